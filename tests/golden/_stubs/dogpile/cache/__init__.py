class _Region:
    def configure(self, *a, **k): return self
    def get(self, key=None): return None
    def set(self, key=None, value=None): pass
    def cache_on_arguments(self, *a, **k): return lambda f: f
def make_region(): return _Region()
