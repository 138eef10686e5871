class NullBackend: pass
