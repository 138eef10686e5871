class _Comm:
    rank = 0
    size = 1
    def bcast(self, x, root=0): return x
    def allreduce(self, x, op=None): return x
    def Barrier(self): pass
    def tompi4py(self): return self
COMM_WORLD = _Comm()
COMM_SELF = _Comm()
INT64_T = 'int64'
SUM = 'sum'
MAX = 'max'
