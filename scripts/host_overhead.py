"""Host-side cost of the binding layer, measured without a GPU by timing argument marshalling only (the foreign call is
replaced by a no-op C function with the same signature count).  usage: python scripts/host_overhead.py"""
import ctypes, time, sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
import mt3d_amd
from mt3d_amd.engine import lib as L
from mt3d_amd.engine.lib import RxAct, I3
from ctypes import byref, c_void_p

t = torch.empty((2, 4, 4, 4, 32))
def bench(name, fn, n=20000):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f"{name:38s} {(time.perf_counter() - t0) / n * 1e6:6.2f} us")
bench("RxAct(...) from tensor", lambda: RxAct(t.data_ptr(), *t.shape[:4], 32, 32))
d = RxAct(t.data_ptr(), 2, 4, 4, 4, 32, 32)
bench("byref(desc)", lambda: byref(d))
bench("I3(3,3,3)", lambda: I3(3, 3, 3))
bench("c_void_p(t.data_ptr())", lambda: c_void_p(t.data_ptr()))
bench("torch.empty_like(small)", lambda: torch.empty_like(t))
bench("torch.device('cpu')", lambda: torch.device("cpu"))
libc = ctypes.CDLL(None)
f = libc.getpid
bench("foreign call (0 args)", lambda: f())
g = libc.labs; g.argtypes = [ctypes.c_long]; g.restype = ctypes.c_long
bench("foreign call (1 arg)", lambda: g(3))
