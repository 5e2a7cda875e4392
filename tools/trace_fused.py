"""Time stamps of the fused order-exact sum (xs_fused, xsum.hip): when the driver saw each batch of 64 groups ready
and when it had walked it, relative to the driver's start.  python tools/trace_fused.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
L.cgh_xsum_set_trace.argtypes = [C.c_void_p]
L.cgh_xsum_set_trace.restype = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 ** 3
rng = np.random.default_rng(1)
a = cga.DeviceVector.from_host(rng.random(n) + 0.5)
b = cga.DeviceVector.from_host(rng.random(n) + 0.5)
out = C.c_double()
for _ in range(3):
    cga.check(L.cgh_dot(a.ptr, b.ptr, n, 1, C.byref(out)))
tr = cga.DeviceVector(224); tr.zero()
L.cgh_xsum_set_trace(tr.ptr)
cga.check(L.cgh_dot(a.ptr, b.ptr, n, 1, C.byref(out)))
L.cgh_xsum_set_trace(None)
t = tr.to_host().view(np.int64)
ncells = (n + 127) // 128
ngroups = (ncells + 63) // 64
nb = ngroups if ncells <= 2048 else (ngroups + 63) // 64   # few cells: a batch is one group of 64 cells
print("n=%d batches=%d  (us after the driver's start)" % (n, nb))
for bb in range(min(nb, 72)):
    print("batch %2d  seen ready %7.2f  walked %7.2f" % (bb, t[8 + bb] / 100.0, t[152 + bb] / 100.0))
