"""Timeline of ONE walk of the order-exact sum driver (xsum.hip, XS_TRACE): python tools/trace_drive.py [nx ny nz]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 256, 256)
n = shape[0] * shape[1] * shape[2]
b = cga.DeviceVector(n); x = cga.DeviceVector(n)
A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
cga.cg_solve(A, b, x, fixed_iters=3, strict=1)
prof = cga.DeviceVector(16 + 2 * 256); prof.zero()
L.cgh_xsum_set_prof(prof.ptr)
cga.cg_solve(A, b, x, fixed_iters=10, strict=1)
L.cgh_xsum_set_prof(None)
p = prof.to_host().view(np.int64)
last = 0
for i in range(240):
    code, t = int(p[16 + 2 * i]), int(p[17 + 2 * i])
    if code == 0:
        break
    print("%5d  at %7.2f us  (+%.2f)" % (code, t / 100.0, (t - last) / 100.0))
    last = t
