"""us/iteration of the single-GPU throughput loop at 256^3 7-point (or the shape given) under the env's variants."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
shape = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [256, 256, 256]
n = shape[0] * shape[1] * shape[2]
b = cga.DeviceVector(n); x = cga.DeviceVector(n)
A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
sv = cga.Solver(A, 700)
sv.start(b, x, tol=-1.0, strict=0)
sv.iterate(50, 0)
cga.check(L.cgh_sync())
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    sv.iterate(200, 0)
    cga.check(L.cgh_sync())
    best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
print("n=%d %8.1f us/it = %7.1f it/s  [ELL=%s VEC=%s]" % (n, best, 1e6 / best, os.environ.get("CGH_ELL_VARIANT", "-"),
      os.environ.get("CGH_VEC_VARIANT", "-")), flush=True)
