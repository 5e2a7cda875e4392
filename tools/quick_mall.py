"""Does the 256 MiB Infinity Cache help when the whole working set fits (2.1M rows: ~200 MB)?  us/iteration of the
single-GPU throughput loop under the env's load/store variants (CGH_VEC_VARIANT, CGH_ELL_VARIANT, CGH_ELL_DICT)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
for shape in ((256, 256, 16), (256, 256, 32), (256, 256, 64), (256, 256, 128)):
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    sv = cga.Solver(A, 1300)
    sv.start(b, x, tol=-1.0, strict=0)
    sv.iterate(300, 0)
    cga.check(L.cgh_sync())
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        sv.iterate(300, 0)
        cga.check(L.cgh_sync())
        best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    print("%s n=%8d %7.1f us/it  [VEC=%s ELL=%s DICT=%s LLC_MB=%s]" % (shape, n, best, os.environ.get("CGH_VEC_VARIANT", "-"),
          os.environ.get("CGH_ELL_VARIANT", "-"), os.environ.get("CGH_ELL_DICT", "-"), os.environ.get("CGH_LLC_MB", "-")), flush=True)
    sv.free()
