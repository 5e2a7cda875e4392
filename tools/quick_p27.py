"""27-point 256^3 SpMV alone (no CG loop) under the env's kernel variants and layouts: ms and algorithmic TB/s."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
dims = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else [256, 256, 256]   # nx ny nz
n = dims[0] * dims[1] * dims[2]
layout = int(os.environ.get("LAYOUT", str(cga.LAYOUT_ELL)))
b = cga.DeviceVector(n); y = cga.DeviceVector(n)
points = int(os.environ.get("POINTS", "27"))
A = cga.Matrix.poisson(dims[0], dims[1], dims[2], points, 0x5EED0001, layout, b)
alg = 12 * A.nnz_offdiag + 24 * n
ms = C.c_double(); best = 1e9
for _ in range(3):
    cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 10, C.byref(ms))); best = min(best, ms.value)
print("layout=%d generic=%s variant=%s dict=%s wgs=%s %d-point %dx%dx%d spmv %.4f ms %.2f TB/s  %.2f ps/row" % (
    layout, os.environ.get("CGH_ELL_GENERIC"), os.environ.get("CGH_ELL_VARIANT"), os.environ.get("CGH_ELL_DICT"),
    os.environ.get("CGH_ELL_WGS"), points, dims[0], dims[1], dims[2], best, alg / best / 1e9, best * 1e9 / n), flush=True)
