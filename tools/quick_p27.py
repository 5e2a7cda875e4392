import sys; sys.path.insert(0, ".")
import ctypes as C, os
import conjugate_gradient_amd as cga
cga.init(0)
N = 256; n = N ** 3
b = cga.DeviceVector(n); y = cga.DeviceVector(n)
A = cga.Matrix.poisson(N, N, N, 27, 0x5EED0001, 0, b)
alg = 12 * A.nnz_offdiag + 24 * n
ms = C.c_double(); best = 1e9
for _ in range(3):
    cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 10, C.byref(ms))); best = min(best, ms.value)
print("generic=%s variant=%s p27_256 spmv %.4f ms %.2f TB/s" % (os.environ.get("CGH_ELL_GENERIC"), os.environ.get("CGH_ELL_VARIANT"), best, alg / best / 1e9), flush=True)
