import ctypes as C, os
import sys; sys.path.insert(0, ".")
import conjugate_gradient_amd as cga
cga.init(0)
ms = C.c_double()
N = 256; n = N ** 3
b = cga.DeviceVector(n); y = cga.DeviceVector(n)
A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, 1, b)
alg = 12 * A.nnz_offdiag + 24 * n
cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 20, C.byref(ms)))
print("window=%s p7_256 csr spmv %.4f ms %.2f TB/s" % (os.environ.get("CGH_CSR_WINDOW"), ms.value, alg / ms.value / 1e9), flush=True)
A.free()
n2 = 8 * 1024 * 1024
b2 = cga.DeviceVector(n2); y2 = cga.DeviceVector(n2)
A = cga.Matrix.powerlaw(n2, 2000, 0x5EED0005, 1, b2)
alg = 12 * A.nnz_offdiag + 24 * n2
cga.check(cga.lib().cgh_time_spmv(A.h, b2.ptr, y2.ptr, 20, C.byref(ms)))
print("window=%s powerlaw csr spmv %.4f ms %.2f TB/s" % (os.environ.get("CGH_CSR_WINDOW"), ms.value, alg / ms.value / 1e9), flush=True)
