"""GPU tuning script: 256^3 7-point ELL SpMV / CG timing (env CGH_ELL_VARIANT selects a kernel variant)."""
import sys; sys.path.insert(0, ".")
import sys, os
import ctypes as C
import conjugate_gradient_amd as cga
cga.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = cga.DeviceVector(N**3)
A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
x = cga.DeviceVector(A.n); y = cga.DeviceVector(A.n)
ms = C.c_double()
alg = 12 * A.nnz_offdiag + 24 * A.n
best = 1e9
for rep in range(3):
    cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 30, C.byref(ms)))
    best = min(best, ms.value)
rep, hist = cga.cg_solve(A, b, x, fixed_iters=200, time_kernels=0, history=False)
print("variant=%s spmv %.4f ms %.2f TB/s (%.1f%%) | cg %.4f ms/iter %.1f it/s" % (os.environ.get("CGH_ELL_VARIANT", "dflt"), best, alg / best / 1e9, alg / best / 1e9 / 80, rep.solve_ms / 200, 2e5 / rep.solve_ms), flush=True)
