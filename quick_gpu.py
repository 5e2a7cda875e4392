"""first-contact GPU script: build a 256^3 7-point ELL on the device, time SpMV and copy."""
import sys, time
import numpy as np
import conjugate_gradient_amd as cga
cga.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = time.time()
b = cga.DeviceVector(N**3)
A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
cga.lib().cgh_sync()
print("gen %.2fs n=%d K=%d nnz_off=%d" % (time.time() - t, A.n, A.K, A.nnz_offdiag), flush=True)
x = cga.DeviceVector(A.n); y = cga.DeviceVector(A.n)
import ctypes as C
ms = C.c_double()
cga.binding.check(cga.lib().cgh_time_copy(y.ptr, b.ptr, A.n * 8, 20, C.byref(ms)))
print("copy %.3f ms  %.2f TB/s (r+w)" % (ms.value, 2 * A.n * 8 / ms.value / 1e9), flush=True)
cga.binding.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 20, C.byref(ms)))
alg = 12 * A.nnz_offdiag + 24 * A.n
print("spmv %.3f ms  alg %.3f GB -> %.2f TB/s (%.1f%% of 8 TB/s)" % (ms.value, alg / 1e9, alg / ms.value / 1e9, alg / ms.value / 1e9 / 8 * 100), flush=True)
rep, hist = cga.cg_solve(A, b, x, fixed_iters=100, time_kernels=1)
print("cg 100 iters: %.3f ms/iter, spmv avg %.3f ms, res0 %.6g -> %.6g" % (rep.solve_ms / 100, rep.spmv_ms_avg, rep.initial_res, rep.final_res), flush=True)
rep, hist = cga.cg_solve(A, b, x, fixed_iters=100, time_kernels=0)
print("cg 100 iters (no kernel events): %.3f ms/iter -> %.1f iter/s" % (rep.solve_ms / 100, 1e5 / rep.solve_ms), flush=True)
rep, hist = cga.cg_solve(A, b, x, max_iter=1000, tol=1e-6 * hist[0, 0])
print("cg to tol: ret=%d iters=%d final=%.3e ms=%.1f" % (rep.ret, rep.iterations, rep.final_res, rep.solve_ms), flush=True)
