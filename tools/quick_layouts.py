"""GPU measurement script: SpMV time per layout on the 256^3 7-point matrix, and CG rate at the foam sizes."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import conjugate_gradient_amd as cga
import foamio
cga.init(0)
ms = C.c_double()
N = 256
n = N ** 3
b = cga.DeviceVector(n); y = cga.DeviceVector(n); x = cga.DeviceVector(n)
for name, layout in (("ell", 0), ("csr", 1), ("sell", 2)):
    A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, layout, b)
    alg = 12 * A.nnz_offdiag + 24 * n
    cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 20, C.byref(ms)))
    rep, _ = cga.cg_solve(A, b, x, fixed_iters=100, history=False)
    print("p7_256 %-5s spmv %.4f ms %.2f TB/s | cg %.1f it/s | stored %.2f GB" % (name, ms.value, alg / ms.value / 1e9, 1e5 / rep.solve_ms, A.stored_bytes / 1e9), flush=True)
    A.free()
for tag in ("2000", "8000", "32k"):
    s = foamio.foam(tag)
    nn = s["n"]
    A = cga.Matrix.from_coo(nn, s["diag"], s["upper"], s["rows"], s["cols"], 0)
    bb = cga.DeviceVector.from_host(s["source"]); xx = cga.DeviceVector(nn)
    for strict in (0, 1):
        t = time.perf_counter()
        rep, _ = cga.cg_solve(A, bb, xx, max_iter=1000, tol=1e-6, strict=strict, history=False)
        dt = time.perf_counter() - t
        print("foam %-5s strict=%d ret=%d iters=%d loop %.2f ms (%.1f us/iter, %.0f it/s) wall %.1f ms" % (tag, strict, rep.ret, rep.iterations, rep.solve_ms, rep.solve_ms / rep.iterations * 1e3, rep.iterations / rep.solve_ms * 1e3, dt * 1e3), flush=True)
