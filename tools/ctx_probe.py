"""Does the kernel BEFORE an SpMV change the SpMV's time?  Stand-alone launches of the 256^3 SpMV (POINTS=7 | 27), each preceded
by nothing / an axpy that rewrites x / one that rewrites an unrelated vector / one that reads y: median ms per SpMV."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import conjugate_gradient_amd as cga
cga.init(0); L = cga.lib()
pts = int(os.environ.get("POINTS", "7"))
N = 256; n = N**3
b = cga.DeviceVector(n); y = cga.DeviceVector(n); z = cga.DeviceVector(n); w = cga.DeviceVector(n)
A = cga.Matrix.poisson(N, N, N, pts, 0x5EED0001, cga.LAYOUT_ELL, b)
x = b
ms = C.c_double()
L.cgh_axpy.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64]
for _ in range(3): cga.check(L.cgh_time_spmv(A.h, x.ptr, y.ptr, 5, C.byref(ms)))
for mode in ("none", "axpy_x", "axpy_other", "axpy_y", "none"):
    ts = []
    for i in range(40):
        if mode == "axpy_x": cga.check(L.cgh_axpy(x.ptr, z.ptr, 0.0, x.ptr, n))
        elif mode == "axpy_other": cga.check(L.cgh_axpy(w.ptr, z.ptr, 0.0, w.ptr, n))
        elif mode == "axpy_y": cga.check(L.cgh_axpy(y.ptr, z.ptr, 0.0, w.ptr, n))
        cga.check(L.cgh_time_spmv(A.h, x.ptr, y.ptr, 1, C.byref(ms))); ts.append(ms.value)
    print("%d-point preceded by %-11s spmv median %.4f ms  min %.4f" % (pts, mode, float(np.median(ts[5:])), min(ts[5:])), flush=True)
