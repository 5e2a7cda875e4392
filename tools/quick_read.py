import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import conjugate_gradient_amd as cga
cga.init(0)
L = cga.lib()
n = 1 << 30
src = cga.DeviceVector(n // 8); sink = cga.DeviceVector(1024)
ms = C.c_double()
for w in (2, 3, 4, 5, 6, 7, 8):
    best = 1e9
    for _ in range(3):
        cga.check(L.cgh_time_read_variant(src.ptr, sink.ptr, n, 10, w, C.byref(ms))); best = min(best, ms.value)
    print("read: %d workgroups per CU: %.4f ms = %.0f GB/s" % (w, best, n / best / 1e6), flush=True)
