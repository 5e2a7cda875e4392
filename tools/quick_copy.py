"""The device copy yardstick in its variants (cgh_time_copy_variant): loads in flight per lane, cache hints, workgroups per CU.
TB/s counted as bytes read + bytes written; 1 GiB buffers (far beyond the 256 MiB Infinity Cache)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
nbytes = 1 << 30
src = cga.DeviceVector(nbytes // 8); dst = cga.DeviceVector(nbytes // 8)
src.zero(); dst.zero()
ms = C.c_double()
for un in (1, 2, 4, 8):
    for hints in ((0, 0), (1, 0), (0, 1), (1, 1)) if un > 2 else ((0, 0), (1, 1)):
        for wgs in (4, 8, 16, 32, 64):
            v = un + 16 * hints[0] + 32 * hints[1] + 256 * wgs
            cga.check(L.cgh_time_copy_variant(dst.ptr, src.ptr, nbytes, 10, v, C.byref(ms)))
            print("copy: %d loads in flight, nt loads %d, nt stores %d, %2d workgroups/CU: %.4f ms = %.0f GB/s (read + written)"
                  % (un, hints[0], hints[1], wgs, ms.value, 2 * nbytes / ms.value / 1e6), flush=True)
sink = cga.DeviceVector(1024)
cga.check(L.cgh_time_read(src.ptr, sink.ptr, nbytes, 10, C.byref(ms)))
print("read only: %.4f ms = %.0f GB/s" % (ms.value, nbytes / ms.value / 1e6))
