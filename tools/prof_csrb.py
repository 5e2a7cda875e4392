"""One layout of config 5 under the profiler: builds the power-law matrix and runs the SpMV a few times.
usage: python3 tools/prof_csrb.py layout [rows] [rpw slab_bits wgs]   (layout: 1 CSR, 3 CSRB, 4 SELLS)"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import conjugate_gradient_amd as cga

cga.init(0)
layout = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8 * 1024 * 1024
if len(sys.argv) > 5:
    cga.config_csrb(*[int(t) for t in sys.argv[3:9]])
b = cga.DeviceVector(n)
y = cga.DeviceVector(n)
A = cga.Matrix.powerlaw(n, 2000, 0x5EED0005, layout, b)
ms = C.c_double()
cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 10, C.byref(ms)))
alg = 12 * A.nnz_offdiag + 24 * n
print("layout %d: spmv %.4f ms, %.2f TB/s algorithmic, algorithmic bytes %d" % (layout, ms.value, alg / ms.value / 1e9, alg))
# calibration stream for FETCH_SIZE at these access widths: 64M x (8 + 4) bytes = 805306368 bytes per launch
cnt = 64 * 1024 * 1024
s8 = cga.DeviceVector(cnt, np.uint64); s4 = cga.DeviceVector(cnt, np.uint32); sink = cga.DeviceVector(2048, np.uint32)
cga.lib().cgh_time_read_8_4(s8.ptr, s4.ptr, sink.ptr, cnt, 5, C.byref(ms))
print("calibration read8_4: %d bytes per launch, %.4f ms, %.2f TB/s" % (12 * cnt, ms.value, 12 * cnt / ms.value / 1e9))
