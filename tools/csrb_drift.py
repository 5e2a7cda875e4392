"""How far apart are the waves of the column-blocked SpMV in their sweep over the slabs of x?  (diagnostic build)"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import conjugate_gradient_amd as cga

cga.init(0)
n = 8 * 1024 * 1024
cb = int(sys.argv[1]) if len(sys.argv) > 1 else 17
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
win = int(sys.argv[3]) if len(sys.argv) > 3 else -1
cga.config_csrb(1024, cb, wgs, win, int(sys.argv[4]) if len(sys.argv) > 4 else 0, 0)
b = cga.DeviceVector(n); y = cga.DeviceVector(n)
A = cga.Matrix.powerlaw(n, 2000, 0x5EED0005, cga.LAYOUT_CSRB, b)
nb = (n // 64 + 15) // 16
st = cga.DeviceVector(nb * 34, np.uint64)
L = cga.lib()
L.cgh_csrb_debug_stamps.argtypes = [C.c_void_p]
L.cgh_csrb_debug_stamps.restype = None
A.spmv(b, y)
st.zero()
L.cgh_csrb_debug_stamps(st.ptr)
A.spmv(b, y)
L.cgh_csrb_debug_stamps(None)
raw = st.to_host()
g = raw[nb*33:]
print('gate: looks %d, gates that needed a look %d of %d, blocks that gave up %d' % ((g >> np.uint64(32)).sum(), ((g & np.uint64(0xffffffff)) >> np.uint64(1)).sum(), nb * ((n + (1 << cb) - 1) >> cb), (g & np.uint64(1)).sum()))
t = raw[:nb*33].reshape(nb, 33).astype(np.int64)
t0 = t[t > 0].min()
t = (t - t0) / 100.0   # us
nslabs = (n + (1 << cb) - 1) >> cb
marks = [k for k in range(32) if 8 * k < nslabs]
print("blocks %d, slabs %d; columns: enter slab 8k (us since first stamp): min / median / max over blocks" % (nb, nslabs))
start = t[:, 0]
first_round = start < np.median(start) + 20
for k in marks + [32]:
    for name, sel in (("round1", first_round), ("round2", ~first_round)):
        c = t[sel, k]
        print("  mark %2d %s  n=%d  min %8.1f  p10 %8.1f  med %8.1f  p90 %8.1f  max %8.1f" % (k, name, c.size, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
# entries per block
wg = np.arange(nb) // 4
grid = 256 * 4 if wgs == 0 else 256 * wgs
for k in (1, 32):
    print("mark %d by group (blockIdx %% 8), round 1: median us / max us / gave up" % k)
    for grp in range(8):
        sel = first_round & ((wg % grid) % 8 == grp)
        print("   group %d: n=%d  med %8.1f  max %8.1f   gave up %d  looks %d" % (grp, sel.sum(), np.median(t[sel, k]), t[sel, k].max(), (g[sel] & np.uint64(1)).sum(), (g[sel] >> np.uint64(32)).sum()))
