"""Workgroup time stamps of one two-phase SpMV on config 5 (cgh_pb_set_trace, spmv_pb.hip): when the workgroups of
pb_expand / pb_accumulate start and end, how long one lives per entry, how long the tail of each launch is.
    python tools/trace_pb.py [rows]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
L.cgh_pb_set_trace.argtypes = [C.c_void_p]
L.cgh_pb_set_trace.restype = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 1024 * 1024
b = cga.DeviceVector(n)
A = cga.Matrix.powerlaw(n, 2000, 0x5EED0005, cga.LAYOUT_PB, b)
x = cga.DeviceVector.from_host(np.random.default_rng(1).standard_normal(n))
y = cga.DeviceVector(n)
for _ in range(3):
    A.spmv(x, y)
tr = cga.DeviceVector(8 * 4096)
tr.zero()
L.cgh_pb_set_trace(tr.ptr)
A.spmv(x, y)
L.cgh_pb_set_trace(None)
y.to_host()
t = tr.to_host().view(np.int64).reshape(2, 4096, 4)
t0 = None
for ph, name in ((0, "pb_expand"), (1, "pb_accumulate")):
    u = t[ph]
    used = u[:, 0] != 0
    st, mid, en, ent = (u[used, k].astype(np.float64) for k in range(4))
    if t0 is None:
        t0 = st.min()
    st, mid, en = (st - t0) / 100.0, (mid - t0) / 100.0, (en - t0) / 100.0
    dur = en - st
    print("%s: %d workgroups, entries per workgroup %d .. %d (mean %d)" % (name, used.sum(), ent.min(), ent.max(), ent.mean()))
    print("  first start %.1f us, last start %.1f, first end %.1f, last end %.1f  (launch lives %.1f us)" % (
        st.min(), st.max(), en.min(), en.max(), en.max() - st.min()))
    print("  lifetime min %.1f median %.1f max %.1f us; ns per 1000 entries: median %.1f (first 256: %.1f, later: %.1f)" % (
        dur.min(), np.median(dur), dur.max(), np.median(dur / ent * 1e6), np.median((dur / ent * 1e6)[:256]),
        np.median((dur / ent * 1e6)[256:]) if used.sum() > 256 else float("nan")))
    if ph == 0:
        print("  staging of x and tables: median %.2f us, max %.2f" % (np.median(mid - st), (mid - st).max()))
    else:
        print("  epilogue (y = diag x + sums): median %.2f us, max %.2f" % (np.median(en - mid), (en - mid).max()))
    lo = st.min()
    for q in range(0, int(en.max() - lo) + 20, 20):
        tt = lo + q
        print("    t=%4d us: %3d workgroups live, %3d done" % (q, int(((st <= tt) & (en > tt)).sum()), int((en <= tt).sum())))
