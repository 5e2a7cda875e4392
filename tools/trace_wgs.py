"""Start / end stamps of every workgroup of one fused order-exact sum (xs_fused, xsum.hip; cgh_xsum_set_wgtrace):
how many workgroups run at a time, how long one lives, when the last summary is in and when the driver is done.
    python tools/trace_wgs.py [n] [dot|rr]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
L.cgh_xsum_set_wgtrace.argtypes = [C.c_void_p]
L.cgh_xsum_set_wgtrace.restype = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 ** 3
what = sys.argv[2] if len(sys.argv) > 2 else "dot"
nwg = 4096
tr = cga.DeviceVector(4 * nwg); tr.zero()
if what == "rr":
    # the launch of the sums of r of a strict CG iteration (r.r, |r|_1 and the riding x += alpha p): the last fused launch
    # of a one-iteration solve of the 256^3 system
    N = round(n ** (1.0 / 3.0))
    bb = cga.DeviceVector(n); xx = cga.DeviceVector(n)
    A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, cga.LAYOUT_ELL, bb)
    cga.cg_solve(A, bb, xx, fixed_iters=3, strict=1)
    L.cgh_xsum_set_wgtrace(tr.ptr)
    cga.cg_solve(A, bb, xx, fixed_iters=1, strict=1)
    L.cgh_xsum_set_wgtrace(None)
else:
    rng = np.random.default_rng(1)
    a = cga.DeviceVector.from_host(rng.random(n) + 0.5)
    b = cga.DeviceVector.from_host(rng.random(n) + 0.5)
    out = C.c_double()
    for _ in range(3):
        cga.check(L.cgh_dot(a.ptr, b.ptr, n, 1, C.byref(out)))
    L.cgh_xsum_set_wgtrace(tr.ptr)
    cga.check(L.cgh_dot(a.ptr, b.ptr, n, 1, C.byref(out)))
    L.cgh_xsum_set_wgtrace(None)
t = tr.to_host().view(np.int64).reshape(nwg, 4)
used = t[:, 0] != 0
st, en = t[used, 0].astype(np.float64), t[used, 1].astype(np.float64)
t0 = st.min()
st = (st - t0) / 100.0; en = (en - t0) / 100.0
print("n=%d: %d workgroups; driver (workgroup 0) lives %.1f .. %.1f us" % (n, used.sum(), st[0], en[0]))
nd = 2 if what == "rr" else 1
role = t[used, 2][nd:]
s, e = st[nd:], en[nd:]
if what == "rr":
    xs_, xe_ = s[role == 2], e[role == 2]
    print("x-update workgroups: %d, start %.1f .. %.1f, end %.1f .. %.1f, lifetime median %.1f max %.1f us; second driver ends %.1f"
          % (xs_.size, xs_.min(), xs_.max(), xe_.min(), xe_.max(), np.median(xe_ - xs_), (xe_ - xs_).max(), en[1]))
    s, e = s[role != 2], e[role != 2]
dur = e - s
print("summarising workgroups: start %.1f .. %.1f us, end %.1f .. %.1f us; lifetime min %.1f median %.1f max %.1f us"
      % (s.min(), s.max(), e.min(), e.max(), dur.min(), np.median(dur), dur.max()))
for lo in range(0, int(e.max()) + 5, 5):
    live = int(((s <= lo) & (e > lo)).sum())
    done = int((e <= lo).sum())
    print("t=%3d us: %4d running, %4d finished" % (lo, live, done))
for k in (0, 63, 64, 767, 768, 1535, 2047):
    if k < s.size:
        print("group %4d: %.1f .. %.1f" % (k, s[k], e[k]))
