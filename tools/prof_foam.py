"""Strict (bit-identical) vs throughput iteration cost on the OpenFOAM systems the reference ships, with the profile of
the serial walks of the order-exact sums (xsum.hip, XS_PROF_ADD)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conjugate_gradient_amd as cga
import foamio
cga.init(0)
L = cga.lib()
for tag in ("2000", "8000", "32k"):
    s = foamio.foam(tag)
    n = s["n"]
    A = cga.Matrix.from_coo(n, s["diag"], s["upper"], s["rows"], s["cols"], cga.LAYOUT_ELL)
    b = cga.DeviceVector.from_host(s["source"]); x = cga.DeviceVector(n)
    cga.cg_solve(A, b, x, fixed_iters=3, strict=1)
    prof = cga.DeviceVector(16); prof.zero()
    L.cgh_xsum_set_prof(prof.ptr)
    iters = 200
    r, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
    L.cgh_xsum_set_prof(None)
    p = prof.to_host().view(np.int64)
    walks = max(int(p[6]), 1)
    rt, _ = cga.cg_solve(A, b, x, fixed_iters=iters, strict=0)
    print("foam %s n=%d K=%d strict %.1f us/it (throughput %.1f) | per walk (%d): batches descended into %.1f chains %.1f crossings/ties applied whole %.1f | %.1f us in the driver: %.1f staging + joins, %.1f walk" % (
        tag, n, A.K, r.solve_ms / iters * 1e3, rt.solve_ms / iters * 1e3, walks, p[0] / walks, p[2] / walks, p[7] / walks, p[3] / walks / 100.0, p[4] / walks / 100.0, p[5] / walks / 100.0), flush=True)
