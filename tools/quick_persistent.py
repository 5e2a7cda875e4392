"""Small systems, throughput mode: the resident launch (csrc/cg_persistent.h) against three launches per iteration --
same bits? us per iteration of each?   python tools/quick_persistent.py [nx ny nz]..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
args = [int(v) for v in sys.argv[1:]]
shapes = [args[i:i + 3] for i in range(0, len(args), 3)] or [[64, 64, 31], [64, 64, 32], [32, 32, 32], [20, 10, 10], [8, 8, 4]]
ITERS = 300


def run(A, b, n, on, iters, reps):
    cga.check(L.cgh_config_persistent(on))
    x = cga.DeviceVector(n)
    sv = cga.Solver(A, iters * (reps + 1) + 2)
    sv.start(b, x, tol=-1.0, strict=0)
    sv.iterate(iters, 0)
    cga.check(L.cgh_sync())
    hist = sv.history(0, iters + 1).copy()
    xs = x.to_host().copy()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        sv.iterate(iters, 0)
        cga.check(L.cgh_sync())
        best = min(best, (time.perf_counter() - t0) / iters * 1e6)
    d, q = sv.poll()
    return hist, xs, best


for shape in shapes:
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    h0, x0, t0 = run(A, b, n, 0, ITERS, 3)
    h1, x1, t1 = run(A, b, n, 1, ITERS, 3)
    same = np.array_equal(h0.view(np.uint64), h1.view(np.uint64)) and np.array_equal(x0.view(np.uint64), x1.view(np.uint64))
    bad = -1
    if not same:
        neq = np.nonzero((h0.view(np.uint64) != h1.view(np.uint64)).any(axis=1))[0]
        bad = int(neq[0]) if len(neq) else -2
    # where the time goes: one traced launch of ITERS iterations
    tr = cga.DeviceVector(256 * 8)   # 8-byte slots
    cga.check(L.cgh_persistent_set_trace(tr.ptr))
    cga.check(L.cgh_config_persistent(1))
    xt = cga.DeviceVector(n)
    svt = cga.Solver(A, ITERS + 2)
    svt.start(b, xt, tol=-1.0, strict=0)
    svt.iterate(ITERS, 0)
    cga.check(L.cgh_sync())
    cga.check(L.cgh_persistent_set_trace(None))
    t = tr.to_host().view(np.int64).reshape(256, 8)[:(n + 511) // 512, :5] / ITERS * 0.01   # us per iteration
    names = ["spmv+tile sum", "wait p.Ap", "r update+tile sums", "wait r.r + r gather", "sums, p update"]
    print("   per iteration, us (median / max over workgroups): " + "; ".join(
        "%s %.2f / %.2f" % (nm, float(np.median(t[:, k])), float(t[:, k].max())) for k, nm in enumerate(names)), flush=True)
    print("n=%7d (%s)  three launches %6.2f us/it   resident %6.2f us/it   same bits: %s%s   res[-1]=%r" % (
        n, "x".join(map(str, shape)), t0, t1, same, "" if same else " (first differing entry %d)" % bad, float(h1[-1, 0])), flush=True)
