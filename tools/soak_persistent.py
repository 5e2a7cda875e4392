"""Soak of the resident launch (csrc/cg_persistent.h): many solves with fresh right-hand sides on one solver object and on
fresh ones, every history and solution compared bit for bit with the three-launch path.  python tools/soak_persistent.py [solves]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
solves = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(7)
bad = 0
t_start = time.time()
for shape, iters in (((64, 64, 31), 60), ((40, 20, 10), 90), ((257, 2, 1), 50), ((64, 64, 8), 70)):
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    x = cga.DeviceVector(n)
    sv = [cga.Solver(A, iters + 2), cga.Solver(A, iters + 2)]
    for k in range(solves):
        bh = rng.standard_normal(n)
        bb = cga.DeviceVector.from_host(bh)
        outs = []
        for on in (0, 1):
            cga.check(L.cgh_config_persistent(on))
            s = sv[on]
            s.start(bb, x, tol=-1.0, strict=0)
            # uneven batches: the launch is entered and left several times per solve
            left = iters
            for c in (1, 7, 19):
                s.iterate(c, 0)
                left -= c
            s.iterate(left, 0)
            d, q = s.poll()
            outs.append((s.history(0, iters + 1).copy(), x.to_host().copy()))
        same = np.array_equal(outs[0][0].view(np.uint64), outs[1][0].view(np.uint64)) and np.array_equal(
            outs[0][1].view(np.uint64), outs[1][1].view(np.uint64))
        if not same or not np.isfinite(outs[1][0][1:, 0]).all():
            bad += 1
            print("MISMATCH shape", shape, "solve", k, flush=True)
    print("shape %s: %d solves x %d iterations done, mismatches so far %d (%.0f s)" % (shape, solves, iters, bad, time.time() - t_start), flush=True)
print("soak: %s" % ("OK" if bad == 0 else "%d MISMATCHES" % bad))
sys.exit(1 if bad else 0)
