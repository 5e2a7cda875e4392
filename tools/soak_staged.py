"""Random banded ELL matrices of wide widths through the tile-blocked / LDS-staged SpMV (and through the knobs that switch
them off), bit for bit against a plain numpy loop in slot order: python tools/soak_staged.py [trials] [seed]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
kinds = {}
for trial in range(trials):
    K = int(rng.integers(9, 41)) if trial % 3 else int(rng.choice([13, 26]))
    n = 2 * int(rng.integers(1, 12000)) if trial % 5 else 2 * int(rng.integers(1, 400))
    spread = int(rng.integers(K, 7000))
    pool = np.unique(np.concatenate([rng.integers(-spread, spread + 1, 4 * K), np.arange(1, K + 2), -np.arange(1, K + 2)]))
    offs = rng.permutation(pool[pool != 0])[:K]
    i = np.arange(n)
    ec = np.full((K, n), -1, dtype=np.int32)
    ev = np.zeros((K, n))
    for s in range(K):
        c = i + offs[s]
        ok = (c >= 0) & (c < n) & (rng.random(n) < 0.85)
        ec[s, ok] = c[ok]
        ev[s, ok] = rng.standard_normal(int(ok.sum()))
    diag = rng.standard_normal(n) + 4.0
    xh = rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))
    want = 0.0 + diag * xh
    for s in range(K):
        ok = ec[s] >= 0
        t = want + ev[s] * xh[np.where(ok, ec[s], 0)]
        want = np.where(ok, t, want)
    x = cga.DeviceVector.from_host(xh); y = cga.DeviceVector(n)
    for knobs in ({}, {"CGH_ELL_STAGE_TILES": "3"}, {"CGH_ELL_STAGE": "0"}, {"CGH_ELL_GENERIC": "1"}, {"CGH_ELL_BLOCKED": "0"}):
        for k in ("CGH_ELL_STAGE_TILES", "CGH_ELL_STAGE", "CGH_ELL_GENERIC", "CGH_ELL_BLOCKED"):
            os.environ.pop(k, None)
        os.environ.update(knobs)
        cga.check(L.cgh_config_reload())
        A = cga.Matrix.from_ell(n, K, diag, ev.ravel(), ec.ravel())
        A.spmv(x, y)
        got = y.to_host()
        key = (A.dict_mode, A.stage_doubles > 0)
        kinds[key] = kinds.get(key, 0) + 1
        if not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
            bad += 1
            print("MISMATCH trial %d K=%d n=%d spread=%d knobs=%s mode=%d staged=%d: %d rows differ" % (
                trial, K, n, spread, knobs, A.dict_mode, A.stage_doubles, int((got.view(np.uint64) != want.view(np.uint64)).sum())), flush=True)
        A.free()
print("trials %d x 5 variants, mismatches %d; (dict_mode, staged) counts: %s" % (trials, bad, sorted(kinds.items())), flush=True)
sys.exit(1 if bad else 0)
