"""GPU tests of the multi-GPU path on ONE device: (1) G ranks in one process (cgh_dist_group_*: same kernels,
index lists and buffers as the RCCL loop, exchange/reduction by device copies) vs the reference goldens;
(2) the z-slab device generator vs the host partition of the same global matrix (bit-exact);
(3) the RCCL-driven loop itself with a 1-rank communicator."""
import numpy as np
import pytest

import foamio
from gpu_common import bits, golden, inputs, same_bits_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


def _run_group(cga, ranks, bs, n_locs, g, check=8):
    xs = [cga.DeviceVector(nl) for nl in n_locs]
    mi, tol = int(g["max_iter"]), float(g["tol"])
    cga.group_start(ranks, bs, xs, tol, mi)
    done, queued = 0, 0
    while queued < mi - 1 and not done:
        step = min(check, mi - 1 - queued)
        cga.group_iterate(ranks, step)
        queued += step
        # all ranks carry the same flag (same reduced scalars): read it through rank 0's history
        hist = ranks[0].history(0, queued + 1)
        conv = np.nonzero(hist[1:, 0] < tol)[0]
        done = int(conv[0]) + 1 if conv.size else 0
    iters = done if done else queued
    hists = [r.history(0, iters + 1) for r in ranks]
    for h in hists[1:]:
        assert np.array_equal(bits(np.nan_to_num(h)), bits(np.nan_to_num(hists[0])))   # ranks agree bit for bit
    x = np.concatenate([v.to_host() for v in xs])
    return (done + 1 if done else -1), hists[0], x


def _check_against_golden(ret, hist, x, g):
    want = int(g["cg_ret"][4])
    assert (ret == -1 and want == -1) or abs(ret - want) <= 1
    gt = g["traj"][:, 0]
    m = min(gt.size, hist.shape[0])
    rel = np.abs(hist[:m, 0] - gt[:m]) / gt[:m]
    assert rel[:11].max() < 1e-10
    if want > 0:
        assert np.abs(x - g["cg_x"]).max() <= 1e-5 * max(1.0, np.abs(g["cg_x"]).max())


@pytest.mark.parametrize("G", [2, 4, 8])
@pytest.mark.parametrize("tag", ["cylinder_2000", "poisson27_12x10x8", "poisson7_16"])
def test_rank_group_from_partition(cga, tag, G):
    s, g = inputs(tag), golden(tag)
    parts = [cga.Partition(s["n"], s["diag"], s["upper"], s["rows"], s["cols"], G, r) for r in range(G)]
    ranks = [cga.DistRank.from_partition(p) for p in parts]
    bs = [cga.DeviceVector.from_host(s["source"][p.lo:p.lo + p.n_loc]) for p in parts]
    ret, hist, x = _run_group(cga, ranks, bs, [p.n_loc for p in parts], g)
    _check_against_golden(ret, hist, x, g)


@pytest.mark.parametrize("shape,G", [((12, 10, 8, 27), 2), ((12, 10, 8, 27), 3), ((16, 16, 16, 7), 4), ((5, 3, 8, 27), 8),
                                     ((4, 4, 9, 7), 2)])
def test_slab_generator_matches_host_partition(cga, shape, G):
    nx, ny, nz, pts = shape
    seed = 0x5EED0001
    s = foamio.poisson_coo(nx, ny, nz, seed, pts)
    nxy = nx * ny
    starts = np.array([nxy * (nz * r // G) for r in range(G + 1)], dtype=np.int64)
    for r in range(G):
        P = cga.Partition(s["n"], s["diag"], s["upper"], s["rows"], s["cols"], G, r, starts)
        b = cga.DeviceVector(P.n_loc)
        D = cga.DistRank.poisson(nx, ny, nz, pts, seed, G, r, b)
        assert (D.n_loc, D.n_halo) == (P.n_loc, P.n_halo)
        assert D.nnz_offdiag == int((P.cols >= 0).sum())
        diag, ev, ec = D.download_local()
        K = D.K
        assert K >= P.K
        assert np.array_equal(bits(diag), bits(P.diag))
        assert np.array_equal(ec.reshape(K, -1)[:P.K], P.cols.reshape(P.K, -1)) and np.all(ec.reshape(K, -1)[P.K:] == -1)
        assert np.array_equal(bits(ev.reshape(K, -1)[:P.K]), bits(P.vals.reshape(P.K, -1)))
        assert np.array_equal(bits(b.to_host()), bits(s["source"][P.lo:P.lo + P.n_loc]))


@pytest.mark.parametrize("G", [2, 4, 5])
def test_rank_group_slab_poisson(cga, G):
    g = golden("poisson7_32x24x20")
    bs, ranks = [], []
    for r in range(G):
        nl = 32 * 24 * (20 * (r + 1) // G - 20 * r // G)
        b = cga.DeviceVector(nl)
        ranks.append(cga.DistRank.poisson(32, 24, 20, 7, 0x5EED0001, G, r, b))
        bs.append(b)
    ret, hist, x = _run_group(cga, ranks, bs, [r.n_loc for r in ranks], g)
    _check_against_golden(ret, hist, x, g)


def test_rccl_loop_single_rank(cga):
    """the RCCL-driven loop (streams, events, ncclAllReduce) with a 1-rank communicator"""
    g = golden("poisson7_32x24x20")
    cga.comm_init(1, 0, cga.comm_unique_id())
    n = 32 * 24 * 20
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    D = cga.DistRank.poisson(32, 24, 20, 7, 0x5EED0001, 1, 0, b)
    mi, tol = int(g["max_iter"]), float(g["tol"])
    D.start(b, x, tol, mi)
    done, queued = 0, 0
    while queued < mi - 1 and not done:
        step = min(16, mi - 1 - queued)
        D.iterate(step, 1)
        done, queued = D.poll()
    assert done > 0 and D.spmv_ms() > 0
    hist = D.history(0, done + 1)
    _check_against_golden(done + 1, hist, x.to_host(), g)
    # the launcher helper bench_dist.py uses for its barriers / max-over-ranks
    assert cga.comm_allreduce([1.5, -2.0, 7.0], "sum").tolist() == [1.5, -2.0, 7.0]
    assert cga.comm_allreduce([3.25], "max").tolist() == [3.25] and cga.comm_allreduce([3.25], "min").tolist() == [3.25]
    cga.lib().cgh_comm_destroy()
    with pytest.raises(RuntimeError):
        cga.comm_allreduce([1.0])


@pytest.mark.parametrize("mode", ["normal", "mute"])
def test_p2p_allreduce_two_processes_one_gpu(tmp_path, mode):
    """the peer-to-peer scalar all-reduce (p2p.hip) between two PROCESSES: hipIpc mailboxes, tagged 64-bit words, the
    waits inside the kernel.  Both ranks share the one GPU of the test box (RCCL refuses that, peer stores do not care);
    400 exchanges per rank with deliberate skew.  "mute": rank 1 connects and stays silent -- rank 0's wait must end
    after its 10 s limit with an error instead of hanging, and later exchanges must not wait again."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "p2p_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(tmp_path), mode], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-2000:]


@pytest.mark.parametrize("G,shape,points", [(2, (32, 24, 20), 7), (4, (32, 24, 20), 7), (8, (32, 24, 20), 7),
                                            (3, (12, 10, 8), 27), (8, (12, 10, 8), 27)])
def test_multi_process_loop_on_one_gpu(tmp_path, G, shape, points):
    """The distributed CG loop with G PROCESSES (one rank each, all on the test box's single GPU): peer-to-peer scalar
    all-reduces and peer-to-peer halo exchange between real processes, host threads running at their own pace, the last
    rank deliberately late.  Every rank's history equals every other's bit for bit and the golden trajectory of the
    reference within 1e-10; the gathered x equals the golden solution within the solver tolerance."""
    import os
    import subprocess
    import sys
    tag = "poisson7_32x24x20" if points == 7 else "poisson27_12x10x8"
    g = golden(tag)
    mi, tol = int(g["max_iter"]), float(g["tol"])
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    args = [str(v) for v in (*shape, points, mi, tol)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(G), str(tmp_path)] + args, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(G)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=170)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-3000:]
    hists = [np.load(os.path.join(tmp_path, "hist_%d.npy" % r)) for r in range(G)]
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_%d.npy" % r)) for r in range(G)])
    for h in hists[1:]:
        assert same_bits_nan(h, hists[0])
    _check_against_golden(hists[0].shape[0], hists[0], xs, g)


@pytest.mark.parametrize("tag,G,env", [("cylinder_2000", 3, {}), ("cylinder_2000", 4, {"CGH_FUSED_PUSH": "0"}),
                                       ("cylinder_8000", 5, {}), ("poisson27_12x10x8", 4, {"CGH_DIST_LL_SPMV": "0"}),
                                       ("poisson27_12x10x8", 2, {"CGH_DIST_LL_SPMV": "1", "CGH_DIST_ONE_SPMV": "0"}),
                                       ("poisson27_12x10x8", 4, {"CGH_DIST_LL_SPMV": "1"}),   # 27-point: unpacked by default
                                       ("poisson7_16", 4, {"CGH_FUSED_PUSH": "0", "CGH_DIST_ONE_SPMV": "0"}),
                                       ("poisson7_16", 2, {"CGH_LLC_MB": "0"}),          # the streaming (non-temporal) instantiations
                                       ("poisson27_12x10x8", 3, {"CGH_LLC_MB": "0", "CGH_ELL_DICT": "0"})])
def test_multi_process_loop_from_partition(tmp_path, tag, G, env):
    """The same between processes for ROW BLOCKS OF A STORED SYSTEM (cgh_dist_from_partition): send lists that are not
    contiguous (stand-alone push kernel with index lists), local sizes the vector SpMV cannot take (deposit unpacked
    into the halo tail of p first), and the knobs that switch the fused pieces off one at a time.  Every rank's history
    identical; the golden trajectory within 1e-10; x within the solver tolerance."""
    import os
    import subprocess
    import sys
    g = golden(tag)
    mi, tol = int(g["max_iter"]), float(g["tol"])
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(G), str(tmp_path), "tag", tag, "0", "0", str(mi), str(tol)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **env)) for r in range(G)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=170)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-3000:]
    hists = [np.load(os.path.join(tmp_path, "hist_%d.npy" % r)) for r in range(G)]
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_%d.npy" % r)) for r in range(G)])
    for h in hists[1:]:
        assert same_bits_nan(h, hists[0])
    _check_against_golden(hists[0].shape[0], hists[0], xs, g)


@pytest.mark.parametrize("tag,G,composed", [("cylinder_2000", 2, 1), ("cylinder_8000", 4, 1), ("cylinder_32k", 2, 1), ("poisson7_16", 2, 1),
                                            ("poisson27_12x10x8", 4, 1), ("poisson7_32x24x20", 8, 1), ("cylinder_2000", 8, 1),
                                            ("poisson7_32x24x20", 3, 1), ("cylinder_32k", 8, 1),
                                            ("cylinder_8000", 3, 0), ("poisson7_32x24x20", 4, 0)])
def test_multi_process_strict_mode_is_bit_identical(tmp_path, tag, G, composed):
    """cgh_dist_set_strict: with 2, 3, 4 and 8 PROCESSES the whole trajectory (residual, r.r, alpha, beta of every
    iteration), the converging iteration and the gathered x equal the reference's goldens of the UNdistributed system bit
    for bit -- the OpenFOAM systems included, where the throughput-mode loop only agrees to rounding.  The order-exact
    sums are COMPOSED from the ranks' batch summaries (every rank summarises its part at once, every rank walks all of
    them: xsum.hip, xc_walk); composed = 0 (CGH_DIST_STRICT_COMPOSED=0): chained rank by rank as in round 2."""
    import os
    import subprocess
    import sys
    g = golden(tag)
    mi, tol = int(g["max_iter"]), float(g["tol"])
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(G), str(tmp_path), "tag", tag, "0", "0", str(mi), str(tol), "strict"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, CGH_DIST_STRICT_COMPOSED=str(composed))) for r in range(G)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=170)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-3000:]
    hists = [np.load(os.path.join(tmp_path, "hist_%d.npy" % r)) for r in range(G)]
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_%d.npy" % r)) for r in range(G)])
    for h in hists[1:]:
        assert same_bits_nan(h, hists[0])
    want = g["traj"]
    assert hists[0].shape[0] == want.shape[0], (hists[0].shape, want.shape)
    assert same_bits_nan(hists[0], want)
    assert np.array_equal(bits(xs), bits(g["cg_x"]))


def test_multi_process_loop_silent_neighbour(tmp_path):
    """rank 1 connects and then says nothing: rank 0's waits inside the SpMV (halo entries) and inside the vector
    kernels (reduction messages) end after their 10 s limit, the results are NaN, cgh_dist_poll reports it, and work
    queued afterwards does not wait again"""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    args = [str(v) for v in (32, 24, 20, 7, 40, 1e-9)] + ["mute"]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(tmp_path)] + args, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=170)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-3000:]


def test_multi_process_loop_larger_halo(tmp_path, cga):
    """4 processes on one GPU, 128x128x64 7-point (1M rows, halo planes of 16384 entries, 50 iterations): every rank's
    history identical, and equal to the single-GPU throughput-mode trajectory of the same matrix within rounding"""
    import os
    import subprocess
    import sys
    G, shape, iters = 4, (128, 128, 64), 50
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    args = [str(v) for v in (*shape, 7, iters + 1, -1.0)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(G), str(tmp_path)] + args, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(G)]
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    rep, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=0)
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=170)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-3000:]
    hists = [np.load(os.path.join(tmp_path, "hist_%d.npy" % r)) for r in range(G)]
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_%d.npy" % r)) for r in range(G)])
    for h in hists[1:]:
        assert same_bits_nan(h, hists[0])
    assert hists[0].shape[0] == iters + 1
    rel = np.abs(hists[0][:, :2] - hist[:, :2]) / np.abs(hist[:, :2])
    assert rel.max() < 1e-9
    xr = x.to_host()
    assert np.abs(xs - xr).max() <= 1e-9 * np.abs(xr).max()


@pytest.mark.parametrize("points,shape", [(7, (16, 12, 4)), (27, (10, 8, 3)), (7, (64, 64, 6))])
def test_loopback_harness_is_the_periodic_slab(cga, points, shape):
    """cgh_dist_loopback (the one-GPU timing harness of the multi-GPU iteration): an inner slab whose two neighbours are
    the rank itself solves the slab's operator closed periodically in z.  Same kernels, flags and deposits as between
    processes; checked against a numpy CG on that periodic matrix (1e-10 on the residual trajectory)."""
    import ctypes as C
    L = cga.lib()
    buf = C.create_string_buffer(64)
    cga.check(L.cgh_p2p_local_handle(buf))
    cga.check(L.cgh_p2p_connect(1, 0, buf.raw))
    cga.check(L.cgh_comm_init_local(1, 0))
    try:
        nx, ny, nzl = shape
        nxy, nl = nx * ny, nx * ny * nzl
        b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
        D = cga.DistRank.poisson(nx, ny, 8 * nzl, points, 0x5EED0001, 8, 3, b)
        diag, ev, ec = D.download_local()
        K = D.K
        D.loopback()
        assert L.cgh_dist_halo_is_p2p(D.h) == 1 and D.n_halo == 2 * nxy
        iters = 30
        D.start(b, x, -1.0, iters + 1)
        D.iterate(iters, 0)
        done, queued = D.poll()
        assert done == 0 and queued == iters
        hist = D.history(0, iters + 1)
        D.free()
        # halo column nl+h: h < nxy is the lower neighbour's top plane = my own top plane, the rest my own bottom plane
        cols = ec.astype(np.int64).reshape(K, nl)
        lo = (cols >= nl) & (cols < nl + nxy)
        hi = cols >= nl + nxy
        cols = np.where(lo, cols - nl + (nl - nxy), np.where(hi, cols - nl - nxy, cols))
        vals = ev.reshape(K, nl)
        bh = b.to_host()

        def mv(v):
            return diag * v + (vals * v[cols]).sum(axis=0)
        xx = np.zeros(nl); r = bh - mv(xx); p = r.copy(); rr = r @ r
        want = [np.abs(r).sum() / 2]
        for _ in range(iters):
            Ap = mv(p); alpha = rr / (p @ Ap)
            xx += alpha * p; r -= alpha * Ap
            rn = r @ r; want.append(np.abs(r).sum() / 2)
            p = r + (rn / rr) * p; rr = rn
        want = np.array(want)
        assert np.all(np.isfinite(hist[:, 0])) and want[-1] < want[0]
        assert np.all(np.abs(hist[:, 0] - want) <= 1e-10 * want + 1e-15 * want[0])   # (the trajectory spans 12 decades)
        assert np.max(np.abs(x.to_host() - xx)) <= 1e-10 * np.max(np.abs(xx))
    finally:
        L.cgh_comm_destroy()
