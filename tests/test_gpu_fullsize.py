"""BASELINE's big configurations at FULL size under -m gpu:

  config 3  256^3 7-point, row-partitioned over 8 processes      vs the one-GPU strict (= reference-order) trajectory
  config 4  512^3 27-point (3.5e9 stored entries), 8 processes   A*1 exactly through the distributed SpMV (halo rows
                                                                 included), throughput loop vs the one-GPU loop, the
                                                                 bit-identical mode vs the one-GPU strict solve
  config 5  8.4M-row power-law matrix                            CG on the two-phase and the column-blocked layouts
                                                                 (strict = 0) vs the exact-row CSR strict solve

The processes of a multi-rank case share the one GPU of the test box (the peer-to-peer transport does not care; RCCL
refuses it), tests/dist_fullsize_worker.py is one rank.  The reference itself cannot produce outputs at these sizes in a
test's time (config 4: its `int` indexing cannot even address the matrix), so the anchor is the one-GPU strict loop, which
tests/test_gpu_xsum.py pins to the reference's own code bit for bit at 256^3."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_common import bits, same_bits_nan

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cga():
    import conjugate_gradient_amd as m
    m.init(0)
    return m


def _ranks(tmp_path, G, shape, points, fast_iters, strict_iters, check_ones, timeout):
    worker = os.path.join(HERE, "dist_fullsize_worker.py")
    args = [str(v) for v in (*shape, points, fast_iters, strict_iters, check_ones)]
    return [subprocess.Popen([sys.executable, worker, str(r), str(G), str(tmp_path)] + args, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True) for r in range(G)], timeout


def _join(procs_timeout):
    procs, timeout = procs_timeout
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (code, out) in enumerate(outs):
        assert code == 0 and "rank %d ok" % r in out, out[-3000:]
    return [o for _, o in outs]


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


def test_config3_256cubed_as_8_ranks_vs_one_gpu_strict(tmp_path, cga):
    """256^3 7-point over 8 processes, 100 iterations of the throughput loop: every rank's history bit-identical to every
    other's; residual and r.r within 1e-10 of the one-GPU STRICT trajectory (the reference's order); x within 1e-9."""
    G, N, iters, strict_iters = 8, 256, 100, 30
    procs = _ranks(tmp_path, G, (N, N, N), 7, iters, strict_iters, 1, 280)
    n = N ** 3
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(N, N, N, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    rep, want = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
    xr = x.to_host()
    rep, want_s = cga.cg_solve(A, b, x, fixed_iters=strict_iters, strict=1)
    xs1 = x.to_host()
    A.free()
    outs = _join(procs)
    # the bit-identical mode over the 8 ranks (order-exact sums composed from the ranks' batch summaries): history and x
    # equal the one-GPU strict solve bit for bit
    for r in range(G):
        assert same_bits_nan(np.load(os.path.join(tmp_path, "hist_strict_%d.npy" % r)), want_s)
    assert np.array_equal(bits(np.concatenate([np.load(os.path.join(tmp_path, "x_strict_%d.npy" % r)) for r in range(G)])), bits(xs1))
    print("\n[config 3, 8 ranks on one GPU, bit-identical mode] " + " | ".join(l for l in outs[0].strip().splitlines() if "strict:" in l))
    hists = [np.load(os.path.join(tmp_path, "hist_fast_%d.npy" % r)) for r in range(G)]
    for h in hists[1:]:
        assert same_bits_nan(h, hists[0])
    assert hists[0].shape == want.shape == (iters + 1, 4)
    dev = _rel(hists[0][:, :2], want[:, :2])
    print("\n[config 3, 8 ranks on one GPU] max relative deviation of (res, r.r) from the one-GPU strict trajectory over %d iterations: %.3g; %s"
          % (iters, dev, outs[0].strip().splitlines()[0]))
    assert dev <= 1e-10
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_fast_%d.npy" % r)) for r in range(G)])
    assert np.abs(xs - xr).max() <= 1e-9 * np.abs(xr).max()


@pytest.mark.parametrize("G", [8, 2])
def test_config4_512cubed_27point_partitioned(tmp_path, cga, G):
    """BASELINE config 4 at full size as G processes (z-slabs of 512/G planes, two 262144-entry halo planes per inner rank):
    (1) A*1 = diag - neighbours EXACTLY in every rank's rows, through the distributed SpMV (each worker checks its slab);
    (2) 20 iterations of the partitioned throughput loop: all ranks bit-identical, within 1e-10 of the one-GPU loop;
    (3) 3 iterations in the bit-identical mode (cgh_dist_set_strict): history AND x equal the one-GPU strict solve bit for bit."""
    N, iters, strict_iters = 512, 20, 3 if G == 8 else 2
    procs = _ranks(tmp_path, G, (N, N, N), 27, iters, strict_iters, 1, 560)
    n = N ** 3
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(N, N, N, 27, 0x5EED0001, cga.LAYOUT_ELL, b)
    rep, fast = cga.cg_solve(A, b, x, fixed_iters=iters, strict=0)
    xf = x.to_host()
    rep, strict = cga.cg_solve(A, b, x, fixed_iters=strict_iters, strict=1)
    xs1 = x.to_host()
    A.free()
    outs = _join(procs)
    hists = [np.load(os.path.join(tmp_path, "hist_fast_%d.npy" % r)) for r in range(G)]
    for h in hists[1:]:
        assert same_bits_nan(h, hists[0])
    assert hists[0].shape == fast.shape == (iters + 1, 4)
    dev = _rel(hists[0][:, :2], fast[:, :2])
    print("\n[config 4, %d ranks on one GPU] (res, r.r) vs the one-GPU loop over %d iterations: %.3g; %s" % (
        G, iters, dev, " | ".join(l for l in outs[0].strip().splitlines() if "iterations in" in l)))
    assert dev <= 1e-10
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_fast_%d.npy" % r)) for r in range(G)])
    assert np.abs(xs - xf).max() <= 1e-9 * np.abs(xf).max()
    del xs, xf
    hs = [np.load(os.path.join(tmp_path, "hist_strict_%d.npy" % r)) for r in range(G)]
    for h in hs:
        assert same_bits_nan(h, strict)
    xs = np.concatenate([np.load(os.path.join(tmp_path, "x_strict_%d.npy" % r)) for r in range(G)])
    assert np.array_equal(bits(xs), bits(xs1))


def test_config5_cg_on_the_throughput_layouts_vs_exact_rows_strict(cga):
    """BASELINE config 5 at full size: CG (strict = 0) on CGH_LAYOUT_PB (the headline) and CGH_LAYOUT_CSRB against the
    parity-capable path -- CGH_LAYOUT_CSR with exact rows, strict = 1: every row's products and every dot in the
    reference's order (src/csr.c:190-229, src/vec.c:67-85) -- over 100 iterations: the residual trajectory within 1e-10
    relative, and the iterations needed for tol = 1e-6 * ||b||_1 / 2 within +-1."""
    n, max_len, seed, iters = 8 * 1024 * 1024, 2000, 0x5EED0005, 100
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.powerlaw(n, max_len, seed, cga.LAYOUT_CSR, b)
    A.set_exact_rows(True)
    rep, want = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
    tol = 1e-6 * want[0, 0]
    rep, _ = cga.cg_solve(A, b, x, max_iter=1000, tol=tol, strict=1)
    ret_ref = rep.ret
    assert ret_ref > 0
    A.free()
    for name, layout in (("two-phase", cga.LAYOUT_PB), ("column-blocked CSR", cga.LAYOUT_CSRB)):
        A = cga.Matrix.powerlaw(n, max_len, seed, layout, b)
        rep, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=0)
        live = want[:, 0] > 1e-280                       # (a residual that has underflowed has no relative accuracy left)
        dev = _rel(hist[live, 0], want[live, 0])
        rep, _ = cga.cg_solve(A, b, x, max_iter=1000, tol=tol, strict=0)
        print("\n[config 5, %s] residual vs exact-row CSR strict over %d iterations: max relative deviation %.3g; iterations to tol %d vs %d"
              % (name, int(live.sum()) - 1, dev, rep.ret, ret_ref))
        assert dev <= 1e-10
        assert abs(rep.ret - ret_ref) <= 1
        A.free()
