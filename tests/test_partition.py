"""CPU-only tests of the multi-GPU path's host logic (partition.cpp) -- needs no GPU:
(1) every rank's local block reproduces the serial SpMV bit for bit once its halo is filled, for 1..8 ranks;
(2) a world_size-2 gloo run of the distributed CG loop (local SpMV by the oracle, halo exchange and the two
    scalar all-reduces over torch.distributed) matches the reference goldens."""
import os
import socket
import sys

import numpy as np
import pytest

import foamio
import oracle_lib as O
from oracle_lib import bits

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(tag):
    return np.load(os.path.join(foamio.GOLDEN, "golden_%s.npz" % tag))


def _parts(s, G, row_starts=None):
    import conjugate_gradient_amd as cga
    return [cga.Partition(s["n"], s["diag"], s["upper"], s["rows"], s["cols"], G, g, row_starts) for g in range(G)]


def _exchange(parts, owned):
    """fill each rank's extended vector [owned | halo] the way the send/recv lists say"""
    ext = [np.concatenate([owned[g], np.full(p.n_halo, np.nan)]) for g, p in enumerate(parts)]
    for g, p in enumerate(parts):
        for nb in p.neighbours:
            q = parts[nb["rank"]]
            back = [m for m in q.neighbours if m["rank"] == g]
            assert len(back) == 1 and back[0]["send"].size == nb["recv_cnt"]
            ext[g][p.n_loc + nb["recv_off"]: p.n_loc + nb["recv_off"] + nb["recv_cnt"]] = owned[nb["rank"]][back[0]["send"]]
    return ext


CASES = [("cylinder_2000", lambda: foamio.foam("2000")), ("cylinder_8000", lambda: foamio.foam("8000")),
         ("poisson27_12x10x8", lambda: foamio.poisson_coo(12, 10, 8, 0x5EED0001, 27)),
         ("poisson7_16", lambda: foamio.poisson_coo(16, 16, 16, 0x5EED0001, 7))]


@pytest.mark.parametrize("G", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_partitioned_spmv_is_bit_identical_to_serial(case, G):
    tag, make = case
    s, g = make(), _golden(tag)
    n = s["n"]
    parts = _parts(s, G)
    assert sum(p.n_loc for p in parts) == n and parts[0].lo == 0
    x = g["x_probe"]
    owned = [x[p.lo:p.lo + p.n_loc] for p in parts]
    ext = _exchange(parts, owned)
    y = np.concatenate([O.mv_ell_full(p.n_loc, p.K, p.diag, p.vals, p.cols, np.ascontiguousarray(e)) if p.K else p.diag * e[:p.n_loc]
                        for p, e in zip(parts, ext)])
    assert not np.isnan(y).any()                                   # every referenced halo slot was filled
    assert np.array_equal(bits(y), bits(g["y_probe"]))
    for p in parts:
        # local rows are the serial ELL's rows: same slot order, columns renumbered
        a, b = p.interior
        assert 0 <= a <= b <= p.n_loc
        cols = p.cols.reshape(p.K, p.n_loc)
        assert np.all(cols[:, a:b] < p.n_loc)                        # interior rows reference no halo column
        glob = np.where(cols >= p.n_loc, p.halo_global[np.clip(cols - p.n_loc, 0, max(p.n_halo - 1, 0))] if p.n_halo else -1,
                        np.where(cols >= 0, cols + p.lo, -1))
        want = g["ell_full_cols"].reshape(-1, n)[:, p.lo:p.lo + p.n_loc]
        k = min(p.K, want.shape[0])
        assert np.array_equal(glob[:k], want[:k]) and np.all(want[k:] == -1)
        assert np.all(np.diff(p.halo_global) > 0)


def test_uneven_row_starts_and_bad_input():
    s = foamio.poisson_coo(6, 5, 7, 3, 27)
    n = s["n"]
    starts = np.array([0, 30, 30, 150, n])          # an empty rank in the middle
    parts = _parts(s, 4, starts)
    assert [p.n_loc for p in parts] == [30, 0, 120, n - 150]
    x = np.random.default_rng(1).standard_normal(n)
    ext = _exchange(parts, [x[p.lo:p.lo + p.n_loc] for p in parts])
    y = np.concatenate([O.mv_ell_full(p.n_loc, p.K, p.diag, p.vals, p.cols, np.ascontiguousarray(e)) for p, e in zip(parts, ext) if p.n_loc])
    K = O.max_nnz_full(n, s["rows"], s["cols"])
    ev, ec, _ = O.coo_to_ell_full(n, s["upper"], s["rows"], s["cols"], K)
    assert np.array_equal(bits(y), bits(O.mv_ell_full(n, K, s["diag"], ev, ec, x)))
    import conjugate_gradient_amd as cga
    bad = s["rows"].copy(); bad[5] = s["cols"][5]
    with pytest.raises(RuntimeError):
        cga.Partition(n, s["diag"], s["upper"], bad, s["cols"], 2, 0)


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _gloo_worker(rank, world, port, tag, out, strict=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import conjugate_gradient_amd as cga
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    s = dict(CASES)[tag]()
    g = _golden(tag)
    P = cga.Partition(s["n"], s["diag"], s["upper"], s["rows"], s["cols"], world, rank)
    nl = P.n_loc
    b = s["source"][P.lo:P.lo + nl]

    def halo_exchange(pext):
        reqs = []
        bufs = []
        for nb in P.neighbours:
            send = torch.from_numpy(np.ascontiguousarray(pext[nb["send"]]))
            recv = torch.empty(nb["recv_cnt"], dtype=torch.float64)
            reqs.append(dist.isend(send, nb["rank"]))
            reqs.append(dist.irecv(recv, nb["rank"]))
            bufs.append((nb, recv, send))
        for r_ in reqs:
            r_.wait()
        for nb, recv, _ in bufs:
            pext[nl + nb["recv_off"]: nl + nb["recv_off"] + nb["recv_cnt"]] = recv.numpy()

    def allsum(*vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t)
        return t.tolist()

    def spmv(pext):
        return O.mv_ell_full(nl, P.K, P.diag, P.vals, P.cols, pext)

    def chained(*term_vectors):
        """the bit-identical mode of dist.hip (strict_chain): the reference's left-to-right sums over the GLOBAL vector --
        round q: rank q continues the exact running sums over its own part, everybody else contributes zeros, one
        all-reduce (s + 0 = s exactly) hands them on.  np.cumsum adds strictly in order."""
        run = [0.0] * len(term_vectors)
        for q in range(world):
            mine = [float(np.cumsum(np.concatenate(([s0], t)))[-1]) if q == rank else 0.0 for s0, t in zip(run, term_vectors)]
            run = allsum(*mine)
        return run

    if strict:
        def sums_r(r):
            return chained(r * r, np.abs(r))

        def sum_pAp(p, Ap):
            return chained(p * Ap)
    else:
        def sums_r(r):
            return allsum(float(np.dot(r, r)), float(np.abs(r).sum()))

        def sum_pAp(p, Ap):
            return allsum(float(np.dot(p, Ap)))

    # the loop of dist.hip / reference cg.c:414-508
    x = np.zeros(nl)
    pext = np.zeros(nl + P.n_halo)
    r = b - spmv(pext)
    pext[:nl] = r
    rr, l1 = sums_r(r)
    traj = [l1 / 2.0]
    ret = -1
    for it in range(1, int(g["max_iter"])):
        halo_exchange(pext)
        Ap = spmv(pext)
        (pAp,) = sum_pAp(pext[:nl], Ap)
        alpha = rr / pAp
        x += alpha * pext[:nl]
        r -= alpha * Ap
        rr_new, l1 = sums_r(r)
        traj.append(l1 / 2.0)
        if l1 / 2.0 < float(g["tol"]):
            ret = it + 1
            break
        pext[:nl] = r + (rr_new / rr) * pext[:nl]
        rr = rr_new
    xs = [torch.empty(p_, dtype=torch.float64) for p_ in [0] * world]
    gathered = [None] * world
    dist.all_gather_object(gathered, (P.lo, x))
    if rank == 0:
        full = np.concatenate([xx for _, xx in sorted(gathered, key=lambda t: t[0])])
        np.savez(out, ret=ret, traj=np.array(traj), x=full)
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["cylinder_2000", "poisson27_12x10x8"])
def test_two_rank_gloo_cg_matches_reference(tag, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    out = str(tmp_path / "dist.npz")
    mp.spawn(_gloo_worker, args=(2, port, tag, out), nprocs=2, join=True)
    res, g = np.load(out), _golden(tag)
    assert abs(int(res["ret"]) - int(g["cg_ret"][4])) <= 1            # iterations within +-1
    gt = g["traj"][:, 0]
    m = min(gt.size, res["traj"].size)
    rel = np.abs(res["traj"][:m] - gt[:m]) / gt[:m]
    assert rel[:11].max() < 1e-10                                      # same bar as the single-GPU throughput mode
    assert np.abs(res["x"] - g["cg_x"]).max() <= 1e-5 * max(1.0, np.abs(g["cg_x"]).max())


@pytest.mark.parametrize("tag,world", [("cylinder_2000", 2), ("poisson27_12x10x8", 3)])
def test_gloo_strict_chain_is_bit_identical(tag, world, tmp_path):
    """The bit-identical multi-GPU mode (cgh_dist_set_strict) restated on the CPU with gloo: sums chained through the ranks
    in rank order give the reference's trajectory, return value and x bit for bit (the GPU version of this test:
    tests/test_gpu_dist.py::test_multi_process_strict_mode_is_bit_identical)."""
    import torch.multiprocessing as mp
    port = _free_port()
    out = str(tmp_path / "dist_strict.npz")
    mp.spawn(_gloo_worker, args=(world, port, tag, out, True), nprocs=world, join=True)
    res, g = np.load(out), _golden(tag)
    assert int(res["ret"]) == int(g["cg_ret"][4])
    assert np.array_equal(res["traj"].view(np.uint64), np.ascontiguousarray(g["traj"][:, 0]).view(np.uint64))
    assert np.array_equal(res["x"].view(np.uint64), g["cg_x"].view(np.uint64))
