"""One rank of the FULL-SIZE multi-process tests (tests/test_gpu_fullsize.py): BASELINE configs 3 and 4 row-partitioned
over N processes that share the one GPU of the test box (peer-to-peer transport: mailbox all-reduces, deposit-buffer halo;
handles through files, as tests/dist_worker.py).
    python dist_fullsize_worker.py <rank> <nranks> <dir> <nx> <ny> <nz> <points> <fast_iters> <strict_iters> <check_ones>
check_ones: the DISTRIBUTED SpMV on the vector of ones (the first iteration of a solve whose right-hand side is 1 computes
A*1 into the rank's A p): off-diagonals are -1 and diag = (points - 1) + 1e-3 h(i), so (A*1)[i] = diag[i] - neighbours(i)
exactly, halo rows included.  Then `fast_iters` iterations of the throughput loop and `strict_iters` of the bit-identical
one; writes hist_fast_<rank>.npy, x_fast_<rank>.npy, hist_strict_<rank>.npy, x_strict_<rank>.npy."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

rank, nranks, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nx, ny, nz, points, fast_iters, strict_iters, check_ones = (int(v) for v in sys.argv[4:11])


def publish(name, data):
    with open(os.path.join(d, name + ".tmp"), "wb") as f:
        f.write(data)
    os.replace(os.path.join(d, name + ".tmp"), os.path.join(d, name))


def collect(prefix, size):
    out = []
    for r in range(nranks):
        path = os.path.join(d, "%s%d" % (prefix, r))
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 240:
                raise SystemExit("rank %d: %s never appeared" % (rank, path))
            time.sleep(0.01)
        with open(path, "rb") as f:
            data = f.read()
        assert len(data) == size, (path, len(data))
        out.append(data)
    return b"".join(out)


os.environ["CGH_DIST_GRID_MAX"] = "64"   # the ranks share one GPU: see alloc_solver_state (dist.hip)
cga.init(0)
L = cga.lib()
buf = C.create_string_buffer(64)
cga.check(L.cgh_p2p_local_handle(buf), "local_handle")
publish("handle%d" % rank, buf.raw)
cga.check(L.cgh_p2p_connect(nranks, rank, collect("handle", 64)), "p2p connect")
publish("connected%d" % rank, b"1")
collect("connected", 1)
cga.check(L.cgh_comm_init_local(nranks, rank), "comm_init_local")

z0, z1 = nz * rank // nranks, nz * (rank + 1) // nranks
nl = nx * ny * (z1 - z0)
b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
D = cga.DistRank.poisson(nx, ny, nz, points, 0x5EED0001, nranks, rank, b)
assert L.cgh_dist_halo_is_p2p(D.h) == 1 and D.n_loc == nl
publish("ready%d" % rank, b"1")
collect("ready", 1)


def run(iters, strict):
    D.start(b, x, -1.0, iters + 1, strict)       # tol < 0: never converges
    queued = 0
    while queued < iters:
        D.iterate(min(8, iters - queued), 0)
        done, queued = D.poll()
        assert done == 0
    return D.history(0, iters + 1)


if check_ones:
    bh = b.to_host()
    ones = np.ones(nl)
    cga.check(L.cgh_h2d(b.ptr, ones.ctypes.data, nl * 8), "h2d")
    run(1, 0)                                      # p0 = r0 = b = 1, so the iteration's SpMV leaves A*1 in A p
    got = D.vector_to_host(2)
    diag, _, _ = D.download_local()
    if points == 27:
        c = lambda m: np.where((np.arange(m) == 0) | (np.arange(m) == m - 1), 2, 3).astype(np.int64)  # noqa: E731
        cz = c(nz)[z0:z1]
        cnt = (cz[:, None, None] * c(ny)[None, :, None] * c(nx)[None, None, :]).reshape(-1) - 1
    else:
        e = lambda m: 2 - ((np.arange(m) == 0).astype(np.int64) + (np.arange(m) == m - 1))  # noqa: E731
        cnt = (e(nz)[z0:z1][:, None, None] + e(ny)[None, :, None] + e(nx)[None, None, :]).reshape(-1)
    want = diag - cnt
    assert got.tobytes() == want.tobytes(), "rank %d: A*1 through the distributed SpMV differs in %d rows" % (
        rank, int((got.view(np.uint64) != want.view(np.uint64)).sum()))
    cga.check(L.cgh_h2d(b.ptr, bh.ctypes.data, nl * 8), "h2d")
    del got, diag, want, cnt, bh

if fast_iters:
    t0 = time.time()
    hist = run(fast_iters, 0)
    np.save(os.path.join(d, "hist_fast_%d.npy" % rank), hist)
    np.save(os.path.join(d, "x_fast_%d.npy" % rank), x.to_host())
    print("rank %d fast: %d iterations in %.2f s" % (rank, fast_iters, time.time() - t0), flush=True)
if strict_iters:
    t0 = time.time()
    hist = run(strict_iters, 1)
    np.save(os.path.join(d, "hist_strict_%d.npy" % rank), hist)
    np.save(os.path.join(d, "x_strict_%d.npy" % rank), x.to_host())
    print("rank %d strict: %d iterations in %.2f s" % (rank, strict_iters, time.time() - t0), flush=True)
publish("finished%d" % rank, b"1")         # nobody unmaps a buffer a peer may still write to
collect("finished", 1)
D.free()
L.cgh_p2p_disconnect()
print("rank %d ok" % rank, flush=True)
