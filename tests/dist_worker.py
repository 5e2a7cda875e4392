"""One rank of the multi-PROCESS test of the distributed CG loop (tests/test_gpu_dist.py): all ranks share the one GPU
of the test box, so the transport is the peer-to-peer one throughout (mailbox all-reduce + deposit-buffer halo); handles
and descriptors travel through files.
    python dist_worker.py <rank> <nranks> <dir> <nx> <ny> <nz> <points> <max_iter> <tol>     z-slabs from the device generator
    python dist_worker.py <rank> <nranks> <dir> tag <golden input> 0 0 <max_iter> <tol>      row blocks of a stored system
Writes hist_<rank>.npy and x_<rank>.npy.  A trailing "strict" selects the bit-identical mode (cgh_dist_set_strict).  A trailing "mute" makes rank 1 connect and then stay silent: rank 0's waits (halo
entries in the SpMV, reduction messages in the vector kernels) must end after their limit and surface as an error."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

rank, nranks, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
tag = sys.argv[5] if sys.argv[4] == "tag" else None
nx, ny, nz, points = (0, 0, 0, 0) if tag else (int(v) for v in sys.argv[4:8])
max_iter, tol = int(sys.argv[8]), float(sys.argv[9])
strict = 1 if sys.argv[-1] == "strict" else 0


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
            if time.time() - t0 > 90:
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
publish("connected%d" % rank, b"1")      # barrier: every mailbox is zeroed and mapped before anybody sends (creating
collect("connected", 1)                  # a DistRank already does: it connects its halo through the mailboxes)
cga.check(L.cgh_comm_init_local(nranks, rank), "comm_init_local")

if tag:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_common import inputs  # noqa: E402
    sy = inputs(tag)
    P = cga.Partition(sy["n"], sy["diag"], sy["upper"], sy["rows"], sy["cols"], nranks, rank)
    D = cga.DistRank.from_partition(P)
    nl = P.n_loc
    b = cga.DeviceVector.from_host(sy["source"][P.lo:P.lo + nl]); x = cga.DeviceVector(nl)
else:
    nl = nx * ny * (nz * (rank + 1) // nranks - nz * rank // nranks)
    b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
    D = cga.DistRank.poisson(nx, ny, nz, points, 0x5EED0001, nranks, rank, b)
assert L.cgh_dist_halo_is_p2p(D.h) == 1          # connected by the constructor, through the mailboxes
desc = C.create_string_buffer(512)               # ... and once more by hand, through files (the explicit API)
cga.check(L.cgh_dist_halo_descriptor(D.h, desc), "halo descriptor")
publish("desc%d" % rank, desc.raw)
cga.check(L.cgh_dist_halo_connect(D.h, collect("desc", 512)), "halo connect")
assert L.cgh_dist_halo_is_p2p(D.h) == 1
publish("ready%d" % rank, b"1")          # barrier: every mailbox / deposit buffer is zeroed and mapped before anybody sends
collect("ready", 1)

if sys.argv[-1] == "mute":                   # two ranks
    if rank == 1:
        t0 = time.time()
        while not os.path.exists(os.path.join(d, "finished0")):   # says nothing, pushes nothing; leaves after rank 0
            assert time.time() - t0 < 150
            time.sleep(0.05)
        D.free()
        L.cgh_p2p_disconnect()
        print("rank 1 ok (mute)", flush=True)
        sys.exit(0)
    t0 = time.time()
    D.start(b, x, tol, max_iter)             # (start's own reduction already waits for the silent peer)
    failed = ""
    try:
        D.iterate(3, 0)
        D.poll()
    except RuntimeError as e:
        failed = str(e)
    hist = D.history(0, 4)
    waited = time.time() - t0
    assert failed and "did not arrive" in failed, failed
    assert not np.all(np.isfinite(hist[1:, 0])), hist
    assert 9.0 < waited < 60.0, waited       # each kind of wait gives up once, then nobody waits again
    t1 = time.time()
    D.iterate(3, 0)                          # later work does not wait at all
    try:
        D.poll()
    except RuntimeError:
        pass
    assert time.time() - t1 < 5.0
    publish("finished0", b"1")
    print("rank 0 ok timeout after %.1f s: %s" % (waited, failed), flush=True)
    sys.exit(0)

D.start(b, x, tol, max_iter, strict)
done, queued = 0, 0
while queued < max_iter - 1 and not done:
    D.iterate(min(7, max_iter - 1 - queued), 0)
    done, queued = D.poll()
    if rank == nranks - 1:
        time.sleep(0.002)                  # skew: the last rank is always late
iters = done if done else queued
hist = D.history(0, iters + 1)
np.save(os.path.join(d, "hist_%d.npy" % rank), hist)
np.save(os.path.join(d, "x_%d.npy" % rank), x.to_host())
# The same solve once more on the same object.  After convergence the ranks have queued different numbers of frozen
# iterations (each stops at its own poll): the message tags count (solve, iteration), so the second solve must still
# find every rank in step -- and give the same bits.
x2 = cga.DeviceVector(nl)
D.start(b, x2, tol, max_iter, strict)
done2, queued2 = 0, 0
step = 5 + 3 * rank                        # ... and queue in different batch sizes this time
while queued2 < max_iter - 1 and not done2:
    D.iterate(min(step, max_iter - 1 - queued2), 0)
    done2, queued2 = D.poll()
iters2 = done2 if done2 else queued2
assert iters2 == iters, (iters, iters2)
hist2 = D.history(0, iters2 + 1)
assert np.nan_to_num(hist2).tobytes() == np.nan_to_num(hist).tobytes()
assert x2.to_host().tobytes() == x.to_host().tobytes()
publish("finished%d" % rank, b"1")         # nobody unmaps a buffer a peer may still write to
collect("finished", 1)
D.free()
L.cgh_p2p_disconnect()
print("rank %d ok done=%d" % (rank, done), flush=True)
