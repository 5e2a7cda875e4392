"""One rank of the 2-process test of the peer-to-peer all-reduce (tests/test_gpu_dist.py): both ranks run on the same
GPU, exchange their mailbox handles through files and check every sum.  python p2p_worker.py <rank> <nranks> <dir>"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

rank, nranks, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
mode = sys.argv[4] if len(sys.argv) > 4 else "normal"     # "mute": rank 1 never sends (timeout test)


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
            if time.time() - t0 > 60:
                raise SystemExit("rank %d: %s never appeared" % (rank, path))
            time.sleep(0.01)
        with open(path, "rb") as f:
            data = f.read()
        assert len(data) == size
        out.append(data)
    return out


cga.init(0)
L = cga.lib()
buf = C.create_string_buffer(64)
cga.check(L.cgh_p2p_local_handle(buf), "local_handle")
publish("handle%d" % rank, buf.raw)
handles = b"".join(collect("handle", 64))
cga.check(L.cgh_p2p_connect(nranks, rank, handles), "connect")
publish("ready%d" % rank, b"1")          # barrier: every mailbox is zeroed and mapped before anybody sends
collect("ready", 1)
if mode == "mute":
    if rank == 1:
        time.sleep(16)                     # connected, mailbox mapped, but silent
        print("rank 1 ok", flush=True)
        sys.exit(0)
    v = np.array([1.0])
    t0 = time.time()
    rc = L.cgh_p2p_allreduce_f64(v.ctypes.data_as(C.POINTER(C.c_double)), 1)
    waited = time.time() - t0
    assert rc != 0 and 8.0 < waited < 14.0 and L.cgh_p2p_error() == 1, (rc, waited)
    t0 = time.time()
    rc = L.cgh_p2p_allreduce_f64(v.ctypes.data_as(C.POINTER(C.c_double)), 1)   # broken: no second 10 s wait
    assert rc != 0 and time.time() - t0 < 1.0
    print("rank 0 ok", flush=True)
    sys.exit(0)
tot = nranks * (nranks + 1) / 2.0
for k in range(1, 201):
    v = np.array([(rank + 1) * k * 0.5, -(rank + 1.0) * k])
    cga.check(L.cgh_p2p_allreduce_f64(v.ctypes.data_as(C.POINTER(C.c_double)), 2), "allreduce")
    assert v[0] == tot * k * 0.5 and v[1] == -tot * k, (rank, k, v)
    if k % 50 == 0 and rank == 0:
        time.sleep(0.05)                   # skew between the ranks: the other one waits in its kernel
    w = np.array([1.0 / (rank + 3) + k])
    cga.check(L.cgh_p2p_allreduce_f64(w.ctypes.data_as(C.POINTER(C.c_double)), 1), "allreduce")
    want = 0.0
    for r in range(nranks):
        want += 1.0 / (r + 3) + k          # rank order, as the kernel adds
    assert w[0] == want, (rank, k, w[0], want)
assert L.cgh_p2p_error() == 0
L.cgh_p2p_disconnect()
print("rank %d ok" % rank, flush=True)
