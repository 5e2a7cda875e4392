"""One-GPU timing of the multi-GPU iteration: rank 3 of 8 of the benchmark matrix (256^3 7-point: a 256x256x32 slab,
2.1M rows, two 65536-entry halo planes) with itself as both neighbours (cgh_dist_loopback), against the same slab
without neighbours (1 rank of 1) and the single-GPU loop on a matrix of the same size.  us/iteration."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
buf = C.create_string_buffer(64)
cga.check(L.cgh_p2p_local_handle(buf))
cga.check(L.cgh_p2p_connect(1, 0, buf.raw))
cga.check(L.cgh_comm_init_local(1, 0))
points = int(sys.argv[1]) if len(sys.argv) > 1 else 7
nx = ny = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nzl = int(sys.argv[3]) if len(sys.argv) > 3 else 32
iters = 400
tk = int(os.environ.get("TIME_KERNELS", "0"))   # 1: HIP events around every SpMV, as bench_dist's timed loop has them


def timed(D, b, x):
    D.start(b, x, -1.0, 3 * iters + 8)
    D.iterate(iters, 0)
    cga.check(L.cgh_sync())
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter()
        D.iterate(iters, tk)
        cga.check(L.cgh_sync())
        best = min(best, (time.perf_counter() - t0) / iters * 1e6)
    h = D.history(0, 3 * iters + 1)
    return best, h[1, 0], h[-1, 0]


nl = nx * ny * nzl
b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
D = cga.DistRank.poisson(nx, ny, 8 * nzl, points, 0x5EED0001, 8, 3, b)
D.loopback()
assert L.cgh_dist_halo_is_p2p(D.h) == 1
print("inner slab, own neighbour (rank 3 of 8, loopback):  %7.1f us/it   res %.3e -> %.3e" % timed(D, b, x), flush=True)
D.free()
D = cga.DistRank.poisson(nx, ny, nzl, points, 0x5EED0001, 1, 0, b)
print("same rows, no neighbours (1 rank of 1):             %7.1f us/it   res %.3e -> %.3e" % timed(D, b, x), flush=True)
D.free()
A = cga.Matrix.poisson(nx, ny, nzl, points, 0x5EED0001, cga.LAYOUT_ELL, b)
sv = cga.Solver(A, 3 * iters + 8)
sv.start(b, x, tol=-1.0, strict=0)
sv.iterate(iters, 0)
cga.check(L.cgh_sync())
t0 = time.perf_counter()
sv.iterate(iters, 0)
cga.check(L.cgh_sync())
print("single-GPU loop, same rows:                         %7.1f us/it" % ((time.perf_counter() - t0) / iters * 1e6), flush=True)
