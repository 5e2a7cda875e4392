"""Config 5, the two-phase (propagation-blocking) SpMV: correctness against the exact-order CSR kernel (small and full
size), then SpMV timings.  usage: python tools/quick_pb.py [rows] [span_bits,target,split ...]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import conjugate_gradient_amd as cga

cga.init(0)
ms = C.c_double()


def spmv_time(A, x, y, reps=20):
    cga.check(cga.lib().cgh_time_spmv(A.h, x.ptr, y.ptr, reps, C.byref(ms)))
    return ms.value


def compare(n, max_len, cfgs, time_csr=True):
    rng = np.random.default_rng(7)
    xh = rng.standard_normal(n)
    x = cga.DeviceVector.from_host(xh)
    y = cga.DeviceVector(n)
    b = cga.DeviceVector(n)
    A = cga.Matrix.powerlaw(n, max_len, 0x5EED0005, cga.LAYOUT_CSR, b)
    A.set_exact_rows(True)
    A.spmv(x, y)
    ref = y.to_host()
    A.set_exact_rows(False)
    alg = 12 * A.nnz_offdiag + 24 * n
    csr_bytes = A.stored_bytes
    if time_csr:
        t = spmv_time(A, x, y)
        print("n=%d nnz_off=%d K=%d  CSR spmv %.4f ms  %.2f TB/s alg (%.1f%% of 8)" % (
            n, A.nnz_offdiag, A.K, t, alg / t / 1e9, alg / t / 8e7), flush=True)
    scale = np.abs(ref).max()
    A.free()
    for span_bits, target, split in cfgs:
        cga.config_pb(span_bits, target, split)
        t0 = time.time()
        A = cga.Matrix.powerlaw(n, max_len, 0x5EED0005, cga.LAYOUT_PB, b)
        tb = time.time() - t0
        A.spmv(x, y)
        got = y.to_host()
        err = np.abs(got - ref).max() / scale
        same = 0
        for _ in range(5):
            A.spmv(x, y)
            same += int(np.array_equal(y.to_host().view(np.uint64), got.view(np.uint64)))
        t = spmv_time(A, x, y)
        print("  PB span_bits=%d target=%d split=%d build %.2fs bytes %.3fx CSR  max err %.2e  same bits in %d of 5 reruns  spmv %.4f ms  %.2f TB/s alg (%.1f%% of 8)" % (
            span_bits, target, split, tb, A.stored_bytes / csr_bytes, err, same, t, alg / t / 1e9, alg / t / 8e7), flush=True)
        assert err < 1e-13, err
        A.free()


cfgs = []
for a in sys.argv[2:]:
    cfgs.append(tuple(int(t) for t in a.split(",")))
if not cfgs:
    cfgs = [(14, 131072, 1)]
compare(300, 40, [(14, 131072, 1), (13, 100, 1)], False)
compare(6000, 300, [(14, 131072, 1), (13, 2000, 2), (14, 500, 3)], False)
compare(200000, 2000, [(14, 131072, 1), (13, 20000, 2)], False)
compare(int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 1024 * 1024, 2000, cfgs)
