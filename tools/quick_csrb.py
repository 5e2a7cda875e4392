"""Config 5 layouts side by side: correctness of the column-blocked CSR and the sigma-sorted sliced ELL against the
plain CSR kernel (small and full size), then SpMV timings.  usage: python tools/quick_csrb.py [rows] [rpw,slab_bits,wgs ...]"""
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


def compare(n, max_len, cfgs):
    rng = np.random.default_rng(7)
    xh = rng.standard_normal(n)
    x = cga.DeviceVector.from_host(xh)
    y = cga.DeviceVector(n)
    b = cga.DeviceVector(n)
    t0 = time.time()
    A = cga.Matrix.powerlaw(n, max_len, 0x5EED0005, cga.LAYOUT_CSR, b)
    t_csr = time.time() - t0
    A.set_exact_rows(True)
    A.spmv(x, y)
    ref = y.to_host()
    A.set_exact_rows(False)
    alg = 12 * A.nnz_offdiag + 24 * n
    t = spmv_time(A, x, y)
    csr_bytes = A.stored_bytes
    print("n=%d nnz_off=%d K=%d  CSR build %.2fs spmv %.4f ms  %.2f TB/s alg (%.1f%% of 8)" % (
        n, A.nnz_offdiag, A.K, t_csr, t, alg / t / 1e9, alg / t / 8e7), flush=True)
    scale = np.abs(ref).max()
    A.free()
    for sigma in (65536,):
        cga.config_sell_sigma(sigma)
        t0 = time.time()
        A = cga.Matrix.powerlaw(n, max_len, 0x5EED0005, cga.LAYOUT_SELLS, b)
        tb = time.time() - t0
        A.spmv(x, y)
        got = y.to_host()
        same = np.array_equal(got.view(np.uint64), ref.view(np.uint64))
        t = spmv_time(A, x, y)
        print("  SELLS sigma=%d build %.2fs bytes %.3fx CSR  bit-identical to exact CSR: %s  spmv %.4f ms %.2f TB/s" % (
            sigma, tb, A.stored_bytes / csr_bytes, same, t, alg / t / 1e9), flush=True)
        A.free()
    for cfg in cfgs:
        rpw, cb, wgs = cfg[:3]
        win = cfg[3] if len(cfg) > 3 else -1
        cga.config_csrb(rpw, cb, wgs, win, cfg[4] if len(cfg) > 4 else 0, cfg[5] if len(cfg) > 5 else -1)
        t0 = time.time()
        A = cga.Matrix.powerlaw(n, max_len, 0x5EED0005, cga.LAYOUT_CSRB, b)
        tb = time.time() - t0
        A.spmv(x, y)
        got = y.to_host()
        A.spmv(x, y)
        again = y.to_host()
        err = np.abs(got - ref).max() / scale
        det = np.array_equal(got.view(np.uint64), again.view(np.uint64))
        t = spmv_time(A, x, y)
        print("  CSRB rpw=%d slab_bits=%d wgs=%d win=%d build %.2fs bytes %.3fx CSR  max err %.2e  deterministic %s  spmv %.4f ms  %.2f TB/s alg (%.1f%% of 8)" % (
            rpw, cb, wgs, win, tb, A.stored_bytes / csr_bytes, err, det, t, alg / t / 1e9, alg / t / 8e7), flush=True)
        A.free()


cfgs = []
for a in sys.argv[2:]:
    cfgs.append(tuple(int(t) for t in a.split(",")))
if not cfgs:
    cfgs = [(1024, 17, 0)]
compare(6000, 300, [(512, 8, 0), (1024, 10, 0), (2048, 17, 0)])
compare(200000, 2000, [(512, 12, 0), (1024, 14, 0)])
compare(int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 1024 * 1024, 2000, cfgs)
