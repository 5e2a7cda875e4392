"""bench_convert.py -- device-side format conversion against the reference's own converter on one
host core: upper-triangular COO of the 7-point Poisson stencil -> slot-major full ELL.
    reference: compute_max_nnz_row_full + coo_to_ell_symmetric_full_colmajor (src/ell.c:360-396, 791-858), from
               oracle/_ref/libref.so (the reference's sources compiled where they lie)
    here:      cgh_matrix_from_coo_device (count -> scan -> bucket -> place, convert.hip), inputs resident in HBM
The two outputs are compared bit for bit.  One JSON line.   python bench_convert.py [N ...]   (grid N^3, default 128 256)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conjugate_gradient_amd as cga  # noqa: E402
import foamio  # noqa: E402


def poisson7_upper_coo(N):
    """row-sorted upper COO of the N^3 7-point stencil without a sort (+1, +N, +N^2 in that order inside a row)"""
    n = N ** 3
    idx = np.arange(n, dtype=np.int64)
    x, y, z = idx % N, (idx // N) % N, idx // (N * N)
    ok = np.stack([x + 1 < N, y + 1 < N, z + 1 < N], axis=1)
    cols = (idx[:, None] + np.array([1, N, N * N], dtype=np.int64)[None, :])[ok]
    rows = np.repeat(idx, 3).reshape(n, 3)[ok]
    return n, rows.astype(np.int32), cols.astype(np.int32), np.full(rows.size, -1.0), 6.0 + 1e-3 * foamio.hash01(0x5EED0001, idx)


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [128, 256]
    cga.init(0)
    L = cga.lib()
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref.so")
    R = C.CDLL(ref_so) if os.path.exists(ref_so) else None
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    out = {"what": "upper COO -> slot-major full ELL, 7-point Poisson N^3", "cases": []}
    for N in sizes:
        n, rows, cols, upper, diag = poisson7_upper_coo(N)
        U = rows.size
        d_rows, d_cols = cga.DeviceVector.from_host(rows), cga.DeviceVector.from_host(cols)
        d_up, d_diag = cga.DeviceVector.from_host(upper), cga.DeviceVector.from_host(diag)
        best = 1e9
        for rep in range(3):
            cga.check(L.cgh_sync())
            t0 = time.perf_counter()
            h = L.cgh_matrix_from_coo_device(n, U, d_diag.ptr, d_up.ptr, d_rows.ptr, d_cols.ptr, cga.LAYOUT_ELL)
            cga.check(L.cgh_sync())
            dt = time.perf_counter() - t0
            A = cga.Matrix(h)
            best = min(best, dt)
            if rep < 2:
                A.free()
        K = A.K
        case = {"N": N, "rows": n, "upper_entries": int(U), "K": int(K), "device_ms": best * 1e3,
                "device_Mentries_per_s": U / best / 1e6,
                # read rows, cols, values once; write the padded ELL once (the minimum any converter moves)
                "device_GBs_min_traffic": (16.0 * U + 12.0 * n * K) / best / 1e9}
        if R is not None:
            R.compute_max_nnz_row_full.argtypes = [C.c_int, C.c_int, ip, ip]
            ev = np.empty(n * K); ec = np.empty(n * K, dtype=np.int32)
            t0 = time.perf_counter()
            Kref = R.compute_max_nnz_row_full(n, U, rows.ctypes.data_as(ip), cols.ctypes.data_as(ip))
            R.coo_to_ell_symmetric_full_colmajor(n, U, upper.ctypes.data_as(dp), rows.ctypes.data_as(ip), cols.ctypes.data_as(ip),
                                                 ev.ctypes.data_as(dp), ec.ctypes.data_as(ip), K)
            cpu = time.perf_counter() - t0
            gv, gc = A.download_ell()
            case.update({"reference_1core_ms": cpu * 1e3, "speedup": cpu / best, "K_reference": int(Kref),
                         "bit_identical": bool(Kref == K and np.array_equal(gc, ec) and gv.tobytes() == ev.tobytes())})
        A.free()
        out["cases"].append(case)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
