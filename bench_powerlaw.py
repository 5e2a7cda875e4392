#!/usr/bin/env python3
"""BASELINE config 5: SPD matrix with power-law row lengths (8M rows, 1..2000 entries/row): CSR-adaptive vs padded
layouts.  Prints one JSON report (bytes stored per layout, SpMV time and algorithmic GB/s, CG iterations/s)."""
import argparse
import ctypes as C
import json

import conjugate_gradient_amd as cga

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8 * 1024 * 1024)
ap.add_argument("--max-len", type=int, default=2000)
ap.add_argument("--iters", type=int, default=100)
args = ap.parse_args()
cga.init(0)
n = args.rows
out = {"workload": "power-law SPD, %d rows, row length 1..%d (Pareto exponent 2), seed 0x5EED0005" % (n, args.max_len)}
b = cga.DeviceVector(n)
x = cga.DeviceVector(n)
y = cga.DeviceVector(n)
ms = C.c_double()
for name, layout in (("csr_stream_adaptive", cga.LAYOUT_CSR), ("sliced_ell_64", cga.LAYOUT_SELL)):
    A = cga.Matrix.powerlaw(n, args.max_len, 0x5EED0005, layout, b)
    alg = 12 * A.nnz_offdiag + 24 * n
    cga.check(cga.lib().cgh_time_spmv(A.h, b.ptr, y.ptr, 20, C.byref(ms)))
    rep, _ = cga.cg_solve(A, b, x, fixed_iters=args.iters, history=False)
    out[name] = {"stored_bytes": int(A.stored_bytes), "spmv_ms": ms.value, "spmv_GBs_algorithmic": alg / ms.value / 1e6,
                 "spmv_frac_of_8TBs": alg / ms.value / 1e6 / 8000.0, "stored_GBs": A.stored_bytes / ms.value / 1e6,
                 "cg_iterations_per_s": args.iters / rep.solve_ms * 1e3}
    if layout == cga.LAYOUT_CSR:
        out["rows"] = n
        out["nnz_offdiag"] = int(A.nnz_offdiag)
        out["row_length_min_mean_max"] = [A.min_row, A.nnz_offdiag / n, A.K]
        out["algorithmic_spmv_bytes"] = int(alg)
        out["ell_padded"] = {"width": A.K, "stored_bytes": int(n * A.K * 12 + n * 8),
                             "padding_ratio_vs_csr": (n * A.K * 12 + n * 8) / A.stored_bytes,
                             "note": "not instantiated: the reference's single-width ELL would store this many bytes"}
    A.free()
out["sliced_ell_64"]["padding_ratio_vs_csr"] = out["sliced_ell_64"]["stored_bytes"] / out["csr_stream_adaptive"]["stored_bytes"]
print(json.dumps(out))
