"""CPU-only: libcg_hip.so loads without a GPU and exports every symbol include/*.h declares; without a
device the product path fails loudly instead of computing on the CPU."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        text = re.sub(r"enum\s*\{.*?\}\s*;", " ", text, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_]\w*)\s*\([^;{}]*\)\s*;", text):
            names.append((os.path.basename(h), m.group(1)))
    return names


def test_headers_declare_the_reference_api():
    names = {n for _, n in declared_functions()}
    # the hot-path entry points of the reference
    for must in ("conjugate_gradient_coo", "conjugate_gradient_csr", "conjugate_gradient_ell",
                 "conjugate_gradient_ell_colmajor", "conjugate_gradient_ell_full_colmajor",
                 "mv_ell_symmetric_full_colmajor", "mv_ell_symmetric_full_colmajor_sdtint",
                 "mv_ell_symmetric_upper_colmajor", "mv_ell_symmetric_upper", "mv_ell_symmetric_upper_opt", "mv_ell",
                 "mv_csr_symmetric", "mv_coo_symmetric", "compute_max_nnz_row_upper", "compute_max_nnz_row_full",
                 "coo_to_ell_symmetric_upper", "coo_to_ell_symmetric_upper_colmajor",
                 "coo_to_ell_symmetric_full_colmajor", "coo_to_ell_symmetric_full_colmajor_sdtint",
                 "coo_to_csr", "sorting_coo_to_csr", "vec_sub", "vec_add", "vec_assign", "vec_dot", "vec_axpy",
                 "vec_l1norm", "rmse", "euclidean_distance", "max_difference"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    import conjugate_gradient_amd as cga
    L = cga.lib()
    decl = declared_functions()
    assert len(decl) > 60
    missing = [(h, n) for h, n in decl if not hasattr(L, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu():
    import conjugate_gradient_amd as cga
    L = cga.lib()
    if L.cgh_device_count() > 0:
        pytest.skip("a GPU is present")
    os.environ["CG_HIP_SILENT_ERRORS"] = "1"
    assert L.cgh_init(0) != 0
    with pytest.raises(RuntimeError):
        cga.init(0)
    assert L.cgh_alloc(64) is None
    # reference-signature solver: -1 and x untouched, never a CPU solve
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.conjugate_gradient_ell_full_colmajor.argtypes = [C.c_int, dp, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_double]
    n = 4
    diag = np.full(n, 2.0); ev = np.zeros(n); ec = np.full(n, -1, dtype=np.int32); b = np.ones(n); x = np.full(n, 123.0)
    ret = L.conjugate_gradient_ell_full_colmajor(n, diag.ctypes.data_as(dp), 0, ev.ctypes.data_as(dp), ec.ctypes.data_as(ip), 1,
                                                 b.ctypes.data_as(dp), x.ctypes.data_as(dp), 10, 1e-6)
    assert ret == -1 and np.all(x == 123.0)
    L.vec_dot.restype = C.c_double
    assert np.isnan(L.vec_dot(b.ctypes.data_as(dp), b.ctypes.data_as(dp), n))
