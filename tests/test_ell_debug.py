"""CPU-only: the host-side ELL inspection helpers of libcg_hip (csrc/ell_debug.c) against the reference's own
(src/ell.c:11-134, :401-707, compiled from its sources into oracle/_ref/libref.so): same bytes on stdout, same dense
matrices, same return value.  And the boundary check the helpers exist for: the reference's tests/tests_cg.c links
against libcg_hip.so with no undefined symbol (build container only; nothing of the reference is copied)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import conjugate_gradient_amd as cga

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
REF_DIR = "/root/reference"
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


def _capture(fn):
    """run fn() with the C-level stdout redirected to a file; returns the bytes written"""
    libc = C.CDLL(None)
    libc.fflush(None)
    with tempfile.TemporaryFile() as tmp:
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
        try:
            ret = fn()
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        return tmp.read(), ret


def _ell(rng, n, K, layout, mismatch=False):
    vals = np.zeros(n * K); cols = np.full(n * K, -1, dtype=np.int32)
    for r in range(n):
        live = rng.integers(0, K + 1)
        cs = rng.choice(np.setdiff1d(np.arange(n), [r]), size=min(live, n - 1), replace=False)
        for k, c in enumerate(cs):
            q = r * K + k if layout == "row" else k * n + r
            vals[q] = np.round(rng.standard_normal() * 10, 2) or 1.5
            cols[q] = c
    if mismatch:
        cols[np.flatnonzero(vals == 0.0)[:3]] = 0          # padding values with a column id: the warning line
    return vals, cols


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref.so not built")
    L, R = cga.lib(), C.CDLL(REF_SO)
    for lib in (L, R):
        lib.analyze_ell_matrix_full_colmajor.restype = C.c_double
    return L, R


@pytest.mark.parametrize("shape", [(5, 2), (7, 4), (40, 4), (3, 1)])
@pytest.mark.parametrize("mismatch", [False, True])
def test_analyze_printers_same_stdout_and_return(libs, shape, mismatch):
    L, R = libs
    n, K = shape
    rng = np.random.default_rng(n * 10 + K)
    for name, layout in (("analyze_ell_matrix", "row"), ("analyze_ell_matrix_colmajor", "col"), ("analyze_ell_matrix_full_colmajor", "col")):
        vals, cols = _ell(rng, n, K, layout, mismatch)
        args = (n, K, vals.ctypes.data_as(dp), cols.ctypes.data_as(ip))
        mine, r1 = _capture(lambda: getattr(L, name)(*args))
        ref, r2 = _capture(lambda: getattr(R, name)(*args))
        assert mine == ref and len(ref) > 40, name
        if name.endswith("full_colmajor"):
            assert r1 == r2 == float(np.count_nonzero(vals == 0.0)) / (n * K)


@pytest.mark.parametrize("name,layout", [("convert_ell_to_dense", "row"), ("convert_ell_to_dense_colmajor", "col"),
                                         ("convert_ell_full_to_dense_rowmajor", "row"), ("convert_ell_full_to_dense_colmajor", "col")])
def test_dense_converters_same_matrix(libs, name, layout):
    L, R = libs
    n, K = 9, 3
    rng = np.random.default_rng(17)
    vals, cols = _ell(rng, n, K, layout)
    diag = rng.standard_normal(n)
    out = []
    for lib in (L, R):
        A = np.full((n, n), 0.0 if "full" not in name else 99.0)     # the `full` converters clear A themselves
        rows = (dp * n)(*[A[r].ctypes.data_as(dp) for r in range(n)])
        getattr(lib, name)(n, diag.ctypes.data_as(dp), vals.ctypes.data_as(dp), cols.ctypes.data_as(ip), K, rows)
        out.append(A)
    assert np.array_equal(out[0].view(np.uint64), out[1].view(np.uint64))
    assert np.array_equal(out[0], out[0].T) and np.array_equal(np.diag(out[0]), diag)
    text_mine, _ = _capture(lambda: L.print_dense_matrix((dp * n)(*[out[0][r].ctypes.data_as(dp) for r in range(n)]), n))
    text_ref, _ = _capture(lambda: R.print_dense_matrix((dp * n)(*[out[1][r].ctypes.data_as(dp) for r in range(n)]), n))
    assert text_mine == text_ref and text_ref.count(b"\n") == n


def test_reference_tests_cg_links_against_libcg_hip(tmp_path):
    """The reference's own driver (tests/tests_cg.c, with its utils.c and parser.c) compiled BY PATH and linked against
    libcg_hip.so instead of src/{vec,coo,csr,ell,cg}.c: no undefined symbol.  Build container only."""
    src = [os.path.join(REF_DIR, "tests", "tests_cg.c"), os.path.join(REF_DIR, "src", "utils.c"), os.path.join(REF_DIR, "src", "parser.c")]
    if not all(os.path.exists(f) for f in src):
        pytest.skip("the reference sources are not present (GPU box)")
    exe = str(tmp_path / "test_cg_hip_link")
    libdir = os.path.join(ROOT, "conjugate_gradient_amd")
    cmd = ["gcc", "-std=c99", "-fgnu89-inline", "-O1", "-w", "-I" + os.path.join(REF_DIR, "include"), "-o", exe] + src + [
        "-L" + libdir, "-lcg_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--no-undefined", "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    for sym in ("analyze_ell_matrix_full_colmajor", "conjugate_gradient_ell_full_colmajor", "coo_to_ell_symmetric_full_colmajor",
                "compute_max_nnz_row_full", "rmse", "max_difference", "euclidean_distance"):
        assert sym in und                                   # resolved at load time from libcg_hip.so
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libcg_hip.so" in ldd and "not found" not in ldd.split("libcg_hip.so")[1].split("\n")[0]
