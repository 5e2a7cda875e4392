#!/usr/bin/env python3
"""Golden arrays for the OpenFOAM readers: what the REFERENCE's own parseDoubleArray / parseIntArray /
parseDoubleArraySolution (src/parser.c:29-177, inside oracle/_ref/libref.so, compiled from the reference's sources by
oracle/Makefile) return for the three cylinder systems and the Phi field kept under tests/golden/.  Build container only;
the output tests/golden/golden_parser.npz is data and is committed; tests/test_host_parser.py compares libcg_hip's parser
with it."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")
ref = C.CDLL(os.path.join(HERE, "_ref", "libref.so"))
libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p
libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
libc.fclose.argtypes = [C.c_void_p]
libc.free.argtypes = [C.c_void_p]
for f in ("parseDoubleArray", "parseIntArray", "parseDoubleArraySolution"):
    getattr(ref, f).restype = C.c_void_p
    getattr(ref, f).argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]


def take(fn, fp, key, ctype):
    cnt = C.c_int(-1)
    ptr = fn(fp, key, C.byref(cnt))
    assert ptr and cnt.value >= 0, key
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt.value,)).copy()
    libc.free(ptr)
    return a


out = {}
for name in ("2000", "8000", "32k"):
    fp = libc.fopen(os.path.join(GOLDEN, "cylinder_%s.system" % name).encode(), b"r")
    assert fp
    # file order, as the reference's drivers ask for them (tests/tests_cg.c:1095-1099)
    for key, fn, ctype in (("source", ref.parseDoubleArray, C.c_double), ("diag", ref.parseDoubleArray, C.c_double),
                           ("upper", ref.parseDoubleArray, C.c_double), ("upperAddr", ref.parseIntArray, C.c_int),
                           ("lowerAddr", ref.parseIntArray, C.c_int)):
        out["%s_%s" % (name, key)] = take(fn, fp, key.encode(), ctype)
    libc.fclose(fp)
fp = libc.fopen(os.path.join(GOLDEN, "Phi_2000").encode(), b"r")
out["phi_2000"] = take(ref.parseDoubleArraySolution, fp, b"internalField   nonuniform List<scalar>", C.c_double)
libc.fclose(fp)
np.savez_compressed(os.path.join(GOLDEN, "golden_parser.npz"), **out)
for k, v in out.items():
    print(k, v.shape, v.dtype)
