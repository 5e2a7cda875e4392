"""Order-exact parallel summation (csrc/xsum.h): the host model (oracle/xsum_model.cpp) must give, bit for bit, the
plain left-to-right sum the reference computes (vec_dot / vec_l1norm, /root/reference/src/vec.c:67-85), whatever the
data -- binade crossings, cancellation, exact ties, zeros, subnormals, huge elements, non-finite values.
CPU only; the device kernels are checked against the same sequential sum in tests/test_gpu_xsum.py."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "oracle", "libxsum_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB):
        pytest.skip("oracle/libxsum_model.so not built (run `make -C oracle`)")
    lib = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    lib.xsum_model_sum.restype = C.c_double
    lib.xsum_model_sum.argtypes = [dp, C.c_int64, C.POINTER(C.c_int64)]
    lib.xsum_model_sequential.restype = C.c_double
    lib.xsum_model_sequential.argtypes = [dp, C.c_int64]
    return lib


def run(lib, t):
    t = np.ascontiguousarray(t, dtype=np.float64)
    st = (C.c_int64 * 4)()
    p = t.ctypes.data_as(C.POINTER(C.c_double))
    got = lib.xsum_model_sum(p, t.size, st)
    want = lib.xsum_model_sequential(p, t.size)
    return got, want, list(st)


def same_bits(a, b):
    return np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64) or (np.isnan(a) and np.isnan(b))


def cases():
    rng = np.random.default_rng(20240611)
    n = 300_000
    yield "squares", rng.standard_normal(n) ** 2
    yield "abs", np.abs(rng.standard_normal(n))
    yield "mixed_sign", rng.standard_normal(n)
    yield "mostly_positive", rng.standard_normal(n) + 0.3
    yield "negative_sum", -(rng.standard_normal(n) ** 2)
    yield "wide_range", rng.standard_normal(n) ** 2 * 10.0 ** rng.integers(-30, 30, n)
    yield "decaying", rng.standard_normal(n) ** 2 * np.exp(-np.arange(n) / 2000.0)
    yield "growing", rng.standard_normal(n) ** 2 * np.exp(np.arange(n) / 20000.0)
    yield "zeros", np.zeros(5000)
    yield "zeros_then_data", np.concatenate([np.zeros(1000), rng.standard_normal(20000) ** 2])
    yield "data_then_zeros", np.concatenate([rng.standard_normal(20000) ** 2, np.zeros(30000)])
    # all-zero cells are identity summaries (CG vectors of small support keep long zero prefixes)
    yield "long_zero_prefix", np.concatenate([np.zeros(20000 + 37), rng.standard_normal(12000) ** 2])
    yield "zero_holes", np.where((np.arange(n) // 1000) % 3 == 1, 0.0, rng.standard_normal(n))
    yield "negative_zeros", np.concatenate([np.full(5000, -0.0), -(rng.standard_normal(3000) ** 2), np.full(7000, -0.0), rng.standard_normal(3000)])
    yield "only_negative_zeros", np.full(4000, -0.0)
    yield "zeros_cancel_zeros", np.concatenate([np.zeros(300), [1.5, -1.5], np.full(3000, -0.0), [2.0 ** -1060, -(2.0 ** -1060)], np.zeros(900), [7.0]])
    yield "subnormal", rng.standard_normal(40000) ** 2 * 5e-324 * 1000
    yield "tiny_and_huge", np.where(rng.random(n) < 0.001, 1e300, 1e-300) * rng.random(n)
    yield "ties", rng.integers(0, 8, n) * 0.5 + 2.0 ** 40 * (np.arange(n) == 0)      # every element is k/2 ulp-ish of the sum
    yield "ties_exact", np.concatenate([[2.0 ** 53], np.ones(50000)])                # 2^53 + 1 is a tie every time
    yield "ties_half_ulp", np.concatenate([[1.0], np.full(50000, 2.0 ** -53)])
    yield "cancel_to_zero", np.concatenate([rng.standard_normal(10000), [0.0]]) * np.concatenate([np.ones(10000), [0.0]])
    x = rng.standard_normal(30000)
    yield "exact_cancellation", np.concatenate([x, -x[::-1]])
    yield "sawtooth_binade", np.tile(np.concatenate([np.full(300, 1.0 / 256), np.full(300, -1.0 / 256)]), 200) + 0.0 + (np.arange(120000) == 0) * 1.0
    yield "powers_of_two", 2.0 ** rng.integers(-20, 20, n).astype(np.float64)
    yield "inf", np.concatenate([rng.random(3000), [np.inf], rng.random(3000)])
    yield "inf_minus_inf", np.concatenate([rng.random(3000), [np.inf], rng.random(300), [-np.inf], rng.random(3000)])
    yield "nan", np.concatenate([rng.random(3000), [np.nan], rng.random(3000)])
    yield "overflow", np.full(5000, 1e308)
    yield "one_element", np.array([3.25])
    yield "empty", np.array([])
    for m in (1, 2, 127, 128, 129, 8191, 8192, 8193, 524287, 524288, 524289):
        yield f"len_{m}", rng.standard_normal(m) ** 2


@pytest.mark.parametrize("name,t", list(cases()), ids=[c[0] for c in cases()])
def test_model_equals_sequential(model, name, t):
    got, want, st = run(model, t)
    assert same_bits(got, want), (name, got, want, st)


def test_model_uses_the_parallel_path(model):
    """On CG-like data (sums of squares / absolute values) nearly every group goes through the integer path."""
    rng = np.random.default_rng(7)
    t = rng.standard_normal(1 << 22) ** 2
    got, want, st = run(model, t)
    assert same_bits(got, want)
    groups = (t.size // 128 + 63) // 64
    assert st[0] >= groups - 64, st       # all but a few dozen groups applied whole
    assert st[2] <= 128, st               # element-by-element cells: binade crossings and ties only


def test_random_trials(model):
    rng = np.random.default_rng(99)
    for trial in range(200):
        n = int(rng.integers(1, 40000))
        kind = trial % 4
        if kind == 0:
            t = rng.standard_normal(n) ** 2
        elif kind == 1:
            t = rng.standard_normal(n) * 2.0 ** rng.integers(-8, 8, n)
        elif kind == 2:
            t = np.round(rng.standard_normal(n) * 1024) / 1024      # many exactly representable ties
        else:
            t = rng.standard_normal(n) + 1.0
        got, want, st = run(model, t)
        assert same_bits(got, want), (trial, n, got, want, st)
