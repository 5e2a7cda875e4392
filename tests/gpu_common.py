"""Shared bits of the -m gpu parity tests: they call the HIP path through the C ABI
(conjugate_gradient_amd.binding -> libcg_hip.so) and compare with the oracle / goldens."""
import os

import numpy as np

import foamio

GOLDENS = ["cylinder_2000", "cylinder_8000", "cylinder_32k",
           "poisson7_16", "poisson7_32x24x20", "poisson27_12x10x8"]


def inputs(tag):
    if tag.startswith("cylinder_"):
        return foamio.foam(tag.split("_")[1])
    if tag == "poisson7_16":
        return foamio.poisson_coo(16, 16, 16, 0x5EED0001, 7)
    if tag == "poisson7_32x24x20":
        return foamio.poisson_coo(32, 24, 20, 0x5EED0001, 7)
    return foamio.poisson_coo(12, 10, 8, 0x5EED0001, 27)


def golden(tag):
    return np.load(os.path.join(foamio.GOLDEN, "golden_%s.npz" % tag))


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def same_bits_nan(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return (a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
            and np.array_equal(bits(np.nan_to_num(a)), bits(np.nan_to_num(b))))


def setknob(monkeypatch, name, value):
    """set a CGH_* environment knob for this test and have the library read its knobs again"""
    import conjugate_gradient_amd as cga
    monkeypatch.setenv(name, str(value))
    cga.check(cga.lib().cgh_config_reload(), "cgh_config_reload")
