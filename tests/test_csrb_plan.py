"""CPU-only: the host-side planner of the column-blocked CSR (cgh_csrb_plan_host, spmv_csrb.hip) deals 64-row segments
to wave-blocks of equal capacity and about equal weight.  No reference counterpart (the reference has one format per
SpMV and no GPU); the properties below are what the kernel relies on."""
import numpy as np
import pytest

import conjugate_gradient_amd as cga


def _check(weights, spb):
    w = np.asarray(weights, dtype=np.int32)
    nb, sb, ss = cga.csrb_plan(w, spb)
    assert nb == (w.size + spb - 1) // spb
    assert sb.min() >= 0 and sb.max() < nb and ss.min() >= 0 and ss.max() < spb
    # every (block, slot) is used at most once, every segment exactly once
    assert np.unique(sb.astype(np.int64) * spb + ss).size == w.size
    load = np.bincount(sb, weights=w, minlength=nb)
    used = np.bincount(sb, minlength=nb)
    assert used.max() <= spb
    return load, used


def test_every_segment_placed_once_and_capacity_respected():
    rng = np.random.default_rng(1)
    for nseg, spb in ((1, 16), (15, 16), (16, 16), (17, 16), (1000, 8), (4097, 32)):
        _check(rng.integers(0, 5000, nseg), spb)


def test_heavy_tailed_weights_are_balanced():
    """segments of a power-law matrix: weights from ~300 to tens of thousands, trend along the rows included"""
    rng = np.random.default_rng(2)
    nseg, spb = 65536, 16
    w = (64 * 4.5 * (1.0 + 3.0 * np.arange(nseg) / nseg)).astype(np.int64) + np.minimum(3000, (30.0 / rng.random(nseg))).astype(np.int64)
    load, used = _check(w, spb)
    assert used.min() == spb                       # nseg divides evenly: every block is full
    spread = (load.max() - load.min()) / load.mean()
    assert spread < 0.05, spread                   # the pacing of the sweep relies on near-equal streams
    naive = w.reshape(-1, spb).sum(1)              # contiguous blocks for comparison
    assert (naive.max() - naive.min()) / naive.mean() > 10 * spread


def test_equal_weights_keep_segment_order_and_determinism():
    w = np.full(64, 7, dtype=np.int32)
    nb, sb, ss = cga.csrb_plan(w, 4)
    nb2, sb2, ss2 = cga.csrb_plan(w, 4)
    assert np.array_equal(sb, sb2) and np.array_equal(ss, ss2)
    assert nb == 16 and np.all(np.bincount(sb) == 4)


def test_degenerate_inputs():
    assert cga.lib().cgh_csrb_plan_host(0, None, 16, None, None) == 0
    load, used = _check(np.zeros(100, dtype=np.int32), 16)   # all-empty segments (a diagonal matrix)
    assert load.sum() == 0 and used.sum() == 100
