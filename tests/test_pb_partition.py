"""CPU-only: the host-side partition of the two-phase SpMV layout (cgh_pb_partition_host, spmv_pb.hip) cuts the rows,
in units of 256, into consecutive blocks of bounded span holding about equal numbers of entries."""
import numpy as np

import conjugate_gradient_amd as cga


def _check(cum, max_groups, target):
    bnd = cga.pb_partition(cum, max_groups, target)
    G = len(cum) - 1
    assert bnd[0] == 0 and bnd[-1] == G and np.all(np.diff(bnd) >= 1) and np.all(np.diff(bnd) <= max_groups)
    w = cum[bnd[1:]] - cum[bnd[:-1]]
    single = np.diff(bnd) == 1
    assert np.all(w[~single] <= target)                  # only a single group may be heavier than the target
    # greedy: a block stops short of the span cap only because one more group would overshoot the target
    for k in range(len(bnd) - 2):
        if bnd[k + 1] - bnd[k] < max_groups:
            assert cum[bnd[k + 1] + 1] - cum[bnd[k]] > target
    return bnd


def test_partition_uniform_and_skewed():
    rng = np.random.default_rng(11)
    cum = np.concatenate([[0], np.cumsum(np.full(1000, 2300))])
    b = _check(cum, 64, 131072)
    assert np.all(np.diff(b)[:-1] == 131072 // 2300)
    w = (rng.pareto(1.5, 5000) * 500).astype(np.int64) + 10
    cum = np.concatenate([[0], np.cumsum(w)])
    for mg, t in ((64, 131072), (32, 20000), (1, 1), (64, 10 ** 12)):
        _check(cum, mg, t)
    assert len(cga.pb_partition(cum, 64, 10 ** 12)) - 1 == (5000 + 63) // 64   # span cap only


def test_partition_degenerate():
    assert list(cga.pb_partition(np.array([0, 0]), 64, 100)) == [0, 1]           # one empty group
    assert list(cga.pb_partition(np.array([0, 5, 5, 5, 9]), 2, 0)) == [0, 1, 3, 4]   # empty groups join for free
    assert cga.lib().cgh_pb_partition_host(0, None, 64, 100, None) == 0
