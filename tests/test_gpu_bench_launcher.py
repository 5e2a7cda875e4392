"""The driver's multi-GPU contract, rehearsed on the ONE GPU of the test box: `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...` must print exactly ONE JSON line (rank 0) with the contract's keys.
RCCL refuses two ranks on one device, so the ranks bootstrap through files (CGH_BENCH_BOOTSTRAP=files) and share
device 0 (CGH_BENCH_SHARE_GPU=1); everything after the bootstrap -- partitioned generation, the three-kernel
iteration with its peer-to-peer halo and all-reduces, barriers, max-over-ranks timing, the JSON line -- is the code
the 2/4/8-GPU runs execute.  The numbers mean nothing here (the ranks share one GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("N", [2, 3])
def test_launcher_contract_with_ranks_sharing_the_gpu(N):
    env = dict(os.environ, CGH_BENCH_BOOTSTRAP="files", CGH_BENCH_SHARE_GPU="1", CGH_DIST_GRID_MAX="64")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(N), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--steps", "30", "--warmup", "5",
           "--workload", "p7_128", "--no-config4"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == N and d["steps"] == 30 and d["warmup"] == 5 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["rows"] == 128 ** 3 and d["config"]["halo_exchange"] == "p2p"
    assert d["config"]["scalar_allreduce"].startswith("p2p")
    first, last = d["config"]["residual_l1_half_first_last"]
    assert 0 < last < first                                   # the partitioned iteration converges
    assert abs(d["value"] - 30 / (d["ms_per_step"] * 30e-3)) < 1e-6 * d["value"]

    # contract-complete for a driver-run multi-GPU record: the reference's CPU path beside it (rank 0), and one
    # self-diagnosis line per rank on stderr (transport chosen and why, all-reduce latency alone, halo cost inside the SpMV)
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] in ("reference", "port") and "compiled with" in cb["sample"]
    diags = [json.loads(ln.split("[cg_hip bench_dist] ", 1)[1]) for ln in p.stderr.splitlines() if ln.startswith("[cg_hip bench_dist] ")]
    assert sorted(g["rank"] for g in diags) == list(range(N))
    for g in diags:
        assert "diagnostics_failed" not in g, g
        assert g["scalar_allreduce"].startswith("p2p") and g["halo_exchange"] == "p2p" and g["p2p_error_word"] == 0
        assert g["allreduce_alone_us_per_exchange"]["p2p_mailbox_us"] > 0
        assert g["own_spmv_ms"] > 0 and g["same_slab_without_neighbours_spmv_ms"] > 0
