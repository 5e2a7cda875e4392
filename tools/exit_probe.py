"""Bisect a crash at interpreter exit of the multi-rank bench: python tools/exit_probe.py <stage>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stage = int(sys.argv[1])
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
if stage >= 1:
    cga.check(cga.lib().cgh_comm_preload(), "preload")
if stage >= 2:
    import torch  # noqa: F401
    import torch.distributed as dist
if stage >= 3:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + stage))
    dist.init_process_group("gloo", rank=0, world_size=1)
if stage >= 4:
    ids = cga.comm_unique_id()
    cga.comm_init(1, 0, ids)
if stage >= 5:
    n = 64 * 64 * 64
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    D = cga.DistRank.poisson(64, 64, 64, 7, 1, 1, 0, b)
    D.start(b, x, -1.0, 12)
    D.iterate(10, 0)
    cga.check(cga.lib().cgh_sync())
    if stage >= 6:
        D.free()
if stage >= 4:
    cga.lib().cgh_comm_destroy()
if stage >= 3:
    dist.destroy_process_group()
print("stage", stage, "done", flush=True)
