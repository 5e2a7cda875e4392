"""Bisect the abort at interpreter exit seen when this library, RCCL and PyTorch share a process.

    python tools/exit_probe.py <case>

Cases (each prints the HIP / HSA / RCCL copies mapped into the process just before it exits):
  init            cgh_init only
  init_torch      cgh_init, then `import torch`                      (two HIP runtimes, each in its own scope)
  global_torch    cgh_init, ROCm's librccl dlopen'ed RTLD_GLOBAL, then `import torch`   (what cgh_comm_preload did)
  preload_torch   cgh_init, cgh_comm_preload (RTLD_LOCAL now), then `import torch`
  torch_init      `import torch`, then cgh_init                        (one HIP runtime: torch's, matched by SONAME)
  full            preload_torch + gloo group + communicator + a 10-iteration distributed solve
"""
import atexit
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
case = sys.argv[1]


def mapped():
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.split()[-1]
            b = os.path.basename(p)
            if any(k in b for k in ("amdhip64", "librccl", "hsa-runtime", "libcg_hip")) and p not in seen:
                seen.append(p)
    for p in seen:
        print("  mapped", p, flush=True)


atexit.register(mapped)

if case == "torch_init":
    import torch  # noqa: F401
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
if case == "global_torch":
    ctypes.CDLL("/opt/rocm/lib/librccl.so.1", mode=ctypes.RTLD_GLOBAL)
if case in ("preload_torch", "full"):
    cga.check(cga.lib().cgh_comm_preload(), "preload")
if case in ("init_torch", "global_torch", "preload_torch", "full"):
    import torch  # noqa: F401,F811
    torch.zeros(4, device="cuda").sum().item()
if case == "full":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("gloo", rank=0, world_size=1)
    cga.comm_init(1, 0, cga.comm_unique_id())
    n = 64 * 64 * 64
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    D = cga.DistRank.poisson(64, 64, 64, 7, 1, 1, 0, b)
    D.start(b, x, -1.0, 12)
    D.iterate(10, 0)
    cga.check(cga.lib().cgh_sync())
    D.free()
    cga.lib().cgh_comm_destroy()
    dist.destroy_process_group()
print("case", case, "done", flush=True)
