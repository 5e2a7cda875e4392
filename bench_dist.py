"""bench_dist.py -- the N>1 leg of bench.py: one rank per GPU (torch.distributed.run), the benchmark matrix
row-block (z-slab) partitioned over the ranks, halo exchange + all-reduces over RCCL inside libcg_hip.
torch.distributed (gloo) is used only for the rendezvous: broadcast of the ncclUniqueIds, barriers, max-over-ranks."""
import json
import os
import time

import numpy as np

import conjugate_gradient_amd as cga

HBM_PEAK_GBS = 8000.0


def run(args, rank, world, local_rank, shape, seed):
    import torch
    import torch.distributed as dist

    nx, ny, nz, points = shape
    dist.init_process_group("gloo", rank=rank, world_size=world)   # MASTER_ADDR/PORT from the launcher's env
    ids = [cga.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    cga.comm_init(world, rank, ids[0])

    nl = nx * ny * (nz * (rank + 1) // world - nz * rank // world)
    b = cga.DeviceVector(nl)
    x = cga.DeviceVector(nl)
    D = cga.DistRank.poisson(nx, ny, nz, points, seed, world, rank, b)
    cap = args.warmup + args.steps + 1
    D.start(b, x, -1.0, cap)                    # tol < 0: never converges, every step does full work
    D.iterate(args.warmup, 0)
    cga.check(cga.lib().cgh_sync())
    dist.barrier()
    t0 = time.perf_counter()
    D.iterate(args.steps, 1)                    # timed region: exactly K iterations
    cga.check(cga.lib().cgh_sync())
    dist.barrier()
    dt = time.perf_counter() - t0
    spmv_ms = D.spmv_ms()
    t = torch.tensor([dt, spmv_ms], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)     # slowest rank
    dt, spmv_ms = t.tolist()
    nnz = torch.tensor([D.nnz_offdiag], dtype=torch.int64)
    dist.all_reduce(nnz)
    hist = D.history(0, cap)
    assert np.all(np.isfinite(hist[1:, 0])), "non-finite residual: the iteration did not run"
    if rank == 0:
        n = nx * ny * nz
        nnz_off = int(nnz.item())
        alg_spmv = 12 * nnz_off + 24 * n
        achieved = alg_spmv / (spmv_ms * 1e-3) / 1e9
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "CG iterations/sec (SpMV achieved HBM GB/s in roofline)",
            "value": args.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: 3-D %d-point Poisson %dx%dx%d, %d rows, slot-major ELL, unpreconditioned CG, "
                                   "z-slab row blocks over %d GPUs, p-halo by ncclSend/Recv overlapped with interior SpMV, "
                                   "2 ncclAllReduce per iteration" % (args.workload, points, nx, ny, nz, n, world),
                       "rows": n, "nnz_offdiag": nnz_off, "parallelism": "row-block x%d" % world,
                       "iter_GBs_algorithmic": (12 * nnz_off + 96 * n) / (ms_step * 1e-3) / 1e9,
                       "residual_l1_half_first_last": [float(hist[0, 0]), float(hist[-1, 0])]},
            "roofline": {"bound": "hbm", "kernel": "spmv_ell_r2 (interior + boundary launches, incl. exposed halo wait; slowest rank)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "bytes_per_launch": int(alg_spmv),
                         "ms_per_launch": spmv_ms, "traffic": None},
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    cga.lib().cgh_comm_destroy()
    dist.destroy_process_group()
