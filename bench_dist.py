"""bench_dist.py -- the N>1 leg of bench.py: one rank per GPU (torch.distributed.run), the benchmark matrix
row-block (z-slab) partitioned over the ranks, halo exchange + all-reduces over RCCL inside libcg_hip.
The launcher (python -m torch.distributed.run) only provides RANK / WORLD_SIZE / LOCAL_RANK: the ranks of the one node
hand the ncclUniqueIds over through a file in /tmp named after the launcher's pid, and every later exchange (barriers,
max-over-ranks) is an RCCL all-reduce of the library's own communicator.  torch is not imported here: nothing on this path needs it (its wheel brings a second HIP
runtime and RCCL into the process; harmless since the library loads its RCCL RTLD_LOCAL, DESIGN.md section 1, but dead weight).

The JSON line's `value` is the workload bench.py was asked for (default p7_256: the 16.7M-row matrix of BASELINE's
north-star target split over the N GPUs, strong scaling).  For the default workload the line also carries
`config4`: BASELINE config 4 (27-point 512^3, 134M rows, 42 GB of ELL) partitioned over the same N GPUs."""
import json
import os
import sys
import time

import numpy as np

import conjugate_gradient_amd as cga

HBM_PEAK_GBS = 8000.0


def launcher_start_time():
    """wall-clock start of the launcher process (the parent of every rank): files of THIS run are younger than it"""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])      # field 22: starttime, in clock ticks since boot
        with open("/proc/stat") as f:
            boot = next(int(line.split()[1]) for line in f if line.startswith("btime"))
        return boot + ticks / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, StopIteration):
        return 0.0


def run_key():
    """names the files of one launch: the launcher's pid AND start time (a recycled pid differs in the second), its
    rendezvous port and, if the launcher is torchrun, its run id"""
    rid = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "") if ch.isalnum())[:16]
    return "%d_%d_%s_%s" % (os.getppid(), int(launcher_start_time()), os.environ.get("MASTER_PORT", "0"), rid)


def read_fresh(path, size):
    """the file's bytes if it is complete and was written during this launch, else None (a leftover of a crashed run
    with the same key would otherwise hand a dead id to ncclCommInitRank, which then hangs without a word)"""
    try:
        if os.stat(path).st_mtime + 1.0 < launcher_start_time():
            return None
        with open(path, "rb") as f:
            data = f.read()
        return data if len(data) == size else None
    except FileNotFoundError:
        return None


def publish_exclusive(path, data):
    """create-or-fail (O_EXCL) under a temporary name, then rename: nobody reads half a file, nobody follows a planted link"""
    for stale in (path, path + ".tmp"):
        try:
            os.unlink(stale)
        except FileNotFoundError:
            pass
    fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(data)
    os.replace(path + ".tmp", path)


def exchange_unique_id(rank, world):
    """rank 0 creates the 256-byte id and publishes it atomically; the other ranks of the node wait for the file"""
    path = "/tmp/cgh_uid_%s" % run_key()
    if rank == 0:
        ident = cga.comm_unique_id()
        publish_exclusive(path, ident)
        return ident, path
    deadline = time.time() + 300.0
    while time.time() < deadline:
        ident = read_fresh(path, 256)
        if ident is not None:
            return ident, path
        time.sleep(0.02)
    raise SystemExit("bench_dist: rank %d never saw %s" % (rank, path))


def connect_without_rccl(rank, world):
    """CGH_BENCH_BOOTSTRAP=files: no RCCL at all -- the ranks hand their mailbox handles over through files and every
    later exchange (barriers, max-over-ranks, halo descriptors) goes through the mailboxes.  This is how the launcher
    protocol is rehearsed with several ranks on ONE GPU (tests/test_gpu_bench_launcher.py), which RCCL refuses."""
    import ctypes as C
    L = cga.lib()
    prefix = "/tmp/cgh_boot_%s_" % run_key()

    def publish(name, data):
        publish_exclusive(prefix + name, data)

    def collect(name, size):
        out = []
        for r in range(world):
            deadline = time.time() + 300.0
            while True:
                data = read_fresh(prefix + name + str(r), size)
                if data is not None:
                    break
                if time.time() > deadline:
                    raise SystemExit("bench_dist: rank %d never saw %s%s%d" % (rank, prefix, name, r))
                time.sleep(0.02)
            out.append(data)
        return b"".join(out)

    buf = C.create_string_buffer(64)
    cga.check(L.cgh_p2p_local_handle(buf), "cgh_p2p_local_handle")
    publish("handle%d" % rank, buf.raw)
    cga.check(L.cgh_p2p_connect(world, rank, collect("handle", 64)), "cgh_p2p_connect")
    publish("ready%d" % rank, b"1")          # every mailbox is zeroed and mapped before anybody sends
    collect("ready", 1)
    cga.check(L.cgh_comm_init_local(world, rank), "cgh_comm_init_local")
    barrier()
    for name in ("handle%d" % rank, "ready%d" % rank):
        os.unlink(prefix + name)


def barrier():
    cga.comm_allreduce([0.0], "sum")


def validate_p2p(rank, world, seed):
    """The peer-to-peer all-reduce passed the library's self-test at start-up; here the SOLVER runs 12 iterations of a
    small partitioned problem with it and 12 without (RCCL all-reduces): the residual histories must agree to rounding
    on every rank, or the benchmark goes on with RCCL.  Returns the path in force."""
    L = cga.lib()
    if not L.cgh_p2p_connected():
        return "rccl"
    shape = (64, 64, 8 * world)
    nl = shape[0] * shape[1] * 8
    hists = []
    for on in (1, 0):
        L.cgh_comm_set_p2p(on)
        b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
        D = cga.DistRank.poisson(shape[0], shape[1], shape[2], 7, seed, world, rank, b)
        D.start(b, x, -1.0, 14)
        D.iterate(12, 0)
        cga.check(L.cgh_sync())
        hists.append(D.history(0, 13))
        D.free()
    a, r = hists
    ok = bool(np.all(np.isfinite(a[:, :2]))) and bool(np.all(r[:, :2] != 0.0)) and float(np.max(np.abs(a[:, :2] - r[:, :2]) / np.abs(r[:, :2]))) < 1e-9
    ok = cga.comm_allreduce([1.0 if ok else 0.0], "min")[0] == 1.0
    L.cgh_comm_set_p2p(1 if ok else 0)
    return "p2p" if ok else "rccl (p2p disagreed with the RCCL all-reduce in a 12-iteration check)"


def probe_transports(rank, world, shape, seed):
    """Both transports on the benchmark's own matrix, 60 timed iterations each (after 10): the timed run then uses the
    faster one, and the line reports both.  Returns (use_p2p, {"p2p_us_per_iter": .., "rccl_us_per_iter": ..})."""
    L = cga.lib()
    nx, ny, nz, points = shape
    nl = nx * ny * (nz * (rank + 1) // world - nz * rank // world)
    us = {}
    for name, on in (("p2p", 1), ("rccl", 0)):
        L.cgh_comm_set_p2p(on)
        b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
        D = cga.DistRank.poisson(nx, ny, nz, points, seed, world, rank, b)
        D.start(b, x, -1.0, 72)
        D.iterate(10, 0)
        cga.check(L.cgh_sync())
        barrier()
        t0 = time.perf_counter()
        D.iterate(60, 0)
        cga.check(L.cgh_sync())
        barrier()
        dt = cga.comm_allreduce([time.perf_counter() - t0], "max")[0]
        us[name + "_us_per_iter"] = dt / 60 * 1e6
        D.free()
    use = us["p2p_us_per_iter"] <= us["rccl_us_per_iter"]
    L.cgh_comm_set_p2p(1 if use else 0)
    return use, us


def measure(rank, world, shape, seed, steps, warmup, p2p_in_force):
    """One partitioned workload: generate this rank's z-slab, warm up, time `steps` iterations (barrier + device sync on
    both sides, MAX over ranks).  Returns a dict on every rank, or None if some rank could not set the problem up."""
    nx, ny, nz, points = shape
    nl = nx * ny * (nz * (rank + 1) // world - nz * rank // world)
    ok = 1
    try:
        b = cga.DeviceVector(nl)
        x = cga.DeviceVector(nl)
        D = cga.DistRank.poisson(nx, ny, nz, points, seed, world, rank, b)
    except Exception as e:  # noqa: BLE001  (reported; the collective part below is skipped on every rank)
        print("[rank %d] set-up of %dx%dx%d/%d-point failed: %s" % (rank, nx, ny, nz, points, e), flush=True)
        ok = 0
    if cga.comm_allreduce([ok], "min")[0] == 0:
        return None
    halo = "p2p" if cga.lib().cgh_dist_halo_is_p2p(D.h) else "rccl"
    spmv_steps = min(steps, 50)
    cap = warmup + steps + spmv_steps + 1
    D.start(b, x, -1.0, cap)                    # tol < 0: never converges, every step does full work
    D.iterate(warmup, 0)
    cga.check(cga.lib().cgh_sync())
    barrier()
    t0 = time.perf_counter()
    D.iterate(steps, 0)                         # timed region: exactly K iterations
    cga.check(cga.lib().cgh_sync())
    barrier()
    dt = time.perf_counter() - t0
    # The SpMV's own duration comes from a second pass of the same iterations with HIP events around every SpMV: at 2M
    # rows per GPU an iteration is ~57 us, and two event records per iteration cost 6 us of it (measured,
    # tools/quick_dist_loop.py with TIME_KERNELS=1) -- they stay out of the region `value` is computed from.
    D.iterate(spmv_steps, 1)
    cga.check(cga.lib().cgh_sync())
    barrier()
    spmv_ms = D.spmv_ms()
    own = {"ms_per_step": dt / steps * 1e3, "spmv_ms": spmv_ms, "rows": int(nl), "halo_entries": int(D.n_halo)}
    dt, spmv_ms = cga.comm_allreduce([dt, spmv_ms], "max")      # slowest rank
    nnz_total, halo_total = cga.comm_allreduce([float(D.nnz_offdiag), float(D.n_halo)], "sum")   # exact: far below 2^53
    hist = D.history(0, cap)
    finite = bool(np.all(np.isfinite(hist[1:, 0]))) and (not p2p_in_force or cga.lib().cgh_p2p_error() == 0)
    finite = cga.comm_allreduce([1.0 if finite else 0.0], "min")[0] == 1.0     # one verdict for all ranks
    n = nx * ny * nz
    nnz_off = int(nnz_total)
    alg_spmv = 12 * nnz_off + 24 * n
    D.free()
    return {"finite": finite, "halo": halo, "n": n, "nnz_off": nnz_off, "alg_spmv": alg_spmv, "dt": dt, "spmv_ms": spmv_ms, "steps": steps,
            "ms_step": dt / steps * 1e3, "achieved": alg_spmv / (spmv_ms * 1e-3) / 1e9,
            "res": [float(hist[0, 0]), float(hist[-1, 0])], "own": own, "halo_total": int(halo_total)}


def spmv_alone_ms(shape, rank, world, seed):
    """this rank's slab as a matrix of its own (no neighbours, no halo columns): SpMV time inside a plain one-GPU loop.
    The difference to the partitioned loop's SpMV is what the halo costs INSIDE the SpMV (deposit reads + exposed waits)."""
    nx, ny, nz, points = shape
    nzl = nz * (rank + 1) // world - nz * rank // world
    nl = nx * ny * nzl
    b = cga.DeviceVector(nl); x = cga.DeviceVector(nl)
    A = cga.Matrix.poisson(nx, ny, nzl, points, seed, cga.LAYOUT_ELL, b)
    sv = cga.Solver(A, 72)
    sv.start(b, x, tol=-1.0, strict=0)
    sv.iterate(10, 0)
    sv.iterate(60, 1)
    ms = sv.spmv_ms()
    sv.free(); A.free()
    return ms


def allreduce_latencies(no_rccl):
    """the two scalar all-reduces of an iteration, alone: 1000 exchanges back to back per transport, us per exchange"""
    import ctypes as C
    L = cga.lib()
    out = {}
    v = C.c_double()
    if L.cgh_p2p_connected():
        rc = L.cgh_comm_bench_allreduce(1000, 1, C.byref(v))
        out["p2p_mailbox_us"] = v.value if rc == 0 else "failed: " + L.cgh_last_error().decode()
    if not no_rccl:
        rc = L.cgh_comm_bench_allreduce(1000, 0, C.byref(v))
        out["rccl_us"] = v.value if rc == 0 else "failed: " + L.cgh_last_error().decode()
    return out


def run(args, rank, world, local_rank, shape, seed):
    nx, ny, nz, points = shape
    no_rccl = os.environ.get("CGH_BENCH_BOOTSTRAP") == "files"
    if no_rccl:
        connect_without_rccl(rank, world)
    else:
        ident, path = exchange_unique_id(rank, world)
        # RCCL prints its version banner on stdout when the communicator comes up; stdout is for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            cga.comm_init(world, rank, ident)
            barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        if rank == 0:
            os.unlink(path)

    path = "p2p (no RCCL in this run: CGH_BENCH_BOOTSTRAP=files)" if no_rccl else validate_p2p(rank, world, seed)
    p2p = path.startswith("p2p")
    probe = None
    if p2p and not no_rccl:
        p2p, probe = probe_transports(rank, world, shape, seed)
        if not p2p:
            path = "rccl (faster than p2p in the probe)"
    m = measure(rank, world, shape, seed, args.steps, args.warmup, p2p)
    if m is None:
        raise SystemExit("bench_dist: the workload could not be set up on every rank")
    if not m["finite"] and p2p and not no_rccl:
        # the peer-to-peer reductions misbehaved under load after passing both checks: measure again over RCCL
        cga.lib().cgh_comm_set_p2p(0)
        p2p, path = False, "rccl (the run with p2p reductions produced non-finite residuals)"
        m = measure(rank, world, shape, seed, args.steps, args.warmup, p2p)
    if m is None or not m["finite"]:
        raise SystemExit("bench_dist: non-finite residual: the iteration did not run")
    # Self-diagnosis, one line per rank on stderr (a driver-run record can be read without a second run): which transport
    # carries the reductions and the halo and why, both transports' probe timings, the two all-reduce latencies measured
    # alone, this rank's own iteration and SpMV time, and what the halo costs inside the SpMV (against the same slab as a
    # matrix without neighbours).
    try:
        lat = allreduce_latencies(no_rccl)
        alone = spmv_alone_ms(shape, rank, world, seed)
        diag = {"rank": rank, "of": world, "device": local_rank, "scalar_allreduce": path, "halo_exchange": m["halo"],
                "transport_probe_us_per_iter": probe, "allreduce_alone_us_per_exchange": lat,
                "p2p_error_word": int(cga.lib().cgh_p2p_error()), "last_library_message": cga.lib().cgh_last_error().decode()[:200],
                "own_ms_per_step": m["own"]["ms_per_step"], "slowest_ms_per_step": m["ms_step"],
                "own_spmv_ms": m["own"]["spmv_ms"], "same_slab_without_neighbours_spmv_ms": alone,
                "halo_cost_inside_spmv_us": (m["own"]["spmv_ms"] - alone) * 1e3,
                "rows": m["own"]["rows"], "halo_entries": m["own"]["halo_entries"]}
    except Exception as e:  # noqa: BLE001  (a diagnostic must not cost the run its line)
        diag = {"rank": rank, "diagnostics_failed": str(e)}
    barrier()
    sys.stderr.flush()
    os.write(2, ("[cg_hip bench_dist] " + json.dumps(diag) + "\n").encode())   # one write: the ranks' lines do not interleave
    extra = None
    if args.workload == "p7_256" and not args.no_config4:
        extra = measure(rank, world, (512, 512, 512, 27), seed, min(args.steps, 20), min(args.warmup, 3), p2p)
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and m["n"] <= (1 << 25):
        # the reference's CPU path beside it (rank 0 only, the others wait at the last barrier): the WHOLE matrix, generated
        # on this rank's GPU once the timed regions are over, handed to the same routine the one-GPU line uses
        try:
            import bench
            bb = cga.DeviceVector(m["n"])
            A1 = cga.Matrix.poisson(nx, ny, nz, points, seed, cga.LAYOUT_ELL, bb)
            cpu = bench.cpu_baseline(A1, bb, budget_s=8.0)
            A1.free()
            del bb
        except Exception as e:  # noqa: BLE001
            cpu = {"error": str(e)}
    if rank == 0:
        out = {
            "metric": "CG iters/sec + SpMV achieved HBM GB/s (% of peak): value = CG iterations/s, the SpMV figures are in roofline",
            "value": args.steps / m["dt"], "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": m["ms_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: 3-D %d-point Poisson %dx%dx%d, %d rows, slot-major ELL, unpreconditioned CG, "
                                   "z-slab row blocks over %d GPUs, %s" % (
                                       args.workload, points, nx, ny, nz, m["n"], world,
                                       "3 kernels per iteration on one stream: p-halo pushed over xGMI by the p-update kernel into "
                                       "tagged deposits the SpMV reads (halo rows last), the 2 scalar all-reduces folded into the "
                                       "prologues of the kernels that consume them (RCCL: bootstrap and fallback)"
                                       if p2p else
                                       "p-halo by ncclSend/Recv overlapped with the interior SpMV, 2 ncclAllReduce per iteration"),
                       "rows": m["n"], "nnz_offdiag": m["nnz_off"], "parallelism": "row-block x%d" % world,
                       "scalar_allreduce": path, "halo_exchange": m["halo"], "transport_probe": probe,
                       "iter_GBs_algorithmic": (12 * m["nnz_off"] + 96 * m["n"]) / (m["ms_step"] * 1e-3) / 1e9,
                       "residual_l1_half_first_last": m["res"]},
            "roofline": {"bound": "hbm", "kernel": "spmv_ell_r2 (all local rows, halo rows last, incl. exposed halo wait; slowest rank; HIP events in a second pass of min(steps, 50) iterations)",
                         "achieved": m["achieved"], "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": m["achieved"] / (HBM_PEAK_GBS * world), "bytes_per_launch": int(m["alg_spmv"]),
                         "ms_per_launch": m["spmv_ms"], "traffic": None},
        }
        # HBM bytes per launch: the PMC figure of the one-GPU launch of the same kernel family (separate rocprofv3 passes,
        # profiles/) -- the N slabs together read the same matrix and vectors -- plus the 16 bytes per halo entry the
        # deposit reads add (two tagged words); counters cannot be collected inside a driver-run multi-rank launch
        root = os.path.dirname(os.path.abspath(__file__))
        for rnd in ("r03", "r02", "r01"):
            tpath = os.path.join(root, "profiles", "%s_traffic_%s.json" % (rnd, args.workload))
            if os.path.exists(tpath):
                with open(tpath) as f:
                    one = json.load(f)["hbm_bytes_per_launch"]
                out["roofline"]["traffic"] = int(one + 16 * m["halo_total"])
                out["roofline"]["traffic_source"] = ("profiles/%s (one-GPU PMC passes of the same kernel family: %d bytes) + 16 bytes x %d halo entries "
                                                     "read from the deposits; not collected in this run" % (os.path.basename(tpath), one, m["halo_total"]))
                out["roofline"]["frac_real_bytes"] = out["roofline"]["traffic"] / (m["spmv_ms"] * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)
                break
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if extra is not None and extra["finite"]:
            out["config4"] = {"workload": "p27_512: 3-D 27-point Poisson 512^3, %d rows, %d off-diagonals, z-slabs over %d GPUs"
                                          % (extra["n"], extra["nnz_off"], world),
                              "value": extra["steps"] / extra["dt"], "unit": "iterations/s", "steps": extra["steps"],
                              "ms_per_step": extra["ms_step"],
                              "roofline": {"bound": "hbm", "kernel": "spmv_ell_r2 (K=26, all local rows of the slowest rank, halo unpacked first; HIP events around every SpMV in a second pass)",
                                           "achieved": extra["achieved"], "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                           "frac": extra["achieved"] / (HBM_PEAK_GBS * world),
                                           "bytes_per_launch": int(extra["alg_spmv"]), "ms_per_launch": extra["spmv_ms"],
                                           "traffic": None},
                              "iter_GBs_algorithmic": (12 * extra["nnz_off"] + 96 * extra["n"]) / (extra["ms_step"] * 1e-3) / 1e9,
                              "one_gpu_reference": "profiles/r03_bench_p27_512_n1.json (124 it/s on one MI355X)"}
        print(json.dumps(out), flush=True)
    barrier()
    cga.lib().cgh_comm_destroy()
