"""ctypes binding of libcg_hip.so (include/cg_hip.h).  No fallback: a missing
library or a failing call raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcg_hip.so")

LAYOUT_ELL, LAYOUT_CSR, LAYOUT_SELL, LAYOUT_CSRB, LAYOUT_SELLS, LAYOUT_PB = 0, 1, 2, 3, 4, 5

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class CgOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tol", C.c_double), ("strict", C.c_int),
                ("check_every", C.c_int), ("fixed_iters", C.c_int), ("time_kernels", C.c_int), ("precond", C.c_int)]


class CgReport(C.Structure):
    _fields_ = [("ret", C.c_int), ("iterations", C.c_int), ("initial_res", C.c_double),
                ("final_res", C.c_double), ("r0", C.c_double * 5), ("solve_ms", C.c_double),
                ("spmv_ms_avg", C.c_double)]


_lib = None
_inited = None


def lib():
    """Load libcg_hip.so (built by `make -C conjugate_gradient_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libcg_hip.so is not built: run `make -C conjugate_gradient_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    i64 = C.c_int64
    L.cgh_init.argtypes = [C.c_int]
    L.cgh_last_error.restype = C.c_char_p
    L.cgh_alloc.argtypes = [C.c_size_t]
    L.cgh_alloc.restype = vp
    L.cgh_free.argtypes = [vp]
    L.cgh_free.restype = None
    for f in ("cgh_h2d", "cgh_d2h", "cgh_d2d"):
        getattr(L, f).argtypes = [vp, vp, C.c_size_t]
    L.cgh_memset.argtypes = [vp, C.c_int, C.c_size_t]
    L.cgh_matrix_from_ell_host.argtypes = [i64, C.c_int, dp, dp, ip]
    L.cgh_matrix_from_ell_host.restype = vp
    L.cgh_matrix_from_coo_host.argtypes = [i64, i64, dp, dp, ip, ip, C.c_int]
    L.cgh_matrix_from_coo_host.restype = vp
    L.cgh_matrix_from_coo_device.argtypes = [i64, i64, vp, vp, vp, vp, C.c_int]
    L.cgh_matrix_from_coo_device.restype = vp
    L.cgh_matrix_poisson.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, vp]
    L.cgh_matrix_poisson.restype = vp
    L.cgh_matrix_powerlaw.argtypes = [i64, C.c_int, C.c_uint64, C.c_int, vp]
    L.cgh_matrix_powerlaw.restype = vp
    L.cgh_matrix_min_row.argtypes = [vp]
    L.cgh_matrix_set_exact_rows.argtypes = [vp, C.c_int]
    L.cgh_matrix_dict_offsets.argtypes = [vp]
    L.cgh_matrix_dict_mode.argtypes = [vp]
    L.cgh_matrix_stage_doubles.argtypes = [vp]
    L.cgh_config_csrb.argtypes = [C.c_int] * 6
    L.cgh_config_sell_sigma.argtypes = [C.c_int]
    L.cgh_config_persistent.argtypes = [C.c_int]
    L.cgh_persistent_set_trace.argtypes = [C.c_void_p]
    L.cgh_time_read_variant.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.cgh_solver_is_resident.argtypes = [C.c_void_p]
    L.cgh_persistent_launches.restype = C.c_longlong
    L.cgh_csrb_plan_host.argtypes = [i64, ip, C.c_int, ip, ip]
    L.cgh_csrb_plan_host.restype = i64
    L.cgh_config_pb.argtypes = [C.c_int, i64, C.c_int, C.c_int]
    L.cgh_pb_partition_host.argtypes = [i64, C.POINTER(C.c_int64), C.c_int, i64, C.POINTER(C.c_int64)]
    L.cgh_pb_partition_host.restype = i64
    L.cgh_matrix_free.argtypes = [vp]
    L.cgh_matrix_free.restype = None
    for f in ("cgh_matrix_rows", "cgh_matrix_nnz_offdiag", "cgh_matrix_stored_bytes"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = i64
    for f in ("cgh_matrix_width", "cgh_matrix_layout"):
        getattr(L, f).argtypes = [vp]
    L.cgh_matrix_download_ell.argtypes = [vp, dp, ip]
    L.cgh_matrix_download_diag.argtypes = [vp, dp]
    L.cgh_matrix_download_csr.argtypes = [vp, C.POINTER(C.c_int64), ip, dp]
    L.cgh_spmv.argtypes = [vp, vp, vp]
    L.cgh_dot.argtypes = [vp, vp, i64, C.c_int, dp]
    L.cgh_l1norm.argtypes = [vp, i64, C.c_int, dp]
    L.cgh_axpy.argtypes = [vp, vp, C.c_double, vp, i64]
    L.cgh_cg_solve.argtypes = [vp, vp, vp, C.POINTER(CgOpts), C.POINTER(CgReport), dp]
    L.cgh_cg_solve_noalloc.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.c_int, vp, vp]
    L.cgh_solver_create.argtypes = [vp, C.c_int]
    L.cgh_solver_create.restype = vp
    L.cgh_solver_free.argtypes = [vp]
    L.cgh_solver_free.restype = None
    L.cgh_solver_start.argtypes = [vp, vp, vp, C.c_double, C.c_int]
    L.cgh_solver_set_precond.argtypes = [vp, C.c_int]
    L.cgh_solver_iterate.argtypes = [vp, C.c_int, C.c_int]
    L.cgh_solver_poll.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cgh_solver_history.argtypes = [vp, C.c_int, C.c_int, dp]
    L.cgh_solver_rebase.argtypes = [vp]
    L.cgh_solver_base.argtypes = [vp]
    L.cgh_solver_spmv_ms.argtypes = [vp, dp]
    L.cgh_solver_residual.argtypes = [vp]
    L.cgh_solver_residual.restype = vp
    i64p = C.POINTER(C.c_int64)
    L.cgh_partition_from_coo_host.argtypes = [i64, i64, dp, dp, ip, ip, C.c_int, C.c_int, i64p]
    L.cgh_partition_from_coo_host.restype = vp
    L.cgh_partition_free.argtypes = [vp]
    L.cgh_partition_free.restype = None
    for f in ("cgh_partition_rows", "cgh_partition_first_row", "cgh_partition_halo"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = i64
    for f in ("cgh_partition_width", "cgh_partition_neighbours"):
        getattr(L, f).argtypes = [vp]
    L.cgh_partition_interior.argtypes = [vp, i64p, i64p]
    L.cgh_partition_interior.restype = None
    L.cgh_partition_neighbour.argtypes = [vp, C.c_int, ip, i64p, i64p, i64p]
    L.cgh_partition_copy_send_list.argtypes = [vp, C.c_int, ip]
    L.cgh_partition_copy_local.argtypes = [vp, dp, dp, ip, i64p]
    L.cgh_comm_unique_id.argtypes = [C.c_char_p]
    L.cgh_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p]
    L.cgh_comm_destroy.restype = None
    L.cgh_comm_allreduce_f64.argtypes = [dp, C.c_int, C.c_int]
    L.cgh_comm_bench_allreduce.argtypes = [C.c_int, C.c_int, dp]
    L.cgh_p2p_local_handle.argtypes = [C.c_char_p]
    L.cgh_comm_init_local.argtypes = [C.c_int, C.c_int]
    L.cgh_dist_halo_descriptor.argtypes = [vp, C.c_char_p]
    L.cgh_dist_halo_connect.argtypes = [vp, C.c_char_p]
    L.cgh_dist_halo_is_p2p.argtypes = [vp]
    L.cgh_dist_loopback.argtypes = [vp]
    L.cgh_p2p_connect.argtypes = [C.c_int, C.c_int, C.c_char_p]
    L.cgh_p2p_disconnect.restype = None
    L.cgh_p2p_allreduce_f64.argtypes = [dp, C.c_int]
    L.cgh_dist_from_partition.argtypes = [vp]
    L.cgh_dist_from_partition.restype = vp
    L.cgh_dist_poisson.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, vp]
    L.cgh_dist_poisson.restype = vp
    L.cgh_dist_free.argtypes = [vp]
    L.cgh_dist_free.restype = None
    for f in ("cgh_dist_rows", "cgh_dist_nnz_offdiag", "cgh_dist_halo"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = i64
    L.cgh_dist_width.argtypes = [vp]
    L.cgh_dist_download_local.argtypes = [vp, dp, dp, ip]
    L.cgh_dist_vector.argtypes = [vp, C.c_int]
    L.cgh_dist_vector.restype = vp
    L.cgh_dist_start.argtypes = [vp, vp, vp, C.c_double, C.c_int]
    L.cgh_dist_set_strict.argtypes = [vp, C.c_int]
    L.cgh_dist_iterate.argtypes = [vp, C.c_int, C.c_int]
    L.cgh_dist_poll.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.cgh_dist_history.argtypes = [vp, C.c_int, C.c_int, dp]
    L.cgh_dist_spmv_ms.argtypes = [vp, dp]
    vpp = C.POINTER(vp)
    L.cgh_dist_group_start.argtypes = [vpp, C.c_int, vpp, vpp, C.c_double, C.c_int]
    L.cgh_dist_group_iterate.argtypes = [vpp, C.c_int, C.c_int]
    L.cgh_time_spmv.argtypes = [vp, vp, vp, C.c_int, dp]
    L.cgh_time_copy.argtypes = [vp, vp, C.c_size_t, C.c_int, dp]
    L.cgh_time_copy_variant.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, dp]
    L.cgh_time_read.argtypes = [vp, vp, C.c_size_t, C.c_int, dp]
    L.cgh_time_read_8_4.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, dp]
    L.cgh_xsum_set_prof.argtypes = [vp]
    L.cgh_xsum_set_prof.restype = None
    L.cgh_alloc_calls.restype = C.c_longlong
    L.cgh_mem_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cgh_capture_end_launch.argtypes = [C.c_int]
    _lib = L
    return L


def check(rc, what="libcg_hip call"):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().cgh_last_error().decode()))


def config_csrb(rows_per_wave_block=0, slab_bits=0, wgs_per_cu=0, window=-1, quorum_div=0, lds_atomics=-1):
    """tuning of the column-blocked CSR matrices constructed after the call (see cgh_config_csrb in include/cg_hip.h)"""
    check(lib().cgh_config_csrb(rows_per_wave_block, slab_bits, wgs_per_cu, window, quorum_div, lds_atomics), "cgh_config_csrb")


def config_sell_sigma(sigma):
    check(lib().cgh_config_sell_sigma(sigma), "cgh_config_sell_sigma")


def config_pb(span_bits=0, target_entries=0, split=0, ordered=-1):
    """tuning of the two-phase (propagation-blocking) matrices constructed after the call (cgh_config_pb in include/cg_hip.h)"""
    check(lib().cgh_config_pb(span_bits, target_entries, split, ordered), "cgh_config_pb")


def pb_partition(cum, max_groups, target):
    """host-side partition of the two-phase layout -> block boundaries in groups of 256 rows"""
    c = np.ascontiguousarray(cum, dtype=np.int64)
    out = np.zeros(c.size, dtype=np.int64)
    B = lib().cgh_pb_partition_host(c.size - 1, c.ctypes.data_as(C.POINTER(C.c_int64)), max_groups, target,
                                    out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:B + 1]


def csrb_plan(weights, spb):
    """host-side planner of the column-blocked CSR -> (nblocks, seg_block, seg_slot)"""
    w = np.ascontiguousarray(weights, dtype=np.int32)
    sb = np.empty(w.size, dtype=np.int32)
    ss = np.empty(w.size, dtype=np.int32)
    nb = lib().cgh_csrb_plan_host(w.size, _I(w), spb, _I(sb), _I(ss))
    return nb, sb, ss


def have_gpu():
    return lib().cgh_device_count() > 0


def init(device=0):
    global _inited
    if _inited != device:
        check(lib().cgh_init(device), "cgh_init")
        _inited = device


def _D(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(dp)


def _I(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(ip)


class DeviceVector:
    """A device buffer of `count` elements of `dtype` (hipMalloc via the library)."""

    def __init__(self, count, dtype=np.float64):
        self.count = int(count)
        self.dtype = np.dtype(dtype)
        self.nbytes = self.count * self.dtype.itemsize
        self.ptr = lib().cgh_alloc(max(self.nbytes, 16))
        if not self.ptr:
            raise MemoryError(lib().cgh_last_error().decode())

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        v = cls(a.size, a.dtype)
        check(lib().cgh_h2d(v.ptr, a.ctypes.data, v.nbytes), "h2d")
        return v

    def to_host(self):
        out = np.empty(self.count, dtype=self.dtype)
        check(lib().cgh_d2h(out.ctypes.data, self.ptr, self.nbytes), "d2h")
        return out

    def zero(self):
        check(lib().cgh_memset(self.ptr, 0, self.nbytes), "memset")

    def free(self):
        if self.ptr:
            lib().cgh_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Matrix:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("matrix construction failed: " + lib().cgh_last_error().decode())
        self.h = handle

    @classmethod
    def from_ell(cls, n, K, diag, ev, ec):
        return cls(lib().cgh_matrix_from_ell_host(n, K, _D(diag), _D(ev), _I(ec)))

    @classmethod
    def from_coo(cls, n, diag, upper, rows, cols, layout=LAYOUT_ELL):
        return cls(lib().cgh_matrix_from_coo_host(n, rows.size, _D(diag), _D(upper), _I(rows), _I(cols), layout))

    @classmethod
    def poisson(cls, nx, ny, nz, points, seed, layout=LAYOUT_ELL, b=None):
        return cls(lib().cgh_matrix_poisson(nx, ny, nz, points, seed, layout, b.ptr if b is not None else None))

    @classmethod
    def powerlaw(cls, n, max_len, seed, layout=LAYOUT_CSR, b=None):
        return cls(lib().cgh_matrix_powerlaw(n, max_len, seed, layout, b.ptr if b is not None else None))

    min_row = property(lambda self: lib().cgh_matrix_min_row(self.h))

    dict_offsets = property(lambda self: lib().cgh_matrix_dict_offsets(self.h))
    dict_mode = property(lambda self: lib().cgh_matrix_dict_mode(self.h))
    stage_doubles = property(lambda self: lib().cgh_matrix_stage_doubles(self.h))

    def set_exact_rows(self, exact=True):
        check(lib().cgh_matrix_set_exact_rows(self.h, int(exact)), "set_exact_rows")
    n = property(lambda self: lib().cgh_matrix_rows(self.h))
    K = property(lambda self: lib().cgh_matrix_width(self.h))
    nnz_offdiag = property(lambda self: lib().cgh_matrix_nnz_offdiag(self.h))
    layout = property(lambda self: lib().cgh_matrix_layout(self.h))
    stored_bytes = property(lambda self: lib().cgh_matrix_stored_bytes(self.h))

    def download_ell(self):
        ev = np.empty(self.n * self.K)
        ec = np.empty(self.n * self.K, dtype=np.int32)
        check(lib().cgh_matrix_download_ell(self.h, _D(ev), _I(ec)), "download_ell")
        return ev, ec

    def download_csr(self):
        """CSR layout: (row_ptr int64 [n+1], cols int32, vals) of the full rows as stored"""
        rp = np.empty(self.n + 1, dtype=np.int64)
        cols = np.empty(self.nnz_offdiag, dtype=np.int32)
        vals = np.empty(self.nnz_offdiag)
        check(lib().cgh_matrix_download_csr(self.h, rp.ctypes.data_as(C.POINTER(C.c_int64)), _I(cols), _D(vals)), "download_csr")
        return rp, cols, vals

    def download_diag(self):
        d = np.empty(self.n)
        check(lib().cgh_matrix_download_diag(self.h, _D(d)), "download_diag")
        return d

    def spmv(self, x, y):
        check(lib().cgh_spmv(self.h, x.ptr, y.ptr), "cgh_spmv")

    def free(self):
        if self.h:
            lib().cgh_matrix_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def cg_solve(mat, b, x, max_iter=1000, tol=1e-6, strict=0, check_every=0, fixed_iters=0, time_kernels=0,
             history=True, precond=0):
    """-> (CgReport, history[iterations+1, 4] or None)"""
    opts = CgOpts(max_iter, tol, strict, check_every, fixed_iters, time_kernels, precond)
    rep = CgReport()
    cap = (fixed_iters + 2) if fixed_iters > 0 else max_iter + 1
    hist = np.full((cap, 4), np.nan) if history else None
    check(lib().cgh_cg_solve(mat.h, b.ptr, x.ptr, C.byref(opts), C.byref(rep),
                             _D(hist) if history else None), "cgh_cg_solve")
    return rep, (hist[:rep.iterations + 1] if history else None)


class Solver:
    """Resumable device CG (cgh_solver_*): queue iterations in batches, poll when you choose."""

    def __init__(self, mat, capacity):
        self.mat = mat
        self.h = lib().cgh_solver_create(mat.h, capacity)
        if not self.h:
            raise RuntimeError("cgh_solver_create failed: " + lib().cgh_last_error().decode())

    def start(self, b, x, tol=1e-6, strict=0):
        self._keep = (b, x)
        check(lib().cgh_solver_start(self.h, b.ptr, x.ptr, tol, strict), "cgh_solver_start")

    def iterate(self, count, time_kernels=0):
        check(lib().cgh_solver_iterate(self.h, count, time_kernels), "cgh_solver_iterate")

    def poll(self):
        d, q = C.c_int(), C.c_int()
        check(lib().cgh_solver_poll(self.h, C.byref(d), C.byref(q)), "cgh_solver_poll")
        return d.value, q.value

    def history(self, first, count):
        out = np.empty((count, 4))
        check(lib().cgh_solver_history(self.h, first, count, _D(out)), "cgh_solver_history")
        return out

    def spmv_ms(self):
        v = C.c_double()
        check(lib().cgh_solver_spmv_ms(self.h, C.byref(v)), "cgh_solver_spmv_ms")
        return v.value

    def free(self):
        if self.h:
            lib().cgh_solver_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Partition:
    """One rank's share of a row-block partitioned matrix (host-side plan; needs no GPU)."""

    def __init__(self, n, diag, upper, rows, cols, nranks, rank, row_starts=None):
        rs = None
        if row_starts is not None:
            self._rs = np.ascontiguousarray(row_starts, dtype=np.int64)
            rs = self._rs.ctypes.data_as(C.POINTER(C.c_int64))
        self.h = lib().cgh_partition_from_coo_host(n, rows.size, _D(diag), _D(upper), _I(rows), _I(cols), nranks, rank, rs)
        if not self.h:
            raise RuntimeError("cgh_partition_from_coo_host failed (malformed input)")
        L = lib()
        self.nranks, self.rank = nranks, rank
        self.n_loc = L.cgh_partition_rows(self.h)
        self.lo = L.cgh_partition_first_row(self.h)
        self.n_halo = L.cgh_partition_halo(self.h)
        self.K = L.cgh_partition_width(self.h)
        a, b = C.c_int64(), C.c_int64()
        L.cgh_partition_interior(self.h, C.byref(a), C.byref(b))
        self.interior = (a.value, b.value)
        self.diag = np.empty(self.n_loc)
        self.vals = np.empty(self.n_loc * self.K)
        self.cols = np.empty(self.n_loc * self.K, dtype=np.int32)
        self.halo_global = np.empty(self.n_halo, dtype=np.int64)
        check(L.cgh_partition_copy_local(self.h, _D(self.diag), _D(self.vals), _I(self.cols),
                                         self.halo_global.ctypes.data_as(C.POINTER(C.c_int64))), "partition_copy_local")
        self.neighbours = []
        for k in range(L.cgh_partition_neighbours(self.h)):
            r, sc, ro, rc = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64()
            check(L.cgh_partition_neighbour(self.h, k, C.byref(r), C.byref(sc), C.byref(ro), C.byref(rc)), "partition_neighbour")
            send = np.empty(sc.value, dtype=np.int32)
            check(L.cgh_partition_copy_send_list(self.h, k, _I(send)), "partition_send_list")
            self.neighbours.append(dict(rank=r.value, send=send, recv_off=ro.value, recv_cnt=rc.value))

    def free(self):
        if self.h:
            lib().cgh_partition_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DistRank:
    """Device side of one rank (cgh_dist_*)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("dist construction failed: " + lib().cgh_last_error().decode())
        self.h = handle

    @classmethod
    def from_partition(cls, part):
        return cls(lib().cgh_dist_from_partition(part.h))

    @classmethod
    def poisson(cls, nx, ny, nz, points, seed, nranks, rank, b=None):
        return cls(lib().cgh_dist_poisson(nx, ny, nz, points, seed, nranks, rank, b.ptr if b is not None else None))

    n_loc = property(lambda self: lib().cgh_dist_rows(self.h))
    nnz_offdiag = property(lambda self: lib().cgh_dist_nnz_offdiag(self.h))
    n_halo = property(lambda self: lib().cgh_dist_halo(self.h))
    K = property(lambda self: lib().cgh_dist_width(self.h))

    def download_local(self):
        d = np.empty(self.n_loc); ev = np.empty(self.n_loc * self.K); ec = np.empty(self.n_loc * self.K, dtype=np.int32)
        check(lib().cgh_dist_download_local(self.h, _D(d), _D(ev), _I(ec)), "dist_download_local")
        return d, ev, ec

    def vector_to_host(self, which):
        """r (0), p (1) or A p (2) of the solve in flight: the owned rows, copied to the host"""
        out = np.empty(self.n_loc)
        ptr = lib().cgh_dist_vector(self.h, which)
        if not ptr:
            raise RuntimeError("cgh_dist_vector: no solve in flight")
        check(lib().cgh_d2h(out.ctypes.data, ptr, out.nbytes), "d2h")
        return out

    def loopback(self):
        """one-GPU timing harness: this inner slab becomes its own two neighbours (cgh_dist_loopback)"""
        check(lib().cgh_dist_loopback(self.h), "cgh_dist_loopback")

    def start(self, b, x, tol, capacity, strict=0):
        self._keep = (b, x)
        check(lib().cgh_dist_set_strict(self.h, int(strict)), "cgh_dist_set_strict")
        check(lib().cgh_dist_start(self.h, b.ptr, x.ptr, tol, capacity), "cgh_dist_start")

    def iterate(self, count, time_kernels=0):
        check(lib().cgh_dist_iterate(self.h, count, time_kernels), "cgh_dist_iterate")

    def poll(self):
        d, q = C.c_int(), C.c_int()
        check(lib().cgh_dist_poll(self.h, C.byref(d), C.byref(q)), "cgh_dist_poll")
        return d.value, q.value

    def history(self, first, count):
        out = np.empty((count, 4))
        check(lib().cgh_dist_history(self.h, first, count, _D(out)), "cgh_dist_history")
        return out

    def spmv_ms(self):
        v = C.c_double()
        check(lib().cgh_dist_spmv_ms(self.h, C.byref(v)), "cgh_dist_spmv_ms")
        return v.value

    def free(self):
        if self.h:
            lib().cgh_dist_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def group_start(ranks, bs, xs, tol, capacity):
    G = len(ranks)
    arr = (C.c_void_p * G)(*[r.h for r in ranks])
    ba = (C.c_void_p * G)(*[b.ptr for b in bs])
    xa = (C.c_void_p * G)(*[x.ptr for x in xs])
    check(lib().cgh_dist_group_start(arr, G, ba, xa, tol, capacity), "cgh_dist_group_start")


def group_iterate(ranks, count):
    G = len(ranks)
    arr = (C.c_void_p * G)(*[r.h for r in ranks])
    check(lib().cgh_dist_group_iterate(arr, G, count), "cgh_dist_group_iterate")


def comm_unique_id():
    buf = C.create_string_buffer(256)
    check(lib().cgh_comm_unique_id(buf), "cgh_comm_unique_id")
    return buf.raw


def comm_allreduce(values, op="sum"):
    """all-reduce a short list of host doubles over the ranks (RCCL); also a barrier"""
    a = np.ascontiguousarray(values, dtype=np.float64).copy()
    check(lib().cgh_comm_allreduce_f64(_D(a), a.size, {"sum": 0, "max": 1, "min": 2}[op]), "cgh_comm_allreduce_f64")
    return a


def comm_init(nranks, rank, id_bytes):
    assert len(id_bytes) == 256
    check(lib().cgh_comm_init(nranks, rank, id_bytes), "cgh_comm_init")
