/*
 * cg_hip.h -- device-resident C ABI of libcg_hip (MI355X / gfx950).
 *
 * The reference (Gabrocecco/conjugate_gradient) has no handles: its API is the
 * plain functions of include/cg.h, csr.h, ell.h, vec.h, coo.h, all taking host
 * pointers (reference include/cg.h:4-56, ell.h:75-158, csr.h:26-53,
 * vec.h:26-46, coo.h:23-30).  This repo's include/{cg,csr,ell,vec,coo}.h keep
 * those signatures; each such entry point is a thin wrapper
 *     upload -> cgh_* below -> download
 * so the functions here are what a caller uses to keep the matrix and the
 * vectors resident in HBM across calls (benchmarks, long solves, multi-GPU).
 *
 * Conventions: plain C, no structs cross the boundary except the two POD
 * option/report structs below; sizes are 64-bit; every function returning int
 * returns 0 on success and a negative CGH_E* code on failure, with a message
 * available from cgh_last_error().  Pointers named d_* are device pointers
 * obtained from cgh_alloc().  The library is not thread-safe; one host thread
 * drives one device (one process per GPU for multi-GPU).
 */
#ifndef CG_HIP_H
#define CG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CGH_OK = 0,
    CGH_EHIP = -1,       /* a HIP runtime call failed */
    CGH_EINVAL = -2,     /* bad argument (shape, alignment, NULL) */
    CGH_ENOMEM = -3,     /* device or host allocation failed */
    CGH_ENODEV = -4,     /* no usable GPU / library not initialised */
    CGH_ECOMM = -5,      /* RCCL failure */
    CGH_EINPUT = -6      /* malformed matrix (non-upper entry, row overflow): partial output kept, as the reference does */
};

/* ---- runtime ---------------------------------------------------------- */
int         cgh_init(int device);          /* select device, create streams; idempotent for the same device */
void        cgh_shutdown(void);
int         cgh_device_count(void);
const char *cgh_last_error(void);
int         cgh_sync(void);                /* drain every stream the library owns */

void *cgh_alloc(size_t bytes);             /* hipMalloc; NULL on failure */
void  cgh_free(void *d_ptr);
int   cgh_h2d(void *d_dst, const void *h_src, size_t bytes);
int   cgh_d2h(void *h_dst, const void *d_src, size_t bytes);
int   cgh_d2d(void *d_dst, const void *d_src, size_t bytes);
int   cgh_memset(void *d_dst, int byte, size_t bytes);

/* ---- matrices ----------------------------------------------------------
 * A cgh_matrix is diag[n] plus the off-diagonals in one device layout.  In
 * every layout a row's stored entries are kept in the ORDER in which the
 * reference's SpMV for the source format accumulates them, so y = A x is
 * bit-identical to the reference's (see DESIGN.md "Accumulation order").
 */
typedef struct cgh_matrix cgh_matrix;

enum {
    CGH_LAYOUT_ELL = 0,   /* slot-major full ELL: entry (row i, slot s) at s*n+i; pad = value 0.0, col -1 (reference ell.c:791-858 layout) */
    CGH_LAYOUT_CSR = 1,   /* full (both triangles) CSR, int64 row_ptr, rows in reference accumulation order; vector/adaptive kernels */
    CGH_LAYOUT_SELL = 2,  /* sliced ELL: slices of 64 rows, each padded to its own width, slot-major inside the slice */
    CGH_LAYOUT_CSRB = 3,  /* column-blocked CSR (throughput layout for skewed rows / scattered columns, BASELINE config 5):
                             rows dealt in 64-row segments to wave-blocks of equal weight, a block's entries sorted by
                             column slab so that every wave on the chip gathers from the same ~1 MiB window of x at the
                             same time; per-row sums in a fixed order (deterministic) that is NOT the reference's order */
    CGH_LAYOUT_SELLS = 4, /* sigma-sorted sliced ELL (SELL-C-sigma, C = 64): rows sorted by length inside windows of
                             sigma rows before slicing, the permutation folded into y; per-row order = the reference's */
    CGH_LAYOUT_PB = 5     /* two-phase SpMV (propagation blocking; throughput layout for scattered columns, BASELINE config 5):
                             entries sorted by (row block, column block) with 2-byte local indices; phase 1 multiplies by x
                             staged in LDS and streams the products, phase 2 adds them into row sums held in LDS -- no random
                             access leaves the CU; a row's sum is added in the order of the product stream (the same bits on every
                             run; not the reference's order: agrees with it to rounding, not bit for bit) */
};

/* Host slot-major full ELL (what coo_to_ell_symmetric_full_colmajor produced) -> device. */
cgh_matrix *cgh_matrix_from_ell_host(int64_t n, int K, const double *diag, const double *ell_vals, const int *ell_cols);
/* Host upper-triangular COO/LDU triple (row-sorted or not) -> device conversion kernels -> `layout`.
 * Slot / entry order inside a row is the reference's: rank of the placement among all earlier
 * placements into that row, entry k placing (i,j) first and (j,i) second (reference ell.c:815-855). */
cgh_matrix *cgh_matrix_from_coo_host(int64_t n, int64_t nnz_upper, const double *diag, const double *upper,
                                     const int *rows, const int *cols, int layout);
/* Same, inputs already on the device. */
cgh_matrix *cgh_matrix_from_coo_device(int64_t n, int64_t nnz_upper, const double *d_diag, const double *d_upper,
                                       const int *d_rows, const int *d_cols, int layout);
/* Seeded synthetic SPD matrices generated on the device (DESIGN.md "Generators"); d_b (may be NULL) receives the rhs. */
cgh_matrix *cgh_matrix_poisson(int nx, int ny, int nz, int points /*7|27*/, uint64_t seed, int layout, double *d_b);
/* Power-law (Pareto, exponent 2) row lengths 1..max_len (<= 2048), strictly diagonally dominant SPD (BASELINE config 5). */
cgh_matrix *cgh_matrix_powerlaw(int64_t n, int max_len, uint64_t seed, int layout, double *d_b);
void        cgh_matrix_free(cgh_matrix *m);

int64_t cgh_matrix_rows(const cgh_matrix *m);
int     cgh_matrix_width(const cgh_matrix *m);        /* K: max off-diagonals per row */
int64_t cgh_matrix_nnz_offdiag(const cgh_matrix *m);  /* stored non-pad off-diagonals (= 2U) */
int     cgh_matrix_layout(const cgh_matrix *m);
int64_t cgh_matrix_stored_bytes(const cgh_matrix *m); /* HBM bytes of the matrix arrays incl. padding */
int     cgh_matrix_min_row(const cgh_matrix *m);      /* shortest row of a CSR-layout matrix (-1 otherwise) */
/* CSR layout: 1 = every row's products are added strictly left to right (the reference's order) however long the row;
 * 0 (default) = rows of more than 64 entries use a wave/workgroup tree.  The include/cg.h, csr.h, coo.h entry points set 1. */
int     cgh_matrix_set_exact_rows(cgh_matrix *m, int exact);
/* Tuning of the CGH_LAYOUT_CSRB / CGH_LAYOUT_SELLS matrices constructed AFTER the call (0 keeps the current value unless
 * stated): rows per wave-block (512 | 1024 | 2048); log2 of the columns per slab (4..22); workgroups per CU (0 = what
 * LDS admits); pacing of the sweep over x: slabs a wave may run ahead of its group (0 = no pacing, < 0 keeps) and the
 * quorum divisor d (a gate opens once all but 1/d of the group's workgroups have passed; 2..255); row sums by LDS
 * atomics (1) or by segmented scan + read-add-write (0, the default: order fixed by construction; < 0 keeps).
 * Sorting window (rows) of the sigma-sorted sliced ELL. */
int     cgh_config_csrb(int rows_per_wave_block, int slab_bits, int wgs_per_cu, int window, int quorum_div, int lds_atomics);
int     cgh_config_sell_sigma(int sigma_rows);
/* Tuning of the CGH_LAYOUT_PB matrices constructed after the call (0 keeps): log2 of the most rows per block (13 | 14), entries
 * per block the partition aims at, workgroups sharing one column block in phase 1 (1..16); ordered: 1 = phase 2 adds the
 * products in the order of the stream (the same bits on every run; the default), 0 = as the LDS atomics retire (about 2 %
 * faster, results differ from run to run in the last bits), < 0 keeps. */
int     cgh_config_pb(int span_bits, int64_t target_entries, int split, int ordered);
/* Throughput-mode (strict = 0) solves of ELL matrices of at most 131 072 rows and at most 8 slots run their iterations as
 * ONE resident launch per batch (csrc/cg_persistent.h: workgroups keep their rows in registers and exchange p and the
 * partial sums as tagged words; same bits as the three launches per iteration).  0 switches that off (also CGH_PERSISTENT=0),
 * 1 on (the default).  The launch needs its <= 256 workgroups resident together: on a GPU shared with other work a wait
 * gives up after 2 s and cgh_solver_poll reports it. */
int     cgh_config_persistent(int on);
struct cgh_solver;
int     cgh_solver_is_resident(const struct cgh_solver *sv);  /* after cgh_solver_start: 1 if its iterations take that path */
long long cgh_persistent_launches(void);                      /* tests: resident launches queued by this process so far */
/* Tools (tools/quick_persistent.py): the resident launch leaves, per workgroup, the ticks (100 MHz) its first lane spent in
 * five sections of the iteration, summed over the launch, in d_trace[8 w + k] (device memory, 256 x 8 values); NULL = off. */
int     cgh_persistent_set_trace(long long *d_trace);
/* Host-side partition of the two-phase layout (no GPU needed; exported for tests): cuts G groups of 256 rows (cum[g] =
 * entries before group g, cum[G] = all) into consecutive blocks of at most max_groups groups and, where that leaves a
 * choice, at most `target` entries.  bnd_g[0..B] receives the boundaries in groups.  Returns B. */
int64_t cgh_pb_partition_host(int64_t G, const int64_t *cum, int max_groups, int64_t target, int64_t *bnd_g);
/* The CGH_* environment knobs (DESIGN.md lists them) are read ONCE, by cgh_init, never in a launch path; a host that
 * changes the environment afterwards (tests do) calls this to have them read again. */
int     cgh_config_reload(void);
/* Values set through cgh_config_csrb / cgh_config_sell_sigma / cgh_config_pb win over defaults and environment and stay in
 * force across cgh_init and cgh_config_reload; this forgets them (defaults + environment again).  Tests call it between cases. */
int     cgh_config_reset(void);
/* Host-side planner of the column-blocked CSR (no GPU needed; exported for tests): deals nseg segments of the given
 * weights (entries) to ceil(nseg/spb) blocks of at most spb segments each, heaviest first to the lightest block with a
 * free slot.  seg_block[s], seg_slot[s] receive the placement.  Returns the number of blocks. */
int64_t cgh_csrb_plan_host(int64_t nseg, const int *weights, int spb, int *seg_block, int *seg_slot);
/* ELL layout: the SpMV reads 1-byte dictionary codes instead of 4-byte column ids when the matrix has at most 255
 * distinct (col - row) offsets (banded / stencil matrices; built on the first SpMV; CGH_ELL_DICT=0 disables).  Returns the
 * number of distinct offsets in use, 0 if the matrix does not qualify, -1 before the first SpMV / other layouts. */
int     cgh_matrix_dict_offsets(const cgh_matrix *m);
/* How the coded columns are laid out: 1 = one byte per entry, slot-major (rows of up to 8 slots); 2 = the codes of 4 slots
 * of a row in one 32-bit word, slot-major; 3 = the same words AND a copy of the values "tile-blocked": the K slots of every
 * 512-row tile lie together, so a workgroup reads one contiguous run per tile instead of K runs n*8 bytes apart (wide rows,
 * K > 8; costs 8 more bytes of HBM per stored slot; CGH_ELL_BLOCKED=0 keeps 2).  0 = int32 columns / not built / other layouts. */
int     cgh_matrix_dict_mode(const cgh_matrix *m);
/* Wide banded rows (dict_mode 3, any width above 8 slots): the SpMV stages the windows of x a 512-row tile reads (row + d for the matrix's distinct
 * offsets d, overlapping runs merged) in LDS with coalesced loads and gathers from there instead of from L1 / L2.  Returns
 * the doubles of LDS per workgroup in use for this matrix, 0 if its launches gather from the caches (CGH_ELL_STAGE=0 forces
 * that; the windows of a matrix must fit 44 KiB -- 50 KiB where two 512-row tiles are staged together, CGH_ELL_STAGE_TILES --
 * and all its columns lie below n). */
int     cgh_matrix_stage_doubles(const cgh_matrix *m);
/* Download as slot-major full ELL of width K=cgh_matrix_width (any layout); buffers hold n*K entries. */
int     cgh_matrix_download_ell(const cgh_matrix *m, double *ell_vals, int *ell_cols);
int     cgh_matrix_download_diag(const cgh_matrix *m, double *diag);
/* CGH_LAYOUT_CSR only: the full rows as stored (row_ptr[n+1], cols / vals [cgh_matrix_nnz_offdiag]). */
int     cgh_matrix_download_csr(const cgh_matrix *m, int64_t *row_ptr, int *cols, double *vals);

/* ---- kernels ------------------------------------------------------------ */
int cgh_spmv(const cgh_matrix *m, const double *d_x, double *d_y);           /* y = A x */
/* Fused BLAS-1 (device scalars are produced into d_out[0..]) */
int cgh_dot(const double *d_a, const double *d_b, int64_t n, int strict, double *h_out);
int cgh_l1norm(const double *d_a, int64_t n, int strict, double *h_out);
int cgh_axpy(const double *d_a, const double *d_b, double alpha, double *d_out, int64_t n); /* out = a + alpha*b */

/* ---- CG ------------------------------------------------------------------
 * The loop of reference src/cg.c:414-508 with the matrix and all vectors
 * resident: x=0; r=b; p=r; for iter=1..max_iter-1 { Ap=A p; alpha=rr/(p.Ap);
 * x+=alpha p; r-=alpha Ap; rr'=r.r; if l1(r)/2<tol return iter+1;
 * beta=rr'/rr; p=r+beta p }.  Returns the reference's value in report->ret.
 */
typedef struct cgh_cg_opts {
    int    max_iter;      /* reference passes 1000 */
    double tol;           /* reference passes 1e-6; compared with l1norm(r)/2 */
    int    strict;        /* 1: dots/norms summed strictly left to right (bit-identical trajectory to the reference);
                             0: wave-shuffle + LDS tree reductions (throughput) */
    int    check_every;   /* iterations queued between host looks at the device's converged flag (>=1; 0 -> default 32).
                             The device freezes x at the converging iteration, so the result does not depend on this. */
    int    fixed_iters;   /* >0: run exactly this many iterations, never test convergence (benchmark mode) */
    int    time_kernels;  /* N > 0: bracket the SpMV kernel of every N-th iteration with HIP events (benchmark mode; two event
                           * records cost ~6 us, which shows in iterations of a few tens of us) */
    int    precond;       /* CGH_PRECOND_NONE (0, the reference's loop) or CGH_PRECOND_JACOBI: z = r/diag, p = z + beta p,
                             alpha = r.z/(p.Ap); no reference counterpart, strict must be 0; history[.][1] = r.z */
} cgh_cg_opts;
enum { CGH_PRECOND_NONE = 0, CGH_PRECOND_JACOBI = 1 };

typedef struct cgh_cg_report {
    int    ret;           /* iter+1 on convergence, -1 otherwise (reference convention) */
    int    iterations;    /* iterations executed */
    double initial_res;   /* l1norm(r0)/2 */
    double final_res;     /* l1norm(r)/2 after the last executed iteration */
    double r0[5];         /* r[0..4] after r0 = b - A*0 (the reference prints them, cg.c:444-449) */
    double solve_ms;      /* wall time of the iteration loop (HIP events) */
    double spmv_ms_avg;   /* average SpMV kernel time when time_kernels=1, else 0 */
} cgh_cg_report;

/* h_history (may be NULL) receives 4 doubles per iteration k=0..iterations:
 * { l1norm(r)/2, r.r, alpha, beta } (NaN where the reference computes none). Needs 4*max_iter doubles
 * of HOST memory if given; the device keeps a bounded window of it whatever max_iter is. */
int cgh_cg_solve(const cgh_matrix *m, const double *d_b, double *d_x,
                 const cgh_cg_opts *opts, cgh_cg_report *report, double *h_history);

/* The one entry point that ALLOCATES NOTHING.  It has no counterpart in the reference; it is for callers that cannot call
 * hipMalloc: code under HIP graph capture, tight serving loops that solve one small system after another, harnesses that
 * forbid allocation.  It is cgh_cg_solve (same kernels, same launch shapes, same bits) with the solver's workspace
 * (r, p, A p, the per-workgroup partials, the converged flag) in a __device__ array built into the library: 25.2 MB
 * that every GPU the library runs on holds until the process ends (28.3 MB with the workspace of the order-exact sums that
 * strict = 1 uses from 4096 rows on and the images of the resident launch that strict = 0 takes up to 131 072 rows).  Limit: at most 2^20 = 1048576 rows, and a matrix with rows -- anything else is
 * refused with CGH_EINVAL and a message.  Asynchronous: all max_iter-1 iterations are queued on the library's stream
 * (the device freezes x, r, p at the converging iteration) and the call returns; after cgh_sync() d_x holds the
 * solution, d_history[4*k..] = { l1norm(r)/2, r.r, alpha, beta } for k = 0 .. converging iteration (4*max_iter doubles,
 * later entries untouched), d_status[0] = converging iteration (0 = not converged; return value of the reference =
 * d_status[0] + 1 or -1), d_status[1..3] reserved.  All calls on one GPU share the one workspace: calls that overlap on
 * a GPU (two host threads) race. */
int cgh_cg_solve_noalloc(const cgh_matrix *m, const double *d_b, double *d_x, int max_iter, double tol, int strict,
                         double *d_history, int *d_status);

/* Helpers for such callers and for the test of the entry point above: number of device allocations the library has made
 * so far (cgh_alloc -> hipMalloc); hipMemGetInfo of the current device; capture of everything the library queues on its
 * compute stream between the two calls into a HIP graph (global capture mode: hipMalloc, hipFree and synchronisations
 * inside it fail), which cgh_capture_end_launch instantiates, replays `replays` times, waits for and releases. */
long long cgh_alloc_calls(void);
int       cgh_mem_info(size_t *free_bytes, size_t *total_bytes);
int       cgh_capture_begin(void);
int       cgh_capture_end_launch(int replays);

/* Resumable form of the same loop (what cgh_cg_solve is built from): the workspace lives in the
 * solver, iterations are queued asynchronously in batches, the host polls when it chooses. */
typedef struct cgh_solver cgh_solver;
cgh_solver   *cgh_solver_create(const cgh_matrix *m, int iteration_capacity);
void          cgh_solver_free(cgh_solver *s);
int           cgh_solver_set_precond(cgh_solver *s, int kind);   /* before cgh_solver_start; default CGH_PRECOND_NONE */
int           cgh_solver_start(cgh_solver *s, const double *d_b, double *d_x, double tol, int strict);
int           cgh_solver_iterate(cgh_solver *s, int count, int time_kernels);
int           cgh_solver_poll(cgh_solver *s, int *done_iter, int *queued);
int           cgh_solver_history(cgh_solver *s, int first, int count, double *h_out);
/* A solve longer than iteration_capacity continues in a fresh history window: call between polls, while not converged.
 * The last entry becomes entry 0, iteration numbers (cgh_solver_poll, cgh_solver_history) restart at 0 and
 * cgh_solver_base() returns how many iterations lie before the window.  cgh_cg_solve uses windows of 4096 iterations,
 * so its device memory does not grow with max_iter (the reference's loop is O(1) in max_iter, src/cg.c:414-508). */
int           cgh_solver_rebase(cgh_solver *s);
int           cgh_solver_base(const cgh_solver *s);
int           cgh_solver_spmv_ms(cgh_solver *s, double *ms_avg);
const double *cgh_solver_residual(const cgh_solver *s);   /* device pointer to r */

/* ---- multi-GPU: row-block partition + halo exchange (one process per GPU, RCCL over xGMI) ---------------
 * No reference counterpart (the reference is serial).  Rank g owns a contiguous row block; per iteration the
 * boundary entries of p go to the neighbouring ranks (ncclSend/ncclRecv on a communication stream, overlapped
 * with the SpMV of the rows that need no halo) and the two scalar reductions are ncclAllReduce.
 */
typedef struct cgh_partition cgh_partition;   /* host-side description of one rank's share (no GPU needed) */
typedef struct cgh_dist cgh_dist;             /* device-side distributed solver of one rank */

/* Rank `rank`'s share of the n-row matrix given as a global upper-triangular COO on the host.
 * row_starts: nranks+1 block starts, or NULL for an even split.  NULL on malformed input. */
cgh_partition *cgh_partition_from_coo_host(int64_t n, int64_t nnz_upper, const double *diag, const double *upper,
                                           const int *rows, const int *cols, int nranks, int rank,
                                           const int64_t *row_starts);
void    cgh_partition_free(cgh_partition *p);
int64_t cgh_partition_rows(const cgh_partition *p);        /* owned rows */
int64_t cgh_partition_first_row(const cgh_partition *p);   /* global index of the first owned row */
int64_t cgh_partition_halo(const cgh_partition *p);        /* halo entries (distinct off-block columns) */
int     cgh_partition_width(const cgh_partition *p);       /* local ELL width */
int     cgh_partition_neighbours(const cgh_partition *p);
void    cgh_partition_interior(const cgh_partition *p, int64_t *begin, int64_t *end);  /* rows needing no halo */
int     cgh_partition_neighbour(const cgh_partition *p, int k, int *rank, int64_t *send_count, int64_t *recv_off,
                                int64_t *recv_count);
int     cgh_partition_copy_send_list(const cgh_partition *p, int k, int *out_local_rows);
/* local slot-major ELL (rows x width, local column ids: owned < rows <= halo), diag, global column of each halo slot */
int     cgh_partition_copy_local(const cgh_partition *p, double *diag, double *ell_vals, int *ell_cols,
                                 int64_t *halo_global);

/* RCCL bootstrap: rank 0 calls cgh_comm_unique_id, the 256 bytes (two ncclUniqueIds: one communicator for the
 * halo exchange, one for the all-reduces) travel to the other ranks by any means (bench_dist.py: a file in /tmp of
 * the one node), then every rank calls cgh_comm_init. */
int cgh_comm_preload(void);   /* dlopen RCCL now (call before another copy of librccl gets loaded, e.g. before importing torch) */
int cgh_comm_unique_id(char out_256_bytes[256]);
int cgh_comm_init(int nranks, int rank, const char id_256_bytes[256]);
void cgh_comm_destroy(void);
/* Peer-to-peer scalar all-reduce (p2p.hip): cgh_comm_init sets it up by itself over the RCCL communicator (handles
 * all-gathered, self-test, all ranks agree; CGH_P2P=0 disables) and cgh_dist_iterate then uses it for the two
 * reductions of an iteration instead of ncclAllReduce.  The pieces are exported for hosts that bootstrap differently
 * and for tests: every rank takes its mailbox handle (64 bytes), the handles travel by any means, every rank connects;
 * a barrier must separate the connects from the first all-reduce. */
int  cgh_p2p_local_handle(char out_64_bytes[64]);
int  cgh_p2p_connect(int nranks, int rank, const char *handles_nranks_x_64_bytes);
void cgh_p2p_disconnect(void);
int  cgh_p2p_connected(void);
int  cgh_comm_init_local(int nranks, int rank);   /* ranks without RCCL (mailboxes connected by the caller); several may share one GPU */
int  cgh_comm_set_p2p(int on);                    /* use / bypass it for the following solves (all ranks alike); returns the state */
int  cgh_p2p_error(void);                          /* 1 after a timed-out wait (results are NaN from then on) */
int  cgh_p2p_allreduce_f64(double *h_vals, int count);   /* 1 or 2 host doubles, summed in rank order */
/* launcher helper: all-reduce up to 64 host doubles over the ranks (op 0 sum, 1 max, 2 min); one value = a barrier */
int cgh_comm_allreduce_f64(double *h_vals, int count, int op);
/* launcher diagnostic (collective): `reps` all-reduces of two doubles back to back through the peer-to-peer mailboxes (1) or
 * ncclAllReduce (0), microseconds per exchange (HIP events around the loop) */
int cgh_comm_bench_allreduce(int reps, int use_p2p, double *us_per_exchange);

/* Device side of a rank: from a host partition, or generated on the device for the z-slab partition of the
 * seeded Poisson stencils (rank g owns planes [nz*g/G, nz*(g+1)/G)); d_b (may be NULL) gets the local rhs. */
cgh_dist *cgh_dist_from_partition(const cgh_partition *p);
cgh_dist *cgh_dist_poisson(int nx, int ny, int nz, int points, uint64_t seed, int nranks, int rank, double *d_b);
void      cgh_dist_free(cgh_dist *d);
int64_t   cgh_dist_rows(const cgh_dist *d);
int64_t   cgh_dist_nnz_offdiag(const cgh_dist *d);
/* The resumable CG loop on the rank's rows (same contract as cgh_solver_*; collective: every rank calls it
 * with the same counts).  transport: RCCL communicator from cgh_comm_init. */
int cgh_dist_start(cgh_dist *d, const double *d_b_local, double *d_x_local, double tol, int iteration_capacity);
/* Bit-identical mode of the distributed loop for the solves started afterwards (every rank makes the same choice): the two
 * dots and the L1 norm become the reference's left-to-right sums over the GLOBAL vector (src/vec.c:67-85 there) --
 * order-exact sums chained through the ranks in rank order -- so trajectory, x and the return value equal the reference's
 * on the undistributed system bit for bit, whatever the number of ranks.  Serial in the number of ranks: a verification
 * mode.  No reference counterpart (the reference is single-process). */
int cgh_dist_set_strict(cgh_dist *d, int strict);
int cgh_dist_iterate(cgh_dist *d, int count, int time_kernels);
int cgh_dist_poll(cgh_dist *d, int *done_iter, int *queued);
int cgh_dist_history(cgh_dist *d, int first, int count, double *h_out);
int cgh_dist_spmv_ms(cgh_dist *d, double *ms_avg);
/* Test harness: the same per-rank code paths with `nranks` ranks living in ONE process on ONE device, the halo
 * exchange and the reductions done by device copies instead of RCCL (the collective calls themselves excepted,
 * this runs every kernel and every index list of the multi-GPU path on a single GPU). */
int cgh_dist_group_start(cgh_dist **ranks, int nranks, const double *const *d_b, double *const *d_x, double tol, int iteration_capacity);
int cgh_dist_group_iterate(cgh_dist **ranks, int nranks, int count);
/* test/debug access to a rank's local block */
int     cgh_dist_width(const cgh_dist *d);
int64_t cgh_dist_halo(const cgh_dist *d);
int     cgh_dist_download_local(const cgh_dist *d, double *diag, double *ell_vals, int *ell_cols);
const double *cgh_dist_vector(const cgh_dist *d, int which);   /* device pointer to r (0), p (1) or A p (2) of the solve in flight */
/* Peer-to-peer halo exchange (no library in the loop): each rank deposits its boundary entries of p straight into a
 * buffer of the neighbour (hipIpc) and raises a flag; see dist.hip.  With an RCCL communicator whose peer-to-peer path
 * is up, cgh_dist_from_partition / cgh_dist_poisson connect it by themselves (collectively: every rank must create
 * its cgh_dist at the same point); otherwise the host moves the 512-byte descriptors (rank order), then a barrier. */
int     cgh_dist_halo_descriptor(cgh_dist *d, char out_512_bytes[512]);
int     cgh_dist_halo_connect(cgh_dist *d, const char *descriptors_nranks_x_512_bytes);
int     cgh_dist_halo_is_p2p(const cgh_dist *d);
/* One-GPU timing harness: an inner slab (created as rank r of N under a ONE-rank communicator) becomes its own two
 * neighbours, i.e. the slab's operator closed periodically in z; per-iteration kernels, bytes and waits are those of an
 * inner rank of the N-GPU run, minus the xGMI hop.  No reference counterpart. */
int     cgh_dist_loopback(cgh_dist *d);

/* ---- timing helpers (benchmarks) ---------------------------------------- */
/* Launch `reps` back-to-back SpMVs and return the average kernel time in ms (HIP events on the launch stream). */
int cgh_time_spmv(const cgh_matrix *m, const double *d_x, double *d_y, int reps, double *ms_avg);
/* float4-style device copy of `bytes` bytes, `reps` times: the attainable-HBM yardstick (ms per copy). */
int cgh_time_copy(void *d_dst, const void *d_src, size_t bytes, int reps, double *ms_avg);
/* the same with the copy's shape chosen by the caller: variant = loads in flight per lane (1 | 2 | 4 | 8) + 16 (non-temporal
 * loads) + 32 (non-temporal stores) + 256 * workgroups per CU (0 = 64); cgh_time_copy is variant 4 + 16 + 32 */
int cgh_time_copy_variant(void *d_dst, const void *d_src, size_t bytes, int reps, int variant, double *ms_avg);
/* profile of the serial walks of the order-exact sums (xsum.hip: XS_PROF_ADD): 16 zeroed 64-bit device words, or NULL */
void cgh_xsum_set_prof(void *dev_16_words);
/* read-only stream of `bytes` from d_src (d_sink_8k: 8 KiB of device scratch): the ceiling for read-dominated kernels */
int cgh_time_read(const void *d_src, void *d_sink_8k, size_t bytes, int reps, double *ms_avg);
int     cgh_time_read_variant(const void *d_src, void *d_sink_8k, size_t bytes, int reps, int wgs_per_cu, double *ms_avg);  /* the same with 1..8 workgroups per CU */
/* `count` 8-byte words from d_src8 plus `count` 4-byte words from d_src4, one of each per lane and step (how the CSR-family
 * kernels stream a matrix): a stream of known size that calibrates the FETCH_SIZE counter for those access widths */
int cgh_time_read_8_4(const void *d_src8, const void *d_src4, void *d_sink_8k, size_t count, int reps, double *ms_avg);

#ifdef __cplusplus
}
#endif
#endif
