// cgh_internal.h -- shared internals of libcg_hip (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/cg_hip.h"

struct cgh_context {
    bool        ready = false;
    int         device = -1;
    int         num_cu = 256;
    hipStream_t stream = nullptr;       // compute stream: every kernel of the solver
    hipStream_t comm_stream = nullptr;  // halo exchange / collectives (multi-GPU)
    char        err[512] = {0};
};
extern cgh_context g_cgh;

void cgh_set_error(const char *fmt, ...);

#define CGH_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            cgh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return CGH_EHIP;                                                                \
        }                                                                                   \
    } while (0)

#define CGH_HIP_NULL(call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            cgh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return nullptr;                                                                 \
        }                                                                                   \
    } while (0)

#define CGH_REQUIRE_READY(ret)                                                              \
    do {                                                                                    \
        if (!g_cgh.ready) {                                                                 \
            cgh_set_error("libcg_hip: cgh_init() has not been called (or no GPU)");          \
            return ret;                                                                     \
        }                                                                                   \
    } while (0)

// ---------------------------------------------------------------- matrices
constexpr int CGH_STAGE_WINDOWS = 12;  // windows of x per staging round at most (spmv_ell_staged): e.g. the nine (y, z) lines of a 27-point stencil on a wide grid
struct cgh_matrix {
    int64_t n = 0;
    int     K = 0;               // max off-diagonals per row
    int     layout = CGH_LAYOUT_ELL;
    int64_t nnz_off = 0;         // stored non-pad off-diagonals
    double *diag = nullptr;      // [n]
    // ELL (slot-major): vals/cols [K*n]
    // CSR: vals/cols [nnz_off], row_ptr [n+1]
    // SELL: vals/cols [slice_ptr[nslices]], slice_ptr [nslices+1] (entry offsets), slice width = (ptr[s+1]-ptr[s])/64
    double  *vals = nullptr;
    int     *cols = nullptr;
    int64_t *row_ptr = nullptr;
    int64_t *slice_ptr = nullptr;
    int64_t  nslices = 0;
    int64_t  stored = 0;         // stored entries incl. padding
    // CSR-adaptive row bins (row ids sorted into length classes), built once at construction
    int     *bin_rows = nullptr; // [n] row ids grouped by bin
    int64_t  bin_start[5] = {0, 0, 0, 0, 0}; // short / sub-wave / wave / block rows
    // ELL only: dictionary-coded columns, built on the first SpMV (spmv_ell.hip).  cols8[s*n+i] = index of (col - i)
    // in off_tab (<= 255 distinct offsets: banded / stencil matrices), 255 = pad; 1 byte per entry instead of 4.
    mutable unsigned char *cols8 = nullptr;
    mutable int           *off_tab = nullptr;  // [256] sorted offsets, device
    mutable int            dict_state = 0;     // 0 not tried, 1 in use, 2 not applicable
    mutable int            n_offsets = 0;
    mutable int            dict_mode = 0;      // 1: cols8 = one byte per entry (slot-major); 2: 4 slots of a row per 32-bit word; 3: the same words and the values tile-blocked (vals_blk)
    // gathers served from LDS (spmv_ell_staged): windows of x per tile, relative to the tile's first row; 0 windows = not in use.
    // off_tab then holds 768 ints: the offsets, and per code the LDS index of lane 0's operand for its first / second row
    mutable int            stage_nwin = 0, stage_total = 0, stage_first[CGH_STAGE_WINDOWS] = {0}, stage_len[CGH_STAGE_WINDOWS] = {0},
                           stage_base[CGH_STAGE_WINDOWS] = {0};
    mutable int            stage_tiles = 1;        // 512-row tiles per staging round (2 for wide rows if the plan fits)
    mutable bool           stage_any_row = false;  // every column lies below n: any row range may be staged (else only interior rows)
    mutable double        *vals_blk = nullptr; // dict_mode 3 (wide rows): the values with the K slots of every 512-row tile together
    bool     owns = true;
    bool     exact_rows = false;  // CSR: add every row's products strictly left to right, however long the row is
    // CSRB (column-blocked CSR, spmv_csrb.hip): rows are dealt in 64-row segments to wave-blocks of rpw rows holding about
    // equal numbers of entries; a block's entries are sorted by (column slab of 2^cb_bits columns, local row) and stored as
    // vals[] + cols[] reinterpreted as packed words (local row << cb_bits | column inside the slab)
    int      cb_bits = 0, nslabs = 0, rpw = 0;
    int64_t  nblocks = 0;
    int     *blk_seg = nullptr;    // [nblocks * rpw/64] first row of each segment of the block, -1 = none
    int64_t *blk_base = nullptr;   // [nblocks + 1] entry offsets of the blocks
    int     *tile_off = nullptr;   // [nblocks * (nslabs + 1)] start of each slab's entries, relative to blk_base[b]
    unsigned *gate_prog = nullptr; // progress counters of the sweep's pacing (zeroed before every launch)
    int      gate_window = 0;      // pacing word: window (slabs a wave may run ahead; 0 = no pacing) | quorum divisor << 8 | LDS atomics << 16
    // SELLS (sigma-sorted sliced ELL): rows sorted by length inside windows of sigma rows before slicing;
    // perm[p] = row stored at position p (slice p/64, lane p%64), -1 beyond the last row
    int     *perm = nullptr;
    int      sigma = 0;
    // PB (two-phase SpMV, spmv_pb.hip): rows = columns cut into pb_blocks blocks (pb_bnd, at most pb_span rows each); vals[] and
    // the 2-byte local columns sorted by (column block, row block), the 2-byte local rows and the launch's product stream
    // pb_prod by (row block, column block) (pb_prod is a workspace of the matrix: launches on one matrix must not overlap)
    int             pb_blocks = 0, pb_span = 0, pb_split = 1;
    bool            pb_ordered = true;   // phase 2 adds the stream's pieces in stream order (same bits on every run)
    int            *pb_bnd = nullptr;   // [pb_blocks + 1] first row of each block
    int            *pb_pre = nullptr;   // [pb_blocks][pb_blocks + 1] per column block: flat position of each of its segments
    unsigned       *pb_seg0 = nullptr;  // [pb_blocks][pb_blocks] per column block: position of each of its segments in the product stream
    int64_t        *pb_bin = nullptr;   // [pb_blocks + 1] position of each row block in the product stream
    int64_t        *pb_cbase = nullptr; // [pb_blocks] start of each column block's run of values / local columns
    unsigned short *pb_lcol = nullptr, *pb_lrow = nullptr;  // [nnz_off]
    double         *pb_prod = nullptr;  // [nnz_off]
};

// What a caller may ask of ONE SpMV launch beyond y = A x (explicit arguments: nothing about a launch travels through
// process-wide state, so two solvers may interleave their launches).
struct cgh_spmv_opts {
    double *cell_sums = nullptr;         // strict mode: the kernel may leave x.y summed per 128-row cell here (xsum.hip's
    bool   *cell_sums_written = nullptr; // binade prediction); *cell_sums_written says whether the kernel it chose did
    int     reverse = 0;                 // 1: walk the rows from the end (serpentine sweeps)
    bool    no_alloc = false;            // the call must not allocate: an ELL matrix's column dictionary is not built now
    bool    interior_rows = false;       // the rows of this launch hold no column >= n (the interior of a multi-GPU block, whose
                                         // other rows read halo entries behind x's n owned ones): the LDS-staged kernel may serve them
};

// Knobs read from the environment ONCE, in cgh_init (never in a launch path); cgh_config_* setters override them.
struct cgh_config {
    int    ell_generic = 0;     // CGH_ELL_GENERIC: runtime-K ELL kernel for every width (tests)
    int    ell_dict = -1;       // CGH_ELL_DICT: 0 no dictionary-coded columns, 1 / 2 force the byte / packed layout, -1 automatic
    int    ell_variant = -1;    // CGH_ELL_VARIANT: cache-hint variant of the ELL SpMV, -1 = by working-set size
    int    ell_blocked = 1;     // CGH_ELL_BLOCKED: wide dictionary-coded rows (K > 8) read tile-blocked copies of values and codes (0: slot-major)
    int    ell_stage = 1;       // CGH_ELL_STAGE: 1 = wide dictionary-coded ELL rows (tile-blocked, K = 13 / 26) gather x from windows staged in LDS; 2 = narrow rows too; 0 = everything from L1 / L2
    int    ell_stage_tiles = 2; // CGH_ELL_STAGE_TILES: 512-row tiles per staging round of wide rows: 2 = two where that saves >= 30 % of the staged doubles per row and fits 50 KiB; 1 = always one; 3 = two whenever they fit
    int    ell_wgs = 0;         // CGH_ELL_WGS: workgroups per CU of the ELL SpMV's tile walk (0 = by kernel: 5 narrow byte-coded rows, else 8)
    double llc_mb = 224.0;      // CGH_LLC_MB: working sets below this many MiB run without streaming hints
    int    csr_window = 1024;   // CGH_CSR_WINDOW: entries per row block of the CSR stream kernel (512 | 1024 | 2048)
    int    csr_exact = 0;       // CGH_CSR_EXACT: every CSR row summed strictly left to right
    int64_t xsum_min_rows = 4096; // CGH_XSUM_MIN_ROWS: shortest vector taking the order-exact parallel sums (-1 never)
    int    xsum_xupd = 1;       // CGH_XSUM_XUPD: the fused launch of the sums of r also carries x += alpha p
    int    xsum_fused = 1;      // CGH_XSUM_FUSED: summaries and walk of the order-exact sums in one launch (0: two launches)
    int    xsum_tail = 1;       // CGH_XSUM_TAIL: few cells: the vector update behind a sum's verdict stays in the sum's launch
    int    serpentine = 0;      // CGH_SERPENTINE: alternate the sweep direction of the iteration's kernels
    int    vec_variant = -1;    // CGH_VEC_VARIANT: cache-hint variant of the vector kernels, -1 = by working-set size
    int    graph = 0;           // CGH_GRAPH: replay batches of iterations from HIP graphs
    int    persistent = 1;      // CGH_PERSISTENT: throughput-mode solves of <= 131072 rows run as one resident launch per batch
    int64_t graph_max_rows = 1 << 21; // CGH_GRAPH_MAX_ROWS
    int csrb_rpw = 1024;        // CGH_CSRB_RPW: rows per wave-block of the column-blocked CSR (512 | 1024 | 2048)
    int csrb_slab_bits = 17;    // CGH_CSRB_SLAB_BITS: log2 of the columns per slab (2^17 doubles = 1 MiB of x)
    int csrb_wgs_per_cu = 0;    // CGH_CSRB_WGS: workgroups per CU of the column-blocked kernel (0 = what LDS admits)
    int csrb_window = 3;        // CGH_CSRB_WINDOW: slabs a wave may run ahead of the pack (0 = no pacing)
    int csrb_quorum_div = 2;    // CGH_CSRB_QUORUM: a gate opens once all but 1/div of the group's workgroups have passed
    int csrb_lds_atomics = 0;   // CGH_CSRB_LDS_ATOMICS: row sums by LDS atomics instead of segmented scan + read-add-write
    int sell_sigma = 65536;     // CGH_SELL_SIGMA: sorting window (rows) of the sigma-sorted sliced ELL
    int pb_span_bits = 14;      // CGH_PB_SPAN_BITS: log2 of the most rows a block of the two-phase layout holds (13 | 14)
    int64_t pb_target_entries = 131072; // CGH_PB_TARGET: entries per block the partition aims at
    int pb_split = 1;           // CGH_PB_SPLIT: workgroups sharing one column block in phase 1
    int pb_ordered = 1;         // CGH_PB_ORDERED: 1 = phase 2 adds in stream order (same bits on every run; the default); 0 = as the LDS atomics retire (2 % faster)
};
extern cgh_config g_cfg;
// What a cgh_config_* setter writes: the field of g_cfg now, and again after every later cgh_init / cgh_config_reload
// (which start from the defaults and the environment) -- until cgh_config_reset() forgets the setters' values.
void cgh_config_set(int *field_of_g_cfg, int v);
void cgh_config_set(int64_t *field_of_g_cfg, int64_t v);

// ------------------------------------------------------------ launch shapes
constexpr int CGH_BLOCK = 256;          // 4 waves of 64
constexpr int CGH_MAX_PARTIALS = 2048;  // upper bound on any reducing grid
constexpr int CGH_HIST = 4;             // doubles per history entry of the CG driver: res, rr, alpha, beta

// Number of blocks for a grid-stride streaming / reducing kernel over `tiles` tiles.
static inline int cgh_stream_grid(int64_t tiles, int blocks_per_cu)
{
    int64_t g = (int64_t)g_cgh.num_cu * blocks_per_cu;
    if (g > CGH_MAX_PARTIALS) g = CGH_MAX_PARTIALS;
    if (g > tiles) g = tiles;
    if (g < 1) g = 1;
    return (int)g;
}

// --------------------------------------------------------------- kernels (host launchers)
// spmv.hip
int cgh_launch_spmv(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                    const int *done_flag, hipStream_t s, const cgh_spmv_opts *opts = nullptr);
// spmv_ell.hip: rows [rb, re) (+ [rb2, re2)) of an ELL matrix; x may be longer than n (multi-GPU halo entries follow)
int cgh_launch_spmv_ell_range(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                              const int *done, hipStream_t s, int64_t rb, int64_t re, const cgh_spmv_opts *opts = nullptr);
int cgh_launch_spmv_ell_range2(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                               const int *done, hipStream_t s, int64_t rb, int64_t re, int64_t rb2, int64_t re2,
                               const cgh_spmv_opts *opts = nullptr);
int cgh_spmv_grid(const cgh_matrix *m);
void cgh_ell_dict_free(const cgh_matrix *m);

// convert.hip
cgh_matrix *cgh_matrix_from_entries_device(int64_t n, int64_t U, const double *d_diag, const double *d_upper,
                                           const int *d_rows, const int *d_cols, int layout, bool any_entries = false);

// cg.hip: cgh_cg_solve with the history handed over window by window (dropin.hip prints from it)
typedef void (*cgh_history_sink)(void *ctx, int first, int count, const double *entries, int converged_iteration,
                                 const cgh_cg_report *report);
int cgh_cg_solve_sink(const cgh_matrix *m, const double *d_b, double *d_x, const cgh_cg_opts *opts, cgh_cg_report *rep,
                      cgh_history_sink sink, void *ctx);

// xsum.hip -- order-exact parallel sums (the strict mode of large vectors)
struct cgh_xsum;
cgh_xsum *cgh_xsum_create(int64_t n);
void      cgh_xsum_free(cgh_xsum *w);
// the same workspace in memory of the caller's (no allocation at all): see xsum.hip
size_t    cgh_xsum_object_bytes(void);
size_t    cgh_xsum_device_bytes(int64_t n);
cgh_xsum *cgh_xsum_place(void *obj, size_t obj_bytes, int64_t n, void *dev, size_t dev_bytes, hipStream_t s);
// Same contract as the seq_sums kernel (cg_kernels.h): dot = sum a[i]*b[i] (a != NULL), l1 = sum |c[i]| (c != NULL), both
// added strictly left to right; mode 0: out = {dot, l1}; 1: alpha; 2: res, rr, beta, done; 3: initial res, rr.
// a, b, c must either be (a, b, NULL), (NULL, NULL, c) or (v, v, v).
// presummed: the approximate per-cell sums (128 elements each; slot 0 = dot at [0, cells), slot 1 = l1 at [cells, 2 cells))
// were already written to cgh_xsum_cellsums() by the kernel that produced the vector; else a sweep computes them.
// xu_x / xu_p: vector updates the launch may take along.  Mode 2 (a == b == c = r), xu_x = x, xu_p = p: x += alpha p of the
// iteration in flight (alpha from the history) -- *xu_taken = 1, the caller's cg_update_xp then leaves x alone; with few cells
// also p = r + beta p behind the verdict (*xu_taken = 2: nothing is left for cg_update_xp).  Mode 1 (a = p, b = Ap), xu_x = r,
// few cells: r -= alpha Ap behind the verdict, with the per-cell sums of the new r (*xu_taken = 2: no cg_update_r).
int cgh_xsum_launch(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, int mode, double *hist,
                    int iter, double tol, int *done, double *out, hipStream_t s, bool presummed = false, double *xu_x = nullptr,
                    const double *xu_p = nullptr, int *xu_taken = nullptr);
// a vector spread over several ranks: each rank's part in turn (s_in -> s_out, device), then the verdict on every rank
int cgh_xsum_launch_part(cgh_xsum *w, int64_t n, const double *a, const double *b, const double *c, const double *s_in,
                         double *s_out, const int *done, hipStream_t s, bool presummed);
int cgh_xsum_finalize(const double *sums, int mode, double *hist, int iter, double tol, int *done, hipStream_t s);
// ... or composed: every rank summarises its part at once, the ranks exchange at most 32 batch summaries each and every
// rank walks them all (xsum.hip, xc_walk).  Collective over the ranks of L (mailboxes connected); idx numbers the sums of a
// solve (the same on every rank); offsets: 2 doubles of device scratch.  CGH_EINVAL: the part is too long for one message.
struct cgh_link;
int cgh_xsum_composed_batches(const cgh_xsum *w);
int cgh_xsum_launch_composed(cgh_xsum *w, const cgh_link *L, int64_t n, const double *a, const double *b, const double *c, int mode,
                             double *hist, int iter, double tol, int *done, bool check_done, unsigned epoch, unsigned idx, double *offsets,
                             hipStream_t s, bool presummed);
double *cgh_xsum_cellsums(cgh_xsum *w);
double *cgh_xsum_groupsums(cgh_xsum *w);       // [2][groups of 64 cells]: a producer may leave them too ...
void    cgh_xsum_groups_written(cgh_xsum *w);   // ... and says so before the launch that consumes them
int64_t cgh_xsum_cells(const cgh_xsum *w);

// p2p.hip -- scalar all-reduce over peer stores (multi-GPU path); W-interleaved partials in, W sums out
int cgh_p2p_launch(int W, const double *partials, int count, double *out, hipStream_t s);
bool cgh_ell_vector_path(const cgh_matrix *m, const double *x, const double *y);  // spmv_ell.hip: two rows per lane, 16-byte loads
bool cgh_fits_llc(const cgh_matrix *m, double times);   // spmv_ell.hip: one iteration's working set fits `times` Infinity Caches
struct cgh_link;
struct cgh_ll_halo;
int cgh_p2p_link(cgh_link *L);   // the connected mailboxes as the iteration's kernels see them (halo_ll.h)
