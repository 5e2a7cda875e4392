// dist.hip -- multi-GPU CG: one process per GPU, row-block partition, halo exchange of p and the two
// scalar reductions over RCCL (xGMI).  New module, no reference counterpart.
//
// Per iteration on rank g (s = compute stream, c = communication stream):
//     c: wait(p updated) ; pack boundary p entries (skipped when the list is one contiguous run) ;
//        ncclGroup{ ncclSend to / ncclRecv from each neighbour, straight into the halo tail of p } ; record(halo)
//     s: SpMV of the interior rows [a,b) (they reference no halo column)            -- overlaps the exchange
//        wait(halo) ; SpMV of the boundary rows [0,a) and [b,n_loc)
//        local p.Ap -> ncclAllReduce(1 double) ; r update (+ local r.r, |r|_1) -> ncclAllReduce(2 doubles) ;
//        x and p update, convergence flag (identical on every rank: all see the same reduced scalars)
// Scalars stay on the device; the host only queues work and polls the flag every `check_every` iterations.
//
// xGMI is point-to-point (a dedicated link per GPU pair), so the neighbour exchange is plain send/recv pairs,
// not a ring; the all-reduces carry 8-16 bytes and are pure latency.
//
// That is the FALLBACK.  When the ranks can map each other's memory (hipIpc over xGMI; cgh_comm_init tries it and all
// ranks agree), RCCL only bootstraps and an iteration is three kernels on ONE stream with no event and no library call:
//     s: SpMV, the rows with halo columns last: they read those from the deposit buffer the neighbours wrote into ;
//        dist_update_r  (prologue: p.Ap all-reduced through the peers' mailboxes) ;
//        dist_update_xp (prologue: r.r, |r|_1 all-reduced ; epilogue: boundary rows of the new p pushed to the neighbours)
// (halo_ll.h, dist_kernels.h; measured: a cross-stream event costs ~11 us each way, a small kernel ~5 us.)
//
// A second driver (cgh_dist_group_*) runs G ranks inside ONE process on ONE device with the exchange and the
// reductions done by device kernels: it exists so that every kernel, index list and buffer layout of this
// path is exercised by the single-GPU test box; only the RCCL calls themselves are not.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

#include "cg_kernels.h"
#include "dist_kernels.h"
#include "partition.h"

bool cgh_spmv_ell_halo_capable(const cgh_matrix *m, const double *x, double *y, int64_t rb, int64_t re, int64_t rb2, int64_t re2);
int cgh_launch_spmv_ell_halo(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                             const int *done, hipStream_t s, int64_t rb, int64_t re, int64_t rb2, int64_t re2,
                             const cgh_ll_halo *hl);

// ------------------------------------------------------------------------------------- RCCL, loaded lazily
namespace {

struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // two communicators, one per stream: halo send/recv lives on the communication stream, the scalar all-reduces
    // on the compute stream; operations of ONE communicator are never in flight on two streams at once
    ncclComm_t comm = nullptr;       // all-reduces (compute stream)
    ncclComm_t comm_halo = nullptr;  // neighbour exchange (communication stream)
    int        nranks = 0, rank = -1;
    bool       p2p = false;          // the scalar all-reduces go through p2p.hip (set up and self-tested in cgh_comm_init)
    unsigned   epoch = 0;            // solves started so far (part of the tags of halo_ll.h)
} g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return CGH_OK;
    // The ROCm installation's RCCL (the one this library's HIP runtime belongs to) first.  RTLD_LOCAL on purpose: a
    // global load promotes librccl.so.1 AND its dependencies (libamdhip64.so.7, libhsa-runtime64.so.1) into the
    // process-wide lookup scope, where their hip*/nccl* symbols interpose those of a second ROCm set loaded later
    // (PyTorch bundles one under other file names, so the loader does not merge them); a process then runs half of
    // one runtime against the other and aborts in the exit handlers (tools/exit_probe.py).
    // Every entry point is taken with dlsym from the handle, so nothing here needs the global scope.
    const char *names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
        g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        cgh_set_error("RCCL: cannot load librccl.so (%s)", dlerror());
        return CGH_ECOMM;
    }
#define CGH_SYM(field, name)                                                          \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
    if (!g_rccl.field) {                                                              \
        cgh_set_error("RCCL: symbol %s missing", name);                               \
        return CGH_ECOMM;                                                             \
    }
    CGH_SYM(GetUniqueId, "ncclGetUniqueId")
    CGH_SYM(CommInitRank, "ncclCommInitRank")
    CGH_SYM(CommDestroy, "ncclCommDestroy")
    CGH_SYM(AllReduce, "ncclAllReduce")
    CGH_SYM(AllGather, "ncclAllGather")
    CGH_SYM(Send, "ncclSend")
    CGH_SYM(Recv, "ncclRecv")
    CGH_SYM(GroupStart, "ncclGroupStart")
    CGH_SYM(GroupEnd, "ncclGroupEnd")
    CGH_SYM(GetErrorString, "ncclGetErrorString")
#undef CGH_SYM
    return CGH_OK;
}

#define CGH_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            cgh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
            return CGH_ECOMM;                                                                       \
        }                                                                                           \
    } while (0)

// ------------------------------------------------------------------------------------------ small kernels
__global__ __launch_bounds__(256) void gather_rows(int64_t count, const int *__restrict__ idx, const double *__restrict__ src,
                                                  double *__restrict__ dst)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < count; k += (int64_t)gridDim.x * 256) dst[k] = src[idx[k]];
}

// gs_local[slot..slot+W) = sum of W-interleaved partials (fixed order)
template <int W>
__global__ __launch_bounds__(CGH_BLOCK) void reduce_local(const double *__restrict__ partials, int count, double *__restrict__ out)
{
    // one workgroup per scalar (gridDim.x == W): the W sums run side by side instead of one after the other
    __shared__ double lds[4];
    const int         w = blockIdx.x;
    const double      v = block_sum_partials(partials + w, count, W, lds);
    if (threadIdx.x == 0) out[w] = v;
}

__global__ void finish_init_dist(const double *__restrict__ gs, double *hist)
{
    hist[0] = gs[2] / 2.0;
    hist[1] = gs[1];
    hist[2] = NAN;
    hist[3] = NAN;
}

// single-process rank group: out_g[w] = sum over ranks (rank order) of local_g[w], for every rank g
__global__ void group_allreduce(int nranks, double *const *__restrict__ locals, double *const *__restrict__ outs, int first, int count)
{
    const int w = first + threadIdx.x;
    if ((int)threadIdx.x >= count) return;
    double s = 0.0;
    for (int g = 0; g < nranks; ++g) s += locals[g][w];
    for (int g = 0; g < nranks; ++g) outs[g][w] = s;
}

// z-slab share of the seeded Poisson stencil (see generate.hip for the global matrix): rows of planes [z0,z1),
// local column ids: owned c-lo | lower plane n_loc + (c-(lo-nxny)) | upper plane n_loc + h_lo + (c-hi)
template <int POINTS>
__global__ __launch_bounds__(256) void poisson_ell_slab(int nx, int ny, int nz, int z0, int z1, int K, uint64_t seed,
                                                       double *__restrict__ diag, double *__restrict__ vals,
                                                       int *__restrict__ cols, double *__restrict__ b)
{
    const int64_t nxy = (int64_t)nx * ny, lo = nxy * z0, hi = nxy * z1, nl = hi - lo;
    const int64_t h_lo = z0 > 0 ? nxy : 0;
    for (int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x; li < nl; li += (int64_t)gridDim.x * 256) {
        const int64_t i = lo + li;
        const int     x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / nxy);
        int           slot = 0;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dz == 0 && dy == 0 && dx == 0) continue;
                    if (POINTS == 7 && (dz != 0) + (dy != 0) + (dx != 0) != 1) continue;
                    const int xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || xx >= nx || yy < 0 || yy >= ny || zz < 0 || zz >= nz) continue;
                    const int64_t c = i + dx + (int64_t)nx * dy + nxy * dz;
                    int64_t       lc;
                    if (c >= lo && c < hi) lc = c - lo;
                    else if (c < lo) lc = nl + (c - (lo - nxy));
                    else lc = nl + h_lo + (c - hi);
                    vals[(int64_t)slot * nl + li] = -1.0;
                    cols[(int64_t)slot * nl + li] = (int)lc;
                    ++slot;
                }
        for (; slot < K; ++slot) {
            vals[(int64_t)slot * nl + li] = 0.0;
            cols[(int64_t)slot * nl + li] = -1;
        }
        diag[li] = (double)(POINTS - 1) + 1e-3 * cgh_hash01(seed, (uint64_t)i);
        if (b != nullptr) b[li] = cgh_hash01(seed + 1, (uint64_t)i) - 0.5;
    }
}

inline int grid_for(int64_t work) { return cgh_stream_grid((work + 255) / 256, 8); }

// stored entries whose column is a halo entry (id >= n_loc)
__global__ __launch_bounds__(256) void count_halo_refs(int64_t total, const int *__restrict__ cols, int n_loc, unsigned long long *out)
{
    unsigned long long c = 0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) c += cols[k] >= n_loc;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

} // namespace

// ------------------------------------------------------------------------------------------ per-rank state
struct cgh_dist_nb {
    int      rank = -1;
    int     *d_send_idx = nullptr;   // local rows to send (device), NULL when contiguous
    int64_t  send_cnt = 0;
    int64_t  contig_start = -1;      // >= 0: the send list is rows [contig_start, contig_start+send_cnt)
    double  *d_sendbuf = nullptr;    // packed boundary entries (only when not contiguous)
    int64_t  recv_off = 0, recv_cnt = 0;
};

struct cgh_dist {
    int        nranks = 1, rank = 0;
    int64_t    n_loc = 0, n_halo = 0, nnz_off = 0;
    int64_t    int_begin = 0, int_end = 0;
    cgh_matrix mat;                   // local slot-major ELL; columns index the extended p vector
    std::vector<cgh_dist_nb> nbs;
    // solver state
    double *p = nullptr;              // [n_loc + n_halo]: owned entries, then the halo
    double *r = nullptr, *Ap = nullptr, *x = nullptr;
    double *partA = nullptr, *partB = nullptr, *hist = nullptr;
    double *gs_local = nullptr, *gs = nullptr;   // [0] p.Ap  [1] r.r  [2] |r|_1   (local / reduced over ranks)
    int    *done = nullptr;
    int     capacity = 0, queued = 0, vgrid = 0;
    double  tol = 0.0;
    hipEvent_t ev_p = nullptr, ev_halo = nullptr;
    std::vector<hipEvent_t> kev;
    // group driver only
    double **d_locals = nullptr, **d_outs = nullptr;
    // peer-to-peer halo (optional, see "peer-to-peer halo" below): the neighbours deposit their boundary entries here
    char    *hbuf = nullptr;                       // fine-grained, IPC-shared: 2 slots x n_halo entries x 2 tagged words
    std::vector<void *> h_opened;                  // per neighbour: its hbuf as mapped here (hipIpcOpenMemHandle)
    std::vector<halo_xfer> h_push;                 // per neighbour: where my boundary entries go
    halo_xfer *d_push = nullptr;                   // the same on the device (stand-alone push kernel)
    int     *d_herr = nullptr;                     // device: 1 after a timed-out wait
    bool     halo_p2p = false;
    bool     fused_push = false;                   // dist_update_xp pushes (all send ranges contiguous, at most CGH_PUSH_MAX)
    bool     loopback = false;                     // timing harness: every neighbour is this rank itself (cgh_dist_loopback)
    unsigned epoch = 0;                            // number of this solve (same on every rank): part of every tag
    int      pushed_iter = 0;                      // the exchange dist_update_xp has already pushed
    bool     ll_spmv = false;                      // the boundary SpMV reads the deposit itself (vector path)
    int64_t  halo_refs = -1;                       // stored entries with a halo column (counted at the first solve)
    // bit-identical mode (cgh_dist_set_strict): the dots and the norm are the reference's left-to-right sums over the
    // GLOBAL vector -- order-exact sums (xsum.hip) chained through the ranks in rank order
    bool      strict = false;
    cgh_xsum *xs = nullptr;                        // this rank's part of the sums
    bool      composed = false;                    // the sums are composed from the ranks' summaries (xsum.hip, xc_walk) instead of chained rank by rank
    double   *chain = nullptr;                     // [0..1] running exact sums as they travel, [2..3] this rank's contribution, [4..5] zeros
};

namespace {

void free_solver_state(cgh_dist *d)
{
    for (hipEvent_t e : d->kev) (void)hipEventDestroy(e);
    d->kev.clear();
    cgh_free(d->p);
    cgh_free(d->r);
    cgh_free(d->Ap);
    cgh_free(d->partA);
    cgh_free(d->partB);
    cgh_free(d->hist);
    cgh_free(d->gs_local);
    cgh_free(d->gs);
    cgh_free(d->done);
    cgh_free(d->d_locals);
    cgh_free(d->d_outs);
    cgh_free(d->chain);
    cgh_xsum_free(d->xs);

    d->chain = nullptr;
    d->xs = nullptr;
    d->p = d->r = d->Ap = d->partA = d->partB = d->hist = d->gs_local = d->gs = nullptr;
    d->done = nullptr;
    d->d_locals = d->d_outs = nullptr;
}

int alloc_solver_state(cgh_dist *d, int capacity)
{
    free_solver_state(d);
    const size_t nl = (size_t)d->n_loc;
    d->capacity = capacity;
    d->p = (double *)cgh_alloc(sizeof(double) * (nl + (size_t)d->n_halo));
    d->r = (double *)cgh_alloc(sizeof(double) * nl);
    d->Ap = (double *)cgh_alloc(sizeof(double) * nl);
    d->partA = (double *)cgh_alloc(sizeof(double) * 3 * CGH_MAX_PARTIALS);
    d->partB = (double *)cgh_alloc(sizeof(double) * 2 * CGH_MAX_PARTIALS);
    d->hist = (double *)cgh_alloc(sizeof(double) * HIST * (size_t)(capacity + 1));
    d->gs_local = (double *)cgh_alloc(sizeof(double) * 8);
    d->gs = (double *)cgh_alloc(sizeof(double) * 8);
    d->done = (int *)cgh_alloc(sizeof(int) * 4);
    d->vgrid = cgh_stream_grid((d->n_loc + 2 * CGH_BLOCK - 1) / (2 * CGH_BLOCK), 8);
    // Every workgroup of the two vector kernels waits for the peers' reduction messages.  Ranks on GPUs of their own
    // never notice; ranks SHARING one GPU (the test harness) must leave room for each other's workgroups, or the
    // waiting ones keep out the one that would send: CGH_DIST_GRID_MAX caps the grid for that case.
    if (const char *env = getenv("CGH_DIST_GRID_MAX")) {
        const int cap = atoi(env);
        if (cap >= 1 && d->vgrid > cap) d->vgrid = cap;
    }
    if (!d->p || !d->r || !d->Ap || !d->partA || !d->partB || !d->hist || !d->gs_local || !d->gs || !d->done) return CGH_ENOMEM;
    if (!d->ev_p) {
        CGH_HIP(hipEventCreateWithFlags(&d->ev_p, hipEventDisableTiming));
        CGH_HIP(hipEventCreateWithFlags(&d->ev_halo, hipEventDisableTiming));
    }
    return CGH_OK;
}

// Ap[rows] = (A p)[rows] for the three row ranges; partials of p.Ap land in partA[off..); returns the partial count.
int spmv_ranges(cgh_dist *d, bool interior, bool boundary, bool with_dot, int *count_io, hipStream_t s, const int *done)
{
    const int64_t a = d->int_begin, b = d->int_end, nl = d->n_loc;
    int           g = 0, rc;
    if (interior) {
        cgh_spmv_opts so;
        so.interior_rows = true;  // rows [a, b) read no halo entry: wide rows may gather from LDS (spmv_ell_staged)
        rc = cgh_launch_spmv_ell_range(&d->mat, d->p, d->Ap, with_dot ? d->partA + *count_io : nullptr, &g, done, s, a, b, &so);
        if (rc) return rc;
        *count_io += g;
    }
    if (boundary) {
        rc = cgh_launch_spmv_ell_range2(&d->mat, d->p, d->Ap, with_dot ? d->partA + *count_io : nullptr, &g, done, s, 0, a, b, nl);
        if (rc) return rc;
        *count_io += g;
    }
    return CGH_OK;
}

// RCCL halo exchange of d->p on the communication stream.
int exchange_rccl(cgh_dist *d, hipStream_t c)
{
    if (d->nbs.empty()) return CGH_OK;
    for (cgh_dist_nb &nb : d->nbs)
        if (nb.contig_start < 0 && nb.send_cnt > 0)
            hipLaunchKernelGGL(gather_rows, dim3(grid_for(nb.send_cnt)), dim3(256), 0, c, nb.send_cnt, (const int *)nb.d_send_idx,
                               (const double *)d->p, nb.d_sendbuf);
    CGH_HIP(hipGetLastError());
    CGH_NCCL(g_rccl.GroupStart());
    for (cgh_dist_nb &nb : d->nbs) {
        if (nb.send_cnt > 0) {
            const double *src = nb.contig_start >= 0 ? d->p + nb.contig_start : nb.d_sendbuf;
            CGH_NCCL(g_rccl.Send(src, (size_t)nb.send_cnt, ncclDouble, nb.rank, g_rccl.comm_halo, c));
        }
        if (nb.recv_cnt > 0)
            CGH_NCCL(g_rccl.Recv(d->p + d->n_loc + nb.recv_off, (size_t)nb.recv_cnt, ncclDouble, nb.rank, g_rccl.comm_halo, c));
    }
    CGH_NCCL(g_rccl.GroupEnd());
    return CGH_OK;
}

int finish_neighbours(cgh_dist *d)
{
    for (cgh_dist_nb &nb : d->nbs)
        if (nb.contig_start < 0 && nb.send_cnt > 0) {
            nb.d_sendbuf = (double *)cgh_alloc(sizeof(double) * (size_t)nb.send_cnt);
            if (!nb.d_sendbuf) return CGH_ENOMEM;
        }
    return CGH_OK;
}


// ------------------------------------------------------------------------------------------ peer-to-peer halo
// The same exchange without a library and without a launch of its own.  Every rank owns a deposit buffer (fine-grained
// device memory shared through hipIpc): two slots, each an image of the halo tail of p with every entry as two tagged
// 64-bit words (halo_ll.h; per slot the low words of all entries, then the high words).  The kernel that writes p (dist_update_xp) also stores the rows a neighbour needs into that
// neighbour's deposit, tagged with the NEXT iteration's exchange; the boundary-row SpMV of that iteration reads its halo
// columns straight from its own deposit and waits, per entry, only if the tag is not there yet.  Two slots alternate: a
// rank cannot push exchange k+2 before every neighbour has consumed exchange k (it needs their reduction messages of
// iteration k+1, which they send after their SpMV of k+1).  Tags count (solve, iteration), so ranks stay in step
// whatever they queue after convergence.
constexpr int    HALO_MAX_NB = 16;
constexpr size_t HALO_DESC_BYTES = 512;
struct halo_desc {                                // what a rank tells the others about its deposit buffer
    char    handle[64];
    int32_t rank, n;
    int64_t n_halo;
    struct {
        int32_t src, pad;
        int64_t recv_off, cnt;
    } e[HALO_MAX_NB];
};
static_assert(sizeof(halo_desc) <= HALO_DESC_BYTES, "descriptor size");

void halo_release(cgh_dist *d)
{
    for (void *p : d->h_opened)
        if (p) (void)hipIpcCloseMemHandle(p);
    d->h_opened.clear();
    d->h_push.clear();
    cgh_free(d->d_push);
    cgh_free(d->d_herr);
    d->d_push = nullptr;
    d->d_herr = nullptr;
    if (d->hbuf) (void)hipFree(d->hbuf);
    d->hbuf = nullptr;
    d->halo_p2p = false;
    d->fused_push = false;
}

int halo_descriptor(cgh_dist *d, char *out)
{
    if ((int)d->nbs.size() > HALO_MAX_NB) {
        cgh_set_error("peer-to-peer halo: more than %d neighbours", HALO_MAX_NB);
        return CGH_EINVAL;
    }
    if (!d->hbuf) {
        const size_t bytes = (size_t)(d->n_halo > 0 ? d->n_halo : 1) * 2 * 2 * sizeof(uint64_t);  // 2 slots x 2 words per entry
        void        *p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            CGH_HIP(hipMalloc(&p, bytes));
        }
        d->hbuf = (char *)p;
        CGH_HIP(hipMemset(d->hbuf, 0, bytes));  // tag 0 = nothing deposited yet
        CGH_HIP(hipDeviceSynchronize());
    }
    halo_desc h;
    memset(&h, 0, sizeof h);
    hipIpcMemHandle_t ih;
    CGH_HIP(hipIpcGetMemHandle(&ih, d->hbuf));
    memcpy(h.handle, &ih, 64);
    h.rank = d->rank;
    h.n = (int32_t)d->nbs.size();
    h.n_halo = d->n_halo;
    for (size_t i = 0; i < d->nbs.size(); ++i) {
        h.e[i].src = d->nbs[i].rank;
        h.e[i].recv_off = d->nbs[i].recv_off;
        h.e[i].cnt = d->nbs[i].recv_cnt;
    }
    memset(out, 0, HALO_DESC_BYTES);
    memcpy(out, &h, sizeof h);
    return CGH_OK;
}

// all: nranks x 512 bytes, rank order.  Maps every neighbour's deposit buffer and builds the transfer lists.
int halo_connect(cgh_dist *d, const char *all)
{
    if (!d->hbuf) return CGH_EINVAL;
    const size_t nnb = d->nbs.size();
    // (a second connect -- by hand after the automatic one -- starts from a clean slate, keeping the deposit buffer)
    for (void *p : d->h_opened)
        if (p) (void)hipIpcCloseMemHandle(p);
    cgh_free(d->d_push);
    cgh_free(d->d_herr);
    d->d_push = nullptr;
    d->d_herr = nullptr;
    d->halo_p2p = false;
    d->h_push.assign(nnb, halo_xfer{nullptr, nullptr, 0, 0, 0});
    d->h_opened.assign(nnb, nullptr);
    bool contiguous = true;
    for (size_t i = 0; i < nnb; ++i) {
        const cgh_dist_nb &nb = d->nbs[i];
        halo_desc          h;
        memcpy(&h, all + HALO_DESC_BYTES * (size_t)nb.rank, sizeof h);
        int at = -1;
        for (int k = 0; k < h.n && k < HALO_MAX_NB; ++k)
            if (h.e[k].src == d->rank) at = k;
        if (d->loopback) at = (int)(nnb - 1 - i);  // what leaves through the lower face comes back through the upper one
        if (h.rank != nb.rank || at < 0 || h.e[at].cnt != nb.send_cnt) {
            cgh_set_error("peer-to-peer halo: rank %d does not expect %lld entries from rank %d", nb.rank, (long long)nb.send_cnt, d->rank);
            return CGH_ECOMM;
        }
        hipIpcMemHandle_t ih;
        memcpy(&ih, h.handle, 64);
        void *base = nullptr;
        if (nb.rank == d->rank) {
            base = d->hbuf;
        } else {
            const hipError_t e = hipIpcOpenMemHandle(&base, ih, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                cgh_set_error("peer-to-peer halo: hipIpcOpenMemHandle(rank %d) -> %s", nb.rank, hipGetErrorString(e));
                return CGH_ECOMM;
            }
            d->h_opened[i] = base;
        }
        halo_xfer &x = d->h_push[i];
        x.dst = reinterpret_cast<uint64_t *>(base) + (size_t)h.e[at].recv_off;
        x.idx = nb.contig_start >= 0 ? nullptr : nb.d_send_idx;
        x.start = nb.contig_start >= 0 ? nb.contig_start : 0;
        x.cnt = nb.send_cnt;
        x.plane = h.n_halo;
        contiguous = contiguous && nb.contig_start >= 0;
    }
    d->d_push = (halo_xfer *)cgh_alloc(sizeof(halo_xfer) * (nnb ? nnb : 1));
    d->d_herr = (int *)cgh_alloc(sizeof(int));
    if (!d->d_push || !d->d_herr) return CGH_ENOMEM;
    int rc = nnb ? cgh_h2d(d->d_push, d->h_push.data(), sizeof(halo_xfer) * nnb) : CGH_OK;
    if (rc == CGH_OK) rc = cgh_memset(d->d_herr, 0, sizeof(int));
    if (rc) return rc;
    const char *env = getenv("CGH_FUSED_PUSH");
    d->fused_push = contiguous && nnb <= (size_t)CGH_PUSH_MAX && !(env && atoi(env) == 0);
    d->halo_p2p = true;
    return CGH_OK;
}

// the stand-alone push of exchange `iter` (first exchange of a solve; send lists that dist_update_xp cannot serve)
void halo_push_launch(cgh_dist *d, int iter, hipStream_t s)
{
    int64_t most = 1;
    for (const cgh_dist_nb &nb : d->nbs) most = nb.send_cnt > most ? nb.send_cnt : most;
    const int per = (int)((most + 511) / 512 < 64 ? (most + 511) / 512 : 64);
    hipLaunchKernelGGL(halo_push_ll, dim3(per, (unsigned)d->nbs.size()), dim3(256), 0, s, (const halo_xfer *)d->d_push, (const double *)d->p,
                       cgh_ll_tag(d->epoch, (unsigned)iter), iter & 1, (const int *)d->done);
}

cgh_ll_halo halo_view(const cgh_dist *d, int iter)
{
    cgh_ll_halo h;
    h.dep = reinterpret_cast<const uint64_t *>(d->hbuf) + (size_t)(iter & 1) * 2 * (size_t)d->n_halo;
    h.plane = d->n_halo;
    h.nl = d->n_loc;
    h.tag = cgh_ll_tag(d->epoch, (unsigned)iter);
    h.err = d->d_herr;
    return h;
}

} // namespace

// Ranks without RCCL (cgh_comm_init_local): every rank's `bytes` bytes (a multiple of 4), in rank order, through the
// mailboxes.  A 32-bit word travels as an integer-valued double inside a sum whose other terms are zero, so the bits
// arrive exactly.  Collective; slow (one exchange per two words and rank) but only used at set-up and by launchers.
static int local_allgather(const void *mine, size_t bytes, void *all)
{
    const int N = g_rccl.nranks, me = g_rccl.rank;
    if (bytes % 4 || !cgh_p2p_connected()) return CGH_EINVAL;
    const size_t    words = bytes / 4;
    const uint32_t *src = static_cast<const uint32_t *>(mine);
    uint32_t       *dst = static_cast<uint32_t *>(all);
    for (int r = 0; r < N; ++r)
        for (size_t w = 0; w < words; w += 2) {
            double v[2] = {0.0, 0.0};
            if (r == me) {
                v[0] = (double)src[w];
                if (w + 1 < words) v[1] = (double)src[w + 1];
            }
            const int rc = cgh_p2p_allreduce_f64(v, 2);
            if (rc) return rc;
            dst[(size_t)r * words + w] = (uint32_t)v[0];
            if (w + 1 < words) dst[(size_t)r * words + w + 1] = (uint32_t)v[1];
        }
    return CGH_OK;
}

// Peer-to-peer scalar all-reduce: all-gather the mailbox handles over the new communicator, connect, prove the path on
// sums whose value is known, and let the ranks agree.  Any failure anywhere leaves the RCCL all-reduces in place.
static bool setup_p2p(int nranks, int rank)
{
    hipStream_t s = g_cgh.stream;
    char        mine[64];
    bool        ok = cgh_p2p_local_handle(mine) == CGH_OK;
    char       *d_all = (char *)cgh_alloc(64 * (size_t)nranks + 64);
    std::vector<char> all(64 * (size_t)nranks);
    ok = ok && d_all != nullptr && cgh_h2d(d_all + 64 * (size_t)nranks, mine, 64) == CGH_OK;
    // (every rank takes part in the collectives below whatever its own state, or the others would wait for it)
    if (d_all) {
        const ncclResult_t r = g_rccl.AllGather(d_all + 64 * (size_t)nranks, d_all, 64, ncclChar, g_rccl.comm, s);
        ok = ok && r == ncclSuccess && cgh_d2h(all.data(), d_all, all.size()) == CGH_OK;
    }
    ok = ok && cgh_p2p_connect(nranks, rank, all.data()) == CGH_OK;
    double *buf = (double *)cgh_alloc(sizeof(double) * 8);
    double  flag = ok && buf ? 1.0 : 0.0;
    // barrier + agreement #1: everybody connected (mailboxes zeroed) before anybody sends
    if (buf) {
        bool c = cgh_h2d(buf, &flag, sizeof flag) == CGH_OK &&
                 g_rccl.AllReduce(buf, buf + 1, 1, ncclDouble, ncclMin, g_rccl.comm, s) == ncclSuccess &&
                 cgh_d2h(&flag, buf + 1, sizeof flag) == CGH_OK;
        if (!c) flag = 0.0;
    }
    if (flag == 1.0) {
        // self-test: rank r contributes (r+1)*k/2 and -(r+1)*k in round k; the sums are exact small numbers
        for (int k = 1; k <= 6 && ok; ++k) {
            const double v[2] = {(rank + 1) * k * 0.5, -(double)(rank + 1) * k};
            const double tot = nranks * (nranks + 1) / 2.0;
            double       got[2] = {0.0, 0.0};
            ok = cgh_h2d(buf, v, sizeof v) == CGH_OK && cgh_p2p_launch(2, buf, 1, buf + 2, s) == CGH_OK &&
                 cgh_d2h(got, buf + 2, sizeof got) == CGH_OK && got[0] == tot * k * 0.5 && got[1] == -tot * k;
        }
        ok = ok && cgh_p2p_error() == 0;
        // agreement #2: everybody passed
        flag = ok ? 1.0 : 0.0;
        bool c = cgh_h2d(buf, &flag, sizeof flag) == CGH_OK &&
                 g_rccl.AllReduce(buf, buf + 1, 1, ncclDouble, ncclMin, g_rccl.comm, s) == ncclSuccess &&
                 cgh_d2h(&flag, buf + 1, sizeof flag) == CGH_OK;
        if (!c) flag = 0.0;
    }
    cgh_free(buf);
    cgh_free(d_all);
    if (flag != 1.0) {
        cgh_p2p_disconnect();
        if (getenv("CGH_P2P_VERBOSE")) fprintf(stderr, "[cg_hip] rank %d: peer-to-peer all-reduce not used (%s)\n", rank, cgh_last_error());
        return false;
    }
    return true;
}

// With RCCL up and the peer-to-peer path agreed on, connect the halo of a new cgh_dist the same way: descriptors
// all-gathered, every rank maps its neighbours, a min-reduce decides for all.  Collective.
static void auto_connect_halo(cgh_dist *d)
{
    const char *env = getenv("CGH_P2P_HALO");
    if (!g_rccl.p2p || (env && atoi(env) == 0) || g_rccl.nranks != d->nranks) return;
    if (!g_rccl.comm) {
        // ranks without RCCL: the descriptors travel through the mailboxes; a failure anywhere is everybody's failure
        // (there is no other transport to fall back to: cgh_dist_start then refuses)
        const size_t      n = (size_t)d->nranks;
        std::vector<char> all(HALO_DESC_BYTES * n), mine(HALO_DESC_BYTES);
        bool              ok = halo_descriptor(d, mine.data()) == CGH_OK;
        if (!ok) memset(mine.data(), 0, mine.size());
        ok = local_allgather(mine.data(), HALO_DESC_BYTES, all.data()) == CGH_OK && ok;
        ok = ok && halo_connect(d, all.data()) == CGH_OK;
        double v[2] = {ok ? 1.0 : 0.0, 0.0};
        if (cgh_p2p_allreduce_f64(v, 1) != CGH_OK || v[0] != (double)d->nranks) halo_release(d);
        return;
    }
    hipStream_t       s = g_cgh.stream;
    const size_t      n = (size_t)d->nranks;
    std::vector<char> all(HALO_DESC_BYTES * n), mine(HALO_DESC_BYTES);
    bool              ok = halo_descriptor(d, mine.data()) == CGH_OK;
    char             *d_all = (char *)cgh_alloc(HALO_DESC_BYTES * (n + 1));
    double           *flag = (double *)cgh_alloc(sizeof(double) * 2);
    ok = ok && d_all && flag && cgh_h2d(d_all + HALO_DESC_BYTES * n, mine.data(), HALO_DESC_BYTES) == CGH_OK;
    if (d_all) {
        const ncclResult_t r = g_rccl.AllGather(d_all + HALO_DESC_BYTES * n, d_all, HALO_DESC_BYTES, ncclChar, g_rccl.comm, s);
        ok = ok && r == ncclSuccess && cgh_d2h(all.data(), d_all, all.size()) == CGH_OK;
    }
    ok = ok && halo_connect(d, all.data()) == CGH_OK;
    double v = ok ? 1.0 : 0.0;
    if (flag) {
        const bool c = cgh_h2d(flag, &v, sizeof v) == CGH_OK &&
                       g_rccl.AllReduce(flag, flag + 1, 1, ncclDouble, ncclMin, g_rccl.comm, s) == ncclSuccess &&
                       cgh_d2h(&v, flag + 1, sizeof v) == CGH_OK;
        if (!c) v = 0.0;
    } else {
        v = 0.0;
    }
    cgh_free(d_all);
    cgh_free(flag);
    if (v != 1.0) halo_release(d);  // somebody could not: everybody keeps the RCCL send/recv
}

// ------------------------------------------------------------------------------------------ bit-identical mode
// One of the reference's sums over a vector spread over the ranks (vec_dot / vec_l1norm, src/vec.c:67-85 there): rank 0's
// rows come first, so rank q may only start when the exact running sum of ranks < q is known.  Round q: rank q
// summarises its part (binades predicted from the exact incoming sum) and walks it from that sum; everybody else
// contributes zeros; one all-reduce of two doubles (s + 0 + ... + 0 = s exactly) hands the running sums to all ranks.
// After the last round every rank holds the same totals and finalises for itself.  Serial in the number of ranks by
// construction: this mode is for verification against the reference, not for speed.
// `sum` = which sum of the iteration (0: p.Ap or the initial residual, 1: r.r and |r|_1)
static int strict_chain(cgh_dist *d, const double *a, const double *b, const double *c, int mode, int iter, int sum, bool presummed,
                        hipStream_t s)
{
    double *run = d->chain, *mine = d->chain + 2, *zeros = d->chain + 4;
    const int  N = d->nranks;
    const bool frozen_check = mode != 3;
    if (d->composed) {
        // every rank summarises its part at once, the ranks exchange their (at most 32) batch summaries and every rank
        // walks them all: one round instead of N (xsum.hip, "sums over several ranks, composed")
        cgh_link L;
        const int rc = cgh_p2p_link(&L);
        if (rc) return rc;
        return cgh_xsum_launch_composed(d->xs, &L, d->n_loc, a, b, c, mode, d->hist, iter, d->tol, d->done, frozen_check, d->epoch,
                                        (unsigned)iter * 2u + (unsigned)sum, d->chain + 6, s, presummed);
    }
    CGH_HIP(hipMemsetAsync(run, 0, sizeof(double) * 2, s));
    for (int q = 0; q < N; ++q) {
        if (q == d->rank) {
            const int rc = cgh_xsum_launch_part(d->xs, d->n_loc, a, b, c, run, mine, frozen_check ? d->done : nullptr, s, presummed);
            if (rc) return rc;
        }
        const double *contrib = q == d->rank ? mine : zeros;
        if (g_rccl.p2p) {
            cgh_link L;
            const int rc = cgh_p2p_link(&L);
            if (rc) return rc;
            hipLaunchKernelGGL(chain_round, dim3(1), dim3(CGH_BLOCK), 0, s, L, contrib, run, (const int *)d->done, frozen_check ? 1 : 0,
                               d->epoch, ((unsigned)iter * 2u + (unsigned)sum) * (unsigned)N + (unsigned)q);
        } else if (N > 1) {
            CGH_NCCL(g_rccl.AllReduce(contrib, run, 2, ncclDouble, ncclSum, g_rccl.comm, s));
        } else {
            CGH_HIP(hipMemcpyAsync(run, contrib, sizeof(double) * 2, hipMemcpyDeviceToDevice, s));
        }
    }
    CGH_HIP(hipGetLastError());
    return cgh_xsum_finalize(run, mode, d->hist, iter, d->tol, d->done, s);
}

extern "C" {

// ------------------------------------------------------------------------------------------ communicator
int cgh_comm_preload(void) { return load_rccl(); }

int cgh_comm_unique_id(char out[256])
{
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    for (int k = 0; k < 2; ++k) {
        ncclUniqueId id;
        CGH_NCCL(g_rccl.GetUniqueId(&id));
        memcpy(out + 128 * k, &id, 128);
    }
    return CGH_OK;
}

int cgh_comm_init(int nranks, int rank, const char id_bytes[256])
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    int rc = load_rccl();
    if (rc) return rc;
    if (g_rccl.comm) cgh_comm_destroy();
    ncclUniqueId id0, id1;
    memcpy(&id0, id_bytes, 128);
    memcpy(&id1, id_bytes + 128, 128);
    CGH_HIP(hipSetDevice(g_cgh.device));
    CGH_NCCL(g_rccl.CommInitRank(&g_rccl.comm, nranks, id0, rank));
    CGH_NCCL(g_rccl.CommInitRank(&g_rccl.comm_halo, nranks, id1, rank));
    g_rccl.nranks = nranks;
    g_rccl.rank = rank;
    g_rccl.p2p = false;
    const char *env = getenv("CGH_P2P");
    if (!(env && atoi(env) == 0) && nranks <= 16) g_rccl.p2p = setup_p2p(nranks, rank);
    return CGH_OK;
}

// Host-side helper for launchers: all-reduce `count` (<= 64) host doubles over the ranks (op 0 sum, 1 max, 2 min) with the
// communicator of cgh_comm_init; with one value it doubles as a barrier.  Synchronises the compute stream.
int cgh_comm_allreduce_f64(double *h_vals, int count, int op)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!h_vals || count < 1 || count > 64 || op < 0 || op > 2) return CGH_EINVAL;
    if (!g_rccl.comm && g_rccl.p2p && cgh_p2p_connected()) {
        // ranks without RCCL: gather everybody's values through the mailboxes, reduce here in rank order
        const int           N = g_rccl.nranks;
        std::vector<double> all((size_t)N * (size_t)count);
        const int           rc = local_allgather(h_vals, sizeof(double) * (size_t)count, all.data());
        if (rc) return rc;
        for (int i = 0; i < count; ++i) {
            double acc = all[i];
            for (int r = 1; r < N; ++r) {
                const double v = all[(size_t)r * count + i];
                acc = op == 0 ? acc + v : (op == 1 ? (v > acc ? v : acc) : (v < acc ? v : acc));
            }
            h_vals[i] = acc;
        }
        return CGH_OK;
    }
    if (!g_rccl.comm) {
        cgh_set_error("cgh_comm_allreduce_f64: no communicator (cgh_comm_init first)");
        return CGH_ECOMM;
    }
    double *buf = (double *)cgh_alloc(sizeof(double) * 128);
    if (!buf) return CGH_ENOMEM;
    hipStream_t s = g_cgh.stream;
    int         rc = cgh_h2d(buf, h_vals, sizeof(double) * (size_t)count);
    if (rc == CGH_OK) {
        const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
        const ncclResult_t r = g_rccl.AllReduce(buf, buf + 64, (size_t)count, ncclDouble, ops[op], g_rccl.comm, s);
        if (r != ncclSuccess) {
            cgh_set_error("cgh_comm_allreduce_f64: %s", g_rccl.GetErrorString(r));
            rc = CGH_ECOMM;
        }
    }
    if (rc == CGH_OK) rc = cgh_d2h(h_vals, buf + 64, sizeof(double) * (size_t)count);
    cgh_free(buf);
    return rc;
}

// Ranks without RCCL: the scalar reductions and the halo exchange both go peer to peer (the caller connects the mailboxes
// with cgh_p2p_connect and every cgh_dist with cgh_dist_halo_connect, e.g. through files).  This is also how several
// ranks can share ONE GPU, which RCCL refuses -- the test harness of the multi-rank loop.
int cgh_comm_init_local(int nranks, int rank)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (nranks < 1 || rank < 0 || rank >= nranks || !cgh_p2p_connected()) {
        cgh_set_error("cgh_comm_init_local: connect the peer-to-peer mailboxes first (cgh_p2p_connect)");
        return CGH_EINVAL;
    }
    if (g_rccl.comm) return CGH_EINVAL;
    g_rccl.nranks = nranks;
    g_rccl.rank = rank;
    g_rccl.p2p = true;
    return CGH_OK;
}

// This rank's deposit buffer for the peer-to-peer halo as a 512-byte descriptor; every rank's descriptor (rank order)
// goes to cgh_dist_halo_connect on every rank, then a barrier, then cgh_dist_start.
int cgh_dist_halo_descriptor(cgh_dist *d, char out_512_bytes[512])
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || !out_512_bytes) return CGH_EINVAL;
    return halo_descriptor(d, out_512_bytes);
}

int cgh_dist_halo_connect(cgh_dist *d, const char *descriptors)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || !descriptors) return CGH_EINVAL;
    const int rc = halo_connect(d, descriptors);
    if (rc != CGH_OK) {
        const bool had = d->hbuf != nullptr;
        (void)had;
        d->halo_p2p = false;
    }
    return rc;
}

int cgh_dist_halo_is_p2p(const cgh_dist *d) { return d && d->halo_p2p ? 1 : 0; }

// Timing harness for ONE GPU: turn an inner slab (created as rank r of N while the communicator has one rank) into a
// one-rank problem whose two neighbours are this rank itself -- what it pushes through its lower face arrives in its
// own upper halo and vice versa, i.e. the slab's operator closed periodically in z (still symmetric and diagonally
// dominant).  Kernels, bytes, flags and waits per iteration are those of an inner rank of the N-GPU run; only the
// xGMI hop is missing.
int cgh_dist_loopback(cgh_dist *d)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || d->nbs.size() != 2 || !g_rccl.p2p || g_rccl.nranks != 1 || d->nbs[0].send_cnt != d->nbs[1].recv_cnt ||
        d->nbs[1].send_cnt != d->nbs[0].recv_cnt) {
        cgh_set_error("cgh_dist_loopback: needs an inner slab (two neighbours) and a one-rank peer-to-peer communicator");
        return CGH_EINVAL;
    }
    halo_release(d);
    d->nranks = 1;
    d->rank = 0;
    for (cgh_dist_nb &nb : d->nbs) nb.rank = 0;
    d->loopback = true;
    char desc[HALO_DESC_BYTES];
    int  rc = halo_descriptor(d, desc);
    if (rc == CGH_OK) rc = halo_connect(d, desc);
    if (rc != CGH_OK) d->halo_p2p = false;
    return rc;
}

// Use (1) or bypass (0) the peer-to-peer scalar all-reduce for the following solves; it can only be switched on if
// cgh_comm_init connected it.  Every rank must make the same choice.  Returns the state now in force.
int cgh_comm_set_p2p(int on)
{
    if (!g_rccl.comm) return g_rccl.p2p ? 1 : 0;  // without RCCL there is nothing to fall back to
    g_rccl.p2p = on != 0 && cgh_p2p_connected() != 0;
    return g_rccl.p2p ? 1 : 0;
}

void cgh_comm_destroy(void)
{
    cgh_p2p_disconnect();
    g_rccl.p2p = false;
    if (g_rccl.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_rccl.comm);
    if (g_rccl.comm_halo && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_rccl.comm_halo);
    g_rccl.comm = g_rccl.comm_halo = nullptr;
}

// ------------------------------------------------------------------------------------------ construction
cgh_dist *cgh_dist_from_partition(const cgh_partition *P)
{
    CGH_REQUIRE_READY(nullptr);
    if (!P) return nullptr;
    cgh_dist *d = new cgh_dist();
    d->nranks = P->nranks;
    d->rank = P->rank;
    d->n_loc = P->n_loc;
    d->n_halo = (int64_t)P->halo_global.size();
    d->nnz_off = P->nnz_off;
    d->int_begin = P->int_begin;
    d->int_end = P->int_end;
    cgh_matrix &m = d->mat;
    m.n = P->n_loc;
    m.K = P->K;
    m.layout = CGH_LAYOUT_ELL;
    m.nnz_off = P->nnz_off;
    m.stored = P->n_loc * (int64_t)P->K;
    m.diag = (double *)cgh_alloc(sizeof(double) * (size_t)(m.n ? m.n : 1));
    m.vals = (double *)cgh_alloc(sizeof(double) * (size_t)(m.stored ? m.stored : 1));
    m.cols = (int *)cgh_alloc(sizeof(int) * (size_t)(m.stored ? m.stored : 1));
    bool ok = m.diag && m.vals && m.cols;
    ok = ok && cgh_h2d(m.diag, P->diag.data(), sizeof(double) * (size_t)m.n) == CGH_OK;
    ok = ok && cgh_h2d(m.vals, P->vals.data(), sizeof(double) * (size_t)m.stored) == CGH_OK;
    ok = ok && cgh_h2d(m.cols, P->cols.data(), sizeof(int) * (size_t)m.stored) == CGH_OK;
    for (const cgh_neighbour &src : P->nbs) {
        if (!ok) break;
        cgh_dist_nb nb;
        nb.rank = src.rank;
        nb.send_cnt = (int64_t)src.send_idx.size();
        nb.recv_off = src.recv_off;
        nb.recv_cnt = src.recv_cnt;
        bool contig = nb.send_cnt > 0;
        for (size_t k = 1; k < src.send_idx.size() && contig; ++k) contig = src.send_idx[k] == src.send_idx[k - 1] + 1;
        if (contig) {
            nb.contig_start = src.send_idx[0];
        } else if (nb.send_cnt > 0) {
            nb.d_send_idx = (int *)cgh_alloc(sizeof(int) * (size_t)nb.send_cnt);
            ok = nb.d_send_idx && cgh_h2d(nb.d_send_idx, src.send_idx.data(), sizeof(int) * (size_t)nb.send_cnt) == CGH_OK;
        }
        d->nbs.push_back(nb);
    }
    ok = ok && finish_neighbours(d) == CGH_OK;
    if (!ok) {
        cgh_dist_free(d);
        return nullptr;
    }
    auto_connect_halo(d);
    return d;
}

cgh_dist *cgh_dist_poisson(int nx, int ny, int nz, int points, uint64_t seed, int nranks, int rank, double *d_b)
{
    CGH_REQUIRE_READY(nullptr);
    if (nx < 1 || ny < 1 || nz < nranks || nranks < 1 || rank < 0 || rank >= nranks || (points != 7 && points != 27)) {
        cgh_set_error("cgh_dist_poisson: bad arguments (%dx%dx%d, %d points, rank %d of %d; need nz >= ranks)", nx, ny, nz, points,
                      rank, nranks);
        return nullptr;
    }
    const int     z0 = (int)((int64_t)nz * rank / nranks), z1 = (int)((int64_t)nz * (rank + 1) / nranks);
    const int64_t nxy = (int64_t)nx * ny, nl = nxy * (z1 - z0);
    const int64_t h_lo = z0 > 0 ? nxy : 0, h_hi = z1 < nz ? nxy : 0;
    if (nl + h_lo + h_hi > 0x7fffffffLL) {
        cgh_set_error("cgh_dist_poisson: local block exceeds int32 column ids");
        return nullptr;
    }
    auto      m3 = [](int v) { return v < 3 ? v : 3; };
    const int K = points == 27 ? m3(nx) * m3(ny) * m3(nz) - 1 : (m3(nx) - 1) + (m3(ny) - 1) + (m3(nz) - 1);
    cgh_dist *d = new cgh_dist();
    d->nranks = nranks;
    d->rank = rank;
    d->n_loc = nl;
    d->n_halo = h_lo + h_hi;
    // rows needing no halo: everything but the first plane (if a lower neighbour exists) and the last (upper)
    int64_t a = h_lo, b = nl - h_hi;
    if (a > b) a = b = 0;
    a = (a + 1) & ~(int64_t)1;
    if (b < nl) b &= ~(int64_t)1;
    if (a > b) a = b = 0;
    d->int_begin = a;
    d->int_end = b;
    // stored off-diagonals of the slab
    const int64_t inplane = points == 27 ? (3LL * nx - 2) * (3LL * ny - 2) - nxy : 2 * ((int64_t)(nx - 1) * ny + (int64_t)nx * (ny - 1));
    const int64_t perz = points == 27 ? (3LL * nx - 2) * (3LL * ny - 2) : nxy;
    int64_t       nnz = 0;
    for (int z = z0; z < z1; ++z) nnz += inplane + perz * ((z > 0) + (z < nz - 1));
    d->nnz_off = nnz;
    cgh_matrix &m = d->mat;
    m.n = nl;
    m.K = K;
    m.layout = CGH_LAYOUT_ELL;
    m.nnz_off = nnz;
    m.stored = nl * (int64_t)K;
    m.diag = (double *)cgh_alloc(sizeof(double) * (size_t)nl);
    m.vals = (double *)cgh_alloc(sizeof(double) * (size_t)(m.stored ? m.stored : 1));
    m.cols = (int *)cgh_alloc(sizeof(int) * (size_t)(m.stored ? m.stored : 1));
    if (!m.diag || !m.vals || !m.cols) {
        cgh_dist_free(d);
        return nullptr;
    }
    hipStream_t s = g_cgh.stream;
    if (points == 7)
        hipLaunchKernelGGL(poisson_ell_slab<7>, dim3(grid_for(nl)), dim3(256), 0, s, nx, ny, nz, z0, z1, K, seed, m.diag, m.vals, m.cols, d_b);
    else
        hipLaunchKernelGGL(poisson_ell_slab<27>, dim3(grid_for(nl)), dim3(256), 0, s, nx, ny, nz, z0, z1, K, seed, m.diag, m.vals, m.cols, d_b);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        cgh_set_error("cgh_dist_poisson: generator kernel failed");
        cgh_dist_free(d);
        return nullptr;
    }
    if (h_lo) {
        cgh_dist_nb nb;
        nb.rank = rank - 1;
        nb.send_cnt = nxy;
        nb.contig_start = 0;
        nb.recv_off = 0;
        nb.recv_cnt = nxy;
        d->nbs.push_back(nb);
    }
    if (h_hi) {
        cgh_dist_nb nb;
        nb.rank = rank + 1;
        nb.send_cnt = nxy;
        nb.contig_start = nl - nxy;
        nb.recv_off = h_lo;
        nb.recv_cnt = nxy;
        d->nbs.push_back(nb);
    }
    auto_connect_halo(d);
    return d;
}

void cgh_dist_free(cgh_dist *d)
{
    if (!d) return;
    free_solver_state(d);
    if (d->ev_p) (void)hipEventDestroy(d->ev_p);
    if (d->ev_halo) (void)hipEventDestroy(d->ev_halo);
    for (cgh_dist_nb &nb : d->nbs) {
        cgh_free(nb.d_send_idx);
        cgh_free(nb.d_sendbuf);
    }
    cgh_free(d->mat.diag);
    cgh_free(d->mat.vals);
    cgh_free(d->mat.cols);
    cgh_ell_dict_free(&d->mat);
    halo_release(d);
    delete d;
}

int64_t cgh_dist_rows(const cgh_dist *d) { return d ? d->n_loc : 0; }
int64_t cgh_dist_nnz_offdiag(const cgh_dist *d) { return d ? d->nnz_off : 0; }

// Debug / test access: the local ELL (slot-major, local column ids) and the neighbour table.
int cgh_dist_download_local(const cgh_dist *d, double *diag, double *ell_vals, int *ell_cols)
{
    int rc = cgh_d2h(diag, d->mat.diag, sizeof(double) * (size_t)d->n_loc);
    if (rc) return rc;
    rc = cgh_d2h(ell_vals, d->mat.vals, sizeof(double) * (size_t)d->mat.stored);
    if (rc) return rc;
    return cgh_d2h(ell_cols, d->mat.cols, sizeof(int) * (size_t)d->mat.stored);
}
int cgh_dist_width(const cgh_dist *d) { return d->mat.K; }
// test/debug access to the vectors of the solve in flight (device pointers, owned rows first): 0 = r, 1 = p, 2 = A p
const double *cgh_dist_vector(const cgh_dist *d, int which)
{
    if (!d) return nullptr;
    return which == 0 ? d->r : which == 1 ? d->p : which == 2 ? d->Ap : nullptr;
}
int64_t cgh_dist_halo(const cgh_dist *d) { return d->n_halo; }

// Bit-identical mode for the solves started from now on (every rank makes the same choice): trajectory, x and the
// converging iteration equal the reference's on the undistributed system, whatever the number of ranks.
int cgh_dist_set_strict(cgh_dist *d, int strict)
{
    if (!d) return CGH_EINVAL;
    d->strict = strict != 0;
    return CGH_OK;
}

// ------------------------------------------------------------------------------------------ RCCL-driven loop
int cgh_dist_start(cgh_dist *d, const double *d_b, double *d_x, double tol, int capacity)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || !d_b || !d_x || capacity < 1) return CGH_EINVAL;
    const bool no_library = g_rccl.p2p && (d->halo_p2p || d->nbs.empty());   // nothing left for RCCL to do
    if ((!g_rccl.comm && !no_library) || g_rccl.nranks != d->nranks || g_rccl.rank != d->rank) {
        cgh_set_error("cgh_dist_start: no communicator for rank %d of %d (call cgh_comm_init first)", d->rank, d->nranks);
        return CGH_ECOMM;
    }
    int rc = alloc_solver_state(d, capacity);
    if (rc) return rc;
    hipStream_t   s = g_cgh.stream;
    const int64_t nl = d->n_loc;
    d->x = d_x;
    d->tol = tol;
    d->queued = 0;
    CGH_HIP(hipMemsetAsync(d->done, 0, sizeof(int) * 4, s));
    CGH_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)nl, s));
    CGH_HIP(hipMemsetAsync(d->p, 0, sizeof(double) * (size_t)(nl + d->n_halo), s));  // x0 = 0, halo of x0 = 0
    d->epoch = ++g_rccl.epoch;   // every rank starts its solves in the same order: the same number everywhere
    d->pushed_iter = 0;
    // The SpMV reads halo entries straight from the deposit (uncached, tagged words) when each is wanted about once --
    // the 7-point stencil.  With the 27-point stencil every halo entry is wanted by 9 rows: there one pass that unpacks
    // the deposit into the halo tail of p, then cached gathers, is far cheaper (inner rank of 8 of 512^3, us per
    // iteration: 1376 direct, 1069 unpacked, 1041 without neighbours).
    if (d->halo_refs < 0 && d->n_halo > 0) {
        unsigned long long *cnt = (unsigned long long *)cgh_alloc(sizeof(unsigned long long));
        unsigned long long  h = 0;
        if (!cnt || cgh_memset(cnt, 0, sizeof h) != CGH_OK) {
            cgh_free(cnt);
            return CGH_ENOMEM;
        }
        hipLaunchKernelGGL(count_halo_refs, dim3(grid_for(d->mat.stored)), dim3(256), 0, s, d->mat.stored, (const int *)d->mat.cols, (int)nl, cnt);
        rc = cgh_d2h(&h, cnt, sizeof h);
        cgh_free(cnt);
        if (rc) return rc;
        d->halo_refs = (int64_t)h;
    }
    d->ll_spmv = cgh_spmv_ell_halo_capable(&d->mat, d->p, d->Ap, 0, d->int_begin, d->int_end, nl) && d->halo_refs <= 2 * d->n_halo;
    if (const char *env = getenv("CGH_DIST_LL_SPMV")) d->ll_spmv = atoi(env) != 0 && cgh_spmv_ell_halo_capable(&d->mat, d->p, d->Ap, 0, d->int_begin, d->int_end, nl);
    int cnt = 0;
    rc = spmv_ranges(d, true, true, false, &cnt, s, nullptr);  // Ap = A*0, as the reference computes it (cg.c:441)
    if (rc) return rc;
    hipLaunchKernelGGL(cg_init_rp, dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d_b, (const double *)d->Ap, d->r, d->p, d->partB);
    if (d->strict) {
        d->xs = cgh_xsum_create(nl);
        d->chain = (double *)cgh_alloc(sizeof(double) * 8);
        if (!d->xs || !d->chain) return CGH_ENOMEM;
        CGH_HIP(hipMemsetAsync(d->chain, 0, sizeof(double) * 8, s));
        // Composed sums need the mailboxes and every rank's part within 32 batches of 64 groups (16.7M rows): the ranks
        // agree (a host-level exchange, once per solve).  CGH_DIST_STRICT_COMPOSED=0 keeps the rank-by-rank chain of round 2.
        d->composed = false;
        if (g_rccl.p2p && !(getenv("CGH_DIST_STRICT_COMPOSED") && atoi(getenv("CGH_DIST_STRICT_COMPOSED")) == 0)) {
            const int nb = cgh_xsum_composed_batches(d->xs);
            double    ok[1] = {nb >= 1 && nb <= CGH_XC_MAX_BATCHES ? 1.0 : 0.0};
            if (d->nranks > 1) {
                CGH_HIP(hipStreamSynchronize(s));
                rc = cgh_p2p_allreduce_f64(ok, 1);
                if (rc) return rc;
            } else {
                ok[0] = ok[0] * 1.0;
            }
            d->composed = ok[0] == (double)d->nranks;
        }
        return strict_chain(d, d->r, d->r, d->r, 3, 0, 1, false, s);  // (rounds N .. 2N-1: iteration 1 goes on at 2N)
    }
    if (g_rccl.p2p) {
        rc = cgh_p2p_launch(2, d->partB, d->vgrid, d->gs + 1, s);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(reduce_local<2>, dim3(2), dim3(CGH_BLOCK), 0, s, (const double *)d->partB, d->vgrid, d->gs_local + 1);
        CGH_HIP(hipGetLastError());
        CGH_NCCL(g_rccl.AllReduce(d->gs_local + 1, d->gs + 1, 2, ncclDouble, ncclSum, g_rccl.comm, s));
    }
    hipLaunchKernelGGL(finish_init_dist, dim3(1), dim3(1), 0, s, (const double *)d->gs, d->hist);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_dist_iterate(cgh_dist *d, int count, int time_kernels)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || count < 0 || d->queued + count > d->capacity) {
        cgh_set_error("cgh_dist_iterate: count exceeds the capacity given to cgh_dist_start");
        return CGH_EINVAL;
    }
    hipStream_t   s = g_cgh.stream, c = g_cgh.comm_stream;
    const int64_t nl = d->n_loc;
    static const bool one_spmv = !(getenv("CGH_DIST_ONE_SPMV") && atoi(getenv("CGH_DIST_ONE_SPMV")) == 0);
    for (int k = 0; k < count; ++k) {
        const int  iter = ++d->queued;
        const bool halo = !d->nbs.empty(), direct = halo && g_rccl.p2p && d->halo_p2p, folded = g_rccl.p2p;
        int        rc;
        if (halo && !direct) {
            // RCCL halo exchange of p on the communication stream, behind the kernel that last wrote p
            CGH_HIP(hipEventRecord(d->ev_p, s));
            CGH_HIP(hipStreamWaitEvent(c, d->ev_p, 0));
            rc = exchange_rccl(d, c);
            if (rc) return rc;
            CGH_HIP(hipEventRecord(d->ev_halo, c));
        }
        if (direct && d->pushed_iter != iter) halo_push_launch(d, iter, s);  // (else dist_update_xp of iter-1 pushed it)
        if (time_kernels) {
            hipEvent_t a, b;
            CGH_HIP(hipEventCreate(&a));
            CGH_HIP(hipEventCreate(&b));
            d->kev.push_back(a);
            d->kev.push_back(b);
            CGH_HIP(hipEventRecord(a, s));
        }
        int cnt = 0;
        if (direct) {
            const cgh_ll_halo hl = halo_view(d, iter);
            const int64_t     a = d->int_begin, b = d->int_end;
            int               g = 0;
            if (d->ll_spmv && one_spmv) {
                // ONE launch over all rows (1.8 us less than two at 2M rows), walked so that the rows with halo columns
                // come last: interior [a,b) and upper boundary [b,nl) first, lower boundary [0,a) at the end
                rc = cgh_launch_spmv_ell_halo(&d->mat, d->p, d->Ap, d->partA, &g, d->done, s, a, nl, 0, a, &hl);
                if (rc) return rc;
                cnt = g;
            } else {
                rc = spmv_ranges(d, true, false, true, &cnt, s, d->done);  // interior rows: the neighbours' pushes arrive meanwhile
                if (rc) return rc;
                if (d->ll_spmv) {
                    rc = cgh_launch_spmv_ell_halo(&d->mat, d->p, d->Ap, d->partA + cnt, &g, d->done, s, 0, a, b, nl, &hl);
                    if (rc) return rc;
                    cnt += g;
                } else {  // this matrix' SpMV cannot read the deposit: unpack it into the halo tail of p first
                    const int64_t blocks = (d->n_halo + 255) / 256;
                    hipLaunchKernelGGL(halo_pull_ll, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, s, hl, d->n_halo, d->p + nl,
                                       (const int *)d->done);
                    rc = spmv_ranges(d, false, true, true, &cnt, s, d->done);
                    if (rc) return rc;
                }
            }
        } else {
            rc = spmv_ranges(d, true, false, true, &cnt, s, d->done);   // interior rows: overlap the exchange
            if (rc) return rc;
            if (halo) CGH_HIP(hipStreamWaitEvent(s, d->ev_halo, 0));
            rc = spmv_ranges(d, false, true, true, &cnt, s, d->done);   // boundary rows
            if (rc) return rc;
        }
        if (time_kernels) CGH_HIP(hipEventRecord(d->kev.back(), s));
        if (d->strict) {
            // alpha, beta and the residual come from the chained order-exact sums; the vector kernels are the single-GPU
            // strict ones (scalars from the history), the next halo leaves with the stand-alone push of the next iteration
            rc = strict_chain(d, d->p, d->Ap, nullptr, 1, iter, 0, false, s);
            if (rc) return rc;
            hipLaunchKernelGGL((cg_update_r<1, 3>), dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->r, (const double *)d->Ap,
                               (const double *)nullptr, 0, d->partB, d->hist, iter, (const int *)d->done, 0, (const double *)nullptr,
                               cgh_xsum_cellsums(d->xs), cgh_xsum_cells(d->xs));
            rc = strict_chain(d, d->r, d->r, d->r, 2, iter, 1, true, s);
            if (rc) return rc;
            hipLaunchKernelGGL((cg_update_xp<1, 3>), dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->x, (const double *)d->r, d->p,
                               (const double *)nullptr, 0, d->hist, iter, d->tol, d->done, 0, (const double *)nullptr);
            CGH_HIP(hipGetLastError());
            continue;
        }
        if (folded) {  // the all-reduces happen inside the kernels that consume them
            cgh_link L;
            rc = cgh_p2p_link(&L);
            if (rc) return rc;
            L.self = d->loopback ? 1 : 0;
            L.tag = cgh_ll_tag(d->epoch, 2u * (unsigned)iter);
            const bool cached = cgh_fits_llc(&d->mat, 2.0);  // hints off when the caches can hold the vectors between kernels
            if (cached)
                hipLaunchKernelGGL(dist_update_r<0>, dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->r, (const double *)d->Ap,
                                   (const double *)d->partA, cnt, d->partB, d->hist, iter, (const int *)d->done, L);
            else
                hipLaunchKernelGGL(dist_update_r<3>, dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->r, (const double *)d->Ap,
                                   (const double *)d->partA, cnt, d->partB, d->hist, iter, (const int *)d->done, L);
            L.tag = cgh_ll_tag(d->epoch, 2u * (unsigned)iter + 1u);
            cgh_pushes P;
            memset(&P, 0, sizeof P);
            if (direct && d->fused_push) {  // exchange iter+1 leaves with the p it is made of
                P.n = (int)d->h_push.size();
                P.tag = cgh_ll_tag(d->epoch, (unsigned)(iter + 1));
                for (int j = 0; j < P.n; ++j) {
                    P.start[j] = d->h_push[j].start;
                    P.cnt[j] = d->h_push[j].cnt;
                    P.dst[j] = d->h_push[j].dst + (size_t)((iter + 1) & 1) * 2 * (size_t)d->h_push[j].plane;
                    P.plane[j] = d->h_push[j].plane;
                    const int64_t chunk = (P.start[j] / 2) / CGH_BLOCK;
                    if (chunk > P.rot) P.rot = chunk;  // start the sweep at the send range highest in memory
                }
                if (P.rot >= (nl / 2 + CGH_BLOCK - 1) / CGH_BLOCK) P.rot = 0;
                d->pushed_iter = iter + 1;
            }
            if (cached)
                hipLaunchKernelGGL(dist_update_xp<0>, dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->x, (const double *)d->r, d->p,
                                   (const double *)d->partB, d->vgrid, d->hist, iter, d->tol, d->done, L, P);
            else
                hipLaunchKernelGGL(dist_update_xp<3>, dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->x, (const double *)d->r, d->p,
                                   (const double *)d->partB, d->vgrid, d->hist, iter, d->tol, d->done, L, P);
        } else {
            hipLaunchKernelGGL(reduce_local<1>, dim3(1), dim3(CGH_BLOCK), 0, s, (const double *)d->partA, cnt, d->gs_local);
            CGH_NCCL(g_rccl.AllReduce(d->gs_local, d->gs, 1, ncclDouble, ncclSum, g_rccl.comm, s));
            hipLaunchKernelGGL((cg_update_r<2, 3>), dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->r, (const double *)d->Ap,
                               (const double *)nullptr, 0, d->partB, d->hist, iter, (const int *)d->done, 0, (const double *)d->gs,
                               (double *)nullptr, (int64_t)0);
            hipLaunchKernelGGL(reduce_local<2>, dim3(2), dim3(CGH_BLOCK), 0, s, (const double *)d->partB, d->vgrid, d->gs_local + 1);
            CGH_NCCL(g_rccl.AllReduce(d->gs_local + 1, d->gs + 1, 2, ncclDouble, ncclSum, g_rccl.comm, s));
            hipLaunchKernelGGL((cg_update_xp<2, 3>), dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, nl, d->x, (const double *)d->r, d->p,
                               (const double *)nullptr, 0, d->hist, iter, d->tol, d->done, 0, (const double *)d->gs);
        }
        CGH_HIP(hipGetLastError());
    }
    return CGH_OK;
}

// Launcher diagnostic: `reps` all-reduces of two doubles back to back on the compute stream, through the transport named
// (1 = the peer-to-peer mailboxes of p2p.hip as a kernel of its own, 0 = ncclAllReduce), timed with HIP events around the
// whole loop: microseconds per exchange as the slowest-to-arrive rank sees them.  Collective: every rank calls it with
// the same arguments.
int cgh_comm_bench_allreduce(int reps, int use_p2p, double *us_per_exchange)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (reps < 1 || !us_per_exchange) return CGH_EINVAL;
    if (use_p2p ? !cgh_p2p_connected() : (g_rccl.comm == nullptr)) {
        cgh_set_error("cgh_comm_bench_allreduce: that transport is not up");
        return CGH_ECOMM;
    }
    hipStream_t s = g_cgh.stream;
    double     *buf = (double *)cgh_alloc(sizeof(double) * 8);
    if (!buf) return CGH_ENOMEM;
    hipEvent_t a = nullptr, b = nullptr;
    int        rc = CGH_OK;
    const double v[2] = {1.0, -0.5};
    if (cgh_h2d(buf, v, sizeof v) != CGH_OK || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) rc = CGH_EHIP;
    for (int k = -8; rc == CGH_OK && k < reps; ++k) {  // (8 untimed exchanges first)
        if (k == 0 && hipEventRecord(a, s) != hipSuccess) rc = CGH_EHIP;
        if (use_p2p) rc = rc ? rc : cgh_p2p_launch(2, buf, 1, buf + 2, s);
        else if (g_rccl.AllReduce(buf, buf + 2, 2, ncclDouble, ncclSum, g_rccl.comm, s) != ncclSuccess) rc = CGH_ECOMM;
    }
    float ms = 0.f;
    if (rc == CGH_OK && (hipEventRecord(b, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess))
        rc = CGH_EHIP;
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    cgh_free(buf);
    if (rc == CGH_OK) *us_per_exchange = (double)ms * 1e3 / reps;
    return rc;
}

int cgh_dist_poll(cgh_dist *d, int *done_iter, int *queued)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d) return CGH_EINVAL;
    hipStream_t s = g_cgh.stream;
    // every rank derives the flag from the same reduced scalars; the max-reduce makes agreement explicit, so that
    // all ranks stop queueing (collectives!) at the same poll
    if (g_rccl.comm && d->nranks > 1) CGH_NCCL(g_rccl.AllReduce(d->done, d->done + 3, 1, ncclInt32, ncclMax, g_rccl.comm, s));
    int v[4] = {0, 0, 0, 0};  // [0] this rank's flag, [1],[2] iteration counters (cg_kernels.h), [3] max over ranks
    CGH_HIP(hipMemcpyAsync(v, d->done, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    CGH_HIP(hipStreamSynchronize(s));
    CGH_HIP(hipStreamSynchronize(g_cgh.comm_stream));
    if (g_rccl.p2p && d->halo_p2p && d->d_herr) {
        int he = 0;
        if (cgh_d2h(&he, d->d_herr, sizeof he) != CGH_OK || he) {
            cgh_set_error("cgh_dist_poll: a neighbour's halo did not arrive (timeout in the peer-to-peer exchange)");
            return CGH_ECOMM;
        }
    }
    if (g_rccl.p2p && cgh_p2p_error()) {
        cgh_set_error("cgh_dist_poll: a peer's reduction message did not arrive (timeout in the peer-to-peer all-reduce)");
        return CGH_ECOMM;
    }
    if (done_iter) *done_iter = (g_rccl.comm && d->nranks > 1) ? v[3] : v[0];
    if (queued) *queued = d->queued;
    return CGH_OK;
}

int cgh_dist_history(cgh_dist *d, int first, int count, double *h_out)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || first < 0 || count < 0 || first + count > d->capacity + 1) return CGH_EINVAL;
    return cgh_d2h(h_out, d->hist + (size_t)first * HIST, sizeof(double) * HIST * (size_t)count);
}

int cgh_dist_spmv_ms(cgh_dist *d, double *ms_avg)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!d || !ms_avg) return CGH_EINVAL;
    CGH_HIP(hipStreamSynchronize(g_cgh.stream));
    double tot = 0.0;
    for (size_t k = 0; k + 1 < d->kev.size(); k += 2) {
        float t = 0.f;
        CGH_HIP(hipEventElapsedTime(&t, d->kev[k], d->kev[k + 1]));
        tot += t;
    }
    *ms_avg = d->kev.empty() ? 0.0 : tot / (double)(d->kev.size() / 2);
    return CGH_OK;
}

// ------------------------------------------------------------------------------------------ single-process rank group
static cgh_dist_nb *find_nb(cgh_dist *d, int rank)
{
    for (cgh_dist_nb &nb : d->nbs)
        if (nb.rank == rank) return &nb;
    return nullptr;
}

static int group_exchange(cgh_dist **R, int G, hipStream_t s)
{
    for (int g = 0; g < G; ++g)
        for (cgh_dist_nb &nb : R[g]->nbs) {
            if (nb.recv_cnt == 0) continue;
            cgh_dist    *q = R[nb.rank];
            cgh_dist_nb *back = find_nb(q, g);
            if (!back || back->send_cnt != nb.recv_cnt) {
                cgh_set_error("group exchange: rank %d expects %lld entries from rank %d, which sends %lld", g, (long long)nb.recv_cnt,
                              nb.rank, back ? (long long)back->send_cnt : -1LL);
                return CGH_EINVAL;
            }
            double *dst = R[g]->p + R[g]->n_loc + nb.recv_off;
            if (back->contig_start >= 0)
                CGH_HIP(hipMemcpyAsync(dst, q->p + back->contig_start, sizeof(double) * (size_t)nb.recv_cnt, hipMemcpyDeviceToDevice, s));
            else
                hipLaunchKernelGGL(gather_rows, dim3(grid_for(nb.recv_cnt)), dim3(256), 0, s, nb.recv_cnt, (const int *)back->d_send_idx,
                                   (const double *)q->p, dst);
        }
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_dist_group_start(cgh_dist **R, int G, const double *const *d_b, double *const *d_x, double tol, int capacity)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!R || G < 1 || G > 64) return CGH_EINVAL;
    hipStream_t s = g_cgh.stream;
    std::vector<double *> locals((size_t)G), outs((size_t)G);
    for (int g = 0; g < G; ++g) {
        int rc = alloc_solver_state(R[g], capacity);
        if (rc) return rc;
        locals[(size_t)g] = R[g]->gs_local;
        outs[(size_t)g] = R[g]->gs;
    }
    cgh_dist *d0 = R[0];
    d0->d_locals = (double **)cgh_alloc(sizeof(double *) * (size_t)G);
    d0->d_outs = (double **)cgh_alloc(sizeof(double *) * (size_t)G);
    if (!d0->d_locals || !d0->d_outs) return CGH_ENOMEM;
    int rc = cgh_h2d(d0->d_locals, locals.data(), sizeof(double *) * (size_t)G);
    if (rc == CGH_OK) rc = cgh_h2d(d0->d_outs, outs.data(), sizeof(double *) * (size_t)G);
    if (rc) return rc;
    for (int g = 0; g < G; ++g) {
        cgh_dist *d = R[g];
        d->x = d_x[g];
        d->tol = tol;
        d->queued = 0;
        CGH_HIP(hipMemsetAsync(d->done, 0, sizeof(int) * 4, s));
        CGH_HIP(hipMemsetAsync(d->x, 0, sizeof(double) * (size_t)d->n_loc, s));
        CGH_HIP(hipMemsetAsync(d->p, 0, sizeof(double) * (size_t)(d->n_loc + d->n_halo), s));
        int cnt = 0;
        rc = spmv_ranges(d, true, true, false, &cnt, s, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(cg_init_rp, dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, d->n_loc, d_b[g], (const double *)d->Ap, d->r, d->p, d->partB);
        hipLaunchKernelGGL(reduce_local<2>, dim3(2), dim3(CGH_BLOCK), 0, s, (const double *)d->partB, d->vgrid, d->gs_local + 1);
    }
    hipLaunchKernelGGL(group_allreduce, dim3(1), dim3(64), 0, s, G, (double *const *)d0->d_locals, (double *const *)d0->d_outs, 1, 2);
    for (int g = 0; g < G; ++g) hipLaunchKernelGGL(finish_init_dist, dim3(1), dim3(1), 0, s, (const double *)R[g]->gs, R[g]->hist);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

int cgh_dist_group_iterate(cgh_dist **R, int G, int count)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!R || G < 1 || !R[0]->d_locals) return CGH_EINVAL;
    hipStream_t s = g_cgh.stream;
    cgh_dist   *d0 = R[0];
    for (int k = 0; k < count; ++k) {
        std::vector<int> cnt((size_t)G, 0);
        // interior rows first (they never read the halo), then the exchange, then the boundary rows: the same order of
        // dependencies as the two-stream RCCL loop, serialised on one stream
        for (int g = 0; g < G; ++g) {
            if (R[g]->queued + 1 > R[g]->capacity) return CGH_EINVAL;
            int rc = spmv_ranges(R[g], true, false, true, &cnt[(size_t)g], s, R[g]->done);
            if (rc) return rc;
        }
        int rc = group_exchange(R, G, s);
        if (rc) return rc;
        for (int g = 0; g < G; ++g) {
            rc = spmv_ranges(R[g], false, true, true, &cnt[(size_t)g], s, R[g]->done);
            if (rc) return rc;
            hipLaunchKernelGGL(reduce_local<1>, dim3(1), dim3(CGH_BLOCK), 0, s, (const double *)R[g]->partA, cnt[(size_t)g], R[g]->gs_local);
        }
        hipLaunchKernelGGL(group_allreduce, dim3(1), dim3(64), 0, s, G, (double *const *)d0->d_locals, (double *const *)d0->d_outs, 0, 1);
        for (int g = 0; g < G; ++g) {
            cgh_dist *d = R[g];
            const int iter = ++d->queued;
            hipLaunchKernelGGL((cg_update_r<2, 3>), dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, d->n_loc, d->r, (const double *)d->Ap,
                               (const double *)nullptr, 0, d->partB, d->hist, iter, (const int *)d->done, 0, (const double *)d->gs);
            hipLaunchKernelGGL(reduce_local<2>, dim3(2), dim3(CGH_BLOCK), 0, s, (const double *)d->partB, d->vgrid, d->gs_local + 1);
        }
        hipLaunchKernelGGL(group_allreduce, dim3(1), dim3(64), 0, s, G, (double *const *)d0->d_locals, (double *const *)d0->d_outs, 1, 2);
        for (int g = 0; g < G; ++g) {
            cgh_dist *d = R[g];
            hipLaunchKernelGGL((cg_update_xp<2, 3>), dim3(d->vgrid), dim3(CGH_BLOCK), 0, s, d->n_loc, d->x, (const double *)d->r, d->p,
                               (const double *)nullptr, 0, d->hist, d->queued, d->tol, d->done, 0, (const double *)d->gs);
        }
        CGH_HIP(hipGetLastError());
    }
    return CGH_OK;
}

} // extern "C"
