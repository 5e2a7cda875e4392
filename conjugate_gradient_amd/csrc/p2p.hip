// p2p.hip -- the two scalar all-reduces of a multi-GPU CG iteration as ONE small kernel each, over xGMI peer stores.
// New module, no reference counterpart.
//
// The all-reduces carry 8-16 bytes: they are pure latency, and a library collective costs a kernel launch plus
// 2(N-1) ring hops on top of the reduction kernel that feeds it.  xGMI is point-to-point, every GPU reaches every
// other one directly, so for a message this small an all-to-all is one hop:
//     p2p_allreduce<W>   (one workgroup per rank)
//       1. sum this rank's partials (same order as reduce_local of dist.hip)
//       2. store the W doubles into the mailbox of every rank, itself included -- each 64-bit word carries 32 bits of
//          data and the 32-bit sequence number of the exchange, so a word is valid exactly when its tag matches
//          and no fence or flag ordering is needed (the "LL" encoding of NCCL's low-latency protocol)
//       3. poll its own mailbox until the N messages of this exchange are there, add them in rank order 0..N-1
//          (identical bits on every rank), write the result where the consumer kernel expects it
// Mailboxes are fine-grained device memory shared through hipIpc handles; slots rotate over 4 sequence numbers (a rank
// cannot be more than one exchange ahead of the slowest: it needs everybody's message to finish its own).
// A poll gives up after ~10 s, poisons the result with NaN and raises an error word the host checks at its next poll.
//
// cgh_comm_init (dist.hip) connects the ranks, runs a self-test against known sums and lets all ranks agree (RCCL
// min-reduce) whether to use this path; on any failure the RCCL all-reduces stay.  CGH_P2P=0 disables it.
#include "cgh_internal.h"
#include "device_utils.h"
#include "halo_ll.h"

namespace {

constexpr int P2P_MAX_RANKS = CGH_LINK_MAX_RANKS, P2P_SLOTS = CGH_LINK_SLOTS, P2P_WORDS = CGH_LINK_WORDS;
// region 0: the exchanges of this file (numbered by g_p2p.seq); region 1: those folded into the kernels of the
// iteration (halo_ll.h, numbered by solve and iteration)
constexpr size_t P2P_BOX_WORDS = 3 * CGH_LINK_REGION_WORDS + CGH_XC_REGION_WORDS;  // host-numbered | per iteration | rounds of the strict chain | composed sums

struct p2p_state {
    bool      connected = false;
    int       nranks = 0, rank = -1;
    uint64_t *box = nullptr;                       // this rank's mailbox (device, fine-grained if available)
    void     *opened[P2P_MAX_RANKS] = {nullptr};   // hipIpcOpenMemHandle results (to close), NULL for self
    uint64_t **d_peers = nullptr;                  // device array [nranks]: every rank's mailbox as seen from here
    int       *d_err = nullptr;                    // device: 1 after a timed-out poll
    unsigned long long *d_rounds = nullptr;        // device: rounds of the strict chain executed so far (dist_kernels.h, chain_round)
    unsigned   seq = 0;                            // exchanges so far (same on every rank)
} g_p2p;

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(int W, const double *__restrict__ partials, int count,
                                                           uint64_t *const *__restrict__ peers, const uint64_t *box, int n, int me,
                                                           unsigned seq, double *__restrict__ out, int *err)
{
    __shared__ double   lds[4];
    __shared__ double   mine[2];
    __shared__ unsigned halves[P2P_MAX_RANKS * P2P_WORDS];
    for (int w = 0; w < W; ++w) {
        const double v = block_sum_partials(partials + w, count, W, lds);
        if (threadIdx.x == 0) mine[w] = v;
    }
    __syncthreads();
    const int  slot = (int)(seq % P2P_SLOTS), words = 2 * W, t = threadIdx.x;
    const bool broken = *err != 0;  // an earlier exchange timed out: keep sending (the peers may still be fine), never wait again
    if (t < n * words) {
        const int      peer = t / words, wi = t % words;
        const uint64_t bits = (uint64_t)__double_as_longlong(mine[wi >> 1]);
        const uint64_t half = (wi & 1) ? bits >> 32 : bits & 0xffffffffull;
        // 1. to `peer`: my message's word wi
        __hip_atomic_store(peers[peer] + ((size_t)(slot * P2P_MAX_RANKS + me) * P2P_WORDS + wi), ((uint64_t)seq << 32) | half,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // 2. from `peer` (same index arithmetic, my own mailbox): its message's word wi
        const uint64_t *q = box + ((size_t)(slot * P2P_MAX_RANKS + peer) * P2P_WORDS + wi);
        const long long t0 = wall_clock64();
        uint64_t        v = 0;
        while (!broken) {
            v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(v >> 32) == seq) break;
            if (wall_clock64() - t0 > 1000000000ll) {  // 10 s at the 100 MHz constant clock
                atomicExch(err, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        halves[peer * P2P_WORDS + wi] = (unsigned)v;
    }
    __syncthreads();
    if (t < W) {
        double s = 0.0;
        for (int r = 0; r < n; ++r) {
            const uint64_t bits = ((uint64_t)halves[r * P2P_WORDS + 2 * t + 1] << 32) | halves[r * P2P_WORDS + 2 * t];
            s += __longlong_as_double((long long)bits);
        }
        out[t] = (broken || *err) ? NAN : s;
    }
}

void p2p_release()
{
    for (int r = 0; r < P2P_MAX_RANKS; ++r)
        if (g_p2p.opened[r]) {
            (void)hipIpcCloseMemHandle(g_p2p.opened[r]);
            g_p2p.opened[r] = nullptr;
        }
    cgh_free(g_p2p.d_peers);
    cgh_free(g_p2p.d_err);
    cgh_free(g_p2p.d_rounds);
    g_p2p.d_rounds = nullptr;
    g_p2p.d_peers = nullptr;
    g_p2p.d_err = nullptr;
    g_p2p.connected = false;
}

} // namespace

extern "C" {

// This rank's mailbox as a 64-byte IPC handle (allocates the mailbox on first use).
int cgh_p2p_local_handle(char out_64_bytes[64])
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!g_p2p.box) {
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, P2P_BOX_WORDS * sizeof(uint64_t), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            CGH_HIP(hipMalloc(&p, P2P_BOX_WORDS * sizeof(uint64_t)));
        }
        g_p2p.box = (uint64_t *)p;
        CGH_HIP(hipMemset(g_p2p.box, 0, P2P_BOX_WORDS * sizeof(uint64_t)));
        CGH_HIP(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    CGH_HIP(hipIpcGetMemHandle(&h, g_p2p.box));
    memcpy(out_64_bytes, &h, 64);
    return CGH_OK;
}

// handles: nranks x 64 bytes (rank order), as returned by cgh_p2p_local_handle on every rank.  All ranks must have
// zeroed mailboxes (fresh from cgh_p2p_local_handle) before anyone sends: callers put a barrier after this call.
int cgh_p2p_connect(int nranks, int rank, const char *handles)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (nranks < 1 || nranks > P2P_MAX_RANKS || rank < 0 || rank >= nranks || !handles || !g_p2p.box) {
        cgh_set_error("cgh_p2p_connect: bad arguments (ranks %d, rank %d; at most %d ranks)", nranks, rank, P2P_MAX_RANKS);
        return CGH_EINVAL;
    }
    p2p_release();
    // tags of an earlier connection must not be mistaken for this one's (every rank does this before the barrier that
    // follows the call, so nobody is sending yet)
    CGH_HIP(hipMemset(g_p2p.box, 0, P2P_BOX_WORDS * sizeof(uint64_t)));
    CGH_HIP(hipDeviceSynchronize());
    uint64_t *peers[P2P_MAX_RANKS] = {nullptr};
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) {
            peers[r] = g_p2p.box;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, handles + 64 * (size_t)r, 64);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            cgh_set_error("cgh_p2p_connect: hipIpcOpenMemHandle(rank %d) -> %s", r, hipGetErrorString(e));
            p2p_release();
            return CGH_ECOMM;
        }
        g_p2p.opened[r] = p;
        peers[r] = (uint64_t *)p;
    }
    g_p2p.d_peers = (uint64_t **)cgh_alloc(sizeof(uint64_t *) * P2P_MAX_RANKS);
    g_p2p.d_err = (int *)cgh_alloc(sizeof(int));
    g_p2p.d_rounds = (unsigned long long *)cgh_alloc(sizeof(unsigned long long));
    if (!g_p2p.d_peers || !g_p2p.d_err || !g_p2p.d_rounds || cgh_memset(g_p2p.d_rounds, 0, sizeof(unsigned long long)) != CGH_OK || cgh_h2d(g_p2p.d_peers, peers, sizeof peers) != CGH_OK ||
        cgh_memset(g_p2p.d_err, 0, sizeof(int)) != CGH_OK) {
        p2p_release();
        return CGH_ENOMEM;
    }
    g_p2p.nranks = nranks;
    g_p2p.rank = rank;
    g_p2p.seq = 0;
    g_p2p.connected = true;
    return CGH_OK;
}

void cgh_p2p_disconnect(void)
{
    p2p_release();
    if (g_p2p.box) (void)hipFree(g_p2p.box);
    g_p2p.box = nullptr;
}

int cgh_p2p_connected(void) { return g_p2p.connected ? 1 : 0; }

// 1 if a poll of this process ever timed out (the results since then are NaN).  Synchronises the compute stream.
int cgh_p2p_error(void)
{
    if (!g_p2p.connected) return 0;
    int e = 0;
    if (cgh_d2h(&e, g_p2p.d_err, sizeof e) != CGH_OK) return 1;
    return e;
}

// Test / launcher helper: all-reduce (sum, rank order) of 1 or 2 host doubles over the connected ranks.
int cgh_p2p_allreduce_f64(double *h_vals, int count)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!g_p2p.connected || !h_vals || count < 1 || count > 2) return CGH_EINVAL;
    double *buf = (double *)cgh_alloc(sizeof(double) * 4);
    if (!buf) return CGH_ENOMEM;
    int rc = cgh_h2d(buf, h_vals, sizeof(double) * (size_t)count);
    if (rc == CGH_OK) rc = cgh_p2p_launch(count, buf, 1, buf + 2, g_cgh.stream);
    if (rc == CGH_OK) rc = cgh_d2h(h_vals, buf + 2, sizeof(double) * (size_t)count);
    cgh_free(buf);
    if (rc == CGH_OK && cgh_p2p_error()) {
        cgh_set_error("cgh_p2p_allreduce_f64: a peer's message did not arrive (timeout)");
        rc = CGH_ECOMM;
    }
    return rc;
}

} // extern "C"

// The mailboxes as the kernels of the iteration see them (dist.hip fills in tag and self).
int cgh_p2p_link(cgh_link *L)
{
    if (!g_p2p.connected) return CGH_EINVAL;
    L->peers = (uint64_t *const *)g_p2p.d_peers;
    L->box = g_p2p.box;
    L->err = g_p2p.d_err;
    L->nranks = g_p2p.nranks;
    L->me = g_p2p.rank;
    L->self = 0;
    L->tag = 0;
    L->region = 1;
    L->rounds = g_p2p.d_rounds;
    return CGH_OK;
}

// out[w] = sum over ranks of (sum of this rank's W-interleaved partials, `count` per scalar), w < W <= 2.
int cgh_p2p_launch(int W, const double *partials, int count, double *out, hipStream_t s)
{
    if (!g_p2p.connected || W < 1 || W > 2) return CGH_EINVAL;
    if (++g_p2p.seq == 0) ++g_p2p.seq;  // tag 0 is the value of an empty mailbox
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(256), 0, s, W, partials, count, (uint64_t *const *)g_p2p.d_peers,
                       (const uint64_t *)g_p2p.box, g_p2p.nranks, g_p2p.rank, g_p2p.seq, out, g_p2p.d_err);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}
