// halo_ll.h -- what the kernels of the multi-GPU iteration share: tagged 64-bit words over peer-mapped memory.
// New module, no reference counterpart.
//
// A double travels as TWO 64-bit words, each carrying 32 bits of data below the 32-bit tag of its exchange (the "LL"
// encoding of NCCL's low-latency protocol): a word is valid exactly when its tag matches, so neither side needs a fence,
// a flag or an atomic counter -- the writer stores with system scope (write-through), the reader polls the very words
// it needs.  Tags are derived from (solve number, iteration, which exchange of the iteration), never from a host-side
// launch count, so ranks may queue different numbers of frozen (post-convergence) iterations without losing step.
//
//   cgh_link      the two scalar all-reduces of an iteration, folded into the prologue of the kernel that consumes the
//                 sum (link_allreduce): workgroup 0 stores this rank's value into every peer's mailbox, every workgroup
//                 polls its own mailbox for the peers' values and adds the N values in rank order (same bits everywhere)
//   cgh_ll_halo   the deposit buffer a boundary SpMV reads halo entries of p from (ll_read), written by the neighbours'
//                 p-update kernels (ll_store_pair): per slot, the low words of all entries, then the high words
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int CGH_LINK_MAX_RANKS = 16, CGH_LINK_SLOTS = 4, CGH_LINK_WORDS = 4;
// mailbox: [region][slot][sender rank][word]; region 0 = exchanges numbered by a host counter (p2p.hip: start-up, tests),
// region 1 = exchanges numbered by iteration (this header), region 2 = rounds of the chained order-exact sums
constexpr size_t CGH_LINK_REGION_WORDS = (size_t)CGH_LINK_SLOTS * CGH_LINK_MAX_RANKS * CGH_LINK_WORDS;
constexpr long long CGH_LL_TIMEOUT_TICKS = 1000000000ll;  // 10 s of the 100 MHz constant clock
// region 3 = the composed order-exact sums of the bit-identical multi-GPU mode (xsum.hip, xc_*): per slot and sender
//   words [0, 4)                       the sender's approximate totals of the two sums (binade prediction of the ranks behind it)
//   word  4                            the number of batches of 64 groups its part has
//   words 8 + ((s 32 + b) 10 + w)      word w of the summary of batch b of sum s (five 64-bit words, low half first)
//   words 648 + (s 32 + b) 2 + {0, 1}  the exact running sum behind batch b of sum s, when the sender had to walk that batch
constexpr int    CGH_XC_MAX_BATCHES = 32, CGH_XC_WORDS = 800, CGH_XC_SUMMARY0 = 8, CGH_XC_RESULT0 = 648;
constexpr size_t CGH_XC_REGION0 = 3 * CGH_LINK_REGION_WORDS;
constexpr size_t CGH_XC_REGION_WORDS = (size_t)CGH_LINK_SLOTS * CGH_LINK_MAX_RANKS * CGH_XC_WORDS;

struct cgh_link {
    uint64_t *const *peers;  // [nranks]: every rank's mailbox as mapped here
    const uint64_t  *box;    // this rank's mailbox
    int             *err;    // device word, 1 after a timed-out wait (then nobody waits again)
    int              nranks, me;
    int              self;   // 1: this rank's own value goes through its mailbox too (one-rank timing harness)
    unsigned         tag;    // of this exchange; its slot is (tag & 3)
    unsigned         region; // of the mailbox: 1 = the two reductions of an iteration, 2 = rounds of the strict chain
    unsigned long long *rounds;  // device: strict-chain rounds executed so far on this rank (the same on every rank)
};

struct cgh_ll_halo {
    const uint64_t *dep;     // this exchange's slot of my deposit buffer: the low words of all halo entries (halo order),
    int64_t         plane;   // then, `plane` words further, their high words
    int64_t         nl;      // owned entries of p: column c >= nl is halo entry c - nl
    unsigned        tag;
    int            *err;
};

constexpr int CGH_PUSH_MAX = 4;
struct cgh_pushes {          // contiguous send ranges of p, pushed by the kernel that writes p
    int       n;
    unsigned  tag;
    int64_t   rot;                 // the kernel starts its sweep at this 512-row chunk (and wraps): the send range highest in
                                   // memory is then pushed in the first wave of workgroups too, not at the end of the kernel
    int64_t   start[CGH_PUSH_MAX], cnt[CGH_PUSH_MAX];
    uint64_t *dst[CGH_PUSH_MAX];   // the neighbour's deposit slot for this exchange, at my region (low words), as mapped here
    int64_t   plane[CGH_PUSH_MAX]; // words from there to the high words
};

// tag of exchange `idx` of solve `epoch`: never 0 (the value of a fresh buffer), low 2 bits = idx & 3
__host__ __device__ inline unsigned cgh_ll_tag(unsigned epoch, unsigned idx)
{
    return 0x80000000u | ((epoch & 0x7ffu) << 20) | (idx & 0xfffffu);
}

#ifdef __HIPCC__
typedef unsigned long long cgh_ull2 __attribute__((ext_vector_type(2)));

// One entry: low word to *lo, high word to *hi (system scope: written through, visible to a polling peer).
__device__ __forceinline__ void ll_store(uint64_t *lo, uint64_t *hi, double v, unsigned tag)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(v), t = (uint64_t)tag << 32;
    __hip_atomic_store(lo, t | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(hi, t | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Two consecutive entries, at words lo_word and hi_word (both even) of `base` (the same in every lane: a neighbour's
// deposit slot): two 16-byte stores with the cache bits of a system-scope store (sc0 sc1: written through), so that a
// wave writes whole, contiguous lines across xGMI.  Every aligned 64-bit word still arrives whole, which is all the tags
// need.  Through the raw-buffer builtin, NOT a volatile access: the compiler completes every volatile access on its own
// (s_waitcnt vmcnt(0) after each store -- a round trip over xGMI per store for the lanes that push, found in round 3),
// and a volatile cast also loses the address space (flat_ instead of global_ instructions).
typedef unsigned cgh_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ll_store_pair(uint64_t *base, int64_t lo_word, int64_t hi_word, double v0, double v1, unsigned tag)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    cgh_u4                       l, h;
    l.x = (unsigned)__double2loint(v0);
    l.y = tag;
    l.z = (unsigned)__double2loint(v1);
    l.w = tag;
    h.x = (unsigned)__double2hiint(v0);
    h.y = tag;
    h.z = (unsigned)__double2hiint(v1);
    h.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(l, rs, (unsigned)(lo_word * 8), 0, 17 /* sc0 sc1 */);
    __builtin_amdgcn_raw_buffer_store_b128(h, rs, (unsigned)(hi_word * 8), 0, 17);
}

// Wait for the words at lo and hi to carry `tag`; NaN (and *err = 1) after 10 s, or at once if an earlier wait timed out.
__device__ __forceinline__ double ll_wait(const uint64_t *lo_p, const uint64_t *hi_p, unsigned tag, int *err)
{
    uint64_t lo = __hip_atomic_load(lo_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint64_t hi = __hip_atomic_load(hi_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(lo >> 32) != tag || (unsigned)(hi >> 32) != tag) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return __builtin_nan("");
        const long long t0 = wall_clock64();
        while (true) {
            __builtin_amdgcn_s_sleep(2);
            lo = __hip_atomic_load(lo_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            hi = __hip_atomic_load(hi_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
            if (wall_clock64() - t0 > CGH_LL_TIMEOUT_TICKS) {
                atomicExch(err, 1);
                return __builtin_nan("");
            }
        }
    }
    return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
}

// one 32-bit datum under its tag
__device__ __forceinline__ void ll_store32(uint64_t *q, unsigned data, unsigned tag)
{
    __hip_atomic_store(q, ((uint64_t)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned ll_wait32(const uint64_t *q, unsigned tag, int *err, bool *failed)
{
    uint64_t v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(v >> 32) != tag) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            *failed = true;
            return 0;
        }
        const long long t0 = wall_clock64();
        while (true) {
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(v >> 32) == tag) break;
            if (wall_clock64() - t0 > CGH_LL_TIMEOUT_TICKS) {
                atomicExch(err, 1);
                *failed = true;
                return 0;
            }
        }
    }
    return (unsigned)v;
}

__device__ __forceinline__ double ll_read(const cgh_ll_halo &h, int64_t entry)
{
    return ll_wait(h.dep + entry, h.dep + h.plane + entry, h.tag, h.err);
}

// out[w] = sum over ranks (rank order) of mine[w], w < W <= 2.  Called by all 256 threads of every workgroup of the
// launch with the same `mine`; `ex` holds 2*16 doubles of LDS.  Every thread returns the same bits on every rank.
template <int W>
__device__ __forceinline__ void link_allreduce(const cgh_link &L, const double *mine, double *out, double *ex)
{
    const int t = threadIdx.x;
    if (L.nranks > 1 || L.self) {
        const int    slot = (int)(L.tag & (CGH_LINK_SLOTS - 1));
        const size_t base = (size_t)L.region * CGH_LINK_REGION_WORDS + (size_t)slot * CGH_LINK_MAX_RANKS * CGH_LINK_WORDS;
        if (t < L.nranks * W) {
            const int peer = t / W, w = t % W;
            if (peer == L.me && !L.self) {
                ex[t] = mine[w];
            } else {
                if (blockIdx.x == 0) {
                    uint64_t *q = L.peers[peer] + base + (size_t)L.me * CGH_LINK_WORDS + 2 * w;
                    ll_store(q, q + 1, mine[w], L.tag);
                }
                const uint64_t *q = L.box + base + (size_t)peer * CGH_LINK_WORDS + 2 * w;
                ex[t] = ll_wait(q, q + 1, L.tag, L.err);
            }
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < W; ++w) {
            double s = 0.0;
            for (int r = 0; r < L.nranks; ++r) s += ex[r * W + w];
            out[w] = s;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int w = 0; w < W; ++w) out[w] = 0.0 + mine[w];
    }
}
#endif
