// spmv_csrb.hip -- column-blocked CSR: y = A x for matrices whose rows are skewed AND whose columns are scattered
// (BASELINE config 5: 8.4M rows, row lengths 1..2000, columns uniformly random).
//
// No reference counterpart as a format: the reference's SpMV for such a matrix is mv_csr_symmetric
// (src/csr.c:190-229), a serial scatter.  On the GPU the plain row-wise gather (spmv_csr.hip) is bound by x: every
// x[col] is a 64-byte fabric request for 8 useful bytes (x = 67 MB does not fit an XCD's 4 MiB L2), 4.3x the
// algorithmic traffic and 10 % of the HBM roofline.  This layout makes the gathers hit L2:
//
//   * rows are cut into 64-row SEGMENTS; segments are dealt to WAVE-BLOCKS of RPW rows (RPW/64 segments each) so that
//     every block holds about the same number of entries (heaviest segment first to the lightest block with a free
//     slot: cgh_csrb_plan_host).  One wave owns one block; its RPW row sums live in LDS.
//   * a block's entries are sorted by (column slab, local row), a slab being 2^cb_bits consecutive columns (1 MiB of
//     x by default), and streamed in that order.  All blocks hold the same number of entries and all waves stream at
//     about the same rate, so at any moment every wave on the chip gathers from the same one or two slabs of x: the
//     XCD's L2 serves them, x is fetched from the Infinity Cache once per XCD and sweep instead of once per entry.
//   * an entry is 12 bytes, as in CSR: the value and one packed word (local row << cb_bits | column inside the slab);
//     the slab itself follows from the entry's position (per-block slab offsets, nslabs+1 ints per block).
//   * inside a 64-entry chunk, entries of one (slab, row) are adjacent: a segmented scan sums them, the last lane of
//     each run adds the run's sum to the row's LDS word (read-add-write: the wave owns the word, lanes hit distinct
//     words), slab by slab where a chunk spans more than one.  The order of additions into a row is therefore FIXED
//     (slab by slab, tree inside a run): results are deterministic run to run, but not the reference's row order --
//     this is a throughput layout; parity callers use CGH_LAYOUT_CSR with exact rows.
#include <algorithm>
#include <queue>
#include <exception>
#include <vector>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "cgh_internal.h"
#include "device_utils.h"

cgh_config g_cfg;

namespace {

constexpr int SEG = 64;            // rows per segment (one lane per row in the epilogue)
constexpr int MAX_SLABS = 256;     // slab cursor fits the 8 bits above the 16-bit local row in the scan key
constexpr int GATE_ROUNDS = 4;     // rounds of the persistent grid that keep progress counters (later rounds run ungated)
constexpr int GATE_COUNTERS = 8 * GATE_ROUNDS * MAX_SLABS;  // one counter per (group, round, slab)
constexpr int GATE_SPINS = 600;    // looks (~0.5 us each) before a wave stops waiting for the others for good
constexpr int GATE_WORDS = GATE_COUNTERS + 8 * 64;  // counters + one 256-byte line per workgroup group (blockIdx % 8)

struct dbuf {
    void *p = nullptr;
    explicit dbuf(size_t bytes) { p = cgh_alloc(bytes ? bytes : 16); }
    ~dbuf() { cgh_free(p); }
    dbuf(const dbuf &) = delete;
    dbuf &operator=(const dbuf &) = delete;
    template <typename T> T *as() { return (T *)p; }
};

inline int grid_for(int64_t work) { return cgh_stream_grid((work + 255) / 256, 8); }

__global__ __launch_bounds__(256) void segment_weights(int64_t n, int64_t nseg, const int64_t *__restrict__ row_ptr, int *__restrict__ w)
{
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * 256) {
        const int64_t r0 = s * SEG, r1 = r0 + SEG < n ? r0 + SEG : n;
        w[s] = (int)(row_ptr[r1] - row_ptr[r0]);
    }
}

// key of entry e of row r: ((block * nslabs + slab) << 16) | local row; one wave per row keeps the loads coalesced
__global__ __launch_bounds__(256) void entry_keys(int64_t n, int nslabs, int cb, const int64_t *__restrict__ row_ptr,
                                                 const int *__restrict__ cols, const int *__restrict__ seg_block,
                                                 const int *__restrict__ seg_slot, unsigned long long *__restrict__ keys,
                                                 unsigned *__restrict__ idx)
{
    const int     lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        const int64_t            sg = r / SEG;
        const unsigned long long blk = (unsigned long long)seg_block[sg];
        const unsigned long long lr = (unsigned long long)(seg_slot[sg] * SEG + (int)(r % SEG));
        const int64_t            e1 = row_ptr[r + 1];
        for (int64_t e = row_ptr[r] + lane; e < e1; e += 64) {
            const unsigned long long slab = (unsigned long long)(cols[e] >> cb);
            keys[e] = ((blk * (unsigned long long)nslabs + slab) << 16) | lr;
            idx[e] = (unsigned)e;
        }
    }
}

__global__ __launch_bounds__(256) void gather_sorted(int64_t nnz, int cb, const unsigned long long *__restrict__ keys,
                                                    const unsigned *__restrict__ idx, const double *__restrict__ vals,
                                                    const int *__restrict__ cols, double *__restrict__ svals,
                                                    unsigned *__restrict__ packed)
{
    const unsigned cmask = (1u << cb) - 1u;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * 256) {
        const unsigned e = idx[q];
        svals[q] = vals[e];
        packed[q] = ((unsigned)(keys[q] & 0xffffu) << cb) | ((unsigned)cols[e] & cmask);
    }
}

// start[t] = first sorted position whose tile id (key >> 16) is >= t, for t = 0..ntiles (start[ntiles] = nnz)
__global__ __launch_bounds__(256) void tile_starts(int64_t nnz, int64_t ntiles, const unsigned long long *__restrict__ keys,
                                                  int64_t *__restrict__ start)
{
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q <= nnz; q += (int64_t)gridDim.x * 256) {
        const int64_t lo = q == 0 ? 0 : (int64_t)(keys[q - 1] >> 16) + 1;
        const int64_t hi = q == nnz ? ntiles : (int64_t)(keys[q] >> 16);
        for (int64_t t = lo; t <= hi; ++t) start[t] = q;
    }
}

__global__ __launch_bounds__(256) void block_offsets(int64_t nblocks, int nslabs, const int64_t *__restrict__ start,
                                                    int64_t *__restrict__ blk_base, int *__restrict__ tile_off)
{
    const int64_t total = nblocks * (int64_t)(nslabs + 1);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / (nslabs + 1);
        const int     t = (int)(i % (nslabs + 1));
        const int64_t base = start[b * nslabs];
        tile_off[i] = (int)(start[b * nslabs + t] - base);
        if (t == 0) blk_base[b] = base;
        if (t == nslabs && b == nblocks - 1) blk_base[nblocks] = start[b * nslabs + t];
    }
}

// ------------------------------------------------------------------------------------------------ the kernel
//
// Pacing (speed only, never correctness).  Left alone, the waves drift 5-10 slabs apart during a sweep
// (tools/csrb_drift.py) and half the gathers miss L2.  So the workgroups of one GROUP (blockIdx % 8: observed to share
// an XCD, hence an L2) keep together:
//   * the 4 waves of a workgroup count themselves per slab in LDS; the last one to leave slab j adds 1 to the group's
//     counter of (round, j) with a RETURNING atomic, looks at the answer one chunk later (its latency is never waited
//     for), and if its add was the one that reached the threshold (all but 1/fdiv of the group's workgroups) it stores
//     the new FRONT -- "slab j is behind us" -- into the group's line: a plain store, read by the others past L1
//     (an L2 hit if the group does share an XCD);
//   * a wave gathers from slab j only once the front has passed slab j - window.  It reads the front one chunk ahead
//     too; only a wave that must wait spins on the line (and every 8th look reads the counter itself, in case the
//     line is not shared after all).
// Measured on the way here: 4096 waves adding to ONE word per slab, or polling one word: 155 us per slab (same-address
// atomics and loads at the memory side run at ~90 per microsecond); consumers polling the counters on demand: every
// gate a 2-3 us round trip.  A wrong guess about the XCD, a stale line, a lost or reordered store cost time only.
// Every wait is bounded; a wave that ran out of patience once stops waiting for good.
// DPP moves (VALU, no LDS): lanes without a source keep `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// one step of the segmented scan: add the value of the lane the DPP control names if it belongs to the same run
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(double &pr, unsigned key)
{
    const unsigned kk = dpp_u32<CTRL, ROW_MASK>(0xfffffffeu, key);
    const unsigned lo = dpp_u32<CTRL, ROW_MASK>(0u, (unsigned)__double2loint(pr));
    const unsigned hi = dpp_u32<CTRL, ROW_MASK>(0u, (unsigned)__double2hiint(pr));
    if (kk == key) pr += __hiloint2double((int)hi, (int)lo);
}

struct csrb_gate {
    unsigned *cnt;     // this group's counters of this round: [MAX_SLABS]
    unsigned *line;    // this group's cached front (word 0)
    unsigned  thresh;  // adds that open a gate
    int       round, window, waves_act;
};

template <int RPW, bool DOT, bool STAMP = false, bool LDSADD = false>
__global__ __launch_bounds__(CGH_BLOCK) void spmv_csrb(int64_t n, int64_t nblocks, int nslabs, int cb,
                                                      const int *__restrict__ blk_seg, const int64_t *__restrict__ blk_base,
                                                      const int *__restrict__ tile_off, const double *__restrict__ vals,
                                                      const unsigned *__restrict__ packed, const double *__restrict__ diag,
                                                      const double *__restrict__ x, double *__restrict__ y,
                                                      double *__restrict__ partials, const int *__restrict__ done,
                                                      unsigned *__restrict__ prog, int wparam,
                                                      unsigned long long *__restrict__ stamps = nullptr)
{
    constexpr int SPB = RPW / SEG;
    __shared__ double acc[4 * RPW];
    __shared__ int    tps[4][MAX_SLABS + 4];
    __shared__ int    wgc[MAX_SLABS];
    __shared__ double red[4];
    if (done != nullptr && *done != 0) return;
    const int      lane = threadIdx.x & 63;
    const int      wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double        *my = acc + wave * RPW;
    int           *tp = tps[wave];
    const unsigned cmask = (1u << cb) - 1u;
    double         local = 0.0;
    for (int64_t g = blockIdx.x; g * 4 < nblocks; g += gridDim.x) {
        const int64_t b = g * 4 + wave;  // wave-uniform
        if (prog != nullptr) {           // the workgroup's per-slab wave counts of this round
            __syncthreads();
            for (int t = threadIdx.x; t < MAX_SLABS; t += CGH_BLOCK) wgc[t] = 0;
            __syncthreads();
        }
        if (b < nblocks) {
#pragma unroll
            for (int k = 0; k < SPB; ++k) my[k * SEG + lane] = 0.0;
            for (int t = lane; t <= nslabs; t += 64) tp[t] = tile_off[b * (nslabs + 1) + t];
            // the wave owns these LDS words; lanes read what other lanes of the SAME wave wrote (in-order LDS):
            // the fences only keep the compiler from moving the accesses
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t base = blk_base[b];
            const int     cnt = (int)(blk_base[b + 1] - base);
            int           mark = -1, nlooks = 0, ngates = 0;  // STAMP: diagnostic build only (tools/csrb_drift.py)

            csrb_gate G;
            G.round = (int)((g - blockIdx.x) / gridDim.x);
            const bool gate_on = prog != nullptr && G.round < GATE_ROUNDS;
            {
                const int     grp = blockIdx.x & 7;
                const int64_t wgs_left = (nblocks + 3) / 4 - (int64_t)G.round * gridDim.x;
                const int     wgs_act = (int)(wgs_left < (int64_t)gridDim.x ? wgs_left : (int64_t)gridDim.x);
                const int     in_grp = wgs_act > grp ? (wgs_act - grp + 7) / 8 : 0;
                const int     fdiv = (wparam >> 8) ? (wparam >> 8) : 2;
                G.window = wparam & 255;
                G.thresh = (unsigned)(in_grp - in_grp / fdiv);
                G.waves_act = nblocks - g * 4 < 4 ? (int)(nblocks - g * 4) : 4;
                G.cnt = prog + ((size_t)grp * GATE_ROUNDS + (size_t)(gate_on ? G.round : 0)) * MAX_SLABS;
                G.line = prog + GATE_COUNTERS + grp * 64;
            }
            bool     gating = gate_on;
            int      cleared = G.window - 1;  // slabs <= cleared may be gathered from
            int      wdone = 0;               // slabs < wdone have been reported as left behind
            unsigned f_ahead = 0;             // lane 0: the front as loaded during the previous chunk
            unsigned rep_old = 0;             // lane k: what the returning add for slab rep_base + k returned
            bool     rep_mine = false;        // lane k: this lane made such an add
            int      rep_base = 0;

            // Software pipeline, one chunk deep: the gather of chunk i+1 (and the stream loads of chunk i+2) are in
            // flight while chunk i is summed into LDS.  Loads complete in issue order, so each wait leaves the younger
            // ones outstanding.
            int      s_n = 0, tnext_n = tp[1];  // slab cursor of the chunk whose gather is being issued
            unsigned pk_c = 0xffffffffu, pk_n = 0xffffffffu, pk_nn = 0xffffffffu;
            double   v_c = 0.0, v_n = 0.0, v_nn = 0.0, gx_c = 0.0, gx_n = 0.0;
            int      s_c = 0;
            bool     act_c = false, act_n = false;
            if (lane < cnt) {
                pk_n = __builtin_nontemporal_load(packed + base + lane);
                v_n = __builtin_nontemporal_load(vals + base + lane);
            }
            if (lane + 64 < cnt) {
                pk_nn = __builtin_nontemporal_load(packed + base + lane + 64);
                v_nn = __builtin_nontemporal_load(vals + base + lane + 64);
            }
            for (int q0 = -64; q0 < cnt; q0 += 64) {
                // ---- stage A, chunk at q0 + 64: slab cursor, pacing, gather
                const int qa = q0 + 64;
                act_n = qa + lane < cnt;
                if (qa < cnt) {
                    if (act_n)
                        while (qa + lane >= tnext_n) tnext_n = tp[++s_n + 1];  // tp[nslabs] = cnt > q: terminates
                    if constexpr (STAMP) {
                        const int s8 = __builtin_amdgcn_readfirstlane(s_n) >> 3;
                        if (s8 != mark) {
                            mark = s8;
                            if (lane == 0 && s8 < 32) stamps[b * 33 + s8] = __builtin_amdgcn_s_memrealtime();
                        }
                    }
                    const int s_lo = __builtin_amdgcn_readfirstlane(s_n);
                    if (gate_on) {
                        // (a) the adds of the previous chunk have returned by now: did one of them open a gate?
                        const unsigned long long hit = __ballot(rep_mine && rep_old + 1u == G.thresh);
                        unsigned                 f_old = __builtin_amdgcn_readfirstlane(f_ahead);
                        if (hit) {
                            const unsigned code = (unsigned)(G.round * 256 + rep_base + (63 - __clzll((long long)hit)) + 1);
                            if (code > f_old) {
                                if (lane == 0) __hip_atomic_store(G.line, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                f_old = code;
                            }
                        }
                        rep_mine = false;
                        // (b) the gate, on the front as it was one chunk ago; a wave that must wait spins on the line
                        const int nact = cnt - qa < 64 ? cnt - qa : 64;
                        const int s_hi = __builtin_amdgcn_readlane(s_n, nact - 1);
                        if (gating && s_hi > cleared) {
                            const int      need_slab = s_hi - G.window;
                            const unsigned need = (unsigned)(G.round * 256 + need_slab + 1);
                            unsigned       f = f_old;
                            int            spins = 0;
                            while (f < need) {
                                if (spins >= GATE_SPINS) {
                                    gating = false;
                                    break;
                                }
                                if ((spins & 7) == 7) {  // the counter itself, in case the line is not shared after all
                                    unsigned c = 0;
                                    if (lane == 0) c = __hip_atomic_load(G.cnt + need_slab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if (__builtin_amdgcn_readfirstlane(c) >= G.thresh) {
                                        f = need;
                                        break;
                                    }
                                }
                                __builtin_amdgcn_s_sleep(8);
                                unsigned t = 0;
                                if (lane == 0) t = __hip_atomic_load(G.line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                f = __builtin_amdgcn_readfirstlane(t);
                                ++spins;
                            }
                            if (f >= need) cleared = (int)(f >> 8) > G.round ? nslabs : (int)(f & 255u) - 1 + G.window;
                            if constexpr (STAMP) {
                                nlooks += spins;
                                ngates += 1;
                            }
                        }
                        if (gating && lane == 0) f_ahead = __hip_atomic_load(G.line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    gx_n = 0.0;
                    if (act_n) gx_n = x[((int64_t)s_n << cb) + (int64_t)(pk_n & cmask)];
                    // (c) slabs this wave has left behind: counted per workgroup in LDS, the last wave tells the group
                    if (gate_on && s_lo > wdone) {
                        const int j = wdone + lane;  // at most 64 slabs per report; the rest follows with the next chunk
                        const int upto = s_lo - wdone > 64 ? wdone + 64 : s_lo;
                        if (j < upto && atomicAdd(&wgc[j], 1) == G.waves_act - 1) {
                            rep_old = __hip_atomic_fetch_add(G.cnt + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            rep_mine = true;
                        }
                        rep_base = wdone;
                        wdone = upto;
                    }
                }
                // the stream loads two chunks ahead go out behind the gather
                unsigned  pk_l = 0xffffffffu;
                double    v_l = 0.0;
                const int ql = q0 + 192 + lane;
                if (ql < cnt) {
                    pk_l = __builtin_nontemporal_load(packed + base + ql);
                    v_l = __builtin_nontemporal_load(vals + base + ql);
                }
                // ---- stage B, chunk at q0: products, runs of one (slab, row), read-add-write into the row's LDS word
                if (q0 >= 0) {
                    const unsigned lr = pk_c >> cb;
                    double         pr = v_c * gx_c;
                    if constexpr (LDSADD) {
                        // one LDS atomic per lane; lanes of one instruction that meet on a row are applied by the
                        // hardware one after the other (observed in lane order: bit-identical run to run)
                        if (act_c) unsafeAtomicAdd(&my[lr], pr);
                    } else {
                        const unsigned key = act_c ? (((unsigned)s_c << 16) | lr) : 0xffffffffu;
                        const unsigned kprev = dpp_u32<0x138>(0xfffffffeu, key);  // wave_shr:1
                        if (__ballot(act_c && kprev == key)) {
                            // segmented inclusive scan (keys are sorted, so equal ends mean one run): 4 steps inside the
                            // rows of 16 lanes, then the last lane of row 0 / 2 into row 1 / 3, then lane 31 into rows 2, 3
                            seg_step<0x111, 0xf>(pr, key);
                            seg_step<0x112, 0xf>(pr, key);
                            seg_step<0x114, 0xf>(pr, key);
                            seg_step<0x118, 0xf>(pr, key);
                            seg_step<0x142, 0xa>(pr, key);
                            seg_step<0x143, 0xc>(pr, key);
                        }
                        const unsigned knext = dpp_u32<0x130>(0xfffffffeu, key);  // wave_shl:1
                        bool           pending = act_c && knext != key;            // last lane of its run
                        unsigned long long m = __ballot(pending);
                        while (m) {
                            const int cs = __builtin_amdgcn_readlane(s_c, __ffsll((long long)m) - 1);
                            if (pending && s_c == cs) {  // one slab at a time: the rows of these lanes are distinct
                                my[lr] += pr;
                                pending = false;
                            }
                            m = __ballot(pending);
                        }
                    }
                }
                pk_c = pk_n;
                v_c = v_n;
                gx_c = gx_n;
                s_c = s_n;
                act_c = act_n;
                pk_n = pk_nn;
                v_n = v_nn;
                pk_nn = pk_l;
                v_nn = v_l;
            }
            if (gate_on) {
                // the rest of the slabs are behind this wave now; whoever completes a count may still open a gate
                for (int j0 = wdone; j0 < nslabs; j0 += 64) {
                    const int j = j0 + lane;
                    if (j < nslabs && atomicAdd(&wgc[j], 1) == G.waves_act - 1) {
                        const unsigned old = __hip_atomic_fetch_add(G.cnt + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old + 1u == G.thresh && j == nslabs - 1)
                            __hip_atomic_store(G.line, (unsigned)(G.round * 256 + nslabs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if constexpr (STAMP)
                if (lane == 0) {
                    stamps[b * 33 + 32] = __builtin_amdgcn_s_memrealtime();
                    stamps[nblocks * 33 + b] = ((unsigned long long)nlooks << 32) | ((unsigned long long)ngates << 1) | (gate_on && !gating ? 1ull : 0ull);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
            for (int k = 0; k < SPB; ++k) {
                const int r0 = blk_seg[b * SPB + k];  // wave-uniform
                if (r0 < 0) continue;
                const int64_t r = (int64_t)r0 + lane;
                if (r < n) {
                    const double xr = x[r];
                    const double yv = add_rn(0.0, mul_rn(diag[r], xr)) + my[k * SEG + lane];
                    __builtin_nontemporal_store(yv, y + r);
                    if (DOT) local += xr * yv;
                }
            }
        }
    }
    if constexpr (DOT) {
        const double tot = block_sum_256(local, red);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

template <int RPW>
int lds_bytes()
{
    return 4 * RPW * 8 + 4 * (MAX_SLABS + 4) * 4 + MAX_SLABS * 4 + 32;
}

unsigned long long *g_stamps = nullptr;  // diagnostic: per-block time stamps of the slab sweep (cgh_csrb_debug_stamps)

template <int RPW, bool DOT, bool STAMP, bool LDSADD>
void launch_as(const cgh_matrix *m, const double *x, double *y, double *dot_partials, const int *done, hipStream_t s, int grid,
               unsigned long long *stamps)
{
    unsigned *prog = (m->gate_window & 255) > 0 ? m->gate_prog : nullptr;
    hipLaunchKernelGGL((spmv_csrb<RPW, DOT, STAMP, LDSADD>), dim3(grid), dim3(CGH_BLOCK), 0, s, m->n, m->nblocks, m->nslabs, m->cb_bits,
                       (const int *)m->blk_seg, (const int64_t *)m->blk_base, (const int *)m->tile_off, (const double *)m->vals,
                       (const unsigned *)m->cols, (const double *)m->diag, x, y, dot_partials, done, prog, m->gate_window & 0xffff, stamps);
}

template <int RPW>
void launch(const cgh_matrix *m, const double *x, double *y, double *dot_partials, const int *done, hipStream_t s, int grid)
{
    const bool ldsadd = (m->gate_window >> 16) & 1;
    if (g_stamps != nullptr && !dot_partials && RPW == 1024 && !ldsadd)
        launch_as<1024, false, true, false>(m, x, y, nullptr, done, s, grid, g_stamps);
    else if (dot_partials && ldsadd) launch_as<RPW, true, false, true>(m, x, y, dot_partials, done, s, grid, nullptr);
    else if (dot_partials) launch_as<RPW, true, false, false>(m, x, y, dot_partials, done, s, grid, nullptr);
    else if (ldsadd) launch_as<RPW, false, false, true>(m, x, y, nullptr, done, s, grid, nullptr);
    else launch_as<RPW, false, false, false>(m, x, y, nullptr, done, s, grid, nullptr);
}

} // namespace

int cgh_launch_spmv_csrb(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out, const int *done,
                         hipStream_t s)
{
    const int     lds = m->rpw == 512 ? lds_bytes<512>() : m->rpw == 2048 ? lds_bytes<2048>() : lds_bytes<1024>();
    int           per_cu = (160 * 1024) / lds;
    if (per_cu > 8) per_cu = 8;
    if (g_cfg.csrb_wgs_per_cu > 0 && g_cfg.csrb_wgs_per_cu < per_cu) per_cu = g_cfg.csrb_wgs_per_cu;
    const int64_t groups = (m->nblocks + 3) / 4;
    const int     grid = cgh_stream_grid(groups, per_cu);
    if (grid_out) *grid_out = grid;
    if ((m->gate_window & 255) > 0 && m->gate_prog)
        CGH_HIP(hipMemsetAsync(m->gate_prog, 0, sizeof(unsigned) * GATE_WORDS, s));
    if (m->rpw == 512) launch<512>(m, x, y, dot_partials, done, s, grid);
    else if (m->rpw == 2048) launch<2048>(m, x, y, dot_partials, done, s, grid);
    else launch<1024>(m, x, y, dot_partials, done, s, grid);
    CGH_HIP(hipGetLastError());
    return CGH_OK;
}

// Heaviest segment first, to the lightest block that still has a free slot (ties: lower block id; equal weights keep
// segment order).  Every block ends with at most spb segments; blocks differ by little more than one light segment.
extern "C" int64_t cgh_csrb_plan_host(int64_t nseg, const int *weights, int spb, int *seg_block, int *seg_slot)
{
    if (nseg <= 0 || spb <= 0 || !weights || !seg_block || !seg_slot) return 0;
    const int64_t        nb = (nseg + spb - 1) / spb;
    std::vector<int64_t> order((size_t)nseg);
    for (int64_t i = 0; i < nseg; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return weights[a] > weights[b]; });
    typedef std::pair<int64_t, int64_t> load_block;  // (entries so far, block id): smallest on top
    std::priority_queue<load_block, std::vector<load_block>, std::greater<load_block>> heap;
    std::vector<int>                                                                   used((size_t)nb, 0);
    for (int64_t b = 0; b < nb; ++b) heap.push(load_block(0, b));
    for (int64_t k = 0; k < nseg; ++k) {
        const int64_t    sg = order[(size_t)k];
        const load_block top = heap.top();
        heap.pop();
        const int64_t b = top.second;
        seg_block[sg] = (int)b;
        seg_slot[sg] = used[(size_t)b]++;
        if (used[(size_t)b] < spb) heap.push(load_block(top.first + weights[sg], b));
    }
    return nb;
}

// CSR (full rows, device) -> CSRB.  `csr` is consumed: its diag is adopted, its arrays are freed.
static cgh_matrix *cgh_csrb_from_csr_impl(cgh_matrix *csr)
{
    hipStream_t   s = g_cgh.stream;
    const int64_t n = csr->n, nnz = csr->nnz_off;
    const int     rpw = (g_cfg.csrb_rpw == 512 || g_cfg.csrb_rpw == 2048) ? g_cfg.csrb_rpw : 1024;
    int           cb = g_cfg.csrb_slab_bits;
    if (cb < 4) cb = 4;
    while (((n + (1ll << cb) - 1) >> cb) > MAX_SLABS) ++cb;
    int rb = 0;
    while ((1 << rb) < rpw) ++rb;
    if (cb + rb > 32 || nnz >= 0xffffffffll) {
        cgh_set_error("column-blocked CSR: %lld rows / %lld entries exceed the packed 32-bit (row, column) words (rpw=%d)",
                      (long long)n, (long long)nnz, rpw);
        cgh_matrix_free(csr);
        return nullptr;
    }
    const int     nslabs = (int)((n + (1ll << cb) - 1) >> cb);
    const int     spb = rpw / SEG;
    const int64_t nseg = (n + SEG - 1) / SEG;
    const int64_t nb = (nseg + spb - 1) / spb;
    const int64_t ntiles = nb * nslabs;
    const size_t  e = (size_t)(nnz ? nnz : 1);

    cgh_matrix *m = new cgh_matrix();
    m->n = n;
    m->K = csr->K;
    m->layout = CGH_LAYOUT_CSRB;
    m->nnz_off = nnz;
    m->stored = nnz;
    m->cb_bits = cb;
    m->nslabs = nslabs;
    m->rpw = rpw;
    m->nblocks = nb;
    bool ok = true;
    {
        dbuf             w(sizeof(int) * (size_t)nseg), sb(sizeof(int) * (size_t)nseg), ss(sizeof(int) * (size_t)nseg);
        std::vector<int> hw((size_t)nseg), hb((size_t)nseg), hs((size_t)nseg), hseg((size_t)(nb * spb), -1);
        ok = w.p && sb.p && ss.p;
        if (ok) {
            hipLaunchKernelGGL(segment_weights, dim3(grid_for(nseg)), dim3(256), 0, s, n, nseg, (const int64_t *)csr->row_ptr, w.as<int>());
            ok = cgh_d2h(hw.data(), w.p, sizeof(int) * (size_t)nseg) == CGH_OK;
        }
        if (ok) {
            cgh_csrb_plan_host(nseg, hw.data(), spb, hb.data(), hs.data());
            for (int64_t sg = 0; sg < nseg; ++sg) hseg[(size_t)(hb[(size_t)sg] * (int64_t)spb + hs[(size_t)sg])] = (int)(sg * SEG);
            m->blk_seg = (int *)cgh_alloc(sizeof(int) * (size_t)(nb * spb));
            ok = m->blk_seg && cgh_h2d(m->blk_seg, hseg.data(), sizeof(int) * (size_t)(nb * spb)) == CGH_OK &&
                 cgh_h2d(sb.p, hb.data(), sizeof(int) * (size_t)nseg) == CGH_OK && cgh_h2d(ss.p, hs.data(), sizeof(int) * (size_t)nseg) == CGH_OK;
        }
        dbuf k0(sizeof(unsigned long long) * e), k1(sizeof(unsigned long long) * e), i0(sizeof(unsigned) * e), i1(sizeof(unsigned) * e);
        ok = ok && k0.p && k1.p && i0.p && i1.p;
        if (ok && nnz > 0) {
            hipLaunchKernelGGL(entry_keys, dim3(g_cgh.num_cu * 8), dim3(256), 0, s, n, nslabs, cb, (const int64_t *)csr->row_ptr,
                               (const int *)csr->cols, (const int *)sb.p, (const int *)ss.p, k0.as<unsigned long long>(), i0.as<unsigned>());
            int bits = 16;
            while (bits < 64 && (ntiles >> (bits - 16)) != 0) ++bits;
            size_t     tmp_bytes = 0;
            hipError_t he = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0.as<unsigned long long>(), k1.as<unsigned long long>(),
                                                      i0.as<unsigned>(), i1.as<unsigned>(), (size_t)nnz, 0u, (unsigned)bits, s);
            dbuf       tmp(tmp_bytes);
            ok = he == hipSuccess && tmp.p;
            if (ok)
                he = rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k0.as<unsigned long long>(), k1.as<unsigned long long>(),
                                               i0.as<unsigned>(), i1.as<unsigned>(), (size_t)nnz, 0u, (unsigned)bits, s);
            ok = ok && he == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
        dbuf start(sizeof(int64_t) * (size_t)(ntiles + 1));
        if (ok) {
            m->vals = (double *)cgh_alloc(sizeof(double) * e);
            m->cols = (int *)cgh_alloc(sizeof(unsigned) * e);
            m->blk_base = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(nb + 1));
            m->tile_off = (int *)cgh_alloc(sizeof(int) * (size_t)(nb * (nslabs + 1)));
            m->gate_prog = (unsigned *)cgh_alloc(sizeof(unsigned) * GATE_WORDS);
            m->gate_window = (g_cfg.csrb_window & 255) | ((g_cfg.csrb_quorum_div & 255) << 8) | (g_cfg.csrb_lds_atomics ? 1 << 16 : 0);
            ok = m->vals && m->cols && m->blk_base && m->tile_off && start.p && m->gate_prog;
        }
        if (ok) {
            if (nnz > 0)
                hipLaunchKernelGGL(gather_sorted, dim3(grid_for(nnz)), dim3(256), 0, s, nnz, cb, (const unsigned long long *)k1.p,
                                   (const unsigned *)i1.p, (const double *)csr->vals, (const int *)csr->cols, m->vals, (unsigned *)m->cols);
            hipLaunchKernelGGL(tile_starts, dim3(grid_for(nnz + 1)), dim3(256), 0, s, nnz, ntiles, (const unsigned long long *)k1.p,
                               start.as<int64_t>());
            hipLaunchKernelGGL(block_offsets, dim3(grid_for(nb * (nslabs + 1))), dim3(256), 0, s, nb, nslabs, (const int64_t *)start.p,
                               m->blk_base, m->tile_off);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
    }
    if (ok) {
        m->diag = csr->diag;  // adopt
        csr->diag = nullptr;
    }
    cgh_matrix_free(csr);
    if (!ok) {
        if (!g_cgh.err[0]) cgh_set_error("column-blocked CSR: construction failed");
        cgh_matrix_free(m);
        return nullptr;
    }
    return m;
}

// (host containers inside: nothing may leave through the extern "C" callers as an exception)
cgh_matrix *cgh_csrb_from_csr(cgh_matrix *csr)
{
    try {
        return cgh_csrb_from_csr_impl(csr);
    } catch (const std::exception &e) {
        cgh_set_error("column-blocked CSR: %s", e.what());
        return nullptr;  // (csr may be lost with it: the process is out of host memory)
    }
}

// diagnostic build of the kernel (rows per wave-block 1024, no dot): block b writes the 100 MHz clock at which it enters
// slabs 0, 8, 16, ... (33 words per block, the last = end of its stream).  NULL switches it off.
extern "C" void cgh_csrb_debug_stamps(void *dev_words) { g_stamps = (unsigned long long *)dev_words; }

extern "C" int cgh_config_csrb(int rpw, int slab_bits, int wgs_per_cu, int window, int quorum_div, int lds_atomics)
{
    if ((rpw != 0 && rpw != 512 && rpw != 1024 && rpw != 2048) || (slab_bits != 0 && (slab_bits < 4 || slab_bits > 22)) || wgs_per_cu < 0 ||
        window > 64 || quorum_div > 255 || quorum_div == 1) {
        cgh_set_error("cgh_config_csrb: rows per wave-block 512|1024|2048, slab bits 4..22 (0 = keep), window <= 64, quorum divisor 2..255");
        return CGH_EINVAL;
    }
    if (rpw) cgh_config_set(&g_cfg.csrb_rpw, rpw);
    if (slab_bits) cgh_config_set(&g_cfg.csrb_slab_bits, slab_bits);
    cgh_config_set(&g_cfg.csrb_wgs_per_cu, wgs_per_cu);
    if (window >= 0) cgh_config_set(&g_cfg.csrb_window, window);
    if (quorum_div > 0) cgh_config_set(&g_cfg.csrb_quorum_div, quorum_div);
    if (lds_atomics >= 0) cgh_config_set(&g_cfg.csrb_lds_atomics, (int)(lds_atomics != 0));
    return CGH_OK;
}
