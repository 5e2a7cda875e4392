// device_utils.h -- wave64 / workgroup reduction helpers and exact-rounding arithmetic.
#pragma once

#include <hip/hip_runtime.h>

// Rounded multiply then rounded add, never contracted into an FMA: the
// reference is built without FMA contraction (its Makefile: -std=c99), and its
// results are reproduced bit for bit only if each product is rounded first.
// The translation units are also compiled with -ffp-contract=off.
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double madd_rn(double acc, double a, double b) { return __dadd_rn(acc, __dmul_rn(a, b)); }

// Sum over the 64 lanes of a wave (result valid in lane 0; fixed order).
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// The same with DPP moves only (no LDS crossbar round trips): prefix inside each row of 16 lanes, then the row totals
// handed on (row_bcast:15, row_bcast:31).  Result valid in lane 63.  For the kernels that reduce once per 128 rows
// (the per-cell sums that predict the binades of the order-exact sums, xsum.hip): any order will do there.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v)
{
    v += dpp_mov_f64<0x111, 0xf>(v);  // row_shr:1 (lanes without a source lane add +0.0)
    v += dpp_mov_f64<0x112, 0xf>(v);
    v += dpp_mov_f64<0x114, 0xf>(v);
    v += dpp_mov_f64<0x118, 0xf>(v);
    v += dpp_mov_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_mov_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Sum over a 256-thread workgroup; result valid in thread 0.  `lds` holds >= 4 doubles.
// The order (lane tree, then waves 0..3 left to right) is fixed, so the result is
// run-to-run deterministic for a fixed launch shape.
__device__ __forceinline__ double block_sum_256(double v, double *lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0) total = ((lds[0] + lds[1]) + lds[2]) + lds[3];
    __syncthreads();
    return total;
}

// Every thread of a 256-thread workgroup gets the sum of parts[0..count) (count <= 2048),
// accumulated in a fixed order that is the same in every workgroup, so all workgroups of
// a launch derive bit-identical scalars from the same partials.
__device__ __forceinline__ double block_sum_partials(const double *parts, int count, int stride, double *lds)
{
    // The loads are issued eight at a time before any is consumed (a plain strided loop serialises ~8 L2 round trips,
    // which is most of the run time of the small kernels that call this); the additions keep the order k = tid,
    // tid+256, ... so the result does not depend on the batching.
    double v = 0.0;
    for (int base = threadIdx.x; base < count; base += 8 * 256) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = base + u * 256;
            t[u] = k < count ? parts[(size_t)k * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (base + u * 256 < count) v += t[u];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double total = ((lds[0] + lds[1]) + lds[2]) + lds[3];
    __syncthreads();
    return total;
}

// Counter-based hash -> double in [0,1): splitmix64 finaliser of (seed ^ idx*golden).  Integer-only, so the
// host twin (tests/foamio.py hash01) gives the same bits.
__device__ __host__ inline double cgh_hash01(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed ^ (idx * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

