// matrix.hip -- cgh_matrix lifetime, SpMV dispatch and the benchmark timing helpers.
#include <vector>

#include "cgh_internal.h"

int cgh_launch_spmv_ell(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                        const int *done, hipStream_t s, const cgh_spmv_opts *opts);
int cgh_launch_spmv_csr(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                        const int *done, hipStream_t s);
int cgh_launch_spmv_sell(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                         const int *done, hipStream_t s);
int cgh_launch_spmv_csrb(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                         const int *done, hipStream_t s);
int cgh_launch_spmv_pb(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                       const int *done, hipStream_t s);
int cgh_csr_to_ell_download(const cgh_matrix *m, double *ell_vals, int *ell_cols);

int cgh_launch_spmv(const cgh_matrix *m, const double *x, double *y, double *dot_partials, int *grid_out,
                    const int *done, hipStream_t s, const cgh_spmv_opts *opts)
{
    if (opts && opts->cell_sums_written) *opts->cell_sums_written = false;  // only the ELL kernels can leave them
    switch (m->layout) {
    case CGH_LAYOUT_ELL: return cgh_launch_spmv_ell(m, x, y, dot_partials, grid_out, done, s, opts);
    case CGH_LAYOUT_CSR: return cgh_launch_spmv_csr(m, x, y, dot_partials, grid_out, done, s);
    case CGH_LAYOUT_SELL:
    case CGH_LAYOUT_SELLS: return cgh_launch_spmv_sell(m, x, y, dot_partials, grid_out, done, s);
    case CGH_LAYOUT_CSRB: return cgh_launch_spmv_csrb(m, x, y, dot_partials, grid_out, done, s);
    case CGH_LAYOUT_PB: return cgh_launch_spmv_pb(m, x, y, dot_partials, grid_out, done, s);
    }
    cgh_set_error("cgh_spmv: unknown layout %d", m->layout);
    return CGH_EINVAL;
}

namespace {

typedef unsigned u4_t __attribute__((ext_vector_type(4)));

// The device-to-device copy that serves as the attainable-HBM yardstick of the bench line: UN independent 16-byte loads
// in flight per lane (a workgroup moves tiles of 256 x UN x 16 bytes, consecutive lanes on consecutive 16 bytes), loads
// and stores non-temporal or not.  (Round 2's yardstick had ONE load in flight per lane and temporal stores: 4.84 TB/s,
// 23 % below what the part does on a copy.)
template <int UN, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy16(const u4_t *__restrict__ src, u4_t *__restrict__ dst, size_t count)
{
    const size_t tile = (size_t)256 * UN, tiles = count / tile;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
        u4_t         v[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) v[q] = NTL ? __builtin_nontemporal_load(src + i + q * 256) : src[i + q * 256];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            if (NTS) __builtin_nontemporal_store(v[q], dst + i + q * 256);
            else dst[i + q * 256] = v[q];
        }
    }
    for (size_t i = tiles * tile + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

struct timing_events {
    hipEvent_t a = nullptr, b = nullptr;
    ~timing_events()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

constexpr int CGH_READ_WGS_PER_CU = 3;  // of the read ceiling: 2-3 per CU 7.13-7.16 TB/s, 5 or 7 7.0, 6 6.55, but 4 5.73 and 8 (rounds 1-2) 5.90 -- power-of-two strides between the workgroups alias in the memory channels (profiles/r03_read_sweep.txt)
// read-only stream: 4 independent 16-byte non-temporal loads in flight per lane, one word per workgroup written
__global__ __launch_bounds__(256) void read16(const u4_t *__restrict__ src, unsigned *__restrict__ sink, size_t count)
{
    unsigned     acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t       i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < count; i += 4 * stride) {
        const u4_t a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                   c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < count; i += stride) {
        const u4_t a = __builtin_nontemporal_load(src + i);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9e3779b9u) sink[blockIdx.x] = acc;  // (practically never: keeps the loads alive)
}

// the same with 8-byte and 4-byte loads per lane (the access widths of the CSR-family kernels): calibrates FETCH_SIZE
__global__ __launch_bounds__(256) void read8_4(const unsigned long long *__restrict__ src8, const unsigned *__restrict__ src4,
                                               unsigned *__restrict__ sink, size_t count)
{
    unsigned long long acc = 0;
    const size_t       stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride)
        acc ^= __builtin_nontemporal_load(src8 + i) ^ (unsigned long long)__builtin_nontemporal_load(src4 + i);
    if (acc == 0x9e3779b97f4a7c15ull) sink[blockIdx.x] = (unsigned)acc;
}

} // namespace

extern "C" {

// `count` 8-byte words from d_src8 and `count` 4-byte words from d_src4, one of each per lane and step (12 bytes per lane,
// as the CSR-family kernels stream a matrix): a calibration stream for the FETCH_SIZE counter.
int cgh_time_read_8_4(const void *d_src8, const void *d_src4, void *d_sink_8k, size_t count, int reps, double *ms_avg)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (reps < 1 || !d_src8 || !d_src4 || !d_sink_8k) return CGH_EINVAL;
    hipStream_t s = g_cgh.stream;
    const int   grid = g_cgh.num_cu * 8;
    if (grid > 2048) return CGH_EINVAL;  // the sink holds 8 KiB
    timing_events ev;                    // (destroyed on every way out)
    CGH_HIP(hipEventCreate(&ev.a));
    CGH_HIP(hipEventCreate(&ev.b));
    CGH_HIP(hipEventRecord(ev.a, s));
    for (int k = 0; k < reps; ++k)
        hipLaunchKernelGGL(read8_4, dim3(grid), dim3(256), 0, s, (const unsigned long long *)d_src8, (const unsigned *)d_src4,
                           (unsigned *)d_sink_8k, count);
    CGH_HIP(hipEventRecord(ev.b, s));
    CGH_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    *ms_avg = (double)ms / reps;
    return CGH_OK;
}

cgh_matrix *cgh_matrix_from_ell_host(int64_t n, int K, const double *diag, const double *ell_vals, const int *ell_cols)
{
    CGH_REQUIRE_READY(nullptr);
    if (n <= 0 || K < 0 || !diag || (K > 0 && (!ell_vals || !ell_cols))) {
        cgh_set_error("cgh_matrix_from_ell_host: bad arguments (n=%lld K=%d)", (long long)n, K);
        return nullptr;
    }
    cgh_matrix *m = new cgh_matrix();
    m->n = n;
    m->K = K;
    m->layout = CGH_LAYOUT_ELL;
    m->stored = n * (int64_t)K;
    m->diag = (double *)cgh_alloc(sizeof(double) * (size_t)n);
    m->vals = (double *)cgh_alloc(sizeof(double) * (size_t)m->stored);
    m->cols = (int *)cgh_alloc(sizeof(int) * (size_t)m->stored);
    if (!m->diag || !m->vals || !m->cols) {
        cgh_matrix_free(m);
        return nullptr;
    }
    if (cgh_h2d(m->diag, diag, sizeof(double) * (size_t)n) || cgh_h2d(m->vals, ell_vals, sizeof(double) * (size_t)m->stored) ||
        cgh_h2d(m->cols, ell_cols, sizeof(int) * (size_t)m->stored)) {
        cgh_matrix_free(m);
        return nullptr;
    }
    int64_t nz = 0;
    for (int64_t k = 0; k < m->stored; ++k) nz += ell_cols[k] != -1;
    m->nnz_off = nz;
    return m;
}

void cgh_matrix_free(cgh_matrix *m)
{
    if (!m) return;
    if (m->owns) {
        cgh_free(m->diag);
        cgh_free(m->vals);
        cgh_free(m->cols);
        cgh_free(m->row_ptr);
        cgh_free(m->slice_ptr);
        cgh_free(m->bin_rows);
        cgh_free(m->blk_seg);
        cgh_free(m->blk_base);
        cgh_free(m->tile_off);
        cgh_free(m->gate_prog);
        cgh_free(m->perm);
        cgh_free(m->pb_bnd);
        cgh_free(m->pb_pre);
        cgh_free(m->pb_seg0);
        cgh_free(m->pb_bin);
        cgh_free(m->pb_cbase);
        cgh_free(m->pb_lcol);
        cgh_free(m->pb_lrow);
        cgh_free(m->pb_prod);
    }
    cgh_ell_dict_free(m);
    delete m;
}

int64_t cgh_matrix_rows(const cgh_matrix *m) { return m ? m->n : 0; }
int     cgh_matrix_width(const cgh_matrix *m) { return m ? m->K : 0; }
int64_t cgh_matrix_nnz_offdiag(const cgh_matrix *m) { return m ? m->nnz_off : 0; }
int     cgh_matrix_layout(const cgh_matrix *m) { return m ? m->layout : -1; }

int64_t cgh_matrix_stored_bytes(const cgh_matrix *m)
{
    if (!m) return 0;
    int64_t b = m->stored * 12 + m->n * 8;
    if (m->layout == CGH_LAYOUT_CSR) b += (m->n + 1) * 8;
    if (m->layout == CGH_LAYOUT_SELL) b += (m->nslices + 1) * 8;
    if (m->layout == CGH_LAYOUT_SELLS) b += (m->nslices + 1) * 8 + m->nslices * 64 * 4;
    if (m->layout == CGH_LAYOUT_CSRB) b += (m->nblocks + 1) * 8 + m->nblocks * (int64_t)(m->nslabs + 1) * 4 + m->nblocks * (m->rpw / 64) * 4;
    if (m->layout == CGH_LAYOUT_PB) b += (int64_t)m->pb_blocks * (m->pb_blocks + 1) * 4 + (int64_t)m->pb_blocks * m->pb_blocks * 4 + (m->pb_blocks + 1) * 12;
    return b;
}

int cgh_matrix_download_diag(const cgh_matrix *m, double *diag)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    return cgh_d2h(diag, m->diag, sizeof(double) * (size_t)m->n);
}

int cgh_matrix_download_ell(const cgh_matrix *m, double *ell_vals, int *ell_cols)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m) return CGH_EINVAL;
    if (m->layout == CGH_LAYOUT_ELL) {
        int rc = cgh_d2h(ell_vals, m->vals, sizeof(double) * (size_t)m->stored);
        if (rc) return rc;
        return cgh_d2h(ell_cols, m->cols, sizeof(int) * (size_t)m->stored);
    }
    return cgh_csr_to_ell_download(m, ell_vals, ell_cols);
}

// CSR layout only: the full rows as stored (row_ptr[n+1] 64-bit, cols / vals [nnz_offdiag]), in the reference's
// per-row accumulation order.
int cgh_matrix_download_csr(const cgh_matrix *m, int64_t *row_ptr, int *cols, double *vals)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || m->layout != CGH_LAYOUT_CSR || !row_ptr || !cols || !vals) {
        cgh_set_error("cgh_matrix_download_csr: needs a CGH_LAYOUT_CSR matrix and three host buffers");
        return CGH_EINVAL;
    }
    int rc = cgh_d2h(row_ptr, m->row_ptr, sizeof(int64_t) * (size_t)(m->n + 1));
    if (rc == CGH_OK) rc = cgh_d2h(cols, m->cols, sizeof(int) * (size_t)m->nnz_off);
    if (rc == CGH_OK) rc = cgh_d2h(vals, m->vals, sizeof(double) * (size_t)m->nnz_off);
    return rc;
}

int cgh_spmv(const cgh_matrix *m, const double *d_x, double *d_y)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || !d_x || !d_y) {
        cgh_set_error("cgh_spmv: NULL argument");
        return CGH_EINVAL;
    }
    return cgh_launch_spmv(m, d_x, d_y, nullptr, nullptr, nullptr, g_cgh.stream);
}

int cgh_time_spmv(const cgh_matrix *m, const double *d_x, double *d_y, int reps, double *ms_avg)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (!m || reps < 1) return CGH_EINVAL;
    hipStream_t   s = g_cgh.stream;
    timing_events ev;
    CGH_HIP(hipEventCreate(&ev.a));
    CGH_HIP(hipEventCreate(&ev.b));
    int rc = cgh_launch_spmv(m, d_x, d_y, nullptr, nullptr, nullptr, s);  // warm-up
    if (rc) return rc;
    CGH_HIP(hipEventRecord(ev.a, s));
    for (int k = 0; k < reps; ++k) {
        rc = cgh_launch_spmv(m, d_x, d_y, nullptr, nullptr, nullptr, s);
        if (rc) return rc;
    }
    CGH_HIP(hipEventRecord(ev.b, s));
    CGH_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    *ms_avg = (double)ms / reps;
    return CGH_OK;
}

// variant = unroll (1 | 2 | 4 | 8) + 16 * (non-temporal loads) + 32 * (non-temporal stores) + 256 * workgroups per CU (0 -> 64)
int cgh_time_copy_variant(void *d_dst, const void *d_src, size_t bytes, int reps, int variant, double *ms_avg)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (bytes % 16 || reps < 1 || !d_dst || !d_src || !ms_avg) return CGH_EINVAL;
    hipStream_t   s = g_cgh.stream;
    const int     un = variant & 15, wgs = (variant >> 8) > 0 ? (variant >> 8) : 64;
    const bool    ntl = (variant & 16) != 0, nts = (variant & 32) != 0;
    const int     grid = g_cgh.num_cu * wgs;
    const size_t  count = bytes / 16;
    timing_events ev;
    CGH_HIP(hipEventCreate(&ev.a));
    CGH_HIP(hipEventCreate(&ev.b));
    for (int k = 0; k <= reps; ++k) {  // (the first launch is a warm-up)
        if (k == 1) CGH_HIP(hipEventRecord(ev.a, s));
#define CGH_COPY(U, L, S) hipLaunchKernelGGL((copy16<U, L, S>), dim3(grid), dim3(256), 0, s, (const u4_t *)d_src, (u4_t *)d_dst, count)
        if (un == 8) {
            if (ntl && nts) CGH_COPY(8, true, true);
            else if (ntl) CGH_COPY(8, true, false);
            else if (nts) CGH_COPY(8, false, true);
            else CGH_COPY(8, false, false);
        } else if (un == 4) {
            if (ntl && nts) CGH_COPY(4, true, true);
            else if (ntl) CGH_COPY(4, true, false);
            else if (nts) CGH_COPY(4, false, true);
            else CGH_COPY(4, false, false);
        } else if (un == 2) {
            if (ntl && nts) CGH_COPY(2, true, true);
            else CGH_COPY(2, false, false);
        } else if (un == 1) {
            if (ntl && nts) CGH_COPY(1, true, true);
            else CGH_COPY(1, false, false);  // round 2's yardstick, kept for comparison
        } else {
            cgh_set_error("cgh_time_copy_variant: unroll %d is not built (1, 4, 8)", un);
            return CGH_EINVAL;
        }
#undef CGH_COPY
    }
    CGH_HIP(hipEventRecord(ev.b, s));
    CGH_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    *ms_avg = (double)ms / reps;
    return CGH_OK;
}

// the yardstick itself: 4 loads in flight per lane, non-temporal loads and stores, 64 workgroups per CU (tools/quick_copy.py,
// profiles/r03_copy_variants.txt: the best of the variants on MI355X, 6.2 TB/s read + written)
int cgh_time_copy(void *d_dst, const void *d_src, size_t bytes, int reps, double *ms_avg)
{
    return cgh_time_copy_variant(d_dst, d_src, bytes, reps, 4 + 16 + 32, ms_avg);
}

// Read-only ceiling: `bytes` streamed from d_src once per repetition (16-byte non-temporal loads), nothing written.
// wgs_per_cu: workgroups per CU of the grid-stride walk (1..8; tools/quick_copy.py sweeps it).
int cgh_time_read_variant(const void *d_src, void *d_sink_8k, size_t bytes, int reps, int wgs_per_cu, double *ms_avg);
int cgh_time_read(const void *d_src, void *d_sink_8k, size_t bytes, int reps, double *ms_avg)
{
    return cgh_time_read_variant(d_src, d_sink_8k, bytes, reps, CGH_READ_WGS_PER_CU, ms_avg);
}

int cgh_time_read_variant(const void *d_src, void *d_sink_8k, size_t bytes, int reps, int wgs_per_cu, double *ms_avg)
{
    CGH_REQUIRE_READY(CGH_ENODEV);
    if (bytes % 16 || reps < 1 || !d_src || !d_sink_8k || wgs_per_cu < 1 || wgs_per_cu > 8) return CGH_EINVAL;
    hipStream_t s = g_cgh.stream;
    const int   grid = g_cgh.num_cu * wgs_per_cu;  // <= 2048 workgroups: the sink holds 8 KiB
    if (grid > 2048) return CGH_EINVAL;
    timing_events ev;
    CGH_HIP(hipEventCreate(&ev.a));
    CGH_HIP(hipEventCreate(&ev.b));
    hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, s, (const u4_t *)d_src, (unsigned *)d_sink_8k, bytes / 16);
    CGH_HIP(hipEventRecord(ev.a, s));
    for (int k = 0; k < reps; ++k)
        hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, s, (const u4_t *)d_src, (unsigned *)d_sink_8k, bytes / 16);
    CGH_HIP(hipEventRecord(ev.b, s));
    CGH_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CGH_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    *ms_avg = (double)ms / reps;
    return CGH_OK;
}

} // extern "C"
