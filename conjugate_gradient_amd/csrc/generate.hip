// generate.hip -- seeded synthetic SPD matrices, generated on the device.
//
// No reference counterpart (the reference's generators use rand(), src/coo.c:31,116 there); these
// are the BASELINE.json benchmark shapes.  Every value is a pure function of (seed, index) through
// one 64-bit integer hash, so the numpy twin in tests/foamio.py produces the same bits.
//
// Stencil matrices (7/27-point on an nx*ny*nz grid, row = x + nx*(y + ny*z)): off-diagonals -1.0,
// diag = (points-1) + 1e-3*hash01(seed,row), b = hash01(seed+1,row) - 0.5.  The upper-triangular
// COO is row-sorted with offsets in lexicographic (dz,dy,dx) order; converting it with the
// reference's coo_to_ell_symmetric_full_colmajor (src/ell.c:791-858) puts a row's neighbours in
// ascending column order (lower neighbours arrive from earlier rows in entry order, then its own
// upper entries), so the full ELL can be written directly, one thread per row, bit-identical to
// that conversion (tests/test_generate.py checks it through the oracle).
#include "cgh_internal.h"
#include "device_utils.h"

namespace {

template <int POINTS>
__global__ __launch_bounds__(256) void poisson_ell(int nx, int ny, int nz, int K, uint64_t seed, double *__restrict__ diag,
                                                  double *__restrict__ vals, int *__restrict__ cols, double *__restrict__ b)
{
    const int64_t n = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((int64_t)nx * ny));
        int       slot = 0;
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
                    const int64_t c = i + dx + (int64_t)nx * (dy + (int64_t)ny * dz);
                    vals[(int64_t)slot * n + i] = -1.0;
                    cols[(int64_t)slot * n + i] = (int)c;
                    ++slot;
                }
        for (; slot < K; ++slot) {
            vals[(int64_t)slot * n + i] = 0.0;
            cols[(int64_t)slot * n + i] = -1;
        }
        diag[i] = (double)(POINTS - 1) + 1e-3 * cgh_hash01(seed, (uint64_t)i);
        if (b != nullptr) b[i] = cgh_hash01(seed + 1, (uint64_t)i) - 0.5;
    }
}

// Upper-triangular COO of the same matrix: count pass (per-row upper degree), scan on the host of
// a tiny per-row formula is avoided by computing each row's start in closed form per (y,z) line.
template <int POINTS>
__device__ inline int upper_degree(int x, int y, int z, int nx, int ny, int nz)
{
    int d = 0;
    for (int dz = 0; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                if (dz == 0 && (dy < 0 || (dy == 0 && dx <= 0))) continue;
                if (POINTS == 7 && (dz != 0) + (dy != 0) + (dx != 0) != 1) continue;
                const int xx = x + dx, yy = y + dy, zz = z + dz;
                if (xx < 0 || xx >= nx || yy < 0 || yy >= ny || zz >= nz) continue;
                ++d;
            }
    return d;
}

template <int POINTS>
__global__ __launch_bounds__(256) void poisson_upper_degree(int nx, int ny, int nz, int64_t *__restrict__ deg)
{
    const int64_t n = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((int64_t)nx * ny));
        deg[i] = upper_degree<POINTS>(x, y, z, nx, ny, nz);
    }
}

template <int POINTS>
__global__ __launch_bounds__(256) void poisson_coo_fill(int nx, int ny, int nz, const int64_t *__restrict__ start,
                                                       double *__restrict__ upper, int *__restrict__ rows,
                                                       int *__restrict__ cols)
{
    const int64_t n = (int64_t)nx * ny * nz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((int64_t)nx * ny));
        int64_t   at = start[i];
        for (int dz = 0; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dz == 0 && (dy < 0 || (dy == 0 && dx <= 0))) continue;
                    if (POINTS == 7 && (dz != 0) + (dy != 0) + (dx != 0) != 1) continue;
                    const int xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || xx >= nx || yy < 0 || yy >= ny || zz >= nz) continue;
                    upper[at] = -1.0;
                    rows[at] = (int)i;
                    cols[at] = (int)(i + dx + (int64_t)nx * (dy + (int64_t)ny * dz));
                    ++at;
                }
    }
}

__global__ __launch_bounds__(256) void poisson_diag_rhs(int64_t n, int points, uint64_t seed, double *__restrict__ diag,
                                                       double *__restrict__ b)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        diag[i] = (double)(points - 1) + 1e-3 * cgh_hash01(seed, (uint64_t)i);
        if (b != nullptr) b[i] = cgh_hash01(seed + 1, (uint64_t)i) - 0.5;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Power-law SPD matrix (BASELINE config 5: skewed row lengths, 1..max_len entries per row).  Row i asks for
// L_i = min(max_len, max(1, floor(1/u_i))) entries (Pareto, exponent 2; u_i = hash01(seed, i)) and receives
// m_i = min(ceil(L_i/2), n-i-1) strictly-upper columns  j = i+1 + hash64(seed+1, i*2048+t) mod (n-i-1), t < m_i,
// first occurrence kept; value = -hash01(seed+2, i*2048+t) in (-1, 0].  The lower triangle mirrors it, so a full row
// holds its own upper entries plus what other rows point at it.  diag_i = 1 + sum_j |a_ij| over the full row, added in
// the row's stored (reference) order -- strictly diagonally dominant, hence SPD.  b = hash01(seed+3, i) - 0.5.
__device__ __host__ inline uint64_t cgh_hash64(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed ^ (idx * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline int pl_want(uint64_t seed, int64_t i, int64_t n, int max_len)
{
    const double u = cgh_hash01(seed, (uint64_t)i);
    double       L = u > 0.0 ? floor(1.0 / u) : (double)max_len;
    if (L > (double)max_len) L = (double)max_len;
    if (L < 1.0) L = 1.0;
    int64_t m = ((int64_t)L + 1) / 2;
    if (m > n - i - 1) m = n - i - 1;
    return (int)m;
}

__device__ inline int64_t pl_col(uint64_t seed, int64_t i, int64_t n, int t)
{
    return i + 1 + (int64_t)(cgh_hash64(seed + 1, (uint64_t)i * 2048u + (uint64_t)t) % (uint64_t)(n - i - 1));
}

// FILL=false: deg[i] = distinct columns of row i.  FILL=true: write them (generation order) from start[i].
template <bool FILL>
__global__ __launch_bounds__(256) void powerlaw_rows(int64_t n, uint64_t seed, int max_len, int64_t *__restrict__ deg,
                                                    const int64_t *__restrict__ start, double *__restrict__ upper,
                                                    int *__restrict__ rows, int *__restrict__ cols)
{
    // one wave per row: lanes test candidates t, t+64, ... against all earlier candidates
    const int     lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int m = pl_want(seed, i, n, max_len);
        int64_t   at = FILL ? start[i] : 0;
        int       kept = 0;
        for (int t0 = 0; t0 < m; t0 += 64) {
            const int t = t0 + lane;
            bool      keep = false;
            int64_t   c = 0;
            if (t < m) {
                c = pl_col(seed, i, n, t);
                keep = true;
                for (int s = 0; s < t && keep; ++s) keep = pl_col(seed, i, n, s) != c;
            }
            const unsigned long long mask = __ballot(keep);
            if (FILL && keep) {
                const int64_t pos = at + __popcll(mask & ((1ull << lane) - 1ull));
                upper[pos] = -cgh_hash01(seed + 2, (uint64_t)i * 2048u + (uint64_t)t);
                rows[pos] = (int)i;
                cols[pos] = (int)c;
            }
            at += __popcll(mask);
            kept += __popcll(mask);
        }
        if (!FILL && lane == 0) deg[i] = kept;
    }
}

__global__ __launch_bounds__(256) void powerlaw_diag_rhs(int64_t n, uint64_t seed, const int64_t *__restrict__ row_ptr,
                                                        const double *__restrict__ vals, double *__restrict__ diag,
                                                        double *__restrict__ b)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double s = 1.0;
        for (int64_t q = row_ptr[i]; q < row_ptr[i + 1]; ++q) s += fabs(vals[q]);
        diag[i] = s;
        if (b != nullptr) b[i] = cgh_hash01(seed + 3, (uint64_t)i) - 0.5;
    }
}

__global__ __launch_bounds__(256) void copy_diag(int64_t n, const double *__restrict__ src, double *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

int width_of(int nx, int ny, int nz, int points)
{
    auto m3 = [](int d) { return d < 3 ? d : 3; };
    if (points == 27) return m3(nx) * m3(ny) * m3(nz) - 1;
    return (m3(nx) - 1) + (m3(ny) - 1) + (m3(nz) - 1);
}

int64_t offdiag_of(int64_t nx, int64_t ny, int64_t nz, int points)
{
    if (points == 27) return (3 * nx - 2) * (3 * ny - 2) * (3 * nz - 2) - nx * ny * nz;
    return 2 * ((nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1));
}

} // namespace

int cgh_exclusive_scan_i64(const int64_t *d_in, int64_t *d_out, int64_t count, int64_t *h_total, hipStream_t s);

extern "C" {

cgh_matrix *cgh_matrix_poisson(int nx, int ny, int nz, int points, uint64_t seed, int layout, double *d_b)
{
    CGH_REQUIRE_READY(nullptr);
    if (nx < 1 || ny < 1 || nz < 1 || (points != 7 && points != 27)) {
        cgh_set_error("cgh_matrix_poisson: bad shape %dx%dx%d points=%d", nx, ny, nz, points);
        return nullptr;
    }
    const int64_t n = (int64_t)nx * ny * nz;
    if (n > 0x7fffffffLL) {
        cgh_set_error("cgh_matrix_poisson: %lld rows exceed int32 column ids", (long long)n);
        return nullptr;
    }
    hipStream_t s = g_cgh.stream;
    const int   grid = g_cgh.num_cu * 8;
    if (layout != CGH_LAYOUT_ELL) {
        // general path: device COO -> device conversion (exercises convert.hip at full size)
        const int64_t U = offdiag_of(nx, ny, nz, points) / 2;
        int64_t      *deg = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(n + 1));
        double       *diag = (double *)cgh_alloc(sizeof(double) * (size_t)n);
        double       *up = (double *)cgh_alloc(sizeof(double) * (size_t)(U ? U : 1));
        int          *rows = (int *)cgh_alloc(sizeof(int) * (size_t)(U ? U : 1));
        int          *cols = (int *)cgh_alloc(sizeof(int) * (size_t)(U ? U : 1));
        cgh_matrix   *m = nullptr;
        if (deg && diag && up && rows && cols) {
            if (points == 7)
                hipLaunchKernelGGL(poisson_upper_degree<7>, dim3(grid), dim3(256), 0, s, nx, ny, nz, deg);
            else
                hipLaunchKernelGGL(poisson_upper_degree<27>, dim3(grid), dim3(256), 0, s, nx, ny, nz, deg);
            int64_t total = 0;
            if (cgh_exclusive_scan_i64(deg, deg, n, &total, s) == CGH_OK && total == U) {
                if (points == 7)
                    hipLaunchKernelGGL(poisson_coo_fill<7>, dim3(grid), dim3(256), 0, s, nx, ny, nz, (const int64_t *)deg, up, rows, cols);
                else
                    hipLaunchKernelGGL(poisson_coo_fill<27>, dim3(grid), dim3(256), 0, s, nx, ny, nz, (const int64_t *)deg, up, rows, cols);
                hipLaunchKernelGGL(poisson_diag_rhs, dim3(grid), dim3(256), 0, s, n, points, seed, diag, d_b);
                m = cgh_matrix_from_coo_device(n, U, diag, up, rows, cols, layout);
            } else {
                cgh_set_error("cgh_matrix_poisson: upper count mismatch (%lld vs %lld)", (long long)total, (long long)U);
            }
        }
        cgh_free(deg);
        cgh_free(diag);
        cgh_free(up);
        cgh_free(rows);
        cgh_free(cols);
        return m;
    }
    cgh_matrix *m = new cgh_matrix();
    m->n = n;
    m->K = width_of(nx, ny, nz, points);
    m->layout = CGH_LAYOUT_ELL;
    m->stored = n * (int64_t)m->K;
    m->nnz_off = offdiag_of(nx, ny, nz, points);
    m->diag = (double *)cgh_alloc(sizeof(double) * (size_t)n);
    m->vals = (double *)cgh_alloc(sizeof(double) * (size_t)m->stored);
    m->cols = (int *)cgh_alloc(sizeof(int) * (size_t)m->stored);
    if (!m->diag || !m->vals || !m->cols) {
        cgh_matrix_free(m);
        return nullptr;
    }
    if (points == 7)
        hipLaunchKernelGGL(poisson_ell<7>, dim3(grid), dim3(256), 0, s, nx, ny, nz, m->K, seed, m->diag, m->vals, m->cols, d_b);
    else
        hipLaunchKernelGGL(poisson_ell<27>, dim3(grid), dim3(256), 0, s, nx, ny, nz, m->K, seed, m->diag, m->vals, m->cols, d_b);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        cgh_set_error("cgh_matrix_poisson: generator kernel failed");
        cgh_matrix_free(m);
        return nullptr;
    }
    return m;
}

} // extern "C"

extern "C" {

// Power-law SPD matrix in `layout` (CSR is built first: its rows define the diagonal's summation order).
cgh_matrix *cgh_matrix_powerlaw(int64_t n, int max_len, uint64_t seed, int layout, double *d_b)
{
    CGH_REQUIRE_READY(nullptr);
    if (n < 2 || n > 0x7fffffffLL || max_len < 1 || max_len > 2048) {
        cgh_set_error("cgh_matrix_powerlaw: bad arguments (n=%lld max_len=%d; max_len <= 2048)", (long long)n, max_len);
        return nullptr;
    }
    hipStream_t s = g_cgh.stream;
    const int   grid = g_cgh.num_cu * 8;
    int64_t    *deg = (int64_t *)cgh_alloc(sizeof(int64_t) * (size_t)(n + 1));
    double     *zero = (double *)cgh_alloc(sizeof(double) * (size_t)n);
    if (!deg || !zero) {
        cgh_free(deg);
        cgh_free(zero);
        return nullptr;
    }
    cgh_matrix *csr = nullptr, *out = nullptr;
    double     *up = nullptr;
    int        *rows = nullptr, *cols = nullptr;
    hipLaunchKernelGGL(powerlaw_rows<false>, dim3(grid), dim3(256), 0, s, n, seed, max_len, deg, (const int64_t *)nullptr,
                       (double *)nullptr, (int *)nullptr, (int *)nullptr);
    int64_t U = 0;
    bool    ok = cgh_exclusive_scan_i64(deg, deg, n, &U, s) == CGH_OK && hipMemsetAsync(zero, 0, sizeof(double) * (size_t)n, s) == hipSuccess;
    if (ok) {
        up = (double *)cgh_alloc(sizeof(double) * (size_t)(U ? U : 1));
        rows = (int *)cgh_alloc(sizeof(int) * (size_t)(U ? U : 1));
        cols = (int *)cgh_alloc(sizeof(int) * (size_t)(U ? U : 1));
        ok = up && rows && cols;
    }
    if (ok) {
        hipLaunchKernelGGL(powerlaw_rows<true>, dim3(grid), dim3(256), 0, s, n, seed, max_len, (int64_t *)nullptr, (const int64_t *)deg, up,
                           rows, cols);
        ok = hipGetLastError() == hipSuccess;
    }
    if (ok) csr = cgh_matrix_from_coo_device(n, U, zero, up, rows, cols, CGH_LAYOUT_CSR);
    if (csr) {
        hipLaunchKernelGGL(powerlaw_diag_rhs, dim3(grid), dim3(256), 0, s, n, seed, (const int64_t *)csr->row_ptr, (const double *)csr->vals,
                           csr->diag, d_b);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        if (ok && layout == CGH_LAYOUT_CSR) {
            out = csr;
            csr = nullptr;
        } else if (ok) {
            out = cgh_matrix_from_coo_device(n, U, csr->diag, up, rows, cols, layout);
        }
    }
    cgh_matrix_free(csr);
    cgh_free(deg);
    cgh_free(zero);
    cgh_free(up);
    cgh_free(rows);
    cgh_free(cols);
    return out;
}

} // extern "C"
