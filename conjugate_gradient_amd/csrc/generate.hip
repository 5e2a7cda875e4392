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
