#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cg_hip.h"
extern "C" void cgh_set_error(const char *, ...) {}
int main()
{
    // 7-point Poisson nx*ny*nz upper COO, row-sorted; also a skewed random matrix
    for (int variant = 0; variant < 2; ++variant)
        for (int G : {1, 2, 3, 5, 8}) {
            const int nx = 7, ny = 5, nz = 9, n = nx * ny * nz;
            std::vector<double> diag(n, 6.5), up;
            std::vector<int> rows, cols;
            for (int i = 0; i < n; ++i) {
                const int x = i % nx, y = (i / nx) % ny, z = i / (nx * ny);
                if (variant == 0) {
                    if (x + 1 < nx) { rows.push_back(i); cols.push_back(i + 1); up.push_back(-1); }
                    if (y + 1 < ny) { rows.push_back(i); cols.push_back(i + nx); up.push_back(-1); }
                    if (z + 1 < nz) { rows.push_back(i); cols.push_back(i + nx * ny); up.push_back(-1); }
                } else {
                    for (int t = 0; t < (i % 11 == 0 ? 40 : 2); ++t) {
                        const int j = i + 1 + (int)((i * 2654435761u + t * 40503u) % (unsigned)(n - i > 1 ? n - i - 1 : 1));
                        if (j < n) { rows.push_back(i); cols.push_back(j); up.push_back(-0.25); }
                    }
                }
            }
            for (int r = 0; r < G; ++r) {
                cgh_partition *P = cgh_partition_from_coo_host(n, (int64_t)rows.size(), diag.data(), up.data(), rows.data(), cols.data(), G, r, nullptr);
                if (!P) { std::printf("variant %d G %d rank %d: NULL\n", variant, G, r); continue; }
                const int64_t nl = cgh_partition_rows(P); const int K = cgh_partition_width(P);
                std::vector<double> d(nl), ev((size_t)nl * K); std::vector<int> ec((size_t)nl * K);
                std::vector<int64_t> hg(cgh_partition_halo(P) + 1);
                cgh_partition_copy_local(P, d.data(), ev.data(), ec.data(), hg.data());
                for (int k = 0; k < cgh_partition_neighbours(P); ++k) {
                    int rk; int64_t sc, ro, rc;
                    cgh_partition_neighbour(P, k, &rk, &sc, &ro, &rc);
                    std::vector<int> sl(sc + 1);
                    cgh_partition_copy_send_list(P, k, sl.data());
                }
                cgh_partition_free(P);
            }
        }
    std::puts("partition ok");
    return 0;
}
