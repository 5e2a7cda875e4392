// temporary stubs until convert.hip / spmv_csr.hip land
#include "cgh_internal.h"
int cgh_launch_spmv_csr(const cgh_matrix *, const double *, double *, double *, int *, const int *, hipStream_t) { cgh_set_error("csr: not built yet"); return CGH_EINVAL; }
int cgh_launch_spmv_sell(const cgh_matrix *, const double *, double *, double *, int *, const int *, hipStream_t) { cgh_set_error("sell: not built yet"); return CGH_EINVAL; }
int cgh_csr_to_ell_download(const cgh_matrix *, double *, int *) { return CGH_EINVAL; }
int cgh_exclusive_scan_i64(const int64_t *, int64_t *, int64_t, int64_t *, hipStream_t) { return CGH_EINVAL; }
extern "C" cgh_matrix *cgh_matrix_from_coo_device(int64_t, int64_t, const double *, const double *, const int *, const int *, int) { return nullptr; }
extern "C" cgh_matrix *cgh_matrix_from_coo_host(int64_t, int64_t, const double *, const double *, const int *, const int *, int) { return nullptr; }
