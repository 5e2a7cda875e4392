// partition.h -- internal layout of a rank's share (see partition.cpp); C API in include/cg_hip.h.
#pragma once
#include <cstdint>
#include <vector>

struct cgh_neighbour {
    int              rank = -1;
    std::vector<int> send_idx;          // my local rows the neighbour needs (ascending)
    int64_t          recv_off = 0;      // offset of its share inside the halo segment
    int64_t          recv_cnt = 0;
};

struct cgh_partition {
    int64_t              n_global = 0;
    int                  nranks = 1, rank = 0;
    std::vector<int64_t> starts;        // nranks+1 block starts
    int64_t              lo = 0, n_loc = 0;
    int                  K = 0;
    int64_t              nnz_off = 0;
    std::vector<double>  diag, vals;    // local slot-major ELL (n_loc x K), local column ids
    std::vector<int>     cols;
    std::vector<int64_t> halo_global;   // global column of each halo slot (sorted)
    std::vector<cgh_neighbour> nbs;
    int64_t              int_begin = 0, int_end = 0;  // rows [int_begin,int_end) reference no halo column
};
