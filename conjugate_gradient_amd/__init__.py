"""conjugate_gradient_amd -- MI355X-native CG solver behind the reference's C API.

The product is libcg_hip.so (hand-written HIP for gfx950, C ABI in include/*.h).
This package is only the ctypes view of that ABI used by tests/ and bench.py;
it carries no compute of its own and raises if the library or a GPU is missing.
"""
from .binding import (CgOpts, CgReport, DeviceVector, Matrix, lib, init, have_gpu,  # noqa: F401
                      LAYOUT_ELL, LAYOUT_CSR, LAYOUT_SELL, LAYOUT_CSRB, LAYOUT_SELLS, LAYOUT_PB, LIB_PATH,
                      config_csrb, config_sell_sigma, csrb_plan, config_pb, pb_partition, cg_solve, Solver, check,
                      Partition, DistRank, group_start, group_iterate, comm_unique_id, comm_init, comm_allreduce)
from . import binding  # noqa: F401
