"""rocprofv3 target: a few strict-mode iterations at two sizes (per-kernel cost of the order-exact sums)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
os.environ["CGH_XSUM_MIN_ROWS"] = "0"
shape = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [400, 320, 1]
n = shape[0] * shape[1] * shape[2]
b = cga.DeviceVector(n); x = cga.DeviceVector(n)
A = cga.Matrix.poisson(shape[0], shape[1], shape[2], 7, 0x5EED0001, cga.LAYOUT_ELL, b)
rep, hist = cga.cg_solve(A, b, x, fixed_iters=20, strict=1)
print(n, rep.solve_ms / 20)
