"""us/iteration of the strict (bit-identical) loop at a few sizes (or at the one given: quick_strict.py nx ny nz [iters])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
SHAPES = (((40, 40, 20), 100), ((400, 320, 1), 100), ((128, 128, 128), 60), ((256, 256, 256), 40))
if len(sys.argv) >= 4:
    SHAPES = ((tuple(int(v) for v in sys.argv[1:4]), int(sys.argv[4]) if len(sys.argv) > 4 else 40),)
for shape, iters in SHAPES:
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    cga.cg_solve(A, b, x, fixed_iters=3, strict=1)
    best = 1e9
    for rep in range(2):
        r, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
        best = min(best, r.solve_ms / iters * 1e3)
    print("n=%9d strict %8.1f us/it   res[-1]=%.17g" % (n, best, hist[-1, 0]), flush=True)
