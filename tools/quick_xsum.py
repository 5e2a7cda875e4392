"""Strict-mode CG cost per iteration: order-exact parallel sums (xsum.hip) vs the one-lane sums (seq_sums) vs throughput mode.
python tools/quick_xsum.py  (GPU)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)


def run(nx, ny, nz, iters, strict, xsum_min):
    os.environ["CGH_XSUM_MIN_ROWS"] = str(xsum_min)
    n = nx * ny * nz
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(nx, ny, nz, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    cga.cg_solve(A, b, x, fixed_iters=2, strict=strict)          # warm-up
    rep, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=strict)
    return rep.solve_ms / iters * 1e3, hist


for (nx, ny, nz, it_seq) in ((16, 16, 8, 50), (16, 16, 16, 50), (20, 20, 20, 50), (40, 40, 20, 50), (400, 320, 1, 50), (128, 128, 64, 20), (256, 256, 64, 5), (256, 256, 256, 2)):
    n = nx * ny * nz
    t_x, h_x = run(nx, ny, nz, 50, 1, 0)
    t_s, h_s = run(nx, ny, nz, it_seq, 1, -1)
    t_f, _ = run(nx, ny, nz, 50, 0, 0)
    m = min(len(h_x), len(h_s))
    same = (h_x[1:m, :2].tobytes() == h_s[1:m, :2].tobytes())
    print("n=%9d  xsum %9.1f us/it   one-lane %11.1f us/it   throughput %8.1f us/it   same bits: %s" % (n, t_x, t_s, t_f, same), flush=True)
