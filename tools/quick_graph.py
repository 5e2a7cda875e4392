"""Eager vs HIP-graph replay of the 3-kernel iteration at several sizes (us/iteration, graph already instantiated)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
os.environ["CGH_GRAPH_MAX_ROWS"] = "1000000000"
for shape in ((40, 40, 20), (400, 320, 1), (128, 128, 64), (128, 128, 128), (256, 256, 128)):
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    res = {}
    for g in ("0", "1"):
        os.environ["CGH_GRAPH"] = g
        cga.check(cga.lib().cgh_config_reload())   # the knobs are read at cgh_init / on request only
        sv = cga.Solver(A, 1300)
        sv.start(b, x, tol=-1.0, strict=0)
        sv.iterate(200, 0)                      # (instantiates the 200-iteration graph)
        cga.check(cga.lib().cgh_sync())
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            sv.iterate(200, 0)
            cga.check(cga.lib().cgh_sync())
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        res[g] = best
        sv.free()
    print("n=%9d  eager %7.1f us/it   graph replay %7.1f us/it" % (n, res["0"], res["1"]), flush=True)
