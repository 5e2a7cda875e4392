"""Where the serial walk of the order-exact sums spends its time (xsum.hip, XS_PROF_ADD): per walk, averaged over the
strict-mode iterations of a Poisson problem.  python tools/prof_drive.py [nx ny nz]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conjugate_gradient_amd as cga  # noqa: E402

cga.init(0)
L = cga.lib()
for shape in ((256, 256, 256), (128, 128, 128), (400, 320, 1), (40, 40, 20)):
    n = shape[0] * shape[1] * shape[2]
    b = cga.DeviceVector(n); x = cga.DeviceVector(n)
    A = cga.Matrix.poisson(*shape, 7, 0x5EED0001, cga.LAYOUT_ELL, b)
    cga.cg_solve(A, b, x, fixed_iters=3, strict=1)
    prof = cga.DeviceVector(16); prof.zero()
    L.cgh_xsum_set_prof(prof.ptr)
    iters = 30
    r, hist = cga.cg_solve(A, b, x, fixed_iters=iters, strict=1)
    L.cgh_xsum_set_prof(None)
    p = prof.to_host().view(np.int64)
    walks = max(int(p[6]), 1)
    print("n=%9d  %7.1f us/it | per walk (%d walks): batches descended into %.1f  groups cell-by-cell %.1f  cells element-by-element %.1f  "
          "runs with a crossing or a tie applied whole %.1f | %.1f us in the driver: %.1f staging + joins, %.1f walk; xs_batch calls %.1f, scans %.1f; start-up %.1f us" % (
              n, r.solve_ms / iters * 1e3, walks, p[0] / walks, p[1] / walks, p[2] / walks, p[7] / walks, p[3] / walks / 100.0, p[4] / walks / 100.0, p[5] / walks / 100.0, p[9] / walks, p[8] / walks, p[10] / walks / 100.0), flush=True)
