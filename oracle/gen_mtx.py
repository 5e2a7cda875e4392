#!/usr/bin/env python3
"""gen_mtx.py -- writes the MatrixMarket fixtures under tests/golden/mtx/ (TEST INFRASTRUCTURE).

The reference's own .mtx inputs (data/mkt_real_symmetric/bcsstk*.mtx, read by tests/test_vec.c:803) are not copied
into this repo; these files have the same shape of content -- `coordinate real symmetric`, lower triangle stored,
sorted by column then row, values in %.13e -- and are synthetic, seeded, SPD by diagonal dominance:

  plate_24x18.mtx       9-point 2-D stencil with element stiffnesses spread over 6 decades (n = 432)
  skewed_1500.mtx       power-law row lengths (a few hub rows of several hundred entries), for the CSR-adaptive path
  general_60.mtx        `general`, both triangles present, unsorted, '%' comments and blank lines, a repeated diagonal
  pattern_40.mtx        `pattern symmetric` (no values)
  integer_30.mtx        `integer symmetric`

Run from the repo root:  python oracle/gen_mtx.py
"""
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mtx")


def write(name, field, symmetry, n, entries, comments=()):
    with open(os.path.join(OUT, name), "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n" % (field, symmetry))
        for c in comments:
            f.write(c + "\n")
        f.write("%d %d %d\n" % (n, n, len(entries)))
        for e in entries:
            if field == "pattern":
                f.write("%d %d\n" % (e[0] + 1, e[1] + 1))
            elif field == "integer":
                f.write("%d %d %d\n" % (e[0] + 1, e[1] + 1, int(e[2])))
            else:
                f.write("%d %d %20.13e\n" % (e[0] + 1, e[1] + 1, e[2]))


def lower_sorted(n, upper):
    """upper: dict (i<j) -> v.  Lower-triangle storage sorted by column then row, diagonal dominance."""
    rowsum = np.zeros(n)
    for (i, j), v in upper.items():
        rowsum[i] += abs(v)
        rowsum[j] += abs(v)
    ent = [(i, i, rowsum[i] * 1.05 + 1e-3) for i in range(n)]
    ent += [(j, i, v) for (i, j), v in upper.items()]      # stored as (row=j > col=i)
    ent.sort(key=lambda e: (e[1], e[0]))
    return ent


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0xC0FFEE)

    nx, ny = 24, 18
    n = nx * ny
    up = {}
    for y in range(ny):
        for x in range(nx):
            i = x + nx * y
            for dx, dy in ((1, 0), (-1, 1), (0, 1), (1, 1)):
                xx, yy = x + dx, y + dy
                if 0 <= xx < nx and 0 <= yy < ny:
                    up[(i, xx + nx * yy)] = -(10.0 ** rng.uniform(0, 6)) * rng.uniform(0.5, 1.0)
    write("plate_24x18.mtx", "real", "symmetric", n, lower_sorted(n, up))

    n = 1500
    up = {}
    length = np.minimum(600, np.maximum(1, (rng.random(n) ** -1.0).astype(int)))
    for i in range(n - 1):
        for j in rng.integers(i + 1, n, int(np.ceil(length[i] / 2))):
            up[(i, int(j))] = -rng.random()
    write("skewed_1500.mtx", "real", "symmetric", n, lower_sorted(n, up))

    n = 60
    ent = []
    for _ in range(400):
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        ent.append((i, j, float(rng.standard_normal())))
    ent += [(i, i, 50.0 + i) for i in range(n)] + [(7, 7, 99.5)]          # the later (7,7) wins
    order = rng.permutation(len(ent))
    ent = [ent[k] for k in order if not (ent[k][0] == 7 and ent[k][1] == 7)] + [(7, 7, 50.0 + 7), (7, 7, 99.5)]
    write("general_60.mtx", "real", "general", n, ent, comments=("% synthetic, unsorted, both triangles", "%", ""))

    n = 40
    ent = sorted({(int(max(a, b)), int(min(a, b))) for a, b in rng.integers(0, n, (150, 2))}, key=lambda e: (e[1], e[0]))
    write("pattern_40.mtx", "pattern", "symmetric", n, ent)

    n = 30
    ent = [(i, i, 20 + i) for i in range(n)] + [(i + 1, i, -(1 + i % 3)) for i in range(n - 1)] + [(i + 5, i, -1) for i in range(n - 5)]
    ent.sort(key=lambda e: (e[1], e[0]))
    write("integer_30.mtx", "integer", "symmetric", n, ent)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
