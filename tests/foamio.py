"""Test-side readers for the OpenFOAM LDU dumps under tests/golden/ and the
seeded synthetic generators (numpy twins of the device generators).

Format as read by the reference's parser (src/parser.c:29-107 there): scan for a
keyword token, then `<count> ( v0 v1 ... )`.  Keyword order in the file is
source, diag, upper, upperAddr, lowerAddr (tests/tests_cg.c:1095-1099 there);
row = lowerAddr, col = upperAddr.
"""
import os
import re

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _array(text, key, dtype):
    m = re.search(r"(?<![A-Za-z])" + key + r"\s+(\d+)\s*\(", text)
    if m is None:
        raise ValueError("keyword %r not found" % key)
    count = int(m.group(1))
    end = text.index(")", m.end())
    vals = np.array(text[m.end():end].split(), dtype=dtype)
    if vals.size != count:
        raise ValueError("%s: expected %d values, got %d" % (key, count, vals.size))
    return vals


def read_system(path):
    """-> dict(n, U, source, diag, upper, rows(lowerAddr), cols(upperAddr))"""
    with open(path) as f:
        text = f.read()
    source = _array(text, "source", np.float64)
    diag = _array(text, "diag", np.float64)
    upper = _array(text, "upper", np.float64)
    cols = _array(text, "upperAddr", np.int32)
    rows = _array(text, "lowerAddr", np.int32)
    assert source.size == diag.size and upper.size == cols.size == rows.size
    return dict(n=int(diag.size), U=int(upper.size), source=source, diag=diag,
                upper=upper, rows=rows, cols=cols)


def read_foam_field(path):
    """internalField nonuniform List<scalar> reader (src/parser.c:109-177 there)."""
    with open(path) as f:
        lines = f.read().splitlines()
    for k, line in enumerate(lines):
        if "internalField   nonuniform List<scalar>" in line:
            count = int(lines[k + 1].split()[0])
            j = k + 2
            while "(" not in lines[j]:
                j += 1
            return np.array([float(s) for s in lines[j + 1:j + 1 + count]], dtype=np.float64)
    raise ValueError("internalField not found")


def foam(name):
    return read_system(os.path.join(GOLDEN, "cylinder_%s.system" % name))


# --------------------------------------------------------------------------
# Seeded synthetic matrices.  Pure functions of (seed, index): the device
# generators in conjugate_gradient_amd/csrc/generate.hip use the same integer
# hash, so host and device agree bit for bit (DESIGN.md "Generators").
# --------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash01(seed, idx):
    """splitmix64 finaliser of (seed ^ idx*golden) -> double in [0,1)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) ^ (np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def poisson_coo(nx, ny, nz, seed, points=7):
    """Upper-triangular COO (row-sorted, columns ascending per row) of the 7- or
    27-point stencil on an nx*ny*nz grid, row = x + nx*(y + ny*z).
    off-diagonals -1.0; diag = (points-1) + 1e-3*hash01(seed,i);
    b = hash01(seed+1,i) - 0.5."""
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64)
    x = idx % nx
    y = (idx // nx) % ny
    z = idx // (nx * ny)
    if points == 7:
        offs = [(0, 0, 1), (0, 1, 0), (1, 0, 0)]  # (dz,dy,dx): +1, +nx, +nx*ny
    else:
        offs = [(dz, dy, dx) for dz in (0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)
                if (dz, dy, dx) > (0, 0, 0)]
    rows, cols, order = [], [], []
    for t, (dz, dy, dx) in enumerate(offs):
        ok = ((x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz < nz))
        r = idx[ok]
        rows.append(r)
        cols.append(r + dx + nx * (dy + ny * dz))
        order.append(np.full(r.size, t))
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    order = np.concatenate(order)
    perm = np.lexsort((order, rows))
    rows = rows[perm].astype(np.int32)
    cols = cols[perm].astype(np.int32)
    upper = np.full(rows.size, -1.0)
    diag = float(points - 1) + 1e-3 * hash01(seed, idx)
    b = hash01(seed + 1, idx) - 0.5
    return dict(n=n, U=int(rows.size), diag=diag, upper=upper, rows=rows, cols=cols, source=b)


def hash64(seed, idx):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) ^ (np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def powerlaw_coo(n, max_len, seed):
    """numpy twin of cgh_matrix_powerlaw's upper-triangular COO (generate.hip): row-sorted, generation order inside a
    row, duplicates dropped (first kept).  The diagonal (1 + sum |a_ij| in the row's stored order) is the caller's job."""
    rows, cols, vals = [], [], []
    u = hash01(seed, np.arange(n))
    with np.errstate(divide="ignore"):
        L = np.where(u > 0, np.floor(1.0 / np.where(u > 0, u, 1.0)), max_len)
    L = np.clip(L, 1, max_len).astype(np.int64)
    m = np.minimum((L + 1) // 2, n - np.arange(n) - 1)
    for i in range(n):
        if m[i] <= 0:
            continue
        t = np.arange(m[i], dtype=np.uint64)
        c = i + 1 + (hash64(seed + 1, np.uint64(i) * np.uint64(2048) + t) % np.uint64(n - i - 1)).astype(np.int64)
        _, first = np.unique(c, return_index=True)
        keep = np.sort(first)
        rows.append(np.full(keep.size, i, dtype=np.int32))
        cols.append(c[keep].astype(np.int32))
        vals.append(-hash01(seed + 2, np.uint64(i) * np.uint64(2048) + t[keep]))
    return dict(n=n, rows=np.concatenate(rows), cols=np.concatenate(cols), upper=np.concatenate(vals),
                source=hash01(seed + 3, np.arange(n)) - 0.5)


def mtx_split(path):
    """Pure-Python restatement of the reference's inline MatrixMarket reader (tests/test_vec.c:443-560 there):
    diag / strict-upper COO in file order, lower entries of a `symmetric` file flipped, of a `general` file dropped."""
    with open(path) as f:
        head = f.readline().split()
        field, symmetry = head[3].lower(), head[4].lower()
        line = f.readline()
        while line.strip() == "" or line.lstrip().startswith("%"):
            line = f.readline()
        n, m, nz = (int(t) for t in line.split())
        assert n == m
        tok = f.read().split()
    per = 2 if field == "pattern" else 3
    diag = np.zeros(n)
    rows, cols, vals = [], [], []
    for k in range(nz):
        i, j = int(tok[per * k]) - 1, int(tok[per * k + 1]) - 1
        v = 1.0 if field == "pattern" else float(tok[per * k + 2])
        if i == j:
            diag[i] = v
        elif i < j:
            rows.append(i); cols.append(j); vals.append(v)
        elif symmetry == "symmetric":
            rows.append(j); cols.append(i); vals.append(v)
    return {"n": n, "U": len(rows), "diag": diag, "upper": np.array(vals, dtype=np.float64),
            "rows": np.array(rows, dtype=np.int32), "cols": np.array(cols, dtype=np.int32), "symmetric": symmetry == "symmetric"}
