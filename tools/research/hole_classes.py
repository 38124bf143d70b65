"""SURVEY 8 f-1, the half VERDICT r3 asked for: which connected classes of invalid vectors have a Delaunay-linear fill
(PIVbackend.py:284-308: LinearNDInterpolator over the ring points, np.argwhere order) that does NOT depend on the
triangulation Qhull happens to pick -- i.e. that a device kernel may write down as fixed weights of ring values?

Method: embed one component of a class at a random position of a 48 x 48 field together with a few random far holes (the
"context": Qhull's tie-breaks on co-circular lattice points depend on the whole point set), read the interpolation WEIGHTS
of every hole cell of the component off SciPy (values = unit vectors of the ring points), and count the distinct weight
patterns per class over many contexts.  One pattern = triangulation-independent (fixed weights); more = Qhull decides.

    python tools/research/hole_classes.py [n_contexts]
"""
import sys

import numpy as np
from scipy.interpolate import LinearNDInterpolator

CLASSES = {
    "isolated":        [(0, 0)],
    "run2_h":          [(0, 0), (0, 1)],
    "run2_v":          [(0, 0), (1, 0)],
    "run3_h":          [(0, 0), (0, 1), (0, 2)],
    "run4_v":          [(0, 0), (1, 0), (2, 0), (3, 0)],
    "L3":              [(0, 0), (1, 0), (1, 1)],
    "L4":              [(0, 0), (1, 0), (2, 0), (2, 1)],
    "block2x2":        [(0, 0), (0, 1), (1, 0), (1, 1)],
    "T4":              [(0, 0), (0, 1), (0, 2), (1, 1)],
    "plus5":           [(0, 1), (1, 0), (1, 1), (1, 2), (2, 1)],
    "S4":              [(0, 0), (0, 1), (1, 1), (1, 2)],
    "diag_pair":       [(0, 0), (1, 1)],                      # two isolated holes touching at a corner (8-connected)
    "diag_triple":     [(0, 0), (1, 1), (2, 2)],
    "knight_pair":     [(0, 0), (1, 2)],                      # rings share no cell but the diamonds overlap
    "gap_pair_h":      [(0, 0), (0, 2)],                      # two holes with one valid cell between them
}


def ring_of(hole):
    dil = hole.copy()
    dil[1:, :] |= hole[:-1, :]
    dil[:-1, :] |= hole[1:, :]
    dil[:, 1:] |= hole[:, :-1]
    dil[:, :-1] |= hole[:, 1:]
    return dil & ~hole


def weights(hole, cells):
    """{cell: tuple of (ring offset relative to the cell, weight)} for the given hole cells, or None if Qhull refuses."""
    pts = np.argwhere(ring_of(hole))
    vals = np.eye(len(pts))
    try:
        w = LinearNDInterpolator(pts, vals)(np.array(cells, dtype=np.float64))
    except Exception:          # noqa: BLE001
        return None
    out = {}
    for k, c in enumerate(cells):
        if np.isnan(w[k]).any():
            out[c] = "nan"
            continue
        nz = np.flatnonzero(np.abs(w[k]) > 1e-12)
        out[c] = tuple(sorted(((int(pts[i][0] - c[0]), int(pts[i][1] - c[1])), round(float(w[k][i]), 9)) for i in nz))
    return out


def survey(n_ctx=400, size=48, seed=0):
    rng = np.random.default_rng(seed)
    report = {}
    for name, shape in CLASSES.items():
        pats = {i: set() for i in range(len(shape))}
        for _ in range(n_ctx):
            hole = np.zeros((size, size), bool)
            r0, c0 = rng.integers(6, size - 10, 2)
            cells = [(int(r0 + dr), int(c0 + dc)) for dr, dc in shape]
            for r, c in cells:
                hole[r, c] = True
            for _ in range(int(rng.integers(0, 6))):           # far context holes (at least 4 cells away from the component)
                r, c = rng.integers(2, size - 2, 2)
                if all(abs(r - rr) > 4 or abs(c - cc) > 4 for rr, cc in cells):
                    hole[r, c] = True
            w = weights(hole, cells)
            if w is None:
                continue
            for i, c in enumerate(cells):
                pats[i].add(w[c])
        report[name] = [len(p) for p in pats.values()]
    return report


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rep = survey(n)
    for name, counts in rep.items():
        verdict = "fixed weights" if max(counts) == 1 else "Qhull decides"
        print(f"{name:12s} distinct weight patterns per hole cell over {n} contexts: {counts}  -> {verdict}")
