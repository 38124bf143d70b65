"""Which diagonal does SciPy/Qhull choose for the co-circular diamond around an isolated invalid vector?
(PIVbackend.py:284-308: LinearNDInterpolator over the ring points in np.argwhere order.)"""
import numpy as np
from scipy.spatial import Delaunay


def ring_points(hole):
    inv = hole
    dil = inv.copy()
    dil[1:, :] |= inv[:-1, :]; dil[:-1, :] |= inv[1:, :]; dil[:, 1:] |= inv[:, :-1]; dil[:, :-1] |= inv[:, 1:]
    return np.argwhere(dil & ~inv)


def diagonals(hole):
    """for every isolated hole (all 4 neighbours valid, and they are ring cells): 'NS' or 'EW'"""
    pts = ring_points(hole)
    tri = Delaunay(pts)
    index = {tuple(p): i for i, p in enumerate(pts)}
    edges = set()
    for s in tri.simplices:
        for a in range(3):
            for b in range(a + 1, 3):
                edges.add((min(s[a], s[b]), max(s[a], s[b])))
    out = {}
    for r, c in np.argwhere(hole):
        nb = [(r - 1, c), (r + 1, c), (r, c - 1), (r, c + 1)]
        if not all(n in index for n in nb):
            continue
        n, s_, w, e = (index[t] for t in nb)
        ns = (min(n, s_), max(n, s_)) in edges
        ew = (min(w, e), max(w, e)) in edges
        out[(r, c)] = "NS" if ns and not ew else ("EW" if ew and not ns else "??")
    return out, pts, tri


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    nr, nc = 63, 63
    # one isolated hole at various positions
    res = {}
    for r in range(2, 12):
        for c in range(2, 12):
            h = np.zeros((nr, nc), bool); h[r, c] = True
            d, _, _ = diagonals(h)
            res[(r, c)] = d[(r, c)]
    print("single hole:", set(res.values()))
    # two holes
    stats = {}
    for trial in range(300):
        h = np.zeros((nr, nc), bool)
        k = rng.integers(2, 8)
        cells = set()
        while len(cells) < k:
            r, c = rng.integers(2, nr - 2), rng.integers(2, nc - 2)
            if all(abs(r - r2) > 2 or abs(c - c2) > 2 for r2, c2 in cells):
                cells.add((r, c))
        for r, c in cells:
            h[r, c] = True
        d, pts, tri = diagonals(h)
        order = sorted(cells)            # argwhere order of the holes
        for i, cell in enumerate(order):
            stats.setdefault((k, i), []).append(d[cell])
    for key in sorted(stats):
        v = stats[key]
        print(key, {t: v.count(t) for t in set(v)})
