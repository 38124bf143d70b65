"""Local fill rules that hold in EVERY Delaunay triangulation of the ring points.

A hole cell h (origin) gets the barycentric value of a triangle T of ring points that contains it.  If the closed
circumdisc of T holds no other ring point, T belongs to every Delaunay triangulation of the point set (the empty-circle
property, strict), so the value is fixed whatever Qhull does elsewhere -- a rule the device can apply from the class map
alone: "the vertices are ring cells and no other lattice cell of the closed disc is a ring cell".  Degenerate triangles
(h on the segment of two ring points) are the straight-run rule the device already had; its circle family is handled there.

This script enumerates the rules with small discs, prints them as a C table (vertex offsets, weights as fractions,
blocker offsets) and checks every rule against SciPy on random fields that satisfy its premise.

    python tools/research/fill_rules.py [max_r2] > torchpiv_amd/csrc/postval_rules.inc
"""
import itertools
import sys
from fractions import Fraction

import numpy as np


def circumcircle(p, q, r):
    (ax, ay), (bx, by), (cx, cy) = p, q, r
    d = 2 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by))
    if d == 0:
        return None
    ux = Fraction((ax * ax + ay * ay) * (by - cy) + (bx * bx + by * by) * (cy - ay) + (cx * cx + cy * cy) * (ay - by), d)
    uy = Fraction((ax * ax + ay * ay) * (cx - bx) + (bx * bx + by * by) * (ax - cx) + (cx * cx + cy * cy) * (bx - ax), d)
    r2 = (ax - ux) ** 2 + (ay - uy) ** 2
    return ux, uy, r2


def barycentric(p, q, r):
    """weights of the origin in triangle (p, q, r) as Fractions, or None if outside"""
    (ax, ay), (bx, by), (cx, cy) = p, q, r
    det = (by - cy) * (ax - cx) + (cx - bx) * (ay - cy)
    if det == 0:
        return None
    w0 = Fraction((by - cy) * (0 - cx) + (cx - bx) * (0 - cy), det)
    w1 = Fraction((cy - ay) * (0 - cx) + (ax - cx) * (0 - cy), det)
    w2 = 1 - w0 - w1
    if min(w0, w1, w2) < 0:
        return None
    return w0, w1, w2


def rules(max_r2=Fraction(5, 2), reach=3):
    pts = [(r, c) for r in range(-reach, reach + 1) for c in range(-reach, reach + 1) if (r, c) != (0, 0)]
    out = []
    for tri in itertools.combinations(pts, 3):
        w = barycentric(*tri)
        if w is None or min(w) == 0:           # origin outside, or on an edge (the straight-run rule's business)
            continue
        cc = circumcircle(*tri)
        if cc is None or cc[2] > max_r2:
            continue
        ux, uy, r2 = cc
        rad = int(float(r2) ** 0.5) + 2
        block = [(r, c) for r in range(int(ux) - rad, int(ux) + rad + 1) for c in range(int(uy) - rad, int(uy) + rad + 1)
                 if (r - ux) ** 2 + (c - uy) ** 2 <= r2 and (r, c) not in tri and (r, c) != (0, 0)]
        # the 4-neighbours of the origin that are not vertices must be holes for the premise to be satisfiable at all
        # (a valid neighbour of a hole is a ring point); keep the rule anyway, the premise says so
        out.append((tri, w, block, r2))
    out.sort(key=lambda t: (t[3], t[0]))
    return out


def check(rule, trials=200, size=24, seed=1):
    """fields that satisfy the premise (vertices ring, blockers not ring) + random far holes: SciPy must give the weights"""
    from scipy.interpolate import LinearNDInterpolator
    tri, w, block, _ = rule
    rng = np.random.default_rng(seed)
    done = 0
    for _ in range(trials * 20):
        if done >= trials:
            break
        hole = rng.random((size, size)) < rng.choice([0.0, 0.02, 0.06])
        o = (size // 2, size // 2)
        hole[o] = True
        for b in block:                          # blockers next to the origin would be ring points: make them holes;
            hole[o[0] + b[0], o[1] + b[1]] = True     # far ones too (simplest premise that is always satisfiable)
        for v in tri:
            hole[o[0] + v[0], o[1] + v[1]] = False
        dil = hole.copy()
        dil[1:, :] |= hole[:-1, :]; dil[:-1, :] |= hole[1:, :]; dil[:, 1:] |= hole[:, :-1]; dil[:, :-1] |= hole[:, 1:]
        ring = dil & ~hole
        if not all(ring[o[0] + v[0], o[1] + v[1]] for v in tri):
            continue
        if any(ring[o[0] + b[0], o[1] + b[1]] for b in block):
            continue
        pts = np.argwhere(ring)
        vals = rng.standard_normal(len(pts))
        try:
            got = float(LinearNDInterpolator(pts, vals)(np.array([o], dtype=np.float64))[0])
        except Exception:          # noqa: BLE001
            continue
        idx = {tuple(p): k for k, p in enumerate(pts)}
        want = sum(float(wi) * vals[idx[(o[0] + v[0], o[1] + v[1])]] for wi, v in zip(w, tri))
        if not abs(got - want) <= 1e-12 * (1 + abs(want)):
            return False, done
        done += 1
    return True, done


if __name__ == "__main__":
    max_r2 = Fraction(sys.argv[1]) if len(sys.argv) > 1 else Fraction(5, 2)
    rs = rules(max_r2)
    bad = 0
    for rule in rs:
        ok, n = check(rule)
        tri, w, block, r2 = rule
        print(tri, [str(x) for x in w], "blockers", block, "r2", r2, "->", "ok" if ok else "MISMATCH", n, file=sys.stderr)
        bad += not ok
    if bad:
        raise SystemExit(f"{bad} rule(s) disagree with SciPy")
    nb = max(len(r[2]) for r in rs)
    print(f"// Generated by tools/research/fill_rules.py {max_r2}: {len(rs)} local fill rules that hold in every Delaunay")
    print("// triangulation of the ring points (closed circumdisc of the triangle free of other ring points), each checked")
    print("// against scipy.interpolate.LinearNDInterpolator on 200 random fields.  Offsets are (row, column) from the hole cell.")
    print(f"constexpr int PV_N_RULES = {len(rs)}, PV_MAX_BLOCK = {nb};")
    print("struct PvRule { signed char v[3][2]; double w[3]; signed char nb; signed char b[PV_MAX_BLOCK][2]; };")
    print("__constant__ PvRule PV_RULES[PV_N_RULES] = {")
    for tri, w, block, r2 in rs:
        vs = ", ".join(f"{{{a}, {b}}}" for a, b in tri)
        ws = ", ".join(f"{x.numerator}.0 / {x.denominator}.0" for x in w)
        bs = ", ".join(f"{{{a}, {b}}}" for a, b in block + [(0, 0)] * (nb - len(block)))
        print(f"    {{{{{vs}}}, {{{ws}}}, {len(block)}, {{{bs}}}}},      // r^2 = {r2}")
    print("};")
