"""The triangulation step of the reference's fillMissingValues (PIVbackend.py:300-302) as a pure function of
small arrays.  This module imports numpy and scipy ONLY: it is the target of OfflinePIV's worker processes
(`fill_workers`), which therefore never import torch, never load libtorchpiv_hip.so and never touch the GPU.
(The workers are started with the `spawn` method, which re-imports the caller's `__main__`: scripts that use
`fill_workers > 0` need the usual `if __name__ == "__main__":` guard.)"""
import numpy as np  # noqa: F401


def qhull_fill(points, values, targets):
    """Delaunay-linear interpolation of `values` [n, k] given at integer `points` [n, 2], evaluated at
    `targets` [m, 2] (scipy's LinearNDInterpolator = Qhull, as the reference calls it).  None when Qhull
    refuses the points (the reference's bare `except` then drops the pair)."""
    from scipy.interpolate import LinearNDInterpolator
    try:
        return LinearNDInterpolator(points, values)(targets)
    except Exception:
        return None
