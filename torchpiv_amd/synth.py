"""Deterministic synthetic particle-image pairs.

Used by bench.py (generated on the GPU so decode/H2D stay outside the timed
region) and by the tests.  Particle positions are displaced by a smooth field
(frame b = particles of frame a moved by (dx, dy)), not by image warping, so
both frames are band-limited alike.
"""
from __future__ import annotations

import math

import torch

BASE_SEED = 1234


def flow_field(px: torch.Tensor, py: torch.Tensor, H: int, W: int, kind: str = "wavy"):
    """Displacement (dx, dy) in pixels at particle positions (px, py)."""
    if kind == "wavy":      # the bench field: |d| <= ~6 px
        dx = 4.0 * torch.sin(2 * math.pi * py / H) + 1.3
        dy = 3.0 * torch.cos(2 * math.pi * px / W) - 0.7
    elif kind == "uniform":
        dx = torch.full_like(px, 2.3)
        dy = torch.full_like(py, -1.6)
    elif kind == "shear":
        dx = 5.0 * (py / H) - 1.0
        dy = torch.full_like(py, 0.4)
    elif kind == "vortex":
        cx, cy = W / 2.0, H / 2.0
        r2 = (px - cx) ** 2 + (py - cy) ** 2
        s = 6.0 * torch.exp(-r2 / (2 * (0.25 * min(H, W)) ** 2)) / (0.25 * min(H, W))
        dx = -(py - cy) * s
        dy = (px - cx) * s
    elif kind == "zero":
        dx = torch.zeros_like(px)
        dy = torch.zeros_like(py)
    else:
        raise KeyError(kind)
    return dx, dy


def _render(px, py, amp, H, W, sigma, device):
    img = torch.zeros(H * W, dtype=torch.float32, device=device)
    cx = torch.round(px)
    cy = torch.round(py)
    for oy in range(-3, 4):
        yy = cy + oy
        wy = torch.exp(-((yy - py) ** 2) / (2 * sigma * sigma))
        oky = (yy >= 0) & (yy < H)
        for ox in range(-3, 4):
            xx = cx + ox
            w = wy * torch.exp(-((xx - px) ** 2) / (2 * sigma * sigma))
            ok = oky & (xx >= 0) & (xx < W)
            idx = yy.long() * W + xx.long()          # (in float32 the flat index is inexact beyond 2^24 pixels)
            img.index_add_(0, idx[ok], (amp * w)[ok])
    return img.view(H, W)


def make_pair(H: int, W: int, index: int = 0, kind: str = "wavy", density: float = 0.03,
              sigma: float = 1.0, noise: float = 0.0, offset: float = 8.0,
              device="cpu"):
    """One (frame_a, frame_b) pair of uint8 [H, W] tensors on `device`.
    Random draws are made on the CPU generator (seed = 1234 + index) so that a
    pair is identical whichever device renders it (up to float32 rounding of
    the accumulation order)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(BASE_SEED + int(index))
    pad = 10.0
    n = int(density * (H + 2 * pad) * (W + 2 * pad))
    px = (torch.rand(n, generator=g, dtype=torch.float64) * (W + 2 * pad) - pad).float()
    py = (torch.rand(n, generator=g, dtype=torch.float64) * (H + 2 * pad) - pad).float()
    amp = (0.5 + 0.5 * torch.rand(n, generator=g, dtype=torch.float64)).float() * 200.0
    na = torch.randn(H, W, generator=g) * noise if noise > 0 else None
    nb = torch.randn(H, W, generator=g) * noise if noise > 0 else None
    px, py, amp = px.to(device), py.to(device), amp.to(device)
    dx, dy = flow_field(px, py, H, W, kind)
    a = _render(px, py, amp, H, W, sigma, device) + offset
    b = _render(px + dx, py + dy, amp, H, W, sigma, device) + offset
    if na is not None:
        a = a + na.to(device)
        b = b + nb.to(device)
    a = a.round().clamp_(0, 255).to(torch.uint8)
    b = b.round().clamp_(0, 255).to(torch.uint8)
    return a, b


def make_batch(n_pairs: int, H: int, W: int, first_index: int = 0, kind: str = "wavy",
               noise: float = 0.0, device="cpu"):
    """uint8 tensors A, B of shape [n_pairs, H, W] resident on `device`."""
    A = torch.empty(n_pairs, H, W, dtype=torch.uint8, device=device)
    B = torch.empty(n_pairs, H, W, dtype=torch.uint8, device=device)
    for i in range(n_pairs):
        A[i], B[i] = make_pair(H, W, first_index + i, kind=kind, noise=noise, device=device)
    return A, B
