"""Synthetic Manhattan rooms rendered to HorizonNet's 1-D signals (TEST INFRASTRUCTURE ONLY).

There is no dataset offline; the post-processing parity tests need structured inputs:
ceiling / floor boundary rows per image column and a wall-wall corner probability, as the network would emit
for a camera inside an axis-aligned (Manhattan) room.  Geometry follows the reference's conventions
(misc/post_proc.py:22-66): u = ((x+0.5)/W-0.5)*2pi is the column's longitude, the floor-plan direction of
column u is (sin u, -cos u), rows map to latitude v = -((y+0.5)/H-0.5)*pi, ceiling at height z0 above the camera.
"""
import numpy as np

W, H = 1024, 512


def manhattan_polygon(rng, n_corners):
    """Axis-aligned simple polygon (counter-clockwise, metres) containing the origin: a rectangle with
    (n_corners - 4) / 2 rectangular notches cut from its corners."""
    x0, x1 = -rng.uniform(1.5, 4.0), rng.uniform(1.5, 4.0)
    y0, y1 = -rng.uniform(1.5, 4.0), rng.uniform(1.5, 4.0)
    pts = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
    notches = (n_corners - 4) // 2
    corners_used = rng.permutation(4)[:notches]
    out = []
    for i, (px, py) in enumerate(pts):
        if i in corners_used:
            sx = 1.0 if px > 0 else -1.0
            sy = 1.0 if py > 0 else -1.0
            dx, dy = rng.uniform(0.4, 0.45 * abs(px)), rng.uniform(0.4, 0.45 * abs(py))
            a, b, c = (px, py - sy * dy), (px - sx * dx, py - sy * dy), (px - sx * dx, py)
            # keep counter-clockwise order around the notch
            out += [a, b, c] if i in (1, 3) else [c, b, a]
        else:
            out.append((px, py))
    return np.array(out, np.float64)


def render(poly, z_ceil=1.2, z_floor=1.5, noise=0.0, rng=None):
    """-> (bon [2,1024] rows of ceiling / floor boundary, cor [1024] corner probability, corner columns)."""
    u = ((np.arange(W) + 0.5) / W - 0.5) * 2 * np.pi
    d = np.stack([np.sin(u), -np.cos(u)], 1)                       # ray directions on the floor plan
    dist = np.full(W, np.inf)
    n = len(poly)
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        e = b - a
        den = d[:, 0] * e[1] - d[:, 1] * e[0]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (a[0] * e[1] - a[1] * e[0]) / den                  # ray parameter
            s = (a[0] * d[:, 1] - a[1] * d[:, 0]) / den            # edge parameter
        ok = (np.abs(den) > 1e-12) & (t > 0) & (s >= -1e-9) & (s <= 1 + 1e-9)
        dist = np.where(ok & (t < dist), t, dist)
    v_c = np.arctan2(z_ceil, dist)
    v_f = np.arctan2(-z_floor, dist)
    bon = np.stack([(-v_c / np.pi + 0.5) * H - 0.5, (-v_f / np.pi + 0.5) * H - 0.5])
    ang = np.arctan2(poly[:, 0], -poly[:, 1])
    cols = np.sort(((ang / (2 * np.pi) + 0.5) * W - 0.5) % W)
    x = np.arange(W)[:, None]
    dcol = np.abs(x - cols[None, :])
    dcol = np.minimum(dcol, W - dcol).min(1)
    cor = 0.96 ** dcol                                              # the training label of dataset.py:108-120
    if noise > 0:
        rng = rng or np.random
        bon = bon + rng.normal(0, noise, bon.shape) * np.array([[1.0], [1.0]])
        cor = np.clip(cor + rng.normal(0, noise * 0.02, cor.shape), 0, 1)
    return bon.astype(np.float32), cor.astype(np.float32), cols


def drop_corner(cor, cols, k):
    """Erase the k-th corner's probability bump (a missed wall-wall corner)."""
    cor = cor.copy()
    x = np.arange(W)
    d = np.abs(x - cols[k])
    cor[np.minimum(d, W - d) < 40] *= 0.02
    return cor


def add_corner(cor, col, height=0.6):
    """Add a spurious probability bump at column col (a false wall-wall corner)."""
    x = np.arange(W)
    d = np.abs(x - col)
    return np.maximum(cor, height * 0.96 ** np.minimum(d, W - d)).astype(np.float32)


def encode_image(bon_rows, cor):
    """Pack the signals into a [1,3,512,1024] float32 'panorama' that `SignalNet` decodes: channel 0 / 1 carry the
    ceiling / floor row as (row+0.5)/H, channel 2 the corner probability (every image row identical)."""
    import torch
    x = np.empty((1, 3, H, W), np.float32)
    x[0, 0] = ((bon_rows[0] + 0.5) / H)[None, :]
    x[0, 1] = ((bon_rows[1] + 0.5) / H)[None, :]
    x[0, 2] = cor[None, :]
    return torch.from_numpy(x)


class SignalNet:
    """Stand-in network for testing ``inference()`` end to end without trained weights: reads the signals back from
    image row 0 with exactly-rounded elementwise arithmetic only (identical on CPU and GPU up to the caller's sigmoid)
    and is equivariant to horizontal flips / rolls like the real model's ideal behaviour."""

    def __call__(self, x):
        import torch
        bon = (x[:, :2, 0, :] - 0.5) * np.float32(np.pi)
        p = x[:, 2:3, 0, :].clamp(1e-4, 1 - 1e-4)
        return bon, torch.log(p) - torch.log1p(-p)
