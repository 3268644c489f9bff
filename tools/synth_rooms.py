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
            out += [a, b, c] if i in (0, 2) else [c, b, a]
        else:
            out.append((px, py))
    return np.array(out, np.float64)


def cast(poly):
    """Per image column: horizontal distance to the first wall hit and the index of that wall (edge i -> i+1)."""
    u = ((np.arange(W) + 0.5) / W - 0.5) * 2 * np.pi
    d = np.stack([np.sin(u), -np.cos(u)], 1)                       # ray directions on the floor plan
    dist = np.full(W, np.inf)
    wall = np.zeros(W, np.int64)
    n = len(poly)
    for i in range(n):
        a, b = poly[i], poly[(i + 1) % n]
        e = b - a
        den = d[:, 0] * e[1] - d[:, 1] * e[0]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (a[0] * e[1] - a[1] * e[0]) / den                  # ray parameter
            s = (a[0] * d[:, 1] - a[1] * d[:, 0]) / den            # edge parameter
        ok = (np.abs(den) > 1e-12) & (t > 0) & (s >= -1e-9) & (s <= 1 + 1e-9) & (t < dist)
        dist = np.where(ok, t, dist)
        wall = np.where(ok, i, wall)
    return dist, wall


def render(poly, z_ceil=1.2, z_floor=1.5, noise=0.0, rng=None):
    """-> (bon [2,1024] rows of ceiling / floor boundary, cor [1024] corner probability, corner columns)."""
    dist, _ = cast(poly)
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


def room_corners(poly, z_ceil=1.2, z_floor=1.5):
    """Ground-truth ``label_cor`` rows of the reference's dataset format (README_PREPARE_DATASET.md:3-14): for every
    polygon vertex (counter-clockwise = increasing longitude) the ceiling then the floor image point, float32 [2N,2],
    starting at the smallest column."""
    ang = np.arctan2(poly[:, 0], -poly[:, 1])
    col = (ang / (2 * np.pi) + 0.5) * W - 0.5
    rng_ = np.sqrt((poly ** 2).sum(1))
    row_c = (-np.arctan2(z_ceil, rng_) / np.pi + 0.5) * H - 0.5
    row_f = (-np.arctan2(-z_floor, rng_) / np.pi + 0.5) * H - 0.5
    cor = np.stack([np.repeat(col, 2), np.stack([row_c, row_f], 1).reshape(-1)], 1)
    return np.roll(cor, -2 * int(np.argmin(col)), axis=0).astype(np.float32)


def render_image(poly, z_ceil=1.2, z_floor=1.5, rng=None, noise=6.0):
    """A synthetic equirectangular RGB panorama [512,1024,3] uint8 of the room seen from the origin: flat-shaded walls
    (one random colour each, darkened with grazing incidence and distance), chequered floor, plain ceiling, sensor
    noise.  Crude, but its edges are exactly the layout's boundaries and corners."""
    rng = rng or np.random
    dist, wall = cast(poly)
    n = len(poly)
    u = ((np.arange(W) + 0.5) / W - 0.5) * 2 * np.pi
    v = -((np.arange(H) + 0.5) / H - 0.5) * np.pi                  # latitude, positive up
    tanv = np.tan(v)[:, None]
    ceil_mask = tanv > (z_ceil / dist)[None, :]
    floor_mask = tanv < (-z_floor / dist)[None, :]
    # walls
    e = np.roll(poly, -1, axis=0) - poly
    nrm = np.stack([e[:, 1], -e[:, 0]], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ray = np.stack([np.sin(u), -np.cos(u)], 1)
    incidence = np.abs((ray * nrm[wall]).sum(1))
    base = rng.uniform(70, 200, (n, 3))
    shade = (0.45 + 0.55 * incidence) * np.clip(1.6 / (0.6 + 0.35 * dist), 0.5, 1.2)
    img = np.broadcast_to((base[wall] * shade[:, None])[None], (H, W, 3)).copy()
    img *= (1.0 - 0.12 * np.abs(np.tan(v))[:, None, None] * 0 + 0.08 * np.sin(v)[:, None, None])
    # floor: chequer in plan coordinates; ceiling: plain with a soft falloff
    with np.errstate(divide="ignore"):
        rf = np.where(floor_mask, -z_floor / np.minimum(tanv, -1e-6), 0.0)
        rc = np.where(ceil_mask, z_ceil / np.maximum(tanv, 1e-6), 0.0)
    px, py = rf * np.sin(u)[None, :], -rf * np.cos(u)[None, :]
    chk = ((np.floor(px / 0.5) + np.floor(py / 0.5)) % 2)
    fcol = rng.uniform(60, 140, 3)
    floor_rgb = fcol[None, None, :] * (0.75 + 0.25 * chk)[..., None] * np.clip(1.4 / (0.7 + 0.3 * rf), 0.5, 1.2)[..., None]
    ccol = rng.uniform(170, 235, 3)
    ceil_rgb = ccol[None, None, :] * np.clip(1.3 / (0.8 + 0.15 * rc), 0.7, 1.05)[..., None]
    img = np.where(floor_mask[..., None], floor_rgb, img)
    img = np.where(ceil_mask[..., None], ceil_rgb, img)
    img += rng.normal(0, noise, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def write_dataset(root, n, seed, corner_mix=(4, 4, 4, 4, 4, 6, 6, 8), visible_only=True, noise=6.0):
    """Write n synthetic samples in the reference's on-disk layout (root/img/*.png, root/label_cor/*.txt)."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    os.makedirs(os.path.join(root, "label_cor"), exist_ok=True)
    k = 0
    while k < n:
        poly = manhattan_polygon(rng, corner_mix[k % len(corner_mix)])
        zc, zf = rng.uniform(1.0, 1.6), rng.uniform(1.2, 1.7)
        cor = room_corners(poly, zc, zf)
        if visible_only and np.any(np.diff(cor[::2, 0]) <= 0):
            continue                                               # a reflex corner hides part of the room
        Image.fromarray(render_image(poly, zc, zf, rng, noise)).save(os.path.join(root, "img", "room_%05d.png" % k))
        with open(os.path.join(root, "label_cor", "room_%05d.txt" % k), "w") as f:
            for x, y in cor:
                f.write("%d %d\n" % (round(float(x)), round(float(y))))
        k += 1
