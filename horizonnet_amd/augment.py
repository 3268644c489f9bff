"""Host-side stretch-parameter sampling -- reference ``dataset.py:70-81`` (+ ``cor2xybound`` ``:189-208``).

Scalar work per image (microseconds): it stays on the CPU and feeds the per-image (kx, ky) of
``pano_stretch_batch``.  Corners are image-space (x, y) rows alternating ceiling / floor like the
reference's ``label_cor`` files."""
import numpy as np


def cor2xybound(cor, w=1024, h=512):
    """Room extent used to clamp the stretch factors: (min|x|, min|y|, max|x|, max|y|) of the wall corners
    in a frame where the room height is normalised to 3 (ceiling plane fixed 50 units above the camera)."""
    cor = np.asarray(cor)
    top, bottom = cor[0::2], cor[1::2]
    z_top = -50
    lon = ((top[:, 0] + 0.5) / w - 0.5) * 2 * np.pi
    lat_top = ((top[:, 1] + 0.5) / h - 0.5) * np.pi
    lat_bot = ((bottom[:, 1] + 0.5) / h - 0.5) * np.pi
    r = z_top / np.tan(lat_top)                      # signed horizontal range of each ceiling corner
    x, y = r * np.cos(lon), r * np.sin(lon)
    z_bot = np.sqrt(x ** 2 + y ** 2) * np.tan(lat_bot)
    scale = 3 / abs(z_bot.mean() - z_top)
    ex = sorted((abs(x.min() * scale), abs(x.max() * scale)))
    ey = sorted((abs(y.min() * scale), abs(y.max() * scale)))
    return ex[0], ey[0], ex[1], ey[1]


def sample_stretch(cor, max_stretch=2.0, rng=np.random):
    """kx, ky ~ U(1, max_stretch), each inverted with probability 1/2, clamped so the stretched room
    keeps a sane size (no wall closer than 0.5 / farther than 10 camera heights)."""
    xmin, ymin, xmax, ymax = cor2xybound(cor)
    kx = rng.uniform(1.0, max_stretch)
    ky = rng.uniform(1.0, max_stretch)
    if rng.randint(2) == 0:
        kx = max(1.0 / kx, min(0.5 / xmin, 1.0))
    else:
        kx = min(kx, max(10.0 / xmax, 1.0))
    if rng.randint(2) == 0:
        ky = max(1.0 / ky, min(0.5 / ymin, 1.0))
    else:
        ky = min(ky, max(10.0 / ymax, 1.0))
    return kx, ky
