"""CPU restatement of ``misc/panostretch.pano_stretch`` (image half + corner half).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Plain numpy, float64
coordinates exactly as the reference (``misc/panostretch.py:6-25,81-117``); the
bilinear gather restates ``scipy.ndimage.map_coordinates(order=1, mode='wrap')``
-- SciPy's *legacy* wrap, whose period is ``len-1`` (SURVEY.md section 4 KAT 1) --
so the file has no SciPy dependency.  Pinned against the real reference + SciPy
by ``oracle/gen_golden.py``.
"""
import numpy as np


def uv_tri(w, h):
    """misc/panostretch.py:6-25 -- sin u, cos u per column; tan v per row (float64)."""
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    u = ((x + 0.5) / w - 0.5) * 2 * np.pi
    v = ((y + 0.5) / h - 0.5) * np.pi
    return np.sin(u), np.cos(u), np.tan(v)


def stretch_coords(h, w, kx, ky):
    """misc/panostretch.py:91-96 -> refy [H,W], refx [H,W] float64 source coordinates."""
    sin_u, cos_u, tan_v = uv_tri(w, h)
    u0 = np.arctan2(sin_u * kx / ky, cos_u)                       # [W]
    v0 = np.arctan(tan_v[:, None] * np.sin(u0)[None, :] / sin_u[None, :] * ky)   # [H,W]
    refx = (u0 / (2 * np.pi) + 0.5) * w - 0.5
    refy = (v0 / np.pi + 0.5) * h - 0.5
    return refy, np.broadcast_to(refx[None, :], (h, w))


def scipy_wrap(c, n):
    """SciPy legacy 'wrap' coordinate rule (period n-1), vectorised."""
    c = np.array(c, dtype=np.float64, copy=True)
    sz = float(n - 1)
    neg = c < 0
    c[neg] += sz * (np.floor(-c[neg] / sz) + 1)       # (npy_intp)(-in/sz) truncation == floor for >0
    big = c > sz
    c[big] -= sz * np.floor(c[big] / sz)
    return c


def bilinear_wrap(img2d, refy, refx):
    """order=1 spline == lerp between floor(c) and min(floor(c)+1, n-1); double accumulate."""
    h, w = img2d.shape
    cy = scipy_wrap(refy, h)
    cx = scipy_wrap(refx, w)
    y0 = np.floor(cy).astype(np.int64)
    x0 = np.floor(cx).astype(np.int64)
    ty = cy - y0
    tx = cx - x0
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    im = img2d.astype(np.float64)
    out = ((1 - ty) * (1 - tx) * im[y0, x0] + (1 - ty) * tx * im[y0, x1]
           + ty * (1 - tx) * im[y1, x0] + ty * tx * im[y1, x1])
    return out.astype(img2d.dtype)


def stretch_corners(corners, kx, ky, w, h):
    """misc/panostretch.py:104-115 (+ :28-41)."""
    corners = np.asarray(corners)
    cu0 = ((corners[:, 0] + 0.5) / w - 0.5) * 2 * np.pi
    cv0 = ((corners[:, 1] + 0.5) / h - 0.5) * np.pi
    cu = np.arctan2(np.sin(cu0) * ky / kx, np.cos(cu0))
    c2 = (np.sin(cu0) * ky) ** 2 + (np.cos(cu0) * kx) ** 2
    cv = np.arctan2(np.sin(cv0), np.cos(cv0) * np.sqrt(c2))
    cx = (cu / (2 * np.pi) + 0.5) * w - 0.5
    cy = (cv / np.pi + 0.5) * h - 0.5
    return np.stack([cx, cy], axis=-1)


def pano_stretch(img, corners, kx, ky, order=1):
    """Same signature / return as misc/panostretch.py:81-117 (order=1 only)."""
    if order != 1:
        raise NotImplementedError("oracle restates order=1 (the only order dataset.py:82 uses)")
    h, w = img.shape[:2]
    refy, refx = stretch_coords(h, w, kx, ky)
    out = np.stack([bilinear_wrap(img[..., i], refy, refx) for i in range(img.shape[-1])], axis=-1)
    return out, stretch_corners(corners, kx, ky, w, h)
