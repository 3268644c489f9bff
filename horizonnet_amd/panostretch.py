"""Pano-Stretch -- drop-in for reference ``misc/panostretch.pano_stretch`` (``:81-117``).

The image half (``:91-102``) is the HIP gather kernel behind ``hn_pano_stretch``; the corner
half (``:104-115``) is a closed form on <= ~30 points and stays on the host, in the same dtype
rules numpy applies in the reference (float32 corners stay float32).
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _stretch_corners(corners, kx, ky, w, h):
    corners = np.asarray(corners)
    lon = ((corners[:, 0] + 0.5) / w - 0.5) * 2 * np.pi
    lat = ((corners[:, 1] + 0.5) / h - 0.5) * np.pi
    lon2 = np.arctan2(np.sin(lon) * ky / kx, np.cos(lon))
    shrink = np.sqrt((np.sin(lon) * ky) ** 2 + (np.cos(lon) * kx) ** 2)
    lat2 = np.arctan2(np.sin(lat), np.cos(lat) * shrink)
    return np.stack([(lon2 / (2 * np.pi) + 0.5) * w - 0.5, (lat2 / np.pi + 0.5) * h - 0.5], axis=-1)


_TABLE_CACHE = {}


def stretch_tables(kx, ky, H, W):
    """The per-column / per-row terms of the warp exactly as the reference computes them (misc/panostretch.py:6-25,92,95),
    numpy float64: -> (col [B,3,W] = refx, sin(u0), sin(u); tan_v [H]; mirror_symmetric).  mirror_symmetric: column W-1-x is the
    exact mirror image of column x (true for even W with numpy's odd sin / arctan2), the precondition of the kernel that shares
    one arctangent between the four mirror images of a pixel."""
    x = np.arange(W).astype(np.float64)
    y = np.arange(H).astype(np.float64)
    u = ((x + 0.5) / W - 0.5) * 2 * np.pi
    v = ((y + 0.5) / H - 0.5) * np.pi
    sin_u, cos_u, tan_v = np.sin(u), np.cos(u), np.tan(v)
    kx = np.asarray(kx, np.float64)[:, None]
    ky = np.asarray(ky, np.float64)[:, None]
    u0 = np.arctan2(sin_u[None] * kx / ky, cos_u[None])
    sin_u0 = np.sin(u0)
    refx = (u0 / (2 * np.pi) + 0.5) * W - 0.5
    refx_m = (-u0 / (2 * np.pi) + 0.5) * W - 0.5                       # what the mirrored column must hold
    sym = bool(np.array_equal(sin_u[::-1], -sin_u) and np.array_equal(sin_u0[:, ::-1], -sin_u0)
               and np.array_equal(refx[:, ::-1], refx_m) and np.array_equal(tan_v[::-1], -tan_v))
    col = np.ascontiguousarray(np.stack([refx, sin_u0, np.broadcast_to(sin_u[None], refx.shape)], 1))
    return col, np.ascontiguousarray(tan_v), sym


def pano_stretch_batch(imgs, kx, ky, out=None, host_tables=True):
    """imgs: cuda float32 [B,H,W,C] (HWC per image); kx, ky: length-B sequences.  Returns cuda [B,H,W,C].
    host_tables (default): the column / row terms come from numpy (``stretch_tables``: the reference's own values, 24 KB per
    image uploaded with the call); False: the kernel computes them itself (device libm; within 1e-13 of the reference's,
    which only matters ON SciPy's wrap discontinuity -- column 0 when kx == ky)."""
    if not (isinstance(imgs, torch.Tensor) and imgs.is_cuda):
        raise RuntimeError("pano_stretch_batch needs a cuda/ROCm tensor (no CPU fallback)")
    if imgs.dim() != 4 or imgs.dtype != torch.float32:
        raise ValueError("imgs must be float32 [B,H,W,C]")
    imgs = imgs.contiguous()
    B, H, W, C = (int(v) for v in imgs.shape)
    kx = np.ascontiguousarray(np.broadcast_to(np.asarray(kx, np.float64), (B,)))
    ky = np.ascontiguousarray(np.broadcast_to(np.asarray(ky, np.float64), (B,)))
    if out is None:
        out = torch.empty_like(imgs)
    lib = _lib.load()
    dp = ctypes.POINTER(ctypes.c_double)
    with torch.cuda.device(imgs.device):
        if host_tables and B > 0:
            key = (kx.tobytes(), ky.tobytes(), H, W, imgs.device.index)
            if _TABLE_CACHE.get("key") != key:               # (a repeated call with the same factors re-uses the uploaded tables)
                col, tan_v, sym = stretch_tables(kx, ky, H, W)
                _TABLE_CACHE.update(key=key, col=torch.from_numpy(col).to(imgs.device), tan=torch.from_numpy(tan_v).to(imgs.device), sym=sym)
            d_col, d_tan, sym = _TABLE_CACHE["col"], _TABLE_CACHE["tan"], _TABLE_CACHE["sym"]
            _lib.check(lib.hn_pano_stretch_tables(_lib.ptr(imgs), _lib.ptr(out), kx.ctypes.data_as(dp), ky.ctypes.data_as(dp),
                                                  _lib.ptr(d_col), _lib.ptr(d_tan), int(sym), B, H, W, C,
                                                  _lib.stream_ptr(imgs.device)), "hn_pano_stretch_tables")
        else:
            _lib.check(lib.hn_pano_stretch(_lib.ptr(imgs), _lib.ptr(out), kx.ctypes.data_as(dp), ky.ctypes.data_as(dp),
                                           B, H, W, C, _lib.stream_ptr(imgs.device)), "hn_pano_stretch")
    return out


def pano_stretch(img, corners, kx, ky, order=1, device="cuda"):
    """Same signature and return types as the reference: numpy [H,W,C] in, numpy out."""
    if order != 1:
        raise NotImplementedError("only order=1 (the order dataset.py:82 uses) is implemented on the MI355X engine")
    img = np.asarray(img)
    src = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(device)
    out = pano_stretch_batch(src[None], [float(kx)], [float(ky)])[0]
    stretched = out.cpu().numpy().astype(img.dtype, copy=False)
    return stretched, _stretch_corners(corners, kx, ky, img.shape[1], img.shape[0])
