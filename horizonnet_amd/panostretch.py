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


def pano_stretch_batch(imgs, kx, ky, out=None):
    """imgs: cuda float32 [B,H,W,C] (HWC per image); kx, ky: length-B sequences.  Returns cuda [B,H,W,C]."""
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
