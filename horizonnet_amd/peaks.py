"""Corner-index extraction -- drop-in for reference ``inference.find_N_peaks`` (``inference.py:21-29``)
with the maximum filter + comparison done on the device (``hn_find_peaks``)."""
import numpy as np
import torch

from . import _lib


def find_peaks_batch(signal, r, min_v, apply_sigmoid=False):
    """signal: cuda float32 [B,n].  Returns (mask uint8 [B,n], prob float32 [B,n]) on the device."""
    if not (isinstance(signal, torch.Tensor) and signal.is_cuda):
        raise RuntimeError("find_peaks_batch needs a cuda/ROCm tensor (no CPU fallback)")
    signal = signal.contiguous().float()
    B, n = (int(v) for v in signal.shape)
    mask = torch.empty((B, n), dtype=torch.uint8, device=signal.device)
    prob = torch.empty((B, n), dtype=torch.float32, device=signal.device)
    lib = _lib.load()
    with torch.cuda.device(signal.device):
        _lib.check(lib.hn_find_peaks(_lib.ptr(signal), B, n, int(r), float(min_v), int(bool(apply_sigmoid)),
                                     _lib.ptr(mask), _lib.ptr(prob), _lib.stream_ptr(signal.device)), "hn_find_peaks")
    return mask, prob


def find_N_peaks(signal, r=29, min_v=0.05, N=None, device="cuda"):
    """numpy [n] in -> (indices, values), exactly the reference's return convention."""
    signal = np.asarray(signal)
    dev = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float32)).to(device)[None]
    mask, _ = find_peaks_batch(dev, r, min_v)
    pk_loc = np.where(mask[0].cpu().numpy() != 0)[0]
    if N is not None:
        order = np.argsort(-signal[pk_loc])
        pk_loc = pk_loc[order[:N]]
        pk_loc = pk_loc[np.argsort(pk_loc)]
    return pk_loc, signal[pk_loc]
