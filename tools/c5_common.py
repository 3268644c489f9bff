"""Shared pieces of the BASELINE configs[4] ("config 5") measurement: the seeded synthetic Structured3D-shaped
rooms, and the compact codec of the briefly-trained checkpoint they are evaluated with.

TEST / MEASUREMENT INFRASTRUCTURE (no reference restatement here, nothing imported from ``oracle/``).

Why a codec: there is no dataset and no checkpoint offline.  The weights are trained on the MI355X engine
(``tools/c5_train_export.py``), and both sides of the parity check need the SAME weights: the unmodified reference
``inference()`` in the build container (``oracle/gen_config5.py``) and the engine on the GPU box.  gpurun returns at
most 64 MiB per call and the float32 state_dict is 326 MB, so every >= 2-D tensor is stored as per-output-channel
symmetric integers (4 bits for the tensors above 4 M elements, 6 bits otherwise) and every 1-D tensor as float32; the
config-5 model is DEFINED as the de-quantised tensors (BatchNorm statistics re-estimated and the 1-D parameters
fine-tuned after quantisation, so the stored model is the evaluated model, bit for bit).
"""
import io
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ST3D_MIX = [4] * 63 + [6] * 17 + [8] * 8 + [10] * 12        # README_ST3D.md:49-56 corner-count mix
CKPT_PATH = os.path.join(ROOT, "tests", "golden", "config5", "ckpt_q.npz")
VAL_SEED0 = 50_000                                           # room i of the 1000-panorama set = seed VAL_SEED0 + i
TRAIN_SEED0 = 10_000


def usable_cores():
    """CPUs this process may really use: the affinity mask AND the cgroup quota (the MI355X box shows 256 logical CPUs
    and grants 16)."""
    import math
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.floor(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return n


def bits_for(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return 4 if n > 4_000_000 else 6


def quantize_tensor(w, bits):
    """w: float32 array [out, ...] -> (q int8 same shape, scale float32 [out]); w ~= q * scale[:, None...]."""
    w = np.asarray(w, np.float32)
    flat = w.reshape(w.shape[0], -1)
    qmax = (1 << (bits - 1)) - 1
    amax = np.abs(flat).max(1)
    scale = np.where(amax > 0, amax / qmax, 1.0).astype(np.float32)
    q = np.clip(np.rint(flat / scale[:, None]), -qmax, qmax).astype(np.int8)
    return q.reshape(w.shape), scale


def dequantize_tensor(q, scale):
    q = np.asarray(q)
    return (q.reshape(q.shape[0], -1).astype(np.float32) * scale[:, None].astype(np.float32)).reshape(q.shape)


def encode_state_dict(sd):
    """state_dict (name -> torch tensor) -> dict of numpy arrays for ``np.savez_compressed``."""
    out = {}
    for k, v in sd.items():
        a = v.detach().cpu().numpy()
        if a.ndim >= 2:
            q, s = quantize_tensor(a, bits_for(a.shape))
            out["q/" + k] = q
            out["s/" + k] = s
        else:
            out["f/" + k] = a
    return out


def decode_state_dict(path=CKPT_PATH):
    """-> ordered dict name -> torch tensor (float32 / int64 as stored)."""
    import torch
    z = np.load(path)
    names = [str(n) for n in z["names"]]
    sd = {}
    for k in names:
        if "q/" + k in z.files:
            sd[k] = torch.from_numpy(dequantize_tensor(z["q/" + k], z["s/" + k]))
        else:
            sd[k] = torch.from_numpy(np.array(z["f/" + k]))
    return sd


def save_checkpoint(sd, path=CKPT_PATH):
    enc = encode_state_dict(sd)
    enc["names"] = np.array(list(sd.keys()))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **enc)
    return os.path.getsize(path)


# ---- rooms --------------------------------------------------------------------------------------------------------
def make_room(args):
    from tools import synth_rooms as sr
    seed, n_corners = args
    rng = np.random.RandomState(seed)
    for _ in range(1000):                                   # bounded: ~10-20 % of notched rooms hide a corner
        poly = sr.manhattan_polygon(rng, n_corners)
        zc, zf = rng.uniform(1.0, 1.6), rng.uniform(1.2, 1.7)
        cor = sr.room_corners(poly, zc, zf)
        if np.all(np.diff(cor[::2, 0]) > 10) and cor[0, 0] + 1024 - cor[-2, 0] > 10:    # every corner visible, none merged
            return sr.render_image(poly, zc, zf, rng), cor
    raise RuntimeError("no fully visible room found")


def room_jobs(n, seed0, first=0, mix=ST3D_MIX):
    return [(seed0 + i, mix[(i * 37) % len(mix)]) for i in range(first, first + n)]


def make_rooms(n, seed0, pool=None, first=0, mix=ST3D_MIX):
    """-> (uint8 [n,512,1024,3], list of label_cor arrays).  Room i depends only on (seed0 + i) (and the corner mix;
    mix=[4] gives PanoContext-shaped cuboids)."""
    jobs = room_jobs(n, seed0, first, mix)
    res = pool.map(make_room, jobs) if pool is not None else [make_room(j) for j in jobs]
    return np.stack([r[0] for r in res]), [r[1] for r in res]


def image_crc(img):
    """Checksum of a rendered panorama: lets the GPU box prove it rendered the same pixels as the build container."""
    return zlib.crc32(np.ascontiguousarray(img).tobytes()) & 0xFFFFFFFF
