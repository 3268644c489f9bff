"""Panorama -> room-layout corners: drop-in for reference ``inference.inference`` (``inference.py:65-141``).

The network forward (and the test-time flip / rotate augmentation around it) runs on the device through the HIP engine,
the corner-peak detection runs on the device (``hn_find_peaks``), the Manhattan fit (``postproc``) on the host as in the
reference.  ``inference_batch`` is the MI355X-shaped entry: B panoramas x A augmentations go through ONE engine forward.
"""
import sys

import numpy as np
import torch

from . import postproc
from .peaks import find_N_peaks as _hip_find_N_peaks


def augment(x_img, flip, rotate):
    """[B,3,H,W] -> ([B*(1+flip+len(rotate)),3,H,W], tags).  Same stacking order and tags as inference.py:32-43;
    works on the tensor where it lives (host or device) instead of a numpy round trip."""
    tags = [""]
    views = [x_img]
    if flip:
        tags.append("flip")
        views.append(torch.flip(x_img, dims=[-1]))
    for frac in rotate:
        shift = int(round(frac * x_img.shape[-1]))
        tags.append("rotate %d" % shift)
        views.append(torch.roll(x_img, shift, dims=-1))
    return torch.cat(views, 0).float(), tags


def augment_undo(y, tags):
    """Inverse of `augment` on a network output [len(tags)*B, C, W] -> numpy [len(tags), B, C, W] (inference.py:46-62)."""
    sz = y.shape[0] // len(tags)
    out = []
    for i, tag in enumerate(tags):
        part = y[i * sz:(i + 1) * sz]
        if tag == "flip":
            part = torch.flip(part, dims=[-1])
        elif tag.startswith("rotate"):
            part = torch.roll(part, -int(tag.split()[-1]), dims=-1)
        elif tag != "":
            raise NotImplementedError()
        out.append(part)
    return torch.stack(out, 0).cpu().numpy()


def visualize_a_data(x, y_bon, y_cor):
    """Raw-output strip (dataset.py:211-229): corner probability band, white rule, dimmed panorama with the two
    boundaries in green."""
    img = (x.cpu().numpy().transpose([1, 2, 0]) * 255).astype(np.uint8)
    rows = ((np.asarray(y_bon) / np.pi + 0.5) * img.shape[0]).round().astype(int)
    band = np.zeros((30, img.shape[1], 3), np.uint8)
    band[:] = np.asarray(y_cor)[0][None, :, None] * 255
    rule = np.zeros((3, img.shape[1], 3), np.uint8) + 255
    dim = (img.copy() * 0.5).astype(np.uint8)
    cols = np.arange(rows.shape[1])
    dim[rows[0], cols, 1] = 255
    dim[rows[1], cols, 1] = 255
    return np.concatenate([band, rule, dim], 0)


def layout_from_signals(y_bon, y_cor, H=512, W=1024, force_cuboid=False, force_raw=False, min_v=None, r=0.05,
                        peaks_fn=None):
    """The host half of inference.py:89-141.  y_bon [2,W] float32 latitudes in radians (ceiling, floor), y_cor [W]
    float32 corner probability -> (cor_id [2N,2] float32 normalised (x,y), z0, z1)."""
    if peaks_fn is None:
        peaks_fn = _hip_find_N_peaks
    y_bon = (y_bon / np.pi + 0.5) * H - 0.5
    y_bon[0] = np.clip(y_bon[0], 1, H / 2 - 1)
    y_bon[1] = np.clip(y_bon[1], H / 2 + 1, H - 2)

    z0 = 50
    _, z1 = postproc.refine_by_fix_z(y_bon[0], y_bon[1], z0)

    if force_raw:
        cor = np.stack([np.arange(1024), y_bon[0]], 1)
    else:
        if min_v is None:
            min_v = 0 if force_cuboid else 0.05
        r = int(round(W * r / 2))
        N = 4 if force_cuboid else None
        peaks = peaks_fn(y_cor, r=r, min_v=min_v, N=N)[0]
        tol = abs(0.16 * z1 / 1.6)
        cor, walls = postproc.gen_ww(peaks, y_bon[0], z0, tol=tol, force_cuboid=force_cuboid)
        if not force_cuboid:
            plan = np.zeros((len(walls), 2), np.float32)
            for i in range(len(walls)):
                plan[i, walls[i]["type"]] = walls[i]["val"]
                plan[i, walls[i - 1]["type"]] = walls[i - 1]["val"]
            if not postproc.polygon_is_simple(plan):
                print("Fail to generate valid general layout!! Generate cuboid as fallback.", file=sys.stderr)
                peaks = peaks_fn(y_cor, r=r, min_v=0, N=4)[0]
                cor, walls = postproc.gen_ww(peaks, y_bon[0], z0, tol=tol, force_cuboid=True)

    cor = np.hstack([cor, postproc.infer_coory(cor[:, 1], z1 - z0, z0)[:, None]])
    cor_id = np.zeros((len(cor) * 2, 2), np.float32)
    cor_id[0::2, 0] = cor[:, 0]
    cor_id[0::2, 1] = cor[:, 1]
    cor_id[1::2, 0] = cor[:, 0]
    cor_id[1::2, 1] = cor[:, 2]
    cor_id[:, 0] /= W
    cor_id[:, 1] /= H
    return cor_id, z0, z1


_TABLES = {}


def _trig_tables(W):
    """np.sin / np.cos of the longitude of every image column (post_proc.py:22-23), plus entry W: of u = -1, the default the
    reference's inferred walls carry (post_proc.py:251-262).  numpy's values, so that hn_layout_fit_batch reproduces them."""
    if W not in _TABLES:
        u = np.concatenate([postproc.col_to_u(np.arange(W), W), [-1.0]])
        _TABLES[W] = (np.ascontiguousarray(np.sin(u)), np.ascontiguousarray(np.cos(u)))
    return _TABLES[W]


_QUARTILE_CHECKED = {"ok": None}


def _quartiles_f32(zz):
    """np.percentile(zz, 25, axis=1), np.percentile(zz, 75, axis=1) for float32 rows of length 1024 from ONE partition:
    numpy's 'linear' method puts the quartiles at virtual indices 255.75 / 767.25 and interpolates a + (b - a) t for t < 0.5,
    b - (b - a)(1 - t) otherwise, in the array's own type.  The closed form is checked against np.percentile on the first
    batch it sees; if numpy ever changes its formula the function falls back to np.percentile for good."""
    if zz.shape[1] != 1024 or zz.dtype != np.float32 or _QUARTILE_CHECKED["ok"] is False:
        return np.percentile(zz, 25, axis=1), np.percentile(zz, 75, axis=1)      # (two calls: a [25, 75] list interpolates in float64)
    p = np.partition(zz, [255, 256, 767, 768], axis=1)
    a, b = p[:, 255], p[:, 256]
    lo = b - (b - a) * np.float32(0.25)
    a, b = p[:, 767], p[:, 768]
    hi = a + (b - a) * np.float32(0.25)
    if _QUARTILE_CHECKED["ok"] is None:
        _QUARTILE_CHECKED["ok"] = bool(np.array_equal(lo, np.percentile(zz, 25, axis=1)) and np.array_equal(hi, np.percentile(zz, 75, axis=1)))
        if not _QUARTILE_CHECKED["ok"]:
            return np.percentile(zz, 25, axis=1), np.percentile(zz, 75, axis=1)
    return lo, hi


def layouts_from_signals_batch(y_bon, y_cor, mask_main, mask_zero=None, H=512, W=1024, force_cuboid=False, threads=1):
    """The host half of inference.py:89-141 for a BATCH: y_bon [B,2,W] float32 latitudes (radians), y_cor [B,W] float32 corner
    probability, mask_main / mask_zero [B,W] uint8 corner peaks at the caller's threshold / at 0 (the cuboid fallback's)
    -> list of (cor_id, z0, z1), bit-identical to B calls of ``layout_from_signals``.

    Every transcendental function is evaluated by numpy on whole [B, W] arrays (numpy's elementwise results do not depend on
    the array shape), the data-dependent part -- votes, the wall state machine, the validity test, the cuboid fallback --
    runs in the native library on `threads` threads (hn_layout_fit_batch), and the corner points go back through numpy."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    B = int(y_bon.shape[0])
    if B == 0:
        return []
    assert W == 1024 and H == 512, "the reference's floor-plan constants are those of a 1024 x 512 panorama"
    PI = postproc.PI
    yb = (np.asarray(y_bon, np.float32) / np.pi + 0.5) * H - 0.5
    yb[:, 0] = np.clip(yb[:, 0], 1, H / 2 - 1)
    yb[:, 1] = np.clip(yb[:, 1], H / 2 + 1, H - 2)
    z0 = 50
    # np_refine_by_fix_z (post_proc.py:109-123), float32 like the per-panorama path: only z1 is used downstream
    v0 = postproc.row_to_v(yb[:, 0], H)
    v1 = postproc.row_to_v(yb[:, 1], H)
    zz = (z0 / np.tan(v0)) * np.tan(v1)
    lo, hi = _quartiles_f32(zz)
    z1 = np.empty(B, np.float32)
    zz = np.ascontiguousarray(zz, np.float32)
    vpf = lambda a: ctypes.c_void_p(a.ctypes.data)                           # noqa: E731
    _lib.check(lib.hn_interquartile_mean_f32(vpf(zz), vpf(np.ascontiguousarray(lo)), vpf(np.ascontiguousarray(hi)), B, W, vpf(z1)),
               "hn_interquartile_mean_f32")
    tol = np.abs(0.16 * z1 / 1.6)                                           # inference.py:111, float32 arithmetic
    # np_coor2xy of (column, ceiling row) (post_proc.py:29-43) in float64, all panoramas at once
    sin_u, cos_u = _trig_tables(W)
    v = postproc.row_to_v(yb[:, 0].astype(np.float64), H)
    c = z0 / np.tan(v)
    xs = np.ascontiguousarray(c * sin_u[None, :W] + W / 2 - 0.5)
    ys = np.ascontiguousarray(-c * cos_u[None, :W] + H / 2 - 0.5)
    prob = np.asarray(y_cor, np.float32)
    tol64 = np.ascontiguousarray(tol, np.float64)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)                            # noqa: E731

    def top4_mask(mask_row, p_row):
        """inference.py:25-28 with N = 4: the four most probable peaks (numpy's argsort decides ties), as a mask."""
        pk = np.where(mask_row != 0)[0]
        pk = pk[np.argsort(-p_row[pk])[:4]]
        m = np.zeros(W, np.uint8)
        m[pk] = 1
        return m

    def run(idx, masks, cuboid):
        n = len(idx)
        pts_, npts_, flags_ = np.empty((n, 64, 2), np.float64), np.zeros(n, np.int32), np.zeros(n, np.int32)
        xs_, ys_, tol_ = (xs, ys, tol64) if n == B else (np.ascontiguousarray(xs[idx]), np.ascontiguousarray(ys[idx]), np.ascontiguousarray(tol64[idx]))
        _lib.check(lib.hn_layout_fit_batch(vp(xs_), vp(ys_), vp(masks), vp(sin_u), vp(cos_u), vp(tol_), n, W, int(cuboid),
                                           int(max(1, min(threads, n))), vp(pts_), vp(npts_), vp(flags_)), "hn_layout_fit_batch")
        return pts_, npts_, flags_

    everyone = np.arange(B)
    if force_cuboid:
        masks = np.ascontiguousarray(np.stack([top4_mask(mask_main[b], prob[b]) for b in range(B)]))
        pts, npts, flags = run(everyone, masks, True)
    else:
        pts, npts, flags = run(everyone, np.ascontiguousarray(mask_main, np.uint8), False)
        redo = np.nonzero(flags == 1)[0]
        if len(redo):                                                        # inference.py:121-126: invalid general layout -> cuboid
            mz = mask_main if mask_zero is None else mask_zero
            for _ in redo:
                print("Fail to generate valid general layout!! Generate cuboid as fallback.", file=sys.stderr)
            p2, n2, f2 = run(redo, np.ascontiguousarray(np.stack([top4_mask(mz[b], prob[b]) for b in redo])), True)
            pts[redo], npts[redo], flags[redo] = p2, n2, f2
    # flag 3: a layout with more walls than one call's output rows hold (HN_FIT_MAX_CORNERS; e.g. a saturated corner signal whose
    # plateau columns are all peaks).  The reference has no cap: those panoramas go through the per-panorama numpy path below, the
    # rest of the batch keeps its native results.  flag 2 (an assertion of the reference's gen_ww / vote) is reported PER panorama.
    over = np.nonzero(flags == 3)[0]
    slow = {}
    for b in over:
        peaks_of = _MaskPeaks({(0.0 if force_cuboid else 0.05): np.asarray(mask_main[b]),
                               0.0: np.asarray(mask_main[b] if (force_cuboid or mask_zero is None) else mask_zero[b])})
        try:
            slow[int(b)] = layout_from_signals(np.array(y_bon[b], np.float32), prob[b], H, W, force_cuboid=force_cuboid, peaks_fn=peaks_of)
        except AssertionError as exc:
            slow[int(b)] = exc
    bad = [int(b) for b in np.nonzero(flags == 2)[0]] + [b for b, v in slow.items() if isinstance(v, AssertionError)]
    if bad:
        err = AssertionError("layout fit: panorama(s) %s of the batch hit an assertion of the reference's gen_ww / vote "
                             "(too few corner peaks, degenerate vote); the other panoramas of the batch are unaffected" % sorted(bad))
        err.failed_panoramas = sorted(bad)
        raise err
    if len(over):
        keep = np.nonzero(flags != 3)[0]
        done = dict(slow)
        if len(keep):
            sub = _finish_layouts(pts[keep], npts[keep], z0, z1[keep], W, H)
            for i, b in enumerate(keep):
                done[int(b)] = sub[i]
        return [done[b] for b in range(B)]
    return _finish_layouts(pts, npts, z0, z1, W, H)


def _finish_layouts(pts, npts, z0, z1, W, H):
    """np_xy2coor + infer_coory + the roll to the left-most corner + cor_id for the corner points of a batch (second half of
    layouts_from_signals_batch)."""
    PI = postproc.PI
    B = len(npts)
    # np_xy2coor + infer_coory (post_proc.py:46-66,126-131) on all corner points at once
    off = np.concatenate([[0], np.cumsum(npts)])
    P = np.concatenate([pts[b, :npts[b]] for b in range(B)], 0)
    cor = postproc.plan_to_pano(P, z0, W, H, W, H)
    zh = np.repeat((z0 + (z1 - z0)).astype(np.float64), npts)              # z0 + h with h = z1 - z0 in float32 (inference.py:128)
    vv = postproc.row_to_v(cor[:, 1], H)
    floor_rows = (-np.arctan2(zh, z0 / np.tan(vv)) / PI + 0.5) * H - 0.5
    # gen_ww's np.roll(cor, -2 * argmin, axis=0) (post_proc.py:357) as one gather over all panoramas, then cor_id (inference.py:128-137)
    perm = np.empty(int(off[-1]), np.int64)
    for b in range(B):
        n, o = int(npts[b]), int(off[b])
        k = (2 * int(cor[o:o + n:2, 0].argmin())) % n
        perm[o:o + n] = o + (np.arange(n) + k) % n
    cor, floor_rows = cor[perm], floor_rows[perm]
    ids = np.zeros((2 * len(cor), 2), np.float32)
    ids[0::2, 0] = cor[:, 0]
    ids[0::2, 1] = cor[:, 1]
    ids[1::2, 0] = cor[:, 0]
    ids[1::2, 1] = floor_rows
    ids[:, 0] /= W
    ids[:, 1] /= H
    return [(ids[2 * off[b]:2 * off[b + 1]].copy(), z0, z1[b]) for b in range(B)]


def _forward_signals(net, x, device, flip, rotate):
    """-> (x_aug, y_bon [B,2,W] radians, y_cor [B,1,W] probability) numpy, augmentations undone and averaged."""
    B = x.shape[0]
    x_aug, tags = augment(x.to(device), flip, rotate)
    y_bon, y_cor = net(x_aug)
    y_bon = augment_undo(y_bon, tags).mean(0)
    y_cor = augment_undo(torch.sigmoid(y_cor), tags).mean(0)
    assert y_bon.shape[0] == B
    return x_aug, y_bon, y_cor


def inference(net, x, device, flip=False, rotate=[], visualize=False,
              force_cuboid=False, force_raw=False, min_v=None, r=0.05, peaks_fn=None):
    """Same arguments and return value as the reference: x [1,3,512,1024] -> (cor_id, z0, z1, vis_out)."""
    H, W = tuple(x.shape[2:])
    x_aug, y_bon, y_cor = _forward_signals(net, x, device, flip, rotate)
    vis_out = visualize_a_data(x_aug[0], y_bon[0], y_cor[0]) if visualize else None
    cor_id, z0, z1 = layout_from_signals(y_bon[0], y_cor[0, 0], H, W, force_cuboid, force_raw, min_v, r, peaks_fn)
    return cor_id, z0, z1, vis_out


class _MaskPeaks:
    """peaks_fn over peak masks computed on the device for the whole batch (picklable: goes to the fit workers)."""

    def __init__(self, masks):
        self.masks = masks                       # {min_v: uint8 [W]}

    def __call__(self, signal, r, min_v, N):
        pk_loc = np.where(self.masks[float(min_v)] != 0)[0]
        if N is not None:                        # inference.py:25-28: the N largest, in column order
            order = np.argsort(-signal[pk_loc])
            pk_loc = pk_loc[order[:N]]
            pk_loc = pk_loc[np.argsort(pk_loc)]
        return pk_loc, signal[pk_loc]


def _fit_one(job):
    y_bon, y_cor, H, W, force_cuboid, force_raw, min_v, r, masks = job
    return layout_from_signals(y_bon, y_cor, H, W, force_cuboid, force_raw, min_v, r, _MaskPeaks(masks))


_POOL = {"pool": None, "n": 0}


def _usable_cores():
    """Host cores THIS rank may use for the fit workers (hostcores.rank_cores: the cgroup grant divided by the ranks on the node)."""
    from .hostcores import rank_cores
    return rank_cores()


_SHM = {"bon": None, "cor": None, "mask": None}      # process-shared signal buffers (created before the pool forks)
_SHM_B, _SHM_W, _SHM_MASKS = 256, 1024, 2


def _shared_signals():
    """numpy views [B,2,W] f32, [B,W] f32, [masks,B,W] u8 over anonymous shared memory; the fit workers inherit them at fork."""
    if _SHM["bon"] is None:
        import ctypes
        import multiprocessing as mp
        _SHM["bon"] = np.frombuffer(mp.RawArray(ctypes.c_float, _SHM_B * 2 * _SHM_W), np.float32).reshape(_SHM_B, 2, _SHM_W)
        _SHM["cor"] = np.frombuffer(mp.RawArray(ctypes.c_float, _SHM_B * _SHM_W), np.float32).reshape(_SHM_B, _SHM_W)
        _SHM["mask"] = np.frombuffer(mp.RawArray(ctypes.c_uint8, _SHM_MASKS * _SHM_B * _SHM_W), np.uint8).reshape(_SHM_MASKS, _SHM_B, _SHM_W)
    return _SHM["bon"], _SHM["cor"], _SHM["mask"]


def _fit_shared(task):
    """Worker side of the shared-memory hand-over: panorama b of the batch lying in the shared buffers."""
    b, W, H, force_cuboid, force_raw, min_v, r, keys = task
    bon, cor, mask = _shared_signals()
    return layout_from_signals(bon[b, :, :W].copy(), cor[b, :W].copy(), H, W, force_cuboid, force_raw, min_v, r,
                               _MaskPeaks({k: mask[i, b, :W].copy() for i, k in enumerate(keys)}))


def _fit_pool(workers):
    """Process pool of the NON-default per-panorama fit (native=False; numpy + hn_vote_scan, the workers never touch the GPU).
    It forks lazily, i.e. possibly after the HIP runtime has been initialised in this process: call it once before the first
    GPU use if you rely on this path.  The default path (hn_layout_fit_batch) needs no processes at all."""
    import multiprocessing as mp
    if _POOL["pool"] is None or _POOL["n"] != workers:
        if _POOL["pool"] is not None:
            _POOL["pool"].terminate()
        _shared_signals()                    # must exist BEFORE the fork so that parent and workers map the same pages
        _POOL["pool"] = mp.get_context("fork").Pool(workers)
        _POOL["n"] = workers
    return _POOL["pool"]


def _undo_mean_device(y, tags):
    """`augment_undo(y, tags).mean(0)` without leaving the device: the flips / rolls are exact, the mean is numpy's own
    arithmetic (slices summed in order, then ONE true division), so the result is bit-identical to the host path."""
    sz = y.shape[0] // len(tags)
    acc = None
    for i, tag in enumerate(tags):
        part = y[i * sz:(i + 1) * sz]
        if tag == "flip":
            part = torch.flip(part, dims=[-1])
        elif tag.startswith("rotate"):
            part = torch.roll(part, -int(tag.split()[-1]), dims=-1)
        elif tag != "":
            raise NotImplementedError()
        acc = part if acc is None else acc + part
    if len(tags) > 1:       # a 0-dim DEVICE divisor: a python scalar would be turned into a multiplication by 1/A
        acc = acc / torch.tensor(float(len(tags)), dtype=acc.dtype, device=acc.device)
    return acc


class _HostRing:
    """Pinned host buffers for the signals / peak masks of the batches in flight (page-locked: the copies are asynchronous)."""

    def __init__(self):
        self.slots = {}
        self.turn = 0

    def take(self, B, W, nmask, depth):
        self.turn += 1
        key = (B, W, nmask, self.turn % (depth + 1))
        if key not in self.slots:
            self.slots[key] = (torch.empty((B, 2, W), dtype=torch.float32).pin_memory(),
                               torch.empty((B, W), dtype=torch.float32).pin_memory(),
                               torch.empty((nmask, B, W), dtype=torch.uint8).pin_memory())
        return self.slots[key]


_RING = _HostRing()


def _stage_submit(net, x, device, flip, rotate, pipelined=True):
    """First device stage of one batch: test-time augmentation + the engine forward, submitted asynchronously
    (``forward_async``: in bf16 mode the recurrent head runs on the engine's own stream beside the NEXT batch's trunk).
    pipelined=False (``inference_batch``): the plain ``net(x)`` of ``inference()`` -- in float32 the pipelined entry's recurrence kernel
    sums in another order (~1e-6 on the logits, enough to flip a peak at the min_v threshold), and ``inference_batch`` promises the
    results of B calls of ``inference()``; in bf16 the two entries are bit-identical, so the stream form keeps the faster one."""
    x_aug, tags = augment(x.to(device), flip, rotate)
    use_async = hasattr(net, "forward_async") and (pipelined or getattr(net, "precision", "f32") == "bf16")
    fwd = net.forward_async(x_aug) if use_async else None
    return {"pending": fwd, "out": None if fwd is not None else net(x_aug), "tags": tags, "B": int(x.shape[0]), "W": int(x.shape[-1]),
            "net": net, "device": device}


def _stage_post(sub, force_cuboid, min_v, r, depth):
    """Second device stage, enqueued without any host synchronisation: wait (stream-side) for the forward's head, augmentation
    undo + mean, sigmoid, one hn_find_peaks launch per threshold, asynchronous copies into pinned host buffers, one event."""
    from .peaks import find_peaks_batch
    W, B, tags, device = sub["W"], sub["B"], sub["tags"], sub["device"]
    y_bon, y_cor = sub["pending"].result() if sub["pending"] is not None else sub["out"]
    bon = _undo_mean_device(y_bon, tags)
    prob = _undo_mean_device(torch.sigmoid(y_cor), tags)[:, 0].contiguous()
    mv = (0.0 if force_cuboid else 0.05) if min_v is None else float(min_v)
    rr = int(round(W * r / 2))
    keys = [mv] + ([0.0] if mv != 0.0 else [])      # the cuboid fallback of inference.py:121-126 re-detects with min_v = 0
    h_bon, h_cor, h_mask = _RING.take(B, W, len(keys), depth)
    h_bon.copy_(bon, non_blocking=True)
    h_cor.copy_(prob, non_blocking=True)
    for i, k in enumerate(keys):
        h_mask[i].copy_(find_peaks_batch(prob, rr, k)[0], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(torch.device(device)))
    return {"event": ev, "bon": h_bon, "cor": h_cor, "mask": h_mask, "keys": keys, "B": B, "keep": (bon, prob), "net": sub["net"],
            "device": device}


def _stage_device(net, x, device, flip, rotate, force_cuboid, min_v, r, depth):
    """Both device stages of one batch back to back (the unpipelined form)."""
    return _stage_post(_stage_submit(net, x, device, flip, rotate, pipelined=False), force_cuboid, min_v, r, depth)


def _raise_if_engine_failed(st):
    """The forward posted an asynchronous read of the persistent LSTM's status word BEFORE the event this batch was just
    synchronised on, so the word is on the host now: a timed-out recurrence means THIS batch's signals are garbage -- raise
    here instead of on the next engine call (which, for the last batch of a stream, never comes)."""
    net, device = st.get("net"), st.get("device")
    states = getattr(net, "_hip_states", None)
    if states:
        ds = states.get(torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
        if ds is not None:
            ds.raise_if_failed()


def _finish_host(st, H, W, force_cuboid, force_raw, min_v, r, workers, shared_signals=False, native=True):
    """Host half: wait for THIS batch's copies, then the Manhattan fits.  native (default): the batched fit -- numpy for the
    transcendental arrays, hn_layout_fit_batch on `workers` THREADS inside the library for the decisions (no worker
    processes, nothing pickled, no fork).  native=False: the per-panorama numpy restatement, inline or on a process pool
    (shared_signals: shared-memory hand-over instead of pickles); kept as the cross-check of the native path."""
    st["event"].synchronize()
    _raise_if_engine_failed(st)
    B = st["B"]
    y_bon, y_cor, masks = st["bon"].numpy(), st["cor"].numpy(), st["mask"].numpy()
    if native and not force_raw and W == 1024 and H == 512:
        keys = list(st["keys"])
        return layouts_from_signals_batch(y_bon[:B], y_cor[:B], masks[0][:B], masks[keys.index(0.0)][:B] if 0.0 in keys else None,
                                          H, W, force_cuboid, threads=_usable_cores() if not workers else workers)
    if workers is None:
        workers = min(_usable_cores(), 32) if B >= 8 else 0
    if shared_signals and workers > 1 and B <= _SHM_B and W <= _SHM_W and len(st["keys"]) <= _SHM_MASKS:
        pool = _fit_pool(workers)            # (creates the shared buffers before forking)
        sb, sc, sm = _shared_signals()
        sb[:B, :, :W] = y_bon
        sc[:B, :W] = y_cor
        sm[:len(st["keys"]), :B, :W] = masks
        tasks = [(b, W, H, force_cuboid, force_raw, min_v, r, tuple(st["keys"])) for b in range(B)]
        return pool.map(_fit_shared, tasks, chunksize=max(1, -(-B // workers)))     # blocking: the buffers are free again on return
    jobs = [(y_bon[b].copy(), y_cor[b].copy(), H, W, force_cuboid, force_raw, min_v, r,
             {k: masks[i, b].copy() for i, k in enumerate(st["keys"])}) for b in range(B)]
    if workers <= 1:
        return [_fit_one(j) for j in jobs]
    return _fit_pool(workers).map(_fit_one, jobs, chunksize=max(1, B // (4 * workers)))


def inference_batch(net, x, device, flip=False, rotate=[], force_cuboid=False, force_raw=False, min_v=None, r=0.05,
                    peaks_fn=None, workers=None, shared_signals=False, native=True):
    """x [B,3,512,1024]: one engine forward for all B x augmentations, ONE peak-detection launch per threshold for the
    whole batch (hn_find_peaks), then the host Manhattan fit of the B panoramas as ONE batched call (numpy for the
    transcendental arrays, hn_layout_fit_batch on `workers` threads inside the library; default: this rank's share of the
    host cores).  native=False: the per-panorama numpy restatement on `workers` forked processes (0 = inline), kept as a
    cross-check.  -> list of (cor_id, z0, z1), identical to B calls of ``inference``."""
    H, W = tuple(x.shape[2:])
    B = int(x.shape[0])
    if peaks_fn is not None:                     # caller-supplied peak finder: the per-panorama path
        _, y_bon, y_cor = _forward_signals(net, x, device, flip, rotate)
        return [layout_from_signals(y_bon[b], y_cor[b, 0], H, W, force_cuboid, force_raw, min_v, r, peaks_fn) for b in range(B)]
    st = _stage_device(net, x, device, flip, rotate, force_cuboid, min_v, r, depth=1)
    return _finish_host(st, H, W, force_cuboid, force_raw, min_v, r, workers, shared_signals, native)


def inference_stream(net, batches, device, flip=False, rotate=[], force_cuboid=False, force_raw=False, min_v=None, r=0.05,
                     workers=None, depth=2, shared_signals=False, native=True):
    """Pipelined ``inference_batch`` over an iterable of batches x [B,3,512,1024]: yields one list of (cor_id, z0, z1) per
    batch, in order, identical to ``inference_batch`` batch by batch.  The device half of the next `depth - 1` batches
    (forward, peaks, copies to pinned memory: all asynchronous) is enqueued BEFORE the host half of the current one (the
    Manhattan fits) runs, so the GPU works on batch i + 1 while the host cores fit batch i -- the per-batch loop of
    inference.py:187-209 leaves each side idle while the other works.  shared_signals=True hands the batch to the fit workers
    through shared memory instead of pickles (same results, tests/test_postproc_cpu.py; off by default until it has been timed
    on the target box)."""
    from collections import deque
    pending = deque()
    sub = None                         # the batch whose forward is submitted but whose second device stage is not yet enqueued:
    try:
        for x in batches:              # batch i+1's trunk goes onto the stream BEFORE batch i's wait for its recurrent head
            H, W = tuple(x.shape[2:])
            nxt = (_stage_submit(net, x, device, flip, rotate), H, W)
            if sub is not None:
                pending.append((_stage_post(sub[0], force_cuboid, min_v, r, depth + 1), sub[1], sub[2]))
            sub = nxt
            if len(pending) >= depth:
                st, h, w = pending.popleft()
                yield _finish_host(st, h, w, force_cuboid, force_raw, min_v, r, workers, shared_signals, native)
        if sub is not None:
            pending.append((_stage_post(sub[0], force_cuboid, min_v, r, depth + 1), sub[1], sub[2]))
            sub = None
        while pending:
            st, h, w = pending.popleft()
            yield _finish_host(st, h, w, force_cuboid, force_raw, min_v, r, workers, shared_signals, native)
    finally:
        # the generator was closed early (or a stage raised): the forward of `sub` is still writing its outputs from the engine's own
        # stream -- collect it before its tensors go back to the allocator (PendingForward.__del__ does the same as a last resort)
        if sub is not None and sub[0].get("pending") is not None:
            try:
                sub[0]["pending"].result()
            except Exception as exc:        # noqa: BLE001 -- never replace the exception that is already propagating (or surface one from close())
                import warnings
                warnings.warn("inference_stream: collecting the in-flight forward during clean-up failed: %s: %s" % (type(exc).__name__, exc),
                              RuntimeWarning)
