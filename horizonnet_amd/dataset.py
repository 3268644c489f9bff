"""Training data pipeline -- drop-in for reference ``dataset.PanoCorBonDataset`` (``dataset.py:12-134``) with the
image half on the MI355X.

* ``PanoCorBonDataset``: same constructor, same on-disk layout (``root/img/*.png|jpg``, ``root/label_cor/*.txt``), same
  per-sample return value, same ``np.random`` draw order -- for code that indexes samples one by one.
* ``DeviceBatcher``: the MI355X-shaped path.  The decoded dataset sits in HBM as uint8 (1.5 MB per panorama; a
  20k-panorama training set is 31 GB of the 288 GB); a training batch is ONE fused launch (``hn_augment_batch``:
  /255, Pano-Stretch, flip, roll, gamma, HWC->CHW) straight into the float32 NCHW tensor the engine consumes.  The
  labels are rasterised on the device too (``hn_labels_rasterise``, csrc/labels.hip): the host keeps the per-corner
  scalars (<= ~30 corners, the reference's float32 flow) and every per-column operation runs in one more launch;
  ``device_labels=False`` keeps the host rasterisation (``labels.py``, ~1.5 ms per panorama).

The augmentation parameters are drawn by ``draw_augmentation`` in the reference's order (stretch: uniform, uniform,
randint, randint; flip: randint; rotate: randint; gamma: uniform, randint), so a seeded run reproduces the reference's
samples.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .augment import sample_stretch
from .labels import (LABEL_MAX_CORNERS, LABEL_MAX_EDGES, LABEL_REC_FLOATS, cor_2_1d, corner_probability, device_label_record, find_occlusion,
                     flip_labels, roll_labels)
from .panostretch import _stretch_corners


def read_label(path, W=1024):
    """label_cor text -> (cor [2N,2] float32 starting at the smallest column, occlusion flags [2N])
    (dataset.py:56-67)."""
    with open(path) as f:
        cor = np.array([line.strip().split() for line in f if line.strip()], np.float32)
    cor = np.roll(cor[:, :2], -2 * np.argmin(cor[::2, 0]), 0)
    occlusion = find_occlusion(cor[::2].copy()).repeat(2)
    assert (np.abs(cor[0::2, 0] - cor[1::2, 0]) > W / 100).sum() == 0, path
    assert (cor[0::2, 1] > cor[1::2, 1]).sum() == 0, path
    return cor, occlusion


def draw_augmentation(cor, W, flip, rotate, gamma, stretch, max_stretch=2.0, rng=np.random):
    """One sample's augmentation parameters, consuming random numbers exactly like dataset.py:70-104."""
    a = {"kx": 1.0, "ky": 1.0, "flip": 0, "rotate": False, "roll": 0, "gamma": 1.0}
    if stretch:
        a["kx"], a["ky"] = sample_stretch(cor, max_stretch, rng)
    if flip and rng.randint(2) == 0:
        a["flip"] = 1
    if rotate:
        a["rotate"], a["roll"] = True, int(rng.randint(W))
    if gamma:
        p = rng.uniform(1, 2)
        if rng.randint(2) == 0:
            p = 1 / p
        a["gamma"] = p
    return a


def make_labels(cor, occlusion, aug, H, W, p_base=0.96):
    """Corner list + augmentation -> (bon [2,W] float64, y_cor [1,W] float64, augmented cor) (dataset.py:82-120)."""
    if not (aug["kx"] == 1.0 and aug["ky"] == 1.0):
        cor = _stretch_corners(cor, aug["kx"], aug["ky"], W, H)
    bon = cor_2_1d(cor, H, W)
    if aug["flip"]:
        bon, cor = flip_labels(bon, cor, W)
    if aug["rotate"]:                      # also for dx = 0: the modulo wraps columns a stretch pushed below 0
        bon, cor = roll_labels(bon, cor, aug["roll"], W)
    return bon, corner_probability(cor[~occlusion, 0], W, p_base), cor


def augment_images(data, index, augs, out=None):
    """data: cuda uint8 [N,H,W,3]; index: B ints; augs: B dicts from `draw_augmentation` -> cuda float32 [B,3,H,W]."""
    if not (isinstance(data, torch.Tensor) and data.is_cuda and data.dtype == torch.uint8 and data.dim() == 4
            and data.shape[3] == 3):
        raise RuntimeError("augment_images needs a cuda/ROCm uint8 [N,H,W,3] tensor (no CPU fallback)")
    data = data.contiguous()
    N, H, W, _ = (int(v) for v in data.shape)
    B = len(index)
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=data.device)
    idx = np.ascontiguousarray(index, dtype=np.int32)
    kx = np.array([a["kx"] for a in augs], np.float64)
    ky = np.array([a["ky"] for a in augs], np.float64)
    flip = np.array([a["flip"] for a in augs], np.int32)
    roll = np.array([a["roll"] for a in augs], np.int32)
    gam = np.array([a["gamma"] for a in augs], np.float64)
    ip, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    lib = _lib.load()
    with torch.cuda.device(data.device):
        _lib.check(lib.hn_augment_batch(_lib.ptr(data), N, idx.ctypes.data_as(ip), _lib.ptr(out),
                                        kx.ctypes.data_as(dp), ky.ctypes.data_as(dp), flip.ctypes.data_as(ip),
                                        roll.ctypes.data_as(ip), gam.ctypes.data_as(dp), B, H, W,
                                        _lib.stream_ptr(data.device)), "hn_augment_batch")
    return out


_IDENTITY_AUG = {"kx": 1.0, "ky": 1.0, "flip": 0, "roll": 0, "gamma": 1.0}


def images_to_input(data, index, out=None):
    """uint8 HWC panoramas resident in HBM -> the float32 NCHW batch ``net(x)`` consumes, in ONE pass of the augmentation kernel
    with every augmentation off: ``np.array(img)[..., :3].transpose(2, 0, 1) / 255`` -> FloatTensor of reference
    inference.py:196-200 (the same 256 float32 values i / 255) for a whole batch, 1.5 MB read + 6.3 MB written per panorama."""
    return augment_images(data, index, [_IDENTITY_AUG] * len(index), out=out)


def _list_samples(root_dir):
    img_dir, cor_dir = os.path.join(root_dir, "img"), os.path.join(root_dir, "label_cor")
    names = sorted(f for f in os.listdir(img_dir) if f.endswith(".jpg") or f.endswith(".png"))
    labels = ["%s.txt" % f[:-4] for f in names]
    for f in labels:
        assert os.path.isfile(os.path.join(cor_dir, f)), "%s not found" % os.path.join(cor_dir, f)
    return img_dir, cor_dir, names, labels


def _decode(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path))[..., :3])


class PanoCorBonDataset(torch.utils.data.Dataset):
    """Per-sample interface of the reference (dataset.py:12-134): returns [x (3,H,W), bon (2,W), y_cor (1,W)] float
    CPU tensors (+ cor, + path on request).  The image goes through the same fused HIP kernel as the batched path."""

    def __init__(self, root_dir, flip=False, rotate=False, gamma=False, stretch=False, p_base=0.96, max_stretch=2.0,
                 normcor=False, return_cor=False, return_path=False, device="cuda"):
        self.img_dir, self.cor_dir, self.img_fnames, self.txt_fnames = _list_samples(root_dir)
        self.flip, self.rotate, self.gamma, self.stretch = flip, rotate, gamma, stretch
        self.p_base, self.max_stretch = p_base, max_stretch
        self.normcor, self.return_cor, self.return_path = normcor, return_cor, return_path
        self.device = device

    def __len__(self):
        return len(self.img_fnames)

    def __getitem__(self, idx):
        img_path = os.path.join(self.img_dir, self.img_fnames[idx])
        img = _decode(img_path)
        H, W = img.shape[:2]
        cor, occlusion = read_label(os.path.join(self.cor_dir, self.txt_fnames[idx]), W)
        aug = draw_augmentation(cor, W, self.flip, self.rotate, self.gamma, self.stretch, self.max_stretch)
        bon, y_cor, cor = make_labels(cor, occlusion, aug, H, W, self.p_base)
        x = augment_images(torch.from_numpy(img)[None].to(self.device), [0], [aug])[0].cpu()
        out = [x, torch.FloatTensor(bon.copy()), torch.FloatTensor(y_cor.copy())]
        if self.return_cor:
            out.append(cor)
        if self.return_path:
            out.append(img_path)
        return out


class DeviceBatcher:
    """The whole dataset decoded once into HBM (uint8), labels parsed once on the host; `batch(indices)` returns the
    augmented training batch on the device: (x [B,3,H,W], bon [B,2,W], y_cor [B,1,W]) float32."""

    def __init__(self, root_dir=None, images=None, corners=None, device="cuda", flip=False, rotate=False, gamma=False,
                 stretch=False, p_base=0.96, max_stretch=2.0, device_labels=True):
        self.device_labels = bool(device_labels)
        self._unexpected_flags = None                      # device scalar: kernel-flagged rows the host half had not predicted (unexpected_label_flags)
        if root_dir is not None:
            img_dir, cor_dir, names, labels = _list_samples(root_dir)
            images = np.stack([_decode(os.path.join(img_dir, n)) for n in names])
            parsed = [read_label(os.path.join(cor_dir, t), images.shape[2]) for t in labels]
        else:
            images = np.ascontiguousarray(images, dtype=np.uint8)
            parsed = [(np.asarray(c, np.float32), find_occlusion(np.asarray(c, np.float32)[::2].copy()).repeat(2))
                      for c in corners]
        self.data = torch.from_numpy(images).to(device)
        self.labels = parsed
        self.flip, self.rotate, self.gamma, self.stretch = flip, rotate, gamma, stretch
        self.p_base, self.max_stretch = p_base, max_stretch
        self.H, self.W = int(images.shape[1]), int(images.shape[2])
        self._staging, self._next_slot, self._ring = [], 0, 4     # pinned label staging ring (device_half)

    def __len__(self):
        return len(self.labels)

    def host_half(self, indices, rng=np.random):
        """The host part of one batch: augmentation draws (reference order) + the labels' host half ->
        (indices, augs, bon, y_cor).  device_labels: bon = the per-panorama records [B, LABEL_REC_FLOATS] float32 for
        hn_labels_rasterise and y_cor = {row: (bon [2,W], y_cor [1,W])} for the panoramas that must be rasterised on the host
        (an outline that does not close; normally empty).  Otherwise bon [B,2,W] / y_cor [B,1,W] float32 from labels.py."""
        augs, bons, cors = [], [], []
        recs = np.empty((len(indices), LABEL_REC_FLOATS), np.float32) if self.device_labels else None
        on_host = {}
        for row, i in enumerate(indices):
            cor, occ = self.labels[i]
            a = draw_augmentation(cor, self.W, self.flip, self.rotate, self.gamma, self.stretch, self.max_stretch, rng)
            augs.append(a)
            if self.device_labels:
                c = cor if (a["kx"] == 1.0 and a["ky"] == 1.0) else _stretch_corners(cor, a["kx"], a["ky"], self.W, self.H)
                _, closed = device_label_record(c, occ, a["flip"], a["roll"] if a["rotate"] else None, self.H, self.W, out=recs[row])
                if not closed:
                    bon, y_cor, _ = make_labels(cor, occ, a, self.H, self.W, self.p_base)
                    on_host[row] = (bon.astype(np.float32), y_cor.astype(np.float32))
                continue
            bon, y_cor, _ = make_labels(cor, occ, a, self.H, self.W, self.p_base)
            bons.append(bon)
            cors.append(y_cor)
        if self.device_labels:
            return list(indices), augs, recs, on_host
        return list(indices), augs, np.stack(bons).astype(np.float32), np.stack(cors).astype(np.float32)

    def device_half(self, indices, augs, bon, y_cor):
        """The device part: one fused augmentation launch + the labels (device_labels: the record upload and ONE rasterisation
        launch; otherwise the upload of the host-rasterised vectors).  Uploads go through a ring of PINNED staging tensors: a copy
        from pageable memory is a host-side wait for the stream on ROCm, i.e. one full host <-> GPU synchronisation per step."""
        x = augment_images(self.data, indices, augs)
        dev = self.data.device
        B = len(indices)
        k = self._next_slot % self._ring
        self._next_slot += 1
        if k == len(self._staging):
            self._staging.append(None)
        slot = self._staging[k]
        if slot is None or slot["bon"].shape[0] < B:
            slot = {"bon": torch.empty((B, 2, self.W), dtype=torch.float32).pin_memory(),
                    "cor": torch.empty((B, 1, self.W), dtype=torch.float32).pin_memory(),
                    "rec": torch.empty((B, LABEL_REC_FLOATS), dtype=torch.float32).pin_memory(), "done": torch.cuda.Event()}
            self._staging[k] = slot
        else:
            slot["done"].synchronize()                  # the upload that last used this slot (ring-size batches ago)
        if self.device_labels:
            slot["rec"][:B].copy_(torch.from_numpy(bon))
            with torch.cuda.device(dev):
                d_rec = slot["rec"][:B].to(dev, non_blocking=True)
                slot["done"].record()
                d_bon = torch.empty((B, 2, self.W), dtype=torch.float32, device=dev)
                d_cor = torch.empty((B, 1, self.W), dtype=torch.float32, device=dev)
                status = torch.empty((B,), dtype=torch.int32, device=dev)
                _lib.check(_lib.load().hn_labels_rasterise(_lib.ptr(d_rec), LABEL_REC_FLOATS, LABEL_MAX_EDGES, LABEL_MAX_CORNERS, B, self.H, self.W,
                                                          float(self.p_base), _lib.ptr(d_bon), _lib.ptr(d_cor), _lib.ptr(status),
                                                          _lib.stream_ptr(dev)), "hn_labels_rasterise")
                for row, (hb, hc) in y_cor.items():     # outlines the device kernel does not take (host_half)
                    d_bon[row].copy_(torch.from_numpy(hb))
                    d_cor[row].copy_(torch.from_numpy(hc))
            self.last_label_status = status             # device int32 [B]: 1 = a column without a trace point (tests read it)
            # The host half predicts which outlines the kernel cannot close (`on_host` rows, rasterised there and copied in above); a row the
            # KERNEL flags but the host did not predict would carry a wrong boundary label silently.  Counted on the device (no sync here),
            # read once per epoch by the training loop (unexpected_label_flags), which refuses to go on.
            unexpected = status
            if y_cor:                                   # (rare: a clone and a masked fill only for batches that have host-rasterised rows)
                keep = torch.ones(B, dtype=torch.int32)
                keep[sorted(y_cor.keys())] = 0
                unexpected = status * keep.pin_memory().to(dev, non_blocking=True)
            self._unexpected_flags = unexpected.sum() if self._unexpected_flags is None else self._unexpected_flags + unexpected.sum()
            return x, d_bon, d_cor
        slot["bon"][:B].copy_(torch.from_numpy(bon))
        slot["cor"][:B].copy_(torch.from_numpy(y_cor))
        with torch.cuda.device(dev):
            d_bon = slot["bon"][:B].to(dev, non_blocking=True)
            d_cor = slot["cor"][:B].to(dev, non_blocking=True)
            slot["done"].record()
        return x, d_bon, d_cor

    def unexpected_label_flags(self):
        """Panoramas (since the last call) whose device-rasterised labels had a column without a trace point although the host half had
        predicted a closed outline -- must be 0; synchronises with the device (call it once per epoch, not per batch)."""
        n = 0 if self._unexpected_flags is None else int(self._unexpected_flags.item())
        self._unexpected_flags = None
        return n

    def batch(self, indices, rng=np.random):
        return self.device_half(*self.host_half(indices, rng))

    def stream(self, index_batches, rng=np.random, depth=2, side_stream=False):
        """Iterate over batches with the host half running `depth` batches AHEAD in a background thread -- the role of
        the reference's DataLoader workers (train.py:153-160).  `index_batches`: an iterable of index arrays; the thread
        consumes it and `rng` in order, so a seeded run draws exactly what the unthreaded loop draws.  The engine's
        forward / backward are single foreign calls (GIL released), so the thread genuinely overlaps with them.
        side_stream: the device half (augmentation gather, label rasterisation, uploads: ~1.5 ms of device time at B = 64) is enqueued on a
        stream of its own and the caller's stream only waits for its event -- the host runs about a step ahead of the GPU, so the launches of
        batch i + 1 execute beside the tail of step i (the recurrence's adjoint leaves most compute units idle) instead of behind it: +0.5 % on the
        step.  OFF by default: the first torch side stream of a process creates torch's pool of 32 HIP streams, which re-maps every stream of the
        process onto the (4) hardware queues -- the engine's trunk / branch / head streams of a LATER pipelined inference in the same process then
        share queues and lose their overlap (bench.py: the layout leg behind the training leg 4900 -> 3900 panoramas/s, also after the stream was
        released)."""
        return _BatchStream(self, index_batches, rng, depth, side_stream)


class _BatchStream:
    def __init__(self, owner, index_batches, rng, depth, side_stream=False):
        import queue
        import threading
        self.owner = owner
        self.side = None
        self.use_side = ((bool(side_stream) or os.environ.get("HN_BATCH_SIDE_STREAM", "0") == "1")          # (1: A/B runs)
                         and isinstance(getattr(owner, "data", None), torch.Tensor) and owner.data.is_cuda)
        self.q = queue.Queue(maxsize=max(1, int(depth)))
        self.stop = threading.Event()

        def work():
            try:
                for idx in index_batches:
                    if self.stop.is_set():
                        return
                    item = owner.host_half(idx, rng)
                    while not self.stop.is_set():
                        try:
                            self.q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                self.q.put(None)
            except BaseException as e:                   # surfaces in the consumer, not in a dead thread
                self.q.put(e)

        self.thread = threading.Thread(target=work, name="horizonnet-batch-host-half", daemon=True)
        self.thread.start()

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        if not self.use_side:
            return self.owner.device_half(*item)
        dev = self.owner.data.device
        if self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        with torch.cuda.stream(self.side):
            out = self.owner.device_half(*item)
            done = torch.cuda.Event()
            done.record(self.side)
        main.wait_event(done)
        for t in out:                      # allocated on the side stream, consumed on the caller's: the allocator must not recycle them early
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)
        return out

    def close(self):
        if self.side is not None:
            # blocks the side stream allocated stay in ITS pool of torch's caching allocator: whatever runs next on the caller's stream (a validation
            # pass, inference) would have to hipMalloc afresh (bench.py: the layout leg behind the training leg lost 20 %).  Hand them back.
            self.side.synchronize()
            self.side = None
            torch.cuda.empty_cache()
        self.stop.set()
        while self.thread.is_alive():
            try:
                self.q.get_nowait()
            except Exception:
                pass
            self.thread.join(timeout=0.05)
