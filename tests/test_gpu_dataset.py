"""hn_augment_batch / PanoCorBonDataset / DeviceBatcher on the MI355X against the reference's own samples
(tests/golden/dataset.npz, generated from the unmodified dataset.py) and the oracle image pipeline."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from horizonnet_amd import dataset as ds  # noqa: E402
from oracle import dataset_ref  # noqa: E402
from hiputil import DEV  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.join(GOLD, "synth_ds")


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_dataset_samples_match_reference():
    """Per-sample drop-in: same seeds as the reference run -> the image is bit-identical where no gamma is applied and
    within 1 ulp (float32 pow) where it is; labels identical."""
    g = np.load(os.path.join(GOLD, "dataset.npz"))
    meta = json.load(open(os.path.join(GOLD, "dataset.json")))
    worst_ulp, n_exact = 0, 0
    for m in meta:
        d = ds.PanoCorBonDataset(ROOT, return_cor=True, device=DEV, **m["cfg"])
        np.random.seed(m["seed"])
        x, bon, y_cor, cor = d[m["index"]]
        k = m["key"]
        assert x.shape == (3, 512, 1024) and x.dtype == torch.float32
        got = x.numpy()[:, 3::16, 5::16]
        if m["cfg"].get("gamma"):
            u = ulp_diff(got, g["x_" + k])
            worst_ulp = max(worst_ulp, int(u.max()))
            assert u.max() <= 1, (k, int(u.max()))
            assert abs(float(x.numpy().astype(np.float64).sum()) - float(g["xsum_" + k])) < 1e-6 * float(g["xsum_" + k])
        else:
            assert np.array_equal(got, g["x_" + k]), k
            assert float(x.numpy().astype(np.float64).sum()) == float(g["xsum_" + k]), k
            n_exact += 1
        assert np.array_equal(bon.numpy(), g["bon_" + k]) and np.array_equal(y_cor.numpy(), g["ycor_" + k])
        assert np.array_equal(cor, g["cor_" + k])
    print("[parity] dataset samples: %d bit-identical images, gamma images within %d ulp" % (n_exact, worst_ulp))


def test_augment_batch_vs_oracle_full_images():
    """One fused launch for a mixed batch (every augmentation on/off combination, repeated source images) against
    the oracle pipeline on every pixel."""
    batcher = ds.DeviceBatcher(ROOT, device=DEV)
    rng = np.random.RandomState(3)
    index, augs = [], []
    for b in range(12):
        i = int(rng.randint(len(batcher)))
        a = {"kx": 1.0, "ky": 1.0, "flip": 0, "rotate": False, "roll": 0, "gamma": 1.0}
        if b & 1:
            a["kx"], a["ky"] = float(rng.uniform(0.5, 2)), float(rng.uniform(0.5, 2))
        if b & 2:
            a["flip"] = 1
        if b & 4:
            a["rotate"], a["roll"] = True, int(rng.randint(1024))
        if b & 8:
            a["gamma"] = float(rng.uniform(0.5, 2))
        index.append(i)
        augs.append(a)
    out = ds.augment_images(batcher.data, index, augs).cpu().numpy()
    src = batcher.data.cpu().numpy()
    for b, (i, a) in enumerate(zip(index, augs)):
        want = dataset_ref.augment_image(src[i], a["kx"], a["ky"], a["flip"], a["roll"] if a["rotate"] else None, a["gamma"])
        if a["gamma"] == 1.0:
            assert np.array_equal(out[b], want), (b, a)
        else:
            # numpy's float32 power is a SIMD approximation (~1 ulp) that varies with the host CPU; the kernel rounds a
            # float64 pow once, so the two agree to 1 ulp, not bit for bit
            u = ulp_diff(out[b], want)
            assert u.max() <= 1, (b, a, int(u.max()))


def test_augment_symmetric_kernel_equals_per_pixel_kernel():
    """augment_sym_kernel (one arctangent per four mirror-image pixels; flip / roll applied on the way out) against the
    per-pixel kernel: identical bits for every augmentation combination."""
    batcher = ds.DeviceBatcher(ROOT, device=DEV)
    rng = np.random.RandomState(13)
    index, augs = [], []
    for b in range(16):
        a = {"kx": 1.0, "ky": 1.0, "flip": 0, "rotate": False, "roll": 0, "gamma": 1.0}
        if b & 1:
            a["kx"], a["ky"] = float(rng.uniform(0.5, 2)), float(rng.uniform(0.5, 2))
        if b & 2:
            a["flip"] = 1
        if b & 4:
            a["rotate"], a["roll"] = True, int(rng.randint(1024))
        if b & 8:
            a["gamma"] = float(rng.uniform(0.5, 2))
        index.append(int(rng.randint(len(batcher))))
        augs.append(a)
    os.environ["HN_STRETCH_SYM"] = "0"
    try:
        want = ds.augment_images(batcher.data, index, augs).clone()
    finally:
        os.environ.pop("HN_STRETCH_SYM", None)
    got = ds.augment_images(batcher.data, index, augs)
    torch.cuda.synchronize()
    assert torch.equal(got, want), "symmetric augment kernel differs in %d values" % int((got != want).sum())


def test_device_batcher_equals_per_sample_path():
    kw = dict(flip=True, rotate=True, gamma=True, stretch=True)
    batcher = ds.DeviceBatcher(ROOT, device=DEV, **kw)
    per = ds.PanoCorBonDataset(ROOT, device=DEV, **kw)
    idx = [2, 0, 3, 3, 1]
    np.random.seed(77)
    x, bon, y_cor = batcher.batch(idx)
    assert x.shape == (5, 3, 512, 1024) and x.is_cuda and bon.shape == (5, 2, 1024) and y_cor.shape == (5, 1, 1024)
    np.random.seed(77)
    for b, i in enumerate(idx):
        xs, bs, cs = per[i]
        assert torch.equal(x[b].cpu(), xs)
        # labels: rasterised on the device (hn_labels_rasterise) against the per-sample host path -- <= 1 float32 ulp (device tan / atan2 / pow)
        assert ulp_diff(bon[b].cpu().numpy(), bs.numpy()).max() <= 1 and ulp_diff(y_cor[b].cpu().numpy(), cs.numpy()).max() <= 1
    assert int(batcher.last_label_status.sum()) == 0
    host = ds.DeviceBatcher(ROOT, device=DEV, device_labels=False, **kw)
    np.random.seed(77)
    xh, bh, ch = host.batch(idx)
    np.random.seed(77)
    for b, i in enumerate(idx):
        xs, bs, cs = per[i]
        assert torch.equal(xh[b].cpu(), xs) and torch.equal(bh[b].cpu(), bs) and torch.equal(ch[b].cpu(), cs)


def test_device_labels_vs_host_labels_on_many_rooms():
    """hn_labels_rasterise (csrc/labels.hip) against labels.make_labels (= reference dataset.py:84-120, pinned bit for bit by
    tests/golden/dataset.npz) on 300 synthetic rooms -- star-shaped and L-shaped with hidden corners -- under random stretch / flip /
    roll: every element within 1 float32 ulp, >= 99.9 % bit-identical, no panorama flagged; an open outline is rasterised on the host."""
    from horizonnet_amd import labels as lb
    from test_labels_device_cpu import rooms
    rs = rooms(300)
    imgs = np.zeros((1, 512, 1024, 3), np.uint8)
    batcher = ds.DeviceBatcher(images=np.repeat(imgs, len(rs), 0), corners=[c for c, _ in rs], device=DEV, flip=True, rotate=True, stretch=True)
    host = ds.DeviceBatcher(images=imgs, corners=[rs[0][0]], device=DEV, device_labels=False, flip=True, rotate=True, stretch=True)
    host.labels = batcher.labels
    worst, exact, total = 0, 0, 0
    for lo in range(0, len(rs), 50):
        idx = list(range(lo, lo + 50))
        _, augs, recs, on_host = batcher.host_half(idx, np.random.RandomState(lo))
        assert not on_host
        _, augs2, hb, hc = host.host_half(idx, np.random.RandomState(lo))
        assert augs == augs2
        _, bon, y_cor = batcher.device_half(idx, augs, recs, on_host)
        torch.cuda.synchronize()
        assert int(batcher.last_label_status.sum()) == 0
        for got, want in ((bon.cpu().numpy(), hb), (y_cor.cpu().numpy(), hc)):
            u = ulp_diff(got, want)
            worst = max(worst, int(u.max()))
            exact += int((u == 0).sum())
            total += u.size
    print("[parity] device labels vs host labels: %d of %d elements bit-identical, worst %d ulp" % (exact, total, worst))
    assert worst <= 1 and exact >= 0.999 * total
    # an outline that does not close: flagged by the host half, rasterised there, and the kernel agrees it could not
    open_cor = np.array([[100, 200], [100, 300], [400, 210], [400, 310]], np.float32)
    b2 = ds.DeviceBatcher(images=imgs, corners=[open_cor], device=DEV)
    idx, augs, recs, on_host = b2.host_half([0])
    assert list(on_host) == [0]
    _, bon, y_cor = b2.device_half(idx, augs, recs, on_host)
    torch.cuda.synchronize()
    assert int(b2.last_label_status[0]) == 1
    want_bon, want_cor, _ = ds.make_labels(open_cor, np.zeros(4, bool), augs[0], 512, 1024)
    assert np.array_equal(bon[0].cpu().numpy(), want_bon.astype(np.float32)) and np.array_equal(y_cor[0].cpu().numpy(), want_cor.astype(np.float32))


def test_batch_stream_equals_unthreaded_loop():
    """DeviceBatcher.stream (host half two batches ahead in a thread, pinned label staging ring) hands out exactly the
    batches the plain loop does for the same seed, survives more batches than the ring has slots, and raises the
    thread's exception in the consumer."""
    kw = dict(flip=True, rotate=True, gamma=True, stretch=True)
    batcher = ds.DeviceBatcher(ROOT, device=DEV, **kw)
    draws = [[(3 * i + j) % 4 for j in range(1 + i % 3)] for i in range(11)]          # ragged batch sizes, 11 > ring of 4
    rng = np.random.RandomState(5)
    want = [tuple(t.cpu() for t in batcher.batch(idx, rng)) for idx in draws]
    rng = np.random.RandomState(5)
    stream = batcher.stream(iter(draws), rng, depth=2)
    got = [tuple(t.cpu() for t in b) for b in stream]
    stream.close()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert all(torch.equal(a, b) for a, b in zip(g, w))

    bad = batcher.stream(iter([[0], [99]]), np.random.RandomState(1))                 # index 99 does not exist
    next(bad)
    with pytest.raises(IndexError):
        next(bad)
    bad.close()
    early = batcher.stream(iter([[0]] * 50), np.random.RandomState(1), depth=2)       # closing mid-way must not hang
    next(early)
    early.close()
    assert not early.thread.is_alive()


def test_augment_batch_contract_errors():
    batcher = ds.DeviceBatcher(ROOT, device=DEV)
    a = {"kx": 1.0, "ky": 1.0, "flip": 0, "rotate": False, "roll": 0, "gamma": 1.0}
    with pytest.raises(RuntimeError):
        ds.augment_images(batcher.data.cpu(), [0], [a])                 # host tensor: no CPU fallback
    with pytest.raises(RuntimeError):
        ds.augment_images(batcher.data, [99], [a])                      # index outside the dataset
    with pytest.raises(RuntimeError):
        ds.augment_images(batcher.data, [0], [dict(a, kx=-1.0)])
    assert ds.augment_images(batcher.data, [], []).shape == (0, 3, 512, 1024)
    big = ds.augment_images(batcher.data, [i % 4 for i in range(70)], [dict(a, roll=i, rotate=True) for i in range(70)])
    want = dataset_ref.augment_image(batcher.data[1].cpu().numpy(), roll=69)
    assert np.array_equal(big[69].cpu().numpy(), want)                   # second chunk of a > 64 batch


def test_images_to_input_equals_reference_conversion():
    """dataset.images_to_input (the augmentation kernel with every augmentation off) == reference inference.py:196-200:
    np.array(img)[..., :3].transpose(2, 0, 1) / 255 -> FloatTensor, bit for bit, for all 256 byte values."""
    rng = np.random.RandomState(1)
    imgs = rng.randint(0, 256, (3, 512, 1024, 3)).astype(np.uint8)
    imgs[0, 0, :256, 0] = np.arange(256)
    want = torch.FloatTensor(np.array([im.transpose(2, 0, 1) / 255 for im in imgs]))
    got = ds.images_to_input(torch.from_numpy(imgs).to(DEV), [0, 1, 2, 1])
    assert got.shape == (4, 3, 512, 1024) and torch.equal(got[:3].cpu(), want) and torch.equal(got[3].cpu(), want[1])
