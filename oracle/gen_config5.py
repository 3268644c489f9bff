"""BASELINE configs[4] ("config 5") reference side: the UNMODIFIED reference ``inference.inference()``
(``/root/reference/inference.py:65-141``, default general-layout path) on the 1000 seeded synthetic
Structured3D-shaped panoramas of ``tools/c5_common.py``, with the briefly-trained checkpoint committed as
``tests/golden/config5/ckpt_q.npz`` loaded into the unmodified reference ``model.HorizonNet``.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs ``/root/reference``):

    python -m oracle.gen_config5 [N]        # ~10 min of CPU for N = 1000

Writes ``tests/golden/config5/reference_layouts.npz``: per panorama the reference's ``cor_id`` (normalised corner list),
``z0``/``z1``, a CRC of the rendered input (so the GPU box can prove it rendered the same pixels), float64 sums of the
raw network signals, and the full signals for the first 64 panoramas.
"""
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "standins"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from tools import c5_common as c5  # noqa: E402

FULL_SIGNALS = 64


class Recorder:
    """Callable around the reference network that keeps the last raw outputs (inference() does not return them)."""

    def __init__(self, net):
        self.net = net
        self.last = None

    def __call__(self, x):
        out = self.net(x)
        self.last = (out[0].detach().numpy().copy(), out[1].detach().numpy().copy())
        return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    import model as ref_model                    # reference model.py, unmodified (torchvision stand-in on sys.path)
    import inference as ref_inf                  # reference inference.py, unmodified (shapely stand-in on sys.path)
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    sd = c5.decode_state_dict()
    net = ref_model.HorizonNet("resnet50", True)
    net.load_state_dict(sd, strict=True)
    net.eval()
    rec = Recorder(net)
    cor_ids, counts, z1s, crcs, sums, sig_bon, sig_cor, fallbacks = [], [], [], [], [], [], [], 0
    t0 = time.perf_counter()
    err = sys.stderr
    for i in range(n):
        img, _ = c5.make_room(c5.room_jobs(1, c5.VAL_SEED0, i)[0])
        x = torch.FloatTensor(np.array([img.transpose(2, 0, 1) / 255]))         # inference.py:199-200
        with torch.no_grad():
            cor_id, z0, z1, _ = ref_inf.inference(net=rec, x=x, device="cpu")   # defaults of inference.py:144-170
        assert z0 == 50
        cor_ids.append(np.asarray(cor_id, np.float32))
        counts.append(len(cor_id))
        z1s.append(float(z1))
        crcs.append(c5.image_crc(img))
        bon, cor = rec.last
        sums.append([float(bon.astype(np.float64).sum()), float(cor.astype(np.float64).sum())])
        if i < FULL_SIGNALS:
            sig_bon.append(bon[0])
            sig_cor.append(cor[0])
        if i % 25 == 0:
            print("%d/%d  %.1f s" % (i, n, time.perf_counter() - t0), file=err, flush=True)
    out = os.path.join(ROOT, "tests", "golden", "config5", "reference_layouts.npz")
    np.savez_compressed(out, cor_id=np.concatenate(cor_ids), count=np.array(counts, np.int32), z1=np.array(z1s, np.float64),
                        crc=np.array(crcs, np.uint32), signal_sum=np.array(sums, np.float64),
                        bon=np.stack(sig_bon), cor=np.stack(sig_cor), n=np.int64(n), seed0=np.int64(c5.VAL_SEED0))
    print("wrote %s: %d panoramas, corner-count histogram %s, %.0f s" % (
        out, n, dict(zip(*np.unique(np.array(counts) // 2, return_counts=True))), time.perf_counter() - t0))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "cli"):
    main()


def gen_inference_cli():
    """The reference's own ``inference.py`` run AS A SCRIPT (unmodified, stand-ins on PYTHONPATH) on the committed synthetic
    dataset's images with the config-5 checkpoint in ``save_model`` format: the JSON files it writes are the expected
    outputs of tests/test_gpu_integration.py::test_inference_entry_point_sequence_matches_reference_json."""
    import argparse
    import glob
    import json
    import shutil
    import subprocess
    import tempfile
    from collections import OrderedDict
    out_dir = os.path.join(ROOT, "tests", "golden", "config5", "inference_cli")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    tmp = tempfile.mkdtemp()
    pth = os.path.join(tmp, "c5.pth")
    torch.save(OrderedDict([("args", {"id": "config5"}), ("kwargs", {"backbone": "resnet50", "use_rnn": True}),
                            ("state_dict", c5.decode_state_dict())]), pth)                     # misc/utils.py:49-58 format
    flags = {"flip": True, "rotate": [0.25]}
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(HERE, "standins"), REF]))
    cmd = [sys.executable, os.path.join(REF, "inference.py"), "--pth", pth, "--img_glob",
           os.path.join(ROOT, "tests", "golden", "synth_ds", "img", "*.png"), "--output_dir", out_dir, "--no_cuda", "--flip",
           "--rotate", "0.25"]
    subprocess.check_call(cmd, env=env, cwd=tmp)
    with open(os.path.join(out_dir, "args.json"), "w") as f:
        json.dump(flags, f)
    print("inference.py wrote:", sorted(os.path.basename(p) for p in glob.glob(os.path.join(out_dir, "*.json"))))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cli":
    gen_inference_cli()
