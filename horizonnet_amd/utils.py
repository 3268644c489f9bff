"""Training utilities with the reference's names and on-disk formats (``misc/utils.py:35-65``): the polynomial
learning-rate schedule with warm-up, ``save_model`` / ``load_trained_model`` (the ``{'args', 'kwargs', 'state_dict'}``
checkpoint every reference tool reads), and the rolling ``checkpoint.pth.tar`` of ``train.py:33-37,336-346`` together
with the resume the reference never implemented."""
import os
import shutil
from collections import OrderedDict

import torch


def adjust_learning_rate(optimizer, args):
    """Linear warm-up from ``warmup_lr`` over ``warmup_iters`` iterations, then ``lr * (1 - progress) ** lr_pow``
    (misc/utils.py:35-46).  Reads ``args.cur_iter``, writes ``args.running_lr`` and every param group's lr."""
    it, warm = args.cur_iter, args.warmup_iters
    if it < warm:
        args.running_lr = args.warmup_lr + (args.lr - args.warmup_lr) * (it / warm)
    else:
        progress = (float(it) - warm) / (args.max_iters - warm)
        args.running_lr = args.lr * max(1.0 - progress, 0.0) ** args.lr_pow
    for group in optimizer.param_groups:
        group["lr"] = args.running_lr


def unwrap(net):
    """The HorizonNet inside an ``nn.DataParallel`` / DDP wrapper (the reference forgets this at train.py:202,252,350)."""
    return net.module if hasattr(net, "module") else net


def save_model(net, path, args):
    """misc/utils.py:49-58 format: what ``inference.py --pth`` and ``train.py --pth`` load."""
    core = unwrap(net)
    torch.save(OrderedDict([("args", dict(vars(args))),
                            ("kwargs", {"backbone": core.backbone, "use_rnn": core.use_rnn}),
                            ("state_dict", core.state_dict())]), path)


def load_trained_model(Net, path):
    """misc/utils.py:61-65."""
    blob = torch.load(path, map_location="cpu")
    net = Net(**blob["kwargs"])
    net.load_state_dict(blob["state_dict"])
    return net


def save_checkpoint(state, is_best, checkpoint_dir, epoch):
    """train.py:33-37: rolling ``checkpoint.pth.tar`` (+ ``best_model_<epoch>.pth.tar``).  Written to a temporary name
    first so that a job killed mid-write leaves the previous checkpoint intact."""
    final = os.path.join(checkpoint_dir, "checkpoint.pth.tar")
    tmp = final + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, final)
    if is_best:
        shutil.copyfile(final, os.path.join(checkpoint_dir, "best_model_%d.pth.tar" % epoch))


def make_checkpoint(net, optimizer, epoch, args):
    """The dictionary of train.py:336-346, plus what a bit-continuous resume needs (iteration counter, RNG states)."""
    core = unwrap(net)
    return {"epoch": epoch, "state_dict": core.state_dict(), "optimizer": optimizer.state_dict(),
            "best_valid_score": args.best_valid_score, "backbone": getattr(core, "backbone", None),
            "use_rnn": getattr(core, "use_rnn", True), "cur_iter": args.cur_iter,
            "rng": {"torch": torch.get_rng_state(), "numpy": __import__("numpy").random.get_state()}}


def resume_checkpoint(path, net, optimizer, args, device):
    """Restore model, optimiser, epoch / iteration counters and RNG streams from ``checkpoint.pth.tar``; returns the
    epoch to continue with.  Also accepts the reference's own checkpoints (no 'cur_iter' / 'rng': the iteration counter
    is rebuilt from the epoch)."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    unwrap(net).load_state_dict(blob["state_dict"])
    optimizer.load_state_dict(blob["optimizer"])
    for state in getattr(optimizer, "state", {}).values():            # torch optimisers; FusedAdam moves its own buffers
        for k, v in state.items():
            if torch.is_tensor(v) and v.is_floating_point() and v.dim() > 0:
                state[k] = v.to(device)
    args.best_valid_score = blob.get("best_valid_score", 0)
    args.cur_iter = blob.get("cur_iter", blob["epoch"] * getattr(args, "iters_per_epoch", 0))
    rng = blob.get("rng")
    if rng is not None:
        import numpy as np
        torch.set_rng_state(rng["torch"])
        np.random.set_state(rng["numpy"])
    return int(blob["epoch"]) + 1
