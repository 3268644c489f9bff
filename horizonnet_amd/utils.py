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


def _rng_states():
    import numpy as np
    return {"torch": torch.get_rng_state(), "numpy": np.random.get_state()}


def gather_rng_states():
    """Every rank's host RNG streams (torch: dropout seeds of the training step; numpy: augmentation draws), rank-indexed.
    A COLLECTIVE when torch.distributed is initialised with more than one rank: call it on every rank (train.py does, just
    before rank 0 writes the checkpoint)."""
    import torch.distributed as dist
    mine = _rng_states()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        states = [None] * dist.get_world_size()
        dist.all_gather_object(states, mine)
        return states
    return [mine]


def make_checkpoint(net, optimizer, epoch, args, rng_ranks=None):
    """The dictionary of train.py:336-346, plus what a bit-continuous resume needs (iteration counter, RNG states of EVERY
    rank: `rng_ranks` from gather_rng_states(); default: this process only)."""
    core = unwrap(net)
    rng_ranks = [_rng_states()] if rng_ranks is None else rng_ranks
    return {"epoch": epoch, "state_dict": core.state_dict(), "optimizer": optimizer.state_dict(),
            "best_valid_score": args.best_valid_score, "backbone": getattr(core, "backbone", None),
            "use_rnn": getattr(core, "use_rnn", True), "cur_iter": args.cur_iter,
            "rng": rng_ranks[0], "rng_ranks": rng_ranks}


def _restore_rng(blob, rank, world):
    """Each rank continues ITS OWN streams.  A checkpoint without per-rank states (older file, or written by a different
    world size) cannot do that for ranks > 0: they are re-seeded with a deterministic function of (rank 0's saved state,
    rank) so that replicas stay decorrelated instead of all replaying rank 0's draws."""
    import numpy as np
    ranks = blob.get("rng_ranks")
    if ranks is not None and len(ranks) == world:
        torch.set_rng_state(ranks[rank]["torch"])
        np.random.set_state(ranks[rank]["numpy"])
        return "own"
    rng = blob.get("rng")
    if rng is None:
        return "none"
    if rank == 0:
        torch.set_rng_state(rng["torch"])
        np.random.set_state(rng["numpy"])
        return "own"
    import zlib
    base = zlib.crc32(rng["torch"].numpy().tobytes()) ^ zlib.crc32(np.asarray(rng["numpy"][1]).tobytes())
    seed = (base + 0x9E3779B1 * rank) & 0x7FFFFFFF
    torch.manual_seed(seed)
    np.random.seed(seed)
    return "reseeded"


def resume_checkpoint(path, net, optimizer, args, device, rank=None, world=None):
    """Restore model, optimiser, epoch / iteration counters and RNG streams from ``checkpoint.pth.tar``; returns the
    epoch to continue with.  Also accepts the reference's own checkpoints (torch.optim.Adam state, no 'cur_iter' / 'rng':
    the iteration counter is rebuilt from the epoch)."""
    if rank is None or world is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank() if on else 0
        world = dist.get_world_size() if on else 1
    blob = torch.load(path, map_location="cpu", weights_only=False)
    unwrap(net).load_state_dict(blob["state_dict"])
    optimizer.load_state_dict(blob["optimizer"])
    for state in getattr(optimizer, "state", {}).values():            # torch optimisers; FusedAdam moves its own buffers
        for k, v in state.items():
            if torch.is_tensor(v) and v.is_floating_point() and v.dim() > 0:
                state[k] = v.to(device)
    args.best_valid_score = blob.get("best_valid_score", 0)
    args.cur_iter = blob.get("cur_iter", blob["epoch"] * getattr(args, "iters_per_epoch", 0))
    args.rng_restore = _restore_rng(blob, rank, world)
    return int(blob["epoch"]) + 1
