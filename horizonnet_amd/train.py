"""Training entry point on the MI355X engine -- the flags, schedule, losses and checkpoint files of reference
``train.py`` (:61-352), re-shaped for one process per GPU:

    python -m horizonnet_amd.train --id run0 --train_root_dir data/train --valid_root_dir data/valid [--resume]
    python -m torch.distributed.run --nproc-per-node 8 -m horizonnet_amd.train --id run0 ...        # data parallel

Differences from the reference, all deliberate (SURVEY.md section 8 e / f2 / f4):

* the training set is decoded once into HBM and every batch is ONE fused augmentation launch (``DeviceBatcher``) instead
  of DataLoader worker processes; ``--num_workers`` is accepted and ignored;
* ``--multi_gpu`` / ``nn.DataParallel`` is replaced by ``torch.distributed`` over RCCL: every rank owns a shard of each
  epoch's permutation, gradients are all-reduced inside ``loss.backward()`` (``horizonnet_amd.parallel``);
* mixed precision is ``--precision bf16`` (bf16 matrix cores, float32 accumulation / master weights) instead of fp16
  autocast + ``GradScaler`` (train.py:227,273-280): no loss scaling is needed;
* ``--resume`` continues from ``<ckpt>/<id>/checkpoint.pth.tar`` (the reference writes that file, train.py:336-346, but
  never reads it back), including the iteration counter and the RNG streams, so the learning-rate schedule and the
  augmentation sequence continue where they stopped;
* the ``nn.DataParallel`` attribute slips of train.py:202,252,350 (``net.feature_extractor`` on a wrapped module) are gone.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import HorizonNet
from .dataset import DeviceBatcher, PanoCorBonDataset
from .inference import inference
from .evaluation import test_general
from . import utils
from .parallel import broadcast_module_


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--id", required=True, help="experiment id to name checkpoints and logs")
    p.add_argument("--ckpt", default="./ckpt", help="folder to output checkpoints")
    p.add_argument("--logs", default="./logs", help="folder to logging")
    p.add_argument("--pth", default=None, help="path to load saved checkpoint (finetuning)")
    p.add_argument("--backbone", default="resnet50", help="backbone of the network (the engine implements resnet50)")
    p.add_argument("--no_rnn", action="store_true", help="whether to remove rnn or not (not implemented by the engine)")
    p.add_argument("--train_root_dir", default="data/layoutnet_dataset/train")
    p.add_argument("--valid_root_dir", default="data/layoutnet_dataset/valid")
    p.add_argument("--no_flip", action="store_true")
    p.add_argument("--no_rotate", action="store_true")
    p.add_argument("--no_gamma", action="store_true")
    p.add_argument("--no_pano_stretch", action="store_true")
    p.add_argument("--num_workers", default=8, type=int, help="ignored: augmentation runs on the device")
    p.add_argument("--freeze_earlier_blocks", default=-1, type=int)
    p.add_argument("--batch_size_train", default=8, type=int, help="training mini-batch size PER GPU")
    p.add_argument("--batch_size_valid", default=2, type=int)
    p.add_argument("--epochs", default=300, type=int)
    p.add_argument("--optim", default="Adam", help="SGD or Adam")
    p.add_argument("--lr", default=1e-4, type=float)
    p.add_argument("--lr_pow", default=0.9, type=float)
    p.add_argument("--warmup_lr", default=1e-6, type=float)
    p.add_argument("--warmup_epochs", default=0, type=int)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--weight_decay", default=0, type=float)
    p.add_argument("--bn_momentum", type=float)
    p.add_argument("--no_cuda", action="store_true", help="not supported: the engine has no CPU path")
    p.add_argument("--multi_gpu", action="store_true", help="ignored: launch one process per GPU with torch.distributed.run")
    p.add_argument("--device", default="0", help="GPU index for a single-process run")
    p.add_argument("--seed", default=594277, type=int)
    p.add_argument("--disp_iter", type=int, default=1)
    p.add_argument("--save_every", type=int, default=25)
    # engine additions
    p.add_argument("--precision", choices=["f32", "bf16"], default="bf16", help="training arithmetic of the conv GEMMs")
    p.add_argument("--resume", action="store_true", help="continue from <ckpt>/<id>/checkpoint.pth.tar when it exists")
    p.add_argument("--stop_after_epoch", type=int, default=None, help="leave after this epoch (tests: simulated kill)")
    return p


class _FusedObjective(torch.autograd.Function):
    """train.py:53-56 as ONE HIP launch (hn_loss_l1_bce: both means, their sum and both gradients) + one launch in the backward
    (hn_scale2: the incoming adjoint, read on the device) -- instead of the ~12 elementwise / reduce launches of the two torch losses."""

    @staticmethod
    def forward(ctx, bon, y_bon, cor, y_cor):
        from . import _lib
        dev = bon.device
        bon, y_bon, cor, y_cor = bon.contiguous(), y_bon.contiguous(), cor.contiguous(), y_cor.contiguous()
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        dbon, dcor = torch.empty_like(bon), torch.empty_like(cor)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().hn_loss_l1_bce(_lib.ptr(bon), _lib.ptr(y_bon), bon.numel(), _lib.ptr(cor), _lib.ptr(y_cor), cor.numel(),
                                                  _lib.ptr(losses), _lib.ptr(total), _lib.ptr(dbon), _lib.ptr(dcor), _lib.stream_ptr(dev)), "hn_loss_l1_bce")
        ctx.save_for_backward(dbon, dcor)
        ctx.mark_non_differentiable(losses)
        return total, losses

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        from . import _lib
        dbon, dcor = ctx.saved_tensors
        if getattr(ctx, "consumed", False):      # hn_scale2 scales the saved gradients IN PLACE: a second pass would multiply them by g again
            raise RuntimeError("horizonnet_amd.train.objective: backward through the fused objective ran twice (retain_graph=True / a loss-scaling "
                               "wrapper that replays the graph); call objective() again, or use the two torch losses of reference train.py:53-56")
        ctx.consumed = True
        g = g_total.contiguous().float()
        with torch.cuda.device(dbon.device):
            _lib.check(_lib.load().hn_scale2(_lib.ptr(dbon), dbon.numel(), _lib.ptr(dcor), dcor.numel(), _lib.ptr(g), _lib.stream_ptr(dbon.device)),
                       "hn_scale2")
        return dbon, None, dcor, None


def objective(y_bon_, y_bon, y_cor_, y_cor):
    """-> {"bon", "cor", "total"} of reference train.py:53-56; "total" carries the gradient.  Float32 ROCm tensors go through the fused HIP
    objective (a second backward through the same graph raises: it scales its saved gradients in place); anything else (the outputs
    under torch.autocast arrive in the autocast dtype; CPU tensors in the host-side tests) through the reference's two torch calls."""
    if (y_bon_.is_cuda and y_bon_.dtype == torch.float32 and y_cor_.dtype == torch.float32 and y_bon.dtype == torch.float32 and y_cor.dtype == torch.float32
            and y_bon.shape == y_bon_.shape and y_cor.shape == y_cor_.shape and y_bon.device == y_bon_.device and y_cor.device == y_bon_.device):
        total, parts = _FusedObjective.apply(y_bon_, y_bon, y_cor_, y_cor)
        return {"bon": parts[0], "cor": parts[1], "total": total}
    losses = {"bon": F.l1_loss(y_bon_, y_bon), "cor": F.binary_cross_entropy_with_logits(y_cor_, y_cor)}
    losses["total"] = losses["bon"] + losses["cor"]
    return losses


def feed_forward(net, x, y_bon, y_cor):
    """train.py:44-58: L1 on the two boundaries + BCE-with-logits on the corner channel."""
    y_bon_, y_cor_ = net(x)
    return objective(y_bon_, y_bon, y_cor_, y_cor)


def freeze_blocks(net, upto, set_eval_only=False):
    """train.py:200-208 (requires_grad off, once) and :250-256 (eval() of the frozen blocks, every epoch)."""
    if upto < 0:
        return
    blocks = utils.unwrap(net).feature_extractor.list_blocks()
    for i in range(upto + 1):
        for m in blocks[i]:
            if set_eval_only:
                m.eval()
            else:
                for q in m.parameters():
                    q.requires_grad = False


def validate(net, dataset_valid, device):
    """train.py:290-325: per-sample objective + layout metrics of the raw 1024-vertex polygon."""
    net.eval()
    total = {}
    for j in range(len(dataset_valid)):
        x, y_bon, y_cor, gt_cor_id = dataset_valid[j]
        x, y_bon, y_cor = x[None].to(device), y_bon[None].to(device), y_cor[None].to(device)
        with torch.no_grad():
            losses = {k: float(v) for k, v in feed_forward(net, x, y_bon, y_cor).items()}
            acc = dict((n, {"2DIoU": [], "3DIoU": [], "rmse": [], "delta_1": []}) for n in ["4", "6", "8", "10+", "odd", "overall"])
            try:
                dt_cor_id = inference(net, x, device, force_raw=True)[0]
                dt_cor_id[:, 0] *= 1024
                dt_cor_id[:, 1] *= 512
            except Exception:
                dt_cor_id = np.array([[k // 2 * 1024, 256 - ((k % 2) * 2 - 1) * 120] for k in range(8)])
            test_general(dt_cor_id, gt_cor_id, 1024, 512, acc)
            for k in ("2DIoU", "3DIoU", "rmse", "delta_1"):
                losses[k] = float(np.mean(acc["overall"][k])) if acc["overall"][k] else 0.0
        for k, v in losses.items():
            total[k] = total.get(k, 0.0) + v
    return {k: v / max(1, len(dataset_valid)) for k, v in total.items()}


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.no_cuda:
        raise SystemExit("horizonnet_amd.train runs on the MI355X engine only (--no_cuda is not supported)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", args.device.split(",")[0]))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # generous collective time-out: a validation pass on rank 0 may outlast RCCL's default watchdog
        # (no device_id=: torch's eager RCCL init bound to the device costs the engine's HBM-bound bf16 kernels 4-5 %; the device is
        #  selected with torch.cuda.set_device before the first collective)
        dist.init_process_group(backend=os.environ.get("HN_DIST_BACKEND", "nccl"), timeout=datetime.timedelta(hours=2))
    np.random.seed(args.seed + rank)
    torch.manual_seed(args.seed + rank)
    ckpt_dir = os.path.join(args.ckpt, args.id)
    os.makedirs(ckpt_dir, exist_ok=True)

    train = DeviceBatcher(root_dir=args.train_root_dir, device=device, flip=not args.no_flip, rotate=not args.no_rotate,
                          gamma=not args.no_gamma, stretch=not args.no_pano_stretch)
    dataset_valid = None
    if args.valid_root_dir and rank == 0:
        dataset_valid = PanoCorBonDataset(args.valid_root_dir, return_cor=True, device=device)
    B = args.batch_size_train
    iters_per_epoch = len(train) // (B * world)
    if iters_per_epoch < 1:
        raise SystemExit("training set (%d) smaller than one global batch (%d)" % (len(train), B * world))

    if args.pth is not None:
        net = utils.load_trained_model(HorizonNet, args.pth).to(device)
    else:
        net = HorizonNet(args.backbone, not args.no_rnn).to(device)
    net.train_precision = args.precision
    broadcast_module_(net)
    freeze_blocks(net, args.freeze_earlier_blocks)
    if args.bn_momentum:
        for m in net.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                m.momentum = args.bn_momentum
    live = [q for q in net.parameters() if q.requires_grad]
    if args.optim == "SGD":
        optimizer = torch.optim.SGD(live, lr=args.lr, momentum=args.beta1, weight_decay=args.weight_decay)
    elif args.optim == "Adam":
        from .optim import FusedAdam           # torch.optim.Adam's arithmetic, one launch over the engine's flat gradient buffer
        optimizer = FusedAdam(net, lr=args.lr, betas=(args.beta1, 0.999), weight_decay=args.weight_decay)
    else:
        raise NotImplementedError()

    args.iters_per_epoch = iters_per_epoch
    args.warmup_iters = args.warmup_epochs * iters_per_epoch
    args.max_iters = args.epochs * iters_per_epoch
    args.running_lr = args.warmup_lr if args.warmup_epochs > 0 else args.lr
    args.cur_iter = 0
    args.best_valid_score = 0
    first_epoch = 1
    ckpt_file = os.path.join(ckpt_dir, "checkpoint.pth.tar")
    if args.resume and os.path.isfile(ckpt_file):
        first_epoch = utils.resume_checkpoint(ckpt_file, net, optimizer, args, device, rank=rank, world=world)
        if rank == 0:
            print("resumed from %s: continuing with epoch %d (iteration %d)" % (ckpt_file, first_epoch, args.cur_iter))

    history = []
    for epoch in range(first_epoch, args.epochs + 1):
        net.train()
        freeze_blocks(net, args.freeze_earlier_blocks, set_eval_only=True)
        perm = np.random.permutation(len(train))          # every rank draws the same permutation only if seeds agree:
        if dist is not None:                               # rank 0's is broadcast
            t = torch.from_numpy(perm).to(device)
            dist.broadcast(t, src=0)
            perm = t.cpu().numpy()
        # host half of every batch (augmentation draws, label rasterisation) two batches ahead in a thread: the role of the
        # reference's DataLoader workers (train.py:153-160); same np.random draw order as the unthreaded loop
        batches = train.stream([perm[(it * world + rank) * B:(it * world + rank) * B + B] for it in range(iters_per_epoch)], depth=2)
        for it in range(iters_per_epoch):
            utils.adjust_learning_rate(optimizer, args)
            args.cur_iter += 1
            x, y_bon, y_cor = next(batches)
            losses = feed_forward(net, x, y_bon, y_cor)
            optimizer.zero_grad(set_to_none=True)
            losses["total"].backward()
            optimizer.step()
            if args.cur_iter % args.disp_iter == 0 or it == iters_per_epoch - 1:
                row = {"epoch": epoch, "iter": args.cur_iter, "lr": args.running_lr,
                       "bon": float(losses["bon"]), "cor": float(losses["cor"])}
                history.append(row)
                if rank == 0:
                    print("ep %d it %d lr %.3e bon %.4f cor %.4f" % (epoch, args.cur_iter, args.running_lr, row["bon"], row["cor"]),
                          flush=True)
        batches.close()
        torch.cuda.synchronize(device)
        status = int(net.hip_status(device) != 0)
        bad_labels = train.unexpected_label_flags() if hasattr(train, "unexpected_label_flags") else 0
        if bad_labels:                                      # (same exit path as the status word: every rank leaves together; status 2 = labels)
            print("horizonnet_amd.train: %d panoramas of epoch %d were rasterised on the device with an uncovered column the host half had not "
                  "predicted (labels.device_label_record vs csrc/labels.hip disagree)" % (bad_labels, epoch), flush=True)
            status = 2
        if world > 1:                                       # EVERY rank must leave together: the collectives below would otherwise
            flag = torch.tensor([status], dtype=torch.int32, device=device)       # block the healthy ranks until the process-group time-out
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            status = int(flag.item())
        if status == 2:
            raise RuntimeError("horizonnet_amd.train: a rank's device label rasterisation flagged panoramas the host half had not predicted in epoch %d "
                               "(boundary labels of those rows would be wrong); refusing to go on -- see the rank's message above" % epoch)
        if status != 0:                                     # a timed-out persistent LSTM kernel produced garbage this epoch:
            raise RuntimeError("horizonnet_amd.train: the engine's status word is non-zero after epoch %d; refusing to "
                               "validate / checkpoint weights updated from invalid activations" % epoch)
        rng_ranks = utils.gather_rng_states()               # collective: every rank's streams go into rank 0's checkpoint
        if dataset_valid is not None:
            scores = validate(net, dataset_valid, device)
            now = scores.get("3DIoU", 0.0)
            print("Ep%3d %.4f vs. Best %.4f" % (epoch, now, args.best_valid_score))
            is_best = now > args.best_valid_score
            if is_best:
                args.best_valid_score = now
            utils.save_checkpoint(utils.make_checkpoint(net, optimizer, epoch, args, rng_ranks), is_best, ckpt_dir, epoch)
            history.append({"epoch": epoch, "valid": scores})
        elif rank == 0:
            utils.save_checkpoint(utils.make_checkpoint(net, optimizer, epoch, args, rng_ranks), False, ckpt_dir, epoch)
        if rank == 0 and epoch % args.save_every == 0:
            utils.save_model(net, os.path.join(ckpt_dir, "epoch_%d.pth" % epoch), args)
        if dist is not None:
            dist.barrier()                                  # nobody reads a checkpoint (or starts the next epoch's collectives) before rank 0 has written it
        if args.stop_after_epoch is not None and epoch >= args.stop_after_epoch:
            break
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return history


if __name__ == "__main__":
    main(sys.argv[1:])
