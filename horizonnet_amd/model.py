"""``HorizonNet(nn.Module)`` -- same constructor, attributes, 448-key ``state_dict`` and
``forward`` contract as reference ``model.py:185-281``; the forward runs in the HIP engine.

The module tree below holds ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.LSTM`` /
``nn.Linear`` objects purely as PARAMETER CONTAINERS, so that
``misc/utils.load_trained_model`` (reference ``misc/utils.py:61-65``), ``save_model``
(``:49-58``), ``feature_extractor.list_blocks()`` freezing (``train.py:200-208``), the
``bn_momentum`` loop (``train.py:210-213``) and ``flatten_parameters`` (``train.py:39-42``)
keep working unchanged.  None of the containers' own ``forward`` methods is ever called.
"""
import ctypes
import os
import warnings

import torch
import torch.nn as nn

from . import _lib

_STAGES = ((64, 3), (128, 4), (256, 6), (512, 3))     # (bottleneck width, blocks) of ResNet-50
_SYNC_BYTES = 4096                                    # HN_SYNC_WORDS * 4: first bytes of every workspace
_STATUS_BYTE = 512 * 4                                # HN_STATUS_WORD * 4


class LR_PAD(nn.Module):
    """Marker for circular left/right padding (reference model.py:32-39).  The padding itself is
    index arithmetic inside the HIP convolution's tile loader; this module only keeps the
    reference's ``Sequential(LR_PAD, Conv2d)`` naming (``...conv2.1.weight``)."""

    def __init__(self, padding=1):
        super().__init__()
        self.padding = padding

    def forward(self, x):
        raise RuntimeError("LR_PAD is a parameter-tree marker; the HIP engine applies the circular padding")


def _lr_conv(cin, cout, k, stride, bias):
    p = k // 2
    return nn.Sequential(LR_PAD(p), nn.Conv2d(cin, cout, k, stride=stride, padding=(p, 0), bias=bias))


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _lr_conv(width, width, 3, stride, False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(width * 4))


class _ResNet50Trunk(nn.Module):
    """Children in torchvision's order (conv1, bn1, relu, maxpool, layer1..4) because
    ``list_blocks`` slices ``children()`` positionally (reference model.py:84-91)."""

    def __init__(self):
        super().__init__()
        self.conv1 = _lr_conv(3, 64, 7, 2, False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for i, (width, n) in enumerate(_STAGES):
            blocks = []
            for j in range(n):
                blocks.append(_Bottleneck(cin, width, 2 if (j == 0 and i > 0) else 1, j == 0))
                cin = width * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def _torchvision_resnet50_state_dict():
    """ImageNet weights of torchvision's resnet50 (``ResNet50_Weights.IMAGENET1K_V1``, what reference model.py:66-69,
    204-207 builds the encoder from), or None.  Sources, in order: the file named by ``HORIZONNET_RESNET50_WEIGHTS``;
    torch hub's cache (``resnet50-0676ba61.pth``); an installed torchvision (which may download)."""
    cands = [os.environ.get("HORIZONNET_RESNET50_WEIGHTS"),
             os.path.join(torch.hub.get_dir(), "checkpoints", "resnet50-0676ba61.pth")]
    for path in cands:
        if path and os.path.isfile(path):
            return torch.load(path, map_location="cpu")
    try:
        import torchvision.models as tvm
        try:
            return tvm.resnet50(weights=tvm.ResNet50_Weights.IMAGENET1K_V1).state_dict()
        except AttributeError:                                   # torchvision < 0.13 (model.py:10-14)
            return tvm.resnet50(pretrained=True).state_dict()
    except Exception:
        return None


def _remap_torchvision_key(k):
    """torchvision key -> key of the LR_PAD-wrapped trunk (reference model.py:42-55 turns every Conv2d with horizontal
    padding into Sequential(LR_PAD, conv): conv1 and every bottleneck conv2 gain a '.1')."""
    if k.startswith("fc."):
        return None
    if k == "conv1.weight":
        return "conv1.1.weight"
    if k.endswith(".conv2.weight"):
        return k[:-len("weight")] + "1.weight"
    return k


class Resnet(nn.Module):
    def __init__(self, backbone="resnet50", pretrained=True):
        super().__init__()
        self.encoder = _ResNet50Trunk()
        self.pretrained_loaded = False
        if pretrained:
            tv = _torchvision_resnet50_state_dict()
            if tv is None:
                warnings.warn(
                    "horizonnet_amd: no ImageNet ResNet-50 weights found (torchvision is not installed, "
                    "HORIZONNET_RESNET50_WEIGHTS is unset and torch hub's cache has no resnet50-0676ba61.pth): the encoder "
                    "starts from Kaiming-random weights, unlike reference model.py:204-207.  Harmless when a checkpoint is "
                    "loaded afterwards (inference.py / --pth); training from scratch will be worse than the reference's.",
                    RuntimeWarning, stacklevel=3)
            else:
                sd = {}
                for k, v in tv.items():
                    nk = _remap_torchvision_key(k)
                    if nk is not None:
                        sd[nk] = v
                self.encoder.load_state_dict(sd, strict=True)
                self.pretrained_loaded = True

    def list_blocks(self):
        ch = list(self.encoder.children())
        return ch[:4], ch[4:5], ch[5:6], ch[6:7], ch[7:8]


class ConvCompressH(nn.Module):
    def __init__(self, in_c, out_c, ks=3):
        super().__init__()
        assert ks % 2 == 1
        self.layers = nn.Sequential(_lr_conv(in_c, out_c, ks, (2, 1), True), nn.BatchNorm2d(out_c),
                                    nn.ReLU(inplace=True))


class GlobalHeightConv(nn.Module):
    def __init__(self, in_c, out_c):
        super().__init__()
        self.layer = nn.Sequential(ConvCompressH(in_c, in_c // 2), ConvCompressH(in_c // 2, in_c // 2),
                                   ConvCompressH(in_c // 2, in_c // 4), ConvCompressH(in_c // 4, out_c))


class GlobalHeightStage(nn.Module):
    def __init__(self, c1, c2, c3, c4, out_scale=8):
        super().__init__()
        self.cs = c1, c2, c3, c4
        self.out_scale = out_scale
        self.ghc_lst = nn.ModuleList([GlobalHeightConv(c, c // out_scale) for c in self.cs])


class _DeviceState:
    """Per-device engine handle + packed weights + workspaces."""

    def __init__(self, device):
        self.lib = _lib.load()
        self.device = device
        h = ctypes.c_void_p()
        _lib.check(self.lib.hn_create(ctypes.byref(h), device.index), "hn_create")
        self.handle = h
        self.packed = torch.empty(self.lib.hn_packed_bytes(), dtype=torch.uint8, device=device)
        self.packed_h = None               # bf16 conv / input-GEMM weights, allocated on first bf16 forward
        self.signature_h = None
        self.signature = None
        self.ptr_signature = None
        self.keepalive = None
        self.workspaces = {}                 # kind ("f32" | "bf16" | "train") -> (B, tensor): one batch size per kind
        self.train_generation = 0            # bumped whenever the train workspace's saved activations are overwritten / freed
        self.bn_flags = None                 # last per-BatchNorm eval flags pushed with hn_set_bn_eval
        self.status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.status_event = None

    def _workspace(self, kind, B, nbytes):
        cur = self.workspaces.get(kind)
        if cur is not None and cur[0] == B:
            return cur[1]
        if kind == "train":
            self.train_generation += 1       # a pending backward of the old buffer must not run on the new one
        self.workspaces.pop(kind, None)      # one batch size resident per kind; the other kinds stay (a validation forward
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)      # between a train forward and its backward is fine)
        ws[:_SYNC_BYTES].zero_()             # arrival counters + the STICKY status word of the persistent LSTM
        if kind in ("bf16p", "f32p"):        # the second head slot's sync area lies behind the plain plan
            off = self.pipelined_status_bytes(B, kind == "bf16p")[1] - _STATUS_BYTE
            ws[off:off + _SYNC_BYTES].zero_()
        self.workspaces[kind] = (B, ws)
        return ws

    def workspace(self, B):
        return self._workspace("f32", B, self.lib.hn_workspace_bytes(B))

    def workspace_bf16(self, B):
        return self._workspace("bf16", B, self.lib.hn_workspace_bf16_bytes(B))

    def train_workspace(self, B):
        return self._workspace("train", B, self.lib.hn_train_workspace_bytes(B))

    def workspace_pipelined(self, B, bf16):
        if bf16:
            return self._workspace("bf16p", B, self.lib.hn_workspace_bf16_pipelined_bytes(B))
        return self._workspace("f32p", B, self.lib.hn_workspace_pipelined_bytes(B))

    def pipelined_status_bytes(self, B, bf16=True):
        """Byte offsets of the two head slots' sticky status words inside the pipelined workspace."""
        fn = self.lib.hn_pipelined_status_offset if bf16 else self.lib.hn_pipelined_status_offset_f32
        offs = []
        for slot in (0, 1):
            o = ctypes.c_size_t()
            _lib.check(fn(B, slot, ctypes.byref(o)), "hn_pipelined_status_offset")
            offs.append(int(o.value))
        return offs

    def post_status_read(self, ws, byte_off=_STATUS_BYTE):
        """Asynchronous copy of the workspace's sticky LSTM status word to pinned host memory (no synchronisation)."""
        self.status_host.copy_(ws[byte_off:byte_off + 4].view(torch.int32), non_blocking=True)
        if self.status_event is None:
            self.status_event = torch.cuda.Event()
        self.status_event.record(torch.cuda.current_stream(self.device))

    def raise_if_failed(self):
        """Called at the start of every engine call: if an EARLIER call's status word has arrived and is non-zero, the
        persistent LSTM kernel hit its spin limit (its workgroups were not co-resident: CUs held by another stream or
        process, a partitioned GPU) and that call's bon / cor were garbage."""
        if self.status_event is not None and self.status_event.query() and int(self.status_host[0]) != 0:
            raise _lib.HipEngineError(
                "the persistent bi-LSTM kernel of an earlier forward timed out waiting for its peer workgroups (status "
                "word %d): its outputs were invalid.  The kernel needs all 256 workgroups co-resident -- do not share the "
                "GPU between processes / concurrent streams while it runs." % int(self.status_host[0]))

    def __del__(self):
        try:
            self.lib.hn_destroy(self.handle)
        except Exception:
            pass


class PendingForward:
    """Handle of one ``HorizonNet.forward_async`` call.  ``result()`` makes the CURRENT stream wait for the call's recurrent
    head (a stream-side wait, the host does not block) and returns (bon, cor) -- the reference's ``net(x)`` outputs."""

    def __init__(self, st, slot, bon, cor, ws, status_off, check, bf16=True):
        self._st, self._slot, self._bon, self._cor, self._ws = st, slot, bon, cor, ws
        self._status_off, self._check, self._done, self._bf16 = status_off, check, slot is None, bf16

    def result(self):
        if not self._done:
            dev = self._bon.device
            with torch.cuda.device(dev):
                collect = self._st.lib.hn_forward_bf16_collect if self._bf16 else self._st.lib.hn_forward_collect
                _lib.check(collect(self._st.handle, self._slot, _lib.stream_ptr(dev)), "hn_forward_collect")
                if self._check:
                    self._st.post_status_read(self._ws, self._status_off)
            self._done = True
        return _follow_autocast(self._bon, self._cor)          # same dtype rule as forward() (the pipelined entry is forward() split in two)

    def __del__(self):
        # bon / cor / the workspace are torch allocations of the CALLER's stream, written by the engine-owned head stream the caching
        # allocator knows nothing about: a handle dropped without result() must not hand them back while the head kernels still write
        if not getattr(self, "_done", True):
            try:
                self.result()
            except Exception:                                # interpreter shutdown, destroyed engine: nothing left to protect
                pass


class _HipTrainStep(torch.autograd.Function):
    """autograd node of one train-mode forward on the HIP engine (hn_train_forward / hn_train_backward)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        st = net._hip_state(x.device)
        B, C_in = int(x.shape[0]), int(x.shape[1])
        if net.train_precision not in ("f32", "bf16"):
            raise ValueError("train_precision must be 'f32' or 'bf16'")
        bf16 = net.train_precision == "bf16"
        if bf16:
            net._pack_bf16(st, x.device)
        _lib.check(st.lib.hn_set_train_precision(st.handle, int(bf16)), "hn_set_train_precision")
        st.raise_if_failed()
        ws = st.train_workspace(B)
        bon = torch.empty((B, 2, 1024), dtype=torch.float32, device=x.device)
        cor = torch.empty((B, 1, 1024), dtype=torch.float32, device=x.device)
        named_bns = [(k, m) for k, m in net.named_modules() if isinstance(m, nn.BatchNorm2d)]
        bns = [m for _, m in named_bns]
        # a BatchNorm in eval() inside a training net (train.py:245-256: frozen blocks) uses its running statistics
        flags = tuple(not m.training for m in bns)
        net._push_bn_flags(st, named_bns, flags)
        live = [m for m in bns if m.training]
        momentum = live[0].momentum if (live and live[0].momentum is not None) else 0.1
        seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())        # host RNG: follows torch.manual_seed
        p_rnn, p_head = float(net.bi_rnn.dropout), float(net.drop_out.p)
        _lib.check(st.lib.hn_train_forward(st.handle, _lib.ptr(x), B, C_in, _lib.ptr(bon), _lib.ptr(cor), _lib.ptr(ws), ws.numel(),
                                           p_rnn, p_head, float(momentum), seed, _lib.stream_ptr(x.device)), "hn_train_forward")
        if live:
            torch._foreach_add_([m.num_batches_tracked for m in live], 1)
        net._train_steps += 1                                    # running stats changed under torch's feet
        st.train_generation += 1
        st.post_status_read(ws)
        ctx.net, ctx.B, ctx.seed, ctx.p, ctx.bf16 = net, B, seed, (p_rnn, p_head), bf16
        ctx.generation, ctx.bn_flags, ctx.named_bns = st.train_generation, flags, named_bns
        ctx.names = net._param_names
        ctx.needs = [p.requires_grad for p in params]
        return bon, cor

    @staticmethod
    def backward(ctx, dbon, dcor):
        net = ctx.net
        dev = dbon.device if dbon is not None else dcor.device
        st = net._hip_states[dev.index]
        B = ctx.B
        dbon = torch.zeros((B, 2, 1024), device=dev) if dbon is None else dbon.contiguous().float()
        dcor = torch.zeros((B, 1, 1024), device=dev) if dcor is None else dcor.contiguous().float()
        if st.train_generation != ctx.generation:
            raise RuntimeError(
                "horizonnet_amd: the activations saved by this train-mode forward were overwritten before backward() ran "
                "(another train-mode forward, or a different batch size, used the engine's training workspace in between). "
                "Call backward() before the next training forward -- e.g. accumulate gradients step by step instead of "
                "summing the losses of two forwards.")
        st.raise_if_failed()
        ws = st.train_workspace(B)
        net._push_bn_flags(st, ctx.named_bns, ctx.bn_flags)
        _lib.check(st.lib.hn_set_train_precision(st.handle, int(ctx.bf16)), "hn_set_train_precision")   # as in the forward that saved the tensors
        flat = torch.empty(st.lib.hn_grad_floats(), dtype=torch.float32, device=dev)
        import ctypes
        import torch.distributed as dist
        world = dist.get_world_size(net.process_group) if (net.sync_gradients and dist.is_available() and dist.is_initialized()) else 1
        args = (st.handle, _lib.ptr(dbon), _lib.ptr(dcor), B, _lib.ptr(ws), ws.numel(), _lib.ptr(flat), ctx.p[0], ctx.p[1], ctx.seed)
        with torch.cuda.device(dev):
            if world == 1 and not getattr(net, "segmented_backward", False):
                _lib.check(st.lib.hn_train_backward(*args, _lib.stream_ptr(dev)), "hn_train_backward")
            else:
                # data-parallel replicas: the backward runs in gradient-completion segments (Linear + LSTM, height
                # compression, layer4, layer3, layer2..stem); each finished range of the flat buffer starts its RCCL
                # all-reduce at once, overlapping with the segments still to run
                from .parallel import allreduce_sum_async
                works, casts = [], []
                half_wire = getattr(net, "allreduce_dtype", "f32") == "bf16" and world > 1
                # net.allreduce_trace = [] asks for the timing of ONE backward: device events at the start, after every segment's
                # kernels, and (on a side stream that waits for the collective) at the end of every bucket -- bench.py turns them into
                # the per-bucket table of its allreduce_overlap block.  Off (None) in normal steps: nothing is recorded.
                trace = getattr(net, "allreduce_trace", None)
                if trace is not None:
                    side = getattr(net, "_trace_stream", None)
                    if side is None:
                        side = net._trace_stream = torch.cuda.Stream(device=dev)
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    trace.append({"kind": "start", "event": ev0})
                lo, cnt = ctypes.c_int64(), ctypes.c_int64()
                for seg in range(st.lib.hn_grad_segments()):
                    _lib.check(st.lib.hn_train_backward_segment(*args, seg, _lib.stream_ptr(dev)), "hn_train_backward_segment")
                    _lib.check(st.lib.hn_grad_segment_range(seg, ctypes.byref(lo), ctypes.byref(cnt)), "hn_grad_segment_range")
                    part = flat[lo.value:lo.value + cnt.value]
                    if half_wire:                                    # 163 MB instead of 326 MB over xGMI (net.allreduce_dtype = "bf16")
                        half = part.to(torch.bfloat16)
                        casts.append((part, half))
                        part = half
                    new_works = allreduce_sum_async(part, net.process_group)
                    if trace is not None:
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record()                                  # this segment's kernels (and the cast) are done: its buckets may start
                        trace.append({"kind": "segment", "segment": seg, "event": ev, "bytes": part.numel() * part.element_size()})
                        per = max(1, (64 << 20) // part.element_size())
                        for k, w in enumerate(new_works):
                            with torch.cuda.stream(side):
                                w.wait()                             # the side stream waits for the collective; the compute stream does not
                                evd = torch.cuda.Event(enable_timing=True)
                                evd.record(side)
                            trace.append({"kind": "bucket", "segment": seg, "bucket": k, "event": evd,
                                          "bytes": min(per, part.numel() - k * per) * part.element_size()})
                    works += new_works
                for w in works:
                    w.wait()
                for full, half in casts:                             # bf16 exchange: the summed halves back into the f32 buffer
                    full.copy_(half)
                net._grad_mean_scale = 1.0
                if world > 1:
                    ref = getattr(net, "_fused_adam_ref", None)
                    if getattr(net, "defer_grad_mean", False) and ref is not None and ref() is not None:   # a LIVE FusedAdam folds the 1/N into hn_adam_step's grad_scale
                        net._grad_mean_scale = 1.0 / world
                    else:
                        flat.mul_(1.0 / world)
        grads = []
        for (name, shape), need in zip(ctx.names, ctx.needs):
            if not need:
                grads.append(None)
                continue
            off = st.lib.hn_grad_offset(name.encode())
            n = 1
            for d in shape:
                n *= d
            grads.append(flat[off:off + n].view(shape))
        return (None, None) + tuple(grads)


def _follow_autocast(bon, cor):
    """"dtype follows autocast".  The reference's last op under ``torch.autocast`` (train.py:51,273) is the
    autocast ``nn.Linear`` (model.py:266), so its outputs arrive in the autocast dtype; the engine computes its head in float32
    and casts the two outputs here (differentiably), so that code written against the reference sees the same dtypes."""
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
        return bon.to(dt), cor.to(dt)
    return bon, cor


def _guard_data_parallel():
    """``nn.DataParallel(net, device_ids=[0, 1, ...])`` (reference train.py:190-192) replicates the module onto several GPUs inside ONE
    process -- the pattern this engine replaces with one process per GPU over RCCL.  Fail where the user
    writes the wrap, with the command to run instead, not at the first forward of a replica."""
    import torch.nn as nn
    if getattr(nn.DataParallel.__init__, "_horizonnet_amd_guard", False):
        return
    orig = nn.DataParallel.__init__

    def checked_init(self, module, *args, **kwargs):          # (signature-agnostic: whatever torch's DataParallel takes is passed on untouched)
        device_ids = kwargs.get("device_ids", args[0] if args else None)
        ids = device_ids if device_ids is not None else list(range(torch.cuda.device_count()))
        if isinstance(module, HorizonNet) and len(ids) > 1:
            raise RuntimeError(
                "horizonnet_amd.HorizonNet cannot be wrapped by nn.DataParallel over %d devices: the HIP engine runs one process per "
                "GPU.  Launch the same script with\n    python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                "train.py ...\n(horizonnet_amd.train / horizonnet_amd.parallel shard the batches and all-reduce the gradients over RCCL); "
                "nn.DataParallel(net, device_ids=[one device]) keeps working." % (len(ids), len(ids)))
        orig(self, module, *args, **kwargs)

    checked_init._horizonnet_amd_guard = True
    nn.DataParallel.__init__ = checked_init


class HorizonNet(nn.Module):
    x_mean = torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    x_std = torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)

    def __init__(self, backbone, use_rnn):
        super().__init__()
        if backbone != "resnet50" or not use_rnn:
            # north_star scope: resnet50_rnn only; fail like the reference does for
            # unknown encoders (model.py:204) instead of silently building something else.
            raise NotImplementedError("the MI355X engine implements backbone='resnet50', use_rnn=True only")
        self.backbone = backbone
        self.use_rnn = use_rnn
        self.out_scale = 8
        self.step_cols = 4
        self.rnn_hidden_size = 512
        self.feature_extractor = Resnet(backbone)
        c1, c2, c3, c4 = 256, 512, 1024, 2048
        c_last = (c1 * 8 + c2 * 4 + c3 * 2 + c4 * 1) // self.out_scale
        self.reduce_height_module = GlobalHeightStage(c1, c2, c3, c4, self.out_scale)
        self.bi_rnn = nn.LSTM(input_size=c_last, hidden_size=self.rnn_hidden_size, num_layers=2, dropout=0.5,
                              batch_first=False, bidirectional=True)
        self.drop_out = nn.Dropout(0.5)
        self.linear = nn.Linear(2 * self.rnn_hidden_size, 3 * self.step_cols)
        with torch.no_grad():                                  # head bias init, reference model.py:231-233
            self.linear.bias[0 * self.step_cols:1 * self.step_cols].fill_(-1)
            self.linear.bias[1 * self.step_cols:2 * self.step_cols].fill_(-0.478)
            self.linear.bias[2 * self.step_cols:3 * self.step_cols].fill_(0.425)
        self._hip_states = {}
        self._engine_options = {}
        self._train_steps = 0
        self.precision = "f32"              # eval mode: "f32" (exact, the parity path) or "bf16"
        self.train_precision = "f32"        # train mode: "f32", or "bf16" = forward + data-gradient convs on the bf16 matrix
                                            # cores (f32 accumulation, BN / weight gradients / LSTM / master weights stay f32)
        self.check_status_async = True      # every forward posts an asynchronous read of the LSTM status word; the NEXT call raises
        self.sync_gradients = True          # all-reduce gradients over torch.distributed when it is initialised (world > 1)
        self.allreduce_dtype = "f32"        # "bf16": the gradient ranges travel as bf16 (half the xGMI bytes; the sum is rounded to bf16)
        self.defer_grad_mean = False        # True (set by FusedAdam): backward leaves the SUM, the optimiser applies 1/world in its launch
        self._grad_mean_scale = 1.0
        self.process_group = None
        self._param_names = [(k, tuple(p.shape)) for k, p in self.named_parameters()]

    # ---- engine plumbing --------------------------------------------------------------------
    def _state_tensors(self):
        """(key, tensor) of every bound state tensor, in state_dict order.  The module tree is fixed after construction, so the
        (owner module, kind, name, key) slots are resolved once; per call this is 379 dictionary look-ups instead of a
        ``state_dict()`` walk (0.74 ms of host time per forward in round 2).  Tensors are looked up afresh every time:
        ``.to()`` replaces buffers, ``load_state_dict`` / optimiser steps change them in place."""
        slots = self.__dict__.get("_state_slots")
        if slots is None:
            slots = []
            for prefix, mod in self.named_modules():
                for name in mod._parameters:
                    if mod._parameters[name] is not None:
                        slots.append((mod, 0, name, (prefix + "." if prefix else "") + name))
                for name in mod._buffers:
                    if mod._buffers[name] is not None and name not in mod._non_persistent_buffers_set and name != "num_batches_tracked":
                        slots.append((mod, 1, name, (prefix + "." if prefix else "") + name))
            order = {k: i for i, k in enumerate(self.state_dict(keep_vars=True).keys())}
            slots.sort(key=lambda s_: order[s_[3]])
            self.__dict__["_state_slots"] = slots
        return [(key, (mod._buffers if kind else mod._parameters)[name]) for mod, kind, name, key in slots]

    def _hip_state(self, device):
        st = self._hip_states.get(device.index)
        if st is None:
            st = _DeviceState(device)
            for k, v in self._engine_options.items():
                _lib.check(st.lib.hn_set_option(st.handle, k.encode(), int(v)), "hn_set_option(%s)" % k)
            self._hip_states[device.index] = st
        tensors = list(self._state_tensors())
        ptr_sig = tuple(t.data_ptr() for _, t in tensors)
        sig = tuple(t._version for _, t in tensors) + (self._train_steps,) + ptr_sig
        if sig != st.signature:
            if ptr_sig != st.ptr_signature:          # storages moved (first call, .to(), load_state_dict into new tensors): re-bind
                keep = []
                for k, t in tensors:
                    if t.device != device:
                        raise RuntimeError("parameter %s lives on %s but the input is on %s" % (k, t.device, device))
                    d = t.detach()
                    if d.dtype != torch.float32 or not d.is_contiguous():
                        d = d.float().contiguous()
                    keep.append(d)
                    _lib.check(st.lib.hn_bind_tensor(st.handle, k.encode(), _lib.ptr(d), d.numel()), "hn_bind_tensor(%s)" % k)
                st.keepalive = keep
                # (a converted copy is a snapshot: remember the ORIGINAL pointers only when every tensor was bound in place)
                st.ptr_signature = ptr_sig if all(a.data_ptr() == b.data_ptr() for a, (_, b) in zip(keep, tensors)) else None
            # values changed (optimiser step, in-place edits): the 379 bindings stand, only the packed copies are rebuilt
            _lib.check(st.lib.hn_pack_weights(st.handle, _lib.ptr(st.packed), st.packed.numel(),
                                              _lib.stream_ptr(device)), "hn_pack_weights")
            st.signature = sig
        return st

    def set_engine_option(self, name, value):
        """hn_set_option on every device state of this module (and on states created later), e.g.
        ``set_engine_option("branch_stream", 0)`` keeps the whole bf16 forward on the caller's stream."""
        self._engine_options[name] = int(value)
        for st in self._hip_states.values():
            _lib.check(st.lib.hn_set_option(st.handle, name.encode(), int(value)), "hn_set_option(%s)" % name)

    @staticmethod
    def _push_bn_flags(st, named_bns, flags):
        if flags != st.bn_flags:
            for (k, _), f in zip(named_bns, flags):
                _lib.check(st.lib.hn_set_bn_eval(st.handle, k.encode(), int(f)), "hn_set_bn_eval(%s)" % k)
            st.bn_flags = flags

    def _pack_bf16(self, st, device):
        if st.packed_h is None:
            st.packed_h = torch.empty(st.lib.hn_packed_bf16_bytes(), dtype=torch.uint8, device=device)
        if st.signature_h != st.signature:
            _lib.check(st.lib.hn_pack_weights_bf16(st.handle, _lib.ptr(st.packed_h), st.packed_h.numel(),
                                                   _lib.stream_ptr(device)), "hn_pack_weights_bf16")
            st.signature_h = st.signature

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_hip_states"] = {}                # (engine options are kept: they are plain ints)
        d.pop("_state_slots", None)
        return d

    def forward(self, x):
        if x.shape[2] != 512 or x.shape[3] != 1024:
            raise NotImplementedError()
        if not x.is_cuda:
            raise RuntimeError("horizonnet_amd.HorizonNet runs on the MI355X HIP engine only: move the input "
                               "(and the module) to a cuda/ROCm device; there is no CPU fallback")
        B, C_in = int(x.shape[0]), int(x.shape[1])
        if self.training:
            if C_in < 3:
                raise RuntimeError("expected at least 3 input channels")
            xin = x.detach()
            if xin.dtype != torch.float32 or not xin.is_contiguous():
                xin = xin.float().contiguous()
            params = list(self.parameters())
            if len(params) != len(self._param_names):
                # nn.DataParallel's per-forward replicas carry plain tensors instead of Parameters: autograd could not
                # route gradients back, and the reference's GPU0-rooted replicate/scatter/gather is what this engine
                # replaces.  Fail loudly instead of training nothing.
                raise RuntimeError(
                    "horizonnet_amd.HorizonNet: train-mode forward on a module without its Parameters (an nn.DataParallel "
                    "replica?).  Multi-GPU training runs one process per GPU over torch.distributed/RCCL "
                    "(horizonnet_amd.parallel); nn.DataParallel(net, device_ids=[one device]) works because it calls the "
                    "module itself.")
            with torch.cuda.device(x.device):
                return _follow_autocast(*_HipTrainStep.apply(self, xin, *params))
        if C_in < 3:
            raise RuntimeError("expected at least 3 input channels")
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        with torch.cuda.device(x.device):
            st = self._hip_state(x.device)
            st.raise_if_failed()
            bon = torch.empty((B, 2, 1024), dtype=torch.float32, device=x.device)
            cor = torch.empty((B, 1, 1024), dtype=torch.float32, device=x.device)
            if self.precision == "bf16":
                self._pack_bf16(st, x.device)
                ws = st.workspace_bf16(B)
                _lib.check(st.lib.hn_forward_bf16(st.handle, _lib.ptr(xin), B, C_in, _lib.ptr(bon), _lib.ptr(cor), _lib.ptr(ws),
                                                  ws.numel(), _lib.stream_ptr(x.device)), "hn_forward_bf16")
            elif self.precision == "f32":
                ws = st.workspace(B)
                _lib.check(st.lib.hn_forward(st.handle, _lib.ptr(xin), B, C_in, _lib.ptr(bon), _lib.ptr(cor), _lib.ptr(ws),
                                             ws.numel(), _lib.stream_ptr(x.device)), "hn_forward")
            else:
                raise ValueError("precision must be 'f32' or 'bf16'")
            if self.check_status_async:
                st.post_status_read(ws)
        return _follow_autocast(bon, cor)

    def forward_async(self, x):
        """Two-stage form of ``forward`` for back-to-back batches (serving loops, ``inference_stream``): enqueues the
        convolutional trunk on the current stream and the recurrent head (bi-LSTM + Linear, reference model.py:263-269) on an
        engine-owned stream behind it, and returns a ``PendingForward`` at once.  The head's recurrence kernel needs 32 (bf16)
        / 64 (f32) of the 256 compute units for 32 panoramas, so the trunk of the NEXT ``forward_async`` call runs beside it:
        call ``forward_async(x[i+1])`` BEFORE ``pending[i].result()``.  bf16: bit-identical to ``forward``; f32: exact float32
        with a different summation order in the recurrence (agrees with ``forward`` to ~1e-6).  Train mode runs ``forward``
        and returns a completed handle."""
        if self.training:
            bon, cor = self.forward(x)
            return PendingForward(None, None, bon, cor, None, 0, False)
        if self.precision not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' or 'bf16'")
        bf16 = self.precision == "bf16"
        if x.shape[2] != 512 or x.shape[3] != 1024:
            raise NotImplementedError()
        if not x.is_cuda:
            raise RuntimeError("horizonnet_amd.HorizonNet runs on the MI355X HIP engine only: move the input "
                               "(and the module) to a cuda/ROCm device; there is no CPU fallback")
        B, C_in = int(x.shape[0]), int(x.shape[1])
        if C_in < 3:
            raise RuntimeError("expected at least 3 input channels")
        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        with torch.cuda.device(x.device):
            st = self._hip_state(x.device)
            st.raise_if_failed()
            bon = torch.empty((B, 2, 1024), dtype=torch.float32, device=x.device)
            cor = torch.empty((B, 1, 1024), dtype=torch.float32, device=x.device)
            if bf16:
                self._pack_bf16(st, x.device)
            ws = st.workspace_pipelined(B, bf16)
            slot = st.pipe_slot = 1 - getattr(st, "pipe_slot", 1)
            submit = st.lib.hn_forward_bf16_submit if bf16 else st.lib.hn_forward_submit
            _lib.check(submit(st.handle, _lib.ptr(xin), B, C_in, _lib.ptr(bon), _lib.ptr(cor), _lib.ptr(ws), ws.numel(), slot,
                              _lib.stream_ptr(x.device)), "hn_forward_submit")
            # bon / cor are written on the engine's head stream: the handle keeps them (and the input the trunk reads on this
            # stream) alive until result() has ordered them on the caller's stream
            pend = PendingForward(st, slot, bon, cor, ws, st.pipelined_status_bytes(B, bf16)[slot], self.check_status_async, bf16)
            pend._keep = xin
        return pend

    _TAP_SHAPES = {"stem": (256, 512, 64), "pool": (128, 256, 64), "c1": (128, 256, 256), "c2": (64, 128, 512),
                   "c3": (32, 64, 1024), "c4": (16, 32, 2048)}

    def forward_with_taps(self, x, names=("stem", "pool", "c1", "c2", "c3", "c4", "feature", "lstm")):
        """Eval-mode forward that also returns the named intermediates (hn_set_forward_tap) in the REFERENCE's layouts:
        stem / pool / c1..c4 as NCHW views [B,C,H,W], "feature" [B,1024,256] (model.py:259), "lstm" [256,B,1024]
        (model.py:264).  float32 in f32 mode; bf16 tensors (lstm: float32) in bf16 mode.  For parity tests."""
        assert not self.training and x.is_cuda
        B = int(x.shape[0])
        st = self._hip_state(x.device)
        dt = torch.float32 if self.precision == "f32" else torch.bfloat16
        bufs = {}
        for n in names:
            if n in self._TAP_SHAPES:
                bufs[n] = torch.empty((B,) + self._TAP_SHAPES[n], dtype=dt, device=x.device)
            elif n == "feature":
                bufs[n] = torch.empty((256, B, 1024), dtype=dt, device=x.device)
            elif n == "lstm":
                bufs[n] = torch.empty((256, B, 1024), dtype=torch.float32, device=x.device)
            else:
                raise KeyError(n)
            _lib.check(st.lib.hn_set_forward_tap(st.handle, n.encode(), _lib.ptr(bufs[n])), "hn_set_forward_tap")
        try:
            bon, cor = self.forward(x)
            torch.cuda.synchronize(x.device)
        finally:
            st.lib.hn_set_forward_tap(st.handle, None, None)
        taps = {}
        for n, t in bufs.items():
            if n in self._TAP_SHAPES:
                taps[n] = t.permute(0, 3, 1, 2)
            elif n == "feature":
                taps[n] = t.permute(1, 2, 0)
            else:
                taps[n] = t
        return bon, cor, taps

    def profile_forward(self, x):
        """One forward with per-launch-group HIP-event timing (hn_set_profiling).  Returns
        (bon, cor, [(name, ms, algorithmic_flops), ...]); synchronises.  Not for timed regions."""
        st = self._hip_state(x.device)
        _lib.check(st.lib.hn_set_profiling(st.handle, 1), "hn_set_profiling")
        try:
            bon, cor = self.forward(x)
            torch.cuda.synchronize(x.device)
            entries = []
            name = ctypes.create_string_buffer(256)
            ms = ctypes.c_float()
            fl = ctypes.c_double()
            for i in range(st.lib.hn_profile_count(st.handle)):
                _lib.check(st.lib.hn_profile_entry(st.handle, i, name, 256, ctypes.byref(ms), ctypes.byref(fl)),
                           "hn_profile_entry")
                entries.append((name.value.decode(), float(ms.value), float(fl.value)))
        finally:
            st.lib.hn_set_profiling(st.handle, 0)
        return bon, cor, entries

    def hip_status(self, device=None):
        """Blocking read of the engine's device-side status word (0 = ok)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        st = self._hip_states.get(device.index)
        if st is None or not st.workspaces:
            return 0
        worst = 0
        for kind, (B, ws) in st.workspaces.items():  # the status word is sticky per workspace
            val = ctypes.c_int(0)
            _lib.check(st.lib.hn_check_status(st.handle, _lib.ptr(ws), ctypes.byref(val)), "hn_check_status")
            worst = max(worst, val.value)
            if kind in ("bf16p", "f32p"):            # second head slot of the pipelined workspace
                off = st.pipelined_status_bytes(B, kind == "bf16p")[1]
                worst = max(worst, int(ws[off:off + 4].view(torch.int32).item()))
        return worst


_guard_data_parallel()
