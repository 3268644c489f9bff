"""``FusedAdam``: Adam over all 241 parameter tensors of ``HorizonNet`` in ONE kernel launch (``hn_adam_step``).

Replaces ``torch.optim.Adam(net.parameters(), …)`` at reference ``train.py:216-225`` / ``optimizer.step()`` at
``:279``: same arithmetic (no amsgrad; L2 ``weight_decay`` folded into the gradient; bias-corrected step), same
``param_groups[0]['lr']`` knob for ``adjust_learning_rate`` (``misc/utils.py:35-46``), ``state_dict`` /
``load_state_dict`` for checkpoints.  It works on the engine's FLAT gradient buffer: after ``loss.backward()`` every
``p.grad`` is a view into one 326 MB tensor laid out by ``hn_grad_offset``; the kernel reads that buffer, keeps the two
moments as flat buffers of the same layout, and writes the parameters in place -- one launch instead of one
multi-tensor pass per dtype/shape bucket, no per-step Python loop over 241 tensors.  Frozen parameters
(``requires_grad = False``, ``--freeze_earlier_blocks``) are skipped.
"""
import ctypes

import torch

from . import _lib


class FusedAdam:
    def __init__(self, net, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.net = net.module if hasattr(net, "module") else net
        self.lib = _lib.load()
        self.named = list(self.net.named_parameters())
        dev = self.named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam works on the MI355X engine's device tensors: move the module to the GPU first")
        self.device = dev
        self.param_groups = [{"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": float(weight_decay),
                              "params": [p for _, p in self.named]}]
        self.total = int(self.lib.hn_grad_floats())
        offs, order = [], []
        for k, p in self.named:
            o = int(self.lib.hn_grad_offset(k.encode()))
            if o < 0 or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedAdam: parameter %s is not a contiguous float32 tensor of the engine's layout" % k)
            offs.append(o)
            order.append((o, k, p))
        order.sort(key=lambda r: r[0])
        self.order = order
        self.offsets = torch.tensor([r[0] for r in order], dtype=torch.int64, device=dev)
        self.ends = torch.tensor([r[0] + r[2].numel() for r in order], dtype=torch.int64, device=dev)
        for (o, k, p), nxt in zip(order, [r[0] for r in order[1:]] + [self.total]):
            assert o + p.numel() <= nxt, "flat layout overlaps at %s" % k           # (alignment gaps between tensors are fine)
        self.m = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.step_count = 0
        # While THIS optimiser is alive a data-parallel backward leaves the gradient SUM in p.grad and step() applies 1 / world in its own
        # launch.  The module only holds a weak reference: once the optimiser is gone (another optimiser, gradient clipping or logging
        # takes over) backward goes back to storing the mean.  Code that reads p.grad between backward() and step() calls
        # mean_gradients_() first.
        import weakref
        self.net.defer_grad_mean = True
        self.net._fused_adam_ref = weakref.ref(self)
        self._ptr_sig = None
        self._ptrs = None
        self._active = None

    @torch.no_grad()
    def mean_gradients_(self):
        """Turn the rank SUM a data-parallel backward left in p.grad into the mean now (for gradient clipping / logging in front of
        step()); step() then applies no further 1 / world factor."""
        scale = float(getattr(self.net, "_grad_mean_scale", 1.0))
        if scale != 1.0:
            torch._foreach_mul_([p.grad for _, p in self.named if p.grad is not None], scale)
            self.net._grad_mean_scale = 1.0

    def zero_grad(self, set_to_none=True):
        for _, p in self.named:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _tables(self):
        sig = tuple((p.data_ptr(), p.requires_grad) for _, _, p in self.order)
        if sig != self._ptr_sig:
            self._ptrs = torch.tensor([p.data_ptr() for _, _, p in self.order], dtype=torch.int64, device=self.device)
            self._active = torch.tensor([1 if p.requires_grad else 0 for _, _, p in self.order], dtype=torch.uint8, device=self.device)
            self._ptr_sig = sig
        return self._ptrs, self._active

    def _flat_grads(self):
        """Device address of a flat gradient buffer in the engine's layout.  After loss.backward() on the engine every
        p.grad is normally a VIEW into the one buffer hn_train_backward filled (autograd adopts the views it is handed);
        if some gradient is not (accumulated over several backward passes, replaced by the caller) the gradients are
        gathered into an internal flat buffer with one multi-tensor copy."""
        base, views_ok = None, True
        live = [(o, k, p) for o, k, p in self.order if p.requires_grad]
        for o, k, p in live:
            if p.grad is None:
                raise RuntimeError("FusedAdam.step(): parameter %s has no gradient (call loss.backward() first)" % k)
            b = p.grad.data_ptr() - 4 * o
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or (base is not None and b != base):
                views_ok = False
                break
            base = b
        if not live:
            return None
        if views_ok:
            self.used_views = True
            return base
        self.used_views = False
        if getattr(self, "_gather", None) is None:
            self._gather = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        torch._foreach_copy_([self._gather[o:o + p.numel()].view(p.shape) for o, k, p in live], [p.grad.float() for o, k, p in live])
        return self._gather.data_ptr()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        base = self._flat_grads()
        if base is None:
            return
        ptrs, active = self._tables()
        st = getattr(self.net, "_hip_states", {}).get(self.device.index)
        if st is not None:
            st.raise_if_failed()     # an EARLIER step's recurrence timed out (its status word has arrived): stop before more updates
        self.step_count += 1
        with torch.cuda.device(self.device):
            _lib.check(self.lib.hn_adam_step(_lib.ptr(ptrs), _lib.ptr(self.offsets), _lib.ptr(self.ends), _lib.ptr(active), len(self.order),
                                             ctypes.c_void_p(base), _lib.ptr(self.m), _lib.ptr(self.v), self.total, g["lr"], g["betas"][0],
                                             g["betas"][1], g["eps"], g["weight_decay"], self.step_count,
                                             float(grad_scale) * float(getattr(self.net, "_grad_mean_scale", 1.0)),
                                             _lib.stream_ptr(self.device)), "hn_adam_step")
        self.net._grad_mean_scale = 1.0      # consumed: the next backward sets it again
        self.net._train_steps += 1           # parameters changed behind torch's version counters: the engine re-packs

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        """Own layout {'step', 'm', 'v', 'param_groups'}, or ``torch.optim.Adam``'s {'state', 'param_groups'} -- what the
        reference's ``checkpoint.pth.tar`` holds (see ``torch_adam_state_to_flat``)."""
        if "m" in sd and "v" in sd and "step" in sd:
            self.step_count = int(sd["step"])
            self.m.copy_(sd["m"].to(self.device))
            self.v.copy_(sd["v"].to(self.device))
            for g, s in zip(self.param_groups, sd["param_groups"]):
                g.update({k: v for k, v in s.items() if k != "params"})
            return
        entries = [(int(self.lib.hn_grad_offset(k.encode())), k, tuple(p.shape), bool(p.requires_grad)) for k, p in self.named]
        m, v, step, hyper = torch_adam_state_to_flat(sd, entries, self.total)
        self.m.copy_(m.to(self.device))
        self.v.copy_(v.to(self.device))
        self.step_count = step
        self.param_groups[0].update(hyper)


def torch_adam_state_to_flat(sd, entries, total):
    """``torch.optim.Adam.state_dict()`` -> (m, v, step, hyper-parameters) in the engine's flat layout.

    The reference builds its optimiser over the parameters with ``requires_grad`` in ``net.parameters()`` order
    (train.py:216-225) and stores ``optimizer.state_dict()`` in ``checkpoint.pth.tar`` (train.py:336-346): state index i is
    the i-th such parameter.  `entries`: (flat offset, name, shape, requires_grad) per parameter in ``named_parameters()``
    order.  ``exp_avg`` / ``exp_avg_sq`` are scattered to the offsets; parameters without state (never stepped) keep zero
    moments; ``step`` is the largest per-parameter step.  A state that is not Adam's (SGD momentum buffers) cannot seed the
    moments: they restart at zero with a warning, the hyper-parameters are still taken."""
    if "state" not in sd or "param_groups" not in sd:
        raise KeyError("FusedAdam.load_state_dict: neither FusedAdam's {'step','m','v','param_groups'} nor torch.optim's "
                       "{'state','param_groups'} layout (keys: %s)" % sorted(sd.keys()))
    ids = [i for g in sd["param_groups"] for i in g["params"]]
    live = [e for e in entries if e[3]]
    if len(ids) == len(live):
        targets = live
    elif len(ids) == len(entries):
        targets = entries
    else:
        raise ValueError("FusedAdam.load_state_dict: the torch optimiser state covers %d parameters, this module has %d "
                         "trainable of %d" % (len(ids), len(live), len(entries)))
    m = torch.zeros(total, dtype=torch.float32)
    v = torch.zeros(total, dtype=torch.float32)
    steps, adam_like = [], True
    for i, (o, k, shape, _) in zip(ids, targets):
        st = sd["state"].get(i)
        if st is None:
            continue
        if "exp_avg" not in st or "exp_avg_sq" not in st:
            adam_like = False
            break
        if tuple(st["exp_avg"].shape) != tuple(shape):
            raise ValueError("FusedAdam.load_state_dict: state %d has shape %s, parameter %s has %s"
                             % (i, tuple(st["exp_avg"].shape), k, tuple(shape)))
        n = st["exp_avg"].numel()
        m[o:o + n] = st["exp_avg"].reshape(-1).to("cpu", torch.float32)
        v[o:o + n] = st["exp_avg_sq"].reshape(-1).to("cpu", torch.float32)
        steps.append(int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"]))
    if not adam_like:
        import warnings
        warnings.warn("FusedAdam.load_state_dict: the checkpoint's optimiser state is not Adam's (no exp_avg / exp_avg_sq): "
                      "moments and step counter restart at zero", RuntimeWarning, stacklevel=3)
        m.zero_()
        v.zero_()
        steps = []
    hyper = {}
    g0 = sd["param_groups"][0]
    for key in ("lr", "betas", "eps", "weight_decay"):
        if key in g0:
            hyper[key] = tuple(g0[key]) if key == "betas" else float(g0[key])
    return m, v, (max(steps) if steps else 0), hyper
