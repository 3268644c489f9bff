"""Compare the engine's intermediate training gradients of selected units with torch autograd (CPU oracle)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from horizonnet_amd import HorizonNet, _lib
from oracle import horizonnet_ref
from oracle.weights import make_state_dict
from oracle.hostinfo import usable_cores
torch.set_num_threads(usable_cores())
DEV = "cuda:0"
B = 2
sd = make_state_dict(11, "random")
x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(12))
g = torch.Generator().manual_seed(13); wb = torch.rand(B, 2, 1024, generator=g) * 2 - 1
g = torch.Generator().manual_seed(14); wc = torch.rand(B, 1, 1024, generator=g) * 2 - 1

# oracle with hooks on the ghc unit outputs (post-ReLU) and conv outputs (pre-BN)
ref = {k: v.clone() for k, v in sd.items()}
for k, v in ref.items():
    if v.dtype == torch.float32 and "running_" not in k: v.requires_grad_(True)
taps = {}
orig_conv, orig_bn = horizonnet_ref._conv, horizonnet_ref._bn
last = {"k": None}
def conv_hook(xx, sdd, k, stride, ks):
    z = orig_conv(xx, sdd, k, stride, ks); z.retain_grad(); taps["z:" + k] = z; last["k"] = k; return z
orig_relu = F.relu
def relu_hook(t, *a, **kw):
    y = orig_relu(t); y.retain_grad(); taps["y:" + str(last["k"])] = y; return y
horizonnet_ref.F.relu = relu_hook
def bn_hook(xx, sdd, k):
    y = orig_bn(xx, sdd, k); return y
horizonnet_ref._conv = conv_hook
rb, rc = horizonnet_ref.forward_train(x, ref, 0.1)
((rb * wb).sum() + (rc * wc).sum()).backward()

net = HorizonNet("resnet50", True); net.load_state_dict(sd); net = net.to(DEV).train()
net.bi_rnn.dropout = 0.0; net.drop_out.p = 0.0
L = _lib.load()
names = [c[0] for c in __import__("oracle.weights", fromlist=["conv_specs"]).conv_specs()]
out8 = (ctypes.c_int64 * 8)()
L.hn_train_debug_unit(B, 0, out8); nunits = out8[7]
info = []
for u in range(nunits):
    L.hn_train_debug_unit(B, u, out8); info.append(tuple(out8))
want = sys.argv[1:] or ["ghc_lst.3.layer.3", "ghc_lst.3.layer.2", "ghc_lst.3.layer.1", "ghc_lst.2.layer.2", "ghc_lst.2.layer.1"]
for w in want:
    u = [i for i, t in enumerate(info) if w in names[t[0]]][0]
    ci, M, C = info[u][0], info[u][1], info[u][2]
    dy = torch.zeros(M * C, device=DEV); dz = torch.zeros(M * C, device=DEV)
    bon, cor = net(x.to(DEV))
    st = net._hip_states[0]
    _lib.check(L.hn_train_debug_set(st.handle, u, _lib.ptr(dy), _lib.ptr(dz)), "set")
    for p in net.parameters(): p.grad = None
    ((bon * wb.to(DEV)).sum() + (cor * wc.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    zt = taps["z:" + names[ci]]
    ref_dz = zt.grad.permute(0, 2, 3, 1).reshape(M, C)
    e_dz = (dz.cpu().view(M, C) - ref_dz).abs().max().item() / ref_dz.abs().max().item()
    yt = taps.get("y:" + names[ci])
    e_dy = float("nan")
    if yt is not None and yt.grad is not None:
        ref_dy = yt.grad.permute(0, 2, 3, 1).reshape(M, C)
        e_dy = (dy.cpu().view(M, C) - ref_dy).abs().max().item() / ref_dy.abs().max().item()
    # engine-side saved tensors of this unit vs the oracle's
    ws = st.train_workspace(B).view(torch.float32)
    zoff, yoff = info[u][4], info[u][5]
    ez = (ws[zoff:zoff + M * C].cpu().view(M, C) - zt.detach().permute(0, 2, 3, 1).reshape(M, C)).abs().max().item()
    ey = (ws[yoff:yoff + M * C].cpu().view(M, C) - yt.detach().permute(0, 2, 3, 1).reshape(M, C)).abs().max().item() if yt is not None else float("nan")
    # recompute the BN adjoint in torch from the engine's own inputs
    stoff = info[u][6]
    zz = ws[zoff:zoff + M * C].cpu().view(M, C).double(); yy = ws[yoff:yoff + M * C].cpu().view(M, C)
    mean = ws[stoff:stoff + C].cpu().double(); invstd = ws[stoff + C:stoff + 2 * C].cpu().double()
    gamma = sd[names[ci].replace("layers.0.1", "layers.1") + ".weight"].double() if "layers.0.1" in names[ci] else None
    if gamma is not None:
        gdy = dy.cpu().view(M, C).double() * (yy > 0)
        zh = (zz - mean) * invstd
        S1 = gdy.sum(0); S2 = (gdy * zh).sum(0)
        dz_t = gamma * invstd * (gdy - S1 / M - zh * S2 / M)
        print("      torch-BN-adjoint(engine inputs) vs oracle dz: %.3e ; vs engine dz: %.3e ; mean/invstd vs batch stats: %.2e %.2e" % (
            (dz_t - ref_dz.double()).abs().max().item() / ref_dz.abs().max().item(),
            (dz_t - dz.cpu().view(M, C).double()).abs().max().item() / ref_dz.abs().max().item(),
            (mean - zz.mean(0)).abs().max().item(), (invstd - 1 / torch.sqrt(zz.var(0, unbiased=False) + 1e-5)).abs().max().item()))
    print("unit %3d %-58s M=%6d C=%4d  dy rel-err %.3e  dz rel-err %.3e  saved z err %.2e y err %.2e" % (u, names[ci][-58:], M, C, e_dy, e_dz, ez, ey))
    L.hn_train_debug_set(st.handle, -1, None, None)
