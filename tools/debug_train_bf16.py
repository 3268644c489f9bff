"""Diagnostic: dy / dz of selected units, train_precision bf16 against f32, same weights and input."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from horizonnet_amd import HorizonNet, _lib
from oracle.weights import make_state_dict, conv_specs
DEV = "cuda:0"
B = 1
sd = make_state_dict(11, "random")
x = torch.rand(B, 3, 512, 1024, generator=torch.Generator().manual_seed(12)).to(DEV)
g = torch.Generator().manual_seed(13); wb = (torch.rand(B, 2, 1024, generator=g) * 2 - 1).to(DEV)
g = torch.Generator().manual_seed(14); wc = (torch.rand(B, 1, 1024, generator=g) * 2 - 1).to(DEV)
net = HorizonNet("resnet50", True); net.load_state_dict(sd); net = net.to(DEV).train()
net.bi_rnn.dropout = 0.0; net.drop_out.p = 0.0
L = _lib.load()
names = [c[0] for c in conv_specs()]
out8 = (ctypes.c_int64 * 8)()
L.hn_train_debug_unit(B, 0, out8); nunits = out8[7]
info = []
for u in range(nunits):
    L.hn_train_debug_unit(B, u, out8); info.append(tuple(out8))
want = sys.argv[1:] or ["ghc_lst.3.layer.3", "ghc_lst.3.layer.2", "ghc_lst.3.layer.0", "layer4.2.conv3", "layer4.2.conv2.1", "layer4.2.conv1",
                        "layer4.0.downsample", "layer4.0.conv1", "layer3.5.conv3", "layer2.0.conv2", "layer1.0.conv1", "encoder.conv1"]
for w in want:
    u = [i for i, t in enumerate(info) if w in names[t[0]]][0]
    M, C = info[u][1], info[u][2]
    res = {}
    for prec in ("f32", "bf16"):
        net.train_precision = prec
        dy = torch.zeros(M * C, device=DEV); dz = torch.zeros(M * C, device=DEV)
        bon, cor = net(x)
        st = net._hip_states[0]
        _lib.check(L.hn_train_debug_set(st.handle, u, _lib.ptr(dy), _lib.ptr(dz)), "set")
        for p in net.parameters(): p.grad = None
        ((bon * wb).sum() + (cor * wc).sum()).backward()
        torch.cuda.synchronize()
        ws = st.train_workspace(B).view(torch.float32)
        z = ws[info[u][4]:info[u][4] + M * C].clone()
        res[prec] = (dy.clone(), dz.clone(), z, bon.detach().clone())
    a, b = res["f32"], res["bf16"]
    rel = lambda p, q: float((p - q).norm() / (p.norm() + 1e-30))
    print("%-45s M=%7d C=%4d  z %.3e  dy %.3e  dz %.3e  (bon %.3e)" % (names[info[u][0]][-45:], M, C, rel(a[2], b[2]), rel(a[0], b[0]), rel(a[1], b[1]), rel(a[3], b[3])))
