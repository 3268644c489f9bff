"""The algebra behind horizonnet_amd/csrc/bn_fold.hip ("BatchNorm-folded 1x1 units"), checked in float64 against torch
autograd on the CPU: for z = a W^T (a 1x1 / stride-1 conv, reference model.py:78-81 via torchvision's Bottleneck conv3 + bn3) followed by a
batch-statistics BatchNorm (+ residual, ReLU), the statistics and every gradient follow from three small matrices

    P = g^T a   (the weight-gradient GEMM on g = dy x ReLU mask),   G = a^T a,   A = colsum(a)

without ever forming z, z-hat or dz.  The GPU tests (test_bn_folded_adjoint_equals_classical_adjoint, test_bn_folded_forward_unit) pin the
kernels to the classical passes; this file pins the formulas themselves, so a disagreement on the GPU can be put on the kernels."""
import pytest
import torch

EPS = 1e-5


def _case(seed, M, K, N, residual):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g, dtype=torch.float64).relu()          # a unit's input is a ReLU output
    W = torch.randn(N, K, generator=g, dtype=torch.float64) / K ** 0.5
    gamma = torch.rand(N, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(N, generator=g, dtype=torch.float64) * 0.1
    res = torch.randn(M, N, generator=g, dtype=torch.float64) if residual else None
    dy = torch.randn(M, N, generator=g, dtype=torch.float64)
    return a, W, gamma, beta, res, dy


def _classical(a, W, gamma, beta, res, dy):
    a = a.clone().requires_grad_(True)
    W = W.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True)
    beta = beta.clone().requires_grad_(True)
    z = a @ W.t()
    mean = z.mean(0)
    var = z.var(0, unbiased=False)
    y = (z - mean) / torch.sqrt(var + EPS) * gamma + beta
    if res is not None:
        y = y + res
    out = y.relu()
    out.backward(dy)
    return out.detach(), mean.detach(), var.detach(), a.grad, W.grad, gamma.grad, beta.grad


@pytest.mark.parametrize("M,K,N,residual", [(4096, 64, 256, True), (1000, 128, 64, False), (37, 64, 64, True)])
def test_folded_forward_statistics_and_adjoint_equal_autograd(M, K, N, residual):
    a, W, gamma, beta, res, dy = _case(7 + M, M, K, N, residual)
    out, mean, var, da_ref, dW_ref, dgamma_ref, dbeta_ref = _classical(a, W, gamma, beta, res, dy)

    # ---- forward: the affine is known before the conv runs (bn_fold_stats_kernel) ----
    G = a.t() @ a
    A = a.sum(0)
    WG = W @ G
    mean_f = (W @ A) / M
    ez2 = (WG * W).sum(1) / M                                    # diag(W G W^T) / M
    var_f = ez2 - mean_f ** 2
    assert float((mean_f - mean).abs().max()) < 1e-12
    assert float((var_f - var).abs().max()) < 1e-11
    invstd = 1.0 / torch.sqrt(var_f + EPS)
    scale, shift = gamma * invstd, beta - gamma * invstd * mean_f
    y = (a @ W.t()) * scale + shift                              # (the fused conv epilogue: z lives in the accumulators only)
    if res is not None:
        y = y + res
    assert float((y.relu() - out).abs().max()) < 1e-11
    mask = (y > 0).to(torch.float64)

    # ---- adjoint (bn_fold_coef_kernel / bn_fold_finish_kernel; the two data-gradient convs) ----
    g = dy * mask
    P = g.t() @ a                                                # [N][K]
    S1 = g.sum(0)                                                # d beta
    S2 = invstd * ((W * P).sum(1) - mean_f * S1)                 # d gamma = sum_m g z-hat
    c1 = gamma * invstd
    dW = c1[:, None] * (P - S1[:, None] / M * A[None, :] - (S2 * invstd / M)[:, None] * (WG - mean_f[:, None] * A[None, :]))
    e = c1 * invstd * S2 / M
    Q = W.t() @ (e[:, None] * W)                                 # [K][K]
    r = ((c1 * S1 / M - e * mean_f)[:, None] * W).sum(0)         # [K]
    da = g @ (c1[:, None] * W) - a @ Q - r[None, :]

    def rel(x, ref):
        return float((x - ref).abs().max() / ref.abs().max())

    assert rel(S1, dbeta_ref) < 1e-12
    assert rel(S2, dgamma_ref) < 1e-10
    assert rel(dW, dW_ref) < 1e-10
    assert rel(da, da_ref) < 1e-10


def test_folded_transfers_per_unit_match_design_table():
    """12.75 activation-sized transfers per classical conv3 unit against ~7 folded, counted in units of the OUTPUT tensor
    (M x N x 2 bytes) for the backbone's K = N / 4."""
    q = 0.25                                                     # input / output size
    classical_fwd = (q + 1) + (1 + 1 + 1)                        # conv: read a, write z; affine: read z, read res, write y
    classical_bwd = (1 + 1) + (1 + 1 + 1) + (1 + q) + (1 + q) + 1          # reduce: dy, z; apply: dy, z -> dz; wgrad: dz, a; dgrad: dz -> da; + mask/res-grad traffic
    folded_fwd = q + (q + 1 + 1)                                 # Gram: read a; fused conv: read a, read res, write y (+ 1/16 for the mask)
    folded_bwd = (1 + q + 1) + (1 + q) + (q + q)                 # P-GEMM: dy, a, write g; conv A: g -> da; conv B: a -> da (accumulate)
    assert classical_fwd == pytest.approx(4.25)
    assert classical_fwd + classical_bwd == pytest.approx(12.75)
    assert 6.5 <= folded_fwd + folded_bwd <= 7.5
