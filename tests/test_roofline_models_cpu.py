"""bench.py's roofline models (the denominators of every `roofline.frac` it prints) pinned on the CPU: the algorithmic work per panorama of
SURVEY.md 8(d) / BASELINE.md, and the ordering / three-GEMMs-per-conv structure of the training bounds -- a changed model can no longer move a
reported fraction without this file noticing."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_work_per_panorama_is_the_surveyed_figure():
    # 142.9 GFLOP per 512x1024 panorama (convs + both LSTM layers), independent of the batch size
    for B in (1, 32):
        _, fl, _, _ = bench.f32_mixed_roofline(B)
        assert fl / B == pytest.approx(142.9e9, rel=2e-3)
    t16, fl16, by16, hbm16 = bench.bf16_mixed_roofline(32)
    t32, fl32, by32, hbm32 = bench.f32_mixed_roofline(32)
    assert fl16 == pytest.approx(fl32, rel=1e-12)                 # the same GEMMs ...
    assert by16 < 0.6 * by32                                      # ... on half-width operands (the f32 gate pre-activations stay 4 bytes)
    assert hbm16 > 0 and t16 < t32
    # 53 convs of the ResNet-50 trunk + 16 of the height compression = the 69 conv launches of DESIGN.md section 4
    assert len(bench.conv_table()) == 69


def test_training_bounds_are_ordered_and_count_three_gemms_per_conv():
    t, fl, by, t_hbm = bench.train_mixed_roofline(64)
    tm, flm, bym, t_hbm_m = bench.train_mixed_roofline(64, fused_minimum=True)
    assert flm == fl and bym < by and tm < t and t_hbm_m < t_hbm  # fusing BatchNorm passes removes bytes, never FLOP
    _, f_fwd, _, _ = bench.bf16_mixed_roofline(64)
    assert 2.9 < fl / f_fwd < 3.0                                 # forward + data gradient + weight gradient (the stem has no data gradient)
    assert 0.70 < tm / t < 0.75                                   # the fused-minimum design bound is ~27 % below the round-4 bound
