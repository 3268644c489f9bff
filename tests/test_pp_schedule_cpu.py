"""The ping-pong conv kernels' phase schedules against the wave-level model (tools/pp_schedule_model.py): no LDS region is read
before all eight waves' LDS-DMA pieces of it have landed, none is re-filled while a wave may still read it, every wave executes the
same number of barriers -- under random wave interleavings and adversarial landing times; and the model DOES catch broken schedules."""
import pytest

from tools import pp_schedule_model as m


@pytest.mark.parametrize("nk,ntiles", [(2, 1), (2, 3), (3, 2), (4, 3), (9, 2)])
def test_pingpong_schedule_is_race_free(nk, ntiles):
    m.sweep(lambda w: m.pp_program(w, nk, ntiles), seeds=6)


@pytest.mark.parametrize("apw,bpw", [(2, 2), (4, 1)])
@pytest.mark.parametrize("nsteps,ntiles", [(1, 1), (1, 3), (3, 2), (6, 2)])
def test_dw_reuse_schedule_is_race_free(apw, bpw, nsteps, ntiles):
    m.sweep(lambda w: m.dwr_program(w, nsteps, ntiles, apw, bpw), seeds=6)


@pytest.mark.parametrize("bug", ["wait1", "wait4", "lgk3", "norealign"])
def test_model_detects_broken_pingpong_schedules(bug):
    with pytest.raises(AssertionError):
        for nk in (2, 3, 4):
            for nt in (1, 2, 3):
                m.sweep(lambda w: m.pp_program(w, nk, nt, bug), seeds=10)


@pytest.mark.parametrize("bug", ["wait1", "wait4", "lgk3"])
def test_model_detects_broken_dw_reuse_schedules(bug):
    with pytest.raises(AssertionError):
        for ns in (1, 2, 3):
            for nt in (1, 2, 3):
                m.sweep(lambda w: m.dwr_program(w, ns, nt, 4, 1, bug), seeds=10)


@pytest.mark.parametrize("nsteps,ntiles", [(1, 1), (1, 3), (2, 2), (3, 3), (6, 2)])
def test_dw_reuse_64_column_schedule_is_race_free(nsteps, ntiles):
    """conv3x3_dwr64_bf16.hip: two phases per chunk, weight ring of four, the epilogue slabs inside region A1 of the finished buffer."""
    m.sweep(lambda w: m.dwr64_program(w, nsteps, ntiles), seeds=6)


@pytest.mark.parametrize("bug", ["wait", "lgk", "slab", "norealign"])
def test_model_detects_broken_64_column_schedules(bug):
    with pytest.raises(AssertionError):
        for ns in (1, 2, 3):
            for nt in (2, 3):
                m.sweep(lambda w: m.dwr64_program(w, ns, nt, bug), seeds=10)
