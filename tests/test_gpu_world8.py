"""World size 8 without an 8-GPU node: the driver's N = 8 command lines of bench.py, eight processes time-slicing the
box's ONE GPU over gloo.  A file of its own, named to be collected LAST: eight extra processes on the device are a rig, not a deployment
(one rank per GPU), and whatever they leave behind -- one launch in ~10 lost a rank to SIGABRT in round 5 -- must not sit in front of the
parity tests of the other files."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hiputil import bench_ranks  # noqa: E402


def test_bench_eight_rank_self_launch_forward():
    """`bench.py --gpus 8` re-exec, rendezvous of eight ranks on 127.0.0.1, pin_rank_affinity at (cores / 8) per rank, eight engines and
    workspaces alive together, shard_for_rank over 8, barrier + max-over-ranks timing, one JSON line, clean teardown."""
    rec = bench_ranks(["--batch", "2", "--steps", "2", "--warmup", "1"], ranks=8)
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 16 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["rccl"]["world_size_seen_by_rccl"] == 8 and len(rec["rccl"]["devices"]) == 8
    assert rec["config"]["parallelism"].startswith("dp8")


def test_bench_eight_rank_self_launch_train():
    """configs[3] at world size 8 (B = 1 per rank, bf16 step, gradients on the wire as bf16, deferred mean): per-rank rooms, broadcast of
    rank 0's weights, the five-segment backward with its seven all-reduce buckets (traced table), the status word MAX-reduced, teardown."""
    rec = bench_ranks(["--mode", "train", "--dtype", "bf16", "--allreduce-dtype", "bf16", "--batch", "1", "--steps", "2", "--warmup", "1", "--rooms", "4"],
                           ranks=8, timeout=1800)
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 8 and rec["value"] > 0 and np.isfinite(rec["final_loss"])
    assert rec["rccl"]["world_size_seen_by_rccl"] == 8
    tr = rec["allreduce_overlap"]["traced_step"]
    # (64 MB buckets: 7 when the gradients travel as float32 -- 326 MB --, 6 on the bf16 wire: the height-compression range is 91 MB = 2 buckets)
    assert len(tr["segment_kernels_done_ms"]) == 5 and len(tr["buckets"]) == 6, [b["MB"] for b in tr["buckets"]]
    assert sum(b["MB"] for b in tr["buckets"]) == pytest.approx(163.1, abs=1.0)
    print("[parity] 8 ranks on one GPU, train: buckets (MB) %s, host cores per rank %s" % ([b["MB"] for b in tr["buckets"]], rec.get("host_cores_per_rank")))


def test_bench_eight_rank_self_launch_layout():
    """configs[4] over 8 ranks: this host's cores / 8 per rank for the layout fit -- the rate a first 8-GPU run of the layout leg should be
    compared with (the forward does not depend on host cores; the fit does)."""
    rec = bench_ranks(["--mode", "layout", "--panoramas", "128", "--batch", "8"], ranks=8, timeout=1800)
    assert rec["n_gpus"] == 8 and rec["render_crc_mismatches"] == 0 and rec["host_cores_per_rank"] >= 1
    par = rec["iou3d_parity_vs_reference_inference"]
    assert par["f32"]["iou3d_failed"] == 0 and par["f32"]["iou3d_mean"] > 0.9999 and par["f32"]["corner_count_mismatches"] == 0
    print("[parity] 8 ranks on one GPU, layout: host cores per rank %s, %s panoramas/s end to end" % (rec["host_cores_per_rank"], rec["value"]))
