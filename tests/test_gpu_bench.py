"""bench.py's own code paths on the GPU box: the world > 1 branch (two ranks on ONE device over gloo -- the driver's
8-GPU run uses the same code with RCCL), the strong-scaling mode, and the fields the record must carry."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--steps", "4", "--warmup", "3", "--num-pts", "20000", "--resolution", "128", "--no-cpu-baseline", "--no-live-pmc"]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("strong", [False, True])
def test_bench_two_ranks_on_one_device(strong):
    port = 24500 + (os.getpid() % 2000) + int(strong)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo", "--no-dropin",
           "--sustained-steps", "30"] + SMALL
    if strong:
        cmd += ["--global-batch", "2"]
    r = _run(cmd)
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["higher_is_better"] is True
    assert r["scaling"] == ("strong" if strong else "weak")
    assert r["config"]["renders_per_step"] == 16  # 2 x 8 per GPU (weak) or the fixed 4 x 2 x 2 step (strong)
    assert r["skipped_steps"] == {"timed_region": 0, "whole_run": 0}
    assert r["allreduce_exposed_ms_per_step"] is not None and r["allreduce_exposed_ms_per_step"] >= 0.0
    assert r["value"] > 0 and abs(r["value"] - 16 * 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]
    assert r["roofline"]["bound"] in ("hbm", "mfma", "valu") and r["roofline"]["achieved"] > 0
    assert r["replicas_bit_identical"] is True and len(r["rank_phases_ms"]) == 2


@pytest.mark.timeout(1800)
def test_bench_eight_ranks_strong_scaling_shape_on_one_device():
    """The driver's 8-GPU command shape (`--gpus 8`, one process per rank) with the reference's own batch (b = 2: a
    FIXED step of 4 x 2 x 2 = 16 renders, main_train_dimo.py:266-293, two per rank) -- eight ranks over gloo sharing the
    one device this box has.  What it can show without an 8-GPU node: the ranks shard, reduce and update as one
    (replicas bit-identical after every step of the run: the parameter buckets' checksums agree), no step is skipped,
    and every rank reports its step by phase.  (Nothing about speed: eight processes time-share one GPU here.)"""
    port = 25600 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "8", "--backend", "gloo", "--no-dropin",
           "--sustained-steps", "20", "--global-batch", "2"] + SMALL
    r = _run(cmd, timeout=1600)
    assert r["n_gpus"] == 8 and r["ranks_seen"] == 8 and r["scaling"] == "strong"
    assert r["config"]["renders_per_step"] == 16
    assert r["replicas_bit_identical"] is True
    assert r["skipped_steps"] == {"timed_region": 0, "whole_run": 0}
    ph = r["rank_phases_ms"]
    assert len(ph) == 8 and all(p["renders_per_step_this_rank"] == 2 for p in ph)
    assert all(p["head_ms"] > 0 and p["chains_ms"] > 0 and p["tail_ms"] > 0 for p in ph)


@pytest.mark.timeout(1200)
def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (the command shape the driver uses for N = 1): the script spawns
    its own ranks, and rank 0's line says how many ranks took part."""
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--no-dropin", "--sustained-steps", "30"]
             + SMALL)
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["backend"] == "gloo"
    assert r["config"]["renders_per_step"] == 16 and r["config"]["parallelism"] == "dp2"
    assert r["sustained"]["steps"] == 30 and r["sustained"]["frames_per_s"] > 0
    assert r["skipped_steps"]["whole_run"] == 0


@pytest.mark.timeout(1200)
def test_bench_single_rank_record_fields():
    r = _run([sys.executable, "bench.py"] + SMALL)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "skipped_steps", "dropin_frames_per_s",
                "kernel_rooflines"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["dtype"] == "f32" and "workload" in r["config"]
    assert r["dropin_frames_per_s"] and r["dropin_frames_per_s"] < r["value"]
    assert r["skipped_steps"]["whole_run"] == 0
    assert r["kernel_rooflines"]["blend_bwd"]["ms"] > 0
    assert r["ranks_seen"] == 1
    # every kernel group is priced with ITS renders per launch: in the timed schedule every stage runs per motion batch
    # (4 renders); the roofline clock is taken in an extra pass with ONE backward launch over the step's 8
    assert r["kernel_rooflines"]["blend_fwd"]["renders_per_launch"] == 4
    assert r["kernel_rooflines"]["blend_bwd"]["renders_per_launch"] == 4
    assert r["roofline"]["renders_per_launch"] == 8 and r["roofline"]["timed_region"]["renders_per_launch"] == 4
    assert r["roofline"]["avg_ms"] > 0 and r["roofline"]["timed_region"]["avg_ms"] > 0
    assert "literal_loop_with_logging_reads" in r["dropin_detail"]
    assert r["dropin_detail"]["launch_chains_per_step"] == 1.0
    s = r["sustained"]  # default: 1200 consecutive steps, across the stage-s2 prune of step 2000
    assert s["steps"] >= 1000 and s["s2_prunes_crossed"] >= 1 and s["frames_per_s"] > 0
    assert s["gaussians_start_end"][0] == 20000 and s["gaussians_start_end"][1] <= 20000


@pytest.mark.timeout(600)
def test_rccl_calls_of_the_data_parallel_step_on_a_one_rank_communicator():
    """No multi-GPU box has ever been available: the RCCL calls of the sharded step (async all-reduce of the
    per-Gaussian head of the gradient bucket under the TimeNet backward, the tail + overflow flag after it, the s1
    statistics) are at least EXECUTED here -- one process, backend nccl (= RCCL), a communicator of one rank, the
    trainer told it is rank 0 of 2 (so it renders half of the step's triples and reduces with itself)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
tr, pol = bench.make_trainer(torch.device("cuda", 0), 0, 1, 20000, 128)
tr.world = 2          # shard as rank 0 of 2; every collective runs on the one-rank RCCL communicator
tr.time_allreduce = True
p0 = tr.renderer.gaussians.flat_params.clone()
n = sum(tr.train_step() for _ in range(3))
torch.cuda.synchronize()
assert n == 3 * 4, n   # half of the 8 triples per step
assert len(tr.allreduce_events) == 3 and all(a.elapsed_time(b) >= 0.0 for a, b in tr.allreduce_events)
assert torch.isfinite(tr.last_loss) and torch.isfinite(tr.renderer.gaussians.flat_params).all()
assert not torch.equal(p0, tr.renderer.gaussians.flat_params) and tr.skipped_steps == 0
t = torch.ones(4, device="cuda"); dist.all_reduce(t); assert float(t.sum()) == 4.0
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", dist.is_nccl_available())
'''
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ), capture_output=True, text=True,
                       timeout=500)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK True" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
