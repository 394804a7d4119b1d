"""bench.py's N > 1 control flow on a 1-GPU box: two ranks share cuda:0 over gloo (MSFL_BENCH_SHARED_GPU=1).
Checks what the driver relies on: exactly one JSON line, from rank 0, whole-job value, no CPU baseline leg."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("launched", [True, False], ids=["torch.distributed.run", "plain python (self-launch)"])
def test_two_ranks_print_one_whole_job_line(launched):
    env = dict(os.environ, MSFL_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29541"] if launched else [sys.executable]
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scans", "64", "--cpu-sample", "0"]
    env.pop("RANK", None)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["config"]["scans_per_gpu"] == 64
    # whole-job value: both ranks' registrations over the max-over-ranks time
    assert abs(d["value"] - 2 * 64 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["cpu_baseline"] is None
    assert d["rccl_ranks"] == 2 and len(d["ms_per_step_per_rank"]) == 2
    assert abs(max(d["ms_per_step_per_rank"]) - d["ms_per_step"]) < 1e-9
    assert d["n_failed"] == 0


@pytest.mark.gpu
def test_strong_scaling_shards_one_job_and_broadcasts_the_map():
    """`--scaling strong --total-scans N` (the shape of BASELINE configs[3]): rank r registers shard_bounds(N, r, G) of ONE
    job whose scans depend on their global number only, rank 0 builds the map and broadcasts it over the process group.
    Two shared-GPU ranks must produce, scan for scan, the poses one rank produces (sha1 over the gathered poses), report
    the whole job's rate, and an uneven split (N odd) must work."""
    def run(nproc, port):
        env = dict(os.environ, MSFL_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
        env.pop("RANK", None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
               "--scaling", "strong", "--total-scans", "65", "--cpu-sample", "0", "--no-h2d"]
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])
    one, two = run(1, 29551), run(2, 29552)
    for d, g in ((one, 1), (two, 2)):
        assert d["scaling"] == "strong" and d["n_gpus"] == g and d["config"]["total_scans"] == 65 and d["n_failed"] == 0
        assert abs(d["value"] - 65 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert two["config"]["scans_per_gpu"] == 33 and "broadcast" in two["config"]["map_source"]
    assert one["poses_sha1"] == two["poses_sha1"], "the job's poses must not depend on the number of ranks"


@pytest.mark.gpu
def test_one_rank_over_rccl_executes_every_collective_of_the_multi_gpu_path():
    """The first 8-GPU run must not be the first RCCL run.  One rank under torch.distributed.run WITHOUT the shared-GPU hook:
    init_process_group("nccl", device_id=...) (nccl IS RCCL on ROCm), the map broadcast and the ragged pose gather forced on
    (MSFL_BENCH_FORCE_COLLECTIVES=1), the per-step all_gather_into_tensor on device tensors and the timing all_gather -- and
    the job's poses equal, bit for bit, those of the plain single-process run."""
    def run(launched, port=29561):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "MSFL_BENCH_SHARED_GPU", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        if launched:
            env["MSFL_BENCH_FORCE_COLLECTIVES"] = "1"
        launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                    "--master-port", str(port)] if launched else [sys.executable]
        cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scaling", "strong",
                          "--total-scans", "64", "--cpu-sample", "0", "--no-h2d", "--no-stages"]
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])
    rccl, plain = run(True), run(False)
    assert rccl["rccl_ranks"] == 1 and rccl["n_gpus"] == 1 and rccl["n_failed"] == 0
    assert rccl["config"]["process_group_backend"] == "nccl" and "broadcast" in rccl["config"]["map_source"]
    assert plain["config"]["process_group_backend"] is None
    assert rccl["poses_sha1"] == plain["poses_sha1"]
