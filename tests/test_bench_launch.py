"""CPU: bench.py's N > 1 launch path, end to end, with the gloo backend and the kernels stubbed out (`--dry-run-gloo`):
the self-launch (os.execve into torch.distributed.run), the driver's own launch line, the rank logic (barriers, max over
ranks, per-rank rates), the broadcast of the REAL packed weight blob, the pose all-gather, and stdout == exactly one JSON
line.  The GPU box has one GPU, so this is the only place the N > 1 branches of bench.py execute before the driver's
8-GPU node (VERDICT r2 missing #2)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    env["OMP_NUM_THREADS"] = "2"
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must be exactly one JSON line, got %d:\n%s" % (len(lines), stdout[:2000])
    return json.loads(lines[0])


def _check(out, n):
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["backend"] == "gloo"
    assert len(out["per_rank_pairs_per_s"]) == n and all(v > 0 for v in out["per_rank_pairs_per_s"])
    assert out["config"]["global_batch"] == 64 * n and out["scaling"] == "weak"
    assert "dry_run" in out and "NOT measurements" in out["dry_run"]
    chk = out["dry_run_checks"]
    assert chk["gather_ok"] and chk["gathered_rank_markers"] == [float(r + 1) for r in range(n)]
    assert chk["blob_identical_on_all_ranks"]
    assert out["pipelined"]["outputs_bit_identical_to_single_stream"] and len(out["pipelined"]["per_rank_pairs_per_s"]) == n
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in out
    return chk


@pytest.mark.timeout(600)
def test_self_launch_two_ranks_gloo_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-gloo", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    chk = _check(_one_json_line(r.stdout), 2)
    # the broadcast blob is the float32 panels only (~54 MB): the split-f16 panels of the f16x3 mode are derived on the device
    assert 50e6 < chk["blob_bytes"] < 58e6, chk


@pytest.mark.timeout(900)
def test_self_launch_eight_ranks_gloo_dry_run():
    """the rank count of the driver's scaling node (BASELINE configs[3]: 8 x 64 = 512 pairs): the launch path has counted to eight"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run-gloo", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=dict(_env(), OMP_NUM_THREADS="1"), cwd=ROOT, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    chk = _check(_one_json_line(r.stdout), 8)
    assert chk["gathered_rank_markers"] == [float(r + 1) for r in range(8)]


@pytest.mark.timeout(600)
def test_driver_launch_line_two_ranks_gloo_dry_run():
    """The round-end driver's command: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-gloo", "--gather-every-step"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry_run_checks"]["gather_ok"]
    assert "every step" in out["config"]["parallelism"]


def test_missing_gpus_is_reported_unmeasured_not_extrapolated():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3"], capture_output=True, text=True, env=_env(),
                       cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "UNMEASURED" in r.stderr and r.stdout.strip() == ""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("every_step", [False, True], ids=["gather-once", "gather-every-step"])
def test_rccl_path_of_bench_at_world_size_one(every_step):
    """RCCL itself entered on the GPU box.  The driver's launch line with ONE rank and SE3TN_FORCE_DIST=1:
    `nccl` backend (= RCCL), a real ncclBroadcast of the 54 MB packed blob into se3tn_bind_weights
    (dist.load_weights_everywhere), a real all_gather_into_tensor of the poses (once per timed region, and the asynchronous
    per-step variant), the parity block checked on the very batch that was computed from the broadcast weights, stdout exactly
    one JSON line.  A 1-GPU box cannot give a scaling curve and none is derived from this."""
    env = _env()
    env["SE3TN_FORCE_DIST"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
           "--track-frames", "0"]
    if every_step:
        cmd.append("--gather-every-step")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=840)
    assert r.returncode == 0, r.stderr[-4000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["backend"] == "nccl", {k: out.get(k) for k in ("n_gpus", "rccl_ranks", "backend")}
    assert out["parity"]["ok"], out["parity"]
    assert out["value"] > 0 and len(out["per_rank_pairs_per_s"]) == 1
    assert "dry_run" not in out
    if every_step:
        assert "every step" in out["config"]["parallelism"]
