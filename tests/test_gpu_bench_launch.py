"""-m gpu: bench.py launched the way the driver launches the N-GPU runs (torch.distributed.run, RCCL backend).
With one rank the filter-bank broadcast, the barrier and the max-over-ranks reduction still go through RCCL on
the GPU, so the collective plumbing of the multi-GPU runs is exercised on a 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_torchrun_single_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames-per-step", "6", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1
    assert j["value"] > 0 and j["unit"] == "MP/s" and j["scaling"] == "weak"
    assert j["roofline"]["bound"] in ("hbm", "mfma") and 0 < j["roofline"]["frac"] < 1


def test_bench_plain_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames-per-step", "6",
           "--cpu-sample-frames", "2"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["config"]["workload"].startswith("C2")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in j["cpu_baseline"], k
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_on_one_gpu_plumbing():
    """world_size 2 on the one GPU of the test box (gloo instead of RCCL, which refuses two ranks on one device): rank 1
    receives the filter bank by broadcast, the ranks own disjoint frames, the time is the max over ranks, one JSON line."""
    env = dict(os.environ, RAISR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-step", "8", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "frame-shard x2"
    assert j["value"] > 0 and abs(j["value"] - j["config"]["fps"] * 3840 * 2160 / 1e6) < 1.0
