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
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames-per-step", "6", "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1
    assert j["value"] > 0 and j["unit"] == "MP/s" and j["scaling"] == "weak"
    assert j["roofline"]["bound"] in ("hbm", "mfma") and 0 < j["roofline"]["frac"] < 1


def test_bench_plain_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames-per-step", "6",
           "--cpu-sample-frames", "2", "--extra-frames", "8"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert out.stdout.strip().splitlines()[-1].startswith("{"), "the JSON line must be the last line of stdout"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "c3_2pass", "end_to_end", "parity"):
        assert k in j, k
    # the side legs: 2-pass (north_star's target config), host->host through the plugin API, parity vs the oracle
    assert j["c3_2pass"]["value"] > 0 and j["c3_2pass"]["unit"] == "MP/s"
    assert j["config"]["mode"] == "exact"
    assert "fast_mode" not in j                                   # the non-bit-exact mode left the product in round 4
    assert j["end_to_end"]["value"] > 0 and "RNLHandler_Process" in j["end_to_end"]["what"]
    assert j["parity"]["mismatches"] == 0 and j["parity"]["psnr"] == "inf" and j["parity"]["pixels"] == 3840 * 2160
    if "stream" in j:
        assert j["stream"]["value"] > 0
    assert "AVX" in j["cpu_baseline"]["sample"]                       # names the ISA the CPU leg was built for
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
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-step", "8", "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "frame-shard x2"
    assert j["value"] > 0 and abs(j["value"] - j["config"]["fps"] * 3840 * 2160 / 1e6) < 1.0


def test_bench_two_ranks_report_a_streamed_leg():
    """N > 1 without --no-extras: beside the HBM-resident headline every rank streams host frames through its ring at the same
    time, so a scaling record sees PCIe / host-memory contention (SURVEY s8e's scaling risks)."""
    env = dict(os.environ, RAISR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "8", "--no-cpu-baseline", "--extra-frames", "12"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert j["n_gpus"] == 2 and j["stream"]["value"] > 0 and j["stream"]["frames_per_rank"] == 12
    assert "c3_2pass" not in j                      # the single-GPU side legs stay single-GPU


def test_bench_gpus_flag_is_honoured_or_fails_loudly():
    """`python bench.py --gpus 2` (no torchrun around it) must start two ranks itself; on a box with fewer than two
    devices it has to fail -- never print a line for a world size it did not run."""
    import torch
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "8",
           "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if torch.cuda.device_count() >= 2:
        assert out.returncode == 0, out.stderr[-3000:]
        assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    else:
        assert out.returncode != 0
        assert not lines, lines
        assert "needs 2 visible devices" in (out.stderr + out.stdout)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "8", "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_single_process_multi_device_ring():
    """`bench.py --gpus N --single-process`: ONE process drives N device slots through raisr_hip_stream_create_multi (the C++-host
    counterpart of the torchrun launch).  One GPU here: RAISR_BENCH_DEVICES=0,0 lists it twice."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--steps", "2", "--warmup", "1", "--frames-per-step", "24",
           "--config", "C1"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, RAISR_BENCH_DEVICES="0,0"))
    assert out.returncode == 0, out.stderr[-3000:]
    j = _one_json_line(out.stdout)
    assert j["n_gpus"] == 2 and j["config"]["devices"] == [0, 0] and j["value"] > 0 and j["unit"] == "MP/s"
    assert "single process" in j["mode"] and j["stream"]["frames"] == 48
