"""The driver-facing contracts: bench.py prints ONE JSON line with the agreed keys (GPU), refuses to run without a GPU
(no CPU fallback), and __graft_entry__.smoke() passes on the device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*args, timeout=280):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run_bench("--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_line_keeps_the_contract():
    r = _run_bench("--steps", "24", "--warmup", "8", "--no-cpu-baseline", "--latency")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 8
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # the frame the timed loop rendered is the golden config-3 frame: 4 589 295 rays, and value = rays / time
    assert d["config"]["rays_per_frame"] == 4589295
    assert abs(d["value"] - 4589295 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["avg_launch_ms"] > 0 and rf["launches_per_call"] == 4
    assert 0.1 < d["config"]["single_frame_latency_ms"] < 50


@pytest.mark.gpu
def test_smoke_entry_point():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"], capture_output=True, text=True,
                       timeout=280, cwd=ROOT)
    assert r.returncode == 0 and "smoke ok" in r.stdout, (r.stdout + r.stderr)[-2000:]


@pytest.mark.gpu
def test_bench_multi_rank_control_flow_on_one_gpu():
    """bench.py under torch.distributed.run with two ranks that share GPU 0 and exchange over gloo (RCCL refuses two ranks
    on one device): the N > 1 branch of the script end to end — tile partition, per-rank ray counts summed, gather to
    rank 0, device assemble.  The assembled frame is the single-GPU frame; the throughput it prints is not a measurement."""
    one = _run_bench("--steps", "8", "--warmup", "8", "--no-cpu-baseline")
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "8",
                        "--rehearse-on-one-gpu"], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 32 and "rehearsal" in d
    assert d["config"]["rays_per_frame"] == ref["config"]["rays_per_frame"] == 4589295      # the two parts' rays add up
    assert d["config"]["frame_checksum"] == ref["config"]["frame_checksum"]                # gathered + assembled = the 1-GPU frame
    assert d["config"]["frames_per_call"] == 16 and "cpu_baseline" not in d
