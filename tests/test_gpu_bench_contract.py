"""bench.py contract on the GPU box: one JSON line with the required keys at N = 1, and the N = 2 code path (site partition x2,
max-over-ranks timing, packed all-reduce of the results) rehearsed with two ranks sharing the one GPU over gloo
(BENCH_REHEARSAL=1; on the 8-GPU node the same code runs over RCCL with one rank per device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline"}


def last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_small():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--sites", "8", "--cells", "10", "--lld", "12",
                        "--no-green"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    rf = d["roofline"]
    # fractions are quoted in the flops the operator's block structure requires: a kernel cannot beat the pipe's peak in them
    assert d["value"] > 0 and rf["bound"] == "mfma" and 0 < rf["frac"] <= 1 and 0 < rf["frac_step"] <= 1
    assert rf["frac"] <= rf["frac_algorithmic"] and 0.5 <= rf["required_per_algorithmic"] <= 1.0      # collinear bcc Fe: 14 of 15 blocks spin-diagonal
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_rehearsal():
    """`bench.py --gpus 2` with no launcher around it: the script starts its two ranks itself (BENCH_REHEARSAL: they share the box's
    one GPU and reduce over gloo; on a multi-GPU node the same command runs one rank per device over RCCL)."""
    env = dict(os.environ, BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--sites", "8", "--cells", "10", "--lld", "12",
           "--master-port", "29533"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 2 and "x2" in d["config"]["parallelism"] and d["value"] > 0
    assert "cpu_baseline" not in d                    # CPU leg on rank 0 at N = 1 only
    assert "gloo" in d["config"]["collective"]


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr


def test_bench_step_fractions():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--sites", "8", "--cells", "10", "--lld", "12",
                        "--no-green", "--no-cpu", "--recur", "chebyshev"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    rf = d["roofline"]
    assert abs(rf["frac_step_algorithmic"] - d["value"] * 1e-3 / d["n_gpus"] / rf["peak"]) < 1e-9 and rf["frac"] == rf["frac_kernel"]
    assert 0 < rf["frac_step"] <= rf["frac_step_algorithmic"] and rf["frac_step"] <= 1
    assert "Chebyshev" in d["config"]["workload"]
