"""bench.py as the driver invokes it (gpu-marked: it runs the hot path)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ, **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]      # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (round 2 raised: it needed an external torch.distributed.run):
    the script re-executes itself under torch.distributed.run.  On this one-GPU box both ranks share device 0 over gloo
    (H2R_BENCH_ONE_GPU=1: a functional check of the N > 1 path -- shards, result gather, per-shard checks -- not a measurement)."""
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "640", "--chunks", "2", "--no-cpu-baseline",
                 "--placement-candidates", "0", "--pmc-traffic", "off"], {"H2R_BENCH_ONE_GPU": "1"})
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2
    assert line["config"]["global_batch"] == 2 * 2 * 640 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


def test_bench_default_line_has_roofline_and_cpu_baseline():
    line = _run(["--steps", "5", "--warmup", "2", "--placement-candidates", "4"])
    assert line["n_gpus"] == 1 and line["metric"].startswith("RSA-2048") and line["unit"] == "assigns/s"
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.2 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # traffic is measured by this very run when rocprofv3 is on the box (else the committed figure): within 0.9 .. 1.3 x algorithmic
    assert rf["traffic"] is None or 0.9 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.3, rf
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "logical_cpus" in cb["host"]


def test_bench_long_exponent_and_two_producers():
    """A small BASELINE-config-5-shaped run (2,048-bit exponent: every call walked as 16 segments of the exponent's bits, 16 chain + 16
    record launches per call) and `--producers 2` (two pipelines on two streams, calls alternate): both finish with the post-run checks
    of bench.py green (results = pow(x, e, n), first record of the timed trace)."""
    line = _run(["--workload", "rsa2048_e2048bit", "--batch", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                 "--placement-candidates", "0", "--pmc-traffic", "off"])
    assert line["value"] > 0 and line["config"]["mul_mods_per_assign"] > 2048
    assert line["roofline"]["launches_per_call"] == 16
    line = _run(["--producers", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--placement-candidates", "0", "--pmc-traffic", "off"])
    assert line["value"] > 0 and "2 producers" in line["config"]["pipeline"]
