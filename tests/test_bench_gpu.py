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
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2        # (ranks: what the communicator / process group reports)
    assert "communicator" in line["config"] and "rccl_version" in line["config"]
    assert "audited in place" in line["config"]["post_run_check"]       # one full call per shard checked on the device, verdicts gathered
    assert line["config"]["global_batch"] == 2 * 2 * 640 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    assert "scale_anchor" not in line                                   # (the anchor belongs to the N = 1 line)


def test_bench_n1_line_carries_the_scale_anchor():
    """The driver's N = 1 point is config 2 (1,024 signatures per step), its N > 1 points run 8,192 per GPU as four calls of 2,048: the
    N = 1 line also reports that per-GPU workload on the one GPU (`scale_anchor`), the like-for-like origin of a 1 -> 8 curve."""
    line = _run(["--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--pmc-traffic", "off", "--placement-candidates", "4"])
    sa = line["scale_anchor"]
    assert "error" not in sa, sa
    assert sa["per_gpu_batch"] == 8192 and sa["calls_per_step"] == 4 and sa["value"] > 0 and 0.2 < sa["roofline_frac"] < 1.0
    assert line["config"]["warmup_calls_total"] >= 1 + 2


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


def test_bench_advice_line():
    """--advice: the prover-consumable witness as the product -- one line with path = "advice image", the cells kernel's roofline on the
    pow rows it writes, and bench.py's own check of the timed image against the image of a call with records."""
    line = _run(["--advice", "--batch", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pmc-traffic", "off", "--placement-candidates", "3"])
    assert line["config"]["path"] == "advice image" and line["roofline"]["kernel"].startswith("cells_kernel")
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 256 * (2 + 19 * 3974) * 160
    assert line["value"] > 0 and 0.2 < line["roofline"]["frac"] < 1.0
    assert line["config"]["buffer_placement"]["candidates"] == 3


def test_bench_advice_in_the_provers_representation():
    """--advice --columns --montgomery: planar Montgomery-form columns; the line says so, and bench.py's own post-run checks ran (the timed
    image against the image of the records in the same representation, and h2r_advice_check over every row of the last timed image)."""
    line = _run(["--advice", "--columns", "--montgomery", "--batch", "128", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pmc-traffic", "off",
                 "--placement-candidates", "0"])
    rp = line["config"]["representation"]
    assert rp["columns"] and rp["montgomery"] and rp["col_stride"] % 4096 == 0 and rp["col_stride"] >= (2 + 1532 + 19 * 3974) * 32
    assert "Montgomery" in line["roofline"]["kernel"] and line["value"] > 0
    assert "0 violations" in line["config"]["post_run_audit"]


def test_bench_advice_whole_verify_element():
    """--advice --verify: the whole verify_pkcs1v15_signature element through h2r_pipeline_verify_pkcs1v15_advice; bench.py's own checks ran
    (is_valid and the timed image against the record-based call's, h2r_advice_check with RSAChip's table over every row)."""
    line = _run(["--advice", "--verify", "--batch", "128", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--pmc-traffic", "off",
                 "--placement-candidates", "0"])
    assert line["config"]["path"] == "advice image (verify element)" and "77219 rows" in line["config"]["workload"]
    assert "h2r_pipeline_verify_pkcs1v15_advice" in line["config"]["pipeline"]
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 128 * (2 + 19 * 3974) * 160
    assert line["value"] > 0 and "0 violations" in line["config"]["post_run_audit"]


def test_bench_records_free_flow_line():
    """--records-free-flow: image + multiplicities from the image + A', S' per batch; bench.py's own checks ran (pow(), multiplicities against the records')."""
    line = _run(["--records-free-flow", "--batch", "64", "--steps", "3", "--warmup", "2"])
    assert line["unit"] == "circuits/s" and line["value"] > 0 and 0.05 < line["roofline"]["frac"] < 1.0
    bb = line["config"]["bytes_per_batch"]
    assert bb["advice_image"] == 64 * 77219 * 160 and bb["lookup_columns"] == 2 * 64 * 5 * ((1 << 17) - 6) * 32


def test_bench_lookup_line():
    line = _run(["--lookup", "--batch", "32", "--steps", "3", "--warmup", "1", "--pmc-traffic", "off", "--placement-candidates", "3"])
    assert line["unit"] == "GB/s" and line["roofline"]["kernel"] == "lookup_fill_kernel" and 0.1 < line["roofline"]["frac"] < 1.0
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 2 * 32 * 5 * ((1 << 17) - 6) * 32
    assert line["config"]["buffer_placement"]["candidates"] >= 3 and 0 < line["whole_call_hbm_frac"] <= line["roofline"]["frac"] + 1e-6


def test_bench_default_line_carries_the_sub_runs():
    """What the driver runs (`--gpus 1 --steps K --warmup W`, nothing else) also reports, from fresh processes: the headline as allocated,
    the advice image in both representations (each audited by h2r_advice_check), BASELINE configs 4 and 5, the lookup argument."""
    line = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--pmc-traffic", "off"], timeout=900)
    for key in ("plain_allocations", "advice", "advice_columns_montgomery", "advice_verify_element", "other_configs", "records_free_flow", "lookup", "scale_anchor"):
        assert key in line, key
    assert line["plain_allocations"]["value"] > 0 and 0.2 < line["plain_allocations"]["frac"] < 1.0
    for key in ("advice", "advice_columns_montgomery", "advice_verify_element"):
        assert line[key]["error"] is None and line[key]["value"] > 0 and "0 violations" in line[key]["audit"], line[key]
    assert line["other_configs"]["C4"]["value"] > 0 and line["other_configs"]["C5"]["value"] > 0
    assert line["lookup"]["error"] is None and line["lookup"]["whole_call_GBps"] > 0
    assert line["records_free_flow"]["error"] is None and line["records_free_flow"]["value"] > 0
    assert line["config"]["pipeline_form"]["record_form"] in ("two-queue", "one-launch step")
