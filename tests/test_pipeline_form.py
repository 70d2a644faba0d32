"""The pipelined RSA-2048 form is chosen by MEASUREMENT, not by an environment variable: h2r_pipeline_create_ex's two-queue form needs the
caller's stream and the two side streams on three hardware queues; the pipeline times three one-wave spinners the first time it meets a
caller stream and falls back to the one-launch step when they share a queue (h2r_pipeline_info reports which)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, random, sys, torch
sys.path.insert(0, %r)
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
chip = H.BigIntChip(64, 2048)
pipe = H.Pipeline(chip, depth=3, side_streams=2)
rng = random.Random(1)
B, e = 1024, 65537
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
x, n = chip.assign_integer(X), chip.assign_integer(N)
pl = chip.pow_fixed_layout(e)
info = pipe.info(B)
small = pipe.info(4096)
one = H.Pipeline(chip, depth=2, side_streams=1).info(B)
# a caller compiled against the struct before probe_span_ms existed (struct_size 24) is still served, and the field beyond it is not written
import ctypes
from halo2_rsa_amd import _lib
raw = _lib.H2RPipelineInfo(); raw.struct_size = 24; raw.probe_span_ms = -7.0
rc = _lib.lib().h2r_pipeline_info(pipe._p, chip._stream(), B, ctypes.byref(raw))
old_abi = [rc, raw.three_queues, raw.probe_span_ms]
bufs = [dict(t=torch.empty(B * pl.elem_stride, dtype=torch.uint8, device="cuda"), w=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
             o=torch.empty((B, 32), dtype=torch.int64, device="cuda"), s=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(3)]
for k in range(5):
    b = bufs[k %% 3]
    pipe.modpow_public_key(x, e, n, b["t"], b["w"], b["o"], b["s"])
pipe.join()
torch.cuda.synchronize()
got = H.AssignedInteger(bufs[4 %% 3]["o"], 64).to_big_uint()
ok = all(got[i] == pow(X[i], e, N[i]) for i in (0, 1, 500, 1023)) and int(bufs[4 %% 3]["s"].max().item()) == 0
print(json.dumps({"form": info.record_form, "three": info.three_queues, "probe_ms": info.probe_ms, "span_ms": info.probe_span_ms, "old_abi": old_abi, "form4096": small.record_form, "three4096": small.three_queues,
                  "form_one_stream": one.record_form, "ok": ok}))
""" % ROOT


def _probe(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", PROBE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_queue_form_only_when_the_streams_overlap():
    free = _probe({"GPU_MAX_HW_QUEUES": "8"})
    assert free["ok"] and free["three"] in (0, 1)
    assert (free["form"] == 1) == (free["three"] == 1)            # two-queue iff the probe saw three queues
    assert free["form4096"] == 0 and free["three4096"] == 2        # above 2,048 per call: the step, nothing to measure
    assert free["form_one_stream"] == 0
    # ONE hardware queue for the whole process: every stream shares it -- the probe must see that, and the calls take the one-launch step
    shared = _probe({"GPU_MAX_HW_QUEUES": "1"})
    assert shared["ok"], shared
    assert shared["three"] == 0 and shared["form"] == 0, shared
    assert shared["probe_ms"] > 0.29 and shared["span_ms"] > 0.29, shared   # three 150 us spinners back to back, on both clocks
    if free["three"] == 1:
        assert 0.150 <= free["span_ms"] <= 0.157, free             # the device-clock span decides (profiles/r06_queue_probe.txt)
    assert free["old_abi"][0] == 0 and free["old_abi"][1] == free["three"] and free["old_abi"][2] == -7.0, free
    # the default (no variable at all): whatever the probe finds, the results are right
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", PROBE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    dflt = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert dflt["ok"] and (dflt["form"] == 1) == (dflt["three"] == 1)
