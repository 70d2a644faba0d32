#!/usr/bin/env python3
"""bench.py -- RSA-2048 pkcs1v15 witness assignments / second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (RSAChip::modpow_public_key witness generation: chain kernel +
trace kernel through the C ABI) over one batch of synthetic signatures that is already resident in
HBM.  At N GPUs every rank processes its own shard of `--batch` signatures (weak scaling, no
data-path collective: signatures are independent); the only collectives are the configuration
broadcast before and the result gather after the timed region, plus the barrier / MAX-reduce that
brackets the timing.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (trace_kernel, HBM-write bound): algorithmic bytes per launch
                  / average launch duration measured with HIP events on the launch stream during
                  the timed steps.
  cpu_baseline -- the CPU oracle ("port" of the reference's path, oracle/h2r_oracle.c) timed on this
                  host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import halo2_rsa_amd as H  # noqa: E402
from halo2_rsa_amd import _lib  # noqa: E402
from halo2_rsa_amd.dist import DistEnv  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def pmc_traffic(kernel_name, batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: WRITE_SIZE + 2*FETCH_SIZE, KB units, collected in separate --pmc
    passes as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process,
    so the number is the measured one for the same kernel/batch, else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        ent = d.get(kernel_name)
        if ent is None:   # template arguments after the shape (workgroup size) are part of the profiler's name
            ent = next((v for k, v in d.items() if k.startswith(kernel_name[:-1] + ",")), None)
        if ent and ent.get("batch") == batch:
            return int(ent["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None

WORKLOADS = {
    # name: (limb_width, bits_len, exponent)
    "rsa2048_e65537": (64, 2048, 65537),   # BASELINE configs[1] (batch 1024) / configs[2] shards
    "rsa4096_w32_e65537": (32, 4096, 65537),  # configs[3]
    "rsa1024_e65537": (64, 1024, 65537),
    "rsa4096_e65537": (64, 4096, 65537),      # RSAChip's own limb width at 4096 bits
    # configs[4]: full 2048-step square-and-multiply (seeded 2048-bit exponent with the top bit set)
    "rsa2048_e2048bit": (64, 2048, random.Random(0x68327273 + 5).getrandbits(2048) | (1 << 2047)),
}


def synth_inputs(w, bits, batch, seed, golden_first=True):
    """Seeded synthetic batch: odd moduli with the top bit set, x uniform mod n; the
    reference's two valid and one invalid RSA-2048 signatures occupy elements 0-2."""
    rng = random.Random(seed)
    L = bits // w
    ns, xs = [], []
    if golden_first and (w, bits) == (64, 2048):
        with open(os.path.join(ROOT, "tests", "golden", "halo2_rsa_golden.json")) as f:
            for k in json.load(f)["rsa_kats"]:
                ns.append(int(k["n"])); xs.append(int(k["sig"]))
    while len(ns) < batch:
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        ns.append(n); xs.append(rng.randrange(n))
    ns, xs = ns[:batch], xs[:batch]
    un = H.UnassignedInteger.from_ints(ns, L, w)
    ux = H.UnassignedInteger.from_ints(xs, L, w)
    return ns, xs, un, ux


def cpu_baseline(w, bits, e, un, ux, target_seconds=12.0):
    """Time the CPU oracle (checker used as the reported baseline) on a bounded sample of the same synthetic batch:
    persistent threads (one per host core), each writing the full op-trace stream of its signatures into its own
    reusable buffer; a short calibration pass sizes the timed run to about `target_seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, pow_mod_fixed_exp_timed
    o = Oracle(w, bits // w)
    cores = os.cpu_count() or 1
    sample = min(un.limbs.shape[0], 1024)
    x, n = ux.limbs[:sample], un.limbs[:sample]
    cal, bad = pow_mod_fixed_exp_timed(o, x, n, e, 1, cores)
    passes = max(1, min(100000, int(target_seconds / max(cal, 1e-4))))
    while passes * sample < 16 * cores:   # at least 16 signatures per thread
        passes += 1
    sec, bad = pow_mod_fixed_exp_timed(o, x, n, e, passes, cores)
    assert bad == 0
    one = min(sample, 64)
    sec1, _ = pow_mod_fixed_exp_timed(o, x[:one], n[:one], e, 1, 1)
    return {"value": round(passes * sample / sec, 1), "unit": "assigns/s", "cores": cores, "kind": "port",
            "sample": "%d passes over %d signatures of the same synthetic batch (%.1f s, %d threads, %.0f signatures per "
                      "thread), full op-trace stream written to per-thread buffers" % (passes, sample, sec, cores, passes * sample / cores),
            "single_thread_value": round(one / sec1, 1), "single_thread_sample": "%d signatures, 1 thread" % one}


def ensure_built():
    """The bench needs libh2r.so (built in-tree by __graft_entry__.build(); git-ignored, but it travels with the
    working tree).  If it is missing, local rank 0 builds it with
    hipcc and the other ranks wait for it -- never eight concurrent compiles into one file."""
    from halo2_rsa_amd import _build
    if os.path.exists(_build.LIB):
        return
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        tmp = _build.build_lib(out=_build.LIB + ".tmp%d" % os.getpid())
        os.replace(tmp, _build.LIB)
    else:
        deadline = time.time() + 600
        while not os.path.exists(_build.LIB):
            if time.time() > deadline:
                raise RuntimeError("libh2r.so did not appear (local rank 0 builds it)")
            time.sleep(0.5)


def main():
    # the image exports NCCL_DEBUG=VERSION, which makes RCCL print a banner on stdout; rank 0's stdout is ONE JSON line
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ.pop("NCCL_DEBUG")
    ensure_built()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="signatures per GPU per step")
    ap.add_argument("--workload", default="rsa2048_e65537", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline-depth", type=int, default=2, help="buffer sets the pipelined calls rotate through (2..4)")
    ap.add_argument("--side-streams", type=int, default=1,
                    help="streams the record kernels alternate between; 2 lets consecutive record kernels overlap "
                         "(higher throughput, but each launch's duration then includes the overlap)")
    ap.add_argument("--verify", action="store_true",
                    help="time the whole verify_pkcs1v15_signature witness (in-field + modpow + encoded-message check) "
                         "instead of modpow_public_key alone (RSA-2048 workloads, pipelined mode)")
    ap.add_argument("--shared-modulus", action="store_true",
                    help="one key, many signatures (H2R_F_SHARED_MODULUS): every element uses element 0's modulus")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="developer: do not arm the C ABI's per-kernel event timing (roofline fields become null)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="fully stream-ordered calls (chain then trace per step) instead of the two-stream pipeline")
    args = ap.parse_args()

    env = DistEnv.from_environment(args.gpus, force=bool(os.environ.get("H2R_FORCE_DIST")))
    w, bits, e = WORKLOADS[args.workload]
    torch.cuda.set_device(env.local_rank)
    env.init("nccl")
    # configuration broadcast (rank 0 decides e / batch): the only pre-run collective
    cfg = env.broadcast_ints([e, args.batch, args.steps, args.warmup])
    e, batch, steps, warmup = cfg

    chip = H.BigIntChip(w, bits, device=env.local_rank)
    ns, xs, un, ux = synth_inputs(w, bits, batch, 0x68327273 + 2 + 1000 * env.rank)
    if args.shared_modulus:   # SURVEY 8d's shared-n variant: x reduced modulo the one modulus
        ns = [ns[0]] * batch
        xs = [x % ns[0] for x in xs]
        un, ux = H.UnassignedInteger.from_ints(ns[:1], bits // w, w), H.UnassignedInteger.from_ints(xs, bits // w, w)
    n_dev, x_dev = chip.assign_integer(un), chip.assign_integer(ux)
    pl = chip.pow_fixed_layout(e)
    dev = "cuda:%d" % env.local_rank
    # two buffer sets: in pipeline mode step k+1's chain kernel overlaps step k's trace kernel
    nbuf = 1 if args.no_pipeline else args.pipeline_depth
    elem_stride = pl.elem_stride
    verify = args.verify and not args.no_pipeline and (w, bits) == (64, 2048)
    if verify:   # whole verifier witness: the element also holds the in-field and encoded-message regions
        import ctypes
        vl = _lib.H2RVerifyLayout()
        eb = e.to_bytes((e.bit_length() + 7) // 8, "little")
        _lib.check(_lib.lib().h2r_verify_layout_fixed(chip._ctx, eb, len(eb), ctypes.byref(vl)), "h2r_verify_layout_fixed")
        elem_stride = vl.elem_stride
        hashed_dev = torch.randint(-2**62, 2**62, (batch, 4), dtype=torch.int64, device=dev)
        valids = [torch.zeros(batch, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    # zero-filled, so every page is resident before the first (possibly un-warmed) timed step touches it
    trace_bufs = [torch.zeros(batch * elem_stride, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    workspaces = [torch.zeros(chip.workspace_bytes(batch, pl.num_mul_mods), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    outs = [torch.empty((batch, chip.num_limbs), dtype=chip.torch_dtype, device=dev) for _ in range(nbuf)]
    statuses = [torch.zeros(batch, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    # modpow_public_key's assert_in_field witness (src/chip.rs:106): its own small buffer per set
    in_fields = [torch.zeros(batch * chip.in_field_layout()[0], dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    pipe = None if args.no_pipeline else H.Pipeline(chip, depth=args.pipeline_depth, side_streams=args.side_streams)
    counter = [0]

    def step():
        b = counter[0] % nbuf
        counter[0] += 1
        if pipe is None:
            chip.pow_mod_fixed_exp(x_dev, e, n_dev, want_trace=True, trace_buf=trace_bufs[b], check_in_field=True,
                                   workspace=workspaces[b], out=outs[b], status=statuses[b], in_field_buf=in_fields[b])
        elif verify:
            pipe.verify_pkcs1v15(x_dev, e, n_dev, hashed_dev, trace_bufs[b], workspaces[b], outs[b], valids[b], statuses[b])
        else:
            pipe.modpow_public_key(x_dev, e, n_dev, trace_bufs[b], workspaces[b], outs[b], statuses[b], in_fields[b])
        return b

    # initialisation that is not part of any step (code-object load, stream / event creation on first use): one call,
    # then the W warm-up steps the caller asked for
    step()
    counter[0] = 0
    for _ in range(warmup):
        step()
    if pipe is not None:
        pipe.join()
    torch.cuda.synchronize()
    _lib.profile_enable(0 if args.no_kernel_timing else 2 * steps + 8)
    env.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    if pipe is not None:
        pipe.join()   # every step's trace is complete before the clock stops
    torch.cuda.synchronize()
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    trace_ms = _lib.profile_read(_lib.KERNEL_TRACE)
    chain_ms = _lib.profile_read(_lib.KERNEL_CHAIN)
    _lib.profile_enable(0)

    # post-run: correctness of what was timed + the result gather (rank 0 receives every shard's x^e mod n)
    out, status = outs[last], statuses[last]
    assert int(status.max().item()) == 0 or (w, bits) != (64, 2048), "unexpected per-element status"
    got = H.AssignedInteger(out, w).to_big_uint()
    # the trace that was timed is the real thing: element 0 of the last step, byte-exact vs pow() through q*n+r
    tr = H.Trace(chip, trace_bufs[last], batch, pl)
    q0 = int.from_bytes(tr.plane(0, 0, "Q").tobytes(), "little")
    r0 = int.from_bytes(tr.plane(0, 0, "R").tobytes(), "little")
    assert xs[0] * xs[0] == q0 * ns[0] + r0, "first mul_mod record of the timed trace is wrong"
    for i in (0, 1, 2, batch - 1):
        if i < batch:
            assert got[i] == pow(xs[i], e, ns[i]), "GPU result differs from pow(x, e, n)"
    gathered = env.gather_to_rank0(out)
    if env.rank == 0:
        assert gathered.shape[0] == env.world * batch

    if env.rank == 0:
        # written (pow stream + the assert_in_field stream of modpow_public_key) + inputs read
        algo_bytes_per_assign = pl.stream_bytes + chip.in_field_layout()[1] + 2 * chip.num_limbs * chip.layout.limb_bytes
        trace_bytes_per_launch = batch * (pl.num_mul_mods * chip.layout.stream_bytes)         # trace_kernel's algorithmic output
        avg_trace_s = (sum(trace_ms) / len(trace_ms)) / 1e3 if trace_ms else float("nan")
        achieved = trace_bytes_per_launch / avg_trace_s / 1e9 if trace_ms else None
        line = {
            "metric": "RSA-2048 pkcs1v15 witness assigns/sec" if bits == 2048 else "RSA-%d witness assigns/sec" % bits,
            "value": round(env.world * batch * steps / dt, 1),
            "unit": "assigns/s",
            "n_gpus": env.world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * dt / steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u%d" % w, "data": "synthetic",
            "config": {"workload": "%s batch=%d per GPU, %d-bit limbs, full op-trace (%d B/assign)" %
                       (args.workload, batch, w, algo_bytes_per_assign),
                       "per_gpu_batch": batch, "global_batch": env.world * batch,
                       "path": "verify_pkcs1v15_signature (in-field + modpow + EM check)" if verify else "modpow_public_key",
                       "mul_mods_per_assign": pl.num_mul_mods, "parallelism": "signature-sharded x%d" % env.world,
                       "pipeline": ("chain k+1 || trace k, %d buffer sets, %d record stream(s)" % (args.pipeline_depth, args.side_streams))
                                   if pipe is not None else "none"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": pmc_traffic("trace_kernel<%d,%d>" % (w, chip.num_limbs), batch),
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this kernel at this batch, "
                                           "committed; PMC counters cannot be read from inside the bench process)",
                         "kernel": "trace_kernel<%d,%d>" % (w, chip.num_limbs),
                         "avg_launch_ms": round(1e3 * avg_trace_s, 4) if trace_ms else None,
                         "algorithmic_bytes_per_launch": trace_bytes_per_launch,
                         "chain_kernel_avg_ms": round(sum(chain_ms) / len(chain_ms), 4) if chain_ms else None},
            "whole_path_hbm_frac": round(env.world * batch * steps / dt * algo_bytes_per_assign / (env.world * HBM_PEAK_GBS * 1e9), 4),
        }
        if env.world == 1 and not args.no_cpu_baseline and not args.shared_modulus:
            line["cpu_baseline"] = cpu_baseline(w, bits, e, un, ux)
        print(json.dumps(line))
    env.finalize()


if __name__ == "__main__":
    main()
