#!/usr/bin/env python3
"""bench.py -- RSA-2048 pkcs1v15 witness assignments / second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (RSAChip::modpow_public_key witness generation: assert_in_field
witness + chain kernel + record kernel through the C ABI) over one batch of synthetic signatures that
is already resident in HBM.
  --gpus 1 (default): BASELINE configs[1] -- one 1,024-signature call per step.
  --gpus N > 1: BASELINE configs[2] -- the seeded global batch (8,192 x N signatures: 65,536 at N = 8) is
    sharded contiguously (halo2_rsa_amd.dist.shard_range); every rank walks its 8,192-signature shard as
    four pipelined 2,048-signature calls per step and keeps the traces resident on its own GPU.  Weak
    scaling, no data-path collective (signatures are independent); the only collectives are the
    configuration broadcast before and the result all-gather after the timed region (rank 0 checks
    samples of EVERY shard against pow()), plus the barrier / MAX-reduce that brackets the timing.

The calls are pipelined (h2r_pipeline_*): RSA-2048's chain kernels on the caller's stream, its record kernels alternating between
two side streams of the pipeline over three buffer sets (call k + 1's record kernel starts while call k's tail drains); the other
shapes one launch per call (step_kernel: records of call k + chains of call k + 1).  --pipeline-depth / --side-streams select the form.
  --advice: the prover-consumable witness as the product (h2r_pipeline_modpow_public_key_advice; roofline on cells_kernel).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (trace_kernel / step_kernel / cells_kernel, HBM-write bound): algorithmic bytes per launch /
                  average launch time -- for launches that overlap on two side streams their PERIOD (timed wall time / launches;
                  avg_launch_ms_in_flight carries a launch's own duration from HIP events stamped by the dispatch), for the one-launch
                  steps the distance of two HIP events on the launch stream / launches, with --per-launch-timing per-launch HIP events.
  cpu_baseline -- the CPU oracle ("port" of the reference's path, oracle/h2r_oracle.c) timed on this
                  host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import random
import sys
import time

# the image exports NCCL_DEBUG=VERSION, which makes RCCL print a banner on stdout when a communicator is created (the library
# reads the variable when it is loaded, i.e. at `import torch`); rank 0's stdout is ONE JSON line
if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
    os.environ.pop("NCCL_DEBUG")

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), handed out in creation order, and kernels of two
# streams that share a queue do not overlap: the pipeline's side streams then sometimes land behind torch's or RCCL's (under torchrun the
# overlapped RSA-2048 form ran at 4.8 M assigns/s instead of 5.5 M on every run, a plain process now and then at 5.3 instead of 5.45 M).
# Eight queues make room (same-box A/B: profiles/r04_two_queue.txt).  The runtime reads the variable when it initialises: before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import halo2_rsa_amd as H  # noqa: E402
from halo2_rsa_amd import _lib  # noqa: E402
from halo2_rsa_amd.dist import DistEnv, shard_range  # noqa: E402

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
    "rsa1536_e65537": (64, 1536, 65537),      # num_limbs = 24
    "rsa3072_e65537": (64, 3072, 65537),      # num_limbs = 48: not a power of two
    "rsa4096_e65537": (64, 4096, 65537),      # RSAChip's own limb width at 4096 bits
    # configs[4]: full 2048-step square-and-multiply (seeded 2048-bit exponent with the top bit set)
    "rsa2048_e2048bit": (64, 2048, random.Random(0x68327273 + 5).getrandbits(2048) | (1 << 2047)),
}


GLOBAL_SEED = 0x68327273


def synth_element(w, bits, g, golden):
    """Element g of the seeded global synthetic batch: an odd modulus with the top bit set and x uniform
    mod n; the reference's two valid and one invalid RSA-2048 signatures are global elements 0-2.  Any rank can
    regenerate any element (rank 0 re-derives samples of every shard for the post-run check)."""
    if golden is not None and g < len(golden):
        return int(golden[g]["n"]), int(golden[g]["sig"])
    rng = random.Random((GLOBAL_SEED << 24) + g)
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    return n, rng.randrange(n)


def load_golden(w, bits):
    if (w, bits) != (64, 2048):
        return None
    with open(os.path.join(ROOT, "tests", "golden", "halo2_rsa_golden.json")) as f:
        return json.load(f)["rsa_kats"]


def to_limbs(values, w, bits):
    """maingate::decompose_big: little-endian limbs, [len(values), bits / w]."""
    raw = b"".join(int(v).to_bytes(bits // 8, "little") for v in values)
    return np.frombuffer(raw, dtype=np.uint64 if w == 64 else np.uint32).reshape(len(values), bits // w).copy()


def synth_inputs(w, bits, lo, hi):
    """Global elements [lo, hi) of the synthetic batch: (ns, xs, UnassignedInteger n, UnassignedInteger x)."""
    golden = load_golden(w, bits)
    pairs = [synth_element(w, bits, g, golden) for g in range(lo, hi)]
    ns, xs = [p[0] for p in pairs], [p[1] for p in pairs]
    return ns, xs, H.UnassignedInteger(to_limbs(ns, w, bits)), H.UnassignedInteger(to_limbs(xs, w, bits))


def cpu_baseline(w, bits, e, un, ux, target_seconds=8.0):
    """Time the CPU oracle (checker used as the reported baseline) on a bounded sample of the same synthetic batch:
    persistent threads (one per host core), each writing the full op-trace stream of its signatures into its own
    reusable buffer; a short calibration pass sizes the timed run to about `target_seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, pow_mod_fixed_exp_timed
    o = Oracle(w, bits // w)
    host = host_cpu_info()
    # threads = what the host may actually run at once: the GPU boxes expose 256 logical CPUs under a cgroup quota of 16
    # (cpu.max "1600000 100000"); 256 threads on 16 CPUs' worth of time measured 10x one thread, not 256x
    cores = max(1, min(host.get("logical_cpus") or 1, host.get("affinity_cpus") or 1 << 30, host.get("quota_cpus") or 1 << 30))
    sample = min(un.limbs.shape[0], 1024)
    x, n = ux.limbs[:sample], un.limbs[:sample]
    pow_mod_fixed_exp_timed(o, x, n, e, 1, cores)                         # threads / pages warm
    cal, bad = pow_mod_fixed_exp_timed(o, x, n, e, 2, cores)
    cal /= 2
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
            "host": host,
            "single_thread_value": round(one / sec1, 1), "single_thread_sample": "%d signatures, 1 thread" % one,
            # what the host actually delivers: os.cpu_count() threads may share far fewer physical cores / a CPU quota
            "parallel_speedup": round((passes * sample / sec) / (one / sec1), 1)}


def host_cpu_info():
    """What the host can deliver, for reading cpu_baseline: logical CPUs, the CPUs this process may run on, distinct
    physical cores (/proc/cpuinfo), and the container's CPU quota (cgroup v2 cpu.max, v1 cfs quota)."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        cores, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        if cores:
            info["physical_cores"] = len(cores)
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                raw = f.read().strip()
            info["cgroup_" + os.path.basename(path)] = raw
            if path.endswith("cpu.max"):
                q, per = raw.split()[0], raw.split()[1]
                if q != "max":
                    info["quota_cpus"] = max(1, int(int(q) / int(per)))
            elif int(raw) > 0:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    info["quota_cpus"] = max(1, int(int(raw) / int(f.read().strip())))
            break
        except Exception:
            continue
    return info


def measured_pmc_traffic(argv_workload, kernel_prefix, timeout_s=150):
    """HBM bytes per launch of the dominant kernel, measured NOW: two rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE: separate
    runs, counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes) of this very script on a short run of the
    same workload.  Units: KB per dispatch (calibrated in round 1: a 1024-byte fill reports WRITE_SIZE = 1.0); FETCH_SIZE is
    doubled (gfx950 reports half of wide coalesced reads).  Returns (bytes, description) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROFILER_", "ROCP_TOOL", "ROCPROF_")) for k in os.environ):
        return None, "already running under a profiler"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="h2r_pmc_", dir="/tmp")
    try:
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "r", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                   "--pmc-traffic", "off", "--scale-anchor", "off", "--placement-candidates", "0"] + argv_workload
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            except Exception as ex:
                return None, "rocprofv3 --pmc %s pass failed: %s" % (counter, str(ex)[:80])
            fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
            if not fs:
                return None, "no counter_collection.csv from the %s pass" % counter
            acc = []
            with open(fs[0]) as f:
                for r in csv.DictReader(f):
                    if r.get("Counter_Name") == counter and ("h2r::" + kernel_prefix) in r.get("Kernel_Name", ""):
                        acc.append(float(r["Counter_Value"]))
            if not acc:
                return None, "no %s dispatches in the %s pass" % (kernel_prefix, counter)
            vals[counter] = sum(acc) / len(acc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    hbm = int(round(1024 * (vals["WRITE_SIZE"] + 2 * vals["FETCH_SIZE"])))
    return hbm, ("measured in this run: rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes of bench.py "
                 "--steps 4, plain allocations), mean per %s dispatch; WRITE_SIZE %.0f KB + 2 x FETCH_SIZE %.0f KB" %
                 (kernel_prefix, vals["WRITE_SIZE"], vals["FETCH_SIZE"]))


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the driver's
    own multi-GPU command line does exactly this; WORLD_SIZE is then set and this is skipped)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def check_gathered(gathered, audit_all, world, shard, total, w, bits, e, sample=True, strict=True, expect=None):
    """Rank 0's post-run check of the gathered results: sizes, audit verdict bytes, and samples from EVERY shard against pow() of the
    regenerated inputs (expect = None), or against `expect(x, n)` (the dry run's stand-in for the GPU result).  Returns (ok, message)."""
    try:
        assert gathered.shape[0] == total and audit_all.shape[0] == total, "gathered %d results / %d verdict bytes for a batch of %d" % (gathered.shape[0], audit_all.shape[0], total)
        n_flagged = int((audit_all != 0).sum().item())
        assert n_flagged == 0 or not strict, "%d elements with a status or a violated witness relation" % n_flagged
        if sample:
            golden = load_golden(w, bits)
            host = gathered.cpu().numpy().view(np.uint64 if w == 64 else np.uint32)   # [total, num_limbs] little-endian limbs
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                for off in sorted({0, (hi - lo) // 2 + 1 if hi - lo > 2 else 0, hi - lo - 1}):
                    g = lo + off
                    n_g, x_g = synth_element(w, bits, g, golden)
                    v = sum(int(t) << (w * i) for i, t in enumerate(host[g]))
                    if not (golden is not None and g < 3 and x_g >= n_g):
                        want = pow(x_g, e, n_g) if expect is None else expect(x_g, n_g)
                        assert v == want, "shard %d element %d differs from %s" % (r, off, "pow(x, e, n)" if expect is None else "the dry run's stand-in result")
    except AssertionError as ex:
        return False, str(ex)
    return True, ""


def dist_dry_run(args):
    """--dry-run-dist: the N > 1 plumbing of this script WITHOUT the GPU -- argument handling, the relaunch under torch.distributed.run,
    rendezvous, configuration broadcast, contiguous shards of the seeded global batch (a batch the ranks need not divide: --dry-total),
    barrier + MAX timing, the gather of results + verdict bytes to rank 0, rank 0's samples from every shard, the agreed verdict and the
    exit status.  Only the GPU call is replaced: the stand-in "result" of element (x, n) is x itself, its verdict byte 0.  gloo on CPU.
    H2R_DRY_FAIL_RANK=r: rank r reports a failed audit of its shard;  H2R_DRY_LATE_RANK=r: rank r arrives 5 s late at the first collective."""
    from halo2_rsa_amd.dist import finish_with_verdict
    env = DistEnv.from_environment(args.gpus, force=bool(os.environ.get("H2R_FORCE_DIST")))
    w, bits, e = WORKLOADS[args.workload]
    if env.rank == int(os.environ.get("H2R_DRY_LATE_RANK", "-1")):
        time.sleep(5.0)
    env.init("gloo")
    chunks = args.chunks if args.chunks else (1 if args.gpus == 1 else 4)
    chunk = args.batch if args.batch else (1024 if args.gpus == 1 else 8192 // chunks)
    e, chunk, chunks, steps, warmup, total = env.broadcast_ints([e, chunk, chunks, args.steps, args.warmup, args.dry_total or 0])
    total = total or chunk * chunks * env.world
    lo, hi = shard_range(total, env.rank, env.world)
    ns, xs, un, ux = synth_inputs(w, bits, lo, hi)
    env.barrier()
    t0 = time.perf_counter()
    result = torch.from_numpy(ux.limbs.view(np.int64).copy())          # <- where the hot path would run
    audit = torch.zeros(hi - lo, dtype=torch.uint8)
    if env.rank == int(os.environ.get("H2R_DRY_FAIL_RANK", "-1")) and hi > lo:
        audit[(hi - lo) // 2] = 1
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    gathered, audit_all = env.gather_to_rank0(result, audit, total=total)
    ok, msg = True, ""
    n_mine = int((audit != 0).sum().item())
    if n_mine:
        ok, msg = False, "%d elements of this rank's shard with a status or a violated witness relation" % n_mine
    elif env.rank == 0:
        ok, msg = check_gathered(gathered, audit_all, env.world, None, total, w, bits, e, sample=True, strict=False, expect=lambda x, n: x)
    finish_with_verdict(env, ok, msg)
    if env.rank == 0:
        sizes = [shard_range(total, r, env.world)[1] - shard_range(total, r, env.world)[0] for r in range(env.world)]
        print(json.dumps({"dry_run": True, "n_gpus": env.world, "global_batch": total, "shard_sizes": sizes, "steps": steps, "warmup": warmup,
                          "e": e, "max_over_ranks_s": round(dt, 6), "collective_backend": "torch.distributed gloo (CPU)",
                          "post_run_check": "samples from every shard regenerated on rank 0; verdict agreed by every rank"}))
    env.finalize()


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


def advice_bench(args):
    """--advice: the PROVER-CONSUMABLE witness as the product.  One step = one 1,024-signature RSAChip::modpow_public_key call whose
    output is the 5-column advice image (what the reference writes cell by cell, main_gate.mul_add
    big_integer/chip.rs:408, range_chip.assign :590, :598, :880-885, is_equal_muled :851-893): chain kernel (no record planes) +
    assert_in_field witness and its rows + cells_kernel, which writes the pow rows directly from the operands.  The chain kernels of
    call k + 1 run next to the cells kernel of call k (two workspaces, two images): ONE export per call,
    h2r_pipeline_modpow_public_key_advice (the pipeline owns the side streams and the events).  roofline: cells_kernel, HBM-write bound, algorithmic bytes = the pow rows it writes
    (12,081,280 B per RSA-2048 e = 65537 element)."""
    import ctypes
    env = DistEnv.from_environment(args.gpus, force=bool(os.environ.get("H2R_FORCE_DIST")))
    w, bits, e = WORKLOADS[args.workload]
    one_gpu = bool(os.environ.get("H2R_BENCH_ONE_GPU"))
    if one_gpu:
        env.local_rank = 0
    torch.cuda.set_device(env.local_rank)
    if env.world > 1 or env.force:
        env.init("gloo" if one_gpu else "nccl")
    chunk = args.batch if args.batch else 1024
    e, chunk, steps, warmup = env.broadcast_ints([e, chunk, args.steps, args.warmup])
    global_batch = chunk * env.world
    lo, hi = shard_range(global_batch, env.rank, env.world)
    # --columns / --montgomery: the image in the PROVER'S representation (h2r_advice_repr): planar column vectors (here 4 KB-aligned
    # columns just long enough for the element's rows; a prover's are 2^k rows) of x * R mod p cells
    probe = H.BigIntChip(w, bits, device=env.local_rank)
    pl = probe.pow_fixed_layout(e)
    L = _lib.lib()
    # --verify: the WHOLE RSAChip::verify_pkcs1v15_signature element (src/chip.rs:128-199: is_eq seed, assert_in_field, the pow rows, the
    # encoded-message check) through h2r_pipeline_verify_pkcs1v15_advice -- no records, a 10 KB witness per element
    whole = bool(args.verify)
    vl = None
    if whole:
        assert w == 64, "--advice --verify: RSAChip::LIMB_WIDTH = 64 (src/chip.rs:203)"
        eb = e.to_bytes((e.bit_length() + 7) // 8, "little")
        full, vl = _lib.H2RVerifyLayout(), _lib.H2RVerifyLayout()
        _lib.check(L.h2r_verify_layout_fixed(probe._ctx, eb, len(eb), ctypes.byref(full)), "h2r_verify_layout_fixed")
        _lib.check(L.h2r_verify_layout_compact(probe._ctx, ctypes.byref(full), ctypes.byref(vl)), "h2r_verify_layout_compact")
        sec4 = (ctypes.c_uint64 * 4)()
        rows = int(L.h2r_verify_advice_rows(probe._ctx, ctypes.byref(vl), sec4))
        pow_rows, pow_off = int(sec4[2]), int(sec4[0]) + int(sec4[1])
        sec = [pow_off, pow_rows]
    else:
        sec = (ctypes.c_uint64 * 2)()
        rows = int(L.h2r_modpow_public_key_advice_rows(probe._ctx, ctypes.byref(pl), sec))
        pow_rows, pow_off = int(sec[1]), int(sec[0])
    col_stride = ((rows * 32 + 4095) // 4096) * 4096 if args.columns else 0
    chip = H.BigIntChip(w, bits, device=env.local_rank, columns=args.columns, montgomery=args.montgomery, col_stride=col_stride) if (args.columns or args.montgomery) else probe
    row_bytes = 32 if args.columns else 160
    ns, xs, un, ux = synth_inputs(w, bits, lo, hi)
    n_dev, x_dev = chip.assign_integer(un), chip.assign_integer(ux)
    dev = "cuda:%d" % env.local_rank
    elem_bytes = 5 * col_stride if args.columns else rows * 160
    nimg = int(os.environ.get("H2R_BENCH_ADV_DEPTH", "2"))   # image / workspace sets the calls rotate through (developer: 2..4)
    wss = [torch.zeros(chip.workspace_bytes(chunk, pl.num_mul_mods), dtype=torch.uint8, device=dev) for _ in range(nimg)]
    # Placement: like the record kernel's trace regions an image buffer has a store rate of its own, stable for
    # the life of the allocation -- 5.1-5.4, 6.1-6.3, 6.6-6.8 or 7.0-7.2 TB/s for the same launch, by buffer (profiles/r04_cells_placement.txt;
    # the cause is not known).  A prover allocates its image buffers once, so the bench looks: `cand` allocations (16, or what three quarters of the free memory hold), the cells kernel
    # timed on each (all held during the look so that they are different memory), the fastest two kept.  --placement-candidates 0: as allocated.
    cand = args.placement_candidates if args.placement_candidates >= 0 else 16
    free_b = torch.cuda.mem_get_info(env.local_rank)[0]
    cand = max(0, min(cand, int(free_b * 0.75) // (chunk * elem_bytes)))
    placement = "as allocated"
    if cand > nimg:
        first = chip.pow_mod_fixed_exp(x_dev, e, n_dev, want_trace=False, check_in_field=True, workspace=wss[0])
        torch.cuda.synchronize()
        pool, ms = [], []
        for _ in range(cand):
            buf = torch.empty(chunk * elem_bytes, dtype=torch.uint8, device=dev)
            first.emit_modpow_advice(out=buf)            # (pages touched, code loaded)
            ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ta.record()
            for _ in range(2):
                _lib.check(L.h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pl), n_dev.data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0, wss[0].data_ptr(),
                                                       chunk, first.status.data_ptr(), buf.data_ptr() + int(sec[0]) * row_bytes, elem_bytes, chip._stream()),
                           "h2r_pow_trace_emit_advice")
            tb.record()
            torch.cuda.synchronize()
            pool.append(buf)
            ms.append(ta.elapsed_time(tb) / 2)
        order = sorted(range(cand), key=lambda i: ms[i])
        images = [pool[i] for i in order[:nimg]]
        placement = {"candidates": cand, "cells_kernel_alone_ms_per_candidate": [round(t, 4) for t in ms], "kept_ms": [round(ms[i], 4) for i in order[:nimg]]}
        pair_pool = pool if (nimg == 2 and os.environ.get("H2R_BENCH_ADV_PAIR_LOOK", "1") != "0") else None   # (looked at again below, under the pipelined call)
        del buf, first
        if pair_pool is None:
            del pool
            torch.cuda.empty_cache()
    else:
        images = [torch.zeros(chunk * elem_bytes, dtype=torch.uint8, device=dev) for _ in range(nimg)]
        pair_pool = None
    ifs = vl.elem_stride if whole else chip.in_field_layout()[0]
    ifb = [torch.zeros(chunk * ifs, dtype=torch.uint8, device=dev) for _ in range(nimg)]   # (--verify: the in-field + encoded-message witness)
    if whole:
        hrng = random.Random(0x68327273 + 23)
        hashed_dev = torch.tensor([[hrng.getrandbits(63) for _ in range(4)] for _ in range(chunk)], dtype=torch.int64, device=dev)   # synthetic digests
        valids = [torch.zeros(chunk, dtype=torch.uint8, device=dev) for _ in range(nimg)]
    outs = [torch.zeros((chunk, chip.num_limbs), dtype=chip.torch_dtype, device=dev) for _ in range(nimg)]
    sts = [torch.zeros(chunk, dtype=torch.uint8, device=dev) for _ in range(nimg)]
    # The chain stream has the higher priority: its short kernels get the CUs a retiring cells wave frees (same-box A/B,
    # profiles/r04_advice_ab.txt: 1.955 against 1.989 ms per step; everything on one stream: 2.146).  Developer A/B:
    # H2R_BENCH_ADV=noprio | serial.
    adv_mode = os.environ.get("H2R_BENCH_ADV", "")
    # default: ONE export per call, h2r_pipeline_modpow_public_key_advice (the pipeline owns the side streams and the events); the
    # developer modes streams | noprio | serial compose the same thing here from plain exports, two torch streams and events
    if whole:
        adv_mode = ""
    pipe = H.Pipeline(chip, nimg, 2) if adv_mode == "" else None
    s_chain = torch.cuda.Stream() if adv_mode == "noprio" else torch.cuda.Stream(priority=-1)
    s_cells = s_chain if adv_mode == "serial" else torch.cuda.Stream()
    chain_done = [torch.cuda.Event() for _ in range(nimg)]
    cells_done = [torch.cuda.Event() for _ in range(nimg)]
    issued = [0]
    results = [None] * nimg

    class _Res:     # what the post-run check reads of a call
        pass

    def step():
        k = issued[0] % nimg
        if pipe is not None:
            with torch.cuda.stream(s_chain):
                if whole:
                    pipe.verify_pkcs1v15_advice(x_dev, e, n_dev, hashed_dev, ifb[k], wss[k], outs[k], valids[k], sts[k], images[k].view(chunk, elem_bytes))
                else:
                    pipe.modpow_public_key_advice(x_dev, e, n_dev, wss[k], outs[k], sts[k], ifb[k], images[k].view(chunk, elem_bytes))
            r = _Res(); r.status = sts[k]; r.value = outs[k]
            results[k] = r
            issued[0] += 1
            return k
        with torch.cuda.stream(s_chain):
            if issued[0] >= nimg:
                s_chain.wait_event(cells_done[k])        # the workspace / witness of call k - 2 have been consumed
            results[k] = chip.pow_mod_fixed_exp(x_dev, e, n_dev, want_trace=False, check_in_field=True, workspace=wss[k], out=outs[k],
                                                status=sts[k], in_field_buf=ifb[k])
            chain_done[k].record(s_chain)                # (the cells kernel needs the operands only)
            # the assert_in_field rows of the element (2 % of its bytes) on this stream too: next to the previous call's cells kernel
            _lib.check(L.h2r_fresh_op_emit_advice(chip._ctx, _lib.FRESH_OPS.index("is_in_field"), _lib.H2R_ADVICE_ASSERT_ONE, x_dev.data_ptr(),
                                                  n_dev.data_ptr(), None, ifb[k].data_ptr(), 0, 0, chunk, sts[k].data_ptr(), images[k].data_ptr(),
                                                  elem_bytes, chip._stream()), "h2r_fresh_op_emit_advice")
        with torch.cuda.stream(s_cells):
            s_cells.wait_event(chain_done[k])
            # the pow rows, written directly from the call's operands (h2r_modpow_public_key_emit_advice = the two exports in a row)
            _lib.check(L.h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pl), n_dev.data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0, wss[k].data_ptr(),
                                                   chunk, sts[k].data_ptr(), images[k].data_ptr() + int(sec[0]) * row_bytes, elem_bytes, chip._stream()),
                       "h2r_pow_trace_emit_advice")
            cells_done[k].record(s_cells)
        issued[0] += 1
        return k

    step()
    torch.cuda.synchronize()
    if cand > nimg and pipe is not None and pair_pool is not None:
        # Two cells kernels are in flight at a time, one per image: what counts is how a PAIR of images takes them.  (The lookup columns showed it:
        # allocations fall into two placement classes and two concurrent store streams are fast when their buffers are of different classes,
        # profiles/r05_lookup_placement.txt.)  The fastest image alone stays; every other candidate is tried as its partner under the pipelined call.
        def pair_ms(other):
            images[1] = other
            for _ in range(2):
                step()
            with torch.cuda.stream(s_chain):
                pipe.join()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(6):
                step()
            with torch.cuda.stream(s_chain):
                pipe.join()
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / 6 * 1e3
        if issued[0] % 2:
            step()                                        # (an even number of calls so far: images[0] stays the first image of a pair)
        pms = {i: pair_ms(pair_pool[i]) for i in order[1:]}
        bi = min(pms, key=pms.get)
        images[1] = pair_pool[bi]
        placement["pair_look_ms_per_call"] = [round(pms[i], 4) for i in order[1:]]
        placement["kept_pair_ms_per_call"] = round(pms[bi], 4)
        placement["kept_ms"] = [round(ms[order[0]], 4), round(ms[bi], 4)]
        del pair_pool, pool
        torch.cuda.empty_cache()
    t_ramp = time.perf_counter()
    ramp = 0
    while ramp < args.clock_warmup_calls // 8 and time.perf_counter() - t_ramp < 0.5:   # (a call is 2 ms: a dozen keep the clocks up)
        step()
        ramp += 1
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(0 if args.no_kernel_timing else 8 * steps + 8)
    env.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    if pipe is not None:
        with torch.cuda.stream(s_chain):
            pipe.join()
    torch.cuda.synchronize()
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    cells_ms = _lib.profile_read(_lib.KERNEL_CELLS)
    chain_ms = _lib.profile_read(_lib.KERNEL_CHAIN)
    emit_ms = _lib.profile_read(_lib.KERNEL_EMIT)
    _lib.profile_enable(0)
    # what was timed is the real thing: results against pow(), and the timed image against the image of a call WITH records
    assert int(sts[last].max().item()) == 0 or (w, bits) != (64, 2048)
    got = H.AssignedInteger(outs[last].contiguous(), w).to_big_uint()
    for i in (0, 1, 2, chunk - 1):
        if i < chunk and xs[i] < ns[i]:
            assert got[i] == pow(xs[i], e, ns[i]), "GPU result differs from pow(x, e, n)"
    sample = min(chunk, 8)
    if whole:   # the record-based whole element (h2r_verify_pkcs1v15_batch + h2r_verify_emit_advice) of the first signatures
        rsa = H.RSAChip(bits, 5, device=env.local_rank, columns=args.columns, montgomery=args.montgomery, col_stride=col_stride) if (args.columns or args.montgomery) \
            else H.RSAChip(bits, 5, device=env.local_rank)
        pk = H.RSAPublicKey(H.AssignedInteger(n_dev.limbs_dev[:sample].contiguous(), w), H.Fix(e))
        ref = rsa.verify_pkcs1v15_signature(pk, hashed_dev[:sample].contiguous(), H.RSASignature(H.AssignedInteger(x_dev.limbs_dev[:sample].contiguous(), w)))
        ref_img = ref.emit_advice()
        assert torch.equal(ref.is_valid, valids[last][:sample]), "is_valid of the timed call differs from the record-based call's"
    else:
        ref = chip.pow_mod_fixed_exp(H.AssignedInteger(x_dev.limbs_dev[:sample].contiguous(), w), e, H.AssignedInteger(n_dev.limbs_dev[:sample].contiguous(), w),
                                     check_in_field=True)
        ref_img = ref.emit_modpow_advice(direct=False)
    torch.cuda.synchronize()
    timed = images[last].view(chunk, elem_bytes)[:sample]
    if args.columns:   # (the columns are longer than the element's rows: compare what the image covers)
        timed = timed.view(sample, 5, col_stride)[:, :, :rows * 32].reshape(sample, -1)
        ref_img = ref_img.view(sample, 5, col_stride)[:, :, :rows * 32].reshape(sample, -1)
    ok = ref.status.cpu().numpy() == 0
    assert torch.equal(timed[torch.from_numpy(ok).to(timed.device)], ref_img[torch.from_numpy(ok).to(ref_img.device)]), \
        "the timed advice image differs from the image of the records"
    # ... and the WHOLE last timed image through the device-side MockProver (h2r_advice_check): the main-gate equation of every row, the
    # range-check table of every lookup-enabled cell, every copy pair of the pow rows -- 1,024 x 77,021 rows where they lie
    import numpy as np
    if whole:
        kinds_all = np.zeros(rows, dtype=np.uint8)
        _lib.check(L.h2r_verify_row_kinds(chip._ctx, ctypes.byref(vl), kinds_all.ctypes.data), "h2r_verify_row_kinds")
    else:
        k_if = chip.fresh_op_row_kinds(_lib.FRESH_OPS.index("is_in_field"), assert_one=True)
        k_pow = np.zeros(pow_rows, dtype=np.uint8)
        _lib.check(L.h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), k_pow.ctypes.data), "h2r_pow_row_kinds")
        kinds_all = np.concatenate([k_if, k_pow])
    t_chk = time.perf_counter()
    bad, first = chip.advice_check(kinds_all, images[last].view(chunk, elem_bytes), chunk, status=sts[last],
                                   copies=chip.pow_copy_map(pl, e, row_offset=pow_off), src_a=x_dev, src_n=n_dev,
                                   lookup=H.LookupArgument(chip, rsa_chip=whole))
    n_bad = int((bad != 0).sum().item())
    assert n_bad == 0, "h2r_advice_check: %d elements of the timed image violate a gate / lookup / copy (first: %#x)" % (n_bad, int(first[bad != 0][0].item()))
    audit_s = time.perf_counter() - t_chk
    # ... and the lookup multiplicities counted from the timed image (h2r_lookup_hist_advice: what the lookup argument needs of a witness that has
    # no records) against those of the first signatures' records
    la = H.LookupArgument(chip, rsa_chip=whole)
    if whole:
        want_h = la.hist_verify(ref, la.new_hist(sample))
    else:
        want_h = la.hist_records(ref.trace, la.new_hist(sample), status=ref.status)
        la.hist_fresh_op("is_in_field", ref.in_field.buf, ref.in_field.elem_stride, sample, want_h)
    got_h = la.hist_advice(kinds_all, images[last].view(chunk, elem_bytes)[:sample], sample, la.new_hist(sample), status=sts[last][:sample].contiguous())
    torch.cuda.synchronize()
    okh = torch.from_numpy(ok).to(got_h.device)
    assert torch.equal(got_h[okh], want_h[okh]), "the lookup multiplicities of the timed image differ from those of the records"
    if env.rank == 0:
        algo = chunk * pow_rows * 160
        stamped_s = (sum(cells_ms) / len(cells_ms)) / 1e3 if cells_ms else float("nan")
        # The pipelined export keeps TWO cells launches in flight (the pipeline's two side streams: call k + 1's starts as soon as its chains
        # are done and fills what call k's tail leaves), so a launch's own duration counts the time it shares: the roofline takes the
        # launches' PERIOD -- the timed region's wall time per launch, an upper bound of what one launch costs -- instead.
        overlapped = pipe is not None
        avg_s = (dt / steps) if overlapped else stamped_s
        achieved = algo / avg_s / 1e9 if cells_ms else None
        line = {
            "metric": "RSA-2048 pkcs1v15 witness assigns/sec" if bits == 2048 else "RSA-%d witness assigns/sec" % bits,
            "value": round(global_batch * steps / dt, 1), "unit": "assigns/s", "n_gpus": env.world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * dt / steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u%d" % w, "data": "synthetic",
            "config": {"workload": ("%s batch=%d per GPU, %d-bit limbs, advice image of the whole verify_pkcs1v15_signature element (%d rows = %d B/assign: 1 is_eq row + %d "
                                    "assert_in_field rows + %d pow rows + %d encoded-message rows)" % (args.workload, chunk, w, rows, rows * 160, int(sec4[1]), pow_rows, int(sec4[3])))
                                   if whole else
                                   "%s batch=%d per GPU, %d-bit limbs, advice image (%d rows = %d B/assign: %d assert_in_field rows + %d pow rows)" %
                                   (args.workload, chunk, w, rows, rows * 160, int(sec[0]), pow_rows),
                       "path": "advice image (verify element)" if whole else "advice image",
                       "representation": {"columns": bool(args.columns), "montgomery": bool(args.montgomery), "col_stride": col_stride,
                                          "note": "planar: one contiguous vector per advice column; montgomery: cells = x * 2^256 mod p (the in-memory form of a halo2 field element)"},
                       "per_gpu_batch": chunk, "global_batch": global_batch, "calls_per_step": 1, "signatures_per_call": chunk,
                       "mul_mods_per_assign": pl.num_mul_mods, "parallelism": "signature-sharded x%d" % env.world, "ranks": env.world,
                       "pipeline": ("h2r_pipeline_verify_pkcs1v15_advice: chains, in-field / EM witness and the three short row programs of call k+1 on the caller's stream next to "
                                    "cells_kernel of call k on the pipeline's side stream (2 workspaces, 2 images; no records)") if whole else
                                   ("h2r_pipeline_modpow_public_key_advice: chain kernels of call k+1 on the caller's stream next to cells_kernel of call k on the "
                                    "pipeline's side stream (2 workspaces, 2 images)") if pipe is not None else
                                   "chain kernels of call k+1 on a second stream next to cells_kernel of call k (2 workspaces, 2 images), stream-ordered exports + events (H2R_BENCH_ADV=%s)" % adv_mode,
                       "untimed_clock_warmup_calls": ramp, "warmup_calls_total": 1 + ramp + warmup, "buffer_placement": placement,
                       "post_run_audit": "h2r_advice_check on the last timed image: %d elements x %d rows, gate + lookup + %s copy pairs per element, 0 violations (%.2f s incl. the copy map); "
                                         "h2r_lookup_hist_advice of its first %d elements = the multiplicities of their records" % (chunk, rows, "pow-row", audit_s, sample)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": None,
                         "traffic_source": "not measured", "kernel": "cells_kernel<%d%s>" % (w, ", Montgomery" if args.montgomery else ""), "launches_timed": len(cells_ms),
                         "signatures_per_launch": chunk, "avg_launch_ms": round(1e3 * avg_s, 4) if cells_ms else None,
                         "timing": ("avg_launch_ms = the launches' period (timed wall time / launches): two cells launches are in flight at a time; "
                                    "avg_launch_ms_in_flight = a launch's own start-to-end time (HIP events stamped by the dispatch)") if overlapped else
                                   "per-launch HIP events stamped by the dispatch packets",
                         "avg_launch_ms_in_flight": round(1e3 * stamped_s, 4) if (overlapped and cells_ms) else None,
                         "algorithmic_bytes_per_launch": algo,
                         "chain_kernel_avg_ms": round(sum(chain_ms) / len(chain_ms), 4) if chain_ms else None,
                         "in_field_rows_kernel_avg_ms": round(sum(emit_ms) / len(emit_ms), 4) if emit_ms else None},
            "whole_path_hbm_frac": round(global_batch * steps / dt * rows * 160 / (env.world * HBM_PEAK_GBS * 1e9), 4),
        }
        if env.world == 1 and args.pmc_traffic == "auto" and cells_ms:
            hbm, how = measured_pmc_traffic(["--advice", "--workload", args.workload, "--batch", str(chunk)] + (["--columns"] if args.columns else []) +
                                            (["--montgomery"] if args.montgomery else []) + (["--verify"] if whole else []), "cells_kernel")   # (the passes run with --placement-candidates 0)
            if hbm is not None:
                line["roofline"]["traffic"] = hbm
                line["roofline"]["traffic_source"] = how
            else:
                line["roofline"]["traffic_source"] = "live measurement skipped: " + how
        if env.world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, bits, e, un, ux)
        print(json.dumps(line))
    env.finalize()


def lookup_bench(args):
    """--lookup: the lookup ARGUMENT of the range-check batch (what `range_chip.load_table` + `create_proof` do with the sub-limb cells,
    reference benches/bench.rs:141-142, 321-329): one step = h2r_lookup_permuted_columns for 256 circuits (RSA-2048 modpow_public_key
    records, k = 17: 131,066 usable rows) x 5 arguments -- halo2's A' / S' columns, 10.7 GB written per call by lookup_fill_kernel
    behind the set-up kernel that ranks the table.  The output pair lies in regions of the image arena (h2r_image_arena_create) unless
    --placement-candidates 0.  roofline: lookup_fill_kernel, HBM-write bound, algorithmic bytes = the two columns."""
    import ctypes
    torch.cuda.set_device(0)
    w, bits, e = WORKLOADS["rsa2048_e65537"]
    B = args.batch if args.batch else 256
    chip = H.BigIntChip(w, bits, device=0, montgomery=args.montgomery)
    ns, xs, un, ux = synth_inputs(w, bits, 0, B)
    res = chip.pow_mod_fixed_exp(chip.assign_integer(ux), e, chip.assign_integer(un))
    la = H.LookupArgument(chip)
    usable = (1 << 17) - 6
    hist = la.new_hist(B)
    la.hist_records(res.trace, hist)
    torch.cuda.synchronize()
    del res
    torch.cuda.empty_cache()
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = random.Random(0x68327273 + 11)
    thetas = [rng.randrange(P) for _ in range(B)]
    col_bytes = B * 5 * usable * 32
    cand = args.placement_candidates if args.placement_candidates >= 0 else 16   # (eight pairs: with four, a box now and then offers no fast one)
    arena, placement = None, "as allocated"
    # roles (default): every candidate as S' next to one fixed A', then every other one as A' next to the best S' -- what the data say is that
    # allocations fall into TWO classes and the call is fast exactly when its two columns lie in DIFFERENT classes (profiles/r05_lookup_placement.txt);
    # call: adjacent pairs timed with the call; arena: h2r_image_arena_create's streaming fill (does not predict this call)
    look = os.environ.get("H2R_BENCH_LOOKUP_LOOK", "roles")
    if cand >= 4 and look == "call":
        # The store rate of a buffer depends on the buffer AND on the kernel that writes it (profiles/r05_lookup_placement.txt: the image arena's
        # streaming fill ranks candidates 0.78-0.93 ms, and two boxes whose kept pair measured 0.78 ms ran this call at 0.69 and 0.84 of the
        # peak): the candidates are looked at with the call that will write them -- pairs (A', S') of plain allocations, the fastest pair kept.
        pool = [torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda") for _ in range(cand - cand % 2)]
        ms = []
        for i in range(0, len(pool), 2):   # (ranking single candidates next to one fixed partner does not predict a pair: 0.61-0.78 for pairs ranked 1.63-1.66 ms)
            la.permuted_columns(hist, thetas, usable, out=(pool[i], pool[i + 1]))       # (pages touched)
            ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ta.record()
            for _ in range(2):
                la.permuted_columns(hist, thetas, usable, out=(pool[i], pool[i + 1]))
            tb.record()
            torch.cuda.synchronize()
            ms.append(ta.elapsed_time(tb) / 2)
        best = min(range(len(ms)), key=lambda i: ms[i])
        out = (pool[2 * best], pool[2 * best + 1])
        placement = {"candidates": len(pool), "look": "the call itself on pairs (A', S') of plain allocations, the fastest pair kept",
                     "call_ms_per_pair": [round(t, 4) for t in ms], "kept_ms": round(ms[best], 4)}
        del pool
        torch.cuda.empty_cache()
    elif cand >= 4 and look == "roles":
        pool = [torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda") for _ in range(cand)]

        def t_pair(ia, is_):
            la.permuted_columns(hist, thetas, usable, out=(pool[ia], pool[is_]))
            ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ta.record()
            for _ in range(2):
                la.permuted_columns(hist, thetas, usable, out=(pool[ia], pool[is_]))
            tb.record()
            torch.cuda.synchronize()
            return ta.elapsed_time(tb) / 2
        ms_s = {i: t_pair(0, i) for i in range(1, cand)}
        bs = min(ms_s, key=ms_s.get)
        ms_a = {j: t_pair(j, bs) for j in range(cand) if j != bs}
        ba = min(ms_a, key=ms_a.get)
        out = (pool[ba], pool[bs])
        placement = {"candidates": cand, "look": "roles: each candidate as S' next to one A', then each as A' next to the best S'",
                     "ms_as_S": [round(ms_s[i], 4) for i in sorted(ms_s)], "ms_as_A": [round(ms_a[j], 4) for j in sorted(ms_a)], "kept_ms": round(ms_a[ba], 4)}
        del pool
        torch.cuda.empty_cache()
    elif cand >= 2:
        arena = H.TraceArena.for_images(chip, col_bytes, regions=2, candidates=cand)
        out = tuple(r.view(B, 5, usable, 32) for r in arena.regions)
        placement = {"candidates": len(arena.measurements_ms), "fill_ms_per_candidate": [round(t, 4) for t in arena.measurements_ms],
                     "kept_ms": [round(t, 4) for t in arena.region_ms]}
    else:
        out = (torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda"), torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda"))
    for _ in range(max(1, args.warmup)):
        la.permuted_columns(hist, thetas, usable, out=out)
    torch.cuda.synchronize()
    steps = args.steps
    _lib.profile_enable(steps + 4)
    t0 = time.perf_counter()
    for _ in range(steps):
        a_perm, s_perm, status = la.permuted_columns(hist, thetas, usable, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fill = _lib.profile_read(_lib.KERNEL_LOOKUP)
    _lib.profile_enable(0)
    assert not status.cpu().numpy().any()
    # what was timed is the real thing: every column sorted, and a multiset-preserving permutation of (inputs, table) for sampled circuits
    av = a_perm[0, 0].view(torch.int64).view(usable, 4)
    assert bool((av[1:, 3] >= av[:-1, 3]).all().item()) or args.montgomery, "A' is not sorted"
    del av, a_perm, s_perm, out          # (views of the arena's mapped memory: gone before the arena unmaps it)
    torch.cuda.synchronize()
    if arena is not None:
        arena.close()
    algo = 2 * col_bytes
    fill_s = (sum(fill) / len(fill)) / 1e3
    line = {"metric": "lookup argument: permuted columns A', S' written", "value": round(algo * steps / dt / 1e9, 1), "unit": "GB/s", "n_gpus": 1,
            "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": round(1e3 * dt / steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "h2r_lookup_permuted_columns: %d RSA-2048 modpow_public_key circuits x 5 arguments x (A', S') x %d usable rows (k = 17), %d table rows"
                                   % (B, usable, la.n_rows), "montgomery": bool(args.montgomery), "buffer_placement": placement},
            "roofline": {"bound": "hbm", "achieved": round(algo / fill_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo / fill_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": "not measured",
                         "kernel": "lookup_fill_kernel", "launches_timed": len(fill), "avg_launch_ms": round(1e3 * fill_s, 4),
                         "algorithmic_bytes_per_launch": algo},
            "whole_call_hbm_frac": round(algo * steps / dt / 1e9 / HBM_PEAK_GBS, 4)}
    if args.pmc_traffic == "auto":
        hbm, how = measured_pmc_traffic(["--lookup", "--batch", str(B)] + (["--montgomery"] if args.montgomery else []), "lookup_fill_kernel")
        if hbm is not None:
            line["roofline"]["traffic"] = hbm
            line["roofline"]["traffic_source"] = how
        else:
            line["roofline"]["traffic_source"] = "live measurement skipped: " + how
    print(json.dumps(line))


def flow_bench(args):
    """--records-free-flow: everything a column-taking prover needs of the witness of `batch` one-signature circuits (k = 17), no record written:
    the whole verify_pkcs1v15_signature element image (h2r_pipeline_verify_pkcs1v15_advice), its lookup multiplicities from the image
    (h2r_lookup_hist_advice + h2r_lookup_hist_values for assign_integer(sig), assign_integer(n)) and halo2's A', S' of the five lookup arguments
    (h2r_lookup_permuted_columns).  The image call of batch k + 1 is issued in front of the multiplicities and columns of batch k."""
    import ctypes
    torch.cuda.set_device(0)
    w, bits, e = WORKLOADS["rsa2048_e65537"]
    B = args.batch if args.batch else 1024
    usable = (1 << 17) - 6
    rows0 = 77219
    kw = dict(columns=True, montgomery=True, col_stride=((rows0 * 32 + 4095) // 4096) * 4096) if (args.columns or args.montgomery) else {}
    rsa = H.RSAChip(bits, 5, **kw)
    chip = rsa.bigint_chip()
    la = H.LookupArgument(chip, rsa_chip=True)
    ns, xs, un, ux = synth_inputs(w, bits, 0, B)
    n, sig = chip.assign_integer(un), chip.assign_integer(ux)
    rng = random.Random(0x68327273 + 31)
    hashed = torch.tensor([[rng.getrandbits(63) for _ in range(4)] for _ in range(B)], dtype=torch.int64, device="cuda")
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    thetas = [rng.randrange(P) for _ in range(B)]
    pipe = H.Pipeline(chip, 2, 2)
    vl = pipe.verify_compact_layout(e)
    L = _lib.lib()
    sec = (ctypes.c_uint64 * 4)()
    rows = int(L.h2r_verify_advice_rows(chip._ctx, ctypes.byref(vl), sec))
    kinds = np.zeros(rows, dtype=np.uint8)
    _lib.check(L.h2r_verify_row_kinds(chip._ctx, ctypes.byref(vl), kinds.ctypes.data), "h2r_verify_row_kinds")
    kd = torch.from_numpy(kinds).cuda()
    eb = chip.image_bytes(rows)
    sets = [dict(img=torch.empty((B, eb), dtype=torch.uint8, device="cuda"), wit=torch.zeros((B, vl.elem_stride), dtype=torch.uint8, device="cuda"),
                 ws=torch.empty(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 powed=torch.zeros((B, chip.num_limbs), dtype=torch.int64, device="cuda"), valid=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                 st=torch.zeros(B, dtype=torch.uint8, device="cuda"), hist=la.new_hist(B)) for _ in range(2)]
    col_b1 = B * 5 * usable * 32
    n_cand = int(min(6, (torch.cuda.mem_get_info(0)[0] * 0.8) // col_b1)) if args.placement_candidates != 0 else 2
    pool = [torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda") for _ in range(max(2, n_cand))]
    cols = (pool[0], pool[1])
    state = {}

    def step(k):
        s = sets[k & 1]
        pipe.verify_pkcs1v15_advice(sig, e, n, hashed, s["wit"], s["ws"], s["powed"], s["valid"], s["st"], s["img"])
        if k:
            p = sets[(k - 1) & 1]
            p["hist"].zero_()
            la.hist_advice(kd, p["img"], B, p["hist"], status=p["st"])
            la.hist_values(sig.limbs_dev, 64, 8, p["hist"])
            la.hist_values(n.limbs_dev, 64, 8, p["hist"])
            state["out"] = la.permuted_columns(p["hist"], thetas, usable, out=cols)

    for k in range(max(2, args.warmup)):
        step(k)
    pipe.join()
    torch.cuda.synchronize()
    placement = "as allocated (no look)"
    if len(pool) > 2:
        # the two columns in buffers of different placement classes (profiles/r05_lookup_placement.txt): every candidate as S' next to pool[0], then
        # every other one as A' next to the best S', timed with the call on the multiplicities of the warm-up batch
        hp = sets[(max(2, args.warmup) - 2) & 1]["hist"]

        def t_pair(ia, is_):
            la.permuted_columns(hp, thetas, usable, out=(pool[ia], pool[is_]))
            ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ta.record()
            la.permuted_columns(hp, thetas, usable, out=(pool[ia], pool[is_]))
            tb.record()
            torch.cuda.synchronize()
            return ta.elapsed_time(tb)
        ms_s = {i: t_pair(0, i) for i in range(1, len(pool))}
        bs = min(ms_s, key=ms_s.get)
        ms_a = {j: t_pair(j, bs) for j in range(len(pool)) if j != bs}
        ba = min(ms_a, key=ms_a.get)
        cols = (pool[ba], pool[bs])
        placement = {"lookup_column_candidates": len(pool), "look": "roles (as bench.py --lookup)", "ms_as_S": [round(ms_s[i], 3) for i in sorted(ms_s)],
                     "ms_as_A": [round(ms_a[j], 3) for j in sorted(ms_a)], "kept_ms": round(ms_a[ba], 3)}
        pool = None
        torch.cuda.empty_cache()
    k0, steps = max(2, args.warmup), args.steps
    for k in range(k0, k0 + 2):        # (the pipeline again in step with the loop below: two untimed batches)
        step(k)
    k0 += 2
    t0 = time.perf_counter()
    for k in range(k0, k0 + steps):
        step(k)
    pipe.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    a_perm, s_perm, status = state["out"]
    assert not status.cpu().numpy().any()
    # what was timed is the real thing: results against pow(), the multiplicities of the first circuits against the record-based call's
    last = sets[(k0 + steps - 2) & 1]
    got = H.AssignedInteger(last["powed"].contiguous(), w).to_big_uint()
    for i in (0, 1, B - 1):
        if xs[i] < ns[i]:
            assert got[i] == pow(xs[i], e, ns[i]), "GPU result differs from pow(x, e, n)"
    sample = min(B, 4)
    ref = rsa.verify_pkcs1v15_signature(H.RSAPublicKey(H.AssignedInteger(n.limbs_dev[:sample].contiguous(), w), H.Fix(e)), hashed[:sample].contiguous(),
                                        H.RSASignature(H.AssignedInteger(sig.limbs_dev[:sample].contiguous(), w)))
    want = la.hist_verify(ref, la.new_hist(sample))
    la.hist_values(sig.limbs_dev[:sample].contiguous(), 64, 8, want)
    la.hist_values(n.limbs_dev[:sample].contiguous(), 64, 8, want)
    torch.cuda.synchronize()
    assert torch.equal(last["hist"][:sample], want), "multiplicities counted from the image differ from the records'"
    img_b, col_b = B * rows * 160, 2 * B * 5 * usable * 32
    line = {"metric": "records-free prover inputs: circuits/sec (verify element image + lookup multiplicities + A', S')", "value": round(B / dt, 1), "unit": "circuits/s",
            "n_gpus": 1, "steps": steps, "warmup": max(2, args.warmup), "ms_per_step": round(1e3 * dt, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "rsa2048_e65537, %d one-signature circuits per batch (k = 17): %d-row verify element image + 5 x (A', S') x %d usable rows" % (B, rows, usable),
                       "representation": {"columns": bool(kw), "montgomery": bool(kw)}, "bytes_per_batch": {"advice_image": img_b, "lookup_columns": col_b},
                       "placement": placement},
            "roofline": {"bound": "hbm", "achieved": round((img_b + col_b) / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((img_b + col_b) / dt / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "kernel": "lookup_fill_kernel + cells_kernel (whole flow)", "algorithmic_bytes_per_launch": img_b + col_b}}
    print(json.dumps(line))


def sub_run(extra, timeout=200):
    """`bench.py <extra>` in a fresh process (same steps cap, no CPU baseline, no counters, no nested sub-runs): the parsed line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--pmc-traffic", "off", "--scale-anchor", "off", "--sub-runs", "off"] + extra
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, check=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        d["_wall_s"] = round(time.perf_counter() - t0, 1)
        return d
    except subprocess.CalledProcessError as ex:
        return {"error": ("rc %d: " % ex.returncode) + (ex.stderr or "").strip()[-300:]}
    except Exception as ex:
        return {"error": str(ex)[:200]}


def sub_run_lines(args):
    """The numbers a driver that runs only the default line cannot see, each from a fresh process of this script: the headline as
    allocated (no placement look), the advice image (default and prover representation), BASELINE configs 4 and 5, the lookup argument."""
    st = ["--steps", str(args.steps), "--warmup", str(args.warmup)]
    out = {}
    d = sub_run(st + ["--placement-candidates", "0"])
    out["plain_allocations"] = {"what": "this line's workload with every buffer as hipMalloc hands it out (no arena, no candidates)", "value": d.get("value"),
                                "frac": d.get("roofline", {}).get("frac"), "whole_path_hbm_frac": d.get("whole_path_hbm_frac"), "wall_s": d.get("_wall_s"), "error": d.get("error")}
    for key, extra in (("advice", []), ("advice_columns_montgomery", ["--columns", "--montgomery"]), ("advice_verify_element", ["--verify"])):
        d = sub_run(st + ["--advice"] + extra)     # (the default look: 16 candidate images; with 8 the kept pair is 0.2 ms slower on most boxes)
        what = {"advice": "row-major canonical", "advice_columns_montgomery": "planar Montgomery-form",
                "advice_verify_element": "whole verify_pkcs1v15_signature element's (is_eq + assert_in_field + pow + encoded-message rows) row-major canonical"}[key]
        out[key] = {"what": "bench.py --advice %s: elements/s of the %s advice image (12.3 MB each), cells_kernel" % (" ".join(extra), what),
                    "value": d.get("value"), "frac": d.get("roofline", {}).get("frac"), "whole_path_hbm_frac": d.get("whole_path_hbm_frac"),
                    "audit": d.get("config", {}).get("post_run_audit"), "wall_s": d.get("_wall_s"), "error": d.get("error")}
    oc = {}
    for key, extra in (("C4", ["--workload", "rsa4096_w32_e65537", "--batch", "4096", "--steps", "8", "--warmup", "2"]),
                       ("C5", ["--workload", "rsa2048_e2048bit", "--batch", "256", "--steps", "8", "--warmup", "2"]),
                       # [r6] RSA-1024: the key size of the reference's only enabled bench (benches/bench.rs:393-407); one-wave chains (h2r_chain_wave.hpp)
                       ("rsa1024_2048_per_call", ["--workload", "rsa1024_e65537", "--batch", "2048", "--steps", "20", "--warmup", "5"])):
        d = sub_run(extra)
        oc[key] = {"workload": " ".join(extra), "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"),
                   "frac": d.get("roofline", {}).get("frac"), "whole_path_hbm_frac": d.get("whole_path_hbm_frac"), "wall_s": d.get("_wall_s"), "error": d.get("error")}
    out["other_configs"] = oc
    d = sub_run(["--records-free-flow", "--steps", "6", "--warmup", "3"])
    out["records_free_flow"] = {"what": "bench.py --records-free-flow: per batch of 1,024 one-signature circuits the verify element image (no records), its lookup multiplicities "
                                        "counted from the image and A', S' of the five lookup arguments (55.6 GB per batch, plain allocations)", "value": d.get("value"), "unit": d.get("unit"),
                                "ms_per_batch": d.get("ms_per_step"), "GBps": d.get("roofline", {}).get("achieved"), "frac": d.get("roofline", {}).get("frac"), "wall_s": d.get("_wall_s"),
                                "error": d.get("error")}
    d = sub_run(["--lookup", "--steps", "8", "--warmup", "2"])
    out["lookup"] = {"what": "bench.py --lookup: h2r_lookup_permuted_columns, 256 circuits x 5 arguments (10.7 GB per call)", "whole_call_GBps": d.get("value"),
                     "whole_call_frac": d.get("whole_call_hbm_frac"), "fill_kernel_frac": d.get("roofline", {}).get("frac"),
                     "placement": d.get("config", {}).get("buffer_placement"), "wall_s": d.get("_wall_s"), "error": d.get("error")}
    return out


def main():
    # the image exports NCCL_DEBUG=VERSION, which makes RCCL print a banner on stdout; rank 0's stdout is ONE JSON line
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ.pop("NCCL_DEBUG")
    ensure_built()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="signatures per call (default 1024)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="calls per step = chunks of the GPU's shard (default 1 at --gpus 1: 1,024 signatures per GPU; 4 at --gpus > 1: 8,192 per GPU)")
    ap.add_argument("--workload", default="rsa2048_e65537", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline-depth", type=int, default=0, help="buffer sets the pipelined calls rotate through (2..4; default: 3 for the "
                                                                    "RSA-2048 and RSA-1024 shapes, 2 otherwise)")
    ap.add_argument("--side-streams", type=int, default=0,
                    help="streams the record kernels alternate between (default: 2 for the RSA-2048 and RSA-1024 shapes, whose pipelined calls (RSA-1024: of 1,536 .. 4,096) then go out as "
                         "chain kernels on the caller's stream and record kernels alternating between two side streams -- call k + 1's record "
                         "kernel starts while call k's tail drains; 1 otherwise: one launch per call, or one side stream)")
    ap.add_argument("--producers", type=int, default=1,
                    help="independent producers: P pipelines, each on a stream of its own with its own pipeline-depth buffer sets; the calls "
                         "alternate between them (one pipeline per producer thread is the C ABI's threading contract).  For latency-bound "
                         "shapes (BASELINE config 5: 256 chains of 3,072 dependent mul_mods) the chain kernels of two calls then run side by side")
    ap.add_argument("--verify", action="store_true",
                    help="time the whole verify_pkcs1v15_signature witness (in-field + modpow + encoded-message check) "
                         "instead of modpow_public_key alone (RSA-2048 workloads, pipelined mode)")
    ap.add_argument("--messages", type=int, default=0, metavar="LEN",
                    help="with --verify: start every call from LEN-byte message bytes (RSASignatureVerifier, src/lib.rs:183-246: "
                         "SHA-256 + hashed-message limbs on the device in the timed region) instead of precomputed digests")
    ap.add_argument("--advice", action="store_true",
                    help="time the prover-consumable witness: every step's output is the 5-column advice image of its modpow_public_key "
                         "elements (assert_in_field rows + pow rows written directly from the operands by cells_kernel), no record planes; "
                         "with --verify: of its whole verify_pkcs1v15_signature elements (h2r_pipeline_verify_pkcs1v15_advice)")
    ap.add_argument("--lookup", action="store_true", help="the lookup argument's permuted columns (h2r_lookup_permuted_columns) as the product")
    ap.add_argument("--records-free-flow", action="store_true",
                    help="the whole records-free flow per batch of one-signature circuits: verify element image + multiplicities from the image + A', S'")
    ap.add_argument("--sub-runs", choices=["auto", "off"], default="auto",
                    help="auto: the default N = 1 line also carries plain_allocations, advice, advice_columns_montgomery, other_configs (C4, C5, RSA-1024) and lookup, "
                         "each measured in a fresh process of this script")
    ap.add_argument("--columns", action="store_true", help="--advice: planar column vectors instead of 160-byte rows (H2R_ADVICE_COLUMNS)")
    ap.add_argument("--montgomery", action="store_true", help="--advice: cells in Montgomery form, x * 2^256 mod p (H2R_ADVICE_MONTGOMERY)")
    ap.add_argument("--shared-modulus", action="store_true",
                    help="one key, many signatures (H2R_F_SHARED_MODULUS): every element uses element 0's modulus")
    ap.add_argument("--per-launch-timing", action="store_true",
                    help="stamp every dispatch of the timed region with its own events (the round-3 way: costs 6-7 us per 1,024-signature step) "
                         "instead of two events over the region")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="developer: do not arm the C ABI's per-kernel event timing (roofline fields become null)")
    ap.add_argument("--user-stream", action="store_true",
                    help="developer: issue the calls on a created stream instead of torch's default (null) stream")
    ap.add_argument("--no-in-field", action="store_true",
                    help="developer: pow_mod_fixed_exp only (no assert_in_field witness kernel) in the pipelined call")
    ap.add_argument("--placement-candidates", type=int, default=-1,
                    help="trace regions the library's placement-aware arena (h2r_arena_create) maps and measures; the calls "
                         "rotate through the fastest ones (where a trace buffer lies physically decides whether the record "
                         "kernel writes it at 5.65 or up to 6.8 TB/s, DESIGN.md section 5).  Default 24 (8 for regions of more than 4 GB; "
                         "at most pipeline-depth + 1 regions are mapped at any time); 0 = plain allocations, taken as they come")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="fully stream-ordered calls (chain then trace per step) instead of the two-stream pipeline")
    ap.add_argument("--clock-warmup-calls", type=int, default=96,
                    help="untimed pipelined calls issued right before the W warm-up steps so that the timed steps do not sit in the "
                         "device's clock ramp after idle (0 = none; see profiles/r03_clock_ramp.txt)")
    ap.add_argument("--collectives", choices=["h2r", "torch"], default="h2r",
                    help="N > 1: the broadcast / result all-gather / timing barrier go through libh2r's RCCL exports (h2r_dist_*, default) "
                         "or through torch.distributed")
    ap.add_argument("--scale-anchor", choices=["auto", "off"], default="auto",
                    help="N = 1, default workload: also run the N > 1 per-GPU workload (8,192 signatures per step as four calls of 2,048) on this "
                         "GPU and report it as `scale_anchor`: the like-for-like origin of a 1 -> 8 scaling curve")
    ap.add_argument("--pmc-traffic", choices=["auto", "off"], default="auto",
                    help="roofline.traffic: auto = measure it now with two rocprofv3 --pmc passes of a short run of the same workload "
                         "(N = 1, rank 0; falls back to the committed profiles/pmc_traffic.json when rocprofv3 is unavailable); off = committed file only")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="the N > 1 plumbing without the GPU (gloo on CPU): see dist_dry_run; tests/test_dist_cpu.py runs it at N = 8")
    ap.add_argument("--dry-total", type=int, default=0, help="--dry-run-dist: the global batch (need not be a multiple of N)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    if args.dry_run_dist:
        return dist_dry_run(args)
    if args.records_free_flow:
        return flow_bench(args)
    if args.lookup:
        return lookup_bench(args)
    if args.advice:
        return advice_bench(args)

    env = DistEnv.from_environment(args.gpus, force=bool(os.environ.get("H2R_FORCE_DIST")))
    w, bits, e = WORKLOADS[args.workload]
    # RSA-2048 (32 x 64-bit limbs): three buffer sets and two record streams select the library's overlapped two-queue form (h2r.h,
    # h2r_pipeline_create_ex; same-box A/B against the one-launch step: tools/two_queue_ab.sh, profiles/r04_two_queue.txt)
    if args.pipeline_depth == 0:
        args.pipeline_depth = 3 if (w, bits) in ((64, 2048), (64, 1024)) else 2
    if args.side_streams == 0:
        args.side_streams = 2 if (w, bits) in ((64, 2048), (64, 1024)) else 1
    # developer: H2R_BENCH_ONE_GPU=1 runs every rank on GPU 0 over gloo -- the N > 1 code path (shards, gather, checks) on a
    # one-GPU box; not a measurement of anything
    one_gpu = bool(os.environ.get("H2R_BENCH_ONE_GPU"))
    if one_gpu:
        env.local_rank = 0
    torch.cuda.set_device(env.local_rank)
    # Collectives: world > 1 (or H2R_FORCE_DIST) goes through libh2r's own RCCL exports (h2r_dist_*: what a Rust host binds);
    # --collectives torch keeps torch.distributed ("nccl" = RCCL), which is also the automatic fall-back if the C-ABI
    # communicator cannot be created -- config.collective_backend says which one ran.
    use_h2r_dist = args.collectives == "h2r" and not one_gpu and (env.world > 1 or env.force)
    if not use_h2r_dist:
        env.init("gloo" if one_gpu else "nccl")
    if args.user_stream:
        torch.cuda.set_stream(torch.cuda.Stream())
    # Workload: N = 1 -> BASELINE configs[1] (one 1,024-signature call per step).  N > 1 -> configs[2]: every GPU owns a
    # contiguous shard of 8,192 signatures of the global batch (65,536 at N = 8); a step is one pass over the shard as FOUR
    # pipelined calls of 2,048 signatures, each into its own arena region (measured on one GPU: 5.30-5.32 M assigns/s; two calls
    # of 4,096: 5.19-5.44 M; one call of 8,192: 5.00 M; eight of 1,024: 4.5-4.7 M -- fewer kernel boundaries against regions
    # small enough for a wide candidate search).  --batch / --chunks override both.
    chunks = args.chunks if args.chunks else (1 if args.gpus == 1 else 4)
    chunk = args.batch if args.batch else (1024 if args.gpus == 1 else 8192 // chunks)
    if use_h2r_dist:
        from halo2_rsa_amd.dist import H2RDist, agree_all
        h2r_env = None
        try:
            h2r_env = H2RDist(H.BigIntChip(w, bits, device=env.local_rank), env.rank, env.world, env.local_rank)
        except Exception as ex:   # no librccl, communicator refused ...: torch.distributed instead
            sys.stderr.write("h2r_dist unavailable on rank %d (%s)\n" % (env.rank, str(ex)[:200]))
        # every rank takes the same backend: the C-ABI communicator only if it came up everywhere
        if agree_all("h2r_dist_up", h2r_env is not None, env.rank, env.world):
            env = h2r_env
        else:
            if h2r_env is not None:
                h2r_env._lib.h2r_dist_destroy(h2r_env._d)
            sys.stderr.write("rank %d: falling back to torch.distributed\n" % env.rank)
            env.init("nccl")
    # configuration broadcast (rank 0 decides): the only pre-run collective
    e, chunk, chunks, steps, warmup = env.broadcast_ints([e, chunk, chunks, args.steps, args.warmup])
    shard = chunk * chunks
    global_batch = shard * env.world
    lo, hi = shard_range(global_batch, env.rank, env.world)
    assert hi - lo == shard

    chip = H.BigIntChip(w, bits, device=env.local_rank)
    ns, xs, un, ux = synth_inputs(w, bits, lo, hi)
    if args.shared_modulus:   # SURVEY 8d's shared-n variant: x reduced modulo the one modulus
        ns = [ns[0]] * shard
        xs = [x % ns[0] for x in xs]
        # (H2R_BENCH_REPLICATE_N: developer check -- the one modulus handed over as a per-element array, no SHARED flag)
        un = H.UnassignedInteger(to_limbs(ns if os.environ.get("H2R_BENCH_REPLICATE_N") else ns[:1], w, bits))
        ux = H.UnassignedInteger(to_limbs(xs, w, bits))
    n_dev, x_dev = chip.assign_integer(un), chip.assign_integer(ux)
    pl = chip.pow_fixed_layout(e)
    dev = "cuda:%d" % env.local_rank
    producers = 1 if args.no_pipeline else max(1, args.producers)
    nbuf = 1 if args.no_pipeline else args.pipeline_depth * producers   # scratch sets the calls rotate through
    elem_stride = pl.elem_stride
    verify = (args.verify or args.messages > 0) and not args.no_pipeline and w == 64 and bits >= 1024   # --messages implies --verify; RSAChip::LIMB_WIDTH = 64
    if verify:   # whole verifier witness: the element also holds the in-field and encoded-message regions
        import ctypes
        vl = _lib.H2RVerifyLayout()
        eb = e.to_bytes((e.bit_length() + 7) // 8, "little")
        _lib.check(_lib.lib().h2r_verify_layout_fixed(chip._ctx, eb, len(eb), ctypes.byref(vl)), "h2r_verify_layout_fixed")
        elem_stride = vl.elem_stride
        hashed_dev = torch.randint(-2**62, 2**62, (shard, 4), dtype=torch.int64, device=dev)
        valid = torch.zeros(shard, dtype=torch.uint8, device=dev)
        if args.messages:   # the caller's step: message bytes -> SHA-256 -> hashed-message limbs, per call, on the device
            msgs_dev = torch.randint(0, 256, (shard, args.messages), dtype=torch.uint8, device=dev)
            digest_dev = torch.zeros((shard, 32), dtype=torch.uint8, device=dev)
            hm_dev = torch.zeros((shard, _lib.H2R_HASHED_MSG_STREAM_BYTES), dtype=torch.uint8, device=dev)
    # The shard's traces stay resident on the GPU that produced them: with one call per step the calls
    # rotate through `nbuf` trace regions, with several calls per step every call has its own region of the shard's
    # trace.  Zero-filled, so every page is resident before the first (possibly un-warmed) timed step touches it.
    regions = nbuf if chunks == 1 else chunks
    # Placement: the record kernel's store rate depends on where its output lies physically (per 1.25 GB region one of
    # ~5.65 / 5.8 / 6.35 / 6.8 TB/s, stable for the life of the allocation; tools/buffer_speed_probe.py).  A service allocates
    # its trace arena once, so it can afford to look: the library's arena (h2r_arena_create) maps `cand` candidate regions,
    # runs the record kernel on each and keeps the `nbuf` fastest -- untimed initialisation, reported in the JSON line.
    cand = args.placement_candidates
    if cand < 0:   # (the arena holds at most nbuf + 1 regions at any time: the look also fits the 44-50 GB traces of configs 4 and 5)
        cand = 24 if chunk * elem_stride <= (4 << 30) else 8
    if cand <= regions:
        cand = 0
    placement, arena = None, None
    if cand:
        off_rec = vl.pow.off_records if verify else pl.off_records
        try:
            # N > 1: every rank bounds what its look may hold besides the kept regions (a tenth of the device's free memory)
            look_cap = int(torch.cuda.mem_get_info(env.local_rank)[0] // 10) if env.world > 1 else 0
            arena = H.TraceArena(chip, elem_stride, off_rec, pl.num_mul_mods, chunk, regions=regions, candidates=cand, max_look_bytes=look_cap)
        except Exception as ex:   # (virtual-memory API unavailable, out of memory ...): plain allocations, said so in the line
            arena, placement = None, "as allocated (arena failed: %s)" % str(ex)[:120]
    if arena is not None:
        trace_regions = arena.regions
        placement = {"arena_candidates": cand, "record_kernel_alone_ms_per_candidate": [round(t, 4) for t in arena.measurements_ms],
                     "kept_ms": [round(t, 4) for t in arena.region_ms]}
    else:
        trace_buf = torch.zeros(regions * chunk * elem_stride, dtype=torch.uint8, device=dev)
        trace_regions = [trace_buf[r * chunk * elem_stride:(r + 1) * chunk * elem_stride] for r in range(regions)]
    ifs = chip.in_field_layout()[0]
    in_field_buf = torch.zeros(regions * chunk * ifs, dtype=torch.uint8, device=dev)   # assert_in_field witness (src/chip.rs:106)
    out = torch.zeros((regions * chunk, chip.num_limbs), dtype=chip.torch_dtype, device=dev)
    status = torch.zeros(regions * chunk, dtype=torch.uint8, device=dev)
    workspaces = [torch.zeros(chip.workspace_bytes(chunk, pl.num_mul_mods), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    xc = [H.AssignedInteger(x_dev.limbs_dev[c * chunk:(c + 1) * chunk], w) for c in range(chunks)]
    nc = [n_dev if (args.shared_modulus and not os.environ.get("H2R_BENCH_REPLICATE_N")) else H.AssignedInteger(n_dev.limbs_dev[c * chunk:(c + 1) * chunk], w) for c in range(chunks)]
    pipes = [] if args.no_pipeline else [H.Pipeline(chip, depth=args.pipeline_depth, side_streams=args.side_streams) for _ in range(producers)]
    pipe = pipes[0] if pipes else None
    pipeline_form = None
    if pipe is not None:   # what the library chose for calls of this size on this stream (it measures whether its streams sit on three hardware queues)
        pi = pipe.info(chunk)
        pipeline_form = {"record_form": ["one-launch step", "two-queue", "side stream"][pi.record_form], "three_queues": {0: False, 1: True}.get(pi.three_queues),
                         "probe_ms": round(pi.probe_ms, 3), "probe_span_ms": round(pi.probe_span_ms, 4), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    # producer p issues calls p, p + P, p + 2P, ... on its own stream; buffer set k % (depth * P) belongs to producer k % P
    prod_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(producers - 1)]
    counter = [0]

    def join_all():
        for pp, ss in zip(pipes, prod_streams):
            with torch.cuda.stream(ss):
                pp.join()

    def call(c):
        """One 1,024-signature (chunk) call: chunk c of the shard."""
        if producers > 1:
            with torch.cuda.stream(prod_streams[counter[0] % producers]):
                return call_on(c, pipes[counter[0] % producers])
        return call_on(c, pipe)

    def call_on(c, pipe):
        k = counter[0]
        counter[0] += 1
        r = (k % nbuf) if chunks == 1 else c          # trace / result region
        ws = workspaces[k % nbuf]
        sl = slice(r * chunk, (r + 1) * chunk)
        tb = trace_regions[r]
        fb = in_field_buf[r * chunk * ifs:(r + 1) * chunk * ifs]
        if pipe is None:
            chip.pow_mod_fixed_exp(xc[c], e, nc[c], want_trace=True, trace_buf=tb, check_in_field=True,
                                   workspace=ws, out=out[sl], status=status[sl], in_field_buf=fb)
        elif verify:
            if args.messages:   # the SHA-256 / hashed-message step rides on the call's step launch (h2r_pipeline_signature_verifier)
                cs = slice(c * chunk, (c + 1) * chunk)
                pipe.signature_verifier(msgs_dev[cs], None, args.messages, xc[c], e, nc[c], tb, ws, out[sl], valid[sl], status[sl],
                                        hashed_dev[cs], digest_dev[cs], hm_dev[cs])
                return r
            pipe.verify_pkcs1v15(xc[c], e, nc[c], hashed_dev[c * chunk:(c + 1) * chunk], tb, ws, out[sl], valid[sl], status[sl])
        else:
            pipe.modpow_public_key(xc[c], e, nc[c], tb, ws, out[sl], status[sl], None if args.no_in_field else fb)
        return r

    def step():
        for c in range(chunks):
            last = call(c)
        return last

    # initialisation that is not part of any step (code-object load, stream / event creation on first use): one call,
    # then the W warm-up steps the caller asked for
    call(0)
    counter[0] = 0
    if pipe is not None:   # the next call reuses buffer set 0: its records are complete first (and every producer stream starts behind the set-up)
        join_all()
    torch.cuda.synchronize()
    # Clock ramp: after the idle stretch of the set-up above (allocations, arena search, host work) the first ~50 launches of
    # sustained work run 3-12 % slower than the rest (tools/clock_ramp_probe.py, profiles/r03_clock_ramp.txt: step launch 0.196 /
    # 0.207 / 0.188 ms for launches 0-19 / 20-39 / 40-59, 0.184 ms from then on) -- the device's power management, not this code.
    # W = 5 warm-up steps end inside that ramp, so the bench first keeps the GPU busy with `--clock-warmup-calls` untimed calls
    # of the same workload (reported in config.untimed_clock_warmup_calls); the W warm-up steps and the K timed steps follow at once.
    ramp_steps = 0
    if pipe is not None and args.clock_warmup_calls > 0:
        t_ramp = time.perf_counter()
        while ramp_steps * chunks < args.clock_warmup_calls:
            step()
            ramp_steps += 1
            if ramp_steps % 8 == 0:   # bounded in time as well: at most ~0.3 s of work whatever the workload
                torch.cuda.synchronize()
                if time.perf_counter() - t_ramp > 0.3:
                    break
    # The W warm-up steps run with the library's per-launch timing armed: that is where the bench learns how a call is issued (one step
    # launch per call?) and gets per-launch durations -- NOT in the timed region, where stamping every dispatch costs 6-7 us per
    # 1,024-signature step (same-box A/B, profiles/r04_kernel_timing_tax.txt: 0.2001 -> 0.1932 ms per step).
    n_prof = (3 + 32 + 2 * (chunk // 256)) * max(steps, warmup) * chunks + 8   # chain + record + in-field kernel per call; (+32: a long exponent walked as up to 16 segments)
    _lib.profile_enable(0 if args.no_kernel_timing else n_prof)
    for _ in range(warmup):
        step()
    if pipe is not None:
        join_all()
    torch.cuda.synchronize()
    w_trace, w_chain, w_step = _lib.profile_read(_lib.KERNEL_TRACE), _lib.profile_read(_lib.KERNEL_CHAIN), _lib.profile_read(_lib.KERNEL_STEP)
    _lib.profile_enable(0)
    # One step launch per call (the RSA-2048 / RSA-1024 pipeline): the timed region carries TWO events on the launch stream instead -- behind
    # the first call (a chain kernel: the pipeline starts empty) and behind the last step launch -- and the dominant kernel's average
    # launch time is their distance / (calls - 1): the period of back-to-back launches, an upper bound of the kernel's own duration.
    span_ok = (not args.no_kernel_timing and not args.per_launch_timing and pipe is not None and producers == 1 and steps * chunks >= 2
               and warmup * chunks > 0 and len(w_step) == warmup * chunks)
    # Record kernels alternating between two side streams (no step launches, one record launch per call in the warm-up): two of them are in
    # flight at a time, so a launch's own duration counts the time it shares -- the roofline takes the launches' PERIOD (timed wall time /
    # launches) instead, and the timed region carries no per-launch stamps at all.
    overlap_ok = (not span_ok and not args.no_kernel_timing and not args.per_launch_timing and pipe is not None and producers == 1 and args.side_streams == 2
                  and not w_step and warmup * chunks > 0 and len(w_trace) >= warmup * chunks)
    # (record launches per call, from the warm-up: 1, or more where the library walks a large call as sub-batches)
    launches_per_call_w = len(w_trace) / float(warmup * chunks) if (overlap_ok and warmup * chunks) else 1.0
    if not span_ok and not overlap_ok:
        _lib.profile_enable(0 if args.no_kernel_timing else n_prof)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    env.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_i in range(steps):
        for c_i in range(chunks):
            last = call(c_i)
            if span_ok and s_i == 0 and c_i == 0:
                ev0.record()
    if span_ok:
        ev1.record()
    if pipe is not None:
        join_all()   # every step's trace is complete before the clock stops
    torch.cuda.synchronize()
    env.barrier()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    if span_ok:
        period_ms = ev0.elapsed_time(ev1) / (steps * chunks - 1)
        step_ms = [period_ms] * (steps * chunks - 1)
        trace_ms = [sum(w_trace) / len(w_trace)] if w_trace else []      # (the record kernel alone appears once, at the join: the warm-up's)
        chain_ms = w_chain
    elif overlap_ok:
        step_ms = []
        n_rec = max(1, int(round(steps * chunks * launches_per_call_w)))
        trace_ms = [1e3 * dt / n_rec] * n_rec
        chain_ms = w_chain
    else:
        trace_ms = _lib.profile_read(_lib.KERNEL_TRACE)
        chain_ms = _lib.profile_read(_lib.KERNEL_CHAIN)
        # RSA-2048 pipelined: the library issues a step as ONE launch (records of call k + chains of call k+1); the record kernel
        # alone then appears once, at the join
        step_ms = _lib.profile_read(_lib.KERNEL_STEP)
        _lib.profile_enable(0)

    # post-run: correctness of what was timed + the result gather (rank 0 receives every shard's x^e mod n)
    assert int(status.max().item()) == 0 or (w, bits) != (64, 2048), "unexpected per-element status"
    c_last = chunks - 1                                    # shard chunk the last call processed
    res = out[last * chunk:(last + 1) * chunk]
    got = H.AssignedInteger(res.contiguous(), w).to_big_uint()
    base = c_last * chunk
    # the trace that was timed is the real thing: first element of the last call, byte-exact vs pow() through q*n+r
    tr = H.Trace(chip, trace_regions[last], chunk, pl)
    tr.elem_stride = elem_stride
    q0 = int.from_bytes(tr.plane(0, 0, "Q").tobytes(), "little")
    r0 = int.from_bytes(tr.plane(0, 0, "R").tobytes(), "little")
    assert xs[base] * xs[base] == q0 * ns[base] + r0, "first mul_mod record of the timed trace is wrong"
    for i in (0, 1, 2, chunk - 1):
        if i < chunk:
            assert got[i] == pow(xs[base + i], e, ns[base + i]), "GPU result differs from pow(x, e, n)"
    # one FULL call per shard audited in place on the device (h2r_pow_trace_check: every record of every element of the LAST call
    # against SURVEY Appendix C's relations, independently of the producing kernels); the verdict bytes travel with the results
    import ctypes as _ct
    bad = torch.zeros(chunk, dtype=torch.int32, device=dev)
    first_bad = torch.zeros(chunk, dtype=torch.int32, device=dev)
    eb_ = e.to_bytes((e.bit_length() + 7) // 8, "little")
    st_last = status[last * chunk:(last + 1) * chunk]
    _lib.check(_lib.lib().h2r_pow_trace_check(chip._ctx, _ct.byref(vl.pow if verify else pl), xc[c_last].data_ptr(), nc[c_last].data_ptr(), eb_, len(eb_),
                                              chip._flags(nc[c_last], chunk), trace_regions[last].data_ptr(), elem_stride,
                                              workspaces[(counter[0] - 1) % nbuf].data_ptr(), chunk, st_last.data_ptr(), bad.data_ptr(),
                                              first_bad.data_ptr(), chip._stream()), "h2r_pow_trace_check")
    torch.cuda.synchronize()
    audit = status[:shard].clone() if chunks > 1 else st_last.clone()
    aud_last = ((bad != 0) | (st_last != 0)).to(torch.uint8)
    if chunks > 1:
        audit[last * chunk:(last + 1) * chunk] = aud_last
    else:
        audit = aud_last
    shard_out = out[:shard].contiguous() if chunks > 1 else res.contiguous()
    gathered, audit_all = env.gather_to_rank0(shard_out, audit.contiguous())
    # every rank judges its own shard, rank 0 also the gathered results; the verdict is then AGREED (halo2_rsa_amd.dist.finish_with_verdict):
    # a failed audit anywhere makes every rank exit non-zero with one line, and no result line is printed
    verdict_ok, verdict_msg = True, ""
    try:
        n_mine = int((audit != 0).sum().item())
        assert n_mine == 0 or (w, bits) != (64, 2048) or args.shared_modulus, "%d elements of this rank's shard with a status or a violated witness relation" % n_mine
        if env.rank == 0:
            verdict_ok, verdict_msg = check_gathered(gathered, audit_all, env.world, shard, shard * env.world, w, bits, e,
                                                     sample=(chunks > 1 or env.world > 1 or shard > 1024) and not args.shared_modulus,
                                                     strict=(w, bits) == (64, 2048) and not args.shared_modulus)
    except AssertionError as ex:
        verdict_ok, verdict_msg = False, str(ex)
    if env.world > 1:
        from halo2_rsa_amd.dist import finish_with_verdict
        finish_with_verdict(env, verdict_ok, verdict_msg)
    assert verdict_ok, verdict_msg

    if env.rank == 0:
        # written (pow stream + the assert_in_field stream of modpow_public_key) + inputs read
        algo_bytes_per_assign = pl.stream_bytes + chip.in_field_layout()[1] + 2 * chip.num_limbs * chip.layout.limb_bytes
        # trace_kernel's algorithmic output per launch.  A large pipelined call that finds the pipeline empty is walked by
        # the library as sub-batches of growing size: the profile then holds more launches than calls, so the figure is
        # the mean over the timed launches (all signatures of the timed region / launches) -- equal to chunk * bytes when
        # every call is one launch.
        n_launches = (len(trace_ms) + len(step_ms)) if (trace_ms or step_ms) else steps * chunks
        per_launch_batch = chunk * steps * chunks / n_launches
        trace_bytes_per_launch = int(round(per_launch_batch * (pl.num_mul_mods * chip.layout.stream_bytes)))
        if step_ms:   # the dominant kernel is the step launch; its algorithmic bytes are the records it writes
            kd = bits // 32
            dom_ms, dom_name = step_ms, "step_kernel<%d,%d,%d,%d> (records of call k + chains of call k+1, one launch)" % (
                kd, 4 if kd <= 64 else (6 if kd == 96 else 8), w, chip.num_limbs)
        else:
            dom_ms, dom_name = trace_ms, "trace_kernel<%d,%d>" % (w, chip.num_limbs)
        avg_trace_s = (sum(dom_ms) / len(dom_ms)) / 1e3 if dom_ms else float("nan")
        achieved = trace_bytes_per_launch / avg_trace_s / 1e9 if dom_ms else None
        if chunks == 1 and env.world == 1:
            wl = "%s batch=%d per GPU, %d-bit limbs, full op-trace (%d B/assign)" % (args.workload, chunk, w, algo_bytes_per_assign)
        elif chunks == 1:
            wl = ("%s global batch=%d sharded x%d (%d per GPU = one pipelined call per step), %d-bit limbs, full op-trace "
                  "(%d B/assign), traces resident on the producing GPU" %
                  (args.workload, global_batch, env.world, shard, w, algo_bytes_per_assign))
        else:
            wl = ("%s global batch=%d sharded x%d (%d per GPU, walked as %d pipelined calls of %d), %d-bit limbs, full op-trace "
                  "(%d B/assign), traces resident on the producing GPU" %
                  (args.workload, global_batch, env.world, shard, chunks, chunk, w, algo_bytes_per_assign))
        line = {
            "metric": "RSA-2048 pkcs1v15 witness assigns/sec" if bits == 2048 else "RSA-%d witness assigns/sec" % bits,
            "value": round(global_batch * steps / dt, 1),
            "unit": "assigns/s",
            "n_gpus": env.world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * dt / steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u%d" % w, "data": "synthetic",
            "config": {"workload": wl,
                       "per_gpu_batch": shard, "global_batch": global_batch, "calls_per_step": chunks, "signatures_per_call": chunk,
                       "path": ("RSASignatureVerifier from %d-byte messages (SHA-256 + hashed-message limbs + in-field + modpow + EM check)" % args.messages
                                if verify and args.messages else
                                "verify_pkcs1v15_signature (in-field + modpow + EM check)" if verify else "modpow_public_key"),
                       "mul_mods_per_assign": pl.num_mul_mods, "parallelism": "signature-sharded x%d" % env.world,
                       "ranks": env.describe()["ranks"], "rccl_version": env.describe()["rccl_version"], "communicator": env.describe()["communicator"],
                       "collective_backend": (env.backend + (" (RCCL)" if env.backend == "nccl" else "")) if env.initialised else "none (single process)",
                       "post_run_check": "results of every shard vs pow(); last call of every shard audited in place (all %d elements, every record), verdict bytes gathered with the results" % chunk,
                       "pipeline": ((("one launch per step: records of call k + chains of call k+1 (step_kernel), %d buffer sets" % args.pipeline_depth)
                                     if step_ms else
                                     ("chain k+1 || trace k, %d buffer sets, %d record stream(s)" % (args.pipeline_depth, args.side_streams))) +
                                    (", %d producers (a pipeline and a stream each, calls alternate)" % producers if producers > 1 else ""))
                                   if pipe is not None else "none",
                       "pipeline_form": pipeline_form,
                       "untimed_clock_warmup_calls": ramp_steps * chunks,
                       # everything that ran before the timed region: 1 set-up call + the clock warm-up calls + the W warm-up steps
                       "warmup_calls_total": 1 + ramp_steps * chunks + warmup * chunks,
                       "buffer_placement": placement if placement else "as allocated"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": pmc_traffic(dom_name.split(" ")[0], int(per_launch_batch)) if per_launch_batch == chunk else None,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this kernel at this batch, "
                                           "committed; PMC counters cannot be read from inside the bench process)",
                         "kernel": dom_name,
                         "launches_timed": len(dom_ms), "signatures_per_launch": round(per_launch_batch, 1),
                         "timing": ("avg_launch_ms = the record launches' period (timed wall time / launches): the record kernels alternate between two side "
                                    "streams and two are in flight at a time; avg_launch_ms_in_flight = a launch's own start-to-end time in the stamped warm-up steps, while the "
                                    "pipeline fills (at steady state two launches share the device throughout and a launch lasts about two periods: the kernel trace "
                                    "under rocprofv3, profiles/r04_timeline_pipeline.txt, shows duration, overlap and period per launch)") if overlap_ok else
                                   ("two HIP events on the launch stream over the timed region (behind the first call and behind the last step launch): "
                                    "avg_launch_ms = the launches' period, an upper bound of the kernel's duration; per-launch dispatch stamps "
                                    "(which cost 6-7 us per step) only in the warm-up steps: avg_launch_ms_stamped_warmup") if span_ok else
                                   "per-launch HIP events stamped by the dispatch packets, in the timed region",
                         "avg_launch_ms_stamped_warmup": round(sum(w_step) / len(w_step), 4) if w_step else None,
                         "avg_launch_ms_in_flight": round(sum(w_trace) / len(w_trace), 4) if (overlap_ok and w_trace) else None,
                         "launches_per_call": round(n_launches / (steps * chunks), 2),   # > 1: sub-batches of a large call, or segments of a long exponent
                         "avg_launch_ms": round(1e3 * avg_trace_s, 4) if dom_ms else None,
                         "algorithmic_bytes_per_launch": trace_bytes_per_launch,
                         "record_kernel_alone_avg_ms": round(sum(trace_ms) / len(trace_ms), 4) if (step_ms and trace_ms) else None,
                         "chain_kernel_avg_ms": round(sum(chain_ms) / len(chain_ms), 4) if chain_ms else None},
            "whole_path_hbm_frac": round(global_batch * steps / dt * algo_bytes_per_assign / (env.world * HBM_PEAK_GBS * 1e9), 4),
        }
        if env.world == 1 and args.pmc_traffic == "auto" and dom_ms and per_launch_batch == chunk:
            wl_args = ["--workload", args.workload, "--batch", str(chunk), "--chunks", str(chunks), "--pipeline-depth", str(args.pipeline_depth),
                       "--side-streams", str(args.side_streams)]
            wl_args += ["--verify"] if args.verify else []
            wl_args += ["--messages", str(args.messages)] if args.messages else []
            wl_args += ["--no-pipeline"] if args.no_pipeline else []
            hbm, how = measured_pmc_traffic(wl_args, dom_name.split("<")[0])
            if hbm is not None:
                line["roofline"]["traffic"] = hbm
                line["roofline"]["traffic_source"] = how
            else:
                line["roofline"]["traffic_source"] += "; live measurement skipped: " + how
        if env.world == 1 and not args.no_cpu_baseline and not args.shared_modulus:
            line["cpu_baseline"] = cpu_baseline(w, bits, e, un, ux)
        if env.world == 1 and args.gpus == 1 and args.scale_anchor == "auto" and chunks == 1 and chunk == 1024 and not verify and args.workload == "rsa2048_e65537":
            # The driver's N = 1 point is BASELINE config 2 (1,024 signatures per step), its N > 1 points config 3 (8,192 per GPU as four
            # calls of 2,048): a like-for-like origin for the 1 -> 8 curve is the N > 1 per-GPU workload run on this one GPU.
            line["scale_anchor"] = scale_anchor_line(args)
            if args.sub_runs == "auto" and args.placement_candidates < 0:
                line.update(sub_run_lines(args))
        print(json.dumps(line))
    env.finalize()


def scale_anchor_line(args):
    """`bench.py --gpus 1 --batch 2048 --chunks 4` (the per-GPU workload of every N > 1 run) in a fresh process, same steps / warm-up."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--batch", "2048", "--chunks", "4", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--pmc-traffic", "off", "--scale-anchor", "off"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, check=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        return {"what": "the N > 1 per-GPU workload (8,192 signatures per step as four pipelined calls of 2,048) on this one GPU: compare per-GPU values of the N > 1 runs with THIS",
                "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "per_gpu_batch": d["config"]["per_gpu_batch"],
                "calls_per_step": d["config"]["calls_per_step"], "roofline_frac": d["roofline"]["frac"], "whole_path_hbm_frac": d["whole_path_hbm_frac"]}
    except Exception as ex:
        return {"error": str(ex)[:200]}


if __name__ == "__main__":
    main()
