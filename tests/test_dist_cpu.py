"""N > 1 harness on CPU: world_size-2 gloo run of the sharding / barrier / MAX-reduce / broadcast /
gather plumbing that bench.py uses around the (GPU-only) hot path."""
import os
import subprocess
import sys
import textwrap

from halo2_rsa_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_the_batch():
    for total in (0, 1, 7, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)   # BASELINE config 3: 8,192 per GPU


def test_gloo_world2_harness(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent('''
        import sys, torch
        sys.path.insert(0, %r)
        from halo2_rsa_amd.dist import DistEnv, shard_range
        env = DistEnv.from_environment(2)
        env.init("gloo")
        cfg = env.broadcast_ints([65537, 1024, 5, 1] if env.rank == 0 else [0, 0, 0, 0])
        assert cfg == [65537, 1024, 5, 1]
        lo, hi = shard_range(10, env.rank, env.world)
        shard = torch.arange(lo, hi, dtype=torch.int64).reshape(-1, 1).repeat(1, 4)
        env.barrier()
        t = env.max_over_ranks(1.0 + env.rank)
        assert t == 2.0
        g = env.gather_to_rank0(shard)
        if env.rank == 0:
            assert g.shape == (10, 4) and g[:, 0].tolist() == list(range(10))
        # results + one status byte per element in one call (bench.py's post-run audit verdicts travel this way)
        g2, st2 = env.gather_to_rank0(shard, torch.full((hi - lo,), env.rank + 1, dtype=torch.uint8))
        if env.rank == 0:
            assert g2.shape == (10, 4) and st2.tolist() == [1] * 5 + [2] * 5
        assert env.describe()["ranks"] == 2
        # bench.py's config-3 sharding: every rank synthesises ITS shard of the seeded global batch; rank 0 can
        # regenerate any element of any shard (that is how it checks samples of every shard against pow())
        import numpy as np
        import bench
        lo, hi = shard_range(12, env.rank, env.world)
        ns, xs, un, ux = bench.synth_inputs(64, 2048, lo, hi)
        assert un.limbs.shape == (6, 32) and all(x < n and n >> 2047 == 1 and n & 1 for n, x in zip(ns, xs))
        gx = env.gather_to_rank0(torch.from_numpy(ux.limbs.view(np.int64)))
        if env.rank == 0:
            golden = bench.load_golden(64, 2048)
            host = gx.numpy().view(np.uint64)
            for gidx in range(12):
                n_g, x_g = bench.synth_element(64, 2048, gidx, golden)
                assert sum(int(t) << (64 * i) for i, t in enumerate(host[gidx])) == x_g
            assert int(golden[0]["sig"]) == bench.synth_element(64, 2048, 0, golden)[1]
            print("GLOO_OK")
        env.finalize()
        # the out-of-band rendezvous of the C-ABI communicator (h2r_dist_init needs rank 0's 128-byte RCCL id on every rank):
        # through the agent's TCPStore under torchrun, exactly as bench.py --gpus N does before h2r_dist_init
        from halo2_rsa_amd.dist import exchange_bytes
        payload = bytes(range(128)) if env.rank == 0 else bytes(128)
        got = exchange_bytes("test_id", payload, env.rank, env.world)
        assert got == bytes(range(128))
        print("ID_OK %%d" %% env.rank)
        # the backend agreement of bench.py: all ranks ok -> True everywhere; one rank not ok -> False everywhere
        from halo2_rsa_amd.dist import agree_all
        assert agree_all("agree_a", True, env.rank, env.world) is True
        assert agree_all("agree_b", env.rank != 1, env.rank, env.world) is False
        # the same key a second time in one job (a second communicator, a restarted worker group): a fresh namespace, not the
        # first use's stale id / counters
        got2 = exchange_bytes("test_id", bytes(reversed(range(128))) if env.rank == 0 else bytes(128), env.rank, env.world)
        assert got2 == bytes(reversed(range(128)))
        assert agree_all("agree_b", True, env.rank, env.world) is True
        print("AGREE_OK %%d" %% env.rank)
    ''' % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", str(script)],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GLOO_OK" in out.stdout
    assert "ID_OK 0" in out.stdout and "ID_OK 1" in out.stdout
    assert "AGREE_OK 0" in out.stdout and "AGREE_OK 1" in out.stdout


# ---- world 8 on CPU: bench.py's own N > 1 path (argument handling, relaunch under torch.distributed.run, shards, gather, verdict) with
# only the GPU call replaced (bench.py --dry-run-dist).  The 1 -> 8 curve itself stays unmeasured on hardware. ----

def _dry_run(extra_env, *extra_args, timeout=300):
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", **extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-dist", "--steps", "7", "--warmup", "2", *extra_args],
                         capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines


def test_world8_batch_not_divisible_by_8():
    """`python bench.py --gpus 8 ...` without a launcher re-executes itself under torch.distributed.run (relaunch_under_torchrun); 1,003 signatures over
    8 ranks = three shards of 126 and five of 125, gathered (padded) to rank 0, samples of EVERY shard checked there, one JSON line."""
    out, lines = _dry_run({}, "--dry-total", "1003")
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 8 and ln["global_batch"] == 1003 and ln["shard_sizes"] == [126, 126, 126, 125, 125, 125, 125, 125]
    assert ln["steps"] == 7 and ln["warmup"] == 2 and ln["e"] == 65537          # rank 0's arguments reached every rank (broadcast)
    assert "post-run check" not in out.stderr


def test_world8_one_failed_audit_fails_every_rank():
    """Rank 5 reports a violated witness relation in its shard: every rank exits non-zero with ONE line, rank 5's says what failed, nobody prints a
    result line, nobody hangs in a collective."""
    out, lines = _dry_run({"H2R_DRY_FAIL_RANK": "5"}, "--dry-total", "1003")
    assert out.returncode != 0
    assert lines == []
    said = [ln for ln in out.stderr.splitlines() if "post-run check" in ln]
    assert len(said) == 8 and sorted(int(ln.split()[1].rstrip(":")) for ln in said) == list(range(8))
    failed = [ln for ln in said if "FAILED:" in ln]
    assert len(failed) == 1 and failed[0].startswith("rank 5:") and "1 elements of this rank's shard" in failed[0]
    assert sum("FAILED on another rank" in ln for ln in said) == 7


def test_world8_a_rank_arrives_late():
    """Rank 2 reaches the rendezvous 5 s after the others: the run completes with the same result line (the timing barrier brackets the step, so the
    late start is not in max_over_ranks_s)."""
    import time
    t0 = time.time()
    out, lines = _dry_run({"H2R_DRY_LATE_RANK": "2"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert time.time() - t0 >= 5.0
    assert len(lines) == 1 and lines[0]["global_batch"] == 65536       # BASELINE config 3: 8,192 per GPU (the N > 1 default)
    assert lines[0]["shard_sizes"] == [8192] * 8 and lines[0]["max_over_ranks_s"] < 4.0
