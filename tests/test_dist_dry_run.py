"""BASELINE config 3 (64k RSA-2048 signatures sharded over 8 GPUs, results gathered over RCCL) to the limit of ONE GPU: the h2r_dist
communicator is created under torchrun (world = 1, H2R_FORCE_DIST semantics: a real RCCL communicator), the eight 8,192-signature shards
run one after the other on the one device, every shard's results and audit verdicts travel through h2r_dist_gather_results into the
64k x 256 B + 64k x 1 B receive buffers AT THE OFFSETS of an 8-rank job (h2r_dist_shard_range), and rank 0 runs the N = 8 post-run check
over all of it: samples of every shard against pow(), every element's status, every shard audited in place.  What it cannot show is the
eight-rank collective itself: the 1 -> 8 curve stays unmeasured on hardware (README)."""
import os
import subprocess
import sys
import textwrap

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_buffers_and_post_run_check_with_one_rank(tmp_path):
    script = tmp_path / "dry.py"
    script.write_text(textwrap.dedent('''
        import ctypes, os, sys, torch
        sys.path.insert(0, %r)
        import numpy as np
        import bench
        import halo2_rsa_amd as H
        from halo2_rsa_amd import _lib
        from halo2_rsa_amd.dist import H2RDist, shard_range
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert world == 1
        torch.cuda.set_device(0)
        chip = H.BigIntChip(64, 2048)
        dist = H2RDist(chip, rank, world, 0)                       # h2r_dist_init: a real RCCL communicator (one rank)
        assert dist.describe()["ranks"] == 1 and dist.describe()["rccl_version"] > 0
        L = _lib.lib()
        TOTAL, RANKS, e = 65536, 8, 65537
        res_all = torch.full((TOTAL, 32), -1, dtype=torch.int64, device="cuda")     # 64k x 256 B
        st_all = torch.full((TOTAL,), 0xEE, dtype=torch.uint8, device="cuda")        # 64k x 1 B
        assert res_all.numel() * 8 + st_all.numel() == 65536 * 257
        pl = chip.pow_fixed_layout(e)
        golden = bench.load_golden(64, 2048)
        for r in range(RANKS):
            lo, hi = shard_range(TOTAL, r, RANKS)
            lo_c, hi_c = ctypes.c_uint64(), ctypes.c_uint64()
            assert L.h2r_dist_shard_range(TOTAL, r, RANKS, ctypes.byref(lo_c), ctypes.byref(hi_c)) == 0 and (lo_c.value, hi_c.value) == (lo, hi) == (r * 8192, (r + 1) * 8192)
            ns, xs, un, ux = bench.synth_inputs(64, 2048, lo, hi)
            x, n = chip.assign_integer(ux), chip.assign_integer(un)
            shard = chip.pow_mod_fixed_exp(x, e, n, check_in_field=True)       # RSAChip::modpow_public_key of the shard
            bad, first = shard.audit()                                                # the in-place audit of every record of the shard
            verdict = ((shard.status != 0) | (bad != 0)).to(torch.uint8)
            # the shard's slice of the 8-rank receive buffers: rank r's results land at element r * 8192
            _lib.check(L.h2r_dist_gather_results(dist._d, shard.value.limbs_dev.data_ptr(), verdict.data_ptr(), hi - lo,
                                                 res_all.data_ptr() + lo * 256, st_all.data_ptr() + lo, chip._stream()), "h2r_dist_gather_results")
            torch.cuda.synchronize()
            del shard
        # ---- rank 0's post-run check of an N = 8 job ----
        assert int(st_all.max().item()) == 0, "a shard reported a failed element or a violated record"
        host = res_all.cpu().numpy().view(np.uint64)
        for r in range(RANKS):
            for g in (r * 8192, r * 8192 + 1, r * 8192 + 4097, (r + 1) * 8192 - 1):
                n_g, x_g = bench.synth_element(64, 2048, g, golden)
                got = sum(int(t) << (64 * i) for i, t in enumerate(host[g]))
                assert got == pow(x_g, e, n_g), g
        # KAT1 / KAT2 sit at elements 0, 1 of the global batch
        assert sum(int(t) << (64 * i) for i, t in enumerate(host[0])) == pow(int(golden[0]["sig"]), e, int(golden[0]["n"]))
        dist.finalize()
        print("DRY_RUN_OK", TOTAL)
    ''' % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", H2R_FORCE_DIST="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29581", str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "DRY_RUN_OK 65536" in out.stdout
