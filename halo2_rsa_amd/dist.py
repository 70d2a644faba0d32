"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).

Signatures are independent, so the hot path shards with NO data-path collective: each
rank owns a contiguous slice of the batch and keeps its traces resident on its own GPU.  Collectives
are only: the configuration broadcast, the timing barrier / MAX-reduce, and the gather of the small
per-signature results (num_limbs limbs + status) to rank 0.
"""
import os
from dataclasses import dataclass
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous batch shards: rank g gets [g*total/world, (g+1)*total/world) (remainder spread low)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class DistEnv:
    rank: int
    local_rank: int
    world: int
    initialised: bool = False
    backend: str = ""
    force: bool = False   # initialise the process group even for world == 1 (exercises the RCCL path)

    @staticmethod
    def from_environment(expected_world: int = 1, force: bool = False) -> "DistEnv":
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != expected_world and world != 1:
            raise RuntimeError("--gpus %d but WORLD_SIZE=%d" % (expected_world, world))
        if expected_world > 1 and world == 1:
            raise RuntimeError("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                               "--nproc-per-node %d" % (expected_world, expected_world))
        return DistEnv(rank, local, world, force=force)

    def init(self, backend: str):
        self.backend = backend
        if self.world > 1 or self.force:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self.initialised = True

    def _dev(self):
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.initialised:
            if self.backend == "nccl":
                dist.barrier(device_ids=[self.local_rank])
            else:
                dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self.initialised:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast_ints(self, values: List[int]) -> List[int]:
        if not self.initialised:
            return list(values)
        t = torch.tensor(values, dtype=torch.int64, device=self._dev())
        dist.broadcast(t, src=0)
        return [int(v) for v in t.tolist()]

    def gather_to_rank0(self, shard: torch.Tensor, status: torch.Tensor = None, total: int = None):
        """Gather the per-rank result tensors; rank 0 returns the concatenation, others None.  With `status` (uint8 [elems]) the pair
        (results, status) -- (None, None) on the other ranks.  Shards are equally sized unless `total` is given: then rank r holds the
        shard_range(total, r, world) elements of a batch that the ranks do not divide (the shards travel padded to the largest one)."""
        return gather_with_total(self, DistEnv._gather_equal, shard, status, total)

    def _gather_equal(self, shard: torch.Tensor, status: torch.Tensor = None):
        if status is not None:
            return self._gather_one(shard), self._gather_one(status)
        return self._gather_one(shard)

    def _gather_one(self, shard: torch.Tensor):
        if not self.initialised:
            return shard
        if self.backend == "nccl":
            outs = [torch.empty_like(shard) for _ in range(self.world)]
            dist.all_gather(outs, shard)  # RCCL all-gather over xGMI; results are tiny (num_limbs limbs each)
            return torch.cat(outs, 0) if self.rank == 0 else None
        host = shard.cpu()   # gloo (CPU tests, one-GPU developer runs): gather on the host, hand back on the shard's device
        outs = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(host, outs, dst=0)
        return torch.cat(outs, 0).to(shard.device) if self.rank == 0 else None

    def describe(self) -> dict:
        return {"ranks": (dist.get_world_size() if self.initialised else 1), "rank0": (dist.get_rank() if self.initialised else 0),
                "rccl_version": None, "communicator": ("torch.distributed " + self.backend) if self.initialised else "none (single process)"}

    def finalize(self):
        if self.initialised:
            dist.barrier() if self.backend != "nccl" else dist.barrier(device_ids=[self.local_rank])
            dist.destroy_process_group()
            self.initialised = False


def gather_with_total(env, gather_equal, shard: torch.Tensor, status, total):
    """Uneven shards over an equal-size gather: pad to ceil(total / world) elements, gather, and let rank 0 cut every rank's padding off."""
    if total is None or env.world == 1:
        return gather_equal(env, shard, status)
    lo, hi = shard_range(total, env.rank, env.world)
    if shard.shape[0] != hi - lo or (status is not None and status.shape[0] != hi - lo):
        raise ValueError("rank %d holds %d elements, shard_range(%d, %d, %d) is %d" % (env.rank, shard.shape[0], total, env.rank, env.world, hi - lo))
    most = -(-total // env.world)
    pad = most - (hi - lo)
    if pad:
        shard = torch.cat([shard, shard.new_zeros((pad,) + tuple(shard.shape[1:]))], 0)
        if status is not None:
            status = torch.cat([status, status.new_zeros(pad)], 0)
    got = gather_equal(env, shard.contiguous(), status.contiguous() if status is not None else None)
    if env.rank != 0:
        return got
    sizes = [shard_range(total, r, env.world)[1] - shard_range(total, r, env.world)[0] for r in range(env.world)]
    cut = lambda t: torch.cat([t[r * most:r * most + sizes[r]] for r in range(env.world)], 0)
    return (cut(got[0]), cut(got[1])) if status is not None else cut(got)


def finish_with_verdict(env, ok: bool, message: str = "", key: str = "post_run_check"):
    """The end of a multi-rank run: every rank leaves with the SAME verdict.  ok = this rank's own post-run check; if ANY rank failed, every
    rank writes one line to stderr (the failing ones say what failed) and exits with status 3 -- no rank prints a result line next to a
    failed audit elsewhere, and none is left waiting in a collective for a rank that raised."""
    import sys
    all_ok = agree_all(key, ok, env.rank, env.world)
    if all_ok:
        return
    sys.stderr.write("rank %d: post-run check %s\n" % (env.rank, ("FAILED: " + message) if not ok else "passed here, FAILED on another rank: leaving without a result line"))
    sys.stderr.flush()
    try:
        env.finalize()
    except Exception:
        pass
    sys.exit(3)


_key_uses = {}


def _run_key(key: str) -> str:
    """Keys in torchrun's long-lived agent store are namespaced per run, per worker restart and per use: a restarted worker group
    (or a second H2RDist in one job) must not read the previous incarnation's RCCL id or find its counters already at `world`.
    Every rank makes the same sequence of calls, so the per-process use counter agrees across ranks."""
    k = _key_uses.get(key, 0)
    _key_uses[key] = k + 1
    return "%s/%s/%s/%d" % (key, os.environ.get("TORCHELASTIC_RUN_ID", "norun"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), k)


def exchange_bytes(key: str, payload, rank: int, world: int) -> bytes:
    """Rank 0's `payload` (bytes) to every rank, out of band of any collective library: through the TCPStore torchrun's agent
    already serves at MASTER_ADDR:MASTER_PORT (every worker is a client of it), or -- launched any other way -- one that rank 0
    serves itself.  This is all the rendezvous h2r_dist_init needs (the 128-byte RCCL id)."""
    from datetime import timedelta
    if world == 1:
        return bytes(payload)
    key = _run_key(key)
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29531")), world,
                          is_master=(rank == 0 and not agent), timeout=timedelta(seconds=120))
    if rank == 0:
        store.set(key, bytes(payload))
        out = bytes(payload)
    else:
        out = bytes(store.get(key))
    # keep the store alive until every rank has read the key (rank 0 may be its server)
    store.add(key + "/seen", 1)
    if rank == 0:
        import time
        deadline = time.time() + 120
        while int(store.add(key + "/seen", 0)) < world and time.time() < deadline:
            time.sleep(0.01)
    return out


def agree_all(key: str, ok: bool, rank: int, world: int, timeout_s: float = 120.0) -> bool:
    """True iff EVERY rank passed ok = True (same out-of-band key-value socket as exchange_bytes; no collective library
    involved).  bench.py uses it so that all ranks take the same collective backend: a communicator that came up on some
    ranks only must not leave the others waiting in a different library."""
    import time
    from datetime import timedelta
    if world == 1:
        return bool(ok)
    key = _run_key(key)
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29531")), world,
                          is_master=(rank == 0 and not agent), timeout=timedelta(seconds=timeout_s))
    store.add(key + "/bad", 0 if ok else 1)
    store.add(key + "/n", 1)
    deadline = time.time() + timeout_s
    while int(store.add(key + "/n", 0)) < world:
        if time.time() > deadline:   # a rank never arrived: raise on every rank that notices, rather than a verdict some ranks do not share
            raise TimeoutError("agree_all(%s): %d of %d ranks after %.0f s" % (key, int(store.add(key + "/n", 0)), world, timeout_s))
        time.sleep(0.01)
    all_ok = int(store.add(key + "/bad", 0)) == 0
    store.add(key + "/seen", 1)                 # keep rank 0's server alive until everybody has read the verdict
    if rank == 0 and not agent:
        while int(store.add(key + "/seen", 0)) < world and time.time() < deadline:
            time.sleep(0.01)
    return all_ok


class H2RDist:
    """The same plumbing over libh2r's own RCCL exports (h2r_dist_*: SURVEY section 2 component C1 behind the C ABI) -- what a
    Rust prover service binds.  Only the 128-byte RCCL id travels out of band: here through a torch.distributed.TCPStore at
    MASTER_ADDR:MASTER_PORT (a key-value socket, no process group); a service uses its own launcher."""

    def __init__(self, chip, rank: int, world: int, local_rank: int):
        import ctypes
        from ._lib import check, lib
        self.chip, self.rank, self.world, self.local_rank = chip, rank, world, local_rank
        self._lib, self._check, self._ct = lib(), check, ctypes
        idb = (ctypes.c_uint8 * 128)()
        if rank == 0:
            check(self._lib.h2r_dist_unique_id(idb), "h2r_dist_unique_id")
        raw = exchange_bytes("h2r_dist_id", bytes(idb), rank, world)
        ctypes.memmove(idb, raw, 128)
        self._d = ctypes.c_void_p()
        check(self._lib.h2r_dist_init(chip._ctx, idb, rank, world, ctypes.byref(self._d)), "h2r_dist_init")
        self.initialised, self.backend = True, "h2r_dist (RCCL via the C ABI)"
        self._dev = torch.device("cuda", local_rank)

    def _stream(self):
        return self.chip._stream()

    def barrier(self):
        self._check(self._lib.h2r_dist_allreduce_max_f64(self._d, None, 0, self._stream()), "h2r_dist barrier")
        torch.cuda.synchronize(self._dev)

    def max_over_ranks(self, value: float) -> float:
        t = torch.tensor([value], dtype=torch.float64, device=self._dev)
        self._check(self._lib.h2r_dist_allreduce_max_f64(self._d, t.data_ptr(), 1, self._stream()), "h2r_dist_allreduce_max_f64")
        return float(t.item())

    def broadcast_ints(self, values: List[int]) -> List[int]:
        t = torch.tensor(values, dtype=torch.int64, device=self._dev)
        self._check(self._lib.h2r_dist_bcast(self._d, t.data_ptr(), t.numel() * 8, 0, self._stream()), "h2r_dist_bcast")
        return [int(v) for v in t.tolist()]

    def gather_to_rank0(self, shard: torch.Tensor, status: torch.Tensor = None, total: int = None):
        """shard: [elems, num_limbs] limbs.  Every rank receives every shard (all-gather); rank 0 returns the concatenation.
        status (uint8 [elems], optional) travels in the same group call: rank 0 then returns (results, status).  `total`: as DistEnv's."""
        return gather_with_total(self, H2RDist._gather_equal, shard, status, total)

    def _gather_equal(self, shard: torch.Tensor, status: torch.Tensor = None):
        shard = shard.contiguous()
        out = torch.empty((self.world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        st_all = torch.empty(self.world * shard.shape[0], dtype=torch.uint8, device=shard.device) if status is not None else None
        self._check(self._lib.h2r_dist_gather_results(self._d, shard.data_ptr(), status.contiguous().data_ptr() if status is not None else None,
                                                      shard.shape[0], out.data_ptr(), st_all.data_ptr() if st_all is not None else None,
                                                      self._stream()), "h2r_dist_gather_results")
        torch.cuda.synchronize(self._dev)
        if status is not None:
            return (out, st_all) if self.rank == 0 else (None, None)
        return out if self.rank == 0 else None

    def describe(self) -> dict:
        """What ran: ranks as the communicator reports them, and the RCCL the C ABI's collectives went through."""
        return {"ranks": int(self._lib.h2r_dist_world(self._d)), "rank0": int(self._lib.h2r_dist_rank(self._d)),
                "rccl_version": int(self._lib.h2r_dist_version()), "communicator": "h2r_dist_init on every rank (agreed out of band before the first collective)"}

    def finalize(self):
        if self._d:
            self.barrier()
            self._lib.h2r_dist_destroy(self._d)
            self._d = self._ct.c_void_p()
