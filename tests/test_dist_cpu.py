"""Multi-process CPU test of the N>1 path: shard bounds + the batch-axis all-gather, world_size 2 and 3 over gloo
(the GPU path is the same code with backend nccl = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from diffsol_amd.dist import gather_batch_axis, gather_batch_axis_async, shard_bounds, solve_ensemble_sharded  # noqa: E402


def test_shard_bounds_partition_exactly():
    for n in (1, 2, 7, 100, 100_000, 262_144):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                covered.extend(range(lo, hi) if n <= 1000 else [lo, hi])
            if n <= 1000:
                assert covered == list(range(n))
            sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


class _StubSolver:
    """Stands in for diffsol_amd.Solver on a CPU-only box: analytic exponential decay 'trajectories' per parameter set."""
    cpu_stub = True

    def __init__(self, p):
        self.p = p
        self.n = 2
        self.nbatch = p.shape[0]

    def solve_dense(self, t_eval, want_host=True, dev_ptr=None):
        t = np.asarray(t_eval)[:, None, None]
        y = self.p[None, :, 1:2] * np.exp(-self.p[None, :, 0:1] * t) * np.ones((1, 1, 2))
        return y, 2

    def solve_dense_adaptive(self, t_eval, want_host=True, dev_ptr=None, group=1):
        y, _ = self.solve_dense(t_eval)
        return y, {"number_of_steps": 5 * self.nbatch * group, "number_of_nonlinear_solver_iterations": 9 * self.nbatch}

    def stats(self):
        return {"number_of_steps": 7, "number_of_nonlinear_solver_iterations": 11}


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        params = np.stack([rng.uniform(0.1, 1.0, n_total), rng.uniform(1.0, 2.0, n_total)], axis=1)
        t_eval = [0.0, 0.5, 1.0]
        y, stats = solve_ensemble_sharded("exponential_decay", params, t_eval, rank=rank, world=world, device=0, solver_factory=_StubSolver)
        ref = params[None, :, 1:2] * np.exp(-params[None, :, 0:1] * np.asarray(t_eval)[:, None, None]) * np.ones((1, 1, 2))
        ref = np.transpose(ref, (0, 2, 1))  # [nt, n, N] batch-fastest
        ok = tuple(y.shape) == (3, 2, n_total) and np.array_equal(y.numpy(), ref)
        # the device-resident modes go through the same shard / gather path
        lo_, hi_ = shard_bounds(n_total, rank, world)
        for g_ in (1, 64):
            y2, st2 = solve_ensemble_sharded("exponential_decay", params, t_eval, rank=rank, world=world, device=0, solver_factory=_StubSolver, resident=g_)
            ok = ok and np.array_equal(y2.numpy(), ref) and st2["number_of_steps"] == 5 * (hi_ - lo_) * g_
        # plain gather of a ragged last axis
        lo, hi = shard_bounds(n_total, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float64).repeat(4, 1)
        g = gather_batch_axis(local, n_total, rank, world)
        ok = ok and torch.equal(g, torch.arange(n_total, dtype=torch.float64).repeat(4, 1))
        # the overlapped form (bench.py, N > 1): several gathers in flight over two buffers, a buffer reused only after its gather has finished
        bufs = [torch.empty((3, 2, hi - lo), dtype=torch.float64) for _ in range(2)]
        pend = [None, None]
        got = []
        for k in range(5):
            i = k % 2
            if pend[i] is not None:
                got.append(pend[i].finish())
            bufs[i].copy_(torch.arange(lo, hi, dtype=torch.float64).repeat(3, 2, 1) + 1000.0 * k)  # "the solve of step k"
            pend[i] = gather_batch_axis_async(bufs[i], n_total, rank, world)
        for i in (5 % 2, 4 % 2):  # older first
            got.append(pend[i].finish())
        for k, gk in enumerate(got):
            ok = ok and tuple(gk.shape) == (3, 2, n_total) and torch.equal(gk, torch.arange(n_total, dtype=torch.float64).repeat(3, 2, 1) + 1000.0 * k)
        ok = ok and len(got) == 5 and got[4] is pend[0].finish()  # finish() is idempotent
        q.put((rank, bool(ok), stats["number_of_steps"]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 7), (3, 8)])
def test_sharded_solve_and_gather_over_gloo(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results), results


def _compile_rank(rank, cache_dir, q):
    os.environ["DSH_JIT_CACHE"] = cache_dir
    os.environ["WORLD_SIZE"] = "4"
    from diffsol_amd import _ffi, diffsl as fe
    import diffsl_models as D
    m = fe.DiffslModel(D.heat1d(24))
    m.precompile(fe.FAMILY_RESIDENT_BDF)
    q.put((rank, int(_ffi.load_device_lib().dsh_jit_compile_count())))
    m.release()


def test_ranks_that_meet_the_same_model_on_a_cold_cache_compile_it_once(tmp_path):
    """One process per GPU: four ranks start together, each builds the same DiffSL model against one cold on-disk cache (DSH_JIT_CACHE).  The
    entry lock of dsh_jit.hip makes ONE of them run hiprtc (seconds to minutes per kernel family) while the others wait and load its code object:
    dsh_jit_compile_count sums to the number of distinct modules, not to four times that; every rank ends with the module loaded."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_compile_rank, args=(r, str(tmp_path), q)) for r in range(4)]
    for p in procs: p.start()
    for p in procs: p.join(600)
    assert all(p.exitcode == 0 for p in procs)
    counts = dict(q.get(timeout=10) for _ in range(4))
    files = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(files) >= 1 and not [f for f in os.listdir(tmp_path) if ".tmp" in f or f.endswith(".lock")]
    assert sum(counts.values()) == len(files), (counts, files)


def test_c_abi_shard_bounds_equal_the_python_partition():
    """dsh_dist_shard_bounds (include/diffsol_hip.h; what a Rust / C caller shards with) is the partition of diffsol_amd.dist.shard_bounds: contiguous, sizes differ
    by at most one, covers [0, n_total) — no GPU needed for the index arithmetic."""
    import ctypes as C

    import __graft_entry__
    __graft_entry__.build()
    from diffsol_amd import _ffi
    from diffsol_amd.dist import shard_bounds
    L = _ffi.load_device_lib()
    for n_total in (0, 1, 2, 7, 8, 9, 100_000, 262_144, 262_145):
        for world in (1, 2, 3, 4, 7, 8):
            prev = 0
            for rank in range(world):
                lo, hi = C.c_int64(), C.c_int64()
                assert L.dsh_dist_shard_bounds(n_total, rank, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_bounds(n_total, rank, world)
                assert lo.value == prev
                prev = hi.value
            assert prev == n_total
    lo, hi = C.c_int64(), C.c_int64()
    assert L.dsh_dist_shard_bounds(10, 3, 3, C.byref(lo), C.byref(hi)) < 0  # rank out of range
