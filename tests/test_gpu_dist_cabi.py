"""Multi-GPU trajectory collection at the C ABI (include/diffsol_hip.h dsh_dist_*, csrc/dsh_dist.hip: librccl bound by the library itself).  The GPU tier has ONE GPU:
  * the communicator and the gather run for real at world = 1 (ncclCommInitRank, ncclAllGather on the communicator's own stream behind the solver's stream, async form
    with two buffers in turn);
  * the two layout copies — everything of the N > 1 path that is not the RCCL call — are driven with synthetic buffers for world = 2, 3, 8 and uneven shards against
    numpy (the same index arithmetic the gloo tests of the CPU tier check for the Python path);
  * the REAL solver runs under two processes at once on the one GPU (shards of one ensemble, gathered over gloo on the host): stream / context / JIT-cache mistakes
    that a one-process tier cannot see, and the sharded result must equal the one-process result bit for bit."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_gather_on_rccl_world_one_blocking_and_overlapped():
    import torch

    import diffsol_amd
    from bench import ATOL, RTOL, T_EVAL, robertson_params
    from diffsol_amd.dist import CabiCommunicator

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    nb = 1000
    s = diffsol_amd.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, device=0)
    comm = CabiCommunicator(s.context_handle(), 0, 1, CabiCommunicator.unique_id())
    try:
        bufs = [torch.empty((len(T_EVAL), 3, nb), dtype=torch.float64, device="cuda:0") for _ in range(2)]
        outs = [torch.zeros_like(bufs[0]) for _ in range(2)]
        s.solve_dense(T_EVAL, want_host=False, dev_ptr=bufs[0].data_ptr())
        g = comm.gather(bufs[0], nb, out=outs[0])
        assert torch.equal(g, bufs[0])
        y_host, _ = s.solve_dense(T_EVAL)
        assert np.array_equal(np.transpose(g.cpu().numpy(), (0, 2, 1)), np.asarray(y_host))
        # overlapped: the gather of solve k in flight on the communicator's stream while solve k + 1 runs on the solver's stream
        for k in range(6):
            i = k % 2
            outs[i].zero_()
            s.solve_dense(T_EVAL, want_host=False, dev_ptr=bufs[i].data_ptr())
            comm.gather(bufs[i], nb, out=outs[i], wait=False)
            comm.wait()
            assert torch.equal(outs[i], g)
    finally:
        comm.close()


@pytest.mark.parametrize("world,n_total,lead", [(2, 11, 5), (3, 10, 4), (8, 262_144 // 64 + 5, 7), (4, 3, 2), (1, 9, 3)])
def test_pack_and_unpack_reassemble_uneven_shards_for_worlds_the_box_does_not_have(world, n_total, lead):
    import torch

    from diffsol_amd import _ffi
    from diffsol_amd.dist import max_shard, shard_bounds
    from diffsol_amd.la import HipContext

    L = _ffi.load_device_lib()
    ctx = HipContext(device=0)
    rng = np.random.default_rng(world * 1000 + n_total)
    full = rng.standard_normal((lead, n_total))
    m = max_shard(n_total, world)
    recv = torch.empty((world, lead, m), dtype=torch.float64, device="cuda:0")
    for r in range(world):  # what ncclAllGather delivers: every rank's padded shard, rank-major
        lo, hi = shard_bounds(n_total, r, world)
        local = torch.from_numpy(np.ascontiguousarray(full[:, lo:hi])).cuda()
        torch.cuda.synchronize()
        _ffi.check(L.dsh_dist_pack_shard(ctx._h, None, _ffi.vp(local.data_ptr()), lead, hi - lo, m, _ffi.vp(recv[r].data_ptr())))
        ctx.sync()
        packed = recv[r].cpu().numpy()
        assert np.array_equal(packed[:, : hi - lo], full[:, lo:hi]) and not packed[:, hi - lo:].any()
    out = torch.full((lead, n_total), np.nan, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()  # torch fills on ITS stream; the library's stream does not wait for it
    _ffi.check(L.dsh_dist_unpack_gathered(ctx._h, None, _ffi.vp(recv.data_ptr()), lead, n_total, world, _ffi.vp(out.data_ptr())))
    ctx.sync()
    assert np.array_equal(out.cpu().numpy(), full)


_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import diffsol_amd
from diffsol_amd.dist import shard_bounds, solve_ensemble_sharded
from bench import ATOL, RTOL, T_EVAL, robertson_params
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_total = 3001
p = robertson_params(n_total)
# every rank on the ONE GPU of the box: two processes, two contexts, two streams, at the same time
y, stats = solve_ensemble_sharded("robertson_ode", p, T_EVAL, rank=rank, world=world, device=0, model_size=1, rtol=RTOL, atol=ATOL, gather=False)
lo, hi = shard_bounds(n_total, rank, world)
assert y.shape[-1] == hi - lo
from diffsol_amd.dist import gather_batch_axis
g = gather_batch_axis(y.cpu(), n_total, rank, world)
if rank == 0:
    np.save(sys.argv[2], g.numpy())
dist.barrier()
dist.destroy_process_group()
"""


def test_the_real_solver_under_two_ranks_on_one_gpu_equals_the_one_process_shards(tmp_path):
    import diffsol_amd
    from bench import ATOL, RTOL, T_EVAL, robertson_params
    from diffsol_amd.dist import shard_bounds

    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = tmp_path / "gathered.npy"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29719",
                        str(script), ROOT, str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g = np.load(out)  # [nt, n, n_total]
    n_total = g.shape[-1]
    p = robertson_params(n_total)
    # each shard is its own default-mode ensemble (wavefront lock-step groups start at the shard's first member): compare with one-process solves of the same shards
    for rank in range(2):
        lo, hi = shard_bounds(n_total, rank, 2)
        s = diffsol_amd.Solver("robertson_ode", p[lo:hi], nbatch=hi - lo, model_size=1, rtol=RTOL, atol=ATOL, device=0)
        y, _ = s.solve_dense(T_EVAL)  # [nt, nb, n]
        assert np.array_equal(np.transpose(g[:, :, lo:hi], (0, 2, 1)), y)
