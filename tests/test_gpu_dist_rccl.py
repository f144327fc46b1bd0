"""The one collective of the multi-GPU path (diffsol_amd/dist.py: the padded all-gather of the trajectories along the batch axis) on RCCL itself.  The GPU tier has ONE
GPU, so this is a one-rank `nccl` group: it cannot show scaling, but it runs torch.distributed's RCCL backend on device tensors of the library's layout through the
same call (all_gather_into_tensor on the flattened padded shard) that the N-rank job issues — initialisation, dtype / contiguity requirements, stream ordering with the
solver's output buffer.  The N > 1 logic (uneven shards, padding, reassembly) is covered by the gloo world-2/3 tests of the CPU tier (tests/test_dist_cpu.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_the_trajectory_gather_runs_on_rccl_with_device_tensors():
    import torch
    import torch.distributed as dist

    import diffsol_amd
    from bench import robertson_params, T_EVAL, RTOL, ATOL
    from diffsol_amd.dist import gather_batch_axis, gather_batch_axis_async

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1)
    try:
        nb = 1000
        s = diffsol_amd.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, device=0)
        out = torch.empty((len(T_EVAL), 3, nb), dtype=torch.float64, device="cuda:0")
        s.solve_dense(T_EVAL, want_host=False, dev_ptr=out.data_ptr())  # the solver writes on its own stream and returns after the launch completed
        g = gather_batch_axis(out, nb, 0, 1, force_collective=True)
        torch.cuda.synchronize()
        assert g.shape == out.shape and torch.equal(g, out)
        y_host, _ = s.solve_dense(T_EVAL)
        assert np.array_equal(np.transpose(g.cpu().numpy(), (0, 2, 1)), np.asarray(y_host))
        # the overlapped form of bench.py (N > 1): the gather of solve k is in flight on RCCL's stream while solve k + 1 runs on the solver's stream; two buffers in turn
        bufs = [out, torch.empty_like(out)]
        pend = [None, None]
        got = []
        for k in range(6):
            i = k % 2
            if pend[i] is not None:
                got.append(pend[i].finish())
            s.solve_dense(T_EVAL, want_host=False, dev_ptr=bufs[i].data_ptr())
            pend[i] = gather_batch_axis_async(bufs[i], nb, 0, 1)
        got += [pend[0].finish(), pend[1].finish()]
        assert len(got) == 6 and all(torch.equal(gk, g) for gk in got)
    finally:
        dist.destroy_process_group()
