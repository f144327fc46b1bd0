"""Multi-GPU ensembles: one process per GPU, contiguous shards of the ensemble, no data-path collective during integration.

The reference has no distributed layer (no NCCL/MPI/threads); systems of an ensemble are independent, so the ensemble
partitions trivially: rank g integrates systems [g*N/G, (g+1)*N/G) with its own lock-step (t, h, order) sequence on its own GPU
and stream.  The only exchange is the final trajectory collection: one all-gather (RCCL over xGMI; `gloo` in the CPU tests) of the
`solve_dense` output along the batch axis.  torch.distributed is used purely as the collective transport.
"""
import numpy as np


def shard_bounds(n_total, rank, world):
    """Contiguous range [lo, hi) of ensemble members owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def max_shard(n_total, world):
    return (int(n_total) + int(world) - 1) // int(world)


def gather_batch_axis(local, n_total, rank, world, group=None, force_collective=False):
    """All-gather `local` ([..., nb_local], batch-fastest like the device layout) along its last axis into [..., n_total] on every rank.

    Shards may differ in size by one: each rank pads to the maximum shard size, one all_gather_into_tensor moves the data, the padding is
    dropped on the way out.  `local` may be a CUDA tensor (RCCL) or a CPU tensor (gloo).  force_collective: issue the collective also for world == 1 (the
    GPU tier runs the RCCL call that way on the one GPU it has)."""
    import torch
    import torch.distributed as dist

    lo, hi = shard_bounds(n_total, rank, world)
    assert local.shape[-1] == hi - lo, (local.shape, lo, hi)
    if world == 1 and not force_collective:
        return local
    m = max_shard(n_total, world)
    lead = tuple(local.shape[:-1])
    padded = local.new_zeros(lead + (m,))
    padded[..., : hi - lo] = local
    flat_in = padded.contiguous().view(-1)
    flat_out = local.new_empty(world * flat_in.numel())  # 1-D in / 1-D out is accepted by both RCCL and gloo
    dist.all_gather_into_tensor(flat_out, flat_in, group=group)
    out = flat_out.view((world,) + lead + (m,))
    pieces = []
    for r in range(world):
        l, h = shard_bounds(n_total, r, world)
        pieces.append(out[r][..., : h - l])
    return torch.cat(pieces, dim=-1)


class PendingGather:
    """An all-gather of a shard's trajectories that has been issued but not waited for (gather_batch_axis_async).  `finish()` waits — for device tensors it makes
    the host wait too, because the solver that will overwrite the shard's buffer launches on its OWN stream, which no stream-level wait of torch orders — and
    returns the gathered [..., n_total] tensor (every rank the same), dropping the padding of uneven shards."""

    def __init__(self, work, flat_in, flat_out, lead, m, n_total, world):
        self._work, self._flat_in, self._flat_out = work, flat_in, flat_out
        self._lead, self._m, self._n_total, self._world = lead, m, n_total, world
        self._result = None

    def finish(self):
        import torch

        if self._result is None:
            self._work.wait()
            if self._flat_out.is_cuda:
                torch.cuda.current_stream(self._flat_out.device).synchronize()
            out = self._flat_out.view((self._world,) + self._lead + (self._m,))
            pieces = []
            for r in range(self._world):
                l, h = shard_bounds(self._n_total, r, self._world)
                pieces.append(out[r][..., : h - l])
            self._result = torch.cat(pieces, dim=-1)
            self._flat_in = None
        return self._result


def gather_batch_axis_async(local, n_total, rank, world, group=None):
    """gather_batch_axis without the wait: the collective is issued (async_op) and a PendingGather comes back, so that the next ensemble solve of this rank runs while
    the links move the last one's trajectories (bench.py: two output buffers in turn; a buffer is reused only after the gather that read it has finished).  `local` must
    not be written until `finish()` has returned."""
    import torch.distributed as dist

    lo, hi = shard_bounds(n_total, rank, world)
    assert local.shape[-1] == hi - lo, (local.shape, lo, hi)
    m = max_shard(n_total, world)
    lead = tuple(local.shape[:-1])
    if hi - lo == m and local.is_contiguous():
        flat_in = local.view(-1)  # even shards: the collective reads the solver's buffer itself
    else:
        padded = local.new_zeros(lead + (m,))
        padded[..., : hi - lo] = local
        flat_in = padded.contiguous().view(-1)
    flat_out = local.new_empty(world * flat_in.numel())
    work = dist.all_gather_into_tensor(flat_out, flat_in, group=group, async_op=True)
    return PendingGather(work, flat_in, flat_out, lead, m, n_total, world)


class CabiCommunicator:
    """The gather at the C ABI (include/diffsol_hip.h dsh_dist_*: librccl bound by the library itself, no torch.distributed in the data path): what a Rust / C caller
    of libdiffsol_hip.so uses.  `unique_id` (128 bytes) comes from rank 0's CabiCommunicator.unique_id() and reaches the other ranks by any side channel (here: the
    launcher's store, a file, an environment variable)."""

    def __init__(self, ctx_handle, rank, world, unique_id, owner=None):
        """owner: the Solver (or context object) `ctx_handle` came from — kept alive as long as the communicator, whose gathers run behind that context's stream"""
        import ctypes as C

        from . import _ffi
        self._owner = owner
        self._L = _ffi.load_device_lib()
        self._h = _ffi.vp()
        assert len(unique_id) == 128
        _ffi.check(self._L.dsh_dist_init(ctx_handle, int(rank), int(world), bytes(unique_id), C.byref(self._h)))
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id():
        import ctypes as C

        from . import _ffi
        buf = C.create_string_buffer(128)
        _ffi.check(_ffi.load_device_lib().dsh_dist_unique_id(buf))
        return buf.raw

    def gather(self, local, n_total, out=None, wait=True):
        """local: CUDA tensor [..., nb_local] (batch-fastest, contiguous) -> out [..., n_total] on every rank; wait=False leaves the gather in flight (call wait())"""
        import torch

        from . import _ffi
        assert local.is_cuda and local.is_contiguous() and local.dtype == torch.float64
        lead = int(np.prod(local.shape[:-1])) if local.dim() > 1 else 1
        if out is None:
            out = torch.empty(tuple(local.shape[:-1]) + (int(n_total),), dtype=torch.float64, device=local.device)
        f = self._L.dsh_gather_batch_axis if wait else self._L.dsh_gather_batch_axis_async
        _ffi.check(f(self._h, _ffi.vp(local.data_ptr()), lead, int(n_total), _ffi.vp(out.data_ptr())))
        self._keep = (local, out)
        return out

    def wait(self):
        from . import _ffi
        _ffi.check(self._L.dsh_gather_wait(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsh_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def solve_ensemble_sharded(model, params, t_eval, *, rank, world, device, method=0, model_size=0, gather=True, group=None, solver_factory=None,
                           resident=None, **solver_kw):
    """Integrate this rank's shard of the ensemble and (optionally) gather the interpolated trajectories.

    params: [n_total, nparams] on every rank (the parameter sweep is generated deterministically from a seed, so nothing is scattered).
    resident: None = host-driven lock-step solve_dense; 1 / 64 = the device-resident kernel with per-member / wavefront lock-step control
    (Solver.solve_dense_adaptive; `stats` are then the shard's totals).
    Returns (y, stats) with y a torch tensor [nt, nstates, n_total] if gather else [nt, nstates, nb_local] (batch-fastest)."""
    import torch

    params = np.asarray(params, dtype=np.float64)
    n_total = params.shape[0]
    lo, hi = shard_bounds(n_total, rank, world)
    if solver_factory is None:
        from .solver import Solver

        def solver_factory(p_local):
            return Solver(model, p_local, nbatch=p_local.shape[0], model_size=model_size, method=method, device=device, **solver_kw)

    s = solver_factory(params[lo:hi])
    nt = len(t_eval)
    on_gpu = torch.cuda.is_available() and not getattr(s, "cpu_stub", False)
    stats = None
    if on_gpu:
        out = torch.empty((nt, s.n, hi - lo), dtype=torch.float64, device=f"cuda:{device}")
        if resident:
            _, stats = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr(), group=resident)
        else:
            s.solve_dense(t_eval, want_host=False, dev_ptr=out.data_ptr())
        torch.cuda.synchronize(device)
    else:
        if resident:
            y_host, stats = s.solve_dense_adaptive(t_eval, group=resident)  # [nt, nb, n]
        else:
            y_host, _ = s.solve_dense(t_eval)
        out = torch.from_numpy(np.ascontiguousarray(np.transpose(y_host, (0, 2, 1))))
    if stats is None:
        stats = s.stats()
    if gather:
        out = gather_batch_axis(out, n_total, rank, world, group=group)
    return out, stats
