"""Agents sharded over ranks: one process and one BatchSolver per GPU under torch.distributed (RCCL on the GPUs,
gloo in the CPU tests).  SURVEY.md section 8(e): the id-sorted block list (multi_agent_problem.hpp:41-49) is cut
into contiguous ranges; the sequential and trust-region strategies exchange nothing; the line-search strategy
needs the total cost over all agents (nash.hpp:45-49,119,134-143), which `make_cost_exchange` provides."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous id range [lo, hi) of `rank`; sizes differ by at most one agent."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def use_library_cost_exchange(solver, n_total: int, group=None):
    """line-search strategy on GPUs: the total cost over all ranks' agents is formed INSIDE the library (RCCL all-gather of the
    shards' costs + one wavefront's serial sum in global block order, mas_hip_set_comm_cost_exchange) -- no cost leaves the device."""
    solver.set_comm_cost_exchange(library_comm(group), n_total)


def make_cost_exchange(n_total: int, group=None, device=None):
    """(host tensors / gloo; on GPUs prefer use_library_cost_exchange)  Callback for BatchSolver.set_cost_exchange: all-gather of the shards' per-agent costs, then the serial sum
    in global block order -- bit-for-bit the total a single process holding all agents computes, so every rank
    takes the same accept / reject decisions.  One all-gather of n_total doubles per call (32 KiB at 4096 agents)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [shard_range(n_total, world, r) for r in range(world)]
    width = max(hi - lo for lo, hi in spans)             # all_gather wants equal shapes: pad the short shards
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def exchange(local_costs: np.ndarray) -> float:
        lo, hi = spans[rank]
        assert local_costs.shape == (hi - lo,)
        mine = torch.zeros(width, dtype=torch.float64)
        mine[:hi - lo] = torch.from_numpy(local_costs)
        mine = mine.to(device)
        parts = [torch.empty(width, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        total = 0.0
        for (l, h), part in zip(spans, parts):           # detail::compute_total_cost: c += cost, in block order
            for v in part[:h - l].cpu().tolist():
                total += v
        return total

    return exchange


_COMMS = {}


def library_comm(group=None, device_index=None):
    """The library's own RCCL communicator over the ranks of a torch.distributed group (include/mas_hip.h: mas_hip_comm_*): rank 0
    makes the unique id, torch.distributed carries its 128 bytes to the others, every rank joins.  One per (group, device)."""
    import torch
    import torch.distributed as dist
    from .batch import Comm
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if device_index is None:
        device_index = torch.cuda.current_device()
    key = (id(group), device_index)
    if key not in _COMMS:
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        _COMMS[key] = Comm(box[0], world, rank, device_index)
    return _COMMS[key]


def all_gather_solution(solver, n_total: int, group=None, device=None):
    """Every rank receives the whole Solution (solution.hpp:9-15): states [n_total, T+1, nx], controls
    [n_total, T, nu] and costs [n_total] in block order -- the publish step of the Nash strategies
    (`update_initial_with_best`, nash.hpp:66-71) for callers whose agents are coupled through each other's
    trajectories.  On GPUs this is ONE call into the library (mas_hip_all_gather_solution): the shard is converted to the
    reference layout on the device straight to its displacement in the contiguous result and travels as one RCCL collective
    per array (uneven shards by displacement, nothing padded, no torch.cat).  On host tensors (gloo, the CPU tests) the
    shards travel through torch.distributed.  Returns torch tensors on `device`."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [shard_range(n_total, world, r) for r in range(world)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n_local = spans[rank][1] - spans[rank][0]
    assert n_local == solver.N
    if device.type == "cuda":
        comm = library_comm(group, device.index if device.index is not None else torch.cuda.current_device())
        X = torch.empty((n_total, solver.T + 1, solver.nx), dtype=torch.float64, device=device)
        U = torch.empty((n_total, solver.T, solver.nu), dtype=torch.float64, device=device)
        cost = torch.empty(n_total, dtype=torch.float64, device=device)
        # stream-ordering contract of the *_device entry points (include/mas_hip.h): the handle works on its OWN stream; torch has
        # nothing pending on these fresh buffers (torch.empty enqueues no kernel), and the call synchronises the handle's stream
        torch.cuda.current_stream(device).synchronize()
        solver.all_gather_solution_device(comm, n_total, X.data_ptr(), U.data_ptr(), cost.data_ptr())
        return {"X": X, "U": U, "cost": cost}
    # host tensors (gloo): through the host entry point -- a device handle must never be handed a host pointer as a device one
    width = max(hi - lo for lo, hi in spans)
    X = torch.zeros((width, solver.T + 1, solver.nx), dtype=torch.float64)
    U = torch.zeros((width, solver.T, solver.nu), dtype=torch.float64)
    cost = torch.zeros(width, dtype=torch.float64)
    sol = solver.solution()
    X[:n_local] = torch.from_numpy(sol["X"]); U[:n_local] = torch.from_numpy(sol["U"]); cost[:n_local] = torch.from_numpy(sol["cost"])
    out = {}
    for name, mine in (("X", X), ("U", U), ("cost", cost)):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out[name] = torch.cat([part[:hi - lo] for (lo, hi), part in zip(spans, parts)])
    return out
