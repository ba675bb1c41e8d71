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


def make_cost_exchange(n_total: int, group=None, device=None):
    """Callback for BatchSolver.set_cost_exchange: all-gather of the shards' per-agent costs, then the serial sum
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


def all_gather_solution(solver, n_total: int, group=None, device=None):
    """Every rank receives the whole Solution (solution.hpp:9-15): states [n_total, T+1, nx], controls
    [n_total, T, nu] and costs [n_total] in block order -- the publish step of the Nash strategies
    (`update_initial_with_best`, nash.hpp:66-71) for callers whose agents are coupled through each other's
    trajectories.  The shard leaves the solver through the device-pointer entry point (no host copy on a GPU) and
    travels as ONE all-gather per array over RCCL/xGMI; shards are padded to equal size because an all-gather wants
    equal shapes.  Returns torch tensors on `device`."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    spans = [shard_range(n_total, world, r) for r in range(world)]
    width = max(hi - lo for lo, hi in spans)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n_local = spans[rank][1] - spans[rank][0]
    assert n_local == solver.N
    X = torch.zeros((width, solver.T + 1, solver.nx), dtype=torch.float64, device=device)
    U = torch.zeros((width, solver.T, solver.nu), dtype=torch.float64, device=device)
    cost = torch.zeros(width, dtype=torch.float64, device=device)
    if device.type == "cuda":
        # stream-ordering contract of the *_device entry points (include/mas_hip.h): the handle works on its OWN stream, so the
        # zero-fills torch just enqueued on its current stream must have finished before the solver writes into the buffers;
        # on return from solution_device the results are complete (the call synchronises the handle's stream)
        torch.cuda.current_stream(device).synchronize()
        solver.solution_device(X.data_ptr(), U.data_ptr(), cost.data_ptr())       # fills the first n_local agents
    else:
        # host tensors (gloo): through the host entry point -- a device handle must never be handed a host pointer as a device one
        sol = solver.solution()
        X[:n_local] = torch.from_numpy(sol["X"]); U[:n_local] = torch.from_numpy(sol["U"]); cost[:n_local] = torch.from_numpy(sol["cost"])
    out = {}
    for name, mine in (("X", X), ("U", U), ("cost", cost)):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out[name] = torch.cat([part[:hi - lo] for (lo, hi), part in zip(spans, parts)])
    return out
