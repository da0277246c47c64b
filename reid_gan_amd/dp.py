"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" is RCCL on ROCm, over xGMI).

The hot path has exactly three kinds of exchange per step (see train_step.py): the two hinge means of DLoss
(8 bytes, before the non-linear hinge), the discriminator's flat gradient buffer (twice) and the generator's
flat gradient buffer (once).  Because each network's gradients live in ONE flat fp32 buffer, each exchange is a
single all-reduce of 17-18 MB, large enough to run RCCL's direct (all-links) algorithms rather than many
latency-bound per-tensor calls.
"""
from __future__ import annotations

import os
from typing import Optional

import torch


def allreduce_mean(t: torch.Tensor, group=None, world: Optional[int] = None) -> None:
    """In place: t <- mean over ranks of t.  No-op for a single rank."""
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(t, group=group)
        t.mul_(1.0 / world)


def init_from_env(backend: str = "nccl"):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, local_rank, world)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def hinge_terms(mean_pos: torch.Tensor, mean_neg: torch.Tensor, n_per_half: int):
    """DLoss on batch means (src/loss.py:141-142) and its per-element gradients, as the device kernel computes
    them: loss = relu(1 - mean_pos) + relu(1 + mean_neg); d/dpos_i = -[1 - mean_pos > 0]/n; d/dneg_i = +[...]/n.
    With data parallelism the means must be the GLOBAL means (all-reduced) and n the per-rank count: averaging
    the resulting per-rank gradients over ranks then equals the single-process gradient at the global batch."""
    a, b = 1.0 - mean_pos, 1.0 + mean_neg
    loss = torch.clamp(a, min=0) + torch.clamp(b, min=0)
    cpos = torch.where(a > 0, -1.0 / n_per_half, 0.0)
    cneg = torch.where(b > 0, 1.0 / n_per_half, 0.0)
    return loss, cpos, cneg
