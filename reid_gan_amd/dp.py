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


class Exchanger:
    """Mean-over-ranks all-reduces of the training step, in two forms:

      sync(t)              the result is needed at once (the two hinge means);
      issue(t) -> handle   t is complete on the current stream: start its all-reduce and return at once, so the
      wait(handle)         kernels launched next run while it is in flight; wait() makes the current stream wait
                           for the result (no host synchronisation).

    Backend "nccl" (RCCL over xGMI): issue() is an asynchronous collective on RCCL's own stream, ordered after the
    current stream by an event; the mean is ReduceOp.AVG where the library has it (else a sum and one scaling
    kernel in wait()).  Any other backend (gloo: CPU tests, and rehearsals with several ranks sharing one card,
    which RCCL refuses) goes through host memory: the stream is drained, the tensor copied out, reduced and copied
    back inside issue(); wait() is then a no-op.  Handles must be waited for in the order they were issued.
    A single-rank group still issues its collectives (that is how the multi-rank code path is exercised on one
    GPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.native = dist.get_backend(group) == "nccl"
        self.avg = False
        self.issued = 0            # collectives started so far (read by watchdogs)
        if self.native:
            try:                   # one tiny collective: does this RCCL build take ReduceOp.AVG?
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                self.avg = abs(t.item() - 1.0) < 1e-6
            except Exception:
                self.avg = False

    def _host_reduce(self, t: torch.Tensor):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        h = t.detach().to("cpu", copy=True)
        self.dist.all_reduce(h, group=self.group)
        if self.world > 1:
            h.mul_(1.0 / self.world)
        t.copy_(h)

    def issue(self, t: torch.Tensor):
        self.issued += 1
        if not self.native:
            self._host_reduce(t)
            return None
        op = self.dist.ReduceOp.AVG if self.avg else self.dist.ReduceOp.SUM
        return (self.dist.all_reduce(t, op=op, group=self.group, async_op=True), t)

    def wait(self, handle):
        if handle is None:
            return
        work, t = handle
        work.wait()                # the current stream waits for RCCL's stream; the host does not block
        if not self.avg and self.world > 1:
            t.mul_(1.0 / self.world)

    def sync(self, t: torch.Tensor):
        self.wait(self.issue(t))


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
