"""Data-parallel reduction logic on CPU: two gloo ranks, each with half of the global batch, must reproduce the
single-process discriminator loss and gradients of the oracle at the global batch (the hinge of src/loss.py:141-143
acts on the batch MEAN, so the means are exchanged before the hinge, not the losses after it).
Tolerance: fp32 summation order only (1e-5 relative)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import ried_oracle as O
from reid_gan_amd import dp, rinet
from reid_gan_amd.synth import synth_batch, synth_state

B, H = 4, 32            # global batch; 2 per rank


def _d_state():
    return synth_state(rinet.state_spec(rinet.ImageDiscriminator()), 202)


def _forward(sd, y, pred):
    """D on cat[y, pred] with fresh buffers (power iteration from the same u) -> (pos, neg, leaf params)."""
    leaves = {k: (v.clone().requires_grad_(True) if O.trainable(k) else v.clone()) for k, v in sd.items()}
    out = O.discriminator_forward(O.Params(leaves), torch.cat([y, pred], 0), training=True)
    n = y.shape[0]
    return out[:n], out[n:], leaves


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, _, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    import torch.distributed as dist
    sd = _d_state()
    x, y = synth_batch(B, H, seed=1234)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    pos, neg, leaves = _forward(sd, y[lo:hi], x[lo:hi])
    means = torch.stack([pos.mean().detach(), neg.mean().detach()])
    dp.allreduce_mean(means)                                   # the 8-byte exchange before the hinge
    loss, cpos, cneg = dp.hinge_terms(means[0], means[1], pos.numel())
    names = [k for k in leaves if leaves[k].requires_grad]
    grads = torch.autograd.grad([pos, neg], [leaves[k] for k in names],
                                [torch.full_like(pos, cpos.item()), torch.full_like(neg, cneg.item())],
                                allow_unused=True)
    flat = torch.cat([(g if g is not None else torch.zeros_like(leaves[k])).flatten() for k, g in zip(names, grads)])
    dp.allreduce_mean(flat)                                    # the one flat-gradient exchange
    if rank == 0:
        q.put((loss.item(), flat.numpy(), names))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_hinge_and_gradient_equal_global_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    loss2, flat2, names = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, global batch
    sd = _d_state()
    x, y = synth_batch(B, H, seed=1234)
    pos, neg, leaves = _forward(sd, y, x)
    loss = O.d_loss(pos, neg)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    flat = torch.cat([(g if g is not None else torch.zeros_like(leaves[k])).flatten() for k, g in zip(names, grads)])
    assert abs(loss2 - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-7
    err = np.abs(flat2 - flat.numpy()).max()
    assert err <= 1e-5 * np.abs(flat.numpy()).max() + 1e-9, err


def _xch_worker(rank, world, port, q):
    """The exchange protocol of the training step (dp.Exchanger) between two gloo ranks: a blocking exchange, then
    four all-reduces started back to back over slices of one flat buffer (the discriminator's gradient and the
    generator's three buckets, last bucket first) and awaited in issue order; then the generator-loss terms of
    src/loss.py:177-197 from per-rank shards - plain means over equal shards, so the mean over ranks of the per-rank
    values must equal the value at the global batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dp.init_from_env(backend="gloo")
    import torch.distributed as dist
    from reid_gan_amd.synth import synth_vgg
    xch = dp.Exchanger(dist.group.WORLD)
    assert xch.world == world and not xch.native
    means = torch.tensor([1.0 + rank, -2.0 * rank])
    xch.sync(means)
    g = torch.Generator().manual_seed(10 + rank)
    d_flat, g_flat = torch.randn(1000, generator=g), torch.randn(3000, generator=g)
    ranges = [(0, 700), (700, 2200), (2200, 3000)]
    handles = [("d", xch.issue(d_flat))]
    for b in (2, 1, 0):
        lo, hi = ranges[b]
        handles.append((b, xch.issue(g_flat[lo:hi])))
    for _, h in handles:
        xch.wait(h)
    assert xch.issued == 5
    # generator-loss terms on this rank's shard
    x, y = synth_batch(B, H, seed=77)
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    d_out = torch.randn(B, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        total, vl, sl, gl = O.compound_loss(synth_vgg(), x[lo:hi], y[lo:hi], d_out[lo:hi])
    terms = torch.stack([total, vl, sl, gl])
    xch.sync(terms)
    if rank == 0:
        q.put((means.numpy(), d_flat.numpy(), g_flat.numpy(), terms.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchanger_protocol_and_generator_loss_terms_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_xch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    means, d_flat, g_flat, terms = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(means, [1.5, -1.0])
    gens = [torch.Generator().manual_seed(10 + r) for r in range(2)]
    ds, gs = zip(*[(torch.randn(1000, generator=g), torch.randn(3000, generator=g)) for g in gens])
    assert np.allclose(d_flat, ((ds[0] + ds[1]) / 2).numpy(), atol=1e-6)
    assert np.allclose(g_flat, ((gs[0] + gs[1]) / 2).numpy(), atol=1e-6)
    from reid_gan_amd.synth import synth_vgg
    x, y = synth_batch(B, H, seed=77)
    d_out = torch.randn(B, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = torch.stack(O.compound_loss(synth_vgg(), x, y, d_out)).numpy()
    assert np.allclose(terms, ref, rtol=2e-5, atol=1e-7), (terms, ref)


def test_batchnorm_statistics_stay_per_replica():
    """What each rank computes in a data-parallel step is the generator applied to ITS shard: BatchNorm normalises
    with the shard's statistics (as the reference's commented nn.DataParallel would, src/train.py:95-96), so a rank's
    prediction equals the single-process prediction of that shard alone, not a slice of the global-batch prediction."""
    gsd = synth_state(rinet.state_spec(rinet.RIGAN()), 101)
    x, _ = synth_batch(4, 32, seed=3)
    x[2:] = x[2:] * 0.5 + 200.0                      # shards with different statistics
    with torch.no_grad():
        whole = O.rigan_forward(O.Params({k: v.clone() for k, v in gsd.items()}), x, True)
        shard = O.rigan_forward(O.Params({k: v.clone() for k, v in gsd.items()}), x[:2], True)
        again = O.rigan_forward(O.Params({k: v.clone() for k, v in gsd.items()}), x[:2].clone(), True)
    assert torch.equal(shard, again)
    assert (whole[:2] - shard).abs().mean() > 1e-3 * shard.abs().mean()


def test_allreduce_mean_single_rank_is_identity():
    t = torch.arange(5.0)
    dp.allreduce_mean(t)
    assert torch.equal(t, torch.arange(5.0))


def test_hinge_terms_inactive_sides():
    loss, cpos, cneg = dp.hinge_terms(torch.tensor(1.5), torch.tensor(-2.0), 10)
    assert loss.item() == 0 and cpos.item() == 0 and cneg.item() == 0
    loss, cpos, cneg = dp.hinge_terms(torch.tensor(0.25), torch.tensor(0.5), 10)
    assert abs(loss.item() - 2.25) < 1e-6 and abs(cpos.item() + 0.1) < 1e-7 and abs(cneg.item() - 0.1) < 1e-7
