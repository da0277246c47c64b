"""Child processes of the multi-rank GPU tests (not collected by pytest: no test_ prefix).

The data-parallel step needs one process per rank, and a process group must not be created and destroyed inside
the long-lived pytest process (RCCL's helper threads would live beside every later test).  So the ranks run here,
in fresh processes:

    python tests/dp_child.py launcher OUTDIR        starts the rank processes of every scenario, one scenario after
                                                    the other, each under a time limit; never touches the GPU itself
    python tests/dp_child.py rank SCENARIO RANK WORLD PORT OUTDIR

tests/conftest.py starts the launcher when the session starts (before the pytest process has made any GPU call),
and tests/test_g8_multirank_gpu.py waits for it and checks what the ranks wrote: OUTDIR/<scenario>.rank<r>.npz
(results) and OUTDIR/<scenario>.rank<r>.log (output).  A rank that fails writes OUTDIR/<scenario>.rank<r>.err.

Scenarios (all on cuda:0; two ranks share the one card over gloo, which RCCL refuses to do):
  onerank_rccl    a group of one rank over RCCL: the segmented graphs with real all-reduces between them against the
                  single-graph step of the same process
  tworank         two ranks, four patches each, two steps on two batches: TrainStep(process_group=WORLD), captured
                  graph segments, per-replica BatchNorm statistics
  tworank_syncbn  the same with sync_bn=True (eager): equals a single process at the global batch of eight
"""
import json
import os
import socket
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENARIOS = (("onerank_rccl", 1), ("tworank", 2), ("tworank_syncbn", 2))
B_RANK, H = 4, 64
SEEDS = (1234, 4321)           # the two global batches
LIMIT_S = 420                  # per scenario


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher(outdir: str) -> int:
    os.makedirs(outdir, exist_ok=True)
    status = {}
    for name, world in SCENARIOS:
        port = free_port()
        procs = []
        for r in range(world):
            log = open(os.path.join(outdir, f"{name}.rank{r}.log"), "w")
            procs.append((subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), "rank", name, str(r),
                                            str(world), str(port), outdir], stdout=log, stderr=subprocess.STDOUT,
                                           cwd=ROOT), log))
        deadline = time.time() + LIMIT_S
        codes = []
        for p, log in procs:
            try:
                codes.append(p.wait(timeout=max(1.0, deadline - time.time())))
            except subprocess.TimeoutExpired:
                p.kill()                     # this exact child
                p.wait()
                codes.append("timeout")
            log.close()
        status[name] = codes
        with open(os.path.join(outdir, "status.json"), "w") as f:
            json.dump(status, f)
    with open(os.path.join(outdir, "done"), "w") as f:
        f.write("done\n")
    return 0


def states():
    from reid_gan_amd import rinet
    from reid_gan_amd.synth import synth_state, synth_vgg
    gsd = synth_state(rinet.state_spec(rinet.RIGAN()), 101)
    dsd = synth_state(rinet.state_spec(rinet.ImageDiscriminator()), 202)
    return gsd, dsd, synth_vgg()


def global_batches(world: int):
    from reid_gan_amd.synth import synth_batch
    return [synth_batch(world * B_RANK, H, seed=s) for s in SEEDS]


def run_rank(name: str, rank: int, world: int, port: int, outdir: str):
    import numpy as np
    import torch
    import torch.distributed as dist
    from reid_gan_amd import engine, ops
    from reid_gan_amd.train_step import TrainStep
    ops.install_crash_backtrace(os.path.join(outdir, f"{name}.rank{rank}.crash"))
    torch.cuda.set_device(0)
    backend = "nccl" if name == "onerank_rccl" else "gloo"
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gsd, dsd, vgg = states()
    out = {}
    if name == "onerank_rccl":
        from reid_gan_amd.synth import synth_batch
        engine.BN_FOLD = False               # fixed-order reductions: the two plans must agree closely
        x, y = synth_batch(B_RANK, H, seed=98)
        one = TrainStep(gsd, dsd, vgg, B_RANK, H, H, lr=1e-4, use_graph=True)
        par = TrainStep(gsd, dsd, vgg, B_RANK, H, H, lr=1e-4, use_graph=True, process_group=dist.group.WORLD)
        for ts in (one, par):
            ts.step(x.cuda(), y.cuda())
            ts.step()
        torch.cuda.synchronize()
        out["n_graphs_par"], out["n_graphs_one"] = len(par.graphs), len(one.graphs)
        for tag, ts in (("one", one), ("par", par)):
            for k, v in ts.losses().items():
                out[f"{tag}.loss.{k}"] = v
            out[f"{tag}.g_flat"] = ts.gs.flat.cpu().numpy()
            out[f"{tag}.d_flat"] = ts.ds.flat.cpu().numpy()
        one.close(); par.close()
    else:
        sync = name == "tworank_syncbn"
        ts = TrainStep(gsd, dsd, vgg, B_RANK, H, H, lr=1e-4, use_graph=not sync, process_group=dist.group.WORLD,
                       sync_bn=sync)
        assert ts.world == world and ts.sync_bn == sync
        lo, hi = rank * B_RANK, (rank + 1) * B_RANK
        for i, (x, y) in enumerate(global_batches(world)):
            ts.step(x[lo:hi].cuda(), y[lo:hi].cuda())
            torch.cuda.synchronize()
            for k, v in ts.losses().items():
                out[f"step{i}.loss.{k}"] = v
            out[f"step{i}.pred"] = ts.G.out.cpu().numpy()
            if rank == 0 and i == 0:      # the averaged gradients of the first step (identical on every rank)
                out["step0.g_grad"] = ts.gs.flat_grad.cpu().numpy()
                out["step0.d_grad"] = ts.ds.flat_grad.cpu().numpy()
        out["n_graphs"] = 0 if ts.graphs is None else len(ts.graphs)
        out["g_flat"] = ts.gs.flat.cpu().numpy()
        out["d_flat"] = ts.ds.flat.cpu().numpy()
        out["g_names"] = np.array(list(ts.gs.offs.keys()))
        out["g_offs"] = np.array(list(ts.gs.offs.values()))
        out["d_names"] = np.array(list(ts.ds.offs.keys()))
        out["d_offs"] = np.array(list(ts.ds.offs.values()))
        out["bn_rm"] = ts.gs.b["conv2.bn.running_mean"].cpu().numpy()
        out["exchanges"] = ts.xch.issued
        ts.close()
    dist.barrier()
    dist.destroy_process_group()
    np.savez(os.path.join(outdir, f"{name}.rank{rank}.npz"), **out)


def main(argv):
    if argv[1] == "launcher":
        return launcher(argv[2])
    name, rank, world, port, outdir = argv[2], int(argv[3]), int(argv[4]), int(argv[5]), argv[6]
    try:
        run_rank(name, rank, world, port, outdir)
    except BaseException:
        with open(os.path.join(outdir, f"{name}.rank{rank}.err"), "w") as f:
            f.write(traceback.format_exc())
        traceback.print_exc()
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
