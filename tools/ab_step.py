"""A/B timing of the captured training step (batch 128 of 64x64) with engine switches flipped, in one process
and interleaved (A B A B ...) so that box-to-box and warm-up drift cancel.
usage: ab_step.py SWITCH[=value] [SWITCH[=value] ...]     e.g.  ab_step.py BN_BWD_ONE=1
Each argument names a module constant of reid_gan_amd.engine (or reid_gan_amd.train_step); A is the default tree, B has all of them set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd import engine, ops, train_step
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep

g, d, vgg = bench.synth_states()
x, y = synth_batch(128, 64, seed=1234)


def timed(reps=60):
    ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4, lookahead=LOOKAHEAD, use_graph="--eager" not in sys.argv)
    ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
    for _ in range(5):
        ts.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ts.step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    losses = ts.losses()
    del ts
    return ms, losses


LOOKAHEAD = False

if __name__ == "__main__":
    flips = {}
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    LOOKAHEAD = "--lookahead" in sys.argv           # both arms with the look-ahead plan
    ab_lookahead = "--ab-lookahead" in sys.argv      # A: plain plan, B: look-ahead plan
    for a in args:
        k, _, v = a.partition("=")
        mod = engine if hasattr(engine, k) else train_step
        flips[k] = type(getattr(mod, k))(int(v)) if v else (not getattr(mod, k))
    base = {k: getattr(engine if hasattr(engine, k) else train_step, k) for k in flips}
    res = {"A": [], "B": []}
    for rep in range(3):
        for name, vals in (("A", base), ("B", flips)):
            for k, v in vals.items():
                setattr(engine if hasattr(engine, k) else train_step, k, v)
            if ab_lookahead:
                LOOKAHEAD = name == "B"
            ms, losses = timed()
            res[name].append(ms)
            print(f"{name} {vals} {ms:.3f} ms  gen {losses['gen']:.4f}", flush=True)
    for name in res:
        print(name, " ".join(f"{v:.3f}" for v in res[name]), "min %.3f" % min(res[name]))
    print("barrier timeouts", ops.barrier_timeouts())
