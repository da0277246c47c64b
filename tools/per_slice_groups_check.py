"""How many per-image BatchNorm finalizations of a per-slice-statistics plan read a sibling group's shared rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import engine, ops
from reid_gan_amd.synth import synth_state, synth_batch
from oracle import ried_oracle as O
calls = {"shared": 0, "own": 0}
real = ops.bn_finalize_images


def counting(*a, **kw):
    calls["shared" if kw.get("ldstat", 0) > 0 else "own"] += 1
    return real(*a, **kw)


ops.bn_finalize_images = counting
st = engine.State(synth_state([(n, s) for n, s, _ in O.rigan_spec()], 101), "cuda")
G = engine.Generator(st, 2, 256, 256, grads=False, per_image_stats=True)
x, _ = synth_batch(2, 256, seed=3)
out = G.forward(x.cuda())
torch.cuda.synchronize()
print(calls, float(out.abs().mean()))
