"""Per-step losses of three consecutive training steps: HIP engine vs the oracle (bf16 emulation and fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ried_oracle as O
from reid_gan_amd import rinet
from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep

gsd = synth_state(rinet.state_spec(rinet.RIGAN()), 101)
dsd = synth_state(rinet.state_spec(rinet.ImageDiscriminator()), 202)
vgg = synth_vgg()
B, H = 4, 64
x, y = synth_batch(B, H, seed=1234)
ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
oe = O.TrainState(gsd, dsd, vgg, lr=1e-4)
of = O.TrainState(gsd, dsd, vgg, lr=1e-4)
keys = ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen")
for s in range(3):
    ts.step(x.cuda(), y.cuda())
    got = ts.losses()
    O.set_bf16_emulation(True)
    re_ = O.train_step(oe, x, y)
    O.set_bf16_emulation(False)
    rf = O.train_step(of, x, y)
    print(f"step {s + 1}")
    for k in keys:
        print(f"  {k:7s} hip {got[k]:.5f}  emu {re_[k]:.5f}  fp32 {rf[k]:.5f}")
