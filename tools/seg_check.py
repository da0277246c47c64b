"""Segmented capture against the single graph: largest weight differences after two steps (deterministic reductions)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from reid_gan_amd import engine
from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep
from oracle import ried_oracle as O
engine.BN_FOLD = False
B, H = 4, 64
gsd = synth_state([(n, s) for n, s, _ in O.rigan_spec()], 101)
dsd = synth_state([(n, s) for n, s, _ in O.discriminator_spec()], 202)
vgg = synth_vgg()
x, y = synth_batch(B, H, seed=99)
for trial in range(2):
    runs = []
    for segd in (False, True, False):
        ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
        ts.segmented = segd
        ts.step(x.cuda(), y.cuda()); ts.step()
        torch.cuda.synchronize()
        runs.append(ts)
    a, b, c = runs
    print("trial", trial, "segments", len(b.graphs),
          "one vs seg: G", (a.gs.flat - b.gs.flat).abs().max().item(), "D", (a.ds.flat - b.ds.flat).abs().max().item(),
          "| one vs one: G", (a.gs.flat - c.gs.flat).abs().max().item(), "D", (a.ds.flat - c.ds.flat).abs().max().item(), flush=True)
