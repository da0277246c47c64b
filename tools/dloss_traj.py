"""Loss trajectory of the benchmark step over rotating batches: when do the hinge gates of DLoss close?
usage: python tools/dloss_traj.py [steps] [nbatches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g, d, vgg = bench.synth_states()
ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4, use_graph=True, lookahead=True)
bs = [tuple(t.cuda() for t in synth_batch(128, 64, seed=1234 + i)) for i in range(nb)]
for i in range(steps):
    x, y = bs[i % nb]
    ts.step(x, y, y_next=bs[(i + 1) % nb][1])
    l = ts.losses()
    print(i, " ".join(f"{k}={l[k]:.4f}" for k in ("dloss1", "dloss2", "gloss", "gen", "vgg", "dssim")), flush=True)
ts.close()
