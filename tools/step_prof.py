"""Run a few graph replays of the full training step (for rocprofv3 --kernel-trace; summarise with prof_one.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep
g, d, vgg = bench.synth_states()
ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4, lookahead=not os.environ.get("RIED_NO_LOOKAHEAD"))
x, y = synth_batch(128, 64, seed=1234)
ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
for _ in range(8):
    ts.step()
torch.cuda.synchronize()
