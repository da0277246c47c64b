"""Eager training steps with the plan's main stream at high priority against default priority (does the dispatcher let
the generator backward's small kernels through while the discriminator backward's large ones run on a side lane?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep

g, d, vgg = bench.synth_states()
x, y = synth_batch(128, 64, seed=1234)
ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4, lookahead=True, use_graph=False)
ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
for prio in (0, -1, 0, -1):
    s = torch.cuda.Stream(priority=prio)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ts.step()
        e1.record(); torch.cuda.synchronize()
    print("priority", prio, "%.3f ms per eager step" % (e0.elapsed_time(e1) / 30), flush=True)
