"""A few replays of the 512x512 inference batch (for rocprofv3 --kernel-trace; summarise with prof_summary.py:
the plan is built and captured first, so divide by the 1 eager + N replayed passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import InferenceEngine
g, _, _ = bench.synth_states()
eng = InferenceEngine(g, 32, 512, 512)
x, y = synth_batch(32, 512, seed=1234); x, y = x.cuda(), y.cuda()
for _ in range(7):
    eng.run(x, y)
torch.cuda.synchronize()
