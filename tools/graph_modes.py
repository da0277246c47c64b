"""Does the step time of a captured TrainStep depend on what the process did before the capture?  Builds several
TrainSteps in one process (optionally creating K extra streams before each) and prints each one's step time.
usage: graph_modes.py [n=6] [extra_streams_before_each=0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g, d, vgg = bench.synth_states()
x, y = synth_batch(128, 64, seed=1234)
keep = []
for i in range(n):
    for _ in range(extra):
        s = torch.cuda.Stream(); keep.append(s)
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
    ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4)
    ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
    for _ in range(3):
        ts.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ts.step()
    e1.record(); torch.cuda.synchronize()
    print(f"TrainStep #{i}: {e0.elapsed_time(e1) / 30:.3f} ms  (queues {os.environ.get('GPU_MAX_HW_QUEUES')})", flush=True)
    del ts
