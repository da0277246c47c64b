"""Throughput of SliceInference (per-slice BatchNorm statistics) at batch 32 of 512x512 for several lane counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import SliceInference
g, _, _ = bench.synth_states()
x, y = synth_batch(32, 512, seed=1234); x, y = x.cuda(), y.cuda()
for lanes in [int(v) for v in sys.argv[1:]] or [8]:
    eng = SliceInference(g, 32, 512, 512, lanes=lanes)
    for _ in range(3):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.run(x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"lanes {lanes}: {32 / ms * 1e3:.1f} slices/s  {ms:.3f} ms per 32 slices", flush=True)
    del eng
    torch.cuda.empty_cache()
