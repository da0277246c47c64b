"""Grouped sibling layers against the ungrouped plan: same losses, same parameters after two steps; step times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd import engine
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep
engine.BN_FOLD = False
g, d, vgg = bench.synth_states()
x, y = synth_batch(128, 64, seed=1234); x, y = x.cuda(), y.cuda()
res = {}
for grouped in (True, False):
    engine.SIBLING_GROUPS = grouped
    ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4)
    ts.step(x, y); ts.step()
    torch.cuda.synchronize()
    res[grouped] = (ts.losses(), ts.gs.flat.clone(), ts.ds.flat.clone(), ts.gs.flat_grad.clone())
    for _ in range(10):
        ts.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ts.step()
    e1.record(); torch.cuda.synchronize()
    print("grouped" if grouped else "ungrouped", f"{e0.elapsed_time(e1) / 50:.3f} ms/step", flush=True)
    del ts
a, b = res[True], res[False]
print("losses", a[0], b[0])
for name, i in (("G params", 1), ("D params", 2), ("G grads", 3)):
    da = (a[i] - b[i]).abs()
    print(name, "max diff", da.max().item(), "mean diff", da.mean().item(), "ref mean abs", b[i].abs().mean().item(),
          "cos", torch.nn.functional.cosine_similarity(a[i], b[i], dim=0).item())
