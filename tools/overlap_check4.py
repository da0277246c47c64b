"""After every step: is dpred_v what conv_c1_dgrad gives from the final a0.g?  Where not: which pixels, by how much."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from reid_gan_amd import engine, train_step, ops
from reid_gan_amd.synth import synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep
from gpu_util import g_state, d_state, DEV

engine.BN_FOLD = False
B, H = int(os.environ.get("B", "4")), 64
gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
batches = [tuple(t.to(DEV) for t in synth_batch(B, H, seed=60 + i)) for i in range(4)]
nbad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    V = ts.V
    for i in range(3):
        if os.environ.get("PREFILL"):
            ts.dpred_v.fill_(1.0)          # the step clears it itself: a stale read would show as ~1
        ts.step(*batches[i])
        torch.cuda.synchronize()
        tmp = torch.zeros_like(ts.dpred_v)
        ops.conv_c1_dgrad(V.a0.images(0, B).g, V.w0, tmp, scale=V.in_scale, accumulate=True)
        torch.cuda.synchronize()
        d = (tmp - ts.dpred_v).abs()
        if d.max().item() > 0:
            nbad += 1
            idx = (d.flatten() > 0).nonzero().flatten().tolist()
            pos = [(j // (H * H), (j // H) % H, j % H) for j in idx]
            print(f"rep {rep} step {i}: {len(idx)} pixels differ:", pos[:10],
                  "stored", [f"{ts.dpred_v.flatten()[j].item():.3e}" for j in idx[:6]],
                  "recomputed", [f"{tmp.flatten()[j].item():.3e}" for j in idx[:6]], flush=True)
    del ts
print("done, bad steps:", nbad)
