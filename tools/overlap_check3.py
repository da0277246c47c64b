"""Three steps, many identical runs, with the perceptual pass on its side stream: the first step and buffer that differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from reid_gan_amd import engine, train_step
from reid_gan_amd.synth import synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep
from gpu_util import g_state, d_state, DEV

engine.BN_FOLD = False
B, H = int(os.environ.get("B", "4")), 64
gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
batches = [tuple(t.to(DEV) for t in synth_batch(B, H, seed=60 + i)) for i in range(4)]
GRAPH = bool(int(os.environ.get("GRAPH", "0")))
ref = {}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=GRAPH)
    V = ts.V
    for i in range(3):
        ts.step(*batches[i])
        torch.cuda.synchronize()
        tens = {"pred": ts.G.out.clone(), "dpred_v": ts.dpred_v.clone(), "dpred": ts.dpred.clone(),
                "a0g": V.a0.images(0, B).g.torch().float().clone(), "a0v": V.a0.images(0, B).v.torch().float().clone(),
                "Dout": ts.D.out.v.torch().float().clone(), "G": ts.gs.flat.clone(), "D": ts.ds.flat.clone(),
                "Ggrad": ts.gs.flat_grad.clone(), "Dgrad": ts.ds.flat_grad.clone(), "gmap": ts.gmap.clone()}
        for j, l in enumerate(V.layers):
            if l[0] == "conv":
                tens[f"v{j}"] = l[3].images(0, 2 * B).v.torch().float().clone()
                tens[f"g{j}"] = l[3].images(0, B).g.torch().float().clone()
        if rep == 0:
            ref[i] = tens
            continue
        bad = []
        for k in tens:
            d = (tens[k] - ref[i][k]).abs()
            if d.max().item() > 0:
                bad.append(f"{k} {d.max().item():.2e}/{ref[i][k].abs().max().item():.2e} ({int((d > 0).sum())} of {d.numel()})")
        bad = [b for b in bad if not b.startswith(("Ggrad", "G ", "Dgrad", "D "))] if i == 0 else bad
        if bad:
            print(f"rep {rep} step {i}:", "; ".join(bad[:14]), flush=True)
            break
    del ts
print("done")
