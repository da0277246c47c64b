"""Do two identical runs of the step agree bit for bit with the perceptual pass on its side stream (a race would show
as run-to-run differences), and do the look-ahead plans agree with the plain one?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from reid_gan_amd import engine, train_step
from reid_gan_amd.synth import synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep
from gpu_util import g_state, d_state, DEV

engine.BN_FOLD = False
B, H = 4, 64
gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
batches = [tuple(t.to(DEV) for t in synth_batch(B, H, seed=60 + i)) for i in range(4)]


def run(kind, overlap, graph=True):
    train_step.OVERLAP_VGG_PRED = overlap
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=graph, lookahead=kind != "plain")
    for i in range(3):
        x, y = batches[i]
        if kind == "ahead":
            ts.step(x, y, y_next=batches[i + 1][1])
        elif kind == "wrong":
            ts.step(x, y, y_next=batches[(i + 2) % 4][1])
        else:
            ts.step(x, y)
    torch.cuda.synchronize()
    return ts.gs.flat.clone(), ts.ds.flat.clone(), ts.losses()


def diff(a, b):
    return "G %.3e (%d > 1e-6)  D %.3e" % ((a[0] - b[0]).abs().max().item(), int(((a[0] - b[0]).abs() > 1e-6).sum()),
                                           (a[1] - b[1]).abs().max().item())


for graph in (True, False):
    base = run("plain", False, graph)
    print("graph", graph)
    print("  plain inline  vs plain inline :", diff(base, run("plain", False, graph)))
    print("  plain inline  vs plain overlap:", diff(base, run("plain", True, graph)))
    print("  plain overlap vs plain overlap:", diff(run("plain", True, graph), run("plain", True, graph)))
    for kind in ("ahead", "blind", "wrong"):
        print(f"  plain inline  vs {kind} inline :", diff(base, run(kind, False, graph)))
        print(f"  plain inline  vs {kind} overlap:", diff(base, run(kind, True, graph)))
