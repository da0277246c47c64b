"""Time G forward+backward, D forward+backward and VGG forward+backward alone (graph replays) on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reid_gan_amd import engine, ops
from reid_gan_amd.engine import State, Generator, Discriminator, Vgg
from reid_gan_amd.synth import synth_batch

B = 128
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""       # "G", "D" or "V": time (and profile) one network only
g, d, vgg = bench.synth_states()
x, y = synth_batch(B, 64, seed=1)
x, y = x.cuda(), y.cuda()

def timeit(name, fn, reps=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps:.3f} ms", flush=True)

gs = State(g, "cuda"); G = Generator(gs, B, 64, 64)
dout = torch.randn(B, 1, 64, 64, device="cuda")
def gfb():
    G.forward(x); gs.zero_grad(); G.backward(dout)
if ONLY in ("", "G"):
    timeit("G forward", lambda: G.forward(x))
    timeit("G forward+backward", gfb)
if ONLY == "G":
    sys.exit(0)
ds = State(d, "cuda"); D = Discriminator(ds, 2 * B, 64, 64)
din = torch.cat([y, x], 0).contiguous()
def dfb():
    out = D.forward(din); ds.zero_grad(); ops.fill(out.g, 1e-3); D.backward(0, 2 * B, wgrad=True)
def dgb():
    out = D.forward(din); ops.fill(out.g.images(B, 2 * B), 1e-3); D.backward(B, 2 * B, wgrad=False, dimg=dout)
if ONLY in ("", "D"):
    timeit("D forward", lambda: D.forward(din))
    timeit("D forward+backward(data only, half)", dgb)
    timeit("D forward+backward(wgrad)", dfb)
if ONLY == "D":
    sys.exit(0)
V = Vgg(vgg, B, 64, 64)
loss = torch.zeros(1, device="cuda")
def vfb():
    V.forward(din); V.loss_and_seed(loss, 0.3); V.backward(dout)
if ONLY in ("", "V"):
    timeit("VGG forward", lambda: V.forward(din))
    timeit("VGG forward+loss+backward", vfb)
