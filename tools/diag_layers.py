"""Per-layer error report of the HIP plans against the CPU oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ried_oracle as O
from reid_gan_amd import engine, ops
from reid_gan_amd.synth import synth_state, synth_batch

def nchw(v): return v.torch().float().cpu().permute(0, 3, 1, 2).contiguous()
def rep(name, got, ref):
    e = (got.double() - ref.double()).abs()
    print(f"{name:28s} max {e.max().item()/ref.abs().max().item():.4f}  mean {e.mean().item()/ref.abs().mean().item():.4f}  |ref| max {ref.abs().max().item():.3g} mean {ref.abs().mean().item():.3g}")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
O.set_bf16_emulation(len(sys.argv) > 2)
sd = synth_state([(n, s) for n, s, _ in O.rigan_spec()], 101)
x, _ = synth_batch(B, 64, seed=7)
acts = {}
with torch.no_grad():
    ref = O.rigan_forward(O.Params({k: v.clone() for k, v in sd.items()}), x, True, acts)
G = engine.Generator(engine.State(sd, "cuda"), B, 64, 64, grads=False)
out = G.forward(x.cuda())
for n, s in G.tops.items(): rep(n, nchw(s.v), acts[n])
rep("out", out.cpu(), ref)
dsd = synth_state([(n, s) for n, s, _ in O.discriminator_spec()], 202)
xd, yd = synth_batch(B, 64, seed=9)
d_in = torch.cat([yd, xd], 0)
acts = {}
with torch.no_grad():
    ref = O.discriminator_forward(O.Params({k: v.clone() for k, v in dsd.items()}), d_in, True, acts)
D = engine.Discriminator(engine.State(dsd, "cuda"), 2 * B, 64, 64, grads=False)
D.forward(d_in.cuda())
for i in range(7): rep(f"d{i}", nchw(D.acts[i].v), acts[f"d{i}"])
