"""Gradient diagnostics of the HIP plans against the (bf16-emulating) oracle; run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ried_oracle as O
from reid_gan_amd import engine, ops
from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep

def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item(), (a.norm() / (b.norm() + 1e-30)).item()

O.set_bf16_emulation(True)
B = 2
sd = synth_state([(n, s) for n, s, _ in O.rigan_spec()], 101)
x, _ = synth_batch(B, 64, seed=7)
osd = {k: (v.clone().requires_grad_(True) if O.trainable(k) else v.clone()) for k, v in sd.items()}
ref = O.rigan_forward(O.Params(osd), x, True)
loss = (ref / 500.0).square().mean()
names = [k for k in osd if O.trainable(k)]
grads = dict(zip(names, torch.autograd.grad(loss, [osd[k] for k in names])))
dout = (2.0 / 500.0 ** 2 / ref.numel()) * ref.detach()
st = engine.State(sd, "cuda")
G = engine.Generator(st, B, 64, 64)
G.forward(x.cuda()); st.zero_grad(); G.backward(dout.cuda().contiguous()); torch.cuda.synchronize()
print("== generator backward, per parameter (cos, norm ratio)")
for k in names:
    c, r = cos(st.g[k].cpu(), grads[k])
    if c < 0.98 or abs(r - 1) > 0.08 or k.endswith("conv.weight") and ("conv1" in k or "deconv" in k or "out_conv" in k):
        print(f"{k:55s} cos {c:.4f} ratio {r:.4f} |ref| {grads[k].norm().item():.3g}")

print("== training step: d gloss / d pred")
gsd, dsd, vgg = sd, synth_state([(n, s) for n, s, _ in O.discriminator_spec()], 202), synth_vgg()
x, y = synth_batch(4, 64, seed=1234)
ts = TrainStep(gsd, dsd, vgg, 4, 64, 64, use_graph=False)
# run the pieces by hand to see each part of dpred
ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
B = 4
pred = ts.G.forward(ts.x)
assert pred.data_ptr() == ts.d_in[B:].data_ptr()      # y and pred are the two halves of d_in
ts._d_iteration(last=False)
# second iteration by hand up to the G loss
out = ts.D.forward(ts.d_in)
n = out.v.npix // 2 * out.v.C
ts.dpred.zero_(); ts.sc.zero_()
v_in = torch.cat([pred, ts.y])
ts.V.forward(v_in); ts.V.ctx.zero_backward(); ts.V.loss_and_seed(ts.sc[5:], 0.3); ts.V.backward(ts.dpred)
d_vgg = ts.dpred.clone()
ts.ssim_sums.zero_()
ops.ssim_fwd(pred, None, ts.y, 1e-4, 9e-4, ts.ssim_sums, gmap=ts.gmap)
ops.ssim_bwd(pred, ts.y, ts.gmap, -0.5 * 0.5 / (B * 64 * 64), ts.dpred, accumulate=True)
d_ssim = ts.dpred - d_vgg
ts.D.ctx.zero_backward()
ops.fill(out.g.images(B, 2 * B), -0.2 / n)
ts.D.backward(B, 2 * B, wgrad=False, dimg=ts.dpred)
d_gen = ts.dpred - d_vgg - d_ssim
torch.cuda.synchronize()
ost = O.TrainState(gsd, dsd, vgg)
r = O.train_step(ost, x, y, want_grads=True)
for nm, g in (("vgg", d_vgg), ("dssim", d_ssim), ("gen", d_gen), ("total", ts.dpred)):
    ref = r["dpred" if nm == "total" else "dpred_" + nm]
    print(nm, "cos/ratio", cos(g.cpu(), ref), "|ref|", ref.norm().item())
