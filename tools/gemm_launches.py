"""Per-launch table of the GEMM-family kernels in one eager training step (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reid_gan_amd import ops
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g, d, vgg = bench.synth_states()
ts = TrainStep(g, d, vgg, B, 64, 64, use_graph=False)
x, y = synth_batch(B, 64, seed=1234)
ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
ts.step(); ts.step()
if os.environ.get("FORCE_TILE"):          # e.g. "0,0,3": tuning hooks of ried_conv2d_force_tile
    from reid_gan_amd._lib import lib
    lib.ried_conv2d_force_tile(*[int(v) for v in os.environ["FORCE_TILE"].split(",")])
rec = []
orig = {k: getattr(ops, k) for k in ("conv2d", "conv2d_wgrad", "bgemm")}
def wrap(name, fn):
    def inner(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(*a, **kw); e1.record()
        if name == "conv":
            xx, w, yy, kh, kw_, sh, sw, ph, pw = a[:9]
            tr = kw.get("transposed", False)
            fl = 2.0 * (xx.npix if tr else yy.npix) * xx.C * yy.C * kh * kw_
            desc = f"conv{'T' if tr else ' '} N{xx.N} {xx.H}x{xx.W}x{xx.C} -> {yy.H}x{yy.W}x{yy.C} k{kh}x{kw_} s{sh} stats={kw.get('stats') is not None}"
        elif name == "wgrad":
            gg, pp, kh, kw_ = a[:4]
            fl = 2.0 * pp.npix * pp.C * gg.C * kh * kw_
            desc = f"wgrad g {gg.H}x{gg.W}x{gg.C} p {pp.H}x{pp.W}x{pp.C} k{kh}x{kw_} splitk {a[-1]}"
        else:
            M, N, K, batch = a[8:12]
            fl = 2.0 * M * N * K * batch
            desc = f"bgemm M{M} N{N} K{K} b{batch} tr{a[3]}{a[7]}"
        rec.append((desc, fl, e0, e1))
    return inner
ops.conv2d = wrap("conv", orig["conv2d"]); ops.conv2d_wgrad = wrap("wgrad", orig["conv2d_wgrad"]); ops.bgemm = wrap("bgemm", orig["bgemm"])
ts._run_eager(); torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for desc, fl, e0, e1 in rec:
    a = agg[desc]; a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e3
tot_t = sum(v[2] for v in agg.values()); tot_f = sum(v[1] for v in agg.values())
print(f"{len(rec)} launches, {tot_t/1e3:.2f} ms, {tot_f/1e12:.3f} TFLOP, {tot_f/tot_t/1e6:.1f} TF/s")
for desc, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:60]:
    print(f"{v[2]:8.1f} us  x{v[0]:<3d} {v[1]/1e9:8.2f} GF {v[1]/v[2]/1e6:7.1f} TF/s  {desc}")
