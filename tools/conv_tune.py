"""Time every distinct convolution of the training step under each tile configuration (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
from reid_gan_amd._lib import lib

# (N, H, W, Cin, Cout, kh, kw, stride, transposed, count per step)
SHAPES = ([l.split() for l in open(sys.argv[1]).read().strip().splitlines()]
          if len(sys.argv) > 1 and sys.argv[1] != "infer" else None)
CONFIGS = [(128, 128, 2), (128, 64, 2), (64, 128, 2), (64, 64, 2), (64, 64, 4)]
if os.environ.get("TUNE_CONFIGS"):
    CONFIGS = [tuple(int(v) for v in c.split("x")) for c in os.environ["TUNE_CONFIGS"].split(",")]


def run(N, H, W, Cin, Cout, kh, kw, s, tr, reps=20):
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    if kh == 4:
        ph = pw = 1
    if tr:
        Ho, Wo = H * s, W * s
    else:
        Ho, Wo = (H + 2 * ph - kh) // s + 1, (W + 2 * pw - kw) // s + 1
    x = ops.new_view(N, H, W, Cin); x.base.normal_()
    y = ops.new_view(N, Ho, Wo, Cout)
    w = torch.randn(Cout * kh * kw * Cin, device="cuda").to(torch.bfloat16)
    out = {}
    stats = None
    for cfg in CONFIGS:
        if Cout <= 64 and cfg[0] > 64:
            continue
        lib.ried_conv2d_force_tile(*cfg)
        if os.environ.get("TUNE_STATS"):    # with the BatchNorm statistics rows in the epilogue (one row per pixel tile
            # of the FORCED shape: sized after forcing it - the library cannot check the caller's buffer)
            rows = ops.conv2d_stat_rows(N, Ho, Wo, Cin, Cout, kh, kw, s, s, bool(tr))
            stats = torch.zeros(rows * 2 * Cout, device="cuda")
        try:
            f = lambda: ops.conv2d(x, w, y, kh, kw, s, s, ph, pw, transposed=bool(tr), stats=stats)
            f(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()           # graph replay: launch overhead off the clock
            with torch.cuda.graph(g):
                for _ in range(reps):
                    f()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record(); torch.cuda.synchronize()
            out[cfg] = e0.elapsed_time(e1) * 1e3 / reps
        except RuntimeError as e:
            out[cfg] = float("nan")
    lib.ried_conv2d_force_tile(0, 0, 0)
    return out, 2.0 * N * (H * W if tr else Ho * Wo) * Cin * Cout * kh * kw


def gather_shapes(B=128):
    import collections
    import bench
    from reid_gan_amd.synth import synth_batch
    from reid_gan_amd.train_step import TrainStep
    g, d, vgg = bench.synth_states()
    ts = TrainStep(g, d, vgg, B, 64, 64, use_graph=False)
    x, y = synth_batch(B, 64, seed=1234)
    ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
    seen = collections.Counter()
    orig = ops.conv2d
    def rec(xx, w, yy, kh, kw, sh, sw, ph, pw, **k):
        seen[(xx.N, xx.H, xx.W, xx.C, yy.C, kh, kw, sh, int(k.get("transposed", False)))] += 1
        return orig(xx, w, yy, kh, kw, sh, sw, ph, pw, **k)
    ops.conv2d = rec
    ts._run_eager(); torch.cuda.synchronize()
    ops.conv2d = orig
    del ts
    torch.cuda.empty_cache()
    return [list(k) + [c] for k, c in seen.items()]


def gather_infer_shapes(B=32, S=512):
    import collections
    import bench
    from reid_gan_amd.synth import synth_batch
    from reid_gan_amd.train_step import InferenceEngine
    g, _, _ = bench.synth_states()
    eng = InferenceEngine(g, B, S, S, use_graph=False)
    x, y = synth_batch(B, S, seed=1234)
    seen = collections.Counter()
    orig = ops.conv2d
    def rec(xx, w, yy, kh, kw, sh, sw, ph, pw, **k):
        seen[(xx.N, xx.H, xx.W, xx.C, yy.C, kh, kw, sh, int(k.get("transposed", False)))] += 1
        return orig(xx, w, yy, kh, kw, sh, sw, ph, pw, **k)
    ops.conv2d = rec
    eng.run(x.cuda(), y.cuda()); torch.cuda.synchronize()
    ops.conv2d = orig
    del eng
    torch.cuda.empty_cache()
    return [list(k) + [c] for k, c in seen.items()]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "infer":
        SHAPES = gather_infer_shapes()
    if SHAPES is None:
        SHAPES = gather_shapes()
    rows = []
    for sh in SHAPES:
        N, H, W, Cin, Cout, kh, kw, s, tr, cnt = map(int, sh)
        res, fl = run(N, H, W, Cin, Cout, kh, kw, s, tr)
        best = min(res, key=lambda k: res[k])
        line = " ".join(f"{c[0]}x{c[1]}x{c[2]}:{v:6.1f}" for c, v in res.items())
        io = 2.0 * N * (H * W * Cin + (H * W * s * s if tr else (H // s) * (W // s)) * Cout)      # x + y once, bf16
        print(f"N{N} {H}x{W}x{Cin}->{Cout} k{kh}x{kw} s{s} tr{tr} x{cnt} | best {best} {res[best]:.1f}us {fl/res[best]/1e6:.0f}TF/s "
              f"{io/res[best]/1e3:.0f}GB/s | {line}", flush=True)
