#!/usr/bin/env python3
"""Benchmark of the RIED-GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one full training iteration (G forward, two D iterations with Adam, VGG16-perceptual + DSSIM + GAN
generator loss, G backward, G Adam: src/train.py:153-177 of the reference) on a batch of synthetic 1x64x64 HU
patches per GPU (BASELINE.json configs[1]: batch 128, bf16 storage / fp32 accumulate).  Weak scaling: every
rank processes its own 128 patches; gradients are averaged with RCCL all-reduces.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.

Extra fields
  roofline      the dominant kernel family (the implicit-GEMM convolution kernel): algorithmic FLOPs of all
                its launches in one step / their summed duration, measured with HIP events on the launch
                stream in one eager step after the timed region, against the dense bf16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline  the CPU oracle (oracle/ried_oracle.py, a port of the reference arithmetic on stock torch CPU
                ops, fp32) timed on this host's cores on BASELINE.json configs[0] (batch 4 of 1x64x64).
  loader        the device patch pipeline (src/loader.py), patches/s with its own HBM roofline (N = 1 only).
  infer         512x512 slice inference + SSIM/PSNR/RMSE (configs[2], batch 32), slices/s (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import reid_gan_amd  # noqa: F401  (sets the HIP hardware-queue count before the runtime initialises)
import torch

PEAK_MFMA_BF16 = 2.5e15        # dense bf16 MFMA peak of one MI355X, FLOP/s
PEAK_HBM = 8.0e12              # HBM3E peak, B/s
G_CONV_OUT_PER_PIXEL = 716288 / 4096.0   # conv + deconv output elements of RIGAN per input pixel
FLOP_PER_PATCH_64 = 17.716e9   # 3 G + 13 D + 3 VGG forward-equivalents at 64x64 (needed work only)


def synth_states(seed=0):
    from reid_gan_amd import rinet
    from reid_gan_amd.synth import synth_state, synth_vgg
    g = synth_state(rinet.state_spec(rinet.RIGAN()), 101 + seed)
    d = synth_state(rinet.state_spec(rinet.ImageDiscriminator()), 202 + seed)
    return g, d, synth_vgg()


def instrumented_step(ts):
    """One eager step with HIP events around every launch of the MFMA kernels (convolutions, weight gradients,
    attention).  Returns (families, templates): time, flops and algorithmic bytes summed per kernel family and per
    kernel template (the template a call runs comes from ried_conv2d_which / the tile rules of the weight gradient)."""
    from reid_gan_amd import ops
    from reid_gan_amd._lib import lib
    rec = []            # (family, template, flops, bytes, event, event)
    names = ("conv2d", "conv2d_tap_mse", "conv2d_wgrad", "bgemm", "attn_fwd", "attn_bwd", "conv2d_signs",
             "conv2d_s2_dgrad_c1fold", "conv2d_tap_mse_bf16")
    orig = {k: getattr(ops, k) for k in names}

    def timed(family, template, flops, nbytes, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        rec.append((family, template, flops, nbytes, e0, e1))

    def conv(x, w, y, kh, kw_, sh, sw, ph, pw, **k):
        """Algorithmic HBM bytes of one launch: each operand once (bf16), the fp32 tap copy, re-read on accumulate."""
        nb = 2.0 * (x.npix * x.C + w.numel() + y.npix * y.C * (2 if k.get("accumulate") else 1))
        if k.get("y2") is not None:
            nb += 4.0 * y.npix * y.C
        if k.get("mask") is not None:
            nb += 2.0 * y.npix * y.C
        fl = 2.0 * (x.npix if k.get("transposed") else y.npix) * x.C * y.C * kh * kw_
        timed("conv", ops.conv2d_which(x, w, y, kh, kw_, sh, sw, ph, pw, **k), fl, nb,
              lambda: orig["conv2d"](x, w, y, kh, kw_, sh, sw, ph, pw, **k))

    def conv_signs(x, w, y, kh, kw_, sh, sw, ph, pw, signs, **k):      # the mask is one byte per eight channels
        nb = 2.0 * (x.npix * x.C + w.numel() + y.npix * y.C) + y.npix * y.C / 8.0
        fl = 2.0 * (x.npix if k.get("transposed") else y.npix) * x.C * y.C * kh * kw_
        kk = {a: b for a, b in k.items() if a != "mask_slope"}
        timed("conv", ops.conv2d_which(x, w, y, kh, kw_, sh, sw, ph, pw, **kk), fl, nb,
              lambda: orig["conv2d_signs"](x, w, y, kh, kw_, sh, sw, ph, pw, signs, **k))

    def c1fold(dy, wt, mask, mask_slope, stats, ximg, dw, scratch, **k):
        # data gradient of the 3x3 stride-2 layer into 64 channels at twice the resolution (never stored) + the first
        # layer's 64 x 9 weight gradient from it; bytes: dy, the weights, the mask, the fp32 image
        npo = 4.0 * dy.npix
        mb = npo * 8.0 if not isinstance(mask, ops.View) else 2.0 * npo * 64
        timed("conv", "conv_halo_s2t_kernel<16,fold>", 2.0 * dy.npix * dy.C * 64 * 9 + 2.0 * npo * 64 * 9,
              2.0 * (dy.npix * dy.C + wt.numel()) + mb + 4.0 * npo,
              lambda: orig["conv2d_s2_dgrad_c1fold"](dy, wt, mask, mask_slope, stats, ximg, dw, scratch, **k))

    def tap(x, w, bias, y, target, gout, gscale, stats):      # x, w, y, gradient seed (bf16) + the fp32 target tap
        nb = 2.0 * (x.npix * x.C + w.numel() + 2 * y.npix * y.C) + 4.0 * y.npix * y.C
        timed("conv", ops.conv2d_which(x, w, y, 3, 3, 1, 1, 1, 1, bias=bias, act=1, stats=stats) + " +tap MSE",
              2.0 * y.npix * y.C * x.C * 9, nb, lambda: orig["conv2d_tap_mse"](x, w, bias, y, target, gout, gscale, stats))

    def tap16(x, w, bias, y, target, gout, gscale, stats):    # ... with the target tap as stored bf16 activations
        nb = 2.0 * (x.npix * x.C + w.numel() + 3 * y.npix * y.C)
        timed("conv", ops.conv2d_which(x, w, y, 3, 3, 1, 1, 1, 1, bias=bias, act=1, stats=stats) + " +tap MSE",
              2.0 * y.npix * y.C * x.C * 9, nb,
              lambda: orig["conv2d_tap_mse_bf16"](x, w, bias, y, target, gout, gscale, stats))

    def wgrad(g, p, kh, kw_, sh, sw, ph, pw, slab, splitk):
        bm, bn = (64 if g.C <= 64 else 128), (64 if p.C <= 64 else 128)
        nb = 2.0 * (g.npix * g.C + p.npix * p.C) + 4.0 * splitk * kh * kw_ * p.C * g.C
        nt = lib.ried_conv2d_wgrad_taps_per_block(g.C, p.C, kh, kw_, p.npix)
        timed("wgrad", f"wgrad_glds_kernel<{bm},{bn},2" + (f",{nt}>" if nt > 1 else ">"),
              2.0 * p.npix * p.C * g.C * kh * kw_, nb,
              lambda: orig["conv2d_wgrad"](g, p, kh, kw_, sh, sw, ph, pw, slab, splitk))

    def bgemm(A, lda, bsa, a_tr, B, ldb, bsb, b_tr, M, N, K, batch, *a, **k):
        timed("attention", "igemm_kernel<*>", 2.0 * M * N * K * batch, 2.0 * batch * (M * K + N * K + M * N),
              lambda: orig["bgemm"](A, lda, bsa, a_tr, B, ldb, bsb, b_tr, M, N, K, batch, *a, **k))

    def attn_fwd(qkv, x, out, O, lse, gamma, d):
        n, C, nimg = qkv.H * qkv.W, O.C, qkv.N
        timed("attention", f"attn_fwd_kernel<{n},{d},{C}>", 2.0 * nimg * n * n * (d + C),
              2.0 * nimg * n * (2 * d + 4 * C), lambda: orig["attn_fwd"](qkv, x, out, O, lse, gamma, d))

    def attn_bwd(qkv, dqkv, dout, O, lse_ptr, gamma, dgamma, d):
        n, C, nimg = qkv.H * qkv.W, O.C, qkv.N
        timed("attention", f"attn_bwd_kernel<{n},{d},{C}>", 2.0 * nimg * n * n * (3 * d + 2 * C),
              2.0 * nimg * n * (4 * d + 4 * C), lambda: orig["attn_bwd"](qkv, dqkv, dout, O, lse_ptr, gamma, dgamma, d))

    ops.conv2d, ops.conv2d_tap_mse, ops.conv2d_wgrad, ops.bgemm = conv, tap, wgrad, bgemm
    ops.conv2d_signs, ops.conv2d_s2_dgrad_c1fold, ops.conv2d_tap_mse_bf16 = conv_signs, c1fold, tap16
    ops.attn_fwd, ops.attn_bwd = attn_fwd, attn_bwd
    launch_wj, launch_cj = ops.WgradJobs.launch, ops.ConvJobs.launch      # the grouped launches of the generator

    def wj(self):
        # (bn carries the jobs' taps per block above bit 16)
        launch_wj(self, around=lambda bm, bn, fn: timed(
            "wgrad", f"wgrad_glds_jobs_kernel<{bm},{bn & 0xFFFF},2" + (f",{(bn >> 16) + 1}>" if bn >> 16 else ">"),
            self.group_flops[(bm, bn)], self.group_bytes[(bm, bn)], fn))

    def cj(self):
        timed("conv", "conv_glds_jobs_kernel<64,64,2>", self.flops, self.bytes, lambda: launch_cj(self))
    ops.WgradJobs.launch, ops.ConvJobs.launch = wj, cj
    # kernels timed one at a time: no side streams while the events are in place
    saved = (ts.overlap_vgg, ts.overlap_gbwd, os.environ.get("RIED_SERIAL_BRANCHES"), ts.overlap_dreal,
             ts.overlap_vgg_pred)
    ts.overlap_vgg = ts.overlap_gbwd = ts.overlap_dreal = ts.overlap_vgg_pred = False
    os.environ["RIED_SERIAL_BRANCHES"] = "1"
    try:
        # An event pair also spans the time the GPU waits for the host to submit the launch between them, so the
        # stream is given a head start of dummy work: the host then stays ahead and the pairs time kernels only.
        pad = torch.empty(1 << 29, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(48):
            pad.zero_()
        ts._run_eager()
        torch.cuda.synchronize()
        del pad
    finally:
        for k, v in orig.items():
            setattr(ops, k, v)
        ops.WgradJobs.launch, ops.ConvJobs.launch = launch_wj, launch_cj
        ts.overlap_vgg, ts.overlap_gbwd, ts.overlap_dreal, ts.overlap_vgg_pred = saved[0], saved[1], saved[3], saved[4]
        if saved[2] is None:
            del os.environ["RIED_SERIAL_BRANCHES"]
        else:
            os.environ["RIED_SERIAL_BRANCHES"] = saved[2]
    fam, tem = {}, {}
    for family, template, fl, nb, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        for table, key in ((fam, family), (tem, template)):
            d = table.setdefault(key, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["flops"] += fl
            d["ms"] += ms
            d["bytes"] += nb
    return fam, tem


def _pmc_tables():
    """Counter evidence committed under profiles/ (rocprofv3 --pmc passes over tools/step_prof.py): HBM bytes per launch
    and MFMA-busy fraction per kernel name; the newest round's files win."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    traffic, busy = None, {}
    for r in ("r03", "r02", "r01"):
        f = os.path.join(root, f"{r}_pmc_conv_traffic.json")
        if traffic is None and os.path.exists(f):
            traffic = json.load(open(f))
        f = os.path.join(root, f"{r}_pmc_mfma_busy.json")
        if not busy and os.path.exists(f):
            busy = {k.replace("void ", "").replace(" ", ""): v["mfma_busy_frac"]
                    for k, v in json.load(open(f))["kernels"].items()}
    return traffic, busy


def cpu_baseline(budget_s=12.0):
    from oracle import ried_oracle as O
    from reid_gan_amd.synth import synth_batch
    cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    g, d, vgg = synth_states()
    x, y = synth_batch(4, 64, seed=1234)
    st = O.TrainState(g, d, vgg)
    O.train_step(st, x, y)                       # warm-up
    t0, n = time.perf_counter(), 0
    while n < 400 and (time.perf_counter() - t0 < budget_s or n < 2):
        O.train_step(st, x, y)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": 4 * n / dt, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"BASELINE configs[0]: {n} full steps at batch 4 of 1x64x64, fp32, stock torch CPU ops "
                      f"({dt:.1f} s)"}


def infer_bench(steps=5, warmup=2, batch=32, size=512, slice_mode=False):
    from reid_gan_amd.synth import synth_batch
    from reid_gan_amd.train_step import InferenceEngine
    g, _, _ = synth_states()
    eng = InferenceEngine(g, batch, size, size)
    x, y = synth_batch(batch, size, seed=1234)
    x, y = x.cuda(), y.cuda()
    for _ in range(warmup):
        eng.run(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.run(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    # metric kernel alone (one read of x, y, pred: 12 B/pixel)
    from reid_gan_amd import ops
    pred = eng.G.out
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0.record()
    for _ in range(20):
        ops.ssim_fwd(x, pred, y, 16.0, 144.0, eng.sums, clamp=(-160.0, 240.0))
    m1.record()
    torch.cuda.synchronize()
    mms = m0.elapsed_time(m1) / 20
    bytes_alg = batch * size * size * 12.0
    # the reference's loop as written (batch-1 loader: BatchNorm statistics per slice), batched by running the
    # slices' batch-1 plans side by side
    per_slice = None
    if slice_mode:
        del eng
        torch.cuda.empty_cache()
        se = InferenceEngine(g, batch, size, size, per_slice_stats=True)
        for _ in range(2):
            se.run(x, y)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(steps):
            se.run(x, y)
        s1.record()
        torch.cuda.synchronize()
        sms = s0.elapsed_time(s1) / steps
        per_slice = {"value": batch / (sms * 1e-3), "unit": "slices/s", "ms_per_batch": sms,
                     "note": "BatchNorm statistics of each slice alone, inside the batched kernels (src/test.py:107-131 "
                             "runs a batch-1 loader); 'value' above is the generator applied to the whole batch"}
        del se
        torch.cuda.empty_cache()
    # HBM roofline of the whole pass.  Algorithmic bytes: every convolution / transposed-convolution output of the
    # generator written and read once in bf16 (174.875 output elements per input pixel: 716,288 per 64 x 64 image,
    # the same at every size), the fp32 input slice read once, and the metrics' one read of x, y and the prediction
    # (12 B per pixel).  BatchNorm passes, pools, upsampling and concatenation are not in it: they are what an ideal
    # implementation folds away.  `traffic`: HBM bytes per batch from the committed counter passes
    # (tools/pmc_infer.py over tools/infer_prof.py).
    bytes_pass = batch * size * size * (G_CONV_OUT_PER_PIXEL * 4.0 + 4.0 + 12.0)
    traffic = None
    f = os.path.join(ROOT, "profiles", "r03_pmc_infer_traffic.json")
    if os.path.exists(f) and batch == 32 and size == 512:
        traffic = json.load(open(f))["traffic_MB_per_batch"] * 1e6
    roof = {"bound": "hbm", "achieved": bytes_pass / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
            "frac": bytes_pass / (ms * 1e-3) / PEAK_HBM, "algorithmic_bytes_per_batch": bytes_pass,
            "traffic": traffic, "traffic_over_algorithmic": None if traffic is None else traffic / bytes_pass}
    return {"metric": "512x512 slices/sec (generator inference + SSIM/PSNR/RMSE)", "value": batch / (ms * 1e-3),
            "roofline": roof,
            "per_slice_statistics": per_slice,
            "unit": "slices/s", "ms_per_batch": ms, "batch": batch,
            "mfma_frac": batch / (ms * 1e-3) * 34.67e9 / PEAK_MFMA_BF16,
            "metrics_kernel": {"bound": "hbm", "ms": mms, "achieved": bytes_alg / (mms * 1e-3) / 1e9,
                               "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": bytes_alg / (mms * 1e-3) / PEAK_HBM}}


def loader_bench(batch=128, patch=64, patch_n=16, nslices=64, size=512, iters=50, cpu=True):
    """Patch pipeline (src/loader.py): `batch` patches per launch cut from int16 512x512 slice pairs resident in
    HBM, fp32 out.  Algorithmic bytes per patch: 2 tensors x P*P x (2 read + 4 written) = 12 B/pixel.  The CPU leg
    is the oracle's restatement of get_patch + the .float() of train.py:136 on the same slices (one core)."""
    import numpy as np
    from reid_gan_amd import loader as L
    from reid_gan_amd.synth_slices import synth_slices
    inp, tg = synth_slices(nslices, size, size, seed=5)
    a, b = torch.from_numpy(inp).cuda(), torch.from_numpy(tg).cuda()
    nsel = batch // patch_n
    for _ in range(3):
        L._crop(a, b, 0, nsel, patch_n, patch, None, 1, 0, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        L._crop(a, b, (i * nsel) % (nslices - nsel + 1), nsel, patch_n, patch, None, 1, i * batch, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_alg = batch * patch * patch * 12.0
    out = {"metric": "training patches/sec cropped + converted on the device", "value": batch / (ms * 1e-3),
           "unit": "patches/s", "ms_per_batch": ms, "batch": batch, "patch": patch,
           "roofline": {"bound": "hbm", "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9,
                        "unit": "GB/s", "frac": bytes_alg / (ms * 1e-3) / PEAK_HBM}}
    if cpu:
        from oracle import ried_oracle as O
        np.random.seed(0)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            for s in range(nsel):
                px, py = O.get_patch(inp[s], tg[s], patch_n, patch)
                px.astype(np.float32); py.astype(np.float32)
            n += batch
        out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "patches/s", "cores": 1,
                               "kind": "port", "sample": "numpy get_patch + float conversion, same slices, 2 s"}
    return out


def vgg_bench(batch=256, size=256, iters=4):
    """The perceptual-loss-heavy configuration: VGG16 taps of `batch` prediction and `batch` target patches of
    size x size, the five-tap MSE and its gradient w.r.t. the prediction (src/loss.py:179-189 and its autograd).
    Needed work per patch: two forwards + one data-gradient pass = 3 x 305,856 MACs per input pixel."""
    from reid_gan_amd.engine import Vgg
    from reid_gan_amd.synth import synth_batch, synth_vgg
    V = Vgg(synth_vgg(), batch, size, size)
    nb = min(batch, 8)                              # the synthetic images repeat: generating 512 of them is host time
    p, y = synth_batch(nb, size, seed=1234)
    reps = (batch + nb - 1) // nb
    imgs = torch.cat([p.repeat(reps, 1, 1, 1)[:batch], y.repeat(reps, 1, 1, 1)[:batch]], 0).cuda()
    loss = torch.zeros(1, device="cuda")
    dpred = torch.zeros(batch, 1, size, size, device="cuda")

    def step():
        V.forward(imgs, batch, 2 * batch)
        V.ctx.zero_backward()
        V.forward_pred_fused(imgs, 0.3)
        V.loss_from_rows(loss)
        V.backward(dpred)
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * 3 * 305856 * size * size * batch
    return {"metric": "patches/sec through the VGG16 perceptual loss (two forwards + data gradient)",
            "value": batch / (ms * 1e-3), "unit": "patches/s", "ms_per_step": ms, "batch": batch, "patch": size,
            "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_MFMA_BF16 / 1e12,
                         "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / PEAK_MFMA_BF16}}


def prep_bench(nslices=208, size=512, iters=20, cpu=True):
    """get_pixels_hu (src/prep.py:46-59) on one series resident in HBM: 2 B read + 2 B written per pixel."""
    import numpy as np
    from reid_gan_amd._lib import lib
    r = np.random.default_rng(1)
    stack = r.integers(0, 4096, (nslices, size, size)).astype(np.uint16)
    raw = torch.from_numpy(stack.view(np.int16)).cuda()
    slope = torch.ones(nslices, dtype=torch.float64, device="cuda")
    icpt = torch.full((nslices,), -1024, dtype=torch.int32, device="cuda")
    outp = torch.empty_like(raw)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        lib.ried_hu_convert(raw.data_ptr(), slope.data_ptr(), icpt.data_ptr(), outp.data_ptr(), nslices, size * size, s)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 4.0 * stack.size
    out = {"metric": "512x512 slices/sec converted to HU on the device", "value": nslices / (ms * 1e-3),
           "unit": "slices/s", "ms_per_series": ms, "slices": nslices,
           "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                        "frac": nbytes / (ms * 1e-3) / PEAK_HBM}}
    if cpu:
        from oracle import ried_oracle as O
        t0 = time.perf_counter()
        O.get_pixels_hu(list(stack), [1.0] * nslices, [-1024.0] * nslices)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": nslices / dt, "unit": "slices/s", "cores": 1, "kind": "port",
                               "sample": f"numpy get_pixels_hu on the same {nslices} slices, once ({dt:.2f} s)"}
    return out


def legacy_bench(batch=128, size=64, iters=20, cpu=True):
    """GNet / DNet of src/networks.py:460-556: forward + backward + Adam on `batch` patches, eager launches.
    Algorithmic MACs per 64x64 patch, forward: GNet 2.744e9 (ten 3x3 256->256 convs at 16x16 are 1.51e9 of it),
    DNet 7.61e7; forward + both gradients = 3x (the first layers have no data gradient: < 0.1 %)."""
    from reid_gan_amd import networks as NW
    from reid_gan_amd.engine import State
    from reid_gan_amd.synth import synth_state, synth_batch
    px = size * size
    macs = {"GNet": 64 * 9 * px + 128 * 64 * 9 * px // 4 + 256 * 128 * 9 * px // 16 + 10 * 256 * 256 * 9 * px // 16 +
                    256 * 256 * 16 * px // 16 + 384 * 128 * 16 * px // 4 + 192 * 9 * px,
            "DNet": 64 * 9 * px // 4 + sum((64 << i) * (128 << i) * 9 * (px >> (4 + 2 * i)) for i in range(4))}
    x, _ = synth_batch(batch, size, seed=77)
    x = x.cuda()
    out = {}
    for name, mod, plan in (("GNet", NW.GNet, NW.GNetPlan), ("DNet", NW.DNet, NW.DNetPlan)):
        sd = synth_state([(k, tuple(v.shape)) for k, v in mod().state_dict().items()], 9)
        st = State(sd, "cuda")
        P = plan(st, batch, size, size)
        dout = torch.randn_like(P.forward(x))

        def step():
            P.forward(x)
            st.zero_grad()
            P.backward(dout)
            st.adam(1e-4)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flops = 6.0 * macs[name] * batch
        out[name] = {"value": batch / (ms * 1e-3), "unit": "patches/s (forward + backward + Adam)", "ms_per_step": ms,
                     "batch": batch, "patch": size,
                     "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_MFMA_BF16 / 1e12,
                                  "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / PEAK_MFMA_BF16}}
        del P, st
    if cpu:
        from oracle import ried_oracle as O
        xs = x[:4].cpu()
        for name, fwd, spec in (("GNet", O.gnet_forward, O.gnet_spec()), ("DNet", O.dnet_forward, O.dnet_spec())):
            sd = synth_state([(n, s) for n, s, _ in spec], 9)
            sd = {k: (v.requires_grad_(True) if O.trainable(k) else v) for k, v in sd.items()}
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < 3.0 or n < 2:
                o = fwd(O.Params(dict(sd)), xs, True)
                torch.autograd.grad(o.square().mean(), [v for k, v in sd.items() if O.trainable(k)])
                n += 1
            out[name]["cpu_baseline"] = {"value": 4 * n / (time.perf_counter() - t0), "unit": "patches/s",
                                         "cores": torch.get_num_threads(), "kind": "port",
                                         "sample": f"{n} forward + backward passes at batch 4, fp32, stock torch CPU ops"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="patches per GPU")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="prepare every step's targets inside that step (default on one GPU: at the end of the step before)")
    ap.add_argument("--reseed-d", type=int, default=8,
                    help="put the discriminator back to its initial state every that many steps (0: never), so that "
                         "the hinge gates of DLoss stay open in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    args = ap.parse_args()

    from reid_gan_amd import dp, ops
    from reid_gan_amd.synth import synth_batch
    from reid_gan_amd.train_step import TrainStep
    try:          # native frames on stderr if the process dies on a signal (diagnostics only: never fatal itself)
        ops.install_crash_backtrace(None)
    except Exception:          # noqa: BLE001
        pass
    if os.environ.get("RIED_HANG_DUMP"):      # debugging aid: every thread's stack after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RIED_HANG_DUMP"]), exit=True)
    # RIED_DP_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (ranks then share cards;
    # RCCL itself refuses two ranks on one device)
    backend = os.environ.get("RIED_DP_BACKEND", "nccl")
    if backend != "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    rank, local, world = dp.init_from_env(backend)
    force_pg = world == 1 and bool(os.environ.get("RIED_FORCE_PG"))      # one-rank group: the multi-rank code path
    if force_pg:
        import torch.distributed as dist0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist0.init_process_group(backend=backend, rank=0, world_size=1)
    if world != args.gpus:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local if backend == "nccl" else local % torch.cuda.device_count())
    import torch.distributed as dist
    pg = dist.group.WORLD if (world > 1 or force_pg) else None
    multi = pg is not None

    g, d, vgg = synth_states()
    ts = TrainStep(g, d, vgg, args.batch, args.size, args.size, lr=1e-4, use_graph=not args.no_graph,
                   process_group=pg, lookahead=not args.no_lookahead)
    if os.environ.get("RIED_HANG_DUMP"):
        import threading

        def watch():
            while True:
                time.sleep(10)
                print(f"[rank {rank}] steps done {ts.steps_done}, segment {ts.segment}, exchange {ts.exchange}",
                      file=sys.stderr, flush=True)
        threading.Thread(target=watch, daemon=True).start()
    # The regime the step trains in: NB resident batches rotate through step(x, y, y_next=...) - the call a
    # prefetching loader makes, so a look-ahead plan finds every batch's targets announced - and the discriminator
    # is put back to its initial state every --reseed-d steps: on synthetic data it separates target from
    # prediction within five steps, after which both hinge gates of DLoss (batch means, src/loss.py:141-142) are
    # closed and every discriminator gradient kernel would be multiplying exact zeros.
    NB = 4
    batches = [tuple(t.cuda() for t in synth_batch(args.batch, args.size, seed=1234 + rank + 1000 * i))
               for i in range(NB)]
    dsnap = ts.snapshot_discriminator() if args.reseed_d > 0 else None
    counter = [0]

    def one_step():
        i = counter[0]
        counter[0] += 1
        if dsnap is not None and i % args.reseed_d == 0 and i > 0:
            ts.reseed_discriminator(dsnap)
        x, y = batches[i % NB]
        if ts.lookahead:
            ts.step(x, y, y_next=batches[(i + 1) % NB][1])
        else:
            ts.step(x, y)

    first = None
    for i in range(args.warmup):
        one_step()
        if i == 0:
            first = ts.losses()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    losses = ts.losses()                 # of the last timed step
    # Are the hinge gates open in the timed pattern?  Untimed: the same loop goes on until the step after the next
    # re-seed, whose losses are read (a host synchronisation the timed loop does not have).
    gates = None
    if dsnap is not None:
        while (counter[0] - 1) % args.reseed_d != 1:
            one_step()
        gates = ts.losses()
        assert gates["dloss1"] > 0.0 or gates["dloss2"] > 0.0, ("the hinge gates are closed right after a re-seed", gates)
    value = world * args.batch * args.steps / dt
    out = {
        "metric": "train patches/sec (G+D+VGG step)", "value": value, "unit": "patches/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: full RIED-GAN training step, batch {args.batch} of "
                               f"1x{args.size}x{args.size} patches per GPU, bf16 storage / fp32 accumulate",
                   "global_batch": world * args.batch, "patch": args.size, "parallelism": f"dp{world}",
                   "hip_graph": not args.no_graph,
                   # every timed step does the whole step's work; with look-ahead the part that needs only the target
                   # batch (VGG target features, D's weight preparation and target-half pass) is the NEXT step's,
                   # run at the end of this one beside the generator backward
                   "lookahead": ts.lookahead, "resident_batches": NB, "announced_y_next": ts.lookahead,
                   "d_reseed_every": args.reseed_d, "losses": losses, "losses_first_step": first,
                   "losses_step_after_reseed": gates},
    }
    if args.size == 64:
        out["step_mfma_frac"] = value / world * FLOP_PER_PATCH_64 / PEAK_MFMA_BF16
    if rank == 0:
        fam, tem = instrumented_step(ts)
        dom = max(fam, key=lambda k: fam[k]["ms"])
        a = fam[dom]["flops"] / (fam[dom]["ms"] * 1e-3)
        names = {"conv": "conv_glds_kernel<*>, conv_glds_jobs_kernel, conv_halo_kernel<*>, conv_halo_s2t_kernel, "
                         "conv_halo_d4_kernel (conv / transposed conv forward and data gradient)",
                 "wgrad": "wgrad_glds_kernel / wgrad_glds_jobs_kernel<*> (weight gradient)",
                 "attention": "attn_fwd_kernel / attn_bwd_kernel (fused self-attention)"}
        # HBM bytes per launch of the conv family from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as the
        # gfx950 correction for wide reads, + WRITE_SIZE); the algorithmic bytes per launch are reported beside it
        pmc_traffic, pmc_busy = _pmc_tables()
        traffic = pmc_traffic["traffic_MB_per_launch"] * 1e6 if (dom == "conv" and pmc_traffic) else None
        # per kernel template: launches and time per step, flop rate, algorithmic HBM rate, MFMA-busy (counter pass)
        templates = []
        for k, v in sorted(tem.items(), key=lambda kv: -kv[1]["ms"]):
            key = k.split(" ")[0].replace(" ", "")
            busy = next((b for n, b in pmc_busy.items() if n.startswith(key.rstrip(">")) or
                         key.replace(",", ", ") in n), None)
            templates.append({"kernel": k, "launches": v["launches"], "us": round(v["ms"] * 1e3, 1),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                              "mfma_frac": round(v["flops"] / (v["ms"] * 1e-3) / PEAK_MFMA_BF16, 4),
                              "hbm_gbs_algorithmic": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                              "mfma_busy_pmc": None if busy is None else round(busy, 4)})
        out["roofline"] = {"bound": "mfma", "kernel": names[dom], "achieved": a / 1e12, "peak": PEAK_MFMA_BF16 / 1e12,
                           "unit": "TFLOP/s", "frac": a / PEAK_MFMA_BF16, "traffic": traffic,
                           "algorithmic_bytes_per_launch": fam[dom]["bytes"] / fam[dom]["launches"],
                           "launches_per_step": fam[dom]["launches"], "ms_per_step": fam[dom]["ms"],
                           "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                            "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                        for k, v in fam.items()},
                           "templates": templates}
    else:
        ts._run_eager()       # keep the collectives of the instrumented step matched on every rank
    if multi:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_infer:
            del ts
            torch.cuda.empty_cache()
            out["infer"] = infer_bench(slice_mode=True)
            out["loader"] = loader_bench(cpu=not args.no_cpu_baseline)
            out["legacy"] = legacy_bench(cpu=not args.no_cpu_baseline)
            out["prep"] = prep_bench(cpu=not args.no_cpu_baseline)
            out["vgg256"] = vgg_bench()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
