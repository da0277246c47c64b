import sys, os, torch
sys.path.insert(0, "/root/repo")
from reid_gan_amd import ops
N, H, C = 256, 64, 64
x = torch.randn(N, 1, H, H, device="cuda")
dy = ops.new_view(N, H, H, C); dy.base.normal_()
dw = torch.zeros(C, 9, device="cuda")
scr = torch.empty(ops.conv_c1_wgrad_scratch_floats(C), device="cuda")
y = ops.new_view(N, H, H, C)
w = torch.randn(C, 9, device="cuda")
dimg = torch.zeros(N, 1, H, H, device="cuda")
def t(name, f, reps=20):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) * 1e3 / reps, "us")
t("c1_wgrad", lambda: ops.conv_c1_wgrad(x, dy, 1, dw, scr))
t("c1_fwd", lambda: ops.conv_c1_fwd(x, w, y, 1, bias=None, act=2))
t("c1_dgrad", lambda: ops.conv_c1_dgrad(dy, w, dimg, accumulate=True))
h = ops.new_view(128, 64, 64, 32); h.base.normal_()
dh = ops.new_view(128, 64, 64, 32)
xo = torch.randn(128, 1, 64, 64, device="cuda"); do = torch.randn(128, 1, 64, 64, device="cuda")
wo = torch.randn(1, 33, 1, 1, device="cuda"); dwo = torch.zeros(33, device="cuda")
t("out_conv_bwd", lambda: ops.out_conv_bwd(xo, h, wo, do, dh, dwo))
