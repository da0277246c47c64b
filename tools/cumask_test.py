"""Prototype: eager training steps with the chip split by CU masks while the generator backward (a chain of small
dependent kernels) runs beside the last discriminator backward (large kernels): the chain on G_CUS compute units of
its own, the discriminator side on the rest, the generator's grouped weight gradients afterwards on the whole chip.
usage: cumask_test.py [G_CUS ...]   (0 = no masks)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd import engine, ops, train_step
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use."""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


g, d, vgg = bench.synth_states()
x, y = synth_batch(128, 64, seed=1234)
ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4, lookahead=True, use_graph=False)
ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
orig_side = engine.side_stream
orig_gb = TrainStep._g_backward
state = {"g": None, "d": None}


def side_stream(lane):
    if lane == train_step.DBWD_LANE and state["d"] is not None:
        return state["d"]
    return orig_side(lane)


def g_backward(self, side=None):
    if state["g"] is None:
        yield from orig_gb(self, side)
        return
    main = torch.cuda.current_stream()
    gs = state["g"]
    gs.wait_stream(main)
    G = self.G
    with torch.cuda.stream(gs):
        self.gs.zero_grad()
        G.ctx.zero_backward()
        ops.out_conv_bwd(G.x, G.hlast.v, G.st.p["out_conv.weight"], self.dpred, G.hlast.g, G.st.g["out_conv.weight"])
        for l in reversed(G.body.layers):
            l.backward()
        G.conv1.backward()
    main.wait_stream(gs)
    G.ctx.end_backward(0)          # the grouped weight gradients: large kernels, whole chip
    if side is not None:
        self._join(side)
    yield []


train_step.side_stream = side_stream
engine.side_stream = side_stream
TrainStep._g_backward = g_backward


def timed(reps=30):
    for _ in range(5):
        ts.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ts.step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for a in (sys.argv[1:] or ["0", "32", "48", "64", "96", "0"]):
    k = int(a)
    if k:
        state["g"] = masked_stream(range(k))
        state["d"] = masked_stream(range(k, 256))
    else:
        state["g"] = state["d"] = None
    print("generator chain on", k, "CUs: %.3f ms per eager step" % timed(), " gen loss %.4f" % ts.losses()["gen"], flush=True)
