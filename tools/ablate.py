"""Critical-path attribution for the captured training step: time the step with one piece of it left out (the
results are then wrong - timing only).  The difference to the full step is what that piece costs on the critical
path, which is not its kernel time when it runs beside other work.
usage: ablate.py [case ...]   (default: every case)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from reid_gan_amd import ops
from reid_gan_amd.synth import synth_batch
from reid_gan_amd.train_step import TrainStep

g, d, vgg = bench.synth_states()
x, y = synth_batch(128, 64, seed=1234)


def timed(patch=None, reps=40):
    ts = TrainStep(g, d, vgg, 128, 64, 64, lr=1e-4, lookahead=not os.environ.get("RIED_NO_LOOKAHEAD"))
    ts.x.copy_(x.cuda()); ts.y.copy_(y.cuda())
    if patch:
        patch(ts)
    for _ in range(3):
        ts.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ts.step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def skip_nth(obj, name, which):
    """Replace obj.name so that the calls whose index (per step, counted from 0) is in `which` do nothing."""
    fn = getattr(obj, name)
    state = {"n": 0}
    def wrapper(*a, **k):
        i = state["n"]; state["n"] += 1
        if i in which:
            return None
        return fn(*a, **k)
    setattr(obj, name, wrapper)
    return state


def per_step(ts, states):
    body = ts._body
    def wrapped():
        for s in states:
            s["n"] = 0
        yield from body()
    ts._body = wrapped


def case_g_wgrads(ts):
    ts.G.ctx.end_backward = lambda bucket=None: None

def case_g_backward(ts):
    def bb(dout):
        yield 0
    ts.G.backward_buckets = bb

def case_g_forward(ts):
    ts.G.forward(ts.x)
    ts.G.forward = lambda x: ts.G.out

def case_vgg_target(ts):
    ts.V.forward(ts.y, ts.B, 2 * ts.B, src0=0)
    ts.V.forward = lambda *a, **k: None

def case_d_last_backward(ts):       # D.backward calls per step: it1 own, G-loss (data only), it2 own
    per_step(ts, [skip_nth(ts.D, "backward", {2})])

def case_d_first_backward(ts):
    per_step(ts, [skip_nth(ts.D, "backward", {0})])

def case_d_gloss_backward(ts):
    per_step(ts, [skip_nth(ts.D, "backward", {1})])

def case_d_forwards(ts):
    ts.D.forward(ts.d_in)
    ts.D.forward = lambda x: ts.D.out

def case_vgg_pred(ts):
    ts.V.forward_pred_fused = lambda *a, **k: None
    ts.V.backward = lambda *a, **k: None

def case_ssim(ts):
    import reid_gan_amd.train_step as T
    T.ops = type("O", (), {})()
    for k in dir(ops):
        setattr(T.ops, k, getattr(ops, k))
    T.ops.ssim_fwd = lambda *a, **k: None
    T.ops.ssim_bwd = lambda *a, **k: None

def case_g_bn_apply_fwd(ts):       # upper bound of what fusing the BatchNorm apply into the consumer would save
    from reid_gan_amd import engine
    real = ops.bn_apply_train
    def f(*a, **k):
        if k.get("accumulate"):
            return real(*a, **k)
    engine.ops = type("O", (), {})()
    for k in dir(ops):
        setattr(engine.ops, k, getattr(ops, k))
    engine.ops.bn_apply_train = f
    engine.ops.bn_apply_train_jobs = lambda *a, **k: None

def case_sn_and_packs(ts):
    ts.D.ctx.sn_sigmas = lambda *a, **k: None
    ts.D.ctx.pack_all = lambda *a, **k: None
    ts.D.ctx.sn_finish_wgrads = lambda *a, **k: None

cases = {k[5:]: v for k, v in globals().items() if k.startswith("case_")}
want = sys.argv[1:] or list(cases)
base = timed()
print(f"full step: {base:.3f} ms", flush=True)
for name in want:
    t = timed(cases[name])
    print(f"without {name:18s}: {t:.3f} ms  (critical-path cost {base - t:+.3f} ms)", flush=True)
    import reid_gan_amd.train_step as T
    from reid_gan_amd import engine
    T.ops = ops
    engine.ops = ops
