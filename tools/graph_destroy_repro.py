"""Minimal sequence for the host crash seen after DESTROYING a captured multi-stream graph while a later capture is
alive (why train_step used to retain its graphs): capture A, capture B, destroy A, replay B, capture C, replay C.
usage: graph_destroy_repro.py torch|step [pool]
  torch: graphs of plain torch kernels forked over two side streams (no library code of this package involved)
  step : graphs of the real training step (TrainStep) at batch 8
  pool : B and C are captured into A's memory pool (as the segmented step chains its graphs)
faulthandler prints the Python stack of a fatal signal; the exit code says how the run ended."""
import sys, os, gc, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
share_pool = len(sys.argv) > 2 and sys.argv[2] == "pool"
print("mode", mode, "shared pool", share_pool, flush=True)

if mode == "torch":
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    x = torch.randn(1 << 20, device="cuda")
    outs = [torch.empty_like(x) for _ in range(3)]

    def body():
        main = torch.cuda.current_stream()
        torch.mul(x, 2.0, out=outs[0])
        for i, s in enumerate(side):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                torch.add(outs[0], float(i), out=outs[1 + i])
        for s in side:
            main.wait_stream(s)
        torch.add(outs[1], outs[2], out=outs[0])

    def capture(pool=None):
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            body()
        return g
    A = capture()
    B = capture(A.pool() if share_pool else None)
    A.replay(); B.replay(); torch.cuda.synchronize()
    print("destroying A", flush=True)
    del A; gc.collect(); torch.cuda.synchronize()
    print("replaying B", flush=True)
    B.replay(); torch.cuda.synchronize()
    print("capturing C", flush=True)
    C = capture(B.pool() if share_pool else None)
    print("replaying C", flush=True)
    C.replay(); torch.cuda.synchronize()
    print("no crash", flush=True)
else:
    import bench
    import reid_gan_amd.train_step as T
    from reid_gan_amd.synth import synth_batch
    g, d, vgg = bench.synth_states()
    x, y = synth_batch(8, 64, seed=1)

    def make():
        ts = T.TrainStep(g, d, vgg, 8, 64, 64, lr=1e-4)
        ts.step(x.cuda(), y.cuda()); ts.step(); torch.cuda.synchronize()
        return ts
    A = make(); B = make()
    print("destroying A", flush=True)
    del A; gc.collect(); torch.cuda.synchronize()
    print("replaying B", flush=True)
    B.step(); torch.cuda.synchronize()
    print("capturing C", flush=True)
    C = make()
    print("replaying B and C", flush=True)
    B.step(); C.step(); torch.cuda.synchronize()
    print("destroying B, replaying C", flush=True)
    del B; gc.collect(); torch.cuda.synchronize()
    C.step(); torch.cuda.synchronize()
    print("no crash", flush=True)
