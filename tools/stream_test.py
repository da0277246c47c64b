"""Do parallel branches of a captured graph overlap on this runtime?  4 independent chains of small kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
from reid_gan_amd.engine import side_stream

N, H, C = 128, 8, 64
bufs = [[ops.new_view(N, H, H, C, zero=True) for _ in range(2)] for _ in range(4)]

def chain(b, n=50):
    for i in range(n):
        ops.ew(ops.EW_COPY, bufs[b][i & 1], bufs[b][(i + 1) & 1])

def timeit(name, fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us", flush=True)

def serial():
    for b in range(4):
        chain(b)

def parallel():
    main = torch.cuda.current_stream()
    for b in range(1, 4):
        s = side_stream(b); s.wait_stream(main)
        with torch.cuda.stream(s):
            chain(b)
    chain(0)
    for b in range(1, 4):
        main.wait_stream(side_stream(b))

def forkjoin(n=25):
    """fork/join around every pair of kernels, like an Inception block with short branches"""
    main = torch.cuda.current_stream()
    for _ in range(n):
        for b in range(1, 4):
            s = side_stream(b); s.wait_stream(main)
            with torch.cuda.stream(s):
                chain(b, 2)
        chain(0, 2)
        for b in range(1, 4):
            main.wait_stream(side_stream(b))

timeit("one chain of 50", lambda: chain(0))
timeit("4 chains serial (200 kernels)", serial)
timeit("4 chains on 4 streams", parallel)
timeit("25 x fork/join of 4 x 2 kernels (200 kernels)", forkjoin)
