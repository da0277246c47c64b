"""Does a reader launched after a writer on ONE stream ever see stale data while other streams are busy?  (stock torch
kernels only: a check of the runtime, not of this library)"""
import torch
dev = "cuda"
n_streams = 6
streams = [torch.cuda.Stream() for _ in range(n_streams)]
a = torch.zeros(4 * 64 * 64 * 64, device=dev, dtype=torch.bfloat16)
b = torch.zeros_like(a)
bad = torch.zeros(1, device=dev, dtype=torch.int64)
m1 = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
m2 = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
big = torch.zeros(64 * 1024 * 1024, device=dev)
main = torch.cuda.current_stream()
for it in range(3000):
    v = float(it % 200 + 1)
    for s in streams[1:4]:
        s.wait_stream(main)
        with torch.cuda.stream(s):
            (m1 @ m2); big.add_(1.0)
    s = streams[0]
    s.wait_stream(main)
    with torch.cuda.stream(s):
        a.fill_(v)
        b.copy_(a)
        bad += (b != v).sum()
    for t in streams[:4]:
        main.wait_stream(t)
torch.cuda.synchronize()
print("stale elements seen:", int(bad.item()))
