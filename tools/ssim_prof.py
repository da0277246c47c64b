"""The metrics kernel alone (three images of 32 slices of 512 x 512) for rocprofv3 counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
n, s = 32, 512
g = torch.Generator(device="cuda").manual_seed(1)
x, y, p = (torch.randn(n, 1, s, s, device="cuda", generator=g) * 100 - 500 for _ in range(3))
sums = torch.zeros(n, 4, device="cuda")
for _ in range(5):
    ops.ssim_fwd(x, p, y, (0.01 * 400) ** 2, (0.03 * 400) ** 2, sums, clamp=(-160.0, 240.0))
torch.cuda.synchronize()
print(sums[0].tolist())
