"""Times ried_bn_apply_train alone on the layer shapes of the 512 x 512 inference batch (32 slices) and of the
training step, and prints the achieved GB/s (one read and one write of the tensor).
RIED_BN_SLICE_BLOCKS=0 gives the fixed two passes per block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops

SHAPES = [(32 * 256 * 256, 32), (32 * 128 * 128, 64), (32 * 64 * 64, 32), (32 * 64 * 64, 64), (32 * 64 * 64, 96),
          (32 * 64 * 64, 128), (32 * 64 * 64, 192), (32 * 64 * 64, 224), (32 * 128 * 128, 96), (128 * 32 * 32, 32),
          (128 * 8 * 8, 64), (128 * 8 * 8, 192), (128 * 4 * 4, 224)]
for npix, C in SHAPES:
    x = ops.new_view(1, 1, npix, C)
    x.torch().normal_()
    y = ops.new_view(1, 1, npix, C)
    rows = ops.bn_fold_rows()
    stats = torch.rand(rows * 2 * C, device="cuda") * npix / rows
    save = torch.empty(2 * C, device="cuda")
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for acc in (False, True):
        for _ in range(5):
            ops.bn_apply_train(x, y, stats, rows, save, gamma, beta, accumulate=acc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.bn_apply_train(x, y, stats, rows, save, gamma, beta, accumulate=acc)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        gb = npix * C * 2 * (3 if acc else 2) / 1e9
        print(f"npix {npix:8d} C {C:4d} acc {int(acc)}  {us:7.1f} us  {gb / us * 1e6:7.0f} GB/s", flush=True)
