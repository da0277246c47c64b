"""The perceptual-loss configuration of bench.py (256 prediction + 256 target patches of 256 x 256 through the VGG16
taps, five-tap MSE, data gradient) for rocprofv3 --kernel-trace: 1 + 4 passes; summarise with prof_summary.py DIR 5."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(bench.vgg_bench(iters=4))
