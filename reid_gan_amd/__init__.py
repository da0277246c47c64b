"""MI355X-native RIED-GAN hot path (generator, discriminator, losses, metrics, training step)."""
import os

# The captured training step is a graph of parallel paths (Inception branches, the discriminator's backward beside
# the generator's); the HIP runtime spreads the streams that replay them over GPU_MAX_HW_QUEUES hardware queues per
# process (default 4), in the order the process created its streams.  Measured on MI355X with tools/graph_modes.py:
# with 4 queues every capture in a process replays in 8.05 ms; with 8 or 16 the same step takes 7.95, 10.6, 12.0 or
# 13.9 ms depending on how many streams the process had created before the capture (one more stream is enough to
# change it).  The default is therefore left alone; RIED_HW_QUEUES=<n> sets it for experiments.  It is read by the
# runtime when it initialises, so this must run before the first HIP call of the process.
_q = os.environ.get("RIED_HW_QUEUES", "0")
if _q != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = _q
