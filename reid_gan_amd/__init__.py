"""MI355X-native RIED-GAN hot path (generator, discriminator, losses, metrics, training step)."""
import os

# The captured training step is a graph of parallel paths (Inception branches, the discriminator's backward beside
# the generator's); the HIP runtime spreads them over GPU_MAX_HW_QUEUES hardware queues per process (default 4).
# Measured on MI355X: 8 queues 8.41 ms per step against 8.51 with 4 (6, 12 and 16 are slower: 15.7 / 11.6 / 11.6 ms).
# Read by the runtime when it initialises, so this must run before the first HIP call of the process;
# RIED_HW_QUEUES overrides (0: leave the environment alone).
_q = os.environ.get("RIED_HW_QUEUES", "8")
if _q != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = _q
