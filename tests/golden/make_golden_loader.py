"""Generate tests/golden/loader.npz by importing the reference's src/loader.py.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_loader.py

Seeded synthetic int16 slices (reid_gan_amd.synth_slices) are written to a temporary directory in the file layout
of src/prep.py:23-25; the reference's get_patch and get_loader run on them with numpy's global generator seeded;
the patches they return are stored.  No reference source is copied; only numbers are stored.
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import numpy as np

import loader           # reference
from reid_gan_amd.synth_slices import synth_slices, write_slice_files

arrs = {}
# get_patch alone
inp, tg = synth_slices(1, 96, 80, seed=41)
np.random.seed(4242)
px, py = loader.get_patch(inp[0], tg[0], 5, 32)
arrs["gp_x"], arrs["gp_y"] = px, py

# the whole loader: 3 patients x 3 slices of 96x96, the test patient held out, 4 patches of 32 per slice, batches
# of 4 slices (so the second batch is short: 6 training slices)
with tempfile.TemporaryDirectory() as d:
    write_slice_files(d, ["L019", "L064", "L150"], 3, 96, 96, seed=50)
    np.random.seed(777)
    dl = loader.get_loader(mode="train", load_mode=1, saved_path=d, test_patient="L064", patch_n=4, patch_size=32,
                           transform=None, batch_size=4, num_workers=0)
    for bi, (x, y) in enumerate(dl):
        arrs[f"train_x{bi}"], arrs[f"train_y{bi}"] = x.numpy(), y.numpy()
    arrs["train_batches"] = np.array([bi + 1])
    dl = loader.get_loader(mode="test", load_mode=0, saved_path=d, test_patient="L064", patch_n=None, patch_size=None,
                           transform=None, batch_size=2, num_workers=0)
    for bi, (x, y) in enumerate(dl):
        arrs[f"test_x{bi}"], arrs[f"test_y{bi}"] = x.numpy(), y.numpy()
    arrs["test_batches"] = np.array([bi + 1])
np.savez_compressed(os.path.join(HERE, "loader.npz"), **arrs)
print("wrote loader.npz", {k: v.shape for k, v in arrs.items()})
