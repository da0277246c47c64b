"""Seeded synthetic stand-ins for the .npy slice files that src/prep.py:24-25 writes (int16 HU, one file per
slice and role), for tests and benchmarks: there is no patient data offline."""
import os

import numpy as np


def synth_slices(n: int, h: int, w: int, seed: int):
    """(inputs, targets): two int16 arrays [n][h][w] with CT-like statistics (HU mean about -470, std about 520)."""
    r = np.random.default_rng(seed)
    tg = np.clip(-470.0 + 520.0 * r.standard_normal((n, h, w)), -1024, 3071)
    inp = np.clip(tg + 30.0 * r.standard_normal((n, h, w)), -1024, 3071)
    return inp.astype(np.int16), tg.astype(np.int16)


def write_slice_files(path: str, patients, per_patient: int, h: int, w: int, seed: int):
    """Files named {patient}_{idx}_{input|target}.npy as src/prep.py:23 names them."""
    os.makedirs(path, exist_ok=True)
    for pi, patient in enumerate(patients):
        inp, tg = synth_slices(per_patient, h, w, seed + pi)
        for i in range(per_patient):
            np.save(os.path.join(path, f"{patient}_{i}_input.npy"), inp[i])
            np.save(os.path.join(path, f"{patient}_{i}_target.npy"), tg[i])
