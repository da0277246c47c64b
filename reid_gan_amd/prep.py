"""DICOM series -> int16 Hounsfield-unit slices, with the reference's names (src/prep.py): load_scan,
get_pixels_hu, normalize_, save_dataset.

The reference decodes with pydicom and converts with numpy on the host.  Here the files are decoded by
reid_gan_amd.dicom (no pydicom), the stored pixel values of a whole series go to the GPU once, and
ried_hu_convert applies the reference's conversion there (src/prep.py:46-59, bit for bit).  get_pixels_hu returns
the numpy array the reference returns; get_pixels_hu_device keeps the result in HBM in the layout the patch
loader reads, so a series can go from .dcm files to training patches without a host round trip or .npy files.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch

from . import dicom
from ._lib import lib
from .ops import _p, _stream, require_gpu


def load_scan(path: str) -> List[dicom.Slice]:
    """src/prep.py:33-43: every file of the directory, sorted by the z of ImagePositionPatient; SliceThickness is
    set to the spacing of the first two slices."""
    slices = [dicom.read_file(os.path.join(path, s)) for s in os.listdir(path)]
    slices.sort(key=lambda x: float(x.ImagePositionPatient[2]))
    try:
        slice_thickness = np.abs(slices[0].ImagePositionPatient[2] - slices[1].ImagePositionPatient[2])
    except Exception:
        slice_thickness = np.abs(slices[0].SliceLocation - slices[1].SliceLocation)
    for s in slices:
        s.SliceThickness = slice_thickness
    return slices


def get_pixels_hu_device(slices: Sequence[dicom.Slice], device="cuda") -> torch.Tensor:
    """int16 (n, H, W) device tensor of Hounsfield units (src/prep.py:46-59)."""
    require_gpu()
    if len(slices) == 0:
        raise ValueError("need at least one slice to stack")            # np.stack([]) raises in the reference
    image = np.stack([s.pixel_array for s in slices])                    # raises on ragged shapes, as there
    if image.dtype.itemsize != 2:
        raise ValueError("16-bit stored values expected")
    n, h, w = image.shape
    if (h * w) % 8:
        raise ValueError("slices must hold a multiple of 8 pixels")
    raw = torch.from_numpy(np.ascontiguousarray(image).view(np.int16)).to(device)   # astype(int16) keeps the bits
    slope = torch.tensor([float(s.RescaleSlope) for s in slices], dtype=torch.float64, device=device)
    icpt = torch.tensor([int(np.int16(s.RescaleIntercept)) for s in slices], dtype=torch.int32, device=device)
    out = torch.empty(n, h, w, dtype=torch.int16, device=device)
    lib.ried_hu_convert(_p(raw), _p(slope), _p(icpt), _p(out), n, h * w, _stream())
    return out


def get_pixels_hu(slices: Sequence[dicom.Slice]) -> np.ndarray:
    return get_pixels_hu_device(slices).cpu().numpy()


def normalize_(image, MIN_B=-1024.0, MAX_B=3072.0):
    """src/prep.py:62-64 (not applied by save_dataset there either)."""
    return (image - MIN_B) / (MAX_B - MIN_B)


PATIENTS = ['L019', 'L033', 'L058', 'L064', 'L077', 'L114', 'L116', 'L134', 'L145', 'L150']      # src/prep.py:12


def load_patient_device(data_path: str, patient: str, mm: int = 3, device="cuda"):
    """(input, target) int16 (n, H, W) device tensors of one patient's quarter-dose and full-dose series
    (directory names as at src/prep.py:14-17)."""
    out = []
    for dose in ("quarter", "full"):
        out.append(get_pixels_hu_device(load_scan(os.path.join(data_path, patient, f"{dose}_{mm}mm")), device))
    return tuple(out)


def save_dataset(args):
    """src/prep.py:7-30: writes {patient}_{i}_{input|target}.npy for every slice of every patient."""
    if not os.path.exists(args.save_path):
        os.makedirs(args.save_path)
        print('Create path : {}'.format(args.save_path))
    for p_ind, patient in enumerate(PATIENTS):
        for dose, io in (("quarter", "input"), ("full", "target")):
            path_ = os.path.join(args.data_path, patient, "{}_{}mm".format(dose, args.mm))
            full_pixels = get_pixels_hu(load_scan(path_))
            for pi in range(len(full_pixels)):
                np.save(os.path.join(args.save_path, '{}_{}_{}.npy'.format(patient, pi, io)), full_pixels[pi])
        printProgressBar(p_ind, len(PATIENTS), prefix='save image ..', suffix='Complete', length=25)
        print(' ')


def printProgressBar(iteration, total, prefix='', suffix='', decimals=1, length=100, fill=' '):
    percent = ("{0:." + str(decimals) + "f}").format(100 * (iteration / float(total)))
    filled = int(length * iteration // total)
    print('\r%s |%s| %s%% %s' % (prefix, fill * filled + '=' * (length - filled), percent, suffix), end='\r')
    if iteration == total:
        print()
