"""DICOM decode and the HU conversion of src/prep.py:33-59 on the CPU side: the reader on files this test writes
(both uncompressed transfer syntaxes, nested undefined-length sequences), the oracle's get_pixels_hu on
hand-computed cases, and - where the reference's shipped data is present - the reader on its files."""
import os

import numpy as np
import pytest

from oracle import ried_oracle as O
from reid_gan_amd import dicom
from tests.dicom_writer import write_slice


@pytest.mark.parametrize("explicit", [True, False])
@pytest.mark.parametrize("signed", [False, True])
def test_reader_roundtrip(tmp_path, explicit, signed):
    r = np.random.default_rng(3)
    px = r.integers(0, 4000, (16, 24)).astype(np.int16 if signed else np.uint16)
    if signed:
        px[0, 0] = -2000
    p = str(tmp_path / "a.dcm")
    write_slice(p, px, -361.5, slope=1.5, intercept=-1000.0, explicit=explicit, signed=signed)
    s = dicom.read_file(p)
    assert np.array_equal(s.pixel_array, px) and s.pixel_array.dtype == px.dtype
    assert s.ImagePositionPatient == [-160.5, -280.25, -361.5] and s.SliceLocation == -361.5
    assert (s.RescaleSlope, s.RescaleIntercept, s.Rows, s.Columns) == (1.5, -1000.0, 16, 24)
    assert s.PixelRepresentation == int(signed) and s.SliceThickness == 3.0


def test_reader_refuses_what_it_cannot_decode(tmp_path):
    px = np.zeros((8, 8), np.uint16)
    p = str(tmp_path / "jpeg.dcm")
    write_slice(p, px, 0.0, syntax="1.2.840.10008.1.2.4.70")          # JPEG lossless
    with pytest.raises(dicom.DicomError, match="transfer syntax"):
        dicom.read_file(p)
    q = str(tmp_path / "notdicom.dcm")
    open(q, "wb").write(b"\x00" * 200)
    with pytest.raises(dicom.DicomError, match="DICM"):
        dicom.read_file(q)
    write_slice(p, px, 0.0)
    data = open(p, "rb").read()
    open(p, "wb").write(data[:-40])                                     # pixel data cut short
    with pytest.raises(dicom.DicomError):
        dicom.read_file(p)


def test_load_scan_sorts_by_position_and_sets_thickness(tmp_path):
    from reid_gan_amd import prep
    zs = [12.0, -3.0, 4.5, 0.0]
    for i, z in enumerate(zs):
        write_slice(str(tmp_path / f"1-{i:03d}.dcm"), np.full((8, 8), i, np.uint16), z)
    slices = prep.load_scan(str(tmp_path))
    assert [s.ImagePositionPatient[2] for s in slices] == sorted(zs)
    assert [int(s.pixel_array[0, 0]) for s in slices] == [1, 3, 2, 0]
    assert all(s.SliceThickness == 3.0 for s in slices)               # |z0 - z1| of the sorted series
    assert np.allclose(prep.normalize_(np.array([-1024.0, 3072.0, 1024.0])), [0.0, 1.0, 0.5])


def test_oracle_hu_known_answers():
    a = np.array([[0, 1000, 63536, 2000]], dtype=np.uint16)            # 63536 is -2000 as int16
    out = O.get_pixels_hu([a, a, a], [1.0, 2.0, 0.5], [-1024.0, -1024.0, 10.0])
    assert out.dtype == np.int16
    assert out[0].tolist() == [[-1024, -24, -1024, 976]]              # -2000 -> 0 (air outside the field of view)
    assert out[1].tolist() == [[-1024, 976, -1024, 2976]]
    assert out[2].tolist() == [[10, 510, 10, 1010]]
    # truncation toward zero of slope * value, and 16-bit wrap of the intercept add
    b = np.array([[3, -3, 32767]], dtype=np.int16)
    out = O.get_pixels_hu([b, b], [0.5, 1.0], [0.0, 1.0])
    assert out[0].tolist() == [[1, -1, 16383]]
    assert out[1].tolist() == [[4, -2, -32768]]


REF = "/root/reference/patient"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's shipped DICOM data is not on this machine")
def test_reader_on_the_reference_data():
    """One of the shipped AAPM slices: geometry and rescale tags, and the HU statistics of that slice."""
    s = dicom.read_file(os.path.join(REF, "L064", "quarter_3mm", "1-101.dcm"))
    assert s.pixel_array.shape == (512, 512) and s.pixel_array.dtype == np.uint16
    assert (s.RescaleSlope, s.RescaleIntercept) == (1.0, -1024.0) and len(s.ImagePositionPatient) == 3
    hu = O.get_pixels_hu([s.pixel_array], [s.RescaleSlope], [s.RescaleIntercept])[0].astype(np.float64)
    assert abs(hu.mean() + 469.0) < 1.0 and abs(hu.std() - 520.7) < 1.0 and hu.min() == -1024.0
    from reid_gan_amd import prep
    series = prep.load_scan(os.path.join(REF, "L064", "quarter_3mm"))
    z = [x.ImagePositionPatient[2] for x in series]
    assert len(series) == 208 and z == sorted(z) and len(set(z)) == len(z)
