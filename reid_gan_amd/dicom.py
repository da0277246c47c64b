"""A reader for the DICOM CT slices the reference's src/prep.py opens with pydicom.read_file (src/prep.py:35).

Only what prep.py touches is decoded: ImagePositionPatient, SliceLocation, RescaleIntercept, RescaleSlope, the image
geometry and the stored pixel values (`pixel_array`).  Supported: DICOM part-10 files with the uncompressed
little-endian transfer syntaxes (explicit VR 1.2.840.10008.1.2.1 - what the AAPM low-dose CT files are - and
implicit VR 1.2.840.10008.1.2), 16 bits allocated, one sample per pixel.  Anything else raises: there is no
silent guess.  Sequences are skipped structurally (defined or undefined length, nested).
"""
from __future__ import annotations

import struct
from typing import List, Optional

import numpy as np

_LONG_VR = {b"OB", b"OW", b"OF", b"SQ", b"UT", b"UN", b"OD", b"OL", b"UC", b"UR", b"OV", b"SV", b"UV"}
_UNDEF = 0xFFFFFFFF
_EXPLICIT_LE, _IMPLICIT_LE = "1.2.840.10008.1.2.1", "1.2.840.10008.1.2"
_ITEM, _ITEM_END, _SEQ_END = (0xFFFE, 0xE000), (0xFFFE, 0xE00D), (0xFFFE, 0xE0DD)


class DicomError(ValueError):
    pass


def _header(b: bytes, pos: int, explicit: bool):
    """(group, element, vr, value length, header size) of the element at pos."""
    if pos + 8 > len(b):
        raise DicomError("truncated element header")
    g, e = struct.unpack_from("<HH", b, pos)
    if (g, e) in (_ITEM, _ITEM_END, _SEQ_END) or not explicit:
        return g, e, b"", struct.unpack_from("<I", b, pos + 4)[0], 8
    vr = b[pos + 4:pos + 6]
    if vr in _LONG_VR:
        if pos + 12 > len(b):
            raise DicomError("truncated element header")
        return g, e, vr, struct.unpack_from("<I", b, pos + 8)[0], 12
    return g, e, vr, struct.unpack_from("<H", b, pos + 6)[0], 8


def _skip_sequence(b: bytes, pos: int, explicit: bool) -> int:
    """pos: first byte after the header of an undefined-length sequence; returns the position after its end."""
    while True:
        g, e, _, ln, hs = _header(b, pos, explicit)
        pos += hs
        if (g, e) == _SEQ_END:
            return pos
        if (g, e) != _ITEM:
            raise DicomError(f"expected a sequence item, found ({g:04x},{e:04x})")
        if ln != _UNDEF:
            pos += ln
            continue
        while True:                                   # elements of an undefined-length item
            g, e, vr, ln, hs = _header(b, pos, explicit)
            pos += hs
            if (g, e) == _ITEM_END:
                break
            if ln == _UNDEF:
                pos = _skip_sequence(b, pos, explicit)
            else:
                pos += ln


def _text(v: bytes) -> str:
    return v.decode("ascii", "replace").strip(" \x00")


class Slice:
    """The attributes of a pydicom dataset that src/prep.py:33-59 reads or writes."""

    def __init__(self):
        self.ImagePositionPatient: Optional[List[float]] = None
        self.SliceLocation: Optional[float] = None
        self.RescaleIntercept: float = 0.0
        self.RescaleSlope: float = 1.0
        self.SliceThickness: Optional[float] = None
        self.Rows = self.Columns = 0
        self.BitsAllocated, self.PixelRepresentation, self.SamplesPerPixel = 16, 0, 1
        self.TransferSyntaxUID = _IMPLICIT_LE
        self.pixel_array: Optional[np.ndarray] = None
        self.filename = ""


def read_file(path: str) -> Slice:
    b = open(path, "rb").read()
    if len(b) < 132 or b[128:132] != b"DICM":
        raise DicomError(f"{path}: not a DICOM part-10 file (no DICM marker)")
    s = Slice()
    s.filename = path
    pos = 132
    while pos < len(b):                               # file meta information: always explicit VR little endian
        g, e, vr, ln, hs = _header(b, pos, True)
        if g != 0x0002:
            break
        if (g, e) == (0x0002, 0x0010):
            s.TransferSyntaxUID = _text(b[pos + hs:pos + hs + ln])
        pos += hs + ln
    if s.TransferSyntaxUID not in (_EXPLICIT_LE, _IMPLICIT_LE):
        raise DicomError(f"{path}: transfer syntax {s.TransferSyntaxUID} is not uncompressed little endian")
    explicit = s.TransferSyntaxUID == _EXPLICIT_LE
    pixels = None
    while pos < len(b):
        g, e, vr, ln, hs = _header(b, pos, explicit)
        pos += hs
        if (g, e) == (0x7FE0, 0x0010):
            if ln == _UNDEF:
                raise DicomError(f"{path}: encapsulated (compressed) pixel data")
            pixels = (pos, ln)
            break
        if ln == _UNDEF:
            pos = _skip_sequence(b, pos, explicit)
            continue
        v = b[pos:pos + ln]
        pos += ln
        if (g, e) == (0x0020, 0x0032):
            s.ImagePositionPatient = [float(t) for t in _text(v).split("\\")]
        elif (g, e) == (0x0020, 0x1041):
            s.SliceLocation = float(_text(v))
        elif (g, e) == (0x0018, 0x0050) and _text(v):
            s.SliceThickness = float(_text(v))
        elif (g, e) == (0x0028, 0x1052):
            s.RescaleIntercept = float(_text(v))
        elif (g, e) == (0x0028, 0x1053):
            s.RescaleSlope = float(_text(v))
        elif g == 0x0028 and e in (0x0002, 0x0010, 0x0011, 0x0100, 0x0103) and ln == 2:
            name = {0x0002: "SamplesPerPixel", 0x0010: "Rows", 0x0011: "Columns", 0x0100: "BitsAllocated",
                    0x0103: "PixelRepresentation"}[e]
            setattr(s, name, struct.unpack("<H", v)[0])
    if pixels is None:
        raise DicomError(f"{path}: no pixel data element")
    if s.BitsAllocated != 16 or s.SamplesPerPixel != 1:
        raise DicomError(f"{path}: only 16-bit single-sample images are supported")
    off, ln = pixels
    n = s.Rows * s.Columns
    if n == 0 or ln < 2 * n or off + 2 * n > len(b):
        raise DicomError(f"{path}: pixel data shorter than Rows x Columns")
    dt = np.dtype("<i2") if s.PixelRepresentation else np.dtype("<u2")
    s.pixel_array = np.frombuffer(b, dtype=dt, count=n, offset=off).reshape(s.Rows, s.Columns)
    return s
