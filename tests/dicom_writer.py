"""Test helper: writes minimal DICOM part-10 CT slices (uncompressed little endian, explicit or implicit VR) with
the elements src/prep.py reads, a nested undefined-length sequence in front of them and the pixel data last."""
import struct

import numpy as np

_LONG = {b"OB", b"OW", b"SQ", b"UN", b"UT"}


def _el(g, e, vr, value, explicit):
    if len(value) % 2:
        value += b" " if vr not in (b"OB", b"OW", b"UI") else b"\x00"
    if not explicit:
        return struct.pack("<HHI", g, e, len(value)) + value
    if vr in _LONG:
        return struct.pack("<HH", g, e) + vr + b"\x00\x00" + struct.pack("<I", len(value)) + value
    return struct.pack("<HH", g, e) + vr + struct.pack("<H", len(value)) + value


def _undefined_sq(g, e, explicit, inner):
    """An undefined-length sequence holding one undefined-length item with `inner` elements."""
    head = struct.pack("<HH", g, e) + (b"SQ\x00\x00" if explicit else b"") + struct.pack("<I", 0xFFFFFFFF)
    item = struct.pack("<HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + inner + struct.pack("<HHI", 0xFFFE, 0xE00D, 0)
    return head + item + struct.pack("<HHI", 0xFFFE, 0xE0DD, 0)


def write_slice(path, pixels, z, slope=1.0, intercept=-1024.0, explicit=True, signed=False,
                syntax=None, with_position=True):
    pixels = np.ascontiguousarray(pixels, dtype=np.int16 if signed else np.uint16)
    ts = syntax or ("1.2.840.10008.1.2.1" if explicit else "1.2.840.10008.1.2")
    meta = _el(0x0002, 0x0010, b"UI", ts.encode(), True)
    meta = _el(0x0002, 0x0000, b"UL", struct.pack("<I", len(meta)), True) + meta
    ds = b""
    ds += _el(0x0008, 0x0060, b"CS", b"CT", explicit)
    nested = _undefined_sq(0x0008, 0x1140, explicit, _el(0x0008, 0x1150, b"UI", b"1.2.3", explicit))
    ds += _undefined_sq(0x0008, 0x9121, explicit, _el(0x0008, 0x0100, b"SH", b"X", explicit) + nested)
    ds += _el(0x0018, 0x0050, b"DS", b"3.0", explicit)
    if with_position:
        ds += _el(0x0020, 0x0032, b"DS", f"-160.5\\-280.25\\{z}".encode(), explicit)
    ds += _el(0x0020, 0x1041, b"DS", f"{z}".encode(), explicit)
    ds += _el(0x0028, 0x0002, b"US", struct.pack("<H", 1), explicit)
    ds += _el(0x0028, 0x0010, b"US", struct.pack("<H", pixels.shape[0]), explicit)
    ds += _el(0x0028, 0x0011, b"US", struct.pack("<H", pixels.shape[1]), explicit)
    ds += _el(0x0028, 0x0100, b"US", struct.pack("<H", 16), explicit)
    ds += _el(0x0028, 0x0103, b"US", struct.pack("<H", int(signed)), explicit)
    ds += _el(0x0028, 0x1052, b"DS", f"{intercept}".encode(), explicit)
    ds += _el(0x0028, 0x1053, b"DS", f"{slope}".encode(), explicit)
    ds += _el(0x7FE0, 0x0010, b"OW", pixels.tobytes(), explicit)
    with open(path, "wb") as f:
        f.write(b"\x00" * 128 + b"DICM" + meta + ds)
