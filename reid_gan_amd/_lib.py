"""ctypes binding of libried_gan.so, generated from the prototypes in include/ried_gan.h.

The product path has no fallback: if the shared library is missing the import
of anything that computes raises, so a silent eager/CPU path cannot stand in
for the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RIED_LIB_PATH") or os.path.join(_PKG, "libried_gan.so")     # (another build, for A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "ried_gan.h")

_CTYPES = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "ried_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """{name: (return type, [(ctype string, arg name), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|long long|int)\s+(ried_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        out = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.*?)(\w+)$", a)
                out.append((am.group(1).strip(), am.group(2)))
        protos[name] = (ret, out)
    return protos


def _ctype_of(t: str):
    if "*" in t:      # every pointer, whatever it points to, is passed as an address
        return ctypes.c_void_p
    t = t.replace("const ", "").strip()
    return _CTYPES[t]


_VALUE_SUFFIXES = ("_rows", "_floats", "_version", "_cus", "_tiles", "_bytes", "_supported", "_ok", "_per_block")


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with ./build.sh (hipcc --offload-arch=gfx950). "
                    "There is no fallback path.")
            dll = ctypes.CDLL(LIB_PATH)
            for name, (ret, args) in self.protos.items():
                fn = getattr(dll, name)          # AttributeError if the header declares what the .so lacks
                fn.restype = (ctypes.c_char_p if "char" in ret else
                              ctypes.c_longlong if ret == "long long" else ctypes.c_int)
                fn.argtypes = [_ctype_of(t) for t, _ in args]
            self._dll = dll
            self._set_ssim_window(dll)
            if os.environ.get("RIED_FORCE_TILE"):      # tuning hook, e.g. "0,0,-1" (see ried_conv2d_force_tile)
                dll.ried_conv2d_force_tile(*[int(v) for v in os.environ["RIED_FORCE_TILE"].split(",")])
        return self._dll

    @staticmethod
    def _set_ssim_window(dll):
        """The SSIM kernels get the Gaussian window exactly as the reference builds it (src/loss.py:24-26: a torch
        fp32 tensor divided by its torch sum), bit for bit: on raw HU the loss value depends on the last bit of
        the weights' sum."""
        import torch
        from math import exp
        g = torch.Tensor([exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
        g = (g / g.sum()).contiguous()
        if dll.ried_ssim_set_window(ctypes.c_void_p(g.data_ptr()), 11) != 0:
            raise RuntimeError("ried_ssim_set_window failed")

    def call(self, name: str, *args):
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if name.endswith(_VALUE_SUFFIXES):       # size / count queries return their value, not a status
            return rc
        if rc != 0:
            msg = dll.ried_last_error()
            raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")

    def __getattr__(self, name):
        if name.startswith("ried_"):
            return lambda *a: self.call(name, *a)
        raise AttributeError(name)


lib = _Lib()
