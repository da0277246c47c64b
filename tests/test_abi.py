"""The C-ABI library loads and exports every symbol include/ried_gan.h declares (no GPU needed)."""
import ctypes
import os

from reid_gan_amd import _lib


def test_header_parses_and_names_entry_points():
    protos = _lib.parse_header()
    for name in ("ried_conv2d", "ried_conv2d_wgrad", "ried_bgemm", "ried_bn_apply", "ried_bn_backward",
                 "ried_ssim_fwd", "ried_ssim_bwd", "ried_adam", "ried_sn_sigma", "ried_last_error"):
        assert name in protos, name
    assert len(protos) >= 35
    ret, args = protos["ried_conv2d"]
    assert ret == "int" and len(args) == 31 and args[-1][0] == "ried_stream_t"


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build the library first: ./build.sh"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(dll, name), f"{name} declared in include/ried_gan.h but not exported"
    dll.ried_abi_version.restype = ctypes.c_int
    assert dll.ried_abi_version() == 1


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected before any launch, with a message."""
    _lib.lib.load()
    try:
        _lib.lib.ried_conv2d(None, 8, None, None, 8, 1, 4, 4, 8, 4, 4, 8, 3, 3, 1, 1, 1, 1, 0, None, 0, None, 0, 0, None, 0, None, 0, 0.0, 0, None)
    except RuntimeError as e:
        assert "null pointer" in str(e)
    else:
        raise AssertionError("expected RuntimeError")
