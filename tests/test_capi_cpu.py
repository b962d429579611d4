"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads,
exports every symbol include/kmersets_hip.h declares, and refuses to compute
without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C

import pytest

from kmersets import capi


@pytest.fixture(scope="module")
def lib():
    capi.build()
    return capi.lib()


def test_header_symbols_exported(lib):
    names = capi.exported_symbols()
    assert len(names) >= 18
    raw = C.CDLL(capi.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), "%s is declared in kmersets_hip.h but not exported" % name


def test_version_and_status_codes(lib):
    assert lib.ksh_version() >= 1
    # absl codes the reference uses: kInternal = 13, kFailedPrecondition = 9
    assert (capi.KSH_INTERNAL, capi.KSH_FAILED_PRECONDITION, capi.KSH_INVALID_ARGUMENT) == (13, 9, 3)


def test_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = lib.ksh_ctx_create(0, None, C.byref(h))
    assert rc == capi.KSH_INTERNAL and not h.value
    assert b"device" in lib.ksh_last_error()
    with pytest.raises(RuntimeError):
        capi.Context(0)


def test_geometry_validation_is_host_side(lib):
    g = capi.geom(23, 14)
    assert (g.k, g.n_bucket_bits, g.key_bytes) == (23, 14, 4)
    assert capi.geom(31, 14).key_bytes == 8
    assert capi.geom(15, 14).key_bytes == 2   # the reference's (15, 14, uint16_t)
    assert capi.geom(9, 10, 1).key_bytes == 2 and capi.geom(19, 10).key_bytes == 4
