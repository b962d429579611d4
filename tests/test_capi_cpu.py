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


def test_custom_transport_struct_is_size_tagged(lib):
    """ksh_comm_fns starts with its own size (ksh_version() >= 2): a struct that does not say how large it is, is
    refused on the host, before any device is touched (a caller compiled against an older, shorter struct would
    otherwise be read past its end)."""
    assert lib.ksh_version() >= 2
    assert capi.CommFns._fields_[0][0] == "struct_size"
    fns = capi.CommFns()          # zero-initialised: struct_size == 0
    h = C.c_void_p()
    # (no context without a GPU: the argument check comes first either way)
    rc = lib.ksh_comm_create_custom(None, 0, 2, C.byref(fns), C.byref(h))
    assert rc == capi.KSH_INVALID_ARGUMENT and not h.value


def test_repeat_rich_generator_twins_agree():
    """synth.plant_repeats (numpy) and synth_torch.plant_repeats plant the same bases, so the repeat-rich family of
    bench_loop.py --repeats is the one the parity tests check against the oracle at small sizes."""
    import numpy as np
    import torch

    from kmersets import synth, synth_torch

    g = synth.random_genome(150_000, 9)
    a = synth.plant_repeats(g, 200, 5, 13)
    b = synth_torch.plant_repeats(torch.from_numpy(g.astype(np.int64)), 200, 5, 13).numpy().astype(np.uint8)
    assert (a != g).sum() > 10_000 and np.array_equal(a, b)
    s1 = synth.phylogeny_sets(23, 2, 100_000, 3, repeats=(150, 6))
    s2 = synth_torch.phylogeny_sets(23, 2, 100_000, 3, torch.device("cpu"), repeats=(150, 6))
    assert all(np.array_equal(x, y.numpy().astype(np.uint64)) for x, y in zip(s1, s2))
    # the repeats make the set smaller than the genome: shared k-mers collapse
    assert len(s1[0]) < len(synth.phylogeny_sets(23, 2, 100_000, 3)[0])
