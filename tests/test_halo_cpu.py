"""CPU-only checks of the native multi-GPU layer's boundary: libsummersph_halo.so builds, loads (RCCL and all) and exports
every symbol include/summersph_halo.h declares; nothing in it can run without a GPU."""
import os
import re

import pytest

from conftest import ROOT
from summersph_amd import capi, halo


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(halo.LIB_PATH) or not os.path.exists(capi.LIB_PATH):
        ge.build()
    return halo.load()


def test_header_symbols_match_binding_list():
    hdr = open(os.path.join(ROOT, "include", "summersph_halo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sph_halo_[a-z_]+)\s*\(", hdr))
    assert declared == set(halo.SYMBOLS), declared ^ set(halo.SYMBOLS)


def test_library_exports_every_symbol(lib):
    for s in halo.SYMBOLS:
        assert hasattr(lib, s), s


def test_halo_library_is_a_client_of_the_c_abi():
    """the orchestration layer calls the public entry points only: no internal header, no oracle"""
    src = open(os.path.join(ROOT, "summersph_amd", "csrc", "halo.hip")).read()
    assert "sph_internal.hpp" not in src and "oracle" not in src
    assert "ncclSend" in src and "ncclRecv" in src and "ncclAllGather" in src


def test_fortran_binding_declares_the_halo_entry_points():
    f90 = open(os.path.join(ROOT, "summersph_amd", "host", "sph_hip_halo_binding.f90")).read()
    for s in halo.SYMBOLS:
        if s in ("sph_halo_hub_create", "sph_halo_hub_destroy", "sph_halo_create_inproc", "sph_halo_attach"):
            continue            # test transport / hosts that already own a communicator: C callers
        assert f"name='{s}'" in f90, s


def test_hub_argument_checks(lib):
    assert lib.sph_halo_hub_create(0) is None and lib.sph_halo_hub_create(65) is None
    hub = lib.sph_halo_hub_create(2)
    assert hub
    lib.sph_halo_hub_destroy(hub)
