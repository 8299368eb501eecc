"""CPU-only checks of the drop-in boundary: the library builds for gfx950, loads, exports every
symbol include/summersph.h declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from summersph_amd import capi


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(capi.LIB_PATH):
        ge.build()
    return capi.load()


def test_header_symbols_match_binding_list():
    hdr = open(os.path.join(ROOT, "include", "summersph.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sph_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)


def test_library_exports_every_symbol(lib):
    for s in capi.SYMBOLS:
        assert hasattr(lib, s), s


def test_abi_version_and_defaults(lib):
    assert lib.sph_abi_version() == 1
    p = capi.default_params()
    # the reference's constants, REAL(4)-rounded literals included (SUMMER_SPH.f90:7,11,317,373,855,857)
    assert (p.h, p.gamma, p.gamma_m1, p.nq) == (2.5, 1.4, 0.4, 5000)
    assert p.G == 39.478416442871094
    assert p.visc_eps == 0.009999999776482582
    assert p.alpha_decay == 0.15000000596046448
    assert (p.dt_max, p.dt_min) == (0.10000000149011612, 9.999999747378752e-05)
    assert p.kernel_pi == 3.14159265359 and p.bounding_size == 1500.0


def test_struct_sizes_match_header(lib):
    # sph_params: 3 doubles, 2 int32, 9 + 5 + 1 doubles
    assert ctypes.sizeof(capi.Params) == 3 * 8 + 2 * 4 + 15 * 8
    # sph_stats: 2 int64, 3+1+1+1 int32, double, 4 int64, int64, double, 2 int32, double, int64
    assert ctypes.sizeof(capi.Stats) == 16 + 24 + 8 + 32 + 8 + 8 + 8 + 8 + 8


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SphError) as e:
        capi.Context()
    assert e.value.status == 2   # SPH_ERR_NO_DEVICE


def test_product_does_not_touch_the_oracle():
    """the shipped path (package sources + Fortran host) must never reference oracle/"""
    pkg = os.path.join(ROOT, "summersph_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".f90", ".h")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "liborc" not in txt and "sph_oracle" not in txt and "from oracle" not in txt \
                    and "import oracle" not in txt, os.path.join(dp, fn)
