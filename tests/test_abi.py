"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/wnoise.h declares; compute entry points refuse to run without a GPU (no CPU fallback)."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT

PKG = "wavelet-noise-in-ray-tracing_amd"


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, PKG, "libwnoise_hip.so")):
        ge.build()
    return importlib.import_module(PKG + "._capi")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wnoise.h")).read()
    return sorted(set(re.findall(r"WN_API\s+[\w\s\*]+?\b(wn_\w+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(capi):
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/wnoise.h but not exported"
        assert n in capi.SIGNATURES, f"{n} has no ctypes signature in _capi.py"
    assert set(capi.SIGNATURES) == set(names)


def test_grid_struct_layout_matches_header(capi):
    assert C.sizeof(capi.wn_grid) == 12 * 4


def test_host_side_setup_streams_match_oracle(capi, ora, gold):
    """wn_gaussian_fill / wn_perlin_permutation are host helpers (libstdc++ <random>): they run
    without a GPU and must give the reference's streams."""
    lib = capi.load()
    g = np.zeros(4096, np.float32)
    assert lib.wn_gaussian_fill(12345, g.size, g.ctypes.data_as(C.c_void_p)) == 0
    assert (g.view(np.uint32) == gold["gauss_12345"].view(np.uint32)).all()
    for s, table in zip(gold["perm_seeds"], gold["perm_tables"]):
        p = np.zeros(512, np.int32)
        assert lib.wn_perlin_permutation(int(s), p.ctypes.data_as(C.c_void_p)) == 0
        assert (p == table).all()
    assert lib.wn_tile_even_size(7) == 8 and lib.wn_tile_even_size(128) == 128


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback(capi):
    lib = capi.load()
    h = C.c_void_p()
    rc = lib.wn_tile_generate(16, 3, 1, C.byref(h))
    assert rc == capi.WN_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.wn_last_error()
    rc = lib.wn_perm_create_seeded(1, C.byref(h))
    assert rc == capi.WN_ERR_NO_DEVICE
    g = capi.wn_grid(8, 8, 8, 0, 8, 4.0, 16.0, 2.0, 0, 0.0, 1.0, 0)
    assert lib.wn_eval3d_grid(None, C.byref(g), None, None) == capi.WN_ERR_NO_DEVICE
    assert lib.wn_eval3d_points(None, None, 0, None, None) == capi.WN_ERR_NO_DEVICE


def test_package_import_fails_loudly_without_library(tmp_path, capi):
    """Importing the product without libwnoise_hip.so must raise, not fall back."""
    import subprocess
    import sys
    code = (
        "import importlib, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        f"m = importlib.import_module({PKG + '._capi'!r})\n"
        f"m.LIB_PATH = {str(tmp_path / 'missing.so')!r}\n"
        "m._lib = None\n"
        "try:\n    m.load()\nexcept ImportError as e:\n    print('RAISED', 'no CPU implementation' in str(e))\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


# ---- libwnoise_shard.so (include/wnoise_shard.h): the gather of the sharded dense-grid path over RCCL ----------------------
def shard_declared_symbols():
    text = open(os.path.join(ROOT, "include", "wnoise_shard.h")).read()
    return sorted(set(re.findall(r"WN_SHARD_API\s+[\w\s\*]+?\b(wn_\w+)\s*\(", text)))


def test_shard_header_symbols_all_exported_and_bound_and_bounds_match_python():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, PKG, "libwnoise_shard.so")):
        ge.build()
    cs = importlib.import_module(PKG + "._capi_shard")
    lib = cs.load()  # links librccl: present in the image, needs no GPU to load
    names = shard_declared_symbols()
    assert len(names) == 7, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/wnoise_shard.h but not exported"
    assert set(cs.SIGNATURES) == set(names)
    wn = importlib.import_module(PKG)
    for nz, world in ((2048, 8), (512, 3), (7, 4), (0, 2), (5, 8)):
        seen = 0
        for r in range(world):
            z0, z1 = C.c_int(-1), C.c_int(-1)
            assert lib.wn_shard_bounds(nz, world, r, C.byref(z0), C.byref(z1)) == 0
            assert (z0.value, z1.value) == wn.slab_bounds(nz, world, r)  # the C ABI and shard.py cut the same slabs
            assert z0.value == seen
            seen = z1.value
        assert seen == nz
    z0, z1 = C.c_int(), C.c_int()
    assert lib.wn_shard_bounds(8, 2, 2, C.byref(z0), C.byref(z1)) != 0 and b"rank=2" in lib.wn_shard_last_error()
    assert lib.wn_gather_volume(None, None, 8, 8, 8, 0, None, 0, None) != 0  # NULL communicator is refused, not dereferenced
