"""CPU: the C-ABI library loads and exports every symbol include/neuralcx.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "neuralcx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ncx_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    from neuralcx import _lib
    names = _declared()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_host_only_entry_points():
    from neuralcx import _lib
    from neuralcx._lib import NcxDims
    assert "gfx950" in _lib.version()
    d = NcxDims(B=512, K=24, dv=2048, dq=2400, dz=360, da=2400, A=2000, H=256, L=1, n_img=82783, flags=15)
    assert _lib.lib().ncx_input_size(ctypes.byref(d)) == 14089          # cx.py:245-251
    nbytes = _lib.lib().ncx_workspace_bytes(ctypes.byref(d))
    assert 16 << 20 < nbytes < 1 << 30
    bad = NcxDims(B=0, K=24, dv=1, dq=1, dz=1, da=1, A=1, H=1, L=1, n_img=1)
    assert _lib.lib().ncx_workspace_bytes(ctypes.byref(bad)) == 0
    # argument validation happens before any launch: NULL pointers are rejected without a GPU
    assert _lib.lib().ncx_loss_rank(None, None, 4, 24, 0.0, None, None, None, None, None, None) == -1
    assert _lib.lib().ncx_adam_step(None, None, None, None, 4, 1e-4, 0.9, 0.999, 1e-8, 1, 1.0, None) == -1
    # the RCCL handle validates its arguments before it loads RCCL
    assert _lib.lib().ncx_allreduce(None, None, 4, None) == -1 and _lib.lib().ncx_comm_unique_id(None) == -1
    comm = ctypes.c_void_p()
    assert _lib.lib().ncx_comm_create(ctypes.create_string_buffer(128), 2, 2, ctypes.byref(comm)) == -2     # rank >= nranks


def test_struct_layout_matches_header():
    from neuralcx._lib import NcxDims, NcxInputs, NcxParams, NcxGrads
    assert ctypes.sizeof(NcxDims) == 64
    assert ctypes.sizeof(NcxInputs) == 80 and ctypes.sizeof(NcxParams) == 72 == ctypes.sizeof(NcxGrads)


def test_missing_library_fails_loudly(monkeypatch):
    from neuralcx import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libneuralcx_hip.so")
    with pytest.raises(_lib.NcxError):
        _lib.lib()


def test_workgroup_maps_are_bijections():
    """WgMap (csrc/ncx_gemm.h): every (tile, k-chunk) gets exactly one workgroup id, in the interleaved layout and in
    the chunk-per-XCD layout (S | 8 with padding ids, 8 | S), for ragged tile counts on either side."""
    from neuralcx import _lib
    L = _lib.lib()
    for tm in (1, 2, 3, 4, 7, 32):
        for tn in (1, 2, 5, 19, 103):
            for S in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24):
                assert L.ncx_wgmap_check(tm, tn, S) == 0, (tm, tn, S)
    assert L.ncx_wgmap_check(0, 1, 1) < 0
