"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, exports every
symbol the headers under include/ declare, and its host-only entry points behave (no compute
calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from plenoctree_amd import _lib, build
from oracle import nerf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def _header_symbols():
    inc = os.path.join(ROOT, "include")
    src = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pxo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound_and_exported(lib):
    syms = _header_symbols()
    assert len(syms) >= 24
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True)
    exported = set(re.findall(r"\bT (pxo_[a-z0-9_]+)", nm.stdout))
    assert set(syms) <= exported, sorted(set(syms) - exported)


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["strings", "-a", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "mlp_fwd_kernel" in out and "wgrad_kernel" in out


def test_param_layout_matches_reference_shapes(lib):
    for deg in range(5):
        cfg = _lib.make_cfg(sh_deg=deg)
        leaves = (_lib.PxoLeaf * _lib.NUM_LEAVES)()
        n = ctypes.c_int64(0)
        assert lib.pxo_param_layout(ctypes.byref(cfg), leaves, ctypes.byref(n)) == 0
        shapes = O.layer_shapes(O.Cfg(sh_deg=deg))
        off = 0
        for l, (fi, fo) in enumerate(shapes):
            k, b = leaves[2 * l], leaves[2 * l + 1]
            assert (k.layer, k.is_bias, k.offset, k.rows, k.cols) == (l, 0, off, fi, fo)
            off += fi * fo
            assert (b.layer, b.is_bias, b.offset, b.rows) == (l, 1, off, fo)
            off += fo
        assert n.value == off
    # 505,649 (SH16) / 512,588 (SH25) parameters per MLP
    cfg = _lib.make_cfg(sh_deg=3)
    assert lib.pxo_param_layout(ctypes.byref(cfg), None, ctypes.byref(n)) == 0 and n.value == 505649
    cfg = _lib.make_cfg(sh_deg=4)
    assert lib.pxo_param_layout(ctypes.byref(cfg), None, ctypes.byref(n)) == 0 and n.value == 512588


def test_error_reporting(lib):
    cfg = _lib.make_cfg(sh_deg=7)
    n = ctypes.c_int64(0)
    assert lib.pxo_packed_sizes(ctypes.byref(cfg), ctypes.byref(n), None) == -1
    assert b"sh_deg" in lib.pxo_last_error()
    cfg = _lib.make_cfg(max_deg_point=4)
    assert lib.pxo_packed_sizes(ctypes.byref(cfg), ctypes.byref(n), None) == -1
    cfg = _lib.make_cfg(num_coarse_samples=200, num_fine_samples=128)
    assert lib.pxo_packed_sizes(ctypes.byref(cfg), ctypes.byref(n), None) == -1
    with pytest.raises(_lib.PxoError):
        _lib.check(-1, "unit")
    with pytest.raises(ValueError):
        _lib.make_cfg(nonsense=1)


def test_size_queries(lib):
    cfg = _lib.make_cfg()
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.pxo_packed_sizes(ctypes.byref(cfg), ctypes.byref(a), ctypes.byref(b)) == 0
    # 8 trunk layers (64 + 4*256 + 320 + 2*256 rows) x 256 + head 256x64 + biases 8*256+64
    assert a.value == (64 + 4 * 256 + 320 + 2 * 256) * 256 + 256 * 64 + 8 * 256 + 64
    assert b.value == 64 * 256 + 7 * 256 * 256
    # 1 bit per activation, whole 128-row tile slots, + 512 spare slots for the half-height tail tiles of a ragged
    # last round (pxo_common.h TileSched)
    tm = lib.pxo_tile_rows()
    assert tm == 128
    slot = 8 * 128 * 256 // 8
    assert lib.pxo_relu_mask_bytes(128) == (1 + 512) * slot
    assert lib.pxo_relu_mask_bytes(129) == (2 + 512) * slot
    assert lib.pxo_dbias_partial_bytes(128) == (1 + 512) * (9 * 256 + 1) * 4   # one [9][256] partial + a live byte per tile slot
    nbytes = ctypes.c_size_t(0)
    assert lib.pxo_train_workspace_bytes(ctypes.byref(cfg), 4096, ctypes.byref(nbytes)) == 0
    # dominated by saved activations + dz: 2 * 8 layers * (4096*256 + 10000) rows * 1 KiB
    rows = 4096 * 256 + 10000
    assert 2 * 8 * rows * 1024 < nbytes.value < 2.5 * 8 * rows * 1024
    assert lib.pxo_render_workspace_bytes(ctypes.byref(cfg), 4096, ctypes.byref(nbytes)) == 0
    assert nbytes.value < 1 << 30


def test_no_cpu_fallback_without_gpu():
    import torch
    from plenoctree_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PxoError):
        ops.posenc(torch.zeros(4, 3))
