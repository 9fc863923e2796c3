"""CPU test: libyakhip.so loads and exports every entry point include/yakhip.h declares.
(No compute call is made: there is no GPU here.)"""
import ctypes
import os
import re

import pytest

from yak_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "yakhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yakhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for must in ("yakhip_integrate", "yakhip_clear_volume", "yakhip_compute_dists", "yakhip_ctx_create"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "libyakhip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, "declared in yakhip.h but not exported: %s" % missing


def test_python_binding_covers_every_declared_symbol():
    assert sorted(capi.EXPORTS) == declared_symbols()
    capi.load()


def test_volume_struct_layout_matches_header():
    # struct yakhip_volume: void*, int[3], int, int, float[3], float, int, void*
    assert ctypes.sizeof(capi.VolumeDesc) == 8 + 12 + 8 + 12 + 4 + 4 + 8
    assert capi.VolumeDesc.data.offset == 0 and capi.VolumeDesc.dims.offset == 8
    assert capi.VolumeDesc.z0.offset == 20 and capi.VolumeDesc.voxel_size.offset == 28
    assert capi.VolumeDesc.max_weight.offset == 44 and capi.VolumeDesc.surface_map.offset == 48


def test_no_gpu_is_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.YakHipError) as e:
        capi.Context(0, use_torch_stream=False)
    assert e.value.code == -3  # YAKHIP_ENOGPU


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "yak_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "liboracle" not in src and "yak_oracle" not in src, f
