"""The C-ABI library loads and exports every symbol include/polaris_hip.h declares (no GPU needed);
entry points that cannot work without a device fail with a status code, never a crash."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(built):
    from polaris_amd import ctypes_api as T

    lib = T.load_library()
    header = open(os.path.join(ROOT, "include", "polaris_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(polaris_hip_[a-z_]+)\s*\(", header)))
    assert declared, "no declarations found in polaris_hip.h"
    for name in declared:
        assert hasattr(lib, name), f"libpolaris_hip.so does not export {name}"
    assert sorted(T.C_ABI_SYMBOLS) == declared, "ctypes_api.C_ABI_SYMBOLS is out of sync with polaris_hip.h"
    assert lib.polaris_hip_abi_version() == 4


def test_struct_sizes_match_header(built):
    from polaris_amd import ctypes_api as T

    assert C.sizeof(T.BlockRequest) == 48          # 12 x 4 bytes, tracer/tracer.go:6-34
    assert C.sizeof(T.TraceStats) == 7 * 8 + 2 * 32 * 8 + 8
    assert T.BVH_NODE.itemsize == 32 and T.MESH_INSTANCE.itemsize == 80
    assert T.MATERIAL_NODE.itemsize == 64 and T.EMISSIVE.itemsize == 80 and T.TEXTURE_META.itemsize == 16
    assert C.sizeof(T.IpcExport) == 544             # 8 x 4 + 4 x 64 (hipIpcMemHandle_t per slot) + 4 x 64 (hipIpcEventHandle_t per slot)


def test_no_device_is_an_error_not_a_crash(built):
    import torch

    from polaris_amd import ctypes_api as T

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = T.load_library()
    assert lib.polaris_hip_device_count() == 0
    h = C.c_void_p()
    rc = lib.polaris_hip_create(0, C.byref(h))
    assert rc == 3 and not h            # POLARIS_E_NO_DEVICE
    assert b"out of range" in lib.polaris_hip_last_error(None)
    # null handles are rejected, not dereferenced
    assert lib.polaris_hip_resize(None, 4, 4) == 2
    assert lib.polaris_hip_trace(None, None, None, 0, None) == 2
    assert lib.polaris_hip_ipc_export(None, 3, None) == 2 and lib.polaris_hip_merge_ipc(None, None, 0, None) == 2
    assert lib.polaris_hip_ipc_open(None, None, None) == 2 and lib.polaris_hip_ipc_close(None, None) == 2
    lib.polaris_hip_destroy(None)


def test_missing_library_fails_loudly(tmp_path):
    from polaris_amd import ctypes_api as T

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        T.load_library(str(tmp_path / "libpolaris_hip.so"))
