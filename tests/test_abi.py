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
    assert lib.polaris_hip_abi_version() == 5


def test_struct_sizes_match_header(built):
    from polaris_amd import ctypes_api as T

    assert C.sizeof(T.BlockRequest) == 48          # 12 x 4 bytes, tracer/tracer.go:6-34
    assert C.sizeof(T.TraceStats) == 7 * 8 + 2 * 32 * 8 + 8
    assert T.BVH_NODE.itemsize == 32 and T.MESH_INSTANCE.itemsize == 80
    assert T.MATERIAL_NODE.itemsize == 64 and T.EMISSIVE.itemsize == 80 and T.TEXTURE_META.itemsize == 16
    assert C.sizeof(T.IpcExport) == 576             # 8 x 4 + 4 x 64 (hipIpcMemHandle_t per slot) + 4 x 64 (hipIpcEventHandle_t per slot) + 32 (PCI bus id, ABI 5)
    assert C.sizeof(T.DeviceIdentity) == 168 and C.sizeof(T.PeerInfo) == 72 and C.sizeof(T.BvhBuildInput) == 80
    assert T.BvhBuildInput().struct_size == 80 and T.DeviceIdentity().struct_size == 168     # (set by the mirrors' constructors: the library refuses another size)


def test_header_struct_sizes_as_the_c_compiler_sees_them(tmp_path):
    """The ctypes mirrors above against sizeof() of include/polaris_hip.h itself, compiled as C (what cgo does with it)."""
    import subprocess

    from polaris_amd import ctypes_api as T

    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "polaris_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %d\\n", sizeof(PolarisIpcExport), '
                   'sizeof(PolarisDeviceIdentity), sizeof(PolarisPeerInfo), sizeof(PolarisBvhBuildInput), sizeof(PolarisBlockRequest), sizeof(PolarisTraceStats), '
                   'POLARIS_MERGE_BRANCHES); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    assert got == [C.sizeof(T.IpcExport), C.sizeof(T.DeviceIdentity), C.sizeof(T.PeerInfo), C.sizeof(T.BvhBuildInput), C.sizeof(T.BlockRequest),
                   C.sizeof(T.TraceStats), len(T.MERGE_BRANCHES)]


def test_experiment_patches_still_apply_to_the_product_sources():
    """The product translation units carry no experiment code: the loop / prologue instrumentation of
    k_trace and the coherence-reorder experiment live as patches under polaris_amd/csrc/experiments/, applied to a COPY by
    scripts/build_variant.sh.  A patch that no longer applies is a rotten experiment: caught here, not on the GPU box."""
    import glob
    import shutil
    import subprocess
    import tempfile

    csrc = os.path.join(ROOT, "polaris_amd", "csrc")
    patches = sorted(glob.glob(os.path.join(csrc, "experiments", "*.patch")))
    assert [os.path.basename(p) for p in patches] == ["profile_loops.patch", "profile_prologue.patch", "reorder.patch", "timing_inexact.patch", "tiny_lds_transposed.patch", "trace_spill.patch"]
    for f in ("kernels.h", "polaris_hip.hip"):
        text = open(os.path.join(csrc, f)).read()
        assert "POLARIS_EXP_REORDER" not in text and "POLARIS_PROFILE_" not in text and "hipcub" not in text, f
    with tempfile.TemporaryDirectory() as tmp:
        for f in glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")):
            shutil.copy(f, tmp)
        for name in ("reorder.patch", "profile_loops.patch", "profile_prologue.patch"):   # in this order they also combine (scripts/wave_lines.py)
            r = subprocess.run(["patch", "-s", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(csrc, "experiments", name)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 0, (name, r.stdout, r.stderr)
        text = open(os.path.join(tmp, "kernels.h")).read()
        assert "POLARIS_EXP_REORDER" in text and "POLARIS_PROFILE_LOOPS" in text and "POLARIS_PROFILE_PROLOGUE" in text
    for name in ("timing_inexact.patch", "trace_spill.patch", "tiny_lds_transposed.patch"):       # each of these applies to the product sources on its own
        with tempfile.TemporaryDirectory() as tmp:
            for f in glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")):
                shutil.copy(f, tmp)
            r = subprocess.run(["patch", "-s", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(csrc, "experiments", name)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 0, (name, r.stdout, r.stderr)


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
    # ABI 5: identity / peer / merge-branch queries
    ident = T.DeviceIdentity()
    assert lib.polaris_hip_device_identity(0, C.byref(ident)) == 3 and lib.polaris_hip_device_identity(0, None) == 2
    ident.struct_size = 8
    assert lib.polaris_hip_device_identity(0, C.byref(ident)) == 2 and b"struct_size" in lib.polaris_hip_last_error(None)
    can = C.c_int(7)
    assert lib.polaris_hip_can_access_peer(0, 1, C.byref(can)) == 3 and can.value == 0
    assert lib.polaris_hip_peer_info(None, None) == 2 and lib.polaris_hip_merge_counts(None, None) == 2
    # a build input of another layout is refused before anything is read from it
    bad = T.BvhBuildInput()
    bad.struct_size = 72
    n = C.c_uint32()
    assert lib.polaris_hip_build_bvh(0, C.byref(bad), C.c_void_p(1), 8, C.byref(n), C.c_void_p(1), C.c_void_p(1), None) == 2
    assert b"struct_size" in lib.polaris_hip_build_bvh_error()
    lib.polaris_hip_destroy(None)


def test_missing_library_fails_loudly(tmp_path):
    from polaris_amd import ctypes_api as T

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        T.load_library(str(tmp_path / "libpolaris_hip.so"))
