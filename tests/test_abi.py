"""The C-ABI library builds for gfx950, loads, and exports every symbol include/vrc_hip.h
declares.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vrc_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    from libre_amd import vrc
    L = vrc.load_library()
    declared = _declared("vrc_hip.h")
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), "libvrc_hip.so does not export %s" % name
    assert sorted(vrc.EXPORTS) == declared
    assert L.vrc_abi_version() == vrc.ABI_VERSION == 4  # the product build; a -DVRC_DEV_BUILD library reports -4
    assert L.vrc_is_dev_build() == 0
    assert L.vrc_last_kernel() == b""


def test_comm_of_one_rank_needs_no_rccl_and_no_gpu_for_argument_checks(built):
    # vrc_comm_* / vrc_gather_tiles (sort-first exchange, livre/eq/Channel.cpp:519-523): argument errors come
    # back as codes before any device or RCCL call
    from libre_amd import vrc
    L = vrc.load_library()
    comm = C.c_void_p()
    assert L.vrc_comm_create(None, 0, 1, None, C.byref(comm)) == vrc.VRC_EINVAL
    assert b"bad ctx" in L.vrc_last_error()
    assert L.vrc_comm_info(None, None, None) == vrc.VRC_EINVAL
    assert L.vrc_gather_tiles(None, None, None, 0, 16, 16, 1, None, 0, None, 0, 0, None) == vrc.VRC_EINVAL
    assert C.sizeof(vrc.Band) == 12
    L.vrc_comm_destroy(None)


def test_library_is_gfx950_code_object(built):
    from libre_amd import vrc
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + vrc.LIB_PATH], capture_output=True, text=True)
    blob = open(vrc.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"vrc_k_raycast" in blob
    del out


def test_pod_layouts_match_reference_shapes(built):
    from libre_amd import vrc
    # cuda/Renderer.cuh:35-66: NodeData = 12 floats; RenderData = 4 uint + 2 float
    assert C.sizeof(vrc.NodeData) == 48
    assert C.sizeof(vrc.RenderData) == 24
    assert C.sizeof(vrc.ViewData) == (3 + 4 + 16 * 3 + 3 + 3 + 1) * 4


def test_missing_library_fails_loudly(tmp_path):
    from libre_amd import vrc
    with pytest.raises(FileNotFoundError):
        vrc.load_library(str(tmp_path / "nope.so"))


def test_no_oracle_in_product_tree():
    # the product never references the oracle or the CPU harness
    for d in ("libre_amd", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(root, f), errors="replace").read()
                    assert "livre_oracle" not in txt and "orc_raycast" not in txt, os.path.join(root, f)
                    assert "libharness" not in txt, os.path.join(root, f)


def test_developer_switches_need_the_dev_build_guard(tmp_path):
    # vrc_core.h: a stray -DVRC_ABLATE_NO_FETCH (wrong pixels by design) or any other kernel switch without
    # -DVRC_DEV_BUILD must not compile; with the guard it does (and such a library reports vrc_abi_version() < 0)
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include "%s/libre_amd/csrc/vrc_core.h"\nint main() { return 0; }\n'
                   % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wno-unknown-pragmas", str(src)]
    assert subprocess.run(base, capture_output=True).returncode == 0
    for sw in ("-DVRC_ABLATE_NO_FETCH", "-DVRC_ZRUN", "-DVRC_PIPELINE", "-DVRC_LAYOUT=5", "-DVRC_LANES_ROWMAJOR",
               "-DVRC_GROUP=4", "-DVRC_LDS_STATS", "-DVRC_TEST_BIAS_ENTRY"):
        r = subprocess.run(base + [sw], capture_output=True, text=True)
        assert r.returncode != 0 and "VRC_DEV_BUILD" in r.stderr, sw
        assert subprocess.run(base + [sw, "-DVRC_DEV_BUILD"], capture_output=True).returncode == 0, sw
