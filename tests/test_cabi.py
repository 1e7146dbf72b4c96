"""The C-ABI library: loads without a GPU, exports every symbol include/ftx.h declares, and its
argument checks answer before anything is launched (no compute calls here)."""
import ctypes
import os
import re

from fusiontransformer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ftx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ftx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(ftx_lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(ftx_lib, s), s
    assert sorted(_lib.SIGNATURES) == syms, "fusiontransformer_amd/_lib.py SIGNATURES must mirror include/ftx.h"


def test_version_and_capacity(ftx_lib):
    assert ftx_lib.ftx_version() >= 100
    for n in (0, 1, 31, 32, 33, 1000, 81237):
        cap = ftx_lib.ftx_hashtable_capacity(n)
        assert cap >= 64 and cap >= 2 * n and cap & (cap - 1) == 0


def test_argument_errors_are_reported_not_thrown(ftx_lib):
    assert ftx_lib.ftx_hash(None, -1, None, None) == -1
    assert b"n < 0" in ftx_lib.ftx_last_error()
    assert ftx_lib.ftx_hash(None, 0, None, None) == 0          # empty input is a no-op, not an error
    assert ftx_lib.ftx_spconv_pairs_gemm(None, 0, None, None, 0, None, 8, 6, 8, 27, None, None) == -1   # channels not multiple of 4
    assert b"multiples of 4" in ftx_lib.ftx_last_error()
    assert ftx_lib.ftx_hashtable_build(None, 0, None, None, 100, None) == -1               # capacity not a power of two
    assert ftx_lib.ftx_lift_gather_fwd(None, None, None, 4, 1, 24, 24, 95, 370, 1226, None, None) == -1


def test_header_has_no_torch_types():
    text = open(os.path.join(ROOT, "include", "ftx.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)   # declarations only
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code


def test_stream_scratch_entry_points_answer_without_a_gpu(ftx_lib):
    n = ftx_lib.ftx_stream_scratch_bytes()
    assert 4096 < n < (4 << 20)
    assert ftx_lib.ftx_stream_scratch_attach(None, None, n) == -1                 # null buffer
    assert b"256-byte aligned" in ftx_lib.ftx_last_error()
    assert ftx_lib.ftx_stream_scratch_attach(None, ctypes.c_void_p(4096), 16) == -1   # too small: refused before anything is touched
    assert b"ftx_stream_scratch_bytes" in ftx_lib.ftx_last_error()
    assert ftx_lib.ftx_stream_scratch_reset(None) == 0 and ftx_lib.ftx_stream_scratch_release(None) == 0   # nothing attached: no-ops


def test_no_process_wide_switches_are_exported(ftx_lib):
    """Round 2 selected kernel variants through process-wide setters; they are gone (tilings are per-call arguments)."""
    for name in ("ftx_spconv_set_gemm_variant", "ftx_spconv_set_split", "ftx_attn_set_config"):
        assert not hasattr(ftx_lib, name), name


def test_wgrad_workspace_is_a_function_of_the_arguments(ftx_lib):
    """Size query and launch must agree in every environment: the tile length comes from a table, not from an occupancy query."""
    a = ftx_lib.ftx_spconv_pairs_wgrad_workspace_bytes(382735, 128, 96, 27)
    assert a == ftx_lib.ftx_spconv_pairs_wgrad_workspace_bytes(382735, 128, 96, 27) and a > 0
    assert ftx_lib.ftx_spconv_wgrad_resident_blocks(32, 32) == 8 and ftx_lib.ftx_spconv_wgrad_resident_blocks(128, 96) == 2
    assert ftx_lib.ftx_spconv_wgrad_table_blocks(4, 1, 1, 1) == -1


def test_wgrad_occupancy_table_matches_the_code_object():
    """The table in csrc/ftx_spconv.hip against the registers / LDS the compiler really allocated (llvm-readelf notes of the gfx950
    code object): min(8, 512 / VGPRs rounded up to 8, 160 KiB / LDS).  A stale table costs speed only, never results."""
    import shutil
    import subprocess
    import tempfile
    bundler, readelf = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    obj = os.path.join(ROOT, "fusiontransformer_amd", "csrc", "ftx_spconv.o")
    if not (os.path.exists(bundler) and os.path.exists(readelf) and os.path.exists(obj)):
        import pytest
        pytest.skip("ROCm LLVM tools or the built object are not here")
    lib = _lib.load()
    with tempfile.TemporaryDirectory() as d:
        dev = os.path.join(d, "dev.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
                        os.path.join(ROOT, "fusiontransformer_amd", "csrc", "ftx_spconv.hip"), "-o", dev], check=True, cwd=d)
        co = os.path.join(d, "k.co")
        subprocess.run([bundler, "--unbundle", "--type=o", "--input=" + dev, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
    seen = 0
    for blk in notes.split("- .agpr_count")[1:]:
        m = re.search(r"pairs_wgrad_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)E", blk)
        if not m:
            continue
        mi, ni, wmg, wng = (int(x) for x in m.groups())
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        occ = min(8, 512 // ((vg + 7) // 8 * 8), (160 * 1024) // lds)
        assert lib.ftx_spconv_wgrad_table_blocks(mi, wmg, ni, wng) == occ, ((mi, ni, wmg, wng), vg, lds, occ)
        seen += 1
    assert seen == 16
