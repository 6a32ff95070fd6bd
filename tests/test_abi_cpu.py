"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly the
symbols `include/dfdclip.h` declares; the ctypes table mirrors the header; the host side fails
loudly (no silent CPU fallback) when handed CPU tensors."""
import os
import re

import pytest
import torch

from dfd_clip_amd import capi
from dfd_clip_amd.build import LIB_PATH, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    return capi.load_library()


def header_functions():
    text = open(os.path.join(ROOT, "include", "dfdclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfd_[a-z0-9_]+)\s*\(", text)))


def test_header_and_ctypes_table_agree():
    assert header_functions() == sorted(capi.SIGNATURES)


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.dfd_abi_version() == capi.ABI_VERSION
    assert os.path.exists(LIB_PATH)


def test_invalid_arguments_are_reported_not_launched(lib):
    # null pointers / bad shapes are rejected on the host before any launch (no GPU needed)
    rc = lib.dfd_layernorm(None, 0, None, None, None, 0, 0, 1, 8, 1e-5, 0.0, None)
    assert rc == -1 and b"null pointer" in lib.dfd_last_error()
    rc = lib.dfd_gemm(1 << 12, 48, 1 << 12, 48, capi.BF16, 1 << 12, 8, capi.BF16, None, capi.EPI_BIAS, None, 4, 8, 48, None)
    assert rc == -1 and b"multiple of 32" in lib.dfd_last_error()
    rc = lib.dfd_attention_fwd(1 << 12, 384, 1 << 12, 128, capi.F32, 1, 5, 2, 32, 0.1, None)
    assert rc == -1 and b"head_dim" in lib.dfd_last_error()
    assert lib.dfd_decoder_attn_workspace(16, 12, 64, 8) == 16 * 8 * 12 * 130 * 4


def test_cpu_tensors_fail_loudly(lib):
    x = torch.zeros(4, 8)
    with pytest.raises(capi.DfdError):
        capi.layernorm(x, torch.ones(8), torch.zeros(8), torch.empty(4, 8))


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(capi.DfdError):
        capi.load_library(str(tmp_path / "nope.so"))
