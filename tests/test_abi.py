"""The C-ABI library: loads, exports every symbol the header declares, fails loudly without a GPU."""

import ctypes
import os
import re

import pytest

from iscc_search_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    with open(os.path.join(ROOT, "include", "isccsearch.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isccsearch_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_the_same_entry_points():
    assert _header_functions() == sorted(_lib.EXPORTS)


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load_library()
    for name in _header_functions():
        assert hasattr(lib, name), name


def test_record_layout_matches_header():
    assert _lib.RECORD_DTYPE.itemsize == 24
    assert [_lib.RECORD_DTYPE.fields[n][1] for n in ("key_hi", "key_lo", "dist_rank", "hamming", "prefix_bits")] == [0, 8, 16, 20, 22]


def test_header_cites_reference_call_sites():
    with open(os.path.join(ROOT, "include", "isccsearch.h")) as f:
        text = f.read()
    for cite in ("usearch/index.py:2037", "usearch_core.py:165", "usearch/index.py:440", "usearch_core.py:221"):
        assert cite in text


def test_create_fails_loudly_without_a_device():
    """No GPU in the CPU tier: create must return an error code and a message, never fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load_library()
    h = ctypes.c_void_p()
    rc = lib.isccsearch_create(0, ctypes.byref(h))
    assert rc < 0 and not h.value
    assert _lib.last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)
    from iscc_search_amd.engine import HipEngine

    with pytest.raises(RuntimeError):
        HipEngine(0)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under iscc_search_amd/ may reference it."""
    pkg = os.path.join(ROOT, "iscc_search_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
                assert "liboracle" not in src, fn


def test_plain_c_client_compiles_and_links_against_the_header(tmp_path):
    """tests/abi_client.c is the ABI as a C host sees it (the GPU tier runs it: tests/test_gpu_abi_client.py)."""
    import subprocess

    csrc = os.path.join(ROOT, "iscc_search_amd", "csrc")
    subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_client.c"), "-o", str(tmp_path / "abi_client"), "-L", csrc, "-lisccsearch_hip",
                    "-Wl,-rpath," + csrc], check=True)
