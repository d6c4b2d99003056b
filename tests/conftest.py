"""
Test configuration.

Markers:
  gpu -- needs a real MI355X; run with ``pytest -m gpu`` on the GPU box.  Everything else runs on CPU.
"""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (gfx950) device")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, as build() would."""
    import subprocess

    lib = os.path.join(ROOT, "iscc_search_amd", "csrc", "libisccsearch_hip.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "iscc_search_amd", "csrc")], check=True, capture_output=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)


@pytest.fixture(scope="session")
def hip_engine():
    """One engine for the whole GPU session (a single process on the card)."""
    from iscc_search_amd.engine import HipEngine

    eng = HipEngine(0)
    # ISCC_HIP_OPTS="fold=1,queries_per_pass=16": run the whole GPU tier under non-default engine options
    for item in filter(None, os.environ.get("ISCC_HIP_OPTS", "").split(",")):
        name, value = item.split("=")
        eng.set_option(name.strip(), int(value))
    yield eng
    eng.close()
