"""
INTEGRATION.md section 1 shows the wiring as unified diffs against the reference tree.  The reference cannot be imported
here, so the ADDED lines of each hunk are extracted from the document and executed against small stand-ins for the
objects they touch (`search_opts`, `IndexConfig`, the config manager): a documented snippet that does not construct a
working `HipIndexManager` fails this test.
"""

import os
import re
import textwrap
import types
from urllib.parse import urlparse

import pytest

from iscc_search_amd.index import HipIndexManager
from iscc_search_amd.schema import IsccIndex
from oracle_engine import OracleEngine

DOC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")


def added_lines(section_title):
    """The '+' lines (without the marker) of the ```diff block that follows the given heading."""
    with open(DOC) as f:
        text = f.read()
    start = text.index(section_title)
    block = re.search(r"```diff\n(.*?)```", text[start:], re.S).group(1)
    return [ln[1:] for ln in block.splitlines() if ln.startswith("+") and not ln.startswith("+++")]


@pytest.fixture
def no_gpu(monkeypatch):
    """The snippets construct HipIndexManager(uri): give it the oracle engine instead of a GPU (CPU tier)."""
    real = HipIndexManager.__init__

    def init(self, uri="hip:///", engine=None, options=None):
        real(self, uri, engine=engine or OracleEngine(), options=options)

    monkeypatch.setattr(HipIndexManager, "__init__", init)


def test_factory_branch_of_options_get_index(no_gpu, tmp_path):
    lines = added_lines("### 1a.")
    body = [ln for ln in lines if "supported = " not in ln]
    assert any('parsed.scheme == "hip"' in ln for ln in body)
    src = "def get_index(uri, search_opts):\n    parsed = urlparse(uri)\n" + "\n".join(body) + "\n    raise ValueError('unsupported')\n"
    ns = {"urlparse": urlparse}
    exec(compile(src, "INTEGRATION.md#1a", "exec"), ns)
    # exactly the knobs SearchOptions has (iscc_search/options.py:139-164): an attribute the hunk reads beyond them fails here
    opts = types.SimpleNamespace(match_threshold_units=0.8, match_threshold_simprints=0.7, confidence_exponent=3, oversampling_factor=10)
    for uri in ("hip:///", f"hip://{tmp_path}/store?device=0"):
        m = ns["get_index"](uri, opts)
        assert isinstance(m, HipIndexManager)
        assert (m._opts.match_threshold_units, m._opts.confidence_exponent, m._opts.oversampling_factor, m._opts.max_dim) == (0.8, 3, 10, 256)
        m.create_index(IsccIndex(name="x"))
        assert m.get_index("x").assets == 0
        m.close()
    with pytest.raises(ValueError, match="unsupported"):
        ns["get_index"]("redis://x", opts)


def test_config_entry_and_cli_branch(no_gpu, tmp_path):
    cfg_lines = added_lines("### 1b.")
    cls_src = textwrap.dedent("\n".join(ln for ln in cfg_lines if not ln.lstrip().startswith(("elif type_", "return HipIndexConfig"))))
    ns = {}
    exec("class IndexConfig:\n    def __init__(self, name, type_):\n        self.name, self.type = name, type_\n", ns)
    exec(compile(cls_src, "INTEGRATION.md#1b", "exec"), ns)
    cfg = ns["HipIndexConfig"]("films", uri=f"hip://{tmp_path}/cli")
    assert cfg.to_dict() == {"type": "hip", "uri": f"hip://{tmp_path}/cli"} and cfg.name == "films"
    # the from_dict hunk
    branch = [ln.strip() for ln in cfg_lines if ln.lstrip().startswith(("elif type_", "return HipIndexConfig"))]
    assert branch == ['elif type_ == "hip":', 'return HipIndexConfig(name=name, uri=data.get("uri", "hip:///"))']

    cli_lines = added_lines("### 1c.")
    body = [ln for ln in cli_lines if "import get_config_manager" not in ln]
    assert body[0].strip() == "elif isinstance(index_config, HipIndexConfig):"
    src = "def branch(index_config, target_name):\n    if False:\n        pass\n" + "\n".join(body) + "\n"
    # the reference's IsccIndex is imported inside the hunk; hand it this package's (same field names)
    import sys

    fake = types.ModuleType("iscc_search.schema")
    fake.IsccIndex = IsccIndex
    pkg = types.ModuleType("iscc_search")
    sys.modules.setdefault("iscc_search", pkg)
    sys.modules["iscc_search.schema"] = fake
    try:
        cli = {"HipIndexConfig": ns["HipIndexConfig"]}
        exec(compile(src, "INTEGRATION.md#1c", "exec"), cli)
        manager, name = cli["branch"](cfg, "films")
        assert isinstance(manager, HipIndexManager) and name == "films" and manager.get_index("films").assets == 0
        manager.close()
        again, _ = cli["branch"](cfg, "films")          # second CLI invocation: the index is found on disk, not re-created
        assert [i.name for i in again.list_indexes()] == ["films"]
        again.close()
    finally:
        sys.modules.pop("iscc_search.schema", None)
        if sys.modules.get("iscc_search") is pkg:
            sys.modules.pop("iscc_search", None)
