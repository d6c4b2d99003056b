"""
Snapshot / restore (the reference's flush/close/reopen of
``usearch://`` indexes, tests/test_indexes_usearch_persistence.py): raw column files per table,
``assets.jsonl`` + ``index.json`` per index, ``hip:///abs/path`` manager.
"""

import json
import os

import numpy as np
import pytest

from helpers import flip_bits, make_asset, sp
from iscc_search_amd import codec
from iscc_search_amd.index import HipIndexManager
from iscc_search_amd.schema import IsccIndex, IsccQuery
from oracle import oracle_topk
from oracle_engine import OracleEngine


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    if request.param == "oracle":
        yield OracleEngine()
    else:
        from iscc_search_amd.engine import HipEngine

        e = HipEngine(0)
        yield e
        e.close()


def _dump(res):
    return json.dumps(res.model_dump(mode="json"), sort_keys=True)


def test_table_roundtrip_files_and_results(engine, tmp_path):
    rng = np.random.default_rng(0)
    n = 5000
    t = engine.open_table(1, 1, 32)
    lens = rng.choice([8, 16, 32, 12], size=n).astype(np.uint8)
    words = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    for j in range(4):
        valid = np.clip(lens.astype(np.int64) - 8 * j, 0, 8)
        mask = np.where(valid == 8, ~np.uint64(0), np.where(valid == 0, np.uint64(0), (~np.uint64(0)) << ((8 * (8 - valid)) % 64).astype(np.uint64)))
        words[:, j] &= mask
    keys = rng.permutation(n).astype(np.uint64) + np.uint64(9)
    t.add(keys, words, lens)
    path = str(tmp_path / "tbl")
    t.save(path, chunk_rows=1000)        # several chunks per segment
    meta = json.load(open(os.path.join(path, "table.json")))
    assert meta["metric"] == 1 and meta["key_words"] == 1 and meta["max_bytes"] == 32
    assert {int(b): r for b, r in meta["segments"].items()} == {int(b): int((lens == b).sum()) for b in (8, 12, 16, 32)}
    # one raw little-endian file per 64-bit word column = the device layout
    assert os.path.getsize(os.path.join(path, "seg32.w3.u64")) == int((lens == 32).sum()) * 8
    assert os.path.getsize(os.path.join(path, "seg12.w1.u64")) == int((lens == 12).sum()) * 8
    assert not os.path.exists(os.path.join(path, "seg12.w2.u64"))
    t2 = engine.open_table(1, 1, 32)
    t2.load(path, chunk_rows=777)
    assert t2.size == n
    q = words[[1, 50, 999]].copy()
    q[:, 0] ^= np.uint64(6)
    qn = lens[[1, 50, 999]]
    # both the original and the reloaded table against the oracle on the arrays they were built from
    want = oracle_topk(1, keys, words, lens, q, qn, 12)
    for name, tbl in (("saved", t), ("reloaded", t2)):
        for a, b, field in zip(tbl.search(q, qn, 12), want, ("keys", "hamming", "prefix_bits", "count")):
            np.testing.assert_array_equal(a, b, err_msg=f"{name} table: {field}")
    # restored tables are fully mutable again
    assert t2.contains(keys[:3]).all() and t2.remove(keys[:3]) == 3 and t2.size == n - 3
    wrong = engine.open_table(0, 1, 8)
    with pytest.raises(ValueError, match="does not match"):
        wrong.load(path)
    for x in (t, t2, wrong):
        x.drop()


def test_manager_snapshot_and_reopen(engine, tmp_path):
    rng = np.random.default_rng(3)
    uri = f"hip://{tmp_path}/store"
    s = [rng.integers(0, 256, size=16, dtype=np.uint8).tobytes() for _ in range(3)]
    assets = [make_asset(rng, i, metadata={"source": f"https://example.com/{i}"}) for i in range(30)]
    assets[0] = assets[0].model_copy(update={"simprints": {"CONTENT_TEXT_V0": [sp(s[0], 0, 10), sp(s[1], 10, 10)]}})
    assets[1] = assets[1].model_copy(update={"simprints": {"CONTENT_TEXT_V0": [sp(flip_bits(s[0], 3), 5, 5)]}})
    m = HipIndexManager(uri, engine=engine)
    m.create_index(IsccIndex(name="main"))
    m.create_index(IsccIndex(name="empty"))
    m.add_assets("main", assets)
    q_units = IsccQuery(iscc_code=assets[7].iscc_code)
    q_sp = IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(s[0])]})
    q_id = IsccQuery(iscc_id=assets[3].iscc_id)
    before = [_dump(m.search_assets("main", q, limit=10)) for q in (q_units, q_sp, q_id)]
    m.close()
    assert os.path.exists(tmp_path / "store" / "main" / "index.json")
    assert os.path.exists(tmp_path / "store" / "main" / "units" / "META_NONE_V0" / "table.json")

    m2 = HipIndexManager(uri, engine=engine)
    assert sorted((i.name, i.assets) for i in m2.list_indexes()) == [("empty", 0), ("main", 30)]
    assert m2.get_index("main").assets == 30                      # answered from index.json, nothing loaded yet
    with pytest.raises(FileExistsError):
        m2.create_index(IsccIndex(name="main"))
    assert m2.get_asset("main", assets[5].iscc_id).metadata == assets[5].metadata
    after = [_dump(m2.search_assets("main", q, limit=10)) for q in (q_units, q_sp, q_id)]
    assert after == before
    # keep writing after a restore: update + flush + reopen
    newer = make_asset(rng, 7)
    assert m2.add_assets("main", [newer])[0].status.value == "updated"
    m2.flush()
    m2.delete_index("empty")
    assert not os.path.exists(tmp_path / "store" / "empty")
    m2.close()
    m3 = HipIndexManager(uri, engine=engine)
    assert [i.name for i in m3.list_indexes()] == ["main"]
    assert m3.get_asset("main", newer.iscc_id).iscc_code == newer.iscc_code
    assert [x.iscc_id for x in m3.search_assets("main", IsccQuery(iscc_code=newer.iscc_code)).global_matches] == [newer.iscc_id]
    assert m3.search_assets("main", q_units).global_matches == [] or all(
        x.iscc_id != assets[7].iscc_id or x.score < 1.0 for x in m3.search_assets("main", q_units).global_matches
    )
    m3.close()


def test_volatile_manager_writes_nothing(tmp_path):
    m = HipIndexManager("hip:///", engine=OracleEngine())
    assert m.base_path is None
    m.create_index(IsccIndex(name="x"))
    m.flush()
    m.close()
    assert list(tmp_path.iterdir()) == []
