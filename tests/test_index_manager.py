"""
Behavioural contract of the ``hip:///`` backend at the ``IsccIndexProtocol`` boundary.

Restates (own assertions, own sample data) what the reference asserts for every backend:
  tests/test_indexes_memory_index.py   lifecycle, error messages, status, get/search, isolation
  tests/test_server_search.py          search result shape, limit, self-exclusion, iscc_id precedence
  tests/test_protocols_index.py        structural protocol conformance
  tests/test_indexes_usearch_index.py  score semantics (:141-215), threshold filter (:803-826)

Runs twice: on CPU with the oracle-backed engine (host logic), and -- marked gpu -- through the
real HIP engine, where both must agree.
"""

import inspect

import numpy as np
import pytest

from helpers import flip_bits, make_asset, make_iscc_id, rnd_unit
from iscc_search_amd import codec
from iscc_search_amd.index import HipIndexManager, HipOptions, get_index, normalize_query
from iscc_search_amd.schema import IsccEntry, IsccIndex, IsccQuery, Status
from oracle_engine import OracleEngine

PROTOCOL_METHODS = ("list_indexes", "create_index", "get_index", "delete_index", "add_assets", "get_asset", "search_assets", "close")


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def manager(request):
    if request.param == "oracle":
        m = HipIndexManager("hip:///", engine=OracleEngine())
    else:
        m = HipIndexManager("hip:///")
    yield m
    m.close()


@pytest.fixture
def rng():
    return np.random.default_rng(42)


def test_protocol_conformance_by_method_names():
    """tests/test_protocols_index.py:160-186: conformance is structural."""
    for name in PROTOCOL_METHODS:
        assert callable(getattr(HipIndexManager, name))
    sig = inspect.signature(HipIndexManager.search_assets)
    assert list(sig.parameters)[1:] == ["index_name", "query", "limit"] and sig.parameters["limit"].default == 100


def test_factory_scheme_dispatch():
    """tests/test_options.py:126-155 pattern for the new scheme."""
    m = get_index("hip:///", engine=OracleEngine())
    assert isinstance(m, HipIndexManager)
    with pytest.raises(ValueError, match="explicit scheme"):
        get_index("/some/path")
    with pytest.raises(ValueError, match="Unsupported index URI scheme"):
        get_index("redis://x")
    assert HipIndexManager("hip:///?device=3", engine=OracleEngine()).device_id == 3


def test_index_lifecycle_and_errors(manager):
    assert manager.list_indexes() == []
    created = manager.create_index(IsccIndex(name="alpha"))
    assert (created.name, created.assets, created.size) == ("alpha", 0, 0)
    with pytest.raises(FileExistsError, match="Index 'alpha' already exists"):
        manager.create_index(IsccIndex(name="alpha"))
    manager.create_index(IsccIndex(name="beta2"))
    assert sorted(i.name for i in manager.list_indexes()) == ["alpha", "beta2"]
    assert manager.get_index("alpha").assets == 0
    with pytest.raises(FileNotFoundError, match="Index 'nope' not found"):
        manager.get_index("nope")
    manager.delete_index("alpha")
    with pytest.raises(FileNotFoundError, match="Index 'alpha' not found"):
        manager.delete_index("alpha")
    assert [i.name for i in manager.list_indexes()] == ["beta2"]


@pytest.mark.parametrize("bad", ["Upper", "1abc", "with-dash", "under_score", "", "x" * 33])
def test_invalid_index_names(bad):
    m = HipIndexManager("hip:///", engine=OracleEngine())
    with pytest.raises(ValueError):
        m.create_index(IsccIndex.model_construct(name=bad))


def test_add_assets_status_and_count(manager, rng):
    manager.create_index(IsccIndex(name="t"))
    a, b = make_asset(rng, 0), make_asset(rng, 1)
    res = manager.add_assets("t", [a, b])
    assert [(r.iscc_id, r.status) for r in res] == [(a.iscc_id, Status.created), (b.iscc_id, Status.created)]
    assert manager.get_index("t").assets == 2
    # re-adding the same iscc_id updates, the count is unchanged (tests/test_indexes_memory_index.py:169-189)
    a2 = make_asset(rng, 0)
    res = manager.add_assets("t", [a2])
    assert res[0].status == Status.updated and manager.get_index("t").assets == 2
    assert manager.get_asset("t", a.iscc_id).iscc_code == a2.iscc_code
    # identical re-add is an idempotent no-op that still reports 'updated'
    assert manager.add_assets("t", [a2])[0].status == Status.updated
    # duplicate ids inside one batch: every input gets a positional result, last one wins
    c1, c2 = make_asset(rng, 5), make_asset(rng, 5)
    res = manager.add_assets("t", [c1, c2])
    assert [r.status for r in res] == [Status.created, Status.updated]
    assert manager.get_asset("t", c1.iscc_id).iscc_code == c2.iscc_code


def test_add_assets_errors(manager, rng):
    with pytest.raises(FileNotFoundError, match="Index 'missing' not found"):
        manager.add_assets("missing", [make_asset(rng, 0)])
    manager.create_index(IsccIndex(name="t"))
    no_id = make_asset(rng, 1).model_copy(update={"iscc_id": None})
    with pytest.raises(ValueError, match="Asset must have iscc_id field when adding to index"):
        manager.add_assets("t", [no_id])
    manager.add_assets("t", [make_asset(rng, 2)])
    other_realm = make_asset(rng, 3).model_copy(update={"iscc_id": make_iscc_id(3, realm=1)})
    with pytest.raises(ValueError, match="Realm ID mismatch"):
        manager.add_assets("t", [other_realm])
    assert manager.add_assets("t", []) == []


def test_get_asset(manager, rng):
    manager.create_index(IsccIndex(name="t"))
    a = make_asset(rng, 0, metadata={"source": "https://example.com/a", "name": "A"})
    manager.add_assets("t", [a])
    got = manager.get_asset("t", a.iscc_id)
    assert (got.iscc_id, got.iscc_code, got.units, got.metadata) == (a.iscc_id, a.iscc_code, a.units, a.metadata)
    with pytest.raises(FileNotFoundError, match=f"Asset '{make_iscc_id(9)}' not found in index 't'"):
        manager.get_asset("t", make_iscc_id(9))
    with pytest.raises(FileNotFoundError, match="Index 'zzz' not found"):
        manager.get_asset("zzz", a.iscc_id)
    with pytest.raises(ValueError):
        manager.get_asset("t", "ISCC:AAAUHBUDQUT3LPWR")   # a unit, not an ISCC-ID


def test_search_exact_match_scores_one_and_shape(manager, rng):
    manager.create_index(IsccIndex(name="t"))
    assets = [make_asset(rng, i, metadata={"source": f"https://example.com/{i}"}) for i in range(8)]
    manager.add_assets("t", assets)
    q = IsccQuery(iscc_code=assets[3].iscc_code)
    res = manager.search_assets("t", q, limit=5)
    assert res.query.iscc_code == assets[3].iscc_code and res.query.units     # normalised: units derived
    assert res.chunk_matches == []
    top = res.global_matches[0]
    assert top.iscc_id == assets[3].iscc_id and top.score == 1.0
    assert set(top.types) == {"META_NONE_V0", "CONTENT_TEXT_V0", "DATA_NONE_V0", "INSTANCE_NONE_V0"}
    assert all(v == 1.0 for v in top.types.values())
    assert top.source == "https://example.com/3"
    assert len(res.global_matches) <= 5
    scores = [m.score for m in res.global_matches]
    assert scores == sorted(scores, reverse=True)


def test_search_limit_and_empty_index(manager, rng):
    manager.create_index(IsccIndex(name="t"))
    base = make_asset(rng, 0)
    assert manager.search_assets("t", IsccQuery(iscc_code=base.iscc_code)).global_matches == []
    same = [base.model_copy(update={"iscc_id": make_iscc_id(i)}) for i in range(10)]
    manager.add_assets("t", same)
    res = manager.search_assets("t", IsccQuery(iscc_code=base.iscc_code), limit=4)
    assert len(res.global_matches) == 4 and all(m.score == 1.0 for m in res.global_matches)
    # ties are ordered by ascending key in the engine and kept by the stable sort
    assert [m.iscc_id for m in res.global_matches] == [make_iscc_id(i) for i in range(4)]
    with pytest.raises(FileNotFoundError, match="Index 'other' not found"):
        manager.search_assets("other", IsccQuery(iscc_code=base.iscc_code))


def test_search_by_iscc_id_precedence_and_self_exclusion(manager, rng):
    """tests/test_server_search.py:184-263."""
    manager.create_index(IsccIndex(name="t"))
    base = make_asset(rng, 0)
    twin = base.model_copy(update={"iscc_id": make_iscc_id(1)})
    other = make_asset(rng, 2)
    manager.add_assets("t", [base, twin, other])
    # iscc_id wins over a contradicting iscc_code; the query asset itself is excluded
    res = manager.search_assets("t", IsccQuery(iscc_id=base.iscc_id, iscc_code=other.iscc_code))
    ids = [m.iscc_id for m in res.global_matches]
    assert base.iscc_id not in ids and twin.iscc_id in ids and other.iscc_id not in ids
    assert res.query.iscc_code == base.iscc_code
    with pytest.raises(FileNotFoundError, match="not found in index 't'"):
        manager.search_assets("t", IsccQuery(iscc_id=make_iscc_id(77)))


def test_similarity_scores_follow_nphd(manager, rng):
    """score = 1 - NPHD per unit; aggregate = sum(s^4)/sum(s) over units >= 0.75 (usearch/index.py:808-828)."""
    manager.create_index(IsccIndex(name="t"))
    body = bytes([255, 170, 85, 0] * 4)                     # tests/conftest.py:209-228 similar_units
    base = codec.encode_unit(codec.MT_META, 0, 0, body)
    near = codec.encode_unit(codec.MT_META, 0, 0, flip_bits(body, 1))
    far = codec.encode_unit(codec.MT_META, 0, 0, bytes([0, 85, 170, 255] * 4))
    mk = lambda i, meta: IsccEntry(iscc_id=make_iscc_id(i), units=[meta, rnd_unit(rng, codec.MT_DATA), rnd_unit(rng, codec.MT_INSTANCE)])
    manager.add_assets("t", [mk(0, base), mk(1, near), mk(2, far)])
    res = manager.search_assets("t", IsccQuery(units=[base]))
    by_id = {m.iscc_id: m for m in res.global_matches}
    assert by_id[make_iscc_id(0)].types == {"META_NONE_V0": 1.0}
    assert by_id[make_iscc_id(1)].types == {"META_NONE_V0": 1.0 - 1 / 128}        # 0.9921875
    assert by_id[make_iscc_id(1)].score == pytest.approx((1 - 1 / 128) ** 3)
    assert make_iscc_id(2) not in by_id                                            # score 0.0 < threshold
    # 64-bit query against 128-bit rows: only the common prefix counts
    short = codec.encode_unit(codec.MT_META, 0, 0, body[:8])
    res = manager.search_assets("t", IsccQuery(units=[short]))
    top2 = {m.iscc_id: m.types["META_NONE_V0"] for m in res.global_matches}
    assert top2[make_iscc_id(0)] == 1.0 and top2[make_iscc_id(1)] == 1.0 - 1 / 64


def test_threshold_filters_low_confidence(rng):
    """tests/test_indexes_usearch_index.py:803-826: nothing passes a 0.99 threshold but exact matches."""
    m = HipIndexManager("hip:///", engine=OracleEngine(), options=HipOptions(match_threshold_units=0.99))
    m.create_index(IsccIndex(name="t"))
    assets = [make_asset(rng, i) for i in range(20)]
    m.add_assets("t", assets)
    res = m.search_assets("t", IsccQuery(units=[rnd_unit(rng, codec.MT_META), rnd_unit(rng, codec.MT_DATA), rnd_unit(rng, codec.MT_INSTANCE)]))
    assert res.global_matches == []
    res = m.search_assets("t", IsccQuery(iscc_code=assets[4].iscc_code))
    assert [x.iscc_id for x in res.global_matches] == [assets[4].iscc_id]


def test_instance_units_match_by_prefix_both_ways(manager, rng):
    """tests/test_indexes_usearch_index.py:141-215: any INSTANCE prefix match scores exactly 1.0."""
    manager.create_index(IsccIndex(name="t"))
    inst256 = rng.integers(0, 256, size=32, dtype=np.uint8).tobytes()
    mk = lambda i, inst: IsccEntry(iscc_id=make_iscc_id(i), units=[rnd_unit(rng, codec.MT_DATA), codec.encode_unit(codec.MT_INSTANCE, 0, 0, inst)])
    manager.add_assets("t", [mk(0, inst256), mk(1, inst256[:16]), mk(2, inst256[:8]), mk(3, flip_bits(inst256, 1)[:8])])
    for qlen in (8, 16, 32):
        res = manager.search_assets("t", IsccQuery(units=[codec.encode_unit(codec.MT_INSTANCE, 0, 0, inst256[:qlen])]))
        got = {m.iscc_id: m.types for m in res.global_matches}
        assert set(got) == {make_iscc_id(0), make_iscc_id(1), make_iscc_id(2)}
        assert all(t == {"INSTANCE_NONE_V0": 1.0} for t in got.values())


def test_more_instance_matches_than_the_first_short_list(manager, rng):
    """An INSTANCE match asks for 64 records first; 150 assets sharing a prefix come back through the second, full-length request."""
    from iscc_search_amd.index import INSTANCE_FIRST_K

    manager.create_index(IsccIndex(name="t"))
    inst = rng.integers(0, 256, size=8, dtype=np.uint8).tobytes()
    n = 150
    assert n > INSTANCE_FIRST_K
    manager.add_assets("t", [IsccEntry(iscc_id=make_iscc_id(i), units=[rnd_unit(rng, codec.MT_DATA), codec.encode_unit(codec.MT_INSTANCE, 0, 0, inst)]) for i in range(n)]
                       + [IsccEntry(iscc_id=make_iscc_id(n), units=[rnd_unit(rng, codec.MT_DATA), rnd_unit(rng, codec.MT_INSTANCE)])])
    res = manager.search_assets("t", IsccQuery(units=[codec.encode_unit(codec.MT_INSTANCE, 0, 0, inst)]), limit=1000)
    assert {m.iscc_id for m in res.global_matches} == {make_iscc_id(i) for i in range(n)}
    idx = manager._indexes["t"]
    assert len(idx._search_instance_unit("INSTANCE_NONE_V0", inst)) == n


def test_single_unit_helpers_agree_with_the_batched_path(manager, rng):
    """``_search_similarity_unit`` / ``_search_instance_unit`` (the reference's per-unit methods, usearch/index.py:2024-2045,
    :1957-2022) return what ``_search_units`` (one engine call for all units of a request) merges."""
    manager.create_index(IsccIndex(name="t"))
    assets = [make_asset(rng, i, bits=128) for i in range(30)]
    manager.add_assets("t", assets)
    idx = manager._index("t")
    units = assets[3].units
    merged = idx._search_units(units, 10)
    for unit_str in units:
        unit = codec.Iscc(unit_str)
        if unit.unit_type.startswith("INSTANCE_"):
            single = idx._search_instance_unit(unit.unit_type, unit.body)
        else:
            single = idx._search_similarity_unit(unit.unit_type, unit.body, 10)
        assert single == {key: types[unit.unit_type] for key, types in merged.items() if unit.unit_type in types}


def test_units_only_query_and_isolation(manager, rng):
    manager.create_index(IsccIndex(name="a"))
    manager.create_index(IsccIndex(name="b"))
    x, y = make_asset(rng, 0), make_asset(rng, 1)
    manager.add_assets("a", [x])
    manager.add_assets("b", [y])
    res = manager.search_assets("a", IsccQuery(units=x.units))
    assert [m.iscc_id for m in res.global_matches] == [x.iscc_id]
    assert res.query.iscc_code == x.iscc_code                    # derived from the units
    assert manager.search_assets("b", IsccQuery(units=x.units)).global_matches == []
    with pytest.raises(FileNotFoundError):
        manager.get_asset("b", x.iscc_id)


def test_normalize_query_cases():
    """iscc_search/indexes/common.py:275-330, the five cases."""
    units = ["ISCC:AAAUHBUDQUT3LPWR", "ISCC:GAAVB2JS4SVPWSEE", "ISCC:IAATI64Q5HJYOXFF"]
    code = codec.gen_iscc_code(units)
    both = IsccQuery(iscc_code=code, units=units)
    assert normalize_query(both) is both
    assert normalize_query(IsccQuery(units=units)).iscc_code == code
    assert normalize_query(IsccQuery(iscc_code=code)).units == units
    meta_only = normalize_query(IsccQuery(units=units[:1]))
    assert meta_only.iscc_code is None and meta_only.units == units[:1]
    sp_only = IsccQuery(simprints={"CONTENT_TEXT_V0": ["AAAAAAAAAAA"]})
    assert normalize_query(sp_only) is sp_only
    with pytest.raises(ValueError, match="Query must have 'iscc_code', 'units', or 'simprints' for search"):
        normalize_query(IsccQuery())


def test_update_replaces_vectors(manager, rng):
    manager.create_index(IsccIndex(name="t"))
    old = make_asset(rng, 0)
    manager.add_assets("t", [old])
    new = make_asset(rng, 0)
    manager.add_assets("t", [new])
    assert manager.search_assets("t", IsccQuery(iscc_code=old.iscc_code)).global_matches == []
    assert [m.iscc_id for m in manager.search_assets("t", IsccQuery(iscc_code=new.iscc_code)).global_matches] == [new.iscc_id]


def test_update_that_drops_the_instance_unit_stops_matching_it(manager, rng):
    """usearch/index.py:338-348: an INSTANCE row the update no longer carries is deleted."""
    manager.create_index(IsccIndex(name="t"))
    full = make_asset(rng, 0)
    other = make_asset(rng, 1)
    manager.add_assets("t", [full, other])
    instance_unit = [u for u in full.units if codec.Iscc(u).unit_type.startswith("INSTANCE_")][0]
    hit = manager.search_assets("t", IsccQuery(units=[instance_unit])).global_matches
    assert [m.iscc_id for m in hit] == [full.iscc_id] and hit[0].score == 1.0
    kept = [u for u in full.units if u != instance_unit]
    without = full.model_copy(update={"units": kept, "iscc_code": None})      # no INSTANCE unit: no ISCC-CODE can be composed
    assert manager.add_assets("t", [without])[0].status == Status.updated
    assert manager.search_assets("t", IsccQuery(units=[instance_unit])).global_matches == []
    # the similarity units it still carries keep matching
    assert [m.iscc_id for m in manager.search_assets("t", IsccQuery(units=kept[:1])).global_matches][:1] == [full.iscc_id]


def test_failed_device_update_is_rolled_back_and_can_be_retried(rng):
    """host state must not run ahead of the device, or the idempotent gate makes the asset unsearchable for good."""
    from helpers import sp

    engine = OracleEngine()
    m = HipIndexManager("hip:///", engine=engine)
    m.create_index(IsccIndex(name="t"))
    s0 = bytes(range(16))
    first = make_asset(rng, 0)
    m.add_assets("t", [first])
    second = make_asset(rng, 0, simprints={"CONTENT_TEXT_V0": [sp(s0, 0, 10)]})
    idx = m._indexes["t"]
    real_add = type(idx._unit_table("DATA_NONE_V0")).add
    calls = {"n": 0}

    def failing_add(self, keys, vectors):
        calls["n"] += 1
        if calls["n"] == 2:
            raise MemoryError("hipMalloc failed (injected)")
        return real_add(self, keys, vectors)

    type(idx._unit_table("DATA_NONE_V0")).add = failing_add
    try:
        with pytest.raises(MemoryError):
            m.add_assets("t", [second])
    finally:
        type(idx._unit_table("DATA_NONE_V0")).add = real_add
    # the host still describes the version that is fully on the device ...
    assert m.get_asset("t", first.iscc_id).iscc_code == first.iscc_code
    # ... and the SAME batch goes through on retry (no idempotent-skip) and becomes searchable in every table
    assert m.add_assets("t", [second])[0].status == Status.updated
    assert [x.iscc_id for x in m.search_assets("t", IsccQuery(iscc_code=second.iscc_code)).global_matches] == [second.iscc_id]
    assert m.search_assets("t", IsccQuery(iscc_code=first.iscc_code)).global_matches == []
    chunk = m.search_assets("t", IsccQuery(simprints={"CONTENT_TEXT_V0": [codec.encode_base64(s0)]})).chunk_matches
    assert [c.iscc_id for c in chunk] == [second.iscc_id] and chunk[0].types["CONTENT_TEXT_V0"].matches == 1
    m.close()


def test_limits_beyond_the_engine_cap_are_refused_not_truncated(manager, rng):
    """the reference passes `limit` on unbounded (usearch/index.py:2037); a silent cut is not acceptable."""
    manager.create_index(IsccIndex(name="t"))
    a = make_asset(rng, 0)
    manager.add_assets("t", [a])
    assert len(manager.search_assets("t", IsccQuery(iscc_code=a.iscc_code), limit=4096).global_matches) == 1
    with pytest.raises(ValueError, match="exceeds the 4096 neighbours"):
        manager.search_assets("t", IsccQuery(iscc_code=a.iscc_code), limit=4097)


def test_close_is_idempotent(rng):
    m = HipIndexManager("hip:///", engine=OracleEngine())
    m.create_index(IsccIndex(name="t"))
    m.add_assets("t", [make_asset(rng, 0)])
    m.close()
    m.close()
    assert m.list_indexes() == []


def test_config1_ten_thousand_units_plumbing(rng):
    """BASELINE config 1: 10 000 random 64-bit units through the full protocol on CPU (plumbing only)."""
    m = HipIndexManager("hip:///", engine=OracleEngine())
    m.create_index(IsccIndex(name="big"))
    assets = [make_asset(rng, i, with_meta=False, with_content=False) for i in range(2500)]   # 2 units x 2500 x ... = 10 000 codes with twins below
    twins = [a.model_copy(update={"iscc_id": make_iscc_id(10_000 + i)}) for i, a in enumerate(assets)]
    m.add_assets("big", assets + twins)
    assert m.get_index("big").assets == 5000
    res = m.search_assets("big", IsccQuery(iscc_id=assets[17].iscc_id), limit=10)
    assert res.global_matches[0].iscc_id == twins[17].iscc_id and res.global_matches[0].score == 1.0
    assert assets[17].iscc_id not in [x.iscc_id for x in res.global_matches]
    assert len(res.global_matches) <= 10
