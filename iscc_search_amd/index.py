"""
``hip:///`` backend: ``HipIndexManager`` implements the reference's ``IsccIndexProtocol``
(``iscc_search/protocols/index.py:19-174``) over the HIP engine.

Layout mirrors the reference's usearch backend (``iscc_search/indexes/usearch/``):
  ``HipIndexManager``  ~ ``UsearchIndexManager`` (``manager.py:25-335``): named indexes, protocol methods
  ``HipIndex``         ~ ``UsearchIndex`` (``index.py:87-2045``): one index = asset store + one NPHD
                         table per unit type + one simprint table per simprint type
What differs by design: there is no LMDB and no HNSW file -- assets live in a host dict (as in the
reference's ``memory://`` backend, ``memory/index.py``), codes live in HBM, and every similarity
search is the exact GPU scan.  Scoring, thresholds, aggregation, self-exclusion, ordering and error
messages follow ``index.py:735-881`` and ``:1357-1469``.
"""

import json
import os
import re
import shutil
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional
from urllib.parse import parse_qs, urlparse

import numpy as np

from iscc_search_amd import codec
from iscc_search_amd._lib import MAX_K

INSTANCE_FIRST_K = 64   # records per query an INSTANCE prefix match asks for first (a full list is asked again up to MAX_K)
from iscc_search_amd.engine import pack_bytes
from iscc_search_amd.nphd import HipNphdIndex
from iscc_search_amd.schema import (
    IsccAddResult, IsccChunkMatch, IsccEntry, IsccGlobalMatch, IsccIndex, IsccMatchedChunk, IsccQuery,
    IsccSearchResult, Status, Types,
)
from iscc_search_amd.simprint import HipSimprintIndex, pack_chunk_pointer

INDEX_NAME_RE = re.compile(r"^[a-z][a-z0-9]*$")


@dataclass
class HipOptions:
    """Knobs with the reference's names and defaults (``iscc_search/options.py:139-164, :95-100``)."""

    match_threshold_units: float = 0.75
    match_threshold_simprints: float = 0.75
    confidence_exponent: int = 4
    oversampling_factor: int = 20
    max_dim: int = 256


def validate_index_name(name):
    # type: (str) -> None
    if not isinstance(name, str) or not INDEX_NAME_RE.match(name) or len(name) > 32:
        raise ValueError(
            f"Invalid index name: '{name}'. Must match pattern ^[a-z][a-z0-9]*$ "
            f"(start with lowercase letter, followed by lowercase letters/digits only)"
        )


def normalize_query(query):
    # type: (IsccQuery) -> IsccQuery
    """Bidirectional units <-> iscc_code normalisation (``iscc_search/indexes/common.py:275-330``)."""
    if query.units and query.iscc_code:
        return query
    if query.units and not query.iscc_code:
        try:
            return query.model_copy(update={"iscc_code": codec.gen_iscc_code(list(query.units), wide=True)})
        except ValueError:
            return query
    if query.iscc_code and not query.units:
        return query.model_copy(update={"units": [str(u) for u in codec.code_units(query.iscc_code)]})
    if query.simprints:
        return query
    raise ValueError("Query must have 'iscc_code', 'units', or 'simprints' for search")


def _sp_string(s):
    return s.root if hasattr(s, "root") else s


def _checked_limit(limit):
    # type: (int) -> int
    """
    The engine returns at most ``MAX_K`` (4 096) neighbours per query.  The reference hands ``limit`` to usearch
    unbounded (``usearch/index.py:2037``); rather than silently returning a shorter list than was asked for, a larger
    ``limit`` is refused (ValueError -> HTTP 400, ``server/search.py:43-46``).
    """
    if limit > MAX_K:
        raise ValueError(f"limit {limit} exceeds the {MAX_K} neighbours per query this backend returns")
    return max(1, limit)


def _check_instance_hits(count, unit_type):
    # type: (int, str) -> None
    """An identity (INSTANCE) match list that fills the engine's cap would be cut silently: refuse instead."""
    if count >= MAX_K:
        raise ValueError(f"more than {MAX_K - 1} assets share the queried {unit_type} prefix; refine the query (longer code)")


class HipIndex:
    """One named index: host asset store + device tables."""

    def __init__(self, engine, options=None):
        # type: (object, HipOptions | None) -> None
        self._engine = engine
        self._opts = options or HipOptions()
        self._lock = threading.RLock()
        self._realm_id = None  # type: Optional[int]
        self._assets = {}  # type: Dict[int, IsccEntry]          key -> entry (without simprints)
        self._asset_units = {}  # type: Dict[int, Dict[str, bytes]]  key -> {unit_type: body} as indexed
        self._unit_tables = {}  # type: Dict[str, HipNphdIndex]
        self._sp_tables = {}  # type: Dict[str, HipSimprintIndex]
        self._sp_assets = {}  # type: Dict[str, Dict[bytes, list]]   sp_type -> body -> [(sp_bytes, chunk_ptr)]
        self._suspect = set()  # asset keys whose last device update failed half-way: their next add cleans up first
        self.dirty = False

    # -- helpers ---------------------------------------------------------------------------------
    def __len__(self):
        return len(self._assets)

    def _unit_table(self, unit_type):
        t = self._unit_tables.get(unit_type)
        if t is None:
            t = self._unit_tables[unit_type] = HipNphdIndex(self._engine, max_dim=self._opts.max_dim)
        return t

    def _sp_table(self, sp_type, ndim):
        t = self._sp_tables.get(sp_type)
        if t is None:
            t = self._sp_tables[sp_type] = HipSimprintIndex(self._engine, ndim=ndim, oversampling_factor=self._opts.oversampling_factor)
            self._sp_assets[sp_type] = {}
        return t

    @staticmethod
    def _fingerprint(sp_list):
        return tuple(sorted((codec.decode_base64(sp.simprint), sp.offset, sp.size) for sp in sp_list))

    # -- add ---------------------------------------------------------------------------------------
    def add_assets(self, assets):
        # type: (List[IsccEntry]) -> List[IsccAddResult]
        """Semantics of ``usearch/index.py:194-537`` (status, batch dedup keep-last, remove-before-add)."""
        if not assets:
            return []
        with self._lock:
            if self._realm_id is None:
                if assets[0].iscc_id is None:
                    raise ValueError("Asset must have iscc_id field when adding to index")
                self._realm_id = codec.validate_iscc_id(assets[0].iscc_id).stype
            results = []
            unit_batches = {}  # type: Dict[str, Dict[int, bytes]]
            updated_keys = set()
            sp_batches = {}  # type: Dict[str, tuple]
            sp_deleted = {}  # type: Dict[str, list]
            last_occurrence = {a.iscc_id: i for i, a in enumerate(assets)}
            batch_seen = set()
            staged = []
            for i, asset in enumerate(assets):
                if asset.iscc_id is None:
                    raise ValueError("Asset must have iscc_id field when adding to index")
                id_obj = codec.validate_iscc_id(asset.iscc_id)
                if id_obj.stype != self._realm_id:
                    raise ValueError(
                        f"Realm ID mismatch: index has realm={self._realm_id}, but asset '{asset.iscc_id}' "
                        f"has realm={id_obj.stype}. All assets in an index must have the same realm ID."
                    )
                key = int.from_bytes(id_obj.body, "big")
                existing = self._assets.get(key)
                status = Status.updated if (existing is not None or key in batch_seen) else Status.created
                batch_seen.add(key)
                results.append(IsccAddResult(iscc_id=asset.iscc_id, status=status))
                if i != last_occurrence[asset.iscc_id]:
                    continue
                stored = asset.model_copy(update={"simprints": None})
                fingerprints = {t: self._fingerprint(lst) for t, lst in (asset.simprints or {}).items()}
                # idempotent re-add: nothing to do when entry and simprints are already indexed identically
                if existing is not None and key not in self._suspect and existing == stored and all(
                    self._fingerprint_of(t, id_obj.body) == fp for t, fp in fingerprints.items()
                ):
                    continue
                if existing is not None:
                    updated_keys.add(key)
                # validate / decode units before touching any state
                unit_map = {}
                for unit_str in asset.units or []:
                    unit = codec.Iscc(unit_str)
                    unit_map[unit.unit_type] = unit.body   # same type at two lengths: last one wins (:423-430)
                sp_decoded = {}
                for sp_type, sp_list in (asset.simprints or {}).items():
                    if not sp_list:
                        continue           # a type listed without chunks indexes nothing
                    sp_decoded[sp_type] = [(codec.decode_base64(sp.simprint), pack_chunk_pointer(id_obj.body, sp.offset, sp.size)) for sp in sp_list]
                staged.append((key, id_obj.body, stored, unit_map, sp_decoded))

            # host state is published first (searches read it without the lock) and rolled back if a device call fails,
            # so that a failed batch can simply be retried (the reference keeps its no-op gate honest the same way:
            # _nphd_units_present / _simprints_present_in_derived, usearch/index.py:539-679)
            undo = []                              # (key, body, old entry, old unit map, {sp_type: old pairs})
            dropped = {}  # type: Dict[str, list]  unit_type -> keys whose update no longer carries that type
            for key, body, stored, unit_map, sp_decoded in staged:
                old_units = self._asset_units.get(key, {})
                undo.append((key, body, self._assets.get(key), self._asset_units.get(key),
                             {t: self._sp_assets.get(t, {}).get(body) for t in sp_decoded}))
                self._assets[key] = stored
                for unit_type, ubody in unit_map.items():
                    unit_batches.setdefault(unit_type, {})[key] = ubody
                # an INSTANCE unit the new version no longer carries must leave its table: left behind it would keep
                # prefix-matching as a 1.0 identity hit (usearch/index.py:338-348).  Similarity units the update drops
                # stay, as in the reference, which removes only from the indexes of the types the NEW version carries (:432-441)
                for unit_type in old_units:
                    if unit_type not in unit_map and unit_type.startswith("INSTANCE_"):
                        dropped.setdefault(unit_type, []).append(key)
                self._asset_units[key] = unit_map
                for sp_type, pairs in sp_decoded.items():
                    self._sp_table(sp_type, len(pairs[0][0]) * 8)
                    old = self._sp_assets[sp_type].pop(body, None)
                    if old is not None:
                        sp_deleted.setdefault(sp_type, []).extend(ptr for _, ptr in old)
                    if key in self._suspect:       # rows of a half-applied earlier attempt may sit under the NEW pointers
                        sp_deleted.setdefault(sp_type, []).extend(ptr for _, ptr in pairs)
                    self._sp_assets[sp_type][body] = pairs
                    keys_b, vecs_b = sp_batches.setdefault(sp_type, ([], []))
                    for sp_bytes, ptr in pairs:
                        keys_b.append(ptr)
                        vecs_b.append(np.frombuffer(sp_bytes, dtype=np.uint8))

            # device side: remove-before-add for updated assets, then one batched add per table
            try:
                for unit_type, keys_gone in dropped.items():
                    self._unit_tables[unit_type].remove(keys_gone)
                for unit_type, items in unit_batches.items():
                    table = self._unit_table(unit_type)
                    to_remove = [k for k in items if k in updated_keys or k in self._suspect]
                    if to_remove:
                        table.remove(to_remove)
                    table.add(list(items.keys()), list(items.values()))
                for sp_type, (ckeys, vecs) in sp_batches.items():
                    table = self._sp_tables[sp_type]
                    if sp_type in sp_deleted:
                        table.remove(sp_deleted[sp_type])
                    table.add_raw(ckeys, vecs)
            except Exception:
                for key, body, old_entry, old_units, old_sp in undo:
                    if old_entry is None:
                        self._assets.pop(key, None)
                        self._asset_units.pop(key, None)
                    else:
                        self._assets[key] = old_entry
                        self._asset_units[key] = old_units if old_units is not None else {}
                    for sp_type, pairs in old_sp.items():
                        if pairs is None:
                            self._sp_assets.get(sp_type, {}).pop(body, None)
                        else:
                            self._sp_assets[sp_type][body] = pairs
                    self._suspect.add(key)
                raise
            for key, *_ in staged:
                self._suspect.discard(key)
            if staged:
                self.dirty = True
            return results

    def _fingerprint_of(self, sp_type, body):
        pairs = self._sp_assets.get(sp_type, {}).get(body)
        if pairs is None:
            return None
        from iscc_search_amd.simprint import unpack_chunk_pointer

        return tuple(sorted((sp, *unpack_chunk_pointer(ptr)[1:]) for sp, ptr in pairs))

    # -- get ---------------------------------------------------------------------------------------
    def get_asset(self, iscc_id):
        # type: (str) -> IsccEntry
        if self._realm_id is not None:
            codec.validate_iscc_id(iscc_id, expected_realm=self._realm_id)
        key = codec.iscc_id_to_int(iscc_id)
        with self._lock:
            asset = self._assets.get(key)
        if asset is None:
            raise FileNotFoundError(f"Asset '{iscc_id}' not found in index")
        return asset

    # -- search ------------------------------------------------------------------------------------
    def _search_similarity_unit(self, unit_type, body, limit):
        # type: (str, bytes, int) -> Dict[int, float]
        """``usearch/index.py:2024-2045``: score = max(0, 1 - NPHD)."""
        matches = self._unit_tables[unit_type].search(np.frombuffer(body, dtype=np.uint8), count=_checked_limit(limit))
        out = {}
        for key, distance in zip(matches.keys, matches.distances):
            out[int(key)] = max(0.0, 1.0 - float(distance))
        return out

    def _search_instance_unit(self, unit_type, body):
        # type: (str, bytes) -> Dict[int, float]
        """
        Bidirectional prefix match of identity codes, every hit scoring 1.0 (``usearch/index.py:1957-2022``).
        A stored code is a hit iff it agrees with the query on their common prefix, i.e. Hamming distance 0
        over the compared prefix: one range-limited GPU scan answers it.  Capped at MAX_K hits per query.
        """
        table = self._unit_tables.get(unit_type)
        if table is None:
            return {}
        m = table.search_within(np.frombuffer(body, dtype=np.uint8), count=INSTANCE_FIRST_K, max_hamming=0)
        if len(m.keys) == INSTANCE_FIRST_K:      # the list may go on: ask for everything up to the cap
            m = table.search_within(np.frombuffer(body, dtype=np.uint8), count=MAX_K, max_hamming=0)
        _check_instance_hits(len(m.keys), unit_type)
        return {int(key): 1.0 for key in m.keys}

    def _search_units(self, units, limit):
        # type: (List[str], int) -> Dict[int, Dict[str, float]]
        """
        The per-unit searches of one request (``usearch/index.py:786-806``: one ``search`` per similarity unit, one
        prefix match per INSTANCE unit) as ONE engine call with one device synchronisation; merged exactly as
        the reference merges its per-unit dicts (max per (key, unit_type), :806; INSTANCE hits score 1.0, :2010).
        """
        plan, requests = [], []
        for unit_str in units:
            unit = codec.parse(unit_str)
            index = self._unit_tables.get(unit.unit_type)
            if index is None:
                continue
            words, nbytes = pack_bytes([unit.body], index._table.max_words)
            if unit.unit_type.startswith("INSTANCE_"):
                # identity matches are few: a short list first (64 records per query to select, exchange between shards and unpack instead
                # of 4 096); a list that comes back full is asked again up to the cap
                requests.append((index._table, words, nbytes, INSTANCE_FIRST_K, 0))
            else:
                requests.append((index._table, words, nbytes, _checked_limit(limit), None))
            plan.append(unit.unit_type)
        aggregated = {}  # type: Dict[int, Dict[str, float]]
        for unit_type, request, (keys, ham, pbits, cnt) in zip(plan, requests, self._engine.search_many(requests)):
            c = int(cnt[0])
            if unit_type.startswith("INSTANCE_"):
                if c == INSTANCE_FIRST_K:
                    keys, ham, pbits, cnt = self._engine.search_many([request[:3] + (MAX_K, 0)])[0]
                    c = int(cnt[0])
                _check_instance_hits(c, unit_type)
                for key in keys[0, :c].tolist():
                    aggregated.setdefault(key, {})[unit_type] = 1.0
                continue
            # float32 NPHD as HipNphdIndex.search hands it out, then the reference's float64 `1.0 - d` clamp (:2041-2043) -- the same
            # IEEE operations on the whole list at once (a float32 widens to float64 exactly), then plain Python numbers
            dist = ham[0, :c].astype(np.float32) / pbits[0, :c].astype(np.float32)
            scores = np.maximum(0.0, 1.0 - dist.astype(np.float64)).tolist()
            for key, score in zip(keys[0, :c].tolist(), scores):
                slot = aggregated.get(key)
                if slot is None:
                    aggregated[key] = {unit_type: score}
                elif score > slot.get(unit_type, 0.0):          # max per (key, unit_type), :806; scores are >= 0.0
                    slot[unit_type] = score
                else:
                    slot.setdefault(unit_type, 0.0)
        return aggregated

    def search_assets(self, query, limit=100, exact=False):
        # type: (IsccQuery, int, bool) -> IsccSearchResult
        """``exact=True`` matches simprints by collision only (``usearch/index.py:735-778, :1261-1304``)."""
        query_iscc_id = None
        if query.iscc_id:
            query_iscc_id = query.iscc_id
            asset = self.get_asset(query.iscc_id)
            query = IsccQuery(iscc_code=asset.iscc_code, units=asset.units, simprints=None)
        query = normalize_query(query)
        # No index-wide lock here: the engine serialises (and combines) concurrent searches itself, the host
        # dicts are only read, and writers (add_assets) publish whole entries -- as with the reference, a search
        # that overlaps an add may or may not see that batch.
        chunk_matches = []
        if self._sp_tables and query.simprints:
            chunk_matches = self._search_simprints(query, limit, exact=exact)
        matches = []
        if query.units:
            aggregated = self._search_units(query.units, limit)
            scored = []
            thr, exp = self._opts.match_threshold_units, self._opts.confidence_exponent
            for key, unit_scores in aggregated.items():
                confident = {t: s for t, s in unit_scores.items() if s >= thr}
                if not confident:
                    continue
                weight_sum = sum(confident.values())
                total = sum(s**exp for s in confident.values()) / weight_sum if weight_sum > 0 else 0.0
                scored.append((key, total, unit_scores))
            if query_iscc_id:
                qkey = codec.iscc_id_to_int(query_iscc_id)
                scored = [r for r in scored if r[0] != qkey]
            scored.sort(key=lambda r: r[1], reverse=True)   # stable, as the reference (:836)
            for key, total, unit_scores in scored[:limit]:
                asset = self._assets.get(key)
                source = metadata = None
                if asset is not None and asset.metadata:
                    source = asset.metadata.get("source")
                    metadata = asset.metadata
                matches.append(IsccGlobalMatch(
                    iscc_id=codec.iscc_id_from_int(key, self._realm_id or 0), score=min(1.0, total),
                    types=unit_scores, source=source, metadata=metadata,
                ))
        if query_iscc_id:
            chunk_matches = [m for m in chunk_matches if m.iscc_id != query_iscc_id]
        return IsccSearchResult(query=query, global_matches=matches, chunk_matches=chunk_matches)

    def _search_simprints(self, query, limit, exact=False):
        # type: (IsccQuery, int, bool) -> List[IsccChunkMatch]
        """
        Per-type search, mean over types, order (-score, iscc_id): approximate-mode scoring of
        ``usearch/index.py:1357-1469``, or with ``exact`` the hard-boundary collision search of ``:1261-1355``.
        """
        total_assets = len(self._assets)
        per_asset = {}  # type: Dict[bytes, Dict[str, object]]
        for sp_type, simprint_objs in query.simprints.items():
            table = self._sp_tables.get(sp_type)
            if table is None:
                continue
            q_bytes = [codec.decode_base64(_sp_string(s)) for s in simprint_objs]
            if exact:
                raw = table.search_exact(simprints=q_bytes, limit=limit * 2, threshold=self._opts.match_threshold_simprints, detailed=True)
            else:
                raw = table.search_raw(
                    simprints=q_bytes, limit=limit * 2, threshold=self._opts.match_threshold_simprints, detailed=True,
                    total_assets=total_assets, device_doc_freq=True,   # lmdb_ops.count_doc_freq, on the device
                )
            for r in raw:
                per_asset.setdefault(r.iscc_id_body, {})[sp_type] = r
        if not per_asset:
            return []
        ranked = []
        for body, type_results in per_asset.items():
            score = sum(r.score for r in type_results.values()) / len(type_results)
            digest = codec.decode_base32(codec.iscc_id_from_int(int.from_bytes(body, "big"), self._realm_id or 0)[5:])
            ranked.append((score, digest, body, type_results))
        ranked.sort(key=lambda x: (-x[0], x[1]))
        out = []
        for score, digest, body, type_results in ranked[:limit]:
            asset = self._assets.get(int.from_bytes(body, "big"))
            source = metadata = None
            if asset is not None and asset.metadata:
                source = asset.metadata.get("source")
                metadata = asset.metadata
            types = {}
            for sp_type, r in type_results.items():
                chunks = None
                if r.chunks is not None:
                    chunks = [
                        IsccMatchedChunk(query=codec.encode_base64(c.query), match=codec.encode_base64(c.match),
                                         score=c.score, freq=max(1, c.freq), offset=c.offset, size=c.size, content=None)
                        for c in r.chunks
                    ]
                types[sp_type] = Types(score=r.score, matches=r.matches, queried=r.queried, chunks=chunks)
            out.append(IsccChunkMatch(iscc_id="ISCC:" + codec.encode_base32(digest), score=score, types=types, source=source, metadata=metadata))
        return out

    # -- snapshot ------------------------------------------------------------------------------------
    # The reference persists through LMDB + HNSW shard files and `flush()` / `close()`
    # (usearch/index.py:883-967).  Here a snapshot is: index.json, assets.jsonl, and the raw code columns
    # of every table (units/<type>/, simprints/<type>/) exactly as they sit in HBM.
    def save(self, path):
        # type: (str) -> None
        with self._lock:
            os.makedirs(path, exist_ok=True)
            # sharded index (hip:///path?devices=N): every rank writes its own table shards, rank 0 the host files
            writer = getattr(self._engine, "rank", 0) == 0
            if writer:
                with open(os.path.join(path, "assets.jsonl.tmp"), "w") as f:
                    for key, asset in self._assets.items():
                        f.write(json.dumps({"key": key, "asset": asset.model_dump(mode="json", exclude_none=True)}, separators=(",", ":")) + "\n")
                os.replace(os.path.join(path, "assets.jsonl.tmp"), os.path.join(path, "assets.jsonl"))
            for unit_type, table in self._unit_tables.items():
                table.save(os.path.join(path, "units", unit_type))
            for sp_type, table in self._sp_tables.items():
                table.save(os.path.join(path, "simprints", sp_type))
                if writer:
                    # the per-asset chunk lists are host state every rank keeps whole: written once, by rank 0, beside the table
                    # shards, so that a restore reads a file instead of gathering every shard's rows to every rank
                    pairs = [pair for chunks in self._sp_assets.get(sp_type, {}).values() for pair in chunks]
                    nb = table.ndim // 8
                    ptrs = np.frombuffer(b"".join(ptr for _, ptr in pairs), dtype=np.uint8).reshape(len(pairs), 16)
                    sps = np.frombuffer(b"".join(sp for sp, _ in pairs), dtype=np.uint8).reshape(len(pairs), nb)
                    tmp = os.path.join(path, "simprints", sp_type, "host_chunks.tmp.npz")
                    np.savez(tmp, pointers=ptrs, simprints=sps)
                    os.replace(tmp, os.path.join(path, "simprints", sp_type, "host_chunks.npz"))
            if writer:
                meta = {
                    "format": 1, "realm_id": self._realm_id, "assets": len(self._assets),
                    "unit_types": sorted(self._unit_tables),
                    "simprint_types": {t: tbl.ndim for t, tbl in self._sp_tables.items()},
                    "ranks": getattr(self._engine, "world_size", 1),
                }
                with open(os.path.join(path, "index.json.tmp"), "w") as f:
                    json.dump(meta, f)
                os.replace(os.path.join(path, "index.json.tmp"), os.path.join(path, "index.json"))
            if hasattr(self._engine, "all_gather_object"):
                self._engine.all_gather_object(None)       # nobody returns before rank 0 has written index.json
            self.dirty = False

    @classmethod
    def load(cls, engine, path, options=None):
        # type: (object, str, HipOptions | None) -> HipIndex
        from iscc_search_amd.simprint import unpack_chunk_pointer

        with open(os.path.join(path, "index.json")) as f:
            meta = json.load(f)
        if meta.get("ranks", 1) != getattr(engine, "world_size", 1):
            raise ValueError(f"snapshot at {path} was written by {meta.get('ranks', 1)} rank(s), this manager runs {getattr(engine, 'world_size', 1)}")
        idx = cls(engine, options)
        idx._realm_id = meta["realm_id"]
        assets_file = os.path.join(path, "assets.jsonl")
        if os.path.exists(assets_file):
            with open(assets_file) as f:
                for line in f:
                    rec = json.loads(line)
                    asset = IsccEntry(**rec["asset"])
                    idx._assets[rec["key"]] = asset
                    units = {}
                    for unit_str in asset.units or []:
                        u = codec.Iscc(unit_str)
                        units[u.unit_type] = u.body
                    idx._asset_units[rec["key"]] = units
        for unit_type in meta["unit_types"]:
            idx._unit_table(unit_type).load(os.path.join(path, "units", unit_type))
        for sp_type, ndim in meta["simprint_types"].items():
            table = idx._sp_table(sp_type, ndim)
            table.load(os.path.join(path, "simprints", sp_type))
            # the host-side per-asset chunk lists (host state every rank keeps whole): from the file rank 0 wrote; snapshots of
            # before that file existed derive them from the stored rows (of every shard, gathered)
            host_chunks = os.path.join(path, "simprints", sp_type, "host_chunks.npz")
            if os.path.exists(host_chunks):
                with np.load(host_chunks) as z:
                    rows = sorted((z["pointers"][i].tobytes(), z["simprints"][i].tobytes()) for i in range(len(z["pointers"])))
            else:
                rows = list(table.rows())
                if hasattr(engine, "all_gather_object"):
                    rows = sorted(r for part in engine.all_gather_object(rows) for r in part)
            for ckey, sp_bytes in rows:
                body = unpack_chunk_pointer(ckey)[0]
                idx._sp_assets[sp_type].setdefault(body, []).append((sp_bytes, ckey))
        return idx

    def close(self):
        # type: () -> None
        with self._lock:
            for t in list(self._unit_tables.values()):
                t.close()
            for t in list(self._sp_tables.values()):
                t.close()
            self._unit_tables.clear()
            self._sp_tables.clear()


class HipIndexManager:
    """
    Protocol-conformant manager of named ``hip:///`` indexes on one GPU.

    The engine (library load, GPU context, stream) is created lazily on first use and released by
    ``close()``, which is idempotent (``protocols/index.py:162-172``).  Thread-safe: FastAPI calls
    the sync protocol methods from a thread pool (``docs/explanation/architecture.md:120-126``).

    ``engine`` is a seam for sharing one ``HipEngine`` between managers and for the CPU test tier (which injects
    an oracle-backed stand-in from ``tests/``); left ``None`` -- as ``get_index("hip:///")`` leaves it -- the manager
    creates a ``HipEngine`` and raises if the HIP library or the GPU is missing.  There is no CPU fallback.
    """

    def __init__(self, uri="hip:///", engine=None, options=None, shard_engine_factory=None):
        # type: (str, object | None, HipOptions | None, str | None) -> None
        parsed = urlparse(uri)
        if parsed.scheme != "hip":
            raise ValueError(f"HipIndexManager requires a hip:// URI, got '{uri}'")
        qs = parse_qs(parsed.query)
        self.device_id = int(qs.get("device", ["0"])[0])
        # hip:///?devices=N  -> the index row-sharded over N GPUs, one process per GPU (iscc_search_amd/sharded_engine.py).
        #   * constructed where NO torch.distributed group exists (the reference's server / CLI: one process): this process
        #     becomes the leader, starts the N - 1 other ranks itself and broadcasts every call to them (shard_front.py);
        #     `backend=gloo` / `same_gpu=1` in the query select the rehearsal transport (CPU tests, ranks sharing one GPU)
        #   * constructed on every rank of a running group of N (torch.distributed.run): SPMD, every rank makes the same calls
        self.devices = int(qs.get("devices", ["1"])[0])
        if self.devices < 1:
            raise ValueError(f"devices must be >= 1, got {self.devices}")
        self._uri = uri
        self._shard_backend = qs.get("backend", ["nccl"])[0]
        self._shard_same_gpu = qs.get("same_gpu", ["0"])[0] == "1"
        self._shard_engine_factory = shard_engine_factory
        self._leading = False          # this manager drives a leader front (shard_front.LeaderEngine) as its engine
        # hip:///            -> volatile (like memory://)
        # hip:///abs/path    -> snapshots under that directory: loaded lazily, written by flush()/close()
        self.base_path = parsed.path if parsed.path not in ("", "/") else None
        self._engine = engine
        self._owns_engine = engine is None
        self._opts = options or HipOptions()
        self._indexes = {}  # type: Dict[str, HipIndex]
        self._on_disk = set()
        self._lock = threading.RLock()
        # sharded mode is SPMD: every rank must issue the same calls in the same order (each holds collectives), so calls that
        # reach the engine are serialised per manager; a single-GPU manager lets them run concurrently (the engine combines them)
        self._spmd = threading.RLock() if self.devices > 1 else None
        self._closed = False
        if self.base_path:
            os.makedirs(self.base_path, exist_ok=True)
            for name in sorted(os.listdir(self.base_path)):
                if INDEX_NAME_RE.match(name) and os.path.exists(os.path.join(self.base_path, name, "index.json")):
                    self._on_disk.add(name)

    def _leads(self):
        """
        Whether this manager is the ONE calling process of a sharded index (``shard_front.LeaderEngine``: it starts the other
        ranks itself) -- as opposed to unsharded, an injected engine, or one rank of a launcher's SPMD group.
        """
        if self.devices == 1 or (self._engine is not None and not self._leading):
            return False
        if self._leading:
            return True
        import torch.distributed as dist

        from iscc_search_amd import shard_front

        # a process group that exists and is NOT a leader front's belongs to a launcher: SPMD
        return not dist.is_initialized() or shard_front.leads_this_process()

    @property
    def _leader(self):
        """The leader front this manager drives (None before its first use and in every other mode)."""
        return self._engine if self._leading else None

    def _get_engine(self):
        if self._engine is None:
            from iscc_search_amd.engine import HipEngine   # raises loudly without library / GPU

            if self.devices > 1 and self._leads():
                from iscc_search_amd.shard_front import leader_engine

                with self._lock:
                    if self._engine is None:
                        self._engine = leader_engine(self.devices, backend=self._shard_backend, engine_factory=self._shard_engine_factory,
                                                     same_gpu=self._shard_same_gpu)
                        self._leading = True
            elif self.devices > 1:
                import torch.distributed as dist

                from iscc_search_amd.sharded_engine import ShardedEngine

                if not dist.is_initialized() or dist.get_world_size() != self.devices:
                    raise ValueError(
                        f"hip:///?devices={self.devices}: this rank's engine needs the process group of {self.devices} ranks (one per GPU); "
                        f"a single calling process gets it from shard_front.LeaderEngine, a launcher from torch.distributed.run"
                    )
                local = int(os.environ.get("LOCAL_RANK", dist.get_rank()))
                self._engine = ShardedEngine(HipEngine(local), device=f"cuda:{local}")
            else:
                self._engine = HipEngine(self.device_id)
        return self._engine

    def _index(self, name):
        # type: (str) -> HipIndex
        idx = self._indexes.get(name)
        if idx is None and name in self._on_disk:
            idx = self._indexes[name] = HipIndex.load(self._get_engine(), os.path.join(self.base_path, name), self._opts)
        if idx is None:
            raise FileNotFoundError(f"Index '{name}' not found")
        return idx

    def _names(self):
        return sorted(set(self._indexes) | self._on_disk)

    def _asset_count(self, name):
        if name in self._indexes:
            return len(self._indexes[name])
        with open(os.path.join(self.base_path, name, "index.json")) as f:
            return json.load(f).get("assets", 0)

    # Every public method runs `_guarded`:
    #   * a sharded index opened by ONE process: the host logic runs here, alone, over the leader front's engine, which broadcasts
    #     every TABLE operation to the shard workers in its own order and combines the searches of concurrent callers
    #     (shard_front.py) -- no manager-wide lock, as on one GPU; a front that is down fails every call, also those that need no table;
    #   * a sharded index under a launcher (SPMD) takes `_spmd` around the WHOLE method: flush, close, the lazy snapshot load in
    #     `_index` and delete_index reach collectives just as add / search do;
    #   * one GPU: no extra lock, searches of concurrent callers are combined by the engine.
    def _guarded(self, method, impl, *args, **kwargs):
        if self._leads():
            self._get_engine().check()
            return impl(*args, **kwargs)
        if self._spmd is not None:
            with self._spmd:
                return impl(*args, **kwargs)
        return impl(*args, **kwargs)

    def flush(self):
        # type: () -> None
        """Write every modified index to its snapshot directory (no-op for volatile managers)."""
        return self._guarded("flush", self._flush)

    def _flush(self):
        if not self.base_path:
            return
        with self._lock:
            for name, idx in self._indexes.items():
                if idx.dirty or name not in self._on_disk:
                    idx.save(os.path.join(self.base_path, name))
                    self._on_disk.add(name)

    # -- protocol ----------------------------------------------------------------------------------
    def list_indexes(self):
        # type: () -> List[IsccIndex]
        return self._guarded("list_indexes", self._list_indexes)

    def _list_indexes(self):
        with self._lock:
            return [IsccIndex(name=n, assets=self._asset_count(n), size=0) for n in self._names()]

    def create_index(self, index):
        # type: (IsccIndex) -> IsccIndex
        return self._guarded("create_index", self._create_index, index)

    def _create_index(self, index):
        validate_index_name(index.name)
        with self._lock:
            if index.name in self._indexes or index.name in self._on_disk:
                raise FileExistsError(f"Index '{index.name}' already exists")
            self._indexes[index.name] = HipIndex(self._get_engine(), self._opts)
        return IsccIndex(name=index.name, assets=0, size=0)

    def get_index(self, name):
        # type: (str) -> IsccIndex
        return self._guarded("get_index", self._get_index, name)

    def _get_index(self, name):
        with self._lock:
            if name not in self._indexes and name not in self._on_disk:
                raise FileNotFoundError(f"Index '{name}' not found")
            return IsccIndex(name=name, assets=self._asset_count(name), size=0)

    def delete_index(self, name):
        # type: (str) -> None
        return self._guarded("delete_index", self._delete_index, name)

    def _delete_index(self, name):
        with self._lock:
            if name not in self._indexes and name not in self._on_disk:
                raise FileNotFoundError(f"Index '{name}' not found")
            if name in self._indexes:
                self._indexes.pop(name).close()
            if name in self._on_disk:
                if getattr(self._engine, "rank", 0) == 0:          # one snapshot directory, shared by the ranks of a node
                    shutil.rmtree(os.path.join(self.base_path, name), ignore_errors=True)
                self._on_disk.discard(name)

    def add_assets(self, index_name, assets):
        # type: (str, List[IsccEntry]) -> List[IsccAddResult]
        return self._guarded("add_assets", self._add_assets, index_name, assets)

    def _add_assets(self, index_name, assets):
        with self._lock:
            idx = self._index(index_name)
        return idx.add_assets(assets)

    def get_asset(self, index_name, iscc_id):
        # type: (str, str) -> IsccEntry
        return self._guarded("get_asset", self._get_asset, index_name, iscc_id)

    def _get_asset(self, index_name, iscc_id):
        with self._lock:
            idx = self._index(index_name)
        try:
            return idx.get_asset(iscc_id)
        except FileNotFoundError:
            raise FileNotFoundError(f"Asset '{iscc_id}' not found in index '{index_name}'")

    def search_assets(self, index_name, query, limit=100):
        # type: (str, IsccQuery, int) -> IsccSearchResult
        return self._guarded("search_assets", self._search_assets, index_name, query, limit)

    def _search_assets(self, index_name, query, limit=100):
        with self._lock:
            idx = self._index(index_name)
        try:
            return idx.search_assets(query, limit)
        except FileNotFoundError:
            raise FileNotFoundError(f"Asset '{query.iscc_id}' not found in index '{index_name}'")

    def close(self):
        # type: () -> None
        if self._leading:
            if self._engine is not None and self._engine.broken is not None:
                # nothing can be flushed through a front that is down: release what is left of it
                with self._lock:
                    if not self._closed:
                        self._closed = True
                        self._indexes.clear()
                        self._engine.close()
                        self._engine = None
                return
            return self._close()
        if self._spmd is not None and not self._leads():
            with self._spmd:
                return self._close()
        return self._close()

    def _close(self):
        with self._lock:
            if self._closed:
                return
            self._closed = True
            self._flush()
            for idx in self._indexes.values():
                idx.close()
            self._indexes.clear()
            if self._owns_engine and self._engine is not None:
                self._engine.close()
                self._engine = None


def get_index(uri="hip:///", **kwargs):
    # type: (str, object) -> HipIndexManager
    """
    Factory for the ``hip`` scheme, the branch a maintainer adds to the reference's
    ``options.get_index()`` next to ``iscc_search/options.py:360-371``.
    """
    parsed = urlparse(uri)
    if not parsed.scheme:
        raise ValueError(f"index URI requires an explicit scheme, got '{uri}'")
    if parsed.scheme != "hip":
        raise ValueError(f"Unsupported index URI scheme: '{parsed.scheme}' (this package serves hip://)")
    return HipIndexManager(uri, **kwargs)
