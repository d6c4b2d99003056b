"""
Wire models of the ``IsccIndexProtocol`` boundary (field names and shapes of the reference's
``iscc_search/schema.py``, which is generated from its OpenAPI spec and cannot be imported here).

Only the models the protocol exchanges are mirrored: ``IsccIndex`` (:18-41), ``IsccSimprint`` (:45-72),
``IsccQuery`` (:95-139), ``Status``/``IsccAddResult`` (:142-162), ``IsccMatchedChunk`` (:188-249),
``IsccEntry`` (:329-381), ``IsccGlobalMatch`` (:384-415), ``Types`` (:418-442), ``IsccChunkMatch``
(:445-530), ``IsccSearchResult`` (:533-558).  ``HipIndexManager`` duck-types on attributes, so the
reference's own pydantic objects can be passed in when both packages are installed side by side.
"""

from enum import Enum
from typing import Any, Dict, List, Optional, Union

from pydantic import BaseModel, ConfigDict, Field

ISCC_ID_PATTERN = r"^ISCC:[A-Z2-7]{16}$"
ISCC_CODE_PATTERN = r"^ISCC:[A-Z2-7]{16,}$"
SIMPRINT_PATTERN = r"^[A-Za-z0-9+/_=-]+$"
INDEX_NAME_PATTERN = r"^[a-z][a-z0-9]*$"


class IsccIndex(BaseModel):
    name: str = Field(min_length=1, max_length=32, pattern=INDEX_NAME_PATTERN)
    assets: Optional[int] = Field(default=None, ge=0)
    size: Optional[int] = Field(default=None, ge=0)
    sizes: Optional[Dict[str, int]] = None


class IsccSimprint(BaseModel):
    simprint: str = Field(min_length=11, pattern=SIMPRINT_PATTERN)
    offset: int = Field(ge=0, le=4294967295)
    size: int = Field(ge=0, le=4294967295)


class IsccQuery(BaseModel):
    iscc_id: Optional[str] = Field(default=None, pattern=ISCC_ID_PATTERN)
    iscc_code: Optional[str] = Field(default=None, pattern=ISCC_CODE_PATTERN)
    units: Optional[List[str]] = Field(default=None, min_length=1)
    simprints: Optional[Dict[str, List[str]]] = None


class Status(str, Enum):
    created = "created"
    updated = "updated"


class IsccAddResult(BaseModel):
    iscc_id: str = Field(pattern=ISCC_ID_PATTERN)
    status: Status


class IsccMetadata(BaseModel):
    model_config = ConfigDict(extra="allow")
    name: Optional[str] = None
    source: Optional[str] = None


class IsccMatchedChunk(BaseModel):
    query: str = Field(pattern=SIMPRINT_PATTERN)
    match: str = Field(pattern=SIMPRINT_PATTERN)
    score: float = Field(ge=0.0, le=1.0)
    freq: int = Field(ge=1)
    offset: int = Field(ge=0, le=4294967295)
    size: int = Field(ge=0, le=4294967295)
    content: Optional[str] = None


class IsccEntry(BaseModel):
    iscc_id: Optional[str] = Field(default=None, pattern=ISCC_ID_PATTERN)
    iscc_code: Optional[str] = Field(default=None, pattern=ISCC_CODE_PATTERN)
    units: Optional[List[str]] = Field(default=None, min_length=2)
    simprints: Optional[Dict[str, List[IsccSimprint]]] = None
    metadata: Optional[Dict[str, Any]] = None


class IsccGlobalMatch(BaseModel):
    iscc_id: str = Field(pattern=ISCC_ID_PATTERN)
    score: float = Field(ge=0.0, le=1.0)
    types: Dict[str, float] = Field(min_length=1)
    source: Optional[str] = None
    metadata: Optional[Union[IsccMetadata, Dict[str, Any]]] = None


class Types(BaseModel):
    score: float = Field(ge=0.0, le=1.0)
    matches: int = Field(ge=0)
    queried: int = Field(ge=1)
    chunks: Optional[List[IsccMatchedChunk]] = None


class IsccChunkMatch(BaseModel):
    iscc_id: str = Field(pattern=ISCC_ID_PATTERN)
    score: float = Field(ge=0.0, le=1.0)
    types: Dict[str, Types] = Field(min_length=1)
    source: Optional[str] = None
    metadata: Optional[Union[IsccMetadata, Dict[str, Any]]] = None


class IsccSearchResult(BaseModel):
    query: IsccQuery
    global_matches: List[IsccGlobalMatch] = Field(default_factory=list)
    chunk_matches: List[IsccChunkMatch] = Field(default_factory=list)
