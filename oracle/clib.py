"""
ctypes binding of ``oracle/liboracle.so`` (test infrastructure only -- see ``oracle/__init__.py``).
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load_oracle(rebuild=False):
    # type: (bool) -> ctypes.CDLL
    """Load (building first when missing) the C oracle."""
    global _LIB
    if _LIB is not None and not rebuild:
        return _LIB
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "nphd_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if rebuild or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    try:
        lib = ctypes.CDLL(so)
    except OSError:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
        lib = ctypes.CDLL(so)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    u16p = ctypes.POINTER(ctypes.c_uint16)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    lib.oracle_topk.restype = ctypes.c_int
    lib.oracle_topk.argtypes = [
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
        ctypes.c_uint64, u64p, u64p, u8p,
        ctypes.c_uint32, u64p, u8p, ctypes.c_uint32,
        u64p, u32p, u16p, u32p, ctypes.c_int,
    ]
    lib.oracle_fill_splitmix64.restype = None
    lib.oracle_fill_splitmix64.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
    lib.oracle_splitmix64.restype = ctypes.c_uint64
    lib.oracle_splitmix64.argtypes = [ctypes.c_uint64]
    lib.oracle_num_threads.restype = ctypes.c_int
    _LIB = lib
    return lib


def _ptr(arr, ctype):
    if arr is None:
        return None
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


def oracle_topk(metric, keys, code_words, nbytes, q_words, q_nbytes, k, fixed_nbytes=0, threads=0):
    # type: (int, np.ndarray, np.ndarray, np.ndarray | None, np.ndarray, np.ndarray | None, int, int, int) -> tuple
    """
    Exact top-k under ascending (distance, key).

    :param metric: 0 = fixed-length Hamming, 1 = NPHD
    :param keys: uint64 [n] or [n, 2] (128-bit keys as hi, lo)
    :param code_words: uint64 [n, max_words] big-endian packed codes
    :param nbytes: uint8 [n] code lengths in bytes, or None for fixed length
    :param q_words: uint64 [nq, max_words]
    :param q_nbytes: uint8 [nq] or None
    :param k: neighbours per query
    :param fixed_nbytes: byte length of a fixed-length table (0 = max_words * 8)
    :return: (keys [nq, k(,2)], hamming [nq, k], prefix_bits [nq, k], count [nq])
    """
    lib = load_oracle()
    code_words = np.ascontiguousarray(code_words, dtype=np.uint64)
    q_words = np.ascontiguousarray(q_words, dtype=np.uint64)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = code_words.shape[0] if code_words.ndim == 2 else 0
    max_words = q_words.shape[1]
    if n:
        assert code_words.shape[1] == max_words
    nq = q_words.shape[0]
    key_words = 2 if keys.ndim == 2 else 1
    if nbytes is not None:
        nbytes = np.ascontiguousarray(nbytes, dtype=np.uint8)
    if q_nbytes is not None:
        q_nbytes = np.ascontiguousarray(q_nbytes, dtype=np.uint8)
    shape = (nq, k, 2) if key_words == 2 else (nq, k)
    out_keys = np.zeros(shape, dtype=np.uint64)
    out_h = np.zeros((nq, k), dtype=np.uint32)
    out_p = np.zeros((nq, k), dtype=np.uint16)
    out_c = np.zeros(nq, dtype=np.uint32)
    rc = lib.oracle_topk(
        metric, key_words, max_words, fixed_nbytes, n,
        _ptr(keys, ctypes.c_uint64), _ptr(code_words, ctypes.c_uint64), _ptr(nbytes, ctypes.c_uint8),
        nq, _ptr(q_words, ctypes.c_uint64), _ptr(q_nbytes, ctypes.c_uint8), k,
        _ptr(out_keys, ctypes.c_uint64), _ptr(out_h, ctypes.c_uint32), _ptr(out_p, ctypes.c_uint16),
        _ptr(out_c, ctypes.c_uint32), threads,
    )
    if rc != 0:
        raise ValueError("oracle_topk: bad arguments")
    return out_keys, out_h, out_p, out_c


def oracle_splitmix64_fill(n, seed, first=0, stride=1, lane=0):
    # type: (int, int, int, int, int) -> np.ndarray
    """out[i] = splitmix64(seed + stride*(first+i) + lane) -- SURVEY.md section 8d generator."""
    lib = load_oracle()
    out = np.empty(n, dtype=np.uint64)
    lib.oracle_fill_splitmix64(_ptr(out, ctypes.c_uint64), n, seed & (2**64 - 1), first, stride, lane)
    return out


def oracle_num_threads():
    # type: () -> int
    return load_oracle().oracle_num_threads()
