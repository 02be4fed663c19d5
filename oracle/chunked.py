"""ctypes front end of oracle/chunked_attention.c (ORACLE - test infrastructure only).

`build()` compiles the C restatement with gcc into oracle/_build/libfa_oracle.so;
`chunked_attention(qkv, nq, nk)` runs the chunked online-softmax program the reference builds
(FA.cpp:175-334, incl. its cyclic causal-mask blocks FA.cpp:47-173) on the CPU in fp32 or fp64.
"""

import ctypes
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_SRC = _HERE / "chunked_attention.c"
_LIB = _HERE / "_build" / "libfa_oracle.so"
_lib = None


def build(force: bool = False) -> pathlib.Path:
    if force or not _LIB.exists() or _LIB.stat().st_mtime < _SRC.stat().st_mtime:
        _LIB.parent.mkdir(exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-o", str(_LIB), str(_SRC), "-lm"]
        )
    return _LIB


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(build()))
        for name, ct in (("fa_oracle_chunked_f32", ctypes.c_float), ("fa_oracle_chunked_f64", ctypes.c_double)):
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.POINTER(ct), ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                           ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ct), ctypes.POINTER(ct)]
        _lib.fa_oracle_triu_offsets.restype = ctypes.c_int
        _lib.fa_oracle_triu_offsets.argtypes = [ctypes.c_uint32, ctypes.c_uint32,
                                                ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
    return _lib


def triu_offsets(num_rows: int, num_cols: int):
    """getTriuOffsetSequence (FA.cpp:47-126). Raises for the shapes on which the reference hangs."""
    lib = _load()
    cap = num_rows + num_cols + 4
    buf = (ctypes.c_int32 * cap)()
    n = lib.fa_oracle_triu_offsets(num_rows, num_cols, buf, cap)
    if n < 0:
        raise RuntimeError("reference getTriuOffsetSequence does not terminate for this block shape")
    return list(buf[:n])


def chunked_attention(qkv: np.ndarray, num_chunks_q: int, num_chunks_kv: int):
    """qkv: [3,G,L,D] float32/float64 ndarray -> (out [G,L,D], lse [G,L]) in the same dtype."""
    lib = _load()
    qkv = np.ascontiguousarray(qkv)
    assert qkv.ndim == 4 and qkv.shape[0] == 3
    _, G, L, D = qkv.shape
    if qkv.dtype == np.float32:
        fn, ct = lib.fa_oracle_chunked_f32, ctypes.c_float
    elif qkv.dtype == np.float64:
        fn, ct = lib.fa_oracle_chunked_f64, ctypes.c_double
    else:
        raise TypeError("chunked oracle runs in float32 or float64")
    out = np.empty((G, L, D), dtype=qkv.dtype)
    lse = np.empty((G, L), dtype=qkv.dtype)
    rc = fn(qkv.ctypes.data_as(ctypes.POINTER(ct)), G, L, D, num_chunks_q, num_chunks_kv,
            out.ctypes.data_as(ctypes.POINTER(ct)), lse.ctypes.data_as(ctypes.POINTER(ct)))
    if rc == -1:
        raise ValueError("sequence length must be divisible by both chunk counts (FA.cpp:190-191)")
    if rc == -2:
        raise RuntimeError("kv chunk length 1: the reference's mask-offset loop never terminates (FA.cpp:106-124)")
    return out, lse
