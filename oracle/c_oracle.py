"""ctypes front-end of oracle/thr_oracle.c  --  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libthr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "thr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def doc_norms(docs: np.ndarray) -> np.ndarray:
    docs = np.ascontiguousarray(docs, dtype=np.float32)
    out = np.empty(docs.shape[0], dtype=np.float64)
    lib().oracle_doc_norms_f64(_p(docs, C.c_float), C.c_int64(docs.shape[0]),
                               C.c_int(docs.shape[1]), _p(out, C.c_double))
    return out


def dense_topk_exact(docs, queries, k, doc_id_base=0, dnorm=None):
    docs = np.ascontiguousarray(docs, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    if dnorm is None:
        dnorm = doc_norms(docs)
    dnorm = np.ascontiguousarray(dnorm, dtype=np.float64)
    nq = queries.shape[0]
    S = np.full((nq, k), -np.inf, dtype=np.float64)
    I = np.full((nq, k), -1, dtype=np.int64)
    cnt = np.zeros(nq, dtype=np.int32)
    lib().oracle_dense_topk_exact(_p(docs, C.c_float), C.c_int64(docs.shape[0]),
                                  C.c_int(docs.shape[1]), _p(dnorm, C.c_double),
                                  _p(queries, C.c_float), C.c_int(nq), C.c_int(k),
                                  C.c_int64(doc_id_base), _p(S, C.c_double), _p(I, C.c_int64),
                                  _p(cnt, C.c_int32))
    return S, I, cnt


def bm25_scores(rowptr, post_doc, post_tf, doclen, idf, avgdl, terms, n_docs, k1=1.2, b=0.75):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    post_doc = np.ascontiguousarray(post_doc, dtype=np.int32)
    post_tf = np.ascontiguousarray(post_tf, dtype=np.int32)
    doclen = np.ascontiguousarray(doclen, dtype=np.float32)
    idf = np.ascontiguousarray(idf, dtype=np.float64)
    terms = np.ascontiguousarray(terms, dtype=np.int32)
    out = np.empty(n_docs, dtype=np.float64)
    lib().oracle_bm25_scores(_p(rowptr, C.c_int64), _p(post_doc, C.c_int32), _p(post_tf, C.c_int32),
                             _p(doclen, C.c_float), _p(idf, C.c_double), C.c_double(avgdl),
                             _p(terms, C.c_int32), C.c_int(len(terms)), C.c_int64(n_docs),
                             C.c_double(k1), C.c_double(b), _p(out, C.c_double))
    return out


def maxsim(qtok: np.ndarray, dtok: np.ndarray, cand: np.ndarray) -> np.ndarray:
    qtok = np.ascontiguousarray(qtok, dtype=np.float16)
    dtok = np.ascontiguousarray(dtok, dtype=np.float16)
    cand = np.ascontiguousarray(cand, dtype=np.int32)
    out = np.empty(cand.shape, dtype=np.float64)
    lib().oracle_maxsim(_p(qtok.view(np.uint16), C.c_uint16), C.c_int(qtok.shape[0]),
                        C.c_int(qtok.shape[1]), C.c_int(qtok.shape[2]),
                        _p(dtok.view(np.uint16), C.c_uint16), C.c_int(dtok.shape[1]),
                        _p(cand, C.c_int32), C.c_int(cand.shape[1]), _p(out, C.c_double))
    return out
