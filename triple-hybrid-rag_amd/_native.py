"""ctypes binding of libthr_hip.so (include/thr_hip.h) over PyTorch-ROCm tensors.

PyTorch is used for device memory, streams and torch.distributed only; every
scorer is a hand-written HIP kernel behind the C ABI.  There is NO CPU or
eager-PyTorch fallback: if the library is missing or a call fails, this module
raises.  Operand shapes/dtypes/devices are validated here, on the host, before
any kernel is launched.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _build

THR_FLAG_CERTIFIED = 1
THR_FLAG_OVERFLOW = 2
THR_FLAG_EXACT = 4
THR_DENSE_MAX_K = 256
THR_BM25_MAX_TERMS = 32
THR_GRAPH_MAX_SEEDS = 16
THR_RRF_MAX_PER_CHANNEL = 128
THR_TOPK_MAX = 128
ABI_VERSION = 9

_lib = None


class NativeError(RuntimeError):
    pass


_i64, _i32, _dbl, _sz, _vp = C.c_int64, C.c_int, C.c_double, C.c_size_t, C.c_void_p
_SIGNATURES = {
    "thr_abi_version": (C.c_int, []),
    "thr_error_string": (C.c_char_p, [_i32]),
    "thr_device_info": (_i32, [C.POINTER(_i32), C.POINTER(_i64), C.c_char_p, _i32]),
    "thr_embed_postproc": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "thr_doc_norms": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "thr_dense_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "thr_dense_topk": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp, _sz, _vp]),
    "thr_dense_exact_workspace_bytes": (_sz, [_i64, _i32]),
    "thr_dense_topk_exact": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _sz, _vp]),
    "thr_dense_scan_probe": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _sz, _vp]),
    "thr_dense_rescue_workspace_bytes": (_sz, [_i32, _i32]),
    "thr_dense_rescue": (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _sz, _vp]),
    "thr_dense_quantize_f16": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "thr_dense_f16_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "thr_dense_f16_copy_bytes": (_sz, [_i64, _i32]),
    "thr_dense_f16_query_tile": (_i32, [_i32, _i32, _i32]),
    "thr_dense_f16_max_queries": (_i32, [_i32, _i32]),
    "thr_dense_topk_f16": (_i32, [_vp, _vp, _dbl, _vp, _vp, _i64, _i32, _i64, _vp, _i32, _i32,
                                  _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "thr_dense_shortlist_f16": (_i32, [_vp, _vp, _dbl, _vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp,
                                       _vp, _sz, _vp]),
    "thr_dense_floor": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "thr_dense_finish_f16": (_i32, [_vp, _vp, _dbl, _vp, _vp, _i64, _i32, _i64, _vp, _i32, _i32,
                                    _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "thr_dense_scan_probe_f16": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _sz, _vp]),
    "thr_dense_scan_stamps_f16": (_i32, [_vp, _i64, _i32, _i32, _vp, _sz, _vp, C.POINTER(_i32), _vp]),
    "thr_lexical_build_workspace_bytes": (_sz, [_i64, _i64]),
    "thr_lexical_build": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "thr_bm25_block_count": (_sz, [_i64]),
    "thr_bm25_bounds": (_i32, [_vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _i64, _i64, _vp, _vp, _vp, _vp]),
    "thr_bm25_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "thr_bm25_dense_stride": (_i64, [_i64]),
    "thr_bm25_dense_rows": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp]),
    "thr_bm25_topk": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _dbl, _i64, _i64, _i64,
                             _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "thr_graph_workspace_bytes": (_sz, [_i32, _i64]),
    "thr_graph_topk": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32,
                              _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "thr_rrf_fuse": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _dbl, _dbl, _dbl, _i32, _i32,
                            _vp, _vp, _vp, _vp, _vp]),
    "thr_rrf_fuse_standalone": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _dbl, _dbl, _dbl, _i32, _i32,
                                       _vp, _vp, _vp, _vp, _vp]),
    "thr_fuse_post": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _dbl, _i32,
                             _dbl, _i32, _i32, _vp, _vp, _vp, _vp]),
    "thr_rerank_order": (_i32, [_vp, _i32, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "thr_maxsim": (_i32, [_vp, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    "thr_maxsim_ids": (_i32, [_vp, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _i32, _vp]),
    "thr_maxsim_pack": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "thr_merge_topk": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load libthr_hip.so; raises NativeError when it cannot (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        if not build_if_missing:
            raise NativeError(f"{path} is missing: build it with __graft_entry__.build()")
        _build.build_native()
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise NativeError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.thr_abi_version() != ABI_VERSION:
        raise NativeError("libthr_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().thr_error_string(rc).decode()
        raise NativeError(f"{what} failed: {msg} (code {rc})")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Optional[torch.Tensor], dtype, name: str, ndim: Optional[int] = None) -> int:
    if t is None:
        return 0
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeError(f"{name}: expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise NativeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise NativeError(f"{name}: must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise NativeError(f"{name}: expected {ndim} dims, got {t.dim()}")
    return t.data_ptr()


def device_info() -> Tuple[int, int, str]:
    cus, hbm = C.c_int(0), C.c_int64(0)
    arch = C.create_string_buffer(128)
    torch.cuda.current_device()  # make sure the HIP context exists on the right device
    _check(load().thr_device_info(C.byref(cus), C.byref(hbm), arch, 128), "thr_device_info")
    return cus.value, hbm.value, arch.value.decode()


# --------------------------------------------------------------------- a1
def embed_postproc(full: torch.Tensor, store_dim: int) -> torch.Tensor:
    p = _dev(full, torch.float32, "full", 2)
    n, fd = full.shape
    out = torch.empty((n, min(fd, store_dim)), dtype=torch.float32, device=full.device)
    if n:
        _check(load().thr_embed_postproc(p, n, fd, store_dim, out.data_ptr(), _stream()),
               "thr_embed_postproc")
    return out


def doc_norms(docs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    p = _dev(docs, torch.float32, "docs", 2)
    n, d = docs.shape
    dn = torch.empty(n, dtype=torch.float64, device=docs.device)
    inv = torch.empty(n, dtype=torch.float32, device=docs.device)
    if n:
        _check(load().thr_doc_norms(p, n, d, dn.data_ptr(), inv.data_ptr(), _stream()),
               "thr_doc_norms")
    return dn, inv


# --------------------------------------------------------------------- a2
def dense_workspace_bytes(n_docs: int, dim: int, n_queries: int, kprime: int) -> int:
    return int(load().thr_dense_workspace_bytes(n_docs, dim, n_queries, kprime))


def _alloc_out(nq: int, k: int, device):
    # scores and ids are the two halves of ONE [2, nq, k] 8-byte tile: the multi-GPU exchange
    # all-gathers that tile as it is (distributed.gather_topk), no packing copy
    tile = torch.empty((2, nq, k), dtype=torch.int64, device=device)
    return (tile[0].view(torch.float64), tile[1],
            torch.empty(nq, dtype=torch.int32, device=device),
            torch.empty(nq, dtype=torch.int32, device=device))


def _coll(doc_coll, query_coll, n: int, nq: int):
    """(doc_coll ptr, query_coll ptr) of the collection filter, or (None, None)."""
    if query_coll is None:
        return None, None
    if doc_coll is None:
        raise NativeError("query_coll without doc_coll")
    if doc_coll.shape != (n,) or query_coll.shape != (nq,):
        raise NativeError("doc_coll / query_coll have the wrong length")
    return _dev(doc_coll, torch.int32, "doc_coll", 1), _dev(query_coll, torch.int32, "query_coll", 1)


def dense_topk(docs, dnorm, inv_norm, queries, k: int, kprime: int, id_base: int = 0,
               workspace: Optional[torch.Tensor] = None, doc_coll=None, query_coll=None):
    """-> (scores f64 [nq,k], ids i64 [nq,k], counts i32 [nq], flags i32 [nq])."""
    pd = _dev(docs, torch.float32, "docs", 2)
    n, d = docs.shape
    pq = _dev(queries, torch.float32, "queries", 2)
    nq = queries.shape[0]
    if queries.shape[1] != d:
        raise NativeError(f"queries dim {queries.shape[1]} != docs dim {d}")
    pn = _dev(dnorm, torch.float64, "dnorm", 1)
    pi = _dev(inv_norm, torch.float32, "inv_norm", 1)
    if dnorm.shape[0] != n or inv_norm.shape[0] != n:
        raise NativeError("dnorm / inv_norm length != n_docs")
    need = dense_workspace_bytes(n, d, nq, kprime)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=docs.device)
    pw = _dev(workspace, workspace.dtype, "workspace")
    S, I, cnt, flg = _alloc_out(nq, k, docs.device)
    pdc, pqc = _coll(doc_coll, query_coll, n, nq)
    _check(load().thr_dense_topk(pd, pn, pi, n, d, id_base, pq, nq, k, kprime, pdc, pqc, S.data_ptr(),
                                 I.data_ptr(), cnt.data_ptr(), flg.data_ptr(), pw,
                                 workspace.numel() * workspace.element_size(), _stream()),
           "thr_dense_topk")
    return S, I, cnt, flg


def dense_topk_exact(docs, dnorm, queries, k: int, id_base: int = 0, doc_coll=None, query_coll=None):
    pd = _dev(docs, torch.float32, "docs", 2)
    n, d = docs.shape
    pq = _dev(queries, torch.float32, "queries", 2)
    nq = queries.shape[0]
    if queries.shape[1] != d:
        raise NativeError(f"queries dim {queries.shape[1]} != docs dim {d}")
    pn = _dev(dnorm, torch.float64, "dnorm", 1)
    if dnorm.shape[0] != n:
        raise NativeError("dnorm length != n_docs")
    need = int(load().thr_dense_exact_workspace_bytes(n, nq))
    ws = torch.empty(need, dtype=torch.uint8, device=docs.device)
    S, I, cnt, flg = _alloc_out(nq, k, docs.device)
    pdc, pqc = _coll(doc_coll, query_coll, n, nq)
    _check(load().thr_dense_topk_exact(pd, pn, n, d, id_base, pq, nq, k, pdc, pqc, S.data_ptr(),
                                       I.data_ptr(), cnt.data_ptr(), flg.data_ptr(),
                                       ws.data_ptr(), need, _stream()), "thr_dense_topk_exact")
    return S, I, cnt, flg


def dense_scan_probe(docs, inv_norm, queries, workspace: torch.Tensor) -> None:
    pd = _dev(docs, torch.float32, "docs", 2)
    n, d = docs.shape
    pq = _dev(queries, torch.float32, "queries", 2)
    if queries.shape[1] != d or inv_norm.shape[0] != n:
        raise NativeError("probe: shape mismatch")
    pi = _dev(inv_norm, torch.float32, "inv_norm", 1)
    pw = _dev(workspace, workspace.dtype, "workspace")
    _check(load().thr_dense_scan_probe(pd, pi, n, d, pq, queries.shape[0], pw,
                                       workspace.numel() * workspace.element_size(), _stream()),
           "thr_dense_scan_probe")


def dense_f16_query_tile(dim: int, packed: bool, n_queries: int) -> int:
    return int(load().thr_dense_f16_query_tile(dim, 1 if packed else 0, n_queries))


def dense_f16_max_queries(dim: int, packed: bool) -> int:
    """Largest batch of one thr_dense_topk_f16 call (32-bit candidate-segment offsets)."""
    return int(load().thr_dense_f16_max_queries(dim, 1 if packed else 0))


def dense_rescue_workspace_bytes(n_queries: int, k: int) -> int:
    return int(load().thr_dense_rescue_workspace_bytes(n_queries, k))


def dense_rescue(docs, dnorm, queries, S, I, cnt, flg, id_base: int = 0,
                 workspace: Optional[torch.Tensor] = None, doc_coll=None, query_coll=None) -> torch.Tensor:
    """Redo, in place and without a host read-back, the queries of a dense_topk[_f16] result
    whose flags lack THR_FLAG_CERTIFIED.  -> device int32[1]: how many were redone."""
    pd = _dev(docs, torch.float32, "docs", 2)
    pn = _dev(dnorm, torch.float64, "dnorm", 1)
    pq = _dev(queries, torch.float32, "queries", 2)
    n, d = docs.shape
    nq, k = I.shape
    need = int(load().thr_dense_rescue_workspace_bytes(nq, k))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=docs.device)
    n_rescued = torch.zeros(1, dtype=torch.int32, device=docs.device)
    pdc, pqc = _coll(doc_coll, query_coll, n, nq)
    _check(load().thr_dense_rescue(pd, pn, n, d, id_base, pq, nq, k, pdc, pqc, S.data_ptr(), I.data_ptr(),
                                   cnt.data_ptr(), flg.data_ptr(), n_rescued.data_ptr(),
                                   workspace.data_ptr(),
                                   workspace.numel() * workspace.element_size(), _stream()),
           "thr_dense_rescue")
    return n_rescued


def dense_quantize_f16(docs: torch.Tensor, keep_copy: bool = True):
    """-> (docs16 or None, max relative row error of the rounding as a float).  docs16 is the
    fragment-major float16 image of the rows (opaque; f16 [ceil32(n), D] elements)."""
    p = _dev(docs, torch.float32, "docs", 2)
    n, d = docs.shape
    d16 = None
    if keep_copy:
        nbytes = int(load().thr_dense_f16_copy_bytes(n, d))
        d16 = torch.empty((nbytes // (2 * d), d), dtype=torch.float16, device=docs.device)
    err = torch.zeros(1, dtype=torch.float32, device=docs.device)
    _check(load().thr_dense_quantize_f16(p, n, d, d16.data_ptr() if keep_copy else None,
                                         err.data_ptr(), _stream()), "thr_dense_quantize_f16")
    return d16, float(err.item())


def dense_f16_layout(dim: int) -> str:
    """Tag of the fragment-major layout thr_dense_quantize_f16 writes in this process: the
    register image of the MFMA shape the copy scan runs with (16x16x32, or 32x32x16 under the
    A/B knob THR_DENSE_MFMA=32, which the library reads the same way).  A saved float16 image is
    reused only under the tag it was written with."""
    return f"fragment-major/mfma{32 if os.environ.get('THR_DENSE_MFMA') == '32' else 16}/dim{dim}"


def dense_f16_workspace_bytes(n_docs: int, dim: int, n_queries: int, kprime: int) -> int:
    return int(load().thr_dense_f16_workspace_bytes(n_docs, dim, n_queries, kprime))


def dense_topk_f16(docs, docs16, doc_rel_err: float, dnorm, inv_norm, queries, k: int,
                   kprime: int, id_base: int = 0, workspace: Optional[torch.Tensor] = None,
                   doc_coll=None, query_coll=None):
    pd = _dev(docs, torch.float32, "docs", 2)
    ph = _dev(docs16, torch.float16, "docs16", 2) if docs16 is not None else None
    n, d = docs.shape
    if docs16 is not None and tuple(docs16.shape) != ((n + 31) // 32 * 32, d):
        raise NativeError("docs16 is not the fragment-major copy of docs (thr_dense_quantize_f16)")
    pq = _dev(queries, torch.float32, "queries", 2)
    nq = queries.shape[0]
    if queries.shape[1] != d:
        raise NativeError(f"queries dim {queries.shape[1]} != docs dim {d}")
    pn = _dev(dnorm, torch.float64, "dnorm", 1)
    pi = _dev(inv_norm, torch.float32, "inv_norm", 1)
    if dnorm.shape[0] != n or inv_norm.shape[0] != n:
        raise NativeError("dnorm / inv_norm length != n_docs")
    need = dense_f16_workspace_bytes(n, d, nq, kprime)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=docs.device)
    pw = _dev(workspace, workspace.dtype, "workspace")
    S, I, cnt, flg = _alloc_out(nq, k, docs.device)
    pdc, pqc = _coll(doc_coll, query_coll, n, nq)
    _check(load().thr_dense_topk_f16(pd, ph, float(doc_rel_err), pn, pi, n, d, id_base, pq, nq, k,
                                     kprime, pdc, pqc, S.data_ptr(), I.data_ptr(), cnt.data_ptr(),
                                     flg.data_ptr(), pw,
                                     workspace.numel() * workspace.element_size(), _stream()),
           "thr_dense_topk_f16")
    return S, I, cnt, flg


def dense_shortlist_f16(docs, docs16, doc_rel_err: float, inv_norm, queries, kprime: int, m: int,
                        workspace: torch.Tensor, doc_coll=None, query_coll=None) -> torch.Tensor:
    """First half of the search split around the shards' exchange (thr_hip.h e1): the filter scan
    (candidate lists left in ``workspace``) and per query the ``m`` largest scan scores as lower
    bounds of ||q|| x cosine -> float32 [nq, m], -inf padded."""
    pd = _dev(docs, torch.float32, "docs", 2)
    ph = _dev(docs16, torch.float16, "docs16", 2) if docs16 is not None else None
    n, d = docs.shape
    pq = _dev(queries, torch.float32, "queries", 2)
    nq = queries.shape[0]
    if queries.shape[1] != d or inv_norm.shape[0] != n:
        raise NativeError("shortlist: shape mismatch")
    pi = _dev(inv_norm, torch.float32, "inv_norm", 1)
    need = dense_f16_workspace_bytes(n, d, nq, kprime)
    if workspace.numel() * workspace.element_size() < need:
        raise NativeError("shortlist: workspace smaller than thr_dense_f16_workspace_bytes")
    pw = _dev(workspace, workspace.dtype, "workspace")
    pdc, pqc = _coll(doc_coll, query_coll, n, nq)
    lb = torch.empty((nq, m), dtype=torch.float32, device=docs.device)
    _check(load().thr_dense_shortlist_f16(pd, ph, float(doc_rel_err), pi, n, d, pq, nq, kprime, pdc, pqc,
                                          m, lb.data_ptr(), pw,
                                          workspace.numel() * workspace.element_size(), _stream()),
           "thr_dense_shortlist_f16")
    return lb


def dense_floor(top_lb: torch.Tensor, k: int) -> torch.Tensor:
    """[n_shards, nq, m] gathered lower bounds -> float32 [nq]: the k-th largest per query, a lower
    bound of ||q|| x the k-th best cosine over all the shards (-inf: no floor)."""
    p = _dev(top_lb, torch.float32, "top_lb", 3)
    g, nq, m = top_lb.shape
    out = torch.empty(nq, dtype=torch.float32, device=top_lb.device)
    _check(load().thr_dense_floor(p, g, nq, m, k, out.data_ptr(), _stream()), "thr_dense_floor")
    return out


def dense_finish_f16(docs, docs16, doc_rel_err: float, dnorm, inv_norm, queries, k: int, kprime: int,
                     gfloor: Optional[torch.Tensor], id_base: int, workspace: torch.Tensor,
                     doc_coll=None, query_coll=None, lb_all: Optional[torch.Tensor] = None):
    """Second half: band, float64 rescoring, order and certificate on the candidate lists the
    matching dense_shortlist_f16 call left in ``workspace`` (same arguments), with the shards'
    common floor: ``gfloor`` [nq] (dense_floor's output), or ``lb_all`` [n_shards, nq, m] -- the
    gathered bounds themselves, the k-th largest found inside the band kernel."""
    pd = _dev(docs, torch.float32, "docs", 2)
    ph = _dev(docs16, torch.float16, "docs16", 2) if docs16 is not None else None
    n, d = docs.shape
    pq = _dev(queries, torch.float32, "queries", 2)
    nq = queries.shape[0]
    pn = _dev(dnorm, torch.float64, "dnorm", 1)
    pi = _dev(inv_norm, torch.float32, "inv_norm", 1)
    if queries.shape[1] != d or dnorm.shape[0] != n or inv_norm.shape[0] != n:
        raise NativeError("finish: shape mismatch")
    pg = None
    if gfloor is not None:
        pg = _dev(gfloor, torch.float32, "gfloor", 1)
        if gfloor.shape[0] != nq:
            raise NativeError("gfloor: one value per query")
    pl, g, m = None, 0, 0
    if lb_all is not None:
        if gfloor is not None:
            raise NativeError("finish: gfloor or lb_all, not both")
        pl = _dev(lb_all, torch.float32, "lb_all", 3)
        g, nq_l, m = lb_all.shape
        if nq_l != nq:
            raise NativeError("lb_all: [n_shards, nq, m]")
    pw = _dev(workspace, workspace.dtype, "workspace")
    S, I, cnt, flg = _alloc_out(nq, k, docs.device)
    pdc, pqc = _coll(doc_coll, query_coll, n, nq)
    _check(load().thr_dense_finish_f16(pd, ph, float(doc_rel_err), pn, pi, n, d, id_base, pq, nq, k,
                                       kprime, pdc, pqc, pg, pl, g, m, S.data_ptr(), I.data_ptr(),
                                       cnt.data_ptr(), flg.data_ptr(), pw,
                                       workspace.numel() * workspace.element_size(), _stream()),
           "thr_dense_finish_f16")
    return S, I, cnt, flg


def dense_scan_probe_f16(docs, docs16, inv_norm, queries, workspace: torch.Tensor) -> None:
    pd = _dev(docs, torch.float32, "docs", 2)
    ph = _dev(docs16, torch.float16, "docs16", 2) if docs16 is not None else None
    n, d = docs.shape
    pq = _dev(queries, torch.float32, "queries", 2)
    if queries.shape[1] != d or inv_norm.shape[0] != n:
        raise NativeError("probe: shape mismatch")
    pi = _dev(inv_norm, torch.float32, "inv_norm", 1)
    pw = _dev(workspace, workspace.dtype, "workspace")
    _check(load().thr_dense_scan_probe_f16(pd, ph, pi, n, d, pq, queries.shape[0], pw,
                                           workspace.numel() * workspace.element_size(),
                                           _stream()), "thr_dense_scan_probe_f16")


def dense_scan_stamps_f16(docs16, n_docs: int, queries_n: int, workspace: torch.Tensor):
    """Phase stamps of the float16-copy scan -> int64 [n_waves, 8] (see thr_hip.h)."""
    ph = _dev(docs16, torch.float16, "docs16", 2)
    d = docs16.shape[1]
    pw = _dev(workspace, workspace.dtype, "workspace")
    nbytes = workspace.numel() * workspace.element_size()
    nw = C.c_int(0)
    _check(load().thr_dense_scan_stamps_f16(ph, n_docs, d, queries_n, pw, nbytes, None, C.byref(nw),
                                            _stream()), "thr_dense_scan_stamps_f16")
    out = torch.zeros((nw.value, 8), dtype=torch.int64, device=docs16.device)
    _check(load().thr_dense_scan_stamps_f16(ph, n_docs, d, queries_n, pw, nbytes, out.data_ptr(),
                                            C.byref(nw), _stream()), "thr_dense_scan_stamps_f16")
    return out


# --------------------------------------------------------------------- f1
def lexical_build(doc: torch.Tensor, term: torch.Tensor, tf: Optional[torch.Tensor], n_docs: int,
                  n_vocab: int):
    """Tokenised rows on the device -> (rowptr i64 [V+1], post_doc i32 [nnz], post_tf i32 [nnz],
    doclen f32 [n_docs], df i64 [V]): the CSR inverted index of this shard.  ``tf`` None: one
    entry per token occurrence.  One host read-back at the end (the number of postings, to cut
    the posting arrays to size)."""
    pdoc = _dev(doc, torch.int32, "doc", 1)
    pterm = _dev(term, torch.int32, "term", 1)
    ptf = _dev(tf, torch.int32, "tf", 1) if tf is not None else None
    n = doc.shape[0]
    if term.shape[0] != n or (tf is not None and tf.shape[0] != n):
        raise NativeError("lexical_build: doc / term / tf lengths differ")
    dev = doc.device
    rowptr = torch.empty(n_vocab + 1, dtype=torch.int64, device=dev)
    post_doc = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    post_tf = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    doclen = torch.empty(n_docs, dtype=torch.float32, device=dev)
    df = torch.empty(n_vocab, dtype=torch.int64, device=dev)
    nnz = torch.zeros(1, dtype=torch.int64, device=dev)
    if n == 0:
        return rowptr.zero_(), post_doc[:0], post_tf[:0], doclen.zero_(), df.zero_()
    need = int(load().thr_lexical_build_workspace_bytes(n, n_vocab))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _check(load().thr_lexical_build(pdoc, pterm, ptf, n, n_docs, n_vocab, rowptr.data_ptr(),
                                    post_doc.data_ptr(), post_tf.data_ptr(), doclen.data_ptr(),
                                    df.data_ptr(), nnz.data_ptr(), ws.data_ptr(), need, _stream()),
           "thr_lexical_build")
    k = int(nnz.item())
    return rowptr, post_doc[:k].clone(), post_tf[:k].clone(), doclen, df


# --------------------------------------------------------------------- a3
def bm25_bounds(rowptr, post_doc, post_tf, doclen, idf, avgdl: float, k1: float = 1.2,
                b: float = 0.75):
    """Index set-up -> (term_ub f64 [V], block_ub f64 [ceil(nnz/128)], post_imp u8 [nnz]): the
    upper bounds thr_bm25_topk prunes with (per term, per 128 postings, per posting)."""
    pr = _dev(rowptr, torch.int64, "rowptr", 1)
    pdoc = _dev(post_doc, torch.int32, "post_doc", 1)
    ptf = _dev(post_tf, torch.int32, "post_tf", 1)
    pdl = _dev(doclen, torch.float32, "doclen", 1)
    pidf = _dev(idf, torch.float64, "idf", 1)
    v, nnz = idf.shape[0], post_doc.shape[0]
    tub = torch.zeros(v, dtype=torch.float64, device=rowptr.device)
    bub = torch.zeros(max(int(load().thr_bm25_block_count(nnz)), 1), dtype=torch.float64,
                      device=rowptr.device)
    imp = torch.zeros(nnz + 4, dtype=torch.uint8, device=rowptr.device)   # (+4: read as aligned 32-bit words)
    if nnz:
        _check(load().thr_bm25_bounds(pr, pdoc, ptf, pdl, pidf, avgdl, k1, b, v, nnz, tub.data_ptr(),
                                      bub.data_ptr(), imp.data_ptr(), _stream()), "thr_bm25_bounds")
    return tub, bub, imp


def bm25_dense_terms(rowptr, post_doc, post_tf, post_imp, n_docs: int, min_share: float = 0.125,
                     max_terms: int = 512, max_bytes: int = 12 << 30):
    """Index set-up -> (dense_slot i32 [V], dense_imp u8 [T, stride], dense_tf u16 [T, stride],
    stride) for the terms held by at least ``min_share`` of the docs (the ``max_terms`` longest
    of them, term frequencies <= 65535), or None when there is none: per-doc rows of impacts and
    term frequencies, so that thr_bm25_topk never walks a stop word's postings.  At most
    ``max_bytes`` of rows (12 GiB: 400 terms of a 10M-doc shard)."""
    _dev(rowptr, torch.int64, "rowptr", 1)
    df = rowptr[1:] - rowptr[:-1]
    cand = torch.nonzero(df.to(torch.float64) >= min_share * n_docs).flatten()
    if cand.numel() == 0 or n_docs <= 0:
        return None
    # 3 bytes per doc and term: the longest lists first, as many as ``max_bytes`` holds
    max_terms = min(max_terms, max_bytes // (3 * int(load().thr_bm25_dense_stride(n_docs))))
    if max_terms <= 0:
        return None
    keep = []
    for t in cand[torch.argsort(df[cand], descending=True, stable=True)][:max_terms].tolist():
        lo, hi = int(rowptr[t]), int(rowptr[t + 1])
        if int(post_tf[lo:hi].max()) <= 65535:
            keep.append(t)
    if not keep:
        return None
    terms = torch.tensor(sorted(keep), dtype=torch.int32, device=rowptr.device)
    stride = int(load().thr_bm25_dense_stride(n_docs))
    dimp = torch.empty((len(keep), stride), dtype=torch.uint8, device=rowptr.device)
    dtf = torch.empty((len(keep), stride), dtype=torch.int16, device=rowptr.device)   # (read as uint16)
    _check(load().thr_bm25_dense_rows(_dev(rowptr, torch.int64, "rowptr", 1), _dev(post_doc, torch.int32, "post_doc", 1),
                                      _dev(post_tf, torch.int32, "post_tf", 1), _dev(post_imp, torch.uint8, "post_imp", 1),
                                      terms.data_ptr(), len(keep), n_docs, int(df[terms.long()].max()),
                                      dimp.data_ptr(), dtf.data_ptr(), _stream()), "thr_bm25_dense_rows")
    slot = torch.full((rowptr.shape[0] - 1,), -1, dtype=torch.int32, device=rowptr.device)
    slot[terms.long()] = torch.arange(len(keep), dtype=torch.int32, device=rowptr.device)
    return slot, dimp, dtf, stride


def bm25_workspace_bytes(n_queries: int, max_terms: int, k: int) -> int:
    return int(load().thr_bm25_workspace_bytes(n_queries, max_terms, k))


def bm25_topk(rowptr, post_doc, post_tf, doclen, idf, avgdl: float, query_terms, k: int,
              id_base: int = 0, k1: float = 1.2, b: float = 0.75, bounds=None,
              conjunctive: bool = False, doc_coll=None, query_coll=None,
              workspace: Optional[torch.Tensor] = None, dense=None):
    """``dense``: bm25_dense_terms' tuple (needs ``bounds`` with the impacts).
    ``workspace``: a uint8 device tensor of >= bm25_workspace_bytes(nq, max_terms, k) bytes
    (item list, slice edges and per-slice lists of the work decomposition); allocated when
    missing or too small."""
    pr = _dev(rowptr, torch.int64, "rowptr", 1)
    pdoc = _dev(post_doc, torch.int32, "post_doc", 1)
    ptf = _dev(post_tf, torch.int32, "post_tf", 1)
    pdl = _dev(doclen, torch.float32, "doclen", 1)
    pidf = _dev(idf, torch.float64, "idf", 1)
    pqt = _dev(query_terms, torch.int32, "query_terms", 2)
    if post_doc.shape != post_tf.shape or idf.shape[0] + 1 != rowptr.shape[0]:
        raise NativeError("bm25: CSR arrays are inconsistent")
    nq, mt = query_terms.shape
    if mt > THR_BM25_MAX_TERMS or k > THR_TOPK_MAX:
        raise NativeError("bm25: too many terms per query or k too large")
    ptu = pbu = pim = None
    if bounds is not None:
        ptu, pbu = _dev(bounds[0], torch.float64, "term_ub", 1), _dev(bounds[1], torch.float64, "block_ub", 1)
        if bounds[0].shape[0] != idf.shape[0]:
            raise NativeError("bm25: term_ub length != vocabulary size")
        if len(bounds) > 2 and bounds[2] is not None:
            pim = _dev(bounds[2], torch.uint8, "post_imp", 1)
            if bounds[2].shape[0] < post_doc.shape[0]:
                raise NativeError("bm25: post_imp shorter than the posting array")
    pds = pdi = pdt = None
    dstride = 0
    if dense is not None:
        if pim is None:
            raise NativeError("bm25: dense rows need the bounds with the impacts")
        pds, pdi = _dev(dense[0], torch.int32, "dense_slot", 1), _dev(dense[1], torch.uint8, "dense_imp", 2)
        pdt, dstride = _dev(dense[2], torch.int16, "dense_tf", 2), int(dense[3])
        if dense[0].shape[0] != idf.shape[0] or dense[1].shape != dense[2].shape or dense[1].shape[1] != dstride:
            raise NativeError("bm25: dense rows are inconsistent")
    pdc = pqc = None
    if query_coll is not None:
        if doc_coll is None:
            raise NativeError("bm25: query_coll without doc_coll")
        pdc, pqc = _dev(doc_coll, torch.int32, "doc_coll", 1), _dev(query_coll, torch.int32, "query_coll", 1)
        if doc_coll.shape[0] != doclen.shape[0] or query_coll.shape[0] != nq:
            raise NativeError("bm25: doc_coll / query_coll have the wrong length")
    S, I, cnt, _ = _alloc_out(nq, k, rowptr.device)
    need = bm25_workspace_bytes(nq, mt, k)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=rowptr.device)
    pw = _dev(workspace, workspace.dtype, "workspace")
    _check(load().thr_bm25_topk(pr, pdoc, ptf, pdl, pidf, ptu, pbu, pim, pds, pdi, pdt, dstride,
                                avgdl, k1, b, doclen.shape[0],
                                idf.shape[0], id_base, pqt, nq, mt, k, 1 if conjunctive else 0, pdc,
                                pqc, S.data_ptr(), I.data_ptr(), cnt.data_ptr(), pw,
                                workspace.numel() * workspace.element_size(), _stream()),
           "thr_bm25_topk")
    return S, I, cnt


# --------------------------------------------------------------------- a4
def graph_workspace_bytes(n_queries: int, n_entities: int) -> int:
    return int(load().thr_graph_workspace_bytes(n_queries, n_entities))


def graph_topk(ent_rowptr, ent_col, men_rowptr, men_chunk, men_conf, query_seeds, hops: int,
               k: int, chunk_base: int, n_chunks: int, transposed=None,
               workspace: Optional[torch.Tensor] = None):
    """-> (scores, ids, counts, flags).  ``transposed`` = (tmen_rowptr i64 [n_chunks+1], tmen_ent
    i32, tmen_conf f32) from ``graph_transpose_mentions`` enables the capacity-free third tier."""
    per = _dev(ent_rowptr, torch.int64, "ent_rowptr", 1)
    pec = _dev(ent_col, torch.int32, "ent_col", 1)
    pmr = _dev(men_rowptr, torch.int64, "men_rowptr", 1)
    pmc = _dev(men_chunk, torch.int32, "men_chunk", 1)
    pmw = _dev(men_conf, torch.float32, "men_conf", 1)
    pqs = _dev(query_seeds, torch.int32, "query_seeds", 2)
    if ent_rowptr.shape != men_rowptr.shape or men_chunk.shape != men_conf.shape:
        raise NativeError("graph: CSR arrays are inconsistent")
    ptr = pte = ptw = None
    n_ent = ent_rowptr.shape[0] - 1
    if transposed is not None:
        tr, te, tw = transposed
        ptr, pte, ptw = (_dev(tr, torch.int64, "tmen_rowptr", 1), _dev(te, torch.int32, "tmen_ent", 1),
                         _dev(tw, torch.float32, "tmen_conf", 1))
        if tr.shape[0] != n_chunks + 1 or te.shape != tw.shape:
            raise NativeError("graph: transposed mention CSR is inconsistent")
    nq, ms = query_seeds.shape
    if ms > THR_GRAPH_MAX_SEEDS or k > THR_TOPK_MAX:
        raise NativeError("graph: too many seeds per query or k too large")
    need = int(load().thr_graph_workspace_bytes(nq, n_ent if transposed is not None else 0))
    ws = workspace
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty(max(need, 8), dtype=torch.uint8, device=ent_rowptr.device)
    S, I, cnt, flg = _alloc_out(nq, k, ent_rowptr.device)
    _check(load().thr_graph_topk(per, pec, n_ent, pmr, pmc, pmw, ptr, pte, ptw, chunk_base,
                                 n_chunks, pqs, nq, ms, hops, k, S.data_ptr(), I.data_ptr(),
                                 cnt.data_ptr(), flg.data_ptr(), ws.data_ptr(), need, _stream()),
           "thr_graph_topk")
    return S, I, cnt, flg


def graph_transpose_mentions(men_rowptr, men_chunk, men_conf, chunk_base: int, n_chunks: int):
    """Index set-up (device tensors, torch ops): the entity-major mention CSR restricted to this
    shard's chunks, re-sorted chunk-major with a STABLE sort, so a chunk's entries stay in
    (entity asc, mention) order -- the oracle's summation order."""
    ne = men_rowptr.shape[0] - 1
    ent = torch.repeat_interleave(torch.arange(ne, dtype=torch.int32, device=men_rowptr.device),
                                  (men_rowptr[1:] - men_rowptr[:-1]))
    local = men_chunk.to(torch.int64) - chunk_base
    keep = (local >= 0) & (local < n_chunks)
    ent, conf, local = ent[keep], men_conf[keep], local[keep]
    order = torch.sort(local, stable=True).indices
    counts = torch.bincount(local, minlength=n_chunks)
    rowptr = torch.zeros(n_chunks + 1, dtype=torch.int64, device=men_rowptr.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, ent[order].contiguous(), conf[order].contiguous()


# ----------------------------------------------------------------- a5 + a6
def rrf_fuse(lex_ids, sem_ids, graph_ids, top_k: int, w_lex: float = 0.7, w_sem: float = 0.8,
             w_graph: float = 1.0, rrf_k: int = 60, want_ranks: bool = False):
    chans = [lex_ids, sem_ids, graph_ids]
    ref = next(c for c in chans if c is not None)
    nq = ref.shape[0]
    ptrs, widths = [], []
    for name, c in zip(("lex_ids", "sem_ids", "graph_ids"), chans):
        if c is None:
            ptrs.append(0)
            widths.append(0)
            continue
        ptrs.append(_dev(c, torch.int64, name, 2))
        if c.shape[0] != nq or c.shape[1] > THR_RRF_MAX_PER_CHANNEL:
            raise NativeError(f"rrf: bad shape for {name}")
        widths.append(c.shape[1])
    dev = ref.device
    out_ids = torch.empty((nq, top_k), dtype=torch.int64, device=dev)
    out_sc = torch.empty((nq, top_k), dtype=torch.float64, device=dev)
    out_rk = torch.empty((nq, top_k, 3), dtype=torch.int32, device=dev) if want_ranks else None
    cnt = torch.empty(nq, dtype=torch.int32, device=dev)
    _check(load().thr_rrf_fuse(ptrs[0], widths[0], ptrs[1], widths[1], ptrs[2], widths[2], nq,
                               w_lex, w_sem, w_graph, rrf_k, top_k, out_ids.data_ptr(),
                               out_sc.data_ptr(), out_rk.data_ptr() if want_ranks else 0,
                               cnt.data_ptr(), _stream()), "thr_rrf_fuse")
    return out_ids, out_sc, out_rk, cnt


def rrf_fuse_standalone(lex_ids, sem_ids, graph_ids, top_k: int, w_lex: float = 0.7, w_sem: float = 0.8,
                        w_graph: float = 1.0, two_channels: bool = False, want_ranks: bool = True):
    """The standalone package's RRFFusion.fuse / fuse_two_channels on id lists [nq, n_c] (-1
    padded) -> (ids i64 [nq, top_k], scores f64, ranks i32 [nq, top_k, 3] or None, counts)."""
    chans = [lex_ids, sem_ids, graph_ids]
    ref = next(c for c in chans if c is not None)
    nq = ref.shape[0]
    ptrs, widths = [], []
    for name, c in zip(("lex_ids", "sem_ids", "graph_ids"), chans):
        if c is None:
            ptrs.append(0)
            widths.append(0)
            continue
        ptrs.append(_dev(c, torch.int64, name, 2))
        if c.shape[0] != nq or c.shape[1] > THR_RRF_MAX_PER_CHANNEL:
            raise NativeError(f"rrf: bad shape for {name}")
        widths.append(c.shape[1])
    dev = ref.device
    out_ids = torch.empty((nq, top_k), dtype=torch.int64, device=dev)
    out_sc = torch.empty((nq, top_k), dtype=torch.float64, device=dev)
    out_rk = torch.empty((nq, top_k, 3), dtype=torch.int32, device=dev) if want_ranks else None
    cnt = torch.empty(nq, dtype=torch.int32, device=dev)
    _check(load().thr_rrf_fuse_standalone(ptrs[0], widths[0], ptrs[1], widths[1], ptrs[2], widths[2], nq,
                                          w_lex, w_sem, w_graph, 1 if two_channels else 0, top_k,
                                          out_ids.data_ptr(), out_sc.data_ptr(),
                                          out_rk.data_ptr() if want_ranks else 0, cnt.data_ptr(),
                                          _stream()), "thr_rrf_fuse_standalone")
    return out_ids, out_sc, out_rk, cnt


def fuse_post(ids, scores, ranks=None, counts=None, channel_scores=(None, None, None),
              safety_threshold: float = 0.0, denoise_quantile: Optional[float] = None,
              normalize: bool = False, top_k: int = 0):
    """Safety threshold / percentile denoise / truncation / min-max normalisation of a batch of
    fused lists (thr_fuse_post).  denoise_quantile = ((1 - alpha) * 100) / 100 or None."""
    pi = _dev(ids, torch.int64, "ids", 2)
    ps = _dev(scores, torch.float64, "scores", 2)
    nq, n = ids.shape
    if tuple(scores.shape) != (nq, n):
        raise NativeError("fuse_post: ids / scores shape mismatch")
    pr = _dev(ranks, torch.int32, "ranks", 3) if ranks is not None else None
    if ranks is not None and tuple(ranks.shape) != (nq, n, 3):
        raise NativeError("fuse_post: ranks must be [nq, n, 3]")
    pc = _dev(counts, torch.int32, "counts", 1) if counts is not None else None
    cp, cw = [], []
    for name, c in zip(("lex_scores", "sem_scores", "graph_scores"), channel_scores):
        cp.append(_dev(c, torch.float64, name, 2) if c is not None else None)
        cw.append(c.shape[1] if c is not None else 0)
        if c is not None and c.shape[0] != nq:
            raise NativeError(f"fuse_post: bad shape for {name}")
    out_ids = torch.empty_like(ids)
    out_sc = torch.empty_like(scores)
    out_c = torch.empty(nq, dtype=torch.int32, device=ids.device)
    _check(load().thr_fuse_post(pi, ps, pr, pc, nq, n, cp[0], cw[0], cp[1], cw[1], cp[2], cw[2],
                                float(safety_threshold), 0 if denoise_quantile is None else 1,
                                float(denoise_quantile or 0.0), 1 if normalize else 0, int(top_k),
                                out_ids.data_ptr(), out_sc.data_ptr(), out_c.data_ptr(), _stream()),
           "thr_fuse_post")
    return out_ids, out_sc, out_c


# --------------------------------------------------------------------- a8
def maxsim_pack(dtok: torch.Tensor) -> torch.Tensor:
    """Row-major token store -> fragment-major image for maxsim(..., packed=True)."""
    pdt = _dev(dtok, torch.float16, "dtok", 3)
    nd, dt, td = dtok.shape
    if dt % 32 or td % 16:
        raise NativeError("maxsim: d_tokens must be a multiple of 32, tok_dim of 16")
    out = torch.empty_like(dtok)
    _check(load().thr_maxsim_pack(pdt, nd, dt, td, out.data_ptr(), _stream()), "thr_maxsim_pack")
    return out


def maxsim(qtok: torch.Tensor, dtok: torch.Tensor, cand: torch.Tensor,
           packed: bool = False) -> torch.Tensor:
    """cand holds LOCAL doc indices; negative or out-of-range entries score -inf."""
    pq = _dev(qtok, torch.float16, "qtok", 3)
    pdt = _dev(dtok, torch.float16, "dtok", 3)
    pc = _dev(cand, torch.int32, "cand", 2)
    nq, qt, td = qtok.shape
    nd, dt, td2 = dtok.shape
    if td != td2 or cand.shape[0] != nq:
        raise NativeError("maxsim: shape mismatch")
    if qt % 32 or dt % 32 or td % 16:
        raise NativeError("maxsim: q_tokens/d_tokens must be multiples of 32, tok_dim of 16")
    out = torch.empty(cand.shape, dtype=torch.float32, device=qtok.device)
    _check(load().thr_maxsim(pq, nq, qt, pdt, nd, dt, td, pc, cand.shape[1], out.data_ptr(),
                             1 if packed else 0, _stream()), "thr_maxsim")
    return out


def maxsim_ids(qtok: torch.Tensor, dtok: torch.Tensor, cand_ids: torch.Tensor, id_base: int,
               packed: bool = False) -> torch.Tensor:
    """cand_ids holds GLOBAL doc ids (int64) of a shard starting at ``id_base``; negative ids and
    ids outside the shard score -inf."""
    pq = _dev(qtok, torch.float16, "qtok", 3)
    pdt = _dev(dtok, torch.float16, "dtok", 3)
    pc = _dev(cand_ids, torch.int64, "cand_ids", 2)
    nq, qt, td = qtok.shape
    nd, dt, td2 = dtok.shape
    if td != td2 or cand_ids.shape[0] != nq:
        raise NativeError("maxsim: shape mismatch")
    if qt % 32 or dt % 32 or td % 16:
        raise NativeError("maxsim: q_tokens/d_tokens must be multiples of 32, tok_dim of 16")
    out = torch.empty(cand_ids.shape, dtype=torch.float32, device=qtok.device)
    _check(load().thr_maxsim_ids(pq, nq, qt, pdt, nd, dt, td, pc, int(id_base), cand_ids.shape[1],
                                 out.data_ptr(), 1 if packed else 0, _stream()), "thr_maxsim_ids")
    return out


def rerank_order(scores: torch.Tensor, ids: torch.Tensor, counts: Optional[torch.Tensor], top_k: int):
    """scores f32 [nq, n] or [n_lists, nq, n] (per-shard MaxSim lists: -inf where the shard does
    not own the candidate) -> (ids i64 [nq, top_k], scores f64 [nq, top_k], counts i32 [nq]):
    the reference's stable descending sort on ``rerank_score or 0`` (retrieval.py:455)."""
    if scores.dim() == 2:
        scores = scores.unsqueeze(0)
    ps = _dev(scores, torch.float32, "scores", 3)
    pi = _dev(ids, torch.int64, "ids", 2)
    pc = _dev(counts, torch.int32, "counts", 1) if counts is not None else None
    nl, nq, n = scores.shape
    if tuple(ids.shape) != (nq, n):
        raise NativeError("rerank_order: ids shape != scores shape")
    out_ids = torch.empty((nq, top_k), dtype=torch.int64, device=ids.device)
    out_s = torch.empty((nq, top_k), dtype=torch.float64, device=ids.device)
    out_c = torch.empty(nq, dtype=torch.int32, device=ids.device)
    _check(load().thr_rerank_order(ps, nl, nq * n, pi, pc, nq, n, top_k, out_ids.data_ptr(),
                                   out_s.data_ptr(), out_c.data_ptr(), _stream()), "thr_rerank_order")
    return out_ids, out_s, out_c


# ------------------------------------------------------------- multi-GPU
def merge_topk(in_scores: torch.Tensor, in_ids: torch.Tensor, k_out: int):
    """in_* are [n_lists, n_queries, k_in] (the layout an all-gather of each rank's
    [n_queries, k_in] block produces); they may be strided views of one gathered
    [n_lists, 2, n_queries, k_in] tile (same list stride, contiguous [n_queries, k_in] blocks)."""
    if in_scores.shape != in_ids.shape or in_scores.dim() != 3:
        raise NativeError("merge: shape mismatch")
    if in_scores.dtype != torch.float64 or in_ids.dtype != torch.int64 or not in_scores.is_cuda:
        raise NativeError("merge: in_scores must be float64, in_ids int64, on the GPU")
    nl, nq, kin = in_scores.shape
    for t in (in_scores, in_ids):
        if t.stride(2) != 1 or t.stride(1) != kin or (nl > 1 and t.stride(0) < nq * kin):
            raise NativeError("merge: lists must be contiguous [n_queries, k_in] blocks")
    if nl > 1 and in_scores.stride(0) != in_ids.stride(0):
        raise NativeError("merge: scores and ids must share the list stride")
    stride = in_scores.stride(0) if nl > 1 else nq * kin
    S, I, cnt, _ = _alloc_out(nq, k_out, in_scores.device)
    _check(load().thr_merge_topk(in_scores.data_ptr(), in_ids.data_ptr(), nq, nl, kin, stride,
                                 k_out, S.data_ptr(), I.data_ptr(), cnt.data_ptr(), _stream()),
           "thr_merge_topk")
    return S, I, cnt
