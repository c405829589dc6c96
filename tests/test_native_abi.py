"""CPU-side checks of the C-ABI library: it builds, loads and exports every symbol
include/thr_hip.h declares (no compute calls without a GPU)."""
import os
import re

import triple_hybrid_rag_amd as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "thr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(thr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    T._build.build_native()
    lib = T._native.load()
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in thr_hip.h but not exported"
    assert sorted(T._native.EXPORTED_SYMBOLS) == names
    assert lib.thr_abi_version() == T._native.ABI_VERSION == 9
    assert lib.thr_error_string(-3) == b"workspace too small"


def test_host_side_argument_checks_do_not_need_a_gpu():
    lib = T._native.load()
    # null pointers / bad sizes are rejected before any launch
    assert lib.thr_dense_topk(None, None, None, 10, 768, 0, None, 1, 10, 128, None, None, None, None,
                              None, None, None, 0, None) == -1
    # bm25: a vocabulary size is part of the call (term ids >= V are ignored, not dereferenced)
    assert lib.thr_bm25_topk(None, None, None, None, None, None, None, None, None, None, None, 0, 1.0, 1.2, 0.75,
                             10, 5, 0, None, 1, 4, 10, 0, None, None, None, None, None, None, 0, None) == -1
    # dense-term rows: one window of zero padding behind the shard's docs, 16-byte aligned
    assert lib.thr_bm25_dense_stride(1000) == 1008 + 65536 and lib.thr_bm25_dense_stride(0) == 0
    assert lib.thr_bm25_dense_rows(None, None, None, None, None, 1, 10, 5, None, None, None) == -1
    # the work decomposition's item list, slice edges and per-slice lists: grows with nq, terms, k
    assert lib.thr_bm25_workspace_bytes(2048, 4, 50) > (2048 + 8192) * 50 * 16
    assert lib.thr_bm25_workspace_bytes(2048, 32, 50) > lib.thr_bm25_workspace_bytes(2048, 4, 50)
    assert lib.thr_bm25_workspace_bytes(0, 4, 50) == 0
    assert lib.thr_bm25_block_count(129) == 2 and lib.thr_bm25_block_count(0) == 0
    assert lib.thr_graph_workspace_bytes(2, 1000) == 2 * 8192 * 8 + 64 * 1024
    assert lib.thr_dense_workspace_bytes(1_000_000, 768, 1024, 128) > 2 ** 20
    assert lib.thr_maxsim(None, 1, 32, None, 1, 128, 128, None, 1, None, 0, None) == -1
    assert lib.thr_rrf_fuse(None, 0, None, 0, None, 0, 1, 0.7, 0.8, 1.0, 60, 10, None, None, None,
                            None, None) == -1


def test_host_side_planning_functions():
    """Sizes and the query-tile choice are pure host arithmetic (no launch)."""
    N = T._native
    N.load()
    # f16 copy scan: 32 queries per wave (their B operands live in registers), 8 waves per block;
    # at dim 1024 4 waves (one per SIMD) of 48 queries -- 384 B-operand registers each
    assert N.dense_f16_query_tile(768, True, 1024) == 256
    assert N.dense_f16_query_tile(512, True, 1536) == 256
    assert N.dense_f16_query_tile(1024, True, 1536) == 192
    assert N.dense_f16_query_tile(768, False, 1536) == 64     # in-flight rounding: transpose tiles
    assert N.dense_f16_query_tile(1024, False, 1536) == 32
    assert N.dense_f16_query_tile(640, True, 64) == 0         # no f16 kernel at that dim
    # the copy scan's candidate-segment offsets are 32 bits: 131072 bytes per query, whole tiles
    assert N.dense_f16_max_queries(768, True) == 32512 and N.dense_f16_max_queries(1024, True) == 32640
    assert N.dense_f16_max_queries(768, True) * 131072 < 2 ** 32
    assert N.dense_f16_max_queries(768, False) == 2 ** 31 - 1
    lib = N.load()
    assert lib.thr_dense_f16_copy_bytes(33, 768) == 64 * 768 * 2   # rows padded to tiles of 32
    assert lib.thr_dense_rescue_workspace_bytes(1024, 100) == 1024 * 64 * 100 * 16
    assert lib.thr_dense_f16_workspace_bytes(1_000_000, 768, 1536, 192) >= \
        lib.thr_dense_f16_workspace_bytes(1_000_000, 768, 1024, 192)
    # the threshold sample grows with the corpus (n * 64 / 4096 rows, one float per query and
    # row): a capped sample would let tens of thousands of rows per query through on a 10M-row
    # shard and overflow every candidate list
    grow = lib.thr_dense_f16_workspace_bytes(10_000_000, 768, 1536, 192) - \
        lib.thr_dense_f16_workspace_bytes(1_000_000, 768, 1536, 192)
    assert abs(grow - 1536 * (10_000_000 - 1_000_000) * 64 // 4096 * 4) < 64 * 1536 * 4 + 4096


def test_no_cpu_fallback_in_the_product_package():
    """The product path must not import the oracle or fall back to CPU math."""
    pkg = os.path.join(ROOT, "triple-hybrid-rag_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
