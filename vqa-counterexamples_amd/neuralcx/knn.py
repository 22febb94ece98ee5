"""k nearest neighbours of image-feature rows on the GPU.

Same result as the reference's knn.py:41-58 -- sklearn `NearestNeighbors(n_neighbors=k).fit(features)` followed by
`kneighbors(features[i:i+b])` over the whole table (brute force, euclidean; each row's first neighbour is itself) --
computed by ncx_knn: one fp32 MFMA GEMM per block of queries + an on-device select / exact re-rank.
"""
import ctypes as C

import torch

from . import _lib
from .ops import _ptr, _stream


def knn(table: torch.Tensor, k: int = 25, queries: torch.Tensor = None, block_rows: int = 4096):
    """table [n, dv] fp32 on the GPU; queries default to the table itself (what knn.py does).
    -> (indices int64 [nq, k], distances fp32 [nq, k]), neighbours in ascending distance (ties by row index)."""
    if not table.is_cuda:
        raise _lib.NcxError("knn needs the feature table on the GPU (no CPU fallback)")
    if table.dtype != torch.float32 or table.dim() != 2:
        raise ValueError("table must be a [n, dv] float32 tensor")
    table = table.contiguous()
    q = table if queries is None else queries.to(table.device, torch.float32).contiguous()
    n, dv = table.shape
    if q.dim() != 2 or q.shape[1] != dv:
        raise ValueError("queries must be [nq, %d]" % dv)
    if not 1 <= k <= min(n, 120):
        raise ValueError("k must be in 1..min(n, 120)")
    nq = q.shape[0]
    block_rows = max(1, min(block_rows, nq))
    L = _lib.lib()
    need = L.ncx_knn_workspace_bytes(n, block_rows) + 256
    ws = torch.empty(need, dtype=torch.uint8, device=table.device)
    base = (ws.data_ptr() + 255) // 256 * 256
    idx = torch.empty(nq, k, dtype=torch.int64, device=table.device)
    dist = torch.empty(nq, k, dtype=torch.float32, device=table.device)
    for i in range(0, nq, block_rows):
        m = min(block_rows, nq - i)
        qb, ib, db = q[i:i + m], idx[i:i + m], dist[i:i + m]
        _lib.check(L.ncx_knn(_ptr(table, torch.float32, "table"), n, C.c_void_p(qb.data_ptr()), m, dv, k, 1 if i else 0,
                             C.c_void_p(base), C.c_size_t(need - 256), C.c_void_p(ib.data_ptr()), C.c_void_p(db.data_ptr()),
                             _stream()), "ncx_knn")
    return idx, dist
