"""Query-sharded data parallelism.

Every (query, candidate) pair is independent, so the path shards with no data-path collective:
contiguous QUERY blocks per rank (a query's <= 30 candidates stay together for local top-k / nDCG),
weights replicated.  The single exchange step is an all-gather of per-rank scores (fp32, plus
int64 query/product ids once) -- RCCL over xGMI on the GPU box (backend "nccl"), gloo in CPU tests.
The reference itself is single-GPU at inference (evaluate_normal.py:46, run_pretraining_predict_score.py:319-323).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def query_block(n_queries: int, world: int, rank: int):
    """Contiguous [lo, hi) block of query indices owned by ``rank`` (sizes differ by at most 1)."""
    q, r = divmod(n_queries, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pair_slice_for_queries(query_of_pair: np.ndarray, lo: int, hi: int):
    """Pairs are stored grouped by query (ascending); returns the [start, stop) pair range of queries lo..hi."""
    start = int(np.searchsorted(query_of_pair, lo, side="left"))
    stop = int(np.searchsorted(query_of_pair, hi, side="left"))
    return start, stop


def gather_scores(scores: torch.Tensor, query_id: torch.Tensor = None, product_id: torch.Tensor = None, group=None):
    """All-gather ragged per-rank score vectors.  Returns (scores, query_id, product_id) concatenated in
    rank order on every rank.  One collective for the scores (padded to the max shard) and, when ids
    are given, one more for the packed int64 ids."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return scores, query_id, product_id
    world = dist.get_world_size(group)
    n = torch.tensor([scores.numel()], device=scores.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = torch.zeros(m, device=scores.device, dtype=scores.dtype)
    pad[: scores.numel()] = scores.reshape(-1)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    all_scores = torch.cat([o[:c] for o, c in zip(out, counts)])
    all_q = all_p = None
    if query_id is not None and product_id is not None:
        ids = torch.zeros((m, 2), device=scores.device, dtype=torch.int64)
        ids[: scores.numel(), 0] = query_id
        ids[: scores.numel(), 1] = product_id
        outs = [torch.empty_like(ids) for _ in range(world)]
        dist.all_gather(outs, ids, group=group)
        cat = torch.cat([o[:c] for o, c in zip(outs, counts)])
        all_q, all_p = cat[:, 0], cat[:, 1]
    return all_scores, all_q, all_p
