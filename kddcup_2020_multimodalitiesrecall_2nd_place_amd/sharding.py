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


def shard_sizes(query_of_pair: np.ndarray, n_queries: int, world: int):
    """Pairs owned by each rank when ``n_queries`` queries (ids 0..n_queries-1, pairs grouped by ascending query) are cut into
    contiguous query blocks: static for a job, so every rank can compute the whole list once and no step needs a size exchange."""
    out = []
    for r in range(world):
        lo, hi = query_block(n_queries, world, r)
        a, b = pair_slice_for_queries(query_of_pair, lo, hi)
        out.append(b - a)
    return out


def gather_scores(scores: torch.Tensor, query_id: torch.Tensor = None, product_id: torch.Tensor = None, group=None, counts=None,
                  force_collective: bool = False):
    """All-gather ragged per-rank score vectors.  Returns (scores, query_id, product_id) concatenated in
    rank order on every rank.  One collective for the scores (padded to the max shard) and, when ids
    are given, one more for the packed int64 ids.

    ``counts``: the per-rank shard sizes (``shard_sizes``), static for a job.  With them a step is ONE collective and no
    host synchronisation; without them the sizes are exchanged first (an extra small all-gather and a host read).

    ``force_collective``: run the collective even in a world of one rank (where the result is the input) -- the same RCCL calls an 8-GPU job
    issues, on the one GPU a test box has (tests/test_rccl_world1_gpu.py; ``bench.py`` under a launcher with WORLD_SIZE=1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return scores, query_id, product_id
    if dist.get_world_size(group) == 1 and not force_collective:
        return scores, query_id, product_id
    world = dist.get_world_size(group)
    if counts is None:
        n = torch.tensor([scores.numel()], device=scores.device, dtype=torch.int64)
        cl = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(cl, n, group=group)
        counts = [int(c.item()) for c in cl]
    else:
        counts = [int(c) for c in counts]
        assert len(counts) == world and counts[dist.get_rank(group)] == scores.numel()
    m = max(counts)
    if all(c == m for c in counts):      # equal shards: gather straight into the output, no padding copies
        all_scores = torch.empty(world * m, device=scores.device, dtype=scores.dtype)
        dist.all_gather_into_tensor(all_scores, scores.reshape(-1).contiguous(), group=group)
    else:
        pad = torch.zeros(m, device=scores.device, dtype=scores.dtype)
        pad[: scores.numel()] = scores.reshape(-1)
        out = torch.empty(world * m, device=scores.device, dtype=scores.dtype)
        dist.all_gather_into_tensor(out, pad, group=group)
        all_scores = torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)])
    all_q = all_p = None
    if query_id is not None and product_id is not None:
        ids = torch.zeros((m, 2), device=scores.device, dtype=torch.int64)
        ids[: scores.numel(), 0] = query_id
        ids[: scores.numel(), 1] = product_id
        outs = torch.empty(world * m * 2, device=scores.device, dtype=torch.int64)
        dist.all_gather_into_tensor(outs, ids.reshape(-1), group=group)
        outs = outs.reshape(world, m, 2)
        cat = torch.cat([outs[r, :c] for r, c in enumerate(counts)])
        all_q, all_p = cat[:, 0], cat[:, 1]
    return all_scores, all_q, all_p
