"""CPU suite: the N > 1 path (query-sharded scoring + score all-gather) on gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kddcup_2020_multimodalitiesrecall_2nd_place_amd import sharding


def test_query_blocks_partition_every_query_once():
    for n, w in ((1000, 8), (994, 8), (7, 8), (496, 3), (5, 1)):
        blocks = [sharding.query_block(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_pair_slice_keeps_candidate_sets_whole():
    per_q = np.array([3, 1, 4, 2, 5])
    qop = np.repeat(np.arange(5), per_q)
    lo, hi = sharding.query_block(5, 2, 1)
    s, e = sharding.pair_slice_for_queries(qop, lo, hi)
    assert (qop[s:e] >= lo).all() and (qop[s:e] < hi).all() and e - s == per_q[lo:hi].sum()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_q = 7
    per_q = np.array([3, 1, 4, 2, 5, 2, 3])
    qop = np.repeat(np.arange(n_q), per_q)
    lo, hi = sharding.query_block(n_q, world, rank)
    s, e = sharding.pair_slice_for_queries(qop, lo, hi)
    # "score" of global pair i is f(i): every rank computes only its own shard
    idx = torch.arange(s, e)
    scores = (idx.float() * 0.5 + 1.0)
    qid = torch.as_tensor(qop[s:e]).long()
    pid = idx.long() * 7
    all_s, all_q, all_p = sharding.gather_scores(scores, qid, pid)
    ok = (all_s.numel() == per_q.sum()
          and torch.equal(all_s, torch.arange(per_q.sum()).float() * 0.5 + 1.0)
          and torch.equal(all_q, torch.as_tensor(qop).long())
          and torch.equal(all_p, torch.arange(per_q.sum()).long() * 7))
    s2, _, _ = sharding.gather_scores(scores)
    ok = ok and torch.equal(s2, all_s)
    # static shard sizes (known to every rank up front): one collective per step, no size exchange
    counts = sharding.shard_sizes(qop, n_q, world)
    ok = ok and counts[rank] == e - s and sum(counts) == per_q.sum()
    s3, q3, p3 = sharding.gather_scores(scores, qid, pid, counts=counts)
    ok = ok and torch.equal(s3, all_s) and torch.equal(q3, all_q) and torch.equal(p3, all_p)
    # equal shards take the no-padding route
    eq = torch.full((4,), float(rank))
    s4, _, _ = sharding.gather_scores(eq, counts=[4] * world)
    ok = ok and torch.equal(s4, torch.arange(world).float().repeat_interleave(4))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_gather_scores_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))


def test_gather_is_identity_without_process_group():
    s = torch.arange(5).float()
    a, q, p = sharding.gather_scores(s)
    assert a is s and q is None and p is None
