"""One rank of the N > 1 scoring path with the HIP scorer (launched by test_multirank_gpu.py with RANK / WORLD_SIZE / MASTER_* set;
all ranks share GPU 0 on the single-GPU test box and exchange over gloo).  Every rank scores its contiguous query block of a
testB-like ragged set, the shards are all-gathered with static counts; rank 0 also scores the WHOLE set alone and writes whether the
gathered (query id, product id, score) triples are identical to that, bit for bit."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, sharding, synth, weights  # noqa: E402
from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import ZkConfig  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ZkConfig(layers=2, vocab=4096, inter=1024)
    w = weights.make_weights(cfg)
    NQ = 25      # ~480 pairs: whole set and shards all run the same GEMM engine (rows < 16384, gemm_dispatch.hip) => bitwise comparable
    whole = synth.make_pairs(NQ, (8, 30), vocab=cfg.vocab, tag="/mr")
    qop = whole.query_id - whole.query_id.min()
    counts = sharding.shard_sizes(qop, NQ, world)
    lo, hi = sharding.query_block(NQ, world, rank)
    a, e = sharding.pair_slice_for_queries(qop, lo, hi)
    mine = whole.take(slice(a, e))
    s = scorers.ZkScorer(cfg, w)
    _, probs = scorers.score_batch(s, synth.zk_batch(mine, cfg.text_len))
    score = probs[:, 1].contiguous().cpu()
    all_s, all_q, all_p = sharding.gather_scores(score, torch.as_tensor(mine.query_id), torch.as_tensor(mine.product_id), counts=counts)
    ok = None
    if rank == 0:
        _, pw = scorers.score_batch(s, synth.zk_batch(whole, cfg.text_len))
        ref = pw[:, 1].contiguous().cpu()
        ok = bool(torch.equal(all_s, ref) and np.array_equal(all_q.numpy(), whole.query_id) and np.array_equal(all_p.numpy(), whole.product_id)
                  and len(set(counts)) > 1)
        json.dump({"ok": ok, "pairs": int(whole.n), "counts": counts}, open(sys.argv[1], "w"))
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
