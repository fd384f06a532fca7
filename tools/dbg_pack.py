import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import small_cfg
from kddcup_2020_multimodalitiesrecall_2nd_place_amd import scorers, synth, weights
cfg = small_cfg("zk")
w = weights.make_weights(cfg)
ps = synth.make_pairs(4, (3, 5), vocab=cfg.vocab, tag="/packedge")
b = synth.batch_for(cfg, ps)
b["len_query_"][0] = 0; b["num_boxes"][0] = 0
b["len_query_"][1] = 0
b["num_boxes"][2] = 0
b["len_query_"][3] = 20; b["num_boxes"][3] = 13
print("len", b["len_query_"], "nb", b["num_boxes"])
def run(**kw):
    s = scorers.make_scorer(cfg, w, **kw); l, p = scorers.score_batch(s, b); torch.cuda.synchronize(); o = l.cpu().numpy(); s.close(); return o
d1 = run(pack_tokens=False); d2 = run(pack_tokens=False); p1 = run(pack_tokens=True); p2 = run(pack_tokens=True)
print("dense vs dense", np.abs(d1-d2).max(1))
print("packed vs packed", np.abs(p1-p2).max(1))
print("packed vs dense", np.abs(p1-d1).max(1))
for stop in (0, 1, 2):
    sd = scorers.make_scorer(cfg, w, stop_after=stop, pack_tokens=False); scorers.score_batch(sd, b); hd = sd.read_hidden(ps.n*30).cpu().numpy().reshape(ps.n, 30, 768); sd.close()
    sp = scorers.make_scorer(cfg, w, stop_after=stop, pack_tokens=True); scorers.score_batch(sp, b); hp = sp.read_hidden(ps.n*30).cpu().numpy(); sp.close()
    # packed row offsets
    off = 0
    for i in range(ps.n):
        lq = min(max(int(b["len_query_"][i]),0),20); nb = min(max(int(b["num_boxes"][i]),0),10)
        if lq + nb == 0: nt, nv = 20, 10
        else: nt, nv = max(lq,1), nb
        rows = list(range(nt)) + [20+j for j in range(nv)]
        dd = np.abs(hp[off:off+len(rows)] - hd[i, rows]).max(1)
        print("stop", stop, "pair", i, "rows", len(rows), "maxdiff", dd.max(), "cls", dd[0], "argmax row", rows[int(dd.argmax())])
        off += len(rows)
