"""Generate golden vectors from the REFERENCE ITSELF (run only in the build container).

Imports /root/reference/code/lxmert/src (PyTorch code, importable under torch 2.10 -- recipe in
SURVEY.md Appendix B), loads this build's seeded weights into the reference modules, runs the
reference forward on this build's synthetic pair sets and stores ONLY outputs (inputs and weights
are regenerated from their seeds by the tests).  Nothing from /root/reference is copied.

Produces:
  lxmert_full.npz        reference KDDModel (9/5/5 layers, full width) : logit, pooled for 16 pairs,
                         intermediates for the first 2 pairs
  lxmert_shallow.npz     reference KDDModel (2/1/2 layers, inter 1024, vocab 4096), 12 ragged pairs,
                         all intermediates
  bertlayer_tanh_S30.npz / _S40.npz
                         reference lxrt.modeling.BertLayer with a tanh-GELU callable as hidden_act
                         (modeling.py:398-401 accepts callables) driven with TF-named weights ->
                         pins the zk/lds encoder-layer restatement (S=30 masked, S=40 unmasked)

  tf_pins.npz            the pieces of the TF models (zk, lds) whose arithmetic the imported PyTorch reference SHARES, driven with
                         this build's TF-named seeded weights (kernels transposed as modeling.py:99-102 does):
                           pooler      lxrt.modeling.BertPooler (modeling.py:596-608 == pixelbert.py:258-266)
                           textemb     lxrt.modeling.BertEmbeddings on 20-token rows, token type 0 (modeling.py:269-297 == the
                                       TEXT rows of zk's / lds's embedding_output: LayerNorm is per row)
                           amcos       KDDModel.forward under args.task_match = args.task_amsloss = True (kdd_model.py:207-210):
                                       cosine logits x_norm @ w_norm == the core of model_triple.py:56-70 before margin and scale
                           stack12_S30 / stack12_S40
                                       12 reference BertLayers (tanh-GELU callable) chained, S = 30 masked / S = 40 unmasked
                                       == pixelbert.transformer_model at the depth zk / lds run it
  lxmert_fp32ckpt.npz    reference KDDModel (2/1/2, inter 1024, vocab 4096) holding fp32 weights that are NOT bf16-representable
                         (this build's seeded generator with bf16_matrices=False, i.e. a real checkpoint's situation), its
                         logits, and the key list + shapes of KDDModel.state_dict() -- the fixture of the end-to-end importer
                         test (state_dict -> weights.from_torch_state_dict -> precision auto = mode 3 -> HIP logits)

Usage:  python tests/golden/make_lxmert_golden.py [all|shallow|full|bertlayer|tfpins|fp32ckpt]
"""
import json
import math
import os
import shutil
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/code"
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")


def _enter_reference(vocab=None, inter=None):
    tmp = tempfile.mkdtemp(prefix="mms_golden_")
    os.makedirs(os.path.join(tmp, "user_data"))
    os.makedirs(os.path.join(tmp, "run"))
    cfg = json.load(open(os.path.join(REF, "user_data", "bert_config.json")))
    if vocab:
        cfg["vocab_size"] = vocab
    if inter:
        cfg["intermediate_size"] = inter
    json.dump(cfg, open(os.path.join(tmp, "user_data", "bert_config.json"), "w"))
    shutil.copy(os.path.join(REF, "user_data", "vocab.txt"), os.path.join(tmp, "user_data", "vocab.txt"))
    import torch
    torch.save({}, os.path.join(tmp, "user_data", "pytorch_model.bin"))
    os.chdir(os.path.join(tmp, "run"))
    sys.dont_write_bytecode = True
    if os.path.join(REF, "lxmert", "src") not in sys.path:
        sys.path.insert(0, os.path.join(REF, "lxmert", "src"))
    sys.argv = ["kdd.py"]
    return tmp


def _run_kdd(cfg, n_q, cands, tag, n_inter, out_name):
    import torch
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import synth, weights
    from param import args
    args.load = None
    args.llayers, args.xlayers, args.rlayers = cfg.l_layers, cfg.x_layers, cfg.r_layers
    from tasks.kdd_model import KDDModel
    torch.manual_seed(0)
    m = KDDModel().eval()
    w = weights.make_lxmert_weights(cfg)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("cls.") or k == "logit_W" for k in missing), missing
    ps = synth.make_pairs(n_q, cands, vocab=cfg.vocab, tag=tag)
    b = synth.lxmert_batch(ps, cfg.text_len)
    inter = {}
    enc = m.lxrt_encoder.model.bert.encoder

    def grab(name, pick=lambda o: o):
        def hook(_mod, _inp, out):
            inter[name] = pick(out)
        return hook
    m.lxrt_encoder.model.bert.embeddings.register_forward_hook(
        lambda _m, _i, out: inter.setdefault("lang_emb", out) if out.dim() == 3 and out.shape[1] == cfg.text_len else None)
    enc.visn_fc.register_forward_hook(grab("visn_emb"))
    enc.layer[cfg.l_layers - 1].register_forward_hook(grab("lang_l"))
    enc.r_layers[cfg.r_layers - 1].register_forward_hook(grab("visn_r"))
    for i in range(cfg.x_layers):
        enc.x_layers[i].register_forward_hook(grab("lang_x%d" % i, lambda o: o[0]))
        enc.x_layers[i].register_forward_hook(grab("visn_x%d" % i, lambda o: o[1]))
    m.lxrt_encoder.model.bert.pooler.register_forward_hook(grab("pooled"))
    t = lambda a, dt: torch.tensor(a, dtype=dt)
    with torch.no_grad():
        _, _, logit = m(t(b["input_ids"], torch.long), t(b["boxes_label_input_ids"], torch.long), None,
                        t(b["input_mask"], torch.long), None, t(b["boxes_label_input_mask"], torch.long),
                        t(b["feats"], torch.float), t(b["boxes"], torch.float),
                        t(b["visual_attention_mask"], torch.float))
    save = {"logit": logit.numpy(), "pooled": inter["pooled"].numpy()}
    for k, v in inter.items():
        if k != "pooled":
            save[k] = v.numpy()[:n_inter]
    meta = dict(l_layers=cfg.l_layers, r_layers=cfg.r_layers, x_layers=cfg.x_layers, vocab=cfg.vocab,
                inter=cfg.inter, n_queries=n_q, cands=list(cands) if not isinstance(cands, int) else cands,
                tag=tag, source="reference code/lxmert/src tasks.kdd_model.KDDModel, torch %s" % torch.__version__)
    np.savez_compressed(os.path.join(OUT, out_name), meta=json.dumps(meta), **save)
    print(out_name, {k: v.shape for k, v in save.items()})


def _run_bertlayer(S, masked, out_name):
    import torch
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import weights
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import ZkConfig
    from lxrt.modeling import BertConfig, BertLayer

    def gelu_tanh(x):
        return x * 0.5 * (1.0 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))
    conf = BertConfig(vocab_size_or_config_json_file=100, hidden_size=768, num_hidden_layers=1,
                      num_attention_heads=12, intermediate_size=3072, hidden_act=gelu_tanh)
    layer = BertLayer(conf).eval()
    w = weights.make_zk_weights(ZkConfig(layers=1, vocab=128))
    p = "bert/encoder/layer_0"
    T = lambda n: torch.from_numpy(np.ascontiguousarray(w[n].T))
    V = lambda n: torch.from_numpy(w[n])
    sd = {}
    for n in ("query", "key", "value"):
        sd["attention.self.%s.weight" % n] = T("%s/attention/self/%s/kernel" % (p, n))
        sd["attention.self.%s.bias" % n] = V("%s/attention/self/%s/bias" % (p, n))
    sd["attention.output.dense.weight"] = T(p + "/attention/output/dense/kernel")
    sd["attention.output.dense.bias"] = V(p + "/attention/output/dense/bias")
    sd["attention.output.LayerNorm.weight"] = V(p + "/attention/output/LayerNorm/gamma")
    sd["attention.output.LayerNorm.bias"] = V(p + "/attention/output/LayerNorm/beta")
    sd["intermediate.dense.weight"] = T(p + "/intermediate/dense/kernel")
    sd["intermediate.dense.bias"] = V(p + "/intermediate/dense/bias")
    sd["output.dense.weight"] = T(p + "/output/dense/kernel")
    sd["output.dense.bias"] = V(p + "/output/dense/bias")
    sd["output.LayerNorm.weight"] = V(p + "/output/LayerNorm/gamma")
    sd["output.LayerNorm.bias"] = V(p + "/output/LayerNorm/beta")
    layer.load_state_dict(sd, strict=True)
    B = 3
    x = weights.normal("bertlayer/x/S%d" % S, (B, S, 768), 20200823)
    if masked:
        keep = np.array([S, S - 7, 5])
        mask = (np.arange(S)[None, :] < keep[:, None]).astype(np.float32)
        add = torch.from_numpy((1.0 - mask) * -10000.0)[:, None, None, :]
    else:
        mask = np.ones((B, S), np.float32)
        add = None
    with torch.no_grad():
        y = layer(torch.from_numpy(x), add).numpy()
    np.savez_compressed(os.path.join(OUT, out_name), y=y, mask=mask,
                        meta=json.dumps(dict(S=S, masked=masked, x_seed_name="bertlayer/x/S%d" % S,
                                             source="reference lxrt.modeling.BertLayer(hidden_act=tanh-gelu)")))
    print(out_name, y.shape)


def _tf_layer_sd(w, i):
    """TF-named encoder layer i of this build's zk weights -> state_dict of the reference's BertLayer."""
    import torch
    p = "bert/encoder/layer_%d" % i
    T = lambda n: torch.from_numpy(np.ascontiguousarray(w[n].T))
    V = lambda n: torch.from_numpy(w[n])
    sd = {}
    for n in ("query", "key", "value"):
        sd["attention.self.%s.weight" % n] = T("%s/attention/self/%s/kernel" % (p, n))
        sd["attention.self.%s.bias" % n] = V("%s/attention/self/%s/bias" % (p, n))
    sd["attention.output.dense.weight"] = T(p + "/attention/output/dense/kernel")
    sd["attention.output.dense.bias"] = V(p + "/attention/output/dense/bias")
    sd["attention.output.LayerNorm.weight"] = V(p + "/attention/output/LayerNorm/gamma")
    sd["attention.output.LayerNorm.bias"] = V(p + "/attention/output/LayerNorm/beta")
    sd["intermediate.dense.weight"] = T(p + "/intermediate/dense/kernel")
    sd["intermediate.dense.bias"] = V(p + "/intermediate/dense/bias")
    sd["output.dense.weight"] = T(p + "/output/dense/kernel")
    sd["output.dense.bias"] = V(p + "/output/dense/bias")
    sd["output.LayerNorm.weight"] = V(p + "/output/LayerNorm/gamma")
    sd["output.LayerNorm.bias"] = V(p + "/output/LayerNorm/beta")
    return sd


def _run_tf_pins(out_name):
    import torch
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import weights
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import ZkConfig
    from lxrt.modeling import BertConfig, BertEmbeddings, BertLayer, BertPooler

    def gelu_tanh(x):
        return x * 0.5 * (1.0 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))
    cfg = ZkConfig(layers=12, vocab=4096)
    w = weights.make_zk_weights(cfg)
    conf = BertConfig(vocab_size_or_config_json_file=cfg.vocab, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                      intermediate_size=3072, hidden_act=gelu_tanh, max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab,
                      hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    save = {}
    with torch.no_grad():
        # --- pooler ---
        pool = BertPooler(conf).eval()
        pool.load_state_dict({"dense.weight": torch.from_numpy(np.ascontiguousarray(w["bert/pooler/dense/kernel"].T)),
                              "dense.bias": torch.from_numpy(w["bert/pooler/dense/bias"])}, strict=True)
        xs = weights.normal("tfpins/pool_x", (6, 30, 768), 20200823)
        save["pooler"] = pool(torch.from_numpy(xs)).numpy()
        # --- text embedding rows ---
        emb = BertEmbeddings(conf).eval()
        emb.load_state_dict({"word_embeddings.weight": torch.from_numpy(w["bert/embeddings/word_embeddings"]),
                             "position_embeddings.weight": torch.from_numpy(w["bert/embeddings/position_embeddings"]),
                             "token_type_embeddings.weight": torch.from_numpy(w["bert/embeddings/token_type_embeddings"]),
                             "LayerNorm.weight": torch.from_numpy(w["bert/embeddings/LayerNorm/gamma"]),
                             "LayerNorm.bias": torch.from_numpy(w["bert/embeddings/LayerNorm/beta"])}, strict=True)
        ids = (106 + np.floor(weights.uniform01("tfpins/ids", 5 * 20, 20200823) * (cfg.vocab - 106))).astype(np.int64).reshape(5, 20)
        ids[:, 0] = 101
        ids[1, 7:] = 0
        save["textemb"] = emb(torch.from_numpy(ids), torch.zeros_like(torch.from_numpy(ids))).numpy()
        save["textemb_ids"] = ids
        # --- 12-layer stacks ---
        layers = []
        for i in range(12):
            l = BertLayer(conf).eval()
            l.load_state_dict(_tf_layer_sd(w, i), strict=True)
            layers.append(l)
        for S, masked in ((30, True), (40, False)):
            x = weights.normal("tfpins/stack_x/S%d" % S, (3, S, 768), 20200823)
            if masked:
                keep = np.array([S, S - 9, 4])
                mask = (np.arange(S)[None, :] < keep[:, None]).astype(np.float32)
                add = torch.from_numpy((1.0 - mask) * -10000.0)[:, None, None, :]
            else:
                mask = np.ones((3, S), np.float32)
                add = None
            y = torch.from_numpy(x)
            for l in layers:
                y = l(y, add)
            save["stack12_S%d" % S] = y.numpy()
            save["stack12_S%d_mask" % S] = mask
    np.savez_compressed(os.path.join(OUT, out_name), meta=json.dumps(dict(vocab=cfg.vocab, source="reference lxrt.modeling BertPooler / BertEmbeddings / BertLayer x12 (tanh-gelu callable)")), **save)
    print(out_name, {k: v.shape for k, v in save.items()})


def _run_amcos(out_name):
    """KDDModel under task_match + task_amsloss: logit = x_norm @ w_norm (kdd_model.py:204-210)."""
    import torch
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import synth, weights
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LxmertConfig
    from param import args
    cfg = LxmertConfig(l_layers=1, r_layers=1, x_layers=1, vocab=4096, inter=1024)
    args.load = None
    args.llayers, args.xlayers, args.rlayers = cfg.l_layers, cfg.x_layers, cfg.r_layers
    args.task_match = True
    args.task_amsloss = True
    from tasks.kdd_model import KDDModel
    torch.manual_seed(0)
    m = KDDModel().eval()
    w = weights.make_lxmert_weights(cfg)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    am = weights.normal("amcos/logit_W", (768, 2), 20200823, 0.05)
    sd["logit_W"] = torch.from_numpy(am)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("cls.") for k in missing), (missing, unexpected)
    ps = synth.make_pairs(3, (3, 5), vocab=cfg.vocab, tag="/amcos")
    b = synth.lxmert_batch(ps, cfg.text_len)
    t = lambda a, dt: torch.tensor(a, dtype=dt)
    pooled = {}
    m.lxrt_encoder.model.bert.pooler.register_forward_hook(lambda _m, _i, o: pooled.setdefault("p", o))
    with torch.no_grad():
        x_norm, _, logit = m(t(b["input_ids"], torch.long), t(b["boxes_label_input_ids"], torch.long), None, t(b["input_mask"], torch.long), None,
                             t(b["boxes_label_input_mask"], torch.long), t(b["feats"], torch.float), t(b["boxes"], torch.float),
                             t(b["visual_attention_mask"], torch.float))
    np.savez_compressed(os.path.join(OUT, out_name), pooled=pooled["p"].numpy(), x_norm=x_norm.numpy(), cos=logit.numpy(),
                        meta=json.dumps(dict(source="reference KDDModel.forward, args.task_match = args.task_amsloss = True", am_seed_name="amcos/logit_W")))
    print(out_name, logit.shape)


def _run_fp32ckpt(out_name):
    import torch
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd import synth, weights
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LxmertConfig
    from param import args
    cfg = LxmertConfig(l_layers=2, r_layers=1, x_layers=2, vocab=4096, inter=1024)
    args.load = None
    args.llayers, args.xlayers, args.rlayers = cfg.l_layers, cfg.x_layers, cfg.r_layers
    from tasks.kdd_model import KDDModel
    torch.manual_seed(0)
    m = KDDModel().eval()
    w = weights.make_lxmert_weights(cfg, bf16_matrices=False)      # fp32 values with live low mantissa bits
    assert weights.auto_precision(w) == 3
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected
    ps = synth.make_pairs(3, (3, 5), vocab=cfg.vocab, tag="/fp32ckpt")
    b = synth.lxmert_batch(ps, cfg.text_len)
    t = lambda a, dt: torch.tensor(a, dtype=dt)
    with torch.no_grad():
        x_norm, _, logit = m(t(b["input_ids"], torch.long), t(b["boxes_label_input_ids"], torch.long), None, t(b["input_mask"], torch.long), None,
                             t(b["boxes_label_input_mask"], torch.long), t(b["feats"], torch.float), t(b["boxes"], torch.float),
                             t(b["visual_attention_mask"], torch.float))
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}    # what torch.save(model.state_dict()) holds (kdd_model.py:131-152)
    meta = dict(l_layers=cfg.l_layers, r_layers=cfg.r_layers, x_layers=cfg.x_layers, vocab=cfg.vocab, inter=cfg.inter, tag="/fp32ckpt",
                n_queries=3, cands=[3, 5], state_dict_keys=keys,
                source="reference KDDModel holding make_lxmert_weights(cfg, bf16_matrices=False); torch %s" % torch.__version__)
    np.savez_compressed(os.path.join(OUT, out_name), logit=logit.numpy(), x_norm=x_norm.numpy(), meta=json.dumps(meta))
    print(out_name, logit.shape, len(keys), "state_dict keys")


def main():
    from kddcup_2020_multimodalitiesrecall_2nd_place_amd.config import LxmertConfig
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "shallow"):
        # the reference caches VISUAL_CONFIG / bert_config per process: one process per model size
        if which == "all":
            import subprocess
            for sub in ("shallow", "full", "bertlayer", "tfpins", "amcos", "fp32ckpt"):
                subprocess.check_call([sys.executable, os.path.abspath(__file__), sub])
            return
        cfg = LxmertConfig(l_layers=2, r_layers=1, x_layers=2, vocab=4096, inter=1024)
        tmp = _enter_reference(vocab=cfg.vocab, inter=cfg.inter)
        _run_kdd(cfg, 3, (3, 5), "/lx_shallow", 4, "lxmert_shallow.npz")
        shutil.rmtree(tmp, ignore_errors=True)
    elif which == "full":
        cfg = LxmertConfig()
        tmp = _enter_reference()
        _run_kdd(cfg, 2, 8, "/lx_full", 2, "lxmert_full.npz")
        shutil.rmtree(tmp, ignore_errors=True)
    elif which == "tfpins":
        tmp = _enter_reference()
        _run_tf_pins("tf_pins.npz")
        shutil.rmtree(tmp, ignore_errors=True)
    elif which == "amcos":
        tmp = _enter_reference(vocab=4096, inter=1024)
        _run_amcos("kdd_amcos.npz")
        shutil.rmtree(tmp, ignore_errors=True)
    elif which == "fp32ckpt":
        tmp = _enter_reference(vocab=4096, inter=1024)
        _run_fp32ckpt("lxmert_fp32ckpt.npz")
        shutil.rmtree(tmp, ignore_errors=True)
    elif which == "bertlayer":
        tmp = _enter_reference()
        _run_bertlayer(30, True, "bertlayer_tanh_S30.npz")
        _run_bertlayer(40, False, "bertlayer_tanh_S40.npz")
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
