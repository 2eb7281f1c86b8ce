"""CPU oracle of the e5 embed path.  TEST INFRASTRUCTURE ONLY (never imported by the product).

The reference computes text embeddings as (minivectordb/embedding_model.py:62-71)
    outputs = AutoModel(...)(**batch_dict); average_pool(last_hidden_state, mask); F.normalize(p=2)
where AutoModel is transformers' BertModel for multilingual-e5-small (XLMRobertaModel for e5-large).
The arithmetic lives in the third-party `transformers` + `torch` packages (reference pins
transformers==4.37.2, requirements.txt:5; this image has 5.x — same maths).  Two checkers:
  * hf_forward      : the real transformers model on CPU with the given weights (the library the
                      reference itself calls) + the reference's average_pool / F.normalize
  * numpy_forward   : a float64 numpy restatement of modeling_bert.py's op order (adjudicator)
Pretrained e5 weights are not available offline, so weights are seeded random tensors with
NON-trivial biases and LayerNorm affine parameters (HF's default init zeroes them, which would hide
bias bugs).  PARITY is therefore pinned on architecture + arithmetic, not on the released weights.
"""
import numpy as np

CONFIGS = {
    # name: (model_type, vocab, hidden, layers, heads, intermediate, max_pos)
    "tiny": ("bert", 100, 64, 2, 2, 128, 64),
    "hd64": ("bert", 120, 128, 2, 2, 256, 64),
    "e5-small-dims": ("bert", 1000, 384, 12, 12, 1536, 512),
    "xlmr-tiny": ("xlm-roberta", 150, 128, 2, 4, 256, 80),
    # multilingual-e5-large / bge-m3 widths (XLM-R large: H 1024, 16 heads of 64, FFN 4096), 2 layers
    "xlmr-large-dims": ("xlm-roberta", 300, 1024, 2, 16, 4096, 600),
    # widths that are multiples of 32 but not of 64 (three heads of 32, FFN 160): rows that do not fill every lane slot
    # of the row kernels, GEMM column tiles that end inside a 128-column block
    "h96": ("bert", 90, 96, 2, 3, 160, 64),
}


def make_config(name):
    mt, v, h, l, nh, f, p = CONFIGS[name]
    return {"model_type": mt, "vocab_size": v, "hidden_size": h, "num_hidden_layers": l,
            "num_attention_heads": nh, "intermediate_size": f, "max_position_embeddings": p,
            "type_vocab_size": 2 if mt == "bert" else 1, "layer_norm_eps": 1e-12 if mt == "bert" else 1e-5,
            "hidden_act": "gelu", "pad_token_id": 0 if mt == "bert" else 1}


def weight_names(cfg):
    names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
             "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
    per = ["attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
           "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
           "attention.output.dense.weight", "attention.output.dense.bias", "attention.output.LayerNorm.weight",
           "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
           "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]
    for i in range(cfg["num_hidden_layers"]):
        names += [f"encoder.layer.{i}.{p}" for p in per]
    return names


def make_weights(cfg, seed):
    """Deterministic (numpy RandomState) weights keyed like an HF state_dict."""
    rs = np.random.RandomState(seed)
    H, F = cfg["hidden_size"], cfg["intermediate_size"]
    shapes = {"embeddings.word_embeddings.weight": (cfg["vocab_size"], H),
              "embeddings.position_embeddings.weight": (cfg["max_position_embeddings"], H),
              "embeddings.token_type_embeddings.weight": (cfg["type_vocab_size"], H)}
    out = {}
    for name in weight_names(cfg):
        if name in shapes:
            w = rs.standard_normal(shapes[name]) * 0.5
        elif name.endswith("LayerNorm.weight"):
            w = 1.0 + 0.2 * rs.standard_normal(H)
        elif name.endswith("LayerNorm.bias"):
            w = 0.1 * rs.standard_normal(H)
        elif name.endswith("intermediate.dense.weight"):
            w = rs.standard_normal((F, H)) * (1.5 / np.sqrt(H))
        elif name.endswith("intermediate.dense.bias"):
            w = 0.1 * rs.standard_normal(F)
        elif name.endswith("output.dense.weight") and "attention" not in name:
            w = rs.standard_normal((H, F)) * (1.0 / np.sqrt(F))
        elif name.endswith(".weight"):
            w = rs.standard_normal((H, H)) * (1.5 / np.sqrt(H))
        else:
            w = 0.1 * rs.standard_normal(H)
        out[name] = w.astype(np.float32)
    return out


def make_inputs(cfg, B, S, seed, ragged=True):
    rs = np.random.RandomState(seed)
    lo = 2 if cfg["model_type"] != "bert" else 1
    ids = rs.randint(lo, cfg["vocab_size"], size=(B, S)).astype(np.int32)
    lens = rs.randint(1, S + 1, size=B) if ragged else np.full(B, S)
    lens[0] = S
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int32)
    ids = np.where(mask == 1, ids, cfg["pad_token_id"]).astype(np.int32)
    return ids, mask


def pool_normalize(hidden, mask):
    """reference average_pool (embedding_model.py:50-53) + F.normalize(p=2, dim=1, eps=1e-12) (:70)"""
    import torch
    import torch.nn.functional as F
    h = torch.as_tensor(hidden)
    m = torch.as_tensor(mask).long()
    last_hidden = h.masked_fill(~m[..., None].bool(), 0.0)
    emb = last_hidden.sum(dim=1) / m.sum(dim=1)[..., None]
    return F.normalize(emb, p=2, dim=1).numpy()


def hf_forward(cfg, weights, ids, mask):
    """transformers' own model on CPU (fp32).  Returns (last_hidden_state [B,S,H], embeddings [B,H])."""
    import torch
    if cfg["model_type"] == "bert":
        from transformers import BertConfig as C, BertModel as M
    else:
        from transformers import XLMRobertaConfig as C, XLMRobertaModel as M
    kw = {k: v for k, v in cfg.items() if k != "model_type"}
    conf = C(**kw)
    try:
        conf._attn_implementation = "eager"
    except Exception:
        pass
    model = M(conf, add_pooling_layer=False)
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    model.eval()
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids).long(), attention_mask=torch.from_numpy(mask).long())
    hidden = out.last_hidden_state.numpy()
    return hidden, pool_normalize(hidden, mask)


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def numpy_forward(cfg, weights, ids, mask):
    """float64 restatement (modeling_bert.py op order).  Returns (hidden [B,S,H], embeddings [B,H])."""
    from scipy.special import erf
    W = {k: v.astype(np.float64) for k, v in weights.items()}
    B, S = ids.shape
    H, nh = cfg["hidden_size"], cfg["num_attention_heads"]
    hd = H // nh
    eps = cfg["layer_norm_eps"]
    if cfg["model_type"] == "bert":
        pos = np.broadcast_to(np.arange(S)[None, :], (B, S))
    else:
        pad = cfg["pad_token_id"]
        pos = (np.cumsum(mask, axis=1) * mask + pad).astype(np.int64)
    x = (W["embeddings.word_embeddings.weight"][ids] + W["embeddings.token_type_embeddings.weight"][0]
         + W["embeddings.position_embeddings.weight"][pos])
    x = _ln(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], eps)
    neg = np.where(mask[:, None, None, :] == 1, 0.0, -np.inf)
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        lin = lambda t, n: t @ W[p + n + ".weight"].T + W[p + n + ".bias"]  # noqa: E731
        split = lambda t: t.reshape(B, S, nh, hd).transpose(0, 2, 1, 3)  # noqa: E731
        q, k, v = split(lin(x, "attention.self.query")), split(lin(x, "attention.self.key")), split(
            lin(x, "attention.self.value"))
        s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(hd) + neg
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr /= pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, S, H)
        x = _ln(lin(ctx, "attention.output.dense") + x, W[p + "attention.output.LayerNorm.weight"],
                W[p + "attention.output.LayerNorm.bias"], eps)
        h = lin(x, "intermediate.dense")
        h = 0.5 * h * (1.0 + erf(h / np.sqrt(2.0)))
        x = _ln(lin(h, "output.dense") + x, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)
    m = mask.astype(np.float64)
    emb = (x * m[..., None]).sum(1) / m.sum(1)[:, None]
    emb = emb / np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-12)
    return x, emb
