"""The PRODUCTION load path of the drop-in EmbeddingModel — what a user of the reference gets when they pass no weights:
``AutoTokenizer.from_pretrained`` + ``AutoModel.from_pretrained`` (minivectordb/embedding_model.py:55-60), an HF config
OBJECT (not a dict), checkpoint key names as `state_dict()` spells them, `pad_token_id` -> XLM-R position offset, and a
real HF tokenizer's output handed to the encoder (minivectordb_amd/embedding_model.py: load_alternative_model,
GpuEncoder.__init__, _tokenize).  The released checkpoints cannot be downloaded here, so a seeded model of the same
architecture and a trained-at-test-time tokenizer are `save_pretrained` into a local directory and loaded back through
`model_path=`; the expected embedding is transformers' own forward on the CPU + the reference's pooling."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TEXTS = ["i like dogs", "a much longer sentence about vector databases and the embeddings they store", "x"]


def _save_model(tmp_path, kind, tok):
    import torch
    from transformers import BertConfig, BertModel, XLMRobertaConfig, XLMRobertaModel
    torch.manual_seed({"small": 3, "large": 4, "bgem3": 5}[kind])
    if kind == "small":  # multilingual-e5-small is a BertModel (model card): its real widths
        cfg = BertConfig(vocab_size=tok.vocab_size + 7, hidden_size=384, num_hidden_layers=12, num_attention_heads=12,
                         intermediate_size=1536, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                         pad_token_id=1)
        model = BertModel(cfg)
    else:  # multilingual-e5-large / bge-m3 are XLMRobertaModels: a small one of that architecture (heads of 64)
        cfg = XLMRobertaConfig(vocab_size=tok.vocab_size + 7, hidden_size=256, num_hidden_layers=3, num_attention_heads=4,
                               intermediate_size=512, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                               pad_token_id=1, bos_token_id=0, eos_token_id=2)
        model = XLMRobertaModel(cfg)
    with torch.no_grad():  # HF's init zeroes every bias and sets LayerNorm to (1, 0): perturb them, or bias bugs pass
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
            elif "LayerNorm.weight" in name:
                p.add_(0.2 * torch.randn_like(p))
            elif name.endswith("dense.weight") or "self." in name:
                p.mul_(3.0)  # 0.02-sigma weights make every layer a near-identity
    model.eval()
    d = tmp_path / kind
    model.save_pretrained(str(d))
    tok.save_pretrained(str(d))
    return model


@pytest.mark.parametrize("kind", ["small", "large", "bgem3"])
def test_from_pretrained_directory(gpu, tmp_path, kind):
    import torch
    import torch.nn.functional as F
    from minivectordb_amd import AlternativeModel, EmbeddingModel
    from spm_tokenizer import hf_fast_tokenizer
    tok = hf_fast_tokenizer()
    hf = _save_model(tmp_path, kind, tok)
    m = EmbeddingModel(use_quantized_onnx_model=False, alternative_model=AlternativeModel(kind),
                       model_path=str(tmp_path / kind))
    assert m.tokenizer is not None and type(m.tokenizer).__name__ != "SpmXlmrTokenizer"
    assert m.model.cfg.position_offset == (0 if kind == "small" else 2)       # pad_token_id + 1 for XLM-R, 0 for BERT
    assert m.model.cfg.pooling == (1 if kind == "bgem3" else 0)
    for text in TEXTS:
        got = m.extract_embeddings(text)
        assert isinstance(got, list) and len(got) == hf.config.hidden_size
        # the reference, statement by statement (embedding_model.py:62-71 / :73-79), on transformers' CPU forward
        prompt = text if kind == "bgem3" else f"passage {text}"
        batch = tok([prompt], max_length=512, padding=True, truncation=True, return_tensors="pt")
        with torch.no_grad():
            out = hf(**batch)
        if kind == "bgem3":
            want = F.normalize(out.last_hidden_state[:, 0], p=2, dim=1)
        else:
            last = out.last_hidden_state.masked_fill(~batch["attention_mask"][..., None].bool(), 0.0)
            want = F.normalize(last.sum(dim=1) / batch["attention_mask"].sum(dim=1)[..., None], p=2, dim=1)
        np.testing.assert_allclose(got, want[0].numpy(), atol=2e-5, rtol=0)
    if kind != "bgem3":  # right-padded batch through the real tokenizer: row i == the single-sentence call
        rows = m.extract_embeddings_batch(TEXTS)
        for i, text in enumerate(TEXTS):
            np.testing.assert_allclose(rows[i], m.extract_embeddings(text), atol=3e-6, rtol=0)
