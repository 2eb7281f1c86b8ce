"""The text -> token ids -> embedding path of the drop-in EmbeddingModel (minivectordb/embedding_model.py:62-79):
prompt prefix, truncation at 512, XLM-R id layout and position ids with right padding, CLS pooling for bge-m3 —
with a SentencePiece model trained at test time (tests/spm_tokenizer.py) standing in for the released tokenizer
files.  CPU part: what `_tokenize` hands the encoder.  GPU part: extract_embeddings end to end against
transformers' own model on the same ids."""
import numpy as np
import pytest

from oracle import encoder as E
from spm_tokenizer import SpmXlmrTokenizer


@pytest.fixture(scope="module")
def tok():
    pytest.importorskip("sentencepiece")
    return SpmXlmrTokenizer()


def _model_cpu(tok, alt):
    """EmbeddingModel with the encoder construction skipped (no GPU here): only the host text path is exercised."""
    from minivectordb_amd import EmbeddingModel
    m = EmbeddingModel.__new__(EmbeddingModel)
    m.use_quantized_onnx_model = False
    m.alternative_model = alt
    m.tokenizer = tok
    m.model = None
    return m


def test_e5_prompt_prefix_truncation_and_layout(tok):
    from minivectordb_amd import AlternativeModel
    m = _model_cpu(tok, AlternativeModel.small)
    ids, mask = m._tokenize(["i like dogs"])
    want = [0] + tok.encode_pieces("passage i like dogs") + [2]
    assert ids.dtype == np.int32 and ids[0].tolist() == want and mask[0].all()
    assert tok.encode_pieces("passage i like dogs")[:len(tok.encode_pieces("passage"))] == tok.encode_pieces("passage")
    # batch: right padding with <pad> = 1, mask 0 there
    ids, mask = m._tokenize(["i like dogs", "x"])
    n1 = int(mask[1].sum())
    assert n1 < ids.shape[1] and (ids[1, n1:] == 1).all() and not mask[1, n1:].any()
    assert ids[1, 0] == 0 and ids[1, n1 - 1] == 2
    # max_length = 512 with truncation: <s> + 510 pieces + </s> (embedding_model.py:64)
    long = " ".join(["vector databases store embeddings"] * 400)
    ids, mask = m._tokenize([long])
    assert ids.shape == (1, 512) and ids[0, 0] == 0 and ids[0, -1] == 2 and mask.all()
    assert ids[0, 1:-1].tolist() == tok.encode_pieces("passage " + long)[:510]


@pytest.mark.gpu
@pytest.mark.parametrize("alt,cfg_name,wseed", [("small", "e5-small-dims", 10), ("large", "xlmr-large-dims", 12),
                                                 ("bgem3", "xlmr-large-dims", 12)])
def test_extract_embeddings_end_to_end(gpu, tok, alt, cfg_name, wseed):
    import torch
    from minivectordb_amd import AlternativeModel, EmbeddingModel
    cfg = E.make_config(cfg_name)
    assert tok.vocab_size <= cfg["vocab_size"]
    w = E.make_weights(cfg, wseed)
    m = EmbeddingModel(use_quantized_onnx_model=False, alternative_model=AlternativeModel(alt),
                       state_dict={k: torch.from_numpy(v) for k, v in w.items()}, config=cfg, tokenizer=tok)
    texts = ["i like dogs", "a much longer sentence about vector databases and the embeddings they store", "x",
             " ".join(["passage retrieval with dense vectors"] * 150)]   # the last one hits the 512 cap
    for t in texts:
        got = m.extract_embeddings(t)
        assert isinstance(got, list) and len(got) == cfg["hidden_size"]
        batch = tok([t if alt == "bgem3" else f"passage {t}"], max_length=512, padding=True, truncation=True)
        ids, mask = batch["input_ids"].astype(np.int32), batch["attention_mask"].astype(np.int32)
        assert ids.shape[1] <= 512
        hidden, mean_emb = E.hf_forward(cfg, w, ids, mask)   # transformers' BertModel / XLMRobertaModel on CPU
        if alt == "bgem3":   # FlagEmbedding dense_vecs: L2-normalised CLS state (embedding_model.py:74-78)
            want = hidden[0, 0] / np.linalg.norm(hidden[0, 0])
        else:
            want = mean_emb[0]
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
    if alt != "bgem3":
        # right-padded batch (XLM-R position ids skip the padding): row i == the single-sentence call
        rows = m.extract_embeddings_batch(texts[:3])
        for i, t in enumerate(texts[:3]):
            np.testing.assert_allclose(rows[i], m.extract_embeddings(t), atol=3e-6, rtol=0)
