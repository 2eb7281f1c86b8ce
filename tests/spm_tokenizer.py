"""Test helper: an HF-style tokenizer (the call signature minivectordb/embedding_model.py:64 uses) over a
SentencePiece unigram model TRAINED AT TEST TIME — the released e5 / bge-m3 tokenizer files are not available
offline, and transformers 5.x no longer loads a bare .model file into XLMRobertaTokenizer.

Restates XLM-R's id layout (transformers/models/xlm_roberta/tokenization_xlm_roberta.py): <s> = 0, <pad> = 1,
</s> = 2, <unk> = 3, SentencePiece piece p -> p + 1 (the "fairseq offset"; SentencePiece's own <unk> = 0 -> 3);
a sequence is <s> pieces </s>; truncation keeps the first max_length - 2 pieces; padding on the right with <pad>.
multilingual-e5-small/large and bge-m3 all use this tokenizer."""
import io

import numpy as np

CORPUS = [
    "the quick brown fox jumps over the lazy dog", "vector databases store embeddings and metadata",
    "passage retrieval with dense vectors", "hello world this is a test of the tokenizer",
    "i like dogs and cats", "ein zwei drei vier fuenf", "bonjour le monde", "a much longer sentence about vector databases",
    "queries and passages share one encoder", "cosine similarity of normalised embeddings",
]


def train(vocab_size=200):
    import sentencepiece as spm
    buf = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(CORPUS * 20), model_writer=buf, vocab_size=vocab_size,
                                   model_type="unigram", character_coverage=1.0, hard_vocab_limit=False,
                                   minloglevel=2)
    return spm.SentencePieceProcessor(model_proto=buf.getvalue())


class SpmXlmrTokenizer:
    bos_token_id, pad_token_id, eos_token_id, unk_token_id = 0, 1, 2, 3

    def __init__(self, sp=None):
        self.sp = sp or train()
        self.vocab_size = self.sp.get_piece_size() + 2  # + fairseq offset + <mask>

    def encode_pieces(self, text):
        return [p + 1 if p != 0 else self.unk_token_id for p in self.sp.encode(text)]

    def __call__(self, texts, max_length=512, padding=True, truncation=True, return_tensors="np"):
        rows = []
        for t in texts:
            pieces = self.encode_pieces(t)
            if truncation:
                pieces = pieces[:max_length - 2]
            rows.append([self.bos_token_id] + pieces + [self.eos_token_id])
        S = max(len(r) for r in rows)
        ids = np.full((len(rows), S), self.pad_token_id, np.int64)
        mask = np.zeros((len(rows), S), np.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
            mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}


def hf_fast_tokenizer(vocab_size=200):
    """A REAL HF tokenizer object that `save_pretrained` / `AutoTokenizer.from_pretrained` round-trip offline: a
    `tokenizers` Unigram model trained on CORPUS with XLM-R's special-token layout (<s> = 0, <pad> = 1, </s> = 2,
    <unk> = 3) and its `<s> A </s>` template, wrapped as PreTrainedTokenizerFast.  Used by the production-load-path
    test (tests/test_load_path_gpu.py); the released e5 / bge-m3 tokenizer files are not available offline."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.Unigram())
    tok.pre_tokenizer = pre_tokenizers.Metaspace()
    tok.decoder = decoders.Metaspace()
    trainer = trainers.UnigramTrainer(vocab_size=vocab_size, special_tokens=["<s>", "<pad>", "</s>", "<unk>"],
                                      unk_token="<unk>", show_progress=False)
    tok.train_from_iterator(CORPUS * 20, trainer)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                       special_tokens=[("<s>", 0), ("</s>", 2)])
    return PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", pad_token="<pad>",
                                   unk_token="<unk>", cls_token="<s>", sep_token="</s>")
