"""EmbeddingModel — drop-in for ``minivectordb.embedding_model.EmbeddingModel`` whose transformer
forward pass, mean pooling and L2 normalisation run on an MI355X through libmvdb.so
(``mvdb_encoder_*``), instead of torch-CPU inside ``transformers``.

Reference behaviour kept (minivectordb/embedding_model.py):
  * constructor arguments and the ``AlternativeModel`` str-enum (:11-35), incl. the legacy
    ``e5_model_size`` keyword (:27-28);
  * e5 path (:62-71): prompt ``f'passage {text}'`` (no colon), ``max_length=512``, truncation,
    attention-masked mean pool, L2 normalise, result returned as a python ``list`` of floats;
  * ``extract_embeddings`` dispatch (:84-91).
PyTorch is used only as the container of the weights (state_dict -> device tensors) and the HF
tokenizer stays on the host, as in the reference.

BGE-M3 (:73-79) is served by the same encoder with XLM-R position ids and CLS pooling
(``dense_vecs`` = normalised first-token state); its sparse / ColBERT heads are not used by the
reference and are not built.

Not available in this build (SURVEY.md §8f "next"): the quantised ONNX USE model — its blob is
absent from the reference tree (.MISSING_LARGE_BLOBS) and cannot be restated.  Selecting it raises
``NotImplementedError`` instead of silently computing something else.

Extras for offline / batched use: ``model_path`` (local HF directory), or ``state_dict`` +
``config`` (+ ``tokenizer``) to inject weights; ``extract_embeddings_batch`` and ``encode_ids``.
"""
import ctypes
import os
from enum import Enum

import numpy as np


class AlternativeModel(str, Enum):
    small = "small"
    large = "large"
    bgem3 = "bgem3"


class GpuEncoder:
    """mvdb_encoder* + the torch tensors that own its weights."""

    def __init__(self, config, state_dict, device=0, pooling="mean"):
        import torch
        from . import _native
        self._native = _native
        lib = _native.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("minivectordb_amd.EmbeddingModel needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", device)
        get = (lambda k, d=None: config.get(k, d)) if isinstance(config, dict) else (
            lambda k, d=None: getattr(config, k, d))
        model_type = get("model_type", "bert")
        pad = get("pad_token_id", 0) or 0
        self.cfg = _native.EncoderCfg(
            vocab_size=get("vocab_size"), hidden=get("hidden_size"), layers=get("num_hidden_layers"),
            heads=get("num_attention_heads"), intermediate=get("intermediate_size"),
            max_positions=get("max_position_embeddings"), type_vocab=get("type_vocab_size", 2),
            position_offset=(pad + 1) if model_type in ("xlm-roberta", "roberta") else 0,
            ln_eps=float(get("layer_norm_eps", 1e-12)), pooling={"mean": 0, "cls": 1}[pooling])
        if get("hidden_act", "gelu") != "gelu":
            raise ValueError("only the erf-GELU activation of BERT/XLM-R is implemented")
        n = lib.mvdb_encoder_weight_count(ctypes.byref(self.cfg))
        if n < 0:
            _native.check(_native.ERR_ARG)
        # accept bare keys or a single model prefix ("bert.", "roberta.", "model.")
        keys = list(state_dict.keys())
        prefix = ""
        probe = "embeddings.word_embeddings.weight"
        if probe not in state_dict:
            hits = [k for k in keys if k.endswith(probe)]
            if not hits:
                raise KeyError(f"state_dict has no '{probe}'")
            prefix = hits[0][:-len(probe)]
        self._weights = []
        table = (ctypes.c_void_p * n)()
        for i in range(n):
            name = lib.mvdb_encoder_weight_name(ctypes.byref(self.cfg), i).decode()
            t = state_dict[prefix + name]
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            if t.data_ptr() % 16:  # the kernels read weights and biases in 16-byte pieces (a view into a larger tensor)
                t = t.clone()
            self._weights.append(t)
            table[i] = t.data_ptr()
        torch.cuda.synchronize(self.device)
        self._h = ctypes.c_void_p()
        _native.check(lib.mvdb_encoder_create(ctypes.byref(self.cfg), table, device, ctypes.byref(self._h)))
        self.hidden = self.cfg.hidden
        # arithmetic of the GEMMs when a call does not say: 2 = split-precision fp16 x 3 on the 16-bit matrix cores
        # (fp32-equivalent to ~2^-21: embeddings within 6e-7 of transformers' fp32 output, as close as the exact mode,
        # tests/test_encoder_gpu.py),
        # 0 = exact fp32 matrix cores
        self.default_compute = int(os.environ.get("MVDB_ENCODER_COMPUTE", "2"))

    def close(self):
        if getattr(self, "_h", None):
            self._native.lib().mvdb_encoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def walks(self, B, S):
        """True when a [B,S] forward is of the shape the one layer-walking launch serves (exact fp32 in both modes)."""
        return bool(self._native.lib().mvdb_encoder_walks(self._h, int(B), int(S)))

    def walk_stats(self):
        """Walking launches abandoned by their bounded waits, forwards the host entry re-ran on the per-op kernels, and forwards
        left on the per-op kernels before the next walking attempt (include/mvdb.h: mvdb_encoder_walk_stats)."""
        a, f, n = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0), ctypes.c_int(0)
        self._native.check(self._native.lib().mvdb_encoder_walk_stats(self._h, ctypes.byref(a), ctypes.byref(f), ctypes.byref(n)))
        return {"aborts": int(a.value), "fallbacks": int(f.value), "suspended_calls": int(n.value)}

    def forward(self, ids, mask, compute=None):
        """ids, mask: int arrays [B,S] (host).  Returns pooled + normalised float32 [B,H]."""
        compute = self.default_compute if compute is None else compute
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        if ids.ndim != 2 or ids.shape != mask.shape:
            raise ValueError("ids and mask must be [B,S] arrays of the same shape")
        B, S = ids.shape
        out = np.empty((B, self.hidden), dtype=np.float32)
        self._native.check(self._native.lib().mvdb_encoder_forward(
            self._h, ctypes.c_void_p(ids.ctypes.data), ctypes.c_void_p(mask.ctypes.data), B, S, compute,
            ctypes.c_void_p(out.ctypes.data)))
        if compute == 2 and not np.isfinite(out).all():
            # the split-precision GEMMs take activations as fp16 pieces: an activation beyond 65504 (no BERT-sized
            # encoder gets near it) overflows to inf.  Same kernels' exact-fp32 mode instead of a NaN embedding.
            self._native.check(self._native.lib().mvdb_encoder_forward(
                self._h, ctypes.c_void_p(ids.ctypes.data), ctypes.c_void_p(mask.ctypes.data), B, S, 0,
                ctypes.c_void_p(out.ctypes.data)))
        return out

    @property
    def overflow_flag_ptr(self):
        """Device address of the uint32 a forward sets to 1 when a pooled row of a non-empty sentence is not finite."""
        return int(self._native.lib().mvdb_encoder_overflow_flag(self._h) or 0)

    def overflow_flag(self):
        """The flag as a 1-element int32 torch tensor living in the encoder's own device word (no copy, no synchronisation):
        `enc.overflow_flag().item()` behind the caller's own stream wait, or a device-side test inside a graph."""
        import torch

        class _Word:  # __cuda_array_interface__ view of the library's word
            def __init__(self, ptr):
                self.__cuda_array_interface__ = {"shape": (1,), "typestr": "<i4", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Word(self.overflow_flag_ptr), device=self.device)

    def forward_device(self, ids, mask, compute=None, want_hidden=False, rerun_on_overflow=False):
        """ids, mask: int32 torch tensors [B,S] on this device.  Returns (pooled [B,H], hidden [B,S,H] or
        None) as torch tensors; enqueued on torch's current stream.  The split-precision mode (compute = 2) overflows when an
        activation leaves the fp16 range: the forward then raises the device-side overflow flag (`overflow_flag()`), which a
        caller can test without reading the embeddings; rerun_on_overflow=True does it here (ONE stream wait per call) and runs
        the exact mode again when it is set, like the host entry."""
        import torch
        compute = self.default_compute if compute is None else compute
        if rerun_on_overflow and compute == 2:
            out, hidden = self.forward_device(ids, mask, compute=2, want_hidden=want_hidden)
            torch.cuda.current_stream().synchronize()
            if int(self.overflow_flag().item()):
                return self.forward_device(ids, mask, compute=0, want_hidden=want_hidden)
            return out, hidden
        B, S = ids.shape
        out = torch.empty((B, self.hidden), dtype=torch.float32, device=self.device)
        hidden = torch.empty((B, S, self.hidden), dtype=torch.float32, device=self.device) if want_hidden else None
        self._native.check(self._native.lib().mvdb_encoder_forward_device(
            self._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(mask.data_ptr()), B, S, compute,
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(hidden.data_ptr() if want_hidden else 0),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out, hidden


class EmbeddingModel:

    def __init__(self, use_quantized_onnx_model=True, alternative_model: AlternativeModel = AlternativeModel.bgem3,
                 onnx_model_cpu_core_count=None, **kwargs):
        self.use_quantized_onnx_model = use_quantized_onnx_model
        self.onnx_model_cpu_core_count = onnx_model_cpu_core_count
        assert isinstance(self.onnx_model_cpu_core_count, int) or self.onnx_model_cpu_core_count is None

        # "e5_model_size" is the reference's legacy spelling of alternative_model
        if 'e5_model_size' in kwargs:
            self.alternative_model = AlternativeModel(kwargs['e5_model_size'])
        else:
            self.alternative_model = alternative_model

        self._device = kwargs.get('device', 0)
        self._model_path = kwargs.get('model_path')
        self._state_dict = kwargs.get('state_dict')
        self._config = kwargs.get('config')
        self.tokenizer = kwargs.get('tokenizer')
        self.model = None

        if self.use_quantized_onnx_model:
            self.load_onnx_model()
        else:
            self.load_alternative_model()

    def load_onnx_model(self):
        raise NotImplementedError(
            "the quantised ONNX USE-multilingual model is not available in minivectordb_amd: its weights blob "
            "(minivectordb/resources/embedding_model_quantized.onnx) is not part of the reference tree, so the graph "
            "cannot be restated for the GPU.  Use EmbeddingModel(use_quantized_onnx_model=False, "
            "alternative_model=AlternativeModel.small) for the e5 encoder.")

    def average_pool(self, last_hidden_states, attention_mask):
        """Kept for API parity (the GPU path fuses this into the pooling kernel)."""
        last_hidden = last_hidden_states.masked_fill(~attention_mask[..., None].bool(), 0.0)
        return last_hidden.sum(dim=1) / attention_mask.sum(dim=1)[..., None]

    def load_alternative_model(self):
        if self.alternative_model in (AlternativeModel.small, AlternativeModel.large):
            if self._state_dict is not None:
                self.model = GpuEncoder(self._config, self._state_dict, device=self._device)
                return
            from transformers import AutoModel, AutoTokenizer
            name = self._model_path or f'intfloat/multilingual-e5-{self.alternative_model.value}'
            if self.tokenizer is None:
                self.tokenizer = AutoTokenizer.from_pretrained(name)
            hf = AutoModel.from_pretrained(name)
            self.model = GpuEncoder(hf.config, hf.state_dict(), device=self._device)
            del hf
        elif self.alternative_model == AlternativeModel.bgem3:
            # BGE-M3's dense vector = L2-normalised CLS state of an XLM-R-large encoder (what
            # FlagEmbedding's BGEM3FlagModel.encode(...)['dense_vecs'] returns, embedding_model.py:74-78)
            if self._state_dict is not None:
                self.model = GpuEncoder(self._config, self._state_dict, device=self._device, pooling="cls")
                return
            from transformers import AutoModel, AutoTokenizer
            name = self._model_path or 'BAAI/bge-m3'
            if self.tokenizer is None:
                self.tokenizer = AutoTokenizer.from_pretrained(name)
            hf = AutoModel.from_pretrained(name)
            self.model = GpuEncoder(hf.config, hf.state_dict(), device=self._device, pooling="cls")
            del hf

    # ---- e5 path ---------------------------------------------------------------------------------------
    def _tokenize(self, texts):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer loaded: pass tokenizer=... or model_path=... to EmbeddingModel")
        batch = self.tokenizer([f'passage {t}' for t in texts], max_length=512, padding=True, truncation=True,
                               return_tensors='np')
        return np.asarray(batch['input_ids'], dtype=np.int32), np.asarray(batch['attention_mask'], dtype=np.int32)

    def encode_ids(self, input_ids, attention_mask, compute=None):
        """Token ids / mask [B,S] -> float32 [B,H] pooled, normalised embeddings (GPU)."""
        return self.model.forward(input_ids, attention_mask, compute=compute)

    def extract_embeddings_e5_multi(self, text):
        ids, mask = self._tokenize([text])
        return self.encode_ids(ids, mask).tolist()[0]

    def extract_embeddings_batch(self, texts):
        """Batched variant (BASELINE config 5: 256 sentences per forward); no reference counterpart —
        row i equals extract_embeddings(texts[i]) up to fp32 rounding."""
        ids, mask = self._tokenize(list(texts))
        return self.encode_ids(ids, mask)

    def extract_embeddings_bgem3(self, text):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer loaded: pass tokenizer=... or model_path=... to EmbeddingModel")
        batch = self.tokenizer([text], max_length=512, padding=True, truncation=True, return_tensors='np')
        ids = np.asarray(batch['input_ids'], dtype=np.int32)
        mask = np.asarray(batch['attention_mask'], dtype=np.int32)
        return self.encode_ids(ids, mask)[0].tolist()

    def extract_embeddings_quant_onnx(self, text):
        raise NotImplementedError("the quantised ONNX model is not available in minivectordb_amd")

    def extract_embeddings(self, text):
        if self.use_quantized_onnx_model:
            return self.extract_embeddings_quant_onnx(text)
        else:
            if self.alternative_model == AlternativeModel.small or self.alternative_model == AlternativeModel.large:
                return self.extract_embeddings_e5_multi(text)
            else:
                return self.extract_embeddings_bgem3(text)
