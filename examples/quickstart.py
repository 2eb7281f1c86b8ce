#!/usr/bin/env python3
"""The reference README's usage (store, search, filters, autocut, persist) against minivectordb_amd.
Needs an MI355X (no CPU fallback).  Embeddings here are synthetic; with model files available use
EmbeddingModel(use_quantized_onnx_model=False, alternative_model=AlternativeModel.small,
model_path=...) and pass extract_embeddings(text) instead."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from minivectordb_amd import VectorDatabase  # noqa: E402

rng = np.random.default_rng(0)
db = VectorDatabase(storage_file=os.path.join(tempfile.mkdtemp(), "db.pkl"))
sentences = ["i like animals", "i like cars", "i like programming", "dogs are friendly", "the market fell"]
vectors = rng.standard_normal((len(sentences), 512)).astype(np.float32)
for i, (s, v) in enumerate(zip(sentences, vectors), start=1):
    db.store_embedding(i, v, {"text": s, "category": "animals" if "animal" in s or "dog" in s else "other", "n": i})

query = vectors[0] + 0.1 * rng.standard_normal(512).astype(np.float32)
ids, distances, metadatas = db.find_most_similar(query, k=3)
print("top-3:", ids, [round(float(d), 4) for d in distances], [m["text"] for m in metadatas])

ids, distances, metadatas = db.find_most_similar(query, metadata_filter={"category": "animals"}, k=3)
print("category = animals:", ids)
ids, _, _ = db.find_most_similar(query, or_filters=[{"n": {"$gte": 4}}, {"text": "i like cars"}], exclude_filter={"n": 5}, k=5)
print("or / exclude:", ids)
ids, distances, _ = db.find_most_similar(query, k=5, autocut=True)
print("autocut:", ids, [round(float(d), 4) for d in distances])

db.delete_embedding(2)
db.persist_to_disk()
db2 = VectorDatabase(storage_file=db.storage_file)
print("after delete + reload:", db2.find_most_similar(query, k=5)[0])
