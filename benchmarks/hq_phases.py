#!/usr/bin/env python3
"""Where a stage of the query-split fp16 nomination kernel (flat_scan_hq_kernel) spends its cycles — ablation build only:
    make -C minivectordb_amd/csrc ABLATE=1
    MVDB_LIBMVDB=minivectordb_amd/lib/libmvdb_ablate.so python benchmarks/hq_phases.py [nq] [d] [rows]
Wave 0 of every block stamps the shader clock around the phases of each stage (K-half of a 32-row tile); prints the mean
cycles per stage and phase over a batch search of nq queries."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from minivectordb_amd import _native as native  # noqa: E402
from oracle import flat  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
idx = native.FlatIndex(d)
idx.reserve(n)
idx.add_synthetic(n, 1234, normalize=True)
q = flat.synth(nq, d, 5678)
flat.normalize_l2(q)
fn = native.lib().mvdb_debug_hq_phases
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
buf = np.zeros(8, np.uint64)
for _ in range(3):
    idx.search(q, 10)
assert fn(buf.ctypes.data) == 0
reps = 10
for _ in range(reps):
    idx.search(q, 10)
assert fn(buf.ctypes.data) == 0
st = float(buf[4])
names = ["wait for the raw stage (vmcnt)", "raw reads + MFMAs (+ interleaved refill / conversion)",
         "behind the MFMAs (refill / conversion if not interleaved, gate)", "lgkmcnt + barrier"]
tot = float(buf[:4].sum())
print(f"nq {nq} d {d} rows {n}: {st / reps:.0f} stages per search (wave 0 of each block), {tot / st:.0f} cycles per stage")
for i, nme in enumerate(names):
    print(f"  {nme:62s} {float(buf[i]) / st:8.0f} cycles  {100 * float(buf[i]) / tot:5.1f} %")
