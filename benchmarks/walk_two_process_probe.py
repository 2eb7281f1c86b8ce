#!/usr/bin/env python3
"""Two (or N) processes embedding one sentence per call on ONE GPU, with or without the cross-process gate
(MVDB_WALK_LOCK=0): do they all finish, how long does each take, how many walking launches were abandoned by their bounded
waits, and do the embeddings agree with a process alone.  usage: walk_two_process_probe.py [--procs 2] [--n 2000] [--large]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, {root!r})
from oracle import encoder as E
from minivectordb_amd.embedding_model import GpuEncoder
cfg = E.make_config({shape!r})
w = E.make_weights(cfg, 61)
enc = GpuEncoder(cfg, {{k: torch.from_numpy(v) for k, v in w.items()}}, device=0)
rs = np.random.RandomState(62)
lens = rs.randint(4, 61, size={n})
sents = [rs.randint(5, cfg["vocab_size"], size=(1, int(L))).astype(np.int32) for L in lens]
enc.forward(sents[0], np.ones_like(sents[0]))
open({ready!r} + str(os.getpid()), "w").close()
while len([f for f in os.listdir(os.path.dirname({ready!r})) if f.startswith(os.path.basename({ready!r}))]) < {procs}:
    time.sleep(0.01)
t0 = time.perf_counter()
out = np.concatenate([enc.forward(s, np.ones_like(s)) for s in sents])
dt = time.perf_counter() - t0
np.save({out!r} + str(os.getpid()) + ".npy", out)
print(json.dumps({{"pid": os.getpid(), "seconds": round(dt, 3), **enc.walk_stats()}}))
"""


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def spawn(tmp, tag, procs, n, shape, env):
    ready, out = os.path.join(tmp, f"ready_{tag}_"), os.path.join(tmp, f"out_{tag}_")
    code = CHILD.format(root=ROOT, n=n, ready=ready, procs=procs, out=out, shape=shape)
    e = dict(os.environ)
    e.update(env)
    ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e) for _ in range(procs)]
    res = []
    for p in ps:
        so, se = p.communicate(timeout=900)
        if p.returncode:
            raise SystemExit(se[-3000:])
        res.append(json.loads(so.strip().splitlines()[-1]))
    return res, [np.load(out + str(r["pid"]) + ".npy") for r in res]


procs, n = int(arg("--procs", "2")), int(arg("--n", "2000"))
shape = "xlmr-large-dims" if "--large" in sys.argv else "e5-small-dims"
with tempfile.TemporaryDirectory() as tmp:
    alone, ref = spawn(tmp, "alone", 1, n, shape, {})
    print(json.dumps({"case": "one process alone", "shape": shape, "sentences": n, "children": alone}), flush=True)
    for tag, env in (("gate", {}), ("nogate", {"MVDB_WALK_LOCK": "0"})):
        res, outs = spawn(tmp, tag, procs, n, shape, env)
        print(json.dumps({"case": f"{procs} processes, " + ("advisory lock" if tag == "gate" else "MVDB_WALK_LOCK=0: bounded waits alone"),
                          "children": res, "bit_identical_to_alone": [bool(np.array_equal(o, ref[0])) for o in outs],
                          "max_abs_diff_vs_alone": [float(np.nanmax(np.abs(o - ref[0]))) for o in outs],
                          "non_finite_rows": [int((~np.isfinite(o).all(axis=1)).sum()) for o in outs]}), flush=True)
