"""The tiers behind a refused certificate under the profiler: clustered corpus 10M x 512, 256 queries per call, 6 calls.
Run under `rocprofv3 --kernel-trace --stats` (kernel times) and, separately, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (HBM bytes of
the rescue launch against its algorithmic N x d x 2): see benchmarks/collect_rescue_trace.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minivectordb_amd import _native as native
dev = torch.device("cuda", 0)
n, d, k, nq = 10_000_000, 512, 10, 256
fam = 2 << 56
idx = native.FlatIndex(d)
idx.reserve(n)
idx.add_synthetic(n, 1234 | fam, normalize=True)
stream = torch.cuda.current_stream().cuda_stream
q = torch.empty((nq, d), dtype=torch.float32, device=dev)
native.check(native.lib().mvdb_synth_fill_device(q.data_ptr(), nq, d, 5678 | fam, 0, 1, 0, stream))
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    idx.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), stream=stream)
torch.cuda.synchronize()
print("refused chunks so far", native.split_rerun_count())
