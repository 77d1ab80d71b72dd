import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import numpy as np
import parquet_go_amd as pq
from gen import pqgen
w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64 << 20
pv = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
chunk, pages = pqgen.dict_chunk(5, n, w, min(1 << w, 65536), pv)
t = torch.empty(chunk.size + 64, dtype=torch.uint8, device='cuda'); t[:chunk.size].copy_(torch.from_numpy(chunk)); torch.cuda.synchronize()
d = pq.Decoder(0, profiling=True)
for i in range(reps):
    d.submit_device(t.data_ptr(), chunk.size, pages, pq.INT32); r = d.wait()
    assert r.status == 0
print(w, n, pv, {k: round(v[0], 4) for k, v in d.kernel_times().items()})
