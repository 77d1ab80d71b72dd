import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
import numpy as np, ctypes as C
import parquet_go_amd as pq
from parquet_go_amd import binding
binding.lib_path = lambda: '/root/repo/parquet-go_amd/lib/libpqdecode_stamps.so'
from gen import pqgen
for w, dn in ((1, 2), (8, 256), (20, 65536)):
    chunk, pages = pqgen.dict_chunk(5, 8 << 20, w, dn, 1 << 20)
    d = pq.Decoder(0)
    for i in range(2):
        d._is_bytes = False
        d.submit(chunk, pages, pq.INT32); r = d.wait()
    print("w", w, "status", r.status, file=sys.stderr)
    pq.lib().pqd_debug_dump(d._h)
