import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from gen import pqgen
from oracle import oracle as orc
for w, dn in ((1, 2), (8, 256), (12, 4096), (20, 65536)):
    t = time.time(); chunk, pages = pqgen.dict_chunk(5, 4 << 20, w, dn, 1 << 20); t1 = time.time() - t
    t = time.time(); rc, n = orc.read_chunk_timed(chunk, pages, orc.INT32); t2 = time.time() - t
    print("dict w", w, "gen %.2fs oracle rc %d n %d %.2fs -> %.1f Mvals/s" % (t1, rc, n, t2, n / t2 / 1e6), "bytes", int(pages['length'].sum()))
chunk, pages = pqgen.dict_chunk(5, 1 << 20, 32, 1000, 2000, style=1, rep_max=3)
rc, n = orc.read_chunk_timed(chunk, pages, orc.INT32); print("pq-go-like", rc, n, len(pages))
for w in (0, 4, 12, 20, 40):
    t = time.time(); chunk, pages = pqgen.delta_chunk(6, 4 << 20, w, 128 << 10); t1 = time.time() - t
    t = time.time(); rc, n = orc.read_chunk_timed(chunk, pages, orc.INT64); t2 = time.time() - t
    print("delta w", w, "gen %.2fs oracle rc %d n %d %.2fs -> %.1f Mvals/s" % (t1, rc, n, t2, n / t2 / 1e6), "bytes", int(pages['length'].sum()))
tb = orc.read_chunk(chunk, pages, orc.INT64)
v = tb.values.view('<i8').reshape(-1)
print(v[:4], np.all(np.diff(v[:131072]) >= 1000))
chunk, pages = pqgen.delta_chunk(6, 1 << 20, 12, 128 << 10, elem_size=4, start=1000)
rc, n = orc.read_chunk_timed(chunk, pages, orc.INT32); print("delta32", rc, n)
