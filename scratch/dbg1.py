import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, ctypes as C
import pagegen as pg
import parquet_go_amd as pq
from oracle import oracle as orc
rng = np.random.default_rng(101)
cb = pg.ChunkBuilder(gap=2, lead=6)
cb.dict_page(np.arange(2, dtype=np.int32).tobytes(), 2)
cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 1, bytes([1, 3, 0]))
cb.data_page_v1(pg.ENC_RLE_DICTIONARY, 8, bytes([1, 3, 0xA5]))
chunk, pages = cb.finish()
d = pq.Decoder(0)
g = d.decode(chunk, pages, pq.INT32)
print("status", g.status, g.bad_page)
pq.lib().pqd_debug_dump(d._h)
o = orc.read_chunk(chunk, pages, orc.INT32)
print("oracle", o.status, o.values.view('<i4').reshape(-1) if o.status == 0 else None)
if g.status == 0: print(g.values.view('<i4').reshape(-1))
