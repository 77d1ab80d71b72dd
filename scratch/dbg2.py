import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
x = torch.ones(4, device='cuda'); torch.cuda.synchronize()
import numpy as np
import pagegen as pg
import parquet_go_amd as pq
from oracle import oracle as orc
rng = np.random.default_rng(77)
cb = pg.ChunkBuilder(lead=16)
nd = 4096
cb.dict_page(rng.integers(0, 256, nd * 4, dtype=np.uint8).tobytes(), nd)
idx = rng.integers(0, nd, 250000)
cb.data_page_v1(pg.ENC_RLE_DICTIONARY, len(idx), pg.dict_indices_body(idx, 12, rng, "arrow"))
chunk, pages = cb.finish()
t = torch.frombuffer(bytearray(chunk), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
dec = pq.Decoder(0, stream=torch.cuda.current_stream().cuda_stream)
dec.submit_device(t.data_ptr(), t.numel(), pages, pg.INT32)
g = dec.fetch()
o = orc.read_chunk(chunk, pages, orc.INT32)
print(g.status, np.array_equal(g.values, o.values))
import os
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l][:3])
