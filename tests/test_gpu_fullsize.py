"""The bench configurations at their full sizes, checked through properties that do not need the CPU oracle to
decode hundreds of millions of values: a chunk decodes to the concatenation of its halves, sampled pages equal the
oracle bit for bit, decoded deltas lie in the range the generator drew them from, two encodings of the same strings
decode to the same bytes, null scatter and row counts agree with the levels and with the PLAIN payload in the file."""
import numpy as np
import pytest

from gen import pqgen

pytestmark = pytest.mark.gpu


class _DevArr:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def dev_tensor(ptr, n, typestr):
    import torch
    return torch.as_tensor(_DevArr(ptr, n, typestr), device="cuda")


def to_device(a):
    import torch
    t = torch.empty(a.size + 64, dtype=torch.uint8, device="cuda")
    t[:a.size].copy_(torch.from_numpy(a))
    t[a.size:].zero_()
    return t


def checksum(t):
    """Order-sensitive: sum of value * (position mod 65521 + 1), and the plain sum, both in wrapping int64."""
    import torch
    pos = torch.arange(t.numel(), device=t.device, dtype=torch.int64) % 65521 + 1
    v = t.to(torch.int64)
    return int((v * pos).sum().item()), int(v.sum().item())


@pytest.fixture(scope="module")
def pq():
    import torch  # before the library: torch carries its own HIP runtime
    assert torch.cuda.is_available()
    import parquet_go_amd as pq
    return pq


@pytest.mark.parametrize("w", [12, 1, 8, 20])
def test_dictionary_chunk_equals_its_halves_and_the_oracle(pq, oracle, w):
    # (w = 20: a dictionary of 64 Ki entries, larger than LDS; w = 1: two entries)
    import torch
    n, page_values = 256 << 20, 1 << 20
    chunk, pages = pqgen.dict_chunk(0x5EED0102 + w, n, w, min(1 << w, 65536), page_values)
    dev = to_device(chunk)
    dec = pq.Decoder(0)
    dec.submit_device(dev.data_ptr(), chunk.size, pages, pq.INT32)
    r = dec.wait(check=True)
    assert r.num_slots == n and r.num_rows == n and r.num_nulls == 0
    full = dev_tensor(dec.device_result().values, n, "<i4").clone()
    torch.cuda.synchronize()   # the copy runs on torch's stream: it must be done before the handle's buffer is written again
    half = 1 + (pages.size - 1) // 2
    parts = []
    for sub in (pages[:half], np.concatenate([pages[:1], pages[half:]])):
        dec.submit_device(dev.data_ptr(), chunk.size, sub, pq.INT32)
        r2 = dec.wait(check=True)
        parts.append(dev_tensor(dec.device_result().values, r2.num_slots, "<i4").clone())
        torch.cuda.synchronize()
    assert parts[0].numel() + parts[1].numel() == n
    assert torch.equal(full, torch.cat(parts))
    assert checksum(full) == checksum(torch.cat(parts))
    # sampled pages against the oracle
    rng = np.random.default_rng(5 + w)
    for k in sorted(rng.choice(np.arange(1, pages.size), 3, replace=False)):
        o = oracle.read_chunk(chunk, np.concatenate([pages[:1], pages[k:k + 1]]), oracle.INT32)
        assert o.status == 0
        got = full[(k - 1) * page_values:k * page_values].cpu().numpy()
        assert np.array_equal(got, o.values.view("<i4").ravel())
    dec.close()


def test_delta_int64_steps_stay_in_the_generators_range(pq, oracle):
    import torch
    n, page_values, w, base_step = 512 << 20, 128 << 10, 12, 1000
    chunk, pages = pqgen.delta_chunk(0x5EED0103, n, w, page_values, 8, 1_700_000_000_000_000, base_step)
    dev = to_device(chunk)
    dec = pq.Decoder(0)
    dec.submit_device(dev.data_ptr(), chunk.size, pages, pq.INT64)
    r = dec.wait(check=True)
    assert r.num_slots == n
    v = dev_tensor(dec.device_result().values, n, "<i8").view(n // page_values, page_values)
    step = v[:, 1:] - v[:, :-1]
    assert int(step.min().item()) >= base_step and int(step.max().item()) < base_step + (1 << w)
    del step
    for k in (0, 1777, pages.size - 1):
        o = oracle.read_chunk(chunk, pages[k:k + 1], oracle.INT64)
        assert o.status == 0 and np.array_equal(v[k].cpu().numpy(), o.values.view("<i8").ravel())
    dec.close()


def test_two_encodings_of_the_same_strings_decode_alike(pq, oracle):
    import torch
    n, page_values = 100_000_000, 64 << 10
    out = {}
    dec = pq.Decoder(0)
    for mode in (0, 1):   # DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY
        chunk, pages = pqgen.strings_chunk(0x5EED0104, n, page_values, mode)
        dev = to_device(chunk)
        dec.submit_device(dev.data_ptr(), chunk.size, pages, pq.BYTE_ARRAY)
        r = dec.wait(check=True)
        assert r.num_slots == n
        d = dec.device_result()
        out[mode] = (dev_tensor(d.offsets, n + 1, "<i8").clone(), dev_tensor(d.bytes, r.bytes_len, "|u1").clone())
        torch.cuda.synchronize()   # the copies run on torch's stream: done before the handle's buffers are written again
        if mode == 1:
            o = oracle.read_chunk(chunk, pages[:1], oracle.BYTE_ARRAY)
            assert o.status == 0
            k = int(o.offsets[-1])
            assert np.array_equal(out[1][0][:page_values + 1].cpu().numpy(), o.offsets)
            assert np.array_equal(out[1][1][:k].cpu().numpy(), o.bytes)
        del dev
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    lens = out[0][0][1:] - out[0][0][:-1]
    assert int(lens.min().item()) >= 8 and int(lens.max().item()) <= 64
    dec.close()


def test_list_chunk_counts_and_scatter(pq, oracle):
    import torch
    n, page_slots = 128 << 20, 128 << 10
    chunk, pages = pqgen.list_chunk(0x5EED0105, n, page_slots)
    dev = to_device(chunk)
    dec = pq.Decoder(0)
    dec.submit_device(dev.data_ptr(), chunk.size, pages, pq.INT64, 0, 3, 1)
    r = dec.wait(check=True)
    assert r.num_slots == n
    d = dec.device_result()
    vals = dev_tensor(d.values, n, "<i8")
    dl = dev_tensor(d.def_levels, n, "<i4")
    rl = dev_tensor(d.rep_levels, n, "<i4")
    present = dl == 3
    assert int((rl == 0).sum().item()) == r.num_rows
    assert int((~present).sum().item()) == r.num_nulls
    assert int(dl.min().item()) >= 0 and int(dl.max().item()) <= 3 and int(rl.max().item()) <= 1
    assert int((vals[~present] != 0).sum().item()) == 0          # null slots hold zero
    # the non-null values, in order, are the PLAIN payload at the end of every page
    import ctypes
    from parquet_go_amd import binding as B
    pc = np.zeros((pages.size, 3), np.int64)
    B._check(B.lib().pqd_page_counts(dec._h, ctypes.c_void_p(pc.ctypes.data), pages.size))
    nn = pc[:, 2] - pc[:, 0]                                    # (nulls, rows, slots) per page: slots - nulls
    ends = pages["offset"] + pages["length"]
    dense = np.concatenate([chunk[int(e) - 8 * int(k):int(e)] for e, k in zip(ends, nn)]).view("<i8")
    assert dense.size == int(present.sum().item())
    assert torch.equal(vals[present], torch.from_numpy(dense.copy()).cuda())
    # sampled pages against the oracle: levels, values, nulls, rows (layout/page.go:835-931)
    rng = np.random.default_rng(55)
    slot0 = np.concatenate([[0], np.cumsum(pc[:, 2])])
    for k in sorted(rng.choice(pages.size, 3, replace=False)):
        o = oracle.read_chunk(chunk, pages[k:k + 1], oracle.INT64, 0, 3, 1)
        assert o.status == 0 and (o.n_slots, o.n_nulls, o.n_rows) == (int(pc[k, 2]), int(pc[k, 0]), int(pc[k, 1]))
        lo, hi = int(slot0[k]), int(slot0[k + 1])
        assert np.array_equal(rl[lo:hi].cpu().numpy(), o.rep_levels) and np.array_equal(dl[lo:hi].cpu().numpy(), o.def_levels)
        assert np.array_equal(vals[lo:hi].cpu().numpy(), o.values.view("<i8").ravel())
    # row starts: one entry per row and the end marker
    rs = dec.row_starts(r)
    assert rs.size == r.num_rows + 1 and rs[0] == 0 and rs[-1] == n
    assert torch.equal(torch.from_numpy(rs[:-1]).cuda(), torch.nonzero(rl == 0).ravel())
    dec.close()


def test_list_column_as_1024_row_groups(pq, oracle):
    """The nested-list configuration at its stated shape: 1024 row-group chunks of 1 Mi level slots each (8 pages of
    128 Ki slots), decoded 64 chunks per call through pqd_decode_chunks_device - the calls the bench times.  Every call
    is checked through its totals and levels; sampled row groups are compared with the oracle slot for slot: repetition
    and definition levels, values, nulls and rows (layout/page.go:835-931)."""
    import torch
    n_rg, per_call, rg_slots = 1024, 64, 1 << 20
    rng = np.random.default_rng(1024)
    sampled = set(int(x) for x in rng.choice(n_rg, 10, replace=False)) | {0, n_rg - 1}
    dec = pq.Decoder(0)
    checked = 0
    for c0 in range(0, n_rg, per_call):
        parts = [pqgen.list_chunk(0x5EED2000 + k, rg_slots, 128 << 10) for k in range(c0, c0 + per_call)]
        devs = [to_device(c) for c, _ in parts]
        torch.cuda.synchronize()
        dec.submit_chunks_device([t.data_ptr() for t in devs], [c.size for c, _ in parts], [p for _, p in parts], pq.INT64, 0, 3, 1)
        r = dec.wait(check=True)
        n = per_call * rg_slots
        assert r.num_slots == n
        d = dec.device_result()
        vals, dl, rl = dev_tensor(d.values, n, "<i8"), dev_tensor(d.def_levels, n, "<i4"), dev_tensor(d.rep_levels, n, "<i4")
        present = dl == 3
        assert int((rl == 0).sum().item()) == r.num_rows and int((~present).sum().item()) == r.num_nulls
        assert int((vals[~present] != 0).sum().item()) == 0
        # every row group starts a row: its first repetition level is 0
        assert int(rl[::rg_slots].abs().sum().item()) == 0
        for k in range(c0, c0 + per_call):
            if k not in sampled:
                continue
            chunk, pages = parts[k - c0]
            o = oracle.read_chunk(chunk, pages, oracle.INT64, 0, 3, 1)
            assert o.status == 0 and o.n_slots == rg_slots
            lo = (k - c0) * rg_slots
            assert np.array_equal(rl[lo:lo + rg_slots].cpu().numpy(), o.rep_levels), k
            assert np.array_equal(dl[lo:lo + rg_slots].cpu().numpy(), o.def_levels), k
            assert np.array_equal(vals[lo:lo + rg_slots].cpu().numpy(), o.values.view("<i8").ravel()), k
            assert int((rl[lo:lo + rg_slots] == 0).sum().item()) == o.n_rows
            assert int((dl[lo:lo + rg_slots] != 3).sum().item()) == o.n_nulls
            checked += 1
        torch.cuda.synchronize()
        del devs
    assert checked == len(sampled)
    dec.close()
