"""Spec-conformant builders of Parquet page bodies and column chunks for the tests (numpy).

They produce what other writers put in files and the reference's own writer never does: bit-packed runs
mixed with RLE runs at any width, delta blocks at any width, trimmed last blocks, v2 pages.  Page
descriptors point straight at the page bodies (no thrift headers): header parsing is tested separately
on pyarrow-written files.
"""
import struct

import numpy as np

PAGE_DESC_DTYPE = np.dtype([
    ("page_type", "<i4"), ("encoding", "<i4"), ("num_values", "<i4"), ("rep_len", "<i4"),
    ("def_len", "<i4"), ("reserved", "<i4"), ("offset", "<i8"), ("length", "<i8")])

BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY = range(8)
ENC_PLAIN, ENC_PLAIN_DICTIONARY, ENC_RLE = 0, 2, 3
ENC_DELTA_BINARY_PACKED, ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_DELTA_BYTE_ARRAY, ENC_RLE_DICTIONARY = 5, 6, 7, 8
DATA_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2 = 0, 2, 3


def varint(n):
    n = int(n)
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def zigzag64(v):
    v = int(v)
    return ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF


def bitpack(vals, w):
    """LSB-first packing of len(vals) values at width w (len(vals)*w must be a multiple of 8)."""
    if w == 0 or len(vals) == 0:
        return b""
    v = np.asarray(vals).astype(np.uint64)
    bits = ((v[:, None] >> np.arange(w, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8)
    return np.packbits(bits.reshape(-1), bitorder="little").tobytes()


def hybrid_encode(vals, w, rng=None, style="mixed", max_groups=63):
    """RLE / bit-packed hybrid runs that decode to vals (bit-packed runs are padded with zeros to 8).

    style: 'mixed' (random run kinds and lengths), 'rle' (RLE runs only, as parquet-go writes them),
    'bp' (bit-packed only), 'arrow' (RLE for repeats >= 8, else bit-packed groups).
    """
    vals = np.asarray(vals, dtype=np.int64)
    n = len(vals)
    out = bytearray()
    nb = (w + 7) // 8
    i = 0
    rng = rng or np.random.default_rng(0)

    def rle_run(v, c):
        out.extend(varint(c << 1))
        out.extend(int(v).to_bytes(8, "little", signed=False)[:nb])

    def bp_run(chunk):
        g = (len(chunk) + 7) // 8
        pad = np.zeros(g * 8, np.int64)
        pad[:len(chunk)] = chunk
        out.extend(varint((g << 1) | 1))
        out.extend(bitpack(pad, w))

    while i < n:
        if style == "rle":
            j = i + 1
            while j < n and vals[j] == vals[i]:
                j += 1
            rle_run(vals[i], j - i)
            i = j
        elif style == "bp":
            c = min(n - i, 8 * int(rng.integers(1, max_groups + 1)))
            bp_run(vals[i:i + c])
            i += c
        elif style == "arrow":
            j = i + 1
            while j < n and vals[j] == vals[i]:
                j += 1
            if j - i >= 8:
                rle_run(vals[i], j - i)
                i = j
            else:
                # literal groups until a repeat of 8 or more starts on a group boundary
                k = i
                while k < n and k - i < 8 * max_groups:
                    if k > i and (k - i) % 8 == 0:
                        m = k + 1
                        while m < n and m - k < 8 and vals[m] == vals[k]:
                            m += 1
                        if m - k >= 8:
                            break
                    k += 1
                bp_run(vals[i:k])
                i = k
        else:  # mixed
            if rng.random() < 0.5:
                # RLE run over the real repeat (possibly split)
                j = i + 1
                while j < n and vals[j] == vals[i]:
                    j += 1
                c = int(rng.integers(1, j - i + 1))
                rle_run(vals[i], c)
                i += c
            else:
                c = 8 * int(rng.integers(1, max_groups + 1))
                if i + c > n:
                    c = n - i
                elif rng.random() < 0.7:
                    pass
                bp_run(vals[i:i + c])
                i += c
    return bytes(out)


def level_block_v1(levels, max_level, rng=None, style="mixed"):
    """v1 level section: 4-byte length prefix + hybrid runs (layout/page.go:234-238)."""
    w = int(max_level).bit_length()
    body = hybrid_encode(levels, w, rng, style)
    return struct.pack("<i", len(body)) + body


def delta_encode(vals, dtype=np.int64, block=128, n_mb=4, trim=False, rng=None):
    """DELTA_BINARY_PACKED stream.  trim=False: every block carries all its miniblocks, short blocks
    padded with the minimum delta (what parquet-go writes); trim=True: miniblocks past the last value are
    omitted and their widths written as 0."""
    vals = np.asarray(vals, dtype=dtype)
    n = len(vals)
    bits = 64 if np.dtype(dtype).itemsize == 8 else 32
    mask = (1 << bits) - 1
    vm = block // n_mb
    out = bytearray()
    out += varint(block) + varint(n_mb) + varint(n)
    first = int(vals[0]) if n else 0
    zz = ((first << 1) ^ (first >> (bits - 1))) & mask
    out += varint(zz)
    if n <= 1:
        return bytes(out)
    iv = [int(x) for x in vals]
    deltas = [((iv[i] - iv[i - 1] + (1 << (bits - 1))) & mask) - (1 << (bits - 1)) for i in range(1, n)]
    for b0 in range(0, len(deltas), block):
        blk = deltas[b0:b0 + block]
        md = min(blk)
        used_mb = (len(blk) + vm - 1) // vm
        blk = blk + [md] * (block - len(blk))
        adj = [(d - md) & mask for d in blk]
        widths = []
        for m in range(n_mb):
            if trim and m >= used_mb:
                widths.append(0)
            else:
                widths.append(max(adj[m * vm:(m + 1) * vm]).bit_length())
        out += varint(((md << 1) ^ (md >> (bits - 1))) & mask)
        out += bytes(widths)
        for m in range(n_mb):
            if trim and m >= used_mb:
                continue
            out += bitpack(np.array(adj[m * vm:(m + 1) * vm], dtype=np.uint64), widths[m])
    return bytes(out)


def delta_encode_fast(vals, dtype=np.int64):
    """Vectorised DELTA_BINARY_PACKED encoder (block 128 / 4 miniblocks, untrimmed) for larger inputs."""
    vals = np.asarray(vals, dtype=dtype)
    n = len(vals)
    bits = 64 if np.dtype(dtype).itemsize == 8 else 32
    ut = np.uint64 if bits == 64 else np.uint32
    out = [varint(128), varint(4), varint(n)]
    first = int(vals[0]) if n else 0
    out.append(varint(((first << 1) ^ (first >> (bits - 1))) & ((1 << bits) - 1)))
    if n <= 1:
        return b"".join(out)
    d = (vals[1:].view(ut) - vals[:-1].view(ut)).view(dtype)
    nb = (len(d) + 127) // 128
    pad = np.empty(nb * 128, dtype=dtype)
    pad[:len(d)] = d
    for b in range(nb):
        blk = pad[b * 128:(b + 1) * 128]
        k = min(128, len(d) - b * 128)
        md = blk[:k].min()
        blk[k:] = md
        adj = (blk.view(ut) - ut(md)).astype(np.uint64) if bits == 64 else (blk.view(ut) - ut(np.uint32(md))).astype(np.uint64)
        mdi = int(md)
        out.append(varint(((mdi << 1) ^ (mdi >> (bits - 1))) & ((1 << bits) - 1)))
        ws = [int(adj[m * 32:(m + 1) * 32].max()).bit_length() for m in range(4)]
        out.append(bytes(ws))
        for m in range(4):
            out.append(bitpack(adj[m * 32:(m + 1) * 32], ws[m]))
    return b"".join(out)


class ChunkBuilder:
    """Concatenates page bodies into one chunk buffer and records the page descriptors."""

    def __init__(self, gap=0, lead=0):
        self.buf = bytearray(b"\xEE" * lead)   # bytes that stand where a thrift header would be
        self.pages = []
        self.gap = gap

    def _add(self, page_type, encoding, num_values, body, rep_len=0, def_len=0):
        if self.gap:
            self.buf.extend(b"\xEE" * self.gap)
        self.pages.append((page_type, encoding, num_values, rep_len, def_len, 0, len(self.buf), len(body)))
        self.buf.extend(body)

    def dict_page(self, body, num_values, encoding=ENC_PLAIN):
        self._add(DICTIONARY_PAGE, encoding, num_values, body)

    def data_page_v1(self, encoding, num_values, values_body, rep_block=b"", def_block=b""):
        self._add(DATA_PAGE, encoding, num_values, rep_block + def_block + values_body)

    def data_page_v2(self, encoding, num_values, values_body, rep_runs=b"", def_runs=b""):
        self._add(DATA_PAGE_V2, encoding, num_values, rep_runs + def_runs + values_body, len(rep_runs), len(def_runs))

    def finish(self):
        pages = np.array(self.pages, dtype=PAGE_DESC_DTYPE) if self.pages else np.zeros(0, PAGE_DESC_DTYPE)
        return bytes(self.buf), pages


def dict_indices_body(idx, w, rng=None, style="mixed"):
    """Value section of a dictionary-encoded data page: width byte + hybrid runs (layout/page.go:644-655)."""
    return bytes([w]) + hybrid_encode(idx, w, rng, style)


def plain_byte_array(strings):
    """PLAIN BYTE_ARRAY: u32 length + bytes per value."""
    return b"".join(struct.pack("<I", len(x)) + x for x in strings)


def dlba_encode(strings, trim=False):
    """DELTA_LENGTH_BYTE_ARRAY: DELTA_BINARY_PACKED lengths, then the bytes."""
    if not strings:
        return delta_encode(np.zeros(0, np.int32), np.int32)
    return delta_encode(np.array([len(x) for x in strings], np.int32), np.int32, trim=trim) + b"".join(strings)


def dba_encode(strings, trim=False):
    """DELTA_BYTE_ARRAY: prefix lengths (shared with the previous value), then the suffixes as DELTA_LENGTH_BYTE_ARRAY."""
    prefix, suffixes, prev = [], [], b""
    for x in strings:
        k = 0
        m = min(len(prev), len(x))
        while k < m and prev[k] == x[k]:
            k += 1
        prefix.append(k)
        suffixes.append(x[k:])
        prev = x
    return delta_encode(np.array(prefix, np.int32), np.int32, trim=trim) + dlba_encode(suffixes, trim)
