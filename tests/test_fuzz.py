"""Malformed input: seeded edits of valid chunks (tests/fuzzcases.py).

What must hold for every case:
  * the oracle (and so the reference's algorithm) and the library agree on the status, on the failing page and,
    when the chunk decodes, on every level and value;
  * a page with several defects may fail with another error kind than the oracle's (the reference meets the defects
    in a data-dependent order: a count that is too small shows only after every stream of the page has been read),
    but it is the same page that fails, and such cases stay rare;
  * except where the library documents a refusal: PQD_ERR_BIT_WIDTH (delta miniblock wider than 64 bits, hybrid
    width above 32) and PQD_ERR_REFUSED (a BYTE_ARRAY value running past its page, which the reference zero pads;
    v2 level sections that contradict the column's levels).  A refusal may come at or before the page where the
    oracle fails, or on a chunk the oracle decodes - never later, and never as a silently different table.
"""
import numpy as np
import pytest

import fuzzcases

REFUSALS = (-7, -13)


def disagreement(g, o):
    if g.status in REFUSALS:
        if o.status != 0 and g.bad_page > o.bad_page:
            return "refused page %d after the oracle failed at page %d" % (g.bad_page, o.bad_page)
        return None
    if o.status != 0 and g.status != 0:
        if g.bad_page != o.bad_page:
            return "failing page gpu=%d oracle=%d" % (g.bad_page, o.bad_page)
        return None if g.status == o.status else "kind"
    if g.status != o.status:
        return "status gpu=%d oracle=%d (pages %d / %d)" % (g.status, o.status, g.bad_page, o.bad_page)
    if (g.n_slots, g.n_rows, g.n_nulls) != (o.n_slots, o.n_rows, o.n_nulls):
        return "counts %s vs %s" % ((g.n_slots, g.n_rows, g.n_nulls), (o.n_slots, o.n_rows, o.n_nulls))
    if not (np.array_equal(g.def_levels, o.def_levels) and np.array_equal(g.rep_levels, o.rep_levels)):
        return "levels differ"
    if o.kind == "fixed":
        return None if np.array_equal(g.values, o.values) else "values differ"
    return None if np.array_equal(g.offsets, o.offsets) and np.array_equal(g.bytes, o.bytes) else "bytes differ"


def test_oracle_survives_malformed_chunks(oracle):
    seen = set()
    for name, chunk, pages, ptype, tl, md, mr in fuzzcases.cases(120):
        o = oracle.read_chunk(chunk, pages, ptype, tl, md, mr)
        assert o.status in (0, -1, -2, -3, -4), name
        seen.add(o.status)
    assert 0 in seen and len(seen) >= 3   # the edits do break chunks, and not all of them


@pytest.mark.gpu
@pytest.mark.parametrize("seed,variant", [(20240, None), (777000, "wave"), (31337, "wide"), (162718, "poison")])
def test_library_and_oracle_agree_on_malformed_chunks(oracle, seed, variant):
    import parquet_go_amd as pq
    dec = pq.Decoder(0)
    if variant:
        dec.set_option(pq.binding.OPT_PRESCAN, 1 if variant == "wide" else 0)   # both run-header walks see the malformed streams
    if variant == "poison":
        # every work and output buffer filled with 0xFF before each call: a kernel that reads what it (or a failed
        # page) never wrote shows up as a difference (this seed holds a bit-packed run cut short many tiles early)
        dec.set_option(pq.binding.OPT_POISON, 1)
    bad, refused, other_kind = [], 0, []
    for name, chunk, pages, ptype, tl, md, mr in fuzzcases.cases(300, seed):
        o = oracle.read_chunk(chunk, pages, ptype, tl, md, mr)
        g = dec.decode(chunk, pages, ptype, tl, md, mr)
        refused += g.status in REFUSALS
        msg = disagreement(g, o)
        if msg == "kind":
            other_kind.append("%s: gpu %d oracle %d" % (name, g.status, o.status))
        elif msg:
            bad.append("%s: %s" % (name, msg))
    dec.close()
    assert not bad, "\n".join(bad)
    assert refused < 100          # refusals are the exception
    assert len(other_kind) <= 6, "\n".join(other_kind)
