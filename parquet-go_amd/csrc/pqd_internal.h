// Internal structures shared by the host orchestration (pqd_api.hip) and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "../../include/pqdecode.h"

namespace pqd {

constexpr int kTile = 4096;        // values per expand tile (256 threads x 16 values)
constexpr int kRunSlice = 1024;    // run-table entries of one tile kept in LDS
constexpr int kPlainTile = 16384;  // bytes per block-iteration of the flat copy kernels
// ChunkCtx::chunk_status[kWorkFlag]: set by the run-header walk of the dictionary-index streams when a stream is left for
// the tile expansion (cleared by k_pages_finalize before); while it is clear the expansion kernels return at once
constexpr int kWorkFlag = 4;
constexpr int kDirectCount = 5;    // chunk_status[kDirectCount]: index streams expanded straight from their pages (k_collect_status)
constexpr int kLayoutMismatch = 6; // chunk_status[kLayoutMismatch]: a DELTA_LENGTH_BYTE_ARRAY page holds bytes behind its last string, so the output
                                   // offsets taken from the pages' payload sizes are off: the host computes them again from the lengths
constexpr int kStatusWord = 16;     // chunk_status sits this many 32-bit words into the handle's block of totals

// One page as the host describes it (built from pqd_page_desc).
struct PageIn {
  int64_t body_off;
  int64_t body_len;
  int32_t page_type;
  int32_t encoding;
  int32_t num_values;
  int32_t rep_len;
  int32_t def_len;
  int32_t data_idx;   // index among the data pages of the call, -1 for a dictionary page
  int64_t slot_base;  // level slots of all earlier data pages
  // The dictionary in effect: for a data page the last dictionary page before it in its chunk (ColumnBufferType keeps
  // the latest one and drops it at the next row group, reader/columnbuffer.go:101, :132-135), -1 when there is none;
  // for a dictionary page the page itself.
  int32_t dict_page;
  int32_t dict_n;     // entries of that dictionary
  int64_t dict_off;   // its first entry in the handle's dictionary table (entries of all dictionary pages back to back)
};

// Device-side per-page state, filled by the kernels.
struct PageState {
  int32_t status;
  int32_t n_nulls;     // slots with def != max_def
  int32_t n_rows;      // slots with rep == 0
  int32_t cnt;         // non-null values = num_values - n_nulls
  int64_t dense_base;  // non-null values of all earlier data pages
  int64_t byte_base;   // BYTE_ARRAY: payload bytes of all earlier data pages
  int64_t byte_len;    // BYTE_ARRAY: payload bytes of this page
  int64_t val_off;     // chunk offset of the value section (after the levels)
  int64_t val_len;
  int64_t aux0;        // DELTA_LENGTH/DELTA_BYTE_ARRAY: chunk offset of the byte payload
  int64_t aux1;
  int32_t delta_fast;  // the page's current delta stream has a block table: decoded block-parallel
  int32_t spare;
};

// One 128-value block of a DELTA_BINARY_PACKED stream whose headers have been walked (k_delta_blocks): where its
// first miniblock starts (bytes from the start of the stream), the four miniblock widths (0 for a miniblock the
// stream does not hold) and the minimum delta as the varint holds it (zigzag still applied).
struct __attribute__((aligned(16))) DeltaBlock {
  uint32_t data_off;
  uint32_t widths;
  uint64_t min_delta;
};

// One unit of 64 consecutive blocks of such a stream (k_delta_units): all a wavefront needs to know to start on it,
// in one 64-byte record.  page < 0: the unit holds nothing.  win_start / win_end: the bytes of its blocks, relative to
// the start of the stream.
struct __attribute__((aligned(64))) DeltaUnit {
  int32_t page;
  uint32_t g;          // index of the unit in its page
  uint32_t n_blocks;   // blocks of the whole page
  uint32_t win_start;
  uint32_t win_end;
  uint32_t spare;
  int64_t tbase;       // index of the page's first DeltaBlock
  int64_t src_off;     // chunk offset of the stream
  int64_t out_base;    // index of the page's first value in the output
  int64_t cnt;         // values the page wants
  uint64_t first_raw;  // the stream's first value as its varint holds it
};

// One RLE/bit-packed hybrid stream (repetition levels, definition levels or dictionary indices of one page).
struct Stream {
  int64_t off;       // chunk offset of stream byte 0
  int32_t len;       // length the reference allocates for the stream buffer
  int32_t avail;     // bytes of it that exist; the rest reads as zero (bytes.Reader.Read semantics)
  int32_t w;         // bit width
  int32_t cnt;       // values wanted from the stream
  int64_t out_base;  // element index of the stream's first value in the output array
  int64_t run_base;  // first entry of this stream in the run table
  int32_t run_cap;
  int32_t tile_base; // first entry of this stream in the tile->run table
  int32_t page;
  int32_t n_runs;    // written by the prescan
  int32_t status;    // written by setup / prescan
  int32_t pad;
  int64_t decoded;   // values the whole stream holds (saturating)
  int32_t direct;    // the stream was expanded by k_rle_only_pages: no run table, no tiles
  int32_t spare;
  // index streams of dictionary-encoded pages: the dictionary in effect, copied from the page entry so that a kernel
  // that starts from the stream needs no second dependent load (k_pages_values)
  int64_t dict_off;  // its first entry in the handle's dictionary table
  int32_t dict_n;    // its entries
  int32_t dict_ok;   // 1: the page is dictionary-encoded and a dictionary is in effect
};

// What the host knows about a stream before the page bodies are looked at; k_page_setup builds the Stream from it.
struct StreamSeed {
  int64_t run_base;
  int32_t run_cap;
  int32_t tile_base;
};

// Where the three groups of streams (repetition levels, definition levels, values) start in the run table, the
// tile -> run table and the level tile numbering, and which of them a call decodes: all the seeds of a call follow
// from these and the page table.
struct TableBases {
  int64_t run_base[3];
  int64_t tile_base[3];
  int64_t tlv_base[2];
  int32_t wanted[3];
  int32_t rle_typed;   // Encoding_RLE value pages are index streams of this column (BOOLEAN, INT32, INT64)
  int32_t is_real;     // FLOAT / DOUBLE: BYTE_STREAM_SPLIT pages count as plain tiles
  int32_t pad;
};

// Run-table entry: x = out_start | (bit_packed << 31); y = RLE value, or byte offset of the packed
// payload inside the stream; z = work items of all earlier runs of the stream, an item being up to eight
// consecutive values of one run (a run of n values has ceil(n / 8) items; a bit-packed item is one group of
// w bytes).  The entry after a stream's last run holds the end position and the item total.
struct __attribute__((aligned(16))) RunEntry {
  uint32_t x;
  uint32_t y;
  uint32_t z;
  uint32_t spare;
};

// Everything a block needs to expand one tile (kTile consecutive values of one hybrid stream), precomputed
// by one thread per tile so that the expanding block starts its loads after a single 64-byte read.
struct __attribute__((aligned(64))) TileDesc {
  int64_t in_off;     // chunk offset of stream byte 0
  int64_t out_elem;   // output element index of the stream's value 0
  int64_t run0;       // run-table index of the first run of the slice
  uint32_t item_lo;   // first work item of the tile (stream-wide numbering)
  int32_t n_items;
  int32_t t0, t1;     // values [t0, t1) of the stream
  int32_t org;        // stream position of the first staged byte (16-byte aligned in memory)
  int32_t span;       // bytes from org the tile can touch
  int32_t nr;         // slice entries (runs of the tile + the entry after them)
  int32_t avail;      // stream bytes that exist
  int32_t w;
  int32_t flags;      // bit 0: valid, bit 1: slice and bytes fit the LDS budget, bit 2: definition levels
  int32_t page;
  int32_t pad;
};

struct ChunkCtx {
  const uint8_t *in;     // chunk bytes
  int64_t in_len;
  const PageIn *pages;
  PageState *state;
  Stream *streams;       // [rep x nd | def x nd | values x nd]
  const StreamSeed *seeds; // same order, host-filled
  RunEntry *runs;
  int32_t *tile_run;
  int32_t n_pages;
  int32_t nd;            // data pages
  int32_t max_def, max_rep;
  int32_t physical_type, type_length, elem_size;
  int32_t *chunk_status; // [0] status, [1] bad page
};

}  // namespace pqd
