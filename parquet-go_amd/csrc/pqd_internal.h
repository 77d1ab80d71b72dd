// Internal structures shared by the host orchestration (pqd_api.hip) and the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "../../include/pqdecode.h"

namespace pqd {

constexpr int kTile = 4096;        // values per expand tile (256 threads x 16 values)
constexpr int kRunSlice = 1024;    // run-table entries of one tile kept in LDS
constexpr int kPlainTile = 16384;  // bytes per block-iteration of the flat copy kernels

// One page as the host describes it (built from pqd_page_desc).
struct PageIn {
  int64_t body_off;
  int64_t body_len;
  int32_t page_type;
  int32_t encoding;
  int32_t num_values;
  int32_t rep_len;
  int32_t def_len;
  int32_t data_idx;   // index among the chunk's data pages, -1 for a dictionary page
  int64_t slot_base;  // level slots of all earlier data pages
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
};

// Run-table entry: x = out_start | (bit_packed << 31); y = RLE value, or byte offset of the packed
// payload inside the stream.  The entry after a stream's last run holds the end position.
struct RunEntry {
  uint32_t x;
  uint32_t y;
};

struct ChunkCtx {
  const uint8_t *in;     // chunk bytes
  int64_t in_len;
  const PageIn *pages;
  PageState *state;
  Stream *streams;       // [rep x nd | def x nd | values x nd]
  RunEntry *runs;
  int32_t *tile_run;
  int32_t n_pages;
  int32_t nd;            // data pages
  int32_t max_def, max_rep;
  int32_t physical_type, type_length, elem_size;
  int32_t *chunk_status; // [0] status, [1] bad page
};

}  // namespace pqd
