// Launchers of the gfx950 kernels (one per kernel family); all enqueue on `stream` and return the
// hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include "pqd_internal.h"

namespace pqd {

// ---- k_hybrid.hip ----
hipError_t launch_page_setup(const ChunkCtx &c, hipStream_t stream);
// stream seeds, tile prefixes and tile -> stream maps from the uploaded page table (what the host otherwise writes and uploads)
// (work: stream_tables_work_bytes(n_pages) bytes, 8-byte aligned)
size_t stream_tables_work_bytes(int n_pages);
hipError_t launch_stream_tables(const ChunkCtx &c, const TableBases &b, void *work, StreamSeed *seeds, int32_t *tp_levels, int32_t *tp_dict, int32_t *tp_plain,
                                int32_t *ts_levels, int32_t *ts_dict, hipStream_t stream);
// streams [s0, s0 + n): one wavefront walks the run headers of each stream
// wide: one 576-thread workgroup per stream (few long streams) instead of one wavefront per stream
// jump: the wavefront variant that also resolves spans of short runs at once (4 KiB more LDS per stream)
hipError_t launch_hybrid_prescan(const ChunkCtx &c, int s0, int n, bool wide, bool jump, hipStream_t stream);
// dictionary-index streams made of one-byte-header RLE runs only (pages of up to 64 KiB, 4- or 8-byte entries) are
// expanded straight from the page; they are marked and skipped by the prescan and the tile expansion
// (dict_table: the handle's dictionary table; every page looks its values up at its own PageIn.dict_off)
hipError_t launch_rle_only_pages(const ChunkCtx &c, const uint8_t *dict_table, uint8_t *values_out, hipStream_t stream);
// the same for the repetition / definition level streams (streams [0, 2 nd)); counts nulls and row starts per page
hipError_t launch_rle_only_levels(const ChunkCtx &c, int32_t *rep_out, int32_t *def_out, hipStream_t stream);
// Dictionary-index streams of large pages (at least min_len bytes, 4- or 8-byte entries): run headers walked and values
// written by one workgroup per page, the page read once; streams it finishes are marked and skipped by the prescan and
// the tile expansion, anything unusual is left to them.  max_dict_n: entries of the largest dictionary of the call.
hipError_t launch_page_fused(const ChunkCtx &c, const uint8_t *dict_table, uint8_t *values_out, int max_dict_n, int min_len,
                             hipStream_t stream);
hipError_t launch_hybrid_prescan_wide(const ChunkCtx &c, int s0, int n, hipStream_t stream);
// definition / repetition levels -> int32 arrays (+ null and row counts)
// stage_bytes: LDS to reserve for the packed bytes of one tile (tiles that need more take a slower path)
// valid_bits / valid_cnt (or nullptr): per definition-level tile (tiles def_tile0 .. n_tiles of the launch), one bit per slot
// that holds a value (128 words a tile) and their number (valid_cnt zeroed by the caller)
hipError_t launch_expand_levels(const ChunkCtx &c, const int32_t *tile_prefix, const int32_t *tile_stream, int n_tiles,
                                int stage_bytes, int32_t *rep_out, int32_t *def_out, TileDesc *descs, int n_cu,
                                uint32_t *valid_bits, int32_t *valid_cnt, int def_tile0, hipStream_t stream);
// block_sums: 32 bytes per 256 pages (device work space)
hipError_t launch_pages_finalize(const ChunkCtx &c, int64_t *totals, int64_t *block_sums, hipStream_t stream);
hipError_t launch_collect_status(const ChunkCtx &c, int32_t *out, hipStream_t stream);
// dictionary indices -> gathered values, for the tiles [tile_lo, tile_hi) (the data pages that share one dictionary;
// `dict` points at its first entry; dict_n < 0: they have none and their indices pass through).  split_mode 0 / 1 pins the mid-size dictionary variant
hipError_t launch_expand_dict(const ChunkCtx &c, const int32_t *tile_prefix, const int32_t *tile_stream, int tile_lo, int tile_hi,
                              int stage_bytes, const uint8_t *dict, int32_t dict_n, uint8_t *values_out, TileDesc *descs,
                              int n_cu, int split_mode, hipStream_t stream);

// ---- k_plain.hip ----
// dict_pages: device list of the call's dictionary pages (page indices)
hipError_t launch_dict_plain_fixed(const ChunkCtx &c, const int32_t *dict_pages, int n_dict, uint8_t *dict_out, hipStream_t stream);
hipError_t launch_plain_fixed(const ChunkCtx &c, const int32_t *tile_prefix, int n_tiles, int64_t max_page_bytes, uint8_t *values_out,
                              hipStream_t stream);
hipError_t launch_byte_stream_split(const ChunkCtx &c, const int32_t *tile_prefix, int n_tiles, uint8_t *values_out,
                                    hipStream_t stream);
hipError_t launch_plain_boolean(const ChunkCtx &c, const int32_t *tile_prefix, int n_tiles, uint8_t *values_out,
                                hipStream_t stream);
hipError_t launch_scatter_rank(const int32_t *def_levels, int64_t n_slots, int max_def, int64_t *partial, int n_tiles,
                               hipStream_t stream);
// slot index of every row start (repetition level 0) + end marker; rep_levels NULL: a flat column
hipError_t launch_row_starts(const int32_t *rep_levels, int64_t n_slots, int64_t n_rows, int64_t *scratch, int64_t *out,
                             hipStream_t stream);
hipError_t launch_scatter_fixed(const ChunkCtx &c, const int32_t *def_levels, const uint8_t *dense, uint8_t *out,
                                int64_t n_slots, int64_t *scratch, hipStream_t stream);
// the same for a column whose data pages are all PLAIN with 4- or 8-byte values: reads the pages, no dense copy
// (scratch: 12 bytes per 4096 slots + 64)
hipError_t launch_scatter_plain(const ChunkCtx &c, const int32_t *def_levels, uint8_t *out, int64_t n_slots, int64_t *scratch,
                                hipStream_t stream);
// the same from the validity bits the level expansion left per definition-level tile: the levels are not read again
// (tile_prefix / tile_stream: the level tiles' tables; n_def_tiles tiles from def_tile0 on; partial: n_def_tiles + 1 int64)
// plan: scatter_plan_bytes(n_def_tiles) bytes of work space
size_t scatter_plan_bytes(int n_def_tiles);
hipError_t launch_scatter_plain_bits(const ChunkCtx &c, const int32_t *tile_prefix, const int32_t *tile_stream, int def_tile0, int n_def_tiles,
                                     const uint32_t *valid_bits, const int32_t *valid_cnt, int64_t *partial, void *plan, uint8_t *out, hipStream_t stream);

// ---- k_delta.hip ----
// one wavefront per page whose value encoding is `encoding`; elem_size 4 or 8; src_sel 1 continues at the end of
// the page's previous delta stream
// work: delta_work_bytes(slots, pages) bytes (slots = the level slots of the call's data pages), or nullptr to
// decode every page with the wavefront-per-page kernel
size_t delta_work_bytes(int64_t slots, int nd);
// bytes_per_value: what the pages of this encoding spend per value (sets how many blocks a wavefront takes at once); 0: not known
// spec: pages whose blocks all look one length long skip the walk of their block headers (the units check the guess)
// narrow (elem_size 4, work given): the values are lengths of strings, computed and kept as 32-bit words; whatever could make
// that differ from the 64-bit values the reference computes sets chunk_status[kLayoutMismatch]
hipError_t launch_delta_decode(const ChunkCtx &c, uint8_t *values_out, int elem_size, int encoding, int src_sel, void *work, int64_t slots,
                               double bytes_per_value, bool spec, bool narrow, hipStream_t stream);

// ---- k_bytes.hip ----
// dict_pages == nullptr: the PLAIN data pages; else the listed dictionary pages
hipError_t launch_plain_ba_walk(const ChunkCtx &c, const int32_t *dict_pages, int n_dict, uint64_t *pairs_out, hipStream_t stream);
hipError_t launch_dlba_pairs(const ChunkCtx &c, int encoding, const int64_t *lens, uint64_t *pairs_out, hipStream_t stream);
// A chunk of DELTA_LENGTH_BYTE_ARRAY pages only: offsets straight from the lengths, page by page, each page starting where the
// payload sizes of the pages in front say (page_base: nd + 1 int64 of work space).  tail: slots may lie behind the last value
// (nullable column).  chunk_status[kLayoutMismatch] is set when a page holds bytes behind its last string: the offsets behind
// it are then off and launch_len_scan computes them again.
hipError_t launch_dlba_offsets(const ChunkCtx &c, const int32_t *lens, int64_t *page_base, const int64_t *n_dense, int64_t n_slots, bool tail,
                               int64_t *doff, hipStream_t stream);
hipError_t launch_void_failed_pages(const ChunkCtx &c, uint64_t *pairs, int64_t *prefix, hipStream_t stream);
hipError_t launch_len_offsets(const uint64_t *pairs, const int64_t *prefix, int64_t n, int64_t *partial, int64_t *doff,
                              hipStream_t stream);
// The same offsets in one pass (every tile finds the sum of the tiles before it by looking back).  lens == nullptr: lengths
// from the pairs (+ prefix).  lens != nullptr: a chunk of DELTA_LENGTH_BYTE_ARRAY pages only, offsets straight from the
// decoded lengths - no pairs; *n_dense (device) values count, the pages are checked as k_dlba_pairs would (page_flags: one
// int32 per data page of work space).  total_out (device, may be nullptr) also receives the byte total.
size_t len_scan_work_bytes(int64_t n);
hipError_t launch_len_scan(const ChunkCtx &c, const uint64_t *pairs, const int64_t *prefix, const int64_t *lens, const int64_t *n_dense,
                           int64_t n, void *work, int64_t *doff, int64_t *total_out, int32_t *page_flags, hipStream_t stream);
// FIXED_LEN_BYTE_ARRAY columns under the two delta byte-array encodings: doff[i] = i * type_length for i in [0, n], and
// pages holding a string of another length are refused
hipError_t launch_flba_delta_offsets(const ChunkCtx &c, const uint64_t *pairs, const int64_t *prefix, int64_t n, int64_t *doff,
                                     hipStream_t stream);
// The copies write nothing when the byte total (*total, on the device) exceeds cap, the capacity of bytes_out.
hipError_t launch_copy_values(const ChunkCtx &c, const uint64_t *pairs, const int64_t *doff, int flat_dlba, uint8_t *bytes_out,
                              const int64_t *total, int64_t cap, hipStream_t stream);
// (pairs == nullptr: a page's bytes start where its length stream ended)
// max_page_bytes: at least the longest page's bytes (sets the grid); plan: copy_plan_bytes(nd) bytes of work space
size_t copy_plan_bytes(int nd);
hipError_t launch_copy_dlba_pages(const ChunkCtx &c, const uint64_t *pairs, const int64_t *doff, uint8_t *bytes_out,
                                  const int64_t *total, int64_t cap, int64_t max_page_bytes, void *plan, hipStream_t stream);
// A chunk of DELTA_BYTE_ARRAY pages only: offsets and suffix positions from sums per 64 strings (blk: 8 bytes per value slot,
// bmin: 4 bytes per value slot, page_work: (2 nd + 2) * 8 bytes); chunk_status[kLayoutMismatch] is set when a page is not
// clean, and the call is then decoded by the kernels that keep a descriptor per value.
hipError_t launch_dba_lean_offsets(const ChunkCtx &c, int32_t *prefix, const int32_t *lens, void *blk, int32_t *bmin, int64_t *page_work,
                                   const int64_t *n_dense, int64_t n_slots, bool tail, int64_t *doff, hipStream_t stream);
hipError_t launch_dba_lean_chain(const ChunkCtx &c, const int32_t *prefix, const int32_t *lens, const void *blk, const int32_t *bmin,
                                 const int64_t *page_work, int64_t *doff, uint8_t *bytes_out, const int64_t *total, int64_t cap,
                                 hipStream_t stream);
hipError_t launch_dba_resolve(const ChunkCtx &c, int64_t *prefix, const uint64_t *pairs, int32_t *parent, const int64_t *doff,
                              int32_t *page_flag, uint8_t *bytes_out, int phase, const int64_t *total, int64_t cap,
                              hipStream_t stream);
hipError_t launch_scatter_offsets(const int32_t *def, int64_t n_slots, int max_def, const int64_t *partial, const int64_t *doff,
                                  int64_t *slot_off, hipStream_t stream);

}  // namespace pqd
