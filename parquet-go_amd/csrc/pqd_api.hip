// C ABI of libpqdecode.so: handles, workspaces and the kernel sequence of one decode call.
//
// Mirrors, for the pages of one or more column chunks of a leaf column, the loop ColumnBufferType.ReadPage runs per
// page (reader/columnbuffer.go:105-156): layout.ReadPage (layout/page.go:721-937) -> Page.Decode (:139-164)
// -> Table.Merge (layout/table.go:52-71), with the row-group advance of NextRowGroup (:53-103) between chunks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "pqd_kernels.h"

using namespace pqd;

namespace {

thread_local std::string g_last_error;

int fail_hip(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return PQD_ERR_HIP;
}
int fail_arg(const char *what) {
  g_last_error = what;
  return PQD_ERR_INVALID_ARG;
}

#define HIP_TRY(expr)                                \
  do {                                               \
    hipError_t _e = (expr);                          \
    if (_e != hipSuccess) return fail_hip(_e, #expr); \
  } while (0)

// Nothing may unwind across the C ABI: allocation failures of the host containers become PQD_ERR_NOMEM.
#define PQD_GUARDED(...)                                                              \
  try { __VA_ARGS__ }                                                                 \
  catch (const std::bad_alloc &) { g_last_error = "host allocation failed"; return PQD_ERR_NOMEM; } \
  catch (...) { g_last_error = "unexpected exception"; return PQD_ERR_INVALID_ARG; }

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// A fixed region of device memory that buffers are carved from, one call at a time (the entry point that may not
// allocate works out of one).
struct Arena {
  uint8_t *base = nullptr;
  size_t cap = 0, used = 0;
};

// Device buffer of a handle.  By default it owns its memory and only ever grows; with `arena` set it is carved
// anew from the arena on every call; with `fixed` set it is memory of the caller and never changes.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  Arena *arena = nullptr;
  bool fixed = false;
  int ensure(size_t n) {
    if (n <= cap) return PQD_OK;
    if (fixed) { g_last_error = "an output buffer of the caller is too small"; return PQD_ERR_INVALID_ARG; }
    if (arena) {
      const size_t at = align_up(arena->used, 256), want = align_up(n, 256);
      if (at + want > arena->cap) { g_last_error = "the call needs more work space than this entry point holds"; return PQD_ERR_NOMEM; }
      p = arena->base + at; cap = want; arena->used = at + want;
      return PQD_OK;
    }
    size_t want = std::max(n, cap + cap / 2);
    want = (want + 255) / 256 * 256;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return fail_hip(e, "hipFree"); }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e == hipErrorOutOfMemory ? PQD_ERR_NOMEM : fail_hip(e, "hipMalloc"); }
    cap = want;
    return PQD_OK;
  }
  void release() { if (p && !arena && !fixed) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Host staging.  Page-locked by default; `pageable` (the entry point that may not allocate device or page-locked
// memory) keeps it in ordinary heap memory, which HIP copies through its own staging.
struct HostBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool pageable = false;
  int ensure(size_t n) {
    if (n <= cap) return PQD_OK;
    const size_t want = std::max(n, cap + cap / 2);
    if (pageable) {
      void *q = std::realloc(p, want);
      if (!q) { g_last_error = "host allocation failed"; return PQD_ERR_NOMEM; }
      p = q; cap = want;
      return PQD_OK;
    }
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; return fail_hip(e, "hipHostMalloc"); }
    cap = want;
    return PQD_OK;
  }
  // grows like ensure() and keeps the first `keep` bytes
  int ensure_keep(size_t n, size_t keep) {
    if (n <= cap) return PQD_OK;
    if (pageable || keep == 0 || !p) return ensure(n);
    void *old = p;
    p = nullptr;
    const size_t old_cap = cap;
    cap = 0;
    const int rc = ensure(std::max(n, old_cap + old_cap / 2));
    if (rc == PQD_OK) std::memcpy(p, old, std::min(keep, old_cap));
    (void)hipHostFree(old);
    return rc;
  }
  void release() {
    if (p) { if (pageable) std::free(p); else (void)hipHostFree(p); }
    p = nullptr; cap = 0;
  }
};

// A typed stretch of a host block (the tables of a call are built where they are uploaded from).
template <class V>
struct HostSpan {
  V *p = nullptr;
  size_t n = 0;
  V &operator[](size_t i) const { return p[i]; }
  V *data() const { return p; }
  size_t size() const { return n; }
  V &back() const { return p[n - 1]; }
};

struct KernelEvent {
  const char *name;
  hipEvent_t start, stop;
};

// Process-wide defaults of the per-handle options, read from the environment once (PQD_PRESCAN=wide|wave,
// PQD_DIRECT_RLE=0|1, PQD_JUMP=0|1, PQD_NO_PLAIN_SCATTER, PQD_SPLIT=0|1, PQD_POISON=1, PQD_FUSED=0|1, PQD_BYTE_OFFSETS=0|1, PQD_DEVICE_TABLES=0|1, PQD_HEAVY_LANES=0..4).
struct Defaults {
  int opt[PQD_OPT_COUNT];
  int heavy_lanes;
};
constexpr int kMaxLanes = 4;
const Defaults &defaults() {
  static const Defaults d = [] {
    Defaults x;
    for (int &o : x.opt) o = -1;
    x.opt[PQD_OPT_POISON] = 0;
    if (const char *e = std::getenv("PQD_PRESCAN")) x.opt[PQD_OPT_PRESCAN] = (e[0] == 'w' && e[1] == 'i') ? 1 : (e[0] == 'w' && e[1] == 'a') ? 0 : -1;
    if (const char *e = std::getenv("PQD_DIRECT_RLE")) x.opt[PQD_OPT_DIRECT_RLE] = e[0] == '1';
    if (const char *e = std::getenv("PQD_JUMP")) x.opt[PQD_OPT_JUMP] = e[0] == '1';
    if (std::getenv("PQD_NO_PLAIN_SCATTER")) x.opt[PQD_OPT_PLAIN_SCATTER] = 0;
    if (const char *e = std::getenv("PQD_SPLIT")) x.opt[PQD_OPT_SPLIT] = e[0] == '1';
    if (const char *e = std::getenv("PQD_POISON")) x.opt[PQD_OPT_POISON] = e[0] == '1';
    if (const char *e = std::getenv("PQD_DELTA_BLOCKS")) x.opt[PQD_OPT_DELTA_BLOCKS] = e[0] - '0';
    if (const char *e = std::getenv("PQD_FUSED")) x.opt[PQD_OPT_FUSED] = e[0] == '1';
    if (const char *e = std::getenv("PQD_BYTE_OFFSETS")) x.opt[PQD_OPT_BYTE_OFFSETS] = e[0] == '1';
    if (const char *e = std::getenv("PQD_DEVICE_TABLES")) x.opt[PQD_OPT_DEVICE_TABLES] = e[0] == '1';
    const char *e = std::getenv("PQD_HEAVY_LANES");
    const int v = e ? std::atoi(e) : 0;
    x.heavy_lanes = v < 0 ? 0 : (v > kMaxLanes ? kMaxLanes : v);
    return x;
  }();
  return d;
}

}  // namespace

struct pqd_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;
  bool profiling = false;
  bool no_alloc = false;             // work space from an arena, outputs in the caller's memory, nothing page-locked
  int opt[PQD_OPT_COUNT];

  DevBuf in, meta, runs, tile_run, tile_desc, dict, values, dense, def, rep, offsets, bytes, scratch, small;
  DevBuf pairs, lens64, prefix64, parent, doff, dict_pairs;  // BYTE_ARRAY work arrays
  DevBuf row_starts, dba_flags, delta_blocks, lvbits;
  int64_t bytes_len = 0;
  HostBuf stage_in[2], stage_out, stage_meta, host_small;
  hipEvent_t ev_h2d[2] = {nullptr, nullptr};   // the copy out of stage_in[k] has been done
  bool h2d_pending[2] = {false, false};
  int stage_next = 0;

  // description of the last decode call
  HostSpan<PageIn> h_pages;        // inside stage_meta: valid until the handle's next call
  std::vector<int32_t> h_dict_pages;
  struct DictGroup { int dict_of, first_page; int64_t tiles; };      // a run of data pages under one dictionary (or none): its index tiles
  std::vector<DictGroup> h_dict_groups;
  pqd_column_desc col{};
  int n_pages = 0, nd = 0;
  int64_t total_slots = 0;
  int elem_size = 0;
  bool busy = false;
  bool have_result = false;
  bool levels_only = false;          // the last call decoded levels and counts only (skip path)
  pqd_chunk_result result{};
  ChunkCtx ctx{};

  // BYTE_ARRAY: what the byte copies of the last call consist of (they are enqueued against an estimate of the
  // output size and done again by pqd_wait when the strings turn out to need more)
  struct BytesPlan {
    bool active = false, any_plain = false, any_dict_enc = false, any_dlba = false, any_dba = false, nullable = false;
    bool fast_dlba = false;     // DELTA_LENGTH_BYTE_ARRAY pages only, offsets straight from the decoded lengths: no pairs
    bool fast_dba = false;      // DELTA_BYTE_ARRAY pages only: offsets and sources from sums per 64 strings, no pairs
    int64_t slots = 0;
    int64_t max_page_bytes = 0; // the longest data page
  } plan;

  hipEvent_t ev_ready = nullptr, ev_done = nullptr;  // hand-over to and from the device's shared stream

  std::vector<KernelEvent> events;   // pool
  size_t n_events = 0;               // used by the last call

  DevBuf *all_bufs_[24] = {&in, &meta, &runs, &tile_run, &tile_desc, &dict, &values, &dense, &def, &rep, &offsets, &bytes,
                           &scratch, &small, &pairs, &lens64, &prefix64, &parent, &doff, &dict_pairs, &row_starts, &dba_flags,
                           &delta_blocks, &lvbits};
};

namespace {

struct Guard {
  int prev = -1;
  explicit Guard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Prof {
  pqd_handle *h;
  KernelEvent *ev = nullptr;
  hipStream_t st;
  Prof(pqd_handle *hh, const char *name, hipStream_t on = nullptr) : h(hh), st(on ? on : hh->stream) {
    if (!h->profiling) return;
    if (h->n_events == h->events.size()) {
      KernelEvent e{name, nullptr, nullptr};
      if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess) return;
      h->events.push_back(e);
    }
    ev = &h->events[h->n_events++];
    ev->name = name;
    (void)hipEventRecord(ev->start, st);
  }
  ~Prof() { if (ev) (void)hipEventRecord(ev->stop, st); }
};

int elem_size_of(const pqd_column_desc &c) {
  switch (c.physical_type) {
    case PQD_BOOLEAN: return 1;
    case PQD_INT32: case PQD_FLOAT: return 4;
    case PQD_INT64: case PQD_DOUBLE: return 8;
    case PQD_INT96: return 12;
    case PQD_FIXED_LEN_BYTE_ARRAY: return c.type_length;
    default: return 0;
  }
}

// Handles with a decode in flight (submitted, not yet waited for), per device.  A chunk decoded while others are in
// flight on its GPU gets the kernel variants that leave most of each CU to them; a lone chunk gets the ones that
// use the whole GPU to cut its latency.
constexpr int kMaxDevices = 64;
std::atomic<int> g_inflight[kMaxDevices];

void set_busy(pqd_handle *h, bool busy) {
  if (busy == h->busy) return;
  h->busy = busy;
  if (h->device >= 0 && h->device < kMaxDevices) g_inflight[h->device].fetch_add(busy ? 1 : -1, std::memory_order_relaxed);
}
int inflight_others(const pqd_handle *h) {
  if (h->device < 0 || h->device >= kMaxDevices) return 0;
  return g_inflight[h->device].load(std::memory_order_relaxed) - (h->busy ? 1 : 0);
}

// Optional shared streams per device for the bandwidth-bound expansion kernels (PQD_HEAVY_LANES = 1..4): every
// handle walks run headers on its own stream and hands the expansion over to a shared one, so that at most that
// many expansions are resident at once and each runs close to its stand-alone speed.  Off by default: on MI355X
// letting all expansions share the GPU freely finishes a batch of chunks about 20 % sooner, at the price of
// per-kernel durations that say little (each kernel holds a fraction of the CUs for several times its
// stand-alone duration).
std::mutex g_heavy_mu;
hipStream_t g_heavy[kMaxDevices][kMaxLanes] = {};
std::atomic<unsigned> g_heavy_next{0};

hipStream_t heavy_stream(int device) {
  const int lanes = defaults().heavy_lanes;
  if (device < 0 || device >= kMaxDevices || lanes == 0) return nullptr;
  const int lane = (int)(g_heavy_next.fetch_add(1, std::memory_order_relaxed) % (unsigned)lanes);
  std::lock_guard<std::mutex> lk(g_heavy_mu);
  if (!g_heavy[device][lane]) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    g_heavy[device][lane] = s;
  }
  return g_heavy[device][lane];
}

bool is_dict_enc(int e) { return e == PQD_ENC_PLAIN_DICTIONARY || e == PQD_ENC_RLE_DICTIONARY; }

// layout of the 256-byte block every call copies back: totals | status | byte total
constexpr int kSmallStatus = kStatusWord;  // int32 index: [16] status, [17] bad page, [20] work flag
constexpr int kSmallByteTotal = 16;    // int64 index: byte total of a BYTE_ARRAY call

// (2 nd + 2) int64 words behind the page flags: output bytes and output base of every page of a chunk of DELTA_BYTE_ARRAY pages only
int64_t *dba_page_work(pqd_handle *h) { return reinterpret_cast<int64_t *>(static_cast<uint8_t *>(h->dba_flags.p) + align_up(4 * (size_t)h->nd + 64, 256)); }

// copy_plan_bytes(nd) bytes behind them: what the copy of DELTA_LENGTH_BYTE_ARRAY pages notes per page
void *copy_plan(pqd_handle *h) { return static_cast<uint8_t *>(h->dba_flags.p) + align_up(4 * (size_t)h->nd + 64, 256) + align_up(8 * (2 * (size_t)h->nd + 2) + 64, 256); }

// The byte copies of a BYTE_ARRAY call, against the current capacity of h->bytes.
int launch_byte_copies(pqd_handle *h) {
  const pqd_handle::BytesPlan &pl = h->plan;
  const ChunkCtx &c = h->ctx;
  const uint64_t *pairs = static_cast<const uint64_t *>(h->pairs.p);
  int64_t *prefix64 = static_cast<int64_t *>(h->prefix64.p);
  const int64_t *doff = static_cast<const int64_t *>(pl.nullable ? h->doff.p : h->offsets.p);
  const int64_t *d_total = doff + pl.slots;
  const int64_t cap = (int64_t)h->bytes.cap - 64;
  uint8_t *bytes_out = static_cast<uint8_t *>(h->bytes.p);
  const bool all_dlba = pl.any_dlba && !pl.any_dba && !pl.any_plain && !pl.any_dict_enc;
  if (pl.any_plain || pl.any_dict_enc || (pl.any_dlba && !all_dlba)) {
    Prof pf(h, "k_copy_values");
    HIP_TRY(launch_copy_values(c, pairs, doff, all_dlba ? 1 : 0, bytes_out, d_total, cap, h->stream));
  }
  if (all_dlba) { Prof pf(h, "k_copy_dlba_pages"); HIP_TRY(launch_copy_dlba_pages(c, pl.fast_dlba ? nullptr : pairs, doff, bytes_out, d_total, cap, pl.max_page_bytes, copy_plan(h), h->stream)); }
  if (pl.fast_dba) {
    Prof pf(h, "k_dba_chain");
    HIP_TRY(launch_dba_lean_chain(c, reinterpret_cast<const int32_t *>(prefix64), static_cast<const int32_t *>(h->lens64.p), h->pairs.p, static_cast<const int32_t *>(h->parent.p),
                                  dba_page_work(h), const_cast<int64_t *>(doff), bytes_out, d_total, cap, h->stream));
  } else if (pl.any_dba) {
    Prof pf(h, "k_dba_resolve");
    HIP_TRY(launch_dba_resolve(c, prefix64, pairs, static_cast<int32_t *>(h->parent.p), doff, static_cast<int32_t *>(h->dba_flags.p), bytes_out, 1,
                               d_total, cap, h->stream));
  }
  return PQD_OK;
}

// One decode call over pages of one leaf column.  `base` + PageIn.body_off addresses the page bodies in device
// memory; chunk_first[k] .. chunk_first[k + 1] are the pages of chunk k and chunk_off[k] is the offset of its byte 0
// from `base` (the pqd_page_desc offsets are relative to their chunk).
int decode_on_device(pqd_handle *h, const uint8_t *base, size_t span, const pqd_page_desc *pages, int32_t n_pages,
                     int32_t n_chunks, const int32_t *chunk_first, const int64_t *chunk_off, const size_t *chunk_len,
                     const pqd_column_desc *col, bool levels_only) {
  // a handle runs one call at a time: the page table, the meta block and the work buffers of a call still in
  // flight are about to be rewritten
  if (h->busy) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    set_busy(h, false);
  }
  static const bool host_timing = std::getenv("PQD_HOST_TIMING") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_begin = tnow();
  h->have_result = false;
  h->plan = pqd_handle::BytesPlan{};
  const int esz = elem_size_of(*col);
  if (col->physical_type < 0 || col->physical_type > PQD_FIXED_LEN_BYTE_ARRAY) return fail_arg("unknown physical type");
  if (col->physical_type == PQD_FIXED_LEN_BYTE_ARRAY && col->type_length < 0) return fail_arg("negative type_length");
  if (col->max_def < 0 || col->max_rep < 0) return fail_arg("negative max level");
  if (reinterpret_cast<uintptr_t>(base) & 15u) return fail_arg("device chunk buffer must be 16-byte aligned");
  const bool is_bytes = col->physical_type == PQD_BYTE_ARRAY;
  const bool is_flba = col->physical_type == PQD_FIXED_LEN_BYTE_ARRAY;
  if ((is_bytes || is_flba) && span > ((size_t)1 << 40)) return fail_arg("byte-array chunks of one call must lie within 1 TiB of device address space");
  if (is_bytes && h->no_alloc) return fail_arg("BYTE_ARRAY columns need the allocating entry points");

  // ---- host-side page table, built in the block it is uploaded from; the dictionary in effect follows the page
  //      order of each chunk ----
  int rc;
  const size_t pages_bytes = sizeof(PageIn) * (size_t)n_pages;
  if ((rc = h->stage_meta.ensure(pages_bytes + 256))) return rc;
  h->h_pages = HostSpan<PageIn>{static_cast<PageIn *>(h->stage_meta.p), (size_t)n_pages};
  // Chunks of very many pages: the page entries go to the device slice by slice while the host is still walking the
  // rest, into the meta block as the call before left it.  Should that block or the staging block have to grow
  // afterwards, the slices are uploaded again with everything else.
  const int early_slice = h->opt[PQD_OPT_DEVICE_TABLES] == 1 ? 1024 : 4096;
  void *const early_meta = (!h->no_alloc && !h->stage_meta.pageable && h->opt[PQD_OPT_POISON] <= 0 && h->opt[PQD_OPT_DEVICE_TABLES] != 0 &&
                            n_pages >= 2 * early_slice && h->meta.p && h->meta.cap >= pages_bytes) ? h->meta.p : nullptr;
  int early_done = 0;      // pages [0, early_done) are on their way
  h->h_dict_pages.clear();
  h->h_dict_groups.clear();
  double index_bpv = 0.0;      // most page bytes per value among the pages whose values are (or may be) an index stream
  int max_dict_n = 0;
  int nd = 0;
  int64_t slots = 0, dict_total = 0, data_bytes = 0, all_bytes = 0, lv_runs = 0, lv_tiles = 0, max_page_bytes = 0;
  int64_t ix_runs = 0, ix_tiles = 0, plain_tiles = 0;      // run room and tiles of the pages whose values are an index stream; tiles of the flat copies
  int host_status = PQD_OK, host_bad = -1;
  bool any_dict_enc = false, any_plain = false, any_delta = false, any_bss = false, any_rle = false;
  bool any_dlba = false, any_dba = false;
  const bool is_real = col->physical_type == PQD_FLOAT || col->physical_type == PQD_DOUBLE;
  const bool rle_typed = col->physical_type == PQD_BOOLEAN || col->physical_type == PQD_INT32 || col->physical_type == PQD_INT64;
  auto host_fail = [&](int code, int page) { if (host_status == PQD_OK) { host_status = code; host_bad = page; } };
  for (int ch = 0; ch < n_chunks; ch++) {
    int cur_dict = -1;   // DictPage = nil at every row group (reader/columnbuffer.go:101)
    if (chunk_first[ch] < 0 || chunk_first[ch + 1] < chunk_first[ch] || chunk_first[ch + 1] > n_pages) return fail_arg("bad page counts");
    for (int p = chunk_first[ch]; p < chunk_first[ch + 1]; p++) {
      const pqd_page_desc &d = pages[p];
      PageIn &pi = h->h_pages[(size_t)p];
      if (d.offset < 0 || d.length < 0 || (uint64_t)d.offset + (uint64_t)d.length > chunk_len[ch]) return fail_arg("page outside the chunk buffer");
      if (d.length > 0x7FFFFFF0ll) return fail_arg("page larger than 2 GiB");
      pi.body_off = chunk_off[ch] + d.offset; pi.body_len = d.length; pi.page_type = d.page_type; pi.encoding = d.encoding;
      pi.num_values = d.num_values; pi.rep_len = d.rep_len; pi.def_len = d.def_len; pi.slot_base = slots;
      pi.data_idx = -1; pi.dict_page = -1; pi.dict_n = 0; pi.dict_off = 0;
      all_bytes += d.length;
      if (d.page_type == PQD_DATA_PAGE || d.page_type == PQD_DATA_PAGE_V2) {
        pi.data_idx = nd++;
        data_bytes += d.length;
        if (d.length > max_page_bytes) max_page_bytes = d.length;
        if (d.num_values > 0) slots += d.num_values;
        {   // what a level stream of the page takes in the run table and the tile table
          const int64_t nv = d.num_values > 0 ? d.num_values : 0;
          lv_runs += std::min<int64_t>(nv, d.length) + 2;
          lv_tiles += (nv + kTile - 1) / kTile;
          if (is_dict_enc(d.encoding) || d.encoding == PQD_ENC_RLE)
            index_bpv = std::max(index_bpv, d.num_values >= kTile ? (double)d.length / (double)d.num_values : 4.25);
          if (!levels_only) {
            if (h->h_dict_groups.empty() || h->h_dict_groups.back().dict_of != cur_dict) h->h_dict_groups.push_back({cur_dict, p, 0});
            if (is_dict_enc(d.encoding) || (d.encoding == PQD_ENC_RLE && rle_typed && !is_bytes)) {
              ix_runs += std::min<int64_t>(nv, d.length) + 2;
              ix_tiles += (nv + kTile - 1) / kTile;
              h->h_dict_groups.back().tiles += (nv + kTile - 1) / kTile;
            }
            if (d.encoding == PQD_ENC_PLAIN || (d.encoding == PQD_ENC_BYTE_STREAM_SPLIT && is_real)) {
              const int64_t units = col->physical_type == PQD_BOOLEAN ? nv : nv * (int64_t)esz;
              plain_tiles += (units + kPlainTile - 1) / kPlainTile;
            }
          }
        }
        if (cur_dict >= 0) {
          const PageIn &dp = h->h_pages[(size_t)cur_dict];
          pi.dict_page = cur_dict; pi.dict_n = dp.dict_n; pi.dict_off = dp.dict_off;
        }
        const int e = d.encoding;
        if (is_dict_enc(e)) {
          any_dict_enc = true;
          // Page.Decode(nil) leaves the indices in place of the values (layout/page.go:139-143): no typed result
          if (cur_dict < 0 && !levels_only) host_fail(PQD_ERR_NO_DICTIONARY, p);
        } else if (e == PQD_ENC_PLAIN) any_plain = true;
        else if ((is_bytes || is_flba) && e == PQD_ENC_DELTA_LENGTH_BYTE_ARRAY) any_dlba = true;   // layout/page.go:684-694
        else if ((is_bytes || is_flba) && e == PQD_ENC_DELTA_BYTE_ARRAY) any_dba = true;           // layout/page.go:696-706
        else if (!is_bytes && e == PQD_ENC_DELTA_BINARY_PACKED && (col->physical_type == PQD_INT32 || col->physical_type == PQD_INT64)) any_delta = true;
        else if (!is_bytes && e == PQD_ENC_BYTE_STREAM_SPLIT && is_real) any_bss = true;   // layout/page.go:707-714
        else if (!is_bytes && e == PQD_ENC_RLE && rle_typed) any_rle = true;               // layout/page.go:657-667
        else if (!levels_only) host_fail(e == PQD_ENC_RLE ? PQD_ERR_UNSUPPORTED_TYPE : PQD_ERR_UNSUPPORTED_ENCODING, p);  // layout/page.go:671, :682, :713, :716
      } else if (d.page_type == PQD_DICTIONARY_PAGE) {
        pi.dict_page = p; pi.dict_n = d.num_values > 0 ? d.num_values : 0; pi.dict_off = dict_total;
        dict_total += pi.dict_n;
        max_dict_n = std::max(max_dict_n, pi.dict_n);
        cur_dict = p;
        h->h_dict_pages.push_back(p);
      } else {
        host_fail(PQD_ERR_UNSUPPORTED_PAGE_TYPE, p);   // INDEX_PAGE and unknown types, layout/page.go:813,934
      }
      if (early_meta && p + 1 - early_done >= early_slice) {
        HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(early_meta) + sizeof(PageIn) * (size_t)early_done, &h->h_pages[(size_t)early_done],
                               sizeof(PageIn) * (size_t)(p + 1 - early_done), hipMemcpyHostToDevice, h->stream));
        early_done = p + 1;
      }
    }
  }
  auto t_pages = tnow();
  if (slots > 0x7FFFFFFF0ll) return fail_arg("too many level slots in one call");
  const int n_dict = (int)h->h_dict_pages.size();

  // ---- meta block: pages | seeds | tile prefixes | tile -> stream maps | dictionary pages; then, device only,
  //      state | streams ----
  const size_t n_streams = (size_t)3 * nd;
  const size_t off_pages = 0;
  const size_t off_seeds = align_up(off_pages + pages_bytes, 256);
  const size_t off_tp_lv = align_up(off_seeds + sizeof(StreamSeed) * n_streams, 256);
  const size_t off_tp_dc = align_up(off_tp_lv + 4 * (2 * (size_t)nd + 1), 256);
  const size_t off_tp_pl = align_up(off_tp_dc + 4 * ((size_t)nd + 1), 256);
  const size_t off_ts_lv = align_up(off_tp_pl + 4 * ((size_t)nd + 1), 256);
  if (early_done && h->stage_meta.cap < off_ts_lv) HIP_TRY(hipStreamSynchronize(h->stream));      // (the slices on their way are read from the block that is about to move)
  if ((rc = h->stage_meta.ensure_keep(off_ts_lv, pages_bytes))) return rc;
  HostSpan<StreamSeed> h_seeds;
  HostSpan<int32_t> tp_levels, tp_dict, tp_plain;
  auto place_tables = [&] {      // (again after the block has grown)
    uint8_t *m = static_cast<uint8_t *>(h->stage_meta.p);
    h->h_pages = HostSpan<PageIn>{reinterpret_cast<PageIn *>(m + off_pages), (size_t)n_pages};
    h_seeds = HostSpan<StreamSeed>{reinterpret_cast<StreamSeed *>(m + off_seeds), n_streams};
    tp_levels = HostSpan<int32_t>{reinterpret_cast<int32_t *>(m + off_tp_lv), 2 * (size_t)nd + 1};
    tp_dict = HostSpan<int32_t>{reinterpret_cast<int32_t *>(m + off_tp_dc), (size_t)nd + 1};
    tp_plain = HostSpan<int32_t>{reinterpret_cast<int32_t *>(m + off_tp_pl), (size_t)nd + 1};
  };
  place_tables();
  // Run-table ranges, tile-table ranges and tile prefixes of the three groups of streams (repetition levels,
  // definition levels, values), one pass over the pages: the level groups take the same room each (added up with the
  // page table), so every group's base is known beforehand.  A group no stream of which is decoded has empty run
  // ranges and one (end) entry per stream in the tile table.
  // Chunks of very many pages leave that pass, the tile -> stream maps and their upload to a kernel (k_stream_tables):
  // the totals the host needs for its buffers were added up with the page table.
  bool device_tables = nd >= 4096;
  if (h->opt[PQD_OPT_DEVICE_TABLES] >= 0) device_tables = h->opt[PQD_OPT_DEVICE_TABLES] == 1;
  int64_t run_total = 0, tile_total = 0;
  TableBases tbases{};
  int32_t n_lv_tiles = 0, n_lv_tiles_rep = 0, n_dc_tiles = 0, n_pl_tiles = 0;   // level tiles (all; those of the repetition levels), index tiles, plain tiles
  {
    const bool wanted[3] = {col->max_rep > 0, col->max_def > 0, !levels_only};
    int64_t (&run_base)[3] = tbases.run_base, (&tile_base)[3] = tbases.tile_base, (&tlv_base)[2] = tbases.tlv_base;
    int64_t t_lv = 0;
    for (int g = 0; g < 2; g++) {
      run_base[g] = run_total; tile_base[g] = tile_total; tlv_base[g] = t_lv;
      run_total += wanted[g] ? lv_runs : 0;
      tile_total += wanted[g] ? lv_tiles + nd : nd;
      t_lv += wanted[g] ? lv_tiles : 0;
    }
    run_base[2] = run_total; tile_base[2] = tile_total;
    for (int g = 0; g < 3; g++) tbases.wanted[g] = wanted[g];
    tbases.rle_typed = rle_typed && !is_bytes; tbases.is_real = is_real;
    if (tile_total > 0x7FFFFFF0ll || t_lv > 0x7FFFFFF0ll) return fail_arg("chunk too large (tile table)");
    if (tile_total + ix_tiles + nd > 0x7FFFFFF0ll || plain_tiles > 0x7FFFFFF0ll) return fail_arg("chunk too large (tiles)");
    n_lv_tiles = (int32_t)t_lv; n_lv_tiles_rep = (int32_t)tlv_base[1];
    n_dc_tiles = (int32_t)ix_tiles; n_pl_tiles = (int32_t)plain_tiles;
    int64_t acc_runs = 0, acc_tiles = 0, v_runs = 0, v_tiles = 0, t_dc = 0, t_pl = 0;
    StreamSeed *s0 = h_seeds.data(), *s1 = s0 + nd, *s2 = s1 + nd;
    for (int p = 0; p < n_pages && !device_tables; p++) {
      const PageIn &pi = h->h_pages[(size_t)p];
      if (pi.data_idx < 0) continue;
      const int d = pi.data_idx;
      const int64_t nv = pi.num_values > 0 ? pi.num_values : 0;
      const int64_t tiles = (nv + kTile - 1) / kTile;
      const int64_t rcap = std::min<int64_t>(nv, pi.body_len) + 2;
      StreamSeed *const sg[2] = {s0 + d, s1 + d};
      for (int g = 0; g < 2; g++) {
        if (wanted[g]) {
          sg[g]->run_base = run_base[g] + acc_runs; sg[g]->run_cap = (int32_t)rcap; sg[g]->tile_base = (int32_t)(tile_base[g] + acc_tiles + d);
          tp_levels[(size_t)g * nd + d] = (int32_t)(tlv_base[g] + acc_tiles);
        } else {
          sg[g]->run_base = run_base[g]; sg[g]->run_cap = 0; sg[g]->tile_base = (int32_t)(tile_base[g] + d);
          tp_levels[(size_t)g * nd + d] = (int32_t)tlv_base[g];
        }
      }
      acc_runs += rcap; acc_tiles += tiles;
      if (wanted[2]) {
        const bool want = is_dict_enc(pi.encoding) || (pi.encoding == PQD_ENC_RLE && rle_typed && !is_bytes);
        s2[d].run_base = run_base[2] + v_runs; s2[d].run_cap = (int32_t)(want ? rcap : 0); s2[d].tile_base = (int32_t)(tile_base[2] + v_tiles + d);
        tp_dict[(size_t)d] = (int32_t)t_dc;
        tp_plain[(size_t)d] = (int32_t)t_pl;
        if (want) { v_runs += rcap; v_tiles += tiles; t_dc += tiles; }
        if (pi.encoding == PQD_ENC_PLAIN || (pi.encoding == PQD_ENC_BYTE_STREAM_SPLIT && is_real)) {
          const int64_t units = col->physical_type == PQD_BOOLEAN ? nv : nv * (int64_t)esz;
          t_pl += (units + kPlainTile - 1) / kPlainTile;
        }
        if (tile_base[2] + v_tiles + d > 0x7FFFFFF0ll || t_pl > 0x7FFFFFF0ll) return fail_arg("chunk too large (tiles)");
      } else {
        s2[d].run_base = run_base[2]; s2[d].run_cap = 0; s2[d].tile_base = (int32_t)(tile_base[2] + d);
        tp_dict[(size_t)d] = 0; tp_plain[(size_t)d] = 0;
      }
    }
    run_total += wanted[2] ? ix_runs : 0;
    tile_total += (wanted[2] ? ix_tiles : 0) + nd;
    if (tile_total > 0x7FFFFFF0ll) return fail_arg("chunk too large (tile table)");
    if (!device_tables) {
      tp_levels[2 * (size_t)nd] = (int32_t)t_lv;
      tp_dict[(size_t)nd] = (int32_t)t_dc;
      tp_plain[(size_t)nd] = (int32_t)t_pl;
      if (v_runs != (wanted[2] ? ix_runs : 0) || t_dc != n_dc_tiles || t_pl != n_pl_tiles) return fail_arg("internal: table totals disagree");
    }
  }
  auto t_seeds = tnow();
  const size_t off_ts_dc = align_up(off_ts_lv + 4 * (size_t)n_lv_tiles, 256);
  const size_t off_dictp = align_up(off_ts_dc + 4 * (size_t)n_dc_tiles, 256);
  const size_t upload_bytes = align_up(off_dictp + 4 * (size_t)n_dict, 256);
  const size_t off_state = upload_bytes;
  const size_t off_streams = align_up(off_state + sizeof(PageState) * (size_t)n_pages, 256);
  const size_t off_pgsum = align_up(off_streams + sizeof(Stream) * n_streams, 256);
  const size_t off_tbsum = align_up(off_pgsum + 32 * (((size_t)n_pages + 255) / 256 + 1), 256);      // (sums of k_stream_table_sums)
  const size_t meta_bytes = align_up(off_tbsum + (device_tables ? stream_tables_work_bytes(n_pages) : 0), 256);

  // a nullable column of PLAIN pages only: the scatter takes the values from the pages
  const bool plain_scatter = !levels_only && col->max_def > 0 && slots > 0 && any_plain && !any_dict_enc && !any_delta && !any_bss && !any_rle &&
                             !any_dlba && !any_dba && host_status == PQD_OK &&
                             (col->physical_type == PQD_INT32 || col->physical_type == PQD_INT64 || col->physical_type == PQD_FLOAT ||
                              col->physical_type == PQD_DOUBLE) && h->opt[PQD_OPT_PLAIN_SCATTER] != 0;
  // ... and from the validity bits the level expansion leaves per tile, when every level stream goes through it
  const int n_def_tiles = col->max_def > 0 ? n_lv_tiles - n_lv_tiles_rep : 0;
  bool direct_rle = nd > 0 && data_bytes / nd <= 64 * 1024;      // small pages (the reference's writer cuts them at 8 KiB): try the RLE-only expansion first
  if (h->opt[PQD_OPT_DIRECT_RLE] >= 0) direct_rle = h->opt[PQD_OPT_DIRECT_RLE] == 1;
  const bool scatter_bits = plain_scatter && !direct_rle && n_def_tiles > 0 && h->opt[PQD_OPT_PLAIN_SCATTER] != 2;

  // ---- workspaces and outputs (every allocation of the call happens here, before anything is enqueued) ----
  if (h->no_alloc) {
    // carved anew from the arena on every call
    for (DevBuf *b : h->all_bufs_) if (b->arena) { b->p = nullptr; b->cap = 0; }
    if (h->small.arena) h->small.arena->used = 0;
  }
  if (early_done && h->stage_meta.cap < upload_bytes) HIP_TRY(hipStreamSynchronize(h->stream));
  if ((rc = h->meta.ensure(meta_bytes)) || (rc = h->stage_meta.ensure_keep(upload_bytes, off_ts_lv))) return rc;
  place_tables();
  if ((rc = h->small.ensure(256)) || (rc = h->host_small.ensure(256))) return rc;
  if ((rc = h->runs.ensure(sizeof(RunEntry) * (size_t)(run_total + 64)))) return rc;
  if ((rc = h->tile_run.ensure(8 * (size_t)(tile_total + 64)))) return rc;
  if ((rc = h->tile_desc.ensure(sizeof(TileDesc) * ((size_t)std::max(n_lv_tiles, n_dc_tiles) + 1)))) return rc;
  const size_t val_bytes = (size_t)slots * (size_t)esz + 64;
  const bool nullable = col->max_def > 0;
  if (levels_only) {
    // nothing but the level arrays
  } else if (!is_bytes) {
    if ((slots > 0 || !h->values.fixed) && (rc = h->values.ensure(val_bytes))) return rc;   // (a chunk without slots writes no values: a caller may pass none)
    if (nullable && (rc = h->dense.ensure(val_bytes))) return rc;
    if (nullable && slots > 0 && (rc = h->scratch.ensure(12 * ((size_t)((slots + 4095) / 4096) + (size_t)n_def_tiles) + 128))) return rc;
    if (scatter_bits && (rc = h->lvbits.ensure(align_up((size_t)n_def_tiles * (kTile / 8 + 4) + 64, 256) + scatter_plan_bytes(n_def_tiles) + 64))) return rc;
    if (any_dlba || any_dba) {   // FIXED_LEN_BYTE_ARRAY pages under the delta byte-array encodings: the work arrays of the BYTE_ARRAY path
      const size_t n8 = 8 * ((size_t)slots + 2) + 64;
      if ((rc = h->pairs.ensure(n8)) || (rc = h->doff.ensure(n8)) || (rc = h->lens64.ensure(n8))) return rc;
      if ((rc = h->dba_flags.ensure(align_up(4 * (size_t)nd + 64, 256) + align_up(8 * (2 * (size_t)nd + 2) + 64, 256) + copy_plan_bytes(nd) + 64))) return rc;
      if (any_dba && ((rc = h->prefix64.ensure(n8)) ||
                      (rc = h->parent.ensure(4 * ((size_t)slots + 2) + 64)))) return rc;
    }
  } else {
    const size_t n8 = 8 * ((size_t)slots + 2) + 64;
    if ((rc = h->pairs.ensure(n8)) || (rc = h->doff.ensure(n8)) || (rc = h->offsets.ensure(n8))) return rc;
    if ((any_dlba || any_dba) && (rc = h->lens64.ensure(n8))) return rc;
    if (any_dba && ((rc = h->prefix64.ensure(n8)) || (rc = h->parent.ensure(4 * ((size_t)slots + 2) + 64)))) return rc;
    if (any_dlba && (rc = h->parent.ensure(8 * ((size_t)nd + 2) + 64))) return rc;      // (page bases of a chunk of DELTA_LENGTH_BYTE_ARRAY pages only)
    if ((rc = h->scratch.ensure(std::max<size_t>(8 * ((size_t)(slots + 1 + 4095) / 2048 + 8) + 64, len_scan_work_bytes(slots) + 64)))) return rc;
    if ((rc = h->dba_flags.ensure(align_up(4 * (size_t)nd + 64, 256) + align_up(8 * (2 * (size_t)nd + 2) + 64, 256) + copy_plan_bytes(nd) + 64))) return rc;
    // the byte output: every PLAIN / DELTA_LENGTH value lies inside its page, so the page bytes bound it; dictionary
    // and DELTA_BYTE_ARRAY pages may expand beyond that and are then copied again by pqd_wait
    if ((rc = h->bytes.ensure((size_t)all_bytes + 64))) return rc;
  }
  if (!levels_only && (any_delta || any_dlba || any_dba) &&
      (rc = h->delta_blocks.ensure(delta_work_bytes(slots, nd)))) return rc;
  if (col->max_def > 0 && (rc = h->def.ensure(4 * (size_t)slots + 64))) return rc;
  if (col->max_rep > 0 && (rc = h->rep.ensure(4 * (size_t)slots + 64))) return rc;
  if (!levels_only) {
    // (at least one entry, so that a page of an empty dictionary never reads outside the table before its
    //  out-of-range index is reported)
    if (!is_bytes && (rc = h->dict.ensure((size_t)std::max<int64_t>(dict_total, 1) * (size_t)std::max(esz, 1) + 64))) return rc;
    if (is_bytes && (rc = h->dict_pairs.ensure(8 * (size_t)std::max<int64_t>(dict_total, 1) + 64))) return rc;
  }

  // PQD_OPT_POISON: every work and output buffer starts the call filled with 0xFF, so that a kernel reading what an
  // earlier call (or a failed page) left behind shows up as a difference from the reference or as a refused index
  if (h->opt[PQD_OPT_POISON] > 0) {
    for (DevBuf *b : h->all_bufs_)
      if (b != &h->in && b->p && b->cap) {
        // (a buffer of the caller is recorded 64 bytes longer than it is - slack that no kernel writes - and must not be filled past its end)
        const size_t n = b->fixed ? (b->cap > 64 ? b->cap - 64 : 0) : b->cap;
        if (n) HIP_TRY(hipMemsetAsync(b->p, 0xFF, n, h->stream));
      }
  }

  auto t_ensure = tnow();
  h->levels_only = levels_only;
  h->n_events = 0;
  h->col = *col;
  h->n_pages = n_pages;
  h->elem_size = esz;
  h->nd = nd;
  h->total_slots = slots;
  h->bytes_len = 0;
  {
    uint8_t *m = static_cast<uint8_t *>(h->stage_meta.p);
    // (page table, seeds and tile prefixes were built in place)
    if (n_dict) std::memcpy(m + off_dictp, h->h_dict_pages.data(), 4 * (size_t)n_dict);
    // tile -> stream maps of the two hybrid expansions, written in place
    int32_t *ts_lv = reinterpret_cast<int32_t *>(m + off_ts_lv), *ts_dc = reinterpret_cast<int32_t *>(m + off_ts_dc);
    for (size_t i = 0; i + 1 < tp_levels.size() && !device_tables; i++)
      for (int32_t t = tp_levels[i]; t < tp_levels[i + 1]; t++) ts_lv[t] = (int32_t)i;
    for (size_t i = 0; i + 1 < tp_dict.size() && !device_tables; i++)
      for (int32_t t = tp_dict[i]; t < tp_dict[i + 1]; t++) ts_dc[t] = (int32_t)i;
  }
  auto t_stage = tnow();
  // (page entries that went ahead and whose block is still where it was are not sent again)
  const size_t sent = early_done && early_meta == h->meta.p ? sizeof(PageIn) * (size_t)early_done : 0;
  if (device_tables) {      // the page table and the list of dictionary pages; the tables between them are the kernel's to write
    uint8_t *m = static_cast<uint8_t *>(h->stage_meta.p);
    if (pages_bytes > sent) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(h->meta.p) + sent, m + sent, pages_bytes - sent, hipMemcpyHostToDevice, h->stream));
    if (n_dict) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(h->meta.p) + off_dictp, m + off_dictp, 4 * (size_t)n_dict, hipMemcpyHostToDevice, h->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(h->meta.p) + sent, static_cast<uint8_t *>(h->stage_meta.p) + sent, upload_bytes - sent, hipMemcpyHostToDevice, h->stream));
  }
  auto t_h2d = tnow();

  uint8_t *mb = static_cast<uint8_t *>(h->meta.p);
  ChunkCtx c;
  c.in = base; c.in_len = (int64_t)span;
  c.pages = reinterpret_cast<const PageIn *>(mb + off_pages);
  c.state = reinterpret_cast<PageState *>(mb + off_state);
  c.streams = reinterpret_cast<Stream *>(mb + off_streams);
  c.seeds = reinterpret_cast<const StreamSeed *>(mb + off_seeds);
  c.runs = static_cast<RunEntry *>(h->runs.p);
  c.tile_run = static_cast<int32_t *>(h->tile_run.p);
  c.n_pages = n_pages; c.nd = nd; c.max_def = col->max_def; c.max_rep = col->max_rep;
  c.physical_type = col->physical_type; c.type_length = col->type_length; c.elem_size = esz;
  c.chunk_status = static_cast<int32_t *>(h->small.p) + kSmallStatus;
  h->ctx = c;
  const int32_t *d_tp_lv = reinterpret_cast<const int32_t *>(mb + off_tp_lv);
  const int32_t *d_tp_dc = reinterpret_cast<const int32_t *>(mb + off_tp_dc);
  const int32_t *d_tp_pl = reinterpret_cast<const int32_t *>(mb + off_tp_pl);
  const int32_t *d_ts_lv = reinterpret_cast<const int32_t *>(mb + off_ts_lv);
  const int32_t *d_ts_dc = reinterpret_cast<const int32_t *>(mb + off_ts_dc);
  const int32_t *d_dictp = reinterpret_cast<const int32_t *>(mb + off_dictp);
  int64_t *d_totals = static_cast<int64_t *>(h->small.p);

  uint8_t *dense_out = static_cast<uint8_t *>(nullable ? h->dense.p : h->values.p);
  void *dtab = h->opt[PQD_OPT_DELTA_BLOCKS] == 0 ? nullptr : h->delta_blocks.p;
  const bool dspec = h->opt[PQD_OPT_DELTA_BLOCKS] != 2;     // (2: every stream's block headers are walked)

  // LDS reserved per tile for packed bytes.  Levels: the width is known here.  Dictionary indices: the width is
  // inside the page, so size it from the densest page's bytes per value (a tile that needs more still decodes,
  // on a slower path); nullable columns spend their page bytes on fewer values, so they get the maximum.
  auto bit_len = [](int x) { int n = 0; while (x) { n++; x >>= 1; } return n; };
  const int w_lv = std::max(bit_len(col->max_def), bit_len(col->max_rep));
  const int stage_levels = kTile * w_lv / 8 + kTile / 4 + 64;
  int stage_dict = 18 * 1024;
  if (!nullable) stage_dict = (int)(index_bpv * kTile * 1.25) + 512;      // (added up with the page table)

  // Few long streams: a workgroup per stream takes the decoding off the serial header walk.  Many short ones
  // (small pages): a wavefront per stream keeps more of them in flight.
  const bool shared_gpu = inflight_others(h) >= 2;
  bool wide_dict = !shared_gpu && nd > 0 && nd <= 4 * h->n_cu && data_bytes / nd >= 64 * 1024;
  bool wide_levels = wide_dict && slots / nd >= 256 * 1024;
  if (h->opt[PQD_OPT_PRESCAN] >= 0) wide_dict = wide_levels = h->opt[PQD_OPT_PRESCAN] == 1;
  // Level streams of nullable and nested columns are where runs of a few values each are the rule; dictionary
  // indices of low-cardinality columns have them too (the variant costs 6 KiB of LDS per stream).
  const bool jump = h->opt[PQD_OPT_JUMP] != 0;
  // large pages: the index stream of a page walked and expanded by one workgroup, in one pass over its bytes
  constexpr int kFusedMinLen = 16 * 1024;
  bool fused = nd > 0 && data_bytes / nd >= 2 * kFusedMinLen;
  if (h->opt[PQD_OPT_FUSED] >= 0) fused = h->opt[PQD_OPT_FUSED] == 1;

  // hand-over of the dictionary expansion to the device's shared stream and back
  hipStream_t heavy = h->no_alloc ? nullptr : heavy_stream(h->device);
  if (!heavy || !h->ev_ready || !h->ev_done) heavy = h->stream;
  // (after the one-pass kernel the tile expansion has little or nothing left to do: it stays on the handle's stream,
  //  and for a mid-size dictionary it takes the plain gather - the variant that keeps the dictionary's head in LDS is
  //  made of workgroups that need a CU to themselves, and on a busy GPU a launch that finds nothing left would wait for
  //  CUs to drain before it could return)
  const bool fused_ran = fused && any_dict_enc && (is_bytes || esz == 4 || esz == 8);
  const int split_mode = fused_ran && h->opt[PQD_OPT_SPLIT] < 0 ? 0 : h->opt[PQD_OPT_SPLIT];
  if (fused_ran) heavy = h->stream;
  const bool jump_dict = jump;
  auto to_heavy = [&]() -> hipError_t {
    if (heavy == h->stream) return hipSuccess;
    hipError_t e = hipEventRecord(h->ev_ready, h->stream);
    return e != hipSuccess ? e : hipStreamWaitEvent(heavy, h->ev_ready, 0);
  };
  auto from_heavy = [&]() -> hipError_t {
    if (heavy == h->stream) return hipSuccess;
    hipError_t e = hipEventRecord(h->ev_done, heavy);
    return e != hipSuccess ? e : hipStreamWaitEvent(h->stream, h->ev_done, 0);
  };

  // Expansion of the index streams: one launch per run of data pages that share a dictionary (the launch keeps its
  // dictionary in LDS where it fits); Encoding_RLE pages ride along with whatever launch covers their tiles.
  auto expand_dict_groups = [&](const ChunkCtx &cc, const uint8_t *table, int entry_size, uint8_t *out) -> int {
    int64_t t_lo = 0;      // (index tiles are numbered in page order; the groups were found with the page table)
    for (const pqd_handle::DictGroup &g : h->h_dict_groups) {
      const int64_t t_hi = t_lo + g.tiles;
      if (t_hi > t_lo) {
        const PageIn &dp = h->h_pages[(size_t)(g.dict_of >= 0 ? g.dict_of : g.first_page)];
        const uint8_t *dptr = g.dict_of >= 0 ? table + (size_t)dp.dict_off * (size_t)entry_size : table;
        const int dn = g.dict_of >= 0 ? dp.dict_n : -1;   // -1: no dictionary in effect
        HIP_TRY(launch_expand_dict(cc, d_tp_dc, d_ts_dc, (int)t_lo, (int)t_hi, stage_dict, dptr, dn, out, static_cast<TileDesc *>(h->tile_desc.p), h->n_cu,
                                   split_mode, heavy));
      }
      t_lo = t_hi;
    }
    return PQD_OK;
  };

  // ---- kernels ----
  if (n_pages > 0) {
    if (device_tables) {
      Prof pf(h, "k_stream_tables");
      HIP_TRY(launch_stream_tables(c, tbases, mb + off_tbsum, reinterpret_cast<StreamSeed *>(mb + off_seeds), const_cast<int32_t *>(d_tp_lv), const_cast<int32_t *>(d_tp_dc),
                                   const_cast<int32_t *>(d_tp_pl), const_cast<int32_t *>(d_ts_lv), const_cast<int32_t *>(d_ts_dc), h->stream));
    }
    { Prof pf(h, "k_page_setup"); HIP_TRY(launch_page_setup(c, h->stream)); }
    if ((col->max_rep > 0 || col->max_def > 0) && nd > 0) {
      if (direct_rle) { Prof pf(h, "k_rle_only_pages(levels)"); HIP_TRY(launch_rle_only_levels(c, static_cast<int32_t *>(h->rep.p), static_cast<int32_t *>(h->def.p), h->stream)); }
      { Prof pf(h, "k_hybrid_prescan(levels)"); HIP_TRY(launch_hybrid_prescan(c, 0, 2 * nd, wide_levels, jump, h->stream)); }
      uint32_t *vbits = scatter_bits ? static_cast<uint32_t *>(h->lvbits.p) : nullptr;
      int32_t *vcnt = scatter_bits ? reinterpret_cast<int32_t *>(static_cast<uint8_t *>(h->lvbits.p) + (size_t)n_def_tiles * (kTile / 8)) : nullptr;
      if (scatter_bits) HIP_TRY(hipMemsetAsync(vcnt, 0, 4 * (size_t)n_def_tiles, h->stream));
      { Prof pf(h, "k_hybrid_expand<levels>");
        HIP_TRY(launch_expand_levels(c, d_tp_lv, d_ts_lv, n_lv_tiles, stage_levels, static_cast<int32_t *>(h->rep.p), static_cast<int32_t *>(h->def.p), static_cast<TileDesc *>(h->tile_desc.p), h->n_cu,
                                     vbits, vcnt, n_lv_tiles_rep, h->stream)); }
    }
  }
  { Prof pf(h, "k_pages_finalize"); HIP_TRY(launch_pages_finalize(c, d_totals, reinterpret_cast<int64_t *>(mb + off_pgsum), h->stream)); }
  if (levels_only) {
    // skip path (layout/page.go:433-559): levels, null and row counts; the value sections stay as they are
  } else if (is_bytes) {
    // ---- BYTE_ARRAY: value descriptors (pairs) per encoding, offsets by a scan over all slots, then the copies ----
    ChunkCtx c8 = c;
    c8.elem_size = 8;
    uint64_t *pairs = static_cast<uint64_t *>(h->pairs.p);
    int64_t *lens64 = static_cast<int64_t *>(h->lens64.p);
    int64_t *prefix64 = static_cast<int64_t *>(h->prefix64.p);
    // a chunk of DELTA_LENGTH_BYTE_ARRAY pages only: a string's place in the output follows from the lengths alone, and its
    // bytes from where its page's length stream ended - no descriptor per value is needed
    const bool one_pass = h->opt[PQD_OPT_BYTE_OFFSETS] != 0;
    const bool fast_dlba = one_pass && dtab && any_dlba && !any_dba && !any_plain && !any_dict_enc && host_status == PQD_OK;
    // ... and a chunk of DELTA_BYTE_ARRAY pages only gets by with sums per 64 strings (dba_general_again when a page is not clean)
    const bool fast_dba = one_pass && dtab && any_dba && !any_dlba && !any_plain && !any_dict_enc && host_status == PQD_OK;
    // (both decode their lengths as 32-bit words: what those paths read and write per value is halved)
    if (fast_dlba || fast_dba) HIP_TRY(hipMemsetAsync(c.chunk_status + kLayoutMismatch, 0, 4, h->stream));
    if (!fast_dlba && !fast_dba) HIP_TRY(hipMemsetAsync(pairs, 0, 8 * ((size_t)slots + 2), h->stream));  // failed pages and the slots beyond the values read as empty strings
    // (a dictionary walk that fails part way leaves the remaining entries as they are: they must read as empty
    //  strings, not as whatever the buffer held, because the data pages are still expanded through them)
    if (n_dict) HIP_TRY(hipMemsetAsync(h->dict_pairs.p, 0, 8 * (size_t)std::max<int64_t>(dict_total, 1) + 64, h->stream));
    if (n_dict) { Prof pf(h, "k_plain_ba_walk(dict)"); HIP_TRY(launch_plain_ba_walk(c, d_dictp, n_dict, static_cast<uint64_t *>(h->dict_pairs.p), h->stream)); }
    if (any_dict_enc) {
      if (direct_rle) { Prof pf(h, "k_rle_only_pages"); HIP_TRY(launch_rle_only_pages(c8, static_cast<const uint8_t *>(h->dict_pairs.p), reinterpret_cast<uint8_t *>(pairs), h->stream)); }
      if (fused) { Prof pf(h, "k_page_fused");   // (the dictionary entries are the 8-byte descriptors of the strings)
        HIP_TRY(launch_page_fused(c8, static_cast<const uint8_t *>(h->dict_pairs.p), reinterpret_cast<uint8_t *>(pairs), max_dict_n,
                                  h->opt[PQD_OPT_FUSED] == 1 ? 1 : kFusedMinLen, h->stream)); }
      { Prof pf(h, "k_hybrid_prescan(dict)"); HIP_TRY(launch_hybrid_prescan(c, 2 * nd, nd, wide_dict, jump_dict, h->stream)); }
      HIP_TRY(to_heavy());
      { Prof pf(h, "k_hybrid_expand<dict>", heavy);
        if ((rc = expand_dict_groups(c8, static_cast<const uint8_t *>(h->dict_pairs.p), 8, reinterpret_cast<uint8_t *>(pairs)))) return rc; }
      HIP_TRY(from_heavy());
    }
    if (any_plain) { Prof pf(h, "k_plain_ba_walk"); HIP_TRY(launch_plain_ba_walk(c, nullptr, 0, pairs, h->stream)); }
    if (any_dlba) {
      { Prof pf(h, "k_delta_decode(lengths)"); HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(lens64), fast_dlba ? 4 : 8, PQD_ENC_DELTA_LENGTH_BYTE_ARRAY, 0, dtab, slots, 0.0, dspec, fast_dlba, h->stream)); }
      if (!fast_dlba) { Prof pf(h, "k_dlba_pairs"); HIP_TRY(launch_dlba_pairs(c, PQD_ENC_DELTA_LENGTH_BYTE_ARRAY, lens64, pairs, h->stream)); }
    }
    if (any_dba) {
      if (!fast_dba) HIP_TRY(hipMemsetAsync(prefix64, 0, 8 * ((size_t)slots + 2), h->stream));
      { Prof pf(h, "k_delta_decode(prefix)"); HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(prefix64), fast_dba ? 4 : 8, PQD_ENC_DELTA_BYTE_ARRAY, 0, dtab, slots, 0.0, dspec, fast_dba, h->stream)); }
      { Prof pf(h, "k_delta_decode(lengths)"); HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(lens64), fast_dba ? 4 : 8, PQD_ENC_DELTA_BYTE_ARRAY, 1, dtab, slots, 0.0, dspec, fast_dba, h->stream)); }
      if (!fast_dba) {
        { Prof pf(h, "k_dlba_pairs"); HIP_TRY(launch_dlba_pairs(c, PQD_ENC_DELTA_BYTE_ARRAY, lens64, pairs, h->stream)); }
        { Prof pf(h, "k_dba_parents"); HIP_TRY(launch_dba_resolve(c, prefix64, pairs, static_cast<int32_t *>(h->parent.p), nullptr, static_cast<int32_t *>(h->dba_flags.p), nullptr, 0, nullptr, 0, h->stream)); }
      }
    }
    // Offsets: a scan over all `slots` descriptors.  The dense values come first; the descriptors behind them are
    // the zeros written above, so the entries from the number of values on all equal the byte total.
    int64_t *doff = static_cast<int64_t *>(nullable ? h->doff.p : h->offsets.p);
    if (fast_dlba) {
      // page by page, every page starting where the payload sizes of the pages in front put it; should a page hold bytes
      // behind its last string, pqd_wait computes the offsets again from the lengths alone (dlba_offsets_again)
      Prof pf(h, "k_dlba_offsets");
      HIP_TRY(launch_dlba_offsets(c, reinterpret_cast<const int32_t *>(lens64), static_cast<int64_t *>(h->parent.p), d_totals + 1, slots, nullable, doff, h->stream));
    } else if (fast_dba) {
      Prof pf(h, "k_dba_offsets");
      HIP_TRY(launch_dba_lean_offsets(c, reinterpret_cast<int32_t *>(prefix64), reinterpret_cast<const int32_t *>(lens64), pairs, static_cast<int32_t *>(h->parent.p), dba_page_work(h), d_totals + 1, slots, nullable,
                                      doff, h->stream));
    } else {
      HIP_TRY(launch_void_failed_pages(c, pairs, any_dba ? prefix64 : nullptr, h->stream));
      Prof pf(h, one_pass ? "k_len_scan" : "k_len_offsets");
      if (one_pass) HIP_TRY(launch_len_scan(c, pairs, any_dba ? prefix64 : nullptr, nullptr, nullptr, slots, h->scratch.p, doff, nullptr, nullptr, h->stream));
      else HIP_TRY(launch_len_offsets(pairs, any_dba ? prefix64 : nullptr, slots, static_cast<int64_t *>(h->scratch.p), doff, h->stream));
    }
    HIP_TRY(hipMemcpyAsync(d_totals + kSmallByteTotal, doff + slots, 8, hipMemcpyDeviceToDevice, h->stream));
    h->plan.active = true; h->plan.any_plain = any_plain; h->plan.any_dict_enc = any_dict_enc; h->plan.any_dlba = any_dlba;
    h->plan.any_dba = any_dba; h->plan.nullable = nullable; h->plan.slots = slots; h->plan.fast_dlba = fast_dlba; h->plan.fast_dba = fast_dba; h->plan.max_page_bytes = max_page_bytes;
    if ((rc = launch_byte_copies(h))) return rc;
    if (nullable) {
      const int n_tiles = (int)((slots + 1 + 4095) / 4096);
      Prof pf(h, "k_scatter_offsets");
      HIP_TRY(launch_scatter_rank(static_cast<const int32_t *>(h->def.p), slots, col->max_def, static_cast<int64_t *>(h->scratch.p), n_tiles, h->stream));
      HIP_TRY(launch_scatter_offsets(static_cast<const int32_t *>(h->def.p), slots, col->max_def, static_cast<const int64_t *>(h->scratch.p), doff,
                                     static_cast<int64_t *>(h->offsets.p), h->stream));
    }
  } else {
    // every dictionary page is decoded as PLAIN whatever its header says (layout/page.go:802)
    if (n_dict) { Prof pf(h, "k_dict_plain_fixed"); HIP_TRY(launch_dict_plain_fixed(c, d_dictp, n_dict, static_cast<uint8_t *>(h->dict.p), h->stream)); }
    if (any_dict_enc || any_rle) {
      if (direct_rle && any_dict_enc && (esz == 4 || esz == 8)) { Prof pf(h, "k_rle_only_pages"); HIP_TRY(launch_rle_only_pages(c, static_cast<const uint8_t *>(h->dict.p), dense_out, h->stream)); }
      if (fused && any_dict_enc && (esz == 4 || esz == 8)) { Prof pf(h, "k_page_fused");
        HIP_TRY(launch_page_fused(c, static_cast<const uint8_t *>(h->dict.p), dense_out, max_dict_n, h->opt[PQD_OPT_FUSED] == 1 ? 1 : kFusedMinLen, h->stream)); }
      { Prof pf(h, "k_hybrid_prescan(dict)"); HIP_TRY(launch_hybrid_prescan(c, 2 * nd, nd, wide_dict, jump_dict, h->stream)); }
      HIP_TRY(to_heavy());
      { Prof pf(h, "k_hybrid_expand<dict>", heavy);
        if ((rc = expand_dict_groups(c, static_cast<const uint8_t *>(h->dict.p), esz, dense_out))) return rc; }
      HIP_TRY(from_heavy());
    }
    if (any_plain && !plain_scatter) {
      if (col->physical_type == PQD_BOOLEAN) { Prof pf(h, "k_plain_boolean"); HIP_TRY(launch_plain_boolean(c, d_tp_pl, n_pl_tiles, dense_out, h->stream)); }
      else { Prof pf(h, "k_plain_fixed"); HIP_TRY(launch_plain_fixed(c, d_tp_pl, n_pl_tiles, max_page_bytes, dense_out, h->stream)); }
    }
    if (any_bss) { Prof pf(h, "k_byte_stream_split"); HIP_TRY(launch_byte_stream_split(c, d_tp_pl, n_pl_tiles, dense_out, h->stream)); }
    if (any_delta) {
      // what the DELTA_BINARY_PACKED pages spend per value (levels included, which only makes the units smaller)
      int64_t db = 0, dv = 0;
      for (int p = 0; p < n_pages; p++) {
        const PageIn &pi = h->h_pages[(size_t)p];
        if (pi.data_idx >= 0 && pi.encoding == PQD_ENC_DELTA_BINARY_PACKED && pi.num_values > 0) { db += pi.body_len; dv += pi.num_values; }
      }
      Prof pf(h, "k_delta_decode");
      HIP_TRY(launch_delta_decode(c, dense_out, esz, PQD_ENC_DELTA_BINARY_PACKED, 0, dtab, slots, dv > 0 ? (double)db / (double)dv : 0.0, dspec, false, h->stream));
    }
    if (any_dlba || any_dba) {
      // FIXED_LEN_BYTE_ARRAY under DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY: the kernels of the BYTE_ARRAY path, their
      // offsets fixed at type_length bytes per value, copy straight into the column's image
      uint64_t *pairs = static_cast<uint64_t *>(h->pairs.p);
      int64_t *lens64 = static_cast<int64_t *>(h->lens64.p);
      int64_t *prefix64 = static_cast<int64_t *>(h->prefix64.p);
      int64_t *doff = static_cast<int64_t *>(h->doff.p);
      HIP_TRY(hipMemsetAsync(pairs, 0, 8 * ((size_t)slots + 2), h->stream));
      if (any_dlba) {
        { Prof pf(h, "k_delta_decode(lengths)"); HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(lens64), 8, PQD_ENC_DELTA_LENGTH_BYTE_ARRAY, 0, dtab, slots, 0.0, dspec, false, h->stream)); }
        { Prof pf(h, "k_dlba_pairs"); HIP_TRY(launch_dlba_pairs(c, PQD_ENC_DELTA_LENGTH_BYTE_ARRAY, lens64, pairs, h->stream)); }
      }
      if (any_dba) {
        HIP_TRY(hipMemsetAsync(prefix64, 0, 8 * ((size_t)slots + 2), h->stream));
        { Prof pf(h, "k_delta_decode(prefix)"); HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(prefix64), 8, PQD_ENC_DELTA_BYTE_ARRAY, 0, dtab, slots, 0.0, dspec, false, h->stream)); }
        { Prof pf(h, "k_delta_decode(lengths)"); HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(lens64), 8, PQD_ENC_DELTA_BYTE_ARRAY, 1, dtab, slots, 0.0, dspec, false, h->stream)); }
        { Prof pf(h, "k_dlba_pairs"); HIP_TRY(launch_dlba_pairs(c, PQD_ENC_DELTA_BYTE_ARRAY, lens64, pairs, h->stream)); }
        { Prof pf(h, "k_dba_parents"); HIP_TRY(launch_dba_resolve(c, prefix64, pairs, static_cast<int32_t *>(h->parent.p), nullptr, static_cast<int32_t *>(h->dba_flags.p), nullptr, 0, nullptr, 0, h->stream)); }
      }
      { Prof pf(h, "k_flba_delta_check"); HIP_TRY(launch_flba_delta_offsets(c, pairs, any_dba ? prefix64 : nullptr, slots, doff, h->stream)); }
      const int64_t cap = slots * (int64_t)esz;
      if (any_dlba) { Prof pf(h, "k_copy_dlba_pages"); HIP_TRY(launch_copy_dlba_pages(c, pairs, doff, dense_out, doff + slots, cap, max_page_bytes, copy_plan(h), h->stream)); }
      if (any_dba) { Prof pf(h, "k_dba_resolve");
        HIP_TRY(launch_dba_resolve(c, prefix64, pairs, static_cast<int32_t *>(h->parent.p), doff, static_cast<int32_t *>(h->dba_flags.p), dense_out, 1, doff + slots, cap, h->stream)); }
    }
    if (nullable && slots > 0) {
      Prof pf(h, scatter_bits ? "k_scatter(plain pages, validity bits)" : plain_scatter ? "k_scatter(plain pages)" : "k_scatter");
      if (scatter_bits)
        HIP_TRY(launch_scatter_plain_bits(c, d_tp_lv, d_ts_lv, n_lv_tiles_rep, n_def_tiles, static_cast<const uint32_t *>(h->lvbits.p),
                                          reinterpret_cast<const int32_t *>(static_cast<const uint8_t *>(h->lvbits.p) + (size_t)n_def_tiles * (kTile / 8)),
                                          static_cast<int64_t *>(h->scratch.p),
                                          static_cast<uint8_t *>(h->lvbits.p) + align_up((size_t)n_def_tiles * (kTile / 8 + 4) + 64, 256),
                                          static_cast<uint8_t *>(h->values.p), h->stream));
      else if (plain_scatter)
        HIP_TRY(launch_scatter_plain(c, static_cast<const int32_t *>(h->def.p), static_cast<uint8_t *>(h->values.p), slots,
                                     static_cast<int64_t *>(h->scratch.p), h->stream));
      else
        HIP_TRY(launch_scatter_fixed(c, static_cast<const int32_t *>(h->def.p), static_cast<const uint8_t *>(h->dense.p),
                                     static_cast<uint8_t *>(h->values.p), slots, static_cast<int64_t *>(h->scratch.p), h->stream));
    }
  }
  { Prof pf(h, "k_collect_status"); HIP_TRY(launch_collect_status(c, c.chunk_status, h->stream)); }
  if (!h->no_alloc) HIP_TRY(hipMemcpyAsync(h->host_small.p, h->small.p, 256, hipMemcpyDeviceToHost, h->stream));

  h->result = pqd_chunk_result{};
  h->result.status = host_status;
  h->result.bad_page = host_bad;
  set_busy(h, true);
  if (host_timing) {
    auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
    auto t_end = tnow();
    std::fprintf(stderr, "pqd host: pages %.1f us, seeds %.1f, ensure %.1f, staging %.1f, h2d call %.1f, launches %.1f (n_pages %d, upload %zu bytes)\n",
                 us(t_begin, t_pages), us(t_pages, t_seeds), us(t_seeds, t_ensure), us(t_ensure, t_stage), us(t_stage, t_h2d), us(t_h2d, t_end), n_pages, device_tables ? pages_bytes + 4 * (size_t)n_dict : upload_bytes);
  }
  return PQD_OK;
}

// A chunk of DELTA_LENGTH_BYTE_ARRAY pages whose offsets were taken from the pages' payload sizes, one of which turned out to
// hold bytes behind its last string (the reference never looks at them): the offsets once more from the lengths alone, then
// what depends on them.
int dlba_offsets_again(pqd_handle *h) {
  const ChunkCtx &c = h->ctx;
  const int64_t slots = h->plan.slots;
  int64_t *d_totals = static_cast<int64_t *>(h->small.p);
  int64_t *doff = static_cast<int64_t *>(h->plan.nullable ? h->doff.p : h->offsets.p);
  // (the lengths were kept as 32-bit words: once more as the 64-bit words the kernels below read - and should the flag have
  //  been raised by a block that does not fit 32-bit arithmetic, these are the values the reference computes)
  void *dtab = h->opt[PQD_OPT_DELTA_BLOCKS] == 0 ? nullptr : h->delta_blocks.p;
  HIP_TRY(hipMemsetAsync(c.chunk_status + kLayoutMismatch, 0, 4, h->stream));      // (the copies below look at it)
  HIP_TRY(launch_delta_decode(c, static_cast<uint8_t *>(h->lens64.p), 8, PQD_ENC_DELTA_LENGTH_BYTE_ARRAY, 0, dtab, slots, 0.0,
                              h->opt[PQD_OPT_DELTA_BLOCKS] != 2, false, h->stream));
  HIP_TRY(launch_len_scan(c, nullptr, nullptr, static_cast<const int64_t *>(h->lens64.p), d_totals + 1, slots, h->scratch.p, doff, nullptr,
                          static_cast<int32_t *>(h->dba_flags.p), h->stream));
  HIP_TRY(hipMemcpyAsync(d_totals + kSmallByteTotal, doff + slots, 8, hipMemcpyDeviceToDevice, h->stream));
  int rc;
  if ((rc = launch_byte_copies(h))) return rc;
  if (h->plan.nullable) {
    const int n_tiles = (int)((slots + 1 + 4095) / 4096);
    HIP_TRY(launch_scatter_rank(static_cast<const int32_t *>(h->def.p), slots, h->col.max_def, static_cast<int64_t *>(h->scratch.p), n_tiles, h->stream));
    HIP_TRY(launch_scatter_offsets(static_cast<const int32_t *>(h->def.p), slots, h->col.max_def, static_cast<const int64_t *>(h->scratch.p), doff,
                                   static_cast<int64_t *>(h->offsets.p), h->stream));
  }
  HIP_TRY(launch_collect_status(c, c.chunk_status, h->stream));
  HIP_TRY(hipMemcpyAsync(h->host_small.p, h->small.p, 256, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return PQD_OK;
}

// A chunk of DELTA_BYTE_ARRAY pages only that turned out not to be clean (a page that failed, a length out of range, a string
// above the in-order chain's limit, suffixes running past their page): the kernels that keep a descriptor per value decide
// what every page is, starting from the decoded lengths again.
int dba_general_again(pqd_handle *h) {
  const ChunkCtx &c = h->ctx;
  const int64_t slots = h->plan.slots;
  int64_t *d_totals = static_cast<int64_t *>(h->small.p);
  uint64_t *pairs = static_cast<uint64_t *>(h->pairs.p);
  int64_t *lens64 = static_cast<int64_t *>(h->lens64.p), *prefix64 = static_cast<int64_t *>(h->prefix64.p);
  int64_t *doff = static_cast<int64_t *>(h->plan.nullable ? h->doff.p : h->offsets.p);
  void *dtab = h->opt[PQD_OPT_DELTA_BLOCKS] == 0 ? nullptr : h->delta_blocks.p;
  const bool dspec = h->opt[PQD_OPT_DELTA_BLOCKS] != 2;
  h->plan.fast_dba = false;
  HIP_TRY(hipMemsetAsync(c.chunk_status + kLayoutMismatch, 0, 4, h->stream));
  HIP_TRY(hipMemsetAsync(pairs, 0, 8 * ((size_t)slots + 2), h->stream));
  HIP_TRY(hipMemsetAsync(prefix64, 0, 8 * ((size_t)slots + 2), h->stream));
  HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(prefix64), 8, PQD_ENC_DELTA_BYTE_ARRAY, 0, dtab, slots, 0.0, dspec, false, h->stream));
  HIP_TRY(launch_delta_decode(c, reinterpret_cast<uint8_t *>(lens64), 8, PQD_ENC_DELTA_BYTE_ARRAY, 1, dtab, slots, 0.0, dspec, false, h->stream));
  HIP_TRY(launch_dlba_pairs(c, PQD_ENC_DELTA_BYTE_ARRAY, lens64, pairs, h->stream));
  HIP_TRY(launch_dba_resolve(c, prefix64, pairs, static_cast<int32_t *>(h->parent.p), nullptr, static_cast<int32_t *>(h->dba_flags.p), nullptr, 0, nullptr, 0, h->stream));
  HIP_TRY(launch_void_failed_pages(c, pairs, prefix64, h->stream));
  HIP_TRY(launch_len_scan(c, pairs, prefix64, nullptr, nullptr, slots, h->scratch.p, doff, nullptr, nullptr, h->stream));
  HIP_TRY(hipMemcpyAsync(d_totals + kSmallByteTotal, doff + slots, 8, hipMemcpyDeviceToDevice, h->stream));
  int rc;
  if ((rc = launch_byte_copies(h))) return rc;
  if (h->plan.nullable) {
    const int n_tiles = (int)((slots + 1 + 4095) / 4096);
    HIP_TRY(launch_scatter_rank(static_cast<const int32_t *>(h->def.p), slots, h->col.max_def, static_cast<int64_t *>(h->scratch.p), n_tiles, h->stream));
    HIP_TRY(launch_scatter_offsets(static_cast<const int32_t *>(h->def.p), slots, h->col.max_def, static_cast<const int64_t *>(h->scratch.p), doff,
                                   static_cast<int64_t *>(h->offsets.p), h->stream));
  }
  HIP_TRY(launch_collect_status(c, c.chunk_status, h->stream));
  HIP_TRY(hipMemcpyAsync(h->host_small.p, h->small.p, 256, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return PQD_OK;
}

// Completes the call in flight: totals, status, and for BYTE_ARRAY the byte copies once more when the output was
// sized too small.
int finish_call(pqd_handle *h) {
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->no_alloc) HIP_TRY(hipMemcpy(h->host_small.p, h->small.p, 256, hipMemcpyDeviceToHost));
  if (h->plan.active && h->plan.fast_dlba && (static_cast<const int32_t *>(h->host_small.p) + kSmallStatus)[kLayoutMismatch] != 0) {
    int rc = dlba_offsets_again(h);
    if (rc) return rc;
  }
  if (h->plan.active && h->plan.fast_dba && (static_cast<const int32_t *>(h->host_small.p) + kSmallStatus)[kLayoutMismatch] != 0) {
    int rc = dba_general_again(h);
    if (rc) return rc;
  }
  set_busy(h, false);
  const int64_t *tot = static_cast<const int64_t *>(h->host_small.p);
  const int32_t *st = static_cast<const int32_t *>(h->host_small.p) + kSmallStatus;
  pqd_chunk_result &r = h->result;
  r.num_slots = tot[0]; r.num_values = tot[1]; r.num_nulls = tot[2]; r.num_rows = tot[3];
  r.elem_size = h->elem_size;
  r.has_levels = (h->col.max_def > 0 ? 1 : 0) | (h->col.max_rep > 0 ? 2 : 0);
  // the first failing page in page order, whether the host or a kernel found it (a kernel's finding wins on the
  // same page: what the page's own bytes do to the reference comes before what its caller does with the result)
  if (r.status == PQD_OK) { r.status = st[0]; r.bad_page = st[1]; }
  else if (st[0] != 0 && st[1] >= 0 && (r.bad_page < 0 || st[1] <= r.bad_page)) { r.status = st[0]; r.bad_page = st[1]; }
  if (r.status == PQD_OK) r.bad_page = -1;
  if (h->plan.active) {
    const int64_t total = tot[kSmallByteTotal];
    if (r.status == PQD_OK) {
      if (total < 0 || total > ((int64_t)1 << 40)) { g_last_error = "implausible byte total"; r.status = PQD_ERR_MALFORMED; r.bad_page = 0; }
      else if (total > (int64_t)h->bytes.cap - 64) {
        int rc;
        if ((rc = h->bytes.ensure((size_t)total + 64))) return rc;
        if ((rc = launch_byte_copies(h))) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
      }
      h->bytes_len = r.status == PQD_OK ? total : 0;
    }
  }
  r.bytes_len = h->bytes_len;
  h->have_result = true;
  return PQD_OK;
}

// chunk bytes in host memory -> the handle's device input buffer through page-locked staging
int stage_chunks(pqd_handle *h, int32_t n_chunks, const uint8_t *const *chunks, const size_t *lens, std::vector<int64_t> &chunk_off,
                 size_t &span) {
  chunk_off.resize((size_t)n_chunks);
  size_t total = 0;
  for (int k = 0; k < n_chunks; k++) {
    if (lens[k] > 0 && !chunks[k]) return fail_arg("NULL chunk");
    chunk_off[(size_t)k] = (int64_t)total;
    total += align_up(lens[k] + 64, 256);
  }
  span = total;
  int rc;
  if ((rc = h->in.ensure(std::max<size_t>(total, 256)))) return rc;
  // bytes the caller already put into one of the handle's input buffers are sent as they are
  int k_in = -1;
  if (n_chunks == 1 && lens[0] > 0)
    for (int k = 0; k < 2; k++) if (chunks[0] == h->stage_in[k].p && h->stage_in[k].cap >= total) k_in = k;
  const int s = k_in >= 0 ? k_in : h->stage_next;
  if (k_in < 0) {
    // the caller's memory may not be retained (cgo rule): copy into page-locked staging, then DMA.  Two staging
    // buffers take turns, so this copy only ever waits for the transfer out of the same buffer two calls ago.
    if (h->h2d_pending[s]) { HIP_TRY(hipEventSynchronize(h->ev_h2d[s])); h->h2d_pending[s] = false; }
    if ((rc = h->stage_in[s].ensure(std::max<size_t>(total, 256)))) return rc;
    h->stage_next = s ^ 1;
  }
  uint8_t *st = static_cast<uint8_t *>(h->stage_in[s].p);
  for (int k = 0; k < n_chunks; k++) {
    uint8_t *dst = st + chunk_off[(size_t)k];
    if (k_in < 0 && lens[k]) std::memcpy(dst, chunks[k], lens[k]);
    const size_t padded = align_up(lens[k] + 64, 256);
    std::memset(dst + lens[k], 0, padded - lens[k]);
  }
  if (total) HIP_TRY(hipMemcpyAsync(h->in.p, st, total, hipMemcpyHostToDevice, h->stream));
  if (h->ev_h2d[s] && hipEventRecord(h->ev_h2d[s], h->stream) == hipSuccess) h->h2d_pending[s] = true;
  else HIP_TRY(hipStreamSynchronize(h->stream));
  return PQD_OK;
}

int decode_host(pqd_handle *h, int32_t n_chunks, const uint8_t *const *chunks, const size_t *lens, const pqd_page_desc *pages,
                const int32_t *pages_per_chunk, const pqd_column_desc *col, bool levels_only) {
  if (!h || !col || n_chunks < 0 || (n_chunks > 0 && (!chunks || !lens || !pages_per_chunk))) return fail_arg("NULL argument");
  if (h->no_alloc) return fail_arg("not a handle of this entry point");
  Guard g(h->device);
  if (h->busy) { HIP_TRY(hipStreamSynchronize(h->stream)); set_busy(h, false); }   // the device input buffer is about to be rewritten
  std::vector<int32_t> first((size_t)n_chunks + 1, 0);
  for (int k = 0; k < n_chunks; k++) {
    if (pages_per_chunk[k] < 0 || first[(size_t)k] > 0x7FFFFFF0 - pages_per_chunk[k]) return fail_arg("bad page counts");
    first[(size_t)k + 1] = first[(size_t)k] + pages_per_chunk[k];
  }
  if (first.back() > 0 && !pages) return fail_arg("NULL argument");
  std::vector<int64_t> chunk_off;
  size_t span = 0;
  int rc;
  if ((rc = stage_chunks(h, n_chunks, chunks, lens, chunk_off, span))) return rc;
  return decode_on_device(h, static_cast<const uint8_t *>(h->in.p), span, pages, first.back(), n_chunks, first.data(), chunk_off.data(), lens,
                          col, levels_only);
}

int decode_device(pqd_handle *h, int32_t n_chunks, const void *const *d_chunks, const size_t *lens, const pqd_page_desc *pages,
                  const int32_t *pages_per_chunk, const pqd_column_desc *col, bool levels_only) {
  if (!h || !col || n_chunks < 0 || (n_chunks > 0 && (!d_chunks || !lens || !pages_per_chunk))) return fail_arg("NULL argument");
  Guard g(h->device);
  std::vector<int32_t> first((size_t)n_chunks + 1, 0);
  std::vector<int64_t> chunk_off((size_t)n_chunks, 0);
  uintptr_t lo = ~(uintptr_t)0, hi = 0;
  for (int k = 0; k < n_chunks; k++) {
    if (pages_per_chunk[k] < 0 || first[(size_t)k] > 0x7FFFFFF0 - pages_per_chunk[k]) return fail_arg("bad page counts");
    first[(size_t)k + 1] = first[(size_t)k] + pages_per_chunk[k];
    if (lens[k] > 0 && !d_chunks[k]) return fail_arg("NULL chunk");
    if (lens[k] == 0) continue;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_chunks[k]);
    if (a & 15u) return fail_arg("device chunk buffers must be 16-byte aligned");
    lo = std::min(lo, a);
    hi = std::max(hi, a + lens[k]);
  }
  if (first.back() > 0 && !pages) return fail_arg("NULL argument");
  if (hi == 0) { lo = 0; }
  for (int k = 0; k < n_chunks; k++) chunk_off[(size_t)k] = lens[k] ? (int64_t)(reinterpret_cast<uintptr_t>(d_chunks[k]) - lo) : 0;
  return decode_on_device(h, reinterpret_cast<const uint8_t *>(lo), hi - lo, pages, first.back(), n_chunks, first.data(), chunk_off.data(), lens,
                          col, levels_only);
}

// ---- the entry point that allocates nothing ----
// Work space of pqd_decode_chunk_device_noalloc: one array per GPU, held from the moment the library is loaded on it
// until the process ends.  Host code never uses the array's name as an address: the address on each device comes
// from hipGetSymbolAddress, once, and is what the kernels and copies get.
constexpr size_t kNoAllocBytes = (size_t)32 << 20;
__device__ __align__(256) unsigned char g_noalloc_scratch[kNoAllocBytes];

struct NoAllocDevice {
  std::mutex mu;          // calls on one GPU share its array: one at a time
  bool ready = false;
  Arena arena;
  pqd_handle h;
};
NoAllocDevice g_noalloc[kMaxDevices];

// with `device` current
int noalloc_setup(NoAllocDevice &na, int device) {
  if (na.ready) return PQD_OK;
  void *ptr = nullptr;
  HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_noalloc_scratch)));
  if (!ptr) return fail_arg("no address for the work space array");
  na.arena.base = static_cast<uint8_t *>(ptr);
  na.arena.cap = kNoAllocBytes;
  na.arena.used = 0;
  pqd_handle &h = na.h;
  h.device = device;
  h.no_alloc = true;
  for (int i = 0; i < PQD_OPT_COUNT; i++) h.opt[i] = defaults().opt[i];
  for (DevBuf *b : h.all_bufs_) b->arena = &na.arena;
  for (DevBuf *b : {&h.values, &h.def, &h.rep}) { b->arena = nullptr; b->fixed = true; }
  h.stage_meta.pageable = true;
  h.host_small.pageable = true;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h.n_cu = prop.multiProcessorCount;
  na.ready = true;
  return PQD_OK;
}

}  // namespace

extern "C" {

const char *pqd_version(void) { return "pqdecode 0.2 (gfx950)"; }
const char *pqd_last_error(void) { return g_last_error.c_str(); }

const char *pqd_strerror(int code) {
  switch (code) {
    case PQD_OK: return "ok";
    case PQD_ERR_EOF: return "EOF";
    case PQD_ERR_UNEXPECTED_EOF: return "unexpected EOF";
    case PQD_ERR_MALFORMED: return "malformed page";
    case PQD_ERR_UNSUPPORTED_ENCODING: return "unsupported encoding";
    case PQD_ERR_UNSUPPORTED_PAGE_TYPE: return "unsupported page type";
    case PQD_ERR_UNSUPPORTED_TYPE: return "unsupported physical type / encoding combination";
    case PQD_ERR_BIT_WIDTH: return "unsupported bit width";
    case PQD_ERR_INVALID_ARG: return "invalid argument";
    case PQD_ERR_HIP: return "HIP runtime error";
    case PQD_ERR_NOMEM: return "out of memory";
    case PQD_ERR_NO_DICTIONARY: return "dictionary-encoded page without a dictionary page";
    case PQD_ERR_THRIFT: return "page header could not be parsed";
    case PQD_ERR_REFUSED: return "malformed input the reference pads or reinterprets; refused";
    case PQD_ERR_CAPACITY: return "output array of the caller too small: grow it and call again";
    default: return "unknown error";
  }
}

int pqd_device_count(int *n_out) {
  if (!n_out) return fail_arg("n_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *n_out = 0; return fail_hip(e, "hipGetDeviceCount"); }
  *n_out = n;
  return PQD_OK;
}

int pqd_handle_create(int device, void *stream, pqd_handle **out) {
  PQD_GUARDED(
  if (!out) return fail_arg("out is NULL");
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail_arg("no such device");
  Guard g(device);
  pqd_handle *h = new pqd_handle();
  h->device = device;
  for (int i = 0; i < PQD_OPT_COUNT; i++) h->opt[i] = defaults().opt[i];
  if (stream) { h->stream = static_cast<hipStream_t>(stream); h->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail_hip(e, "hipStreamCreate"); }
    h->own_stream = true;
  }
  if (hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming) != hipSuccess) h->ev_ready = nullptr;
  if (hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess) h->ev_done = nullptr;
  for (int k = 0; k < 2; k++)
    if (hipEventCreateWithFlags(&h->ev_h2d[k], hipEventDisableTiming) != hipSuccess) h->ev_h2d[k] = nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
  *out = h;
  return PQD_OK;
  )
}

int pqd_handle_destroy(pqd_handle *h) {
  if (!h) return PQD_OK;
  Guard g(h->device);
  (void)hipStreamSynchronize(h->stream);
  set_busy(h, false);
  for (auto &e : h->events) { (void)hipEventDestroy(e.start); (void)hipEventDestroy(e.stop); }
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  if (h->ev_done) (void)hipEventDestroy(h->ev_done);
  for (int k = 0; k < 2; k++) if (h->ev_h2d[k]) (void)hipEventDestroy(h->ev_h2d[k]);
  for (DevBuf *b : h->all_bufs_) b->release();
  h->stage_in[0].release(); h->stage_in[1].release(); h->stage_out.release(); h->stage_meta.release(); h->host_small.release();
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return PQD_OK;
}

int pqd_set_profiling(pqd_handle *h, int on) {
  if (!h) return fail_arg("handle is NULL");
  h->profiling = on != 0;
  return PQD_OK;
}

int pqd_set_option(pqd_handle *h, int option, int value) {
  if (!h) return fail_arg("handle is NULL");
  if (option < 0 || option >= PQD_OPT_COUNT) return fail_arg("no such option");
  h->opt[option] = value < 0 ? -1 : value;
  return PQD_OK;
}

int pqd_decode_chunk_device(pqd_handle *h, const void *d_chunk, size_t len, const pqd_page_desc *pages, int32_t n_pages,
                            const pqd_column_desc *col) {
  PQD_GUARDED(
  if (n_pages < 0) return fail_arg("negative page count");
  const void *chunks[1] = {d_chunk};
  return decode_device(h, 1, chunks, &len, pages, &n_pages, col, false);
  )
}

int pqd_decode_chunks_device(pqd_handle *h, int32_t n_chunks, const void *const *d_chunks, const size_t *lens, const pqd_page_desc *pages,
                             const int32_t *pages_per_chunk, const pqd_column_desc *col, int32_t flags) {
  PQD_GUARDED(
  return decode_device(h, n_chunks, d_chunks, lens, pages, pages_per_chunk, col, (flags & PQD_LEVELS_ONLY) != 0);
  )
}

int pqd_decode_chunk_device_noalloc(int device, void *stream, const void *d_chunk, size_t len, const pqd_page_desc *pages,
                                    int32_t n_pages, const pqd_column_desc *col, void *d_values, size_t values_bytes,
                                    int32_t *d_def_levels, int32_t *d_rep_levels, size_t level_slots, pqd_chunk_result *res) {
  PQD_GUARDED(
  if (!col || !res || n_pages < 0) return fail_arg("NULL argument");
  if (device < 0 || device >= kMaxDevices) return fail_arg("no such device");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device >= n) return fail_arg("no such device");
  Guard g(device);
  NoAllocDevice &na = g_noalloc[device];
  std::lock_guard<std::mutex> lk(na.mu);
  int rc = noalloc_setup(na, device);
  if (rc) return rc;
  pqd_handle *h = &na.h;
  h->stream = static_cast<hipStream_t>(stream);
  h->values.p = d_values; h->values.cap = d_values ? values_bytes + 64 : 0;   // (the sizes asked for carry 64 bytes of slack that no kernel writes)
  h->def.p = d_def_levels; h->def.cap = d_def_levels ? 4 * level_slots + 64 : 0;
  h->rep.p = d_rep_levels; h->rep.cap = d_rep_levels ? 4 * level_slots + 64 : 0;
  const void *chunks[1] = {d_chunk};
  if ((rc = decode_device(h, 1, chunks, &len, pages, &n_pages, col, false))) {
    // a launch sequence that failed part way may have left kernels on the stream: they use the shared array, which the
    // next caller carves anew as soon as the lock is free
    (void)hipStreamSynchronize(h->stream);
    set_busy(h, false);
    return rc;
  }
  if ((rc = finish_call(h))) { (void)hipStreamSynchronize(h->stream); set_busy(h, false); return rc; }
  *res = h->result;
  return h->result.status;
  )
}

int pqd_input_buffer(pqd_handle *h, size_t len, uint8_t **buf) {
  PQD_GUARDED(
  if (!h || !buf) return fail_arg("NULL argument");
  if (h->no_alloc) return fail_arg("not a handle of this entry point");
  Guard g(h->device);
  int rc;
  const int s = h->stage_next;
  if (h->h2d_pending[s]) { HIP_TRY(hipEventSynchronize(h->ev_h2d[s])); h->h2d_pending[s] = false; }  // a transfer may still be reading the buffer
  if ((rc = h->stage_in[s].ensure(align_up(len + 64, 256)))) return rc;
  h->stage_next = s ^ 1;
  *buf = static_cast<uint8_t *>(h->stage_in[s].p);
  return PQD_OK;
  )
}

int pqd_decode_chunk(pqd_handle *h, const uint8_t *chunk, size_t len, const pqd_page_desc *pages, int32_t n_pages,
                     const pqd_column_desc *col) {
  PQD_GUARDED(
  if (n_pages < 0) return fail_arg("negative page count");
  const uint8_t *chunks[1] = {chunk};
  return decode_host(h, 1, chunks, &len, pages, &n_pages, col, false);
  )
}

int pqd_decode_chunks(pqd_handle *h, int32_t n_chunks, const uint8_t *const *chunks, const size_t *lens, const pqd_page_desc *pages,
                      const int32_t *pages_per_chunk, const pqd_column_desc *col, int32_t flags) {
  PQD_GUARDED(
  return decode_host(h, n_chunks, chunks, lens, pages, pages_per_chunk, col, (flags & PQD_LEVELS_ONLY) != 0);
  )
}

int pqd_decode_levels_device(pqd_handle *h, const void *d_chunk, size_t len, const pqd_page_desc *pages, int32_t n_pages,
                             const pqd_column_desc *col) {
  PQD_GUARDED(
  if (n_pages < 0) return fail_arg("negative page count");
  const void *chunks[1] = {d_chunk};
  return decode_device(h, 1, chunks, &len, pages, &n_pages, col, true);
  )
}

int pqd_decode_levels(pqd_handle *h, const uint8_t *chunk, size_t len, const pqd_page_desc *pages, int32_t n_pages,
                      const pqd_column_desc *col) {
  PQD_GUARDED(
  if (n_pages < 0) return fail_arg("negative page count");
  const uint8_t *chunks[1] = {chunk};
  return decode_host(h, 1, chunks, &len, pages, &n_pages, col, true);
  )
}

int pqd_wait(pqd_handle *h, pqd_chunk_result *res) {
  PQD_GUARDED(
  if (!h) return fail_arg("handle is NULL");
  Guard g(h->device);
  if (h->busy) {
    int rc = finish_call(h);
    if (rc) { set_busy(h, false); return rc; }
  }
  if (!h->have_result) return fail_arg("no decode call to wait for");
  if (res) *res = h->result;
  return h->result.status;
  )
}

int pqd_result_device(pqd_handle *h, pqd_device_result *out) {
  if (!h || !out) return fail_arg("NULL argument");
  if (!h->have_result) return fail_arg("no completed decode call");
  out->values = h->levels_only ? nullptr : h->values.p;
  out->def_levels = h->col.max_def > 0 ? static_cast<int32_t *>(h->def.p) : nullptr;
  out->rep_levels = h->col.max_rep > 0 ? static_cast<int32_t *>(h->rep.p) : nullptr;
  const bool is_bytes = h->col.physical_type == PQD_BYTE_ARRAY;
  if (is_bytes) out->values = nullptr;
  out->offsets = is_bytes && !h->levels_only ? static_cast<int64_t *>(h->offsets.p) : nullptr;
  out->bytes = is_bytes && !h->levels_only ? static_cast<uint8_t *>(h->bytes.p) : nullptr;
  return PQD_OK;
}

int pqd_copy_out(pqd_handle *h, void *values, int32_t *def_levels, int32_t *rep_levels, int64_t *offsets, uint8_t *bytes) {
  if (!h) return fail_arg("handle is NULL");
  if (!h->have_result) return fail_arg("no completed decode call");
  Guard g(h->device);
  const size_t n = (size_t)h->total_slots;
  const bool is_bytes = h->col.physical_type == PQD_BYTE_ARRAY;
  if (h->levels_only && (values || offsets || bytes)) return fail_arg("the last call decoded levels only");
  if (values && n && !is_bytes) HIP_TRY(hipMemcpyAsync(values, h->values.p, n * (size_t)h->elem_size, hipMemcpyDeviceToHost, h->stream));
  if (is_bytes && h->result.status == PQD_OK) {
    if (offsets) HIP_TRY(hipMemcpyAsync(offsets, h->offsets.p, 8 * (n + 1), hipMemcpyDeviceToHost, h->stream));
    if (bytes && h->bytes_len) HIP_TRY(hipMemcpyAsync(bytes, h->bytes.p, (size_t)h->bytes_len, hipMemcpyDeviceToHost, h->stream));
  }
  if (def_levels) {
    if (h->col.max_def > 0) { if (n) HIP_TRY(hipMemcpyAsync(def_levels, h->def.p, 4 * n, hipMemcpyDeviceToHost, h->stream)); }
    else std::memset(def_levels, 0, 4 * n);
  }
  if (rep_levels) {
    if (h->col.max_rep > 0) { if (n) HIP_TRY(hipMemcpyAsync(rep_levels, h->rep.p, 4 * n, hipMemcpyDeviceToHost, h->stream)); }
    else std::memset(rep_levels, 0, 4 * n);
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return PQD_OK;
}

int pqd_result_host(pqd_handle *h, pqd_device_result *out) {
  PQD_GUARDED(
  if (!h || !out) return fail_arg("NULL argument");
  if (!h->have_result) return fail_arg("no completed decode call");
  if (h->no_alloc) return fail_arg("not a handle of this entry point");
  Guard g(h->device);
  *out = pqd_device_result{};
  const size_t n = (size_t)h->total_slots;
  const bool is_bytes = h->col.physical_type == PQD_BYTE_ARRAY;
  const bool with_values = !h->levels_only && !(is_bytes && h->result.status != PQD_OK);
  const size_t val_bytes = with_values ? (is_bytes ? 8 * (n + 1) : n * (size_t)h->elem_size) : 0;
  const size_t byte_bytes = with_values && is_bytes ? (size_t)h->bytes_len : 0;
  const size_t lvl_bytes = 4 * n;
  // one pinned block: values or offsets | bytes | def | rep, each part 256-byte aligned
  const size_t o_bytes = align_up(val_bytes, 256), o_def = o_bytes + align_up(byte_bytes, 256);
  const size_t o_rep = o_def + (h->col.max_def > 0 ? align_up(lvl_bytes, 256) : 0);
  const size_t total = o_rep + (h->col.max_rep > 0 ? align_up(lvl_bytes, 256) : 0);
  int rc;
  if ((rc = h->stage_out.ensure(total + 256))) return rc;
  uint8_t *m = static_cast<uint8_t *>(h->stage_out.p);
  if (val_bytes) HIP_TRY(hipMemcpyAsync(m, is_bytes ? h->offsets.p : h->values.p, val_bytes, hipMemcpyDeviceToHost, h->stream));
  if (byte_bytes) HIP_TRY(hipMemcpyAsync(m + o_bytes, h->bytes.p, byte_bytes, hipMemcpyDeviceToHost, h->stream));
  if (h->col.max_def > 0 && n) HIP_TRY(hipMemcpyAsync(m + o_def, h->def.p, lvl_bytes, hipMemcpyDeviceToHost, h->stream));
  if (h->col.max_rep > 0 && n) HIP_TRY(hipMemcpyAsync(m + o_rep, h->rep.p, lvl_bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (with_values) {
    if (is_bytes) { out->offsets = reinterpret_cast<int64_t *>(m); out->bytes = m + o_bytes; }
    else out->values = m;
  }
  if (h->col.max_def > 0) out->def_levels = reinterpret_cast<int32_t *>(m + o_def);
  if (h->col.max_rep > 0) out->rep_levels = reinterpret_cast<int32_t *>(m + o_rep);
  return PQD_OK;
  )
}

int pqd_row_starts(pqd_handle *h, int64_t *host_out, int64_t cap, int64_t **device_out) {
  PQD_GUARDED(
  if (!h) return fail_arg("handle is NULL");
  if (!h->have_result) return fail_arg("no completed decode call");
  if (h->result.status != PQD_OK) return fail_arg("the last decode call failed");
  Guard g(h->device);
  const int64_t n_rows = h->result.num_rows, n_slots = h->total_slots;
  if (host_out && cap < n_rows + 1) return fail_arg("row_starts needs num_rows + 1 entries");
  int rc;
  if ((rc = h->row_starts.ensure(8 * (size_t)(n_rows + 2) + 64))) return rc;
  if ((rc = h->scratch.ensure(8 * ((size_t)n_slots / 4096 + 8) + 64))) return rc;
  int64_t *d = static_cast<int64_t *>(h->row_starts.p);
  HIP_TRY(launch_row_starts(h->col.max_rep > 0 ? static_cast<const int32_t *>(h->rep.p) : nullptr, n_slots, n_rows,
                            static_cast<int64_t *>(h->scratch.p), d, h->stream));
  if (host_out) HIP_TRY(hipMemcpyAsync(host_out, d, 8 * (size_t)(n_rows + 1), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (device_out) *device_out = d;
  return PQD_OK;
  )
}

int pqd_page_counts(pqd_handle *h, int64_t *out, int32_t cap_pages) {
  PQD_GUARDED(
  if (!h || !out) return fail_arg("NULL argument");
  if (!h->have_result) return fail_arg("no completed decode call");
  Guard g(h->device);
  const int n = std::min<int>(cap_pages, h->n_pages);
  if (n <= 0) return PQD_OK;
  std::vector<PageState> st((size_t)n);
  HIP_TRY(hipMemcpy(st.data(), h->ctx.state, sizeof(PageState) * (size_t)n, hipMemcpyDeviceToHost));
  for (int p = 0; p < n; p++) {
    const bool data = h->h_pages[(size_t)p].data_idx >= 0;
    out[3 * p + 0] = data ? st[(size_t)p].n_nulls : 0;
    out[3 * p + 1] = data ? st[(size_t)p].n_rows : 0;
    out[3 * p + 2] = data ? std::max(h->h_pages[(size_t)p].num_values, 0) : 0;
  }
  return PQD_OK;
  )
}

int pqd_direct_streams(pqd_handle *h, int32_t *n_out) {
  if (!h || !n_out) return fail_arg("NULL argument");
  if (!h->have_result) return fail_arg("no completed decode call");
  *n_out = (static_cast<const int32_t *>(h->host_small.p) + kSmallStatus)[kDirectCount];
  return PQD_OK;
}

int pqd_kernel_times(pqd_handle *h, pqd_kernel_time *out, int32_t cap, int32_t *n_out) {
  if (!h || !n_out) return fail_arg("NULL argument");
  Guard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  int n = 0;
  for (size_t i = 0; i < h->n_events; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->events[i].start, h->events[i].stop) != hipSuccess) continue;
    int j = 0;
    for (; j < n; j++) if (std::strcmp(out[j].name, h->events[i].name) == 0) break;
    if (j == n) {
      if (n >= cap) continue;
      std::memset(&out[n], 0, sizeof out[n]);
      std::strncpy(out[n].name, h->events[i].name, sizeof(out[n].name) - 1);
      n++;
    }
    out[j].ms += ms;
    out[j].launches += 1;
  }
  *n_out = n;
  return PQD_OK;
}

}  // extern "C"
