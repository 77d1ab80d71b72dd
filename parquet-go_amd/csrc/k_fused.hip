// One pass over the dictionary-index stream of a large page on gfx950: run headers walked and values written by the
// same workgroup, the page's bytes read from memory once.
//
// The run-header walk of an RLE / bit-packed hybrid stream is a chain (the place of a header is known only after the
// header before it), so a page is walked by one wavefront whatever else happens.  Run tables written by one kernel and
// expanded by another make every page byte cross the memory bus twice and put the two serial steps of a chunk one
// behind the other; here a workgroup owns a page and works on it in three stages: wavefront 0 follows the chain
// through a ring of stream bytes in LDS and notes, per step (a window of the ring, or kFRunCap runs of it), where the
// headers are; wavefront 1 turns the positions of the step before into a short list of runs; the other wavefronts
// turn the list of the step before that into values (unpack, dictionary lookup, stores) out of the same ring.  One
// barrier per step.  What bounds it: alone on the GPU a page takes as long as its walk (about 75 ns a run and 0.3 us a
// step); with two or three pages per CU in flight - several chunks decoded side by side, or a call of many pages -
// the memory system (about 4.6 TB/s of page bytes read and values written).
//
// Only clean streams are finished here: headers of one or two bytes, no empty runs, every run inside the stream, the
// stream ending with the run that completes the page's count.  Anything else leaves the stream to the general path
// (k_hybrid_prescan + k_hybrid_expand), which writes the same places again, so the reference's behaviour on odd input
// stays in one place.
//
// Reference behaviour reproduced: encoding/encodingread.go:127-268 (varint, RLE run, bit-packed run, hybrid),
// layout/page.go:644-655 (index stream of a dictionary-encoded page), :139-164 (dictionary gather).
#include <hip/hip_runtime.h>
#include "pqd_kernels.h"
#include "pqd_device.h"

namespace pqd {

constexpr int kFWin = 8192;                 // bytes per ring slot (one window of the walk)
constexpr int kFSlots = 4;
constexpr int kFRing = kFWin * kFSlots;
constexpr int kFMirror = 64;                // the ring's first bytes once more behind its end: reads run on across the wrap
constexpr int kFRunCap = 64;                // runs per step of the walk (a window of more takes several steps)
constexpr uint32_t kFBp = 0x80000000u;

// One run, or the part of a bit-packed run whose groups start inside the window.
struct __attribute__((aligned(16))) FusedEntry {
  uint32_t o;      // index of its first value in the page
  uint32_t y;      // RLE: the value; bit-packed: position of its first group (bytes from the 16-byte boundary before the stream)
  uint32_t item0;  // work items of the entries before it (an item: up to eight consecutive values)
  uint32_t nv;     // values | bit-packed flag << 31
};
struct __attribute__((aligned(16))) FusedList {
  uint32_t n, items, flags, pad;            // flags: 1 = the page ends with this list, 2 = left to the general path
  FusedEntry e[kFRunCap + 2];
};

// What the walk leaves per step: the positions of the headers it passed, where it stands, which window it was on.
struct __attribute__((aligned(16))) FusedPos {
  uint32_t n, q_end, R, last;
  uint32_t pos[kFRunCap];
};

constexpr int kFusedFixedLds = kFRing + kFMirror + 3 * (int)sizeof(FusedPos) + 2 * (int)sizeof(FusedList);

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

__device__ __forceinline__ uint32_t dpp_shr(uint32_t x, int n) {
  switch (n) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  }
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
  x += dpp_shr(x, 1); x += dpp_shr(x, 2); x += dpp_shr(x, 4); x += dpp_shr(x, 8);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, true);   // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, true);   // row_bcast:31
  return x;
}

template <class T, int DMODE>
__device__ __forceinline__ void fused_lookup8(const T *s_dict, const T *dict, uint32_t dict_lds_n, const uint32_t x[8], T y[8]) {
  typedef const __attribute__((address_space(3))) T *LdsPtr;
  typedef const __attribute__((address_space(1))) T *GlobalPtr;
  if (DMODE == 1) {
#pragma unroll
    for (int j = 0; j < 8; j++) y[j] = ((LdsPtr)s_dict)[x[j]];
    return;
  }
  // head of the dictionary in LDS, the rest in global memory: every lane reads LDS for all eight (entry 0 where the
  // entry is not there) and only the lanes beyond the head load from global memory, into registers of their own
  T yg[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    yg[j] = 0;
    if (x[j] >= dict_lds_n) yg[j] = ((GlobalPtr)dict)[x[j]];
  }
  T yl[8];
#pragma unroll
  for (int j = 0; j < 8; j++) yl[j] = ((LdsPtr)s_dict)[x[j] < dict_lds_n ? x[j] : 0u];
#pragma unroll
  for (int j = 0; j < 8; j++) y[j] = x[j] < dict_lds_n ? yl[j] : yg[j];
}

// Stores the first j1 values of an item (eight output elements starting at `outp`); a short item - the end of an RLE
// run or of the page - as one vector and up to three single elements.
__device__ __forceinline__ void fused_store(uint32_t *outp, const uint32_t y[8], uint32_t j1) {
  if (j1 == 8u) {
    U32x4Unaligned o0 = {{y[0], y[1], y[2], y[3]}}, o1 = {{y[4], y[5], y[6], y[7]}};
    reinterpret_cast<U32x4Unaligned *>(outp)[0] = o0;
    reinterpret_cast<U32x4Unaligned *>(outp)[1] = o1;
  } else if (j1 != 0u) {
    // (only an RLE item is ever short of eight, or the page's last item: all its values are y[0] in the first case)
    uint32_t k = 0;
    if (j1 >= 4u) { U32x4Unaligned o0 = {{y[0], y[1], y[2], y[3]}}; reinterpret_cast<U32x4Unaligned *>(outp)[0] = o0; k = 4; }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((uint32_t)j >= k && (uint32_t)j < j1) outp[j] = y[j];
  }
}
__device__ __forceinline__ void fused_store(uint64_t *outp, const uint64_t y[8], uint32_t j1) {
  if (j1 == 8u) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      U64x2Unaligned o = {{y[2 * q], y[2 * q + 1]}};
      reinterpret_cast<U64x2Unaligned *>(outp)[q] = o;
    }
  } else if (j1 != 0u) {
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((uint32_t)j < j1) outp[j] = y[j];
  }
}

// The items of one list: wavefront `ew` of `ne` takes 64 consecutive items per step.  The entry holding an item: every
// lane keeps the item count of one entry (two, for a long list), so the entries of the step's first and last item
// come from two ballots, and the lanes in between step through the few entries that separate them.
template <class T, int DMODE, int W>
__device__ __forceinline__ void fused_expand(const FusedList *L, const uint32_t *ring32, const T *s_dict, const T *dict,
                                             uint32_t dict_n, uint32_t dict_lds_n, T *out, uint32_t ucnt, int ew, int ne, int lane,
                                             bool &bad_index) {
  const uint32_t n = rfl(L->n), total = rfl(L->items);
  const bool check_bp = !(W < 32 && (1u << (W & 31)) <= dict_n);   // can a packed index be out of range
  if ((uint32_t)ew * 64u >= total) return;
  const uint32_t z0 = (uint32_t)lane < n ? L->e[lane].item0 : 0xFFFFFFFFu;
  const uint32_t z1 = (uint32_t)lane + 64u < n ? L->e[lane + 64].item0 : 0xFFFFFFFFu;
  const uint32_t z2 = (uint32_t)lane + 128u < n ? L->e[lane + 128].item0 : 0xFFFFFFFFu;   // (a list holds up to kFRunCap + 1 entries)
  static_assert(kFRunCap + 2 <= 192, "three entries per lane cover a list");
  auto z_of = [&](uint32_t d) -> uint32_t {
    return d < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)z0, (int)d)
                   : d < 128u ? (uint32_t)__builtin_amdgcn_readlane((int)z1, (int)d - 64) : (uint32_t)__builtin_amdgcn_readlane((int)z2, (int)d - 128);
  };
  auto last_le = [&](uint32_t it) -> uint32_t {     // the last entry whose items start at or before `it` (uniform)
    return (uint32_t)(__builtin_popcountll(__ballot(z0 <= it)) + __builtin_popcountll(__ballot(z1 <= it)) +
                      __builtin_popcountll(__ballot(z2 <= it))) - 1u;
  };
  for (uint32_t it0 = (uint32_t)ew * 64u; it0 < total; it0 += (uint32_t)ne * 64u) {
    const uint32_t it = it0 + (uint32_t)lane;
    const uint32_t it_hi = it0 + 63u < total ? it0 + 63u : total - 1u;
    const uint32_t lo_u = last_le(it0), hi_u = last_le(it_hi);
    uint32_t lo = lo_u;
    for (uint32_t d = lo_u + 1u; d <= hi_u; d++)
      if (z_of(d) <= it) lo = d;
    if (it < total) {
      const FusedEntry e = L->e[lo];
      const uint32_t k = it - e.item0, nv = e.nv & 0x7FFFFFFFu;
      const bool bp = (e.nv >> 31) != 0u;
      const uint32_t o = e.o + 8u * k;
      uint32_t j1 = nv - 8u * k < 8u ? nv - 8u * k : 8u;
      if (o >= ucnt) j1 = 0;                       // values beyond the page's count are read by the reference and dropped
      else if (ucnt - o < j1) j1 = ucnt - o;
      uint32_t x[8];
      unpack8<W>(ring32, bp ? (e.y + k * (uint32_t)W) & (uint32_t)(kFRing - 1) : 0u, x);
      if (check_bp) {
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (x[j] >= dict_n) { bad_index = bad_index || (bp && (uint32_t)j < j1); x[j] = 0; }   // dict.Values[index] out of range
      }
      uint32_t ry = e.y;
      if (!bp && ry >= dict_n) { bad_index = bad_index || j1 != 0u; ry = 0; }
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = bp ? x[j] : ry;
      T y[8];
      fused_lookup8<T, DMODE>(s_dict, dict, dict_lds_n, x, y);
      fused_store(out + o, y, j1);
    }
  }
}

// T: uint32_t / uint64_t dictionary entries.  DMODE 1: the whole dictionary in LDS; 2: its head in LDS, the rest gathered
// from global memory.  NW wavefronts: one walks, one makes the lists, NW - 2 expand.
template <class T, int DMODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_page_fused(ChunkCtx c, const T *__restrict__ dict_table, T *__restrict__ values_out,
                                                        int dict_lds_bytes, int min_len) {
  extern __shared__ uint4 smem[];
  uint8_t *lds = reinterpret_cast<uint8_t *>(smem);
  // LDS: the ring first (a position's place in it is the position masked), then the step's header positions, the two
  // lists and the dictionary
  uint32_t *ring32 = reinterpret_cast<uint32_t *>(lds);
  uint4 *ring4 = reinterpret_cast<uint4 *>(ring32);
  FusedPos *s_pos = reinterpret_cast<FusedPos *>(lds + kFRing + kFMirror);   // header positions of three steps
  FusedList *lists = reinterpret_cast<FusedList *>(s_pos + 3);
  T *s_dict = reinterpret_cast<T *>(lds + kFusedFixedLds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (int)rfl((uint32_t)tid >> 6);
  constexpr int NT = NW * 64;

  Stream *sp = &c.streams[2 * c.nd + (int)blockIdx.x];
  const int w = sp->w, cnt = sp->cnt, len = sp->len;
  if (sp->status != 0 || sp->direct || cnt <= 0 || w <= 0 || w > 32 || len != sp->avail || len < min_len) return;
  const PageIn &pg = c.pages[sp->page];
  if ((pg.encoding != PQD_ENC_PLAIN_DICTIONARY && pg.encoding != PQD_ENC_RLE_DICTIONARY) || pg.dict_page < 0 || pg.dict_n <= 0) return;
  const T *dict = dict_table + pg.dict_off;
  const uint32_t dict_n = (uint32_t)pg.dict_n;
  const uint32_t lds_room = (uint32_t)dict_lds_bytes / (uint32_t)sizeof(T);
  const uint32_t dict_lds_n = dict_n < lds_room ? dict_n : lds_room;
  if (DMODE == 1 && dict_lds_n < dict_n) return;

  const uint8_t *abs0 = c.in + sp->off;
  const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(abs0) & 15u);
  const uint8_t *base16 = abs0 - a;
  const uint8_t *in_end = c.in + c.in_len;
  const uint32_t qend = a + (uint32_t)len;     // positions count from the 16-byte boundary before the stream
  const uint32_t ucnt = (uint32_t)cnt, uw = (uint32_t)w, nb_rle = (uint32_t)(w + 7) / 8u;
  const uint32_t rle_mask = nb_rle >= 4 ? 0xFFFFFFFFu : ((1u << (8u * nb_rle)) - 1u);
  T *out = values_out + sp->out_base;

  // window j of the stream <-> ring slot j & 3
  auto load_vec = [&](uint32_t j, int i) -> uint4 {
    const uint32_t qpos = j * (uint32_t)kFWin + 16u * (uint32_t)i;
    const uint8_t *p = base16 + (size_t)j * kFWin + 16 * i;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    // only granules that hold bytes of the stream are read (in a call of several chunks the bytes behind a chunk's last
    // page may not be mapped); what the ring holds beyond the stream's end is never used for a value that is kept
    if (qpos < qend && p < in_end) v = *reinterpret_cast<const uint4 *>(p);
    return v;
  };
  auto put_vec = [&](uint32_t j, int i, uint4 v) {
    const uint32_t slot = j & (uint32_t)(kFSlots - 1);
    ring4[slot * (kFWin / 16) + i] = v;
    if (slot == 0 && i < kFMirror / 16) ring4[kFRing / 16 + i] = v;
  };

  for (int i = tid; i < kFWin / 16; i += NT) put_vec(0u, i, load_vec(0u, i));
  if ((reinterpret_cast<uintptr_t>(dict) & 15u) == 0) {
    const int nvec = (int)((dict_lds_n * sizeof(T) + 15) / 16);
    for (int i = tid; i < nvec; i += NT) reinterpret_cast<uint4 *>(s_dict)[i] = reinterpret_cast<const uint4 *>(dict)[i];
  } else {
    for (uint32_t i = tid; i < dict_lds_n; i += NT) s_dict[i] = dict[i];
  }
  __syncthreads();

  // ---- three stages, one barrier per step ----
  // Wavefront 0 walks: a step covers the headers of one window (kFRunCap of them when it holds more) and leaves
  // nothing but their positions.  Wavefront 1 turns the positions of the step before into the list of runs: what each
  // run is, where its values and work items start; the groups of a bit-packed run that start beyond the window are
  // carried into the next list.  The other wavefronts expand the list of the step before that.  While window R is
  // walked for the first time, window R + 1 comes in: the walk stays inside its window - a header in a window's last
  // byte (its second byte may belong to the window on its way) is left to the next step - the list maker and the
  // expanding wavefronts are at most one and two windows behind and read a few bytes ahead, which makes four slots.
  uint32_t vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));   // a zero the compiler keeps in a vector register (see the hop loop)
  typedef __attribute__((address_space(3))) uint32_t *Lds32;
  const uint32_t ring_addr = (uint32_t)(uintptr_t)(Lds32)ring32;

  // -- the walk (wavefront 0; everything wave-uniform) --
  uint32_t q = a, R = 0;
  bool first = true;
  auto hop_step = [&](FusedPos *P) {
    const uint32_t E = (R + 1u) * (uint32_t)kFWin;
    const uint32_t whi = E - 1u < qend ? E - 1u : qend;
    // loads that write LDS themselves (no registers held across the walk; a wavefront-instruction fills 64 x 16
    // consecutive bytes), waited for at the step's barrier
    constexpr int NV = kFWin / 16 / 64;
    typedef __attribute__((address_space(3))) uint32_t *LdsDst;
    typedef const __attribute__((address_space(1))) uint32_t *GlobalSrc;
    const uint32_t jl = R + 1u;
    if (first && jl * (uint32_t)kFWin < qend) {
      const uint8_t *p = base16 + (size_t)jl * kFWin + 16 * lane;
      const uint32_t qp = jl * (uint32_t)kFWin + 16u * (uint32_t)lane;   // position of this lane's first granule
      const uint32_t slot = jl & (uint32_t)(kFSlots - 1);
      uint8_t *dst = reinterpret_cast<uint8_t *>(ring32) + slot * (uint32_t)kFWin;
      const bool whole = (jl + 1u) * (uint32_t)kFWin <= qend;            // the window lies inside the stream
#pragma unroll
      for (int k = 0; k < NV; k++)
        if (whole || qp + 1024u * (uint32_t)k < qend)    // (only granules that hold bytes of the stream)
          __builtin_amdgcn_global_load_lds((GlobalSrc)(p + 1024 * k), (LdsDst)(dst + 1024 * k), 16, 0, 0);
      if (slot == 0 && lane < kFMirror / 16 && (whole || qp < qend))
        __builtin_amdgcn_global_load_lds((GlobalSrc)p, (LdsDst)(reinterpret_cast<uint8_t *>(ring32) + kFRing), 16, 0, 0);
    }
    uint32_t n = 0;
    if (q < whi) {
      // The chain itself - header at q, next header at q + its length + the run's bytes - is all that has to happen one
      // run after the other, and a lone wavefront issues an instruction every four cycles or so: the loop does nothing
      // but note the position and step on (vector instructions on a value all lanes share: no trips between the scalar
      // and the vector side; written out by hand, 21 instructions a run).  It does not look at what it steps over: a
      // header of three bytes or more, an empty run or a run reaching past the stream is found when the list is made,
      // and then nothing of the page is kept.
      uint32_t vq = q + vz;
      uint32_t vpp = (uint32_t)(uintptr_t)(Lds32)P->pos + vz;
      const uint32_t cap = (uint32_t)kFRunCap;
      uint32_t va, b0, b1, g, t;
      const uint32_t v_negnb = vz - nb_rle, v_onb = vz + 1u + nb_rle;
      asm volatile(
          "s_waitcnt lgkmcnt(0)\n"
          "1:\n"
          "v_and_b32 %[va], 0x7fff, %[vq]\n"
          "v_add_u32 %[va], %[rb], %[va]\n"
          "ds_read_u8 %[b0], %[va]\n"
          "ds_read_u8 %[b1], %[va] offset:1\n"
          "ds_write_b32 %[vpp], %[vq]\n"
          "v_add_u32 %[vpp], 4, %[vpp]\n"
          "s_add_u32 %[n], %[n], 1\n"
          "s_waitcnt lgkmcnt(1)\n"                          // LDS operations complete in order: both reads are in
          "v_bfe_u32 %[g], %[b0], 1, 6\n"                   // one-byte header: groups or values
          "v_and_b32 %[b1], 0x7f, %[b1]\n"
          "v_lshl_or_b32 %[b1], %[b1], 6, %[g]\n"           // two-byte header
          "v_cmp_lt_u32 vcc, 0x7f, %[b0]\n"
          "v_cndmask_b32 %[g], %[g], %[b1], vcc\n"
          "v_mad_i32_i24 %[t], %[g], %[w], %[negnb]\n"      // bit-packed: w bytes per group, against the RLE value's bytes
          "v_and_b32 %[b0], 1, %[b0]\n"
          "v_mad_i32_i24 %[vq], %[b0], %[t], %[vq]\n"
          "v_addc_co_u32 %[vq], vcc, %[onb], %[vq], vcc\n"  // + header byte + RLE value bytes + second header byte
          "v_cmp_gt_u32 vcc, %[whi], %[vq]\n"
          "s_cbranch_vccz 2f\n"
          "s_cmp_lt_u32 %[n], %[cap]\n"
          "s_cbranch_scc1 1b\n"
          "2:\n"
          : [vq] "+v"(vq), [vpp] "+v"(vpp), [n] "+s"(n), [va] "=&v"(va), [b0] "=&v"(b0), [b1] "=&v"(b1), [g] "=&v"(g), [t] "=&v"(t)
          : [rb] "s"(ring_addr), [w] "s"(uw), [negnb] "v"(v_negnb), [onb] "v"(v_onb), [whi] "s"(whi), [cap] "s"(cap)
          : "vcc", "scc", "memory");
      q = rfl(vq);
    }
    if (lane == 0) { P->n = n; P->q_end = q; P->R = R; P->last = q >= qend ? 1u : 0u; }
    first = q >= whi;
    if (first) R++;
  };

  // -- the lists (wavefront 1) --
  uint32_t o32 = 0;
  uint32_t cp = 0, cg = 0, co = 0;   // carried groups: position of the first, how many, index of their first value
  bool fin = false, bail = false;
  auto make_list = [&](const FusedPos *P, FusedList *L) {
    const uint32_t n = rfl(P->n), qs = rfl(P->q_end), Rs = rfl(P->R), plast = rfl(P->last);
    const uint32_t E = (Rs + 1u) * (uint32_t)kFWin;
    uint32_t n_list = 0, items = 0;
    if (cg) {
      const uint32_t fit = (E - cp + uw - 1u) / uw;        // its groups that start before E (cp < E)
      const uint32_t here = cg < fit ? cg : fit;
      if (lane == 0) L->e[0] = FusedEntry{co, cp, 0u, (here * 8u) | kFBp};
      n_list = 1; items = here;
      cp += here * uw; co += here * 8u; cg -= here;
    }
    if (n > 0 && !fin && !bail) {
      if (cg != 0 || qs > qend) bail = true;       // (headers inside a run that goes on; a run the stream does not hold whole)
      // What each noted run is, and where its values and items start: one run per lane, wave prefix sums.
      for (uint32_t b = 0; b < n && !bail; b += 64u) {
        const uint32_t nbt = n - b < 64u ? n - b : 64u;
        const bool lastb = b + 64u >= n;
        const bool in = (uint32_t)lane < nbt;
        const uint32_t posv = in ? P->pos[b + (uint32_t)lane] : a;
        const uint32_t ro = posv & (uint32_t)(kFRing - 1), dj = ro >> 2, sj = (ro & 3u) * 8u;
        const uint32_t e0 = ring32[dj], e1 = ring32[dj + 1], e2 = ring32[dj + 2];
        const uint32_t y0 = __builtin_amdgcn_alignbit(e1, e0, sj), y1 = __builtin_amdgcn_alignbit(e2, e1, sj);
        const uint32_t two = (y0 >> 7) & 1u;
        const uint32_t header = (y0 & 0x7Fu) | (two ? ((y0 >> 1) & 0x3F80u) : 0u);
        const uint32_t hlen = 1u + two, isbp = header & 1u, g = header >> 1;
        if (__ballot(in && ((y0 & 0x8080u) == 0x8080u || g == 0u))) { bail = true; break; }   // not for this kernel
        const uint32_t val = (uint32_t)((((uint64_t)y1 << 32) | y0) >> (8u * hlen)) & rle_mask;   // RLE value: the low 32 bits (encodingread.go:160)
        const uint32_t Pp = posv + hlen;
        const uint32_t nvi = in ? g << (isbp * 3u) : 0u;
        const uint32_t incl_v = wave_incl_sum(nvi);
        uint32_t iti = in ? (isbp ? g : (nvi + 7u) >> 3) : 0u;
        uint32_t nv_here = nvi;
        // only the last run walked can reach beyond the window: its groups that start at E or later are carried
        if (lastb && qs > E && __builtin_amdgcn_readlane((int)isbp, (int)nbt - 1)) {
          const uint32_t Pl = (uint32_t)__builtin_amdgcn_readlane((int)Pp, (int)nbt - 1);
          const uint32_t gl = (uint32_t)__builtin_amdgcn_readlane((int)g, (int)nbt - 1);
          const uint32_t sl = o32 + (uint32_t)__builtin_amdgcn_readlane((int)(incl_v - nvi), (int)nbt - 1);
          const uint32_t fit = Pl >= E ? 0u : (E - Pl + uw - 1u) / uw;
          const uint32_t here = gl < fit ? gl : fit;
          if (here < gl) {
            cg = gl - here; cp = Pl + here * uw; co = sl + here * 8u;
            if ((uint32_t)lane == nbt - 1u) { iti = here; nv_here = here * 8u; }
          }
        }
        const uint32_t incl_i = wave_incl_sum(iti);
        if (in) L->e[n_list + (uint32_t)lane] = FusedEntry{o32 + incl_v - nvi, isbp ? Pp : val, items + incl_i - iti, nv_here | (isbp << 31)};
        const uint64_t full = __ballot(in && o32 + incl_v >= ucnt);   // the run that completes the page's count
        if (full) {
          fin = true;
          // runs or bytes behind it: the reference reads them too, and may fail on them
          if ((uint32_t)__builtin_ctzll(full) != nbt - 1u || !lastb || qs != qend) bail = true;
        }
        o32 += (uint32_t)__builtin_amdgcn_readlane((int)incl_v, (int)nbt - 1);
        items += (uint32_t)__builtin_amdgcn_readlane((int)incl_i, (int)nbt - 1);
        n_list += nbt;
      }
    }
    if (!fin && !bail && cg == 0 && plast) bail = true;   // the stream holds too few values
    if (lane == 0) { L->n = n_list; L->items = items; L->flags = ((fin && cg == 0) ? 1u : 0u) | (bail ? 2u : 0u); }
  };

  // The barrier of a step: what the stages hand each other is in LDS, so a wavefront waits for its LDS operations only
  // - the values it stored (and the dictionary entries it asked memory for) stay in flight across the barrier.  The
  // walking wavefront also waits for the window it asked for (loads that write LDS count as memory operations).
  auto step_barrier = [&]() {
    if (wave == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // the walk is the one serial chain of the page: its wavefront goes first whenever it has an instruction ready
  if (wave == 0) { __builtin_amdgcn_s_setprio(3); hop_step(&s_pos[0]); }
  step_barrier();
  if (wave == 0) hop_step(&s_pos[1]);
  else if (wave == 1) make_list(&s_pos[0], &lists[0]);
  step_barrier();

  bool bad_index = false;
  bool left = false;
  for (uint32_t r = 0;; r++) {
    const FusedList *L = &lists[r & 1u];
    const uint32_t fl = rfl(L->flags);
    if (fl & 2u) { left = true; break; }
    if (wave == 0) {
      hop_step(&s_pos[(r + 2u) % 3u]);
    } else if (wave == 1) {
      if (!(fl & 1u)) make_list(&s_pos[(r + 1u) % 3u], &lists[(r + 1u) & 1u]);
    } else {
      switch (w) {
#define PQD_CASE(W) case W: fused_expand<T, DMODE, W>(L, ring32, s_dict, dict, dict_n, dict_lds_n, out, ucnt, wave - 2, NW - 2, lane, bad_index); break;
        PQD_CASE(1) PQD_CASE(2) PQD_CASE(3) PQD_CASE(4) PQD_CASE(5) PQD_CASE(6) PQD_CASE(7) PQD_CASE(8)
        PQD_CASE(9) PQD_CASE(10) PQD_CASE(11) PQD_CASE(12) PQD_CASE(13) PQD_CASE(14) PQD_CASE(15) PQD_CASE(16)
        PQD_CASE(17) PQD_CASE(18) PQD_CASE(19) PQD_CASE(20) PQD_CASE(21) PQD_CASE(22) PQD_CASE(23) PQD_CASE(24)
        PQD_CASE(25) PQD_CASE(26) PQD_CASE(27) PQD_CASE(28) PQD_CASE(29) PQD_CASE(30) PQD_CASE(31) PQD_CASE(32)
#undef PQD_CASE
        default: break;
      }
    }
    step_barrier();
    if (fl & 1u) break;
  }

  // ---- the verdict ----
  const int any_bad = __syncthreads_or(bad_index ? 1 : 0);
  if (left) return;    // (uniform) the general path decodes the stream and finds whatever is wrong with it
  if (tid == 0) {
    if (any_bad) atomicCAS(&c.state[sp->page].status, 0, PQD_ERR_MALFORMED);
    sp->direct = 1; sp->n_runs = 0; sp->decoded = (int64_t)cnt;
  }
}

constexpr int kLdsPerCu = 160 * 1024;

template <class T>
static hipError_t launch_fused_t(const ChunkCtx &c, const uint8_t *dict_table, uint8_t *values_out, int max_dict_n, int min_len,
                                 hipStream_t stream) {
  const size_t dict_bytes = ((size_t)max_dict_n * sizeof(T) + 15) / 16 * 16;
  const T *dt = reinterpret_cast<const T *>(dict_table);
  T *vo = reinterpret_cast<T *>(values_out);
  if (dict_bytes <= 64 * 1024) {
    // the whole dictionary beside the ring: two or three workgroups of eight wavefronts per CU
    const size_t shmem = dict_bytes + kFusedFixedLds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_page_fused<T, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_page_fused<T, 1, 8>), dim3(c.nd), dim3(512), shmem, stream, c, dt, vo, (int)dict_bytes, min_len);
  } else {
    // larger dictionaries: one workgroup of sixteen wavefronts per CU, all the LDS the ring leaves holds the head of the
    // dictionary (a random 4-byte gather from L2 is bound by its request rate: every entry that sits in LDS counts)
    size_t room = (size_t)(kLdsPerCu - kFusedFixedLds - 512) / 16 * 16;
    if (room > dict_bytes) room = dict_bytes;
    const size_t shmem = room + kFusedFixedLds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_page_fused<T, 2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_page_fused<T, 2, 16>), dim3(c.nd), dim3(1024), shmem, stream, c, dt, vo, (int)room, min_len);
  }
  return hipGetLastError();
}

hipError_t launch_page_fused(const ChunkCtx &c, const uint8_t *dict_table, uint8_t *values_out, int max_dict_n, int min_len,
                             hipStream_t stream) {
  if (c.nd <= 0 || max_dict_n <= 0) return hipSuccess;
  if (c.elem_size == 4) return launch_fused_t<uint32_t>(c, dict_table, values_out, max_dict_n, min_len, stream);
  if (c.elem_size == 8) return launch_fused_t<uint64_t>(c, dict_table, values_out, max_dict_n, min_len, stream);
  return hipSuccess;
}

}  // namespace pqd
