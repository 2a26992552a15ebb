// sw_long_kernel.h — score-only sweep of ONE LONG query (2049+ rows): the strips of a tile pipelined over the wavefronts of
// a workgroup, so that a tile can be as long as the reference allows and the warm-up columns in front of it are paid once
// per WORKGROUP instead of once per wavefront.
//
// Same job as sw_score_kernel's strip-mined instances (Similarity_Matrix::iterate + the value half of
// find_index_of_maximum for one alignment, reference src/aligner/similaritymatrix.cpp:99-264, :21-28; per piece:
// src/aligner/plocalaligner.cpp:110-129), other decomposition:
//   * sw_score_kernel<.., STRIPS>: a tile belongs to ONE wavefront, which sweeps its columns once per 2048-row strip.
//     Filling 1024 SIMDs then takes ~2 k tiles, and each pays the warm-up margin m + ceil(smax m / g) (25 k columns for a
//     10 kbp query): 19 % redundant work on a 250 Mbp reference, 62 % on one rank's eighth of it.
//   * here: strip s of a tile = rows [s*64*R, (s+1)*64*R) belongs to wavefront s of the workgroup; all strips sweep the tile's
//     columns at the same time, each one or two 64-column segments behind the strip above it, whose bottom row it takes
//     from an LDS ring through the `old` operand of its wave_shr:1 DPP move (the mechanism of sw_strip_kernel.h).  One
//     workgroup per CU fills the chip with 256 tiles: 976 k own columns per tile on 250 Mbp (2.5 % warm-up), 122 k on an
//     eighth (20 %).
// The whole query profile has to be in LDS at once for that: float16 entries (score / 2048, exact for integer scores up to
// 2048), prof[code][strip][lane][row], read with ds_read_b128 and fed to the float32 cell through v_fma_mix_f32
// (x = clamp(score * scale + NW), scale = 2048 * 2^-k: the same value the float32 profile of sw_score_kernel holds).
// Cells, scaling and results are those of Cell<kSemF32> (sw_score_kernel.h): keys (max << 32 | ~sub-chunk) by atomicMax,
// per-sub-chunk values for the sampled maximum (MK = 4).
//
// Float32 profile, several workgroups per tile (P32 = true): the float16 profile costs a v_fma_mix_f32 (4 cycles per wave64
// instruction) where the float32 one takes a v_add_f32 clamp (2: it issues at the double rate), 12 % of the cell.  A float32
// profile of the whole query does not fit one CU (10 kbp x 5 codes x 4 B = 205 KB), so the strips of a tile are dealt to
// `groups` workgroups — on as many CUs — of `spg` strips each (10 kbp: 2 x 4 strips of 1280 rows, 102 KB per CU); the bottom
// row of a workgroup's last strip reaches the next workgroup's first strip through a global row (one float per column, written
// once, read once: 8 B per column and tile against 10^4 cell updates) with an agent-scope progress counter per (tile, group).
// The chain of waits is acyclic (group g only waits for group g - 1) and every workgroup of the launch is resident (the host
// never launches more workgroups than the chip holds at once), so the waits terminate; they are bounded all the same.
//
// Flow control between the wavefronts of a pipeline: sw_strip_kernel.h's (produced / consumed counters per wavefront,
// release stores once per segment, acquire polls with s_sleep, every wait bounded; on expiry the workgroup raises
// *status and drains, the host reports an error).  `pipes` pipelines (tiles) may share one workgroup and its profile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sw_score_kernel.h"

namespace mi355sw {

constexpr int kLongRing = 256;           // boundary positions held per strip (4 segments)
constexpr int kLongMaxWaves = 16;
constexpr int kLongSpinLimit = 1 << 22;

struct LongArgs {
  const uint8_t *refcodes;     // reference as dense codes
  const int64_t *range_lo;     // [nranges] sub-problems (pieces), [lo, hi)
  const int64_t *range_hi;
  int64_t chunk_len;           // own columns per tile (multiple of sub_len)
  int64_t sub_len;             // columns per reported sub-chunk maximum (multiple of 64)
  int64_t warm;                // warm-up columns in front of a tile (multiple of 64)
  const uint8_t *qbytes;       // the query's bytes
  int32_t qlen;
  int32_t qid;                 // its id (column of keys)
  int32_t nq;                  // row length of keys
  const uint16_t *htab;        // [256][ncodes] float16 bits of score / 2048; column ncodes-1 = padding
  const float *ftab;           // P32: [256][ncodes] score * 2^-k
  int32_t ncodes;
  float gap_s;                 // gap * 2^-k
  float scale;                 // 2048 * 2^-k
  uint32_t pubmax;             // != 0: published maxima are clamped to this float bit pattern (uint8 engine swept unsaturated)
  unsigned long long *keys;    // [nranges][nq]
  uint32_t *submax_out;        // MK > 1: value of every sub-chunk (float bits), range r at submax_out + r * submax_range_stride; else null
  int64_t submax_range_stride;
  int32_t nstrips;             // strips of the query
  int32_t groups;              // workgroups a tile's strips are dealt to (1: the whole pipeline in one workgroup)
  int32_t spg;                 // strips per workgroup = wavefronts per pipeline and workgroup (ceil(nstrips / groups))
  int32_t pipes;               // pipelines (tiles) per workgroup
  int64_t tiles_stride;        // tiles per range, rounded up to whole workgroups (indexes gbound / gcount)
  float *gbound;               // groups > 1: [(range * tiles_stride + tile) * (groups - 1) + g][gstride] bottom rows of group g
  int64_t gstride;
  long long *gcount;           // ... and the positions of it that are complete (zero at launch)
  int32_t subs_per_tile;       // chunk_len / sub_len
  int32_t *status;             // set to 1 when a pipeline wait expired
  // Saved state for the finish (host_saved.h): the sweep passes every cell of the matrix once; what a later exact window
  // would have to recompute behind a zero border and a warm-up margin of its own is kept instead —
  //   colsave  the H column 64 positions in front of every sub-chunk (stream position warm + b * sub_len - 64: the column
  //            just left of the window locate re-runs for sub-chunk b), every row of the query: lane l is at that column
  //            at step l of the segment, and stores its R values then;
  //   rowsave  the bottom row of every strip but the last (what already travels to the strip below), every position.
  // A block of 64 * R rows x one sub-chunk of columns is then an independent exact problem with known left column and top
  // row.  Values are the cells' own (H * 2^-k).  Null: nothing is saved.
  float *colsave;              // [(range * col_subs + tile * subs_per_tile + b) * col_rows + row]
  int64_t col_subs, col_rows;  // sub-chunks per range; nstrips * 64 * R
  float *rowsave;              // [((range * tiles_stride + tile) * (nstrips - 1) + strip) * row_stride + position]
  int64_t row_stride;
  int32_t fault;               // test hook (option fault_inject = long_stall): the waits BETWEEN workgroups expire at once
};

// Dwords between the float16 profile rows of adjacent lanes.  R = 24 (twelve dwords): 16-byte reads, stride
// ≡ 4 (mod 8) dwords as in sw_score_kernel.  Otherwise (R = 20: ten dwords) 8-byte reads at the plain stride R / 2: the 32
// two-dword windows of a ds_read_b64 half-wavefront start at 10 k mod 64 — 32 different even banks — and the stride between
// codes is a multiple of 64 dwords, so the map does not depend on the code a lane looks up.  No padding: 20 % less LDS.
// R = 32 (sixteen dwords): 8-byte reads at stride 18 — 18 k mod 64 are again 32 different even banks.
__host__ __device__ constexpr bool long_wide(int R) { return R == 24; }
__host__ __device__ constexpr int long_lane_stride(int R) { return long_wide(R) ? lane_stride(R / 2) : ((R / 2) % 8 == 0 ? R / 2 + 2 : R / 2); }

// bytes of dynamic LDS of a workgroup that holds `spg` strips of `pipes` tiles (p32: float32 profile entries)
__host__ __device__ inline size_t long_lds_bytes(int ncodes, int spg, int pipes, int R, int subs_per_tile, bool p32 = false) {
  const size_t prof = (size_t)ncodes * spg * 64 * (p32 ? lane_stride(R) : long_lane_stride(R)) * 4;
  const size_t waves = (size_t)spg * pipes;
  return prof + waves * kLongRing * 4 + (size_t)pipes * subs_per_tile * 4;
}

// wavefronts a workgroup of the R-rows-per-lane instance may have: the register file holds four per SIMD at <= 128 VGPRs
// (R = 20 / 24), three at <= 168 (R = 32: 2048-row strips, a third less per-step overhead per cell)
__host__ __device__ constexpr int long_max_waves(int R) { return R >= 32 ? 12 : kLongMaxWaves; }

template <int R, int MK, bool P32 = false>
__global__ __launch_bounds__(64 * long_max_waves(R)) void sw_long_kernel(const LongArgs a) {
  static_assert(R % 4 == 0, "two float16 profile entries per dword, whole 8-byte reads");
  static_assert(MK == 1 || MK == 4 || MK == 8, "running maximum every step, every 4th or every 8th");
  constexpr int LSH = P32 ? lane_stride(R) : long_lane_stride(R);   // dwords between the profile rows of adjacent lanes
  constexpr bool WIDE = P32 || long_wide(R);   // ds_read_b128 (else ds_read_b64)
  constexpr int UNR = MK > 4 ? MK : 4;         // steps per unrolled group: the fold steps are then compile-time
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ long long produced[kLongMaxWaves], consumed[kLongMaxWaves + 1];
  __shared__ int dead;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;   // (w in an SGPR: everything derived from it branches uniformly)
  const int nwaves = a.spg * a.pipes;
  const int grp = (int)(blockIdx.x % (unsigned)a.groups);                  // which part of the tile's strips this workgroup holds
  const int64_t tgroup = blockIdx.x / (unsigned)a.groups;
  uint32_t *prof = smem;                                                   // [ncodes][spg][64][LSH]
  float *ring_all = reinterpret_cast<float *>(prof + (size_t)a.ncodes * a.spg * 64 * LSH);
  uint32_t *submax_all = reinterpret_cast<uint32_t *>(ring_all + (size_t)nwaves * kLongRing);

  // ---- the profile of this workgroup's strips (rows grp*spg*64*R ...) ----------------------------
  {
    const int per_code = a.spg * 64 * R;
    const int row0 = grp * a.spg * 64 * R;
    uint16_t *p16 = reinterpret_cast<uint16_t *>(prof);
    for (int e = tid; e < a.ncodes * per_code; e += blockDim.x) {
      const int c = e / per_code;
      const int rem = e - c * per_code;
      const int sl = rem / R, r = rem - sl * R;                            // sl = local strip * 64 + lane
      const int i = row0 + sl * R + r;
      if (P32) {
        prof[(size_t)(c * a.spg * 64 + sl) * LSH + r] = __float_as_uint((i < a.qlen) ? a.ftab[(int)a.qbytes[i] * a.ncodes + c] : kPadScoreF);
      } else {
        const uint16_t v = (i < a.qlen) ? a.htab[(int)a.qbytes[i] * a.ncodes + c] : (uint16_t)0xC800;   // padding rows: -8 = -16384 / 2048
        p16[((size_t)(c * a.spg * 64 + sl) * LSH) * 2 + r] = v;
      }
    }
  }
  if (tid < kLongMaxWaves) produced[tid] = 0;
  if (tid <= kLongMaxWaves) consumed[tid] = 0;
  if (tid == 0) dead = 0;
  for (int e = tid; e < a.pipes * a.subs_per_tile; e += blockDim.x) submax_all[e] = 0u;
  __syncthreads();

  const int pipe = w / a.spg, ls = w - pipe * a.spg;                       // pipeline, strip within this workgroup
  const int strip = grp * a.spg + ls;                                      // strip of the query
  const int range = blockIdx.y;
  const int64_t rlo = a.range_lo[range], rhi = a.range_hi[range];
  const int64_t ntiles = (rhi - rlo + a.chunk_len - 1) / a.chunk_len;
  const int64_t tile = tgroup * a.pipes + pipe;
  const bool active = w < nwaves && tile < ntiles && strip < a.nstrips;
  const int64_t own_lo = rlo + tile * a.chunk_len;
  const int64_t own_hi = (own_lo + a.chunk_len < rhi) ? own_lo + a.chunk_len : rhi;
  const int64_t s0 = own_lo - a.warm;                                      // reference index of stream position 0
  const int64_t nb = a.warm + a.chunk_len;                                 // stream positions that hold columns
  const int nseg = (int)((nb + 64 + 63) / 64);                             // lane 63 reaches position nb - 1
  const uint32_t pad = (uint32_t)(a.ncodes - 1);
  uint32_t *submax = submax_all + (size_t)pipe * a.subs_per_tile;
  bool ok = true;

  if (active) {
    auto wait_for = [&](long long *counter, long long need) {
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        if (__hip_atomic_load(&dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || ++spins > kLongSpinLimit) {
          __hip_atomic_store(&dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ok = false;
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    };
    auto wait_global = [&](long long *counter, long long need) {          // progress of the workgroup above (device scope)
      int spins = a.fault ? kLongSpinLimit : 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
        if (__hip_atomic_load(&dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || ++spins > kLongSpinLimit) {
          __hip_atomic_store(&dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ok = false;
          return;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    };
    const bool has_in = strip > 0, has_out = strip + 1 < a.nstrips;
    const bool in_global = has_in && ls == 0, out_global = has_out && ls == a.spg - 1;
    const float *rin = ring_all + (size_t)((has_in && !in_global) ? w - 1 : w) * kLongRing;
    float *rout = ring_all + (size_t)w * kLongRing;
    const size_t gslot = ((size_t)range * (size_t)a.tiles_stride + (size_t)tile) * (size_t)(a.groups > 1 ? a.groups - 1 : 1);
    const float *gin = in_global ? a.gbound + (gslot + (size_t)(grp - 1)) * (size_t)a.gstride : nullptr;
    float *gout = out_global ? a.gbound + (gslot + (size_t)grp) * (size_t)a.gstride : nullptr;
    long long *gin_count = in_global ? a.gcount + gslot + (size_t)(grp - 1) : nullptr;
    long long *gout_count = out_global ? a.gcount + gslot + (size_t)grp : nullptr;
    auto stage_load = [&](int seg) -> uint32_t {                           // code of stream position seg*64 + l
      const int64_t col = s0 + (int64_t)seg * 64 + l;
      return (col >= rlo && col < own_hi) ? (uint32_t)a.refcodes[col] : pad;
    };
    const char *prof_lane = reinterpret_cast<const char *>(prof + (size_t)(ls * 64 + l) * LSH);
    const uint32_t code_stride = (uint32_t)a.spg * 64 * LSH * 4;           // bytes per reference code (a multiple of 64 dwords:
                                                                           // the lane -> bank map is the same for every code)
    float gv = a.gap_s, sv = a.scale;
    asm volatile("" : "+v"(gv), "+v"(sv));                                 // VGPR operands: an SGPR operand halves the issue rate of v_sub_f32
    // one-lane shifts across the wavefront.  shr1: lanes take the lane above, lane 0 keeps `old` (pass `old` as a value that
    // is dead afterwards and the move is done in place, no copy); rot1: lanes take the lane below (lane 63: don't care);
    // shl1_insert: lanes take the lane below, lane 63 takes `ins`.
    auto shr1 = [](uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); };
    auto rot1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true); };
    auto shl1_insert = [](uint32_t ins, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)ins, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); };

    float H[R], Hg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { H[r] = 0.0f; Hg[r] = -gv; }
    uint32_t up_prev = 0u;
    float mx = 0.0f;
    // The inner loop touches LDS for the profile only.  What a step needs from outside the lane travels through registers:
    //   code    this lane's reference code; every step it moves one lane up (the lane above was at this column one step ago)
    //           and lane 0 takes the next code of the segment from lane 0 of cseg, which rotates one lane down per step;
    //   bseg    the segment's 64 boundary values from the strip above (one ring read per lane and SEGMENT), rotating the same
    //           way: lane 0 of it is the `old` operand of the H shift — where the single-strip kernels get the zero border;
    //   oseg    collects lane 63's bottom-row value of every step (shift down, insert at lane 63): one ring write per segment.
    uint32_t code = pad;
    uint32_t curc = stage_load(0), nextc = stage_load(1);
    uint32_t oprev = 0u;
    float *rsave = (has_out && a.rowsave != nullptr)
                       ? a.rowsave + (((size_t)range * (size_t)a.tiles_stride + (size_t)tile) * (size_t)(a.nstrips - 1) + (size_t)strip) * (size_t)a.row_stride
                       : nullptr;
    const int segs_per_sub = (int)(a.sub_len / 64);
    const int warm_segs = (int)(a.warm / 64);
    int sub = 0;
    auto fold_sub = [&](int s) {
      float m = mx;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      if (l == 0 && m > 0.0f) atomicMax(&submax[s], __float_as_uint(m));  // non-negative floats order like their bits
      mx = 0.0f;
    };

    for (int seg = 0; seg < nseg && ok; ++seg) {
      // input: boundary positions seg*64 .. seg*64+63 (the strip above finishes them during ITS segment seg + 1)
      const long long need_in = ((int64_t)(seg + 1) * 64 < nb) ? (long long)(seg + 1) * 64 : (long long)nb;
      if (in_global) wait_global(gin_count, need_in);
      else if (has_in) wait_for(&produced[w - 1], need_in);
      // ring space: this segment stores positions <= seg*64 over the slots of positions <= seg*64 - kLongRing
      // (the global row holds every position: nothing to wait for)
      if (has_out && !out_global) wait_for(&consumed[w + 1], (long long)seg * 64 - kLongRing + 64);
      if (!ok) break;
      // positions >= nb (the segment after the last column: nb is a multiple of 64) are only reached by lagging lanes; the
      // strip above has not produced them: zero boundary there (padding columns, lower values only)
      uint32_t bseg = 0u;
      if (has_in && (int64_t)seg * 64 < nb)
        bseg = in_global ? __float_as_uint(__hip_atomic_load(gin + (size_t)seg * 64 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                         : __float_as_uint(rin[(seg * 64 + l) & (kLongRing - 1)]);
      uint32_t cseg = curc;
      curc = nextc;
      nextc = stage_load(seg + 2);
      uint32_t oseg = 0u;
      // The saved copy of the PREVIOUS segment's bottom row goes out here, at the start of the steps (the waits above and the
      // release operations at the segment's end wait for every memory operation of the wavefront still in flight), and as a
      // WRITE-THROUGH store (system scope: sc0 sc1): the chip has one L2 per XCD, so every agent-scope release between the
      // workgroups of a tile writes the L2's dirty lines back (buffer_wbl2) — with 12 GB of saved state passing through as
      // ordinary stores those write-backs cost the sweep a third (202 -> 268 ms on 10 kbp x 250 Mbp).
      if (rsave != nullptr && seg > 0) {
        const int64_t t = (int64_t)(seg - 1) * 64 + l - 63;
        if (t >= 0) __hip_atomic_store(rsave + t, __uint_as_float(oprev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // the segment in front of own sub-chunk bdone / segs_per_sub: lane l stands at the column to save at step l
      const int bdone = seg + 1 - warm_segs;
      const bool save_seg = a.colsave != nullptr && bdone >= 0 && bdone % segs_per_sub == 0 && bdone / segs_per_sub < a.subs_per_tile;
      const size_t csave_at = save_seg ? ((size_t)range * (size_t)a.col_subs + (size_t)tile * a.subs_per_tile + (size_t)(bdone / segs_per_sub)) * (size_t)a.col_rows +
                                             (size_t)strip * 64 * R
                                       : 0;
#pragma unroll UNR
      for (int k = 0; k < 64; ++k) {
        {
          const uint32_t head = cseg;                                      // lane 0: this step's code
          cseg = rot1(cseg);
          code = shr1(head, code);
        }
        uint32_t p[P32 ? R : R / 2];
        if constexpr (P32) {
          const u32x4 *pp = static_cast<const u32x4 *>(__builtin_assume_aligned(prof_lane + __umul24(code, code_stride), 16));
#pragma unroll
          for (int q = 0; q < R / 4; ++q) {
            const u32x4 v = pp[q];
            p[4 * q + 0] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
          }
        } else if constexpr (WIDE) {
          const u32x4 *pp = static_cast<const u32x4 *>(__builtin_assume_aligned(prof_lane + __umul24(code, code_stride), 16));
#pragma unroll
          for (int q = 0; q < R / 8; ++q) {
            const u32x4 v = pp[q];
            p[4 * q + 0] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
          }
        } else {
          const u32x2 *pp = static_cast<const u32x2 *>(__builtin_assume_aligned(prof_lane + __umul24(code, code_stride), 8));
#pragma unroll
          for (int q = 0; q < R / 4; ++q) {
            const u32x2 v = pp[q];
            p[2 * q + 0] = v.x; p[2 * q + 1] = v.y;
          }
        }
        uint32_t up;                                                       // H(row above the lane's first, this column)
        if (has_in) { const uint32_t head = bseg; bseg = rot1(bseg); up = shr1(head, __float_as_uint(H[R - 1])); }
        else up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
        float diag = __uint_as_float(up_prev);
        up_prev = up;
        float ng;
        asm("v_sub_f32 %0, %1, %2" : "=v"(ng) : "v"(__uint_as_float(up)), "v"(gv));
        float tpend = 0.0f;
        (void)tpend;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float wv = H[r];
          float x, h;
          // x = clamp(score * scale + NW): the [0, 1] clamp is the zero floor (cells hold H * 2^-k in [0, 1))
          if constexpr (P32) asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(x) : "v"(diag), "v"(p[r]));
          else if (r & 1) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(x) : "v"(p[r >> 1]), "v"(sv), "v"(diag));
          else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] clamp" : "=v"(x) : "v"(p[r >> 1]), "v"(sv), "v"(diag));
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(h) : "v"(x), "v"(Hg[r]), "v"(ng));
          if (MK == 1 || (k & (MK - 1)) == MK - 1) {                       // (compile-time per unrolled step)
            if (r & 1) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(tpend), "v"(h));
            else tpend = h;                                                // (R is even: every row is in some max3)
          }
          diag = wv;
          H[r] = h;
          asm("v_sub_f32 %0, %1, %2" : "=v"(ng) : "v"(h), "v"(gv));
          Hg[r] = ng;
        }
        if (has_out) oseg = shl1_insert(__float_as_uint(H[R - 1]), oseg);  // lane 63 inserts, the others pass down
        if (save_seg) {                                                    // (uniform: a scalar branch in all other segments —
          asm volatile("" ::: "memory");                                   //  kept apart from the per-lane test below)
          if (k == l) {                                                    // this lane stands at stream position seg * 64
            float *csave = a.colsave + csave_at + (size_t)l * R;             // (write-through stores, see the row save above:
#pragma unroll                                                             //  16 bytes each — as single dwords they cost 2.6 % of the sweep)
            for (int r = 0; r < R; r += 4) {
              typedef float f32x4 __attribute__((ext_vector_type(4)));
              const f32x4 v = {H[r], H[r + 1], H[r + 2], H[r + 3]};
              asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(csave + r), "v"(v) : "memory");
            }
          }
        }
      }
      // lane j of oseg holds lane 63's value of step j: stream position seg*64 + j - 63
      // positions <= seg*64 are stored; this wavefront has read positions <= seg*64 + 63
      if (out_global) {
        const int64_t t = (int64_t)seg * 64 + l - 63;
        if (t >= 0) __hip_atomic_store(gout + t, __uint_as_float(oseg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // (all lanes' stores before lane 0's counter)
        if (l == 0) __hip_atomic_store(gout_count, (long long)seg * 64 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else if (has_out) {
        const int t = seg * 64 + l - 63;
        if (t >= 0) rout[t & (kLongRing - 1)] = __uint_as_float(oseg);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // (all lanes' ring stores before lane 0's flag)
        if (l == 0) __hip_atomic_store(&produced[w], (long long)seg * 64 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (l == 0) __hip_atomic_store(&consumed[w], (long long)(seg + 1) * 64, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      oprev = oseg;                                                        // (saved at the start of the next segment's steps, see there)
      // lane 0 has just finished a sub-chunk (and it is not the tile's last): report and restart the maximum.  Lanes lag
      // lane 0 by up to 63 columns, so the tail of a sub-chunk is reported with the next one (the host widens its windows).
      const int done = seg + 1 - warm_segs;
      if (done > 0 && done % segs_per_sub == 0 && done / segs_per_sub < a.subs_per_tile) fold_sub(sub++);
    }
    if (ok && rsave != nullptr && nseg > 0) {                              // the last segment's bottom row
      const int64_t t = (int64_t)(nseg - 1) * 64 + l - 63;
      if (t >= 0) __hip_atomic_store(rsave + t, __uint_as_float(oprev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (ok) fold_sub(a.subs_per_tile - 1);                                 // the tile's last (or only) sub-chunk
    // whatever happened, the workgroup below must not wait for this one any more
    if (out_global && l == 0) __hip_atomic_store(gout_count, 0x7FFFFFFFFFFFFFFFll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (l == 0 && w < kLongMaxWaves) {
    // whatever happened, nobody may wait on this wavefront any more
    __hip_atomic_store(&produced[w], 0x7FFFFFFFFFFFFFFFll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&consumed[w], 0x7FFFFFFFFFFFFFFFll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!ok) *a.status = 1;
  }
  __syncthreads();                                                         // every wavefront gets here: all waits are bounded

  // ---- publish: wavefront 0 of each pipeline ------------------------------------------------------
  if (active && ls == 0 && __hip_atomic_load(&dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
    uint32_t bv = 0u, bs = 0xFFFFFFFFu;                                    // best value, first sub-chunk that holds it
    for (int s = l; s < a.subs_per_tile; s += 64) {
      uint32_t v = submax[s];
      // (several workgroups may hold strips of this tile: the values merge by maximum; the host zeroes the rows)
      if (MK > 1 && a.submax_out != nullptr && v != 0u)
        atomicMax(&a.submax_out[(size_t)range * (size_t)a.submax_range_stride + (size_t)tile * a.subs_per_tile + s], v);
      if (a.pubmax != 0u && v > a.pubmax) v = a.pubmax;
      if (v > bv) { bv = v; bs = (uint32_t)s; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const uint32_t ov = (uint32_t)__shfl_xor((int)bv, off), os = (uint32_t)__shfl_xor((int)bs, off);
      if (ov > bv || (ov == bv && os < bs)) { bv = ov; bs = os; }
    }
    if (l == 0 && bv != 0u) {
      const unsigned long long tag = 0xFFFFFFFFull - (unsigned long long)(tile * a.subs_per_tile + bs);
      const unsigned long long v = ((unsigned long long)bv << 32) | tag;
      unsigned long long *addr = a.keys + (size_t)range * a.nq + a.qid;
      if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(addr, v);
    }
  }
}

}  // namespace mi355sw
