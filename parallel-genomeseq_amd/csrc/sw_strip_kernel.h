// sw_strip_kernel.h — exact sweep of ONE LONG query over a window: the wavefronts of a workgroup form a pipeline.
//
// The register wavefront of sw_wave_kernel.h (ORIENT 0: lanes hold rows of x, the stream runs over a window of
// y) with whole-wavefront strips: strip s owns rows [s*64*R, (s+1)*64*R) of x — lane l its R consecutive rows —
// and sweeps the window one column per step.  Wavefront w of the workgroup runs strips w, w + nw, w + 2*nw, ...
// (one per round).  The bottom row of a strip enters the next strip through an LDS ring (lane 63 stores one
// value per step, lane 0 of the next wavefront picks it up through the DPP `old` operand, where the single-strip
// kernels get the zero border row), so the strips of a round run concurrently, each two 64-column segments
// behind the one above it; the last wavefront of a round hands its bottom row to wavefront 0 of the next round
// through a global scratch row.  This is the multi-wavefront cooperative sweep of one alignment (what the
// reference's fine-grained OpenMP variants attempt, similaritymatrix.cpp:118-245), kept inside one workgroup so
// that all participants are resident by construction.
//
// Flow control: per wavefront a count of produced and of consumed boundary positions (monotonic over rounds),
// published with release stores once per segment and polled with acquire loads; a strip waits for input
// (positions of the coming segment) and for ring space (the strip below must have consumed what is about to be
// overwritten).  The waits cannot form a cycle (a producer blocks only when >= kStripRing-64 positions ahead, a
// consumer only when < 128 behind), every wavefront runs the same number of segments per round, and every wait
// is bounded: on expiry the workgroup raises `status` and drains.
//
// MODE kStripDirs : one greedy traceback decision per cell (smithwaterman.cpp:51-72), 2 bits, laid out
//                   dirs[stream position][64 * strip + lane][W] dwords (W = 1 for R <= 16) — the layout of
//                   sw_wave_kernel.h with 64*nstrips lanes instead of 16; sw_wave_walk_kernel reads both.
// MODE kStripMax  : the MAXIMUM over the stream positions >= own_lo and the first cell holding it in the float engine's
//                   storage order (similaritymatrix.cpp:21-28) -> *best, cell — for the sub-chunks in which the float
//                   engine's saturating float16 score sweep reached its cap (host_pipeline.h locate_saturated).
// MODE kStripTrack: the first cell in the engine's storage order (order_key<>, sw_exact_kernel.h) among the cells
//                   equal to `target` at stream positions >= own_lo (the locate step of DESIGN.md §4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_exact_kernel.h"  // order_key
#include "sw_wave_kernel.h"   // WaveScoring, kDir*

namespace mi355sw {

struct StripProblem {
  const uint8_t *a;      // x (rows)
  const uint8_t *b;      // window of y: stream position t is b[t]
  int32_t na, nb;
  int32_t nstrips;       // ceil(na / (64*R))
  int32_t nw;            // wavefronts that take part: min(nstrips, blockDim.x / 64)
  float *gbound;         // 2 x gstride floats: bottom rows that cross rounds (null when nstrips <= nw)
  int64_t gstride;
  uint32_t *dirs;        // kStripDirs: [nb][64*nstrips][W]
  // kStripTrack
  float target;
  int32_t own_lo;        // first stream position (0-based) that competes
  int64_t col_offset;    // true column of stream position t = col_offset + t + 1
  int64_t full_n;        // |y| of the full problem (uint8 storage order)
  int64_t *cell;         // [2 per workgroup of the problem] row, true column of the first cell equal to target; row 0 when none
  float *best;           // kStripMax: [per workgroup] the maximum (0 when no positive cell competes)
  int32_t *status;       // 0 = complete, 1 = a pipeline wait expired (result unusable)
  int32_t spg;           // > 0: the strips are dealt to several WORKGROUPS, spg consecutive strips each (one wavefront per
                         // strip, one round); the bottom row of a workgroup's last strip reaches the next workgroup
                         // through gbound + g * gstride, progress through gcount[g] (global, zero at launch)
  long long *gcount;
  int32_t fault;         // test hook (option fault_inject = strip_stall): wavefront 0 never reports progress, so the
                         // strip below it runs into the bounded wait and the workgroup takes the expiry path
  // A BLOCK of a larger problem whose left column and top row the score sweep saved (sw_long_kernel.h colsave / rowsave,
  // host_saved.h): rows row0 + 1 .. row0 + na of the query (a = x + row0), starting right of the saved column instead of a
  // zero border; lane registers start from init, the first strip's boundary row comes complete from top.  Null: zero border.
  const float *init;     // init[ai] = H(row0 + ai + 1, column in front of stream position 0) * in_scale^-1, ai = -1 .. (row0 + ai >= 0)
  const float *top;      // top[t] = H(row0, column of stream position t) * in_scale^-1 (row0 > 0)
  float in_scale;        // the saved values are the sweep's scaled cells: H = value * in_scale
  int32_t row0;          // rows of the query above this block (true row = row0 + local row)
  int32_t na_full;       // |x| of the full problem (uint8 storage order)
  int32_t lt;            // kStripDirs: lanes of a decision row (64 * strips of the FULL problem); 0: 64 * nstrips
};

enum : int { kStripDirs = 0, kStripTrack = 1, kStripMax = 2 };
constexpr int kStripRing = 512;          // boundary positions held per strip (8 segments)
constexpr int kStripMaxWaves = 16;
constexpr int kStripSpinLimit = 1 << 22; // polls (with s_sleep) before a wait is declared dead

// LUT = true (float engine): scores come from a table tab[257][ncodes] in dynamic LDS (row = query byte, row 256 =
// padding row; column = reference code, column ncodes-1 = padding) and the stream is the window of reference
// CODES; LUT = false: identity scoring on raw bytes, no table.
// groups > 1: the grid holds `groups` workgroups per problem — the sweep of ONE long alignment's window on several CUs (a
// workgroup of sixteen wavefronts is bound by its CU's issue rate: 31 ms for the decisions of config 5's 26 k-column window,
// 8.6 ms for the candidate windows of its locate step); all workgroups of a launch are resident at once (the host keeps the
// grid far below the CU count), the chain of waits is acyclic (strip s only waits for strip s - 1) and every wait is bounded
// as before.  kStripMax / kStripTrack: every workgroup reports the best cell of ITS strips (best[grp], cell[2 grp ..]); the
// host merges them by (value, storage-order key).
template <int R, bool U8, int MODE, bool LUT = false>
__global__ __launch_bounds__(64 * kStripMaxWaves) void sw_strip_kernel(const StripProblem *probs, const WaveScoring sc,
                                                                       const float *gtab = nullptr, int ncodes = 0, int groups = 1) {
  extern __shared__ float tab[];
  __shared__ float ring[kStripMaxWaves][kStripRing];                // [w] = bottom row of wavefront w's strip (the strip below reads it)
  __shared__ long long produced[kStripMaxWaves], consumed[kStripMaxWaves + 1];
  __shared__ int dead;
  __shared__ unsigned long long wkey[kStripMaxWaves];
  __shared__ long long wi[kStripMaxWaves], wj[kStripMaxWaves];
  const bool multi = groups > 1;
  const StripProblem P = probs[multi ? blockIdx.x / groups : blockIdx.x];
  const int grp = multi ? blockIdx.x % groups : 0;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;   // (w in an SGPR: what derives from it branches uniformly)
  if (tid < kStripMaxWaves) produced[tid] = 0;
  if (tid <= kStripMaxWaves) consumed[tid] = 0;
  if (tid == 0) dead = 0;
  if (LUT) {
    for (int e = tid; e < 257 * ncodes; e += blockDim.x) tab[e] = e < 256 * ncodes ? gtab[e] : -1.0e30f;
  }
  __syncthreads();
  const int na = P.na, nb = P.nb;
  const int nfull = P.na_full > 0 ? P.na_full : na;                  // rows of the full problem (a block knows only its own)
  const int nstrips = P.nstrips;
  const int s_first = multi ? grp * P.spg : 0;                       // this workgroup's first strip
  const int nw = multi ? min(P.spg, nstrips - s_first) : P.nw;       // wavefronts that take part (multi: one per strip)
  if (nw <= 0) return;
  const int LT = P.lt > 0 ? P.lt : 64 * nstrips;
  constexpr int W = (R + 15) / 16;
  const bool top_saved = P.top != nullptr;                           // block of a larger problem: the row above it is in HBM
  const int nseg = (nb + 64 + 63) / 64;                  // lane 63 reaches stream position nb - 1
  const long long NBP = (long long)nseg * 64;            // counter units per round
  const int rounds = multi ? 1 : (nstrips + nw - 1) / nw;
  const float gpen = U8 ? sc.u8G : sc.gap;
  bool ok = true;

  unsigned long long bkey = ~0ull;                       // kStripTrack: this lane's first competing cell
  long long bi = 0, bj = 0;
  float bval = 0.0f;                                     // kStripMax: this lane's best value so far

  auto wait_for = [&](long long *counter, long long need) {
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
      if (__hip_atomic_load(&dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || ++spins > kStripSpinLimit) {
        __hip_atomic_store(&dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ok = false;
        return;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  };

  auto wait_global = [&](long long *counter, long long need) {        // progress of the workgroup above (device scope)
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
      if (__hip_atomic_load(&dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 || ++spins > kStripSpinLimit) {
        __hip_atomic_store(&dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ok = false;
        return;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  };

  // stream byte of position seg*64 + l, flagged where the stream has ended (LUT: the padding code; else bit 8, which no
  // query byte carries: positions outside the stream compare unequal to everything)
  const uint32_t off_stream = LUT ? (uint32_t)(ncodes - 1) : 0x100u;
  auto stage_load = [&](int seg) -> uint32_t {
    const int t = seg * 64 + l;
    return (t < nb) ? (uint32_t)P.b[t] : off_stream;
  };
  // one-lane shifts across the wavefront (as sw_long_kernel.h): shr1 = lanes take the lane above, lane 0 keeps `old`;
  // rot1 = lanes take the lane below; shl1_insert = lanes take the lane below, lane 63 takes `ins`
  auto shr1 = [](uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); };
  auto rot1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true); };
  auto shl1_insert = [](uint32_t ins, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)ins, (int)v, 0x130 /*wave_shl:1*/, 0xf, 0xf, false); };

  for (int round = 0; round < rounds && ok && w < nw; ++round) {
    const int s = s_first + round * nw + w;              // this wavefront's strip in this round
    if (s >= nstrips) break;
    const long long base = (long long)round * NBP;
    const bool has_in = s > 0 || top_saved, has_out = s + 1 < nstrips;
    const bool in_saved = top_saved && s == 0;           // complete in HBM before the launch: nothing to wait for
    const bool in_global = has_in && w == 0;             // from the last wavefront of the previous round
    const bool out_global = has_out && w == nw - 1;      // to wavefront 0 of the next round
    const float *rin = ring[w > 0 ? w - 1 : 0];
    float *rout = ring[w];
    const float *gin = in_saved ? P.top : (in_global ? P.gbound + (size_t)(multi ? grp - 1 : ((round + 1) & 1)) * (size_t)P.gstride : nullptr);
    float *gout = out_global ? P.gbound + (size_t)(multi ? grp : (round & 1)) * (size_t)P.gstride : nullptr;

    uint32_t ca[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ai = (s * 64 + l) * R + r;
      if (LUT) ca[r] = ((ai < na) ? (uint32_t)P.a[ai] : 256u) * (uint32_t)ncodes;   // row offset into tab
      else ca[r] = (ai < na) ? (uint32_t)P.a[ai] : 0xFFFFu;                          // padding rows never match
    }
    // The inner loop touches no LDS (LUT: the score table only).  What a step needs from outside the lane travels through
    // registers: `code` moves one lane up per step and lane 0 takes the next byte of the segment from lane 0 of cseg, which
    // rotates one lane down per step; bseg holds the segment's 64 boundary values (one ring / global read per lane and
    // SEGMENT) and rotates the same way, lane 0 of it being the `old` operand of the H shift; oseg collects lane 63's
    // bottom-row value of every step: one ring / global write per segment.  (With one wavefront per SIMD — a few long
    // problems dealt to several workgroups — two dependent LDS round trips per step were most of the step.)
    uint32_t code = off_stream;
    uint32_t curc = stage_load(0), nextc = stage_load(1);

    float H[R];
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = 0.0f;
    uint32_t up_prev = 0;

    for (int seg = 0; seg < nseg && ok; ++seg) {
      // input: boundary positions seg*64 .. seg*64+63 (the strip above finishes them during ITS segment seg+1)
      if (has_in && !in_saved) {
        const long long need = (seg + 1) * 64 < nb ? (seg + 1) * 64 : nb;
        if (in_global) {
          if (multi) wait_global(P.gcount + (grp - 1), need);
          else wait_for(&produced[nw - 1], base - NBP + need);
        } else {
          wait_for(&produced[w - 1], base + need);
        }
      }
      // ring space: this segment stores positions <= seg*64, over the slots of positions <= seg*64 - kStripRing
      if (has_out && !out_global) wait_for(&consumed[w + 1], base + seg * 64 - kStripRing + 64);
      if (!ok) break;
      uint32_t bseg = 0u;
      {
        const int tl = seg * 64 + l;
        if (has_in && tl < nb) {
          if (in_saved) bseg = __float_as_uint(gin[tl] * P.in_scale);
          else bseg = __float_as_uint(in_global ? __hip_atomic_load(gin + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rin[tl & (kStripRing - 1)]);
        }
      }
      // a block that starts right of a saved column: lane l reaches stream position 0 at step l of the first segment and
      // takes its rows of that column (and the row above them, for its first diagonal term) then
      float hinit[R];
      float dinit = 0.0f;
      const bool seeded = P.init != nullptr && seg == 0;
      if (seeded) {
        const int ai0 = (s * 64 + l) * R;
#pragma unroll
        for (int r = 0; r < R; ++r) hinit[r] = (ai0 + r < na) ? P.init[ai0 + r] * P.in_scale : 0.0f;
        dinit = (P.row0 + ai0 > 0) ? P.init[ai0 - 1] * P.in_scale : 0.0f;
      }
      uint32_t cseg = curc;
      curc = nextc;
      nextc = stage_load(seg + 2);
      uint32_t oseg = 0u;
#pragma unroll 2
      for (int k = 0; k < 64; ++k) {
        const int t0 = seg * 64 + k;                                     // lane 0's stream position
        const int t = t0 - l;
        {
          const uint32_t head = cseg;
          cseg = rot1(cseg);
          code = shr1(head, code);
        }
        const uint32_t cb = code;
        if (seeded && k == l) {
#pragma unroll
          for (int r = 0; r < R; ++r) H[r] = hinit[r];
          up_prev = __float_as_uint(dinit);
        }
        uint32_t up;                                                     // H(first row of the strip - 1, this column)
        if (has_in) { const uint32_t head = bseg; bseg = rot1(bseg); up = shr1(head, __float_as_uint(H[R - 1])); }
        else up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
        float diag = __uint_as_float(up_prev);
        float north = __uint_as_float(up);
        up_prev = up;
        uint32_t dpack[W];
#pragma unroll
        for (int d = 0; d < W; ++d) dpack[d] = 0;
        bool hit = false;                                                // kStripTrack: some cell of this step equals target
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float wv = H[r];
          const bool eq = ca[r] == cb;
          float x;
          if (LUT) x = diag + tab[ca[r] + cb];
          else if (U8) x = eq ? fminf(diag + sc.u8M, 255.0f) : fmaxf(diag - sc.u8X, 0.0f);
          else x = diag + (eq ? sc.match : sc.mismatch);
          const float tmx = fmaxf(wv, north);
          const float h = fmaxf(fmaxf(x, tmx - gpen), 0.0f);
          if (MODE == kStripDirs) {
            // smithwaterman.cpp:51,59,66,72 at this cell (n1 = NW = diag, n2 = W = wv, n3 = N = north): stop when a
            // neighbour is 0, else NW if it is >= both others, else W if it is >= N, else N
            // (arithmetic on the three conditions: written as nested selects the compiler turns it into branches)
            const float lowest = fminf(fminf(diag, wv), north);
            const uint32_t c_go = lowest != 0.0f ? 1u : 0u, c_nw = diag >= tmx ? 1u : 0u, c_w = wv >= north ? 1u : 0u;
            const uint32_t dir = c_go * (3u - c_w - c_nw * (2u - c_w));   // 0 stop, 1 NW, 2 W, 3 N
            dpack[r >> 4] |= (uint32_t)dir << (2 * (r & 15));
          } else if (MODE == kStripMax) {
            hit |= h >= bval && h > 0.0f;
          } else {
            hit |= h == P.target;
          }
          diag = wv;
          H[r] = h;
          north = h;
        }
        if (MODE == kStripTrack) {
          // rare path, out of the recurrence: which rows, and where they stand in the storage order
          if (hit && t >= P.own_lo && t < nb) {
            const long long j = P.col_offset + t + 1;
            uint32_t rows = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) rows |= (H[r] == P.target ? 1u : 0u) << r;
            while (rows) {
              const int r = __builtin_ctz(rows);
              rows &= rows - 1;
              const long long il = (long long)(s * 64 + l) * R + r + 1, i = il + P.row0;
              if (il <= na) {
                const unsigned long long key = U8 ? order_key<1>(i, j, nfull, P.full_n) : order_key<0>(i, j, nfull, P.full_n);
                if (key < bkey) { bkey = key; bi = i; bj = j; }
              }
            }
          }
        }
        if (MODE == kStripMax) {
          // seldom: a cell reaches this lane's best so far (a later strip of the same lane sweeps the same columns again,
          // so ties are settled by the storage-order key, not by time)
          if (hit && t >= P.own_lo && t < nb) {
            const long long j = P.col_offset + t + 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const long long il = (long long)(s * 64 + l) * R + r + 1, i = il + P.row0;
              if (H[r] >= bval && H[r] > 0.0f && il <= na) {
                const unsigned long long key = U8 ? order_key<1>(i, j, nfull, P.full_n) : order_key<0>(i, j, nfull, P.full_n);
                if (H[r] > bval || key < bkey) { bval = H[r]; bkey = key; bi = i; bj = j; }
              }
            }
          }
        }
        if (has_out) oseg = shl1_insert(__float_as_uint(H[R - 1]), oseg);    // lane 63 inserts, the others pass down
        if (MODE == kStripDirs) {
          if (t >= 0 && t < nb) {
            uint32_t *dst = P.dirs + ((size_t)t * LT + (size_t)(s * 64 + l)) * W;
#pragma unroll
            for (int d = 0; d < W; ++d) dst[d] = dpack[d];
          }
        }
      }
      // lane j of oseg holds lane 63's value of step j = stream position seg*64 + j - 63: positions <= seg*64 are complete;
      // this wavefront has read positions <= seg*64 + 63
      if (has_out) {
        const int t = seg * 64 - 63 + l;
        if (out_global) {
          if (t >= 0 && t < nb) __hip_atomic_store(gout + t, __uint_as_float(oseg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        } else {
          if (t >= 0) rout[t & (kStripRing - 1)] = __uint_as_float(oseg);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
      }
      if (multi && out_global && l == 0)                               // (the release covers this wavefront's stores above)
        __hip_atomic_store(P.gcount + grp, (long long)seg * 64 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (l == 0) {
        if (has_out && !(P.fault && w == 0)) __hip_atomic_store(&produced[w], base + seg * 64 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&consumed[w], base + (seg + 1) * 64, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (multi && out_global && l == 0 && ok) __hip_atomic_store(P.gcount + grp, NBP, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (l == 0 && ok && !(P.fault && w == 0)) {
      // the whole round of this strip is done: releases every wait of this round on this wavefront
      __hip_atomic_store(&produced[w], base + NBP, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&consumed[w], base + NBP, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (l == 0 && w < nw) {
    // whatever happened, nobody may wait on this wavefront any more
    __hip_atomic_store(&produced[w], 0x7FFFFFFFFFFFFFFFll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&consumed[w], 0x7FFFFFFFFFFFFFFFll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!ok) *P.status = 1;
    if (multi && w == nw - 1) __hip_atomic_store(P.gcount + grp, 0x7FFFFFFFFFFFFFFFll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == kStripMax) {
    __shared__ float wval[kStripMaxWaves];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(bval, off);
      const unsigned long long ok2 = __shfl_xor(bkey, off);
      const long long oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
      if (ov > bval || (ov == bval && ok2 < bkey)) { bval = ov; bkey = ok2; bi = oi; bj = oj; }
    }
    if (l == 0) { wval[w] = bval; wkey[w] = bkey; wi[w] = bi; wj[w] = bj; }
    __syncthreads();                                                     // every wavefront gets here: all waits are bounded
    if (tid == 0) {
      for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
        if (wval[k] > bval || (wval[k] == bval && wkey[k] < bkey)) { bval = wval[k]; bkey = wkey[k]; bi = wi[k]; bj = wj[k]; }
      P.best[grp] = bval;
      P.cell[2 * grp] = bval > 0.0f ? bi : 0;
      P.cell[2 * grp + 1] = bval > 0.0f ? bj : 0;
    }
  }
  if (MODE == kStripTrack) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long ok2 = __shfl_xor(bkey, off);
      const long long oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
      if (ok2 < bkey) { bkey = ok2; bi = oi; bj = oj; }
    }
    if (l == 0) { wkey[w] = bkey; wi[w] = bi; wj[w] = bj; }
    __syncthreads();                                                     // every wavefront gets here: all waits are bounded
    if (tid == 0) {
      for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
        if (wkey[k] < bkey) { bkey = wkey[k]; bi = wi[k]; bj = wj[k]; }
      P.cell[2 * grp] = bkey != ~0ull ? bi : 0;
      P.cell[2 * grp + 1] = bkey != ~0ull ? bj : 0;
    }
  }
}

}  // namespace mi355sw
