// sw_wave_kernel.h — wavefront-in-registers exact kernel for SMALL problems (one side <= 16*R <= 512).
//
// Same register wavefront as sw_score_kernel.h (16 lanes = one problem, lane l owns R consecutive cells of the
// short side, one DPP row_shr:1 per step), but in float32 cells, one problem per slot, identity scoring, and with
// the two things the score kernel leaves to sw_exact_kernel.h:
//   TRACK  the first maximum in the float engine's storage order (similaritymatrix.cpp:21-28: columns outer,
//          rows inner, strict '>') -> (score, row, column);
//   DIRS   one greedy traceback decision per cell (smithwaterman.cpp:51-72), 2 bits, laid out
//          dirs[stream step][lane][W] dwords (W = 1 for R <= 16, else 2): every lane stores its R decisions of a
//          step as one or two dwords, a slot stores 64*W contiguous bytes per step.
// It serves (a) the many-small-alignments batch (src/mpi_sw_solve_uniprot.cpp:95-138 shape: each database
// sequence x against one short query y) and (b) the traceback windows of the score-kernel path.
//
// ORIENT = 0: lanes hold rows of x (i), the stream runs over columns of y (j)   [short read, long window]
// ORIENT = 1: lanes hold columns of y (j), the stream runs over rows of x (i)   [long sequence, short query]
// U8 = true evaluates the uint8 engine's cell rule (similaritymatrix.cpp:75-81) on integer-valued floats.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_exact_kernel.h"   // kDir*
#include "sw_score_kernel.h"   // lane_stride, u32x4, kPadScoreF

namespace mi355sw {

struct WaveProblem {
  const uint8_t *a;      // sequence held on lanes: x (ORIENT 0) or y (ORIENT 1); na <= 16*R
  const uint8_t *b;      // streamed sequence: window of y (ORIENT 0) or window of x (ORIENT 1)
  int32_t na, nb;
  int64_t b_offset;      // true (1-based) stream index = b_offset + t + 1 for stream position t
  uint32_t *dirs;        // DIRS: [nb + 15][16][W] packed decisions (2 bits per cell, cell r of a lane at bit 2*(r%16)); lane l's
                         // decisions for stream position t are in row t + l (the step they were made at); or null
  float *best;           // TRACK: maximum (0 when no positive cell)
  int64_t *cell;         // TRACK: [2] = row (into x), column (into y), 1-based, of the first maximum
  // KEYED tracking (ORIENT 0): the first cell in the engine's storage order (order_key<>, sw_exact_kernel.h) among
  // the cells equal to `target` at stream positions >= own_lo; best = target when found, else -1
  float target;
  int32_t own_lo;
  int64_t full_n;        // |y| of the full problem (uint8 storage order)
  // sw_wave_prof_kernel only (checkpointed whole problems, host_batch.h).  TRACK without DIRS: where the state of the slot's
  // wavefront is saved after every kCkptEvery-th step (null: nowhere).  DIRS without TRACK: k0 > 0 resumes the problem at step k0
  // (a multiple of kCkptEvery) from the state saved there; nb is then the END of the rows to run, dirs rows count from step k0.
  float *ckpt;
  int32_t k0;
  // DIRS resumed from a state sw_wave_prof16_kernel saved: that kernel stores its packed registers as they are, one row of
  // 16 x (R + 1) dwords per saved state for the PAIR of problems a slot runs (ckpt of the pair's first problem); 1 / 2 = this
  // problem was the low / high half (0: float32 states of its own, sw_wave_prof_kernel<TRACK>)
  int32_t ck_half;
  // DIRS of a window that ENDS at the argmax (host_batch.h): the walk only moves up and to the left, so decisions are wanted of
  // the lanes up to the argmax column's; the launch then runs nb - k0 + lanes_used steps instead of nb - k0 + 16 (0: all lanes)
  int32_t lanes_used;
};

struct WaveScoring {
  float match, mismatch, gap;
  float u8M, u8X, u8G;   // uint8 engine parameters as floats
};

constexpr int kWaveSeg = 64;
constexpr int kWaveBuf = 16 + kWaveSeg;
constexpr int kCkptEvery = 32;           // steps between two saved states of a checkpointed pass (sw_wave_prof_kernel / sw_wave_prof16_kernel);
                                         // 16 was measured: decision pass 1.51 -> 1.32 ms, first pass 2.81 -> 2.93 ms, twice the states: not taken
constexpr int kCkptPerSeg = kWaveSeg / kCkptEvery;
static_assert(kWaveSeg % kCkptEvery == 0 && kCkptEvery % 4 == 0, "states are saved inside and at the end of every 64-step segment");

template <int R, int ORIENT, bool U8, bool TRACK, bool DIRS, bool KEYED = false>
__global__ __launch_bounds__(256) void sw_wave_kernel(const WaveProblem *probs, int nprob, const WaveScoring sc) {
  static_assert(!KEYED || (TRACK && !DIRS && ORIENT == 0), "keyed tracking: locate windows, lanes = rows of x");
  __shared__ __attribute__((aligned(16))) uint8_t win[16 * kWaveBuf];
  const int tid = threadIdx.x;
  const int l = tid & 15;
  const int slot = tid >> 4;
  const int pid = blockIdx.x * 16 + slot;
  const bool active = pid < nprob;
  WaveProblem P;
  if (active) P = probs[pid];
  else { P.a = nullptr; P.b = nullptr; P.na = 0; P.nb = 0; P.b_offset = 0; P.dirs = nullptr; P.best = nullptr; P.cell = nullptr;
         P.target = -1.0f; P.own_lo = 0; P.full_n = 0; P.ckpt = nullptr; P.k0 = 0; P.ck_half = 0; P.lanes_used = 0; }
  const int na = P.na, nb = P.nb;

  // this lane's R characters of the short side (0xFFFF = padding, never equal to a byte)
  uint32_t ca[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ai = l * R + r;
    ca[r] = (ai < na) ? (uint32_t)P.a[ai] : 0xFFFFu;
  }

  // stream window: 16 B history + 64 B segment per slot, refilled every 64 steps (as sw_score_kernel.h)
  uint8_t *buf = win + slot * kWaveBuf;
  uint32_t *buf32 = reinterpret_cast<uint32_t *>(buf);
  const uint8_t *buf_lane = buf + 16 - l;
  auto stage_load = [&](int seg) -> uint32_t {
    const int c0 = seg * kWaveSeg + 4 * l;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = c0 + k;
      const uint32_t ch = (t < nb) ? (uint32_t)P.b[t] : 0u;
      w |= ch << (8 * k);
    }
    return w;
  };
  // steps this wavefront needs: the longest of its four slots (+15 skew); wave-uniform
  int steps = nb + 16;
  steps = max(steps, __shfl_xor(steps, 16));
  steps = max(steps, __shfl_xor(steps, 32));
  const int nseg = (steps + kWaveSeg - 1) / kWaveSeg;

  uint32_t nextc = stage_load(0);
  if (l < 4) buf32[l] = 0u;
  buf32[4 + l] = nextc;
  nextc = stage_load(1);

  float H[R];
#pragma unroll
  for (int r = 0; r < R; ++r) H[r] = 0.0f;
  uint32_t up_prev = 0;
  // TRACK state: ORIENT 0 -> one (best, step, row) per lane; ORIENT 1 -> one (best, step) per lane row
  float tb0 = 0.0f;
  int tt0 = 0, tr0 = 0;
  float tbr[ORIENT == 1 && TRACK ? R : 1];
  int ttr[ORIENT == 1 && TRACK ? R : 1];
  if (ORIENT == 1 && TRACK) {
#pragma unroll
    for (int r = 0; r < R; ++r) { tbr[r] = 0.0f; ttr[r] = 0; }
  }
  const float gpen = U8 ? sc.u8G : sc.gap;
  unsigned long long bkey = ~0ull;                                 // KEYED: this lane's first competing cell
  long long kbi = 0, kbj = 0;

  for (int seg = 0; seg < nseg; ++seg) {
#pragma unroll 4
    for (int k = 0; k < kWaveSeg; ++k) {
      const int t = seg * kWaveSeg + k - l;                        // this lane's stream position
      // positions outside the stream compare unequal to every byte (bit 8 set)
      const uint32_t cb = (uint32_t)buf_lane[k] | ((uint32_t)t >= (uint32_t)nb ? 0x100u : 0u);
      const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x111, 0xf, 0xf, true);
      float diag = __uint_as_float(up_prev);
      float north = __uint_as_float(up);                           // previous lane row, same step
      up_prev = up;
      constexpr int W = (R + 15) / 16;
      uint32_t dpack[W];
      if (DIRS) {
#pragma unroll
        for (int d = 0; d < W; ++d) dpack[d] = 0;
      }
      bool hit = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float w = H[r];                                      // same lane row, previous step
        const bool eq = ca[r] == cb;
        float x;
        if (U8) x = eq ? fminf(diag + sc.u8M, 255.0f) : fmaxf(diag - sc.u8X, 0.0f);
        else x = diag + (eq ? sc.match : sc.mismatch);
        const float tmx = fmaxf(w, north);
        const float y = tmx - gpen;
        const float h = fmaxf(fmaxf(x, y), 0.0f);
        if (DIRS) {
          // neighbours in the reference's terms: n1 = NW, n2 = W (same row of x, previous column of y), n3 = N
          const float n1 = diag;
          const float n2 = ORIENT == 0 ? w : north;
          const float n3 = ORIENT == 0 ? north : w;
          // stop when a neighbour is 0 (all are >= 0), else NW if it is >= both others, else W if it is >= N, else N
          // (arithmetic on the three conditions: written as nested selects the compiler turns it into branches)
          const float lowest = fminf(fminf(n1, n2), n3);
          const uint32_t c_go = lowest != 0.0f ? 1u : 0u, c_nw = n1 >= tmx ? 1u : 0u, c_w = n2 >= n3 ? 1u : 0u;
          const uint32_t dir = c_go * (3u - c_w - c_nw * (2u - c_w));   // 0 stop, 1 NW, 2 W, 3 N
          dpack[r >> 4] |= (uint32_t)dir << (2 * (r & 15));
        }
        if (KEYED) hit |= h == P.target;
        else if (TRACK) {
          if (ORIENT == 0) { if (h > tb0) { tb0 = h; tt0 = t; tr0 = r; } }
          else { if (h > tbr[r]) { tbr[r] = h; ttr[r] = t; } }
        }
        diag = w;
        H[r] = h;
        north = h;
      }
      if (KEYED) {
        // rare path, out of the recurrence: which rows, and where they stand in the storage order
        if (hit && t >= P.own_lo && t < nb) {
          const long long j = P.b_offset + t + 1;
          uint32_t rows = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) rows |= (H[r] == P.target ? 1u : 0u) << r;
          while (rows) {
            const int r = __builtin_ctz(rows);
            rows &= rows - 1;
            const long long i = (long long)l * R + r + 1;
            if (i <= na) {
              const unsigned long long key = U8 ? order_key<1>(i, j, na, P.full_n) : order_key<0>(i, j, na, P.full_n);
              if (key < bkey) { bkey = key; kbi = i; kbj = j; }
            }
          }
        }
      }
      if (DIRS) {
        // row = the STEP (t + l): the sixteen lanes of a slot write one contiguous 64 W-byte row per step
        if (P.dirs != nullptr && t >= 0 && t < nb) {
          uint32_t *dst = P.dirs + ((size_t)(seg * kWaveSeg + k) * 16 + (size_t)l) * W;
#pragma unroll
          for (int d = 0; d < W; ++d) dst[d] = dpack[d];
        }
      }
    }
    const uint32_t hist = buf32[kWaveSeg / 4 + (l & 3)];
    if (l < 4) buf32[l] = hist;
    buf32[4 + l] = nextc;
    nextc = stage_load(seg + 2);
  }

  if (KEYED) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      const unsigned long long ok = __shfl_xor(bkey, off, 16);
      const long long oi = __shfl_xor(kbi, off, 16), oj = __shfl_xor(kbj, off, 16);
      if (ok < bkey) { bkey = ok; kbi = oi; kbj = oj; }
    }
    if (l == 0 && active) {
      *P.best = bkey != ~0ull ? P.target : -1.0f;
      P.cell[0] = bkey != ~0ull ? kbi : 0;
      P.cell[1] = bkey != ~0ull ? kbj : 0;
    }
  } else if (TRACK) {
    // per-lane winner in storage order (column of y first, then row of x), then across the 16 lanes
    float bv;
    long long bi, bj;
    if (ORIENT == 0) {
      bv = tb0; bi = (long long)l * R + tr0 + 1; bj = P.b_offset + tt0 + 1;
    } else {
      bv = 0.0f; bi = 0; bj = 0;
#pragma unroll
      for (int r = 0; r < R; ++r)                                  // ascending column: strict '>' keeps the first
        if (tbr[r] > bv) { bv = tbr[r]; bj = (long long)l * R + r + 1; bi = P.b_offset + ttr[r] + 1; }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 16);
      const long long oi = __shfl_xor(bi, off, 16);
      const long long oj = __shfl_xor(bj, off, 16);
      if (ov > bv || (ov == bv && ov > 0.0f && (oj < bj || (oj == bj && oi < bi)))) { bv = ov; bi = oi; bj = oj; }
    }
    if (l == 0 && active) {
      *P.best = bv;
      P.cell[0] = bv > 0.0f ? bi : 0;
      P.cell[1] = bv > 0.0f ? bj : 0;
    }
  }
}

// sw_wave_prof_kernel — ORIENT 1 of sw_wave_kernel (lanes hold columns of the SHARED second sequence y, the stream runs
// over the rows of a database sequence x: the many-small-alignments batch, src/mpi_sw_solve_uniprot.cpp:95-138) on the cell of
// the score kernels instead of compare-and-select arithmetic.  sw_wave_kernel spends ~10 VALU instructions per cell there
// with the first-maximum tracking (compare, select, add, max, sub, max, max; compare + two selects per cell for the
// tracking) and is VALU-issue-bound (profiles/r03_pmc_config4_*.json); here
//   * every problem of a workgroup has the same y, so ONE query profile serves all sixteen: prof[code][lane][r] = score of
//     y's column lane*R + r against the letter `code` (y's distinct letters + "other"), float32 scaled by 2^-k, in LDS;
//     the stream's bytes are translated to codes as they are staged into the slot's window;
//   * cell: x = clamp(NW + s) (v_add_f32 clamp: the [0, 1] clamp is the zero floor), H = max3(x, W - g, N - g), keep H - g:
//     three instructions, two of them at the double issue rate;
//   * TRACK: the first maximum in storage order (column of y, then row of x) is kept per LANE as ONE orderable key and the
//     row it was first seen at.  The host takes this kernel only when every reachable cell value has its five lowest
//     mantissa bits clear (scores that are multiples of a power of two, small against 2^18 of it): the key of a cell is its
//     float bits OR (31 - column within the lane) — still a positive float, ordered by (value, then smaller column) — so one
//     v_or per cell, one max3 per two cells and, per STEP, a compare, a max and a select (strict '>' keeps the first row)
//     replace the per-cell compare + two selects; nothing branches.
// Values, decisions and the winner are those of sw_wave_kernel<R, 1, false, TRACK, DIRS> (a power-of-two scaling commutes
// with every add, subtract, maximum and comparison).
struct WaveProfArgs {
  const uint8_t *lut;        // [256] byte -> code; ncodes - 1 = "other" (a byte that y does not contain)
  const uint8_t *byte_of;    // [ncodes - 1] code -> byte
  int32_t ncodes;
  float match_s, mismatch_s, gap_s;   // scores * 2^-k
  float unscale;             // 2^k
  float ck16_scale;          // a state saved by sw_wave_prof16_kernel: this kernel's cell = that float16 value * ck16_scale
};

template <int R, bool TRACK, bool DIRS>
__global__ __launch_bounds__(256) void sw_wave_prof_kernel(const WaveProblem *probs, int nprob, const WaveProfArgs sa) {
  constexpr int LS = lane_stride(R);                               // dwords between the profile rows of adjacent lanes
  constexpr int NQ4 = (R + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) uint32_t wsmem[];
  __shared__ __attribute__((aligned(16))) uint8_t win[16 * kWaveBuf];
  __shared__ uint8_t lut_s[256];
  float *prof = reinterpret_cast<float *>(wsmem);                  // [ncodes][16][LS]
  const int tid = threadIdx.x;
  const int l = tid & 15;
  const int slot = tid >> 4;
  const int pid = blockIdx.x * 16 + slot;
  const bool active = pid < nprob;
  WaveProblem P;
  if (active) P = probs[pid];
  else { P.a = nullptr; P.b = nullptr; P.na = 0; P.nb = 0; P.b_offset = 0; P.dirs = nullptr; P.best = nullptr; P.cell = nullptr;
         P.target = -1.0f; P.own_lo = 0; P.full_n = 0; P.ckpt = nullptr; P.k0 = 0; P.ck_half = 0; P.lanes_used = 0; }
  const int nb = P.nb;
  // the lane side is the same for every problem of the launch (the range of the resident reference)
  const uint8_t *ya = probs[blockIdx.x * 16].a;
  const int na = probs[blockIdx.x * 16].na;
  const uint32_t other = (uint32_t)(sa.ncodes - 1);
  lut_s[tid] = sa.lut[tid];
  for (int e = tid; e < sa.ncodes * 16 * R; e += 256) {
    const int c = e / (16 * R);
    const int rem = e - c * 16 * R;
    const int ll = rem / R, r = rem - ll * R;
    const int j = ll * R + r;
    float v = kPadScoreF;                                          // padding columns: clamp to 0, never a maximum
    if (j < na) v = ((uint32_t)c < other && ya[j] == sa.byte_of[c]) ? sa.match_s : sa.mismatch_s;
    prof[(c * 16 + ll) * LS + r] = v;
  }
  __syncthreads();

  // stream window of CODES: 16 B history + 64 B segment per slot, refilled every 64 steps
  uint8_t *buf = win + slot * kWaveBuf;
  uint32_t *buf32 = reinterpret_cast<uint32_t *>(buf);
  const uint8_t *buf_lane = buf + 16 - l;
  const int k0 = (DIRS && !TRACK) ? P.k0 : 0;                       // first step of this launch (a resumed problem: > 0)
  auto stage_load = [&](int seg) -> uint32_t {                     // (seg = -1: the sixteen positions in front of step k0)
    const int c0 = k0 + seg * kWaveSeg + 4 * l;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = c0 + k;
      const uint32_t ch = ((uint32_t)t < (uint32_t)nb && (seg >= 0 || l >= 12)) ? (uint32_t)lut_s[P.b[t]] : other;   // outside the stream: matches nothing
      w |= ch << (8 * k);
    }
    return w;
  };
  int steps = nb - k0 + ((DIRS && !TRACK && P.lanes_used > 0) ? P.lanes_used : 16);
  steps = max(steps, __shfl_xor(steps, 16));
  steps = max(steps, __shfl_xor(steps, 32));
  const int nseg = (steps + kWaveSeg - 1) / kWaveSeg;
  const int steps4 = (steps + 3) & ~3;                             // the last segment stops at the wavefront's last step (in fours)

  uint32_t nextc = stage_load(0);
  {
    // history: the 16 positions in front of the first step (lanes 12..15 of stage_load(-1) cover k0 - 16 .. k0 - 1)
    const uint32_t hw = stage_load(-1);
    const uint32_t h4 = (uint32_t)__shfl((int)hw, 12 + (l & 3), 16);
    if (l < 4) buf32[l] = h4;
  }
  buf32[4 + l] = nextc;
  nextc = stage_load(1);

  float gv = sa.gap_s;
  asm volatile("" : "+v"(gv));                                     // (a VGPR operand: v_sub_f32 then issues at the double rate)
  float H[R], Hg[R];
  uint32_t up_prev = 0;
  if (DIRS && !TRACK && k0 > 0 && P.ckpt != nullptr) {
    const float *ck = P.ckpt + (size_t)l * (R + 1);
    if (P.ck_half == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) { H[r] = ck[r]; Hg[r] = H[r] - gv; }
      up_prev = __float_as_uint(ck[R]);
    } else {
      const uint32_t *ck2 = reinterpret_cast<const uint32_t *>(ck);
      const int sh = P.ck_half == 2 ? 16 : 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        H[r] = (float)__builtin_bit_cast(_Float16, (uint16_t)(ck2[r] >> sh)) * sa.ck16_scale;
        Hg[r] = H[r] - gv;
      }
      up_prev = __float_as_uint((float)__builtin_bit_cast(_Float16, (uint16_t)(ck2[R] >> sh)) * sa.ck16_scale);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) { H[r] = 0.0f; Hg[r] = -gv; }
  }
  float blk = 0.0f;                                                // TRACK: this lane's best key (value | 31 - column in the lane) ...
  int tl = 0;                                                      // ... and the stream position it was first seen at
  const float *prof_lane = prof + l * LS;

  // TRACK without DIRS: the slot's wavefront as it stands after step kCkptEvery (c + 1) - 1 — what a later launch needs to
  // resume there
  auto save_state = [&](int c) {
    if (P.ckpt != nullptr && (c + 1) * kCkptEvery < nb + 16) {
      float *ck = P.ckpt + ((size_t)c * 16 + (size_t)l) * (R + 1);
#pragma unroll
      for (int r = 0; r < R; ++r) ck[r] = H[r];
      ck[R] = __uint_as_float(up_prev);
    }
  };
  for (int seg = 0; seg < nseg; ++seg) {
    const int kq = min(kWaveSeg, steps4 - seg * kWaveSeg) >> 2;
    for (int k4 = 0; k4 < kq; ++k4) {
    if (TRACK && !DIRS && k4 != 0 && (4 * k4) % kCkptEvery == 0) save_state(kCkptPerSeg * seg + 4 * k4 / kCkptEvery - 1);
#pragma unroll
    for (int ku = 0; ku < 4; ++ku) {
      const int k = 4 * k4 + ku;
      const int t = k0 + seg * kWaveSeg + k - l;                   // this lane's stream position
      const uint32_t c = (uint32_t)buf_lane[k];
      const u32x4 *pp = static_cast<const u32x4 *>(__builtin_assume_aligned(prof_lane + c * (16 * LS), 16));
      uint32_t p[NQ4 * 4];
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const u32x4 v = pp[q];
        p[4 * q + 0] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
      }
      const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x111, 0xf, 0xf, true);
      float diag = __uint_as_float(up_prev);
      float north = __uint_as_float(up);                           // previous lane row, same step
      up_prev = up;
      float ng;
      asm("v_sub_f32 %0, %1, %2" : "=v"(ng) : "v"(north), "v"(gv));
      constexpr int W = (R + 15) / 16;
      uint32_t dpack[W];
      if (DIRS) {
#pragma unroll
        for (int d = 0; d < W; ++d) dpack[d] = 0;
      }
      float m = 0.0f, tpend = 0.0f;
      (void)tpend;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float w = H[r];                                      // same lane row, previous step
        float x, h;
        asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(x) : "v"(diag), "v"(__uint_as_float(p[r])));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(h) : "v"(x), "v"(Hg[r]), "v"(ng));
        if (DIRS) {
          // neighbours in the reference's terms (ORIENT 1): n1 = NW = diag, n2 = W = north (previous column of y), n3 = N = w.
          // Stop when a neighbour is 0 (all are >= 0), else NW if it is >= both others, else W if it is >= N, else N:
          // 0 stop, 1 NW, 2 W, 3 N.  The three comparisons land in wavefront masks, their combination is scalar work, and
          // each of the two bits is shifted into the lane's word by one add-with-carry (word = 2 * word + bit).
          float tmx, lowest;                                       // (asm: fmaxf / fminf would first canonicalise the asm-made inputs)
          asm("v_max_f32 %0, %1, %2" : "=v"(tmx) : "v"(w), "v"(north));
          asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lowest) : "v"(diag), "v"(north), "v"(w));
          const uint64_t c_go = __builtin_amdgcn_ballot_w64(lowest != 0.0f), c_nw = __builtin_amdgcn_ballot_w64(diag >= tmx),
                         c_w = __builtin_amdgcn_ballot_w64(north >= w);
          const uint64_t b_hi = c_go & ~c_nw, b_lo = c_go & (c_nw | ~c_w);
          uint64_t carry_out;
          asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(dpack[r >> 4]), "=s"(carry_out) : "v"(dpack[r >> 4]), "s"(b_lo));
          asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(dpack[r >> 4]), "=s"(carry_out) : "v"(dpack[r >> 4]), "s"(b_hi));
          (void)carry_out;
        }
        if (TRACK) {
          const float hk = __uint_as_float(__float_as_uint(h) | (uint32_t)(31 - r));     // (value, smaller column first)
          if (r & 1) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(tpend), "v"(hk));
          else if (r + 1 < R) tpend = hk;
          else m = fmaxf(m, hk);
        }
        diag = w;
        H[r] = h;
        north = h;
        asm("v_sub_f32 %0, %1, %2" : "=v"(ng) : "v"(h), "v"(gv));
        Hg[r] = ng;
      }
      if (TRACK) {
        // strict '>': an equal key (same value, same column) at a later row does not replace the first.  Cells beyond the
        // stream's end and padding columns hold values strictly below some real cell: they can lead a lane for a while, never
        // the slot.
        tl = m > blk ? t : tl;
        blk = fmaxf(blk, m);
      }
      if (DIRS) {
        // row = the STEP (t + l): the sixteen lanes of a slot write one contiguous 64 W-byte row per step
        if (P.dirs != nullptr && t >= 0 && t < nb) {
          uint32_t *dst = P.dirs + ((size_t)(seg * kWaveSeg + k) * 16 + (size_t)l) * W;
#pragma unroll
          for (int d = 0; d < W; ++d) {
            // the first cell's bits were shifted in first: reversed and moved down, cell r of the word sits at bit 2 * (r % 16)
            constexpr int full = 16;
            const int cells = (d + 1) * full <= R ? full : R - d * full;
            dst[d] = __builtin_bitreverse32(dpack[d]) >> (32 - 2 * cells);
          }
        }
      }
    }
    }
    if (TRACK && !DIRS && kq == kWaveSeg / 4) save_state(kCkptPerSeg * seg + kCkptPerSeg - 1);
    const uint32_t hist = buf32[kWaveSeg / 4 + (l & 3)];
    if (l < 4) buf32[l] = hist;
    buf32[4 + l] = nextc;
    nextc = stage_load(seg + 2);
  }

  if (TRACK) {
    // the lane's winner -> across the 16 lanes: value, then column of y, then row of x
    const uint32_t kb = __float_as_uint(blk);
    float bv = __uint_as_float(kb & ~31u) * sa.unscale;
    long long bj = (long long)l * R + (31 - (int)(kb & 31u)) + 1, bi = P.b_offset + tl + 1;
    if (!(bv > 0.0f)) { bv = 0.0f; bi = 0; bj = 0; }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 16);
      const long long oi = __shfl_xor(bi, off, 16);
      const long long oj = __shfl_xor(bj, off, 16);
      if (ov > bv || (ov == bv && ov > 0.0f && (oj < bj || (oj == bj && oi < bi)))) { bv = ov; bi = oi; bj = oj; }
    }
    if (l == 0 && active) {
      *P.best = bv;
      P.cell[0] = bv > 0.0f ? bi : 0;
      P.cell[1] = bv > 0.0f ? bj : 0;
    }
  }
}

// sw_wave_prof16_kernel — the TRACK pass of sw_wave_prof_kernel on PACKED FLOAT16 cells: every register holds the same cell of
// TWO database sequences (low half = problem 2s, high half = problem 2s + 1 of the launch; neighbours in a length-sorted
// list), so a 16-lane slot runs two problems and a cell costs half the instructions.
//   * cells hold H / (q 2048), q = the power of two all three scores are multiples of: every value below 2048 q is exact, and the
//     host takes this kernel only when match * (|y| + 1) stays below that (DESIGN.md §3.3 lemma L13);
//   * the two halves see different stream letters, so the profile exists twice in LDS: profLo[code][lane][r] = (score | 1.0 in
//     the high half) and profHi[code][lane][r] = (1.0 | score in the high half).  ONE v_pk_fma_f16 with the clamp modifier,
//     profLo[cA] * profHi[cB] + NW, is then (clamp(NW_A + score_A), clamp(NW_B + score_B)): no merge instruction, and the cell is
//     fma, maximum3, add (H - g), or (the key), half a maximum3 (the lane's best key) = 4.5 ops for two cells;
//   * the first maximum in storage order: key = cell bits | (15 - column within the lane) as in sw_wave_prof_kernel — the four
//     lowest mantissa bits of a float16 are free while the value is below 128 q (cell < 2^-4).  A slot whose best key reaches
//     2^-4 is NOT decided here: best = -1, and the host hands that sequence to the float32 path (a database of mostly unrelated
//     sequences: a handful).  The row of the first sight of a lane's best key is kept for both halves as two 16-bit counts of
//     the steps since (the host bounds the stream at 65 000 rows): max, sub, saturating sub, mad on packed integers per STEP.
// Saved states: the packed registers as they are, one row per state for the slot's PAIR of problems, at the first problem's
// ckpt (rows for both problems were laid out back to back: room for the longer of the two); the decision pass converts the half
// it resumes (WaveProblem::ck_half).
struct WaveProf16Args {
  const uint8_t *lut;        // [256] byte -> code; ncodes - 1 = "other"
  const uint8_t *byte_of;    // [ncodes - 1] code -> byte
  int32_t ncodes;
  uint32_t match_h, mismatch_h;   // float16 bits of score / (q 2048)
  uint32_t ngap2;            // float16 bits of -gap / (q 2048) in both halves
  float unscale;             // q * 2048
};
constexpr uint32_t kProf16KeyLimit = 0x2C00u;                      // float16 2^-4 = 128 / 2048: keys below it are exact
constexpr uint32_t kProf16One = 0x3C00u;                           // float16 1.0
constexpr uint32_t kProf16Pad = 0xC800u;                           // float16 -8: padding columns clamp to 0

template <int R>
__global__ __launch_bounds__(256) void sw_wave_prof16_kernel(const WaveProblem *probs, int nprob, const WaveProf16Args sa) {
  static_assert(R <= 16, "the key holds the column within the lane in four bits");
  constexpr int LS = lane_stride(R);
  constexpr int NQ4 = (R + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) uint32_t wsmem[];
  __shared__ __attribute__((aligned(16))) uint8_t win[2 * 16 * kWaveBuf];
  __shared__ uint8_t lut_s[256];
  uint32_t *profLo = wsmem;                                        // [ncodes][16][LS]
  uint32_t *profHi = wsmem + sa.ncodes * 16 * LS;
  const int tid = threadIdx.x;
  const int l = tid & 15;
  const int slot = tid >> 4;
  const int pidA = (blockIdx.x * 16 + slot) * 2, pidB = pidA + 1;
  const bool activeA = pidA < nprob, activeB = pidB < nprob;
  const uint8_t *bA = nullptr, *bB = nullptr;
  int nbA = 0, nbB = 0;
  if (activeA) { bA = probs[pidA].b; nbA = probs[pidA].nb; }
  if (activeB) { bB = probs[pidB].b; nbB = probs[pidB].nb; }
  // the lane side is the same for every problem of the launch (the range of the resident reference)
  const uint8_t *ya = probs[0].a;
  const int na = probs[0].na;
  const uint32_t other = (uint32_t)(sa.ncodes - 1);
  lut_s[tid] = sa.lut[tid];
  for (int e = tid; e < sa.ncodes * 16 * R; e += 256) {
    const int c = e / (16 * R);
    const int rem = e - c * 16 * R;
    const int ll = rem / R, r = rem - ll * R;
    const int j = ll * R + r;
    uint32_t v = kProf16Pad;
    if (j < na) v = ((uint32_t)c < other && ya[j] == sa.byte_of[c]) ? sa.match_h : sa.mismatch_h;
    profLo[(c * 16 + ll) * LS + r] = v | (kProf16One << 16);
    profHi[(c * 16 + ll) * LS + r] = kProf16One | (v << 16);
  }
  __syncthreads();

  // stream windows of CODES, one per half: 16 B history + 64 B segment, refilled every 64 steps
  uint8_t *bufA = win + (2 * slot) * kWaveBuf, *bufB = bufA + kWaveBuf;
  uint32_t *bufA32 = reinterpret_cast<uint32_t *>(bufA), *bufB32 = reinterpret_cast<uint32_t *>(bufB);
  const uint8_t *bufA_lane = bufA + 16 - l, *bufB_lane = bufB + 16 - l;
  auto stage_load = [&](const uint8_t *b, int nb, int seg) -> uint32_t {
    const int c0 = seg * kWaveSeg + 4 * l;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = c0 + k;
      const uint32_t ch = (uint32_t)t < (uint32_t)nb ? (uint32_t)lut_s[b[t]] : other;     // outside the stream: matches nothing
      w |= ch << (8 * k);
    }
    return w;
  };
  int steps = max(nbA, nbB) + 16;
  steps = max(steps, __shfl_xor(steps, 16));
  steps = max(steps, __shfl_xor(steps, 32));
  const int nseg = (steps + kWaveSeg - 1) / kWaveSeg;
  const int steps4 = (steps + 3) & ~3;

  uint32_t nextA = stage_load(bA, nbA, 0), nextB = stage_load(bB, nbB, 0);
  if (l < 4) { bufA32[l] = other * 0x01010101u; bufB32[l] = other * 0x01010101u; }
  bufA32[4 + l] = nextA; bufB32[4 + l] = nextB;
  nextA = stage_load(bA, nbA, 1); nextB = stage_load(bB, nbB, 1);

  uint32_t gv = sa.ngap2;
  asm volatile("" : "+v"(gv));
  uint32_t one2 = 0x00010001u;
  asm volatile("" : "+v"(one2));
  uint32_t H[R], Hg[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { H[r] = 0u; Hg[r] = gv; }
  uint32_t up_prev = 0;
  uint32_t blk = 0;                                                // best key of this lane, both halves
  uint32_t s2 = 0;                                                 // ... and the steps since it was first seen (16 bits each)
  const uint32_t *profLo_lane = profLo + l * LS, *profHi_lane = profHi + l * LS;
  float *ckA = activeA ? probs[pidA].ckpt : nullptr;              // (the pair's states go where its first problem's would)

  // the slot's wavefront as it stands after step kCkptEvery (c + 1) - 1: the packed registers as they are, for the pair
  const int nbmax = max(nbA, nbB);
  auto save_state = [&](int c) {
    if (ckA != nullptr && (c + 1) * kCkptEvery < nbmax + 16) {
      uint32_t *ck = reinterpret_cast<uint32_t *>(ckA) + ((size_t)c * 16 + (size_t)l) * (R + 1);
#pragma unroll
      for (int r = 0; r < R; ++r) ck[r] = H[r];
      ck[R] = up_prev;
    }
  };
  for (int seg = 0; seg < nseg; ++seg) {
    const int kq = min(kWaveSeg, steps4 - seg * kWaveSeg) >> 2;
    for (int k4 = 0; k4 < kq; ++k4) {
    if (k4 != 0 && (4 * k4) % kCkptEvery == 0) save_state(kCkptPerSeg * seg + 4 * k4 / kCkptEvery - 1);
#pragma unroll
    for (int ku = 0; ku < 4; ++ku) {
      const int k = 4 * k4 + ku;
      const uint32_t cA = (uint32_t)bufA_lane[k], cB = (uint32_t)bufB_lane[k];
      const u32x4 *pa = static_cast<const u32x4 *>(__builtin_assume_aligned(profLo_lane + cA * (16 * LS), 16));
      const u32x4 *pb = static_cast<const u32x4 *>(__builtin_assume_aligned(profHi_lane + cB * (16 * LS), 16));
      uint32_t A[NQ4 * 4], B[NQ4 * 4];
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const u32x4 v = pa[q];
        A[4 * q + 0] = v.x; A[4 * q + 1] = v.y; A[4 * q + 2] = v.z; A[4 * q + 3] = v.w;
        const u32x4 u = pb[q];
        B[4 * q + 0] = u.x; B[4 * q + 1] = u.y; B[4 * q + 2] = u.z; B[4 * q + 3] = u.w;
      }
      const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H[R - 1], 0x111, 0xf, 0xf, true);
      uint32_t diag = up_prev;
      up_prev = up;
      uint32_t ng;
      asm("v_pk_add_f16 %0, %1, %2" : "=v"(ng) : "v"(up), "v"(gv));
      uint32_t m = 0, tpend = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t w = H[r];
        uint32_t x, h;
        asm("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(x) : "v"(A[r]), "v"(B[r]), "v"(diag));
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(h) : "v"(x), "v"(Hg[r]), "v"(ng));
        const uint32_t hk = h | ((uint32_t)(15 - r) * 0x00010001u);
        if (r & 1) asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(tpend), "v"(hk));
        else if (r + 1 < R) tpend = hk;
        else asm("v_pk_max_f16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(hk));
        diag = w;
        H[r] = h;
        asm("v_pk_add_f16 %0, %1, %2" : "=v"(ng) : "v"(h), "v"(gv));
        Hg[r] = ng;
      }
      {
        // strict '>' per half: a key that grew restarts the count of steps since the best key was first seen, an equal key
        // (same value, same column, a later row) does not: s2 <- (s2 + 1) * [key did not grow]
        uint32_t nbk, d, nd1;
        asm("v_pk_max_f16 %0, %1, %2" : "=v"(nbk) : "v"(blk), "v"(m));
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(nbk), "v"(blk));
        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(nd1) : "v"(one2), "v"(d));
        asm("v_pk_mad_u16 %0, %1, %2, %2" : "=v"(s2) : "v"(s2), "v"(nd1));
        blk = nbk;
      }
    }
    }
    if (kq == kWaveSeg / 4) save_state(kCkptPerSeg * seg + kCkptPerSeg - 1);
    const uint32_t histA = bufA32[kWaveSeg / 4 + (l & 3)], histB = bufB32[kWaveSeg / 4 + (l & 3)];
    if (l < 4) { bufA32[l] = histA; bufB32[l] = histB; }
    bufA32[4 + l] = nextA; bufB32[4 + l] = nextB;
    nextA = stage_load(bA, nbA, seg + 2); nextB = stage_load(bB, nbB, seg + 2);
  }

  // per half: the lane's winner -> across the 16 lanes: value, then column of y, then row of x
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const bool active = half ? activeB : activeA;
    const uint32_t kb = (blk >> (16 * half)) & 0xFFFFu;
    const int tl = steps4 - 1 - l - (int)((s2 >> (16 * half)) & 0xFFFFu);   // (the lane's last stream position - steps since)
    uint32_t kmax = kb;                                            // (keys are positive float16: ordered as integers)
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 16));
    float bv = (float)__builtin_bit_cast(_Float16, (uint16_t)(kb & ~15u)) * sa.unscale;
    const long long boff = active ? probs[half ? pidB : pidA].b_offset : 0;
    long long bj = (long long)l * R + (15 - (int)(kb & 15u)) + 1, bi = boff + tl + 1;
    if (!(bv > 0.0f)) { bv = 0.0f; bi = 0; bj = 0; }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 16);
      const long long oi = __shfl_xor(bi, off, 16);
      const long long oj = __shfl_xor(bj, off, 16);
      if (ov > bv || (ov == bv && ov > 0.0f && (oj < bj || (oj == bj && oi < bi)))) { bv = ov; bi = oi; bj = oj; }
    }
    if (l == 0 && active) {
      const WaveProblem *P = probs + (half ? pidB : pidA);
      const bool undecided = kmax >= kProf16KeyLimit;
      *P->best = undecided ? -1.0f : bv;
      P->cell[0] = (bv > 0.0f && !undecided) ? bi : 0;
      P->cell[1] = (bv > 0.0f && !undecided) ? bj : 0;
    }
  }
}

// Greedy walk over sw_wave_kernel's decisions (smithwaterman.cpp:40-78).
struct WaveWalk {
  const uint8_t *x;        // full x
  const uint8_t *y;        // full y (of the range)
  const uint32_t *dirs;    // [nb][lanes][W] as WaveProblem::dirs (lanes = 16) or StripProblem::dirs (64 per wavefront)
  int32_t na, nb, orient;
  int32_t R;               // rows per lane of the instance that wrote dirs
  int32_t lanes;           // lanes that share one stream position in dirs
  int32_t skew;            // 1: the decisions of (stream position t, lane) are in row t + lane - row0 (sw_wave_kernel); 0: in row t
  int32_t row0;            // skew = 1: the step the decision rows start at (a resumed problem, sw_wave_prof_kernel); a cell in
                           // front of it is outside the window (status 1)
  float need_slope;        // > 0: a cell with lane-side index a is already exact a + ceil(a * need_slope) + 2 stream
                           // positions into the window (DESIGN.md §3.3 for a path confined to a rows/columns)
  int64_t b_offset;        // as WaveProblem
  int64_t start_i, start_j;  // true 1-based start cell (row of x, column of y)
  int64_t exact_from;      // stream index (true, 1-based) from which every cell is exact; 0 = all
  int32_t cap;             // longest consensus the caller accepts
  int64_t *out;            // [0] length, [1] pos, [2] status (0 ok, 1 window too small, 2 capacity)
  // Decisions made from the score sweep's saved columns / rows (host_saved.h): a cell is as exact as the sweep's TILE that
  // owns its column made it — tile T = (column - 1 + 63) / zchunk started from a zero border at column T * zchunk - zwarm
  // (tile 0: the matrix border itself).  zchunk > 0 replaces b_offset by that border in the need_slope rule; exact_from = 0.
  int64_t zchunk, zwarm;
};

// Many walks take two passes over the same decisions: kWalkMeasure yields (length, pos, status); the host then
// lays the strings out back to back and kWalkWrite emits them at cons + offs[p] (x) and cons + offs[p] + length
// (y), so that only the bytes that exist cross PCIe.  A few walks take one pass (kWalkBoth): x at cons + offs[p],
// y at cons + offs[p] + cap.
enum : int { kWalkMeasure = 0, kWalkWrite = 1, kWalkBoth = 2 };
template <int MODE>
__global__ void sw_wave_walk_kernel(const WaveWalk *probs, int n, char *cons, const int64_t *offs) {
  constexpr bool WRITE = MODE != kWalkMeasure;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const WaveWalk W = probs[p];
  char *cons_x = nullptr, *cons_y = nullptr;
  if (MODE == kWalkWrite) {
    if (W.out[2] != 0) return;
    cons_x = cons + offs[p];
    cons_y = cons_x + W.out[0];
  }
  if (MODE == kWalkBoth) {
    cons_x = cons + offs[p];
    cons_y = cons_x + W.cap;
  }
  long long ix = W.start_i, iy = W.start_j;
  int len = 0;
  long long status = 0, pos = 0;
  if (ix <= 0 || iy <= 0) { W.out[0] = 0; W.out[1] = 0; W.out[2] = 0; return; }
  for (;;) {
    const long long sidx = W.orient == 0 ? iy : ix;               // stream-side true index of this cell
    const long long aidx = W.orient == 0 ? ix : iy;               // lane-side true index (1-based)
    const long long t = sidx - W.b_offset - 1;
    // the decision at (ix, iy) reads cells one step back along the stream: they must be exact and inside the window
    if (t < 0 || t >= W.nb || aidx > W.na) { status = 1; break; }
    if (W.exact_from > 0) {
      long long need = W.exact_from;
      if (W.need_slope > 0.0f) {
        const long long rn = W.b_offset + aidx + (long long)ceilf((float)aidx * W.need_slope) + 2;
        need = rn < need ? rn : need;
      }
      if (sidx - 1 < need) { status = 1; break; }
    }
    if (len >= W.cap) { status = 2; break; }
    const int lane = (int)((aidx - 1) / W.R), r = (int)((aidx - 1) % W.R);
    const int wd = (W.R + 15) / 16;
    const long long row = W.skew ? t + lane - W.row0 : t;
    if (row < 0) { status = 1; break; }                            // in front of a resumed problem's first step
    const int dir = (int)((W.dirs[((size_t)row * (size_t)W.lanes + (size_t)lane) * wd + (r >> 4)] >> (2 * (r & 15))) & 3u);
    if (dir == kDirStop) {
      if (WRITE) { cons_x[len] = (char)W.x[ix - 1]; cons_y[len] = (char)W.y[iy - 1]; }
      ++len;
      pos = iy;
      break;
    } else if (dir == kDirNW) {
      if (WRITE) { cons_x[len] = (char)W.x[ix - 1]; cons_y[len] = (char)W.y[iy - 1]; }
      ++len; --ix; --iy;
    } else if (dir == kDirW) {
      if (WRITE) { cons_x[len] = '-'; cons_y[len] = (char)W.y[iy - 1]; }
      ++len; --iy;
    } else {
      if (WRITE) { cons_x[len] = (char)W.x[ix - 1]; cons_y[len] = '-'; }
      ++len; --ix;
    }
  }
  if (MODE == kWalkWrite) return;
  W.out[0] = len; W.out[1] = pos; W.out[2] = status;
}

// The same walk for a FEW LONG alignments (one 10 kbp query: ~10 000 dependent decision reads, each a cache miss, at
// ~1.3 us per step on one thread): one wavefront per walk.  Lane p reads the decision p steps ahead ALONG THE NW
// DIAGONAL (where a walk spends ~99 % of its steps); the leading run of NW decisions is consumed at once, the first other
// decision (W, N, stop, or a cell outside the exact zone) is applied, and the lanes look ahead again from there.
// Results as sw_wave_walk_kernel<kWalkBoth>.  ORIENT 0 only (lanes of the decision kernel = rows of x).
__global__ __launch_bounds__(64) void sw_wave_walk_long_kernel(const WaveWalk *probs, int n, char *cons, const int64_t *offs) {
  const int w = blockIdx.x;
  if (w >= n) return;
  const WaveWalk W = probs[w];
  const int p = threadIdx.x;
  char *cons_x = cons + offs[w];
  char *cons_y = cons_x + W.cap;
  long long ix = W.start_i, iy = W.start_j;
  long long len = 0, pos = 0, status = 0;
  if (ix <= 0 || iy <= 0) { if (p == 0) { W.out[0] = 0; W.out[1] = 0; W.out[2] = 0; } return; }
  const int wd = (W.R + 15) / 16;
  for (;;) {
    const long long cx = ix - p, cy = iy - p;                        // this lane's cell, p NW steps ahead
    int code;                                                          // 0 stop, 1 NW, 2 W, 3 N, 4 window, 5 capacity, 6 beyond the border
    if (cx < 1 || cy < 1) code = 6;
    else {
      const long long t = cy - W.b_offset - 1;
      bool inwin = !(t < 0 || t >= W.nb || cx > W.na);
      if (inwin && W.exact_from > 0) {
        long long need = W.exact_from;
        if (W.need_slope > 0.0f) {
          const long long rn = W.b_offset + cx + (long long)ceilf((float)cx * W.need_slope) + 2;
          need = rn < need ? rn : need;
        }
        if (cy - 1 < need) inwin = false;
      }
      if (inwin && W.zchunk > 0) {
        const long long tile = (cy - 1 + 63) / W.zchunk;
        if (tile > 0 && cy - 1 < tile * W.zchunk - W.zwarm + cx + (long long)ceilf((float)cx * W.need_slope) + 2) inwin = false;
      }
      if (!inwin) code = 4;
      else if (len + p >= W.cap) code = 5;
      else {
        const int lane = (int)((cx - 1) / W.R), r = (int)((cx - 1) % W.R);
        const long long row = W.skew ? t + lane - W.row0 : t;
        code = row < 0 ? 4 : (int)((W.dirs[((size_t)row * (size_t)W.lanes + (size_t)lane) * wd + (r >> 4)] >> (2 * (r & 15))) & 3u);
      }
    }
    const unsigned long long other = __ballot(code != kDirNW);
    const int run = other ? __builtin_ctzll(other) : 64;
    if (p < run) { cons_x[len + p] = (char)W.x[cx - 1]; cons_y[len + p] = (char)W.y[cy - 1]; }
    if (run == 64) { ix -= 64; iy -= 64; len += 64; continue; }
    const int ev = __shfl(code, run);
    const long long ex = ix - run, ey = iy - run, elen = len + run;
    if (ev == kDirStop) {
      if (p == 0) { cons_x[elen] = (char)W.x[ex - 1]; cons_y[elen] = (char)W.y[ey - 1]; }
      len = elen + 1; pos = ey;
      break;
    } else if (ev == kDirW) {
      if (p == 0) { cons_x[elen] = '-'; cons_y[elen] = (char)W.y[ey - 1]; }
      len = elen + 1; ix = ex; iy = ey - 1;
    } else if (ev == kDirN) {
      if (p == 0) { cons_x[elen] = (char)W.x[ex - 1]; cons_y[elen] = '-'; }
      len = elen + 1; ix = ex - 1; iy = ey;
    } else {
      len = elen;
      status = ev == 5 ? 2 : 1;
      break;
    }
  }
  if (p == 0) { W.out[0] = len; W.out[1] = pos; W.out[2] = status; }
}

}  // namespace mi355sw
