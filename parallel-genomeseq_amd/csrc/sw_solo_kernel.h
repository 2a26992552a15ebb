// sw_solo_kernel.h — ONE short query against a long reference, everything after the score pass in one kernel.
//
// The one-by-one loop of the reference's drivers (a new SWAligner per read, src/sw_solve_big.cpp:78-92,
// src/sw_solve_small.cpp:82-93) is latency-bound here: after sw_score_kernel has left (maximum, first sub-chunk) in the
// query's key, the general pipeline runs three more kernels with a host round trip after each (locate, traceback
// decisions, walk).  This kernel chains them on the device: it reads the key itself, derives the candidate window(s)
// exactly as host_pipeline.h locate_fast does, sweeps each window ONCE on a register wavefront (64 lanes x R rows) —
// tracking the first cell equal to the maximum in the engine's storage order (find_index_of_maximum,
// similaritymatrix.cpp:21-28 / :291-299) and keeping every cell's greedy decision (smithwaterman.cpp:51-72) in LDS —
// and lets the workgroup that holds the winning cell walk the traceback out of LDS (smithwaterman.cpp:40-78).
// The host launches it right behind the score kernel and synchronises once.
//
//   float engine: one candidate window (the first sub-chunk that reached the maximum holds the first maximum in
//                 column-major order), one workgroup;
//   uint8 engine: up to five candidate windows (first, first + 1, and the corner triangles 0, last - 1, last: the
//                 skewed storage order visits them early), one workgroup each; the smallest storage-order key wins
//                 (64-bit atomicMin + arrival counter; all five are resident at once, the wait is bounded).
// Windows that do not fit LDS, walks that leave the exact zone and anything else unusual raise a status and the host
// falls back to the general path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_exact_kernel.h"  // order_key, kDir*
#include "sw_wave_kernel.h"   // WaveScoring

namespace mi355sw {

enum : int { kSoloOk = 0, kSoloWindow = 1, kSoloCapacity = 2, kSoloLds = 3, kSoloExpired = 4, kSoloLost = 5 };

struct SoloResult {            // header of the result block (device; zeroed by the call's upload); cons_x at +64, cons_y at +64 + cap
  float score;
  int32_t status;              // kSolo*
  int64_t ix, iy;              // argmax row, argmax column (1-based, relative to the range)
  int64_t len, pos;            // consensus length, position where the walk stopped (relative to the range)
  int32_t written;             // 1 when the block was filled by this launch
  int32_t piece;               // which range held the maximum (0 for a single range)
  int32_t pad[4];
};
static_assert(sizeof(SoloResult) == 64, "the strings start at +64");

struct SoloArgs {
  const unsigned long long *key;   // [nranges] the query's keys of this call's score pass, one per range
  const int64_t *range_lo;         // [nranges] the ranges (pieces of the reference, plocalaligner.cpp:44-67) the score pass swept;
  const int64_t *range_hi;         //           the FIRST one with the strictly greatest maximum is located and traced (:122-129)
  int32_t nranges;
  unsigned long long *gmin;        // smallest storage-order key over the workgroups (~0 at launch)
  unsigned int *done;              // workgroups that have finished their sweep (0 at launch)
  const uint8_t *x;                // query bytes (device)
  int32_t m;
  const uint8_t *yref;             // first byte of the resident reference (ranges index it)
  int64_t sub_len;                 // granularity of the key's tag
  int32_t keykind;                 // 2: float16 bits of H / 2048; 4: float32 bits of H * 2^-fshift
  int32_t fshift;
  float mg_smax, mg_g;             // margins (host_common.h Margin): gap columns per row <= mg_smax / mg_g
  int64_t warm;                    // the bucket's general margin
  int32_t budget;                  // room for the walk's horizontal excursions
  int32_t want_trace;
  int32_t ncand;                   // workgroups launched
  int32_t cap;                     // capacity of each consensus string
  int32_t lds_steps;               // stream positions whose decisions fit LDS
  WaveScoring sc;
  SoloResult *out;
};

constexpr int kSoloSpinLimit = 1 << 22;

// max / min without the compiler's NaN canonicalisation moves (no value here is ever a NaN): one instruction each
__device__ __forceinline__ float solo_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float solo_max3_0(float a, float b) { float r; asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float solo_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int R, bool U8>
__global__ __launch_bounds__(64) void sw_solo_kernel(const SoloArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int DB = R <= 4 ? 1 : 2;                   // bytes of decisions per lane and stream position (2 bits per row)
  typedef typename std::conditional<DB == 1, uint8_t, uint16_t>::type dir_t;
  const int l = threadIdx.x;
  const int c = blockIdx.x;
  char *cons_out = reinterpret_cast<char *>(a.out) + sizeof(SoloResult);

  // ---- the key(s) of the score pass: first range with the strictly greatest maximum -----------------------------------
  int piece = 0;
  unsigned long long key = __hip_atomic_load(a.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int pc = 1; pc < a.nranges; ++pc) {
    const unsigned long long k2 = __hip_atomic_load(a.key + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((k2 >> 32) > (key >> 32)) { key = k2; piece = pc; }        // non-negative scores order like their bits
  }
  const uint8_t *ybase = a.yref + a.range_lo[piece];
  const int64_t n = a.range_hi[piece] - a.range_lo[piece];
  float score;
  {
    const uint32_t hi = (uint32_t)(key >> 32);
    if (a.keykind == 2) score = (float)__builtin_bit_cast(_Float16, (unsigned short)hi) * 2048.0f;
    else score = ldexpf(__uint_as_float(hi), a.fshift);
  }
  if (!(score > 0.0f)) {                                // all-zero matrix: the defined no-match result
    if (c == 0 && l == 0) {
      a.out->score = 0.0f; a.out->status = kSoloOk; a.out->ix = 0; a.out->iy = 0; a.out->len = 0; a.out->pos = 0;
      a.out->piece = piece;
      a.out->written = 1;
    }
    return;
  }
  const int64_t nsub = (n + a.sub_len - 1) / a.sub_len;
  const int64_t first = (int64_t)(0xFFFFFFFFull - (key & 0xFFFFFFFFull));
  // candidate sub-chunks, in the order of host_pipeline.h locate_fast; duplicates and out-of-range ones drop out
  int64_t mine = -1;
  {
    const int64_t list[5] = {first, first + 1, 0, nsub - 2, nsub - 1};
    const int nlist = U8 ? 5 : 1;
    if (c < nlist) {
      const int64_t v = list[c];
      bool ok = v >= 0 && v < nsub;
      for (int e = 0; e < c; ++e) ok = ok && list[e] != v;
      if (ok) mine = v;
    }
  }
  auto arrive = [&]() { if (l == 0) { __threadfence(); atomicAdd(a.done, 1u); } };
  if (mine < 0) { arrive(); return; }

  // ---- this workgroup's window: the sub-chunk, the lane lag, and the margins of locate and of the traceback -------
  const int m = a.m;
  const float slope = a.mg_smax / a.mg_g;
  const int64_t sub_lo = max((int64_t)0, mine * a.sub_len - 63);
  const int64_t sub_hi = min((mine + 1) * a.sub_len, n);
  const float spare = fmaxf(0.0f, a.mg_smax * (float)m - score);
  // (one column more than the host's double-precision forms of the same margins: these are evaluated in float)
  const int64_t warm1 = min(a.warm, (int64_t)m + (int64_t)ceilf(spare / a.mg_g) + 3);
  const int64_t need_t = a.want_trace ? (int64_t)a.budget + min(a.warm, (int64_t)m + (int64_t)ceilf((float)m * slope) + 3) : 0;
  const int64_t wl = max((int64_t)0, sub_lo - max(warm1, need_t));
  const int nb = (int)(sub_hi - wl);
  const int own_lo = (int)(sub_lo - wl);
  if (nb > a.lds_steps) {
    if (l == 0) atomicMax(reinterpret_cast<unsigned int *>(&a.out->status), (unsigned int)kSoloLds);
    arrive();
    return;
  }
  // LDS: decisions [lds_steps + 1][64] (the last row takes the stores of positions outside the window), the window's
  // bytes with 64 bytes of padding in front (lane l is l positions behind lane 0) and behind, the query, the strings
  dir_t *D = reinterpret_cast<dir_t *>(lds);
  uint8_t *winp = lds + ((size_t)a.lds_steps + 1) * 64 * DB;           // [64 + lds_steps + 64 + 8]
  uint8_t *win = winp + 64;
  uint8_t *xs = winp + (((size_t)a.lds_steps + 136 + 15) & ~(size_t)15); // [64 * R]
  char *cx = reinterpret_cast<char *>(xs + 64 * R);                    // [cap], then cy [cap]
  char *cy = cx + a.cap;
  for (int t = l - 64; t < nb + 72; t += 64) win[t] = (t >= 0 && t < nb) ? ybase[wl + t] : 0;
  for (int e = l; e < 64 * R; e += 64) xs[e] = e < m ? a.x[e] : 0;
  __syncthreads();

  uint32_t ca[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const int ai = l * R + r; ca[r] = ai < m ? (uint32_t)xs[ai] : 0xFFFFu; }
  float H[R];
#pragma unroll
  for (int r = 0; r < R; ++r) H[r] = 0.0f;
  uint32_t up_prev = 0;
  const float gpen = U8 ? a.sc.u8G : a.sc.gap;
  unsigned long long bkey = ~0ull;
  long long bi = 0, bj = 0;
  // per-row decision codes, already shifted to the row's two bits: no shifts in the loop
  uint32_t dNW[R], dW[R], dN[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { dNW[r] = (uint32_t)kDirNW << (2 * r); dW[r] = (uint32_t)kDirW << (2 * r); dN[r] = (uint32_t)kDirN << (2 * r); }

  const int steps = nb + 63;                                           // lane 63 reaches stream position nb - 1
  const int groups = (steps + 3) / 4;
  const uint8_t *wlane = win - l;                                      // + s = byte of this lane's position at step s
  uint32_t w4;
  __builtin_memcpy(&w4, wlane, 4);                                     // the bytes of steps 0..3 (unaligned LDS read)
  for (int g4 = 0; g4 < groups; ++g4) {
    uint32_t w4next;
    __builtin_memcpy(&w4next, wlane + 4 * (g4 + 1), 4);                // one group ahead: the read's latency is hidden
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int s = 4 * g4 + k;
      const int t = s - l;
      const bool in = (uint32_t)t < (uint32_t)nb;
      const uint32_t cb = in ? ((w4 >> (8 * k)) & 0xFFu) : 0x100u;     // outside the window nothing matches
      const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(H[R - 1]), 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
      float diag = __uint_as_float(up_prev);
      float north = __uint_as_float(up);
      up_prev = up;
      uint32_t dpack = 0;
      bool hit = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float wv = H[r];
        const bool eq = ca[r] == cb;
        float xv;
        if (U8) xv = eq ? fminf(diag + a.sc.u8M, 255.0f) : diag - a.sc.u8X;      // (the floor at 0 is the max3 below)
        else xv = diag + (eq ? a.sc.match : a.sc.mismatch);
        const float tmx = solo_max(wv, north);
        const float h = solo_max3_0(xv, tmx - gpen);
        // smithwaterman.cpp:51,59,66,72 at this cell (n1 = NW = diag, n2 = W = wv, n3 = N = north): stop when a neighbour
        // is 0, else NW if it is >= both others, else W if it is >= N, else N
        const float lowest = solo_min3(diag, wv, north);
        uint32_t d = wv >= north ? dW[r] : dN[r];
        d = diag >= tmx ? dNW[r] : d;
        d = lowest != 0.0f ? d : 0u;
        dpack |= d;
        hit |= h == score;
        diag = wv;
        H[r] = h;
        north = h;
      }
      D[(size_t)(in ? t : a.lds_steps) * 64 + l] = (dir_t)dpack;       // outside the window: the spare row
      if (hit && in && t >= own_lo) {                                  // rare: which rows, where in the storage order
        const long long j = wl + t + 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const long long i = (long long)l * R + r + 1;
          if (H[r] == score && i <= m) {
            const unsigned long long k2 = U8 ? order_key<1>(i, j, m, n) : order_key<0>(i, j, m, n);
            if (k2 < bkey) { bkey = k2; bi = i; bj = j; }
          }
        }
      }
    }
    w4 = w4next;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long ok = __shfl_xor(bkey, off);
    const long long oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
    if (ok < bkey) { bkey = ok; bi = oi; bj = oj; }
  }
  __syncthreads();                                                     // decisions of every lane are in LDS

  // ---- which workgroup holds the first maximum ---------------------------------------------------------------------
  if (l != 0) return;                                                  // the rest is one thread's work
  if (bkey != ~0ull) atomicMin(a.gmin, bkey);
  __threadfence();
  atomicAdd(a.done, 1u);
  if (a.ncand > 1) {
    int spins = 0;
    while (__hip_atomic_load(a.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)a.ncand) {
      if (++spins > kSoloSpinLimit) {
        atomicMax(reinterpret_cast<unsigned int *>(&a.out->status), (unsigned int)kSoloExpired);
        return;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  const unsigned long long g = __hip_atomic_load(a.gmin, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
  if (g == ~0ull) {                                                    // nobody found the maximum again
    if (c == 0) { a.out->status = kSoloLost; a.out->piece = piece; a.out->written = 1; }
    return;
  }
  if (bkey != g) return;

  // ---- the winner: result, and the greedy walk over the decisions in LDS (smithwaterman.cpp:40-78) ------------------
  long long ix = bi, iy = bj;
  int len = 0;
  int status = kSoloOk;
  long long pos = 0;
  if (a.want_trace) {
    const long long exact_from = wl == 0 ? 0 : wl + a.warm;
    for (;;) {
      const long long t = iy - wl - 1;
      if (t < 0 || t >= nb || ix > m || ix < 1) { status = kSoloWindow; break; }
      if (exact_from > 0) {
        long long need = exact_from;
        const long long rn = wl + ix + (long long)ceilf((float)ix * slope) + 2;
        need = rn < need ? rn : need;
        if (iy - 1 < need) { status = kSoloWindow; break; }
      }
      if (len >= a.cap) { status = kSoloCapacity; break; }
      const int lane = (int)((ix - 1) / R), r = (int)((ix - 1) % R);
      const int dir = ((uint32_t)D[(size_t)t * 64 + lane] >> (2 * r)) & 3;
      if (dir == kDirStop) { cx[len] = (char)xs[ix - 1]; cy[len] = (char)win[t]; ++len; pos = iy; break; }
      else if (dir == kDirNW) { cx[len] = (char)xs[ix - 1]; cy[len] = (char)win[t]; ++len; --ix; --iy; }
      else if (dir == kDirW) { cx[len] = '-'; cy[len] = (char)win[t]; ++len; --iy; }
      else { cx[len] = (char)xs[ix - 1]; cy[len] = '-'; ++len; --ix; }
    }
    if (status == kSoloOk) {
      for (int k = 0; k < len; ++k) { cons_out[k] = cx[k]; cons_out[a.cap + k] = cy[k]; }
    }
  }
  a.out->score = score;
  a.out->piece = piece;
  a.out->ix = bi; a.out->iy = bj;
  a.out->len = len; a.out->pos = pos;
  atomicMax(reinterpret_cast<unsigned int *>(&a.out->status), (unsigned int)status);
  a.out->written = 1;
}

}  // namespace mi355sw
