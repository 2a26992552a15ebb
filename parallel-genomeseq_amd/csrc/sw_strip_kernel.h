// sw_strip_kernel.h — traceback decisions for ONE LONG query: the wavefronts of a workgroup form a pipeline.
//
// The register wavefront of sw_wave_kernel.h (ORIENT 0: lanes hold rows of x, the stream runs over a window of
// y) with whole-wavefront strips: wavefront w of the workgroup owns rows [w*64*R, (w+1)*64*R) of x — lane l its
// R consecutive rows — and sweeps the window one column per step.  The bottom row of strip w enters strip w+1
// through an LDS ring (lane 63 stores one value per step, lane 0 of the next wavefront picks it up through the
// DPP `old` operand, where the single-strip kernels get the zero border row), so the strips run concurrently,
// each two 64-column segments behind the one above it.  This is the multi-wavefront cooperative sweep of one
// alignment (what the reference's fine-grained OpenMP variants attempt, similaritymatrix.cpp:118-245), kept
// inside one workgroup so that all participants are resident by construction.
//
// Flow control: per strip a count of produced and of consumed boundary positions, published with release
// stores once per segment and polled with acquire loads; a strip waits for input (positions of the coming
// segment) and for ring space (the strip below must have consumed what is about to be overwritten).  The waits
// cannot form a cycle (a producer blocks only when >= kStripRing-64 positions ahead, a consumer only when < 128
// behind), every wavefront runs the same number of segments, and every wait is bounded: on expiry the workgroup
// raises `status` and drains.
//
// Output: one greedy traceback decision per cell (smithwaterman.cpp:51-72), 2 bits, laid out
// dirs[stream position][lane of the workgroup][W] dwords (W = 1 for R <= 16, else 2), the layout of
// sw_wave_kernel.h with 64*nw lanes instead of 16; sw_wave_walk_kernel reads both.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_wave_kernel.h"   // WaveScoring, kDir*

namespace mi355sw {

struct StripProblem {
  const uint8_t *a;      // x (rows)
  const uint8_t *b;      // window of y: stream position t is b[t]
  int32_t na, nb;
  int32_t nw;            // wavefronts that hold rows (ceil(na / (64*R))), <= blockDim.x / 64
  uint32_t *dirs;        // [nb][64*nw][W]
  int32_t *status;       // 0 = complete, 1 = a pipeline wait expired (result unusable)
};

constexpr int kStripRing = 512;          // boundary positions held per strip (8 segments)
constexpr int kStripMaxWaves = 16;
constexpr int kStripSpinLimit = 1 << 22; // polls (with s_sleep) before a wait is declared dead

template <int R, bool U8>
__global__ __launch_bounds__(64 * kStripMaxWaves) void sw_strip_kernel(const StripProblem *probs, const WaveScoring sc) {
  __shared__ float ring[kStripMaxWaves][kStripRing];
  __shared__ int produced[kStripMaxWaves], consumed[kStripMaxWaves + 1];
  __shared__ int dead;
  __shared__ __attribute__((aligned(16))) uint8_t win[kStripMaxWaves][128];
  const StripProblem P = probs[blockIdx.x];
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  if (tid < kStripMaxWaves) produced[tid] = 0;
  if (tid <= kStripMaxWaves) consumed[tid] = 0;
  if (tid == 0) dead = 0;
  __syncthreads();
  const int nw = P.nw;
  if (w >= nw) return;                                   // spare wavefronts of a launch shared with longer queries
  const int na = P.na, nb = P.nb;
  const int LT = 64 * nw;
  constexpr int W = (R + 15) / 16;

  uint32_t ca[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int ai = (w * 64 + l) * R + r;
    ca[r] = (ai < na) ? (uint32_t)P.a[ai] : 0xFFFFu;    // padding rows never match
  }

  // stream window of this wavefront: 64 B of history, then the current 64-column segment
  uint8_t *buf = win[w];
  const uint8_t *buf_lane = buf + 64 - l;                // + k = byte of stream position seg*64 + k - l
  auto stage_load = [&](int seg) -> uint32_t {
    const int t = seg * 64 + l;
    return (t < nb) ? (uint32_t)P.b[t] : 0u;
  };
  const int nseg = (nb + 64 + 63) / 64;                  // lane 63 reaches stream position nb - 1
  uint32_t nextc = stage_load(0);
  buf[l] = 0;
  buf[64 + l] = (uint8_t)nextc;
  nextc = stage_load(1);

  float H[R];
#pragma unroll
  for (int r = 0; r < R; ++r) H[r] = 0.0f;
  uint32_t up_prev = 0;
  const float gpen = U8 ? sc.u8G : sc.gap;
  const float *rin = ring[w > 0 ? w - 1 : 0];
  float *rout = ring[w];
  const bool has_in = w > 0, has_out = w + 1 < nw;
  bool ok = true;

  auto wait_for = [&](int *counter, int need) {
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

  for (int seg = 0; seg < nseg && ok; ++seg) {
    // input: boundary positions seg*64 .. seg*64+63 (the strip above finishes them during ITS segment seg+1)
    if (has_in) {
      const int need = (seg + 1) * 64 < nb ? (seg + 1) * 64 : nb;
      wait_for(&produced[w - 1], need);
    }
    // ring space: this segment stores positions <= seg*64, over the slots of positions <= seg*64 - kStripRing
    if (has_out) wait_for(&consumed[w + 1], seg * 64 - kStripRing + 64);
    if (!ok) break;
#pragma unroll 2
    for (int k = 0; k < 64; ++k) {
      const int t0 = seg * 64 + k;                                     // lane 0's stream position
      const int t = t0 - l;
      const uint32_t cb = (uint32_t)buf_lane[k] | ((uint32_t)t >= (uint32_t)nb ? 0x100u : 0u);
      const float bnd = has_in ? rin[t0 & (kStripRing - 1)] : 0.0f;    // H(first row of the strip - 1, column t0)
      const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(bnd), (int)__float_as_uint(H[R - 1]),
                                                                0x138 /*wave_shr:1*/, 0xf, 0xf, false);
      float diag = __uint_as_float(up_prev);
      float north = __uint_as_float(up);
      up_prev = up;
      uint32_t dpack[W];
#pragma unroll
      for (int d = 0; d < W; ++d) dpack[d] = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float wv = H[r];
        const bool eq = ca[r] == cb;
        float x;
        if (U8) x = eq ? fminf(diag + sc.u8M, 255.0f) : fmaxf(diag - sc.u8X, 0.0f);
        else x = diag + (eq ? sc.match : sc.mismatch);
        const float y = fmaxf(wv, north) - gpen;
        const float h = fmaxf(fmaxf(x, y), 0.0f);
        // smithwaterman.cpp:51,59,66,72 at this cell: n1 = NW, n2 = W, n3 = N
        int dir;
        if (diag == 0.0f || wv == 0.0f || north == 0.0f) dir = kDirStop;
        else if (diag >= wv && diag >= north) dir = kDirNW;
        else if (wv >= diag && wv >= north) dir = kDirW;
        else dir = kDirN;
        dpack[r >> 4] |= (uint32_t)dir << (2 * (r & 15));
        diag = wv;
        H[r] = h;
        north = h;
      }
      if (has_out && l == 63 && t >= 0) rout[t & (kStripRing - 1)] = H[R - 1];
      if (t >= 0 && t < nb) {
        uint32_t *dst = P.dirs + ((size_t)t * LT + (size_t)(w * 64 + l)) * W;
#pragma unroll
        for (int d = 0; d < W; ++d) dst[d] = dpack[d];
      }
    }
    // lane 63 has stored positions <= seg*64; this wavefront has read positions <= seg*64 + 63
    if (l == 0) {
      if (has_out) __hip_atomic_store(&produced[w], seg * 64 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&consumed[w], (seg + 1) * 64, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const uint8_t hist = buf[64 + l];
    buf[l] = hist;
    buf[64 + l] = (uint8_t)nextc;
    nextc = stage_load(seg + 2);
  }
  if (l == 0) {
    // whatever happens next, nobody may wait on this wavefront any more
    __hip_atomic_store(&produced[w], 0x7FFFFFFF, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&consumed[w], 0x7FFFFFFF, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!ok) *P.status = 1;
  }
}

}  // namespace mi355sw
