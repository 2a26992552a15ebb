// sw_score_kernel.h — score-only Smith-Waterman sweep for gfx950 (MI355X), hand-written HIP.
//
// Replaces, for the score pass, Similarity_Matrix::iterate / Similarity_Matrix_Skewed::iterate
// plus the value half of find_index_of_maximum (reference src/aligner/similaritymatrix.cpp:99-264,
// :386-561, :21-28, :291-299).  The argmax POSITION and the traceback are recovered afterwards by
// sw_wave_kernel.h / sw_strip_kernel.h / sw_exact_kernel.h on the one or two sub-chunks that hold the maximum.
//
// Layout:
//   * a tile = (query pair, reference chunk).  16 lanes of a wavefront (one DPP row) own one tile;
//     lane l owns R consecutive query rows, so a 16-lane slot covers 16*R rows.  A wavefront runs
//     4 tiles (4 chunks of the same query pair), a 256-thread workgroup 16.
//   * every VGPR holds TWO cells: low half = query A, high half = query B, same reference column.
//     All arithmetic is packed 16-bit: float16 (v_pk_add_f16 clamp / v_pk_maximum3_f16) where every value fits
//     +-2048 and for the uint8 engine, integer (v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_u16 clamp) beyond; one
//     float32 instance (one query per register) covers everything else.
//   * cells of one anti-diagonal live in the 16 lanes: at step t lane l is at column t - l.  The
//     only cross-lane traffic is ONE v_mov_b32_dpp row_shr:1 per step (last row of the lane above).
//   * the substitution scores come from a per-workgroup query profile in LDS,
//     prof[ref code][lane][row] (packed A|B), read with ds_read_b128; the reference chunk is
//     streamed through a small per-slot LDS byte window (coalesced global loads, once per 64 steps).
//   * float16 / float32 cells: H = max3(clamp(NW + s), W - g, N - g) with H - g kept per cell — add, maximum3, add,
//     and one maximum3 per two cells for the running maximum: 3.5 ops per cell.
//     Integer cells: add, max(W, N), sat-sub, max — the running maximum is folded into t = max(W, N), which the
//     recurrence needs anyway, on odd rows only (t_r covers cell (r, j-1) and cell (r-1, j)): 4.5 ops per cell.
//
// One op = one VOP3P instruction = 4 cycles per wave64 on a SIMD (profiles/r01_valu_instruction_rates*.txt);
// the kernel is VALU-issue-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355sw {

typedef short i16x2 __attribute__((ext_vector_type(2)));
// one 16-byte LDS read (ds_read_b128): a single vector load the compiler cannot split into ds_read2_b64 pairs — the
// profile's lane stride is laid out for b128 lane groups, and the split form ran into bank conflicts on 56 % of its LDS
// cycles (profiles/r02_pmc_wide_tiles.json)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int kSlotLanes = 16;         // lanes of a DPP row; a tile ("slot") takes 16 or 8 of them
constexpr int kSeg = 64;               // steps between two refills of the code window
// bytes of history kept in front of a segment (>= lanes per slot - 1), and the per-slot window size
__host__ __device__ constexpr int hist_bytes(int SL) { return SL > 16 ? SL : 16; }
__host__ __device__ constexpr int codebuf_bytes(int SL) { return hist_bytes(SL) + kSeg; }
constexpr int kBrowFront = 64;         // dwords of front padding of a strip boundary row (>= lanes per slot)
constexpr int kPadScore = -16384;      // substitution score of padding rows / columns
constexpr int kSemI16 = 0;             // Similarity_Matrix semantics on integer scores, two queries per register
constexpr int kSemU8 = 1;              // Similarity_Matrix_Skewed semantics (saturate at 255), two queries per register
constexpr int kSemF32 = 2;             // Similarity_Matrix semantics in float32 cells (any table, any positive gap),
                                       // one query per register: the general instance behind the packed ones
constexpr int kSemF32U8 = 3;           // Similarity_Matrix_Skewed cell rule on integer-valued float32 cells, one query per
                                       // register: a lone uint8-engine query (the packed instance would carry it twice)
constexpr int kSemF16 = 4;             // Similarity_Matrix semantics on small integer scores in packed FLOAT16 cells holding
                                       // H / 2048 (exact while every value stays within +-2048), two queries per register:
                                       // the [0, 1] clamp of v_pk_add_f16 is the zero floor of the diagonal term and gfx950's
                                       // three-input packed maximum takes the two gap terms, kept as H - g: 3.5 ops per cell
constexpr float kF16Scale = 2048.0f;   // cell value = H / kF16Scale
constexpr int kSemU8H = 5;             // Similarity_Matrix_Skewed semantics in packed float16 cells holding (H + 1) / 256: the
                                       // [0, 1] clamp of v_pk_add_f16 is then the saturation at 255 (the lower clamp, H = -1,
                                       // lies below the explicit floor 1/256), so a cell costs the same four ops as kSemF16
constexpr uint32_t kU8HZero = 0x1C001C00u;   // float16 1/256 in both halves: H = 0
constexpr float kPadScoreF = -1.0e30f;
__host__ __device__ constexpr bool sem_is_float(int sem) { return sem == kSemF32 || sem == kSemF32U8; }

// LDS stride (dwords) between the profile rows of two adjacent lanes: a multiple of 4 (b128
// alignment) that is ≡ 4 (mod 8), so that the sixteen 16-byte windows of a ds_read_b128 lane
// group fall on disjoint banks whatever reference code each lane looks up (16*LS ≡ 0 mod 64).
__host__ __device__ constexpr int lane_stride(int R) {
  int rp = (R + 3) / 4 * 4;
  return (rp % 8 == 0) ? rp + 4 : rp;
}

struct ScoreArgs {
  const uint8_t *refcodes;   // [ref_len] reference as dense codes 0..ncodes-2
  int64_t ref_len;
  const int64_t *range_lo;   // [nranges] sub-problems (pieces) of the reference, [lo,hi)
  const int64_t *range_hi;
  int64_t chunk_len;         // own columns per tile
  int64_t sub_len;           // the tile's maximum is reported per sub-chunk of sub_len columns (divides chunk_len)
  int64_t warm;              // warm-up columns recomputed in front of a tile; multiple of 64
  int chunks_per_range;      // max over ranges
  const uint8_t *qbytes;     // concatenated raw query bytes
  const int64_t *qoff;       // [nq] byte offset of each query
  const int32_t *qlen;       // [nq]
  const int32_t *qsel;       // [nq] query ids sorted by length; this launch sweeps qsel[qfirst .. qfirst+qcount)
  int qfirst, qcount;
  int nq;                    // total queries (row length of keys)
  const void *stab;          // [256][ncodes] score(query byte, reference code); column ncodes-1 = pad;
                             // int16 for the packed instances, float for kSemF32
  int ncodes;
  uint32_t gap2;             // gap penalty in both halves (packed) / float bits (kSemF32)
  uint32_t clamp2;           // 255 in both halves (U8SAT)
  uint32_t pubmax;           // != 0: published maxima are clamped to this cell value (bit pattern of the instance's cell
                             // type: float16 in the low half, or float32) — the uint8 engine swept WITHOUT saturation,
                             // see host_score.h make_buckets
  unsigned long long *keys;  // [nranges][nq]  (max << 32) | (0xFFFFFFFF - sub-chunk index in the range)
  // Saturating sweep of the FLOAT engine (kSemF16 beyond its exact range, host_score.h make_buckets): every (query,
  // sub-chunk) whose maximum reaches flag_value — the cell type's cap — is appended to flag_list; those sub-chunks are
  // re-evaluated exactly afterwards (host_pipeline.h locate_saturated).  Null: not in use.
  unsigned int *flag_count;
  uint2 *flag_list;          // {query id, sub-chunk index in the range}
  uint32_t flag_cap, flag_value;
  // Sampled running maximum (template parameter MK > 1): the per-sub-chunk values — lower bounds within (MK - 1) * gap of
  // the truth — of launch-local query position p, sub-chunk s go to submax_out[p * submax_stride + s]; sw_sample_filter
  // turns those within that slack of the query's final key into flagged sub-chunks.  Null: not in use.
  uint16_t *submax_out;
  int64_t submax_stride;
  // strip-mined variant only (queries longer than 16*R rows): boundary rows between strips,
  // two ping-pong buffers of brow_stride dwords per tile, 16 dwords of front padding each
  uint32_t *brow;
  int64_t brow_stride;
};

__device__ __forceinline__ uint32_t as_u32(i16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ i16x2 as_i16x2(uint32_t v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ u16x2 as_u16x2(i16x2 v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ i16x2 to_i16x2(u16x2 v) { return __builtin_bit_cast(i16x2, v); }

// Cell arithmetic of the three instances.  A cell register is handled as raw 32 bits outside these helpers.
template <int SEM> struct Cell {
  typedef i16x2 T;
  static constexpr int kQueries = 2;
  static __device__ __forceinline__ T from_bits(uint32_t v) { return as_i16x2(v); }
  static __device__ __forceinline__ uint32_t bits(T v) { return as_u32(v); }
  static __device__ __forceinline__ T add(T d, T sc, uint32_t clamp2) {
    T x = d + sc;
    if (SEM == kSemU8) x = __builtin_elementwise_min(x, as_i16x2(clamp2));
    return x;
  }
  static __device__ __forceinline__ T vmax(T a, T b) { return __builtin_elementwise_max(a, b); }
  static __device__ __forceinline__ T sub_gap(T t, uint32_t gap2) {      // max(t - g, 0): unsigned saturation
    return to_i16x2(__builtin_elementwise_sub_sat(as_u16x2(t), __builtin_bit_cast(u16x2, gap2)));
  }
  static __device__ __forceinline__ T cell(T x, T y) { return __builtin_elementwise_max(x, y); }
};
template <int SEM> struct CellF {
  typedef float T;
  static constexpr int kQueries = 1;
  static __device__ __forceinline__ T from_bits(uint32_t v) { return __uint_as_float(v); }
  static __device__ __forceinline__ uint32_t bits(T v) { return __float_as_uint(v); }
  static __device__ __forceinline__ T add(T d, T sc, uint32_t) { return SEM == kSemF32U8 ? fminf(d + sc, 255.0f) : d + sc; }
  static __device__ __forceinline__ T vmax(T a, T b) { return fmaxf(a, b); }
  // max(w - g, n - g) == max(w, n) - g exactly (rounding is monotone): similaritymatrix.cpp:49-54
  static __device__ __forceinline__ T sub_gap(T t, uint32_t gap2) { return t - __uint_as_float(gap2); }
  static __device__ __forceinline__ T cell(T x, T y) { return fmaxf(fmaxf(x, y), 0.0f); }
};
// Packed float16 cells, handled as raw 32 bits (two halves); every operation is one VOP3P instruction.
template <> struct Cell<kSemF16> {
  typedef uint32_t T;
  static constexpr int kQueries = 2;
  static __device__ __forceinline__ T from_bits(uint32_t v) { return v; }
  static __device__ __forceinline__ uint32_t bits(T v) { return v; }
  // max(diag + score, 0): values are scaled into [0, 1), so the clamp modifier is exactly the zero floor
  static __device__ __forceinline__ T add(T d, T sc, uint32_t) {
    T r; asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(d), "v"(sc)); return r;
  }
  static __device__ __forceinline__ T vmax(T a, T b) {
    T r; asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
  }
  static __device__ __forceinline__ T vmax3(T a, T b, T c) {
    T r; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
  }
  // gap2 holds -g in both halves; the zero floor is applied by cell()
  static __device__ __forceinline__ T sub_gap(T t, uint32_t gap2) {
    T r; asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(t), "s"(gap2)); return r;
  }
  static __device__ __forceinline__ T cell(T x, T y) {
    T r; asm("v_pk_maximum3_f16 %0, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y)); return r;
  }
};
template <> struct Cell<kSemU8H> : Cell<kSemF16> {
  // adds(nw, +M) saturating at 255 / subs(nw, X): one clamped add in the (H + 1) / 256 representation
  static __device__ __forceinline__ T add(T d, T sc, uint32_t) {
    T r; asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(d), "v"(sc)); return r;
  }
  // max(x, w - g, n - g, 0): the floor H = 0 is the constant 1/256
  static __device__ __forceinline__ T cell(T x, T y) {
    T r; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(kU8HZero)); return r;
  }
};
// float32 cells hold H * 2^-k with 2^k above every value of the call (a pure exponent shift: every add, subtract
// and maximum commutes with it exactly), so that the [0, 1] clamp of the add is the zero floor and the cell takes the
// same three ops as the packed float16 instance: add clamp, max3 with the two kept (H - g) terms, subtract g.
template <> struct Cell<kSemF32> : CellF<kSemF32> {
  static __device__ __forceinline__ T add(T d, T sc, uint32_t) {
    T r; asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(d), "v"(sc)); return r;
  }
  static __device__ __forceinline__ T vmax3(T a, T b, T c) {
    T r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
  }
};
template <> struct Cell<kSemF32U8> : CellF<kSemF32U8> {};

// bit pattern of H = 0 in a cell register
template <int SEM> __host__ __device__ constexpr uint32_t zero_bits() { return SEM == kSemU8H ? kU8HZero : 0u; }

// three-input maximum where the cell type has one (packed float16), else two steps
template <int SEM> __device__ __forceinline__ typename Cell<SEM>::T cell_max3(typename Cell<SEM>::T a, typename Cell<SEM>::T b, typename Cell<SEM>::T c) {
  if constexpr (SEM == kSemF16 || SEM == kSemU8H) return Cell<SEM>::vmax3(a, b, c);
  else return Cell<SEM>::vmax(Cell<SEM>::vmax(a, b), c);
}

// value of the lane above inside the 16-lane DPP row, 0 for the first lane (row H(0,.) = 0)
__device__ __forceinline__ uint32_t row_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}

// SL = lanes per tile ("slot"): 16 (one DPP row) or 8 (half a DPP row; the DPP shift then needs one mask op
// per step, but 8*R rows fit the read length more tightly: 150 bp = 8 x 19 rows instead of 16 x 10).
// STRIPS = false: the whole query (<= SL*R rows) is one strip held in registers.
// STRIPS = true : the query is swept in strips of SL*R rows; the bottom row of strip s over the tile's
// columns goes through a per-tile global scratch row (L2-resident) and enters strip s+1 through the
// DPP `old` operand of lane 0, where the single-strip kernel gets the zero border row.
// TWIN = true (packed instances, SL = 64): ONE query per workgroup; the two halves of every register hold two
// neighbouring TILES of it (chunks 2T and 2T+1).  The halves then see different reference codes, so the profile
// holds 16-bit scores and a step reads it twice (once per code); one v_perm_b32 per row merges the two reads.
// COMB = true (with TWIN, small alphabets): the profile is indexed by the PAIR of codes the two tiles see
// (ncodes^2 entries of ready-made packed pairs), so a step reads it once and needs no merge — the cell costs what it
// costs in the two-query instances.
// MK > 1 (packed float16, two-query tiles): the running maximum is folded only every MK-th step.  A cell holding the
// maximum M decays by exactly one gap per column along its row (H(i, j + k) >= M - k g), so the value seen at the next
// folded step is within (MK - 1) g of M: the sweep's sub-chunk values become lower bounds with that slack, every
// sub-chunk within the slack of the query's final key is re-evaluated exactly (host_pipeline.h), and the cell costs
// 3 + 1/(2 MK) instead of 3.5 ops.
template <int R, int SEM, bool STRIPS = false, int SL = 16, bool TWIN = false, bool COMB = false, int MK = 1>
__global__ __launch_bounds__(256) void sw_score_kernel(const ScoreArgs a) {
  static_assert(MK == 1 || (MK == 4 && !TWIN && ((SEM == kSemF16 && !STRIPS) || SEM == kSemF32)),
                "sampled maximum: packed float16 two-query tiles, or float32 cells (one query per tile)");
  static_assert(SL == 64 || SL == 16 || SL == 8, "a slot is a whole wavefront, a DPP row or half a DPP row");
  static_assert(!(STRIPS && SL == 8), "the strip-mined instances use whole DPP rows or whole wavefronts");
  static_assert(!TWIN || ((SL == 64 || SL == 16) && !sem_is_float(SEM) && R % 2 == 0), "twin tiles: packed cells on whole-wavefront or 16-lane tiles");
  static_assert(!COMB || (TWIN && !STRIPS), "the code-pair profile belongs to the twin instances");
  constexpr bool HALF = TWIN && !COMB;                             // the profile holds 16-bit entries, two rows per dword
  constexpr int LS = HALF ? lane_stride(R / 2) : lane_stride(R);   // dwords between the profile rows of adjacent lanes
  constexpr int NQ4 = HALF ? (R / 2 + 3) / 4 : (R + 3) / 4;
  constexpr int NSLOT = 256 / SL;                                  // tiles per workgroup
  constexpr int CPL = kSeg / SL;                                   // reference codes fetched per lane per segment
  constexpr int PL = SL > 16 ? SL : 16;                            // lane positions of the profile
  constexpr int HIST = hist_bytes(SL);
  constexpr int CB = codebuf_bytes(SL);
  constexpr int VPL = kSeg / SL >= 4 ? 4 : 1;                      // boundary-row values moved per lane per segment
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t *prof = smem;                                           // [ncodes (COMB: ncodes^2)][PL lane positions][LS]
  uint8_t *codebuf = reinterpret_cast<uint8_t *>(smem + (COMB ? a.ncodes * a.ncodes : a.ncodes) * PL * LS);
  // STRIPS: [NSLOT][64] boundary-in window, then [NSLOT][64] boundary-out staging
  uint32_t *bwin = reinterpret_cast<uint32_t *>(codebuf + (TWIN ? 2 : 1) * NSLOT * CB);

  const int tid = threadIdx.x;
  const int ls = tid & (SL - 1);                                   // lane within the slot
  const int slot = tid / SL;                                       // 0..NSLOT-1 within the workgroup
  const int cgroups = (a.chunks_per_range + NSLOT - 1) / NSLOT;
  const int pair = blockIdx.x / cgroups;
  const int cg = blockIdx.x - pair * cgroups;
  const int range = blockIdx.y;
  typedef Cell<SEM> C;
  typedef typename C::T T;
  constexpr int NQ = TWIN ? 1 : C::kQueries;                       // queries per workgroup ("pair")
  const bool hasB = NQ == 2 && (2 * pair + 1) < a.qcount;
  const int qA = a.qsel[a.qfirst + NQ * pair];
  const int qB = hasB ? a.qsel[a.qfirst + 2 * pair + 1] : qA;

  // ---- query profile for this workgroup's pair (rows row0 .. row0 + SL*R - 1) --------------
  // 16 lane positions of a DPP row; with SL = 8 positions 8..15 repeat 0..7, so that the sixteen lanes of a
  // ds_read_b128 lane group still hit disjoint banks
  const int mA = a.qlen[qA], mB = a.qlen[qB];
  auto build_profile = [&](int row0) {
    const uint8_t *xA = a.qbytes + a.qoff[qA];
    const uint8_t *xB = a.qbytes + a.qoff[qB];
    const int per_code = PL * R;
    for (int e = tid; e < (COMB ? a.ncodes * a.ncodes : a.ncodes) * per_code; e += 256) {
      const int c = e / per_code;
      const int rem = e - c * per_code;
      const int ll = rem / R, r = rem - ll * R;
      const int i = row0 + (ll & (SL - 1)) * R + r;
      uint32_t e32;
      if (COMB) {
        // entry (cA * ncodes + cB): low half = this row against the first tile's code, high half = against the second's
        const int16_t *st = static_cast<const int16_t *>(a.stab);
        constexpr int kPadEntry = SEM == kSemF16 ? (int)(int16_t)0xC800 : (SEM == kSemU8H ? (int)(int16_t)0xD400 : kPadScore);
        const int cA = c / a.ncodes, cB = c - cA * a.ncodes;
        const int sa = (i < mA) ? st[(int)xA[i] * a.ncodes + cA] : kPadEntry;
        const int sb = (i < mA) ? st[(int)xA[i] * a.ncodes + cB] : kPadEntry;
        prof[(c * PL + ll) * LS + r] = (uint32_t)(uint16_t)sa | ((uint32_t)(uint16_t)sb << 16);
        continue;
      }
      if (TWIN) {
        const int16_t *st = static_cast<const int16_t *>(a.stab);
        const int sa = (i < mA) ? st[(int)xA[i] * a.ncodes + c]
                                : (SEM == kSemU8H ? (int)(int16_t)0xD400 /* float16 -64 */
                                   : (SEM == kSemF16 ? (int)(int16_t)0xC800 /* float16 -8 */ : kPadScore));
        reinterpret_cast<uint16_t *>(prof)[((c * PL + ll) * LS) * 2 + r] = (uint16_t)sa;
        continue;
      }
      if (sem_is_float(SEM)) {
        const float *ft = static_cast<const float *>(a.stab);
        e32 = __float_as_uint((i < mA) ? ft[(int)xA[i] * a.ncodes + c] : kPadScoreF);
      } else {
        // 16-bit table entries: int16 scores, or float16 bit patterns for the packed float16 instance
        const int16_t *st = static_cast<const int16_t *>(a.stab);
        constexpr int kPadEntry = SEM == kSemF16 ? (int)(int16_t)0xC800 /* float16 -8 = -16384 / 2048 */
                                  : (SEM == kSemU8H ? (int)(int16_t)0xD400 /* float16 -64 = -16384 / 256 */ : kPadScore);
        const int sa = (i < mA) ? st[(int)xA[i] * a.ncodes + c] : kPadEntry;
        const int sb = (i < mB) ? st[(int)xB[i] * a.ncodes + c] : kPadEntry;
        e32 = (uint32_t)(uint16_t)sa | ((uint32_t)(uint16_t)sb << 16);
      }
      prof[(c * PL + ll) * LS + r] = e32;
    }
  };
  build_profile(0);

  // ---- this slot's tile -------------------------------------------------------------------
  const int64_t rlo = a.range_lo[range], rhi = a.range_hi[range];
  const int64_t nchunks = (rhi - rlo + a.chunk_len - 1) / a.chunk_len;
  const int64_t chunk = ((int64_t)cg * NSLOT + slot) * (TWIN ? 2 : 1);   // TWIN: chunk and chunk + 1
  const bool active = chunk < nchunks;
  const int64_t own_lo = rlo + chunk * a.chunk_len;
  const int64_t own_hi = (own_lo + a.chunk_len < rhi) ? own_lo + a.chunk_len : rhi;
  const int64_t s0 = own_lo - a.warm;                 // reference index of stream position 0
  const bool active2 = TWIN && chunk + 1 < nchunks;   // the tile in the high halves
  const int64_t own_lo2 = own_lo + a.chunk_len;
  const int64_t own_hi2 = (own_lo2 + a.chunk_len < rhi) ? own_lo2 + a.chunk_len : rhi;
  const uint32_t pad = (uint32_t)(a.ncodes - 1);
  const uint32_t pad4 = pad * 0x01010101u;

  // codes of stream positions seg*64 + CPL*ls .. +CPL-1 (pad outside [rlo, own_hi)): CPL/4 dwords, or one
  // byte per lane when the slot is a whole wavefront
  struct Codes { uint32_t w[CPL >= 4 ? CPL / 4 : 1]; };
  auto stage_load = [&](int seg, bool second = false) -> Codes {
    Codes out;
    const bool act = second ? active2 : active;
    const int64_t hi = second ? own_hi2 : own_hi;
    const int64_t c0 = s0 + (second ? a.chunk_len : 0) + (int64_t)seg * kSeg + CPL * ls;
    if (CPL >= 4) {
#pragma unroll
      for (int d = 0; d < CPL / 4; ++d) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int64_t col = c0 + 4 * d + b;
          const bool ok = act && col >= rlo && col < hi;
          const uint32_t code = ok ? (uint32_t)a.refcodes[col] : pad;
          w |= code << (8 * b);
        }
        out.w[d] = w;
      }
    } else {
      const bool ok = act && c0 >= rlo && c0 < hi;
      out.w[0] = ok ? (uint32_t)a.refcodes[c0] : pad;
    }
    return out;
  };

  uint8_t *buf = codebuf + slot * CB;
  uint32_t *buf32 = reinterpret_cast<uint32_t *>(buf);
  const uint8_t *buf_lane = buf + HIST - ls;                       // + k = code of step k
  uint8_t *buf2 = codebuf + (NSLOT + slot) * CB;                   // TWIN: window of the second tile
  const uint8_t *buf2_lane = buf2 + HIST - ls;
  const uint32_t *prof_lane = prof + (tid & (PL - 1)) * LS;
  uint32_t *buf2_32 = reinterpret_cast<uint32_t *>(buf2);
  auto window_put = [&](const Codes &c) {
    if (CPL >= 4) {
#pragma unroll
      for (int d = 0; d < CPL / 4; ++d) buf32[HIST / 4 + (CPL / 4) * ls + d] = c.w[d];
    } else {
      buf[HIST + ls] = (uint8_t)c.w[0];
    }
  };
  auto window_put2 = [&](const Codes &c) {                          // TWIN: the second tile's window
    if (CPL >= 4) {
#pragma unroll
      for (int d = 0; d < CPL / 4; ++d) buf2_32[HIST / 4 + (CPL / 4) * ls + d] = c.w[d];
    } else {
      buf2[HIST + ls] = (uint8_t)c.w[0];
    }
  };
  // first fill: history = padding; later: the last HIST bytes of the window move to its front
  auto window_init = [&]() {
    if (SL == 64) { buf[ls] = (uint8_t)pad; if (TWIN) buf2[ls] = (uint8_t)pad; }
    else if (ls < HIST / 4) { buf32[ls] = pad4; if (TWIN) buf2_32[ls] = pad4; }
  };
  auto window_slide = [&]() {
    if (SL == 64) {
      const uint8_t h = buf[kSeg + ls]; buf[ls] = h;
      if (TWIN) { const uint8_t h2 = buf2[kSeg + ls]; buf2[ls] = h2; }
    }
    else {
      const uint32_t h = buf32[kSeg / 4 + (ls & 3)]; if (ls < HIST / 4) buf32[ls] = h;
      if (TWIN) { const uint32_t h2 = buf2_32[kSeg / 4 + (ls & 3)]; if (ls < HIST / 4) buf2_32[ls] = h2; }
    }
  };
  // value of the lane above: DPP inside the row (16/8 lanes) or across the wavefront (64 lanes)
  auto shift_in = [&](uint32_t v, uint32_t border) -> uint32_t {
    if (SL == 64) return (uint32_t)__builtin_amdgcn_update_dpp((int)border, (int)v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp((int)border, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  };

  const int64_t total_steps = a.warm + a.chunk_len + SL;           // + SL-1 skew, + 1 max-fold drain
  const int nseg = (int)((total_steps + kSeg - 1) / kSeg);

  T mx = C::from_bits(0u);
  const int code_stride = PL * LS;                                 // dwords per reference code
  const int strip_rows = SL * R;
  const int mmax = mA > mB ? mA : mB;
  const int nstrips = STRIPS ? (mmax + strip_rows - 1) / strip_rows : 1;
  uint32_t first_lane_zero = ls == 0 ? 0u : 0xFFFFFFFFu;           // SL = 8: zero border row for lane 0 of the slot
  asm volatile("" : "+v"(first_lane_zero));                        // keep it a plain v_and_b32 (2 cycles), not a v_cndmask (4)
  // STRIPS: this tile's ping-pong boundary rows (global), and its LDS windows
  const size_t tile_id = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NSLOT + slot;
  uint32_t *brow0 = STRIPS ? a.brow + tile_id * 2 * (size_t)a.brow_stride : nullptr;
  uint32_t *bin_w = bwin + slot * kSeg;
  uint32_t *bout_w = bwin + NSLOT * kSeg + slot * kSeg;

  // per-sub-chunk maximum -> per-query key.  Lanes lag lane 0 by up to SL-1 columns, so up to SL-1 trailing
  // columns of a sub-chunk are reported with the next one; the host widens its search accordingly.
  const int64_t subs_per_tile = a.chunk_len / a.sub_len;
  uint32_t best_a = zero_bits<SEM>() & 0xFFFFu, best_b = best_a;   // this tile's best published value per query (H = 0: nothing to report)
  auto slot_max = [&]() -> uint32_t {                              // maximum of mx over the slot's lanes
    uint32_t m32 = C::bits(mx);
#pragma unroll
    for (int off = SL / 2; off >= 1; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)m32, off, SL);
      m32 = C::bits(C::vmax(C::from_bits(m32), C::from_bits(o)));
    }
    return m32;
  };
  // atomicMax on the query's key, skipped when the key already holds something at least as large: a lone query's
  // tiles (tens of thousands) all aim at ONE address, and the serialised atomics cost more than the sweep's last 15 %
  auto key_max = [&](unsigned long long *addr, unsigned long long v) {
    if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(addr, v);
  };
  auto publish_value = [&](int64_t sub, uint32_t m32) {
    // Only a sub-chunk that strictly beats the tile's earlier ones can become the query's (max, first
    // sub-chunk) key, so all others skip the atomic (a tile publishes O(log) times, not once per sub-chunk).
    if (ls == 0 && active) {
      const unsigned long long tag = 0xFFFFFFFFull - (unsigned long long)(chunk * subs_per_tile + sub);
      unsigned long long *k = a.keys + (size_t)range * a.nq;
      if (sem_is_float(SEM)) {
        if (MK > 1 && a.submax_out != nullptr)                       // (one query per workgroup: launch-local position = pair)
          reinterpret_cast<uint32_t *>(a.submax_out)[(size_t)pair * (size_t)a.submax_stride + (size_t)(chunk * subs_per_tile + sub)] = m32;
        if (a.pubmax != 0u && m32 > a.pubmax) m32 = a.pubmax;       // non-negative floats order like their bits
        if (m32 > best_a) {
          best_a = m32;
          key_max(k + qA, ((unsigned long long)m32 << 32) | tag);
        }
      } else {
        uint32_t va = m32 & 0xFFFFu, vb = m32 >> 16;
        if (a.flag_list != nullptr) {                                 // (positive float16 values order like their bits)
          const unsigned int gsub = (unsigned int)(chunk * subs_per_tile + sub);
          if (va >= a.flag_value) { const unsigned int at = atomicAdd(a.flag_count, 1u); if (at < a.flag_cap) a.flag_list[at] = make_uint2((unsigned int)qA, gsub); }
          if (hasB && vb >= a.flag_value) { const unsigned int at = atomicAdd(a.flag_count, 1u); if (at < a.flag_cap) a.flag_list[at] = make_uint2((unsigned int)qB, gsub); }
          if (TWIN && active2 && vb >= a.flag_value) {
            const unsigned int at = atomicAdd(a.flag_count, 1u);
            if (at < a.flag_cap) a.flag_list[at] = make_uint2((unsigned int)qA, gsub + (unsigned int)subs_per_tile);
          }
        }
        if (MK > 1 && a.submax_out != nullptr) {
          const size_t at = (size_t)(2 * pair) * (size_t)a.submax_stride + (size_t)(chunk * subs_per_tile + sub);
          a.submax_out[at] = (uint16_t)va;
          if (hasB) a.submax_out[at + (size_t)a.submax_stride] = (uint16_t)vb;
        }
        if (a.pubmax != 0u) { va = va > a.pubmax ? a.pubmax : va; vb = vb > a.pubmax ? a.pubmax : vb; }
        if (va > best_a) { best_a = va; key_max(k + qA, ((unsigned long long)va << 32) | tag); }
        if (hasB && vb > best_b) { best_b = vb; key_max(k + qB, ((unsigned long long)vb << 32) | tag); }
        if (TWIN && active2 && vb > best_b) {                        // the second tile of the same query
          best_b = vb;
          key_max(k + qA, ((unsigned long long)vb << 32) | (tag - (unsigned long long)subs_per_tile));
        }
      }
    }
  };
  auto publish = [&](int64_t sub) {
    publish_value(sub, slot_max());
    mx = C::from_bits(0u);
  };
  // STRIPS: a sub-chunk's maximum accumulates over all strips in LDS before it can be published
  uint32_t *submax = bwin + 2 * NSLOT * kSeg + slot * 64;
  auto fold_sub = [&](int64_t sub) {
    const uint32_t m32 = slot_max();
    if (ls == 0) submax[sub] = C::bits(C::vmax(C::from_bits(submax[sub]), C::from_bits(m32)));
    mx = C::from_bits(0u);
  };
  if (STRIPS) {
    for (int e = ls; e < 64; e += SL) submax[e] = 0u;
  }
  const int segs_per_sub = (int)(a.sub_len / kSeg);
  const int warm_segs = (int)(a.warm / kSeg);
  int64_t sub = 0;

  for (int strip = 0; strip < nstrips; ++strip) {
    if (STRIPS) sub = 0;
    if (STRIPS && strip > 0) {
      // the boundary row written by this tile's own lanes in the previous strip is re-read below:
      // drain the stores, invalidate this CU's L1 (it may hold the row's lines from two strips ago)
      __threadfence();
      __syncthreads();                                             // everyone done with the old profile
      build_profile(strip * strip_rows);
    }
    const uint32_t *bin_g = STRIPS ? brow0 + (size_t)(strip & 1) * a.brow_stride + kBrowFront : nullptr;
    uint32_t *bout_g = STRIPS ? brow0 + (size_t)((strip + 1) & 1) * a.brow_stride + kBrowFront : nullptr;
    const bool rd = STRIPS && strip > 0, wr = STRIPS && strip + 1 < nstrips;
    // boundary values of stream positions seg*64 + VPL*ls .. +VPL-1
    auto bin_load = [&](int seg) -> uint4 {
      if (!rd) return make_uint4(zero_bits<SEM>(), zero_bits<SEM>(), zero_bits<SEM>(), zero_bits<SEM>());
      if (VPL == 4) return *reinterpret_cast<const uint4 *>(bin_g + (size_t)seg * kSeg + 4 * ls);
      return make_uint4(bin_g[(size_t)seg * kSeg + ls], 0, 0, 0);
    };
    auto bin_put = [&](const uint4 &v) {
      if (VPL == 4) *reinterpret_cast<uint4 *>(bin_w + 4 * ls) = v;
      else bin_w[ls] = v.x;
    };

    Codes nextcodes = stage_load(0);
    Codes nextcodes2 = nextcodes;
    window_init();
    window_put(nextcodes);
    nextcodes = stage_load(1);
    if (TWIN) {
      window_put2(stage_load(0, true));
      nextcodes2 = stage_load(1, true);
    }
    uint4 nextb = make_uint4(zero_bits<SEM>(), zero_bits<SEM>(), zero_bits<SEM>(), zero_bits<SEM>());
    if (STRIPS) {
      bin_put(bin_load(0));
      nextb = bin_load(1);
    }
    __syncthreads();                                               // profile + first window ready

    T H[R];
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = C::from_bits(zero_bits<SEM>());
    uint32_t up_prev = zero_bits<SEM>();
    constexpr bool kKeepsHg = SEM == kSemF16 || SEM == kSemF32;    // instances that keep H - g of every cell
    T Hg[kKeepsHg ? R : 1];
#pragma unroll
    for (int r = 0; r < (kKeepsHg ? R : 1); ++r) Hg[r] = C::sub_gap(C::from_bits(0u), a.gap2);   // 0 - g

    for (int seg = 0; seg < nseg; ++seg) {
#pragma unroll 4
      for (int k = 0; k < kSeg; ++k) {
        const uint32_t c = COMB ? (uint32_t)buf_lane[k] * (uint32_t)a.ncodes + (uint32_t)buf2_lane[k] : (uint32_t)buf_lane[k];
        const u32x4 *pp = static_cast<const u32x4 *>(__builtin_assume_aligned(prof_lane + c * code_stride, 16));
        uint32_t p[HALF ? R : NQ4 * 4];
        if (HALF) {
          // 16-bit scores of this lane's rows for the two tiles' codes, merged row by row: low half = first tile
          const u32x4 *pp2 = static_cast<const u32x4 *>(__builtin_assume_aligned(prof_lane + (uint32_t)buf2_lane[k] * code_stride, 16));
          uint32_t d1[NQ4 * 4], d2[NQ4 * 4];
#pragma unroll
          for (int q = 0; q < NQ4; ++q) {
            const u32x4 v = pp[q], u = pp2[q];
            d1[4 * q + 0] = v.x; d1[4 * q + 1] = v.y; d1[4 * q + 2] = v.z; d1[4 * q + 3] = v.w;
            d2[4 * q + 0] = u.x; d2[4 * q + 1] = u.y; d2[4 * q + 2] = u.z; d2[4 * q + 3] = u.w;
          }
#pragma unroll
          for (int r = 0; r < R; ++r)
            p[r] = __builtin_amdgcn_perm(d2[r >> 1], d1[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
        } else {
#pragma unroll
          for (int q = 0; q < NQ4; ++q) {
            const u32x4 v = pp[q];
            p[4 * q + 0] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
          }
        }
        uint32_t up;                                               // H(i0-1, j) of the lane above
        if (STRIPS) {
          // lane 0 takes the previous strip's bottom row through the DPP `old` operand
          up = shift_in(C::bits(H[R - 1]), bin_w[k]);
        } else {
          // zero border row H(0, .): bound_ctrl supplies it (no `old` operand to set up)
          if (SEM == kSemU8H) {
            // the border row is the bit pattern of H = 0, not zero: `old` operand, and a bit-select for 8-lane tiles
            up = shift_in(C::bits(H[R - 1]), kU8HZero);
            if (SL == 8) up = (up & first_lane_zero) | (kU8HZero & ~first_lane_zero);
          } else {
            if (SL == 64) up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C::bits(H[R - 1]), 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
            else up = row_shr1(C::bits(H[R - 1]));
            if (SL == 8) up &= first_lane_zero;                    // lane 8 of the DPP row starts another tile
          }
        }
        T diag = C::from_bits(up_prev);                            // H(i0-1, j-1)
        T north = C::from_bits(up);
        up_prev = up;
        T tpend = C::from_bits(0u);
        (void)tpend;
        if constexpr (SEM == kSemF16 || SEM == kSemF32) {
          // H = max(clamp0(NW + s), W - g, N - g): the cell keeps H (next step's diagonal) and H - g (this row's west
          // term next step, the row below's north term now) — add, maximum3, add per cell; the running maximum takes
          // two cells per maximum3
          T ng = C::sub_gap(north, a.gap2);                        // (row above the lane's first) - g
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const T w = H[r];
            const T x = C::add(diag, C::from_bits(p[r]), a.clamp2);
            const T h = C::vmax3(x, Hg[r], ng);
            if (MK == 1 || (k & (MK - 1)) == MK - 1) {             // (compile-time per unrolled step)
              if (r & 1) mx = C::vmax3(mx, tpend, h);
              else if (r + 1 < R) tpend = h;
              else mx = C::vmax(mx, h);
            }
            diag = w;
            H[r] = h;
            ng = Hg[r] = C::sub_gap(h, a.gap2);
          }
        } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const T w = H[r];
          const T x = C::add(diag, C::from_bits(p[r]), a.clamp2);
          const T t = C::vmax(w, north);
          if (SEM == kSemU8H) {
            // three-input maximum: two odd rows per running-maximum op (t covers cells (r, j-1) and (r-1, j))
            if ((r & 3) == 1) { if (r + 2 < R) tpend = t; else mx = C::vmax(mx, t); }
            if ((r & 3) == 3) mx = cell_max3<SEM>(mx, tpend, t);
          } else if (r & 1) mx = C::vmax(mx, t);                   // covers (r, j-1) and (r-1, j)
          const T y = C::sub_gap(t, a.gap2);
          const T h = C::cell(x, y);
          diag = w;
          H[r] = h;
          north = h;
        }
        if (R & 1) mx = C::vmax(mx, H[R - 1]);                     // odd R: the last (even) row is in no tracked t
        }
        if (STRIPS) {
          if (ls == SL - 1) bout_w[k] = C::bits(H[R - 1]);         // bottom row at stream position seg*64+k-(SL-1)
        }
      }
      // slide the code window: keep the last HIST bytes as history, append the prefetched segment
      window_slide();
      window_put(nextcodes);
      nextcodes = stage_load(seg + 2);
      if (TWIN) {
        window_put2(nextcodes2);
        nextcodes2 = stage_load(seg + 2, true);
      }
      {
        // lane 0 has just finished a sub-chunk (and it is not the tile's last): report and restart the maximum
        const int done = seg + 1 - warm_segs;
        if (done > 0 && done % segs_per_sub == 0 && done / segs_per_sub < subs_per_tile) {
          if (STRIPS) fold_sub(sub++); else publish(sub++);
        }
      }
      if (STRIPS) {
        if (wr) {
          // flush 64 bottom-row values: positions seg*64 - (SL-1) + (0..63)
          if (VPL == 4) {
            uint32_t *g = bout_g + (int64_t)seg * kSeg - (SL - 1) + 4 * ls;
            const uint4 v = *reinterpret_cast<const uint4 *>(bout_w + 4 * ls);
            g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
          } else {
            bout_g[(int64_t)seg * kSeg - (SL - 1) + ls] = bout_w[ls];
          }
        }
        bin_put(nextb);
        nextb = bin_load(seg + 2);
      }
    }
    if (STRIPS) fold_sub(sub);                                     // this strip's last (or only) sub-chunk
  }

  if (STRIPS) {
    // (the last sub-chunk of every strip was folded at the end of the strip loop body below)
    for (int64_t k2 = 0; k2 < subs_per_tile; ++k2) {
      const uint32_t v = submax[k2];                               // written by lane 0 of this slot, same wavefront
      publish_value(k2, v);
    }
  } else {
    publish(sub);                                                  // the tile's last (or only) sub-chunk
  }
}

// Sampled sweep (MK > 1): every sub-chunk whose value lies within `slack` (in H units) of its query's final key may hold
// the true maximum — append it to the flag list.  grid.y = launch-local query position, threads stride over sub-chunks.
// F32V: float32 cells (values and key are float bit patterns in the sweep's scaled units, and so is `slack`).
// qcnt / per_query_cap: candidates are also counted per query, and a query appends at most per_query_cap + 1 of them — one
// low-complexity read with a hundred thousand near-equal hits cannot crowd the list; the host re-sweeps exactly the
// queries whose count exceeds the cap (host_pipeline.h) and keeps everybody else's candidates.
template <bool F32V>
__global__ __launch_bounds__(256) void sw_sample_filter(const void *submax, int64_t stride, int64_t nsub, const int32_t *qsel,
                                                        int qfirst, int qcount, const unsigned long long *keys, float slack,
                                                        unsigned int *flag_count, uint2 *flag_list, uint32_t flag_cap,
                                                        unsigned int *qcnt = nullptr, uint32_t per_query_cap = 0,
                                                        uint32_t sub_offset = 0) {
  const int pos = blockIdx.y;
  if (pos >= qcount) return;
  const int q = qsel[qfirst + pos];
  const float best = F32V ? __uint_as_float((uint32_t)(keys[q] >> 32))
                          : (float)__builtin_bit_cast(_Float16, (uint16_t)(keys[q] >> 32)) * 2048.0f;
  if (!(best > 0.0f)) return;
  const float thr = best - slack;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nsub; s += (int64_t)gridDim.x * blockDim.x) {
    const size_t at = (size_t)pos * (size_t)stride + (size_t)s;
    const float v = F32V ? __uint_as_float(static_cast<const uint32_t *>(submax)[at])
                         : (float)__builtin_bit_cast(_Float16, static_cast<const uint16_t *>(submax)[at]) * 2048.0f;
    if (v > 0.0f && v >= thr) {
      if (qcnt != nullptr && atomicAdd(&qcnt[q], 1u) > per_query_cap) continue;
      const unsigned int at = atomicAdd(flag_count, 1u);
      if (at < flag_cap) flag_list[at] = make_uint2((unsigned int)q, (unsigned int)s + sub_offset);
    }
  }
}

// The FIRST candidates, in ascending sub-chunk order, of the queries that exceeded their cap in sw_sample_filter (one wavefront
// per query position; everybody else returns at once).  The uint8 engine's maximum is capped at 255: a query whose key sits at the
// cap is decided by the first sub-chunk that truly holds a 255 and its right neighbour (the skewed storage order runs along
// anti-diagonals, host_pipeline.h), so its first few candidates usually settle it without a second sweep.
// first[q * (K + 1)]: count (bit 31: the scan stopped before the end of the row), then the sub-chunk indices.
constexpr int kFirstCandidates = 8;
template <bool F32V>
__global__ __launch_bounds__(64) void sw_sample_first(const void *submax, int64_t stride, int64_t nsub, const int32_t *qsel,
                                                      int qfirst, int qcount, const unsigned long long *keys, float slack,
                                                      const unsigned int *qcnt, uint32_t per_query_cap, uint32_t sub_offset,
                                                      uint32_t *first) {
  constexpr int K = kFirstCandidates;
  const int pos = blockIdx.x;
  if (pos >= qcount) return;
  const int q = qsel[qfirst + pos];
  if (qcnt[q] <= per_query_cap) return;
  const float best = F32V ? __uint_as_float((uint32_t)(keys[q] >> 32))
                          : (float)__builtin_bit_cast(_Float16, (uint16_t)(keys[q] >> 32)) * 2048.0f;
  const float thr = best - slack;
  const int lane = threadIdx.x;
  uint32_t *out = first + (size_t)q * (K + 1);
  int n = 0;
  int64_t base = 0;
  for (; base < nsub && n < K; base += 64) {
    const int64_t s = base + lane;
    float v = 0.0f;
    if (s < nsub) {
      const size_t at = (size_t)pos * (size_t)stride + (size_t)s;
      v = F32V ? __uint_as_float(static_cast<const uint32_t *>(submax)[at])
               : (float)__builtin_bit_cast(_Float16, static_cast<const uint16_t *>(submax)[at]) * 2048.0f;
    }
    const bool hit = v > 0.0f && v >= thr;
    const unsigned long long mask = __ballot(hit);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (hit && n + rank < K) out[1 + n + rank] = (uint32_t)s + sub_offset;
    n += __popcll(mask);
  }
  if (lane == 0) out[0] = (uint32_t)(n < K ? n : K) | ((base < nsub || n > K) ? 0x80000000u : 0u);
}

}  // namespace mi355sw
