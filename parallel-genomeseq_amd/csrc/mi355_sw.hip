// mi355_sw.hip — host side of the C-ABI in include/mi355_sw.h (gfx950 only, no CPU fallback).
//
// Pipeline for every alignment:
//   1. score pass      sw_score_kernel  — packed 16-bit wavefront sweep over (query pair x chunk)
//                                         tiles, per-query (max, first chunk) by 64-bit atomicMax
//   2. locate          sw_exact_kernel  — the tile(s) that can hold the first maximum in the
//                                         reference's storage order -> argmax cell
//   3. traceback       sw_exact_kernel  — window left of the argmax -> greedy decisions,
//                      sw_walk_kernel   — the walk itself (smithwaterman.cpp:40-78)
// Problems the score kernel does not cover (see bucket_fast_ok) run 2+3 on the whole matrix.
#include "../../include/mi355_sw.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <future>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sw_exact_kernel.h"
#include "sw_score_kernel.h"
#include "sw_wave_kernel.h"
#include "sw_strip_kernel.h"

using namespace mi355sw;

namespace {

constexpr size_t kProfileLdsMax = 120 * 1024;  // LDS budget of the query profile (ncodes x 16 lanes x stride x 4 B)
constexpr int kMaxRowsFast = 512;             // 16 lanes x R <= 32 rows in one strip; longer queries are strip-mined
constexpr size_t kDirsBudget = 16ull << 30;    // bytes of traceback decisions per exact launch
constexpr size_t kExactLdsMax = 159 * 1024;    // dynamic part; the wide instance adds < 1 KiB of static LDS

// MI355_SW_TRACE=1: wall-clock of the host-side phases of every call on stderr (diagnostic)
struct HostTrace {
  const char *name;
  std::chrono::steady_clock::time_point t0;
  explicit HostTrace(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~HostTrace() {
    static const bool on = std::getenv("MI355_SW_TRACE") != nullptr;
    if (on) std::fprintf(stderr, "[mi355_sw] %-28s %9.3f ms\n", name,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct RefData {
  DevBuf bytes, codes;
  size_t n = 0;
  int ncodes = 0;                 // incl. pad
  int code_of[256];
  uint8_t byte_of[256];
  void release() { bytes.release(); codes.release(); n = 0; }
};

struct QueryBatch {
  DevBuf bytes, lens, offs, sel;  // concatenated bytes (16-byte aligned starts), lengths, offsets, length-sorted ids
  std::vector<int32_t> len;
  std::vector<int64_t> off;
  std::vector<int32_t> order;     // query ids sorted by length (stable)
  size_t nq = 0;
  int maxlen = 0;
  void release() { bytes.release(); lens.release(); offs.release(); sel.release(); }
};

struct Range { int64_t lo, hi; };

// One alignment's intermediate state on the host
struct Located {
  float score = 0;
  int64_t ix = 0, iy = 0;         // argmax, iy relative to the range start (1-based)
};

}  // namespace

struct mi355_sw_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};
  std::string err;
  RefData ref;                    // resident reference (set_reference)
  QueryBatch batch;               // resident queries (batch_upload)
  RefData adhoc;                  // reference of the last mi355_sw_align-style call, kept while its content hash
  uint64_t adhoc_hash = 0;        // matches (one-by-one driver loops pass the same reference every time)
  bool adhoc_valid = false;
  QueryBatch one;                 // the single query of such a call
  // scratch
  DevBuf keys, ranges, stab, ftab, lut, probs, dirs, outs_f, outs_i, cons, walkp, hmat, brow, wprobs;
  // host sides of small per-call uploads: they must outlive the asynchronous copies, and the tables are only
  // sent again when they change
  std::vector<int64_t> h_ranges;
  std::vector<int16_t> h_stab;
  std::vector<float> h_ftab;
  // event pairs around the score launches of a call, read back after the call's first synchronisation
  std::vector<hipEvent_t> score_ev;
  size_t score_ev_used = 0;
  double timings[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

#define HIPCHK(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return MI355_SW_ENODEV;                                                              \
    }                                                                                      \
  } while (0)

int fail(mi355_sw_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

inline float lut_or(const mi355_sw_params &p, uint8_t a, uint8_t b) {
  if (p.lut) return p.lut[(size_t)a * 256 + b];
  return a == b ? p.match : p.mismatch;
}

// similaritymatrix.cpp:376-384
inline int sat8(float a) { return a < 0 ? 0 : (a > 255 ? 255 : (int)(uint8_t)a); }

struct U8Params { int M, X, G; };
U8Params u8_params(const mi355_sw_params &p) {
  return {sat8(lut_or(p, 'A', 'A')), sat8(-lut_or(p, 'A', 'T')), sat8(p.gap)};   // :389-392
}

// host twin of order_key<> (sw_exact_kernel.h)
unsigned long long host_order_key(int sem, int64_t i, int64_t j, int64_t m, int64_t n) {
  if (sem == MI355_SW_F32) return ((unsigned long long)j << 32) | (unsigned long long)i;
  const int64_t len_x = n + 1, len_y = m + 1;
  const int64_t nrows = std::min(len_x, len_y), ncols = std::max(len_x, len_y);
  const int64_t ti = j, tj = i;
  int64_t ri, rj;
  if (ti + tj < nrows - 1) { ri = ti; rj = ti + tj; }
  else if (ti + tj > ncols - 1) { ri = ti - ncols + len_y; rj = ti + tj - (ncols - 1) - 1; }
  else { ri = (len_x <= len_y) ? ti : len_y - 1 - tj; rj = ti + tj; }
  return ((unsigned long long)rj << 32) | (unsigned long long)ri;
}

int upload_reference(mi355_sw_ctx *ctx, RefData &r, const char *y, size_t ny) {
  HostTrace trace_("upload_reference");
  bool present[256] = {false};
  const uint8_t *u = reinterpret_cast<const uint8_t *>(y);
  for (size_t k = 0; k < ny; ++k) present[u[k]] = true;
  int nc = 0;
  for (int b = 0; b < 256; ++b) {
    r.code_of[b] = -1;
    if (present[b]) { r.code_of[b] = nc; r.byte_of[nc] = (uint8_t)b; ++nc; }
  }
  r.ncodes = nc + 1;   // + pad
  r.n = ny;
  if (r.bytes.ensure(ny + 64) || r.codes.ensure(ny + 64)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(reference) failed");
  std::vector<uint8_t> codes(ny);
  for (size_t k = 0; k < ny; ++k) codes[k] = (uint8_t)r.code_of[u[k]];
  HIPCHK(ctx, hipMemcpyAsync(r.bytes.p, y, ny, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(r.codes.p, codes.data(), ny, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// 64-bit content hash, four independent multiply-rotate lanes (memory-bound; ~3 ms for 50 MB)
uint64_t content_hash_part(const char *p, size_t n) {
  uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ n, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0xD6E8FEB86659FD93ull};
  size_t k = 0;
  for (; k + 32 <= n; k += 32) {
    uint64_t w[4];
    memcpy(w, p + k, 32);
    for (int l = 0; l < 4; ++l) { h[l] = (h[l] ^ w[l]) * 0x9FB21C651E98DF25ull; h[l] = (h[l] << 29) | (h[l] >> 35); }
  }
  for (; k < n; ++k) { h[k & 3] = (h[k & 3] ^ (uint8_t)p[k]) * 0x9FB21C651E98DF25ull; h[k & 3] = (h[k & 3] << 29) | (h[k & 3] >> 35); }
  uint64_t r = h[0];
  for (int l = 1; l < 4; ++l) r = (r ^ h[l]) * 0xBF58476D1CE4E5B9ull + (r >> 31);
  return r ^ (r >> 32);
}

// Hash of a whole buffer: four independent quarters (hashed on helper threads when the buffer is large), combined.
uint64_t content_hash(const char *p, size_t n) {
  if (n < ((size_t)4 << 20)) return content_hash_part(p, n);
  const size_t q = (n / 4) & ~(size_t)31;
  std::future<uint64_t> f[3];
  for (int k = 0; k < 3; ++k) f[k] = std::async(std::launch::async, content_hash_part, p + (size_t)(k + 1) * q, k == 2 ? n - 3 * q : q);
  uint64_t r = content_hash_part(p, q);
  for (int k = 0; k < 3; ++k) r = (r ^ f[k].get()) * 0xBF58476D1CE4E5B9ull + (r >> 29);
  return r;
}

// Reference of a single-alignment call: re-used from the previous call when its bytes are identical.
// `known_hash`: the caller has already hashed y.
int adhoc_reference(mi355_sw_ctx *ctx, const char *y, size_t ny, const RefData **out, const uint64_t *known_hash = nullptr) {
  const uint64_t h = known_hash ? *known_hash : content_hash(y, ny);
  if (!(ctx->adhoc_valid && ctx->adhoc.n == ny && ctx->adhoc_hash == h)) {
    ctx->adhoc_valid = false;
    int rc = upload_reference(ctx, ctx->adhoc, y, ny);
    if (rc) return rc;
    ctx->adhoc_hash = h;
    ctx->adhoc_valid = true;
  }
  *out = &ctx->adhoc;
  return 0;
}

// One-by-one loops against a large reference (src/sw_solve_big.cpp:78-92: a new aligner per read, same reference):
// hashing 50 MB costs as much as aligning against it, so the call starts on the resident copy while a helper
// thread re-hashes the caller's buffer, and is repeated on a fresh upload in the rare case the content changed.
struct AdhocSpeculation {
  std::future<uint64_t> hash;
  bool active = false;
};
int adhoc_begin(mi355_sw_ctx *ctx, const char *y, size_t ny, const RefData **out, AdhocSpeculation &sp) {
  if (ctx->adhoc_valid && ctx->adhoc.n == ny && ny >= ((size_t)1 << 20)) {
    sp.hash = std::async(std::launch::async, content_hash, y, ny);
    sp.active = true;
    *out = &ctx->adhoc;
    return 0;
  }
  return adhoc_reference(ctx, y, ny, out);
}
// true: the resident copy was the right one (or nothing was speculated); false: *out now points at a fresh upload
// (or rc reports why not) and the caller must repeat its work
bool adhoc_confirm(mi355_sw_ctx *ctx, const char *y, size_t ny, const RefData **out, AdhocSpeculation &sp, int &rc) {
  if (!sp.active) return true;
  sp.active = false;
  const uint64_t h = sp.hash.get();
  if (h == ctx->adhoc_hash) return true;
  ctx->adhoc_valid = false;
  rc = adhoc_reference(ctx, y, ny, out, &h);
  return false;
}

int upload_queries(mi355_sw_ctx *ctx, QueryBatch &q, size_t n, const char *const *xs, const size_t *nxs) {
  HostTrace trace_("upload_queries");
  q.nq = n;
  q.len.resize(n);
  q.off.resize(n);
  size_t mx = 0, tot = 0;
  for (size_t k = 0; k < n; ++k) {
    if (nxs[k] > 0x3fffffff) return fail(ctx, MI355_SW_EINVAL, "query too long");
    q.len[k] = (int32_t)nxs[k];
    q.off[k] = (int64_t)tot;
    tot += (nxs[k] + 15) / 16 * 16;
    mx = std::max(mx, nxs[k]);
  }
  q.maxlen = (int)mx;
  q.order.resize(n);
  for (size_t k = 0; k < n; ++k) q.order[k] = (int32_t)k;
  std::stable_sort(q.order.begin(), q.order.end(), [&](int32_t a, int32_t b) { return q.len[a] < q.len[b]; });
  std::vector<uint8_t> host(tot + 16, 0);
  for (size_t k = 0; k < n; ++k) memcpy(&host[(size_t)q.off[k]], xs[k], nxs[k]);
  if (q.bytes.ensure(host.size()) || q.lens.ensure(n * 4 + 16) || q.offs.ensure(n * 8 + 16) || q.sel.ensure(n * 4 + 16))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(queries) failed");
  HIPCHK(ctx, hipMemcpyAsync(q.bytes.p, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(q.lens.p, q.len.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(q.offs.p, q.off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(q.sel.p, q.order.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- what the packed 16-bit score kernel covers --------------------------------------------
// Score table shared by every launch of a call: which (params, reference alphabet) the packed 16-bit
// kernel can represent exactly.
struct ScoreTable {
  bool ok = false;            // some score-kernel instance can represent (params, alphabet) exactly
  bool integral = false;      // the packed 16-bit instances can
  std::string why;
  int gap = 0, smax = 0;      // packed instances
  float gapf = 0, smaxf = 0;  // float32 instance
  std::vector<int16_t> stab;  // [256][ncodes]
  std::vector<float> ftab;    // [256][ncodes]
};

// A run of length-sorted queries swept by one kernel instance.
struct Bucket {
  int first = 0, count = 0;   // positions in QueryBatch::order
  int maxlen = 0;
  int R = 0;
  int SL = 16;                // lanes per tile: 16, or 8 where 8*R rows fit the reads more tightly
  int sem = kSemI16;          // kernel instance: kSemI16 / kSemU8 packed pairs, kSemF32 one query per slot
  bool strips = false;        // queries longer than one 512-row strip
  bool twin = false;          // lone long query: two tiles of it per packed register (sw_score_kernel TWIN)
  int64_t warm = 0;           // exactness margin in columns
  bool fast = false;          // swept by the score kernel (else whole-matrix exact path)
  int64_t chunk_len = 0;      // own columns per tile
  int64_t sub_len = 0;        // granularity at which tile maxima are reported (= what locate re-runs)
};

int pick_R(int maxlen) {
  static const int rs[] = {2, 4, 6, 8, 10, 12, 16, 20, 24, 32};
  const int need = (maxlen + 15) / 16;
  for (int r : rs) if (r >= need) return r;
  return 0;
}

// 8-lane tiles: instances for the common short-read lengths (<= 56, 80, 104, 128, 152, 208, 256 rows)
int pick_R8(int maxlen) {
  static const int rs[] = {7, 10, 13, 16, 19, 26, 32};
  const int need = (maxlen + 7) / 8;
  for (int r : rs) if (r >= need) return r;
  return 0;
}

// (SL, R) with the fewest padded rows; ties go to 8 lanes (fewer per-step overhead ops per cell)
void pick_shape(int len, int &SL, int &R) {
  SL = 16; R = len < 1 ? 2 : pick_R(len);
  const int r8 = len < 36 ? 0 : pick_R8(len);
  if (r8 && 8 * r8 <= 16 * R) { SL = 8; R = r8; }
  if (const char *e = std::getenv("MI355_SW_SLOT")) { if (std::atoi(e) == 16) { SL = 16; R = len < 1 ? 2 : pick_R(len); } }   // tuning aid
}

ScoreTable plan_table(const RefData &ref, const mi355_sw_params &p) {
  ScoreTable f;
  const int nc = ref.ncodes;
  f.stab.assign((size_t)256 * nc, (int16_t)kPadScore);
  if (p.semantics == MI355_SW_U8SAT) {
    const U8Params u = u8_params(p);
    if (u.G < 1) { f.why = "gap penalty saturates to 0: no finite warm-up margin"; return f; }
    f.ftab.assign((size_t)256 * nc, kPadScoreF);
    for (int a = 0; a < 256; ++a)
      for (int c = 0; c < nc - 1; ++c) {
        const int v = (uint8_t)a == ref.byte_of[c] ? u.M : -u.X;
        f.stab[(size_t)a * nc + c] = (int16_t)v;
        f.ftab[(size_t)a * nc + c] = (float)v;
      }
    f.gap = u.G; f.smax = u.M;
    f.gapf = (float)u.G; f.smaxf = (float)u.M;
    f.integral = true;
  } else {
    const float g = p.gap;
    if (!(g > 0.0f) || !std::isfinite(g)) { f.why = "gap penalty is not positive: no finite warm-up margin"; return f; }
    f.ftab.assign((size_t)256 * nc, kPadScoreF);
    bool integral = g >= 1.0f && g == std::floor(g) && g <= 8000;
    float smaxf = 0;
    for (int a = 0; a < 256; ++a)
      for (int c = 0; c < nc - 1; ++c) {
        const float s = lut_or(p, (uint8_t)a, ref.byte_of[c]);
        if (!std::isfinite(s) || std::fabs(s) > 1e6f) { f.why = "substitution score out of range"; return f; }
        f.ftab[(size_t)a * nc + c] = s;
        smaxf = std::max(smaxf, s);
        if (s != std::floor(s) || std::fabs(s) > 8000) integral = false;
        else f.stab[(size_t)a * nc + c] = (int16_t)s;
      }
    f.gapf = g; f.smaxf = smaxf;
    f.integral = integral;
    if (integral) { f.gap = (int)g; f.smax = (int)smaxf; }
  }
  f.ok = true;
  return f;
}

size_t profile_lds_bytes(int ncodes, int R, int SL = 16, bool twin = false) {
  return (size_t)ncodes * (size_t)std::max(16, SL) * lane_stride(twin ? R / 2 : R) * 4;
}

// Length classes of the batch: one bucket per kernel instance (R), plus one strip-mined bucket.
std::vector<Bucket> make_buckets(const RefData &ref, const QueryBatch &q, const ScoreTable &t, const mi355_sw_params &p, int64_t n) {
  std::vector<Bucket> out;
  // queries beyond 512 rows: whole-wavefront tiles (64 lanes x R rows: one strip up to 2048 rows, 2048-row
  // strips beyond) when the 64-position profile fits LDS, else 16-lane tiles in 512-row strips
  const bool wide_ok = profile_lds_bytes(ref.ncodes, 32, 64) <= kProfileLdsMax && std::getenv("MI355_SW_NO_WIDE") == nullptr;
  for (size_t pos = 0; pos < q.nq; ++pos) {
    const int len = q.len[q.order[pos]];
    bool strips = false;
    int SL = 16, R = 32;
    if (len <= kMaxRowsFast) pick_shape(len, SL, R);
    else if (wide_ok) { SL = 64; R = len <= 1024 ? 16 : 32; strips = len > 2048; }
    else strips = true;
    if (out.empty() || out.back().R != R || out.back().SL != SL || out.back().strips != strips) {
      Bucket b;
      b.first = (int)pos; b.R = R; b.SL = SL; b.strips = strips;
      out.push_back(b);
    }
    out.back().count++;
    out.back().maxlen = std::max(out.back().maxlen, len);
  }
  for (Bucket &b : out) {
    const bool twin_ok = b.count == 1 && b.SL == 64 && std::getenv("MI355_SW_NO_TWIN") == nullptr;
    if (p.semantics == MI355_SW_U8SAT) {
      // lone query: two of its tiles per packed register on whole-wavefront tiles, else one query per register
      b.twin = twin_ok;
      b.sem = b.count == 1 && !b.twin ? kSemF32U8 : kSemU8;
    } else {
      // packed 16-bit cells when scores are small integers and the score bound fits; float32 cells otherwise
      const bool fits = t.integral && (int64_t)t.smax * std::min<int64_t>(b.maxlen, std::max<int64_t>(n, 1)) + t.smax <= 32000;
      b.sem = fits ? kSemI16 : kSemF32;
      // a lone query would fill both halves of every packed register with itself; the float32 instance
      // (one query per slot, exact for integer scores below 2^24) sweeps it ~1.5x faster
      if (b.count == 1 && b.sem == kSemI16) {
        if (twin_ok) b.twin = true;                               // long lone query: two of its tiles per register
        else if ((double)t.smax * b.maxlen < 1.6e7) b.sem = kSemF32;
      }
    }
    const double smax = sem_is_float(b.sem) ? (double)t.smaxf : (double)t.smax;
    const double gap = sem_is_float(b.sem) ? (double)t.gapf : (double)t.gap;
    if (smax <= 0 || gap <= 0) b.warm = 0;
    else b.warm = (int64_t)b.maxlen + (int64_t)std::ceil(smax * b.maxlen / gap);
    b.warm = (b.warm + 63) / 64 * 64;
  }
  return out;
}

// May this bucket's queries be swept by the score kernel over a reference range of n columns?
bool bucket_fast_ok(const RefData &ref, const ScoreTable &t, const Bucket &b, int64_t n, const mi355_sw_params &p) {
  if (!t.ok || n < 1 || b.maxlen < 1) return false;
  if (profile_lds_bytes(ref.ncodes, b.R, b.SL) > kProfileLdsMax) return false;  // alphabet too large for this shape
  // the uint8 engine's storage order is only bounded to a few tiles when the reference is the longer side;
  // shorter references take the whole-matrix path (which also holds the |x| == |y| quirk)
  if (p.semantics == MI355_SW_U8SAT && n <= (int64_t)b.maxlen + 1) return false;
  // float32 cells stay exact integers only below 2^24
  if (b.sem == kSemF32 && t.integral && (double)t.smax * (double)std::min<int64_t>(b.maxlen, n) > 1.6e7) return false;
  // the warm-up margin must stay a small fraction of the range (tiny gap penalties)
  if (b.warm > 64 * (int64_t)b.maxlen + 1024) return false;
  // strip-mining re-streams the range once per 512 rows: only worth it on long ranges
  if (b.strips && n < 4096) return false;
  // short references (UniProt shape: many sequences against one 144-residue query): one whole-matrix
  // pass of the exact kernel does score + argmax + decisions at once; the tile machinery would idle
  if (n < 1024) return false;
  return true;
}

template <class K>
void launch_score(K kernel, dim3 grid, size_t shmem, hipStream_t st, const ScoreArgs &a) {
  // large alphabets x many rows per lane need more than the default 64 KiB of dynamic LDS
  if (shmem > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(kernel, grid, dim3(256), shmem, st, a);
}

template <int SEM>
int launch_score_twin(int R, bool strips, dim3 grid, size_t shmem, hipStream_t st, const ScoreArgs &a) {
  if (strips) {
    if (R != 32) return -1;
    launch_score(sw_score_kernel<32, SEM, true, 64, true>, grid, shmem, st, a);
    return 0;
  }
  if (R == 16) launch_score(sw_score_kernel<16, SEM, false, 64, true>, grid, shmem, st, a);
  else if (R == 32) launch_score(sw_score_kernel<32, SEM, false, 64, true>, grid, shmem, st, a);
  else return -1;
  return 0;
}

template <int SEM>
int launch_score_R(int R, int SL, bool strips, dim3 grid, size_t shmem, hipStream_t st, const ScoreArgs &a) {
  if (strips) {
    if (R != 32) return -1;
    if (SL == 64) launch_score(sw_score_kernel<32, SEM, true, 64>, grid, shmem, st, a);
    else if (SL == 16) launch_score(sw_score_kernel<32, SEM, true, 16>, grid, shmem, st, a);
    else return -1;
    return 0;
  }
  if (SL == 64) {
    if (R == 16) launch_score(sw_score_kernel<16, SEM, false, 64>, grid, shmem, st, a);
    else if (R == 32) launch_score(sw_score_kernel<32, SEM, false, 64>, grid, shmem, st, a);
    else return -1;
    return 0;
  }
  if (SL == 8) {
    switch (R) {
#define CASE_R8(r) case r: launch_score(sw_score_kernel<r, SEM, false, 8>, grid, shmem, st, a); return 0;
      CASE_R8(7) CASE_R8(10) CASE_R8(13) CASE_R8(16) CASE_R8(19) CASE_R8(26) CASE_R8(32)
#undef CASE_R8
    }
    return -1;
  }
  switch (R) {
#define CASE_R(r) case r: launch_score(sw_score_kernel<r, SEM, false>, grid, shmem, st, a); return 0;
    CASE_R(2) CASE_R(4) CASE_R(6) CASE_R(8) CASE_R(10) CASE_R(12) CASE_R(16) CASE_R(20) CASE_R(24) CASE_R(32)
#undef CASE_R
  }
  return -1;
}

int64_t pick_chunk_len(int64_t max_range_len, size_t npairs, int64_t warm, int SL = 16, bool twin = false, int maxlen = 0) {
  int64_t cl = 65536;
  while (cl < 8 * warm) cl *= 2;                 // long queries: keep the warm-up redundancy bounded
  // fill the chip: 256 CUs x 32 waves x 4 slots; shrink tiles while they stay >> warm-up
  while (cl > 2048 && cl / 2 >= 4 * warm &&
         (double)npairs * (double)((max_range_len + cl - 1) / cl) < 65536.0) cl /= 2;
  // few tiles (one long query): filling the SIMDs beats the warm-up redundancy down to cl == warm
  // (measured, 10 kbp x 250 Mbp: 1.17 s at 131 k columns, 0.58 s at 32 k; profiles/r01_config5*.log)
  const double few = (SL == 64 ? 1536.0 : 8192.0) * (twin ? 2.0 : 1.0);   // a 64-lane tile is a wavefront of its own (two tiles with twin)
  while (cl / 2 >= std::max<int64_t>(warm, 2048) &&
         (double)npairs * (double)((max_range_len + cl - 1) / cl) < few) cl /= 2;
  // tiny problems (one read against a short reference): the call's latency is one tile's sweep and the chip is
  // mostly idle, so tiles shrink until every CU has a workgroup (down to one sub-chunk: >= 256 columns, >= |x|)
  int64_t floor_cl = 256;
  while (floor_cl < maxlen) floor_cl *= 2;
  const double per_wg = 256.0 / SL * (twin ? 2.0 : 1.0);
  while (cl / 2 >= floor_cl && (double)npairs * (double)((max_range_len + cl - 1) / cl) / per_wg < 256.0) cl /= 2;
  if (const char *e = std::getenv("MI355_SW_CHUNK")) { const long v = std::atol(e); if (v >= 256) cl = v / 64 * 64; }   // tuning aid
  return cl;
}

// Uploads what every score launch of a call shares and clears the keys.
int score_begin(mi355_sw_ctx *ctx, const QueryBatch &q, const std::vector<Range> &ranges, const ScoreTable &t) {
  const size_t nq = q.nq, nr = ranges.size();
  if (nr > 32768) return fail(ctx, MI355_SW_ENOTSUP, "more than 32768 ranges per launch");
  // the previous call's copies out of these host vectors have completed: every call ends synchronised
  std::vector<int64_t> &rl = ctx->h_ranges;
  rl.resize(2 * nr);
  for (size_t k = 0; k < nr; ++k) { rl[k] = ranges[k].lo; rl[nr + k] = ranges[k].hi; }
  const void *stab_was = ctx->stab.p, *ftab_was = ctx->ftab.p;
  if (ctx->ranges.ensure(rl.size() * 8) || ctx->keys.ensure(nq * nr * 8) || ctx->stab.ensure(t.stab.size() * 2) ||
      ctx->ftab.ensure(t.ftab.size() * 4 + 16))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
  HIPCHK(ctx, hipMemcpyAsync(ctx->ranges.p, rl.data(), rl.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->stab.p != stab_was || ctx->h_stab != t.stab) {
    ctx->h_stab = t.stab;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stab.p, ctx->h_stab.data(), ctx->h_stab.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  }
  if (!t.ftab.empty() && (ctx->ftab.p != ftab_was || ctx->h_ftab != t.ftab)) {
    ctx->h_ftab = t.ftab;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ftab.p, ctx->h_ftab.data(), ctx->h_ftab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->keys.p, 0, nq * nr * 8, ctx->stream));
  return 0;
}

// One score-kernel launch: bucket b over all ranges.  Device time is added to ctx->timings[0].
int score_launch(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<Range> &ranges,
                 const mi355_sw_params &p, const ScoreTable &t, Bucket &b) {
  HostTrace trace_("score_launch");
  const size_t nr = ranges.size();
  int64_t maxlen = 0;
  for (auto &r : ranges) maxlen = std::max(maxlen, r.hi - r.lo);
  const size_t npairs = (sem_is_float(b.sem) || b.twin) ? (size_t)b.count : ((size_t)b.count + 1) / 2;   // queries per workgroup: 1 or 2
  b.chunk_len = pick_chunk_len(maxlen, npairs * nr, b.warm, b.SL, b.twin, b.maxlen);
  // report maxima per sub-chunk of >= 256 columns (>= query length, so that the uint8 storage order stays
  // within two neighbouring sub-chunks): that is what locate re-runs; the strip-mined instance reports per tile
  b.sub_len = 256;
  while (b.sub_len < b.maxlen) b.sub_len *= 2;
  if (b.strips) while (b.chunk_len / b.sub_len > 64) b.sub_len *= 2;      // the strip-mined instances keep <= 64 sub-chunk maxima in LDS
  if (b.sub_len > b.chunk_len || b.chunk_len % b.sub_len != 0) b.sub_len = b.chunk_len;
  const int64_t cpr = (maxlen + b.chunk_len - 1) / b.chunk_len;
  const int nslot = 256 / b.SL;                                     // tiles (twin: tile pairs) per workgroup
  const int64_t cgroups = ((b.twin ? (cpr + 1) / 2 : cpr) + nslot - 1) / nslot;
  if ((double)npairs * (double)cgroups > 2.0e9) return fail(ctx, MI355_SW_ENOTSUP, "grid too large");

  ScoreArgs a;
  a.refcodes = ref.codes.as<uint8_t>();
  a.ref_len = (int64_t)ref.n;
  a.range_lo = ctx->ranges.as<int64_t>();
  a.range_hi = ctx->ranges.as<int64_t>() + nr;
  a.chunk_len = b.chunk_len;
  a.sub_len = b.sub_len;
  a.warm = (cpr == 1) ? 0 : b.warm;              // a single tile per range starts at the range's own border
  a.chunks_per_range = (int)cpr;
  a.qbytes = q.bytes.as<uint8_t>();
  a.qoff = q.offs.as<int64_t>();
  a.qlen = q.lens.as<int32_t>();
  a.qsel = q.sel.as<int32_t>();
  a.qfirst = b.first;
  a.qcount = b.count;
  a.nq = (int)q.nq;
  a.stab = sem_is_float(b.sem) ? ctx->ftab.p : ctx->stab.p;
  a.ncodes = ref.ncodes;
  if (sem_is_float(b.sem)) memcpy(&a.gap2, &t.gapf, 4);
  else a.gap2 = (uint32_t)t.gap * 0x00010001u;
  a.clamp2 = 255u * 0x00010001u;
  a.keys = ctx->keys.as<unsigned long long>();

  const int nqw = (sem_is_float(b.sem) || b.twin) ? 1 : 2;          // queries per workgroup
  // keep single launches to a few seconds: split the bucket's pairs over several launches
  double range_cols = 0;
  for (auto &r : ranges) range_cols += (double)(r.hi - r.lo);
  const double cells_per_pair = (double)nqw * std::max(1, b.maxlen) * std::max(1.0, range_cols);
  const size_t pairs_per_launch = (size_t)std::max(1.0, std::min((double)npairs, 5.0e13 / cells_per_pair));
  for (size_t p0 = 0; p0 < npairs; p0 += pairs_per_launch) {
  const size_t pn = std::min(pairs_per_launch, npairs - p0);
  a.qfirst = b.first + (int)(p0 * nqw);
  a.qcount = std::min(b.count - (int)(p0 * nqw), (int)(pn * nqw));
  size_t shmem = profile_lds_bytes(ref.ncodes, b.R, b.SL, b.twin) + (size_t)(b.twin ? 2 : 1) * nslot * codebuf_bytes(b.SL);
  dim3 grid((unsigned)(pn * cgroups), (unsigned)nr);
  a.brow = nullptr;
  a.brow_stride = 0;
  if (b.strips) {
    const int64_t nseg = (a.warm + b.chunk_len + b.SL + kSeg - 1) / kSeg;
    a.brow_stride = (nseg + 3) * kSeg + kBrowFront + 32;
    const size_t slots = (size_t)grid.x * grid.y * nslot;
    const size_t bytes = slots * 2 * (size_t)a.brow_stride * 4;
    if (bytes > ((size_t)64 << 30)) return fail(ctx, MI355_SW_ENOTSUP, "strip-mined sweep needs more than 64 GiB of boundary scratch");
    if (ctx->brow.ensure(bytes)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(strip boundary rows) failed");
    HIPCHK(ctx, hipMemsetAsync(ctx->brow.p, 0, bytes, ctx->stream));
    a.brow = ctx->brow.as<uint32_t>();
    shmem += (size_t)2 * nslot * kSeg * 4 + (size_t)nslot * 64 * 4;   // boundary windows + per-sub-chunk maxima
  }
  if (ctx->score_ev.size() < ctx->score_ev_used + 2) {
    for (int e = 0; e < 2; ++e) { hipEvent_t ev; HIPCHK(ctx, hipEventCreate(&ev)); ctx->score_ev.push_back(ev); }
  }
  HIPCHK(ctx, hipEventRecord(ctx->score_ev[ctx->score_ev_used], ctx->stream));
  int rc = b.twin ? (b.sem == kSemU8 ? launch_score_twin<kSemU8>(b.R, b.strips, grid, shmem, ctx->stream, a)
                                     : launch_score_twin<kSemI16>(b.R, b.strips, grid, shmem, ctx->stream, a))
           : b.sem == kSemU8 ? launch_score_R<kSemU8>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
           : b.sem == kSemF32U8 ? launch_score_R<kSemF32U8>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
           : b.sem == kSemF32 ? launch_score_R<kSemF32>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
                              : launch_score_R<kSemI16>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a);
  if (rc) return fail(ctx, MI355_SW_ENOTSUP, "no score kernel instance for this R");
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipEventRecord(ctx->score_ev[ctx->score_ev_used + 1], ctx->stream));
  ctx->score_ev_used += 2;                                // read by score_fetch, after the launches have drained
  ctx->timings[4] += 1;
  }
  double cells = 0;
  for (int k = 0; k < b.count; ++k)
    for (auto &r : ranges) cells += (double)q.len[q.order[b.first + k]] * (double)(r.hi - r.lo);
  ctx->timings[5] += cells;
  return 0;
}

int score_fetch(mi355_sw_ctx *ctx, size_t count, std::vector<unsigned long long> &keys) {
  keys.resize(count);
  HIPCHK(ctx, hipMemcpyAsync(keys.data(), ctx->keys.p, count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t e = 0; e + 1 < ctx->score_ev_used; e += 2) {     // device time of the score launches
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->score_ev[e], ctx->score_ev[e + 1]));
    ctx->timings[0] += (double)ms * 1000.0;
  }
  ctx->score_ev_used = 0;
  return 0;
}

// ---- exact kernel launches ------------------------------------------------------------------
struct ExactJob {
  int q;                 // query index in the batch
  int64_t ylo;           // window start (absolute reference index of local column 1)
  int32_t nw;
  int64_t col_offset;    // true (range-relative) column = col_offset + jl
  int64_t full_n;
  int32_t own_lo;
  int32_t quirk;
  float target;
  bool want_dirs;
  // results
  float best = -1;
  int64_t ci = 0, cj = 0;
  size_t dirs_off = 0;
};

// bytes of the diagonal-major decision array of an (m x nw) window (sw_exact_kernel.h)
size_t dirs_bytes(int64_t m, int64_t nw) { return (size_t)(m + nw + 1) * (size_t)std::max<int64_t>(1, std::min(m, nw)) + 16; }

size_t exact_lds_bytes(int m, int nw) { return (size_t)3 * (std::min(m, nw) + 2) * 4 + (size_t)m + 16; }

ExactScoring make_scoring(mi355_sw_ctx *ctx, const mi355_sw_params &p, bool &lut_uploaded, int &rc) {
  ExactScoring s;
  rc = 0;
  s.lut = nullptr;
  if (p.lut && p.semantics == MI355_SW_F32) {
    if (!lut_uploaded) {
      if (ctx->lut.ensure(65536 * 4)) { rc = MI355_SW_ENOMEM; return s; }
      if (hipMemcpyAsync(ctx->lut.p, p.lut, 65536 * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = MI355_SW_ENODEV; return s; }
      lut_uploaded = true;
    }
    s.lut = ctx->lut.as<float>();
  }
  s.match = p.match; s.mismatch = p.mismatch; s.gap = p.gap;
  const U8Params u = u8_params(p);
  s.u8M = u.M; s.u8X = u.X; s.u8G = u.G;
  return s;
}

// Runs jobs[lo,hi) in one launch.  Decisions (if wanted) land in ctx->dirs at job.dirs_off.
int run_exact(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const mi355_sw_params &p,
              std::vector<ExactJob> &jobs, size_t lo, size_t hi, float *hout /* device or null, single job */) {
  const size_t n = hi - lo;
  if (n == 0) return 0;
  size_t dirs_total = 0, lds = 0;
  for (size_t k = lo; k < hi; ++k) {
    ExactJob &j = jobs[k];
    lds = std::max(lds, exact_lds_bytes(q.len[j.q], j.nw));
    if (j.want_dirs) { j.dirs_off = dirs_total; dirs_total += dirs_bytes(q.len[j.q], j.nw); dirs_total = (dirs_total + 15) & ~(size_t)15; }
  }
  if (lds > kExactLdsMax) return fail(ctx, MI355_SW_ENOTSUP, "anti-diagonal longer than the exact kernel's LDS window");
  if (ctx->probs.ensure(n * sizeof(ExactProblem)) || ctx->outs_f.ensure(n * 4) || ctx->outs_i.ensure(n * 16) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(exact scratch) failed");
  // few problems with long diagonals: sixteen wavefronts per problem; the others one wavefront each
  std::vector<size_t> slot(n);                     // position of job lo + k in the problem array: wide ones first
  size_t nwide = 0, lds_narrow = 0;
  {
    std::vector<size_t> wide_k, narrow_k;
    for (size_t k = 0; k < n; ++k)
      (std::min<int>(q.len[jobs[lo + k].q], jobs[lo + k].nw) >= 1024 ? wide_k : narrow_k).push_back(k);
    if (wide_k.size() > 2048) { narrow_k.insert(narrow_k.end(), wide_k.begin(), wide_k.end()); wide_k.clear(); }
    nwide = wide_k.size();
    lds_narrow = 0;
    for (size_t k : narrow_k) lds_narrow = std::max(lds_narrow, exact_lds_bytes(q.len[jobs[lo + k].q], jobs[lo + k].nw));
    for (size_t t = 0; t < wide_k.size(); ++t) slot[wide_k[t]] = t;
    for (size_t t = 0; t < narrow_k.size(); ++t) slot[narrow_k[t]] = nwide + t;
  }
  std::vector<ExactProblem> pr(n);
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    ExactProblem &e = pr[slot[k]];
    e.x = q.bytes.as<uint8_t>() + q.off[j.q];
    e.y = ref.bytes.as<uint8_t>() + j.ylo;
    e.m = q.len[j.q];
    e.nw = j.nw;
    e.col_offset = j.col_offset;
    e.full_n = j.full_n;
    e.own_lo = j.own_lo;
    e.square_quirk = j.quirk;
    e.target = j.target;
    e.dirs = j.want_dirs ? ctx->dirs.as<uint8_t>() + j.dirs_off : nullptr;
    e.hout = hout;
    e.best = ctx->outs_f.as<float>() + slot[k];
    e.cell = ctx->outs_i.as<int64_t>() + 2 * slot[k];
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->probs.p, pr.data(), n * sizeof(ExactProblem), hipMemcpyHostToDevice, ctx->stream));
  bool lut_up = false;
  int rc = 0;
  const ExactScoring sc = make_scoring(ctx, p, lut_up, rc);
  if (rc) return fail(ctx, rc, "scoring table upload failed");
  const ExactProblem *dp = ctx->probs.as<ExactProblem>();
  const size_t nnarrow = n - nwide;
  if (p.semantics == MI355_SW_U8SAT) {
    if (nwide) hipLaunchKernelGGL((sw_exact_kernel<1, 1024>), dim3((unsigned)nwide), dim3(1024), lds, ctx->stream, dp, sc);
    if (nnarrow) hipLaunchKernelGGL((sw_exact_kernel<1, 64>), dim3((unsigned)nnarrow), dim3(64), lds_narrow, ctx->stream, dp + nwide, sc);
  } else {
    if (nwide) hipLaunchKernelGGL((sw_exact_kernel<0, 1024>), dim3((unsigned)nwide), dim3(1024), lds, ctx->stream, dp, sc);
    if (nnarrow) hipLaunchKernelGGL((sw_exact_kernel<0, 64>), dim3((unsigned)nnarrow), dim3(64), lds_narrow, ctx->stream, dp + nwide, sc);
  }
  HIPCHK(ctx, hipGetLastError());
  std::vector<float> bf(n);
  std::vector<int64_t> ci(2 * n);
  HIPCHK(ctx, hipMemcpyAsync(bf.data(), ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ci.data(), ctx->outs_i.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k) { jobs[lo + k].best = bf[slot[k]]; jobs[lo + k].ci = ci[2 * slot[k]]; jobs[lo + k].cj = ci[2 * slot[k] + 1]; }
  return 0;
}

struct TraceOut {
  std::string cx, cy;
  uint32_t pos = 0;
};

// Walk over decisions of jobs[lo,hi) (all with want_dirs), starting at (start_i, local nw...).
int run_walk(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<ExactJob> &jobs,
             size_t lo, size_t hi, const std::vector<std::pair<int32_t, int32_t>> &starts,
             const std::vector<int32_t> &exact_lo, std::vector<TraceOut> &outs, std::vector<int> &status,
             float need_slope = 0.0f) {
  const size_t n = hi - lo;
  if (n == 0) return 0;
  std::vector<WalkProblem> wp(n);
  std::vector<size_t> coff(n);
  size_t ctot = 0;
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    const int cap = q.len[j.q] + j.nw + 2;
    coff[k] = ctot;
    ctot += 2 * (size_t)cap;
  }
  if (ctx->cons.ensure(ctot + 16) || ctx->walkp.ensure(n * sizeof(WalkProblem) + n * 24))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(walk scratch) failed");
  int64_t *wout = reinterpret_cast<int64_t *>(ctx->walkp.as<uint8_t>() + n * sizeof(WalkProblem));
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    WalkProblem &w = wp[k];
    const int cap = q.len[j.q] + j.nw + 2;
    w.x = q.bytes.as<uint8_t>() + q.off[j.q];
    w.y = ref.bytes.as<uint8_t>() + j.ylo;
    w.dirs = ctx->dirs.as<uint8_t>() + j.dirs_off;
    w.m = q.len[j.q]; w.nw = j.nw;
    w.start_i = starts[k].first; w.start_jl = starts[k].second;
    w.exact_lo = exact_lo[k];
    w.need_slope = need_slope;
    w.col_offset = j.col_offset;
    w.cons_x = ctx->cons.as<char>() + coff[k];
    w.cons_y = w.cons_x + cap;
    w.cap = cap;
    w.out = wout + 3 * k;
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->walkp.p, wp.data(), n * sizeof(WalkProblem), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(sw_walk_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, ctx->walkp.as<WalkProblem>(), (int)n);
  HIPCHK(ctx, hipGetLastError());
  std::vector<int64_t> wo(3 * n);
  std::vector<char> cons(ctot);
  HIPCHK(ctx, hipMemcpyAsync(wo.data(), wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(cons.data(), ctx->cons.p, ctot, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  outs.resize(n); status.resize(n);
  for (size_t k = 0; k < n; ++k) {
    const ExactJob &j = jobs[lo + k];
    const int cap = q.len[j.q] + j.nw + 2;
    status[k] = (int)wo[3 * k + 2];
    const size_t len = (size_t)wo[3 * k];
    outs[k].cx.assign(cons.data() + coff[k], len);
    outs[k].cy.assign(cons.data() + coff[k] + cap, len);
    outs[k].pos = (uint32_t)wo[3 * k + 1];
  }
  return 0;
}

// ---- wavefront exact kernel (sw_wave_kernel.h): small problems, identity scoring ---------------
constexpr int kWaveMaxLanesSide = 512;

// Scoring the wave kernel evaluates: no table, and penalties that strictly lower a path (so that padding cells
// can never reach the maximum).
bool wave_scoring_ok(const mi355_sw_params &p) {
  if (p.semantics == MI355_SW_U8SAT) { const U8Params u = u8_params(p); return u.M > 0; }
  return p.lut == nullptr && p.match > 0 && p.mismatch < 0 && p.gap > 0 && std::isfinite(p.match) &&
         std::isfinite(p.mismatch) && std::isfinite(p.gap);
}

// rows per lane of the wave kernel instance that covers `na` cells on the lane side, and its decision bytes
int wave_R(int na) { return na <= 160 ? 10 : (na <= 320 ? 20 : 32); }
size_t wave_dirs_bytes(int64_t nb, int R) { return (size_t)nb * 16 * (size_t)((R + 15) / 16) * 4 + 64; }

struct WaveJob {
  int q;                  // query index
  int orient;             // 0: lanes = rows of x, stream = columns of y; 1: lanes = columns of y, stream = rows of x
  int64_t s_lo;           // stream window start (0-based, range-relative), nb positions
  int32_t nb;
  bool track, dirs;
  bool keyed = false;     // wave kernel, track: first cell equal to target in storage order (else: first maximum)
  float target = 0;       // strip kernel / keyed: only cells equal to target compete ...
  int32_t own_lo = 0;     // ... at stream positions >= own_lo (0-based)
  // results
  float best = 0;
  int64_t ci = 0, cj = 0;
  size_t dirs_off = 0;
};

template <int R, int ORIENT, bool U8>
void launch_wave_flags(bool track, bool dirs, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (track && dirs) hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, U8, true, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else if (track) hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, U8, true, false>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else hipLaunchKernelGGL((sw_wave_kernel<R, ORIENT, U8, false, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
}

template <int R>
void launch_wave_keyed(bool u8, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (u8) hipLaunchKernelGGL((sw_wave_kernel<R, 0, true, true, false, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
  else hipLaunchKernelGGL((sw_wave_kernel<R, 0, false, true, false, true>), dim3(blocks), dim3(256), 0, st, pr, n, sc);
}

template <int R>
void launch_wave_R(int orient, bool u8, bool track, bool dirs, unsigned blocks, hipStream_t st, const WaveProblem *pr, int n, const WaveScoring &sc) {
  if (orient == 0) { if (u8) launch_wave_flags<R, 0, true>(track, dirs, blocks, st, pr, n, sc); else launch_wave_flags<R, 0, false>(track, dirs, blocks, st, pr, n, sc); }
  else { if (u8) launch_wave_flags<R, 1, true>(track, dirs, blocks, st, pr, n, sc); else launch_wave_flags<R, 1, false>(track, dirs, blocks, st, pr, n, sc); }
}

// One launch: all jobs share orientation and flags; lanes side <= 512.
int run_wave(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
             std::vector<WaveJob> &jobs) {
  HostTrace trace_("run_wave");
  const size_t n = jobs.size();
  if (n == 0) return 0;
  const int orient = jobs[0].orient;
  const bool track = jobs[0].track, dirs = jobs[0].dirs;
  const int64_t nref = rg.hi - rg.lo;
  size_t dirs_total = 0;
  int maxna = 0;
  for (WaveJob &j : jobs) maxna = std::max(maxna, orient == 0 ? q.len[j.q] : (int)nref);
  if (maxna > kWaveMaxLanesSide) return fail(ctx, MI355_SW_EINVAL, "internal: wave kernel side too long");
  const int R = wave_R(maxna);
  for (WaveJob &j : jobs)
    if (dirs) { j.dirs_off = dirs_total; dirs_total += wave_dirs_bytes(j.nb, R); }
  if (ctx->wprobs.ensure(n * sizeof(WaveProblem)) || ctx->outs_f.ensure(n * 4) || ctx->outs_i.ensure(n * 16) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(wave scratch) failed");
  std::vector<WaveProblem> pr(n);
  for (size_t k = 0; k < n; ++k) {
    const WaveJob &j = jobs[k];
    WaveProblem &w = pr[k];
    const uint8_t *xq = q.bytes.as<uint8_t>() + q.off[j.q];
    const uint8_t *yr = ref.bytes.as<uint8_t>() + rg.lo;
    if (orient == 0) { w.a = xq; w.na = q.len[j.q]; w.b = yr + j.s_lo; }
    else { w.a = yr; w.na = (int32_t)nref; w.b = xq + j.s_lo; }
    w.nb = j.nb;
    w.b_offset = j.s_lo;
    w.dirs = dirs ? reinterpret_cast<uint32_t *>(ctx->dirs.as<uint8_t>() + j.dirs_off) : nullptr;
    w.best = ctx->outs_f.as<float>() + k;
    w.cell = ctx->outs_i.as<int64_t>() + 2 * k;
    w.target = j.target; w.own_lo = j.own_lo; w.full_n = nref;
  }
  const bool keyed = jobs[0].keyed;
  HIPCHK(ctx, hipMemcpyAsync(ctx->wprobs.p, pr.data(), n * sizeof(WaveProblem), hipMemcpyHostToDevice, ctx->stream));
  WaveScoring sc;
  sc.match = p.match; sc.mismatch = p.mismatch; sc.gap = p.gap;
  const U8Params u = u8_params(p);
  sc.u8M = (float)u.M; sc.u8X = (float)u.X; sc.u8G = (float)u.G;
  const bool u8 = p.semantics == MI355_SW_U8SAT;
  const unsigned blocks = (unsigned)((n + 15) / 16);
  const WaveProblem *dp = ctx->wprobs.as<WaveProblem>();
  if (keyed) {
    if (R == 10) launch_wave_keyed<10>(u8, blocks, ctx->stream, dp, (int)n, sc);
    else if (R == 20) launch_wave_keyed<20>(u8, blocks, ctx->stream, dp, (int)n, sc);
    else launch_wave_keyed<32>(u8, blocks, ctx->stream, dp, (int)n, sc);
  }
  else if (R == 10) launch_wave_R<10>(orient, u8, track, dirs, blocks, ctx->stream, dp, (int)n, sc);
  else if (R == 20) launch_wave_R<20>(orient, u8, track, dirs, blocks, ctx->stream, dp, (int)n, sc);
  else launch_wave_R<32>(orient, u8, track, dirs, blocks, ctx->stream, dp, (int)n, sc);
  HIPCHK(ctx, hipGetLastError());
  if (track) {
    std::vector<float> bf(n);
    std::vector<int64_t> ci(2 * n);
    HIPCHK(ctx, hipMemcpyAsync(bf.data(), ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ci.data(), ctx->outs_i.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < n; ++k) { jobs[k].best = bf[k]; jobs[k].ci = ci[2 * k]; jobs[k].cj = ci[2 * k + 1]; }
  }
  return 0;
}

// Rows per lane of the strip kernel instance for a query of `na` rows, and its strips of 64*R rows (sixteen run
// concurrently; longer queries take several rounds).
int strip_R(int na) { return na <= 64 * kStripMaxWaves * 10 ? 10 : 16; }
int strip_count(int na, int R) { return std::max(1, (na + 64 * R - 1) / (64 * R)); }
size_t strip_dirs_bytes(int64_t nb, int nstrips, int R) { return (size_t)nb * 64 * (size_t)nstrips * (size_t)((R + 15) / 16) * 4 + 64; }

template <int R>
void launch_strip(bool u8, bool track, dim3 grid, dim3 block, hipStream_t st, const StripProblem *dp, const WaveScoring &sc,
                  const float *gtab, int ncodes) {
  if (gtab) {                                   // table scoring (float engine): tab[257][ncodes] in dynamic LDS
    const size_t lds = (size_t)257 * ncodes * 4;
    if (track) hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripTrack, true>), grid, block, lds, st, dp, sc, gtab, ncodes);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripDirs, true>), grid, block, lds, st, dp, sc, gtab, ncodes);
    return;
  }
  if (track) {
    if (u8) hipLaunchKernelGGL((sw_strip_kernel<R, true, kStripTrack>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripTrack>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0);
  } else {
    if (u8) hipLaunchKernelGGL((sw_strip_kernel<R, true, kStripDirs>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0);
    else hipLaunchKernelGGL((sw_strip_kernel<R, false, kStripDirs>), grid, block, 0, st, dp, sc, (const float *)nullptr, 0);
  }
}

// Long queries (ORIENT 0 windows) on the pipelined strip kernel, one workgroup per job: traceback decisions
// (jobs[.].dirs) or the first cell equal to jobs[.].target in storage order (jobs[.].track -> ci, cj; ci = 0: none).
// The score table of general (non-identity) float scoring the strip kernel can hold in LDS next to its rings.
bool strip_table_ok(const RefData &ref, const mi355_sw_params &p) {
  return p.semantics == MI355_SW_F32 && (size_t)257 * ref.ncodes * 4 <= 96 * 1024;
}
// Which long queries the strip kernel takes: identity scoring in both engines, any table in the float engine.
bool strip_scoring_ok(const RefData &ref, const mi355_sw_params &p) {
  if (std::getenv("MI355_SW_NO_STRIP") != nullptr) return false;
  return wave_scoring_ok(p) || strip_table_ok(ref, p);
}

int run_strip(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
              std::vector<WaveJob> &jobs, int R) {
  HostTrace trace_("run_strip");
  const bool use_table = !wave_scoring_ok(p);            // ctx->ftab holds plan_table()'s [256][ncodes] (score_begin)
  const size_t n = jobs.size();
  if (n == 0) return 0;
  const bool track = jobs[0].track;
  size_t dirs_total = 0, gtotal = 0;
  int nwmax = 1;
  std::vector<size_t> goff(n, 0);
  for (size_t k = 0; k < n; ++k) {
    WaveJob &j = jobs[k];
    const int ns = strip_count(q.len[j.q], R);
    nwmax = std::max(nwmax, std::min(ns, kStripMaxWaves));
    if (!track) { j.dirs_off = dirs_total; dirs_total += strip_dirs_bytes(j.nb, ns, R); }
    if (ns > kStripMaxWaves) { goff[k] = gtotal; gtotal += 2 * ((size_t)j.nb + 192); }
  }
  if (ctx->wprobs.ensure(n * sizeof(StripProblem)) || ctx->outs_i.ensure(n * 16) || ctx->outs_f.ensure(n * 4) ||
      (dirs_total && ctx->dirs.ensure(dirs_total)) || (gtotal && ctx->brow.ensure(gtotal * 4)))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(strip scratch) failed");
  std::vector<StripProblem> pr(n);
  for (size_t k = 0; k < n; ++k) {
    const WaveJob &j = jobs[k];
    StripProblem &s = pr[k];
    s.a = q.bytes.as<uint8_t>() + q.off[j.q];
    s.na = q.len[j.q];
    s.b = (use_table ? ref.codes.as<uint8_t>() : ref.bytes.as<uint8_t>()) + rg.lo + j.s_lo;
    s.nb = j.nb;
    s.nstrips = strip_count(q.len[j.q], R);
    s.nw = std::min(s.nstrips, kStripMaxWaves);
    s.gbound = s.nstrips > kStripMaxWaves ? ctx->brow.as<float>() + goff[k] : nullptr;
    s.gstride = (int64_t)j.nb + 192;
    s.dirs = track ? nullptr : reinterpret_cast<uint32_t *>(ctx->dirs.as<uint8_t>() + j.dirs_off);
    s.target = j.target;
    s.own_lo = j.own_lo;
    s.col_offset = j.s_lo;
    s.full_n = rg.hi - rg.lo;
    s.cell = ctx->outs_i.as<int64_t>() + 2 * k;
    s.status = ctx->outs_f.as<int32_t>() + k;
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->outs_f.p, 0, n * 4, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->wprobs.p, pr.data(), n * sizeof(StripProblem), hipMemcpyHostToDevice, ctx->stream));
  WaveScoring sc;
  sc.match = p.match; sc.mismatch = p.mismatch; sc.gap = p.gap;
  const U8Params u = u8_params(p);
  sc.u8M = (float)u.M; sc.u8X = (float)u.X; sc.u8G = (float)u.G;
  const bool u8 = p.semantics == MI355_SW_U8SAT;
  const StripProblem *dp = ctx->wprobs.as<StripProblem>();
  const dim3 grid((unsigned)n), block((unsigned)(64 * nwmax));
  const float *gtab = use_table ? ctx->ftab.as<float>() : nullptr;
  if (use_table && (size_t)257 * ref.ncodes * 4 > 48 * 1024) {
    const int lds = 257 * ref.ncodes * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<10, false, kStripTrack, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<10, false, kStripDirs, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<16, false, kStripTrack, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_strip_kernel<16, false, kStripDirs, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  if (R == 10) launch_strip<10>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes);
  else launch_strip<16>(u8, track, grid, block, ctx->stream, dp, sc, gtab, ref.ncodes);
  HIPCHK(ctx, hipGetLastError());
  std::vector<int32_t> st(n);
  std::vector<int64_t> ci(2 * n);
  HIPCHK(ctx, hipMemcpyAsync(st.data(), ctx->outs_f.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (track) HIPCHK(ctx, hipMemcpyAsync(ci.data(), ctx->outs_i.p, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; ++k) {
    if (st[k] != 0) return fail(ctx, MI355_SW_ENODEV, "internal: strip pipeline wait expired");
    if (track) { jobs[k].ci = ci[2 * k]; jobs[k].cj = ci[2 * k + 1]; jobs[k].best = ci[2 * k] > 0 ? jobs[k].target : -1.0f; }
  }
  return 0;
}

// Traceback of located alignments with the wave kernel: decisions over a window that ends at the argmax along
// the streamed side, grown on demand; then the greedy walk.  orient as WaveJob.
int wave_trace(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
               int orient, const std::vector<int> &qidx, const std::vector<Located> &loc, std::vector<TraceOut> &tout,
               bool strips = false /* long queries: pipelined strip kernel (orient 0 only) */,
               const ScoreTable *table = nullptr /* strips with table scoring: its smax / gap bound the margins */) {
  HostTrace trace_("wave_trace");
  const int64_t nref = rg.hi - rg.lo;
  tout.assign(qidx.size(), TraceOut());
  std::vector<size_t> todo;
  for (size_t k = 0; k < qidx.size(); ++k) if (loc[k].score > 0) todo.push_back(k);
  // exactness margin along the stream: a positive path ending at a stream index spans fewer than
  // na + smax*na/g stream positions (DESIGN.md §3.3 with the roles of the two sequences as given)
  double smax, g;
  if (p.semantics == MI355_SW_U8SAT) { const U8Params u = u8_params(p); smax = u.M; g = u.G; }
  else if (table != nullptr && table->ok) { smax = table->smaxf; g = table->gapf; }
  else { smax = p.match; g = p.gap; }
  // ... and a cell whose lane-side index is a (its path is confined to a rows / columns) is exact a + ceil(a*smax/g)
  // positions into the window: the window needs that margin at the argmax plus room for the walk's excursions
  // along the stream; the walk kernel checks every cell it visits
  const float slope = g > 0 ? (float)(smax / g) : 0.0f;
  auto lane_need = [&](int64_t a) { return a + (int64_t)std::ceil((double)a * (double)slope) + 2; };
  std::vector<int64_t> budget(qidx.size()), warm(qidx.size());
  for (size_t k : todo) {
    const int64_t na = orient == 0 ? q.len[qidx[k]] : nref;
    budget[k] = na / 8 + 64;
    warm[k] = g > 0 ? na + (int64_t)std::ceil(smax * (double)na / g) : (int64_t)1 << 40;
  }
  while (!todo.empty()) {
    std::vector<size_t> next;
    size_t pos = 0;
    while (pos < todo.size()) {
      std::vector<WaveJob> jobs;
      std::vector<size_t> owner;
      size_t bytes = 0;
      while (pos < todo.size() && jobs.size() < 262144) {
        const size_t k = todo[pos];
        const int qi = qidx[k];
        const int64_t na = orient == 0 ? q.len[qi] : nref;
        const int64_t s_end = orient == 0 ? loc[k].iy : loc[k].ix;   // 1-based stream index of the argmax
        const int64_t a_end = orient == 0 ? loc[k].ix : loc[k].iy;   // lane-side index of the argmax
        const int64_t wl = std::max<int64_t>(0, s_end - (budget[k] + std::min(warm[k], lane_need(a_end))));
        const int64_t nb = s_end - wl;
        const size_t need = strips ? strip_dirs_bytes(nb, strip_count((int)na, 10), 10)
                                   : wave_dirs_bytes(nb, 32);      // upper bound whatever instance the group gets
        if (need > kDirsBudget) return fail(ctx, MI355_SW_ENOTSUP, "traceback window exceeds the device scratch budget");
        if (!jobs.empty() && bytes + need > kDirsBudget) break;
        WaveJob j;
        j.q = qi; j.orient = orient; j.s_lo = wl; j.nb = (int32_t)nb; j.track = false; j.dirs = true;
        jobs.push_back(j); owner.push_back(k);
        bytes += need;
        ++pos;
      }
      int gmax = 0;
      for (const WaveJob &j : jobs) gmax = std::max(gmax, orient == 0 ? q.len[j.q] : (int)nref);
      const int groupR = strips ? strip_R(gmax) : wave_R(gmax);     // the instance this group runs on
      int rc = strips ? run_strip(ctx, ref, q, rg, p, jobs, groupR) : run_wave(ctx, ref, q, rg, p, jobs);
      if (rc) return rc;
      // walk: measure, lay out, write (only the bytes that exist are copied back)
      const size_t n = jobs.size();
      std::vector<WaveWalk> wp(n);
      if (ctx->walkp.ensure(n * sizeof(WaveWalk) + n * 24 + n * 8 + 64))
        return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(walk scratch) failed");
      int64_t *wout = reinterpret_cast<int64_t *>(ctx->walkp.as<uint8_t>() + n * sizeof(WaveWalk));
      int64_t *woffs = wout + 3 * n;
      for (size_t t = 0; t < n; ++t) {
        const WaveJob &j = jobs[t];
        const size_t k = owner[t];
        WaveWalk &w = wp[t];
        const int na = orient == 0 ? q.len[j.q] : (int)nref;
        w.x = q.bytes.as<uint8_t>() + q.off[j.q];
        w.y = ref.bytes.as<uint8_t>() + rg.lo;
        w.dirs = reinterpret_cast<const uint32_t *>(ctx->dirs.as<uint8_t>() + j.dirs_off);
        w.na = na;
        w.nb = j.nb;
        w.orient = orient;
        w.R = groupR;
        w.lanes = strips ? 64 * strip_count(na, groupR) : 16;
        w.need_slope = slope;
        w.b_offset = j.s_lo;
        w.start_i = loc[k].ix; w.start_j = loc[k].iy;
        w.exact_from = j.s_lo == 0 ? 0 : j.s_lo + warm[k];
        w.cap = na + j.nb + 2;                                      // a walk inside the window emits <= na + nb pairs
        w.out = wout + 3 * t;
      }
      HIPCHK(ctx, hipMemcpyAsync(ctx->walkp.p, wp.data(), n * sizeof(WaveWalk), hipMemcpyHostToDevice, ctx->stream));
      const unsigned wblocks = (unsigned)((n + 63) / 64);
      std::vector<int64_t> wo(3 * n), offs(n);
      std::vector<char> cons;
      size_t captot = 0;
      for (size_t t = 0; t < n; ++t) captot += 2 * (size_t)wp[t].cap;
      const bool one_pass = n <= 4096 && captot <= ((size_t)32 << 20);
      if (one_pass) {
        // few walks: each writes into a buffer of its own capacity (x at offs, y at offs + cap) in one pass
        size_t at = 0;
        for (size_t t = 0; t < n; ++t) { offs[t] = (int64_t)at; at += 2 * (size_t)wp[t].cap; }
        if (ctx->cons.ensure(captot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(consensus) failed");
        HIPCHK(ctx, hipMemcpyAsync(woffs, offs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkBoth>, dim3(wblocks), dim3(64), 0, ctx->stream, ctx->walkp.as<WaveWalk>(), (int)n,
                           ctx->cons.as<char>(), (const int64_t *)woffs);
        HIPCHK(ctx, hipGetLastError());
        cons.resize(captot + 1);
        HIPCHK(ctx, hipMemcpyAsync(wo.data(), wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(cons.data(), ctx->cons.p, captot, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkMeasure>, dim3(wblocks), dim3(64), 0, ctx->stream, ctx->walkp.as<WaveWalk>(), (int)n,
                           (char *)nullptr, (const int64_t *)nullptr);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(wo.data(), wout, n * 24, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        size_t ctot = 0;
        for (size_t t = 0; t < n; ++t) { offs[t] = (int64_t)ctot; if (wo[3 * t + 2] == 0) ctot += 2 * (size_t)wo[3 * t]; }
        if (ctx->cons.ensure(ctot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(consensus) failed");
        HIPCHK(ctx, hipMemcpyAsync(woffs, offs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(sw_wave_walk_kernel<kWalkWrite>, dim3(wblocks), dim3(64), 0, ctx->stream, ctx->walkp.as<WaveWalk>(), (int)n,
                           ctx->cons.as<char>(), (const int64_t *)woffs);
        HIPCHK(ctx, hipGetLastError());
        cons.resize(ctot + 1);
        if (ctot) HIPCHK(ctx, hipMemcpyAsync(cons.data(), ctx->cons.p, ctot, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      }
      for (size_t t = 0; t < n; ++t) {
        const size_t k = owner[t];
        const int st = (int)wo[3 * t + 2];
        if (st == 0) {
          const size_t len = (size_t)wo[3 * t];
          tout[k].cx.assign(cons.data() + offs[t], len);
          tout[k].cy.assign(cons.data() + offs[t] + (one_pass ? (size_t)wp[t].cap : len), len);
          tout[k].pos = (uint32_t)wo[3 * t + 1];
        } else if (st == 1) { budget[k] *= 4; next.push_back(k); }
        else return fail(ctx, MI355_SW_ENOTSUP, "consensus longer than |x| + |y|");
      }
    }
    todo.swap(next);
  }
  return 0;
}

void set_result(mi355_sw_result &r, float score, int64_t ix, int64_t iy, const TraceOut *t) {
  r.score = score;
  r.end_x = score > 0 ? ix : 0;
  r.end_y = score > 0 ? iy : 0;
  r.pos = t ? t->pos : 0;
  const std::string empty;
  const std::string &cx = t ? t->cx : empty, &cy = t ? t->cy : empty;
  r.cons_len = cx.size();
  // both strings live in ONE allocation owned through cons_x (mi355_sw_free_result frees only that)
  r.cons_x = (char *)malloc(cx.size() + cy.size() + 2);
  r.cons_y = r.cons_x + cx.size() + 1;
  memcpy(r.cons_x, cx.data(), cx.size()); r.cons_x[cx.size()] = 0;
  memcpy(r.cons_y, cy.data(), cy.size()); r.cons_y[cy.size()] = 0;
}

// Traceback for located alignments of one range: windows left of the argmax, grown on demand.
int trace_located(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                  const mi355_sw_params &p, const std::vector<int64_t> &qwarm, const ScoreTable &table,
                  const std::vector<int> &qidx, const std::vector<Located> &loc, std::vector<TraceOut> &tout) {
  tout.assign(qidx.size(), TraceOut());
  // A cell in row i is exact once it lies i + ceil(i * smax / g) columns into a window (the bound of DESIGN.md
  // §3.3 for a path that can only use rows 1..i), so the window needs that margin at the argmax row plus room
  // for the horizontal excursions of the walk; the walk kernel checks every cell it visits against the bound.
  const float slope = table.gapf > 0 ? table.smaxf / table.gapf : 0.0f;
  auto row_need = [&](int64_t i) { return i + (int64_t)std::ceil((double)i * (double)slope) + 2; };
  std::vector<size_t> todo;
  // short reads with identity scoring: decisions by the register-wavefront kernel (lanes = rows of x);
  // long queries (identity scoring, or any table in the float engine): the pipelined strip kernel
  const bool wave_ok = wave_scoring_ok(p) && std::getenv("MI355_SW_NO_WAVE") == nullptr;
  const bool strip_ok = strip_scoring_ok(ref, p);
  for (int pass = 0; pass < 2; ++pass) {
    std::vector<int> sub;
    std::vector<Located> sl;
    std::vector<size_t> owner;
    for (size_t k = 0; k < qidx.size(); ++k) {
      if (!(loc[k].score > 0)) continue;
      const bool is_long = q.len[qidx[k]] > kWaveMaxLanesSide;
      if (is_long != (pass == 1)) continue;
      if (is_long ? strip_ok : wave_ok) { sub.push_back(qidx[k]); sl.push_back(loc[k]); owner.push_back(k); }
      else todo.push_back(k);
    }
    if (sub.empty()) continue;
    std::vector<TraceOut> t2;
    int rc = wave_trace(ctx, ref, q, rg, p, 0, sub, sl, t2, pass == 1, &table);
    if (rc) return rc;
    for (size_t t = 0; t < sub.size(); ++t) tout[owner[t]] = t2[t];
  }
  std::vector<int64_t> budget(qidx.size());
  for (size_t k : todo) budget[k] = (int64_t)q.len[qidx[k]] / 8 + 64;
  while (!todo.empty()) {
    // build jobs in memory-bounded groups
    std::vector<size_t> next;
    size_t pos = 0;
    while (pos < todo.size()) {
      std::vector<ExactJob> jobs;
      std::vector<size_t> owner;
      std::vector<std::pair<int32_t, int32_t>> starts;
      std::vector<int32_t> exlo;
      size_t bytes = 0;
      while (pos < todo.size()) {
        const size_t k = todo[pos];
        const int qi = qidx[k];
        const int64_t iy = loc[k].iy;
        const int64_t warm = qwarm[qi];
        int64_t wl = iy - (budget[k] + row_need(loc[k].ix));   // range-relative 0-based start of window
        if (wl < 0) wl = 0;
        const int64_t nw = iy - wl;
        const size_t need = dirs_bytes(q.len[qi], nw) + 16;
        if (need > kDirsBudget) return fail(ctx, MI355_SW_ENOTSUP, "traceback window exceeds the device scratch budget");
        if (!jobs.empty() && bytes + need > kDirsBudget) break;
        ExactJob j;
        j.q = qi; j.ylo = rg.lo + wl; j.nw = (int32_t)nw; j.col_offset = wl; j.full_n = rg.hi - rg.lo;
        j.own_lo = (int32_t)(nw + 1);                   // nothing competes: decisions only
        j.quirk = 0;                                    // |x| == |y| never reaches the score path (bucket_fast_ok)
        j.target = 1e30f; j.want_dirs = true;
        jobs.push_back(j); owner.push_back(k);
        starts.emplace_back((int32_t)loc[k].ix, (int32_t)nw);
        exlo.push_back(wl == 0 ? 0 : (int32_t)warm);
        bytes += need;
        ++pos;
      }
      int rc = run_exact(ctx, ref, q, p, jobs, 0, jobs.size(), nullptr);
      if (rc) return rc;
      std::vector<TraceOut> outs;
      std::vector<int> st;
      rc = run_walk(ctx, ref, q, jobs, 0, jobs.size(), starts, exlo, outs, st, slope);
      if (rc) return rc;
      for (size_t t = 0; t < jobs.size(); ++t) {
        const size_t k = owner[t];
        if (st[t] == 0) tout[k] = outs[t];
        else if (st[t] == 1) { budget[k] *= 4; next.push_back(k); }
        else return fail(ctx, MI355_SW_ENOTSUP, "consensus longer than |x| + window");
      }
    }
    todo.swap(next);
  }
  return 0;
}

// Whole-matrix path on the LDS anti-diagonal kernel (sw_exact_kernel.h) for the listed queries over one range.
int exact_full_lds(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                   const mi355_sw_params &p, const std::vector<int> &qidx, bool want_trace,
                   std::vector<Located> &loc, std::vector<TraceOut> &tout) {
  const int64_t n = rg.hi - rg.lo;
  loc.assign(qidx.size(), Located());
  tout.assign(qidx.size(), TraceOut());
  size_t pos = 0;
  while (pos < qidx.size()) {
    std::vector<ExactJob> jobs;
    std::vector<size_t> owner;
    size_t bytes = 0;
    while (pos < qidx.size()) {
      const int qi = qidx[pos];
      const size_t need = dirs_bytes(q.len[qi], n) + 16;
      if (want_trace && need > kDirsBudget) return fail(ctx, MI355_SW_ENOTSUP, "problem needs the score kernel but is outside its coverage");
      if (!jobs.empty() && want_trace && bytes + need > kDirsBudget) break;
      if (jobs.size() >= 65536) break;
      ExactJob j;
      j.q = qi; j.ylo = rg.lo; j.nw = (int32_t)n; j.col_offset = 0; j.full_n = n; j.own_lo = 1;
      j.quirk = (p.semantics == MI355_SW_U8SAT && q.len[qi] == n) ? 1 : 0;
      j.target = -1.0f; j.want_dirs = want_trace;
      jobs.push_back(j); owner.push_back(pos);
      bytes += need;
      ++pos;
    }
    int rc = run_exact(ctx, ref, q, p, jobs, 0, jobs.size(), nullptr);
    if (rc) return rc;
    std::vector<std::pair<int32_t, int32_t>> starts;
    std::vector<int32_t> exlo(jobs.size(), 0);
    for (size_t t = 0; t < jobs.size(); ++t) {
      Located &L = loc[owner[t]];
      L.score = jobs[t].best > 0 ? jobs[t].best : 0;
      L.ix = jobs[t].ci; L.iy = jobs[t].cj;
      starts.emplace_back(L.score > 0 ? (int32_t)L.ix : 0, L.score > 0 ? (int32_t)L.iy : 0);
    }
    if (want_trace) {
      std::vector<TraceOut> outs;
      std::vector<int> st;
      rc = run_walk(ctx, ref, q, jobs, 0, jobs.size(), starts, exlo, outs, st);
      if (rc) return rc;
      for (size_t t = 0; t < jobs.size(); ++t) {
        if (st[t] != 0) return fail(ctx, MI355_SW_ENOTSUP, "traceback walk failed on a whole-matrix window");
        tout[owner[t]] = outs[t];
      }
    }
  }
  return 0;
}

// Whole-matrix path for the listed queries over one range (problems the score kernel does not take).
// Small problems with identity scoring run on the register-wavefront kernel (sw_wave_kernel.h): argmax tracking
// for the float engine, traceback decisions for both engines; everything else on the LDS anti-diagonal kernel.
int exact_full(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
               const mi355_sw_params &p, const std::vector<int> &qidx, bool want_trace,
               std::vector<Located> &loc, std::vector<TraceOut> &tout) {
  HostTrace trace_("exact_full");
  const int64_t n = rg.hi - rg.lo;
  loc.assign(qidx.size(), Located());
  tout.assign(qidx.size(), TraceOut());
  const bool wave_ok = wave_scoring_ok(p) && std::getenv("MI355_SW_NO_WAVE") == nullptr;
  const bool u8 = p.semantics == MI355_SW_U8SAT;
  // orientation per query: -1 = LDS kernel for everything
  std::vector<int> orient(qidx.size(), -1);
  for (size_t k = 0; k < qidx.size(); ++k) {
    const int m = q.len[qidx[k]];
    if (!wave_ok || m < 1 || n < 1) continue;
    if (u8 && m == n) continue;                                   // |x| == |y| quirk lives in the LDS kernel
    if (m <= kWaveMaxLanesSide && m <= n) orient[k] = 0;
    else if (n <= kWaveMaxLanesSide) orient[k] = 1;
  }
  // 1. score + argmax
  std::vector<int> lds_all, lds_score_only;                       // positions k
  for (size_t k = 0; k < qidx.size(); ++k) {
    if (orient[k] < 0) lds_all.push_back((int)k);
    else if (u8) lds_score_only.push_back((int)k);                // uint8 storage order: LDS kernel's order_key
  }
  auto run_lds = [&](const std::vector<int> &ks, bool trace) -> int {
    if (ks.empty()) return 0;
    std::vector<int> sub(ks.size());
    for (size_t t = 0; t < ks.size(); ++t) sub[t] = qidx[ks[t]];
    std::vector<Located> l2;
    std::vector<TraceOut> t2;
    int rc = exact_full_lds(ctx, ref, q, rg, p, sub, trace, l2, t2);
    if (rc) return rc;
    for (size_t t = 0; t < ks.size(); ++t) { loc[ks[t]] = l2[t]; tout[ks[t]] = t2[t]; }
    return 0;
  };
  int rc = run_lds(lds_all, want_trace);
  if (rc) return rc;
  rc = run_lds(lds_score_only, false);
  if (rc) return rc;
  if (!u8) {
    for (int o = 0; o < 2; ++o) {
      std::vector<WaveJob> jobs;
      std::vector<size_t> owner;
      for (size_t k = 0; k < qidx.size(); ++k) {
        if (orient[k] != o) continue;
        WaveJob j;
        j.q = qidx[k]; j.orient = o; j.s_lo = 0; j.nb = o == 0 ? (int32_t)n : q.len[qidx[k]]; j.track = true; j.dirs = false;
        jobs.push_back(j); owner.push_back(k);
      }
      rc = run_wave(ctx, ref, q, rg, p, jobs);
      if (rc) return rc;
      for (size_t t = 0; t < jobs.size(); ++t) {
        Located &L = loc[owner[t]];
        L.score = jobs[t].best > 0 ? jobs[t].best : 0;
        L.ix = jobs[t].ci; L.iy = jobs[t].cj;
      }
    }
  }
  // 2. traceback of the wave-eligible ones
  if (want_trace) {
    for (int o = 0; o < 2; ++o) {
      std::vector<int> sub;
      std::vector<Located> sl;
      std::vector<size_t> owner;
      for (size_t k = 0; k < qidx.size(); ++k)
        if (orient[k] == o) { sub.push_back(qidx[k]); sl.push_back(loc[k]); owner.push_back(k); }
      if (sub.empty()) continue;
      std::vector<TraceOut> t2;
      rc = wave_trace(ctx, ref, q, rg, p, o, sub, sl, t2);
      if (rc) return rc;
      for (size_t t = 0; t < sub.size(); ++t) tout[owner[t]] = t2[t];
    }
  }
  return 0;
}

// Argmax cells for the score-kernel queries (qfast[q] != 0) over one range, from the score pass' keys.
// qchunk / qwarm: tile geometry of each query's bucket.
int locate_fast(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                const mi355_sw_params &p, const std::vector<char> &qfast, const std::vector<int64_t> &qchunk,
                const std::vector<int64_t> &qwarm, const std::vector<char> &qfloat, const unsigned long long *keys,
                const ScoreTable &table, std::vector<Located> &loc) {
  HostTrace trace_("locate_fast");
  const size_t nq = q.nq;
  const int64_t n = rg.hi - rg.lo;
  std::vector<ExactJob> jobs;
  std::vector<WaveJob> sjobs;                  // long queries with identity scoring: pipelined strip kernel
  std::vector<WaveJob> wjobs;                  // short queries, float engine, identity scoring: register wavefront
  const bool strip_ok = strip_scoring_ok(ref, p);
  // float order = (column, row): no cell left of the sub-chunk can equal the maximum (it would have been reported
  // by an earlier sub-chunk), so the wave kernel's plain first-maximum tracking over the whole window is the answer
  // the uint8 order needs the storage-order key of every cell that equals the maximum: keyed tracking
  const bool wave_locate = wave_scoring_ok(p) && std::getenv("MI355_SW_NO_WAVE") == nullptr;
  const bool wave_keyed = p.semantics == MI355_SW_U8SAT;
  auto key_score = [&](size_t k) {
    float score;
    if (qfloat[k]) { const uint32_t bits = (uint32_t)(keys[k] >> 32); memcpy(&score, &bits, 4); }
    else score = (float)(int)(keys[k] >> 32);
    return score;
  };
  // long queries are few and each re-run occupies one workgroup: cut their sub-chunk into pieces (each with its
  // own margin) so that the idle CUs share the work
  size_t nlong = 0;                       // workgroups the long queries' sub-chunks need before cutting
  for (size_t k = 0; k < nq; ++k)
    if (qfast[k] && q.len[k] > 512 && key_score(k) > 0) nlong += p.semantics == MI355_SW_U8SAT ? 5 : 1;
  for (size_t k = 0; k < nq; ++k) {
    if (!qfast[k]) continue;
    const unsigned long long key = keys[k];
    const float score = key_score(k);
    if (!(score > 0)) continue;
    const int64_t chunk_len = qchunk[k];           // sub-chunk granularity of this query's bucket
    const int64_t nchunks = (n + chunk_len - 1) / chunk_len;
    // only cells equal to the known maximum compete: a path that reaches `score` within |x| diagonal steps can
    // afford fewer gap columns than the general margin allows (DESIGN.md §3.3 with the score subtracted)
    int64_t warm = qwarm[k];
    if (table.gapf > 0) {
      const double spare = std::max(0.0, (double)table.smaxf * (double)q.len[k] - (double)score);
      warm = std::min<int64_t>(warm, (int64_t)q.len[k] + (int64_t)std::ceil(spare / (double)table.gapf) + 2);
    }
    const int64_t first = (int64_t)(0xFFFFFFFFull - (key & 0xFFFFFFFFull));
    loc[k].score = score;
    int64_t cand[5];
    int nc = 0;
    cand[nc++] = first;
    if (p.semantics == MI355_SW_U8SAT) {
      // storage order = anti-diagonal (mod ncols): the first maximum lies in the first tile that
      // reached the maximum or the next one, or in the corner triangles (first / last two tiles)
      const int64_t extra[4] = {first + 1, 0, nchunks - 2, nchunks - 1};
      for (int64_t c : extra) {
        if (c < 0 || c >= nchunks) continue;
        bool dup = false;
        for (int t = 0; t < nc; ++t) dup |= cand[t] == c;
        if (!dup) cand[nc++] = c;
      }
    }
    for (int t = 0; t < nc; ++t) {
      // lanes lag by up to 63 columns (whole-wavefront tiles): the end of the previous sub-chunk is reported with this one
      const int64_t sub_lo = std::max<int64_t>(0, cand[t] * chunk_len - 63);   // range-relative, 0-based
      const int64_t sub_hi = std::min((cand[t] + 1) * chunk_len, n);
      int64_t pieces = 1;
      if (q.len[k] > 512) pieces = std::max<int64_t>(1, std::min<int64_t>((sub_hi - sub_lo) / 256, 224 / (int64_t)nlong));   // one 1024-thread workgroup per CU
      const int64_t plen = (sub_hi - sub_lo + pieces - 1) / pieces;
      for (int64_t own_lo = sub_lo; own_lo < sub_hi; own_lo += plen) {
        const int64_t own_hi = std::min(own_lo + plen, sub_hi);
        const int64_t wl = std::max<int64_t>(0, own_lo - warm);
        if (wave_locate && q.len[k] <= kWaveMaxLanesSide) {
          WaveJob wj;
          wj.q = (int)k; wj.orient = 0; wj.s_lo = wl; wj.nb = (int32_t)(own_hi - wl); wj.track = true; wj.dirs = false;
          wj.target = score; wj.keyed = wave_keyed; wj.own_lo = (int32_t)(own_lo - wl);
          wjobs.push_back(wj);
          continue;
        }
        if (strip_ok && q.len[k] > kWaveMaxLanesSide) {
          WaveJob sj;
          sj.q = (int)k; sj.orient = 0; sj.s_lo = wl; sj.nb = (int32_t)(own_hi - wl); sj.track = true; sj.dirs = false;
          sj.target = score; sj.own_lo = (int32_t)(own_lo - wl);
          sjobs.push_back(sj);
          continue;
        }
        ExactJob j;
        j.q = (int)k; j.ylo = rg.lo + wl; j.nw = (int32_t)(own_hi - wl); j.col_offset = wl; j.full_n = n;
        j.own_lo = (int32_t)(own_lo - wl + 1); j.quirk = 0; j.target = score; j.want_dirs = false;
        jobs.push_back(j);
      }
    }
  }
  for (size_t lo = 0; lo < jobs.size(); lo += 65536) {
    int rc = run_exact(ctx, ref, q, p, jobs, lo, std::min(jobs.size(), lo + 65536), nullptr);
    if (rc) return rc;
  }
  std::vector<unsigned long long> bestkey(nq, ~0ull);
  for (const ExactJob &j : jobs) {
    if (j.best != j.target) continue;
    const unsigned long long kk = host_order_key(p.semantics, j.ci, j.cj, q.len[j.q], n);
    if (kk < bestkey[j.q]) { bestkey[j.q] = kk; loc[j.q].ix = j.ci; loc[j.q].iy = j.cj; }
  }
  for (size_t lo = 0; lo < wjobs.size(); lo += 262144) {
    std::vector<WaveJob> part(wjobs.begin() + lo, wjobs.begin() + std::min(wjobs.size(), lo + 262144));
    int rc = run_wave(ctx, ref, q, rg, p, part);
    if (rc) return rc;
    for (const WaveJob &j : part) {
      if (j.best != j.target) continue;
      const unsigned long long kk = host_order_key(p.semantics, j.ci, j.cj, q.len[j.q], n);
      if (kk < bestkey[j.q]) { bestkey[j.q] = kk; loc[j.q].ix = j.ci; loc[j.q].iy = j.cj; }
    }
  }
  if (!sjobs.empty()) {
    // one launch per kernel instance (rows per lane)
    for (int R : {10, 16}) {
      std::vector<WaveJob> group;
      for (const WaveJob &j : sjobs) if (strip_R(q.len[j.q]) == R) group.push_back(j);
      for (size_t lo = 0; lo < group.size(); lo += 4096) {
        std::vector<WaveJob> part(group.begin() + lo, group.begin() + std::min(group.size(), lo + 4096));
        int rc = run_strip(ctx, ref, q, rg, p, part, R);
        if (rc) return rc;
        for (const WaveJob &j : part) {
          if (j.ci <= 0) continue;
          const unsigned long long kk = host_order_key(p.semantics, j.ci, j.cj, q.len[j.q], n);
          if (kk < bestkey[j.q]) { bestkey[j.q] = kk; loc[j.q].ix = j.ci; loc[j.q].iy = j.cj; }
        }
      }
    }
  }
  for (size_t k = 0; k < nq; ++k)
    if (qfast[k] && loc[k].score > 0 && bestkey[k] == ~0ull)
      return fail(ctx, MI355_SW_ENODEV, "internal: maximum of the score pass not found again by the exact kernel");
  return 0;
}

float elapsed_us(mi355_sw_ctx *ctx, hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0;
  (void)ctx;
  return ms * 1000.0f;
}

// All queries of `q` against one range of the reference.
int align_range(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                const mi355_sw_params &p, int flags, mi355_sw_result *outs) {
  HostTrace trace_("align_range");
  const bool want_trace = !(flags & MI355_SW_SCORE_ONLY);
  const size_t nq = q.nq;
  const int64_t n = rg.hi - rg.lo;
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  std::vector<Located> loc(nq);
  std::vector<TraceOut> tout(nq);
  if (n >= 1 && nq > 0) {
    const ScoreTable table = plan_table(ref, p);
    std::vector<Bucket> buckets = make_buckets(ref, q, table, p, n);
    std::vector<char> qfast(nq, 0), qfloat(nq, 0);
    std::vector<int64_t> qchunk(nq, 0), qwarm(nq, 0);
    bool any_fast = false;
    for (Bucket &b : buckets) { b.fast = bucket_fast_ok(ref, table, b, n, p); any_fast |= b.fast; }
    if (any_fast) {
      const std::vector<Range> ranges{rg};
      int rc = score_begin(ctx, q, ranges, table);
      if (rc) return rc;
      for (Bucket &b : buckets) {
        if (!b.fast) continue;
        rc = score_launch(ctx, ref, q, ranges, p, table, b);
        if (rc) return rc;
        for (int k = 0; k < b.count; ++k) {
          const int id = q.order[b.first + k];
          qfast[id] = 1; qchunk[id] = b.sub_len; qwarm[id] = b.warm; qfloat[id] = sem_is_float(b.sem);
        }
      }
      std::vector<unsigned long long> keys;
      rc = score_fetch(ctx, nq, keys);
      if (rc) return rc;
      HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      rc = locate_fast(ctx, ref, q, rg, p, qfast, qchunk, qwarm, qfloat, keys.data(), table, loc);
      if (rc) return rc;
      HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      ctx->timings[1] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
      if (want_trace) {
        std::vector<int> fq;
        std::vector<Located> floc;
        for (size_t k = 0; k < nq; ++k) if (qfast[k]) { fq.push_back((int)k); floc.push_back(loc[k]); }
        std::vector<TraceOut> ft;
        HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
        rc = trace_located(ctx, ref, q, rg, p, qwarm, table, fq, floc, ft);
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->timings[2] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
        for (size_t t = 0; t < fq.size(); ++t) tout[fq[t]] = ft[t];
      }
    }
    std::vector<int> slow;
    for (size_t k = 0; k < nq; ++k) if (!qfast[k]) slow.push_back((int)k);
    if (!slow.empty()) {
      std::vector<Located> sl;
      std::vector<TraceOut> st;
      HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      int rc = exact_full(ctx, ref, q, rg, p, slow, want_trace, sl, st);
      if (rc) {
        if (!table.ok && ctx->err.find("outside its coverage") != std::string::npos) ctx->err += " (" + table.why + ")";
        return rc;
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      ctx->timings[2] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
      for (size_t t = 0; t < slow.size(); ++t) { loc[slow[t]] = sl[t]; tout[slow[t]] = st[t]; }
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  ctx->timings[3] += elapsed_us(ctx, ctx->ev[4], ctx->ev[5]);
  HostTrace trace_results("set_results");
  for (size_t k = 0; k < nq; ++k) {
    set_result(outs[k], loc[k].score, loc[k].ix, loc[k].iy, (want_trace && loc[k].score > 0) ? &tout[k] : nullptr);
    outs[k].timings_us[0] = (float)(ctx->timings[0] > 0 ? ctx->timings[0] : ctx->timings[3]);
    outs[k].timings_us[1] = 0;
  }
  return 0;
}

// Per-range maxima of every query (value half of find_index_of_maximum per piece).
int range_maxima(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<Range> &ranges,
                 const mi355_sw_params &p, float *maxima /* [nranges][nq] */) {
  const size_t nq = q.nq, nr = ranges.size();
  if (nq == 0 || nr == 0) return 0;
  const ScoreTable table = plan_table(ref, p);
  int64_t maxn = 0;
  for (auto &r : ranges) maxn = std::max(maxn, r.hi - r.lo);
  std::vector<Bucket> buckets = make_buckets(ref, q, table, p, maxn);
  std::vector<char> qfast(nq, 0), qfloat(nq, 0);
  for (Bucket &b : buckets) {
    b.fast = true;
    for (auto &r : ranges) b.fast = b.fast && bucket_fast_ok(ref, table, b, r.hi - r.lo, p);
  }
  for (size_t lo = 0; lo < nr; lo += 32768) {
    const size_t hi = std::min(nr, lo + 32768);
    const std::vector<Range> sub(ranges.begin() + lo, ranges.begin() + hi);
    bool any = false;
    for (Bucket &b : buckets) any |= b.fast;
    if (!any) break;
    int rc = score_begin(ctx, q, sub, table);
    if (rc) return rc;
    for (Bucket &b : buckets) {
      if (!b.fast) continue;
      rc = score_launch(ctx, ref, q, sub, p, table, b);
      if (rc) return rc;
      for (int k = 0; k < b.count; ++k) { qfast[q.order[b.first + k]] = 1; qfloat[q.order[b.first + k]] = sem_is_float(b.sem); }
    }
    std::vector<unsigned long long> keys;
    rc = score_fetch(ctx, nq * sub.size(), keys);
    if (rc) return rc;
    for (size_t r = 0; r < sub.size(); ++r)
      for (size_t k = 0; k < nq; ++k)
        if (qfast[k]) {
          const uint32_t hi32 = (uint32_t)(keys[r * nq + k] >> 32);
          float v;
          if (qfloat[k]) memcpy(&v, &hi32, 4); else v = (float)hi32;
          maxima[(lo + r) * nq + k] = v;
        }
  }
  std::vector<int> slow;
  for (size_t k = 0; k < nq; ++k) if (!qfast[k]) slow.push_back((int)k);
  for (size_t r = 0; r < nr && !slow.empty(); ++r) {
    std::vector<Located> loc;
    std::vector<TraceOut> t;
    int rc = exact_full(ctx, ref, q, ranges[r], p, slow, false, loc, t);
    if (rc) return rc;
    for (size_t i = 0; i < slow.size(); ++i) maxima[r * nq + slow[i]] = loc[i].score;
  }
  return 0;
}

int check_params(mi355_sw_ctx *ctx, const mi355_sw_params *p) {
  if (!ctx) return MI355_SW_EINVAL;
  if (!p) return fail(ctx, MI355_SW_EINVAL, "params is NULL");
  if (p->semantics != MI355_SW_F32 && p->semantics != MI355_SW_U8SAT) return fail(ctx, MI355_SW_EINVAL, "unknown semantics");
  return 0;
}

void reset_timings(mi355_sw_ctx *ctx) { for (double &t : ctx->timings) t = 0; ctx->score_ev_used = 0; }

}  // namespace

// ================================= C-ABI ======================================================
extern "C" {

const char *mi355_sw_build_info(void) { return "mi355_sw gfx950 hip; score kernel R={2,4,6,8,10,12,16,20,24,32} x {i16 pairs,u8sat pairs,f32} + strip-mined R=32"; }

void mi355_sw_default_params(mi355_sw_params *p) {
  if (!p) return;
  p->lut = nullptr; p->match = 3.0f; p->mismatch = -3.0f; p->gap = 2.0f; p->semantics = MI355_SW_F32;
}

int mi355_sw_create(mi355_sw_ctx **out, int device) {
  if (!out) return MI355_SW_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MI355_SW_ENODEV;
  if (hipSetDevice(device) != hipSuccess) return MI355_SW_ENODEV;
  mi355_sw_ctx *c = new (std::nothrow) mi355_sw_ctx();
  if (!c) return MI355_SW_ENOMEM;
  c->device = device;
  if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return MI355_SW_ENODEV; }
  for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return MI355_SW_ENODEV; }
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<0, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_exact_kernel<1, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kExactLdsMax);
  *out = c;
  return 0;
}

void mi355_sw_destroy(mi355_sw_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  DevBuf *bufs[] = {&c->ref.bytes, &c->ref.codes, &c->batch.bytes, &c->batch.lens, &c->keys, &c->ranges, &c->stab,
                    &c->batch.offs, &c->batch.sel, &c->ftab, &c->lut, &c->probs, &c->dirs, &c->outs_f, &c->outs_i, &c->cons, &c->walkp, &c->hmat, &c->brow, &c->wprobs};
  for (DevBuf *b : bufs) b->release();
  c->adhoc.release(); c->one.release();
  for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto &e : c->score_ev) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *mi355_sw_last_error(const mi355_sw_ctx *c) { return c ? c->err.c_str() : "null context"; }

int mi355_sw_set_reference(mi355_sw_ctx *ctx, const char *y, size_t ny) {
  if (!ctx || (!y && ny)) return MI355_SW_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return upload_reference(ctx, ctx->ref, y, ny);
}

int mi355_sw_batch_upload(mi355_sw_ctx *ctx, size_t n, const char *const *xs, const size_t *nxs) {
  if (!ctx || !xs || !nxs) return MI355_SW_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return upload_queries(ctx, ctx->batch, n, xs, nxs);
}

int mi355_sw_batch_run(mi355_sw_ctx *ctx, const mi355_sw_params *params, int flags, mi355_sw_result *outs) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!outs) return fail(ctx, MI355_SW_EINVAL, "outs is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  if (ctx->batch.nq == 0) return 0;
  return align_range(ctx, ctx->ref, ctx->batch, Range{0, (int64_t)ctx->ref.n}, *params, flags, outs);
}

int mi355_sw_align_batch(mi355_sw_ctx *ctx, size_t n, const char *const *xs, const size_t *nxs,
                         const mi355_sw_params *params, int flags, mi355_sw_result *outs) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  rc = mi355_sw_batch_upload(ctx, n, xs, nxs);
  if (rc) return rc;
  return mi355_sw_batch_run(ctx, params, flags, outs);
}

int mi355_sw_align(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                   const mi355_sw_params *params, mi355_sw_result *out) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!out || (!x && nx) || (!y && ny)) return fail(ctx, MI355_SW_EINVAL, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  const RefData *ref = nullptr;
  AdhocSpeculation sp;
  memset(out, 0, sizeof *out);
  rc = adhoc_begin(ctx, y, ny, &ref, sp);
  if (!rc) rc = upload_queries(ctx, ctx->one, 1, &x, &nx);
  if (!rc) rc = align_range(ctx, *ref, ctx->one, Range{0, (int64_t)ny}, *params, 0, out);
  int rc2 = 0;
  if (!adhoc_confirm(ctx, y, ny, &ref, sp, rc2)) {             // the caller's buffer changed since the last call
    mi355_sw_free_result(out);
    rc = rc2;
    if (!rc) rc = align_range(ctx, *ref, ctx->one, Range{0, (int64_t)ny}, *params, 0, out);
  }
  return rc;
}

int mi355_sw_argmax(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                    const mi355_sw_params *params, int64_t *index_x, int64_t *index_y, float *mx) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  mi355_sw_result r;
  memset(&r, 0, sizeof r);
  const RefData *ref = nullptr;
  AdhocSpeculation sp;
  rc = adhoc_begin(ctx, y, ny, &ref, sp);
  if (!rc) rc = upload_queries(ctx, ctx->one, 1, &x, &nx);
  if (!rc) rc = align_range(ctx, *ref, ctx->one, Range{0, (int64_t)ny}, *params, MI355_SW_SCORE_ONLY, &r);
  int rc2 = 0;
  if (!adhoc_confirm(ctx, y, ny, &ref, sp, rc2)) {
    mi355_sw_free_result(&r);
    rc = rc2;
    if (!rc) rc = align_range(ctx, *ref, ctx->one, Range{0, (int64_t)ny}, *params, MI355_SW_SCORE_ONLY, &r);
  }
  if (rc) return rc;
  if (index_x) *index_x = r.end_x;
  if (index_y) *index_y = r.end_y;
  if (mx) *mx = r.score;
  mi355_sw_free_result(&r);
  return 0;
}

int mi355_sw_make_string_range(int npiece, int64_t shortlen, int64_t longlen, float overlap_ratio,
                               int64_t *lefts, int64_t *rights) {
  // plocalaligner.cpp:44-67
  if (npiece < 1 || !lefts || !rights) return MI355_SW_EINVAL;
  const int64_t overlap = (int64_t)((float)shortlen * overlap_ratio);
  if (npiece == 1) { lefts[0] = 0; rights[0] = longlen; return 0; }
  const int64_t piecelen = (longlen + (int64_t)(npiece - 1) * overlap) / npiece;
  if (!(overlap <= piecelen)) return MI355_SW_ERANGE;
  int64_t left = 0, right = piecelen;
  int k = 0;
  lefts[k] = left; rights[k] = right; ++k;
  while (k < npiece - 1) {
    left = std::max<int64_t>(0, right - overlap);
    right = std::min(left + piecelen, longlen);
    lefts[k] = left; rights[k] = right; ++k;
  }
  if (!(right < longlen)) return MI355_SW_ERANGE;
  lefts[k] = std::max<int64_t>(0, right - overlap); rights[k] = longlen;
  return 0;
}

int mi355_sw_align_split(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                         const mi355_sw_params *params, int sm_semantics, int la_semantics,
                         int npiece, float overlap_ratio, mi355_sw_result *out, int *winning_piece) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!out || npiece < 1) return fail(ctx, MI355_SW_EINVAL, "bad argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  std::vector<int64_t> lefts(npiece), rights(npiece);
  rc = mi355_sw_make_string_range(npiece, (int64_t)nx, (int64_t)ny, overlap_ratio, lefts.data(), rights.data());
  if (rc) return fail(ctx, rc, "_make_string_range: the reference's asserts would fire for these arguments");
  const RefData *refp = nullptr;
  AdhocSpeculation sp;
  memset(out, 0, sizeof *out);
  rc = adhoc_begin(ctx, y, ny, &refp, sp);
  QueryBatch &q = ctx->one;
  if (!rc) rc = upload_queries(ctx, q, 1, &x, &nx);
  int bp = 0;
  auto work = [&]() -> int {
    mi355_sw_params ps = *params;
    ps.semantics = sm_semantics;
    std::vector<Range> ranges(npiece);
    for (int k = 0; k < npiece; ++k) ranges[k] = Range{lefts[k], rights[k]};
    std::vector<float> pmax(npiece, 0.0f);
    int r = range_maxima(ctx, *refp, q, ranges, ps, pmax.data());
    if (r) return r;
    float best = -1.0f;                                  // plocalaligner.cpp:106,122-129
    bp = 0;
    for (int k = 0; k < npiece; ++k) if (pmax[k] > best) { best = pmax[k]; bp = k; }
    mi355_sw_params pd;
    mi355_sw_default_params(&pd);                        // LAT(x, piece): default scoring (:135)
    pd.semantics = la_semantics;
    const double t_score = ctx->timings[0];
    r = align_range(ctx, *refp, q, ranges[bp], pd, 0, out);
    if (r) return r;
    if (out->score > 0) { out->pos += (uint32_t)lefts[bp]; out->end_y += lefts[bp]; }
    else out->pos = (uint32_t)lefts[bp];
    out->timings_us[0] = (float)t_score;
    out->timings_us[1] = (float)t_score;
    return 0;
  };
  if (!rc) rc = work();
  int rc2 = 0;
  if (!adhoc_confirm(ctx, y, ny, &refp, sp, rc2)) {             // the caller's buffer changed since the last call
    mi355_sw_free_result(out);
    reset_timings(ctx);
    rc = rc2;
    if (!rc) rc = work();
  }
  if (winning_piece) *winning_piece = bp;
  return rc;
}

int mi355_sw_score_ranges(mi355_sw_ctx *ctx, size_t nranges, const int64_t *lefts, const int64_t *rights,
                          const mi355_sw_params *params, float *maxima) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!lefts || !rights || !maxima) return fail(ctx, MI355_SW_EINVAL, "null argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  reset_timings(ctx);
  const QueryBatch &q = ctx->batch;
  if (q.nq == 0 || nranges == 0) return 0;
  std::vector<Range> ranges(nranges);
  int64_t maxlen = 0;
  for (size_t k = 0; k < nranges; ++k) {
    if (lefts[k] < 0 || rights[k] < lefts[k] || rights[k] > (int64_t)ctx->ref.n) return fail(ctx, MI355_SW_EINVAL, "range outside the resident reference");
    ranges[k] = Range{lefts[k], rights[k]};
    maxlen = std::max(maxlen, rights[k] - lefts[k]);
  }
  (void)maxlen;
  return range_maxima(ctx, ctx->ref, q, ranges, *params, maxima);
}

int mi355_sw_fill_matrix(mi355_sw_ctx *ctx, const char *x, size_t nx, const char *y, size_t ny,
                         const mi355_sw_params *params, float *H) {
  int rc = check_params(ctx, params);
  if (rc) return rc;
  if (!H) return fail(ctx, MI355_SW_EINVAL, "H is NULL");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t cells = (nx + 1) * (ny + 1);
  if (cells * 4 > (size_t)8 << 30) return fail(ctx, MI355_SW_ENOTSUP, "matrix larger than 8 GiB");
  if (nx == 0 || ny == 0) { memset(H, 0, cells * 4); return 0; }
  RefData ref;
  QueryBatch q;
  rc = upload_reference(ctx, ref, y, ny);
  if (!rc) rc = upload_queries(ctx, q, 1, &x, &nx);
  if (!rc && ctx->hmat.ensure(cells * 4)) rc = fail(ctx, MI355_SW_ENOMEM, "hipMalloc(matrix) failed");
  if (!rc) {
    if (hipMemsetAsync(ctx->hmat.p, 0, cells * 4, ctx->stream) != hipSuccess) rc = fail(ctx, MI355_SW_ENODEV, "hipMemsetAsync failed");
  }
  if (!rc) {
    std::vector<ExactJob> jobs(1);
    ExactJob &j = jobs[0];
    j.q = 0; j.ylo = 0; j.nw = (int32_t)ny; j.col_offset = 0; j.full_n = (int64_t)ny; j.own_lo = 1;
    j.quirk = (params->semantics == MI355_SW_U8SAT && nx == ny) ? 1 : 0;
    j.target = -1.0f; j.want_dirs = false;
    rc = run_exact(ctx, ref, q, *params, jobs, 0, 1, ctx->hmat.as<float>());
  }
  if (!rc && hipMemcpy(H, ctx->hmat.p, cells * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, MI355_SW_ENODEV, "hipMemcpy(matrix) failed");
  ref.release(); q.release();
  return rc;
}

void mi355_sw_true2raw(size_t nx, size_t ny, size_t ti, size_t tj, size_t *ri, size_t *rj) {
  // similaritymatrix.cpp:353-364 with len_x = |y|+1, len_y = |x|+1 (constructor swap, :274-285)
  const size_t len_x = ny + 1, len_y = nx + 1;
  const size_t nrows = std::min(len_x, len_y), ncols = std::max(len_x, len_y);
  if (ti + tj < nrows - 1) { *ri = ti; *rj = ti + tj; }
  else if (ti + tj > ncols - 1) { *ri = ti - ncols + len_y; *rj = ti + tj - (ncols - 1) - 1; }
  else { *ri = (len_x <= len_y) ? ti : len_y - 1 - tj; *rj = ti + tj; }
}

void mi355_sw_raw2true(size_t nx, size_t ny, size_t ri, size_t rj, size_t *ti, size_t *tj) {
  // similaritymatrix.cpp:330-346
  const size_t len_x = ny + 1, len_y = nx + 1;
  const size_t nrows = std::min(len_x, len_y);
  if (rj < nrows - 1) {
    if (ri <= rj) { *ti = ri; *tj = rj - ri; }
    else { *ti = len_x - nrows + ri; *tj = len_y - ri + rj; }
  } else if (len_x <= len_y) { *ti = ri; *tj = rj - ri; }
  else { *ti = rj - (nrows - 1) + ri; *tj = nrows - 1 - ri; }
}

int mi355_sw_last_timings(const mi355_sw_ctx *ctx, double out[6]) {
  if (!ctx || !out) return MI355_SW_EINVAL;
  for (int k = 0; k < 6; ++k) out[k] = ctx->timings[k];
  return 0;
}

void mi355_sw_free_result(mi355_sw_result *r) {
  if (!r) return;
  free(r->cons_x);                    // cons_y points into the same allocation
  r->cons_x = r->cons_y = nullptr; r->cons_len = 0;
}

void mi355_sw_free_results(mi355_sw_result *r, size_t n) {
  if (!r) return;
  for (size_t k = 0; k < n; ++k) mi355_sw_free_result(r + k);
}

}  // extern "C"
