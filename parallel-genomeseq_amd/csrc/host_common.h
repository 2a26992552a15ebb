// host_common.h — engine context, device buffers, reference / query upload, the resident-reference cache
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
namespace {

constexpr size_t kProfileLdsMax = 120 * 1024;  // LDS budget of the query profile (ncodes x 16 lanes x stride x 4 B)
constexpr int kMaxRowsFast = 512;             // 16 lanes x R <= 32 rows in one strip; longer queries are strip-mined
constexpr size_t kDirsBudget = 16ull << 30;    // bytes of traceback decisions per exact launch
constexpr size_t kExactLdsMax = 159 * 1024;    // dynamic part; the wide instance adds < 1 KiB of static LDS

// ---- options: A/B and diagnostic switches, none of which changes results -------------------------------
// Every context carries its own set: defaults come from the environment (MI355_SW_<NAME>, read ONCE when the context is
// created), mi355_sw_set_option overrides them.  The host code reads them through opt(): each C-ABI entry binds the
// calling thread to its context's options for the duration of the call (a context serves one host thread at a time).
#define MI355_SW_BOOL_OPTIONS(X) \
  X(no_f16) X(no_unsat) X(no_sample) X(no_satflag) X(no_solo) X(no_wave) X(no_comb) X(no_twin) X(no_wide) X(no_strip) \
  X(no_quant) X(no_devlist) X(no_ref_cache) X(no_strip_groups) X(u8_long_twin) X(long_twin) X(no_long) \
  X(no_requery) X(force_f32) X(no_long_p32) X(no_opt_margin) X(no_wave_prof) X(no_wave_window) X(no_first) X(no_long_save) X(u8_sample_short) X(no_wave_pieces) X(no_u8_early) X(no_wave_f16) X(no_devlist_by_id) X(trace)
#define MI355_SW_INT_OPTIONS(X) X(strip_r) X(slot) X(few_r) X(chunk) X(long_pipes) X(long_wgs) X(long_sub) X(long_r) X(long_groups) X(assume_cus) X(long_save_what)
struct Options {
#define X(n) bool n = false;
  MI355_SW_BOOL_OPTIONS(X)
#undef X
#define X(n) long n = 0;
  MI355_SW_INT_OPTIONS(X)
#undef X
  int fault_inject = 0;           // test hook, only through mi355_sw_set_option("fault_inject", "strip_stall" | "long_stall"): never from the environment
};
inline std::string option_env_name(const char *n) {
  std::string e = "MI355_SW_";
  for (const char *c = n; *c; ++c) e += (char)std::toupper((unsigned char)*c);
  return e;
}
inline Options options_from_env() {
  Options o;
#define X(n) o.n = std::getenv(option_env_name(#n).c_str()) != nullptr;
  MI355_SW_BOOL_OPTIONS(X)
#undef X
#define X(n) if (const char *e = std::getenv(option_env_name(#n).c_str())) o.n = std::atol(e);
  MI355_SW_INT_OPTIONS(X)
#undef X
  return o;
}
// 0: set; -1: unknown key.  value NULL / "" / "0" / "off" / "false" = off (0), anything else = on / the integer.
inline int option_set(Options &o, const char *key, const char *value) {
  std::string k;
  for (const char *c = key; *c; ++c) k += (char)std::tolower((unsigned char)*c);
  if (k.compare(0, 9, "mi355_sw_") == 0) k = k.substr(9);
  const std::string v = value ? value : "";
  const bool on = !(v.empty() || v == "0" || v == "off" || v == "false");
#define X(n) if (k == #n) { o.n = on; return 0; }
  MI355_SW_BOOL_OPTIONS(X)
#undef X
#define X(n) if (k == #n) { o.n = on ? std::atol(v.c_str()) : 0; return 0; }
  MI355_SW_INT_OPTIONS(X)
#undef X
  if (k == "fault_inject") { o.fault_inject = v == "strip_stall" ? 1 : (v == "long_stall" ? 2 : 0); return 0; }
  return -1;
}
inline const char *option_names() {
  return ""
#define X(n) #n ","
  MI355_SW_BOOL_OPTIONS(X) MI355_SW_INT_OPTIONS(X)
#undef X
  "fault_inject";
}
thread_local const Options *tl_opt = nullptr;
inline const Options &opt() {
  static const Options env = options_from_env();       // before any context exists on this thread
  return tl_opt ? *tl_opt : env;
}

// What the launch sizing needs to know about the device (hipGetDeviceProperties in mi355_sw_create; option assume_cus overrides the
// CU count for tests and partitioned devices), bound to the calling thread with the options.
struct DevInfo {
  int cus = 256;                  // multiProcessorCount (MI355X: 256; a CPX partition: 32)
  size_t lds = 160 * 1024;        // LDS bytes per CU
};
thread_local const DevInfo *tl_dev = nullptr;
inline int dev_cus() {
  const long v = opt().assume_cus;
  if (v > 0) return (int)std::min<long>(v, 4096);
  return tl_dev ? std::max(1, tl_dev->cus) : 256;
}
inline size_t dev_lds() { return tl_dev ? tl_dev->lds : (size_t)160 * 1024; }
// Set for the rest of a C-ABI call after a kernel whose workgroups wait for each other (sw_long_kernel with several workgroups per
// tile, the grouped sw_strip_kernel) reported an expired wait: the work is repeated on the instances whose waits stay inside one
// workgroup — resident by construction — instead of failing the call (a partitioned or shared device may not hold every
// workgroup of a launch at once, whatever the host assumed).
thread_local bool tl_no_wait = false;

// option `trace`: wall-clock of the host-side phases of every call on stderr (diagnostic)
struct HostTrace {
  const char *name;
  std::chrono::steady_clock::time_point t0;
  explicit HostTrace(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~HostTrace() {
    if (opt().trace) std::fprintf(stderr, "[mi355_sw] %-28s %9.3f ms\n", name,
                                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool view = false;              // aliases memory owned elsewhere: never freed here
  void alias(void *ptr) { p = ptr; cap = 0; view = true; }
  int ensure(size_t bytes) {
    if (view) return -1;
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
    cap = want;
    return 0;
  }
  void release() { if (p && !view) (void)hipFree(p); p = nullptr; cap = 0; view = false; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Pinned host staging (problem lists up, results down): pageable copies of tens of MB per launch were the largest
// single host cost of the many-small-alignments shape.
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return -1; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct RefData {
  DevBuf bytes, codes;
  size_t n = 0;
  int ncodes = 0;                 // incl. pad
  int code_of[256];
  uint8_t byte_of[256];
  uint64_t version = 0;           // bumped by every upload (caches keyed on the reference's content)
  void release() { bytes.release(); codes.release(); n = 0; ++version; }
};

struct QueryBatch {
  DevBuf bytes, lens, offs, sel;  // concatenated bytes (16-byte aligned starts), lengths, offsets, length-sorted ids
  DevBuf cum;                     // [nq + 1] exclusive prefix of the lengths in sorted order (device-built job lists)
  std::vector<int32_t> len;
  std::vector<int64_t> off;
  std::vector<int32_t> order;     // query ids sorted by length (stable)
  std::vector<int64_t> cumlen;    // host copy of `cum`
  size_t nq = 0;
  int maxlen = 0;
  uint64_t version = 0;           // bumped by every upload (caches keyed on the batch's content)
  void release() { bytes.release(); lens.release(); offs.release(); sel.release(); cum.release(); ++version; }
};

struct Range { int64_t lo, hi; };

// A few persistent helper threads per context (re-hashing the caller's reference while the call already runs on the
// resident copy): creating threads per call cost more than a short alignment.
class Helpers {
 public:
  static constexpr int kThreads = 4;
  ~Helpers() { stop(); }
  void submit(int w, std::function<void()> job) {
    Slot &s = slot[w];
    start(w);
    {
      std::lock_guard<std::mutex> g(s.m);
      s.job = std::move(job);
      s.busy = true;
    }
    s.cv.notify_all();
  }
  void wait(int w) {
    Slot &s = slot[w];
    if (!s.th.joinable()) return;
    std::unique_lock<std::mutex> g(s.m);
    s.cv.wait(g, [&] { return !s.busy; });
  }
  void stop() {
    for (Slot &s : slot) {
      if (!s.th.joinable()) continue;
      { std::lock_guard<std::mutex> g(s.m); s.quit = true; }
      s.cv.notify_all();
      s.th.join();
    }
  }

 private:
  struct Slot {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
  };
  Slot slot[kThreads];
  void start(int w) {
    Slot &s = slot[w];
    if (s.th.joinable()) return;
    s.th = std::thread([&s] {
      std::unique_lock<std::mutex> g(s.m);
      for (;;) {
        s.cv.wait(g, [&] { return s.busy || s.quit; });
        if (s.quit) return;
        std::function<void()> job = std::move(s.job);
        g.unlock();
        job();
        g.lock();
        s.busy = false;
        s.cv.notify_all();
      }
    });
  }
};

struct Hash128 {
  uint64_t a = 0, b = 0;
  bool operator==(const Hash128 &o) const { return a == o.a && b == o.b; }
  bool operator!=(const Hash128 &o) const { return !(*this == o); }
};


// One alignment's intermediate state on the host
struct Located {
  float score = 0;
  int64_t ix = 0, iy = 0;         // argmax, iy relative to the range start (1-based)
};

struct TraceOut {                  // consensus strings of one alignment: views into buffers the context keeps until its next call
  const char *cx = nullptr, *cy = nullptr;
  size_t len = 0;
  uint32_t pos = 0;
};

}  // namespace

// What mi355_sw_score_ranges leaves behind for mi355_sw_align_scored_range: the sweep's keys per (range, query) and the
// tile geometry they were made with, valid while reference, batch and scoring stay the same.
struct ScoredRanges {
  bool valid = false;
  const void *ref = nullptr, *batch = nullptr;   // the RefData / QueryBatch objects the sweep ran on, and their versions
  uint64_t ref_version = 0, batch_version = 0;
  mi355_sw_params params = {};
  std::vector<Range> ranges;
  std::vector<unsigned long long> keys;   // [nranges][nq]
  std::vector<char> qfast, qfloat;        // per query: swept by the score kernel; cell type of its key
  std::vector<int64_t> qchunk, qwarm;     // per query: sub-chunk length and warm-up margin of its bucket
  int fshift = 0;
  // winner-only sweeps (mi355_sw_best_range) on the sampled maximum: keys are lower bounds; the ranges that could hold the
  // greatest maximum were re-evaluated exactly and carry their first-maximum cell (one query per batch)
  bool sampled = false;
  std::vector<char> has_located;          // [nranges]
  std::vector<Located> located;           // [nranges]
};

// What the last sw_long_kernel launch saved for the finish (sw_long_kernel.h colsave / rowsave, host_saved.h): the H column
// in front of every sub-chunk and the bottom row of every strip, so that locate and traceback windows start from stored exact
// values instead of a zero border and a warm-up margin.  Valid for the (reference, batch, scoring, ranges) of that launch.
struct LongSaved {
  bool valid = false;
  const void *ref = nullptr, *batch = nullptr;
  uint64_t ref_version = 0, batch_version = 0;
  mi355_sw_params params = {};
  int qid = 0;
  std::vector<Range> ranges;
  int64_t chunk = 0, sub_len = 0, warm = 0;   // tile geometry of the launch (own columns, sub-chunk, warm-up actually used)
  int nstrips = 0, R = 0, spt = 0;            // strips of the query, rows per lane, sub-chunks per tile
  int64_t tiles_stride = 0, col_subs = 0, col_rows = 0, row_stride = 0;
  int fshift = 0;                             // saved values are H * 2^-fshift
};

// results of mi355_sw_batch_run_view: arrays the context owns until its next call
struct ViewStore {
  std::vector<float> score;
  std::vector<uint32_t> pos, cons_len;
  std::vector<int64_t> end_x, end_y;
  std::vector<const char *> cx, cy;
};

struct mi355_sw_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // downloads that need not hold up the kernels behind them (host_batch.h: the records of a batch)
  hipEvent_t ev[8] = {};
  std::string err;
  RefData ref;                    // resident reference (set_reference)
  QueryBatch batch;               // resident queries (batch_upload)
  RefData adhoc;                  // reference of the last mi355_sw_align-style call, kept while its content hash
  Hash128 adhoc_hash;             // matches (one-by-one driver loops pass the same reference every time)
  bool adhoc_valid = false;
  QueryBatch one;                 // the single query of such a call
  // scratch
  uint32_t flag_cap = 0;          // entries of `flags` (score_begin)
  bool first_valid = false;       // sw_sample_first ran in this score pass (ctx->first holds the offenders' first candidates)
  size_t first_settled = 0;       // queries over their candidate cap that their first candidates settled (uint8 engine)
  // per-query results of the running call (kept between calls: half a million alignments per call would otherwise fault in
  // 30 MB of fresh pages every time)
  std::vector<Located> loc_store;
  std::vector<TraceOut> tout_store;
  std::function<void()> while_device_works;   // host work of the running call that does not depend on the launches in flight: run in front of the next wait
  ViewStore *direct_view = nullptr;  // mi355_sw_batch_run_view: where exact_full_device may write finished alignments directly
  size_t devlist_direct = 0;         // ... and how many it wrote there
  std::vector<char> handled_store;   // per query of the running range: finished by the device-built lists (2: and written into direct_view)
  size_t devlist_done = 0;        // alignments of the running range the device-built lists finished (host_batch.h)
  size_t beyond_f16 = 0;          // sequences of the running call whose maximum was beyond the packed float16 pass's key range (host_batch.h)
  size_t left_window = 0;         // walks of the running call that left their decision window (host_batch.h) and were redone whole
  size_t requeried = 0;           // queries of the running call that were swept a second time on the exact instances
  size_t whole_again = 0;         // ... times the whole batch was (most of it exceeded its candidate cap)
  size_t candidates = 0;          // candidate sub-chunks the sampled / saturating sweeps of the call flagged
  const void *wlut_ref = nullptr; // reference (and its version) whose byte -> code tables are in `wlut` (sw_wave_prof_kernel)
  uint64_t wlut_version = 0;
  std::vector<uint8_t> h_wlut;
  int64_t long_margin = 0;        // > 0: warm-up margin the next sw_long_kernel launch must use at least (optimistic margins, host_score.h)
  float long_cert = -1.0f;        // maxima ABOVE this value were swept exactly by the last sw_long_kernel launch (-1: all of them)
  int64_t long_nsub = 0;          // sub-chunks per range of the last sampled sw_long_kernel launch (decodes its flag entries)
  bool long_launched = false;     // a sw_long_kernel launch since the last score_fetch (its status word is flags[1])
  std::string path;               // which kernels / pipelines the running call used (mi355_sw_last_path): space-separated tags
  DevInfo dev;                    // CU count and LDS per CU of this context's device
  size_t early_settled = 0;       // uint8-engine queries of the running call settled by the reference's first and last sub-chunks (no sweep)
  size_t wait_retries = 0;        // launches of the running call that were repeated on a non-waiting instance (tl_no_wait)
  LongSaved lsaved;
  DevBuf colsave, rowsave, pieces, recs;
  std::vector<uint8_t> h_pieces;  // host side of the piece table of the running call (host_batch.h)
  size_t saved_locates = 0, saved_traces = 0, saved_fallbacks = 0;   // finish steps of the running call that started from saved state / fell back
  DevBuf qcnt, sel2, gcnt, wlut, ckpt, first, keys, ranges, stab, ftab, ftab_s, htab, htab8, soloblk, flags, submax, lut, probs, dirs, outs_f, outs_i, cons, walkp, hmat, brow, wprobs, scan;
  // host sides of small per-call uploads: they must outlive the asynchronous copies, and the tables are only
  // sent again when they change
  std::vector<int64_t> h_ranges;
  std::vector<int16_t> h_stab;
  std::vector<float> h_ftab, h_ftab_s;
  int fshift = 0;                 // float32 score instance: cells hold H * 2^-fshift in this call
  std::vector<uint16_t> h_htab, h_htab8;
  // event pairs around the score launches of a call, read back after the call's first synchronisation
  std::vector<hipEvent_t> score_ev;
  size_t score_ev_used = 0;
  // D2H consensus buffers of the running call (TraceOut points into them); cleared when the next call starts
  std::vector<std::vector<char>> arenas;
  // pinned staging: problem / walk descriptors (up), small results (down), and a pool of consensus buffers that
  // live until the next call (cons_used of them are taken)
  PinBuf pin_probs, pin_walk, pin_out, pin_solo_up, pin_solo_down;
  std::vector<PinBuf> pin_cons;
  size_t cons_used = 0;
  double timings[6] = {0, 0, 0, 0, 0, 0};
  Options opts;                   // see Options above
  ViewStore view;
  ScoredRanges scored;
  Helpers helpers;
  mi355_sw_kernel_info last_kernel = {};   // score-kernel instance that swept the most cells in the running call
};

namespace {

// binds the calling thread to a context's options for the duration of a C-ABI call (nests)
struct OptScope {
  const Options *prev;
  const DevInfo *prev_dev;
  explicit OptScope(const mi355_sw_ctx *c) : prev(tl_opt), prev_dev(tl_dev) { if (c) { tl_opt = &c->opts; tl_dev = &c->dev; } }
  ~OptScope() { tl_opt = prev; tl_dev = prev_dev; if (!prev) tl_no_wait = false; }
};

#define HIPCHK(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return MI355_SW_ENODEV;                                                              \
    }                                                                                      \
  } while (0)

// One tag per kernel family / pipeline decision of the running call, each at most once (mi355_sw_last_path: what the parity
// tests assert a switch ENGAGED with — a switch that is silently ignored would still give the oracle's answers).
void path_note(mi355_sw_ctx *ctx, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void path_note(mi355_sw_ctx *ctx, const char *fmt, ...) {
  char buf[192];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  const std::string tag(buf), hay = " " + ctx->path + " ";
  if (ctx->path.size() > 8192 || hay.find(" " + tag + " ") != std::string::npos) return;
  if (!ctx->path.empty()) ctx->path += ' ';
  ctx->path += tag;
}

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

// Column counts derived from smax / gap ratios are computed in double and clamped before the integer cast: a tiny
// positive gap penalty (1e-20) would otherwise overflow it (undefined behaviour; INT64_MIN on x86).
constexpr int64_t kColsMax = (int64_t)1 << 50;
inline int64_t clamp_cols(double v) { return v >= (double)kColsMax ? kColsMax : (v <= 0 ? 0 : (int64_t)v); }

// Exactness margins (lemma L1; rows(): L2) under float32 ROUNDING.  A cell value is the rounded score of a chain of cells
// (follow the neighbour that achieved the maximum) that ends where a cell is 0.  With integer-valued scores every
// operation is exact and a chain with positive value over `rows` rows spans fewer than rows + smax*rows/g columns.
// With fractional scores each of the chain's operations may round by up to u = half an ulp of the largest value
// (<= smax*(rows+1) * 2^-24; taken as 2^-23 of it), so a chain with k_g gap steps still has
// smax*rows - g*k_g + (rows + k_g)*u > 0, i.e. k_g < rows*(smax + u)/(g - u): the same formulas with smax + u and g - u.
// A gap penalty within a factor 64 of u (H - g could stall or shrink by much less than g) has no usable margin.
struct Margin {
  double smax = 0, g = 0;
  bool finite() const { return g > 0 && std::isfinite(smax) && std::isfinite(g); }
  double slope() const { return finite() ? smax / g : 0.0; }
  int64_t cols(double rows) const { return finite() ? clamp_cols(rows + std::ceil(rows * smax / g)) : kColsMax; }
};
inline Margin make_margin(double smax, double g, bool exact, double rows) {
  if (!(g > 0) || !std::isfinite(smax) || !std::isfinite(g)) return Margin{smax, 0};
  if (exact) return Margin{smax, g};
  const double u = std::ldexp(std::max(smax, 0.0) * (rows + 1.0), -23);
  if (u * 64 > g) return Margin{smax, 0};
  return Margin{smax + u, g - u};
}

struct U8Params { int M, X, G; };
U8Params u8_params(const mi355_sw_params &p) {
  return {sat8(lut_or(p, 'A', 'A')), sat8(-lut_or(p, 'A', 'T')), sat8(p.gap)};   // :389-392
}

// float32 <-> float16 bit patterns (round to nearest even; the values here are small integers, exactly representable)
uint16_t half_bits(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t m = x & 0x7FFFFFu;
  if (((x >> 23) & 0xFF) == 0) return (uint16_t)sign;                     // zero / float32 subnormal
  if (e >= 31) return (uint16_t)(sign | 0x7C00u);                        // overflow -> inf
  if (e <= 0) {                                                          // float16 subnormal
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int sh = 14 - e;
    uint32_t h = m >> sh;
    const uint32_t rem = m & ((1u << sh) - 1), halfway = 1u << (sh - 1);
    if (rem > halfway || (rem == halfway && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}
float half_value(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FFu;
  float f;
  if (e == 0) { f = std::ldexp((float)m, -24); uint32_t b; memcpy(&b, &f, 4); b |= sign; memcpy(&f, &b, 4); return f; }
  const uint32_t x = sign | ((e == 31 ? 255u : e - 15 + 127) << 23) | (m << 13);
  memcpy(&f, &x, 4);
  return f;
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
  ++r.version;
  if (r.bytes.ensure(ny + 64) || r.codes.ensure(ny + 64)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(reference) failed");
  std::vector<uint8_t> codes(ny);
  for (size_t k = 0; k < ny; ++k) codes[k] = (uint8_t)r.code_of[u[k]];
  HIPCHK(ctx, hipMemcpyAsync(r.bytes.p, y, ny, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(r.codes.p, codes.data(), ny, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// 128-bit content hash of the caller's reference buffer: two structurally different 64-bit hashes computed in the same
// pass (xor-multiply-rotate lanes, and add-multiply-xorshift lanes with other constants), memory-bound (~3 ms for 50 MB).
// Both halves and the length must agree before a resident copy is used.
Hash128 content_hash_part(const char *p, size_t n) {
  uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ n, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0xD6E8FEB86659FD93ull};
  uint64_t g[4] = {0xA0761D6478BD642Full + n, 0xE7037ED1A0B428DBull, 0x8EBC6AF09C88C6E3ull, 0x589965CC75374CC3ull};
  size_t k = 0;
  for (; k + 32 <= n; k += 32) {
    uint64_t w[4];
    memcpy(w, p + k, 32);
    for (int l = 0; l < 4; ++l) {
      h[l] = (h[l] ^ w[l]) * 0x9FB21C651E98DF25ull; h[l] = (h[l] << 29) | (h[l] >> 35);
      g[l] = (g[l] + w[l]) * 0xD6E8FEB86659FD93ull; g[l] ^= g[l] >> 32;
    }
  }
  for (; k < n; ++k) {
    h[k & 3] = (h[k & 3] ^ (uint8_t)p[k]) * 0x9FB21C651E98DF25ull; h[k & 3] = (h[k & 3] << 29) | (h[k & 3] >> 35);
    g[k & 3] = (g[k & 3] + (uint8_t)p[k] + 1) * 0xD6E8FEB86659FD93ull; g[k & 3] ^= g[k & 3] >> 32;
  }
  Hash128 r;
  r.a = h[0]; r.b = g[0];
  for (int l = 1; l < 4; ++l) {
    r.a = (r.a ^ h[l]) * 0xBF58476D1CE4E5B9ull + (r.a >> 31);
    r.b = (r.b + g[l]) * 0x94D049BB133111EBull; r.b ^= r.b >> 29;
  }
  r.a ^= r.a >> 32;
  return r;
}

// Per-item host loops over a big batch (half a million small alignments per call) on a few PERSISTENT worker threads (one
// process-wide pool, started on first use): creating threads per loop (std::async) cost 0.1-0.2 ms each time — more than the
// loop itself on one rank's eighth of the UniProt-shaped batch, i.e. a per-call constant that did not shrink with the share.
// fn(k0, k1) must only touch items k0 <= k < k1.  Not re-entrant (a context serves one host thread; concurrent contexts
// take turns at the pool).
class WorkerPool {
 public:
  static WorkerPool &get() { static WorkerPool p; return p; }
  // runs job(t) for t = 1 .. nt - 1 on the workers and job(0) on the caller; returns when all are done.
  // The loops of one C-ABI call follow each other within a millisecond or so (defaults while the device works, results, view):
  // a worker that has finished SPINS on the generation counter for kSpinUs before it goes to sleep on the condition variable,
  // and the caller spins on the count of pending parts — waking seven sleepers through one mutex cost 0.05-0.1 ms per loop,
  // a third of a world-8 rank's whole call on the UniProt-shaped batch.
  void run(int nt, const std::function<void(int)> &job) {
    nt = std::min(nt, kWorkers + 1);
    if (nt <= 1) { job(0); return; }
    std::lock_guard<std::mutex> turn(turn_);
    {
      std::lock_guard<std::mutex> g(m_);
      if (th_.empty()) for (int w = 0; w < kWorkers; ++w) th_.emplace_back([this, w] { loop(w); });
    }
    job_ = &job;
    pending_.store(nt - 1, std::memory_order_relaxed);
    // generation and the number of workers it wants travel in ONE word: a worker that sits this generation out may still be
    // looking at it when the next one is posted
    state_.store(((state_.load(std::memory_order_relaxed) >> 8) + 1) << 8 | (uint64_t)(nt - 1), std::memory_order_release);
    {
      std::lock_guard<std::mutex> g(m_);                           // (a worker between its last look at gen_ and its wait holds m_)
      if (sleepers_ > 0) cv_.notify_all();
    }
    job(0);
    for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins) {
      if (spins < 4096) cpu_relax(); else std::this_thread::yield();
    }
    job_ = nullptr;
  }
  // wakes sleeping workers without giving them work: they spin for kSpinUs again.  Called in front of a wait for the device that
  // a pooled loop follows (host_batch.h), so that the loop does not start with the wake-up of seven sleepers.
  void nudge() {
    std::lock_guard<std::mutex> turn(turn_);
    std::lock_guard<std::mutex> g(m_);
    if (th_.empty() || sleepers_ == 0) return;
    state_.store(((state_.load(std::memory_order_relaxed) >> 8) + 1) << 8, std::memory_order_release);
    cv_.notify_all();
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> g(m_); quit_.store(true, std::memory_order_release); }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }

 private:
  static constexpr int kWorkers = 7;
  // (MI355_SW_POOL_SPIN_US in the environment, read once: 0 = sleep at once)
  const int kSpinUs = [] { const char *e = std::getenv("MI355_SW_POOL_SPIN_US"); return e && *e ? std::max(0, atoi(e)) : 1000; }();
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void loop(int w) {
    uint64_t seen = 0;
    for (;;) {
      // spin for a while, then sleep
      bool fresh = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0;; ++spins) {
        if (quit_.load(std::memory_order_acquire)) return;
        if (state_.load(std::memory_order_acquire) >> 8 != seen) { fresh = true; break; }
        cpu_relax();
        if ((spins & 255u) == 255u &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinUs) break;
      }
      if (!fresh) {
        std::unique_lock<std::mutex> g(m_);
        ++sleepers_;
        cv_.wait(g, [&] { return quit_.load(std::memory_order_acquire) || state_.load(std::memory_order_acquire) >> 8 != seen; });
        --sleepers_;
        if (quit_.load(std::memory_order_acquire)) return;
      }
      const uint64_t st = state_.load(std::memory_order_acquire);
      seen = st >> 8;
      if (w >= (int)(st & 0xFF)) continue;
      // (this worker is one of the generation's parts: run() cannot return, let alone post the next one, before it is done)
      const std::function<void(int)> *job = job_;
      (*job)(w + 1);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, turn_;
  std::condition_variable cv_;
  const std::function<void(int)> *job_ = nullptr;
  int sleepers_ = 0;
  std::atomic<int> pending_{0};
  std::atomic<uint64_t> state_{0};                                 // generation << 8 | workers wanted
  std::atomic<bool> quit_{false};
};

// items from which a per-item host loop of the running call goes to the pool (set per call by align_range_core)
thread_local size_t tl_pool_from = 131072;
struct PoolFromScope {                                             // (all loops of a call or none, see parallel_for)
  size_t was = tl_pool_from;
  explicit PoolFromScope(bool) { tl_pool_from = 49152; }
  ~PoolFromScope() { tl_pool_from = was; }
};

template <class F>
void parallel_for(size_t n, F fn) {
  const size_t pool_from = tl_pool_from;
  // Measured on one rank's eighth of the UniProt-shaped batch (70 k alignments, tools/c4_w8_time.py under MI355_SW_POOL_THREADS,
  // C-ABI call).  While the result loops scattered by query id (three loops per call): score + argmax 0.76 ms with every loop on
  // the calling thread, 1.03 ms on the pool; with traceback 1.40 against 1.31 ms — and 1.50 ms when only SOME loops of the call
  // took the pool: a call pools all of its loops or none.  With the records in id order (host_batch.h: one sequential loop + the
  // defaults written while the device works, which also wakes the workers): 0.64-0.69 against 0.76-0.81 ms, 1.00-1.08 against
  // 1.03-1.04 ms: calls pool from 49 152 items (PoolFromScope), anything else from 131 072.
  static const int nt_env = [] { const char *e = std::getenv("MI355_SW_POOL_THREADS"); return e && *e ? std::max(1, atoi(e)) : 0; }();
  if (n < pool_from && !nt_env) { fn((size_t)0, n); return; }
  const int nt = nt_env ? nt_env : 8;
  if (nt <= 1 || n < 1024) { fn((size_t)0, n); return; }
  const size_t step = (n + nt - 1) / nt;
  const Options *caller = tl_opt;                                  // the workers see the calling context's options, not the environment's
  const DevInfo *caller_dev = tl_dev;
  WorkerPool::get().run(nt, [&](int t) {
    const Options *was = tl_opt; const DevInfo *was_dev = tl_dev;
    tl_opt = caller; tl_dev = caller_dev;
    fn(std::min(n, (size_t)t * step), std::min(n, (size_t)(t + 1) * step));
    tl_opt = was; tl_dev = was_dev;
  });
}

inline Hash128 combine_hash(Hash128 r, const Hash128 &o) {
  r.a = (r.a ^ o.a) * 0xBF58476D1CE4E5B9ull + (r.a >> 29);
  r.b = (r.b + o.b) * 0x94D049BB133111EBull; r.b ^= r.b >> 31;
  return r;
}

// Hash of a whole buffer: four independent quarters (hashed on helper threads when the buffer is large), combined.
Hash128 content_hash(const char *p, size_t n) {
  if (n < ((size_t)4 << 20)) return content_hash_part(p, n);
  const size_t q = (n / 4) & ~(size_t)31;
  std::future<Hash128> f[3];
  for (int k = 0; k < 3; ++k) f[k] = std::async(std::launch::async, content_hash_part, p + (size_t)(k + 1) * q, k == 2 ? n - 3 * q : q);
  Hash128 r = content_hash_part(p, q);
  for (int k = 0; k < 3; ++k) r = combine_hash(r, f[k].get());
  return r;
}

// option no_ref_cache: never reuse the resident copy of a single-alignment call's reference (every call uploads).
bool adhoc_cache_enabled() { return !opt().no_ref_cache; }

// Reference of a single-alignment call: re-used from the previous call when length and 128-bit content hash match.
// `known_hash`: the caller has already hashed y.
int adhoc_reference(mi355_sw_ctx *ctx, const char *y, size_t ny, const RefData **out, const Hash128 *known_hash = nullptr) {
  const Hash128 h = known_hash ? *known_hash : (adhoc_cache_enabled() ? content_hash(y, ny) : Hash128());
  if (!(adhoc_cache_enabled() && ctx->adhoc_valid && ctx->adhoc.n == ny && ctx->adhoc_hash == h)) {
    ctx->adhoc_valid = false;
    int rc = upload_reference(ctx, ctx->adhoc, y, ny);
    if (rc) return rc;
    ctx->adhoc_hash = h;
    ctx->adhoc_valid = adhoc_cache_enabled();
  }
  *out = &ctx->adhoc;
  return 0;
}

// One-by-one loops against one reference (src/sw_solve_big.cpp:78-92: a new aligner per read, same reference):
// hashing the reference costs as much as aligning against it, so the call starts on the resident copy while the
// context's helper threads re-hash the caller's buffer (four independent quarters when it is large), and is repeated
// on a fresh upload in the rare case the content changed.
struct AdhocSpeculation {
  Hash128 part[Helpers::kThreads];
  int nparts = 0;
  bool active = false;
};
int adhoc_begin(mi355_sw_ctx *ctx, const char *y, size_t ny, const RefData **out, AdhocSpeculation &sp) {
  if (adhoc_cache_enabled() && ctx->adhoc_valid && ctx->adhoc.n == ny && ny >= ((size_t)64 << 10)) {
    // the same split as content_hash(): one part below 4 MB, else four quarters
    if (ny < ((size_t)4 << 20)) {
      sp.nparts = 1;
      ctx->helpers.submit(0, [&sp, y, ny] { sp.part[0] = content_hash_part(y, ny); });
    } else {
      const size_t q = (ny / 4) & ~(size_t)31;
      sp.nparts = 4;
      for (int k = 0; k < 4; ++k) {
        const char *p0 = y + (size_t)k * q;
        const size_t len = k == 3 ? ny - 3 * q : q;
        ctx->helpers.submit(k, [&sp, k, p0, len] { sp.part[k] = content_hash_part(p0, len); });
      }
    }
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
  for (int k = 0; k < sp.nparts; ++k) ctx->helpers.wait(k);
  Hash128 h = sp.part[0];
  for (int k = 1; k < sp.nparts; ++k) h = combine_hash(h, sp.part[k]);
  if (h == ctx->adhoc_hash) return true;
  ctx->adhoc_valid = false;
  rc = adhoc_reference(ctx, y, ny, out, &h);
  return false;
}

// stable order of the batch by length: a counting sort when the batch is large (561 k UniProt sequences: 40 ms -> 3 ms)
void sort_queries_by_length(QueryBatch &q, size_t mx) {
  const size_t n = q.nq;
  q.order.resize(n);
  if (n >= 4096 && mx <= ((size_t)1 << 22)) {
    std::vector<uint32_t> start(mx + 2, 0);
    for (size_t k = 0; k < n; ++k) start[(size_t)q.len[k] + 1]++;
    for (size_t l = 1; l < start.size(); ++l) start[l] += start[l - 1];
    for (size_t k = 0; k < n; ++k) q.order[start[(size_t)q.len[k]]++] = (int32_t)k;
  } else {
    for (size_t k = 0; k < n; ++k) q.order[k] = (int32_t)k;
    std::stable_sort(q.order.begin(), q.order.end(), [&](int32_t a, int32_t b) { return q.len[a] < q.len[b]; });
  }
  q.cumlen.resize(n + 1);
  q.cumlen[0] = 0;
  for (size_t k = 0; k < n; ++k) q.cumlen[k + 1] = q.cumlen[k] + q.len[q.order[k]];
}

// lengths / offsets / sorted ids / prefix sums to the device (the bytes go separately)
int upload_query_index(mi355_sw_ctx *ctx, QueryBatch &q, size_t tot) {
  const size_t n = q.nq;
  if (q.bytes.ensure(tot + 16) || q.lens.ensure(n * 4 + 16) || q.offs.ensure(n * 8 + 16) || q.sel.ensure(n * 4 + 16) ||
      q.cum.ensure((n + 1) * 8 + 16))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(queries) failed");
  HIPCHK(ctx, hipMemcpyAsync(q.cum.p, q.cumlen.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(q.lens.p, q.len.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(q.offs.p, q.off.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(q.sel.p, q.order.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(q.bytes.as<uint8_t>() + tot, 0, 16, ctx->stream));
  return 0;
}

int upload_queries(mi355_sw_ctx *ctx, QueryBatch &q, size_t n, const char *const *xs, const size_t *nxs) {
  HostTrace trace_("upload_queries");
  q.nq = n;
  ++q.version;
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
  sort_queries_by_length(q, mx);
  // staging copy (16-byte aligned starts; the padding is never read): helper threads share a large batch
  std::unique_ptr<uint8_t[]> host_buf(new uint8_t[tot + 16]);
  uint8_t *host = host_buf.get();
  auto copy_part = [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; ++k) memcpy(host + (size_t)q.off[k], xs[k], nxs[k]); };
  if (tot >= ((size_t)8 << 20) && n >= 8) {
    std::future<void> parts[3];
    const size_t step = n / 4;
    for (int t = 0; t < 3; ++t) parts[t] = std::async(std::launch::async, copy_part, (size_t)(t + 1) * step, t == 2 ? n : (size_t)(t + 2) * step);
    copy_part(0, step);
    for (auto &f : parts) f.get();
  } else {
    copy_part(0, n);
  }
  int rc = upload_query_index(ctx, q, tot);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(q.bytes.p, host, tot, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// The batch as ONE contiguous buffer + n + 1 ascending offsets (what a multi-FASTA reader has; the reference concatenates each
// sequence's lines into one string, src/mpi_sw_solve_uniprot.cpp:97-110): no per-sequence pointers, no staging copy — the
// caller's bytes go to the device as they are (in a few large pieces, so that the first ones travel while the index is sorted
// on the host), sequence k = buf[offsets[k], offsets[k + 1]).
int upload_queries_packed(mi355_sw_ctx *ctx, QueryBatch &q, size_t n, const char *buf, const int64_t *offsets) {
  HostTrace trace_("upload_queries_packed");
  // the offsets are validated BEFORE anything is sized from them or leaves the caller's buffer
  if (n && offsets[0] < 0) return fail(ctx, MI355_SW_EINVAL, "negative offset");
  for (size_t k = 0; k < n; ++k) {
    const int64_t len = offsets[k + 1] - offsets[k];
    if (len < 0 || len > 0x3fffffff) return fail(ctx, MI355_SW_EINVAL, "offsets must ascend (sequence longer than 2^30 or negative length)");
  }
  const size_t base = n ? (size_t)offsets[0] : 0, tot = n ? (size_t)(offsets[n] - offsets[0]) : 0;
  if (q.bytes.ensure(tot + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(queries) failed");
  // every exit after the first enqueue waits for the stream: no copy out of caller-owned memory is in flight when the call returns
  auto leave = [&](int rc) { (void)hipStreamSynchronize(ctx->stream); if (rc) q.nq = 0; return rc; };
  const size_t piece = (size_t)32 << 20;
  for (size_t at = 0; at < tot; at += piece)                         // asynchronous to the host work below
    if (hipMemcpyAsync(q.bytes.as<uint8_t>() + at, buf + base + at, std::min(piece, tot - at), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      return leave(fail(ctx, MI355_SW_ENODEV, "hipMemcpyAsync(queries) failed"));
  q.nq = n;
  ++q.version;
  q.len.resize(n);
  q.off.resize(n);
  size_t mx = 0;
  for (size_t k = 0; k < n; ++k) {
    const int64_t len = offsets[k + 1] - offsets[k];
    q.len[k] = (int32_t)len;
    q.off[k] = offsets[k] - (int64_t)base;
    mx = std::max(mx, (size_t)len);
  }
  q.maxlen = (int)mx;
  sort_queries_by_length(q, mx);
  return leave(upload_query_index(ctx, q, tot));
}

}  // namespace
