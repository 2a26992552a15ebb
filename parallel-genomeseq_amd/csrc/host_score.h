// host_score.h — score pass: score table, length buckets, tile geometry, sw_score_kernel launches
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
namespace {

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
  std::vector<uint16_t> htab; // [256][ncodes] the same scores as float16 bits scaled by 1/2048 (kSemF16), empty when
                              // an entry does not fit (|s| <= 2048; padding -16384)
  std::vector<uint16_t> htab8; // uint8 engine: the scores as float16 bits scaled by 1/256 (kSemU8H)
  std::vector<float> ftab;    // [256][ncodes]
  // margins for queries of up to `rows` rows: exact arithmetic for integer scores (and the uint8 engine), widened by
  // the float32 rounding slack otherwise
  Margin margin(double rows) const { return make_margin(smaxf, gapf, integral, rows); }
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
  bool comb = false;          // twin on 16-lane tiles, small alphabet: profile indexed by the pair of codes (COMB)
  bool unsat = false;         // uint8 engine swept by a float-engine instance WITHOUT saturation, maxima clamped at 255
  bool sampled = false;       // running maximum folded every 4th step (sw_score_kernel MK = 4): sub-chunk values are lower bounds
                              // within 3 gaps of the truth; sub-chunks within that slack of the key are re-evaluated exactly
  bool opt_margin = false;    // sw_long_kernel: optimistic warm-up margin, certified afterwards (long_score_launch)
  bool longp = false;         // lone long query on sw_long_kernel: the strips of a tile pipelined over the wavefronts of a workgroup
  int nstrips = 0;            // ... strips (= wavefronts) per tile
  bool satflag = false;       // float engine swept on float16 cells BEYOND their exact range (the sweep saturates at 2048):
                              // sub-chunks that reach the cap are flagged and re-evaluated exactly (locate_saturated)
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

// Whole-wavefront tiles (queries beyond 512 rows): rows per lane of one strip up to 2048 rows, or of the strips of
// a longer query — the choice with the fewest padded rows (more rows per lane on ties: fewer strips).
void pick_shape64(int len, int &R, bool &strips) {
  static const int one[] = {10, 12, 16, 20, 24, 32};
  strips = len > 2048;
  if (!strips) {
    for (int r : one) if (64 * r >= len) { R = r; return; }
  }
  static const int many[] = {20, 24, 32};
  { const long v = opt().strip_r; if (v == 20 || v == 24 || v == 32) { R = (int)v; return; } }   // tuning aid
  int64_t best = -1;
  for (int r : many) {
    const int64_t rows = (int64_t)((len + 64 * r - 1) / (64 * r)) * 64 * r;
    if (best < 0 || rows <= best) { best = rows; R = r; }
  }
}

// (SL, R) with the fewest padded rows; ties go to 8 lanes (fewer per-step overhead ops per cell)
void pick_shape(int len, int &SL, int &R) {
  SL = 16; R = len < 1 ? 2 : pick_R(len);
  const int r8 = len < 36 ? 0 : pick_R8(len);
  if (r8 && 8 * r8 <= 16 * R) { SL = 8; R = r8; }
  if (opt().slot == 16) { SL = 16; R = len < 1 ? 2 : pick_R(len); }   // tuning aid
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
    // packed float16 instance (kSemU8H): cells hold (H + 1) / 256, scores are s / 256, padding -64
    f.htab8.resize(f.stab.size());
    for (size_t k = 0; k < f.stab.size(); ++k) f.htab8[k] = half_bits((float)f.stab[k] / 256.0f);
    // unsaturated sweep (kSemF16 with clamped publishing, make_buckets): cells hold H / 2048
    f.htab.resize(f.stab.size());
    for (size_t k = 0; k < f.stab.size(); ++k) f.htab[k] = half_bits((float)f.stab[k] / kF16Scale);
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
  if (p.semantics == MI355_SW_F32 && f.integral && f.gap <= 2040) {
    bool fits = true;
    for (int16_t v : f.stab) fits = fits && (v == (int16_t)kPadScore || (v >= -2048 && v <= 2048));
    if (fits) {
      f.htab.resize(f.stab.size());
      for (size_t k = 0; k < f.stab.size(); ++k) f.htab[k] = half_bits((float)f.stab[k] / kF16Scale);   // cells hold H / 2048
    }
  }
  return f;
}

size_t profile_lds_bytes(int ncodes, int R, int SL = 16, bool twin = false, bool comb = false) {
  if (comb) return (size_t)ncodes * (size_t)ncodes * (size_t)std::max(16, SL) * lane_stride(R) * 4;
  return (size_t)ncodes * (size_t)std::max(16, SL) * lane_stride(twin ? R / 2 : R) * 4;
}
// twin tiles on 16-lane slots with the profile indexed by code PAIRS: while the ncodes^2 entries leave room for two
// workgroups per CU (DNA incl. N and the pad code: 36 pairs)
constexpr size_t kCombLdsMax = 64 * 1024;
bool comb_ok(int ncodes, int R) {
  return !opt().no_comb && profile_lds_bytes(ncodes, R, 16, true, true) <= kCombLdsMax;
}

// sw_long_kernel (a lone query beyond 2048 rows): rows per lane and strips per tile such that the whole float16 profile,
// the rings and the sub-chunk maxima fit the CU's LDS with at most sixteen wavefronts — the shape with the fewest padded rows.
inline size_t long_lds_max() { return std::min<size_t>((size_t)156 * 1024, dev_lds() - 4096); }   // (a CU's LDS minus the static part)
constexpr int kLongSubsMax = 1024;            // sub-chunk maxima a tile keeps in LDS
constexpr int kLongMK = 8;                    // sw_long_kernel folds the running maximum every kLongMK-th step when it samples: sub-chunk values
                                              // are lower bounds within (kLongMK - 1) gaps (a cell holding M passes M - k g along its row)
bool long_shape(int ncodes, int len, int &R, int &nstrips) {
  int64_t best = -1;
  const long forced = opt().long_r;                                       // tuning aid
  for (int r : {20, 24, 32}) {
    if ((forced == 20 || forced == 24 || forced == 32) && r != forced) continue;
    const int ns = (len + 64 * r - 1) / (64 * r);
    if (ns < 1) continue;
    // one workgroup with the float16 profile, or up to eight with a float32 profile each (long_score_launch decides)
    const bool one_wg = ns <= long_max_waves(r) && long_lds_bytes(ncodes, ns, 1, r, kLongSubsMax) <= long_lds_max();
    bool many_wg = false;
    for (int g = 1; g <= 8 && !many_wg; g *= 2) {
      const int spg = (ns + g - 1) / g;
      many_wg = spg <= long_max_waves(r) && long_lds_bytes(ncodes, spg, 1, r, kLongSubsMax, true) <= long_lds_max();
    }
    if (!one_wg && (!many_wg || opt().no_long_p32 || tl_no_wait)) continue;   // (tl_no_wait: only layouts whose waits stay inside one workgroup)
    const int64_t rows = (int64_t)ns * 64 * r;
    if (best < 0 || rows < best) { best = rows; R = r; nstrips = ns; }    // ties: fewer rows per lane = more wavefronts (measured,
                                                                          // 10 kbp x 250 Mbp: 8 x R=20 strips 244 ms, 5 x R=32 strips 306 ms)
  }
  return best > 0;
}

// Tile shape of a lone short query on twin tiles: 16 lanes x R rows.  (Round 4 tried 8 lanes x 19 rows with the code-pair profile
// — half the per-step work per cell, 152 instead of 160 rows: 0.646 against 0.658 ms per 150 bp x 50 Mbp sweep, and 0.157 against
// 0.098 ms at 1 Mbp, where half as many wavefronts leave CUs idle: dropped, CHANGELOG.md.)
void twin_shape(const RefData &ref, Bucket &b) { b.SL = 16; b.R = pick_R(b.maxlen); b.comb = comb_ok(ref.ncodes, b.R); }

// Length classes of the batch: one bucket per kernel instance (R), plus one strip-mined bucket.
bool sampled_instance(int SL, int R) {
  if (SL == 8) return R == 13 || R == 16 || R == 19 || R == 26 || R == 32;
  if (SL == 16 || SL == 64) return R == 10 || R == 12 || R == 16 || R == 20 || R == 24 || R == 32;
  return false;
}

std::vector<Bucket> make_buckets(const RefData &ref, const QueryBatch &q, const ScoreTable &t, const mi355_sw_params &p, int64_t n,
                                 bool allow_sat = false, bool allow_sample = false) {
  std::vector<Bucket> out;
  // queries beyond 512 rows: whole-wavefront tiles (64 lanes x R rows: one strip up to 2048 rows, 2048-row
  // strips beyond) when the 64-position profile fits LDS, else 16-lane tiles in 512-row strips
  const bool wide_ok = profile_lds_bytes(ref.ncodes, 32, 64) <= kProfileLdsMax && !opt().no_wide;
  for (size_t pos = 0; pos < q.nq; ++pos) {
    const int len = q.len[q.order[pos]];
    bool strips = false;
    int SL = 16, R = 32;
    if (len <= kMaxRowsFast) pick_shape(len, SL, R);
    else if (wide_ok) { SL = 64; pick_shape64(len, R, strips); }
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
    const bool twin_ok = b.count == 1 && b.SL == 64 && !opt().no_twin;
    const bool twin16_ok = b.count == 1 && b.maxlen >= 1 && b.maxlen <= kMaxRowsFast && !opt().no_twin;
    if (p.semantics == MI355_SW_U8SAT) {
      // lone query: two of its tiles per packed register on whole-wavefront tiles, else one query per register
      b.twin = twin_ok;
      b.sem = b.count == 1 && !b.twin ? kSemF32U8 : kSemU8;
      // the same rule in packed float16 cells scaled by 1/256 (4.25 instead of 5.5 ops per cell); values never leave
      // 0..255, so this holds for every query length
      if (b.sem == kSemU8 && !opt().no_f16) b.sem = kSemU8H;
      // (DESIGN.md §3.3 lemma L7.)  The score pass only has to deliver, per query, the maximum and the FIRST sub-chunk that reaches it.  Sweep with
      // the FLOAT engine's packed float16 cell (kSemF16: max(0, NW + s, W - G, N - G) on the integer scores M, -X, G,
      // whose clamped add saturates the diagonal term at 2048 instead of 255) and clamp what is published at 255: left of
      // the first cell that reaches 255 neither rule has saturated, so both recurrences agree there and that cell holds
      // >= 255 in both; the maximum is min(255, swept maximum) and its first sub-chunk is the same.  Later sub-chunks may
      // differ, but cannot change (maximum, first sub-chunk).  Every value stays an integer <= 2048, exact in float16
      // for ANY query length.  3.5 ops per cell pair instead of 4.25 (kSemU8H) or 6 per cell (lone query, float32).
      // locate and traceback keep the saturating rule.
      if (!opt().no_unsat && !opt().no_f16 && !t.htab.empty() && t.gap <= 2040) {
        if (b.count >= 2) {
          b.sem = kSemF16; b.unsat = true; b.twin = false;
          // sampled maximum: every cell that holds the uint8 maximum reads at least that in the unsaturated sweep (a clamp at
          // 255 only lowers values), hence >= it - 3 gaps at the next folded step: its sub-chunk is a candidate
          // ... provided a random background stays clear of 255: with cheap gaps it grows with the read (about 0.2 M per
          // row at 3 / -3 / 2), longer reads reach 255 everywhere and every sub-chunk would be a candidate (measured: 1000 bp
          // reads overflow the flag budget and the call repeats the sweep unsampled)
          // Longer reads (beyond ~250 bp at 3 / -3 / 2) reach 255 against ANY background, so every sub-chunk is a candidate of
          // every read — but a key at the cap is decided by the FIRST candidates in storage order (sw_sample_first: the first
          // eight in ascending order plus the wrapped triangle's two sub-chunks, align_range_core), not by all of them: the
          // bucket keeps the three-op cell and the filter's per-query cap keeps the list short (round 3 switched the sampling
          // off there: 3.5 ops, 18-19 TCUPS against the float engine's 21).  Option u8_sample_short restores that limit.
          b.sampled = allow_sample && !b.strips && sampled_instance(b.SL, b.R) && !opt().no_sample &&
                      ref.ncodes - 1 >= 4 &&             // (two- and three-letter alphabets: random matches every other column)
                      (!opt().u8_sample_short || 0.3 * (double)t.smax * (double)b.maxlen + 3.0 * (double)t.gap < 230.0);
        }
        else if (twin16_ok && !b.strips && b.SL != 64) {
          b.sem = kSemF16; b.unsat = true; b.twin = true; twin_shape(ref, b);
        }
        // a lone LONG query (whole-wavefront tiles): the float engine's float32 cell (three ops per cell, one query per
        // register; exact below 2^24), unsaturated and clamped at 255 the same way, sweeps it faster than two tiles per packed
        // float16 register, whose halves need two profile reads and a merge op per row (config 5: 265 against 288 ms;
        // 4096 rows x 50 Mbp: 26 against 34 ms).  MI355_SW_U8_LONG_TWIN=1 restores the twin tiles.
        else if (b.count == 1 && b.SL == 64 && (double)t.smax * b.maxlen < 1.6e7 && !opt().u8_long_twin) {
          b.sem = kSemF32; b.unsat = true; b.twin = false;
          int R = 0, ns = 0;
          if (b.strips && !opt().no_long && long_shape(ref.ncodes, b.maxlen, R, ns)) { b.longp = true; b.R = R; b.nstrips = ns; }
        }
        else if (twin_ok) { b.sem = kSemF16; b.unsat = true; b.twin = true; }
      }
    } else {
      // packed 16-bit cells when scores are small integers and the score bound fits; float32 cells otherwise
      // (option force_f32: the float32 instance for everything — what bench.py's parity_check runs the resident batch on again)
      const bool fits = !opt().force_f32 && t.integral && (int64_t)t.smax * std::min<int64_t>(b.maxlen, std::max<int64_t>(n, 1)) + t.smax <= 32000;
      b.sem = fits ? kSemI16 : kSemF32;
      // small scores on short reads: packed float16 cells (clamped add + three-input maximum: 3.5 instead of 4.5 ops per cell)
      if (fits && !t.htab.empty() && !b.strips && b.count >= 2 &&
          (int64_t)t.smax * b.maxlen + t.smax <= 2040 && !opt().no_f16) {
        b.sem = kSemF16;
        // the running maximum every 4th step, the sub-chunks within 3 gaps of the key re-evaluated exactly (sw_score_kernel MK; lemma L5)
        b.sampled = allow_sample && sampled_instance(b.SL, b.R) && !opt().no_sample;
      }
      // (Lemma L6.)  Beyond float16's exact range (reads above 680 bp at match 3) the packed int16 cell costs 4.5 ops.  The float16
      // cell still sweeps them when its clamp is allowed to SATURATE the values at 2048: if the true maximum M is below
      // 2048 nothing saturated and the sweep is exact; if not, the first cell holding M has 2048 in the sweep (every
      // suffix of its best path has a positive sum — else an earlier cell would hold M too — so a walk capped at 2048
      // that has reached the cap is back at it there), so its sub-chunk is among the FLAGGED ones (sub-chunk maximum at
      // the cap), which locate_saturated re-evaluates exactly: few windows per read instead of the whole reference.
      // Only where a random background stays well below the cap (<= 2048 rows) and the exact kernel takes the scoring.
      else if (allow_sat && fits && !t.htab.empty() && !b.strips && b.count >= 2 && b.maxlen <= 2048 &&
               !opt().no_f16 && !opt().no_satflag) {
        b.sem = kSemF16; b.satflag = true;
        // with the sampled maximum the flags come from the filter alone (threshold: key - slack, key <= cap): a cell that holds
        // the true maximum reads 2048 in the saturating sweep and >= 2048 - 3 gaps at the next folded step
        b.sampled = allow_sample && sampled_instance(b.SL, b.R) && !opt().no_sample;
      }
      // a lone query would fill both halves of every packed register with itself: the float32 instance (one query per
      // slot, exact for integer scores below 2^24) sweeps it faster — also than two of its tiles per packed integer
      // register (config 5: 282 ms against 338 ms), which remains the uint8 engine's way (its cells are float16)
      // ... except a short one with small scores: two of its TILES per packed float16 register on 16-lane tiles
      if (b.count == 1 && b.sem == kSemI16 && twin16_ok && !t.htab.empty() && !b.strips && b.SL != 64 &&
          (int64_t)t.smax * b.maxlen + t.smax <= 2040 && !opt().no_f16) {
        b.sem = kSemF16; b.twin = true; twin_shape(ref, b);
      }
      else if (b.count == 1 && b.sem == kSemI16 && twin_ok && opt().long_twin) b.twin = true;   // A/B switch
      else if (b.count == 1 && b.sem == kSemI16 && (double)t.smax * b.maxlen < 1.6e7) b.sem = kSemF32;
      else if (b.count == 1 && b.sem == kSemI16 && twin_ok) b.twin = true;
      // a lone query beyond 2048 rows on float32 cells, integer scores within float16's exact range: the strips of a tile
      // pipelined over the wavefronts of a workgroup (sw_long_kernel.h) instead of one wavefront per tile
      if (b.sem == kSemF32 && b.count == 1 && b.SL == 64 && b.strips && !b.twin && t.integral && !t.htab.empty() && !opt().no_long) {
        int R = 0, ns = 0;
        if (long_shape(ref.ncodes, b.maxlen, R, ns)) { b.longp = true; b.R = R; b.nstrips = ns; }
      }
      // a lone long query on float32 cells (config 5): the same sampled maximum; integer scores only (exact arithmetic)
      if (b.sem == kSemF32 && b.count == 1 && b.SL == 64 && !b.twin && t.integral && allow_sample &&
          (b.strips ? (b.R == 20 || b.R == 24 || b.R == 32) : sampled_instance(64, b.R)) && !opt().no_sample)
        b.sampled = true;
      // ... and batches on float32 cells (fractional scoring; scores beyond 16 bits): the decay bound then holds up to the
      // rounding of three subtractions, which the filter's slack allows for (score_launch)
      if (b.sem == kSemF32 && b.count >= 2 && !b.twin && !b.strips && (b.SL == 8 || b.SL == 16) && sampled_instance(b.SL, b.R) &&
          allow_sample && !opt().no_sample)
        b.sampled = true;
    }
    const Margin mg = t.margin(b.maxlen);
    if (!(mg.smax > 0)) b.warm = 0;
    else b.warm = mg.cols(b.maxlen);                                  // DESIGN.md §3.3 (kColsMax: no finite margin)
    b.warm = std::min(kColsMax, (b.warm + 63) / 64 * 64);
  }
  return out;
}

// May this bucket's queries be swept by the score kernel over a reference range of n columns?
bool bucket_fast_ok(const RefData &ref, const ScoreTable &t, const Bucket &b, int64_t n, const mi355_sw_params &p) {
  if (!t.ok || n < 1 || b.maxlen < 1) return false;
  if (!b.longp && profile_lds_bytes(ref.ncodes, b.R, b.SL) > kProfileLdsMax) return false;  // alphabet too large for this shape
  // codes travel as bytes: with all 256 byte values present the pad code (256) would alias code 0
  if (ref.ncodes > 256) return false;
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

// VALU instructions per cell and lane of the instance's inner loop (cost model of DESIGN.md §3.4): per step and
// lane the recurrence ops of R rows, the running-maximum ops, and the per-step overhead (DPP move, profile address,
// code extract, border mask on 8-lane tiles), over the cells a register row holds (two for the packed instances).
double valu_ops_per_cell(const Bucket &b) {
  const int R = b.R;
  const double over = b.SL == 8 ? 4.0 : 3.0;
  double per_step;
  int cells_per_row = 2;
  switch (b.sem) {
    case kSemF32:   per_step = 3.0 * R + (b.sampled ? (b.longp ? 1.0 / kLongMK : 0.25) : 1.0) * ((R + 1) / 2) + 1 + over; cells_per_row = 1; break;   // add clamp, max3, sub; max3 per two cells
    case kSemF32U8: per_step = 6.0 * R + (R + 1) / 2 + over; cells_per_row = 1; break;          // add, min, max, sub, max, max
    case kSemF16:   per_step = 3.0 * R + (b.sampled ? 0.25 : 1.0) * ((R + 1) / 2) + 1 + over; break;
    case kSemU8H:   { const int odd = R / 2; per_step = 4.0 * R + odd / 2 + odd % 2 + R % 2 + over; break; }
    case kSemU8:    per_step = 5.0 * R + (R + 1) / 2 + R % 2 + over; break;
    default:        per_step = 4.0 * R + (R + 1) / 2 + R % 2 + over; break;
  }
  if (b.twin && !b.comb) per_step += R;                                                         // one v_perm_b32 per row
  return per_step / (double)(cells_per_row * R);
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
    switch (R) {
#define CASE_TS(r) case r: launch_score(sw_score_kernel<r, SEM, true, 64, true>, grid, shmem, st, a); return 0;
      CASE_TS(20) CASE_TS(24) CASE_TS(32)
#undef CASE_TS
    }
    return -1;
  }
  switch (R) {
#define CASE_T(r) case r: launch_score(sw_score_kernel<r, SEM, false, 64, true>, grid, shmem, st, a); return 0;
    CASE_T(10) CASE_T(12) CASE_T(16) CASE_T(20) CASE_T(24) CASE_T(32)
#undef CASE_T
  }
  return -1;
}

// a lone short query on 16-lane tiles, two TILES of it per packed float16 register
int launch_score_twin16(int R, bool comb, dim3 grid, size_t shmem, hipStream_t st, const ScoreArgs &a) {
  if (comb) {
    switch (R) {
#define CASE_C16(r) case r: launch_score(sw_score_kernel<r, kSemF16, false, 16, true, true>, grid, shmem, st, a); return 0;
      CASE_C16(2) CASE_C16(4) CASE_C16(6) CASE_C16(8) CASE_C16(10) CASE_C16(12) CASE_C16(16) CASE_C16(20) CASE_C16(24) CASE_C16(32)
#undef CASE_C16
    }
    return -1;
  }
  switch (R) {
#define CASE_T16(r) case r: launch_score(sw_score_kernel<r, kSemF16, false, 16, true>, grid, shmem, st, a); return 0;
    CASE_T16(2) CASE_T16(4) CASE_T16(6) CASE_T16(8) CASE_T16(10) CASE_T16(12) CASE_T16(16) CASE_T16(20) CASE_T16(24) CASE_T16(32)
#undef CASE_T16
  }
  return -1;
}

// packed float16 cells: reads whose scores stay within +-2048, 8- and 16-lane tiles
template <int SEM>
int launch_score_f16(int R, int SL, bool strips, dim3 grid, size_t shmem, hipStream_t st, const ScoreArgs &a) {
  if (strips) {
    if (SL == 16) { if (R != 32) return -1; launch_score(sw_score_kernel<32, SEM, true, 16>, grid, shmem, st, a); return 0; }
    if (SL != 64) return -1;
    switch (R) {
#define CASE_HS(r) case r: launch_score(sw_score_kernel<r, SEM, true, 64>, grid, shmem, st, a); return 0;
      CASE_HS(20) CASE_HS(24) CASE_HS(32)
#undef CASE_HS
    }
    return -1;
  }
  if (a.submax_out != nullptr) {                // sampled running maximum (MK = 4)
    if constexpr (SEM == kSemF16) {
      switch (SL * 100 + R) {
#define CASE_M(sl, r) case sl * 100 + r: launch_score(sw_score_kernel<r, kSemF16, false, sl, false, false, 4>, grid, shmem, st, a); return 0;
        CASE_M(8, 13) CASE_M(8, 16) CASE_M(8, 19) CASE_M(8, 26) CASE_M(8, 32)
        CASE_M(16, 10) CASE_M(16, 12) CASE_M(16, 16) CASE_M(16, 20) CASE_M(16, 24) CASE_M(16, 32)
        CASE_M(64, 10) CASE_M(64, 12) CASE_M(64, 16) CASE_M(64, 20) CASE_M(64, 24) CASE_M(64, 32)
#undef CASE_M
      }
    }
    return -1;
  }
  if (SL == 8) {
    switch (R) {
#define CASE_H8(r) case r: launch_score(sw_score_kernel<r, SEM, false, 8>, grid, shmem, st, a); return 0;
      CASE_H8(7) CASE_H8(10) CASE_H8(13) CASE_H8(16) CASE_H8(19) CASE_H8(26) CASE_H8(32)
#undef CASE_H8
    }
    return -1;
  }
  if (SL == 64) {
    switch (R) {
#define CASE_H64(r) case r: launch_score(sw_score_kernel<r, SEM, false, 64>, grid, shmem, st, a); return 0;
      CASE_H64(10) CASE_H64(12) CASE_H64(16) CASE_H64(20) CASE_H64(24) CASE_H64(32)
#undef CASE_H64
    }
    return -1;
  }
  if (SL != 16) return -1;
  switch (R) {
#define CASE_H(r) case r: launch_score(sw_score_kernel<r, SEM, false>, grid, shmem, st, a); return 0;
    CASE_H(2) CASE_H(4) CASE_H(6) CASE_H(8) CASE_H(10) CASE_H(12) CASE_H(16) CASE_H(20) CASE_H(24) CASE_H(32)
#undef CASE_H
  }
  return -1;
}

template <int SEM>
int launch_score_R(int R, int SL, bool strips, dim3 grid, size_t shmem, hipStream_t st, const ScoreArgs &a) {
  if (a.submax_out != nullptr) {                // sampled running maximum (MK = 4): a lone long query on float32 cells
    if constexpr (SEM == kSemF32) {
      if (SL == 8 || SL == 16) {                 // batches on float32 cells (fractional scoring)
        if (strips) return -1;
        switch (SL * 100 + R) {
#define CASE_MB(sl, r) case sl * 100 + r: launch_score(sw_score_kernel<r, kSemF32, false, sl, false, false, 4>, grid, shmem, st, a); return 0;
          CASE_MB(8, 13) CASE_MB(8, 16) CASE_MB(8, 19) CASE_MB(8, 26) CASE_MB(8, 32)
          CASE_MB(16, 10) CASE_MB(16, 12) CASE_MB(16, 16) CASE_MB(16, 20) CASE_MB(16, 24) CASE_MB(16, 32)
#undef CASE_MB
        }
        return -1;
      }
      if (SL != 64) return -1;
      switch ((strips ? 100 : 0) + R) {
#define CASE_MF(r) case r: launch_score(sw_score_kernel<r, kSemF32, false, 64, false, false, 4>, grid, shmem, st, a); return 0;
#define CASE_MFS(r) case 100 + r: launch_score(sw_score_kernel<r, kSemF32, true, 64, false, false, 4>, grid, shmem, st, a); return 0;
        CASE_MF(10) CASE_MF(12) CASE_MF(16) CASE_MF(20) CASE_MF(24) CASE_MF(32) CASE_MFS(20) CASE_MFS(24) CASE_MFS(32)
#undef CASE_MF
#undef CASE_MFS
      }
    }
    return -1;
  }
  if (strips) {
    if (SL == 16) { if (R != 32) return -1; launch_score(sw_score_kernel<32, SEM, true, 16>, grid, shmem, st, a); return 0; }
    if (SL != 64) return -1;
    switch (R) {
#define CASE_S(r) case r: launch_score(sw_score_kernel<r, SEM, true, 64>, grid, shmem, st, a); return 0;
      CASE_S(20) CASE_S(24) CASE_S(32)
#undef CASE_S
    }
    return -1;
  }
  if (SL == 64) {
    switch (R) {
#define CASE_W(r) case r: launch_score(sw_score_kernel<r, SEM, false, 64>, grid, shmem, st, a); return 0;
      CASE_W(10) CASE_W(12) CASE_W(16) CASE_W(20) CASE_W(24) CASE_W(32)
#undef CASE_W
    }
    return -1;
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

// Granularity at which a tile reports its maxima = what the locate step re-runs.  uint8 engine: >= the query length
// (a power of two >= 256), so that the skewed storage order stays within two neighbouring sub-chunks.  Float engine,
// lone query (the latency path): one 64-column segment — the first sub-chunk that reaches the maximum holds the first
// maximum whatever the granularity, and a shorter sub-chunk is a shorter window to re-run.
int64_t score_sub_len(int semantics, const Bucket &b) {
  if (semantics == MI355_SW_F32 && b.count == 1 && !b.strips && b.SL != 64) return kSeg;
  int64_t s = 256;
  while (s < b.maxlen) s *= 2;
  // uint8 engine: strictly more than |x| + 64 columns, so that the first 255 in skewed order lies in the first sub-chunk that truly
  // holds one or its right neighbour (what settles a sampled query whose key sits at the cap)
  if (semantics == MI355_SW_U8SAT) while (s <= (int64_t)b.maxlen + 64) s *= 2;
  return s;
}

int64_t pick_chunk_len(int64_t max_range_len, size_t npairs, int64_t warm, int SL = 16, bool twin = false, int maxlen = 0,
                       int64_t sub_len = 0, int64_t quant = 0) {
  int64_t cl = 65536;
  while (cl < 8 * warm) cl *= 2;                 // long queries: keep the warm-up redundancy bounded
  // fill the chip: CUs x 32 waves x 4 slots (65 536 tiles on 256 CUs); shrink tiles while they stay >> warm-up
  const double cus = (double)dev_cus();
  while (cl > 2048 && cl / 2 >= 4 * warm &&
         (double)npairs * (double)((max_range_len + cl - 1) / cl) < 256.0 * cus) cl /= 2;
  // few tiles (one long query): filling the SIMDs beats the warm-up redundancy down to cl == warm
  // (measured, 10 kbp x 250 Mbp: 1.17 s at 131 k columns, 0.58 s at 32 k; profiles/r01_config5*.log)
  const double few = (SL == 64 ? 6.0 : 32.0) * cus * (twin ? 2.0 : 1.0);   // (1536 / 8192 on 256 CUs) a 64-lane tile is a wavefront of its own (two tiles with twin)
  while (cl / 2 >= std::max<int64_t>(warm, 2048) &&
         (double)npairs * (double)((max_range_len + cl - 1) / cl) < few) cl /= 2;
  // tiny problems (one read against a short reference): the call's latency is one tile's sweep and the chip is
  // mostly idle, so tiles shrink until every CU has a workgroup (down to one sub-chunk: >= 256 columns, >= |x|)
  int64_t floor_cl = 256;
  while (floor_cl < maxlen) floor_cl *= 2;
  if (sub_len > floor_cl) floor_cl = sub_len;                                          // (a tile is at least one sub-chunk: the uint8 engine's is > |x| + 64)
  else if (sub_len > 0 && sub_len < floor_cl) floor_cl = std::max<int64_t>(128, sub_len);   // finer sub-chunks allow shorter tiles (measured:
                                                                                       // 0.30 ms per 150 bp x 1 Mbp call at 128 columns, 0.33 at 256)
  const double per_wg = 256.0 / SL * (twin ? 2.0 : 1.0);
  while (cl / 2 >= floor_cl && (double)npairs * (double)((max_range_len + cl - 1) / cl) / per_wg < cus) cl /= 2;
  // Few workgroups per CU: the launch's duration is (workgroups per CU, rounded UP) x (one tile's sweep), so a tile length
  // that lets the workgroup count land just under a multiple of the 256 CUs beats the power of two next to it
  // (50 Mbp, one 400 bp read: 763 workgroups of 2048-column tiles = 3 per CU, 509 of 3072-column tiles = 2 per CU:
  // 2.54 -> 2.14 ms; config 5: 131072 -> 122880 columns).  Candidates are multiples of `quant` (the sub-chunk length).
  const bool no_quant = opt().no_quant;
  if (quant > 0 && !no_quant) {
    const int64_t tiles_per_wg = (int64_t)per_wg;
    auto rounds = [&](int64_t c) {
      const int64_t tiles = (max_range_len + c - 1) / c;
      const double wgs = (double)npairs * (double)((tiles + tiles_per_wg - 1) / tiles_per_wg);
      return std::ceil(wgs / cus);
    };
    auto cost = [&](int64_t c) { return rounds(c) * (double)(c + warm + SL); };
    const double r0 = rounds(cl);
    if (r0 <= 8.0) {
      const double need_rounds = std::min(2.0, r0);                // keep two workgroups per CU where there were two
      // switch for a predicted gain of >= 4 %; >= 7 % when it costs a workgroup per CU (fewer wavefronts to hide latency
      // behind: measured, 1000-row float32 tiles 3 -> 2 per CU: predicted -5 %, measured +3 %)
      const double base = cost(cl);
      int64_t best = cl;
      double best_gain = 0.0;
      const int64_t lo = std::max<int64_t>({quant, floor_cl, (cl / 2 + quant - 1) / quant * quant});
      for (int64_t c = lo; c <= 2 * cl; c += quant) {
        const double rc = rounds(c);
        if (rc < need_rounds) continue;
        const double gain = 1.0 - cost(c) / base;
        if (gain < (rc < r0 ? 0.07 : 0.04)) continue;
        if (gain > best_gain) { best_gain = gain; best = c; }
      }
      cl = best;
    }
  }
  { const long v = opt().chunk; if (v >= 64) cl = v / 64 * 64; }   // tuning aid
  return cl;
}

// The score tables of a call on the device (sent again only when they change), and the scale of the float32 instance.
int score_tables(mi355_sw_ctx *ctx, int maxlen, int64_t maxrange, const ScoreTable &t) {
  const void *stab_was = ctx->stab.p, *ftab_was = ctx->ftab.p;
  if (ctx->stab.ensure(t.stab.size() * 2) || ctx->ftab.ensure(t.ftab.size() * 4 + 16))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
  if (ctx->stab.p != stab_was || ctx->h_stab != t.stab) {
    ctx->h_stab = t.stab;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stab.p, ctx->h_stab.data(), ctx->h_stab.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  }
  if (!t.ftab.empty() && (ctx->ftab.p != ftab_was || ctx->h_ftab != t.ftab)) {
    ctx->h_ftab = t.ftab;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ftab.p, ctx->h_ftab.data(), ctx->h_ftab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  // float32 instance: table and gap scaled by 2^-k, 2^k above every value a cell of this call can take
  if (!t.ftab.empty()) {
    const double bound = (double)t.smaxf * (double)std::min<int64_t>(std::max(1, maxlen), std::max<int64_t>(1, maxrange)) + (double)t.smaxf + 1.0;
    ctx->fshift = std::max(1, std::min(100, std::ilogb(bound) + 2));
    std::vector<float> scaled(t.ftab.size());
    for (size_t k = 0; k < scaled.size(); ++k) scaled[k] = std::ldexp(t.ftab[k], -ctx->fshift);
    const void *was = ctx->ftab_s.p;
    if (ctx->ftab_s.ensure(scaled.size() * 4 + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
    if (ctx->ftab_s.p != was || ctx->h_ftab_s != scaled) {
      ctx->h_ftab_s.swap(scaled);
      HIPCHK(ctx, hipMemcpyAsync(ctx->ftab_s.p, ctx->h_ftab_s.data(), ctx->h_ftab_s.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  if (!t.htab8.empty()) {
    const void *was = ctx->htab8.p;
    if (ctx->htab8.ensure(t.htab8.size() * 2 + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
    if (ctx->htab8.p != was || ctx->h_htab8 != t.htab8) {
      ctx->h_htab8 = t.htab8;
      HIPCHK(ctx, hipMemcpyAsync(ctx->htab8.p, ctx->h_htab8.data(), ctx->h_htab8.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  if (!t.htab.empty()) {
    const void *htab_was = ctx->htab.p;
    if (ctx->htab.ensure(t.htab.size() * 2 + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
    if (ctx->htab.p != htab_was || ctx->h_htab != t.htab) {
      ctx->h_htab = t.htab;
      HIPCHK(ctx, hipMemcpyAsync(ctx->htab.p, ctx->h_htab.data(), ctx->h_htab.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  return 0;
}

constexpr uint32_t kFlagCap = 1u << 20;   // (query, sub-chunk) pairs a saturating float16 sweep may flag before it is abandoned
// candidates one query may have before IT is swept again exactly (sampled sweeps): 64, plus a share of 1024 (a lone query: 1088)
inline uint32_t query_flag_cap(size_t nq) { return 64u + (uint32_t)(1024 / std::max<size_t>(1, nq)); }

// Uploads what every score launch of a call shares and clears the keys.
int score_begin(mi355_sw_ctx *ctx, const QueryBatch &q, const std::vector<Range> &ranges, const ScoreTable &t) {
  const size_t nq = q.nq, nr = ranges.size();
  ctx->long_cert = -1.0f;                                            // (set by a sw_long_kernel launch with an optimistic margin)
  ctx->lsaved.valid = false;                                         // (a new score pass: set again by a sw_long_kernel launch that saves)
  if (nr > 32768) return fail(ctx, MI355_SW_ENOTSUP, "more than 32768 ranges per launch");
  // the previous call's copies out of these host vectors have completed: every call ends synchronised
  std::vector<int64_t> &rl = ctx->h_ranges;
  rl.resize(2 * nr);
  int64_t maxrange = 1;
  for (size_t k = 0; k < nr; ++k) { rl[k] = ranges[k].lo; rl[nr + k] = ranges[k].hi; maxrange = std::max(maxrange, ranges[k].hi - ranges[k].lo); }
  if (ctx->ranges.ensure(rl.size() * 8) || ctx->keys.ensure(nq * nr * 8))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
  HIPCHK(ctx, hipMemcpyAsync(ctx->ranges.p, rl.data(), rl.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  int rc = score_tables(ctx, q.maxlen, maxrange, t);
  if (rc) return rc;
  HIPCHK(ctx, hipMemsetAsync(ctx->keys.p, 0, nq * nr * 8, ctx->stream));
  // room for the flagged (query, sub-chunk) pairs: a few per query (sampled / saturating sweeps), at least kFlagCap
  // (the filter of the sampled sweeps appends at most query_flag_cap + 1 entries per query: its list cannot overflow)
  ctx->flag_cap = (uint32_t)std::min<size_t>(0x7FFFFFFFu, std::max<size_t>(kFlagCap, (size_t)(query_flag_cap(q.nq) + 2) * q.nq + 4096));
  ctx->first_valid = false;
  if (ctx->flags.ensure(8 + (size_t)ctx->flag_cap * 8) || ctx->qcnt.ensure(q.nq * 4 + 16) ||
      ctx->first.ensure(q.nq * (size_t)(kFirstCandidates + 1) * 4 + 16))
    return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
  HIPCHK(ctx, hipMemsetAsync(ctx->flags.p, 0, 8, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->qcnt.p, 0, q.nq * 4, ctx->stream));
  return 0;
}

// Where a score launch finds its ranges and leaves its keys when not in the context's own buffers (host_solo.h).
struct ScoreIO {
  const int64_t *range_lo = nullptr, *range_hi = nullptr;
  unsigned long long *keys = nullptr;
};

// A lone long query on sw_long_kernel: one workgroup per tile (or `pipes` tiles), every strip of the tile on its own
// wavefront.  Tile length: the ranges' columns dealt to as many workgroups as the chip holds at once (LDS decides how many
// per CU), rounded up to whole sub-chunks — never more workgroups than that, a second round would double the launch.
int long_score_launch(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<Range> &ranges,
                      const mi355_sw_params &p, const ScoreTable &t, Bucket &b) {
  HostTrace trace_("long_score_launch");
  const size_t nr = ranges.size();
  const int qid = q.order[b.first];
  double total_cols = 0;
  int64_t maxlen = 0;
  for (auto &r : ranges) { total_cols += (double)(r.hi - r.lo); maxlen = std::max(maxlen, r.hi - r.lo); }
  // float32 profile where the strips can be dealt to at most eight workgroups per tile (the cell's add then issues at the
  // double rate, sw_long_kernel.h); else the float16 profile of the whole query in one workgroup
  bool p32 = !opt().no_long_p32;
  int groups = 1, spg = b.nstrips;
  if (p32) {
    p32 = false;
    for (int g = (opt().long_groups > 0 && !tl_no_wait) ? (int)opt().long_groups : 1; g <= (tl_no_wait ? 1 : 8); g *= 2) {
      const int s1 = (b.nstrips + g - 1) / g;
      if (s1 <= long_max_waves(b.R) && long_lds_bytes(ref.ncodes, s1, 1, b.R, kLongSubsMax, true) <= long_lds_max()) { p32 = true; groups = g; spg = s1; break; }
    }
  }
  if (!p32 && (b.nstrips > long_max_waves(b.R) || long_lds_bytes(ref.ncodes, b.nstrips, 1, b.R, kLongSubsMax) > long_lds_max()))
    return fail(ctx, MI355_SW_ENOTSUP, "internal: no sw_long_kernel layout for this query");
  groups = (b.nstrips + spg - 1) / spg;                                 // (no workgroup without strips)
  int pipes = (int)opt().long_pipes;
  if (pipes < 1) pipes = long_max_waves(b.R) / spg;                     // as many wavefronts per CU as fit: the sweep
                                                                        // is bound by issue slots that only other wavefronts fill
  pipes = std::max(1, std::min(pipes, long_max_waves(b.R) / spg));
  while (pipes > 1 && long_lds_bytes(ref.ncodes, spg, pipes, b.R, kLongSubsMax, p32) > long_lds_max()) --pipes;
  // columns per reported (and saved) sub-chunk.  The finish recomputes blocks of sub_len columns around the winner (host_saved.h):
  // a per-call constant of ~2.3 ms at 2048, ~1.6 at 1024, while the sweep pays 1.4 ms per 250 M columns for the finer grain
  // (measured, tools/c5_whole.py long_sub=...): finer below ~125 M columns — a rank's share of a sharded reference
  int64_t sub_len = opt().long_sub >= 64 ? opt().long_sub / 64 * 64 : (total_cols < 125e6 ? 1024 : 2048);
  // uint8 engine: a power of two >= |x|, so that the skewed storage order stays within two neighbouring sub-chunks (locate_fast)
  if (p.semantics == MI355_SW_U8SAT) sub_len = std::max<int64_t>(sub_len, score_sub_len(p.semantics, b));
  int64_t chunk = 0;
  int wg_per_cu = 1;
  for (;;) {
    const size_t lds = long_lds_bytes(ref.ncodes, spg, pipes, b.R, kLongSubsMax, p32);
    wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(dev_lds() / lds, (size_t)(long_max_waves(b.R) / (spg * pipes))));
    // never more workgroups than the chip holds at once: with several workgroups per tile they WAIT for each other
    const int64_t G = std::max<int64_t>(groups, opt().long_wgs > 0 ? opt().long_wgs : (int64_t)dev_cus() * wg_per_cu);
    chunk = std::max<int64_t>(sub_len, (int64_t)std::ceil(total_cols / (double)((G / groups) * pipes) / (double)sub_len) * sub_len);
    auto wgs = [&](int64_t c) { int64_t n = 0; for (auto &r : ranges) n += ((((r.hi - r.lo) + c - 1) / c + pipes - 1) / pipes) * groups; return n; };
    while (wgs(chunk) > G) chunk += sub_len;
    if (chunk / sub_len <= kLongSubsMax) break;
    sub_len *= 2;
  }
  if (groups == 1) { const long v = opt().chunk; if (v >= sub_len) chunk = v / sub_len * sub_len; }   // tuning aid (one workgroup per tile only:
                                                                                                     // waiting workgroups must all be resident)
  b.chunk_len = chunk;
  b.sub_len = sub_len;
  const int64_t cpr = (maxlen + chunk - 1) / chunk;
  const int subs_per_tile = (int)(chunk / sub_len);
  if (b.sampled && (nr > 4096 || (double)nr * (double)(cpr * subs_per_tile) > 4.0e9)) b.sampled = false;

  LongArgs a;
  a.refcodes = ref.codes.as<uint8_t>();
  a.range_lo = ctx->ranges.as<int64_t>();
  a.range_hi = ctx->ranges.as<int64_t>() + nr;
  a.chunk_len = chunk; a.sub_len = sub_len;
  a.warm = (cpr == 1) ? 0 : b.warm;              // a single tile per range starts at the range's own border
  // Optimistic warm-up margin (DESIGN.md §3.3 lemma L11, from L3).  The full margin m + ceil(smax m / g) makes EVERY cell exact; a cell of value v is already
  // exact behind m + ceil((smax m - v) / g) + 2 columns (a path that reaches v within m diagonal steps affords that many gap
  // columns).  The sweep therefore starts with the margin that is enough for maxima above 11/12 of the best possible score —
  // what a read with a real hit has — and reports the value above which it was exact (long_cert); the callers check their
  // maximum against it and sweep again with the margin that maximum needs when it falls short (host_pipeline.h).
  ctx->long_cert = -1.0f;
  if (b.opt_margin && a.warm > 0 && t.integral) {
    const int64_t m = b.maxlen;
    int64_t w1 = m + std::max<int64_t>(256, m / 8);
    w1 = std::max(w1, ctx->long_margin);
    w1 = (w1 + 63) / 64 * 64;
    if (w1 < a.warm) {
      a.warm = w1;
      ctx->long_cert = (float)((double)t.smax * (double)m - (double)t.gap * (double)(w1 - m - 2));
    }
  }
  a.qbytes = q.bytes.as<uint8_t>() + q.off[qid];
  a.qlen = q.len[qid]; a.qid = qid; a.nq = (int)q.nq;
  a.htab = ctx->htab.as<uint16_t>();
  a.ftab = ctx->ftab_s.as<float>();
  a.ncodes = ref.ncodes;
  a.gap_s = std::ldexp(t.gapf, -ctx->fshift);
  a.scale = std::ldexp(kF16Scale, -ctx->fshift);
  a.pubmax = 0u;
  if (b.unsat) { const float v = std::ldexp(255.0f, -ctx->fshift); memcpy(&a.pubmax, &v, 4); }
  a.keys = ctx->keys.as<unsigned long long>();
  a.submax_out = nullptr;
  a.nstrips = b.nstrips; a.groups = groups; a.spg = spg; a.pipes = pipes; a.subs_per_tile = subs_per_tile;
  const int64_t tgroups = (cpr + pipes - 1) / pipes;                    // workgroup sets per range
  a.tiles_stride = tgroups * pipes;
  a.gbound = nullptr; a.gstride = 0; a.gcount = nullptr;
  if (groups > 1) {
    a.gstride = a.warm + chunk + 192;
    const size_t slots = nr * (size_t)a.tiles_stride * (size_t)(groups - 1);
    if (ctx->brow.ensure(slots * (size_t)a.gstride * 4) || ctx->gcnt.ensure(slots * 8 + 64))
      return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(boundary rows between the workgroups of a tile) failed");
    HIPCHK(ctx, hipMemsetAsync(ctx->gcnt.p, 0, slots * 8, ctx->stream));
    a.gbound = ctx->brow.as<float>();
    a.gcount = ctx->gcnt.as<long long>();
  }
  a.status = reinterpret_cast<int32_t *>(ctx->flags.as<unsigned int>() + 1);      // zeroed by score_begin, read by score_fetch
  a.fault = (opt().fault_inject == 2 && groups > 1) ? 1 : 0;                       // test hook: the waits between workgroups expire at once
  const int64_t nsub = cpr * subs_per_tile;                           // sub-chunks of one range (value rows: one per range)
  a.submax_range_stride = nsub;
  if (b.sampled) {
    if (ctx->submax.ensure((size_t)nr * (size_t)nsub * 4 + 64)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(sub-chunk values) failed");
    a.submax_out = ctx->submax.as<uint32_t>();
    // (the workgroups of a tile merge their values by atomic maximum; tiles beyond a short range's end never run)
    HIPCHK(ctx, hipMemsetAsync(ctx->submax.p, 0, (size_t)nr * (size_t)nsub * 4, ctx->stream));
  }
  // what the finish starts from instead of zero borders and warm-up margins (sw_long_kernel.h colsave / rowsave, host_saved.h):
  // 4 B per row and sub-chunk + 4 B per strip boundary and column — 12 GB for 10 kbp x 250 Mbp, written once while the sweep
  // runs (60 GB/s).  Not for the uint8 engine's unsaturated sweep (its finish evaluates another recurrence).
  ctx->lsaved.valid = false;
  a.colsave = nullptr; a.rowsave = nullptr; a.col_subs = nsub; a.col_rows = (int64_t)b.nstrips * 64 * b.R; a.row_stride = a.warm + chunk + 192;
  if (!opt().no_long_save && !b.unsat && t.integral && p.lut == nullptr) {
    const size_t col_bytes = (size_t)nr * (size_t)nsub * (size_t)a.col_rows * 4;
    const size_t row_bytes = (size_t)nr * (size_t)a.tiles_stride * (size_t)std::max(1, b.nstrips - 1) * (size_t)a.row_stride * 4;
    if (col_bytes + row_bytes <= ((size_t)64 << 30) && !ctx->colsave.ensure(col_bytes + 64) && !ctx->rowsave.ensure(row_bytes + 64)) {
      a.colsave = ctx->colsave.as<float>();
      a.rowsave = b.nstrips > 1 ? ctx->rowsave.as<float>() : nullptr;
      const long what = opt().long_save_what;                           // tuning aid (timing only: the finish needs both): 1 columns, 2 rows
      if (what == 1) a.rowsave = nullptr;
      if (what == 2) a.colsave = nullptr;
      LongSaved &ls = ctx->lsaved;
      ls.ref = &ref; ls.batch = &q; ls.ref_version = ref.version; ls.batch_version = q.version; ls.params = p; ls.qid = qid;
      ls.ranges = ranges; ls.chunk = chunk; ls.sub_len = sub_len; ls.warm = a.warm; ls.nstrips = b.nstrips; ls.R = b.R; ls.spt = subs_per_tile;
      ls.tiles_stride = a.tiles_stride; ls.col_subs = nsub; ls.col_rows = a.col_rows; ls.row_stride = a.row_stride; ls.fshift = ctx->fshift;
      ls.valid = what != 1 && what != 2;                                // (the launch below fills the buffers; every call ends synchronised)
    }
  }
  const size_t shmem = long_lds_bytes(ref.ncodes, spg, pipes, b.R, subs_per_tile, p32);
  const dim3 grid((unsigned)(tgroups * groups), (unsigned)nr);
  const dim3 block((unsigned)(64 * spg * pipes));
  if (ctx->score_ev.size() < ctx->score_ev_used + 2) {
    for (int e = 0; e < 2; ++e) { hipEvent_t ev; HIPCHK(ctx, hipEventCreate(&ev)); ctx->score_ev.push_back(ev); }
  }
  HIPCHK(ctx, hipEventRecord(ctx->score_ev[ctx->score_ev_used], ctx->stream));
  auto launch = [&](auto kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kernel, grid, block, shmem, ctx->stream, a);
  };
#define LONG_CASE(r) \
  if (b.R == r) { \
    if (p32) { if (b.sampled) launch(sw_long_kernel<r, kLongMK, true>); else launch(sw_long_kernel<r, 1, true>); } \
    else { if (b.sampled) launch(sw_long_kernel<r, kLongMK, false>); else launch(sw_long_kernel<r, 1, false>); } \
  } else
  LONG_CASE(20) LONG_CASE(24) LONG_CASE(32)
    return fail(ctx, MI355_SW_ENOTSUP, "no sw_long_kernel instance for this R");
#undef LONG_CASE
  HIPCHK(ctx, hipGetLastError());
  path_note(ctx, "long[R=%d,groups=%d,pipes=%d,p32=%d,sampled=%d,opt_margin=%d,saved=%d,unsat=%d]", b.R, groups, pipes, (int)p32, (int)b.sampled,
            (int)(ctx->long_cert >= 0.0f), (int)(a.colsave != nullptr), (int)b.unsat);
  ctx->long_launched = true;
  if (b.sampled) {
    // one filter launch per range: its value row against ITS key; entries carry the range in the sub-chunk index (r * nsub + s)
    const dim3 fgrid((unsigned)std::min<int64_t>(64, (nsub + 255) / 256), 1u);
    for (size_t r = 0; r < nr; ++r)
      hipLaunchKernelGGL(sw_sample_filter<true>, fgrid, dim3(256), 0, ctx->stream, (const void *)(a.submax_out + r * (size_t)nsub), nsub, nsub,
                         q.sel.as<int32_t>(), b.first, 1, (const unsigned long long *)(a.keys + r * (size_t)q.nq),
                         std::ldexp((float)(kLongMK - 1) * t.gapf, -ctx->fshift),
                         ctx->flags.as<unsigned int>(), reinterpret_cast<uint2 *>(ctx->flags.as<unsigned int>() + 2), ctx->flag_cap,
                         ctx->qcnt.as<unsigned int>(), query_flag_cap(q.nq), (uint32_t)(r * (size_t)nsub));
    HIPCHK(ctx, hipGetLastError());
    ctx->long_nsub = nsub;
  }
  HIPCHK(ctx, hipEventRecord(ctx->score_ev[ctx->score_ev_used + 1], ctx->stream));
  ctx->score_ev_used += 2;
  ctx->timings[4] += 1;
  double cells = 0;
  for (auto &r : ranges) cells += (double)q.len[qid] * (double)(r.hi - r.lo);
  ctx->timings[5] += cells;
  if (cells > ctx->last_kernel.cells) {
    mi355_sw_kernel_info &ki = ctx->last_kernel;
    ki.cell = b.sem; ki.lanes = 64; ki.rows_per_lane = b.R; ki.strips = b.nstrips; ki.twin = 0;
    ki.chunk_len = chunk; ki.sub_len = sub_len; ki.warm = a.warm; ki.cells = cells;
    ki.valu_ops_per_cell = valu_ops_per_cell(b);
    std::snprintf(ki.name, sizeof ki.name, "sw_long_kernel<R=%d, f32 cells, %s profile; %d strips pipelined over %d workgroup(s), %d tile(s) per workgroup>%s%s",
                  b.R, p32 ? "f32" : "f16", b.nstrips, groups, pipes, b.unsat ? " uint8 engine swept unsaturated, maxima clamped at 255" : "",
                  b.sampled ? "; maximum folded every 8th step (candidates re-evaluated)" : "");
  }
  return 0;
}

// One score-kernel launch: bucket b over all ranges.  Device time is added to ctx->timings[0].
int score_launch(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<Range> &ranges,
                 const mi355_sw_params &p, const ScoreTable &t, Bucket &b, const ScoreIO *io = nullptr) {
  if (b.longp && io == nullptr) return long_score_launch(ctx, ref, q, ranges, p, t, b);
  HostTrace trace_("score_launch");
  const size_t nr = ranges.size();
  int64_t maxlen = 0;
  for (auto &r : ranges) maxlen = std::max(maxlen, r.hi - r.lo);
  const size_t npairs = (sem_is_float(b.sem) || b.twin) ? (size_t)b.count : ((size_t)b.count + 1) / 2;   // queries per workgroup: 1 or 2
  // report maxima per sub-chunk of >= 256 columns (>= query length, so that the uint8 storage order stays
  // within two neighbouring sub-chunks): that is what locate re-runs; the strip-mined instance reports per tile
  b.sub_len = score_sub_len(p.semantics, b);
  // float engine, strip-mined: the sub-chunk is any multiple of 64 columns (1/64 of the tile: the strip-mined instances keep
  // <= 64 sub-chunk maxima in LDS); the uint8 engine's stays a power of two >= |x| (storage order, see above)
  const bool free_sub = p.semantics == MI355_SW_F32 && b.strips;
  if (b.strips && !free_sub) {
    // the candidates of the refinement below reach twice the unrefined tile length: settle the sub-chunk for that first
    const int64_t cl0 = pick_chunk_len(maxlen, npairs * nr, b.warm, b.SL, b.twin, b.maxlen, b.sub_len, 0);
    while (2 * cl0 / b.sub_len > 64) b.sub_len *= 2;
  }
  b.chunk_len = pick_chunk_len(maxlen, npairs * nr, b.warm, b.SL, b.twin, b.maxlen, b.sub_len, free_sub ? 64 * kSeg : b.sub_len);
  if (free_sub && b.chunk_len % (64 * kSeg) == 0) b.sub_len = b.chunk_len / 64;
  if (b.strips) while (b.chunk_len / b.sub_len > 64) b.sub_len *= 2;      // the strip-mined instances keep <= 64 sub-chunk maxima in LDS
  if (b.sub_len > b.chunk_len || b.chunk_len % b.sub_len != 0) b.sub_len = b.chunk_len;
  const int64_t cpr = (maxlen + b.chunk_len - 1) / b.chunk_len;
  const int nslot = 256 / b.SL;                                     // tiles (twin: tile pairs) per workgroup
  const int64_t cgroups = ((b.twin ? (cpr + 1) / 2 : cpr) + nslot - 1) / nslot;
  if ((double)npairs * (double)cgroups > 2.0e9) return fail(ctx, MI355_SW_ENOTSUP, "grid too large");

  ScoreArgs a;
  a.refcodes = ref.codes.as<uint8_t>();
  a.ref_len = (int64_t)ref.n;
  a.range_lo = io ? io->range_lo : ctx->ranges.as<int64_t>();
  a.range_hi = io ? io->range_hi : ctx->ranges.as<int64_t>() + nr;
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
  a.stab = b.sem == kSemF32 ? ctx->ftab_s.p
           : (sem_is_float(b.sem) ? ctx->ftab.p : (b.sem == kSemF16 ? ctx->htab.p : (b.sem == kSemU8H ? ctx->htab8.p : ctx->stab.p)));
  a.ncodes = ref.ncodes;
  if (b.sem == kSemF32) { const float gs = std::ldexp(t.gapf, -ctx->fshift); memcpy(&a.gap2, &gs, 4); }
  else if (sem_is_float(b.sem)) memcpy(&a.gap2, &t.gapf, 4);
  else if (b.sem == kSemF16) a.gap2 = (uint32_t)half_bits(-(float)t.gap / kF16Scale) * 0x00010001u;
  else if (b.sem == kSemU8H) a.gap2 = (uint32_t)half_bits(-(float)t.gap / 256.0f) * 0x00010001u;
  else a.gap2 = (uint32_t)t.gap * 0x00010001u;
  a.clamp2 = 255u * 0x00010001u;
  a.pubmax = 0u;
  if (b.unsat) {                                  // uint8 engine swept without saturation: clamp what is published
    if (b.sem == kSemF32) { const float v = std::ldexp(255.0f, -ctx->fshift); memcpy(&a.pubmax, &v, 4); }
    else a.pubmax = (uint32_t)half_bits(255.0f / kF16Scale);
  }
  a.keys = io ? io->keys : ctx->keys.as<unsigned long long>();
  a.flag_count = nullptr; a.flag_list = nullptr; a.flag_cap = 0; a.flag_value = 0;
  a.submax_out = nullptr; a.submax_stride = 0;
  if (b.sampled && nr != 1) b.sampled = false;                        // (per-range value rows are not laid out)
  if (b.satflag) {
    a.flag_count = ctx->flags.as<unsigned int>();
    a.flag_list = reinterpret_cast<uint2 *>(ctx->flags.as<unsigned int>() + 2);
    a.flag_cap = ctx->flag_cap;
    a.flag_value = (uint32_t)half_bits(1.0f);                       // cells hold H / 2048: the clamp's upper end
  }

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
  size_t shmem = profile_lds_bytes(ref.ncodes, b.R, b.SL, b.twin, b.comb) + (size_t)(b.twin ? 2 : 1) * nslot * codebuf_bytes(b.SL);
  const int64_t nsub = cpr * (b.chunk_len / b.sub_len);                // sub-chunks of the range (sampled sweep: one value each)
  if (b.sampled) {
    if (ctx->submax.ensure((size_t)pn * (size_t)nqw * (size_t)nsub * (b.sem == kSemF32 ? 4 : 2) + 64))
      return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(sub-chunk values) failed");
    a.submax_out = ctx->submax.as<uint16_t>();
    a.submax_stride = nsub;
    a.flag_count = nullptr; a.flag_list = nullptr;                  // (the filter below appends, not the sweep)
  }
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
    // boundary rows start as H = 0 (a non-zero bit pattern in the scaled float16 instance)
    if (b.sem == kSemU8H) HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->brow.p, (int)kU8HZero, bytes / 4, ctx->stream));
    else HIPCHK(ctx, hipMemsetAsync(ctx->brow.p, 0, bytes, ctx->stream));
    a.brow = ctx->brow.as<uint32_t>();
    shmem += (size_t)2 * nslot * kSeg * 4 + (size_t)nslot * 64 * 4;   // boundary windows + per-sub-chunk maxima
  }
  if (ctx->score_ev.size() < ctx->score_ev_used + 2) {
    for (int e = 0; e < 2; ++e) { hipEvent_t ev; HIPCHK(ctx, hipEventCreate(&ev)); ctx->score_ev.push_back(ev); }
  }
  HIPCHK(ctx, hipEventRecord(ctx->score_ev[ctx->score_ev_used], ctx->stream));
  int rc = b.twin ? (b.sem == kSemF16 ? (b.SL == 64 ? launch_score_twin<kSemF16>(b.R, b.strips, grid, shmem, ctx->stream, a)
                                                    : launch_score_twin16(b.R, b.comb, grid, shmem, ctx->stream, a))
                     : b.sem == kSemU8H ? launch_score_twin<kSemU8H>(b.R, b.strips, grid, shmem, ctx->stream, a)
                     : b.sem == kSemU8 ? launch_score_twin<kSemU8>(b.R, b.strips, grid, shmem, ctx->stream, a)
                                       : launch_score_twin<kSemI16>(b.R, b.strips, grid, shmem, ctx->stream, a))
           : b.sem == kSemF16 ? launch_score_f16<kSemF16>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
           : b.sem == kSemU8H ? launch_score_R<kSemU8H>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
           : b.sem == kSemU8 ? launch_score_R<kSemU8>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
           : b.sem == kSemF32U8 ? launch_score_R<kSemF32U8>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
           : b.sem == kSemF32 ? launch_score_R<kSemF32>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a)
                              : launch_score_R<kSemI16>(b.R, b.SL, b.strips, grid, shmem, ctx->stream, a);
  if (rc) return fail(ctx, MI355_SW_ENOTSUP, "no score kernel instance for this R");
  HIPCHK(ctx, hipGetLastError());
  {
    static const char *cellname[] = {"i16", "u8i16", "f32", "u8f32", "f16", "u8f16"};
    path_note(ctx, "score[cell=%s,SL=%d,R=%d,strips=%d,twin=%d,comb=%d,sampled=%d,satflag=%d,unsat=%d,pow2=%d]", cellname[b.sem], b.SL, b.R, (int)b.strips,
              (int)b.twin, (int)b.comb, (int)b.sampled, (int)b.satflag, (int)b.unsat, (int)((b.chunk_len & (b.chunk_len - 1)) == 0));
  }
  if (b.sampled) {
    // grid.y = query positions of this launch, at most 65535 per filter launch
    for (int f0 = 0; f0 < a.qcount; f0 += 65535) {
      const int fc = std::min(65535, a.qcount - f0);
      const dim3 fgrid((unsigned)std::min<int64_t>(64, (nsub + 255) / 256), (unsigned)fc);
      const void *rows = reinterpret_cast<const uint8_t *>(a.submax_out) + (size_t)f0 * (size_t)nsub * (b.sem == kSemF32 ? 4 : 2);
      if (b.sem == kSemF32)
        hipLaunchKernelGGL(sw_sample_filter<true>, fgrid, dim3(256), 0, ctx->stream, rows, nsub, nsub,
                           (const int32_t *)a.qsel, a.qfirst + f0, fc, (const unsigned long long *)a.keys,
                           std::ldexp(3.0f * t.gapf + (t.integral ? 0.0f : std::ldexp(t.smaxf * (float)(b.maxlen + 1), -20)), -ctx->fshift),
                           ctx->flags.as<unsigned int>(), reinterpret_cast<uint2 *>(ctx->flags.as<unsigned int>() + 2), ctx->flag_cap,
                           ctx->qcnt.as<unsigned int>(), query_flag_cap(q.nq));
      else
        hipLaunchKernelGGL(sw_sample_filter<false>, fgrid, dim3(256), 0, ctx->stream, rows, nsub, nsub,
                           (const int32_t *)a.qsel, a.qfirst + f0, fc, (const unsigned long long *)a.keys, 3.0f * (float)t.gap,
                           ctx->flags.as<unsigned int>(), reinterpret_cast<uint2 *>(ctx->flags.as<unsigned int>() + 2), ctx->flag_cap,
                           ctx->qcnt.as<unsigned int>(), query_flag_cap(q.nq));
      // uint8 engine: the first candidates, in order, of the queries over their cap (settled without a second sweep when the
      // key sits at 255, align_range_core)
      if (b.unsat && b.sem != kSemF32 && nr == 1 && !opt().no_first) {
        hipLaunchKernelGGL(sw_sample_first<false>, dim3((unsigned)fc), dim3(64), 0, ctx->stream, rows, nsub, nsub,
                           (const int32_t *)a.qsel, a.qfirst + f0, fc, (const unsigned long long *)a.keys, 3.0f * (float)t.gap,
                           (const unsigned int *)ctx->qcnt.as<unsigned int>(), query_flag_cap(q.nq), 0u, ctx->first.as<uint32_t>());
        ctx->first_valid = true;
        path_note(ctx, "sample_first");
      }
      HIPCHK(ctx, hipGetLastError());
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->score_ev[ctx->score_ev_used + 1], ctx->stream));
  ctx->score_ev_used += 2;                                // read by score_fetch, after the launches have drained
  ctx->timings[4] += 1;
  }
  double cells = 0;
  for (int k = 0; k < b.count; ++k)
    for (auto &r : ranges) cells += (double)q.len[q.order[b.first + k]] * (double)(r.hi - r.lo);
  ctx->timings[5] += cells;
  if (cells > ctx->last_kernel.cells) {
    mi355_sw_kernel_info &ki = ctx->last_kernel;
    ki.cell = b.sem; ki.lanes = b.SL; ki.rows_per_lane = b.R; ki.strips = b.strips; ki.twin = b.twin;
    ki.chunk_len = b.chunk_len; ki.sub_len = b.sub_len; ki.warm = a.warm; ki.cells = cells;
    ki.valu_ops_per_cell = valu_ops_per_cell(b);
    static const char *cellname[] = {"i16x2", "u8 as i16x2", "f32", "u8 as f32", "f16x2", "u8 as f16x2"};
    std::snprintf(ki.name, sizeof ki.name, "sw_score_kernel<R=%d, %s, SL=%d%s%s>%s", b.R, cellname[b.sem], b.SL,
                  b.strips ? ", strips" : "", b.twin ? (b.comb ? ", twin, code-pair profile" : ", twin") : "",
                  b.unsat ? " uint8 engine swept unsaturated, maxima clamped at 255"
                  : b.satflag ? (b.sampled ? " float engine swept saturating at 2048" : " float engine swept saturating at 2048, saturated sub-chunks re-evaluated exactly") : "");
    if (b.sampled) {
      const size_t at = std::strlen(ki.name);
      std::snprintf(ki.name + at, sizeof ki.name - at, "; maximum folded every 4th step (candidates re-evaluated)");
    }
  }
  return 0;
}

constexpr int kRetryNoWait = 1;   // score_fetch: not an error — sweep again, tl_no_wait is set

int score_fetch(mi355_sw_ctx *ctx, size_t count, std::vector<unsigned long long> &keys) {
  keys.resize(count);
  HIPCHK(ctx, hipMemcpyAsync(keys.data(), ctx->keys.p, count * 8, hipMemcpyDeviceToHost, ctx->stream));
  int32_t long_status = 0;
  if (ctx->long_launched) HIPCHK(ctx, hipMemcpyAsync(&long_status, ctx->flags.as<unsigned int>() + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->long_launched = false;
  if (long_status != 0) {
    // the workgroups of a tile wait for each other (several per tile): on a device that does not hold them all at once the wait
    // expires — the caller sweeps again on a layout whose waits stay inside one workgroup (once per call)
    ctx->score_ev_used = 0;
    if (!tl_no_wait) { tl_no_wait = true; ctx->wait_retries += 1; return kRetryNoWait; }
    return fail(ctx, MI355_SW_ENODEV, "sw_long_kernel: a pipeline wait expired (a wavefront of the workgroup made no progress)");
  }
  for (size_t e = 0; e + 1 < ctx->score_ev_used; e += 2) {     // device time of the score launches
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->score_ev[e], ctx->score_ev[e + 1]));
    ctx->timings[0] += (double)ms * 1000.0;
  }
  ctx->score_ev_used = 0;
  return 0;
}

}  // namespace
