// host_pipeline.h — the per-call pipeline: locate, traceback routing, whole-matrix path, align_range, range_maxima
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
namespace {

void set_result(mi355_sw_result &r, float score, int64_t ix, int64_t iy, const TraceOut *t) {
  r.score = score;
  r.end_x = score > 0 ? ix : 0;
  r.end_y = score > 0 ? iy : 0;
  r.pos = t ? t->pos : 0;
  const size_t len = t ? t->len : 0;
  r.cons_len = len;
  // both strings live in ONE allocation owned through cons_x (mi355_sw_free_result frees only that)
  r.cons_x = (char *)malloc(2 * len + 2);
  r.cons_y = r.cons_x + len + 1;
  if (len) { memcpy(r.cons_x, t->cx, len); memcpy(r.cons_y, t->cy, len); }
  r.cons_x[len] = 0;
  r.cons_y[len] = 0;
}

// Traceback for located alignments of one range: windows left of the argmax, grown on demand.
int trace_located(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                  const mi355_sw_params &p, const std::vector<int64_t> &qwarm, const ScoreTable &table,
                  const std::vector<int> &qidx, const std::vector<Located> &loc, std::vector<TraceOut> &tout) {
  tout.assign(qidx.size(), TraceOut());
  // A cell in row i is exact once it lies i + ceil(i * smax / g) columns into a window (lemma L2: the bound
  // for a path that can only use rows 1..i), so the window needs that margin at the argmax row plus room
  // for the horizontal excursions of the walk; the walk kernel checks every cell it visits against the bound.
  const Margin mg = table.margin(q.maxlen);
  const float slope = (float)mg.slope();
  auto row_need = [&](int64_t i) { return mg.finite() ? clamp_cols((double)i + std::ceil((double)i * (double)slope) + 2.0) : kColsMax; };
  std::vector<size_t> todo;
  // short reads with identity scoring: decisions by the register-wavefront kernel (lanes = rows of x);
  // long queries (identity scoring, or any table in the float engine): the pipelined strip kernel
  const bool wave_ok = wave_scoring_ok(p) && !opt().no_wave;
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
    // a lone long query whose sweep saved its columns / strip rows (host_saved.h): decisions from stored exact values, one
    // block per (strip, sub-chunk) the walk can reach; what the walk's checks refuse takes the zero-border windows below
    if (pass == 1 && q.nq == 1 && sub.size() == 1) {
      const int sr = saved_range_index(ctx, ref, q, rg, p, sub[0]);
      if (sr >= 0) {
        TraceOut t1;
        const int rc1 = trace_from_saved(ctx, ref, q, rg, p, sr, sub[0], sl[0], table, t1);
        if (rc1 < 0) return rc1;
        if (rc1 == 0) { tout[owner[0]] = t1; continue; }
        ctx->saved_fallbacks += 1;
      }
    }
    if (sub.empty()) continue;
    std::vector<TraceOut> t2;
    const bool latency_mode = pass == 0 && sub.size() <= kLatencyJobs && strip_ok;
    int rc = wave_trace(ctx, ref, q, rg, p, 0, sub, sl, t2, pass == 1 || latency_mode, &table);
    if (rc) return rc;
    for (size_t t = 0; t < sub.size(); ++t) tout[owner[t]] = std::move(t2[t]);
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
        if (st[t] == 0) tout[k] = std::move(outs[t]);
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
        tout[owner[t]] = std::move(outs[t]);
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
  const bool wave_ok = wave_scoring_ok(p) && !opt().no_wave;
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
    for (size_t t = 0; t < ks.size(); ++t) { loc[ks[t]] = l2[t]; tout[ks[t]] = std::move(t2[t]); }
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
      jobs.reserve(qidx.size()); owner.reserve(qidx.size());
      for (size_t k = 0; k < qidx.size(); ++k) {
        if (orient[k] != o) continue;
        WaveJob j;
        j.q = qidx[k]; j.orient = o; j.s_lo = 0; j.nb = o == 0 ? (int32_t)n : q.len[qidx[k]]; j.track = true; j.dirs = false;
        jobs.push_back(j); owner.push_back(k);
      }
      rc = run_wave(ctx, ref, q, rg, p, jobs);
      if (rc) return rc;
      parallel_for(jobs.size(), [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t) {
          Located &L = loc[owner[t]];
          L.score = jobs[t].best > 0 ? jobs[t].best : 0;
          L.ix = jobs[t].ci; L.iy = jobs[t].cj;
        }
      });
    }
  }
  // 2. traceback of the wave-eligible ones
  if (want_trace) {
    for (int o = 0; o < 2; ++o) {
      std::vector<int> sub;
      std::vector<Located> sl;
      std::vector<size_t> owner;
      sub.reserve(qidx.size()); sl.reserve(qidx.size()); owner.reserve(qidx.size());
      for (size_t k = 0; k < qidx.size(); ++k)
        if (orient[k] == o) { sub.push_back(qidx[k]); sl.push_back(loc[k]); owner.push_back(k); }
      if (sub.empty()) continue;
      std::vector<TraceOut> t2;
      rc = wave_trace(ctx, ref, q, rg, p, o, sub, sl, t2);
      if (rc) return rc;
      for (size_t t = 0; t < sub.size(); ++t) tout[owner[t]] = std::move(t2[t]);
    }
  }
  return 0;
}

// Argmax cells for the score-kernel queries (qfast[q] != 0) over one range, from the score pass' keys.
// qchunk / qwarm: tile geometry of each query's bucket.
int locate_fast(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                const mi355_sw_params &p, const std::vector<char> &qfast_in, const std::vector<int64_t> &qchunk,
                const std::vector<int64_t> &qwarm, const std::vector<char> &qfloat, const unsigned long long *keys,
                const ScoreTable &table, std::vector<Located> &loc, const std::vector<char> *skip = nullptr) {
  HostTrace trace_("locate_fast");
  const size_t nq = q.nq;
  std::vector<char> qfast = qfast_in;                    // (queries already located elsewhere are not ours)
  if (skip) for (size_t k = 0; k < nq; ++k) if ((*skip)[k]) qfast[k] = 0;
  const int64_t n = rg.hi - rg.lo;
  std::vector<ExactJob> jobs;
  std::vector<WaveJob> sjobs;                  // long queries with identity scoring: pipelined strip kernel
  std::vector<WaveJob> wjobs;                  // short queries, float engine, identity scoring: register wavefront
  const bool strip_ok = strip_scoring_ok(ref, p);
  // float order = (column, row): no cell left of the sub-chunk can equal the maximum (it would have been reported
  // by an earlier sub-chunk), so the wave kernel's plain first-maximum tracking over the whole window is the answer
  // the uint8 order needs the storage-order key of every cell that equals the maximum: keyed tracking
  const bool wave_locate = wave_scoring_ok(p) && !opt().no_wave;
  const bool wave_keyed = p.semantics == MI355_SW_U8SAT;
  auto key_score = [&](size_t k) {
    float score;
    if (qfloat[k] == 2) score = half_value((uint16_t)(keys[k] >> 32)) * kF16Scale;
    else if (qfloat[k] == 3) score = (uint16_t)(keys[k] >> 32) ? half_value((uint16_t)(keys[k] >> 32)) * 256.0f - 1.0f : 0.0f;
    else if (qfloat[k] == 4) { const uint32_t bits = (uint32_t)(keys[k] >> 32); memcpy(&score, &bits, 4); score = std::ldexp(score, ctx->fshift); }
    else if (qfloat[k]) { const uint32_t bits = (uint32_t)(keys[k] >> 32); memcpy(&score, &bits, 4); }
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
    // afford fewer gap columns than the general margin allows (lemma L3)
    int64_t warm = qwarm[k];
    const Margin mg = table.margin(q.len[k]);
    if (mg.finite()) {
      const double spare = std::max(0.0, mg.smax * (double)q.len[k] - (double)score);
      warm = std::min<int64_t>(warm, clamp_cols((double)q.len[k] + std::ceil(spare / mg.g) + 2.0));
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
      if (q.len[k] > 512) pieces = std::max<int64_t>(1, std::min<int64_t>((sub_hi - sub_lo) / 256, (int64_t)(dev_cus() * 7 / 8) / (int64_t)nlong));   // one 1024-thread workgroup per CU
      const int64_t plen = (sub_hi - sub_lo + pieces - 1) / pieces;
      for (int64_t own_lo = sub_lo; own_lo < sub_hi; own_lo += plen) {
        const int64_t own_hi = std::min(own_lo + plen, sub_hi);
        const int64_t wl = std::max<int64_t>(0, own_lo - warm);
        if (wave_locate && q.len[k] <= kWaveMaxLanesSide) {
          WaveJob wj;
          wj.q = (int)k; wj.orient = 0; wj.s_lo = wl; wj.nb = (int32_t)(own_hi - wl); wj.track = true; wj.dirs = false;
          wj.target = score; wj.keyed = wave_keyed;
          // float order: nothing left of the sub-chunk can equal the maximum, so the whole window may compete
          wj.own_lo = wave_keyed ? (int32_t)(own_lo - wl) : 0;
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
  if (!wjobs.empty() && wjobs.size() <= kLatencyJobs && strip_ok) {
    // latency mode: a handful of short problems — a whole wavefront each, few rows per lane
    sjobs.insert(sjobs.end(), wjobs.begin(), wjobs.end());
    wjobs.clear();
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
    for (int R : {3, 5, 8, 10, 16}) {
      std::vector<WaveJob> group;
      for (const WaveJob &j : sjobs) if (strip_R_few(q.len[j.q], sjobs.size(), true) == R) group.push_back(j);
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

// (L5 — candidates of a sampled sweep —, L6 — flagged sub-chunks of a saturating sweep —, L3 for the margin.)
// Queries whose saturating float16 sweep reached the cap (host_score.h make_buckets): exact maximum and first maximum
// cell from the sub-chunks the sweep flagged — each re-evaluated on the pipelined strip kernel (kStripMax) over a window
// with the general warm-up margin in front.  `flagged` = {query id, sub-chunk}; done[k] is set for every query resolved.
int locate_saturated(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
                     const std::vector<int64_t> &qchunk, const std::vector<int64_t> &qwarm, const std::vector<float> &qlower,
                     const ScoreTable &table,
                     const std::vector<std::pair<uint32_t, uint32_t>> &flagged, std::vector<Located> &loc, std::vector<char> &done) {
  HostTrace trace_("locate_saturated");
  const int64_t n = rg.hi - rg.lo;
  std::vector<WaveJob> jobs;
  jobs.reserve(flagged.size());
  // long queries are few and each window occupies one workgroup: cut their sub-chunks into pieces (each with its own
  // margin) so that the idle CUs share the work, as locate_fast does
  size_t nlong = 0;
  for (const auto &f : flagged) nlong += q.len[f.first] > 512 ? 1 : 0;
  for (const auto &f : flagged) {
    const int k = (int)f.first;
    const int64_t sub_len = qchunk[k];
    const int64_t sub_lo = std::max<int64_t>(0, (int64_t)f.second * sub_len - 63);   // the lane lag of the sweep
    const int64_t sub_hi = std::min(((int64_t)f.second + 1) * sub_len, n);
    if (sub_lo >= sub_hi) continue;
    // only cells that hold the maximum M >= qlower[k] (the sweep's key: a lower bound) must come out exact: a path that
    // reaches M within |x| diagonal steps can afford fewer gap columns than the general margin allows (as locate_fast)
    int64_t warm = qwarm[k];
    const Margin mg = table.margin(q.len[k]);
    if (mg.finite() && qlower[k] > 0) {
      const double spare = std::max(0.0, mg.smax * (double)q.len[k] - (double)qlower[k]);
      warm = std::min<int64_t>(warm, clamp_cols((double)q.len[k] + std::ceil(spare / mg.g) + 2.0));
    }
    int64_t pieces = 1;
    if (q.len[k] > 512) pieces = std::max<int64_t>(1, std::min<int64_t>((sub_hi - sub_lo) / 256, (int64_t)(dev_cus() * 7 / 8) / (int64_t)std::max<size_t>(1, nlong)));
    if (warm >= sub_hi) pieces = 1;                               // (every piece would start at column 0: no point in cutting)
    const int64_t plen = (sub_hi - sub_lo + pieces - 1) / pieces;
    for (int64_t own_lo = sub_lo; own_lo < sub_hi; own_lo += plen) {
      const int64_t own_hi = std::min(own_lo + plen, sub_hi);
      const int64_t wl = std::max<int64_t>(0, own_lo - warm);
      WaveJob j;
      j.q = k; j.orient = 0; j.s_lo = wl; j.nb = (int32_t)(own_hi - wl); j.track = true; j.dirs = false; j.maxmode = true;
      j.target = -1.0f; j.own_lo = (int32_t)(own_lo - wl);
      jobs.push_back(j);
    }
  }
  std::vector<unsigned long long> bestkey(q.nq, ~0ull);
  for (int R : {3, 5, 8, 10, 16}) {
    std::vector<WaveJob> group;
    for (const WaveJob &j : jobs) if (strip_R_few(q.len[j.q], jobs.size(), true) == R) group.push_back(j);
    for (size_t lo = 0; lo < group.size(); lo += 4096) {
      std::vector<WaveJob> part(group.begin() + lo, group.begin() + std::min(group.size(), lo + 4096));
      int rc = run_strip(ctx, ref, q, rg, p, part, R);
      if (rc) return rc;
      for (const WaveJob &j : part) {
        if (!(j.best > 0) || j.ci <= 0) continue;
        const unsigned long long kk = host_order_key(p.semantics, j.ci, j.cj, q.len[j.q], n);
        Located &L = loc[j.q];
        if (j.best > L.score || (j.best == L.score && kk < bestkey[j.q])) { L.score = j.best; L.ix = j.ci; L.iy = j.cj; bestkey[j.q] = kk; }
        done[j.q] = 1;
      }
    }
  }
  return 0;
}

// locate_saturated, with the candidates of a lone long query taken from the state its sweep saved where that state covers
// them (host_saved.h): blocks with known left column and top row instead of windows behind a zero border and a margin.
int locate_flagged(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg, const mi355_sw_params &p,
                   const std::vector<int64_t> &qchunk, const std::vector<int64_t> &qwarm, const std::vector<float> &qlower,
                   const ScoreTable &table, const std::vector<std::pair<uint32_t, uint32_t>> &flagged, std::vector<Located> &loc,
                   std::vector<char> &done) {
  const int qid = q.nq == 1 ? q.order[0] : -1;
  const int sr = qid >= 0 ? saved_range_index(ctx, ref, q, rg, p, qid) : -1;
  if (sr < 0 || flagged.empty() || qchunk[qid] != ctx->lsaved.sub_len)
    return locate_saturated(ctx, ref, q, rg, p, qchunk, qwarm, qlower, table, flagged, loc, done);
  std::vector<uint32_t> subs, rest;
  for (const auto &f : flagged) if ((int)f.first == qid) subs.push_back(f.second);
  Located L;
  bool found = false;
  int rc = locate_from_saved(ctx, ref, q, rg, p, sr, qid, qlower[qid], table, subs, L, found, rest);
  if (rc) return rc;
  if (!rest.empty()) {
    ctx->saved_fallbacks += rest.size();
    std::vector<std::pair<uint32_t, uint32_t>> fr;
    for (uint32_t sgl : rest) fr.push_back({(uint32_t)qid, sgl});
    std::vector<Located> l2(q.nq);
    std::vector<char> d2(q.nq, 0);
    rc = locate_saturated(ctx, ref, q, rg, p, qchunk, qwarm, qlower, table, fr, l2, d2);
    if (rc) return rc;
    if (d2[qid]) {
      const int64_t m = q.len[qid], n = rg.hi - rg.lo;
      const bool better = !found || l2[qid].score > L.score ||
                          (l2[qid].score == L.score && host_order_key(p.semantics, l2[qid].ix, l2[qid].iy, m, n) < host_order_key(p.semantics, L.ix, L.iy, m, n));
      if (better) { L = l2[qid]; found = true; }
    }
  }
  if (found) { loc[qid] = L; done[qid] = 1; }
  return 0;
}

float elapsed_us(mi355_sw_ctx *ctx, hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0;
  (void)ctx;
  return ms * 1000.0f;
}

// All queries of `q` against one range of the reference: argmax cells and traceback views (into buffers the context
// keeps until its next call), indexed by query id.
// `pre` (mi355_sw_align_scored_range): the keys of an earlier mi355_sw_score_ranges sweep over this very range stand in
// for the score pass.
int align_range_core(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                     const mi355_sw_params &p, int flags, std::vector<Located> &loc, std::vector<TraceOut> &tout,
                     const ScoredRanges *pre = nullptr, size_t pre_range = 0) {
  HostTrace trace_("align_range");
  const bool want_trace = !(flags & MI355_SW_SCORE_ONLY);
  const size_t nq = q.nq;
  const int64_t n = rg.hi - rg.lo;
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  // (half a million small alignments per call: the per-query host arrays are sized only where a path reads them)
  PoolFromScope pool_from_(want_trace);
  loc.resize(nq);
  tout.resize(want_trace ? nq : 0);
  auto fill_defaults = [&loc, &tout, nq, want_trace]() {
    HostTrace t_("  per-query arrays");
    parallel_for(nq, [&](size_t k0, size_t k1) {
      std::fill(loc.begin() + k0, loc.begin() + k1, Located());
      if (want_trace) std::fill(tout.begin() + k0, tout.begin() + k1, TraceOut());
    });
  };
  // The many-small-alignments batch (below: nothing takes the score kernel) touches these arrays only once its results are down:
  // the defaults are then written while the device works (exact_full_device runs ctx->while_device_works in front of its first
  // wait) — 31 MB of host stores for config 4's 561 356 alignments
  const bool devlist_first = !pre && n < 1024 && p.semantics == MI355_SW_F32 && wave_scoring_ok(p) && !opt().no_wave && !opt().no_devlist;
  if (devlist_first) ctx->while_device_works = fill_defaults;
  else fill_defaults();
  struct RunPending {                                              // (every exit: the deferred work is done or dropped, never left behind)
    mi355_sw_ctx *c;
    ~RunPending() { c->while_device_works = nullptr; }
  } run_pending_{ctx};
  // No score can be positive (uint8 engine whose match score saturates to 0; float engine whose best substitution
  // score is <= 0 with a positive gap): every cell of the matrix is 0 and the defined no-match result stands.
  bool all_zero = false;
  if (p.semantics == MI355_SW_U8SAT) all_zero = u8_params(p).M == 0;
  if (n >= 1 && nq > 0 && !all_zero) {
    const ScoreTable table = plan_table(ref, p);
    if (p.semantics == MI355_SW_F32 && table.ok && !(table.smaxf > 0)) all_zero = true;
  }
  if (n >= 1 && nq > 0 && !all_zero) {
    const ScoreTable table = plan_table(ref, p);
    // references shorter than 1024 columns never take the score kernel (bucket_fast_ok): skip the length classes
    std::vector<Bucket> buckets;
    // the float engine's saturating float16 sweep (make_buckets) needs the strip kernel for its flagged sub-chunks
    bool allow_sat = p.semantics == MI355_SW_F32 && strip_scoring_ok(ref, p);
    // ... and so does the sampled running maximum (sw_score_kernel MK), for the sub-chunks within its slack of the key
    bool allow_sample = strip_scoring_ok(ref, p);
    const size_t nsweep = (n >= 1024 || pre) ? nq : 0;             // only the score-kernel paths (below) index these
    std::vector<char> qfast(nq, 0), qfloat(nsweep, 0), qsat(nsweep, 0), qdone(nsweep, 0);
    std::vector<int64_t> qchunk(nsweep, 0), qwarm(nsweep, 0);
    std::vector<unsigned long long> keys;
    bool any_fast = false;
    if (pre) {
      qfast = pre->qfast; qfloat = pre->qfloat; qchunk = pre->qchunk; qwarm = pre->qwarm;
      keys.assign(pre->keys.begin() + pre_range * nq, pre->keys.begin() + (pre_range + 1) * nq);
      ctx->fshift = pre->fshift;
      for (size_t k = 0; k < nq; ++k) any_fast |= qfast[k] != 0;
      if (pre->sampled && nq == 1 && pre->has_located[pre_range]) { loc[0] = pre->located[pre_range]; qdone[0] = 1; }
    }
    ctx->long_margin = 0;
    // sw_long_kernel swept with an optimistic warm-up margin (long_score_launch): the lone query's maximum must lie above what
    // that margin certifies, else the sweep is repeated with the margin this maximum needs (which then certifies it)
    auto margin_certified = [&](float best) {
      if (!(ctx->long_cert >= 0.0f) || best > ctx->long_cert) return true;
      const double m = (double)q.maxlen;
      ctx->long_margin = (int64_t)(m + std::ceil(((double)table.smax * m - (double)best) / (double)table.gap)) + 2 + 64;
      ctx->last_kernel.cells = 0; ctx->timings[4] = 0; ctx->timings[5] = 0;
      ctx->whole_again += 1;
      path_note(ctx, "margin_again");
      return false;
    };
    // uint8 engine, EARLY EXIT.  The maximum never exceeds 255, and the answer is the 255 that comes first in the
    // skewed storage order: on the lowest anti-diagonal i + j — so within the first sub-chunk s1 that truly holds a 255 or its right
    // neighbour (sub-chunks are longer than |x| + 64 columns) — unless the wrapped bottom-right triangle (the last two
    // sub-chunks) holds one, which the layout stores in front of everything.  Reads whose random BACKGROUND reaches the cap
    // (beyond ~300 bp at 3 / -3 / 2: the linear regime, about 0.6 per row) hold a 255 within the reference's first few hundred
    // columns: when the exact evaluation of sub-chunks 0 and 1 and of the last two finds a 255 in sub-chunk 0, that is s1 = 0,
    // every candidate has been evaluated, and the sweep of the remaining reference cannot change (maximum, first cell) — it is
    // skipped.  (The reference sweeps it all the same; its answer for such reads is decided where the matrix first saturates.)
    bool all_early = false;
    if (!pre && p.semantics == MI355_SW_U8SAT && n >= 1024 && allow_sample && !opt().no_u8_early && nq <= 4096 && table.ok) {
      std::vector<Bucket> bk = make_buckets(ref, q, table, p, n, false, false);
      std::vector<int64_t> echunk(nq, 0), ewarm(nq, 0);
      std::vector<float> elow(nq, 255.0f);
      std::vector<char> edone(nq, 0);
      std::vector<std::pair<uint32_t, uint32_t>> ff;
      bool likely = !bk.empty();
      for (Bucket &b : bk) {
        likely = likely && bucket_fast_ok(ref, table, b, n, p) && 0.3 * (double)table.smax * (double)q.len[q.order[b.first]] >= 255.0;
        if (!likely) break;
        const int64_t E = score_sub_len(p.semantics, b), nsub = (n + E - 1) / E;
        for (int k = 0; k < b.count; ++k) {
          const int id = q.order[b.first + k];
          echunk[id] = E; ewarm[id] = b.warm;
          int64_t seen[4]; int ns = 0;
          for (int64_t s : {(int64_t)0, (int64_t)1, nsub - 2, nsub - 1}) {
            bool dup = s < 0 || s >= nsub;
            for (int t = 0; t < ns; ++t) dup = dup || seen[t] == s;
            if (!dup) { seen[ns++] = s; ff.push_back({(uint32_t)id, (uint32_t)s}); }
          }
        }
      }
      if (likely && !ff.empty()) {
        HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
        int rc = locate_saturated(ctx, ref, q, rg, p, echunk, ewarm, elow, table, ff, loc, edone);
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->timings[1] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
        all_early = true;
        for (size_t k = 0; k < nq && all_early; ++k) {
          const bool wrapped = loc[k].ix + loc[k].iy > n && n >= (int64_t)q.len[k];
          all_early = edone[k] && loc[k].score == 255.0f && (loc[k].iy - 1 < echunk[k] || wrapped);
        }
        if (all_early) {
          for (size_t k = 0; k < nq; ++k) { qfast[k] = 1; qdone[k] = 1; qchunk[k] = echunk[k]; qwarm[k] = ewarm[k]; }
          any_fast = true;
          ctx->early_settled += nq;
          path_note(ctx, "u8_early");
        } else {
          for (size_t k = 0; k < nq; ++k) loc[k] = Located();       // (the sweep decides for everybody)
        }
      }
    }
    for (int attempt = 0; attempt < 4 && !pre && !all_early; ++attempt) {
      buckets.clear();
      if (n >= 1024) buckets = make_buckets(ref, q, table, p, n, allow_sat, allow_sample);
      any_fast = false;
      bool any_sat = false;
      for (Bucket &b : buckets) {
        b.fast = bucket_fast_ok(ref, table, b, n, p); any_fast |= b.fast; any_sat |= b.fast && (b.satflag || b.sampled);
        // (not for the uint8 engine's unsaturated sweep: there every cell that reaches 255 must be exact, wherever it lies)
        b.opt_margin = b.longp && !b.unsat && nq == 1 && !opt().no_opt_margin;   // (certified below for a lone query only)
      }
      if (!any_fast) break;
      const std::vector<Range> ranges{rg};
      int rc = score_begin(ctx, q, ranges, table);
      if (rc) return rc;
      std::fill(qsat.begin(), qsat.end(), 0);
      for (Bucket &b : buckets) {
        if (!b.fast) continue;
        rc = score_launch(ctx, ref, q, ranges, p, table, b);
        if (rc) return rc;
        for (int k = 0; k < b.count; ++k) {
          const int id = q.order[b.first + k];
          qfast[id] = 1; qchunk[id] = b.sub_len; qwarm[id] = b.warm; qsat[id] = b.sampled ? 2 : (b.satflag ? 1 : 0);
          qfloat[id] = b.sem == kSemF16 ? 2 : (b.sem == kSemU8H ? 3 : (b.sem == kSemF32 ? 4 : (sem_is_float(b.sem) ? 1 : 0)));
        }
      }
      rc = score_fetch(ctx, nq, keys);
      if (rc == kRetryNoWait) {                                       // (an expired wait between workgroups: the non-waiting layout)
        ctx->last_kernel.cells = 0; ctx->timings[4] = 0; ctx->timings[5] = 0;
        continue;
      }
      if (rc) return rc;
      if (!any_sat) {
        if (ctx->long_cert >= 0.0f && nq == 1) {                       // exact keys: the lone query's maximum as swept
          float v; const uint32_t hi32 = (uint32_t)(keys[0] >> 32); memcpy(&v, &hi32, 4);
          if (!margin_certified(std::ldexp(v, ctx->fshift))) continue;
        }
        break;
      }
      // the flagged (query, sub-chunk) pairs of the saturating sweep
      unsigned int nflag = 0;
      HIPCHK(ctx, hipMemcpy(&nflag, ctx->flags.p, 4, hipMemcpyDeviceToHost));
      size_t nsatq = 0;
      for (size_t k = 0; k < nq; ++k) nsatq += qsat[k] ? 1 : 0;
      const bool trace_on = opt().trace;
      if (trace_on) std::fprintf(stderr, "[mi355_sw] saturating / sampled sweep: %u candidate sub-chunks of %zu queries (budget %.0f)\n", nflag, nsatq,
                                 64.0 * (double)nsatq + 1024.0);
      auto whole_batch_again = [&]() {
        // saturated nearly everywhere (a background that reaches the cap): the exact sweep instead, for every query
        allow_sat = false;
        allow_sample = false;
        ctx->last_kernel.cells = 0;                                  // the sweep that counts is the one that follows
        ctx->timings[4] = 0; ctx->timings[5] = 0;                    // launches / cells: only the sweep that produced the result
                                                                     // (its device time stays in timings[0]: honest extra cost)
        ctx->whole_again += 1;
        path_note(ctx, "whole_again");
      };
      if (nflag > ctx->flag_cap) { whole_batch_again(); continue; }  // (only the unfiltered saturating sweep can overflow the list)
      std::vector<uint32_t> raw(2 * (size_t)nflag);
      if (nflag) HIPCHK(ctx, hipMemcpy(raw.data(), ctx->flags.as<unsigned int>() + 2, (size_t)nflag * 8, hipMemcpyDeviceToHost));
      std::vector<std::pair<uint32_t, uint32_t>> flagged(nflag);
      for (size_t f = 0; f < nflag; ++f) flagged[f] = {raw[2 * f], raw[2 * f + 1]};
      std::sort(flagged.begin(), flagged.end());
      flagged.erase(std::unique(flagged.begin(), flagged.end()), flagged.end());
      ctx->candidates += flagged.size();
      // Per QUERY, not per call: a query with more candidates than its cap (a poly-A or microsatellite read against a repeat-rich
      // reference flags thousands of near-equal sub-chunks) is swept again on the exact instances, alone with the other
      // offenders; everybody else keeps the candidates of the first sweep.  Only when most of the batch offends (a background
      // that reaches the cap everywhere) is the whole batch swept again.
      const uint32_t qcap = query_flag_cap(nq);
      std::vector<uint32_t> qcount(nq, 0);
      for (const auto &f : flagged) if (f.first < nq) qcount[f.first]++;
      std::vector<int> offenders;
      for (size_t k = 0; k < nq; ++k) if (qsat[k] && qcount[k] > qcap) offenders.push_back((int)k);
      if (trace_on) std::fprintf(stderr, "[mi355_sw] %zu of %zu queries exceed %u candidates\n", offenders.size(), nsatq, qcap);
      // uint8 engine, key at the cap of 255: the answer is the 255 that comes first in the skewed storage order, i.e. on the
      // lowest anti-diagonal i + j (the bottom-right triangle i + j > |y| aside: the order wraps it to the front, and its two
      // sub-chunks are always evaluated).  Sub-chunks are longer than |x| + 64, so that cell lies in the first sub-chunk s1 that truly
      // holds a 255 or in s1 + 1; a sub-chunk that is no candidate holds none (its unsaturated maximum stays below 255, and the
      // saturating rule never exceeds the unsaturated one).  sw_sample_first listed the query's first candidates in order:
      // evaluate them exactly; when a 255 turns up and every candidate up to two sub-chunks right of the winner was on the
      // list, the query is settled without a second sweep (a read inside a repeat family: thousands of equal candidates).
      if (!offenders.empty() && ctx->first_valid && p.semantics == MI355_SW_U8SAT) {
        constexpr int K = kFirstCandidates;
        std::vector<uint32_t> first(nq * (size_t)(K + 1));
        HIPCHK(ctx, hipMemcpy(first.data(), ctx->first.p, first.size() * 4, hipMemcpyDeviceToHost));
        std::vector<std::pair<uint32_t, uint32_t>> ff;
        std::vector<int> tried;
        std::vector<float> qlow(nq, 0.0f);
        for (int id : offenders) {
          if (qsat[id] != 2 || qfloat[id] != 2 || qchunk[id] <= (int64_t)q.len[id] + 64) continue;
          if (half_value((uint16_t)(keys[id] >> 32)) * kF16Scale != 255.0f) continue;
          const uint32_t *f = &first[(size_t)id * (K + 1)];
          const uint32_t cnt = f[0] & 0x7FFFFFFFu;
          if (cnt == 0 || cnt > (uint32_t)K) continue;
          for (uint32_t e = 0; e < cnt; ++e) ff.push_back({(uint32_t)id, f[1 + e]});
          // ... and the last two sub-chunks: the storage order wraps the bottom-right triangle (i + j > |y|) in front of
          // everything else (order_key<1>, region 3), so a 255 there beats any other (locate_fast lists them for the same reason)
          const int64_t nsub = (n + qchunk[id] - 1) / qchunk[id];
          for (int64_t s : {nsub - 2, nsub - 1}) {
            bool listed = s < 0;
            for (uint32_t e = 0; e < cnt; ++e) listed = listed || (int64_t)f[1 + e] == s;
            if (!listed) ff.push_back({(uint32_t)id, (uint32_t)s});
          }
          tried.push_back(id);
          qlow[id] = 255.0f;
        }
        if (!ff.empty()) {
          HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
          rc = locate_saturated(ctx, ref, q, rg, p, qchunk, qwarm, qlow, table, ff, loc, qdone);
          if (rc) return rc;
          HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
          ctx->timings[1] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
          std::vector<char> settled(nq, 0);
          size_t nsettled = 0;
          for (int id : tried) {
            const uint32_t *f = &first[(size_t)id * (K + 1)];
            const uint32_t cnt = f[0] & 0x7FFFFFFFu;
            bool ok = qdone[id] && loc[id].score == 255.0f;
            // list cut short: every candidate up to two sub-chunks right of the winner must have been on it (f[cnt]: the last one
            // listed) — unless the winner lies in the wrapped triangle, which precedes every cell of the sub-chunks not listed
            if (ok && (f[0] & 0x80000000u) && !(loc[id].ix + loc[id].iy > n && n >= (int64_t)q.len[id]))
              ok = (int64_t)f[cnt] >= (loc[id].iy - 1) / qchunk[id] + 2;
            if (ok) { settled[id] = 1; ++nsettled; }
            else { loc[id] = Located(); qdone[id] = 0; }
          }
          if (nsettled) {
            flagged.erase(std::remove_if(flagged.begin(), flagged.end(), [&](const std::pair<uint32_t, uint32_t> &f) { return f.first < nq && settled[f.first]; }),
                          flagged.end());
            offenders.erase(std::remove_if(offenders.begin(), offenders.end(), [&](int id) { return settled[id] != 0; }), offenders.end());
            ctx->first_settled += nsettled;
            path_note(ctx, "first_settled");
          }
          if (trace_on) std::fprintf(stderr, "[mi355_sw] %zu of %zu offenders settled by their first candidates\n", nsettled, tried.size());
        }
      }
      if (!offenders.empty() && (opt().no_requery || 2 * offenders.size() > nsatq)) { whole_batch_again(); continue; }
      if (!offenders.empty()) {
        flagged.erase(std::remove_if(flagged.begin(), flagged.end(), [&](const std::pair<uint32_t, uint32_t> &f) { return qcount[f.first] > qcap; }),
                      flagged.end());
        // a view of the batch that lists only the offenders (same device bytes / offsets / lengths, own sorted id list)
        QueryBatch qo;
        qo.bytes.alias(q.bytes.p); qo.lens.alias(q.lens.p); qo.offs.alias(q.offs.p); qo.cum.alias(q.cum.p);
        qo.len = q.len; qo.off = q.off;
        qo.order.assign(offenders.begin(), offenders.end());
        std::stable_sort(qo.order.begin(), qo.order.end(), [&](int32_t a, int32_t b2) { return q.len[a] < q.len[b2]; });
        qo.nq = offenders.size();
        qo.maxlen = 0;
        for (int id : offenders) qo.maxlen = std::max(qo.maxlen, (int)q.len[id]);
        if (ctx->sel2.ensure(qo.nq * 4 + 16)) return fail(ctx, MI355_SW_ENOMEM, "hipMalloc(score scratch) failed");
        HIPCHK(ctx, hipMemcpy(ctx->sel2.p, qo.order.data(), qo.nq * 4, hipMemcpyHostToDevice));
        qo.sel.alias(ctx->sel2.p);
        std::vector<Bucket> again = make_buckets(ref, qo, table, p, n, false, false);
        rc = score_begin(ctx, q, ranges, table);                     // (clears every key: the first sweep's are on the host)
        if (rc) return rc;
        for (Bucket &b : again) {
          b.fast = bucket_fast_ok(ref, table, b, n, p);
          if (!b.fast) return fail(ctx, MI355_SW_ENODEV, "internal: an exact instance refuses a query the sampled one took");
          rc = score_launch(ctx, ref, qo, ranges, p, table, b);
          if (rc) return rc;
          for (int k = 0; k < b.count; ++k) {
            const int id = qo.order[b.first + k];
            qchunk[id] = b.sub_len; qwarm[id] = b.warm; qsat[id] = 0;
            qfloat[id] = b.sem == kSemF16 ? 2 : (b.sem == kSemU8H ? 3 : (b.sem == kSemF32 ? 4 : (sem_is_float(b.sem) ? 1 : 0)));
          }
        }
        std::vector<unsigned long long> keys2;
        rc = score_fetch(ctx, nq, keys2);
        if (rc == kRetryNoWait) return fail(ctx, MI355_SW_ENODEV, "sw_long_kernel: a pipeline wait expired");   // (offenders never take sw_long_kernel)
        if (rc) return rc;
        for (int id : offenders) keys[id] = keys2[id];
        ctx->requeried += offenders.size();
        path_note(ctx, "requery");
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      std::vector<float> qlower(nq, 0.0f);                          // the sweep's key: a lower bound of the query's maximum
      for (size_t k = 0; k < nq; ++k) {
        if (!qsat[k]) continue;
        const uint32_t hi32 = (uint32_t)(keys[k] >> 32);
        if (qfloat[k] == 2) qlower[k] = half_value((uint16_t)hi32) * kF16Scale;
        else if (qfloat[k] == 4) { float v; memcpy(&v, &hi32, 4); qlower[k] = std::ldexp(v, ctx->fshift); }
      }
      rc = locate_flagged(ctx, ref, q, rg, p, qchunk, qwarm, qlower, table, flagged, loc, qdone);
      if (rc) return rc;
      HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      ctx->timings[1] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
      // every query whose key sits at the cap must have been resolved by a flagged sub-chunk
      for (size_t k = 0; k < nq; ++k)
        if (qsat[k] && !qdone[k] && (qsat[k] == 2 ? (keys[k] >> 32) != 0 : half_value((uint16_t)(keys[k] >> 32)) * kF16Scale >= kF16Scale))
          return fail(ctx, MI355_SW_ENODEV, "internal: a saturated or sampled query without a flagged sub-chunk");
      if (ctx->long_cert >= 0.0f && nq == 1 && !margin_certified(qdone[0] ? loc[0].score : 0.0f)) {
        loc[0] = Located(); qdone[0] = 0;
        continue;
      }
      break;
    }
    if (any_fast) {
      int rc = 0;
      HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      rc = locate_fast(ctx, ref, q, rg, p, qfast, qchunk, qwarm, qfloat, keys.data(), table, loc, &qdone);
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
        for (size_t t = 0; t < fq.size(); ++t) tout[fq[t]] = std::move(ft[t]);
      }
    }
    // Many small whole problems (short reference or short queries: nothing took the score kernel): float engine with
    // identity scoring runs them on device-built job lists, one sorted range per orientation (host_batch.h)
    std::vector<char> &handled = ctx->handled_store;               // (2: written straight into the caller's view, host_batch.h)
    handled.assign(nq, 0);
    bool all_handled = false;
    size_t z_handled = 0;
    if (!any_fast && p.semantics == MI355_SW_F32 && wave_scoring_ok(p) && !opt().no_wave &&
        !opt().no_devlist) {
      auto len_at = [&](size_t pos) { return (int64_t)q.len[q.order[pos]]; };
      auto first_above = [&](int64_t v) {                          // first sorted position whose length exceeds v
        size_t lo = 0, hi = nq;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (len_at(mid) <= v) lo = mid + 1; else hi = mid; }
        return lo;
      };
      const size_t z = first_above(0);                             // empty queries: score 0, nothing to run
      // lanes = rows of x (|x| <= 512, |x| <= |y|) — unless the range fits the lanes and the profile kernel takes it: then
      // every x streams past the shared profile (|x| + 16 steps instead of |y| + 16, on the cheaper cell)
      const bool all_stream = n <= kWaveMaxLanesSide && wave_prof_ok(ref, p, wave_R((int)n), (int)n, true);
      const size_t e0 = all_stream ? z : first_above(std::min<int64_t>(kWaveMaxLanesSide, n));
      const size_t e1 = n <= kWaveMaxLanesSide ? nq : e0;          // lanes = columns of y (|y| <= 512), x streams
      for (size_t pos = 0; pos < z; ++pos) handled[q.order[pos]] = 1;
      all_handled = true; z_handled = z;
      ctx->devlist_done = 0;
      HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      int rc = exact_full_device(ctx, ref, q, rg, p, z, e0 - z, 0, want_trace, loc, tout, handled);
      if (!rc) rc = exact_full_device(ctx, ref, q, rg, p, e0, e1 - e0, 1, want_trace, loc, tout, handled);
      if (rc) return rc;
      HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      ctx->timings[2] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
    }
    if (ctx->while_device_works) { ctx->while_device_works(); ctx->while_device_works = nullptr; }   // (no launch took it)
    std::vector<int> slow;
    all_handled = all_handled && z_handled + ctx->devlist_done == nq;   // (half a million alignments: no scan when the lists took them all)
    if (!all_handled)
      for (size_t k = 0; k < nq; ++k) if (!qfast[k] && !handled[k]) slow.push_back((int)k);
    // Problems the score kernel does not take (no finite warm-up margin: a gap penalty that is, or truncates to, 0 ...) and whose
    // anti-diagonal does not fit the LDS kernel either: the strip kernel over the WHOLE range as one window — a window that starts
    // at column 0 needs no margin — for the first maximum (locate_saturated with one sub-chunk = the range) and for the decisions
    // (wave_trace starts its windows at 0 when no margin is finite).  Identity scoring; not the uint8 engine's |x| == |y| quirk.
    if (!slow.empty() && !pre && wave_scoring_ok(p) && strip_scoring_ok(ref, p) && n > kWaveMaxLanesSide) {
      std::vector<int> huge, rest;
      for (int id : slow) {
        const int64_t m = q.len[id];
        const bool quirk = p.semantics == MI355_SW_U8SAT && m == n;
        if (m > kWaveMaxLanesSide && !quirk && exact_lds_bytes((int)m, (int)std::min<int64_t>(n, INT32_MAX)) > kExactLdsMax) huge.push_back(id);
        else rest.push_back(id);
      }
      if (!huge.empty()) {
        std::vector<int64_t> hchunk(nq, n), hwarm(nq, kColsMax);
        std::vector<float> hlow(nq, 0.0f);
        std::vector<char> hdone(nq, 0);
        std::vector<std::pair<uint32_t, uint32_t>> whole;
        for (int id : huge) whole.push_back({(uint32_t)id, 0u});
        HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
        int rc = locate_saturated(ctx, ref, q, rg, p, hchunk, hwarm, hlow, table, whole, loc, hdone);
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->timings[1] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
        if (want_trace) {
          std::vector<Located> hl;
          for (int id : huge) hl.push_back(loc[id]);
          std::vector<TraceOut> ht;
          HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
          rc = trace_located(ctx, ref, q, rg, p, hwarm, table, huge, hl, ht);
          if (rc) return rc;
          HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
          ctx->timings[2] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
          for (size_t t = 0; t < huge.size(); ++t) tout[huge[t]] = std::move(ht[t]);
        }
        slow.swap(rest);
      }
    }
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
      for (size_t t = 0; t < slow.size(); ++t) { loc[slow[t]] = sl[t]; if (want_trace) tout[slow[t]] = std::move(st[t]); }
    }
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
  ctx->timings[3] += elapsed_us(ctx, ctx->ev[4], ctx->ev[5]);
  return 0;
}

// ... as an array of results with library-owned (malloc) strings: the C-ABI's classic form
int align_range(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                const mi355_sw_params &p, int flags, mi355_sw_result *outs, const ScoredRanges *pre = nullptr, size_t pre_range = 0) {
  const bool want_trace = !(flags & MI355_SW_SCORE_ONLY);
  const size_t nq = q.nq;
  std::vector<Located> &loc = ctx->loc_store;
  std::vector<TraceOut> &tout = ctx->tout_store;
  int rc = align_range_core(ctx, ref, q, rg, p, flags, loc, tout, pre, pre_range);
  if (rc) return rc;
  HostTrace trace_results("set_results");
  const float t_iter = (float)(ctx->timings[0] > 0 ? ctx->timings[0] : ctx->timings[3]);
  parallel_for(nq, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      set_result(outs[k], loc[k].score, loc[k].ix, loc[k].iy, (want_trace && loc[k].score > 0) ? &tout[k] : nullptr);
      outs[k].timings_us[0] = t_iter;
      outs[k].timings_us[1] = 0;
    }
  });
  return 0;
}

// ... as a struct of arrays in context-owned memory, strings as views into the D2H buffers: nothing per alignment is
// allocated (half a million small alignments per call)
int align_range_view(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const Range &rg,
                     const mi355_sw_params &p, int flags, mi355_sw_batch_view *out) {
  const bool want_trace = !(flags & MI355_SW_SCORE_ONLY);
  const size_t nq = q.nq;
  std::vector<Located> &loc = ctx->loc_store;
  std::vector<TraceOut> &tout = ctx->tout_store;
  PoolFromScope pool_from_(want_trace);
  ViewStore &v = ctx->view;
  v.score.resize(nq); v.pos.resize(nq); v.cons_len.resize(nq); v.end_x.resize(nq); v.end_y.resize(nq);
  v.cx.resize(nq); v.cy.resize(nq);
  // the many-small-alignments batch writes the alignments its device list finishes straight into the view when it has them in
  // id order (exact_full_device, handled = 2): one pass over half a million results instead of two
  ctx->handled_store.clear();
  ctx->devlist_direct = 0;
  ctx->direct_view = &v;
  int rc = align_range_core(ctx, ref, q, rg, p, flags, loc, tout);
  ctx->direct_view = nullptr;
  if (rc) return rc;
  HostTrace trace_results("view_results");
  const std::vector<char> &direct = ctx->handled_store;
  const bool some_direct = direct.size() == nq && ctx->devlist_direct > 0;
  if (!(some_direct && ctx->devlist_direct == nq))
  parallel_for(nq, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; ++k) {
      if (some_direct && direct[k] == 2) continue;
      const bool hit = loc[k].score > 0;
      const TraceOut *t = (want_trace && hit) ? &tout[k] : nullptr;
      v.score[k] = loc[k].score;
      v.end_x[k] = hit ? loc[k].ix : 0;
      v.end_y[k] = hit ? loc[k].iy : 0;
      v.pos[k] = t ? t->pos : 0;
      v.cons_len[k] = t ? (uint32_t)t->len : 0;
      v.cx[k] = t && t->len ? t->cx : nullptr;
      v.cy[k] = t && t->len ? t->cy : nullptr;
    }
  });
  out->n = nq;
  out->score = v.score.data(); out->pos = v.pos.data(); out->end_x = v.end_x.data(); out->end_y = v.end_y.data();
  out->cons_x = v.cx.data(); out->cons_y = v.cy.data(); out->cons_len = v.cons_len.data();
  out->timings_us[0] = (float)(ctx->timings[0] > 0 ? ctx->timings[0] : ctx->timings[3]);
  out->timings_us[1] = 0;
  return 0;
}

// Per-range maxima of every query (value half of find_index_of_maximum per piece).
// winner_only (mi355_sw_best_range; DESIGN.md §3.3 lemmas L10, L11): the caller only needs, per query, the first range with the greatest maximum — what
// OMPParallelLocalAligner does with the per-piece maxima (plocalaligner.cpp:122-129).  A lone long query is then swept with the
// sampled maximum (every 4th step: keys are lower bounds within three gaps), and only the ranges whose key lies within that
// slack of the best key — the only ones that can hold the greatest maximum — have their candidate sub-chunks re-evaluated
// exactly; the maxima of the other ranges stay lower bounds.
// known_best / exact_above (winner_only): the sweep of a lone long query uses an optimistic warm-up margin (long_score_launch)
// and is exact for maxima above *exact_above.  known_best > 0: a lower bound of the greatest maximum the caller already knows
// (from other ranks, or from a first call) — the margin is then the one THAT value needs, so that every range whose maximum
// reaches it comes out exact.  exact_above == nullptr: this function certifies its own best by sweeping again when needed.
int range_maxima(mi355_sw_ctx *ctx, const RefData &ref, const QueryBatch &q, const std::vector<Range> &ranges,
                 const mi355_sw_params &p, float *maxima /* [nranges][nq] */, bool winner_only = false, float known_best = 0.0f,
                 float *exact_above = nullptr) {
  const size_t nq = q.nq, nr = ranges.size();
  ctx->scored.valid = false;
  if (exact_above) *exact_above = -1.0f;
  if (nq == 0 || nr == 0) return 0;
  const ScoreTable table = plan_table(ref, p);
  // no positive score possible (see align_range): every maximum is 0
  if ((p.semantics == MI355_SW_U8SAT && u8_params(p).M == 0) ||
      (p.semantics == MI355_SW_F32 && table.ok && !(table.smaxf > 0))) {
    for (size_t k = 0; k < nr * nq; ++k) maxima[k] = 0.0f;
    return 0;
  }
  int64_t maxn = 0;
  for (auto &r : ranges) maxn = std::max(maxn, r.hi - r.lo);
  const bool try_sample = winner_only && nq == 1 && nr <= 4096 && strip_scoring_ok(ref, p) && !opt().no_sample;
  std::vector<Bucket> buckets = make_buckets(ref, q, table, p, maxn, false, try_sample);
  std::vector<char> qfast(nq, 0), qfloat(nq, 0);
  bool sampled = false;
  for (Bucket &b : buckets) {
    b.fast = true;
    for (auto &r : ranges) b.fast = b.fast && bucket_fast_ok(ref, table, b, r.hi - r.lo, p);
    if (!b.longp) b.sampled = false;                               // only sw_long_kernel lays out one value row per range
    sampled |= b.fast && b.sampled;
    b.opt_margin = winner_only && b.longp && !b.unsat && nq == 1 && !opt().no_opt_margin;
  }
  ctx->long_margin = 0;
  if (winner_only && nq == 1 && known_best > 0.0f && table.integral)
    ctx->long_margin = (int64_t)((double)q.maxlen + std::ceil(((double)table.smax * (double)q.maxlen - (double)known_best) / (double)table.gap)) + 2 + 64;
  // what a following mi355_sw_align_scored_range needs (one launch group only: the geometry is per launch)
  ScoredRanges &sc = ctx->scored;
  sc.valid = false;
  const bool keep = nr <= 32768 && p.lut == nullptr;
  if (keep) {
    sc.ref = &ref; sc.batch = &q; sc.ref_version = ref.version; sc.batch_version = q.version; sc.params = p; sc.ranges = ranges;
    sc.keys.assign(nr * nq, 0ull); sc.qfast.assign(nq, 0); sc.qfloat.assign(nq, 0); sc.qchunk.assign(nq, 0); sc.qwarm.assign(nq, 0);
    sc.sampled = false; sc.has_located.assign(nr, 0); sc.located.assign(nr, Located());
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
      for (int k = 0; k < b.count; ++k) {
        const int id = q.order[b.first + k];
        qfast[id] = 1; qfloat[id] = b.sem == kSemF16 ? 2 : (b.sem == kSemU8H ? 3 : (b.sem == kSemF32 ? 4 : (sem_is_float(b.sem) ? 1 : 0)));
        if (keep) { sc.qchunk[id] = b.sub_len; sc.qwarm[id] = b.warm; }
      }
    }
    std::vector<unsigned long long> keys;
    rc = score_fetch(ctx, nq * sub.size(), keys);
    if (rc == kRetryNoWait) {                                       // (an expired wait between workgroups: the non-waiting layout)
      ctx->timings[4] = 0; ctx->timings[5] = 0; ctx->last_kernel.cells = 0;
      return range_maxima(ctx, ref, q, ranges, p, maxima, winner_only, known_best, exact_above);
    }
    if (rc) return rc;
    if (keep) { std::copy(keys.begin(), keys.end(), sc.keys.begin()); sc.qfast = qfast; sc.qfloat = qfloat; sc.fshift = ctx->fshift; }
    for (size_t r = 0; r < sub.size(); ++r)
      for (size_t k = 0; k < nq; ++k)
        if (qfast[k]) {
          const uint32_t hi32 = (uint32_t)(keys[r * nq + k] >> 32);
          float v;
          if (qfloat[k] == 2) v = half_value((uint16_t)hi32) * kF16Scale;
          else if (qfloat[k] == 3) v = (uint16_t)hi32 ? half_value((uint16_t)hi32) * 256.0f - 1.0f : 0.0f;
          else if (qfloat[k] == 4) { memcpy(&v, &hi32, 4); v = std::ldexp(v, ctx->fshift); }
          else if (qfloat[k]) memcpy(&v, &hi32, 4); else v = (float)hi32;
          maxima[(lo + r) * nq + k] = v;
        }
    if (sampled) {
      // (nq == 1, one launch group.)  The keys above are lower bounds within 3 gaps: re-evaluate the contenders exactly.
      unsigned int nflag = 0;
      HIPCHK(ctx, hipMemcpy(&nflag, ctx->flags.p, 4, hipMemcpyDeviceToHost));
      const uint32_t qcap = query_flag_cap(nq);
      float top = 0.0f;
      for (size_t r = 0; r < nr; ++r) top = std::max(top, maxima[r]);
      const float slack = (float)(kLongMK - 1) * table.gapf;
      bool ok = nflag <= ctx->flag_cap && nflag <= qcap;            // (the filter stops appending beyond the cap)
      std::vector<std::vector<std::pair<uint32_t, uint32_t>>> per_range(nr);
      if (ok && nflag) {
        std::vector<uint32_t> raw(2 * (size_t)nflag);
        HIPCHK(ctx, hipMemcpy(raw.data(), ctx->flags.as<unsigned int>() + 2, (size_t)nflag * 8, hipMemcpyDeviceToHost));
        for (size_t f = 0; f < nflag; ++f) {
          const size_t r = (size_t)(raw[2 * f + 1] / (uint64_t)ctx->long_nsub);
          if (r < nr) per_range[r].push_back({raw[2 * f], (uint32_t)(raw[2 * f + 1] % (uint64_t)ctx->long_nsub)});
        }
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      for (size_t r = 0; r < nr && ok; ++r) {
        if (!(maxima[r] > 0.0f) || maxima[r] < top - slack) continue;                     // cannot hold the greatest maximum
        auto &fl = per_range[r];
        std::sort(fl.begin(), fl.end());
        fl.erase(std::unique(fl.begin(), fl.end()), fl.end());
        std::vector<Located> loc(1);
        std::vector<char> done(1, 0);
        const std::vector<float> qlower(1, maxima[r]);
        std::vector<int64_t> qchunk(1, 0), qwarm(1, 0);
        for (Bucket &b : buckets) if (b.fast) { qchunk[0] = b.sub_len; qwarm[0] = b.warm; }
        rc = locate_flagged(ctx, ref, q, ranges[r], p, qchunk, qwarm, qlower, table, fl, loc, done);
        if (rc) return rc;
        if (!done[0]) { ok = false; break; }
        maxima[r] = loc[0].score;
        if (keep) { sc.has_located[r] = 1; sc.located[r] = loc[0]; }
        ctx->candidates += fl.size();
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      ctx->timings[1] += elapsed_us(ctx, ctx->ev[2], ctx->ev[3]);
      if (!ok) {                                                     // too many near-equal candidates: the exact sweep instead
        ctx->whole_again += 1;
        ctx->timings[4] = 0; ctx->timings[5] = 0; ctx->last_kernel.cells = 0;
        return range_maxima(ctx, ref, q, ranges, p, maxima, false);
      }
      if (keep) sc.sampled = true;
    }
    if (winner_only && nq == 1 && ctx->long_cert >= 0.0f) {
      // optimistic margin: exact above long_cert.  The caller merges several ranks' results and checks the GLOBAL best
      // against it (exact_above); without a caller's check, this call certifies its own best
      float top = 0.0f;
      for (size_t r = 0; r < nr; ++r) top = std::max(top, maxima[r]);
      if (exact_above) *exact_above = ctx->long_cert;
      else if (!(top > ctx->long_cert)) {
        ctx->whole_again += 1;
        ctx->timings[4] = 0; ctx->timings[5] = 0; ctx->last_kernel.cells = 0;
        // the margin `top` needs; a second call is exact for every maximum >= top, hence for the true best
        return range_maxima(ctx, ref, q, ranges, p, maxima, true, std::max(top, 1.0f), nullptr);
      }
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
  sc.valid = keep;
  return 0;
}

int check_params(mi355_sw_ctx *ctx, const mi355_sw_params *p) {
  if (!ctx) return MI355_SW_EINVAL;
  if (!p) return fail(ctx, MI355_SW_EINVAL, "params is NULL");
  if (p->semantics != MI355_SW_F32 && p->semantics != MI355_SW_U8SAT) return fail(ctx, MI355_SW_EINVAL, "unknown semantics");
  return 0;
}

void reset_timings(mi355_sw_ctx *ctx) {
  for (double &t : ctx->timings) t = 0;
  ctx->score_ev_used = 0; ctx->arenas.clear(); ctx->cons_used = 0;
  ctx->last_kernel = mi355_sw_kernel_info{};
  ctx->requeried = 0; ctx->whole_again = 0; ctx->candidates = 0; ctx->left_window = 0; ctx->beyond_f16 = 0; ctx->first_settled = 0;
  ctx->saved_locates = 0; ctx->saved_traces = 0; ctx->saved_fallbacks = 0; ctx->wait_retries = 0; ctx->early_settled = 0;
  ctx->path.clear();
}

}  // namespace
