// host_multi.h — several GPUs of one node behind ONE handle, in one process: mi355_sw_multi_* (include/mi355_sw.h).
//
// What the reference spreads over OpenMP threads and MPI ranks becomes devices here:
//   * OMPParallelLocalAligner::calculateScore (src/aligner/plocalaligner.cpp:105-143): piece p of _make_string_range is
//     swept by device p mod ndev (per-piece maxima, :110-115), the maxima are merged under the SERIAL rule (:122-129:
//     strict '>', the lowest piece wins ties) as one 64-bit key (score bits << 32 | ~piece) per device — MAX over
//     devices on the host, or ncclAllReduce(ncclMax, ncclUint64) over xGMI with MI355_SW_MULTI_RCCL — and the owner of
//     the winning piece re-aligns it with default scoring (:132-141).
//   * the independent alignments of src/mpi_sw_solve_uniprot.cpp:95-138 (task farm + writer rank): queries dealt to
//     the devices by length (snake order over the length-sorted batch, so every device gets the same mix of cell
//     counts), reference replicated, no exchange during compute; the batch best (score, lowest index) is the same
//     64-bit key merge.
// One engine context + one host thread per device per call; the per-device work is the single-device pipeline
// (host_pipeline.h), so results are identical to a single device by construction.
// Part of the single translation unit mi355_sw.hip (included there, in order; not a standalone header).
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed when MI355_SW_MULTI_RCCL is asked for

#include <thread>

namespace {

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool load(std::string &err) {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("RCCL not loadable: ") + dlerror(); return false; }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    GetVersion = reinterpret_cast<decltype(GetVersion)>(dlsym(lib, "ncclGetVersion"));
    if (!CommInitAll || !CommDestroy || !AllReduce || !GetErrorString || !GetVersion) { err = "RCCL symbols missing"; return false; }
    return true;
  }
};

// (score, index) -> key: MAX picks the highest score, then the LOWEST index (non-negative floats order like their bits)
inline unsigned long long pack_best(float score, uint32_t index) {
  uint32_t bits;
  memcpy(&bits, &score, 4);
  return ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - index);
}

}  // namespace

struct mi355_sw_multi {
  std::vector<mi355_sw_ctx *> ctx;
  std::vector<int> devices;
  int flags = 0;
  std::string err;
  RcclApi rccl;
  std::vector<ncclComm_t> comms;          // one per device (MI355_SW_MULTI_RCCL)
  std::vector<DevBuf> keybuf;             // 2 x 8 bytes per device: send, receive
  int rccl_version = 0;
  double timings[6] = {0, 0, 0, 0, 0, 0};
};

namespace {

int mfail(mi355_sw_multi *m, int code, const std::string &msg) {
  if (m) m->err = msg;
  return code;
}

// fn(d) on one host thread per device (device 0 on the calling thread); first failure wins
template <class F>
int on_devices(mi355_sw_multi *m, F fn) {
  const size_t n = m->ctx.size();
  std::vector<int> rc(n, 0);
  std::vector<std::thread> th;
  for (size_t d = 1; d < n; ++d) th.emplace_back([&rc, &fn, d]() { rc[d] = fn((int)d); });
  rc[0] = fn(0);
  for (auto &t : th) t.join();
  for (size_t d = 0; d < n; ++d)
    if (rc[d]) { m->err = "device " + std::to_string(m->devices[d]) + ": " + m->ctx[d]->err; return rc[d]; }
  return 0;
}

// MAX over the devices' keys with one ncclAllReduce(ncclMax, ncclUint64) per device over xGMI, in two on_devices passes:
// everything that can fail on ONE device alone (hipSetDevice, staging the key) happens in the first; the collective is entered
// in the second, and only when the first succeeded on every device — a device that failed would otherwise leave the others
// waiting in the all-reduce forever.
int merge_keys_rccl(mi355_sw_multi *m, std::vector<unsigned long long> &keys) {
  int rc = on_devices(m, [&](int d) -> int {
    mi355_sw_ctx *c = m->ctx[d];
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(m->keybuf[d].as<unsigned long long>(), &keys[d], 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
  });
  if (rc) return rc;
  return on_devices(m, [&](int d) -> int {
    mi355_sw_ctx *c = m->ctx[d];
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long *buf = m->keybuf[d].as<unsigned long long>();
    const ncclResult_t r = m->rccl.AllReduce(buf, buf + 1, 1, ncclUint64, ncclMax, m->comms[d], c->stream);
    if (r != ncclSuccess) return fail(c, MI355_SW_ENODEV, std::string("ncclAllReduce: ") + m->rccl.GetErrorString(r));
    HIPCHK(c, hipMemcpyAsync(&keys[d], buf + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
  });
}

void multi_timings(mi355_sw_multi *m) {
  for (double &t : m->timings) t = 0;
  for (mi355_sw_ctx *c : m->ctx) {
    for (int k = 0; k < 4; ++k) m->timings[k] = std::max(m->timings[k], c->timings[k]);   // devices run side by side
    m->timings[4] += c->timings[4];
    m->timings[5] += c->timings[5];
  }
}

}  // namespace

extern "C" {

int mi355_sw_multi_create(mi355_sw_multi **out, int ndev, const int *devices, int flags) {
  if (!out) return MI355_SW_EINVAL;
  *out = nullptr;
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return MI355_SW_ENODEV;
  std::vector<int> devs;
  if (ndev <= 0 || !devices) { for (int d = 0; d < visible; ++d) devs.push_back(d); }     // all visible devices
  else devs.assign(devices, devices + ndev);
  if (devs.size() > 64) return MI355_SW_EINVAL;
  mi355_sw_multi *m = new (std::nothrow) mi355_sw_multi();
  if (!m) return MI355_SW_ENOMEM;
  m->devices = devs;
  m->flags = flags;
  for (int d : devs) {
    mi355_sw_ctx *c = nullptr;
    const int rc = mi355_sw_create(&c, d);
    if (rc) { mi355_sw_multi_destroy(m); return rc; }
    m->ctx.push_back(c);
  }
  if (flags & MI355_SW_MULTI_RCCL) {
    // one communicator per device, all in this process; RCCL needs every device at most once
    std::vector<int> sorted = devs;
    std::sort(sorted.begin(), sorted.end());
    const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    std::string why;
    if (!distinct || !m->rccl.load(why)) { mi355_sw_multi_destroy(m); return MI355_SW_ENOTSUP; }
    m->comms.assign(devs.size(), nullptr);
    const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) { m->comms.clear(); mi355_sw_multi_destroy(m); return MI355_SW_ENODEV; }
    (void)m->rccl.GetVersion(&m->rccl_version);
    m->keybuf.resize(devs.size());
    for (size_t d = 0; d < devs.size(); ++d) {
      (void)hipSetDevice(devs[d]);
      if (m->keybuf[d].ensure(16)) { mi355_sw_multi_destroy(m); return MI355_SW_ENOMEM; }
    }
  }
  *out = m;
  return 0;
}

void mi355_sw_multi_destroy(mi355_sw_multi *m) {
  if (!m) return;
  for (size_t d = 0; d < m->comms.size(); ++d)
    if (m->comms[d]) (void)m->rccl.CommDestroy(m->comms[d]);
  for (size_t d = 0; d < m->keybuf.size(); ++d) { (void)hipSetDevice(m->devices[d]); m->keybuf[d].release(); }
  for (mi355_sw_ctx *c : m->ctx) mi355_sw_destroy(c);
  delete m;
}

const char *mi355_sw_multi_last_error(const mi355_sw_multi *m) { return m ? m->err.c_str() : "null handle"; }
int mi355_sw_multi_set_option(mi355_sw_multi *m, const char *key, const char *value) {
  if (!m || !key) return MI355_SW_EINVAL;
  for (mi355_sw_ctx *c : m->ctx)
    if (option_set(c->opts, key, value)) return mfail(m, MI355_SW_EINVAL, std::string("unknown option: ") + key);
  return 0;
}
int mi355_sw_multi_device_count(const mi355_sw_multi *m) { return m ? (int)m->ctx.size() : 0; }
int mi355_sw_multi_rccl_version(const mi355_sw_multi *m) { return m ? m->rccl_version : 0; }

int mi355_sw_multi_last_timings(const mi355_sw_multi *m, double out[6]) {
  if (!m || !out) return MI355_SW_EINVAL;
  for (int k = 0; k < 6; ++k) out[k] = m->timings[k];
  return 0;
}

int mi355_sw_multi_align_split(mi355_sw_multi *m, const char *x, size_t nx, const char *y, size_t ny,
                               const mi355_sw_params *params, int sm_semantics, int la_semantics,
                               int npiece, float overlap_ratio, mi355_sw_result *out, int *winning_piece) {
  if (!m || m->ctx.empty()) return MI355_SW_EINVAL;
  int rc = check_params(m->ctx[0], params);
  if (rc) return mfail(m, rc, m->ctx[0]->err);
  if (!out || npiece < 1 || (!x && nx) || (!y && ny)) return mfail(m, MI355_SW_EINVAL, "bad argument");
  std::vector<int64_t> lefts(npiece), rights(npiece);
  rc = mi355_sw_make_string_range(npiece, (int64_t)nx, (int64_t)ny, overlap_ratio, lefts.data(), rights.data());
  if (rc) return mfail(m, rc, "_make_string_range: the reference's asserts would fire for these arguments");
  memset(out, 0, sizeof *out);
  const int ndev = (int)m->ctx.size();
  // Every device holds ONLY its own pieces (piece p -> device p mod ndev; each piece's matrix is built on its own sub-string
  // only, plocalaligner.cpp:96-102): the pieces are packed back to back into one staging buffer per device, hashed (the
  // resident copy is reused when a later call passes the same bytes) and uploaded — 1/ndev of the reference plus the overlaps
  // per device instead of all of it.  One device: the caller's buffer as it is.
  OptScope opt_scope0_(m->ctx[0]);
  mi355_sw_params ps = *params;
  ps.semantics = sm_semantics;
  std::vector<unsigned long long> gkey(ndev, 0ull);
  std::vector<const RefData *> refs(ndev, nullptr);
  std::vector<std::vector<Range>> local(ndev);                     // device d's pieces inside ITS resident buffer
  std::vector<std::vector<int>> ids(ndev);
  // first pass: every device stages, hashes and uploads its pieces and the query
  rc = on_devices(m, [&](int d) -> int {
    mi355_sw_ctx *c = m->ctx[d];
    OptScope opt_scope_(c);
    HIPCHK(c, hipSetDevice(c->device));
    reset_timings(c);
    size_t total = 0;
    for (int p = d; p < npiece; p += ndev) {                       // piece p -> device p mod ndev
      ids[d].push_back(p);
      local[d].push_back(ndev == 1 ? Range{lefts[p], rights[p]} : Range{(int64_t)total, (int64_t)total + (rights[p] - lefts[p])});
      total += (size_t)(rights[p] - lefts[p]);
    }
    if (ids[d].empty()) return 0;
    int r = 0;
    if (ndev == 1) {
      r = adhoc_reference(c, y, ny, &refs[d]);
    } else {
      std::unique_ptr<char[]> stage(new char[total + 1]);
      for (size_t k = 0; k < ids[d].size(); ++k)
        memcpy(stage.get() + local[d][k].lo, y + lefts[ids[d][k]], (size_t)(rights[ids[d][k]] - lefts[ids[d][k]]));
      r = adhoc_reference(c, stage.get(), total, &refs[d]);
    }
    if (!r) r = upload_queries(c, c->one, 1, &x, &nx);
    return r;
  });
  if (rc) return rc;
  // Winner-only sweeps (what mi355_sw_best_range offers a rank of the one-process-per-GPU form): all the reference does with the
  // per-piece maxima is pick the first piece with the strictly greatest one (plocalaligner.cpp:122-129).  A lone long query is
  // swept behind an optimistic warm-up margin; the certification is GLOBAL: every device reports the value above which its
  // sweep was exact (the same on all of them), the merged best must exceed it, else every device sweeps again with the
  // margin the merged best needs — decided here, once, for all devices (at most three rounds).
  unsigned long long best = 0;
  float known = 0.0f;
  for (int round = 0; round < 3; ++round) {
    std::vector<float> above(ndev, -1.0f);
    rc = on_devices(m, [&](int d) -> int {
      mi355_sw_ctx *c = m->ctx[d];
      OptScope opt_scope_(c);
      HIPCHK(c, hipSetDevice(c->device));
      unsigned long long key = 0;                                  // "no piece": below every real key
      if (!ids[d].empty()) {
        std::vector<float> mx(ids[d].size(), 0.0f);
        const int r = range_maxima(c, *refs[d], c->one, local[d], ps, mx.data(), true, known, &above[d]);
        if (r) return r;
        for (size_t k = 0; k < ids[d].size(); ++k) key = std::max(key, pack_best(mx[k], (uint32_t)ids[d][k]));
      }
      gkey[d] = key;
      return 0;
    });
    if (rc) return rc;
    if (m->flags & MI355_SW_MULTI_RCCL) {
      // second phase, entered only when every device finished its sweep: nobody can be left waiting in the collective
      rc = merge_keys_rccl(m, gkey);
      if (rc) return rc;
      best = gkey[0];
      for (int d = 1; d < ndev; ++d)
        if (gkey[d] != best) return mfail(m, MI355_SW_ENODEV, "internal: devices disagree after the all-reduce");
    } else {
      best = 0;
      for (int d = 0; d < ndev; ++d) best = std::max(best, gkey[d]);
    }
    float gbest, cert = -1.0f;
    { const uint32_t bits = (uint32_t)(best >> 32); memcpy(&gbest, &bits, 4); }
    for (int d = 0; d < ndev; ++d) cert = std::max(cert, above[d]);
    if (gbest > cert) break;
    if (round == 2) return mfail(m, MI355_SW_ENODEV, "internal: the merged best is not above what the sweeps certify after three rounds");
    known = std::max(gbest, 1.0f);
  }
  const int bp = (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull));   // first piece with the strictly greatest maximum
  const int owner = bp % ndev;
  mi355_sw_ctx *c = m->ctx[owner];
  OptScope opt_scope_(c);
  double t_score = 0;
  for (mi355_sw_ctx *cc : m->ctx) t_score = std::max(t_score, cc->timings[0]);
  if (hipSetDevice(c->device) != hipSuccess) return mfail(m, MI355_SW_ENODEV, "hipSetDevice failed");
  mi355_sw_params pd;
  mi355_sw_default_params(&pd);                                    // LAT(x, piece): default scoring (plocalaligner.cpp:135)
  pd.semantics = la_semantics;
  // default scoring in both roles: the owner finishes its piece from the sweep's keys instead of sweeping it a second time
  // (the reference does sweep twice, :132-136)
  const bool same_sweep = params->lut == nullptr && params->match == 3.0f && params->mismatch == -3.0f && params->gap == 2.0f &&
                          sm_semantics == la_semantics;
  const size_t li = (size_t)(bp / ndev);
  const ScoredRanges &sc = c->scored;
  const bool from_keys = same_sweep && sc.valid && sc.ref == (const void *)refs[owner] && sc.batch == (const void *)&c->one &&
                         li < sc.ranges.size() && (!sc.sampled || sc.has_located[li]);
  rc = align_range(c, *refs[owner], c->one, local[owner][li], pd, 0, out, from_keys ? &sc : nullptr, li);
  if (rc) return mfail(m, rc, c->err);
  if (out->score > 0) { out->pos += (uint32_t)lefts[bp]; out->end_y += lefts[bp]; }
  else out->pos = (uint32_t)lefts[bp];
  out->timings_us[0] = (float)t_score;
  double t_sum = 0;
  for (mi355_sw_ctx *cc : m->ctx) t_sum += cc == c ? t_score : cc->timings[0];
  out->timings_us[1] = (float)t_sum;
  if (winning_piece) *winning_piece = bp;
  multi_timings(m);
  return 0;
}

int mi355_sw_multi_set_reference(mi355_sw_multi *m, const char *y, size_t ny) {
  if (!m || (!y && ny)) return MI355_SW_EINVAL;
  return on_devices(m, [&](int d) -> int {
    mi355_sw_ctx *c = m->ctx[d];
    OptScope opt_scope_(c);
    HIPCHK(c, hipSetDevice(c->device));
    return upload_reference(c, c->ref, y, ny);                     // replicated: every device streams all of it
  });
}

int mi355_sw_multi_align_batch(mi355_sw_multi *m, size_t n, const char *const *xs, const size_t *nxs,
                               const mi355_sw_params *params, int flags, mi355_sw_result *outs, int64_t *best_index) {
  if (!m || m->ctx.empty()) return MI355_SW_EINVAL;
  int rc = check_params(m->ctx[0], params);
  if (rc) return mfail(m, rc, m->ctx[0]->err);
  if ((n && (!xs || !nxs || !outs))) return mfail(m, MI355_SW_EINVAL, "null argument");
  if (n >= 0xFFFFFFFFull) return mfail(m, MI355_SW_EINVAL, "more than 2^32 - 2 alignments per call");
  const int ndev = (int)m->ctx.size();
  if (n) memset(outs, 0, n * sizeof *outs);                      // so that a failed call can release what was filled
  // deal by length: snake order over the length-sorted batch (device loads differ by at most one item per round)
  std::vector<uint32_t> order(n);
  {
    size_t mx = 0;
    for (size_t k = 0; k < n; ++k) mx = std::max(mx, nxs[k]);
    if (n >= 4096 && mx <= ((size_t)1 << 22)) {
      std::vector<uint32_t> start(mx + 2, 0);
      for (size_t k = 0; k < n; ++k) start[nxs[k] + 1]++;
      for (size_t l = 1; l < start.size(); ++l) start[l] += start[l - 1];
      for (size_t k = 0; k < n; ++k) order[start[nxs[k]]++] = (uint32_t)k;
    } else {
      for (size_t k = 0; k < n; ++k) order[k] = (uint32_t)k;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return nxs[a] < nxs[b]; });
    }
  }
  std::vector<std::vector<uint32_t>> part(ndev);
  for (int d = 0; d < ndev; ++d) part[d].reserve(n / ndev + 1);
  for (size_t k = 0; k < n; ++k) {
    const size_t round = k / ndev, pos = k % ndev;
    part[(round & 1) ? ndev - 1 - pos : pos].push_back(order[n - 1 - k]);   // longest first
  }
  std::vector<unsigned long long> gkey(ndev, 0ull);
  rc = on_devices(m, [&](int d) -> int {
    mi355_sw_ctx *c = m->ctx[d];
    OptScope opt_scope_(c);
    HIPCHK(c, hipSetDevice(c->device));
    reset_timings(c);
    const std::vector<uint32_t> &idx = part[d];
    unsigned long long key = 0;
    if (!idx.empty()) {
      std::vector<const char *> sx(idx.size());
      std::vector<size_t> sn(idx.size());
      for (size_t k = 0; k < idx.size(); ++k) { sx[k] = xs[idx[k]]; sn[k] = nxs[idx[k]]; }
      int r = upload_queries(c, c->batch, idx.size(), sx.data(), sn.data());
      if (r) return r;
      std::vector<mi355_sw_result> res(idx.size());
      r = align_range(c, c->ref, c->batch, Range{0, (int64_t)c->ref.n}, *params, flags, res.data());
      if (r) return r;
      for (size_t k = 0; k < idx.size(); ++k) {
        outs[idx[k]] = res[k];                                     // the strings move with the struct
        key = std::max(key, pack_best(res[k].score, idx[k]));
      }
    }
    gkey[d] = key;
    return 0;
  });
  if (!rc && (m->flags & MI355_SW_MULTI_RCCL))                     // second phase: every device finished its share
    rc = merge_keys_rccl(m, gkey);
  if (rc) {                                                        // nothing half-filled leaves the call
    mi355_sw_free_results(outs, n);
    memset(outs, 0, n * sizeof *outs);
    return rc;
  }
  unsigned long long best = 0;
  for (int d = 0; d < ndev; ++d) best = std::max(best, gkey[d]);
  if (best_index) *best_index = n ? (int64_t)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull)) : -1;
  multi_timings(m);
  return 0;
}

}  // extern "C"
