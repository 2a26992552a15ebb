// test_multi.cpp — mi355_sw_multi_* (several GPUs behind one handle) against the single-device entry points:
// the same split (OMPParallelLocalAligner, src/aligner/plocalaligner.cpp:105-143) and the same many-alignment batch
// (src/mpi_sw_solve_uniprot.cpp:95-138 shape) on {0}, {0, 0} (two independent contexts on one GPU: the dealing and
// merging logic without a second card), and — when the box has them — {0, 1} with the host merge and with the RCCL
// all-reduce, and all visible devices.  Every result must be identical to the single-device one.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mi355_sw.h"

#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

static unsigned long long st = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

static bool same(const mi355_sw_result &a, const mi355_sw_result &b) {
  return a.score == b.score && a.pos == b.pos && a.end_x == b.end_x && a.end_y == b.end_y && a.cons_len == b.cons_len &&
         (a.cons_len == 0 || (memcmp(a.cons_x, b.cons_x, a.cons_len) == 0 && memcmp(a.cons_y, b.cons_y, a.cons_len) == 0));
}

int main() {
  mi355_sw_ctx *ctx = nullptr;
  EXPECT(mi355_sw_create(&ctx, 0) == 0);
  // reference with two equal hits of the query in different pieces (the lowest piece must win), fractional scoring
  std::string ref;
  for (int k = 0; k < 400000; ++k) ref.push_back("ACGT"[rnd() & 3]);
  std::string q = ref.substr(123456, 700);
  q[350] = q[350] == 'A' ? 'C' : 'A';
  ref.replace(301000, 700, ref.substr(123456, 700));
  std::vector<std::string> reads;
  for (int k = 0; k < 301; ++k) {
    const size_t len = 30 + rnd() % 600;
    std::string r = ref.substr(rnd() % (ref.size() - len), len);
    if (k % 3 == 0) r[len / 2] = 'N';
    reads.push_back(r);
  }
  std::vector<const char *> xs;
  std::vector<size_t> nxs;
  for (auto &r : reads) { xs.push_back(r.data()); nxs.push_back(r.size()); }

  mi355_sw_params p;
  mi355_sw_default_params(&p);
  struct Case { int sm, la, npiece; float match, mismatch, gap; };
  const Case cases[] = {{MI355_SW_F32, MI355_SW_F32, 7, 3, -3, 2}, {MI355_SW_U8SAT, MI355_SW_U8SAT, 5, 3, -3, 2},
                        {MI355_SW_F32, MI355_SW_U8SAT, 16, 2.5f, -1.5f, 0.5f}, {MI355_SW_F32, MI355_SW_F32, 1, 3, -3, 2}};
  std::vector<mi355_sw_result> split_ref;
  std::vector<int> piece_ref;
  for (const Case &c : cases) {
    mi355_sw_params pc = p;
    pc.match = c.match; pc.mismatch = c.mismatch; pc.gap = c.gap; pc.semantics = c.sm;
    mi355_sw_result r;
    int piece = -1;
    EXPECT(mi355_sw_align_split(ctx, q.data(), q.size(), ref.data(), ref.size(), &pc, c.sm, c.la, c.npiece, 2.0f, &r, &piece) == 0);
    EXPECT(r.score > 1000 || c.la == MI355_SW_U8SAT);
    split_ref.push_back(r);
    piece_ref.push_back(piece);
  }
  EXPECT(mi355_sw_set_reference(ctx, ref.data(), ref.size()) == 0);
  std::vector<mi355_sw_result> batch_ref(reads.size());
  EXPECT(mi355_sw_align_batch(ctx, reads.size(), xs.data(), nxs.data(), &p, 0, batch_ref.data()) == 0);
  size_t best_ref = 0;
  for (size_t k = 0; k < reads.size(); ++k) if (batch_ref[k].score > batch_ref[best_ref].score) best_ref = k;

  int visible = 1;
  {
    mi355_sw_multi *all = nullptr;
    EXPECT(mi355_sw_multi_create(&all, 0, nullptr, 0) == 0);
    visible = mi355_sw_multi_device_count(all);
    mi355_sw_multi_destroy(all);
  }
  std::printf("visible devices: %d\n", visible);
  struct Set { std::vector<int> devs; int flags; const char *name; };
  std::vector<Set> sets = {{{0}, 0, "{0}"}, {{0, 0}, 0, "{0,0}"}, {{0, 0, 0}, 0, "{0,0,0}"}, {{0}, MI355_SW_MULTI_RCCL, "{0} rccl"}};
  if (visible >= 2) {
    sets.push_back({{0, 1}, 0, "{0,1}"});
    sets.push_back({{0, 1}, MI355_SW_MULTI_RCCL, "{0,1} rccl"});
    sets.push_back({{1, 0, 1}, 0, "{1,0,1}"});
    std::vector<int> all;
    for (int d = 0; d < visible; ++d) all.push_back(d);
    sets.push_back({all, MI355_SW_MULTI_RCCL, "all rccl"});
  }
  for (const Set &s : sets) {
    mi355_sw_multi *m = nullptr;
    const int rc = mi355_sw_multi_create(&m, (int)s.devs.size(), s.devs.data(), s.flags);
    if (rc != 0) { std::printf("FAILED create %s rc=%d\n", s.name, rc); return 1; }
    EXPECT(mi355_sw_multi_device_count(m) == (int)s.devs.size());
    if (s.flags & MI355_SW_MULTI_RCCL) EXPECT(mi355_sw_multi_rccl_version(m) > 0);
    for (int rep = 0; rep < 2; ++rep)                        // second pass: every device reuses its resident copy
      for (size_t k = 0; k < sizeof cases / sizeof cases[0]; ++k) {
        const Case &c = cases[k];
        mi355_sw_params pc = p;
        pc.match = c.match; pc.mismatch = c.mismatch; pc.gap = c.gap; pc.semantics = c.sm;
        mi355_sw_result r;
        int piece = -1;
        const int rc2 = mi355_sw_multi_align_split(m, q.data(), q.size(), ref.data(), ref.size(), &pc, c.sm, c.la, c.npiece, 2.0f, &r, &piece);
        if (rc2 != 0) { std::printf("FAILED split %s: %s\n", s.name, mi355_sw_multi_last_error(m)); return 1; }
        EXPECT(piece == piece_ref[k]);
        EXPECT(same(r, split_ref[k]));
        mi355_sw_free_result(&r);
      }
    EXPECT(mi355_sw_multi_set_reference(m, ref.data(), ref.size()) == 0);
    std::vector<mi355_sw_result> res(reads.size());
    int64_t best = -2;
    const int rc3 = mi355_sw_multi_align_batch(m, reads.size(), xs.data(), nxs.data(), &p, 0, res.data(), &best);
    if (rc3 != 0) { std::printf("FAILED batch %s: %s\n", s.name, mi355_sw_multi_last_error(m)); return 1; }
    EXPECT(best == (int64_t)best_ref);
    for (size_t k = 0; k < reads.size(); ++k) EXPECT(same(res[k], batch_ref[k]));
    mi355_sw_free_results(res.data(), res.size());
    int64_t none = 0;
    EXPECT(mi355_sw_multi_align_batch(m, 0, nullptr, nullptr, &p, 0, nullptr, &none) == 0 && none == -1);
    double t[6];
    EXPECT(mi355_sw_multi_last_timings(m, t) == 0);
    mi355_sw_multi_destroy(m);
    std::printf("%s ok\n", s.name);
  }
  {  // RCCL needs distinct devices
    mi355_sw_multi *m = nullptr;
    const int d2[2] = {0, 0};
    EXPECT(mi355_sw_multi_create(&m, 2, d2, MI355_SW_MULTI_RCCL) == MI355_SW_ENOTSUP && m == nullptr);
  }
  for (auto &r : split_ref) mi355_sw_free_result(&r);
  mi355_sw_free_results(batch_ref.data(), batch_ref.size());
  mi355_sw_destroy(ctx);
  std::printf("device sets run: %zu (devices visible: %d)%s\n", sets.size(), visible, visible >= 2 ? "" : "; SKIPPED two-device sets: one device visible");
  std::printf("ALL OK\n");
  return 0;
}
