// gtest.h — build-owned MINIMAL stand-in for the googletest API that the reference's own test files use
// (TEST, TEST_F, testing::Test with SetUp / TearDown, ASSERT_EQ, InitGoogleTest, RUN_ALL_TESTS), so that
// /root/reference/test/*.cpp compile WHERE THEY LIE against include/parseq/*.h (tests/cpp/build_dropin.sh) — the
// "drivers and tests compile unchanged" proof.  TEST INFRASTRUCTURE ONLY; not googletest, no code of it.
#pragma once
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace testing {

class Test {
 public:
  virtual ~Test() {}
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
};

struct Registry {
  struct Case { std::string name; std::function<Test *()> make; };
  static std::vector<Case> &cases() { static std::vector<Case> c; return c; }
  static bool &failed() { static bool f = false; return f; }
  static int add(const char *suite, const char *name, std::function<Test *()> make) {
    cases().push_back(Case{std::string(suite) + "." + name, std::move(make)});
    return 0;
  }
};

inline void InitGoogleTest(int *, char **) {}

inline int RunAll() {
  int nfail = 0;
  for (auto &c : Registry::cases()) {
    std::printf("[ RUN      ] %s\n", c.name.c_str());
    Registry::failed() = false;
    Test *t = c.make();
    t->SetUp();
    if (!Registry::failed()) t->TestBody();
    t->TearDown();
    delete t;
    if (Registry::failed()) { ++nfail; std::printf("[  FAILED  ] %s\n", c.name.c_str()); }
    else std::printf("[       OK ] %s\n", c.name.c_str());
  }
  std::printf("[==========] %zu tests ran.\n", Registry::cases().size());
  if (nfail) std::printf("[  FAILED  ] %d tests.\n", nfail);
  else std::printf("[  PASSED  ] %zu tests.\n", Registry::cases().size());
  return nfail ? 1 : 0;
}

}  // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAll()

#define GTEST_SHIM_CASE_(suite, name, base)                                                          \
  class suite##_##name##_Test : public base {                                                        \
   public:                                                                                           \
    void TestBody() override;                                                                        \
  };                                                                                                 \
  static int suite##_##name##_registered_ =                                                          \
      ::testing::Registry::add(#suite, #name, []() -> ::testing::Test * { return new suite##_##name##_Test; }); \
  void suite##_##name##_Test::TestBody()

#define TEST(suite, name) GTEST_SHIM_CASE_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) GTEST_SHIM_CASE_(fixture, name, fixture)

// fatal assertion: marks the case failed and leaves the current function (as googletest's ASSERT_* do)
#define ASSERT_EQ(a, b)                                                                              \
  do {                                                                                               \
    if (!((a) == (b))) {                                                                             \
      std::printf("%s:%d: Failure\n  Expected equality of: %s\n                        %s\n", __FILE__, __LINE__, #a, #b); \
      ::testing::Registry::failed() = true;                                                          \
      return;                                                                                        \
    }                                                                                                \
  } while (0)
#define ASSERT_TRUE(c) ASSERT_EQ(static_cast<bool>(c), true)
#define EXPECT_EQ(a, b)                                                                              \
  do {                                                                                               \
    if (!((a) == (b))) {                                                                             \
      std::printf("%s:%d: Failure\n  Expected equality of: %s\n                        %s\n", __FILE__, __LINE__, #a, #b); \
      ::testing::Registry::failed() = true;                                                          \
    }                                                                                                \
  } while (0)
