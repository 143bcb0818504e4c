// Tiny assertion helper for the host-layer test programs (gtest is not available here).
#ifndef MINITEST_H_
#define MINITEST_H_
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace minitest {
inline int &failures() { static int f = 0; return f; }
inline int &checks() { static int c = 0; return c; }
struct Test { const char *name; void (*fn)(); };
inline std::vector<Test> &registry() { static std::vector<Test> r; return r; }
struct Registrar { Registrar(const char *n, void (*f)()) { registry().push_back({n, f}); } };
// filter: only the tests whose "Suite.Name" starts with it (nullptr: all)
inline int run_all(const char *filter = nullptr) {
  for (auto &t : registry()) {
    if (filter && std::string(t.name).rfind(filter, 0) != 0) continue;
    const int before = failures();
    try { t.fn(); }
    catch (std::exception &e) { std::printf("  uncaught exception in %s: %s\n", t.name, e.what()); failures()++; }
    std::printf("[%s] %s\n", failures() == before ? " OK " : "FAIL", t.name);
  }
  std::printf("%d checks, %d failures\n", checks(), failures());
  return failures() ? 1 : 0;
}
}  // namespace minitest

#define TEST(suite, name)                                                         \
  static void suite##_##name();                                                   \
  static minitest::Registrar reg_##suite##_##name(#suite "." #name, suite##_##name); \
  static void suite##_##name()
#define EXPECT_TRUE(c)                                                                     \
  do { minitest::checks()++; if (!(c)) { minitest::failures()++;                           \
       std::printf("  %s:%d: EXPECT_TRUE(%s) failed\n", __FILE__, __LINE__, #c); } } while (0)
#define EXPECT_FALSE(c) EXPECT_TRUE(!(c))
#define EXPECT_EQ(a, b)                                                                    \
  do { minitest::checks()++; if (!((a) == (b))) { minitest::failures()++;                  \
       std::printf("  %s:%d: EXPECT_EQ(%s, %s) failed\n", __FILE__, __LINE__, #a, #b); } } while (0)
#define EXPECT_NE(a, b) EXPECT_TRUE(!((a) == (b)))
#define EXPECT_THROW(stmt, ex)                                                             \
  do { minitest::checks()++; bool caught_ = false; try { stmt; } catch (ex &) { caught_ = true; } \
       catch (...) {}                                                                      \
       if (!caught_) { minitest::failures()++;                                             \
       std::printf("  %s:%d: EXPECT_THROW(%s, %s) failed\n", __FILE__, __LINE__, #stmt, #ex); } } while (0)
#define EXPECT_NO_THROW(stmt)                                                              \
  do { minitest::checks()++; try { stmt; } catch (std::exception &e_) { minitest::failures()++; \
       std::printf("  %s:%d: EXPECT_NO_THROW(%s) threw %s\n", __FILE__, __LINE__, #stmt, e_.what()); } } while (0)
#endif
