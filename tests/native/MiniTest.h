// MiniTest.h -- a few-line stand-in for the googletest macros the reference's renderer tests use
// (TEST_F, EXPECT_*), so tests/native/RendererTest.cpp reads like tests/OptiXRendererTests/RendererTest.h.
// googletest is not in this image.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <sstream>
#include <string>
#include <vector>

namespace minitest {

struct TestCase { std::string name; bool needs_gpu; std::function<void()> run; };
inline std::vector<TestCase>& registry() { static std::vector<TestCase> r; return r; }
inline int& failure_count() { static int f = 0; return f; }
struct Registrar { Registrar(const char* name, bool needs_gpu, std::function<void()> run) { registry().push_back({name, needs_gpu, run}); } };

inline void report(const char* file, int line, const std::string& message) {
    ++failure_count();
    if (failure_count() <= 20) fprintf(stderr, "%s:%d: Failure\n  %s\n", file, line, message.c_str());
}

// main: `--cpu` runs only the tests that need no device, `--gpu` only those that do, no flag runs all.
inline int run_all(int argc, char** argv) {
    bool cpu_only = false, gpu_only = false;
    const char* filter = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--cpu")) cpu_only = true;
        else if (!strcmp(argv[i], "--gpu")) gpu_only = true;
        else filter = argv[i];
    }
    int failed_tests = 0, ran = 0;
    for (TestCase& t : registry()) {
        if ((cpu_only && t.needs_gpu) || (gpu_only && !t.needs_gpu)) continue;
        if (filter && t.name.find(filter) == std::string::npos) continue;
        int before = failure_count();
        printf("[ RUN      ] %s\n", t.name.c_str());
        fflush(stdout);
        t.run();
        bool ok = failure_count() == before;
        printf("[ %s ] %s\n", ok ? "      OK" : " FAILED ", t.name.c_str());
        failed_tests += !ok;
        ++ran;
    }
    printf("%d tests ran, %d failed.\n", ran, failed_tests);
    return failed_tests ? 1 : 0;
}

} // namespace minitest

#define MINITEST_F(fixture, name, needs_gpu)                                                                           \
    struct fixture##_##name : fixture { void TestBody(); };                                                            \
    static minitest::Registrar registrar_##fixture##_##name(#fixture "." #name, needs_gpu, [] {                        \
        fixture##_##name t; t.SetUp();                                                                                 \
        if (!(needs_gpu) || t.usable()) t.TestBody(); else minitest::report(__FILE__, __LINE__, "fixture is not usable (no renderer)"); \
        t.TearDown(); });                                                 \
    void fixture##_##name::TestBody()

#define GPU_TEST_F(fixture, name) MINITEST_F(fixture, name, true)
#define CPU_TEST_F(fixture, name) MINITEST_F(fixture, name, false)

#define EXPECT_TRUE(cond) do { if (!(cond)) minitest::report(__FILE__, __LINE__, "expected true: " #cond); } while (0)
#define EXPECT_FALSE(cond) EXPECT_TRUE(!(cond))
#define EXPECT_EQ(expected, actual) do { auto e_ = (expected); auto a_ = (actual); if (!(e_ == a_)) { std::ostringstream s_; s_ << #actual " is " << a_ << ", expected " << e_; minitest::report(__FILE__, __LINE__, s_.str()); } } while (0)
#define EXPECT_FLOAT_EQ_EPS(expected, actual, eps) do { double e_ = (expected), a_ = (actual); if (!(std::fabs(e_ - a_) <= (eps))) { std::ostringstream s_; s_ << #actual " is " << a_ << ", expected " << e_ << " +- " << (eps); minitest::report(__FILE__, __LINE__, s_.str()); } } while (0)
