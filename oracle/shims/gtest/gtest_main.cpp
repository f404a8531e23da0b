// gtest_main.cpp -- runner of the GoogleTest stand-in (oracle/shims/gtest/gtest/gtest.h): GoogleTest's output format
// ("[ RUN      ]", "[       OK ]", "[  FAILED  ]", "[  PASSED  ] N tests.") and --gtest_filter=POS[:POS...][-NEG[:NEG...]]
// with * and ? wildcards, --gtest_list_tests.  Test infrastructure only.
#include <chrono>
#include <cstring>
#include <exception>

#include "gtest/gtest.h"

namespace {

bool wild(const char* p, const char* s) {
    if (*p == 0) return *s == 0;
    if (*p == '*') return wild(p + 1, s) || (*s && wild(p, s + 1));
    return *s && (*p == '?' || *p == *s) && wild(p + 1, s + 1);
}
bool any_of(const std::string& patterns, const std::string& name) {
    size_t b = 0;
    while (b <= patterns.size()) {
        size_t e = patterns.find(':', b);
        if (e == std::string::npos) e = patterns.size();
        if (e > b && wild(patterns.substr(b, e - b).c_str(), name.c_str())) return true;
        b = e + 1;
    }
    return false;
}

}  // namespace

namespace testing {

int RunAllTests(int argc, char** argv) {
    std::string pos = "*", neg;
    bool list = false;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (std::strncmp(a, "--gtest_filter=", 15) == 0) {
            const std::string f = a + 15;
            const size_t d = f.find('-');
            pos = d == std::string::npos ? f : f.substr(0, d);
            neg = d == std::string::npos ? "" : f.substr(d + 1);
            if (pos.empty()) pos = "*";
        } else if (std::strcmp(a, "--gtest_list_tests") == 0) {
            list = true;
        }
    }
    for (auto& e : internal::expanders()) e();
    internal::expanders().clear();
    std::vector<std::string> failed;
    size_t ran = 0, passed = 0, skipped = 0;
    for (const auto& t : internal::registry()) {
        const std::string full = t.suite + "." + t.name;
        if (!any_of(pos, full) || (!neg.empty() && any_of(neg, full))) continue;
        if (list) {
            std::cout << full << "\n";
            continue;
        }
        std::cout << "[ RUN      ] " << full << std::endl;
        internal::State::current() = internal::State{};
        const auto t0 = std::chrono::steady_clock::now();
        try {
            t.run();
        } catch (const std::exception& e) {
            internal::State::current().failed = true;
            std::cout << "unknown file: Failure\nC++ exception with description \"" << e.what() << "\" thrown in the test body." << std::endl;
        } catch (...) {
            internal::State::current().failed = true;
            std::cout << "unknown file: Failure\nUnknown C++ exception thrown in the test body." << std::endl;
        }
        const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        ++ran;
        if (internal::State::current().failed) {
            failed.push_back(full);
            std::cout << "[  FAILED  ] " << full << " (" << ms << " ms)" << std::endl;
        } else if (internal::State::current().skipped) {
            ++skipped;
            std::cout << "[  SKIPPED ] " << full << " (" << ms << " ms)" << std::endl;
        } else {
            ++passed;
            std::cout << "[       OK ] " << full << " (" << ms << " ms)" << std::endl;
        }
    }
    if (list) return 0;
    std::cout << "[==========] " << ran << " tests ran." << std::endl;
    std::cout << "[  PASSED  ] " << passed << " tests." << std::endl;
    if (skipped) std::cout << "[  SKIPPED ] " << skipped << " tests." << std::endl;
    if (!failed.empty()) {
        std::cout << "[  FAILED  ] " << failed.size() << " tests, listed below:" << std::endl;
        for (const auto& f : failed) std::cout << "[  FAILED  ] " << f << std::endl;
    }
    return failed.empty() ? 0 : 1;
}

}  // namespace testing

int main(int argc, char** argv) { return ::testing::RunAllTests(argc, argv); }
