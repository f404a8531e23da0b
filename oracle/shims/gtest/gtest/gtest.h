// gtest.h -- a small stand-in for GoogleTest (absent from this image), test infrastructure only: just enough of its
// surface (TEST, TEST_P + INSTANTIATE_TEST_CASE_P with Values / Combine, the EXPECT_* / ASSERT_* macros with streamed
// messages, GTEST_SKIP, --gtest_filter) to compile the reference's own C++ tests FROM WHERE THEY LIE
// (/root/reference/tests/*.cpp) against this repo's mirror of the ouster_core API.  See oracle/Makefile (reftests).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace testing {

class Message {
   public:
    template <class T>
    Message& operator<<(const T& v) {
        ss_ << v;
        return *this;
    }
    Message& operator<<(std::ostream& (*f)(std::ostream&)) {
        ss_ << f;
        return *this;
    }
    std::string str() const { return ss_.str(); }

   private:
    std::ostringstream ss_;
};

namespace internal {

struct State {
    bool failed = false, skipped = false;
    static State& current() {
        static State s;
        return s;
    }
};

template <class T, class = void>
struct Streamable : std::false_type {};
template <class T>
struct Streamable<T, std::void_t<decltype(std::declval<std::ostream&>() << std::declval<const T&>())>> : std::true_type {};

template <class T>
std::string show(const T& v) {
    if constexpr (std::is_same_v<T, bool>) {
        return v ? "true" : "false";
    } else if constexpr (std::is_enum_v<T>) {
        return std::to_string(static_cast<long long>(v));
    } else if constexpr (std::is_same_v<T, unsigned char> || std::is_same_v<T, signed char> || std::is_same_v<T, char>) {
        return std::to_string(static_cast<int>(v));
    } else if constexpr (std::is_same_v<T, std::nullptr_t>) {
        return "nullptr";
    } else if constexpr (Streamable<T>::value) {
        std::ostringstream ss;
        ss << v;
        return ss.str();
    } else {
        return "<" + std::to_string(sizeof(T)) + "-byte object>";
    }
}

struct Result {
    bool ok;
    std::string msg;
    explicit operator bool() const { return ok; }
};

#define OUSTER_GTEST_SHIM_CMP(name, op)                                                                        \
    template <class A, class B>                                                                                \
    Result name(const char* ea, const char* eb, const A& a, const B& b) {                                      \
        if (a op b) return {true, {}};                                                                         \
        return {false, std::string("Expected: (") + ea + ") " #op " (" + eb + "), actual: " + show(a) + " vs " + show(b)}; \
    }
OUSTER_GTEST_SHIM_CMP(CmpEQ, ==)
OUSTER_GTEST_SHIM_CMP(CmpNE, !=)
OUSTER_GTEST_SHIM_CMP(CmpLT, <)
OUSTER_GTEST_SHIM_CMP(CmpLE, <=)
OUSTER_GTEST_SHIM_CMP(CmpGT, >)
OUSTER_GTEST_SHIM_CMP(CmpGE, >=)
#undef OUSTER_GTEST_SHIM_CMP

inline Result CmpStr(const char* ea, const char* eb, const char* a, const char* b, bool want_equal) {
    const bool eq = (a == nullptr || b == nullptr) ? a == b : std::strcmp(a, b) == 0;
    if (eq == want_equal) return {true, {}};
    return {false, std::string("Expected: ") + ea + (want_equal ? " == " : " != ") + eb + ", actual: \"" + (a ? a : "(null)") +
                       "\" vs \"" + (b ? b : "(null)") + "\""};
}
inline Result CmpStr(const char* ea, const char* eb, const std::string& a, const std::string& b, bool want_equal) {
    return CmpStr(ea, eb, a.c_str(), b.c_str(), want_equal);
}

template <class F>
Result CmpUlp(const char* ea, const char* eb, F a, F b) {   // GoogleTest: equal within 4 units in the last place
    using I = std::conditional_t<sizeof(F) == 4, int32_t, int64_t>;
    auto biased = [](F v) {
        I i;
        std::memcpy(&i, &v, sizeof v);
        return i < 0 ? ~i + 1 : static_cast<I>(i | (I{1} << (sizeof(I) * 8 - 1)));
    };
    bool ok = !(std::isnan(a) || std::isnan(b));
    if (ok) {
        const auto ua = static_cast<std::make_unsigned_t<I>>(biased(a)), ub = static_cast<std::make_unsigned_t<I>>(biased(b));
        ok = (ua > ub ? ua - ub : ub - ua) <= 4;
    }
    if (ok) return {true, {}};
    return {false, std::string("Expected: ") + ea + " ~= " + eb + ", actual: " + show(a) + " vs " + show(b)};
}
inline Result CmpNear(const char* ea, const char* eb, double a, double b, double tol) {
    if (std::fabs(a - b) <= tol) return {true, {}};
    return {false, std::string("Expected: |") + ea + " - " + eb + "| <= " + show(tol) + ", actual: " + show(a) + " vs " + show(b)};
}

class AssertHelper {
   public:
    AssertHelper(const char* file, int line, std::string what, bool skip = false)
        : file_(file), line_(line), what_(std::move(what)), skip_(skip) {}
    void operator=(const Message& m) const {
        const std::string extra = m.str();
        if (skip_) {
            State::current().skipped = true;
            std::cout << file_ << ":" << line_ << ": Skipped" << (extra.empty() ? "" : "\n") << extra << std::endl;
            return;
        }
        State::current().failed = true;
        std::cout << file_ << ":" << line_ << ": Failure\n" << what_ << (extra.empty() ? "" : "\n") << extra << std::endl;
    }

   private:
    const char* file_;
    int line_;
    std::string what_;
    bool skip_;
};

}  // namespace internal

class Test {
   public:
    virtual ~Test() = default;
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
    static void RecordProperty(const std::string&, const std::string&) {}
    static void RecordProperty(const std::string&, int) {}
    static bool HasFailure() { return internal::State::current().failed; }
    static bool HasFatalFailure() { return internal::State::current().failed; }
};

template <class T>
class WithParamInterface {
   public:
    using ParamType = T;
    const T& GetParam() const { return *slot(); }
    static const T*& slot() {
        static const T* p = nullptr;
        return p;
    }
};
template <class T>
class TestWithParam : public Test, public WithParamInterface<T> {};

namespace internal {

struct TestCase {
    std::string suite, name;
    std::function<void()> run;
};
inline std::vector<TestCase>& registry() {
    static std::vector<TestCase> r;
    return r;
}
inline std::vector<std::function<void()>>& expanders() {   // INSTANTIATE_* bodies, run once every TEST_P has registered
    static std::vector<std::function<void()>> e;
    return e;
}
inline void run_one(Test& t) {
    t.SetUp();
    if (!State::current().failed && !State::current().skipped) t.TestBody();
    t.TearDown();
}
template <class T>
bool add_test(const char* suite, const char* name) {
    registry().push_back({suite, name, [] {
                              T t;
                              run_one(t);
                          }});
    return true;
}
template <class Fixture>
std::vector<std::pair<std::string, std::function<void()>>>& param_tests() {
    static std::vector<std::pair<std::string, std::function<void()>>> v;
    return v;
}
template <class Fixture, class T>
bool add_param_test(const char* name) {
    param_tests<Fixture>().push_back({name, [] {
                                          T t;
                                          run_one(t);
                                      }});
    return true;
}

template <class... Ts>
struct ValueArray {
    std::tuple<Ts...> v;
    template <class P>
    std::vector<P> as() const {
        return std::apply([](const auto&... x) { return std::vector<P>{static_cast<P>(x)...}; }, v);
    }
};
template <class... Gs>
struct CombineGen {
    std::tuple<Gs...> g;
};
template <class P, class G>
std::vector<P> generate(const G& g);
template <class P, class... Ts>
std::vector<P> generate_impl(const ValueArray<Ts...>& g) {
    return g.template as<P>();
}
template <class Tuple, class Gens, size_t I>
void combine_rec(const Gens& g, Tuple& cur, std::vector<Tuple>& out) {
    if constexpr (I == std::tuple_size_v<Tuple>) {
        out.push_back(cur);
    } else {
        for (const auto& v : generate<std::tuple_element_t<I, Tuple>>(std::get<I>(g))) {
            std::get<I>(cur) = v;
            combine_rec<Tuple, Gens, I + 1>(g, cur, out);
        }
    }
}
template <class P, class... Gs>
std::vector<P> generate_impl(const CombineGen<Gs...>& g) {
    std::vector<P> out;
    P cur{};
    combine_rec<P, std::tuple<Gs...>, 0>(g.g, cur, out);
    return out;
}
template <class P, class G>
std::vector<P> generate(const G& g) {
    return generate_impl<P>(g);
}

template <class Fixture, class G>
bool instantiate(const char* prefix, const char* fixture, G gen) {
    expanders().push_back([=] {
        using P = typename Fixture::ParamType;
        auto values = std::make_shared<std::vector<P>>(generate<P>(gen));
        for (size_t i = 0; i < values->size(); ++i)
            for (const auto& t : param_tests<Fixture>())
                registry().push_back({std::string(prefix) + "/" + fixture, t.first + "/" + std::to_string(i), [values, i, t] {
                                          Fixture::slot() = &(*values)[i];
                                          t.second();
                                      }});
    });
    return true;
}

}  // namespace internal

template <class... Ts>
internal::ValueArray<Ts...> Values(Ts... v) {
    return {std::make_tuple(v...)};
}
template <class... Gs>
internal::CombineGen<Gs...> Combine(Gs... g) {
    return {std::make_tuple(g...)};
}
inline internal::ValueArray<bool, bool> Bool() { return {std::make_tuple(false, true)}; }

int RunAllTests(int argc, char** argv);   // gtest_main.cpp

}  // namespace testing

#define OUSTER_GTEST_CLASS_(suite, name) suite##_##name##_Test

#define TEST(suite, name)                                                                             \
    class OUSTER_GTEST_CLASS_(suite, name) : public ::testing::Test {                                 \
        void TestBody() override;                                                                     \
        static bool registered_;                                                                      \
    };                                                                                                \
    bool OUSTER_GTEST_CLASS_(suite, name)::registered_ =                                              \
        ::testing::internal::add_test<OUSTER_GTEST_CLASS_(suite, name)>(#suite, #name);               \
    void OUSTER_GTEST_CLASS_(suite, name)::TestBody()

#define TEST_F(fixture, name)                                                                         \
    class OUSTER_GTEST_CLASS_(fixture, name) : public fixture {                                       \
        void TestBody() override;                                                                     \
        static bool registered_;                                                                      \
    };                                                                                                \
    bool OUSTER_GTEST_CLASS_(fixture, name)::registered_ =                                            \
        ::testing::internal::add_test<OUSTER_GTEST_CLASS_(fixture, name)>(#fixture, #name);           \
    void OUSTER_GTEST_CLASS_(fixture, name)::TestBody()

#define TEST_P(fixture, name)                                                                         \
    class OUSTER_GTEST_CLASS_(fixture, name) : public fixture {                                       \
        void TestBody() override;                                                                     \
        static bool registered_;                                                                      \
    };                                                                                                \
    bool OUSTER_GTEST_CLASS_(fixture, name)::registered_ =                                            \
        ::testing::internal::add_param_test<fixture, OUSTER_GTEST_CLASS_(fixture, name)>(#name);      \
    void OUSTER_GTEST_CLASS_(fixture, name)::TestBody()

#define INSTANTIATE_TEST_SUITE_P(prefix, fixture, ...)                                                \
    static bool ouster_gtest_inst_##prefix##_##fixture =                                              \
        ::testing::internal::instantiate<fixture>(#prefix, #fixture, __VA_ARGS__)
#define INSTANTIATE_TEST_CASE_P INSTANTIATE_TEST_SUITE_P

#define OUSTER_GTEST_AMBIGUOUS_ELSE_ switch (0) case 0: default:
#define OUSTER_GTEST_NONFATAL_(what) ::testing::internal::AssertHelper(__FILE__, __LINE__, what) = ::testing::Message()
#define OUSTER_GTEST_FATAL_(what) return ::testing::internal::AssertHelper(__FILE__, __LINE__, what) = ::testing::Message()

#define OUSTER_GTEST_PRED_(expr, on_fail) \
    OUSTER_GTEST_AMBIGUOUS_ELSE_ if (const ::testing::internal::Result gtest_r_ = (expr)) ; else on_fail(gtest_r_.msg)
#define OUSTER_GTEST_BOOL_(cond, text, want, on_fail) \
    OUSTER_GTEST_AMBIGUOUS_ELSE_ if (static_cast<bool>(cond) == want) ; else on_fail(std::string("Value of: " text "\n  Expected: ") + (want ? "true" : "false"))

#define EXPECT_TRUE(c) OUSTER_GTEST_BOOL_(c, #c, true, OUSTER_GTEST_NONFATAL_)
#define EXPECT_FALSE(c) OUSTER_GTEST_BOOL_(c, #c, false, OUSTER_GTEST_NONFATAL_)
#define ASSERT_TRUE(c) OUSTER_GTEST_BOOL_(c, #c, true, OUSTER_GTEST_FATAL_)
#define ASSERT_FALSE(c) OUSTER_GTEST_BOOL_(c, #c, false, OUSTER_GTEST_FATAL_)

#define EXPECT_EQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpEQ(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define EXPECT_NE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpNE(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define EXPECT_LT(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpLT(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define EXPECT_LE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpLE(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define EXPECT_GT(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpGT(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define EXPECT_GE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpGE(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define ASSERT_EQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpEQ(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define ASSERT_NE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpNE(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define ASSERT_LT(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpLT(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define ASSERT_LE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpLE(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define ASSERT_GT(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpGT(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define ASSERT_GE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpGE(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define EXPECT_STREQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpStr(#a, #b, a, b, true), OUSTER_GTEST_NONFATAL_)
#define EXPECT_STRNE(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpStr(#a, #b, a, b, false), OUSTER_GTEST_NONFATAL_)
#define ASSERT_STREQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpStr(#a, #b, a, b, true), OUSTER_GTEST_FATAL_)
#define EXPECT_FLOAT_EQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpUlp<float>(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define EXPECT_DOUBLE_EQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpUlp<double>(#a, #b, a, b), OUSTER_GTEST_NONFATAL_)
#define ASSERT_FLOAT_EQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpUlp<float>(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define ASSERT_DOUBLE_EQ(a, b) OUSTER_GTEST_PRED_(::testing::internal::CmpUlp<double>(#a, #b, a, b), OUSTER_GTEST_FATAL_)
#define EXPECT_NEAR(a, b, tol) OUSTER_GTEST_PRED_(::testing::internal::CmpNear(#a, #b, a, b, tol), OUSTER_GTEST_NONFATAL_)
#define ASSERT_NEAR(a, b, tol) OUSTER_GTEST_PRED_(::testing::internal::CmpNear(#a, #b, a, b, tol), OUSTER_GTEST_FATAL_)

// The statement runs in the test body itself (not in a lambda): ASSERT_* inside it may `return` from the test, as with GoogleTest.
#define OUSTER_GTEST_CAT2_(a, b) a##b
#define OUSTER_GTEST_CAT_(a, b) OUSTER_GTEST_CAT2_(a, b)
#define OUSTER_GTEST_THROW_(stmt, extype, on_fail)                                                                   \
    OUSTER_GTEST_AMBIGUOUS_ELSE_ if (::testing::internal::Result gtest_r_ = ::testing::internal::Result{true, {}}) { \
        bool gtest_caught_ = false;                                                                                  \
        try {                                                                                                        \
            stmt;                                                                                                    \
        } catch (const extype&) {                                                                                    \
            gtest_caught_ = true;                                                                                    \
        } catch (...) {                                                                                              \
            gtest_r_ = {false, "Expected: " #stmt " throws " #extype ".\n  Actual: it throws a different type."};    \
            goto OUSTER_GTEST_CAT_(gtest_label_throw_, __LINE__);                                                    \
        }                                                                                                            \
        if (!gtest_caught_) {                                                                                        \
            gtest_r_ = {false, "Expected: " #stmt " throws " #extype ".\n  Actual: it throws nothing."};             \
            goto OUSTER_GTEST_CAT_(gtest_label_throw_, __LINE__);                                                    \
        }                                                                                                            \
    } else                                                                                                           \
        OUSTER_GTEST_CAT_(gtest_label_throw_, __LINE__) : on_fail(gtest_r_.msg)
#define OUSTER_GTEST_NO_THROW_(stmt, on_fail)                                                                        \
    OUSTER_GTEST_AMBIGUOUS_ELSE_ if (::testing::internal::Result gtest_r_ = ::testing::internal::Result{true, {}}) { \
        try {                                                                                                        \
            stmt;                                                                                                    \
        } catch (const std::exception& e) {                                                                          \
            gtest_r_ = {false, std::string("Expected: " #stmt " doesn't throw.\n  Actual: it throws: ") + e.what()}; \
            goto OUSTER_GTEST_CAT_(gtest_label_nothrow_, __LINE__);                                                  \
        } catch (...) {                                                                                              \
            gtest_r_ = {false, "Expected: " #stmt " doesn't throw.\n  Actual: it throws."};                          \
            goto OUSTER_GTEST_CAT_(gtest_label_nothrow_, __LINE__);                                                  \
        }                                                                                                            \
    } else                                                                                                           \
        OUSTER_GTEST_CAT_(gtest_label_nothrow_, __LINE__) : on_fail(gtest_r_.msg)
#define OUSTER_GTEST_ANY_THROW_(stmt, on_fail)                                                                       \
    OUSTER_GTEST_AMBIGUOUS_ELSE_ if (::testing::internal::Result gtest_r_ = ::testing::internal::Result{true, {}}) { \
        bool gtest_caught_ = false;                                                                                  \
        try {                                                                                                        \
            stmt;                                                                                                    \
        } catch (...) {                                                                                              \
            gtest_caught_ = true;                                                                                    \
        }                                                                                                            \
        if (!gtest_caught_) {                                                                                        \
            gtest_r_ = {false, "Expected: " #stmt " throws.\n  Actual: it doesn't."};                                \
            goto OUSTER_GTEST_CAT_(gtest_label_anythrow_, __LINE__);                                                 \
        }                                                                                                            \
    } else                                                                                                           \
        OUSTER_GTEST_CAT_(gtest_label_anythrow_, __LINE__) : on_fail(gtest_r_.msg)
#define EXPECT_THROW(stmt, extype) OUSTER_GTEST_THROW_(stmt, extype, OUSTER_GTEST_NONFATAL_)
#define ASSERT_THROW(stmt, extype) OUSTER_GTEST_THROW_(stmt, extype, OUSTER_GTEST_FATAL_)
#define EXPECT_NO_THROW(stmt) OUSTER_GTEST_NO_THROW_(stmt, OUSTER_GTEST_NONFATAL_)
#define ASSERT_NO_THROW(stmt) OUSTER_GTEST_NO_THROW_(stmt, OUSTER_GTEST_FATAL_)
#define EXPECT_ANY_THROW(stmt) OUSTER_GTEST_ANY_THROW_(stmt, OUSTER_GTEST_NONFATAL_)
#define ASSERT_ANY_THROW(stmt) OUSTER_GTEST_ANY_THROW_(stmt, OUSTER_GTEST_FATAL_)

#define GTEST_SKIP() return ::testing::internal::AssertHelper(__FILE__, __LINE__, "", true) = ::testing::Message()
#define SUCCEED() static_cast<void>(::testing::Message())
#define ADD_FAILURE() OUSTER_GTEST_NONFATAL_("Failed")
#define FAIL() OUSTER_GTEST_FATAL_("Failed")
#define SCOPED_TRACE(msg) static_cast<void>(0)
#define RUN_ALL_TESTS() ::testing::RunAllTests(0, nullptr)
