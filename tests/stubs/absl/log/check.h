// stand-in for absl/log/check.h (syntax check only): CHECK* swallow a streamed message
#ifndef MJPC_TEST_STUB_ABSL_CHECK_H_
#define MJPC_TEST_STUB_ABSL_CHECK_H_
namespace absl_stub { struct Sink { template <class T> Sink& operator<<(const T&) { return *this; } }; }
#define MJPC_STUB_CHECK(c) if (c) {} else ::absl_stub::Sink()
#define CHECK(c) MJPC_STUB_CHECK(c)
#define CHECK_EQ(a, b) MJPC_STUB_CHECK((a) == (b))
#define CHECK_NE(a, b) MJPC_STUB_CHECK((a) != (b))
#define CHECK_LT(a, b) MJPC_STUB_CHECK((a) < (b))
#define CHECK_LE(a, b) MJPC_STUB_CHECK((a) <= (b))
#define CHECK_GT(a, b) MJPC_STUB_CHECK((a) > (b))
#define CHECK_GE(a, b) MJPC_STUB_CHECK((a) >= (b))
#define DCHECK(c) MJPC_STUB_CHECK(c)
#endif
