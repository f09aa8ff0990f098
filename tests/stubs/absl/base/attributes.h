// stand-in for absl/base/attributes.h (syntax check only)
#ifndef MJPC_TEST_STUB_ABSL_ATTRIBUTES_H_
#define MJPC_TEST_STUB_ABSL_ATTRIBUTES_H_
#define ABSL_CONST_INIT
#define ABSL_ATTRIBUTE_UNUSED
#define ABSL_MUST_USE_RESULT
#endif
