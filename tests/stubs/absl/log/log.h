// stand-in for absl/log/log.h (syntax check only)
#ifndef MJPC_TEST_STUB_ABSL_LOG_H_
#define MJPC_TEST_STUB_ABSL_LOG_H_
#include "absl/log/check.h"
#define LOG(sev) ::absl_stub::Sink()
#endif
