// stand-in for absl/container/flat_hash_map.h (syntax check only)
#ifndef MJPC_TEST_STUB_ABSL_FLAT_HASH_MAP_H_
#define MJPC_TEST_STUB_ABSL_FLAT_HASH_MAP_H_
#include <unordered_map>
namespace absl { template <class K, class V> using flat_hash_map = std::unordered_map<K, V>; }
#endif
