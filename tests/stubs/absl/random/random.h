// stand-in for absl/random/random.h (syntax check only)
#ifndef MJPC_TEST_STUB_ABSL_RANDOM_H_
#define MJPC_TEST_STUB_ABSL_RANDOM_H_
namespace absl {
class BitGen { public: using result_type = unsigned long; result_type operator()(); };
template <class T, class G> T Gaussian(G& gen, T mean = 0, T stddev = 1);
template <class T, class G> T Uniform(G& gen, T lo, T hi);
}
#endif
