// declaration-level stand-in for absl::Span (syntax check of integration/*.cc only; see ../../mujoco/mujoco.h)
#ifndef MJPC_TEST_STUB_ABSL_SPAN_H_
#define MJPC_TEST_STUB_ABSL_SPAN_H_
#include <cstddef>
#include <type_traits>
#include <vector>
namespace absl {
template <class T>
class Span {
 public:
  using element_type = T;
  using value_type = std::remove_cv_t<T>;
  using size_type = std::size_t;
  using iterator = T*;
  constexpr Span() noexcept : p_(nullptr), n_(0) {}
  constexpr Span(T* p, size_type n) noexcept : p_(p), n_(n) {}
  template <class V, class = std::enable_if_t<std::is_same_v<typename V::value_type, value_type>>>
  Span(V& v) noexcept : p_(v.data()), n_(v.size()) {}
  template <class V, class = std::enable_if_t<std::is_const_v<T> && std::is_same_v<typename V::value_type, value_type>>>
  Span(const V& v) noexcept : p_(v.data()), n_(v.size()) {}
  template <class U, class = std::enable_if_t<std::is_const_v<T> && std::is_same_v<U, value_type>>>
  Span(const Span<U>& o) noexcept : p_(o.data()), n_(o.size()) {}
  constexpr T* data() const noexcept { return p_; }
  constexpr size_type size() const noexcept { return n_; }
  constexpr bool empty() const noexcept { return n_ == 0; }
  constexpr T& operator[](size_type i) const { return p_[i]; }
  constexpr T* begin() const { return p_; }
  constexpr T* end() const { return p_ + n_; }
  constexpr Span subspan(size_type pos = 0, size_type len = static_cast<size_type>(-1)) const { return Span(p_ + pos, len == static_cast<size_type>(-1) ? n_ - pos : len); }
 private:
  T* p_;
  size_type n_;
};
template <class T> constexpr Span<T> MakeSpan(T* p, std::size_t n) { return Span<T>(p, n); }
template <class T> constexpr Span<const T> MakeConstSpan(T* p, std::size_t n) { return Span<const T>(p, n); }
template <class C> auto MakeSpan(C& c) { return Span<typename C::value_type>(c.data(), c.size()); }
template <class C> auto MakeConstSpan(const C& c) { return Span<const typename C::value_type>(c.data(), c.size()); }
}  // namespace absl
#endif
