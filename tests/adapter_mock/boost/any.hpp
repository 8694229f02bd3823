// tests/adapter_mock/boost/any.hpp -- TEST DOUBLE: boost::any as std::any (type(), any_cast), nothing else.
#pragma once
#include <any>
#include <typeinfo>
namespace boost {
class any {
    std::any a_;
public:
    any() {}
    template <class T> any(const T& v) : a_(v) {}
    const std::type_info& type() const { return a_.type(); }
    template <class T> friend T any_cast(const any& x) { return std::any_cast<T>(x.a_); }
};
template <class T> T any_cast(const any& x);
}  // namespace boost
