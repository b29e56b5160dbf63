/*
 * Stand-in for <boost/geometry.hpp> and <boost/assign.hpp>; see ros/ros.h.
 * Only the road_marker polygon (outside the hot path) uses these, so
 * simplify() is a plain copy.  TEST INFRASTRUCTURE ONLY.
 */
#pragma once
#include <vector>

namespace boost {
namespace geometry {
namespace model {
namespace d2 {
template <class T>
struct point_xy {
    T px, py;
    point_xy(T x = 0, T y = 0) : px(x), py(y) {}
};
}   // namespace d2
template <class P>
struct linestring : std::vector<P> {};
}   // namespace model

template <int I, class T>
T get(const model::d2::point_xy<T>& p)
{
    return I == 0 ? p.px : p.py;
}
template <class C>
void clear(C& c)
{
    c.clear();
}
template <class C, class D>
void simplify(const C& in, C& out, D)
{
    out = in;
}
}   // namespace geometry

namespace assign {
template <class P, class V>
geometry::model::linestring<P>& operator+=(geometry::model::linestring<P>& l, const V& v)
{
    l.push_back(v);
    return l;
}
}   // namespace assign
}   // namespace boost
