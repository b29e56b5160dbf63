/*
 * Stand-in for <boost/geometry.hpp> and <boost/assign.hpp>; see ros/ros.h.
 * Only the road_marker polygon uses these.  simplify() is the Douglas-Peucker restatement of
 * oracle/urf_rdp.c (see urf_rdp.h for what is and is not pinned).  TEST INFRASTRUCTURE ONLY.
 */
#pragma once
#include <vector>

#include "urf_rdp.h"

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
void simplify(const C& in, C& out, D max_distance)
{
    const int n = (int)in.size();
    std::vector<float> x(n), y(n);
    std::vector<unsigned char> keep(n ? n : 1);
    for (int i = 0; i < n; i++) {
        x[i] = in[i].px;
        y[i] = in[i].py;
    }
    urf_rdp_float(x.data(), y.data(), n, (float)max_distance, keep.data());
    for (int i = 0; i < n; i++)
        if (keep[i])
            out.push_back(in[i]);   /* boost appends to the output range */
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
