/* marker.cpp -- see marker.hpp. */
#include "marker.hpp"

namespace urf {

namespace {
float segmentDistance2(const std::array<float, 2>& p, const std::array<float, 2>& a, const std::array<float, 2>& b)
{
    const float vx = b[0] - a[0], vy = b[1] - a[1];
    const float wx = p[0] - a[0], wy = p[1] - a[1];
    const float c1 = wx * vx + wy * vy;
    if (c1 <= 0.0f)
        return wx * wx + wy * wy;
    const float c2 = vx * vx + vy * vy;
    if (c2 <= c1) {
        const float ux = p[0] - b[0], uy = p[1] - b[1];
        return ux * ux + uy * uy;
    }
    const float t = c1 / c2;
    const float qx = a[0] + t * vx, qy = a[1] + t * vy;
    const float dx = p[0] - qx, dy = p[1] - qy;
    return dx * dx + dy * dy;
}

void simplifySpan(const std::vector<std::array<float, 2>>& line, size_t a, size_t b, float tol2, std::vector<char>& keep)
{
    if (b < a + 2)
        return;
    float far2 = -1.0f;
    size_t arg = a;
    for (size_t i = a + 1; i < b; ++i) {
        const float d2 = segmentDistance2(line[i], line[a], line[b]);
        if (d2 > far2) {
            far2 = d2;
            arg = i;
        }
    }
    if (far2 > tol2) {
        keep[arg] = 1;
        simplifySpan(line, a, arg, tol2, keep);
        simplifySpan(line, arg, b, tol2, keep);
    }
}
}   // namespace

std::vector<std::array<float, 2>> simplifyLine(const std::vector<std::array<float, 2>>& line, float tolerance)
{
    if (line.size() < 3 || tolerance < 0.0f)
        return line;
    std::vector<char> keep(line.size(), 0);
    keep.front() = keep.back() = 1;
    simplifySpan(line, 0, line.size() - 1, tolerance * tolerance, keep);
    std::vector<std::array<float, 2>> out;
    for (size_t i = 0; i < line.size(); ++i)
        if (keep[i])
            out.push_back(line[i]);
    return out;
}

}   // namespace urf

/* C view of simplifyLine (include/urf.h): known-answer tests bind it without a C++ compiler */
extern "C" int urf_simplify_line(const float* xy, uint32_t n, float max_distance, uint8_t* keep)
{
    if ((!xy || !keep) && n)
        return URF_ERR_INVALID_ARG;
    std::vector<std::array<float, 2>> line(n);
    for (uint32_t i = 0; i < n; i++)
        line[i] = { xy[2 * i], xy[2 * i + 1] };
    const auto out = urf::simplifyLine(line, max_distance);
    size_t k = 0;   /* the result is a subsequence of the input: mark its members */
    for (uint32_t i = 0; i < n; i++) {
        keep[i] = k < out.size() && out[k] == line[i];
        k += keep[i];
    }
    return k == out.size() ? URF_OK : URF_ERR_INVALID_ARG;
}

namespace urf {

MarkerBuilder::MarkerBuilder() { urf_default_marker_params(&params_); }

/* lidar_segmentation.cpp:471-489 (and :508-526, :544-562): optionally replace the strip's points by the
 * simplified outline at the manual height, append it, start over */
void MarkerBuilder::closeStrip(Marker& strip, MarkerArray& out)
{
    if (params_.simple_poly_allow) {
        strip.points.clear();
        for (const auto& q : simplifyLine(line_, params_.poly_s_param))
            strip.points.push_back({ { (double)q[0], (double)q[1], (double)params_.poly_z_manual } });
    }
    out.markers.push_back(strip);
    strip.points.clear();
    line_.clear();
}

bool MarkerBuilder::build(const float* pts, uint32_t k, MarkerArray& out)
{
    out.markers.clear();
    const int cM = (int)k;
    if (cM <= 2)   /* :371 */
        return false;
    std::vector<float> red(cM);
    for (int i = 0; i < cM; ++i)
        red[i] = pts[4 * i + 3];
    /* :379-415: a point needs a neighbour of its own colour */
    if (red[0] == 0 && red[1] == 1) red[0] = 1;
    if (red[cM - 1] == 0 && red[cM - 2] == 1) red[cM - 1] = 1;
    if (red[0] == 1 && red[1] == 0) red[0] = 0;
    if (red[cM - 1] == 1 && red[cM - 2] == 0) red[cM - 1] = 0;
    for (int i = 2; i <= cM - 3; ++i)
        if (red[i] == 0 && red[i - 1] == 1 && red[i + 1] == 1) red[i] = 1;
    for (int i = 2; i <= cM - 3; ++i)
        if (red[i] == 1 && red[i - 1] == 0 && red[i + 1] == 0) red[i] = 0;

    const std::array<float, 4> green{ { 0.f, 1.f, 0.f, 1.f } }, redc{ { 1.f, 0.f, 0.f, 1.f } };
    Marker strip;
    strip.frame_id = fixed_frame_;
    strip.type = Marker::LINE_STRIP;
    strip.action = Marker::ADD;
    float zavg = 0.0f;
    int stripId = 0;
    auto at = [&](int i) { return std::array<double, 3>{ { (double)pts[4 * i], (double)pts[4 * i + 1], (double)pts[4 * i + 2] } }; };
    auto add = [&](const std::array<double, 3>& p) {
        strip.points.push_back(p);
        line_.push_back({ { (float)p[0], (float)p[1] } });
    };
    for (int i = 0; i < cM; ++i) {   /* :430-579 */
        const auto p = at(i);
        zavg *= (float)i;
        zavg = (float)((double)zavg + p[2]);
        zavg /= (float)(i + 1);
        if (i == 0) {
            add(p);
        } else if (red[i] == red[i - 1]) {
            add(p);
            if (i == cM - 1) {   /* the last strip is only closed on this path (:456) */
                strip.id = stripId;
                strip.color = red[i] == 0 ? green : redc;
                closeStrip(strip, out);
            }
        } else if (red[i] == 0) {   /* red -> green: the joining segment is still red (:495-529) */
            add(p);
            strip.id = stripId++;
            strip.color = redc;
            closeStrip(strip, out);
            add(p);
        } else {                    /* green -> red (:534-577) */
            strip.id = stripId++;
            strip.color = green;
            closeStrip(strip, out);
            add(at(i - 1));
            add(p);
        }
    }
    if (params_.poly_z_avg_allow)   /* :580-589 */
        for (Marker& m : out.markers)
            for (auto& q : m.points)
                q[2] = (double)zavg;
    strip.action = Marker::DELETE;   /* :591-598 obsolete markers of the previous sweep */
    for (int del = stripId; del < ghostcount_; ++del) {
        strip.id++;
        out.markers.push_back(strip);
    }
    ghostcount_ = stripId;
    return true;
}

}   // namespace urf
