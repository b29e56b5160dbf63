/*
 * marker.hpp -- the road_marker MarkerArray of the reference, built on the host from the marker
 * points the GPU finds (urf_marker_points): colour fix-ups, green/red line strips, Douglas-Peucker
 * simplification, polygon height, deletion of obsolete markers
 * (src/lidar_segmentation.cpp:369-602).  Pure host C++, at most 361 points per sweep.
 *
 * visualization_msgs::Marker / MarkerArray are mirrored with the members the reference sets.
 * boost::geometry::simplify (third party, not in the reference checkout nor in this image) is
 * restated as Douglas-Peucker with the point-to-segment distance, float coordinates, "keep iff
 * strictly farther than the tolerance"; the real library cannot be run here, the step is pinned by
 * the worked example of Boost.Geometry's documentation and hand-derived cases (urf_simplify_line).
 */
#ifndef URF_MARKER_HPP
#define URF_MARKER_HPP

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "urf.h"

namespace urf {

struct Marker {
    enum { LINE_STRIP = 4, ADD = 0, DELETE = 2 };
    std::string frame_id;          /* header.frame_id = params::fixedFrame (:424) */
    int32_t type = LINE_STRIP, action = ADD, id = 0;
    std::array<double, 3> position{ { 0, 0, 0 } };
    std::array<double, 4> orientation{ { 0, 0, 0, 1 } };   /* x y z w (:31-34) */
    std::array<double, 3> scale{ { 0.5, 0.5, 0.5 } };      /* :36-38 */
    std::array<float, 4> color{ { 0, 0, 0, 0 } };           /* r g b a */
    std::vector<std::array<double, 3>> points;
};
struct MarkerArray {
    std::vector<Marker> markers;
};

/* Keeps what the reference keeps between callbacks: ghostcount (lidar_segmentation.cpp:23) and the
 * member linestring (data_structures.hpp:139) that holds the points of a strip which was started
 * but not closed. */
class MarkerBuilder {
public:
    MarkerBuilder();
    void setParams(const urf_marker_params& p) { params_ = p; }
    void setFixedFrame(const std::string& f) { fixed_frame_ = f; }
    /* pts: k x {x, y, z, red} from urf_marker_points.  Returns false when the reference would
     * publish no MarkerArray for this sweep (fewer than 3 marker points, :371). */
    bool build(const float* pts, uint32_t k, MarkerArray& out);

private:
    void closeStrip(Marker& strip, MarkerArray& out);
    urf_marker_params params_;
    std::string fixed_frame_;
    int ghostcount_ = 0;
    std::vector<std::array<float, 2>> line_;
};

/* Douglas-Peucker on float points; returns the kept points in order. */
std::vector<std::array<float, 2>> simplifyLine(const std::vector<std::array<float, 2>>& line, float tolerance);

}   // namespace urf
#endif
