/*
 * urf_rdp.h -- Douglas-Peucker line simplification, the algorithm behind
 * boost::geometry::simplify(linestring, out, max_distance) that the reference calls at
 * src/lidar_segmentation.cpp:475,512,548.  TEST INFRASTRUCTURE (oracle/): used by the stand-in
 * <boost/geometry.hpp> of oracle A and by oracle B.
 *
 * Boost.Geometry is a third-party dependency that is neither in the reference checkout nor in
 * this image (the reference pins no version; README.md:7 names ROS Kinetic/Melodic, i.e. Boost
 * 1.58 / 1.65).  Restated from its documented behaviour (strategy simplify::douglas_peucker with
 * the projected-point distance strategy): the first and last point are kept; for a span (a, b) the
 * interior point farthest from the SEGMENT a-b is kept iff its distance is strictly greater than
 * max_distance, and both halves are processed recursively; lines of fewer than 3 points and
 * max_distance < 0 are copied.  Coordinates are float (xy = point_xy<float>,
 * data_structures.hpp:38), comparisons are made on squared distances in float.
 * The real library cannot be run here; the behaviour is pinned by known answers instead
 * (tests/test_simplify_kat.py): the worked example of Boost.Geometry's documentation of simplify
 * and hand-derived cases for the documented strategy.
 */
#ifndef URF_RDP_H
#define URF_RDP_H
#ifdef __cplusplus
extern "C" {
#endif
/* in: n points (x[i], y[i]); keep[i] set to 1 for the points of the simplified line (in order) */
void urf_rdp_float(const float* x, const float* y, int n, float max_distance, unsigned char* keep);
#ifdef __cplusplus
}
#endif
#endif
