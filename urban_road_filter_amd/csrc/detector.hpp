/*
 * detector.hpp -- C++ host adapter with the shape of the reference's Detector
 * (include/urban_road_filter/data_structures.hpp:110-141), on top of the C ABI.
 *
 *   reference                                        here
 *   ------------------------------------------------ -----------------------------------------
 *   Detector::Detector(ros::NodeHandle*)             urf::Detector::Detector(device, max_points)
 *     subscribe + advertise + beam_init()              urf_create(): scratch + beam_init tables
 *     (lidar_segmentation.cpp:51-65)
 *   paramsCallback(config, level) (main.cpp:4-34)    urf::Detector::setParams(const urf_params&)
 *   void filtered(const pcl::PointCloud<PointXYZI>&) bool filtered(const PointCloud&)
 *     (lidar_segmentation.cpp:95)                      false <=> the reference returns without
 *                                                      publishing (< 30 ROI points, :124-126)
 *   pub_road/pub_high/pub_box/pub_pobroad.publish    road() curb() roi() road_probably()
 *     (lidar_segmentation.cpp:612-621)                 clouds carrying the input header
 *
 * A ROS node keeps its subscriber/publishers and calls this class from its
 * callback; see INTEGRATION.md.  Points keep their intensity (the reference copies whole
 * pcl::PointXYZI records into its output clouds, lidar_segmentation.cpp:238-242, 354-367); the order
 * inside the output clouds is input order, or, after setReferenceOrder(true), exactly the
 * reference's (ring-major, azimuth ascending).
 *
 * Nothing is allocated per sweep once the clouds have reached their working size (the reference
 * allocates channels x piece x 64 B per callback, :207): the sweep goes through the four-slot
 * asynchronous path of the C ABI (urf_classify_pc2_async / _wait), the label bytes are read in place
 * from the slot's pinned result buffer (urf_result_labels), the index scratch of the reference
 * order lives in the object.  submit() / collect() expose the slots: up to URF_MAX_IN_FLIGHT sweeps in
 * flight, so that a node's subscriber callback returns after the submission.
 */
#ifndef URF_DETECTOR_HPP
#define URF_DETECTOR_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "marker.hpp"
#include "urf.h"

namespace urf {

/* layout of pcl::PointXYZI: 32 bytes, x y z at 0/4/8, intensity at 16 */
struct alignas(16) PointXYZI {
    float x = 0, y = 0, z = 0, pad0 = 1.0f;
    float intensity = 0;
    float pad1[3] = { 0, 0, 0 };
};

struct Header {
    uint32_t seq = 0;
    uint64_t stamp = 0;
    std::string frame_id;
};

struct PointCloud {
    Header header;
    std::vector<PointXYZI> points;
};

/* sensor_msgs/PointField and sensor_msgs/PointCloud2, field for field */
struct PointField {
    enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
    std::string name;
    uint32_t offset = 0;
    uint8_t datatype = 0;
    uint32_t count = 1;
};
struct PointCloud2 {
    Header header;
    uint32_t height = 1, width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    uint32_t point_step = 0, row_step = 0;
    std::vector<uint8_t> data;
    bool is_dense = false;
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class Detector {
public:
    explicit Detector(int device = 0, uint32_t max_points = 1u << 20);
    ~Detector();
    Detector(const Detector&) = delete;
    Detector& operator=(const Detector&) = delete;

    /* Emit road / curb / road_probably in the reference's own order (ring by ring, ascending
     * azimuth inside a ring -- lidar_segmentation.cpp:289-291, 354-367) instead of input order.
     * Costs one extra per-ring sort on the GPU; roi is in input order either way. */
    void setReferenceOrder(bool on) { reference_order_ = on; }

    /* Also build the "road_marker" MarkerArray (lidar_segmentation.cpp:295-351, 369-602, topic :59,601);
     * fixedFrame = params::fixedFrame (cfg:10).  Off by default: the polygon is a visualisation product. */
    void enableRoadMarker(bool on, const std::string& fixed_frame = "left_os1/os1_lidar")
    {
        marker_on_ = on;
        marker_.setFixedFrame(fixed_frame);
    }
    void setMarkerParams(const urf_marker_params& p) { marker_.setParams(p); }
    /* nullptr when the reference would not publish a MarkerArray for the last sweep */
    const MarkerArray* road_marker() const { return marker_published_ ? &markers_ : nullptr; }

    /* main.cpp:4-34 paramsCallback: callable between scans */
    void setParams(const urf_params& p);
    urf_params params() const;

    /* lidar_segmentation.cpp:95 Detector::filtered.  Returns false when nothing is published. */
    bool filtered(const PointCloud& cloud);
    /* The same for a raw sensor_msgs/PointCloud2 payload: data, point_step, the byte offsets of the
     * x / y / z FLOAT32 fields and (optional, -1 = the message has none) of the FLOAT32 intensity field.
     * The output clouds carry x, y, z and that intensity (0 without the field: what pcl::fromROSMsg leaves
     * in a pcl::PointXYZI whose field the message lacks). */
    bool filtered(const uint8_t* data, uint32_t n_points, uint32_t point_step,
                  uint32_t off_x, uint32_t off_y, uint32_t off_z, const Header& header = Header(),
                  int64_t off_intensity = -1);

    /* The wire message itself: resolves the x / y / z and intensity FLOAT32 fields by name, as
     * pcl::fromROSMsg does for pcl::PointXYZI (every other field is ignored), and classifies
     * width*height points. */
    bool filtered(const PointCloud2& msg);

    /* The same in two halves, up to URF_MAX_IN_FLIGHT sweeps in flight: submit() returns as soon as the
     * sweep is on its way to the device, collect() blocks until it is done and fills road() ... labels().
     * Tickets must be collected in the order they were submitted; the message (cloud.points / data /
     * msg.data) must stay alive and unchanged until its ticket has been collected -- the output clouds are
     * built from it.  A submission while URF_MAX_IN_FLIGHT sweeps are in flight throws Error(URF_ERR_BUSY). */
    uint32_t submit(const PointCloud& cloud);
    uint32_t submit(const uint8_t* data, uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                    uint32_t off_z, const Header& header = Header(), int64_t off_intensity = -1);
    uint32_t submit(const PointCloud2& msg);
    bool collect(uint32_t ticket);

    const PointCloud& road() const { return road_; }                    /* topic "road" */
    const PointCloud& curb() const { return curb_; }                    /* topic "curb" */
    const PointCloud& roi() const { return roi_; }                      /* topic "roi" */
    const PointCloud& road_probably() const { return road_probably_; }  /* topic "road_probably" */
    /* one urf.h label byte per input point of the sweep collected last: n_labels() bytes in the library's
     * pinned result buffer, valid until URF_MAX_IN_FLIGHT further sweeps have been submitted */
    const uint8_t* labels() const { return labels_; }
    uint32_t n_labels() const { return n_labels_; }
    const urf_scan_info& info() const { return info_; }

private:
    struct Pending {
        const uint8_t* data = nullptr;
        uint32_t n = 0, step = 0, ox = 0, oy = 0, oz = 0, ticket = 0;
        int64_t oi = -1;
        Header header;
        bool used = false;
    };
    void check(int rc, const char* what) const;
    void split(const Pending& m);
    static void resolve(const PointCloud2& msg, uint32_t off[3], int64_t& off_intensity, uint64_t& n);
    urf_ctx* ctx_ = nullptr;
    bool reference_order_ = false;
    bool marker_on_ = false, marker_published_ = false;
    MarkerBuilder marker_;
    MarkerArray markers_;
    Pending pending_[URF_MAX_IN_FLIGHT];
    const uint8_t* labels_ = nullptr;
    uint32_t n_labels_ = 0;
    urf_scan_info info_{};
    PointCloud road_, curb_, roi_, road_probably_;
    std::vector<uint32_t> ord_;   /* 3 x max_points: the index lists of the reference order (sized once) */
    uint32_t max_points_ = 0;
};

}   // namespace urf
#endif
