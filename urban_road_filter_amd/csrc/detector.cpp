/* detector.cpp -- see detector.hpp.  Host code; all classification happens behind the C ABI. */
#include "detector.hpp"

#include <cstring>

namespace urf {

Detector::Detector(int device, uint32_t max_points)
{
    const int rc = urf_create(&ctx_, device, max_points, 1);
    if (rc != URF_OK)
        throw Error(rc, std::string("urf_create: ") + urf_strerror(rc));
}

Detector::~Detector()
{
    if (ctx_)
        urf_destroy(ctx_);
}

void Detector::check(int rc, const char* what) const
{
    if (rc < 0)
        throw Error(rc, std::string(what) + ": " + urf_strerror(rc) + " " + urf_last_error(ctx_));
}

void Detector::setParams(const urf_params& p) { check(urf_set_params(ctx_, &p), "urf_set_params"); }

urf_params Detector::params() const
{
    urf_params p;
    check(urf_get_params(ctx_, &p), "urf_get_params");
    return p;
}

void Detector::split(const PointXYZI* pts, uint32_t n, const Header& h)
{
    road_.points.clear();
    curb_.points.clear();
    roi_.points.clear();
    road_probably_.points.clear();
    road_.header = curb_.header = roi_.header = road_probably_.header = h;   /* lidar_segmentation.cpp:612-615 */
    marker_published_ = false;
    if (info_.status != URF_OK)
        return;
    road_.points.reserve(info_.n_road);
    curb_.points.reserve(info_.n_curb);
    roi_.points.reserve(info_.n_roi);
    road_probably_.points.reserve(info_.n_ring10);
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t l = labels_[i];
        if (!(l & URF_FLAG_ROI))
            continue;
        roi_.points.push_back(pts[i]);
        if ((l & URF_LABEL_MASK) == URF_LABEL_ROAD)
            road_.points.push_back(pts[i]);
        else if ((l & URF_LABEL_MASK) == URF_LABEL_CURB)
            curb_.points.push_back(pts[i]);
        if (l & URF_FLAG_RING10)
            road_probably_.points.push_back(pts[i]);
    }
    if (marker_on_) {
        float mp[361 * 4];
        uint32_t k = 0;
        check(urf_marker_points(ctx_, 0, mp, &k), "urf_marker_points");
        marker_published_ = marker_.build(mp, k, markers_);
    }
    if (reference_order_) {
        std::vector<uint32_t> ro(n), co(n), po(n);
        uint32_t cnt[3] = { 0, 0, 0 };
        check(urf_ordered_indices(ctx_, 0, ro.data(), co.data(), po.data(), cnt), "urf_ordered_indices");
        road_.points.clear();
        curb_.points.clear();
        road_probably_.points.clear();
        for (uint32_t i = 0; i < cnt[0]; i++)
            road_.points.push_back(pts[ro[i]]);
        for (uint32_t i = 0; i < cnt[1]; i++)
            curb_.points.push_back(pts[co[i]]);
        for (uint32_t i = 0; i < cnt[2]; i++)
            road_probably_.points.push_back(pts[po[i]]);
    }
}

bool Detector::filtered(const PointCloud& cloud)
{
    const uint32_t n = (uint32_t)cloud.points.size();
    labels_.assign(n, 0);
    check(urf_classify_pc2(ctx_, (const uint8_t*)cloud.points.data(), n, (uint32_t)sizeof(PointXYZI), 0, 4, 8,
                           labels_.data(), &info_),
          "urf_classify_pc2");
    split(cloud.points.data(), n, cloud.header);
    return info_.status == URF_OK;
}

bool Detector::filtered(const uint8_t* data, uint32_t n_points, uint32_t point_step,
                        uint32_t off_x, uint32_t off_y, uint32_t off_z, const Header& header)
{
    labels_.assign(n_points, 0);
    check(urf_classify_pc2(ctx_, data, n_points, point_step, off_x, off_y, off_z, labels_.data(), &info_),
          "urf_classify_pc2");
    std::vector<PointXYZI> pts(n_points);
    for (uint32_t i = 0; i < n_points; i++) {
        const uint8_t* p = data + (size_t)i * point_step;
        std::memcpy(&pts[i].x, p + off_x, 4);
        std::memcpy(&pts[i].y, p + off_y, 4);
        std::memcpy(&pts[i].z, p + off_z, 4);
        pts[i].intensity = (float)i;
    }
    split(pts.data(), n_points, header);
    return info_.status == URF_OK;
}

bool Detector::filtered(const PointCloud2& msg)
{
    if (msg.is_bigendian)
        throw Error(URF_ERR_INVALID_ARG, "big-endian PointCloud2 is not supported");
    uint32_t off[3] = { 0, 0, 0 };
    bool have[3] = { false, false, false };
    for (const PointField& f : msg.fields)
        for (int k = 0; k < 3; k++)
            if (f.name == (k == 0 ? "x" : k == 1 ? "y" : "z")) {
                if (f.datatype != PointField::FLOAT32)
                    throw Error(URF_ERR_INVALID_ARG, "field " + f.name + " is not FLOAT32");
                off[k] = f.offset;
                have[k] = true;
            }
    if (!have[0] || !have[1] || !have[2])
        throw Error(URF_ERR_INVALID_ARG, "PointCloud2 without x/y/z fields");
    const uint64_t n = (uint64_t)msg.width * msg.height;
    if (msg.point_step < 4 || n * msg.point_step > msg.data.size())
        throw Error(URF_ERR_INVALID_ARG, "PointCloud2 data shorter than width*height*point_step");
    return filtered(msg.data.data(), (uint32_t)n, msg.point_step, off[0], off[1], off[2], msg.header);
}

}   // namespace urf
