/* detector.cpp -- see detector.hpp.  Host code; all classification happens behind the C ABI. */
#include "detector.hpp"

#include <cstring>

namespace urf {

Detector::Detector(int device, uint32_t max_points) : max_points_(max_points)
{
    /* four scratch rows: the slots of the asynchronous path get a row and a stream each */
    const int rc = urf_create(&ctx_, device, max_points, URF_MAX_IN_FLIGHT);
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

namespace {

/* point i of the message as a pcl::PointXYZI */
struct RecordsXYZI {   /* the message IS an array of pcl::PointXYZI (32 bytes, x y z at 0 / 4 / 8, intensity at 16) */
    const uint8_t* data;
    inline void get(uint32_t i, PointXYZI& out) const { std::memcpy(&out, data + (size_t)i * sizeof(PointXYZI), sizeof(PointXYZI)); }
};
struct RecordsAny {    /* any point_step and field offsets; intensity optional */
    const uint8_t* data;
    uint32_t step, ox, oy, oz;
    int64_t oi;
    inline void get(uint32_t i, PointXYZI& out) const
    {
        const uint8_t* p = data + (size_t)i * step;
        PointXYZI q;   /* pcl::PointXYZI's defaults for what the message does not carry */
        std::memcpy(&q.x, p + ox, 4);
        std::memcpy(&q.y, p + oy, 4);
        std::memcpy(&q.z, p + oz, 4);
        if (oi >= 0)
            std::memcpy(&q.intensity, p + oi, 4);
        out = q;
    }
};
struct RecordsXYZ_I {   /* x y z side by side and intensity right behind them (the layout of the Velodyne / Ouster drivers' messages) */
    const uint8_t* data;
    uint32_t step, ox;
    inline void get(uint32_t i, PointXYZI& out) const
    {
        const uint8_t* p = data + (size_t)i * step + ox;
        PointXYZI q;
        float v[4];
        std::memcpy(v, p, 16);
        q.x = v[0];
        q.y = v[1];
        q.z = v[2];
        q.intensity = v[3];
        out = q;
    }
};

/* lidar_segmentation.cpp:354-367, 605-608, 620: the four clouds from the label bytes, input order.  The clouds have
 * their final sizes already (the sweep's summary holds the four counts), so every point is ONE indexed store -- no
 * push_back, no capacity check; eight labels at a time: a region of interest that drops whole azimuth ranges leaves long
 * runs of zero bytes.  Returns false if the labels do not add up to the counts (cannot happen). */
template <class REC>
bool materialise(const REC& rec, const uint8_t* lab, uint32_t n, bool all_roi, bool lists, std::vector<PointXYZI>& roi,
                 std::vector<PointXYZI>& road, std::vector<PointXYZI>& curb, std::vector<PointXYZI>& probably)
{
    PointXYZI* const o_roi = roi.data();
    PointXYZI* const o_road = road.data();
    PointXYZI* const o_curb = curb.data();
    PointXYZI* const o_prob = probably.data();
    size_t k_roi = all_roi ? roi.size() : 0, k_road = 0, k_curb = 0, k_prob = 0;
    const size_t c_roi = roi.size(), c_road = lists ? road.size() : 0, c_curb = lists ? curb.size() : 0, c_prob = lists ? probably.size() : 0;
    bool ok = true;
    PointXYZI q;
    auto one = [&](uint32_t i, uint8_t l) {
        if (!(l & URF_FLAG_ROI))
            return;
        const bool need = !all_roi || (lists && (l & (URF_LABEL_MASK | URF_FLAG_RING10)));
        if (!need)
            return;
        rec.get(i, q);
        if (!all_roi) {
            if (k_roi < c_roi)
                o_roi[k_roi] = q;
            k_roi++;
        }
        if (lists) {
            if ((l & URF_LABEL_MASK) == URF_LABEL_ROAD) {
                if (k_road < c_road)
                    o_road[k_road] = q;
                k_road++;
            } else if ((l & URF_LABEL_MASK) == URF_LABEL_CURB) {
                if (k_curb < c_curb)
                    o_curb[k_curb] = q;
                k_curb++;
            }
            if (l & URF_FLAG_RING10) {
                if (k_prob < c_prob)
                    o_prob[k_prob] = q;
                k_prob++;
            }
        }
    };
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, lab + i, 8);
        if (w == 0)
            continue;
        for (uint32_t k = 0; k < 8; k++)
            one(i + k, (uint8_t)(w >> (8 * k)));
    }
    for (; i < n; i++)
        one(i, lab[i]);
    ok = k_roi == c_roi && (!lists || (k_road == c_road && k_curb == c_curb && k_prob == c_prob));
    return ok;
}

}   // namespace

void Detector::split(const Pending& m)
{
    road_.header = curb_.header = roi_.header = road_probably_.header = m.header;   /* lidar_segmentation.cpp:612-615 */
    marker_published_ = false;
    if (info_.status != URF_OK) {
        road_.points.clear();
        curb_.points.clear();
        roi_.points.clear();
        road_probably_.points.clear();
        return;
    }
    /* the clouds at their final sizes (a vector that shrinks or stays keeps its storage and initialises nothing: in a stream of
     * sweeps only growth beyond the largest sweep so far costs anything) */
    const bool lists = !reference_order_;   /* in the reference's order they come from the index lists below */
    roi_.points.resize(info_.n_roi);
    road_.points.resize(info_.n_road);
    curb_.points.resize(info_.n_curb);
    road_probably_.points.resize(info_.n_ring10);
    const uint32_t n = m.n;
    const bool xyzi = m.step == sizeof(PointXYZI) && m.ox == 0 && m.oy == 4 && m.oz == 8 && m.oi == 16;
    /* every point inside the region of interest and the message already an array of pcl::PointXYZI: "roi" is the message */
    const bool all_roi = xyzi && info_.n_roi == n && ((uintptr_t)m.data % alignof(PointXYZI)) == 0;
    if (all_roi)
        std::memcpy((void*)roi_.points.data(), m.data, (size_t)n * sizeof(PointXYZI));
    const bool xyz_i = !xyzi && m.oy == m.ox + 4 && m.oz == m.ox + 8 && m.oi == (int64_t)m.ox + 12 && (uint64_t)m.ox + 16 <= m.step;
    bool ok = true;
    if (all_roi && !lists)
        ;   /* nothing left for the label scan */
    else if (xyzi)
        ok = materialise(RecordsXYZI{ m.data }, labels_, n, all_roi, lists, roi_.points, road_.points, curb_.points, road_probably_.points);
    else if (xyz_i)
        ok = materialise(RecordsXYZ_I{ m.data, m.step, m.ox }, labels_, n, false, lists, roi_.points, road_.points, curb_.points,
                         road_probably_.points);
    else
        ok = materialise(RecordsAny{ m.data, m.step, m.ox, m.oy, m.oz, m.oi }, labels_, n, false, lists, roi_.points, road_.points, curb_.points,
                         road_probably_.points);
    if (!ok)
        throw Error(URF_ERR_HIP, "label bytes and summary counters disagree");
    if (marker_on_) {
        float mp[361 * 4];
        uint32_t k = 0;
        check(urf_marker_points(ctx_, 0, mp, &k), "urf_marker_points");
        marker_published_ = marker_.build(mp, k, markers_);
    }
    if (reference_order_) {
        if (ord_.size() < 3 * (size_t)max_points_)
            ord_.resize(3 * (size_t)max_points_);   /* once */
        uint32_t* ro = ord_.data();
        uint32_t* co = ro + max_points_;
        uint32_t* po = co + max_points_;
        uint32_t cnt[3] = { 0, 0, 0 };
        check(urf_ordered_indices(ctx_, 0, ro, co, po, cnt), "urf_ordered_indices");
        const RecordsAny rec{ m.data, m.step, m.ox, m.oy, m.oz, m.oi };
        const RecordsXYZI recx{ m.data };
        PointXYZI q;
        auto fill = [&](std::vector<PointXYZI>& out, const uint32_t* idx, uint32_t c) {
            out.resize(c);   /* (= the summary's count) */
            PointXYZI* const o = out.data();
            for (uint32_t i = 0; i < c; i++) {
                if (xyzi)
                    recx.get(idx[i], q);
                else
                    rec.get(idx[i], q);
                o[i] = q;
            }
        };
        fill(road_.points, ro, cnt[0]);
        fill(curb_.points, co, cnt[1]);
        fill(road_probably_.points, po, cnt[2]);
    }
}

uint32_t Detector::submit(const uint8_t* data, uint32_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                          uint32_t off_z, const Header& header, int64_t off_intensity)
{
    if (off_intensity >= 0 && (uint64_t)off_intensity + 4 > point_step)
        throw Error(URF_ERR_INVALID_ARG, "intensity field outside the record");
    uint32_t t = 0;
    check(urf_classify_pc2_async(ctx_, data, n_points, point_step, off_x, off_y, off_z, &t), "urf_classify_pc2_async");
    Pending& m = pending_[t % URF_MAX_IN_FLIGHT];
    m.data = data;
    m.n = n_points;
    m.step = point_step;
    m.ox = off_x;
    m.oy = off_y;
    m.oz = off_z;
    m.oi = off_intensity;
    m.header = header;
    m.ticket = t;
    m.used = true;
    return t;
}

uint32_t Detector::submit(const PointCloud& cloud)
{
    return submit((const uint8_t*)cloud.points.data(), (uint32_t)cloud.points.size(), (uint32_t)sizeof(PointXYZI), 0, 4, 8,
                  cloud.header, 16);
}

bool Detector::collect(uint32_t ticket)
{
    Pending& m = pending_[ticket % URF_MAX_IN_FLIGHT];
    if (!m.used || m.ticket != ticket)
        throw Error(URF_ERR_INVALID_ARG, "collect: no such ticket in flight");
    m.used = false;
    check(urf_classify_pc2_wait(ctx_, ticket, nullptr, &info_), "urf_classify_pc2_wait");
    check(urf_result_labels(ctx_, ticket, &labels_), "urf_result_labels");   /* read in place: the slot's pinned result buffer */
    n_labels_ = m.n;
    split(m);
    return info_.status == URF_OK;
}

bool Detector::filtered(const PointCloud& cloud)
{
    if (cloud.points.empty()) {   /* (the C ABI refuses an empty message; the reference returns without publishing) */
        Pending m;
        m.header = cloud.header;
        info_ = urf_scan_info{};
        info_.status = URF_TOO_FEW_POINTS;
        labels_ = nullptr;
        n_labels_ = 0;
        split(m);
        return false;
    }
    return collect(submit(cloud));
}

bool Detector::filtered(const uint8_t* data, uint32_t n_points, uint32_t point_step,
                        uint32_t off_x, uint32_t off_y, uint32_t off_z, const Header& header, int64_t off_intensity)
{
    if (n_points == 0) {
        Pending m;
        m.header = header;
        info_ = urf_scan_info{};
        info_.status = URF_TOO_FEW_POINTS;
        labels_ = nullptr;
        n_labels_ = 0;
        split(m);
        return false;
    }
    return collect(submit(data, n_points, point_step, off_x, off_y, off_z, header, off_intensity));
}

/* pcl::fromROSMsg for pcl::PointXYZI: the fields x, y, z and intensity by name (FLOAT32), everything else ignored */
void Detector::resolve(const PointCloud2& msg, uint32_t off[3], int64_t& off_intensity, uint64_t& n)
{
    if (msg.is_bigendian)
        throw Error(URF_ERR_INVALID_ARG, "big-endian PointCloud2 is not supported");
    bool have[3] = { false, false, false };
    off_intensity = -1;
    for (const PointField& f : msg.fields) {
        for (int k = 0; k < 3; k++)
            if (f.name == (k == 0 ? "x" : k == 1 ? "y" : "z")) {
                if (f.datatype != PointField::FLOAT32)
                    throw Error(URF_ERR_INVALID_ARG, "field " + f.name + " is not FLOAT32");
                off[k] = f.offset;
                have[k] = true;
            }
        if (f.name == "intensity" && f.datatype == PointField::FLOAT32)   /* (another type: left at its default, as PCL does with a mismatching field) */
            off_intensity = (int64_t)f.offset;
    }
    if (!have[0] || !have[1] || !have[2])
        throw Error(URF_ERR_INVALID_ARG, "PointCloud2 without x/y/z fields");
    n = (uint64_t)msg.width * msg.height;
    if (msg.point_step < 4 || n * msg.point_step > msg.data.size() || n > 0xffffffffull)
        throw Error(URF_ERR_INVALID_ARG, "PointCloud2 data shorter than width*height*point_step");
}

uint32_t Detector::submit(const PointCloud2& msg)
{
    uint32_t off[3] = { 0, 0, 0 };
    int64_t oi = -1;
    uint64_t n = 0;
    resolve(msg, off, oi, n);
    return submit(msg.data.data(), (uint32_t)n, msg.point_step, off[0], off[1], off[2], msg.header, oi);
}

bool Detector::filtered(const PointCloud2& msg)
{
    uint32_t off[3] = { 0, 0, 0 };
    int64_t oi = -1;
    uint64_t n = 0;
    resolve(msg, off, oi, n);
    return filtered(msg.data.data(), (uint32_t)n, msg.point_step, off[0], off[1], off[2], msg.header, oi);
}

}   // namespace urf
