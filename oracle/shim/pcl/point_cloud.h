/*
 * Stand-in for <pcl/point_cloud.h> + <pcl/point_types.h> (PCL is third party,
 * not installed here); see ros/ros.h.  TEST INFRASTRUCTURE ONLY.
 */
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace boost {
using std::make_shared;
using std::shared_ptr;
}   // namespace boost

namespace pcl {
/* same size, alignment and field offsets as pcl::PointXYZI (32 B, 16-aligned) */
struct alignas(16) PointXYZI {
    float x = 0, y = 0, z = 0, _pad = 1.0f;
    float intensity = 0;
    float _pad2[3] = { 0, 0, 0 };
};

struct PCLHeader {
    std::uint32_t seq = 0;
    std::uint64_t stamp = 0;
    std::string frame_id;
};
struct PCLPointCloud2 {};

template <class T>
struct PointCloud {
    PCLHeader header;
    std::vector<T> points;
    void push_back(const T& p) { points.push_back(p); }
};

template <class T>
struct ConditionBase {
    virtual ~ConditionBase() {}
    virtual bool evaluate(const T&) const = 0;
};

/* topic -> points of the last message published on it */
inline std::map<std::string, std::vector<PointXYZI>>& shim_store()
{
    static std::map<std::string, std::vector<PointXYZI>> s;
    return s;
}
inline void shim_capture(const std::string& topic, const PointCloud<PointXYZI>& c)
{
    shim_store()[topic] = c.points;
}
inline void shim_capture(const std::string& topic, const std::shared_ptr<PointCloud<PointXYZI>>& c)
{
    shim_store()[topic] = c->points;
}
}   // namespace pcl
