/* Stand-in for visualization_msgs/Marker(.h|Array.h); see ros/ros.h.  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include <ros/ros.h>

namespace visualization_msgs {
struct Marker {
    enum { LINE_STRIP = 4, ADD = 0, DELETE = 2 };
    std_msgs::Header header;
    int type = 0, action = 0, id = 0;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color;
    ros::Duration lifetime;
    std::vector<geometry_msgs::Point> points;
};
struct MarkerArray {
    std::vector<Marker> markers;
};
/* the last MarkerArray published (topic "road_marker"); null when nothing was published */
inline MarkerArray*& shim_markers()
{
    static MarkerArray* m = nullptr;
    return m;
}
inline void shim_capture(const std::string&, const MarkerArray& ma)
{
    delete shim_markers();
    shim_markers() = new MarkerArray(ma);
}
}   // namespace visualization_msgs
