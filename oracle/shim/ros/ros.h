/*
 * Stand-in for <ros/ros.h> (roscpp is a third-party dependency that is not
 * part of the reference checkout and is not installed here).  It declares just
 * enough of the roscpp surface for the reference's hot-path translation units
 * to compile UNMODIFIED; publishing a message hands it to shim_capture(), which
 * the oracle harness (oracle/ref_harness.cpp) uses to read the clouds back.
 * TEST INFRASTRUCTURE ONLY -- no reference code in here.
 */
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#define ROS_INFO(...) ((void)0)

namespace ros {
struct Time {};
struct Duration {
    Duration(double = 0) {}
};
struct Subscriber {};

struct Publisher {
    std::string topic;
    template <class T>
    void publish(const T& m) const
    {
        shim_capture(topic, m);   /* found by ADL in the message's namespace */
    }
};

struct NodeHandle {
    template <class M, class C>
    Subscriber subscribe(const std::string&, int, void (C::*)(const M&), C*)
    {
        return Subscriber();
    }
    template <class T>
    Publisher advertise(const std::string& topic, int)
    {
        Publisher p;
        p.topic = topic;
        return p;
    }
};
}   // namespace ros

namespace std_msgs {
struct ColorRGBA {
    float r = 0, g = 0, b = 0, a = 0;
};
struct Header {
    std::string frame_id;
    ros::Time stamp;
};
}   // namespace std_msgs

namespace geometry_msgs {
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 0;
};
struct Vector3 {
    double x = 0, y = 0, z = 0;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
}   // namespace geometry_msgs
