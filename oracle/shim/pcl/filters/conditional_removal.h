/*
 * Stand-in for <pcl/filters/conditional_removal.h>; see ros/ros.h.
 * Semantics of pcl::ConditionalRemoval with keep_organized == false: the
 * output holds, in input order, the points for which the condition evaluates
 * to true (points with non-finite coordinates are dropped first).
 * TEST INFRASTRUCTURE ONLY.
 */
#pragma once
#include <cmath>
#include <pcl/point_cloud.h>

namespace pcl {
template <class T>
class ConditionalRemoval {
public:
    void setCondition(std::shared_ptr<ConditionBase<T>> c) { cond_ = c; }
    void setInputCloud(std::shared_ptr<PointCloud<T>> in) { in_ = in; }
    void filter(PointCloud<T>& out)
    {
        std::vector<T> kept;   /* `out` may alias the input cloud */
        kept.reserve(in_->points.size());
        for (const T& p : in_->points) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z))
                continue;
            if (cond_->evaluate(p))
                kept.push_back(p);
        }
        PCLHeader h = in_->header;
        out.points.swap(kept);
        out.header = h;
    }

private:
    std::shared_ptr<ConditionBase<T>> cond_;
    std::shared_ptr<PointCloud<T>> in_;
};
}   // namespace pcl
