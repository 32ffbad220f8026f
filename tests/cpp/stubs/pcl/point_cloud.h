// TEST-ONLY stand-in for <pcl/point_cloud.h>: the members Locator::update's callers touch (samples/main.cpp:58-61,
// src/locate/locate.cpp:158-171).  Ptr is std::shared_ptr as in PCL >= 1.11.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
namespace pcl {
template <class PointT>
class PointCloud {
   public:
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    std::vector<PointT> points;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void push_back(const PointT& p) { points.push_back(p); }
    const PointT& operator[](std::size_t i) const { return points[i]; }
};
}  // namespace pcl
