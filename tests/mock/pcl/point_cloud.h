// TEST-ONLY stand-in (tests/mock/README.md): pcl::PointCloud<T> as registration.hpp uses it (points, size(), width / height / is_dense).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
namespace pcl {
template <typename PointT>
class PointCloud
{
public:
    std::vector<PointT> points;
    uint32_t width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); }
    const PointT& operator[](size_t i) const { return points[i]; }
    PointT& operator[](size_t i) { return points[i]; }
};
}  // namespace pcl
