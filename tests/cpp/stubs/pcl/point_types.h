// TEST-ONLY stand-in for <pcl/point_types.h> (this image has no PCL): pcl::PointXYZ with PCL's 16-byte SSE-padded
// layout (x, y, z, 1.0f).  See tests/cpp/stubs/opencv2/core.hpp for why these exist.
#pragma once
namespace pcl {
struct alignas(16) PointXYZ {
    float x = 0, y = 0, z = 0, data_w = 1.f;
    PointXYZ() = default;
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ is 16 bytes");
}  // namespace pcl
