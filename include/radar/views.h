// views.h -- minimal value types standing in for the OpenCV / PCL types in the reference's
// signatures (cv::Size, cv::Rect, cv::Rect2f, cv::Point3f, cv::Matx33f, cv::Matx44f, cv::Mat,
// pcl::PointCloud<pcl::PointXYZ>).  When OpenCV / PCL headers are available the adapters at the
// bottom accept the real types, so reference call sites compile unchanged.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>

namespace radar {

struct Size {
    int width = 0, height = 0;
    constexpr Size() = default;
    constexpr Size(int w, int h) : width(w), height(h) {}
};
template <class T>
struct Rect_ {
    T x{}, y{}, width{}, height{};
    constexpr Rect_() = default;
    constexpr Rect_(T x_, T y_, T w_, T h_) : x(x_), y(y_), width(w_), height(h_) {}
};
using Rect = Rect_<int>;
using Rect2f = Rect_<float>;
struct Point3f {
    float x = 0, y = 0, z = 0;
};
using Matx33f = std::array<float, 9>;   // row-major
using Matx44f = std::array<float, 16>;  // row-major

// cv::Mat as the hot path sees it: BGR u8, HWC (borrowed for the call, never retained)
struct ImageView {
    const std::uint8_t* data = nullptr;
    int width = 0, height = 0;
    std::size_t stride = 0;  // bytes per row
    bool on_device = false;
    int channels() const { return 3; }
    Size size() const { return Size(width, height); }
};

// pcl::PointCloud<pcl::PointXYZ> as the hot path sees it: n points, stride_bytes apart (16 for
// pcl::PointXYZ), millimetres
struct CloudView {
    const float* xyz = nullptr;
    int size = 0;
    int stride_bytes = 16;
    bool on_device = false;
    bool empty() const { return size == 0; }
};

}  // namespace radar

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
namespace radar {
inline ImageView view(const cv::Mat& m) { return ImageView{m.data, m.cols, m.rows, m.step[0], false}; }
inline Matx33f matx(const cv::Matx33f& m) { Matx33f o; for (int i = 0; i < 9; ++i) o[i] = m.val[i]; return o; }
inline Matx44f matx(const cv::Matx44f& m) { Matx44f o; for (int i = 0; i < 16; ++i) o[i] = m.val[i]; return o; }
}  // namespace radar
#endif
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace radar {
inline CloudView view(const pcl::PointCloud<pcl::PointXYZ>::Ptr& c) {
    if (!c || c->empty()) return CloudView{};
    return CloudView{&c->points[0].x, (int)c->size(), (int)sizeof(pcl::PointXYZ), false};
}
}  // namespace radar
#endif
