// views.h -- the value types of the reference's signatures (cv::Size, cv::Rect, cv::Rect2f, cv::Point3f, cv::Matx33f,
// cv::Matx44f, cv::Mat, pcl::PointCloud<pcl::PointXYZ>::Ptr) as this library sees them.
//
// With OpenCV on the include path (<opencv2/core.hpp>) radar::Size / Rect / Rect2f / Point3f / Matx33f / Matx44f ARE the
// OpenCV types, and with PCL (<pcl/point_cloud.h>) the classes take the PCL cloud pointer, so the reference's call sites
// -- RobotDetector(car, armor, cv::Size, ...), detector.detect(cv::Mat), Locator(w, h, cv::Matx33f, cv::Matx44f,
// cv::Matx44f), locator.update(cloud_ptr), cv::rectangle(image, robot.rect().value(), ...) (samples/sample_radar.h:57-127,
// 170-180) -- compile unchanged (tests/cpp/sample_calls.cpp compiles exactly that sequence).  Without them the
// dependency-free stand-ins below have the same member names.  ImageView / CloudView are this library's own additions:
// borrowed views that may also name DEVICE memory (frames / clouds already resident in HBM), which cv::Mat and
// pcl::PointCloud cannot express.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <type_traits>

#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define RADAR_HAVE_OPENCV 1
#endif
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#define RADAR_HAVE_PCL 1
#endif

namespace radar {

#ifdef RADAR_HAVE_OPENCV
using Size = cv::Size;
template <class T>
using Rect_ = cv::Rect_<T>;
using Rect = cv::Rect;
using Rect2f = cv::Rect2f;
using Point3f = cv::Point3f;
using Matx33f = cv::Matx33f;  // row-major .val[9]
using Matx44f = cv::Matx44f;  // row-major .val[16]
#else
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
#endif

namespace detail {
// element i (row-major) of a cv::Matx (.val) or of the std::array stand-in
template <class M>
inline float mat_at(const M& m, int i) {
    if constexpr (requires { m.val[0]; })
        return m.val[i];
    else
        return m[(std::size_t)i];
}
}  // namespace detail

// cv::Mat as the hot path sees it: BGR u8, HWC (borrowed for the call, never retained)
struct ImageView {
    const std::uint8_t* data = nullptr;
    int width = 0, height = 0;
    std::size_t stride = 0;  // bytes per row
    bool on_device = false;
    ImageView() = default;
    ImageView(const std::uint8_t* d, int w, int h, std::size_t s, bool dev = false)
        : data(d), width(w), height(h), stride(s), on_device(dev) {}
#ifdef RADAR_HAVE_OPENCV
    // a CV_8UC3 cv::Mat (what cv::imread returns, samples/main.cpp:36); an empty Mat is an empty view
    ImageView(const cv::Mat& m) : data(m.data), width(m.cols), height(m.rows), stride(m.empty() ? 0 : (std::size_t)m.step[0]) {}
#endif
    int channels() const { return 3; }
    Size size() const { return Size(width, height); }
    bool empty() const { return !data || width <= 0 || height <= 0; }
};

// pcl::PointCloud<pcl::PointXYZ> as the hot path sees it: n points, stride_bytes apart (16 for
// pcl::PointXYZ), millimetres
struct CloudView {
    const float* xyz = nullptr;
    int size = 0;
    int stride_bytes = 16;
    bool on_device = false;
    CloudView() = default;
    CloudView(const float* p, int n, int stride = 16, bool dev = false) : xyz(p), size(n), stride_bytes(stride), on_device(dev) {}
#ifdef RADAR_HAVE_PCL
    // a null pointer or an empty cloud is an empty view (Locator::update(Ptr) tells the two apart itself, to print the
    // reference's two messages, locate.cpp:160-171)
    CloudView(const pcl::PointCloud<pcl::PointXYZ>::Ptr& c)
        : xyz(c && !c->empty() ? &c->points[0].x : nullptr), size(c ? (int)c->size() : 0), stride_bytes((int)sizeof(pcl::PointXYZ)) {}
#endif
    bool empty() const { return size == 0; }
};

#ifdef RADAR_HAVE_OPENCV
inline ImageView view(const cv::Mat& m) { return ImageView(m); }
#endif
#ifdef RADAR_HAVE_PCL
inline CloudView view(const pcl::PointCloud<pcl::PointXYZ>::Ptr& c) { return CloudView(c); }
#endif

}  // namespace radar
