// TEST-ONLY stand-in for <opencv2/core.hpp> (this image has no OpenCV): just the members the reference's call sites of
// radar::Detector / RobotDetector / Locator / Tracker / Robot touch (src/detect/detector.h:87-134,173-184,
// src/locate/locator.h:59-71, samples/sample_radar.h:57-127, samples/main.cpp:12-22).  Exists so that
// tests/cpp/sample_calls.cpp can prove that include/radar/*.h accepts the REAL argument types when OpenCV is on the
// include path.  It is never used to build any reference source and never ships with the library.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC3 16

namespace cv {

typedef unsigned char uchar;

template <typename T>
inline T saturate_cast(float v) { return (T)v; }
template <>
inline int saturate_cast<int>(float v) { return (int)std::lrintf(v); }  // round half to even, as cvRound

template <class T>
struct Size_ {
    T width{}, height{};
    Size_() = default;
    Size_(T w, T h) : width(w), height(h) {}
};
using Size = Size_<int>;

template <class T>
struct Rect_ {
    T x{}, y{}, width{}, height{};
    Rect_() = default;
    Rect_(T x_, T y_, T w_, T h_) : x(x_), y(y_), width(w_), height(h_) {}
    template <class U>
    operator Rect_<U>() const {
        return Rect_<U>(saturate_cast<U>(x), saturate_cast<U>(y), saturate_cast<U>(width), saturate_cast<U>(height));
    }
};
using Rect = Rect_<int>;
using Rect2f = Rect_<float>;

template <class T>
struct Point3_ {
    T x{}, y{}, z{};
    Point3_() = default;
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
using Point3f = Point3_<float>;

template <class T, int m, int n>
struct Matx {
    T val[m * n];
    Matx() { for (T& v : val) v = T(0); }
    template <class... A>
    Matx(A... a) : val{(T)a...} { static_assert(sizeof...(A) == m * n, "one value per element"); }
    T operator()(int i, int j) const { return val[i * n + j]; }
};
using Matx33f = Matx<float, 3, 3>;
using Matx44f = Matx<float, 4, 4>;

// reference-counted like cv::Mat: copies share the pixels, clone() copies them
class Mat {
   public:
    struct Step {
        std::size_t p[2] = {0, 0};
        std::size_t operator[](int i) const { return p[i]; }
    };
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    Step step;
    Mat() = default;
    Mat(int rows_, int cols_, int type) : rows(rows_), cols(cols_), own_(std::make_shared<std::vector<uchar>>((std::size_t)rows_ * cols_ * 3)) {
        (void)type;
        data = own_->data();
        step.p[0] = (std::size_t)cols * 3, step.p[1] = 3;
    }
    Mat(int rows_, int cols_, int type, void* external, std::size_t step_bytes = 0) : rows(rows_), cols(cols_), data((uchar*)external) {
        (void)type;
        step.p[0] = step_bytes ? step_bytes : (std::size_t)cols * 3, step.p[1] = 3;
    }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    int type() const { return CV_8UC3; }
    int channels() const { return 3; }
    Size size() const { return Size(cols, rows); }
    Mat clone() const {
        Mat m(rows, cols, CV_8UC3);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + r * m.step[0], data + r * step[0], (std::size_t)cols * 3);
        return m;
    }

   private:
    std::shared_ptr<std::vector<uchar>> own_;
};

}  // namespace cv
