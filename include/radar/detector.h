// detector.h -- radar::Detector / radar::RobotDetector with the reference's signatures
// (src/detect/detector.h:84-190), calling only the C-ABI of librmr.so.
// Error behaviour as the reference: constructors throw (std::invalid_argument detector.cpp:80,181;
// std::runtime_error :184,199,205); hot-path methods are noexcept and abort on a device error
// (common.h:54-62).
#pragma once
#include <concepts>
#include <cstdio>
#include <cstdlib>
#include <optional>
#include <span>
#include <stdexcept>
#include <string>
#include <string_view>
#include <type_traits>
#include <vector>

#include "../rmr.h"
#include "detection.h"
#include "preparam.h"
#include "robot.h"
#include "views.h"

namespace radar {

namespace detail {
[[noreturn]] inline void throw_status(rmr_status s) {
    const std::string msg = rmr_last_error();
    if (s == RMR_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    if (s == RMR_ERR_LOGIC) throw std::logic_error(msg);
    throw std::runtime_error(msg);
}
inline void check_or_abort(rmr_status s) noexcept {
    if (s == RMR_OK) return;
    std::fprintf(stderr, "librmr error: %s\n", rmr_last_error());
    std::abort();
}
inline rmr_image to_c(const ImageView& v) {
    return rmr_image{v.data, v.width, v.height, (int)v.stride, v.on_device ? RMR_MEM_DEVICE : RMR_MEM_HOST};
}
}  // namespace detail

namespace detail {
// one image: this library's view, or (with OpenCV) the reference's cv::Mat
template <typename T>
inline constexpr bool is_image_v = std::is_same_v<std::decay_t<T>, ImageView>
#ifdef RADAR_HAVE_OPENCV
                                   || std::is_same_v<std::decay_t<T>, cv::Mat>
#endif
    ;
}  // namespace detail

// detector.h:70-77: an image (cv::Mat / ImageView) or a container of images (std::vector<cv::Mat>, std::span<cv::Mat>, ...)
template <typename T>
concept ImageOrImages = detail::is_image_v<T> || requires(T t) {
    typename std::decay_t<T>::value_type;
    { t.begin() } -> std::same_as<typename std::decay_t<T>::iterator>;
    { t.end() } -> std::same_as<typename std::decay_t<T>::iterator>;
    requires detail::is_image_v<typename std::decay_t<T>::value_type>;
};

class Detector {
   public:
    Detector() = delete;
    explicit Detector(std::string_view engine_path, int classes, Size image_size, int max_batch_size,
                      std::optional<int> opt_batch_size = std::nullopt, float nms_thresh = 0.65f,
                      float conf_thresh = 0.25f, int input_width = 640, int input_height = 640,
                      std::string_view input_name = "images", int input_channels = 3, int opt_level = 3,
                      int device = 0)
        : path_(engine_path) {
        (void)input_name, (void)opt_level;
        rmr_detector_cfg c;
        rmr_detector_cfg_default(&c);
        c.engine_path = path_.c_str();
        c.classes = classes;
        c.image_width = image_size.width, c.image_height = image_size.height;
        c.max_batch_size = max_batch_size;
        c.opt_batch_size = opt_batch_size.value_or(0);
        c.nms_thresh = nms_thresh, c.conf_thresh = conf_thresh;
        c.input_width = input_width, c.input_height = input_height, c.input_channels = input_channels;
        c.device = device;
        if (rmr_status s = rmr_detector_create(&c, &h_); s != RMR_OK) detail::throw_status(s);
    }
    ~Detector() { rmr_detector_destroy(h_); }
    Detector(const Detector&) = delete;
    Detector& operator=(const Detector&) = delete;

    // detector.h:117-134
    template <ImageOrImages T>
    auto detect(T&& input) noexcept {
        if constexpr (detail::is_image_v<T>) {   // std::vector<Detection>
            const ImageView one(input);
            return run(std::span<const ImageView>(&one, 1))[0];
        } else {                                 // std::vector<std::vector<Detection>>
            std::vector<ImageView> v;
            for (const auto& im : input) v.emplace_back(im);
            return run(v);
        }
    }

   private:
    std::vector<std::vector<Detection>> run(std::span<const ImageView> imgs) noexcept {
        const int n = (int)imgs.size();
        std::vector<std::vector<Detection>> out(n);
        if (n == 0) return out;
        std::vector<rmr_image> ci(n);
        for (int i = 0; i < n; ++i) ci[i] = detail::to_c(imgs[i]);
        const int cap = rmr_detector_anchors(h_);
        std::vector<rmr_detection> buf((size_t)n * cap);
        std::vector<int> counts(n);
        detail::check_or_abort(rmr_detector_detect(h_, ci.data(), nullptr, n, buf.data(), counts.data(), cap));
        for (int i = 0; i < n; ++i) {
            const auto* p = reinterpret_cast<const Detection*>(buf.data() + (size_t)i * cap);
            out[i].assign(p, p + counts[i]);
        }
        return out;
    }
    std::string path_;
    rmr_detector* h_ = nullptr;
};

class RobotDetector {
   public:
    RobotDetector() = delete;
    explicit RobotDetector(std::string_view car_engine_path, std::string_view armor_engine_path, Size image_size,
                           int armor_classes, int max_cars, int opt_cars, float iou_thresh = 0.75f,
                           float car_nms_thresh = 0.65f, float car_conf_thresh = 0.25f,
                           float armor_nms_thresh = 0.65f, float armor_conf_thresh = 0.50f, float input_width = 640,
                           float input_height = 640, std::string_view input_name = "images",
                           int input_channels = 3, int opt_level = 5, int device = 0)
        : car_(car_engine_path), armor_(armor_engine_path), max_cars_(max_cars) {
        (void)input_name, (void)opt_level;
        rmr_robot_detector_cfg c;
        rmr_robot_detector_cfg_default(&c);
        c.car_engine_path = car_.c_str(), c.armor_engine_path = armor_.c_str();
        c.image_width = image_size.width, c.image_height = image_size.height;
        c.armor_classes = armor_classes, c.max_cars = max_cars, c.opt_cars = opt_cars;
        c.iou_thresh = iou_thresh;
        c.car_nms_thresh = car_nms_thresh, c.car_conf_thresh = car_conf_thresh;
        c.armor_nms_thresh = armor_nms_thresh, c.armor_conf_thresh = armor_conf_thresh;
        c.input_width = (int)input_width, c.input_height = (int)input_height, c.input_channels = input_channels;
        c.device = device;
        if (rmr_status s = rmr_robot_detector_create(&c, &h_); s != RMR_OK) detail::throw_status(s);
    }
    ~RobotDetector() { rmr_robot_detector_destroy(h_); }
    RobotDetector(const RobotDetector&) = delete;
    RobotDetector& operator=(const RobotDetector&) = delete;

#ifdef RADAR_HAVE_OPENCV
    // detector.h:184, the reference's own signature
    std::vector<Robot> detect(const cv::Mat& image) { return detect(ImageView(image)); }
#endif
    // detector.cpp:413-455
    std::vector<Robot> detect(const ImageView& image) {
        const rmr_image ci = detail::to_c(image);
        std::vector<rmr_robot> buf(max_cars_);
        int n = 0;
        if (rmr_status s = rmr_robot_detector_detect(h_, &ci, buf.data(), &n, max_cars_); s != RMR_OK)
            detail::throw_status(s);
        std::vector<Robot> out;
        out.reserve(n);
        for (int i = 0; i < n; ++i) out.emplace_back(buf[i]);
        return out;
    }

   private:
    std::string car_, armor_;
    int max_cars_;
    rmr_robot_detector* h_ = nullptr;
};

}  // namespace radar
