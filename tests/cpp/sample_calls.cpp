// sample_calls.cpp -- the reference application's call sites against include/radar/*.h with the REAL argument types:
// cv::Size, cv::Matx33f / cv::Matx44f, cv::Point3f, cv::Mat, pcl::PointCloud<pcl::PointXYZ>::Ptr, cv::Rect out of
// Robot::rect() (samples/sample_radar.h:57-127, samples/main.cpp:12-22,64-97).  This image has neither OpenCV nor PCL, so the
// TU is compiled with -I tests/cpp/stubs (test-only stand-ins carrying just the members those call sites touch); a
// maintainer of the reference compiles the same lines against the real libraries.  What runs here is the whole cycle
// for every frame -- update + cluster on one thread while detect runs on another, join, search, tracker -- and the robots
// are dumped as hex floats for tests/test_cpp_api.py to compare with the Python mirror on the same inputs.
//
// usage: sample_calls <car.rmrw> <armor.rmrw> <frames.bin> <clouds.bin> <car_conf> <armor_conf>
//   frames.bin: int32 n, w, h, then n * h * w * 3 BGR bytes
//   clouds.bin: int32 clouds, points, then clouds * points * 3 floats (mm, lidar frame; cloud 0 = the background cloud,
//               clouds 1 .. n the frames'); then 9 + 16 + 16 floats (intrinsic, lidar_to_camera, world_to_camera)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <memory>
#include <opencv2/opencv.hpp>
#include <optional>
#include <string_view>
#include <vector>

#include "radar/radar.h"

using namespace radar;

static_assert(std::is_same_v<radar::Size, cv::Size> && std::is_same_v<radar::Rect, cv::Rect> &&
                  std::is_same_v<radar::Matx44f, cv::Matx44f>,
              "with OpenCV on the include path the value types ARE the OpenCV types");

static constexpr int kClassNum = 12, kMaxBatchSize = 20, kOptBatchSize = 4;  // sample_radar.h:32-34

// what samples/frame.h:27-84 gives runOnce: optionals of an image, a cloud pointer, a time stamp
struct Frame {
    cv::Mat image_;
    pcl::PointCloud<pcl::PointXYZ>::Ptr cloud_;
    std::chrono::high_resolution_clock::time_point stamp_;
    std::optional<cv::Mat> image() const { return image_.empty() ? std::nullopt : std::make_optional(image_); }
    std::optional<pcl::PointCloud<pcl::PointXYZ>::Ptr> point_cloud() const { return cloud_ ? std::make_optional(cloud_) : std::nullopt; }
    std::optional<std::chrono::high_resolution_clock::time_point> timestamp() const { return stamp_; }
};

class Cycle {
   public:
    // sample_radar.h:57-69, argument for argument (the two thresholds are extra: the test's packs want their own)
    Cycle(std::string_view car_path, std::string_view armor_path, cv::Size image_size, const cv::Matx33f& intrinsic,
          const cv::Matx44f lidar_to_camera, const cv::Matx44f& world_to_camera, const cv::Point3f& lidar_noise,
          float car_conf, float armor_conf)
        : detector_(std::make_unique<RobotDetector>(car_path, armor_path, image_size, kClassNum, kMaxBatchSize, kOptBatchSize,
                                                    0.75f, 0.65f, car_conf, 0.65f, armor_conf)),
          locator_(std::make_unique<Locator>(image_size.width, image_size.height, intrinsic, lidar_to_camera, world_to_camera)),
          tracker_(std::make_unique<Tracker>(lidar_noise, kClassNum)) {}

    // sample_radar.h:94-97
    void updateBackgroundCloud(const pcl::PointCloud<pcl::PointXYZ>::Ptr& cloud) { locator_->update(cloud); }

    // sample_radar.h:106-127 without visualize()
    std::vector<Robot> runOnce(const Frame& frame) {
        auto future_locate = std::async(std::launch::async, [&] {
            locator_->update(frame.point_cloud().value_or(nullptr));
            locator_->cluster();
        });
        auto future_detect = std::async(std::launch::async, [&] { return detector_->detect(frame.image().value_or(cv::Mat())); });
        future_locate.get();
        auto robots = future_detect.get();
        locator_->search(robots);
        tracker_->update(robots, frame.timestamp().value_or(std::chrono::high_resolution_clock::now()));
        return robots;
    }

   private:
    std::unique_ptr<RobotDetector> detector_;
    std::unique_ptr<Locator> locator_;
    std::unique_ptr<Tracker> tracker_;
};

int main(int argc, char** argv) {
    if (argc != 7) return std::fprintf(stderr, "usage: see the header of sample_calls.cpp\n"), 2;
    const float car_conf = (float)std::atof(argv[5]), armor_conf = (float)std::atof(argv[6]);
    std::FILE* f = std::fopen(argv[3], "rb");
    if (!f) return std::perror(argv[3]), 2;
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2];
    std::vector<uint8_t> pix((size_t)n * w * h * 3);
    if (std::fread(pix.data(), 1, pix.size(), f) != pix.size()) return 2;
    std::fclose(f);
    std::vector<cv::Mat> images;
    for (int i = 0; i < n; ++i) images.emplace_back(h, w, CV_8UC3, pix.data() + (size_t)i * w * h * 3);

    f = std::fopen(argv[4], "rb");
    if (!f) return std::perror(argv[4]), 2;
    int32_t ch[2];
    if (std::fread(ch, 4, 2, f) != 2 || ch[0] != n + 1) return 2;
    std::vector<pcl::PointCloud<pcl::PointXYZ>::Ptr> clouds;
    std::vector<float> xyz((size_t)ch[1] * 3);
    for (int c = 0; c < ch[0]; ++c) {
        if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) return 2;
        pcl::PointCloud<pcl::PointXYZ>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZ>());
        for (int i = 0; i < ch[1]; ++i) cloud->push_back(pcl::PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        clouds.emplace_back(cloud);
    }
    float m[9 + 16 + 16];
    if (std::fread(m, 4, 41, f) != 41) return 2;
    std::fclose(f);
    const cv::Size image_size(w, h);
    const cv::Matx33f intrinsic(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]);
    const float *l = m + 9, *wc = m + 25;
    const cv::Matx44f lidar_to_camera(l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7], l[8], l[9], l[10], l[11], l[12], l[13], l[14], l[15]);
    const cv::Matx44f world_to_camera(wc[0], wc[1], wc[2], wc[3], wc[4], wc[5], wc[6], wc[7], wc[8], wc[9], wc[10], wc[11], wc[12],
                                      wc[13], wc[14], wc[15]);
    const cv::Point3f lidar_noise(0.4, 0.4, 0.4);  // main.cpp:22

    Cycle radar(argv[1], argv[2], image_size, intrinsic, lidar_to_camera, world_to_camera, lidar_noise, car_conf, armor_conf);
    // main.cpp:83-97
    const auto start_time = std::chrono::high_resolution_clock::time_point(std::chrono::seconds(1000));
    const auto duration = std::chrono::milliseconds(100);
    radar.updateBackgroundCloud(clouds[0]);
    for (size_t i = 0; i < images.size(); ++i) {
        const auto& image = images[i];
        const auto& cloud = clouds[i + 1];
        const auto timestamp = start_time + i * duration;
        Frame frame{image, cloud, std::chrono::time_point_cast<std::chrono::high_resolution_clock::duration>(timestamp)};
        const std::vector<Robot> robots = radar.runOnce(frame);
        std::printf("frame %zu robots %zu\n", i, robots.size());
        for (const auto& robot : robots) {
            const cv::Rect rect = robot.rect().value();                 // what visualize() hands cv::rectangle (:171-172)
            const cv::Rect2f rf = robot.rect2f().value();
            std::printf("robot %a %a %a %a int %d %d %d %d label %d conf %a state %d", rf.x, rf.y, rf.width, rf.height, rect.x, rect.y,
                        rect.width, rect.height, robot.label().value_or(-1), robot.confidence().value_or(0.0f),
                        robot.track_state().has_value() ? (int)robot.track_state().value() : 0);
            if (robot.location().has_value()) {
                const cv::Point3f p = robot.location().value();
                std::printf(" loc %a %a %a\n", p.x, p.y, p.z);
            } else {
                std::printf(" loc none\n");
            }
        }
    }
    // the null and the empty cloud print the reference's messages and clear the frame (locate.cpp:160-171)
    radar.updateBackgroundCloud(nullptr);
    radar.updateBackgroundCloud(pcl::PointCloud<pcl::PointXYZ>::Ptr(new pcl::PointCloud<pcl::PointXYZ>()));
    // Detector::detect<T> with the reference's two argument kinds (detector.h:117-134)
    {
        Detector det(argv[1], 1, image_size, (int)images.size(), std::nullopt, 0.65f, car_conf);
        const std::vector<Detection> one = det.detect(images[0]);
        const std::vector<std::vector<Detection>> many = det.detect(images);
        std::span<cv::Mat> sp(images);
        const std::vector<std::vector<Detection>> viaspan = det.detect(sp);
        // (one image alone and the same image inside a batch may run on different kernels -- another f32 summation order -- so a
        // detection at the threshold may exist in one and not in the other: both counts are printed, the Python mirror's are the bar)
        if (many.size() != images.size() || viaspan.size() != images.size() || viaspan[0].size() != many[0].size()) return 3;
        // ... but the divergence is BOUNDED: at most one detection crosses the threshold either way, and every
        // detection of the single call that is clear of the threshold has a partner in the batch call with the same label at
        // IoU >= 0.99 (the north-star bar between two runs of one network)
        const long diff = (long)one.size() - (long)many[0].size();
        if (diff > 1 || diff < -1) return 4;
        const auto iou = [](const Detection& a, const Detection& b) {
            const float x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
            const float x2 = std::min(a.x + a.width, b.x + b.width), y2 = std::min(a.y + a.height, b.y + b.height);
            const float inter = std::max(0.f, x2 - x1) * std::max(0.f, y2 - y1);
            const float uni = a.width * a.height + b.width * b.height - inter;
            return uni > 0.f ? inter / uni : 1.f;
        };
        for (const Detection& a : one) {
            if (a.confidence < car_conf + 0.02f) continue;
            bool found = false;
            for (const Detection& b : many[0]) found = found || (a.label == b.label && iou(a, b) >= 0.99f);
            if (!found) return 5;
        }
        std::printf("detect %zu %zu %zu\n", one.size(), many[0].size(), many.size());
    }
    std::puts("sample_calls ok");
    return 0;
}
