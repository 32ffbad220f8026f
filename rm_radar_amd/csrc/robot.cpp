// robot.cpp -- host-side Robot assembly of the two-stage detector
// (src/robot/robot.cpp:41-74, src/detect/detector.cpp:324-349, 427-454).
#include "robot.h"

#include <algorithm>
#include <cmath>
#include <map>

namespace rmr {

// Robot::setDetection (robot.cpp:41-74)
void robot_set_detection(rmr_robot& r, const rmr_detection& car, const rmr_detection* armors,
                         int n_armors) {
    std::memset(&r, 0, sizeof(r));
    r.rect[0] = car.x;
    r.rect[1] = car.y;
    r.rect[2] = car.width;
    r.rect[3] = car.height;
    r.label = -1;
    if (n_armors <= 0) return;
    // label, confidence (and through them the tracker feature) are voted by ALL armors, as in the
    // reference; only the stored list is bounded by the struct (RMR_MAX_ARMORS)

    // std::map<int, float>: per-label confidence sums, accumulated in armor order
    std::map<int, float> score;
    for (int i = 0; i < n_armors; ++i) score[(int)armors[i].label] += armors[i].confidence;
    // max_element over the ordered map: the first maximal sum in ascending label wins
    auto best = score.begin();
    for (auto it = score.begin(); it != score.end(); ++it)
        if (best->second < it->second) best = it;
    const int label = best->first;
    float confidence = best->second;
    int count = 0;
    for (int i = 0; i < n_armors; ++i)
        if (armors[i].label == (float)label) ++count;
    confidence /= (float)count;

    r.has_label = 1;
    r.label = label;
    r.confidence = confidence;
    r.n_armors = std::min(n_armors, (int)RMR_MAX_ARMORS);
    // armor boxes move from crop coordinates to image coordinates (robot.cpp:69-73)
    for (int i = 0; i < r.n_armors; ++i) {
        r.armors[i] = armors[i];
        r.armors[i].x += car.x;
        r.armors[i].y += car.y;
    }
}

// computeIoU (detector.cpp:324-349): intersection over the area of the BOUNDING rectangle
float compute_iou(const float a[4], const float b[4]) {
    float x1 = std::max(a[0], b[0]);
    float y1 = std::max(a[1], b[1]);
    float x2 = std::min(a[0] + a[2], b[0] + b[2]);
    float y2 = std::min(a[1] + a[3], b[1] + b[3]);
    float iw = 0, ih = 0;
    if (x1 < x2 && y1 < y2) {
        iw = x2 - x1;
        ih = y2 - y1;
    }
    x1 = std::min(a[0], b[0]);
    y1 = std::min(a[1], b[1]);
    x2 = std::max(a[0] + a[2], b[0] + b[2]);
    y2 = std::max(a[1] + a[3], b[1] + b[3]);
    const float inter = iw * ih;
    const float uni = (x2 - x1) * (y2 - y1);
    if (uni > 0) return inter / uni;
    return 0.0f;
}

// Robot::rect() hands out cv::Rect: Rect2f -> Rect rounds half to even (robot.h:111)
static void rounded_rect(const rmr_robot& r, float out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = (float)(int)lrintf(r.rect[i]);
}

// RobotDetector::detect tail (detector.cpp:427-454)
std::vector<rmr_robot> group_robots(const rmr_robot* in, int n, float iou_thresh) {
    std::vector<rmr_robot> robots;
    robots.reserve(n);
    std::map<int, rmr_robot> by_label;
    for (int i = 0; i < n; ++i) {
        const rmr_robot& robot = in[i];
        if (!robot.has_label) {
            robots.push_back(robot);
            continue;
        }
        auto it = by_label.find(robot.label);
        if (it == by_label.end()) {
            by_label.emplace(robot.label, robot);
            continue;
        }
        float ra[4], rb[4];
        rounded_rect(it->second, ra);
        rounded_rect(robot, rb);
        if (compute_iou(ra, rb) > iou_thresh) continue;
        if (it->second.confidence < robot.confidence) it->second = robot;
    }
    for (const auto& kv : by_label) robots.push_back(kv.second);
    return robots;
}

}  // namespace rmr
