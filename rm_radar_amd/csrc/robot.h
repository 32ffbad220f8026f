// robot.h -- host-side Robot assembly (src/robot/robot.cpp, src/detect/detector.cpp:324-454).
#pragma once
#include <vector>

#include "common.h"

namespace rmr {

void robot_set_detection(rmr_robot& r, const rmr_detection& car, const rmr_detection* armors,
                         int n_armors);
float compute_iou(const float a[4], const float b[4]);
std::vector<rmr_robot> group_robots(const rmr_robot* in, int n, float iou_thresh);

}  // namespace rmr
