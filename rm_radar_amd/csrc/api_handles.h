// api_handles.h -- the opaque handle types of include/rmr.h.
#pragma once
#include "detector.h"
#include "locator.h"

struct rmr_detector {
    rmr::Detector impl;
    explicit rmr_detector(const rmr_detector_cfg& c) : impl(c) {}
};
struct rmr_robot_detector {
    rmr::RobotDetector impl;
    explicit rmr_robot_detector(const rmr_robot_detector_cfg& c) : impl(c) {}
};
struct rmr_locator {
    rmr::Locator impl;
    explicit rmr_locator(const rmr_locator_cfg& c) : impl(c) {}
};
