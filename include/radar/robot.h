// robot.h -- radar::Robot / radar::Label / radar::TrackState (src/robot/robot.h:32-164, track.h:26).
// The track-related state is written by radar::Tracker::update (tracker.h), which applies
// Robot::setTrack (robot.cpp:81-94) inside the library.
#pragma once
#include <optional>
#include <ostream>
#include <vector>

#include "../rmr.h"
#include "detection.h"
#include "views.h"

namespace radar {

enum Label {
    BlueHero = 0, BlueEngineer = 1, BlueInfantryThree = 2, BlueInfantryFour = 3, BlueInfantryFive = 4,
    RedHero = 5, RedEngineer = 6, RedInfantryThree = 7, RedInfantryFour = 8, RedInfantryFive = 9,
    BlueSentry = 10, RedSentry = 11
};

enum class TrackState { Tentative = RMR_TRACK_TENTATIVE, Confirmed = RMR_TRACK_CONFIRMED, Deleted = RMR_TRACK_DELETED };

class Robot {
   public:
    Robot() = default;
    // robot.cpp:29-74
    Robot(const Detection& car, const std::vector<Detection>& armors) { setDetection(car, armors); }
    explicit Robot(const rmr_robot& r) { fromC(r); }

    bool isDetected() const noexcept { return armors_.has_value(); }
    bool isLocated() const noexcept { return location_.has_value(); }
    bool isTracked() const noexcept { return track_state_.has_value(); }
    std::optional<TrackState> track_state() const noexcept { return track_state_; }
    // robot.cpp:102-122
    std::vector<float> feature(int class_num) const {
        std::vector<float> f((size_t)class_num, 0.f);
        const rmr_robot c = toC();
        rmr_robot_feature(&c, class_num, f.data());
        return f;
    }

    void setDetection(const Detection& car, const std::vector<Detection>& armors) noexcept {
        rmr_robot r{};
        rmr_robot_set_detection(&r, reinterpret_cast<const rmr_detection*>(&car),
                                reinterpret_cast<const rmr_detection*>(armors.data()), (int)armors.size());
        fromC(r);
    }
    // robot.h:93-95: millimetres -> metres
    void setLocation(const Point3f& mm) noexcept {
        location_ = Point3f{(float)(mm.x * 1e-3), (float)(mm.y * 1e-3), (float)(mm.z * 1e-3)};
    }
    void setLocationMetres(const Point3f& m) noexcept { location_ = m; }
    std::optional<int> label() const noexcept { return label_; }
    // robot.h:111: Rect2f -> Rect rounds half to even (cv::saturate_cast<int>)
    std::optional<Rect> rect() const noexcept {
        if (!rect_) return std::nullopt;
        return Rect((int)__builtin_lrintf(rect_->x), (int)__builtin_lrintf(rect_->y),
                    (int)__builtin_lrintf(rect_->width), (int)__builtin_lrintf(rect_->height));
    }
    std::optional<Rect2f> rect2f() const noexcept { return rect_; }
    std::optional<float> confidence() const noexcept { return confidence_; }
    std::optional<std::vector<Detection>> armors() const noexcept { return armors_; }
    std::optional<Point3f> location() const noexcept { return location_; }

    rmr_robot toC() const {
        rmr_robot r{};
        if (rect_) r.rect[0] = rect_->x, r.rect[1] = rect_->y, r.rect[2] = rect_->width, r.rect[3] = rect_->height;
        r.has_label = label_.has_value();
        r.label = label_.value_or(-1);
        r.confidence = confidence_.value_or(0.f);
        if (armors_) {
            r.n_armors = (int)std::min<size_t>(armors_->size(), RMR_MAX_ARMORS);
            for (int i = 0; i < r.n_armors; ++i) r.armors[i] = reinterpret_cast<const rmr_detection&>((*armors_)[i]);
        }
        if (location_) r.has_location = 1, r.location[0] = location_->x, r.location[1] = location_->y, r.location[2] = location_->z;
        r.track_state = track_state_ ? (int)*track_state_ : RMR_TRACK_NONE;
        return r;
    }
    void fromC(const rmr_robot& r) {
        rect_ = Rect2f(r.rect[0], r.rect[1], r.rect[2], r.rect[3]);
        armors_.reset(), label_.reset(), confidence_.reset(), location_.reset(), track_state_.reset();
        if (r.has_label) label_ = r.label;  // from the armors, or from a track (robot.cpp:81-94)
        if (r.n_armors > 0) {               // isDetected()
            confidence_ = r.confidence;
            armors_ = std::vector<Detection>(reinterpret_cast<const Detection*>(r.armors),
                                             reinterpret_cast<const Detection*>(r.armors) + r.n_armors);
        }
        if (r.track_state != RMR_TRACK_NONE) track_state_ = (TrackState)r.track_state;
        if (r.has_location) location_ = Point3f{r.location[0], r.location[1], r.location[2]};
    }
    friend std::ostream& operator<<(std::ostream& os, const Robot& rb) {
        os << "Robot: { Label: ";
        rb.label_ ? os << *rb.label_ : os << "None";
        os << ", Confidence: ";
        rb.confidence_ ? os << *rb.confidence_ : os << "None";
        os << ", Location: ";
        rb.location_ ? os << "[" << rb.location_->x << ", " << rb.location_->y << ", " << rb.location_->z << "]" : os << "None";
        return os << " }";
    }

   private:
    std::optional<std::vector<Detection>> armors_;
    std::optional<Point3f> location_;
    std::optional<Rect2f> rect_;
    std::optional<int> label_;
    std::optional<float> confidence_;
    std::optional<TrackState> track_state_;
};

}  // namespace radar
