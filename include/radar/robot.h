// robot.h -- radar::Robot / radar::Label (src/robot/robot.h:32-164) for the detect + locate path.
// Track-related members (setTrack, track_state, feature) belong to the out-of-scope tracker.
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

class Robot {
   public:
    Robot() = default;
    // robot.cpp:29-74
    Robot(const Detection& car, const std::vector<Detection>& armors) { setDetection(car, armors); }
    explicit Robot(const rmr_robot& r) { fromC(r); }

    bool isDetected() const noexcept { return armors_.has_value(); }
    bool isLocated() const noexcept { return location_.has_value(); }

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
        return r;
    }
    void fromC(const rmr_robot& r) {
        rect_ = Rect2f(r.rect[0], r.rect[1], r.rect[2], r.rect[3]);
        armors_.reset(), label_.reset(), confidence_.reset(), location_.reset();
        if (r.has_label) {
            label_ = r.label;
            confidence_ = r.confidence;
            armors_ = std::vector<Detection>(reinterpret_cast<const Detection*>(r.armors),
                                             reinterpret_cast<const Detection*>(r.armors) + r.n_armors);
        }
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
};

}  // namespace radar
