// detection.h -- radar::Detection (src/detect/detection.h:25-68): six f32, layout-identical to
// rmr_detection so arrays cross the C-ABI without conversion.
#pragma once
#include <ostream>
#include <type_traits>

#include "../rmr.h"

namespace radar {

struct Detection {
    Detection() = default;
    Detection(float x, float y, float width, float height, float label, float confidence)
        : x{x}, y{y}, width{width}, height{height}, label{label}, confidence{confidence} {}
    friend std::ostream& operator<<(std::ostream& os, const Detection& d) {
        return os << "{ x: " << d.x << ", y: " << d.y << ", width: " << d.width << ", height: " << d.height
                  << ", label: " << d.label << ", confidence: " << d.confidence << " }";
    }
    float x = 0, y = 0, width = 0, height = 0, label = 0, confidence = 0;
};
static_assert(std::is_standard_layout_v<Detection> && sizeof(Detection) == sizeof(rmr_detection));

}  // namespace radar
