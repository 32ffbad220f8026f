// preparam.h -- radar::detect::PreParam (src/detect/preparam.h:25-59)
#pragma once
#include "../rmr.h"
#include "views.h"

namespace radar::detect {

struct PreParam {
    PreParam() = default;
    PreParam(float width, float height, float ratio, float dw, float dh)
        : width{width}, height{height}, ratio{ratio}, dw{dw}, dh{dh} {}
    PreParam(Size input, Size output) {
        rmr_preparam p{};
        rmr_preparam_make(input.width, input.height, output.width, output.height, &p);
        width = p.width, height = p.height, ratio = p.ratio, dw = p.dw, dh = p.dh;
    }
    float width = 0, height = 0, ratio = 1, dw = 0, dh = 0;
};

}  // namespace radar::detect
