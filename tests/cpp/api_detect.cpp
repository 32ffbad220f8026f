// api_detect.cpp -- the reference-shaped C++20 classes on the hot path, for real: radar::Detector::detect
// (single image and container overloads, src/detect/detector.h:117-134), radar::RobotDetector::detect
// (detector.h:184, detector.cpp:413-455) and radar::Locator::update / cluster / search
// (src/locate/locator.h:59-71) on inputs written by tests/test_cpp_api.py, results dumped as hex floats so
// that the test can compare them byte for byte with the ctypes path on the same packs, images and clouds.
//
// usage: api_detect <car.rmrw> <armor.rmrw> <frames.bin> <clouds.bin> <car_conf> <armor_conf>
//   frames.bin: int32 n, w, h, then n * h * w * 3 BGR bytes
//   clouds.bin: int32 frames, points, then frames * points * 3 floats (mm, lidar frame); then 9 + 16 + 16
//               floats (intrinsic, lidar_to_camera, world_to_camera), int32 n_rects, n_rects * 4 floats
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "radar/radar.h"

using namespace radar;

static void dump(const char* tag, const Detection& d) {
    std::printf("%s %a %a %a %a %a %a\n", tag, d.x, d.y, d.width, d.height, d.label, d.confidence);
}

int main(int argc, char** argv) {
    if (argc != 7) return std::fprintf(stderr, "usage: see the header of api_detect.cpp\n"), 2;
    const float car_conf = (float)std::atof(argv[5]), armor_conf = (float)std::atof(argv[6]);
    std::FILE* f = std::fopen(argv[3], "rb");
    if (!f) return std::perror(argv[3]), 2;
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2];
    std::vector<uint8_t> pix((size_t)n * w * h * 3);
    if (std::fread(pix.data(), 1, pix.size(), f) != pix.size()) return 2;
    std::fclose(f);
    std::vector<ImageView> frames;
    for (int i = 0; i < n; ++i) frames.push_back(ImageView{pix.data() + (size_t)i * w * h * 3, w, h, (size_t)w * 3, false});

    // ---- Detector: one image, then a container of images -------------------------------------------
    {
        Detector det(argv[1], 1, Size(w, h), n, std::nullopt, 0.65f, car_conf);
        const std::vector<Detection> one = det.detect(frames[0]);
        std::printf("detect_one %zu\n", one.size());
        for (const Detection& d : one) dump("d", d);
        const std::vector<std::vector<Detection>> many = det.detect(frames);
        std::printf("detect_many %zu\n", many.size());
        for (const auto& v : many) {
            std::printf("image %zu\n", v.size());
            for (const Detection& d : v) dump("d", d);
        }
    }
    // ---- RobotDetector: car stage, crops, armor stage, grouping ---------------------------------------
    std::vector<Robot> robots;
    {
        RobotDetector rd(argv[1], argv[2], Size(w, h), 12, 6, 4, 0.75f, 0.65f, car_conf, 0.65f, armor_conf);
        robots = rd.detect(frames[0]);
        std::printf("robots %zu\n", robots.size());
        for (const Robot& r : robots) {
            const Rect2f rc = *r.rect2f();
            const std::optional<std::vector<Detection>> armors = r.armors();  // by value, as robot.h:115
            std::printf("robot %a %a %a %a label %d conf %a armors %zu\n", rc.x, rc.y, rc.width, rc.height, r.label().value_or(-1),
                        r.confidence().value_or(0.f), armors ? armors->size() : (size_t)0);
            if (armors)
                for (const Detection& d : *armors) dump("a", d);
        }
    }
    // ---- Locator: update + cluster per frame, search the given rects after the last one --------------
    f = std::fopen(argv[4], "rb");
    if (!f) return std::perror(argv[4]), 2;
    int32_t ch[2];
    if (std::fread(ch, 4, 2, f) != 2) return 2;
    std::vector<float> pts((size_t)ch[0] * ch[1] * 3);
    if (std::fread(pts.data(), 4, pts.size(), f) != pts.size()) return 2;
    Matx33f K;
    Matx44f l2c, w2c;
    int32_t nr = 0;
    if (std::fread(K.data(), 4, 9, f) != 9 || std::fread(l2c.data(), 4, 16, f) != 16 || std::fread(w2c.data(), 4, 16, f) != 16 ||
        std::fread(&nr, 4, 1, f) != 1)
        return 2;
    std::vector<float> rects((size_t)nr * 4);
    if (std::fread(rects.data(), 4, rects.size(), f) != rects.size()) return 2;
    std::fclose(f);
    {
        Locator loc(w, h, K, l2c, w2c);
        for (int fr = 0; fr < ch[0]; ++fr) {
            loc.update(CloudView{pts.data() + (size_t)fr * ch[1] * 3, ch[1], 12, false});
            loc.cluster();
        }
        std::vector<Robot> rb(nr);
        for (int i = 0; i < nr; ++i) {
            rmr_robot r{};
            for (int k = 0; k < 4; ++k) r.rect[k] = rects[(size_t)i * 4 + k];
            rb[i].fromC(r);
        }
        loc.search(rb);
        std::printf("located %d\n", nr);
        for (const Robot& r : rb) {
            if (r.isLocated())
                std::printf("loc %a %a %a\n", r.location()->x, r.location()->y, r.location()->z);
            else
                std::printf("loc none\n");
        }
    }
    std::puts("api_detect ok");
    return 0;
}
