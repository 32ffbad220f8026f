// api_smoke.cpp -- the reference-shaped C++20 API (include/radar/) over librmr.so.
// Without a GPU: constructors must throw (no CPU fallback).  With a GPU: a tiny locate pass,
// mirroring test/locate/locator_test.cpp:121-168 (two pixel blobs -> rect is located).
#include <cstdio>
#include <random>
#include <stdexcept>
#include <vector>

#include "radar/radar.h"

using namespace radar;

int main() {
    const Matx33f eye3{1, 0, 0, 0, 1, 0, 0, 0, 1};
    const Matx44f eye4{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    detect::PreParam pp(Size(810, 1080), Size(640, 640));  // detector_test.cpp:38-41
    if (pp.dw != 80 || pp.dh != 0) return std::puts("FAIL preparam"), 1;
    Robot vote(Detection(10, 10, 100, 50, 0, 0.9f),
               {Detection(1, 1, 5, 5, 3, 0.6f), Detection(2, 2, 5, 5, 5, 0.5f), Detection(3, 3, 5, 5, 3, 0.3f)});
    if (!vote.isDetected() || vote.label().value() != 3) return std::puts("FAIL vote"), 1;

    // the tracker stage is host code: a robot seen five times becomes a confirmed track
    // (tracker.h:25-30 defaults: init_thresh 4)
    {
        Tracker tracker(Point3f{0.1f, 0.1f, 0.1f}, 12);
        const auto t0 = std::chrono::high_resolution_clock::time_point{} + std::chrono::seconds(1);
        std::vector<Robot> seen;
        for (int i = 0; i < 5; ++i) {
            seen.assign(1, vote);
            seen[0].setLocationMetres(Point3f{1.f + 0.01f * i, 2.f, 3.f});
            tracker.update(seen, t0 + std::chrono::milliseconds(100 * i));
        }
        if (!seen[0].isTracked() || *seen[0].track_state() != TrackState::Confirmed) return std::puts("FAIL tracker"), 1;
        if (seen[0].feature(12)[3] < 0.6f) return std::puts("FAIL feature"), 1;
    }

    try {
        Detector d("/nonexistent/car.rmrw", 1, Size(640, 640), 1);
        return std::puts("FAIL: missing engine accepted"), 1;
    } catch (const std::invalid_argument&) {
        if (rmr_device_count() == 0) return std::puts("FAIL: expected a device error without a GPU"), 1;
    } catch (const std::runtime_error&) {
        if (rmr_device_count() > 0) return std::puts("FAIL: expected invalid_argument for a missing engine"), 1;
    }
    if (rmr_device_count() == 0) {
        try {
            Locator l(640, 480, eye3, eye4, eye4);
            return std::puts("FAIL: Locator constructed without a GPU"), 1;
        } catch (const std::runtime_error&) {
        }
        std::puts("api_smoke ok (no GPU: constructors fail loudly)");
        return 0;
    }

    Locator loc(640, 480, eye3, eye4, eye4, 0.5f, 5, 0.05f, 5.0f, 100.f, 10, 1000, 20.0f);
    std::vector<float> diff(320 * 240, 0.f);
    std::mt19937 gen(0);
    std::normal_distribution<> x1(160, 10), y1(120, 10), x2(80, 10), y2(60, 10);
    std::uniform_real_distribution<> d1(5, 6), d2(1, 2);
    auto put = [&](double x, double y, double d) {
        const int xi = std::min(std::max((int)x, 0), 319), yi = std::min(std::max((int)y, 0), 239);
        diff[yi * 320 + xi] = (float)d;
    };
    for (int i = 0; i < 500; ++i) put(x1(gen), y1(gen), d1(gen)), put(x2(gen), y2(gen), d2(gen));
    if (rmr_locator_write_image(loc.handle(), RMR_LOC_DIFF, diff.data()) != RMR_OK) return std::puts(rmr_last_error()), 1;
    loc.cluster();
    if (rmr_locator_num_clusters(loc.handle()) != 2) return std::puts("FAIL clusters"), 1;
    std::vector<Robot> robots(1);
    rmr_robot r{};
    r.rect[0] = 140, r.rect[1] = 100, r.rect[2] = 40, r.rect[3] = 40;
    robots[0].fromC(r);
    loc.search(robots);
    if (!robots[0].isLocated()) return std::puts("FAIL search"), 1;
    loc.update(CloudView{});  // null cloud: message + early return (locate.cpp:160-165)
    // throughput form: update + cluster of a batch of frames (here one, the null cloud: nothing in the foreground)
    loc.updateClusterBatch({CloudView{}});
    if (rmr_locator_num_clusters(loc.handle()) != 0) return std::puts("FAIL updateClusterBatch"), 1;
    std::printf("api_smoke ok: located at [%f %f %f] m\n", robots[0].location()->x, robots[0].location()->y,
                robots[0].location()->z);
    return 0;
}
