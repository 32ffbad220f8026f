// detector.h -- Detector and RobotDetector back ends (src/detect/detector.h:84-190).
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include "common.h"
#include "preprocess.h"
#include "yolov8.h"

namespace rmr {

// Keeps device copies of host frames for the duration of one detect call.
class FrameStage {
   public:
    explicit FrameStage(DeviceCtx&) {}
    ~FrameStage();
    // returns, per image, a device pointer + geometry (device images pass through)
    struct Frame {
        const uint8_t* dev;
        int width, height, stride;
    };
    const std::vector<Frame>& stage(hipStream_t s, const rmr_image* imgs, int n);

   private:
    DevBuf<uint8_t> dev_;
    PinnedBuf<uint8_t> pin_;
    std::vector<Frame> frames_;
    // a large frame's memcpy into pinned memory is shared with one helper thread (started at the first such frame)
    struct Helper;
    Helper* helper_ = nullptr;
};

class Detector {
   public:
    explicit Detector(const rmr_detector_cfg& cfg);
    ~Detector();

    // Detector::detect<T> (detector.h:117-134)
    void detect(const rmr_image* imgs, const int* crops, int n, rmr_detection* out, int* counts, int cap);
    void infer(const rmr_image* imgs, const int* crops, int n, float* net_out, rmr_preparam* pp);

    // the same on frames already resident on the device; descs carry src/crop only
    // in_flight (optional): host work that needs nothing from this call, run after everything is enqueued and before the wait for
    // the results -- it travels under the network instead of in front of it
    void detect_staged(std::vector<LetterboxDesc>& descs, std::vector<std::vector<rmr_detection>>& out,
                       const std::function<void()>& in_flight = nullptr);

    // Parity hook: images [first, first + n) of the LAST call's network output ([4 + classes][anchors] f32 each, the
    // tensor handed to postprocess) and their letterbox parameters; returns the number of images of that call
    int read_heads(int first, int n, float* out, rmr_preparam* pp);

    Yolov8& net() { return *net_; }
    hipStream_t stream() { return stream_; }
    DeviceCtx& ctx() { return ctx_; }
    int max_batch() const { return cfg_.max_batch_size; }

   private:
    void enqueue(std::vector<LetterboxDesc>& descs, bool post);

    rmr_detector_cfg cfg_;
    DeviceCtx& ctx_;
    hipStream_t stream_ = nullptr;
    hipEvent_t io_done_ = nullptr;   // behind the H2D copy of the last call's descriptor block (the pinned side is reused)
    std::unique_ptr<Yolov8> net_;
    FrameStage stage_;
    static constexpr int kHeadRows = 64;  // rows per image fetched with the counts
    int det_cap_ = 0;
    int last_n_ = 0;  // images of the last enqueue()
    // descriptors and letterbox parameters of a call travel as ONE block ([n descs][n params], one H2D copy: a small
    // hipMemcpyAsync is a blit kernel of its own, ~4 us + a launch boundary in the batch-1 trace)
    DevBuf<unsigned char> io_dev_;
    PinnedBuf<unsigned char> io_pin_;
    static size_t pp_offset(int n) { return ((size_t)n * sizeof(LetterboxDesc) + 15) & ~(size_t)15; }
    LetterboxDesc* descs_dev() { return (LetterboxDesc*)io_dev_.p; }
    LetterboxDesc* descs_pin() { return (LetterboxDesc*)io_pin_.p; }
    rmr_preparam* pp_dev(int n) { return (rmr_preparam*)(io_dev_.p + pp_offset(n)); }
    rmr_preparam* pp_pin(int n) { return (rmr_preparam*)(io_pin_.p + pp_offset(n)); }
    DevBuf<uint8_t> post_scratch_;
    DevBuf<rmr_detection> dets_dev_;
    DevBuf<int> counts_dev_;
    DevBuf<uint8_t> heads_dev_;      // [B][kHeadRows] rows + B counts, gathered for one contiguous D2H copy
    PinnedBuf<uint8_t> heads_pin_;
};

class RobotDetector {
   public:
    explicit RobotDetector(const rmr_robot_detector_cfg& cfg);
    size_t arena_bytes() { return car_->net().arena_bytes() + armor_->net().arena_bytes(); }
    Detector& stage(int i) { return i == 0 ? *car_ : *armor_; }
    // RobotDetector::detect (detector.cpp:413-455), batched over frames
    // after_cars (optional) runs once stage 1 is known -- the car boxes per frame -- and the armor stage
    // has been ENQUEUED (it only needs the boxes, so its host time travels under the armor network);
    // car_index_out (optional, [n_frames][cap]) names the car each output robot came from.  Together
    // they let the caller start work that only needs the car boxes (Locator::search) while the armor
    // stage runs.  car_in_flight (optional) runs while the CAR stage is in flight: host work that needs
    // nothing from the detector (a single frame's Locator::update / cluster enqueue).
    using AfterCars = std::function<void(const std::vector<std::vector<rmr_detection>>&)>;
    void detect_batch(const rmr_image* imgs, int n_frames, const int* forced_crops, int forced_per_frame,
                      rmr_robot* out, int* n_out, int cap, const AfterCars& after_cars = nullptr,
                      int* car_index_out = nullptr, const std::function<void()>& car_in_flight = nullptr);

   private:
    rmr_robot_detector_cfg cfg_;
    std::unique_ptr<Detector> car_, armor_;
    FrameStage stage_;
};

}  // namespace rmr
