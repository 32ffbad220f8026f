// detector.cpp -- Detector and RobotDetector back ends (src/detect/detector.cpp:48-161,
// 377-455; src/detect/detector.cu:380-582).  Per call: ONE letterbox launch for the whole batch,
// the network, ONE decode+NMS+restore launch, one small D2H -- instead of the reference's
// 3 + 3 launches, one stream and one 202 KB D2H per image.
#include "detector.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "postprocess.h"
#include "robot.h"

namespace rmr {

// ---- FrameStage ------------------------------------------------------------------------------------

// The helper thread of a large frame's staging copy.  Pieces are CLAIMED (one atomic counter), not dealt: the calling thread
// and the helper both take the next unclaimed piece, so a helper that wakes up late (a condition-variable wake-up is tens of
// microseconds, now and then milliseconds: p99 of configs[1] went to 5 ms when the pieces were dealt odd / even) costs
// nothing but its share -- the caller simply copies more.  done[i] is published with release order.
struct FrameStage::Helper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false, has_job = false;
    uint64_t gen = 0;                 // job number (under mu); the claim counter carries it in its upper half
    uint8_t* dst = nullptr;
    const uint8_t* src = nullptr;
    size_t bytes = 0, piece = 0, pieces = 0;
    std::atomic<uint64_t> next{0};    // (job number << 32) | next unclaimed piece
    std::vector<std::atomic<int>> done;
    Helper() : done(64) {
        th = std::thread([this] {
            for (;;) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || has_job; });
                if (stop) return;
                has_job = false;
                const uint64_t g = gen;
                lk.unlock();
                while (copy_next(g)) {
                }
            }
        });
    }
    ~Helper() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_one();
        th.join();
    }
    // claims and copies one piece of job g; false: none left, or the counter belongs to a later job (a helper that was away
    // for a whole frame must not take a piece of a job whose fields it has not read).  A thread that holds a claim on a piece
    // of job g reads stable fields: start() of job g + 1 only runs after the caller has seen EVERY piece of job g done.
    bool copy_next(uint64_t g) {
        uint64_t v = next.load(std::memory_order_acquire);
        for (;;) {
            if ((v >> 32) != g || (v & 0xffffffffu) >= pieces) return false;
            if (next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
        }
        const size_t i = (size_t)(v & 0xffffffffu), o = i * piece;
        std::memcpy(dst + o, src + o, std::min(piece, bytes - o));
        done[i].store(1, std::memory_order_release);
        return true;
    }
    uint64_t start(uint8_t* d, const uint8_t* s_, size_t n, size_t pc) {
        std::lock_guard<std::mutex> lk(mu);
        const size_t np = (n + pc - 1) / pc;
        if (done.size() < np) done = std::vector<std::atomic<int>>(np);
        for (size_t i = 0; i < np; ++i) done[i].store(0, std::memory_order_relaxed);
        dst = d, src = s_, bytes = n, piece = pc, pieces = np;
        ++gen;
        next.store(gen << 32, std::memory_order_release);
        has_job = true;
        cv.notify_one();
        return gen;
    }
};

FrameStage::~FrameStage() { delete helper_; }

const std::vector<FrameStage::Frame>& FrameStage::stage(hipStream_t s, const rmr_image* imgs, int n) {
    frames_.resize(n);
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        const rmr_image& im = imgs[i];
        if (!im.data || im.width <= 0 || im.height <= 0 || im.stride < im.width * 3)
            fail(RMR_ERR_INVALID_ARGUMENT, "image %d: bad data/size/stride", i);
        if (im.mem != RMR_MEM_DEVICE) need += ((size_t)im.stride * im.height + 255) & ~(size_t)255;
    }
    if (need > dev_.n) {
        RMR_HIP(hipStreamSynchronize(s));
        dev_.alloc(need);
        pin_.alloc(need);
    }
    // A large frame (the reference sample's 2592 x 2048 x 3 = 15.9 MB: 0.5 ms of memcpy + 0.3 ms of H2D, 30 % of a batch-1
    // frame when they run one after the other: profiles/r06_bench_config1_a.json) travels in pieces, each piece's H2D behind
    // the memcpy of the next, so the frame costs max(memcpy, H2D) + one piece: p50 2.67 -> 2.49 ms; the memcpy (0.5 ms for one
    // thread) is then what is left, so a helper thread copies every other piece.  (640 x 640 frames: one piece -- uploading them
    // in 4 or 8 pieces gave p50 1.919 / 1.94 against 1.911 ms, the extra copies cost what the overlap returns.)
    static const size_t piece = [] {
        const char* e = std::getenv("RMR_STAGE_PIECE_KB");
        return (size_t)(e ? std::max(64, std::atoi(e)) : 2048) << 10;
    }();
    size_t off = 0, sent = 0;
    bool any_host = false;
    for (int i = 0; i < n; ++i) {
        const rmr_image& im = imgs[i];
        Frame& f = frames_[i];
        f.width = im.width;
        f.height = im.height;
        f.stride = im.stride;
        if (im.mem == RMR_MEM_DEVICE) {
            f.dev = im.data;
            continue;
        }
        if (!any_host) {
            // the pinned buffer is reused across calls: the previous upload must have landed
            RMR_HIP(hipStreamSynchronize(s));
            any_host = true;
        }
        const size_t bytes = (size_t)im.stride * im.height;
        f.dev = dev_.p + off;
        if (bytes >= 4 * piece) {
            if (off > sent) RMR_HIP(hipMemcpyAsync(dev_.p + sent, pin_.p + sent, off - sent, hipMemcpyHostToDevice, s));
            if (!helper_) helper_ = new Helper();
            const uint64_t job = helper_->start(pin_.p + off, im.data, bytes, piece);
            const size_t np = helper_->pieces;
            size_t queued = 0;   // pieces whose H2D is enqueued: in order, each as soon as it has been copied (by either thread)
            while (queued < np) {
                const bool took = helper_->copy_next(job);
                while (queued < np && helper_->done[queued].load(std::memory_order_acquire)) {
                    const size_t o = queued * piece;
                    RMR_HIP(hipMemcpyAsync(dev_.p + off + o, pin_.p + off + o, std::min(piece, bytes - o), hipMemcpyHostToDevice, s));
                    ++queued;
                }
                if (!took && queued < np) __builtin_ia32_pause();   // every piece is claimed: the helper is finishing its last one
            }
            off += (bytes + 255) & ~(size_t)255;
            sent = off;
            continue;
        }
        std::memcpy(pin_.p + off, im.data, bytes);  // detector.cu:388: memcpy into pinned memory
        off += (bytes + 255) & ~(size_t)255;
    }
    if (off > sent) RMR_HIP(hipMemcpyAsync(dev_.p + sent, pin_.p + sent, off - sent, hipMemcpyHostToDevice, s));
    return frames_;
}

// ---- Detector -----------------------------------------------------------------------------------------

// Detector::Detector (detector.cpp:48-149)
Detector::Detector(const rmr_detector_cfg& cfg) : cfg_(cfg), ctx_(device_ctx(cfg.device)), stage_(ctx_) {
    if (!cfg.engine_path || !*cfg.engine_path) fail(RMR_ERR_INVALID_ARGUMENT, "Detector: engine_path is empty");
    if (cfg.classes <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "Detector: classes must be positive");
    if (cfg.max_batch_size < 1) fail(RMR_ERR_INVALID_ARGUMENT, "Detector: max_batch_size must be >= 1");
    if (cfg.input_channels != 3) fail(RMR_ERR_INVALID_ARGUMENT, "Detector: only 3-channel BGR input is supported");
    // highest stream priority: its own hardware queue, so the network is never stuck behind the
    // Locator's long tail of tiny launches (measured: car stage delayed 25 ms when they shared one)
    int prio_lo = 0, prio_hi = 0;
    RMR_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    RMR_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, prio_hi));
    RMR_HIP(hipEventCreateWithFlags(&io_done_, hipEventDisableTiming));
    if (cfg.precision != RMR_PRECISION_F16 && cfg.precision != RMR_PRECISION_FP8)
        fail(RMR_ERR_INVALID_ARGUMENT, "Detector: precision %d is not one of RMR_PRECISION_*", cfg.precision);
    net_ = std::make_unique<Yolov8>(ctx_, cfg.engine_path, cfg.classes, cfg.input_width, cfg.input_height,
                                    cfg.max_batch_size, cfg.precision == RMR_PRECISION_FP8);
    const int B = cfg.max_batch_size;
    io_dev_.alloc(pp_offset(B) + (size_t)B * sizeof(rmr_preparam));
    io_pin_.alloc(io_dev_.n);
    post_scratch_.alloc(postprocess_scratch_bytes(B, net_->anchors()));
    det_cap_ = net_->anchors();  // every anchor can survive, as in the reference (detector.cu:549)
    dets_dev_.alloc((size_t)B * det_cap_);
    // [B][kHeadRows] rows, then B counts: one contiguous block on both sides
    heads_dev_.alloc((size_t)B * kHeadRows * sizeof(rmr_detection) + (size_t)B * sizeof(int));
    heads_pin_.alloc(heads_dev_.n);
    counts_dev_.alloc(B);
}

Detector::~Detector() {
    if (stream_) {
        (void)hipStreamSynchronize(stream_);
        (void)hipStreamDestroy(stream_);
    }
    if (io_done_) (void)hipEventDestroy(io_done_);
}

// preprocess (detector.cu:380-502) + enqueueV3 (detector.h:122) + the device half of
// postprocess (detector.cu:522-555), all on one stream, no host synchronisation inside
void Detector::enqueue(std::vector<LetterboxDesc>& descs, bool post) {
    const int n = (int)descs.size();
    if (n > cfg_.max_batch_size)
        fail(RMR_ERR_CAPACITY, "Detector: batch of %d exceeds max_batch_size %d", n, cfg_.max_batch_size);
    ctx_.use();
    last_n_ = n;
    // the pinned descriptor block is reused: the previous call's copy of it must be done.  (Only that copy: a stream
    // synchronisation here also waited for the frame upload enqueued a moment ago -- 25 us of a batch-1 frame during which
    // the descriptors could have been written and the network launched behind it.)
    RMR_HIP(hipEventSynchronize(io_done_));
    for (int i = 0; i < n; ++i) {
        LetterboxDesc& d = descs[i];
        const rmr_preparam p = make_preparam(d.crop_w, d.crop_h, cfg_.input_width, cfg_.input_height);
        letterbox_geometry(p, d.rw, d.rh, d.top, d.left);
        descs_pin()[i] = d;
        pp_pin(n)[i] = p;
    }
    RMR_HIP(hipMemcpyAsync(io_dev_.p, io_pin_.p, pp_offset(n) + (size_t)n * sizeof(rmr_preparam), hipMemcpyHostToDevice, stream_));
    RMR_HIP(hipEventRecord(io_done_, stream_));
    // letterbox (fill 128, 1/255) + network; the first layer samples the frames itself where it can
    net_->forward(stream_, n, descs_dev(), 128, 1 / 255.f);
    if (!post) return;
    launch_postprocess(ctx_, stream_, net_->output(), n, net_->channels(), net_->anchors(), net_->nc(),
                       cfg_.nms_thresh, cfg_.conf_thresh, pp_dev(n), post_scratch_.p, dets_dev_.p,
                       counts_dev_.p, det_cap_);
    // D2H: the first kHeadRows rows of every image and the counts, gathered on the device into one block and fetched
    // with ONE contiguous copy (the reference copies all 8400 rows of every image, detector.cu:549-551); images with
    // more survivors are topped up after the sync in detect_staged()
    launch_gather_heads(stream_, dets_dev_.p, counts_dev_.p, det_cap_, kHeadRows, n, heads_dev_.p);
    RMR_HIP(hipMemcpyAsync(heads_pin_.p, heads_dev_.p, (size_t)n * kHeadRows * sizeof(rmr_detection) + (size_t)n * sizeof(int),
                           hipMemcpyDeviceToHost, stream_));
}

void Detector::detect_staged(std::vector<LetterboxDesc>& descs, std::vector<std::vector<rmr_detection>>& out,
                             const std::function<void()>& in_flight) {
    const int n = (int)descs.size();
    out.assign(n, {});
    if (n == 0) {        // Q10d: the reference would abort on an empty batch; return nothing --
        last_n_ = 0;     // and read_heads() must not hand out the previous call's heads as this call's
        if (in_flight) in_flight();
        return;
    }
    enqueue(descs, true);
    if (in_flight) {   // the GPU is busy with this batch: the caller's host work costs nothing here
        try {
            in_flight();
        } catch (...) {
            // the network is still sampling the staged frames: nothing may unwind (and let the next call's upload overwrite
            // them) before the stream has drained
            (void)hipStreamSynchronize(stream_);
            throw;
        }
    }
    RMR_HIP(hipStreamSynchronize(stream_));
    bool extra = false;
    const rmr_detection* rows = (const rmr_detection*)heads_pin_.p;
    const int* counts = (const int*)(heads_pin_.p + (size_t)n * kHeadRows * sizeof(rmr_detection));
    for (int i = 0; i < n; ++i) {
        const int c = counts[i];
        out[i].resize(c);
        std::copy(rows + (size_t)i * kHeadRows, rows + (size_t)i * kHeadRows + std::min(c, kHeadRows), out[i].begin());
        if (c > kHeadRows) {
            RMR_HIP(hipMemcpyAsync(out[i].data() + kHeadRows, dets_dev_.p + (size_t)i * det_cap_ + kHeadRows,
                                   (size_t)(c - kHeadRows) * sizeof(rmr_detection), hipMemcpyDeviceToHost, stream_));
            extra = true;
        }
    }
    if (extra) RMR_HIP(hipStreamSynchronize(stream_));
}

static void fill_descs(const std::vector<FrameStage::Frame>& fr, const int* crops, std::vector<LetterboxDesc>& descs) {
    const int n = (int)fr.size();
    descs.resize(n);
    for (int i = 0; i < n; ++i) {
        LetterboxDesc& d = descs[i];
        d.src = fr[i].dev;
        d.src_stride = fr[i].stride;
        d.crop_x = crops ? crops[4 * i + 0] : 0;
        d.crop_y = crops ? crops[4 * i + 1] : 0;
        d.crop_w = crops ? crops[4 * i + 2] : fr[i].width;
        d.crop_h = crops ? crops[4 * i + 3] : fr[i].height;
        if (d.crop_x < 0 || d.crop_y < 0 || d.crop_w <= 0 || d.crop_h <= 0 || d.crop_x + d.crop_w > fr[i].width ||
            d.crop_y + d.crop_h > fr[i].height)
            fail(RMR_ERR_INVALID_ARGUMENT, "image %d: crop outside the image", i);
        // the sampling kernels address a frame with 32-bit byte offsets
        if ((unsigned long long)fr[i].stride * (unsigned long long)fr[i].height >= 0xfffffff0ull)
            fail(RMR_ERR_CAPACITY, "image %d: frames of 4 GiB or more are not supported", i);
    }
}

void Detector::detect(const rmr_image* imgs, const int* crops, int n, rmr_detection* out, int* counts, int cap) {
    if (n < 0 || (n > 0 && (!imgs || !out || !counts)) || cap <= 0)
        fail(RMR_ERR_INVALID_ARGUMENT, "Detector::detect: bad arguments");
    ctx_.use();
    std::vector<LetterboxDesc> descs;
    fill_descs(stage_.stage(stream_, imgs, n), crops, descs);
    std::vector<std::vector<rmr_detection>> res;
    detect_staged(descs, res);
    bool over = false;
    for (int i = 0; i < n; ++i) {
        counts[i] = (int)res[i].size();
        const int m = std::min(counts[i], cap);
        std::copy(res[i].begin(), res[i].begin() + m, out + (size_t)i * cap);
        over |= counts[i] > cap;
    }
    if (over) fail(RMR_ERR_CAPACITY, "Detector::detect: more detections than the caller's cap %d", cap);
}

void Detector::infer(const rmr_image* imgs, const int* crops, int n, float* net_out, rmr_preparam* pp) {
    if (n <= 0 || !imgs || !net_out) fail(RMR_ERR_INVALID_ARGUMENT, "Detector::infer: bad arguments");
    ctx_.use();
    std::vector<LetterboxDesc> descs;
    fill_descs(stage_.stage(stream_, imgs, n), crops, descs);
    enqueue(descs, false);
    RMR_HIP(hipMemcpyAsync(net_out, net_->output(), (size_t)n * net_->channels() * net_->anchors() * sizeof(float),
                           hipMemcpyDeviceToHost, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
    if (pp)
        for (int i = 0; i < n; ++i) pp[i] = pp_pin(n)[i];
}

int Detector::read_heads(int first, int n, float* out, rmr_preparam* pp) {
    if (n == 0) return last_n_;
    if (first < 0 || n < 0 || first + n > last_n_ || !out)
        fail(RMR_ERR_INVALID_ARGUMENT, "read_heads: images [%d, %d) are not inside the last call's %d", first, first + n, last_n_);
    ctx_.use();
    const size_t per = (size_t)net_->channels() * net_->anchors();
    RMR_HIP(hipMemcpyAsync(out, net_->output() + (size_t)first * per, (size_t)n * per * sizeof(float), hipMemcpyDeviceToHost, stream_));
    RMR_HIP(hipStreamSynchronize(stream_));
    if (pp)
        for (int i = 0; i < n; ++i) pp[i] = pp_pin(last_n_)[first + i];
    return last_n_;
}

// ---- RobotDetector -------------------------------------------------------------------------------------

static rmr_detector_cfg sub_cfg(const rmr_robot_detector_cfg& c, const char* path, int classes, int max_batch,
                                float nms, float conf) {
    rmr_detector_cfg d{};
    d.engine_path = path;
    d.classes = classes;
    d.image_width = c.image_width;
    d.image_height = c.image_height;
    d.max_batch_size = max_batch;
    d.opt_batch_size = 0;
    d.nms_thresh = nms;
    d.conf_thresh = conf;
    d.input_width = c.input_width;
    d.input_height = c.input_height;
    d.input_channels = c.input_channels;
    d.device = c.device;
    d.precision = c.precision;
    return d;
}

// RobotDetector::RobotDetector (detector.cpp:377-397): the car detector has 1 class and
// batch 1 (per frame); the armor detector is batched over the cars
RobotDetector::RobotDetector(const rmr_robot_detector_cfg& cfg)
    : cfg_(cfg), stage_(device_ctx(cfg.device)) {
    if (cfg.max_cars < 1) fail(RMR_ERR_INVALID_ARGUMENT, "RobotDetector: max_cars must be >= 1");
    if (cfg_.max_frames < 1) cfg_.max_frames = 1;
    car_ = std::make_unique<Detector>(sub_cfg(cfg, cfg.car_engine_path, 1, cfg_.max_frames, cfg.car_nms_thresh, cfg.car_conf_thresh));
    armor_ = std::make_unique<Detector>(sub_cfg(cfg, cfg.armor_engine_path, cfg.armor_classes,
                                                 cfg_.max_frames * cfg.max_cars, cfg.armor_nms_thresh,
                                                 cfg.armor_conf_thresh));
}

// RobotDetector::detect (detector.cpp:413-455) for n_frames independent frames
void RobotDetector::detect_batch(const rmr_image* imgs, int n_frames, const int* forced_crops, int forced_per_frame,
                                 rmr_robot* out, int* n_out, int cap, const AfterCars& after_cars,
                                 int* car_index_out, const std::function<void()>& car_in_flight) {
    if (n_frames <= 0 || !imgs || !out || !n_out || cap <= 0)
        fail(RMR_ERR_INVALID_ARGUMENT, "RobotDetector::detect: bad arguments");
    if (n_frames > cfg_.max_frames)
        fail(RMR_ERR_CAPACITY, "RobotDetector::detect: %d frames exceed max_frames %d", n_frames, cfg_.max_frames);
    if (forced_crops && (forced_per_frame < 0 || forced_per_frame > cfg_.max_cars))
        fail(RMR_ERR_INVALID_ARGUMENT, "RobotDetector::detect: forced_per_frame must be in 0..max_cars");
    car_->ctx().use();
    const auto& frames = stage_.stage(car_->stream(), imgs, n_frames);

    // stage 1: cars on the full frames (detector.cpp:415)
    std::vector<LetterboxDesc> descs;
    fill_descs(frames, nullptr, descs);
    std::vector<std::vector<rmr_detection>> cars;
    struct StageTag {   // profile names of the launches enqueued meanwhile: "car|..." / "armor|..."
        explicit StageTag(int s) { Profiler::stage = s; }
        ~StageTag() { Profiler::stage = 0; }
    };
    {
        StageTag tag(1);
        car_->detect_staged(descs, cars, car_in_flight);
    }

    if (forced_crops) {
        // throughput benches with synthetic weights: the car stage ran in full, but the crops
        // handed to the armor stage are the caller's
        for (int f = 0; f < n_frames; ++f) {
            cars[f].clear();
            for (int k = 0; k < forced_per_frame; ++k) {
                const int* r = forced_crops + ((size_t)f * forced_per_frame + k) * 4;
                cars[f].push_back(rmr_detection{(float)r[0], (float)r[1], (float)r[2], (float)r[3], 0.f, 1.f});
            }
        }
    }

    for (int f = 0; f < n_frames; ++f)
        if ((int)cars[f].size() > cfg_.max_cars) cars[f].resize(cfg_.max_cars);  // Q10d
    // (after_cars runs below, once the armor stage is in flight)

    // stage 2: one armor batch over every car crop of every frame (detector.cpp:417-425)
    descs.clear();
    std::vector<int> slot_of;  // per (frame, car): index into the armor batch or -1
    for (int f = 0; f < n_frames; ++f) {
        for (const rmr_detection& c : cars[f]) {
            // cv::Rect(x, y, w, h) from floats truncates (detector.cpp:420-421)
            const int x = (int)c.x, y = (int)c.y, w = (int)c.width, h = (int)c.height;
            if (w <= 0 || h <= 0 || x < 0 || y < 0 || x + w > frames[f].width || y + h > frames[f].height) {
                slot_of.push_back(-1);  // Q10d: empty crop -> no armor batch entry
                continue;
            }
            LetterboxDesc d{};
            d.src = frames[f].dev;
            d.src_stride = frames[f].stride;
            d.crop_x = x;
            d.crop_y = y;
            d.crop_w = w;
            d.crop_h = h;
            slot_of.push_back((int)descs.size());
            descs.push_back(d);
        }
    }
    std::vector<std::vector<rmr_detection>> armors;
    {
        StageTag tag(2);
        armor_->detect_staged(descs, armors, after_cars ? std::function<void()>([&] { after_cars(cars); }) : std::function<void()>());
    }

    // Robot assembly + per-label de-duplication (detector.cpp:427-454)
    size_t k = 0;
    bool over = false;
    for (int f = 0; f < n_frames; ++f) {
        std::vector<rmr_robot> robots(cars[f].size());
        for (size_t i = 0; i < cars[f].size(); ++i, ++k) {
            const int s = slot_of[k];
            static const std::vector<rmr_detection> none;
            const auto& a = s >= 0 ? armors[s] : none;
            robot_set_detection(robots[i], cars[f][i], a.data(), (int)a.size());
            robots[i].track_state = (int)i + 1;  // tag: which car (grouping copies robots whole)
        }
        auto grouped = group_robots(robots.data(), (int)robots.size(), cfg_.iou_thresh);
        n_out[f] = (int)grouped.size();
        const int m = std::min(n_out[f], cap);
        for (int i = 0; i < m; ++i) {
            if (car_index_out) car_index_out[(size_t)f * cap + i] = grouped[i].track_state - 1;
            grouped[i].track_state = RMR_TRACK_NONE;
        }
        std::copy(grouped.begin(), grouped.begin() + m, out + (size_t)f * cap);
        over |= n_out[f] > cap;
    }
    if (over) fail(RMR_ERR_CAPACITY, "RobotDetector::detect: more robots than the caller's cap %d", cap);
}

}  // namespace rmr
