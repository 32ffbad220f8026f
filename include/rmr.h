/*
 * rmr.h -- C-ABI of librmr.so: the MI355X-native detect+locate hot path that
 * replaces zmsbruce/rm_radar's Detector / RobotDetector / Locator back ends.
 *
 * The reference has no FFI: its boundary is the C++ class API of
 * src/detect/detector.h, src/locate/locator.h and src/robot/robot.h
 * (SURVEY.md 8b).  The C++20 wrappers in include/radar/ keep that API and call
 * ONLY the functions below.  Every entry point cites the reference interface
 * it stands in for (file:line under /root/reference).
 *
 * Conventions: plain pointers and sizes, opaque handles, caller-owned output
 * buffers, int status returns (0 = ok), no exceptions cross this boundary.
 * The library fails loudly (RMR_ERR_DEVICE) when no gfx950 device is usable;
 * there is no CPU fallback.
 */
#ifndef RMR_H
#define RMR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMR_ABI_VERSION 6

typedef int rmr_status;
enum {
    RMR_OK = 0,
    RMR_ERR_INVALID_ARGUMENT = 1, /* reference: std::invalid_argument (detector.cpp:80,181) */
    RMR_ERR_RUNTIME = 2,          /* reference: std::runtime_error (detector.cpp:184,199,205) */
    RMR_ERR_LOGIC = 3,            /* reference: std::logic_error (tensor.h:90) */
    RMR_ERR_DEVICE = 4,           /* reference: CUDA_CHECK failure -> abort (common.h:54-62) */
    RMR_ERR_CAPACITY = 5          /* a caller-owned or internal buffer was too small */
};

enum { RMR_MEM_HOST = 0, RMR_MEM_DEVICE = 1 };

/* last error message of the calling thread ("" when none) */
const char* rmr_last_error(void);
int rmr_abi_version(void);
/* number of usable gfx950 devices (0 => every create call returns RMR_ERR_DEVICE) */
int rmr_device_count(void);

/* ---------------------------------------------------------------- value types */

/* cv::Mat as the hot path sees it: BGR u8, HWC, row stride in bytes
 * (detector.cu:380-389).  mem says where `data` lives. */
typedef struct {
    const uint8_t* data;
    int width, height, stride;
    int mem;
} rmr_image;

/* radar::Detection (src/detect/detection.h:25-68): six f32, same layout */
typedef struct {
    float x, y, width, height, label, confidence;
} rmr_detection;

/* radar::detect::PreParam (src/detect/preparam.h:25-59) */
typedef struct {
    float width, height, ratio, dw, dh;
} rmr_preparam;

#define RMR_MAX_ARMORS 64

/* radar::Robot as filled by detect + locate (src/robot/robot.h:53-164) */
typedef struct {
    float rect[4];       /* Rect2f x,y,w,h */
    int has_label;       /* isDetected() */
    int label;           /* radar::Label (robot.h:32-45) */
    float confidence;
    int n_armors;
    rmr_detection armors[RMR_MAX_ARMORS];
    int has_location;    /* isLocated() */
    float location[3];   /* metres (robot.h:93-95) */
    int track_state;     /* RMR_TRACK_*: track_state() (robot.h:139-141), set by rmr_tracker_update */
} rmr_robot;

/* radar::TrackState (track.h:26) with "no track" as the zero value */
#define RMR_TRACK_NONE 0
#define RMR_TRACK_TENTATIVE 1
#define RMR_TRACK_CONFIRMED 2
#define RMR_TRACK_DELETED 3

/* compact fixed-size record used for the cross-GPU gather of the robot list */
typedef struct {
    float rect[4];
    float location[3];
    float confidence;
    int32_t label;      /* -1 when not detected */
    int32_t flags;      /* bit0 has_label, bit1 has_location, bit2 slot filled */
    int32_t stream_id;
    int32_t frame_id;
} rmr_robot_record;

/* ---------------------------------------------------------------- multi-GPU (one process per GPU)
 * No reference counterpart: the reference pins device 0 (detector.cpp:61) and runs one camera / LiDAR stream
 * (samples/sample_radar.h:106-127).  Streams shard over ranks (a stream's Locator state never migrates); the
 * only exchange of the path is one all-gather of the final robot list per batch of frames. */

typedef struct rmr_comm rmr_comm;
enum {
    RMR_TRANSPORT_RCCL = 0, /* ncclAllGather over xGMI; librccl.so is opened on first use            */
    RMR_TRANSPORT_FILE = 1  /* a shared directory (the id is its path): hosts / CI without GPUs      */
};
#define RMR_COMM_ID_BYTES 128

/* rank that owns stream s: s % world */
int rmr_stream_owner(int stream, int world);
/* the streams of `rank` in ascending order into out[cap]; returns their number */
int rmr_streams_of_rank(int n_streams, int rank, int world, int* out, int cap);
/* rank 0 creates the id; the host application hands it to the other ranks (env, file, MPI, socket ...) */
rmr_status rmr_comm_unique_id(int transport, char* id /* [RMR_COMM_ID_BYTES] */);
/* collective over all ranks (RCCL: ncclCommInitRank on `device`; FILE: rank 0 hands every rank the communicator's epoch in
 * a token handshake through the directory, so a second communicator on one path -- a restarted rank, a caller-supplied
 * directory -- never reads what an earlier one left behind; returns when every rank has joined, 120 s at most) */
rmr_status rmr_comm_create(int transport, int device, int rank, int world, const char* id, rmr_comm** out);
void rmr_comm_destroy(rmr_comm* comm);
/* every rank contributes n records (the same n on every rank); all[world][n] on every rank on return */
rmr_status rmr_comm_all_gather_records(rmr_comm* comm, const rmr_robot_record* mine, int n, rmr_robot_record* all);
/* robots[n_frames][cap] + counts[n_frames] -> out[n_frames][max_per_frame], zero padded; flags bit 2 marks a
 * filled slot */
rmr_status rmr_pack_robot_records(const rmr_robot* robots, const int* counts, int n_frames, int cap, int stream_id,
                                  int max_per_frame, rmr_robot_record* out);

/* ---------------------------------------------------------------- geometry (host) */

/* PreParam(cv::Size, cv::Size)  (preparam.h:46-52) */
rmr_status rmr_preparam_make(int in_w, int in_h, int out_w, int out_h, rmr_preparam* out);
/* resized size + border offsets as Detector::preprocess derives them (detector.cu:394-405) */
rmr_status rmr_letterbox_geometry(const rmr_preparam* pp, int* resized_w, int* resized_h,
                                  int* top, int* left);
/* Detector::restoreDetection (detector.cpp:258-268) */
rmr_status rmr_restore_detection(rmr_detection* d, const rmr_preparam* pp);

/* ---------------------------------------------------------------- unit kernels */

enum {
    RMR_FMT_U8_HWC = 0,   /* letterboxed canvas, u8, same channel order as the source   */
    RMR_FMT_F32_NCHW = 1  /* the reference's network input blob: f32 planar RGB * scale */
};

/* One fused launch standing in for resizeKernel + copyMakeBorderKernel + blobKernel
 * (detector.cu:40-81, 102-133, 151-171) with explicit geometry, so the reference's
 * kernel_test.cu vectors can be reproduced on the GPU.  For image i the sub-image
 * crops[4i..4i+3] = (x,y,w,h) (NULL = whole image) is resized to (resized_w,resized_h),
 * pasted at (left,top) into an out_w x out_h canvas filled with `fill`.
 * `out` is a HOST buffer of n * out_w*out_h*3 elements (u8 or f32 by `fmt`). */
rmr_status rmr_letterbox(int device, const rmr_image* imgs, const int* crops, int n,
                         int resized_w, int resized_h, int top, int left, int out_w, int out_h,
                         int fill, float scale, int fmt, void* out);

/* Detector::preprocess (detector.cu:380-421, 439-502): geometry from PreParam, fill 128,
 * scale 1/255, f32 NCHW RGB blob to a HOST buffer [n][3][out_h][out_w]; pp[n] out. */
rmr_status rmr_preprocess(int device, const rmr_image* imgs, const int* crops, int n, int out_w,
                          int out_h, float* blob, rmr_preparam* pp);

/* Detector::postprocess (detector.cu:522-582): net_out is HOST [n][channels][anchors] f32
 * (channels = 4 + classes).  Survivors per image in ascending anchor order, restored with
 * pp[i]; out has room for cap detections per image; counts[i] = number found (may exceed
 * cap: then only cap were written and RMR_ERR_CAPACITY is returned). */
rmr_status rmr_postprocess(int device, const float* net_out, int n, int channels, int anchors,
                           int classes, float nms_thresh, float conf_thresh,
                           const rmr_preparam* pp, rmr_detection* out, int* counts, int cap);

/* transposeKernel (detector.cu:185-203), kept only so its known-answer test has a target */
rmr_status rmr_transpose(int device, const float* src, float* dst, int rows, int cols);

/* One conv+bias(+SiLU)(+residual) layer through the MFMA implicit-GEMM engine.
 * Host f32 in/out (converted to the engine's f16 NHWC internally): x [n][h][w][cin],
 * w [cout][cin][kh][kw] (OIHW), bias [cout], residual/y [n][ho][wo][cout]. tile<0 = auto;
 * otherwise the kernel to test: 0..99 conv_igemm tile, 100..199 conv_dma tile (+ 1000 * split for
 * split-K), 200..299 conv_halo tile, 300..399 conv_ws variant, 400..499 conv_direct tile, 500 conv_stem,
 * 600..699 conv_ws_s2 variant, 700..799 conv_pw variant, 800..899 conv_t32 tile, 900..999 conv_t32f8 tile
 * (e4m3 operands: the input is quantised on the device, the weights on the host);
 * RMR_ERR_INVALID_ARGUMENT if it cannot run the layer. */
rmr_status rmr_conv2d(int device, const float* x, int n, int h, int w, int cin, const float* wt,
                      const float* bias, int cout, int kh, int kw, int stride, int pad, int silu,
                      const float* residual, float* y, int tile);

/* Version of the tuning-cache / kernel-plan file format this library reads and writes ('<pack>.tune', RMR_PLAN): the second
 * field of the file's header line.  It moves whenever the set of candidate kernels does, so a plan file of another version is
 * ignored (an engine cache of another TensorRT version, detector.cpp:74-99).  No GPU needed. */
int rmr_tune_file_version(void);

/* Host side of the fp8 weight packer: OCP e4m3fn bytes of x[n], round to nearest even, saturating at 448
 * (conv_t32f8.hip; the counterpart of choosing kFP16 / kINT8 at detector.cpp:208-231). */
rmr_status rmr_f32_to_e4m3(const float* x, int n, unsigned char* out);

/* The device quantiser of the fp8 plan (f16 -> e4m3, unit scale) on host data, n a multiple of 16: parity
 * hook, must equal rmr_f32_to_e4m3 of the f16-rounded values. */
rmr_status rmr_quant_e4m3(int device, const float* x, int n, unsigned char* out);

/* Development hook: times one conv layer (bias + SiLU, f16 in / f16 out, optional residual) on
 * device-resident pseudo-random data with HIP events; kernel = a tiled family id as in rmr_conv2d
 * (0..299, 800..899).  *ms_out = mean launch time over `reps` launches.  No reference counterpart
 * (TensorRT's builder times its tactics the same way, detector.cpp:208-231). */
rmr_status rmr_conv_bench(int device, int n, int h, int w, int cin, int cout, int k, int stride,
                          int residual, int kernel, int reps, float* ms_out);

/* ---------------------------------------------------------------- Detector */

typedef struct rmr_detector rmr_detector;

/* Arithmetic of the network (the reference sets one TensorRT builder flag, detector.cpp:226: kFP16):
 * F16 = f16 operands everywhere (f32 accumulate); FP8 = BASELINE configs[4]: the 3x3 / stride-1 convolutions
 * with >= 64 input channels (~70 % of the FLOPs) take OCP e4m3 weights (one scale per output channel) and e4m3
 * activations on the MX MFMA, everything else and every stored activation stay f16. */
enum { RMR_PRECISION_F16 = 0, RMR_PRECISION_FP8 = 1 };

/* Detector::Detector arguments (detector.h:87-93).  engine_path names this library's
 * weight pack (*.rmrw, see rm_radar_amd/weights.py) instead of a TensorRT engine. */
typedef struct {
    const char* engine_path;
    int classes;
    int image_width, image_height; /* cv::Size image_size: sizes the staging buffer */
    int max_batch_size;
    int opt_batch_size;            /* <=0 : nullopt */
    float nms_thresh, conf_thresh; /* 0.65 / 0.25 */
    int input_width, input_height; /* 640 x 640 */
    int input_channels;            /* 3 */
    int device;                    /* the reference hard-wires cudaSetDevice(0) (detector.cpp:61) */
    int precision;                 /* RMR_PRECISION_*: the builder flag of detector.cpp:226 (kFP16 there)  */
} rmr_detector_cfg;

void rmr_detector_cfg_default(rmr_detector_cfg* cfg);
rmr_status rmr_detector_create(const rmr_detector_cfg* cfg, rmr_detector** out);
void rmr_detector_destroy(rmr_detector* det);

/* Detector::detect<T> (detector.h:117-134) for n images (n=1: the cv::Mat overload).
 * crops may be NULL.  out: cap detections per image; counts[n]. */
rmr_status rmr_detector_detect(rmr_detector* det, const rmr_image* imgs, const int* crops, int n,
                               rmr_detection* out, int* counts, int cap);
/* preprocess + network only: the tensor TensorRT hands to postprocess,
 * HOST [n][4+classes][anchors] f32 (detector.cpp:129-130); pp[n] optional. */
rmr_status rmr_detector_infer(rmr_detector* det, const rmr_image* imgs, const int* crops, int n,
                              float* net_out, rmr_preparam* pp);
/* Parity hook: the output of backbone / neck stage `name` ("model.0" ... "model.21", Ultralytics module names)
 * for image `img` of the detector's last call, as HOST f32 [h][w][c]; dims = {h, w, c}; out may be NULL
 * (dimensions only).  RMR_ERR_INVALID_ARGUMENT for an unknown stage.  TensorRT offers the same through
 * marked network outputs (detector.cpp:187-231 builds the network from the ONNX graph). */
rmr_status rmr_detector_read_feature(rmr_detector* det, const char* name, int img, float* out, int* dims);
/* bytes of activation memory the detector holds on its GPU (sized for rmr_detector_chunk images per launch) */
double rmr_detector_arena_bytes(const rmr_detector* det);
int rmr_detector_chunk(const rmr_detector* det);
int rmr_detector_anchors(const rmr_detector* det);
int rmr_detector_channels(const rmr_detector* det);
/* algorithmic FLOPs of one 640x640 forward (2*MAC over all convs) */
double rmr_detector_flops_per_image(const rmr_detector* det);

/* ---------------------------------------------------------------- RobotDetector */

typedef struct rmr_robot_detector rmr_robot_detector;

/* RobotDetector::RobotDetector arguments (detector.h:173-180) */
typedef struct {
    const char* car_engine_path;
    const char* armor_engine_path;
    int image_width, image_height;
    int armor_classes;
    int max_cars, opt_cars;
    float iou_thresh;                         /* 0.75 */
    float car_nms_thresh, car_conf_thresh;    /* 0.65 / 0.25 */
    float armor_nms_thresh, armor_conf_thresh;/* 0.65 / 0.50 */
    int input_width, input_height, input_channels;
    int device;
    int max_frames;                           /* frames per detect_batch call (>=1) */
    int precision;                            /* RMR_PRECISION_* for both networks */
} rmr_robot_detector_cfg;

void rmr_robot_detector_cfg_default(rmr_robot_detector_cfg* cfg);
/* bytes of activation memory both networks hold on the GPU */
double rmr_robot_detector_arena_bytes(rmr_robot_detector* rd);
rmr_status rmr_robot_detector_create(const rmr_robot_detector_cfg* cfg, rmr_robot_detector** out);
void rmr_robot_detector_destroy(rmr_robot_detector* rd);
/* RobotDetector::detect (detector.cpp:413-455): one frame -> robots (cap entries) */
rmr_status rmr_robot_detector_detect(rmr_robot_detector* rd, const rmr_image* img, rmr_robot* out,
                                     int* n_out, int cap);
/* throughput mode: n_frames independent frames in one call (car stage batched over frames,
 * armor stage batched over all crops).  forced_crops != NULL injects forced_per_frame crop
 * rects (x,y,w,h) per frame in place of the car detections (synthetic-weight benches);
 * the car stage still runs in full.  out: cap robots per frame; n_out[n_frames]. */
rmr_status rmr_robot_detector_detect_batch(rmr_robot_detector* rd, const rmr_image* imgs,
                                           int n_frames, const int* forced_crops,
                                           int forced_per_frame, rmr_robot* out, int* n_out,
                                           int cap);
/* Parity hook: what the networks of the LAST detect / detect_batch / pipeline call handed to postprocess.  stage 0 = the
 * car network (one image per frame), 1 = the armor network (one image per non-empty car crop, frames in order, cars in
 * order).  Copies images [first, first + n) as HOST f32 [n][4 + classes][anchors] with their letterbox parameters
 * (pp may be NULL); n_last (may be NULL) receives the number of images of that call; n = 0 only asks for it. */
rmr_status rmr_robot_detector_read_heads(rmr_robot_detector* rd, int stage, int first, int n, float* out,
                                         rmr_preparam* pp, int* n_last);
/* host-side pieces, exposed for parity tests (robot.cpp:41-74, detector.cpp:324-349, 427-454) */
rmr_status rmr_robot_set_detection(rmr_robot* r, const rmr_detection* car,
                                   const rmr_detection* armors, int n_armors);
float rmr_compute_iou(const float rect_a[4], const float rect_b[4]);
rmr_status rmr_group_robots(const rmr_robot* in, int n, float iou_thresh, rmr_robot* out,
                            int* n_out);

/* ---------------------------------------------------------------- Locator */

typedef struct rmr_locator rmr_locator;

/* Locator::Locator arguments (locator.h:59-65) */
typedef struct {
    int image_width, image_height;
    float intrinsic[9];        /* cv::Matx33f row-major */
    float lidar_to_camera[16]; /* cv::Matx44f row-major */
    float world_to_camera[16];
    float zoom_factor;         /* 0.5 */
    int queue_size;            /* 3 */
    float min_depth_diff, max_depth_diff; /* 500 / 4000 mm */
    float cluster_tolerance;   /* 400 mm */
    int min_cluster_size, max_cluster_size; /* 8 / 1000 */
    float max_distance;        /* 29300 mm */
    int device;
    int max_points;            /* capacity of one cloud (default 262144) */
    int max_foreground;        /* capacity of the foreground list (default 32768) */
    int max_frames;            /* per-frame results kept for batched search (default 1) */
} rmr_locator_cfg;

void rmr_locator_cfg_default(rmr_locator_cfg* cfg);
rmr_status rmr_locator_create(const rmr_locator_cfg* cfg, rmr_locator** out);
void rmr_locator_destroy(rmr_locator* loc);
/* Locator::update (locate.cpp:158-220).  xyz: first of n points, stride_bytes apart
 * (16 for pcl::PointXYZ), millimetres.  NULL / n<=0 = the reference's null/empty cloud. */
rmr_status rmr_locator_update(rmr_locator* loc, const float* xyz, int n, int stride_bytes, int mem);
/* Locator::cluster (locate.cpp:231-264) */
rmr_status rmr_locator_cluster(rmr_locator* loc);
/* Locator::search(std::vector<Robot>&) (locate.cpp:323-326): fills location / has_location */
rmr_status rmr_locator_search(rmr_locator* loc, rmr_robot* robots, int n);
/* keep the current frame's cluster() result in slot `frame` (0..max_frames-1), and search
 * robots against a kept slot: lets a batch of frames be located after a batched detect */
rmr_status rmr_locator_keep(rmr_locator* loc, int frame);
rmr_status rmr_locator_search_kept(rmr_locator* loc, int frame, rmr_robot* robots, int n);
/* throughput mode: Locator::search over the kept frames 0 .. n_frames-1 in one pass (robots laid
 * out as rmr_robot_detector_detect_batch returns them: cap per frame, counts[f] valid) */
rmr_status rmr_locator_search_batch(rmr_locator* loc, rmr_robot* robots, const int* counts,
                                    int n_frames, int cap);
/* throughput mode: update + cluster + keep(f) for the n_frames consecutive frames of this stream (frame f =
 * clouds[f], n_points[f] points), with the same results -- the updates run in order (the background image and the
 * depth queue are the stream's history, locator.h:90-91), the cluster stage (locate.cpp:231-264) as ONE pass over
 * all frames.  Afterwards the current frame is the last one.  Needs max_frames >= n_frames. */
rmr_status rmr_locator_update_cluster_batch(rmr_locator* loc, const float* const* clouds, const int* n_points,
                                            int stride_bytes, int mem, int n_frames);

/* private members the reference's tests reach via `#define private public`
 * (test/locate/locator_test.cpp:6-13) */
enum { RMR_LOC_DEPTH = 0, RMR_LOC_BACKGROUND = 1, RMR_LOC_DIFF = 2 };
enum { RMR_XF_LIDAR_TO_WORLD = 0, RMR_XF_CAMERA_TO_LIDAR = 1, RMR_XF_LIDAR_TO_CAMERA = 2 };
int rmr_locator_width(const rmr_locator* loc);  /* image_width_zoomed_ */
int rmr_locator_height(const rmr_locator* loc);
rmr_status rmr_locator_read_image(rmr_locator* loc, int which, float* host_out);
rmr_status rmr_locator_write_image(rmr_locator* loc, int which, const float* host_in);
/* Snapshot of the temporal state the reference keeps only in memory (locator.h:90-91: background
 * image + queue of depth images): lets a stream restart, or move to another GPU, without
 * re-accumulating its background.  The blob is host memory of rmr_locator_state_bytes bytes. */
rmr_status rmr_locator_state_bytes(const rmr_locator* loc, size_t* bytes);
rmr_status rmr_locator_save_state(rmr_locator* loc, void* host_out, size_t cap);
rmr_status rmr_locator_load_state(rmr_locator* loc, const void* host_in, size_t bytes);
rmr_status rmr_locator_transform(const rmr_locator* loc, int which, const float in[3], float out[3]);
rmr_status rmr_locator_zoom(const rmr_locator* loc, const int rect[4], int out[4]);
/* cluster() products: foreground points in scan order (lidar frame, mm), their pixel
 * (v*W+u) and cluster id (-1 = unclustered); clusters_.size() */
rmr_status rmr_locator_foreground(rmr_locator* loc, float* xyz, int* pixel, int* cluster, int cap,
                                  int* n);
int rmr_locator_num_clusters(rmr_locator* loc);

/* ---------------------------------------------------------------- whole path, throughput mode
 * SampleRadar::runOnce (samples/sample_radar.h:106-127) over n_frames frames of ONE stream:
 * Locator update + cluster of every frame (in order) on a helper thread while the two-stage
 * detect runs, join, then one batched search.  clouds[f] / n_points[f]: the frame's points as
 * rmr_locator_update takes them; forced_crops as in rmr_robot_detector_detect_batch; the locator
 * must have been created with max_frames >= n_frames.  out: cap robots per frame, located. */
rmr_status rmr_pipeline_run_batch(rmr_robot_detector* rd, rmr_locator* loc, const rmr_image* imgs,
                                  const float* const* clouds, const int* n_points, int stride_bytes,
                                  int mem, int n_frames, const int* forced_crops, int forced_per_frame,
                                  rmr_robot* out, int* n_out, int cap);
/* The same for n_streams camera / LiDAR streams that share this GPU: one detector batch over all n_frames frames
 * (stream-major: stream s owns frames [s * n_frames / n_streams, ...)), one Locator per stream (locs[n_streams],
 * each created with max_frames >= n_frames / n_streams), their locate work on one helper thread and HIP stream
 * each.  State per stream in HBM: (8 + 4 (queue + 2)) bytes per zoomed pixel -- 2.9 MB at 640 x 640, so the
 * 288 GB of a GPU bound the number of streams only through the detector's batch. */
rmr_status rmr_pipeline_run_streams(rmr_robot_detector* rd, rmr_locator* const* locs, int n_streams,
                                    const rmr_image* imgs, const float* const* clouds, const int* n_points,
                                    int stride_bytes, int mem, int n_frames, const int* forced_crops,
                                    int forced_per_frame, rmr_robot* out, int* n_out, int cap);

/* ---------------------------------------------------------------- input staging, throughput mode
 * The reference uploads inside its cycle: every image is memcpy'd into a mapped pinned buffer that the resize kernel
 * reads across PCIe (src/detect/detector.cu:388-399, 455-470).  For batches this ring takes the upload out of the
 * cycle: step i + 1's frames and clouds are copied from the caller's page-locked buffers into device slot (i + 1) %
 * slots on a copy stream of their own while step i computes; the device addresses go into the rmr_image / cloud
 * tables of rmr_pipeline_run_batch (mem = RMR_MEM_DEVICE) after rmr_upload_wait. */
typedef struct rmr_upload rmr_upload;
/* page-locked host memory (hipHostMalloc): a copy from it is one DMA, from pageable memory it is staged by the driver */
rmr_status rmr_pinned_alloc(size_t bytes, void** out);   /* on the calling thread's current device */
rmr_status rmr_pinned_alloc_on(int device, size_t bytes, void** out);
void rmr_pinned_free(void* p);
rmr_status rmr_upload_create(int device, int slots, size_t bytes_per_slot, rmr_upload** out);
void rmr_upload_destroy(rmr_upload* up);
/* starts the copies of n host blocks into `slot` (packed at 256-byte boundaries; blocks adjacent in host memory
 * travel as one copy) and returns at once; dev_out[i] = the device address block i will have.  The slot must not be
 * in use by a running step; RMR_ERR_CAPACITY when the blocks do not fit bytes_per_slot. */
rmr_status rmr_upload_begin(rmr_upload* up, int slot, const void* const* src, const size_t* bytes, int n, void** dev_out);
/* blocks the calling thread until the slot's copies have landed (no-op when none is pending) */
rmr_status rmr_upload_wait(rmr_upload* up, int slot);

/* ---------------------------------------------------------------- per-kernel profile */

/* HIP-event timing of the library's own launches, on the streams they run on */
typedef struct {
    char name[64];
    long long launches;
    double total_ms;
    double flops;  /* algorithmic FLOPs summed over the launches */
    double bytes;  /* algorithmic HBM bytes summed over the launches */
} rmr_kernel_stat;

/* on: 0 off, 1 every launch, 2 only the launches that declare FLOPs (the convolution family: what
 * the roofline of bench.py needs, at a third of the event records) */
rmr_status rmr_profile_enable(int device, int on);
rmr_status rmr_profile_reset(int device);
/* resolves pending events (synchronises), writes up to cap entries, *n = entries available */
rmr_status rmr_profile_read(int device, rmr_kernel_stat* out, int cap, int* n);

/* ---------------------------------------------------------------- Tracker (host only)
 * The stage after detect + locate (src/track/): not part of the accelerated path, CPU code as
 * in the reference.  Matrices are row-major f32. */

/* KalmanFilter<N, M> / ExtendedKalmanFilter<N, M> (kalman_filter.h:77-296).  F, Q, H may be NULL
 * for an extended filter that receives them per call. */
typedef struct rmr_kalman rmr_kalman;
rmr_status rmr_kalman_create(int n, int m, const float* x0, const float* P0, const float* F,
                             const float* Q, const float* H, const float* R, rmr_kalman** out);
void rmr_kalman_destroy(rmr_kalman* kf);
rmr_status rmr_kalman_predict(rmr_kalman* kf);                    /* kalman_filter.h:116-121 */
rmr_status rmr_kalman_update(rmr_kalman* kf, const float* z);     /* kalman_filter.h:129-152 */
/* ExtendedKalmanFilter::predict / update with the transition matrix, process noise, predicted
 * measurement and observation Jacobian already evaluated by the caller (kalman_filter.h:221-248) */
rmr_status rmr_kalman_predict_ekf(rmr_kalman* kf, const float* F, const float* Q);
rmr_status rmr_kalman_update_ekf(rmr_kalman* kf, const float* z, const float* hx, const float* H);
rmr_status rmr_kalman_state(const rmr_kalman* kf, float* x, float* P); /* either may be NULL */

/* SingerEKF (singer.h:33-132): x0[9] = [x vx ax y vy ay z vz az], P0[81], R[9] */
typedef struct rmr_singer rmr_singer;
rmr_status rmr_singer_create(const float* x0, const float* P0, float max_a, float tau,
                             const float* R, rmr_singer** out);
void rmr_singer_destroy(rmr_singer* s);
rmr_status rmr_singer_predict(rmr_singer* s, float dt);
rmr_status rmr_singer_update(rmr_singer* s, const float* z);
rmr_status rmr_singer_state(const rmr_singer* s, float* x, float* P);

/* auction(value_matrix, max_iter) (auction.h:49-127): values[agents][tasks] -> assignment[agents],
 * -1 = not matched */
rmr_status rmr_auction(const float* values, int agents, int tasks, int max_iter, int* assignment);
/* Robot::feature(class_num) (robot.cpp:102-122) */
rmr_status rmr_robot_feature(const rmr_robot* r, int class_num, float* out);

/* Tracker::Tracker arguments (tracker.h:25-30) */
typedef struct {
    float observation_noise[3];          /* metres */
    int class_num;
    int init_thresh;                     /* 4 */
    int miss_thresh;                     /* 10 */
    float max_acceleration;              /* 2.0 */
    float acceleration_correlation_time; /* 1.0 */
    float distance_weight;               /* 0.40 */
    float feature_weight;                /* 0.60 */
    int max_iter;                        /* 100 */
    float distance_thresh;               /* 0.8 */
} rmr_tracker_cfg;

typedef struct {
    int id, state, label, init_count, miss_count;
    float location[3];
    float state_vector[9];
} rmr_track_info;

typedef struct rmr_tracker rmr_tracker;
void rmr_tracker_cfg_default(rmr_tracker_cfg* cfg);
rmr_status rmr_tracker_create(const rmr_tracker_cfg* cfg, rmr_tracker** out);
void rmr_tracker_destroy(rmr_tracker* t);
/* Tracker::update(robots, timestamp) (tracker.cpp:126-220); timestamp in nanoseconds.  Robots are
 * updated in place (label, location, track_state) as Robot::setTrack does (robot.cpp:81-94). */
rmr_status rmr_tracker_update(rmr_tracker* t, rmr_robot* robots, int n, int64_t timestamp_ns);
/* the live tracks, in the tracker's order (tests / diagnostics) */
rmr_status rmr_tracker_tracks(const rmr_tracker* t, rmr_track_info* out, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* RMR_H */
