/*
 * rmr_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the detect+locate hot path of zmsbruce/rm_radar
 * (SURVEY.md section 8a, Appendix A).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (rm_radar_amd/, include/) never does.
 *
 * Parity pin: checked against every golden vector the reference's own tests
 * hold for the path (test/detect/kernel_test.cu:71-205,
 * test/detect/detector_test.cpp:38-67, test/locate/locator_test.cpp:43-168);
 * see tests/test_oracle_kat.py.  Functions the reference's tests do not pin
 * (decode, NMS, restore, grouping, Locator::update, search tie-breaks) follow
 * the deterministic readings of SURVEY.md Appendix A and are "parity
 * unpinned" against the reference itself (it cannot be built here: needs
 * CUDA, TensorRT, OpenCV, PCL).
 *
 * All citations are file:line under /root/reference.
 */
#ifndef RMR_ORACLE_H
#define RMR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- detect: geometry ------------------------------------------------- */

/* src/detect/preparam.h:25-59 */
typedef struct {
    float width, height, ratio, dw, dh;
} orc_preparam;

/* src/detect/detection.h:25-68 : six f32, standard layout */
typedef struct {
    float x, y, width, height, label, confidence;
} orc_detection;

/* preparam.h:46-52 */
void orc_preparam_make(int in_w, int in_h, int out_w, int out_h, orc_preparam* p);

/* detector.cu:394-405 : resized size (truncated) and border offsets (rounded) */
void orc_letterbox_geometry(const orc_preparam* p, int* resized_w, int* resized_h,
                            int* top, int* bottom, int* left, int* right);

/* ---- detect: the six kernels, restated one-to-one --------------------- */

/* detector.cu:40-81 */
void orc_resize_u8(const uint8_t* src, uint8_t* dst, int channels, int src_w, int src_h,
                   int dst_w, int dst_h);
/* detector.cu:102-133 */
void orc_copy_make_border_u8(const uint8_t* src, uint8_t* dst, int channels, int src_w,
                             int src_h, int top, int bottom, int left, int right);
/* detector.cu:151-171 */
void orc_blob(const uint8_t* src, float* dst, int width, int height, int channels,
              float scale);
/* detector.cu:185-203 */
void orc_transpose(const float* src, float* dst, int rows, int cols);
/* detector.cu:219-251 ; src is [anchors][channels] (post-transpose) */
void orc_decode(const float* src, orc_detection* dst, int channels, int anchors, int classes);
/* detector.cu:271-293 */
float orc_iou(float x1, float y1, float w1, float h1, float x2, float y2, float w2, float h2);
/* detector.cu:315-360, "any-higher" deterministic reading:
 * every comparison sees PRE-NMS labels.  In place: suppressed rows get label=NaN. */
void orc_nms(orc_detection* dets, float nms_thresh, float score_thresh, int anchors);
/* detector.cpp:258-268 */
void orc_restore(orc_detection* d, const orc_preparam* p);

/* ---- detect: composed host sequencing --------------------------------- */

/* Detector::preprocess (detector.cu:380-421) for one image or one crop of it.
 * src is BGR u8 HWC with row stride src_stride bytes; (crop_x,crop_y,crop_w,crop_h)
 * selects the sub-image (the reference clones the crop first, detector.cpp:417-424).
 * Output: f32 NCHW RGB [3][out_h][out_w] scaled by 1/255.
 * Defined variant of Q2: the canvas is pre-filled with 128 and the resized image
 * is pasted at (left, top) with canvas stride out_w. */
void orc_preprocess(const uint8_t* src, int src_stride, int crop_x, int crop_y, int crop_w,
                    int crop_h, int out_w, int out_h, float* blob, orc_preparam* pp);

/* Detector::postprocess (detector.cu:522-582) for one image.
 * net_out is the network output [channels][anchors] (pre-transpose, channels=4+classes).
 * Writes survivors in ascending anchor order, restored; returns the count (<= cap kept). */
int orc_postprocess(const float* net_out, int channels, int anchors, int classes,
                    float nms_thresh, float conf_thresh, const orc_preparam* pp,
                    orc_detection* out, int cap);

/* ---- robot grouping ---------------------------------------------------- */

#define ORC_MAX_ARMORS 64

typedef struct {
    float rect[4];      /* Rect2f x,y,w,h (robot.cpp:44) */
    int has_label;      /* isDetected() */
    int label;
    float confidence;
    int n_armors;
    orc_detection armors[ORC_MAX_ARMORS];
    int has_location;
    float location[3];  /* metres (robot.h:93-95) */
    int track_state;    /* 0 none, 1 tentative, 2 confirmed, 3 deleted (track.h:26) -- set by the tracker stage */
} orc_robot;

/* robot.cpp:41-74 */
void orc_robot_set_detection(orc_robot* r, const orc_detection* car,
                             const orc_detection* armors, int n_armors);
/* Rect2f -> Rect conversion used by Robot::rect() (robot.h:111): OpenCV
 * saturate_cast<int>(float) = round-half-to-even [OpenCV behaviour]. */
void orc_rect_round(const float rect[4], int out[4]);
/* detector.cpp:324-349 : intersection / bounding-rect area */
float orc_compute_iou_bounding(const float a[4], const float b[4]);
/* detector.cpp:427-454 : undetected robots first (car order), then one per label */
int orc_group_robots(const orc_robot* in, int n, float iou_thresh, orc_robot* out);
/* detector.cpp:417-424 : crop rect from a car detection (float->int truncation) */
void orc_crop_rect(const orc_detection* car, int out[4]);

/* ---- locator ------------------------------------------------------------ */

typedef struct orc_locator orc_locator;

typedef struct {
    int image_width, image_height;
    float intrinsic[9];         /* row-major 3x3 */
    float lidar_to_camera[16];  /* row-major 4x4 */
    float world_to_camera[16];
    float zoom_factor;
    int queue_size;
    float min_depth_diff, max_depth_diff;
    float cluster_tolerance;
    int min_cluster_size, max_cluster_size;
    float max_distance;
} orc_locator_cfg;

/* locator.h:59-65 defaults */
void orc_locator_cfg_default(orc_locator_cfg* c);
/* locate.cpp:112-146 */
orc_locator* orc_locator_create(const orc_locator_cfg* c);
void orc_locator_destroy(orc_locator* l);
/* locate.cpp:158-220 ; xyz points with stride_bytes between points (16 for pcl::PointXYZ) */
void orc_locator_update(orc_locator* l, const float* xyz, int n, int stride_bytes);
/* locate.cpp:231-264 */
void orc_locator_cluster(orc_locator* l);
/* locate.cpp:276-311 ; rect = Rect2f of the robot.  Returns 1 and writes metres if located */
int orc_locator_search(const orc_locator* l, const float rect[4], float xyz_m[3]);
/* locate.cpp:337-350 */
void orc_locator_zoom(const orc_locator* l, const int rect[4], int out[4]);
/* locate.cpp:37-42 / 54-61 / 73-81 */
void orc_locator_lidar_to_world(const orc_locator* l, const float p[3], float out[3]);
void orc_locator_camera_to_lidar(const orc_locator* l, const float uvd[3], float out[3]);
void orc_locator_lidar_to_camera(const orc_locator* l, const float p[3], float out[3]);

/* state access for tests */
int orc_locator_width(const orc_locator* l);
int orc_locator_height(const orc_locator* l);
float* orc_locator_depth_image(orc_locator* l);
float* orc_locator_background_image(orc_locator* l);
float* orc_locator_diff_image(orc_locator* l);
int orc_locator_num_foreground(const orc_locator* l);
const float* orc_locator_foreground_xyz(const orc_locator* l);   /* n x 3 */
const int* orc_locator_foreground_pixel(const orc_locator* l);   /* n, linear index v*W+u */
const int* orc_locator_foreground_cluster(const orc_locator* l); /* n, cluster id or -1 */
int orc_locator_num_clusters(const orc_locator* l);
int orc_locator_cluster_size(const orc_locator* l, int id);

/* OpenCV Matx inverses as the reference calls them (locate.cpp:132-136)
 * [OpenCV behaviour]: 3x3 closed form, 4x4 LU with partial pivoting, f32. */
int orc_inv3x3(const float a[9], float out[9]);
int orc_inv4x4(const float a[16], float out[16]);

/* ---- conv cross-check (plain f32 direct convolution, NCHW) --------------- */
void orc_conv2d_nchw(const float* x, int n, int cin, int h, int w, const float* wt,
                     const float* bias, int cout, int kh, int kw, int stride, int pad,
                     int silu, float* y);

/* ---- whole-frame CPU pass over pre/post/locate used by bench cpu_baseline - */
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
