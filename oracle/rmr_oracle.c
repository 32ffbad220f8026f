/*
 * rmr_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see rmr_oracle.h).
 *
 * Restates, function by function, the reference's detect+locate hot path.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Build with -ffp-contract=off: the f32 operation order
 * written here IS the specification the HIP path is compared against.
 */
#include "rmr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================= */
/* detect: geometry                                                          */
/* ======================================================================= */

/* src/detect/preparam.h:46-52 */
void orc_preparam_make(int in_w, int in_h, int out_w, int out_h, orc_preparam* p) {
    float height = (float)in_h;
    float width = (float)in_w;
    float rh = (float)out_h / height; /* int / float -> float */
    float rw = (float)out_w / width;
    float ratio = 1.0f / (rh < rw ? rh : rw); /* 1 / std::min(...) */
    p->height = height;
    p->width = width;
    p->ratio = ratio;
    p->dw = ((float)out_w - roundf(width / ratio)) * 0.5f;
    p->dh = ((float)out_h - roundf(height / ratio)) * 0.5f;
}

/* src/detect/detector.cu:394-405.  The float padding_width/height are passed to
 * int kernel parameters (truncation, Q2); the border offsets are
 * round(double(d) -/+ 0.1). */
void orc_letterbox_geometry(const orc_preparam* p, int* resized_w, int* resized_h,
                            int* top, int* bottom, int* left, int* right) {
    float padding_width = p->width / p->ratio;
    float padding_height = p->height / p->ratio;
    *resized_w = (int)padding_width;
    *resized_h = (int)padding_height;
    *top = (int)round((double)p->dh - 0.1);
    *bottom = (int)round((double)p->dh + 0.1);
    *left = (int)round((double)p->dw - 0.1);
    *right = (int)round((double)p->dw + 0.1);
}

/* ======================================================================= */
/* detect: kernels                                                           */
/* ======================================================================= */

/* One bilinear sample, src/detect/detector.cu:53-79 (Q1): top-left aligned
 * source coordinate, f32 weights, f32 accumulate in the order tl+tr+bl+br,
 * (unsigned char) truncation. */
static inline void resize_pixel(const uint8_t* src, int src_step, int channels, int src_w,
                                int src_h, int dst_w, int dst_h, int dst_x, int dst_y,
                                uint8_t* out) {
    float src_y = (float)dst_y * (float)src_h / (float)dst_h;
    float src_x = (float)dst_x * (float)src_w / (float)dst_w;
    int y_lo = (int)src_y;
    int y_hi = y_lo + 1 < src_h - 1 ? y_lo + 1 : src_h - 1;
    int x_lo = (int)src_x;
    int x_hi = x_lo + 1 < src_w - 1 ? x_lo + 1 : src_w - 1;
    float ly = src_y - (float)y_lo;
    float lx = src_x - (float)x_lo;
    float hy = 1.f - ly;
    float hx = 1.f - lx;
    for (int c = 0; c < channels; ++c) {
        float tl = (float)src[y_lo * src_step + x_lo * channels + c] * hy * hx;
        float tr = (float)src[y_lo * src_step + x_hi * channels + c] * hy * lx;
        float bl = (float)src[y_hi * src_step + x_lo * channels + c] * ly * hx;
        float br = (float)src[y_hi * src_step + x_hi * channels + c] * ly * lx;
        float value = tl + tr + bl + br;
        out[c] = (uint8_t)value;
    }
}

/* src/detect/detector.cu:40-81 */
void orc_resize_u8(const uint8_t* src, uint8_t* dst, int channels, int src_w, int src_h,
                   int dst_w, int dst_h) {
    int src_step = src_w * channels;
    int dst_step = dst_w * channels;
    for (int y = 0; y < dst_h; ++y)
        for (int x = 0; x < dst_w; ++x)
            resize_pixel(src, src_step, channels, src_w, src_h, dst_w, dst_h, x, y,
                         dst + y * dst_step + x * channels);
}

/* src/detect/detector.cu:102-133 */
void orc_copy_make_border_u8(const uint8_t* src, uint8_t* dst, int channels, int src_w,
                             int src_h, int top, int bottom, int left, int right) {
    int dst_w = src_w + left + right;
    int dst_h = src_h + top + bottom;
    int src_step = src_w * channels;
    int dst_step = dst_w * channels;
    for (int dy = 0; dy < dst_h; ++dy) {
        for (int dx = 0; dx < dst_w; ++dx) {
            int sy = dy - top, sx = dx - left;
            for (int c = 0; c < channels; ++c) {
                if (sy >= 0 && sy < src_h && sx >= 0 && sx < src_w)
                    dst[dy * dst_step + dx * channels + c] = src[sy * src_step + sx * channels + c];
                else
                    dst[dy * dst_step + dx * channels + c] = 128;
            }
        }
    }
}

/* src/detect/detector.cu:151-171 : HWC BGR u8 -> CHW RGB f32 * scale (Q4) */
void orc_blob(const uint8_t* src, float* dst, int width, int height, int channels,
              float scale) {
    (void)channels; /* the reference hard-codes *3 indexing */
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            dst[y * width + x + width * height * 0] = (float)src[(y * width + x) * 3 + 2] * scale;
            dst[y * width + x + width * height * 1] = (float)src[(y * width + x) * 3 + 1] * scale;
            dst[y * width + x + width * height * 2] = (float)src[(y * width + x) * 3 + 0] * scale;
        }
    }
}

/* src/detect/detector.cu:185-203 : [rows][cols] -> [cols][rows] */
void orc_transpose(const float* src, float* dst, int rows, int cols) {
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) dst[c * rows + r] = src[r * cols + c];
}

/* src/detect/detector.cu:219-251 (Q7): strict '>' argmax => first max; x,y via a
 * double intermediate (0.5 is a double literal); w,h unchanged. */
void orc_decode(const float* src, orc_detection* dst, int channels, int anchors, int classes) {
    for (int row = 0; row < anchors; ++row) {
        const float* bbox = src + (size_t)channels * row;
        const float* score = bbox + 4;
        int best = 0;
        for (int j = 0; j < classes; ++j)
            if (score[j] > score[best]) best = j;
        double xd = (double)bbox[0] - 0.5 * (double)bbox[2];
        double yd = (double)bbox[1] - 0.5 * (double)bbox[3];
        dst[row].x = (float)fmax(xd, 0.0);
        dst[row].y = (float)fmax(yd, 0.0);
        dst[row].width = bbox[2];
        dst[row].height = bbox[3];
        dst[row].label = (float)best;
        dst[row].confidence = score[best];
    }
}

/* src/detect/detector.cu:271-293 (Q8) */
float orc_iou(float x1, float y1, float w1, float h1, float x2, float y2, float w2, float h2) {
    /* CUDA max/min on floats are fmaxf/fminf (a NaN operand yields the other one) */
    float x_left = fmaxf(x1, x2);
    float y_top = fmaxf(y1, y2);
    float x_right = fminf(x1 + w1, x2 + w2);
    float y_bottom = fminf(y1 + h1, y2 + h2);
    if (x_right < x_left || y_bottom < y_top) return 0.0f;
    float iw = x_right - x_left;
    float ih = y_bottom - y_top;
    float inter = iw * ih;
    float area1 = w1 * h1;
    float area2 = w2 * h2;
    float uni = area1 + area2 - inter;
    return inter / uni;
}

/* src/detect/detector.cu:315-360 with the "any-higher" reading (Q9): a row is
 * dropped when conf < score_thresh, or when ANY same-label row with strictly
 * higher confidence overlaps it by IoU > nms_thresh -- judged on pre-NMS labels. */
void orc_nms(orc_detection* dets, float nms_thresh, float score_thresh, int anchors) {
    uint8_t* drop = (uint8_t*)calloc((size_t)anchors, 1);
    /* candidates = rows that pass the score threshold; a below-threshold column can
     * still suppress in the reference (its label is only NaN'ed by its own row
     * thread, racily); but such a column has conf < thresh <= row conf, so the
     * 'comp_conf > row_conf' test is false for it anyway. */
    for (int i = 0; i < anchors; ++i) {
        if (dets[i].confidence < score_thresh) {
            drop[i] = 1;
            continue;
        }
        for (int j = 0; j < anchors; ++j) {
            if (dets[j].label == dets[i].label && dets[j].confidence > dets[i].confidence) {
                if (orc_iou(dets[i].x, dets[i].y, dets[i].width, dets[i].height, dets[j].x,
                            dets[j].y, dets[j].width, dets[j].height) > nms_thresh) {
                    drop[i] = 1;
                    break;
                }
            }
        }
    }
    for (int i = 0; i < anchors; ++i)
        if (drop[i]) dets[i].label = NAN;
    free(drop);
}

static inline float clampf(float v, float lo, float hi) {
    /* std::clamp(v, lo, hi): (v < lo) ? lo : (hi < v) ? hi : v */
    return v < lo ? lo : (hi < v ? hi : v);
}

/* src/detect/detector.cpp:258-268 */
void orc_restore(orc_detection* d, const orc_preparam* p) {
    d->x = clampf((d->x - p->dw) * p->ratio, 0.0f, p->width);
    d->y = clampf((d->y - p->dh) * p->ratio, 0.0f, p->height);
    d->width = clampf(d->width * p->ratio, 0.0f, p->width - d->x);
    d->height = clampf(d->height * p->ratio, 0.0f, p->height - d->y);
}

/* ======================================================================= */
/* detect: composed host sequencing                                          */
/* ======================================================================= */

/* src/detect/detector.cu:380-421 (single image) / 439-502 (crops).  Q2 defined
 * variant: 128-filled canvas of stride out_w, resized image pasted at (left, top),
 * rows/cols falling outside the canvas dropped. */
void orc_preprocess(const uint8_t* src, int src_stride, int crop_x, int crop_y, int crop_w,
                    int crop_h, int out_w, int out_h, float* blob, orc_preparam* pp) {
    orc_preparam p;
    orc_preparam_make(crop_w, crop_h, out_w, out_h, &p);
    int rw, rh, top, bottom, left, right;
    orc_letterbox_geometry(&p, &rw, &rh, &top, &bottom, &left, &right);
    (void)bottom;
    (void)right;
    if (pp) *pp = p;

    /* image(Rect).clone(): a contiguous copy of the crop (detector.cpp:417-424) */
    uint8_t* crop = (uint8_t*)malloc((size_t)crop_w * crop_h * 3);
    for (int y = 0; y < crop_h; ++y)
        memcpy(crop + (size_t)y * crop_w * 3, src + (size_t)(crop_y + y) * src_stride + crop_x * 3,
               (size_t)crop_w * 3);

    uint8_t* canvas = (uint8_t*)malloc((size_t)out_w * out_h * 3);
    memset(canvas, 128, (size_t)out_w * out_h * 3);
    if (rw > 0 && rh > 0) {
        uint8_t* resized = (uint8_t*)malloc((size_t)rw * rh * 3);
        orc_resize_u8(crop, resized, 3, crop_w, crop_h, rw, rh);
        for (int y = 0; y < rh; ++y) {
            int cy = y + top;
            if (cy < 0 || cy >= out_h) continue;
            for (int x = 0; x < rw; ++x) {
                int cx = x + left;
                if (cx < 0 || cx >= out_w) continue;
                memcpy(canvas + ((size_t)cy * out_w + cx) * 3, resized + ((size_t)y * rw + x) * 3, 3);
            }
        }
        free(resized);
    }
    orc_blob(canvas, blob, out_w, out_h, 3, 1 / 255.f);
    free(canvas);
    free(crop);
}

/* src/detect/detector.cu:522-582 for one image: transpose -> decode -> NMS ->
 * drop NaN labels -> restore; ascending anchor order (Q10b). */
int orc_postprocess(const float* net_out, int channels, int anchors, int classes,
                    float nms_thresh, float conf_thresh, const orc_preparam* pp,
                    orc_detection* out, int cap) {
    float* t = (float*)malloc(sizeof(float) * (size_t)channels * anchors);
    orc_detection* dets = (orc_detection*)malloc(sizeof(orc_detection) * (size_t)anchors);
    orc_transpose(net_out, t, channels, anchors);
    orc_decode(t, dets, channels, anchors, classes);
    orc_nms(dets, nms_thresh, conf_thresh, anchors);
    int n = 0;
    for (int i = 0; i < anchors; ++i) {
        if (isnan(dets[i].label)) continue;
        orc_detection d = dets[i];
        orc_restore(&d, pp);
        if (n < cap) out[n] = d;
        ++n;
    }
    free(dets);
    free(t);
    return n;
}

/* ======================================================================= */
/* robot grouping                                                            */
/* ======================================================================= */

/* src/robot/robot.cpp:41-74 */
void orc_robot_set_detection(orc_robot* r, const orc_detection* car,
                             const orc_detection* armors, int n_armors) {
    memset(r, 0, sizeof(*r));
    r->rect[0] = car->x;
    r->rect[1] = car->y;
    r->rect[2] = car->width;
    r->rect[3] = car->height;
    if (n_armors <= 0) return;
    if (n_armors > ORC_MAX_ARMORS) n_armors = ORC_MAX_ARMORS;

    /* std::map<int,float> score_map; score_map[armor.label] += armor.confidence */
    int keys[ORC_MAX_ARMORS];
    float sums[ORC_MAX_ARMORS];
    int nk = 0;
    for (int i = 0; i < n_armors; ++i) {
        int key = (int)armors[i].label;
        int k = 0;
        for (; k < nk; ++k)
            if (keys[k] == key) break;
        if (k == nk) {
            keys[nk] = key;
            sums[nk] = 0.0f;
            ++nk;
        }
        sums[k] += armors[i].confidence;
    }
    /* max_element over the map iterates ascending key; first max wins */
    int best = -1;
    for (int pass_key_idx = 0; pass_key_idx < nk; ++pass_key_idx) {
        /* select the pass_key_idx-th smallest key */
        int sel = -1;
        for (int k = 0; k < nk; ++k) {
            int smaller = 0;
            for (int q = 0; q < nk; ++q)
                if (keys[q] < keys[k]) ++smaller;
            if (smaller == pass_key_idx) sel = k;
        }
        if (best < 0 || sums[best] < sums[sel]) best = sel;
    }
    int label = keys[best];
    float confidence = sums[best];
    int count = 0;
    for (int i = 0; i < n_armors; ++i)
        if (armors[i].label == (float)label) ++count;
    confidence /= (float)count; /* float /= ptrdiff_t : converted to float */
    r->has_label = 1;
    r->label = label;
    r->confidence = confidence;
    r->n_armors = n_armors;
    for (int i = 0; i < n_armors; ++i) {
        r->armors[i] = armors[i];
        r->armors[i].x += car->x;
        r->armors[i].y += car->y;
    }
}

/* OpenCV saturate_cast<int>(float) == cvRound == lrintf under the default
 * rounding mode (round-half-to-even) [OpenCV behaviour, not in reference]. */
static inline int cv_round(float v) { return (int)lrintf(v); }

void orc_rect_round(const float rect[4], int out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = cv_round(rect[i]);
}

/* src/detect/detector.cpp:324-349 (Q11) */
float orc_compute_iou_bounding(const float a[4], const float b[4]) {
    float x1 = a[0] > b[0] ? a[0] : b[0];
    float y1 = a[1] > b[1] ? a[1] : b[1];
    float ar = a[0] + a[2], br = b[0] + b[2], ab = a[1] + a[3], bb = b[1] + b[3];
    float x2 = ar < br ? ar : br;
    float y2 = ab < bb ? ab : bb;
    float iw = 0, ih = 0;
    if (x1 < x2 && y1 < y2) {
        iw = x2 - x1;
        ih = y2 - y1;
    }
    float ux1 = a[0] < b[0] ? a[0] : b[0];
    float uy1 = a[1] < b[1] ? a[1] : b[1];
    float ux2 = ar > br ? ar : br;
    float uy2 = ab > bb ? ab : bb;
    float inter = iw * ih;
    float uni = (ux2 - ux1) * (uy2 - uy1);
    if (uni > 0) return inter / uni;
    return 0.0f;
}

/* src/detect/detector.cpp:427-454.  Robot::rect() hands computeIoU the
 * rounded-int rect converted back to float (robot.h:111, Q11). */
int orc_group_robots(const orc_robot* in, int n, float iou_thresh, orc_robot* out) {
    int n_out = 0;
    /* std::map<int, Robot>: keep (label, robot) sorted by label */
    orc_robot* map = (orc_robot*)malloc(sizeof(orc_robot) * (size_t)(n > 0 ? n : 1));
    int n_map = 0;
    for (int i = 0; i < n; ++i) {
        orc_robot robot = in[i];
        if (!robot.has_label) {
            out[n_out++] = robot;
            continue;
        }
        int k = 0;
        for (; k < n_map; ++k)
            if (map[k].label == robot.label) break;
        if (k == n_map) {
            map[n_map++] = robot;
        } else {
            int ra[4], rb[4];
            orc_rect_round(map[k].rect, ra);
            orc_rect_round(robot.rect, rb);
            float fa[4] = {(float)ra[0], (float)ra[1], (float)ra[2], (float)ra[3]};
            float fb[4] = {(float)rb[0], (float)rb[1], (float)rb[2], (float)rb[3]};
            if (orc_compute_iou_bounding(fa, fb) > iou_thresh) {
                continue;
            } else if (map[k].confidence < robot.confidence) {
                map[k] = robot; /* std::swap(exist_robot, robot); robot then discarded */
            }
        }
    }
    /* emit in ascending label */
    for (int pass = 0; pass < n_map; ++pass) {
        int sel = -1;
        for (int k = 0; k < n_map; ++k) {
            int smaller = 0;
            for (int q = 0; q < n_map; ++q)
                if (map[q].label < map[k].label) ++smaller;
            if (smaller == pass) sel = k;
        }
        out[n_out++] = map[sel];
    }
    free(map);
    return n_out;
}

/* src/detect/detector.cpp:420-421 : cv::Rect(float...) truncates toward zero */
void orc_crop_rect(const orc_detection* car, int out[4]) {
    out[0] = (int)car->x;
    out[1] = (int)car->y;
    out[2] = (int)car->width;
    out[3] = (int)car->height;
}

/* ======================================================================= */
/* small fixed-size matrices, OpenCV Matx semantics [OpenCV behaviour]       */
/* ======================================================================= */

/* cv::Matx multiplication: s = 0; for k: s += a(i,k) * b(k,j)   (f32) */
static void matmul(const float* a, const float* b, float* out, int m, int l, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            float s = 0;
            for (int k = 0; k < l; ++k) s += a[i * l + k] * b[k * n + j];
            out[i * n + j] = s;
        }
}

/* Matx33f::inv() (DECOMP_LU) = closed-form adjugate in f32 */
int orc_inv3x3(const float a[9], float b[9]) {
#define A(i, j) a[(i)*3 + (j)]
    float d = A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) -
              A(0, 1) * (A(1, 0) * A(2, 2) - A(2, 0) * A(1, 2)) +
              A(0, 2) * (A(1, 0) * A(2, 1) - A(2, 0) * A(1, 1));
    if (d == 0) {
        memset(b, 0, 9 * sizeof(float));
        return 0;
    }
    d = 1 / d;
    b[0] = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) * d;
    b[1] = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * d;
    b[2] = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * d;
    b[3] = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) * d;
    b[4] = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * d;
    b[5] = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * d;
    b[6] = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) * d;
    b[7] = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * d;
    b[8] = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * d;
#undef A
    return 1;
}

/* Matx44f::inv() (DECOMP_LU) = cv::LU on [A | I], f32, partial pivoting */
int orc_inv4x4(const float a[16], float out[16]) {
    float A[16], B[16];
    memcpy(A, a, sizeof(A));
    memset(B, 0, sizeof(B));
    for (int i = 0; i < 4; ++i) B[i * 4 + i] = 1.0f;
    const int m = 4, n = 4;
    for (int i = 0; i < m; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (fabsf(A[j * m + i]) > fabsf(A[k * m + i])) k = j;
        if (fabsf(A[k * m + i]) < 1.1920929e-06f /* FLT_EPSILON*10 */) {
            memset(out, 0, 16 * sizeof(float));
            return 0;
        }
        if (k != i) {
            for (int j = i; j < m; ++j) {
                float t = A[i * m + j];
                A[i * m + j] = A[k * m + j];
                A[k * m + j] = t;
            }
            for (int j = 0; j < n; ++j) {
                float t = B[i * n + j];
                B[i * n + j] = B[k * n + j];
                B[k * n + j] = t;
            }
        }
        float d = -1 / A[i * m + i];
        for (int j = i + 1; j < m; ++j) {
            float alpha = A[j * m + i] * d;
            for (int q = i + 1; q < m; ++q) A[j * m + q] += alpha * A[i * m + q];
            for (int q = 0; q < n; ++q) B[j * n + q] += alpha * B[i * n + q];
        }
    }
    for (int i = m - 1; i >= 0; --i)
        for (int j = 0; j < n; ++j) {
            float s = B[i * n + j];
            for (int k = i + 1; k < m; ++k) s -= A[i * m + k] * B[k * n + j];
            B[i * n + j] = s / A[i * m + i];
        }
    memcpy(out, B, sizeof(B));
    return 1;
}

/* ======================================================================= */
/* locator                                                                   */
/* ======================================================================= */

struct orc_locator {
    orc_locator_cfg cfg;
    int wz, hz;
    float K[9], Kinv[9];
    float L2C[16];
    float c2l_R[9], c2l_t[3];
    float C2W[16];
    float* depth;
    float* background;
    float* diff;
    float** queue; /* oldest first */
    int queue_len;
    /* cluster() products */
    int n_fg, cap_fg;
    float* fg_xyz;
    int* fg_pixel;
    int* fg_cluster;
    int* pixel_to_fg; /* hz*wz, -1 when none */
    int n_clusters;
    int* cluster_sizes;
};

void orc_locator_cfg_default(orc_locator_cfg* c) {
    memset(c, 0, sizeof(*c));
    c->zoom_factor = 0.5f;
    c->queue_size = 3;
    c->min_depth_diff = 500;
    c->max_depth_diff = 4000;
    c->cluster_tolerance = 400;
    c->min_cluster_size = 8;
    c->max_cluster_size = 1000;
    c->max_distance = 29300;
}

/* src/locate/locate.cpp:112-146.  Q12: images zero-initialised. */
orc_locator* orc_locator_create(const orc_locator_cfg* c) {
    orc_locator* l = (orc_locator*)calloc(1, sizeof(orc_locator));
    l->cfg = *c;
    l->wz = (int)((float)c->image_width * c->zoom_factor);
    l->hz = (int)((float)c->image_height * c->zoom_factor);
    memcpy(l->K, c->intrinsic, sizeof(l->K));
    memcpy(l->L2C, c->lidar_to_camera, sizeof(l->L2C));
    orc_inv3x3(l->K, l->Kinv);
    float c2l[16];
    orc_inv4x4(l->L2C, c2l);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) l->c2l_R[i * 3 + j] = c2l[i * 4 + j];
        l->c2l_t[i] = c2l[i * 4 + 3];
    }
    orc_inv4x4(c->world_to_camera, l->C2W);
    size_t px = (size_t)l->wz * l->hz;
    l->depth = (float*)calloc(px, sizeof(float));
    l->background = (float*)calloc(px, sizeof(float));
    l->diff = (float*)calloc(px, sizeof(float));
    l->queue = (float**)calloc((size_t)c->queue_size + 1, sizeof(float*));
    l->pixel_to_fg = (int*)malloc(px * sizeof(int));
    for (size_t i = 0; i < px; ++i) l->pixel_to_fg[i] = -1;
    return l;
}

void orc_locator_destroy(orc_locator* l) {
    if (!l) return;
    for (int i = 0; i < l->queue_len; ++i) free(l->queue[i]);
    free(l->queue);
    free(l->depth);
    free(l->background);
    free(l->diff);
    free(l->fg_xyz);
    free(l->fg_pixel);
    free(l->fg_cluster);
    free(l->pixel_to_fg);
    free(l->cluster_sizes);
    free(l);
}

/* src/locate/locate.cpp:37-42 : (C2W * L2C) * [p;1] , left-associative Matx products */
void orc_locator_lidar_to_world(const orc_locator* l, const float p[3], float out[3]) {
    float v[4] = {p[0], p[1], p[2], 1.0f};
    float M[16], w[4];
    matmul(l->C2W, l->L2C, M, 4, 4, 4);
    matmul(M, v, w, 4, 4, 1);
    out[0] = w[0];
    out[1] = w[1];
    out[2] = w[2];
}

/* src/locate/locate.cpp:54-61 (Q16): R * ( (Kinv * d) * [u/z, v/z, 1] + t ) */
void orc_locator_camera_to_lidar(const orc_locator* l, const float uvd[3], float out[3]) {
    float z = l->cfg.zoom_factor;
    float cam[3] = {uvd[0] / z, uvd[1] / z, 1.0f};
    float Ks[9];
    for (int i = 0; i < 9; ++i) Ks[i] = l->Kinv[i] * uvd[2];
    float q[3];
    matmul(Ks, cam, q, 3, 3, 1);
    for (int i = 0; i < 3; ++i) q[i] = q[i] + l->c2l_t[i];
    matmul(l->c2l_R, q, out, 3, 3, 1);
}

/* src/locate/locate.cpp:73-81 */
void orc_locator_lidar_to_camera(const orc_locator* l, const float p[3], float out[3]) {
    float v[4] = {p[0], p[1], p[2], 1.0f};
    float c4[4], c[3];
    matmul(l->L2C, v, c4, 4, 4, 1);
    matmul(l->K, c4, c, 3, 3, 1); /* get_minor<3,1>(0,0): first three rows */
    float z = l->cfg.zoom_factor;
    out[0] = c[0] * z / c[2];
    out[1] = c[1] * z / c[2];
    out[2] = c[2];
}

/* src/locate/locate.cpp:158-220.  Deterministic readings: Q13 sequential point
 * order (highest index wins the depth pixel, background = running max), Q14 queue
 * processed oldest -> newest, Q15 u==Wz / v==Hz (and NaN) treated as out of range. */
void orc_locator_update(orc_locator* l, const float* xyz, int n, int stride_bytes) {
    size_t px = (size_t)l->wz * l->hz;
    memset(l->depth, 0, px * sizeof(float));
    memset(l->diff, 0, px * sizeof(float));
    if (!xyz || n <= 0) return; /* locate.cpp:160-171 : early return, nothing queued */

    for (int i = 0; i < n; ++i) {
        const float* pt = (const float*)((const char*)xyz + (size_t)i * stride_bytes);
        if (pt[0] == 0 && pt[1] == 0 && pt[2] == 0) continue;
        if (pt[0] > l->cfg.max_distance) continue;
        float uvd[3];
        orc_locator_lidar_to_camera(l, pt, uvd);
        float u = uvd[0], v = uvd[1], d = uvd[2];
        if (!(u >= 0 && u < (float)l->wz && v >= 0 && v < (float)l->hz)) continue;
        size_t idx = (size_t)(int)v * l->wz + (int)u;
        if (d > l->background[idx]) l->background[idx] = d;
        l->depth[idx] = d;
    }

    float* clone = (float*)malloc(px * sizeof(float));
    memcpy(clone, l->depth, px * sizeof(float));
    l->queue[l->queue_len++] = clone;
    if (l->queue_len > l->cfg.queue_size) {
        free(l->queue[0]);
        memmove(l->queue, l->queue + 1, sizeof(float*) * (size_t)(l->queue_len - 1));
        --l->queue_len;
    }

    for (int q = 0; q < l->queue_len; ++q) {
        const float* img = l->queue[q];
        for (size_t p = 0; p < px; ++p) {
            float value = img[p];
            if (value == 0) continue;
            float diff = l->background[p] - value;
            if (diff >= l->cfg.min_depth_diff && diff <= l->cfg.max_depth_diff) l->diff[p] = value;
        }
    }
}

static int find_root(int* parent, int i) {
    while (parent[i] != i) {
        parent[i] = parent[parent[i]];
        i = parent[i];
    }
    return i;
}

typedef struct {
    int root, size;
} cluster_rank;

static int cmp_rank(const void* a, const void* b) {
    const cluster_rank* x = (const cluster_rank*)a;
    const cluster_rank* y = (const cluster_rank*)b;
    if (x->size != y->size) return y->size - x->size; /* larger first */
    return x->root - y->root;                         /* then lower min index */
}

/* src/locate/locate.cpp:231-264.  The PCL part (Q17) [PCL/FLANN public behaviour,
 * pcl::EuclideanClusterExtraction over search::KdTree, version unpinned by
 * src/locate/CMakeLists.txt:2]: connected components of the graph
 * "squared f32 distance < tolerance^2" (FLANN radius search is strict), components
 * outside [min,max] stay unlabeled, kept clusters ordered by size descending;
 * ties (unstable std::sort in PCL) resolved here by lowest member index. */
void orc_locator_cluster(orc_locator* l) {
    size_t px = (size_t)l->wz * l->hz;
    for (int i = 0; i < l->n_fg; ++i) l->pixel_to_fg[l->fg_pixel[i]] = -1;
    l->n_fg = 0;
    l->n_clusters = 0;

    for (int i = 0; i < l->hz; ++i) {
        for (int j = 0; j < l->wz; ++j) {
            float value = l->diff[(size_t)i * l->wz + j];
            if (value == 0) continue;
            if (l->n_fg == l->cap_fg) {
                l->cap_fg = l->cap_fg ? l->cap_fg * 2 : 1024;
                l->fg_xyz = (float*)realloc(l->fg_xyz, sizeof(float) * 3 * (size_t)l->cap_fg);
                l->fg_pixel = (int*)realloc(l->fg_pixel, sizeof(int) * (size_t)l->cap_fg);
                l->fg_cluster = (int*)realloc(l->fg_cluster, sizeof(int) * (size_t)l->cap_fg);
            }
            float uvd[3] = {(float)j, (float)i, value};
            orc_locator_camera_to_lidar(l, uvd, l->fg_xyz + 3 * (size_t)l->n_fg);
            l->fg_pixel[l->n_fg] = i * l->wz + j;
            l->pixel_to_fg[(size_t)i * l->wz + j] = l->n_fg;
            ++l->n_fg;
        }
    }
    (void)px;
    int n = l->n_fg;
    if (n == 0) return;

    int* parent = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) parent[i] = i;
    float tol2 = l->cfg.cluster_tolerance * l->cfg.cluster_tolerance;
    for (int i = 0; i < n; ++i) {
        const float* a = l->fg_xyz + 3 * (size_t)i;
        for (int j = i + 1; j < n; ++j) {
            const float* b = l->fg_xyz + 3 * (size_t)j;
            float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
            float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < tol2) {
                int ra = find_root(parent, i), rb = find_root(parent, j);
                if (ra < rb)
                    parent[rb] = ra;
                else if (rb < ra)
                    parent[ra] = rb;
            }
        }
    }
    int* size = (int*)calloc((size_t)n, sizeof(int));
    for (int i = 0; i < n; ++i) size[find_root(parent, i)]++;
    cluster_rank* ranks = (cluster_rank*)malloc(sizeof(cluster_rank) * (size_t)n);
    int nr = 0;
    for (int i = 0; i < n; ++i)
        if (size[i] > 0 && size[i] >= l->cfg.min_cluster_size && size[i] <= l->cfg.max_cluster_size) {
            ranks[nr].root = i;
            ranks[nr].size = size[i];
            ++nr;
        }
    qsort(ranks, (size_t)nr, sizeof(cluster_rank), cmp_rank);
    int* root_to_id = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) root_to_id[i] = -1;
    free(l->cluster_sizes);
    l->cluster_sizes = (int*)malloc(sizeof(int) * (size_t)(nr > 0 ? nr : 1));
    for (int k = 0; k < nr; ++k) {
        root_to_id[ranks[k].root] = k;
        l->cluster_sizes[k] = ranks[k].size;
    }
    for (int i = 0; i < n; ++i) l->fg_cluster[i] = root_to_id[find_root(parent, i)];
    l->n_clusters = nr;
    free(root_to_id);
    free(ranks);
    free(size);
    free(parent);
}

/* src/locate/locate.cpp:337-350 (Q19) */
void orc_locator_zoom(const orc_locator* l, const int rect[4], int out[4]) {
    float z = l->cfg.zoom_factor;
    float center_x = (float)rect[0] * z + (float)rect[2] * z * 0.5f;
    float center_y = (float)rect[1] * z + (float)rect[3] * z * 0.5f;
    int ret_width = (int)((float)rect[2] * z);
    int ret_height = (int)((float)rect[3] * z);
    int ret_x = (int)(center_x - (float)ret_width * 0.5f);
    int ret_y = (int)(center_y - (float)ret_height * 0.5f);
    /* cv::Rect &= image_rect */
    int x1 = ret_x > 0 ? ret_x : 0;
    int y1 = ret_y > 0 ? ret_y : 0;
    int x2 = ret_x + ret_width < l->wz ? ret_x + ret_width : l->wz;
    int y2 = ret_y + ret_height < l->hz ? ret_y + ret_height : l->hz;
    int w = x2 - x1, h = y2 - y1;
    if (w <= 0 || h <= 0) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    out[0] = x1;
    out[1] = y1;
    out[2] = w;
    out[3] = h;
}

/* src/locate/locate.cpp:276-311 (Q18).  rect is the robot's Rect2f; Robot::rect()
 * rounds it to ints (robot.h:111).  Winner = bucket with most points, first max in
 * ascending key (-1 first).  Mean = sequential f32 sum in scan order / n; then
 * lidarToWorld; then setLocation: (float)(double(v) * 1e-3) (robot.h:93-95). */
int orc_locator_search(const orc_locator* l, const float rectf[4], float xyz_m[3]) {
    int ri[4], r[4];
    orc_rect_round(rectf, ri);
    orc_locator_zoom(l, ri, r);
    int nb = l->n_clusters + 1; /* bucket 0 = key -1 */
    int* count = (int*)calloc((size_t)nb, sizeof(int));
    for (int v = r[1]; v < r[1] + r[3]; ++v)
        for (int u = r[0]; u < r[0] + r[2]; ++u) {
            float depth = l->diff[(size_t)v * l->wz + u];
            if (depth == 0) continue;
            int idx = l->pixel_to_fg[(size_t)v * l->wz + u];
            int cid = idx >= 0 ? l->fg_cluster[idx] : -1;
            count[cid + 1]++;
        }
    int best = -1;
    for (int b = 0; b < nb; ++b)
        if (count[b] > 0 && (best < 0 || count[best] < count[b])) best = b;
    if (best < 0) {
        free(count);
        return 0;
    }
    float sx = 0, sy = 0, sz = 0;
    for (int v = r[1]; v < r[1] + r[3]; ++v)
        for (int u = r[0]; u < r[0] + r[2]; ++u) {
            float depth = l->diff[(size_t)v * l->wz + u];
            if (depth == 0) continue;
            int idx = l->pixel_to_fg[(size_t)v * l->wz + u];
            int cid = idx >= 0 ? l->fg_cluster[idx] : -1;
            if (cid + 1 != best) continue;
            float uvd[3] = {(float)u, (float)v, depth}, p[3];
            orc_locator_camera_to_lidar(l, uvd, p);
            sx += p[0];
            sy += p[1];
            sz += p[2];
        }
    float nf = (float)count[best];
    float loc[3] = {sx / nf, sy / nf, sz / nf}, w[3];
    orc_locator_lidar_to_world(l, loc, w);
    for (int i = 0; i < 3; ++i) xyz_m[i] = (float)((double)w[i] * 1e-3);
    free(count);
    return 1;
}

int orc_locator_width(const orc_locator* l) { return l->wz; }
int orc_locator_height(const orc_locator* l) { return l->hz; }
float* orc_locator_depth_image(orc_locator* l) { return l->depth; }
float* orc_locator_background_image(orc_locator* l) { return l->background; }
float* orc_locator_diff_image(orc_locator* l) { return l->diff; }
int orc_locator_num_foreground(const orc_locator* l) { return l->n_fg; }
const float* orc_locator_foreground_xyz(const orc_locator* l) { return l->fg_xyz; }
const int* orc_locator_foreground_pixel(const orc_locator* l) { return l->fg_pixel; }
const int* orc_locator_foreground_cluster(const orc_locator* l) { return l->fg_cluster; }
int orc_locator_num_clusters(const orc_locator* l) { return l->n_clusters; }
int orc_locator_cluster_size(const orc_locator* l, int id) {
    return (id >= 0 && id < l->n_clusters) ? l->cluster_sizes[id] : 0;
}

/* ======================================================================= */
/* plain f32 direct convolution (cross-check of the torch network oracle)    */
/* ======================================================================= */

void orc_conv2d_nchw(const float* x, int n, int cin, int h, int w, const float* wt,
                     const float* bias, int cout, int kh, int kw, int stride, int pad,
                     int silu, float* y) {
    int ho = (h + 2 * pad - kh) / stride + 1;
    int wo = (w + 2 * pad - kw) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int oc = 0; oc < cout; ++oc) {
            float* yo = y + ((size_t)b * cout + oc) * ho * wo;
            for (int i = 0; i < ho * wo; ++i) yo[i] = bias ? bias[oc] : 0.0f;
            for (int ic = 0; ic < cin; ++ic) {
                const float* xi = x + ((size_t)b * cin + ic) * h * w;
                for (int r = 0; r < kh; ++r)
                    for (int s = 0; s < kw; ++s) {
                        float wv = wt[(((size_t)oc * cin + ic) * kh + r) * kw + s];
                        for (int oy = 0; oy < ho; ++oy) {
                            int iy = oy * stride - pad + r;
                            if (iy < 0 || iy >= h) continue;
                            for (int ox = 0; ox < wo; ++ox) {
                                int ix = ox * stride - pad + s;
                                if (ix < 0 || ix >= w) continue;
                                yo[oy * wo + ox] += wv * xi[iy * w + ix];
                            }
                        }
                    }
            }
            if (silu)
                for (int i = 0; i < ho * wo; ++i) yo[i] = yo[i] / (1.0f + expf(-yo[i]));
        }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
