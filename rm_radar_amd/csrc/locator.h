// locator.h -- device-resident Locator (src/locate/locator.h:53-98, src/locate/locate.cpp).
#pragma once
#include "common.h"

namespace rmr {

// constants every locator kernel needs, passed by value as a kernel argument
struct LocParams {
    float K[9];        // intrinsic_
    float L2C[16];     // lidar_to_camera_transform_
    float Kinv[9];     // intrinsic_inv_
    float R[9];        // camera_to_lidar_rotate_
    float t[3];        // camera_to_lidar_translate_
    float L2W[16];     // camera_to_world_transform_ * lidar_to_camera_transform_
    float zoom;
    int wz, hz;
    float min_diff, max_diff, max_distance, tol2;
    int min_cluster, max_cluster;
};

// per-frame products of cluster(): what search() needs
struct FrameSlot {
    int* n_fg = nullptr;       // [1] foreground count (clamped to max_foreground)
    int* n_clusters = nullptr; // [1]
    int* fg_pixel = nullptr;   // [max_fg]  v*W+u, scan order
    float* fg_xyz = nullptr;   // [max_fg*3] lidar frame, mm
    int* fg_cluster = nullptr; // [max_fg]  cluster id or -1
};

class Locator {
   public:
    explicit Locator(const rmr_locator_cfg& cfg);
    ~Locator();

    void update(const float* xyz, int n, int stride_bytes, int mem);
    void cluster();
    // update + cluster + keep(f) for f = 0 .. n_frames-1 (same results), the cluster stage as one pass over all frames
    void update_cluster_batch(const float* const* clouds, const int* n_points, int stride_bytes, int mem, int n_frames);
    void search(rmr_robot* robots, int n, int slot);  // slot -1 = current frame
    // kept frames 0 .. n_frames-1 in one pass: robots[f * cap + i], i < counts[f]; one upload of the
    // rects, one launch per frame, one download, ONE synchronisation
    void search_batch(rmr_robot* robots, const int* counts, int n_frames, int cap);
    // the same in two halves: enqueue (no synchronisation) / wait and write the locations
    void search_batch_begin(const rmr_robot* robots, const int* counts, int n_frames, int cap);
    void search_batch_end(rmr_robot* robots, const int* counts, int n_frames, int cap);
    void keep(int frame);

    int width() const { return prm_.wz; }
    int height() const { return prm_.hz; }
    void read_image(int which, float* host_out);
    void write_image(int which, const float* host_in);
    // temporal state (background + depth ring) as one host blob: a stream can restart, or move to
    // another GPU, without re-accumulating its background
    size_t state_bytes() const;
    void save_state(void* host_out, size_t cap);
    void load_state(const void* host_in, size_t bytes);
    void transform(int which, const float in[3], float out[3]) const;
    void zoom(const int rect[4], int out[4]) const;
    void foreground(float* xyz, int* pixel, int* cluster, int cap, int* n);
    int num_clusters();

   private:
    void update_into(const float* xyz, int n, int stride_bytes, int mem, float* diff_out);
    void update_batch(const float* const* clouds, const int* n_points, int stride_bytes, int mem, int n_frames);
    void cluster_frames(const float* diff, int n_frames, int first_slot);
    float* image_ptr(int which);
    FrameSlot make_slot();

    rmr_locator_cfg cfg_;
    DeviceCtx& ctx_;
    hipStream_t stream_ = nullptr;
    LocParams prm_{};
    size_t npx_ = 0;
    int max_clusters_ = 0;

    // update() state
    DevBuf<unsigned long long> key_;  // per pixel (point index + 1) << 32 | depth bits
    DevBuf<float> bg_, diff_;
    DevBuf<float> diff_batch_;        // [max_frames][npx]: the foreground images of a batch (update_cluster_batch)
    DevBuf<unsigned long long> key_batch_;  // [max_frames][npx]: per-frame key images of a batched update ...
    DevBuf<int> fmax_batch_;                // ... and each frame's largest positive depth per pixel (float bits)
    DevBuf<unsigned char> table_dev_;       // [max_frames] CloudTable
    PinnedBuf<unsigned char> table_pin_;
    DevBuf<float> ring_;              // [queue_size][npx]
    int ring_len_ = 0, ring_head_ = 0;  // oldest slot, number of valid slots
    DevBuf<float> cloud_;             // staging for host clouds
    PinnedBuf<float> cloud_pin_;

    // cluster() scratch
    DevBuf<int> blk_count_, blk_offset_;
    DevBuf<int> parent_, csize_, vroot_, vsize_, root_id_, counters_;   // counters_: [max_frames][4]
    DevBuf<int> slot_over_;   // [1 + max_frames]: foreground-capacity overflow of the current frame / of kept frame f
    DevBuf<float> fg_depth_;  // camera depth of each foreground point (prunes the pair tests of cluster())
    DevBuf<int> store_int_;
    DevBuf<float> store_f_;
    std::vector<FrameSlot> slots_;  // [0] = current frame, [1+f] = kept frame f
    DevBuf<int> slot_ints_;
    DevBuf<float> slot_floats_;

    // search() staging
    DevBuf<int> rects_dev_;
    DevBuf<float> loc_dev_;
    PinnedBuf<int> rects_pin_;
    PinnedBuf<float> loc_pin_;
    PinnedBuf<int> search_flags_;  // capacity-overflow flag read back with a batched search
};

// OpenCV-compatible small inverses (cv::Matx::inv, DECOMP_LU) used by the ctor
bool inv3x3_f32(const float a[9], float out[9]);
bool inv4x4_f32(const float a[16], float out[16]);

}  // namespace rmr
