// yolov8.h -- the network behind Detector: what the reference delegates to a TensorRT engine
// built from car.onnx / armor.onnx (src/detect/detector.cpp:177-243, detector.h:122).
// Public Ultralytics YOLOv8 architecture executed as a flat list of
// strided-view ops on one activation arena; weights come from a *.rmrw pack.
#pragma once
#include <hip/hip_fp16.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "preprocess.h"
#include "conv_igemm.h"

namespace rmr {

struct WeightPack {
    float depth = 0, width = 0;
    unsigned max_ch = 0, nc = 0, reg_max = 16;
    struct Tensor {
        std::vector<unsigned> dims;
        std::vector<float> data;
    };
    std::map<std::string, Tensor> tensors;
    static WeightPack load(const std::string& path);
    const Tensor& get(const std::string& name) const;
};

// A strided NHWC view into the arena: element offset of the buffer for image 0, elements per
// pixel, first channel, channels, spatial size.
struct View {
    size_t off = 0;  // in halves (or floats for f32 buffers), per-chunk buffer start
    int cs = 0, co = 0, c = 0, h = 0, w = 0;
};

class Yolov8 {
   public:
    static int tune_file_version();   // the header version of '<pack>.tune' / RMR_PLAN files this build reads and writes
    // in_w/in_h: network input size (multiples of 32); max_batch: largest forward() batch
    // fp8: the 3x3 / stride-1 layers with >= 64 input channels run on e4m3 operands (RMR_FP8=1 forces it)
    Yolov8(DeviceCtx& ctx, const std::string& pack_path, int expect_nc, int in_w, int in_h,
           int max_batch, bool fp8 = false);

    int nc() const { return nc_; }
    int anchors() const { return anchors_; }
    int channels() const { return 4 + nc_; }
    double flops_per_image() const { return flops_; }
    int max_batch() const { return max_batch_; }
    // bytes of activation memory (all arenas, all images of a chunk) and images per chunk
    size_t arena_bytes() const { return arena_.n * sizeof(__half) + arena32_.n * sizeof(float) + arena8_.n; }
    int chunk() const { return chunk_; }
    // Debug / parity hook: the output of a backbone / neck stage ("model.0" ... "model.21") of image `img` of
    // the last forward(), as f32 [h][w][c] on the host (out = nullptr: the dimensions only).  Needs a detector
    // created with RMR_ARENA_REUSE=0 (otherwise later layers have overwritten the stage outputs).
    bool read_feature(hipStream_t s, const std::string& name, int img, float* out, int dims[3]);

    // network input: f16 NHWC with 8 channels per pixel (RGB + 5 zero lanes), [max_batch]
    __half* input() { return input_.p; }
    // network output: f32 [batch][4+nc][anchors], the tensor TensorRT hands to postprocess
    float* output() { return output_.p; }
    void forward(hipStream_t s, int batch);
    // The same, taking the frames / crops themselves: src is a DEVICE array of batch letterbox
    // descriptors (preprocess.h).  Where the first layer has the stem shape it samples the sources
    // itself (conv_stem.hip) and input() is not written; otherwise the canvases are built first.
    void forward(hipStream_t s, int batch, const LetterboxDesc* src, int fill, float scale);
    ~Yolov8();

   private:
    enum OpKind { OP_CONV, OP_SPPF, OP_UP, OP_HEAD, OP_QUANT };
    struct ConvW {
        DevBuf<__half> w;
        DevBuf<__half> w32;  // 3x3 layers with Cin % 32 == 0: the LDS images conv_t32 streams (pack_conv_weights_t32)
        DevBuf<__half> w1d;  // 3x3 layers with Cin % 32 == 0: the Winograd F(2, 3) transformed weights conv_w1d streams
        DevBuf<float> b;
        DevBuf<unsigned char> w8;  // fp8 plan: e4m3 LDS images (pack_conv_weights_t32f8) ...
        DevBuf<float> wscale;      // ... and the scale of every output channel
        int cout = 0, cout_pad = 0, cin = 0, k = 0, K = 0, Kp = 0;
    };
    struct Op {
        OpKind kind;
        int conv = -1;      // index into convs_
        View in, out, res;  // res.c == 0 : none
        View pre;           // pre.c != 0: f32 half-resolution addend before the activation (ConvArgs::pre)
        // planar channel groups (ConvArgs::in_slab_c ...): `in` / `out` is the first slab (cs = slab
        // width, c = all channels), the others follow at equal distances of slab_step elements per image
        int in_slab_c = 0, out_slab_c = 0;
        size_t in_slab_step = 0, out_slab_step = 0;
        bool in_is_input = false;
        bool out_f32 = false;
        // fp8 plan: OP_CONV reads the e4m3 copy of `in` at q_off (bytes per image into arena8_, rows of q_pitch
        // bytes); OP_QUANT writes it
        bool fp8 = false;
        size_t q_off = 0;
        int q_pitch = 0;
        // OP_CONV of the fp8 plan whose output feeds another e4m3 layer: it writes that layer's input itself
        bool q_out = false;
        bool q_only = false;   // ... and nothing else reads the tensor: its f16 copy is not written
        size_t q_out_off = 0;
        int q_out_pitch = 0;
        int stride = 1, act = 1;
        // a C2f bottleneck of the conv_wsf shape (two 3x3 / 48 -> 48 convolutions on 160-wide maps, shortcut = the first one's
        // input, hidden tensor read by nobody else): this op is the FIRST convolution, fuse_with the index of the second.  A
        // tuned choice of 340.. for this op runs both in one launch (conv_wsf: the hidden tensor stays in LDS); the second op
        // then carries the choice kFusedAway and launches nothing
        int fuse_with = -1;
        // a run of consecutive, mutually independent convolutions (the Detect head: the same depth of its branches): index into
        // groups_, -1 = none.  A tuned choice of kSbGroupBase + variant on the group's FIRST op runs all of them in one conv_sb
        // launch; the others then carry kGroupedAway and launch nothing
        int group = -1;
        // OP_HEAD
        View box, cls;
        int head_stride = 0, a_off = 0;
        // ... in the fused form (head_fused_kernel: the last 1x1 convolutions of both branches + the decode): the two
        // convolutions (indices into convs_) and the feature views they read; box / cls are not allocated
        int box_conv = -1, cls_conv = -1;
        View hb, hc;
    };

    View alloc(int h, int w, int c, bool f32 = false);
    // Plan-time liveness: every buffer lives from the first to the last op that touches it; buffers whose
    // lifetimes do not meet share arena memory (27 GiB -> a few GiB for a 256-image armor detector).
    // RMR_ARENA_REUSE=0 keeps one region per buffer, which the stage-output hook (read_feature) needs.
    void compact_arenas();
    struct Alloc {
        size_t off, size;  // per image, in elements of its arena
        int arena;         // 0 f16, 1 f32, 2 e4m3
        int group;         // allocations of one group keep their relative placement (slabs)
    };
    std::vector<Alloc> allocs_;
    int alloc_group_ = 0;   // != 0 while a group is being allocated
    int next_group_ = 1;
    bool arena_reuse_ = true;
    static View slice(const View& v, int co, int c);
    // ci0 / ci_n: the slice of input channels to keep (ci_n = 0: all); no_bias: a zero bias
    int add_conv_weights(const WeightPack& p, const std::string& name, int cin_pad, int ci0 = 0, int ci_n = 0,
                         bool no_bias = false);
    void upload_t32(ConvW& cw, const std::vector<__half>& packed);
    int add_fused_head_weights(const WeightPack& p, const std::string& a, const std::string& b);
    void conv(int widx, const View& in, const View& out, int stride, int act, const View* res = nullptr,
              bool out_f32 = false, bool in_is_input = false, const View* pre = nullptr);
    // up: x is concat[up2x(*up), skip] with the skip in x's trailing channels -- cv1 is then computed
    // as W_skip.skip + up2x(W_up.up) and the upsampled slice of x is never written or read
    View c2f(const WeightPack& p, const std::string& name, const View& x, int n, bool shortcut,
             const View* out_view, const View* up = nullptr);
    bool fuse_up_ = true;  // RMR_FUSE_UP=0: upsample kernel + cv1 over the concat
    bool slabs_ = true;    // RMR_SLABS=0: every C2f keeps its chunks interleaved in one wide buffer
    bool pw_can(int K, int N, int h, int w, bool pre) const;
    void run_op(hipStream_t s, int op_index, int chunk_n, size_t img_base);
    ConvArgs conv_args(int op_index, int chunk_n, size_t img_base);
    int tune_conv(hipStream_t s, const ConvArgs& a, float* best_ms_out = nullptr);
    ConvArgs fused_args(int op_index, int n, size_t img0);   // ConvArgs of a fused bottleneck (op_index = its first convolution)
    static constexpr int kFusedAway = 399;
    std::vector<std::vector<int>> groups_;                                   // op indices, in op order
    std::map<std::pair<int, int>, DevBuf<unsigned char>> group_tables_;      // (first op of a group, images) -> the launch's problem table
    void find_groups();
    std::vector<ConvArgs> group_args(int g, int n);
    void ensure_group_table(int g, int n, int variant);
    void tune_group(hipStream_t s, int g, int n, size_t img0);
    std::map<std::pair<int, int>, float> tuned_ms_;   // the tuner's time of the first convolution of a fusable pair

    DeviceCtx& ctx_;
    int nc_, in_w_, in_h_, max_batch_, chunk_;
    int anchors_ = 0;
    double flops_ = 0;
    std::vector<ConvW> convs_;
    std::vector<Op> ops_;
    struct Named {
        View v;
        int slab_c = 0;        // planar channel groups (0: interleaved)
        size_t slab_step = 0;
    };
    std::map<std::string, Named> named_;  // stage outputs by Ultralytics module name
    size_t arena_halves_ = 0, arena_floats_ = 0;  // per image
    DevBuf<__half> arena_;
    DevBuf<float> arena32_;
    DevBuf<unsigned char> arena8_;   // fp8 plan: quantised inputs of the e4m3 layers
    size_t arena_bytes8_ = 0;        // per image
    bool fp8_ = false;
    bool fp8_layer_ = true;   // planner state: whether the layers being added may take e4m3 operands
    std::vector<__half> last_packed_;  // add_conv_weights' packed f16 weights, for the e4m3 packer
    DevBuf<__half> input_;
    DevBuf<float> output_;
    // autotuned kernel choice per (op, images in the launch): 0..99 conv_igemm tile, 100..199
    // conv_dma tile, 200..299 conv_halo tile; + 1000 * split for split-K (conv_dma only)
    std::map<std::pair<int, int>, int> tuned_;
    bool autotune_ = true;
    bool pinned_ = false;  // RMR_PLAN=<file>: kernels come from that plan, nothing is timed (reproducible outputs)
    bool tuned_dirty_ = false;
    // split-K workspace (partial tiles) and re-arming ticket counters, shared by all layers
    DevBuf<float> splitk_ws_;
    DevBuf<int> splitk_cnt_;
    void launch_choice(hipStream_t s, ConvArgs a, int choice);
    // Small batches are launch-bound (170 kernels of a few microseconds): once every layer of a
    // batch size is tuned, its forward pass is captured into a hipGraph and replayed.
    struct Graph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        const LetterboxDesc* src = nullptr;  // captured by address
    };
    std::map<int, Graph> graphs_;
    int graph_max_batch_ = 8;
    bool all_tuned(int n) const;
    // letterbox fused into the first layer (RMR_FUSE_LB=0 keeps the separate kernel)
    bool fuse_lb_ = true;
    const LetterboxDesc* lb_src_ = nullptr;  // set for the duration of a fused forward()
    int lb_fill_ = 0;
    float lb_scale_ = 0.f;
    std::string tune_path_;  // '<pack>.tune': choices persist like the reference's engine cache
    unsigned long long plan_signature() const;
    void load_tuning();
    bool choice_supported(const ConvArgs& a, int choice) const;
    void save_tuning();
};

}  // namespace rmr
