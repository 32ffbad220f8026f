// api_upload.cpp -- input staging for throughput hosts: the frames and clouds of step i + 1 travel to HBM on a copy
// stream of their own while step i computes, so the PCIe upload (109 MB per 64-frame step at 640 x 640 + 30 k points:
// ~2 ms) disappears behind the ~30 ms of kernels.  The reference has the upload inside its cycle -- every image is
// memcpy'd into one mapped pinned buffer and the resize kernel reads it across PCIe (src/detect/detector.cu:388-399,
// 455-470; detector.cpp:133-141) -- which is what the batch-1 path of this library does as well (FrameStage); this is
// the batched form of it: a ring of device slots, the caller's page-locked buffers as the source, no host-side copy.
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"

using namespace rmr;

struct rmr_upload {
    DeviceCtx& ctx;
    hipStream_t stream = nullptr;
    size_t bytes_per_slot = 0;
    struct Slot {
        DevBuf<unsigned char> buf;
        hipEvent_t done = nullptr;
        bool pending = false;
    };
    std::vector<Slot> slots;
    explicit rmr_upload(DeviceCtx& c) : ctx(c) {}
    ~rmr_upload() {
        if (stream) (void)hipStreamSynchronize(stream);
        for (Slot& s : slots)
            if (s.done) (void)hipEventDestroy(s.done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

rmr_status rmr_pinned_alloc_on(int device, size_t bytes, void** out) {
    return guarded([&] {
        if (!out || bytes == 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_pinned_alloc: bad arguments");
        // page-locked under the context of the device whose upload ring will read it (a rank of a multi-GPU host must
        // not open a context on GPU 0 for this), and portable: visible to every device context of the process
        device_ctx(device).use();  // fails loudly without a GPU
        RMR_HIP(hipHostMalloc(out, bytes, hipHostMallocPortable));
    });
}

rmr_status rmr_pinned_alloc(size_t bytes, void** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;   // the calling thread's current device
    return rmr_pinned_alloc_on(dev, bytes, out);
}

void rmr_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

rmr_status rmr_upload_create(int device, int n_slots, size_t bytes_per_slot, rmr_upload** out) {
    return guarded([&] {
        if (!out || n_slots < 1 || n_slots > 64 || bytes_per_slot == 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_upload_create: bad arguments");
        DeviceCtx& ctx = device_ctx(device);
        ctx.use();
        auto u = std::make_unique<rmr_upload>(ctx);
        u->bytes_per_slot = bytes_per_slot;
        RMR_HIP(hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking));
        u->slots.resize(n_slots);
        for (auto& s : u->slots) {
            s.buf.alloc(bytes_per_slot);
            RMR_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        }
        *out = u.release();
    });
}

void rmr_upload_destroy(rmr_upload* u) { delete u; }

rmr_status rmr_upload_begin(rmr_upload* u, int slot, const void* const* src, const size_t* bytes, int n, void** dev_out) {
    return guarded([&] {
        if (!u || slot < 0 || slot >= (int)u->slots.size() || n < 0 || (n > 0 && (!src || !bytes || !dev_out)))
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_upload_begin: bad arguments");
        u->ctx.use();
        rmr_upload::Slot& s = u->slots[slot];
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            if (!src[i] && bytes[i]) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_upload_begin: block %d is null", i);
            if (off + bytes[i] > u->bytes_per_slot)
                fail(RMR_ERR_CAPACITY, "rmr_upload_begin: %d blocks need more than the slot's %zu bytes", n, u->bytes_per_slot);
            dev_out[i] = s.buf.p + off;
            off = (off + bytes[i] + 255) & ~(size_t)255;
        }
        // consecutive blocks that are consecutive in host memory too (one big capture buffer) travel as ONE copy
        for (int i = 0; i < n;) {
            int j = i;
            size_t run = bytes[i];
            while (j + 1 < n && bytes[j] % 256 == 0 && (const char*)src[j] + bytes[j] == (const char*)src[j + 1]) run += bytes[++j];
            if (run) RMR_HIP(hipMemcpyAsync(dev_out[i], src[i], run, hipMemcpyHostToDevice, u->stream));
            i = j + 1;
        }
        RMR_HIP(hipEventRecord(s.done, u->stream));
        s.pending = true;
    });
}

rmr_status rmr_upload_wait(rmr_upload* u, int slot) {
    return guarded([&] {
        if (!u || slot < 0 || slot >= (int)u->slots.size()) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_upload_wait: bad arguments");
        rmr_upload::Slot& s = u->slots[slot];
        if (!s.pending) return;
        RMR_HIP(hipEventSynchronize(s.done));
        s.pending = false;
    });
}

}  // extern "C"
