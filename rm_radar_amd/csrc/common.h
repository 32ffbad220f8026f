// common.h -- shared host-side plumbing of librmr.so: error reporting, the per-device
// context (streams, pinned staging, scratch) and HIP-event kernel profiling.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rmr.h"

namespace rmr {

// ---- errors -----------------------------------------------------------------------
struct Error : std::exception {
    rmr_status code;
    std::string msg;
    Error(rmr_status c, std::string m) : code(c), msg(std::move(m)) {}
    const char* what() const noexcept override { return msg.c_str(); }
};

void set_last_error(const std::string& s);
const std::string& last_error();

[[noreturn]] inline void fail(rmr_status code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define RMR_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            ::rmr::fail(RMR_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// Wraps a C-ABI body: exceptions never cross the boundary.
template <class F>
inline rmr_status guarded(F&& f) {
    try {
        f();
        return RMR_OK;
    } catch (const Error& e) {
        set_last_error(e.msg);
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return RMR_ERR_RUNTIME;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return RMR_ERR_RUNTIME;
    }
}

// ---- device buffers -----------------------------------------------------------------
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        RMR_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
};

template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        RMR_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
};

// ---- profiling ----------------------------------------------------------------------
struct ProfEntry {
    long long launches = 0;
    double total_ms = 0, flops = 0, bytes = 0;
};

struct Profiler {
    int on = 0;  // 0 off, 1 every launch, 2 only launches that declare FLOPs (the convolution family)
    // which network stage the launches being enqueued belong to (set by RobotDetector around its two detect calls: 1 = car,
    // 2 = armor; 0 = a lone Detector): the stat name gets the prefix "car|" / "armor|", so a step whose two stages launch
    // the same shapes (batch 256: car chunk = armor chunk) can still be told apart.  Thread-local: the locate helper
    // thread's launches are never tagged.
    static thread_local int stage;
    // RMR_PROFILE_ORDER=<file> (read once): every profiled launch is also appended to that file in enqueue order when its
    // events are resolved ("<level> <stage>|<name>|<flops>|<bytes>|<ms>"), so that tools/pmc_traffic.py can give the k-th
    // convolution dispatch of a rocprofv3 --pmc pass its layer (PMC rows carry kernel symbols only) and tools/make_plan.py
    // can read every layer's own time inside the network
    FILE* order_log = nullptr;
    bool order_checked = false;
    std::mutex mu;
    struct Pending {
        hipEvent_t a, b;
        const char* name;
        double flops, bytes;
        int stage;
        int level = 0;   // prof.on when it was pushed
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::map<std::string, ProfEntry> stats;

    hipEvent_t get_event();
    void push(const Pending& p);
    void resolve();
    void reset();
    ~Profiler();
};

// RAII bracket around one launch; a no-op unless profiling is enabled.
struct ProfScope {
    Profiler* p;
    hipStream_t s;
    Profiler::Pending rec{};
    ProfScope(Profiler& prof, hipStream_t stream, const char* name, double flops = 0, double bytes = 0)
        : p(prof.on == 1 || (prof.on == 2 && flops > 0) ? &prof : nullptr), s(stream) {
        if (p) {
            rec = Profiler::Pending{p->get_event(), p->get_event(), name, flops, bytes, Profiler::stage};
            (void)hipEventRecord(rec.a, s);
        }
    }
    ~ProfScope() {
        if (p) {
            (void)hipEventRecord(rec.b, s);
            p->push(rec);
        }
    }
};

// ---- per-device context ---------------------------------------------------------------
struct DeviceCtx {
    int device = 0;
    hipStream_t stream = nullptr;  // unit-level entry points run here
    Profiler prof;
    int num_cus = 256;
    explicit DeviceCtx(int dev);
    ~DeviceCtx();
    void use() const { RMR_HIP(hipSetDevice(device)); }
};

// Fails loudly (RMR_ERR_DEVICE) when `device` is not a usable gfx950 GPU.
DeviceCtx& device_ctx(int device);
int usable_device_count();

}  // namespace rmr
