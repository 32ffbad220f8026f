// common.cpp -- error slot, per-device context, HIP-event profiler.
#include "common.h"

#include <cstdlib>
#include <memory>

namespace rmr {

static thread_local std::string g_last_error;

void set_last_error(const std::string& s) { g_last_error = s; }
const std::string& last_error() { return g_last_error; }

// ---- Profiler ---------------------------------------------------------------------
thread_local int Profiler::stage = 0;

hipEvent_t Profiler::get_event() {
    std::lock_guard<std::mutex> lk(mu);
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    RMR_HIP(hipEventCreate(&e));
    return e;
}

void Profiler::push(const Pending& p) {
    std::lock_guard<std::mutex> lk(mu);
    pending.push_back(p);
    pending.back().level = on;
}

void Profiler::resolve() {
    std::lock_guard<std::mutex> lk(mu);
    if (!order_checked) {
        order_checked = true;
        if (const char* f = std::getenv("RMR_PROFILE_ORDER")) order_log = *f ? std::fopen(f, "a") : nullptr;
    }
    for (auto& p : pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            // RMR_PROFILE_ORDER: every profiled launch in enqueue order with its own duration
            if (order_log)
                std::fprintf(order_log, "%d %s|%s|%.0f|%.0f|%.6f\n", p.level, p.stage == 1 ? "car" : p.stage == 2 ? "armor" : "", p.name, p.flops,
                             p.bytes, ms);
            ProfEntry& e = stats[p.stage == 1 ? std::string("car|") + p.name : p.stage == 2 ? std::string("armor|") + p.name : std::string(p.name)];
            e.launches += 1;
            e.total_ms += ms;
            e.flops += p.flops;
            e.bytes += p.bytes;
        }
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.clear();
    if (order_log) std::fflush(order_log);
}

void Profiler::reset() {
    resolve();
    std::lock_guard<std::mutex> lk(mu);
    stats.clear();
}

Profiler::~Profiler() {
    for (auto& p : pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
    if (order_log) std::fclose(order_log);
}

// ---- DeviceCtx ----------------------------------------------------------------------
DeviceCtx::DeviceCtx(int dev) : device(dev) {
    RMR_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    RMR_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fail(RMR_ERR_DEVICE, "device %d is %s; librmr.so is built for gfx950 (MI355X) only", dev,
             prop.gcnArchName);
    num_cus = prop.multiProcessorCount;
    RMR_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
}

DeviceCtx::~DeviceCtx() {
    if (stream) (void)hipStreamDestroy(stream);
}

int usable_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

DeviceCtx& device_ctx(int device) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<DeviceCtx>> ctxs;
    std::lock_guard<std::mutex> lk(mu);
    int n = usable_device_count();
    if (n <= 0)
        fail(RMR_ERR_DEVICE, "no HIP device available: librmr.so has no CPU fallback");
    if (device < 0 || device >= n) fail(RMR_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device, n - 1);
    auto it = ctxs.find(device);
    if (it == ctxs.end()) it = ctxs.emplace(device, std::make_unique<DeviceCtx>(device)).first;
    it->second->use();
    return *it->second;
}

}  // namespace rmr
