// api_comm.cpp -- the multi-GPU part of the drop-in boundary: which rank owns a camera / LiDAR stream, and the
// ONE exchange of the path -- an all-gather of the final robot list as fixed-size records, once per batch of
// frames.  No reference counterpart: the reference pins device 0 (src/detect/detector.cpp:61) and
// has a single stream (samples/sample_radar.h:106-127).  One process per GPU; the host application hands the
// 128-byte id that rank 0 creates to the other ranks by whatever it has (environment, file, MPI, a socket).
//
// Transports:
//   RMR_TRANSPORT_RCCL  ncclAllGather over xGMI on the rank's GPU.  librccl.so is opened at the first
//                       communicator, so hosts that never go multi-GPU do not load it.
//   RMR_TRANSPORT_FILE  the same exchange through a shared directory (the id is its path): for hosts and CI
//                       boxes without GPUs, and for testing a C++ caller with several processes on one box.
#include <dirent.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"

using namespace rmr;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;   // the dlopen / dlsym message, captured where it happened (dlerror() clears itself when read)
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            const char* e = dlerror();
            r.why += std::string(r.why.empty() ? "" : "; ") + (e ? e : name);
        }
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
            r.why = "ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy missing from the library";
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
        fail(RMR_ERR_DEVICE, "librccl.so cannot be loaded: %s", r.why.c_str());
    return r;
}

// What a failed communicator set-up needs to be diagnosed from a log: the IPC mode (this driver only supports dmabuf IPC:
// HSA_ENABLE_IPC_MODE_LEGACY=0), RCCL's own switches, who we are
std::string comm_env_report(int transport, int rank, int world, int device) {
    const auto ev = [](const char* k) {
        const char* v = std::getenv(k);
        return std::string(k) + "=" + (v ? v : "(unset)");
    };
    return "transport " + std::string(transport == RMR_TRANSPORT_RCCL ? "RCCL" : "FILE") + ", rank " + std::to_string(rank) + " of " + std::to_string(world) +
           ", device " + std::to_string(device) + "; " + ev("HSA_ENABLE_IPC_MODE_LEGACY") + " " + ev("NCCL_DEBUG") + " " + ev("NCCL_SOCKET_IFNAME") + " " +
           ev("NCCL_IB_DISABLE") + " " + ev("HIP_VISIBLE_DEVICES") + " " + ev("ROCR_VISIBLE_DEVICES") +
           " (set NCCL_DEBUG=INFO for RCCL's own account of the failure)";
}

void nccl_check(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) fail(RMR_ERR_DEVICE, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
}

}  // namespace

struct rmr_comm {
    int transport = 0, rank = 0, world = 1, device = 0;
    // RCCL
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<unsigned char> send, recv;
    // FILE: every file of a communicator carries its EPOCH, so a second communicator on a directory -- a rank restarted
    // after a crash, a caller-supplied path -- can never read a record file an earlier one left behind.  The epoch is a
    // COLLECTIVE value (round 4; before, every rank derived it from its own surviving markers and two ranks could
    // disagree after a clean close or a crash): rank 0 picks 1 + the highest epoch of ANY file still in the directory
    // (after a clean close has swept the directory that is epoch 0 again: safe, nothing of the earlier epoch 0 is left to read)
    // and hands it to every other rank in a token handshake (agree_on_epoch), which makes creation a collective call
    // like ncclCommInitRank.
    std::string dir;
    long long epoch = 0, seq = 0;
    bool joined = false;   // the session marker is written: the leave protocol applies
    std::string file(const char* kind, long long s, int r) const {
        return dir + "/e" + std::to_string(epoch) + "." + kind + (s >= 0 ? std::to_string(s) : std::string()) + "." + std::to_string(r);
    }
    bool wait_all(const char* kind, double seconds) const {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < world; ++r) {
            struct stat st;
            while (stat(file(kind, -1, r).c_str(), &st) != 0) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
                std::this_thread::sleep_for(std::chrono::microseconds(500));
            }
        }
        return true;
    }
    static void touch(const std::string& path) { std::ofstream f(path, std::ios::binary | std::ios::trunc); }
    static bool publish(const std::string& path, const std::string& text) {   // readers never see a partial file
        const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
        {
            std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
            f << text;
            if (!f) return false;
        }
        return std::rename(tmp.c_str(), path.c_str()) == 0;
    }
    static std::string slurp(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        std::string t;
        if (f) std::getline(f, t);
        return t;
    }
    // Rank r != 0 publishes hello.<r> = a fresh token and waits for welcome.<r> = "<token> <epoch>" carrying ITS token
    // (a welcome left by an earlier communicator, or one answering a hello a crashed process left, has another token and
    // is ignored).  Rank 0 picks the epoch, then keeps answering whatever token each hello currently holds until that
    // rank's session marker of the new epoch appears (the caller writes it right after this returns).
    // t0: when rmr_comm_create was entered -- ONE deadline of 120 s covers the handshake and rank 0's wait for the session markers
    static long long agree_on_epoch(const std::string& dir, int rank, int world, std::chrono::steady_clock::time_point t0) {
        const auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120); };
        if (rank != 0) {
            std::random_device rd;
            const std::string token = std::to_string(((unsigned long long)rd() << 32) ^ rd() ^ ((unsigned long long)getpid() << 20) ^
                                                     (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
            const std::string hello = dir + "/hello." + std::to_string(rank), welcome = dir + "/welcome." + std::to_string(rank);
            if (!publish(hello, token)) fail(RMR_ERR_RUNTIME, "rmr_comm_create: cannot write into '%s'", dir.c_str());
            for (;;) {
                const std::string w = slurp(welcome);
                const size_t sp = w.find(' ');
                if (sp != std::string::npos && w.compare(0, sp, token) == 0) {
                    std::remove(hello.c_str());
                    std::remove(welcome.c_str());
                    return std::atoll(w.c_str() + sp + 1);
                }
                if (late()) fail(RMR_ERR_RUNTIME, "rmr_comm_create: rank 0 did not answer in '%s' within 120 s", dir.c_str());
                std::this_thread::sleep_for(std::chrono::microseconds(300));
            }
        }
        long long last = -1;
        if (DIR* d = opendir(dir.c_str())) {
            while (const dirent* e = readdir(d)) {
                const char* n = e->d_name;
                if (n[0] == 'e' && n[1] >= '0' && n[1] <= '9') last = std::max(last, std::atoll(n + 1));
            }
            closedir(d);
        }
        const long long epoch = last + 1;
        std::vector<std::string> welcomed(world);
        for (int pending = world - 1; pending > 0;) {
            pending = 0;
            for (int r = 1; r < world; ++r) {
                struct stat st;
                if (stat((dir + "/e" + std::to_string(epoch) + ".session." + std::to_string(r)).c_str(), &st) == 0) continue;
                ++pending;
                const std::string token = slurp(dir + "/hello." + std::to_string(r));
                if (!token.empty() && token != welcomed[r] && publish(dir + "/welcome." + std::to_string(r), token + " " + std::to_string(epoch)))
                    welcomed[r] = token;
            }
            if (pending && late()) fail(RMR_ERR_RUNTIME, "rmr_comm_create: not every rank joined '%s' within 120 s", dir.c_str());
            if (pending) std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
        return epoch;
    }
    // Leaving: a peer may still be reading this rank's last record files, so a rank first says it will read no more
    // ("done"), removes its own files once every rank has said so, and says "gone"; rank 0 waits for that and sweeps
    // the markers (and the directory, when rmr_comm_unique_id made it).  A peer that never arrives (crashed) costs a
    // bounded wait and leaves its epoch's files behind -- which the next communicator's epoch makes harmless.
    void leave_dir() {
        if (!joined) return;
        touch(file("done", -1, rank));
        const bool all = wait_all("done", 5.0);
        if (all) {
            for (long long s = std::max(0LL, seq - 2); s < seq; ++s) std::remove(file("", s, rank).c_str());
        }
        touch(file("gone", -1, rank));
        if (rank != 0 || !all || !wait_all("gone", 5.0)) return;
        for (int r = 0; r < world; ++r)
            for (const char* kind : {"done", "gone", "session"}) std::remove(file(kind, -1, r).c_str());
        const size_t slash = dir.find_last_of('/');
        if (dir.compare(slash == std::string::npos ? 0 : slash + 1, 9, "rmr_comm_") == 0) (void)rmdir(dir.c_str());
    }
    ~rmr_comm() {
        if (comm) (void)rccl().CommDestroy(comm);
        if (stream) (void)hipStreamDestroy(stream);
        if (transport == RMR_TRANSPORT_FILE) leave_dir();
    }
};

extern "C" {

int rmr_stream_owner(int stream, int world) { return world > 0 && stream >= 0 ? stream % world : -1; }

int rmr_streams_of_rank(int n_streams, int rank, int world, int* out, int cap) {
    int n = 0;
    for (int s = 0; s < n_streams; ++s)
        if (rmr_stream_owner(s, world) == rank) {
            if (out && n < cap) out[n] = s;
            ++n;
        }
    return n;
}

rmr_status rmr_comm_unique_id(int transport, char* id) {
    return guarded([&] {
        if (!id) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_unique_id: null id");
        std::memset(id, 0, RMR_COMM_ID_BYTES);
        if (transport == RMR_TRANSPORT_RCCL) {
            static_assert(sizeof(ncclUniqueId) == RMR_COMM_ID_BYTES, "the id is an ncclUniqueId");
            ncclUniqueId u;
            nccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
            std::memcpy(id, &u, sizeof(u));
        } else if (transport == RMR_TRANSPORT_FILE) {
            const char* base = std::getenv("TMPDIR");
            std::string tmpl = std::string(base && *base ? base : "/tmp") + "/rmr_comm_XXXXXX";
            if (tmpl.size() >= RMR_COMM_ID_BYTES) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_unique_id: TMPDIR is too long");
            std::vector<char> buf(tmpl.begin(), tmpl.end());
            buf.push_back(0);
            if (!mkdtemp(buf.data())) fail(RMR_ERR_RUNTIME, "rmr_comm_unique_id: cannot create a directory under %s", tmpl.c_str());
            std::memcpy(id, buf.data(), buf.size());
        } else {
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_unique_id: unknown transport %d", transport);
        }
    });
}

rmr_status rmr_comm_create(int transport, int device, int rank, int world, const char* id, rmr_comm** out) {
    return guarded([&] {
        if (!out || !id || world < 1 || rank < 0 || rank >= world) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_create: bad arguments");
        auto c = std::make_unique<rmr_comm>();
        c->transport = transport, c->rank = rank, c->world = world, c->device = device;
        // RMR_COMM_INJECT_FAILURE=1 (tests): fail where ncclCommInitRank would, with the same account of the environment, so that
        // the first multi-GPU run's log can be checked to carry what a diagnosis needs before such a run exists
        if (const char* inj = std::getenv("RMR_COMM_INJECT_FAILURE"))
            if (std::atoi(inj) != 0)
                fail(RMR_ERR_DEVICE, "communicator set-up failed (injected): %s [%s]", transport == RMR_TRANSPORT_RCCL ? "ncclCommInitRank" : "file handshake",
                     comm_env_report(transport, rank, world, device).c_str());
        if (transport == RMR_TRANSPORT_RCCL) {
            DeviceCtx& ctx = device_ctx(device);  // fails loudly without a usable GPU
            ctx.use();
            ncclUniqueId u;
            std::memcpy(&u, id, sizeof(u));
            RMR_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            const ncclResult_t ie = rccl().CommInitRank(&c->comm, world, u, rank);
            if (ie != ncclSuccess)
                fail(RMR_ERR_DEVICE, "ncclCommInitRank failed: %s [%s]", rccl().GetErrorString ? rccl().GetErrorString(ie) : "RCCL error",
                     comm_env_report(transport, rank, world, device).c_str());
        } else if (transport == RMR_TRANSPORT_FILE) {
            c->dir.assign(id, strnlen(id, RMR_COMM_ID_BYTES));
            struct stat st;
            if (c->dir.empty() || stat(c->dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode))
                fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_create: '%s' is not a directory", c->dir.c_str());
            const auto t_enter = std::chrono::steady_clock::now();
            c->epoch = rmr_comm::agree_on_epoch(c->dir, rank, world, t_enter);
            rmr_comm::touch(c->file("session", -1, rank));   // for ranks != 0 this is also the ack rank 0 waits for
            const double left = 120.0 - std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count();
            if (rank == 0 && !c->wait_all("session", std::max(0.0, left)))
                fail(RMR_ERR_RUNTIME, "rmr_comm_create: not every rank joined '%s' within 120 s", c->dir.c_str());
            c->joined = true;
        } else {
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_create: unknown transport %d", transport);
        }
        *out = c.release();
    });
}

void rmr_comm_destroy(rmr_comm* c) { delete c; }

rmr_status rmr_comm_all_gather_records(rmr_comm* c, const rmr_robot_record* mine, int n, rmr_robot_record* all) {
    return guarded([&] {
        if (!c || !mine || !all || n <= 0) fail(RMR_ERR_INVALID_ARGUMENT, "rmr_comm_all_gather_records: bad arguments");
        const size_t bytes = (size_t)n * sizeof(rmr_robot_record);
        if (c->world == 1 && c->transport == RMR_TRANSPORT_FILE) {
            std::memcpy(all, mine, bytes);
            return;
        }
        if (c->transport == RMR_TRANSPORT_RCCL) {
            // a few KB per rank: latency-bound, so one call per batch of frames.  Host records are staged through
            // device buffers because the collective moves device memory.
            RMR_HIP(hipSetDevice(c->device));
            c->send.ensure(bytes);
            c->recv.ensure(bytes * c->world);
            RMR_HIP(hipMemcpyAsync(c->send.p, mine, bytes, hipMemcpyHostToDevice, c->stream));
            nccl_check(rccl().AllGather(c->send.p, c->recv.p, bytes, ncclChar, c->comm, c->stream), "ncclAllGather");
            RMR_HIP(hipMemcpyAsync(all, c->recv.p, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
            RMR_HIP(hipStreamSynchronize(c->stream));
            return;
        }
        // FILE: publish e<epoch>.<seq>.<rank> (written under another name, then renamed: readers never see a partial file),
        // collect everybody's, retire the files of two rounds ago (every rank has read them by then)
        const long long seq = c->seq++;
        const auto name = [&](long long s, int r) { return c->file("", s, r); };
        {
            const std::string tmp = name(seq, c->rank) + ".tmp";
            std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
            f.write((const char*)mine, (std::streamsize)bytes);
            f.close();
            if (!f || std::rename(tmp.c_str(), name(seq, c->rank).c_str()) != 0)
                fail(RMR_ERR_RUNTIME, "rmr_comm_all_gather_records: cannot publish into '%s'", c->dir.c_str());
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < c->world; ++r) {
            for (;;) {
                std::ifstream f(name(seq, r), std::ios::binary);
                if (f) {
                    f.read((char*)all + (size_t)r * bytes, (std::streamsize)bytes);
                    if ((size_t)f.gcount() == bytes) break;
                    fail(RMR_ERR_RUNTIME, "rmr_comm_all_gather_records: rank %d published %zu bytes, expected %zu (ranks disagree on n)", r,
                         (size_t)f.gcount(), bytes);
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                    fail(RMR_ERR_RUNTIME, "rmr_comm_all_gather_records: rank %d did not arrive within 120 s", r);
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
        if (seq >= 2) std::remove(name(seq - 2, c->rank).c_str());
    });
}

// rmr_robot[n_frames * cap] + counts -> records [n_frames][max_per_frame], zero padded; a slot is valid when
// flags bit 2 is set (bit 0: label, bit 1: location)
rmr_status rmr_pack_robot_records(const rmr_robot* robots, const int* counts, int n_frames, int cap, int stream_id,
                                  int max_per_frame, rmr_robot_record* out) {
    return guarded([&] {
        if (!robots || !counts || !out || n_frames <= 0 || cap <= 0 || max_per_frame <= 0)
            fail(RMR_ERR_INVALID_ARGUMENT, "rmr_pack_robot_records: bad arguments");
        std::memset(out, 0, (size_t)n_frames * max_per_frame * sizeof(rmr_robot_record));
        for (int f = 0; f < n_frames; ++f) {
            const int m = std::min(std::min(counts[f], cap), max_per_frame);
            for (int i = 0; i < m; ++i) {
                const rmr_robot& r = robots[(size_t)f * cap + i];
                rmr_robot_record& o = out[(size_t)f * max_per_frame + i];
                std::memcpy(o.rect, r.rect, sizeof(o.rect));
                std::memcpy(o.location, r.location, sizeof(o.location));
                o.confidence = r.confidence;
                o.label = r.has_label ? r.label : -1;
                o.flags = (r.has_label ? 1 : 0) | (r.has_location ? 2 : 0) | 4;
                o.stream_id = stream_id;
                o.frame_id = f;
            }
        }
    });
}

}  // extern "C"
