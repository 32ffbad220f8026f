// gather_ranks.cpp -- a C++ host of the multi-GPU boundary (include/rmr.h: rmr_stream_owner, rmr_comm_*,
// rmr_pack_robot_records): one process per rank, as a C++ SampleRadar (samples/sample_radar.h:106-127) per
// GPU would be.  Each rank fabricates the robots of its streams, packs them into records and gathers the
// world's records; every rank prints the same table.
// usage: gather_ranks <transport 0=rccl 1=file> <id file> <rank> <world> <n_streams> [rounds]
//   rank 0 creates the id and writes it to <id file>; the others wait for that file.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rmr.h"

int main(int argc, char** argv) {
    if (argc < 6) return std::fprintf(stderr, "usage: see gather_ranks.cpp\n"), 2;
    const int transport = std::atoi(argv[1]), rank = std::atoi(argv[3]), world = std::atoi(argv[4]), n_streams = std::atoi(argv[5]);
    const int rounds = argc > 6 ? std::atoi(argv[6]) : 3;
    char id[RMR_COMM_ID_BYTES];
    if (rank == 0) {
        if (rmr_comm_unique_id(transport, id) != RMR_OK) return std::fprintf(stderr, "%s\n", rmr_last_error()), 1;
        const std::string tmp = std::string(argv[2]) + ".tmp";
        std::FILE* f = std::fopen(tmp.c_str(), "wb");
        std::fwrite(id, 1, sizeof(id), f);
        std::fclose(f);
        std::rename(tmp.c_str(), argv[2]);
    } else {
        std::FILE* f = nullptr;
        for (int i = 0; i < 50000 && !(f = std::fopen(argv[2], "rb")); ++i) usleep(1000);
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) return std::fprintf(stderr, "no id file\n"), 1;
        std::fclose(f);
    }
    rmr_comm* comm = nullptr;
    if (rmr_comm_create(transport, 0, rank, world, id, &comm) != RMR_OK) return std::fprintf(stderr, "%s\n", rmr_last_error()), 1;

    std::vector<int> streams(n_streams);
    const int mine = rmr_streams_of_rank(n_streams, rank, world, streams.data(), n_streams);
    streams.resize(mine);
    for (int s : streams)
        if (rmr_stream_owner(s, world) != rank) return std::fprintf(stderr, "stream assignment is inconsistent\n"), 1;

    const int n_frames = 2, cap = 3, per_rank = (n_streams + world - 1) / world;  // every rank sends the same count
    for (int round = 0; round < rounds; ++round) {
        std::vector<rmr_robot_record> block((size_t)per_rank * n_frames * cap), all(block.size() * world);
        std::memset(block.data(), 0, block.size() * sizeof(rmr_robot_record));
        for (int k = 0; k < mine; ++k) {
            std::vector<rmr_robot> robots((size_t)n_frames * cap);
            std::memset(robots.data(), 0, robots.size() * sizeof(rmr_robot));
            int counts[2] = {2, 1};
            for (int f = 0; f < n_frames; ++f)
                for (int i = 0; i < counts[f]; ++i) {
                    rmr_robot& r = robots[(size_t)f * cap + i];
                    r.rect[0] = 100.f * streams[k] + f, r.rect[1] = (float)i, r.rect[2] = 10.f + round, r.rect[3] = 20.f;
                    r.has_label = i == 0, r.label = 3 + streams[k], r.confidence = 0.5f + 0.1f * i;
                    r.has_location = 1, r.location[0] = 1.f + streams[k], r.location[1] = 2.f + f, r.location[2] = 3.f + i;
                }
            if (rmr_pack_robot_records(robots.data(), counts, n_frames, cap, streams[k], cap,
                                       block.data() + (size_t)k * n_frames * cap) != RMR_OK)
                return std::fprintf(stderr, "%s\n", rmr_last_error()), 1;
        }
        if (rmr_comm_all_gather_records(comm, block.data(), (int)block.size(), all.data()) != RMR_OK)
            return std::fprintf(stderr, "%s\n", rmr_last_error()), 1;
        for (const rmr_robot_record& r : all)
            if (r.flags & 4)
                std::printf("round %d stream %d frame %d rect %g %g %g %g label %d loc %g %g %g\n", round, r.stream_id, r.frame_id, r.rect[0],
                            r.rect[1], r.rect[2], r.rect[3], r.label, r.location[0], r.location[1], r.location[2]);
    }
    rmr_comm_destroy(comm);
    std::printf("gather_ranks ok rank %d\n", rank);
    return 0;
}
