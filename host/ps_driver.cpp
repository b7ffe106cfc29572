// ps_driver.cpp -- the reference's driver loop on top of the C ABI, in C++.
//
// DoParallelProcess (particleSystem.cpp:1843-1928) runs, per iteration, task 3
// (init_iframe), task 8 (build_grid), reads hostGridMax[0], runs task 6 (calc_forces) if
// the biggest chunk is not empty, optionally fetches the buffers back to the host mirrors
// after each stage (pFetchBack), and prints four wall-clock figures per iteration.  This
// program is that loop with the three stage calls going to libpsamd.so; it links against
// nothing but include/psamd.h (no Python, no torch), which is what a maintainer dropping the
// library into the reference would write.  Setup follows the reference too: DoInit sizes,
// then the fill stage (task 5) places the particles in order.
//
// usage: ps_driver [--n N | --cloud file.f32] [--iters K] [--dt DT] [--seed S]
//                  [--fetch-back] [--age A] [--describe]
//   --cloud    raw little-endian float32 x,y,z triples (tests/golden/g2_cloud_*.f32)
//   --describe host-only: print the sizes DoInit derives and exit (no GPU needed)
// Last line of output: "state-hash <16 hex digits> live <n>", a digest of the P_DATA_TYPE
// records of every slot (pad bytes excluded), for comparison with the oracle.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "psamd.h"

namespace {

double now_secs()   // getCurrentTimeInSecs of the reference's common code
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

struct Particle72 {   // P_DATA_TYPE, common.h:94-120
    int32_t id, cell, chunk, seg_type, seg_tid;
    uint8_t seg_fault, is_parent, pad[2];
    float w, age, fertility_age, x, y, z, vx, vy, vz, ax, ay, az;
};
static_assert(sizeof(Particle72) == 72, "P_DATA_TYPE is 72 bytes");

#define CHECK(call)                                                                          \
    do {                                                                                     \
        const int rc_ = (call);                                                              \
        if (rc_ != PSAMD_OK) {                                                               \
            std::fprintf(stderr, "%s failed: %s (%s)\n", #call, psamd_status_string(rc_),    \
                         ctx ? psamd_last_error(ctx) : "");                                  \
            return 1;   /* the reference printf()s and exit(1)s at this point */             \
        }                                                                                    \
    } while (0)

// Position-weighted sum of the 18 dwords of every record (the two pad bytes masked out),
// modulo 2^64: cheap to reproduce with numpy on the oracle's array.
uint64_t state_digest(const std::vector<Particle72> &p)
{
    uint64_t h = 0, k = 0;
    for (const Particle72 &r : p) {
        uint32_t w[18];
        std::memcpy(w, &r, sizeof w);
        w[5] &= 0x0000ffffu;
        for (int j = 0; j < 18; j++, k++) h += (uint64_t)w[j] * (k % 65521u + 1u);
    }
    return h;
}

}  // namespace

int main(int argc, char **argv)
{
    int64_t n = 4096;
    int iters = 10;
    double dt = -1.0;
    uint32_t seed = 12345;
    bool fetch_back = false, describe = false;
    float age0 = -1.0f;
    std::string cloud;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--n") n = std::atoll(next());
        else if (a == "--cloud") cloud = next();
        else if (a == "--iters") iters = std::atoi(next());
        else if (a == "--dt") dt = std::atof(next());
        else if (a == "--seed") seed = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--age") age0 = (float)std::atof(next());
        else if (a == "--fetch-back") fetch_back = true;
        else if (a == "--describe") describe = true;
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }

    psamd_ctx *ctx = nullptr;
    psamd_config cfg;
    CHECK(psamd_default_config(&cfg));
    if (dt > 0) cfg.dt = dt;
    if (n > cfg.max_particles_num) cfg.max_particles_num = (int32_t)n;

    psamd_sizes sz;
    CHECK(psamd_describe(&cfg, &sz, nullptr, nullptr, nullptr, nullptr));   // DoInit, ps.cpp:2204-2222
    std::printf("grid %d^3 cells, %d chunks, container %d slots, %d queue records, cell list %d, chunk list %d\n",
                sz.grid_dim, sz.num_chunks, sz.container_size, sz.queue_info_size, sz.max_per_cell, sz.max_per_chunk);
    if (describe) return 0;

    CHECK(psamd_create(&cfg, &ctx));

    // ---- fill stage (task 5, ps.cpp:915-1048) ----
    std::vector<float> xyz;
    if (!cloud.empty()) {
        FILE *f = std::fopen(cloud.c_str(), "rb");
        if (!f) { std::fprintf(stderr, "cannot open %s\n", cloud.c_str()); return 2; }
        std::fseek(f, 0, SEEK_END);
        const long bytes = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        xyz.resize((size_t)bytes / sizeof(float));
        if (std::fread(xyz.data(), sizeof(float), xyz.size(), f) != xyz.size()) { std::fclose(f); return 2; }
        std::fclose(f);
        n = (int64_t)xyz.size() / 3;
    } else {
        xyz.resize((size_t)n * 3);
        CHECK(psamd_uniform_cloud(ctx, n, seed, xyz.data()));
    }
    // adults from the first step (bodies younger than KID_AGE neither attract nor collide,
    // app_common.cu:240), immortal-by-birth fertility tags as in the golden fixtures
    const float age = age0 >= 0 ? age0 : (float)(40.0 * cfg.dt);
    std::vector<float> ages((size_t)n, age), fert((size_t)n);
    for (int64_t i = 0; i < n; i++) fert[(size_t)i] = (float)(1.0e6 + (double)i);
    int64_t placed = 0;
    CHECK(psamd_fill_particles(ctx, n, xyz.data(), nullptr, nullptr, ages.data(), fert.data(), nullptr, &placed));
    std::printf("placed %lld particles\n", (long long)placed);

    // host mirrors, only touched with --fetch-back (pFetchBack in the reference)
    std::vector<Particle72> hostParticles;
    std::vector<char> hostTdata, hostQueueInfo;
    std::vector<int32_t> hostQueue, hostChunkgrid, hostCellgrid;
    if (fetch_back) {
        hostParticles.resize((size_t)sz.container_size);
        hostTdata.resize((size_t)sz.container_size * 24);
        hostQueueInfo.resize((size_t)sz.queue_info_size * 24);
        hostQueue.resize((size_t)sz.container_size);
        hostChunkgrid.resize((size_t)sz.n_chunkgrid);
        hostCellgrid.resize((size_t)sz.n_cellgrid);
    }
    int32_t hostGridMax[2] = {0, 0};

    const double lStartTime = now_secs();
    for (int niter = 0; niter < iters; niter++) {
        const double time0 = now_secs();
        CHECK(psamd_init_iframe(ctx));                         // task 3
        if (fetch_back) {   // task 3 leaves every list empty and the maxima zero: nothing to fetch
            std::fill(hostChunkgrid.begin(), hostChunkgrid.end(), 0);
            std::fill(hostCellgrid.begin(), hostCellgrid.end(), 0);
            hostGridMax[0] = hostGridMax[1] = 0;
        }
        const double time1 = now_secs();
        CHECK(psamd_build_grid(ctx));                          // task 8
        CHECK(psamd_get_gridmax(ctx, hostGridMax));            // read by the driver at ps.cpp:1900
        if (fetch_back) {
            CHECK(psamd_download_chunkgrid(ctx, hostChunkgrid.data()));
            CHECK(psamd_download_cellgrid(ctx, hostCellgrid.data()));
            CHECK(psamd_download_queues(ctx, hostQueueInfo.data(), hostQueue.data()));
            CHECK(psamd_download_particles(ctx, hostParticles.data(), 0, sz.container_size));
            CHECK(psamd_download_tdata(ctx, hostTdata.data(), 0, sz.container_size));
        }
        const double time2 = now_secs();
        const int biggestChunkSize = hostGridMax[0];
        if (biggestChunkSize > 0) CHECK(psamd_calc_forces(ctx));   // task 6, all chunk subtasks in one call
        else CHECK(psamd_synchronize(ctx));
        if (fetch_back) {
            CHECK(psamd_download_queues(ctx, hostQueueInfo.data(), hostQueue.data()));
            CHECK(psamd_download_particles(ctx, hostParticles.data(), 0, sz.container_size));
        } else {
            CHECK(psamd_synchronize(ctx));
        }
        const double time3 = now_secs();
        std::printf(">>>>>>>>>>> Execution time of iteration (sec): \n%f\n%f\n%f\n%f\n\n\n\n", time3 - time0, time1 - time0,
                    time2 - time1, time3 - time2);
    }
    const double lEndTime = now_secs();
    std::printf("total %f s for %d iterations\n", lEndTime - lStartTime, iters);

    // ---- final state, field by field (the two pad bytes are not part of the state) ----
    hostParticles.resize((size_t)sz.container_size);
    CHECK(psamd_download_particles(ctx, hostParticles.data(), 0, sz.container_size));
    const uint64_t h = state_digest(hostParticles);
    int64_t live = 0;
    for (const Particle72 &p : hostParticles) live += (p.cell >= 0) ? 1 : 0;
    psamd_counters ctr;
    CHECK(psamd_get_counters(ctx, &ctr));
    std::printf("deaths %lld+%lld survives %lld relocations %lld births %lld\n", (long long)ctr.deaths_age,
                (long long)ctr.deaths_collision, (long long)ctr.survives, (long long)ctr.relocations, (long long)ctr.births);
    std::printf("state-hash %016llx live %lld\n", (unsigned long long)h, (long long)live);
    CHECK(psamd_destroy(ctx));
    return 0;
}
