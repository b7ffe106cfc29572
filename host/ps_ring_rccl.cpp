// ps_ring_rccl.cpp -- the multi-GPU step in C++: libpsamd.so's four slab stage calls with the
// messages moved by RCCL (ncclSend / ncclRecv / ncclAllGather) straight between the contexts'
// device buffers, on the stream the stage kernels run on.  No Python, no torch: include/psamd.h,
// the HIP runtime and rccl.h.  This is what the reference's pmlib layer does for it between
// nodes (subscriptions to segments, ps.cpp:380-487); here one process drives one GPU.
//
//   one process per GPU (what a node runs):
//       ps_ring_rccl --world W --rank r --id-file /tmp/id --job J [--n N] [--iters K] [--seed S]
//     rank 0 writes the communicator's ncclUniqueId to the file (tagged with the job's nonce J), the others wait for it.
//   all slabs in ONE process on GPU 0 (what a one-GPU test box can run):
//       ps_ring_rccl --world W --loopback [--n N] [--iters K] [--seed S]
//     The communicator has a single rank; every message is an ncclSend to self matched by
//     an ncclRecv from self in the same group -- RCCL moves every byte, between the buffers of
//     different contexts.  This mode also runs the whole system in one plain context and
//     requires the union of the slabs to equal it byte for byte (P_DATA_TYPE of every slot).
//
// Message routes (particlesystem_amd/slab.py says the same in Python): after slab_build the
// halo snapshots (rank r's halo_out[above] -> rank r+1's halo_in[below]; halo_out[below] ->
// rank r-1's halo_in[above]) and the all-gather of the status records; after slab_pairs the
// force records of lent layers (force_out -> rank r-1's force_in); after slab_apply the
// particles that change owner, on the ring (xfer_out[below] -> rank (r-1)%W's xfer_in[above],
// xfer_out[above] -> rank (r+1)%W's xfer_in[below]).  All sizes are fixed by the plan; a
// message of 0 bytes does not exist.  Between one pair of ranks RCCL matches sends and
// receives by order, so both sides post them ordered by (peer, direction of travel).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "psamd.h"

namespace {

// A rank that fails must not leave its peers blocked in a receive: abort the communicator on the way out
// (psamd's own failures are collective -- every rank returns the error from the same slab_finish -- but a
// HIP or RCCL error, or a failure during set-up, is not).
ncclComm_t g_comm = nullptr;
int bail() { if (g_comm) { (void)ncclCommAbort(g_comm); g_comm = nullptr; } return 1; }
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return bail(); } } while (0)
#define NCCL_OK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return bail(); } } while (0)
#define PS_OK(ctx, call) do { const int rc_ = (call); if (rc_ != PSAMD_OK) { std::fprintf(stderr, "%s failed: %s (%s)\n", #call, psamd_status_string(rc_), (ctx) ? psamd_last_error(ctx) : ""); return bail(); } } while (0)

struct Slab {
    int rank = 0;
    psamd_ctx *ctx = nullptr;
    psamd_slab_buffers b{};
    psamd_slab_plan plan{};
};

struct Msg { void *buf; int64_t bytes; int peer; int dir; int hop; };   // dir: 0 travels down the ring, 1 up; hop: 1 to a ring neighbour, 2 to the rank beyond it

enum Phase { HALO, FORCE, XFER };

// what slab `s` of `world` sends in a phase: (buffer, bytes, destination rank, direction)
std::vector<Msg> sends_of(const Slab &s, int world, Phase ph)
{
    std::vector<Msg> v;
    const int r = s.rank;
    if (ph == HALO) {
        if (r > 0 && s.b.halo_out_bytes[0]) v.push_back({s.b.halo_out[0], s.b.halo_out_bytes[0], r - 1, 0, 1});
        if (r + 1 < world && s.b.halo_out_bytes[1]) v.push_back({s.b.halo_out[1], s.b.halo_out_bytes[1], r + 1, 1, 1});
    } else if (ph == FORCE) {
        if (r > 0 && s.b.force_out_bytes) v.push_back({s.b.force_out, s.b.force_out_bytes, r - 1, 0, 1});
    } else if (world > 1 && s.b.xfer_bytes) {
        v.push_back({s.b.xfer_out[0], s.b.xfer_bytes, (r - 1 + world) % world, 0, 1});
        v.push_back({s.b.xfer_out[1], s.b.xfer_bytes, (r + 1) % world, 1, 1});
        if (world >= 4 && s.b.xfer2_bytes) {     // a two-layer jump over a rank whose state is one layer: straight to rank +-2
            v.push_back({s.b.xfer2_out[0], s.b.xfer2_bytes, (r - 2 + world) % world, 0, 2});
            v.push_back({s.b.xfer2_out[1], s.b.xfer2_bytes, (r + 2) % world, 1, 2});
        }
    }
    return v;
}

// where slab `s` takes a message that rank `from` sent travelling in direction `dir`
Msg recv_of(const Slab &s, Phase ph, int from, int dir, int hop = 1)
{
    // a message travelling down arrives from above, and the other way round
    if (ph == HALO) return {s.b.halo_in[dir == 0 ? 1 : 0], s.b.halo_in_bytes[dir == 0 ? 1 : 0], from, dir, 1};
    if (ph == FORCE) return {s.b.force_in, s.b.force_in_bytes, from, dir, 1};
    if (hop == 2) return {s.b.xfer2_in[dir == 0 ? 1 : 0], s.b.xfer2_bytes, from, dir, 2};
    return {s.b.xfer_in[dir == 0 ? 1 : 0], s.b.xfer_bytes, from, dir, 1};
}

bool by_peer_then_dir(const Msg &a, const Msg &b) { return a.peer != b.peer ? a.peer < b.peer : (a.hop != b.hop ? a.hop < b.hop : a.dir < b.dir); }

// One phase's messages as ONE RCCL group.  `local`: the slabs this process holds (one per
// process in a real run; all of them in loopback mode, where every peer is comm rank 0).
int exchange(const std::vector<Slab> &local, int world, bool loopback, Phase ph, ncclComm_t comm, hipStream_t st, int64_t *moved)
{
    std::vector<Msg> sends, recvs;
    if (loopback) {
        // k-th receive from self = k-th send to self: enumerate the routes once for both lists
        for (const Slab &s : local)
            for (const Msg &m : sends_of(s, world, ph)) {
                const Msg r = recv_of(local[(size_t)m.peer], ph, s.rank, m.dir, m.hop);
                if (r.bytes != m.bytes) { std::fprintf(stderr, "message size mismatch %lld vs %lld\n", (long long)m.bytes, (long long)r.bytes); return 1; }
                sends.push_back({m.buf, m.bytes, 0, m.dir, m.hop});
                recvs.push_back({r.buf, r.bytes, 0, r.dir, r.hop});
            }
    } else {
        const Slab &s = local[0];
        const int r = s.rank;
        sends = sends_of(s, world, ph);
        // what the neighbours send here (a message exists iff its in-buffer has a size)
        if (ph == HALO) {
            if (r + 1 < world && s.b.halo_in_bytes[1]) recvs.push_back(recv_of(s, ph, r + 1, 0));
            if (r > 0 && s.b.halo_in_bytes[0]) recvs.push_back(recv_of(s, ph, r - 1, 1));
        } else if (ph == FORCE) {
            if (r + 1 < world && s.b.force_in_bytes) recvs.push_back(recv_of(s, ph, r + 1, 0));
        } else if (world > 1 && s.b.xfer_bytes) {
            recvs.push_back(recv_of(s, ph, (r + 1) % world, 0));
            recvs.push_back(recv_of(s, ph, (r - 1 + world) % world, 1));
            if (world >= 4 && s.b.xfer2_bytes) {
                recvs.push_back(recv_of(s, ph, (r + 2) % world, 0, 2));
                recvs.push_back(recv_of(s, ph, (r - 2 + world) % world, 1, 2));
            }
        }
        std::sort(sends.begin(), sends.end(), by_peer_then_dir);
        std::sort(recvs.begin(), recvs.end(), by_peer_then_dir);
    }
    if (sends.empty() && recvs.empty()) return 0;
    NCCL_OK(ncclGroupStart());
    for (const Msg &m : sends) { NCCL_OK(ncclSend(m.buf, (size_t)m.bytes, ncclInt8, m.peer, comm, st)); *moved += m.bytes; }
    for (const Msg &m : recvs) NCCL_OK(ncclRecv(m.buf, (size_t)m.bytes, ncclInt8, m.peer, comm, st));
    NCCL_OK(ncclGroupEnd());
    return 0;
}

// an all-gathered buffer pair: the status records (far == false), or the far outboxes of the transfer phase
int gather(const std::vector<Slab> &local, int world, bool loopback, bool far, ncclComm_t comm, hipStream_t st)
{
    auto out_of = [&](const Slab &s) { return far ? s.b.far_out : s.b.status_out; };
    auto in_of = [&](const Slab &s) { return far ? s.b.far_in : s.b.status_in; };
    const size_t nb = (size_t)(far ? local[0].b.far_bytes : local[0].b.status_bytes);
    if (world == 1 || !nb) return 0;
    if (!loopback) {
        NCCL_OK(ncclAllGather(out_of(local[0]), in_of(local[0]), nb, ncclInt8, comm, st));
        return 0;
    }
    // a communicator of one rank: its all-gather is a copy; every slab's record into every slab's block
    for (const Slab &src : local)
        for (const Slab &dst : local)
            NCCL_OK(ncclAllGather(out_of(src), (char *)in_of(dst) + (size_t)src.rank * nb, nb, ncclInt8, comm, st));
    return 0;
}

struct Particle72 { unsigned char bytes[72]; };

}  // namespace

int main(int argc, char **argv)
{
    int world = 2, rank = 0, iters = 8;
    int64_t n = 60000;
    uint32_t seed = 2026;
    bool loopback = false, id_only = false;
    uint64_t job = 0;
    std::string id_file;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--world") world = std::atoi(next());
        else if (a == "--rank") rank = std::atoi(next());
        else if (a == "--iters") iters = std::atoi(next());
        else if (a == "--n") n = std::atoll(next());
        else if (a == "--seed") seed = (uint32_t)std::atoll(next());
        else if (a == "--id-file") id_file = next();
        else if (a == "--loopback") loopback = true;
        else if (a == "--job") job = (uint64_t)std::strtoull(next(), nullptr, 10);
        else if (a == "--id-only") id_only = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (world < 1 || rank < 0 || rank >= world || (!loopback && world > 1 && id_file.empty())) {
        std::fprintf(stderr, "usage: ps_ring_rccl --world W (--loopback | --rank r --id-file F) [--n N] [--iters K] [--seed S]\n");
        return 2;
    }
    const int device = loopback ? 0 : rank;
    hipStream_t st = nullptr;
    if (!id_only) {
        HIP_OK(hipSetDevice(device));
        HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    }

    // the communicator: one rank per process
    ncclComm_t comm;
    ncclUniqueId id;
    const int comm_world = loopback ? 1 : world, comm_rank = loopback ? 0 : rank;
    // The id file carries the job's nonce (--job, the same on every rank of one job) in front of the id: a file
    // left behind by an earlier job is not this job's and is waited past, not read.  Rank 0 removes whatever is
    // there before it writes (tmp + rename: never a half-written file) and again once the communicator is up.
    struct IdFile { uint64_t magic, job; ncclUniqueId id; };
    const uint64_t kMagic = 0x70735f72696e6731ull;        // "ps_ring1"
    if (comm_rank == 0) {
        if (id_only) { uint64_t x = job * 0x9E3779B97F4A7C15ull + 1; for (size_t i = 0; i < sizeof id; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; ((char *)&id)[i] = (char)x; } }
        else NCCL_OK(ncclGetUniqueId(&id));
        if (!id_file.empty()) {
            std::remove(id_file.c_str());
            IdFile rec{kMagic, job, id};
            std::ofstream f(id_file + ".tmp", std::ios::binary);
            f.write((const char *)&rec, sizeof rec);
            f.close();
            if (!f || std::rename((id_file + ".tmp").c_str(), id_file.c_str()) != 0) { std::fprintf(stderr, "cannot write %s\n", id_file.c_str()); return 1; }
        }
    } else {
        for (int tries = 0;; tries++) {
            IdFile rec{};
            std::ifstream f(id_file, std::ios::binary);
            if (f && f.read((char *)&rec, sizeof rec) && rec.magic == kMagic && rec.job == job) { id = rec.id; break; }
            if (tries > 600) { std::fprintf(stderr, "no communicator id of job %llu in %s\n", (unsigned long long)job, id_file.c_str()); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    if (id_only) {                      // (test hook: the rendezvous through the file, without a GPU or RCCL transport)
        unsigned sum = 0;
        for (size_t i = 0; i < sizeof id; i++) sum = sum * 131u + (unsigned char)((const char *)&id)[i];
        std::printf("rank %d of %d: communicator id %08x (job %llu)\n", rank, world, sum, (unsigned long long)job);
        return 0;
    }
    NCCL_OK(ncclCommInitRank(&comm, comm_world, id, comm_rank));
    g_comm = comm;
    if (comm_rank == 0 && !id_file.empty()) std::remove(id_file.c_str());      // every rank has joined: the file has served

    // the slabs this process holds, all shown the same particles (each keeps its own segments')
    std::vector<Slab> local;
    std::vector<float> xyz((size_t)3 * n), age((size_t)n), fert((size_t)n);
    for (int r = 0; r < world; r++) {
        if (!loopback && r != rank) continue;
        Slab s; s.rank = r;
        psamd_config cfg;
        psamd_default_config(&cfg);
        cfg.device = device; cfg.rank = r; cfg.world = world;
        psamd_ctx *ctx = nullptr;
        PS_OK(ctx, psamd_create(&cfg, &ctx));
        s.ctx = ctx;
        if (local.empty()) {
            PS_OK(ctx, psamd_uniform_cloud(ctx, n, seed, xyz.data()));
            uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
            for (int64_t i = 0; i < n; i++) {            // ages of adults, no births (a tag as fertility age)
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                age[(size_t)i] = 15.0f / 7.0f + (7.5f - 15.0f / 7.0f) * (float)((x >> 40) * (1.0 / 16777216.0));
                fert[(size_t)i] = 1.0e6f + (float)i;
            }
        }
        PS_OK(ctx, psamd_fill_particles(ctx, n, xyz.data(), nullptr, nullptr, age.data(), fert.data(), nullptr, nullptr));
        PS_OK(ctx, psamd_set_stream(ctx, (void *)st));
        PS_OK(ctx, psamd_slab_buffers_get(ctx, &s.b));
        PS_OK(ctx, psamd_get_slab_plan(ctx, &s.plan));
        local.push_back(s);
    }

    int64_t moved = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++) {      // DoParallelProcess, ps.cpp:1843-1928, one slab per GPU
        for (Slab &s : local) PS_OK(s.ctx, psamd_slab_build(s.ctx));
        if (exchange(local, world, loopback, HALO, comm, st, &moved)) return 1;
        if (gather(local, world, loopback, false, comm, st)) return 1;
        for (Slab &s : local) PS_OK(s.ctx, psamd_slab_pairs(s.ctx));
        if (exchange(local, world, loopback, FORCE, comm, st, &moved)) return 1;
        for (Slab &s : local) PS_OK(s.ctx, psamd_slab_apply(s.ctx));
        if (exchange(local, world, loopback, XFER, comm, st, &moved)) return 1;
        if (gather(local, world, loopback, true, comm, st)) return 1;        // (births on, four or more ranks)
        for (Slab &s : local) PS_OK(s.ctx, psamd_slab_finish(s.ctx));
    }
    for (Slab &s : local) PS_OK(s.ctx, psamd_synchronize(s.ctx));
    HIP_OK(hipStreamSynchronize(st));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    psamd_sizes sz;
    PS_OK(local[0].ctx, psamd_get_sizes(local[0].ctx, &sz));
    std::vector<Particle72> merged((size_t)sz.container_size), part((size_t)sz.container_size);
    std::memset(merged.data(), 0, merged.size() * sizeof(Particle72));
    int64_t live = 0;
    for (Slab &s : local) {
        PS_OK(s.ctx, psamd_download_particles(s.ctx, part.data(), 0, sz.container_size));
        for (int t = 0; t < 4; t++)
            std::memcpy(merged.data() + s.plan.slot_lo[t], part.data() + s.plan.slot_lo[t],
                        (size_t)(s.plan.slot_hi[t] - s.plan.slot_lo[t]) * sizeof(Particle72));
        int64_t l = 0;
        PS_OK(s.ctx, psamd_live_count(s.ctx, &l));
        live += l;
    }
    std::printf("rank %d of %d%s: %d steps, %.1f MB through RCCL, %.3f ms per step, %lld live here\n", rank, world,
                loopback ? " (all slabs in this process)" : "", iters, moved / 1e6, 1e3 * secs / std::max(1, iters), (long long)live);

    int rc = 0;
    if (loopback) {
        // the same steps in one plain context: the union of the slabs must be its state
        psamd_config cfg;
        psamd_default_config(&cfg);
        cfg.device = 0;
        psamd_ctx *one = nullptr;
        PS_OK(one, psamd_create(&cfg, &one));
        PS_OK(one, psamd_fill_particles(one, n, xyz.data(), nullptr, nullptr, age.data(), fert.data(), nullptr, nullptr));
        PS_OK(one, psamd_step(one, iters));
        PS_OK(one, psamd_download_particles(one, part.data(), 0, sz.container_size));
        psamd_counters cn;
        PS_OK(one, psamd_get_counters(one, &cn));
        // free records: a slab reports the slots it does not own as free records, the merge above took owned ranges only
        size_t bad = 0;
        for (size_t i = 0; i < merged.size(); i++)
            if (std::memcmp(&merged[i], &part[i], sizeof(Particle72)) != 0) bad++;
        std::printf("ring-rccl %s: %zu of %lld records differ from the single context after %d steps (%lld relocations there)\n",
                    bad ? "MISMATCH" : "ok", bad, (long long)sz.container_size, iters, (long long)cn.relocations);
        rc = bad ? 1 : 0;
        psamd_destroy(one);
    }
    for (Slab &s : local) psamd_destroy(s.ctx);
    g_comm = nullptr;
    ncclCommDestroy(comm);
    (void)hipStreamDestroy(st);
    return rc;
}
