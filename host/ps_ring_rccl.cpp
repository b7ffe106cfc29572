// ps_ring_rccl.cpp -- the multi-GPU step in C++: libpsamd.so's four slab stage calls with the
// messages moved by RCCL (ncclSend / ncclRecv / ncclAllGather) straight between the contexts'
// device buffers.  No Python, no torch: include/psamd.h, the HIP runtime and rccl.h.  This is
// what the reference's pmlib layer does for it between nodes (subscriptions to segments,
// ps.cpp:380-487; the stage loop, ps.cpp:1843-1928); here one process drives one GPU.
//
//   one process per GPU (what a node runs; bench.py --gpus N starts these):
//       ps_ring_rccl --world W --rank r --device d --id-file /tmp/id --job J [--bench ...]
//     rank 0 writes the communicator's ncclUniqueId to the file (tagged with the job's nonce J), the others wait for it.
//   all slabs in ONE process on GPU 0 (what a one-GPU test box can run):
//       ps_ring_rccl --world W --loopback [--n N] [--iters K] [--seed S] [--all-pairs] [--births]
//     The communicator has a single rank; every message is an ncclSend to self matched by
//     an ncclRecv from self in the same group -- RCCL moves every byte, between the buffers of
//     different contexts.  This mode also runs the whole system in one plain context and
//     requires the union of the slabs to equal it byte for byte (P_DATA_TYPE of every slot).
//
// Two HIP streams.  The stage kernels run on the COMPUTE stream -- with --graphs 1 each stage's kernels as
// one captured hipGraph, so a rank's step is five submissions, not two dozen launches (measured: a graph
// launch costs ~10 us on the GPU's timeline, the plain launches of a host that runs ahead cost nothing:
// profiles/r4_ab_graphs.txt; off by default) --, every RCCL call on the TRANSFER stream; events order the two: the halo (and the all-pairs snapshot all-gather)
// waits for slab_build and travels while the compute stream runs the interior pair pass (--overlap-interior)
// or simply goes ahead; the all-gather of the status records lands before the first pair-stage call; force and
// transfer messages fork off after slab_pairs / slab_apply and are joined before the stage that reads them.
//
// Message routes (particlesystem_amd/slab.py says the same in Python): after slab_build the
// halo snapshots (rank r's halo_out[above] -> rank r+1's halo_in[below]; halo_out[below] ->
// rank r-1's halo_in[above]), the all-gather of the status records and -- all-pairs forces -- of the
// snapshot blocks; after slab_pairs the force records of lent layers (force_out -> rank r-1's force_in);
// after slab_apply the particles that change owner, on the ring (xfer_out[below] -> rank (r-1)%W's
// xfer_in[above], xfer_out[above] -> rank (r+1)%W's xfer_in[below]; hop-two and far outboxes where the
// plan has them).  All sizes are fixed by the plan; a message of 0 bytes does not exist.  Between one
// pair of ranks RCCL matches sends and receives by order, so both sides post them ordered by
// (peer, hop, direction of travel).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cctype>
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
enum Gather { G_STATUS, G_SNAPSHOT, G_FAR };

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

struct Ring {
    int world = 1;
    bool loopback = false, overlap_interior = false;
    int side = 0;                       // 0: every RCCL call on the compute stream (default); 1: what has compute to travel beside goes on the transfer stream; 2: everything does (round 4)
    ncclComm_t comm = nullptr;
    bool compute_owned = true;          // (false: the compute stream is the first context's own)
    hipStream_t compute = nullptr, transfer = nullptr;      // (transfer == compute without --no-side-stream's opposite)
    hipEvent_t ev_built = nullptr, ev_halo = nullptr, ev_paired = nullptr, ev_force = nullptr, ev_applied = nullptr, ev_xfer = nullptr;
    std::vector<Slab> local;            // the slabs this process holds (one per process in a real run; all of them in loopback mode, where every peer is comm rank 0)
    int64_t moved = 0;
    // Stage timing on the compute stream (benchmark runs, every n-th step): per local slab eight events -- before / after
    // each of the four stage calls.  after(stage k) -> before(stage k + 1) is what the compute stream spent WAITING for
    // the phase's messages (the host enqueues the next stage at once; only the event of the transfer stream holds it).
    std::vector<std::vector<hipEvent_t>> stage_ev;      // [timed step][slab * 8 + e]
    int stage_slot = -1;                                // the timed step being recorded, -1: this step carries no events
};

int mark(Ring &R, size_t slab, int e)
{
    if (R.stage_slot < 0) return 0;
    HIP_OK(hipEventRecord(R.stage_ev[(size_t)R.stage_slot][slab * 8 + (size_t)e], R.compute));
    return 0;
}

// Both ends of every message must agree on its size BEFORE the first step: RCCL matches a send and a receive by order
// alone, and two neighbours that disagree (different halo_cap_cell / xfer_cap / plans) would sit in the transfer until the
// watchdog ends them.  Every rank's sizes are all-gathered once and checked against the routes.
struct SizeTable { int64_t halo_out[2], halo_in[2], force_out, force_in, xfer, xfer2, far, status, allg, pad[5]; };
static_assert(sizeof(SizeTable) == 16 * sizeof(int64_t), "sixteen words");
SizeTable sizes_of(const Slab &s)
{
    SizeTable t{};
    for (int k = 0; k < 2; k++) { t.halo_out[k] = s.b.halo_out_bytes[k]; t.halo_in[k] = s.b.halo_in_bytes[k]; }
    t.force_out = s.b.force_out_bytes; t.force_in = s.b.force_in_bytes;
    t.xfer = s.b.xfer_bytes; t.xfer2 = s.b.xfer2_bytes; t.far = s.b.far_bytes; t.status = s.b.status_bytes; t.allg = s.b.allg_bytes;
    return t;
}
int sizes_agree(const std::vector<SizeTable> &all)
{
    const int W = (int)all.size();
    auto bad = [&](const char *what, int a, int b, long long x, long long y) {
        std::fprintf(stderr, "message sizes disagree: %s of rank %d is %lld bytes, rank %d expects %lld (same halo_cap_cell / xfer_cap / cuts on every rank?)\n", what, a, x, b, y);
        return 1;
    };
    for (int r = 0; r < W; r++) {
        if (r + 1 < W) {
            if (all[r].halo_out[1] != all[r + 1].halo_in[0]) return bad("halo_out[above]", r, r + 1, all[r].halo_out[1], all[r + 1].halo_in[0]);
            if (all[r + 1].halo_out[0] != all[r].halo_in[1]) return bad("halo_out[below]", r + 1, r, all[r + 1].halo_out[0], all[r].halo_in[1]);
            if (all[r + 1].force_out != all[r].force_in) return bad("force_out", r + 1, r, all[r + 1].force_out, all[r].force_in);
        }
        if (all[r].xfer != all[0].xfer) return bad("xfer", r, 0, all[r].xfer, all[0].xfer);
        if (all[r].xfer2 != all[0].xfer2) return bad("xfer2", r, 0, all[r].xfer2, all[0].xfer2);
        if (all[r].far != all[0].far) return bad("far", r, 0, all[r].far, all[0].far);
        if (all[r].status != all[0].status) return bad("status", r, 0, all[r].status, all[0].status);
        if (all[r].allg != all[0].allg) return bad("allg", r, 0, all[r].allg, all[0].allg);
    }
    return 0;
}
int check_sizes(Ring &R)
{
    std::vector<SizeTable> all((size_t)R.world);
    if (R.loopback) { for (const Slab &s : R.local) all[(size_t)s.rank] = sizes_of(s); return sizes_agree(all); }
    if (R.world == 1) return 0;
    SizeTable mine = sizes_of(R.local[0]), *d = nullptr;
    HIP_OK(hipMalloc((void **)&d, sizeof(SizeTable) * ((size_t)R.world + 1)));
    HIP_OK(hipMemcpyAsync(d + R.world, &mine, sizeof mine, hipMemcpyHostToDevice, R.transfer));
    NCCL_OK(ncclAllGather(d + R.world, d, sizeof(SizeTable), ncclInt8, R.comm, R.transfer));
    HIP_OK(hipMemcpyAsync(all.data(), d, sizeof(SizeTable) * (size_t)R.world, hipMemcpyDeviceToHost, R.transfer));
    HIP_OK(hipStreamSynchronize(R.transfer));
    (void)hipFree(d);
    return sizes_agree(all);
}

// One phase's messages as ONE RCCL group on stream `st`.
int exchange(Ring &R, Phase ph, hipStream_t st)
{
    std::vector<Msg> sends, recvs;
    if (R.loopback) {
        // k-th receive from self = k-th send to self: enumerate the routes once for both lists
        for (const Slab &s : R.local)
            for (const Msg &m : sends_of(s, R.world, ph)) {
                const Msg r = recv_of(R.local[(size_t)m.peer], ph, s.rank, m.dir, m.hop);
                if (r.bytes != m.bytes) { std::fprintf(stderr, "message size mismatch %lld vs %lld\n", (long long)m.bytes, (long long)r.bytes); return 1; }
                sends.push_back({m.buf, m.bytes, 0, m.dir, m.hop});
                recvs.push_back({r.buf, r.bytes, 0, r.dir, r.hop});
            }
    } else {
        const Slab &s = R.local[0];
        const int r = s.rank, world = R.world;
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
    for (const Msg &m : sends) { NCCL_OK(ncclSend(m.buf, (size_t)m.bytes, ncclInt8, m.peer, R.comm, st)); R.moved += m.bytes; }
    for (const Msg &m : recvs) NCCL_OK(ncclRecv(m.buf, (size_t)m.bytes, ncclInt8, m.peer, R.comm, st));
    NCCL_OK(ncclGroupEnd());
    return 0;
}

// an all-gathered buffer pair: the status records, the snapshot blocks of an all-pairs run (between slab_build and
// slab_pairs: SURVEY 8(e)'s "all-gather of positions once per step"), or the far outboxes of the transfer phase
int gather(Ring &R, Gather what, hipStream_t st)
{
    auto out_of = [&](const Slab &s) { return what == G_FAR ? s.b.far_out : what == G_SNAPSHOT ? s.b.allg_out : s.b.status_out; };
    auto in_of = [&](const Slab &s) { return what == G_FAR ? s.b.far_in : what == G_SNAPSHOT ? s.b.allg_in : s.b.status_in; };
    const Slab &s0 = R.local[0];
    const size_t nb = (size_t)(what == G_FAR ? s0.b.far_bytes : what == G_SNAPSHOT ? s0.b.allg_bytes : s0.b.status_bytes);
    if (R.world == 1 || !nb) return 0;
    if (!R.loopback) {
        NCCL_OK(ncclAllGather(out_of(s0), in_of(s0), nb, ncclInt8, R.comm, st));
        R.moved += (int64_t)nb;
        return 0;
    }
    // a communicator of one rank: its all-gather is a copy; every slab's record into every slab's block
    for (const Slab &src : R.local)
        for (const Slab &dst : R.local)
            NCCL_OK(ncclAllGather(out_of(src), (char *)in_of(dst) + (size_t)src.rank * nb, nb, ncclInt8, R.comm, st));
    R.moved += (int64_t)nb * (int64_t)R.local.size();
    return 0;
}

// `later` waits for everything enqueued on `earlier` so far (nothing to do when they are one stream)
int order(Ring &R, hipStream_t earlier, hipEvent_t ev, hipStream_t later)
{
    if (earlier == later) return 0;
    HIP_OK(hipEventRecord(ev, earlier));
    HIP_OK(hipStreamWaitEvent(later, ev, 0));
    return 0;
}

// One step of the stage loop (DoParallelProcess, ps.cpp:1843-1928), one slab per GPU.  between(stage): a hook the
// benchmark's frame census uses to read counts back between two stages (nullptr: none).
template <typename Hook>
int ring_step(Ring &R, Hook between)
{
    // Which stream a message travels on.  A dependency that crosses streams costs the GPU's timeline ~15 us each way here
    // (measured, round 5: with every phase on the transfer stream a rank with NO messages at all spent 29 us per phase
    // between two stage kernels -- 88 us of a 720-us rank-step at eight ranks), and pays only where there is compute to
    // travel beside: the halo beside the interior pass, when that is asked for.  The status records must be in before the
    // FIRST pair-stage call (its chunk census decides which particles the stage leaves alone), force and transfer messages
    // before the stage behind them: nothing to travel beside, they go on the compute stream -- an RCCL kernel between two
    // stage kernels, no event.  The default (--side-stream 0) puts EVERYTHING there; 1 is for runs that overlap the halo
    // with the interior pass (--overlap-interior), 2 is round 4's form (every message on the transfer stream), kept for
    // comparison.
    const bool one = R.world == 1;
    hipStream_t s_halo = (R.side == 2 || (R.side == 1 && R.overlap_interior)) ? R.transfer : R.compute;
    hipStream_t s_late = R.side == 2 ? R.transfer : R.compute;      // force, transfer, far outboxes
    for (size_t i = 0; i < R.local.size(); i++) { Slab &s = R.local[i]; if (mark(R, i, 0)) return 1; PS_OK(s.ctx, psamd_slab_build(s.ctx)); if (mark(R, i, 1)) return 1; }
    if (between(0)) return 1;
    if (!one) {
        if (s_halo != R.compute) if (order(R, R.compute, R.ev_built, R.transfer)) return 1;
        if (gather(R, G_STATUS, s_halo)) return 1;                     // first: a 16-KB all-gather, and the interior pass waits for nothing else
        if (s_halo != R.compute) HIP_OK(hipEventRecord(R.ev_force, R.transfer));
        if (exchange(R, HALO, s_halo)) return 1;
        if (gather(R, G_SNAPSHOT, s_halo)) return 1;                   // all-pairs forces only
        if (s_halo != R.compute) HIP_OK(hipEventRecord(R.ev_halo, R.transfer));
    }
    if (R.overlap_interior) {
        if (!one && s_halo != R.compute) HIP_OK(hipStreamWaitEvent(R.compute, R.ev_force, 0));
        for (Slab &s : R.local) PS_OK(s.ctx, psamd_slab_pairs_interior(s.ctx));      // cells whose stencil lies in the own layers: no halo needed
    }
    if (!one && s_halo != R.compute) HIP_OK(hipStreamWaitEvent(R.compute, R.ev_halo, 0));
    for (size_t i = 0; i < R.local.size(); i++) { Slab &s = R.local[i]; if (mark(R, i, 2)) return 1; PS_OK(s.ctx, psamd_slab_pairs(s.ctx)); if (mark(R, i, 3)) return 1; }
    if (between(1)) return 1;
    if (!one) {
        if (s_late != R.compute) { if (order(R, R.compute, R.ev_paired, R.transfer)) return 1; }
        if (exchange(R, FORCE, s_late)) return 1;
        if (s_late != R.compute) { if (order(R, R.transfer, R.ev_force, R.compute)) return 1; }
    }
    for (size_t i = 0; i < R.local.size(); i++) { Slab &s = R.local[i]; if (mark(R, i, 4)) return 1; PS_OK(s.ctx, psamd_slab_apply(s.ctx)); if (mark(R, i, 5)) return 1; }
    if (!one) {
        // (the transfer messages may have grown: every rank adopts the capacity all of them agreed on two steps ago in the same step)
        for (Slab &s : R.local) PS_OK(s.ctx, psamd_slab_buffers_get(s.ctx, &s.b));
        if (s_late != R.compute) { if (order(R, R.compute, R.ev_applied, R.transfer)) return 1; }
        if (exchange(R, XFER, s_late)) return 1;
        if (gather(R, G_FAR, s_late)) return 1;                        // (births on, four or more ranks)
        if (s_late != R.compute) { if (order(R, R.transfer, R.ev_xfer, R.compute)) return 1; }
    }
    for (size_t i = 0; i < R.local.size(); i++) { Slab &s = R.local[i]; if (mark(R, i, 6)) return 1; PS_OK(s.ctx, psamd_slab_finish(s.ctx)); if (mark(R, i, 7)) return 1; }
    return 0;
}
int no_hook(int) { return 0; }

// the shader clock while a timed region runs (sysfs pp_dpm_sclk of the HIP device's PCI function, the level marked current):
// the chip is power-bound under this load, and which clock a figure was taken at is part of the figure
struct ClockWatch {
    std::string path;
    std::vector<int> samples;
    std::atomic<bool> stop{false};
    std::thread th;
    int period_ms = 10;
    explicit ClockWatch(int device, int period = 10) : period_ms(period)
    {
        char bus[64] = {0};
        if (period_ms <= 0) return;
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) == hipSuccess) {
            for (char *c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
            path = std::string("/sys/bus/pci/devices/") + bus + "/pp_dpm_sclk";
            std::ifstream f(path);
            if (!f) path.clear();
        }
    }
    void start()
    {
        if (path.empty()) return;
        stop = false;
        th = std::thread([this]() {
            while (!stop) {
                std::ifstream f(path);
                std::string line;
                while (std::getline(f, line)) {
                    if (line.find('*') == std::string::npos) continue;
                    const size_t c = line.find(':');
                    int v = 0;
                    for (size_t i = c == std::string::npos ? 0 : c + 1; i < line.size(); i++) if (std::isdigit((unsigned char)line[i])) v = v * 10 + (line[i] - '0');
                    if (v) samples.push_back(v);
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
            }
        });
    }
    void end() { if (th.joinable()) { stop = true; th.join(); } }
    std::string json()
    {
        if (samples.empty()) return "null";
        std::vector<int> v = samples;
        std::sort(v.begin(), v.end());
        char b[256];
        std::snprintf(b, sizeof b, "{\"min\": %d, \"median\": %d, \"max\": %d, \"samples\": %zu, \"source\": \"%s\"}", v.front(), v[v.size() / 2], v.back(), v.size(), path.c_str());
        return b;
    }
};

struct Particle72 { unsigned char bytes[72]; };

// force terms one pair pass evaluates: per visited particle its stencil's population (27 cells, not periodic:
// app.cu:352-409), or -- all-pairs -- every listed body
double force_terms(const std::vector<int64_t> &n, const std::vector<int32_t> &f, int G, bool all_pairs)
{
    double total = 0;
    if (all_pairs) {
        double sn = 0, sf = 0;
        for (size_t i = 0; i < n.size(); i++) { sn += (double)n[i]; sf += (double)f[i]; }
        return sn * sf;
    }
    for (int i3 = 0; i3 < G; i3++) for (int i1 = 0; i1 < G; i1++) for (int i2 = 0; i2 < G; i2++) {
        const int fc = f[(size_t)(i3 * G + i1) * G + i2];
        if (!fc) continue;
        int64_t nb = 0;
        for (int a = -1; a <= 1; a++) for (int b = -1; b <= 1; b++) for (int d = -1; d <= 1; d++) {
            const int j3 = i3 + a, j1 = i1 + b, j2 = i2 + d;
            if (j3 < 0 || j3 >= G || j1 < 0 || j1 >= G || j2 < 0 || j2 >= G) continue;
            nb += n[(size_t)(j3 * G + j1) * G + j2];
        }
        total += (double)fc * (double)nb;
    }
    return total;
}

}  // namespace

int main(int argc, char **argv)
{
    int world = 2, rank = 0, iters = 8, device = -1;
    int64_t n = 60000;
    uint32_t seed = 2026;
    bool loopback = false, id_only = false, all_pairs = false, births = false, graphs = false, bench = false, evolve = false;
    bool overlap_interior = false, fast_math = false, launch_check = false, break_sizes = false;
    int side_stream = 0;
    int steps = 200, warmup = 5, chunk_factor = 4, chunk_dim = 4, halo_cap_cell = 0, xfer_cap = 0, timing_period = 8, wait_policy = -1, sustained_steps = 0, clock_ms = 10;
    double settle_seconds = 0.5;
    int64_t max_particles = 0;
    uint64_t job = 0;
    std::string id_file;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--world") world = std::atoi(next());
        else if (a == "--rank") rank = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--iters") iters = std::atoi(next());
        else if (a == "--n") n = std::atoll(next());
        else if (a == "--seed") seed = (uint32_t)std::atoll(next());
        else if (a == "--id-file") id_file = next();
        else if (a == "--loopback") loopback = true;
        else if (a == "--job") job = (uint64_t)std::strtoull(next(), nullptr, 10);
        else if (a == "--id-only") id_only = true;
        else if (a == "--launch-check") launch_check = true;
        else if (a == "--test-size-mismatch") break_sizes = true;      // (test hook: rank 1 is created with other message sizes than its neighbours expect)
        else if (a == "--all-pairs") all_pairs = true;
        else if (a == "--births") births = true;
        else if (a == "--fast-math") fast_math = true;
        else if (a == "--graphs") graphs = std::atoi(next()) != 0;
        else if (a == "--side-stream") side_stream = std::max(0, std::min(2, std::atoi(next())));
        else if (a == "--overlap-interior") overlap_interior = true;
        else if (a == "--wait") wait_policy = std::atoi(next());
        else if (a == "--bench") bench = true;
        else if (a == "--evolve") evolve = true;
        else if (a == "--steps") steps = std::atoi(next());
        else if (a == "--warmup") warmup = std::atoi(next());
        else if (a == "--settle-seconds") settle_seconds = std::atof(next());
        else if (a == "--timing-period") timing_period = std::max(1, std::atoi(next()));
        else if (a == "--clock-period-ms") clock_ms = std::atoi(next());      // how often the shader clock is sampled while a timed region runs (0: not at all)
        else if (a == "--sustained-steps") sustained_steps = std::max(0, std::atoi(next()));
        else if (a == "--chunk-factor") chunk_factor = std::atoi(next());
        else if (a == "--chunk-dim") chunk_dim = std::atoi(next());
        else if (a == "--halo-cap-cell") halo_cap_cell = std::atoi(next());
        else if (a == "--xfer-cap") xfer_cap = std::atoi(next());
        else if (a == "--max-particles") max_particles = std::atoll(next());
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (world < 1 || rank < 0 || rank >= world || (!loopback && world > 1 && id_file.empty())) {
        std::fprintf(stderr, "usage: ps_ring_rccl --world W (--loopback | --rank r --id-file F --job J [--device d]) [--n N] [--iters K] [--seed S] "
                             "[--all-pairs] [--births] [--graphs 0|1] [--side-stream 0|1|2] [--overlap-interior] [--bench --steps K --warmup W ...]\n");
        return 2;
    }
    if (psamd_abi_version() != PSAMD_ABI_VERSION) {
        std::fprintf(stderr, "libpsamd.so has ABI version %d, this program was built against %d: rebuild one of them\n", psamd_abi_version(), PSAMD_ABI_VERSION);
        return 2;
    }
    if (device < 0) device = loopback ? 0 : rank;
    const bool no_gpu = id_only || launch_check;
    Ring R;
    R.world = world; R.loopback = loopback; R.side = side_stream; R.overlap_interior = overlap_interior;
    if (!no_gpu) {
        HIP_OK(hipSetDevice(device));
        HIP_OK(hipStreamCreateWithFlags(&R.compute, hipStreamNonBlocking));
        if (side_stream) HIP_OK(hipStreamCreateWithFlags(&R.transfer, hipStreamNonBlocking));
        else R.transfer = R.compute;
        for (hipEvent_t *e : {&R.ev_built, &R.ev_halo, &R.ev_paired, &R.ev_force, &R.ev_applied, &R.ev_xfer}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }

    // the communicator: one rank per process
    ncclUniqueId id;
    const int comm_world = loopback ? 1 : world, comm_rank = loopback ? 0 : rank;
    // The id file carries the job's nonce (--job, the same on every rank of one job) in front of the id: a file
    // left behind by an earlier job is not this job's and is waited past, not read.  Rank 0 removes whatever is
    // there before it writes (tmp + rename: never a half-written file) and again once the communicator is up.
    struct IdFile { uint64_t magic, job; ncclUniqueId id; };
    const uint64_t kMagic = 0x70735f72696e6731ull;        // "ps_ring1"
    if (comm_rank == 0) {
        if (no_gpu) { uint64_t x = job * 0x9E3779B97F4A7C15ull + 1; for (size_t i = 0; i < sizeof id; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; ((char *)&id)[i] = (char)x; } }
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
    if (no_gpu) {                      // (test hooks: the rendezvous through the file, without a GPU or RCCL transport)
        unsigned sum = 0;
        for (size_t i = 0; i < sizeof id; i++) sum = sum * 131u + (unsigned char)((const char *)&id)[i];
        if (id_only) { std::printf("rank %d of %d: communicator id %08x (job %llu)\n", rank, world, sum, (unsigned long long)job); return 0; }
        // --launch-check: every rank reports in through a file beside the id file; rank 0 waits for all of them and
        // prints the skeleton of the benchmark record -- proves bench.py's launcher of C++ ranks on a machine without GPUs
        const std::string mine = id_file + ".in" + std::to_string(rank);
        { std::ofstream f(mine, std::ios::binary); f << sum << " " << job << "\n"; }
        if (rank == 0) {
            int seen = 0;
            for (int tries = 0; tries < 600 && seen < world; tries++) {
                seen = 0;
                for (int r = 0; r < world; r++) {
                    std::ifstream f(id_file + ".in" + std::to_string(r));
                    unsigned s2 = 0; unsigned long long j2 = 0;
                    if (f && (f >> s2 >> j2) && s2 == sum && j2 == job) seen++;
                }
                if (seen < world) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            for (int r = 0; r < world; r++) std::remove((id_file + ".in" + std::to_string(r)).c_str());
            std::remove(id_file.c_str());
            if (seen < world) { std::fprintf(stderr, "launch check: %d of %d ranks reported in\n", seen, world); return 1; }
            std::printf("{\"psamd_ring\": 1, \"launch_check\": true, \"world\": %d, \"steps\": %d, \"warmup\": %d}\n", seen, steps, warmup);
        }
        return 0;
    }
    const bool no_comm = comm_world == 1 && !loopback && std::getenv("PSAMD_RING_NO_RCCL") != nullptr;      // (A/B: a world of one without a communicator)
    if (!no_comm) NCCL_OK(ncclCommInitRank(&R.comm, comm_world, id, comm_rank));
    g_comm = R.comm;
    if (comm_rank == 0 && !id_file.empty()) std::remove(id_file.c_str());      // every rank has joined: the file has served

    // the slabs this process holds, all shown the same particles (each keeps its own segments')
    std::vector<float> xyz((size_t)3 * n), age((size_t)n), fert((size_t)n);
    psamd_config cfg0;
    psamd_default_config(&cfg0);
    cfg0.chunk_factor = chunk_factor; cfg0.chunk_dim = chunk_dim;
    cfg0.max_particles_num = (int32_t)std::max<int64_t>(std::max<int64_t>(n, max_particles), 1 << 20);
    cfg0.flags = (all_pairs ? PSAMD_FLAG_ALL_PAIRS : 0u) | (births ? PSAMD_FLAG_EXPLOSIONS : 0u) | (fast_math ? PSAMD_FLAG_FAST_MATH : 0u);
    cfg0.halo_cap_cell = halo_cap_cell; cfg0.xfer_cap = xfer_cap;
    cfg0.seed = seed;
    const double life = cfg0.life_steps * cfg0.dt;
    for (int r = 0; r < world; r++) {
        if (!loopback && r != rank) continue;
        Slab s; s.rank = r;
        psamd_config cfg = cfg0;
        cfg.device = device; cfg.rank = r; cfg.world = world;
        if (break_sizes && r == 1) cfg.halo_cap_cell = (cfg.halo_cap_cell > 0 ? cfg.halo_cap_cell : 64) + 8;
        psamd_ctx *ctx = nullptr;
        PS_OK(ctx, psamd_create(&cfg, &ctx));
        s.ctx = ctx;
        if (R.local.empty()) {
            PS_OK(ctx, psamd_uniform_cloud(ctx, n, seed, xyz.data()));
            uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
            for (int64_t i = 0; i < n; i++) {            // ages of adults [MIN_ADULT_AGE, MAX_ADULT_AGE); births: fertility ages they reach within a few steps
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const double u = (double)(x >> 40) * (1.0 / 16777216.0);
                age[(size_t)i] = (float)(life / 7.0 + (life / 2.0 - life / 7.0) * u);
                fert[(size_t)i] = births ? (float)(age[(size_t)i] + cfg0.dt * (double)(1 + (x & 15))) : 1.0e6f + (float)(bench ? 0 : i);
            }
        }
        PS_OK(ctx, psamd_fill_particles(ctx, n, xyz.data(), nullptr, nullptr, age.data(), fert.data(), nullptr, nullptr));
        if (R.local.empty()) {
            // Which stream the stage kernels (and, by default, the RCCL calls) run on: the first context's OWN stream.  Measured
            // (profiles/r5_ab_host.txt): with a stream this program created itself -- before the contexts (PSAMD_RING_STREAM=0,
            // what it did until late in round 5) or after the first one (2) -- a one-rank step takes 0.8-1.2 % longer, all of it
            // inside the force pass's own time; on the context's own stream the C++ host is as fast as the Python host.  Eight
            // slabs in one process: no difference.  The cause is not known (same flags, same kernels, same arguments).
            const char *e = std::getenv("PSAMD_RING_STREAM");
            const int mode = e ? std::atoi(e) : 1;
            if (mode >= 1 && mode <= 4) {
                const bool shared = R.transfer == R.compute;
                (void)hipStreamDestroy(R.compute);
                if (mode == 1) { void *st = nullptr; PS_OK(ctx, psamd_get_stream(ctx, &st)); R.compute = (hipStream_t)st; R.compute_owned = false; }
                else if (mode == 2) HIP_OK(hipStreamCreateWithFlags(&R.compute, hipStreamNonBlocking));
                else { int lo = 0, hi = 0; HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi)); HIP_OK(hipStreamCreateWithPriority(&R.compute, hipStreamNonBlocking, mode == 3 ? hi : lo)); }      // (3: highest priority, 4: lowest)
                if (shared) R.transfer = R.compute;
            }
        }
        PS_OK(ctx, psamd_set_stream(ctx, (void *)R.compute));
        PS_OK(ctx, psamd_set_graphs(ctx, graphs ? 1 : 0));
        if (bench) PS_OK(ctx, psamd_set_tdata_mirror(ctx, 0));      // (this host never fetches the reference's T_DATA buffer)
        if (wait_policy >= 0) PS_OK(ctx, psamd_set_wait_policy(ctx, wait_policy));
        PS_OK(ctx, psamd_slab_buffers_get(ctx, &s.b));
        PS_OK(ctx, psamd_get_slab_plan(ctx, &s.plan));
        R.local.push_back(s);
    }
    psamd_sizes sz;
    PS_OK(R.local[0].ctx, psamd_get_sizes(R.local[0].ctx, &sz));
    if (check_sizes(R)) return bail();          // every message has the size its receiver expects, or nobody starts

    // ---------------------------------------------------------------- benchmark protocol (bench.py --gpus N relays the record)
    if (bench) {
        // collectives on host numbers: a device scratch word, RCCL, the transfer stream (a world of one -- loopback -- has nothing to reduce)
        int64_t *d_red = nullptr;
        const size_t red_words = (size_t)sz.num_cells + 8;
        HIP_OK(hipMalloc((void **)&d_red, red_words * sizeof(int64_t)));
        auto sync_all = [&]() -> int {
            for (Slab &s : R.local) PS_OK(s.ctx, psamd_synchronize(s.ctx));
            HIP_OK(hipStreamSynchronize(R.compute));
            HIP_OK(hipStreamSynchronize(R.transfer));
            return 0;
        };
        auto reduce_i64 = [&](int64_t *host, size_t count, ncclRedOp_t op) -> int {
            if (comm_world == 1) return 0;
            HIP_OK(hipMemcpyAsync(d_red, host, count * sizeof(int64_t), hipMemcpyHostToDevice, R.transfer));
            NCCL_OK(ncclAllReduce(d_red, d_red, count, ncclInt64, op, R.comm, R.transfer));
            HIP_OK(hipMemcpyAsync(host, d_red, count * sizeof(int64_t), hipMemcpyDeviceToHost, R.transfer));
            HIP_OK(hipStreamSynchronize(R.transfer));
            return 0;
        };
        auto barrier = [&]() -> int {              // every rank's device work is done, then all ranks meet, then again nothing is in flight
            if (sync_all()) return 1;
            int64_t one = 1;
            if (reduce_i64(&one, 1, ncclSum)) return 1;
            return 0;
        };
        if (!evolve) for (Slab &s : R.local) PS_OK(s.ctx, psamd_snapshot_save(s.ctx));
        auto one_step = [&]() -> int {
            if (!evolve) for (Slab &s : R.local) PS_OK(s.ctx, psamd_snapshot_restore(s.ctx));
            return ring_step(R, no_hook);
        };
        // (The census -- downloads, host work -- comes BEFORE the settling steps and the warmup, so that the warmup runs
        // straight into the timed region: an idle GPU in between cost the first timed steps their clock.)
        // the frame's census: particles per cell (whole system) and particles the force pass visits per cell (own, and whole system)
        std::vector<int32_t> cellgrid((size_t)sz.n_cellgrid), fc((size_t)sz.num_cells);
        std::vector<int64_t> n_cell((size_t)sz.num_cells), f_all((size_t)sz.num_cells);
        std::vector<int32_t> f_own((size_t)sz.num_cells);
        auto census = [&](double *terms_own, int64_t *with_force, int64_t *live) -> int {
            std::fill(n_cell.begin(), n_cell.end(), 0); std::fill(f_own.begin(), f_own.end(), 0);
            if (!evolve) for (Slab &s : R.local) PS_OK(s.ctx, psamd_snapshot_restore(s.ctx));
            const size_t stride = 1 + (size_t)sz.max_per_cell;
            auto hook = [&](int stage) -> int {
                for (Slab &s : R.local) {
                    if (stage == 0) {
                        PS_OK(s.ctx, psamd_download_cellgrid(s.ctx, cellgrid.data()));
                        for (int c = 0; c < sz.num_cells; c++) n_cell[(size_t)c] += cellgrid[stride * (size_t)c];
                    } else {
                        PS_OK(s.ctx, psamd_download_force_counts(s.ctx, fc.data()));
                        if (&s == &R.local[0]) f_own = fc;
                        for (int c = 0; c < sz.num_cells; c++) f_all[(size_t)c] += fc[(size_t)c];
                    }
                }
                return 0;
            };
            std::fill(f_all.begin(), f_all.end(), 0);
            if (ring_step(R, hook)) return 1;
            if (sync_all()) return 1;
            if (reduce_i64(n_cell.data(), n_cell.size(), ncclSum)) return 1;
            if (reduce_i64(f_all.data(), f_all.size(), ncclSum)) return 1;
            *terms_own = force_terms(n_cell, f_own, sz.grid_dim, all_pairs);
            *with_force = 0; *live = 0;
            for (int c = 0; c < sz.num_cells; c++) { *with_force += f_all[(size_t)c]; *live += n_cell[(size_t)c]; }
            return 0;
        };
        double terms0 = 0, terms1 = 0;
        int64_t wf0 = 0, wf1 = 0, live0 = 0, live1 = 0;
        if (!evolve && census(&terms0, &wf0, &live0)) return bail();
        { psamd_ctx *c = R.local[0].ctx; PS_OK(c, psamd_set_timing(c, 1)); PS_OK(c, psamd_set_timing(c, 0)); }      // (the timers' events exist before the timed region)
        // untimed: let the clocks settle; all ranks must take the same number of steps: they decide together, ten at a time
        int settle = 0;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(settle_seconds);
        for (;;) {
            int64_t go = (!evolve && std::chrono::steady_clock::now() < t_end) ? 1 : 0;
            if (reduce_i64(&go, 1, ncclMin)) return bail();
            if (!go) break;
            for (int k = 0; k < 10; k++) if (one_step()) return bail();
            settle += 10;
        }
        for (int k = 0; k < warmup; k++) if (one_step()) return bail();
        if (barrier()) return bail();
        psamd_ctx *c0 = R.local[0].ctx;
        const int period = std::max(1, std::min(timing_period, steps));
        PS_OK(c0, psamd_set_timing_period(c0, period));
        PS_OK(c0, psamd_set_timing(c0, 1));
        // this host's own events around the four stage calls of every local slab, on the same steps
        const size_t n_local = R.local.size();
        R.stage_ev.assign((size_t)((steps + period - 1) / period), std::vector<hipEvent_t>(n_local * 8));
        for (auto &v : R.stage_ev) for (auto &e : v) HIP_OK(hipEventCreate(&e));
        psamd_counters cn0{}, cn1{};
        int64_t processed0 = 0;
        for (Slab &s : R.local) { PS_OK(s.ctx, psamd_get_counters(s.ctx, &cn0)); processed0 += cn0.particles_processed; }
        ClockWatch clock(device, clock_ms), clock2(device, clock_ms);
        if (barrier()) return bail();
        clock.start();
        const auto t0 = std::chrono::steady_clock::now();
        size_t stage_steps = 0;                             // timed steps that carried the stage events
        for (int k = 0; k < steps; k++) {
            // (the stage events go in on other steps than the library's kernel timers: side by side each delays what the other brackets)
            R.stage_slot = k % period == (period > 1 ? period / 2 : 0) ? k / period : -1;
            if (R.stage_slot >= 0) stage_steps = (size_t)R.stage_slot + 1;
            if (one_step()) return bail();
        }
        R.stage_slot = -1;
        if (barrier()) return bail();
        double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        clock.end();
        double us[PSAMD_NUM_TIMERS], us_med[PSAMD_NUM_TIMERS], us_max[PSAMD_NUM_TIMERS];
        int64_t launches = 0;
        PS_OK(c0, psamd_get_timing(c0, us, &launches));
        PS_OK(c0, psamd_get_timing_stats(c0, us_med, us_max, nullptr));
        PS_OK(c0, psamd_set_timing(c0, 0));
        int64_t processed1 = 0;
        for (Slab &s : R.local) { PS_OK(s.ctx, psamd_get_counters(s.ctx, &cn1)); processed1 += cn1.particles_processed; }
        PS_OK(c0, psamd_get_counters(c0, &cn1));
        const int64_t own_updates = processed1 - processed0;
        // a short timed region says little about the clock a long run holds: the same loop again, long enough (no events)
        double sustained_ms = 0.0;
        if (sustained_steps > steps && !evolve) {
            if (barrier()) return bail();
            clock2.start();
            const auto s0 = std::chrono::steady_clock::now();
            for (int k = 0; k < sustained_steps; k++) if (one_step()) return bail();
            if (barrier()) return bail();
            sustained_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count() / sustained_steps;
            clock2.end();
            int64_t ns = (int64_t)(sustained_ms * 1e6);
            if (reduce_i64(&ns, 1, ncclMax)) return bail();
            sustained_ms = (double)ns * 1e-6;
        }
        // stage and wait times: the median over the timed steps per local slab, then every rank's figures on every rank
        // ([world][7]: build pairs apply finish | wait for halo, force, xfer), in nanoseconds through the int64 all-reduce
        // (each rank fills its own rows, the others' are zero: a sum is an all-gather)
        std::vector<int64_t> tab((size_t)world * 7, 0);
        for (size_t i = 0; i < n_local; i++) {
            const int a[7] = {0, 2, 4, 6, 1, 3, 5}, b[7] = {1, 3, 5, 7, 2, 4, 6};
            for (int k = 0; k < 7; k++) {
                std::vector<float> v;
                // (only the steps that recorded them: asking an event that never was recorded for its time fails AND leaves
                // the error behind for the next launch check to find)
                for (size_t q = 0; q < stage_steps; q++) {
                    const auto &evs = R.stage_ev[q];
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, evs[i * 8 + (size_t)a[k]], evs[i * 8 + (size_t)b[k]]) == hipSuccess) v.push_back(ms);
                }
                std::sort(v.begin(), v.end());
                tab[(size_t)R.local[i].rank * 7 + (size_t)k] = v.empty() ? 0 : (int64_t)(1e6 * (double)v[v.size() / 2]);
            }
        }
        if (reduce_i64(tab.data(), tab.size(), ncclSum)) return bail();
        for (auto &v : R.stage_ev) for (auto &e : v) (void)hipEventDestroy(e);
        (void)hipGetLastError();
        if (census(&terms1, &wf1, &live1)) return bail();
        if (evolve) { terms0 = terms1; wf0 = wf1; }
        int64_t red[2] = {own_updates, (int64_t)(elapsed * 1e9)};
        int64_t upd = own_updates;
        if (reduce_i64(&upd, 1, ncclSum)) return bail();
        if (reduce_i64(&red[1], 1, ncclMax)) return bail();
        elapsed = (double)red[1] * 1e-9;
        int64_t gl = 0, gc = 0;
        const int grc = psamd_get_graph_stats(c0, &gl, &gc);
        int rccl_ranks = R.comm ? 0 : 1;
        if (R.comm) NCCL_OK(ncclCommCount(R.comm, &rccl_ranks));
        if (rank == 0 || loopback) {
            static const char *names[PSAMD_NUM_TIMERS] = {"hist", "scan", "scatter", "sort_cells", "pairs", "apply", "lifecycle", "init_iframe", "collide"};
            auto timers = [&](const double *v, double div) {
                std::string kt;
                for (int k = 0; k < PSAMD_NUM_TIMERS; k++)
                    if (us[k] > 0) { char b[96]; std::snprintf(b, sizeof b, "%s\"%s\": %.3f", kt.empty() ? "" : ", ", names[k], v[k] / div); kt += b; }
                return kt;
            };
            const std::string kt = timers(us, (double)std::max<int64_t>(launches, 1)), kt_med = timers(us_med, 1.0), kt_max = timers(us_max, 1.0);
            // per-rank stage times, and the waits as minimum / maximum over the ranks
            static const char *stage_names[4] = {"build", "pairs", "apply", "finish"}, *wait_names[3] = {"halo", "force", "xfer"};
            std::string stages, waits;
            for (int k = 0; k < 4; k++) {
                stages += std::string(k ? ", " : "") + "\"" + stage_names[k] + "\": [";
                for (int r = 0; r < world; r++) { char b[32]; std::snprintf(b, sizeof b, "%s%.4f", r ? ", " : "", (double)tab[(size_t)r * 7 + (size_t)k] * 1e-6); stages += b; }
                stages += "]";
            }
            for (int k = 0; k < 3; k++) {
                int64_t lo = INT64_MAX, hi = 0;
                for (int r = 0; r < world; r++) { lo = std::min(lo, tab[(size_t)r * 7 + 4 + (size_t)k]); hi = std::max(hi, tab[(size_t)r * 7 + 4 + (size_t)k]); }
                char b[96];
                std::snprintf(b, sizeof b, "%s\"%s\": {\"min\": %.4f, \"max\": %.4f}", k ? ", " : "", wait_names[k], (double)lo * 1e-6, (double)hi * 1e-6);
                waits += b;
            }
            const psamd_slab_buffers &b = R.local[0].b;
            std::printf("{\"psamd_ring\": 1, \"world\": %d, \"loopback\": %s, \"rccl_ranks\": %d, \"n\": %lld, \"grid_dim\": %d, \"steps\": %d, \"warmup\": %d, \"settle_steps\": %d, "
                        "\"elapsed_s\": %.9f, \"updates\": %lld, \"own_updates\": %lld, \"live_after\": %lld, \"particles_with_a_force_term\": %lld, "
                        "\"pairs_rank0\": %.6e, \"kernel_us\": {%s}, \"kernel_us_median\": {%s}, \"kernel_us_max\": {%s}, \"timed_launches\": %lld, \"timing_period\": %d, "
                        "\"stage_ms_per_rank\": {%s}, \"wait_ms\": {%s}, "
                        "\"relocations\": %lld, \"relocations_lost\": %lld, \"cell_overflow_kills\": %lld, "
                        "\"message_bytes_rank0\": {\"halo_up\": %lld, \"halo_down\": %lld, \"force_in\": %lld, \"xfer_each\": %lld, \"status\": %lld, \"snapshot_block\": %lld}, "
                        "\"bytes_per_phase_rank0\": {\"halo\": %lld, \"force\": %lld, \"xfer\": %lld, \"gathers\": %lld}, "
                        "\"rccl_mb_rank0\": %.3f, \"graphs\": %s, \"graph_replays\": %lld, \"graph_captures\": %lld, \"side_stream\": %s, \"overlap_interior\": %s, "
                        "\"all_pairs\": %s, \"fast_math\": %s, \"evolve\": %s, \"halo_cap_cell\": %d, \"xfer_cap\": %d, \"side_stream_mode\": %d, "
                        "\"shader_clock_mhz\": %s, \"sustained_steps\": %d, \"sustained_ms_per_step\": %.6f, \"sustained_shader_clock_mhz\": %s}\n",
                        world, loopback ? "true" : "false", rccl_ranks, (long long)n, sz.grid_dim, steps, warmup, settle, elapsed, (long long)upd, (long long)own_updates,
                        (long long)live1, (long long)wf1, 0.5 * (terms0 + terms1), kt.c_str(), kt_med.c_str(), kt_max.c_str(), (long long)launches, period,
                        stages.c_str(), waits.c_str(),
                        (long long)cn1.relocations, (long long)cn1.relocations_lost, (long long)cn1.cell_overflow_kills,
                        (long long)b.halo_out_bytes[1], (long long)b.halo_out_bytes[0], (long long)b.force_in_bytes, (long long)b.xfer_bytes, (long long)b.status_bytes,
                        (long long)b.allg_bytes,
                        (long long)(b.halo_out_bytes[0] + b.halo_out_bytes[1]), (long long)b.force_out_bytes,
                        (long long)(world > 1 ? 2 * b.xfer_bytes + 2 * b.xfer2_bytes : 0), (long long)(world > 1 ? b.status_bytes + b.allg_bytes + b.far_bytes : 0),
                        R.moved / 1e6, (graphs && grc == PSAMD_OK) ? "true" : "false", (long long)gl, (long long)gc,
                        side_stream ? "true" : "false", overlap_interior ? "true" : "false", all_pairs ? "true" : "false", fast_math ? "true" : "false",
                        evolve ? "true" : "false", halo_cap_cell, xfer_cap, side_stream,
                        clock.json().c_str(), sustained_steps > steps ? sustained_steps : 0, sustained_ms, clock2.json().c_str());
            std::fflush(stdout);
        }
        if (barrier()) return bail();
        (void)hipFree(d_red);
        for (size_t i = R.local.size(); i-- > 0;) psamd_destroy(R.local[i].ctx);      // (the first context last: the others enqueue on its stream)
        g_comm = nullptr;
        if (R.comm) ncclCommDestroy(R.comm);
        return 0;
    }

    // ---------------------------------------------------------------- plain run (and, in loopback mode, the check against one context)
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++)
        if (ring_step(R, no_hook)) return bail();
    for (Slab &s : R.local) PS_OK(s.ctx, psamd_synchronize(s.ctx));
    HIP_OK(hipStreamSynchronize(R.compute));
    HIP_OK(hipStreamSynchronize(R.transfer));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<Particle72> merged((size_t)sz.container_size), part((size_t)sz.container_size);
    std::memset(merged.data(), 0, merged.size() * sizeof(Particle72));
    int64_t live = 0, replays = 0, captures = 0;
    for (Slab &s : R.local) {
        PS_OK(s.ctx, psamd_download_particles(s.ctx, part.data(), 0, sz.container_size));
        for (int t = 0; t < 4; t++)
            std::memcpy(merged.data() + s.plan.slot_lo[t], part.data() + s.plan.slot_lo[t],
                        (size_t)(s.plan.slot_hi[t] - s.plan.slot_lo[t]) * sizeof(Particle72));
        int64_t l = 0, a = 0, b = 0;
        PS_OK(s.ctx, psamd_live_count(s.ctx, &l));
        PS_OK(s.ctx, psamd_get_graph_stats(s.ctx, &a, &b));       // (an error here: the runtime refused to capture a stage)
        live += l; replays += a; captures += b;
    }
    std::printf("rank %d of %d%s: %d steps, %.1f MB through RCCL, %.3f ms per step, %lld live here, %lld graph replays (%lld captures)%s%s\n", rank, world,
                loopback ? " (all slabs in this process)" : "", iters, R.moved / 1e6, 1e3 * secs / std::max(1, iters), (long long)live,
                (long long)replays, (long long)captures, side_stream == 2 ? ", every transfer on a second stream" : side_stream ? ", the status gather (and an overlapped halo) on a second stream" : "",
                overlap_interior ? ", interior pass beside the halo" : "");

    int rc = 0;
    if (loopback) {
        // the same steps in one plain context: the union of the slabs must be its state
        psamd_config cfg = cfg0;
        cfg.device = device;
        psamd_ctx *one = nullptr;
        PS_OK(one, psamd_create(&cfg, &one));
        PS_OK(one, psamd_fill_particles(one, n, xyz.data(), nullptr, nullptr, age.data(), fert.data(), nullptr, nullptr));
        PS_OK(one, psamd_set_graphs(one, graphs ? 1 : 0));
        PS_OK(one, psamd_step(one, iters));
        PS_OK(one, psamd_download_particles(one, part.data(), 0, sz.container_size));
        psamd_counters cn;
        PS_OK(one, psamd_get_counters(one, &cn));
        // free records: a slab reports the slots it does not own as free records, the merge above took owned ranges only
        size_t bad = 0;
        for (size_t i = 0; i < merged.size(); i++)
            if (std::memcmp(&merged[i], &part[i], sizeof(Particle72)) != 0) bad++;
        std::printf("ring-rccl %s: %zu of %lld records differ from the single context after %d steps (%lld relocations, %lld births there)\n",
                    bad ? "MISMATCH" : "ok", bad, (long long)sz.container_size, iters, (long long)cn.relocations, (long long)cn.births);
        rc = bad ? 1 : 0;
        psamd_destroy(one);
    }
    for (size_t i = R.local.size(); i-- > 0;) psamd_destroy(R.local[i].ctx);      // (the first context last: the others enqueue on its stream)
    g_comm = nullptr;
    if (R.comm) ncclCommDestroy(R.comm);
    if (R.transfer != R.compute) (void)hipStreamDestroy(R.transfer);
    if (R.compute_owned) (void)hipStreamDestroy(R.compute);
    return rc;
}
