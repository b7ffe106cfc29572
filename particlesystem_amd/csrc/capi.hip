// capi.hip -- the psamd context and its C ABI (include/psamd.h).
//
// Host side of the drop-in boundary, C++ like the reference's host code.  It mirrors
// the reference driver's view of the path: nine buffers (ps.cpp:70-78), one-off setup
// stages, then per step init_iframe -> build_grid -> calc_forces (ps.cpp:1843-1928).
// All arithmetic of the step runs in the HIP kernels of kernels.hip; there is no CPU
// fallback: without a HIP device psamd_create fails with PSAMD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include <sys/prctl.h>

#include "../../include/psamd.h"
#include "device_types.h"
#include "geometry.hpp"
#include "kernels.h"
#include "partition.hpp"

using namespace psamd;

struct psamd_ctx {
    Geometry geo;
    DevParams P{};
    DevParams P_int{}, P_rest{};      // the pair stage cut in two: interior own cells (no halo needed), the rest
    bool have_interior = false, interior_done = false;
    SegLayout S{};
    DeviceState d;
    hipStream_t stream = nullptr;       // stream in use
    hipStream_t own_stream = nullptr;   // the one this context created
    SlabPlan plan;
    // slab messages (device); index 0 = the rank below, 1 = the rank above
    int *halo_out[2] = {nullptr, nullptr}, *halo_in[2] = {nullptr, nullptr};
    size_t halo_out_bytes[2] = {0, 0}, halo_in_bytes[2] = {0, 0};
    int halo_out_c0[2] = {0, 0}, halo_out_cells[2] = {0, 0}, halo_in_cells[2] = {0, 0};
    int *force_out = nullptr, *force_in = nullptr;
    size_t force_out_bytes = 0, force_in_bytes = 0;
    int *xfer_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, *xfer_in[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [2], [3]: two ranks below / above; [4]: the far outbox and the world's far outboxes, all-gathered
    size_t xfer_bytes = 0, xfer2_bytes = 0, far_bytes = 0;      // xfer_bytes: what travels this step (grows with P.xfer_cap); the buffers have room for P.xfer_cap_max
    std::map<int, int> cap_decisions; // slab: record number -> the transfer capacity all ranks agreed on in that step (adopted two steps on)
    int *status_out = nullptr, *status_in = nullptr;   // status_in: world records, all-gathered
    size_t status_bytes = 0;
    int *allg_out = nullptr, *allg_in = nullptr;       // all-pairs across ranks: own snapshot block, all ranks' blocks (all-gathered)
    size_t allg_bytes = 0;
    int *pack_off[2] = {nullptr, nullptr}, *unpack_off[2] = {nullptr, nullptr};
    int slab_stage = 0;               // 0 idle, 1 built, 2 pairs done, 3 applied
    size_t frame_ints = 0;            // ints zeroed by init_iframe
    std::vector<void *> allocs;
    std::string err;
    // host mirrors
    std::vector<CellInfo> celltab;
    std::vector<QueueInfo> h_qinfo;   // valid while !queues_on_device_newer
    std::vector<int32_t> h_queue;
    bool host_queues_valid = true;    // host mirror == device copy
    FrameScalars *h_fs = nullptr;     // pinned host copies of the per-frame scalars: TWO records, a step's number picks one
    FrameScalars last{};              // the record of the last step the host has read (consume_scalars)
    int64_t processed_total = 0;      // sum over steps of the live particles at build_grid
    int64_t max_bucket_seen = 0;
    int bucket_cap0 = 2048;           // the longest operation list the step's replay instance sorts in LDS (2048 / 4096 / 8192, from the last lists seen)
    char *snapshot = nullptr;         // device image for snapshot_save / _restore
    int snapshot_step = 0;
    void *staging = nullptr;          // device staging for AoS transfers
    size_t staging_bytes = 0;
    // stage state machine
    bool frame_reset = false, grid_built = false, pairs_done = false;
    bool tdata_mirror = true;         // build_grid also writes the reference's T_DATA rows (psamd_set_tdata_mirror)
    bool frame_clean = true;          // the per-frame counts are zero: a finished step leaves them so (its last kernel is the next init_iframe)
    int step = 0;
    int64_t steps_total = 0;
    int live_at_build = -1;           // host copy of fs->live (valid after a sync)
    bool interior_ran = false;        // this step's pair stage ran in two passes (the scalars hold the second pass's task count)
    std::set<int> interior_steps;     // ... the numbers of such steps whose records have not been read yet
    int64_t tasks_last = 0;           // force tasks of the last step (all passes), sizes the next step's balanced pass
    int64_t packs_last = 0;           // ... of which packs of partly filled slices
    // upper bound of the live count at the next build_grid, kept on the host so that the
    // life-cycle kernels can be sized without a read-back (-1 = unknown)
    int64_t live_bound = 0, snapshot_live_bound = 0;
    // timing
    int timing = 0;                    // 0 off, 1 pair pass / apply / life cycle, 2 every stage
    int timing_period = 1;             // events are recorded on every timing_period-th step since set_timing
    int64_t timing_steps = 0;          // steps since set_timing
    int timing_now = 0;                // the level in force for the step being run (0 on the steps in between)
    // The step's scalars: the device numbers the records it hands out (StepState.seq), the host counts the steps it has
    // enqueued (scalars_seq) and the records it has read (scalars_seen).  run_ahead = 1: step k + 1 is enqueued once the
    // record of step k - 1 has been read -- the host is never on a step's critical path; 0: every step's own record is
    // waited for before the call returns.
    int scalars_seq = 0, scalars_seen = 0;
    int run_ahead = 1;
    int pending_status = PSAMD_OK;     // the verdict of a step whose record was read by a call that does not report verdicts (kept for the next that does)
    std::string pending_err;
    bool wedged = false;               // a step's scalars did not arrive within the wall-clock bound: the context refuses further work
    // Timing events: two sets, a timed step takes the one that was read longest ago; a set is read when it is taken again
    // or by psamd_get_timing -- never by the step that recorded it (the host runs ahead of the GPU).
    enum { E_RESET = 0, E_HIST, E_SCAN, E_SCATTER, E_SORT, E_SORT_END, E_COLLIDE, E_FORCE, E_PAIRS_END, E_APPLY, E_LIFE, E_END, E_COUNT };
    hipEvent_t ev[2][E_COUNT]{};
    int ev_level[2] = {0, 0};          // level a set was recorded at, 0: nothing outstanding in it
    int tset = 0;                      // the set of the step being run
    int64_t timed_steps = 0;
    bool ev_made = false;
    double t_us[PSAMD_NUM_TIMERS]{};
    std::vector<float> t_samples[PSAMD_NUM_TIMERS];
    int64_t t_launches = 0;
    // stage sequences as hipGraphs (psamd_set_graphs): per kind of sequence, the shapes captured so far
    struct GraphSlot { uint64_t key; hipGraphExec_t exec; uint64_t stamp; };
    bool graphs = false;
    std::vector<GraphSlot> gcache[5];
    uint64_t gstamp = 0;
    int64_t graph_launches = 0, graph_captures = 0;
    std::string graph_refused;         // why the runtime would not capture (the context then runs without graphs)
    int64_t slab_bound = 0;            // the bound slab_apply sized its launches from; slab_finish uses the same
    int wait_policy = 0;               // how the host waits for the step's scalars: 0 spin, 1 spin briefly, then nap
    double wait_limit_s = 10.0;        // ... and for how long at most (PSAMD_WAIT_LIMIT_S)
};

namespace {

const char *status_text(int s)
{
    switch (s) {
    case PSAMD_OK: return "ok";
    case PSAMD_ERR_INVALID_ARG: return "invalid argument";
    case PSAMD_ERR_NO_DEVICE: return "no usable HIP device (the HIP path is the only path)";
    case PSAMD_ERR_HIP: return "HIP runtime error";
    case PSAMD_ERR_OUT_OF_MEMORY: return "out of device memory";
    case PSAMD_ERR_OUTSIDE_BOX: return "particle location outside box";
    case PSAMD_ERR_QUEUE_EMPTY: return "overflow: reserved space of the segment is full";
    case PSAMD_ERR_CELL_OVERFLOW: return "cell or queue-op capacity exceeded on device";
    case PSAMD_ERR_STATE: return "stage called out of order";
    case PSAMD_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown status";
}

int fail(psamd_ctx *c, int status, const std::string &what)
{
    if (c) c->err = std::string(status_text(status)) + ": " + what;
    return status;
}

int hip_fail(psamd_ctx *c, hipError_t e, const char *what)
{
    return fail(c, e == hipErrorOutOfMemory ? PSAMD_ERR_OUT_OF_MEMORY : PSAMD_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}

#define PS_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail((c), e_, #call); } while (0)

template <typename T>
hipError_t dev_alloc(psamd_ctx *c, T **out, size_t n)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    c->allocs.push_back(p);
    *out = (T *)p;
    // PSAMD_POISON (tests): fresh device memory is usually zero, reused memory is not -- fill every
    // allocation with a pattern so that anything read before it is written shows up
    static const bool poison = std::getenv("PSAMD_POISON") != nullptr;
    if (poison) {
        e = hipMemset(p, 0xA5, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) e = hipDeviceSynchronize();     // (the fill must not overtake the context's own stream)
    }
    return e;
}

int ensure_staging(psamd_ctx *c, size_t bytes)
{
    if (bytes <= c->staging_bytes) return PSAMD_OK;
    if (c->staging) (void)hipFree(c->staging);
    c->staging = nullptr; c->staging_bytes = 0;
    PS_HIP(c, hipMalloc(&c->staging, bytes));
    c->staging_bytes = bytes;
    return PSAMD_OK;
}

// The device keeps the queue array for the owned segments only, back to back (like the slot
// arrays); the host mirrors are whole-container arrays whose foreign parts are never used.
template <typename F>
void for_owned_ranges(const psamd_ctx *c, F fn)
{
    size_t off = 0;
    for (int t = 0; t < 4; t++) {
        const int n = c->P.slot_n[t];
        if (n > 0) fn((size_t)c->P.slot_lo[t], (size_t)n, off);
        off += (size_t)n;
    }
}

int pull_queues(psamd_ctx *c)   // device -> host mirror
{
    if (c->host_queues_valid) return PSAMD_OK;
    PS_HIP(c, hipStreamSynchronize(c->stream));
    PS_HIP(c, hipMemcpy(c->h_qinfo.data(), c->d.qinfo, c->h_qinfo.size() * sizeof(QueueInfo), hipMemcpyDeviceToHost));
    hipError_t e = hipSuccess;
    for_owned_ranges(c, [&](size_t lo, size_t n, size_t off) {
        if (e == hipSuccess) e = hipMemcpy(c->h_queue.data() + lo, c->d.queue + off, n * sizeof(int32_t), hipMemcpyDeviceToHost);
    });
    PS_HIP(c, e);
    c->host_queues_valid = true;
    return PSAMD_OK;
}

int push_queues(psamd_ctx *c)   // host mirror -> device
{
    PS_HIP(c, hipMemcpyAsync(c->d.qinfo, c->h_qinfo.data(), c->h_qinfo.size() * sizeof(QueueInfo), hipMemcpyHostToDevice, c->stream));
    hipError_t e = hipSuccess;
    for_owned_ranges(c, [&](size_t lo, size_t n, size_t off) {
        if (e == hipSuccess) e = hipMemcpyAsync(c->d.queue + off, c->h_queue.data() + lo, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
    });
    PS_HIP(c, e);
    PS_HIP(c, hipStreamSynchronize(c->stream));
    return PSAMD_OK;
}

// q_remove on the host mirror (app_common.cu:305-339), used by the fill stage only
int host_q_remove(psamd_ctx *c, int seg_type, int seg_tid)
{
    QueueInfo &q = c->h_qinfo[(size_t)c->geo.segment_record(seg_type, seg_tid)];
    if (q.count <= 0) return -1;
    const int pos = q.front;
    if (q.count == 1) { q.front = -1; q.rear = -1; }
    else if (q.front == q.rloc + q.seg_size - 1) q.front = q.rloc;
    else q.front++;
    q.count--;
    const int item = c->h_queue[(size_t)pos];
    c->h_queue[(size_t)pos] = -1;
    return item;
}

int check_device_errors(psamd_ctx *c)   // after a sync: sticky error bits raised by kernels
{
    FrameScalars fs{};
    PS_HIP(c, hipMemcpy(&fs, c->d.fs, sizeof fs, hipMemcpyDeviceToHost));
    if (!c->grid_built) for (int k = 0; k < 5; k++) fs.n_out[k] = c->last.n_out[k];      // (between steps the device's record is the next frame's, zeroed)
    if (fs.error & (ERR_BAD_ID | ERR_BAD_POS)) {
        // an upload error is reported once and then cleared: the rejected records stay in the
        // container, the caller is expected to upload valid ones over them
        const int bits = fs.error;
        const int cleared = fs.error & ~(ERR_BAD_ID | ERR_BAD_POS);
        (void)hipMemcpy(&c->d.fs->error, &cleared, sizeof(int), hipMemcpyHostToDevice);
        return fail(c, PSAMD_ERR_INVALID_ARG, (bits & ERR_BAD_ID) ? "uploaded particle with id != slot index"
                                                                  : "uploaded live particle outside the box or with cell >= NUM_CELLS");
    }
    if (fs.error & ERR_CELL_TOO_BIG) return fail(c, PSAMD_ERR_CELL_OVERFLOW, "a cell holds more particles than the sort kernel ranks");
    if (fs.error & ERR_FOREIGN_CELL) return fail(c, PSAMD_ERR_STATE, "a particle stored on this rank sits in a cell layer the rank holds no state for");
    if (fs.error & ERR_HALO_OVERFLOW) {
        // say what was asked for, so that the caller can size the messages (the error may also have come in with a
        // neighbour's message header: then the numbers below are this rank's own and may all fit)
        char buf[512];
        std::snprintf(buf, sizeof buf, "a slab message had no room (raise halo_cap_cell / xfer_cap): this step this rank wanted to send %d / %d "
                      "transfer records down / up (room: %d each now -- it follows the traffic two steps behind, up to xfer_cap_max), %d / %d two ranks away (room %d), %d to a far rank (room %d); "
                      "a halo message holds halo_cap_cell = %d bodies per cell on average over a cell layer",
                      fs.n_out[0], fs.n_out[1], c->P.xfer_cap, fs.n_out[2], fs.n_out[3], c->P.xfer2_cap, fs.n_out[4], c->P.far_cap, c->P.halo_cap_cell);
        return fail(c, PSAMD_ERR_CELL_OVERFLOW, buf);
    }
    if (fs.error & ERR_SLAB_MISMATCH) return fail(c, PSAMD_ERR_STATE, "a slab message does not match the receiver's plan or counts");
    if (fs.error & ERR_REMOTE_RECORD0) return fail(c, PSAMD_ERR_CELL_OVERFLOW, "more cell-overflow kills in one step than a slab's status message carries (ps.cpp:1523-1526 frees them into queue record 0)");
    if (fs.error & ERR_CHUNK_CAP) return fail(c, PSAMD_ERR_CELL_OVERFLOW, "a chunk list passed MAX_PARTICLES_PER_CHUNK: the reference skips its tail (ps.cpp:1502-1508), this library does not reproduce that");
    if (fs.error & ERR_OPS_OVERFLOW) return fail(c, PSAMD_ERR_CELL_OVERFLOW, "lifecycle op buffer overflow");
    if (fs.error & ERR_HANDOFF_TIMEOUT) return fail(c, PSAMD_ERR_STATE, "force pass: a wave never saw the partial sums of the task it continues");
    if (fs.error) return fail(c, PSAMD_ERR_STATE, "device error bits " + std::to_string(fs.error));
    return PSAMD_OK;
}

void make_events(psamd_ctx *c)
{
    if (c->ev_made) return;
    for (auto &set : c->ev) for (auto &e : set) (void)hipEventCreate(&e);
    c->ev_made = true;
}

// read a set of timing events (waits for its step's last kernel: a set is read when it is taken again, two timed
// steps later, or by psamd_get_timing)
void collect_timing(psamd_ctx *c, int set)
{
    const int level = c->ev_level[set];
    if (!level) return;
    c->ev_level[set] = 0;
    if (hipEventSynchronize(c->ev[set][psamd_ctx::E_END]) != hipSuccess) return;
    // timers: hist scan scatter sort | force pass, apply, life cycle | frame reset | flags + active lists (two-pass prologue)
    using X = psamd_ctx;
    const int a[PSAMD_NUM_TIMERS] = {X::E_HIST, X::E_SCAN, X::E_SCATTER, X::E_SORT, X::E_FORCE, X::E_APPLY, X::E_LIFE, X::E_RESET, X::E_COLLIDE};
    const int b[PSAMD_NUM_TIMERS] = {X::E_SCAN, X::E_SCATTER, X::E_SORT, X::E_SORT_END, X::E_PAIRS_END, X::E_LIFE, X::E_END, X::E_HIST, X::E_FORCE};
    for (int k = 0; k < PSAMD_NUM_TIMERS; k++) {
        if (level < 2 && (k < 4 || k == 7)) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[set][a[k]], c->ev[set][b[k]]) == hipSuccess) {
            c->t_us[k] += 1000.0 * ms;
            c->t_samples[k].push_back(1000.0f * ms);
        }
    }
    c->t_launches++;
}

void tick(psamd_ctx *c, int e)        // a timing event on the context's stream, if this step carries them
{
    if (c->timing_now) (void)hipEventRecord(c->ev[c->tset][e], c->stream);
}

}  // namespace

extern "C" {

int psamd_abi_version(void) { return PSAMD_ABI_VERSION; }
const char *psamd_status_string(int status) { return status_text(status); }

int psamd_default_config(psamd_config *cfg)
{
    if (!cfg) return PSAMD_ERR_INVALID_ARG;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->max_particles_num = 1024 * 1024;  // common.h:12
    cfg->x_factor = 2;                     // common.h:13
    cfg->chunk_factor = 4;                 // common.h:29
    cfg->chunk_dim = 4;                    // common.h:30
    cfg->cell_size = 5.0;                  // common.h:52
    cfg->eps2 = 0.2;                       // common.h:53
    cfg->collision_radius = 0.4;           // common.h:54
    cfg->particle_weight = 60.0;           // common.h:55
    cfg->dt = 0.05;                        // common.h:69
    cfg->max_v = 10.0;                     // common.h:66
    cfg->explosion_speed = 3.0;            // common.h:67
    cfg->life_steps = 300.0;               // common.h:58
    cfg->device = 0;
    cfg->flags = 0;
    cfg->seed = 1;                         // RAND_SEED, common.h:56
    cfg->rank = 0;
    cfg->world = 1;
    cfg->drag = 0.0;
    cfg->force_sign = 1.0;
    return PSAMD_OK;
}

// bytes of a transfer message that carries up to `cap` records
static size_t xfer_msg_bytes(int cap) { return ((size_t)MSG_HEADER_WORDS + ((size_t)cap + 1) * (sizeof(XferRec) / sizeof(int))) * sizeof(int); }

static SlabPlan plan_for(const Geometry &g, const psamd_config &cfg)
{
    const bool given = cfg.world >= 1 && cfg.world <= PSAMD_MAX_RANKS && cfg.cuts[cfg.world] != 0;
    return make_slab_plan(g.F, g.D, g.seg_base, g.seg_size_t, g.info_base, cfg.rank, cfg.world, given ? cfg.cuts : nullptr);
}

// DevParams fields that describe what this rank holds (partition.hpp -> device_types.h)
static void fill_slab_params(const Geometry &g, const SlabPlan &pl, const psamd_config &cfg, DevParams &P)
{
    const int GG = g.G * g.G;
    P.rank = pl.rank; P.world = pl.world; P.num_cells_global = g.num_cells;
    const int first[4] = {pl.state_lo, pl.below_lo, pl.lentin_lo, pl.above_lo};
    const int layers[4] = {pl.state_hi - pl.state_lo, pl.lentin_lo - pl.below_lo, pl.lentin_hi - pl.lentin_lo, pl.above_hi - pl.above_lo};
    P.halo_cap_cell = (cfg.halo_cap_cell > 0 && cfg.halo_cap_cell < g.max_per_cell) ? cfg.halo_cap_cell : g.max_per_cell;
    P.xfer_cap = pl.world > 1 ? (cfg.xfer_cap > 0 ? cfg.xfer_cap : std::max(4096, GG * g.max_per_cell / 4)) : 0;
    // how far the transfer messages may grow (their buffers' room): by default a step's worst case -- everything two cell
    // layers hold, and a child of each (a particle moves one layer a step, two when the rounded sum lands on the far face)
    P.xfer_cap0 = P.xfer_cap;
    P.xfer_cap_max = pl.world > 1 ? std::max(P.xfer_cap, cfg.xfer_cap_max > 0 ? cfg.xfer_cap_max
                                                         : (int)std::min<int64_t>(4ll * GG * g.max_per_cell, INT32_MAX / 256)) : 0;
    int64_t slots = 0;
    for (int t = 0; t < 4; t++) { P.slot_lo[t] = pl.slot_lo[t]; P.slot_n[t] = pl.slot_hi[t] - pl.slot_lo[t]; slots += P.slot_n[t];
                                  P.rec_lo[t] = pl.rec_lo[t]; P.rec_hi[t] = pl.rec_hi[t]; }
    P.slots_total = (int)slots;
    int base = 0;
    int64_t sorted = 0;
    for (int r = 0; r < 4; r++) {
        P.reg_first[r] = first[r]; P.reg_layers[r] = layers[r]; P.reg_base[r] = base; P.reg_sorted[r] = (int)sorted;
        base += layers[r] * GG + 1;                                     // + the gap cell
        // the own block can hold every owned slot (overflow-killed entries keep their place in
        // the sorted order for the frame); a remote block what its messages can carry
        sorted += r == 0 ? slots : (int64_t)layers[r] * GG * P.halo_cap_cell;
    }
    P.n_local_cells = base; P.n_own_cells = layers[0] * GG;
    P.sorted_cap = (int)sorted;
    P.own_comp0 = (std::max(pl.cut_lo, pl.state_lo) - pl.state_lo) * GG;
    P.own_comp1 = (std::min(pl.cut_hi, pl.state_hi) - pl.state_lo) * GG;
    if (P.own_comp1 < P.own_comp0) P.own_comp1 = P.own_comp0;
    // the whole pair stage in one pass: the lent cells, then the own ones
    P.comp_lo[0] = P.reg_base[2]; P.comp_hi[0] = P.reg_base[2] + layers[2] * GG;
    P.comp_lo[1] = P.own_comp0; P.comp_hi[1] = P.own_comp1;
    P.comp_lo[2] = P.comp_hi[2] = 0;
    P.lentout_c0 = (pl.lentout_lo - pl.state_lo) * GG; P.lentout_c1 = (pl.lentout_hi - pl.state_lo) * GG;
}

int psamd_create(const psamd_config *cfg, psamd_ctx **out)
{
    if (!cfg || !out) return PSAMD_ERR_INVALID_ARG;
    *out = nullptr;
    psamd_ctx *c = new (std::nothrow) psamd_ctx();
    if (!c) return PSAMD_ERR_OUT_OF_MEMORY;
    *out = c;  // returned even on failure so psamd_last_error can explain; caller destroys it
    if (!c->geo.init(*cfg)) return fail(c, PSAMD_ERR_INVALID_ARG, "bad configuration (chunk_dim >= 3, sizes > 0, container < 2^31)");
    if (cfg->world < 1 || cfg->world > PSAMD_MAX_RANKS || cfg->rank < 0 || cfg->rank >= cfg->world) return fail(c, PSAMD_ERR_INVALID_ARG, "rank/world");
    const Geometry &g = c->geo;
    c->plan = plan_for(g, *cfg);
    if (!c->plan.valid) return fail(c, PSAMD_ERR_UNSUPPORTED, "no slab partition for this grid and world size (every rank needs >= 2 cell layers "
                                                                "and its neighbours must hold every layer it reads)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(c, PSAMD_ERR_NO_DEVICE, "hipGetDeviceCount found none");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(c, PSAMD_ERR_NO_DEVICE, "device ordinal out of range");
    PS_HIP(c, hipSetDevice(cfg->device));
    PS_HIP(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    PS_HIP(c, hipStreamCreateWithFlags(&c->d.side_stream, hipStreamNonBlocking));
    PS_HIP(c, hipEventCreateWithFlags(&c->d.ev_fork, hipEventDisableTiming));
    PS_HIP(c, hipEventCreateWithFlags(&c->d.ev_join, hipEventDisableTiming));

    DevParams &P = c->P;
    P.G = g.G; P.num_cells = g.num_cells; P.num_chunks = g.num_chunks; P.container = g.container;
    P.max_per_cell = g.max_per_cell; P.max_per_chunk = g.max_per_chunk;
    P.slices = (g.max_per_cell + 63) / 64;
    P.flags = cfg->flags;
    P.t = (float)cfg->dt;
    P.kid_thr = float_ceil(g.kid_age);
    P.life_thr = float_floor(g.particle_life);
    // every pair whose fp32 squared distance is at or below this gets the exact
    // collision test; the margin only has to cover the rounding of sqrtf
    P.coll_d2_gate = (float)(cfg->collision_radius * cfg->collision_radius * 1.001 + 1e-30);
    P.dmax = (float)cfg->cell_size;   // MAX_DX = CELL_SIZE, common.h:65
    P.vmax = (float)cfg->max_v;
    P.w_default = (float)cfg->particle_weight;
    P.fert_lo = (float)g.min_fert; P.fert_hi = (float)g.max_fert;
    P.cell_size = cfg->cell_size; P.eps2 = cfg->eps2; P.coll_radius = cfg->collision_radius;
    P.kid_age = g.kid_age; P.life = g.particle_life; P.expl_speed = cfg->explosion_speed;
    P.seed = cfg->seed;
    P.drag = (float)cfg->drag;
    P.force_sign = cfg->force_sign < 0 ? -1.0f : 1.0f;
    if (cfg->drag < 0) return fail(c, PSAMD_ERR_INVALID_ARG, "drag must be >= 0");
    fill_slab_params(g, c->plan, *cfg, P);
    P.status_words = STATUS_CHUNK_OFF + 4 * g.num_chunks;
    if (cfg->world > 1) {
        // the ring neighbours' queue records; and whether any rank's whole state is ONE cell layer: a particle
        // that crosses two layers in a step (from one ulp below a face, moved by exactly CELL_SIZE) can fly over
        // such a rank, to the rank beyond it -- those records get outboxes of their own, sent straight to rank +-2
        bool single = false;
        psamd_config rc = *cfg;
        for (int r = 0; r < cfg->world; r++) {
            rc.rank = r;
            const SlabPlan pr = plan_for(g, rc);
            if (!pr.valid) return fail(c, PSAMD_ERR_UNSUPPORTED, "no slab partition for this grid and world size");
            single |= pr.state_hi - pr.state_lo < 2;
            const int which = r == (cfg->rank + cfg->world - 1) % cfg->world ? 0 : -1, which2 = r == (cfg->rank + 1) % cfg->world ? 1 : -1;
            for (int w : {which, which2})
                if (w >= 0) for (int t = 0; t < 4; t++) { P.nbr_rec_lo[w][t] = pr.rec_lo[t]; P.nbr_rec_hi[w][t] = pr.rec_hi[t]; }
        }
        P.xfer2_cap = (single && cfg->world >= 4) ? 1024 : 0;       // (a ring of two or three has no rank beyond the neighbours)
        // A record for a rank further away: only a particle whose position stopped being a number travels that far (it is
        // filed under one fixed cell wherever it was), and only births make such particles (a child with the direction
        // (0, 0, 0)): with births on, a world of four or more all-gathers a small far outbox in the transfer phase.
        P.far_cap = (cfg->world >= 4 && (cfg->flags & PSAMD_FLAG_EXPLOSIONS)) ? 16 : 0;
    }
    if ((cfg->flags & PSAMD_FLAG_ALL_PAIRS) && cfg->world > 1) {
        // every rank's block of the all-gathered snapshot has the same size: room for the rank with the most cells / slots
        int cells = 0;
        int64_t slots = 0;
        psamd_config rc = *cfg;
        for (int r = 0; r < cfg->world; r++) {
            rc.rank = r;
            const SlabPlan pr = plan_for(g, rc);
            if (!pr.valid) return fail(c, PSAMD_ERR_UNSUPPORTED, "no slab partition for this grid and world size");
            int64_t sl = 0;
            for (int t = 0; t < 4; t++) sl += pr.slot_hi[t] - pr.slot_lo[t];
            cells = std::max(cells, (pr.state_hi - pr.state_lo) * g.G * g.G);
            slots = std::max(slots, sl);
        }
        P.allg_cells = cells;
        P.allg_cap = (int)((slots + 63) & ~(int64_t)63);
        P.allg_block = MSG_HEADER_WORDS + P.allg_cells + 4 * P.allg_cap;
    }
    auto bits_for = [](int64_t n) { int b = 1; while (((int64_t)1 << b) < n) b++; return b; };
    P.key_chunk_shift = 2 + bits_for(g.container);
    P.key_rec_shift = P.key_chunk_shift + bits_for((int64_t)g.num_chunks + 1);
    P.key_bits = P.key_rec_shift + bits_for(g.queue_infos);
    {
        // (r.r + eps2)^3 over every pair of in-box positions, with slack for one wrap of drift
        const double L = (double)g.G * cfg->cell_size;
        const double lo = cfg->eps2 * cfg->eps2 * cfg->eps2, hi = std::pow(3.0 * (2.0 * L) * (2.0 * L) + cfg->eps2, 3.0);
        P.lean_math = (lo > std::ldexp(1.0, -60) && hi < std::ldexp(1.0, 60)) ? 1 : 0;
        // Collision flags before forces (lean modes): a collision needs two bodies within
        // COLLISION_RADIUS, so only bodies that close to a cell face concern the cell beyond it;
        // 2.5 % + 1e-3 of slack covers every rounding between here and the exact test.  Needs
        // the radius to be small against the cell (else the halo is the whole neighbour).
        P.halo_reach = (float)(cfg->collision_radius * 1.025 + 1e-3);
        {   // largest float t with (double)sqrtf(t) <= COLLISION_RADIUS (sqrtf: correctly rounded, monotone)
            auto collides = [&](float t) { return !((double)std::sqrt(t) > cfg->collision_radius); };
            uint32_t lo_b = 0u, hi_b = 0x7f7fffffu;                 // bit patterns of non-negative floats order like the floats
            auto as_f = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
            if (!collides(0.0f)) P.coll_d2_max = -1.0f;
            else {
                while (lo_b < hi_b) { const uint32_t mid = lo_b + (hi_b - lo_b + 1) / 2; if (collides(as_f(mid))) lo_b = mid; else hi_b = mid - 1; }
                P.coll_d2_max = as_f(lo_b);
            }
        }
        P.two_pass = (P.lean_math && P.halo_reach < 0.25 * cfg->cell_size && !std::getenv("PSAMD_ONE_PASS")) ? 1 : 0;
    }
    if (P.key_bits > 63) return fail(c, PSAMD_ERR_UNSUPPORTED, "queue-op key does not fit 64 bits for this configuration");
    if ((cfg->flags & PSAMD_FLAG_ALL_PAIRS) && !P.lean_math) return fail(c, PSAMD_ERR_UNSUPPORTED, "all-pairs forces are built for the lean pair arithmetic only (EPS2 in its validated range)");
    if ((cfg->flags & PSAMD_FLAG_ALL_PAIRS) && !P.two_pass) return fail(c, PSAMD_ERR_UNSUPPORTED, "all-pairs forces need the two-pass pair stage (collision radius small against the cell)");
    for (int k = 0; k < 5; k++) { c->S.seg_base[k] = g.seg_base[k]; c->S.info_base[k] = g.info_base[k]; }
    for (int k = 0; k < 4; k++) c->S.seg_size_t[k] = g.seg_size_t[k];

    DeviceState &d = c->d;
    const size_t C = (size_t)P.slots_total;          // owned slots
    const size_t SC = (size_t)P.sorted_cap + 64;     // sorted-order arrays (+ slack: scalar loads fetch whole groups)
    const size_t LC = (size_t)P.n_local_cells;
    const size_t xf = (size_t)P.xfer_cap_max + (size_t)P.xfer2_cap + (size_t)P.far_cap * (size_t)std::max(1, P.world) / 2 + 1;
    d.ops_cap = (int)std::min<size_t>(3 * C + 2 * xf + 64 + (P.world > 1 ? (size_t)P.world * STATUS_KILL_CAP : 0), (size_t)INT32_MAX / 2);
    d.moves_cap = (int)std::min<size_t>(2 * C + 2 * xf + 64, (size_t)INT32_MAX / 2);
    int *frame = nullptr;
    // cell counts, chunk counts, queue-op counts and cursors per record, halo counts, active-list lengths, hand-off flags of the force pass
    const size_t frame_ints = LC + g.num_chunks + 2 * (size_t)g.queue_infos + LC + LC + 2 * LC * P.slices;   // (flags: one block per pass of the pair stage)
    PS_HIP(c, dev_alloc(c, &d.pos4, C));
    PS_HIP(c, dev_alloc(c, &d.vel4, C));
    PS_HIP(c, dev_alloc(c, &d.acc4, C));
    PS_HIP(c, dev_alloc(c, &d.cell, C));
    PS_HIP(c, dev_alloc(c, &d.pflags, C));
    PS_HIP(c, dev_alloc(c, &d.tdata, 6 * C));
    PS_HIP(c, dev_alloc(c, &d.qinfo, (size_t)g.queue_infos));
    PS_HIP(c, dev_alloc(c, &d.queue, C));
    PS_HIP(c, dev_alloc(c, &frame, frame_ints));
    d.cell_count = frame; d.chunk_count = frame + LC; d.rec_count = d.chunk_count + g.num_chunks;
    d.rec_cursor = d.rec_count + g.queue_infos;
    d.halo_count = d.rec_cursor + g.queue_infos;
    d.active_count = d.halo_count + LC;
    d.task_ready = d.active_count + LC;
    c->frame_ints = frame_ints;
    PS_HIP(c, dev_alloc(c, &d.halo_f, (size_t)3 * LC * HALO_CAP + 64));   // + slack: scalar loads fetch whole groups
    PS_HIP(c, dev_alloc(c, &d.halo_id, LC * HALO_CAP + 64));
    PS_HIP(c, dev_alloc(c, &d.active_list, SC));
    PS_HIP(c, dev_alloc(c, &d.snap_cid, SC));
    PS_HIP(c, dev_alloc(c, &d.task_list2, LC * P.slices));
    PS_HIP(c, dev_alloc(c, &d.merged_tasks, LC));
    PS_HIP(c, dev_alloc(c, &d.task_cost, LC));
    PS_HIP(c, dev_alloc(c, &d.ctask_start, 2 * LC + 2));     // (+ one virtual cell per merged pack)
    PS_HIP(c, dev_alloc(c, &d.cost_start, 2 * LC + 2));
    PS_HIP(c, dev_alloc(c, &d.wave_pos, (size_t)MAX_PAIR_WAVES + 1));
    PS_HIP(c, dev_alloc(c, &d.rec_start, (size_t)g.queue_infos + 1));
    PS_HIP(c, dev_alloc(c, &d.fs, 1));
    PS_HIP(c, dev_alloc(c, &d.st, 1));
    PS_HIP(c, hipMemsetAsync(d.st, 0, sizeof(StepState), c->stream));
    c->wait_policy = cfg->world > 1 ? 1 : 0;
    PS_HIP(c, dev_alloc(c, &d.cell_start, LC + 1));
    PS_HIP(c, dev_alloc(c, &d.cursor, LC));
    PS_HIP(c, dev_alloc(c, &d.task_start, LC + 1));
    PS_HIP(c, dev_alloc(c, &d.task_list, LC * P.slices));
    PS_HIP(c, dev_alloc(c, &d.sorted_id, SC));
    PS_HIP(c, dev_alloc(c, &d.flag_slot, C));
    PS_HIP(c, dev_alloc(c, &d.snap_soa, 4 * (size_t)P.sorted_cap + 64));
    PS_HIP(c, dev_alloc(c, &d.snap_age, SC));
    PS_HIP(c, dev_alloc(c, &d.force4, SC));
    PS_HIP(c, dev_alloc(c, &d.celltab, (size_t)g.num_cells));
    if (P.flags & PSAMD_FLAG_ALL_PAIRS) {
        // partial sums of the all-pairs far pass: one float4 per (part, particle that needs a force) -- dense, 64 to a task;
        // every entry of the sorted order could be one
        d.part_tasks = (int)(SC / 64 + 1);
        PS_HIP(c, dev_alloc(c, &d.part_acc, (size_t)ALLP_PARTS * d.part_tasks * 64));
        PS_HIP(c, dev_alloc(c, &d.act_start, LC + 1));
        PS_HIP(c, dev_alloc(c, &d.dense_gi, SC));
        PS_HIP(c, dev_alloc(c, &d.dense_cell, SC));
    }
    {                                                    // the chunk lists' capacity rule (chunk_cap_block)
        PS_HIP(c, dev_alloc(c, &d.chunk_skip, C));
        PS_HIP(c, dev_alloc(c, &d.chunk_segs, (size_t)g.num_chunks * 27));
        std::vector<int2> segs((size_t)g.num_chunks * 27);
        for (int ch = 0; ch < g.num_chunks; ch++) {
            Pair pk[27];
            g.chunk_segments(ch, pk);
            for (int j = 0; j < 27; j++) {
                const int k = seg_index(pk[j].c);
                segs[(size_t)ch * 27 + j] = make_int2(g.seg_base[k] + pk[j].p * g.seg_size_t[k], g.seg_size_t[k]);
            }
            // slot order (set_pkg_segments lists them so already; the walk must not depend on it)
            std::sort(segs.begin() + (size_t)ch * 27, segs.begin() + (size_t)ch * 27 + 27, [](const int2 &a, const int2 &b) { return a.x < b.x; });
        }
        PS_HIP(c, hipMemcpy(d.chunk_segs, segs.data(), segs.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    PS_HIP(c, dev_alloc(c, &d.op_keys, (size_t)d.ops_cap));
    PS_HIP(c, dev_alloc(c, &d.op_keys_sorted, (size_t)d.ops_cap));
    PS_HIP(c, dev_alloc(c, &d.op_args, (size_t)d.ops_cap));
    PS_HIP(c, dev_alloc(c, &d.op_args_sorted, (size_t)d.ops_cap));
    // the host polls these records (wait_scalars): coherent mapping whatever HIP_HOST_COHERENT says, and zeroed --
    // hipHostMalloc does not promise zeroed pages, and a recycled page whose seq word happened to hold the number
    // the first step waits for would be taken for that step's scalars
    PS_HIP(c, hipHostMalloc((void **)&c->h_fs, 2 * sizeof(FrameScalars), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_fs, 0, 2 * sizeof(FrameScalars));
    if (const char *lim = std::getenv("PSAMD_WAIT_LIMIT_S")) c->wait_limit_s = std::max(0.05, std::atof(lim));
    PS_HIP(c, hipHostGetDevicePointer((void **)&c->d.fs_host, c->h_fs, 0));
    PS_HIP(c, dev_alloc(c, &d.moves, (size_t)d.moves_cap));
    PS_HIP(c, dev_alloc(c, &d.stage, 3 * (size_t)d.moves_cap));
    PS_HIP(c, dev_alloc(c, &d.ctr, (size_t)COUNTER_COPIES));
    PS_HIP(c, dev_alloc(c, &d.trace, 3 * (LC * P.slices + 4)));
    PS_HIP(c, hipMemsetAsync(d.trace, 0, 3 * (LC * P.slices + 4) * sizeof(unsigned long long), c->stream));

    // slab messages: sizes fixed by the plan (see kernels.hip "slab exchange")
    if (P.world > 1) {
        const SlabPlan &pl = c->plan;
        const int GG = g.G * g.G;
        auto halo_words = [&](int cells) { return (size_t)MSG_HEADER_WORDS + (size_t)cells + 6 * (size_t)cells * P.halo_cap_cell; };
        // out: own layers for the rank below / above; in: what they hold for this rank
        c->halo_out_cells[0] = (pl.send_down_hi - pl.send_down_lo) * GG; c->halo_out_c0[0] = (pl.send_down_lo - pl.state_lo) * GG;
        c->halo_out_cells[1] = (pl.send_up_hi - pl.send_up_lo) * GG;     c->halo_out_c0[1] = (pl.send_up_lo - pl.state_lo) * GG;
        c->halo_in_cells[0] = (pl.below_hi - pl.below_lo) * GG;
        c->halo_in_cells[1] = (pl.above_hi - pl.above_lo) * GG;
        for (int k = 0; k < 2; k++) {
            if (c->halo_out_cells[k] > 0) {
                c->halo_out_bytes[k] = halo_words(c->halo_out_cells[k]) * sizeof(int);
                PS_HIP(c, dev_alloc(c, &c->halo_out[k], c->halo_out_bytes[k] / sizeof(int)));
                PS_HIP(c, hipMemsetAsync(c->halo_out[k], 0, c->halo_out_bytes[k], c->stream));
                PS_HIP(c, dev_alloc(c, &c->pack_off[k], (size_t)c->halo_out_cells[k] + 1));
            }
            if (c->halo_in_cells[k] > 0) {
                c->halo_in_bytes[k] = halo_words(c->halo_in_cells[k]) * sizeof(int);
                PS_HIP(c, dev_alloc(c, &c->halo_in[k], c->halo_in_bytes[k] / sizeof(int)));
                PS_HIP(c, hipMemsetAsync(c->halo_in[k], 0, c->halo_in_bytes[k], c->stream));
                PS_HIP(c, dev_alloc(c, &c->unpack_off[k], (size_t)c->halo_in_cells[k] + 1));
            }
        }
        auto force_words = [&](int cells) { return (size_t)MSG_HEADER_WORDS + 4 * (size_t)cells * P.halo_cap_cell; };
        if (P.reg_layers[2] > 0) {
            c->force_out_bytes = force_words(P.reg_layers[2] * GG) * sizeof(int);
            PS_HIP(c, dev_alloc(c, &c->force_out, c->force_out_bytes / sizeof(int)));
            PS_HIP(c, hipMemsetAsync(c->force_out, 0, c->force_out_bytes, c->stream));
        }
        if (P.lentout_c1 > P.lentout_c0) {
            c->force_in_bytes = force_words(P.lentout_c1 - P.lentout_c0) * sizeof(int);
            PS_HIP(c, dev_alloc(c, &c->force_in, c->force_in_bytes / sizeof(int)));
            PS_HIP(c, hipMemsetAsync(c->force_in, 0, c->force_in_bytes, c->stream));
        }
        const size_t xfer_alloc = ((size_t)MSG_HEADER_WORDS + xf * (sizeof(XferRec) / sizeof(int))) * sizeof(int);
        c->xfer_bytes = xfer_msg_bytes(P.xfer_cap);
        c->xfer2_bytes = P.xfer2_cap > 0 ? ((size_t)MSG_HEADER_WORDS + (size_t)P.xfer2_cap * (sizeof(XferRec) / sizeof(int))) * sizeof(int) : 0;
        for (int k = 0; k < 4; k++) {
            const size_t bytes = k < 2 ? xfer_alloc : c->xfer2_bytes;
            if (bytes == 0) continue;
            PS_HIP(c, dev_alloc(c, &c->xfer_out[k], bytes / sizeof(int)));
            PS_HIP(c, dev_alloc(c, &c->xfer_in[k], bytes / sizeof(int)));
            PS_HIP(c, hipMemsetAsync(c->xfer_out[k], 0, bytes, c->stream));
            PS_HIP(c, hipMemsetAsync(c->xfer_in[k], 0, bytes, c->stream));
            d.xfer_out[k] = reinterpret_cast<XferRec *>(c->xfer_out[k] + MSG_HEADER_WORDS);
        }
        if (P.far_cap > 0) {
            c->far_bytes = ((size_t)MSG_HEADER_WORDS + (size_t)P.far_cap * (sizeof(XferRec) / sizeof(int))) * sizeof(int);
            PS_HIP(c, dev_alloc(c, &c->xfer_out[4], c->far_bytes / sizeof(int)));
            PS_HIP(c, dev_alloc(c, &c->xfer_in[4], c->far_bytes / sizeof(int) * (size_t)P.world));
            PS_HIP(c, hipMemsetAsync(c->xfer_out[4], 0, c->far_bytes, c->stream));
            PS_HIP(c, hipMemsetAsync(c->xfer_in[4], 0, c->far_bytes * (size_t)P.world, c->stream));
            d.xfer_out[4] = reinterpret_cast<XferRec *>(c->xfer_out[4] + MSG_HEADER_WORDS);
        }
        if (P.flags & PSAMD_FLAG_ALL_PAIRS) {
            c->allg_bytes = (size_t)P.allg_block * sizeof(int);
            PS_HIP(c, dev_alloc(c, &c->allg_out, (size_t)P.allg_block));
            PS_HIP(c, dev_alloc(c, &c->allg_in, (size_t)P.allg_block * P.world + 64));      // + slack: scalar loads fetch whole groups
            PS_HIP(c, hipMemsetAsync(c->allg_out, 0, c->allg_bytes, c->stream));
            PS_HIP(c, hipMemsetAsync(c->allg_in, 0, (c->allg_bytes * P.world) + 64 * sizeof(int), c->stream));
            PS_HIP(c, dev_alloc(c, &d.gstart, (size_t)g.num_cells + 1));
            PS_HIP(c, dev_alloc(c, &d.gn, (size_t)g.num_cells + 1));
            PS_HIP(c, hipMemsetAsync(d.gstart, 0, ((size_t)g.num_cells + 1) * sizeof(int), c->stream));
            PS_HIP(c, hipMemsetAsync(d.gn, 0, ((size_t)g.num_cells + 1) * sizeof(int), c->stream));
            d.allg_in = c->allg_in;
        }
        c->status_bytes = (size_t)P.status_words * sizeof(int);
        PS_HIP(c, dev_alloc(c, &c->status_out, (size_t)P.status_words));
        PS_HIP(c, dev_alloc(c, &c->status_in, (size_t)P.status_words * P.world));
        PS_HIP(c, hipMemsetAsync(c->status_out, 0, c->status_bytes, c->stream));
        PS_HIP(c, hipMemsetAsync(c->status_in, 0, c->status_bytes * P.world, c->stream));
        d.status_out = c->status_out;
    }

    // From which squared distance on is the fp32 add of EPS2 bit-identical to the
    // reference's double add?  Try a few candidates, each checked on the device for
    // every float up to the largest squared distance two in-box particles can have.
    P.eps2f = (float)cfg->eps2;
    P.eps_f32_from = INFINITY;   // default: always add in double
    if (P.lean_math && cfg->eps2 > 0) {
        const double L = (double)g.G * cfg->cell_size;
        const float d2_max = (float)(3.0 * (2.0 * L) * (2.0 * L));
        unsigned long long *bad = (unsigned long long *)d.fs;   // scratch, zeroed again below
        auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
        // (below 1.5 x EPS2 the sum lies under 0.5, a binade finer, and the rounding error of (float)EPS2 shows: 838 861 of the floats
        // in [1.25 EPS2, 1.5 EPS2) differ at the reference's EPS2 -- lower multiples are not worth a try)
        for (double mult : {1.5, 2.0, 4.0, 8.0, 16.0, 64.0}) {
            const float from = (float)(mult * cfg->eps2);
            if (!(from < d2_max)) break;
            unsigned long long h_bad = 1;
            PS_HIP(c, hipMemsetAsync(bad, 0, sizeof(unsigned long long), c->stream));
            PS_HIP(c, launch_validate_eps(c->stream, fbits(from), fbits(d2_max), cfg->eps2, P.eps2f, bad));
            PS_HIP(c, hipMemcpyAsync(&h_bad, bad, sizeof h_bad, hipMemcpyDeviceToHost, c->stream));
            PS_HIP(c, hipStreamSynchronize(c->stream));
            if (h_bad == 0) { P.eps_f32_from = from; break; }
        }
    }

    P.slow_below = std::max(P.eps_f32_from, std::nextafterf(P.coll_d2_gate, INFINITY));
    {   // Interior own cells: computed layers whose neighbour layers are both own state (or outside
        // the grid): their collision flags and forces need nothing from another rank, so that
        // pass can run while the halo messages travel (two-pass mode; slab_pairs_interior).
        const SlabPlan &pl = c->plan;
        const int GG = g.G * g.G, lo = std::max(pl.cut_lo, pl.state_lo), hi = std::min(pl.cut_hi, pl.state_hi);
        auto own = [&](int l) { return l < 0 || l >= g.G || (l >= pl.state_lo && l < pl.state_hi); };
        int i0 = lo, i1 = lo;
        for (int l = lo; l < hi; l++) if (own(l - 1) && own(l + 1)) { if (i1 == i0) i0 = l; i1 = l + 1; } else if (i1 > i0) break;
        c->P_int = P; c->P_rest = P;
        c->have_interior = P.world > 1 && P.two_pass && P.lean_math && i1 > i0 && (i1 - i0) < (hi - lo) + (pl.lentin_hi - pl.lentin_lo)
                           && !(P.flags & PSAMD_FLAG_ALL_PAIRS);      // (an all-pairs pass needs the gathered snapshot: nothing to do before it lands)
        if (c->have_interior) {
            const int a = (i0 - pl.state_lo) * GG, b = (i1 - pl.state_lo) * GG;
            c->P_int.comp_lo[0] = a; c->P_int.comp_hi[0] = b;
            c->P_int.comp_lo[1] = c->P_int.comp_hi[1] = c->P_int.comp_lo[2] = c->P_int.comp_hi[2] = 0;
            c->P_rest.comp_lo[1] = P.own_comp0; c->P_rest.comp_hi[1] = a;
            c->P_rest.comp_lo[2] = b; c->P_rest.comp_hi[2] = P.own_comp1;
        }
    }

    // init_particles (ps.cpp:722-753): every slot reset, cell = -1
    PS_HIP(c, hipMemsetAsync(d.pos4, 0, std::max<size_t>(C, 1) * sizeof(float4), c->stream));
    PS_HIP(c, hipMemsetAsync(d.vel4, 0, std::max<size_t>(C, 1) * sizeof(float4), c->stream));
    PS_HIP(c, hipMemsetAsync(d.acc4, 0, std::max<size_t>(C, 1) * sizeof(float4), c->stream));
    PS_HIP(c, hipMemsetAsync(d.pflags, 0, std::max<size_t>(C, 1), c->stream));
    PS_HIP(c, hipMemsetAsync(d.force4, 0, SC * sizeof(float4), c->stream));
    PS_HIP(c, hipMemsetAsync(d.flag_slot, 0, std::max<size_t>(C, 1), c->stream));
    PS_HIP(c, hipMemsetAsync(d.fs, 0, sizeof(FrameScalars), c->stream));
    PS_HIP(c, hipMemsetAsync(d.ctr, 0, sizeof(DevCounters) * COUNTER_COPIES, c->stream));
    PS_HIP(c, hipMemsetAsync(frame, 0, frame_ints * sizeof(int), c->stream));
    PS_HIP(c, hipMemsetAsync(d.cell_start, 0, (LC + 1) * sizeof(int), c->stream));
    PS_HIP(c, launch_fill_int(c->stream, d.cell, -1, C));
    PS_HIP(c, launch_init_tdata(c->stream, P, d));
    // q_start_fast (ps.cpp:814-871) and the cell table
    g.initial_queues(c->h_qinfo, c->h_queue);
    c->celltab = g.cell_table();
    PS_HIP(c, hipMemcpyAsync(d.celltab, c->celltab.data(), c->celltab.size() * sizeof(CellInfo), hipMemcpyHostToDevice, c->stream));
    {   // k_sort_cells' order of the own cells: the cells of one segment (they share its slots, so their gathers share
        // cache lines) side by side -- one XCD's L2 then sees a segment's lines once
        const int cell_off = P.reg_first[0] * g.G * g.G;
        std::vector<int> order((size_t)std::max(P.n_own_cells, 1), 0);
        for (int lc = 0; lc < P.n_own_cells; lc++) order[(size_t)lc] = lc;
        const char *e = getenv("PSAMD_CELL_ORDER");
        if (!e || atoi(e) != 0)
            std::stable_sort(order.begin(), order.begin() + P.n_own_cells, [&](int a, int b) {
                const CellInfo &x = c->celltab[(size_t)(a + cell_off)], &y = c->celltab[(size_t)(b + cell_off)];
                if (x.chunk != y.chunk) return x.chunk < y.chunk;
                if (x.seg_type != y.seg_type) return x.seg_type < y.seg_type;
                return x.seg_tid < y.seg_tid;
            });
        PS_HIP(c, dev_alloc(c, &d.cell_order, order.size()));
        PS_HIP(c, hipMemcpy(d.cell_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    int rc = push_queues(c);
    if (rc != PSAMD_OK) return rc;
    c->host_queues_valid = true;
    return PSAMD_OK;
}

int psamd_destroy(psamd_ctx *c)
{
    if (!c) return PSAMD_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &cache : c->gcache) for (auto &g : cache) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->staging) (void)hipFree(c->staging);
    if (c->h_fs) (void)hipHostFree(c->h_fs);
    if (c->ev_made) for (auto &set : c->ev) for (auto &e : set) (void)hipEventDestroy(e);
    if (c->d.ev_fork) (void)hipEventDestroy(c->d.ev_fork);
    if (c->d.ev_join) (void)hipEventDestroy(c->d.ev_join);
    if (c->d.side_stream) (void)hipStreamDestroy(c->d.side_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return PSAMD_OK;
}

const char *psamd_last_error(const psamd_ctx *c) { return c ? c->err.c_str() : "null context"; }

static void fill_sizes(const Geometry &g, psamd_sizes *o)
{
    std::memset(o, 0, sizeof *o);
    o->grid_dim = g.G; o->num_cells = g.num_cells; o->num_chunks = g.num_chunks;
    o->cells_per_chunk = g.cells_per_chunk; o->max_per_cell = g.max_per_cell; o->max_per_chunk = g.max_per_chunk;
    o->container_size = g.container; o->queue_info_size = g.queue_infos;
    o->n_chunkgrid = (int64_t)g.num_chunks * (1 + (int64_t)g.max_per_chunk);
    o->n_cellgrid = (int64_t)g.num_cells * (1 + (int64_t)g.max_per_cell);
    o->n_pkgdistrib = g.num_chunks * 27;
    for (int k = 0; k < 4; k++) { o->seg_count[k] = g.seg_count[k]; o->seg_size_t[k] = g.seg_size_t[k]; o->seg_size[k] = g.seg_size[k]; }
}

int psamd_describe(const psamd_config *cfg, psamd_sizes *sizes, int32_t *cell_table3, int32_t *pkg,
                   void *queue_info24, int32_t *queue)
{
    if (!cfg) return PSAMD_ERR_INVALID_ARG;
    Geometry g;
    if (!g.init(*cfg)) return PSAMD_ERR_INVALID_ARG;
    if (sizes) fill_sizes(g, sizes);
    if (cell_table3)
        for (int i = 0; i < g.num_cells; i++) {
            const CellInfo ci = g.cell_info(i);
            cell_table3[3 * i] = ci.chunk; cell_table3[3 * i + 1] = ci.seg_type; cell_table3[3 * i + 2] = ci.seg_tid;
        }
    if (pkg) for (int ch = 0; ch < g.num_chunks; ch++) g.chunk_segments(ch, (Pair *)pkg + (size_t)ch * 27);
    if (queue_info24 || queue) {
        std::vector<QueueInfo> qi;
        std::vector<int32_t> q;
        g.initial_queues(qi, q);
        if (queue_info24) std::memcpy(queue_info24, qi.data(), qi.size() * sizeof(QueueInfo));
        if (queue) std::memcpy(queue, q.data(), q.size() * sizeof(int32_t));
    }
    return PSAMD_OK;
}

int psamd_get_sizes(const psamd_ctx *c, psamd_sizes *o)
{
    if (!c || !o) return PSAMD_ERR_INVALID_ARG;
    fill_sizes(c->geo, o);
    return PSAMD_OK;
}

int psamd_get_config(const psamd_ctx *c, psamd_config *o)
{
    if (!c || !o) return PSAMD_ERR_INVALID_ARG;
    *o = c->geo.cfg;
    return PSAMD_OK;
}

int psamd_uniform_cloud(const psamd_ctx *c, int64_t n, uint32_t seed, float *xyz)
{
    if (!c || !xyz || n < 0) return PSAMD_ERR_INVALID_ARG;
    // ps.cpp:974-1028 draws r*sign*range per axis from a random_device-seeded mt19937;
    // a fixed seed and one uniform draw per axis give the same distribution reproducibly
    const float half = (float)((c->geo.G / 2) * c->geo.cfg.cell_size);
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> dist(-half, half);
    for (int64_t i = 0; i < n; i++) {
        // a draw can land exactly on a face that belongs to the neighbouring (missing)
        // cell: -half on the negated axes y and z, +half (float rounding) on x; draw again
        int cell;
        do {
            xyz[3 * i] = dist(gen); xyz[3 * i + 1] = dist(gen); xyz[3 * i + 2] = dist(gen);
        } while (!c->geo.locate(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cell));
    }
    return PSAMD_OK;
}

int psamd_fill_particles(psamd_ctx *c, int64_t n, const float *xyz, const float *vxyz, const float *w,
                         const float *age, const float *fert_age, int32_t *ids_out, int64_t *n_done)
{
    if (n_done) *n_done = 0;
    if (!c || n < 0 || (n > 0 && !xyz)) return PSAMD_ERR_INVALID_ARG;
    if (n == 0) return PSAMD_OK;
    int rc = pull_queues(c);
    if (rc != PSAMD_OK) return rc;
    const Geometry &g = c->geo;
    struct Rec { float4 p, v, a; int cell; };
    std::vector<int32_t> ids((size_t)n);
    std::vector<Rec> recs((size_t)n);
    int64_t done = 0, placed = 0;
    int status = PSAMD_OK;
    for (; done < n; done++) {
        const float x = xyz[3 * done], y = xyz[3 * done + 1], z = xyz[3 * done + 2];
        int cell;
        if (!g.locate(x, y, z, cell)) { status = fail(c, PSAMD_ERR_OUTSIDE_BOX, "fill_particles"); break; }
        const CellInfo &ci = c->celltab[(size_t)cell];
        Rec &r = recs[(size_t)done];
        // a slab places only the particles of its own segments (their queues are its own: the
        // order among them is the reference's), the others are their owners' business
        if (!owns_record(c->P, g.segment_record(ci.seg_type, ci.seg_tid))) { ids[(size_t)done] = -1; r.cell = -1; continue; }
        const int nid = host_q_remove(c, ci.seg_type, ci.seg_tid);
        if (nid < 0) { status = fail(c, PSAMD_ERR_QUEUE_EMPTY, "fill_particles"); break; }
        ids[(size_t)done] = nid;
        placed++;
        r.cell = cell;
        r.p = make_float4(x, y, z, w ? w[done] : (float)g.cfg.particle_weight);
        r.v = make_float4(vxyz ? vxyz[3 * done] : 0.f, vxyz ? vxyz[3 * done + 1] : 0.f, vxyz ? vxyz[3 * done + 2] : 0.f,
                          age ? age[done] : 0.f);
        r.a = make_float4(0.f, 0.f, 0.f, fert_age ? fert_age[done] : 0.f);
    }
    // create_particle_s overwrites every field of the slot (app.cu:189-208): ship the
    // records once and let a kernel drop them into their slots
    if (done > 0) {
        const size_t m = (size_t)done;
        std::vector<float4> hp(m), hv(m), ha(m);
        std::vector<int> hc(m);
        for (size_t k = 0; k < m; k++) { hp[k] = recs[k].p; hv[k] = recs[k].v; ha[k] = recs[k].a; hc[k] = recs[k].cell; }
        const size_t bytes = m * (3 * sizeof(float4) + 2 * sizeof(int));
        rc = ensure_staging(c, bytes);
        if (rc != PSAMD_OK) return rc;
        char *base = (char *)c->staging;
        float4 *dp = (float4 *)base, *dv = dp + m, *da = dv + m;
        int *dc = (int *)(da + m), *di = dc + m;
        PS_HIP(c, hipMemcpyAsync(dp, hp.data(), m * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        PS_HIP(c, hipMemcpyAsync(dv, hv.data(), m * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        PS_HIP(c, hipMemcpyAsync(da, ha.data(), m * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        PS_HIP(c, hipMemcpyAsync(dc, hc.data(), m * sizeof(int), hipMemcpyHostToDevice, c->stream));
        PS_HIP(c, hipMemcpyAsync(di, ids.data(), m * sizeof(int), hipMemcpyHostToDevice, c->stream));
        PS_HIP(c, launch_place(c->stream, c->P, (int)done, di, dp, dv, da, dc, c->d));
        PS_HIP(c, hipStreamSynchronize(c->stream));
    }
    rc = push_queues(c);
    if (rc != PSAMD_OK) return rc;
    if (ids_out) std::copy(ids.begin(), ids.begin() + done, ids_out);
    if (n_done) *n_done = done;
    if (c->live_bound >= 0) c->live_bound += placed;
    c->grid_built = false; c->pairs_done = false; c->slab_stage = 0;
    return status;
}

int psamd_upload_particles(psamd_ctx *c, const void *p72, int64_t first, int64_t count)
{
    if (!c || !p72 || first < 0 || count < 0 || first + count > c->geo.container) return PSAMD_ERR_INVALID_ARG;
    if (count == 0) return PSAMD_OK;
    int rc = ensure_staging(c, (size_t)count * 72);
    if (rc != PSAMD_OK) return rc;
    PS_HIP(c, hipMemcpyAsync(c->staging, p72, (size_t)count * 72, hipMemcpyHostToDevice, c->stream));
    // odd grids are not centred (G/2 is an integer division): allow the longer half
    const float half_box = (float)((c->geo.G - c->geo.G / 2) * c->geo.cfg.cell_size);
    PS_HIP(c, launch_unpack_aos(c->stream, c->P, c->staging, (int)first, (int)count, half_box, c->d));
    PS_HIP(c, hipStreamSynchronize(c->stream));
    c->grid_built = false; c->pairs_done = false; c->slab_stage = 0;
    c->live_bound = -1;
    return check_device_errors(c);
}

int psamd_download_particles(psamd_ctx *c, void *p72, int64_t first, int64_t count)
{
    if (!c || !p72 || first < 0 || count < 0 || first + count > c->geo.container) return PSAMD_ERR_INVALID_ARG;
    if (count == 0) return PSAMD_OK;
    int rc = ensure_staging(c, (size_t)count * 72);
    if (rc != PSAMD_OK) return rc;
    PS_HIP(c, launch_pack_aos(c->stream, c->P, c->staging, (int)first, (int)count, c->d));
    PS_HIP(c, hipMemcpyAsync(p72, c->staging, (size_t)count * 72, hipMemcpyDeviceToHost, c->stream));
    PS_HIP(c, hipStreamSynchronize(c->stream));
    return PSAMD_OK;
}

int psamd_download_tdata(psamd_ctx *c, void *t24, int64_t first, int64_t count)
{
    if (!c || !t24 || first < 0 || count < 0 || first + count > c->geo.container) return PSAMD_ERR_INVALID_ARG;
    if (count == 0) return PSAMD_OK;
    if (!c->tdata_mirror) return fail(c, PSAMD_ERR_STATE, "the T_DATA mirror is off (psamd_set_tdata_mirror): build_grid has not been writing the rows");
    PS_HIP(c, hipStreamSynchronize(c->stream));
    // rows of slots another rank owns: as init_particles left them (id, zeros; ps.cpp:743-748)
    uint32_t *out = (uint32_t *)t24;
    if (c->P.world > 1)
        for (int64_t i = 0; i < count; i++) { uint32_t *r = out + 6 * i; r[0] = (uint32_t)(first + i); r[1] = r[2] = r[3] = r[4] = r[5] = 0u; }
    hipError_t e = hipSuccess;
    for_owned_ranges(c, [&](size_t lo, size_t n, size_t off) {
        const int64_t a = std::max<int64_t>(first, (int64_t)lo), b = std::min<int64_t>(first + count, (int64_t)(lo + n));
        if (a < b && e == hipSuccess)
            e = hipMemcpy(out + 6 * (a - first), c->d.tdata + 6 * (off + (size_t)(a - (int64_t)lo)), (size_t)(b - a) * 24, hipMemcpyDeviceToHost);
    });
    PS_HIP(c, e);
    return PSAMD_OK;
}

int psamd_upload_queues(psamd_ctx *c, const void *qi, const int32_t *queue)
{
    if (!c || !qi || !queue) return PSAMD_ERR_INVALID_ARG;
    std::memcpy(c->h_qinfo.data(), qi, c->h_qinfo.size() * sizeof(QueueInfo));
    std::memcpy(c->h_queue.data(), queue, c->h_queue.size() * sizeof(int32_t));
    c->host_queues_valid = true;
    return push_queues(c);
}

int psamd_download_queues(psamd_ctx *c, void *qi, int32_t *queue)
{
    if (!c || !qi || !queue) return PSAMD_ERR_INVALID_ARG;
    int rc = pull_queues(c);
    if (rc != PSAMD_OK) return rc;
    std::memcpy(qi, c->h_qinfo.data(), c->h_qinfo.size() * sizeof(QueueInfo));
    std::memcpy(queue, c->h_queue.data(), c->h_queue.size() * sizeof(int32_t));
    return PSAMD_OK;
}

// Rebuild the reference's fixed-stride lists from the compact sorted arrays.  start[] is
// indexed by the own LOCAL cells (region 0); local cell lc is global cell lc + cell_off.
static int fetch_sorted(psamd_ctx *c, std::vector<int> &start, std::vector<int> &ids)
{
    if (!c->grid_built) return fail(c, PSAMD_ERR_STATE, "grid lists requested before build_grid");
    start.resize((size_t)c->P.n_own_cells + 1);
    PS_HIP(c, hipStreamSynchronize(c->stream));
    PS_HIP(c, hipMemcpy(start.data(), c->d.cell_start, start.size() * sizeof(int), hipMemcpyDeviceToHost));
    ids.resize((size_t)std::max(start.back(), 1));
    PS_HIP(c, hipMemcpy(ids.data(), c->d.sorted_id, (size_t)start.back() * sizeof(int), hipMemcpyDeviceToHost));
    return PSAMD_OK;
}

int psamd_download_cellgrid(psamd_ctx *c, int32_t *out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    std::vector<int> start, ids;
    int rc = fetch_sorted(c, start, ids);
    if (rc != PSAMD_OK) return rc;
    const Geometry &g = c->geo;
    const size_t stride = 1 + (size_t)g.max_per_cell;
    const int cell_off = c->P.reg_first[0] * g.G * g.G;
    std::memset(out, 0, sizeof(int32_t) * stride * (size_t)g.num_cells);
    for (int lc = 0; lc < c->P.n_own_cells; lc++) {
        const int n = std::min(start[(size_t)lc + 1] - start[(size_t)lc], g.max_per_cell);
        int32_t *row = out + stride * (size_t)(lc + cell_off);
        row[0] = n;
        for (int k = 0; k < n; k++) row[1 + k] = ids[(size_t)start[(size_t)lc] + k];
    }
    return PSAMD_OK;
}

int psamd_download_force_counts(psamd_ctx *c, int32_t *out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    if (!c->pairs_done) return fail(c, PSAMD_ERR_STATE, "force counts requested before the pair pass of this frame");
    PS_HIP(c, hipStreamSynchronize(c->stream));
    const Geometry &g = c->geo;
    const DevParams &P = c->P;
    const bool two = P.two_pass && P.lean_math;
    std::memset(out, 0, sizeof(int32_t) * (size_t)g.num_cells);
    std::vector<int> v((size_t)P.n_local_cells + 1);
    if (two) PS_HIP(c, hipMemcpy(v.data(), c->d.active_count, sizeof(int) * (size_t)P.n_local_cells, hipMemcpyDeviceToHost));
    else PS_HIP(c, hipMemcpy(v.data(), c->d.cell_start, sizeof(int) * ((size_t)P.n_local_cells + 1), hipMemcpyDeviceToHost));
    for (int j = 0; j < comp_count(P); j++) {
        const int lc = comp_cell(P, j);
        // one-pass modes evaluate every particle's sum (and discard what is not used)
        out[global_of_local(P, lc)] = two ? v[(size_t)lc] : std::min(v[(size_t)lc + 1] - v[(size_t)lc], g.max_per_cell);
    }
    return PSAMD_OK;
}

int psamd_download_chunkgrid(psamd_ctx *c, int32_t *out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    std::vector<int> start, ids;
    int rc = fetch_sorted(c, start, ids);
    if (rc != PSAMD_OK) return rc;
    const Geometry &g = c->geo;
    const size_t stride = 1 + (size_t)g.max_per_chunk;
    const int cell_off = c->P.reg_first[0] * g.G * g.G;
    std::memset(out, 0, sizeof(int32_t) * stride * (size_t)g.num_chunks);
    // the reference appends in slot order (ps.cpp:1502-1508): per chunk, ids ascending,
    // including the ones the cell-overflow rule then killed (stored as ~id in fetch order)
    std::vector<std::vector<int>> per((size_t)g.num_chunks);
    for (int lc = 0; lc < c->P.n_own_cells; lc++) {
        auto &v = per[(size_t)c->celltab[(size_t)(lc + cell_off)].chunk];
        for (int k = start[(size_t)lc]; k < start[(size_t)lc + 1]; k++) v.push_back(ids[(size_t)k]);
    }
    for (int ch = 0; ch < g.num_chunks; ch++) {
        auto &v = per[(size_t)ch];
        std::sort(v.begin(), v.end());
        int32_t *row = out + stride * (size_t)ch;
        row[0] = (int32_t)v.size();
        const size_t n = std::min(v.size(), (size_t)g.max_per_chunk);
        for (size_t k = 0; k < n; k++) row[1 + k] = v[k];
    }
    return PSAMD_OK;
}

int psamd_get_pkgdistrib(const psamd_ctx *c, int32_t *out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    for (int ch = 0; ch < c->geo.num_chunks; ch++) c->geo.chunk_segments(ch, (Pair *)out + (size_t)ch * 27);
    return PSAMD_OK;
}

int psamd_get_cell_table(const psamd_ctx *c, int32_t *out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    for (int i = 0; i < c->geo.num_cells; i++) {
        out[3 * i] = c->celltab[(size_t)i].chunk;
        out[3 * i + 1] = c->celltab[(size_t)i].seg_type;
        out[3 * i + 2] = c->celltab[(size_t)i].seg_tid;
    }
    return PSAMD_OK;
}

static int drain_scalars(psamd_ctx *c, bool quiet = false);

int psamd_get_gridmax(psamd_ctx *c, int32_t out2[2])
{
    if (!c || !out2) return PSAMD_ERR_INVALID_ARG;
    // inside a frame (after build_grid) the device's record is the frame's; once calc_forces has run the device's record
    // belongs to the next frame already and the step's scalars are in the host's copy (ps.cpp:1900 reads hostGridMax
    // between the stages; the reference's array keeps the build's values until the next init_iframe)
    const int rc = drain_scalars(c);
    if (rc != PSAMD_OK && !c->grid_built) return rc;
    if (c->grid_built) {
        FrameScalars fs{};
        PS_HIP(c, hipMemcpy(&fs, c->d.fs, sizeof fs, hipMemcpyDeviceToHost));
        out2[0] = fs.gridmax[0]; out2[1] = fs.gridmax[1];
        c->live_at_build = fs.live;
    } else { out2[0] = c->last.gridmax[0]; out2[1] = c->last.gridmax[1]; }
    return PSAMD_OK;
}

// ---- stages ---------------------------------------------------------------------
//
// Every stage is a fixed sequence of kernel launches whose arguments do not change from step to step
// (sizes live on the device; the step's number and the scalar records' sequence number too: StepState).
// enq_* functions only enqueue; the host-side state machine is advanced by their callers -- so that a
// sequence can be captured once into a hipGraph and replayed (psamd_set_graphs): one submission per stage
// instead of one per kernel.  A graph is keyed by what shapes its launches: the size of the balanced force
// pass (from the last task count the host has seen) and the hint of the live count the life-cycle grids are sized
// from, rounded up to 64 Ki so that a free-running population does not mean a capture per step.  Steps
// that carry timing events run eagerly (the events sit between the kernels).
//
// NOTHING in a step waits for the host: the step's tail decides everything on the device (lifecycle.hip), and the
// one read-back of a step -- live count, sticky errors, list sizes: what the reference's driver fetches as
// hostGridMax, ps.cpp:1878-1900 -- lands in a pinned record that the host reads a step late (consume_scalars).

static int slab_only(psamd_ctx *c, const char *what)
{
    return fail(c, PSAMD_ERR_STATE, std::string(what) + ": this context is one slab of a multi-rank system; step it with psamd_slab_build / "
                                                         "_pairs / _apply / _finish and exchange the messages in between");
}

static int refuse_wedged(psamd_ctx *c)
{
    return fail(c, PSAMD_ERR_STATE, "the GPU stopped answering (a step's scalars did not arrive within " + std::to_string((int)c->wait_limit_s) +
                                    " s while its stream stayed busy): this context takes no further work; destroy it");
}

// which steps carry timing events is settled when the step begins (before anything is enqueued or replayed)
static void begin_step(psamd_ctx *c)
{
    // (an event between two kernels costs ~6 us of idle GPU: a long timed run records them on every n-th step)
    c->timing_now = (c->timing && c->timing_steps++ % c->timing_period == 0) ? c->timing : 0;
    if (c->timing_now) {
        make_events(c);
        c->tset = (int)(c->timed_steps & 1);
        collect_timing(c, c->tset);          // (the set's last use is two timed steps old: this returns at once)
    }
}

static int enq_init_iframe(psamd_ctx *c)
{
    tick(c, psamd_ctx::E_RESET);
    // cell / chunk / queue-record counts and the per-frame scalars (the sticky error word stays): the last kernel of
    // the step before did it, unless there was none
    if (!c->frame_clean) PS_HIP(c, launch_frame_reset(c->stream, c->d, c->frame_ints, c->P.world > 1 ? 4 * c->geo.num_chunks : 0));
    return PSAMD_OK;
}

// is a cell with more than 1024 ids to be expected?  (the last frames the host has read; a wrong guess only costs time:
// without the crowded cells' instance the ordinary one ranks such a cell through global memory)
static bool big_cells_hint(const psamd_ctx *c) { return c->scalars_seen > 0 && c->last.max_cell_raw > 960; }
static uint64_t build_key(const psamd_ctx *c) { return (c->frame_clean ? 0ull : 1ull) | (c->tdata_mirror ? 2ull : 0ull) | (big_cells_hint(c) ? 4ull : 0ull); }

static int enq_build_grid(psamd_ctx *c)
{
    PS_HIP(c, launch_build_grid(c->stream, c->P, c->d, c->timing_now >= 2 ? &c->ev[c->tset][psamd_ctx::E_HIST] : nullptr, c->tdata_mirror, big_cells_hint(c)));
    return PSAMD_OK;
}

// size of the balanced force pass: the tasks of the last step whose scalars the host has read, else the bound of the
// live count (a pass over part of the cells gets its share of the hint)
static int64_t live_bound_of(const psamd_ctx *c);
static int64_t pairs_hint(const psamd_ctx *c, const DevParams &P)
{
    int64_t tasks_hint = (c->scalars_seen > 0 && c->tasks_last > 0) ? c->tasks_last
                         : (c->live_bound >= 0 ? c->live_bound : (int64_t)c->P.slots_total) / 64 + comp_count(c->P);
    // (high word: about how many packs of partly filled slices the pass will have -- their workgroups hold residency
    // slots of the same launch; in steps of 64 so that the launch shape does not change with every step)
    const int64_t packs = ((c->scalars_seen > 0 ? c->packs_last : 0) + 63) & ~(int64_t)63;
    return (tasks_hint * comp_count(P) / std::max(1, comp_count(c->P))) | ((packs * comp_count(P) / std::max(1, comp_count(c->P))) << 32);
}

static int enq_pairs(psamd_ctx *c, const DevParams &P, int64_t tasks_hint, bool last = true, bool first = true)
{
    if (first) tick(c, psamd_ctx::E_COLLIDE);
    PS_HIP(c, launch_pairs(c->stream, P, c->d, (c->timing_now && first) ? c->ev[c->tset][psamd_ctx::E_FORCE] : nullptr, tasks_hint, first ? 0 : 1, live_bound_of(c)));
    if (last) tick(c, psamd_ctx::E_PAIRS_END);
    return PSAMD_OK;
}

// An upper bound of the particles alive at the NEXT build_grid, as far as the host can know it (< 0 inside: unknown --
// state was uploaded -- every owned slot).  The host's figure comes from the scalars of the last step it has READ,
// which with run-ahead is not the last step enqueued: every step in between may have added a child per particle
// (explosions on) and a slab its arrivals.  Only the all-pairs far pass sizes a launch from this that must cover
// every particle; everything else takes it as a hint.
static int64_t live_bound_of(const psamd_ctx *c)
{
    int64_t b = c->live_bound >= 0 ? c->live_bound : (int64_t)c->P.slots_total;
    for (int k = c->scalars_seen; k < c->scalars_seq && b < c->P.slots_total; k++) {
        if (c->P.flags & PSAMD_FLAG_EXPLOSIONS) b *= 2;
        b += 2 * (int64_t)c->P.xfer_cap + 2 * (int64_t)c->P.xfer2_cap + (int64_t)c->P.far_cap * c->P.world;
    }
    b = std::min<int64_t>(b, c->P.slots_total);
    return c->graphs ? std::min<int64_t>((b + 65535) & ~(int64_t)65535, std::max<int64_t>(c->P.slots_total, 65536)) : b;
}

// kill / survive / integrate / explosion for every own particle; in slab mode the particles
// that leave for a neighbour's segment are in the outboxes when this has run
static int enq_apply(psamd_ctx *c, int64_t bound)
{
    tick(c, psamd_ctx::E_APPLY);
    PS_HIP(c, launch_apply(c->stream, c->P, c->S, c->d));
    if (c->P.world > 1) PS_HIP(c, launch_outbox_close(c->stream, c->P, c->d, bound, c->xfer_out));
    return PSAMD_OK;
}

// arrivals on top of the own particles: about what the op lists and move records of a step hold at most
static int64_t lifecycle_bound(const psamd_ctx *c, int64_t bound)
{
    return bound + 2 * (int64_t)c->P.xfer_cap + 2 * (int64_t)c->P.xfer2_cap + (int64_t)c->P.far_cap * c->P.world
           + (c->P.world > 1 ? (int64_t)c->P.world * STATUS_KILL_CAP : 0);
}

// Which instance replays the lists this step: the longest list of the last step the host has read is the hint (lists
// longer than the instance sorts in LDS are sorted in global memory by the same workgroup: a wrong hint only costs time).
static uint64_t pick_bucket_cap(psamd_ctx *c)
{
    const int last = c->scalars_seen > 0 ? c->last.max_bucket : 0;
    c->bucket_cap0 = last > 4096 ? BUCKET_MAX : last > 2048 ? 4096 : 2048;
    return c->bucket_cap0 > 4096 ? 2ull << 61 : c->bucket_cap0 > 2048 ? 1ull << 61 : 0ull;       // (part of a captured graph's key)
}

// free-slot queues and relocation (in slab mode: after the arrivals were merged in): census, bucketing -- the last
// bucketing workgroup hands the step's scalars to the host's pinned record --, replay + commit; the last launch
// is also the next frame's init_iframe
static int enq_lifecycle(psamd_ctx *c, int64_t bound)
{
    tick(c, psamd_ctx::E_LIFE);
    if (c->P.world > 1) PS_HIP(c, launch_inbox_merge(c->stream, c->P, c->d, c->xfer_in));
    PS_HIP(c, launch_lifecycle(c->stream, c->P, c->d, c->geo.queue_infos, lifecycle_bound(c, bound), c->bucket_cap0,
                               c->frame_ints, c->P.world > 1 ? 4 * c->geo.num_chunks : 0));
    tick(c, psamd_ctx::E_END);
    return PSAMD_OK;
}

// ---- hipGraph cache ----
enum { SEG_BUILD = 0, SEG_PAIRS, SEG_APPLY, SEG_FINISH, SEG_STEP, NSEG };

static void drop_graphs(psamd_ctx *c)
{
    for (auto &cache : c->gcache) {
        for (auto &g : cache) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        cache.clear();
    }
}

extern "C++" {
template <typename F>
static int run_segment(psamd_ctx *c, int seg, uint64_t key, F enqueue)
{
    if (!c->graphs || c->timing_now) return enqueue();
    auto &cache = c->gcache[seg];
    for (auto &g : cache)
        if (g.key == key) {
            g.stamp = ++c->gstamp;
            PS_HIP(c, hipGraphLaunch(g.exec, c->stream));
            c->graph_launches++;
            return PSAMD_OK;
        }
    // not seen with this shape: capture the sequence once, then replay it
    hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
    if (e == hipSuccess) {
        const int rc = enqueue();
        hipGraph_t graph = nullptr;
        e = hipStreamEndCapture(c->stream, &graph);
        hipGraphExec_t exec = nullptr;
        if (rc == PSAMD_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc != PSAMD_OK) return rc;
        if (e == hipSuccess && exec) {
            if (cache.size() >= 8) {                      // the shape that was used longest ago makes room
                size_t old = 0;
                for (size_t k = 1; k < cache.size(); k++) if (cache[k].stamp < cache[old].stamp) old = k;
                (void)hipGraphExecDestroy(cache[old].exec);
                cache.erase(cache.begin() + (long)old);
            }
            cache.push_back({key, exec, ++c->gstamp});
            c->graph_captures++;
            PS_HIP(c, hipGraphLaunch(exec, c->stream));
            c->graph_launches++;
            return PSAMD_OK;
        }
    }
    // the runtime would not capture this: the context goes on without graphs (nothing was executed so far)
    (void)hipGetLastError();
    c->graphs = false;
    c->graph_refused = std::string(hipGetErrorString(e));
    return enqueue();
}
}  // extern "C++"

// The host's and the device's count of the scalar records part ways if a launch fails between the kernel that
// publishes a record and the host's bookkeeping of the step: after an error both are set to what the device holds.
static void resync_scalars(psamd_ctx *c)
{
    if (hipStreamSynchronize(c->stream) != hipSuccess) return;
    StepState st{};
    if (hipMemcpy(&st, c->d.st, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return;
    c->scalars_seq = c->scalars_seen = st.seq;
}

// Wait until the scalars of step `seq` are in the host's record: the publishing workgroup stores the record's
// number last.  The stream is looked at now and then so that a failed launch cannot leave the host waiting, and
// the wall clock too: a stream that stays busy without ever publishing is a wedged GPU, reported as such
// (PSAMD_ERR_STATE, sticky) instead of a host thread that never comes back.
// Policy 0 spins on the word (lowest latency); policy 1, the default of a slab, spins for a few microseconds and
// then sleeps in short naps -- eight ranks of a node do not pin eight cores for the whole run.  The naps need a
// timer slack of ~1 us (the default 50 us would BE the nap): set for the wait, restored before it returns.
static int wait_scalars(psamd_ctx *c, int seq)
{
    volatile int32_t *word = &c->h_fs[seq & 1].seq;
    if (*word == seq) { std::atomic_thread_fence(std::memory_order_acquire); return PSAMD_OK; }
    const bool naps = c->wait_policy == 1;
    long old_slack = -1;
    int rc = PSAMD_OK;
    struct timespec t0;
    (void)clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t spins = 1; *word != seq; spins++) {
        if (naps && spins > 2000) {
            if (old_slack < 0) { old_slack = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL); (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL); }
            struct timespec ts = {0, 5000};
            (void)nanosleep(&ts, nullptr);
        } else
            __builtin_ia32_pause();
        if ((spins & (naps ? 0x3ff : 0x3fff)) == 0) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {
                if (*word == seq) break;
                rc = fail(c, PSAMD_ERR_STATE, "the step's scalars never arrived on the host");
                break;
            }
            if (e != hipErrorNotReady) { rc = hip_fail(c, e, "waiting for the step's scalars"); break; }
            struct timespec t1;
            (void)clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > c->wait_limit_s) { c->wedged = true; rc = refuse_wedged(c); break; }
        }
    }
    if (old_slack >= 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack, 0UL, 0UL, 0UL);
    std::atomic_thread_fence(std::memory_order_acquire);
    return rc;
}

// Read the records of the steps up to number `upto` (waiting for them) and whatever has arrived beyond: the host's
// bookkeeping of a step -- hints for the launches to come, the counters, and the step's verdict.
// A slab fails COLLECTIVELY: only on error bits that were in a step's all-gathered status records,
// which every rank sees alike (status_error) -- all ranks return the error from the same call.  An error this
// rank raised after its status record was closed (a message that did not fit, an arrival for a queue it does not
// hold) stays sticky, goes out with the next step's record and stops every rank there; returning it at once would
// leave the ranks that have not heard of it waiting in the next exchange.  (psamd_synchronize reports whatever is pending.)
static int consume_scalars(psamd_ctx *c, int upto)
{
    int verdict = PSAMD_OK;
    while (c->scalars_seen < c->scalars_seq) {
        const int s = c->scalars_seen + 1;
        if (s <= upto) { const int st = wait_scalars(c, s); if (st != PSAMD_OK) { if (!c->wedged) resync_scalars(c); return st; } }
        else if (*(volatile int32_t *)&c->h_fs[s & 1].seq != s) break;
        std::atomic_thread_fence(std::memory_order_acquire);
        const FrameScalars r = c->h_fs[s & 1];
        c->scalars_seen = s;
        c->last = r;
        c->live_at_build = r.live;
        const int64_t tasks_now = (int64_t)r.n_tasks2 + r.n_merged;       // ordinary tasks + packs of partial slices
        // (a pair stage in two passes -- interior_ran is noted per step below -- reports the second pass's task count)
        const bool two = c->interior_steps.count(s) != 0;
        c->interior_steps.erase(s);
        c->tasks_last = two ? tasks_now * comp_count(c->P) / std::max(1, comp_count(c->P_rest)) : tasks_now;
        c->packs_last = two ? (int64_t)r.n_merged * comp_count(c->P) / std::max(1, comp_count(c->P_rest)) : r.n_merged;
        c->live_bound = std::min<int64_t>(c->P.slots_total, (int64_t)r.live + r.n_moves);   // births and arrivals <= moves
        c->processed_total += r.live;
        c->max_bucket_seen = std::max<int64_t>(c->max_bucket_seen, r.max_bucket);
        if (c->P.world > 1 && r.xfer_cap_next > 0) c->cap_decisions[s] = r.xfer_cap_next;      // (every step's: an absolute number, the same on every rank)
        if (verdict == PSAMD_OK && (c->P.world > 1 ? r.status_error != 0 : r.error != 0)) verdict = check_device_errors(c);
    }
    if (verdict != PSAMD_OK && c->pending_status == PSAMD_OK) { c->pending_status = verdict; c->pending_err = c->err; }
    return PSAMD_OK;
}

static int take_verdict(psamd_ctx *c)
{
    if (c->pending_status == PSAMD_OK) return PSAMD_OK;
    const int st = c->pending_status;
    c->err = c->pending_err;
    c->pending_status = PSAMD_OK;
    return st;
}

// the rest of the step, once enq_lifecycle is enqueued (or replayed): the host's bookkeeping
static int finish_step(psamd_ctx *c)
{
    c->host_queues_valid = false;
    const int seq = ++c->scalars_seq;
    if (c->interior_ran) c->interior_steps.insert(seq);
    c->interior_ran = false;
    if (c->timing_now) { c->ev_level[c->tset] = c->timing_now; c->timed_steps++; }
    c->grid_built = false; c->pairs_done = false;
    c->frame_clean = true;                       // (the step's last kernel zeroed the counts for the frame that follows)
    c->step++; c->steps_total++;
    // run-ahead: this step's record is read when the NEXT step has been enqueued (the record of the step before must be
    // out of the way by then: the two pinned records alternate); else now
    const int rc = consume_scalars(c, c->run_ahead ? seq - 1 : seq);
    return rc != PSAMD_OK ? rc : take_verdict(c);
}

// everything enqueued so far has run: read every record outstanding (psamd_synchronize and the calls that hand
// state or counters to the caller)
static int drain_scalars(psamd_ctx *c, bool quiet)
{
    PS_HIP(c, hipStreamSynchronize(c->stream));
    const int rc = consume_scalars(c, c->scalars_seq);
    return rc != PSAMD_OK ? rc : quiet ? (int)PSAMD_OK : take_verdict(c);
}

int psamd_init_iframe(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->wedged) return refuse_wedged(c);
    if (c->P.world > 1) return slab_only(c, "init_iframe");
    begin_step(c);
    if (c->grid_built) c->frame_clean = false;      // (a frame abandoned after its build: its counts are in the way)
    const int rc = enq_init_iframe(c);
    if (rc != PSAMD_OK) return rc;
    c->frame_clean = true;
    c->frame_reset = true; c->grid_built = false; c->pairs_done = false;
    return PSAMD_OK;
}

int psamd_build_grid(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->wedged) return refuse_wedged(c);
    if (c->P.world > 1) return slab_only(c, "build_grid");
    if (!c->frame_reset) return fail(c, PSAMD_ERR_STATE, "build_grid needs init_iframe first");
    const int rc = enq_build_grid(c);
    if (rc != PSAMD_OK) return rc;
    c->frame_reset = false; c->grid_built = true; c->pairs_done = false; c->frame_clean = false;
    return PSAMD_OK;
}

int psamd_calc_forces_pairs(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->wedged) return refuse_wedged(c);
    if (c->P.world > 1) return slab_only(c, "calc_forces");
    if (!c->grid_built) return fail(c, PSAMD_ERR_STATE, "calc_forces needs build_grid first");
    const int rc = enq_pairs(c, c->P, pairs_hint(c, c->P));
    if (rc != PSAMD_OK) return rc;
    c->pairs_done = true;
    return PSAMD_OK;
}

int psamd_calc_forces_apply(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->wedged) return refuse_wedged(c);
    if (c->P.world > 1) return slab_only(c, "calc_forces");
    if (!c->grid_built || !c->pairs_done) return fail(c, PSAMD_ERR_STATE, "apply needs build_grid and the pair pass first");
    const int64_t bound = live_bound_of(c);
    (void)pick_bucket_cap(c);
    int rc = enq_apply(c, bound);
    if (rc == PSAMD_OK) rc = enq_lifecycle(c, bound);
    if (rc != PSAMD_OK) { resync_scalars(c); return rc; }
    return finish_step(c);
}

int psamd_calc_forces(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    int rc = psamd_calc_forces_pairs(c);
    if (rc != PSAMD_OK) return rc;
    return psamd_calc_forces_apply(c);
}

int psamd_step(psamd_ctx *c, int32_t nsteps)
{
    if (!c || nsteps < 0) return PSAMD_ERR_INVALID_ARG;
    if (c->wedged) return refuse_wedged(c);
    if (c->P.world > 1 && nsteps > 0) return slab_only(c, "step");
    for (int k = 0; k < nsteps; k++) {
        // init_iframe, build_grid, calc_forces: one sequence of launches (one graph)
        begin_step(c);
        if (c->grid_built) c->frame_clean = false;
        const int64_t hint = pairs_hint(c, c->P), bound = live_bound_of(c);
        const uint64_t key = launch_pairs_shape(c->P, hint) | ((uint64_t)bound << 24) | pick_bucket_cap(c) | (build_key(c) << 58);
        int rc = run_segment(c, SEG_STEP, key, [&]() {
            int r = enq_init_iframe(c);
            if (r == PSAMD_OK) r = enq_build_grid(c);
            if (r == PSAMD_OK) r = enq_pairs(c, c->P, hint);
            if (r == PSAMD_OK) r = enq_apply(c, bound);
            if (r == PSAMD_OK) r = enq_lifecycle(c, bound);
            return r;
        });
        c->frame_clean = false;
        if (rc != PSAMD_OK) { resync_scalars(c); return rc; }
        c->frame_reset = false; c->grid_built = true; c->pairs_done = true;
        rc = finish_step(c);
        if (rc != PSAMD_OK) return rc;
    }
    return PSAMD_OK;
}

// ---- slab stages: the step cut where neighbouring ranks exchange messages -----------------

int psamd_slab_build(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->wedged) return refuse_wedged(c);
    begin_step(c);
    if (c->grid_built) c->frame_clean = false;
    {   // The transfer messages' capacity the ranks agreed on two steps ago (k_status_merge) takes effect now, on every rank
        // in this same step: the record of step s - 2 has been read by every host that starts step s, whatever its run-ahead.
        // A decision is an absolute number (grown, kept or shrunk: the rule is k_status_merge's), the same on every rank.
        const int s = c->scalars_seq + 1;
        int cap = c->P.xfer_cap;
        for (auto it = c->cap_decisions.begin(); it != c->cap_decisions.end() && it->first <= s - 2; it = c->cap_decisions.erase(it)) cap = it->second;
        cap = std::max(c->P.xfer_cap0, std::min(cap, c->P.xfer_cap_max));
        if (cap != c->P.xfer_cap) {
            c->P.xfer_cap = c->P_int.xfer_cap = c->P_rest.xfer_cap = cap;
            c->xfer_bytes = xfer_msg_bytes(c->P.xfer_cap);
        }
    }
    const int rc = run_segment(c, SEG_BUILD, build_key(c), [&]() {
        int r = enq_init_iframe(c);
        if (r == PSAMD_OK) r = enq_build_grid(c);
        if (r != PSAMD_OK) return r;
        if (c->allg_out) PS_HIP(c, launch_allg_pack(c->stream, c->P, c->d, c->allg_out));
        if (c->P.world > 1) PS_HIP(c, launch_pack_halos(c->stream, c->P, c->d, c->halo_out_c0, c->halo_out_cells, c->halo_out, c->pack_off));
        return (int)PSAMD_OK;
    });
    if (rc != PSAMD_OK) return rc;
    c->frame_reset = false; c->grid_built = true; c->pairs_done = false; c->frame_clean = false;
    c->slab_stage = 1;
    return PSAMD_OK;
}

// optional, between slab_build and the arrival of the halo: the pair stage of the cells whose
// stencil lies inside this rank's own layers
int psamd_slab_pairs_interior(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->slab_stage != 1) return fail(c, PSAMD_ERR_STATE, "slab_pairs_interior belongs between slab_build and slab_pairs");
    if (!c->have_interior || c->interior_done) return PSAMD_OK;
    const int64_t hint = pairs_hint(c, c->P_int);
    const int rc = run_segment(c, SEG_PAIRS, launch_pairs_shape(c->P_int, hint) | (1ull << 40), [&]() {
        // (the status records have landed: the chunk lists' capacity rule over all ranks decides which particles the stage leaves alone)
        PS_HIP(c, launch_chunk_census(c->stream, c->P, c->d, c->status_in));
        return enq_pairs(c, c->P_int, hint, false, true);
    });
    if (rc != PSAMD_OK) return rc;
    c->interior_done = true; c->interior_ran = true;
    return PSAMD_OK;
}

int psamd_slab_pairs(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->slab_stage != 1) return fail(c, PSAMD_ERR_STATE, "slab_pairs needs slab_build (and the halo exchange) first");
    const DevParams &P = c->P;
    const DevParams &Pp = c->interior_done ? c->P_rest : c->P;
    const bool second = c->interior_done;
    const int64_t hint = pairs_hint(c, Pp);
    // (the all-pairs far pass sizes its launch from the live bound on one GPU only -- a slab takes every entry of the
    // sorted order, see launch_pairs -- so the bound is no part of this key)
    const int rc = run_segment(c, SEG_PAIRS, launch_pairs_shape(Pp, hint) | (second ? 2ull << 40 : 0ull), [&]() {
        const int GG = P.G * P.G;
        // from the rank below: halo layer (region 1), then lent layers (region 2); from the rank above: halo layer (region 3)
        PS_HIP(c, launch_unpack_halos(c->stream, P, c->d, c->halo_in_cells[0], c->halo_in[0], c->unpack_off[0],
                                      c->halo_in_cells[1], c->halo_in[1], c->unpack_off[1]));
        if (c->allg_in) PS_HIP(c, launch_allg_index(c->stream, P, c->d));      // all-pairs: the gathered snapshot, by global cell
        // the chunk lists' capacity rule over all ranks (the status records have landed): which particles the step leaves alone
        if (!second) PS_HIP(c, launch_chunk_census(c->stream, P, c->d, c->status_in));
        const int r = enq_pairs(c, Pp, hint, true, !second);
        if (r != PSAMD_OK) return r;
        if (c->force_out) PS_HIP(c, launch_pack_force(c->stream, P, c->d, c->force_out, P.reg_layers[2] * GG * P.halo_cap_cell));
        return (int)PSAMD_OK;
    });
    c->interior_done = false;
    if (rc != PSAMD_OK) return rc;
    c->pairs_done = true;
    c->slab_stage = 2;
    return PSAMD_OK;
}

int psamd_slab_apply(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->slab_stage != 2) return fail(c, PSAMD_ERR_STATE, "slab_apply needs slab_pairs (and the force exchange) first");
    const int64_t bound = live_bound_of(c);
    const int rc = run_segment(c, SEG_APPLY, (uint64_t)bound | ((uint64_t)c->P.xfer_cap << 32), [&]() {
        // the status records of all ranks (all-gathered since slab_build): error bits, cell-overflow kills for the
        // owner of queue record 0, the transfer messages' next capacity; in
        // the same launch the force records of the lent-out layers (the tail of the snapshot that went up)
        PS_HIP(c, launch_status_merge(c->stream, c->P, c->d, c->status_in, c->halo_out_cells[1] - (c->P.lentout_c1 - c->P.lentout_c0),
                                      c->force_in, c->pack_off[1]));
        return enq_apply(c, bound);
    });
    if (rc != PSAMD_OK) return rc;
    c->slab_bound = bound;
    c->slab_stage = 3;
    return PSAMD_OK;
}

int psamd_slab_finish(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (c->slab_stage != 3) return fail(c, PSAMD_ERR_STATE, "slab_finish needs slab_apply (and the transfer exchange) first");
    c->slab_stage = 0;
    const int64_t bound = c->slab_bound;
    const int rc = run_segment(c, SEG_FINISH, (uint64_t)bound | ((uint64_t)(c->P.xfer_cap & 0x1fffffff) << 32) | pick_bucket_cap(c), [&]() { return enq_lifecycle(c, bound); });
    if (rc != PSAMD_OK) { resync_scalars(c); return rc; }
    return finish_step(c);
}

int psamd_slab_plan_describe(const psamd_config *cfg, psamd_slab_plan *o)
{
    if (!cfg || !o) return PSAMD_ERR_INVALID_ARG;
    Geometry g;
    if (!g.init(*cfg)) return PSAMD_ERR_INVALID_ARG;
    if (cfg->world < 1 || cfg->world > PSAMD_MAX_RANKS || cfg->rank < 0 || cfg->rank >= cfg->world) return PSAMD_ERR_INVALID_ARG;
    const SlabPlan p = plan_for(g, *cfg);
    if (!p.valid) return PSAMD_ERR_UNSUPPORTED;
    std::memset(o, 0, sizeof *o);
    o->world = p.world; o->rank = p.rank; o->grid_dim = p.G;
    o->cut_lo = p.cut_lo; o->cut_hi = p.cut_hi; o->state_lo = p.state_lo; o->state_hi = p.state_hi;
    o->below_lo = p.below_lo; o->below_hi = p.below_hi; o->above_lo = p.above_lo; o->above_hi = p.above_hi;
    o->lentin_lo = p.lentin_lo; o->lentin_hi = p.lentin_hi; o->lentout_lo = p.lentout_lo; o->lentout_hi = p.lentout_hi;
    o->send_up_lo = p.send_up_lo; o->send_up_hi = p.send_up_hi; o->send_down_lo = p.send_down_lo; o->send_down_hi = p.send_down_hi;
    for (int t = 0; t < 4; t++) { o->slot_lo[t] = p.slot_lo[t]; o->slot_hi[t] = p.slot_hi[t]; o->rec_lo[t] = p.rec_lo[t]; o->rec_hi[t] = p.rec_hi[t]; }
    o->up_rank = p.up_rank; o->down_rank = p.down_rank;
    return PSAMD_OK;
}

int psamd_get_slab_plan(const psamd_ctx *c, psamd_slab_plan *o)
{
    if (!c || !o) return PSAMD_ERR_INVALID_ARG;
    return psamd_slab_plan_describe(&c->geo.cfg, o);
}

int psamd_slab_buffers_get(psamd_ctx *c, psamd_slab_buffers *o)
{
    if (!c || !o) return PSAMD_ERR_INVALID_ARG;
    std::memset(o, 0, sizeof *o);
    for (int k = 0; k < 2; k++) {
        o->halo_out[k] = c->halo_out[k]; o->halo_in[k] = c->halo_in[k];
        o->halo_out_bytes[k] = (int64_t)c->halo_out_bytes[k]; o->halo_in_bytes[k] = (int64_t)c->halo_in_bytes[k];
        o->xfer_out[k] = c->xfer_out[k]; o->xfer_in[k] = c->xfer_in[k];
    }
    for (int k = 0; k < 2; k++) { o->xfer2_out[k] = c->xfer_out[2 + k]; o->xfer2_in[k] = c->xfer_in[2 + k]; }
    o->xfer2_bytes = (int64_t)c->xfer2_bytes;
    o->far_out = c->xfer_out[4]; o->far_in = c->xfer_in[4]; o->far_bytes = (int64_t)c->far_bytes;
    o->force_out = c->force_out; o->force_in = c->force_in;
    o->force_out_bytes = (int64_t)c->force_out_bytes; o->force_in_bytes = (int64_t)c->force_in_bytes;
    o->xfer_bytes = (int64_t)c->xfer_bytes;
    o->xfer_bytes_max = c->P.world > 1 ? (int64_t)xfer_msg_bytes(c->P.xfer_cap_max) : 0;
    o->status_out = c->status_out; o->status_in = c->status_in; o->status_bytes = (int64_t)c->status_bytes;
    o->allg_out = c->allg_out; o->allg_in = c->allg_in; o->allg_bytes = (int64_t)c->allg_bytes;
    return PSAMD_OK;
}

static bool slab_msg(psamd_ctx *c, int which, int *&ptr, size_t &bytes)
{
    switch (which) {
    case 0: case 1: ptr = c->halo_out[which]; bytes = c->halo_out_bytes[which]; return true;
    case 2: case 3: ptr = c->halo_in[which - 2]; bytes = c->halo_in_bytes[which - 2]; return true;
    case 4: ptr = c->force_out; bytes = c->force_out_bytes; return true;
    case 5: ptr = c->force_in; bytes = c->force_in_bytes; return true;
    case 6: case 7: ptr = c->xfer_out[which - 6]; bytes = c->xfer_bytes; return true;
    case 8: case 9: ptr = c->xfer_in[which - 8]; bytes = c->xfer_bytes; return true;
    case 10: ptr = c->status_out; bytes = c->status_bytes; return true;
    case 11: ptr = c->status_in; bytes = c->status_bytes * (size_t)std::max(1, c->P.world); return true;
    case 12: ptr = c->allg_out; bytes = c->allg_bytes; return true;
    case 13: ptr = c->allg_in; bytes = c->allg_bytes * (size_t)std::max(1, c->P.world); return true;
    case 14: case 15: ptr = c->xfer_out[which - 12]; bytes = c->xfer2_bytes; return true;
    case 16: case 17: ptr = c->xfer_in[which - 14]; bytes = c->xfer2_bytes; return true;
    case 18: ptr = c->xfer_out[4]; bytes = c->far_bytes; return true;
    case 19: ptr = c->xfer_in[4]; bytes = c->far_bytes * (size_t)std::max(1, c->P.world); return true;
    }
    return false;
}

int psamd_slab_msg_download(psamd_ctx *c, int which, void *host, int64_t bytes)
{
    int *p = nullptr; size_t n = 0;
    if (!c || !host || !slab_msg(c, which, p, n) || bytes < 0 || (size_t)bytes > n || (!p && bytes > 0)) return PSAMD_ERR_INVALID_ARG;
    if (bytes == 0) return PSAMD_OK;
    PS_HIP(c, hipMemcpyAsync(host, p, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    PS_HIP(c, hipStreamSynchronize(c->stream));
    return PSAMD_OK;
}

int psamd_slab_msg_upload(psamd_ctx *c, int which, const void *host, int64_t bytes)
{
    int *p = nullptr; size_t n = 0;
    if (!c || !host || !slab_msg(c, which, p, n) || bytes < 0 || (size_t)bytes > n || (!p && bytes > 0)) return PSAMD_ERR_INVALID_ARG;
    if (bytes == 0) return PSAMD_OK;
    PS_HIP(c, hipMemcpyAsync(p, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    PS_HIP(c, hipStreamSynchronize(c->stream));
    return PSAMD_OK;
}

int psamd_synchronize(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    const int rc = drain_scalars(c);             // (the verdict of every step whose record had not been read yet)
    if (rc != PSAMD_OK) return rc;
    return check_device_errors(c);
}

int psamd_get_counters(psamd_ctx *c, psamd_counters *o)
{
    if (!c || !o) return PSAMD_ERR_INVALID_ARG;
    (void)drain_scalars(c, true);                // (steps / particles_processed count every step enqueued; a step's verdict is psamd_synchronize's to report)
    DevCounters copies[COUNTER_COPIES];
    PS_HIP(c, hipMemcpy(copies, c->d.ctr, sizeof copies, hipMemcpyDeviceToHost));
    DevCounters d{};
    for (const DevCounters &k : copies) {
        d.deaths_age += k.deaths_age; d.deaths_collision += k.deaths_collision; d.survives += k.survives;
        d.integrated += k.integrated; d.relocations += k.relocations; d.relocations_lost += k.relocations_lost;
        d.births += k.births; d.births_failed += k.births_failed; d.cell_overflow_kills += k.cell_overflow_kills;
    }
    o->deaths_age = (int64_t)d.deaths_age; o->deaths_collision = (int64_t)d.deaths_collision;
    o->survives = (int64_t)d.survives; o->integrated = (int64_t)d.integrated;
    o->relocations = (int64_t)d.relocations; o->relocations_lost = (int64_t)d.relocations_lost;
    o->births = (int64_t)d.births; o->births_failed = (int64_t)d.births_failed;
    o->cell_overflow_kills = (int64_t)d.cell_overflow_kills;
    o->steps = c->steps_total;
    o->particles_processed = c->processed_total;
    o->max_ops_one_queue = c->max_bucket_seen;
    return PSAMD_OK;
}

int psamd_live_count(psamd_ctx *c, int64_t *out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    PS_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<int> cells((size_t)std::max(c->P.slots_total, 1));
    PS_HIP(c, hipMemcpy(cells.data(), c->d.cell, (size_t)c->P.slots_total * sizeof(int), hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (int i = 0; i < c->P.slots_total; i++) n += (cells[(size_t)i] >= 0 && cells[(size_t)i] < c->geo.num_cells) ? 1 : 0;
    *out = n;
    return PSAMD_OK;
}

int psamd_device_view_get(psamd_ctx *c, psamd_device_view *o)
{
    if (!c || !o) return PSAMD_ERR_INVALID_ARG;
    o->pos4 = c->d.pos4; o->vel4 = c->d.vel4; o->acc4 = c->d.acc4; o->cell = c->d.cell; o->pflags = c->d.pflags;
    o->sorted_id = c->d.sorted_id; o->snap_soa = c->d.snap_soa; o->sorted_cap = c->P.sorted_cap; o->force4 = c->d.force4; o->cell_start = c->d.cell_start;
    o->container_size = c->P.slots_total; o->num_cells = c->P.n_own_cells;
    o->live = c->live_at_build;
    o->stream = (void *)c->stream;
    return PSAMD_OK;
}

int psamd_download_force4(psamd_ctx *c, void *out, int64_t first, int64_t count)
{
    if (!c || !out || first < 0 || count < 0 || first + count > c->P.sorted_cap) return PSAMD_ERR_INVALID_ARG;
    if (count == 0) return PSAMD_OK;
    int rc = ensure_staging(c, (size_t)count * sizeof(float4));
    if (rc != PSAMD_OK) return rc;
    PS_HIP(c, launch_force_gather(c->stream, c->P, c->d, c->staging, (int)first, (int)count));
    PS_HIP(c, hipMemcpyAsync(out, c->staging, (size_t)count * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    PS_HIP(c, hipStreamSynchronize(c->stream));
    return PSAMD_OK;
}

// particles (pos4, vel4, acc4, cell, pflags) + QUEUE_INFO + queue, back to back
static size_t snapshot_bytes(const psamd_ctx *c)
{
    const size_t C = (size_t)c->P.slots_total;
    return C * (3 * sizeof(float4) + sizeof(int) + 1) + (size_t)c->geo.queue_infos * sizeof(QueueInfo) + C * sizeof(int);
}

static int snapshot_copy(psamd_ctx *c, bool save)
{
    const size_t C = (size_t)c->P.slots_total;
    char *p = c->snapshot;
    char *s_pos = p, *s_vel = s_pos + C * sizeof(float4), *s_acc = s_vel + C * sizeof(float4);
    char *s_cell = s_acc + C * sizeof(float4);
    char *s_qinfo = s_cell + C * sizeof(int);
    char *s_queue = s_qinfo + (size_t)c->geo.queue_infos * sizeof(QueueInfo);
    char *s_flags = s_queue + C * sizeof(int);
    if (save) {
        struct { void *dev; char *snap; size_t bytes; } parts[] = {
            {c->d.pos4, s_pos, C * sizeof(float4)}, {c->d.vel4, s_vel, C * sizeof(float4)},
            {c->d.acc4, s_acc, C * sizeof(float4)}, {c->d.cell, s_cell, C * sizeof(int)},
            {c->d.pflags, s_flags, C},
        };
        for (auto &part : parts) PS_HIP(c, hipMemcpyAsync(part.snap, part.dev, part.bytes, hipMemcpyDeviceToDevice, c->stream));
        PS_HIP(c, hipMemcpyAsync(s_qinfo, c->d.qinfo, (size_t)c->geo.queue_infos * sizeof(QueueInfo), hipMemcpyDeviceToDevice, c->stream));
        PS_HIP(c, hipMemcpyAsync(s_queue, c->d.queue, C * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    } else {
        PS_HIP(c, launch_restore(c->stream, (int)C, s_pos, s_vel, s_acc, s_cell, s_flags, s_queue, s_qinfo,
                                 (int)((size_t)c->geo.queue_infos * sizeof(QueueInfo) / sizeof(int)), c->step, c->d));
    }
    return PSAMD_OK;
}

int psamd_snapshot_save(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (!c->snapshot) PS_HIP(c, dev_alloc(c, &c->snapshot, snapshot_bytes(c)));
    c->snapshot_step = c->step;
    c->snapshot_live_bound = c->live_bound;
    return snapshot_copy(c, true);
}

int psamd_snapshot_restore(psamd_ctx *c)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (!c->snapshot) return fail(c, PSAMD_ERR_STATE, "snapshot_restore without a saved snapshot");
    c->step = c->snapshot_step;
    c->live_bound = c->snapshot_live_bound;
    c->host_queues_valid = false;
    c->frame_reset = false; c->grid_built = false; c->pairs_done = false; c->slab_stage = 0;
    return snapshot_copy(c, false);
}

int psamd_set_stream(psamd_ctx *c, void *hip_stream)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    PS_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PSAMD_OK;
}

int psamd_get_stream(psamd_ctx *c, void **out)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    *out = (void *)c->stream;
    return PSAMD_OK;
}

int psamd_debug_wave_trace(psamd_ctx *c, uint64_t *out, int64_t n_words)
{
    if (!c || !out) return PSAMD_ERR_INVALID_ARG;
    const int64_t have = 3 * ((int64_t)c->P.n_local_cells * c->P.slices + 4);
    PS_HIP(c, hipStreamSynchronize(c->stream));
    PS_HIP(c, hipMemcpy(out, c->d.trace, (size_t)std::min(have, n_words) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PSAMD_OK;
}

int psamd_selftest_math(psamd_ctx *c, uint32_t lo_bits, uint32_t hi_bits, uint64_t out24[24])
{
    if (!c || !out24 || hi_bits < lo_bits) return PSAMD_ERR_INVALID_ARG;
    unsigned long long *d = nullptr;
    PS_HIP(c, hipMalloc((void **)&d, 26 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 26 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = launch_selftest_math(c->stream, lo_bits, hi_bits, d);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out24, d, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(c, e, "selftest_math");
    return PSAMD_OK;
}

int psamd_set_graphs(psamd_ctx *c, int enabled)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    PS_HIP(c, hipStreamSynchronize(c->stream));
    if (!enabled) drop_graphs(c);
    c->graphs = enabled != 0;
    c->graph_refused.clear();
    return PSAMD_OK;
}

int psamd_get_graph_stats(psamd_ctx *c, int64_t *launches, int64_t *captures)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    if (launches) *launches = c->graph_launches;
    if (captures) *captures = c->graph_captures;
    if (!c->graph_refused.empty()) return fail(c, PSAMD_ERR_UNSUPPORTED, "the HIP runtime would not capture a stage sequence (" + c->graph_refused + "); the context runs without graphs");
    return PSAMD_OK;
}

int psamd_set_wait_policy(psamd_ctx *c, int policy)
{
    if (!c || policy < 0 || policy > 1) return PSAMD_ERR_INVALID_ARG;
    c->wait_policy = policy;
    return PSAMD_OK;
}

int psamd_set_tdata_mirror(psamd_ctx *c, int enabled)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    c->tdata_mirror = enabled != 0;
    return PSAMD_OK;
}

int psamd_set_run_ahead(psamd_ctx *c, int steps)
{
    if (!c || steps < 0 || steps > 1) return PSAMD_ERR_INVALID_ARG;
    c->run_ahead = steps;
    return PSAMD_OK;
}

int psamd_set_timing(psamd_ctx *c, int enabled)
{
    if (!c) return PSAMD_ERR_INVALID_ARG;
    collect_timing(c, 0); collect_timing(c, 1);          // (whatever is outstanding belongs to the setting that ends here)
    c->timing = enabled < 0 ? 0 : enabled > 2 ? 2 : enabled;
    c->timing_steps = 0; c->timing_now = 0;
    if (c->timing) make_events(c);
    for (double &v : c->t_us) v = 0.0;
    for (auto &v : c->t_samples) v.clear();
    c->t_launches = 0;
    return PSAMD_OK;
}

int psamd_set_timing_period(psamd_ctx *c, int every)
{
    if (!c || every < 1) return PSAMD_ERR_INVALID_ARG;
    c->timing_period = every;
    c->timing_steps = 0;
    return PSAMD_OK;
}

int psamd_get_timing(psamd_ctx *c, double us_out[PSAMD_NUM_TIMERS], int64_t *launches)
{
    if (!c || !us_out) return PSAMD_ERR_INVALID_ARG;
    collect_timing(c, 0); collect_timing(c, 1);
    for (int k = 0; k < PSAMD_NUM_TIMERS; k++) us_out[k] = c->t_us[k];
    if (launches) *launches = c->t_launches;
    return PSAMD_OK;
}

int psamd_get_timing_stats(psamd_ctx *c, double median_us[PSAMD_NUM_TIMERS], double max_us[PSAMD_NUM_TIMERS], int64_t *samples)
{
    if (!c || !median_us || !max_us) return PSAMD_ERR_INVALID_ARG;
    collect_timing(c, 0); collect_timing(c, 1);
    for (int k = 0; k < PSAMD_NUM_TIMERS; k++) {
        std::vector<float> v = c->t_samples[k];
        median_us[k] = max_us[k] = 0.0;
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        median_us[k] = v[v.size() / 2]; max_us[k] = v.back();
    }
    if (samples) *samples = c->t_launches;
    return PSAMD_OK;
}

}  // extern "C"
