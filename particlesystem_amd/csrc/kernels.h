// kernels.h -- host-visible launch wrappers of kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"
#include "geometry.hpp"

namespace psamd {

// Container layout by segment type (slots and QUEUE_INFO records), device copy.
struct SegLayout {
    int32_t seg_base[5];
    int32_t info_base[5];
    int32_t seg_size_t[4];
};

// Every device allocation of a context.  SoA by slot unless noted.
struct DeviceState {
    // particle state, by slot (id == slot)
    float4 *pos4 = nullptr;       // x, y, z, w
    float4 *vel4 = nullptr;       // vx, vy, vz, age
    float4 *acc4 = nullptr;       // ax, ay, az, fertility_age
    int *cell = nullptr;          // -1 = free slot
    uint8_t *pflags = nullptr;    // bit0 is_parent
    uint32_t *tdata = nullptr;    // T_DATA_TYPE[container], 6 dwords each
    // free-slot queues (reference layout)
    QueueInfo *qinfo = nullptr;
    int *queue = nullptr;
    // per-frame grid
    int *cell_count = nullptr;    // [num_cells]   \ zeroed together
    int *chunk_count = nullptr;   // [num_chunks]  / by init_iframe
    uint8_t *chunk_skip = nullptr;   // [slots] 1: not in its chunk's (capped) list this frame; valid for the slots of over-cap chunks only
    int2 *chunk_segs = nullptr;      // [num_chunks * 27] (first slot, slots) of the segments a chunk's particles live in, in slot order
    FrameScalars *fs = nullptr;
    FrameScalars *fs_host = nullptr;  // the host's TWO pinned records as the device sees them (the step's number picks one; written by the queue-census kernels)
    StepState *st = nullptr;          // the step's number and the scalar records' sequence number, device-resident
    int *cell_start = nullptr;    // [num_cells+1]
    int *cursor = nullptr;        // [num_cells]
    int *task_start = nullptr;    // [num_cells+1] prefix of 64-particle slices per cell
    int *task_list = nullptr;     // [num_cells * slices] non-empty (cell, slice) tasks, cell-major
    // two-pass pair stage (lean modes): collision flags first, forces only where they are used
    int *halo_count = nullptr;    // [num_cells] collision candidates listed by the neighbours (zeroed with the frame)
    float *halo_f = nullptr;      // [3][num_cells * HALO_CAP] x, y, z of those candidates
    int *halo_id = nullptr;       // [num_cells * HALO_CAP] their slot ids
    int *snap_cid = nullptr;      // [container] sorted order: slot id, or -1 for a body that can never collide
    int *active_list = nullptr;   // [container] per cell (at cell_start[c]): sorted indices of the particles that need a force (written by k_collide_cell)
    int *active_count = nullptr;  // [num_cells] zeroed with the frame
    int *task_list2 = nullptr;    // [num_cells * slices]
    // balanced force pass: every wave walks the same number of bodies; a task may be cut at a stencil-cell boundary
    int *task_cost = nullptr;     // [local cells] bodies in the cell's stencil = what one task of the cell walks
    int *ctask_start = nullptr;   // [computed cells + 1] first entry of task_list2 of the j-th computed cell
    long long *cost_start = nullptr;  // [computed cells + 1] cost of all tasks before the j-th computed cell
    long long *wave_pos = nullptr; // [waves + 1] where every wave slot of the balanced pass starts: task index << 32 | cost already walked inside the task
    int *task_ready = nullptr;    // [num_cells * slices] hand-off flags, zeroed with the frame
    int4 *merged_tasks = nullptr; // [num_cells] cells whose leftover slices share one wave (-1: unused)
    // the merged tasks run beside k_pairs on a stream of their own (fork / join by events)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int *sorted_id = nullptr;     // [container] cell-major, id ascending inside a cell
    float *snap_soa = nullptr;    // [4][sorted_cap] sorted order: x, y, z, w_eff as four arrays (what the pair walk streams)
    float *snap_age = nullptr;    // [container] sorted order
    float4 *force4 = nullptr;     // [sorted_cap] sorted order: (ax, ay, az, flag) of the lent region's particles; the hand-off mailbox of the force pass
    int *cell_order = nullptr;    // [n_own_cells] the own (local) cells by (chunk, segment type, cell in the segment): cells that share a segment's slots side by side
    uint8_t *flag_slot = nullptr; // [slots] by slot: the step's collision flag of the own cells' particles (their new acceleration goes into acc4.xyz: ForceBuf, kernels_common.hpp)
    CellInfo *celltab = nullptr;  // [num_cells]
    // lifecycle
    uint64_t *op_keys = nullptr, *op_keys_sorted = nullptr;
    int *op_args = nullptr, *op_args_sorted = nullptr;
    int ops_cap = 0;
    int *rec_count = nullptr;     // [queue_infos] zeroed with the frame
    int *rec_start = nullptr;     // [queue_infos + 1]
    int *rec_cursor = nullptr;    // [queue_infos]
    MoveRec *moves = nullptr;
    int moves_cap = 0;
    float4 *stage = nullptr;      // 3 float4 per move
    XferRec *xfer_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // slab mode: records leaving for the rank below / above, two ranks below / above (inside the messages)
    int *status_out = nullptr;    // slab mode: [0] cell-overflow kills this frame, [1] error bits, [2] live, [16..] the killed slot ids
    // all-pairs across ranks: the gathered snapshot blocks (inside the context's message buffer) and their index by global cell
    const int *allg_in = nullptr;
    int *gstart = nullptr, *gn = nullptr;
    float4 *part_acc = nullptr;   // all-pairs: [ALLP_PARTS][part_tasks * 64] partial sums of the far pass's (dense task, part) waves
    int part_tasks = 0;
    int *act_start = nullptr;     // all-pairs: [num_cells + 1] particles that need a force in the pass's cells before its j-th
    int *dense_gi = nullptr;      // all-pairs: [container] sorted index of the r-th particle that needs a force (cell order)
    int *dense_cell = nullptr;    // all-pairs: [container] its cell
    DevCounters *ctr = nullptr;
    unsigned long long *trace = nullptr;  // 3 words per pair-kernel wave slot (diagnostic builds only)
};

hipError_t launch_unpack_aos(hipStream_t st, const DevParams &P, const void *aos, int first, int count, float half_box,
                             const DeviceState &d);
hipError_t launch_pack_aos(hipStream_t st, const DevParams &P, void *aos, int first, int count, const DeviceState &d);
hipError_t launch_place(hipStream_t st, const DevParams &P, int n, const int *ids, const float4 *p, const float4 *v, const float4 *a,
                        const int *cells, const DeviceState &d);
hipError_t launch_fill_int(hipStream_t st, int *p, int v, size_t n);
hipError_t launch_restore(hipStream_t st, int n, const void *s_pos, const void *s_vel, const void *s_acc,
                          const void *s_cell, const void *s_flags, const void *s_queue, const void *s_qinfo, int qinfo_words,
                          int step, const DeviceState &d);
hipError_t launch_validate_eps(hipStream_t st, uint32_t lo_bits, uint32_t hi_bits, double eps2, float eps2f,
                               unsigned long long *out);
hipError_t launch_selftest_math(hipStream_t st, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out24);
hipError_t launch_init_tdata(hipStream_t st, const DevParams &P, const DeviceState &d);
// ev (optional) = 5 events recorded before hist, scan, scatter, sort and after sort.  tdata_rows: also write the reference's T_DATA rows (a mirror for
// psamd_download_tdata; nothing in the step reads them).  big_cells: launch the instance for cells of more than 1024 ids
// (a hint: without it such a cell is ranked through global memory by the ordinary instance).
hipError_t launch_build_grid(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t *ev, bool tdata_rows, bool big_cells);
// psamd_download_force4: entries [first, first + count) of the sorted order, gathered from where the records live
hipError_t launch_force_gather(hipStream_t st, const DevParams &P, const DeviceState &d, void *out, int first, int count);
// tasks_hint: about how many force tasks the pass will have (sizes the balanced force pass)
// pass: 0 or 1, which of a frame's (up to two) passes of the pair stage this is
// live_bound: at most so many particles are alive (< 0: unknown); sizes the all-pairs far pass's launch
hipError_t launch_pairs(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t ev_force, int64_t tasks_hint, int pass, int64_t live_bound);
// what shapes launch_pairs' launches for this hint, as a number below 2^24 (the key of a captured graph)
uint64_t launch_pairs_shape(const DevParams &P, int64_t tasks_hint);
hipError_t launch_apply(hipStream_t st, const DevParams &P, const SegLayout &S, const DeviceState &d);
hipError_t launch_frame_reset(hipStream_t st, const DeviceState &d, size_t frame_ints, int status_table);   // also clears the status record's header and census table
// The step's tail behind k_apply: census of the queue operations (+ relocation phase 1), bucketing (the step's scalars go
// out to the host's pinned record), replay + commit + the next frame's init_iframe.  live_hint sizes grid-stride
// launches and nothing else; cap0 (2048 / 4096 / 8192) picks the replay instance -- lists longer than it are sorted in
// global memory by the same workgroup.  frame_ints / status_table: what the next frame's reset zeroes.
hipError_t launch_lifecycle(hipStream_t st, const DevParams &P, const DeviceState &d, int nrec, int64_t live_hint, int cap0,
                            size_t frame_ints, int status_table);
// slab exchange (messages are int arrays with a 16-word header, see kernels.hip)
// the snapshots for the rank below (k = 0) / above (k = 1), both in one pair of launches; also closes the status record
hipError_t launch_pack_halos(hipStream_t st, const DevParams &P, const DeviceState &d, const int c0[2], const int ncell[2],
                             int *const msg[2], int *const pack_off[2]);
hipError_t launch_unpack_halos(hipStream_t st, const DevParams &P, const DeviceState &d, int ncell_below, const int *msg_below,
                               int *off_below, int ncell_above, const int *msg_above, int *off_above);
hipError_t launch_pack_force(hipStream_t st, const DevParams &P, const DeviceState &d, int *msg, int cap_bodies);
hipError_t launch_outbox_close(hipStream_t st, const DevParams &P, const DeviceState &d, int64_t live_hint, int *const msgs[5]);
hipError_t launch_inbox_merge(hipStream_t st, const DevParams &P, const DeviceState &d, const int *const msgs[5]);
hipError_t launch_allg_pack(hipStream_t st, const DevParams &P, const DeviceState &d, int *msg);
hipError_t launch_allg_index(hipStream_t st, const DevParams &P, const DeviceState &d);
// the chunk lists' capacity rule over all ranks' status records (start of the pair stage)
hipError_t launch_chunk_census(hipStream_t st, const DevParams &P, const DeviceState &d, const int *status_all);
// status records of all ranks (error bits, cell-overflow kills, the transfer messages' next capacity) + the force records of the lent-out layers (force_msg, may be null)
hipError_t launch_status_merge(hipStream_t st, const DevParams &P, const DeviceState &d, const int *status_all,
                               int force_j0, const int *force_msg, const int *pack_off);

}  // namespace psamd
