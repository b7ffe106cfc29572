// geometry.hpp -- host-side mirror of the reference's configuration arithmetic and
// cell / chunk / segment index math for the step hot path.
//
// The reference fixes all of this at compile time (common.h:12-70) and evaluates
// the index math per particle (get_cell_info, app_common.cu:50-148).  Here the
// configuration is a runtime struct and the per-cell results are tabulated once
// per context and uploaded, so device code does a table lookup instead.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/psamd.h"

namespace psamd {

struct CellInfo { int32_t chunk, seg_type, seg_tid, pad; };  // one int4 per cell on device
struct Pair { int32_t c, p; };                                 // PAIR, common.h:141-145
struct QueueInfo { int32_t front, rear, count, lock, rloc, seg_size; };  // QUEUE_INFO, common.h:134-139

// Segment types are 1, 2, 4, 8 (interior / face / edge / corner of a chunk):
// the product of per-axis factors 1 (inside) or 2 (first/last cell layer).
inline int seg_index(int seg_type) {
    switch (seg_type) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; default: return -1; }
}

struct Geometry {
    psamd_config cfg{};
    int F = 0, D = 0, G = 0;           // chunk_factor, chunk_dim, grid_dim
    int num_cells = 0, num_chunks = 0, cells_per_chunk = 0;
    int max_per_cell = 0, max_per_chunk = 0;
    int seg_cells[4]{}, seg_count[4]{}, seg_size_t[4]{}, seg_size[4]{};
    int seg_base[5]{};                 // first slot of each type's region in the container
    int info_base[5]{};                // first QUEUE_INFO record of each type
    int container = 0, queue_infos = 0;
    double particle_life = 0, kid_age = 0, min_fert = 0, max_fert = 0, min_adult = 0, max_adult = 0;

    // common.h:20-50, 58-65
    bool init(const psamd_config &c) {
        cfg = c;
        F = c.chunk_factor; D = c.chunk_dim;
        if (F < 1 || D < 3 || c.max_particles_num < 1 || c.x_factor < 1) return false;
        if (!(c.cell_size > 0) || !(c.dt > 0)) return false;
        G = F * D;
        const int64_t cells64 = (int64_t)G * G * G;
        if (cells64 > (1 << 24)) return false;
        num_cells = (int)cells64;
        num_chunks = F * F * F;
        cells_per_chunk = D * D * D;
        max_per_cell = (c.max_particles_num / num_cells + 1) * c.x_factor;
        max_per_chunk = max_per_cell * cells_per_chunk;
        const int m = D - 2;
        seg_cells[0] = m * m * m; seg_cells[1] = 2 * m * m; seg_cells[2] = 4 * m; seg_cells[3] = 8;
        seg_count[0] = F * F * F;
        seg_count[1] = 3 * F * F * (F + 1);
        seg_count[2] = 3 * F * (F + 1) * (F + 1);
        seg_count[3] = (F + 1) * (F + 1) * (F + 1);
        int64_t total = 0;
        seg_base[0] = 0; info_base[0] = 0;
        for (int k = 0; k < 4; k++) {
            seg_size_t[k] = seg_cells[k] * max_per_cell;
            const int64_t sz = (int64_t)seg_count[k] * seg_size_t[k];
            total += sz;
            if (total > INT32_MAX) return false;
            seg_size[k] = (int)sz;
            seg_base[k + 1] = seg_base[k] + seg_size[k];
            info_base[k + 1] = info_base[k] + seg_count[k];
        }
        container = seg_base[4];
        queue_infos = info_base[4];
        particle_life = c.life_steps * c.dt;
        kid_age = particle_life / 10.0;
        min_fert = particle_life / 6.0;
        max_fert = particle_life * 2.0;
        min_adult = particle_life / 7.0;
        max_adult = particle_life / 2.0;
        return true;
    }

    // slot -> owning segment (get_id_info, app.cu:24-65)
    bool slot_segment(int slot, int &seg_type, int &seg_tid) const {
        if (slot < 0 || slot >= container) return false;
        int k = 0;
        while (slot >= seg_base[k + 1]) k++;
        seg_type = 1 << k;
        seg_tid = (slot - seg_base[k]) / seg_size_t[k];
        return true;
    }
    // unknown types collapse to 0 like the reference's switches (app_common.cu:6-48)
    int segment_first_slot(int seg_type, int seg_tid) const {
        const int k = seg_index(seg_type);
        return k < 0 ? 0 : seg_base[k] + seg_tid * seg_size_t[k];
    }
    int segment_record(int seg_type, int seg_tid) const {
        const int k = seg_index(seg_type);
        return k < 0 ? 0 : info_base[k] + seg_tid;
    }

    // Which boundary layer of its chunk a cell coordinate lies on, and the index
    // of the grid plane between chunks it belongs to (0..F).
    struct Axis { int chunk, layer, plane; };  // layer: 1 inside, 2 on a chunk face
    Axis axis(int i) const {
        Axis a;
        a.chunk = i / D;
        const int r = i % D;
        if (r == 0)          { a.layer = 2; a.plane = a.chunk; }
        else if (r == D - 1) { a.layer = 2; a.plane = a.chunk + 1; }
        else                 { a.layer = 1; a.plane = a.chunk; }
        return a;
    }

    // cell -> (chunk, segment), get_cell_info, app_common.cu:50-148.
    // Numbering of segments inside a type follows the reference so that slot
    // ranges, queue records and the pkg table stay interchangeable with it.
    CellInfo cell_info(int cell) const {
        const int i3 = cell / (G * G), rem = cell % (G * G), i1 = rem / G, i2 = rem % G;
        const Axis a1 = axis(i1), a2 = axis(i2), a3 = axis(i3);
        const int FF = F * F, S2 = 2 * F * (F + 1), E = (F + 1) * (F + 1);
        CellInfo ci;
        ci.chunk = a3.chunk * FF + a1.chunk * F + a2.chunk;
        ci.seg_type = a1.layer * a2.layer * a3.layer;
        ci.pad = 0;
        const int t1 = a1.plane, t2 = a2.plane, t3 = a3.plane;
        int tid = -1;
        switch (ci.seg_type) {
        case 1: tid = t3 * FF + t1 * F + t2; break;
        case 2:  // one face: numbered layer by layer in i3, faces normal to i3 first
            if (a3.layer == 2)      tid = t3 * (FF + S2) + t1 * F + t2;
            else if (a2.layer == 2) tid = (t3 + 1) * FF + t3 * S2 + (t1 + 1) * F + t1 * (F + 1) + t2;
            else                    tid = (t3 + 1) * FF + t3 * S2 + t1 * (2 * F + 1) + t2;
            break;
        case 4:  // one edge: the axis it runs along is the one with layer == 1
            if (a3.layer == 1)      tid = (t3 + 1) * S2 + t3 * E + t1 * (F + 1) + t2;
            else if (a2.layer == 1) tid = t3 * S2 + t3 * E + t1 * (2 * F + 1) + t2;
            else                    tid = t3 * S2 + t3 * E + t1 * (2 * F + 1) + F + t2;
            break;
        case 8: tid = t3 * E + t1 * (F + 1) + t2; break;
        }
        ci.seg_tid = tid;
        return ci;
    }

    std::vector<CellInfo> cell_table() const {
        std::vector<CellInfo> t((size_t)num_cells);
        for (int c = 0; c < num_cells; c++) t[(size_t)c] = cell_info(c);
        return t;
    }

    // The 27 segments a chunk and its halo touch (set_pkg_segments,
    // app_common.cu:150-232): 1 interior, 6 faces, 12 edges, 8 corners.
    void chunk_segments(int chunk, Pair *out27) const {
        const int i3 = chunk / (F * F), rem = chunk % (F * F), i1 = rem / F, i2 = rem % F;
        const int E = (F + 1) * (F + 1), S = (F + 1) * F, FF = F * F;
        const int layer4 = 2 * S + E, layer2 = FF + 2 * S;
        int n = 0;
        out27[n++] = {1, chunk};
        const int f0 = i3 * layer2 + i1 * F + i2;              // face below (normal i3)
        const int f1 = i3 * layer2 + FF + i1 * (2 * F + 1) + i2;
        const int faces[6] = {f0, f1, f1 + F, f1 + F + 1, f1 + 2 * F + 1, f0 + layer2};
        for (int v : faces) out27[n++] = {2, v};
        const int e0 = i3 * layer4 + i1 * (2 * F + 1) + i2;
        const int e4 = i3 * layer4 + 2 * S + i1 * (F + 1) + i2;
        const int lower[4] = {e0, e0 + F, e0 + F + 1, e0 + 2 * F + 1};
        for (int v : lower) out27[n++] = {4, v};
        const int mid[4] = {e4, e4 + 1, e4 + 1 + F, e4 + 2 + F};
        for (int v : mid) out27[n++] = {4, v};
        for (int v : lower) out27[n++] = {4, v + layer4};
        const int c0 = i3 * E + i1 * (F + 1) + i2;
        const int corners[4] = {c0, c0 + 1, c0 + F + 1, c0 + F + 2};
        for (int v : corners) out27[n++] = {8, v};
        for (int v : corners) out27[n++] = {8, v + E};
    }

    // q_start_fast (ps.cpp:814-871): every slot free, record k spans segment k
    void initial_queues(std::vector<QueueInfo> &info, std::vector<int32_t> &queue) const {
        info.resize((size_t)queue_infos);
        queue.resize((size_t)container);
        for (int i = 0; i < container; i++) queue[(size_t)i] = i;
        int rec = 0;
        for (int k = 0; k < 4; k++)
            for (int j = 0; j < seg_count[k]; j++, rec++) {
                const int rloc = seg_base[k] + j * seg_size_t[k];
                info[(size_t)rec] = {rloc, rloc + seg_size_t[k] - 1, seg_size_t[k], 0, rloc, seg_size_t[k]};
            }
    }

    // Position -> cell index triple as the reference computes it (double floor,
    // axis mapping i1 <- -y, i2 <- +x, i3 <- -z; app.cu:126-128, ps.cpp:921-923).
    bool locate(float x, float y, float z, int &cell) const {
        const double cs = cfg.cell_size;
        const int i1 = (int)(std::floor((-1.0 * y) / cs) + (G / 2));
        const int i2 = (int)(std::floor((1.0 * x) / cs) + (G / 2));
        const int i3 = (int)(std::floor((-1.0 * z) / cs) + (G / 2));
        if (i1 < 0 || i1 >= G || i2 < 0 || i2 >= G || i3 < 0 || i3 >= G) return false;
        cell = i3 * G * G + i1 * G + i2;
        return true;
    }
};

// smallest float f with (double)f >= v : for a float a, ((double)a < v) <=> (a < f)
inline float float_ceil(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    return f;
}
// largest float f with (double)f <= v : for a float a, ((double)a > v) <=> (a > f)
inline float float_floor(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    return f;
}

}  // namespace psamd
