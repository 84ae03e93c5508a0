// lt_api.cpp -- host side of the C ABI declared in include/lt.h.
//
// Built with -ffp-contract=off: the derived scene constants (1/mu_t, mu_a/mu_t,
// triangle edges and normals) are single IEEE operations, as in the reference's
// NumPy float64 code (e.g. primitives.py:105-111).
//
// There is no CPU fallback in this library: every compute entry point runs a
// gfx950 kernel or fails.
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "lt_internal.hpp"

using namespace ltk;

// mesh tables up to this size are staged into LDS by every workgroup; larger meshes are traversed in place
static const size_t kMeshLdsBudget = 64 * 1024;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    hipError_t ensure(size_t n)
    {
        if (n <= bytes && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e == hipSuccess) bytes = n ? n : 16;
        return e;
    }
};

}  // namespace

// One lane of the log pipeline: a stream with its own deposit log, ping-pong copy and bookkeeping tables.  A launch
// uses lane 0 (the ctx stream) alone, or alternates its batches between lanes 0 and 1 so that the bandwidth-bound
// reduction of one batch runs beside the VALU-bound walk of the next (lt_set_overlap).
constexpr int kMaxLanes = 3;      // lanes a launch may be spread over (lt_set_overlap)
constexpr int kAutoLanes = 2;     // auto mode tries 1 and 2: a third lane measured no gain (profiles/r02c_lanes_1_2_3.log)

struct LogLane {
    hipStream_t stream = nullptr;
    // tail split: the stream the tail kernel of a batch runs on (beside the lane's log reduction), its pool and events
    hipStream_t tail_stream = nullptr;
    hipEvent_t ev_bulk = nullptr, ev_tail = nullptr;
    hipEvent_t ev_done = nullptr;          // end of the lane's last batch
    std::vector<hipEvent_t> evs;           // 5 per batch of the last launch: walk start / end, scan end, partition end, reduce end
    size_t ev_used = 0;
    DevBuf log_idx, log_val, tmp_idx, tmp_val, log_fill, meta, hist1, hist_unused, hist, bin_base, bin_cnt, tile_base, tile_cnt, cursor1,
        cursor2, items2, items_c, items_r, itab, head, pool, pool_n;
    size_t alloc_records = 0;              // capacity of the log buffers currently allocated (without the slack)
    int alloc_elem = 0;
    void release_log() { log_idx.release(); log_val.release(); tmp_idx.release(); tmp_val.release(); log_fill.release(); alloc_records = 0; }
    void release_all()
    {
        release_log(); meta.release(); hist1.release(); hist_unused.release(); hist.release(); bin_base.release(); bin_cnt.release(); tile_base.release();
        tile_cnt.release(); cursor1.release(); itab.release();
        cursor2.release(); items2.release(); items_c.release(); items_r.release(); head.release(); pool.release(); pool_n.release();
    }
};

struct lt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr;
    bool stages_valid = false;
    uint64_t last_records = 0, last_overflow = 0, last_batches = 0;
    int last_lanes = 1;
    bool log_stats_pending = false;
    uint64_t pending_photons = 0;
    std::string err;
    hipDeviceProp_t prop;

    // scene (host copies)
    std::vector<lt_medium> media;
    std::vector<double> z_bounds;
    std::vector<int32_t> layer_medium;
    double n_above = 1.0, n_below = 1.0;
    std::vector<double> verts;
    std::vector<int32_t> med_front, med_back;
    std::vector<lt_bvh_node> nodes;
    std::vector<lt_surface_material> surf_mats;
    std::vector<lt_point_light> lights;
    bool have_layers = false, have_mesh = false, have_grid = false, have_source = false;
    int nx = 0, ny = 0, nz = 0, tally = LT_TALLY_F32;
    double origin[3] = {0, 0, 0}, voxel[3] = {1, 1, 1};
    int src_type = LT_SRC_PENCIL, start_medium = 0;
    double src_pos[3] = {0, 0, 0}, src_dir[3] = {0, 0, 1}, src_extra[6] = {0, 0, 0, 0, 0, 0};
    uint32_t max_steps = 1000000;
    int quantity = LT_QUANTITY_ABSORBED;
    uint32_t max_vertices = 0, captured_max_vertices = 0;
    int tally_mode = 2;                 // 0: global atomics, 1: deposit log + partition + tile reduce, 2: auto (default)
    size_t log_budget = (size_t)16 << 30, default_log_budget = (size_t)16 << 30;   // bytes for the logs and their ping-pong copies
    double rec_per_photon = 0.0;        // measured deposit records per photon (sizes the logs and the batches)
    int overlap_mode = 0;               // lt_set_overlap: 0 auto, 1 one lane, 2 two lanes
    double auto_ms_per_photon[kMaxLanes] = {0.0, 0.0, 0.0};   // overlap auto: device time per photon measured with 1 / 2 / 3 lanes
    uint64_t auto_photons[kMaxLanes] = {0, 0, 0};             //               ... and the size of the launch it was measured on
    int auto_pending = -1;              // which of the two the launch in flight is measuring (-1: none)
    uint64_t captured_photons = 0;
    int blocks_per_cu = 0, threads_per_block = 0;

    // device buffers
    DevBuf d_media[2], d_zb[2], d_if[2], d_lm, d_tris[2], d_nodes[2];  // [0]=f64, [1]=f32
    DevBuf d_grid, d_counters, d_head, d_table, d_scratch_in, d_scratch_out, d_scratch_aux;
    DevBuf d_mats, d_lights, d_r0, d_r1, d_lc, d_img, d_xy, d_vtx, d_vcnt, d_clear, d_job, d_gridx[kMaxLanes - 1];   // d_gridx: private grids of lanes 1, 2
    int cn[3] = {0, 0, 0};
    double corg[3] = {0, 0, 0}, ccell[3] = {1, 1, 1};
    bool have_clear = false;
    // march grid (meshes whose f64 tables exceed the LDS budget): cell records, candidate lists, scratch of the builder
    DevBuf d_mcell, d_mlist, d_mcoarse, d_links;      // d_links: front-to-back threading of the BVH (bvh_octant_links)
    MarchGrid mgrid;
    bool have_march = false, have_links = false;
    size_t march_entries = 0;
    LogLane lanes[kMaxLanes];
    // hot-tile form of the two-pass partition (LogReduceParams::dmap): the map is made once per scene from the tile
    // counts a batch on lane 0 left behind (normally the pilot batch) and shared by every lane
    DevBuf d_dmap, d_dmeta;
    bool dmap_valid = false, tile_cnt_ready = false, last_hot = false;
    uint32_t dmap_tiles = 0, dmap_bits2 = 0;      // the partition geometry the map was made for
    bool tables_dirty = true, media_dirty = false;
    // Experiment knobs (lt_set_tuning; -1 = built-in default).  The environment variable LT_<KEY IN CAPITALS> seeds each of
    // them ONCE, in lt_create: nothing in this library reads the environment after that.
    struct Knobs {
        long query_min = -1, log_bits2 = -1, log_hot = -1, overlap_walk_bpc = -1, diag_no_tally = -1, log_timing = -1,
             march_cells = -1, march_scale_milli = -1, no_march = -1, no_clearance = -1, no_near_lists = -1,
             clearance_cells = -1, march_info = -1, force_march = -1, tail_split = -1, part_alone = -1, serial_walks = -1, part_lds = -1;
        std::string overlap_pattern;      // LT_OVERLAP_PATTERN (relative sub-batch sizes; tools/pattern_ab.py)
    } knob;
    long* knob_by_name(const char* key)
    {
        static const struct { const char* k; long Knobs::*m; } tab[] = {
            {"query_min", &Knobs::query_min}, {"log_bits2", &Knobs::log_bits2}, {"log_hot", &Knobs::log_hot},
            {"overlap_walk_bpc", &Knobs::overlap_walk_bpc}, {"diag_no_tally", &Knobs::diag_no_tally}, {"log_timing", &Knobs::log_timing},
            {"march_cells", &Knobs::march_cells}, {"march_scale_milli", &Knobs::march_scale_milli}, {"no_march", &Knobs::no_march},
            {"no_clearance", &Knobs::no_clearance}, {"no_near_lists", &Knobs::no_near_lists}, {"clearance_cells", &Knobs::clearance_cells},
            {"march_info", &Knobs::march_info}, {"force_march", &Knobs::force_march}, {"tail_split", &Knobs::tail_split}, {"part_alone", &Knobs::part_alone}, {"serial_walks", &Knobs::serial_walks}, {"part_lds", &Knobs::part_lds}};
        for (const auto& t : tab) if (std::strcmp(key, t.k) == 0) return &(knob.*(t.m));
        return nullptr;
    }
    bool on(long v) const { return v > 0; }
    bool timed = false;

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    int hip(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return LT_OK;
        return fail(LT_E_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    size_t grid_elem() const { return tally == LT_TALLY_F32 ? 4 : 8; }
    size_t n_vox() const { return (size_t)nx * (size_t)ny * (size_t)nz; }
    void scene_changed() { rec_per_photon = 0.0; for (double& a : auto_ms_per_photon) a = 0.0; for (uint64_t& a : auto_photons) a = 0; auto_pending = -1; dmap_valid = tile_cnt_ready = false; }
};

// Front-to-back threading of a flattened pre-order BVH for each of the 8 sign patterns of a ray direction (bit k set: the
// direction is negative on axis k): first[i] = the child of interior node i that lies on the ray's side of the split plane --
// the second child if the direction is negative on the node's axis, else the first (S/bvh_new.py:455-458) -- and after[i] = the
// node the search goes to once the subtree of i is done (n = finished).  Layout [pattern][first | after][node], int16.
static bool bvh_octant_links(const std::vector<lt_bvh_node>& nd, std::vector<int16_t>& out)
{
    const int n = (int)nd.size();
    if (n <= 0 || n > 32767) return false;
    out.assign((size_t)16 * n, (int16_t)n);
    for (int oct = 0; oct < 8; oct++) {
        int16_t* first = &out[(size_t)oct * 2 * n]; int16_t* after = first + n;
        std::vector<std::pair<int, int>> todo{{0, n}};     // (node, where to go after its subtree)
        while (!todo.empty()) {
            const auto [i, a] = todo.back(); todo.pop_back();
            after[i] = (int16_t)a;
            if (nd[(size_t)i].n_prims > 0) continue;
            const int c0 = i + 1, c1 = nd[(size_t)i].offset;
            const bool neg = (oct >> nd[(size_t)i].axis) & 1;
            const int near_ = neg ? c1 : c0, far_ = neg ? c0 : c1;
            first[i] = (int16_t)near_;
            todo.emplace_back(near_, far_); todo.emplace_back(far_, a);
        }
    }
    return true;
}

#define CHECK_CTX(c) do { if (!(c)) return LT_E_INVALID; } while (0)
#define HIP_TRY(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (c)->hip(e_, #call); } while (0)
#define BIND(c) HIP_TRY(c, hipSetDevice((c)->device))

namespace {

template <typename R>
void fill_media(const std::vector<lt_medium>& in, int quantity, std::vector<MedD<R>>& out)
{
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        const double mu_t = in[i].mu_a + in[i].mu_s;
        MedD<R>& m = out[i];
        m.mu_t = (R)mu_t;
        m.inv_mu_t = (R)(mu_t > 0 ? 1.0 / mu_t : 0.0);
        m.absorb = (R)(mu_t > 0 ? in[i].mu_a / mu_t : 0.0);
        m.g = (R)in[i].g;
        m.n = (R)in[i].n;
        const R g = m.g;  // derived in walk precision, one operation each
        m.one_m_g2 = (R)1 - g * g;
        m.one_p_g2 = (R)1 + g * g;
        m.inv_2g = g != 0 ? (R)1 / ((R)2 * g) : (R)0;
        m.dep = quantity == LT_QUANTITY_FLUENCE ? m.inv_mu_t : m.absorb;
        m.one_m_g = (R)1 - g; m.two_g = (R)2 * g;
        m.pad_ = 0;
    }
}

// PreComputedTriangle fields (primitives.py:105-111) in float64, then narrowed
template <typename R>
void fill_tris(const std::vector<double>& v, const std::vector<int32_t>& mf, const std::vector<int32_t>& mb,
               std::vector<TriD<R>>& out)
{
    const size_t n = v.size() / 9;
    out.resize(n);
    for (size_t i = 0; i < n; i++) {
        const double* a = &v[9 * i];
        const double* b = a + 3;
        const double* c = a + 6;
        double e1[3], e2[3], nn[3];
        for (int k = 0; k < 3; k++) { e1[k] = b[k] - a[k]; e2[k] = c[k] - a[k]; }
        nn[0] = e1[1] * e2[2] - e1[2] * e2[1];
        nn[1] = e1[2] * e2[0] - e1[0] * e2[2];
        nn[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const double l = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        for (int k = 0; k < 3; k++) {
            out[i].a[k] = (R)a[k]; out[i].ab[k] = (R)e1[k]; out[i].ac[k] = (R)e2[k]; out[i].n[k] = (R)(nn[k] / l);
        }
        out[i].med_front = mf[i];
        out[i].med_back = mb[i];
    }
}

template <typename R>
void fill_nodes(const std::vector<lt_bvh_node>& in, std::vector<NodeD<R>>& out)
{
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        for (int k = 0; k < 3; k++) {
            R lo = (R)in[i].lo[k], hi = (R)in[i].hi[k];
            // keep the box conservative when narrowing
            if ((double)lo > in[i].lo[k]) lo = std::nextafter(lo, -std::numeric_limits<R>::infinity());
            if ((double)hi < in[i].hi[k]) hi = std::nextafter(hi, std::numeric_limits<R>::infinity());
            out[i].lo[k] = lo; out[i].hi[k] = hi;
        }
        out[i].offset = in[i].offset;
        out[i].n_prims = in[i].n_prims;
        out[i].axis = in[i].axis;
        out[i].skip = (int32_t)in.size();
    }
    // skip links: pre-order layout => left child = i + 1 ends where the right child (offset) begins,
    // the right child ends where its parent does (parents come before children, so one forward sweep)
    for (size_t i = 0; i < in.size(); i++) {
        if (in[i].n_prims > 0) continue;
        out[i + 1].skip = in[i].offset;
        out[(size_t)in[i].offset].skip = out[i].skip;
    }
}

// interface table of a layered slab (IfD): the indices either side of every plane and their quotients, one IEEE operation
// each in walk precision (this file is built with -ffp-contract=off) -- what the oracle computes per event as n1 / n2
template <typename R>
void fill_ifaces(const lt_ctx* c, std::vector<IfD<R>>& out)
{
    const int n = (int)c->layer_medium.size();
    out.resize((size_t)n + 1);
    for (int k = 0; k <= n; k++) {
        auto index_of = [&](int layer) {     // (an index beyond the media table is refused by lt_launch; this only keeps the read in bounds)
            const int32_t m = c->layer_medium[(size_t)layer];
            return m >= 0 && (size_t)m < c->media.size() ? c->media[(size_t)m].n : 1.0;
        };
        const R up = (R)(k == 0 ? c->n_above : index_of(k - 1));
        const R dn = (R)(k == n ? c->n_below : index_of(k));
        out[(size_t)k].n_up = up; out[(size_t)k].n_dn = dn;
        out[(size_t)k].nr_down = up / dn; out[(size_t)k].nr_up = dn / up;
    }
}

template <typename T>
int upload(lt_ctx* c, DevBuf& b, const std::vector<T>& h)
{
    HIP_TRY(c, b.ensure(h.size() * sizeof(T)));
    if (!h.empty()) HIP_TRY(c, hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return LT_OK;
}

// ---- march grid: which triangles touch which cell (host), exact triangle / box overlap -----------------------------
// Separating-axis test of a triangle against an axis-aligned box (centre bc, half widths bh): the 3 box normals, the
// triangle normal and the 9 cross products of edges and axes.  The box arrives already grown by the margin, so rounding in
// here (1e-16 relative) cannot lose a triangle that really touches the cell.
bool tri_box_overlap(const double bc[3], const double bh[3], const double* tri)
{
    double v[3][3];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) v[i][k] = tri[3 * i + k] - bc[k];
    for (int k = 0; k < 3; k++) {
        const double lo = std::min(v[0][k], std::min(v[1][k], v[2][k])), hi = std::max(v[0][k], std::max(v[1][k], v[2][k]));
        if (lo > bh[k] || hi < -bh[k]) return false;
    }
    const double e[3][3] = {{v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]},
                            {v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2]},
                            {v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2]}};
    for (int j = 0; j < 3; j++)          // axis = unit vector i x edge j
        for (int i = 0; i < 3; i++) {
            const int a = (i + 1) % 3, b = (i + 2) % 3;
            const double ax = -e[j][b], bx = e[j][a];            // the axis has components (ax on a, bx on b)
            const double p0 = ax * v[0][a] + bx * v[0][b], p1 = ax * v[1][a] + bx * v[1][b], p2 = ax * v[2][a] + bx * v[2][b];
            const double r = bh[a] * std::fabs(ax) + bh[b] * std::fabs(bx);
            if (std::min(p0, std::min(p1, p2)) > r || std::max(p0, std::max(p1, p2)) < -r) return false;
        }
    const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2], e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    const double dn = n[0] * v[0][0] + n[1] * v[0][1] + n[2] * v[0][2];
    const double r = bh[0] * std::fabs(n[0]) + bh[1] * std::fabs(n[1]) + bh[2] * std::fabs(n[2]);
    return !(dn > r || dn < -r);
}

// Build the march grid of the ctx mesh: dimensions, candidate lists (host), clearances (device).
int build_march_grid(lt_ctx* c)
{
    c->have_march = false;
    const size_t nt = c->verts.size() / 9;
    double ext[3], longest = 0;
    for (int k = 0; k < 3; k++) { ext[k] = c->nodes[0].hi[k] - c->nodes[0].lo[k]; if (ext[k] > longest) longest = ext[k]; }
    if (!(longest > 0) || !std::isfinite(longest) || nt == 0) return LT_OK;
    // cell size ~ the median triangle's size (sqrt of twice its area): then a cell near the surface lists a handful of
    // triangles and a hop of a few triangle sizes crosses a handful of cells.  32 .. 256 cells along the longest axis;
    // LT_MARCH_CELLS=<n> pins the number, LT_MARCH_SCALE=<x> scales the cell size (tuning).
    std::vector<double> size(nt);
    for (size_t i = 0; i < nt; i++) {
        const double* a = &c->verts[9 * i];
        const double e1[3] = {a[3] - a[0], a[4] - a[1], a[5] - a[2]}, e2[3] = {a[6] - a[0], a[7] - a[1], a[8] - a[2]};
        const double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        size[i] = std::sqrt(std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]));
    }
    std::nth_element(size.begin(), size.begin() + nt / 2, size.end());
    double h = size[nt / 2];
    if (c->knob.march_scale_milli > 10 && c->knob.march_scale_milli < 100000) h *= 1e-3 * (double)c->knob.march_scale_milli;
    int per_axis = h > 0 ? (int)std::lround(longest / h) : 256;
    per_axis = per_axis < 32 ? 32 : (per_axis > 256 ? 256 : per_axis);
    if (c->knob.march_cells >= 4 && c->knob.march_cells <= 512) per_axis = (int)c->knob.march_cells;
    h = longest / (double)per_axis;
    MarchGrid& G = c->mgrid;
    int n[3];
    for (int k = 0; k < 3; k++) {
        n[k] = (int)std::ceil(ext[k] / h); if (n[k] < 1) n[k] = 1;
        G.h[k] = ext[k] > 0 ? ext[k] / n[k] : h;
        // one cell of margin on every side: photons just outside the root bounds -- around an OPEN mesh, next to a flat one
        // (zero extent on an axis: the sheet would otherwise lie ON the grid's wall and half of space outside it) -- are
        // still inside the grid and marched; only origins farther out take the BVH
        n[k] += 2;
        G.org[k] = c->nodes[0].lo[k] - G.h[k]; G.inv[k] = 1.0 / G.h[k];
    }
    G.nx = n[0]; G.ny = n[1]; G.nz = n[2];
    for (int k = 0; k < 3; k++) {
        G.fn[k] = (double)n[k]; G.fn32[k] = (float)n[k];
        G.org32[k] = (float)G.org[k]; G.inv32[k] = (float)G.inv[k]; G.h32[k] = (float)G.h[k];
    }
    const size_t cells = (size_t)n[0] * n[1] * n[2];
    // margin by which a cell is grown before the overlap test: far above the rounding of the f32 walk's positions and hit
    // parameters (1e-6 of the scene) and above the nudge by which the march steps past a cell wall, far below a cell
    const double hmin = std::min(G.h[0], std::min(G.h[1], G.h[2]));
    const double margin = std::max(1e-4 * hmin, 3e-5 * longest);
    G.nudge64 = 1e-9 * hmin; G.nudge32 = 4e-6 * longest;      // (f32 positions carry ~1e-6 of the scene in rounding)
    std::vector<uint32_t> count(cells, 0u);
    std::vector<std::pair<uint32_t, uint32_t>> pairs;      // (cell, triangle), triangles ascending
    pairs.reserve(nt * 8);
    for (size_t i = 0; i < nt; i++) {
        const double* a = &c->verts[9 * i];
        int lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            const double mn = std::min(a[k], std::min(a[3 + k], a[6 + k])) - margin, mx = std::max(a[k], std::max(a[3 + k], a[6 + k])) + margin;
            lo[k] = (int)std::floor((mn - G.org[k]) * G.inv[k]); hi[k] = (int)std::floor((mx - G.org[k]) * G.inv[k]);
            lo[k] = lo[k] < 0 ? 0 : lo[k]; hi[k] = hi[k] >= n[k] ? n[k] - 1 : hi[k];
        }
        const double bh[3] = {0.5 * G.h[0] + margin, 0.5 * G.h[1] + margin, 0.5 * G.h[2] + margin};
        for (int z = lo[2]; z <= hi[2]; z++)
            for (int y = lo[1]; y <= hi[1]; y++)
                for (int x = lo[0]; x <= hi[0]; x++) {
                    const double bc[3] = {G.org[0] + (x + 0.5) * G.h[0], G.org[1] + (y + 0.5) * G.h[1], G.org[2] + (z + 0.5) * G.h[2]};
                    if (!tri_box_overlap(bc, bh, a)) continue;
                    const uint32_t ci = (uint32_t)(((size_t)z * n[1] + y) * n[0] + x);
                    count[ci]++;
                    pairs.emplace_back(ci, (uint32_t)i);
                }
    }
    // record: candidate count in the low bits of x (the device adds c0 above them), start of the candidates in y
    std::vector<uint2> rec(cells);
    size_t total = 0;
    for (size_t i = 0; i < cells; i++) {
        rec[i].x = count[i] < kMarchCountMask ? count[i] : kMarchCountMask;      // 63: a long list -- that query walks the BVH
        rec[i].y = (uint32_t)total;
        total += count[i];
    }
    if (total >= (1u << 26)) return LT_OK;      // queue items address the list with 26 bits: such a mesh keeps the BVH walk (> ~5e6 triangles)
    std::vector<uint32_t> list(total ? total : 1, 0u), fill(cells, 0u);
    for (const auto& pr : pairs) list[rec[pr.first].y + fill[pr.first]++] = pr.second;
    c->march_entries = pairs.size();
    HIP_TRY(c, c->d_mcell.ensure(cells * sizeof(uint2)));
    HIP_TRY(c, c->d_mlist.ensure(list.size() * 4));
    const size_t nc = (size_t)((n[0] + 7) / 8) * ((n[1] + 7) / 8) * ((n[2] + 7) / 8);
    HIP_TRY(c, c->d_mcoarse.ensure(nc * sizeof(double)));
    HIP_TRY(c, hipMemcpyAsync(c->d_mcell.p, rec.data(), cells * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_mlist.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, c->stream));
    G.cell = (const uint2*)c->d_mcell.p; G.list = (const uint32_t*)c->d_mlist.p;
    HIP_TRY(c, launch_march_clearance(c->d_tris[0].p, c->d_nodes[0].p, (int)c->nodes.size(), G, (uint2*)c->d_mcell.p,
                                      (double*)c->d_mcoarse.p, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // rec / list are pageable host memory
    c->have_march = true;
    if (c->on(c->knob.march_info))
        std::fprintf(stderr, "[lt march] %zu triangles, grid %d x %d x %d (cell %.4g), %zu list entries (%.2f per triangle), %zu cells listed\n",
                     nt, n[0], n[1], n[2], h, pairs.size(), (double)pairs.size() / (double)nt,
                     (size_t)std::count_if(count.begin(), count.end(), [](uint32_t v) { return v != 0; }));
    return LT_OK;
}

int upload_tables(lt_ctx* c)
{
    if (!c->tables_dirty && !c->media_dirty) return LT_OK;
    int rc;
    // staging vectors live until the stream has drained (pageable H2D copies)
    std::vector<MedD<double>> m64; std::vector<MedD<float>> m32;
    std::vector<double> z64; std::vector<float> z32;
    std::vector<IfD<double>> i64; std::vector<IfD<float>> i32;
    std::vector<TriD<double>> t64; std::vector<TriD<float>> t32;
    std::vector<NodeD<double>> n64; std::vector<NodeD<float>> n32;
    std::vector<int16_t> links16;
    fill_media(c->media, c->quantity, m64); fill_media(c->media, c->quantity, m32);
    if ((rc = upload(c, c->d_media[0], m64))) return rc;
    if ((rc = upload(c, c->d_media[1], m32))) return rc;
    if (c->have_layers) {      // (layer -> medium indices are checked against the media table by lt_launch before it comes here)
        fill_ifaces(c, i64); fill_ifaces(c, i32);
        if ((rc = upload(c, c->d_if[0], i64))) return rc;
        if ((rc = upload(c, c->d_if[1], i32))) return rc;
    }
    if (!c->tables_dirty) {      // only the media table changed (lt_set_tally_quantity): the geometry tables stay
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->media_dirty = false;
        return LT_OK;
    }
    if (c->have_layers) {
        z64 = c->z_bounds;
        z32.assign(z64.begin(), z64.end());
        if ((rc = upload(c, c->d_zb[0], z64))) return rc;
        if ((rc = upload(c, c->d_zb[1], z32))) return rc;
        if ((rc = upload(c, c->d_lm, c->layer_medium))) return rc;
    }
    if (c->have_mesh) {
        fill_tris(c->verts, c->med_front, c->med_back, t64);
        fill_tris(c->verts, c->med_front, c->med_back, t32);
        fill_nodes(c->nodes, n64); fill_nodes(c->nodes, n32);
        if ((rc = upload(c, c->d_tris[0], t64))) return rc;
        if ((rc = upload(c, c->d_tris[1], t32))) return rc;
        if ((rc = upload(c, c->d_nodes[0], n64))) return rc;
        if ((rc = upload(c, c->d_nodes[1], n32))) return rc;
        c->have_links = bvh_octant_links(c->nodes, links16);
        if (c->have_links && (rc = upload(c, c->d_links, links16))) return rc;
        // Which acceleration data the walks of this mesh use: tables within the LDS budget -> clearance grid with
        // near-triangle lists + BVH in LDS (GEOM 1); beyond it -> march grid (GEOM 2).  The f32 tables are about half the
        // size of the f64 ones, so a mesh may need both.
        Variant vq; std::memset(&vq, 0, sizeof vq); vq.mesh = 1;
        const int nm = (int)c->media.size(), ntr = (int)t64.size(), nno = (int)n64.size();
        const bool lds64 = walk_lds_bytes(vq, nm, 0, ntr, nno) <= kMeshLdsBudget;
        vq.f32 = 1;
        const bool lds32 = walk_lds_bytes(vq, nm, 0, ntr, nno) <= kMeshLdsBudget;
        c->have_march = false;
        if ((!lds64 || c->on(c->knob.force_march)) && !c->on(c->knob.no_clearance) && !c->on(c->knob.no_march)) { if ((rc = build_march_grid(c))) return rc; }
        // clearance grid over the root bounds: 64 cells along the longest axis (LT_NO_CLEARANCE=1 disables it)
        c->have_clear = false;
        // (only for meshes some walk variant stages in LDS: the f64-beyond-LDS kernels use the march grid or the plain BVH,
        // and the builder below is brute force over cells x triangles with 16-bit triangle ids in its records)
        if (!c->on(c->knob.no_clearance) && lds32) {
            if (t64.size() > 0xffff) return c->fail(LT_E_UNSUPPORTED, "lt_set_mesh: %zu triangles within the LDS budget cannot be (near-triangle ids are 16 bits)", t64.size());
            double ext[3], longest = 0;
            for (int k = 0; k < 3; k++) { ext[k] = c->nodes[0].hi[k] - c->nodes[0].lo[k]; if (ext[k] > longest) longest = ext[k]; }
            if (longest > 0 && std::isfinite(longest)) {
                // cells along the longest axis: as fine as a brute-force build (cells x triangles distance evaluations)
                // of ~2e9 evaluations allows, between 32 and 128 (C4's 30 triangles: 128, 8 MiB; a 5000-triangle mesh: 73)
                int per_axis = (int)std::cbrt(2.0e9 / (double)(t64.empty() ? 1 : t64.size()));
                per_axis = per_axis < 32 ? 32 : (per_axis > 128 ? 128 : per_axis);
                // LT_CLEARANCE_CELLS overrides (tuning experiments)
                if (c->knob.clearance_cells >= 8 && c->knob.clearance_cells <= 512) per_axis = (int)c->knob.clearance_cells;
                const double h = longest / (double)per_axis;
                for (int k = 0; k < 3; k++) {
                    c->cn[k] = (int)std::ceil(ext[k] / h); if (c->cn[k] < 1) c->cn[k] = 1;
                    c->ccell[k] = ext[k] > 0 ? ext[k] / c->cn[k] : h; c->corg[k] = c->nodes[0].lo[k];
                }
                const size_t cells = (size_t)c->cn[0] * c->cn[1] * c->cn[2];
                // 16 bytes per cell: the clearance and the cell's nearest triangles (WalkParams::clear; LT_NO_NEAR_LISTS=1
                // leaves the lists empty, every query then walks the BVH)
                HIP_TRY(c, c->d_clear.ensure(cells * 16));
                HIP_TRY(c, launch_build_clearance(c->d_tris[0].p, (int)t64.size(), c->on(c->knob.no_near_lists) ? 0 : 1, c->d_clear.p,
                                                  c->cn[0], c->cn[1], c->cn[2], c->corg, c->ccell, c->stream));
                c->have_clear = true;
            }
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tables_dirty = false; c->media_dirty = false;
    return LT_OK;
}

// Read back the statistics of the last log-mode launch (records logged, records that overflowed to atomics) once its
// streams have drained, and update the deposit-record rate that sizes the next launch's logs and batches.
int collect_log_stats(lt_ctx* c)
{
    if (!c->log_stats_pending) return LT_OK;
    unsigned long long h[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, c->d_job.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // the ctx stream has joined every lane of the launch
    c->log_stats_pending = false;
    c->last_records = h[0]; c->last_overflow = h[1];
    if (c->pending_photons > 0) c->rec_per_photon = (double)(h[0] + h[1]) / (double)c->pending_photons;
    if (c->auto_pending >= 0) {    // overlap auto: remember what this regime cost per photon
        float f = 0;
        if (hipEventElapsedTime(&f, c->ev0, c->ev1) == hipSuccess && c->pending_photons > 0)
        {
            c->auto_ms_per_photon[c->auto_pending] = (double)f / (double)c->pending_photons;
            c->auto_photons[c->auto_pending] = c->pending_photons;
        }
        c->auto_pending = -1;
    }
    return LT_OK;
}

// ---- log-structured tally: planning and enqueueing -------------------------------------------------------------
constexpr uint32_t kMaxLogTiles = 65536;              // tiled record index = tile << 14 | position: 30 bits
constexpr uint64_t kPilotPhotons = 16384;             // pilot batch that measures a new scene's record rate
constexpr uint64_t kOverlapMinPhotons = 1ull << 21;   // overlap auto: launches below this stay on one lane
constexpr uint64_t kTailSplitMinPerWave = 512;        // tail split: photons per launched wave below which a batch walks in one kernel

struct LogGeom { uint32_t ntx, nty, ntz, n_tiles, bits2, nb1; };

LogGeom log_geom(const lt_ctx* c)
{
    LogGeom g;
    g.ntx = ((uint32_t)c->nx + 31u) >> kTileBX; g.nty = ((uint32_t)c->ny + 31u) >> kTileBY; g.ntz = ((uint32_t)c->nz + 15u) >> kTileBZ;
    g.n_tiles = g.ntx * g.nty * g.ntz;   // 32 x 32 x 16-voxel blocks
    // grids of <= 1024 tiles (256^3): ONE partition pass straight to tiles.  Records cluster in the few dozen tiles
    // around the source, so the per-tile runs of a 4096-record work item stay long enough to coalesce.  Larger grids:
    // two passes over digits of <= 7 bits each (<= 1024 level-1 bins).
    uint32_t b = 0;
    if (g.n_tiles > 1024) { b = 1; while ((1u << (2 * b)) < g.n_tiles && b < 7) b++; }
    if (c->knob.log_bits2 >= 0) b = c->knob.log_bits2 > 7 ? 7u : (uint32_t)c->knob.log_bits2;
    while ((g.n_tiles >> b) > 1024 && b < 7) b++;
    g.bits2 = b;
    g.nb1 = b ? (g.n_tiles + (1u << b) - 1) >> b : g.n_tiles;
    return g;
}

// records behind the capacity that padded tile / bin starts and whole-group loads may touch
size_t log_slack_records() { return (size_t)kTileAlign * kMaxLogTiles + (size_t)kBinAlign * 1024 + 2 * (size_t)log_part_item(); }

struct LogPlan {
    int lanes = 1;
    uint32_t cap_chunks[kMaxLanes] = {0, 0, 0};                   // log capacity each lane may use (allocation capped by the budget)
    std::vector<std::pair<int, uint64_t>> batches;     // (lane, photons), in launch order
};

// make lane `ln` hold a log of at least `want` records (not more than `limit`) and its tables
hipError_t ensure_lane(lt_ctx* c, LogLane& ln, size_t want, size_t limit, const LogGeom& G)
{
    hipError_t e;
    const size_t elem = c->grid_elem();
    if (ln.alloc_elem != (int)elem) { ln.release_log(); ln.alloc_elem = (int)elem; }
    if (want > ln.alloc_records) {
        // grow geometrically so that run-to-run jitter of the record rate does not re-allocate tens of GB
        size_t grown = want + want / 8;
        if (grown > limit) grown = limit;
        grown = (grown / kLogChunk) * kLogChunk;
        ln.release_log();
        const size_t n = grown + log_slack_records();
        if ((e = ln.log_idx.ensure(n * 4)) != hipSuccess) return e;
        if ((e = ln.tmp_idx.ensure(n * 4)) != hipSuccess) return e;
        if ((e = ln.log_val.ensure(n * elem)) != hipSuccess) return e;
        if ((e = ln.tmp_val.ensure(n * elem)) != hipSuccess) return e;
        if ((e = ln.log_fill.ensure((grown / kLogChunk) * 4)) != hipSuccess) return e;
        if ((e = ln.itab.ensure((grown / log_part_item() + 1024 + 16) * 16)) != hipSuccess) return e;   // pass-2 item descriptors
        ln.alloc_records = grown;
    }
    // digits of the first partition pass: the tiles (one pass), the level-1 bins (two passes; with hot tiles in front
    // of them up to log_max_digits())
    const size_t nt = G.n_tiles, nd = G.bits2 ? (size_t)log_max_digits() : G.n_tiles;
    if ((e = ln.meta.ensure(LM_WORDS * 4)) != hipSuccess) return e;
    if ((e = ln.head.ensure(sizeof(unsigned long long))) != hipSuccess) return e;
    if ((e = ln.hist.ensure(nt * kLogGroups * 4)) != hipSuccess) return e;
    if ((e = ln.hist1.ensure(nd * kLogGroups * 4)) != hipSuccess) return e;
    if (G.bits2 && (e = ln.hist_unused.ensure(nd * kLogGroups * 4)) != hipSuccess) return e;
    if ((e = ln.bin_base.ensure((nd + 1) * 4)) != hipSuccess) return e;
    if ((e = ln.bin_cnt.ensure((nd + 1) * 4)) != hipSuccess) return e;
    if ((e = ln.tile_base.ensure((nt + 1) * 4)) != hipSuccess) return e;
    if ((e = ln.tile_cnt.ensure((nt + 1) * 4)) != hipSuccess) return e;
    if ((e = ln.cursor1.ensure(nd * kLogGroups * 4)) != hipSuccess) return e;
    if ((e = ln.cursor2.ensure(nt * kLogGroups2 * 4)) != hipSuccess) return e;
    if ((e = ln.items2.ensure((nd + 1) * 4)) != hipSuccess) return e;
    if ((e = ln.items_c.ensure((nd + 1) * 4)) != hipSuccess) return e;
    return ln.items_r.ensure((nt + 1) * 4);
}

// Size the logs for a launch of n photons on `lanes` lanes and cut it into batches.  LT_E_NOMEM: the device cannot
// hold the logs (the caller falls back to fewer lanes or to atomics).
int plan_log(lt_ctx* c, uint64_t n, int lanes, LogPlan* plan)
{
    const LogGeom G = log_geom(c);
    const size_t rec_bytes = 4 + c->grid_elem();
    size_t per_lane = c->log_budget / (2 * rec_bytes) / (size_t)lanes;     // the budget covers log + ping-pong copy of every lane
    if (per_lane > 0xFF000000ull) per_lane = 0xFF000000ull;                // 32-bit record offsets
    per_lane = (per_lane / kLogChunk) * kLogChunk;
    if (per_lane < 16 * (size_t)kLogChunk) {
        if (lanes > 1) return LT_E_NOMEM;      // the budget does not stretch to this many lanes: the caller retries with fewer
        return c->fail(LT_E_INVALID, "lt_launch: log budget too small (%zu bytes)", c->log_budget);
    }
    // lanes 1, 2 tally into grids of their own (run_log_plan): allocated HERE, so that a device without room for them
    // falls back to fewer lanes like a device without room for the logs
    for (int l = 1; l < lanes; l++)
        if (c->d_gridx[l - 1].ensure(c->n_vox() * c->grid_elem()) != hipSuccess) { (void)hipGetLastError(); return LT_E_NOMEM; }
    // no measurement yet (pilot or tiny launch): tissue-like media give 100-400 records per photon
    const double rate = c->rec_per_photon > 0.0 ? c->rec_per_photon : 400.0;
    // Overlapped launches: every lane starts with one large batch (the walks run side by side at full occupancy), then
    // ONE small batch follows whose walk runs beside the first reductions and whose own reduction is all that is left
    // exposed at the end: relative sizes 2 : 2 : 1 on two lanes.  Every further batch costs a drain of its longest
    // photons and a pipeline fill; interleaved in one process (tools/pattern_ab.py, profiles/r02c_batch_layout.log):
    // C2 2,2,1: 40.2 ms, staggered 1,2,2,1: 40.8, five staggered batches: 41.8; 512^3 share 61.5 / 63.1 / 63.0 ms.
    // LT_OVERLAP_PATTERN="w0,w1,..." (relative sizes, dealt to the lanes in turn) overrides the layout for tuning.
    const uint64_t tail = lanes == 1 ? 0 : n / (uint64_t)(2 * lanes + 1);
    const uint64_t b_target = lanes == 1 ? n : (n - tail + (uint64_t)lanes - 1) / (uint64_t)lanes;
    // every resident walk wave holds one partly filled chunk: that many chunks are claimed on top of the records' own
    double waves = 20.0 * (double)c->prop.multiProcessorCount / (double)lanes;          // at most 5 waves per SIMD
    const double launched = 4.0 * (double)((b_target + 255) / 256);                          // 256-thread workgroups
    if (launched < waves) waves = launched;
    const double wave_slack = (double)kLogChunk * waves;
    // (walk workgroup b claims chunks of group b % 16 only: a launch of fewer than 16 workgroups -- under 4096 photons -- reaches
    // 1/16 of the log per workgroup it has, so the log is sized for the groups in use)
    const double groups_used = launched / 4.0 < (double)kLogGroups ? (launched / 4.0 < 1.0 ? 1.0 : launched / 4.0) : (double)kLogGroups;
    const double need = (1.25 * rate * (double)b_target + 1048576.0) * ((double)kLogGroups / groups_used) + wave_slack;
    size_t want = need < (double)per_lane ? (size_t)need : per_lane;
    want = ((want + kLogChunk - 1) / kLogChunk) * kLogChunk;
    size_t cap_use = per_lane;
    for (int l = 0; l < lanes; l++) {
        if (ensure_lane(c, c->lanes[l], want, per_lane, G) != hipSuccess) {
            (void)hipGetLastError();
            for (int k = 0; k < kMaxLanes; k++) c->lanes[k].release_log();
            return LT_E_NOMEM;
        }
        // honour the budget even when an earlier, larger allocation is still there
        const size_t use = c->lanes[l].alloc_records < per_lane ? c->lanes[l].alloc_records : per_lane;
        plan->cap_chunks[l] = (uint32_t)(use / kLogChunk);
        if (use < cap_use) cap_use = use;
    }
    double fit = 0.8 * ((double)cap_use - wave_slack) / rate;
    if (fit < 4096.0) fit = 4096.0;
    plan->lanes = lanes;
    plan->batches.clear();
    uint64_t left = n;
    if (lanes == 1) {
        while (left > 0) { const uint64_t b = (double)left > fit ? (uint64_t)fit : left; plan->batches.emplace_back(0, b); left -= b; }
    } else {
        int l = 0;
        if (const char* e = c->knob.overlap_pattern.empty() ? nullptr : c->knob.overlap_pattern.c_str()) {
            std::vector<double> wts; double sum = 0.0;
            for (const char* q = e; *q;) { char* end = nullptr; const double v_ = std::strtod(q, &end); if (end == q) break; if (v_ > 0) { wts.push_back(v_); sum += v_; } q = *end ? end + 1 : end; }
            for (size_t k = 0; k < wts.size() && left > 0 && sum > 0; k++) {
                uint64_t take = k + 1 == wts.size() ? left : (uint64_t)((double)n * wts[k] / sum);
                if (take < 1) take = 1;
                if (take > left) take = left;
                if ((double)take > fit) take = (uint64_t)fit;
                plan->batches.emplace_back(l, take);
                left -= take; l = (l + 1) % lanes;
            }
        }
        // the body in equal batches (a multiple of the lane count, each within the log), then the tail
        const uint64_t body = left > tail ? left - tail : left;
        const uint64_t cap_b = (double)b_target > fit ? (uint64_t)fit : b_target;
        uint64_t rounds = (body + cap_b * (uint64_t)lanes - 1) / (cap_b * (uint64_t)lanes);
        if (rounds < 1) rounds = 1;
        const uint64_t full = (body + rounds * (uint64_t)lanes - 1) / (rounds * (uint64_t)lanes);
        for (uint64_t k = 0, done_b = 0; k < rounds * (uint64_t)lanes && done_b < body; k++) {
            const uint64_t take = full < body - done_b ? full : body - done_b;
            plan->batches.emplace_back(l, take);
            done_b += take; left -= take; l = (l + 1) % lanes;
        }
        while (left > 0) {
            const uint64_t take = (double)left > fit ? (uint64_t)fit : left;
            plan->batches.emplace_back(l, take);
            left -= take; l = (l + 1) % lanes;
        }
    }
    return LT_OK;
}

// how many lanes this launch uses (lt_set_overlap).  auto: launches that are large enough try two lanes once and one
// lane once (device time per photon, HIP events), then keep the faster regime for this scene -- two streams only
// overlap when the runtime gives them separate hardware queues, which a library cannot guarantee.
int choose_lanes(lt_ctx* c, uint64_t n)
{
    c->auto_pending = -1;
    if (c->overlap_mode == 1) return 1;
    if (c->overlap_mode >= 2) return n >= 8192 ? c->overlap_mode : 1;
    if (c->blocks_per_cu > 0 || n < kOverlapMinPhotons) return 1;    // the caller pinned the launch geometry / small job
    // A regime's cost per photon is only comparable between launches of similar size (a launch's fixed costs -- pipeline fill,
    // the drain of its longest photons -- weigh more on a small one): a measurement taken on a launch more than 2x away from
    // this one's size is measured again.
    for (int k = kAutoLanes - 1; k >= 0; k--) {
        const bool stale = c->auto_photons[k] != 0 && (n > 2 * c->auto_photons[k] || c->auto_photons[k] > 2 * n);
        if (c->auto_ms_per_photon[k] <= 0.0 || stale) { c->auto_pending = k; return k + 1; }
    }
    int best = 0;
    for (int k = 1; k < kAutoLanes; k++) if (c->auto_ms_per_photon[k] < c->auto_ms_per_photon[best]) best = k;
    return best + 1;
}

struct LogRun {
    lt_ctx* c; WalkParams P; Variant v; LaunchCfg cfg; LogGeom G;
};

// Walk train (lt_set_tuning "serial_walks" = 1): the walk kernels of ALL contexts of a device that set the knob run one
// after another -- each waits for the end of the one enqueued before it, whichever context that belongs to -- while every
// context's log reduction stays on its own stream.  A host that keeps several jobs in flight then launches each walk at
// THREE of the four resident workgroups per CU (lt_set_launch_config(3, 256): 92 % of the full walk's speed, measured
// 30.6 against 28.2 ms on config 2) and the fourth slot -- 128 VGPRs per SIMD lane, enough for one partition or one
// tile-reduce workgroup per CU -- carries the reduction of the job in front.  Two walks in flight at two workgroups each
// (round 2-3's regimes) ran at the same total occupancy but left the reductions nothing until a walk had ended.
// The gate is a ring of events that live as long as the process (a context may be destroyed while a later walk still waits).
struct WalkGate {
    std::mutex mu;
    hipEvent_t ring[64] = {};
    unsigned next = 0;
    hipEvent_t last = nullptr;
};
WalkGate& walk_gate(int device)
{
    static WalkGate gates[16];
    return gates[device & 15];
}

hipError_t lane_event(LogLane& ln, hipStream_t s)
{
    if (ln.ev_used == ln.evs.size()) {
        hipEvent_t ev;
        hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) return e;
        ln.evs.push_back(ev);
    }
    return hipEventRecord(ln.evs[ln.ev_used++], s);
}

// Enqueue every batch of `plan` without a host read-back: item counts stay in device memory, the partition / reduce
// kernels are persistent work loops, and the statistics that size the next launch are read lazily at lt_sync.
int run_log_plan(LogRun& R, const LogPlan& plan, uint64_t offset, int* n_batches)
{
    lt_ctx* c = R.c;
    const LogGeom& G = R.G;
    // hot-tile form (two-pass grids with a map, see lt_launch): k_log_count1 counts the digits of pass 1; the walk's
    // level-1 histogram goes to a buffer nobody reads (the walk kernels stay as they are)
    const bool hot = G.bits2 != 0 && c->dmap_valid;
    const uint32_t n_hist = G.bits2 ? G.nb1 : G.n_tiles;
    LaunchCfg cfg = R.cfg;
    cfg.lds_bytes = walk_lds_bytes(R.v, R.P.n_media, R.P.n_layers, R.P.n_tris, R.P.n_nodes, n_hist);
    const int resident = walk_max_blocks_per_cu(R.v, cfg.threads, cfg.lds_bytes);
    if (resident <= 0) return c->fail(LT_E_HIP, "lt_launch: log-mode kernel not resident");
    // two lanes: each walk takes half of the resident workgroups, so that two walks together fill the register file
    // and one walk leaves room for the other lane's partition / reduce workgroups
    int per_cu = c->blocks_per_cu > 0 ? c->blocks_per_cu : (plan.lanes >= 2 ? (resident + 1) / 2 : resident);
    if (plan.lanes >= 2 && c->knob.overlap_walk_bpc >= 1 && c->knob.overlap_walk_bpc <= 8) per_cu = (int)c->knob.overlap_walk_bpc;
    const unsigned long long cap = (unsigned long long)per_cu * (unsigned long long)c->prop.multiProcessorCount;
    if (plan.lanes >= 2) {
        // lanes 1, 2 tally into grids of their own (zeroed here, added to the ctx grid at the join): every lane can then
        // update "its" voxels with plain read-add-writes, and a walk's overflow atomics never race with another lane's
        // reduce
        const size_t gb = c->n_vox() * c->grid_elem();
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
        for (int l = 1; l < plan.lanes; l++) {
            if (!c->lanes[l].stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->lanes[l].stream, hipStreamNonBlocking));
            HIP_TRY(c, hipStreamWaitEvent(c->lanes[l].stream, c->ev_fork, 0));      // (d_gridx: allocated by plan_log)
            HIP_TRY(c, hipMemsetAsync(c->d_gridx[l - 1].p, 0, gb, c->lanes[l].stream));
        }
    }
    for (const auto& bt : plan.batches) {
        LogLane& ln = c->lanes[bt.first];
        hipStream_t s = ln.stream;
        uint32_t* meta = (uint32_t*)ln.meta.p;
        WalkParams P = R.P;
        P.n_photons = bt.second; P.photon_offset = offset;
        if (bt.first >= 1) P.grid = c->d_gridx[bt.first - 1].p;      // (the log's overflow path adds to the lane's grid)
        P.head = (unsigned long long*)ln.head.p;
        P.log_idx = (uint32_t*)ln.log_idx.p; P.log_val = ln.log_val.p; P.log_fill = (uint32_t*)ln.log_fill.p;
        P.log_next = meta + LM_NEXT; P.log_overflow = meta + LM_OVERFLOW; P.log_cap_chunks = plan.cap_chunks[bt.first];
        P.log_hist = (uint32_t*)(hot ? ln.hist_unused.p : (G.bits2 ? ln.hist1.p : ln.hist.p)); P.log_n_hist = n_hist; P.log_hist_shift = kTileShift + G.bits2;
        P.log_ntx = G.ntx; P.log_nty = G.nty;
        const unsigned long long want = (bt.second + (unsigned long long)cfg.threads - 1) / (unsigned long long)cfg.threads;
        cfg.blocks = (int)(want < cap ? want : cap);
        if (cfg.blocks < 1) cfg.blocks = 1;

        LogReduceParams L;
        std::memset(&L, 0, sizeof L);
        L.log_idx = P.log_idx; L.log_val = P.log_val; L.log_fill = P.log_fill;
        L.tmp_idx = (uint32_t*)ln.tmp_idx.p; L.tmp_val = ln.tmp_val.p;
        L.hist1 = (uint32_t*)ln.hist1.p; L.hist = (uint32_t*)ln.hist.p;
        L.bin_base = (uint32_t*)ln.bin_base.p; L.bin_cnt = (uint32_t*)ln.bin_cnt.p;
        L.tile_base = (uint32_t*)ln.tile_base.p; L.tile_cnt = (uint32_t*)ln.tile_cnt.p;
        L.cursor1 = (uint32_t*)ln.cursor1.p; L.cursor2 = (uint32_t*)ln.cursor2.p;
        L.items2 = (uint32_t*)ln.items2.p; L.items_c = (uint32_t*)ln.items_c.p; L.items_r = (uint32_t*)ln.items_r.p;
        L.itab = (uint32_t*)ln.itab.p;
        L.meta = meta; L.job = (unsigned long long*)c->d_job.p; L.cap_chunks = P.log_cap_chunks;
        L.n_tiles = G.n_tiles; L.bits2 = G.bits2;
        L.grid = bt.first >= 1 ? c->d_gridx[bt.first - 1].p : c->d_grid.p; L.n_vox = c->n_vox(); L.tally = c->tally;
        L.nx = (uint32_t)c->nx; L.ny = (uint32_t)c->ny; L.nz = (uint32_t)c->nz; L.ntx = G.ntx; L.nty = G.nty;
        L.flush_atomic = 0;
        if (hot) { L.dmap = (const uint16_t*)c->d_dmap.p; L.dmeta = (const uint32_t*)c->d_dmeta.p; }

        HIP_TRY(c, hipMemsetAsync(ln.head.p, 0, sizeof(unsigned long long), s));
        HIP_TRY(c, hipMemsetAsync(meta, 0, LM_WORDS * 4, s));
        HIP_TRY(c, hipMemsetAsync(ln.hist.p, 0, (size_t)G.n_tiles * (G.bits2 ? kLogGroups2 : kLogGroups) * 4, s));
        HIP_TRY(c, hipMemsetAsync(ln.log_fill.p, 0, (size_t)P.log_cap_chunks * 4, s));   // unclaimed chunk indices read as empty
        if (G.bits2) HIP_TRY(c, hipMemsetAsync(ln.hist1.p, 0, (size_t)(hot ? log_max_digits() : G.nb1) * kLogGroups * 4, s));
        // Tail split (slab walks, XORWOW, batches large enough to have a drain worth hiding): the walk kernel hands the
        // last photons of every wave to a pool and ends; the tail kernel finishes them on the lane's tail stream, with
        // atomic deposits, WHILE the log reduction of this batch runs here.  The reduction then flushes with atomics too.
        // Only where a drain is exposed: the LAST batch of the launch (earlier batches drain beside the next batch's walk),
        // or every batch of a one-lane launch (there the partition waits for each drain).
        // ... and only where the drain is a small part of the batch: the tail's deposits are per-deposit atomics, the slow
        // path the log exists to avoid.  A batch of fewer than kTailSplitMinPerWave photons per launched wave is "all tail"
        // (300 000 photons over 4096 waves: 73 each -- measured: 84 % of the records took the atomic route).
        // lt_set_tuning("tail_split", n >= 2) forces the split with threshold n whatever the size (tests).
        const bool last_batch = &bt == &plan.batches.back();
        const uint64_t waves_launched = (uint64_t)cfg.blocks * (uint64_t)(cfg.threads / 64);
        const bool big_enough = c->knob.tail_split >= 2 || bt.second >= kTailSplitMinPerWave * waves_launched;
        const bool split = R.v.mesh <= 1 && !R.v.table && !R.v.capture && c->knob.tail_split != 0 && big_enough &&
                           ((plan.lanes == 1 && (c->blocks_per_cu == 0 || c->knob.serial_walks > 0) && (!R.v.f32 || c->knob.tail_split >= 2)) ||
                            (plan.lanes > 1 && last_batch));
                           // (one lane, f32 walk: measured a LOSS -- 28.9 -> 31.6 ms on C2: the partition beside the tail kernel takes
                           //  3 ms longer and the f32 drain is short -- so it is left out there.)  (A host that runs walks at
                           // partial occupancy keeps several contexts in flight: their drains are hidden already, a tail kernel only adds contention)
        // alone: one lane AND the caller has not asked for a walk at partial occupancy (lt_set_launch_config: what a host that
        // keeps several contexts in flight does, bench.py's two_jobs / three_jobs): then nothing co-runs with this reduction
        // (a split batch is NOT alone: its tail kernel walks beside the partition and the reduce -- ADVICE r3; the knob
        // part_alone = 0 / 1 pins the build for A/B runs)
        L.alone = (plan.lanes == 1 && c->blocks_per_cu == 0 && !split) ? 1 : 0;
        if (c->knob.part_alone == 0 || c->knob.part_alone == 1) L.alone = (int)c->knob.part_alone;
        L.lds_part = c->knob.part_lds > 0 ? (int)(c->knob.part_lds & 7) : 0;
        Variant vw = R.v;
        if (split) {
            const size_t cap = (size_t)cfg.blocks * (size_t)(cfg.threads / 64) * kDumpPoolLanes;      // every wave hands over at most that many
            P.dump_max = c->knob.tail_split > 1 ? (uint32_t)std::min<long>(c->knob.tail_split, (long)kDumpPoolLanes) : kDumpMaxLanes;
            HIP_TRY(c, ln.pool.ensure(cap * (R.v.f32 ? sizeof(SurvD<float>) : sizeof(SurvD<double>))));
            HIP_TRY(c, ln.pool_n.ensure(4));
            if (!ln.tail_stream) HIP_TRY(c, hipStreamCreateWithFlags(&ln.tail_stream, hipStreamNonBlocking));
            if (!ln.ev_bulk) HIP_TRY(c, hipEventCreateWithFlags(&ln.ev_bulk, hipEventDisableTiming));
            if (!ln.ev_tail) HIP_TRY(c, hipEventCreateWithFlags(&ln.ev_tail, hipEventDisableTiming));
            HIP_TRY(c, hipMemsetAsync(ln.pool_n.p, 0, 4, s));
            P.pool = ln.pool.p; P.pool_n = (uint32_t*)ln.pool_n.p; P.pool_cap = (uint32_t)cap;
            vw.phase = 1;
        }
        const bool train = c->knob.serial_walks > 0;
        if (train) {
            WalkGate& g = walk_gate(c->device);
            std::lock_guard<std::mutex> lk(g.mu);
            if (g.last) HIP_TRY(c, hipStreamWaitEvent(s, g.last, 0));
        }
        HIP_TRY(c, lane_event(ln, s));
        HIP_TRY(c, launch_walk(P, vw, cfg, s));
        HIP_TRY(c, lane_event(ln, s));
        if (split) {
            HIP_TRY(c, hipEventRecord(ln.ev_bulk, s));
            HIP_TRY(c, hipStreamWaitEvent(ln.tail_stream, ln.ev_bulk, 0));
            WalkParams Pt = P;
            Pt.log_idx = nullptr; Pt.log_val = nullptr; Pt.log_n_hist = 0;       // the tail deposits with atomics into the lane's grid
            Variant vt = R.v; vt.phase = 2;
            LaunchCfg ct = cfg;
            ct.lds_bytes = walk_lds_bytes(vt, P.n_media, P.n_layers, P.n_tris, P.n_nodes, 0);
            ct.blocks = (int)((P.pool_cap / 64u + (unsigned)(cfg.threads / 64) - 1) / (unsigned)(cfg.threads / 64));     // 64 pool entries per wave
            if (ct.blocks < 1) ct.blocks = 1;
            HIP_TRY(c, launch_walk(Pt, vt, ct, ln.tail_stream));
            HIP_TRY(c, hipEventRecord(ln.ev_tail, ln.tail_stream));
        }
        if (hot) HIP_TRY(c, launch_log_count1(L, s));
        HIP_TRY(c, G.bits2 ? launch_log_scan_bins(L, s) : launch_log_scan_tiles(L, s));
        if (train) {      // the next walk of the train may start: behind this batch's walk AND its scan -- a 1024-lane workgroup that finds
                          // no CU to run on once the next walk has filled them all (seen: a scan that waited 30 ms for a walk to end)
            WalkGate& g = walk_gate(c->device);
            std::lock_guard<std::mutex> lk(g.mu);
            hipEvent_t& ev = g.ring[g.next++ & 63u];
            if (!ev) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(ev, s));
            g.last = ev;
        }
        HIP_TRY(c, lane_event(ln, s));
        HIP_TRY(c, launch_log_part1(L, s));
        LogReduceParams Lr = L;
        if (G.bits2 == 0) { Lr.log_idx = L.tmp_idx; Lr.log_val = L.tmp_val; }   // one pass: tiles are final in tmp
        else HIP_TRY(c, launch_log_part2(L, s));                                  // tile counts, their scan, pass 2
        HIP_TRY(c, lane_event(ln, s));
        if (split) Lr.flush_atomic = 1;        // the tail kernel may be adding to the same voxels
        HIP_TRY(c, launch_log_reduce(Lr, s));
        HIP_TRY(c, lane_event(ln, s));
        if (split) HIP_TRY(c, hipStreamWaitEvent(s, ln.ev_tail, 0));      // the batch is complete when its tail is (and the pool may be reused)
        if (bt.first == 0 && G.bits2) c->tile_cnt_ready = true;      // lane 0's tile_cnt: this scene's records per tile
        offset += bt.second;
        (*n_batches)++;
    }
    for (int l = 1; l < plan.lanes; l++) {
        HIP_TRY(c, hipEventRecord(c->lanes[l].ev_done, c->lanes[l].stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->lanes[l].ev_done, 0));
        HIP_TRY(c, launch_grid_add(c->d_grid.p, c->d_gridx[l - 1].p, c->tally, c->n_vox(), c->stream));
    }
    return LT_OK;
}

// structural check of a flattened pre-order tree: every node reached exactly once, leaves within the triangle range,
// interior nodes with a valid axis and a second child behind the first subtree.  Iterative: the traversal on the device is
// stackless (skip links), so there is no depth a tree could exceed.
bool bvh_is_valid(const std::vector<lt_bvh_node>& n, int n_tris, int* max_depth)
{
    std::vector<std::pair<int, int>> todo{{0, 0}};     // (node, depth)
    std::vector<char> seen(n.size(), 0);
    size_t visited = 0;
    *max_depth = 0;
    while (!todo.empty()) {
        const auto [idx, depth] = todo.back(); todo.pop_back();
        if (idx < 0 || idx >= (int)n.size() || seen[idx]) return false;
        seen[idx] = 1; visited++;
        if (depth > *max_depth) *max_depth = depth;
        const lt_bvh_node& nd = n[idx];
        if (nd.n_prims > 0) { if (nd.offset < 0 || nd.offset + nd.n_prims > n_tris) return false; continue; }
        if (nd.axis < 0 || nd.axis > 2 || nd.offset <= idx + 1) return false;
        todo.emplace_back(nd.offset, depth + 1); todo.emplace_back(idx + 1, depth + 1);
    }
    return visited == n.size();
}

}  // namespace

extern "C" {

int lt_abi_version(void) { return LT_ABI_VERSION; }

const char* lt_last_error(const lt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int lt_create(lt_ctx** out, int device_id)
{
    if (!out) return LT_E_INVALID;
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "0 devices") +
                         "); this library has no CPU fallback";
        return LT_E_HIP;
    }
    if (device_id < 0 || device_id >= n_dev) { g_create_error = "device_id out of range"; return LT_E_INVALID; }
    lt_ctx* c = new (std::nothrow) lt_ctx();
    if (!c) { g_create_error = "out of host memory"; return LT_E_NOMEM; }
    c->device = device_id;
    {   // the one place the environment is read: LT_QUERY_MIN, LT_LOG_HOT, ... seed the knobs of lt_set_tuning
        static const char* const names[] = {"query_min", "log_bits2", "log_hot", "overlap_walk_bpc", "diag_no_tally", "log_timing", "march_cells",
                                            "march_scale_milli", "no_march", "no_clearance", "no_near_lists", "clearance_cells", "march_info", "force_march", "tail_split", "part_alone", "serial_walks", "part_lds"};
        for (const char* k : names) {
            std::string name = "LT_";
            for (const char* q = k; *q; q++) name += (char)std::toupper((unsigned char)*q);
            if (const char* v = std::getenv(name.c_str())) *c->knob_by_name(k) = std::atol(v);
        }
        if (const char* v = std::getenv("LT_OVERLAP_PATTERN")) c->knob.overlap_pattern = v;
    }
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device_id);
    if (e == hipSuccess && std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device arch ") + c->prop.gcnArchName + " is not gfx950: kernels are built for MI355X only";
        delete c;
        return LT_E_UNSUPPORTED;
    }
    if (e == hipSuccess) {
        size_t quarter = c->prop.totalGlobalMem / 4;
        c->log_budget = c->default_log_budget = quarter < ((size_t)64 << 30) ? quarter : ((size_t)64 << 30);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    c->lanes[0].stream = c->stream;      // (lane 1's stream is created when a launch first overlaps: a process's streams
                                         //  share GPU_MAX_HW_QUEUES hardware queues, idle ones included)
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    for (int k = 0; k < kMaxLanes && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->lanes[k].ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = c->d_job.ensure(2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = c->d_counters.ensure(sizeof(DevCounters));
    if (e == hipSuccess) e = c->d_head.ensure(sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_counters.p, 0, sizeof(DevCounters), c->stream);
    if (e != hipSuccess) {
        g_create_error = std::string("lt_create: ") + hipGetErrorString(e);
        delete c;
        return LT_E_HIP;
    }
    *out = c;
    return LT_OK;
}

int lt_destroy(lt_ctx* c)
{
    if (!c) return LT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 2; i++) { c->d_media[i].release(); c->d_zb[i].release(); c->d_if[i].release(); c->d_tris[i].release(); c->d_nodes[i].release(); }
    c->d_lm.release(); c->d_grid.release(); c->d_counters.release(); c->d_head.release(); c->d_table.release();
    c->d_scratch_in.release(); c->d_scratch_out.release(); c->d_scratch_aux.release();
    c->d_mats.release(); c->d_lights.release(); c->d_r0.release(); c->d_r1.release(); c->d_lc.release();
    c->d_img.release(); c->d_xy.release(); c->d_vtx.release(); c->d_vcnt.release(); c->d_clear.release();
    c->d_job.release(); c->d_dmap.release(); c->d_dmeta.release();
    c->d_mcell.release(); c->d_mlist.release(); c->d_mcoarse.release(); c->d_links.release();
    for (int k = 1; k < kMaxLanes; k++) { c->d_gridx[k - 1].release(); if (c->lanes[k].stream) (void)hipStreamSynchronize(c->lanes[k].stream); }
    for (int k = 0; k < kMaxLanes; k++) {
        c->lanes[k].release_all();
        for (hipEvent_t ev : c->lanes[k].evs) (void)hipEventDestroy(ev);
        if (c->lanes[k].ev_done) (void)hipEventDestroy(c->lanes[k].ev_done);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int k = 0; k < kMaxLanes; k++) {
        if (c->lanes[k].tail_stream) { (void)hipStreamSynchronize(c->lanes[k].tail_stream); (void)hipStreamDestroy(c->lanes[k].tail_stream); }
        if (c->lanes[k].ev_bulk) (void)hipEventDestroy(c->lanes[k].ev_bulk);
        if (c->lanes[k].ev_tail) (void)hipEventDestroy(c->lanes[k].ev_tail);
    }
    for (int k = 1; k < kMaxLanes; k++) if (c->lanes[k].stream) (void)hipStreamDestroy(c->lanes[k].stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return LT_OK;
}

int lt_set_media(lt_ctx* c, const lt_medium* media, int n)
{
    CHECK_CTX(c);
    if (!media || n <= 0 || n > kMaxMedia) return c->fail(LT_E_INVALID, "lt_set_media: need 1..%d media", kMaxMedia);
    for (int i = 0; i < n; i++) {
        const lt_medium& m = media[i];
        if (!(m.mu_a >= 0) || !(m.mu_s >= 0) || !(m.g > -1 && m.g < 1) || !(m.n > 0) || !std::isfinite(m.mu_a + m.mu_s))
            return c->fail(LT_E_INVALID, "lt_set_media: medium %d out of range (mu_a,mu_s >= 0, |g| < 1, n > 0)", i);
    }
    c->media.assign(media, media + n);
    c->scene_changed();
    c->tables_dirty = true;
    return LT_OK;
}

int lt_set_layers(lt_ctx* c, const double* z_bounds, const int32_t* medium_idx, int n, double n_above, double n_below)
{
    CHECK_CTX(c);
    if (!z_bounds || !medium_idx || n <= 0 || n > kMaxLayers)
        return c->fail(LT_E_INVALID, "lt_set_layers: need 1..%d layers", kMaxLayers);
    if (!std::isfinite(z_bounds[0])) return c->fail(LT_E_INVALID, "lt_set_layers: z_bounds[0] must be finite");
    for (int i = 0; i < n; i++) {
        if (!(z_bounds[i + 1] > z_bounds[i])) return c->fail(LT_E_INVALID, "lt_set_layers: z_bounds must ascend");
        if (medium_idx[i] < 0) return c->fail(LT_E_INVALID, "lt_set_layers: negative medium index");
    }
    if (!(n_above > 0) || !(n_below > 0)) return c->fail(LT_E_INVALID, "lt_set_layers: ambient indices must be > 0");
    c->z_bounds.assign(z_bounds, z_bounds + n + 1);
    c->layer_medium.assign(medium_idx, medium_idx + n);
    c->n_above = n_above; c->n_below = n_below;
    c->have_layers = true; c->have_mesh = false;
    c->scene_changed();
    c->tables_dirty = true;
    return LT_OK;
}

int lt_set_mesh(lt_ctx* c, const double* verts, const int32_t* med_front, const int32_t* med_back, int n_tris,
                const lt_bvh_node* nodes, int n_nodes)
{
    CHECK_CTX(c);
    if (!verts || !med_front || !med_back || n_tris <= 0 || !nodes || n_nodes <= 0)
        return c->fail(LT_E_INVALID, "lt_set_mesh: empty mesh or BVH");
    for (size_t i = 0; i < (size_t)n_tris * 9; i++)
        if (!std::isfinite(verts[i])) return c->fail(LT_E_INVALID, "lt_set_mesh: non-finite vertex");
    std::vector<lt_bvh_node> nn(nodes, nodes + n_nodes);
    int depth = 0;
    if (!bvh_is_valid(nn, n_tris, &depth))
        return c->fail(LT_E_INVALID, "lt_set_mesh: BVH is not a valid pre-order tree over %d triangles", n_tris);
    size_t covered = 0;
    for (const auto& nd : nn) if (nd.n_prims > 0) covered += (size_t)nd.n_prims;
    if (covered != (size_t)n_tris) return c->fail(LT_E_INVALID, "lt_set_mesh: leaves cover %zu of %d triangles", covered, n_tris);
    c->verts.assign(verts, verts + (size_t)n_tris * 9);
    c->med_front.assign(med_front, med_front + n_tris);
    c->med_back.assign(med_back, med_back + n_tris);
    c->nodes.swap(nn);
    c->have_mesh = true; c->have_layers = false;
    c->surf_mats.clear();
    c->scene_changed();
    c->tables_dirty = true;
    return LT_OK;
}

int lt_set_grid(lt_ctx* c, int nx, int ny, int nz, const double origin[3], const double voxel[3], int tally_dtype)
{
    CHECK_CTX(c);
    if (nx <= 0 || ny <= 0 || nz <= 0 || !origin || !voxel) return c->fail(LT_E_INVALID, "lt_set_grid: bad shape");
    if (tally_dtype < LT_TALLY_F32 || tally_dtype > LT_TALLY_U64FX) return c->fail(LT_E_INVALID, "lt_set_grid: bad tally dtype");
    for (int k = 0; k < 3; k++)
        if (!(voxel[k] > 0) || !std::isfinite(origin[k])) return c->fail(LT_E_INVALID, "lt_set_grid: bad origin/voxel");
    if ((double)nx * ny * nz > 2147483647.0) return c->fail(LT_E_UNSUPPORTED, "lt_set_grid: more than 2^31-1 voxels");
    BIND(c);
    c->nx = nx; c->ny = ny; c->nz = nz; c->tally = tally_dtype;
    for (int k = 0; k < 3; k++) { c->origin[k] = origin[k]; c->voxel[k] = voxel[k]; }
    HIP_TRY(c, c->d_grid.ensure(c->n_vox() * c->grid_elem()));
    c->have_grid = true;
    c->scene_changed();
    return lt_zero_tally(c);
}

int lt_set_source(lt_ctx* c, int type, const double pos[3], const double dir[3], const double* extra, int start_medium)
{
    CHECK_CTX(c);
    if (!pos || !dir || (type != LT_SRC_PENCIL && type != LT_SRC_COSINE_QUAD)) return c->fail(LT_E_INVALID, "lt_set_source: bad arguments");
    const double l = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    if (!(l > 0) || !std::isfinite(l)) return c->fail(LT_E_INVALID, "lt_set_source: zero direction");
    if (type == LT_SRC_COSINE_QUAD && !extra) return c->fail(LT_E_INVALID, "lt_set_source: quad source needs extra[6]");
    c->src_type = type; c->start_medium = start_medium;
    for (int k = 0; k < 3; k++) { c->src_pos[k] = pos[k]; c->src_dir[k] = dir[k] / l; }
    for (int k = 0; k < 6; k++) c->src_extra[k] = extra ? extra[k] : 0.0;
    c->have_source = true;
    c->scene_changed();
    return LT_OK;
}

int lt_set_max_steps(lt_ctx* c, uint32_t max_steps)
{
    CHECK_CTX(c);
    if (max_steps == 0) return c->fail(LT_E_INVALID, "lt_set_max_steps: must be > 0");
    c->max_steps = max_steps;
    return LT_OK;
}

int lt_set_tuning(lt_ctx* c, const char* key, int64_t value)
{
    CHECK_CTX(c);
    long* k = key ? c->knob_by_name(key) : nullptr;
    if (!k) return c->fail(LT_E_INVALID, "lt_set_tuning: unknown key '%s'", key ? key : "(null)");
    *k = value < 0 ? -1 : (long)value;
    // knobs that shape the hot-tile map or the partition geometry: the map of the current scene is made again
    if (!std::strcmp(key, "log_hot") || !std::strcmp(key, "log_bits2")) { c->dmap_valid = false; }
    return LT_OK;
}

int lt_set_tally_quantity(lt_ctx* c, int quantity)
{
    CHECK_CTX(c);
    if (quantity != LT_QUANTITY_ABSORBED && quantity != LT_QUANTITY_FLUENCE)
        return c->fail(LT_E_INVALID, "lt_set_tally_quantity: LT_QUANTITY_ABSORBED or LT_QUANTITY_FLUENCE");
    if (quantity != c->quantity) { c->quantity = quantity; c->media_dirty = true; }     // the media table carries the factor
    return LT_OK;
}

int lt_set_launch_config(lt_ctx* c, int blocks_per_cu, int threads_per_block)
{
    CHECK_CTX(c);
    if (blocks_per_cu < 0 || threads_per_block < 0 || threads_per_block > 256 || (threads_per_block % 64) != 0)
        return c->fail(LT_E_INVALID, "lt_set_launch_config: threads must be a multiple of 64, <= 256");
    c->blocks_per_cu = blocks_per_cu; c->threads_per_block = threads_per_block;
    return LT_OK;
}

int lt_zero_tally(lt_ctx* c)
{
    CHECK_CTX(c);
    BIND(c);
    if (c->have_grid) HIP_TRY(c, hipMemsetAsync(c->d_grid.p, 0, c->n_vox() * c->grid_elem(), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_counters.p, 0, sizeof(DevCounters), c->stream));
    return LT_OK;
}

int lt_launch(lt_ctx* c, uint64_t n_photons, uint64_t photon_offset, uint64_t seed, const double* rng_table,
              uint64_t table_steps, uint32_t flags)
{
    CHECK_CTX(c);
    if (c->media.empty()) return c->fail(LT_E_STATE, "lt_launch: lt_set_media first");
    if (!c->have_layers && !c->have_mesh) return c->fail(LT_E_STATE, "lt_launch: lt_set_layers or lt_set_mesh first");
    if (!c->have_grid) return c->fail(LT_E_STATE, "lt_launch: lt_set_grid first");
    if (!c->have_source) return c->fail(LT_E_STATE, "lt_launch: lt_set_source first");
    const int n_media = (int)c->media.size();
    if (c->have_layers) {
        for (int32_t m : c->layer_medium) if (m >= n_media) return c->fail(LT_E_INVALID, "lt_launch: layer medium index %d >= %d media", m, n_media);
    } else {
        for (size_t i = 0; i < c->med_front.size(); i++)
            if (c->med_front[i] >= n_media || c->med_back[i] >= n_media || c->med_front[i] < -1 || c->med_back[i] < -1)
                return c->fail(LT_E_INVALID, "lt_launch: triangle %zu medium index out of range", i);
        if (c->start_medium < 0 || c->start_medium >= n_media) return c->fail(LT_E_INVALID, "lt_launch: start_medium out of range");
    }
    Variant v;
    v.f32 = (flags & LT_FLAG_F32_WALK) ? 1 : 0;
    v.mesh = c->have_mesh ? 1 : 0;
    v.table = rng_table ? 1 : 0;
    v.tally = c->tally;
    v.capture = c->max_vertices > 0 ? 1 : 0;
    v.phase = 0;
    if (v.capture && (v.f32 || v.table)) return c->fail(LT_E_UNSUPPORTED, "lt_launch: vertex capture runs the f64 walk with the XORWOW generator");
    if (c->on(c->knob.diag_no_tally)) v.tally = 3;  // diagnostic: time the walk without deposition
    if (v.table && v.f32) return c->fail(LT_E_UNSUPPORTED, "lt_launch: table RNG runs the f64 walk only");
    if (v.table && v.tally == LT_TALLY_F32) return c->fail(LT_E_UNSUPPORTED, "lt_launch: table RNG needs an f64 or u64fx tally");
    if (v.table && table_steps == 0) return c->fail(LT_E_INVALID, "lt_launch: table_steps == 0");
    BIND(c);
    int rc = upload_tables(c);
    if (rc) return rc;
    if (n_photons == 0) { c->timed = false; return LT_OK; }

    if (v.table) {
        const size_t bytes = (size_t)n_photons * (size_t)table_steps * 4 * sizeof(double);
        HIP_TRY(c, c->d_table.ensure(bytes));
        HIP_TRY(c, hipMemcpyAsync(c->d_table.p, rng_table, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller may free rng_table on return
    }

    WalkParams P;
    std::memset(&P, 0, sizeof P);
    P.head = (unsigned long long*)c->d_head.p;
    P.n_photons = n_photons; P.photon_offset = photon_offset; P.seed = seed;
    const int pi = v.f32 ? 1 : 0;
    P.media = c->d_media[pi].p; P.zb = c->d_zb[pi].p; P.ifaces = c->d_if[pi].p; P.layer_medium = (const int32_t*)c->d_lm.p;
    P.tris = c->d_tris[pi].p; P.nodes = c->d_nodes[pi].p; P.links = c->have_mesh && c->have_links ? (const int16_t*)c->d_links.p : nullptr;
    P.n_media = n_media;
    P.n_layers = c->have_layers ? (int)c->layer_medium.size() : 0;
    P.n_tris = c->have_mesh ? (int)c->med_front.size() : 0;
    P.n_nodes = c->have_mesh ? (int)c->nodes.size() : 0;
    P.n_above = c->n_above; P.n_below = c->n_below;
    P.grid = c->d_grid.p; P.nx = c->nx; P.ny = c->ny; P.nz = c->nz; P.tally = c->tally;
    for (int k = 0; k < 3; k++) {
        P.origin[k] = c->origin[k]; P.inv_voxel[k] = 1.0 / c->voxel[k];
        P.src_pos[k] = c->src_pos[k]; P.src_dir[k] = c->src_dir[k];
        P.src_e1[k] = c->src_extra[k]; P.src_e2[k] = c->src_extra[3 + k];
    }
    P.fdim[0] = (double)c->nx; P.fdim[1] = (double)c->ny; P.fdim[2] = (double)c->nz;
    for (int k = 0; k < 3; k++) { P.f32.origin[k] = (float)P.origin[k]; P.f32.inv_voxel[k] = (float)P.inv_voxel[k]; P.f32.fdim[k] = (float)P.fdim[k]; }
    P.src_type = c->src_type; P.start_medium = c->start_medium;
    P.table = (const double*)c->d_table.p; P.table_steps = table_steps;
    P.max_steps = c->max_steps;
    P.counters = (DevCounters*)c->d_counters.p;
    if (c->knob.query_min >= 1 && c->knob.query_min <= 64) P.query_min = (unsigned)c->knob.query_min;
    if (c->have_mesh && c->have_clear) {
        P.clear = (const uint4*)c->d_clear.p; P.cnx = c->cn[0]; P.cny = c->cn[1]; P.cnz = c->cn[2];
        for (int k = 0; k < 3; k++) {
            P.corg[k] = c->corg[k]; P.cinv[k] = 1.0 / c->ccell[k]; P.cdim[k] = (double)c->cn[k];
            P.f32.corg[k] = (float)P.corg[k]; P.f32.cinv[k] = (float)P.cinv[k]; P.f32.cdim[k] = (float)P.cdim[k];
        }
    }
    if (c->have_mesh && c->have_march) P.mg = c->mgrid;
    c->captured_photons = 0;
    if (c->max_vertices > 0) {
        const size_t vb = (size_t)n_photons * c->max_vertices * sizeof(lt_vertex);
        if (vb > ((size_t)64 << 30)) return c->fail(LT_E_NOMEM, "lt_launch: vertex capture needs %zu bytes (> 64 GiB)", vb);
        HIP_TRY(c, c->d_vtx.ensure(vb));
        HIP_TRY(c, c->d_vcnt.ensure((size_t)n_photons * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->d_vcnt.p, 0, (size_t)n_photons * sizeof(uint32_t), c->stream));
        P.vertices = (lt_vertex*)c->d_vtx.p; P.vertex_counts = (uint32_t*)c->d_vcnt.p; P.max_vertices = c->max_vertices;
        c->captured_photons = n_photons; c->captured_max_vertices = c->max_vertices;
    }

    LaunchCfg cfg;
    cfg.threads = c->threads_per_block > 0 ? c->threads_per_block : 256;
    cfg.lds_bytes = walk_lds_bytes(v, P.n_media, P.n_layers, P.n_tris, P.n_nodes);
    if (v.mesh && (cfg.lds_bytes > kMeshLdsBudget || (c->on(c->knob.force_march) && c->have_march && !v.table))) {
        // large mesh: leave triangles and nodes in global memory (L2 / Infinity Cache resident)
        if (v.table) return c->fail(LT_E_UNSUPPORTED, "lt_launch: table RNG with a mesh beyond the LDS budget");
        v.mesh = c->have_march ? 3 : 2;      // with a march grid: walk_kernel_m
        cfg.lds_bytes = walk_lds_bytes(v, P.n_media, P.n_layers, P.n_tris, P.n_nodes);
    }
    int resident = walk_max_blocks_per_cu(v, cfg.threads, cfg.lds_bytes);
    if (resident <= 0) return c->fail(LT_E_HIP, "lt_launch: kernel variant not resident (occupancy query returned %d)", resident);
    int per_cu = c->blocks_per_cu > 0 ? c->blocks_per_cu : resident;
    // persistent threads: no more workgroups than photons can feed
    unsigned long long want = (n_photons + (unsigned long long)cfg.threads - 1) / (unsigned long long)cfg.threads;
    unsigned long long cap = (unsigned long long)per_cu * (unsigned long long)c->prop.multiProcessorCount;
    cfg.blocks = (int)(want < cap ? want : cap);
    if (cfg.blocks < 1) cfg.blocks = 1;

    // ---- log-structured tally: walk -> deposit log -> partition by grid tile -> LDS tile reduce, in batches
    // auto: every walk is paced by the atomic unit (~17-20e9 requests/s) when it deposits with atomics -> log.  (Until
    // the near-triangle lists the f64 mesh walk was the exception: its BVH arithmetic hid the atomics, 58 vs 59 ms on
    // C4; with them: 53.8 ms atomic, 44.3 ms log.)
    const int mode = c->tally_mode == 2 ? 1 : c->tally_mode;
    { int rc3 = collect_log_stats(c); if (rc3) return rc3; }   // stats of the previous launch size this one
    LogGeom G = log_geom(c);
    bool use_log = mode == 1 && !v.table && c->max_vertices == 0 && G.n_tiles <= kMaxLogTiles && !c->on(c->knob.diag_no_tally);
    if (use_log) {
        LogRun R;
        R.c = c; R.P = P; R.v = v; R.cfg = cfg; R.G = G;
        if (c->dmap_valid && (c->dmap_tiles != G.n_tiles || c->dmap_bits2 != G.bits2)) c->dmap_valid = c->tile_cnt_ready = false;   // (LT_LOG_BITS2 changed)
        c->stages_valid = false;
        for (LogLane& ln : c->lanes) ln.ev_used = 0;
        HIP_TRY(c, hipMemsetAsync(c->d_job.p, 0, 2 * sizeof(unsigned long long), c->stream));
        uint64_t done = 0;
        int n_batches = 0;
        // No record rate known for this scene yet: trace a small pilot batch first and read its rate back (a few ms,
        // once per scene), so that the logs and batches of the rest are sized from a measurement, not from a guess.
        if (c->rec_per_photon <= 0.0 && n_photons > kPilotPhotons) {
            LogPlan pilot;
            rc = plan_log(c, kPilotPhotons, 1, &pilot);
            if (rc == LT_OK) rc = run_log_plan(R, pilot, photon_offset, &n_batches);
            if (rc == LT_E_NOMEM) { use_log = false; rc = LT_OK; }
            else if (rc) return rc;
            else {
                c->log_stats_pending = true; c->pending_photons = kPilotPhotons; c->auto_pending = -1;
                if ((rc = collect_log_stats(c))) return rc;
                done = kPilotPhotons;
            }
        }
        if (use_log && G.bits2 && !c->dmap_valid && c->tile_cnt_ready && G.n_tiles <= kMaxHotTiles && G.nb1 < log_max_digits()) {
            // two-pass grid with a measured tile histogram (the pilot's, or an earlier small launch's): make the
            // hot-tile map for this scene.  LT_LOG_HOT caps the number of hot tiles (0: plain two-pass form).
            uint32_t max_hot = log_max_digits() - G.nb1;
            if (c->knob.log_hot >= 0 && (uint32_t)c->knob.log_hot < max_hot) max_hot = (uint32_t)c->knob.log_hot;
            if (max_hot > 0) {
                HIP_TRY(c, c->d_dmap.ensure(((size_t)G.n_tiles + 2) * sizeof(uint16_t)));
                HIP_TRY(c, c->d_dmeta.ensure(4 * sizeof(uint32_t)));
                HIP_TRY(c, launch_log_plan((const uint32_t*)c->lanes[0].tile_cnt.p, G.n_tiles, G.bits2, max_hot, (uint16_t*)c->d_dmap.p,
                                           (uint32_t*)c->d_dmeta.p, c->stream));
                c->dmap_valid = true; c->dmap_tiles = G.n_tiles; c->dmap_bits2 = G.bits2;
            }
        }
        if (use_log) {
            int lanes = choose_lanes(c, n_photons - done);
            LogPlan plan;
            rc = plan_log(c, n_photons - done, lanes, &plan);
            while (rc == LT_E_NOMEM && lanes > 1) {
                lanes--; rc = plan_log(c, n_photons - done, lanes, &plan);
                if (c->auto_pending >= 0) c->auto_pending = lanes - 1;      // what this launch measures is the regime it really runs in
            }
            if (rc == LT_E_NOMEM) { use_log = false; rc = LT_OK; }
            else if (rc) return rc;
            else {
                HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
                if ((rc = run_log_plan(R, plan, photon_offset + done, &n_batches))) return rc;
                HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
                c->stages_valid = true; c->last_batches = (uint64_t)n_batches; c->last_lanes = plan.lanes;
                c->last_hot = G.bits2 != 0 && c->dmap_valid;
                c->log_stats_pending = true; c->pending_photons = n_photons;   // the job counters hold the pilot's records too
                if (c->auto_pending >= 0 && done > 0) c->auto_pending = -1;     // (first launch of a scene: not a clean measurement)
                c->timed = true;
                if (c->on(c->knob.log_timing)) {   // diagnostic: per-stage device times of this launch (synchronises)
                    double ms[4]; uint64_t rec = 0, bat = 0;
                    if (lt_last_log_stages(c, ms, &rec, &bat) == LT_OK)
                        std::fprintf(stderr, "[lt log] %llu photons, %d lane(s), %llu batches, %llu records (%.1f / photon), %llu to atomics; "
                                     "stage sums ms: walk %.2f scan %.2f partition %.2f reduce %.2f\n", (unsigned long long)n_photons,
                                     plan.lanes, (unsigned long long)bat, (unsigned long long)rec, (double)rec / (double)n_photons,
                                     (unsigned long long)c->last_overflow, ms[0], ms[1], ms[2], ms[3]);
                }
                return LT_OK;
            }
        }
        // not enough free HBM for a log (other contexts, a huge grid): the rest of this launch deposits with atomics
        P.n_photons = n_photons - done; P.photon_offset = photon_offset + done;
        want = (P.n_photons + (unsigned long long)cfg.threads - 1) / (unsigned long long)cfg.threads;
        cfg.blocks = (int)(want < cap ? want : cap);
        if (cfg.blocks < 1) cfg.blocks = 1;
    }

    c->stages_valid = false;
    HIP_TRY(c, hipMemsetAsync(c->d_head.p, 0, sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    HIP_TRY(c, launch_walk(P, v, cfg, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return LT_OK;
}

int lt_reserve_log(lt_ctx* c, uint64_t n_photons)
{
    CHECK_CTX(c);
    if (!c->have_grid) return c->fail(LT_E_STATE, "lt_reserve_log: lt_set_grid first");
    BIND(c);
    LogPlan plan;
    int lanes = c->overlap_mode == 1 ? 1 : (c->overlap_mode >= 2 ? c->overlap_mode : ((c->blocks_per_cu == 0 && n_photons >= kOverlapMinPhotons) ? kAutoLanes : 1));
    int rc = plan_log(c, n_photons, lanes, &plan);
    if (rc == LT_E_NOMEM) return c->fail(LT_E_NOMEM, "lt_reserve_log: not enough device memory for the deposit log");
    return rc;
}

int lt_set_tally_mode(lt_ctx* c, int mode, uint64_t log_bytes)
{
    CHECK_CTX(c);
    if (mode < 0 || mode > 2) return c->fail(LT_E_INVALID, "lt_set_tally_mode: mode must be LT_MODE_ATOMIC, LT_MODE_LOG or LT_MODE_AUTO");
    c->tally_mode = mode;
    c->log_budget = log_bytes ? (size_t)log_bytes : c->default_log_budget;
    return LT_OK;
}

int lt_last_log_hot_tiles(lt_ctx* c, uint32_t* hot_tiles, uint32_t* threshold)
{
    CHECK_CTX(c);
    if (!c->stages_valid) return c->fail(LT_E_STATE, "lt_last_log_hot_tiles: the last launch did not use the deposit log");
    BIND(c);
    uint32_t m[2] = {0, 0};
    if (c->last_hot) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(m, c->d_dmeta.p, sizeof m, hipMemcpyDeviceToHost));
    }
    if (hot_tiles) *hot_tiles = m[0];
    if (threshold) *threshold = m[1];
    return LT_OK;
}

int lt_set_overlap(lt_ctx* c, int lanes)
{
    CHECK_CTX(c);
    if (lanes < 0 || lanes > kMaxLanes) return c->fail(LT_E_INVALID, "lt_set_overlap: 0 (auto), 1, 2 or 3");
    c->overlap_mode = lanes;
    for (double& a : c->auto_ms_per_photon) a = 0.0;
    for (uint64_t& a : c->auto_photons) a = 0;
    c->auto_pending = -1;
    return LT_OK;
}

int lt_sync(lt_ctx* c)
{
    CHECK_CTX(c);
    BIND(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return collect_log_stats(c);
}

int lt_last_kernel_ms(lt_ctx* c, double* ms)
{
    CHECK_CTX(c);
    if (!ms) return c->fail(LT_E_INVALID, "lt_last_kernel_ms: null output");
    if (!c->timed) return c->fail(LT_E_STATE, "lt_last_kernel_ms: no launch recorded");
    BIND(c);
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIP_TRY(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = (double)f;
    return LT_OK;
}

int lt_last_log_stages(lt_ctx* c, double ms_out[4], uint64_t* records, uint64_t* batches)
{
    CHECK_CTX(c);
    if (!ms_out) return c->fail(LT_E_INVALID, "lt_last_log_stages: null output");
    if (!c->stages_valid) return c->fail(LT_E_STATE, "lt_last_log_stages: the last launch did not use the log tally");
    BIND(c);
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    { int rc3 = collect_log_stats(c); if (rc3) return rc3; }
    for (int k = 0; k < 4; k++) ms_out[k] = 0.0;
    for (int l = 0; l < kMaxLanes; l++) {
        const LogLane& ln = c->lanes[l];
        for (size_t b = 0; b + 5 <= ln.ev_used; b += 5) {
            float f;
            HIP_TRY(c, hipEventElapsedTime(&f, ln.evs[b], ln.evs[b + 1])); ms_out[0] += f;       // walk
            HIP_TRY(c, hipEventElapsedTime(&f, ln.evs[b + 1], ln.evs[b + 2])); ms_out[1] += f;   // scan
            HIP_TRY(c, hipEventElapsedTime(&f, ln.evs[b + 2], ln.evs[b + 3])); ms_out[2] += f;   // partition pass(es)
            HIP_TRY(c, hipEventElapsedTime(&f, ln.evs[b + 3], ln.evs[b + 4])); ms_out[3] += f;   // tile reduce
        }
    }
    if (records) *records = c->last_records;
    if (batches) *batches = c->last_batches;
    return LT_OK;
}

int lt_last_log_info(lt_ctx* c, uint64_t* records, uint64_t* overflow_records, uint64_t* batches, int* lanes)
{
    CHECK_CTX(c);
    if (!c->stages_valid) return c->fail(LT_E_STATE, "lt_last_log_info: the last launch did not use the log tally");
    BIND(c);
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    { int rc3 = collect_log_stats(c); if (rc3) return rc3; }
    if (records) *records = c->last_records;
    if (overflow_records) *overflow_records = c->last_overflow;
    if (batches) *batches = c->last_batches;
    if (lanes) *lanes = c->last_lanes;
    return LT_OK;
}

int lt_read_grid(lt_ctx* c, void* host_out, size_t bytes)
{
    CHECK_CTX(c);
    if (!c->have_grid) return c->fail(LT_E_STATE, "lt_read_grid: no grid");
    if (!host_out || bytes != c->n_vox() * c->grid_elem())
        return c->fail(LT_E_INVALID, "lt_read_grid: expected %zu bytes", c->n_vox() * c->grid_elem());
    BIND(c);
    HIP_TRY(c, hipMemcpyAsync(host_out, c->d_grid.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_read_grid_f64(lt_ctx* c, double* host_out, size_t n_voxels)
{
    CHECK_CTX(c);
    if (!c->have_grid) return c->fail(LT_E_STATE, "lt_read_grid_f64: no grid");
    if (!host_out || n_voxels != c->n_vox()) return c->fail(LT_E_INVALID, "lt_read_grid_f64: expected %zu voxels", c->n_vox());
    BIND(c);
    if (c->tally == LT_TALLY_F64) return lt_read_grid(c, host_out, n_voxels * 8);
    HIP_TRY(c, c->d_scratch_out.ensure(n_voxels * sizeof(double)));
    HIP_TRY(c, launch_grid_to_f64(c->d_grid.p, c->tally, n_voxels, (double*)c->d_scratch_out.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(host_out, c->d_scratch_out.p, n_voxels * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_read_counters(lt_ctx* c, lt_counters* out)
{
    CHECK_CTX(c);
    if (!out) return c->fail(LT_E_INVALID, "lt_read_counters: null output");
    BIND(c);
    DevCounters h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_counters.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->photons = h.photons; out->steps = h.steps;
    out->w_absorbed = h.w[CW_ABSORBED]; out->w_lost_outside_grid = h.w[CW_LOST];
    out->w_escaped_top = h.w[CW_ESC_TOP]; out->w_escaped_bottom = h.w[CW_ESC_BOT];
    out->w_escaped_mesh = h.w[CW_ESC_MESH]; out->w_specular = h.w[CW_SPECULAR];
    out->w_roulette_net = h.w[CW_ROULETTE]; out->w_capped = h.w[CW_CAPPED];
    return LT_OK;
}

int lt_grid_device_ptr(lt_ctx* c, void** ptr, size_t* bytes)
{
    CHECK_CTX(c);
    if (!c->have_grid) return c->fail(LT_E_STATE, "lt_grid_device_ptr: no grid");
    if (ptr) *ptr = c->d_grid.p;
    if (bytes) *bytes = c->n_vox() * c->grid_elem();
    return LT_OK;
}

int lt_counters_device_ptr(lt_ctx* c, void** ptr, size_t* bytes)
{
    CHECK_CTX(c);
    if (ptr) *ptr = c->d_counters.p;
    if (bytes) *bytes = sizeof(DevCounters);
    return LT_OK;
}

void* lt_stream(lt_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---- RCCL (loaded lazily; no link-time dependency) ------------------------
namespace {
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_reduce_t)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
typedef int (*nccl_group_t)(void);
struct Rccl {
    void* h = nullptr; nccl_allreduce_t allreduce = nullptr; nccl_reduce_t reduce = nullptr;
    nccl_group_t gstart = nullptr, gend = nullptr;
} g_rccl;
// ncclDataType_t / ncclRedOp_t values (nccl.h): ncclUint64 = 5, ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0
enum { kNcclUint64 = 5, kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0 };
}  // namespace

int lt_reduce_grid(lt_ctx* c, void* comm, int root)
{
    CHECK_CTX(c);
    if (!comm) return c->fail(LT_E_INVALID, "lt_reduce_grid: null communicator");
    if (!c->have_grid) return c->fail(LT_E_STATE, "lt_reduce_grid: no grid");
    // contexts are independent across threads (lt.h): the library is loaded exactly once
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        g_rccl.allreduce = (nccl_allreduce_t)dlsym(h, "ncclAllReduce");
        g_rccl.reduce = (nccl_reduce_t)dlsym(h, "ncclReduce");
        g_rccl.gstart = (nccl_group_t)dlsym(h, "ncclGroupStart");
        g_rccl.gend = (nccl_group_t)dlsym(h, "ncclGroupEnd");
        g_rccl.h = h;
    });
    if (!g_rccl.h) return c->fail(LT_E_UNSUPPORTED, "lt_reduce_grid: cannot load librccl.so");
    if (!g_rccl.allreduce || !g_rccl.reduce || !g_rccl.gstart || !g_rccl.gend)
        return c->fail(LT_E_UNSUPPORTED, "lt_reduce_grid: librccl.so lacks ncclAllReduce/ncclReduce");
    BIND(c);
    const int dt = c->tally == LT_TALLY_F32 ? kNcclFloat32 : (c->tally == LT_TALLY_F64 ? kNcclFloat64 : kNcclUint64);
    DevCounters* dc = (DevCounters*)c->d_counters.p;
    int rc = g_rccl.gstart();
    if (root < 0) {
        if (!rc) rc = g_rccl.allreduce(c->d_grid.p, c->d_grid.p, c->n_vox(), dt, kNcclSum, comm, c->stream);
        if (!rc) rc = g_rccl.allreduce(&dc->photons, &dc->photons, 2, kNcclUint64, kNcclSum, comm, c->stream);
        if (!rc) rc = g_rccl.allreduce(dc->w, dc->w, 8, kNcclFloat64, kNcclSum, comm, c->stream);
    } else {
        if (!rc) rc = g_rccl.reduce(c->d_grid.p, c->d_grid.p, c->n_vox(), dt, kNcclSum, root, comm, c->stream);
        if (!rc) rc = g_rccl.reduce(&dc->photons, &dc->photons, 2, kNcclUint64, kNcclSum, root, comm, c->stream);
        if (!rc) rc = g_rccl.reduce(dc->w, dc->w, 8, kNcclFloat64, kNcclSum, root, comm, c->stream);
    }
    int rc2 = g_rccl.gend();
    if (rc || rc2) return c->fail(LT_E_HIP, "lt_reduce_grid: RCCL error %d", rc ? rc : rc2);
    return LT_OK;
}

// ---- device-side queries ---------------------------------------------------
namespace {
int stage_in(lt_ctx* c, DevBuf& b, const void* h, size_t bytes)
{
    HIP_TRY(c, b.ensure(bytes));
    HIP_TRY(c, hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, c->stream));
    return LT_OK;
}
}  // namespace

// the link tables of the ctx mesh on the device (rebuilt per call: a few KB; the mesh may have changed)
static int upload_links(lt_ctx* c)      // (upload_tables has put them on the device with the mesh)
{
    if (!c->have_links) return c->fail(LT_E_UNSUPPORTED, "BVH of %zu nodes: the front-to-back order tables hold 16-bit links", c->nodes.size());
    return LT_OK;
}

int lt_intersect_rays(lt_ctx* c, const double* origins, const double* dirs, const double* tmax, size_t n, int use_bvh,
                      int32_t* prim_out, double* t_out)
{
    CHECK_CTX(c);
    if (!c->have_mesh) return c->fail(LT_E_STATE, "lt_intersect_rays: lt_set_mesh first");
    if (n == 0) return LT_OK;
    if (!origins || !dirs || !prim_out || !t_out) return c->fail(LT_E_INVALID, "lt_intersect_rays: null argument");
    if (use_bvh < 0 || use_bvh > 4) return c->fail(LT_E_INVALID, "lt_intersect_rays: use_bvh must be 0 (brute force), 1 (BVH), 2 (march grid, wave-cooperative), 3 (march grid, lane by lane) or 4 (BVH, front to back)");
    BIND(c);
    if (c->media.empty()) { lt_medium m = {0, 0, 0, 1}; c->media.push_back(m); }
    int rc = upload_tables(c);
    if (rc) return rc;
    if (use_bvh == 4 && (rc = upload_links(c))) return rc;
    if ((use_bvh == 2 || use_bvh == 3) && !c->have_march) {      // small meshes have no march grid of their own: build one on request
        if ((rc = build_march_grid(c))) return rc;
        if (!c->have_march) return c->fail(LT_E_UNSUPPORTED, "lt_intersect_rays: no march grid for this mesh");
    }
    // layout of the staging buffer: origins | dirs | tmax
    const size_t vb = n * 3 * sizeof(double), tb = n * sizeof(double);
    HIP_TRY(c, c->d_scratch_in.ensure(2 * vb + tb));
    char* base = (char*)c->d_scratch_in.p;
    HIP_TRY(c, hipMemcpyAsync(base, origins, vb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + vb, dirs, vb, hipMemcpyHostToDevice, c->stream));
    if (tmax) HIP_TRY(c, hipMemcpyAsync(base + 2 * vb, tmax, tb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, c->d_scratch_out.ensure(tb));
    HIP_TRY(c, c->d_scratch_aux.ensure(n * sizeof(int32_t)));
    HIP_TRY(c, launch_intersect_rays(c->d_tris[0].p, c->d_nodes[0].p, (int)c->med_front.size(), (int)c->nodes.size(),
                                     (const double*)base, (const double*)(base + vb),
                                     tmax ? (const double*)(base + 2 * vb) : nullptr, n, use_bvh, c->have_march ? &c->mgrid : nullptr,
                                     use_bvh == 4 ? (const int16_t*)c->d_links.p : nullptr, (int32_t*)c->d_scratch_aux.p, (double*)c->d_scratch_out.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(prim_out, c->d_scratch_aux.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(t_out, c->d_scratch_out.p, tb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_triangle_intersect(lt_ctx* c, const double* origins, const double* dirs, const double* tris, size_t n, double* t_out)
{
    CHECK_CTX(c);
    if (n == 0) return LT_OK;
    if (!origins || !dirs || !tris || !t_out) return c->fail(LT_E_INVALID, "lt_triangle_intersect: null argument");
    BIND(c);
    const size_t vb = n * 3 * sizeof(double), qb = n * 9 * sizeof(double);
    HIP_TRY(c, c->d_scratch_in.ensure(2 * vb + qb));
    char* base = (char*)c->d_scratch_in.p;
    HIP_TRY(c, hipMemcpyAsync(base, origins, vb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + vb, dirs, vb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + 2 * vb, tris, qb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, c->d_scratch_out.ensure(n * sizeof(double)));
    HIP_TRY(c, launch_triangle_intersect((const double*)base, (const double*)(base + vb), (const double*)(base + 2 * vb), n,
                                         (double*)c->d_scratch_out.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(t_out, c->d_scratch_out.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_intersect_bounds(lt_ctx* c, const double* origins, const double* dirs, const double* tmax, const double* boxes,
                        size_t n, int32_t* hit_out)
{
    CHECK_CTX(c);
    if (n == 0) return LT_OK;
    if (!origins || !dirs || !boxes || !hit_out) return c->fail(LT_E_INVALID, "lt_intersect_bounds: null argument");
    BIND(c);
    const size_t vb = n * 3 * sizeof(double), bb = n * 6 * sizeof(double), tb = n * sizeof(double);
    HIP_TRY(c, c->d_scratch_in.ensure(2 * vb + bb + tb));
    char* base = (char*)c->d_scratch_in.p;
    HIP_TRY(c, hipMemcpyAsync(base, origins, vb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + vb, dirs, vb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(base + 2 * vb, boxes, bb, hipMemcpyHostToDevice, c->stream));
    if (tmax) HIP_TRY(c, hipMemcpyAsync(base + 2 * vb + bb, tmax, tb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, c->d_scratch_aux.ensure(n * sizeof(int32_t)));
    HIP_TRY(c, launch_intersect_bounds((const double*)base, (const double*)(base + vb),
                                       tmax ? (const double*)(base + 2 * vb + bb) : nullptr,
                                       (const double*)(base + 2 * vb), n, (int32_t*)c->d_scratch_aux.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hit_out, c->d_scratch_aux.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_eval(lt_ctx* c, int fn, const double* in, size_t n, double* out)
{
    CHECK_CTX(c);
    static const int k_in[10] = {2, 2, 3, 2, 8, 6, 8, 5, 1, 1}, k_out[10] = {1, 1, 6, 2, 4, 3, 5, 3, 5, 3};
    if (fn < 0 || fn > 9) return c->fail(LT_E_INVALID, "lt_eval: unknown function %d", fn);
    if (n == 0) return LT_OK;
    if (!in || !out) return c->fail(LT_E_INVALID, "lt_eval: null argument");
    BIND(c);
    int rc = stage_in(c, c->d_scratch_in, in, n * k_in[fn] * sizeof(double));
    if (rc) return rc;
    HIP_TRY(c, c->d_scratch_out.ensure(n * k_out[fn] * sizeof(double)));
    HIP_TRY(c, launch_eval(fn, (const double*)c->d_scratch_in.p, n, (double*)c->d_scratch_out.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch_out.p, n * k_out[fn] * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_rng_raw(lt_ctx* c, uint64_t seed, uint64_t photon_id, uint32_t count, uint32_t* out)
{
    CHECK_CTX(c);
    if (count == 0) return LT_OK;
    if (!out) return c->fail(LT_E_INVALID, "lt_rng_raw: null output");
    BIND(c);
    HIP_TRY(c, c->d_scratch_aux.ensure((size_t)count * 4));
    HIP_TRY(c, launch_rng_raw(seed, photon_id, count, (uint32_t*)c->d_scratch_aux.p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch_aux.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_set_vertex_capture(lt_ctx* c, uint32_t max_vertices_per_photon)
{
    CHECK_CTX(c);
    if (max_vertices_per_photon > 4096) return c->fail(LT_E_INVALID, "lt_set_vertex_capture: at most 4096 vertices per photon");
    c->max_vertices = max_vertices_per_photon;
    return LT_OK;
}

int lt_read_vertices(lt_ctx* c, lt_vertex* vertices_out, uint32_t* counts_out, uint64_t n_photons)
{
    CHECK_CTX(c);
    // sizes come from the capturing launch, not from a later lt_set_vertex_capture
    if (c->captured_photons == 0 || c->captured_max_vertices == 0) return c->fail(LT_E_STATE, "lt_read_vertices: the last launch captured nothing");
    if (!vertices_out || !counts_out || n_photons != c->captured_photons)
        return c->fail(LT_E_INVALID, "lt_read_vertices: expected buffers for %llu photons", (unsigned long long)c->captured_photons);
    if (c->max_vertices != c->captured_max_vertices)
        return c->fail(LT_E_STATE, "lt_read_vertices: lt_set_vertex_capture changed (%u -> %u) since the capturing launch",
                       c->captured_max_vertices, c->max_vertices);
    BIND(c);
    HIP_TRY(c, hipMemcpyAsync(counts_out, c->d_vcnt.p, (size_t)n_photons * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(vertices_out, c->d_vtx.p, (size_t)n_photons * c->captured_max_vertices * sizeof(lt_vertex), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_set_surface_materials(lt_ctx* c, const lt_surface_material* mats, int n_tris)
{
    CHECK_CTX(c);
    if (!c->have_mesh) return c->fail(LT_E_STATE, "lt_set_surface_materials: lt_set_mesh first");
    if (!mats || n_tris != (int)c->med_front.size())
        return c->fail(LT_E_INVALID, "lt_set_surface_materials: need one record per mesh triangle (%zu)", c->med_front.size());
    c->surf_mats.assign(mats, mats + n_tris);
    return LT_OK;
}

int lt_set_lights(lt_ctx* c, const lt_point_light* lights, int n)
{
    CHECK_CTX(c);
    if (!lights || n <= 0) return c->fail(LT_E_INVALID, "lt_set_lights: need at least one light sample");
    c->lights.assign(lights, lights + n);
    return LT_OK;
}

static int render_impl(lt_ctx* c, int variant, int choices, int width, int height, int samples, int max_depth,
                       const double camera[3], double f_distance, const double* xs, const double* ys, double* rand_0,
                       const double* rand_1, const int32_t* light_choice, double* image)
{
    CHECK_CTX(c);
    if (!c->have_mesh || c->surf_mats.empty() || c->lights.empty())
        return c->fail(LT_E_STATE, "lt_render_surface: lt_set_mesh, lt_set_surface_materials and lt_set_lights first");
    if (width <= 0 || height <= 0 || samples <= 0 || max_depth <= 0 || !camera || !xs || !ys || !rand_0 || !rand_1 ||
        !light_choice || !image || choices <= 0)
        return c->fail(LT_E_INVALID, "lt_render_surface: bad argument");
    if (variant == 1 && max_depth > kRenderOldMaxDepth)
        return c->fail(LT_E_INVALID, "lt_render_surface_old: max_depth %d > %d (the recursion is unrolled on a fixed stack)",
                       max_depth, kRenderOldMaxDepth);
    const size_t n_tab = (size_t)width * height * samples * max_depth;
    const size_t n_lc = (size_t)width * height * samples * choices;
    for (size_t k = 0; k < n_lc; k++)
        if (light_choice[k] < 0 || light_choice[k] >= (int)c->lights.size())
            return c->fail(LT_E_INVALID, "lt_render_surface: light_choice[%zu] out of range", k);
    BIND(c);
    if (c->media.empty()) { lt_medium m = {0, 0, 0, 1}; c->media.push_back(m); }
    int rc = upload_tables(c);
    if (rc) return rc;
    const size_t n_img = (size_t)width * height * 3;
    HIP_TRY(c, c->d_mats.ensure(c->surf_mats.size() * sizeof(lt_surface_material)));
    HIP_TRY(c, c->d_lights.ensure(c->lights.size() * sizeof(lt_point_light)));
    HIP_TRY(c, c->d_r0.ensure(n_tab * 8)); HIP_TRY(c, c->d_r1.ensure(n_tab * 8)); HIP_TRY(c, c->d_lc.ensure(n_lc * 4));
    HIP_TRY(c, c->d_img.ensure(n_img * 8)); HIP_TRY(c, c->d_xy.ensure((size_t)(width + height) * 8));
    if ((rc = upload_links(c))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_mats.p, c->surf_mats.data(), c->surf_mats.size() * sizeof(lt_surface_material), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_lights.p, c->lights.data(), c->lights.size() * sizeof(lt_point_light), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_r0.p, rand_0, n_tab * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_r1.p, rand_1, n_tab * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_lc.p, light_choice, n_lc * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_img.p, image, n_img * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_xy.p, xs, (size_t)width * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync((double*)c->d_xy.p + width, ys, (size_t)height * 8, hipMemcpyHostToDevice, c->stream));
    RenderParams P;
    std::memset(&P, 0, sizeof P);
    P.tris = c->d_tris[0].p; P.nodes = c->d_nodes[0].p;
    P.mats = (const lt_surface_material*)c->d_mats.p; P.lights = (const lt_point_light*)c->d_lights.p;
    P.links = (const int16_t*)c->d_links.p;
    P.n_tris = (int)c->med_front.size(); P.n_nodes = (int)c->nodes.size(); P.n_lights = (int)c->lights.size();
    P.W = width; P.H = height; P.S = samples; P.D = max_depth;
    for (int k = 0; k < 3; k++) P.cam[k] = camera[k];
    P.f_distance = f_distance;
    P.xs = (const double*)c->d_xy.p; P.ys = (const double*)c->d_xy.p + width;
    P.rand_0 = (double*)c->d_r0.p; P.rand_1 = (const double*)c->d_r1.p; P.light_choice = (const int32_t*)c->d_lc.p;
    P.image = (double*)c->d_img.p;
    P.variant = variant; P.choices = choices;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    HIP_TRY(c, launch_render_surface(P, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    HIP_TRY(c, hipMemcpyAsync(image, c->d_img.p, n_img * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rand_0, c->d_r0.p, n_tab * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_render_surface(lt_ctx* c, int width, int height, int samples, int max_depth, const double camera[3],
                      double f_distance, const double* xs, const double* ys, double* rand_0, const double* rand_1,
                      const int32_t* light_choice, double* image)
{
    return render_impl(c, 0, max_depth, width, height, samples, max_depth, camera, f_distance, xs, ys, rand_0, rand_1,
                       light_choice, image);
}

int lt_render_surface_old(lt_ctx* c, int width, int height, int samples, int max_depth, const double camera[3],
                          double f_distance, const double* xs, const double* ys, double* rand_0, const double* rand_1,
                          const int32_t* light_choice, int choices_per_sample, double* image)
{
    return render_impl(c, 1, choices_per_sample, width, height, samples, max_depth, camera, f_distance, xs, ys, rand_0,
                       rand_1, light_choice, image);
}

int lt_mesh_accel_info(lt_ctx* c, int* kind, int march_dims[3], uint64_t* march_entries, int clearance_dims[3])
{
    CHECK_CTX(c);
    if (!c->have_mesh) return c->fail(LT_E_STATE, "lt_mesh_accel_info: lt_set_mesh first");
    BIND(c);
    if (c->media.empty()) { lt_medium m = {0, 0, 0, 1}; c->media.push_back(m); }
    int rc = upload_tables(c);
    if (rc) return rc;
    if (kind) *kind = (c->have_clear ? 1 : 0) | (c->have_march ? 2 : 0);
    if (march_dims) { march_dims[0] = c->have_march ? c->mgrid.nx : 0; march_dims[1] = c->have_march ? c->mgrid.ny : 0; march_dims[2] = c->have_march ? c->mgrid.nz : 0; }
    if (march_entries) *march_entries = c->have_march ? (uint64_t)c->march_entries : 0;
    if (clearance_dims) for (int k = 0; k < 3; k++) clearance_dims[k] = c->have_clear ? c->cn[k] : 0;
    return LT_OK;
}

int lt_device_info(lt_ctx* c, char* name, size_t name_len, int* n_cus, int* clock_mhz, size_t* hbm_bytes)
{
    CHECK_CTX(c);
    if (name && name_len) { std::snprintf(name, name_len, "%s (%s)", c->prop.name, c->prop.gcnArchName); }
    if (n_cus) *n_cus = c->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = c->prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = c->prop.totalGlobalMem;
    return LT_OK;
}

}  // extern "C"
