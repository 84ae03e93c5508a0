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

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
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

struct lt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // stage boundaries of the last log-mode batch
    bool stages_valid = false;
    uint64_t last_records = 0, last_batches = 0;
    bool log_stats_pending = false, last_overflow = false;
    uint64_t pending_batch = 0;
    uint32_t pending_cap_chunks = 0;
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
    uint32_t max_vertices = 0;
    int tally_mode = 2;                 // 0: global atomics, 1: deposit log + partition + tile reduce, 2: auto (default)
    size_t log_budget = (size_t)16 << 30; // bytes for the log and its ping-pong copy
    double rec_per_photon = 0.0;        // measured deposit records per photon (sizes the batches)
    size_t log_alloc_records = 0;       // capacity of the log buffers currently allocated
    int log_alloc_elem = 0;
    double last_stage_ms[6] = {0, 0, 0, 0, 0, 0};
    uint64_t captured_photons = 0;
    int blocks_per_cu = 0, threads_per_block = 0;

    // device buffers
    DevBuf d_media[2], d_zb[2], d_lm, d_tris[2], d_nodes[2];  // [0]=f64, [1]=f32
    DevBuf d_grid, d_counters, d_head, d_table, d_scratch_in, d_scratch_out, d_scratch_aux;
    DevBuf d_mats, d_lights, d_r0, d_r1, d_lc, d_img, d_xy, d_vtx, d_vcnt, d_clear;
    int cn[3] = {0, 0, 0};
    double corg[3] = {0, 0, 0}, ccell[3] = {1, 1, 1};
    bool have_clear = false;
    DevBuf d_log_idx, d_log_val, d_tmp_idx, d_tmp_val, d_log_fill, d_log_meta, d_hist, d_tile_base, d_cursor1, d_cursor2, d_items2, d_items_r;
    bool tables_dirty = true;
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
};

#define CHECK_CTX(c) do { if (!(c)) return LT_E_INVALID; } while (0)
#define HIP_TRY(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (c)->hip(e_, #call); } while (0)
#define BIND(c) HIP_TRY(c, hipSetDevice((c)->device))

namespace {

template <typename R>
void fill_media(const std::vector<lt_medium>& in, std::vector<MedD<R>>& out)
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

template <typename T>
int upload(lt_ctx* c, DevBuf& b, const std::vector<T>& h)
{
    HIP_TRY(c, b.ensure(h.size() * sizeof(T)));
    if (!h.empty()) HIP_TRY(c, hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return LT_OK;
}

int upload_tables(lt_ctx* c)
{
    if (!c->tables_dirty) return LT_OK;
    int rc;
    // staging vectors live until the stream has drained (pageable H2D copies)
    std::vector<MedD<double>> m64; std::vector<MedD<float>> m32;
    std::vector<double> z64; std::vector<float> z32;
    std::vector<TriD<double>> t64; std::vector<TriD<float>> t32;
    std::vector<NodeD<double>> n64; std::vector<NodeD<float>> n32;
    fill_media(c->media, m64); fill_media(c->media, m32);
    if ((rc = upload(c, c->d_media[0], m64))) return rc;
    if ((rc = upload(c, c->d_media[1], m32))) return rc;
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
        // clearance grid over the root bounds: 64 cells along the longest axis (LT_NO_CLEARANCE=1 disables it)
        c->have_clear = false;
        if (!std::getenv("LT_NO_CLEARANCE")) {
            double ext[3], longest = 0;
            for (int k = 0; k < 3; k++) { ext[k] = c->nodes[0].hi[k] - c->nodes[0].lo[k]; if (ext[k] > longest) longest = ext[k]; }
            if (longest > 0 && std::isfinite(longest)) {
                // cells along the longest axis: as fine as a brute-force build (cells x triangles distance evaluations)
                // of ~2e9 evaluations allows, between 32 and 128 (C4's 30 triangles: 128, 8 MiB; a 5000-triangle mesh: 73)
                int per_axis = (int)std::cbrt(2.0e9 / (double)(t64.empty() ? 1 : t64.size()));
                per_axis = per_axis < 32 ? 32 : (per_axis > 128 ? 128 : per_axis);
                // LT_CLEARANCE_CELLS overrides (tuning experiments)
                if (const char* e = std::getenv("LT_CLEARANCE_CELLS")) { int v = std::atoi(e); if (v >= 8 && v <= 512) per_axis = v; }
                const double h = longest / (double)per_axis;
                for (int k = 0; k < 3; k++) {
                    c->cn[k] = (int)std::ceil(ext[k] / h); if (c->cn[k] < 1) c->cn[k] = 1;
                    c->ccell[k] = ext[k] > 0 ? ext[k] / c->cn[k] : h; c->corg[k] = c->nodes[0].lo[k];
                }
                const size_t cells = (size_t)c->cn[0] * c->cn[1] * c->cn[2];
                HIP_TRY(c, c->d_clear.ensure(cells * sizeof(float)));
                HIP_TRY(c, launch_build_clearance(c->d_tris[0].p, (int)t64.size(), (float*)c->d_clear.p, c->cn[0], c->cn[1], c->cn[2],
                                                  c->corg, c->ccell, c->stream));
                c->have_clear = true;
            }
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tables_dirty = false;
    return LT_OK;
}

// Read back the statistics of the last log-mode batch (chunks claimed, records logged) once the stream has
// drained, and update the deposit-record rate that sizes the next launch's log and batches.
int collect_log_stats(lt_ctx* c)
{
    if (!c->log_stats_pending) return LT_OK;
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h, c->d_log_meta.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->log_stats_pending = false;
    c->last_overflow = h[0] > c->pending_cap_chunks;
    c->last_records = h[2];
    if (c->pending_batch > 0) {
        double rate = (double)h[2] / (double)c->pending_batch;
        if (c->last_overflow) rate *= 2.0;   // part of the batch went through atomics: the true rate is higher
        c->rec_per_photon = rate > c->rec_per_photon ? rate : 0.5 * (rate + c->rec_per_photon);
    }
    return LT_OK;
}

// depth of the flattened tree (bounds the traversal stack) + structural checks
int bvh_depth(const std::vector<lt_bvh_node>& n, int n_tris, int idx, int depth, int* max_depth, int* visited)
{
    if (idx < 0 || idx >= (int)n.size() || depth > 64) return -1;
    (*visited)++;
    if (depth > *max_depth) *max_depth = depth;
    const lt_bvh_node& nd = n[idx];
    if (nd.n_prims > 0) {
        if (nd.offset < 0 || nd.offset + nd.n_prims > n_tris) return -1;
        return 0;
    }
    if (nd.axis < 0 || nd.axis > 2 || nd.offset <= idx + 1) return -1;
    if (bvh_depth(n, n_tris, idx + 1, depth + 1, max_depth, visited)) return -1;
    return bvh_depth(n, n_tris, nd.offset, depth + 1, max_depth, visited);
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
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device_id);
    if (e == hipSuccess && std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device arch ") + c->prop.gcnArchName + " is not gfx950: kernels are built for MI355X only";
        delete c;
        return LT_E_UNSUPPORTED;
    }
    if (e == hipSuccess) {
        size_t quarter = c->prop.totalGlobalMem / 4;
        c->log_budget = quarter < ((size_t)64 << 30) ? quarter : ((size_t)64 << 30);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    for (int k = 0; k < 6 && e == hipSuccess; k++) e = hipEventCreate(&c->evs[k]);
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
    for (int i = 0; i < 2; i++) { c->d_media[i].release(); c->d_zb[i].release(); c->d_tris[i].release(); c->d_nodes[i].release(); }
    c->d_lm.release(); c->d_grid.release(); c->d_counters.release(); c->d_head.release(); c->d_table.release();
    c->d_scratch_in.release(); c->d_scratch_out.release(); c->d_scratch_aux.release();
    c->d_mats.release(); c->d_lights.release(); c->d_r0.release(); c->d_r1.release(); c->d_lc.release();
    c->d_img.release(); c->d_xy.release(); c->d_vtx.release(); c->d_vcnt.release(); c->d_clear.release();
    c->d_log_idx.release(); c->d_log_val.release(); c->d_tmp_idx.release(); c->d_tmp_val.release(); c->d_log_fill.release();
    c->d_log_meta.release(); c->d_hist.release(); c->d_tile_base.release(); c->d_cursor1.release(); c->d_cursor2.release();
    c->d_items2.release(); c->d_items_r.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int k = 0; k < 6; k++) if (c->evs[k]) (void)hipEventDestroy(c->evs[k]);
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
    c->rec_per_photon = 0.0;
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
    int depth = 0, visited = 0;
    if (bvh_depth(nn, n_tris, 0, 0, &depth, &visited) != 0 || visited != n_nodes)
        return c->fail(LT_E_INVALID, "lt_set_mesh: BVH is not a valid pre-order tree over %d triangles", n_tris);
    if (depth >= 31) return c->fail(LT_E_UNSUPPORTED, "lt_set_mesh: BVH depth %d exceeds the traversal stack (31)", depth);
    size_t covered = 0;
    for (const auto& nd : nn) if (nd.n_prims > 0) covered += (size_t)nd.n_prims;
    if (covered != (size_t)n_tris) return c->fail(LT_E_INVALID, "lt_set_mesh: leaves cover %zu of %d triangles", covered, n_tris);
    c->verts.assign(verts, verts + (size_t)n_tris * 9);
    c->med_front.assign(med_front, med_front + n_tris);
    c->med_back.assign(med_back, med_back + n_tris);
    c->nodes.swap(nn);
    c->have_mesh = true; c->have_layers = false;
    c->surf_mats.clear();
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
    return LT_OK;
}

int lt_set_max_steps(lt_ctx* c, uint32_t max_steps)
{
    CHECK_CTX(c);
    if (max_steps == 0) return c->fail(LT_E_INVALID, "lt_set_max_steps: must be > 0");
    c->max_steps = max_steps;
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
    if (std::getenv("LT_DIAG_NO_TALLY")) v.tally = 3;  // diagnostic: time the walk without deposition
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
    P.media = c->d_media[pi].p; P.zb = c->d_zb[pi].p; P.layer_medium = (const int32_t*)c->d_lm.p;
    P.tris = c->d_tris[pi].p; P.nodes = c->d_nodes[pi].p;
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
    P.src_type = c->src_type; P.start_medium = c->start_medium;
    P.table = (const double*)c->d_table.p; P.table_steps = table_steps;
    P.max_steps = c->max_steps;
    P.counters = (DevCounters*)c->d_counters.p;
    if (c->have_mesh && c->have_clear) {
        P.clear = (const float*)c->d_clear.p; P.cnx = c->cn[0]; P.cny = c->cn[1]; P.cnz = c->cn[2];
        for (int k = 0; k < 3; k++) { P.corg[k] = c->corg[k]; P.cinv[k] = 1.0 / c->ccell[k]; }
    }
    c->captured_photons = 0;
    if (c->max_vertices > 0) {
        const size_t vb = (size_t)n_photons * c->max_vertices * sizeof(lt_vertex);
        if (vb > ((size_t)64 << 30)) return c->fail(LT_E_NOMEM, "lt_launch: vertex capture needs %zu bytes (> 64 GiB)", vb);
        HIP_TRY(c, c->d_vtx.ensure(vb));
        HIP_TRY(c, c->d_vcnt.ensure((size_t)n_photons * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(c->d_vcnt.p, 0, (size_t)n_photons * sizeof(uint32_t), c->stream));
        P.vertices = (lt_vertex*)c->d_vtx.p; P.vertex_counts = (uint32_t*)c->d_vcnt.p; P.max_vertices = c->max_vertices;
        c->captured_photons = n_photons;
    }

    LaunchCfg cfg;
    cfg.threads = c->threads_per_block > 0 ? c->threads_per_block : 256;
    cfg.lds_bytes = walk_lds_bytes(v, P.n_media, P.n_layers, P.n_tris, P.n_nodes);
    if (v.mesh && cfg.lds_bytes > kMeshLdsBudget) {
        // large mesh: leave triangles and nodes in global memory (L2 / Infinity Cache resident)
        if (v.table) return c->fail(LT_E_UNSUPPORTED, "lt_launch: table RNG with a mesh beyond the LDS budget");
        v.mesh = 2;
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
    const uint32_t ntx = ((uint32_t)c->nx + 31u) >> kTileBX, nty = ((uint32_t)c->ny + 31u) >> kTileBY,
                   ntz = ((uint32_t)c->nz + 15u) >> kTileBZ;
    const uint32_t n_tiles = ntx * nty * ntz;   // 32 x 32 x 16-voxel blocks
    const uint32_t n_tiles_pre = (uint32_t)((c->n_vox() + kTileSize - 1) >> kTileShift);
    (void)n_tiles_pre;
    // auto: slab walks and f32 mesh walks are paced by the atomic unit -> log (C4, f32: 43 ms log vs 60 ms atomic);
    // the f64 mesh walk's BVH arithmetic hides the atomics, so the log passes would only add time (58 vs 59 ms)
    const int mode = c->tally_mode == 2 ? ((c->have_mesh && !v.f32) ? 0 : 1) : c->tally_mode;
    { int rc3 = collect_log_stats(c); if (rc3) return rc3; }   // stats of the previous launch size this one
    bool use_log = mode == 1 && !v.table && c->max_vertices == 0 && n_tiles <= 16384 && !std::getenv("LT_DIAG_NO_TALLY");
    if (use_log) {
        const size_t rec_bytes = 4 + c->grid_elem();
        size_t budget_records = c->log_budget / (2 * rec_bytes);
        if (budget_records > 0xFFF00000ull) budget_records = 0xFFF00000ull;   // 32-bit record offsets
        if (budget_records < 64 * (size_t)kLogChunk) return c->fail(LT_E_INVALID, "lt_launch: log budget too small (%zu bytes)", c->log_budget);
        // size the log to the job: measured records per photon when known, else a pilot-sized log
        // (a log that turns out too small only diverts the excess deposits to atomics)
        auto size_log = [&](uint64_t photons_left) -> size_t {
            // no measurement yet: assume 200 records per photon (tissue-like media give 100-300); a wrong guess is
            // corrected after the first batch and an undersized log only diverts the excess deposits to atomics
            const double rate = c->rec_per_photon > 0.0 ? c->rec_per_photon : 200.0;
            double need = 1.25 * rate * (double)photons_left + 1048576.0;
            size_t r = need < (double)budget_records ? (size_t)need : budget_records;
            return ((r + kLogChunk - 1) / kLogChunk) * kLogChunk;
        };
        size_t cap_records = size_log(n_photons);
        uint32_t cap_chunks = (uint32_t)(cap_records / kLogChunk);
        auto ensure_log = [&]() -> hipError_t {
            hipError_t e;
            if (c->log_alloc_elem != (int)c->grid_elem()) c->log_alloc_records = 0;
            if (cap_records <= c->log_alloc_records) {   // use everything that is already there
                cap_records = c->log_alloc_records; cap_chunks = (uint32_t)(cap_records / kLogChunk);
                return hipSuccess;
            }
            // grow geometrically so that run-to-run jitter of the record rate does not re-allocate tens of GB
            size_t grown = cap_records + cap_records / 8;
            if (grown > budget_records) grown = budget_records;
            cap_records = (grown / kLogChunk) * kLogChunk; cap_chunks = (uint32_t)(cap_records / kLogChunk);
            c->log_alloc_records = cap_records; c->log_alloc_elem = (int)c->grid_elem();
            if ((e = c->d_log_idx.ensure(cap_records * 4)) != hipSuccess) return e;
            if ((e = c->d_tmp_idx.ensure(cap_records * 4)) != hipSuccess) return e;
            if ((e = c->d_log_val.ensure(cap_records * c->grid_elem())) != hipSuccess) return e;
            if ((e = c->d_tmp_val.ensure(cap_records * c->grid_elem())) != hipSuccess) return e;
            return c->d_log_fill.ensure((size_t)cap_chunks * 4);
        };
        if (ensure_log() != hipSuccess) {
            // not enough free HBM for the log (other contexts, a huge grid): this launch deposits with atomics instead
            (void)hipGetLastError();
            c->d_log_idx.release(); c->d_tmp_idx.release(); c->d_log_val.release(); c->d_tmp_val.release(); c->d_log_fill.release();
            c->log_alloc_records = 0;
            use_log = false;
        }
    }
    if (use_log) {
        // batches use what is allocated; later batches shrink or grow with the measured record rate
        const size_t cap_records = c->log_alloc_records;
        const uint32_t cap_chunks = (uint32_t)(cap_records / kLogChunk);
        HIP_TRY(c, c->d_log_meta.ensure(64));
        uint32_t bits2 = 1; while ((1u << (2 * bits2)) < n_tiles) bits2++;
        // grids of <= 1024 tiles (256^3): ONE partition pass straight to tiles.  Records cluster in the few dozen
        // tiles around the source, so the per-tile runs of a 4096-record work item stay long enough to coalesce.
        if (n_tiles <= 1024) bits2 = 0;
        if (const char* e = std::getenv("LT_LOG_BITS2")) bits2 = (uint32_t)std::atoi(e);
        const uint32_t nb1 = (n_tiles + (1u << bits2) - 1) >> bits2;
        HIP_TRY(c, c->d_hist.ensure((size_t)n_tiles * 4)); HIP_TRY(c, c->d_tile_base.ensure((size_t)(n_tiles + 1) * 4));
        HIP_TRY(c, c->d_cursor1.ensure((size_t)nb1 * 4)); HIP_TRY(c, c->d_cursor2.ensure((size_t)n_tiles * 4));
        HIP_TRY(c, c->d_items2.ensure((size_t)(nb1 + 1) * 4)); HIP_TRY(c, c->d_items_r.ensure((size_t)(n_tiles + 1) * 4));
        uint32_t* meta = (uint32_t*)c->d_log_meta.p;   // [0] next chunk, [2..3] totals
        P.log_idx = (uint32_t*)c->d_log_idx.p; P.log_val = c->d_log_val.p; P.log_fill = (uint32_t*)c->d_log_fill.p;
        P.log_next = meta; P.log_cap_chunks = cap_chunks;
        P.log_hist = (uint32_t*)c->d_hist.p; P.log_n_tiles = n_tiles; P.log_ntx = ntx; P.log_nty = nty;
        cfg.lds_bytes = walk_lds_bytes(v, P.n_media, P.n_layers, P.n_tris, P.n_nodes, n_tiles);
        {
            const int res2 = walk_max_blocks_per_cu(v, cfg.threads, cfg.lds_bytes);
            if (res2 <= 0) return c->fail(LT_E_HIP, "lt_launch: log-mode kernel not resident");
            const int pc = c->blocks_per_cu > 0 ? c->blocks_per_cu : res2;
            cap = (unsigned long long)pc * (unsigned long long)c->prop.multiProcessorCount;
        }
        LogReduceParams L;
        std::memset(&L, 0, sizeof L);
        L.log_idx = P.log_idx; L.log_val = P.log_val; L.log_fill = P.log_fill;
        L.tmp_idx = (uint32_t*)c->d_tmp_idx.p; L.tmp_val = c->d_tmp_val.p;
        L.hist = (uint32_t*)c->d_hist.p; L.tile_base = (uint32_t*)c->d_tile_base.p;
        L.cursor1 = (uint32_t*)c->d_cursor1.p; L.cursor2 = (uint32_t*)c->d_cursor2.p; L.items2 = (uint32_t*)c->d_items2.p; L.items_r = (uint32_t*)c->d_items_r.p;
        L.totals = meta + 2; L.n_tiles = n_tiles; L.bits2 = bits2;
        L.grid = c->d_grid.p; L.n_vox = c->n_vox(); L.tally = c->tally;
        L.nx = (uint32_t)c->nx; L.ny = (uint32_t)c->ny; L.nz = (uint32_t)c->nz; L.ntx = ntx; L.nty = nty;

        L.chunks_used = meta; L.cap_chunks = cap_chunks; L.work = meta + 5;
        // Everything below is enqueued without a host read-back: item counts stay in device memory and the
        // partition / reduce kernels are persistent work loops.  Statistics of the LAST batch (records, overflow)
        // are collected lazily (collect_log_stats) and steer the batch size of the next launch.
        HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        uint64_t done = 0;
        const bool diag = std::getenv("LT_LOG_TIMING") != nullptr;   // prints per-stage device times (syncs per batch)
        double stage[4] = {0, 0, 0, 0};
        int n_batches = 0;
        while (done < n_photons) {
            uint64_t batch = n_photons - done;
            {
                const double rate = c->rec_per_photon > 0.0 ? c->rec_per_photon : 200.0;
                const double fit = 0.8 * (double)cap_records / rate;
                if ((double)batch > fit) batch = fit < 4096.0 ? 4096 : (uint64_t)fit;
            }
            P.n_photons = batch; P.photon_offset = photon_offset + done;
            want = (batch + (unsigned long long)cfg.threads - 1) / (unsigned long long)cfg.threads;
            cfg.blocks = (int)(want < cap ? want : cap);
            if (cfg.blocks < 1) cfg.blocks = 1;
            HIP_TRY(c, hipMemsetAsync(c->d_head.p, 0, sizeof(unsigned long long), c->stream));
            HIP_TRY(c, hipMemsetAsync(meta, 0, 32, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_hist.p, 0, (size_t)n_tiles * 4, c->stream));
            HIP_TRY(c, hipEventRecord(c->evs[0], c->stream));
            HIP_TRY(c, launch_walk(P, v, cfg, c->stream));
            HIP_TRY(c, hipEventRecord(c->evs[1], c->stream));
            HIP_TRY(c, launch_log_scan(L, c->stream));
            HIP_TRY(c, hipEventRecord(c->evs[2], c->stream));
            HIP_TRY(c, launch_log_part1(L, c->stream));
            LogReduceParams Lr = L;
            if (bits2 == 0) { Lr.log_idx = L.tmp_idx; Lr.log_val = L.tmp_val; }   // single pass: tiles are final in tmp
            else HIP_TRY(c, launch_log_part2(L, c->stream));
            HIP_TRY(c, hipEventRecord(c->evs[4], c->stream));
            HIP_TRY(c, launch_log_reduce(Lr, c->stream));
            HIP_TRY(c, hipEventRecord(c->evs[5], c->stream));
            c->log_stats_pending = true; c->pending_batch = batch; c->pending_cap_chunks = cap_chunks;
            if (diag) {
                int rc2 = collect_log_stats(c);
                if (rc2) return rc2;
                float f;
                (void)hipEventElapsedTime(&f, c->evs[0], c->evs[1]); stage[0] += f;
                (void)hipEventElapsedTime(&f, c->evs[1], c->evs[2]); stage[1] += f;
                (void)hipEventElapsedTime(&f, c->evs[2], c->evs[4]); stage[2] += f;
                (void)hipEventElapsedTime(&f, c->evs[4], c->evs[5]); stage[3] += f;
                std::fprintf(stderr, "[lt log] batch %d: %llu photons, %llu records (%.1f / photon)%s\n", n_batches,
                             (unsigned long long)batch, (unsigned long long)c->last_records,
                             (double)c->last_records / (double)batch, c->last_overflow ? ", LOG OVERFLOW -> atomics" : "");
            }
            n_batches++;
            done += batch;
        }
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        c->stages_valid = true; c->last_batches = (uint64_t)n_batches;
        if (diag)
            std::fprintf(stderr, "[lt log] stages ms: walk %.2f scan %.2f partition %.2f reduce %.2f\n", stage[0], stage[1],
                         stage[2], stage[3]);
        c->timed = true;
        return LT_OK;
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
    const size_t rec_bytes = 4 + c->grid_elem();
    size_t budget_records = c->log_budget / (2 * rec_bytes);
    if (budget_records > 0xFFF00000ull) budget_records = 0xFFF00000ull;
    const double rate = c->rec_per_photon > 0.0 ? c->rec_per_photon : 200.0;
    double need = 1.25 * rate * (double)n_photons + 1048576.0;
    size_t r = need < (double)budget_records ? (size_t)need : budget_records;
    r = ((r + kLogChunk - 1) / kLogChunk) * kLogChunk;
    if (c->log_alloc_elem == (int)c->grid_elem() && r <= c->log_alloc_records) return LT_OK;
    HIP_TRY(c, c->d_log_idx.ensure(r * 4)); HIP_TRY(c, c->d_tmp_idx.ensure(r * 4));
    HIP_TRY(c, c->d_log_val.ensure(r * c->grid_elem())); HIP_TRY(c, c->d_tmp_val.ensure(r * c->grid_elem()));
    HIP_TRY(c, c->d_log_fill.ensure((r / kLogChunk) * 4));
    c->log_alloc_records = r; c->log_alloc_elem = (int)c->grid_elem();
    return LT_OK;
}

int lt_set_tally_mode(lt_ctx* c, int mode, uint64_t log_bytes)
{
    CHECK_CTX(c);
    if (mode < 0 || mode > 2) return c->fail(LT_E_INVALID, "lt_set_tally_mode: mode must be LT_MODE_ATOMIC, LT_MODE_LOG or LT_MODE_AUTO");
    c->tally_mode = mode;
    if (log_bytes) c->log_budget = (size_t)log_bytes;
    c->rec_per_photon = 0.0;
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
    HIP_TRY(c, hipEventSynchronize(c->evs[5]));
    { int rc3 = collect_log_stats(c); if (rc3) return rc3; }
    float f;
    HIP_TRY(c, hipEventElapsedTime(&f, c->evs[0], c->evs[1])); ms_out[0] = f;   // walk
    HIP_TRY(c, hipEventElapsedTime(&f, c->evs[1], c->evs[2])); ms_out[1] = f;   // counts readback + scan
    HIP_TRY(c, hipEventElapsedTime(&f, c->evs[2], c->evs[4])); ms_out[2] = f;   // partition pass(es)
    HIP_TRY(c, hipEventElapsedTime(&f, c->evs[4], c->evs[5])); ms_out[3] = f;   // tile reduce
    if (records) *records = c->last_records;
    if (batches) *batches = c->last_batches;
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
    if (!g_rccl.h) {
        g_rccl.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.h) g_rccl.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.h) return c->fail(LT_E_UNSUPPORTED, "lt_reduce_grid: cannot load librccl.so: %s", dlerror());
        g_rccl.allreduce = (nccl_allreduce_t)dlsym(g_rccl.h, "ncclAllReduce");
        g_rccl.reduce = (nccl_reduce_t)dlsym(g_rccl.h, "ncclReduce");
        g_rccl.gstart = (nccl_group_t)dlsym(g_rccl.h, "ncclGroupStart");
        g_rccl.gend = (nccl_group_t)dlsym(g_rccl.h, "ncclGroupEnd");
        if (!g_rccl.allreduce || !g_rccl.reduce || !g_rccl.gstart || !g_rccl.gend)
            return c->fail(LT_E_UNSUPPORTED, "lt_reduce_grid: librccl.so lacks ncclAllReduce/ncclReduce");
    }
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

int lt_intersect_rays(lt_ctx* c, const double* origins, const double* dirs, const double* tmax, size_t n, int use_bvh,
                      int32_t* prim_out, double* t_out)
{
    CHECK_CTX(c);
    if (!c->have_mesh) return c->fail(LT_E_STATE, "lt_intersect_rays: lt_set_mesh first");
    if (n == 0) return LT_OK;
    if (!origins || !dirs || !prim_out || !t_out) return c->fail(LT_E_INVALID, "lt_intersect_rays: null argument");
    BIND(c);
    if (c->media.empty()) { lt_medium m = {0, 0, 0, 1}; c->media.push_back(m); }
    int rc = upload_tables(c);
    if (rc) return rc;
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
                                     tmax ? (const double*)(base + 2 * vb) : nullptr, n, use_bvh,
                                     (int32_t*)c->d_scratch_aux.p, (double*)c->d_scratch_out.p, c->stream));
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
    static const int k_in[9] = {2, 2, 3, 2, 8, 6, 8, 5, 1}, k_out[9] = {1, 1, 6, 2, 4, 3, 5, 3, 5};
    if (fn < 0 || fn > 8) return c->fail(LT_E_INVALID, "lt_eval: unknown function %d", fn);
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
    if (c->captured_photons == 0 || c->max_vertices == 0) return c->fail(LT_E_STATE, "lt_read_vertices: the last launch captured nothing");
    if (!vertices_out || !counts_out || n_photons != c->captured_photons)
        return c->fail(LT_E_INVALID, "lt_read_vertices: expected buffers for %llu photons", (unsigned long long)c->captured_photons);
    BIND(c);
    HIP_TRY(c, hipMemcpyAsync(counts_out, c->d_vcnt.p, (size_t)n_photons * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(vertices_out, c->d_vtx.p, (size_t)n_photons * c->max_vertices * sizeof(lt_vertex), hipMemcpyDeviceToHost, c->stream));
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
