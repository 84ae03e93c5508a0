// lt_internal.hpp -- layouts shared by the host API (lt_api.cpp, g++) and the
// gfx950 kernels (lt_kernels.hip, hipcc).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/lt.h"

namespace ltk {

// Per-medium constants in walk precision R.  Derived values are computed on the
// host in double, one IEEE operation each, then narrowed.
template <typename R>
struct MedD {
    R mu_t, inv_mu_t, absorb, g;
    R n, one_m_g2, one_p_g2, inv_2g;
};

// Triangle record: role of PreComputedTriangle (primitives.py:99-112).
template <typename R>
struct TriD {
    R a[3], ab[3], ac[3], n[3];
    int32_t med_front, med_back;
};

// Flattened BVH node: role of LinearBVHNode (bvh_new.py:60-67).
template <typename R>
struct NodeD {
    R lo[3], hi[3];
    int32_t offset, n_prims, axis;
    int32_t skip;  // first node after this node's subtree (stackless traversal)
};

struct DevCounters {
    unsigned long long photons, steps;
    double w[8];  // absorbed, lost, esc_top, esc_bot, esc_mesh, specular, roulette, capped
};
enum { CW_ABSORBED = 0, CW_LOST, CW_ESC_TOP, CW_ESC_BOT, CW_ESC_MESH, CW_SPECULAR, CW_ROULETTE, CW_CAPPED };

constexpr int kMaxMedia = 32;
constexpr uint32_t kLogChunk = 8192;    // records per log chunk = records per partition work item
constexpr uint32_t kTileShift = 14;     // grid tile = 32 x 32 x 16 voxels = 16384 (128 KiB of f64 in LDS)
constexpr uint32_t kTileSize = 1u << kTileShift;
// Deposit records carry a TILED voxel index: (tile id << 14) | (z&15)<<10 | (y&31)<<5 | (x&31), tile id =
// (tz * nty + ty) * ntx + tx.  3-D blocks follow the compact photon cloud, so far fewer tiles are active than with
// slabs of consecutive linear indices and the partition pass writes longer contiguous runs.
constexpr uint32_t kTileBX = 5, kTileBY = 5, kTileBZ = 4;
constexpr int kMaxLayers = 64;

struct WalkParams {
    // photon queue
    unsigned long long* head;
    unsigned long long n_photons, photon_offset, seed;
    // scene tables (device pointers, element type depends on walk precision)
    const void* media;
    const void* zb;
    const int32_t* layer_medium;
    const void* tris;
    const void* nodes;
    int n_media, n_layers, n_tris, n_nodes;
    double n_above, n_below;
    // voxel grid
    void* grid;
    int nx, ny, nz, tally;
    double origin[3], inv_voxel[3];
    // source
    int src_type, start_medium;
    double src_pos[3], src_dir[3], src_e1[3], src_e2[3];
    // table RNG
    const double* table;
    unsigned long long table_steps;
    unsigned max_steps;
    DevCounters* counters;
    // log-structured tally (null = deposit with global atomics): coalesced deposit log written by the walk,
    // reduced into the grid by the partition / tile-reduce kernels
    uint32_t* log_idx;    // [log_cap_chunks * kLogChunk] voxel index
    void* log_val;        // [log_cap_chunks * kLogChunk] value in the tally's type
    uint32_t* log_fill;   // [log_cap_chunks] valid records per chunk
    uint32_t* log_next;   // next free chunk
    uint32_t log_cap_chunks;
    uint32_t* log_hist;   // [log_n_tiles] records per grid tile, accumulated by the walk (LDS histogram per workgroup)
    uint32_t log_n_tiles, log_ntx, log_nty;
    // clearance grid (mesh scenes; null = off): conservative lower bound of the distance from any point of a cell
    // to any triangle -- a hop shorter than that cannot hit, so the BVH query is skipped
    const float* clear;
    int cnx, cny, cnz;
    double corg[3], cinv[3];
    // light sub-path capture (null = off)
    lt_vertex* vertices;
    uint32_t* vertex_counts;
    unsigned max_vertices;
};

struct LaunchCfg {
    int blocks, threads;
    size_t lds_bytes;
};

// walk variants: precision x geometry x rng are compile-time, tally is too
struct Variant {
    int f32;     // 0: f64 walk, 1: f32 walk
    int mesh;    // 0: layered slab, 1: mesh + BVH staged in LDS, 2: mesh + BVH read from global memory
    int table;   // 0: XORWOW, 1: table RNG
    int tally;   // LT_TALLY_*
};

hipError_t launch_walk(const WalkParams& P, const Variant& v, const LaunchCfg& cfg, hipStream_t s);
size_t walk_lds_bytes(const Variant& v, int n_media, int n_layers, int n_tris, int n_nodes, unsigned n_hist = 0);
// resident-blocks-per-CU the runtime reports for a variant at `threads`
int walk_max_blocks_per_cu(const Variant& v, int threads, size_t lds_bytes);

struct RenderParams {
    const void* tris;    // TriD<double>
    const void* nodes;   // NodeD<double>
    const lt_surface_material* mats;
    const lt_point_light* lights;
    int n_tris, n_nodes, n_lights;
    int W, H, S, D;
    double cam[3], f_distance;
    const double* xs;
    const double* ys;
    double* rand_0;
    const double* rand_1;
    const int32_t* light_choice;
    double* image;
    int variant;   // 0 = path_tracing_fix1.trace_path, 1 = path_tracing_old.trace_path
    int choices;   // variant 1: light_choice entries per sample
};
constexpr int kRenderOldMaxDepth = 24;   // frames of the unrolled recursion (variant 1)
hipError_t launch_build_clearance(const void* tris_f64, int n_tris, float* clear, int nx, int ny, int nz,
                                  const double org[3], const double cell[3], hipStream_t s);
hipError_t launch_render_surface(const RenderParams& P, hipStream_t s);

// log-structured tally pipeline (all on stream s)
struct LogReduceParams {
    const uint32_t* log_idx; const void* log_val; const uint32_t* log_fill; uint32_t n_chunks;   // walk output
    uint32_t* tmp_idx; void* tmp_val;          // ping-pong buffers, same capacity as the log
    uint32_t* hist;                            // [n_tiles]
    uint32_t* tile_base;                       // [n_tiles + 1]
    uint32_t* cursor1;                         // [nb1]
    uint32_t* cursor2;                         // [n_tiles]
    uint32_t* items2;                          // [nb1 + 1] prefix of pass-2 work items per level-1 bin
    uint32_t* items_r;                         // [n_tiles + 1] prefix of reduce work items per tile
    uint32_t* totals;                          // [3]: total records, pass-2 items, reduce items
    const uint32_t* chunks_used;               // [1]: chunks the walk claimed (may exceed cap_chunks on overflow)
    uint32_t cap_chunks;
    uint32_t* work;                            // [3]: work-item counters of part1, part2, reduce (zeroed per batch)
    uint32_t n_tiles, bits2;                   // level-2 digit width; level-1 bins = ceil(n_tiles >> bits2)
    void* grid; size_t n_vox; int tally;
    uint32_t nx, ny, nz, ntx, nty;             // grid shape and tile counts along x, y (tiled record index)
};
hipError_t launch_log_hist(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_scan(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_part1(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_part2(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_reduce(const LogReduceParams& L, hipStream_t s);

hipError_t launch_intersect_rays(const void* tris, const void* nodes, int n_tris, int n_nodes,
                                 const double* o, const double* d, const double* tmax, size_t n,
                                 int use_bvh, int32_t* prim, double* t, hipStream_t s);
hipError_t launch_triangle_intersect(const double* o, const double* d, const double* tris, size_t n,
                                     double* t, hipStream_t s);
hipError_t launch_intersect_bounds(const double* o, const double* d, const double* tmax,
                                   const double* boxes, size_t n, int32_t* hit, hipStream_t s);
hipError_t launch_eval(int fn, const double* in, size_t n, double* out, hipStream_t s);
hipError_t launch_rng_raw(unsigned long long seed, unsigned long long photon_id, unsigned count,
                          uint32_t* out, hipStream_t s);
hipError_t launch_grid_to_f64(const void* grid, int tally, size_t n, double* out, hipStream_t s);

}  // namespace ltk
