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
    R dep;   // what a voxel receives per unit of photon weight at an interaction: absorb, or inv_mu_t (LT_QUANTITY_FLUENCE)
    R one_m_g, two_g;   // the walk's Henyey-Greenstein denominator: (1 - g) + (2 g) xi as one fused product
    R pad_;
};

// Plane interface k of a layered slab (between layer k - 1 above and layer k below; k = 0 / n_layers: the ambient media):
// the refractive indices on both sides and their two quotients, computed once on the host (one IEEE division each, in walk
// precision) instead of by every photon that reaches the plane -- the Fresnel block of S/path_tracing_fix1.py:86-119
// runs whenever ANY lane of a wave sits on an interface, which in a layered medium is most wave-steps.
template <typename R>
struct IfD {
    R n_up, n_dn;        // index above / below the plane
    R nr_down, nr_up;    // n1 / n2 for a photon travelling down (n_up / n_dn) and up (n_dn / n_up)
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
// The first partition pass claims output space with one returning global atomic per (work item, non-empty digit).
// Records cluster in a few dozen tiles, so EVERY work item hits the same hot cursors, and one address sustains only
// ~90 atomics per microsecond (MI355X_MICROARCH "dequeue" / "fanin"); and a digit's region is written in fragments of
// tens of records, which only become whole 128-byte lines if neighbouring fragments meet in ONE XCD's L2.  Both are
// answered by groups: walk workgroup b belongs to group b % kLogGroups, takes its chunks from the group's own
// counter (chunk index = group + kLogGroups * n), and counts its records in the group's histogram column; partition
// workgroup b handles the chunks c = b (mod grid), i.e. one group's, and workgroups b and b + 8 share an XCD (observed
// dispatch order -- speed only, never correctness).  A digit's region is the groups' sub-regions back to back, each
// with its own cursor: 1/16 of the atomics per address, and consecutive fragments of a sub-region come from the same
// XCD shortly after one another.  Pass 2 of the two-pass form does the same with kLogGroups2 groups of work items.
constexpr uint32_t kLogGroups = 16;
constexpr uint32_t kLogGroups2 = 8;
constexpr int kMaxLayers = 64;

// March grid (meshes whose tables do not fit LDS, GEOM 2): a uniform grid over the root bounds of the BVH.  Every cell has
// an 8-byte record -- x: c0 (f32 bits, rounded down to a multiple of 64 ulp), a strictly conservative lower bound of the
// distance from ANY point of the cell to ANY triangle (a hop shorter than that cannot hit: no query at all), with the
// number of the cell's candidates in the 6 low bits; y: where they start in `list` -- and the candidates are every
// triangle that overlaps the cell grown by a margin (exact triangle / box separating-axis test on the host).  A query marches its hop segment through the
// cells front to back (3-D DDA; free space is crossed c0 at a time) and tests the candidates of the cells it visits with the
// walk's own tri_hit: same predicate, same nearest / tie rule, hence the same answer as a brute-force scan -- without a
// per-lane pointer chase through a tree.  Role of intersect_bvh's near-child-first order (S/bvh_new.py:455-458).
constexpr uint32_t kMarchCountMask = 63u;       // low bits of a cell record's x word: candidates of the cell (63: more than 62 -> that query walks the BVH)
struct MarchGrid {
    const uint2* cell;          // [nz][ny][nx]
    const uint32_t* list;
    int nx, ny, nz;
    double org[3], inv[3], h[3];   // origin, 1 / cell size, cell size
    // the dimensions as reals and everything again in f32: a kernel reads its loop invariants in walk precision straight
    // from its arguments (scalar registers) -- converted inside the kernel they are vector registers held through the hot loop
    double fn[3];
    float org32[3], inv32[3], h32[3], fn32[3];
    double nudge64, nudge32;       // by how much the f64 / f32 march steps past a cell wall (<< the margin of the lists)
};

// A photon in flight, as walk_kernel<..., PHASE 1> hands it over and walk_kernel<..., PHASE 2> takes it up (see "tail split"
// in lt_walk_kernel.inc): everything the walk keeps per lane -- position, direction, weight, unused optical depth, layer,
// step count, id, and the XORWOW state (rocrand_state_xorwow without the Box-Muller cache: d + x[5]).
template <typename R>
struct SurvD {
    R p[3], u[3], w, sleft;
    unsigned long long pid;
    int32_t cur;
    uint32_t step;
    uint32_t rng[6];
    uint32_t held[3];        // mesh walks: the decision uniforms of a prepared step that waits for its surface query (HeldU bits)
    uint32_t q_pend;         //             ... and whether there is one
};
constexpr uint32_t kDumpMaxLanes = 32;     // a wave hands its photons over when the queue is empty and at most this many are alive (default)
constexpr uint32_t kDumpPoolLanes = 48;    // what the pool is sized for per wave (the knob tail_split = n sets the threshold, at most this)

struct WalkParams {
    // photon queue
    unsigned long long* head;
    unsigned long long n_photons, photon_offset, seed;
    // scene tables (device pointers, element type depends on walk precision)
    const void* media;
    const void* zb;
    const void* ifaces;         // IfD<R>[n_layers + 1]
    const int32_t* layer_medium;
    const void* tris;
    const void* nodes;
    const int16_t* links;       // [8][2][n_nodes] front-to-back threading of the BVH (bvh_octant_links); null beyond 32767 nodes
    int n_media, n_layers, n_tris, n_nodes;
    double n_above, n_below;
    // voxel grid
    void* grid;
    int nx, ny, nz, tally;
    double origin[3], inv_voxel[3];
    double fdim[3];       // (double)nx, ny, nz -- and below everything the hot loop compares positions with, again in f32: read in
                          // walk precision from the kernel arguments these are scalar operands; an int -> real or f64 -> f32
                          // conversion inside the kernel is hoisted in front of the loop and kept in a vector register
    double cdim[3];       // clearance grid dimensions as reals
    struct { float origin[3], inv_voxel[3], fdim[3], corg[3], cinv[3], cdim[3]; } f32;
    // source
    int src_type, start_medium;
    double src_pos[3], src_dir[3], src_e1[3], src_e2[3];
    // table RNG
    const double* table;
    unsigned long long table_steps;
    unsigned max_steps;
    unsigned query_min;   // mesh walks: lanes of a wave that wait for a surface query before it is served (0: the built-in default)
    DevCounters* counters;
    // log-structured tally (null = deposit with global atomics): coalesced deposit log written by the walk,
    // reduced into the grid by the partition / tile-reduce kernels
    uint32_t* log_idx;    // [log_cap_chunks * kLogChunk] voxel index
    void* log_val;        // [log_cap_chunks * kLogChunk] value in the tally's type
    uint32_t* log_fill;   // [log_cap_chunks] valid records per chunk (zeroed per batch: chunk indices are claimed per group)
    uint32_t* log_next;   // [kLogGroups] chunks claimed per group; chunk index = group + kLogGroups * n
    uint32_t log_cap_chunks;
    uint32_t* log_hist;   // [log_n_hist][kLogGroups] records per partition bin and group, accumulated by the walk (LDS histogram per workgroup):
                          // bin = tile id >> (log_hist_shift - kTileShift): the tile itself for grids of <= 1024 tiles,
                          // the level-1 digit of the two-pass partition otherwise (<= 128 bins: 512 B of LDS)
    uint32_t log_n_hist, log_hist_shift, log_ntx, log_nty;
    uint32_t* log_overflow;   // [1] records that found the log full and went to the grid as global atomics
    // clearance grid (mesh scenes; null = off), one 16-byte record per cell:
    //   x  c0 (f32 bits): conservative lower bound of the distance from any point of the cell to any triangle -- a hop
    //      shorter than that cannot hit, so no query is made at all;
    //   y  c2 | c4 << 16 (f16, rounded down): the same bound for the 3rd / the 5th nearest triangle -- a hop shorter
    //      than c2 (c4) can only hit the 2 (4) nearest ones, which are tested directly instead of walking the BVH;
    //   z  id0 | id1 << 16,  w  id2 | id3 << 16: those triangles, nearest first (0xffff: none)
    const uint4* clear;
    int cnx, cny, cnz;
    double corg[3], cinv[3];
    MarchGrid mg;         // GEOM 2 (cell == null: off -> every hop walks the BVH)
    // tail split (slab walks in log mode): pool of photons handed from the bulk kernel to the tail kernel
    void* pool;           // SurvD<R>[pool_cap]
    uint32_t* pool_n;     // entries claimed (may exceed pool_cap: the excess was not written and those photons walked on)
    uint32_t pool_cap;
    uint32_t dump_max;    // PHASE 1: hand over when at most this many lanes of the wave are alive
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
    int mesh;    // 0: layered slab, 1: mesh + BVH staged in LDS, 2: mesh + BVH read from global memory, 3: the same with a march grid (walk_kernel_m)
    int table;   // 0: XORWOW, 1: table RNG
    int tally;   // LT_TALLY_*
    int capture; // 1: the build that stores light sub-path vertices (f64 walk, XORWOW)
    int phase;   // slab walks: 0 whole walk, 1 bulk (hands the last photons of every wave to a pool), 2 tail (walks the pool)
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
    const int16_t* links;   // [8][2][n_nodes] front-to-back threading of the BVH per direction sign pattern (nearest_bvh_ordered)
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
// march grid: exact point-to-mesh distance per cell by a pruned BVH nearest-point query, seeded from a coarse pass
// (`coarse`: scratch of ceil(n/8)^3 doubles); fills the x word of every cell record
hipError_t launch_march_clearance(const void* tris_f64, const void* nodes_f64, int n_nodes, const MarchGrid& G, uint2* cells,
                                  double* coarse, hipStream_t s);
hipError_t launch_build_clearance(const void* tris_f64, int n_tris, int near_lists, void* clear_records, int nx, int ny, int nz,
                                  const double org[3], const double cell[3], hipStream_t s);
hipError_t launch_render_surface(const RenderParams& P, hipStream_t s);

// Per-batch bookkeeping words of the log pipeline (one u32 array per lane, zeroed before every batch).
enum { LM_NEXT = 0,        // [kLogGroups] chunks the walk claimed per group (may run past the capacity when the log overflowed)
       LM_OVERFLOW = 16,   // records that went to the grid as atomics because the log was full
       LM_RECORDS,         // totals: records in the log
       LM_ITEMS2,          //         pass-2 work items (4096-record slices; statistics only)
       LM_ITEMS_R,         //         reduce work items
       LM_SLICE,           //         records per reduce work item (chosen by the scan from the record count)
       LM_ITEMS_C,         //         pass-2 work units = k_log_count2 work items (two-pass form)
       LM_WORK,            // work-item counter of the reduce (the other passes stride the grid over equal-sized items)
       LM_WORDS = 24 };
static_assert(kLogGroups == 16, "LM_NEXT holds one counter per group");
// tile / bin regions start at multiples of these record counts so that the wide (16-byte) loads of the next pass are
// aligned; the gaps are never read (a region's length comes from the histogram), the buffers carry the slack
constexpr uint32_t kTileAlign = 8, kBinAlign = 8;     // (a hot tile's bin is read by the reduce: 8 records per 16-byte load)

// log-structured tally pipeline (all on stream s)
struct LogReduceParams {
    const uint32_t* log_idx; const void* log_val; const uint32_t* log_fill;   // walk output
    uint32_t* tmp_idx; void* tmp_val;          // ping-pong buffers, same capacity as the log
    uint32_t* hist1;                           // [nb1][kLogGroups] records per level-1 bin and group (two-pass form: the walk's histogram)
    uint32_t* hist;                            // records per tile: one-pass form [n_tiles][kLogGroups], the walk's histogram;
                                               //                   two-pass form [n_tiles][kLogGroups2], counted from pass 1's output
    uint32_t* bin_base; uint32_t* bin_cnt;     // [nb1 + 1] / [nb1] where pass 1 puts each level-1 bin, records in it
    uint32_t* tile_base; uint32_t* tile_cnt;   // [n_tiles + 1] / [n_tiles] where the final pass puts each tile, records in it
    uint32_t* cursor1;                         // [nb1][kLogGroups] (one-pass form: [n_tiles][kLogGroups])
    uint32_t* cursor2;                         // [n_tiles][kLogGroups2]
    uint32_t* items2;                          // [nb1 + 1] prefix of pass-2 work items per level-1 bin
    uint32_t* items_c;                         // [nb1 + 1] prefix of tile-count work items per level-1 bin
    uint32_t* itab;                            // [pass-2 units][4] descriptor of every pass-2 work unit (k_log_items2)
    uint32_t* items_r;                         // [n_tiles + 1] prefix of reduce work items per tile
    uint32_t* meta;                            // [LM_WORDS]
    unsigned long long* job;                   // [2] records / overflowed records of the whole launch (all batches)
    uint32_t cap_chunks;
    uint32_t n_tiles, bits2;                   // level-2 digit width; level-1 bins = ceil(n_tiles >> bits2); 0 = one pass
    void* grid; size_t n_vox; int tally;
    uint32_t nx, ny, nz, ntx, nty;             // grid shape and tile counts along x, y (tiled record index)
    // Hot-tile form of the two-pass partition (null / 0: off).  dmap[tile] = digit of pass 1: the H tiles that held the
    // most records when the map was made ("hot", H = dmeta[0]) have digits 0 .. H-1 of their own and leave pass 1 in
    // their final form; every other tile shares digit H + (tile >> bits2) with its level-1 bin and goes through pass 2.
    // Any map gives the same grid; a good one (k_log_plan, from a measured tile histogram) saves the second pass for
    // most records.  The walk's level-1 histogram is not used in this form: k_log_count1 counts the digits from the log.
    const uint16_t* dmap; const uint32_t* dmeta;
    int alone;                                 // nothing shares the device with this lane's reduction (one lane, walk at full occupancy):
                                               // the partition may take the register budget that leaves no room for a co-running walk
    int flush_atomic;                          // every tile adds to the grid with atomics (another lane of the same
                                               // launch may be updating it at the same time)
    int lds_part;                              // one-pass grids: the LDS-staged partition (k_log_part_lds: one workgroup per CU, the next
                                               // item arrives by LDS-DMA while this one is sorted) instead of k_log_part
};
hipError_t launch_log_scan_bins(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_count1(const LogReduceParams& L, hipStream_t s);      // hot-tile form: pass-1 digit histogram from the log
// hot-tile map from a tile histogram: at most max_hot tiles, the ones with the most records (dmeta[0] = their number)
hipError_t launch_log_plan(const uint32_t* tile_cnt, uint32_t n_tiles, uint32_t bits2, uint32_t max_hot, uint16_t* dmap,
                           uint32_t* dmeta, hipStream_t s);
uint32_t log_max_digits();  // digits one partition pass can tell apart (1024)
constexpr uint32_t kMaxHotTiles = 8192;    // the map lives in LDS (2 bytes per tile: 16 KiB beside a partition item's 56 KiB, two workgroups per CU)
hipError_t launch_log_scan_tiles(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_part1(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_part2(const LogReduceParams& L, hipStream_t s);
hipError_t launch_log_reduce(const LogReduceParams& L, hipStream_t s);
uint32_t log_part_item();   // records per partition work item (the log capacity is a multiple of it)

// use_bvh: 0 brute force, 1 BVH, 2 / 3 march grid (G; falls back to the BVH for origins outside the grid), 4 BVH front to back (links)
hipError_t launch_intersect_rays(const void* tris, const void* nodes, int n_tris, int n_nodes,
                                 const double* o, const double* d, const double* tmax, size_t n,
                                 int use_bvh, const MarchGrid* G, const int16_t* links, int32_t* prim, double* t, hipStream_t s);
hipError_t launch_triangle_intersect(const double* o, const double* d, const double* tris, size_t n,
                                     double* t, hipStream_t s);
hipError_t launch_intersect_bounds(const double* o, const double* d, const double* tmax,
                                   const double* boxes, size_t n, int32_t* hit, hipStream_t s);
hipError_t launch_eval(int fn, const double* in, size_t n, double* out, hipStream_t s);
hipError_t launch_rng_raw(unsigned long long seed, unsigned long long photon_id, unsigned count,
                          uint32_t* out, hipStream_t s);
hipError_t launch_grid_to_f64(const void* grid, int tally, size_t n, double* out, hipStream_t s);
hipError_t launch_grid_add(void* dst, const void* src, int tally, size_t n, hipStream_t s);   // dst += src

}  // namespace ltk
