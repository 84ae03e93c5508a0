/*
 * lt.h -- C ABI of the MI355X photon-transport hot path (liblt_hip.so).
 *
 * The reference (zhouyifan233/light-transport) has no FFI: its only boundary is
 * the Python call  render_scene(scene, primitives, bvh) -> ndarray
 * (LightTransportSimulator/light_transport/src/path_tracing_fix1.py:139-169),
 * fed by constructor objects (primitives.py:99, material.py:28, scene.py:53,
 * bvh_new.py:11,60,148,282).  This header is the boundary a maintainer binds
 * with ctypes from the (empty) reference slot src/photon_tracing.py; see
 * INTEGRATION.md for the binding stub.  Each entry point names the reference
 * construct whose role it takes over.
 *
 * Conventions
 *   - every function returns 0 on success, a negative LT_E_* code on failure;
 *     lt_last_error(ctx) returns a message for the last failure on that ctx
 *     (ctx == NULL: the last failure of lt_create on this thread).
 *   - all host pointers stay owned by the caller; set_* functions copy.
 *   - one ctx = one GPU + one HIP stream.  A ctx is not thread-safe; distinct
 *     ctxs are independent.
 *   - there is NO CPU fallback: lt_create fails when no gfx950-class device /
 *     code object is usable.
 */
#ifndef LT_H_
#define LT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_ABI_VERSION 3

/* error codes */
#define LT_OK 0
#define LT_E_INVALID (-1)  /* bad argument / inconsistent scene            */
#define LT_E_STATE (-2)    /* call order: e.g. launch before set_grid      */
#define LT_E_HIP (-3)      /* HIP runtime error (message has the detail)   */
#define LT_E_NOMEM (-4)
#define LT_E_UNSUPPORTED (-5)

/* tally storage of the voxel grid (lt_set_grid) */
#define LT_TALLY_F32 0   /* global_atomic_add_f32                         */
#define LT_TALLY_F64 1   /* global_atomic_add_f64                         */
#define LT_TALLY_U64FX 2 /* u64 fixed point, 2^40 units per unit weight:
                            order-independent, bit-reproducible.  One voxel
                            holds at most 2^24 = 1.68e7 units of weight
                            before it wraps (C2's hottest voxel absorbs
                            0.0074 per photon: ~2e9 photons per grid);
                            read the grid out and zero it before that      */
#define LT_FX_SCALE 1099511627776.0 /* 2^40 */

/* what an interaction adds to its voxel (lt_set_tally_quantity) */
#define LT_QUANTITY_ABSORBED 0 /* the absorbed weight dw = w mu_a / mu_t (default)           */
#define LT_QUANTITY_FLUENCE 1  /* w / mu_t (= dw / mu_a, defined for mu_a = 0 too): the grid then holds
                                  fluence x voxel volume x photons, correct in heterogeneous media -- no division
                                  by a per-voxel mu_a afterwards.  Role of the normalised tally image / samples,
                                  path_tracing_fix1.py:162-166.  Counters keep booking the absorbed weight.    */

/* photon sources (lt_set_source) */
#define LT_SRC_PENCIL 0      /* pos, dir                                   */
#define LT_SRC_COSINE_QUAD 1 /* quad corner=pos, normal=dir,
                                extra[0..2]=edge1, extra[3..5]=edge2;
                                cosine-weighted about dir (reference
                                utils.py:132-161, light_samples.py:64-116) */

/* lt_launch flags */
#define LT_FLAG_F32_WALK 1u /* walk arithmetic in f32 (default: f64, the
                               reference's dtype, e.g. scene.py:31-50)     */

/* optical medium -- takes the role of Material.ior (material.py:22) plus the
 * volumetric coefficients the reference lacks. */
typedef struct lt_medium {
    double mu_a; /* absorption coefficient  [1/length] */
    double mu_s; /* scattering coefficient  [1/length] */
    double g;    /* Henyey-Greenstein anisotropy (medium_samples.py:14-16) */
    double n;    /* refractive index */
} lt_medium;

/* flattened BVH node, DFS pre-order -- role of LinearBVHNode
 * (bvh_new.py:60-67) with second_child_offset = FIRST index of the right
 * subtree (the reference's own "#TODO: fix this", bvh_new.py:295). */
typedef struct lt_bvh_node {
    double lo[3];
    double hi[3];
    int32_t offset;  /* leaf: primitives_offset; interior: second child   */
    int32_t n_prims; /* >0: leaf                                           */
    int32_t axis;    /* interior: split axis 0..2                          */
    int32_t pad_;
} lt_bvh_node;

/* energy bookkeeping.  Invariant:
 *   photons = w_absorbed + w_lost_outside_grid + w_escaped_top
 *           + w_escaped_bottom + w_escaped_mesh + w_specular
 *           + w_roulette_net + w_capped                                  */
typedef struct lt_counters {
    uint64_t photons;
    uint64_t steps; /* photon-steps: iterations of the hop/drop/spin loop */
    double w_absorbed;
    double w_lost_outside_grid;
    double w_escaped_top;
    double w_escaped_bottom;
    double w_escaped_mesh;
    double w_specular;
    double w_roulette_net;
    double w_capped;
} lt_counters;

typedef struct lt_ctx lt_ctx;

/* ---- lifetime ---------------------------------------------------------- */
int lt_abi_version(void);
/* role of: constructing a Scene (scene.py:53-73) bound to one device */
int lt_create(lt_ctx** out, int device_id);
int lt_destroy(lt_ctx* ctx);
const char* lt_last_error(const lt_ctx* ctx);

/* ---- scene ------------------------------------------------------------- */
/* media table; index = medium id used by layers / triangles */
int lt_set_media(lt_ctx* ctx, const lt_medium* media, int n);
/* layered slab: n layers, z_bounds[n+1] ascending (last may be +inf),
 * ambient indices above z_bounds[0] / below z_bounds[n].  Clears any mesh. */
int lt_set_layers(lt_ctx* ctx, const double* z_bounds, const int32_t* medium_idx,
                  int n, double n_above, double n_below);
/* triangle mesh + flattened BVH.  verts [T][3][3] (role of
 * PreComputedTriangle.vertex_1..3, primitives.py:99-111, in BVH order =
 * ordered_prims of build_bvh, bvh_new.py:148).  med_front/med_back: medium id
 * on the +normal / -normal side, -1 = exterior (photon leaves the scene).
 * Clears any layers. */
int lt_set_mesh(lt_ctx* ctx, const double* verts, const int32_t* med_front,
                const int32_t* med_back, int n_tris, const lt_bvh_node* nodes,
                int n_nodes);
/* Host BVH build + flatten in one call -- role of build_bvh + flatten_bvh (bvh_new.py:148-300) over BoundedBox-es of the
 * triangles (bvh_new.py:11-15), with the reference's defects B1 (second_child_offset) and B2 (partition on a copy) fixed.
 * Pure host code: no ctx, no device.  verts [n_tris][3][3]; split_method 1 = midpoint (the reference's hard-wired choice,
 * :149), 0 = binned surface-area heuristic (its dormant branch, :198-258).  order_out [n_tris]: ordered_prims[i] is input
 * triangle order_out[i]; nodes_out [max_nodes >= 2 n_tris - 1], pre-order, as lt_set_mesh takes them.  The tree equals
 * the Python mirror's (light_transport_amd/src/bvh_new.py build_bvh / flatten_bvh) node for node. */
int lt_build_bvh(const double* verts, int n_tris, int split_method, int32_t* order_out, lt_bvh_node* nodes_out,
                 int max_nodes, int* n_nodes_out);
/* voxel grid (tally) -- role of Scene.image (scene.py:66).  C-order
 * [nz][ny][nx]; allocates and zeroes the device grid and the counters. */
int lt_set_grid(lt_ctx* ctx, int nx, int ny, int nz, const double origin[3],
                const double voxel[3], int tally_dtype);
/* source -- role of sample_light (light_samples.py:90-116).  start_medium:
 * medium id the photon starts in (mesh scenes; ignored for layers). */
int lt_set_source(lt_ctx* ctx, int type, const double pos[3], const double dir[3],
                  const double* extra, int start_medium);
int lt_set_max_steps(lt_ctx* ctx, uint32_t max_steps);
/* Experiment knobs -- tuning constants and diagnostic switches that tests and measurement tools vary; results do not
 * depend on any of them (the tests hold that down bit for bit).  value < 0 restores the built-in default.  Each knob is
 * seeded ONCE, in lt_create, from the environment variable LT_<KEY IN CAPITALS>; the library reads the environment
 * nowhere else.  Keys:
 *   query_min          meshes in LDS: lanes a wave gathers before it serves their surface queries; meshes beyond LDS:
 *                      answered-but-untested queries that trigger a drain of the candidate queue            (1..64)
 *   log_bits2, log_hot two-pass partition: width of the second digit; cap on the number of hot tiles (0: plain two-pass)
 *   overlap_walk_bpc   workgroups per CU of a walk that shares the device with another lane's reduction     (1..8)
 *   diag_no_tally      time the walk without deposition;   log_timing  print per-stage device times of every launch
 *   tail_split         slab walks in log mode: 0 = the whole walk in one kernel; default = split (a wave hands its last
 *                      photons to the tail kernel when at most 32 lanes are alive) where a drain is exposed and the batch
 *                      has >= 512 photons per launched wave; n >= 2 = split with threshold n (<= 48) whatever the size
 *   part_alone         0 / 1: pin the partition build that shares the CU with a walk / the one for an otherwise idle device
 *   part_lds           one-pass grids (<= 1024 tiles): the partition that stages its NEXT item by LDS-DMA while it sorts this one
 *                      -- one workgroup per CU, 136 KiB of LDS, <= 62 VGPRs, no second workgroup needed to hide its loads
 *                      (on an idle device it equals the default build within 5 %).  Bits: 1 = use it where it applies; 2 = and
 *                      make lt_launch fail where it does not (two-pass grids; tests); 4 = the ONE-WAVE-PER-SIMD builds of the
 *                      partition (256 lanes, items of 2048 records, 68 KiB) and of the tile reduce (256 lanes) -- what fits
 *                      beside four walk waves of <= 112 VGPRs (measurement; DESIGN "Overlap")
 *   serial_walks       1: WALK TRAIN -- the walk kernels of every context of this device that sets the knob run one after
 *                      another (each waits for the end of the one enqueued before it; the log reductions stay on their own
 *                      streams).  For hosts that keep several jobs in flight: launch each walk at three of the four resident
 *                      workgroups per CU (lt_set_launch_config(3, 256)) and the free quarter of the register file carries the
 *                      reduction of the job in front (bench.py's walk_train regime)
 *   -- taking effect when the mesh tables are next built (lt_set_mesh + launch / query):
 *   march_cells, march_scale_milli   march grid: cells along the longest axis / cell size in 1/1000 of the default
 *   clearance_cells    clearance grid of meshes in LDS: cells along the longest axis
 *   no_march, no_clearance, no_near_lists   switch the shortcut off (every query then takes the slower exact path)
 *   force_march        meshes that fit LDS take the march grid and walk_kernel_m all the same (measurement)
 *   march_info         print the march grid's dimensions when it is built */
int lt_set_tuning(lt_ctx* ctx, const char* key, int64_t value);
/* LT_QUANTITY_*; applies to subsequent launches (the grid is NOT rescaled: zero it when switching) */
int lt_set_tally_quantity(lt_ctx* ctx, int quantity);
/* launch geometry: resident workgroups per CU and threads per workgroup
 * (multiples of 64).  0 keeps the default. */
int lt_set_launch_config(lt_ctx* ctx, int blocks_per_cu, int threads_per_block);

/* how deposits reach the grid.  LT_MODE_ATOMIC: one no-return global atomic per
 * deposit record.  LT_MODE_LOG: the walk appends records to a coalesced log in
 * HBM which is radix-partitioned by grid tile and reduced tile by tile in LDS
 * (no global atomics); photons are traced in batches sized to log_bytes, the
 * budget for all deposit logs of the ctx and their ping-pong copies (0 restores
 * the default: a quarter of the device memory, at most 64 GiB).  A log that turns
 * out too small diverts the excess records to atomics -- speed, never
 * correctness (lt_last_log_info reports how many).  The first launch of a new
 * scene traces a 16384-photon pilot batch and reads its record rate back (a few
 * ms on the host) so that logs and batches are sized from a measurement.
 * Results are identical for the u64 fixed-point tally and equal up to summation
 * order for float tallies.  Table RNG, vertex capture and grids beyond 65536
 * tiles (2^30 voxels) use the atomic path. */
#define LT_MODE_ATOMIC 0
#define LT_MODE_LOG 1
#define LT_MODE_AUTO 2 /* default: LOG whenever the launch can use it (no RNG table, no vertex capture, grid within the tile index range, free HBM), ATOMIC otherwise */
int lt_set_tally_mode(lt_ctx* ctx, int mode, uint64_t log_bytes);
/* optional: allocate the deposit log(s) for launches of up to n_photons now (otherwise the first launch does it) */
int lt_reserve_log(lt_ctx* ctx, uint64_t n_photons);
/* Overlap inside ONE lt_launch (log mode).  lanes = 2: the launch is cut into
 * sub-batches that alternate between two streams of the ctx, each with its own
 * deposit log, so that the bandwidth-bound reduction of batch k runs beside the
 * VALU-bound walk of batch k+1 (role of Numba's thread pool working through one
 * render_scene call, path_tracing_fix1.py:139-148: the caller still makes one
 * call).  lanes = 1: one stream, batches back to back.  lanes = 0 (default):
 * launches of >= 2^21 photons whose geometry the caller has not pinned
 * (lt_set_launch_config) try both once and keep the faster.  The ctx stream
 * joins both lanes before lt_launch returns, so everything ordered on lt_stream
 * (readback, lt_reduce_grid, lt_zero_tally) stays ordered.  The u64 fixed-point
 * grid is bit-identical for every setting. */
int lt_set_overlap(lt_ctx* ctx, int lanes);

/* ---- run --------------------------------------------------------------- */
/* role of render_scene (path_tracing_fix1.py:139-169): trace photons
 * [photon_offset, photon_offset + n_photons) asynchronously on the ctx stream,
 * ACCUMULATING into the grid and counters.
 *   rng_table == NULL : rocRAND XORWOW, state re-seeded per photon from
 *                       (seed, photon id) -- results do not depend on launch
 *                       geometry or on which rank traces which id range.
 *   rng_table != NULL : "table RNG" of the reference (scene.py:68-69,
 *                       path_tracing_fix1.py:28-29): host array
 *                       [n_photons][table_steps][4] of uniforms in [0,1),
 *                       addressed by (photon - photon_offset, step). */
int lt_launch(lt_ctx* ctx, uint64_t n_photons, uint64_t photon_offset, uint64_t seed,
              const double* rng_table, uint64_t table_steps, uint32_t flags);
int lt_sync(lt_ctx* ctx);
/* milliseconds of device time spent in the transport kernel by the last
 * lt_launch (HIP events on the ctx stream); valid after lt_sync. */
int lt_last_kernel_ms(lt_ctx* ctx, double* ms);
int lt_zero_tally(lt_ctx* ctx); /* zero grid + counters (async)           */
/* device milliseconds per stage, summed over the batches (and lanes) of the last
 * log-mode lt_launch: [0] walk kernel, [1] scan, [2] partition pass(es), [3] tile
 * reduce (with two lanes the stages of different batches overlap in time);
 * deposit records and batches of that launch. */
int lt_last_log_stages(lt_ctx* ctx, double ms_out[4], uint64_t* records, uint64_t* batches);
/* bookkeeping of the last log-mode lt_launch (blocks until it has finished): records written to the
 * deposit log, records that found it full and went to the grid as atomics, batches, lanes used */
int lt_last_log_info(lt_ctx* ctx, uint64_t* records, uint64_t* overflow_records, uint64_t* batches, int* lanes);
/* grids of more than 1024 tiles (a tile = 32 x 32 x 16 voxels) take two partition passes; once a tile histogram of
 * the scene is known (the pilot batch), the tiles that hold most records ("hot") leave the first pass in their final
 * form and only the rest goes through the second.  Reports, for the last log-mode lt_launch (blocking): how many tiles
 * were hot (0: plain two-pass or one-pass form) and the record count of the pilot histogram that made a tile hot.
 * lt_set_tuning("log_hot", n) caps the number of hot tiles (0 switches the form off). */
int lt_last_log_hot_tiles(lt_ctx* ctx, uint32_t* hot_tiles, uint32_t* threshold);

/* ---- readback ---------------------------------------------------------- */
/* blocking D2H of the raw tally ([nz][ny][nx], dtype as set) */
int lt_read_grid(lt_ctx* ctx, void* host_out, size_t bytes);
/* blocking D2H, converted to float64 absorbed weight per voxel */
int lt_read_grid_f64(lt_ctx* ctx, double* host_out, size_t n_voxels);
int lt_read_counters(lt_ctx* ctx, lt_counters* out);
int lt_grid_device_ptr(lt_ctx* ctx, void** ptr, size_t* bytes);
int lt_counters_device_ptr(lt_ctx* ctx, void** ptr, size_t* bytes);
void* lt_stream(lt_ctx* ctx); /* hipStream_t of the ctx */
/* sum-reduce grid + counters over an RCCL communicator (ncclComm_t), on the
 * ctx stream; root < 0 = all-reduce.  librccl is loaded on first use. */
int lt_reduce_grid(lt_ctx* ctx, void* nccl_comm, int root);

/* ---- geometry / sampling queries on the device ------------------------- */
/* These run the SAME __device__ functions the walk uses, one lane per query,
 * so the reference's per-function behaviour can be checked in isolation. */

/* nearest hit of n rays against the ctx mesh -- role of hit_object /
 * intersect_bvh (utils.py:53-68, bvh_new.py:414-482), predicate
 * EPSILON < t < tmax.  prim_out = -1, t_out = +inf when nothing is hit.
 * use_bvh = 0: brute force over all triangles; 1: the flattened BVH; 2: the march grid as the walk uses it for meshes beyond
 * LDS (wave-cooperative: lanes march their rays through the grid, the whole wave tests the candidates; built on request for
 * smaller meshes); 3: the same march lane by lane; 4: the BVH front to back -- the child on the ray's side of the split plane
 * first, the reference's order (bvh_new.py:455-458), threaded per direction sign pattern so that it needs no stack: what the
 * surface renderers use.  All five give the same answer bit for bit. */
int lt_intersect_rays(lt_ctx* ctx, const double* origins, const double* dirs,
                      const double* tmax, size_t n, int use_bvh, int32_t* prim_out,
                      double* t_out);
/* pairwise Moller-Trumbore -- role of triangle_intersect
 * (intersects.py:46-104).  tris [n][3][3]; t_out = NaN for "None". */
int lt_triangle_intersect(lt_ctx* ctx, const double* origins, const double* dirs,
                          const double* tris, size_t n, double* t_out);
/* pairwise slab test -- role of intersect_bounds (intersects.py:179-196).
 * boxes [n][2][3] (min,max); hit_out 0/1. */
int lt_intersect_bounds(lt_ctx* ctx, const double* origins, const double* dirs,
                        const double* tmax, const double* boxes, size_t n,
                        int32_t* hit_out);
/* sampling helpers, n independent evaluations each:
 *  LT_FN_HG_PDF       in: cos_theta, g            out: [1] henyey_greenstein (medium_samples.py:14-16)
 *  LT_FN_HG_SAMPLE    in: xi, g                   out: [1] deflection cosine (Appendix C.6)
 *  LT_FN_ONB          in: n[3]                    out: [6] v2,v3 (utils.py:72-80)
 *  LT_FN_DISK         in: u[2]                    out: [2] (utils.py:115-128)
 *  LT_FN_COSINE_HEMI  in: n[3], wi[3], u[2]       out: [4] dir, pdf (utils.py:132-161)
 *  LT_FN_REFLECT      in: v[3], n[3]              out: [3] (brdf.py:8-9)
 *  LT_FN_BOUNDARY     in: d[3], n[3], n1, n2      out: [5] R_fresnel, cos_t, refracted[3] (Appendix C.5)
 *  LT_FN_SPIN         in: u[3], cos_t, phi_xi     out: [3] rotated direction (Appendix C.6)
 * in/out are row-major [n][k]. */
#define LT_FN_HG_PDF 0
#define LT_FN_HG_SAMPLE 1
#define LT_FN_ONB 2
#define LT_FN_DISK 3
#define LT_FN_COSINE_HEMI 4
#define LT_FN_REFLECT 5
#define LT_FN_BOUNDARY 6
#define LT_FN_SPIN 7
#define LT_FN_WALK_MATH 8 /* in: x in (0,1]  out: [5] -ln x, sin 2 pi x, cos 2 pi x, sqrt x, 1/(1+x) -- the walk's f64 primitives */
#define LT_FN_WALK_MATH_RAW 9 /* in: k = a raw 32-bit draw (as a double)  out: [3] -ln u, sin 2 pi u, cos 2 pi u of u = (k + 1) 2^-32, by the forms the f64 XORWOW walk uses on the raw draw */
int lt_eval(lt_ctx* ctx, int fn, const double* in, size_t n, double* out);
/* first `count` raw 32-bit XORWOW outputs of photon `photon_id` under `seed`
 * (checks the per-photon seeding against the oracle's restatement) */
int lt_rng_raw(lt_ctx* ctx, uint64_t seed, uint64_t photon_id, uint32_t count,
               uint32_t* out);

/* ---- surface path tracing on the same mesh ------------ */
/* per-triangle surface record: the Material fields trace_path reads
 * (material.py:28-37; path_tracing_fix1.py:45-119) + PreComputedTriangle.is_light */
typedef struct lt_surface_material {
    double diffuse[3];   /* material.color.diffuse */
    double emission;
    double ior;
    double transmission;
    int32_t is_diffuse, is_mirror, is_light, pad_;
} lt_surface_material;

/* one point sample of the area light: role of Light (scene.py:12-17).
 * radiance = material.emission * material.color.diffuse (light_samples.py:56) */
typedef struct lt_point_light {
    double source[3];
    double normal[3];
    double radiance[3];
    double total_area;
} lt_point_light;

/* one entry per triangle of the current mesh (lt_set_mesh order) */
int lt_set_surface_materials(lt_ctx* ctx, const lt_surface_material* mats, int n_tris);
int lt_set_lights(lt_ctx* ctx, const lt_point_light* lights, int n);
/* role of render_scene + trace_path (path_tracing_fix1.py:18-169).  One lane
 * per pixel.  xs[width] / ys[height]: screen coordinates (np.linspace of
 * left..right / top..bottom, :141-142).  rand_0 / rand_1: the Scene tables
 * [H][W][S][D] (scene.py:68-69); rand_0 is updated in place with the +inf
 * markers the reference writes for unused bounces (:38,66,130).  light_choice
 * [H][W][S][D]: index of the light sample used by the shadow ray of that bounce
 * (the reference draws it with np.random.choice, light_samples.py:38).
 * image [H][W][3] is ACCUMULATED into: += 0.25 * clip(mean colour) (:166). */
int lt_render_surface(lt_ctx* ctx, int width, int height, int samples, int max_depth,
                      const double camera[3], double f_distance, const double* xs,
                      const double* ys, double* rand_0, const double* rand_1,
                      const int32_t* light_choice, double* image);
/* role of render_scene + trace_path of path_tracing_old.py:17-171, the recursive
 * integrator examples/LTS.ipynb calls: a diffuse hit recurses on the shared ray
 * and carries on from where the callee left it (:68-80), emission counts at
 * bounce 0 only (:45), roulette starts after bounce 3 (:127).  Same tables as
 * lt_render_surface, except that one path casts up to 2^max_depth - 1 shadow
 * rays: light_choice is [H][W][S][choices_per_sample] and the k-th shadow ray of
 * a sample (depth-first order) uses entry k mod choices_per_sample.
 * max_depth <= 24.  image [H][W][3] is OVERWRITTEN with clip(mean colour) (:167). */
int lt_render_surface_old(lt_ctx* ctx, int width, int height, int samples, int max_depth,
                          const double camera[3], double f_distance, const double* xs,
                          const double* ys, double* rand_0, const double* rand_1,
                          const int32_t* light_choice, int choices_per_sample, double* image);

/* ---- light sub-path vertices ("photon map" output) ----- */
/* Role of the Vertex record and of generate_light_subpaths / random_walk
 * (vertex.py:24-37, bdpt.py:18-147,258-268; both unrunnable in the reference):
 * besides depositing into the grid, the walk stores the first
 * max_vertices_per_photon vertices of every photon's path. */
#define LT_VERTEX_LIGHT 5        /* emission point (constants.py:22, Medium.LIGHT)     */
#define LT_VERTEX_REFLECTIVE 3   /* boundary event, reflected (Medium.REFLECTIVE)      */
#define LT_VERTEX_TRANSMISSIVE 4 /* boundary event, refracted / escaped                */
#define LT_VERTEX_VOLUME 7       /* interaction site in a medium (no reference value)  */
typedef struct lt_vertex {
    double point[3];     /* Vertex.point                                              */
    double direction[3]; /* direction of travel on arrival                            */
    double g_norm[3];    /* Vertex.g_norm: the light's normal (light_samples.py:104) / the interface normal facing
                            the arriving photon; zero inside a medium (bdpt.py:275: "on a surface if non-zero") */
    double throughput;   /* photon weight on arrival (Vertex.throughput, scalar)      */
    double pdf_pos;      /* emission vertex: 1 / light area (light_samples.py:100); pencil beam: 1; else 0 */
    double pdf_dir;      /* density of the direction the path LEAVES this vertex in: emission -- |cos| / pi
                            (light_samples.py:83), pencil beam: 1; medium -- the Henyey-Greenstein value
                            (medium_samples.py:14-16) of the sampled deflection; interface (a delta event) -- the
                            probability of the branch taken, R or 1 - R; 0 when the path ends here.  A path's
                            pdf_fwd[k] = pdf_dir[k-1] and pdf_rev[k] = pdf_dir[k+1] (bdpt.py:25-27,137; both
                            distributions are symmetric), in solid-angle measure                           */
    int32_t kind;        /* LT_VERTEX_* (Vertex.medium)                               */
    int32_t medium;      /* layer index (slabs) / medium id (meshes) on arrival       */
    uint32_t step;       /* photon-step index of the event (0 = emission)             */
    uint32_t pad_;
} lt_vertex;
/* 0 disables capture (default).  Applies to subsequent lt_launch calls. */
int lt_set_vertex_capture(lt_ctx* ctx, uint32_t max_vertices_per_photon);
/* vertices [n_photons][max_vertices_per_photon] and counts [n_photons] of the
 * LAST lt_launch (photon index = id - photon_offset); blocking D2H. */
int lt_read_vertices(lt_ctx* ctx, lt_vertex* vertices_out, uint32_t* counts_out, uint64_t n_photons);

/* What accelerates surface queries on the current mesh (built at the first launch / query after lt_set_mesh; this call
 * builds it if need be).  kind: 0 none (BVH only), 1 clearance grid with near-triangle lists (meshes whose tables fit
 * LDS), 2 march grid (meshes beyond LDS), 3 both (the f32 tables fit LDS, the f64 ones do not).  march_dims /
 * clearance_dims: cells along x, y, z (zeros if absent); march_entries: (cell, triangle) pairs in the candidate lists. */
int lt_mesh_accel_info(lt_ctx* ctx, int* kind, int march_dims[3], uint64_t* march_entries, int clearance_dims[3]);

/* device description for bench reports */
int lt_device_info(lt_ctx* ctx, char* name, size_t name_len, int* n_cus,
                   int* clock_mhz, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LT_H_ */
