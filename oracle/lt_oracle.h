/*
 * lt_oracle.h -- CPU oracle for the photon-transport hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under light_transport_amd/ (the product)
 * may include, link, import or call this; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 *
 * What it restates, and how it is pinned:
 *   - the reference functions that exist (henyey_greenstein, triangle_intersect,
 *     intersect_bounds, create_orthonormal_system, concentric_sample_disk,
 *     cosine_weighted_hemisphere_sampling, get_reflected_direction, the BVH
 *     nearest-hit predicate, PreComputedTriangle fields) -- each function below
 *     cites the reference file:line it follows and is checked against golden
 *     vectors captured from the reference's own function bodies
 *     (the .npz files in tests/golden, made by tests/golden/make_golden.py).
 *   - the volumetric hop/drop/spin walk, which the reference does NOT contain
 *     (src/photon_tracing.py is an empty file; src/bdpt.py:40 is a TODO):
 *     PARITY UNPINNED against the reference.  It follows SURVEY.md Appendix C
 *     and is pinned by analytic known answers instead (energy conservation,
 *     Beer-Lambert, <cos theta> = g, <s> = 1/mu_t, MCML published Rd/Tt).
 */
#ifndef LT_ORACLE_H_
#define LT_ORACLE_H_

#include "../include/lt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lto_scene {
    int n_media;
    const lt_medium* media;
    /* layered slab (n_layers > 0) */
    int n_layers;
    const double* z_bounds;       /* [n_layers+1] */
    const int32_t* layer_medium;  /* [n_layers]   */
    double n_above, n_below;
    /* mesh (n_tris > 0), triangles in BVH order */
    int n_tris;
    const double* verts;          /* [n_tris][3][3] */
    const int32_t* med_front;
    const int32_t* med_back;
    int n_nodes;
    const lt_bvh_node* nodes;
    /* voxel grid */
    int nx, ny, nz;
    double origin[3], voxel[3];
    /* source */
    int src_type;
    double src_pos[3], src_dir[3], src_extra[6];
    int start_medium;
    uint32_t max_steps;
    int quantity;                 /* LT_QUANTITY_ABSORBED / LT_QUANTITY_FLUENCE (lt_set_tally_quantity) */
} lto_scene;

/* Walk photons [photon_offset, photon_offset+n).  grid_f64 (nullable) and
 * grid_fx (nullable, u64 fixed point, LT_FX_SCALE) are ACCUMULATED into;
 * counters are accumulated.  n_threads <= 1: single thread, plain adds,
 * deterministic.  walk_f32 != 0: the f32 restatement of the walk. */
int lto_run(const lto_scene* sc, uint64_t n_photons, uint64_t photon_offset,
            uint64_t seed, const double* rng_table, uint64_t table_steps,
            int walk_f32, double* grid_f64, uint64_t* grid_fx,
            lt_counters* counters, int n_threads);

int lto_run_capture(const lto_scene* sc, uint64_t n_photons, uint64_t photon_offset, uint64_t seed,
                    double* grid_f64, lt_counters* counters, lt_vertex* vertices, uint32_t* counts,
                    uint32_t max_vertices);

/* unit functions (double precision), same meaning as lt_eval / lt_* queries */
int lto_eval(int fn, const double* in, size_t n, double* out);
int lto_triangle_intersect(const double* origins, const double* dirs,
                           const double* tris, size_t n, double* t_out);
int lto_intersect_bounds(const double* origins, const double* dirs,
                         const double* tmax, const double* boxes, size_t n,
                         int32_t* hit_out);
int lto_intersect_rays(const lto_scene* sc, const double* origins, const double* dirs,
                       const double* tmax, size_t n, int use_bvh, int32_t* prim_out,
                       double* t_out);
int lto_rng_raw(uint64_t seed, uint64_t photon_id, uint32_t count, uint32_t* out);
/* PreComputedTriangle derived fields (primitives.py:99-112):
 * out[n][16] = centroid[3], edge_1[3], edge_2[3], normal[3], num, pad[3] */
int lto_triangle_fields(const double* tris, size_t n, double* out);

/* Surface path tracer: restates trace_path + render_scene + cast_one_shadow_ray
 * (S/path_tracing_fix1.py:18-169, S/light_samples.py:36-61) in double precision,
 * pixels in row-major order.  Nearest hits use the oracle's BVH (== brute force;
 * the reference traversal's bug B3 is not reproduced).  Pinned by fixture G8. */
int lto_render_surface(const lto_scene* sc, const lt_surface_material* mats, const lt_point_light* lights,
                       int n_lights, int width, int height, int samples, int max_depth, const double camera[3],
                       double f_distance, const double* xs, const double* ys, double* rand_0, const double* rand_1,
                       const int32_t* light_choice, double* image);
/* The recursive ancestor examples/LTS.ipynb calls (S/path_tracing_old.py:17-171), written as the recursion it is;
 * light_choice [H][W][S][choices_per_sample] is consumed in depth-first shadow-ray order; image is overwritten.
 * Pinned by fixture G9. */
int lto_render_surface_old(const lto_scene* sc, const lt_surface_material* mats, const lt_point_light* lights,
                           int n_lights, int width, int height, int samples, int max_depth, const double camera[3],
                           double f_distance, const double* xs, const double* ys, double* rand_0, const double* rand_1,
                           const int32_t* light_choice, int choices_per_sample, double* image);

#ifdef __cplusplus
}
#endif
#endif
