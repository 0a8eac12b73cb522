/*
 * rt_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the per-pixel Raycast hot path of erbuka/raytracing
 * (re::Raytracer behind re::AbstractRaycaster).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (raytracing_amd/) never links, loads or calls it.
 *
 * Every function in rt_oracle.c cites the reference file:line it follows
 * (paths relative to the reference checkout: Raytracer/..., Sandbox/...).
 *
 * Parity status: PINNED — bit-exact against the genuine reference compiled from
 * its own sources (oracle/_ref, see oracle/Makefile) on the scenes in
 * tests/golden/ (see tests/test_oracle_vs_ref.py, tests/test_oracle_golden.py).
 *
 * The "scene builder" entry points (sb_*) are the same C signatures that
 * oracle/ref_harness.cpp (the genuine reference) and the product host library
 * export, so one Python scene description drives all three.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sb_scene sb_scene;

/* ---- scene builder (common signature set) ---- */
sb_scene *sb_scene_new(void);
void sb_scene_free(sb_scene *s);
void sb_camera(sb_scene *s, const double pos[3], const double dir[3]);
int sb_mat_uniform(sb_scene *s, unsigned rgb, double absorptance);
int sb_mat_interp(sb_scene *s, int noise, int m0, int m1);
int sb_noise_checker(sb_scene *s, double domain);
/* table256 == NULL -> table is drawn with rand() % 256 like the reference ctor */
int sb_noise_perlin(sb_scene *s, double domain, const uint8_t *table256);
/* points == NULL -> drawn with rand() like the reference ctor; else div^3*3 doubles */
int sb_noise_worley(sb_scene *s, double domain, int divisions, const double *points);
int sb_noise_marble(sb_scene *s, double domain, double freq, double turb, const uint8_t *table256);
/* parent < 0 -> child of root.  returns node id (root is node 0) */
int sb_node(sb_scene *s, int parent, const double pos[3], const double rot[3], const double scale[3]);
int sb_sphere(sb_scene *s, int node, int mat);
int sb_plane(sb_scene *s, int node, int mat, const double n[3]);
/* verts / normals: ntris * 9 doubles (v0 v1 v2); normal_mode 0 = Face, 1 = Vertex */
int sb_mesh(sb_scene *s, int node, int mat, int normal_mode, int ntris, const double *verts,
            const double *normals);
/* type: 0 Directional, 1 Ambient, 2 Point (reference enum order, Scene.h:17) */
int sb_light(sb_scene *s, int type, unsigned rgb, const double dir[3], const double pos[3],
             double attenuation, int enabled);
void sb_bg_color(sb_scene *s, unsigned rgb);
void sb_bg_skybox(sb_scene *s, unsigned rgb0, unsigned rgb1, int sun_light, const uint8_t *table256);
/* read back noise tables (so fixtures do not depend on libc rand()) */
int sb_get_perlin_table(sb_scene *s, int noise, uint8_t *out256);   /* perlin or marble */
int sb_get_skybox_table(sb_scene *s, uint8_t *out256);
int sb_get_worley_points(sb_scene *s, int noise, double *out);       /* div^3*3 */

/* ---- rendering ---- */
typedef struct {
    int width, height;
    double fovy;          /* <= 0 -> default PI/4 (float PI literal, Common.h:9) */
    int max_recursion;
    int aa_mode;          /* 0 None, 1 SSAA */
    int num_threads;
    int renderer;         /* 0 Raytracer, 1 DebugRaycaster(Normal), 2 DebugRaycaster(Color) */
} sb_render_params;

/* pixels: W*H uint32 (index y*W+x).  shape_ord/tri_idx (nullable): primary-hit
 * pre-order ordinal of the hit shape node among shape nodes (-1 = miss) and the
 * index of the winning triangle in the mesh's triangle list (-1 if not a mesh).
 * ray_count (nullable): number of Scene::CastRay-equivalent queries.
 * sub_x/sub_y/n_sub: if n_sub > 0 only those pixels are rendered; outputs are
 * then n_sub long (sparse golden sets for the big configs). */
int sb_render(sb_scene *s, const sb_render_params *p, uint32_t *pixels, int32_t *shape_ord,
              int32_t *tri_idx, uint64_t *ray_count, const int32_t *sub_x, const int32_t *sub_y,
              int n_sub);

/* ---- known-answer probes (one call = n independent evaluations) ---- */
/* Scene::CastRay on n world rays (o,d: n*3).  out_hit n; out_point/local/normal n*3;
 * out_shape n (ordinal); out_tri n */
int sb_cast_rays(sb_scene *s, int n, const double *o, const double *d, int32_t *out_hit,
                 double *out_point, double *out_local, double *out_normal, int32_t *out_shape,
                 int32_t *out_tri);
/* Noise::Sample on n points */
int sb_noise_sample(sb_scene *s, int noise, int n, const double *p, double *out);
/* Material: out_color n*3, out_absorptance n */
int sb_material_eval(sb_scene *s, int mat, int n, const double *p, double *out_color,
                     double *out_absorptance);
/* Background::GetColor on n directions -> n*3 */
int sb_background_eval(sb_scene *s, int n, const double *d, double *out_color);
/* Raytracer::Raycast (RecursiveRaytrace from depth 0) on n rays -> n*3 */
int sb_raycast(sb_scene *s, int max_recursion, int n, const double *o, const double *d,
               double *out_color);
/* AbstractRaycaster::CreateScreenRay for n (x,y) real pixel coords -> o,d n*3 */
int sb_screen_rays(sb_scene *s, int width, int height, double fovy, int n, const double *xy,
                   double *out_o, double *out_d);
/* kd-tree dump of a mesh shape (pre-order).  Returns node count.  Arrays may be NULL
 * to query sizes.  node_depth/node_ntris/node_split: per node; tri_refs: concatenated
 * leaf triangle indices in pre-order (total returned in *n_refs). */
int sb_kd_dump(sb_scene *s, int mesh_shape_node, int32_t *node_depth, int32_t *node_ntris,
               double *node_split, double *node_bounds, int32_t *tri_refs, int64_t *n_refs);
/* compiled transforms of a node: 3*16 doubles (transform, inverse, normal) */
int sb_node_matrices(sb_scene *s, int node, double *out48);

const char *sb_impl_name(void);

#ifdef __cplusplus
}
#endif
#endif
