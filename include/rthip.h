/*
 * rthip.h — C-ABI of the MI355X (gfx950) Raycast hot path.
 *
 * This is the drop-in boundary below re::AbstractRaycaster (reference Raytracer/Raytracer.h:55-98):
 * a HIP-backed subclass (raytracing_amd/host: re::HipRaytracer) flattens the compiled scene into
 * the POD blobs below at render start and calls these entry points from its detached render
 * thread, exactly where the reference spawns its DoRaytraceThread workers
 * (Raytracer/Raytracer.cpp:39-87).  Plain C: pointers and sizes only, no C++/torch types.
 * Every call returns 0 on success and a negative rthip_status on failure;
 * rthip_last_error() returns the message of the calling thread's last failure.
 *
 * All reals are IEEE double in the reference's operation order (typedef double real,
 * Raytracer/Common.h:7); the kernels are built with -ffp-contract=off.
 */
#ifndef RTHIP_H
#define RTHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rthip_ctx rthip_ctx;

enum rthip_status {
    RTHIP_OK = 0,
    RTHIP_E_INVALID = -1,     /* bad argument / inconsistent scene description */
    RTHIP_E_HIP = -2,         /* a HIP runtime call failed (no device, OOM, launch failure) */
    RTHIP_E_UNSUPPORTED = -3, /* scene uses a feature outside the device path (limits below) */
    RTHIP_E_INTERRUPTED = -4, /* rthip_interrupt() was called during the render */
    RTHIP_E_STATE = -5        /* call order (render before upload, ...) */
};

enum { RTHIP_SHAPE_SPHERE = 0, RTHIP_SHAPE_PLANE = 1, RTHIP_SHAPE_MESH = 2 };
enum { RTHIP_LIGHT_DIRECTIONAL = 0, RTHIP_LIGHT_AMBIENT = 1, RTHIP_LIGHT_POINT = 2 }; /* Scene.h:17 */
enum { RTHIP_NOISE_CHECKER = 0, RTHIP_NOISE_PERLIN = 1, RTHIP_NOISE_WORLEY = 2, RTHIP_NOISE_MARBLE = 3 };
enum { RTHIP_BG_NONE = 0, RTHIP_BG_COLOR = 1, RTHIP_BG_SKYBOX = 2 };
enum { RTHIP_MAT_UNIFORM = 0, RTHIP_MAT_INTERP = 1 };
enum { RTHIP_RENDER_RAYTRACER = 0, RTHIP_RENDER_DEBUG_NORMAL = 1, RTHIP_RENDER_DEBUG_COLOR = 2 };

#define RTHIP_MAX_RECURSION 16   /* Raytracer::MaxRecursion upper bound on the device */
#define RTHIP_MAX_MAT_STACK 8    /* max nesting of InterpolatedMaterial trees */
#define RTHIP_KD_MAX_DEPTH 100   /* KDTreeTriangle depth cap, Scene.cpp:25 */

/* One scene-graph node that carries a Shape, in pre-order (Scene::CastRayRecursive visits
 * nodes in this order and the first node wins distance ties, Scene.cpp:217-233).
 * Matrices are the first three rows (row-major 3x4) of Transform's m_Transform,
 * m_InverseTransform and m_NormalTransform (Scene.cpp:261-263) as computed on the host. */
typedef struct {
    int32_t type;        /* RTHIP_SHAPE_* */
    int32_t material;    /* index into rthip_scene_desc.materials (root op of the shape's Material) */
    int32_t mesh;        /* index into meshes[] for RTHIP_SHAPE_MESH, else -1 */
    uint32_t shape_id;   /* Shape::GetID() (Scene.h:137), reported back in out_shape_id */
    double inv[12];      /* world -> local */
    double fwd[12];      /* local -> world */
    double nrm[12];      /* normal transform */
    double plane_n[4];   /* Plane::Normal (un-normalised, Scene.h:159); [3] unused */
    /* filled by rthip_upload_scene (callers leave them 0): bounds on E = fwd*inv - I, the round-trip
     * error of the two independently computed matrices (rotations come from float cosf/sinf,
     * Common.cpp:401-423): rt_lin >= ||E_3x3||_inf, rt_t >= |E translation|_inf.  They bound how far
     * the reference's self-intersection test (Scene.cpp:196-200) can reach along a shadow ray. */
    double rt_lin, rt_t;
    /* also filled by the upload: a world-space ball that contains the shape (spheres: sigma_max(fwd 3x3) around fwd's
     * translation; meshes: around the transformed centre of the BVH root boxes; planes: infinite), used for a
     * conservative early rejection of rays that pass nowhere near it.  bound_r2 = squared radius. */
    double bound_c[3];
    double bound_r2;
} rthip_shape;

/* kd-tree node, 16 bytes.  Nodes of one mesh are stored in pre-order; the left child of an
 * interior node i (if present) is i+1.  split = the plane BoundingBox::Split was called with
 * (Scene.cpp:56).  Interior: link = index of the right child (mesh-relative), info = axis |
 * HAS_LEFT | HAS_RIGHT.  Leaf: link = first entry in kd_refs (mesh-relative), info = LEAF |
 * (count << 8). */
typedef struct {
    double split;
    uint32_t link;
    uint32_t info;
} rthip_kdnode;
#define RTHIP_KD_AXIS_MASK 0x3u
#define RTHIP_KD_HAS_LEFT 0x4u
#define RTHIP_KD_HAS_RIGHT 0x8u
#define RTHIP_KD_LEAF 0x10u
#define RTHIP_KD_COUNT_SHIFT 8

/* Per-triangle record tested by Mesh::IntersectTriangle (Scene.cpp:421-467): the values
 * Triangle::Update caches (Scene.cpp:509-523).  128 bytes, 128-byte aligned in HBM. */
typedef struct {
    double v0[3];
    double face_normal[3];
    double e0[3];   /* v1 - v0 */
    double e1[3];   /* v2 - v0 */
    double d00, d01, d11, inv_den;
} rthip_tri;

/* Device-oriented bounding-volume hierarchy over the SAME triangles (no reference counterpart: the
 * reference only has the kd-tree above).  Mesh::Intersect returns the strictly closest triangle hit of
 * the whole mesh (argmin over triangles, Scene.cpp:469-493), so any conservative culling structure
 * yields the same result.  This one is 4-wide, 128 bytes per node = one 32-byte slot per child, so the
 * four lanes that trace one ray together each fetch and test one child (one 128-byte line per ray and
 * step).  Child boxes are fp32, rounded OUTWARD.  Child word: RTHIP_BVH_ABSENT = no child; bit 31 set =
 * leaf, bits 28..30 = triangle count - 1 (leaves hold 1..4 triangles, one per lane), bits 0..27 = first
 * triangle in bvh_tris (mesh-relative); else index of the child node (mesh-relative). */
typedef struct {
    struct {
        float lo[3], hi[3];
        uint32_t child;
        uint32_t pad;
    } c[4];
} rthip_bvhnode;
#define RTHIP_BVH_LEAF 0x80000000u
#define RTHIP_BVH_ABSENT 0xffffffffu
#define RTHIP_BVH_COUNT_SHIFT 28
#define RTHIP_BVH_FIRST_MASK 0x0fffffffu
#define RTHIP_BVH_MAX_LEAF 4
#define RTHIP_MAX_BATCH 64 /* frames per rthip_render_batch_device call */
#define RTHIP_BVH_MAX_DEPTH 48

typedef struct {
    uint32_t node_base, node_count;   /* into kd_nodes / kd_bounds */
    uint32_t ref_base, ref_count;     /* into kd_refs (values are mesh-relative triangle indices) */
    uint32_t tri_base, tri_count;     /* into tris / tri_normals */
    int32_t normal_mode;              /* 0 Face, 1 Vertex (Scene.h:18) */
    uint32_t max_depth;               /* deepest kd node, sizes the traversal stack */
    uint32_t bvh_node_base, bvh_node_count;   /* into bvh_nodes */
    uint32_t bvh_tri_base;            /* into bvh_tris / bvh_tri_index (tri_count entries, leaf order) */
    uint32_t bvh_depth;               /* deepest BVH node */
    float abs_max;                    /* max |coordinate| of the mesh's vertices (slab-test slack) */
    uint32_t pad;
    /* kd_planes[plane_base[a] .. +plane_count[a]) = sorted unique plane coordinates of the kd-tree's
     * boxes on axis a (root bounds + split values).  A ray with direction component a exactly 0 whose
     * origin coordinate a equals one of them makes the reference's slab test produce 0/0 = NaN
     * (Common.cpp:349-356) and take decisions no geometric structure reproduces: such rays are routed
     * through the kd-tree even when accel = 0. */
    uint32_t plane_base[3], plane_count[3];
} rthip_mesh;

/* Material trees in postfix order: UNIFORM pushes (color, absorptance); INTERP pops b then a and
 * pushes Lerp(a, b, noise.Sample(p)) (Material.cpp:21-29).  A shape's material index points at
 * the LAST op of its tree; first_op = index of the first op of that tree. */
typedef struct {
    int32_t op;          /* RTHIP_MAT_* */
    int32_t noise;       /* INTERP: index into noises */
    int32_t first_op;    /* first op of the subtree ending here */
    int32_t pad;
    double color[3];     /* UNIFORM: absorbed colour */
    double absorptance;  /* UNIFORM */
} rthip_matop;

typedef struct {
    int32_t type;          /* RTHIP_NOISE_* */
    int32_t table;         /* PERLIN/MARBLE: index into perlin_tables (units of 256 bytes) */
    int32_t points_base;   /* WORLEY: first point in worley_points (units of 3 doubles) */
    int32_t divisions;     /* WORLEY */
    uint32_t octaves;      /* PERLIN/MARBLE (Perlin.h:24), 1..9 (octave i samples at frequency 256 >> i, Perlin.cpp:31) */
    int32_t pad;
    double domain;         /* Noise::m_DomainSize */
    double persistance;    /* Perlin.h:23 */
    double step;           /* Worley.h:21 = 1.0 / divisions */
    double frequency;      /* Marble.h:16 */
    double turbolence;     /* Marble.h:16 */
} rthip_noise;

typedef struct {
    int32_t type;      /* RTHIP_LIGHT_* */
    int32_t enabled;
    double direction[3];
    double position[3];
    double color[3];
    double attenuation;
} rthip_light;

typedef struct {
    int32_t type;        /* RTHIP_BG_* */
    int32_t table;       /* SKYBOX: perlin table index of SkyBox::m_Perlin (domain 1) */
    int32_t sun_light;   /* SKYBOX: index into lights or -1 (m_Sun == nullptr) */
    int32_t pad;
    double color0[3];    /* COLOR: m_Color; SKYBOX: m_SkyColor0 */
    double color1[3];    /* SKYBOX: m_SkyColor1 */
} rthip_background;

/* The flattened scene.  All arrays are host memory owned by the caller; rthip_upload_scene
 * copies them to the device (and keeps nothing pointing into them). */
typedef struct {
    const rthip_shape *shapes;          uint32_t n_shapes;
    const rthip_mesh *meshes;           uint32_t n_meshes;
    const rthip_kdnode *kd_nodes;       uint64_t n_kd_nodes;
    const double *kd_bounds;            /* n_kd_nodes * 6: Min xyz, Max xyz (KDTreeTriangle::Bounds) */
    const uint32_t *kd_refs;            uint64_t n_kd_refs;
    const double *kd_planes;            uint64_t n_kd_planes;
    const rthip_tri *tris;              uint64_t n_tris;
    const double *tri_normals;          /* n_tris * 9: Triangle::Normals */
    const rthip_bvhnode *bvh_nodes;     uint64_t n_bvh_nodes;
    const rthip_tri *bvh_tris;          /* n_tris records in BVH leaf order */
    const uint32_t *bvh_tri_index;      /* n_tris: leaf-order position -> mesh-relative triangle index */
    const rthip_matop *materials;       uint32_t n_materials;
    const rthip_noise *noises;          uint32_t n_noises;
    const uint8_t *perlin_tables;       uint32_t n_perlin_tables;   /* n * 256 bytes */
    const double *worley_points;        uint64_t n_worley_points;   /* n * 3 doubles */
    const rthip_light *lights;          uint32_t n_lights;
    rthip_background background;
    /* only read by rthip_upload_scene_ex with RTHIP_UPLOAD_BUILD_KD (may be null otherwise): */
    const double *tri_verts;            /* n_tris * 9: Triangle::Vertices (three xyz triples) in the order of tris[] */
    const double *mesh_bounds;          /* n_meshes * 6: Mesh::m_BoundingBox Min xyz, Max xyz as Mesh::Compile leaves it (Scene.cpp:391-406) */
} rthip_scene_desc;

/* Camera basis of AbstractRaycaster::CreateScreenRay (Raytracer.cpp:102-121), computed once on
 * the host: forward/right/up normalised, h2 = atan(fovY/2), w2 = h2*W/H. */
typedef struct {
    double position[3];
    double forward[3];
    double right[3];
    double up[3];
    double h2, w2;
} rthip_camera;

typedef struct {
    uint32_t width, height;
    uint32_t max_recursion;   /* Raytracer::MaxRecursion (Raytracer.h:104) */
    int32_t aa_mode;          /* AbstractRaycaster::AAMode: 0 None, 1 SSAA (Raytracer.h:59-63) */
    int32_t renderer;         /* RTHIP_RENDER_* */
    /* screen partition for multi-GPU: this context renders the tiles t (row-major tile index,
     * tiles of tile_w x tile_h pixels) with t % part_count == part_index.  part_count = 1 -> whole frame. */
    uint32_t part_index, part_count;
    uint32_t tile_w, tile_h;  /* 0 -> default 32 x 8 */
    /* rthip_render_device only: 0 = d_pixels is the width*height frame (row-major);
     * 1 = d_pixels is this part's tile-compact buffer: rthip_part_pixels() uint32, tile after tile in
     * this part's tile order, 8x8 sub-blocks inside a tile (what the multi-GPU gather ships) */
    int32_t layout;
    /* mesh acceleration structure: 0 = BVH (default), 1 = the reference's kd-tree (same results;
     * kept for parity tests and as the literal flattening of KDTreeTriangle) */
    int32_t accel;
} rthip_params;

typedef struct {
    uint64_t rays_primary, rays_shadow, rays_reflect;  /* Scene::CastRay-equivalent queries of the last render */
    uint64_t pixels;                                   /* pixels rendered by this context */
    float ms_total;                                    /* device time of the last render (HIP events) */
    float ms_trace;                                    /* summed duration of the dominant kernel (RTHIP_PROFILE_LAUNCHES=1): the
                                                        * mesh-walk kernel - k_trace_quad / k_trace_coop, whichever the level's list
                                                        * took (both launches are inside the bracket) - or k_trace (accel = 1) */
    uint32_t launches_trace;                           /* number of launches of that kernel (= recursion levels traced) */
    uint32_t pad;
    /* only collected by contexts created with RTHIP_COUNTERS=1 in the environment.  The traversal counters count the structure
     * the frame WALKED: with accel = 1 the reference kd-tree (the names they were given in round 1), with accel = 0 - the
     * default - the 4-wide BVH; each has an alias that says so (same storage): */
    union { uint64_t kd_nodes_visited;  uint64_t bvh_steps; };          /* kd node records fetched (16 B each) / BVH node steps (one 128-byte node = 4 child boxes) */
    union { uint64_t kd_tris_tested;    uint64_t tris_tested; };        /* triangle records tested (128 B + 4 B reference each) */
    union { uint64_t kd_leaves_visited; uint64_t bvh_leaf_steps; };     /* leaves visited (BVH: leaf steps of <= 4 triangles) */
    union { uint64_t kd_pops;           uint64_t bvh_culled_pops; };    /* kd: pending subtrees resumed (48 B of bounds each) / BVH: stack entries popped already culled */
    uint64_t shape_tests;        /* shape-table entries tested (world->local transform fetch) */
    uint64_t mesh_hits;          /* mesh hits whose vertex normals were fetched (72 B) */
    union { uint64_t kd_max_nodes_ray; uint64_t max_steps_ray; };       /* most node records / steps any single ray (all its queries) took (tail-latency diagnostic) */
    /* divergence diagnostics: BVH interior steps / triangle-test iterations executed per WAVE (a wave
     * executes a step when any of its lanes does): the largest of any wave over the frame (summed over
     * the grid-stride iterations of that wave), and the totals */
    uint64_t wave_steps_max, wave_tris_max, wave_steps, wave_tris;
    uint64_t wave_cycles_max;    /* longest-lived trace wave of the frame, shader clock cycles */
    uint64_t fetch_cycles;       /* cycles waves waited for BVH node fetches (RTHIP_COUNTERS=2: of the longest wave) */
    /* per recursion level of the last chunk: rays in the level's queue, and (RTHIP_PROFILE_LAUNCHES=1)
     * the duration of the level's dominant trace kernel, of the kernels after it (redo + shade of the
     * listed rays) and of k_trace_easy (which also shades the rays it finishes) */
    uint32_t level_rays[RTHIP_MAX_RECURSION + 1];
    float level_ms_trace[RTHIP_MAX_RECURSION + 1];
    float level_ms_shade[RTHIP_MAX_RECURSION + 1];
    float level_ms_easy[RTHIP_MAX_RECURSION + 1];
    float ms_easy;               /* summed duration of k_trace_easy */
    uint32_t rays_redo;          /* rays (all levels, last chunk) re-traced through the reference kd-tree: kd-degenerate rays and exact distance ties */
    uint64_t rays_hard;          /* rays (all levels) that needed a mesh walk = rays traced by the dominant kernel */
    uint64_t shape_tests_easy;   /* RTHIP_COUNTERS=1: shape-table entries tested by k_trace_easy (shape_tests: by the dominant kernel) */
    /* since the context was created: launch graphs captured (one per frame SHAPE: params, frames per call, output and workspace
     * pointers, scene - the cameras are not part of it) and calls served by replaying one */
    uint32_t graphs_captured, graph_replays;
    /* RTHIP_COUNTERS=1: what the two shading kernels shaded in the last call, [0] = k_trace_easy (the rays it finishes itself),
     * [1] = k_shade (the rays that needed a mesh walk): hits per shape ordinal (shapes >= 63 share the last entry) and
     * Background::GetColor evaluations; the units of the flop accounting of those kernels (bench.py) */
    uint64_t shape_hits[2][64];
    uint64_t background_evals[2];
    /* rays_shadow / rays_reflect split by the kernel that shaded the parent ray (same [0] / [1]; always collected) */
    uint64_t rays_shadow_by_kernel[2], rays_reflect_by_kernel[2];
    /* RTHIP_COUNTERS=1: of shape_tests / shape_tests_easy, the entries that passed the bounding-ball test and went through the
     * world -> local transform and the intersection (meshes: the BVH root boxes / the walk); the others cost the ball test only */
    uint64_t shape_tests_full, shape_tests_easy_full;
} rthip_stats;

/* ---- lifecycle ---- */
/* replaces: the std::async worker pool of AbstractRaycaster::Render (Raytracer.cpp:52-66) */
int rthip_create(int device, rthip_ctx **out);
int rthip_destroy(rthip_ctx *ctx);
const char *rthip_last_error(void);

/* replaces: the pointer-linked scene the reference walks per ray (Scene.cpp:168-237, 469-493);
 * called once per frame after Scene::Compile (Raytracer.cpp:42) */
int rthip_upload_scene(rthip_ctx *ctx, const rthip_scene_desc *scene);
/* rthip_upload_scene with options.
 * RTHIP_UPLOAD_BUILD_BVH: every mesh that has triangles but came without a BVH (bvh_node_count == 0; its bvh_* fields and abs_max
 *   are ignored, and the scene's bvh_tris / bvh_tri_index then hold only the records of the meshes that do bring one, at their
 *   bvh_tri_base) gets its 4-wide BVH built ON THE DEVICE from the triangle records of the upload (csrc/rthip_bvh.hip): the host
 *   builder's own algorithm - top-down binned SAH (16 bins per axis over the centroid bounds, cost = area x ceil(count / 4)) down
 *   to single triangles, level-synchronous above 64 triangles and one wave per subtree below, then the optimal collapse to
 *   4-wide nodes with leaves of <= 4 triangles; boxes rounded outward exactly like the host builder's; every link checked
 *   before anything walks the tree (a failed check or a tree deeper than RTHIP_BVH_MAX_DEPTH is RTHIP_E_UNSUPPORTED: build it
 *   on the host).  replaces: the acceleration-structure half of Mesh::Compile (Scene.cpp:386-414) for the structure the frame
 *   kernels walk - at 2 M triangles 18 ms of device time (39 ms with the read-backs) instead of 0.45 s on 16 host threads, and
 *   0.27 GB less to upload.  The BVH takes no part in a result's arithmetic (see rthip_bvhnode).  RTHIP_BVH_BUILDER=ploc
 *   selects the Morton-order clustering builder kept for comparison.
 * RTHIP_UPLOAD_KEEP_GEOMETRY: rthip_update_scene. */
/* RTHIP_UPLOAD_BUILD_KD: every mesh that has triangles but came without a kd-tree (node_count == 0; its node_* / ref_* / max_depth /
 *   plane_* fields are ignored) gets the REFERENCE kd-tree built ON THE DEVICE from scene->tri_verts and scene->mesh_bounds
 *   (csrc/rthip_kd.hip): KDTreeTriangle's own decisions - depth % 3 axis, split = the mean of the 3n vertex coordinates summed
 *   sequentially in list order, any-vertex <= / >= membership in list order, the 1.5 n rule, depth cap 100 - level by level with
 *   all nodes of a level side by side, emitted in the pre-order layout above together with the kd_planes the flattener derives;
 *   byte-identical to the host builder's arrays.  replaces: KDTreeTriangle::KDTreeTriangle (Scene.cpp:18-95) as called by
 *   Mesh::Compile (Scene.cpp:386-414).  Without the flag a mesh with triangles and no kd nodes is RTHIP_E_INVALID. */
#define RTHIP_UPLOAD_BUILD_BVH 1u
#define RTHIP_UPLOAD_KEEP_GEOMETRY 2u
#define RTHIP_UPLOAD_BUILD_KD 4u
int rthip_upload_scene_ex(rthip_ctx *ctx, const rthip_scene_desc *scene, uint32_t flags);
/* The same for a scene whose MESH GEOMETRY is the one already uploaded: everything is replaced except the mesh
 * arrays (kd_*, tris, tri_normals, bvh_*), whose pointers in `scene` are ignored (may be null); their counts and
 * the meshes[] table must equal the uploaded ones except for normal_mode, which is taken from `scene` (Mesh::NormalMode
 * is a public field the reference reads live, Scene.cpp:456).  A frame-after-frame caller that only moves nodes, lights,
 * materials or the camera (the Sandbox preview) does not ship hundreds of MB per frame.
 * replaces: Mesh::Compile returning early when nothing was invalidated (Scene.cpp:386-388) */
int rthip_update_scene(rthip_ctx *ctx, const rthip_scene_desc *scene);

/* replaces: the per-pixel loop of AbstractRaycaster::DoRaytraceThread + ColorsToPixels
 * (Raytracer.cpp:123-170) for the Raycast of re::Raytracer (Raytracer.cpp:187-315).
 * out_pixels: host buffer, width*height uint32, index y*width+x, value = Color::GetHexValue
 * (Common.cpp:275-282).  With part_count > 1 only this part's pixels are written.
 * out_shape_id / out_tri_index (nullable, same indexing, int32): primary-hit shape ordinal
 * (pre-order index into shapes[], -1 = miss) and mesh-relative triangle index (-1 = none). */
int rthip_render(rthip_ctx *ctx, const rthip_camera *cam, const rthip_params *params,
                 uint32_t *out_pixels, int32_t *out_shape_id, int32_t *out_tri_index);

/* rthip_render for a run of frames: n_frames (1..RTHIP_MAX_BATCH) cameras over the same scene traced as one set of launches
 * (rthip_render_batch_device), then ONE device-to-host copy; out_pixels receives the frames one after the other
 * (width*height uint32 each).  Whole frames only (part_count = 1), no id images.
 * replaces: the frame-after-frame RenderSync loop over a camera path known ahead (Sandbox/Sandbox.cpp:315-359). */
int rthip_render_batch(rthip_ctx *ctx, const rthip_camera *cams, uint32_t n_frames, const rthip_params *params, uint32_t *out_pixels);

/* Page-locked host memory (hipHostMalloc) for a buffer the caller keeps passing as out_pixels, so that the
 * device-to-host copy of rthip_render runs at PCIe rate instead of through the runtime's staging buffers.
 * replaces: the `new unsigned[w * h]` of the renderer-owned m_Pixels (Raytracer.cpp:20-25), which every Render
 * returns.  Free with rthip_host_free before rthip_destroy. */
int rthip_host_alloc(rthip_ctx *ctx, size_t bytes, void **out);
int rthip_host_free(rthip_ctx *ctx, void *ptr);

/* Same frame, output left in device memory (HBM): d_pixels is a device pointer to
 * width*height uint32 (row-major); only this part's pixels are written.  Asynchronous on the
 * context's stream unless `sync` is non-zero.  Used by bench.py and the multi-GPU gather. */
int rthip_render_device(rthip_ctx *ctx, const rthip_camera *cam, const rthip_params *params,
                        void *d_pixels, int sync);
/* A run of frames in one call: n_frames (1..RTHIP_MAX_BATCH) cameras over the same scene and params, traced as one
 * set of launches (the rays of all the frames share the ray queues, so the fixed cost of a frame's ~20 dependent
 * launches is paid once per call).  d_pixels: n_frames buffers one after the other, each width*height (layout 0) or
 * rthip_part_pixels() (layout 1) uint32.  Every frame is bit-identical to what rthip_render_device gives for its
 * camera.  replaces: the frame-after-frame RenderSync loop of the preview (Sandbox/Sandbox.cpp:359), whose camera
 * path is known ahead; also what keeps a GPU busy when its share of a frame is small (part_count > 1). */
int rthip_render_batch_device(rthip_ctx *ctx, const rthip_camera *cams, uint32_t n_frames, const rthip_params *params,
                              void *d_pixels, int sync);
/* rthip_render_device for callers that CANNOT ALLOCATE device memory (tight serving loops, harnesses that forbid hipMalloc,
 * code that must not grow the process's HBM footprint).  No counterpart in the reference.  The call launches on the context's
 * OWN stream and may synchronise it (when an earlier frame's rthip_interrupt flag has to be lowered: hipStreamSynchronize +
 * hipMemset of one word) and may call hipFuncSetAttribute for deep trees: it allocates nothing, but it is not itself
 * capturable into a caller's HIP graph.  Same frame, same kernels, bit-identical output; the
 * per-frame workspace (ray queues, hit / colour records, list counters) lives in a file-scope `static __device__` array of
 * RTHIP_NOALLOC_SCRATCH_BYTES (64 MiB) that every GPU the library runs on holds until the process ends, instead of in
 * hipMalloc'ed buffers, and the call performs no hipMalloc / hipFree / graph instantiation / event creation.  Limits: one
 * frame; the workspace, (36 * (max_recursion + 1) + 216) bytes per level-0 ray (x9 with SSAA) plus ~20 KB, must fit the
 * array (e.g. 512x360 at MaxRecursion 3) - larger frames are rejected with RTHIP_E_UNSUPPORTED and a message that names the
 * size; no id images, no statistics.  All contexts of one GPU share that GPU's array: calls that overlap on one GPU race
 * (the array keeps the previous call's records; the counters that must start at zero are cleared by the call itself). */
#define RTHIP_NOALLOC_SCRATCH_BYTES (64u << 20)
int rthip_render_device_noalloc(rthip_ctx *ctx, const rthip_camera *cam, const rthip_params *params, void *d_pixels, int sync);
/* number of uint32 entries of one part's tile-compact buffer (identical for every part_index:
 * ceil(n_tiles / part_count) * tile_w * tile_h), 0 on bad params */
uint64_t rthip_part_pixels(const rthip_params *params);
/* replaces: nothing in the reference (single process); multi-GPU only.  d_gathered holds
 * part_count tile-compact buffers of rthip_part_pixels() entries each, part after part (the result
 * of an all-gather / gather over RCCL); writes the width*height row-major frame to d_frame. */
int rthip_assemble(rthip_ctx *ctx, const rthip_params *params, const void *d_gathered, void *d_frame, int sync);
/* rthip_assemble for the output of rthip_render_batch_device gathered over the ranks: d_gathered holds, part after
 * part, n_frames tile-compact buffers each; d_frames receives n_frames row-major frames one after the other. */
int rthip_assemble_batch(rthip_ctx *ctx, const rthip_params *params, uint32_t n_frames, const void *d_gathered, void *d_frames,
                         int sync);
/* HIP stream the context launches on (hipStream_t as void*), so callers can order work after it */
void *rthip_stream(rthip_ctx *ctx);
int rthip_synchronize(rthip_ctx *ctx);

/* replaces: RenderStatus::Percent polling and Interrupt() (Raytracer.cpp:91-99, 150-154);
 * both callable from another thread while rthip_render runs */
float rthip_progress(rthip_ctx *ctx);
int rthip_interrupt(rthip_ctx *ctx);

int rthip_get_stats(rthip_ctx *ctx, rthip_stats *out);
/* After an ASYNCHRONOUS rthip_render_device / rthip_render_batch_device (sync = 0): waits for the context's stream, gathers
 * the statistics of that call (params = the ones it was made with) into *out (nullable) and returns RTHIP_E_INTERRUPTED if
 * rthip_interrupt arrived during it.  What the synchronous calls do at their end. */
int rthip_collect_stats(rthip_ctx *ctx, const rthip_params *params, rthip_stats *out);

/* ---- several GPUs of one node from ONE process ----
 * replaces: the NumThreads std::async workers of AbstractRaycaster::Render (Raytracer.cpp:52-87) - the reference's only
 * parallel decomposition - by one context + stream per GPU driven from the calling thread.  The screen is
 * split into interleaved tiles (tile t -> devices[t % n]); every device renders its share into a tile-compact buffer in its own
 * HBM; the parts are moved to devices[0] over xGMI with RCCL (ncclCommInitAll, grouped ncclSend / ncclRecv; librccl is opened
 * with dlopen when the first multi-GPU renderer is created) and assembled there; one D2H copy returns the frame.  The image is
 * bit-identical to the single-GPU image.  Repeated ordinals (the same GPU several times: how a one-GPU box rehearses the path)
 * and hosts without librccl use peer / device-to-device copies instead; RTHIP_MULTI_TRANSPORT=copy|rccl forces either. */
typedef struct rthip_multi rthip_multi;
int rthip_multi_create(const int *devices, int n_devices, rthip_multi **out);
int rthip_multi_destroy(rthip_multi *m);
int rthip_multi_device_count(rthip_multi *m);
rthip_ctx *rthip_multi_context(rthip_multi *m, int i);   /* the context of devices[i] (owned by m): host allocations, probes, statistics */
const char *rthip_multi_transport(rthip_multi *m);   /* "RCCL ncclSend/ncclRecv" or "peer copies" */
/* the scene is replicated on every device; flags: RTHIP_UPLOAD_* as for rthip_upload_scene_ex */
int rthip_multi_upload_scene(rthip_multi *m, const rthip_scene_desc *scene, uint32_t flags);
/* one frame into host memory (width*height uint32, as rthip_render); params->part_* / layout are set by the call;
 * out_stats (nullable): ray counts summed over the devices, ms_total = the slowest device's */
int rthip_multi_render(rthip_multi *m, const rthip_camera *cam, const rthip_params *params, uint32_t *out_pixels, rthip_stats *out_stats);
/* a run of frames (1..RTHIP_MAX_BATCH cameras) split over the devices in ONE round: each device traces its tiles of all the frames
 * as one set of launches, the parts cross xGMI once, one assemble, one device-to-host copy of n_frames * width*height uint32 into
 * out_pixels, one synchronisation (statistics are summed over devices and frames).  A device's share of a single frame is
 * latency-bound; a run gives every device whole frames' worth of rays per launch.
 * replaces: the worker pool of AbstractRaycaster::Render (Raytracer.cpp:52-87) for a queued camera path */
int rthip_multi_render_batch(rthip_multi *m, const rthip_camera *cams, uint32_t n_frames, const rthip_params *params, uint32_t *out_pixels,
                             rthip_stats *out_stats);
float rthip_multi_progress(rthip_multi *m);
int rthip_multi_interrupt(rthip_multi *m);

/* Known-answer probes: run single device functions on n inputs (device parity tests of the
 * S1/P1/P2/T2/T4, N0-N4, M1/M2, B1/B2 and R4 rows of SURVEY.md §8a against the oracle). */
/* Scene::CastRay (Scene.cpp:163-237): o,d n*3 -> hit n, point/local/normal n*3, shape n, tri n */
int rthip_probe_cast(rthip_ctx *ctx, uint32_t n, const double *o, const double *d, int32_t *hit,
                     double *point, double *local, double *normal, int32_t *shape, int32_t *tri);
/* Noise::Sample (noise/Noise.cpp:3-19) */
int rthip_probe_noise(rthip_ctx *ctx, int32_t noise, uint32_t n, const double *p, double *out);
/* Material::GetAbsorbedColor / GetAbsorptance (Material.cpp:21-29) */
int rthip_probe_material(rthip_ctx *ctx, int32_t material, uint32_t n, const double *p,
                         double *color, double *absorptance);
/* Background::GetColor (Scene.cpp:111-151) */
int rthip_probe_background(rthip_ctx *ctx, uint32_t n, const double *d, double *color);
/* Raytracer::Raycast (Raytracer.cpp:312-315) on explicit rays -> unpacked colours n*3 */
int rthip_probe_raycast(rthip_ctx *ctx, uint32_t max_recursion, uint32_t n, const double *o,
                        const double *d, double *color);
/* The BVH of mesh `mesh` as the device holds it (test access; matters for BVHs built by RTHIP_UPLOAD_BUILD_BVH): its mesh-table
 * entry, its nodes (room for cap_nodes) and its leaf-order triangle indices (room for cap_tris); null pointers are skipped */
int rthip_probe_bvh(rthip_ctx *ctx, uint32_t mesh, rthip_mesh *out_mesh, rthip_bvhnode *nodes, uint64_t cap_nodes, uint32_t *tri_index, uint64_t cap_tris);
/* The kd-tree of mesh `mesh` as the device holds it (test access; matters for trees built by RTHIP_UPLOAD_BUILD_KD): its mesh-table
 * entry (node_count, ref_count, max_depth, plane_count[]), nodes / bounds (6 doubles per node; room for cap_nodes nodes), leaf
 * references (room for cap_refs) and the three runs of plane coordinates one after the other (room for cap_planes); null
 * pointers are skipped */
int rthip_probe_kd(rthip_ctx *ctx, uint32_t mesh, rthip_mesh *out_mesh, rthip_kdnode *nodes, double *bounds, uint64_t cap_nodes, uint32_t *refs,
                   uint64_t cap_refs, double *planes, uint64_t cap_planes);
/* what the last rthip_upload_scene* of this context built on the device */
typedef struct {
    uint32_t kd_meshes, kd_nodes, kd_levels;        /* meshes whose kd-tree was built on the device; their nodes; the deepest tree's levels */
    uint32_t walk_coop;                             /* 1: long lists of rays that need a mesh walk go through k_trace_coop (one ray per lane outside the walks; every
                                                     * scene with meshes unless RTHIP_QUAD_COOP / RTHIP_COOP_MAX_TREE_MB say otherwise), 0: every list through k_trace_quad */
    uint64_t kd_refs;
    float kd_device_ms;                             /* device time of those builds (events around them) */
    uint32_t bvh_meshes, bvh_nodes;                 /* meshes whose BVH was built on the device; their nodes */
    float upload_ms;                                /* host wall time of the whole call */
} rthip_upload_info;
int rthip_get_upload_info(rthip_ctx *ctx, rthip_upload_info *out);
/* IEEE arithmetic self-check of the device (add/mul/div/sqrt/floor/fmodf/casts on n inputs):
 * op 0 a+b, 1 a*b, 2 a/b, 3 sqrt(a), 4 floor(a), 5 (double)fmodf((float)a,(float)b),
 * 6 (double)(float)a, 7 sin(a), 8 (double)(int)a, 9 fmax(a,b), 10 fmin(a,b) */
int rthip_probe_arith(rthip_ctx *ctx, int32_t op, uint32_t n, const double *a, const double *b, double *out);

/* diagnostic: number of hipMalloc calls this library has made for scene and frame buffers since it was loaded (all contexts).
 * tests/test_zz_no_alloc.py checks that rthip_render_device_noalloc leaves it unchanged. */
uint64_t rthip_debug_device_allocs(void);
const char *rthip_version(void);

#ifdef __cplusplus
}
#endif
#endif
