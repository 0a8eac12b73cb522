/*
 * ref_harness.cpp — TEST INFRASTRUCTURE (oracle/_ref).  Container-only: it is compiled
 * against the genuine reference sources where they lie under /root/reference (never copied),
 * with -fno-access-control so private tables can be injected/dumped, and exports the same
 * sb_* C entry points as oracle/rt_oracle.h.  Used to (1) pin the C restatement in
 * rt_oracle.c bit-exactly, (2) generate the golden vectors under tests/golden/,
 * (3) time the reference's own std::thread renderer as bench.py's cpu_baseline ("reference").
 *
 * Scene.cpp is included textually because re::KDTreeTriangle is a file-local class; the other
 * reference TUs are compiled separately, unmodified (see oracle/Makefile).
 */
#include "Scene.cpp"  // from -I<reference>/Raytracer
#include "Raytracer.h"
#include "noise/CheckerBoard.h"
#include "noise/Marble.h"
#include "noise/Perlin.h"
#include "noise/Worley.h"

#include <atomic>
#include <cstring>
#include <map>
#include <unordered_map>

#include "rt_oracle.h"

#include "ref_scene.h"

using namespace re;

/* --- ray counter: link-time --wrap of re::Scene::CastRay as seen from Raytracer.o --- */
static std::atomic<uint64_t> g_flushed{0};
struct TlCount {
    uint64_t n = 0;
    ~TlCount() { g_flushed.fetch_add(n, std::memory_order_relaxed); }
};
static thread_local TlCount tl_count;
extern "C" Scene::RaycastResult __real__ZN2re5Scene7CastRayERKNS_3RayE(Scene *self, const Ray &ray);
extern "C" Scene::RaycastResult __wrap__ZN2re5Scene7CastRayERKNS_3RayE(Scene *self, const Ray &ray) {
    tl_count.n++;
    return __real__ZN2re5Scene7CastRayERKNS_3RayE(self, ray);
}

static void sb_forget_tri_indices();
static Vector3 V3(const double *p) { return Vector3(p[0], p[1], p[2]); }
static void set_table(Perlin *p, const uint8_t *t) {
    if (t)
        for (int i = 0; i < 256; i++) p->m_Seed[i] = t[i];
}
static void ordinals(sb_scene *s, SceneNode *n, int &ord) {
    if (n->GetComponentOfType<Shape>() != nullptr) s->ordinal[n] = ord++;
    for (auto &c : n->GetChildren()) ordinals(s, c.get(), ord);
}
static void ensure_compiled(sb_scene *s) {
    if (s->compiled) return;
    s->scene->Compile();
    s->ordinal.clear();
    int ord = 0;
    ordinals(s, s->scene->GetRoot().get(), ord);
    s->compiled = true;
}

extern "C" {

const char *sb_impl_name(void) { return "reference"; }

sb_scene *sb_scene_new(void) {
    auto *s = new sb_scene;
    s->scene = std::make_shared<Scene>();
    s->nodes.push_back(s->scene->GetRoot().get());
    return s;
}
void sb_scene_free(sb_scene *s) {
    sb_forget_tri_indices();
    delete s;
}
void sb_camera(sb_scene *s, const double pos[3], const double dir[3]) {
    s->scene->Camera.Position = V3(pos);
    s->scene->Camera.Direction = V3(dir);
}
int sb_mat_uniform(sb_scene *s, unsigned rgb, double absorptance) {
    s->mats.push_back(std::shared_ptr<Material>(new UniformMaterial(rgb, absorptance)));
    return (int)s->mats.size() - 1;
}
int sb_mat_interp(sb_scene *s, int noise, int m0, int m1) {
    if (noise < 0 || noise >= (int)s->noises.size() || m0 < 0 || m0 >= (int)s->mats.size() || m1 < 0 ||
        m1 >= (int)s->mats.size())
        return -1;
    s->mats.push_back(std::shared_ptr<Material>(new InterpolatedMaterial(s->noises[noise], s->mats[m0], s->mats[m1])));
    return (int)s->mats.size() - 1;
}
int sb_noise_checker(sb_scene *s, double domain) {
    s->noises.push_back(std::shared_ptr<Noise>(new CheckerBoard(domain)));
    s->noise_type.push_back(0);
    return (int)s->noises.size() - 1;
}
int sb_noise_perlin(sb_scene *s, double domain, const uint8_t *table256) {
    auto *p = new Perlin(domain);
    set_table(p, table256);
    s->noises.push_back(std::shared_ptr<Noise>(p));
    s->noise_type.push_back(1);
    return (int)s->noises.size() - 1;
}
int sb_noise_worley(sb_scene *s, double domain, int divisions, const double *points) {
    auto *w = new Worley(domain, divisions);
    if (points)
        for (size_t i = 0; i < w->m_Points.size(); i++) w->m_Points[i] = V3(points + i * 3);
    s->noises.push_back(std::shared_ptr<Noise>(w));
    s->noise_type.push_back(2);
    return (int)s->noises.size() - 1;
}
int sb_noise_marble(sb_scene *s, double domain, double freq, double turb, const uint8_t *table256) {
    auto *m = new Marble(domain, freq, turb);
    set_table(m->m_Perlin.get(), table256);
    s->noises.push_back(std::shared_ptr<Noise>(m));
    s->noise_type.push_back(3);
    return (int)s->noises.size() - 1;
}
int sb_node(sb_scene *s, int parent, const double pos[3], const double rot[3], const double scale[3]) {
    if (parent < 0) parent = 0;
    if (parent >= (int)s->nodes.size()) return -1;
    auto child = s->nodes[parent]->AddChild();
    auto t = child->GetComponentOfType<Transform>();
    t->Position = V3(pos);
    t->Rotation = V3(rot);
    t->Scale = V3(scale);
    s->nodes.push_back(child.get());
    s->compiled = false;
    return (int)s->nodes.size() - 1;
}
static bool node_ok(sb_scene *s, int node, int mat) {
    return node >= 0 && node < (int)s->nodes.size() && mat >= 0 && mat < (int)s->mats.size() &&
           s->nodes[node]->GetComponentOfType<Shape>() == nullptr;
}
int sb_sphere(sb_scene *s, int node, int mat) {
    if (!node_ok(s, node, mat)) return -1;
    auto sh = s->nodes[node]->AddComponent<Sphere>();
    sh->Material = s->mats[mat].get();
    s->compiled = false;
    return 0;
}
int sb_plane(sb_scene *s, int node, int mat, const double n[3]) {
    if (!node_ok(s, node, mat)) return -1;
    auto sh = s->nodes[node]->AddComponent<Plane>();
    sh->Normal = V3(n);
    sh->Material = s->mats[mat].get();
    s->compiled = false;
    return 0;
}
int sb_mesh(sb_scene *s, int node, int mat, int normal_mode, int ntris, const double *verts, const double *normals) {
    if (!node_ok(s, node, mat)) return -1;
    auto mesh = s->nodes[node]->AddComponent<Mesh>();
    for (int i = 0; i < ntris; i++) {
        auto &t = mesh->AddTriangle();
        t.Vertices = {V3(verts + i * 9), V3(verts + i * 9 + 3), V3(verts + i * 9 + 6)};
        if (normals) t.Normals = {V3(normals + i * 9), V3(normals + i * 9 + 3), V3(normals + i * 9 + 6)};
    }
    mesh->NormalMode = normal_mode ? NormalModes::Vertex : NormalModes::Face;
    mesh->Material = s->mats[mat].get();
    s->compiled = false;
    return 0;
}
int sb_light(sb_scene *s, int type, unsigned rgb, const double dir[3], const double pos[3], double attenuation, int enabled) {
    auto l = std::make_shared<Light>();
    l->Type = type == 0 ? LightType::Directional : (type == 1 ? LightType::Ambient : LightType::Point);
    l->Color = rgb;
    l->Direction = V3(dir);
    l->Position = V3(pos);
    l->Attenuation = attenuation;
    l->Enabled = enabled != 0;
    s->scene->Lights.push_back(l);
    s->lights.push_back(l);
    return (int)s->lights.size() - 1;
}
void sb_bg_color(sb_scene *s, unsigned rgb) { s->scene->Background = std::shared_ptr<Background>(new ColorBackground(rgb)); }
void sb_bg_skybox(sb_scene *s, unsigned rgb0, unsigned rgb1, int sun_light, const uint8_t *table256) {
    std::shared_ptr<Light> sun = (sun_light >= 0 && sun_light < (int)s->lights.size()) ? s->lights[sun_light] : nullptr;
    auto *sky = new SkyBox(rgb0, rgb1, sun);
    set_table(sky->m_Perlin.get(), table256);
    s->scene->Background = std::shared_ptr<Background>(sky);
}
int sb_get_perlin_table(sb_scene *s, int noise, uint8_t *out256) {
    if (noise < 0 || noise >= (int)s->noises.size()) return -1;
    Perlin *p = nullptr;
    if (s->noise_type[noise] == 1) p = static_cast<Perlin *>(s->noises[noise].get());
    if (s->noise_type[noise] == 3) p = static_cast<Marble *>(s->noises[noise].get())->m_Perlin.get();
    if (!p) return -1;
    for (int i = 0; i < 256; i++) out256[i] = (uint8_t)p->m_Seed[i];
    return 0;
}
int sb_get_skybox_table(sb_scene *s, uint8_t *out256) {
    auto *sky = dynamic_cast<SkyBox *>(s->scene->Background.get());
    if (!sky) return -1;
    for (int i = 0; i < 256; i++) out256[i] = (uint8_t)sky->m_Perlin->m_Seed[i];
    return 0;
}
int sb_get_worley_points(sb_scene *s, int noise, double *out) {
    if (noise < 0 || noise >= (int)s->noises.size() || s->noise_type[noise] != 2) return -1;
    auto *w = static_cast<Worley *>(s->noises[noise].get());
    for (size_t i = 0; i < w->m_Points.size(); i++) {
        out[i * 3] = w->m_Points[i].X;
        out[i * 3 + 1] = w->m_Points[i].Y;
        out[i * 3 + 2] = w->m_Points[i].Z;
    }
    return 0;
}
}  // extern "C"

/* ---- triangle index recovery: the reference stores leaf triangles by value and has no ids --- */
struct TriKey {
    double v[9];
    bool operator==(const TriKey &o) const { return std::memcmp(v, o.v, sizeof v) == 0; }
};
struct TriKeyHash {
    size_t operator()(const TriKey &k) const {
        uint64_t h = 1469598103934665603ull;
        const unsigned char *p = reinterpret_cast<const unsigned char *>(k.v);
        for (size_t i = 0; i < sizeof k.v; i++) h = (h ^ p[i]) * 1099511628211ull;
        return (size_t)h;
    }
};
static TriKey key_of(const Triangle &t) {
    TriKey k;
    for (int i = 0; i < 3; i++) {
        k.v[i * 3] = t.Vertices[i].X;
        k.v[i * 3 + 1] = t.Vertices[i].Y;
        k.v[i * 3 + 2] = t.Vertices[i].Z;
    }
    return k;
}
typedef std::unordered_map<TriKey, int, TriKeyHash> TriIndex;
static std::map<Mesh *, TriIndex> g_tri_index;
static std::mutex g_tri_index_mu;
static const TriIndex &tri_index(Mesh *m) {
    std::lock_guard<std::mutex> lock(g_tri_index_mu);
    auto it = g_tri_index.find(m);
    if (it != g_tri_index.end() && it->second.size() > 0) return it->second;
    TriIndex &idx = g_tri_index[m];
    for (size_t i = 0; i < m->m_Triangles.size(); i++) idx.emplace(key_of(m->m_Triangles[i]), (int)i);
    return idx;
}
static void sb_forget_tri_indices() {
    std::lock_guard<std::mutex> lock(g_tri_index_mu);
    g_tri_index.clear();
}
/* re-walk in the order of Mesh::IntersectInternal (Scene.cpp:469-493) using the reference's own tests */
static void rewalk(Mesh *m, const Ray &ray, KDTreeTriangle *node, real &distance, const Triangle *&best) {
    if (node->Bounds.Intersect(ray).Hit) {
        for (auto &t : node->Triangles) {
            auto r = m->IntersectTriangle(ray, t);
            auto d = (ray.Origin - r.Point).SquaredLength();
            if (d < distance && r.Hit) {
                distance = d;
                best = &t;
            }
        }
        if (node->Left) rewalk(m, ray, node->Left, distance, best);
        if (node->Right) rewalk(m, ray, node->Right, distance, best);
    }
}
static int winning_triangle(Scene::RaycastResult &r, const Ray &ray) {
    if (!r.Hit) return -1;
    auto mesh = r.Node->GetComponentOfType<Mesh>();
    if (mesh == nullptr) return -1;
    auto transform = r.Node->GetComponentOfType<Transform>();
    Matrix4 itmat;
    transform->GetInverseTransform(itmat);
    Ray local;  // Scene.cpp:185-186
    local.Origin = itmat * Vector4(ray.Origin, 1);
    local.Direction = Vector3(itmat * Vector4(ray.Direction, 0)).Normalized();
    real distance = std::numeric_limits<real>::max();
    const Triangle *best = nullptr;
    rewalk(mesh.get(), local, mesh->m_KdTree, distance, best);
    if (!best) return -1;
    const TriIndex &idx = tri_index(mesh.get());
    auto it = idx.find(key_of(*best));
    return it == idx.end() ? -2 : it->second;
}

namespace {
struct DebugRt : DebugRaycaster {
    using DebugRaycaster::DebugRaycaster;
};
}  // namespace

extern "C" {

int sb_render(sb_scene *s, const sb_render_params *p, uint32_t *pixels, int32_t *shape_ord, int32_t *tri_idx,
              uint64_t *ray_count, const int32_t *sub_x, const int32_t *sub_y, int n_sub) {
    ensure_compiled(s);
    real fovy = p->fovy > 0 ? p->fovy : PI / 4;
    std::unique_ptr<AbstractRaycaster> rc;
    if (p->renderer == 0) {
        auto *rt = new Raytracer(p->width, p->height, fovy);
        rt->MaxRecursion = p->max_recursion;
        rc.reset(rt);
    } else {
        auto *dr = new DebugRaycaster(p->width, p->height, fovy);
        dr->Mode = p->renderer == 1 ? DebugRaycaster::Modes::Normal : DebugRaycaster::Modes::Color;
        rc.reset(dr);
    }
    rc->NumThreads = p->num_threads > 0 ? p->num_threads : 1;
    rc->Antialiasing = p->aa_mode == 1 ? AbstractRaycaster::AAMode::SSAA : AbstractRaycaster::AAMode::None;
    g_flushed = 0;
    tl_count.n = 0;
    Scene *scene = s->scene.get();
    if (n_sub <= 0) {
        unsigned *px = rc->RenderSync(scene);
        std::memcpy(pixels, px, sizeof(uint32_t) * (size_t)p->width * p->height);
    } else {
        for (int i = 0; i < n_sub; i++) {
            unsigned x = sub_x[i];
            int y = sub_y[i];
            if (p->aa_mode == 1) {  // Raytracer.cpp:132-143
                Color result(Color::Black);
                for (int dx = -1; dx <= 1; dx++)
                    for (int dy = -1; dy <= 1; dy++) {
                        Ray ray = rc->CreateScreenRay(scene, x + dx * .5f, y + dy * .5f);
                        result += (rc->Raycast(scene, ray) * (1.0f / 9.0f));
                    }
                pixels[i] = result.GetHexValue();
            } else {
                Ray ray = rc->CreateScreenRay(scene, x, y);
                pixels[i] = rc->Raycast(scene, ray).GetHexValue();
            }
        }
    }
    if (ray_count) *ray_count = g_flushed.load() + tl_count.n;
    if (shape_ord || tri_idx) {
        size_t npx = n_sub > 0 ? (size_t)n_sub : (size_t)p->width * p->height;
        for (size_t i = 0; i < npx; i++) {
            unsigned x = n_sub > 0 ? (unsigned)sub_x[i] : (unsigned)(i % p->width);
            int y = n_sub > 0 ? sub_y[i] : (int)(i / p->width);
            Ray ray = rc->CreateScreenRay(scene, x, y);
            auto r = scene->CastRay(ray);
            if (shape_ord) shape_ord[i] = r.Hit ? s->ordinal[r.Node] : -1;
            if (tri_idx) tri_idx[i] = winning_triangle(r, ray);
        }
    }
    return 0;
}

int sb_cast_rays(sb_scene *s, int n, const double *o, const double *d, int32_t *out_hit, double *out_point,
                 double *out_local, double *out_normal, int32_t *out_shape, int32_t *out_tri) {
    ensure_compiled(s);
    for (int i = 0; i < n; i++) {
        Ray ray;
        ray.Origin = V3(o + i * 3);
        ray.Direction = V3(d + i * 3);
        auto r = s->scene->CastRay(ray);
        out_hit[i] = r.Hit;
        if (out_point) { out_point[i * 3] = r.Point.X; out_point[i * 3 + 1] = r.Point.Y; out_point[i * 3 + 2] = r.Point.Z; }
        if (out_local) { out_local[i * 3] = r.LocalPoint.X; out_local[i * 3 + 1] = r.LocalPoint.Y; out_local[i * 3 + 2] = r.LocalPoint.Z; }
        if (out_normal) { out_normal[i * 3] = r.Normal.X; out_normal[i * 3 + 1] = r.Normal.Y; out_normal[i * 3 + 2] = r.Normal.Z; }
        if (out_shape) out_shape[i] = r.Hit ? s->ordinal[r.Node] : -1;
        if (out_tri) out_tri[i] = winning_triangle(r, ray);
    }
    return 0;
}
int sb_noise_sample(sb_scene *s, int noise, int n, const double *p, double *out) {
    if (noise < 0 || noise >= (int)s->noises.size()) return -1;
    for (int i = 0; i < n; i++) out[i] = s->noises[noise]->Sample(V3(p + i * 3));
    return 0;
}
int sb_material_eval(sb_scene *s, int mat, int n, const double *p, double *out_color, double *out_absorptance) {
    if (mat < 0 || mat >= (int)s->mats.size()) return -1;
    for (int i = 0; i < n; i++) {
        Color c = s->mats[mat]->GetAbsorbedColor(V3(p + i * 3));
        out_color[i * 3] = c.R; out_color[i * 3 + 1] = c.G; out_color[i * 3 + 2] = c.B;
        out_absorptance[i] = s->mats[mat]->GetAbsorptance(V3(p + i * 3));
    }
    return 0;
}
int sb_background_eval(sb_scene *s, int n, const double *d, double *out_color) {
    for (int i = 0; i < n; i++) {
        Color c = s->scene->Background ? s->scene->Background->GetColor(V3(d + i * 3)) : Color::Black;
        out_color[i * 3] = c.R; out_color[i * 3 + 1] = c.G; out_color[i * 3 + 2] = c.B;
    }
    return 0;
}
int sb_raycast(sb_scene *s, int max_recursion, int n, const double *o, const double *d, double *out_color) {
    ensure_compiled(s);
    Raytracer rt(4, 4);
    rt.MaxRecursion = max_recursion;
    for (int i = 0; i < n; i++) {
        Ray ray;
        ray.Origin = V3(o + i * 3);
        ray.Direction = V3(d + i * 3);
        Color c = rt.Raycast(s->scene.get(), ray);
        out_color[i * 3] = c.R; out_color[i * 3 + 1] = c.G; out_color[i * 3 + 2] = c.B;
    }
    return 0;
}
int sb_screen_rays(sb_scene *s, int width, int height, double fovy, int n, const double *xy, double *out_o, double *out_d) {
    Raytracer rt(width, height, fovy > 0 ? fovy : PI / 4);
    for (int i = 0; i < n; i++) {
        Ray r = rt.CreateScreenRay(s->scene.get(), xy[i * 2], xy[i * 2 + 1]);
        out_o[i * 3] = r.Origin.X; out_o[i * 3 + 1] = r.Origin.Y; out_o[i * 3 + 2] = r.Origin.Z;
        out_d[i * 3] = r.Direction.X; out_d[i * 3 + 1] = r.Direction.Y; out_d[i * 3 + 2] = r.Direction.Z;
    }
    return 0;
}
static void kd_dump_rec(Mesh *m, KDTreeTriangle *k, int &ni, int64_t &nr, int32_t *depth, int32_t *ntris, double *split,
                        double *bounds, int32_t *refs) {
    int i = ni++;
    if (depth) depth[i] = (int32_t)k->Depth;
    if (ntris) ntris[i] = (int32_t)k->Triangles.size();
    if (split) {
        unsigned ax = k->Depth % 3;
        split[i] = k->Left ? k->Left->Bounds.Max.Elements[ax] : (k->Right ? k->Right->Bounds.Min.Elements[ax] : 0.0);
    }
    if (bounds) {
        bounds[i * 6 + 0] = k->Bounds.Min.X; bounds[i * 6 + 1] = k->Bounds.Min.Y; bounds[i * 6 + 2] = k->Bounds.Min.Z;
        bounds[i * 6 + 3] = k->Bounds.Max.X; bounds[i * 6 + 4] = k->Bounds.Max.Y; bounds[i * 6 + 5] = k->Bounds.Max.Z;
    }
    if (!k->Triangles.empty()) {
        const TriIndex &idx = tri_index(m);
        for (auto &t : k->Triangles) {
            if (refs) {
                auto it = idx.find(key_of(t));
                refs[nr] = it == idx.end() ? -2 : it->second;
            }
            nr++;
        }
    }
    if (k->Left) kd_dump_rec(m, k->Left, ni, nr, depth, ntris, split, bounds, refs);
    if (k->Right) kd_dump_rec(m, k->Right, ni, nr, depth, ntris, split, bounds, refs);
}
int sb_kd_dump(sb_scene *s, int node, int32_t *node_depth, int32_t *node_ntris, double *node_split,
               double *node_bounds, int32_t *tri_refs, int64_t *n_refs) {
    ensure_compiled(s);
    if (node < 0 || node >= (int)s->nodes.size()) return -1;
    auto mesh = s->nodes[node]->GetComponentOfType<Mesh>();
    if (mesh == nullptr) return -1;
    int ni = 0;
    int64_t nr = 0;
    kd_dump_rec(mesh.get(), mesh->m_KdTree, ni, nr, node_depth, node_ntris, node_split, node_bounds, tri_refs);
    if (n_refs) *n_refs = nr;
    return ni;
}
int sb_node_matrices(sb_scene *s, int node, double *out48) {
    ensure_compiled(s);
    if (node < 0 || node >= (int)s->nodes.size()) return -1;
    auto t = s->nodes[node]->GetComponentOfType<Transform>();
    Matrix4 a, b, c;
    t->GetTransform(a);
    t->GetInverseTransform(b);
    t->GetNormalTransform(c);
    std::memcpy(out48, a.Elements, sizeof(double) * 16);
    std::memcpy(out48 + 16, b.Elements, sizeof(double) * 16);
    std::memcpy(out48 + 32, c.Elements, sizeof(double) * 16);
    return 0;
}
}  // extern "C"
