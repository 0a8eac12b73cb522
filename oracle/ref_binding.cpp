/*
 * ref_binding.cpp — the drop-in binding PROVEN ON THE GENUINE HEADERS (container-only test infrastructure, built into
 * oracle/_ref/libref.so by oracle/Makefile next to the unmodified reference sources; never shipped as product code).
 *
 * What a maintainer of erbuka/raytracing adds to use librthip.so, compiled here against
 * /root/reference/Raytracer/*.h — not against this repository's mirror of that API:
 *   1. re::HipRaytracer : public re::AbstractRaycaster (Raytracer/Raytracer.h:55-98): Render / GetStatus / Interrupt
 *      overridden with the contract of Raytracer.cpp:34-99 (returns at once, detached thread, scene->Compile() first,
 *      promise satisfied exactly once, Pixels = m_Pixels);
 *   2. a flattener re::Scene -> rthip_scene_desc.  The data it needs is private without getters (Scene.h:203-214,
 *      Material.h:43-45, Perlin.h:26, Worley.h:19-21, Marble.h:15-16, Scene.h:60-64); upstream would grant it with
 *      `friend struct re::hipbind::Flattener;` lines — this TU is built with -fno-access-control instead, so the
 *      reference sources stay untouched;
 *   3. the kd-tree in the flat layout: re::KDTreeTriangle is file-local to Scene.cpp, so the binding rebuilds it from
 *      Mesh::m_Triangles with the SAME template code the product's mirror uses (raytracing_amd/host/mesh_build.hpp), whose
 *      splits are those of Scene.cpp:18-95 (pinned: tests/test_host.py::test_host_kdtree_is_the_reference_tree).
 * tests/test_ref_binding.py flattens the BASELINE configs through this file and through the mirror and compares every
 * array of the two rthip_scene_desc byte for byte.
 */
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>

#include "Raytracer.h" /* genuine: -I<reference>/Raytracer */
#include "noise/CheckerBoard.h"
#include "noise/Marble.h"
#include "noise/Perlin.h"
#include "noise/Worley.h"

#include "mesh_build.hpp" /* -I<repo>/raytracing_amd/host : shared kd / BVH builders, no scene classes */
#include "ref_scene.h"
#include "rthip.h"

namespace re {
namespace hipbind {

struct FlatScene {
    std::vector<rthip_shape> shapes;
    std::vector<rthip_mesh> meshes;
    std::vector<rthip_kdnode> kd_nodes;
    std::vector<double> kd_bounds;
    std::vector<uint32_t> kd_refs;
    std::vector<double> kd_planes;
    std::vector<rthip_tri> tris;
    std::vector<double> tri_normals;
    std::vector<rthip_bvhnode> bvh_nodes;
    std::vector<rthip_tri> bvh_tris;
    std::vector<uint32_t> bvh_tri_index;
    std::vector<rthip_matop> materials;
    std::vector<rthip_noise> noises;
    std::vector<uint8_t> perlin_tables;
    std::vector<double> worley_points;
    std::vector<rthip_light> lights;
    rthip_scene_desc desc;
};

// Mesh::m_Triangles as the builders read them
struct TriView {
    const std::vector<Triangle> &t;
    size_t size() const { return t.size(); }
    bool empty() const { return t.empty(); }
    double V(uint32_t tri, int vertex, unsigned axis) const { return t[tri].Vertices[vertex].Elements[axis]; }
};

struct Flattener {
    Scene *scene;
    FlatScene &out;
    std::string &error;
    std::map<const Material *, int32_t> mat_index;
    std::map<const Noise *, int32_t> noise_index;
    std::map<const Perlin *, int32_t> table_index;
    std::map<const Mesh *, int32_t> mesh_index;

    int Unsupported(const std::string &what) {
        error = what;
        return RTHIP_E_UNSUPPORTED;
    }
    static void Rows(const Matrix4 &m, double *dst) { std::memcpy(dst, m.Elements, sizeof(double) * 12); }
    static void Put(double *dst, const Vector3 &v) { dst[0] = v.X; dst[1] = v.Y; dst[2] = v.Z; }
    static void Put(double *dst, const Color &c) { dst[0] = c.R; dst[1] = c.G; dst[2] = c.B; }

    int32_t Table(const Perlin *p) {  // Perlin::m_Seed holds rand() % 256 as reals (Perlin.cpp:16-23)
        auto it = table_index.find(p);
        if (it != table_index.end()) return it->second;
        const int32_t idx = (int32_t)(out.perlin_tables.size() / 256);
        for (size_t i = 0; i < 256; i++) out.perlin_tables.push_back((uint8_t)p->m_Seed[i]);
        return table_index[p] = idx;
    }
    int AddNoise(Noise *n, int32_t &idx) {
        auto it = noise_index.find(n);
        if (it != noise_index.end()) { idx = it->second; return 0; }
        rthip_noise d;
        std::memset(&d, 0, sizeof d);
        d.domain = n->GetDomainSize();
        d.table = -1;
        d.octaves = 8;
        d.persistance = 0.5;
        const Perlin *perlin = nullptr;
        if (dynamic_cast<CheckerBoard *>(n)) {
            d.type = RTHIP_NOISE_CHECKER;
        } else if (auto p = dynamic_cast<Perlin *>(n)) {
            d.type = RTHIP_NOISE_PERLIN;
            perlin = p;
        } else if (auto w = dynamic_cast<Worley *>(n)) {
            d.type = RTHIP_NOISE_WORLEY;
            d.divisions = w->m_Divisions;
            d.step = w->m_Step;
            d.points_base = (int32_t)(out.worley_points.size() / 3);
            for (const Vector3 &p : w->m_Points) {
                out.worley_points.push_back(p.X);
                out.worley_points.push_back(p.Y);
                out.worley_points.push_back(p.Z);
            }
        } else if (auto m = dynamic_cast<Marble *>(n)) {
            d.type = RTHIP_NOISE_MARBLE;
            d.frequency = m->m_Frequency;
            d.turbolence = m->m_Turbolence;
            perlin = m->m_Perlin.get();
        } else {
            return Unsupported("user-defined re::Noise subclass cannot run on the device");
        }
        if (perlin) {
            d.table = Table(perlin);
            d.octaves = perlin->m_Octaves;
            d.persistance = perlin->m_Persistance;
            if (d.octaves < 1 || d.octaves > 9) return Unsupported("Perlin octaves outside 1..9");
        }
        idx = (int32_t)out.noises.size();
        out.noises.push_back(d);
        noise_index[n] = idx;
        return 0;
    }
    // postfix program of the material tree (UNIFORM pushes, INTERP pops two and pushes the Lerp, Material.cpp:21-29)
    int Emit(Material *m, int32_t &root, int depth) {
        if (depth > 64) return Unsupported("material tree too deep (cycle?)");
        rthip_matop op;
        std::memset(&op, 0, sizeof op);
        if (auto u = dynamic_cast<UniformMaterial *>(m)) {
            op.op = RTHIP_MAT_UNIFORM;
            op.noise = -1;
            Put(op.color, u->m_AbsorbedColor);
            op.absorptance = u->m_Absorptance;
            op.first_op = (int32_t)out.materials.size();
            out.materials.push_back(op);
        } else if (auto im = dynamic_cast<InterpolatedMaterial *>(m)) {
            if (!im->m_Noise || !im->m_Material0 || !im->m_Material1) return Unsupported("InterpolatedMaterial with a null member");
            int32_t first = (int32_t)out.materials.size(), r0, r1, ni;
            int rc;
            if ((rc = Emit(im->m_Material0.get(), r0, depth + 1))) return rc;
            if ((rc = Emit(im->m_Material1.get(), r1, depth + 1))) return rc;
            if ((rc = AddNoise(im->m_Noise.get(), ni))) return rc;
            op.op = RTHIP_MAT_INTERP;
            op.noise = ni;
            op.first_op = first;
            out.materials.push_back(op);
        } else {
            return Unsupported("user-defined re::Material subclass cannot run on the device");
        }
        root = (int32_t)out.materials.size() - 1;
        return 0;
    }
    int AddMaterial(Material *m, int32_t &root) {
        if (!m) return Unsupported("Shape::Material is null");
        auto it = mat_index.find(m);
        if (it != mat_index.end()) { root = it->second; return 0; }
        int rc = Emit(m, root, 0);
        if (rc) return rc;
        int sp = 0, maxsp = 0;
        for (int32_t k = out.materials[root].first_op; k <= root; k++) {
            sp += out.materials[k].op == RTHIP_MAT_UNIFORM ? 1 : -1;
            if (sp > maxsp) maxsp = sp;
        }
        if (maxsp > RTHIP_MAX_MAT_STACK) return Unsupported("InterpolatedMaterial nesting deeper than the device stack");
        mat_index[m] = root;
        return 0;
    }
    int AddMesh(Mesh *mesh, int32_t &idx) {
        auto it = mesh_index.find(mesh);
        if (it != mesh_index.end()) { idx = it->second; return 0; }
        const std::vector<Triangle> &tris = mesh->m_Triangles;  // Triangle::Update has run (Mesh::Compile, Scene.cpp:386-414)
        const TriView view{tris};
        // the reference's tree over its own (already computed) bounding box, in the flat pre-order layout
        rtbuild::KdTreeFlat kd;
        {
            const BoundingBox &bb = mesh->m_BoundingBox;
            const rtbuild::Box3 root = {{bb.Min.X, bb.Min.Y, bb.Min.Z}, {bb.Max.X, bb.Max.Y, bb.Max.Z}};
            std::vector<uint32_t> all(tris.size());
            for (size_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
            rtbuild::KdBuilder<TriView> builder{view, kd};
            builder.Build(all, root, 0, rtbuild::BuildForkDepth());
        }
        rtbuild::BvhFlat bvh;
        {
            rtbuild::BvhBuilder<TriView> b{view, bvh, {}, {}, {}, {}};
            b.Run();
        }
        rthip_mesh m;
        std::memset(&m, 0, sizeof m);
        m.node_base = (uint32_t)out.kd_nodes.size();
        m.node_count = (uint32_t)kd.nodes.size();
        m.ref_base = (uint32_t)out.kd_refs.size();
        m.ref_count = (uint32_t)kd.refs.size();
        m.tri_base = (uint32_t)out.tris.size();
        m.tri_count = (uint32_t)tris.size();
        m.normal_mode = mesh->NormalMode == NormalModes::Vertex ? 1 : 0;
        m.max_depth = kd.max_depth;
        for (const auto &n : kd.nodes) out.kd_nodes.push_back({n.split, n.link, n.info});
        out.kd_bounds.insert(out.kd_bounds.end(), kd.bounds.begin(), kd.bounds.end());
        out.kd_refs.insert(out.kd_refs.end(), kd.refs.begin(), kd.refs.end());
        for (unsigned a = 0; a < 3; a++) {  // plane coordinates of every kd box on axis a (kd_degenerate_ray, rt_core.hpp)
            std::vector<double> planes = {kd.bounds[a], kd.bounds[3 + a]};
            for (const auto &n : kd.nodes)
                if (!(n.info & RTHIP_KD_LEAF) && (n.info & RTHIP_KD_AXIS_MASK) == a) planes.push_back(n.split);
            std::sort(planes.begin(), planes.end());
            planes.erase(std::unique(planes.begin(), planes.end()), planes.end());
            m.plane_base[a] = (uint32_t)out.kd_planes.size();
            m.plane_count[a] = (uint32_t)planes.size();
            out.kd_planes.insert(out.kd_planes.end(), planes.begin(), planes.end());
        }
        if (bvh.depth > RTHIP_BVH_MAX_DEPTH) return Unsupported("BVH deeper than the device stack");
        m.bvh_node_base = (uint32_t)out.bvh_nodes.size();
        m.bvh_node_count = (uint32_t)bvh.nodes.size();
        m.bvh_tri_base = (uint32_t)out.bvh_tris.size();
        m.bvh_depth = bvh.depth;
        m.abs_max = bvh.abs_max;
        for (const auto &n : bvh.nodes) {
            rthip_bvhnode d;
            std::memcpy(&d, &n, sizeof d);
            out.bvh_nodes.push_back(d);
        }
        auto pack = [](const Triangle &t) {  // what Triangle::Update caches (Scene.cpp:509-523)
            rthip_tri r;
            Put(r.v0, t.Vertices[0]);
            Put(r.face_normal, t.FaceNormal);
            Put(r.e0, t.Edges[0]);
            Put(r.e1, t.Edges[1]);
            r.d00 = t.m_D00; r.d01 = t.m_D01; r.d11 = t.m_D11; r.inv_den = t.m_InvDen;
            return r;
        };
        for (size_t i = 0; i < tris.size(); i++) {
            out.bvh_tris.push_back(pack(tris[bvh.order[i]]));
            out.bvh_tri_index.push_back(bvh.order[i]);
        }
        for (const Triangle &t : tris) {
            out.tris.push_back(pack(t));
            for (int k = 0; k < 3; k++) {
                out.tri_normals.push_back(t.Normals[k].X);
                out.tri_normals.push_back(t.Normals[k].Y);
                out.tri_normals.push_back(t.Normals[k].Z);
            }
        }
        idx = (int32_t)out.meshes.size();
        out.meshes.push_back(m);
        mesh_index[mesh] = idx;
        return 0;
    }
    // pre-order walk == visiting order of Scene::CastRayRecursive (Scene.cpp:168-237)
    int Walk(SceneNode *node) {
        auto shape = node->GetComponentOfType<Shape>();
        if (shape) {
            auto transform = node->GetComponentOfType<Transform>();
            rthip_shape s;
            std::memset(&s, 0, sizeof s);
            Matrix4 m;
            transform->GetInverseTransform(m); Rows(m, s.inv);
            transform->GetTransform(m);        Rows(m, s.fwd);
            transform->GetNormalTransform(m);  Rows(m, s.nrm);
            s.shape_id = shape->GetID();
            s.mesh = -1;
            int rc;
            if ((rc = AddMaterial(shape->Material, s.material))) return rc;
            if (dynamic_cast<Sphere *>(shape.get())) {
                s.type = RTHIP_SHAPE_SPHERE;
            } else if (auto pl = dynamic_cast<Plane *>(shape.get())) {
                s.type = RTHIP_SHAPE_PLANE;
                Put(s.plane_n, pl->Normal);
            } else if (auto me = dynamic_cast<Mesh *>(shape.get())) {
                s.type = RTHIP_SHAPE_MESH;
                if ((rc = AddMesh(me, s.mesh))) return rc;
            } else {
                return Unsupported("user-defined re::Shape subclass cannot run on the device");
            }
            out.shapes.push_back(s);
        }
        for (auto &child : node->GetChildren()) {
            int rc = Walk(child.get());
            if (rc) return rc;
        }
        return 0;
    }
    static rthip_light MakeLight(const Light &l, bool enabled) {
        rthip_light d;
        std::memset(&d, 0, sizeof d);
        d.type = l.Type == LightType::Directional ? RTHIP_LIGHT_DIRECTIONAL : (l.Type == LightType::Ambient ? RTHIP_LIGHT_AMBIENT : RTHIP_LIGHT_POINT);
        d.enabled = enabled ? 1 : 0;
        Put(d.direction, l.Direction);
        Put(d.position, l.Position);
        Put(d.color, l.Color);
        d.attenuation = l.Attenuation;
        return d;
    }
    int Run() {
        int rc = Walk(scene->GetRoot().get());
        if (rc) return rc;
        for (auto &l : scene->Lights) {
            if (!l) return Unsupported("null entry in Scene::Lights");
            out.lights.push_back(MakeLight(*l, l->Enabled));
        }
        rthip_background &bg = out.desc.background;
        std::memset(&bg, 0, sizeof bg);
        bg.type = RTHIP_BG_NONE;
        bg.table = -1;
        bg.sun_light = -1;
        if (Background *b = scene->Background.get()) {
            if (auto cb = dynamic_cast<ColorBackground *>(b)) {
                bg.type = RTHIP_BG_COLOR;
                Put(bg.color0, cb->m_Color);
            } else if (auto sky = dynamic_cast<SkyBox *>(b)) {
                bg.type = RTHIP_BG_SKYBOX;
                Put(bg.color0, sky->m_SkyColor0);
                Put(bg.color1, sky->m_SkyColor1);
                bg.table = Table(sky->m_Perlin.get());
                if (const Light *sun = sky->m_Sun.get()) {  // read live by SkyBox::GetColor (Scene.cpp:124-139)
                    for (size_t i = 0; i < scene->Lights.size(); i++)
                        if (scene->Lights[i].get() == sun) bg.sun_light = (int32_t)i;
                    if (bg.sun_light < 0) {
                        bg.sun_light = (int32_t)out.lights.size();
                        out.lights.push_back(MakeLight(*sun, false));
                    }
                }
            } else {
                return Unsupported("user-defined re::Background subclass cannot run on the device");
            }
        }
        rthip_scene_desc &d = out.desc;
        d.shapes = out.shapes.data();               d.n_shapes = (uint32_t)out.shapes.size();
        d.meshes = out.meshes.data();               d.n_meshes = (uint32_t)out.meshes.size();
        d.kd_nodes = out.kd_nodes.data();           d.n_kd_nodes = out.kd_nodes.size();
        d.kd_bounds = out.kd_bounds.data();
        d.kd_refs = out.kd_refs.data();             d.n_kd_refs = out.kd_refs.size();
        d.kd_planes = out.kd_planes.data();         d.n_kd_planes = out.kd_planes.size();
        d.tris = out.tris.data();                   d.n_tris = out.tris.size();
        d.tri_normals = out.tri_normals.data();
        d.bvh_nodes = out.bvh_nodes.data();         d.n_bvh_nodes = out.bvh_nodes.size();
        d.bvh_tris = out.bvh_tris.data();
        d.bvh_tri_index = out.bvh_tri_index.data();
        d.materials = out.materials.data();         d.n_materials = (uint32_t)out.materials.size();
        d.noises = out.noises.data();               d.n_noises = (uint32_t)out.noises.size();
        d.perlin_tables = out.perlin_tables.data(); d.n_perlin_tables = (uint32_t)(out.perlin_tables.size() / 256);
        d.worley_points = out.worley_points.data(); d.n_worley_points = out.worley_points.size() / 3;
        d.lights = out.lights.data();               d.n_lights = (uint32_t)out.lights.size();
        d.tri_verts = nullptr;   // (only read by RTHIP_UPLOAD_BUILD_KD: this flattener always brings the kd-tree)
        d.mesh_bounds = nullptr;
        return 0;
    }
};

inline int Flatten(Scene *scene, FlatScene &out, std::string &error) {
    out = FlatScene();
    std::memset(&out.desc, 0, sizeof out.desc);
    Flattener f{scene, out, error, {}, {}, {}, {}};
    return f.Run();
}

// camera basis of AbstractRaycaster::CreateScreenRay (Raytracer.cpp:102-110), once per frame
inline rthip_camera MakeCamera(Scene *scene, size_t width, size_t height, real fovY) {
    const Vector3 forward = scene->Camera.Direction.Normalized();
    const Vector3 right = Cross(forward, Vector3::Up).Normalized();
    const Vector3 up = Cross(right, forward).Normalized();
    const real h2 = std::atan(fovY / 2);
    const real w2 = h2 * (real)width / (real)height;
    rthip_camera cam;
    const Vector3 &p = scene->Camera.Position;
    cam.position[0] = p.X; cam.position[1] = p.Y; cam.position[2] = p.Z;
    cam.forward[0] = forward.X; cam.forward[1] = forward.Y; cam.forward[2] = forward.Z;
    cam.right[0] = right.X; cam.right[1] = right.Y; cam.right[2] = right.Z;
    cam.up[0] = up.X; cam.up[1] = up.Y; cam.up[2] = up.Z;
    cam.h2 = h2;
    cam.w2 = w2;
    return cam;
}

}  // namespace hipbind

// ---- the subclass of INTEGRATION.md §1, against the genuine AbstractRaycaster ---------------------------------------
class HipRaytracer : public AbstractRaycaster {
public:
    unsigned int MaxRecursion = 3;
    int Device = 0;

    HipRaytracer(unsigned int w, unsigned int h, real fovY = PI / 4.0f) : AbstractRaycaster(w, h, fovY) {}
    ~HipRaytracer() {
        {
            std::unique_lock<std::mutex> lock(m_Mutex);
            if (m_Busy && m_Ctx) rthip_interrupt(m_Ctx);
            m_Idle.wait(lock, [this] { return !m_Busy; });
        }
        if (m_Ctx) rthip_destroy(m_Ctx);
    }

    // replaces AbstractRaycaster::Render (Raytracer.cpp:34-89)
    void Render(Scene *scene, std::promise<RenderStatus> p) override {
        {
            std::unique_lock<std::mutex> lock(m_Mutex);
            m_Idle.wait(lock, [this] { return !m_Busy; });
            m_Busy = true;
        }
        m_Finished = false;  // m_Status = { false, false, 0, m_Pixels }, Raytracer.cpp:36 (the base's m_Status is private)
        m_Interrupted = false;
        std::thread([this, scene](std::promise<RenderStatus> p) {
            int rc = 0;
            std::string err;
            scene->Compile();  // Raytracer.cpp:42
            hipbind::FlatScene flat;
            rc = hipbind::Flatten(scene, flat, err);
            if (!rc && !m_Ctx) {
                std::lock_guard<std::mutex> lock(m_Mutex);
                rc = rthip_create(Device, &m_Ctx);
            }
            if (!rc) rc = rthip_upload_scene(m_Ctx, &flat.desc);
            if (!rc && !m_Interrupted) {
                const rthip_camera cam = hipbind::MakeCamera(scene, m_ViewWidth, m_ViewHeight, m_FoVY);
                rthip_params prm;
                std::memset(&prm, 0, sizeof prm);
                prm.width = (uint32_t)m_ViewWidth;
                prm.height = (uint32_t)m_ViewHeight;
                prm.max_recursion = MaxRecursion;
                prm.aa_mode = Antialiasing == AAMode::SSAA ? 1 : 0;
                prm.renderer = RTHIP_RENDER_RAYTRACER;
                prm.part_count = 1;
                rc = rthip_render(m_Ctx, &cam, &prm, m_Pixels, nullptr, nullptr);  // Raytracer.cpp:52-74
                if (rc == RTHIP_E_INTERRUPTED) {
                    m_Interrupted = true;
                    rc = 0;
                }
            }
            if (rc && err.empty()) err = rthip_last_error();
            RenderStatus status = {true, m_Interrupted.load(), m_Interrupted ? 0.f : 1.f, m_Pixels};
            m_Finished = true;  // Raytracer.cpp:76-77
            {
                std::lock_guard<std::mutex> lock(m_Mutex);
                m_Busy = false;
                m_Idle.notify_all();
            }
            if (rc)
                p.set_exception(std::make_exception_ptr(std::runtime_error("rthip error " + std::to_string(rc) + ": " + err)));
            else
                p.set_value(status);  // Raytracer.cpp:79
        }, std::move(p)).detach();  // Raytracer.cpp:84-87
    }
    void Interrupt() override {  // Raytracer.cpp:91-94
        m_Interrupted = true;
        std::lock_guard<std::mutex> lock(m_Mutex);
        if (m_Busy && m_Ctx) rthip_interrupt(m_Ctx);
    }
    RenderStatus GetStatus() override {  // Raytracer.cpp:96-99
        rthip_ctx *ctx;
        {
            std::lock_guard<std::mutex> lock(m_Mutex);
            ctx = m_Ctx;
        }
        const bool fin = m_Finished, intr = m_Interrupted;
        return {fin, intr, fin ? (intr ? 0.f : 1.f) : (ctx ? rthip_progress(ctx) : 0.f), m_Pixels};
    }

protected:
    Color Raycast(Scene *, const Ray &) override { throw std::logic_error("HipRaytracer shades whole frames on the device"); }

private:
    rthip_ctx *m_Ctx = nullptr;
    std::mutex m_Mutex;
    std::condition_variable m_Idle;
    bool m_Busy = false;
    std::atomic<bool> m_Finished{true}, m_Interrupted{false};
};

}  // namespace re

// ---- C facade for tests/test_ref_binding.py (same sb_scene handles as ref_harness.cpp's sb_* entry points) -----------
static std::map<sb_scene *, std::unique_ptr<re::hipbind::FlatScene>> g_flat;
static std::string g_bind_err;

extern "C" {

/* scene -> `const rthip_scene_desc*` through the flattener above (valid until the next call for this scene / refbind_release) */
const void *refbind_flatten(sb_scene *s) {
    s->scene->Compile();
    std::unique_ptr<re::hipbind::FlatScene> f(new re::hipbind::FlatScene);
    if (re::hipbind::Flatten(s->scene.get(), *f, g_bind_err) != 0) return nullptr;
    const void *p = &f->desc;
    g_flat[s] = std::move(f);
    return p;
}
void refbind_release(sb_scene *s) { g_flat.erase(s); }
const char *refbind_last_error(void) { return g_bind_err.c_str(); }
void refbind_camera(sb_scene *s, int width, int height, double fovy, rthip_camera *out) {
    *out = re::hipbind::MakeCamera(s->scene.get(), (size_t)width, (size_t)height, fovy > 0 ? fovy : re::PI / 4);
}
/* HipRaytracer::RenderSync through the genuine Renderer interface (needs a GPU; the container only links it) */
int refbind_render(sb_scene *s, int width, int height, double fovy, int max_recursion, int ssaa, uint32_t *pixels) {
    try {
        re::HipRaytracer rt(width, height, fovy > 0 ? fovy : re::PI / 4.0f);
        rt.MaxRecursion = max_recursion;
        rt.Antialiasing = ssaa ? re::AbstractRaycaster::AAMode::SSAA : re::AbstractRaycaster::AAMode::None;
        re::Renderer *r = &rt;  // only the reference's own virtual interface from here on
        unsigned int *px = r->RenderSync(s->scene.get());
        std::memcpy(pixels, px, sizeof(uint32_t) * (size_t)width * height);
        return 0;
    } catch (const std::exception &e) {
        g_bind_err = e.what();
        return -2;
    }
}
}
