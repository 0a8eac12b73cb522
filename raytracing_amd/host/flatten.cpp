// flatten.cpp — see flatten.hpp.  Re-lays the compiled scene out as the POD struct-of-arrays blobs
// of include/rthip.h: pre-order shape table with host-computed matrices, per-mesh kd nodes /
// bounds / leaf references / triangle records, postfix material programs, noise descriptors +
// tables, lights and background.
#include "flatten.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <map>

#include "re/threads.hpp"

namespace re {
namespace {

constexpr int kGeometryChanged = 1;  // internal: the meshes are not the ones of `previous`

struct Flattener {
    Scene *scene;
    FlatScene &out;
    std::string &error;
    std::map<const Material *, int32_t> mat_index;
    std::map<const Noise *, int32_t> noise_index;
    std::map<const Perlin *, int32_t> table_index;
    std::map<const Mesh *, int32_t> mesh_index;
    const FlatScene *previous;  // non-null: keep its geometry (see FlattenScene)

    int Unsupported(const std::string &what) {
        error = what;
        return RTHIP_E_UNSUPPORTED;
    }
    static void Rows(const Matrix4 &m, double *dst) { std::memcpy(dst, m.Elements, sizeof(double) * 12); }
    static void Put(double *dst, const Vector3 &v) { dst[0] = v.X; dst[1] = v.Y; dst[2] = v.Z; }
    static void Put(double *dst, const Color &c) { dst[0] = c.R; dst[1] = c.G; dst[2] = c.B; }

    int32_t Table(const Perlin *p) {
        auto it = table_index.find(p);
        if (it != table_index.end()) return it->second;
        int32_t idx = (int32_t)(out.perlin_tables.size() / 256);
        out.perlin_tables.insert(out.perlin_tables.end(), p->GetSeedTable().begin(), p->GetSeedTable().end());
        return table_index[p] = idx;
    }
    int AddNoise(const Noise *n, int32_t &idx) {
        auto it = noise_index.find(n);
        if (it != noise_index.end()) { idx = it->second; return 0; }
        rthip_noise d;
        std::memset(&d, 0, sizeof d);
        d.domain = n->GetDomainSize();
        d.table = -1;
        d.octaves = 8;
        d.persistance = 0.5;
        const Perlin *perlin = nullptr;
        if (dynamic_cast<const CheckerBoard *>(n)) {
            d.type = RTHIP_NOISE_CHECKER;
        } else if (auto p = dynamic_cast<const Perlin *>(n)) {
            d.type = RTHIP_NOISE_PERLIN;
            perlin = p;
        } else if (auto w = dynamic_cast<const Worley *>(n)) {
            d.type = RTHIP_NOISE_WORLEY;
            d.divisions = w->GetDivisions();
            d.step = w->GetStep();
            d.points_base = (int32_t)(out.worley_points.size() / 3);
            for (const Vector3 &p : w->GetPoints()) {
                out.worley_points.push_back(p.X);
                out.worley_points.push_back(p.Y);
                out.worley_points.push_back(p.Z);
            }
        } else if (auto m = dynamic_cast<const Marble *>(n)) {
            d.type = RTHIP_NOISE_MARBLE;
            d.frequency = m->GetFrequency();
            d.turbolence = m->GetTurbolence();
            perlin = m->GetPerlin().get();
        } else {
            return Unsupported("user-defined re::Noise subclass cannot run on the device");
        }
        if (perlin) {
            d.table = Table(perlin);
            d.octaves = perlin->GetOctaves();
            d.persistance = perlin->GetPersistance();
            if (d.octaves < 1 || d.octaves > 9) return Unsupported("Perlin octaves outside 1..9");
        }
        idx = (int32_t)out.noises.size();
        out.noises.push_back(d);
        noise_index[n] = idx;
        return 0;
    }
    // emits the postfix program of `m`; returns the index of its last op in `root`
    int Emit(const Material *m, int32_t &root, int depth) {
        if (depth > 64) return Unsupported("material tree too deep (cycle?)");
        rthip_matop op;
        std::memset(&op, 0, sizeof op);
        if (auto u = dynamic_cast<const UniformMaterial *>(m)) {
            op.op = RTHIP_MAT_UNIFORM;
            op.noise = -1;
            Put(op.color, u->GetColor());
            op.absorptance = u->GetAbsorptanceValue();
            op.first_op = (int32_t)out.materials.size();
            out.materials.push_back(op);
        } else if (auto im = dynamic_cast<const InterpolatedMaterial *>(m)) {
            if (!im->GetNoise() || !im->GetMaterial0() || !im->GetMaterial1()) return Unsupported("InterpolatedMaterial with a null member");
            int32_t first = (int32_t)out.materials.size(), r0, r1, ni;
            int rc;
            if ((rc = Emit(im->GetMaterial0().get(), r0, depth + 1))) return rc;
            if ((rc = Emit(im->GetMaterial1().get(), r1, depth + 1))) return rc;
            if ((rc = AddNoise(im->GetNoise().get(), ni))) return rc;
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
    int AddMaterial(const Material *m, int32_t &root) {
        if (!m) return Unsupported("Shape::Material is null");
        auto it = mat_index.find(m);
        if (it != mat_index.end()) { root = it->second; return 0; }
        int rc = Emit(m, root, 0);
        if (rc) return rc;
        // stack depth check (device evaluates with a bounded stack)
        int sp = 0, maxsp = 0;
        for (int32_t k = out.materials[root].first_op; k <= root; k++) {
            sp += out.materials[k].op == RTHIP_MAT_UNIFORM ? 1 : -1;
            if (sp > maxsp) maxsp = sp;
        }
        if (maxsp > RTHIP_MAX_MAT_STACK) return Unsupported("InterpolatedMaterial nesting deeper than the device stack");
        mat_index[m] = root;
        return 0;
    }
    int AddMesh(const Mesh *mesh, int32_t &idx) {
        auto it = mesh_index.find(mesh);
        if (it != mesh_index.end()) { idx = it->second; return 0; }
        if (previous) {  // the next mesh must be the one `previous` has at this position, unchanged
            const size_t k = out.meshes.size();
            if (k >= previous->mesh_signature.size() || previous->mesh_signature[k] != std::make_pair(mesh, mesh->GetRevision()))
                return kGeometryChanged;
            idx = (int32_t)k;
            out.meshes.push_back(previous->meshes[k]);
            // Mesh::NormalMode is a public field the reference reads live per hit (Scene.cpp:456); changing it does not
            // invalidate the mesh, so it is not part of the revision: carry the current value (rthip_update_scene
            // re-uploads the mesh table and accepts a difference in this one field)
            out.meshes.back().normal_mode = mesh->NormalMode == NormalModes::Vertex ? 1 : 0;
            out.mesh_signature.push_back(previous->mesh_signature[k]);
            mesh_index[mesh] = idx;
            return 0;
        }
        const KdTreeFlat &kd = mesh->GetKdTree();
        const auto &tris = mesh->GetTriangles();
        const bool device_kd = mesh->KdBuiltOnDevice() && kd.nodes.empty() && !tris.empty();  // left to RTHIP_UPLOAD_BUILD_KD
        if (device_kd) out.needs_device_kd = true;
        if (kd.nodes.empty() && !device_kd) return Unsupported("Mesh was not compiled");
        {
            const BoundingBox &bb = mesh->GetBounds();
            const double b6[6] = {bb.Min.X, bb.Min.Y, bb.Min.Z, bb.Max.X, bb.Max.Y, bb.Max.Z};
            out.mesh_bounds.insert(out.mesh_bounds.end(), b6, b6 + 6);
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
        if (out.kd_nodes.size() + kd.nodes.size() > 0xfffffff0ull || out.kd_refs.size() + kd.refs.size() > 0xfffffff0ull ||
            out.tris.size() + tris.size() > 0xfffffff0ull)
            return Unsupported("scene exceeds 2^32 kd nodes / references / triangles");
        for (const auto &n : kd.nodes) {
            if ((n.info & RTHIP_KD_LEAF) && (n.info >> RTHIP_KD_COUNT_SHIFT) >= (1u << 24) - 1) return Unsupported("kd leaf with more than 2^24 triangles");
            out.kd_nodes.push_back({n.split, n.link, n.info});
        }
        out.kd_bounds.insert(out.kd_bounds.end(), kd.bounds.begin(), kd.bounds.end());
        out.kd_refs.insert(out.kd_refs.end(), kd.refs.begin(), kd.refs.end());
        std::vector<double> axis_planes[3];
        std::future<void> sorted[3];
        for (unsigned a = 0; a < 3 && !device_kd; a++)  // plane coordinates of every kd box on axis a
            sorted[a] = std::async(std::launch::async, [&, a] {
                std::vector<double> &planes = axis_planes[a];
                planes = {kd.bounds[a], kd.bounds[3 + a]};
                for (const auto &n : kd.nodes)
                    if (!(n.info & RTHIP_KD_LEAF) && (n.info & RTHIP_KD_AXIS_MASK) == a) planes.push_back(n.split);
                std::sort(planes.begin(), planes.end());
                planes.erase(std::unique(planes.begin(), planes.end()), planes.end());
            });
        for (unsigned a = 0; a < 3 && !device_kd; a++) {
            sorted[a].get();
            const std::vector<double> &planes = axis_planes[a];
            m.plane_base[a] = (uint32_t)out.kd_planes.size();
            m.plane_count[a] = (uint32_t)planes.size();
            out.kd_planes.insert(out.kd_planes.end(), planes.begin(), planes.end());
        }
        const BvhFlat &bvh = mesh->GetBvh();
        const bool device_bvh = mesh->BvhBuiltOnDevice() && bvh.nodes.empty() && !tris.empty();  // left to RTHIP_UPLOAD_BUILD_BVH
        if (device_bvh) out.needs_device_bvh = true;
        if (bvh.depth > RTHIP_BVH_MAX_DEPTH) return Unsupported("BVH deeper than the device stack");
        if ((!device_bvh && bvh.order.size() != tris.size()) || tris.size() > RTHIP_BVH_FIRST_MASK) return Unsupported("mesh too large for the BVH leaf encoding");
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
        auto pack = [](const Triangle &t) {
            rthip_tri r;
            Put(r.v0, t.Vertices[0]);
            Put(r.face_normal, t.FaceNormal);
            Put(r.e0, t.Edges[0]);
            Put(r.e1, t.Edges[1]);
            r.d00 = t.D00(); r.d01 = t.D01(); r.d11 = t.D11(); r.inv_den = t.InvDen();
            return r;
        };
        const size_t n = tris.size(), b0 = out.bvh_tris.size(), t0 = out.tris.size();
        if (!device_bvh) {
            out.bvh_tris.resize(b0 + n);
            out.bvh_tri_index.resize(b0 + n);
        }
        out.tris.resize(t0 + n);
        out.tri_normals.resize(9 * (t0 + n));
        if (device_kd) out.tri_verts.resize(9 * (t0 + n));  // (zeros for the meshes in front that bring their tree)
        ParallelSlices(n, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                if (!device_bvh) {
                    const uint32_t ti = bvh.order[i];
                    out.bvh_tris[b0 + i] = pack(tris[ti]);
                    out.bvh_tri_index[b0 + i] = ti;
                }
                const Triangle &t = tris[i];
                out.tris[t0 + i] = pack(t);
                double *nn = &out.tri_normals[9 * (t0 + i)];
                for (int k = 0; k < 3; k++) Put(nn + 3 * k, t.Normals[k]);
                if (device_kd) {
                    double *vv = &out.tri_verts[9 * (t0 + i)];
                    for (int k = 0; k < 3; k++) Put(vv + 3 * k, t.Vertices[k]);
                }
            }
        });
        idx = (int32_t)out.meshes.size();
        out.meshes.push_back(m);
        out.mesh_signature.emplace_back(mesh, mesh->GetRevision());
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
            out.shape_nodes.push_back(node);
        }
        for (auto &child : node->GetChildren()) {
            int rc = Walk(child.get());
            if (rc) return rc;
        }
        return 0;
    }
    static int32_t LightType_(LightType t) {
        return t == LightType::Directional ? RTHIP_LIGHT_DIRECTIONAL : (t == LightType::Ambient ? RTHIP_LIGHT_AMBIENT : RTHIP_LIGHT_POINT);
    }
    static rthip_light MakeLight(const Light &l, bool enabled) {
        rthip_light d;
        std::memset(&d, 0, sizeof d);
        d.type = LightType_(l.Type);
        d.enabled = enabled ? 1 : 0;
        Put(d.direction, l.Direction);
        Put(d.position, l.Position);
        Put(d.color, l.Color);
        d.attenuation = l.Attenuation;
        return d;
    }
    // triangles of every distinct mesh below `node` (the geometry arrays are reserved once: five 2 M-triangle meshes appended to
    // vectors that regrow copied gigabytes)
    static void CountTriangles(SceneNode *node, std::map<const Mesh *, size_t> &seen) {
        if (auto shape = node->GetComponentOfType<Shape>())
            if (auto mesh = dynamic_cast<Mesh *>(shape.get())) seen[mesh] = mesh->GetTriangles().size();
        for (auto &child : node->GetChildren()) CountTriangles(child.get(), seen);
    }
    int Run() {
        if (!previous) {
            std::map<const Mesh *, size_t> seen;
            CountTriangles(scene->GetRoot().get(), seen);
            size_t total = 0, total_host_bvh = 0;
            bool any_device_kd = false;
            for (auto &kv : seen) {
                total += kv.second;
                if (!kv.first->BvhBuiltOnDevice()) total_host_bvh += kv.second;
                any_device_kd = any_device_kd || (kv.first->KdBuiltOnDevice() && kv.second);
            }
            out.tris.reserve(total);
            out.tri_normals.reserve(9 * total);
            if (any_device_kd) out.tri_verts.reserve(9 * total);
            out.bvh_tris.reserve(total_host_bvh);
            out.bvh_tri_index.reserve(total_host_bvh);
        }
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
                Put(bg.color0, cb->GetBackgroundColor());
            } else if (auto sky = dynamic_cast<SkyBox *>(b)) {
                bg.type = RTHIP_BG_SKYBOX;
                Put(bg.color0, sky->GetSkyColor0());
                Put(bg.color1, sky->GetSkyColor1());
                bg.table = Table(sky->GetPerlin().get());
                if (const Light *sun = sky->GetSun().get()) {
                    // the sky reads its Light live (Scene.cpp:124-139); it need not be in Scene::Lights
                    for (size_t i = 0; i < scene->Lights.size(); i++)
                        if (scene->Lights[i].get() == sun) bg.sun_light = (int32_t)i;
                    if (bg.sun_light < 0) {
                        bg.sun_light = (int32_t)out.lights.size();
                        out.lights.push_back(MakeLight(*sun, false));  // disabled: sky only, casts no light
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
        d.tri_verts = nullptr;
        d.mesh_bounds = nullptr;
        if (out.needs_device_kd) {
            out.tri_verts.resize(9 * out.tris.size());
            d.tri_verts = out.tri_verts.data();
            d.mesh_bounds = out.mesh_bounds.data();
        }
        if (previous) {
            if (out.meshes.size() != previous->meshes.size()) return kGeometryChanged;
            out.geometry_kept = true;
            out.needs_device_bvh = previous->needs_device_bvh;
            out.needs_device_kd = previous->needs_device_kd;
            d.tri_verts = nullptr;
            d.mesh_bounds = nullptr;
            const rthip_scene_desc &pd = previous->desc;
            d.kd_nodes = nullptr; d.kd_bounds = nullptr; d.kd_refs = nullptr; d.kd_planes = nullptr; d.tris = nullptr;
            d.tri_normals = nullptr; d.bvh_nodes = nullptr; d.bvh_tris = nullptr; d.bvh_tri_index = nullptr;
            d.n_kd_nodes = pd.n_kd_nodes; d.n_kd_refs = pd.n_kd_refs; d.n_kd_planes = pd.n_kd_planes; d.n_tris = pd.n_tris;
            d.n_bvh_nodes = pd.n_bvh_nodes;
        }
        return 0;
    }
};

}  // namespace

int FlattenScene(Scene *scene, FlatScene &out, std::string &error, const FlatScene *previous) {
    if (previous) {
        out = FlatScene();
        std::memset(&out.desc, 0, sizeof out.desc);
        Flattener f{scene, out, error, {}, {}, {}, {}, previous};
        const int rc = f.Run();
        if (rc != kGeometryChanged) return rc;
    }
    out = FlatScene();
    std::memset(&out.desc, 0, sizeof out.desc);
    Flattener f{scene, out, error, {}, {}, {}, {}, nullptr};
    return f.Run();
}

void ForceHostBvhEverywhere(SceneNode *node) {
    if (auto shape = node->GetComponentOfType<Shape>())
        if (auto mesh = dynamic_cast<Mesh *>(shape.get())) {
            mesh->BuildHostBvh();
            mesh->BuildHostKd();
        }
    for (auto &child : node->GetChildren()) ForceHostBvhEverywhere(child.get());
}

void FlatScene::ReleaseGeometry() {
    std::vector<rthip_kdnode>().swap(kd_nodes);
    std::vector<double>().swap(kd_bounds);
    std::vector<uint32_t>().swap(kd_refs);
    std::vector<double>().swap(kd_planes);
    std::vector<rthip_tri>().swap(tris);
    std::vector<double>().swap(tri_normals);
    std::vector<rthip_bvhnode>().swap(bvh_nodes);
    std::vector<rthip_tri>().swap(bvh_tris);
    std::vector<uint32_t>().swap(bvh_tri_index);
    std::vector<double>().swap(tri_verts);
    rthip_scene_desc &d = desc;  // counts stay: they describe what the device holds
    d.kd_nodes = nullptr; d.kd_bounds = nullptr; d.kd_refs = nullptr; d.kd_planes = nullptr; d.tris = nullptr;
    d.tri_normals = nullptr; d.bvh_nodes = nullptr; d.bvh_tris = nullptr; d.bvh_tri_index = nullptr; d.tri_verts = nullptr;
}

void MakeCamera(const Scene *scene, size_t width, size_t height, real fovY, rthip_camera &cam) {
    MakeCamera(scene->Camera.Position, scene->Camera.Direction, width, height, fovY, cam);
}
void MakeCamera(const Vector3 &position, const Vector3 &direction, size_t width, size_t height, real fovY, rthip_camera &cam) {
    const Vector3 forward = direction.Normalized();
    const Vector3 right = Cross(forward, Vector3::Up).Normalized();
    const Vector3 up = Cross(right, forward).Normalized();
    const real h2 = std::atan(fovY / 2);  // atan, not tan (Raytracer.cpp:109)
    const real w2 = h2 * (real)width / (real)height;
    const Vector3 &p = position;
    cam.position[0] = p.X; cam.position[1] = p.Y; cam.position[2] = p.Z;
    cam.forward[0] = forward.X; cam.forward[1] = forward.Y; cam.forward[2] = forward.Z;
    cam.right[0] = right.X; cam.right[1] = right.Y; cam.right[2] = right.Z;
    cam.up[0] = up.X; cam.up[1] = up.Y; cam.up[2] = up.Z;
    cam.h2 = h2;
    cam.w2 = w2;
}

}  // namespace re
