// sb_host.cpp — C facade over the re:: host mirror for the Python tests and bench.py: the same
// sb_* entry points the oracle exports (oracle/rt_oracle.h documents them), implemented with the
// PRODUCT classes: scenes are built through re::Scene / SceneNode / Shape / Material / Noise,
// rendered by re::HipRaytracer::RenderSync, and the probe calls go through the rthip_* C-ABI.
// Nothing here evaluates rays on the host.
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "flatten.hpp"
#include "scene_io.hpp"
#include "re/re.hpp"
#include "rthip.h"

using namespace re;

struct sb_scene {
    std::shared_ptr<Scene> scene = std::make_shared<Scene>();
    std::vector<SceneNode *> nodes;
    std::vector<std::shared_ptr<Material>> mats;
    std::vector<std::shared_ptr<Noise>> noises;
    std::vector<std::shared_ptr<Light>> lights;
    bool compiled = false;
    std::unique_ptr<FlatScene> flat;  // valid while compiled
    rthip_ctx *ctx = nullptr;         // probe context (lazy)
    bool uploaded = false;
    int device = 0;
    int use_kd = 0;
    std::vector<int> devices;  // more than one: HipRaytracer::Devices (the frame is split over these GPUs)
    HipRenderStats last_stats;
    ~sb_scene() { if (ctx) rthip_destroy(ctx); }
};

static thread_local std::string g_host_err;
static Vector3 V3(const double *p) { return Vector3(p[0], p[1], p[2]); }
static void touch(sb_scene *s) { s->compiled = false; s->uploaded = false; }
static int ensure_flat(sb_scene *s) {
    if (s->compiled) return 0;
    s->scene->Compile();
    s->flat.reset(new FlatScene);
    int rc = FlattenScene(s->scene.get(), *s->flat, g_host_err);
    if (rc) return rc;
    s->compiled = true;
    return 0;
}
static int ensure_device(sb_scene *s) {
    int rc = ensure_flat(s);
    if (rc) return rc;
    if (!s->ctx && (rc = rthip_create(s->device, &s->ctx))) { g_host_err = rthip_last_error(); return rc; }
    if (!s->uploaded) {
        rc = rthip_upload_scene_ex(s->ctx, &s->flat->desc, UploadFlags(*s->flat));
        if (rc == RTHIP_E_UNSUPPORTED && (s->flat->needs_device_bvh || s->flat->needs_device_kd)) {
            // the device would not build a BVH (deeper than its traversal stacks, failed validation): build them on the host
            // after all, as HipRaytracer::RenderThread does
            ForceHostBvhEverywhere(s->scene->GetRoot().get());
            s->compiled = false;
            if ((rc = ensure_flat(s))) return rc;
            rc = rthip_upload_scene_ex(s->ctx, &s->flat->desc, UploadFlags(*s->flat));
        }
        if (rc) { g_host_err = rthip_last_error(); return rc; }
        s->uploaded = true;
    }
    return 0;
}

struct sb_render_params {
    int width, height;
    double fovy;
    int max_recursion, aa_mode, num_threads, renderer;
};

extern "C" {

const char *sb_impl_name(void) { return "raytracing_amd-host"; }
const char *rehost_last_error(void) { return g_host_err.c_str(); }

sb_scene *sb_scene_new(void) {
    auto *s = new sb_scene;
    s->nodes.push_back(s->scene->GetRoot().get());
    return s;
}
void sb_scene_free(sb_scene *s) { delete s; }

// scene inputs of the reference's Sandbox (scene_io.hpp): a scene from a script of the reXxx call subset
sb_scene *rehost_scene_from_script(const char *code, const char *res_dir) {
    try {
        sbio::ScriptScene sc;
        sbio::RunSceneScript(code ? code : "", res_dir ? res_dir : "", sc);
        sb_scene *s = new sb_scene;
        s->scene = sc.Scene;
        s->nodes = sc.ShapeNodes;
        s->mats = sc.Materials;
        s->noises = sc.Noises;
        s->lights = sc.Lights;
        return s;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return nullptr;
    }
}
// scene `index` of a scenes.xml store: name and code copied into the caller's buffers; returns the number of scenes, -1 on error
int rehost_scenes_xml(const char *path, int index, char *name, int name_cap, char *code, int code_cap) {
    try {
        auto scenes = sbio::ReadScenesXml(path);
        if (index >= 0 && index < (int)scenes.size()) {
            if (name && name_cap > 0) { std::strncpy(name, scenes[index].first.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
            if (code && code_cap > 0) { std::strncpy(code, scenes[index].second.c_str(), code_cap - 1); code[code_cap - 1] = 0; }
        }
        return (int)scenes.size();
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}
// Wavefront OBJ group -> triangles (first three vertices / normals of every face, as reObjMesh takes them);
// returns the triangle count (call with null buffers first), -1 on error
int rehost_load_obj(const char *path, const char *group, double *verts9, double *normals9, int cap_tris) {
    try {
        auto data = sbio::LoadWavefront(path);
        auto g = data.find(group);
        if (g == data.end()) { g_host_err = std::string("Invalid obj group: ") + group; return -1; }
        int n = 0;
        for (const auto &f : g->second) {
            if (f.Vertices.size() < 3) { g_host_err = "obj face with fewer than 3 vertices"; return -1; }
            if (verts9 && normals9 && n < cap_tris)
                for (int k = 0; k < 3; k++) {
                    verts9[9 * n + 3 * k] = f.Vertices[k].X; verts9[9 * n + 3 * k + 1] = f.Vertices[k].Y; verts9[9 * n + 3 * k + 2] = f.Vertices[k].Z;
                    normals9[9 * n + 3 * k] = f.Normals[k].X; normals9[9 * n + 3 * k + 1] = f.Normals[k].Y; normals9[9 * n + 3 * k + 2] = f.Normals[k].Z;
                }
            n++;
        }
        return n;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}
int rehost_write_ppm(const char *path, const uint32_t *pixels, int width, int height) {
    try {
        sbio::WritePPM(path, pixels, (size_t)width, (size_t)height);
        return 0;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -1;
    }
}
void sb_camera(sb_scene *s, const double pos[3], const double dir[3]) {
    s->scene->Camera.Position = V3(pos);
    s->scene->Camera.Direction = V3(dir);
}
int sb_mat_uniform(sb_scene *s, unsigned rgb, double absorptance) {
    s->mats.push_back(std::make_shared<UniformMaterial>(Color(rgb), absorptance));
    touch(s);
    return (int)s->mats.size() - 1;
}
int sb_mat_interp(sb_scene *s, int noise, int m0, int m1) {
    if (noise < 0 || noise >= (int)s->noises.size() || m0 < 0 || m0 >= (int)s->mats.size() || m1 < 0 || m1 >= (int)s->mats.size()) return -1;
    s->mats.push_back(std::make_shared<InterpolatedMaterial>(s->noises[noise], s->mats[m0], s->mats[m1]));
    touch(s);
    return (int)s->mats.size() - 1;
}
int sb_noise_checker(sb_scene *s, double domain) {
    s->noises.push_back(std::make_shared<CheckerBoard>(domain));
    return (int)s->noises.size() - 1;
}
int sb_noise_perlin(sb_scene *s, double domain, const uint8_t *table256) {
    auto p = std::make_shared<Perlin>(domain);
    if (table256) p->SetSeedTable(table256);
    s->noises.push_back(p);
    return (int)s->noises.size() - 1;
}
int sb_noise_worley(sb_scene *s, double domain, int divisions, const double *points) {
    if (divisions < 1) return -1;
    auto w = std::make_shared<Worley>(domain, divisions);
    if (points) w->SetPoints(points);
    s->noises.push_back(w);
    return (int)s->noises.size() - 1;
}
int sb_noise_marble(sb_scene *s, double domain, double freq, double turb, const uint8_t *table256) {
    auto m = std::make_shared<Marble>(domain, freq, turb);
    if (table256) m->GetPerlin()->SetSeedTable(table256);
    s->noises.push_back(m);
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
    touch(s);
    return (int)s->nodes.size() - 1;
}
static bool node_ok(sb_scene *s, int node, int mat) {
    return node >= 0 && node < (int)s->nodes.size() && mat >= 0 && mat < (int)s->mats.size() &&
           s->nodes[node]->GetComponentOfType<Shape>() == nullptr;
}
int sb_sphere(sb_scene *s, int node, int mat) {
    if (!node_ok(s, node, mat)) return -1;
    s->nodes[node]->AddComponent<Sphere>()->Material = s->mats[mat].get();
    touch(s);
    return 0;
}
int sb_plane(sb_scene *s, int node, int mat, const double n[3]) {
    if (!node_ok(s, node, mat)) return -1;
    auto pl = s->nodes[node]->AddComponent<Plane>();
    pl->Normal = V3(n);
    pl->Material = s->mats[mat].get();
    touch(s);
    return 0;
}
int sb_mesh(sb_scene *s, int node, int mat, int normal_mode, int ntris, const double *verts, const double *normals) {
    if (!node_ok(s, node, mat) || ntris < 0 || (ntris && !verts)) return -1;
    auto mesh = s->nodes[node]->AddComponent<Mesh>();
    mesh->ReserveTriangles((size_t)ntris);
    for (int i = 0; i < ntris; i++) {
        Triangle &t = mesh->AddTriangle();
        t.Vertices = {V3(verts + i * 9), V3(verts + i * 9 + 3), V3(verts + i * 9 + 6)};
        if (normals) t.Normals = {V3(normals + i * 9), V3(normals + i * 9 + 3), V3(normals + i * 9 + 6)};
    }
    mesh->NormalMode = normal_mode ? NormalModes::Vertex : NormalModes::Face;
    mesh->Material = s->mats[mat].get();
    touch(s);
    return 0;
}
int sb_light(sb_scene *s, int type, unsigned rgb, const double dir[3], const double pos[3], double attenuation, int enabled) {
    auto l = std::make_shared<Light>();
    l->Type = type == 0 ? LightType::Directional : (type == 1 ? LightType::Ambient : LightType::Point);
    l->Color = Color(rgb);
    l->Direction = V3(dir);
    l->Position = V3(pos);
    l->Attenuation = attenuation;
    l->Enabled = enabled != 0;
    s->scene->Lights.push_back(l);
    s->lights.push_back(l);
    touch(s);
    return (int)s->lights.size() - 1;
}
void sb_bg_color(sb_scene *s, unsigned rgb) {
    s->scene->Background = std::make_shared<ColorBackground>(Color(rgb));
    touch(s);
}
void sb_bg_skybox(sb_scene *s, unsigned rgb0, unsigned rgb1, int sun_light, const uint8_t *table256) {
    std::shared_ptr<Light> sun = (sun_light >= 0 && sun_light < (int)s->lights.size()) ? s->lights[sun_light] : nullptr;
    auto sky = std::make_shared<SkyBox>(Color(rgb0), Color(rgb1), sun);
    if (table256) sky->GetPerlin()->SetSeedTable(table256);
    s->scene->Background = sky;
    touch(s);
}
int sb_get_perlin_table(sb_scene *s, int noise, uint8_t *out256) {
    if (noise < 0 || noise >= (int)s->noises.size()) return -1;
    const Perlin *p = dynamic_cast<const Perlin *>(s->noises[noise].get());
    if (auto m = dynamic_cast<const Marble *>(s->noises[noise].get())) p = m->GetPerlin().get();
    if (!p) return -1;
    std::memcpy(out256, p->GetSeedTable().data(), 256);
    return 0;
}
int sb_get_skybox_table(sb_scene *s, uint8_t *out256) {
    auto sky = dynamic_cast<SkyBox *>(s->scene->Background.get());
    if (!sky) return -1;
    std::memcpy(out256, sky->GetPerlin()->GetSeedTable().data(), 256);
    return 0;
}
int sb_get_worley_points(sb_scene *s, int noise, double *out) {
    if (noise < 0 || noise >= (int)s->noises.size()) return -1;
    auto w = dynamic_cast<const Worley *>(s->noises[noise].get());
    if (!w) return -1;
    size_t i = 0;
    for (const Vector3 &p : w->GetPoints()) { out[i++] = p.X; out[i++] = p.Y; out[i++] = p.Z; }
    return 0;
}

// ---- additive: product-only entry points ---------------------------------------------------
// compiled + flattened scene as a `const rthip_scene_desc*` (valid until the scene changes / is freed)
const void *rehost_flatten(sb_scene *s) { return ensure_flat(s) ? nullptr : (const void *)&s->flat->desc; }
int rehost_camera(sb_scene *s, int width, int height, double fovy, rthip_camera *out) {
    MakeCamera(s->scene.get(), (size_t)width, (size_t)height, fovy > 0 ? fovy : PI / 4, *out);
    return 0;
}
void rehost_set_device(sb_scene *s, int device) { s->device = device; }
void rehost_set_devices(sb_scene *s, const int *devices, int n) { s->devices.assign(devices, devices + (n > 0 ? n : 0)); }
void rehost_set_accel(sb_scene *s, int use_reference_kd) { s->use_kd = use_reference_kd; }
int rehost_shape_count(sb_scene *s) { return ensure_flat(s) ? -1 : (int)s->flat->shapes.size(); }
// shape ordinal of a node id (or -1)
int rehost_shape_ordinal(sb_scene *s, int node) {
    if (ensure_flat(s) || node < 0 || node >= (int)s->nodes.size()) return -1;
    for (size_t i = 0; i < s->flat->shape_nodes.size(); i++)
        if (s->flat->shape_nodes[i] == s->nodes[node]) return (int)i;
    return -1;
}
int rehost_last_stats(sb_scene *s, double *out8) {
    const HipRenderStats &t = s->last_stats;
    out8[0] = (double)t.RaysPrimary; out8[1] = (double)t.RaysShadow; out8[2] = (double)t.RaysReflect; out8[3] = (double)t.Pixels;
    out8[4] = t.DeviceMs; out8[5] = t.FlattenMs; out8[6] = t.UploadMs; out8[7] = 0;
    return 0;
}


// Whole frame through the drop-in class: HipRaytracer / HipDebugRaycaster ::RenderSync.
// Primary-hit ids (optional) come from a second, id-only device pass through rthip_render.
int sb_render(sb_scene *s, const sb_render_params *p, uint32_t *pixels, int32_t *shape_ord, int32_t *tri_idx,
              uint64_t *ray_count, const int32_t *sub_x, const int32_t *sub_y, int n_sub) {
    try {
        real fovy = p->fovy > 0 ? p->fovy : PI / 4;
        std::unique_ptr<HipRaytracer> rt;
        if (p->renderer == 0) {
            rt.reset(new HipRaytracer(p->width, p->height, fovy));
        } else {
            auto *d = new HipDebugRaycaster(p->width, p->height, fovy);
            d->Mode = p->renderer == 1 ? HipDebugRaycaster::Modes::Normal : HipDebugRaycaster::Modes::Color;
            rt.reset(d);
        }
        rt->MaxRecursion = p->max_recursion;
        rt->NumThreads = p->num_threads;
        rt->Device = s->device;
        rt->Devices = s->devices;
        rt->UseReferenceKdTree = s->use_kd != 0;
        rt->Antialiasing = p->aa_mode == 1 ? AbstractRaycaster::AAMode::SSAA : AbstractRaycaster::AAMode::None;
        unsigned int *px = rt->RenderSync(s->scene.get());
        s->last_stats = rt->GetLastStats();
        s->compiled = false;  // RenderSync recompiled the scene; keep our cached flat copy coherent
        s->uploaded = false;
        const size_t W = p->width, H = p->height;
        std::vector<int32_t> sh, tr;
        if (shape_ord || tri_idx) {
            int rc = ensure_device(s);
            if (rc) return rc;
            rthip_camera cam;
            MakeCamera(s->scene.get(), W, H, fovy, cam);
            rthip_params prm;
            std::memset(&prm, 0, sizeof prm);
            prm.width = (uint32_t)W; prm.height = (uint32_t)H; prm.max_recursion = 0; prm.aa_mode = 0;
            prm.renderer = RTHIP_RENDER_DEBUG_NORMAL; prm.part_index = 0; prm.part_count = 1;
            prm.accel = s->use_kd;
            std::vector<uint32_t> tmp(W * H);
            sh.resize(W * H); tr.resize(W * H);
            rc = rthip_render(s->ctx, &cam, &prm, tmp.data(), sh.data(), tr.data());
            if (rc) { g_host_err = rthip_last_error(); return rc; }
        }
        if (n_sub > 0) {
            for (int i = 0; i < n_sub; i++) {
                size_t idx = (size_t)sub_y[i] * W + (size_t)sub_x[i];
                pixels[i] = px[idx];
                if (shape_ord) shape_ord[i] = sh[idx];
                if (tri_idx) tri_idx[i] = tr[idx];
            }
        } else {
            std::memcpy(pixels, px, sizeof(uint32_t) * W * H);
            if (shape_ord) std::memcpy(shape_ord, sh.data(), sizeof(int32_t) * W * H);
            if (tri_idx) std::memcpy(tri_idx, tr.data(), sizeof(int32_t) * W * H);
        }
        if (ray_count) *ray_count = s->last_stats.RaysPrimary + s->last_stats.RaysShadow + s->last_stats.RaysReflect;
        return 0;
    } catch (const DeviceError &e) {
        g_host_err = e.what();
        return e.Code;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RTHIP_E_INVALID;
    }
}

// Two frames through ONE renderer, as the Sandbox preview does frame after frame (Sandbox.cpp:359): after the first
// frame every node directly under the root is moved by dx along x and the camera by dx / 2; the second frame and
// whether it kept the device geometry (HipRenderStats::GeometryKept) are returned.  The scene stays moved.
int rehost_render_moved(sb_scene *s, const sb_render_params *p, double dx, uint32_t *pixels_second, int *geometry_kept) {
    try {
        real fovy = p->fovy > 0 ? p->fovy : PI / 4;
        HipRaytracer rt(p->width, p->height, fovy);
        rt.MaxRecursion = p->max_recursion;
        rt.Device = s->device;
        rt.UseReferenceKdTree = s->use_kd != 0;
        rt.RenderSync(s->scene.get());
        for (auto &child : s->scene->GetRoot()->GetChildren()) child->GetComponentOfType<Transform>()->Position.X += dx;
        s->scene->Camera.Position.X += dx / 2;
        unsigned int *px = rt.RenderSync(s->scene.get());
        std::memcpy(pixels_second, px, sizeof(uint32_t) * (size_t)p->width * p->height);
        *geometry_kept = rt.GetLastStats().GeometryKept ? 1 : 0;
        s->last_stats = rt.GetLastStats();
        s->compiled = false;
        s->uploaded = false;
        return 0;
    } catch (const DeviceError &e) {
        g_host_err = e.what();
        return e.Code;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RTHIP_E_INVALID;
    }
}

// A queued camera path through ONE renderer (HipRaytracer::RenderPathSync): n poses (position xyz, direction xyz each) over the
// scene as it is; frames come back one after the other.  `devices` as for sb_render (rehost_set_devices).
int rehost_render_path(sb_scene *s, const sb_render_params *p, int n, const double *poses6, uint32_t *frames) {
    try {
        HipRaytracer rt(p->width, p->height, p->fovy > 0 ? p->fovy : PI / 4);
        rt.MaxRecursion = p->max_recursion;
        rt.Device = s->device;
        rt.Devices = s->devices;
        rt.UseReferenceKdTree = s->use_kd != 0;
        rt.Antialiasing = p->aa_mode == 1 ? AbstractRaycaster::AAMode::SSAA : AbstractRaycaster::AAMode::None;
        std::vector<CameraPose> path((size_t)n);
        for (int i = 0; i < n; i++) path[i] = {V3(poses6 + 6 * i), V3(poses6 + 6 * i + 3)};
        const unsigned int *px = rt.RenderPathSync(s->scene.get(), path);
        s->last_stats = rt.GetLastStats();
        s->compiled = false;
        s->uploaded = false;
        if (!px) { g_host_err = "interrupted"; return RTHIP_E_INTERRUPTED; }
        std::memcpy(frames, px, sizeof(uint32_t) * (size_t)n * p->width * p->height);
        return 0;
    } catch (const DeviceError &e) {
        g_host_err = e.what();
        return e.Code;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RTHIP_E_INVALID;
    }
}

// HipRaytracer::Raycast for n rays, one call each (origin xyz, direction xyz per ray) -> n unpacked colours
int rehost_raycast(sb_scene *s, uint32_t max_recursion, int n, const double *rays6, double *rgb) {
    try {
        HipRaytracer rt(64, 64);
        rt.MaxRecursion = max_recursion;
        rt.Device = s->device;
        rt.UseReferenceKdTree = s->use_kd != 0;
        for (int i = 0; i < n; i++) {
            Ray r;
            r.Origin = V3(rays6 + 6 * i);
            r.Direction = V3(rays6 + 6 * i + 3);
            const Color c = rt.RaycastSync(s->scene.get(), r);
            rgb[3 * i] = c.R; rgb[3 * i + 1] = c.G; rgb[3 * i + 2] = c.B;
        }
        s->compiled = false;
        s->uploaded = false;
        return 0;
    } catch (const DeviceError &e) {
        g_host_err = e.what();
        return e.Code;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RTHIP_E_INVALID;
    }
}

// Two frames through ONE renderer with Mesh::NormalMode of every mesh flipped (Face <-> Vertex) in between: a public field
// the reference reads live per hit (Scene.cpp:456) and whose change does not invalidate the mesh.  Returns the second
// frame and whether the geometry stayed on the device.  The scene keeps the flipped modes.
int rehost_render_flip_normal_mode(sb_scene *s, const sb_render_params *p, uint32_t *pixels_second, int *geometry_kept) {
    try {
        HipRaytracer rt(p->width, p->height, p->fovy > 0 ? p->fovy : PI / 4);
        rt.MaxRecursion = p->max_recursion;
        rt.Device = s->device;
        rt.UseReferenceKdTree = s->use_kd != 0;
        rt.RenderSync(s->scene.get());
        std::function<void(SceneNode *)> flip = [&](SceneNode *n) {
            if (auto sh = n->GetComponentOfType<Shape>())
                if (auto m = dynamic_cast<Mesh *>(sh.get())) m->NormalMode = m->NormalMode == NormalModes::Vertex ? NormalModes::Face : NormalModes::Vertex;
            for (auto &c : n->GetChildren()) flip(c.get());
        };
        flip(s->scene->GetRoot().get());
        unsigned int *px = rt.RenderSync(s->scene.get());
        std::memcpy(pixels_second, px, sizeof(uint32_t) * (size_t)p->width * p->height);
        *geometry_kept = rt.GetLastStats().GeometryKept ? 1 : 0;
        s->last_stats = rt.GetLastStats();
        s->compiled = false;
        s->uploaded = false;
        return 0;
    } catch (const DeviceError &e) {
        g_host_err = e.what();
        return e.Code;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RTHIP_E_INVALID;
    }
}

// ---- asynchronous contract of Renderer::Render / GetStatus / Interrupt (Raytracer.h:36-43) ----
struct rehost_job {
    std::unique_ptr<HipRaytracer> rt;
    std::future<Renderer::RenderStatus> fut;
    int width = 0, height = 0;
};
rehost_job *rehost_render_async(sb_scene *s, const sb_render_params *p) {
    auto *j = new rehost_job;
    j->rt.reset(new HipRaytracer(p->width, p->height, p->fovy > 0 ? p->fovy : PI / 4));
    j->rt->MaxRecursion = p->max_recursion;
    j->rt->Device = s->device;
    j->rt->Antialiasing = p->aa_mode == 1 ? AbstractRaycaster::AAMode::SSAA : AbstractRaycaster::AAMode::None;
    j->width = p->width;
    j->height = p->height;
    std::promise<Renderer::RenderStatus> pr;
    j->fut = pr.get_future();
    s->compiled = false;
    s->uploaded = false;
    j->rt->Render(s->scene.get(), std::move(pr));  // returns immediately
    return j;
}
// out4: Finished, Interruped, Percent, (Pixels != null)
void rehost_job_status(rehost_job *j, float *out4) {
    Renderer::RenderStatus st = j->rt->GetStatus();
    out4[0] = st.Finished; out4[1] = st.Interruped; out4[2] = st.Percent; out4[3] = st.Pixels != nullptr;
}
void rehost_job_interrupt(rehost_job *j) { j->rt->Interrupt(); }
// waits for the promise; returns 0 and copies the pixels, or the DeviceError code. out3: Finished, Interruped, Percent
int rehost_job_wait(rehost_job *j, uint32_t *pixels, float *out3) {
    try {
        Renderer::RenderStatus st = j->fut.get();
        if (out3) { out3[0] = st.Finished; out3[1] = st.Interruped; out3[2] = st.Percent; }
        if (pixels && st.Pixels) std::memcpy(pixels, st.Pixels, sizeof(uint32_t) * (size_t)j->width * j->height);
        return 0;
    } catch (const DeviceError &e) {
        g_host_err = e.what();
        return e.Code;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RTHIP_E_INVALID;
    }
}
void rehost_job_free(rehost_job *j) { delete j; }

int sb_cast_rays(sb_scene *s, int n, const double *o, const double *d, int32_t *out_hit, double *out_point,
                 double *out_local, double *out_normal, int32_t *out_shape, int32_t *out_tri) {
    int rc = ensure_device(s);
    if (rc) return rc;
    std::vector<double> pt(3 * (size_t)n), lp(3 * (size_t)n), nr(3 * (size_t)n);
    std::vector<int32_t> sh(n), tr(n);
    rc = rthip_probe_cast(s->ctx, (uint32_t)n, o, d, out_hit, pt.data(), lp.data(), nr.data(), sh.data(), tr.data());
    if (rc) { g_host_err = rthip_last_error(); return rc; }
    if (out_point) std::memcpy(out_point, pt.data(), sizeof(double) * 3 * n);
    if (out_local) std::memcpy(out_local, lp.data(), sizeof(double) * 3 * n);
    if (out_normal) std::memcpy(out_normal, nr.data(), sizeof(double) * 3 * n);
    if (out_shape) std::memcpy(out_shape, sh.data(), sizeof(int32_t) * n);
    if (out_tri) std::memcpy(out_tri, tr.data(), sizeof(int32_t) * n);
    return 0;
}
// a probe of a noise / material that nothing in the scene references: the flattened scene does not contain it (the tests skip
// exactly this code and nothing else)
static const int kNotInScene = -100;
int sb_noise_sample(sb_scene *s, int noise, int n, const double *p, double *out) {
    // noises that no material references are not part of the flattened scene: attach a probe material
    if (noise < 0 || noise >= (int)s->noises.size()) return -1;
    int rc = ensure_device(s);
    if (rc) return rc;
    // find the flattened index of this noise by evaluating through a temporary material is overkill:
    // rebuild the flattener's mapping order instead (noises appear in first-use order)
    // -> simplest exact route: a one-node scene is not needed; search by descriptor identity.
    const Noise *target = s->noises[noise].get();
    int32_t flat_index = -1;
    {
        // re-run the discovery order: shapes pre-order -> material postfix -> noise first use
        std::vector<const Noise *> order;
        std::vector<const Material *> seen;
        std::function<void(const Material *)> visit = [&](const Material *m) {
            if (auto im = dynamic_cast<const InterpolatedMaterial *>(m)) {
                visit(im->GetMaterial0().get());
                visit(im->GetMaterial1().get());
                const Noise *nz = im->GetNoise().get();
                bool have = false;
                for (auto q : order) have |= q == nz;
                if (!have) order.push_back(nz);
            }
        };
        for (SceneNode *node : s->flat->shape_nodes) {
            const Material *m = node->GetComponentOfType<Shape>()->Material;
            bool have = false;
            for (auto q : seen) have |= q == m;
            if (have) continue;
            seen.push_back(m);
            visit(m);
        }
        for (size_t i = 0; i < order.size(); i++)
            if (order[i] == target) flat_index = (int32_t)i;
    }
    if (flat_index < 0) { g_host_err = "noise is not referenced by any shape material of the scene"; return kNotInScene; }
    rc = rthip_probe_noise(s->ctx, flat_index, (uint32_t)n, p, out);
    if (rc) g_host_err = rthip_last_error();
    return rc;
}
int sb_material_eval(sb_scene *s, int mat, int n, const double *p, double *out_color, double *out_absorptance) {
    if (mat < 0 || mat >= (int)s->mats.size()) return -1;
    int rc = ensure_device(s);
    if (rc) return rc;
    // root op of this material = the `material` field of a shape that uses it
    int32_t root = -1;
    for (size_t i = 0; i < s->flat->shape_nodes.size(); i++)
        if (s->flat->shape_nodes[i]->GetComponentOfType<Shape>()->Material == s->mats[mat].get()) root = s->flat->shapes[i].material;
    if (root < 0) { g_host_err = "material is not used by any shape of the scene"; return kNotInScene; }
    rc = rthip_probe_material(s->ctx, root, (uint32_t)n, p, out_color, out_absorptance);
    if (rc) g_host_err = rthip_last_error();
    return rc;
}
int sb_background_eval(sb_scene *s, int n, const double *d, double *out_color) {
    int rc = ensure_device(s);
    if (rc) return rc;
    rc = rthip_probe_background(s->ctx, (uint32_t)n, d, out_color);
    if (rc) g_host_err = rthip_last_error();
    return rc;
}
int sb_raycast(sb_scene *s, int max_recursion, int n, const double *o, const double *d, double *out_color) {
    int rc = ensure_device(s);
    if (rc) return rc;
    rc = rthip_probe_raycast(s->ctx, (uint32_t)max_recursion, (uint32_t)n, o, d, out_color);
    if (rc) g_host_err = rthip_last_error();
    return rc;
}
// host logic only (no device): kd-tree and matrices as built by the product's Compile()
int sb_kd_dump(sb_scene *s, int node, int32_t *node_depth, int32_t *node_ntris, double *node_split,
               double *node_bounds, int32_t *tri_refs, int64_t *n_refs) {
    if (ensure_flat(s) || node < 0 || node >= (int)s->nodes.size()) return -1;
    auto mesh = s->nodes[node]->GetComponentOfType<Mesh>();
    if (!mesh) return -1;
    const KdTreeFlat &kd = mesh->GetKdTree();
    for (size_t i = 0; i < kd.nodes.size(); i++) {
        const auto &n = kd.nodes[i];
        bool leaf = n.info & RTHIP_KD_LEAF;
        if (node_depth) node_depth[i] = (int32_t)kd.depth[i];
        if (node_ntris) node_ntris[i] = leaf ? (int32_t)(n.info >> RTHIP_KD_COUNT_SHIFT) : 0;
        if (node_split) node_split[i] = leaf ? 0.0 : n.split;
    }
    if (node_bounds) std::memcpy(node_bounds, kd.bounds.data(), sizeof(double) * kd.bounds.size());
    if (tri_refs)
        for (size_t i = 0; i < kd.refs.size(); i++) tri_refs[i] = (int32_t)kd.refs[i];
    if (n_refs) *n_refs = (int64_t)kd.refs.size();
    return (int)kd.nodes.size();
}
int sb_node_matrices(sb_scene *s, int node, double *out48) {
    if (ensure_flat(s) || node < 0 || node >= (int)s->nodes.size()) return -1;
    auto t = s->nodes[node]->GetComponentOfType<Transform>();
    Matrix4 a, b, c;
    t->GetTransform(a); t->GetInverseTransform(b); t->GetNormalTransform(c);
    std::memcpy(out48, a.Elements, sizeof(double) * 16);
    std::memcpy(out48 + 16, b.Elements, sizeof(double) * 16);
    std::memcpy(out48 + 32, c.Elements, sizeof(double) * 16);
    return 0;
}
// AbstractRaycaster::CreateScreenRay is device code here; the host part is MakeCamera (see rehost_camera)
int sb_screen_rays(sb_scene *, int, int, double, int, const double *, double *, double *) {
    g_host_err = "screen rays are generated on the device (k_bounce level 0); use rehost_camera + a render";
    return RTHIP_E_UNSUPPORTED;
}

}  // extern "C"
