// hip_raytracer.cpp — re::HipRaytracer: the AbstractRaycaster subclass that replaces
// re::Raytracer::Render (reference Raytracer/Raytracer.cpp:34-89) with the device path.
// Contract kept: Render() returns immediately; work happens on a detached thread that first
// compiles the scene; the promise is satisfied exactly once (also when interrupted) with
// RenderStatus{Finished=true, Interruped, Percent, Pixels=m_Pixels}; GetStatus()/Interrupt() may
// be called from another thread while rendering.  Device failures are delivered as
// re::DeviceError through the promise — there is no CPU fallback.
#include <chrono>
#include <cstring>
#include <thread>

#include "flatten.hpp"
#include "re/renderer.hpp"
#include "rthip.h"

namespace re {

unsigned int *Renderer::RenderSync(Scene *scene) {
    std::promise<RenderStatus> p;
    auto f = p.get_future();
    Render(scene, std::move(p));
    return f.get().Pixels;
}

AbstractRaycaster::AbstractRaycaster(unsigned int w, unsigned int h, real fovY) : Renderer(w, h, fovY) {
    m_ColorBuffer0 = nullptr;
    m_Pixels = new unsigned int[m_ViewWidth * m_ViewHeight]();
}
AbstractRaycaster::~AbstractRaycaster() { delete[] m_Pixels; }

HipRaytracer::HipRaytracer(unsigned int w, unsigned int h, real fovY) : AbstractRaycaster(w, h, fovY) {}

HipRaytracer::~HipRaytracer() {
    {   // a detached render thread still uses *this: wait for it
        std::unique_lock<std::mutex> lock(m_Mutex);
        if (m_Busy && m_Multi) rthip_multi_interrupt(m_Multi);
        else if (m_Busy && m_Ctx) rthip_interrupt(m_Ctx);
        m_Idle.wait(lock, [this] { return !m_Busy; });
    }
    if (m_Ctx && m_PathFrames) (void)rthip_host_free(m_Ctx, m_PathFrames);
    if (m_Ctx && m_PixelsPinned) {  // m_Pixels is page-locked memory of the library, not the base class's new[]
        (void)rthip_host_free(m_Ctx, m_Pixels);
        m_Pixels = nullptr;
    }
    if (m_Multi) rthip_multi_destroy(m_Multi);  // owns m_Ctx
    else if (m_Ctx) rthip_destroy(m_Ctx);
}

// Raytracer::Raycast (Raytracer.cpp:312-315) for ONE ray, on the device: the scene goes up as for a frame (or stays, when it is the
// one already there) and the ray runs through the same device functions a frame's rays do (rthip_probe_raycast).  A single ray
// costs a launch and a round trip - the reference calls this per pixel from its worker threads, this class renders frames with
// Render instead and keeps Raycast for callers that want the colour along one ray (picking, probes, tests).
Color HipRaytracer::Raycast(Scene *scene, const Ray &ray) {
    if (RendererKind() != RTHIP_RENDER_RAYTRACER)
        throw DeviceError(RTHIP_E_UNSUPPORTED, "HipDebugRaycaster renders whole frames on the device; it has no per-ray Raycast");
    {
        std::unique_lock<std::mutex> lock(m_Mutex);
        m_Idle.wait(lock, [this] { return !m_Busy; });
        m_Busy = true;
        EnsureContext();
    }
    int rc = 0;
    std::string err;
    HipRenderStats stats;
    double c[3] = {0, 0, 0};
    try {
        rc = PrepareScene(scene, stats, err);
        if (!rc) {
            const double o[3] = {ray.Origin.X, ray.Origin.Y, ray.Origin.Z}, d[3] = {ray.Direction.X, ray.Direction.Y, ray.Direction.Z};
            rc = rthip_probe_raycast(m_Ctx, MaxRecursion, 1, o, d, c);
            if (rc) err = rthip_last_error();
        }
    } catch (const std::exception &e) {
        rc = RTHIP_E_INVALID;
        err = e.what();
    }
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Busy = false;
        m_Idle.notify_all();
    }
    if (rc) throw DeviceError(rc, err);
    Color out;  // (the device colour is already clamped: no constructor in between)
    out.R = c[0];
    out.G = c[1];
    out.B = c[2];
    return out;
}

void HipRaytracer::Interrupt() {
    m_Interrupted = true;  // Raytracer.cpp:91-94
    std::lock_guard<std::mutex> lock(m_Mutex);
    if (m_Busy && m_Multi) rthip_multi_interrupt(m_Multi);
    else if (m_Busy && m_Ctx) rthip_interrupt(m_Ctx);
}

Renderer::RenderStatus HipRaytracer::GetStatus() {
    RenderStatus s;
    s.Finished = m_Finished;
    s.Interruped = m_Interrupted;
    s.Pixels = m_Pixels;
    rthip_ctx *ctx;
    rthip_multi *multi;
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        ctx = m_Ctx;
        multi = m_Multi;
    }
    s.Percent = s.Finished ? (s.Interruped ? 0.f : 1.f) : (multi ? rthip_multi_progress(multi) : (ctx ? rthip_progress(ctx) : 0.f));
    return s;
}

HipRenderStats HipRaytracer::GetLastStats() {
    std::lock_guard<std::mutex> lock(m_Mutex);
    return m_Stats;
}

void HipRaytracer::Render(Scene *scene, std::promise<RenderStatus> p) {
    {
        std::unique_lock<std::mutex> lock(m_Mutex);
        m_Idle.wait(lock, [this] { return !m_Busy; });  // one render at a time per renderer
        m_Busy = true;
    }
    m_Finished = false;  // m_Status = { false, false, 0, m_Pixels }, Raytracer.cpp:36
    m_Interrupted = false;
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        EnsureContext();
    }
    std::thread th(&HipRaytracer::RenderThread, this, scene, std::move(p));
    th.detach();  // Raytracer.cpp:84-87
}

// First Render: the device context, and m_Pixels moves to page-locked memory so that the copy back of every frame runs at PCIe
// rate (1080p: 0.18 instead of 0.50 ms).  Done on the caller's thread, because the pixel buffer may only change hands at a
// Render call (RenderStatus.Pixels is valid until the next Render); a failure is reported through the promise by the render
// thread.
void HipRaytracer::EnsureContext() {
    if (m_Ctx) return;
    if (Devices.size() > 1) {
        m_CreateRc = rthip_multi_create(Devices.data(), (int)Devices.size(), &m_Multi);
        if (!m_CreateRc) m_Ctx = rthip_multi_context(m_Multi, 0);
    } else {
        m_CreateRc = rthip_create(Devices.empty() ? Device : Devices[0], &m_Ctx);
    }
    m_CreateErr = m_CreateRc ? rthip_last_error() : "";
    void *pinned = nullptr;
    if (!m_CreateRc && rthip_host_alloc(m_Ctx, sizeof(unsigned int) * m_ViewWidth * m_ViewHeight, &pinned) == 0) {
        delete[] m_Pixels;
        m_Pixels = static_cast<unsigned int *>(pinned);
        m_PixelsPinned = true;
    }
}

// Scene::Compile (Raytracer.cpp:42), flatten, upload - or, when the meshes are the ones already on the device, only the small tables
int HipRaytracer::PrepareScene(Scene *scene, HipRenderStats &stats, std::string &err) {
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
    int rc = 0;
    {
        auto t0 = clock::now();
        scene->Compile();  // Raytracer.cpp:42
        // m_Flat describes what the device holds (geometry arrays released): if the meshes are still the same objects
        // at the same revision, only the small tables are flattened and uploaded again
        FlatScene flat;
        rc = FlattenScene(scene, flat, err, m_Ctx ? m_Flat.get() : nullptr);
        auto t1 = clock::now();
        stats.FlattenMs = ms(t0, t1);
        if (!rc) {
            std::lock_guard<std::mutex> lock(m_Mutex);
            if (!m_Ctx) {  // Render() could not create the context
                rc = m_CreateRc ? m_CreateRc : RTHIP_E_STATE;
                err = m_CreateErr;
            }
        }
        if (!rc) {
            auto upload = [&]() { return m_Multi ? rthip_multi_upload_scene(m_Multi, &flat.desc, UploadFlags(flat)) : rthip_upload_scene_ex(m_Ctx, &flat.desc, UploadFlags(flat)); };
            rc = upload();
            if (rc == RTHIP_E_UNSUPPORTED && (flat.needs_device_bvh || flat.needs_device_kd) && !flat.geometry_kept) {
                // the device would not build a BVH (deeper than its traversal stacks): build them on the host after all
                ForceHostBvhEverywhere(scene->GetRoot().get());
                flat = FlatScene();
                rc = FlattenScene(scene, flat, err, nullptr);
                if (!rc) rc = upload();
            }
            if (rc && err.empty()) err = rthip_last_error();
            if (!rc) {
                stats.GeometryKept = flat.geometry_kept;
                flat.ReleaseGeometry();
                m_Flat.reset(new FlatScene(std::move(flat)));
            } else {
                m_Flat.reset();  // the device scene is in an unknown state: full upload next time
            }
        }
        auto t2 = clock::now();
        stats.UploadMs = ms(t1, t2);
    }
    return rc;
}

static void FillParams(rthip_params &prm, size_t w, size_t h, unsigned max_recursion, bool ssaa, int renderer, unsigned part_index, unsigned part_count,
                       unsigned tile_w, unsigned tile_h, bool use_kd) {
    prm.width = (uint32_t)w;
    prm.height = (uint32_t)h;
    prm.max_recursion = max_recursion;
    prm.aa_mode = ssaa ? 1 : 0;
    prm.renderer = renderer;
    prm.part_index = part_index;
    prm.part_count = part_count;
    prm.tile_w = tile_w;
    prm.tile_h = tile_h;
    prm.layout = 0;
    prm.accel = use_kd ? 1 : 0;
}

const unsigned int *HipRaytracer::RenderPathSync(Scene *scene, const std::vector<CameraPose> &path) {
    {
        std::unique_lock<std::mutex> lock(m_Mutex);
        m_Idle.wait(lock, [this] { return !m_Busy; });
        m_Busy = true;
        EnsureContext();
    }
    m_Finished = false;
    m_Interrupted = false;
    int rc = 0;
    std::string err;
    HipRenderStats stats;
    const size_t frame_px = m_ViewWidth * m_ViewHeight;
    try {
        rc = PrepareScene(scene, stats, err);
        if (!rc && path.size() * frame_px > m_PathCapacity) {
            if (m_PathFrames) (void)rthip_host_free(m_Ctx, m_PathFrames);
            m_PathFrames = nullptr;
            m_PathCapacity = 0;
            void *pinned = nullptr;
            rc = rthip_host_alloc(m_Ctx, sizeof(unsigned int) * path.size() * frame_px, &pinned);
            if (rc) err = rthip_last_error();
            else { m_PathFrames = static_cast<unsigned int *>(pinned); m_PathCapacity = path.size() * frame_px; }
        }
        rthip_params prm;
        FillParams(prm, m_ViewWidth, m_ViewHeight, MaxRecursion, Antialiasing == AAMode::SSAA, RendererKind(), 0, 1, TileWidth, TileHeight, UseReferenceKdTree);
        std::vector<rthip_camera> cams;
        for (size_t f0 = 0; !rc && !m_Interrupted && f0 < path.size(); f0 += RTHIP_MAX_BATCH) {
            const size_t nf = std::min<size_t>(RTHIP_MAX_BATCH, path.size() - f0);
            cams.resize(nf);
            for (size_t k = 0; k < nf; k++) MakeCamera(path[f0 + k].Position, path[f0 + k].Direction, m_ViewWidth, m_ViewHeight, m_FoVY, cams[k]);
            rthip_stats st;
            std::memset(&st, 0, sizeof st);
            if (m_Multi) rc = rthip_multi_render_batch(m_Multi, cams.data(), (uint32_t)nf, &prm, m_PathFrames + f0 * frame_px, &st);
            else rc = rthip_render_batch(m_Ctx, cams.data(), (uint32_t)nf, &prm, m_PathFrames + f0 * frame_px);
            if (rc == RTHIP_E_INTERRUPTED) { m_Interrupted = true; rc = 0; break; }
            if (rc) { err = rthip_last_error(); break; }
            if (m_Multi || rthip_get_stats(m_Ctx, &st) == 0) {
                stats.RaysPrimary += st.rays_primary; stats.RaysShadow += st.rays_shadow; stats.RaysReflect += st.rays_reflect;
                stats.Pixels += st.pixels; stats.DeviceMs += st.ms_total;
            }
        }
    } catch (const std::exception &e) {
        rc = RTHIP_E_INVALID;
        err = e.what();
    }
    m_Finished = true;
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Stats = stats;
        m_Busy = false;
        m_Idle.notify_all();
    }
    if (rc) throw DeviceError(rc, err);
    return m_Interrupted ? nullptr : m_PathFrames;
}

void HipRaytracer::RenderThread(Scene *scene, std::promise<RenderStatus> p) {
    int rc = 0;
    std::string err;
    HipRenderStats stats;
    try {
        rc = PrepareScene(scene, stats, err);
        if (!rc && !m_Interrupted) {
            rthip_camera cam;
            MakeCamera(scene, m_ViewWidth, m_ViewHeight, m_FoVY, cam);
            rthip_params prm;
            FillParams(prm, m_ViewWidth, m_ViewHeight, MaxRecursion, Antialiasing == AAMode::SSAA, RendererKind(), PartIndex, PartCount, TileWidth, TileHeight,
                       UseReferenceKdTree);
            rthip_stats st;
            std::memset(&st, 0, sizeof st);
            if (m_Multi) rc = rthip_multi_render(m_Multi, &cam, &prm, m_Pixels, &st);
            else rc = rthip_render(m_Ctx, &cam, &prm, m_Pixels, nullptr, nullptr);
            if (rc == RTHIP_E_INTERRUPTED) {
                m_Interrupted = true;
                rc = 0;
            } else if (rc) {
                err = rthip_last_error();
            } else {
                if (m_Multi || rthip_get_stats(m_Ctx, &st) == 0) {
                    stats.RaysPrimary = st.rays_primary;
                    stats.RaysShadow = st.rays_shadow;
                    stats.RaysReflect = st.rays_reflect;
                    stats.Pixels = st.pixels;
                    stats.DeviceMs = st.ms_total;
                }
            }
        }
    } catch (const std::exception &e) {
        rc = RTHIP_E_INVALID;
        err = e.what();
    }
    RenderStatus status;
    status.Interruped = m_Interrupted;
    status.Percent = status.Interruped ? 0.f : 1.f;
    status.Pixels = m_Pixels;
    status.Finished = true;
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Stats = stats;
    }
    m_Finished = true;  // Raytracer.cpp:76-77
    // release the renderer BEFORE publishing the result: the owner may destroy it as soon as the future is ready
    // (notified under the lock: a destructor waiting in m_Idle cannot run before this thread is done with *this)
    {
        std::lock_guard<std::mutex> lock(m_Mutex);
        m_Busy = false;
        m_Idle.notify_all();
    }
    if (rc)
        p.set_exception(std::make_exception_ptr(DeviceError(rc, err)));
    else
        p.set_value(status);  // Raytracer.cpp:79
}

}  // namespace re
