// re/renderer.hpp — renderer side of the re:: host mirror (reference Raytracer/Raytracer.h:16-135).
// Renderer and AbstractRaycaster keep the reference's public surface; HipRaytracer is the new
// AbstractRaycaster subclass that is the drop-in for re::Raytracer: Render() returns at once, a
// detached thread compiles the scene, flattens it, and renders the frame on the MI355X through
// the C-ABI in include/rthip.h.  There is no CPU rendering path in this library.
#pragma once
#include <atomic>
#include <condition_variable>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene.hpp"

struct rthip_ctx;
struct rthip_multi;

namespace re {

class Renderer {  // Raytracer.h:16-51
public:
    struct RenderStatus {
        bool Finished;
        bool Interruped;
        float Percent;
        unsigned int *Pixels;
    };
    Renderer(size_t viewWidth, size_t viewHeight, real fovY = PI / 4) : m_ViewWidth(viewWidth), m_ViewHeight(viewHeight), m_FoVY(fovY) {}
    virtual ~Renderer() {}
    // Render + wait (Raytracer.cpp:12-18); rethrows a device error delivered through the promise
    virtual unsigned int *RenderSync(Scene *scene);
    virtual void Render(Scene *scene, std::promise<RenderStatus> promise) = 0;
    virtual RenderStatus GetStatus() = 0;
    virtual void Interrupt() = 0;
    virtual size_t GetViewWidth() { return m_ViewWidth; }
    virtual size_t GetViewHeight() { return m_ViewHeight; }
protected:
    size_t m_ViewWidth, m_ViewHeight;
    real m_FoVY;
};

class AbstractRaycaster : public Renderer {  // Raytracer.h:55-98
public:
    enum class AAMode : int { None = 0, SSAA = 1 };
    AbstractRaycaster(unsigned int viewWidth, unsigned int viewHeight, real fovY = PI / 4.0f);
    ~AbstractRaycaster() override;
    AAMode Antialiasing = AAMode::None;
    unsigned int NumThreads = 4;  // kept for source compatibility; the device path ignores it
protected:
    virtual Color Raycast(Scene *scene, const Ray &ray) = 0;
    Color *m_ColorBuffer0;   // not allocated: the fp64 colour buffer lives in HBM (rec level 0)
    unsigned int *m_Pixels;  // W*H, index y*W+x, 0xAABBGGRR
};

// Error delivered through the promise / RenderSync when the device path cannot run (additive:
// the reference has no error channel, SURVEY.md §8b "errors").
class DeviceError : public std::runtime_error {
public:
    DeviceError(int code, const std::string &what) : std::runtime_error(what), Code(code) {}
    int Code;
};

struct FlatScene;  // flatten.hpp

struct HipRenderStats {
    unsigned long long RaysPrimary = 0, RaysShadow = 0, RaysReflect = 0, Pixels = 0;
    float DeviceMs = 0.f;     // HIP-event time of the frame's kernels
    float FlattenMs = 0.f;    // host: Scene::Compile + flatten
    float UploadMs = 0.f;     // host->device scene upload
    bool GeometryKept = false;  // the meshes were the ones already on the device: only the small tables went up
};

struct CameraPose {  // Scene::Camera's two fields (Scene.h:80-86): one stop of a camera path
    Vector3 Position, Direction;
};

class HipRaytracer : public AbstractRaycaster {  // drop-in for re::Raytracer (Raytracer.h:100-116)
public:
    unsigned int MaxRecursion = 3;
    // additive knobs
    int Device = 0;                         // HIP device ordinal of this process
    // more than one entry: the frame is split over these GPUs of the node from this one process (interleaved tiles, parts
    // gathered to Devices[0] over xGMI with RCCL, rthip_multi_* in include/rthip.h) - the device-side counterpart of
    // NumThreads (Raytracer.h:74).  Set before the first Render.
    std::vector<int> Devices;
    unsigned int PartIndex = 0, PartCount = 1;  // interleaved-tile share rendered by this process
    unsigned int TileWidth = 32, TileHeight = 8;
    bool UseReferenceKdTree = false;        // traverse the reference's kd-tree instead of the BVH (same image)

    HipRaytracer(unsigned int viewWidth, unsigned int viewHeight, real fovY = PI / 4.0f);
    ~HipRaytracer() override;

    void Render(Scene *scene, std::promise<RenderStatus> p) override;
    // additive: a QUEUED CAMERA PATH - one frame per pose over the otherwise unchanged scene (what the Sandbox preview renders
    // frame after frame with RenderSync, Sandbox/Sandbox.cpp:315-359, when the path is known ahead).  Synchronous; the frames are
    // traced in runs of up to 64 as one set of launches each (rthip_render_batch / rthip_multi_render_batch: with several
    // Devices every GPU traces its tiles of the whole run and the parts cross xGMI once per run) and returned one after the
    // other (width*height pixels each, as RenderSync returns one) in a buffer the renderer owns until the next call.
    // Throws DeviceError like RenderSync; an Interrupt() abandons the path and returns nullptr.
    const unsigned int *RenderPathSync(Scene *scene, const std::vector<CameraPose> &path);
    void Interrupt() override;
    RenderStatus GetStatus() override;
    HipRenderStats GetLastStats();
    // additive: AbstractRaycaster::Raycast is protected (Raytracer.h:78); this is the same call for users of the class
    Color RaycastSync(Scene *scene, const Ray &ray) { return Raycast(scene, ray); }
protected:
    // Raytracer::Raycast (Raytracer.cpp:312-315): the colour along ONE ray, traced and shaded on the device (frames go through
    // Render, never through this)
    Color Raycast(Scene *scene, const Ray &ray) override;
    virtual int RendererKind() const { return 0; }  // RTHIP_RENDER_RAYTRACER
private:
    void RenderThread(Scene *scene, std::promise<RenderStatus> p);
    void EnsureContext();   // first Render: the device context(s), m_Pixels moved to page-locked memory (m_Mutex held)
    int PrepareScene(Scene *scene, HipRenderStats &stats, std::string &err);  // Compile + flatten + upload (render thread)
    unsigned int *m_PathFrames = nullptr;  // page-locked, RenderPathSync
    size_t m_PathCapacity = 0;
    rthip_ctx *m_Ctx = nullptr;      // single device; with Devices.size() > 1: the first device's context (owned by m_Multi)
    rthip_multi *m_Multi = nullptr;
    bool m_PixelsPinned = false;
    std::unique_ptr<FlatScene> m_Flat;  // the scene as last uploaded, geometry arrays released (see RenderThread)
    int m_CreateRc = 0;
    std::string m_CreateErr;
    std::mutex m_Mutex;
    std::condition_variable m_Idle;
    bool m_Busy = false;
    std::atomic<bool> m_Finished{true}, m_Interrupted{true};
    HipRenderStats m_Stats;
};

class HipDebugRaycaster : public HipRaytracer {  // drop-in for re::DebugRaycaster (Raytracer.h:119-135)
public:
    enum class Modes : unsigned int { Normal = 0, Color = 1 };
    Modes Mode = Modes::Normal;
    HipDebugRaycaster(unsigned int w, unsigned int h, real fovY = PI / 4.0f) : HipRaytracer(w, h, fovY) {}
protected:
    int RendererKind() const override { return 1 + (int)Mode; }
};

}  // namespace re
