// rthip_multi.cpp — several GPUs of ONE node driven from ONE process ("single process, one host thread per
// frame (ncclCommInitAll), one stream per GPU"): the screen is split into interleaved tiles (tile t -> device t % n, the
// rthip_params partition), every device renders its share into a tile-compact buffer in its own HBM, the parts are moved to
// the first device over xGMI - RCCL grouped ncclSend / ncclRecv (librccl is opened at run time, only when a multi-GPU
// renderer is created) or, for repeated device ordinals / without RCCL, peer copies - and assembled there by k_assemble; one
// D2H copy hands the frame back.  replaces: the NumThreads std::async workers of AbstractRaycaster::Render
// (Raytracer/Raytracer.cpp:52-87), whose scanline scheduler is the reference's only parallel decomposition.
// Built on the public C-ABI of rthip.h (one rthip_ctx per device) + HIP; no kernels of its own.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rthip.h"

extern "C" void rthip_set_last_error(const char *msg);  // rthip.hip: the calling thread's message for rthip_last_error()

namespace {
int mfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    rthip_set_last_error(buf);
    return code;
}
#define MHIP(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return mfail(RTHIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the handful of RCCL entry points the gather needs, bound at run time
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool Load() {
        if (so) return true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (so) break;
        }
        if (!so) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(so, "ncclSend");
        Recv = (decltype(Recv))dlsym(so, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};
Rccl g_rccl;
}  // namespace

struct rthip_multi {
    std::vector<int> devices;
    std::vector<rthip_ctx *> ctx;        // one per entry of `devices`
    std::vector<ncclComm_t> comm;        // RCCL transport: one communicator per device, rank = index
    std::vector<uint32_t *> part;        // per device: its tile-compact part buffer (device 0: a slice of `gathered`)
    std::vector<hipEvent_t> rendered;    // per device: its part is complete
    std::vector<size_t> part_cap;
    uint32_t *gathered = nullptr;        // on devices[0]: n part buffers, part after part
    uint32_t *frame = nullptr;           // on devices[0]: the assembled frame
    uint32_t *self_part = nullptr;       // n = 1 over RCCL only
    size_t gathered_cap = 0, frame_cap = 0, self_cap = 0;
    hipEvent_t gathered_ev = nullptr;
    bool use_rccl = false;
    std::string transport;
};

extern "C" {

int rthip_multi_create(const int *devices, int n, rthip_multi **out) {
    if (!devices || n < 1 || n > 64 || !out) return mfail(RTHIP_E_INVALID, "rthip_multi_create: bad arguments");
    *out = nullptr;
    rthip_multi *m = new rthip_multi;
    m->devices.assign(devices, devices + n);
    bool distinct = true;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
    for (int i = 0; i < n; i++) {
        rthip_ctx *c = nullptr;
        int rc = rthip_create(devices[i], &c);
        if (rc) {
            rthip_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
    }
    m->part.assign(n, nullptr);
    m->part_cap.assign(n, 0);
    m->rendered.assign(n, nullptr);
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(devices[i]) != hipSuccess || hipEventCreateWithFlags(&m->rendered[i], hipEventDisableTiming) != hipSuccess) {
            rthip_multi_destroy(m);
            return mfail(RTHIP_E_HIP, "rthip_multi_create: event creation on device %d failed", devices[i]);
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&m->gathered_ev, hipEventDisableTiming) != hipSuccess) {
        rthip_multi_destroy(m);
        return mfail(RTHIP_E_HIP, "rthip_multi_create: event creation failed");
    }
    // transport: RCCL over xGMI when the ordinals are distinct (RCCL refuses one device twice) and librccl is there;
    // RTHIP_MULTI_TRANSPORT=copy forces peer copies, =rccl makes a missing RCCL an error
    const char *want = getenv("RTHIP_MULTI_TRANSPORT");
    const bool force_copy = want && !strcmp(want, "copy"), force_rccl = want && !strcmp(want, "rccl");
    m->transport = "peer copies";
    if (!force_copy && distinct) {
        if (g_rccl.Load()) {
            m->comm.assign(n, nullptr);
            ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n, devices);
            if (r == ncclSuccess) {
                m->use_rccl = true;
                m->transport = "RCCL ncclSend/ncclRecv";
            } else {
                m->comm.clear();
                if (force_rccl) {
                    rthip_multi_destroy(m);
                    return mfail(RTHIP_E_HIP, "rthip_multi_create: ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
                }
            }
        } else if (force_rccl) {
            rthip_multi_destroy(m);
            return mfail(RTHIP_E_HIP, "rthip_multi_create: librccl could not be loaded: %s", dlerror());
        }
    } else if (force_rccl) {
        rthip_multi_destroy(m);
        return mfail(RTHIP_E_INVALID, "rthip_multi_create: RCCL needs distinct device ordinals");
    }
    if (!m->use_rccl) {  // peer copies between distinct devices need peer access (same device: plain device-to-device copies)
        for (int i = 1; i < n; i++) {
            if (devices[i] == devices[0]) continue;
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, devices[0], devices[i]);
            if (can && hipSetDevice(devices[0]) == hipSuccess) {
                hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    *out = m;
    return 0;
}

int rthip_multi_destroy(rthip_multi *m) {
    if (!m) return 0;
    for (size_t i = 0; i < m->ctx.size(); i++)
        if (m->ctx[i]) (void)rthip_synchronize(m->ctx[i]);
    for (ncclComm_t c : m->comm)
        if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < m->devices.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        if (i > 0 && i < m->part.size() && m->part[i]) (void)hipFree(m->part[i]);
        if (i < m->rendered.size() && m->rendered[i]) (void)hipEventDestroy(m->rendered[i]);
    }
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->gathered) (void)hipFree(m->gathered);
    if (m->frame) (void)hipFree(m->frame);
    if (m->self_part) (void)hipFree(m->self_part);
    if (m->gathered_ev) (void)hipEventDestroy(m->gathered_ev);
    for (rthip_ctx *c : m->ctx)
        if (c) (void)rthip_destroy(c);
    delete m;
    return 0;
}

int rthip_multi_device_count(rthip_multi *m) { return m ? (int)m->devices.size() : 0; }
rthip_ctx *rthip_multi_context(rthip_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }
const char *rthip_multi_transport(rthip_multi *m) { return m ? m->transport.c_str() : ""; }

// the scene is replicated: every device gets its own copy (<= 2.5 GB at config 5)
int rthip_multi_upload_scene(rthip_multi *m, const rthip_scene_desc *scene, uint32_t flags) {
    if (!m || !scene) return mfail(RTHIP_E_INVALID, "rthip_multi_upload_scene: null argument");
    for (rthip_ctx *c : m->ctx) {
        int rc = rthip_upload_scene_ex(c, scene, flags);
        if (rc) return rc;
    }
    return 0;
}

int rthip_multi_interrupt(rthip_multi *m) {
    if (!m) return mfail(RTHIP_E_INVALID, "null multi-GPU renderer");
    for (rthip_ctx *c : m->ctx) (void)rthip_interrupt(c);
    return 0;
}
float rthip_multi_progress(rthip_multi *m) {  // the slowest device's
    if (!m || m->ctx.empty()) return 0.f;
    float p = 1.f;
    for (rthip_ctx *c : m->ctx) {
        const float q = rthip_progress(c);
        p = q < p ? q : p;
    }
    return p;
}

static int ensure(void **p, size_t *cap, size_t bytes) {
    if (*p && *cap >= bytes) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    MHIP(hipMalloc(p, bytes));
    *cap = bytes;
    return 0;
}

int rthip_multi_render(rthip_multi *m, const rthip_camera *cam, const rthip_params *params, uint32_t *out_pixels, rthip_stats *out_stats) {
    return rthip_multi_render_batch(m, cam, 1, params, out_pixels, out_stats);
}

// n_frames cameras in ONE round: every device traces its tiles of all the frames as one set of launches
// (rthip_render_batch_device), the parts of the whole run cross xGMI once (one send / receive per peer), one assemble launch,
// one device-to-host copy, one synchronisation at the end.  A device's share of a single 1080p frame is latency-bound (an
// eighth of a frame takes almost half the time of the whole frame); a run of frames gives every device whole frames' worth of
// rays per launch.
int rthip_multi_render_batch(rthip_multi *m, const rthip_camera *cam, uint32_t n_frames, const rthip_params *params, uint32_t *out_pixels,
                             rthip_stats *out_stats) {
    if (!m || !cam || !params || !out_pixels) return mfail(RTHIP_E_INVALID, "rthip_multi_render: null argument");
    if (n_frames == 0 || n_frames > RTHIP_MAX_BATCH) return mfail(RTHIP_E_INVALID, "rthip_multi_render_batch: %u frames per call, limit %d", n_frames, RTHIP_MAX_BATCH);
    const int n = (int)m->devices.size();
    rthip_params prm = *params;
    prm.part_count = (uint32_t)n;
    prm.part_index = 0;
    prm.layout = 1;
    const size_t part_px = (size_t)rthip_part_pixels(&prm) * n_frames;  // a device's part of the whole run: frame after frame
    if (!part_px) return mfail(RTHIP_E_INVALID, "rthip_multi_render: bad frame size %ux%u", params->width, params->height);
    const size_t part_bytes = part_px * sizeof(uint32_t), frame_px = (size_t)params->width * params->height * n_frames;
    MHIP(hipSetDevice(m->devices[0]));
    int rc;
    if ((rc = ensure((void **)&m->gathered, &m->gathered_cap, part_bytes * n))) return rc;
    if ((rc = ensure((void **)&m->frame, &m->frame_cap, frame_px * sizeof(uint32_t)))) return rc;
    // (n = 1 with RCCL: the part takes a self send / recv, which is how a one-GPU box rehearses the RCCL calls)
    const bool self_gather = n == 1 && m->use_rccl;
    m->part[0] = m->gathered;  // the first device renders straight into its slot
    if (self_gather && (rc = ensure((void **)&m->self_part, &m->self_cap, part_bytes))) return rc;
    if (self_gather) m->part[0] = m->self_part;
    for (int i = 1; i < n; i++) {
        MHIP(hipSetDevice(m->devices[i]));
        if ((rc = ensure((void **)&m->part[i], &m->part_cap[i], part_bytes))) return rc;
    }
    // every device: its interleaved tiles, asynchronously on its own stream
    for (int i = 0; i < n; i++) {
        prm.part_index = (uint32_t)i;
        if ((rc = rthip_render_batch_device(m->ctx[i], cam, n_frames, &prm, m->part[i], 0))) return rc;
        MHIP(hipSetDevice(m->devices[i]));
        MHIP(hipEventRecord(m->rendered[i], (hipStream_t)rthip_stream(m->ctx[i])));
    }
    hipStream_t s0 = (hipStream_t)rthip_stream(m->ctx[0]);
    // the gather to the first device
    if (m->use_rccl) {
        // grouped point-to-point: each peer sends its part once over its own xGMI link; stream order on both sides
        // (send after the peer's render, receive after nothing: it writes slots the first device does not render into)
        ncclResult_t r = g_rccl.GroupStart();
        for (int i = 1; i < n && r == ncclSuccess; i++) {
            r = g_rccl.Send(m->part[i], part_px, ncclUint32, 0, m->comm[i], (hipStream_t)rthip_stream(m->ctx[i]));
            if (r == ncclSuccess) r = g_rccl.Recv(m->gathered + (size_t)i * part_px, part_px, ncclUint32, i, m->comm[0], s0);
        }
        if (self_gather && r == ncclSuccess) {
            r = g_rccl.Send(m->part[0], part_px, ncclUint32, 0, m->comm[0], s0);
            if (r == ncclSuccess) r = g_rccl.Recv(m->gathered, part_px, ncclUint32, 0, m->comm[0], s0);
        }
        ncclResult_t r2 = g_rccl.GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return mfail(RTHIP_E_HIP, "rthip_multi_render: RCCL gather failed: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : r2));
    } else {
        MHIP(hipSetDevice(m->devices[0]));
        for (int i = 1; i < n; i++) {
            MHIP(hipStreamWaitEvent(s0, m->rendered[i], 0));
            if (m->devices[i] == m->devices[0])
                MHIP(hipMemcpyAsync(m->gathered + (size_t)i * part_px, m->part[i], part_bytes, hipMemcpyDeviceToDevice, s0));
            else
                MHIP(hipMemcpyPeerAsync(m->gathered + (size_t)i * part_px, m->devices[0], m->part[i], m->devices[i], part_bytes, s0));
        }
    }
    prm.part_index = 0;
    if ((rc = rthip_assemble_batch(m->ctx[0], &prm, n_frames, m->gathered, m->frame, 0))) return rc;  // [part][frame][slot] -> frames
    rthip_stats sum;
    memset(&sum, 0, sizeof sum);
    int interrupted = 0;
    for (int i = 0; i < n; i++) {  // ray counts of the parts add up to the frame's; an Interrupt() on any device abandons the frame
        rthip_stats st;
        prm.part_index = (uint32_t)i;
        rc = rthip_collect_stats(m->ctx[i], &prm, &st);
        if (rc == RTHIP_E_INTERRUPTED) { interrupted = 1; continue; }
        if (rc) return rc;
        sum.rays_primary += st.rays_primary;
        sum.rays_shadow += st.rays_shadow;
        sum.rays_reflect += st.rays_reflect;
        sum.pixels += st.pixels;
        sum.ms_total = st.ms_total > sum.ms_total ? st.ms_total : sum.ms_total;
    }
    if (interrupted) return mfail(RTHIP_E_INTERRUPTED, "rthip_multi_render: interrupted");  // out_pixels untouched
    MHIP(hipSetDevice(m->devices[0]));
    MHIP(hipMemcpyAsync(out_pixels, m->frame, frame_px * sizeof(uint32_t), hipMemcpyDeviceToHost, s0));
    MHIP(hipStreamSynchronize(s0));
    if (out_stats) *out_stats = sum;
    return 0;
}

}  // extern "C"
