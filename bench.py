#!/usr/bin/env python3
"""bench.py — Mrays/s of the Raycast hot path on N MI355X GPUs of one node.

A "step" is one frame of the workload: BASELINE.json configs[2], the reference's "Demo 4 - Bunny"
(Wavefront-OBJ TriangleMesh with its kd-tree + checker plane + SkyBox sun, 1920x1080,
MaxRecursion 3).  The flattened scene is resident in HBM before the timed
region; every step traces the whole frame.  With N > 1 the frame is partitioned into interleaved
32x8 screen tiles (tile t -> rank t % N), each rank renders its share into a tile-compact buffer in
HBM, the buffers of a call are gathered over RCCL to ONE rank - call k's to rank k % N, so that the receive traffic and the
assemble kernel go round the ranks (--gather-dst 0: always rank 0; --collective all_gather: to every rank) - and assembled
into the frames on that rank's device; total work
is fixed as N grows ("strong" scaling).

Frames are traced in runs: one rthip_render_batch_device call takes --batch frames (default 32 per GPU up to the library's 64 per call - or half the
timed burst when that is shorter; every frame has a camera of its own and is traced in full — nothing is reused between frames) through ONE set of kernel
launches, and --inflight calls (default 3) are in flight on separate HIP streams.  A run of frames is what a camera
path rendered ahead of time looks like; it is also what keeps a GPU busy when its share of a 1080p frame is an
eighth of it (a frame is ~20 dependent launches, each of them latency-bound on so few rays).  --steps K times exactly
K frames: K // batch full calls and one shorter call.  --latency also reports the latency of a single frame.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel - the one that walks the meshes, k_trace_coop on this scene: algorithmic VALU flops from
SURVEY.md §8d's per-unit weights x counts measured on this call, against the fp64 vector peak; the HBM fraction next to it
from the PMC-measured traffic), "config4" (BASELINE configs[3], the 2 M-triangle 4K frame: the config whose roof IS HBM),
"single_frame_latency_ms" / "ms_per_frame_with_d2h" in "config", and "cpu_baseline" (the genuine reference compiled in
oracle/_ref - or the C oracle port - timed on the host cores, NumThreads 16 and 8, medians of 5 frames), and "config2" / "config3_closeup" /
"config5": the other BASELINE configs as lone frames on one GPU, each with its own roofline object.

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own N ranks (torch.distributed.run as a child
process) and relays rank 0's line and the exit code.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, MAX_REC = 1920, 1080, 3
FRAME0_CHECKSUM = -27903797358016  # sum of the int32 pixels of the frame from the scene's own camera (unchanged since round 1; its pixels are
                                   # what tests/test_gpu.py compares with the reference's golden subset)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (spec)
# fp64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flops (FMA) x 2.4 GHz = 78.6 TFLOP/s (the guide's 157.3 TFLOP/s
# fp32 vector figure is twice that through packed fp32).  The kernels are built with -ffp-contract=off (bit parity with the
# non-fused x86-64 reference), so every fp64 instruction is ONE flop: the usable peak of this code is half, 39.3 TFLOP/s.
FP64_VALU_PEAK_TFLOPS = 78.6
FP64_VALU_PEAK_UNFUSED_TFLOPS = 39.3
# the committed PMC summaries the measured side of the rooflines is quoted from (tracked files; tools/prof_round.sh makes them)
def _newest(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return os.path.join("profiles", n)
    return os.path.join("profiles", names[0])


PMC_CFG3 = _newest("r03g_pmc_summary_config3_batch32.json", "r03f_pmc_summary_config3_batch32.json", "r03e_pmc_summary_config3_batch32.json", "r03_pmc_summary_config3_batch32.json", "r03_pmc_summary_config3_batch8.json", "r02_pmc_summary_config3_batch8.json")
PMC_CFG4 = _newest("r03g_pmc_summary_config4.json", "r03f_pmc_summary_config4.json", "r03e_pmc_summary_config4.json", "r03_pmc_summary_config4.json", "r02_pmc_summary_config4.json")
PMC_CFG5 = _newest("r03g_pmc_summary_config5.json", "r03f_pmc_summary_config5.json", "r03_pmc_summary_config5.json")

# ALGORITHMIC flops of the dominant kernel (SURVEY.md §8d "ALGORITHMIC flops per ray": fixed weights x measured counts), for
# k_trace_quad's units of work:
#   72 per BVH step  : 4 child boxes x (6 fma = 12 flops + 6 min/max) - fp32 VALU, counted as issued (unpacked fp32 and fp64
#                      instructions take the same issue slot on CDNA4)
#   54 per triangle  : Mesh::IntersectTriangle + Baricentric in fp64 (Scene.cpp:421-467, 495-507): two 3-dots for the plane
#                      (10 + 3 sub), the hit point (1 div + 6), v2 (3), two more dots (10), v / w / u (10), |O-P|^2 (8), compares
#   180 per shape-table entry tested: SURVEY's weight (two 4x4 mat-vec, normalise, intersect, world point, d^2)
#   140 per ray      : hit record of the winner (hit_details: transforms, barycentric normal blend, two normalisations)
#   25 per shape-table entry that the bounding-ball test rejects (ball_missed: two 3-dots, a difference, the compares) - such
#                      an entry never reaches the transform and the intersection the 180 stand for
FLOPS_NODE, FLOPS_TRI, FLOPS_SHAPE, FLOPS_RAY, FLOPS_BALL = 72, 54, 180, 140, 25


def shape_flops(entries, entries_full):
    return FLOPS_SHAPE * int(entries_full) + FLOPS_BALL * (int(entries) - int(entries_full))


def algorithmic_flops(st):
    return FLOPS_NODE * st.kd_nodes_visited + FLOPS_TRI * st.kd_tris_tested + shape_flops(st.shape_tests, st.shape_tests_full) + FLOPS_RAY * st.rays_hard


# ... and of the two kernels that shade (SURVEY.md §8d's weights): k_trace_easy generates the level-0 rays (30 each), tests every
# ray of a level against the shape table (180 per entry), shades the rays it finishes itself - material (per InterpolatedMaterial
# node: the noise + 24 for the wrap and the lerps; Perlin 500, Worley 300, Marble 525, checker 8), SkyBox on a miss (Perlin 500 + 30
# for the mix and the sun disc), 30 per light term, 12 per reflected ray; k_shade does the shading part for the rays that needed a
# mesh walk.  Counts come from a counting context (rthip_stats.shape_hits / background_evals / rays_*_by_kernel).
FLOPS_RAYGEN, FLOPS_LIGHT, FLOPS_REFLECT, FLOPS_SKY, FLOPS_LERP = 30, 30, 12, 530, 24
FLOPS_NOISE = {0: 8, 1: 500, 2: 300, 3: 525}  # RTHIP_NOISE_CHECKER / PERLIN / WORLEY / MARBLE


def material_flops(desc_ptr):
    """(flops of one material evaluation per shape ordinal, flops of one background evaluation) of a flattened scene"""
    from raytracing_amd import device
    shapes, mats, noises, bg = device.scene_tables(desc_ptr)
    per_shape = []
    for sh in shapes:
        root = sh.material
        per_shape.append(sum(FLOPS_NOISE[noises[mats[k].noise].type] + FLOPS_LERP for k in range(mats[root].first_op, root + 1) if mats[k].op == 1))
    return per_shape, (FLOPS_SKY if bg == 2 else 0)


def walk_kernel(ctx):
    """the kernel that walks the meshes of this context's long lists (>= 300 000 rays per level): k_trace_coop (per-ray work one ray per
    lane, up to 64 rays per wave) unless RTHIP_QUAD_COOP=0 forces k_trace_quad (everything of a ray in its quad), which also takes the
    short lists.  Until a call has reported its list lengths both are launched and one returns at once; the times below bracket the pair."""
    return "k_trace_coop" if ctx.upload_info().walk_coop else "k_trace_quad"


def kernel_flops(st, mat, sky, walk="k_trace_quad"):
    """algorithmic flops of one call, by kernel, from the statistics of a counting context"""
    shaded = lambda k: sum(int(st.shape_hits[k][i]) * mat[min(i, len(mat) - 1)] for i in range(64) if st.shape_hits[k][i]) if mat else 0
    shading = lambda k: shaded(k) + sky * int(st.background_evals[k]) + FLOPS_LIGHT * int(st.rays_shadow_by_kernel[k]) + FLOPS_REFLECT * int(st.rays_reflect_by_kernel[k])
    return {walk: int(algorithmic_flops(st)),
            "k_trace_easy": int(FLOPS_RAYGEN * st.rays_primary + shape_flops(st.shape_tests_easy, st.shape_tests_easy_full) + shading(0)),
            "k_shade": int(shading(1))}


def kernel_ms(st, walk="k_trace_quad"):
    """summed duration of each kernel over the levels of one call (a context created with RTHIP_PROFILE_LAUNCHES=1: HIP events around
    plain launches that have the GPU to themselves)"""
    n = max(1, st.launches_trace)
    return {walk: float(st.ms_trace), "k_trace_easy": float(st.ms_easy), "k_shade": float(sum(st.level_ms_shade[i] for i in range(min(n, 17))))}


def measured_pmc(kernel, rel):
    k = _pmc_kernel(_load(rel), kernel)
    if not k:
        return None
    out = {key: k[key] for key in ("valu_issue_util", "wave_time_waiting", "l2_hit_rate", "active_lanes_per_valu_inst", "valu_insts_per_frame") if key in k}
    if "valu_issue_util" in out and "active_lanes_per_valu_inst" in out:
        out["valu_lane_cycle_util"] = out["valu_issue_util"] * out["active_lanes_per_valu_inst"] / 64.0
    out["source"] = rel
    return out


def config_object(workload, sd, kw, local_rank, frames=5, hbm_pmc=None):
    """One more BASELINE config on ONE GPU, lone frames left in HBM (every config's GPU numbers in the driver line):
    median ms / frame and Mrays/s of `frames` frames replayed from the captured launch graph, the checksum of the frame, the
    start-up times, and the frame's position against the fp64 vector roof: algorithmic flops of the three ray kernels (a counting
    context) over the frame time, with each kernel's own share (per-launch HIP events of a profiling context)."""
    import torch
    from raytracing_amd import device, host
    w, h, mr = kw["width"], kw["height"], kw["max_recursion"]
    tc = time.perf_counter()
    hs = host.HostScene(sd, device_index=local_rank)
    t0 = time.perf_counter()
    desc = hs.flatten()
    t1 = time.perf_counter()
    cam, prm = hs.camera(w, h), device.make_params(w, h, mr)
    fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    mat, sky = material_flops(desc)

    def context(env=None):
        if env:
            os.environ[env] = "1"
        c = device.Context(local_rank)
        if env:
            os.environ[env] = "0"
        c.upload(desc)
        return c

    ctx = context()
    ctx.synchronize()
    t2 = time.perf_counter()
    up = ctx.upload_info()
    walk = "k_trace_quad" + (" / k_trace_coop" if up.walk_coop else "")  # (a lone frame's short lists go through k_trace_quad either way)
    for _ in range(3):
        ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
    st = ctx.stats()
    rays = int(st.rays_primary + st.rays_shadow + st.rays_reflect)
    ts, wall = [], []
    for _ in range(frames):
        t = time.perf_counter()
        ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        wall.append((time.perf_counter() - t) * 1e3)   # the synchronous call: + the statistics read-back
        ts.append(float(ctx.stats().ms_total))          # the frame on the device: HIP events around its launches
    ts.sort()
    wall.sort()
    ms = ts[len(ts) // 2]
    checksum = int(fb.to(torch.int64).sum().item())
    ctx.close()
    c = context("RTHIP_COUNTERS")
    c.render_device(cam, prm, fb.data_ptr(), sync=True)
    cst = c.stats()
    c.close()
    c = context("RTHIP_PROFILE_LAUNCHES")
    for _ in range(2):
        c.render_device(cam, prm, fb.data_ptr(), sync=True)
    pst = c.stats()
    c.close()
    fl, kms = kernel_flops(cst, mat, sky, walk), kernel_ms(pst, walk)
    total = sum(fl.values())
    tf = total / (ms * 1e-3) / 1e12
    out = {"workload": workload, "rays_per_frame": rays, "ms_per_frame": ms, "ms_per_frame_samples": ts, "ms_per_synchronous_call": wall[len(wall) // 2],
           "value": rays / ms / 1e3, "unit": "Mrays/s",
           "frames": frames, "frame_checksum": checksum,
           "scene_construction_s": t0 - tc, "scene_compile_flatten_s": t1 - t0, "scene_upload_and_device_build_s": t2 - t1,
           "trees": {"kd_built_on": "device" if up.kd_meshes else "host", "kd_device_ms": up.kd_device_ms, "bvh_built_on": "device" if up.bvh_meshes else "host"},
           "roofline": {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                        "frac_of_unfused_peak": tf / FP64_VALU_PEAK_UNFUSED_TFLOPS, "algorithmic_flops_per_frame": total, "traffic": None,
                        "note": "whole frame: algorithmic flops of k_trace_easy + the walk kernel + k_shade over the frame's wall time",
                        "kernels": [{"kernel": k, "algorithmic_flops_per_frame": fl[k], "ms_per_frame": kms[k],
                                     "achieved": (fl[k] / (kms[k] * 1e-3) / 1e12) if kms[k] > 0 else None,
                                     "frac": (fl[k] / (kms[k] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if kms[k] > 0 else None} for k in fl],
                        "counts": {"bvh_steps": int(cst.kd_nodes_visited), "triangle_tests": int(cst.kd_tris_tested), "shape_entries_walk": int(cst.shape_tests),
                                   "shape_entries_walk_past_ball_test": int(cst.shape_tests_full), "shape_entries_easy": int(cst.shape_tests_easy),
                                   "shape_entries_easy_past_ball_test": int(cst.shape_tests_easy_full), "rays_walked": int(cst.rays_hard),
                                   "background_evals": [int(cst.background_evals[0]), int(cst.background_evals[1])]}}}
    if hbm_pmc:
        out["roofline"]["hbm"] = hbm_pmc
    del hs
    return out




def _load(rel):
    path = os.path.join(ROOT, rel)
    return json.load(open(path)) if os.path.exists(path) else None


def _pmc_kernel(j, name):
    """the entry of a kernel instantiation in a PMC summary (`name` like "k_trace_coop<false>"; the profiler may spell out more
    template arguments)"""
    if not j:
        return None
    ks = j["kernels"]
    if name in ks:
        return ks[name]
    stem = name.rstrip(">")
    for k in sorted(ks):
        if k.startswith(stem + ",") or k.startswith(stem + ">"):
            return ks[k]
    return None


def measured_traffic(frames_per_call, launches_per_call, rel=PMC_CFG3, kernel="k_trace_quad"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/prof_passes.sh: separate rocprofv3
    --pmc runs of FETCH_SIZE and of WRITE_SIZE over calls of this very shape, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950; tools/prof_summary.py gives the kernel's bytes per frame): x frames per call / launches per call."""
    k = _pmc_kernel(_load(rel), kernel + "<false>")
    if not k or "hbm_bytes_per_frame" not in k:
        return None
    return float(k["hbm_bytes_per_frame"]) * frames_per_call / max(1, launches_per_call)


def measured_valu(rel=PMC_CFG3, kernel="k_trace_quad"):
    """VALU issue utilisation / active lanes / waiting fraction of the dominant kernel from the committed PMC passes over the
    bench's own shape of work (`tools/prof_passes.sh <tag> 3 4 8` + prof_summary.py): the measured side of the VALU roof."""
    src = rel
    j = _load(src)
    if j is None:
        return None
    k = _pmc_kernel(j, kernel + "<false>")
    if k is None:
        return None
    out = {key: k[key] for key in ("valu_issue_util", "wave_time_waiting", "l2_hit_rate", "active_lanes_per_valu_inst") if key in k}
    if "valu_issue_util" in out and "active_lanes_per_valu_inst" in out:
        out["valu_lane_cycle_util"] = out["valu_issue_util"] * out["active_lanes_per_valu_inst"] / 64.0
    out["source"] = src
    return out


def cpu_baseline(sd, frames=5):
    """Times the CPU renderer on the host cores on a bounded sample of the same workload: `frames` whole 1080p frames at
    NumThreads = min(16, host cores) and at NumThreads = 8 (the Sandbox's setting, Sandbox/Sandbox.cpp:139), medians of 5."""
    import oracle
    cores = min(16, os.cpu_count() or 1)
    if oracle.have_ref():
        lib, kind = oracle.load_ref(), "reference"
    else:
        lib, kind = oracle.load_oracle(), "port"
    scene = sd.build(lib)
    scene.render(W // 8, H // 8, max_recursion=MAX_REC, threads=cores)  # compiles the scene (kd build) outside the timed frames

    def median_frame(threads):
        ts, rays = [], 0
        for _ in range(frames):
            t0 = time.time()
            r = scene.render(W, H, max_recursion=MAX_REC, threads=threads)
            ts.append(time.time() - t0)
            rays = r["rays"]
        ts.sort()
        return ts[len(ts) // 2], rays, ts

    dt, rays, all_t = median_frame(cores)
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
           "sample": f"median of {frames} whole frames {W}x{H} of the same scene (MaxRecursion {MAX_REC}, {rays} rays each), NumThreads={cores}: "
                     + ", ".join(f"{t:.2f}" for t in all_t) + " s",
           "ms_per_frame_sample": dt * 1e3}
    if cores != 8:
        dt8, rays8, all8 = median_frame(8)
        out["threads8"] = {"value": rays8 / dt8 / 1e6, "unit": "Mrays/s", "cores": 8, "ms_per_frame_sample": dt8 * 1e3,
                           "sample": f"median of {frames} frames, NumThreads=8 (Sandbox/Sandbox.cpp:139): " + ", ".join(f"{t:.2f}" for t in all8) + " s"}
    return out


def hbm_roofline(pmc_rel, kernel=None):
    """HBM roof of the walk kernel from a committed PMC summary: bytes per frame (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) over the
    kernel's own time in the same passes.  kernel = None: whichever of the two walk kernels did the work in that summary (the long lists of
    a 4K / 8K frame are k_trace_coop's, the short ones of its deep levels k_trace_quad's: the larger one is reported, the other's share named)"""
    j = _load(pmc_rel)
    other = None
    if kernel is None:
        cands = [(n, _pmc_kernel(j, n + "<false>")) for n in ("k_trace_coop", "k_trace_quad")]
        cands = [(n, k) for n, k in cands if k and k.get("hbm_bytes_per_frame")]
        if not cands:
            return None
        cands.sort(key=lambda nk: -nk[1]["gpu_ms_per_frame_serialised"])
        kernel, k = cands[0][0] + "<false>", cands[0][1]
        if len(cands) > 1:
            other = {"kernel": cands[1][0], "kernel_ms_per_frame": cands[1][1]["gpu_ms_per_frame_serialised"], "traffic": cands[1][1]["hbm_bytes_per_frame"]}
    else:
        k = _pmc_kernel(j, kernel)
    if not (k and k.get("hbm_bytes_per_frame")):
        return None
    gbs = k["hbm_bytes_per_frame"] / (k["gpu_ms_per_frame_serialised"] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel.split("<")[0], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": k["hbm_bytes_per_frame"], "kernel_ms_per_frame": k["gpu_ms_per_frame_serialised"],
            "valu_issue_util": k.get("valu_issue_util"), "active_lanes_per_valu_inst": k.get("active_lanes_per_valu_inst"),
            "l2_hit_rate": k.get("l2_hit_rate"), **({"short_lists": other} if other else {}), "source": pmc_rel}


def config4_line(local_rank, frames=5):
    """BASELINE configs[3] on ONE GPU, lone frames left in HBM: the 2 097 152-triangle height field + 16 reflective spheres,
    3840x2160, MaxRecursion 4 - the workload whose BVH + triangles (0.6 GB) do not fit L2, i.e. where HBM is the roof.
    Measured twice: with the mesh's two trees - the 4-wide BVH the kernels walk and the reference's kd-tree - built on the device at
    upload (RTHIP_UPLOAD_BUILD_BVH | RTHIP_UPLOAD_BUILD_KD: the library's default for meshes of 2^18 triangles and more; value /
    ms_per_frame) and with both built by the host library on host threads (the same algorithms, the same frames)."""
    import torch
    from raytracing_amd import device, host, scenes
    sd, kw = scenes.config(4)
    w, h, mr = kw["width"], kw["height"], kw["max_recursion"]
    runs = {}
    saved = os.environ.get("RE_BVH_DEVICE_MIN_TRIS")
    saved_kd = os.environ.get("RE_KD_DEVICE_MIN_TRIS")
    for name, env in (("trees_built_on_device", "1"), ("trees_built_on_host", "0")):
        os.environ["RE_BVH_DEVICE_MIN_TRIS"] = env
        os.environ["RE_KD_DEVICE_MIN_TRIS"] = env
        tc = time.perf_counter()
        hs = host.HostScene(sd, device_index=local_rank)   # the caller's part: 2 M triangles handed to Mesh::AddTriangle through the re:: API
        t0 = time.perf_counter()
        desc = hs.flatten()                                # Scene::Compile (Raytracer.cpp:42) + flatten
        t1 = time.perf_counter()
        ctx = device.Context(local_rank)
        ctx.upload(desc)                                   # H2D + the trees the host left to the device
        ctx.synchronize()
        t2 = time.perf_counter()
        up = ctx.upload_info()
        cam, prm = hs.camera(w, h), device.make_params(w, h, mr)
        fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
        for _ in range(3):
            ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
        st = ctx.stats()
        rays = st.rays_primary + st.rays_shadow + st.rays_reflect
        ts, wall = [], []
        for _ in range(frames):
            t = time.perf_counter()
            ctx.render_device(cam, prm, fb.data_ptr(), sync=True)
            wall.append((time.perf_counter() - t) * 1e3)   # the synchronous call: + the statistics read-back
            ts.append(float(ctx.stats().ms_total))          # the frame on the device (HIP events around its launches)
        ts.sort()
        wall.sort()
        ms = ts[len(ts) // 2]
        runs[name] = {"rays_per_frame": int(rays), "ms_per_frame": ms, "ms_per_synchronous_call": wall[len(wall) // 2], "value": rays / ms / 1e3,
                      "frame_checksum": int(fb.to(torch.int64).sum().item()),
                      "scene_construction_s": t0 - tc, "scene_compile_flatten_s": t1 - t0, "scene_upload_and_device_build_s": t2 - t1,
                      "kd_tree": {"built_on": "device" if up.kd_meshes else "host", "device_ms": up.kd_device_ms, "nodes": int(up.kd_nodes) or None,
                                  "leaf_refs": int(up.kd_refs) or None, "levels": int(up.kd_levels) or None},
                      "bvh": {"built_on": "device" if up.bvh_meshes else "host", "nodes": int(up.bvh_nodes) or None}}
        ctx.close()
        del hs
    for key, val in (("RE_BVH_DEVICE_MIN_TRIS", saved), ("RE_KD_DEVICE_MIN_TRIS", saved_kd)):
        if val is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = val
    best = runs["trees_built_on_device"]  # the default path for a mesh of this size
    out = {"workload": "BASELINE configs[3]: 2 097 152-triangle height field + 16 reflective spheres + SkyBox, 3840x2160, MaxRecursion 4, lone frames",
           "rays_per_frame": best["rays_per_frame"], "ms_per_frame": best["ms_per_frame"], "value": best["value"], "unit": "Mrays/s", "frames": frames,
           "frame_checksum": best["frame_checksum"], "frames_identical": runs["trees_built_on_host"]["frame_checksum"] == best["frame_checksum"],
           "trees_built_on_host": runs["trees_built_on_host"], "trees_built_on_device": runs["trees_built_on_device"]}
    rf = hbm_roofline(PMC_CFG4)
    if rf:
        out["roofline"] = rf
    return out


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a CHILD process (torch.distributed.run; never an
    exec), relay rank 0's JSON line and the exit code.  Runs before torch is imported or HIP is touched in this process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    if r.returncode != 0:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the {args.gpus}-rank run failed (exit code {r.returncode}); no result")
    if len(lines) != 1:
        raise SystemExit(f"bench.py --gpus {args.gpus}: expected one JSON line from rank 0, got {len(lines)}")
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="the timed burst of --steps frames is run this many times; the median is reported")
    ap.add_argument("--inflight", type=int, default=0, help="calls in flight (one context + HIP stream each); 0 = 3")
    ap.add_argument("--batch", type=int, default=0, help="frames traced per call (rthip_render_batch_device: one set of launches for "
                    "the rays of all of them); 0 = 8 per GPU (8, 16, 32, 64 on 1, 2, 4, 8 GPUs; the library's limit is 64)")
    ap.add_argument("--latency", action="store_true", help="(kept for old command lines: lone-frame latency is measured by default now)")
    ap.add_argument("--no-extras", action="store_true", help="skip what runs AFTER the timed region besides the roofline launches: the lone-frame "
                    "latency, the D2H-inclusive frames and the config-4 frames.  For `rocprofv3 --kernel-trace --stats -- python3 bench.py "
                    "--no-extras`: every k_trace_quad launch of the process then has the timed region's size, so the profiler's average "
                    "duration of that kernel is comparable with roofline.avg_launch_ms")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 object (2 M-triangle scene: ~3 s of host build + 0.6 GB upload)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config2 / config3_closeup / config5 objects")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 object (10.5 M triangles at 8K on one GPU: ~40 s of scene construction)")
    ap.add_argument("--collective", choices=["gather", "all_gather"], default="gather", help="framebuffer exchange: gather to rank 0 "
                    "(each peer sends its part once) or all-gather (every rank ends up with the frames)")
    ap.add_argument("--gather-dst", choices=["rotate", "0"], default="rotate", help="with --collective gather: the rank that receives and "
                    "assembles call k's frames - k %% N (default: the receive traffic and the assemble kernel go round the ranks, the frames "
                    "of a call stay in that rank's HBM) or always rank 0 (which then does N times the assembling and bounds the scaling)")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL gather + assemble path even with one rank")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="torch.distributed.run with N ranks that all use GPU 0 and exchange "
                    "over gloo (staged through host memory): walks the multi-rank control flow of this script on a one-GPU box; the "
                    "number it prints is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a run on fewer ranks than asked for is not reported as {args.gpus} GPUs")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the Raycast path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible: refusing to share devices (use --rehearse-on-one-gpu "
                         "to walk the multi-rank control flow on one GPU)")
    torch.cuda.set_device(local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    elif args.force_gather:  # rehearsal of the multi-GPU data path on one GPU
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))

    from raytracing_amd import device, host, parallel, scenes
    sd, _ = scenes.config(3)
    hs = host.HostScene(sd, device_index=local_rank)
    desc = hs.flatten()
    cam = hs.camera(W, H)
    gather = world > 1 or args.force_gather
    prm = device.make_params(W, H, MAX_REC, part_index=rank, part_count=world, layout=1 if gather else 0)

    part_px = device.part_pixels(prm)
    # 32 frames per call on one GPU (32 x 2.07 M primary rays = one workspace chunk of 25 GB): a call is a chain of 14 launches
    # each as long as its slowest wave (0.3 ms per call whatever it carries, tools/gpu_floor.py), so the more frames a call
    # carries the less of it every frame pays - 8 / 16 / 32 frames per call: 11 230 / 11 435 / 11 770 Mrays/s
    # (tools/sweep_bench_chunk.sh).  A rank's share of a frame shrinks with the rank count, so N GPUs take 32 N frames per call
    # up to the library's limit of 64.
    # ... unless the timed region is shorter than two such calls: a burst of K frames is then traced as two calls of K / 2 in
    # flight together (the second call's kernels fill the gaps of the first one's latency-bound levels, and its gather overlaps):
    # 20 frames on one GPU: 2 x 10 -> 11 320 Mrays/s, 3 x 7 -> 11 220, 1 x 20 -> 10 820 (tools/sweep_bench_chunk.sh)
    B_full = min(device.MAX_BATCH, 32 * world)
    B = args.batch if args.batch > 0 else max(1, min(B_full, (args.steps + 1) // 2))  # (200 frames: 6 x 32 + 8 beats 8 x 23 + 16 and 6 x 29 + 26)
    if B > device.MAX_BATCH:
        raise SystemExit(f"--batch {B} > {device.MAX_BATCH}")
    # the shape of call the roofline objects are measured on (counting context, per-launch events, PMC summaries under profiles/):
    # the full-size call, whatever --steps makes of the timed region
    B_R = B if (args.batch > 0 or args.rehearse_on_one_gpu) else B_full
    # Every frame has a camera of its own: a small orbit around the model (+-0.12 rad about the y axis at the scene's camera
    # distance, looking at the origin; period 40 frames; frame 0 = the scene's own camera (0,0,5) -> (0,0,-1)).  The cameras are
    # not part of the launch graph the library replays (they travel through its page-locked camera ring), so this is what the
    # Sandbox preview's moving camera gets (Sandbox/Sandbox.cpp:315-359).
    import math
    PATH = 40
    dist0 = math.sqrt(sum(x * x for x in sd.camera_pos))
    path = []
    for f in range(PATH):
        th = 0.12 * math.sin(2.0 * math.pi * f / PATH)
        path.append(hs.camera_at((dist0 * math.sin(th), 0.0, dist0 * math.cos(th)), (0.0 - math.sin(th), 0.0, 0.0 - math.cos(th)), W, H))
    hs.camera_at(sd.camera_pos, sd.camera_dir, W, H)

    def cam_run(start, n):
        return (device.Camera * n)(*[path[(start + j) % PATH] for j in range(n)])

    # one untimed call on a counting context: ray and BVH-traversal counts of one call of B frames (this rank's share)
    os.environ["RTHIP_COUNTERS"] = "1"
    cctx = device.Context(local_rank)
    os.environ["RTHIP_COUNTERS"] = "0"
    cctx.upload(desc)
    scratch = torch.zeros(B_R * (part_px if gather else W * H), dtype=torch.int32, device="cuda")
    cctx.render_batch_device(cam_run(0, B_R), prm, scratch.data_ptr(), sync=True)
    cst = cctx.stats()
    aflops = algorithmic_flops(cst)  # per call (the first B_R cameras of the path)
    path_rays = []                     # this rank's rays of every camera of the path (on the counting context too: its kernels are
    for f in range(PATH):              # instantiations of their own, so a kernel trace of this script keeps the timed kernels apart)
        cctx.render_batch_device(cam_run(f, 1), prm, scratch.data_ptr(), sync=True)
        st1 = cctx.stats()
        path_rays.append(int(st1.rays_primary + st1.rays_shadow + st1.rays_reflect))
    my_rays_timed = sum(path_rays[f % PATH] for f in range(args.steps))  # exactly the frames of one timed burst
    cctx.close()
    del scratch

    # Timed contexts: `inflight` frames are in flight at once, each on its own context (own HIP stream, own
    # workspace, own output buffers), so that the latency-bound tail of one frame's kernels overlaps with the
    # next frame's work.  Every frame is traced completely; frame k uses context k % inflight.
    # A call traces a run of B frames (one camera each; here the same camera B times, every frame traced in full) as
    # one set of launches: the ~20 dependent launches of a frame then work on B frames' rays, which is what keeps a
    # GPU busy when its share of a 1080p frame is an eighth of it.  One RCCL all-gather per call.
    F = args.inflight if args.inflight > 0 else 3
    ctxs, mines, gathers, frames, exts = [], [], [], [], []
    for k in range(F):
        c = device.Context(local_rank)
        c.upload(desc)
        ctxs.append(c)
        fr = torch.zeros(B * W * H, dtype=torch.int32, device="cuda")
        frames.append(fr)
        mines.append(torch.zeros(B * part_px, dtype=torch.int32, device="cuda") if gather else fr)
        gathers.append(torch.zeros(B * part_px * world, dtype=torch.int32, device="cuda") if gather else None)
        exts.append(torch.cuda.ExternalStream(c.stream()))
    comm = torch.cuda.current_stream()

    def call(k, n, start=0):
        """call k: n (<= B) frames, the cameras start .. start + n - 1 of the path, on context k % F"""
        j = k % F
        c = ctxs[j]
        c.render_batch_device(cam_run(start, n), prm, mines[j].data_ptr(), sync=False)
        if gather:  # device-side ordering only: render -> RCCL all-gather (torch's stream) -> assemble (context's stream)
            done = torch.cuda.Event()
            done.record(exts[j])
            comm.wait_event(done)
            mine, gath = mines[j][:n * part_px], gathers[j][:n * part_px * world]
            complete = parallel.collect_parts(mine, gath, args.collective, dst=k % world if args.gather_dst == "rotate" else 0)
            gathered_ev = torch.cuda.Event()
            gathered_ev.record(comm)
            exts[j].wait_event(gathered_ev)  # (every rank: its part buffer is free again only once the exchange has read it)
            if complete:
                c.assemble_batch(prm, n, gath.data_ptr(), frames[j].data_ptr(), sync=False)

    def run_frames(n_frames):
        """exactly n_frames frames (cameras 0 .. n_frames - 1 of the path): calls of B, then one shorter call for the remainder"""
        k, f = 0, 0
        while f < n_frames:
            n = min(B, n_frames - f)
            call(k, n, f)
            k += 1
            f += n

    def sync_local():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()

    def sync_all():
        sync_local()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up: at least five calls per context of every call size the timed region uses (workspaces allocated; the library
    # captures a call's launch graph the second time it sees that launch sequence and replays it from the third - and the
    # sequence changes once, when the first calls have reported their list lengths and only one walk kernel per level is
    # launched from then on: so two calls per context, a wait for their reports, then three more)
    run_frames(2 * F * B)
    sync_local()
    run_frames(max(args.warmup, 3 * F * B))
    sync_local()
    if args.steps % B:
        for j in range(2 * F):
            call(j, args.steps % B)
        sync_local()
        for j in range(3 * F):
            call(j, args.steps % B)
        sync_local()
    # The timed region: exactly --steps frames between a barrier + synchronize (all ranks start together) and this rank's own
    # synchronize; the slowest rank's time counts (MAX over ranks below), so no second barrier sits inside the region.
    # Repeated --repeats times (default 5); `value` is the MEDIAN burst, every sample is in the line.
    samples = []
    for _ in range(max(1, args.repeats)):
        sync_all()
        t0 = time.perf_counter()
        run_frames(args.steps)
        sync_local()
        samples.append(time.perf_counter() - t0)

    # ---- after the timed region -------------------------------------------------------------------------------------
    # (1) per-launch duration of the dominant kernel (HIP events on the contexts' own streams, plain launches instead of the
    #     replayed graph): of a launch that has the GPU to itself, and of launches in the timed region's regime (F calls in
    #     flight, one host thread per context so that F sets of launches overlap) - the figure `rocprofv3 --kernel-trace
    #     --stats -- python3 bench.py --no-extras` reports as the kernel's average duration.
    import threading
    for c in ctxs:
        c.synchronize()
    for c in ctxs[1:]:  # (their workspaces - 25 GB each at 32 frames per call - make room for the profiling contexts')
        c.close()
    del ctxs[1:]        # (what follows uses context 0 only)
    pouts = [torch.zeros(B_R * (part_px if gather else W * H), dtype=torch.int32, device="cuda") for _ in range(F)]
    os.environ["RTHIP_PROFILE_LAUNCHES"] = "1"
    pctxs = [device.Context(local_rank) for _ in range(F)]
    os.environ["RTHIP_PROFILE_LAUNCHES"] = "0"
    for c in pctxs:
        c.upload(desc)
    for _ in range(3):
        pctxs[0].render_batch_device(cam_run(0, B_R), prm, pouts[0].data_ptr(), sync=True)
    st = pctxs[0].stats()
    walk = walk_kernel(pctxs[0])
    launch_ms = st.ms_trace / max(1, st.launches_trace)
    kfl, kms = kernel_flops(cst, *material_flops(desc), walk), kernel_ms(st, walk)
    over = []

    def overlapped(j):
        for _ in range(4):
            pctxs[j].render_batch_device(cam_run(0, B_R), prm, pouts[j].data_ptr(), sync=True)
            s2 = pctxs[j].stats()
            over.append((s2.ms_trace, s2.launches_trace))
    ths = [threading.Thread(target=overlapped, args=(j,)) for j in range(F)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    launch_ms_overlapped = sum(m for m, _ in over) / max(1, sum(n for _, n in over))
    for c in pctxs:
        c.close()
    del pouts

    # (2) what the reference's own contract costs: ONE frame with nothing else in flight (latency), and the same frame
    #     through rthip_render into page-locked host memory (RenderSync hands back host pixels: D2H-inclusive)
    lat, d2h = [], []
    cfg4 = None
    if not args.no_extras:
        for f in range(3):
            call(0, 1, f)
            sync_all()
        for f in range(3, 3 + 11):   # a lone frame whose camera MOVED since the previous one (the last one is frame 0's camera again)
            t1 = time.perf_counter()
            call(0, 1, f if f < 13 else 0)
            sync_all()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat.sort()
        if world == 1:
            full = device.make_params(W, H, MAX_REC)
            pinned = ctxs[0].host_pixels(W, H)
            for _ in range(3):
                ctxs[0].render(cam, full, out=pinned)
            for _ in range(10):
                t1 = time.perf_counter()
                ctxs[0].render(cam, full, out=pinned)
                d2h.append((time.perf_counter() - t1) * 1e3)
            d2h.sort()
            ctxs[0].host_free(pinned)
    others = {}
    if world == 1 and rank == 0 and not args.no_extras and not args.no_config4:
        cfg4 = config4_line(local_rank)
    if world == 1 and rank == 0 and not args.no_extras and not args.no_other_configs:
        # the other BASELINE configs' GPU numbers: Demo 1 (noise materials, no mesh), the close-up of the bunny the
        # survey asks for (camera (0,0,2.2): the mesh fills the view, the walk dominates), and config 5 on ONE GPU
        others["config2"] = config_object("BASELINE configs[1]: Demo 1 spheres (Marble / Worley / Perlin materials) + checker plane + SkyBox, 1920x1080, "
                                          "MaxRecursion 3, lone frames", *scenes.config(2), local_rank)
        others["config3_closeup"] = config_object("BASELINE configs[2] from camera (0,0,2.2) (SURVEY.md §8d: the bunny fills the view, kd / BVH traversal "
                                                  "dominates), 1920x1080, MaxRecursion 3, lone frames", *scenes.config(3, cam_z=2.2), local_rank)
        if not args.no_config5:
            others["config5"] = config_object("BASELINE configs[4] on ONE GPU: five rotated / scaled 2 097 152-triangle height fields (10 485 760 triangles) + 16 "
                                              "reflective spheres + checker plane + SkyBox, 7680x4320, MaxRecursion 8, lone frames (one workspace chunk of 18 GB)",
                                              *scenes.config(5), local_rank, frames=3, hbm_pmc=hbm_roofline(PMC_CFG5))
    red_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"
    rays_t = torch.tensor([float(my_rays_timed), float(path_rays[0])], dtype=torch.float64, device=red_dev)
    dt_t = torch.tensor(samples, dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(rays_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)   # per burst: the slowest rank
    rays_timed, rays_frame0 = float(rays_t[0].item()), float(rays_t[1].item())  # all ranks: one burst of --steps frames; frame 0 alone
    samples = sorted(float(x) for x in dt_t.tolist())
    dt = samples[len(samples) // 2]
    rays_frame = rays_timed / args.steps
    checksum = int(frames[0][:W * H].to(torch.int64).sum().item()) if rank == 0 else 0  # the last lone frame: camera 0 of the path

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = rays_timed / dt / 1e6
        n_launch = max(1, st.launches_trace)
        tflops = (aflops / n_launch) / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
        traffic = measured_traffic(B_R, n_launch, kernel=walk)
        hbm_gbs = traffic / (launch_ms * 1e-3) / 1e9 if (traffic and launch_ms > 0) else None
        med = lambda v: v[len(v) // 2] if v else None
        out = {
            "metric": "Mrays/sec (primary+bounce) 1080p OBJ KD-tree scene", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            **({"rehearsal": "all ranks on one GPU, gloo exchange through host memory: not a measurement"} if args.rehearse_on_one_gpu else {}),
            "config": {"workload": "BASELINE configs[2]: Demo 4 bunny.obj (4968 tris, kd-tree) + checker plane + SkyBox sun, "
                                   "1920x1080, MaxRecursion 3, camera (0,0,5)", "rays_per_frame": rays_frame,
                       "rays_frame0": int(rays_frame0),
                       "partition": f"interleaved 32x8 screen tiles over {world} GPU(s)" + (f", RCCL {args.collective}" + (f" (to rank {'k % N' if args.gather_dst == 'rotate' else '0'})" if args.collective == "gather" else "") + " + device assemble" if gather else ""),
                       "frames_per_call": B, "calls_in_flight": F,
                       "camera": f"every frame its own camera: orbit of +-0.12 rad about the y axis, period {PATH} frames, frame 0 = the scene's camera; "
                                 "one replayed launch graph serves all of them (cameras travel through the library's camera ring)",
                       "timed_bursts": len(samples), "ms_per_step_samples": [x / args.steps * 1e3 for x in samples],
                       "single_frame_latency_ms": med(lat), "single_frame_mrays_s": (rays_frame / med(lat) / 1e3) if lat else None,
                       "ms_per_frame_with_d2h": med(d2h), "mrays_s_with_d2h": (rays_frame / med(d2h) / 1e3) if d2h else None,
                       "frame_checksum": checksum, "frame_checksum_expected": FRAME0_CHECKSUM,
                       "frame_checksum_ok": (checksum == FRAME0_CHECKSUM) if not args.no_extras else None},
            # The dominant kernel is divergent fp64 / fp32 VALU work over an L2-resident BVH (0.8 MB): its roof is the vector ALU,
            # not HBM.  achieved = algorithmic flops of one launch / its duration when it has the GPU to itself.
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_VALU_PEAK_TFLOPS, "frac_of_unfused_peak": tflops / FP64_VALU_PEAK_UNFUSED_TFLOPS,
                         "traffic": traffic,
                         "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None,
                                 "note": "PMC-measured HBM bytes per launch (`traffic`) / avg_launch_ms"},
                         "kernel": walk, "launches_per_call": int(st.launches_trace), "avg_launch_ms": launch_ms,
                         "avg_launch_ms_calls_in_flight": launch_ms_overlapped,
                         "algorithmic_flops_per_call_rank0": int(aflops),
                         "flop_weights": {"bvh_step_4_boxes_fp32": FLOPS_NODE, "triangle_test_fp64": FLOPS_TRI, "shape_entry_fp64": FLOPS_SHAPE,
                                          "shape_entry_rejected_by_ball_test": FLOPS_BALL, "hit_record_fp64": FLOPS_RAY},
                         "counts_per_call_rank0": {"bvh_steps": int(cst.kd_nodes_visited), "triangle_tests": int(cst.kd_tris_tested),
                                                   "shape_entries": int(cst.shape_tests_full), "shape_entries_rejected_by_ball_test": int(cst.shape_tests - cst.shape_tests_full),
                                                   "rays": int(cst.rays_hard)},
                         "frames_per_call": B_R,
                         "pmc": measured_valu(kernel=walk),
                         # every ray kernel of the timed regime, not only the dominant one: algorithmic flops of one call (the counting
                         # context above) over the kernel's summed launch durations in that call (plain launches alone on the GPU)
                         "kernels": [{"kernel": k, "algorithmic_flops_per_call": kfl[k], "ms_per_call": kms[k], "launches_per_call": int(st.launches_trace),
                                      "achieved": kfl[k] / (kms[k] * 1e-3) / 1e12 if kms[k] > 0 else None,
                                      "frac": kfl[k] / (kms[k] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS if kms[k] > 0 else None,
                                      "frac_of_unfused_peak": kfl[k] / (kms[k] * 1e-3) / 1e12 / FP64_VALU_PEAK_UNFUSED_TFLOPS if kms[k] > 0 else None,
                                      "share_of_call": kms[k] / sum(kms.values()),
                                      "pmc": measured_pmc(k + ("<true>" if k == "k_shade" else "<false>"), PMC_CFG3)} for k in kfl],
                         "note": "peak = 78.6 TFLOP/s fp64 vector with FMA; the kernels are built with -ffp-contract=off (bit parity with the "
                                 "non-fused reference), so an instruction is one flop and the usable peak is 39.3 (frac_of_unfused_peak).  "
                                 "`pmc` is the measured side: share of SIMD cycles issuing VALU x active lanes per instruction.  "
                                 "avg_launch_ms is a launch alone on the GPU; avg_launch_ms_calls_in_flight is with as many calls in flight as the timed "
                                 "region keeps (what rocprofv3 --stats of `bench.py --no-extras` averages); every figure here is for calls of "
                                 "`frames_per_call` frames, the full-size call, whatever --steps made of the timed region"},
        }
        if cfg4 is not None:
            out["config4"] = cfg4
        out.update(others)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
