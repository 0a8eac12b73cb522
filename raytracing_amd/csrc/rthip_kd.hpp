// rthip_kd.hpp — device-side build of a mesh's REFERENCE kd-tree (rthip_kd.hip), used by rthip_upload_scene_ex.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "rthip.h"

struct rthip_device_kd_info {
    uint32_t n_nodes;         // nodes written (pre-order, mesh-relative links)
    uint32_t n_refs;          // leaf references written
    uint32_t max_depth;       // deepest node (root = 0)
    uint32_t plane_count[3];  // sorted unique plane coordinates per axis; the three runs lie one after the other in *d_planes
    float device_ms;          // device time of the build (events around it)
    uint32_t levels;
};
// d_verts: n triangles x 3 vertices x 3 doubles in mesh order (device); root_bounds: Min xyz, Max xyz of Mesh::m_BoundingBox
// as Mesh::Compile leaves it (host).  Allocates the four output arrays with hipMalloc (caller frees): nodes / bounds (6 per
// node) / refs / planes in the layout of include/rthip.h, byte-identical to what rtbuild::KdBuilder + the flattener produce for
// the same triangles.  Synchronises `stream`.  Returns 0, RTHIP_E_HIP or RTHIP_E_UNSUPPORTED with a message in err.
int rthip_kd_build_device(hipStream_t stream, const double *d_verts, uint32_t n, const double root_bounds[6], rthip_kdnode **d_nodes,
                          double **d_bounds, uint32_t **d_refs, double **d_planes, rthip_device_kd_info *info, char *err, size_t err_len);
