// rthip_bvh.hpp — device-side build of a mesh's 4-wide BVH (rthip_bvh.hip), used by rthip_upload_scene_ex.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "rthip.h"

struct rthip_device_bvh_info {
    uint32_t n_nodes;    // wide nodes written (mesh-relative indices, root = 0)
    uint32_t depth;      // deepest wide node (root = 0): sizes the traversal stacks
    float abs_max;       // >= max |coordinate| of the mesh
    rthip_bvhnode root;  // host copy of node 0 (bounding ball of the shape, roundtrip_bounds)
};
// d_tris: n records in mesh order (device).  Writes d_bvh_tris / d_bvh_index (n entries each, device, leaf order) and
// allocates *d_nodes_out with hipMalloc (caller frees; info->n_nodes entries are valid).  Synchronises `stream`.
// Returns 0, RTHIP_E_HIP or RTHIP_E_UNSUPPORTED (deeper than RTHIP_BVH_MAX_DEPTH) with a message in err.
int rthip_bvh_build_device(hipStream_t stream, const rthip_tri *d_tris, uint32_t n, rthip_tri *d_bvh_tris, uint32_t *d_bvh_index,
                           rthip_bvhnode **d_nodes_out, rthip_device_bvh_info *info, char *err, size_t err_len);
