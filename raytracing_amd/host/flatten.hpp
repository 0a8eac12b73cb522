// flatten.hpp — re::Scene (compiled) -> rthip_scene_desc POD blobs (include/rthip.h).
// Runs at render start on the render thread, after Scene::Compile (reference Raytracer.cpp:42).
#pragma once
#include <string>
#include <vector>

#include "re/scene.hpp"
#include "rthip.h"

namespace re {

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
    std::vector<double> tri_verts;    // 9 per triangle, only when needs_device_kd
    std::vector<double> mesh_bounds;  // 6 per mesh
    std::vector<rthip_matop> materials;
    std::vector<rthip_noise> noises;
    std::vector<uint8_t> perlin_tables;
    std::vector<double> worley_points;
    std::vector<rthip_light> lights;
    rthip_scene_desc desc;
    std::vector<SceneNode *> shape_nodes;  // pre-order shape ordinal -> node
    // (mesh, Mesh::GetRevision()) in meshes[] order: identifies the geometry arrays above
    std::vector<std::pair<const Mesh *, uint64_t>> mesh_signature;
    bool needs_device_bvh = false;  // some mesh comes without a BVH: upload with RTHIP_UPLOAD_BUILD_BVH (see Mesh::BvhBuiltOnDevice)
    bool needs_device_kd = false;   // some mesh comes without its kd-tree: upload with RTHIP_UPLOAD_BUILD_KD (see Mesh::KdBuiltOnDevice)
    bool geometry_kept = false;  // flattened against `previous`: geometry arrays empty, desc pointers null, counts kept

    // drops the geometry arrays (they can be hundreds of MB) but keeps what a later FlattenScene(..., this) needs
    void ReleaseGeometry();
};

// Returns 0 or RTHIP_E_UNSUPPORTED (user-defined Shape/Material/Noise/Background subclass, nesting
// too deep, ...) with a message in `error`.  The scene must have been compiled.
// previous != nullptr: if the scene's meshes are exactly the ones `previous` was flattened from (same objects, same
// revisions, same order) only the small tables are rebuilt (shapes / transforms, materials, noises, lights,
// background); the mesh table is copied and the geometry arrays are left out for rthip_update_scene.
int FlattenScene(Scene *scene, FlatScene &out, std::string &error, const FlatScene *previous = nullptr);

// Camera basis of AbstractRaycaster::CreateScreenRay (Raytracer.cpp:102-110) computed once per frame.
// upload flags (rthip_upload_scene_ex) that go with a flattened scene
inline uint32_t UploadFlags(const FlatScene &f) {
    return f.geometry_kept ? RTHIP_UPLOAD_KEEP_GEOMETRY : ((f.needs_device_bvh ? RTHIP_UPLOAD_BUILD_BVH : 0u) | (f.needs_device_kd ? RTHIP_UPLOAD_BUILD_KD : 0u));
}
// every mesh below `node` gets a host-built BVH and kd-tree from now on (after the device refused to build one: deeper than its stacks, ...)
void ForceHostBvhEverywhere(SceneNode *node);

void MakeCamera(const Scene *scene, size_t width, size_t height, real fovY, rthip_camera &cam);
void MakeCamera(const Vector3 &position, const Vector3 &direction, size_t width, size_t height, real fovY, rthip_camera &cam);

}  // namespace re
