// scene.cpp — host-side Compile() of the re:: mirror: transforms (Scene.cpp:239-264), triangle
// caches (Scene.cpp:509-523) and the kd-tree with the reference's splitting rules
// (Scene.cpp:18-95), emitted directly in the flat pre-order layout the device walks.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <future>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>

#include "mesh_build.hpp"
#include "re/scene.hpp"
#include "re/threads.hpp"
#include "rthip.h"

namespace re {

const UniformMaterial UniformMaterial::OpaqueWhite = UniformMaterial(Color::White, 1.0);  // Material.cpp:3-7
const UniformMaterial UniformMaterial::OpaqueBlack = UniformMaterial(Color::Black, 1.0);

// ---- noise tables -------------------------------------------------------------------------
Perlin::Perlin(real domainSize) : Noise(domainSize) { NewSeed(); }
void Perlin::NewSeed() {
    for (size_t i = 0; i < SeedSize; i++) m_Seed[i] = (uint8_t)(std::rand() % SeedSize);
}
void Perlin::SetSeedTable(const uint8_t *t) { std::copy(t, t + SeedSize, m_Seed.begin()); }

template <typename T> static T Random(T lo, T hi) { return (T)(lo + (hi - lo) * ((double)std::rand() / RAND_MAX)); }  // Common.h:17-20

Worley::Worley(real domainSize, int divisions) : Noise(domainSize), m_Divisions(divisions) {
    m_Step = 1.0 / divisions;
    m_Points.reserve((size_t)divisions * divisions * divisions);
    for (int z = 0; z < divisions; z++)
        for (int y = 0; y < divisions; y++)
            for (int x = 0; x < divisions; x++) {
                real fx = x, fy = y, fz = z;
                real px = Random(fx * m_Step, (fx + 1) * m_Step);  // X, Y, Z drawn in this order
                real py = Random(fy * m_Step, (fy + 1) * m_Step);
                real pz = Random(fz * m_Step, (fz + 1) * m_Step);
                m_Points.push_back({px, py, pz});
            }
}
void Worley::SetPoints(const double *xyz) {
    for (size_t i = 0; i < m_Points.size(); i++) m_Points[i] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
}

Marble::Marble(real domainSize, real frequency, real turbolence)
    : Noise(domainSize), m_Perlin(new Perlin(domainSize)), m_Frequency(frequency), m_Turbolence(turbolence) {}

SkyBox::SkyBox(Color color0, Color color1, const std::shared_ptr<Light> &sun)
    : m_Perlin(std::make_shared<Perlin>(1)), m_SkyColor0(color0), m_SkyColor1(color1), m_Sun(sun) {}

// ---- graph --------------------------------------------------------------------------------
SceneNode::SceneNode() { AddComponent<Transform>(); }

std::shared_ptr<SceneNode> SceneNode::AddChild() {
    auto child = std::make_shared<SceneNode>();
    child->m_Parent = this;
    m_Children.push_back(child);
    return child;
}

void SceneNode::Compile() {
    for (auto &c : m_Components) c->Compile();
    for (auto &child : m_Children) child->Compile();
}

void Transform::Compile() {
    Matrix4 prev = Matrix4::Identity, prevInverse = Matrix4::Identity;
    if (SceneNode *parent = m_Owner->GetParent()) {
        auto pt = parent->GetComponentOfType<Transform>();
        pt->GetTransform(prev);
        pt->GetInverseTransform(prevInverse);
    }
    const Matrix4 t = Matrix4::GetTranslation(Position), r = Matrix4::GetRotation(Rotation), s = Matrix4::GetScale(Scale);
    const Matrix4 it = Matrix4::GetTranslation(Vector3::Zero - Position), ir = r.Transpose(), is = Matrix4::GetScale(Vector3::One / Scale);
    m_Transform = prev * t * r * s;                  // ((prev*t)*r)*s
    m_InverseTransform = is * ir * it * prevInverse;  // ((is*ir)*it)*prevInverse
    m_NormalTransform = m_InverseTransform.Transpose();
}

Shape::Shape(SceneNode *owner) : Component(owner) {
    static unsigned int nextID = 1;
    m_ID = nextID;
    nextID = std::max(1u, nextID + 1);
}

// ---- mesh ---------------------------------------------------------------------------------
void Triangle::Update() {
    FaceNormal = Cross(Vertices[1] - Vertices[0], Vertices[2] - Vertices[1]).Normalized();
    Edges = {Vertices[1] - Vertices[0], Vertices[2] - Vertices[0], Vertices[2] - Vertices[1]};
    m_D00 = Edges[0] ^ Edges[0];
    m_D01 = Edges[0] ^ Edges[1];
    m_D11 = Edges[1] ^ Edges[1];
    m_InvDen = 1.0 / (m_D00 * m_D11 - m_D01 * m_D01);
}

namespace {
// the mesh's triangles as the builders of mesh_build.hpp read them
struct MirrorTriView {
    const std::vector<Triangle> &t;
    size_t size() const { return t.size(); }
    bool empty() const { return t.empty(); }
    double V(uint32_t tri, int vertex, unsigned axis) const { return t[tri].Vertices[vertex][axis]; }
};
}  // namespace

size_t Mesh::DeviceBvhMinTriangles() {
    if (const char *e = std::getenv("RE_BVH_DEVICE_MIN_TRIS")) {
        const long long v = atoll(e);
        return v <= 0 ? (size_t)-1 : (size_t)v;
    }
    return (size_t)1 << 18;
}
bool Mesh::BvhBuiltOnDevice() const { return !ForceHostBvh && m_Triangles.size() >= DeviceBvhMinTriangles(); }
void Mesh::BuildHostBvh() {
    ForceHostBvh = true;
    if (m_Invalidated || !m_Bvh.nodes.empty() || m_Triangles.empty()) return;  // not compiled yet (Compile will build it) / already there
    const MirrorTriView view{m_Triangles};
    rtbuild::BvhBuilder<MirrorTriView> bvh{view, m_Bvh, {}, {}, {}, {}};
    bvh.Run();
    static std::atomic<uint64_t> late{1ull << 62};
    m_Revision = ++late;  // the flattened geometry changes: a renderer must not keep what it uploaded
}

size_t Mesh::DeviceKdMinTriangles() {
    if (const char *e = std::getenv("RE_KD_DEVICE_MIN_TRIS")) {
        const long long v = atoll(e);
        return v <= 0 ? (size_t)-1 : (size_t)v;
    }
    return (size_t)1 << 18;
}
bool Mesh::KdBuiltOnDevice() const { return !ForceHostKd && m_Triangles.size() >= DeviceKdMinTriangles(); }
void Mesh::BuildHostKd() {
    ForceHostKd = true;
    if (m_Invalidated || !m_KdTree.nodes.empty()) return;  // not compiled yet (Compile will build it) / already there
    BuildKdTree();
    static std::atomic<uint64_t> late{1ull << 61};
    m_Revision = ++late;  // the flattened geometry changes: a renderer must not keep what it uploaded
}
void Mesh::BuildKdTree() {
    m_KdTree = KdTreeFlat();
    std::vector<uint32_t> all(m_Triangles.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
    const MirrorTriView view{m_Triangles};
    const rtbuild::Box3 root = {{m_BoundingBox.Min.X, m_BoundingBox.Min.Y, m_BoundingBox.Min.Z}, {m_BoundingBox.Max.X, m_BoundingBox.Max.Y, m_BoundingBox.Max.Z}};
    rtbuild::KdBuilder<MirrorTriView> builder{view, m_KdTree};
    builder.Build(all, root, 0, BuildForkDepth());
}

Triangle &Mesh::AddTriangle() {
    m_Triangles.emplace_back();
    return m_Triangles.back();
}

void Mesh::Compile() {
    if (!m_Invalidated) return;
    // the bounding box starts at (0,0,0)/(0,0,0) and is never reset (Scene.cpp:391-406, Common.h:212)
    // (min / max are exact and order-independent: slices on threads give the same box)
    std::mutex mu;
    const BoundingBox start = m_BoundingBox;  // read once, before the slices run (they merge into m_BoundingBox under `mu`)
    ParallelSlices(m_Triangles.size(), [&](size_t i0, size_t i1) {
        BoundingBox bb = start;
        Vector3 &mn = bb.Min, &mx = bb.Max;
        for (size_t i = i0; i < i1; i++) {
            Triangle &t = m_Triangles[i];
            t.Update();
            for (const Vector3 &v : t.Vertices) {
                mn.X = std::min(mn.X, v.X); mn.Y = std::min(mn.Y, v.Y); mn.Z = std::min(mn.Z, v.Z);
                mx.X = std::max(mx.X, v.X); mx.Y = std::max(mx.Y, v.Y); mx.Z = std::max(mx.Z, v.Z);
            }
        }
        std::lock_guard<std::mutex> g(mu);
        Vector3 &gmn = m_BoundingBox.Min, &gmx = m_BoundingBox.Max;
        gmn.X = std::min(gmn.X, mn.X); gmn.Y = std::min(gmn.Y, mn.Y); gmn.Z = std::min(gmn.Z, mn.Z);
        gmx.X = std::max(gmx.X, mx.X); gmx.Y = std::max(gmx.Y, mx.Y); gmx.Z = std::max(gmx.Z, mx.Z);
    });
    auto T0 = std::chrono::steady_clock::now();
    // The reference kd-tree of a big mesh is left to the device as well (RTHIP_UPLOAD_BUILD_KD: the same splits, level by level
    // with all nodes of a level side by side; byte-identical arrays).  RE_KD_DEVICE_MIN_TRIS moves the limit (0 = always here).
    m_KdTree = KdTreeFlat();
    if (!KdBuiltOnDevice()) BuildKdTree();
    const MirrorTriView view{m_Triangles};
    auto T1 = std::chrono::steady_clock::now();
    // The 4-wide BVH of a big mesh is left to the device (rthip_upload_scene_ex + RTHIP_UPLOAD_BUILD_BVH: milliseconds instead
    // of most of a second, and its leaf-order triangle copy is never made or uploaded); small meshes keep the host's
    // binned-SAH build, which costs nothing at their size and walks a little faster.  RE_BVH_DEVICE_MIN_TRIS moves the limit
    // (0 = always on the host).
    m_Bvh = BvhFlat();
    if (!BvhBuiltOnDevice()) {
        rtbuild::BvhBuilder<MirrorTriView> bvh{view, m_Bvh, {}, {}, {}, {}};
        bvh.Run();
    }
    auto T2 = std::chrono::steady_clock::now();
    if (getenv("RE_BUILD_TIMING")) fprintf(stderr, "Mesh::Compile %zu tris: kd %.3f s, bvh %.3f s\n", m_Triangles.size(), std::chrono::duration<double>(T1 - T0).count(), std::chrono::duration<double>(T2 - T1).count());
    m_Invalidated = false;
    static std::atomic<uint64_t> revisions{0};
    m_Revision = ++revisions;
}

}  // namespace re
