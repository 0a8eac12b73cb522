// scene.cpp — host-side Compile() of the re:: mirror: transforms (Scene.cpp:239-264), triangle
// caches (Scene.cpp:509-523) and the kd-tree with the reference's splitting rules
// (Scene.cpp:18-95), emitted directly in the flat pre-order layout the device walks.
#include <algorithm>
#include <cstdlib>

#include "re/scene.hpp"
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

Triangle &Mesh::AddTriangle() {
    m_Triangles.emplace_back();
    return m_Triangles.back();
}

namespace {
// Builder: same decisions as KDTreeTriangle's constructor (Scene.cpp:18-95) on triangle INDEX lists
// (the reference copies 272-byte triangles by value; the order of each list is the same).
struct KdBuilder {
    const std::vector<Triangle> &tris;
    KdTreeFlat &out;

    void Leaf(uint32_t me, const std::vector<uint32_t> &list) {
        out.nodes[me].link = (uint32_t)out.refs.size();
        out.nodes[me].info = RTHIP_KD_LEAF | ((uint32_t)list.size() << RTHIP_KD_COUNT_SHIFT);
        out.refs.insert(out.refs.end(), list.begin(), list.end());
    }
    uint32_t Build(const std::vector<uint32_t> &list, const BoundingBox &bounds, unsigned depth) {
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.push_back({0.0, 0, 0});
        out.depth.push_back(depth);
        const double b[6] = {bounds.Min.X, bounds.Min.Y, bounds.Min.Z, bounds.Max.X, bounds.Max.Y, bounds.Max.Z};
        out.bounds.insert(out.bounds.end(), b, b + 6);
        out.max_depth = std::max(out.max_depth, depth);
        const size_t n = list.size();
        if (n <= 1 || depth == 100) {  // stop condition, Scene.cpp:25
            Leaf(me, list);
            return me;
        }
        const unsigned axis = depth % 3;  // round-robin
        real median = 0;                  // mean of all vertex coordinates, summed in list order
        for (uint32_t ti : list) {
            const Triangle &t = tris[ti];
            median += t.Vertices[0][axis];
            median += t.Vertices[1][axis];
            median += t.Vertices[2][axis];
        }
        median /= 3 * n;
        BoundingBox lb, rb;
        bounds.Split(static_cast<Axis>(axis), median, lb, rb);
        std::vector<uint32_t> left, right;
        for (uint32_t ti : list) {
            const Triangle &t = tris[ti];
            const real a = t.Vertices[0][axis], bb = t.Vertices[1][axis], c = t.Vertices[2][axis];
            if (a <= median || bb <= median || c <= median) left.push_back(ti);
            if (a >= median || bb >= median || c >= median) right.push_back(ti);
        }
        if (left.size() + right.size() > 1.5 * n) {  // > 50 % duplicated: stop subdividing
            Leaf(me, list);
            return me;
        }
        out.nodes[me].split = median;
        uint32_t info = axis;
        if (!left.empty()) {
            info |= RTHIP_KD_HAS_LEFT;
            Build(left, lb, depth + 1);  // lands at me + 1
        }
        std::vector<uint32_t>().swap(left);
        uint32_t link = 0;
        if (!right.empty()) {
            info |= RTHIP_KD_HAS_RIGHT;
            link = Build(right, rb, depth + 1);
        }
        out.nodes[me].link = link;
        out.nodes[me].info = info;
        return me;
    }
};
}  // namespace

void Mesh::Compile() {
    if (!m_Invalidated) return;
    // the bounding box starts at (0,0,0)/(0,0,0) and is never reset (Scene.cpp:391-406, Common.h:212)
    Vector3 &mn = m_BoundingBox.Min, &mx = m_BoundingBox.Max;
    for (Triangle &t : m_Triangles) {
        t.Update();
        for (const Vector3 &v : t.Vertices) {
            mn.X = std::min(mn.X, v.X); mn.Y = std::min(mn.Y, v.Y); mn.Z = std::min(mn.Z, v.Z);
            mx.X = std::max(mx.X, v.X); mx.Y = std::max(mx.Y, v.Y); mx.Z = std::max(mx.Z, v.Z);
        }
    }
    m_KdTree = KdTreeFlat();
    std::vector<uint32_t> all(m_Triangles.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
    KdBuilder builder{m_Triangles, m_KdTree};
    builder.Build(all, m_BoundingBox, 0);
    m_Invalidated = false;
}

}  // namespace re
