// re/scene.hpp — scene-construction API of the re:: host mirror: the same public names, fields and
// signatures a caller of the reference uses to BUILD a scene (Raytracer/Scene.h, Material.h,
// noise/*.h; see Sandbox/Sandbox.cpp:595-750 for the call pattern), so such callers compile
// unchanged.  Differences, all additive or deliberate:
//   * per-ray evaluation (Shape::Intersect, Noise::Sample, Material::Get*, Background::GetColor,
//     Scene::CastRay) is NOT provided on the host: it runs on the device only (no CPU fallback).
//   * read-only accessors / table setters exist so the flattener (flatten.hpp) can reach what the
//     reference keeps private (SURVEY.md §8b "private-data problem").
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <vector>

#include "../mesh_build.hpp"
#include "math.hpp"

namespace re {

class SceneNode;

// ------------------------------------------------------------------ noise (noise/*.h)
class Noise {
public:
    explicit Noise(real domainSize) : m_DomainSize(domainSize) {}
    virtual ~Noise() = default;
    real GetDomainSize() const { return m_DomainSize; }
private:
    real m_DomainSize;
};

class CheckerBoard : public Noise {  // noise/CheckerBoard.h:7-12
public:
    explicit CheckerBoard(real domainSize) : Noise(domainSize) {}
};

class Perlin : public Noise {  // noise/Perlin.h:7-30
public:
    static constexpr size_t SeedSize = 256;
    static constexpr size_t MaxOctaves = 8;
    explicit Perlin(real domainSize);  // draws the table with rand() % 256 like Perlin.cpp:17-23
    void NewSeed();
    void SetPersistance(real p) { m_Persistance = p; }
    void SetOctaves(unsigned int o) { m_Octaves = o; }
    // additive: table access for the flattener / reproducible tests
    const std::array<uint8_t, SeedSize> &GetSeedTable() const { return m_Seed; }
    void SetSeedTable(const uint8_t *table256);
    real GetPersistance() const { return m_Persistance; }
    unsigned int GetOctaves() const { return m_Octaves; }
private:
    real m_Persistance = 0.5f;
    unsigned int m_Octaves = MaxOctaves;
    std::array<uint8_t, SeedSize> m_Seed;
};

class Worley : public Noise {  // noise/Worley.h:8-22
public:
    Worley(real domainSize, int divisions);  // one rand() feature point per cell, Worley.cpp:3-27
    int GetDivisions() const { return m_Divisions; }
    real GetStep() const { return m_Step; }
    const std::vector<Vector3> &GetPoints() const { return m_Points; }  // additive
    void SetPoints(const double *xyz);                                  // additive: div^3 * 3 doubles
private:
    std::vector<Vector3> m_Points;
    int m_Divisions;
    real m_Step;
};

class Marble : public Noise {  // noise/Marble.h:9-17
public:
    Marble(real domainSize, real frequency = 50, real turbolence = 4);
    real GetFrequency() const { return m_Frequency; }
    real GetTurbolence() const { return m_Turbolence; }
    const std::shared_ptr<Perlin> &GetPerlin() const { return m_Perlin; }  // additive
private:
    std::shared_ptr<Perlin> m_Perlin;
    real m_Frequency, m_Turbolence;
};

// ------------------------------------------------------------------ materials (Material.h)
class Material {
public:
    virtual ~Material() = default;
};

class UniformMaterial : public Material {  // Material.h:18-32
public:
    static const UniformMaterial OpaqueWhite, OpaqueBlack;
    UniformMaterial(Color absorbedColor, real absorptance) : m_AbsorbedColor(absorbedColor), m_Absorptance(absorptance) {}
    const Color &GetColor() const { return m_AbsorbedColor; }   // additive
    real GetAbsorptanceValue() const { return m_Absorptance; }  // additive
private:
    Color m_AbsorbedColor;
    real m_Absorptance;
};

class InterpolatedMaterial : public Material {  // Material.h:35-47
public:
    InterpolatedMaterial(std::shared_ptr<Noise> noise, std::shared_ptr<Material> material0, std::shared_ptr<Material> material1)
        : m_Material0(std::move(material0)), m_Material1(std::move(material1)), m_Noise(std::move(noise)) {}
    const std::shared_ptr<Noise> &GetNoise() const { return m_Noise; }          // additive
    const std::shared_ptr<Material> &GetMaterial0() const { return m_Material0; }
    const std::shared_ptr<Material> &GetMaterial1() const { return m_Material1; }
private:
    std::shared_ptr<Material> m_Material0, m_Material1;
    std::shared_ptr<Noise> m_Noise;
};

// ------------------------------------------------------------------ lights / backgrounds (Scene.h:17-65)
enum class LightType { Directional, Ambient, Point };
enum class NormalModes { Face, Vertex };

class Light {
public:
    LightType Type = LightType::Directional;
    Vector3 Direction = Vector3::Up;
    Vector3 Position = Vector3::Zero;
    re::Color Color = re::Color::White;
    real Attenuation = 5.0f;
    bool Enabled = true;
};

class Background {
public:
    virtual ~Background() = default;
};
class ColorBackground : public Background {
public:
    ColorBackground(Color color) : m_Color(color) {}
    const Color &GetBackgroundColor() const { return m_Color; }  // additive
private:
    Color m_Color;
};
class SkyBox : public Background {
public:
    SkyBox(Color color0, Color color1, const std::shared_ptr<Light> &sun);  // owns a Perlin(1), Scene.cpp:105-109
    float SunFactor = 128.0f;  // unused by the reference as well (Scene.h:58)
    const std::shared_ptr<Perlin> &GetPerlin() const { return m_Perlin; }  // additive
    const Color &GetSkyColor0() const { return m_SkyColor0; }
    const Color &GetSkyColor1() const { return m_SkyColor1; }
    const std::shared_ptr<Light> &GetSun() const { return m_Sun; }
private:
    std::shared_ptr<Perlin> m_Perlin;
    Color m_SkyColor0, m_SkyColor1;
    std::shared_ptr<Light> m_Sun;
};

// ------------------------------------------------------------------ components (Scene.h:105-215)
class Component {
public:
    explicit Component(SceneNode *owner) : m_Owner(owner) {}
    virtual ~Component() = default;
    virtual void Compile() = 0;
protected:
    SceneNode *m_Owner;
};

class Transform : public Component {
public:
    explicit Transform(SceneNode *owner) : Component(owner) {}
    Vector3 Position, Rotation, Scale = Vector3::One;
    void Compile() override;  // Scene.cpp:239-264
    void GetTransform(Matrix4 &r) const { r = m_Transform; }
    void GetInverseTransform(Matrix4 &r) const { r = m_InverseTransform; }
    void GetNormalTransform(Matrix4 &r) const { r = m_NormalTransform; }
private:
    Matrix4 m_Transform = Matrix4::Identity, m_InverseTransform = Matrix4::Identity, m_NormalTransform = Matrix4::Identity;
};

class Shape : public Component {
public:
    explicit Shape(SceneNode *owner);  // global id counter, Scene.cpp:365-370
    re::Material *Material = (re::Material *)&(UniformMaterial::OpaqueWhite);
    unsigned int GetID() const { return m_ID; }
    void Compile() override {}
protected:
    unsigned int m_ID;
};
class Sphere : public Shape {  // unit sphere at the local origin
public:
    explicit Sphere(SceneNode *owner) : Shape(owner) {}
};
class Plane : public Shape {  // one-sided plane through the local origin
public:
    explicit Plane(SceneNode *owner) : Shape(owner) {}
    Vector3 Normal = Vector3::Up;
};

struct Triangle {  // Scene.h:167-186
    std::array<Vector3, 3> Vertices, Normals;
    Vector3 FaceNormal;
    std::array<Vector3, 3> Edges;
    void Update();  // Scene.cpp:509-523
    real D00() const { return m_D00; }
    real D01() const { return m_D01; }
    real D11() const { return m_D11; }
    real InvDen() const { return m_InvDen; }
private:
    real m_D00 = 0, m_D01 = 0, m_D11 = 0, m_InvDen = 0;
};

// flat kd-tree / BVH of one mesh: see mesh_build.hpp (shared with the binding against the genuine reference headers)
using KdTreeFlat = rtbuild::KdTreeFlat;
using BvhFlat = rtbuild::BvhFlat;

class Mesh : public Shape {
public:
    explicit Mesh(SceneNode *owner) : Shape(owner) {}
    NormalModes NormalMode = NormalModes::Face;
    Triangle &AddTriangle();
    void ReserveTriangles(size_t n) { m_Triangles.reserve(n); }  // additive: a caller that knows the count avoids the vector's regrowth copies
    void Compile() override;  // Scene.cpp:386-414
    void Invalidate() { m_Invalidated = true; }
    const std::vector<Triangle> &GetTriangles() const { return m_Triangles; }  // additive
    const KdTreeFlat &GetKdTree() const { return m_KdTree; }                    // additive
    const BvhFlat &GetBvh() const { return m_Bvh; }                             // additive (empty when BvhBuiltOnDevice())
    // additive: meshes of at least DeviceBvhMinTriangles() triangles (2^18; env RE_BVH_DEVICE_MIN_TRIS, 0 = never) do not
    // build the 4-wide BVH in Compile(): the device builds it at upload.  ForceHostBvh / BuildHostBvh() bring it back.
    bool ForceHostBvh = false;
    bool BvhBuiltOnDevice() const;
    void BuildHostBvh();
    static size_t DeviceBvhMinTriangles();
    // additive, likewise for the kd-tree: meshes of at least DeviceKdMinTriangles() triangles (2^18; env RE_KD_DEVICE_MIN_TRIS,
    // 0 = never) leave KDTreeTriangle's build (Scene.cpp:18-95) to the device (RTHIP_UPLOAD_BUILD_KD, csrc/rthip_kd.hip: the same
    // splits, byte-identical arrays); GetKdTree() is then empty.  ForceHostKd / BuildHostKd() bring the host build back.
    bool ForceHostKd = false;
    bool KdBuiltOnDevice() const;
    void BuildHostKd();
    static size_t DeviceKdMinTriangles();
    const BoundingBox &GetBounds() const { return m_BoundingBox; }              // additive
    // additive: changes every time Compile() rebuilds the acceleration structures (process-wide unique), so a renderer
    // can tell that the geometry it has already flattened and uploaded is still the current one
    uint64_t GetRevision() const { return m_Revision; }
private:
    void BuildKdTree();
    bool m_Invalidated = true;
    uint64_t m_Revision = 0;
    KdTreeFlat m_KdTree;
    BvhFlat m_Bvh;
    std::vector<Triangle> m_Triangles;
    BoundingBox m_BoundingBox;
};

// ------------------------------------------------------------------ graph (Scene.h:67-103, 217-266)
class SceneNode {
public:
    SceneNode();  // every node owns a Transform (Scene.cpp:266-269)
    virtual ~SceneNode() = default;
    virtual bool IsLeaf() const { return m_Children.empty(); }
    virtual std::vector<std::shared_ptr<SceneNode>> &GetChildren() { return m_Children; }
    std::shared_ptr<SceneNode> AddChild();
    SceneNode *GetParent() const { return m_Parent; }

    template <typename T> std::shared_ptr<T> AddComponent() {
        auto c = std::shared_ptr<Component>(new T(this));
        m_Components.push_back(c);
        return std::dynamic_pointer_cast<T, Component>(c);
    }
    template <typename T> std::shared_ptr<T> GetComponentOfType() {
        for (auto &c : m_Components)
            if (auto cast = std::dynamic_pointer_cast<T, Component>(c)) return cast;
        return nullptr;
    }
    template <typename T> std::vector<std::shared_ptr<T>> GetComponentsOfType() {
        std::vector<std::shared_ptr<T>> out;
        for (auto &c : m_Components)
            if (auto cast = std::dynamic_pointer_cast<T, Component>(c)) out.push_back(cast);
        return out;
    }
    void Compile();  // components first, then children (Scene.cpp:279-292)
protected:
    SceneNode *m_Parent = nullptr;
    std::vector<std::shared_ptr<SceneNode>> m_Children;
    std::vector<std::shared_ptr<Component>> m_Components;
};

class Scene {
public:
    struct {
        Vector3 Position = Vector3::Zero;
        Vector3 Direction = {0, 0, -1};
        void LookAt(const Vector3 &target) { Direction = (target - Position).Normalized(); }
    } Camera;
    std::vector<std::shared_ptr<Light>> Lights;
    std::shared_ptr<re::Background> Background = nullptr;

    Scene() : m_Root(std::make_shared<SceneNode>()) {}
    void Compile() { m_Root->Compile(); }
    std::shared_ptr<SceneNode> GetRoot() { return m_Root; }
private:
    std::shared_ptr<SceneNode> m_Root;
};

}  // namespace re
