// scene_io.hpp — scene inputs of the reference's Sandbox app, restated without its GUI / Lua / TinyXML2
// dependencies: the Wavefront OBJ reader (Sandbox/WavefrontLoader.{h,cpp}), the scene
// script calls of Sandbox::UpdateScene (Sandbox/Sandbox.cpp:578-763) and the <scene> store of
// Sandbox/res/scenes.xml, plus a PPM writer for headless output.  Host-side input parsing only.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "re/re.hpp"

namespace sbio {

// Sandbox/WavefrontLoader.h:11-22
struct WfFace {
    std::vector<re::Vector3> Vertices;
    std::vector<re::Vector3> Normals;
    std::vector<re::Vector2> TexCoords;
};
using WfGroup = std::vector<WfFace>;
using WfData = std::map<std::string, WfGroup>;

// Sandbox/WavefrontLoader.cpp:11-102: tokens v / vt / vn / g / f; face vertices `v`, `v/vt`, `v//vn`,
// `v/vt/vn` (1-based); no triangulation; groups default to "default".  Where the reference has undefined
// behaviour (an index outside the lists read so far, e.g. from trailing blanks on an `f` line) this
// reader throws std::runtime_error naming the line.
WfData LoadWavefront(const std::string &fileName);

// The objects a scene script creates.  The scene holds raw Material pointers (Shape::Material), so the
// vectors must outlive it — same ownership as Sandbox::m_Materials / m_Noises / m_Lights.
struct ScriptScene {
    std::shared_ptr<re::Scene> Scene;
    std::vector<std::shared_ptr<re::Material>> Materials;
    std::vector<std::shared_ptr<re::Noise>> Noises;
    std::vector<std::shared_ptr<re::Light>> Lights;
    std::vector<re::SceneNode *> ShapeNodes;  // in creation order
};

// Evaluates the call subset of the Sandbox's Lua scene scripts: `name = call(...)`, bare calls, nested calls,
// decimal / hex numbers, strings, `--` comments, optional `;` and `local`.  Calls and their effects are those
// registered by Sandbox::UpdateScene (reUniformMaterial, reInterpolatedMaterial, reCheckerBoard, rePerlin,
// reWorley, reMarble, rePosition, reScale, reSphere, rePlane, reObjMesh, reAmbientLight, reDirectionalLight,
// rePointLight, reCameraPos, reSkyBox), executed in script order (Perlin / Worley / SkyBox constructors draw
// from rand() exactly as the reference's do).  resDir is prepended to relative reObjMesh file names.
// Throws std::runtime_error("line N: ...").
void RunSceneScript(const std::string &code, const std::string &resDir, ScriptScene &out);

// <scene name="...">code</scene> entries of a scenes.xml file (Sandbox/res/scenes.xml), in file order.
std::vector<std::pair<std::string, std::string>> ReadScenesXml(const std::string &path);

// binary PPM (P6) from 0xAABBGGRR pixels (R in the low byte, Common.cpp:275-282)
void WritePPM(const std::string &path, const unsigned int *pixels, size_t width, size_t height);

}  // namespace sbio
