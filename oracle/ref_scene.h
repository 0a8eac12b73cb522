/* ref_scene.h — TEST INFRASTRUCTURE (oracle/_ref, container-only): the scene handle behind the sb_* facade of the genuine
 * reference build, shared by ref_harness.cpp (oracle) and ref_binding.cpp (the drop-in binding proven on the genuine headers). */
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "Scene.h"  /* from -I<reference>/Raytracer */

struct sb_scene {
    std::shared_ptr<re::Scene> scene;
    std::vector<re::SceneNode *> nodes;  // id -> node (0 = root)
    std::vector<std::shared_ptr<re::Material>> mats;
    std::vector<std::shared_ptr<re::Noise>> noises;
    std::vector<std::shared_ptr<re::Light>> lights;
    std::vector<int> noise_type;  // 0 checker 1 perlin 2 worley 3 marble
    std::map<re::SceneNode *, int> ordinal;
    bool compiled = false;
};
