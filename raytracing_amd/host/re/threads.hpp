// Host threads for the scene build (Mesh::Compile, the flattener): std::async tasks, no pool (mesh_build.hpp).
#pragma once
#include "../mesh_build.hpp"

namespace re {
using rtbuild::BuildForkDepth;
using rtbuild::ParallelSlices;
}  // namespace re
