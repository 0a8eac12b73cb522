// re/re.hpp — umbrella include of the re:: host API (reference Raytracer/re.h:1-11)
#pragma once
#include "math.hpp"
#include "renderer.hpp"
#include "scene.hpp"
