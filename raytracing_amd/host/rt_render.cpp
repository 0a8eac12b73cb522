// rt_render — headless command line for the MI355X Raycast path: renders a scene of a scenes.xml store
// (the reference's Sandbox/res/scenes.xml format) or a scene script file with re::HipRaytracer and writes a PPM.
// The reference has no command line (its only front end is the GLFW/ImGui Sandbox); this is the
// smallest possible stand-in for Sandbox::StartRaytracer (Sandbox/Sandbox.cpp:556-562) on a GPU box.
//
//   rt_render --xml scenes.xml --scene "Demo 4 - Bunny" [--res DIR] [--size 1920x1080] [--maxrec 3]
//             [--ssaa] [--debug normal|color] [--device 0] [--frames 1] --out frame.ppm
//   rt_render --script scene.lua ...            rt_render --xml scenes.xml --list
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "re/re.hpp"
#include "scene_io.hpp"

int main(int argc, char **argv) {
    std::string xml, script, scene_name, res_dir, out;
    unsigned w = 1280, h = 768, maxrec = 3;  // Main.cpp:4-8 opens a 1280x768 window
    int device = 0, frames = 1, debug = -1;
    bool ssaa = false, list = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--xml") xml = val();
        else if (a == "--script") script = val();
        else if (a == "--scene") scene_name = val();
        else if (a == "--res") res_dir = val();
        else if (a == "--out") out = val();
        else if (a == "--size") { if (std::sscanf(val(), "%ux%u", &w, &h) != 2) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--maxrec") maxrec = (unsigned)std::atoi(val());
        else if (a == "--device") device = std::atoi(val());
        else if (a == "--frames") frames = std::atoi(val());
        else if (a == "--ssaa") ssaa = true;
        else if (a == "--list") list = true;
        else if (a == "--debug") { const std::string m = val(); debug = m == "color" ? 1 : 0; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        std::string code;
        if (!xml.empty()) {
            auto scenes = sbio::ReadScenesXml(xml);
            if (list) {
                for (auto &s : scenes) std::printf("%s\n", s.first.c_str());
                return 0;
            }
            bool found = false;
            for (auto &s : scenes)
                if (s.first == scene_name || (scene_name.empty() && !found)) { code = s.second; scene_name = s.first; found = true; if (!scene_name.empty()) break; }
            if (!found) { std::fprintf(stderr, "no scene '%s' in %s\n", scene_name.c_str(), xml.c_str()); return 1; }
            if (res_dir.empty()) {  // the Sandbox resolves "res/..." relative to its working directory, the parent of res/
                const size_t slash = xml.find_last_of('/');
                const std::string dir = slash == std::string::npos ? "." : xml.substr(0, slash);
                const size_t up = dir.find_last_of('/');
                res_dir = up == std::string::npos ? "." : dir.substr(0, up);
            }
        } else if (!script.empty()) {
            std::ifstream is(script);
            if (!is) { std::fprintf(stderr, "cannot open %s\n", script.c_str()); return 1; }
            std::stringstream ss;
            ss << is.rdbuf();
            code = ss.str();
        } else {
            std::fprintf(stderr, "usage: rt_render (--xml scenes.xml [--scene NAME | --list] | --script FILE) [--res DIR] [--size WxH] "
                                 "[--maxrec N] [--ssaa] [--debug normal|color] [--device N] [--frames N] --out frame.ppm\n");
            return 2;
        }
        sbio::ScriptScene sc;
        sbio::RunSceneScript(code, res_dir, sc);
        std::unique_ptr<re::HipRaytracer> rt;
        if (debug >= 0) {
            auto d = new re::HipDebugRaycaster(w, h);
            d->Mode = debug ? re::HipDebugRaycaster::Modes::Color : re::HipDebugRaycaster::Modes::Normal;
            rt.reset(d);
        } else {
            rt.reset(new re::HipRaytracer(w, h));
        }
        rt->MaxRecursion = maxrec;
        rt->Device = device;
        rt->Antialiasing = ssaa ? re::AbstractRaycaster::AAMode::SSAA : re::AbstractRaycaster::AAMode::None;
        unsigned int *pixels = nullptr;
        for (int f = 0; f < frames; f++) {
            const auto t0 = std::chrono::steady_clock::now();
            pixels = rt->RenderSync(sc.Scene.get());
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            const re::HipRenderStats st = rt->GetLastStats();
            const unsigned long long rays = st.RaysPrimary + st.RaysShadow + st.RaysReflect;
            std::printf("frame %d: %ux%u '%s' %llu rays, device %.3f ms (%.0f Mrays/s), flatten %.1f ms, upload %.1f ms, wall %.1f ms\n", f, w, h,
                        scene_name.c_str(), rays, st.DeviceMs, st.DeviceMs > 0 ? rays / st.DeviceMs / 1e3 : 0.0, st.FlattenMs, st.UploadMs, ms);
        }
        if (!out.empty()) sbio::WritePPM(out, pixels, w, h);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rt_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
