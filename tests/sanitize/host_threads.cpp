// host_threads.cpp — TEST ONLY: the threaded host code under ThreadSanitizer / AddressSanitizer + UBSan.
// Built by tests/sanitize/Makefile from the product's host SOURCES (not the .so) with -fsanitize=..., run by
// tests/test_sanitizers.py in the GPU-less container.  Exercises:
//   * Mesh::Compile on a mesh big enough for the builders to fork (std::async sub-tree tasks of KdBuilder / BvhBuilder,
//     ParallelSlices in Prepare / Mesh::Compile / the flattener) and checks the arrays equal the single-threaded build;
//   * FlattenScene twice (geometry kept the second time);
//   * re::HipRaytracer's render-thread lifetime: Render() + GetStatus() polling + Interrupt() from other threads, the
//     promise kept exactly once, destruction right after the future is ready, destruction while busy.  Without a GPU
//     rthip_create fails and the failure travels through the promise - the thread / condvar / promise code is the same.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "flatten.hpp"
#include "re/renderer.hpp"

using namespace re;

static int g_fail = 0;
#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); g_fail++; } } while (0)

static std::shared_ptr<Scene> MakeScene(int cells, std::shared_ptr<Material> &keep) {
    auto scene = std::make_shared<Scene>();
    keep = std::make_shared<UniformMaterial>(Color(0x88aa66u), 0.8);
    auto node = scene->GetRoot()->AddChild();
    auto mesh = node->AddComponent<Mesh>();
    mesh->Material = keep.get();
    mesh->NormalMode = NormalModes::Vertex;
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0; };
    std::vector<double> h((size_t)(cells + 1) * (cells + 1));
    for (auto &v : h) v = 0.1 * (2 * rnd() - 1);
    auto P = [&](int i, int j) { return Vector3(-8.0 + 16.0 * i / cells, 0.35 * std::sin(1.7 * i / cells) + h[(size_t)j * (cells + 1) + i], -8.0 + 16.0 * j / cells); };
    for (int j = 0; j < cells; j++)
        for (int i = 0; i < cells; i++) {
            Triangle &a = mesh->AddTriangle();
            a.Vertices = {P(i, j), P(i, j + 1), P(i + 1, j + 1)};
            a.Normals = {Vector3::Up, Vector3::Up, Vector3::Up};
            Triangle &b = mesh->AddTriangle();
            b.Vertices = {P(i, j), P(i + 1, j + 1), P(i + 1, j)};
            b.Normals = {Vector3::Up, Vector3::Up, Vector3::Up};
        }
    auto sph = scene->GetRoot()->AddChild();
    sph->GetComponentOfType<Transform>()->Position = Vector3(0, 2, 0);
    sph->AddComponent<Sphere>()->Material = keep.get();
    auto sun = std::make_shared<Light>();
    scene->Lights.push_back(sun);
    scene->Background = std::make_shared<SkyBox>(Color(0x4444ffu), Color(0xffffffu), sun);
    return scene;
}

template <class T> static bool Same(const std::vector<T> &a, const std::vector<T> &b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), sizeof(T) * a.size()) == 0);
}

int main() {
    const int cells = 200;  // 80 000 triangles: above the builders' 32 768-triangle fork threshold
    std::shared_ptr<Material> m1, m2;
    // threaded build == single-threaded build
    setenv("RE_BUILD_THREADS", "8", 1);
    auto threaded = MakeScene(cells, m1);
    threaded->Compile();
    FlatScene ft;
    std::string err;
    CHECK(FlattenScene(threaded.get(), ft, err) == 0);
    setenv("RE_BUILD_THREADS", "1", 1);
    auto serial = MakeScene(cells, m2);
    serial->Compile();
    FlatScene fs;
    CHECK(FlattenScene(serial.get(), fs, err) == 0);
    CHECK(ft.tris.size() == (size_t)2 * cells * cells);
    CHECK(Same(ft.kd_nodes, fs.kd_nodes) && Same(ft.kd_bounds, fs.kd_bounds) && Same(ft.kd_refs, fs.kd_refs) && Same(ft.kd_planes, fs.kd_planes));
    CHECK(Same(ft.bvh_nodes, fs.bvh_nodes) && Same(ft.bvh_tri_index, fs.bvh_tri_index) && Same(ft.bvh_tris, fs.bvh_tris) && Same(ft.tris, fs.tris));
    setenv("RE_BUILD_THREADS", "8", 1);
    // flatten against the previous flatten: geometry kept
    FlatScene again;
    ft.ReleaseGeometry();
    CHECK(FlattenScene(threaded.get(), again, err, &ft) == 0 && again.geometry_kept && again.tris.empty());

    // render-thread lifetime (no GPU here: the device error arrives through the promise)
    for (int round = 0; round < 4; round++) {
        auto rt = std::make_unique<HipRaytracer>(64, 48);
        std::promise<Renderer::RenderStatus> p;
        auto f = p.get_future();
        rt->Render(threaded.get(), std::move(p));  // returns at once
        std::thread poller([&] { for (int i = 0; i < 50; i++) { auto st = rt->GetStatus(); (void)st; std::this_thread::yield(); } });
        std::thread breaker([&] { if (round & 1) rt->Interrupt(); });
        if (round == 2) {  // destroy while (possibly) busy: the destructor waits for the render thread
            poller.join();
            breaker.join();
            rt.reset();
            try { (void)f.get(); } catch (const DeviceError &) {}
            continue;
        }
        bool delivered = false;
        try {
            auto st = f.get();
            delivered = st.Finished;
        } catch (const DeviceError &e) {
            delivered = true;  // the promise was kept with the device's refusal (this container has no GPU)
        }
        CHECK(delivered);
        poller.join();
        breaker.join();
        CHECK(rt->GetStatus().Finished);
        rt.reset();  // right after the future is ready
    }
    // two renders in a row through one renderer (the second waits for the first)
    {
        HipRaytracer rt(32, 32);
        std::promise<Renderer::RenderStatus> p1, p2;
        auto f1 = p1.get_future(), f2 = p2.get_future();
        rt.Render(threaded.get(), std::move(p1));
        rt.Render(threaded.get(), std::move(p2));
        for (auto *f : {&f1, &f2}) {
            try { (void)f->get(); } catch (const DeviceError &) {}
        }
    }
    std::printf(g_fail ? "FAILED %d checks\n" : "host_threads ok\n", g_fail);
    return g_fail ? 1 : 0;
}
