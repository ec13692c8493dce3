// rtr_render -- headless renderer over host/rtr.hpp (the reference has no headless mode: src/main.rs is a GUI).
//
//   rtr_render --scene data/scenes/benchmark.json --width 1920 --height 1080 --spp 256 --seed 1 --out frame.ppm
//   rtr_render --preset gui --mesh-dir data/meshes --out house.ppm      (the scene src/gui.rs:56-119 builds)
//   rtr_render --scene ... --host-only --info                            (no GPU: lowering + BVH statistics only)
//
// Options: --bvh 0|1, --depth N, --mode Full|Clay|Random|Normal, --f32 file (raw mean radiance floats), --device N,
//          --devices 0,1,2,3 (one context over several GPUs: rtr_create_multi divides every frame among them),
//          --fast-trees (rebuild mesh and scene trees with the surface-area heuristic: same hits, faster walk),
//          --device-trees (build the mesh trees on the GPU as linear BVHs: same hits, fast build, slower walk),
//          --sky file.pfm (replace the scene's background by a SkyMap read from a colour Portable Float Map).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "rtr.hpp"

static rtr::Scene gui_scene(const std::string &mesh_dir) { // src/gui.rs:56-119 (SkyMap -> gradient: lake.exr is absent)
    using namespace rtr;
    Scene scene;
    scene.camera.lookfrom = Vec3{-3.0f, 2.0f, 4.0f};
    scene.camera.lookat = Vec3{0.0f, 2.0f, -1.0f};
    scene.camera.aspect_ratio = 16.0f / 9.0f;
    scene.settings.max_ray_depth = 3;
    scene.background = std::make_unique<GradientBackground>(Color{1.0f, 1.0f, 1.0f}, Color{0.5f, 0.7f, 1.0f});
    MaterialHandle lambertian = scene.add_material(std::make_unique<Lambertian>(Color{0.9f, 0.2f, 0.2f}));
    scene.add_material(std::make_unique<Metal>(Color{0.8f, 0.2f, 0.2f}, 0.02f));
    scene.add_material(std::make_unique<Dielectric>(1.5f));
    MaterialHandle ground = scene.add_material(std::make_unique<MixMaterial>(
        std::make_unique<Metal>(Color{0.1f, 0.1f, 0.1f}, 0.0f), std::make_unique<Lambertian>(Color{0.4f, 0.4f, 0.4f}), 0.5f));
    scene.add_material(std::make_unique<Emission>(Color{1.0f, 1.0f, 1.0f}, 50.0f));
    scene.add_material(std::make_unique<Isotropic>(Color{1.0f, 1.0f, 1.0f}));
    auto house = Mesh::from_file(mesh_dir + "/house.obj", lambertian);
    house->build_bvh();
    scene.add_object(std::move(house));
    auto plane = Mesh::from_file(mesh_dir + "/plane.obj", ground);
    plane->build_bvh();
    scene.add_object(std::move(plane));
    return scene;
}

int main(int argc, char **argv) {
    std::string scene_path, preset, mesh_dir = "data/meshes", out, f32_out, mode, sky;
    size_t width = 256, height = 256;
    long spp = -1, depth = -1, bvh = -1;
    uint64_t seed = 0x5EED;
    int device = 0;
    std::vector<int> devices;
    bool info = false, host_only = false, fast_trees = false, device_trees = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--scene") scene_path = next();
        else if (a == "--preset") preset = next();
        else if (a == "--mesh-dir") mesh_dir = next();
        else if (a == "--out") out = next();
        else if (a == "--f32") f32_out = next();
        else if (a == "--width") width = std::strtoul(next(), nullptr, 0);
        else if (a == "--height") height = std::strtoul(next(), nullptr, 0);
        else if (a == "--spp") spp = std::strtol(next(), nullptr, 0);
        else if (a == "--depth") depth = std::strtol(next(), nullptr, 0);
        else if (a == "--bvh") bvh = std::strtol(next(), nullptr, 0);
        else if (a == "--mode") mode = next();
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
        else if (a == "--device") device = (int)std::strtol(next(), nullptr, 0);
        else if (a == "--devices") { // comma-separated GPU indices
            const std::string list = next();
            for (size_t i = 0; i < list.size();) {
                size_t j = list.find(',', i);
                if (j == std::string::npos) j = list.size();
                devices.push_back((int)std::strtol(list.substr(i, j - i).c_str(), nullptr, 0));
                i = j + 1;
            }
        }
        else if (a == "--info") info = true;
        else if (a == "--host-only") host_only = true;
        else if (a == "--fast-trees") fast_trees = true;
        else if (a == "--device-trees") device_trees = true;
        else if (a == "--sky") sky = next();
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        rtr::Scene scene = !scene_path.empty() ? rtr::Scene::from_file(scene_path)
                           : preset == "gui"   ? gui_scene(mesh_dir)
                                               : throw std::runtime_error("give --scene FILE or --preset gui");
        if (!sky.empty()) scene.background = rtr::SkyMap::from_pfm(sky); // SkyMap::new (backgrounds.rs:58-81) from a .pfm
        if (spp >= 0) scene.settings.samples_per_pixel = (uint32_t)spp;
        if (depth >= 0) scene.settings.max_ray_depth = (uint8_t)depth;
        if (bvh >= 0) scene.settings.enable_bvh_tree = bvh != 0;
        if (!mode.empty())
            scene.settings.mode = mode == "Clay" ? rtr::RenderMode::Clay : mode == "Random" ? rtr::RenderMode::Random
                                : mode == "Normal" ? rtr::RenderMode::Normal : rtr::RenderMode::Full;
        if (host_only) devices.assign(devices.size(), -1);
        rtr::Renderer r = devices.empty() ? rtr::Renderer(host_only ? -1 : device) : rtr::Renderer(devices);
        if (fast_trees) { r.set_option("mesh_tree", "sah"); r.set_option("scene_tree", "sah"); }
        if (device_trees) r.set_option("mesh_tree", "lbvh"); // mesh trees built on the GPU (csrc/rtr_lbvh.hip)
        r.upload(scene);
        if (info) {
            rtr_scene_info si = r.info();
            std::printf("objects %u spheres %u meshes %u triangles %u materials %u leaves %u nodes %u height %u spheres_only %u\n",
                        si.n_objects, si.n_spheres, si.n_meshes, si.n_triangles, si.n_materials, si.n_leaves, si.n_nodes,
                        si.tree_height, si.spheres_only);
        }
        if (host_only) return 0;
        rtr::RenderTarget target(width, height);
        auto t0 = std::chrono::steady_clock::now();
        r.render(scene, target, seed);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        rtr_stats st = r.stats();
        std::printf("rendered %zux%zu spp %u: %.3f ms (kernel %.3f ms), %llu ray segments, %.1f Mrays/s\n", width, height,
                    scene.settings.samples_per_pixel, ms, st.kernel_ms, (unsigned long long)st.segments,
                    st.kernel_ms > 0 ? st.segments / st.kernel_ms / 1e3 : 0.0);
        if (!out.empty()) target.save(out); // .png or .ppm
        if (!f32_out.empty()) {
            std::vector<float> mean;
            r.render_f32(scene, width, height, seed, mean);
            FILE *f = std::fopen(f32_out.c_str(), "wb");
            if (!f) throw std::runtime_error("cannot write " + f32_out);
            std::fwrite(mean.data(), sizeof(float), mean.size(), f);
            std::fclose(f);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rtr_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
