// rodent -- the renderer's command line, same options and output as the reference's driver
// (src/driver/driver.cpp:169-232 options, :279-348 frame loop and "min/med/max Msamples/s" line,
// :138-162 tone-mapped PNG), driving the HIP wavefront path tracer through include/rodent_render.h.
//
// The reference bakes scene, device, spp and max path length into the executable at configure time
// (src/CMakeLists.txt:1-22,101-105); here they are run-time options:
//   --scene file        .obj (converted on the fly) or .rscene written by `converter`   (required)
//   --spp n             samples per pixel per frame        (default: the scene file's, 4)
//   --max-path-len n    maximum path length                (default: the scene file's, 64)
//   -dev n              HIP device                          (default 0)
//   --ngpu K            K GPUs of this node (devices -dev .. -dev + K - 1; BASELINE config 5): every GPU holds the scene and renders
//                       the row band split_range(height, r, K) of every frame (rodent_hip_render_rows; one host thread per
//                       device), a frame takes as long as the slowest band; after the last frame ONE RCCL gather (grouped
//                       ncclSend / ncclRecv, host/multi_gpu.h) brings the bands to the first device's film
//   --target t          amdgpu-streaming or amdgpu-megakernel (default: chosen per scene) (converter.cpp:30-35,1032-1037)
//   --sort / --no-sort  streaming target: sort hit rays by material before shading (the reference's loop) / shade in stream order (default)
// Without --bench the reference opens an SDL window and renders until it is closed; this build is
// headless (DISABLE_GUI, driver.cpp:236-242), so --bench or -o is required.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <mutex>
#include <string>
#include <vector>

#include "../multi_gpu.h"
#include "../png_write.h"
#include "../scene.h"

using namespace rodent;

static void usage() {
    std::cout << "Usage: rodent [options]\n"
              << "Available options:\n"
              << "   --help              Shows this message\n"
              << "   --scene  file       Scene to render (.obj or .rscene)\n"
              << "   --spp    n          Samples per pixel per frame\n"
              << "   --max-path-len n    Maximum path length\n"
              << "   -dev     n          GPU device index\n"
              << "   --ngpu   K          Renders on K GPUs (interleaved 16-row tiles, one film gather to the first device)\n"
              << "   --bands             With --ngpu: one contiguous row band per GPU instead of interleaved tiles\n"
              << "   --check             With --ngpu: renders the same frames again on the first GPU alone and compares the two films\n"
              << "   --target t          amdgpu-streaming or amdgpu-megakernel (default: chosen per scene)\n"
              << "   --sort              Sort rays by material before shading (streaming target; default: stream order)\n"
              << "   --no-sort           Do not sort rays by material before shading\n"
              << "   --width  pixels     Sets the viewport horizontal dimension (in pixels)\n"
              << "   --height pixels     Sets the viewport vertical dimension (in pixels)\n"
              << "   --eye    x y z      Sets the position of the camera\n"
              << "   --dir    x y z      Sets the direction vector of the camera\n"
              << "   --up     x y z      Sets the up vector of the camera\n"
              << "   --fov    degrees    Sets the horizontal field of view (in degrees)\n"
              << "   --bench  iterations Enables benchmarking mode and sets the number of iterations\n"
              << "   -o       image.png  Writes the output image to a file" << std::endl;
}

[[noreturn]] static void fail(const std::string& msg) { std::cerr << msg << std::endl; exit(1); }   // common.h:43-59 error() aborts

int main(int argc, char** argv) {
    std::string out_file, scene_file;
    size_t bench_iter = 0, width = 1080, height = 720;
    float fov = 60.0f;
    V3 eye(0.0f), dir(0.0f, 0.0f, 1.0f), up(0.0f, 1.0f, 0.0f);
    int spp = 0, max_path_len = -1, dev = 0, mapping = -1, ngpu = 1;
    bool bands = false, check = false;
    int sort = -1;                                                        // -1: the library's default

    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] != '-') fail(std::string("Unexpected argument '") + argv[i] + "'");
        auto need = [&](int n) {
            if (i + n >= argc) fail(std::string("Option '") + argv[i] + "' expects " + std::to_string(n) + " arguments, got "
            + std::to_string(argc - i)); };
        if (!strcmp(argv[i], "--width")) { need(1); width = strtoul(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "--height")) { need(1); height = strtoul(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "--eye")) { need(3);
            eye = V3(strtof(argv[i + 1], nullptr), strtof(argv[i + 2], nullptr), strtof(argv[i + 3], nullptr)); i += 3; }
        else if (!strcmp(argv[i], "--dir")) { need(3);
            dir = V3(strtof(argv[i + 1], nullptr), strtof(argv[i + 2], nullptr), strtof(argv[i + 3], nullptr)); i += 3; }
        else if (!strcmp(argv[i], "--up")) { need(3);
            up = V3(strtof(argv[i + 1], nullptr), strtof(argv[i + 2], nullptr), strtof(argv[i + 3], nullptr)); i += 3; }
        else if (!strcmp(argv[i], "--fov")) { need(1); fov = strtof(argv[++i], nullptr); }
        else if (!strcmp(argv[i], "--bench")) { need(1); bench_iter = strtoul(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "-o")) { need(1); out_file = argv[++i]; }
        else if (!strcmp(argv[i], "--scene")) { need(1); scene_file = argv[++i]; }
        else if (!strcmp(argv[i], "--spp")) { need(1); spp = strtol(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "--max-path-len")) { need(1); max_path_len = strtol(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "-dev")) { need(1); dev = strtol(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "--ngpu")) { need(1); ngpu = strtol(argv[++i], nullptr, 10); }
        else if (!strcmp(argv[i], "--bands")) bands = true;
        else if (!strcmp(argv[i], "--check")) check = true;
        else if (!strcmp(argv[i], "--no-sort")) sort = 0;
        else if (!strcmp(argv[i], "--sort")) sort = 1;
        else if (!strcmp(argv[i], "--target")) {
            need(1); ++i;
            if (!strcmp(argv[i], "amdgpu-streaming") || !strcmp(argv[i], "amdgpu")) mapping = 0;
            else if (!strcmp(argv[i], "amdgpu-megakernel")) mapping = 1;
            else fail(std::string("Unknown target '") + argv[i] + "' (this build has amdgpu-streaming and amdgpu-megakernel)");
        }
        else if (!strcmp(argv[i], "--help")) { usage(); return 0; }
        else fail(std::string("Unknown option '") + argv[i] + "'");
    }
    if (scene_file.empty()) fail("No scene specified (--scene file.obj|file.rscene)");
    if (bench_iter == 0 && out_file.empty()) fail("This build has no GUI: use --bench iterations and/or -o image.png");
    if (bench_iter == 0) bench_iter = 1;

    SceneData scene;
    const bool is_obj = scene_file.size() > 4 && scene_file.substr(scene_file.size() - 4) == ".obj";
    if (is_obj ? !build_scene_from_obj(scene_file, scene) : !load_scene(scene_file, scene)) fail("Cannot load scene '" + scene_file + "'");
    if (scene.lights.empty()) fail("The scene has no light source");
    if (spp <= 0) spp = scene.default_spp;
    if (max_path_len < 0) max_path_len = scene.default_max_path_len;

    // Camera (driver.cpp:22-39)
    const V3 d = normalize(dir), r = normalize(cross(d, up)), u = normalize(cross(r, d));
    const float w = std::tan(fov * 3.14159265359f / 360.0f), h = w / ((float)width / (float)height);
    const Settings settings{{eye.x, eye.y, eye.z}, {d.x, d.y, d.z}, {u.x, u.y, u.z}, {r.x, r.y, r.z}, w, h};

    if (ngpu < 1) fail("Invalid GPU count");
    DeviceGroup group;
    {
        std::string err;
        if (!group.init(dev, ngpu, &err)) fail("No such GPU device(s): " + err);
    }
    const RodentSceneDesc desc = scene.desc();
    setup_interface(width, height);
    // (the first device last: it stays the current one of render() / get_spp())
    for (int r = ngpu - 1; r >= 0; r--) {
        const int d = group.device(r);
        rodent_hip_set_device(d);
        rodent_hip_scene_create(d, &desc);
        rodent_hip_render_config(d, spp, max_path_len);
        if (mapping >= 0) rodent_hip_render_mapping(d, mapping);
        if (sort >= 0) rodent_hip_render_sort(d, sort);
    }
    clear_pixels();

    std::vector<double> samples_sec, rank_ms(ngpu, 0.0);
    std::mutex device_lock[16];
    uint32_t iter = 0;
    while (samples_sec.size() < bench_iter) {
        const auto ticks = std::chrono::high_resolution_clock::now();
        if (ngpu == 1) render(&settings, iter++);
        else {
            // every GPU its share of this frame, all at once; the call returns when the share is in the device's film
            group.run([&](int r) {
                // RODENT_SHARE_GPUS maps several ranks onto one device: its RenderDevice (control words, slabs, counters, film) serves ONE
                // render call at a time (K threads on one RenderDevice raced).  With one rank per device -- every real run --
                // nobody ever waits here.
                std::lock_guard<std::mutex> one_call(device_lock[group.device(r) & 15]);
                const auto t0 = std::chrono::high_resolution_clock::now();
                if (bands) { const Part band = split_range((int)height, r, ngpu);
                    rodent_hip_render_rows(group.device(r), &settings, (int32_t)iter, band.begin, band.end, nullptr); }
                else rodent_hip_render_tiles(group.device(r), &settings, (int32_t)iter, kTileRows, r, ngpu, nullptr);
                rank_ms[r] = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
            });
            iter++;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - ticks).count();
        samples_sec.emplace_back(1000.0 * double(get_spp() * width * height) / ms);
    }
    double gather_s = 0.0;
    if (ngpu > 1) {
        // the one collective: every peer's rows straight into the first device's film, then that film to the host
        std::vector<DeviceGroup::Piece> pieces;
        float* root_film = nullptr; int32_t fw = 0, fh = 0;
        rodent_get_film_data(group.device(0), &root_film, &fw, &fh);
        for (int r = 1; r < ngpu; r++) {
            float* film = nullptr;
            rodent_get_film_data(group.device(r), &film, &fw, &fh);
            const auto add = [&](Part rows) {
                pieces.push_back({r, film + (size_t)rows.begin * width * 3, root_film + (size_t)rows.begin * width * 3,
                (size_t)rows.size() * width * 3 * sizeof(float)}); };
            if (bands) add(split_range((int)height, r, ngpu));
            else for_each_tile((int)height, r, ngpu, kTileRows, add);
        }
        std::string err;
        gather_s = group.gather_to_root(pieces, &err);
        if (gather_s < 0) fail(err);
        rodent_present(group.device(0));
    }
    // --check: the same frames once more on the first device alone; the two films may differ by the order of their fp32 atomic adds only
    std::string verdict;
    if (ngpu > 1 && check) {
        const size_t words = width * height * 3;
        const std::vector<float> gathered(get_pixels(), get_pixels() + words);
        clear_pixels();
        for (uint32_t it = 0; it < iter; it++) rodent_hip_render_rows(group.device(0), &settings, (int32_t)it, 0, (int32_t)height, nullptr);
        rodent_present(group.device(0));
        const float* alone = get_pixels();
        size_t off = 0; double worst = 0;
        for (size_t k = 0; k < words; k++) {
            const double d = std::fabs((double)alone[k] - gathered[k]), tol = 1e-5 * std::fabs((double)alone[k]) + 1e-6 * iter;
            if (d > tol) off++;
            worst = std::max(worst, d / (std::fabs((double)alone[k]) + 1e-6 * iter));
        }
        verdict = std::string("the gathered film ") + (off == 0 ? "EQUALS"
            : "DIFFERS FROM") + " the first device's own render of the whole frame(s) (1e-5 relative; largest relative difference "
                  + std::to_string(worst) + (off ? ", " + std::to_string(off) + " values off" : "") + ")";
        std::copy(gathered.begin(), gathered.end(), get_pixels());       // the image that is saved is the multi-GPU one
    }

    if (!out_file.empty()) {                                             // driver.cpp:138-162
        const float* film = get_pixels();
        const float inv_iter = 1.0f / iter, inv_gamma = 1.0f / 2.2f;
        std::vector<uint8_t> px(width * height * 4);
        for (size_t k = 0; k < width * height; k++) {
            for (int c = 0; c < 3; c++) px[4 * k + c] =
                (uint8_t)(std::min(std::max(std::pow(film[3 * k + c] * inv_iter, inv_gamma), 0.0f), 1.0f) * 255.0f);
            px[4 * k + 3] = 255;
        }
        if (!write_png(out_file, px.data(), (int)width, (int)height, 4)) fail("Failed to save PNG file '" + out_file + "'");
        std::cout << "Image saved to '" << out_file << "'" << std::endl;
    }
    cleanup_interface();
    for (int r = 0; r < ngpu; r++) rodent_hip_scene_destroy(group.device(r));

    std::sort(samples_sec.begin(), samples_sec.end());
    std::cout << "# " << samples_sec.front() * 1e-6 << "/" << samples_sec[samples_sec.size() / 2] * 1e-6 << "/" << samples_sec.back() * 1e-6
              << " (min/med/max Msamples/s)" << std::endl;
    if (ngpu > 1) {
        const int own_rows = bands ? split_range((int)height, 0, ngpu).size() : tile_rows_of_rank((int)height, 0, ngpu, kTileRows);
        std::cout << "# GPUs: " << ngpu << " (devices" << [&] { std::string l;
            for (int r = 0; r < ngpu; r++) l += " " + std::to_string(group.device(r)); return l;
            }() << "), " << (bands ? "bands of " + std::to_string(own_rows) + " row(s)"
            : "interleaved tiles of " + std::to_string(kTileRows) + " rows")
                  << "; film gather to device " << group.device(0) << ": " << double(height - own_rows) * width * 12 / 1e6 << " MB in "
                      << gather_s * 1e3 << " ms" << std::endl;
        std::cout << "# Collective: " << group.describe() << std::endl;
        std::cout << "# Render ms per rank (last frame):";
        for (double ms : rank_ms) std::cout << " " << ms;
        std::cout << std::endl;
        if (check) std::cout << "# Check: " << verdict << std::endl;
        if (check && verdict.find("DIFFERS") != std::string::npos) return 2;
    }
    return 0;
}
