// services.hip -- the host services of the renderer ABI that load scene data onto a device: what the reference's
// generated kernel side calls back into (prototypes src/render/driver.impala:3-20, implementation
// src/driver/interface.cpp:432-492,584-619,665-673):
//   rodent_load_buffer        one LZ4 buffer file (data/*.bin, src/driver/buffer.h)       -> device bytes
//   rodent_load_bvh{2_tri1,4_tri4,8_tri4}  the matching layout of data/bvh.bin            -> device nodes + tris
//   rodent_load_png / _jpg    a texture file, RGBA8, rows flipped, gamma 2.2 (image.cpp)  -> device texels
//   clock_us                  microsecond clock
// Everything returned is owned by the library, cached by (device, file name) like the reference's Interface
// (interface.cpp:324-339,395-423,456-492) and valid until cleanup_interface().  Failures print a message and abort()
// like the reference's error() (interface.cpp:436,452,462,477,489).  `dev` is the HIP device index (the reference
// packs AnyDSL platform and index into it; its host device 0 has no counterpart here: this library computes nothing on
// the CPU).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "rodent_render.h"
#include "../host/buffer_io.h"
#include "../host/image.h"

namespace {

#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t err_ = (expr);                                                              \
        if (err_ != hipSuccess) {                                                              \
            fprintf(stderr, "rodent_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(err_), __FILE__, __LINE__); \
            abort();                                                                           \
        }                                                                                      \
    } while (0)

[[noreturn]] void fail(const char* what, const char* file) { fprintf(stderr, "rodent_hip: %s '%s'\n", what, file); abort(); }

struct DeviceBlock { void* ptr = nullptr; size_t bytes = 0; };
struct DeviceBvh { DeviceBlock nodes, tris; };
struct DeviceImage { DeviceBlock pixels; int width = 0, height = 0; };

struct Cache {
    std::map<std::pair<int, std::string>, DeviceBlock> buffers;
    std::map<std::tuple<int, int, std::string>, DeviceBvh> bvhs;              // (dev, node size, file)
    std::map<std::pair<int, std::string>, DeviceImage> images;
};
Cache g_cache;
std::mutex g_cache_mutex;

void select_device(int dev) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || dev < 0 || dev >= count) {
        fprintf(stderr, "rodent_hip: no HIP device %d (%d visible)\n", dev, count); abort(); }
    HIP_CHECK(hipSetDevice(dev));
}

// The calling thread's current device is left as it was (the reference's copy_to_device does not touch it either).
DeviceBlock copy_to_device(int dev, const void* host, size_t bytes) {           // interface.cpp:413-430
    int previous = 0;
    HIP_CHECK(hipGetDevice(&previous));
    select_device(dev);
    DeviceBlock b; b.bytes = bytes;
    HIP_CHECK(hipMalloc(&b.ptr, bytes ? bytes : 16));
    if (bytes) HIP_CHECK(hipMemcpy(b.ptr, host, bytes, hipMemcpyHostToDevice));
    HIP_CHECK(hipSetDevice(previous));
    return b;
}

// The loaders return COPIES of the cache entries, made while the lock is held: a concurrent cleanup_interface() may erase the
// map node, the pointers in the copy stay what the caller was promised ("valid until cleanup_interface()").  The cache is
// keyed by (device, file name) like the reference's (interface.cpp:432-468): a file rewritten between two calls is NOT reloaded.
template <typename Node, typename Tri>
DeviceBvh load_bvh(int dev, const char* file) {                                // interface.cpp:395-423,432-454
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    const auto key = std::make_tuple(dev, (int)sizeof(Node), std::string(file));
    auto it = g_cache.bvhs.find(key);
    if (it != g_cache.bvhs.end()) return it->second;
    std::vector<Node> nodes; std::vector<Tri> tris;
    FILE* probe = fopen(file, "rb");
    if (!probe) fail("Cannot open BVH", file);
    fclose(probe);
    if (!rodent::load_bvh_bin(file, nodes, tris)) fail("Invalid BVH file", file);
    DeviceBvh b;
    b.nodes = copy_to_device(dev, nodes.data(), nodes.size() * sizeof(Node));
    b.tris = copy_to_device(dev, tris.data(), tris.size() * sizeof(Tri));
    return g_cache.bvhs[key] = b;
}

DeviceImage load_image(int dev, const char* file, bool png) {                  // interface.cpp:470-492
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    const auto key = std::make_pair(dev, std::string(file));
    auto it = g_cache.images.find(key);
    if (it != g_cache.images.end()) return it->second;
    rodent::ImageRgba8 img; std::string err;
    if (!(png ? rodent::load_png(file, img, &err) : rodent::load_jpg(file, img, &err))) {
        fprintf(stderr, "rodent_hip: Cannot load %s file '%s'%s%s\n", png ? "PNG" : "JPG", file, err.empty() ? "" : ": ", err.c_str());
        abort();
    }
    DeviceImage d; d.width = img.width; d.height = img.height;
    d.pixels = copy_to_device(dev, img.pixels.data(), img.pixels.size());
    return g_cache.images[key] = d;
}

} // namespace

// called by cleanup_interface() (render.hip): the reference frees these with its Interface singleton (interface.cpp:516-518)
void rodent_services_cleanup() {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    int previous = 0;
    const bool have_device = hipGetDevice(&previous) == hipSuccess;
    for (auto& kv : g_cache.buffers) { (void)hipSetDevice(kv.first.first); (void)hipFree(kv.second.ptr); }
    for (auto& kv : g_cache.bvhs) { (void)hipSetDevice(std::get<0>(kv.first)); (void)hipFree(kv.second.nodes.ptr);
        (void)hipFree(kv.second.tris.ptr); }
    for (auto& kv : g_cache.images) { (void)hipSetDevice(kv.first.first); (void)hipFree(kv.second.pixels.ptr); }
    if (have_device) (void)hipSetDevice(previous);
    g_cache = Cache();
}

extern "C" {

uint8_t* rodent_load_buffer(int32_t dev, const char* file) {                   // interface.cpp:456-468,596-599
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    const auto key = std::make_pair((int)dev, std::string(file));
    auto it = g_cache.buffers.find(key);
    if (it != g_cache.buffers.end()) return (uint8_t*)it->second.ptr;
    FILE* f = fopen(file, "rb");
    if (!f) fail("Cannot open buffer", file);
    std::vector<uint8_t> bytes;
    const bool ok = rodent::read_buffer(f, bytes);
    fclose(f);
    if (!ok) fail("Invalid buffer file", file);
    return (uint8_t*)(g_cache.buffers[key] = copy_to_device(dev, bytes.data(), bytes.size())).ptr;
}

int64_t rodent_hip_buffer_size(int32_t dev, const char* file) {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_cache.buffers.find(std::make_pair((int)dev, std::string(file)));
    return it == g_cache.buffers.end() ? -1 : (int64_t)it->second.bytes;
}

void rodent_load_bvh2_tri1(int32_t dev, const char* file, Node2** nodes, Tri1** tris) {     // interface.cpp:601-605
    const DeviceBvh b = load_bvh<Node2, Tri1>(dev, file);
    *nodes = (Node2*)b.nodes.ptr; *tris = (Tri1*)b.tris.ptr;
}
void rodent_load_bvh4_tri4(int32_t dev, const char* file, Node4** nodes, Tri4** tris) {     // interface.cpp:607-611
    const DeviceBvh b = load_bvh<Node4, Tri4>(dev, file);
    *nodes = (Node4*)b.nodes.ptr; *tris = (Tri4*)b.tris.ptr;
}
void rodent_load_bvh8_tri4(int32_t dev, const char* file, Node8** nodes, Tri4** tris) {     // interface.cpp:613-617
    const DeviceBvh b = load_bvh<Node8, Tri4>(dev, file);
    *nodes = (Node8*)b.nodes.ptr; *tris = (Tri4*)b.tris.ptr;
}
void rodent_hip_bvh_counts(int32_t dev, const char* file, int32_t bvh_width, int32_t* num_nodes, int32_t* num_tris) {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    const int node_size = bvh_width == 2 ? (int)sizeof(Node2) : bvh_width == 4 ? (int)sizeof(Node4) : (int)sizeof(Node8);
    auto it = g_cache.bvhs.find(std::make_tuple((int)dev, node_size, std::string(file)));
    *num_nodes = *num_tris = -1;
    if (it == g_cache.bvhs.end()) return;
    *num_nodes = (int32_t)(it->second.nodes.bytes / node_size);
    *num_tris = (int32_t)(it->second.tris.bytes / (bvh_width == 2 ? sizeof(Tri1) : sizeof(Tri4)));
}

void rodent_load_png(int32_t dev, const char* file, uint8_t** pixels, int32_t* width, int32_t* height) {   // interface.cpp:584-589
    const DeviceImage img = load_image(dev, file, true);
    *pixels = (uint8_t*)img.pixels.ptr; *width = img.width; *height = img.height;
}
void rodent_load_jpg(int32_t dev, const char* file, uint8_t** pixels, int32_t* width, int32_t* height) {   // interface.cpp:591-595
    const DeviceImage img = load_image(dev, file, false);
    *pixels = (uint8_t*)img.pixels.ptr; *width = img.width; *height = img.height;
}

#ifndef RODENT_HIP_SOURCE_DIGEST
#define RODENT_HIP_SOURCE_DIGEST "unknown"
#endif
// rodent_amd/build.py source_digest(), passed by the build
const char* rodent_hip_source_digest(void) { return RODENT_HIP_SOURCE_DIGEST; }

// interface.cpp:665-673 (its non-x86 branch: a monotonic clock)
int64_t clock_us(void) {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // extern "C"
