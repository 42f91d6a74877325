/*
 * rodent_render.h -- C ABI of the MI355X wavefront path tracer (librodent_hip.so).
 *
 * Stands where the reference has the AnyDSL-generated `interface.h` (src/CMakeLists.txt:76-79)
 * plus the `extern "C"` services of src/driver/interface.cpp:565-675.  In the reference the
 * scene is compiled into `render()` by the converter (src/driver/converter.cpp:613-967 emits
 * Impala source); here the scene is DATA: tables uploaded once with rodent_hip_scene_create(),
 * and spp / max path length are run-time values (rodent_hip_render_config) instead of
 * converter flags (converter.cpp:1007-1012).
 *
 * Kept from the reference, same names and meaning:
 *   Settings, get_spp(), render(settings, iter)                 src/dummy_main.impala:3-13
 *   RayStream / PrimaryStream / SecondaryStream (SoA slabs)      src/render/driver.impala:24-61
 *   setup_interface / get_pixels / clear_pixels / cleanup_interface   src/driver/interface.cpp:512-526
 *   rodent_get_film_data, rodent_gpu_get_{first,second}_primary_stream,
 *   rodent_gpu_get_secondary_stream, rodent_gpu_get_tmp_buffer, rodent_present   interface.cpp:567-582,631-663
 * Device ids: `dev` is the HIP device ordinal (the reference packs platform and index,
 * src/render/mapping_gpu.impala:538,596; 0 there means "host").
 * Errors: message on stderr + abort(), like src/driver/common.h:43-59.
 */
#ifndef RODENT_RENDER_H
#define RODENT_RENDER_H

#include <stdint.h>
#include <stddef.h>
#include "rodent_traversal.h"

#ifdef __cplusplus
extern "C" {
#endif

struct Vec3 { float x, y, z; };
/* width/height = tan(fov/2), that / ratio (driver.cpp:37-38) */
struct Settings { struct Vec3 eye, dir, up, right; float width, height; };

struct RayStream { int32_t* id; float *org_x, *org_y, *org_z, *dir_x, *dir_y, *dir_z, *tmin, *tmax; };
struct PrimaryStream {
    struct RayStream rays; int32_t *geom_id, *prim_id; float *t, *u, *v;
    uint32_t* rnd; float *mis, *contrib_r, *contrib_g, *contrib_b; int32_t* depth; int32_t size, pad;
};
struct SecondaryStream { struct RayStream rays; int32_t* prim_id; float *color_r, *color_g, *color_b; int32_t size, pad; };

/* ---- scene tables (what the converter bakes into Impala source in the reference) ---- */
enum RodentBsdf { RODENT_BSDF_BLACK = 0, RODENT_BSDF_DIFFUSE = 1, RODENT_BSDF_PHONG = 2,
                  RODENT_BSDF_MIX = 3 /* diffuse (+) phong */, RODENT_BSDF_MIRROR = 4, RODENT_BSDF_GLASS = 5 };
struct RodentMaterial {            /* 64 B; one per geometry id (converter.cpp:858-920) */
    float kd[3]; int32_t type; float ks[3]; float ns; float tf[3]; float ni;
    float mix_k;                   /* lum(ks) / (lum(ks) + lum(kd)) (converter.cpp:900-906); recomputed per hit when textured */
    int32_t emissive;
    /* 0: kd / ks are the constants above; else 1 + index of the map_Kd / map_Ks texture (converter.cpp:881-893) */
    int32_t tex_kd, tex_ks;
};
struct RodentTexture {             /* RGBA8 image, row 0 = bottom row of the file, gamma-corrected (src/driver/image.cpp:10-18,85) */
    int32_t width, height; uint32_t offset /* first texel in the pool */; int32_t pad;
};
struct RodentLight {               /* 80 B; triangle area light (converter.cpp:831-851, light.impala:140-154) */
    float v0[4], v1[4], v2[4]; float n[3]; float inv_area; float color[4];
};
struct RodentSceneDesc {           /* HOST pointers; copied to HBM by rodent_hip_scene_create */
    const float* vertices;         /* float4 per vertex (padded like the GPU targets, converter.cpp:629-632) */
    const float* normals;          /* float4 per vertex */
    const float* face_normals;     /* float4 per triangle */
    const int32_t* indices;        /* int4 per triangle: v0 v1 v2 material (obj.cpp:455-461) */
    const struct Node2* nodes; const struct Tri1* tris;     /* BVH2/Tri1, geom_id = material id */
    const struct RodentMaterial* materials; const struct RodentLight* lights;
    const int32_t* light_ids;      /* per triangle: index into lights (0 if not a light) */
    int32_t num_vertices, num_tris, num_nodes, num_bvh_tris, num_materials, num_lights;
    /* textures (src/render/image.impala:56-92: repeat border, bilinear filter); all three may be NULL when num_textures == 0 */
    const float* texcoords;        /* float4 per vertex: u, v, 0, 0 (converter.cpp:408) */
    const struct RodentTexture* textures; const uint32_t* texels;
    int32_t num_textures; uint32_t num_texels;
};

void    rodent_hip_scene_create(int32_t dev, const struct RodentSceneDesc* desc);   /* replaces the device's current scene */
void    rodent_hip_scene_destroy(int32_t dev);
void    rodent_hip_render_config(int32_t dev, int32_t spp, int32_t max_path_len);   /* defaults 4 / 64 (converter.cpp:1007-1012) */
/* 0 = streaming wavefront loop (src/render/mapping_gpu.impala:308-369), 1 = persistent-threads megakernel
 * (mapping_gpu.impala:371-474; the reference selects it at configure time with the converter target
 * amdgpu-megakernel / nvvm-megakernel, converter.cpp:30-35,1032-1037; `rodent --target amdgpu-megakernel` here),
 * -1 (default) = chosen per scene when the scene is created: the megakernel for hierarchies of at most
 * RODENT_HIP_AUTO_MEGA_MAX_NODES inner nodes (default 128: the whole tree sits in the traversal kernels' LDS image and the
 * wavefront formulation's stream traffic is all that is left to save; the Cornell box renders 1.45 x faster that way), the
 * streaming loop for larger scenes (1.05 x faster at 306 nodes, 1.4-1.5 x from 5 000 nodes on: profiles/r03_mapping_sweep.txt).  The
 * initial value can also be set with the environment variable
 * RODENT_HIP_MAPPING=auto|streaming|mega.  rodent_hip_render_mapping_in_effect: 0 / 1, what the next frame will use. */
void    rodent_hip_render_mapping(int32_t dev, int32_t mapping);
int32_t rodent_hip_render_mapping_in_effect(int32_t dev);
/* Every rodent_hip_render_* option of the device back to its default, or to what its RODENT_HIP_* environment variable says
 * (the values a fresh process starts with); spp / max_path_len and the scene are kept. */
void    rodent_hip_render_defaults(int32_t dev);
/* Rays per ray stream of the streaming mapping: the reference's constant 1 Mi (mapping_gpu.impala:319) is 32 Mi here
 * by default (larger launches amortise their fill and drain on a 256-CU chip: 7.1 GB of streams out of 288 GB; 0 restores the default). */
void    rodent_hip_render_capacity(int32_t dev, int32_t rays);
/* 1: hit rays are sorted by material before shading and misses dropped, as in the reference (mapping_gpu.impala:166-221,347-357:
 * there every material is its own generated shader and the sort is what makes a launch per material possible).
 * 0 (default): no sort -- the shader here is ONE table-driven kernel; it runs in stream order and ends the rays that missed; one
 * stream copy less per bounce.  Measured on every scene at hand (profiles/r03_sort_sweep.txt: Cornell, the atrium, a room with
 * every BSDF kind on neighbouring walls, a textured room) the unsorted loop is 5 ... 22 % faster, so it is the default; the sort
 * stays one call away.  Same paths, same ray counts; RODENT_HIP_SORT=0|1 sets the initial value, `rodent --sort` / `--no-sort`. */
void    rodent_hip_render_sort(int32_t dev, int32_t enable);
/* Hit records inside the library's own wavefront loop: 1 (default) = one 20-byte record per ray in the memory of the stream's geom_id /
 * prim_id / t / u / v arrays (one 16-byte + one 4-byte store per record instead of five scattered 4-byte stores: the traversal launches'
 * write traffic), 0 = the ABI's five arrays.  The stage-level entry points (hip_traverse_primary, hip_shade, ...) always use the five
 * arrays, and so does the loop when the sort by material is on.  RODENT_HIP_HIT_AOS. */
void    rodent_hip_render_hit_records(int32_t dev, int32_t aos);
/* 1 (default): the shadow rays of a bounce are traced on a second HIP stream beside the compaction, regeneration and the
 * next closest-hit pass; 0: one stream.  Same film up to the order of the atomic adds.  RODENT_HIP_OVERLAP=0|1.
 * (Without effect while the joint traversal launch is in use, rodent_hip_render_trace_persistent below: that loop has one stream.) */
void    rodent_hip_render_overlap(int32_t dev, int32_t enable);
/* 0 (default): rays are moved by the sort (copy_primary_ray, mapping_gpu.impala:136-164) and shaded in place.  1: the sort by
 * material only computes the permutation; the shader gathers its rays through it and writes the sorted stream (one copy of the
 * 18-word stream per bounce less, but a gathering shader: measured 3 % slower on the Cornell box, equal on the atrium).
 * Same paths and film.  RODENT_HIP_FUSED_SORT=0|1. */
void    rodent_hip_render_fused_sort(int32_t dev, int32_t enable);
/* Compaction as part of the shader: every ray that goes on is written straight to its compacted slot of the other stream, one
 * read + write of the 15-word stream per bounce less than shading in place and compacting afterwards.
 * 2 (default): a block of 256 rays takes its slots from one atomic counter (order of the blocks in the new stream = order
 * of arrival; inside a block stream order; the reference's own compaction takes one atomic per RAY, mapping_gpu.impala:293).
 * 1: slots from a single-pass block scan with decoupled look-back -- the stable order of the separate pass, reproducible
 * from run to run, but every block waits for its 64 predecessors (measured: the shader 290 -> 535 us per 8 Mi rays).
 * 0: shade in place, then the separate compaction pass (gpu_compact_primary, mapping_gpu.impala:267-300).
 * Same rays in the stream either way, same film up to the order of the atomic adds.  RODENT_HIP_FUSED_COMPACT=0|1|2. */
void    rodent_hip_render_fused_compact(int32_t dev, int32_t enable);
/* 1 (default): the stream traversal kernels run as 2-wave workgroups that stage the first 31 inner nodes of the scene's BVH
 * (breadth first, built at scene creation) in LDS and fetch those with ds_read instead of through the vector-memory pipeline.
 * 0: one wave per workgroup, every node from memory (rounds 1-2).  Same per-ray visit order, same film.  RODENT_HIP_LDS_IMAGE=0|1. */
void    rodent_hip_render_lds_image(int32_t dev, int32_t enable);
/* Megakernel mapping.  0 (default): the reference's sequence of loops (closest hit, shade, shadow; mapping_gpu.impala:371-474).
 * 1: a lane traces the shadow ray of its path vertex and the path's next ray back to back in ONE wave-level loop (a wave needs max
 * over its lanes of the SUM of the two rays' steps instead of the sum of two maxima) -- measured 6 ... 14 % SLOWER (a path that ends
 * with a shadow ray pending holds its lane for one more loop): an option, not the default.
 * Same paths, same ray counts, same film up to the order of the atomic adds.  RODENT_HIP_MEGA_JOINT=0|1. */
void    rodent_hip_render_mega_joint(int32_t dev, int32_t enable);
/* The stream traversal launches of the streaming loop.
 * 0: 2-wave workgroups with a 31-node image (rodent_hip_render_lds_image), the shadow pass on a second stream (rodent_hip_render_overlap).
 * 1: persistent form (one resident generation of 16-wave workgroups, the first 255 inner nodes in LDS, 64-ray chunks drawn from
 *    striped ticket counters -- traversal.hip's default mapping) for streams of at least 524 288 rays.
 * 2: joint -- the shadow pass of an iteration rides in the NEXT iteration's closest-hit launch: one persistent kernel works through
 *    both ray lists (they depend on the same shader run and on nothing else), one stream, no pass waits for the other's tail.
 * -1 (default): per scene -- 2 for every hierarchy the per-scene mapping rule sends to the streaming loop (+3 ... +9 % on the atrium at
 *    306 ... 142 444 nodes, profiles/r03_joint_sweep.txt), 0 for a tree of a few dozen nodes (Cornell box: 2 is 5 % slower).
 * Same film.  RODENT_HIP_TRACE_PERSISTENT=-1|0|1|2. */
void    rodent_hip_render_trace_persistent(int32_t dev, int32_t enable);
/* Lane refill in the persistent traversal launches (1 and 2 above).  Thresholds 1 .. 64: a wave whose idle lanes reach that count
 * retires their rays and draws as many new ones from its stripe's counter instead of waiting for the last ray of a 64-ray chunk
 * (k_trace_refill) -- idle_bounce while it draws from the rays the last bounce left, idle_shadow while it draws shadow rays (64 =
 * whole chunks for that kind); the camera rays a launch holds (the host knows where: the rays generated for it) always go chunk
 * by chunk.  0, 0: whole chunks for everything (k_trace_persist).  -1, -1 (default): per scene -- 40, 40 for hierarchies of 16 384
 * nodes and more (atrium: +8 % at 1920 x 1080 x 16 spp, +9 % at 3840 x 2160 x 32 spp; profiles/r03_refill_sweep.txt), off below
 * (every ray is short there: -4 ... -11 %).  Same paths, same ray counts, same film up to the order of the atomic adds.
 * RODENT_HIP_TRACE_REFILL=<both> or <bounce>,<shadow>. */
void    rodent_hip_render_trace_refill(int32_t dev, int32_t idle_bounce, int32_t idle_shadow);
/* idle_bounce | idle_shadow << 8 for the scene that is loaded (0 = whole chunks) */
int32_t rodent_hip_render_trace_refill_in_effect(int32_t dev);

/* ---- the reference's renderer ABI ---- */
int32_t get_spp(void);
void    render(const struct Settings* settings, int32_t iter);       /* one frame: spp samples per pixel, accumulated into the film */
void    setup_interface(size_t width, size_t height);
float*  get_pixels(void);                                            /* host film, width*height*3 floats, valid after rodent_present */
void    clear_pixels(void);
void    cleanup_interface(void);
void    rodent_get_film_data(int32_t dev, float** pixels, int32_t* width, int32_t* height);   /* DEVICE film */
void    rodent_gpu_get_first_primary_stream(int32_t dev, struct PrimaryStream* primary, int32_t size);
void    rodent_gpu_get_second_primary_stream(int32_t dev, struct PrimaryStream* primary, int32_t size);
void    rodent_gpu_get_secondary_stream(int32_t dev, struct SecondaryStream* secondary, int32_t size);
void    rodent_gpu_get_tmp_buffer(int32_t dev, int32_t** buf, int32_t size);
void    rodent_present(int32_t dev);                                 /* film device -> host */
/* Scene-data services (src/driver/interface.cpp:432-492,584-619; prototypes src/render/driver.impala:11-16): load one of the
 * reference converter's files onto device `dev` and return DEVICE pointers.  Results are cached by (dev, file name), owned
 * by the library and valid until cleanup_interface(); a missing or malformed file prints a message and abort()s like the
 * reference's error().  `dev` is the HIP device index (there is no host device 0 here: nothing is computed on the CPU). */
/* one LZ4 buffer file (data/vertices.bin, ...; src/driver/buffer.h) */
uint8_t* rodent_load_buffer(int32_t dev, const char* file);
/* the matching layout of data/bvh.bin */
void    rodent_load_bvh2_tri1(int32_t dev, const char* file, struct Node2** nodes, struct Tri1** tris);
void    rodent_load_bvh4_tri4(int32_t dev, const char* file, struct Node4** nodes, struct Tri4** tris);
void    rodent_load_bvh8_tri4(int32_t dev, const char* file, struct Node8** nodes, struct Tri4** tris);
/* RGBA8, rows flipped, gamma 2.2 (image.cpp:10-18,85) */
void    rodent_load_png(int32_t dev, const char* file, uint8_t** pixels, int32_t* width, int32_t* height);
void    rodent_load_jpg(int32_t dev, const char* file, uint8_t** pixels, int32_t* width, int32_t* height);
/* Host-side stream slabs with the device slabs' carving, one per calling thread (interface.cpp:341-342,367-373,621-629). */
void    rodent_cpu_get_primary_stream(struct PrimaryStream* primary, int32_t size);
void    rodent_cpu_get_secondary_stream(struct SecondaryStream* secondary, int32_t size);
int64_t clock_us(void);                                              /* interface.cpp:665-673 */

/* ---- additions ---- */
/* Sizes of what the services above loaded (the reference's generated code knows them at compile time):
 * bytes of a loaded buffer (-1 if `file` was not loaded on `dev`); node / primitive counts of a loaded BVH layout. */
int64_t rodent_hip_buffer_size(int32_t dev, const char* file);
void    rodent_hip_bvh_counts(int32_t dev, const char* file, int32_t bvh_width, int32_t* num_nodes, int32_t* num_tris);
void    rodent_hip_set_device(int32_t dev);                          /* device used by render() (the reference bakes it in) */
/* Renders only image rows [y0, y1) (tile sharding across GPUs: seeds depend on absolute (sample, iter, x, y),
 * renderer.impala:28-33, so any tiling reproduces the same samples).  The work is enqueued on `stream`; the call RETURNS WHEN THE
 * ROWS ARE IN THE DEVICE FILM: the wavefront loop reads the stream size back every iteration (as the reference does,
 * mapping_gpu.impala:344-366), so it cannot be asynchronous. */
void    rodent_hip_render_rows(int32_t dev, const struct Settings* settings, int32_t iter, int32_t y0, int32_t y1, void* stream);
/* Renders the interleaved row tiles first_tile, first_tile + tile_stride, ... of tile_rows rows each (the last tile of the film may be
 * shorter): GPU k of K takes first_tile = k, tile_stride = K, which balances the GPUs where contiguous bands do not (SURVEY 8e;
 * the reference deals ~1024-sample tiles dynamically, render/mapping_gpu.impala:374-420).  Same samples, same film as
 * rodent_hip_render_rows over the same rows; synchronous like it. */
void    rodent_hip_render_tiles(int32_t dev, const struct Settings* settings, int32_t iter, int32_t tile_rows, int32_t first_tile,
    int32_t tile_stride, void* stream);
/* Counters of the last render call on this device (render, rodent_hip_render_rows, rodent_hip_render_tiles -- the whole call, however
 * many launches it took): [0] primary rays traced, [1] shadow rays traced, [2] wavefront iterations, [3] rays generated. */
void    rodent_hip_render_counters(int32_t dev, uint64_t* out4);

/* The wavefront stages as separate entry points (the reference's device kernels,
 * src/render/mapping_gpu.impala:18-30,47-80,82-134,166-221,223-265,267-300); all asynchronous on `stream`. */
void    hip_generate_rays(int32_t dev, struct PrimaryStream* primary, int32_t capacity, int32_t first_ray_id, int32_t num_rays,
                          const struct Settings* settings, int32_t iter, int32_t film_width, int32_t film_height,
                          int32_t first_pixel, int32_t spp, void* stream);
void    hip_traverse_primary(int32_t dev, struct PrimaryStream* primary, void* stream);
/* Sorts by geometry id into `other` (stable, deterministic); ray_ends[g] (host, num_geometries+1 ints) receives the
 * exclusive end of bin g like mapping_gpu.impala:203-207.  Synchronises the stream (the reference does too). */
void    hip_sort_primary(int32_t dev, struct PrimaryStream* primary, struct PrimaryStream* other, int32_t* ray_ends, void* stream);
void    hip_shade(int32_t dev, struct PrimaryStream* primary, struct SecondaryStream* secondary, int32_t num_rays, void* stream);
void    hip_traverse_secondary(int32_t dev, struct SecondaryStream* secondary, void* stream);
/* Order-preserving compaction of rays with id >= 0; returns the new size (synchronises the stream). */
int32_t hip_compact_primary(int32_t dev, struct PrimaryStream* primary, struct PrimaryStream* other, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RODENT_RENDER_H */
