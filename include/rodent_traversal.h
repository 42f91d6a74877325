/*
 * rodent_traversal.h -- C ABI of the MI355X traversal library (librodent_hip.so).
 *
 * This is the hand-written equivalent of the header AnyDSL generates for
 * Rodent's bench_traversal ("traversal.h", git-ignored in the reference,
 * produced by anydsl_runtime_wrap at tools/bench_traversal/CMakeLists.txt:13-17).
 * Struct field order and sizes follow the Impala definitions:
 *   Node2/Tri1        src/traversal/mapping_gpu.impala:3-16
 *   Node4/Node8/Tri4  src/traversal/mapping_cpu.impala:3-22
 *   Ray1/4/8 Hit1/4/8 tools/bench_traversal/bench_traversal.impala:25-65
 *
 * Every entry point takes plain pointers and sizes.  GPU entry points take
 * DEVICE pointers that the caller allocated and filled (the reference does
 * the same through anydsl::Array + anydsl::copy, tools/common/load_bvh.h:64-68,
 * tools/common/load_rays.h:85-88); the callee neither allocates nor frees them.
 * All entry points are synchronous unless their name ends in `_async`
 * (the reference syncs the device before returning,
 * tools/bench_traversal/bench_traversal.impala:509).
 *
 * Errors: like the reference (bench_traversal.impala:17-21) a runtime failure
 * prints a message to stderr and abort()s; there are no return codes on the
 * reference-named entry points.
 */
#ifndef RODENT_TRAVERSAL_H
#define RODENT_TRAVERSAL_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BVH / ray / hit layouts (bit-compatible with the reference) -------- */

struct Node2 {              /* 64 B  mapping_gpu.impala:3-7 */
    float   bounds[12];     /* child0: lo_x hi_x lo_y hi_y lo_z hi_z, then child1 */
    int32_t child[2];       /* >0 inner (index+1), <0 leaf (~first tri), 0 none */
    int32_t pad[2];
};

struct Tri1 {               /* 48 B  mapping_gpu.impala:9-16 */
    float   v0[3];  int32_t pad;
    float   e1[3];  int32_t geom_id;   /* e1 = v0 - v1 */
    float   e2[3];  int32_t prim_id;   /* e2 = v2 - v0; bit 31 = last in leaf */
};

struct Node4 {              /* 128 B mapping_cpu.impala:12-16 */
    float   bounds[6][4];   /* rows lo_x hi_x lo_y hi_y lo_z hi_z; column = child */
    int32_t child[4];
    int32_t pad[4];
};

struct Node8 {              /* 256 B mapping_cpu.impala:18-22 */
    float   bounds[6][8];
    int32_t child[8];
    int32_t pad[8];
};

struct Tri4 {               /* 224 B mapping_cpu.impala:3-10 */
    float   v0[3][4], e1[3][4], e2[3][4], n[3][4];
    int32_t prim_id[4];     /* -1 = unused lane; bit 31 of [3] = last packet in leaf */
    int32_t geom_id[4];
};

struct Ray1 { float org[3]; float tmin; float dir[3]; float tmax; };          /*  32 B */
struct Ray4 { float org[3][4], dir[3][4], tmin[4], tmax[4]; };                /* 128 B */
struct Ray8 { float org[3][8], dir[3][8], tmin[8], tmax[8]; };                /* 256 B */
struct Hit1 { int32_t tri_id; float t, u, v; };                               /*  16 B */
struct Hit4 { int32_t tri_id[4]; float t[4], u[4], v[4]; };                   /*  64 B */
struct Hit8 { int32_t tri_id[8]; float t[8], u[8], v[8]; };                   /* 128 B */

/* ---- Entry points that exist in the reference --------------------------- */

/* Replaces tools/bench_traversal/bench_traversal.impala:495-511.
 * Closest hit, one ray per lane, BVH2/Tri1.  hits[i] = {prim id or -1, t, u, v};
 * a miss stores tri_id = -1 and t = rays[i].tmax (intersection.impala:134-136). */
void amdgpu_intersect_single_ray1_bvh2_tri1(int32_t dev,
        const struct Node2* nodes, const struct Tri1* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays);

/* Replaces tools/bench_traversal/bench_traversal.impala:513-529 (any-hit:
 * returns at the first accepted triangle). */
void amdgpu_occluded_single_ray1_bvh2_tri1(int32_t dev,
        const struct Node2* nodes, const struct Tri1* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays);

/* ---- New entry points, same naming pattern ------------------------------ */

/* BVH4/Tri4 and BVH8/Tri4 on the GPU (the reference only has these layouts on the CPU,
 * cpu_{intersect,occluded}_single_ray1_bvh{4,8}_tri4, bench_traversal.impala:399-455).  Per-ray visit order: the
 * reference GPU kernel's branch for arity != 2 (src/traversal/mapping_gpu.impala:136-153,160-169).
 * Which layout for what (one MI355X, profiles/r03_sweep_widths.log): closest hits and any query about incoherent rays are
 * fastest on BVH2 / Tri1; visibility (occluded_*) of COHERENT rays is fastest on BVH8 / Tri4 -- 24-33 % ahead of BVH2. */
void hip_intersect_single_ray1_bvh4_tri4(int32_t dev,
        const struct Node4* nodes, const struct Tri4* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays);
void hip_occluded_single_ray1_bvh4_tri4(int32_t dev,
        const struct Node4* nodes, const struct Tri4* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays);
void hip_intersect_single_ray1_bvh8_tri4(int32_t dev,
        const struct Node8* nodes, const struct Tri4* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays);
void hip_occluded_single_ray1_bvh8_tri4(int32_t dev,
        const struct Node8* nodes, const struct Tri4* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays);

/* Asynchronous forms: enqueue on `stream` (a hipStream_t passed as void*,
 * NULL = the device's null stream) and return without synchronising, so a
 * caller can bracket launches with its own HIP events.  `variant` selects the
 * kernel mapping (see rodent_hip_variant_name); 0 is the default shipped one.
 * Every (device, stream) pair has its own control words and deep-ray list (rays whose stack outgrows the LDS window
 * are traced again with the reference's 64-entry stack: by the launch's last workgroup in the default BVH2 mapping, by a
 * follow-up kernel enqueued behind the launch in the others), so launches on different streams may overlap;
 * up to 64 streams per device. */
void hip_traverse_bvh2_tri1_async(int32_t dev,
        const struct Node2* nodes, const struct Tri1* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays,
        int32_t any_hit, int32_t variant, void* stream);
void hip_traverse_bvh4_tri4_async(int32_t dev,
        const struct Node4* nodes, const struct Tri4* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays,
        int32_t any_hit, int32_t variant, void* stream);
void hip_traverse_bvh8_tri4_async(int32_t dev,
        const struct Node8* nodes, const struct Tri4* tris,
        const struct Ray1* rays, struct Hit1* hits, int32_t num_rays,
        int32_t any_hit, int32_t variant, void* stream);
/* The reference's stack holds 64 entries, unchecked (src/traversal/stack.impala:53-54).  Here a ray that needs more
 * raises a device-side flag: the synchronous entry points above abort() on it like the reference's runtime failures
 * (bench_traversal.impala:17-21); after the asynchronous forms call this -- it waits for `stream`, returns 1 if any
 * launch enqueued on it since the last call overflowed (those rays' hits are undefined), and clears the flag. */
int32_t rodent_hip_check_errors(int32_t dev, void* stream);

/* Introspection / plumbing. */
/* anydsl_get_kernel_time() of the AnyDSL runtime, which the reference's bench_traversal brackets its GPU calls with
 * (tools/bench_traversal/bench_traversal.cpp:125-133): microseconds of KERNEL time accumulated over the synchronous entry points above (HIP
 * events around what a call enqueues, after the context's buffers have been allocated: the call's synchronisation is not in it; where a
 * mapping is two kernels -- launches under rodent_hip_top_min_rays: the one-chunk kernel and its follow-up -- the microsecond between them
 * is.  One sum per process over all devices, like the reference's; one synchronous call at a time per (device, null stream). */
uint64_t    rodent_hip_get_kernel_time(void);
int32_t     rodent_hip_device_count(void);              /* 0 when no GPU is visible */
int32_t     rodent_hip_num_variants(int32_t bvh_width); /* bvh_width: 2, 4 or 8 */
/* The phased BVH2 mappings ("phased-*": capped phases with ray compaction in between) take the single kernel for launches
 * of fewer than this many rays (default 262 144: less than one round of resident waves); < 0 restores the default, 0 makes every launch
 * phased (tests). */
void        rodent_hip_phased_min_rays(int32_t rays);
/* The default BVH2 mapping ("top": top of the tree staged in LDS, persistent workgroups) takes the one-chunk-per-workgroup
 * kernel ("fast") for launches of fewer than this many rays (default 393 216: the measured cross-over of the two kernels, 589 824 until
 * round 4 -- staging and validating the image in every workgroup does not pay below that); < 0 restores the default, 0 sends every launch
 * through the LDS-image kernel (tests). The default BVH4 / BVH8 mapping ("top": persistent workgroups that stage the top 85 / 73 nodes
 * themselves) switches to ITS one-chunk kernel ("single") at the same size. */
void        rodent_hip_top_min_rays(int32_t rays);
/* Schedule history of the default BVH2 mapping (off by default; RODENT_HIP_SCHEDULE_HISTORY=1): every launch records how many
 * wave iterations each 64-ray chunk took, and the next launch of the SAME ray count on that (device, stream) traces its chunks
 * longest first -- frame-to-frame cost feedback for callers that trace similar ray sets again and again (1 Mi primary rays of the
 * atrium: 0.187 -> 0.150 ms).  It changes only the order in which chunks are traced, never a hit record; a launch without a
 * usable history takes the default order.  State is per (device, stream); launches of more than 4 Mi rays do not use it (they are
 * throughput-bound: measured -8 % at 16 Mi). */
void        rodent_hip_schedule_history(int32_t enable);
/* Ray-kind hint of the default BVH2 mapping (OFF by default from round 5 on; RODENT_HIP_KIND_HINT=1).  The default kernel decides per
 * wavefront, once, from the first 64 rays the wave draws, whether they share an origin or a direction (traced as whole 64-ray chunks) or
 * not (idle lanes are refilled: the compaction of BASELINE config 3; reference: render/mapping_gpu.impala:267-300 compacts unconditionally)
 * -- no state between launches. With the hint on, every workgroup also notes what it saw in a host-visible word, and a list -- same
 * pointer, same count -- that all workgroups of the earlier launches found incoherent is traced by the kernel specialised for that
 * ("refill") from its second launch on (+2 ... 4 % on random segments): kernel selection then depends on earlier launches and, for
 * asynchronous callers, on when they finished.  Hit records never depend on it. */
void        rodent_hip_ray_kind_hint(int32_t enable);
/* Camera rays in image order (round 5; RODENT_HIP_RAY_GRID).  The reference's primary-ray dumps are the pixels of an image row by row
 * (tools/ray_gen/ray_gen.cpp:20-58: dir = d + kx r + ky u, not normalised).  Which rays share a wavefront is the callee's business: the
 * default BVH2 kernel recognises such a list from 66 of its rays -- no state between launches, no probe launch -- and gives every wavefront
 * an 8 x 8-pixel tile instead of 64 pixels of one row (the rays of a tile finish closer together and share more nodes: 1 Mi camera rays on
 * the atrium 0.178 -> 0.167 ms).  Per-pixel lists that are no camera dump (ray_gen's shadow mode, tools/ray_gen/ray_gen.cpp:60-85; a
 * renderer's shadow rays in pixel order) are recognised when the width is a multiple of 128 that divides the ray count.  Hit records and
 * their places in `hits` do not change. width: -1 = recognise (default), 0 = never (list order, as until round 4), > 0 = take this image
 * width on trust (experiments; multiples of 8 only, others mean 0). */
void        rodent_hip_ray_grid(int32_t width);
/* 1 = librodent_hip_lab.so (-DRODENT_HIP_LAB: also the measured-and-lost kernels) */
int32_t     rodent_hip_is_lab_build(void);
const char* rodent_hip_variant_name(int32_t bvh_width, int32_t variant);
const char* rodent_hip_kernel_name(int32_t bvh_width, int32_t variant, int32_t any_hit);
const char* rodent_hip_version(void);
/* Digest of the sources this binary was compiled from (rodent_amd/build.py source_digest(): sha256 over the files of csrc/, the host code
 * the library links and every header): a prebuilt library can be checked against the sources lying next to it. */
const char* rodent_hip_source_digest(void);
/* Debug aid: reads and clears the 8 phase counters of the instrumented "stats-*" variants
 * ([0] descent iterations, [1] lanes active in them, [2] leaf iterations, [3] lanes, [4] refills, [5] lanes refilled, [6] outer
 * iterations); from the shipped BVH2 mappings: [2] image width a launch traced 8 x 8 tiles of, [4] launches of the refill kernel by the
 * ray-kind hint, [5] launches whose first wavefront chose the refill loop, [6] launches that ran on a validated LDS image, [7] stack blocks
 * spilled + rays handed to the deep pass. */
void        rodent_hip_read_stats(int32_t dev, uint64_t* out8);
/* Debug aid: per-wave timeline of the instrumented variants.  First call (out may be NULL) arms it;
 * later calls copy 16384 x 4 words {start tick, end tick (100 MHz), hw_id | xcc_id << 32,
 * outer iterations | rays << 32} and clear the buffer. */
void        rodent_hip_read_trace(int32_t dev, uint64_t* out);
/* Lab build: the ray permutation (device pointer, one int per ray, owned by the caller) that the "top-userperm" mapping of the
 * context used last traces through; nullptr = none.  For scheduling experiments (scripts/lpt_experiment.py). */
void        rodent_hip_debug_set_perm(int32_t dev, const int32_t* device_perm);

#ifdef __cplusplus
}
#endif
#endif /* RODENT_TRAVERSAL_H */
