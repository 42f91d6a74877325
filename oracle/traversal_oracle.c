/*
 * traversal_oracle.c -- CPU restatement of Rodent's BVH traversal.  TEST INFRASTRUCTURE.
 *
 * This file is the parity oracle for the HIP traversal kernels.  It is test
 * infrastructure only: nothing under rodent_amd/ (the product) may include,
 * link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * PARITY PINNING: "parity unpinned by the reference" -- by reference OUTPUTS.  The reference's traversal
 * is Impala source that needs the AnyDSL toolchain (absent here), and its only
 * golden vectors (testing/ref-primary.png, ref-random.png) depend on sponza.bvh /
 * sponza-*.rays which are missing from the checkout (.MISSING_LARGE_BLOBS).
 * The oracle is therefore pinned by (1) an exhaustive all-triangles checker
 * (oracle_brute_force below; closest t must equal the minimum over every
 * triangle), (2) an independent float64 Moeller-Trumbore in tests/, (3) fixtures generated on this side
 * (tests/golden/), and (4) hierarchies the REFERENCE'S OWN BUILDER made (oracle/_ref/ref_bvh_builder = the reference's
 * obj.cpp + bvh.h compiled where they lie; tests/golden/<name>-refbuilt.bvh, tests/test_refbuilt.py): B1 / B1g / B2 on those against
 * the exhaustive checker.  Indirectly, the reference does hold it: the renderer
 * oracle (render_oracle.c) traces every ray with oracle_bvh2_tri1 below and reproduces the reference's own
 * golden image testing/ref-cornell.png.
 *
 * Arithmetic: fp32, IEEE, compiled with -ffp-contract=off.  Where a fused
 * multiply-add is used it is written explicitly as fmaf() so that the HIP
 * kernels (also built with -ffp-contract=off) perform the identical sequence
 * of correctly-rounded operations.  The reference leaves fusion to its compiler
 * (-O3 -march=native -ffast-math, CMakeLists.txt:12), so its own results are not
 * reproducible bit-for-bit across machines; this file fixes one rounding.
 *
 * Reference lines restated here:
 *   safe_rcp / prodsign          src/core/common.impala:78-85
 *   make_ray                     src/traversal/intersection.impala:88-99
 *   intersect_ray_tri            src/traversal/intersection.impala:164-192
 *   intersect_ray_box            src/traversal/intersection.impala:194-208
 *   stack                        src/traversal/stack.impala:52-123
 *   batcher / bose-nelson sort   src/core/sort.impala:3-66
 *   GPU single-ray BVH2 (B1)     src/traversal/mapping_gpu.impala:18-70,94-178
 *   CPU single-ray BVH4/8 (B2)   src/traversal/mapping_cpu.impala:24-116,138-256
 *   Hit1 writer                  tools/bench_traversal/bench_traversal.impala:78-83
 */
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <float.h>

struct Node2 { float bounds[12]; int32_t child[2]; int32_t pad[2]; };
struct Tri1  { float v0[3]; int32_t pad; float e1[3]; int32_t geom_id; float e2[3]; int32_t prim_id; };
struct Node4 { float bounds[6][4]; int32_t child[4]; int32_t pad[4]; };
struct Node8 { float bounds[6][8]; int32_t child[8]; int32_t pad[8]; };
struct Tri4  { float v0[3][4], e1[3][4], e2[3][4], n[3][4]; int32_t prim_id[4], geom_id[4]; };
struct Ray1  { float org[3]; float tmin; float dir[3]; float tmax; };
struct Hit1  { int32_t tri_id; float t, u, v; };

/* Per-call visit statistics (sums over all rays); these feed the
 * "algorithmic bytes per ray" figure of SURVEY.md 8(d). */
struct OracleStats {
    uint64_t rays;
    uint64_t inner_nodes;   /* inner nodes fetched                         */
    uint64_t prim_packets;  /* Tri1 (BVH2) or Tri4 packets (BVH4/8) fetched */
    uint64_t hits;          /* rays with tri_id >= 0                        */
    uint32_t max_stack;     /* deepest stack pointer seen (entries in memory) */
    uint32_t pad;
};

#define FLT_MAX_REF 3.4028234664e+38f   /* common.impala:4 */

static inline int32_t f2i(float x) { int32_t i; memcpy(&i, &x, 4); return i; }
static inline float   i2f(int32_t i) { float x; memcpy(&x, &i, 4); return x; }

/* common.impala:78-80 */
static inline float prodsign(float x, float y) {
    return i2f(f2i(x) ^ (f2i(y) & (int32_t)0x80000000u));
}
/* common.impala:82-85 */
static inline float safe_rcp(float x) {
    const float min_rcp = 1e-8f;
    return ((x > 0.0f ? x : -x) < min_rcp) ? prodsign(FLT_MAX_REF, x) : 1.0f / x;
}

/* IEEE minNum/maxNum like ocml_fminf/fmaxf (a NaN operand loses); inline because gcc
 * keeps fminf/fmaxf as libm calls without -ffinite-math-only. */
static inline float fmin_ref(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float fmax_ref(float a, float b) { return (a > b || b != b) ? a : b; }

static inline float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
#define CROSS_X(ax,ay,az,bx,by,bz) fmaf((ay), (bz), -((az) * (by)))
#define CROSS_Y(ax,ay,az,bx,by,bz) fmaf((az), (bx), -((ax) * (bz)))
#define CROSS_Z(ax,ay,az,bx,by,bz) fmaf((ax), (by), -((ay) * (bx)))

struct RayX {
    float ox, oy, oz, dx, dy, dz, idx, idy, idz, iox, ioy, ioz, tmin, tmax;
};

/* intersection.impala:88-99 */
static inline struct RayX make_ray(const struct Ray1* r) {
    struct RayX x;
    x.ox = r->org[0]; x.oy = r->org[1]; x.oz = r->org[2];
    x.dx = r->dir[0]; x.dy = r->dir[1]; x.dz = r->dir[2];
    x.idx = safe_rcp(x.dx); x.idy = safe_rcp(x.dy); x.idz = safe_rcp(x.dz);
    x.iox = -(x.ox * x.idx); x.ioy = -(x.oy * x.idy); x.ioz = -(x.oz * x.idz);
    x.tmin = r->tmin; x.tmax = r->tmax;
    return x;
}

/* intersection.impala:164-192, no back-face culling.  Returns 1 on hit. */
static inline int intersect_tri(const struct RayX* ray,
                                float v0x, float v0y, float v0z,
                                float e1x, float e1y, float e1z,
                                float e2x, float e2y, float e2z,
                                float nx,  float ny,  float nz,
                                float* t_out, float* u_out, float* v_out) {
    const float cx = v0x - ray->ox, cy = v0y - ray->oy, cz = v0z - ray->oz;
    const float rx = CROSS_X(ray->dx, ray->dy, ray->dz, cx, cy, cz);
    const float ry = CROSS_Y(ray->dx, ray->dy, ray->dz, cx, cy, cz);
    const float rz = CROSS_Z(ray->dx, ray->dy, ray->dz, cx, cy, cz);
    const float det = dot3(nx, ny, nz, ray->dx, ray->dy, ray->dz);
    const float abs_det = fabsf(det);
    const float u = prodsign(dot3(rx, ry, rz, e2x, e2y, e2z), det);
    const float v = prodsign(dot3(rx, ry, rz, e1x, e1y, e1z), det);
    if (!(u >= 0.0f) || !(v >= 0.0f) || !(u + v <= abs_det)) return 0;
    const float t = prodsign(dot3(cx, cy, cz, nx, ny, nz), det);
    if (!(abs_det != 0.0f)) return 0;
    if (!(t >= abs_det * ray->tmin) || !(t <= abs_det * ray->tmax)) return 0;
    const float inv_det = 1.0f / abs_det;
    *t_out = t * inv_det; *u_out = u * inv_det; *v_out = v * inv_det;
    return 1;
}

static inline int intersect_tri1(const struct RayX* ray, const struct Tri1* tr,
                                 float* t, float* u, float* v) {
    /* mapping_gpu.impala:57 -- the normal is computed in the kernel */
    const float nx = CROSS_X(tr->e1[0], tr->e1[1], tr->e1[2], tr->e2[0], tr->e2[1], tr->e2[2]);
    const float ny = CROSS_Y(tr->e1[0], tr->e1[1], tr->e1[2], tr->e2[0], tr->e2[1], tr->e2[2]);
    const float nz = CROSS_Z(tr->e1[0], tr->e1[1], tr->e1[2], tr->e2[0], tr->e2[1], tr->e2[2]);
    return intersect_tri(ray, tr->v0[0], tr->v0[1], tr->v0[2], tr->e1[0], tr->e1[1], tr->e1[2],
                         tr->e2[0], tr->e2[1], tr->e2[2], nx, ny, nz, t, u, v);
}

static inline void stats_merge(struct OracleStats* dst, const struct OracleStats* src) {
    if (!dst) return;
    dst->rays += src->rays; dst->inner_nodes += src->inner_nodes;
    dst->prim_packets += src->prim_packets; dst->hits += src->hits;
    if (src->max_stack > dst->max_stack) dst->max_stack = src->max_stack;
}

/* ------------------------------------------------------------------------- */
/* B1: GPU single-ray traversal of BVH2/Tri1 (mapping_gpu.impala:94-178).     */
/* Min/max are fminf/fmaxf (ocml_fminf/fmaxf, mapping_gpu.impala:87-89).      */
/* ------------------------------------------------------------------------- */
#define STACK_CAP 64   /* stack.impala:53-54 (unchecked there; checked here) */

/* Analysis aid (scripts/model_phases.py): when set, oracle_bvh2_tri1 also writes per ray the number of inner nodes and
 * triangles it visited (2 x uint32 per ray).  Not thread-safe; NULL switches it off. */
static uint32_t* g_ray_steps = 0;
void oracle_set_ray_step_trace(uint32_t* buf) { g_ray_steps = buf; }
/* Analysis aid (scripts/model_top_levels.py): when set, oracle_bvh2_tri1 counts the visits of every inner node
 * (one uint64 per node).  Not thread-safe; NULL switches it off. */
static uint64_t* g_node_visits = 0;
void oracle_set_node_visit_trace(uint64_t* buf) { g_node_visits = buf; }
/* Analysis aid (scripts/model_stack_depth.py): when set, oracle_bvh2_tri1 writes every ray's deepest stack pointer (one byte
 * per ray; the HIP kernels' LDS window of N entries holds rays whose value is below N).  Not thread-safe; NULL switches it off. */
static uint8_t* g_ray_depth = 0;
void oracle_set_ray_depth_trace(uint8_t* buf) { g_ray_depth = buf; }

int oracle_bvh2_tri1(const struct Node2* nodes, const struct Tri1* tris,
                     const struct Ray1* rays, struct Hit1* hits, int32_t n,
                     int32_t any_hit, struct OracleStats* stats_out) {
    struct OracleStats st; memset(&st, 0, sizeof st);
    int overflow = 0;
    for (int32_t i = 0; i < n; i++) {
        struct RayX ray = make_ray(&rays[i]);
        int32_t hit_id = -1; float hit_t = ray.tmax, hit_u = 0.0f, hit_v = 0.0f;
        int32_t mem[STACK_CAP + 2];
        int32_t ptr = 0, top = 1; mem[0] = 0;          /* push(1): old top (0) spilled */
        int done = 0, deepest = 0;
        const uint64_t inner_before = st.inner_nodes, prims_before = st.prim_packets;
        while (top != 0 && !done) {
            const struct Node2* nd = &nodes[top - 1];   /* top is NOT popped (:107-108) */
            st.inner_nodes++;
            if (g_node_visits) g_node_visits[top - 1]++;
            int hk[2]; float te[2];
            for (int k = 0; k < 2; k++) {
                const float* b = nd->bounds + 6 * k;
                const float t0x = fmaf(ray.idx, b[0], ray.iox), t1x = fmaf(ray.idx, b[1], ray.iox);
                const float t0y = fmaf(ray.idy, b[2], ray.ioy), t1y = fmaf(ray.idy, b[3], ray.ioy);
                const float t0z = fmaf(ray.idz, b[4], ray.ioz), t1z = fmaf(ray.idz, b[5], ray.ioz);
                const float tentry = fmax_ref(fmax_ref(fmin_ref(t0x, t1x), fmin_ref(t0y, t1y)),
                                           fmax_ref(fmin_ref(t0z, t1z), ray.tmin));
                const float texit  = fmin_ref(fmin_ref(fmax_ref(t0x, t1x), fmax_ref(t0y, t1y)),
                                           fmin_ref(fmax_ref(t0z, t1z), ray.tmax));
                /* Deviation: an empty slot (child 0, bounds +inf/-inf, converter.cpp:343-350) is never
                 * taken.  The unordered min/max test turns the inverted box into an infinite one, so the
                 * reference would push node id 0 (= "stack empty") and stop early on single-leaf scenes. */
                hk[k] = (tentry <= texit) && nd->child[k] != 0; te[k] = tentry;
            }
            if (!hk[0] && !hk[1]) { top = mem[ptr]; ptr--; }
            else if (hk[0] && hk[1]) {
                const int c0first = te[0] < te[1];      /* strict <, equal -> child1 first (:128-129) */
                const int32_t first  = c0first ? nd->child[0] : nd->child[1];
                const int32_t second = c0first ? nd->child[1] : nd->child[0];
                top = first; ptr++;
                if (ptr >= STACK_CAP) { overflow = 1; ptr = STACK_CAP - 1; }
                mem[ptr] = second;
                if ((uint32_t)ptr > st.max_stack) st.max_stack = (uint32_t)ptr;
                if (ptr > deepest) deepest = ptr;
            } else top = hk[0] ? nd->child[0] : nd->child[1];

            while (top < 0) {                           /* leaf loop (:156-174) */
                int32_t j = ~top; top = mem[ptr]; ptr--;
                for (;;) {
                    const struct Tri1* tr = &tris[j++];
                    st.prim_packets++;
                    float t, u, v;
                    if (intersect_tri1(&ray, tr, &t, &u, &v)) {
                        hit_id = tr->prim_id & 0x7FFFFFFF; hit_t = t; hit_u = u; hit_v = v;
                        ray.tmax = t;
                        if (any_hit) { done = 1; break; }
                    }
                    if (tr->prim_id < 0) break;         /* sentinel bit 31 (:63,172) */
                }
                if (done) break;
            }
        }
        hits[i].tri_id = hit_id; hits[i].t = hit_t; hits[i].u = hit_u; hits[i].v = hit_v;
        st.hits += hit_id >= 0;
        if (g_ray_steps) { g_ray_steps[2 * i] = (uint32_t)(st.inner_nodes - inner_before);
            g_ray_steps[2 * i + 1] = (uint32_t)(st.prim_packets - prims_before); }
        if (g_ray_depth) g_ray_depth[i] = (uint8_t)deepest;
    }
    st.rays = (uint64_t)n;
    stats_merge(stats_out, &st);
    return overflow ? -1 : 0;
}

/* ------------------------------------------------------------------------- */
/* B2: CPU single-ray traversal of BVH4/BVH8 with Tri4 leaves                 */
/* (mapping_cpu.impala:138-256).  Min/max are the integer-compare variants    */
/* (bench_traversal.impala:11 enable_cpu_int_min_max, mapping_cpu.impala:123-133). */
/* ------------------------------------------------------------------------- */
static inline float imin(float x, float y) { return f2i(x) < f2i(y) ? x : y; }
static inline float imax(float x, float y) { return f2i(x) > f2i(y) ? x : y; }

struct Ent { int32_t node; float tmin; };

/* sort.impala:39-66 (Batcher odd-even merge sort, comparators on missing elements removed) */
static void batcher_merge(struct Ent* a, int n, int i, int len, int r) {
    const int step = r * 2;
    if (step < len) {
        batcher_merge(a, n, i, len, step);
        batcher_merge(a, n, i + r, len, step);
        for (int j = i + r; j < i + len - r; j += step)
            if (j < n && j + r < n && a[j].tmin < a[j + r].tmin) { struct Ent t = a[j]; a[j] = a[j + r]; a[j + r] = t; }
    } else if (i < n && i + r < n && a[i].tmin < a[i + r].tmin) { struct Ent t = a[i]; a[i] = a[i + r]; a[i + r] = t; }
}
static void batcher_sort_rec(struct Ent* a, int n, int i, int len) {
    if (len > 1) { const int m = len / 2; batcher_sort_rec(a, n, i, m); batcher_sort_rec(a, n, i + m, m); batcher_merge(a, n, i, len, 1); }
}
static void batcher_sort(struct Ent* a, int n) {
    int p = 0; while (n > (1 << p)) p++;               /* common.impala:107-116 ilog2 = ceil(log2) */
    batcher_sort_rec(a, n, 0, 1 << p);
}
/* sort.impala:3-37 (Bose-Nelson) */
#define CMPSWAP(a, i, j) do { if ((a)[i].tmin < (a)[j].tmin) { struct Ent t_ = (a)[i]; (a)[i] = (a)[j]; (a)[j] = t_; } } while (0)
static void bn_bracket(struct Ent* a, int i1, int len1, int i2, int len2) {
    if (len1 == 1 && len2 == 1) CMPSWAP(a, i1, i2);
    else if (len1 == 1 && len2 == 2) { CMPSWAP(a, i1, i2 + 1); CMPSWAP(a, i1, i2); }
    else if (len1 == 2 && len2 == 1) { CMPSWAP(a, i1, i2); CMPSWAP(a, i1 + 1, i2); }
    else {
        const int x = len1 / 2;
        const int y = (len1 % 2 != 0) ? len2 / 2 : (len2 + 1) / 2;
        bn_bracket(a, i1, x, i2, y);
        bn_bracket(a, i1 + x, len1 - x, i2 + y, len2 - y);
        bn_bracket(a, i1 + x, len1 - x, i2, y);
    }
}
static void bn_star(struct Ent* a, int i, int len) {
    if (len > 1) { const int m = len / 2; bn_star(a, i, m); bn_star(a, i + m, len - m); bn_bracket(a, i, m, i + m, len - m); }
}

#define STACK_CAPN 128

static int traverse_single_wide(const float* node_base, int arity, const struct Tri4* tris,
                                const struct Ray1* rays, struct Hit1* hits, int32_t n,
                                int32_t any_hit, int32_t root, struct OracleStats* stats_out) {
    struct OracleStats st; memset(&st, 0, sizeof st);
    const int node_floats = 8 * arity;                  /* 6 bound rows + child + pad, each `arity` wide */
    int overflow = 0;
    for (int32_t i = 0; i < n; i++) {
        struct RayX ray = make_ray(&rays[i]);
        /* intersection.impala:128-132 */
        const int ox = ray.dx > 0.0f, oy = ray.dy > 0.0f, oz = ray.dz > 0.0f;
        int32_t hit_id = -1; float hit_t = ray.tmax, hit_u = 0.0f, hit_v = 0.0f;
        struct Ent mem[STACK_CAPN + 16];
        int32_t ptr = -1;
        struct Ent top = { 0, FLT_MAX_REF };
        /* push(root, tmin) */
        mem[++ptr] = top; top.node = root; top.tmin = ray.tmin;
        for (;;) {
            if (top.node == 0) break;
            if (!any_hit && top.tmin > ray.tmax) { top = mem[ptr--]; continue; }   /* :171-174 */
            int restart = 0;
            while (top.node > 0) {
                const float* nd = node_base + (size_t)(top.node - 1) * node_floats;
                const int32_t* child = (const int32_t*)(nd + 6 * arity);
                top = mem[ptr--];                        /* pop */
                st.inner_nodes++;
                /* ordered boxes, mapping_cpu.impala:51-69,88-106: near plane = lo if dir>0 else hi */
                const float* nx_ = nd + (ox ? 0 : 1) * arity; const float* fx_ = nd + (ox ? 1 : 0) * arity;
                const float* ny_ = nd + (oy ? 2 : 3) * arity; const float* fy_ = nd + (oy ? 3 : 2) * arity;
                const float* nz_ = nd + (oz ? 4 : 5) * arity; const float* fz_ = nd + (oz ? 5 : 4) * arity;
                float tentry[8]; unsigned mask = 0;
                for (int k = 0; k < arity; k++) {
                    const float t0x = fmaf(ray.idx, nx_[k], ray.iox), t1x = fmaf(ray.idx, fx_[k], ray.iox);
                    const float t0y = fmaf(ray.idy, ny_[k], ray.ioy), t1y = fmaf(ray.idy, fy_[k], ray.ioy);
                    const float t0z = fmaf(ray.idz, nz_[k], ray.ioz), t1z = fmaf(ray.idz, fz_[k], ray.ioz);
                    const float te = imax(imax(t0x, t0y), imax(t0z, ray.tmin));
                    const float tx = imin(imin(t1x, t1y), imin(t1z, ray.tmax));
                    tentry[k] = te;
                    if (!(f2i(tx) < f2i(te))) mask |= 1u << k;       /* :182-187 */
                }
                if (mask == 0) { if (any_hit) continue; restart = 1; break; }
                int num = 0;
                for (int k = 0; k < arity; k++) {
                    if (!(mask & (1u << k))) continue;
                    num++;
                    const float t = tentry[k];
                    if (ptr + 1 >= STACK_CAPN) { overflow = 1; break; }
                    if (any_hit || t < top.tmin) { mem[++ptr] = top; top.node = child[k]; top.tmin = t; }
                    else { ++ptr; mem[ptr].node = child[k]; mem[ptr].tmin = t; }
                }
                if ((uint32_t)(ptr + 1) > st.max_stack) st.max_stack = (uint32_t)(ptr + 1);
                if (!any_hit && num >= 3) {             /* :210-218 */
                    struct Ent* a = &mem[ptr - num + 1];
                    if (arity == 8) batcher_sort(a, num); else bn_star(a, 0, num);
                }
            }
            if (restart) continue;
            if (any_hit && top.node == 0) break;
            /* leaf */
            int32_t j = ~top.node; top = mem[ptr--];
            int terminated = 0;
            for (;;) {
                const struct Tri4* P = &tris[j++];
                st.prim_packets++;
                float bt = 0, bu = 0, bv = 0; int bl = -1;
                for (int k = 0; k < 4; k++) {
                    if (P->prim_id[k] == -1) continue;   /* is_valid (:38) */
                    float t, u, v;
                    if (!intersect_tri(&ray, P->v0[0][k], P->v0[1][k], P->v0[2][k],
                                       P->e1[0][k], P->e1[1][k], P->e1[2][k],
                                       P->e2[0][k], P->e2[1][k], P->e2[2][k],
                                       P->n[0][k],  P->n[1][k],  P->n[2][k], &t, &u, &v)) continue;
                    if (any_hit) { if (bl < 0) { bl = k; bt = t; bu = u; bv = v; } }   /* first lane (:233-237) */
                    else if (bl < 0 || f2i(t) < f2i(bt)) { bl = k; bt = t; bu = u; bv = v; } /* min t, lowest lane on ties (:239-243) */
                }
                if (bl >= 0) {
                    hit_id = P->prim_id[bl] & 0x7FFFFFFF; hit_t = bt; hit_u = bu; hit_v = bv;
                    if (any_hit) terminated = 1; else ray.tmax = bt;
                }
                if (P->prim_id[3] < 0) break;            /* is_last (:39,248) */
            }
            if (any_hit && terminated) break;
        }
        hits[i].tri_id = hit_id; hits[i].t = hit_t; hits[i].u = hit_u; hits[i].v = hit_v;
        st.hits += hit_id >= 0;
    }
    st.rays = (uint64_t)n;
    stats_merge(stats_out, &st);
    return overflow ? -1 : 0;
}

int oracle_bvh8_tri4(const struct Node8* nodes, const struct Tri4* tris, const struct Ray1* rays,
                     struct Hit1* hits, int32_t n, int32_t any_hit, struct OracleStats* st) {
    return traverse_single_wide((const float*)nodes, 8, tris, rays, hits, n, any_hit, 1, st);
}
int oracle_bvh4_tri4(const struct Node4* nodes, const struct Tri4* tris, const struct Ray1* rays,
                     struct Hit1* hits, int32_t n, int32_t any_hit, struct OracleStats* st) {
    return traverse_single_wide((const float*)nodes, 4, tris, rays, hits, n, any_hit, 1, st);
}

/* ------------------------------------------------------------------------- */
/* B1g: the reference GPU kernel's "general case" for arity != 2              */
/* (mapping_gpu.impala:136-153) on the CPU layouts Node4/Node8 + Tri4:         */
/* pop the node; unordered box test of every child with fminf/fmaxf; the        */
/* nearest hit child (strict <, starting from ray.tmax) is pushed on top, the   */
/* others go underneath in slot order; nothing is culled on pop.  Triangles of  */
/* a packet are tested one after the other, tmax shrinking in between            */
/* (mapping_gpu.impala:160-169).  This is what k_bvh8_lane implements.          */
/* ------------------------------------------------------------------------- */
static int traverse_gpu_wide(const float* node_base, int arity, const struct Tri4* tris,
                             const struct Ray1* rays, struct Hit1* hits, int32_t n,
                             int32_t any_hit, struct OracleStats* stats_out) {
    struct OracleStats st; memset(&st, 0, sizeof st);
    const int node_floats = 8 * arity;
    int overflow = 0;
    for (int32_t i = 0; i < n; i++) {
        struct RayX ray = make_ray(&rays[i]);
        int32_t hit_id = -1; float hit_t = ray.tmax, hit_u = 0.0f, hit_v = 0.0f;
        int32_t mem[STACK_CAP + 16];
        int32_t ptr = 0, top = 1; mem[0] = 0;
        int done = 0;
        while (top != 0 && !done) {
            const float* nd = node_base + (size_t)(top - 1) * node_floats;
            const int32_t* child = (const int32_t*)(nd + 6 * arity);
            top = mem[ptr]; ptr--;                                     /* pop (:138) */
            st.inner_nodes++;
            float tnear = ray.tmax;
            for (int k = 0; k < arity; k++) {
                const float t0x = fmaf(ray.idx, nd[0 * arity + k], ray.iox), t1x = fmaf(ray.idx, nd[1 * arity + k], ray.iox);
                const float t0y = fmaf(ray.idy, nd[2 * arity + k], ray.ioy), t1y = fmaf(ray.idy, nd[3 * arity + k], ray.ioy);
                const float t0z = fmaf(ray.idz, nd[4 * arity + k], ray.ioz), t1z = fmaf(ray.idz, nd[5 * arity + k], ray.ioz);
                const float tentry = fmax_ref(fmax_ref(fmin_ref(t0x, t1x), fmin_ref(t0y, t1y)), fmax_ref(fmin_ref(t0z, t1z), ray.tmin));
                const float texit  = fmin_ref(fmin_ref(fmax_ref(t0x, t1x), fmax_ref(t0y, t1y)), fmin_ref(fmax_ref(t0z, t1z), ray.tmax));
                if (!(tentry <= texit) || child[k] == 0) continue;          /* empty slots: see B1 */
                if (ptr + 1 >= STACK_CAP) { overflow = 1; continue; }
                if (any_hit || tentry < tnear) { mem[++ptr] = top; top = child[k]; tnear = tentry; }
                else mem[++ptr] = child[k];
            }
            if ((uint32_t)(ptr + 1) > st.max_stack) st.max_stack = (uint32_t)(ptr + 1);
            while (top < 0) {
                int32_t j = ~top; top = mem[ptr]; ptr--;
                for (;;) {
                    const struct Tri4* P = &tris[j++];
                    st.prim_packets++;
                    for (int k = 0; k < 4; k++) {
                        if (P->prim_id[k] == -1) break;
                        float t, u, v;
                        if (!intersect_tri(&ray, P->v0[0][k], P->v0[1][k], P->v0[2][k],
                                           P->e1[0][k], P->e1[1][k], P->e1[2][k],
                                           P->e2[0][k], P->e2[1][k], P->e2[2][k],
                                           P->n[0][k],  P->n[1][k],  P->n[2][k], &t, &u, &v)) continue;
                        hit_id = P->prim_id[k] & 0x7FFFFFFF; hit_t = t; hit_u = u; hit_v = v;
                        ray.tmax = t;
                        if (any_hit) { done = 1; break; }
                    }
                    if (done || P->prim_id[3] < 0) break;
                }
                if (done) break;
            }
        }
        hits[i].tri_id = hit_id; hits[i].t = hit_t; hits[i].u = hit_u; hits[i].v = hit_v;
        st.hits += hit_id >= 0;
    }
    st.rays = (uint64_t)n;
    stats_merge(stats_out, &st);
    return overflow ? -1 : 0;
}

int oracle_gpu_bvh8_tri4(const struct Node8* nodes, const struct Tri4* tris, const struct Ray1* rays,
                         struct Hit1* hits, int32_t n, int32_t any_hit, struct OracleStats* st) {
    return traverse_gpu_wide((const float*)nodes, 8, tris, rays, hits, n, any_hit, st);
}
int oracle_gpu_bvh4_tri4(const struct Node4* nodes, const struct Tri4* tris, const struct Ray1* rays,
                         struct Hit1* hits, int32_t n, int32_t any_hit, struct OracleStats* st) {
    return traverse_gpu_wide((const float*)nodes, 4, tris, rays, hits, n, any_hit, st);
}

/* ------------------------------------------------------------------------- */
/* Exhaustive checker: every ray against every triangle, no hierarchy.        */
/* closest: best = lexicographic min of (t, prim id) over all accepted        */
/* triangles using the ray's ORIGINAL [tmin,tmax]; second_t[i] = smallest t    */
/* among accepted triangles whose prim id differs from the winner (tmax if     */
/* none) -- tests use it to recognise rays whose winner is order-dependent.    */
/* ------------------------------------------------------------------------- */
void oracle_brute_force_tri1(const struct Tri1* tris, int32_t num_tris,
                             const struct Ray1* rays, struct Hit1* hits, float* second_t, int32_t n) {
    for (int32_t i = 0; i < n; i++) {
        const struct RayX ray = make_ray(&rays[i]);
        int32_t best = -1; float bt = ray.tmax, bu = 0, bv = 0, st2 = ray.tmax;
        for (int32_t j = 0; j < num_tris; j++) {
            float t, u, v;
            if (!intersect_tri1(&ray, &tris[j], &t, &u, &v)) continue;
            const int32_t id = tris[j].prim_id & 0x7FFFFFFF;
            if (best < 0 || t < bt || (t == bt && id < best)) {
                if (best >= 0 && id != best && bt < st2) st2 = bt;
                best = id; bt = t; bu = u; bv = v;
            } else if (id != best && t < st2) st2 = t;
        }
        hits[i].tri_id = best; hits[i].t = bt; hits[i].u = bu; hits[i].v = bv;
        if (second_t) second_t[i] = st2;
    }
}

void oracle_brute_force_tri4(const struct Tri4* tris, int32_t num_packets,
                             const struct Ray1* rays, struct Hit1* hits, float* second_t, int32_t n) {
    for (int32_t i = 0; i < n; i++) {
        const struct RayX ray = make_ray(&rays[i]);
        int32_t best = -1; float bt = ray.tmax, bu = 0, bv = 0, st2 = ray.tmax;
        for (int32_t j = 0; j < num_packets; j++) {
            const struct Tri4* P = &tris[j];
            for (int k = 0; k < 4; k++) {
                if (P->prim_id[k] == -1) continue;
                float t, u, v;
                if (!intersect_tri(&ray, P->v0[0][k], P->v0[1][k], P->v0[2][k],
                                   P->e1[0][k], P->e1[1][k], P->e1[2][k],
                                   P->e2[0][k], P->e2[1][k], P->e2[2][k],
                                   P->n[0][k],  P->n[1][k],  P->n[2][k], &t, &u, &v)) continue;
                const int32_t id = P->prim_id[k] & 0x7FFFFFFF;
                if (best < 0 || t < bt || (t == bt && id < best)) {
                    if (best >= 0 && id != best && bt < st2) st2 = bt;
                    best = id; bt = t; bu = u; bv = v;
                } else if (id != best && t < st2) st2 = t;
            }
        }
        hits[i].tri_id = best; hits[i].t = bt; hits[i].u = bu; hits[i].v = bv;
        if (second_t) second_t[i] = st2;
    }
}

uint32_t oracle_abi_sizes(int which) {
    switch (which) {
        case 0: return sizeof(struct Node2); case 1: return sizeof(struct Tri1);
        case 2: return sizeof(struct Node4); case 3: return sizeof(struct Node8);
        case 4: return sizeof(struct Tri4);  case 5: return sizeof(struct Ray1);
        case 6: return sizeof(struct Hit1);  case 7: return sizeof(struct OracleStats);
    }
    return 0;
}
