/*
 * render_oracle.c -- CPU restatement of Rodent's path tracer.  TEST INFRASTRUCTURE.
 *
 * One path at a time (emit -> closest-hit traverse -> shade -> shadow any-hit -> bounce),
 * which is what the reference's wavefront / megakernel mappings compute per path; the
 * order in which paths are interleaved does not change a path's arithmetic because the
 * RNG state and throughput live in the path (src/render/mapping_gpu.impala:82-134,371-474).
 * Only tests/, __graft_entry__.smoke() and bench.py may use this file.
 *
 * PARITY PINNING: the reference cannot be compiled here (Impala / AnyDSL absent), so this
 * restatement is pinned by the reference's own golden image testing/ref-cornell.png
 * (src/CMakeLists.txt:131-134: cornell_box.obj, 1080x720, eye 0 1 2.7, dir 0 0 -1, up 0 1 0)
 * through an MSE check in tests/, plus energy / pdf-normalisation property tests.
 *
 * Arithmetic: fp32, plain IEEE operations in source order (-ffp-contract=off), division and
 * sqrt correctly rounded.  sin/cos of the sampling routines go through sincos_2pi() below, a
 * fixed polynomial, instead of the platform's cosf/sinf (the reference takes them from the
 * AnyDSL runtime, unpinned -- SURVEY.md 8c), so that the HIP kernels can reproduce every path
 * bit for bit.
 *
 * Reference lines restated:
 *   RNG, hashes, samplers, RR        src/core/random.impala:7-11,22-30,49-131
 *   fastlog2 / fastpow2 / fastpow    src/core/common.impala:42-61
 *   orthonormal basis                src/core/matrix.impala:29-39
 *   colour helpers                   src/core/color.impala:19-35
 *   camera                           src/render/camera.impala:29-44 (w, h: src/driver/driver.cpp:37-38)
 *   emitter / on_hit / on_shadow / on_bounce   src/render/renderer.impala:26-40,62-162
 *   surface element                  src/render/geometry.impala:21-54
 *   BSDFs                            src/render/material.impala:63-192
 *   triangle light                   src/render/light.impala:46-102,122-154
 *   material selection from MTL      src/driver/converter.cpp:858-920 (done by the scene loader)
 *   film accumulation                src/render/mapping_gpu.impala:32-45
 *   traversal                        traversal_oracle.c (B1: mapping_gpu.impala:94-178)
 */
#include <stdint.h>
#include <string.h>
#include <math.h>

struct Node2 { float bounds[12]; int32_t child[2]; int32_t pad[2]; };
struct Tri1  { float v0[3]; int32_t pad; float e1[3]; int32_t geom_id; float e2[3]; int32_t prim_id; };
struct Ray1  { float org[3]; float tmin; float dir[3]; float tmax; };
struct Hit1  { int32_t tri_id; float t, u, v; };
struct OracleStats;
int oracle_bvh2_tri1(const struct Node2*, const struct Tri1*, const struct Ray1*, struct Hit1*, int32_t, int32_t, struct OracleStats*);

/* Scene tables (same layout as include/rodent_render.h) */
enum { MAT_BLACK = 0, MAT_DIFFUSE = 1, MAT_PHONG = 2, MAT_MIX = 3, MAT_MIRROR = 4, MAT_GLASS = 5 };
struct Material { float kd[3]; int32_t type; float ks[3]; float ns; float tf[3]; float ni; float mix_k; int32_t emissive;
    int32_t tex_kd, tex_ks; };
struct Texture  { int32_t width, height; uint32_t offset; int32_t pad; };
struct Light    { float v0[4], v1[4], v2[4]; float n[3]; float inv_area; float color[4]; };
struct Scene {
    const float* vertices;      /* float4 per vertex */
    const float* normals;       /* float4 per vertex */
    const float* face_normals;  /* float4 per triangle */
    const int32_t* indices;     /* int4 per triangle: v0 v1 v2 material */
    const struct Node2* nodes; const struct Tri1* tris;
    const struct Material* materials; const struct Light* lights; const int32_t* light_ids;
    int32_t num_tris, num_materials, num_lights, pad;
    const float* texcoords;     /* float4 per vertex: u, v, 0, 0 */
    const struct Texture* textures; const uint32_t* texels;
};
struct Settings { float eye[3], dir[3], up[3], right[3]; float w, h; };

#define FLT_MAX_REF 3.4028234664e+38f
#define FLT_PI 3.14159265359f

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 mulf(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              /* vector.impala:60 */
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float len(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { return mulf(a, 1.0f / len(a)); }                           /* vector.impala:82 */
static inline v3 reflect(v3 v, v3 n) { return sub(mulf(n, 2.0f * dot(n, v)), v); }            /* vector.impala:74 */
static inline float lerp1(float a, float b, float k) { return (1.0f - k) * a + k * b; }        /* common.impala:118 */
static inline float lerp2(float a, float b, float c, float k1, float k2) { return (1.0f - k1 - k2) * a + k1 * b + k2 * c; }
static inline float positive_cos(v3 a, v3 b) { const float c = dot(a, b); return c >= 0.0f ? c : 0.0f; }
static inline float luminance(v3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }  /* color.impala:33 */
static inline v3 color_lerp(v3 a, v3 b, float t) {
    return V((1.0f - t) * a.x + t * b.x, (1.0f - t) * a.y + t * b.y, (1.0f - t) * a.z + t * b.z); }

static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* random.impala:22-30, 7-11 */
static inline uint32_t xorshift(uint32_t* seed) { uint32_t x = *seed; x = x == 0u ? 1u : x; x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *seed = x; return x; }
static inline float randf(uint32_t* rnd) { return u2f((127u << 23) | (xorshift(rnd) & 0x7FFFFFu)) - 1.0f; }
/* random.impala:116-126 */
static inline uint32_t fnv_hash(uint32_t h, uint32_t d) {
    h = (h * 16777619u) ^ (d & 0xFFu); h = (h * 16777619u) ^ ((d >> 8) & 0xFFu);
    h = (h * 16777619u) ^ ((d >> 16) & 0xFFu); h = (h * 16777619u) ^ ((d >> 24) & 0xFFu); return h;
}

/* common.impala:42-61 */
static inline float fastlog2(float x) {
    const uint32_t vx = f2u(x); const uint32_t mx = (vx & 0x007FFFFFu) | 0x3f000000u;
    const float y = (float)vx * 1.1920928955078125e-7f; const float z = u2f(mx);
    return y - 124.22551499f - 1.498030302f * z - 1.72587999f / (0.3520887068f + z);
}
static inline float fastpow2(float p) {
    const float offset = p < 0.0f ? 1.0f : 0.0f; const float clipp = p < -126.0f ? -126.0f : p;
    const int32_t w = (int32_t)clipp; const float z = clipp - (float)w + offset;
    const int32_t v = (int32_t)((float)(1u << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z));
    return u2f((uint32_t)v);
}
static inline float fastpow(float x, float p) { return fastpow2(p * fastlog2(x)); }

/* cos(2 pi u), sin(2 pi u) for u in [0,1): quadrant reduction + fixed odd/even polynomials on
 * [-pi/4, pi/4] (Taylor coefficients, |error| < 1e-7); every operation is a plain fp32 op. */
static inline void sincos_2pi(float u, float* c_out, float* s_out) {
    const float x = u * 4.0f;                       /* in quarter turns */
    const int32_t k = (int32_t)(x + 0.5f);          /* nearest quadrant 0..4 */
    const float a = (x - (float)k) * 1.57079632679f;
    const float a2 = a * a;
    const float s = a * (1.0f + a2 * (-0.16666667163f + a2 * (0.0083333337680f + a2 * (-0.00019841270114f + a2 * 2.7557314297e-6f))));
    const float c = 1.0f + a2
        * (-0.5f + a2 * (0.041666667908f + a2 * (-0.0013888889225f + a2 * (2.4801587642e-5f + a2 * -2.7557314297e-7f))));
    switch (k & 3) {
        case 0: *c_out = c;  *s_out = s;  break;
        case 1: *c_out = -s; *s_out = c;  break;
        case 2: *c_out = -c; *s_out = -s; break;
        default: *c_out = s; *s_out = -c; break;
    }
}

typedef struct { v3 c0, c1, c2; } m3;
/* matrix.impala:29-39 */
static inline m3 orthonormal(v3 n) {
    const float sign = n.z >= 0.0f ? 1.0f : -1.0f; const float a = -1.0f / (sign + n.z); const float b = n.x * n.y * a;
    m3 m; m.c0 = V(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x); m.c1 = V(b, sign + n.y * n.y * a, -n.y); m.c2 = n; return m;
}
static inline v3 m3_mul(m3 m, v3 v) {   /* matrix.impala mat3x3_mul: rows dotted with v */
    return V(m.c0.x * v.x + m.c1.x * v.y + m.c2.x * v.z, m.c0.y * v.x + m.c1.y * v.y + m.c2.y * v.z,
        m.c0.z * v.x + m.c1.z * v.y + m.c2.z * v.z);
}

typedef struct { v3 dir; float pdf; } DirSample;
static inline DirSample make_dir_sample(float c, float s, float u, float pdf) {      /* random.impala:38-47, phi = 2 pi u */
    float cp, sp; sincos_2pi(u, &cp, &sp); DirSample d; d.dir = V(s * cp, s * sp, c); d.pdf = pdf; return d;
}
static inline float cosine_hemisphere_pdf(float c) { return c * (1.0f / FLT_PI); }
static inline DirSample sample_cosine_hemisphere(float u, float v) {                  /* random.impala:72-79 */
    const float c = sqrtf(1.0f - v), s = sqrtf(v); return make_dir_sample(c, s, u, cosine_hemisphere_pdf(c));
}
static inline float cosine_power_hemisphere_pdf(float c, float k) { return fastpow(c, k) * (k + 1.0f) * (1.0f / (2.0f * FLT_PI)); }
static inline DirSample sample_cosine_power_hemisphere(float k, float u, float v) {    /* random.impala:87-101 */
    const float p = fastpow(v, 1.0f / (k + 1.0f)); const float c = p < 1.0f ? p : 1.0f;
    const float s = sqrtf(1.0f - c * c); const float pow_c_k = c != 0.0f ? v / c : 0.0f;
    return make_dir_sample(c, s, u, pow_c_k * (k + 1.0f) * (1.0f / (2.0f * FLT_PI)));
}

typedef struct { int entering; v3 point, face_normal; m3 local; } Surf;
typedef struct { v3 in_dir; float pdf, cos; v3 color; } BsdfSample;

/* material.impala:63-72 */
static inline BsdfSample make_bsdf_sample(const Surf* s, v3 in_dir, float pdf, float cosv, v3 color, int inverted) {
    const int valid = (pdf > 0.0f) && (inverted ^ (dot(in_dir, s->face_normal) > 0.0f));
    BsdfSample r; r.in_dir = in_dir; r.pdf = valid ? pdf : 1.0f; r.cos = cosv; r.color = valid ? color : V(0, 0, 0); return r;
}
static inline v3 LD3(const float* p) { return V(p[0], p[1], p[2]); }

/* diffuse: material.impala:85-100; phong: :103-123 */
static inline v3 diffuse_eval(const struct Material* m) { return mulf(LD3(m->kd), 1.0f / FLT_PI); }
static inline float diffuse_pdf(const Surf* s, v3 in_dir) { return cosine_hemisphere_pdf(positive_cos(in_dir, s->local.c2)); }
static inline BsdfSample diffuse_sample(const struct Material* m, const Surf* s, uint32_t* rnd) {
    const float u = randf(rnd), v = randf(rnd); const DirSample d = sample_cosine_hemisphere(u, v);
    return make_bsdf_sample(s, m3_mul(s->local, d.dir), d.pdf, d.dir.z, mulf(LD3(m->kd), 1.0f / FLT_PI), 0);
}
static inline v3 phong_eval(const struct Material* m, const Surf* s, v3 in_dir, v3 out_dir) {
    const float c = positive_cos(in_dir, reflect(out_dir, s->local.c2));
    return mulf(LD3(m->ks), fastpow(c, m->ns) * (m->ns + 2.0f) * (1.0f / (2.0f * FLT_PI)));
}
static inline float phong_pdf(const struct Material* m, const Surf* s, v3 in_dir, v3 out_dir) {
    return cosine_power_hemisphere_pdf(positive_cos(in_dir, reflect(out_dir, s->local.c2)), m->ns);
}
static inline BsdfSample phong_sample(const struct Material* m, const Surf* s, uint32_t* rnd, v3 out_dir) {
    const v3 r = reflect(out_dir, s->local.c2);
    const float u = randf(rnd), v = randf(rnd); const DirSample d = sample_cosine_power_hemisphere(m->ns, u, v);
    const v3 in_dir = m3_mul(orthonormal(r), d.dir); const float c = positive_cos(in_dir, s->local.c2);
    return make_bsdf_sample(s, in_dir, d.pdf, c, mulf(LD3(m->ks), d.pdf * (m->ns + 2.0f) / (m->ns + 1.0f)), 0);
}
static inline float fresnel_factor(float k, float ci, float ct) {                      /* material.impala:39-43 */
    const float rs = (k * ci - ct) / (k * ci + ct), rp = (ci - k * ct) / (ci + k * ct); return (rs * rs + rp * rp) * 0.5f;
}

static inline int bsdf_is_specular(const struct Material* m) { return m->type == MAT_MIRROR || m->type == MAT_GLASS; }
static v3 bsdf_eval(const struct Material* m, const Surf* s, v3 in_dir, v3 out_dir) {
    switch (m->type) {
        case MAT_DIFFUSE: return diffuse_eval(m);
        case MAT_PHONG:   return phong_eval(m, s, in_dir, out_dir);
        case MAT_MIX:     return color_lerp(diffuse_eval(m), phong_eval(m, s, in_dir, out_dir), m->mix_k);   /* :166-171 */
        default:          return V(0, 0, 0);
    }
}
static float bsdf_pdf(const struct Material* m, const Surf* s, v3 in_dir, v3 out_dir) {
    switch (m->type) {
        case MAT_DIFFUSE: return diffuse_pdf(s, in_dir);
        case MAT_PHONG:   return phong_pdf(m, s, in_dir, out_dir);
        case MAT_MIX:     return lerp1(diffuse_pdf(s, in_dir), phong_pdf(m, s, in_dir, out_dir), m->mix_k);
        default:          return 0.0f;
    }
}
static BsdfSample bsdf_sample(const struct Material* m, const Surf* s, uint32_t* rnd, v3 out_dir) {
    switch (m->type) {
        case MAT_DIFFUSE: return diffuse_sample(m, s, rnd);
        case MAT_PHONG:   return phong_sample(m, s, rnd, out_dir);
        case MAT_MIX: {                                                                /* material.impala:176-189 */
            BsdfSample r;
            if (randf(rnd) >= m->mix_k) {
                r = diffuse_sample(m, s, rnd);
                const float p = lerp1(r.pdf, phong_pdf(m, s, r.in_dir, out_dir), m->mix_k);
                r.color = color_lerp(r.color, phong_eval(m, s, r.in_dir, out_dir), m->mix_k); r.pdf = p;
            } else {
                r = phong_sample(m, s, rnd, out_dir);
                const float p = lerp1(diffuse_pdf(s, r.in_dir), r.pdf, m->mix_k);
                r.color = color_lerp(diffuse_eval(m), r.color, m->mix_k); r.pdf = p;
            }
            return r;
        }
        case MAT_MIRROR: return make_bsdf_sample(s, reflect(out_dir, s->local.c2), 1.0f, 1.0f, LD3(m->ks), 0);   /* :126-135 */
        case MAT_GLASS: {                                                              /* :138-163, n1 = 1, n2 = Ni, not adjoint */
            const float k = s->entering ? 1.0f / m->ni : m->ni / 1.0f;
            const v3 n = s->local.c2; const float ci = dot(out_dir, n); const float c2t = 1.0f - k * k * (1.0f - ci * ci);
            if (c2t > 0.0f) {
                const float ct = sqrtf(c2t); const float F = fresnel_factor(k, ci, ct);
                if (randf(rnd) > F) {
                    const v3 t = sub(mulf(n, k * ci - ct), mulf(out_dir, k));
                    return make_bsdf_sample(s, t, 1.0f, 1.0f, mulf(LD3(m->tf), 1.0f), 1);
                }
            }
            return make_bsdf_sample(s, reflect(out_dir, n), 1.0f, 1.0f, LD3(m->ks), 0);
        }
        default: { BsdfSample r; r.in_dir = out_dir; r.pdf = 1.0f; r.cos = 1.0f; r.color = V(0, 0, 0); return r; }   /* black :75-82 */
    }
}

/* geometry.impala:21-54 */
static Surf surface_element(const struct Scene* sc, v3 org, v3 dir, int32_t prim, float t, float u, float v) {
    const int32_t* idx = sc->indices + 4 * prim;
    const v3 fn = LD3(sc->face_normals + 4 * prim);
    const float* n0 = sc->normals + 4 * idx[0]; const float* n1 = sc->normals + 4 * idx[1]; const float* n2 = sc->normals + 4 * idx[2];
    const v3 nrm = normalize(V(lerp2(n0[0], n1[0], n2[0], u, v), lerp2(n0[1], n1[1], n2[1], u, v), lerp2(n0[2], n1[2], n2[2], u, v)));
    Surf s; s.entering = dot(dir, fn) <= 0.0f; s.point = add(org, mulf(dir, t));
    s.face_normal = s.entering ? fn : neg(fn); s.local = orthonormal(dot(dir, nrm) <= 0.0f ? nrm : neg(nrm)); return s;
}

/* image.impala:24-38 (RGBA8 -> colour), :48-54 (repeat border), :64-86 (bilinear filter) */
static inline v3 texel(const struct Scene* sc, const struct Texture* t, int32_t x, int32_t y) {
    const uint32_t p = sc->texels[t->offset + (uint32_t)y * (uint32_t)t->width + (uint32_t)x];
    return V((float)(p & 0xFFu) * (1.0f / 255.0f), (float)((p >> 8) & 0xFFu) * (1.0f / 255.0f),
        (float)((p >> 16) & 0xFFu) * (1.0f / 255.0f));
}
static v3 tex_lookup(const struct Scene* sc, const struct Texture* t, float tu, float tv) {
    const float ru = tu - floorf(tu), rv = tv - floorf(tv);
    const float u = ru * (float)t->width, v = rv * (float)t->height;
    const int32_t iu = (int32_t)u, iv = (int32_t)v;
    const int32_t x0 = iu < t->width - 1 ? iu : t->width - 1, y0 = iv < t->height - 1 ? iv : t->height - 1;
    const int32_t x1 = x0 + 1 < t->width - 1 ? x0 + 1 : t->width - 1, y1 = y0 + 1 < t->height - 1 ? y0 + 1 : t->height - 1;
    const float kx = u - (float)iu, ky = v - (float)iv;
    const v3 p00 = texel(sc, t, x0, y0), p10 = texel(sc, t, x1, y0), p01 = texel(sc, t, x0, y1), p11 = texel(sc, t, x1, y1);
    return V(lerp1(lerp1(p00.x, p10.x, kx), lerp1(p01.x, p11.x, kx), ky), lerp1(lerp1(p00.y, p10.y, kx), lerp1(p01.y, p11.y, kx), ky),
             lerp1(lerp1(p00.z, p10.z, kx), lerp1(p01.z, p11.z, kx), ky));
}
/* The material of a hit: map_Kd / map_Ks replace kd / ks with texture lookups at the interpolated texture coordinates,
 * and the diffuse/Phong mix weight follows the looked-up colours (converter.cpp:881-906, geometry.impala:30-40). */
static const struct Material* resolve_material(const struct Scene* sc, const struct Material* m, struct Material* tmp, int32_t prim,
    float u, float v) {
    if (!(m->tex_kd | m->tex_ks)) return m;
    const int32_t* idx = sc->indices + 4 * prim;
    const float* t0 = sc->texcoords + 4 * idx[0]; const float* t1 = sc->texcoords + 4 * idx[1];
    const float* t2 = sc->texcoords + 4 * idx[2];
    const float tu = lerp2(t0[0], t1[0], t2[0], u, v), tv = lerp2(t0[1], t1[1], t2[1], u, v);
    *tmp = *m;
    if (m->tex_kd) { const v3 c = tex_lookup(sc, sc->textures + (m->tex_kd - 1), tu, tv); tmp->kd[0] = c.x; tmp->kd[1] = c.y;
        tmp->kd[2] = c.z; }
    if (m->tex_ks) { const v3 c = tex_lookup(sc, sc->textures + (m->tex_ks - 1), tu, tv); tmp->ks[0] = c.x; tmp->ks[1] = c.y;
        tmp->ks[2] = c.z; }
    if (m->type == MAT_MIX) {
        const float ls = luminance(LD3(tmp->ks)), ld = luminance(LD3(tmp->kd));
        tmp->mix_k = (ls + ld == 0.0f) ? 0.0f : ls / (ls + ld);
    }
    return tmp;
}

static inline v3 sample_triangle(float u, float v, v3 v0, v3 v1, v3 v2) {              /* random.impala:49-60 */
    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    return add(add(mulf(v0, 1.0f - v - u), mulf(v1, u)), mulf(v2, v));
}

static int trace(const struct Scene* sc, v3 org, v3 dir, float tmin, float tmax, int any, struct Hit1* h) {
    struct Ray1 r = {{org.x, org.y, org.z}, tmin, {dir.x, dir.y, dir.z}, tmax};
    oracle_bvh2_tri1(sc->nodes, sc->tris, &r, h, 1, any, 0);
    return h->tri_id >= 0;
}

/* One path vertex (what a primary-stream entry holds, driver.impala:63-104) and what shading it produces: an emission
 * sample, at most one shadow ray, at most one continuation.  Order: on_hit, on_shadow, on_bounce (renderer.impala:62-162,
 * mapping_gpu.impala:82-134).  Used by oracle_render below and by the CPU wavefront restatement (cpu_wavefront.cpp). */
struct OracleVertex { float org[3], dir[3]; int32_t prim; float t, u, v; uint32_t rnd; float mis; float contrib[3]; int32_t depth; };
struct OracleShade  { int32_t emits; float emitted[3]; int32_t shadow; float s_org[3], s_dir[3], s_color[3];
                      int32_t bounce; float b_org[3], b_dir[3], contrib[3]; uint32_t rnd; float mis; };
#define ST3(dst, val) do { const v3 tmp_ = (val); (dst)[0] = tmp_.x; (dst)[1] = tmp_.y; (dst)[2] = tmp_.z; } while (0)

void oracle_shade_vertex(const struct Scene* sc, const struct OracleVertex* pv, int32_t max_path_len, struct OracleShade* o) {
    const float pdf_lightpick = 1.0f / (float)sc->num_lights;
    uint32_t rnd = pv->rnd;
    const v3 org = LD3(pv->org), dir = LD3(pv->dir), contrib = LD3(pv->contrib);
    const int32_t prim = pv->prim;
    struct Material textured;
    const struct Material* m = resolve_material(sc, sc->materials + sc->indices[4 * prim + 3], &textured, prim, pv->u, pv->v);
    const Surf s = surface_element(sc, org, dir, prim, pv->t, pv->u, pv->v);
    const v3 out_dir = neg(dir);
    o->emits = 0; o->shadow = 0; o->bounce = 0;
    /* on_hit (renderer.impala:113-128) */
    if (m->emissive && s.entering) {
        const struct Light* L = sc->lights + sc->light_ids[prim];
        const float pdf_dir = cosine_hemisphere_pdf(dot(LD3(L->n), out_dir));
        const v3 intensity = pdf_dir > 0.0f ? LD3(L->color) : V(0, 0, 0);      /* make_emission_value, light.impala:87-102 */
        const float pdf_area = pdf_dir > 0.0f ? L->inv_area : 1.0f;
        const float next_mis = pv->mis * pv->t * pv->t / dot(out_dir, s.local.c2);
        const float w = 1.0f / (1.0f + next_mis * pdf_lightpick * pdf_area);
        o->emits = 1; ST3(o->emitted, mulf(mul(contrib, intensity), w));
    }
    /* on_shadow (renderer.impala:69-111) */
    if (!bsdf_is_specular(m)) {
        const int32_t light_id = (int32_t)(xorshift(&rnd) & 0x7FFFFFFFu) % sc->num_lights;
        const struct Light* L = sc->lights + light_id;
        const float lu = randf(&rnd), lv = randf(&rnd);
        const v3 pos = sample_triangle(lu, lv, LD3(L->v0), LD3(L->v1), LD3(L->v2));
        const v3 from_dir = sub(s.point, pos);
        float lcos = dot(from_dir, LD3(L->n)) / len(from_dir);                    /* light.impala:124-128 */
        v3 intensity = LD3(L->color); float pdf_area = L->inv_area;
        if (!(pdf_area > 0.0f && cosine_hemisphere_pdf(lcos) > 0.0f && lcos > 0.0f)) { intensity = V(0, 0, 0); pdf_area = 1.0f;
            lcos = 0.0f; }
        const v3 light_dir = sub(pos, s.point);
        const float vis = dot(light_dir, s.local.c2);
        if (vis > 0.0f && lcos > 0.0f) {
            const float inv_d = 1.0f / len(light_dir), inv_d2 = inv_d * inv_d;
            const v3 in_dir = mulf(light_dir, inv_d);
            const float pdf_e = bsdf_pdf(m, &s, in_dir, out_dir);
            const float pdf_l = pdf_area * pdf_lightpick, inv_pdf_l = 1.0f / pdf_l;
            const float cos_e = vis * inv_d, cos_l = lcos;
            const float w = 1.0f / (1.0f + pdf_e * cos_l * inv_d2 * inv_pdf_l);
            const float geom = cos_e * cos_l * inv_d2 * inv_pdf_l;
            o->shadow = 1;
            ST3(o->s_color, mulf(mul(intensity, mul(contrib, bsdf_eval(m, &s, in_dir, out_dir))), geom * w));
            ST3(o->s_org, s.point); ST3(o->s_dir, light_dir);
        }
    }
    /* on_bounce (renderer.impala:130-152) */
    const float lum2 = 2.0f * luminance(contrib); const float rr = lum2 > 0.75f ? 0.75f : lum2;
    if (pv->depth >= max_path_len || randf(&rnd) >= rr) return;
    const BsdfSample bs = bsdf_sample(m, &s, &rnd, out_dir);
    const v3 c2 = mul(contrib, bs.color);
    o->bounce = 1;
    o->mis = bsdf_is_specular(m) ? 0.0f : 1.0f / bs.pdf;
    ST3(o->contrib, mulf(c2, bs.cos / (bs.pdf * rr)));
    ST3(o->b_org, s.point); ST3(o->b_dir, bs.in_dir); o->rnd = rnd;
}

/* on_emit (renderer.impala:26-40, camera.impala:35-44): the sample's seed state and camera ray direction */
void oracle_emit_sample(const struct Settings* st, int32_t iter, int32_t width, int32_t height, int32_t x, int32_t y, int32_t sample,
    uint32_t* rnd_out, float* dir3) {
    const v3 cdir = LD3(st->dir), cup = LD3(st->up), cright = LD3(st->right);
    uint32_t rnd = fnv_hash(fnv_hash(fnv_hash(fnv_hash(0x811C9DC5u, (uint32_t)sample), (uint32_t)iter), (uint32_t)x), (uint32_t)y);
    const float kx = 2.0f * ((float)x + randf(&rnd)) / (float)width - 1.0f;
    const float ky = 1.0f - 2.0f * ((float)y + randf(&rnd)) / (float)height;
    ST3(dir3, normalize(add(add(mulf(cright, st->w * kx), mulf(cup, st->h * ky)), cdir)));
    *rnd_out = rnd;
}

/* Renders rows [y0, y1) of one iteration into film (w*h*3 floats, accumulated), spp samples per pixel, one path at a time. */
void oracle_render(const struct Scene* sc, const struct Settings* st, int32_t iter, int32_t spp, int32_t max_path_len,
                   int32_t width, int32_t height, int32_t y0, int32_t y1, float* film, uint64_t* ray_counts) {
    const float offset = 0.001f;
    uint64_t n_primary = 0, n_shadow = 0;
    for (int32_t y = y0; y < y1; y++) for (int32_t x = 0; x < width; x++) for (int32_t sample = 0; sample < spp; sample++) {
        struct OracleVertex pv;
        oracle_emit_sample(st, iter, width, height, x, y, sample, &pv.rnd, pv.dir);
        pv.org[0] = st->eye[0]; pv.org[1] = st->eye[1]; pv.org[2] = st->eye[2];
        float tmin = 0.0f, tmax = FLT_MAX_REF;
        pv.mis = 0.0f; pv.contrib[0] = pv.contrib[1] = pv.contrib[2] = 1.0f; pv.depth = 0;
        float* px = film + 3 * ((size_t)y * width + x);
        const float inv_spp = 1.0f / (float)spp;
        for (;;) {
            struct Hit1 h; n_primary++;
            if (!trace(sc, LD3(pv.org), LD3(pv.dir), tmin, tmax, 0, &h)) break;           /* miss: dropped (mapping_gpu.impala:347-357) */
            pv.prim = h.tri_id; pv.t = h.t; pv.u = h.u; pv.v = h.v;
            struct OracleShade o;
            oracle_shade_vertex(sc, &pv, max_path_len, &o);
            if (o.emits) { px[0] += o.emitted[0] * inv_spp; px[1] += o.emitted[1] * inv_spp; px[2] += o.emitted[2] * inv_spp; }
            if (o.shadow) {
                struct Hit1 sh; n_shadow++;
                if (!trace(sc, LD3(o.s_org), LD3(o.s_dir), offset, 1.0f - offset, 1, &sh)) {      /* mapping_gpu.impala:47-80 */
                    px[0] += o.s_color[0] * inv_spp; px[1] += o.s_color[1] * inv_spp; px[2] += o.s_color[2] * inv_spp;
                }
            }
            if (!o.bounce) break;
            for (int k = 0; k < 3; k++) { pv.org[k] = o.b_org[k]; pv.dir[k] = o.b_dir[k]; pv.contrib[k] = o.contrib[k]; }
            pv.rnd = o.rnd; pv.mis = o.mis; pv.depth++; tmin = offset; tmax = FLT_MAX_REF;
        }
    }
    if (ray_counts) { ray_counts[0] += n_primary; ray_counts[1] += n_shadow; }
}

/* Single-function probes for property tests */
void oracle_sincos_2pi(const float* u, float* c, float* s, int32_t n) { for (int32_t i = 0; i < n; i++) sincos_2pi(u[i], &c[i], &s[i]); }
/* probes for the tests: texture lookup and the per-hit material of a textured scene */
void oracle_tex_lookup(const struct Scene* sc, int32_t tex, const float* uv, float* rgb, int32_t n) {
    for (int32_t i = 0; i < n; i++) { const v3 c = tex_lookup(sc, sc->textures + tex, uv[2 * i], uv[2 * i + 1]); rgb[3 * i] = c.x;
        rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z; }
}
void oracle_hit_material(const struct Scene* sc, int32_t prim, float u, float v, struct Material* out) {
    struct Material tmp; *out = *resolve_material(sc, sc->materials + sc->indices[4 * prim + 3], &tmp, prim, u, v);
}
void oracle_fastpow(const float* x, const float* p, float* out, int32_t n) { for (int32_t i = 0; i < n; i++) out[i] = fastpow(x[i], p[i]); }
void oracle_randf(uint32_t seed, float* out, int32_t n) { for (int32_t i = 0; i < n; i++) out[i] = randf(&seed); }
uint32_t oracle_seed(int32_t sample, int32_t iter, int32_t x, int32_t y) {
    return fnv_hash(fnv_hash(fnv_hash(fnv_hash(0x811C9DC5u, (uint32_t)sample), (uint32_t)iter), (uint32_t)x), (uint32_t)y);
}
/* samples a BSDF n times for a fixed frame (normal +z, entering) and out_dir; outputs in_dir[3], pdf, cos, color[3] per sample */
void oracle_bsdf_samples(const struct Material* m, const float* out_dir3, uint32_t seed, float* out8, int32_t n) {
    Surf s; s.entering = 1; s.point = V(0, 0, 0); s.face_normal = V(0, 0, 1); s.local = orthonormal(V(0, 0, 1));
    const v3 od = LD3(out_dir3);
    for (int32_t i = 0; i < n; i++) {
        const BsdfSample b = bsdf_sample(m, &s, &seed, od);
        float* o = out8 + 10 * i;
        o[0] = b.in_dir.x; o[1] = b.in_dir.y; o[2] = b.in_dir.z; o[3] = b.pdf; o[4] = b.cos; o[5] = b.color.x; o[6] = b.color.y;
        o[7] = b.color.z;
        o[8] = bsdf_pdf(m, &s, b.in_dir, od); const v3 e = bsdf_eval(m, &s, b.in_dir, od); o[9] = e.x;
    }
}
