// shading.h -- device-side path-tracing arithmetic of the wavefront renderer.
//
// Restates (for the GPU) src/core/{random,common,matrix,color,vector}.impala, src/render/{material,
// light,geometry}.impala -- see the line references on each function.  The expressions are written
// as plain fp32 operations in a fixed order and the file is compiled with -ffp-contract=off, so every
// path evaluates bit-identically to the CPU restatement used as the parity oracle (which was written
// first; this header is the same author's device adaptation of those expressions).  sin/cos use the
// fixed polynomial sincos_2pi() instead of the platform's sinf/cosf for the same reason.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "rodent_render.h"

#define RD_FN __device__ __forceinline__

struct SceneDev {                 // device pointers (uploaded by rodent_hip_scene_create)
    const float* vertices; const float* normals; const float* face_normals; const int32_t* indices;
    const Node2* nodes; const Tri1* tris;
    const RodentMaterial* materials; const RodentLight* lights; const int32_t* light_ids;
    int32_t num_tris, num_materials, num_lights, pad;
    const float* texcoords; const RodentTexture* textures; const uint32_t* texels;
    // the first kSceneTopNodes inner nodes, breadth first, as LDS-image records (traversal_device.h); built at scene creation
    const int4* top_image;
    const int4* top_image_large;  // the same with kPersistTopNodes records (persistent stream traversal kernels)
    // What a hit needs of its triangle, gathered once at scene creation into ONE 48-byte record per triangle: face normal, then the three
    // vertex normals (12 floats; rodent_hip_scene_create).  Through indices -> normals the shader waits for two dependent fetches (16 + 12
    // bytes, then 3 x 12 bytes scattered over the vertex array); the record is one fetch of three consecutive 16-byte words.  Same values,
    // same arithmetic.  May be null (old path).
    const float4* tri_shade;
    // textured scenes: the three corners' texture coordinates per triangle, gathered the same way (6 floats); null otherwise
    const float* tri_tex;
};

#define FLT_MAX_REF 3.4028234664e+38f
#define FLT_PI 3.14159265359f

typedef struct { float x, y, z; } v3;
RD_FN v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
RD_FN v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
RD_FN v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
RD_FN v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
RD_FN v3 mulf(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
RD_FN v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
RD_FN float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }              /* vector.impala:60 */
RD_FN v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RD_FN float len(v3 a) { return sqrtf(dot(a, a)); }
RD_FN v3 normalize(v3 a) { return mulf(a, 1.0f / len(a)); }                           /* vector.impala:82 */
RD_FN v3 reflect(v3 v, v3 n) { return sub(mulf(n, 2.0f * dot(n, v)), v); }            /* vector.impala:74 */
RD_FN float lerp1(float a, float b, float k) { return (1.0f - k) * a + k * b; }        /* common.impala:118 */
RD_FN float lerp2(float a, float b, float c, float k1, float k2) { return (1.0f - k1 - k2) * a + k1 * b + k2 * c; }
RD_FN float positive_cos(v3 a, v3 b) { const float c = dot(a, b); return c >= 0.0f ? c : 0.0f; }
RD_FN float luminance(v3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }  /* color.impala:33 */
RD_FN v3 color_lerp(v3 a, v3 b, float t) { return V((1.0f - t) * a.x + t * b.x, (1.0f - t) * a.y + t * b.y, (1.0f - t) * a.z + t * b.z); }

RD_FN uint32_t f2u(float x) { return __float_as_uint(x); }
RD_FN float u2f(uint32_t u) { return __uint_as_float(u); }

/* random.impala:22-30, 7-11 */
RD_FN uint32_t xorshift(uint32_t* seed) { uint32_t x = *seed; x = x == 0u ? 1u : x; x ^= x << 13; x ^= x >> 17; x ^= x << 5; *seed = x;
    return x; }
RD_FN float randf(uint32_t* rnd) { return u2f((127u << 23) | (xorshift(rnd) & 0x7FFFFFu)) - 1.0f; }
/* random.impala:116-126 */
RD_FN uint32_t fnv_hash(uint32_t h, uint32_t d) {
    h = (h * 16777619u) ^ (d & 0xFFu); h = (h * 16777619u) ^ ((d >> 8) & 0xFFu);
    h = (h * 16777619u) ^ ((d >> 16) & 0xFFu); h = (h * 16777619u) ^ ((d >> 24) & 0xFFu); return h;
}

/* common.impala:42-61 */
RD_FN float fastlog2(float x) {
    const uint32_t vx = f2u(x); const uint32_t mx = (vx & 0x007FFFFFu) | 0x3f000000u;
    const float y = (float)vx * 1.1920928955078125e-7f; const float z = u2f(mx);
    return y - 124.22551499f - 1.498030302f * z - 1.72587999f / (0.3520887068f + z);
}
RD_FN float fastpow2(float p) {
    const float offset = p < 0.0f ? 1.0f : 0.0f; const float clipp = p < -126.0f ? -126.0f : p;
    const int32_t w = (int32_t)clipp; const float z = clipp - (float)w + offset;
    const int32_t v = (int32_t)((float)(1u << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z));
    return u2f((uint32_t)v);
}
RD_FN float fastpow(float x, float p) { return fastpow2(p * fastlog2(x)); }

/* cos(2 pi u), sin(2 pi u) for u in [0,1): quadrant reduction + fixed odd/even polynomials on
 * [-pi/4, pi/4] (Taylor coefficients, |error| < 1e-7); every operation is a plain fp32 op. */
RD_FN void sincos_2pi(float u, float* c_out, float* s_out) {
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
RD_FN m3 orthonormal(v3 n) {
    const float sign = n.z >= 0.0f ? 1.0f : -1.0f; const float a = -1.0f / (sign + n.z); const float b = n.x * n.y * a;
    m3 m; m.c0 = V(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x); m.c1 = V(b, sign + n.y * n.y * a, -n.y); m.c2 = n; return m;
}
RD_FN v3 m3_mul(m3 m, v3 v) {   /* matrix.impala mat3x3_mul: rows dotted with v */
    return V(m.c0.x * v.x + m.c1.x * v.y + m.c2.x * v.z, m.c0.y * v.x + m.c1.y * v.y + m.c2.y * v.z,
        m.c0.z * v.x + m.c1.z * v.y + m.c2.z * v.z);
}

typedef struct { v3 dir; float pdf; } DirSample;
RD_FN DirSample make_dir_sample(float c, float s, float u, float pdf) {      /* random.impala:38-47, phi = 2 pi u */
    float cp, sp; sincos_2pi(u, &cp, &sp); DirSample d; d.dir = V(s * cp, s * sp, c); d.pdf = pdf; return d;
}
RD_FN float cosine_hemisphere_pdf(float c) { return c * (1.0f / FLT_PI); }
RD_FN DirSample sample_cosine_hemisphere(float u, float v) {                  /* random.impala:72-79 */
    const float c = sqrtf(1.0f - v), s = sqrtf(v); return make_dir_sample(c, s, u, cosine_hemisphere_pdf(c));
}
RD_FN float cosine_power_hemisphere_pdf(float c, float k) { return fastpow(c, k) * (k + 1.0f) * (1.0f / (2.0f * FLT_PI)); }
RD_FN DirSample sample_cosine_power_hemisphere(float k, float u, float v) {    /* random.impala:87-101 */
    const float p = fastpow(v, 1.0f / (k + 1.0f)); const float c = p < 1.0f ? p : 1.0f;
    const float s = sqrtf(1.0f - c * c); const float pow_c_k = c != 0.0f ? v / c : 0.0f;
    return make_dir_sample(c, s, u, pow_c_k * (k + 1.0f) * (1.0f / (2.0f * FLT_PI)));
}

typedef struct { int entering; v3 point, face_normal; m3 local; } Surf;
typedef struct { v3 in_dir; float pdf, cos; v3 color; } BsdfSample;

/* material.impala:63-72 */
RD_FN BsdfSample make_bsdf_sample(const Surf* s, v3 in_dir, float pdf, float cosv, v3 color, int inverted) {
    const int valid = (pdf > 0.0f) && (inverted ^ (dot(in_dir, s->face_normal) > 0.0f));
    BsdfSample r; r.in_dir = in_dir; r.pdf = valid ? pdf : 1.0f; r.cos = cosv; r.color = valid ? color : V(0, 0, 0); return r;
}
RD_FN v3 LD3(const float* p) { return V(p[0], p[1], p[2]); }

/* diffuse: material.impala:85-100; phong: :103-123 */
RD_FN v3 diffuse_eval(const RodentMaterial* m) { return mulf(LD3(m->kd), 1.0f / FLT_PI); }
RD_FN float diffuse_pdf(const Surf* s, v3 in_dir) { return cosine_hemisphere_pdf(positive_cos(in_dir, s->local.c2)); }
RD_FN BsdfSample diffuse_sample(const RodentMaterial* m, const Surf* s, uint32_t* rnd) {
    const float u = randf(rnd), v = randf(rnd); const DirSample d = sample_cosine_hemisphere(u, v);
    return make_bsdf_sample(s, m3_mul(s->local, d.dir), d.pdf, d.dir.z, mulf(LD3(m->kd), 1.0f / FLT_PI), 0);
}
RD_FN v3 phong_eval(const RodentMaterial* m, const Surf* s, v3 in_dir, v3 out_dir) {
    const float c = positive_cos(in_dir, reflect(out_dir, s->local.c2));
    return mulf(LD3(m->ks), fastpow(c, m->ns) * (m->ns + 2.0f) * (1.0f / (2.0f * FLT_PI)));
}
RD_FN float phong_pdf(const RodentMaterial* m, const Surf* s, v3 in_dir, v3 out_dir) {
    return cosine_power_hemisphere_pdf(positive_cos(in_dir, reflect(out_dir, s->local.c2)), m->ns);
}
RD_FN BsdfSample phong_sample(const RodentMaterial* m, const Surf* s, uint32_t* rnd, v3 out_dir) {
    const v3 r = reflect(out_dir, s->local.c2);
    const float u = randf(rnd), v = randf(rnd); const DirSample d = sample_cosine_power_hemisphere(m->ns, u, v);
    const v3 in_dir = m3_mul(orthonormal(r), d.dir); const float c = positive_cos(in_dir, s->local.c2);
    return make_bsdf_sample(s, in_dir, d.pdf, c, mulf(LD3(m->ks), d.pdf * (m->ns + 2.0f) / (m->ns + 1.0f)), 0);
}
RD_FN float fresnel_factor(float k, float ci, float ct) {                      /* material.impala:39-43 */
    const float rs = (k * ci - ct) / (k * ci + ct), rp = (ci - k * ct) / (ci + k * ct); return (rs * rs + rp * rp) * 0.5f;
}

RD_FN int bsdf_is_specular(const RodentMaterial* m) { return m->type == RODENT_BSDF_MIRROR || m->type == RODENT_BSDF_GLASS; }
RD_FN v3 bsdf_eval(const RodentMaterial* m, const Surf* s, v3 in_dir, v3 out_dir) {
    switch (m->type) {
        case RODENT_BSDF_DIFFUSE: return diffuse_eval(m);
        case RODENT_BSDF_PHONG:   return phong_eval(m, s, in_dir, out_dir);
        case RODENT_BSDF_MIX:     return color_lerp(diffuse_eval(m), phong_eval(m, s, in_dir, out_dir), m->mix_k);   /* :166-171 */
        default:          return V(0, 0, 0);
    }
}
RD_FN float bsdf_pdf(const RodentMaterial* m, const Surf* s, v3 in_dir, v3 out_dir) {
    switch (m->type) {
        case RODENT_BSDF_DIFFUSE: return diffuse_pdf(s, in_dir);
        case RODENT_BSDF_PHONG:   return phong_pdf(m, s, in_dir, out_dir);
        case RODENT_BSDF_MIX:     return lerp1(diffuse_pdf(s, in_dir), phong_pdf(m, s, in_dir, out_dir), m->mix_k);
        default:          return 0.0f;
    }
}
RD_FN BsdfSample bsdf_sample(const RodentMaterial* m, const Surf* s, uint32_t* rnd, v3 out_dir) {
    switch (m->type) {
        case RODENT_BSDF_DIFFUSE: return diffuse_sample(m, s, rnd);
        case RODENT_BSDF_PHONG:   return phong_sample(m, s, rnd, out_dir);
        case RODENT_BSDF_MIX: {                                                                /* material.impala:176-189 */
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
        case RODENT_BSDF_MIRROR: return make_bsdf_sample(s, reflect(out_dir, s->local.c2), 1.0f, 1.0f, LD3(m->ks), 0);   /* :126-135 */
        case RODENT_BSDF_GLASS: {                                                              /* :138-163, n1 = 1, n2 = Ni, not adjoint */
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
RD_FN Surf surface_element(const SceneDev* sc, v3 org, v3 dir, int32_t prim, float t, float u, float v) {
    v3 fn, nrm;
    if (sc->tri_shade) {                                       /* wave-uniform: the gathered record (see SceneDev::tri_shade) */
        const float4* rec = sc->tri_shade + 3 * (size_t)prim;
        const float4 a = rec[0], b = rec[1], c = rec[2];       /* fn.xyz n0.x | n0.yz n1.xy | n1.z n2.xyz */
        fn = V(a.x, a.y, a.z);
        nrm = normalize(V(lerp2(a.w, b.z, c.y, u, v), lerp2(b.x, b.w, c.z, u, v), lerp2(b.y, c.x, c.w, u, v)));
    } else {
        const int32_t* idx = sc->indices + 4 * prim;
        fn = LD3(sc->face_normals + 4 * prim);
        const float* n0 = sc->normals + 4 * idx[0]; const float* n1 = sc->normals + 4 * idx[1]; const float* n2 = sc->normals + 4 * idx[2];
        nrm = normalize(V(lerp2(n0[0], n1[0], n2[0], u, v), lerp2(n0[1], n1[1], n2[1], u, v), lerp2(n0[2], n1[2], n2[2], u, v)));
    }
    Surf s; s.entering = dot(dir, fn) <= 0.0f; s.point = add(org, mulf(dir, t));
    s.face_normal = s.entering ? fn : neg(fn); s.local = orthonormal(dot(dir, nrm) <= 0.0f ? nrm : neg(nrm)); return s;
}

/* image.impala:24-38 (RGBA8 -> colour), :48-54 (repeat border), :64-86 (bilinear filter) */
RD_FN v3 texel(const SceneDev* sc, const RodentTexture* t, int32_t x, int32_t y) {
    const uint32_t p = sc->texels[t->offset + (uint32_t)y * (uint32_t)t->width + (uint32_t)x];
    return V((float)(p & 0xFFu) * (1.0f / 255.0f), (float)((p >> 8) & 0xFFu) * (1.0f / 255.0f),
        (float)((p >> 16) & 0xFFu) * (1.0f / 255.0f));
}
RD_FN v3 tex_lookup(const SceneDev* sc, const RodentTexture* t, float tu, float tv) {
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
RD_FN const RodentMaterial* resolve_material(const SceneDev* sc, const RodentMaterial* m, RodentMaterial* tmp, int32_t prim, float u,
    float v) {
    if (!(m->tex_kd | m->tex_ks)) return m;
    float tu, tv;
    if (sc->tri_tex) {                                         /* wave-uniform: the gathered corners (SceneDev::tri_tex) */
        const float* tc = sc->tri_tex + 6 * (size_t)prim;
        tu = lerp2(tc[0], tc[2], tc[4], u, v); tv = lerp2(tc[1], tc[3], tc[5], u, v);
    } else {
        const int32_t* idx = sc->indices + 4 * prim;
        const float* t0 = sc->texcoords + 4 * idx[0]; const float* t1 = sc->texcoords + 4 * idx[1];
        const float* t2 = sc->texcoords + 4 * idx[2];
        tu = lerp2(t0[0], t1[0], t2[0], u, v); tv = lerp2(t0[1], t1[1], t2[1], u, v);
    }
    *tmp = *m;
    if (m->tex_kd) { const v3 c = tex_lookup(sc, sc->textures + (m->tex_kd - 1), tu, tv); tmp->kd[0] = c.x; tmp->kd[1] = c.y;
        tmp->kd[2] = c.z; }
    if (m->tex_ks) { const v3 c = tex_lookup(sc, sc->textures + (m->tex_ks - 1), tu, tv); tmp->ks[0] = c.x; tmp->ks[1] = c.y;
        tmp->ks[2] = c.z; }
    if (m->type == RODENT_BSDF_MIX) {
        const float ls = luminance(LD3(tmp->ks)), ld = luminance(LD3(tmp->kd));
        tmp->mix_k = (ls + ld == 0.0f) ? 0.0f : ls / (ls + ld);
    }
    return tmp;
}

RD_FN v3 sample_triangle(float u, float v, v3 v0, v3 v1, v3 v2) {              /* random.impala:49-60 */
    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    return add(add(mulf(v0, 1.0f - v - u), mulf(v1, u)), mulf(v2, v));
}

