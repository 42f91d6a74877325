/*
 * model_packet.c -- CPU model of wave-packet traversal of BVH2/Tri1 (analysis aid for scripts/model_packet.py).
 * LAB CODE: not part of the product, not part of the parity oracle.  It reuses the oracle's ray / box / triangle arithmetic
 * (included below) so that what it counts is what a HIP kernel with the same operation sequence would do.
 *
 * The reference's CPU default is a packet kernel with a single-ray fallback (src/traversal/mapping_cpu.impala:259-384: one
 * shared stack per packet, every node fetched once per packet, thresholds :267-272, fallback :305-321).  This file simulates the
 * wave64 form of that idea on 64-ray chunks of a ray list:
 *   packet phase   one shared stack of (node, lane mask); an inner node is fetched once, every lane of the mask tests both
 *                  child boxes against its own ray and tmax, ballots decide descend / push / pop, the order by the majority
 *                  of the lanes' own near-child preference; a leaf's triangles are fetched once each and tested by the mask.
 *   fallback       a subtree entered by fewer than T lanes is left to those lanes' own per-lane traversal (the single-step
 *                  schedule of k_bvh2_top_persist: one node step or one triangle test per lane per wave iteration),
 *                  either at once while the other lanes idle (mode 0, the reference's order) or DEFERRED: the lanes put the
 *                  node on their own stacks, the packet goes on, and all lanes drain their stacks together afterwards (mode 1).
 * T = 65 is the existing kernel (everything falls back at the root): its results must equal oracle B1 bit for bit, which
 * checks the simulator.  Per chunk the model returns step counts by kind; the Python side prices them.
 */
#include "../oracle/traversal_oracle.c"

#define W 64
#define PSTACK 256

struct LaneState {
    struct RayX ray;
    int32_t hit_id; float hit_t, hit_u, hit_v;
    int32_t top;             /* per-lane loop: 0 = idle */
    int32_t mem[4 * STACK_CAP]; int32_t ptr;
    int done;                /* any-hit: finished */
    int32_t deferred[STACK_CAP]; int ndeferred;
};

struct ChunkCounts {          /* per chunk, all uint32 */
    uint32_t p_node_img, p_node_mem, p_tri;       /* packet steps: inner node in the LDS image / from memory, triangle */
    uint32_t p_lanes_node, p_lanes_tri;           /* sum of active lanes over those steps */
    uint32_t f_it_node, f_it_mixed, f_it_tri;     /* per-lane phase: wave iterations with node lanes only / both kinds / triangle lanes only */
    uint32_t f_lane_steps;                        /* lanes stepping, summed over those iterations */
    uint32_t f_phases;                            /* fallbacks taken (mode 0: per-lane phases; mode 1: deferred entries) */
    uint32_t max_deferred;                        /* mode 1: most entries one lane had to keep */
    uint32_t window_overflows;                    /* mode 2: rays whose kept entries + own depth exceed the LDS window (WINDOW entries): they would go to the deep list */
};

static inline void box2(const struct RayX* ray, const struct Node2* nd, int* h0, int* h1, float* te0, float* te1) {
    int hk[2]; float te[2];
    for (int k = 0; k < 2; k++) {
        const float* b = nd->bounds + 6 * k;
        const float t0x = fmaf(ray->idx, b[0], ray->iox), t1x = fmaf(ray->idx, b[1], ray->iox);
        const float t0y = fmaf(ray->idy, b[2], ray->ioy), t1y = fmaf(ray->idy, b[3], ray->ioy);
        const float t0z = fmaf(ray->idz, b[4], ray->ioz), t1z = fmaf(ray->idz, b[5], ray->ioz);
        const float tentry = fmax_ref(fmax_ref(fmin_ref(t0x, t1x), fmin_ref(t0y, t1y)), fmax_ref(fmin_ref(t0z, t1z), ray->tmin));
        const float texit  = fmin_ref(fmin_ref(fmax_ref(t0x, t1x), fmax_ref(t0y, t1y)), fmin_ref(fmax_ref(t0z, t1z), ray->tmax));
        hk[k] = (tentry <= texit) && nd->child[k] != 0; te[k] = tentry;
    }
    *h0 = hk[0]; *h1 = hk[1]; *te0 = te[0]; *te1 = te[1];
}

/* one triangle for one lane; returns 1 when the leaf ends with it (sentinel) */
static inline void tri_test(struct LaneState* L, const struct Tri1* tr, int any_hit) {
    float t, u, v;
    if (intersect_tri1(&L->ray, tr, &t, &u, &v)) {
        L->hit_id = tr->prim_id & 0x7FFFFFFF; L->hit_t = t; L->hit_u = u; L->hit_v = v;
        L->ray.tmax = t;
        if (any_hit) L->done = 1;
    }
}

/* one step of the single-step schedule for one lane (bvh2_step): returns 1 for a node step, 2 for a triangle step */
static inline int lane_step(struct LaneState* L, const struct Node2* nodes, const struct Tri1* tris, int any_hit) {
    if (L->top > 0) {
        const struct Node2* nd = &nodes[L->top - 1];
        int h0, h1; float te0, te1;
        box2(&L->ray, nd, &h0, &h1, &te0, &te1);
        if (!h0 && !h1) { L->top = L->mem[L->ptr]; L->ptr--; }
        else if (h0 && h1) {
            const int c0first = te0 < te1;
            L->mem[++L->ptr] = c0first ? nd->child[1] : nd->child[0];
            L->top = c0first ? nd->child[0] : nd->child[1];
        } else L->top = h0 ? nd->child[0] : nd->child[1];
        return 1;
    }
    const int32_t j = ~L->top;
    const struct Tri1* tr = &tris[j];
    tri_test(L, tr, any_hit);
    if (any_hit && L->done) { L->top = 0; return 2; }
    if (tr->prim_id < 0) { L->top = L->mem[L->ptr]; L->ptr--; }
    else L->top = L->top - 1;                       /* ~(j + 1) */
    return 2;
}

/* lock-step per-lane loop over the lanes whose top != 0 */
/* analysis aid: lanes stepping / iterations run, by iteration index of the per-lane loop (2 x 512 uint64; NULL = off) */
static uint64_t* g_iter_profile = 0;
void model_set_iteration_profile(uint64_t* buf) { g_iter_profile = buf; }

static int g_window = 1 << 30;      /* entries of the per-lane LDS window (mode 2) */
static void lane_loop(struct LaneState* lanes, const struct Node2* nodes, const struct Tri1* tris, int any_hit, struct ChunkCounts* c) {
    uint64_t over = 0;
    for (int it = 0;; it++) {
        int nn = 0, nt = 0;
        for (int l = 0; l < W; l++) {
            if (lanes[l].top == 0) continue;
            if (lane_step(&lanes[l], nodes, tris, any_hit) == 1) nn++; else nt++;
            if (lanes[l].ptr >= g_window && !((over >> l) & 1)) { over |= 1ull << l; c->window_overflows++; }
        }
        if (nn + nt == 0) break;
        if (nt == 0) c->f_it_node++; else if (nn == 0) c->f_it_tri++; else c->f_it_mixed++;
        c->f_lane_steps += (uint32_t)(nn + nt);
        if (g_iter_profile) { const int k = it < 511 ? it : 511; g_iter_profile[k] += (uint64_t)(nn + nt); g_iter_profile[512 + k]++; }
    }
}

static inline int popc(uint64_t m) { return __builtin_popcountll(m); }

/*
 * mode 3, "steal": the per-lane kernel with WORK STEALING INSIDE THE WAVE.  The launch at 1 Mi rays ends with waves in which a few lanes walk long
 * paths while the others idle; a lane's stack entries are independent subtrees, so from iteration i0 on, every `every` iterations, each idle lane takes
 * the top stack entry of a lane that has one (its ray and its tmax are SHARED: the thief tests against the owner's current tmax and shortens it, a hit
 * record belongs to the ray).  The per-ray visit order is not the reference's any more: t stays the minimum over all accepted triangles, exact ties may
 * resolve to another triangle.  Returns through `c` the iterations by kind (as lane_loop), the stealing events in f_phases, entries moved in max_deferred.
 */
struct Worker { int owner; int32_t top; int32_t mem[4 * STACK_CAP]; int ptr; };
static void steal_loop(struct LaneState* lanes, uint64_t valid, const struct Node2* nodes, const struct Tri1* tris, int any_hit, int i0, int every, struct ChunkCounts* c) {
    static struct Worker w[W];
    for (int l = 0; l < W; l++) { w[l].owner = l; w[l].top = ((valid >> l) & 1) ? 1 : 0; w[l].ptr = 0; w[l].mem[0] = 0; }
    for (int it = 0;; it++) {
        int nn = 0, nt = 0;
        for (int l = 0; l < W; l++) {
            struct Worker* k = &w[l];
            if (k->top == 0) continue;
            struct LaneState* L = &lanes[k->owner];
            if (any_hit && L->done) { k->top = 0; continue; }
            if (k->top > 0) {
                const struct Node2* nd = &nodes[k->top - 1];
                int h0, h1; float te0, te1;
                box2(&L->ray, nd, &h0, &h1, &te0, &te1);
                if (!h0 && !h1) { k->top = k->mem[k->ptr]; k->ptr--; }
                else if (h0 && h1) { const int c0first = te0 < te1; k->mem[++k->ptr] = c0first ? nd->child[1] : nd->child[0]; k->top = c0first ? nd->child[0] : nd->child[1]; }
                else k->top = h0 ? nd->child[0] : nd->child[1];
                nn++;
            } else {
                const struct Tri1* tr = &tris[~k->top];
                tri_test(L, tr, any_hit);
                if (any_hit && L->done) k->top = 0;
                else if (tr->prim_id < 0) { k->top = k->mem[k->ptr]; k->ptr--; }
                else k->top = k->top - 1;
                nt++;
            }
        }
        if (nn + nt == 0) break;
        if (nt == 0) c->f_it_node++; else if (nn == 0) c->f_it_tri++; else c->f_it_mixed++;
        c->f_lane_steps += (uint32_t)(nn + nt);
        if (g_iter_profile) { const int q = it < 511 ? it : 511; g_iter_profile[q] += (uint64_t)(nn + nt); g_iter_profile[512 + q]++; }
        if (it >= i0 && (it - i0) % every == 0) {
            int thief = 0, moved = 0;
            for (int d = 0; d < W; d++) {
                if (w[d].top == 0 || w[d].ptr < 1) continue;
                while (thief < W && w[thief].top != 0) thief++;
                if (thief >= W) break;
                w[thief].owner = w[d].owner; w[thief].top = w[d].mem[w[d].ptr--]; w[thief].ptr = 0; w[thief].mem[0] = 0;
                moved++; thief++;
            }
            if (moved) { c->f_phases++; c->max_deferred += (uint32_t)moved; }
        }
    }
}

/*
 * in_image[i] != 0: node i (0-based) is in the LDS image.  mode: 0 immediate fallback, 1 deferred.  threshold T: a subtree
 * entered by fewer than T lanes falls back.  hits: Hit1 per ray.  counts: one record per chunk.
 */
int model_packet(const struct Node2* nodes, const struct Tri1* tris, const struct Ray1* rays, struct Hit1* hits, int32_t n,
                 int32_t any_hit, int32_t mode, int32_t T, const uint8_t* in_image, struct ChunkCounts* counts, uint64_t* hist /* [65] active lanes per packet visit */) {
    static struct LaneState lanes[W];
    int overflow = 0;
    g_window = mode == 2 ? 15 : (1 << 30);
    for (int32_t first = 0, chunk = 0; first < n; first += W, chunk++) {
        struct ChunkCounts c; memset(&c, 0, sizeof c);
        uint64_t valid = 0;
        for (int l = 0; l < W; l++) {
            struct LaneState* L = &lanes[l];
            const int32_t i = first + l;
            if (i < n) { L->ray = make_ray(&rays[i]); valid |= 1ull << l; }
            L->hit_id = -1; L->hit_t = L->ray.tmax; L->hit_u = 0.0f; L->hit_v = 0.0f;
            L->top = 0; L->ptr = 0; L->mem[0] = 0; L->done = 0; L->ndeferred = 0;
        }
        int32_t snode[PSTACK]; uint64_t smask[PSTACK]; int sp = 0;
        int32_t cur = 1; uint64_t curmask = valid;
        if (mode == 3) {
            steal_loop(lanes, valid, nodes, tris, any_hit, T & 255, T >> 8, &c);
        } else
        if (mode == 2) {
            /* the buildable form: no masks on the shared stack (every live lane tests every node the packet pops: child boxes lie inside their
             * parent's, so a lane that missed the parent misses the children); the fallback is decided at the PARENT -- a child that fewer than T
             * lanes hit goes onto those lanes' own stacks (each lane its nearer child first) unless one of them already keeps dlim entries */
            const int dlim = T >> 8; const int TT = T & 255;
            uint64_t alive = valid;
            if (popc(valid) < TT) { for (int l = 0; l < W; l++) if ((valid >> l) & 1) lanes[l].deferred[lanes[l].ndeferred++] = 1; cur = 0; }
            for (;;) {
                if (cur == 0) { if (sp == 0) break; sp--; cur = snode[sp]; }
                if (any_hit) { for (int l = 0; l < W; l++) if (lanes[l].done) alive &= ~(1ull << l); }
                if (alive == 0) break;
                if (cur > 0) {
                    const struct Node2* nd = &nodes[cur - 1];
                    if (in_image[cur - 1]) c.p_node_img++; else c.p_node_mem++;
                    uint64_t m0 = 0, m1 = 0, near0 = 0; int pref0 = 0, pref1 = 0;
                    for (int l = 0; l < W; l++) {
                        if (!((alive >> l) & 1)) continue;
                        int h0, h1; float te0, te1;
                        box2(&lanes[l].ray, nd, &h0, &h1, &te0, &te1);
                        if (h0) m0 |= 1ull << l;
                        if (h1) m1 |= 1ull << l;
                        if (h0 && (!h1 || te0 < te1)) { pref0++; near0 |= 1ull << l; } else if (h1) pref1++;
                    }
                    c.p_lanes_node += (uint32_t)popc(m0 | m1);
                    hist[popc(m0 | m1)]++;
                    uint64_t full = 0;
                    for (int l = 0; l < W; l++) if (lanes[l].ndeferred >= dlim) full |= 1ull << l;
                    const int pk0 = m0 && (popc(m0) >= TT || (m0 & full)), pk1 = m1 && (popc(m1) >= TT || (m1 & full));
                    /* deferred children: each lane its own nearer child first */
                    for (int l = 0; l < W; l++) {
                        const int d0 = !pk0 && ((m0 >> l) & 1), d1 = !pk1 && ((m1 >> l) & 1);
                        if (!d0 && !d1) continue;
                        const int first = (d0 && d1) ? (((near0 >> l) & 1) ? 0 : 1) : (d0 ? 0 : 1);
                        lanes[l].deferred[lanes[l].ndeferred++] = nd->child[first]; c.f_phases++;
                        if (d0 && d1) { lanes[l].deferred[lanes[l].ndeferred++] = nd->child[1 - first]; c.f_phases++; }
                        if (lanes[l].ndeferred >= STACK_CAP - 2) overflow = 1;
                    }
                    if (pk0 && pk1) {
                        const int c0first = pref0 >= pref1;
                        if (sp >= PSTACK) { overflow = 1; cur = 0; continue; }
                        snode[sp++] = c0first ? nd->child[1] : nd->child[0];
                        cur = c0first ? nd->child[0] : nd->child[1];
                    } else cur = pk0 ? nd->child[0] : (pk1 ? nd->child[1] : 0);
                } else {
                    int32_t j = ~cur;
                    for (;;) {
                        const struct Tri1* tr = &tris[j++];
                        c.p_tri++; c.p_lanes_tri += (uint32_t)popc(alive);
                        for (int l = 0; l < W; l++) if (((alive >> l) & 1) && !lanes[l].done) tri_test(&lanes[l], tr, any_hit);
                        if (tr->prim_id < 0) break;
                    }
                    cur = 0;
                }
            }
        } else
        for (;;) {
            if (cur == 0) {
                if (sp == 0) break;
                sp--; cur = snode[sp]; curmask = smask[sp];
            }
            if (any_hit) { for (int l = 0; l < W; l++) if (lanes[l].done) curmask &= ~(1ull << l); }
            if (curmask == 0) { cur = 0; continue; }
            if (popc(curmask) < T) {
                if (mode == 0) {
                    for (int l = 0; l < W; l++) if ((curmask >> l) & 1) { lanes[l].top = cur; lanes[l].ptr = 0; lanes[l].mem[0] = 0; }
                    lane_loop(lanes, nodes, tris, any_hit, &c);
                    c.f_phases++;
                } else {
                    for (int l = 0; l < W; l++) if ((curmask >> l) & 1) {
                        if (lanes[l].ndeferred >= STACK_CAP) { overflow = 1; continue; }
                        lanes[l].deferred[lanes[l].ndeferred++] = cur;
                        c.f_phases++;
                    }
                }
                cur = 0; continue;
            }
            hist[popc(curmask)]++;
            if (cur > 0) {
                const struct Node2* nd = &nodes[cur - 1];
                if (in_image[cur - 1]) c.p_node_img++; else c.p_node_mem++;
                c.p_lanes_node += (uint32_t)popc(curmask);
                uint64_t m0 = 0, m1 = 0; int pref0 = 0, pref1 = 0;
                for (int l = 0; l < W; l++) {
                    if (!((curmask >> l) & 1)) continue;
                    int h0, h1; float te0, te1;
                    box2(&lanes[l].ray, nd, &h0, &h1, &te0, &te1);
                    if (h0) m0 |= 1ull << l;
                    if (h1) m1 |= 1ull << l;
                    if (h0 && (!h1 || te0 < te1)) pref0++; else if (h1) pref1++;
                }
                if (!m0 && !m1) cur = 0;
                else if (m0 && m1) {
                    const int c0first = pref0 >= pref1;
                    if (sp >= PSTACK) { overflow = 1; cur = 0; continue; }
                    snode[sp] = c0first ? nd->child[1] : nd->child[0]; smask[sp] = c0first ? m1 : m0; sp++;
                    cur = c0first ? nd->child[0] : nd->child[1]; curmask = c0first ? m0 : m1;
                } else { cur = m0 ? nd->child[0] : nd->child[1]; curmask = m0 ? m0 : m1; }
            } else {
                int32_t j = ~cur;
                for (;;) {
                    const struct Tri1* tr = &tris[j++];
                    c.p_tri++; c.p_lanes_tri += (uint32_t)popc(curmask);
                    for (int l = 0; l < W; l++) if (((curmask >> l) & 1) && !lanes[l].done) tri_test(&lanes[l], tr, any_hit);
                    if (tr->prim_id < 0) break;
                }
                cur = 0;
            }
        }
        if (mode == 1 || mode == 2) {
            /* the lanes drain what they kept: entries in the order the packet met them (near first), i.e. reversed onto the stack */
            for (int l = 0; l < W; l++) {
                struct LaneState* L = &lanes[l];
                if ((uint32_t)L->ndeferred > c.max_deferred) c.max_deferred = (uint32_t)L->ndeferred;
                if (L->ndeferred == 0 || L->done) { L->top = 0; continue; }
                L->ptr = 0; L->mem[0] = 0;
                for (int k = L->ndeferred - 1; k >= 1; k--) L->mem[++L->ptr] = L->deferred[k];
                L->top = L->deferred[0];
            }
            lane_loop(lanes, nodes, tris, any_hit, &c);
        }
        for (int l = 0; l < W; l++) {
            const int32_t i = first + l;
            if (i >= n) break;
            hits[i].tri_id = lanes[l].hit_id; hits[i].t = lanes[l].hit_t; hits[i].u = lanes[l].hit_u; hits[i].v = lanes[l].hit_v;
        }
        counts[chunk] = c;
    }
    return overflow;
}
