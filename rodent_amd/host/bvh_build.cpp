#include "bvh_build.h"

#include <cassert>
#include <cstring>
#include <limits>
#include <numeric>

namespace rodent {
namespace {

struct Ref { uint32_t id; Box bb; };

struct Cand {
    std::vector<Ref> refs;
    Box   bb;
    float cost = 0;
    bool  tested = false;
    void finish() { cost = (float)refs.size() * bb.half_area(); }
};

inline float centroid(const Ref& r, int a) { return 0.5f * (r.bb.lo[a] + r.bb.hi[a]); }

// Bounding box of triangle ∩ {lo <= x[axis] <= hi} (Sutherland-Hodgman on two planes).
Box clipped_box(const Triangle& t, int axis, float lo, float hi) {
    V3 a[10], b[10];
    int na = 3; a[0] = t.v0; a[1] = t.v1; a[2] = t.v2;
    for (int pass = 0; pass < 2; pass++) {
        const float plane = pass == 0 ? lo : hi;
        const float sgn = pass == 0 ? 1.0f : -1.0f;          // keep sgn*(x - plane) >= 0
        int nb = 0;
        for (int i = 0; i < na; i++) {
            const V3 p = a[i], q = a[(i + 1) % na];
            const float dp = sgn * (p[axis] - plane), dq = sgn * (q[axis] - plane);
            if (dp >= 0) b[nb++] = p;
            if ((dp > 0 && dq < 0) || (dp < 0 && dq > 0)) {
                const float s = dp / (dp - dq);
                V3 x = p + (q - p) * s;
                x[axis] = plane;
                b[nb++] = x;
            }
        }
        na = nb;
        std::memcpy(a, b, sizeof(V3) * nb);
        if (na == 0) break;
    }
    Box r;
    for (int i = 0; i < na; i++) r.grow(a[i]);
    return r;
}

struct SplitChoice {
    bool  valid = false, spatial = false;
    int   axis = 0;
    float cost = std::numeric_limits<float>::max();
    size_t left_count = 0;      // object split (sweep): refs sorted on axis, first left_count go left
    int   bin = -1;             // object split (binned): centroid bin <= bin goes left
    float cmin = 0, cscale = 0; // binned: bin = (c - cmin) * cscale
    float plane = 0;            // spatial split position
    Box   lbox, rbox;
};

constexpr int    kObjBins = 64;
constexpr int    kSpatialBins = 64;
constexpr size_t kSweepLimit = 4096;

void sort_axis(std::vector<Ref>& refs, int axis) {
    std::sort(refs.begin(), refs.end(), [axis](const Ref& a, const Ref& b) {
        const float ca = centroid(a, axis), cb = centroid(b, axis);
        return ca < cb || (ca == cb && a.id < b.id);
    });
}

void find_object_split_sweep(Cand& c, SplitChoice& best, std::vector<Box>& scratch) {
    const size_t n = c.refs.size();
    scratch.resize(n);
    for (int axis = 0; axis < 3; axis++) {
        sort_axis(c.refs, axis);
        Box acc;
        for (size_t i = n - 1; i > 0; i--) { acc.grow(c.refs[i].bb); scratch[i - 1] = acc; }
        acc = Box();
        for (size_t i = 0; i + 1 < n; i++) {
            acc.grow(c.refs[i].bb);
            const float cost = (float)(i + 1) * acc.half_area() + (float)(n - i - 1) * scratch[i].half_area();
            if (cost < best.cost) {
                best.valid = true; best.spatial = false; best.bin = -1; best.axis = axis; best.cost = cost;
                best.left_count = i + 1; best.lbox = acc; best.rbox = scratch[i];
            }
        }
    }
}

void find_object_split_binned(Cand& c, SplitChoice& best) {
    Box cb;
    for (auto& r : c.refs) cb.grow(V3(centroid(r, 0), centroid(r, 1), centroid(r, 2)));
    for (int axis = 0; axis < 3; axis++) {
        const float ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0)) continue;
        const float scale = (float)kObjBins / ext;
        Box bb[kObjBins]; size_t cnt[kObjBins] = {};
        for (auto& r : c.refs) {
            const int b = std::min(kObjBins - 1, std::max(0, (int)((centroid(r, axis) - cb.lo[axis]) * scale)));
            bb[b].grow(r.bb); cnt[b]++;
        }
        Box right[kObjBins]; Box acc;
        for (int i = kObjBins - 1; i > 0; i--) { acc.grow(bb[i]); right[i - 1] = acc; }
        acc = Box(); size_t nl = 0;
        for (int i = 0; i + 1 < kObjBins; i++) {
            acc.grow(bb[i]); nl += cnt[i];
            const size_t nr = c.refs.size() - nl;
            if (nl == 0 || nr == 0) continue;
            const float cost = (float)nl * acc.half_area() + (float)nr * right[i].half_area();
            if (cost < best.cost) {
                best.valid = true; best.spatial = false; best.axis = axis; best.cost = cost; best.bin = i;
                best.cmin = cb.lo[axis]; best.cscale = scale; best.left_count = nl;
                best.lbox = acc; best.rbox = right[i];
            }
        }
    }
}

void find_spatial_split(const Cand& c, const std::vector<Triangle>& tris, SplitChoice& best) {
    for (int axis = 0; axis < 3; axis++) {
        const float lo = c.bb.lo[axis], hi = c.bb.hi[axis];
        if (!(hi > lo)) continue;
        const float width = (hi - lo) / kSpatialBins, inv = 1.0f / width;
        Box bb[kSpatialBins]; size_t entry[kSpatialBins] = {}, exits[kSpatialBins] = {};
        auto plane_at = [&](int i) { return i >= kSpatialBins ? hi : lo + i * width; };
        for (auto& r : c.refs) {
            const int b0 = std::min(kSpatialBins - 1, std::max(0, (int)((r.bb.lo[axis] - lo) * inv)));
            const int b1 = std::min(kSpatialBins - 1, std::max(b0, (int)((r.bb.hi[axis] - lo) * inv)));
            if (b0 == b1) bb[b0].grow(r.bb);
            else for (int b = b0; b <= b1; b++) {
                Box cl = clipped_box(tris[r.id], axis, plane_at(b), plane_at(b + 1));
                cl.clip(r.bb);
                if (!cl.empty()) bb[b].grow(cl);
            }
            entry[b0]++; exits[b1]++;
        }
        Box right[kSpatialBins]; Box acc;
        for (int i = kSpatialBins - 1; i > 0; i--) { acc.grow(bb[i]); right[i - 1] = acc; }
        acc = Box(); size_t nl = 0, nr = c.refs.size();
        for (int i = 0; i + 1 < kSpatialBins; i++) {
            acc.grow(bb[i]); nl += entry[i]; nr -= exits[i];
            if (nl == 0 || nr == 0 || nl == c.refs.size() || nr == c.refs.size()) continue;
            const float cost = (float)nl * acc.half_area() + (float)nr * right[i].half_area();
            if (cost < best.cost) {
                best.valid = true; best.spatial = true; best.axis = axis; best.cost = cost; best.plane = plane_at(i + 1);
            }
        }
    }
}

// Returns false if the split degenerates (one side empty); l and r are filled otherwise.
bool apply_split(Cand& c, const SplitChoice& s, const std::vector<Triangle>& tris, Cand& l, Cand& r) {
    if (!s.spatial) {
        if (s.bin < 0) {
            sort_axis(c.refs, s.axis);
            l.refs.assign(c.refs.begin(), c.refs.begin() + s.left_count);
            r.refs.assign(c.refs.begin() + s.left_count, c.refs.end());
        } else {
            for (auto& ref : c.refs) {
                const int b = std::min(kObjBins - 1, std::max(0, (int)((centroid(ref, s.axis) - s.cmin) * s.cscale)));
                (b <= s.bin ? l : r).refs.push_back(ref);
            }
        }
        l.bb = s.lbox; r.bb = s.rbox;
    } else {
        const int a = s.axis; const float p = s.plane;
        std::vector<Ref> mid;
        for (auto& ref : c.refs) {
            if (ref.bb.hi[a] <= p)      { l.refs.push_back(ref); l.bb.grow(ref.bb); }
            else if (ref.bb.lo[a] >= p) { r.refs.push_back(ref); r.bb.grow(ref.bb); }
            else mid.push_back(ref);
        }
        for (auto& ref : mid) {
            Box lb = clipped_box(tris[ref.id], a, -FLT_MAX, p); lb.clip(ref.bb);
            Box rb = clipped_box(tris[ref.id], a, p, FLT_MAX);  rb.clip(ref.bb);
            Box lu = l.bb; lu.grow(ref.bb);                    // unsplit to the left
            Box ru = r.bb; ru.grow(ref.bb);                    // unsplit to the right
            Box ld = l.bb; ld.grow(lb);
            Box rd = r.bb; rd.grow(rb);
            const float nl = (float)l.refs.size(), nr = (float)r.refs.size();
            const float c_l = (nl + 1) * lu.half_area() + nr * r.bb.half_area();
            const float c_r = nl * l.bb.half_area() + (nr + 1) * ru.half_area();
            const float c_d = (nl + 1) * ld.half_area() + (nr + 1) * rd.half_area();
            const bool can_dup = !lb.empty() && !rb.empty();
            if (can_dup && c_d < c_l && c_d < c_r) {
                l.refs.push_back({ref.id, lb}); l.bb = ld;
                r.refs.push_back({ref.id, rb}); r.bb = rd;
            } else if (c_l <= c_r) { l.refs.push_back(ref); l.bb = lu; }
            else                   { r.refs.push_back(ref); r.bb = ru; }
        }
    }
    if (l.refs.empty() || r.refs.empty()) return false;
    l.finish(); r.finish();
    return true;
}

} // namespace

WideBvh build_wide_bvh(const std::vector<Triangle>& tris, const BuildParams& p) {
    WideBvh out; out.arity = p.arity;
    const int N = p.arity;
    assert(N >= 2 && N <= 8);

    Cand root;
    root.refs.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        Box b; b.grow(tris[i].v0); b.grow(tris[i].v1); b.grow(tris[i].v2);
        root.refs[i] = {(uint32_t)i, b}; root.bb.grow(b);
    }
    root.finish();
    const float spatial_threshold = root.bb.half_area() * p.alpha;

    struct Task { Cand c; int parent, slot, level; };
    std::vector<Task> stack;
    stack.push_back({std::move(root), -1, 0, 0});
    std::vector<Box> scratch;

    auto make_leaf = [&](const Cand& c) {
        std::vector<uint32_t> ids(c.refs.size());
        for (size_t i = 0; i < ids.size(); i++) ids[i] = c.refs[i].id;
        out.num_refs += ids.size();
        out.leaves.push_back(std::move(ids));
        return ~(int)(out.leaves.size() - 1);
    };

    while (!stack.empty()) {
        Task t = std::move(stack.back()); stack.pop_back();
        out.depth = std::max(out.depth, t.level);
        std::vector<Cand> kids; kids.reserve(N);
        kids.push_back(std::move(t.c));
        if (t.level >= p.max_depth) kids[0].tested = true;

        while ((int)kids.size() < N) {
            int pick = -1;
            for (int i = 0; i < (int)kids.size(); i++)
                if (!kids[i].tested && (pick < 0 || kids[i].cost > kids[pick].cost)) pick = i;
            if (pick < 0) break;
            Cand& c = kids[pick];
            if (c.refs.size() <= (size_t)p.leaf_threshold) { c.tested = true; continue; }

            SplitChoice s;
            if (c.refs.size() <= kSweepLimit) find_object_split_sweep(c, s, scratch);
            else find_object_split_binned(c, s);
            if (!s.valid && c.refs.size() > kSweepLimit) find_object_split_sweep(c, s, scratch);  // all centroids equal
            const SplitChoice object_choice = s;
            if (p.spatial_splits && s.valid) {
                Box ov = s.lbox; ov.clip(s.rbox);
                if (!ov.empty() && ov.half_area() > spatial_threshold) find_spatial_split(c, tris, s);
            }
            if (!s.valid || s.cost + p.traversal_cost * c.bb.half_area() >= c.cost) { c.tested = true; continue; }

            Cand l, r;
            bool ok = apply_split(c, s, tris, l, r);
            if (!ok && s.spatial) { l = Cand(); r = Cand(); ok = apply_split(c, object_choice, tris, l, r); if (ok) out.object_splits++; }
            else if (ok) (s.spatial ? out.spatial_splits : out.object_splits)++;
            if (!ok) { c.tested = true; continue; }
            kids[pick] = std::move(l);
            kids.push_back(std::move(r));
        }

        if (kids.size() == 1) {
            const int leaf = make_leaf(kids[0]);
            if (t.parent < 0) {                       // a single-leaf scene still gets a root node (bvh.h:218-224)
                WideNode n; n.count = 1; n.box[0] = kids[0].bb; n.child[0] = leaf;
                out.nodes.push_back(n);
            } else out.nodes[t.parent].child[t.slot] = leaf;
            continue;
        }

        std::stable_sort(kids.begin(), kids.end(), [](const Cand& a, const Cand& b) { return a.refs.size() > b.refs.size(); });
        const int me = (int)out.nodes.size();
        out.nodes.emplace_back();
        if (t.parent >= 0) out.nodes[t.parent].child[t.slot] = me;
        out.nodes[me].count = (int)kids.size();
        for (int i = 0; i < (int)kids.size(); i++) out.nodes[me].box[i] = kids[i].bb;
        for (int i = (int)kids.size() - 1; i >= 0; i--) {
            if (kids[i].tested) out.nodes[me].child[i] = make_leaf(kids[i]);
            else stack.push_back({std::move(kids[i]), me, i, t.level + 1});
        }
    }

    // SAH cost relative to the root area (traversal 1 per inner node, 1 per referenced triangle)
    Box rb; for (int i = 0; i < out.nodes[0].count; i++) rb.grow(out.nodes[0].box[i]);
    double cost = rb.half_area();
    for (auto& n : out.nodes)
        for (int i = 0; i < n.count; i++)
            cost += n.child[i] >= 0 ? n.box[i].half_area() : n.box[i].half_area() * (double)out.leaves[~n.child[i]].size();
    out.sah_cost = (float)(cost / std::max(rb.half_area(), 1e-30f));
    return out;
}

// ---------------------------------------------------------------------------------------------

namespace {

const float kInf = std::numeric_limits<float>::infinity();

template <typename NodeT, int N>
void fill_wide_nodes(const WideBvh& bvh, const std::vector<size_t>& leaf_first, std::vector<NodeT>& nodes) {
    nodes.resize(bvh.nodes.size());
    for (size_t i = 0; i < bvh.nodes.size(); i++) {
        const WideNode& w = bvh.nodes[i];
        NodeT& n = nodes[i];
        std::memset(&n, 0, sizeof n);
        for (int j = 0; j < N; j++) {
            if (j < w.count) {
                n.bounds[0][j] = w.box[j].lo.x; n.bounds[1][j] = w.box[j].hi.x;
                n.bounds[2][j] = w.box[j].lo.y; n.bounds[3][j] = w.box[j].hi.y;
                n.bounds[4][j] = w.box[j].lo.z; n.bounds[5][j] = w.box[j].hi.z;
                n.child[j] = w.child[j] >= 0 ? w.child[j] + 1 : ~(int32_t)leaf_first[~w.child[j]];
            } else {                                 // empty slot (converter.cpp:185-195)
                n.bounds[0][j] = n.bounds[2][j] = n.bounds[4][j] = kInf;
                n.bounds[1][j] = n.bounds[3][j] = n.bounds[5][j] = -kInf;
                n.child[j] = 0;
            }
        }
    }
}

void pack_tri4(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
               std::vector<size_t>& leaf_first, std::vector<Tri4>& out) {
    leaf_first.resize(bvh.leaves.size());
    out.clear();
    for (size_t l = 0; l < bvh.leaves.size(); l++) {
        leaf_first[l] = out.size();
        const auto& ids = bvh.leaves[l];
        for (size_t i = 0; i < ids.size(); i += 4) {                 // converter.cpp:219-257
            Tri4 t; std::memset(&t, 0, sizeof t);
            const size_t c = std::min<size_t>(4, ids.size() - i);
            for (size_t j = 0; j < c; j++) {
                const uint32_t id = ids[i + j];
                const Triangle& in = tris[id];
                const V3 e1 = in.v0 - in.v1, e2 = in.v2 - in.v0, n = cross(e1, e2);
                for (int k = 0; k < 3; k++) { t.v0[k][j] = in.v0[k]; t.e1[k][j] = e1[k]; t.e2[k][j] = e2[k]; t.n[k][j] = n[k]; }
                t.prim_id[j] = (int32_t)id;
                t.geom_id[j] = geom_ids ? (int32_t)geom_ids[id] : 0;
            }
            for (size_t j = c; j < 4; j++) t.prim_id[j] = -1;
            out.push_back(t);
        }
        if (!bvh.leaves[l].empty()) out.back().prim_id[3] |= (int32_t)0x80000000u;
    }
}

} // namespace

void layout_bvh2_tri1(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
                      std::vector<Node2>& nodes, std::vector<Tri1>& out) {
    assert(bvh.arity == 2);
    std::vector<size_t> leaf_first(bvh.leaves.size());
    out.clear();
    for (size_t l = 0; l < bvh.leaves.size(); l++) {
        leaf_first[l] = out.size();
        for (uint32_t id : bvh.leaves[l]) {                          // converter.cpp:365-380
            const Triangle& in = tris[id];
            const V3 e1 = in.v0 - in.v1, e2 = in.v2 - in.v0;
            Tri1 t;
            t.v0[0] = in.v0.x; t.v0[1] = in.v0.y; t.v0[2] = in.v0.z; t.pad = 0;
            t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z; t.geom_id = geom_ids ? (int32_t)geom_ids[id] : 0;
            t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z; t.prim_id = (int32_t)id;
            out.push_back(t);
        }
        if (!bvh.leaves[l].empty()) out.back().prim_id |= (int32_t)0x80000000u;
    }
    nodes.resize(bvh.nodes.size());
    for (size_t i = 0; i < bvh.nodes.size(); i++) {
        const WideNode& w = bvh.nodes[i];
        Node2& n = nodes[i];
        std::memset(&n, 0, sizeof n);
        for (int j = 0; j < 2; j++) {
            float* b = n.bounds + 6 * j;
            if (j < w.count) {
                b[0] = w.box[j].lo.x; b[1] = w.box[j].hi.x; b[2] = w.box[j].lo.y;
                b[3] = w.box[j].hi.y; b[4] = w.box[j].lo.z; b[5] = w.box[j].hi.z;
                n.child[j] = w.child[j] >= 0 ? w.child[j] + 1 : ~(int32_t)leaf_first[~w.child[j]];
            } else { b[0] = b[2] = b[4] = kInf; b[1] = b[3] = b[5] = -kInf; n.child[j] = 0; }
        }
    }
}

void layout_bvh4_tri4(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
                      std::vector<Node4>& nodes, std::vector<Tri4>& out) {
    assert(bvh.arity == 4);
    std::vector<size_t> leaf_first;
    pack_tri4(bvh, tris, geom_ids, leaf_first, out);
    fill_wide_nodes<Node4, 4>(bvh, leaf_first, nodes);
}

void layout_bvh8_tri4(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
                      std::vector<Node8>& nodes, std::vector<Tri4>& out) {
    assert(bvh.arity == 8);
    std::vector<size_t> leaf_first;
    pack_tri4(bvh, tris, geom_ids, leaf_first, out);
    fill_wide_nodes<Node8, 8>(bvh, leaf_first, nodes);
}

} // namespace rodent
