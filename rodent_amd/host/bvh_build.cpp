#include "bvh_build.h"

#include <atomic>
#include <cassert>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>

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



// A subtree as the serial builder numbers it: nodes in the order their tasks are taken (pre-order), leaves in the order they are made.
struct Piece {
    std::vector<WideNode> nodes;
    std::vector<std::vector<uint32_t>> leaves;
    size_t num_refs = 0, object_splits = 0, spatial_splits = 0;
    int depth = 0;
    bool root_is_leaf = false;                 // the piece's root did not split: leaves[0] is the piece
};

// One task of the build (bvh.h:61-82): candidate `c` is split greedily -- always the untested child of largest cost -- into up to
// p.arity children, returned in order of decreasing reference count; kids[i].tested = "this child becomes a leaf".
std::vector<Cand> expand(Cand&& c, int level, const std::vector<Triangle>& tris, const BuildParams& p, float spatial_threshold,
                         std::vector<Box>& scratch, Piece& stats) {
    const int N = p.arity;
    std::vector<Cand> kids; kids.reserve(N);
    kids.push_back(std::move(c));
    if (level >= p.max_depth) kids[0].tested = true;
    while ((int)kids.size() < N) {
        int pick = -1;
        for (int i = 0; i < (int)kids.size(); i++)
            if (!kids[i].tested && (pick < 0 || kids[i].cost > kids[pick].cost)) pick = i;
        if (pick < 0) break;
        Cand& k = kids[pick];
        if (k.refs.size() <= (size_t)p.leaf_threshold) { k.tested = true; continue; }

        SplitChoice s;
        if (k.refs.size() <= kSweepLimit) find_object_split_sweep(k, s, scratch);
        else find_object_split_binned(k, s);
        if (!s.valid && k.refs.size() > kSweepLimit) find_object_split_sweep(k, s, scratch);  // all centroids equal
        const SplitChoice object_choice = s;
        if (p.spatial_splits && s.valid) {
            Box ov = s.lbox; ov.clip(s.rbox);
            if (!ov.empty() && ov.half_area() > spatial_threshold) find_spatial_split(k, tris, s);
        }
        if (!s.valid || s.cost + p.traversal_cost * k.bb.half_area() >= k.cost) { k.tested = true; continue; }

        Cand l, r;
        bool ok = apply_split(k, s, tris, l, r);
        if (!ok && s.spatial) { l = Cand(); r = Cand(); ok = apply_split(k, object_choice, tris, l, r); if (ok) stats.object_splits++; }
        else if (ok) (s.spatial ? stats.spatial_splits : stats.object_splits)++;
        if (!ok) { k.tested = true; continue; }
        kids[pick] = std::move(l);
        kids.push_back(std::move(r));
    }
    if (kids.size() > 1) std::stable_sort(kids.begin(), kids.end(),
        [](const Cand& a, const Cand& b) { return a.refs.size() > b.refs.size(); });
    return kids;
}

std::vector<uint32_t> leaf_ids(const Cand& c) {
    std::vector<uint32_t> ids(c.refs.size());
    for (size_t i = 0; i < ids.size(); i++) ids[i] = c.refs[i].id;
    return ids;
}

// The serial build of one subtree: a stack of tasks, depth first, child 0 first.
void build_piece(Cand&& root, int root_level, const std::vector<Triangle>& tris, const BuildParams& p, float spatial_threshold,
    Piece& out) {
    struct Task { Cand c; int parent, slot, level; };
    std::vector<Task> stack;
    stack.push_back({std::move(root), -1, 0, root_level});
    std::vector<Box> scratch;
    auto make_leaf = [&](const Cand& c) {
        out.num_refs += c.refs.size();
        out.leaves.push_back(leaf_ids(c));
        return ~(int)(out.leaves.size() - 1);
    };
    while (!stack.empty()) {
        Task t = std::move(stack.back()); stack.pop_back();
        out.depth = std::max(out.depth, t.level);
        std::vector<Cand> kids = expand(std::move(t.c), t.level, tris, p, spatial_threshold, scratch, out);
        if (kids.size() == 1) {
            const int leaf = make_leaf(kids[0]);
            if (t.parent < 0) out.root_is_leaf = true;
            else out.nodes[t.parent].child[t.slot] = leaf;
            continue;
        }
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
}

} // namespace

// Large inputs are built on several host threads WITHOUT changing the result: the serial builder works through a subtree completely before
// it touches the next one, so a subtree's nodes and leaves are contiguous runs of its output.  The top of the tree (tasks above
// `piece_limit` references) is expanded first, in the serial order, into a list of events -- node, leaf, or "a subtree goes here" --; the
// subtrees are built independently, each numbering its nodes and leaves from zero; then the events are replayed and every subtree is
// spliced in with its offsets.  Same nodes, same order, same leaves as one thread would produce (tests/test_builder.py).
WideBvh build_wide_bvh(const std::vector<Triangle>& tris, const BuildParams& p) {
    WideBvh out; out.arity = p.arity;
    assert(p.arity >= 2 && p.arity <= 8);

    Cand root;
    root.refs.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        Box b; b.grow(tris[i].v0); b.grow(tris[i].v1); b.grow(tris[i].v2);
        root.refs[i] = {(uint32_t)i, b}; root.bb.grow(b);
    }
    root.finish();
    const float spatial_threshold = root.bb.half_area() * p.alpha;
    const Box root_box = root.bb;

    int threads = p.threads > 0 ? p.threads : (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    constexpr size_t kParallelMinRefs = 65536, kMinPieceRefs = 8192;
    Piece whole;
    if (threads == 1 || tris.size() < kParallelMinRefs) {
        build_piece(std::move(root), 0, tris, p, spatial_threshold, whole);
        if (whole.root_is_leaf) {                  // a single-leaf scene still gets a root node (bvh.h:218-224)
            WideNode n; n.count = 1; n.box[0] = root_box; n.child[0] = ~0;
            whole.nodes.push_back(n);
        }
    } else {
        const size_t piece_limit = std::max(kMinPieceRefs, tris.size() / (size_t)(8 * threads));
        // kind 0: node, 1: leaf, 2: piece; parent = index of the parent's node EVENT
        struct Event { int kind; int parent, slot; WideNode node; std::vector<uint32_t> ids; int piece; };
        std::vector<Event> events;
        struct PieceIn { Cand c; int level; };
        std::vector<PieceIn> todo;
        {
            struct Task { Cand c; int parent, slot, level; };
            std::vector<Task> stack;
            stack.push_back({std::move(root), -1, 0, 0});
            std::vector<Box> scratch;
            while (!stack.empty()) {
                Task t = std::move(stack.back()); stack.pop_back();
                if (t.parent >= 0 && t.c.refs.size() <= piece_limit) {
                    events.push_back({2, t.parent, t.slot, WideNode(), {}, (int)todo.size()});
                    todo.push_back({std::move(t.c), t.level});
                    continue;
                }
                whole.depth = std::max(whole.depth, t.level);
                std::vector<Cand> kids = expand(std::move(t.c), t.level, tris, p, spatial_threshold, scratch, whole);
                if (kids.size() == 1) {                // (cannot be the root: it holds >= kParallelMinRefs references and would have split)
                    events.push_back({1, t.parent, t.slot, WideNode(), leaf_ids(kids[0]), -1});
                    continue;
                }
                const int me = (int)events.size();
                Event ev{0, t.parent, t.slot, WideNode(), {}, -1};
                ev.node.count = (int)kids.size();
                for (int i = 0; i < (int)kids.size(); i++) ev.node.box[i] = kids[i].bb;
                events.push_back(std::move(ev));
                for (int i = (int)kids.size() - 1; i >= 0; i--) {
                    if (kids[i].tested) events.push_back({1, me, i, WideNode(), leaf_ids(kids[i]), -1});
                    else stack.push_back({std::move(kids[i]), me, i, t.level + 1});
                }
            }
        }
        std::vector<Piece> pieces(todo.size());
        {
            // largest first: the pieces differ in size by what the top's splits left
            std::vector<int> order(todo.size());
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return todo[a].c.refs.size() > todo[b].c.refs.size(); });
            std::atomic<size_t> next{0};
            auto worker = [&] {
                for (size_t k; (k = next.fetch_add(1)) < order.size();) {
                    const int i = order[k];
                    build_piece(std::move(todo[i].c), todo[i].level, tris, p, spatial_threshold, pieces[i]);
                }
            };
            std::vector<std::thread> pool;
            for (int k = 1; k < std::min<int>(threads, (int)order.size()); k++) pool.emplace_back(worker);
            worker();
            for (auto& th : pool) th.join();
        }
        // replay: the serial numbering
        std::vector<int> node_of_event(events.size(), -1);
        for (size_t e = 0; e < events.size(); e++) {
            Event& ev = events[e];
            int child;
            if (ev.kind == 0) {
                child = (int)whole.nodes.size();
                node_of_event[e] = child;
                whole.nodes.push_back(ev.node);
            } else if (ev.kind == 1) {
                whole.num_refs += ev.ids.size();
                whole.leaves.push_back(std::move(ev.ids));
                child = ~(int)(whole.leaves.size() - 1);
            } else {
                Piece& pc = pieces[ev.piece];
                const int node_off = (int)whole.nodes.size(), leaf_off = (int)whole.leaves.size();
                child = pc.root_is_leaf ? ~leaf_off : node_off;
                for (WideNode n : pc.nodes) {
                    for (int i = 0; i < n.count; i++) n.child[i] = n.child[i] >= 0 ? n.child[i] + node_off : ~(~n.child[i] + leaf_off);
                    whole.nodes.push_back(n);
                }
                for (auto& l : pc.leaves) whole.leaves.push_back(std::move(l));
                whole.num_refs += pc.num_refs; whole.object_splits += pc.object_splits; whole.spatial_splits += pc.spatial_splits;
                whole.depth = std::max(whole.depth, pc.depth);
            }
            if (ev.parent >= 0) whole.nodes[node_of_event[ev.parent]].child[ev.slot] = child;
        }
    }
    out.nodes = std::move(whole.nodes); out.leaves = std::move(whole.leaves);
    out.num_refs = whole.num_refs; out.object_splits = whole.object_splits; out.spatial_splits = whole.spatial_splits;
    out.depth = whole.depth;

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
