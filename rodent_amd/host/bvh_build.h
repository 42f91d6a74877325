// Host-side BVH construction for the scene tools (bvh_extractor, rodent).
//
// Own implementation of a spatial-split SAH builder (Stich et al. 2009) that
// produces N-ary trees by greedy in-node splitting.  It stands where the
// reference uses src/driver/bvh.h (SplitBvhBuilder) and follows the same cost
// model and stopping rules so that trees are comparable:
//   leaf cost = count * half_area, traversal cost = half_area   converter.cpp:120-127
//   stop splitting at <= leaf_threshold refs (2 for every layout) converter.cpp:144,288
//   spatial splits only if overlap half-area > alpha * root half-area, alpha 1e-5   bvh.h:106,160-166
//   accept a split iff split_cost + traversal_cost < leaf_cost    bvh.h:172-176
//   N-ary: keep splitting the untested child of largest cost until N children,
//   then order children by decreasing reference count             bvh.h:61-82,215
// It is a build step, not part of the GPU hot path.
#pragma once
#include <cstdint>
#include <vector>
#include "vec.h"
#include "../../include/rodent_traversal.h"

namespace rodent {

struct WideNode {
    int   count = 0;
    Box   box[8];
    int   child[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // >= 0: index into nodes; < 0: ~index into leaves
};
struct WideBvh {
    int arity = 2;
    std::vector<WideNode> nodes;                  // nodes[0] is the root
    std::vector<std::vector<uint32_t>> leaves;    // triangle ids per leaf
    size_t num_refs = 0, object_splits = 0, spatial_splits = 0;
    int depth = 0;
    float sah_cost = 0;                           // sum(area*1 per inner) + sum(area*count per leaf), / root area
};

struct BuildParams {
    int   arity = 2;
    int   leaf_threshold = 2;
    // cost of visiting an inner node, in units of one triangle test, times its half area (the reference: 1, converter.cpp:120-127)
    float traversal_cost = 1.0f;
    float alpha = 1e-5f;
    bool  spatial_splits = true;
    int   max_depth = 56;        // keeps the traversal stacks (64 entries, stack.impala:53) safe
    // host threads for inputs of 65 536 triangles and more (0 = all hardware threads); the result does not depend on it
    int   threads = 0;
};

WideBvh build_wide_bvh(const std::vector<Triangle>& tris, const BuildParams& p);

// Layout writers: geom_ids[i] is stored in Tri1::geom_id / Tri4::geom_id
// (material id, converter.cpp:246,374; pass nullptr for 0 like extract_bvh2.cpp:85-99).
void layout_bvh2_tri1(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
                      std::vector<Node2>& nodes, std::vector<Tri1>& out);
void layout_bvh4_tri4(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
                      std::vector<Node4>& nodes, std::vector<Tri4>& out);
void layout_bvh8_tri4(const WideBvh& bvh, const std::vector<Triangle>& tris, const uint32_t* geom_ids,
                      std::vector<Node8>& nodes, std::vector<Tri4>& out);

} // namespace rodent
