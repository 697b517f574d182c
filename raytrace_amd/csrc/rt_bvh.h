// Host-side BVH construction over the ordinary spheres.  Replaces the device-heap, randomly
// axis-split, insertion-sorted build of cuda/bvh.h:55-89 by a deterministic binned-SAH build
// that emits the linear 64-byte node array the traversal kernel reads (layout: rt_device.h).
#pragma once

#include <cstdint>
#include <vector>

namespace rt {

// A leaf code is ~(first << kLeafBits | count - 1): up to 2^kLeafBits spheres per leaf.
constexpr int kLeafBits = 3;
constexpr int kLeafCap = 1 << kLeafBits;

struct BvhResult {
    std::vector<float> nodes;     // 16 floats per inner node (child codes stored as int bits)
    std::vector<int32_t> order;   // leaf order: order[k] = index of the input sphere at slot k
    int32_t root = 0x7fffffff;    // inner index, leaf code, or kNoBvh when there are no spheres
    int32_t depth = 0;            // inner levels on the longest root-to-leaf path
};

// centers: n x 3 doubles, radii: n doubles.  max_depth bounds `depth` (traversal stack size).
// Ranges of at most leaf_max spheres become leaves: all of them when node_cost is 0, else where
// the surface-area heuristic says a leaf is cheaper than a split (node_cost = cost of an inner
// node visit in units of one sphere test).
BvhResult build_bvh(const double* centers, const double* radii, int64_t n, int max_depth, int leaf_max = 4, float node_cost = 0.0f);

// Four-wide form of a tree's 32-byte quantised binary nodes (rt_lbvh.h), for walks out of HBM: a
// node takes the place of up to three binary nodes and holds up to four children, so a ray makes
// half as many dependent node fetches.  64 bytes, sixteen dwords:
//   dwords 3 i + a (i = 0..3, a = x, y, z): child i's planes along a on the same 16-bit grid, lo | hi << 16
//   dwords 12 + i: child i's code, a four-wide node index (>= 0) or the binary tree's leaf code (< 0)
// The planes are those of the binary nodes, bit for bit, so a box holds what it held there.  An
// unused slot has lo = 65535, hi = 0 on every axis (no ray meets it) and `empty_leaf` as its code,
// a leaf nothing can hit, should arithmetic at the edge of the float range let one in after all.
struct Bvh4qResult {
    std::vector<uint32_t> nodes;
    int32_t root = 0x7fffffff;    // four-wide index, leaf code, or kNoBvh
    int32_t depth = 0;            // four-wide inner levels on the longest root-to-leaf path
    int32_t stack_need = 0;       // most entries a traversal stack holds at once: the sum of (children - 1) down the worst path
};

// qnodes2: the binary tree's eight-dword nodes; root2: its root code.  Greedy by surface area: the
// largest inner child is opened until the node has four children or only leaves.  With stack_cap > 0
// a child is only opened while the worst path through the node still fits a stack of that many
// entries (the node then stays two- or three-wide).
Bvh4qResult collapse_bvh4q(const uint32_t* qnodes2, size_t n_nodes2, int32_t root2, int32_t empty_leaf, int stack_cap = 0);

}  // namespace rt
