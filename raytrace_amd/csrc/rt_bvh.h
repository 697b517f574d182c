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

}  // namespace rt
