// BVH construction on the GPU: Morton-ordered radix tree (one kernel per phase, rocPRIM for the
// sort and the scan) emitting the same 64-byte nodes and leaf-ordered sphere arrays as the host
// builder of rt_bvh.h.  Takes the place of the device-side recursive build of cuda/bvh.h:55-89
// where a scene is rebuilt often or is too large for a host build per frame.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct LbvhResult {
    float4* nodes = nullptr;    // device, n_nodes x 4 float4; owned by the caller (hipFree)
    int* order = nullptr;       // device, n ints: order[k] = input index of the sphere at leaf slot k; owned by the caller
    int n_nodes = 0;
    int root = 0;               // inner index 0 (the builder is only used for n > 4)
    int depth = 0;              // inner levels on the longest root-to-leaf path
};

// sph64: n spheres (cx, cy, cz, r) in input order, device memory; n > 4.  Subtrees of at most
// leaf_max (1 .. 4) spheres become leaves.  Synchronises `stream` twice (node count, depth).  On
// failure nothing is left allocated.
hipError_t build_lbvh(const double4* sph64, int n, int leaf_max, hipStream_t stream, LbvhResult* out);

// Leaf-order copies of the per-sphere arrays: dst[k] = src[order[k]] for k < n.  Entries
// n .. n_padded-1 of the destinations must have been filled by the caller.
hipError_t gather_spheres(const int* order, int n, const float4* sph, const double4* sph64, const int2* info,
                          const int* xform, float4* sph_out, double4* sph64_out, int2* info_out, int* xform_out,
                          hipStream_t stream);

// 32-byte form of the binary nodes for trees walked out of HBM: each of the twelve box planes of a
// node as a 16-bit step count from the low corner of the scene's box (`lo`, steps of `step` per
// axis), the low planes rounded down and the high planes up, so a quantised box contains its box;
// then the two child codes.  Per node two 16-byte loads instead of four.
//   dwords 0-2: left  box x, y, z as (lo | hi << 16)    dwords 3-5: right box    dwords 6, 7: left, right
// *n_coarse: the number of nodes for which the grid is too coarse, that is, blows the surface area
// of either box up to more than twice its size (fine detail in a wide scene).  Such a node would be
// visited, and descended from, that much more often: a tree that has any is better walked in its
// float form.  Synchronises `stream`.
hipError_t quantize_nodes(const float4* nodes, int n_nodes, const float lo[3], const float step[3], uint4* out, int* n_coarse,
                          hipStream_t stream);

}  // namespace rt
