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

// sph64: n spheres (cx, cy, cz, r) in input order, device memory; n > 4.  Synchronises `stream`
// twice (node count, depth).  On failure nothing is left allocated.
hipError_t build_lbvh(const double4* sph64, int n, hipStream_t stream, LbvhResult* out);

// Leaf-order copies of the per-sphere arrays: dst[k] = src[order[k]] for k < n.  Entries
// n .. n_padded-1 of the destinations must have been filled by the caller.
hipError_t gather_spheres(const int* order, int n, const float4* sph, const double4* sph64, const int2* info,
                          const int* xform, float4* sph_out, double4* sph64_out, int2* info_out, int* xform_out,
                          hipStream_t stream);

}  // namespace rt
