// Device-side records of the flattened scene (all in HBM) and the launch interface between
// rt_api.cpp (host) and rt_kernels.hip (gfx950 kernels).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

// ---- scene records ----------------------------------------------------------------------

// Ordinary sphere, single precision, in BVH leaf order: (cx, cy, cz, r).            16 B
// The same sphere in double for the polish of the winning hit: (cx, cy, cz, r).     32 B
// Its list position and material: (obj, mat).                                        8 B

struct DWide {            // sphere too large for the single-precision test; always tested
    double cx, cy, cz, r; // centre already in world space
    double r2;            // r * r
    int obj, mat;
    int xform;            // 0, or 1 + index of the instance transform (only turns the texture coordinates)
    int pad1;
    double pad2;
};

// Instance transform (cuda/sceneobject.h:31-109 plus a translation):
// world = R_y * object + t, with R_y = [c 0 s; 0 1 0; -s 0 c].
// Kept in double: the ray is taken into the instance's frame in double, so that the distance to a
// rect's plane is a difference of nearby numbers formed before rounding to float.
struct DXform {
    double s, c;
    double tx, ty, tz;
    double pad;
};

struct DRect {            // cuda/rect.h: plane axis `ka` at k, in-plane axes (aa, ba)
    float a0, a1, b0, b1;
    float k;
    int axes;             // ka | aa << 2 | ba << 4 | (k does not fit a float) << 6 | (1 + instance transform, or 0) << 8
    int obj, mat;
};

struct DMaterial {        // tagged union: material.h:15-92, cuda/material.h:33-47,103-119
    int type, tex;
    float r, g, b;        // albedo / emitted colour when tex < 0; dielectric: 1/ir, Schlick r0 seen from outside, from inside
    float fuzz, ir, pad;
};

struct DTexture {         // tagged union: cuda/texture.h:17-78
    int type;
    int a, b;             // checker: even, odd texture | image: width, height
    uint32_t off;         // image: byte offset into the texel blob
    float r, g, bl, pad;  // solid colour
};

// BVH2 node, 64 B = four float4 loads.  Both children's boxes live in the parent so one
// fetch decides both.  child >= 0: inner node index; child < 0: leaf, ~child = first << kLeafBits |
// (count - 1) into the leaf-ordered sphere arrays (kLeafBits = 3 in rt_bvh.h; leaves hold at most four).
//   q0 = (lmin.x lmin.y lmin.z lmax.x)  q1 = (lmax.y lmax.z rmin.x rmin.y)
//   q2 = (rmin.z rmax.x rmax.y rmax.z)  q3 = (left, right, -, -) as int bits
struct DScene {
    const float4* sph;
    const double4* sph64;
    const int2* sph_info;
    int n_sph;
    int n_sph_padded;     // n_sph rounded up to a multiple of 4: the linear scan's extent (the padding cannot be hit);
                          // the arrays hold four more records, for leaves that are read four at a time
    const float4* nodes;
    const uint4* qnodes;  // the same nodes in 32 bytes (16-bit planes on the grid q_lo + k * q_step), or null when the grid is too coarse for some node: what frames walk out of HBM
    float q_lo[3], q_step[3];
    float tree_reach[3];  // largest |coordinate| of the tree's box planes, per axis (the slab test's rounding margins)
    int n_nodes;
    int root;             // inner index, leaf code, or kNoBvh
    int bvh_depth;
    int use_bvh;          // 0: scan all ordinary spheres in order
    int ids_fit_16;       // every list position and material index is below 65536
    int has_textures;     // bit 0: some material reads a texture; bit 1: some rect, or some sphere whose material reads a
                          // texture, stands under an instance transform; bit 2: some material emits (one word: the order and number of the fields
                          // of a kernel-argument struct move the register allocation of every kernel)
    const DWide* wide;
    int n_wide;
    const DRect* rects;
    const double* rect_k; // per rect, the plane coordinate as given (read where the float one does not do:
                          // instances, and planes that do not fit a float)
    int n_rects;
    const DXform* xforms;
    const int* sph_xform; // per ordinary sphere (leaf order): 0 or 1 + transform; null when no sphere has one
    const DMaterial* mats;
    const DTexture* texs;
    const uint8_t* texels;
    int background;
    float bg[3];
    int* stack_spill;     // deep trees: per lane of a full launch, the traversal stack entries beyond the LDS stack (see kLdsStackCap), or null
    int stack_spill_entries;  // entries per lane in stack_spill
};

struct DCamera {          // camera.h:8-16 after the constructor, single precision
    float origin[3];
    float llc_rel[3];     // lower_left_corner - origin, subtracted in double on the host
    float horizontal[3];
    float vertical[3];
    float u[3], v[3];
    float lens_radius;
};

struct DRender {
    int width, height;
    int spp, sample_begin, max_depth;
    uint32_t seed_lo, seed_hi;
    int tile_rows, tile_first, tile_step;   // normalised: tile_rows >= 1, tile_step >= 1
    int rows;                               // rows this launch renders
    float t_min;
    uint32_t flags;
    int lanes_per_pixel;                    // K: power of two, 1..64; a wave's tile holds 64/K pixels
    float inv_wm1, inv_hm1;                 // 1/(W-1), 1/(H-1)  (main.cc:117-118)
    int stack_depth;                        // size of the LDS traversal stack in units of kBlock ints (tree depth + 2 entries per lane; 16-bit entries with the tree in LDS)
    int slice;                              // low 16 bits: node steps a wave spends on traversal per trip (0 = no limit, 0xffff = choose at launch); above them: exit share in 128ths (0 = none, 255 = the default of the walk that is launched)
    int leaf_quorum;                        // time-sliced walk: leaves are tested once (lanes waiting at a leaf) x this >= lanes still walking (0 = wait for all, -1 = default)
    int stack_cap;                          // most entries the LDS stack of a walk out of HBM may hold before it spills (kLdsStackCap, or the caller's)
    int stack_entries;                      // entries per lane the LDS stack holds (the walk spills beyond them where the kernel is built for it)
};

// counters[0] rays, [1] samples, [2] box tests, [3] sphere tests, [4] sphere hits,
// [5] rect tests, [6] wide tests, [7] shades
constexpr int kNumCounters = 8;
constexpr int kNumPhaseClocks = 5;         // extra slots behind the counters (diagnostic build)
constexpr int kNumCounterSlots = kNumCounters + kNumPhaseClocks + 4;   // + eight 32-bit tile counters of the persistent waves
constexpr int kNoBvh = 0x7fffffff;
constexpr int kLdsStackCap = 24;            // LDS stack entries of a walk out of HBM that still leave six workgroups per CU (24 KB + 2 KB of accumulators each)
constexpr int kSpillLanes = 256 * 6 * 256;  // lanes of the largest grid such a kernel is launched with (256 CUs x 6 workgroups x 256 threads)
constexpr int kExitShare = 64;             // a time-sliced walk is also left once no more than this many 128ths of the lanes that came in are unfinished
constexpr int kRoundShare = 32;            // the same for a plain walk that is left between rounds of leaves
constexpr int kMaxStack = 48;               // deepest BVH the traversal stack holds

struct Hit {              // rt_hit of include/rt.h, device side (same layout)
    int id;
    float t;
    float p[3];
    float n[3];
    int front;
    float u, v;
};

// launchers (rt_kernels.hip)
hipError_t launch_render(const DScene& sc, const DCamera& cam, const DRender& rp, float4* out,
                         unsigned long long* counters, hipStream_t stream);
hipError_t launch_trace(const DScene& sc, const float* rays, int n, float t_min, uint32_t flags,
                        Hit* out, hipStream_t stream);
hipError_t launch_resolve(const unsigned long long* sums, long long n_pixels, bool mean, float4* rgba,
                          hipStream_t stream);
hipError_t launch_quantize(const float4* rgba, long long n_pixels, int spp, uint8_t* rgb8,
                           hipStream_t stream);

hipError_t launch_guides(const DScene& sc, const DCamera& cam, int width, int height, float t_min, float4* out,
                         hipStream_t stream);

// rt_reconstruct.hip: the edge-avoiding a-trous passes over a frame and its guides
struct DReconstruct {
    int width, height, iterations;
    float sigma_color;        // <= 0: 1.6 / sqrt(samples of the pixel)
    float sigma_normal, sigma_depth, albedo_floor;
    int no_lds;               // every pass through the caches (for comparison with the passes staged in LDS)
};
// tmp: two float4 images of the frame's size; in, out may be the same buffer
hipError_t launch_reconstruct(const DReconstruct& rp, const float4* rgba, const float4* guides, float4* tmp, float4* out,
                              hipStream_t stream);

// rp.width / rp.height: the high resolution; the sigmas and the albedo floor as in launch_reconstruct
hipError_t launch_upscale(const DReconstruct& rp, int low_width, int low_height, const float4* rgba_low, const float4* guides_low,
                          const float4* guides_high, float4* out, hipStream_t stream);

}  // namespace rt
