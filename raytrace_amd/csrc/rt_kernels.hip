// gfx950 kernels of the per-pixel render path: camera ray -> closest hit -> scatter ->
// accumulate (main.cc:10-35,111-126), written for 64-wide wavefronts.
//
// Work decomposition: persistent waves draw tiles of 64/K pixels (K = lanes_per_pixel) from a device
// counter.  A tile's (pixel, sample) units are handed out dynamically: whenever lanes of the wave are
// free they take the next units (ballot + mbcnt rank), and every trip of the path loop traces ONE ray
// on every live lane, whichever sample or bounce that lane is at, so a lane whose path ended starts
// another sample in the very next trip instead of idling until the wave's longest path ends.  A
// pixel's radiance is summed in LDS as 32.32 fixed point with integer atomics: the sum does not
// depend on which lane rendered which sample or in what order, so an image is a pure function of
// (scene, camera, params) whatever K or the GPU count.
//
// Numerics: single precision everywhere except (a) spheres flagged "wide" (radius above
// rt_scene_desc.wide_radius: the r = 1000 ground of main.cc:42), whose quadratic cancels
// catastrophically in float and is solved in double exactly as sphere.h:23-41 writes it, and
// (b) a double-precision Newton polish of the winning ordinary sphere's root, so that hit
// distance and normal agree with the double-precision reference to float rounding.
#include "rt_bvh.h"
#include "rt_device.h"

#ifndef RT_WG_PER_CU
#define RT_WG_PER_CU 5                 // workgroups of 256 threads per CU = waves per SIMD the tree-walking kernels are compiled for
#endif
#ifndef RT_WG_PER_CU_ROUNDS
#define RT_WG_PER_CU_ROUNDS 5          // the same for walks in LDS that may be left between rounds of leaves (16 registers spill; at 4, without spills: 17.5 -> 19.3 ms)
#endif
#ifndef RT_WG_PER_CU_HBM
#define RT_WG_PER_CU_HBM 6             // the same for walks of a tree in HBM: more waves to wait on memory with
#endif
#ifndef RT_WG_PER_CU_SCAN
#define RT_WG_PER_CU_SCAN 6            // the same for the linear-scan kernels (fewer live values: 6 measured 5 % faster than 5)
#endif
#include "rt_philox.h"

namespace rt {

namespace {

constexpr int kBlock = 256;
constexpr float kInf = __builtin_huge_valf();

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// a + s*b
__device__ __forceinline__ V3 madd(V3 a, float s, V3 b) { return mk(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z)); }
__device__ __forceinline__ float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct Counters { unsigned box, sph, sph_hit, rect, wide, shade, steps, wave_steps; };

// true in exactly one of the currently active lanes
__device__ __forceinline__ bool first_active_lane() {
    int lane = (int)(threadIdx.x & 63);
    return __builtin_amdgcn_readfirstlane(lane) == lane;
}

// ---- closest hit ---------------------------------------------------------------------------

struct Cand {
    float tn;     // distance along the unit direction (= t * |d|)
    int kind;     // -1 none, 0 ordinary sphere, 1 wide sphere, 2 rect
    int idx;      // index into the kind's array
    int obj;      // list position: exact ties go to the later object (sceneobject_list.h:31-36
                  // with the inclusive range test of sphere.h:35)
};

__device__ __forceinline__ bool better(float tn, int obj, const Cand& b) {
    return tn < b.tn || (tn == b.tn && obj > b.obj);
}

// Sphere too large for the single-precision test (sphere.h:18-41), against the unit direction.
// What single precision cannot do here is form |o - c|^2 - r^2 and (o - c).d without losing the
// answer to cancellation (r = 1000: 0.06 of absolute error on a difference of a few units), so
// those three sums, |d|^2 and the discriminant are taken in double (17 operations).  The roots then
// come from the cancellation-free form of the quadratic formula, q = -(hb + sign(hb) sqrt(disc)),
// roots q and c / q, in single precision: each is good to a few ulps, like an ordinary sphere's,
// and the winner is polished in double like an ordinary sphere's.
__device__ __forceinline__ bool wide_hit(const DWide& w, V3 o, V3 dh, float tn_min, float tn_max, bool far_only, float& tn_out) {
    const double ocx = (double)o.x - w.cx, ocy = (double)o.y - w.cy, ocz = (double)o.z - w.cz;
    const double hb = fma(ocz, (double)dh.z, fma(ocy, (double)dh.y, ocx * (double)dh.x));
    const double cc = fma(ocz, ocz, fma(ocy, ocy, fma(ocx, ocx, -w.r2)));
    // |dh|^2 is 1 only to single precision (1e-7), and that error is multiplied by cc: for a sphere
    // seen from 30 radii away at a grazing angle it is as large as the discriminant itself
    const double a = fma((double)dh.z, (double)dh.z, fma((double)dh.y, (double)dh.y, (double)dh.x * (double)dh.x));
    const double disc = fma(hb, hb, -a * cc);
    if (disc < 0.0) return false;
    const float hbf = (float)hb, ccf = (float)cc;
    const float sq = __builtin_amdgcn_sqrtf((float)disc);
    const float q = hbf < 0.0f ? sq - hbf : -hbf - sq;
    const float other = ccf * __builtin_amdgcn_rcpf(q);   // q = 0 only for a tangent ray from the surface: NaN, no hit
    const float near = hbf < 0.0f ? other : q, far = hbf < 0.0f ? q : other;
    float root = near;
    if (far_only || !(root >= tn_min && root <= tn_max)) {
        root = far;
        if (!(root >= tn_min && root <= tn_max)) return false;
    }
    tn_out = root;
    return true;
}

// Ordinary sphere against a unit direction: distance of the centre's foot point along the ray
// (tca), squared distance of the centre from the ray line (l2), half chord (thc).  Works on
// differences of nearby numbers only, so it stays accurate where b*b - 4ac cancels.
// lo_near / lo_far: smallest distance accepted for the near and the far root (both tn_min, except
// on the sphere the ray starts on: see Origin).
__device__ __forceinline__ bool sphere_test(float4 s, V3 o, V3 dh, float lo_near, float lo_far, float tn_max, float& tn_out) {
    V3 L = mk(s.x - o.x, s.y - o.y, s.z - o.z);
    float tca = dot(L, dh);
    V3 l = madd(L, -tca, dh);
    float gap = fmaf(s.w, s.w, -dot(l, l));         // r^2 - (distance of the centre from the line)^2, as in hit_list
    if (!(gap >= 0.0f)) return false;
    float thc = __builtin_amdgcn_sqrtf(gap);        // 1 ulp; the winner's root is polished in double
    // sphere.h:29-37: the near root, else the far one.  A near root beyond tn_max leaves the far one
    // beyond it as well, so the far root is only ever wanted when the near one lies before its
    // lower bound: one select and three compares, no branch
    const float near_root = tca - thc;
    const bool use_far = !(near_root >= lo_near);
    const float root = use_far ? tca + thc : near_root, lo = use_far ? lo_far : lo_near;
    if (!(root >= lo && root <= tn_max)) return false;
    tn_out = root;
    return true;
}

// The primitive a ray starts on, as the scatter that made the ray recorded it.  In exact
// arithmetic such a ray meets that primitive again at t = 0 (rejected: below t_min) and, for a
// sphere entered or crossed from inside, at the far side; the double-precision reference computes
// exactly that.  The single-precision origin lies up to an ulp of its coordinates (3e-5 at
// |p| = 500) off the surface, and a root that should be 0 can come out above t_min * |d|.  So:
// a rect is not tested against the ray it emitted, a sphere left towards its outside is not
// tested, a sphere left towards its inside takes only its far root.
//   code < 0: none; else kind << 29 | inward << 28 | index   (kind 0 sphere, 1 wide sphere, 2 rect)
struct Origin {
    int code;
    __device__ __forceinline__ int sphere() const { return (code >= 0 && (code >> 29) == 0) ? (code & 0x0fffffff) : -1; }
    __device__ __forceinline__ int wide() const { return (code >= 0 && (code >> 29) == 1) ? (code & 0x0fffffff) : -1; }
    __device__ __forceinline__ int rect() const { return (code >= 0 && (code >> 29) == 2) ? (code & 0x0fffffff) : -1; }
    __device__ __forceinline__ bool inward() const { return (code >> 28) & 1; }
};
__device__ __forceinline__ Origin origin_of(int kind, int idx, bool inward) {
    Origin g; g.code = (kind << 29) | ((inward ? 1 : 0) << 28) | idx; return g;
}
__device__ __forceinline__ Origin no_origin() { Origin g; g.code = -1; return g; }

// cuda/sceneobject.h:84-88: a world-space ray seen from the instance (the translation comes off
// first), in double and rounded once; `num` is the distance from the origin to the plane
// coordinate k along axis ka, formed before rounding.  cuda/sceneobject.h:98-102: an
// object-space direction back in the world.
struct ObjRay { V3 o, d; float num; };
__device__ __forceinline__ ObjRay to_object(const DXform& x, V3 o, V3 d, int ka, double k) {
    double qx = (double)o.x - x.tx, qy = (double)o.y - x.ty, qz = (double)o.z - x.tz;
    double ox = fma(x.c, qx, -x.s * qz), oz = fma(x.s, qx, x.c * qz);
    double dx = fma(x.c, (double)d.x, -x.s * (double)d.z), dz = fma(x.s, (double)d.x, x.c * (double)d.z);
    ObjRay r;
    r.o = mk((float)ox, (float)qy, (float)oz);
    r.d = mk((float)dx, d.y, (float)dz);
    r.num = (float)(k - (ka == 0 ? ox : (ka == 1 ? qy : oz)));
    return r;
}
__device__ __forceinline__ V3 to_object_dir(const DXform& x, V3 v) {
    float c = (float)x.c, s = (float)x.s;
    return mk(fmaf(c, v.x, -s * v.z), v.y, fmaf(s, v.x, c * v.z));
}
__device__ __forceinline__ V3 to_world_dir(const DXform& x, V3 v) {
    float c = (float)x.c, s = (float)x.s;
    return mk(fmaf(c, v.x, s * v.z), v.y, fmaf(-s, v.x, c * v.z));
}

// cuda/rect.h:29-52 (xy), 75-99 (xz), 122-146 (yz); parametric t; num = k - origin along the
// plane's axis, inv = 1 / direction along that axis (polished_rcp: a division costs ten
// instructions per rect, the reciprocals three per ray and axis).  A ray parallel to the plane never
// hits (inv is then NaN).
__device__ __forceinline__ float polished_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
// KA: the plane's axis; the in-plane axes follow from it (yz: y, z; xz: x, z; xy: x, y).
template <int KA>
__device__ __forceinline__ bool rect_test(const DRect& r, float num, float inv, V3 o, V3 d, float t_min, float t_max, float& t_out) {
    constexpr int aa = KA == 0 ? 1 : 0, ba = KA == 2 ? 1 : 2;
    float t = num * inv;
    if (!(t >= t_min && t <= t_max)) return false;
    float av = fmaf(t, comp(d, aa), comp(o, aa));
    float bv = fmaf(t, comp(d, ba), comp(o, ba));
    if (!(av >= r.a0 && av <= r.a1 && bv >= r.b0 && bv <= r.b1)) return false;
    t_out = t;
    return true;
}

// Slab test (cuda/bounding_box.h:18-29) against [tn_min, tn_max] with the per-ray constants
// idir = 1/d and oid = -o/d, so a plane distance is one fused multiply-add.  Which of an axis'
// two planes the ray meets first follows from the sign of its direction alone, so the near and far
// plane are picked by that sign (a select) instead of by min/max of the two distances: min, max
// and compare issue at half the rate of a select or a multiply-add on this hardware.  Boxes are
// padded on the host.  The rounding error of the fused form is proportional to |o/d| of its axis:
// it is folded into the constants, taken off for the planes a ray enters through (oid_in) and
// added for the planes it leaves through (oid_out), so rounding can only add visits, never lose a
// hit, and the test itself has no term for it.  The bound is per axis on purpose: |o/d| is huge on
// an axis the ray barely moves along, and one bound for all three (the largest) would let such a
// ray into every box along its way, whichever axes the box misses it on; rays like that are few,
// but a wave waits for its slowest lane.
struct SlabSigns { bool x, y, z; };                    // direction component negative
// the test proper, from the three planes the ray enters through and the three it leaves through
__device__ __forceinline__ bool box_planes(float enter_x, float enter_y, float enter_z, float leave_x, float leave_y, float leave_z,
                                           V3 idir, V3 oid_in, V3 oid_out, float tn_min, float tn_max, float& t_enter) {
    float nx = fmaf(enter_x, idir.x, oid_in.x), fx = fmaf(leave_x, idir.x, oid_out.x);
    float ny = fmaf(enter_y, idir.y, oid_in.y), fy = fmaf(leave_y, idir.y, oid_out.y);
    float nz = fmaf(enter_z, idir.z, oid_in.z), fz = fmaf(leave_z, idir.z, oid_out.z);
    float tnear = fmaxf(fmaxf(fmaxf(nx, ny), nz), tn_min);
    float tfar = fminf(fminf(fminf(fx, fy), fz), tn_max);
    t_enter = tnear;
    return tnear <= tfar;
}
__device__ __forceinline__ bool box_test(float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                         V3 idir, V3 oid_in, V3 oid_out, SlabSigns neg, float tn_min, float tn_max, float& t_enter) {
    return box_planes(neg.x ? bx1 : bx0, neg.y ? by1 : by0, neg.z ? bz1 : bz0, neg.x ? bx0 : bx1, neg.y ? by0 : by1, neg.z ? bz0 : bz1,
                      idir, oid_in, oid_out, tn_min, tn_max, t_enter);
}

// the low / high 16 bits of a dword as a float, in one instruction
__device__ __forceinline__ float half_lo(unsigned w) {
    float f;
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "v"(w));
    return f;
}
__device__ __forceinline__ float half_hi(unsigned w) {
    float f;
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(w));
    return f;
}

// Per-ray constants of the slab test.  A direction component of zero (or too small for a finite
// reciprocal) takes +-1e30 as its reciprocal: the two plane distances of that axis then come out
// as huge numbers of opposite sign when the origin lies between the planes and of equal sign when
// it does not, which is the answer for a ray parallel to the slab (no error term there: the signs
// decide).  (With an infinite reciprocal b * inf - o * inf is NaN whenever b and o have the same
// sign.)
struct SlabRay { V3 idir, oid, oid_in, oid_out; SlabSigns neg; };
__device__ __forceinline__ float slab_margin(float idir, float size) {
    const float e = fabsf(idir) < 1.0e30f ? size : 0.0f;
    return isfinite(e) ? e : 0.0f;
}
// reach: the largest |coordinate| of any box plane, per axis.  A plane distance is at most
// reach * |1/d| + |o/d|; 6e-7 of that covers the reciprocals of the three axes being an ulp off each
// (distances of different axes are compared with each other), 4e-7 of |o/d| the rounding of the
// fused form.
__device__ __forceinline__ SlabRay slab_ray(V3 o, V3 dh, V3 reach) {
    SlabRay r;
    const float lim = 1.0e30f;
    r.idir = mk(fminf(fmaxf(__builtin_amdgcn_rcpf(dh.x), -lim), lim), fminf(fmaxf(__builtin_amdgcn_rcpf(dh.y), -lim), lim),
                fminf(fmaxf(__builtin_amdgcn_rcpf(dh.z), -lim), lim));
    const V3 oid = mk(-o.x * r.idir.x, -o.y * r.idir.y, -o.z * r.idir.z);
    r.oid = oid;
    r.neg.x = r.idir.x < 0.0f; r.neg.y = r.idir.y < 0.0f; r.neg.z = r.idir.z < 0.0f;
    const float ex = slab_margin(r.idir.x, fmaf(6.0e-7f * reach.x, fabsf(r.idir.x), 1.0e-6f * fabsf(oid.x))),
                ey = slab_margin(r.idir.y, fmaf(6.0e-7f * reach.y, fabsf(r.idir.y), 1.0e-6f * fabsf(oid.y))),
                ez = slab_margin(r.idir.z, fmaf(6.0e-7f * reach.z, fabsf(r.idir.z), 1.0e-6f * fabsf(oid.z)));
    r.oid_in = mk(oid.x - ex, oid.y - ey, oid.z - ez);
    r.oid_out = mk(oid.x + ex, oid.y + ey, oid.z + ez);
    return r;
}

// "Traversal finished" as a node reference.  Not a leaf code: a leaf code is ~(first << kLeafBits |
// count - 1) over fewer than 2^28 spheres (rt_upload_scene's limit), so it stays above -2^31; with
// the tree in LDS references are 16 bits wide and leaf codes range over fewer than 4088 sphere slots,
// so they stay above -32768.  The entry below a lane's traversal stack holds this value: a walk pops
// without asking whether anything is left, and what comes off an empty stack ends it.
constexpr int kTravDone = (int)0x80000000;
constexpr int kTravDone16 = -32768;

// Where the traversal reads its geometry from: HBM (through L1/L2), or a copy in LDS that the
// workgroup makes at kernel start when the whole BVH and sphere list fit.
struct GeoGlobal {
    static constexpr bool kPrefetchLeaf = true;
    static constexpr bool kPlaneAddressed = false;
    static constexpr bool kQuant = false;
    static constexpr bool kDeep = false;
    static constexpr int kDone = kTravDone;
    const float4* nodes; const float4* sph; const int2* info;
    V3 reach;                                                    // largest |coordinate| of the tree's box planes, per axis
    __device__ __forceinline__ int root_ref(int root) const { return root; }
    __device__ __forceinline__ float4 node(int i, int q) const { return nodes[4 * (size_t)i + q]; }
    __device__ __forceinline__ float4 sphere(int i) const { return sph[i]; }
    __device__ __forceinline__ int obj(int i) const { return info[i].x; }
    __device__ __forceinline__ int mat(int i) const { return info[i].y; }
};
// The tree in HBM as 32-byte nodes (rt_lbvh.h, quantize_nodes): two 16-byte loads per node instead of
// four.  The walk of a large scene is bound by the L1's tag lookups (the lanes of a wave stand at
// different nodes, so every 16-byte load is a lookup per lane), not by bytes or arithmetic.
// DEEP: the tree is deeper than the LDS stack the kernel's occupancy leaves room for (kLdsStackCap
// entries): the entries beyond it live in this lane's strip of a spill array in HBM.  A walk rarely
// gets there (the stack holds one entry per level at which BOTH children were met), but when it
// does it must not lose one.
template <bool DEEP>
struct GeoGlobalQT {
    static constexpr bool kPrefetchLeaf = true;
    static constexpr bool kPlaneAddressed = false;
    static constexpr bool kQuant = true;
    static constexpr bool kDeep = DEEP;
    static constexpr int kDone = kTravDone;
    const uint4* qnodes; const float4* sph; const int2* info;
    float lo[3], step[3];
    V3 reach;
    int* spill;                                                  // this lane's strip (DEEP)
    __device__ __forceinline__ int root_ref(int root) const { return root; }
    // (32-bit byte offset from a uniform base: one shift per step instead of 64-bit address
    // arithmetic; the quantised form is only made for trees of fewer than 2^27 nodes)
    __device__ __forceinline__ uint4 qnode(int i, int q) const {
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(qnodes) + (((unsigned)i << 5) + 16u * q));
    }
    __device__ __forceinline__ float4 sphere(int i) const { return sph[i]; }
    __device__ __forceinline__ int obj(int i) const { return info[i].x; }
    __device__ __forceinline__ int mat(int i) const { return info[i].y; }
};
using GeoGlobalQ = GeoGlobalQT<false>;
// The LDS copy of the nodes is laid out for the slab test: 80 bytes per node,
//   [x: left lo, left hi, left lo, right lo, right hi, right lo][y: the same][z: the same][left ref, right ref]
// so that the plane a ray enters a child's slab through sits at (axis base + 4 bytes if the
// direction component is negative) and the plane it leaves through in the next word, whatever the
// sign (the low plane is stored on both sides of the high one).  A lane reads the six planes of
// both children it needs through three addresses that carry its ray's signs, instead of reading
// all twelve and selecting; the selects were a fifth of the node step's VALU time and LDS has
// bandwidth to spare.  Inner-node references are the nodes' LDS byte addresses in units of 16
// bytes ((node_base + index * 80) / 16), leaves keep their negative codes.
#ifndef RT_FUSE_DRAWS
#define RT_FUSE_DRAWS 1                // which kernels draw a trip's scatter and camera numbers in one place (bits: see render_kernel)
#endif
#ifndef RT_LDS_NODE_WORDS
#define RT_LDS_NODE_WORDS 20
#endif
constexpr int kLdsNodeBytes = 4 * RT_LDS_NODE_WORDS;
// inner-node references in LDS: the node's byte address >> kLdsRefShift (16-byte units for the
// 80-byte node; an 84-byte node, whose bases spread over all 32 banks, is addressed in dwords)
constexpr int kLdsRefShift = (RT_LDS_NODE_WORDS % 4 == 0) ? 4 : 2;
typedef const float __attribute__((address_space(3))) lds_cfloat;
typedef const int __attribute__((address_space(3))) lds_cint;
typedef int __attribute__((address_space(3))) lds_int;
struct GeoLds {
    static constexpr bool kPrefetchLeaf = false;
    static constexpr bool kPlaneAddressed = true;
    static constexpr bool kQuant = false;
    static constexpr bool kDeep = false;
    static constexpr int kDone = kTravDone16;                     // what the 16-bit stack's bottom entry reads as
    unsigned node_base;                                          // LDS byte address of node 0
    const float4* sph; const int* objs;                          // both point into LDS
    V3 reach;
    __device__ __forceinline__ int root_ref(int root) const { return root >= 0 ? (int)((node_base + (unsigned)root * kLdsNodeBytes) >> kLdsRefShift) : root; }   // a tiny scene's root is a leaf
    __device__ __forceinline__ float4 sphere(int i) const { return sph[i]; }
    // list position and material share a word (both below 65536 in a scene that fits in LDS)
    __device__ __forceinline__ int obj(int i) const { return objs[i] & 0xffff; }
    __device__ __forceinline__ int mat(int i) const { return (int)((unsigned)objs[i] >> 16); }
};

template <bool COUNT, class G>
__device__ __forceinline__ void leaf_sphere(const G& geo, int i, float4 s, V3 o, V3 dh, float tn_min, int self, Cand& best, Counters& cn) {
    float tn;
    if (COUNT) cn.sph++;
    const float lo = i == self ? kInf : tn_min;        // the sphere the ray starts on is settled before the traversal
    if (sphere_test(s, o, dh, lo, lo, best.tn, tn)) {
        int obj = geo.obj(i);
        if (COUNT) cn.sph_hit++;
        // the range test let through tn <= best.tn, so better() comes down to "not an exact tie that
        // the earlier list entry loses": one compare on the common path instead of four
        bool take = true;
        if (tn == best.tn) take = obj > best.obj;
        if (take) { best.tn = tn; best.kind = 0; best.idx = i; best.obj = obj; }
    }
}

template <bool COUNT, class G>
__device__ __forceinline__ void leaf_spheres(const G& geo, int first, int count, V3 o, V3 dh,
                                             float tn_min, int self, Cand& best, Counters& cn) {
    if constexpr (G::kPrefetchLeaf) {
        // geometry in HBM: a leaf's records are requested two at a time, so that a leaf of one or
        // two spheres (most leaves of a large scene) costs one memory round trip and two tag
        // lookups per lane (the arrays are padded for the reads past a shorter leaf)
        float4 s0 = geo.sphere(first), s1 = geo.sphere(first + 1);
        leaf_sphere<COUNT>(geo, first, s0, o, dh, tn_min, self, best, cn);
        if (count > 1) leaf_sphere<COUNT>(geo, first + 1, s1, o, dh, tn_min, self, best, cn);
        if (count > 2) {
            float4 s2 = geo.sphere(first + 2), s3 = geo.sphere(first + 3);
            leaf_sphere<COUNT>(geo, first + 2, s2, o, dh, tn_min, self, best, cn);
            if (count > 3) leaf_sphere<COUNT>(geo, first + 3, s3, o, dh, tn_min, self, best, cn);
        }
        for (int i = first + 4; i < first + count; i++) leaf_sphere<COUNT>(geo, i, geo.sphere(i), o, dh, tn_min, self, best, cn);
    } else {
        for (int i = first; i < first + count; i++) leaf_sphere<COUNT>(geo, i, geo.sphere(i), o, dh, tn_min, self, best, cn);
    }
}

__device__ __forceinline__ Cand no_hit() {
    Cand best;
    best.tn = kInf; best.kind = -1; best.idx = -1; best.obj = -1;
    return best;
}

// Per-ray constants of the closest-hit query.
struct RayN {
    V3 dh;          // unit direction
    float len;      // |d|
    float inv_len;
    float tn_min;   // t_min along the unit direction
};
__device__ __forceinline__ RayN normalise(V3 d, float t_min) {
    RayN r;
    // hardware reciprocal square root (1 ulp): a scale error e of dh enters the foot-point test
    // only as e^2, and the winner's t is polished in double from (o, d) themselves
    const float d2 = dot(d, d);
    r.inv_len = __builtin_amdgcn_rsqf(d2);
    r.len = d2 * r.inv_len;
    r.dh = r.inv_len * d;
    r.tn_min = t_min * r.len;
    return r;
}

// Always-tested primitives: wide spheres (double) and rects.  They run first and tighten the
// distance bound the BVH traversal prunes with.
// FEAT: what the scene holds besides plainly coloured spheres (kFeat* bits).  Each kernel carries the
// code of its scene's features only: code a scene never runs still costs the code that does run
// registers and scheduling freedom when both share a kernel (the rect code: 3 % of a frame of
// spheres; the texture code in the room of rects and the rect and instance code in the field of
// textured spheres: the registers those kernels spilled).
//   kFeatRects      rects (cuda/rect.h:29-52)
//   kFeatTextures   some material reads a texture (cuda/texture.h:17-78): surface coordinates, texture_value
//   kFeatInstances  some primitive stands under rotate_y / translate (cuda/sceneobject.h:31-109)
//   kFeatEmitters   some material emits (cuda/material.h:33-47,103-119): without one a path's radiance is
//                   its throughput times the background where it leaves the scene, and the running sum
//                   of emitted light (three registers for the whole path) is not kept
constexpr int kFeatRects = 1, kFeatTextures = 2, kFeatInstances = 4, kFeatEmitters = 8, kFeatAll = 15;
// A room's rects (at most kLdsRects of them: kFeatRoom, kFeatTexturedRoom) are read from a copy in LDS, every
// lane the same record at a time (a broadcast): as scalar-load operands they make every instruction of the
// rect test issue at half rate, and a loop over six of them cannot hide six scalar-load latencies.
constexpr int kLdsRects = 16;
typedef const DRect __attribute__((address_space(3))) lds_crect;
__host__ __device__ constexpr bool staged_rects(int feat) { return (feat & kFeatRects) != 0 && feat != kFeatAll; }
template <int FEAT>
__device__ __forceinline__ DRect rect_record(const DScene& sc, lds_crect* staged, int i) {
    if constexpr (staged_rects(FEAT)) {
        lds_crect& s = staged[i];
        DRect r;
        r.a0 = s.a0; r.a1 = s.a1; r.b0 = s.b0; r.b1 = s.b1; r.k = s.k; r.axes = s.axes; r.obj = s.obj; r.mat = s.mat;
        return r;
    } else {
        return sc.rects[i];
    }
}
constexpr int kFeatRoom = kFeatRects | kFeatEmitters;   // the combinations kernels are built for besides none and all:
constexpr int kFeatTexturedRoom = kFeatRoom | kFeatTextures;   // a room with its light, the same with textures, kFeatTextures alone
template <bool COUNT, int FEAT>
__device__ __forceinline__ void hit_always(const DScene& sc, lds_crect* staged, V3 o, V3 d, const RayN& rn, float t_min, Origin from,
                                           Cand& best, Counters& cn) {
    const int from_wide = from.wide(), from_rect = from.rect();
    for (int i = 0; i < sc.n_wide; i++) {
        const DWide w = sc.wide[i];
        float tn;
        if (COUNT) cn.wide++;
        if (i == from_wide && !from.inward()) continue;
        if (wide_hit(w, o, rn.dh, rn.tn_min, best.tn, i == from_wide, tn) && better(tn, w.obj, best)) {
            best.tn = tn; best.kind = 1; best.idx = i; best.obj = w.obj;
        }
    }
    constexpr bool RECTS = (FEAT & kFeatRects) != 0, INST = (FEAT & kFeatInstances) != 0;
    V3 inv_d = mk(0.0f, 0.0f, 0.0f);
    if constexpr (RECTS)
        if (sc.n_rects > 0) inv_d = mk(polished_rcp(d.x), polished_rcp(d.y), polished_rcp(d.z));
    if constexpr (RECTS)
    for (int i = 0; i < sc.n_rects; i++) {
        const DRect r = rect_record<FEAT>(sc, staged, i);
        float t;
        if (COUNT) cn.rect++;
        if (i == from_rect) continue;
        // the record is the same in every lane (it arrives through scalar loads), so what kind of
        // rect this is and which axis its plane cuts are wave-uniform branches: one copy of the test
        // per plane axis, each on the ray's components themselves
        bool hit;
        const int axes = __builtin_amdgcn_readfirstlane(r.axes);      // (the same in every lane: tests on it are scalar)
        const int ka = axes & 3;
        if (INST && (axes >> 8)) {           // instance: a rotation keeps |d|, so t means the same
            const ObjRay q = to_object(sc.xforms[(axes >> 8) - 1], o, d, ka, sc.rect_k[i]);
            if (ka == 0) hit = rect_test<0>(r, q.num, polished_rcp(q.d.x), q.o, q.d, t_min, kInf, t);
            else if (ka == 1) hit = rect_test<1>(r, q.num, polished_rcp(q.d.y), q.o, q.d, t_min, kInf, t);
            else hit = rect_test<2>(r, q.num, polished_rcp(q.d.z), q.o, q.d, t_min, kInf, t);
        } else {
            // the distance to the plane: in float when the plane coordinate is a float (the
            // subtraction is then as good as the double one rounded), else in double
            const bool k64 = (axes & 64) != 0;
            if (ka == 0) hit = rect_test<0>(r, k64 ? (float)(sc.rect_k[i] - (double)o.x) : r.k - o.x, inv_d.x, o, d, t_min, kInf, t);
            else if (ka == 1) hit = rect_test<1>(r, k64 ? (float)(sc.rect_k[i] - (double)o.y) : r.k - o.y, inv_d.y, o, d, t_min, kInf, t);
            else hit = rect_test<2>(r, k64 ? (float)(sc.rect_k[i] - (double)o.z) : r.k - o.z, inv_d.z, o, d, t_min, kInf, t);
        }
        if (hit) {
            float tn = t * rn.len;
            if (better(tn, r.obj, best)) { best.tn = tn; best.kind = 2; best.idx = i; best.obj = r.obj; }
        }
    }
}

// sceneobject_list.h:26-39: every ordinary sphere, in list order.  The sphere index is the same
// in all lanes, so the records arrive through scalar loads and sit in SGPRs; four spheres are
// screened per trip with the branch-free part of the test (is the centre closer to the ray's
// line than the radius?) and only a group in which some lane passes runs the full test.  The
// array is padded to a multiple of four with spheres nothing can hit.
template <bool COUNT, bool STAGED>
__device__ __forceinline__ void hit_list(const DScene& sc, const float4* staged, V3 o, const RayN& rn, Origin from, Cand& best, Counters& cn) {
    const int self = from.sphere();
    const bool self_in = from.inward();
    // constant address space: tells the compiler nothing in the kernel writes these records, which
    // is what it needs to fetch them with scalar loads
    typedef const float __attribute__((address_space(4))) cfloat;
    typedef const int __attribute__((address_space(4))) cint;
    const cfloat* sph = (const cfloat*)(unsigned long long)sc.sph;
    const cint* info = (const cint*)(unsigned long long)sc.sph_info;
    const V3 dh = rn.dh;
    for (int i = 0; i < sc.n_sph_padded; i += 4) {
        float tca[4], gap[4];                          // gap = r^2 - (distance of the centre from the line)^2
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float sx, sy, sz, sr;
            // (the padding of a short list is not screened: a wave-uniform skip.  Only there: in the
            // long list's loop the test cost 15 % of the frame for the one group it can skip)
            if constexpr (STAGED)
                if (k >= 2 && i + k >= sc.n_sph) { tca[k] = 0.0f; gap[k] = -1.0f; continue; }
            if constexpr (STAGED) {
                // from the workgroup's LDS copy: every lane reads the same address (a broadcast)
                const float4 s4 = staged[i + k];
                sx = s4.x; sy = s4.y; sz = s4.z; sr = s4.w;
            } else {
                sx = sph[4 * (i + k)]; sy = sph[4 * (i + k) + 1]; sz = sph[4 * (i + k) + 2]; sr = sph[4 * (i + k) + 3];
            }
            V3 L = mk(sx - o.x, sy - o.y, sz - o.z);
            tca[k] = dot(L, dh);
            V3 l = madd(L, -tca[k], dh);
            gap[k] = fmaf(sr, sr, -dot(l, l));
        }
        if (COUNT) cn.sph += 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // wave-uniform branch: skipped when no lane's line comes within the radius; inside,
            // the rest of sphere_test without branches.  The scan runs in list order, so among
            // spheres "later object wins an exact tie" is "accept when root <= closest so far".
            if (__ballot(gap[k] >= 0.0f) != 0ull) {
                float thc = __builtin_amdgcn_sqrtf(fmaxf(gap[k], 0.0f));
                float t0 = tca[k] - thc, t1 = tca[k] + thc;
                const bool own = i + k == self;                   // the sphere this ray starts on
                // an exact tie goes to the later list entry; the candidate so far is an earlier
                // sphere or one of the always-tested primitives, which may come later in the list:
                // then the bound is the next float below its distance
                const int obj = info[2 * (i + k)];
                const float lim = __int_as_float(__float_as_int(best.tn) - (obj > best.obj ? 0 : 1));
                bool ok0 = t0 >= rn.tn_min && t0 <= lim && !own;
                bool ok1 = t1 >= rn.tn_min && t1 <= lim && !(own && !self_in);
                bool hit = gap[k] >= 0.0f && (ok0 || ok1);
                if (COUNT && hit) cn.sph_hit++;
                best.tn = hit ? (ok0 ? t0 : t1) : best.tn;
                best.idx = hit ? i + k : best.idx;
                best.kind = hit ? 0 : best.kind;
                best.obj = hit ? obj : best.obj;
            }
        }
    }
}

// The sphere a ray starts on, when the ray heads into it (refraction, internal reflection): only
// its far side can be hit.  Settled here, before the traversal, which then skips that sphere.
template <class G>
__device__ __forceinline__ void hit_own_sphere(const G& geo, V3 o, const RayN& rn, Origin from, Cand& best) {
    const int self = from.sphere();
    if (self < 0 || !from.inward()) return;
    float tn;
    if (sphere_test(geo.sphere(self), o, rn.dh, kInf, rn.tn_min, best.tn, tn)) {
        int obj = geo.obj(self);
        if (better(tn, obj, best)) { best.tn = tn; best.kind = 0; best.idx = self; best.obj = obj; }
    }
}

// BVH traversal, near child first, as two nested loops: the inner one walks inner nodes until
// every lane of the wave stands on a leaf (or has finished), then all those lanes test their
// leaves together.  A leaf costs as much as two node steps and only one lane step in eight is
// a leaf, so testing leaves as they come would run the sphere code for a handful of lanes on
// nearly every trip.  `stack` is this lane's column of the workgroup's LDS traversal stack
// (entry e at stack[e * kBlock]).
//
// The traversal is resumable: (cur, sp) and `best` are its whole state.  With budget > 0 the
// wave leaves after about that many node steps; lanes whose traversal is not finished
// (cur != G::kDone) come back later with the same state, so one long ray does not hold the
// other 63 lanes of the wave.

typedef short __attribute__((address_space(3))) lds_short;
template <bool SHORT>
__device__ __forceinline__ void push_ref(unsigned top, int ref) {
    if constexpr (SHORT) *(lds_short*)(size_t)top = (short)ref; else *(lds_int*)(size_t)top = ref;
}
template <bool SHORT>
__device__ __forceinline__ int pop_ref(unsigned top) {
    if constexpr (SHORT) return (int)*(lds_short*)(size_t)top; else return *(lds_int*)(size_t)top;
}

template <bool COUNT, int SL, class G>     // SL: 0 a walk ends in the call; 1 time slices; 2 left between rounds only
__device__ __forceinline__ void hit_bvh(const G& geo, V3 o, const RayN& rn, Origin from, int* stack, int lds_entries, int budget, int quorum,
                                        int& cur, int& sp, Cand& best, Counters& cn) {
    const V3 dh = rn.dh;
    const int self = from.sphere();
    const float tn_min = rn.tn_min;
    const SlabRay sr = slab_ray(o, dh, geo.reach);
    unsigned enter_x = 0u, enter_y = 0u, enter_z = 0u;    // plane-addressed nodes: where this ray's entry planes sit in a node
    if constexpr (G::kPlaneAddressed) {
        enter_x = sr.neg.x ? 4u : 0u;
        enter_y = sr.neg.y ? 28u : 24u;
        enter_z = sr.neg.z ? 52u : 48u;
    }
    // quantised nodes: a plane is lo + k * step, so its distance along the ray is k * (step / d) +
    // (lo - o) / d: grid origin and step are folded into the ray's constants, and the plane a ray
    // enters through is the low or the high half of a dword by the sign of its direction
    V3 qa = mk(0.0f, 0.0f, 0.0f), qb_in = mk(0.0f, 0.0f, 0.0f), qb_out = mk(0.0f, 0.0f, 0.0f);
    unsigned sh_x = 0u, sh_y = 0u, sh_z = 0u;
    if constexpr (G::kQuant) {
        qa = mk(geo.step[0] * sr.idir.x, geo.step[1] * sr.idir.y, geo.step[2] * sr.idir.z);
        const V3 oid = sr.oid;
        const V3 qb = mk(fmaf(geo.lo[0], sr.idir.x, oid.x), fmaf(geo.lo[1], sr.idir.y, oid.y), fmaf(geo.lo[2], sr.idir.z, oid.z));
        sh_x = sr.neg.x ? 16u : 0u; sh_y = sr.neg.y ? 16u : 0u; sh_z = sr.neg.z ? 16u : 0u;
        // rounding error bound of k * qa + qb per axis, from the sizes of the terms that were added up,
        // plus 6e-7 of the largest distance there can be for the reciprocals being an ulp off
        const float ex = slab_margin(sr.idir.x, 1.4e-6f * (fabsf(oid.x) + fabsf(geo.lo[0] * sr.idir.x) + 65536.0f * fabsf(qa.x)));
        const float ey = slab_margin(sr.idir.y, 1.4e-6f * (fabsf(oid.y) + fabsf(geo.lo[1] * sr.idir.y) + 65536.0f * fabsf(qa.y)));
        const float ez = slab_margin(sr.idir.z, 1.4e-6f * (fabsf(oid.z) + fabsf(geo.lo[2] * sr.idir.z) + 65536.0f * fabsf(qa.z)));
        qb_in = mk(qb.x - ex, qb.y - ey, qb.z - ez);
        qb_out = mk(qb.x + ex, qb.y + ey, qb.z + ez);
    }
    // the stack top as an LDS byte address: a push or pop is one add and one LDS access.  With
    // the tree in LDS a reference fits 16 bits (a node's address in units of 16 bytes, or a leaf
    // code over fewer than 4096 sphere slots): half the stack, which is what lets the 80-byte nodes
    // in beside it at five workgroups per CU.
    // Entry 0 of a lane's column holds G::kDone for the whole launch (render_kernel and trace_kernel
    // write it): popping it ends the walk, so no pop asks whether the stack is empty.
    constexpr unsigned kStackStride = kBlock * (G::kPlaneAddressed ? sizeof(short) : sizeof(int));
    constexpr int kDone = G::kDone;
    const unsigned bottom = (unsigned)(size_t)(__attribute__((address_space(3))) char*)stack + kStackStride;
    unsigned top = bottom + (unsigned)sp * kStackStride;
    // (G::kDeep) `top` keeps counting past the end of the LDS stack; entries from `limit` on are in the spill strip
    unsigned limit = 0u;
    if constexpr (G::kDeep) limit = bottom + (unsigned)(lds_entries - 1) * kStackStride;
    int wave_steps = 0;
    bool stop = false;
    // (SLICED) budget = steps | share << 16: the wave also leaves the walk as soon as the lanes whose walk
    // is unfinished are no more than share / 128 of the lanes that came in
    constexpr bool SLICED = SL == 1, RESUMED = SL != 0;
    const int exit_ratio = RESUMED ? budget >> 16 : 0;
    if constexpr (RESUMED) budget &= 0xffff;
    const int entered = RESUMED ? __popcll(__ballot(true)) : 0;
    while (cur != kDone && !stop) {
        // Deep trees: a lane meets a leaf every seven or eight node steps, and waiting until all 64
        // lanes stand on one leaves most of them idle for most of the walk.  The wave therefore
        // turns to the leaves as soon as the lanes waiting at one, times `quorum`, outnumber the
        // lanes still walking (both wave-uniform: built from ballots only); the walkers carry on
        // afterwards.  Shallow trees keep the plain rule: there the two extra ballots per step
        // cost more than the shorter waits return.
        int waiting = SLICED ? __popcll(__ballot(cur < 0)) : 0;
        bool leave = false;
        while (cur >= 0 && !stop && !leave) {
            if (COUNT) { cn.steps++; if (first_active_lane()) cn.wave_steps++; }
            int left, right;
            float tl, tr;
            bool hl, hr;
            if (COUNT) cn.box += 2;
            if constexpr (G::kPlaneAddressed) {
                // per axis: words 0 and 3 from the ray's address are the planes it enters the left and
                // the right child through, words 1 and 4 the ones it leaves through
                lds_cfloat* px = (lds_cfloat*)(size_t)(((unsigned)cur << kLdsRefShift) + enter_x);
                lds_cfloat* py = (lds_cfloat*)(size_t)(((unsigned)cur << kLdsRefShift) + enter_y);
                lds_cfloat* pz = (lds_cfloat*)(size_t)(((unsigned)cur << kLdsRefShift) + enter_z);
                lds_cint* refs = (lds_cint*)(size_t)(((unsigned)cur << kLdsRefShift) + 72u);
                left = refs[0]; right = refs[1];
                float l_ex = px[0], r_ex = px[3], l_lx = px[1], r_lx = px[4];
                float l_ey = py[0], r_ey = py[3], l_ly = py[1], r_ly = py[4];
                float l_ez = pz[0], r_ez = pz[3], l_lz = pz[1], r_lz = pz[4];
                hl = box_planes(l_ex, l_ey, l_ez, l_lx, l_ly, l_lz, sr.idir, sr.oid_in, sr.oid_out, tn_min, best.tn, tl);
                hr = box_planes(r_ex, r_ey, r_ez, r_lx, r_ly, r_lz, sr.idir, sr.oid_in, sr.oid_out, tn_min, best.tn, tr);
            } else if constexpr (G::kQuant) {
                const uint4 a = geo.qnode(cur, 0), b = geo.qnode(cur, 1);
                left = (int)b.z; right = (int)b.w;
                // per dword one rotation by the ray's shift (0 or 16) puts the plane the ray enters
                // through into the low half; the halves are converted straight out of the dword
                // (sub-dword operand select), three instructions per axis and box
                const unsigned lx = __builtin_amdgcn_alignbit(a.x, a.x, sh_x), ly = __builtin_amdgcn_alignbit(a.y, a.y, sh_y),
                               lz = __builtin_amdgcn_alignbit(a.z, a.z, sh_z), rx = __builtin_amdgcn_alignbit(a.w, a.w, sh_x),
                               ry = __builtin_amdgcn_alignbit(b.x, b.x, sh_y), rz = __builtin_amdgcn_alignbit(b.y, b.y, sh_z);
                hl = box_planes(half_lo(lx), half_lo(ly), half_lo(lz), half_hi(lx), half_hi(ly), half_hi(lz), qa, qb_in, qb_out, tn_min, best.tn, tl);
                hr = box_planes(half_lo(rx), half_lo(ry), half_lo(rz), half_hi(rx), half_hi(ry), half_hi(rz), qa, qb_in, qb_out, tn_min, best.tn, tr);
            } else {
                float4 q0 = geo.node(cur, 0), q1 = geo.node(cur, 1), q2 = geo.node(cur, 2), q3 = geo.node(cur, 3);
                left = __float_as_int(q3.x); right = __float_as_int(q3.y);
                hl = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, sr.idir, sr.oid_in, sr.oid_out, sr.neg, tn_min, best.tn, tl);
                hr = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, sr.idir, sr.oid_in, sr.oid_out, sr.neg, tn_min, best.tn, tr);
            }
            // (a straight-line version of this step, with selects and unconditional stack traffic,
            // measured 1 % slower)
            if (hl && hr) {
                bool left_first = tl <= tr;
                if constexpr (G::kDeep) {
                    if (top < limit) push_ref<false>(top, left_first ? right : left);
                    else geo.spill[(top - limit) / kStackStride] = left_first ? right : left;
                } else {
                    push_ref<G::kPlaneAddressed>(top, left_first ? right : left);
                }
                top += kStackStride;
                cur = left_first ? left : right;
            } else if (hl || hr) {
                cur = hl ? left : right;
            } else {
                top -= kStackStride;
                if constexpr (G::kDeep) cur = top >= limit ? geo.spill[(top - limit) / kStackStride] : pop_ref<false>(top);
                else cur = pop_ref<G::kPlaneAddressed>(top);
            }
            if constexpr (SLICED) {
                waiting += __popcll(__ballot(cur < 0 && cur != kDone));
                const int walking = __popcll(__ballot(cur >= 0));
                leave = waiting * quorum >= walking;
                // the count is taken from the first active lane; a lane that sat out some steps at
                // a leaf brings an older count back when it becomes that lane, so a slice lasts at
                // least `budget` steps and about twice that in practice (measured 66 for 32)
                wave_steps = __builtin_amdgcn_readfirstlane(wave_steps) + 1;
                stop = wave_steps >= budget || (walking + waiting) * 128 <= entered * exit_ratio;
            }
        }
        if (cur < 0 && cur != kDone) {
            if (COUNT) { cn.steps++; if (first_active_lane()) cn.wave_steps++; }
            int code = ~cur;
            leaf_spheres<COUNT>(geo, code >> kLeafBits, (code & (kLeafCap - 1)) + 1, o, dh, tn_min, self, best, cn);
            top -= kStackStride;
            if constexpr (G::kDeep) cur = top >= limit ? geo.spill[(top - limit) / kStackStride] : pop_ref<false>(top);
            else cur = pop_ref<G::kPlaneAddressed>(top);
        }
        // (SL == 2) the plain walk, left after a round of leaves once the lanes with an unfinished walk
        // are no more than share / 128 of those that came in: one ballot a round, none per step
        if constexpr (SL == 2) stop = __popcll(__ballot(cur != kDone)) * 128 <= entered * exit_ratio;
    }
    // (a finished walk leaves top one entry below the bottom: its sp is not used again)
    if constexpr (RESUMED) sp = (int)(top - bottom) / (int)kStackStride;
}

// ---- hit record ------------------------------------------------------------------------------

struct Rec {
    V3 p, n;        // n faces against the ray (sceneobject.h:16-23)
    float t;        // parametric
    float u, v;
    int front, mat, obj;
    bool hollow;    // sphere of negative radius: its outward normal points at the centre
};

// cuda/sphere.h:20-26 on the unit outward normal.  theta = acos(-y) is taken as
// atan2(|(x,z)|, -y): the same angle, but well conditioned at the poles where acos turns a
// 1e-7 error of y into 4e-4 of theta.
__device__ __forceinline__ void sphere_uv(V3 on, float& u, float& v) {
    const float pi = 3.14159265358979323846f;
    float theta = atan2f(sqrtf(fmaf(on.x, on.x, on.z * on.z)), -on.y);
    float phi = atan2f(-on.z, on.x) + pi;
    u = phi * (1.0f / (2.0f * pi));
    v = theta * (1.0f / pi);
}

// texture coordinates of a sphere instance come from the normal in its own frame
__device__ __forceinline__ void sphere_uv_inst(const DScene& sc, int xform, V3 on, float& u, float& v) {
    if (xform) on = to_object_dir(sc.xforms[xform - 1], on);
    sphere_uv(on, u, v);
}

__device__ __forceinline__ void face(Rec& r, V3 d, V3 outward) {
    r.front = dot(d, outward) < 0.0f;
    r.n = r.front ? outward : -outward;
}

// Fill the record of the winning candidate.  Ordinary spheres: Newton steps on
// f(t) = a t^2 + 2 hb t + c in double from the single-precision root (the correction itself is
// a float division), then p and the normal from the polished t.
// `s64` is the winning ordinary sphere in double and `mat` its material, both loaded by the caller.
template <int FEAT>
__device__ __forceinline__ Rec make_record(const DScene& sc, lds_crect* staged, const Cand& c, V3 o, V3 d, float inv_len, bool refine, bool want_uv,
                                           double4 s64, int mat) {
    constexpr bool RECTS = (FEAT & kFeatRects) != 0, TEX = (FEAT & kFeatTextures) != 0, INST = (FEAT & kFeatInstances) != 0;
    if constexpr (!TEX) want_uv = false;
    Rec r;
    r.obj = c.obj;
    r.u = 0.0f; r.v = 0.0f;
    r.hollow = false;
    if (RECTS && c.kind == 2) {
        const DRect q = rect_record<FEAT>(sc, staged, c.idx);
        int ka = q.axes & 3, aa = (q.axes >> 2) & 3, ba = (q.axes >> 4) & 3;
        float t = c.tn * inv_len;
        r.t = t;
        r.p = madd(o, t, d);
        V3 axis = mk(ka == 0 ? 1.0f : 0.0f, ka == 1 ? 1.0f : 0.0f, ka == 2 ? 1.0f : 0.0f);
        V3 po = r.p;
        // the hit lies in the rect's plane: that coordinate of the point is k itself (as the
        // reference's o + t d is to 1e-12), not a single-precision o + t d, which strays by an ulp
        // of the scene's coordinates and lets a solid texture flip where it changes across the plane
        if (INST && (q.axes >> 8)) {
            // the instance's own hit record, then point and normal back in the world.  The side
            // of the surface is the one the instance saw (cuda/sceneobject.h:105 asks again with
            // the object-space ray against the world-space normal, which mixes the two frames).
            const DXform x = sc.xforms[(q.axes >> 8) - 1];
            const double kd = sc.rect_k[c.idx];
            ObjRay oq = to_object(x, o, d, ka, kd);
            po = madd(oq.o, t, oq.d);
            const double pox = ka == 0 ? kd : (double)po.x, poy = ka == 1 ? kd : (double)po.y, poz = ka == 2 ? kd : (double)po.z;
            r.p = mk((float)(fma(x.c, pox, x.s * poz) + x.tx), (float)(poy + x.ty), (float)(fma(-x.s, pox, x.c * poz) + x.tz));
            face(r, oq.d, axis);
            r.n = to_world_dir(x, r.n);
        } else {
            if (ka == 0) r.p.x = q.k; else if (ka == 1) r.p.y = q.k; else r.p.z = q.k;      // q.k is k rounded to float
            face(r, d, axis);
        }
        if (want_uv) {                         // cuda/rect.h:45-46; only a textured material looks at them
            r.u = (comp(po, aa) - q.a0) / (q.a1 - q.a0);
            r.v = (comp(po, ba) - q.b0) / (q.b1 - q.b0);
        }
        r.mat = q.mat;
        return r;
    }
    double cx, cy, cz, rad, t;
    int xform = 0;
    const float tf = c.tn * inv_len;
    if (c.kind == 1) {
        const DWide w = sc.wide[c.idx];
        cx = w.cx; cy = w.cy; cz = w.cz; rad = w.r;
        r.mat = w.mat;
        if constexpr (INST) xform = w.xform;
    } else {
        if constexpr (INST)
            if (want_uv && sc.sph_xform) xform = sc.sph_xform[c.idx];
        r.mat = mat;
        cx = s64.x; cy = s64.y; cz = s64.z; rad = s64.w;
    }
    if (!refine) {
        r.hollow = rad < 0.0;
        r.t = tf;
        r.p = madd(o, tf, d);
        V3 on = (1.0f / (float)rad) * mk(r.p.x - (float)cx, r.p.y - (float)cy, r.p.z - (float)cz);
        face(r, d, on);
        if (want_uv) sphere_uv_inst(sc, xform, on, r.u, r.v);
        return r;
    }
    {
        double ocx = (double)o.x - cx, ocy = (double)o.y - cy, ocz = (double)o.z - cz;
        double dx = d.x, dy = d.y, dz = d.z;
        double a = dx * dx + dy * dy + dz * dz;
        double hb = ocx * dx + ocy * dy + ocz * dz;
        double cc = (ocx * ocx + ocy * ocy + ocz * ocz) - rad * rad;
        t = (double)tf;
        const float af = (float)a;
        for (int it = 0; it < 4; it++) {
            double g = fma(a, t, hb);                 // f'(t) / 2
            double f = fma(g + hb, t, cc);            // (a t + 2 hb) t + c
            float fp = 2.0f * (float)g;
            if (fp == 0.0f) break;
            float inv_fp = __builtin_amdgcn_rcpf(fp);
            float corr = (float)f * inv_fp;
            t -= (double)corr;
            // a step leaves corr^2 f'' / (2 f') behind.  Nearly always the first step ends it; a
            // ray that grazes (f' small, and a poorer float estimate to start from) takes more
            if (corr * corr * fabsf(af * inv_fp) <= 3e-8f * fabsf(tf)) break;
        }
    }
    r.hollow = rad < 0.0;
    double px = fma(t, (double)d.x, (double)o.x), py = fma(t, (double)d.y, (double)o.y), pz = fma(t, (double)d.z, (double)o.z);
    float inv_r = __builtin_amdgcn_rcpf((float)rad);
    double ir = (double)inv_r;
    ir = ir * (2.0 - rad * ir);                       // one Newton step: 1/r to double accuracy
    V3 on = mk((float)((px - cx) * ir), (float)((py - cy) * ir), (float)((pz - cz) * ir));
    r.t = (float)t;
    r.p = mk((float)px, (float)py, (float)pz);
    face(r, d, on);
    if (want_uv) sphere_uv_inst(sc, xform, on, r.u, r.v);
    return r;
}

// ---- textures (cuda/texture.h) --------------------------------------------------------------------

__device__ __forceinline__ V3 texture_value(const DScene& sc, int tex, float u, float v, V3 p) {
    for (int guard = 0; guard < 16; guard++) {
        const DTexture t = sc.texs[tex];
        if (t.type == 0) return mk(t.r, t.g, t.bl);
        if (t.type == 1) {                             // cuda/texture.h:43-52
            float sines = sinf(10.0f * p.x) * sinf(10.0f * p.y) * sinf(10.0f * p.z);
            tex = sines < 0.0f ? t.b : t.a;
            continue;
        }
        // cuda/texture.h:63-76 with the sampler state of cuda/main.cu:421-425 restated in
        // arithmetic: normalised coordinates, wrap addressing, point filter on a 3W x H array
        // of single bytes.
        int W = t.a, H = t.b;
        const uint8_t* img = sc.texels + t.off;
        float vf = 1.0f - v;
        uint32_t col = (uint32_t)(u * (float)W);
        float yw = vf - floorf(vf);
        int iy = (int)floorf(yw * (float)H);
        iy = iy >= H ? H - 1 : iy;
        float out[3];
        float denom = 1.0f / (float)(3 * W);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float xn = (3.0f * (float)col + 0.5f + (float)c) * denom;
            float xw = xn - floorf(xn);
            int ix = (int)floorf(xw * (float)(3 * W));
            ix = ix >= 3 * W ? 3 * W - 1 : ix;
            out[c] = (float)img[(size_t)iy * (size_t)(3 * W) + ix] / 255.0f;
        }
        return mk(out[0], out[1], out[2]);
    }
    return mk(0.0f, 0.0f, 0.0f);
}

__device__ __forceinline__ V3 material_color(const DScene& sc, const DMaterial& m, const Rec& r) {
    if (m.tex >= 0) return texture_value(sc, m.tex, r.u, r.v, r.p);
    return mk(m.r, m.g, m.b);
}

// ---- random stream -----------------------------------------------------------------------------------

struct Key { uint32_t seed_lo, seed_hi, pixel, sample; };

__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }  // 2^-24
// random_double(-1,1) = 2 u - 1 with u = k 2^-24: k 2^-23 - 1 rounded once either way, so one multiply-add gives the same bits as two steps
__device__ __forceinline__ float pm1(uint32_t w) { return fmaf((float)(w >> 8), 1.1920928955078125e-07f, -1.0f); }

// The draws of a segment beyond its Philox block: one step of the xoshiro128 linear engine
// (Blackman, Vigna) per rejection ATTEMPT, started from the block as state and stepped twice before
// the first such attempt (every state word then carries w.w, which no direct draw of a scatter
// segment uses).  An attempt's draws are s0 + s3, s1 + s2 and, for a sphere triple, s0 - s3 of the new
// state: bits 8..31, the ones a uniform uses, are exactly uniform and independent for a uniform
// state (disjoint words; s0 - s3 = (s0 + s3) - 2 s3 shares only bit 0 with s0 + s3), and the sum is
// xoshiro128+'s scrambler, weak only in its lowest bits.  A wave iterates a rejection loop until its
// unluckiest lane is done (five more trips for thirty lanes), so the loop body is what counts: ten
// integer operations per attempt instead of thirty for three calls of xoshiro128++.
struct Xo { uint32_t s0, s1, s2, s3; };
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return __builtin_rotateleft32(x, k); }
__device__ __forceinline__ void xo_step(Xo& g) {
    uint32_t t = g.s1 << 9;
    g.s2 ^= g.s0; g.s3 ^= g.s1; g.s1 ^= g.s2; g.s0 ^= g.s3;
    g.s2 ^= t;
    g.s3 = rotl32(g.s3, 11);
}

// vec3.h:130-136.  The first rejection attempt of a scatter segment reads words 0,1,2 of the
// segment's Philox block (first draw z, then y, then x: the argument evaluation order of the
// reference build, vec3.h:60); later attempts come from the stepped engine, in the same order.
__device__ __forceinline__ V3 in_unit_sphere(uint4 w) {
    V3 p = mk(pm1(w.z), pm1(w.y), pm1(w.x));
    if (dot(p, p) >= 1.0f) {
        Xo g; g.s0 = w.x; g.s1 = w.y; g.s2 = w.z; g.s3 = w.w;
        xo_step(g);
#ifdef RT_REJECT_PAIRS
        // two attempts per trip of the loop (the second one's answer is used only if the first is
        // rejected: the same point as one attempt per trip, in half the trips)
        for (int i = 0; i < 128; i++) {
            xo_step(g);
            const V3 a = mk(pm1(g.s0 - g.s3), pm1(g.s1 + g.s2), pm1(g.s0 + g.s3));
            xo_step(g);
            const V3 b = mk(pm1(g.s0 - g.s3), pm1(g.s1 + g.s2), pm1(g.s0 + g.s3));
            const bool ok_a = !(dot(a, a) >= 1.0f), ok_b = !(dot(b, b) >= 1.0f);
            p = ok_a ? a : b;
            if (ok_a || ok_b) break;
        }
#else
        for (int i = 0; i < 256; i++) {
            xo_step(g);
            float z = pm1(g.s0 + g.s3), y = pm1(g.s1 + g.s2), x = pm1(g.s0 - g.s3);
            p = mk(x, y, z);
            if (!(dot(p, p) >= 1.0f)) break;
        }
#endif
    }
    return p;
}

// ---- materials (material.h:15-92; emitters cuda/material.h:33-47,103-119) -------------------------------------

// vec3.h:159
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return madd(v, -2.0f * dot(v, n), n); }

// Returns true when the path continues along (so, sd).  `emitted` is always written.
// `w` is the Philox block of the path's current scatter segment; `rs` a point in the unit sphere
// drawn from it (both materials that need one read the same draws, so it is drawn once).
template <bool EMIT>
__device__ __forceinline__ bool scatter(const DScene& sc, const DMaterial& m, V3 ud, const Rec& r, uint4 w, V3 rs,
                                        V3& att, V3& sd, V3& emitted) {
    emitted = mk(0.0f, 0.0f, 0.0f);
    switch (EMIT ? m.type : (m.type > 2 ? 0 : m.type)) {       // (no emitters in the scene: the upload says so)
    case 0: {                                          // lambertian, material.h:21-32
        V3 s = rs;
        V3 uvec = __builtin_amdgcn_rsqf(dot(s, s)) * s;   // vec3.h:139
        V3 dir = r.n + uvec;
        const float thr = 1e-8f;                       // vec3.h:64-68
        if (fabsf(dir.x) < thr && fabsf(dir.y) < thr && fabsf(dir.z) < thr) dir = r.n;
        sd = dir;
        att = material_color(sc, m, r);
        return true;
    }
    case 1: {                                          // metal, material.h:42-47
        V3 refl = reflect(ud, r.n);                    // ud = unit_vector(r_in.direction())
        if (m.fuzz > 0.0f) refl = madd(refl, m.fuzz, rs);
        sd = refl;
        att = mk(m.r, m.g, m.b);
        return dot(sd, r.n) > 0.0f;
    }
    case 2: {                                          // dielectric, material.h:58-91
        att = mk(1.0f, 1.0f, 1.0f);
        float ratio = r.front ? m.r : m.ir;            // m.r = 1 / ir, m.g / m.b = r0 for this side (set at upload)
        float cos_theta = fminf(dot(-ud, r.n), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool tir = ratio * sin_theta > 1.0f;
        float r0 = r.front ? m.g : m.b;                // ((1 - ratio) / (1 + ratio))^2
        float x = 1.0f - cos_theta;
        float x2 = x * x;
        float refl_prob = r0 + (1.0f - r0) * (x2 * x2 * x);
        if (tir || refl_prob > u01(w.x)) {
            sd = reflect(ud, r.n);
        } else {                                       // vec3.h:163-169
            V3 perp = ratio * madd(ud, cos_theta, r.n);
            float par = -sqrtf(fabsf(1.0f - dot(perp, perp)));
            sd = madd(perp, par, r.n);
        }
        return true;
    }
    default:                                           // diffuse_light, lambertian_bg
        if constexpr (EMIT) emitted = material_color(sc, m, r);
        att = mk(0.0f, 0.0f, 0.0f);
        sd = mk(0.0f, 0.0f, 0.0f);
        return false;
    }
}

// ---- camera (camera.h:40-45, main.cc:117-118) ---------------------------------------------------------------------

// Segment 0 of a sample: its Philox block holds the two pixel jitters (words 0,1) and the first
// lens-disk attempt (word 2 -> y, word 3 -> x: vec3.h:152 evaluated right to left); further
// attempts come from the stepped xoshiro128 engine (see Xo).
template <bool XOR3, bool ORDERED>
__device__ __forceinline__ void camera_ray(const DCamera& cam /* the workgroup's LDS copy */, const DRender& rp, const Key& k, int i, int j, V3& o, V3& d) {
    uint4 w = philox4x32_10<XOR3, ORDERED>(k.pixel, k.sample, 0u, 0u, k.seed_lo, k.seed_hi);
    float s = ((float)i + u01(w.x)) * rp.inv_wm1;
    float t = ((float)j + u01(w.y)) * rp.inv_hm1;
    V3 offset = mk(0.0f, 0.0f, 0.0f);
    if (cam.lens_radius != 0.0f) {
        float py = pm1(w.z), px = pm1(w.w);
        if (fmaf(px, px, py * py) >= 1.0f) {
            Xo g; g.s0 = w.x; g.s1 = w.y; g.s2 = w.z; g.s3 = w.w;
            xo_step(g);
            for (int a = 0; a < 256; a++) {
                xo_step(g);
                py = pm1(g.s0 + g.s3); px = pm1(g.s1 + g.s2);
                if (!(fmaf(px, px, py * py) >= 1.0f)) break;
            }
        }
        float rx = cam.lens_radius * px, ry = cam.lens_radius * py;
        offset = madd(rx * ld3(cam.u), ry, ld3(cam.v));
    }
    o = ld3(cam.origin) + offset;
    d = madd(madd(ld3(cam.llc_rel), s, ld3(cam.horizontal)), t, ld3(cam.vertical)) - offset;
}

// main.cc:30-34
__device__ __forceinline__ V3 background(const DScene& sc, V3 ud) {      // ud: unit direction
    if (sc.background == 0) {
        float y = ud.y;
        float t = 0.5f * (y + 1.0f);
        return mk(fmaf(t, 0.5f, 1.0f - t), fmaf(t, 0.7f, 1.0f - t), fmaf(t, 1.0f, 1.0f - t));
    }
    return mk(sc.bg[0], sc.bg[1], sc.bg[2]);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- render kernel ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned long long clk() { return __builtin_amdgcn_s_memtime(); }

// Shade one finished closest-hit query: returns true when the path ended (sample complete).
// L collects the path's radiance: emitted light along the way and the background at the end.
template <bool COUNT, int FEAT, class G>
__device__ __forceinline__ bool shade(const DScene& sc, lds_crect* staged, const G& geo, const DRender& rp, const Cand& c, const Key& key, int depth,
                                      bool refine, const RayN& rn, V3& o, V3& d, V3& T, V3& L, Origin& from, Counters& cn) {
    constexpr bool RECTS = (FEAT & kFeatRects) != 0, TEX = (FEAT & kFeatTextures) != 0, EMIT = (FEAT & kFeatEmitters) != 0;
    // without emitters nothing is added along the way: L is written where a path ends, and only there
    if constexpr (!EMIT) L = mk(0.0f, 0.0f, 0.0f);
    if (c.kind < 0) {
        L = L + T * background(sc, rn.dh);
        return true;
    }
    // the loads the record and the material need go out first; the Philox block below is pure
    // arithmetic and runs while they are in flight
    const int mat = c.kind == 0 ? geo.mat(c.idx) : ((!RECTS || c.kind == 1) ? sc.wide[c.idx].mat : (staged_rects(FEAT) ? staged[c.idx].mat : sc.rects[c.idx].mat));
    const double4 s64 = c.kind == 0 ? sc.sph64[c.idx] : make_double4(0.0, 0.0, 0.0, 1.0);
    DMaterial m = sc.mats[mat];
    if (!TEX) m.tex = -1;          // a kernel for scenes without textures keeps none of that code
    // one Philox block per scatter segment, shared by whichever material consumes it
    // (the fused exclusive-or wherever the geometry is in LDS or the kernel carries no rect and texture
    // code: the walk out of HBM with that code is the one kernel it costs registers it does not have;
    // the kernels that read their geometry from HBM want their loads kept behind the block: rt_philox.h)
    const uint4 w = philox4x32_10<FEAT == 0 || G::kPlaneAddressed, !G::kPlaneAddressed>(key.pixel, key.sample, (uint32_t)depth, 0u, key.seed_lo, key.seed_hi);
    Rec r = make_record<FEAT>(sc, staged, c, o, d, rn.inv_len, refine, m.tex >= 0, s64, mat);
    V3 att, sd, em;
    if (COUNT) cn.shade++;
    V3 rs = mk(0.0f, 0.0f, 0.0f);
    if (m.type == 0 || (m.type == 1 && m.fuzz > 0.0f)) rs = in_unit_sphere(w);
    bool cont = scatter<EMIT>(sc, m, rn.dh, r, w, rs, att, sd, em);
    if constexpr (EMIT) L = L + T * em;
    T = T * att;
    o = r.p;
    d = sd;
    // does the new ray head into the ball?  r.n faces against the old ray: it is the outward
    // normal on a front face; a hollow sphere's outward normal points at its centre
    const bool into = (dot(sd, r.n) < 0.0f) == ((r.front != 0) != r.hollow);
    from = origin_of(c.kind, c.idx, into);
    return !cont || depth >= rp.max_depth;             // main.cc:13-14: the limit ends in black
}

// QUANT: 0 single-precision nodes, 1 the 32-byte quantised nodes, 2 those with a stack that spills to HBM (deep trees)
template <bool LDS, int QUANT>
struct GeoOf { using type = GeoGlobal; };
template <>
struct GeoOf<false, 1> { using type = GeoGlobalQT<false>; };
template <>
struct GeoOf<false, 2> { using type = GeoGlobalQT<true>; };
template <int QUANT>
struct GeoOf<true, QUANT> { using type = GeoLds; };

// Copies the BVH and the sphere list into LDS (LDS variant) behind `lds`.
template <bool LDS, int QUANT>
__device__ __forceinline__ typename GeoOf<LDS, QUANT>::type stage_geometry(const DScene& sc, int* lds) {
    if constexpr (LDS) {
        float* nodes = reinterpret_cast<float*>(lds);
        float4* sph = reinterpret_cast<float4*>(nodes + (kLdsNodeBytes / 4) * sc.n_nodes);
        int* objs = reinterpret_cast<int*>(sph + sc.n_sph_padded);
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)nodes;
        for (int i = threadIdx.x; i < sc.n_nodes; i += kBlock) {
            // device layout: left box lo xyz, hi xyz, right box lo xyz, hi xyz, left, right
            const float4 q0 = sc.nodes[4 * i], q1 = sc.nodes[4 * i + 1], q2 = sc.nodes[4 * i + 2], q3 = sc.nodes[4 * i + 3];
            const int l = __float_as_int(q3.x), r = __float_as_int(q3.y);
            float* n = nodes + (kLdsNodeBytes / 4) * i;
            n[0] = q0.x; n[1] = q0.w; n[2] = q0.x; n[3] = q1.z; n[4] = q2.y; n[5] = q1.z;
            n[6] = q0.y; n[7] = q1.x; n[8] = q0.y; n[9] = q1.w; n[10] = q2.z; n[11] = q1.w;
            n[12] = q0.z; n[13] = q1.y; n[14] = q0.z; n[15] = q2.x; n[16] = q2.w; n[17] = q2.x;
            n[18] = __int_as_float(l >= 0 ? (int)((base + (unsigned)l * kLdsNodeBytes) >> kLdsRefShift) : l);
            n[19] = __int_as_float(r >= 0 ? (int)((base + (unsigned)r * kLdsNodeBytes) >> kLdsRefShift) : r);
        }
        for (int i = threadIdx.x; i < sc.n_sph; i += kBlock) { sph[i] = sc.sph[i]; int2 f = sc.sph_info[i]; objs[i] = f.x | (f.y << 16); }
        for (int i = sc.n_sph + threadIdx.x; i < sc.n_sph_padded; i += kBlock) { sph[i] = sc.sph[i]; objs[i] = 0; }   // the scan's padding
        GeoLds g = {base, sph, objs, mk(sc.tree_reach[0], sc.tree_reach[1], sc.tree_reach[2])};
        return g;
    } else if constexpr (QUANT != 0) {
        int* strip = nullptr;
        if constexpr (QUANT == 2) strip = sc.stack_spill + ((size_t)blockIdx.x * kBlock + threadIdx.x) * (size_t)sc.stack_spill_entries;
        GeoGlobalQT<QUANT == 2> g = {sc.qnodes, sc.sph, sc.sph_info, {sc.q_lo[0], sc.q_lo[1], sc.q_lo[2]}, {sc.q_step[0], sc.q_step[1], sc.q_step[2]}, mk(sc.tree_reach[0], sc.tree_reach[1], sc.tree_reach[2]), strip};
        return g;
    } else {
        GeoGlobal g = {sc.nodes, sc.sph, sc.sph_info, mk(sc.tree_reach[0], sc.tree_reach[1], sc.tree_reach[2])};
        return g;
    }
}

// Radiance sums are kept per pixel in LDS as unsigned 32.32 fixed point and added with integer
// atomics.  Integer addition is associative, so a pixel's sum does not depend on which lane
// rendered which sample or in what order: that is what lets lanes take samples dynamically, and
// it makes an image a pure function of (scene, camera, params).  A sample is truncated to
// 2^-32 (float samples >= 2^-8 are represented exactly), so the sum is also closer to the
// reference's double accumulation (main.cc:121) than a running float sum would be.
__device__ __forceinline__ unsigned long long to_fixed(float v) {
    v = fminf(fmaxf(v, 0.0f), 1.0e9f);                 // also maps NaN to 0
    float hi = floorf(v);
    unsigned lo = (unsigned)((v - hi) * 4294967296.0f);
    return ((unsigned long long)(unsigned)hi << 32) | lo;
}
__device__ __forceinline__ float from_fixed(unsigned long long q) { return (float)((double)q * 2.3283064365386963e-10); }


// One wave renders a tile of P = 64 / lanes_per_pixel pixels, all samples.  The work units
// (pixel, sample) of the tile are numbered sample-major and handed out dynamically: whenever
// lanes are free (their path ended) they take the next units, so every trip of the loop traces
// one ray on all 64 lanes until the tile runs out of samples, whatever the path lengths were.
// Dynamic LDS: [traversal stack: stack_depth x kBlock ints, a lane's column starting with the entry that ends a walk][accumulators and pixel tables: 4 waves x (8P + 1) u64]
// then, for the LDS-resident variant, [nodes][spheres][list positions].
// MODE: 0 linear scan, 1 BVH, 2 BVH walked in time slices, 3 BVH walk that may be left between rounds of leaves.  Each combination is a kernel of its
// own: code that a scene never runs (rects, the scan, the tree walk) still costs the code that
// does run registers and scheduling freedom when it shares a kernel with it.
template <bool COUNT, bool LDS, bool PROF, int MODE, int FEAT, int QUANT>
__global__ __launch_bounds__(kBlock, MODE == 0 ? RT_WG_PER_CU_SCAN : (LDS ? (MODE == 3 ? RT_WG_PER_CU_ROUNDS : RT_WG_PER_CU) : RT_WG_PER_CU_HBM)) void render_kernel(DScene sc, DCamera cam, DRender rp, float4* __restrict__ out,
                                                        unsigned long long* __restrict__ counters) {
    constexpr bool SLICED = MODE >= 2, BVH = MODE != 0;
    extern __shared__ __align__(16) int s_dyn[];
    // (16-bit entries with the tree in LDS: see hit_bvh)
    int* stack = (LDS && MODE != 0) ? reinterpret_cast<int*>(reinterpret_cast<short*>(s_dyn) + threadIdx.x) : s_dyn + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int P = 64 / rp.lanes_per_pixel;             // pixels of this wave's tile
    // per wave: two tiles in flight (the one whose units are being handed out and the one before
    // it, whose last paths are still running), each with rgb sums per pixel (3P), a pixel table (P)
    // and the tile's first row in the framebuffer (an int)
    const int accum_words = 8 * P + 1;
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(s_dyn + rp.stack_depth * kBlock) + (threadIdx.x >> 6) * accum_words;
    // pixel table: (x, j) of the q-th pixel of the two tiles, x = -1 where a tile hangs over the image
    int2* tile_pix = reinterpret_cast<int2*>(accum + 6 * P);
    int* tile_row = reinterpret_cast<int*>(accum + 8 * P);   // first framebuffer row of the two tiles
    const auto geo = stage_geometry<LDS, QUANT>(sc, s_dyn + rp.stack_depth * kBlock + (kBlock / 64) * accum_words * 2);
    const float4* staged_list = geo.sph;               // the scan reads it when LDS (else it takes the records through scalar loads)
    (void)staged_list;
    constexpr int kDone = decltype(geo)::kDone;
    // The camera is read from an LDS copy: as kernel arguments its 19 words sit in scalar registers for
    // the whole launch (the kernels run out of those and park them in vector-register lanes, which
    // costs vector instructions on every trip), and a vector operation with a scalar operand issues at
    // half rate.
    __shared__ DCamera s_cam;
    if (threadIdx.x < sizeof(DCamera) / sizeof(float)) reinterpret_cast<float*>(&s_cam)[threadIdx.x] = reinterpret_cast<const float*>(&cam)[threadIdx.x];
    // a room's rects (the launch picks these kernels for at most kLdsRects of them)
    __shared__ __align__(16) DRect s_rects[staged_rects(FEAT) ? kLdsRects : 1];
    lds_crect* staged = (lds_crect*)(size_t)(__attribute__((address_space(3))) char*)s_rects;
    if constexpr (staged_rects(FEAT)) {
        static_assert(sizeof(DRect) == 32, "a rect record is eight words");
        if (threadIdx.x < 8 * sc.n_rects) reinterpret_cast<int*>(s_rects)[threadIdx.x] = reinterpret_cast<const int*>(sc.rects)[threadIdx.x];
    }
    // the entry below every walk's stack (see hit_bvh)
    if constexpr (BVH) {
        if constexpr (LDS) *reinterpret_cast<short*>(stack) = (short)kDone; else *stack = kDone;
    }
    __syncthreads();

    const int logP = 31 - __clz(P);
    // tile shape: 8x8, 4x8, 4x4, 4x2, 2x2, 2x1, 1x1 pixels for P = 64 ... 1 (squarer tiles measured
    // 0.5 - 3 % faster than rows of eight pixels: the lanes' primary rays lie closer together)
    const int TW = P >= 64 ? 8 : (P >= 8 ? 4 : (P >= 2 ? 2 : 1)), TH = P / TW;
    const int tiles_x = (rp.width + TW - 1) / TW;
    const int tiles_y = (rp.rows + TH - 1) / TH;
    const unsigned n_tiles = (unsigned)(tiles_x * tiles_y);
    unsigned* tile_counter = reinterpret_cast<unsigned*>(counters + kNumCounters + kNumPhaseClocks);

    Key key;
    key.seed_lo = rp.seed_lo; key.seed_hi = rp.seed_hi;
    key.pixel = 0u; key.sample = 0u;
    unsigned nrays = 0, nsamples = 0;
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool refine = (rp.flags & 2u) == 0u;
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, tq = 0;

    // Persistent waves: the grid is sized to fill the chip once and every wave keeps drawing
    // tiles from one device-wide counter until none are left, so the geometry is staged once per
    // workgroup and light (sky) and heavy tiles balance out across the waves.
    // Tiles are dealt in eight contiguous bands, one per XCD (each XCD has its own L2, so the rays
    // of one band keep re-using the BVH nodes that XCD has cached); a wave whose band has run
    // out helps with the next one.  With the geometry in LDS there is nothing to re-use in L2 and
    // one band (bottom rows, the heaviest, first) leaves the shortest tail.
    constexpr int kBands = LDS ? 1 : 8;
    int band = LDS ? 0 : (int)(__builtin_amdgcn_s_getreg(6164) & 7u), bands_tried = 0;   // HW_REG_XCC_ID[3:0]

    // A wave does not wait for the last paths of a tile before it starts on the next one: the
    // tile whose units have all been handed out ("older") keeps its accumulators while free lanes
    // already take units of the next tile ("current") into the other buffer.  All of this state is
    // wave-uniform.  Only when the current tile runs out as well while the older one still has a
    // path running (a path longer than a whole tile takes: rare) do free lanes idle.
    const unsigned tile_units = (rp.max_depth > 0) ? ((unsigned)rp.spp << logP) : 0u;   // < 2^31: the launch picks P for that
    int cur_buf = 0;                                   // buffer of the current tile; the older tile owns the other one
    bool cur_valid = false, old_valid = false, more_tiles = true;
    unsigned next = 0u;                                // first unit of the current tile not handed out yet

    // one float4 store per pixel of a finished tile
    auto write_tile = [&](int buf) {
        __builtin_amdgcn_wave_barrier();               // LDS serves one wave's accesses in issue order
        if (lane < P) {
            const int slot = buf * P + lane;
            if (tile_pix[slot].x >= 0) {
                const unsigned long long* acc = accum + slot * 3;
                float a = (float)rp.spp;
                const size_t at = (size_t)(tile_row[buf] + lane / TW) * rp.width + tile_pix[slot].x;
                if (rp.flags & 64u) {                  // the sums as they are: two 16-byte stores per pixel
                    ulonglong2* o2 = reinterpret_cast<ulonglong2*>(out) + 2 * at;
                    o2[0] = make_ulonglong2(acc[0], acc[1]);
                    o2[1] = make_ulonglong2(acc[2], (unsigned long long)rp.spp);
                } else {
                    V3 total_rgb = mk(from_fixed(acc[0]), from_fixed(acc[1]), from_fixed(acc[2]));
                    if (rp.flags & 1u) total_rgb = (1.0f / a) * total_rgb;
                    out[at] = make_float4(total_rgb.x, total_rgb.y, total_rgb.z, a);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    bool has = false;                                  // this lane is on a path
    int q = 0, depth = 0;                              // pixel slot (buffer * P + pixel of the tile), rays traced on the path
    Origin from = no_origin();                         // the primitive the ray starts on
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), T = mk(1, 1, 1), L = mk(0, 0, 0);
    // closest-hit query in progress: candidate so far and traversal position.  cur == kDone
    // on a lane with a path means "ray not started yet".
    Cand c = no_hit();
    int cur = kDone, sp = 0;
    // RT_FUSE_DRAWS bits: 1 walks in LDS left between rounds, 2 scans, 4 the other walks in LDS, 8 walks out of HBM
    constexpr bool FUSE = ((RT_FUSE_DRAWS & 1) && MODE == 3 && LDS) || ((RT_FUSE_DRAWS & 2) && MODE == 0) ||
                          ((RT_FUSE_DRAWS & 4) && (MODE == 1 || MODE == 2) && LDS) || ((RT_FUSE_DRAWS & 8) && MODE != 0 && !LDS);
    if constexpr (FUSE) {
    // The same loop with the random draws of a trip in one place.  A query that ends on a scattering surface
    // leaves its scatter pending (the surface point in o, the incoming unit direction in d, the normal and
    // the material's numbers beside them); a lane that takes a new unit is `fresh`.  Both kinds then draw
    // together: one Philox block per lane (segment 0 for a fresh lane, the path's segment for a pending
    // one) and one rejection loop (a point in the unit disk for the lens, in the unit sphere for the
    // scatter: the disk's is the sphere's with the third coordinate held at zero, the same arithmetic
    // bit for bit), instead of a block and a loop for the third of the lanes that start a sample and
    // another of each for the lanes that scatter.
    constexpr bool EMIT = (FEAT & kFeatEmitters) != 0;
    Cand& c_member = c;
    int &cur_member = cur, &sp_member = sp;
    int pend = 0;                                      // 0: nothing pending; else 1 + material type
    bool fresh = false, sflip = false;
    int fx = 0, fj = 0;
    V3 sn = mk(0, 0, 0);
    float sa = 0.0f, sb = 0.0f;
    for (;;) {
        if (PROF) tq = clk();
        if (old_valid && __ballot(has && (q >> logP) != cur_buf) == 0ull) {
            write_tile(cur_buf ^ 1);
            old_valid = false;
        }
        if (cur_valid && next >= tile_units && !old_valid) {
            old_valid = true;
            cur_buf ^= 1; cur_valid = false;
        }
        if (!cur_valid && more_tiles) {
            unsigned tile = n_tiles;
            while (bands_tried < kBands) {
                const unsigned b0 = (unsigned)(((unsigned long long)n_tiles * band) / kBands), b1 = (unsigned)(((unsigned long long)n_tiles * (band + 1)) / kBands);
                unsigned t = 0u;
                if (lane == 0) t = atomicAdd(tile_counter + band, 1u);
                t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
                if (b0 + t < b1) { tile = b0 + t; break; }
                band = (band + 1) % kBands;
                bands_tried++;
            }
            if (tile >= n_tiles) more_tiles = false;
            else {
                cur_valid = true; next = 0u;
                if (lane < P) {
                    const int slot = cur_buf * P + lane;
                    accum[slot * 3 + 0] = 0ull; accum[slot * 3 + 1] = 0ull; accum[slot * 3 + 2] = 0ull;
                    const int x = (int)(tile % (unsigned)tiles_x) * TW + (lane % TW), ly = (int)(tile / (unsigned)tiles_x) * TH + (lane / TW);
                    const int j = (rp.tile_first + (ly / rp.tile_rows) * rp.tile_step) * rp.tile_rows + (ly % rp.tile_rows);
                    tile_pix[slot] = make_int2((x < rp.width && ly < rp.rows && j < rp.height) ? x : -1, j);
                    if (lane == 0) tile_row[cur_buf] = ly;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long need = __ballot(!has);
        if (need != 0ull && cur_valid && next < tile_units) {
            const unsigned u = next + __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            if (!has && u < tile_units) {
                q = (int)(u & (unsigned)(P - 1)) + (cur_buf << logP);
                const int2 xj = tile_pix[q];
                if (xj.x >= 0) {
                    fx = xj.x; fj = xj.y;
                    key.pixel = (uint32_t)(fj * rp.width + fx);
                    key.sample = (uint32_t)rp.sample_begin + (uint32_t)(u >> logP);
                    T = mk(1.0f, 1.0f, 1.0f);
                    L = mk(0.0f, 0.0f, 0.0f);
                    depth = 0;
                    from = no_origin();
                    nsamples++;
                    has = true; fresh = true;
                }
            }
            next += (unsigned)__popcll(need);
        }
        if (PROF) { unsigned long long t = clk(); ph[0] += t - tq; tq = t; }
        if (__ballot(has) == 0ull) {
            if (!cur_valid && !old_valid && !more_tiles) break;
            continue;
        }
        if (has && (fresh || pend != 0)) {
            const uint4 w = philox4x32_10<FEAT == 0 || LDS, !LDS>(key.pixel, key.sample, fresh ? 0u : (uint32_t)depth, 0u, key.seed_lo, key.seed_hi);
            const bool lens = s_cam.lens_radius != 0.0f;
            const bool want = fresh ? lens : (pend == 1 || (pend == 2 && sa > 0.0f));
            // sphere: (X, Y, Z) = the point, first attempt from words z, y, x of the block (vec3.h:130-136, in_unit_sphere);
            // disk: X = py from word z, Y = px from word w, Z = 0 (vec3.h:150-156, camera_ray)
            // (the choice between the two by a per-lane bit mask: selects on a lane condition issue at half rate)
            const uint32_t km = fresh ? 0xffffffffu : 0u;
            float X = pm1(w.z), Y = pm1((w.w & km) | (w.y & ~km)), Z = __uint_as_float(__float_as_uint(pm1(w.x)) & ~km);
            if (want && fmaf(Z, Z, fmaf(Y, Y, X * X)) >= 1.0f) {
                Xo g; g.s0 = w.x; g.s1 = w.y; g.s2 = w.z; g.s3 = w.w;
                xo_step(g);
                for (int i = 0; i < 256; i++) {
                    xo_step(g);
                    const uint32_t ua = __float_as_uint(pm1(g.s0 + g.s3)), uc = __float_as_uint(pm1(g.s0 - g.s3));
                    Y = pm1(g.s1 + g.s2);
                    X = __uint_as_float((ua & km) | (uc & ~km));
                    Z = __uint_as_float(ua & ~km);
                    if (!(fmaf(Z, Z, fmaf(Y, Y, X * X)) >= 1.0f)) break;
                }
            }
            if (fresh) {                                   // camera.h:40-45, main.cc:117-118 (camera_ray)
                const float s = ((float)fx + u01(w.x)) * rp.inv_wm1;
                const float t = ((float)fj + u01(w.y)) * rp.inv_hm1;
                V3 offset = mk(0.0f, 0.0f, 0.0f);
                if (lens) {
                    const float rx = s_cam.lens_radius * Y, ry = s_cam.lens_radius * X;
                    offset = madd(rx * ld3(s_cam.u), ry, ld3(s_cam.v));
                }
                o = ld3(s_cam.origin) + offset;
                d = madd(madd(ld3(s_cam.llc_rel), s, ld3(s_cam.horizontal)), t, ld3(s_cam.vertical)) - offset;
                fresh = false;
            } else {                                       // material.h:21-32, 42-47, 58-91 (scatter)
                const V3 rs = mk(X, Y, Z), ud = d;
                V3 sd;
                bool cont = true;
                if (pend == 1) {
                    const V3 uvec = __builtin_amdgcn_rsqf(dot(rs, rs)) * rs;
                    V3 dir = sn + uvec;
                    const float thr = 1e-8f;
                    if (fabsf(dir.x) < thr && fabsf(dir.y) < thr && fabsf(dir.z) < thr) dir = sn;
                    sd = dir;
                } else if (pend == 2) {
                    V3 refl = reflect(ud, sn);
                    if (sa > 0.0f) refl = madd(refl, sa, rs);
                    sd = refl;
                    cont = dot(sd, sn) > 0.0f;
                } else {
                    const float ratio = sa, r0 = sb;
                    const float cos_theta = fminf(dot(-ud, sn), 1.0f);
                    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                    const bool tir = ratio * sin_theta > 1.0f;
                    const float x = 1.0f - cos_theta;
                    const float x2 = x * x;
                    const float refl_prob = r0 + (1.0f - r0) * (x2 * x2 * x);
                    if (tir || refl_prob > u01(w.x)) {
                        sd = reflect(ud, sn);
                    } else {
                        const V3 perp = ratio * madd(ud, cos_theta, sn);
                        const float par = -sqrtf(fabsf(1.0f - dot(perp, perp)));
                        sd = madd(perp, par, sn);
                    }
                }
                d = sd;
                const bool into = (dot(sd, sn) < 0.0f) == sflip;
                from.code |= (into ? 1 : 0) << 28;
                pend = 0;
                if (!cont) {                               // absorbed: the path ends with what it has collected
                    if constexpr (EMIT) {
                        atomicAdd(&accum[q * 3 + 0], to_fixed(L.x));
                        atomicAdd(&accum[q * 3 + 1], to_fixed(L.y));
                        atomicAdd(&accum[q * 3 + 2], to_fixed(L.z));
                    }
                    has = false;
                }
            }
        }
        // (what the draws have consumed is dead from here on; saying so keeps it out of the walk's registers)
        pend = 0; fresh = false; sflip = false; fx = 0; fj = 0; sa = 0.0f; sb = 0.0f; sn = mk(0.0f, 0.0f, 0.0f);
        if (PROF) { unsigned long long t = clk(); ph[3] += t - tq; tq = t; }
        if (has) {
            const RayN rn = normalise(d, rp.t_min);
            // without time slices a query starts and ends inside one trip: its state is local
            Cand c_trip = no_hit();
            int cur_trip = kDone, sp_trip = 0;
            Cand& c = *(SLICED ? &c_member : &c_trip);
            int& cur = *(SLICED ? &cur_member : &cur_trip);
            int& sp = *(SLICED ? &sp_member : &sp_trip);
            if (cur == kDone) {
                c = no_hit();
                hit_always<COUNT, FEAT>(sc, staged, o, d, rn, rp.t_min, from, c, cn);
                nrays++;
                depth++;
                if constexpr (!BVH) hit_list<COUNT, LDS>(sc, staged_list, o, rn, from, c, cn);
                else if (sc.root != kNoBvh) { hit_own_sphere(geo, o, rn, from, c); cur = geo.root_ref(sc.root); sp = 0; }
            }
            if (PROF) { unsigned long long t = clk(); ph[1] += t - tq; tq = t; }
            if constexpr (BVH)
                if (cur != kDone) hit_bvh<COUNT, MODE == 2 ? 1 : (MODE == 3 ? 2 : 0)>(geo, o, rn, from, stack, rp.stack_entries, rp.slice, rp.leaf_quorum, cur, sp, c, cn);
            if (PROF) { unsigned long long t = clk(); ph[2] += t - tq; tq = t; }
            if (cur == kDone) {                        // query finished: the record and the material, no draws
                bool ended;
                if constexpr (!EMIT) L = mk(0.0f, 0.0f, 0.0f);
                if (c.kind < 0) {
                    L = L + T * background(sc, rn.dh);
                    ended = true;
                } else {
                    constexpr bool RECTS = (FEAT & kFeatRects) != 0, TEX = (FEAT & kFeatTextures) != 0;
                    const int mat = c.kind == 0 ? geo.mat(c.idx) : ((!RECTS || c.kind == 1) ? sc.wide[c.idx].mat : (staged_rects(FEAT) ? staged[c.idx].mat : sc.rects[c.idx].mat));
                    const double4 s64 = c.kind == 0 ? sc.sph64[c.idx] : make_double4(0.0, 0.0, 0.0, 1.0);
                    DMaterial m = sc.mats[mat];
                    if (!TEX) m.tex = -1;
                    const Rec r = make_record<FEAT>(sc, staged, c, o, d, rn.inv_len, refine, m.tex >= 0, s64, mat);
                    if (COUNT) cn.shade++;
                    const int type = EMIT ? m.type : (m.type > 2 ? 0 : m.type);
                    if (type > 2) {                        // diffuse_light, lambertian_bg
                        if constexpr (EMIT) L = L + T * material_color(sc, m, r);
                        ended = true;
                    } else {
                        const V3 att = type == 0 ? material_color(sc, m, r) : (type == 1 ? mk(m.r, m.g, m.b) : mk(1.0f, 1.0f, 1.0f));
                        T = T * att;
                        ended = depth >= rp.max_depth;     // main.cc:13-14: the limit ends in black
                        o = r.p;
                        d = rn.dh;
                        sn = r.n;
                        sa = type == 1 ? m.fuzz : (r.front ? m.r : m.ir);
                        sb = r.front ? m.g : m.b;
                        sflip = (r.front != 0) != r.hollow;
                        from = origin_of(c.kind, c.idx, false);
                        pend = ended ? 0 : type + 1;
                    }
                }
                if (ended) {
                    atomicAdd(&accum[q * 3 + 0], to_fixed(L.x));
                    atomicAdd(&accum[q * 3 + 1], to_fixed(L.y));
                    atomicAdd(&accum[q * 3 + 2], to_fixed(L.z));
                    has = false;
                }
            }
            if (PROF) { unsigned long long t = clk(); ph[4] += t - tq; tq = t; }
        }
    }
    } else
    for (;;) {
        if (PROF) tq = clk();
        // the older tile's last path has ended: write its pixels
        if (old_valid && __ballot(has && (q >> logP) != cur_buf) == 0ull) {
            write_tile(cur_buf ^ 1);
            old_valid = false;
        }
        // every unit of the current tile is handed out: it becomes the older tile
        if (cur_valid && next >= tile_units && !old_valid) {
            old_valid = true;
            cur_buf ^= 1; cur_valid = false;
        }
        if (!cur_valid && more_tiles) {                    // draw a tile into the free buffer
            unsigned tile = n_tiles;
            while (bands_tried < kBands) {
                const unsigned b0 = (unsigned)(((unsigned long long)n_tiles * band) / kBands), b1 = (unsigned)(((unsigned long long)n_tiles * (band + 1)) / kBands);
                unsigned t = 0u;
                if (lane == 0) t = atomicAdd(tile_counter + band, 1u);
                t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
                if (b0 + t < b1) { tile = b0 + t; break; }
                band = (band + 1) % kBands;
                bands_tried++;
            }
            if (tile >= n_tiles) more_tiles = false;
            else {
                cur_valid = true; next = 0u;
                if (lane < P) {
                    const int slot = cur_buf * P + lane;
                    accum[slot * 3 + 0] = 0ull; accum[slot * 3 + 1] = 0ull; accum[slot * 3 + 2] = 0ull;
                    const int x = (int)(tile % (unsigned)tiles_x) * TW + (lane % TW), ly = (int)(tile / (unsigned)tiles_x) * TH + (lane / TW);
                    const int j = (rp.tile_first + (ly / rp.tile_rows) * rp.tile_step) * rp.tile_rows + (ly % rp.tile_rows);
                    tile_pix[slot] = make_int2((x < rp.width && ly < rp.rows && j < rp.height) ? x : -1, j);
                    if (lane == 0) tile_row[cur_buf] = ly;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long need = __ballot(!has);
        if (need != 0ull && cur_valid && next < tile_units) {
            const unsigned u = next + __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            if (!has && u < tile_units) {
                q = (int)(u & (unsigned)(P - 1)) + (cur_buf << logP);
                const int2 xj = tile_pix[q];
                const int x = xj.x, j = xj.y;
                if (x >= 0) {
                    key.pixel = (uint32_t)(j * rp.width + x);
                    key.sample = (uint32_t)rp.sample_begin + (uint32_t)(u >> logP);
                    camera_ray<FEAT == 0 || LDS, !LDS>(s_cam, rp, key, x, j, o, d);
                    T = mk(1.0f, 1.0f, 1.0f);
                    L = mk(0.0f, 0.0f, 0.0f);
                    depth = 0;
                    from = no_origin();
                    nsamples++;
                    has = true;
                }
            }
            next += (unsigned)__popcll(need);
        }
        if (PROF) { unsigned long long t = clk(); ph[0] += t - tq; tq = t; }
        if (__ballot(has) == 0ull) {
            // nothing in flight: either more units or tiles follow (each trip then hands out units,
            // retires a tile or draws one), or the wave is done
            if (!cur_valid && !old_valid && !more_tiles) break;
            continue;
        }
        if (has) {
            const RayN rn = normalise(d, rp.t_min);
            // without time slices a query starts and ends inside one trip: its state is local
            Cand c_trip = no_hit();
            int cur_trip = kDone, sp_trip = 0;
            Cand& cq = *(SLICED ? &c : &c_trip);
            int& curq = *(SLICED ? &cur : &cur_trip);
            int& spq = *(SLICED ? &sp : &sp_trip);
            if (curq == kDone) {                       // a new ray: always-tested primitives first
                cq = no_hit();
                hit_always<COUNT, FEAT>(sc, staged, o, d, rn, rp.t_min, from, cq, cn);
                nrays++;
                depth++;
                if constexpr (!BVH) hit_list<COUNT, LDS>(sc, staged_list, o, rn, from, cq, cn);
                else if (sc.root != kNoBvh) { hit_own_sphere(geo, o, rn, from, cq); curq = geo.root_ref(sc.root); spq = 0; }
            }
            if (PROF) { unsigned long long t = clk(); ph[1] += t - tq; tq = t; }
            if constexpr (BVH)
                if (curq != kDone) hit_bvh<COUNT, MODE == 2 ? 1 : (MODE == 3 ? 2 : 0)>(geo, o, rn, from, stack, rp.stack_entries, rp.slice, rp.leaf_quorum, curq, spq, cq, cn);
            if (PROF) { unsigned long long t = clk(); ph[2] += t - tq; tq = t; }
            if (curq == kDone) {                       // query finished (else: resumed next trip)
                if (shade<COUNT, FEAT>(sc, staged, geo, rp, cq, key, depth, refine, rn, o, d, T, L, from, cn)) {
                    atomicAdd(&accum[q * 3 + 0], to_fixed(L.x));
                    atomicAdd(&accum[q * 3 + 1], to_fixed(L.y));
                    atomicAdd(&accum[q * 3 + 2], to_fixed(L.z));
                    has = false;
                }
            }
            if (PROF) { unsigned long long t = clk(); ph[4] += t - tq; tq = t; }
        }
    }

    unsigned long long r_all = wave_sum(nrays), s_all = wave_sum(nsamples);
    if (lane == 0 && r_all) {
        atomicAdd(&counters[0], r_all);
        atomicAdd(&counters[1], s_all);
    }
    if (PROF && lane == 0)
        for (int k = 0; k < 5; k++) atomicAdd(&counters[kNumCounters + k], ph[k]);
    if (COUNT) {
        unsigned long long v[8] = {cn.box, cn.sph, cn.sph_hit, cn.rect, cn.wide, cn.shade, cn.steps, cn.wave_steps};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned long long t = wave_sum(v[k]);
            // traversal step counts share the slots of the phase clocks (a different build)
            if (lane == 0 && t) atomicAdd(&counters[k < 6 ? 2 + k : kNumCounters + (k - 6)], t);
        }
    }
}

// ---- closest-hit kernel for explicit rays (parity hook) ------------------------------------------------------------------

// Closest hit over the whole scene (QUANT: a tree is walked through its 32-byte nodes, as a frame
// walks a tree that does not fit in LDS).
template <bool COUNT, int QUANT>
__device__ __forceinline__ Cand closest_hit(const DScene& sc, V3 o, V3 d, float t_min, int* stack, Counters& cn) {
    Cand best = no_hit();
    RayN rn = normalise(d, t_min);
    hit_always<COUNT, kFeatAll>(sc, nullptr, o, d, rn, t_min, no_origin(), best, cn);
    const auto geo = stage_geometry<false, QUANT>(sc, nullptr);
    if (!sc.use_bvh) hit_list<COUNT, false>(sc, nullptr, o, rn, no_origin(), best, cn);
    else if (sc.root != kNoBvh) {
        int cur = sc.root, sp = 0;
        hit_bvh<COUNT, 0>(geo, o, rn, no_origin(), stack, kMaxStack, 0, 0, cur, sp, best, cn);
    }
    return best;
}


// The same query through the walk a frame of a small scene uses: tree and spheres staged into LDS (plane-addressed
// 80-byte nodes, 16-bit stack entries), so that explicit rays can be held against the checker on that walk too.
// Dynamic LDS: [stack: stack_depth x kBlock ints holding 16-bit entries][nodes][spheres][list positions].
__global__ __launch_bounds__(kBlock) void trace_lds_kernel(DScene sc, const float* __restrict__ rays, int n, float t_min,
                                                           uint32_t flags, int stack_depth, Hit* __restrict__ out) {
    extern __shared__ __align__(16) int s_dyn[];
    int* stack = reinterpret_cast<int*>(reinterpret_cast<short*>(s_dyn) + threadIdx.x);
    *reinterpret_cast<short*>(stack) = (short)kTravDone16;
    const GeoLds geo = stage_geometry<true, 0>(sc, s_dyn + stack_depth * kBlock);
    __syncthreads();
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0};
    Cand c = no_hit();
    const RayN rn = normalise(d, t_min);
    hit_always<false, kFeatAll>(sc, nullptr, o, d, rn, t_min, no_origin(), c, cn);
    if (sc.root != kNoBvh) {
        int cur = geo.root_ref(sc.root), sp = 0;
        hit_bvh<false, 0>(geo, o, rn, no_origin(), stack, 2 * stack_depth, 0, 0, cur, sp, c, cn);
    }
    Hit h;
    if (c.kind < 0) {
        h.id = -1; h.t = 0.0f; h.front = 0; h.u = 0.0f; h.v = 0.0f;
        for (int a = 0; a < 3; a++) { h.p[a] = 0.0f; h.n[a] = 0.0f; }
    } else {
        const double4 s64 = c.kind == 0 ? sc.sph64[c.idx] : make_double4(0.0, 0.0, 0.0, 1.0);
        Rec r = make_record<kFeatAll>(sc, nullptr, c, o, d, rn.inv_len, (flags & 2u) == 0u, true, s64, c.kind == 0 ? sc.sph_info[c.idx].y : 0);
        h.id = r.obj; h.t = r.t; h.front = r.front; h.u = r.u; h.v = r.v;
        h.p[0] = r.p.x; h.p[1] = r.p.y; h.p[2] = r.p.z;
        h.n[0] = r.n.x; h.n[1] = r.n.y; h.n[2] = r.n.z;
    }
    out[i] = h;
}

template <int QUANT>
__global__ __launch_bounds__(kBlock) void trace_kernel(DScene sc, const float* __restrict__ rays, int n, float t_min,
                                                       uint32_t flags, Hit* __restrict__ out) {
    __shared__ int s_stack[kMaxStack * kBlock];
    int* stack = s_stack + threadIdx.x;
    *stack = kTravDone;                                // the entry below the walk's stack (see hit_bvh)
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    V3 d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0};
    Cand c = closest_hit<false, QUANT>(sc, o, d, t_min, stack, cn);
    Hit h;
    if (c.kind < 0) {
        h.id = -1; h.t = 0.0f; h.front = 0; h.u = 0.0f; h.v = 0.0f;
        for (int a = 0; a < 3; a++) { h.p[a] = 0.0f; h.n[a] = 0.0f; }
    } else {
        const double4 s64 = c.kind == 0 ? sc.sph64[c.idx] : make_double4(0.0, 0.0, 0.0, 1.0);
        Rec r = make_record<kFeatAll>(sc, nullptr, c, o, d, normalise(d, t_min).inv_len, (flags & 2u) == 0u, true, s64, c.kind == 0 ? sc.sph_info[c.idx].y : 0);
        h.id = r.obj; h.t = r.t; h.front = r.front; h.u = r.u; h.v = r.v;
        h.p[0] = r.p.x; h.p[1] = r.p.y; h.p[2] = r.p.z;
        h.n[0] = r.n.x; h.n[1] = r.n.y; h.n[2] = r.n.z;
    }
    out[i] = h;
}

// ---- first-hit guides of the reconstruction stage ------------------------------------------------------------------------------

// Per pixel, the closest hit of the ray through the pixel's centre from the centre of the lens
// (camera.h:40-45 with a zero lens offset; main.cc:117-118 with the jitter at its mean): plane 0 holds
// (normal facing against the ray, distance t), plane 1 (albedo of the surface there, list position as
// int bits; -1 and white where the ray leaves the scene).  What rt_reconstruct weighs its taps by.
template <int QUANT>
__global__ __launch_bounds__(kBlock) void guide_kernel(DScene sc, DCamera cam, int width, int height, float inv_wm1, float inv_hm1,
                                                       float t_min, float4* __restrict__ out) {
    __shared__ int s_stack[kMaxStack * kBlock];
    int* stack = s_stack + threadIdx.x;
    *stack = kTravDone;
    const long long n = (long long)width * height, i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % width), y = (int)(i / width);
    const float s = ((float)x + 0.5f) * inv_wm1, t = ((float)y + 0.5f) * inv_hm1;
    const V3 o = ld3(cam.origin);
    const V3 d = madd(madd(ld3(cam.llc_rel), s, ld3(cam.horizontal)), t, ld3(cam.vertical));
    Counters cn = {0, 0, 0, 0, 0, 0, 0, 0};
    const Cand c = closest_hit<false, QUANT>(sc, o, d, t_min, stack, cn);
    float4 g0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g1 = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(-1));
    if (c.kind >= 0) {
        const double4 s64 = c.kind == 0 ? sc.sph64[c.idx] : make_double4(0.0, 0.0, 0.0, 1.0);
        const Rec r = make_record<kFeatAll>(sc, nullptr, c, o, d, normalise(d, t_min).inv_len, true, true, s64, c.kind == 0 ? sc.sph_info[c.idx].y : 0);
        const DMaterial m = sc.mats[r.mat];
        V3 a = mk(1.0f, 1.0f, 1.0f);                                   // a dielectric has no colour (its record holds 1/ir and r0 there)
        if (m.type != 2) a = material_color(sc, m, r);
        g0 = make_float4(r.n.x, r.n.y, r.n.z, r.t);
        g1 = make_float4(fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f), fmaxf(a.z, 0.0f), __int_as_float(r.obj));
    }
    out[i] = g0;
    out[n + i] = g1;
}

// ---- write_color (color.h:9-26) ------------------------------------------------------------------------------------------------

// The reference's `scale` is a float and the rest of the expression is double (color.h:17-23);
// kept that way so the bytes equal the reference's for the same sums.
__global__ __launch_bounds__(kBlock) void quantize_kernel(const float4* __restrict__ rgba, long long n, int spp,
                                                          uint8_t* __restrict__ rgb8) {
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float scale = (float)(1.0 / (double)spp);
    float4 c = rgba[i];
    float v[3] = {c.x, c.y, c.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double g = sqrt((double)scale * (double)v[a]);
        g = g < 0.0 ? 0.0 : (g > 0.999 ? 0.999 : g);
        rgb8[3 * i + a] = (uint8_t)(int)(256.0 * g);
    }
}

// Fixed-point sums (RT_FLAG_FIXED_SUMS layout: r, g, b, samples per pixel) -> float4, the conversion
// render_kernel applies to its own sums.
__global__ __launch_bounds__(kBlock) void resolve_kernel(const ulonglong2* __restrict__ sums, long long n, int mean,
                                                         float4* __restrict__ rgba) {
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const ulonglong2 a = sums[2 * i], b = sums[2 * i + 1];
    const float spp = (float)b.y;
    V3 total_rgb = mk(from_fixed(a.x), from_fixed(a.y), from_fixed(b.x));
    if (mean) total_rgb = (1.0f / spp) * total_rgb;
    rgba[i] = make_float4(total_rgb.x, total_rgb.y, total_rgb.z, spp);
}

}  // namespace

hipError_t launch_resolve(const unsigned long long* sums, long long n_pixels, bool mean, float4* rgba, hipStream_t stream) {
    if (n_pixels <= 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((n_pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       reinterpret_cast<const ulonglong2*>(sums), n_pixels, mean ? 1 : 0, rgba);
    return hipGetLastError();
}

hipError_t launch_render(const DScene& sc, const DCamera& cam, const DRender& rp, float4* out,
                         unsigned long long* counters, hipStream_t stream) {
    const int K = rp.lanes_per_pixel;
    const int P = 64 / K;
    const int TW = P >= 64 ? 8 : (P >= 8 ? 4 : (P >= 2 ? 2 : 1)), TH = P / TW;
    const long long tiles = (long long)((rp.width + TW - 1) / TW) * ((rp.rows + TH - 1) / TH);
    if (tiles <= 0) return hipSuccess;
    if (tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    long long blocks = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
    // dynamic LDS: the traversal stack (one entry per BVH level), the per-pixel accumulators and,
    // when it all fits in a budget that still leaves four workgroups per CU, a copy of the BVH
    // and the sphere list
    DRender r = rp;
    // one push per inner level at most, and the entry below the stack that ends a walk;
    // stack_depth is in units of kBlock ints
    const int entries = sc.use_bvh ? sc.bvh_depth + 2 : 1;
    const size_t accum_bytes = (size_t)(kBlock / 64) * (8 * (64 / rp.lanes_per_pixel) + 1) * sizeof(unsigned long long);
    const size_t geo_bytes = (size_t)sc.n_nodes * kLdsNodeBytes + (size_t)sc.n_sph_padded * 20;
    // the tree walk keeps tree and spheres in LDS up to 40 KB per workgroup (its stack then holds
    // 16-bit references).  The scan reads a
    // short list (a room with a few balls) from LDS: 6 % faster than through scalar loads, whose
    // latency a loop of one or two trips cannot hide; a long list is faster through scalar loads
    // (487 spheres: by 6 %), which run ahead of the arithmetic and cost no vector registers
    const int short_depth = (entries + 1) / 2;
    const bool in_lds = (rp.flags & 32u) == 0u && sc.n_sph > 0 &&
                        sc.ids_fit_16 &&                  // the LDS copy packs list position and material into one word
                        (sc.use_bvh ? sc.n_sph_padded + 8 < 4096 && (size_t)short_depth * kBlock * sizeof(int) + accum_bytes + geo_bytes <= 40 * 1024
                                    : sc.n_sph_padded <= 32);
    // Deep trees (large scenes) have a long-tailed traversal length: there a wave spends at most
    // `slice` node steps per trip on traversal, leaves earlier once half the lanes that came in
    // have finished their walk (the exit share: those lanes then shade and start new rays together,
    // at a utilisation a fixed budget alone does not give them), and unfinished lanes resume on the
    // next trip.  Slabs of 10^5 and 10^6 spheres (18 and 22 levels): 23.3 and 76.8 ms with the
    // share against 24.9 and 79.8 with the best budget alone; 64 to 128 steps then measure alike.
    // Up to 16 levels the plain walk is as fast or faster (2 x 10^4 spheres, 15 levels: 9.8 ms
    // against 10.9; 3.5 x 10^4, 16 levels: 13.7 against 14.0; 5 x 10^4, 16 levels: 16.7 against
    // 16.4; 7 x 10^4, 18 levels: 22.1 against 19.3).  The image does not depend on it.
    // A tree of at most 16 levels under nothing but spheres is walked plainly, and the walk is left
    // between two rounds of leaves once a quarter of the lanes that came in, or fewer, are unfinished
    // (one ballot a round; those lanes resume next trip, like a time slice's): the default scene
    // 18.4 -> 17.5 ms, 5 x 10^4 spheres 16.9 -> 16.2, 2 x 10^4 spheres and the default scene's
    // tree read from HBM unchanged.  With rects or textures in the scene it loses (a world of image
    // textures and boxes 14.2 -> 15.3 ms: the lanes that resume a walk sit out the always-tested
    // primitives of the lanes that start a ray), so there the walk stays whole.
    bool rounds = false;
    {
        int steps = r.slice & 0xffff, share = (r.slice >> 16) & 0xff;
        const bool chosen = steps == 0xffff;
        if (chosen) steps = sc.bvh_depth > 16 ? 64 : 0;
        rounds = chosen && steps == 0 && share > 0 && sc.use_bvh && sc.n_rects == 0 && !sc.has_textures && (rp.flags & 256u) == 0u &&   // (has_textures: any of its bits)
                 (in_lds || (sc.qnodes != nullptr && (rp.flags & 128u) == 0u));
        if (share == 255) share = steps > 0 ? kExitShare : kRoundShare;
        r.slice = (steps > 0 || rounds) ? (steps | (share << 16)) : 0;
    }
    // (with six waves per SIMD on a tree in HBM the walkers may be a little fewer before the wave
    // turns to its leaves: 10^6 spheres -2.5 % against a ratio of 2)
    if (r.leaf_quorum < 0) r.leaf_quorum = in_lds ? 2 : 4;

    const bool sliced = sc.use_bvh && (r.slice & 0xffff) > 0;
    // A tree in HBM deeper than kLdsStackCap - 2 levels: the LDS stack stops at kLdsStackCap entries (more
    // would cost a workgroup per CU, the sixth wave per SIMD that hides the fetches) and the rest of a
    // lane's stack lives in the scene's spill array (only the quantised walk is built that way)
    const int cap = rp.stack_cap >= 2 ? rp.stack_cap : kLdsStackCap;
    const bool deep = sliced && !in_lds && entries > cap && sc.stack_spill != nullptr && sc.qnodes != nullptr && (rp.flags & 128u) == 0u &&
                      entries - cap <= sc.stack_spill_entries;
    r.stack_depth = (in_lds && sc.use_bvh) ? short_depth : (deep ? cap : entries);
    r.stack_entries = (in_lds && sc.use_bvh) ? 2 * short_depth : r.stack_depth;
    size_t lds = (size_t)r.stack_depth * kBlock * sizeof(int) + accum_bytes;
    if (in_lds) lds += geo_bytes;
    // persistent waves: enough workgroups to fill 256 CUs at the kernel's occupancy, never more
    // than there are tiles
    static_assert(kSpillLanes == 256 * RT_WG_PER_CU_HBM * kBlock, "one spill strip per lane of the largest grid");
    const int wg_per_cu = sc.use_bvh ? (in_lds ? (rounds ? RT_WG_PER_CU_ROUNDS : RT_WG_PER_CU) : RT_WG_PER_CU_HBM) : RT_WG_PER_CU_SCAN;
    if (blocks > 256 * wg_per_cu) blocks = 256 * wg_per_cu;
    const bool count = (rp.flags & 4u) != 0u;
    const bool prof = (rp.flags & 8u) != 0u;
    dim3 grid((unsigned)blocks), block(kBlock);
#define RT_LAUNCH(...) hipLaunchKernelGGL((render_kernel<__VA_ARGS__>), grid, block, lds, stream, sc, cam, r, out, counters)
    // a tree read from HBM is walked through its 32-byte quantised nodes
    const bool quant = sc.use_bvh && !in_lds && sc.qnodes != nullptr && (rp.flags & 128u) == 0u;
    // what the scene holds besides plainly coloured spheres picks the kernel (see hit_always).  Instances
    // come with everything else (scenes of turned boxes are rooms with textures more often than not); the
    // counting and the phase-clock builds (diagnostics: their counts do not depend on it) exist without
    // any of it and with all of it
    int feat = (sc.n_rects > 0 ? kFeatRects : 0) | ((sc.has_textures & 1) ? kFeatTextures : 0) | ((sc.has_textures & 2) ? kFeatInstances : 0) |
               ((sc.has_textures & 4) ? kFeatEmitters : 0);
    if ((feat & kFeatInstances) || ((count || prof) && feat != 0) || (rp.flags & 256u) != 0u) feat = kFeatAll;
    else if (feat != 0 && feat != kFeatTextures) feat = (feat & kFeatTextures) ? kFeatTexturedRoom : kFeatRoom;   // the built combination that covers it
    if (staged_rects(feat) && sc.n_rects > kLdsRects) feat = kFeatAll;       // more rects than a room's: read through scalar loads
#define RT_LAUNCH_Q(C, L, P, M, F) do { if (!L && M == 2 && quant && deep) RT_LAUNCH(C, L, P, M, F, ((!L && M == 2) ? 2 : 0)); \
        else if (!L && M != 0 && quant) RT_LAUNCH(C, L, P, M, F, ((!L && M != 0) ? 1 : 0)); else RT_LAUNCH(C, L, P, M, F, 0); } while (0)
#define RT_LAUNCH_F(C, L, P, M) do { if (feat == 0) RT_LAUNCH_Q(C, L, P, M, 0); else if (feat == kFeatAll) RT_LAUNCH_Q(C, L, P, M, kFeatAll); \
        else if constexpr (!C && !P) { if (feat == kFeatRoom) RT_LAUNCH_Q(C, L, P, M, kFeatRoom); else if (feat == kFeatTextures) RT_LAUNCH_Q(C, L, P, M, kFeatTextures); \
                                       else RT_LAUNCH_Q(C, L, P, M, kFeatTexturedRoom); } } while (0)
#define RT_LAUNCH_M(C, P) do { \
        if (!sc.use_bvh) { if (in_lds) RT_LAUNCH_F(C, true, P, 0); else RT_LAUNCH_F(C, false, P, 0); } \
        else if (in_lds) { if (sliced) RT_LAUNCH_F(C, true, P, 2); else if (rounds) RT_LAUNCH(C, true, P, 3, 0, 0); else RT_LAUNCH_F(C, true, P, 1); } \
        else { if (sliced) RT_LAUNCH_F(C, false, P, 2); else if (rounds) RT_LAUNCH(C, false, P, 3, 0, 1); else RT_LAUNCH_F(C, false, P, 1); } \
    } while (0)
    if (prof) RT_LAUNCH_M(false, true);
    else if (count) RT_LAUNCH_M(true, false);
    else RT_LAUNCH_M(false, false);
#undef RT_LAUNCH_M
#undef RT_LAUNCH_F
#undef RT_LAUNCH_Q
#undef RT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_trace(const DScene& sc, const float* rays, int n, float t_min, uint32_t flags, Hit* out,
                        hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    // flags & 512: the walk a frame of this scene uses: tree and spheres in LDS where they fit (launch_render's rule)
    if ((flags & 512u) != 0u && (flags & 32u) == 0u && sc.use_bvh && sc.n_sph > 0 && sc.ids_fit_16 && sc.n_sph_padded + 8 < 4096) {
        const int short_depth = (sc.bvh_depth + 2 + 1) / 2;
        const size_t lds = (size_t)short_depth * kBlock * sizeof(int) + (size_t)sc.n_nodes * kLdsNodeBytes + (size_t)sc.n_sph_padded * 20;
        if (lds <= 40 * 1024) {
            hipLaunchKernelGGL(trace_lds_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, stream, sc, rays, n, t_min, flags, short_depth, out);
            return hipGetLastError();
        }
    }
    // else the walk a frame of this scene uses when the tree does not fit in LDS
    if (sc.use_bvh && sc.qnodes != nullptr && (flags & 128u) == 0u)
        hipLaunchKernelGGL(trace_kernel<1>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sc, rays, n, t_min, flags, out);
    else
        hipLaunchKernelGGL(trace_kernel<0>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sc, rays, n, t_min, flags, out);
    return hipGetLastError();
}

hipError_t launch_guides(const DScene& sc, const DCamera& cam, int width, int height, float t_min, float4* out, hipStream_t stream) {
    const long long n = (long long)width * height;
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    const float inv_wm1 = (float)(1.0 / (double)(width - 1)), inv_hm1 = (float)(1.0 / (double)(height - 1));
    if (sc.use_bvh && sc.qnodes != nullptr)
        hipLaunchKernelGGL(guide_kernel<1>, grid, block, 0, stream, sc, cam, width, height, inv_wm1, inv_hm1, t_min, out);
    else
        hipLaunchKernelGGL(guide_kernel<0>, grid, block, 0, stream, sc, cam, width, height, inv_wm1, inv_hm1, t_min, out);
    return hipGetLastError();
}

hipError_t launch_quantize(const float4* rgba, long long n_pixels, int spp, uint8_t* rgb8, hipStream_t stream) {
    if (n_pixels <= 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((n_pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       rgba, n_pixels, spp, rgb8);
    return hipGetLastError();
}

}  // namespace rt
