// BVH construction on the GPU (interface: rt_lbvh.h).
//
//   bounds      centroid bounds of all spheres (wave reduction + ordered-integer atomics)
//   morton      63-bit Morton key per sphere on a cubic grid over those bounds
//   sort        rocPRIM radix sort of (key, sphere index)
//   hierarchy   binary radix tree over the sorted keys: node i owns a contiguous range of the
//               sorted spheres and splits it where the keys' common prefix ends (one thread
//               per inner node, no dependence between nodes)
//   collapse    subtrees of <= leaf_max spheres become leaves; the surviving inner nodes are
//               renumbered by a prefix sum
//   fit         boxes bottom-up: the second thread to arrive at a node owns it and goes on
//   assemble    64-byte nodes holding both children's boxes (layout: rt_device.h)
#include "rt_lbvh.h"

#include "rt_bvh.h"

#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace rt {

namespace {

constexpr int kThreads = 256;
constexpr int kLeafMax = 4;

// ---- helpers ---------------------------------------------------------------------------------

// float <-> unsigned with the same order, so that atomicMin/atomicMax on the integers order floats
__device__ __forceinline__ unsigned ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float unordered(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

// cuda/sphere.h:30-33 (centre -+ radius), padded exactly as the host builder pads: outwards by a
// relative 1e-5 of the radius and one ulp, so that single-precision rounding in the slab test
// can never cut off a sphere the double-precision scan would hit
__device__ __forceinline__ void sphere_box(double4 s, float* lo, float* hi) {
    double r = fabs(s.w) * (1.0 + 1e-5) + 1e-30;
    double c[3] = {s.x, s.y, s.z};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = nextafterf((float)(c[a] - r), -INFINITY);
        hi[a] = nextafterf((float)(c[a] + r), INFINITY);
    }
}

// bits of a 21-bit number moved to every third position
__device__ __forceinline__ unsigned long long spread21(unsigned v) {
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// ---- kernels -----------------------------------------------------------------------------------

__global__ void init_bounds_kernel(unsigned* lohi) {
    if (threadIdx.x < 3) lohi[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) lohi[threadIdx.x] = 0u;
}

__global__ __launch_bounds__(kThreads) void bounds_kernel(const double4* __restrict__ sph, int n, unsigned* lohi) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        double4 s = sph[i];
        float c[3] = {(float)s.x, (float)s.y, (float)s.z};
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], c[a]); hi[a] = fmaxf(hi[a], c[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            atomicMin(&lohi[a], ordered(lo[a]));
            atomicMax(&lohi[3 + a], ordered(hi[a]));
        }
    }
}

__global__ __launch_bounds__(kThreads) void morton_kernel(const double4* __restrict__ sph, int n, const unsigned* __restrict__ lohi,
                                                         unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    float lo[3] = {unordered(lohi[0]), unordered(lohi[1]), unordered(lohi[2])};
    float ext = fmaxf(fmaxf(unordered(lohi[3]) - lo[0], unordered(lohi[4]) - lo[1]), unordered(lohi[5]) - lo[2]);
    // one scale for the three axes: cubic cells, so that a split by key prefix is a split by
    // position whatever the shape of the scene
    float scale = ext > 0.0f ? 2097152.0f / ext : 0.0f;
    double4 s = sph[i];
    float c[3] = {(float)s.x, (float)s.y, (float)s.z};
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float v = (c[a] - lo[a]) * scale;
        v = fminf(fmaxf(v, 0.0f), 2097151.0f);
        q[a] = (unsigned)v;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; equal keys are told
// apart by their positions, so every pair differs somewhere; -1 outside the array
__device__ __forceinline__ int prefix(const unsigned long long* __restrict__ keys, int n, int i, unsigned long long ki, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long x = ki ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Inner node i of the radix tree over n sorted keys (n - 1 inner nodes; node 0 is the root).
// child: >= 0 inner node, < 0 single sphere ~k at sorted position k.  parent is indexed by
// "unified" node number: inner node i -> i, sorted position k -> (n - 1) + k.
__global__ __launch_bounds__(kThreads) void hierarchy_kernel(const unsigned long long* __restrict__ keys, int n,
                                                            int2* __restrict__ child, int2* __restrict__ range,
                                                            int* __restrict__ parent) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1) return;
    const unsigned long long ki = keys[i];
    // direction in which the node's range extends from i, and the prefix every key outside shares less of
    int d = prefix(keys, n, i, ki, i + 1) - prefix(keys, n, i, ki, i - 1) >= 0 ? 1 : -1;
    int dmin = prefix(keys, n, i, ki, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, ki, i + lmax * d) > dmin) lmax <<= 1;      // ends: prefix() is -1 outside the array
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (prefix(keys, n, i, ki, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    // the split: last position that still shares more than the node's own prefix with i
    int dnode = prefix(keys, n, i, ki, j);
    int s = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (prefix(keys, n, i, ki, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int g = i + s * d + (d < 0 ? -1 : 0);
    int first = i < j ? i : j, last = i < j ? j : i;
    int lc = first == g ? ~g : g;
    int rc = last == g + 1 ? ~(g + 1) : g + 1;
    child[i] = make_int2(lc, rc);
    range[i] = make_int2(first, last);
    parent[lc >= 0 ? lc : (n - 1) + ~lc] = i;
    parent[rc >= 0 ? rc : (n - 1) + ~rc] = i;
    if (i == 0) parent[0] = -1;
}

__global__ __launch_bounds__(kThreads) void kept_kernel(const int2* __restrict__ range, int n, int leaf_max, int* __restrict__ kept) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1) return;
    int2 r = range[i];
    kept[i] = (r.y - r.x + 1) > leaf_max ? 1 : 0;
}

// Boxes and heights bottom-up.  A thread starts at every subtree that becomes a leaf (<= leaf_max
// spheres under a parent that holds more) and climbs; at each inner node the first of its two
// arrivals stops and the second, which finds the sibling's result published, goes on.  Every
// thread stops at the latest at the root.  Results of other threads are read with agent-scope
// atomic loads so that they come from L2, not from a stale line in this CU's vector cache.
__global__ __launch_bounds__(kThreads) void fit_kernel(const double4* __restrict__ sph, const int* __restrict__ vals, int n,
                                                      const int2* __restrict__ child, const int2* __restrict__ range,
                                                      const int* __restrict__ parent, unsigned* __restrict__ kbox,
                                                      int* __restrict__ height, int* __restrict__ arrivals, int* __restrict__ depth_out, int leaf_max) {
    int u = blockIdx.x * kThreads + threadIdx.x;
    if (u >= 2 * n - 1) return;
    int first, count;
    if (u < n - 1) { int2 r = range[u]; first = r.x; count = r.y - r.x + 1; }
    else { first = u - (n - 1); count = 1; }
    if (count > leaf_max) return;
    int p = parent[u];
    if (p < 0) return;                                    // cannot happen for n > 4: the root holds more than 4
    { int2 r = range[p]; if (r.y - r.x + 1 <= leaf_max) return; }   // inside a subtree that is collapsed higher up

    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = first; k < first + count; k++) {
        float l[3], h[3];
        sphere_box(sph[vals[k]], l, h);
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], l[a]); hi[a] = fmaxf(hi[a], h[a]); }
    }
    int h = 0, cur = u;
    for (int guard = 0; guard < 128; guard++) {          // a path to the root is at most 64 + 32 nodes long
        unsigned* b = kbox + 6 * (size_t)cur;
#pragma unroll
        for (int a = 0; a < 3; a++) { b[a] = __float_as_uint(lo[a]); b[3 + a] = __float_as_uint(hi[a]); }
        height[cur] = h;
        if (p < 0) { *depth_out = h; return; }            // cur is the root
        __threadfence();                                  // publish before announcing the arrival
        if (atomicAdd(&arrivals[p], 1) == 0) return;      // the sibling's thread will take it from here
        int2 ch = child[p];
        int cl = ch.x >= 0 ? ch.x : (n - 1) + ~ch.x, cr = ch.y >= 0 ? ch.y : (n - 1) + ~ch.y;
        int sib = cl == cur ? cr : cl;
        const unsigned* sb = kbox + 6 * (size_t)sib;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], __uint_as_float(__hip_atomic_load(sb + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
            hi[a] = fmaxf(hi[a], __uint_as_float(__hip_atomic_load(sb + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        }
        int hs = __hip_atomic_load(height + sib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h = 1 + (h > hs ? h : hs);
        cur = p;
        p = parent[cur];
    }
}

__global__ __launch_bounds__(kThreads) void assemble_kernel(int n, const int2* __restrict__ child, const int2* __restrict__ range,
                                                           const int* __restrict__ kept, const int* __restrict__ newidx,
                                                           const unsigned* __restrict__ kbox, float4* __restrict__ nodes) {
    int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1 || !kept[i]) return;
    int2 ch = child[i];
    int code[2];
    float bx[2][6];
#pragma unroll
    for (int side = 0; side < 2; side++) {
        int c = side == 0 ? ch.x : ch.y;
        int u;
        if (c < 0) { u = (n - 1) + ~c; code[side] = ~(((~c) << kLeafBits) | 0); }
        else {
            u = c;
            if (kept[c]) code[side] = newidx[c];
            else { int2 r = range[c]; code[side] = ~((r.x << kLeafBits) | (r.y - r.x)); }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) bx[side][a] = __uint_as_float(kbox[6 * (size_t)u + a]);
    }
    float4* q = nodes + 4 * (size_t)newidx[i];
    q[0] = make_float4(bx[0][0], bx[0][1], bx[0][2], bx[0][3]);
    q[1] = make_float4(bx[0][4], bx[0][5], bx[1][0], bx[1][1]);
    q[2] = make_float4(bx[1][2], bx[1][3], bx[1][4], bx[1][5]);
    q[3] = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), 0.0f, 0.0f);
}

__global__ __launch_bounds__(kThreads) void gather_kernel(const int* __restrict__ order, int n, const float4* __restrict__ sph,
                                                         const double4* __restrict__ sph64, const int2* __restrict__ info,
                                                         const int* __restrict__ xform, float4* __restrict__ sph_out,
                                                         double4* __restrict__ sph64_out, int2* __restrict__ info_out,
                                                         int* __restrict__ xform_out) {
    int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n) return;
    int i = order[k];
    sph_out[k] = sph[i];
    sph64_out[k] = sph64[i];
    info_out[k] = info[i];
    if (xform) xform_out[k] = xform[i];
}

struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
    void release(void* p) {                               // ownership passes to the caller
        for (auto& q : ptrs) if (q == p) { q = ptrs.back(); ptrs.pop_back(); return; }
    }
};

inline int blocks(long long n) { return (int)((n + kThreads - 1) / kThreads); }

}  // namespace

#define LB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

hipError_t build_lbvh(const double4* sph64, int n, int leaf_max, hipStream_t stream, LbvhResult* out) {
    *out = LbvhResult{};
    if (leaf_max < 1 || leaf_max > kLeafMax || n <= kLeafMax || n > (1 << 28)) return hipErrorInvalidValue;
    Scratch mem;
    unsigned* lohi; unsigned long long *keys, *keys_sorted; int *vals, *vals_sorted;
    int2 *child, *range; int *parent, *kept, *newidx, *arrivals, *height, *depth_dev; unsigned* kbox;
    LB(mem.get(&lohi, 6));
    LB(mem.get(&keys, n)); LB(mem.get(&keys_sorted, n)); LB(mem.get(&vals, n)); LB(mem.get(&vals_sorted, n));
    LB(mem.get(&child, n - 1)); LB(mem.get(&range, n - 1)); LB(mem.get(&parent, 2 * (size_t)n - 1));
    LB(mem.get(&kept, n - 1)); LB(mem.get(&newidx, n - 1)); LB(mem.get(&arrivals, n - 1));
    LB(mem.get(&height, 2 * (size_t)n - 1)); LB(mem.get(&depth_dev, 1)); LB(mem.get(&kbox, 6 * (2 * (size_t)n - 1)));

    hipLaunchKernelGGL(init_bounds_kernel, dim3(1), dim3(64), 0, stream, lohi);
    hipLaunchKernelGGL(bounds_kernel, dim3(blocks(n) < 1024 ? blocks(n) : 1024), dim3(kThreads), 0, stream, sph64, n, lohi);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks(n)), dim3(kThreads), 0, stream, sph64, n, lohi, keys, vals);
    LB(hipGetLastError());

    size_t tmp_bytes = 0;
    LB(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys_sorted, vals, vals_sorted, (size_t)n, 0u, 63u, stream));
    size_t scan_bytes = 0;
    LB(rocprim::exclusive_scan(nullptr, scan_bytes, kept, newidx, 0, (size_t)(n - 1), rocprim::plus<int>(), stream));
    char* tmp;
    LB(mem.get(&tmp, tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes));
    LB(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_sorted, vals, vals_sorted, (size_t)n, 0u, 63u, stream));

    hipLaunchKernelGGL(hierarchy_kernel, dim3(blocks(n - 1)), dim3(kThreads), 0, stream, keys_sorted, n, child, range, parent);
    hipLaunchKernelGGL(kept_kernel, dim3(blocks(n - 1)), dim3(kThreads), 0, stream, range, n, leaf_max, kept);
    LB(hipGetLastError());
    LB(rocprim::exclusive_scan(tmp, scan_bytes, kept, newidx, 0, (size_t)(n - 1), rocprim::plus<int>(), stream));
    LB(hipMemsetAsync(arrivals, 0, (size_t)(n - 1) * sizeof(int), stream));
    LB(hipMemsetAsync(depth_dev, 0, sizeof(int), stream));
    hipLaunchKernelGGL(fit_kernel, dim3(blocks(2 * (long long)n - 1)), dim3(kThreads), 0, stream, sph64, vals_sorted, n, child, range,
                       parent, kbox, height, arrivals, depth_dev, leaf_max);
    LB(hipGetLastError());

    int last_kept = 0, last_idx = 0, depth = 0;
    LB(hipMemcpyAsync(&last_kept, kept + (n - 2), sizeof(int), hipMemcpyDeviceToHost, stream));
    LB(hipMemcpyAsync(&last_idx, newidx + (n - 2), sizeof(int), hipMemcpyDeviceToHost, stream));
    LB(hipMemcpyAsync(&depth, depth_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    LB(hipStreamSynchronize(stream));
    const int n_nodes = last_idx + last_kept;
    if (n_nodes < 1 || depth < 1) return hipErrorUnknown;

    float4* nodes;
    LB(mem.get(&nodes, 4 * (size_t)n_nodes));
    hipLaunchKernelGGL(assemble_kernel, dim3(blocks(n - 1)), dim3(kThreads), 0, stream, n, child, range, kept, newidx, kbox, nodes);
    LB(hipGetLastError());
    LB(hipStreamSynchronize(stream));                     // the scratch arrays are freed on return

    mem.release(nodes);
    mem.release(vals_sorted);
    out->nodes = nodes;
    out->order = vals_sorted;
    out->n_nodes = n_nodes;
    out->root = 0;
    out->depth = depth;
    return hipSuccess;
}

// One thread per node.  The plane a traversal works with is lo + q * step in exact arithmetic (it
// folds lo and step into the ray's constants and rounds once); q is chosen here, in double, so that
// this exact plane lies on the outer side of the node's own plane.
constexpr double kCoarseArea = 2.0;

__global__ __launch_bounds__(kThreads) void quantize_kernel(const float4* __restrict__ nodes, int n, float lx, float ly, float lz,
                                                            float sx, float sy, float sz, uint4* __restrict__ out, int* __restrict__ n_coarse) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float4 q0 = nodes[4 * (size_t)i], q1 = nodes[4 * (size_t)i + 1], q2 = nodes[4 * (size_t)i + 2], q3 = nodes[4 * (size_t)i + 3];
    const float blo[2][3] = {{q0.x, q0.y, q0.z}, {q1.z, q1.w, q2.x}}, bhi[2][3] = {{q0.w, q1.x, q1.y}, {q2.y, q2.z, q2.w}};
    const double lo[3] = {lx, ly, lz}, st[3] = {sx, sy, sz};
    unsigned w[6];
    bool is_coarse = false;
    for (int c = 0; c < 2; c++) {
        double ext[3], qext[3];
        for (int a = 0; a < 3; a++) {
            double ql = floor(((double)blo[c][a] - lo[a]) / st[a]), qh = ceil(((double)bhi[c][a] - lo[a]) / st[a]);
            if (lo[a] + ql * st[a] > (double)blo[c][a]) ql -= 1.0;
            if (lo[a] + qh * st[a] < (double)bhi[c][a]) qh += 1.0;
            ql = ql < 0.0 ? 0.0 : (ql > 65535.0 ? 65535.0 : ql);
            qh = qh < 0.0 ? 0.0 : (qh > 65535.0 ? 65535.0 : qh);
            w[3 * c + a] = (unsigned)ql | ((unsigned)qh << 16);
            ext[a] = (double)bhi[c][a] - (double)blo[c][a];
            qext[a] = (qh - ql) * st[a];
        }
        const double area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
        const double qarea = qext[0] * qext[1] + qext[1] * qext[2] + qext[2] * qext[0];
        is_coarse = is_coarse || !(qarea <= kCoarseArea * area);
    }
    out[2 * (size_t)i] = make_uint4(w[0], w[1], w[2], w[3]);
    out[2 * (size_t)i + 1] = make_uint4(w[4], w[5], __float_as_uint(q3.x), __float_as_uint(q3.y));
    if (is_coarse) atomicAdd(n_coarse, 1);
}

hipError_t quantize_nodes(const float4* nodes, int n_nodes, const float lo[3], const float step[3], uint4* out, int* n_coarse,
                          hipStream_t stream) {
    *n_coarse = 0;
    if (n_nodes <= 0) return hipSuccess;
    int* count = nullptr;
    hipError_t e = hipMalloc(&count, sizeof(int));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(count, 0, sizeof(int), stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(quantize_kernel, dim3(blocks(n_nodes)), dim3(kThreads), 0, stream, nodes, n_nodes, lo[0], lo[1], lo[2],
                           step[0], step[1], step[2], out, count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(n_coarse, count, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(count);
    return e;
}

hipError_t gather_spheres(const int* order, int n, const float4* sph, const double4* sph64, const int2* info,
                          const int* xform, float4* sph_out, double4* sph64_out, int2* info_out, int* xform_out,
                          hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3(blocks(n)), dim3(kThreads), 0, stream, order, n, sph, sph64, info, xform,
                       sph_out, sph64_out, info_out, xform_out);
    return hipGetLastError();
}

}  // namespace rt
