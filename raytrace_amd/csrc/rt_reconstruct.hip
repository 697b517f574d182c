// Low-to-high sample-count reconstruction of a frame (rt_reconstruct of include/rt.h): the edge-avoiding
// a-trous wavelet transform of Dammertz, Sewtz, Hanika and Lensch (HPG 2010) over the frame's mean
// radiance, guided by what the ray through each pixel's centre hits first (rt_render_guides: normal,
// distance, list position, albedo).  The reference names such a stage as ongoing work (README.md:7) and
// holds no code of it; the frame pairs it is meant for are those of cuda/main.cu:557-576.
//
// N passes of a 5x5 B3-spline kernel whose taps stand 2^i pixels apart in pass i.  A tap counts by
//   h(dx) h(dy) * [same object] * cos(normals)^sigma_normal * exp(-|t_p - t_q| / (sigma_depth t_p))
//             * exp(-|c_p - c_q|^2 / (sigma_color 2^-i)^2)
// on the colour demodulated by the first hit's albedo (texture detail is divided out before the
// passes and multiplied back in after the last one).
//
// Bandwidth-bound: per pass and pixel 36 bytes in (colour, normal + distance, object) and 16 out once,
// and 25 taps through L2.  The blockIdx -> tile map hands each XCD one horizontal band of the image
// (workgroups go round the eight XCDs, each with its own L2), so the rows a tile's taps reach above and
// below it are rows its own XCD's other tiles read as well.
#include "rt_device.h"

namespace rt {
namespace {

constexpr int kTileW = 32, kTileH = 8, kThreads = kTileW * kTileH;

// mean radiance / albedo; the sample count rides along in w
__global__ __launch_bounds__(kThreads) void demodulate_kernel(const float4* __restrict__ rgba, const float4* __restrict__ g1,
                                                              long long n, float albedo_floor, float4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float4 c = rgba[i], a = g1[i];
    const float inv = 1.0f / fmaxf(c.w, 1.0f);
    out[i] = make_float4(c.x * inv / fmaxf(a.x, albedo_floor), c.y * inv / fmaxf(a.y, albedo_floor),
                         c.z * inv / fmaxf(a.z, albedo_floor), c.w);
}

// One pass.  LAST: the result is multiplied by albedo and sample count again (the convention of rt_render's images).
template <bool LAST>
__global__ __launch_bounds__(kThreads) void atrous_kernel(const float4* __restrict__ cin, const float4* __restrict__ g0,
                                                          const float4* __restrict__ g1, int width, int height, int step,
                                                          float sigma_color, float pass_scale /* 4^pass */, float sigma_normal,
                                                          float inv_sigma_depth, float albedo_floor, int band_rows, int tiles_x,
                                                          float4* __restrict__ cout) {
    // workgroup b runs on XCD b % 8: that XCD's band of tile rows, tiles in row-major order within it
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ty = xcd * band_rows + slot / tiles_x, tx = slot % tiles_x;
    if (slot / tiles_x >= band_rows) return;
    const int x = tx * kTileW + (threadIdx.x % kTileW), y = ty * kTileH + (threadIdx.x / kTileW);
    if (x >= width || y >= height) return;
    const long long at = (long long)y * width + x;
    const float4 cp = cin[at], gp = g0[at];
    const int idp = __float_as_int(g1[at].w);
    const bool hit = idp >= 0;
    const float sc = sigma_color > 0.0f ? sigma_color : 1.6f * __builtin_amdgcn_rsqf(fmaxf(cp.w, 1.0f));
    const float k_color = -1.4426950408889634f * pass_scale / (sc * sc);          // exponents are taken to base 2
    const float k_depth = hit ? -1.4426950408889634f * inv_sigma_depth / fmaxf(gp.w, 1.0e-6f) : 0.0f;
    const float k_normal = hit ? sigma_normal : 0.0f;
    const float b3[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float ax = 0.0f, ay = 0.0f, az = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
        const int qy = y + dy * step;
        if (qy < 0 || qy >= height) continue;
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
            const int qx = x + dx * step;
            if (qx < 0 || qx >= width) continue;
            const long long q = (long long)qy * width + qx;
            const int idq = __float_as_int(g1[q].w);
            if (idq != idp) continue;
            const float4 cq = cin[q], gq = g0[q];
            const float cosn = fminf(fmaxf(fmaf(gp.z, gq.z, fmaf(gp.y, gq.y, gp.x * gq.x)), 1.0e-6f), 1.0f);
            const float ex = cq.x - cp.x, ey = cq.y - cp.y, ez = cq.z - cp.z;
            const float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            const float e = fmaf(k_normal, __builtin_amdgcn_logf(cosn), fmaf(k_depth, fabsf(gp.w - gq.w), k_color * d2));
            const float w = b3[dy + 2] * b3[dx + 2] * __builtin_amdgcn_exp2f(e);
            ax = fmaf(w, cq.x, ax); ay = fmaf(w, cq.y, ay); az = fmaf(w, cq.z, az);
            wsum += w;
        }
    }
    const float inv = 1.0f / wsum;                   // the centre tap always counts: wsum >= 9/64
    float4 r = make_float4(ax * inv, ay * inv, az * inv, cp.w);
    if (LAST) {
        const float4 a = g1[at];
        r.x *= fmaxf(a.x, albedo_floor) * cp.w; r.y *= fmaxf(a.y, albedo_floor) * cp.w; r.z *= fmaxf(a.z, albedo_floor) * cp.w;
    }
    cout[at] = r;
}

// The same pass with its taps read from LDS: the tile's neighbourhood, (TW + 4 step) x (TH + 4 step) pixels, is staged
// once (eight planes: colour, normal, distance, object) with coalesced loads, and the 25 taps of every pixel are LDS
// reads; through the caches each pixel was fetched 25 times over.  Steps 1, 2 and 4 take 32x8 tiles (14 to 37 KB);
// step 8 a 64x16 tile of 1024 threads, one workgroup per CU (147 KB); wider steps stay on the caches.  Pixels beyond
// the image carry an object number no pixel has, so they fail the first test like any tap on another object.
// FIRST: the input is the frame as rt_render leaves it, and staging divides it by sample count and albedo (the
// demodulation pass is folded into the first a-trous pass).  Same arithmetic in the same order as demodulate_kernel
// and atrous_kernel: the same bits.
constexpr int kNoPixel = (int)0x80000000;
template <bool FIRST, bool LAST, int TW, int TH>
__global__ __launch_bounds__(TW * TH) void atrous_lds_kernel(const float4* __restrict__ cin, const float4* __restrict__ g0,
                                                             const float4* __restrict__ g1, int width, int height, int step,
                                                             float sigma_color, float pass_scale, float sigma_normal,
                                                             float inv_sigma_depth, float albedo_floor, int band_rows, int tiles_x,
                                                             float4* __restrict__ cout) {
    extern __shared__ float s_tile[];
    constexpr int kWg = TW * TH;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ty = xcd * band_rows + slot / tiles_x, tx = slot % tiles_x;
    if (slot / tiles_x >= band_rows) return;
    const int halo = 2 * step, rw = TW + 2 * halo, rh = TH + 2 * halo, rn = rw * rh;
    float* pc0 = s_tile; float* pc1 = pc0 + rn; float* pc2 = pc1 + rn;
    float* pn0 = pc2 + rn; float* pn1 = pn0 + rn; float* pn2 = pn1 + rn;
    float* pt = pn2 + rn;
    int* pid = reinterpret_cast<int*>(pt + rn);
    const int x0 = tx * TW - halo, y0 = ty * TH - halo;
    for (int i = threadIdx.x; i < rn; i += kWg) {
        const int qx = x0 + i % rw, qy = y0 + i / rw;
        if (qx >= 0 && qx < width && qy >= 0 && qy < height) {
            const long long q = (long long)qy * width + qx;
            const float4 c = cin[q], g = g0[q];
            if (FIRST) {
                const float4 a = g1[q];
                const float inv = 1.0f / fmaxf(c.w, 1.0f);
                pc0[i] = c.x * inv / fmaxf(a.x, albedo_floor); pc1[i] = c.y * inv / fmaxf(a.y, albedo_floor);
                pc2[i] = c.z * inv / fmaxf(a.z, albedo_floor);
                pid[i] = __float_as_int(a.w);
            } else {
                pc0[i] = c.x; pc1[i] = c.y; pc2[i] = c.z;
                pid[i] = __float_as_int(g1[q].w);
            }
            pn0[i] = g.x; pn1[i] = g.y; pn2[i] = g.z; pt[i] = g.w;
        } else {
            pid[i] = kNoPixel;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
    const int x = tx * TW + lx, y = ty * TH + ly;
    if (x >= width || y >= height) return;
    const long long at = (long long)y * width + x;
    const int ci = (ly + halo) * rw + lx + halo;
    const float cpx = pc0[ci], cpy = pc1[ci], cpz = pc2[ci], gpx = pn0[ci], gpy = pn1[ci], gpz = pn2[ci], gpw = pt[ci];
    const int idp = pid[ci];
    const float spp = cin[at].w;
    const bool hit = idp >= 0;
    const float sc = sigma_color > 0.0f ? sigma_color : 1.6f * __builtin_amdgcn_rsqf(fmaxf(spp, 1.0f));
    const float k_color = -1.4426950408889634f * pass_scale / (sc * sc);
    const float k_depth = hit ? -1.4426950408889634f * inv_sigma_depth / fmaxf(gpw, 1.0e-6f) : 0.0f;
    const float k_normal = hit ? sigma_normal : 0.0f;
    const float b3[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float ax = 0.0f, ay = 0.0f, az = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
            const int q = ci + (dy * rw + dx) * step;
            if (pid[q] != idp) continue;
            const float cqx = pc0[q], cqy = pc1[q], cqz = pc2[q];
            const float cosn = fminf(fmaxf(fmaf(gpz, pn2[q], fmaf(gpy, pn1[q], gpx * pn0[q])), 1.0e-6f), 1.0f);
            const float ex = cqx - cpx, ey = cqy - cpy, ez = cqz - cpz;
            const float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
            const float e = fmaf(k_normal, __builtin_amdgcn_logf(cosn), fmaf(k_depth, fabsf(gpw - pt[q]), k_color * d2));
            const float w = b3[dy + 2] * b3[dx + 2] * __builtin_amdgcn_exp2f(e);
            ax = fmaf(w, cqx, ax); ay = fmaf(w, cqy, ay); az = fmaf(w, cqz, az);
            wsum += w;
        }
    }
    const float inv = 1.0f / wsum;
    float4 r = make_float4(ax * inv, ay * inv, az * inv, spp);
    if (LAST) {
        const float4 a = g1[at];
        r.x *= fmaxf(a.x, albedo_floor) * spp; r.y *= fmaxf(a.y, albedo_floor) * spp; r.z *= fmaxf(a.z, albedo_floor) * spp;
    }
    cout[at] = r;
}

// Joint bilateral upsampling (Kopf, Cohen, Lischinski, Uyttendaele 2007) of a low-resolution frame onto the
// pixel grid of guides rendered at the high resolution: a high-resolution pixel takes the 3x3 low-resolution
// pixels around its own position (tent weights of width 1.5 low pixels), each counted by how well the first hit
// of ITS centre ray agrees with the first hit of the high-resolution pixel's (same object, normal, distance), on
// colours divided by the low-resolution albedo; the result is multiplied by the HIGH-resolution albedo, so
// texture detail and silhouettes come from the high-resolution guides.  A pixel none of whose neighbours
// agrees (a feature thinner than a low-resolution pixel) falls back to the tent weights alone.
// Pixel centres map as main.cc:117-118 places them: s = (x + 0.5) / (W - 1) on either grid.
__global__ __launch_bounds__(kThreads) void upscale_kernel(const float4* __restrict__ clow, const float4* __restrict__ g0l,
                                                           const float4* __restrict__ g1l, int lw, int lh,
                                                           const float4* __restrict__ g0h, const float4* __restrict__ g1h, int hw, int hh,
                                                           float sigma_normal, float inv_sigma_depth, float albedo_floor,
                                                           float4* __restrict__ out) {
    const int x = blockIdx.x * kTileW + (threadIdx.x % kTileW), y = blockIdx.y * kTileH + (threadIdx.x / kTileW);
    if (x >= hw || y >= hh) return;
    const long long at = (long long)y * hw + x;
    const float4 gp = g0h[at], ap = g1h[at];
    const int idp = __float_as_int(ap.w);
    const bool hit = idp >= 0;
    const float ux = ((float)x + 0.5f) * ((float)(lw - 1) / (float)(hw - 1)) - 0.5f;
    const float uy = ((float)y + 0.5f) * ((float)(lh - 1) / (float)(hh - 1)) - 0.5f;
    const int bx = (int)floorf(ux + 0.5f), by = (int)floorf(uy + 0.5f);      // nearest low-resolution pixel
    const float k_depth = hit ? -1.4426950408889634f * inv_sigma_depth / fmaxf(gp.w, 1.0e-6f) : 0.0f;
    const float k_normal = hit ? sigma_normal : 0.0f;
    float ax = 0.0f, ay = 0.0f, az = 0.0f, wsum = 0.0f, bx_ = 0.0f, by_ = 0.0f, bz_ = 0.0f, bsum = 0.0f, spp = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int qx = min(max(bx + dx, 0), lw - 1), qy = min(max(by + dy, 0), lh - 1);
            const float tent = fmaxf(1.0f - fabsf((float)(bx + dx) - ux) * (1.0f / 1.5f), 0.0f) *
                               fmaxf(1.0f - fabsf((float)(by + dy) - uy) * (1.0f / 1.5f), 0.0f);
            const long long q = (long long)qy * lw + qx;
            const float4 cq = clow[q], gq = g0l[q], aq = g1l[q];
            const float inv = 1.0f / fmaxf(cq.w, 1.0f);
            const float rx = cq.x * inv / fmaxf(aq.x, albedo_floor), ry = cq.y * inv / fmaxf(aq.y, albedo_floor),
                        rz = cq.z * inv / fmaxf(aq.z, albedo_floor);
            if (dx == 0 && dy == 0) spp = cq.w;
            bx_ = fmaf(tent, rx, bx_); by_ = fmaf(tent, ry, by_); bz_ = fmaf(tent, rz, bz_); bsum += tent;
            if (__float_as_int(aq.w) != idp) continue;
            const float cosn = fminf(fmaxf(fmaf(gp.z, gq.z, fmaf(gp.y, gq.y, gp.x * gq.x)), 1.0e-6f), 1.0f);
            const float w = tent * __builtin_amdgcn_exp2f(fmaf(k_normal, __builtin_amdgcn_logf(cosn), k_depth * fabsf(gp.w - gq.w)));
            ax = fmaf(w, rx, ax); ay = fmaf(w, ry, ay); az = fmaf(w, rz, az); wsum += w;
        }
    }
    const bool agreed = wsum > 1.0e-4f;
    const float inv = 1.0f / (agreed ? wsum : bsum);         // (the centre tap's tent weight is at least 4/9)
    const float sx = (agreed ? ax : bx_) * inv, sy = (agreed ? ay : by_) * inv, sz = (agreed ? az : bz_) * inv;
    out[at] = make_float4(sx * fmaxf(ap.x, albedo_floor) * spp, sy * fmaxf(ap.y, albedo_floor) * spp, sz * fmaxf(ap.z, albedo_floor) * spp, spp);
}

}  // namespace

hipError_t launch_upscale(const DReconstruct& rp, int low_width, int low_height, const float4* rgba_low, const float4* guides_low,
                          const float4* guides_high, float4* out, hipStream_t stream) {
    const long long nl = (long long)low_width * low_height, nh = (long long)rp.width * rp.height;
    if (nl <= 0 || nh <= 0) return hipSuccess;
    const dim3 grid((unsigned)((rp.width + kTileW - 1) / kTileW), (unsigned)((rp.height + kTileH - 1) / kTileH)), block(kThreads);
    hipLaunchKernelGGL(upscale_kernel, grid, block, 0, stream, rgba_low, guides_low, guides_low + nl, low_width, low_height,
                       guides_high, guides_high + nh, rp.width, rp.height, rp.sigma_normal, 1.0f / rp.sigma_depth, rp.albedo_floor, out);
    return hipGetLastError();
}

// one pass staged in LDS, with the tile shape of its step
template <bool FIRST, bool LAST, int TW, int TH>
static hipError_t launch_lds_pass(const DReconstruct& rp, int step, float scale, const float4* src, const float4* g0, const float4* g1,
                                  float4* dst, hipStream_t stream) {
    const int tiles_x = (rp.width + TW - 1) / TW, tiles_y = (rp.height + TH - 1) / TH;
    const int band_rows = (tiles_y + 7) / 8;
    const size_t lds = (size_t)(TW + 4 * step) * (TH + 4 * step) * 8 * sizeof(float);
    auto kernel = atrous_lds_kernel<FIRST, LAST, TW, TH>;
    if (lds > 64 * 1024) {                         // beyond the default limit of a workgroup: ask for it (160 KB per CU)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)(8 * band_rows * tiles_x)), dim3(TW * TH), lds, stream, src, g0, g1, rp.width, rp.height, step,
                       rp.sigma_color, scale, rp.sigma_normal, 1.0f / rp.sigma_depth, rp.albedo_floor, band_rows, tiles_x, dst);
    return hipGetLastError();
}

hipError_t launch_reconstruct(const DReconstruct& rp, const float4* rgba, const float4* guides, float4* tmp, float4* out,
                              hipStream_t stream) {
    const long long n = (long long)rp.width * rp.height;
    if (n <= 0) return hipSuccess;
    const float4* g0 = guides;
    const float4* g1 = guides + n;
    float4* a = tmp;
    float4* b = tmp + n;
    // the first pass (step 1) stages its tile in LDS and demodulates while it does; without LDS the demodulated frame
    // is written out first
    // (a single pass in place would read pixels other workgroups have already overwritten: that one goes through the
    // scratch image like the unfused path)
    const bool fused = !rp.no_lds && !(rp.iterations == 1 && out == rgba);
    if (!fused)
        hipLaunchKernelGGL(demodulate_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, rgba, g1, n,
                           rp.albedo_floor, a);
    const int tiles_x = (rp.width + kTileW - 1) / kTileW, tiles_y = (rp.height + kTileH - 1) / kTileH;
    const int band_rows = (tiles_y + 7) / 8;
    const dim3 grid((unsigned)(8 * band_rows * tiles_x)), block(kThreads);
    float scale = 1.0f;
    for (int it = 0; it < rp.iterations; it++, scale *= 4.0f) {
        const bool last = it == rp.iterations - 1;
        const float4* src = (it == 0 && fused) ? rgba : a;
        float4* dst = last ? out : b;
        const int step = 1 << it;
        hipError_t e = hipSuccess;
        if (it == 0 && fused) {
            e = last ? launch_lds_pass<true, true, kTileW, kTileH>(rp, step, scale, src, g0, g1, dst, stream)
                     : launch_lds_pass<true, false, kTileW, kTileH>(rp, step, scale, src, g0, g1, dst, stream);
        } else if (step <= 4 && !rp.no_lds) {
            e = last ? launch_lds_pass<false, true, kTileW, kTileH>(rp, step, scale, src, g0, g1, dst, stream)
                     : launch_lds_pass<false, false, kTileW, kTileH>(rp, step, scale, src, g0, g1, dst, stream);
        } else if (step == 8 && !rp.no_lds) {
            e = last ? launch_lds_pass<false, true, 64, 16>(rp, step, scale, src, g0, g1, dst, stream)
                     : launch_lds_pass<false, false, 64, 16>(rp, step, scale, src, g0, g1, dst, stream);
        } else if (last) {
            hipLaunchKernelGGL(atrous_kernel<true>, grid, block, 0, stream, src, g0, g1, rp.width, rp.height, step, rp.sigma_color, scale,
                               rp.sigma_normal, 1.0f / rp.sigma_depth, rp.albedo_floor, band_rows, tiles_x, dst);
        } else {
            hipLaunchKernelGGL(atrous_kernel<false>, grid, block, 0, stream, src, g0, g1, rp.width, rp.height, step, rp.sigma_color, scale,
                               rp.sigma_normal, 1.0f / rp.sigma_depth, rp.albedo_floor, band_rows, tiles_x, dst);
        }
        if (e != hipSuccess) return e;
        // the pass read `src` and wrote `dst`: the next one reads what was written, and writes the other scratch image
        if (!last) { float4* t = a; a = b; b = t; }
    }
    return hipGetLastError();
}

}  // namespace rt
