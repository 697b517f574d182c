// C-ABI of the render path (include/rt.h): context, scene flattening into HBM, launches.
#include "../../include/rt.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_lbvh.h"

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;

    // scene
    bool has_scene = false, has_camera = false;
    rt::DScene scene{};
    rt::DCamera camera{};
    std::vector<void*> scene_allocs;
    int accel_used = RT_ACCEL_LIST;
    int bvh_nodes = 0, bvh_depth = 0;
    float build_ms = 0.0f;
    double max_radiance = 1.0;     // brightest thing a sample can see: emitters and the background (see validate_render)
    int coarse_nodes = 0;          // nodes the 16-bit grid is too coarse for (any: the tree is walked in its float form)
    int* stack_spill = nullptr;    // deep trees walked out of HBM: a strip per lane for the stack entries beyond the LDS stack
    int stack_spill_entries = 0;   // entries per strip

    // per-render state
    unsigned long long* counters = nullptr;   // device, rt::kNumCounters
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    int last_k = 0;
    float4* staging = nullptr;                // device image for rt_render_host
    size_t staging_bytes = 0;
    unsigned char* trace_buf = nullptr;       // device copies of rays and hits for rt_trace_primary (grown on demand)
    size_t trace_rays = 0;                    // rays it holds
    float4* recon_tmp = nullptr;              // two images of scratch for rt_reconstruct (grown on demand)
    size_t recon_pixels = 0;
    unsigned char* noalloc = nullptr;         // this device's address of the fixed staging array of rt_trace_primary_noalloc
};

namespace {

// ---- rt_trace_primary_noalloc: device staging that is never allocated -------------------------------
// The device copies of the rays and of the hits of rt_trace_primary_noalloc live in this array, which
// is part of the library's code object: every GPU the library runs on holds its own copy at its own
// address from the moment the code object is loaded until the process ends.  In host code the name
// `scratch` is only a handle for hipGetSymbolAddress; the address a kernel or a copy may use is the
// one that call returns, looked up once per device (rt_create) and kept in g_scratch_ptr.
constexpr size_t kNoAllocRayBytes = (size_t)RT_TRACE_NOALLOC_MAX_RAYS * 6 * sizeof(float);
constexpr size_t kNoAllocHitBytes = (size_t)RT_TRACE_NOALLOC_MAX_RAYS * sizeof(rt::Hit);
static __device__ __align__(256) unsigned char scratch[kNoAllocRayBytes + kNoAllocHitBytes];
static_assert(kNoAllocRayBytes % 256 == 0, "the hits start on a 256-byte boundary");

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;                       // guards g_scratch_ptr
unsigned char* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_in_use[kMaxDevices];         // one call at a time per GPU stages through that GPU's array

// Device address of `scratch` on `device`, which must be the current device.
hipError_t resolve_scratch(int device, unsigned char** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

int fail(rt_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    return code;
}

#define RT_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void free_scene(rt_context* ctx) {
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->has_scene = false;
    ctx->scene = rt::DScene{};
}

template <typename T>
int upload(rt_context* ctx, const std::vector<T>& host, const T** dev) {
    *dev = nullptr;
    if (host.empty()) return RT_OK;
    void* p = nullptr;
    RT_HIP(ctx, hipMalloc(&p, host.size() * sizeof(T)));
    ctx->scene_allocs.push_back(p);
    RT_HIP(ctx, hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<const T*>(p);
    return RT_OK;
}

bool finite3(const double* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// camera.h:19-38 in double; the result is rounded to float once at the end.
void build_camera(const rt_camera_desc& c, rt::DCamera* out) {
    const double pi = 3.1415926535897932385;
    double theta = c.vfov * pi / 180.0;
    double h = std::tan(theta / 2);
    double vh = 2.0 * h, vw = c.aspect * vh;
    double w[3], u[3], v[3], t[3];
    for (int a = 0; a < 3; a++) t[a] = c.lookfrom[a] - c.lookat[a];
    double len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int a = 0; a < 3; a++) w[a] = (1 / len) * t[a];
    double cr[3] = { c.vup[1] * w[2] - c.vup[2] * w[1], c.vup[2] * w[0] - c.vup[0] * w[2],
                     c.vup[0] * w[1] - c.vup[1] * w[0] };
    len = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    for (int a = 0; a < 3; a++) u[a] = (1 / len) * cr[a];
    v[0] = w[1] * u[2] - w[2] * u[1]; v[1] = w[2] * u[0] - w[0] * u[2]; v[2] = w[0] * u[1] - w[1] * u[0];
    for (int a = 0; a < 3; a++) {
        double hor = c.focus_dist * vw * u[a], ver = c.focus_dist * vh * v[a];
        double llc = c.lookfrom[a] - hor / 2 - ver / 2 - c.focus_dist * w[a];
        out->origin[a] = (float)c.lookfrom[a];
        out->llc_rel[a] = (float)(llc - c.lookfrom[a]);
        out->horizontal[a] = (float)hor;
        out->vertical[a] = (float)ver;
        out->u[a] = (float)u[a];
        out->v[a] = (float)v[a];
    }
    out->lens_radius = (float)(c.aperture / 2);
}

int shard_rows(int height, int tile_rows, int tile_first, int tile_step) {
    if (tile_rows <= 0 || tile_step <= 0) return height;
    int n_tiles = (height + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int t = tile_first; t < n_tiles; t += tile_step) {
        int r0 = t * tile_rows, r1 = r0 + tile_rows;
        rows += (r1 > height ? height : r1) - r0;
    }
    return rows;
}

int choose_lanes_per_pixel(long long pixels, int spp) {
    // A wave's tile holds 64/K pixels.  K = 8 (a 4x2 tile, eight samples of a pixel in flight at
    // once) measured best on whole 1080p frames: the lanes' primary rays are more coherent than
    // with a whole 8x8 tile, and a wave keeps two tiles' accumulators in LDS (the one it hands out
    // and the one still finishing), which at K = 4 costs the tree-in-LDS kernels a workgroup per CU.
    // The persistent grid has 5120 waves and a tile is the unit they draw; with too few tiles per
    // wave the last tiles of a launch leave most of the chip idle.  So K grows with the number of
    // tiles the launch (a rank's shard, not the frame) would have; each doubling costs some
    // coherence, so it asks for less.  The image does not depend on K (integer accumulation).
    int k = spp >= 8 ? 8 : (spp >= 4 ? 4 : (spp >= 2 ? 2 : 1));
    if (k == 8 && 2 * k <= spp && pixels * 8 / 64 < 50000) k = 16;
    while (k < 64 && 2 * k <= spp && pixels * k / 64 < 5120) k *= 2;      // tiny frames: at least a tile per wave
    return k;
}

int validate_render(rt_context* ctx, const rt_render_params* p, rt::DRender* rp) {
    if (!ctx) return RT_ERR_INVALID;
    // three layouts are accepted: the whole struct, the one that ends before stack_lds_entries, and the
    // one that ends before the traversal tuning fields
    const bool has_stack = p && p->struct_size == sizeof(rt_render_params);
    const bool has_tuning = has_stack || (p && p->struct_size == offsetof(rt_render_params, stack_lds_entries));
    if (!p || (!has_tuning && p->struct_size != offsetof(rt_render_params, traversal_slice)))
        return fail(ctx, RT_ERR_INVALID, "rt_render_params: null or struct_size mismatch");
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "no scene uploaded");
    if (!ctx->has_camera) return fail(ctx, RT_ERR_NO_CAMERA, "no camera set");
    if (p->width < 2 || p->height < 2 || p->width > 65536 || p->height > 65536)
        return fail(ctx, RT_ERR_INVALID, "width/height must be in [2, 65536]");
    if ((long long)p->width * p->height > 0x7fffffffLL)
        return fail(ctx, RT_ERR_UNSUPPORTED, "more than 2^31 - 1 pixels (the random stream is keyed by a 32-bit pixel index)");
    if (p->spp < 1 || p->sample_begin < 0 || p->max_depth < 0)
        return fail(ctx, RT_ERR_INVALID, "spp >= 1, sample_begin >= 0, max_depth >= 0 required");
    if ((long long)p->sample_begin + p->spp > 0x7fffffffLL)
        return fail(ctx, RT_ERR_INVALID, "sample range overflows");
    // a pixel's sum of sample radiance must stay below 2^32 (32.32 fixed point): the samples up to the end
    // of this call's window, each at most the scene's brightest emitter
    if (((double)p->sample_begin + (double)p->spp) * ctx->max_radiance >= 4294967296.0)
        return fail(ctx, RT_ERR_UNSUPPORTED, "a pixel's radiance sum could reach 2^32: " + std::to_string((long long)p->sample_begin + p->spp) +
                                             " samples of an emitter (or background) of " + std::to_string(ctx->max_radiance) +
                                             "; render fewer samples per call and combine the means in floating point");
    bool whole = p->tile_rows == 0 && p->tile_first == 0 && p->tile_step == 0;
    if (!whole && (p->tile_rows < 1 || p->tile_step < 1 || p->tile_first < 0 || p->tile_first >= p->tile_step))
        return fail(ctx, RT_ERR_INVALID, "tiles: tile_rows >= 1, 0 <= tile_first < tile_step required");
    int k = p->lanes_per_pixel;
    if (k < 0 || k > 64 || (k & (k - 1)) != 0)
        return fail(ctx, RT_ERR_INVALID, "lanes_per_pixel must be 0 or a power of two <= 64");
    if (!(p->t_min >= 0.0f) || !std::isfinite(p->t_min)) return fail(ctx, RT_ERR_INVALID, "t_min must be >= 0");
    rp->width = p->width; rp->height = p->height;
    rp->spp = p->spp; rp->sample_begin = p->sample_begin; rp->max_depth = p->max_depth;
    rp->seed_lo = (uint32_t)p->seed; rp->seed_hi = (uint32_t)(p->seed >> 32);
    rp->tile_rows = whole ? p->height : p->tile_rows;
    rp->tile_first = whole ? 0 : p->tile_first;
    rp->tile_step = whole ? 1 : p->tile_step;
    rp->rows = shard_rows(p->height, rp->tile_rows, rp->tile_first, rp->tile_step);
    if (k == 0) k = choose_lanes_per_pixel((long long)p->width * rp->rows, p->spp);
    rp->t_min = p->t_min == 0.0f ? 0.001f : p->t_min;
    rp->flags = p->flags;
    while (k < 64 && (long long)p->spp * (64 / k) > 0x7fffffffLL) k *= 2;   // a tile's (pixel, sample) units are counted in 32 bits
    rp->lanes_per_pixel = k;
    rp->inv_wm1 = (float)(1.0 / (double)(p->width - 1));
    rp->inv_hm1 = (float)(1.0 / (double)(p->height - 1));
    const int slice = has_tuning ? p->traversal_slice : 0, quorum = has_tuning ? p->leaf_quorum : 0;
    // (the kernel multiplies the quorum by a lane count: 64 lanes make anything above 64 mean the same)
    if (slice < -1 || slice > 65534 || quorum < -1 || quorum > 64)
        return fail(ctx, RT_ERR_INVALID, "traversal_slice must be in [-1, 65534] and leaf_quorum in [-1, 64]");
    const int share = has_stack ? p->traversal_exit_share : 0;
    if (share < -1 || share > 128) return fail(ctx, RT_ERR_INVALID, "traversal_exit_share must be -1, 0 or in [1, 128]");
    // device side: steps in the low 16 bits (0xffff = chosen at launch from the tree depth, 0 = unsliced), exit share above them
    rp->slice = (slice == 0 ? 0xffff : (slice < 0 ? 0 : slice)) | ((share == 0 ? 255 : (share < 0 ? 0 : share)) << 16);
    rp->leaf_quorum = quorum == 0 ? -1 : (quorum < 0 ? 0 : quorum);     // device side: -1 = default ratio, 0 = wait for all
    const int lds_entries = has_stack ? p->stack_lds_entries : 0;
    if (lds_entries != 0 && (lds_entries < 2 || lds_entries > rt::kMaxStack))
        return fail(ctx, RT_ERR_INVALID, "stack_lds_entries must be 0 or in [2, 48]");
    rp->stack_cap = lds_entries == 0 ? rt::kLdsStackCap : lds_entries;
    if ((p->flags & RT_FLAG_FIXED_SUMS) && (p->flags & RT_FLAG_MEAN))
        return fail(ctx, RT_ERR_INVALID, "RT_FLAG_FIXED_SUMS excludes RT_FLAG_MEAN (take the mean in rt_resolve_sums)");
    return RT_OK;
}

int do_render(rt_context* ctx, const rt::DRender& rp, float4* out_dev) {   // with RT_FLAG_FIXED_SUMS out_dev holds 32 bytes per pixel
    // a tree in HBM deeper than the LDS stack of its walk: a strip per lane for the rest of the stack
    // (rt_device.h, kLdsStackCap), grown on demand and kept by the context
    const int spill_need = (ctx->scene.use_bvh && ctx->scene.qnodes) ? ctx->scene.bvh_depth + 2 - rp.stack_cap : 0;
    if (spill_need > ctx->stack_spill_entries) {
        if (ctx->stack_spill) (void)hipFree(ctx->stack_spill);     // (waits for the launches that use it)
        ctx->stack_spill = nullptr; ctx->stack_spill_entries = 0;
        RT_HIP(ctx, hipMalloc((void**)&ctx->stack_spill, (size_t)rt::kSpillLanes * spill_need * sizeof(int)));
        ctx->stack_spill_entries = spill_need;
    }
    ctx->scene.stack_spill = ctx->stack_spill;
    ctx->scene.stack_spill_entries = ctx->stack_spill_entries;
    RT_HIP(ctx, hipMemsetAsync(ctx->counters, 0, rt::kNumCounterSlots * sizeof(unsigned long long), ctx->stream));
    RT_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    RT_HIP(ctx, rt::launch_render(ctx->scene, ctx->camera, rp, out_dev, ctx->counters, ctx->stream));
    RT_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed = true;
    ctx->last_k = rp.lanes_per_pixel;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_version(void) { return "raytrace_amd 0.3 (gfx950)"; }

int rt_create(int device, rt_context** out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return RT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RT_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RT_ERR_NO_DEVICE;   // kernels are gfx950 only
    rt_context* ctx = new (std::nothrow) rt_context;
    if (!ctx) return RT_ERR_INVALID;
    ctx->device = device;
    DeviceGuard g(device);
    if (hipMalloc((void**)&ctx->counters, rt::kNumCounterSlots * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess ||
        resolve_scratch(device, &ctx->noalloc) != hipSuccess) {
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_context* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->staging) (void)hipFree(ctx->staging);
    if (ctx->trace_buf) (void)hipFree(ctx->trace_buf);
    if (ctx->stack_spill) (void)hipFree(ctx->stack_spill);
    if (ctx->recon_tmp) (void)hipFree(ctx->recon_tmp);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    delete ctx;
}

int rt_set_stream(rt_context* ctx, void* hip_stream) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return RT_OK;
}

const char* rt_last_error(rt_context* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int rt_shard_rows(int32_t height, int32_t tile_rows, int32_t tile_first, int32_t tile_step) {
    if (height < 0) return RT_ERR_INVALID;
    bool whole = tile_rows == 0 && tile_first == 0 && tile_step == 0;
    if (!whole && (tile_rows < 1 || tile_step < 1 || tile_first < 0 || tile_first >= tile_step)) return RT_ERR_INVALID;
    return shard_rows(height, tile_rows, tile_first, tile_step);
}

int rt_upload_scene(rt_context* ctx, const rt_scene_desc* d) {
    if (!ctx) return RT_ERR_INVALID;
    // two layouts are accepted: the whole struct, or the one that ends before the instance transforms
    const bool has_xf_fields = d && d->struct_size == sizeof(rt_scene_desc);
    if (!d || (!has_xf_fields && d->struct_size != offsetof(rt_scene_desc, n_xforms)))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_desc: null or struct_size mismatch");
    const int n_xf = has_xf_fields ? d->n_xforms : 0;
    const int32_t* prim_xf = has_xf_fields && n_xf > 0 ? d->prim_xform : nullptr;
    if (n_xf < 0 || (n_xf > 0 && !d->xform)) return fail(ctx, RT_ERR_INVALID, "instance transform array missing");
    if (n_xf > (1 << 22)) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 2^22 instance transforms");
    if (d->n_prims < 0 || d->n_mats < 0 || d->n_tex < 0) return fail(ctx, RT_ERR_INVALID, "negative count");
    if (d->n_prims > 0 && (!d->prim_type || !d->prim_geom || !d->prim_mat))
        return fail(ctx, RT_ERR_INVALID, "primitive arrays missing");
    if (d->n_mats > 0 && (!d->mat_type || !d->mat_param || !d->mat_tex))
        return fail(ctx, RT_ERR_INVALID, "material arrays missing");
    if (d->n_tex > 0 && (!d->tex_type || !d->tex_color || !d->tex_a || !d->tex_b || !d->tex_off))
        return fail(ctx, RT_ERR_INVALID, "texture arrays missing");
    if (d->accel < RT_ACCEL_AUTO || d->accel > RT_ACCEL_BVH_DEVICE) return fail(ctx, RT_ERR_INVALID, "unknown accel");
    if (d->background != RT_BG_SKY && d->background != RT_BG_SOLID) return fail(ctx, RT_ERR_INVALID, "unknown background");
    // leaf codes hold first << kLeafBits | count - 1 in 31 bits, Origin codes an index in 28
    static_assert((1LL << 28) <= (0x7fffffffLL >> rt::kLeafBits) + 1, "a leaf code holds any sphere index");
    if (d->n_prims > (1 << 28)) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 2^28 primitives");
    const double wide_radius = d->wide_radius > 0 ? d->wide_radius : 64.0;

    // textures
    std::vector<rt::DTexture> texs((size_t)d->n_tex);
    for (int i = 0; i < d->n_tex; i++) {
        rt::DTexture& t = texs[i];
        t = rt::DTexture{};
        t.type = d->tex_type[i];
        if (t.type == RT_TEX_SOLID) {
            t.r = (float)d->tex_color[3 * i]; t.g = (float)d->tex_color[3 * i + 1]; t.bl = (float)d->tex_color[3 * i + 2];
        } else if (t.type == RT_TEX_CHECKER) {
            t.a = d->tex_a[i]; t.b = d->tex_b[i];
            if (t.a < 0 || t.a >= d->n_tex || t.b < 0 || t.b >= d->n_tex)
                return fail(ctx, RT_ERR_INVALID, "checker texture refers outside the texture array");
        } else if (t.type == RT_TEX_IMAGE) {
            t.a = d->tex_a[i]; t.b = d->tex_b[i];
            int64_t off = d->tex_off[i];
            if (t.a < 1 || t.b < 1 || t.a > 32768 || t.b > 32768 || off < 0 || !d->texels ||
                (uint64_t)off + (uint64_t)t.a * t.b * 3 > d->texel_bytes || (uint64_t)off > 0xffffffffULL)
                return fail(ctx, RT_ERR_INVALID, "image texture outside the texel blob");
            t.off = (uint32_t)off;
        } else {
            return fail(ctx, RT_ERR_INVALID, "unknown texture type");
        }
    }
    // checker nesting must end in a leaf texture within the kernel's 16-level walk
    for (int i = 0; i < d->n_tex; i++) {
        std::vector<int> todo{i};
        for (int level = 0; !todo.empty(); level++) {
            if (level > 8) return fail(ctx, RT_ERR_UNSUPPORTED, "checker textures nested deeper than 8 (or cyclic)");
            std::vector<int> next;
            for (int t : todo)
                if (texs[t].type == RT_TEX_CHECKER) { next.push_back(texs[t].a); next.push_back(texs[t].b); }
            todo.swap(next);
        }
    }

    // materials
    std::vector<rt::DMaterial> mats((size_t)d->n_mats);
    for (int i = 0; i < d->n_mats; i++) {
        rt::DMaterial& m = mats[i];
        m = rt::DMaterial{};
        m.type = d->mat_type[i];
        if (m.type < RT_LAMBERTIAN || m.type > RT_LAMBERTIAN_BG) return fail(ctx, RT_ERR_INVALID, "unknown material type");
        const double* p = d->mat_param + 5 * (size_t)i;
        m.r = (float)p[0]; m.g = (float)p[1]; m.b = (float)p[2];
        m.fuzz = (float)(p[3] < 1 ? p[3] : 1.0);           // material.h:40
        m.ir = (float)p[4];
        m.tex = d->mat_tex[i];
        if (m.tex >= d->n_tex) return fail(ctx, RT_ERR_INVALID, "material refers outside the texture array");
        if (m.tex < 0) m.tex = -1;
        if ((m.type == RT_METAL || m.type == RT_DIELECTRIC) && m.tex >= 0) m.tex = -1;   // neither reads a texture
        if (m.type == RT_DIELECTRIC && !(m.ir > 0.0f)) return fail(ctx, RT_ERR_INVALID, "dielectric needs ir > 0");
        if (m.type == RT_DIELECTRIC) {
            // material.h:61,88-90 per material instead of per hit: the refraction ratio seen from
            // outside and Schlick's r0 for either side, in the same single-precision operations
            // the kernel would use (a dielectric has no colour, so they take the colour's place)
            const float inv_ir = 1.0f / m.ir;
            const float r0_front = (1.0f - inv_ir) / (1.0f + inv_ir), r0_back = (1.0f - m.ir) / (1.0f + m.ir);
            m.r = inv_ir; m.g = r0_front * r0_front; m.b = r0_back * r0_back;
        }
    }

    // The brightest radiance a sample can return: a path's throughput is a product of albedos (at most 1
    // for any physical material), so a sample is at most the brightest emitter or the background.  A
    // pixel's sum is kept in unsigned 32.32 fixed point: rt_render refuses a sample count that could
    // carry it past 2^32 instead of letting it wrap (main.cc:121 sums in double, which does not).
    double max_radiance = d->background == RT_BG_SKY ? 1.0 : std::fmax(d->bg_color[0], std::fmax(d->bg_color[1], d->bg_color[2]));
    {
        std::vector<double> tex_max((size_t)d->n_tex, 1.0);       // image textures: bytes / 255
        for (int pass = 0; pass < 9; pass++)                      // checkers nest at most 8 deep (checked above)
            for (int i = 0; i < d->n_tex; i++) {
                if (texs[i].type == RT_TEX_SOLID) tex_max[i] = std::fmax(texs[i].r, std::fmax(texs[i].g, texs[i].bl));
                else if (texs[i].type == RT_TEX_CHECKER) tex_max[i] = std::fmax(tex_max[texs[i].a], tex_max[texs[i].b]);
            }
        for (const rt::DMaterial& m : mats)
            if (m.type == RT_DIFFUSE_LIGHT || m.type == RT_LAMBERTIAN_BG)
                max_radiance = std::fmax(max_radiance, m.tex >= 0 ? tex_max[m.tex] : std::fmax(m.r, std::fmax(m.g, m.b)));
        if (!(max_radiance >= 0.0)) max_radiance = 0.0;           // (NaN colours render as 0)
    }

    // instance transforms: sine and cosine in double (cuda/sceneobject.h:50-52 takes them in float)
    std::vector<rt::DXform> xforms((size_t)n_xf);
    std::vector<double> xf_sin((size_t)n_xf), xf_cos((size_t)n_xf);
    for (int i = 0; i < n_xf; i++) {
        const double* x = d->xform + 4 * (size_t)i;
        if (!std::isfinite(x[0]) || !finite3(x + 1)) return fail(ctx, RT_ERR_INVALID, "instance transform must be finite");
        double rad = x[0] * 3.1415926535897932385 / 180.0;
        xf_sin[i] = std::sin(rad); xf_cos[i] = std::cos(rad);
        rt::DXform& o = xforms[i];
        o = rt::DXform{};
        o.s = xf_sin[i]; o.c = xf_cos[i];
        o.tx = x[1]; o.ty = x[2]; o.tz = x[3];
    }

    // primitives: ordinary spheres (BVH / scan), wide spheres and rects (always tested)
    std::vector<double> sc_c, sc_r;          // ordinary spheres, input order
    std::vector<int32_t> sc_obj, sc_mat, sc_xf;
    bool any_sphere_xf = false;
    std::vector<rt::DWide> wide;
    std::vector<rt::DRect> rects;
    std::vector<double> rect_k;
    for (int i = 0; i < d->n_prims; i++) {
        const double* g = d->prim_geom + 6 * (size_t)i;
        int mat = d->prim_mat[i];
        if (mat < 0 || mat >= d->n_mats) return fail(ctx, RT_ERR_INVALID, "primitive refers outside the material array");
        int type = d->prim_type[i];
        int xf = prim_xf ? prim_xf[i] : -1;
        if (xf >= n_xf) return fail(ctx, RT_ERR_INVALID, "primitive refers outside the instance transform array");
        if (xf < 0) xf = -1;
        if (type == RT_SPHERE) {
            if (!finite3(g) || !std::isfinite(g[3]) || g[3] == 0.0) return fail(ctx, RT_ERR_INVALID, "sphere needs finite centre and non-zero radius");
            // a turned and shifted sphere is a sphere about the moved centre; the transform is kept
            // only to turn its texture coordinates back
            double c[3] = { g[0], g[1], g[2] };
            if (xf >= 0) {
                const double* x = d->xform + 4 * (size_t)xf;
                c[0] = xf_cos[xf] * g[0] + xf_sin[xf] * g[2] + x[1];
                c[1] = g[1] + x[2];
                c[2] = -xf_sin[xf] * g[0] + xf_cos[xf] * g[2] + x[3];
            }
            if (std::fabs(g[3]) > wide_radius) {
                rt::DWide w{};
                w.cx = c[0]; w.cy = c[1]; w.cz = c[2]; w.r = g[3]; w.r2 = g[3] * g[3]; w.obj = i; w.mat = mat; w.xform = xf + 1;
                wide.push_back(w);
            } else {
                sc_c.insert(sc_c.end(), c, c + 3); sc_r.push_back(g[3]);
                sc_obj.push_back(i); sc_mat.push_back(mat); sc_xf.push_back(xf + 1);
                any_sphere_xf = any_sphere_xf || xf >= 0;
            }
        } else if (type == RT_XY_RECT || type == RT_XZ_RECT || type == RT_YZ_RECT) {
            for (int a = 0; a < 5; a++) if (!std::isfinite(g[a])) return fail(ctx, RT_ERR_INVALID, "rect needs finite bounds");
            rt::DRect r{};
            r.a0 = (float)g[0]; r.a1 = (float)g[1]; r.b0 = (float)g[2]; r.b1 = (float)g[3]; r.k = (float)g[4];
            int ka = type == RT_XY_RECT ? 2 : (type == RT_XZ_RECT ? 1 : 0);
            int aa = type == RT_YZ_RECT ? 1 : 0;
            int ba = type == RT_XY_RECT ? 1 : 2;
            const int k_inexact = (double)r.k != g[4] ? 1 : 0;     // the plane coordinate does not fit a float
            r.axes = ka | (aa << 2) | (ba << 4) | (k_inexact << 6) | ((xf + 1) << 8);
            r.obj = i; r.mat = mat;
            rects.push_back(r);
            rect_k.push_back(g[4]);
        } else {
            return fail(ctx, RT_ERR_INVALID, "unknown primitive type");
        }
    }
    const int64_t ns = (int64_t)sc_r.size();

    bool use_bvh = d->accel == RT_ACCEL_BVH || d->accel == RT_ACCEL_BVH_DEVICE || (d->accel == RT_ACCEL_AUTO && ns > 32);
    // the GPU builder needs an inner root; a handful of spheres goes through the host builder
    // RT_ACCEL_AUTO: from 65536 spheres on the host build (0.5 s per million) costs more than the
    // few percent of traversal time its tree saves on a frame
    bool device_build = (d->accel == RT_ACCEL_BVH_DEVICE && ns > 4) || (d->accel == RT_ACCEL_AUTO && ns >= 65536);
    const auto t_build = std::chrono::steady_clock::now();
    DeviceGuard g(ctx->device);
    rt::DScene s{};
    int n_nodes = 0, root = rt::kNoBvh, depth = 0;
    for (;;) {
        rt::BvhResult bvh;
        std::vector<int32_t> order((size_t)ns);
        for (int64_t i = 0; i < ns; i++) order[i] = (int32_t)i;
        if (use_bvh && !device_build) {
            // Large scenes (walked out of HBM): leaves where the surface-area heuristic prefers them to
            // another split, which is almost nowhere for spheres that do not overlap: 1.7 spheres
            // per leaf instead of 3.  10^6 spheres: 7 sphere tests per ray instead of 22 for 7 % more
            // box tests, frame 343 -> 336 ms; 2 x 10^4 spheres: no gain, so the rule starts above that.
            const bool fine = ns > 65536;
            bvh = rt::build_bvh(sc_c.data(), sc_r.data(), ns, rt::kMaxStack - 4, fine ? 8 : 4, fine ? 1.5f : 0.0f);
            if (bvh.depth > rt::kMaxStack - 2) return fail(ctx, RT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
            order = bvh.order;
        }
        // sphere arrays in leaf order (input order when the GPU builder will permute them)
        // a multiple of four (the linear scan's group size) plus four (a leaf is read as four records)
        const int64_t ns_pad = (ns + 3) / 4 * 4 + 4;
        std::vector<float4> sph((size_t)ns_pad, make_float4(1e30f, 1e30f, 1e30f, 0.0f));
        std::vector<double4> sph64((size_t)ns_pad, make_double4(1e30, 1e30, 1e30, 1.0));
        std::vector<int2> info((size_t)ns_pad, make_int2(-1, 0));
        std::vector<int> sph_xf(any_sphere_xf ? (size_t)ns_pad : 0, 0);
        for (int64_t k = 0; k < ns; k++) {
            int32_t i = order[k];
            if (any_sphere_xf) sph_xf[k] = sc_xf[i];
            sph[k] = make_float4((float)sc_c[3 * i], (float)sc_c[3 * i + 1], (float)sc_c[3 * i + 2], (float)std::fabs(sc_r[i]));
            sph64[k] = make_double4(sc_c[3 * i], sc_c[3 * i + 1], sc_c[3 * i + 2], sc_r[i]);
            info[k] = make_int2(sc_obj[i], sc_mat[i]);
        }

        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        free_scene(ctx);
        s = rt::DScene{};
        int rc;
        std::vector<float4> nodes4(bvh.nodes.size() / 4);
        if (!nodes4.empty()) std::memcpy(nodes4.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(float));
        std::vector<uint8_t> texels;
        if (d->texel_bytes && d->texels) texels.assign(d->texels, d->texels + d->texel_bytes);
        if ((rc = upload(ctx, sph, &s.sph)) || (rc = upload(ctx, sph64, &s.sph64)) || (rc = upload(ctx, info, &s.sph_info)) ||
            (rc = upload(ctx, nodes4, &s.nodes)) || (rc = upload(ctx, wide, &s.wide)) || (rc = upload(ctx, rects, &s.rects)) || (rc = upload(ctx, rect_k, &s.rect_k)) ||
            (rc = upload(ctx, xforms, &s.xforms)) || (rc = upload(ctx, sph_xf, &s.sph_xform)) ||
            (rc = upload(ctx, mats, &s.mats)) || (rc = upload(ctx, texs, &s.texs)) || (rc = upload(ctx, texels, &s.texels))) {
            std::string keep = ctx->error;
            free_scene(ctx);
            ctx->error = keep;
            return rc;
        }
        n_nodes = (int)(bvh.nodes.size() / 16); root = use_bvh ? bvh.root : rt::kNoBvh; depth = bvh.depth;
        if (device_build) {
            // build over the uploaded (input order) spheres, then permute the arrays into leaf
            // order; the second set of arrays starts as a copy so that it carries the padding
            rt::LbvhResult lb;
            hipError_t e = rt::build_lbvh(s.sph64, (int)ns, ns > 65536 ? 2 : 4, ctx->stream, &lb);   // small leaves for large scenes, as the host builder
            const float4* sph2 = nullptr; const double4* sph642 = nullptr; const int2* info2 = nullptr; const int* xf2 = nullptr;
            if (e == hipSuccess) {
                ctx->scene_allocs.push_back(lb.nodes);
                if (lb.depth > rt::kMaxStack - 2) e = hipErrorInvalidValue;
            }
            if (e == hipSuccess && ((rc = upload(ctx, sph, &sph2)) || (rc = upload(ctx, sph64, &sph642)) ||
                                    (rc = upload(ctx, info, &info2)) || (rc = upload(ctx, sph_xf, &xf2)))) e = hipErrorOutOfMemory;
            if (e == hipSuccess)
                e = rt::gather_spheres(lb.order, (int)ns, s.sph, s.sph64, s.sph_info, s.sph_xform, const_cast<float4*>(sph2),
                                       const_cast<double4*>(sph642), const_cast<int2*>(info2), const_cast<int*>(xf2), ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (lb.order) (void)hipFree(lb.order);
            if (e != hipSuccess) {
                free_scene(ctx);
                // the caller did not ask for this builder: a tree its Morton order makes too deep
                // (strongly clustered or multi-scale scenes), or a failed build, goes to the host
                // builder, whose depth is bounded
                if (d->accel == RT_ACCEL_AUTO) { device_build = false; continue; }
                return fail(ctx, lb.depth > rt::kMaxStack - 2 ? RT_ERR_UNSUPPORTED : RT_ERR_HIP,
                            lb.depth > rt::kMaxStack - 2 ? std::string("BVH deeper than the traversal stack")
                                                         : std::string("GPU BVH build: ") + hipGetErrorString(e));
            }
            s.sph = sph2; s.sph64 = sph642; s.sph_info = info2; s.sph_xform = xf2;   // the input-order copies stay in scene_allocs until the next upload
            s.nodes = lb.nodes;
            n_nodes = lb.n_nodes; root = lb.root; depth = lb.depth;
        }
        break;
    }
    ctx->coarse_nodes = 0;
    if (use_bvh && n_nodes > 0 && root >= 0 && root != rt::kNoBvh) {
        // the root's two boxes span the tree
        float rn[16];
        RT_HIP(ctx, hipMemcpy(rn, s.nodes + 4 * (size_t)root, sizeof(rn), hipMemcpyDeviceToHost));
        for (int a = 0; a < 3; a++) {
            const float l = std::fmin(rn[a], rn[6 + a]), h = std::fmax(rn[3 + a], rn[9 + a]);
            s.tree_reach[a] = std::fmax(std::fabs(l), std::fabs(h));
            float step = (float)(((double)h - (double)l) * (1.0 + 1e-6) / 65535.0);
            if (!(step > 0.0f) || !std::isfinite(step)) step = 1e-30f;
            s.q_lo[a] = l; s.q_step[a] = step;
        }
    }
    if (use_bvh && n_nodes > 0 && n_nodes < (1 << 27) && root >= 0 && root != rt::kNoBvh) {
        // 32-byte nodes for walks out of HBM: planes on a 16-bit grid over the root's box
        float lo[3] = {s.q_lo[0], s.q_lo[1], s.q_lo[2]}, step[3] = {s.q_step[0], s.q_step[1], s.q_step[2]};
        void* qn = nullptr;
        RT_HIP(ctx, hipMalloc(&qn, (size_t)n_nodes * 32));
        ctx->scene_allocs.push_back(qn);
        RT_HIP(ctx, rt::quantize_nodes(s.nodes, n_nodes, lo, step, static_cast<uint4*>(qn), &ctx->coarse_nodes, ctx->stream));
        // a grid too coarse for some node (fine detail in a wide scene): the tree keeps its float form
        if (ctx->coarse_nodes == 0) {
            s.qnodes = static_cast<const uint4*>(qn);
        } else {
            ctx->scene_allocs.pop_back();
            (void)hipFree(qn);
        }
    }
    ctx->build_ms = use_bvh ? std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_build).count() : 0.0f;
    s.n_sph = (int)ns;
    s.n_sph_padded = (int)((ns + 3) / 4 * 4);      // what the linear scan walks; the arrays hold four records more
    s.n_nodes = n_nodes;
    s.root = root;
    s.bvh_depth = depth;
    s.use_bvh = use_bvh ? 1 : 0;
    s.ids_fit_16 = (d->n_prims < 65536 && d->n_mats < 65536) ? 1 : 0;
    s.has_textures = 0;
    for (const rt::DMaterial& m : mats) if (m.tex >= 0) s.has_textures = 1;
    // instance transforms the kernels have to look at: a rect's, and a sphere's where its material may
    // read a texture (a turned sphere is the sphere about the moved centre; only its texture turns)
    bool instances = false;
    for (const rt::DRect& r : rects) if (r.axes >> 8) instances = true;
    if (s.has_textures) {
        if (any_sphere_xf) instances = true;
        for (const rt::DWide& w : wide) if (w.xform) instances = true;
    }
    if (instances) s.has_textures |= 2;
    for (const rt::DMaterial& m : mats) if (m.type == RT_DIFFUSE_LIGHT || m.type == RT_LAMBERTIAN_BG) s.has_textures |= 4;
    s.n_wide = (int)wide.size();
    s.n_rects = (int)rects.size();
    s.background = d->background;
    for (int a = 0; a < 3; a++) s.bg[a] = d->bg_color[a];
    ctx->scene = s;
    ctx->accel_used = use_bvh ? RT_ACCEL_BVH : RT_ACCEL_LIST;
    ctx->bvh_nodes = n_nodes;
    ctx->bvh_depth = depth;
    ctx->max_radiance = max_radiance;
    ctx->has_scene = true;
    return RT_OK;
}

int rt_set_camera(rt_context* ctx, const rt_camera_desc* c) {
    if (!ctx) return RT_ERR_INVALID;
    if (!c) return fail(ctx, RT_ERR_INVALID, "null camera");
    if (!finite3(c->lookfrom) || !finite3(c->lookat) || !finite3(c->vup) || !(c->vfov > 0 && c->vfov < 180) ||
        !(c->aspect > 0) || !(c->aperture >= 0) || !std::isfinite(c->focus_dist))
        return fail(ctx, RT_ERR_INVALID, "camera: finite vectors, 0 < vfov < 180, aspect > 0, aperture >= 0 required");
    rt::DCamera cam{};
    build_camera(*c, &cam);
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(cam.u[a]) || !std::isfinite(cam.v[a]))
            return fail(ctx, RT_ERR_INVALID, "camera: lookfrom == lookat or vup parallel to the view direction");
    ctx->camera = cam;
    ctx->has_camera = true;
    return RT_OK;
}

// A caller's device buffer is a bare pointer at this boundary, and a kernel that writes past the end
// of one faults the GPU (which on a shared machine takes other people's work with it).  Where the
// runtime knows the allocation a pointer lies in, its extent is checked on the host before anything
// is launched: a buffer that is too small is an error code.  Pointers the runtime cannot place
// (sub-allocators over mapped ranges, memory of other APIs) are taken on trust, as before.
static int check_device_range(rt_context* ctx, const void* p, size_t bytes, const char* what) {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) != hipSuccess) {
        (void)hipGetLastError();                       // (not an error of this call: the pointer is just not the runtime's)
        return RT_OK;
    }
    const size_t off = (size_t)((const char*)p - (const char*)base);
    if (off > size || bytes > size - off)
        return fail(ctx, RT_ERR_INVALID, std::string(what) + ": the device buffer ends " + std::to_string(size - std::min(off, size)) +
                                         " bytes after the pointer, " + std::to_string(bytes) + " are written");
    return RT_OK;
}

int rt_render(rt_context* ctx, const rt_render_params* params, float* rgba_device) {
    rt::DRender rp{};
    int rc = validate_render(ctx, params, &rp);
    if (rc) return rc;
    if (!rgba_device) return fail(ctx, RT_ERR_INVALID, "null output");
    DeviceGuard g(ctx->device);
    const size_t bytes = (size_t)rp.rows * rp.width * ((rp.flags & RT_FLAG_FIXED_SUMS) ? 4 * sizeof(uint64_t) : sizeof(float4));
    if ((rc = check_device_range(ctx, rgba_device, bytes, "rt_render"))) return rc;
    return do_render(ctx, rp, reinterpret_cast<float4*>(rgba_device));
}

int rt_render_host(rt_context* ctx, const rt_render_params* params, float* rgba_host) {
    rt::DRender rp{};
    int rc = validate_render(ctx, params, &rp);
    if (rc) return rc;
    if (!rgba_host) return fail(ctx, RT_ERR_INVALID, "null output");
    DeviceGuard g(ctx->device);
    size_t bytes = (size_t)rp.rows * rp.width * ((rp.flags & RT_FLAG_FIXED_SUMS) ? 4 * sizeof(uint64_t) : sizeof(float4));
    if (bytes > ctx->staging_bytes) {
        if (ctx->staging) (void)hipFree(ctx->staging);
        ctx->staging = nullptr; ctx->staging_bytes = 0;
        RT_HIP(ctx, hipMalloc((void**)&ctx->staging, bytes));
        ctx->staging_bytes = bytes;
    }
    if ((rc = do_render(ctx, rp, ctx->staging))) return rc;
    RT_HIP(ctx, hipMemcpyAsync(rgba_host, ctx->staging, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

static int trace_primary(rt_context* ctx, const float* rays_host, int32_t n, float t_min, uint32_t flags, rt_hit* hits_host);

int rt_trace_primary(rt_context* ctx, const float* rays_host, int32_t n, float t_min, rt_hit* hits_host) {
    return trace_primary(ctx, rays_host, n, t_min, 0u, hits_host);
}

int rt_trace_primary_with(rt_context* ctx, const float* rays_host, int32_t n, float t_min, uint32_t flags, rt_hit* hits_host) {
    if (!ctx) return RT_ERR_INVALID;
    if (flags & ~(uint32_t)(RT_FLAG_NO_LDS_SCENE | RT_FLAG_FLOAT_NODES)) return fail(ctx, RT_ERR_INVALID, "rt_trace_primary_with: unknown flag");
    return trace_primary(ctx, rays_host, n, t_min, flags | 512u, hits_host);      // 512: the LDS walk where the scene fits
}

static int trace_primary(rt_context* ctx, const float* rays_host, int32_t n, float t_min, uint32_t flags, rt_hit* hits_host) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!rays_host || !hits_host))) return fail(ctx, RT_ERR_INVALID, "bad ray arrays");
    if (!(t_min >= 0.0f)) return fail(ctx, RT_ERR_INVALID, "t_min must be >= 0");
    if (n == 0) return RT_OK;
    static_assert(sizeof(rt_hit) == sizeof(rt::Hit), "rt_hit layout");
    DeviceGuard g(ctx->device);
    // rays in the first part of the context's buffer, hits behind them (on a 256-byte boundary)
    if ((size_t)n > ctx->trace_rays) {
        if (ctx->trace_buf) (void)hipFree(ctx->trace_buf);
        ctx->trace_buf = nullptr; ctx->trace_rays = 0;
        const size_t cap = ((size_t)n + 1023) / 1024 * 1024;
        RT_HIP(ctx, hipMalloc((void**)&ctx->trace_buf, cap * (6 * sizeof(float) + sizeof(rt::Hit))));
        ctx->trace_rays = cap;
    }
    float* d_rays = reinterpret_cast<float*>(ctx->trace_buf);
    rt::Hit* d_hits = reinterpret_cast<rt::Hit*>(ctx->trace_buf + ctx->trace_rays * 6 * sizeof(float));
    hipError_t e = hipMemcpyAsync(d_rays, rays_host, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt::launch_trace(ctx->scene, d_rays, n, t_min == 0.0f ? 0.001f : t_min, flags, d_hits, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hits_host, d_hits, (size_t)n * sizeof(rt::Hit), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, std::string("rt_trace_primary: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_trace_primary_noalloc(rt_context* ctx, const float* rays_host, int32_t n, float t_min, rt_hit* hits_host) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!rays_host || !hits_host))) return fail(ctx, RT_ERR_INVALID, "bad ray arrays");
    if (!(t_min >= 0.0f)) return fail(ctx, RT_ERR_INVALID, "t_min must be >= 0");
    if (n > RT_TRACE_NOALLOC_MAX_RAYS)
        return fail(ctx, RT_ERR_UNSUPPORTED, "rt_trace_primary_noalloc: " + std::to_string(n) + " rays exceed the fixed staging array of " +
                                             std::to_string(RT_TRACE_NOALLOC_MAX_RAYS) + " (split the batch, or use rt_trace_primary)");
    if (n == 0) return RT_OK;
    if (!ctx->noalloc) return fail(ctx, RT_ERR_HIP, "rt_trace_primary_noalloc: the staging array has no device address");
    DeviceGuard g(ctx->device);
    // every context and thread of this process shares the GPU's one array: calls on one GPU take turns
    // (the call waits for its stream anyway, so the lock costs nothing a lone caller could notice)
    std::lock_guard<std::mutex> turn(g_scratch_in_use[ctx->device]);
    // rays in the first part of the array, hits behind them.  Nothing has to start at zero: the
    // copy writes rays [0, n), the kernel writes every hit of [0, n) and only those are read back,
    // so what an earlier (longer) call left behind n is never looked at.
    float* d_rays = reinterpret_cast<float*>(ctx->noalloc);
    rt::Hit* d_hits = reinterpret_cast<rt::Hit*>(ctx->noalloc + kNoAllocRayBytes);
    hipError_t e = hipMemcpyAsync(d_rays, rays_host, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rt::launch_trace(ctx->scene, d_rays, n, t_min == 0.0f ? 0.001f : t_min, 0u, d_hits, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hits_host, d_hits, (size_t)n * sizeof(rt::Hit), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, std::string("rt_trace_primary_noalloc: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_quantize(rt_context* ctx, const float* rgba_device, int64_t n_pixels, int32_t spp, uint8_t* rgb8_device) {
    if (!ctx) return RT_ERR_INVALID;
    if (n_pixels < 0 || spp < 1 || (n_pixels > 0 && (!rgba_device || !rgb8_device)))
        return fail(ctx, RT_ERR_INVALID, "rt_quantize: bad arguments");
    DeviceGuard g(ctx->device);
    if (n_pixels > 0) {
        int rc;
        if ((rc = check_device_range(ctx, rgba_device, (size_t)n_pixels * sizeof(float4), "rt_quantize (sums)")) ||
            (rc = check_device_range(ctx, rgb8_device, (size_t)n_pixels * 3, "rt_quantize (bytes)"))) return rc;
    }
    RT_HIP(ctx, rt::launch_quantize(reinterpret_cast<const float4*>(rgba_device), n_pixels, spp, rgb8_device, ctx->stream));
    return RT_OK;
}

int rt_resolve_sums(rt_context* ctx, const uint64_t* sums_device, int64_t n_pixels, int32_t mean, float* rgba_device) {
    if (!ctx) return RT_ERR_INVALID;
    if (n_pixels < 0 || (n_pixels > 0 && (!sums_device || !rgba_device)))
        return fail(ctx, RT_ERR_INVALID, "rt_resolve_sums: bad arguments");
    DeviceGuard g(ctx->device);
    if (n_pixels > 0) {
        int rc;
        if ((rc = check_device_range(ctx, sums_device, (size_t)n_pixels * 4 * sizeof(uint64_t), "rt_resolve_sums (sums)")) ||
            (rc = check_device_range(ctx, rgba_device, (size_t)n_pixels * sizeof(float4), "rt_resolve_sums (image)"))) return rc;
    }
    RT_HIP(ctx, rt::launch_resolve(reinterpret_cast<const unsigned long long*>(sums_device), n_pixels, mean != 0,
                                   reinterpret_cast<float4*>(rgba_device), ctx->stream));
    return RT_OK;
}

int rt_render_guides(rt_context* ctx, int32_t width, int32_t height, float* guides_device) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "no scene uploaded");
    if (!ctx->has_camera) return fail(ctx, RT_ERR_NO_CAMERA, "no camera set");
    if (width < 2 || height < 2 || width > 65536 || height > 65536) return fail(ctx, RT_ERR_INVALID, "width/height must be in [2, 65536]");
    if (!guides_device) return fail(ctx, RT_ERR_INVALID, "null output");
    DeviceGuard g(ctx->device);
    int rc;
    if ((rc = check_device_range(ctx, guides_device, (size_t)width * height * 2 * sizeof(float4), "rt_render_guides"))) return rc;
    RT_HIP(ctx, rt::launch_guides(ctx->scene, ctx->camera, width, height, 0.001f, reinterpret_cast<float4*>(guides_device), ctx->stream));
    return RT_OK;
}

int rt_reconstruct(rt_context* ctx, const float* rgba_device, const float* guides_device, int32_t width, int32_t height,
                   const rt_reconstruct_params* p, float* rgba_out_device) {
    if (!ctx) return RT_ERR_INVALID;
    if (p && p->struct_size != sizeof(rt_reconstruct_params)) return fail(ctx, RT_ERR_INVALID, "rt_reconstruct_params: struct_size mismatch");
    if (width < 1 || height < 1 || width > 65536 || height > 65536) return fail(ctx, RT_ERR_INVALID, "width/height must be in [1, 65536]");
    if (!rgba_device || !guides_device || !rgba_out_device) return fail(ctx, RT_ERR_INVALID, "rt_reconstruct: null buffer");
    rt::DReconstruct rp{};
    rp.width = width; rp.height = height;
    rp.iterations = p && p->iterations ? p->iterations : 4;
    rp.sigma_color = p ? p->sigma_color : 0.0f;
    rp.sigma_normal = p && p->sigma_normal != 0.0f ? p->sigma_normal : 16.0f;
    rp.sigma_depth = p && p->sigma_depth != 0.0f ? p->sigma_depth : 0.05f;
    rp.albedo_floor = p && p->albedo_floor != 0.0f ? p->albedo_floor : 0.02f;
    rp.no_lds = p && (p->flags & RT_RECON_NO_LDS) ? 1 : 0;
    if (rp.iterations < 1 || rp.iterations > 8) return fail(ctx, RT_ERR_INVALID, "rt_reconstruct: iterations must be in [1, 8]");
    if (!(rp.sigma_color >= 0.0f) || !(rp.sigma_normal > 0.0f) || !(rp.sigma_depth > 0.0f) || !(rp.albedo_floor > 0.0f) ||
        !std::isfinite(rp.sigma_color) || !std::isfinite(rp.sigma_normal) || !std::isfinite(rp.sigma_depth) || !std::isfinite(rp.albedo_floor))
        return fail(ctx, RT_ERR_INVALID, "rt_reconstruct: sigmas and albedo_floor must be finite and positive (sigma_color 0 = default)");
    DeviceGuard g(ctx->device);
    const size_t n = (size_t)width * height;
    int rc;
    if ((rc = check_device_range(ctx, rgba_device, n * sizeof(float4), "rt_reconstruct (frame)")) ||
        (rc = check_device_range(ctx, guides_device, n * 2 * sizeof(float4), "rt_reconstruct (guides)")) ||
        (rc = check_device_range(ctx, rgba_out_device, n * sizeof(float4), "rt_reconstruct (result)"))) return rc;
    if (n > ctx->recon_pixels) {
        if (ctx->recon_tmp) (void)hipFree(ctx->recon_tmp);          // (waits for the launches that use it)
        ctx->recon_tmp = nullptr; ctx->recon_pixels = 0;
        RT_HIP(ctx, hipMalloc((void**)&ctx->recon_tmp, 2 * n * sizeof(float4)));
        ctx->recon_pixels = n;
    }
    RT_HIP(ctx, rt::launch_reconstruct(rp, reinterpret_cast<const float4*>(rgba_device), reinterpret_cast<const float4*>(guides_device),
                                       ctx->recon_tmp, reinterpret_cast<float4*>(rgba_out_device), ctx->stream));
    return RT_OK;
}

int rt_upscale(rt_context* ctx, const float* rgba_low_device, const float* guides_low_device, int32_t low_width, int32_t low_height,
               const float* guides_high_device, int32_t width, int32_t height, const rt_reconstruct_params* p, float* rgba_out_device) {
    if (!ctx) return RT_ERR_INVALID;
    if (p && p->struct_size != sizeof(rt_reconstruct_params)) return fail(ctx, RT_ERR_INVALID, "rt_reconstruct_params: struct_size mismatch");
    if (low_width < 2 || low_height < 2 || width < 2 || height < 2 || width > 65536 || height > 65536)
        return fail(ctx, RT_ERR_INVALID, "rt_upscale: sizes must be in [2, 65536]");
    if (width < low_width || height < low_height || width > 4 * low_width || height > 4 * low_height)
        return fail(ctx, RT_ERR_INVALID, "rt_upscale: the high resolution must be 1 to 4 times the low one per axis");
    if (!rgba_low_device || !guides_low_device || !guides_high_device || !rgba_out_device) return fail(ctx, RT_ERR_INVALID, "rt_upscale: null buffer");
    rt::DReconstruct rp{};
    rp.width = width; rp.height = height; rp.iterations = 0; rp.sigma_color = 0.0f;
    rp.sigma_normal = p && p->sigma_normal != 0.0f ? p->sigma_normal : 16.0f;
    rp.sigma_depth = p && p->sigma_depth != 0.0f ? p->sigma_depth : 0.05f;
    rp.albedo_floor = p && p->albedo_floor != 0.0f ? p->albedo_floor : 0.02f;
    if (!(rp.sigma_normal > 0.0f) || !(rp.sigma_depth > 0.0f) || !(rp.albedo_floor > 0.0f) || !std::isfinite(rp.sigma_normal) ||
        !std::isfinite(rp.sigma_depth) || !std::isfinite(rp.albedo_floor))
        return fail(ctx, RT_ERR_INVALID, "rt_upscale: sigmas and albedo_floor must be finite and positive");
    DeviceGuard g(ctx->device);
    const size_t nl = (size_t)low_width * low_height, nh = (size_t)width * height;
    int rc;
    if ((rc = check_device_range(ctx, rgba_low_device, nl * sizeof(float4), "rt_upscale (low frame)")) ||
        (rc = check_device_range(ctx, guides_low_device, nl * 2 * sizeof(float4), "rt_upscale (low guides)")) ||
        (rc = check_device_range(ctx, guides_high_device, nh * 2 * sizeof(float4), "rt_upscale (high guides)")) ||
        (rc = check_device_range(ctx, rgba_out_device, nh * sizeof(float4), "rt_upscale (result)"))) return rc;
    RT_HIP(ctx, rt::launch_upscale(rp, low_width, low_height, reinterpret_cast<const float4*>(rgba_low_device),
                                   reinterpret_cast<const float4*>(guides_low_device), reinterpret_cast<const float4*>(guides_high_device),
                                   reinterpret_cast<float4*>(rgba_out_device), ctx->stream));
    return RT_OK;
}

int rt_get_stats(rt_context* ctx, rt_stats* out) {
    if (!ctx || !out) return RT_ERR_INVALID;
    // the caller says how large its rt_stats is; fields added after its header was written are not
    // written into it
    constexpr size_t kFirstSized = offsetof(rt_stats, bvh_coarse_nodes) + sizeof(int32_t);
    const size_t caller = out->struct_size;
    if (caller < kFirstSized) return fail(ctx, RT_ERR_INVALID, "rt_stats: set struct_size to sizeof(rt_stats) before the call");
    DeviceGuard g(ctx->device);
    rt_stats st;
    std::memset(&st, 0, sizeof(st));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long c[rt::kNumCounters + rt::kNumPhaseClocks];
    RT_HIP(ctx, hipMemcpy(c, ctx->counters, sizeof(c), hipMemcpyDeviceToHost));
    st.rays = c[0]; st.samples = c[1];
    st.box_tests = c[2]; st.sphere_tests = c[3]; st.sphere_hits = c[4];
    st.rect_tests = c[5]; st.wide_tests = c[6]; st.shades = c[7];
    for (int q = 0; q < rt::kNumPhaseClocks; q++) st.phase_clocks[q] = c[rt::kNumCounters + q];
    if (ctx->timed) RT_HIP(ctx, hipEventElapsedTime(&st.kernel_ms, ctx->ev_start, ctx->ev_stop));
    st.lanes_per_pixel = ctx->last_k;
    st.accel = ctx->accel_used;
    st.bvh_nodes = ctx->bvh_nodes;
    st.bvh_depth = ctx->bvh_depth;
    st.build_ms = ctx->build_ms;
    st.bvh_coarse_nodes = ctx->coarse_nodes;
    const size_t n = caller < sizeof(rt_stats) ? caller : sizeof(rt_stats);
    st.struct_size = (uint32_t)n;
    std::memcpy(out, &st, n);
    return RT_OK;
}

}  // extern "C"
