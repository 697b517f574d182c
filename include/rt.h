/* rt.h — C-ABI of the MI355X (gfx950) render path.
 *
 * The reference compiles scene and integrator into one translation unit (main.cc:3-8,
 * cuda/main.cu:12-20) and has no plugin boundary of its own; this header is the boundary a
 * maintainer would put where the reference launches its device work:
 *
 *   reference call site                                   entry point here
 *   ---------------------------------------------------   -------------------------
 *   create_world<<<1,1>>>        cuda/main.cu:39-189,520   rt_upload_scene
 *   load_texture: stbi_load      cuda/main.cu:393          rt_load_jpeg_file / rt_decode_jpeg
 *   load_texture (H2D part)      cuda/main.cu:385-436      rt_upload_scene (texels)
 *   rotate_y wrappers            cuda/sceneobject.h:31-109 rt_upload_scene (xform, prim_xform)
 *   camera ctor / update_scene   camera.h:19-38,
 *                                cuda/main.cu:220-279,564  rt_set_camera
 *   init_render + render<<<>>>   cuda/main.cu:281-355,582  rt_render / rt_render_host
 *   pixel/sample loop, ray_color main.cc:10-35,111-126     rt_render / rt_render_host
 *   world.hit (closest hit)      main.cc:18,
 *                                sceneobject_list.h:26-39  rt_trace_primary (+ _with, _noalloc)
 *   write_color / output_image   color.h:9-26,
 *                                cuda/main.cu:358-383      rt_quantize
 *   low / high frame pairs       cuda/main.cu:557-576,
 *                                README.md:7               rt_render_guides + rt_reconstruct + rt_upscale (no code in the reference)
 *   gpuErrchk                    cuda/main.cu:24-32        return codes + rt_last_error
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_error; nothing exits or
 * throws.  A context belongs to one GPU and is driven by one host thread at a time;
 * different contexts may be driven concurrently.  All descriptors are deep-copied by the
 * call that takes them.  Images are float4 per pixel, row 0 = BOTTOM row (the reference's
 * framebuffer orientation, cuda/main.cu:338,364), rgb = sum (or mean) of sample radiance,
 * a = number of samples.
 */
#ifndef RT_H
#define RT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

typedef enum rt_error {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument or descriptor */
    RT_ERR_NO_DEVICE = -2,    /* no such GPU / not a gfx950 code object */
    RT_ERR_HIP = -3,          /* a HIP runtime call failed (text in rt_last_error) */
    RT_ERR_NO_SCENE = -4,     /* rt_upload_scene has not succeeded yet */
    RT_ERR_NO_CAMERA = -5,    /* rt_set_camera has not succeeded yet */
    RT_ERR_UNSUPPORTED = -6   /* valid request this build cannot serve (limits in rt_last_error) */
} rt_error;

/* primitive tags, in list order (sceneobject_list.h:21; rect kinds cuda/rect.h:8,54,101) */
enum { RT_SPHERE = 0, RT_XY_RECT = 1, RT_XZ_RECT = 2, RT_YZ_RECT = 3 };
/* material tags (material.h:15,35,51; cuda/material.h:33,103) */
enum { RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIELECTRIC = 2, RT_DIFFUSE_LIGHT = 3, RT_LAMBERTIAN_BG = 4 };
/* texture tags (cuda/texture.h:17,32,55) */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };
/* radiance of a ray that leaves the scene: the sky gradient of main.cc:30-34, or a constant */
enum { RT_BG_SKY = 0, RT_BG_SOLID = 1 };
/* closest-hit strategy: the reference's linear scan (sceneobject_list.h:26-39) or a BVH
 * (cuda/bvh.h); both return the same hit */
enum { RT_ACCEL_AUTO = 0, RT_ACCEL_LIST = 1, RT_ACCEL_BVH = 2,
       RT_ACCEL_BVH_DEVICE = 3 /* BVH built on the GPU from Morton order (where cuda/bvh.h:55-89 builds on
                                  the device too): milliseconds instead of a host build, a somewhat
                                  slower tree; for scenes that are rebuilt often */ };

typedef struct rt_scene_desc {
    uint32_t struct_size;        /* sizeof(rt_scene_desc) */
    int32_t n_prims;
    const int32_t* prim_type;    /* [n_prims] RT_SPHERE.. */
    const double* prim_geom;     /* [n_prims][6] sphere: cx cy cz r 0 0
                                    rect:   a0 a1 b0 b1 k 0 (argument order of cuda/rect.h:19,66,112) */
    const int32_t* prim_mat;     /* [n_prims] index into the material arrays */
    int32_t n_mats;
    const int32_t* mat_type;     /* [n_mats] */
    const double* mat_param;     /* [n_mats][5] albedo r g b, fuzz, index of refraction */
    const int32_t* mat_tex;      /* [n_mats] texture index, or -1: albedo is the colour */
    int32_t n_tex;
    const int32_t* tex_type;     /* [n_tex] */
    const double* tex_color;     /* [n_tex][3] RT_TEX_SOLID */
    const int32_t* tex_a;        /* RT_TEX_CHECKER: even texture | RT_TEX_IMAGE: width  */
    const int32_t* tex_b;        /* RT_TEX_CHECKER: odd texture  | RT_TEX_IMAGE: height */
    const int64_t* tex_off;      /* RT_TEX_IMAGE: byte offset of the image in texels */
    const uint8_t* texels;       /* RGB8 rows, top row first (what stbi_load returns, cuda/main.cu:393) */
    uint64_t texel_bytes;
    int32_t background;          /* RT_BG_* */
    float bg_color[3];
    int32_t accel;               /* RT_ACCEL_* */
    double wide_radius;          /* spheres with a larger radius are intersected in double
                                    (0 = default 64.0) */
    /* Instance transforms (cuda/sceneobject.h:31-109 rotate_y, plus a translation): primitive i
     * with prim_xform[i] = x >= 0 stands at  world = R_y(angle_x) * object + offset_x ; -1 (or a
     * NULL prim_xform) leaves it where its geometry says.  Nested wrappers compose into this one
     * form on the caller's side.  A descriptor whose struct_size ends before n_xforms has none. */
    int32_t n_xforms;
    const double* xform;         /* [n_xforms][4] angle in degrees about +y, offset x y z */
    const int32_t* prim_xform;   /* [n_prims] or NULL */
} rt_scene_desc;

/* camera.h:19-20 argument list */
typedef struct rt_camera_desc {
    double lookfrom[3], lookat[3], vup[3];
    double vfov;                 /* vertical field of view, degrees */
    double aspect, aperture, focus_dist;
} rt_camera_desc;

enum {
    RT_FLAG_MEAN = 1u << 0,       /* rgb = mean radiance instead of the sum over samples */
    RT_FLAG_NO_REFINE = 1u << 1,  /* skip the double-precision polish of the winning hit */
    RT_FLAG_COUNTERS = 1u << 2,   /* also count box/sphere/rect tests (slower; for the roofline) */
    RT_FLAG_PHASE_CLOCKS = 1u << 3, /* diagnostic build: wave clocks per phase into rt_stats.phase_clocks */
    RT_FLAG_NO_LDS_SCENE = 1u << 5, /* read BVH and spheres from HBM even when they would fit in LDS */
    RT_FLAG_FLOAT_NODES = 1u << 7, /* a tree read from HBM is walked through its 64-byte single-precision nodes
                                     instead of the 32-byte quantised ones (same image; for comparison) */
    RT_FLAG_ALL_FEATURES = 1u << 8, /* render with the kernel that carries the code of every scene feature (rects, textures,
                                     instance transforms) instead of the one built for what this scene holds (same image;
                                     for comparison) */
    RT_FLAG_FIXED_SUMS = 1u << 6  /* the image is written as the kernel accumulates it: per pixel four uint64_t
                                     (r, g, b sums of sample radiance in unsigned 32.32 fixed point, sample
                                     count), 32 bytes.  Integer sums of sample windows (sample_begin) add up
                                     exactly, in any order: rt_resolve_sums of the added windows is bit for bit
                                     the frame one call renders with all the samples.  Excludes RT_FLAG_MEAN. */
};

/* Range of a pixel: the kernels keep a pixel's sum of sample radiance per channel in unsigned 32.32 fixed
 * point (what makes an image independent of the order samples are added in, on one GPU or many), so the
 * sum must stay below 2^32, where main.cc:121 sums in double.  A sample is at most the scene's brightest
 * emitter or background colour (albedos are at most 1), so rt_render and rt_render_host return
 * RT_ERR_UNSUPPORTED when (sample_begin + spp) x that brightness reaches 2^32 -- an emitter of 10^7 takes at
 * most 429 samples -- instead of wrapping; split such a frame into calls of fewer samples and combine their
 * means in floating point.  (Integer sums of sample windows added by the caller, RT_FLAG_FIXED_SUMS, are the
 * caller's to keep below 2^32 per channel in the same way; an albedo above 1 is not accounted for.) */
typedef struct rt_render_params {
    uint32_t struct_size;        /* sizeof(rt_render_params) */
    int32_t width, height;       /* full image, also when only some row tiles are rendered */
    int32_t spp;                 /* samples rendered by this call */
    int32_t sample_begin;        /* index of the first one (sample-split: rank g starts at g*spp) */
    int32_t max_depth;           /* ray_bounce_limit (main.cc:94) */
    uint64_t seed;
    int32_t tile_rows;           /* row-tile sharding: rows are cut into tiles of tile_rows rows; */
    int32_t tile_first;          /* this call renders tiles tile_first, tile_first+tile_step, ... */
    int32_t tile_step;           /* and packs their rows in that order.  0,0,0 = whole image */
    float t_min;                 /* main.cc:18 uses 0.001; 0 selects that */
    uint32_t flags;              /* RT_FLAG_* */
    int32_t lanes_per_pixel;     /* K = 1,2,4,...,64: a wavefront's tile holds 64/K pixels whose samples its
                                    64 lanes take dynamically; 0 = choose from the frame size */
    /* Traversal tuning (the image does not depend on any of it).  A struct whose struct_size ends before
     * traversal_slice leaves all of it at the defaults. */
    int32_t traversal_slice;     /* BVH node steps a wavefront spends per trip of its path loop before lanes whose
                                    rays are finished shade and start new rays: 0 = chosen from the tree depth,
                                    -1 = never leave a walk early, n > 0 = n steps (at most 65534; more is
                                    RT_ERR_INVALID) */
    int32_t leaf_quorum;         /* time-sliced walk: leaves are tested once (lanes waiting at a leaf) x this
                                    >= lanes still walking; 0 = default (2 with the tree in LDS, 4 with it in
                                    device memory), -1 = wait for every lane, at most 64 (a wavefront's lanes) */
    int32_t stack_lds_entries;   /* time-sliced walk of a tree in HBM: entries of a lane's traversal stack kept in LDS; a
                                    deeper tree's walk keeps the rest in device memory the context allocates.  0 = default
                                    (24: what leaves six workgroups per CU), else 2 .. 48.  A struct whose struct_size ends
                                    before this field leaves it at the default. */
    int32_t traversal_exit_share; /* the wavefront also leaves a walk as soon as the lanes whose walk is unfinished are no
                                    more than this many 128ths of the lanes that came in with one (finished lanes then
                                    shade and start new rays while most of them are free; the others resume): in a
                                    time-sliced walk after any step, in the plain walk of a shallower tree between two
                                    rounds of leaves (scenes of spheres only, traversal_slice 0).  0 = default (64 and 32),
                                    -1 = off, else 1 .. 128.  Read with stack_lds_entries (it fills what was padding
                                    behind it: zero the struct). */
} rt_render_params;

/* one closest-hit answer (hit_record of sceneobject.h:9-14 plus the object's list position) */
typedef struct rt_hit {
    int32_t id;                  /* position in the primitive list, -1 = miss */
    float t;
    float p[3];
    float n[3];                  /* unit normal facing against the ray (sceneobject.h:16-23) */
    int32_t front_face;
    float u, v;                  /* surface coordinates (cuda/sphere.h:20-26, cuda/rect.h:43-44) */
} rt_hit;

typedef struct rt_stats {
    uint32_t struct_size;        /* IN: sizeof(rt_stats) as the caller was compiled; rt_get_stats writes that many bytes
                                    at most, so the struct can grow at its end without writing past an older caller's
                                    object (since version 0.3: the layout before it had no such field) */
    uint32_t reserved;
    uint64_t rays;               /* closest-hit queries (one per world.hit call, main.cc:18) */
    uint64_t samples;            /* camera samples */
    uint64_t box_tests, sphere_tests, sphere_hits, rect_tests, wide_tests, shades; /* RT_FLAG_COUNTERS */
    uint64_t phase_clocks[5];    /* RT_FLAG_PHASE_CLOCKS: [0] unit hand-out (and camera rays), [1] always-tested primitives / scan,
                                    [2] tree walk, [3] the draws of a trip (kernels that draw in one place: camera rays and scatter),
                                    [4] shading (records, materials; scatter too in the other kernels).  With RT_FLAG_COUNTERS [0] and [1]
                                    hold the traversal steps of all lanes and of the waves instead */
    float kernel_ms;             /* device time of the last render kernel (HIP events on its stream) */
    int32_t lanes_per_pixel;     /* value the last render used */
    int32_t accel;               /* RT_ACCEL_LIST or RT_ACCEL_BVH actually used (either builder) */
    int32_t bvh_nodes, bvh_depth;
    float build_ms;              /* wall time rt_upload_scene spent building the BVH and ordering the spheres */
    int32_t bvh_coarse_nodes;    /* inner nodes whose boxes the 16-bit grid of the 32-byte nodes would blow up to more than
                                    twice their surface area; 0: a walk out of HBM reads the 32-byte nodes, else the
                                    64-byte ones */
} rt_stats;

int rt_create(int device, rt_context** out);
void rt_destroy(rt_context* ctx);

/* Use an existing hipStream_t for all device work of this context (NULL = the null stream). */
int rt_set_stream(rt_context* ctx, void* hip_stream);

int rt_upload_scene(rt_context* ctx, const rt_scene_desc* scene);
int rt_set_camera(rt_context* ctx, const rt_camera_desc* camera);

/* Rows this call would render for the given sharding (for sizing the output buffer). */
int rt_shard_rows(int32_t height, int32_t tile_rows, int32_t tile_first, int32_t tile_step);

/* Render into DEVICE memory: rgba_device holds rt_shard_rows(..) * width float4 (with
 * RT_FLAG_FIXED_SUMS: as many 4 x uint64_t).  Asynchronous on the context's stream.  Where the HIP
 * runtime knows the allocation the pointer lies in (hipMalloc and the like), a buffer that ends before
 * the last byte the call writes is RT_ERR_INVALID, checked on the host before anything is launched;
 * rt_quantize and rt_resolve_sums check their buffers the same way. */
int rt_render(rt_context* ctx, const rt_render_params* params, float* rgba_device);
/* Same into HOST memory; returns when the image is there. */
int rt_render_host(rt_context* ctx, const rt_render_params* params, float* rgba_host);

/* Closest hit of n explicit rays (host arrays; rays = n x (ox oy oz dx dy dz)).  A tree is walked as a
 * frame walks one that does not fit in LDS: through its 32-byte nodes where the scene has them
 * (rt_stats.bvh_coarse_nodes == 0), else through the 64-byte ones.  The device copies of rays and hits live
 * in a buffer the context keeps, and grows when n exceeds every earlier n. */
int rt_trace_primary(rt_context* ctx, const float* rays_host, int32_t n, float t_min, rt_hit* hits_host);

/* rt_trace_primary with the choice of walk a frame has (same call site in the reference: world.hit, main.cc:18):
 * flags 0 = the walk a frame of this scene uses (tree and spheres in LDS where they fit, else the 32-byte nodes out of
 * device memory), RT_FLAG_NO_LDS_SCENE = never the LDS walk (what rt_trace_primary does), RT_FLAG_FLOAT_NODES = the
 * 64-byte single-precision nodes.  Same answers whichever walk; for holding each walk against a checker with explicit
 * rays.  Other flag bits are RT_ERR_INVALID. */
int rt_trace_primary_with(rt_context* ctx, const float* rays_host, int32_t n, float t_min, uint32_t flags, rt_hit* hits_host);

/* rt_trace_primary for callers that cannot allocate device memory: tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference (its world.hit runs on host memory it already owns).
 * Same arguments, same answers; the device copies of rays and hits live in a fixed array inside the
 * library's code object (RT_TRACE_NOALLOC_MAX_RAYS rays: 1.5 MB of rays + 3 MB of hits per GPU, held until
 * the process ends), so the call makes no hipMalloc / hipFree and never grows anything.  n above the
 * maximum is RT_ERR_UNSUPPORTED (split the batch).  Every call on one GPU, from any context, uses that
 * GPU's one array: calls of one process on one GPU take turns (a lock per GPU inside the library, held
 * until the call's answers are on the host), so contexts driven by different threads may call it freely;
 * calls on different GPUs are independent.  The host arrays are copied with hipMemcpyAsync:
 * pass pinned memory where the runtime's own staging of pageable memory is not acceptable either. */
#define RT_TRACE_NOALLOC_MAX_RAYS 65536
int rt_trace_primary_noalloc(rt_context* ctx, const float* rays_host, int32_t n, float t_min, rt_hit* hits_host);

/* write_color (color.h:9-26) on device memory: n_pixels float4 sums -> n_pixels x 3 bytes. */
int rt_quantize(rt_context* ctx, const float* rgba_device, int64_t n_pixels, int32_t spp,
                uint8_t* rgb8_device);

/* RT_FLAG_FIXED_SUMS images -> float4 images: n_pixels x (r, g, b, samples) uint64_t in device memory become
 * n_pixels float4 (rgb = sum, or mean with mean != 0; a = samples), exactly as rt_render converts its own
 * sums.  Asynchronous on the context's stream. */
int rt_resolve_sums(rt_context* ctx, const uint64_t* sums_device, int64_t n_pixels, int32_t mean, float* rgba_device);

/* Low-to-high sample-count reconstruction.  The reference names such a stage as ongoing work (README.md:7:
 * a learned upscaler "similar to DLSS") and lays out the frame pairs for it (cuda/main.cu:557-576: every frame
 * once with few and once with many samples); it holds no code, weights or output of it, so these two entry
 * points have no counterpart there to be identical to.  They turn the low frame of such a pair into an
 * estimate of the high one:
 *
 * rt_render_guides: what the ray through the centre of every pixel (from the centre of the lens) hits first
 *   under the context's scene and camera: two planes of width x height float4, rows bottom-up like the image;
 *   plane 0 = (unit normal facing against the ray, distance t), plane 1 = (albedo r g b of the surface there --
 *   texture value, emitted colour, white for a dielectric --, list position of the object as int32 bits).  A ray
 *   that leaves the scene has normal 0, t 0, albedo 1, position -1.  Asynchronous on the context's stream.
 *
 * rt_reconstruct: the edge-avoiding a-trous wavelet transform (Dammertz, Sewtz, Hanika, Lensch, HPG 2010) of the
 *   frame's mean radiance: `iterations` passes of a 5x5 B3-spline kernel whose taps stand 1, 2, 4, ... pixels
 *   apart, a tap counting by how well it agrees with the centre pixel in first-hit object (equal or not),
 *   normal (cosine to the power sigma_normal), distance (exp(-|t_p - t_q| / (sigma_depth t_p))) and colour
 *   (exp(-|c_p - c_q|^2 / s^2), s = sigma_color halved from pass to pass), on colours divided by the first hit's
 *   albedo (at least albedo_floor) so that texture detail is not blurred.  rgba_device: a frame as rt_render
 *   leaves it (rgb = sum over a samples); rgba_out_device: the same convention (rgb = reconstructed mean x a); the
 *   two may be the same buffer.  The context keeps two images of scratch memory, grown on demand.  params NULL =
 *   defaults.  Asynchronous on the context's stream. */
typedef struct rt_reconstruct_params {
    uint32_t struct_size;        /* sizeof(rt_reconstruct_params) */
    int32_t iterations;          /* 1 .. 8; 0 = default (4) */
    float sigma_color;           /* 0 = default: 1.6 / sqrt(samples of the pixel) */
    float sigma_normal;          /* 0 = default (16) */
    float sigma_depth;           /* 0 = default (0.05) */
    float albedo_floor;          /* 0 = default (0.02) */
    uint32_t flags;              /* RT_RECON_NO_LDS: every pass reads its taps through the caches instead of staging the
                                    close ones (steps 1, 2, 4) in LDS (same image; for comparison) */
} rt_reconstruct_params;
enum { RT_RECON_NO_LDS = 1u << 0 };
int rt_render_guides(rt_context* ctx, int32_t width, int32_t height, float* guides_device);
/* rt_upscale: a frame rendered at a low resolution onto a higher-resolution pixel grid (the "upscaler" half of
 *   README.md:7), by joint bilateral upsampling (Kopf, Cohen, Lischinski, Uyttendaele, SIGGRAPH 2007): a
 *   high-resolution pixel takes the 3x3 low-resolution pixels around its own position in the image, each counted
 *   by a tent in image distance and by how well the first hit of its centre ray (guides_low_device:
 *   rt_render_guides at the low size) agrees with the high-resolution pixel's (guides_high_device:
 *   rt_render_guides at the high size; same camera) in object, normal and distance; colours are divided by the
 *   low-resolution albedo and multiplied by the high-resolution one, so texture detail and silhouettes are those
 *   of the high-resolution guides.  Any ratio of sizes up to 4 per axis; typically the low frame has been through
 *   rt_reconstruct first.  iterations and sigma_color of params are not used.  Same image convention as
 *   rt_reconstruct (the sample count of the nearest low-resolution pixel rides along).  Asynchronous. */
int rt_upscale(rt_context* ctx, const float* rgba_low_device, const float* guides_low_device, int32_t low_width, int32_t low_height,
               const float* guides_high_device, int32_t width, int32_t height, const rt_reconstruct_params* params,
               float* rgba_out_device);
int rt_reconstruct(rt_context* ctx, const float* rgba_device, const float* guides_device, int32_t width, int32_t height,
                   const rt_reconstruct_params* params, float* rgba_out_device);

/* Waits for the context's stream, then reports the last render.  out->struct_size must be set by the caller
 * to sizeof(rt_stats) of the header it was compiled against; a size smaller than the struct of version 0.3
 * (which ends with bvh_coarse_nodes) is RT_ERR_INVALID. */
int rt_get_stats(rt_context* ctx, rt_stats* out);

const char* rt_last_error(rt_context* ctx);
const char* rt_version(void);

/* Texture files (host side, no context): what stbi_load(filename, &w, &h, &n, 3) does for
 * load_texture (cuda/main.cu:393) -- a JPEG file (baseline or progressive, 8 bit, any 1x/2x/4x
 * chroma subsampling, restart intervals, grey / YCbCr / RGB / CMYK) decoded to RGB8 rows, top row
 * first, ready to be an RT_TEX_IMAGE entry.  *rgb8 is allocated by the library: release it with
 * rt_free_image.  Errors: RT_ERR_INVALID (unreadable or corrupt), RT_ERR_UNSUPPORTED (12-bit,
 * arithmetic-coded, lossless); text from rt_image_last_error (per thread). */
int rt_decode_jpeg(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t** rgb8);
int rt_load_jpeg_file(const char* path, int32_t* width, int32_t* height, uint8_t** rgb8);
void rt_free_image(uint8_t* rgb8);
const char* rt_image_last_error(void);

/* Frame files (host side, no context): output_image (cuda/main.cu:358-383) and the output loop of
 * main.cc:109-126 -- "P3", width height, 255, then one pixel per line as decimal text, top row first;
 * binary != 0 writes the same pixels as a binary "P6" file instead.  rgb8: height x width x 3 bytes with
 * row 0 = the BOTTOM row, which is what rt_quantize leaves (the reference's loops count j downwards
 * for the same reason).  Errors: RT_ERR_INVALID (bad arguments, or the file cannot be written: text
 * from rt_image_last_error). */
int rt_write_ppm(const char* path, const uint8_t* rgb8, int32_t width, int32_t height, int32_t binary);

#ifdef __cplusplus
}
#endif
#endif /* RT_H */
