/* TEST INFRASTRUCTURE — CPU restatement of the reference's per-pixel render path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The shipped path (raytrace_amd/) never
 * links, imports or falls back to it.
 *
 * Parity pinning: the double-precision variant (suffix _f64) is checked against
 * vectors and images produced by the reference's own compiled main.cc
 * (oracle/ref_harness.cc -> tests/golden/).  The reference ships no tests or golden
 * files of its own, so those generated fixtures are the only pin.  Features that
 * exist only in the reference's CUDA tree (rects, emitters, textures, camera
 * animation) cannot be compiled here (no nvcc): for them this file restates the
 * CUDA tree's formulas in double and is "parity unpinned" beyond code review.
 *
 * Every function exists twice: `_f64` (all arithmetic in double, like the
 * reference's CPU tree) and `_f32` (all arithmetic in float, like its CUDA tree);
 * the float variant is used to calibrate how far a single-precision renderer may
 * drift from the double one on the same random stream.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* primitive types, in list order (sceneobject_list.h:21 add order) */
enum { RTO_SPHERE = 0, RTO_XY_RECT = 1, RTO_XZ_RECT = 2, RTO_YZ_RECT = 3 };
/* material types */
enum { RTO_LAMBERTIAN = 0, RTO_METAL = 1, RTO_DIELECTRIC = 2,
       RTO_DIFFUSE_LIGHT = 3, RTO_LAMBERTIAN_BG = 4 };
/* texture types */
enum { RTO_TEX_SOLID = 0, RTO_TEX_CHECKER = 1, RTO_TEX_IMAGE = 2 };
/* background on a miss */
enum { RTO_BG_SKY = 0, RTO_BG_SOLID = 1 };

typedef struct {
    int32_t n_prims;
    const int32_t* prim_type;   /* n_prims */
    const double*  prim_geom;   /* n_prims x 6: sphere cx cy cz r 0 0 | rect a0 a1 b0 b1 k 0 */
    const int32_t* prim_mat;    /* n_prims */
    int32_t n_mats;
    const int32_t* mat_type;    /* n_mats */
    const double*  mat_param;   /* n_mats x 5: albedo r g b, fuzz, ir */
    const int32_t* mat_tex;     /* n_mats: texture id or -1 (then albedo is the colour) */
    int32_t n_tex;
    const int32_t* tex_type;    /* n_tex */
    const double*  tex_color;   /* n_tex x 3 (solid) */
    const int32_t* tex_a;       /* checker: even id | image: width  */
    const int32_t* tex_b;       /* checker: odd id  | image: height */
    const int64_t* tex_off;     /* image: byte offset into texels */
    const uint8_t* texels;      /* RGB8 rows, top row first */
    int32_t background;
    double bg_color[3];
    /* instance wrappers (cuda/sceneobject.h:31-109 rotate_y, plus a translation):
     * world = R_y(angle) * object + offset for primitives with prim_xform[i] >= 0 */
    int32_t n_xforms;
    const double*  xform;       /* n_xforms x 4: angle in degrees, offset x y z */
    const int32_t* prim_xform;  /* n_prims, -1 = none; may be NULL */
} rto_scene;

/* camera record: origin(3) lower_left(3) horizontal(3) vertical(3) u(3) v(3) w(3) lens_radius */
#define RTO_CAM_DOUBLES 22

void rto_philox4x32_10_kat(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t rto_draw(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment, uint32_t d);

#define RTO_DECL(S, R)                                                                           \
void rto_camera_init_##S(const double* lookfrom, const double* lookat, const double* vup,       \
                         double vfov, double aspect, double aperture, double focus_dist,          \
                         double* cam);                                                            \
void rto_get_ray_##S(const double* cam, double s, double t, uint64_t seed, uint32_t pixel,       \
                     uint32_t sample, uint32_t first_draw, double* ray, uint32_t* draws_used);   \
void rto_sample_##S(int which, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment, \
                    double* out, uint32_t* draws_used);                                          \
void rto_trace_##S(const rto_scene* sc, int n, const double* rays, double t_min, double t_max,   \
                   int32_t* id, double* t, double* p, double* nrm, int32_t* front, double* uv);  \
int rto_scatter_##S(const rto_scene* sc, int mat, const double* ray_in, const double* p,         \
                    const double* nrm, int front, const double* uv, uint64_t seed,               \
                    uint32_t pixel, uint32_t sample, uint32_t segment, double* out);             \
void rto_texture_value_##S(const rto_scene* sc, int tex, double u, double v, const double* p,    \
                           double* rgb);                                                         \
void rto_ray_color_##S(const rto_scene* sc, const double* ray, int depth, double t_min,          \
                       uint64_t seed, uint32_t pixel, uint32_t sample, double* rgb,              \
                       uint64_t* rays);                                                          \
uint64_t rto_render_##S(const rto_scene* sc, const double* cam, int W, int H, int spp,          \
                        int sample_begin, int depth, double t_min, uint64_t seed,                \
                        int i0, int i1, int j0, int j1, int istep, int jstep,                    \
                        int threads, double* sums);

RTO_DECL(f64, double)
RTO_DECL(f32, float)
#undef RTO_DECL

/* write_color (color.h:9): per pixel sum of spp samples -> three 0..255 ints */
void rto_write_color(int n, const double* sums, int spp, int32_t* out);

/* camera animation of the CUDA tree (cuda/camera.h:59-117, cuda/main.cu:220-279), in float
 * like the original.  state = look_from(3) angles(3); advanced in place by one frame. */
void rto_camera_path_step(int frame, float* state);

#ifdef __cplusplus
}
#endif
#endif
