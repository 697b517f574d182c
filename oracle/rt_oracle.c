/* TEST INFRASTRUCTURE — CPU restatement of the reference's per-pixel render path
 * (see rt_oracle.h for who may load it).  Builds the double and float variants from
 * rt_oracle_impl.h. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "rt_oracle.h"
#include "rt_philox.h"

void rto_philox4x32_10_kat(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    rto_philox4x32_10(ctr, key, out);
}
uint32_t rto_draw(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment, uint32_t d) {
    return rto_draw_k24(seed, pixel, sample, segment, d);
}

#define R double
#define S(name) name##_f64
#define M_SQRT sqrt
#define M_FABS fabs
#define M_FMIN fmin
#define M_FMAX fmax
#define M_TAN tan
#define M_POW pow
#define M_SIN sin
#define M_COS cos
#define M_ACOS acos
#define M_ATAN2 atan2
#include "rt_oracle_impl.h"
#undef R
#undef S
#undef M_SQRT
#undef M_FABS
#undef M_FMIN
#undef M_FMAX
#undef M_TAN
#undef M_POW
#undef M_SIN
#undef M_COS
#undef M_ACOS
#undef M_ATAN2

#define R float
#define S(name) name##_f32
#define M_SQRT sqrtf
#define M_FABS fabsf
#define M_FMIN fminf
#define M_FMAX fmaxf
#define M_TAN tanf
#define M_POW powf
#define M_SIN sinf
#define M_COS cosf
#define M_ACOS acosf
#define M_ATAN2 atan2f
#include "rt_oracle_impl.h"

/* color.h:9-26.  `scale` is a float in the reference (color.h:17). */
static double rto_clamp(double x, double lo, double hi) {   /* utils.h:18 */
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
void rto_write_color(int n, const double* sums, int spp, int32_t* out) {
    float scale = 1.0 / spp;
    for (int i = 0; i < 3 * n; i++) {
        double c = sqrt(scale * sums[i]);
        out[i] = (int)(256 * rto_clamp(c, 0.0, 0.999));
    }
}

/* cuda/camera.h:59-117 and cuda/main.cu:206-279, in float like the original.
 * The rotate_camera_* functions rotate look_from about the world origin; their names do
 * not match the plane they rotate in (the "x" one turns in the xy-plane). */
static float deg2rad_f(float d) { const float pi = 3.1415926535897932385f; return d * pi / 180.0f; }

static void rot_x(float* s, float new_angle) {      /* cuda/camera.h:69-81 */
    float r = deg2rad_f(new_angle - s[3]);
    float nx = s[0] * cosf(r) - s[1] * sinf(r);
    float ny = s[0] * sinf(r) + s[1] * cosf(r);
    s[0] = nx; s[1] = ny; s[3] = new_angle;
}
static void rot_y(float* s, float new_angle) {      /* cuda/camera.h:87-99 */
    float r = deg2rad_f(new_angle - s[4]);
    float nx = s[0] * cosf(r) + s[2] * sinf(r);
    float nz = s[2] * cosf(r) - s[0] * sinf(r);
    s[0] = nx; s[2] = nz; s[4] = new_angle;
}
static void rot_z(float* s, float new_angle) {      /* cuda/camera.h:105-117 */
    float r = deg2rad_f(new_angle - s[5]);
    float ny = s[1] * cosf(r) - s[2] * sinf(r);
    float nz = s[1] * sinf(r) + s[2] * cosf(r);
    s[1] = ny; s[2] = nz; s[5] = new_angle;
}
static void rotate_scene(float* s, float ax, float ay, float az) {   /* cuda/main.cu:206-217 */
    if (ax != s[3]) rot_x(s, ax);
    if (ay != s[4]) rot_y(s, ay);
    if (az != s[5]) rot_z(s, az);
}
void rto_camera_path_step(int frame, float* s) {    /* cuda/main.cu:220-279 */
    if (frame == 0) return;
    if (frame < 200) rotate_scene(s, s[3], s[4] + 0.125f, s[5]);
    else if (frame < 400) rotate_scene(s, s[3], s[4] - 0.125f, s[5]);
    else if (frame < 500) {
        rotate_scene(s, s[3] + 0.125f, s[4], s[5]);
        s[0] += -0.025f; s[1] += 0.025f; s[2] += 0.0f;
    } else if (frame < 600) {
        rotate_scene(s, s[3] - 0.125f, s[4], s[5]);
        s[0] += 0.025f; s[1] += -0.025f; s[2] += 0.0f;
    }
}
