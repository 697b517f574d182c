// TEST INFRASTRUCTURE — builds oracle/_ref/libref_main.so, the compiled
// reference checker.  Nothing shipped links or loads it.
//
// This translation unit compiles the reference CPU renderer's own sources from
// where they lie (the include path points at the reference checkout; no
// reference text is copied here) and exports a C interface around its
// unmodified functions:
//
//   generate_random_scene   main.cc:37     ray_color          main.cc:10
//   camera / get_ray        camera.h:19,40 sphere::hit        sphere.h:18
//   sceneobject_list::hit   sceneobject_list.h:26
//   lambertian/metal/dielectric::scatter   material.h:21,42,58
//   random_in_unit_sphere/_disk/random_unit_vector  vec3.h:130,150,139
//   reflect / refract       vec3.h:159,163 write_color        color.h:9
//
// Two preprocessor hooks make that possible without touching the sources:
//   * `main` is renamed so main.cc can be included as a library;
//   * `rand()` (used only by utils.h:25) is routed through rt_ref_rand(), which
//     either forwards to glibc rand() (scene authoring, CPU-baseline timing) or
//     serves the keyed Philox stream of rt_philox.h, so the unmodified
//     integrator consumes exactly the stream the GPU kernels draw from.
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "rt_philox.h"

extern "C" int rt_ref_rand();
#define rand() rt_ref_rand()
#define main rt_reference_main
#include "main.cc"
#undef main
#undef rand

namespace {

enum { MODE_LIBC = 0, MODE_KEYED = 1, MODE_LIST = 2 };

struct shim_state {
    int mode = MODE_LIBC;
    uint64_t seed = 0;
    uint32_t pixel = 0, sample = 0, segment = 0;
    rto_stream st{};
    const uint32_t* list = nullptr;
    size_t list_n = 0, list_pos = 0;
    uint64_t n_draws = 0;
    uint64_t n_rays = 0;
} g;

// A list whose closest-hit query opens the next RNG segment and counts rays.
// ray_color (main.cc:18) calls world.hit() through the virtual table, so the
// integrator itself stays untouched.
struct counting_list : public sceneobject_list {
    bool hit(const ray& r, double t_min, double t_max, hit_record& rec) const override {
        g.segment++;
        rto_stream_begin(&g.st, g.seed, g.pixel, g.sample, g.segment);
        g.n_rays++;
        return sceneobject_list::hit(r, t_min, t_max, rec);
    }
};

counting_list g_world;
std::map<const material*, int> g_mat_to_obj;
camera* g_cam = nullptr;

void index_world() {
    g_mat_to_obj.clear();
    for (size_t i = 0; i < g_world.objects.size(); i++) {
        auto* s = dynamic_cast<sphere*>(g_world.objects[i].get());
        if (s) g_mat_to_obj[s->mat_ptr.get()] = (int)i;
    }
}

}  // namespace

extern "C" {

int rt_ref_rand() {
    g.n_draws++;
    switch (g.mode) {
    case MODE_KEYED: {
        uint32_t k24 = rto_stream_next(&g.st);
        return (int)(k24 << 7);  // rand()/(RAND_MAX+1.0) == k24 / 2^24
    }
    case MODE_LIST: {
        uint32_t k24 = (g.list_pos < g.list_n) ? g.list[g.list_pos] : 0u;
        g.list_pos++;
        return (int)(k24 << 7);
    }
    default:
        return (rand)();
    }
}

void ref_set_mode_libc(void) { g.mode = MODE_LIBC; }
void ref_set_key(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t segment) {
    g.mode = MODE_KEYED; g.seed = seed; g.pixel = pixel; g.sample = sample;
    g.segment = segment;
    rto_stream_begin(&g.st, seed, pixel, sample, segment);
}
void ref_set_draw_list(const uint32_t* k24, size_t n) {
    g.mode = MODE_LIST; g.list = k24; g.list_n = n; g.list_pos = 0;
}
size_t ref_draw_list_used(void) { return g.list_pos; }
uint64_t ref_draw_count(void) { return g.n_draws; }
uint64_t ref_ray_count(void) { return g.n_rays; }
void ref_reset_counters(void) { g.n_draws = 0; g.n_rays = 0; }

// ---- scene ---------------------------------------------------------------

// The default scene of main.cc:97 under glibc's default seed (no srand call in
// the reference; srand(1) restores that state).
int ref_build_default_scene(void) {
    g.mode = MODE_LIBC;
    srand(1);
    sceneobject_list w = generate_random_scene();
    g_world.objects = w.objects;
    index_world();
    return (int)g_world.objects.size();
}

int ref_scene_count(void) { return (int)g_world.objects.size(); }

// Per object: cx cy cz r type a0 a1 a2 p   (type 0 lambertian, 1 metal, 2 dielectric;
// a = albedo, p = fuzz for metal / ir for dielectric)
int ref_scene_dump(double* out) {
    for (size_t i = 0; i < g_world.objects.size(); i++) {
        auto* s = dynamic_cast<sphere*>(g_world.objects[i].get());
        if (!s) return -1;
        double* o = out + 9 * i;
        o[0] = s->center.x(); o[1] = s->center.y(); o[2] = s->center.z(); o[3] = s->radius;
        const material* m = s->mat_ptr.get();
        if (auto* l = dynamic_cast<const lambertian*>(m)) {
            o[4] = 0; o[5] = l->albedo.x(); o[6] = l->albedo.y(); o[7] = l->albedo.z(); o[8] = 0;
        } else if (auto* me = dynamic_cast<const metal*>(m)) {
            o[4] = 1; o[5] = me->albedo.x(); o[6] = me->albedo.y(); o[7] = me->albedo.z(); o[8] = me->fuzz;
        } else if (auto* d = dynamic_cast<const dielectric*>(m)) {
            o[4] = 2; o[5] = 1; o[6] = 1; o[7] = 1; o[8] = d->ir;
        } else {
            return -2;
        }
    }
    return (int)g_world.objects.size();
}

// Rebuild the world from the same 9-double records through the reference's
// own constructors (sphere.h:13, material.h:19,40,56).
int ref_scene_load(int n, const double* in) {
    g_world.clear();
    for (int i = 0; i < n; i++) {
        const double* o = in + 9 * i;
        shared_ptr<material> m;
        int type = (int)o[4];
        if (type == 0) m = make_shared<lambertian>(color(o[5], o[6], o[7]));
        else if (type == 1) m = make_shared<metal>(color(o[5], o[6], o[7]), o[8]);
        else if (type == 2) m = make_shared<dielectric>(o[8]);
        else return -1;
        g_world.add(make_shared<sphere>(point3(o[0], o[1], o[2]), o[3], m));
    }
    index_world();
    return n;
}

// ---- camera --------------------------------------------------------------

void ref_set_camera(const double* lookfrom, const double* lookat, const double* vup,
                    double vfov, double aspect, double aperture, double focus_dist) {
    delete g_cam;
    g_cam = new camera(point3(lookfrom[0], lookfrom[1], lookfrom[2]),
                       point3(lookat[0], lookat[1], lookat[2]),
                       vec3(vup[0], vup[1], vup[2]), vfov, aspect, aperture, focus_dist);
}

// origin(3) lower_left_corner(3) horizontal(3) vertical(3) u(3) v(3) w(3) lens_radius
void ref_camera_dump(double* out) {
    const vec3* f[7] = { &g_cam->origin, &g_cam->lower_left_corner, &g_cam->horizontal,
                         &g_cam->vertical, &g_cam->u, &g_cam->v, &g_cam->w };
    for (int i = 0; i < 7; i++)
        for (int c = 0; c < 3; c++) out[3 * i + c] = (*f[i])[c];
    out[21] = g_cam->lens_radius;
}

// get_ray(s,t) with the currently selected draw source; out = origin(3) dir(3)
void ref_get_ray(double s, double t, double* out) {
    ray r = g_cam->get_ray(s, t);
    for (int c = 0; c < 3; c++) { out[c] = r.origin()[c]; out[3 + c] = r.direction()[c]; }
}

// ---- closest hit ------------------------------------------------------------

// rays: n x (ox oy oz dx dy dz).  id = list position of the hit object, -1 on miss.
void ref_trace(int n, const double* rays, double t_min, double t_max,
               int32_t* id, double* t, double* p, double* nrm, int32_t* front) {
    for (int i = 0; i < n; i++) {
        const double* q = rays + 6 * i;
        ray r(point3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]));
        hit_record rec;
        bool h = g_world.sceneobject_list::hit(r, t_min, t_max, rec);
        if (!h) {
            id[i] = -1; t[i] = 0; front[i] = 0;
            for (int c = 0; c < 3; c++) { p[3 * i + c] = 0; nrm[3 * i + c] = 0; }
            continue;
        }
        id[i] = g_mat_to_obj.at(rec.mat_ptr.get());
        t[i] = rec.t; front[i] = rec.front_face ? 1 : 0;
        for (int c = 0; c < 3; c++) { p[3 * i + c] = rec.p[c]; nrm[3 * i + c] = rec.normal[c]; }
    }
}

// One object's sphere::hit in isolation (sphere.h:18).
int ref_sphere_hit(int obj, const double* q, double t_min, double t_max,
                   double* t, double* p, double* nrm, int32_t* front) {
    ray r(point3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]));
    hit_record rec;
    if (!g_world.objects[obj]->hit(r, t_min, t_max, rec)) return 0;
    *t = rec.t; *front = rec.front_face ? 1 : 0;
    for (int c = 0; c < 3; c++) { p[c] = rec.p[c]; nrm[c] = rec.normal[c]; }
    return 1;
}

// ---- materials ----------------------------------------------------------------

// scatter() of object `obj`'s material for an incoming ray and a hit record
// (p, normal, front_face), using the currently selected draw source.
// out = attenuation(3) origin(3) dir(3); returns the bool of scatter().
int ref_scatter(int obj, const double* ray_in, const double* p, const double* nrm,
                int front, double* out) {
    auto* s = dynamic_cast<sphere*>(g_world.objects[obj].get());
    ray r(point3(ray_in[0], ray_in[1], ray_in[2]), vec3(ray_in[3], ray_in[4], ray_in[5]));
    hit_record rec;
    rec.p = point3(p[0], p[1], p[2]);
    rec.normal = vec3(nrm[0], nrm[1], nrm[2]);
    rec.front_face = front != 0;
    rec.t = 0;
    rec.mat_ptr = s->mat_ptr;
    color att; ray sc;
    bool ok = s->mat_ptr->scatter(r, rec, att, sc);
    for (int c = 0; c < 3; c++) {
        out[c] = att[c]; out[3 + c] = sc.origin()[c]; out[6 + c] = sc.direction()[c];
    }
    return ok ? 1 : 0;
}

void ref_random_in_unit_sphere(double* out) { vec3 v = random_in_unit_sphere(); for (int c = 0; c < 3; c++) out[c] = v[c]; }
void ref_random_unit_vector(double* out)    { vec3 v = random_unit_vector();    for (int c = 0; c < 3; c++) out[c] = v[c]; }
void ref_random_in_unit_disk(double* out)   { vec3 v = random_in_unit_disk();   for (int c = 0; c < 3; c++) out[c] = v[c]; }
void ref_reflect(const double* v, const double* n, double* out) {
    vec3 r = reflect(vec3(v[0], v[1], v[2]), vec3(n[0], n[1], n[2]));
    for (int c = 0; c < 3; c++) out[c] = r[c];
}
void ref_refract(const double* uv, const double* n, double eta, double* out) {
    vec3 r = refract(vec3(uv[0], uv[1], uv[2]), vec3(n[0], n[1], n[2]), eta);
    for (int c = 0; c < 3; c++) out[c] = r[c];
}

// ---- integrator ------------------------------------------------------------------

// ray_color for one explicit ray under key (seed,pixel,sample); segment counting
// starts at 0 so the first closest-hit query opens segment 1.
void ref_ray_color(const double* q, int depth, uint64_t seed, uint32_t pixel,
                   uint32_t sample, double* out) {
    ref_set_key(seed, pixel, sample, 0);
    ray r(point3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]));
    color c = ray_color(r, g_world, depth);
    for (int k = 0; k < 3; k++) out[k] = c[k];
}

// The pixel/sample loop of main.cc:111-126 over the window [i0,i1) x [j0,j1),
// keyed stream.  Pixel index = j*W + i with j counted from the BOTTOM row
// (main.cc:118 uses the same j).  sums: H x W x 3 doubles, row j at j*W.
// Only the window is written.  Returns rays traced.
uint64_t ref_render_keyed(int W, int H, int spp, int sample_begin, int depth, uint64_t seed,
                          int i0, int i1, int j0, int j1, int istep, int jstep, double* sums) {
    uint64_t rays0 = g.n_rays;
    for (int j = j1 - 1; j >= j0; j--) {
        if ((j - j0) % jstep) continue;
        for (int i = i0; i < i1; i++) {
            if ((i - i0) % istep) continue;
            color pixel_color(0, 0, 0);
            uint32_t pix = (uint32_t)(j * W + i);
            for (int s = sample_begin; s < sample_begin + spp; s++) {
                ref_set_key(seed, pix, (uint32_t)s, 0);
                double u = (i + random_double()) / (W - 1);
                double v = (j + random_double()) / (H - 1);
                ray r = g_cam->get_ray(u, v);
                pixel_color += ray_color(r, g_world, depth);
            }
            for (int c = 0; c < 3; c++) sums[3 * ((size_t)j * W + i) + c] = pixel_color[c];
        }
    }
    return g.n_rays - rays0;
}

// Same loop on glibc rand(), as the reference binary runs it (used for the CPU
// baseline timing; results depend on the global rand() state).
uint64_t ref_render_libc(int W, int H, int spp, int depth,
                         int i0, int i1, int j0, int j1, int istep, int jstep, double* sums) {
    g.mode = MODE_LIBC;
    uint64_t rays0 = g.n_rays;
    for (int j = j1 - 1; j >= j0; j--) {
        if ((j - j0) % jstep) continue;
        for (int i = i0; i < i1; i++) {
            if ((i - i0) % istep) continue;
            color pixel_color(0, 0, 0);
            for (int s = 0; s < spp; s++) {
                double u = (i + random_double()) / (W - 1);
                double v = (j + random_double()) / (H - 1);
                ray r = g_cam->get_ray(u, v);
                pixel_color += ray_color(r, g_world, depth);
            }
            if (sums) for (int c = 0; c < 3; c++) sums[3 * ((size_t)j * W + i) + c] = pixel_color[c];
        }
    }
    return g.n_rays - rays0;
}

// write_color (color.h:9) for n pixels; out = n x 3 ints as the text would hold.
int ref_write_color(int n, const double* sums, int spp, int32_t* out) {
    for (int i = 0; i < n; i++) {
        std::ostringstream os;
        write_color(os, color(sums[3 * i], sums[3 * i + 1], sums[3 * i + 2]), spp);
        std::istringstream is(os.str());
        int a, b, c;
        if (!(is >> a >> b >> c)) return -1;
        out[3 * i] = a; out[3 * i + 1] = b; out[3 * i + 2] = c;
    }
    return n;
}

}  // extern "C"
