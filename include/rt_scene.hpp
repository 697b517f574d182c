// rt_scene.hpp — header-only C++ facade over the C-ABI of rt.h.
//
// It keeps the reference's scene-authoring vocabulary (class names, constructor argument order
// and meaning) so that host code written against the reference keeps compiling:
//
//   vec3 / point3 / color                         vec3.h:13-73
//   sphere(center, radius, material)              sphere.h:13, cuda/sphere.h:16
//   xy_rect / xz_rect / yz_rect(a0,a1,b0,b1,k,m)  cuda/rect.h:19,66,112
//   sceneobject_list::add / clear                 sceneobject_list.h:16-21
//   bvh_node(list)                                cuda/bvh.h:18
//   rotate_y(object, angle)                       cuda/sceneobject.h:31-47 (+ translate, box)
//   lambertian / metal / dielectric               material.h:19,40,56
//   lambertian_bg / diffuse_light                 cuda/material.h:37-38,107-109
//   rgb_color / checker_texture / image_texture   cuda/texture.h:22,39-41,61
//   load_texture(filename)                        cuda/main.cu:385-436
//   camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist)   camera.h:19-20
//   camera(look_from, look_at, angles, vup, vfov, aspect, aperture, focus_dist), translate_camera,
//   rotate_camera_{x,y,z}, compute_camera_scene   cuda/camera.h:29-117
//   rotate_scene / update_scene(camera, frame)    cuda/main.cu:206-279
//
// Objects only record what they were built from; `flat_scene` turns a sceneobject_list into the
// arrays rt_upload_scene takes, and `renderer` owns an rt_context.  Nothing here computes pixels.
#ifndef RT_SCENE_HPP
#define RT_SCENE_HPP

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt.h"

namespace rt_scene {

using std::make_shared;
using std::shared_ptr;

struct vec3 {
    double e[3];
    vec3() : e{0, 0, 0} {}
    vec3(double e0, double e1, double e2) : e{e0, e1, e2} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double operator[](int i) const { return e[i]; }
};
using point3 = vec3;
using color = vec3;

// ---- textures ------------------------------------------------------------------------
struct custom_texture { virtual ~custom_texture() = default; };
struct rgb_color : custom_texture {
    color color_value;
    explicit rgb_color(color c) : color_value(c) {}
    rgb_color(double r, double g, double b) : color_value(r, g, b) {}
};
struct checker_texture : custom_texture {
    shared_ptr<custom_texture> even, odd;
    checker_texture(shared_ptr<custom_texture> e, shared_ptr<custom_texture> o) : even(e), odd(o) {}
    checker_texture(color c1, color c2) : even(make_shared<rgb_color>(c1)), odd(make_shared<rgb_color>(c2)) {}
};
// texels: RGB8 rows, top row first (what stbi_load(...,3) returns, cuda/main.cu:393)
struct image_texture : custom_texture {
    uint32_t texture_width, texture_height;
    std::vector<uint8_t> texels;
    image_texture(uint32_t w, uint32_t h, const uint8_t* rgb8) : texture_width(w), texture_height(h), texels(rgb8, rgb8 + (size_t)w * h * 3) {}
};

// load_texture of cuda/main.cu:385-436: a JPEG file as an image texture (decoded by the library
// to the texels the reference's stb_image produces).
inline shared_ptr<image_texture> load_texture(const char* filename) {
    int32_t w = 0, h = 0;
    uint8_t* rgb = nullptr;
    if (rt_load_jpeg_file(filename, &w, &h, &rgb) != RT_OK)
        throw std::runtime_error(std::string("load_texture: ") + rt_image_last_error());
    auto t = make_shared<image_texture>((uint32_t)w, (uint32_t)h, rgb);
    rt_free_image(rgb);
    return t;
}

// ---- materials --------------------------------------------------------------------------
struct material {
    int type = RT_LAMBERTIAN;
    color albedo{0, 0, 0};
    double fuzz = 0, ir = 0;
    shared_ptr<custom_texture> tex;
    virtual ~material() = default;
};
struct lambertian : material {
    explicit lambertian(const color& a) { type = RT_LAMBERTIAN; albedo = a; }
    explicit lambertian(shared_ptr<custom_texture> a) { type = RT_LAMBERTIAN; tex = a; }
};
struct lambertian_bg : material {
    explicit lambertian_bg(const color& a) { type = RT_LAMBERTIAN_BG; albedo = a; }
    explicit lambertian_bg(shared_ptr<custom_texture> a) { type = RT_LAMBERTIAN_BG; tex = a; }
};
struct metal : material {
    metal(const color& a, double f) { type = RT_METAL; albedo = a; fuzz = f < 1 ? f : 1; }   // material.h:40
};
struct dielectric : material {
    explicit dielectric(double index_of_refraction) { type = RT_DIELECTRIC; albedo = color(1, 1, 1); ir = index_of_refraction; }
};
struct diffuse_light : material {
    explicit diffuse_light(color c) { type = RT_DIFFUSE_LIGHT; albedo = c; }
    explicit diffuse_light(shared_ptr<custom_texture> a) { type = RT_DIFFUSE_LIGHT; tex = a; }
};

// ---- objects ------------------------------------------------------------------------------
struct sceneobject { virtual ~sceneobject() = default; };
struct sphere : sceneobject {
    point3 center; double radius; shared_ptr<material> mat_ptr;
    sphere(point3 cen, double r, shared_ptr<material> m) : center(cen), radius(r), mat_ptr(m) {}
};
struct rect_base : sceneobject {
    int kind; double a0, a1, b0, b1, k; shared_ptr<material> mat;
    rect_base(int kind_, double a0_, double a1_, double b0_, double b1_, double k_, shared_ptr<material> m)
        : kind(kind_), a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_), mat(m) {}
};
struct xy_rect : rect_base { xy_rect(double x0, double x1, double y0, double y1, double k, shared_ptr<material> m) : rect_base(RT_XY_RECT, x0, x1, y0, y1, k, m) {} };
struct xz_rect : rect_base { xz_rect(double x0, double x1, double z0, double z1, double k, shared_ptr<material> m) : rect_base(RT_XZ_RECT, x0, x1, z0, z1, k, m) {} };
struct yz_rect : rect_base { yz_rect(double y0, double y1, double z0, double z1, double k, shared_ptr<material> m) : rect_base(RT_YZ_RECT, y0, y1, z0, z1, k, m) {} };

struct sceneobject_list : sceneobject {
    std::vector<shared_ptr<sceneobject>> objects;
    sceneobject_list() {}
    explicit sceneobject_list(shared_ptr<sceneobject> object) { add(object); }
    void clear() { objects.clear(); }
    void add(shared_ptr<sceneobject> object) { objects.push_back(object); }
};
// Grouping marker (cuda/bvh.h:18): the library builds one BVH over the whole flattened scene.
struct bvh_node : sceneobject_list {
    explicit bvh_node(const sceneobject_list& l) { objects = l.objects; }
};

// cuda/sceneobject.h:31-47: `p` turned by `angle` degrees about the world's +y axis.
struct rotate_y : sceneobject {
    shared_ptr<sceneobject> ptr; double angle;
    rotate_y(shared_ptr<sceneobject> p, double angle_) : ptr(p), angle(angle_) {}
};
// `p` moved by `displacement` (the companion every use of rotate_y needs: an object turns about
// the world's y axis, not its own).
struct translate : sceneobject {
    shared_ptr<sceneobject> ptr; vec3 offset;
    translate(shared_ptr<sceneobject> p, vec3 displacement) : ptr(p), offset(displacement) {}
};
// The six rects of an axis-aligned box in the order of cuda/main.cu:130-137: per axis pair the
// face at p1 first (material `m`), then the face at p0 (`m_low`, default `m`).
inline shared_ptr<sceneobject_list> box(point3 p0, point3 p1, shared_ptr<material> m, shared_ptr<material> m_low = nullptr) {
    if (!m_low) m_low = m;
    auto l = make_shared<sceneobject_list>();
    l->add(make_shared<xy_rect>(p0.x(), p1.x(), p0.y(), p1.y(), p1.z(), m)); l->add(make_shared<xy_rect>(p0.x(), p1.x(), p0.y(), p1.y(), p0.z(), m_low));
    l->add(make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p1.y(), m)); l->add(make_shared<xz_rect>(p0.x(), p1.x(), p0.z(), p1.z(), p0.y(), m_low));
    l->add(make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p1.x(), m)); l->add(make_shared<yz_rect>(p0.y(), p1.y(), p0.z(), p1.z(), p0.x(), m_low));
    return l;
}

// camera.h:19-38, and the CUDA tree's camera (cuda/camera.h:9-126): the constructor with `angles`,
// the mutators and compute_camera_scene.  The mutators work in single precision like the
// original (its vec3 holds floats): look_from and angles always hold float values.  The view
// basis itself is derived by rt_set_camera from what desc() returns.
struct camera {
    point3 look_from, look_at;
    vec3 angles, vup;
    double vfov, aspect_ratio, aperture, focus_dist;

    camera(point3 lookfrom, point3 lookat, vec3 vup_, double vfov_, double aspect_ratio_, double aperture_, double focus_dist_)
        : look_from(lookfrom), look_at(lookat), angles(0, 0, 0), vup(vup_), vfov(vfov_), aspect_ratio(aspect_ratio_),
          aperture(aperture_), focus_dist(focus_dist_) {}
    // cuda/camera.h:29-36
    camera(point3 look_from_, point3 look_at_, vec3 angles_, vec3 vup_, float vfov_, float aspect_ratio_, float aperture_, float focus_dist_)
        : look_from(look_from_), look_at(look_at_), angles(angles_), vup(vup_), vfov(vfov_), aspect_ratio(aspect_ratio_),
          aperture(aperture_), focus_dist(focus_dist_) {}

    rt_camera_desc desc() const {
        rt_camera_desc d{};
        for (int a = 0; a < 3; a++) { d.lookfrom[a] = look_from[a]; d.lookat[a] = look_at[a]; d.vup[a] = vup[a]; }
        d.vfov = vfov; d.aspect = aspect_ratio; d.aperture = aperture; d.focus_dist = focus_dist;
        return d;
    }
    // cuda/camera.h:42-57.  Nothing is cached on the host (rt_set_camera derives u, v, w and the
    // viewport from the descriptor), so this only hands out the descriptor of the current pose.
    rt_camera_desc compute_camera_scene() const { return desc(); }

    void translate_camera(float x, float y, float z) {               // cuda/camera.h:59-63
        look_from = point3((float)look_from[0] + x, (float)look_from[1] + y, (float)look_from[2] + z);
    }
    void rotate_camera_x(float new_x_angle) { turn(0, new_x_angle, 0, 1); }    // cuda/camera.h:69-81: the xy-plane
    void rotate_camera_y(float new_y_angle) {                                   // cuda/camera.h:87-99: the xz-plane
        float c, s;
        delta(1, new_y_angle, c, s);
        const float x = (float)look_from[0], z = (float)look_from[2];
        const float nx = mul(x, c) + mul(z, s), nz = mul(z, c) - mul(x, s);
        look_from = point3(nx, look_from[1], nz);
    }
    void rotate_camera_z(float new_z_angle) { turn(2, new_z_angle, 1, 2); }    // cuda/camera.h:105-117: the yz-plane

private:
    // a single-precision product, rounded once, which no compiler setting can fuse into an addition
    static float mul(float a, float b) { return (float)((double)a * (double)b); }
    // cosine and sine of (new angle - current angle), cuda/utils.h:17 in float; records the new angle
    void delta(int which, float new_angle, float& c, float& s) {
        const float d_angle = new_angle - (float)angles[which];
        const float d_rads = mul(d_angle, 3.1415926535897932385f) / 180.0f;
        c = (float)std::cos((double)d_rads); s = (float)std::sin((double)d_rads);
        angles.e[which] = new_angle;
    }
    void turn(int which, float new_angle, int i, int j) {
        float c, s;
        delta(which, new_angle, c, s);
        const float a = (float)look_from[i], b = (float)look_from[j];
        const float na = mul(a, c) - mul(b, s), nb = mul(a, s) + mul(b, c);
        look_from.e[i] = na; look_from.e[j] = nb;
    }
};

// cuda/main.cu:206-217
inline void rotate_scene(camera& cam, float new_x_angle, float new_y_angle, float new_z_angle) {
    if (new_x_angle != (float)cam.angles.x()) cam.rotate_camera_x(new_x_angle);
    if (new_y_angle != (float)cam.angles.y()) cam.rotate_camera_y(new_y_angle);
    if (new_z_angle != (float)cam.angles.z()) cam.rotate_camera_z(new_z_angle);
}
// cuda/main.cu:220-279: the camera of frame `frame`, given the camera of the frame before it.
inline void update_scene(camera& cam, int frame) {
    const float ax = (float)cam.angles.x(), ay = (float)cam.angles.y(), az = (float)cam.angles.z();
    if (frame == 0) {
    } else if (frame < 200) {
        rotate_scene(cam, ax, ay + 0.125f, az);
    } else if (frame < 400) {
        rotate_scene(cam, ax, ay - 0.125f, az);
    } else if (frame < 500) {
        rotate_scene(cam, ax + 0.125f, ay, az);
        cam.translate_camera(-0.025f, 0.025f, 0.0f);
    } else if (frame < 600) {
        rotate_scene(cam, ax - 0.125f, ay, az);
        cam.translate_camera(0.025f, -0.025f, 0.0f);
    }
    (void)cam.compute_camera_scene();
}

// ---- flattening ------------------------------------------------------------------------------
struct flat_scene {
    std::vector<int32_t> prim_type, prim_mat, prim_xform, mat_type, mat_tex, tex_type, tex_a, tex_b;
    std::vector<double> prim_geom, mat_param, tex_color, xform;     // xform: [n][4] degrees about +y, offset
    // instance transform in force while walking: world = R_y(angle) * object + off
    struct xf { bool any; double angle; double off[3]; };
    static xf no_xf() { xf x; x.any = false; x.angle = 0; x.off[0] = x.off[1] = x.off[2] = 0; return x; }
    std::vector<int64_t> tex_off;
    std::vector<uint8_t> texels;
    std::map<const void*, int> mat_ids, tex_ids;

    int tex_id(const shared_ptr<custom_texture>& t) {
        auto it = tex_ids.find(t.get());
        if (it != tex_ids.end()) return it->second;
        int type, a = 0, b = 0; int64_t off = 0; color c;
        if (auto s = dynamic_cast<rgb_color*>(t.get())) { type = RT_TEX_SOLID; c = s->color_value; }
        else if (auto k = dynamic_cast<checker_texture*>(t.get())) { type = RT_TEX_CHECKER; a = tex_id(k->even); b = tex_id(k->odd); }
        else if (auto im = dynamic_cast<image_texture*>(t.get())) {
            type = RT_TEX_IMAGE; a = (int)im->texture_width; b = (int)im->texture_height; off = (int64_t)texels.size();
            texels.insert(texels.end(), im->texels.begin(), im->texels.end());
        } else throw std::invalid_argument("unknown texture class");
        int id = (int)tex_type.size();
        tex_type.push_back(type); tex_a.push_back(a); tex_b.push_back(b); tex_off.push_back(off);
        for (int q = 0; q < 3; q++) tex_color.push_back(c[q]);
        tex_ids[t.get()] = id;
        return id;
    }
    int mat_id(const shared_ptr<material>& m) {
        auto it = mat_ids.find(m.get());
        if (it != mat_ids.end()) return it->second;
        int tex = m->tex ? tex_id(m->tex) : -1;
        int id = (int)mat_type.size();
        mat_type.push_back(m->type); mat_tex.push_back(tex);
        for (int q = 0; q < 3; q++) mat_param.push_back(m->albedo[q]);
        mat_param.push_back(m->fuzz); mat_param.push_back(m->ir);
        mat_ids[m.get()] = id;
        return id;
    }
    // Wrappers compose into one rotation about y and one offset.  `outer` is the transform
    // above the wrapper; the wrapper's own motion acts first, so world = outer(wrapper(object)).
    static xf under(const xf& outer, double angle, const double* t) {
        const double r = outer.angle * 3.1415926535897932385 / 180.0, c = std::cos(r), s = std::sin(r);
        xf x = no_xf(); x.any = true; x.angle = outer.angle + angle;
        x.off[0] = c * t[0] + s * t[2] + outer.off[0];               // cuda/sceneobject.h:98-99
        x.off[1] = t[1] + outer.off[1];
        x.off[2] = -s * t[0] + c * t[2] + outer.off[2];
        return x;
    }
    int xform_id(const xf& x) {
        if (!x.any) return -1;
        for (size_t i = 0; i < xform.size() / 4; i++)
            if (xform[4 * i] == x.angle && xform[4 * i + 1] == x.off[0] && xform[4 * i + 2] == x.off[1] && xform[4 * i + 3] == x.off[2]) return (int)i;
        xform.push_back(x.angle); xform.insert(xform.end(), x.off, x.off + 3);
        return (int)(xform.size() / 4) - 1;
    }
    void add(const shared_ptr<sceneobject>& o) { add(o, no_xf()); }
    void add(const shared_ptr<sceneobject>& o, const xf& x) {
        if (auto l = dynamic_cast<sceneobject_list*>(o.get())) { for (auto& c : l->objects) add(c, x); return; }
        if (auto r = dynamic_cast<rotate_y*>(o.get())) { const double zero[3] = {0, 0, 0}; add(r->ptr, under(x, r->angle, zero)); return; }
        if (auto t = dynamic_cast<translate*>(o.get())) { add(t->ptr, under(x, 0.0, t->offset.e)); return; }
        prim_xform.push_back(xform_id(x));
        double g[6] = {0, 0, 0, 0, 0, 0};
        if (auto s = dynamic_cast<sphere*>(o.get())) {
            prim_type.push_back(RT_SPHERE); prim_mat.push_back(mat_id(s->mat_ptr));
            g[0] = s->center[0]; g[1] = s->center[1]; g[2] = s->center[2]; g[3] = s->radius;
        } else if (auto r = dynamic_cast<rect_base*>(o.get())) {
            prim_type.push_back(r->kind); prim_mat.push_back(mat_id(r->mat));
            g[0] = r->a0; g[1] = r->a1; g[2] = r->b0; g[3] = r->b1; g[4] = r->k;
        } else throw std::invalid_argument("unknown scene object class");
        prim_geom.insert(prim_geom.end(), g, g + 6);
    }
    explicit flat_scene(const sceneobject_list& world) { for (auto& o : world.objects) add(o); }

    rt_scene_desc desc(int background = RT_BG_SKY, color bg = color(0, 0, 0), int accel = RT_ACCEL_AUTO) const {
        rt_scene_desc d{};
        d.struct_size = sizeof(d);
        d.n_prims = (int32_t)prim_type.size(); d.prim_type = prim_type.data(); d.prim_geom = prim_geom.data(); d.prim_mat = prim_mat.data();
        d.n_mats = (int32_t)mat_type.size(); d.mat_type = mat_type.data(); d.mat_param = mat_param.data(); d.mat_tex = mat_tex.data();
        d.n_tex = (int32_t)tex_type.size(); d.tex_type = tex_type.data(); d.tex_color = tex_color.data();
        d.tex_a = tex_a.data(); d.tex_b = tex_b.data(); d.tex_off = tex_off.data();
        d.texels = texels.data(); d.texel_bytes = texels.size();
        d.background = background; d.bg_color[0] = (float)bg[0]; d.bg_color[1] = (float)bg[1]; d.bg_color[2] = (float)bg[2];
        d.accel = accel; d.wide_radius = 0;
        d.n_xforms = (int32_t)(xform.size() / 4); d.xform = xform.data(); d.prim_xform = prim_xform.data();
        return d;
    }
};

// ---- renderer ------------------------------------------------------------------------------------
class renderer {
public:
    explicit renderer(int device = 0) { check(rt_create(device, &ctx_), "rt_create"); }
    ~renderer() { rt_destroy(ctx_); }
    renderer(const renderer&) = delete;
    renderer& operator=(const renderer&) = delete;

    void upload(const sceneobject_list& world, int background = RT_BG_SKY, color bg = color(0, 0, 0), int accel = RT_ACCEL_AUTO) {
        flat_scene f(world);
        rt_scene_desc d = f.desc(background, bg, accel);
        check(rt_upload_scene(ctx_, &d), "rt_upload_scene");
    }
    void set_camera(const camera& cam) { const rt_camera_desc d = cam.desc(); check(rt_set_camera(ctx_, &d), "rt_set_camera"); }

    // Whole frame into host memory: float4 per pixel, row 0 = bottom row, rgb = radiance sums.
    std::vector<float> render(int width, int height, int samples_per_pixel, int ray_bounce_limit, uint64_t seed = 1) {
        rt_render_params p{};
        p.struct_size = sizeof(p);
        p.width = width; p.height = height; p.spp = samples_per_pixel; p.max_depth = ray_bounce_limit; p.seed = seed;
        std::vector<float> out((size_t)width * height * 4);
        check(rt_render_host(ctx_, &p, out.data()), "rt_render_host");
        return out;
    }
    rt_stats stats() { rt_stats s{}; s.struct_size = sizeof(s); check(rt_get_stats(ctx_, &s), "rt_get_stats"); return s; }
    rt_context* context() { return ctx_; }

private:
    void check(int rc, const char* what) {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + (ctx_ ? rt_last_error(ctx_) : "no gfx950 device"));
    }
    rt_context* ctx_ = nullptr;
};

}  // namespace rt_scene
#endif
