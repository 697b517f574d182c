// Builds the two-turned-boxes room through the C++ facade and prints the flattened arrays, one
// primitive per line: type geom[5] mat xform | then the transforms.  Compared by
// tests/test_host.py with what the Python facade produces for the same scene.
#include <cstdio>

#include "rt_scene.hpp"

using namespace rt_scene;

int main() {
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(7, 7, 7));
    sceneobject_list w;
    w.add(make_shared<yz_rect>(0, 555, 0, 555, 555, green));
    w.add(make_shared<yz_rect>(0, 555, 0, 555, 0, red));
    w.add(make_shared<xz_rect>(75, 270, 76, 280, 554, light));
    w.add(make_shared<xz_rect>(0, 555, 0, 555, 0, white));
    w.add(make_shared<xz_rect>(0, 555, 0, 555, 555, white));
    w.add(make_shared<xy_rect>(0, 555, 0, 555, 555, white));
    w.add(make_shared<translate>(make_shared<rotate_y>(box(point3(0, 0, 0), point3(165, 330, 165), white), 15), vec3(265, 0, 295)));
    w.add(make_shared<translate>(make_shared<rotate_y>(box(point3(0, 0, 0), point3(165, 165, 165), white), -18), vec3(130, 0, 65)));
    w.add(make_shared<translate>(make_shared<rotate_y>(make_shared<sphere>(point3(82.5, 215, 82.5), 50.0, make_shared<dielectric>(1.5)), -18), vec3(130, 0, 65)));
    // a wrapper around wrappers: the whole room turned and moved once more
    sceneobject_list outer;
    outer.add(make_shared<translate>(make_shared<rotate_y>(make_shared<sceneobject_list>(w), 33), vec3(120, -40, 77)));
    for (const sceneobject_list* world : {&w, &outer}) {
        flat_scene f(*world);
        rt_scene_desc d = f.desc(RT_BG_SOLID);
        std::printf("scene %d %d\n", d.n_prims, d.n_xforms);
        for (int i = 0; i < d.n_prims; i++) {
            std::printf("%d", d.prim_type[i]);
            for (int a = 0; a < 5; a++) std::printf(" %.17g", d.prim_geom[6 * i + a]);
            std::printf(" %d %d\n", d.prim_mat[i], d.prim_xform[i]);
        }
        for (int i = 0; i < d.n_xforms; i++)
            std::printf("%.17g %.17g %.17g %.17g\n", d.xform[4 * i], d.xform[4 * i + 1], d.xform[4 * i + 2], d.xform[4 * i + 3]);
    }
    return 0;
}
