// The reference's CPU driver (main.cc:87-129) on the GPU path: same scene-authoring calls, same
// P3 output on stdout, with the pixel/sample loop replaced by one rt_render_host call.
//
//   g++ -std=c++17 -Iinclude examples/rtiow_main.cpp -Lraytrace_amd -lrt_amd -Wl,-rpath,$PWD/raytrace_amd -o rtiow
//   ./rtiow scene.txt 1200 800 500 > image.ppm
//   ./rtiow scene.txt 480 270 1 450 > image_450_low.ppm    (frame 450 of the CUDA driver's camera path,
//                                                           cuda/main.cu:220-279,557-591)
//
// scene.txt holds one sphere per line: cx cy cz r type(0 lambertian,1 metal,2 dielectric) r g b fuzz|ir
// (the reference's own generate_random_scene depends on its compiler's argument evaluation
// order, main.cc:47,54, so the world is read, not re-rolled).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "rt_scene.hpp"

using namespace rt_scene;

static sceneobject_list load_world(const char* path) {
    sceneobject_list world;
    std::ifstream in(path);
    double cx, cy, cz, r, type, a0, a1, a2, p;
    while (in >> cx >> cy >> cz >> r >> type >> a0 >> a1 >> a2 >> p) {
        shared_ptr<material> m;
        if (type == 0) m = make_shared<lambertian>(color(a0, a1, a2));
        else if (type == 1) m = make_shared<metal>(color(a0, a1, a2), p);
        else m = make_shared<dielectric>(p);
        world.add(make_shared<sphere>(point3(cx, cy, cz), r, m));
    }
    return world;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: rtiow scene.txt [width height spp [frame]]\n"; return 2; }
    const int image_width = argc > 2 ? std::atoi(argv[2]) : 1200;
    const int image_height = argc > 3 ? std::atoi(argv[3]) : 800;
    const int samples_per_pixel = argc > 4 ? std::atoi(argv[4]) : 500;
    const int frame = argc > 5 ? std::atoi(argv[5]) : 0;
    const int ray_bounce_limit = 50;
    const float aspect_ratio = (float)image_width / (float)image_height;

    sceneobject_list world = load_world(argv[1]);
    camera cam(point3(6, 3, -13), point3(0, 0, 0), vec3(0, 1, 0), 20, aspect_ratio, 0.1, 10.0);
    for (int f = 0; f <= frame; f++) update_scene(cam, f);             // cuda/main.cu:564, once per frame up to this one

    try {
        renderer gpu(0);
        gpu.upload(world);
        gpu.set_camera(cam);
        std::vector<float> fb = gpu.render(image_width, image_height, samples_per_pixel, ray_bounce_limit);
        rt_stats st = gpu.stats();
        std::cerr << st.rays << " rays in " << st.kernel_ms << " ms\n";

        std::cout << "P3\n" << image_width << ' ' << image_height << "\n255\n";
        const float scale = 1.0 / samples_per_pixel;                     // color.h:17
        for (int j = image_height - 1; j >= 0; j--)
            for (int i = 0; i < image_width; i++) {
                const float* px = &fb[4 * ((size_t)j * image_width + i)];
                for (int c = 0; c < 3; c++) {
                    double v = std::sqrt(scale * (double)px[c]);
                    v = v < 0.0 ? 0.0 : (v > 0.999 ? 0.999 : v);
                    std::cout << static_cast<int>(256 * v) << (c < 2 ? ' ' : '\n');
                }
            }
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
