// Prints the camera of every frame of the reference's animation (cuda/main.cu:220-279) as the C++
// facade computes it: "frame look_from(3) angles(3)" with %.9g, one line per frame.
#include <cstdio>

#include "rt_scene.hpp"

using namespace rt_scene;

int main() {
    // the reference driver's camera (cuda/main.cu:182-187), eight-argument form
    camera cam(point3(450, 278, 200), point3(0, 278, 555), vec3(0, 0, 0), vec3(0, 1, 0), 40.0f, 16.0f / 9.0f, 0.0f, 10.0f);
    for (int frame = 0; frame < 600; frame++) {
        update_scene(cam, frame);
        const rt_camera_desc d = cam.compute_camera_scene();
        std::printf("%d %.9g %.9g %.9g %.9g %.9g %.9g\n", frame, d.lookfrom[0], d.lookfrom[1], d.lookfrom[2],
                    cam.angles.x(), cam.angles.y(), cam.angles.z());
    }
    // the mutators on their own, and the seven-argument constructor of camera.h:19-20
    camera c2(point3(6, 3, -13), point3(0, 0, 0), vec3(0, 1, 0), 20, 1.5, 0.1, 10.0);
    c2.rotate_camera_z(33.0f);
    c2.rotate_camera_x(-12.5f);
    c2.translate_camera(0.1f, 0.2f, 0.3f);
    c2.rotate_camera_y(400.0f);
    std::printf("c2 %.9g %.9g %.9g %.9g %.9g %.9g\n", c2.look_from[0], c2.look_from[1], c2.look_from[2],
                c2.angles.x(), c2.angles.y(), c2.angles.z());
    return 0;
}
