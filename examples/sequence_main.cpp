// The CUDA driver's frame loop (cuda/main.cu:557-591) on the GPU path, from C++ like the reference:
//
//     for frame: update_scene (main.cu:220-279, :564) -> render -> write_color -> output/ppm_images/image_<frame>_low.ppm (:569)
//
// Frames alternate between two contexts on two HIP streams: frame k+1 is launched before frame k is
// read back, so it starts in the tail of frame k on the device and renders while the host formats
// frame k's file (P3 text takes longer to write than a 1080p frame takes to render).  Each frame is
// quantised on the device (rt_quantize = color.h:9-26) and comes back as 3 bytes a pixel.
//
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/sequence_main.cpp
//       -Lraytrace_amd -lrt_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/raytrace_amd -o sequence
//   ./sequence scene.txt out_dir 1920 1080 256 0 240 [p3|p6] [recon]
//
// recon: every frame goes through the low-to-high sample-count reconstruction stage before it is quantised
// (rt_render_guides + rt_reconstruct: what the reference's README.md:7 names as ongoing work for the frame
// pairs of cuda/main.cu:557-576) and is written as image_<frame>_recon.ppm.
//
// scene.txt: one sphere per line, as examples/rtiow_main.cpp reads it.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <vector>

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

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    std::cerr << #call << ": " << hipGetErrorString(e_) << "\n"; return 1; } } while (0)
#define RT_CHECK(ctx, call) do { int rc_ = (call); if (rc_ != RT_OK) { \
    std::cerr << #call << ": " << rt_last_error(ctx) << "\n"; return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 8) { std::cerr << "usage: sequence scene.txt out_dir width height spp first_frame n_frames [p3|p6] [recon]\n"; return 2; }
    const std::string out_dir = argv[2];
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), spp = std::atoi(argv[5]);
    const int first = std::atoi(argv[6]), n_frames = std::atoi(argv[7]);
    const bool text = argc < 9 || std::string(argv[8]) != "p6";
    const bool recon = argc >= 10 && std::string(argv[9]) == "recon";
    const int ray_bounce_limit = 25;                       // cuda/main.cu:452
    const uint64_t seed = 1;
    if (W < 1 || H < 1 || spp < 1 || first < 0 || n_frames < 1) { std::cerr << "bad sizes\n"; return 2; }

    sceneobject_list world = load_world(argv[1]);
    camera cam(point3(6, 3, -13), point3(0, 0, 0), vec3(0, 1, 0), 20, (float)W / (float)H, 0.1, 10.0);
    mkdir(out_dir.c_str(), 0777);
    mkdir((out_dir + "/ppm_images").c_str(), 0777);

    // two contexts with the same scene, each on its own stream with its own buffers
    HIP_OK(hipSetDevice(0));
    rt_context* ctx[2] = {nullptr, nullptr};
    hipStream_t stream[2];
    float* fb[2];
    float* guides[2] = {nullptr, nullptr};                  // recon: two planes of float4 per pixel
    uint8_t* rgb_dev[2];
    std::vector<uint8_t> rgb_host[2];
    flat_scene flat(world);
    rt_scene_desc desc = flat.desc(RT_BG_SKY, color(0, 0, 0), RT_ACCEL_AUTO);
    for (int s = 0; s < 2; s++) {
        if (rt_create(0, &ctx[s]) != RT_OK) { std::cerr << "rt_create: no gfx950 device\n"; return 1; }
        HIP_OK(hipStreamCreate(&stream[s]));
        RT_CHECK(ctx[s], rt_set_stream(ctx[s], stream[s]));
        RT_CHECK(ctx[s], rt_upload_scene(ctx[s], &desc));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&fb[s]), (size_t)W * H * 4 * sizeof(float)));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&rgb_dev[s]), (size_t)W * H * 3));
        if (recon) HIP_OK(hipMalloc(reinterpret_cast<void**>(&guides[s]), (size_t)W * H * 8 * sizeof(float)));
        rgb_host[s].resize((size_t)W * H * 3);
    }

    uint64_t rays = 0;
    double kernel_ms = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    int pending_frame = -1, pending_slot = 0, launched = 0;
    auto finish = [&](int frame, int s) -> int {
        HIP_OK(hipStreamSynchronize(stream[s]));
        rt_stats st{};
        st.struct_size = sizeof(st);
        RT_CHECK(ctx[s], rt_get_stats(ctx[s], &st));
        rays += st.rays; kernel_ms += st.kernel_ms;
        const std::string path = out_dir + "/ppm_images/image_" + std::to_string(frame) + (recon ? "_recon.ppm" : "_low.ppm");
        // output_image (cuda/main.cu:358-383) on bytes that are already write_color's
        if (rt_write_ppm(path.c_str(), rgb_host[s].data(), W, H, text ? 0 : 1) != RT_OK) { std::cerr << rt_image_last_error() << "\n"; return 1; }
        return 0;
    };
    for (int frame = 0; frame < first + n_frames; frame++) {
        update_scene(cam, frame);                          // every frame moves the camera, rendered or not
        if (frame < first) continue;
        const int s = launched++ % 2;
        const rt_camera_desc cd = cam.desc();
        RT_CHECK(ctx[s], rt_set_camera(ctx[s], &cd));
        rt_render_params p{};
        p.struct_size = sizeof(p);
        p.width = W; p.height = H; p.spp = spp; p.max_depth = ray_bounce_limit; p.seed = seed + (uint64_t)frame;
        RT_CHECK(ctx[s], rt_render(ctx[s], &p, fb[s]));
        if (recon) {
            RT_CHECK(ctx[s], rt_render_guides(ctx[s], W, H, guides[s]));
            RT_CHECK(ctx[s], rt_reconstruct(ctx[s], fb[s], guides[s], W, H, nullptr, fb[s]));      // in place, default parameters
        }
        RT_CHECK(ctx[s], rt_quantize(ctx[s], fb[s], (int64_t)W * H, spp, rgb_dev[s]));
        HIP_OK(hipMemcpyAsync(rgb_host[s].data(), rgb_dev[s], rgb_host[s].size(), hipMemcpyDeviceToHost, stream[s]));
        if (pending_frame >= 0 && finish(pending_frame, pending_slot)) return 1;
        pending_frame = frame; pending_slot = s;
    }
    if (pending_frame >= 0 && finish(pending_frame, pending_slot)) return 1;
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << n_frames << " frames, " << rays << " rays, " << kernel_ms << " ms of kernels, " << wall_ms
              << " ms from the first launch to the last file\n";
    for (int s = 0; s < 2; s++) {
        (void)hipFree(fb[s]); (void)hipFree(rgb_dev[s]); (void)hipFree(guides[s]);
        rt_destroy(ctx[s]);
        (void)hipStreamDestroy(stream[s]);
    }
    return 0;
}
