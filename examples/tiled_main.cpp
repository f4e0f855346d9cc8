// tiled_main.cpp -- one frame tiled over every GPU of the node by ONE process, written against the host C++ API and the
// C-ABI only (no Python, no torch): what the reference's main() (kernel.cu:141-302) becomes when the frame is cut into
// 16-row stripes, one replica of the scene per device, and gathered over RCCL (include/rt_hip.h: rt_comm_init_all,
// rt_render_tiled_all).  With one visible GPU it degenerates to a plain render, which is how the tests run it.
//
//   g++ -std=c++17 -O2 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude
//       -Icuda-raytracing_amd/csrc/host examples/tiled_main.cpp -Lcuda-raytracing_amd -lrt_host -lrt_hip
//       -Wl,-rpath,$PWD/cuda-raytracing_amd -o tiled
//   ./tiled mesh.obj out.png [width height [spp bounces lighting]]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <vector>

#include "rt_hip.h"
#include "Camera.h"
#include "ImageIO.hpp"
#include "OBJLoader.hpp"
#include "Scene.h"

int main(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "usage: tiled mesh.obj out.png [width height [spp bounces lighting]]" << std::endl; return 2; }
    const int width = argc > 4 ? atoi(argv[3]) : 1920, height = argc > 4 ? atoi(argv[4]) : 1080;
    int n = 0;
    if (rt_device_count(&n) || n < 1) { std::cerr << "no GPU" << std::endl; return 1; }

    const double s = width / 1920.0;                            // the reference's calibration (kernel.cu:158-164), scaled to the width
    float4 D = make_float4(0.016233999489849514, -0.013875757716177956, 0.03264329940126211, -0.019561619947134234);
    float3x3 K = {(float)(862.097835972576 * s), 0.0, (float)(998.1702383680802 * s), 0, (float)(862.1368447300727 * s), (float)(569.6759403225842 * s), 0, 0, 1};
    Camera camera(width, height, K, D);
    camera.pose.y = -1.6; camera.pose.z = 0.2;
    if (argc > 7) { camera.spp = atoi(argv[5]); camera.bounces = atoi(argv[6]); camera.lighting = atoi(argv[7]) != 0; }

    // one communicator and one replica of the scene per device
    std::vector<RtComm*> comms((size_t)n, nullptr);
    int rc = rt_comm_init_all(nullptr, n, comms.data());
    if (rc) { std::cerr << "rt_comm_init_all: " << rt_error_string(rc) << " " << rt_comm_last_error() << std::endl; return 1; }
    MeshPrimitive mesh = OBJLoader::load(argv[1]);
    std::vector<std::unique_ptr<Scene>> scenes;
    std::vector<RtScene*> handles;
    for (int d = 0; d < n; d++) {
        rt_set_device(d);
        scenes.emplace_back(new Scene);
        Material m = Material();
        m.albedo = make_float3(0.9, 0.5, 0.2); m.metallic = 0.4; m.roughness = 0.05;
        scenes[d]->add_material(m);
        scenes[d]->add_mesh(mesh);
        scenes[d]->add_mesh_instance(MeshInstance(0, 0));
        scenes[d]->upload_to_device();
        if (scenes[d]->last_error) { std::cerr << "upload on device " << d << ": " << rt_error_string(scenes[d]->last_error) << std::endl; return 1; }
        handles.push_back(scenes[d]->d_scene);
    }
    rt_set_device(0);
    uchar3* d_img = nullptr;
    size_t pitch = 0;
    rt_malloc_pitch((void**)&d_img, &pitch, width * sizeof(uchar3), height);

    RtCameraParams p;
    p.width = width; p.height = height;
    lre inv = invert_lre(camera.pose);
    memcpy(p.K_inv, &camera.K_inv, sizeof p.K_inv);
    p.D[0] = D.x; p.D[1] = D.y; p.D[2] = D.z; p.D[3] = D.w;
    memcpy(p.camera_pose, &camera.pose, sizeof p.camera_pose);
    memcpy(p.inv_camera_pose, &inv, sizeof p.inv_camera_pose);
    RtRenderOptions o = {camera.spp, camera.bounces, camera.lighting ? 1 : 0};

    double best = 1e30;
    for (int it = 0; it < 3; it++) {
        auto t0 = std::chrono::steady_clock::now();
        rc = rt_render_tiled_all(handles.data(), comms.data(), n, &p, &o, (uint8_t*)d_img, pitch, 16, 0, nullptr, 1);
        if (rc) { std::cerr << "rt_render_tiled_all: " << rt_error_string(rc) << " " << rt_comm_last_error() << std::endl; return 1; }
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    std::cout << n << " GPU(s), " << width << "x" << height << " x " << camera.spp << " spp: " << best * 1e3 << " ms/frame, "
              << (double)width * height * camera.spp / best / 1e6 << " Mrays/s" << std::endl;
    rc = save_png(argv[2], d_img, width, height, pitch);
    std::cout << (rc ? "could not write " : "wrote ") << argv[2] << std::endl;
    rt_free(d_img);
    for (RtComm* c : comms) rt_comm_destroy(c);
    return rc ? 1 : 0;
}
