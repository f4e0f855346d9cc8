// demo_main.cpp -- the reference application's flow (kernel.cu:141-302: camera, materials, OBJ meshes, instances,
// upload, a render loop that issues two frames per synchronise, prints FPS and ends every pass with display_image ->
// out.png with the FPS overlay) written against this
// project's host API.  It shows what a user of the reference keeps (Scene / Camera / OBJLoader / MeshInstance /
// Material calls) and what changes (rt_hip.h plumbing instead of cudaMallocPitch / cudaDeviceSynchronize / OpenCV).
//
//   g++ -std=c++17 -O2 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude
//       -Icuda-raytracing_amd/csrc/host examples/demo_main.cpp -Lcuda-raytracing_amd -lrt_host -lrt_hip
//       -Wl,-rpath,$PWD/cuda-raytracing_amd -o demo
//   ./demo mesh.obj [out.png] [iterations] [texture.png|.jpg|.ppm]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "rt_hip.h"
#include "Camera.h"
#include "ImageIO.hpp"
#include "OBJLoader.hpp"
#include "Scene.h"

int main(int argc, char** argv)
{
    if (argc < 2) { std::cerr << "usage: demo mesh.obj [out.png] [iterations] [texture.png|.jpg|.ppm]" << std::endl; return 2; }
    const char* out_png = argc > 2 ? argv[2] : "out.png";
    const int iterations = argc > 3 ? atoi(argv[3]) : 100;

    int width = 1920, height = 1080;
    float4 D = make_float4(0.016233999489849514, -0.013875757716177956, 0.03264329940126211, -0.019561619947134234);
    float3x3 K = {862.097835972576, 0.0, 998.1702383680802,
                  0, 862.1368447300727, 569.6759403225842,
                  0, 0, 1};
    Camera camera = Camera(width, height, K, D);
    camera.pose.x = 0;
    camera.pose.y = -1.6;
    camera.pose.z = 0.2;

    Scene scene;
    Material plain = Material();
    plain.albedo = make_float3(0.9, 0.5, 0.2);
    plain.roughness = 0.01;
    scene.add_material(plain);
    Material textured = Material();
    textured.albedo = make_float3(1.0, 1.0, 1.0);
    const bool have_texture = argc > 4 && textured.upload_texture(argv[4]);
    scene.add_material(textured);

    MeshPrimitive mesh = OBJLoader::load(argv[1]);
    mesh.bvh_top.print_stats();
    scene.add_mesh(mesh);

    MeshInstance big = MeshInstance(0, have_texture ? 1 : 0);
    scene.add_mesh_instance(big);
    MeshInstance small_one = MeshInstance(0, 0);
    small_one.pose.x = -0.6; small_one.pose.y = 1.48; small_one.pose.z = 0.73;
    small_one.scale = make_float3(0.4, 0.4, 0.4);
    scene.add_mesh_instance(small_one);

    scene.upload_to_device();
    if (scene.last_error) { std::cerr << "upload failed: " << rt_error_string(scene.last_error) << std::endl; return 1; }

    uchar3 *d_img, *d_img2;
    size_t pitch, pitch2;
    rt_malloc_pitch((void**)&d_img, &pitch, width * sizeof(uchar3), height);
    rt_malloc_pitch((void**)&d_img2, &pitch2, width * sizeof(uchar3), height);

    MouseParams mouse_state;                                    // kernel.cu:258-259 (no window system here: never fed)
    mouse_state.pose = &camera.pose;

    double fps = 0.0;
    int rc = 0;
    for (int l = 0; l < iterations; l++) {
        auto t0 = std::chrono::steady_clock::now();
        camera.render_scene(scene, d_img, pitch);               // two renders per synchronise, as kernel.cu:277-279
        camera.render_scene(scene, d_img2, pitch2);
        rt_device_synchronize();
        if (camera.last_error) { std::cerr << "render failed: " << rt_error_string(camera.last_error) << std::endl; return 1; }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fps = 1.0 / dt;
        std::cout << "FPS: " << fps << "\n";
        rc = display_image(d_img, width, height, pitch, fps, mouse_state, out_png, camera.stream);    // kernel.cu:298
        if (rc) break;
    }
    std::cout << "FPS: " << fps << " (" << 2.0 * width * height * fps / 1e6 << " Mrays/s)" << std::endl;
    std::cout << (rc ? "could not write " : "wrote ") << out_png << std::endl;
    rt_free(d_img);
    rt_free(d_img2);
    return rc ? 1 : 0;
}
