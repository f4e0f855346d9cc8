// Camera.h -- the kernel launcher of the reference (Camera.h:9-31, Camera.cu:6-41): stores K,
// K_inv, D and the pose, and renders a Scene into a caller-owned pitched device image.
#pragma once
#include <cstddef>
#include "Scene.h"
#include "transforms.hpp"

using namespace transforms;

struct RtComm;

// Replacement for display_image()'s `cv::imwrite("out.png", ...)` (kernel.cu:30-43) without OpenCV: copies a pitched BGR
// device image to the host and writes it as an 8-bit RGB PNG (stored, i.e. uncompressed, deflate blocks).
// Returns 0 or an rt_hip.h error code (RT_E_INVALID if the file cannot be written).
int save_png(const char* path, const uchar3* d_img, int width, int height, size_t pitch, void* stream = nullptr);
// same from host memory
int write_png_bgr(const char* path, const unsigned char* bgr, int width, int height, size_t pitch);

class Camera {
public:
    Camera(int width, int height, float3x3 K, float4 D);

    lre pose;
    int width;
    int height;
    float3x3 K;
    float3x3 K_inv;
    float4 D;
    void* stream = nullptr;                        // hipStream_t; the reference always uses the default stream
    // extension (rt_render_ex): with the defaults render_scene() is the reference's one-ray, illumination-1.0 frame
    int spp = 1;                                   // samples per pixel (sample 0 = the reference ray)
    int bounces = 0;                               // specular bounces (Material::metallic / roughness)
    bool lighting = false;                         // sun + shadow pass of raycast.cu:249-287
    int last_error = 0;

    // asynchronous on `stream` unless synchronize (Camera.cu:38-39)
    void render_scene(Scene& scene, uchar3* img_ptr, size_t pitch, bool synchronize = false);
    // extension frame with the optional per-pixel pops plane (device int32 [height][width]) for parity tests
    void render_scene_ex(Scene& scene, uchar3* img_ptr, size_t pitch, int* d_total_pops, bool synchronize = false);
    // `count` frames (<= RT_MAX_BATCH) along a camera path in one launch: frame i uses poses[i] (K, D, size from
    // this camera) and goes to img_ptrs[i]; see rt_render_batch in rt_hip.h
    void render_scene_batch(Scene& scene, const lre* poses, int count, uchar3* const* img_ptrs, size_t pitch, bool synchronize = false);
    // rotate_first >= 0: the stripe owner rotates with the frame index -- frame i renders the stripes of owner
    // (rank + rotate_first + i) % num_ranks, so that every rank's share of a group of frames is the same (rt_hip.h)
    void render_scene_stripes_batch(Scene& scene, const lre* poses, int count, uchar3* const* local_ptrs, size_t local_pitch,
                                    int stripe_rows, int rank, int num_ranks, bool synchronize = false, int rotate_first = -1);
    // One frame tiled over the GPUs of `comm` (rt_comm_init_rank / rt_comm_init_all, rt_hip.h): every rank calls this with
    // its own replica of the scene; the frame arrives in img_ptr on rank `root` (img_ptr may be null elsewhere).
    // Honours spp / bounces / lighting like render_scene.  The multi-GPU form of Camera.cu:18-41.
    void render_scene_tiled(Scene& scene, RtComm* comm, uchar3* img_ptr, size_t pitch, bool synchronize = false, int stripe_rows = 16,
                            int root = 0);
    // this rank's stripes of the frame into a tight local buffer (multi-GPU tiling, rt_hip.h)
    void render_scene_stripes(Scene& scene, uchar3* local_ptr, size_t local_pitch, int stripe_rows, int rank, int num_ranks,
                              bool synchronize = false);
};
