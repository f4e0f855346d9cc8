/*
 * rt_host.h -- C facade over the host C++ API (Scene / Camera / OBJLoader / MeshPrimitive /
 * MeshInstance / Material in cuda-raytracing_amd/csrc/host/), for hosts that cannot include
 * C++ headers (the Python tests and bench.py bind it with ctypes).  Each function names the
 * reference call it stands for.  Returns 0 / a handle on success; errors are negative RT_E_*
 * or positive hipError_t codes as in rt_hip.h.  No exceptions cross this boundary.
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthMesh RthMesh;       /* MeshPrimitive */
typedef struct RthScene RthScene;     /* Scene */
typedef struct RthCamera RthCamera;   /* Camera */

/* OBJLoader::load(fp) (OBJLoader.hpp:15); NULL on failure, message via rth_last_error() */
RthMesh *rth_obj_load(const char *path);
/* OBJLoader::load_lenient: additionally accepts `v//vn` tokens and negative (relative) indices */
RthMesh *rth_obj_load_lenient(const char *path);
/* OBJLoader::load_for_device / MeshPrimitive::for_device_build: no host tree, the GPU builds it at Scene::upload_to_device */
RthMesh *rth_obj_load_for_device(const char *path);
RthMesh *rth_mesh_from_triangles_for_device(const float *tris18, int32_t n);
/* OBJLoader::parse alone (no BVH): the number of triangles, or -1 (rth_last_error()); when out18 is non-NULL and holds
 * `capacity` >= that many 18-float triangles (TrianglePrimitive layout) they are copied there */
int32_t rth_obj_parse(const char *path, int32_t lenient, float *out18, int32_t capacity);
/* the parser's float scanner on token[0..length): 1 and *out = the value std::stof gives, 0 = no conversion */
int rth_scan_float(const char *token, size_t length, float *out);
/* MeshPrimitive(std::vector<TrianglePrimitive>) (MeshPrimitive.h:31); tris18 = n x {v0 v1 v2 normal uv0 uv1 uv2} */
RthMesh *rth_mesh_from_triangles(const float *tris18, int32_t n);
/* the same meshes with the BVH built on the GPU (rt_bvh_build): identical tree, NULL if no device */
RthMesh *rth_mesh_from_triangles_gpu(const float *tris18, int32_t n);
RthMesh *rth_obj_load_gpu(const char *path);
/* TrianglePrimitive(a, b, c) (TrianglePrimitive.hpp:15): one triangle, normal from the winding */
RthMesh *rth_mesh_single_triangle(const float *abc9);
void rth_mesh_free(RthMesh *m);
int32_t rth_mesh_num_triangles(const RthMesh *m);
int32_t rth_mesh_num_nodes(const RthMesh *m);
int32_t rth_mesh_max_level(const RthMesh *m);
void rth_mesh_get_triangles(const RthMesh *m, float *out18);
/* boxes [n][6], child [n][2], leaf_count [n] (0 for interior); returns the total of leaf_count */
int32_t rth_mesh_get_nodes(const RthMesh *m, float *boxes, int32_t *child, int32_t *leaf_count);
void rth_mesh_get_leaf_indices(const RthMesh *m, int32_t *out);
/* BVHTree::print_stats (BVHTree.hpp:117) to stdout */
void rth_mesh_print_stats(const RthMesh *m);

RthScene *rth_scene_create(void);                                                  /* Scene() */
void rth_scene_free(RthScene *s);
/* Scene::add_material (Scene.h:19); texture = BGR bytes or NULL */
int32_t rth_scene_add_material(RthScene *s, const float *albedo3, const uint8_t *texture_bgr, int32_t w, int32_t h, size_t pitch);
/* Material::upload_texture(path) (Material.hpp:29) then add_material; binary PPM only */
int32_t rth_scene_add_material_ppm(RthScene *s, const float *albedo3, const char *ppm_path);
/* Material::roughness / metallic / illumination of material `index` (set before upload_to_device) */
int rth_scene_set_material_params(RthScene *s, int32_t index, float roughness, float metallic, float illumination);
/* Scene::add_mesh (copies the mesh, as the by-value reference call does) */
int32_t rth_scene_add_mesh(RthScene *s, const RthMesh *m);
/* Scene::add_mesh_instance(MeshInstance(mesh, material, pose, scale)) */
int32_t rth_scene_add_mesh_instance(RthScene *s, int32_t mesh, int32_t material, const float *pose6, const float *scale3);
int rth_scene_upload_to_device(RthScene *s);                                       /* Scene::upload_to_device */
int rth_scene_update_mesh_instance(RthScene *s, int32_t index, int32_t mesh, int32_t material, const float *pose6, const float *scale3);
/* ordered on `stream` instead of synchronising (rt_scene_update_instance_async) */
int rth_scene_update_mesh_instance_async(RthScene *s, int32_t index, int32_t mesh, int32_t material, const float *pose6,
                                         const float *scale3, void *stream);
/* Scene::refit_mesh: the mesh's triangles moved (tris18 = n TrianglePrimitives of 18 floats, same count and order as when
 * the mesh was built): new triangle records and refitted BVH bounds on host and device, no rebuild (rt_scene_refit_mesh) */
int rth_scene_refit_mesh(RthScene *s, int32_t mesh_index, const float *tris18, int32_t n, void *stream);
/* Scene::rebuild_mesh: new triangles for a mesh (at most as many as at upload), a new tree built on the GPU in place */
int rth_scene_rebuild_mesh(RthScene *s, int32_t mesh_index, const float *tris18, int32_t n, void *stream);
int32_t rth_scene_num_mesh_instances(const RthScene *s);
/* the RtScene* behind the Scene (for rt_render_debug etc.), NULL before upload */
void *rth_scene_device_handle(RthScene *s);
/* MeshInstance::build_inv (MeshInstance.hpp:39): out26 = pose6 inv_pose6 rotation3 inv_rotation3 scale3 inv_scale3 (minus ids) */
void rth_instance_build(const float *pose6, const float *scale3, float *out24);

RthCamera *rth_camera_create(int32_t width, int32_t height, const float *K9, const float *D4);   /* Camera(w, h, K, D) */
void rth_camera_free(RthCamera *c);
void rth_camera_set_pose(RthCamera *c, const float *pose6);                        /* camera.pose = ... */
void rth_camera_set_stream(RthCamera *c, void *stream);
/* extension options (Camera::spp / bounces / lighting, see rt_render_ex); defaults 1, 0, 0 = the reference frame */
void rth_camera_set_options(RthCamera *c, int32_t spp, int32_t bounces, int32_t lighting);
int rth_camera_render_scene_ex(RthCamera *c, RthScene *s, void *d_img, size_t pitch, int32_t *d_total_pops, int synchronize);
/* the XORWOW stream the extension kernel draws from (curand_init(seed,0,0) seeding), for known-answer tests */
uint32_t rth_xorwow(uint64_t seed, int32_t n, uint32_t *out_bits, float *out_uniform);
/* Camera::render_scene(scene, img_ptr, pitch, synchronize) (Camera.h:25) */
int rth_camera_render_scene(RthCamera *c, RthScene *s, void *d_img, size_t pitch, int synchronize);
int rth_camera_render_scene_stripes(RthCamera *c, RthScene *s, void *d_local, size_t local_pitch,
                                    int32_t stripe_rows, int32_t rank, int32_t num_ranks, int synchronize);
/* Camera::render_scene_batch: `count` frames along a camera path (poses6 = count x lre) in one launch (rt_render_batch) */
/* Camera::render_scene_tiled: one frame over the GPUs of an RtComm (rt_hip.h), frame on rank `root` */
int rth_camera_render_scene_tiled(RthCamera *c, RthScene *s, void *comm, void *d_img, size_t pitch, int32_t stripe_rows,
                                  int32_t root, int synchronize);
int rth_camera_render_scene_batch(RthCamera *c, RthScene *s, const float *poses6, void *const *d_imgs, size_t pitch,
                                  int32_t count, int synchronize);
int rth_camera_render_scene_stripes_batch(RthCamera *c, RthScene *s, const float *poses6, void *const *d_locals,
                                          size_t local_pitch, int32_t count, int32_t stripe_rows, int32_t rank,
                                          int32_t num_ranks, int synchronize);
/* the same with the stripe owner rotating over the frames: frame i renders the stripes of owner (rank + first_frame + i) % num_ranks
 * (rt_render_stripes_batch_rotating) */
int rth_camera_render_scene_stripes_batch_rotating(RthCamera *c, RthScene *s, const float *poses6, void *const *d_locals,
                                                   size_t local_pitch, int32_t count, int32_t stripe_rows, int32_t rank,
                                                   int32_t num_ranks, int32_t first_frame, int synchronize);
/* the RtCameraParams (rt_hip.h) the camera would launch with: 1 + 1 + 9 + 4 + 6 + 6 words */
void rth_camera_params(const RthCamera *c, void *out_RtCameraParams);

/* display_image()'s `cv::imwrite("out.png", ...)` (kernel.cu:30-43): device BGR image -> RGB PNG file; and from host bytes */
int rth_save_png(const char *path, const void *d_img, int32_t width, int32_t height, size_t pitch);
int rth_write_png_bgr(const char *path, const uint8_t *bgr, int32_t width, int32_t height, size_t pitch);

/* ---- image files and the display step without OpenCV (csrc/host/ImageIO.hpp) ----
 * rth_read_image_bgr: PNG / baseline JPEG / binary PPM by signature (the decoders behind Material::upload_texture, which
 * the reference does with cv::imread, Material.hpp:29-43).  bgr may be NULL to query the size only; otherwise it
 * receives width*height*3 tight B,G,R bytes if `capacity` allows (RT_E_INVALID if not, or if the file is refused). */
int rth_read_image_bgr(const char *path, uint8_t *bgr, size_t capacity, int32_t *width, int32_t *height);
int rth_zlib_inflate(const uint8_t *src, size_t n, uint8_t *out, size_t capacity, size_t *out_n);
/* text with its bottom-left corner at (x, y), built-in 5x7 font scaled by `scale` (the cv::putText of kernel.cu:41) */
void rth_overlay_text_bgr(uint8_t *bgr, int32_t width, int32_t height, size_t pitch, const char *text, int32_t x, int32_t y,
                          int32_t scale, uint8_t b, uint8_t g, uint8_t r);
/* display_image (kernel.cu:30-43): download, "FPS: ..." overlay, PNG file */
int rth_display_image(const void *d_img, int32_t width, int32_t height, size_t pitch, double fps, const char *path);
/* on_mouse (kernel.cu:112-139) / the key handling of kernel.cu:51-103 on a pose (6 floats) and the handler's state
 * {last_x, last_y, has_last, is_down}; rth_on_key returns 0 for 'q' (quit), 1 otherwise */
void rth_on_mouse(float *pose6, int32_t *state4, int32_t event, int32_t x, int32_t y);
int rth_on_key(float *pose6, int32_t key);

/* host math with the reference's names, for parity tests (utils.hpp / transforms.hpp) */
float rth_q_rsqrt(float x);
float rth_atanf(float x);                       /* the restated atanf the kernels use */
void rth_normalize(const float *v3, float *out3);
void rth_invert_lre(const float *l6, float *out6);
void rth_apply_lre(const float *l6, const float *v3, float *out3);
void rth_euler2quat(const float *e3, float *out4);
void rth_apply_quat(const float *q4, const float *v3, float *out3);
void rth_invert_intrinsic(const float *K9, float *out9);

const char *rth_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
