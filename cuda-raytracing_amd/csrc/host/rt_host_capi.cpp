// rt_host_capi.cpp -- extern "C" facade (include/rt_host.h) over the host C++ API.
#include <cstring>
#include <exception>
#include <string>

#include "../../../include/rt_hip.h"
#include "../../../include/rt_host.h"
#include "Camera.h"
#include "ImageIO.hpp"
#include "OBJLoader.hpp"
#include "Scene.h"

struct RthMesh { MeshPrimitive mesh; explicit RthMesh(MeshPrimitive m) : mesh(std::move(m)) {} };
struct RthScene { Scene scene; };
struct RthCamera { Camera cam; RthCamera(int w, int h, float3x3 K, float4 D) : cam(w, h, K, D) {} };

static thread_local std::string g_err;
static float3 F3(const float* v) { return make_float3(v[0], v[1], v[2]); }
static lre LRE(const float* p) { lre l; l.x = p[0]; l.y = p[1]; l.z = p[2]; l.yaw = p[3]; l.pitch = p[4]; l.roll = p[5]; return l; }

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

RthMesh* rth_obj_load(const char* path)
{
    try {
        std::vector<TrianglePrimitive> tris;
        std::string err;
        if (!path || !OBJLoader::parse(path, tris, &err)) { g_err = path ? err : "null path"; return nullptr; }
        return new RthMesh(MeshPrimitive(std::move(tris)));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
int32_t rth_obj_parse(const char* path, int32_t lenient, float* out18, int32_t capacity)
{
    try {
        std::vector<TrianglePrimitive> tris;
        std::string err;
        if (!path || !OBJLoader::parse(path, tris, &err, lenient != 0)) { g_err = path ? err : "null path"; return -1; }
        if (out18 && (size_t)capacity >= tris.size() && !tris.empty()) memcpy(out18, tris.data(), tris.size() * 72);
        return (int32_t)tris.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int rth_scan_float(const char* token, size_t length, float* out)
{
    float f = 0.0f;
    if (!token || !out || !OBJLoader::scan_float_token(token, token + length, f)) return 0;
    *out = f;
    return 1;
}
RthMesh* rth_obj_load_for_device(const char* path)
{
    try {
        if (!path) { g_err = "null path"; return nullptr; }
        return new RthMesh(OBJLoader::load_for_device(path));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
RthMesh* rth_mesh_from_triangles_for_device(const float* tris18, int32_t n)
{
    try {
        std::vector<TrianglePrimitive> tris((size_t)(n > 0 ? n : 0));
        if (n > 0) memcpy((void*)tris.data(), tris18, (size_t)n * 72);
        return new RthMesh(MeshPrimitive::for_device_build(std::move(tris)));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
RthMesh* rth_obj_load_lenient(const char* path)
{
    try {
        std::vector<TrianglePrimitive> tris;
        std::string err;
        if (!path || !OBJLoader::parse(path, tris, &err, true)) { g_err = path ? err : "null path"; return nullptr; }
        return new RthMesh(MeshPrimitive(std::move(tris)));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
RthMesh* rth_mesh_from_triangles(const float* tris18, int32_t n)
{
    try {
        static_assert(sizeof(TrianglePrimitive) == 72, "");
        std::vector<TrianglePrimitive> tris((size_t)(n > 0 ? n : 0));
        if (n > 0) memcpy((void*)tris.data(), tris18, (size_t)n * 72);
        return new RthMesh(MeshPrimitive(std::move(tris)));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
RthMesh* rth_mesh_from_triangles_gpu(const float* tris18, int32_t n)
{
    try {
        std::vector<TrianglePrimitive> tris((size_t)(n > 0 ? n : 0));
        if (n > 0) memcpy((void*)tris.data(), tris18, (size_t)n * 72);
        return new RthMesh(MeshPrimitive(std::move(tris), true));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
RthMesh* rth_obj_load_gpu(const char* path)
{
    try {
        std::vector<TrianglePrimitive> tris;
        std::string err;
        if (!path || !OBJLoader::parse(path, tris, &err)) { g_err = path ? err : "null path"; return nullptr; }
        return new RthMesh(MeshPrimitive(std::move(tris), true));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
RthMesh* rth_mesh_single_triangle(const float* abc9)
{
    try {
        std::vector<TrianglePrimitive> tris;
        tris.push_back(TrianglePrimitive(F3(abc9), F3(abc9 + 3), F3(abc9 + 6)));
        return new RthMesh(MeshPrimitive(std::move(tris)));
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void rth_mesh_free(RthMesh* m) { delete m; }
int32_t rth_mesh_num_triangles(const RthMesh* m) { return m->mesh.num_triangles; }
int32_t rth_mesh_num_nodes(const RthMesh* m) { return (int32_t)m->mesh.bvh_top.nodes.size(); }
int32_t rth_mesh_max_level(const RthMesh* m) { return m->mesh.bvh_top.max_level(); }
void rth_mesh_get_triangles(const RthMesh* m, float* out18)
{ if (m->mesh.num_triangles) memcpy(out18, m->mesh.triangle_array().data(), (size_t)m->mesh.num_triangles * 72); }
int32_t rth_mesh_get_nodes(const RthMesh* m, float* boxes, int32_t* child, int32_t* leaf_count)
{
    const auto& N = m->mesh.bvh_top.nodes;
    int32_t total = 0;
    for (size_t i = 0; i < N.size(); i++) {
        const BVHNode& n = N[i];
        const bool leaf = n.child_index_a == -1 && n.child_index_b == -1;
        if (boxes) { boxes[6*i] = n.min.x; boxes[6*i+1] = n.min.y; boxes[6*i+2] = n.min.z; boxes[6*i+3] = n.max.x; boxes[6*i+4] = n.max.y; boxes[6*i+5] = n.max.z; }
        if (child) { child[2*i] = n.child_index_a; child[2*i+1] = n.child_index_b; }
        if (leaf_count) leaf_count[i] = leaf ? n.count : 0;
        if (leaf) total += n.count;
    }
    return total;
}
void rth_mesh_get_leaf_indices(const RthMesh* m, int32_t* out)
{
    const auto& N = m->mesh.bvh_top.nodes;
    size_t k = 0;
    for (const BVHNode& n : N)
        if (n.child_index_a == -1 && n.child_index_b == -1)
            for (int i = 0; i < n.count; i++) out[k++] = m->mesh.bvh_top.order[(size_t)n.first + i];
}
void rth_mesh_print_stats(const RthMesh* m) { m->mesh.bvh_top.print_stats(); }

RthScene* rth_scene_create(void) { try { return new RthScene; } catch (...) { return nullptr; } }
void rth_scene_free(RthScene* s) { delete s; }
int32_t rth_scene_add_material(RthScene* s, const float* albedo3, const uint8_t* tex, int32_t w, int32_t h, size_t pitch)
{
    Material m;
    m.albedo = F3(albedo3);
    if (tex && w > 0 && h > 0) m.set_texture_bgr(tex, w, h, pitch);
    s->scene.add_material(std::move(m));
    return 0;
}
int32_t rth_scene_add_material_ppm(RthScene* s, const float* albedo3, const char* ppm_path)
{
    Material m;
    m.albedo = F3(albedo3);
    if (!m.upload_texture(ppm_path)) { g_err = std::string("cannot read PPM ") + ppm_path; return RT_E_INVALID; }
    s->scene.add_material(std::move(m));
    return 0;
}
int rth_scene_set_material_params(RthScene* s, int32_t index, float roughness, float metallic, float illumination)
{
    if (index < 0 || index >= s->scene.num_materials()) return RT_E_INVALID;
    Material& m = s->scene.material(index);
    m.roughness = roughness; m.metallic = metallic; m.illumination = illumination;
    return 0;
}
int32_t rth_scene_add_mesh(RthScene* s, const RthMesh* m) { try { s->scene.add_mesh(m->mesh); return 0; } catch (...) { return RT_E_NOMEM; } }
int32_t rth_scene_add_mesh_instance(RthScene* s, int32_t mesh, int32_t material, const float* pose6, const float* scale3)
{ s->scene.add_mesh_instance(MeshInstance(mesh, material, LRE(pose6), F3(scale3))); return 0; }
int rth_scene_upload_to_device(RthScene* s)
{ try { s->scene.upload_to_device(); return s->scene.last_error; } catch (const std::exception& e) { g_err = e.what(); return RT_E_NOMEM; } }
int rth_scene_update_mesh_instance(RthScene* s, int32_t index, int32_t mesh, int32_t material, const float* pose6, const float* scale3)
{ s->scene.update_mesh_instance(index, MeshInstance(mesh, material, LRE(pose6), F3(scale3))); return s->scene.last_error; }
int rth_scene_update_mesh_instance_async(RthScene* s, int32_t index, int32_t mesh, int32_t material, const float* pose6,
                                         const float* scale3, void* stream)
{ s->scene.update_mesh_instance(index, MeshInstance(mesh, material, LRE(pose6), F3(scale3)), stream); return s->scene.last_error; }
int rth_scene_refit_mesh(RthScene* s, int32_t mesh_index, const float* tris18, int32_t n, void* stream)
{
    try {
        std::vector<TrianglePrimitive> tris((size_t)(n > 0 ? n : 0));
        if (n > 0) memcpy((void*)tris.data(), tris18, (size_t)n * 72);
        s->scene.refit_mesh(mesh_index, std::move(tris), stream);
        return s->scene.last_error;
    } catch (const std::exception& e) { g_err = e.what(); return RT_E_NOMEM; }
}
int rth_scene_rebuild_mesh(RthScene* s, int32_t mesh_index, const float* tris18, int32_t n, void* stream)
{
    try {
        std::vector<TrianglePrimitive> tris((size_t)(n > 0 ? n : 0));
        if (n > 0) memcpy((void*)tris.data(), tris18, (size_t)n * 72);
        s->scene.rebuild_mesh(mesh_index, std::move(tris), stream);
        return s->scene.last_error;
    } catch (const std::exception& e) { g_err = e.what(); return RT_E_NOMEM; }
}
int32_t rth_scene_num_mesh_instances(const RthScene* s) { return s->scene.num_mesh_instances; }
void* rth_scene_device_handle(RthScene* s) { return s->scene.d_scene; }
void rth_instance_build(const float* pose6, const float* scale3, float* out24)
{
    MeshInstance in(0, 0, LRE(pose6), F3(scale3));
    memcpy(out24, &in.pose, 96);
}

RthCamera* rth_camera_create(int32_t width, int32_t height, const float* K9, const float* D4)
{
    float3x3 K; memcpy(&K, K9, sizeof K);
    try { return new RthCamera(width, height, K, make_float4(D4[0], D4[1], D4[2], D4[3])); } catch (...) { return nullptr; }
}
void rth_camera_free(RthCamera* c) { delete c; }
void rth_camera_set_pose(RthCamera* c, const float* pose6) { c->cam.pose = LRE(pose6); }
void rth_camera_set_stream(RthCamera* c, void* stream) { c->cam.stream = stream; }
void rth_camera_set_options(RthCamera* c, int32_t spp, int32_t bounces, int32_t lighting)
{ c->cam.spp = spp; c->cam.bounces = bounces; c->cam.lighting = lighting != 0; }
int rth_camera_render_scene_ex(RthCamera* c, RthScene* s, void* d_img, size_t pitch, int32_t* d_total_pops, int synchronize)
{ c->cam.render_scene_ex(s->scene, (uchar3*)d_img, pitch, d_total_pops, synchronize != 0); return c->cam.last_error; }
uint32_t rth_xorwow(uint64_t seed, int32_t n, uint32_t* out_bits, float* out_uniform)
{
    rt::Xorwow a, b;
    rt::xorwow_init(a, seed); b = a;
    uint32_t last = 0;
    for (int i = 0; i < n; i++) { last = rt::xorwow_next(a); if (out_bits) out_bits[i] = last; if (out_uniform) out_uniform[i] = rt::xorwow_uniform(b); }
    return last;
}
int rth_camera_render_scene(RthCamera* c, RthScene* s, void* d_img, size_t pitch, int synchronize)
{ c->cam.render_scene(s->scene, (uchar3*)d_img, pitch, synchronize != 0); return c->cam.last_error; }
int rth_camera_render_scene_stripes(RthCamera* c, RthScene* s, void* d_local, size_t local_pitch, int32_t stripe_rows, int32_t rank,
                                    int32_t num_ranks, int synchronize)
{ c->cam.render_scene_stripes(s->scene, (uchar3*)d_local, local_pitch, stripe_rows, rank, num_ranks, synchronize != 0); return c->cam.last_error; }
int rth_camera_render_scene_tiled(RthCamera* c, RthScene* s, void* comm, void* d_img, size_t pitch, int32_t stripe_rows, int32_t root,
                                  int synchronize)
{ c->cam.render_scene_tiled(s->scene, (RtComm*)comm, (uchar3*)d_img, pitch, synchronize != 0, stripe_rows, root); return c->cam.last_error; }
int rth_camera_render_scene_batch(RthCamera* c, RthScene* s, const float* poses6, void* const* d_imgs, size_t pitch, int32_t count,
                                  int synchronize)
{
    if (count < 1 || count > RT_MAX_BATCH) return RT_E_INVALID;
    lre poses[RT_MAX_BATCH];
    for (int i = 0; i < count; i++) poses[i] = LRE(poses6 + 6 * i);
    c->cam.render_scene_batch(s->scene, poses, count, (uchar3* const*)d_imgs, pitch, synchronize != 0);
    return c->cam.last_error;
}
int rth_camera_render_scene_stripes_batch(RthCamera* c, RthScene* s, const float* poses6, void* const* d_locals, size_t local_pitch,
                                          int32_t count, int32_t stripe_rows, int32_t rank, int32_t num_ranks, int synchronize)
{
    if (count < 1 || count > RT_MAX_BATCH) return RT_E_INVALID;
    lre poses[RT_MAX_BATCH];
    for (int i = 0; i < count; i++) poses[i] = LRE(poses6 + 6 * i);
    c->cam.render_scene_stripes_batch(s->scene, poses, count, (uchar3* const*)d_locals, local_pitch, stripe_rows, rank, num_ranks,
                                      synchronize != 0);
    return c->cam.last_error;
}
int rth_camera_render_scene_stripes_batch_rotating(RthCamera* c, RthScene* s, const float* poses6, void* const* d_locals, size_t local_pitch,
                                                   int32_t count, int32_t stripe_rows, int32_t rank, int32_t num_ranks, int32_t first_frame,
                                                   int synchronize)
{
    if (count < 1 || count > RT_MAX_BATCH || first_frame < 0) return RT_E_INVALID;
    lre poses[RT_MAX_BATCH];
    for (int i = 0; i < count; i++) poses[i] = LRE(poses6 + 6 * i);
    c->cam.render_scene_stripes_batch(s->scene, poses, count, (uchar3* const*)d_locals, local_pitch, stripe_rows, rank, num_ranks,
                                      synchronize != 0, first_frame);
    return c->cam.last_error;
}
void rth_camera_params(const RthCamera* c, void* out)
{
    RtCameraParams p;
    p.width = c->cam.width; p.height = c->cam.height;
    memcpy(p.K_inv, &c->cam.K_inv, sizeof p.K_inv);
    p.D[0] = c->cam.D.x; p.D[1] = c->cam.D.y; p.D[2] = c->cam.D.z; p.D[3] = c->cam.D.w;
    lre inv = invert_lre(c->cam.pose);
    memcpy(p.camera_pose, &c->cam.pose, sizeof p.camera_pose);
    memcpy(p.inv_camera_pose, &inv, sizeof p.inv_camera_pose);
    memcpy(out, &p, sizeof p);
}

int rth_save_png(const char* path, const void* d_img, int32_t width, int32_t height, size_t pitch)
{ return save_png(path, (const uchar3*)d_img, width, height, pitch, nullptr); }
int rth_read_image_bgr(const char* path, uint8_t* bgr, size_t capacity, int32_t* width, int32_t* height)
{
    if (!path || !width || !height) return RT_E_INVALID;
    std::vector<uint8_t> px;
    int w = 0, h = 0;
    std::string err;
    if (!read_image_bgr(path, px, w, h, &err)) { g_err = "rth_read_image_bgr: " + err; return RT_E_INVALID; }
    *width = w; *height = h;
    if (bgr) {
        if (capacity < px.size()) return RT_E_INVALID;
        memcpy(bgr, px.data(), px.size());
    }
    return RT_OK;
}
int rth_zlib_inflate(const uint8_t* src, size_t n, uint8_t* out, size_t capacity, size_t* out_n)
{
    if (!src || !out_n) return RT_E_INVALID;
    std::vector<uint8_t> o;
    std::string err;
    // with a destination buffer the output is capped at its capacity; the size query is capped at 1 GiB
    if (!zlib_inflate(src, n, o, &err, out ? capacity : (size_t)1 << 30)) { g_err = "rth_zlib_inflate: " + err; return RT_E_INVALID; }
    *out_n = o.size();
    if (out) {
        if (capacity < o.size()) return RT_E_INVALID;
        memcpy(out, o.data(), o.size());
    }
    return RT_OK;
}
void rth_overlay_text_bgr(uint8_t* bgr, int32_t width, int32_t height, size_t pitch, const char* text, int32_t x, int32_t y,
                          int32_t scale, uint8_t b, uint8_t g, uint8_t r)
{
    overlay_text_bgr(bgr, width, height, pitch, text ? text : "", x, y, scale, b, g, r);
}
int rth_display_image(const void* d_img, int32_t width, int32_t height, size_t pitch, double fps, const char* path)
{
    MouseParams m;
    m.pose = nullptr;
    return display_image((const uchar3*)d_img, width, height, pitch, fps, m, path ? path : "out.png");
}
void rth_on_mouse(float* pose6, int32_t* state4, int32_t event, int32_t x, int32_t y)
{
    lre pose = LRE(pose6);
    MouseParams m;
    m.last_x = state4[0]; m.last_y = state4[1]; m.has_last = state4[2] != 0; m.is_down = state4[3] != 0; m.pose = &pose;
    on_mouse(event, x, y, 0, &m);
    state4[0] = m.last_x; state4[1] = m.last_y; state4[2] = m.has_last ? 1 : 0; state4[3] = m.is_down ? 1 : 0;
    pose6[0] = pose.x; pose6[1] = pose.y; pose6[2] = pose.z; pose6[3] = pose.yaw; pose6[4] = pose.pitch; pose6[5] = pose.roll;
}
int rth_on_key(float* pose6, int32_t key)
{
    lre pose = LRE(pose6);
    MouseParams m;
    m.pose = &pose;
    const bool go_on = on_key(key, m);
    pose6[0] = pose.x; pose6[1] = pose.y; pose6[2] = pose.z; pose6[3] = pose.yaw; pose6[4] = pose.pitch; pose6[5] = pose.roll;
    return go_on ? 1 : 0;
}
int rth_write_png_bgr(const char* path, const uint8_t* bgr, int32_t width, int32_t height, size_t pitch)
{ return write_png_bgr(path, bgr, width, height, pitch); }
float rth_q_rsqrt(float x) { return Q_rsqrt(x); }
float rth_atanf(float x) { return rt::atanf_fdlibm(x); }
void rth_normalize(const float* v, float* o) { float3 r = normalize(F3(v)); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void rth_invert_lre(const float* l, float* o) { lre r = invert_lre(LRE(l)); memcpy(o, &r, sizeof r); }
void rth_apply_lre(const float* l, const float* v, float* o) { float3 r = apply_lre(LRE(l), F3(v)); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void rth_euler2quat(const float* e, float* o) { float4 q = euler2quat(F3(e)); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
void rth_apply_quat(const float* q, const float* v, float* o)
{ float3 r = apply_quat(make_float4(q[0], q[1], q[2], q[3]), F3(v)); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void rth_invert_intrinsic(const float* K9, float* o9) { float3x3 K; memcpy(&K, K9, sizeof K); float3x3 r = invert_intrinsic(K); memcpy(o9, &r, sizeof r); }

}  // extern "C"
