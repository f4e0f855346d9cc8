// ref_probe.cpp -- TEST INFRASTRUCTURE.  A thin extern "C" driver (ours) around the
// reference's OWN sources, compiled where they lie under /root/reference by oracle/Makefile
// into oracle/_ref/libref_probe.so.  Nothing from the reference is copied: the headers are
// #included by path and MeshPrimitive.cpp is compiled from its own location.
//
// What builds here with the image's real CUDA headers (triton/backends/nvidia/include):
//   utils.hpp, transforms.hpp, Ray.hpp, TrianglePrimitive.hpp, BVHTree.hpp, MeshInstance.hpp,
//   MeshPrimitive.h/.cpp, OBJLoader.hpp.
// What does not (treated as unbuildable, restated in rt_oracle.c only):
//   raycast.cu / raycast.h (curand_kernel.h absent), Material.hpp / Scene.* / Camera.* /
//   kernel.cu (opencv2 absent).
//
// Q_rsqrt (utils.hpp:12-27) is compiled with `long` = 32 bits, the LLP64 meaning it has on the
// MSVC target of CudaRaytracer.vcxproj; on LP64 the original is an out-of-bounds read.
#include <cuda_runtime.h>
#include <cstring>
#include <cstdio>
#include <cfloat>
#include <vector>
#include <string>
#include "ref_long32.h"
#include "Ray.hpp"
#include "TrianglePrimitive.hpp"
#include "BVHTree.hpp"
#include "MeshInstance.hpp"
#define private public          // read MeshPrimitive::triangles back out (layout unchanged)
#include "MeshPrimitive.h"
#undef private
#include "OBJLoader.hpp"

#define API extern "C" __attribute__((visibility("default")))

static_assert(sizeof(TrianglePrimitive) == 72, "TrianglePrimitive layout");
static_assert(sizeof(lre) == 24, "lre layout");

static float3 F3(const float* v) { return make_float3(v[0], v[1], v[2]); }
static void S3(float* o, float3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

API float ref_q_rsqrt(float x) { return Q_rsqrt(x); }
API void ref_normalize(const float* v, float* o) { S3(o, normalize(F3(v))); }
API float ref_magnitude(const float* v) { return magnitude(F3(v)); }
API void ref_euler2quat(const float* e, float* q) { float4 r = euler2quat(F3(e)); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
API void ref_apply_quat(const float* q, const float* v, float* o) { S3(o, apply_quat(make_float4(q[0], q[1], q[2], q[3]), F3(v))); }
API void ref_apply_euler(const float* e, const float* v, float* o) { S3(o, apply_euler(F3(e), F3(v))); }
API void ref_invert_lre(const float* l, float* o) { lre a; memcpy(&a, l, 24); lre r = invert_lre(a); memcpy(o, &r, 24); }
API void ref_apply_lre(const float* l, const float* v, float* o) { lre a; memcpy(&a, l, 24); S3(o, apply_lre(a, F3(v))); }
API void ref_lre2homo(const float* l, float* o16) { lre a; memcpy(&a, l, 24); float4x4 h = lre2homo(a); memcpy(o16, &h, 64); }
API void ref_invert_intrinsic(const float* K9, float* o9) { float3x3 k; memcpy(&k, K9, 36); float3x3 r = invert_intrinsic(k); memcpy(o9, &r, 36); }
API void ref_apply_matrix33(const float* K9, const float* v, float* o) { float3x3 k; memcpy(&k, K9, 36); S3(o, apply_matrix(k, F3(v))); }

// Ray::Ray (Ray.hpp:17-23): origin, direction, direction_inv = 1 / d per component, color (1,1,1), illumination 0
API void ref_ray_ctor(const float* o, const float* d, float* out13)
{
    Ray r(F3(o), F3(d), make_uint2(3, 4));
    S3(out13, r.origin); S3(out13 + 3, r.direction); S3(out13 + 6, r.direction_inv); S3(out13 + 9, r.color); out13[12] = r.illumination;
}

API float ref_aabb_ray_intersects(const float* bmin, const float* bmax, const float* o, const float* d)
{
    Ray r(F3(o), F3(d), make_uint2(0, 0));
    d_BVHTree n(F3(bmin), F3(bmax), -1, -1, nullptr, 0);
    return n.ray_intersects(r);
}
API void ref_tri_test(const float* tri18, const float* o, const float* d, float* out5)
{
    TrianglePrimitive t; memcpy(&t, tri18, 72);
    Ray r(F3(o), F3(d), make_uint2(0, 0));
    float3 p = t.ray_intersect(r);
    S3(out5, p);
    if (p.x == FLT_MAX) { out5[3] = FLT_MAX; out5[4] = FLT_MAX; return; }
    float2 uv = t.point_inside(p);
    out5[3] = uv.x; out5[4] = uv.y;
}
API void ref_tri_from_vertices(const float* abc9, float* tri18)
{
    TrianglePrimitive t(F3(abc9), F3(abc9 + 3), F3(abc9 + 6));
    memcpy(tri18, &t, 72);
    memset(tri18 + 12, 0, 24);           // uv_coords are uninitialised by this ctor (:15-23)
}
API void ref_tri_center(const float* tri18, float* o) { TrianglePrimitive t; memcpy(&t, tri18, 72); S3(o, t.center()); }

// MeshInstance(int,int,lre,float3) + build_inv (MeshInstance.hpp:31-46); out26 = the 104-byte struct
API void ref_instance_build(int mesh, int mat, const float* pose6, const float* scale3, void* out104)
{
    lre p; memcpy(&p, pose6, 24);
    MeshInstance in(mesh, mat, p, F3(scale3));
    static_assert(sizeof(MeshInstance) == 104, "MeshInstance layout");
    memcpy(out104, &in, 104);
}

struct RefMesh { MeshPrimitive* mesh; };

API RefMesh* ref_mesh_from_triangles(const float* tris18, int n)
{
    std::vector<TrianglePrimitive> v((size_t)n);
    if (n) memcpy((void*)v.data(), tris18, (size_t)n * 72);
    RefMesh* r = new RefMesh;
    r->mesh = new MeshPrimitive(v);      // heap: BVHTree keeps a pointer to its own root member (MeshPrimitive.cpp:51)
    return r;
}
// OBJLoader::load returns by value; the root BVHTree* in master_list_trees[0] then dangles
// (MeshPrimitive.cpp:51 pushes &this->bvh_top of the temporary) -- re-point it at the copy,
// which holds identical contents, before reading anything back.
API RefMesh* ref_obj_load(const char* path)
{
    RefMesh* r = new RefMesh;
    r->mesh = new MeshPrimitive(OBJLoader::load(std::string(path)));
    r->mesh->bvh_top.master_list_trees->at(0) = &r->mesh->bvh_top;
    return r;
}
API int ref_mesh_num_triangles(const RefMesh* m) { return m->mesh->num_triangles; }
API int ref_mesh_num_nodes(const RefMesh* m) { return (int)m->mesh->bvh_top.master_list_trees->size(); }
API void ref_mesh_get_triangles(const RefMesh* m, float* out18) { memcpy(out18, m->mesh->triangles, (size_t)m->mesh->num_triangles * 72); }
API int ref_mesh_get_nodes(const RefMesh* m, float* boxes, int* child, int* leaf_count)
{
    std::vector<BVHTree*>& L = *m->mesh->bvh_top.master_list_trees;
    int total = 0;
    for (size_t i = 0; i < L.size(); i++) {
        BVHTree* n = L[i];
        bool leaf = (n->child_index_a == -1 && n->child_index_b == -1);     // BVHTree.hpp:100
        if (boxes) { boxes[i*6+0] = n->min.x; boxes[i*6+1] = n->min.y; boxes[i*6+2] = n->min.z;
                     boxes[i*6+3] = n->max.x; boxes[i*6+4] = n->max.y; boxes[i*6+5] = n->max.z; }
        if (child) { child[i*2] = n->child_index_a; child[i*2+1] = n->child_index_b; }
        if (leaf_count) leaf_count[i] = leaf ? (int)n->triangle_indices.size() : 0;
        if (leaf) total += (int)n->triangle_indices.size();
    }
    return total;
}
API void ref_mesh_get_leaf_indices(const RefMesh* m, int* out)
{
    std::vector<BVHTree*>& L = *m->mesh->bvh_top.master_list_trees;
    size_t k = 0;
    for (size_t i = 0; i < L.size(); i++) {
        BVHTree* n = L[i];
        if (n->child_index_a == -1 && n->child_index_b == -1)
            for (size_t j = 0; j < n->triangle_indices.size(); j++) out[k++] = n->triangle_indices[j];
    }
}
