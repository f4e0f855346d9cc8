// MeshPrimitive.h -- a triangle array plus its BVH (MeshPrimitive.h:27-43).  A scene's device copy is
// made as a whole by Scene::upload_to_device through the C-ABI; to_device() exists for callers of the reference's
// per-mesh upload (MeshPrimitive.h:36, MeshPrimitive.cpp:17-36).
#pragma once
#include <vector>
#include "BVHTree.hpp"
#include "TrianglePrimitive.hpp"

struct RtScene;
// What MeshPrimitive::to_device() returns (the reference's d_MeshPrimitive, MeshPrimitive.h:13-25, holds raw device pointers
// to an AoS triangle array and a d_BVHTree array; here the mesh's device form is the record arrays of the C-ABI): a HOST
// object that owns a one-mesh device scene.  Like the reference's, it is never freed unless the caller does
// (rt_scene_destroy(d->device) + delete d).
struct d_MeshPrimitive {
    int num_triangles = 0;
    RtScene* device = nullptr;      // rt_scene_upload of this mesh alone (no materials, no instances); nullptr if the upload failed
};

class MeshPrimitive {
public:
    explicit MeshPrimitive(std::vector<TrianglePrimitive> triangles);
    // build_on_device = true builds the BVH with rt_bvh_build (GPU) instead of the host builder; same tree.
    // Throws std::runtime_error if the device build fails.
    MeshPrimitive(std::vector<TrianglePrimitive> triangles, bool build_on_device);
    // No tree yet: Scene::upload_to_device hands the triangles over and the GPU builds the tree inside the scene's arrays
    // (rt_scene_upload with num_nodes = 0) -- the shortest way from an OBJ file to a renderable scene.  bvh_top stays empty
    // until sync_tree() builds it on the host (print_stats, or anything else that reads the host tree, needs that first).
    static MeshPrimitive for_device_build(std::vector<TrianglePrimitive> triangles);
    // (a mesh that was refitted since its device build goes through sync_tree() instead: the tree it is rendered with, not a new one)
    bool builds_at_upload() const { return tree_needs_rebuild && built_from.empty(); }
    int num_triangles;
    BVHTree bvh_top;
    d_MeshPrimitive* to_device();                               // MeshPrimitive.h:36: this mesh's records on the current device
    const std::vector<TrianglePrimitive>& triangle_array() const { return triangles; }
    // deformation with fixed connectivity: replaces the triangles' vertices and normals (same count; uv_coords stay as they are) and
    // refits the BVH bounds; false if the count differs.
    // defer_tree = true leaves bvh_top's bounds as they are until sync_tree() -- Scene::refit_mesh does that for an uploaded
    // scene, whose device copy is refitted by the GPU at once: walking the host tree costs twenty times the device refit
    // and is only needed if the mesh is uploaded again or bvh_top is read (call sync_tree() first).
    bool refit(std::vector<TrianglePrimitive> moved, bool defer_tree = false);
    // new triangles altogether (any count): the tree is rebuilt, at once or (defer_tree) when sync_tree() is next called --
    // Scene::rebuild_mesh defers, because the device copy gets its new tree from the GPU build
    void replace(std::vector<TrianglePrimitive> triangles, bool defer_tree = false);
    void sync_tree();
    bool tree_is_stale() const { return tree_stale; }

private:
    MeshPrimitive() : num_triangles(0) {}
    std::vector<TrianglePrimitive> triangles;
    std::vector<TrianglePrimitive> built_from;                  // see refit(): the triangles a pending tree is to be built from, once refits have moved them
    bool tree_stale = false, tree_needs_rebuild = false;
};
