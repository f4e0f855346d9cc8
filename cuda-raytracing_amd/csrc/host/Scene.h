// Scene.h -- the host scene graph: what was added, and the device copy made from it.
//
// Same calls as the reference's Scene (Scene.h:19-28): add_* store by value and hand back nothing (indices are the
// insertion order), upload_to_device() (re)builds the device scene, update_mesh_instance() re-poses one instance
// without rebuilding.  The device side is owned by librt_hip.so behind `d_scene` (an RtScene*, see include/rt_hip.h)
// instead of the reference's three raw device pointers.
#pragma once
#include <vector>

#include "Material.hpp"
#include "MeshInstance.hpp"
#include "MeshPrimitive.h"

struct RtScene;

class Scene {
public:
    Scene();
    ~Scene();                                       // releases the device scene
    Scene(const Scene&) = delete;                   // owns device memory
    Scene& operator=(const Scene&) = delete;

    // ---- what the scene contains (host side) ----
    void add_mesh(MeshPrimitive mesh);
    void add_material(Material material);
    void add_mesh_instance(MeshInstance mesh_instance);
    int num_materials() const { return (int)materials.size(); }
    Material& material(int index) { return materials[index]; }      // edit before upload_to_device()

    // ---- device side ----
    void upload_to_device();                        // flatten everything and call rt_scene_upload (Scene.cpp:25-65)
    void update_mesh_instance(int index, MeshInstance mesh_instance);   // rt_scene_update_instance (Scene.cpp:67-74)
    // the same, ordered on a stream instead of synchronising (rt_scene_update_instance_async): for per-frame animation
    void update_mesh_instance(int index, MeshInstance mesh_instance, void* stream);
    // A mesh deforms (same triangle count and order): host copy and device copy get the moved vertices and normals and refitted
    // bounds -- texture coordinates stay; no rebuild, no re-upload of anything else.  Ordered on `stream` like update_mesh_instance(.., stream).
    void refit_mesh(int mesh_index, std::vector<TrianglePrimitive> moved, void* stream = nullptr);
    // A mesh changes beyond what a refit can follow (large motion, or other triangles): the device copy gets a NEW tree, built on
    // the GPU straight into the scene's arrays (rt_scene_rebuild_mesh_device), the host copy rebuilds its tree when it is next
    // needed.  More triangles than the mesh was uploaded with do not fit its part of the arrays: a new device scene is uploaded
    // then (the GPU builds the mesh's tree during the upload) and takes the old one's place once it is complete; that path waits
    // for the whole device, whatever `stream` is.  On EVERY path the host mesh and the device scene change only after the device
    // call has succeeded: on an error (last_error) host and device still describe the old mesh and the old scene still renders.
    // Returns when the new tree is in place.
    void rebuild_mesh(int mesh_index, std::vector<TrianglePrimitive> triangles, void* stream = nullptr);
    RtScene* d_scene = nullptr;
    int num_mesh_instances = 0;
    int last_error = 0;                             // rt_hip.h status of the last device call (the reference ignores errors)

private:
    int upload_as(const std::vector<MeshPrimitive*>& meshes_now, RtScene** out);   // a new device scene from these meshes; touches nothing else
    std::vector<MeshPrimitive> meshes;
    std::vector<Material> materials;
    std::vector<MeshInstance> mesh_instances;
};
