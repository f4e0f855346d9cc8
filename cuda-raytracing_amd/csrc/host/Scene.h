// Scene.h -- host scene graph with the reference's interface (Scene.h:19-28).  upload_to_device()
// flattens everything and hands it to rt_scene_upload (include/rt_hip.h); the device buffers are
// owned by the library behind `d_scene` instead of the three raw device pointers of Scene.h:23-25.
#pragma once
#include <vector>
#include "Material.hpp"
#include "MeshInstance.hpp"
#include "MeshPrimitive.h"

struct RtScene;

class Scene {
    std::vector<Material> materials;
    std::vector<MeshPrimitive> meshes;
    std::vector<MeshInstance> mesh_instances;

public:
    Scene();
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    void add_material(Material material);
    void add_mesh(MeshPrimitive mesh);
    void add_mesh_instance(MeshInstance mesh_instance);
    int num_materials() const { return (int)materials.size(); }
    Material& material(int index) { return materials[index]; }      // edit before upload_to_device()

    RtScene* d_scene = nullptr;
    int num_mesh_instances = 0;
    int last_error = 0;                            // rt_hip.h status of the last device call
    void upload_to_device();
    void update_mesh_instance(int index, MeshInstance mesh_instance);
};
