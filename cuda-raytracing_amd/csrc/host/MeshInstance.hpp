// MeshInstance.hpp -- placement of a mesh in the world; field order and build_inv() as in
// MeshInstance.hpp:6-46 (this struct is exactly RtInstanceDesc of include/rt_hip.h).
#pragma once
#include "utils.hpp"

struct MeshInstance {
    int mesh_index;
    int material_index;
    lre pose;
    lre inv_pose;
    float3 rotation;
    float3 inv_rotation;
    float3 scale;
    float3 inv_scale;

    MeshInstance() : mesh_index(-1), material_index(0) { scale = make_float3(1.0f, 1.0f, 1.0f); build_inv(); }
    MeshInstance(int mesh_index, int material_index) : mesh_index(mesh_index), material_index(material_index)
    { scale = make_float3(1.0f, 1.0f, 1.0f); build_inv(); }
    MeshInstance(int mesh_index, int material_index, lre pose, float3 scale)
        : mesh_index(mesh_index), material_index(material_index), pose(pose), scale(scale) { build_inv(); }

    void build_inv()
    {
        inv_pose = invert_lre(pose);
        inv_scale = make_float3(1 / scale.x, 1 / scale.y, 1 / scale.z);
        rotation = make_float3(pose.yaw, pose.pitch, pose.roll);
        inv_rotation = make_float3(inv_pose.yaw, inv_pose.pitch, inv_pose.roll);
    }
};
static_assert(sizeof(MeshInstance) == 104, "MeshInstance must match RtInstanceDesc");
