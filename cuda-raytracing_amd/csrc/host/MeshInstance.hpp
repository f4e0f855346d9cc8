// MeshInstance.hpp -- placement of one mesh in the world.
//
// Binary layout = RtInstanceDesc of include/rt_hip.h = the reference's MeshInstance (MeshInstance.hpp:6-18):
// two ints, then pose / inverse pose, the two Euler triples, scale / inverse scale -- 104 bytes.  Scene uploads the
// struct with a memcpy, so the field order below is part of the ABI.
#pragma once
#include "utils.hpp"

struct MeshInstance {
    int mesh_index;                 // index into the Scene's meshes, in add_mesh() order
    int material_index;             // index into the Scene's materials, in add_material() order

    lre pose;                       // where the mesh sits (translation + yaw / pitch / roll)
    lre inv_pose;                   // derived, see build_inv()

    float3 rotation;                // (pose.yaw, pose.pitch, pose.roll), derived
    float3 inv_rotation;            // (inv_pose.yaw, inv_pose.pitch, inv_pose.roll), derived

    float3 scale;                   // per-axis size factor
    float3 inv_scale;               // derived

    // Recomputes every derived field from pose and scale (MeshInstance.hpp:39-46).  Scene::upload_to_device and
    // Scene::update_mesh_instance call it, so editing pose / scale after construction is enough.
    void build_inv()
    {
        inv_pose = invert_lre(pose);
        inv_scale = make_float3(1 / scale.x, 1 / scale.y, 1 / scale.z);
        rotation = make_float3(pose.yaw, pose.pitch, pose.roll);
        inv_rotation = make_float3(inv_pose.yaw, inv_pose.pitch, inv_pose.roll);
    }

    // unit scale, identity pose
    MeshInstance(int mesh, int material) : mesh_index(mesh), material_index(material), scale(make_float3(1.0f, 1.0f, 1.0f)) { build_inv(); }
    // explicit placement
    MeshInstance(int mesh, int material, lre placed_at, float3 size) : mesh_index(mesh), material_index(material), pose(placed_at), scale(size) { build_inv(); }
    // the reference's default constructor leaves everything but mesh_index = -1 uninitialised; here it is a valid
    // identity placement of "no mesh"
    MeshInstance() : MeshInstance(-1, 0) {}
};
static_assert(sizeof(MeshInstance) == 104, "MeshInstance must match RtInstanceDesc");
