// BVHTree.hpp -- host BVH builder.  Produces, node for node, the tree that the reference's
// recursive BVHTree::fill builds (BVHTree.hpp:203-292: 5 candidate planes per axis at (s+1)/6 of
// the node extent, centroid partition, cost = half-area x count, strict-less axis choice with
// ties to z, pre-order numbering), because identical topology is what makes node-visit counts
// comparable.  The implementation is not the reference's: the tree is one flat node array over
// one permutation of the triangle indices, and each node's 15 candidate costs come from a single
// binning pass instead of 15 partition passes.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>
#include "TrianglePrimitive.hpp"

struct BVHNode {
    float3 min, max;
    int child_index_a = -1, child_index_b = -1;    // -1 = leaf (BVHTree.hpp:66-67)
    int first = 0, count = 0;                      // this node's triangles: BVHTree::order[first .. first+count)
};

class BVHTree {
public:
    std::vector<BVHNode> nodes;                    // nodes[0] is the root; children follow in pre-order
    std::vector<int> order;                        // triangle indices; every leaf is a contiguous range

    BVHTree() {}
    void build(const TrianglePrimitive* triangles, int num_triangles, int max_depth = 32);   // fill(1, 32)
    // the same tree built on the GPU (rt_bvh_build); returns an rt_hip.h status, the tree is unchanged on failure
    int build_on_device(const TrianglePrimitive* triangles, int num_triangles, int max_depth = 32);
    // same tree over moved triangles: every node's bounds recomputed (what fill()'s bounds pass gives that node), topology,
    // leaf lists and numbering unchanged.  The reference has no refit; the device form is rt_scene_refit_mesh.
    void refit(const TrianglePrimitive* triangles, int num_triangles);
    int max_level() const { return levels_; }
    void print_stats() const;                      // same report as BVHTree.hpp:117-172

    // ---- the reference's per-node queries (BVHTree.hpp:175-201, :294-361) on the flat tree: a node is named by its index ----
    // (the builder itself does not call them: it gets all fifteen candidate costs of a node from one binning pass;
    // tests/test_host_logic.py checks that both give the same numbers)
    void grow_to_include(int node, float3 vertex);                                      // BVHTree.hpp:183-190
    void grow_to_include(int node, const TrianglePrimitive& triangle);                  // BVHTree.hpp:175-181
    float cost(int node) const;                                                         // BVHTree.hpp:192-201: half area x triangle count
    // best of the five candidate planes of `axis` ("x", "y", anything else = z): {cost, split position}, BVHTree.hpp:294-361
    std::pair<float, float> evaluate_split(int node, const std::string& axis, const TrianglePrimitive* triangles) const;
    // The arrays the device upload takes (the role of compile_tree / to_device_compatible, BVHTree.hpp:364-383, whose
    // device-side form is an array of d_BVHTree with per-leaf index lists): per node bounds, children, leaf range; and the
    // leaf index list.  Exactly what RtMeshDesc of include/rt_hip.h points at; Scene::upload_to_device uses it.
    struct DeviceCompatible {
        std::vector<float> node_bounds;            // [node][min xyz, max xyz]
        std::vector<int32_t> node_children;        // [node][a, b], -1 -1 = leaf
        std::vector<int32_t> node_leaf_first, node_leaf_count;
        std::vector<int32_t> leaf_indices;
    };
    DeviceCompatible to_device_compatible() const;
    // BVHTree.hpp:364-383 under its own name: the reference walks the tree's node list, converts every node with
    // to_device_compatible() and uploads the array; here the conversion of the whole tree is one call and the upload is
    // Scene::upload_to_device's (or MeshPrimitive::to_device's), so this returns the arrays that upload takes.
    static DeviceCompatible compile_tree(BVHTree& top) { return top.to_device_compatible(); }

private:
    void fill(int self, int depth, int max_depth);
    const TrianglePrimitive* tris_ = nullptr;
    std::vector<float> centroid_;                  // [n][3]  TrianglePrimitive::center()
    std::vector<float> tbox_;                      // [n][6]  per-triangle bounds
    std::vector<int> scratch_;
    int levels_ = 1;
};
