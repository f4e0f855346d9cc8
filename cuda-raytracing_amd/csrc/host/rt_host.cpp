// rt_host.cpp -- implementation of the host C++ API (Scene / Camera / OBJLoader / MeshPrimitive /
// BVHTree / Material).  Pure host code: talks to the GPU only through the C-ABI of
// include/rt_hip.h.  Built with -ffp-contract=off like every file that touches rt_math.h.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cctype>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <atomic>
#include <stdexcept>
#include <chrono>
#include <thread>
#include <type_traits>

#include "../../../include/rt_hip.h"
#include "Camera.h"
#include "ImageIO.hpp"
#include "OBJLoader.hpp"
#include "Scene.h"

// ------------------------------------------------------------------------------ BVHTree

namespace {
struct Box {
    float mn[3], mx[3];
    Box() { for (int k = 0; k < 3; k++) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; } }       // BVHTree.hpp:73-81
    void grow(const float* lo, const float* hi)
    { for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], lo[k]); mx[k] = fmaxf(mx[k], hi[k]); } }
    void merge(const Box& o) { grow(o.mn, o.mx); }
};
// BVHTree::cost, BVHTree.hpp:192-201
float box_cost(const Box& b, size_t count)
{
    if (count == 0) return FLT_MAX;
    float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    float half_area = sx * (sy + sz) + sy * sz;
    return half_area * (float)count;
}
}  // namespace

void BVHTree::build(const TrianglePrimitive* triangles, int n, int max_depth)
{
    tris_ = triangles;
    nodes.clear();
    order.resize((size_t)n);
    centroid_.resize((size_t)n * 3);
    tbox_.resize((size_t)n * 6);
    scratch_.resize((size_t)n);
    levels_ = 1;
    for (int i = 0; i < n; i++) {
        order[i] = i;
        const TrianglePrimitive& t = triangles[i];
        const float3 c = t.center();                                   // TrianglePrimitive.hpp:81-83
        centroid_[3 * (size_t)i] = c.x; centroid_[3 * (size_t)i + 1] = c.y; centroid_[3 * (size_t)i + 2] = c.z;
        Box b;
        for (int k = 0; k < 3; k++) { const float v[3] = {t.vertices[k].x, t.vertices[k].y, t.vertices[k].z}; b.grow(v, v); }
        memcpy(&tbox_[6 * (size_t)i], b.mn, 12); memcpy(&tbox_[6 * (size_t)i + 3], b.mx, 12);
    }
    BVHNode root;
    root.first = 0; root.count = n;
    nodes.push_back(root);                                             // MeshPrimitive.cpp:49-51
    fill(0, 1, max_depth);                                             // MeshPrimitive.cpp:54
    centroid_.clear(); centroid_.shrink_to_fit();
    tbox_.clear(); tbox_.shrink_to_fit();
    scratch_.clear(); scratch_.shrink_to_fit();
}

void BVHTree::fill(int self, int depth, int max_depth)
{
    const int first = nodes[self].first, count = nodes[self].count;
    if (depth > levels_) levels_ = depth;
    // bounds: vertex by vertex in list order, as BVHTree.hpp:206-209 / :175-190
    Box nb;
    for (int i = 0; i < count; i++) {
        const TrianglePrimitive& t = tris_[order[first + i]];
        for (int k = 0; k < 3; k++) { const float v[3] = {t.vertices[k].x, t.vertices[k].y, t.vertices[k].z}; nb.grow(v, v); }
    }
    nodes[self].min = make_float3(nb.mn[0], nb.mn[1], nb.mn[2]);
    nodes[self].max = make_float3(nb.mx[0], nb.mx[1], nb.mx[2]);
    if (depth >= max_depth) return;                                    // :211-215
    if (count <= 1) return;

    // evaluate_split for x, y, z (BVHTree.hpp:294-361).  Plane s of an axis sits at
    // pos_s = min + (max - min) * ((s + 1) / 6); pos_s is non-decreasing in s, so a centroid c
    // goes left of exactly the planes s >= bin(c) where bin = #{s : c > pos_s}.  One pass bins
    // every triangle's box per axis; plane s then has left = bins 0..s, right = bins s+1..5.
    float pos[3][5], eval_cost[3], eval_split[3];
    for (int ax = 0; ax < 3; ax++)
        for (int s = 0; s < 5; s++) {
            float split_t = ((float)s + 1) / (5.0f + 1);
            pos[ax][s] = nb.mn[ax] + (nb.mx[ax] - nb.mn[ax]) * (split_t);
        }
    Box bin_box[3][6];
    size_t bin_cnt[3][6] = {{0}};
    for (int i = 0; i < count; i++) {
        const int t = order[first + i];
        const float* tb = &tbox_[6 * (size_t)t];
        for (int ax = 0; ax < 3; ax++) {
            const float c = centroid_[3 * (size_t)t + ax];
            int b = 0;
            while (b < 5 && !(c <= pos[ax][b])) b++;                   // tri_check <= pos  (:339)
            bin_box[ax][b].grow(tb, tb + 3);
            bin_cnt[ax][b]++;
        }
    }
    for (int ax = 0; ax < 3; ax++) {
        float best_cost = FLT_MAX, best_split = 0.0f;
        Box right_acc[6];                                              // right_acc[s] = union of bins s+1..5
        size_t right_cnt[6];
        Box acc; size_t n = 0;
        for (int b = 5; b >= 1; b--) { acc.merge(bin_box[ax][b]); n += bin_cnt[ax][b]; right_acc[b - 1] = acc; right_cnt[b - 1] = n; }
        Box left; size_t ln = 0;
        for (int s = 0; s < 5; s++) {
            left.merge(bin_box[ax][s]); ln += bin_cnt[ax][s];
            float cost = box_cost(left, ln) + box_cost(right_acc[s], right_cnt[s]);     // :351
            if (cost < best_cost) { best_cost = cost; best_split = pos[ax][s]; }
        }
        eval_cost[ax] = best_cost; eval_split[ax] = best_split;
    }
    int axis; float split_pos, best_cost;                              // :229-243 (strict <, ties fall to z)
    if (eval_cost[0] < eval_cost[1] && eval_cost[0] < eval_cost[2]) { axis = 0; split_pos = eval_split[0]; best_cost = eval_cost[0]; }
    else if (eval_cost[1] < eval_cost[0] && eval_cost[1] < eval_cost[2]) { axis = 1; split_pos = eval_split[1]; best_cost = eval_cost[1]; }
    else { axis = 2; split_pos = eval_split[2]; best_cost = eval_cost[2]; }
    if (best_cost >= box_cost(nb, (size_t)count)) return;              // :246

    // stable partition of order[first .. first+count) by centroid <= split_pos (:253-277)
    int nl = 0, nr = 0;
    for (int i = 0; i < count; i++) {
        const int t = order[first + i];
        if (centroid_[3 * (size_t)t + axis] <= split_pos) order[first + nl++] = t; else scratch_[first + nr++] = t;
    }
    for (int i = 0; i < nr; i++) order[first + nl + i] = scratch_[first + i];
    if (nl == 0 || nr == 0) return;                                    // :279 (order is unchanged in that case)

    const int a = (int)nodes.size();                                   // :283-285
    BVHNode na; na.first = first; na.count = nl;
    nodes.push_back(na);
    nodes[self].child_index_a = a;
    fill(a, depth + 1, max_depth);
    const int b = (int)nodes.size();                                   // :287-289
    BVHNode nbn; nbn.first = first + nl; nbn.count = nr;
    nodes.push_back(nbn);
    nodes[self].child_index_b = b;
    fill(b, depth + 1, max_depth);
}

void BVHTree::grow_to_include(int node, float3 v)
{
    BVHNode& n = nodes[(size_t)node];
    n.min = make_float3(fminf(n.min.x, v.x), fminf(n.min.y, v.y), fminf(n.min.z, v.z));
    n.max = make_float3(fmaxf(n.max.x, v.x), fmaxf(n.max.y, v.y), fmaxf(n.max.z, v.z));
}
void BVHTree::grow_to_include(int node, const TrianglePrimitive& t) { for (int k = 0; k < 3; k++) grow_to_include(node, t.vertices[k]); }

float BVHTree::cost(int node) const
{
    const BVHNode& n = nodes[(size_t)node];
    Box b;
    b.mn[0] = n.min.x; b.mn[1] = n.min.y; b.mn[2] = n.min.z; b.mx[0] = n.max.x; b.mx[1] = n.max.y; b.mx[2] = n.max.z;
    return box_cost(b, (size_t)n.count);
}

// The reference's own way (BVHTree.hpp:294-361): for each of the five planes partition the node's triangles by centroid,
// grow a left and a right box vertex by vertex, add the two costs.
std::pair<float, float> BVHTree::evaluate_split(int node, const std::string& axis, const TrianglePrimitive* triangles) const
{
    const BVHNode& n = nodes[(size_t)node];
    const int ax = axis == "x" ? 0 : (axis == "y" ? 1 : 2);
    const float lo[3] = {n.min.x, n.min.y, n.min.z}, hi[3] = {n.max.x, n.max.y, n.max.z};
    float best_cost = FLT_MAX, best_split = 0.0f;
    for (int s = 0; s < 5; s++) {
        const float split_t = ((float)s + 1) / (5.0f + 1);
        const float pos = lo[ax] + (hi[ax] - lo[ax]) * (split_t);
        Box left, right;
        size_t nl = 0, nr = 0;
        for (int i = 0; i < n.count; i++) {
            const TrianglePrimitive& t = triangles[order[(size_t)(n.first + i)]];
            const float3 c = t.center();
            const float check = ax == 0 ? c.x : (ax == 1 ? c.y : c.z);
            Box& side = check <= pos ? left : right;
            (check <= pos ? nl : nr)++;
            for (int k = 0; k < 3; k++) { const float v[3] = {t.vertices[k].x, t.vertices[k].y, t.vertices[k].z}; side.grow(v, v); }
        }
        const float c = box_cost(left, nl) + box_cost(right, nr);
        if (c < best_cost) { best_cost = c; best_split = pos; }
    }
    return std::make_pair(best_cost, best_split);
}

BVHTree::DeviceCompatible BVHTree::to_device_compatible() const
{
    DeviceCompatible f;
    f.node_bounds.resize(nodes.size() * 6); f.node_children.resize(nodes.size() * 2);
    f.node_leaf_first.resize(nodes.size()); f.node_leaf_count.resize(nodes.size());
    for (size_t k = 0; k < nodes.size(); k++) {
        const BVHNode& nd = nodes[k];
        f.node_bounds[6 * k] = nd.min.x; f.node_bounds[6 * k + 1] = nd.min.y; f.node_bounds[6 * k + 2] = nd.min.z;
        f.node_bounds[6 * k + 3] = nd.max.x; f.node_bounds[6 * k + 4] = nd.max.y; f.node_bounds[6 * k + 5] = nd.max.z;
        f.node_children[2 * k] = nd.child_index_a; f.node_children[2 * k + 1] = nd.child_index_b;
        const bool leaf = nd.child_index_a == -1 && nd.child_index_b == -1;             // BVHTree.hpp:100
        f.node_leaf_first[k] = nd.first; f.node_leaf_count[k] = leaf ? nd.count : 0;
    }
    f.leaf_indices.assign(order.begin(), order.end());
    return f;
}

void BVHTree::refit(const TrianglePrimitive* triangles, int n)
{
    tris_ = triangles;
    for (int k = (int)nodes.size() - 1; k >= 0; k--) {          // children follow their parents in the array
        BVHNode& nd = nodes[(size_t)k];
        Box b;
        if (nd.child_index_a < 0) {
            for (int i = 0; i < nd.count && nd.first + i < n; i++) {
                const TrianglePrimitive& t = triangles[order[(size_t)(nd.first + i)]];
                for (int j = 0; j < 3; j++) { const float v[3] = {t.vertices[j].x, t.vertices[j].y, t.vertices[j].z}; b.grow(v, v); }
            }
        } else {
            const BVHNode& ca = nodes[(size_t)nd.child_index_a];
            const BVHNode& cb = nodes[(size_t)nd.child_index_b];
            const float amn[3] = {ca.min.x, ca.min.y, ca.min.z}, amx[3] = {ca.max.x, ca.max.y, ca.max.z};
            const float bmn[3] = {cb.min.x, cb.min.y, cb.min.z}, bmx[3] = {cb.max.x, cb.max.y, cb.max.z};
            b.grow(amn, amx); b.grow(bmn, bmx);
        }
        nd.min = make_float3(b.mn[0], b.mn[1], b.mn[2]);
        nd.max = make_float3(b.mx[0], b.mx[1], b.mx[2]);
    }
}

int BVHTree::build_on_device(const TrianglePrimitive* triangles, int n, int max_depth)
{
    std::vector<float> v((size_t)n * 9);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            v[9 * (size_t)i + 3 * k] = triangles[i].vertices[k].x;
            v[9 * (size_t)i + 3 * k + 1] = triangles[i].vertices[k].y;
            v[9 * (size_t)i + 3 * k + 2] = triangles[i].vertices[k].z;
        }
    const size_t cap = n > 0 ? 2 * (size_t)n : 1;
    std::vector<float> bounds(cap * 6);
    std::vector<int32_t> children(cap * 2), lfirst(cap), lcount(cap), leaf((size_t)(n > 0 ? n : 1));
    int32_t num = 0, levels = 1;
    int rc = rt_bvh_build(v.data(), n, max_depth, bounds.data(), children.data(), lfirst.data(), lcount.data(), leaf.data(), &num, &levels);
    if (rc) return rc;
    nodes.assign((size_t)num, BVHNode());
    for (int k = 0; k < num; k++) {
        BVHNode& nd = nodes[k];
        nd.min = make_float3(bounds[6 * k], bounds[6 * k + 1], bounds[6 * k + 2]);
        nd.max = make_float3(bounds[6 * k + 3], bounds[6 * k + 4], bounds[6 * k + 5]);
        nd.child_index_a = children[2 * k]; nd.child_index_b = children[2 * k + 1];
        nd.first = lfirst[k];
    }
    // BVHNode::count is the node's own triangle count for interior nodes too: recover it bottom-up (children follow parents)
    for (int k = num - 1; k >= 0; k--)
        nodes[k].count = nodes[k].child_index_a < 0 ? lcount[k] : nodes[nodes[k].child_index_a].count + nodes[nodes[k].child_index_b].count;
    order.assign(leaf.begin(), leaf.begin() + n);
    levels_ = levels;
    tris_ = triangles;
    return RT_OK;
}

void BVHTree::print_stats() const
{
    int count_nodes = 0, max_t = 0, min_t = 1000000, max_depth = 0, count_leaves = 0;
    std::vector<int> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const BVHNode& n = nodes[stack.back()];
        stack.pop_back();
        max_depth = (int)fmaxf((float)max_depth, (float)stack.size());
        count_nodes++;
        if (n.child_index_a == -1) {
            if (n.count > max_t) max_t = n.count;
            if (n.count < min_t) min_t = n.count;
            count_leaves++;
        } else {
            stack.push_back(n.child_index_a);
            stack.push_back(n.child_index_b);
        }
    }
    float avg = (float)nodes[0].count / (float)count_leaves;
    std::cout << "BVH Stats: " << std::endl;
    std::cout << "Number of nodes: " << count_nodes << std::endl;
    std::cout << "Max triangles per node: " << max_t << std::endl;
    std::cout << "Min triangles per node: " << min_t << std::endl;
    std::cout << "Max depth: " << max_depth << std::endl;
    std::cout << "Number of leaves: " << count_leaves << std::endl;
    std::cout << "Average triangles per leaf: " << avg << std::endl;
}

// -------------------------------------------------------------------------- MeshPrimitive

MeshPrimitive::MeshPrimitive(std::vector<TrianglePrimitive> tris) : triangles(std::move(tris))
{
    num_triangles = (int)triangles.size();
    bvh_top.build(triangles.data(), num_triangles, 32);
}

MeshPrimitive::MeshPrimitive(std::vector<TrianglePrimitive> tris, bool build_on_device) : triangles(std::move(tris))
{
    num_triangles = (int)triangles.size();
    if (!build_on_device) { bvh_top.build(triangles.data(), num_triangles, 32); return; }
    int rc = bvh_top.build_on_device(triangles.data(), num_triangles, 32);
    if (rc) throw std::runtime_error(std::string("MeshPrimitive: GPU BVH build failed: ") + rt_error_string(rc));
}

// MeshPrimitive.cpp:17-36: the reference uploads the triangle array and the compiled tree and returns a device struct of
// raw pointers; the C-ABI's device form of a mesh is its part of a scene's record arrays, so this uploads a scene that holds
// nothing but this mesh.
static void flatten_mesh(const MeshPrimitive& m, std::vector<float>& v, std::vector<float>& n, std::vector<float>& uv)
{
    const auto& tris = m.triangle_array();
    v.resize(tris.size() * 9); n.resize(tris.size() * 3); uv.resize(tris.size() * 6);
    for (size_t t = 0; t < tris.size(); t++) {
        for (int k = 0; k < 3; k++) {
            v[9 * t + 3 * k] = tris[t].vertices[k].x; v[9 * t + 3 * k + 1] = tris[t].vertices[k].y; v[9 * t + 3 * k + 2] = tris[t].vertices[k].z;
            uv[6 * t + 2 * k] = tris[t].uv_coords[k].x; uv[6 * t + 2 * k + 1] = tris[t].uv_coords[k].y;
        }
        n[3 * t] = tris[t].normal.x; n[3 * t + 1] = tris[t].normal.y; n[3 * t + 2] = tris[t].normal.z;
    }
}

d_MeshPrimitive* MeshPrimitive::to_device()
{
    const bool device_build = builds_at_upload();
    if (!device_build) sync_tree();
    std::vector<float> v, n, uv;
    flatten_mesh(*this, v, n, uv);
    RtMeshDesc d;
    memset(&d, 0, sizeof d);
    d.num_triangles = num_triangles;
    d.vertices = v.data(); d.normals = n.data(); d.uvs = uv.data();
    BVHTree::DeviceCompatible tree;
    if (!device_build) {
        tree = BVHTree::compile_tree(bvh_top);                    // BVHTree.hpp:364-383
        d.num_nodes = (int32_t)bvh_top.nodes.size();
        d.node_bounds = tree.node_bounds.data(); d.node_children = tree.node_children.data();
        d.node_leaf_first = tree.node_leaf_first.data(); d.node_leaf_count = tree.node_leaf_count.data();
        d.num_leaf_indices = (int32_t)tree.leaf_indices.size(); d.leaf_indices = tree.leaf_indices.data();
    }
    RtSceneDesc sd;
    memset(&sd, 0, sizeof sd);
    sd.num_meshes = 1; sd.meshes = &d;
    d_MeshPrimitive* out = new d_MeshPrimitive;
    out->num_triangles = num_triangles;
    if (rt_scene_upload(&sd, &out->device) != RT_OK) out->device = nullptr;
    return out;
}

Material* Material::to_device() const
{
    RtMaterialDesc d;
    memset(&d, 0, sizeof d);
    d.roughness = roughness; d.albedo[0] = albedo.x; d.albedo[1] = albedo.y; d.albedo[2] = albedo.z;
    d.metallic = metallic; d.illumination = illumination;
    void* dev = nullptr;
    if (rt_malloc(&dev, sizeof d) != RT_OK) return nullptr;
    if (rt_memcpy_h2d(dev, &d, sizeof d, nullptr) != RT_OK) { (void)rt_free(dev); return nullptr; }
    return reinterpret_cast<Material*>(dev);
}

MeshPrimitive MeshPrimitive::for_device_build(std::vector<TrianglePrimitive> tris)
{
    MeshPrimitive m;
    m.triangles = std::move(tris);
    m.num_triangles = (int)m.triangles.size();
    m.tree_stale = m.tree_needs_rebuild = true;
    return m;
}

// ------------------------------------------------------------------------------ Material

void Material::set_texture_bgr(const uint8_t* bgr, int width, int height, size_t pitch)
{
    texture.assign((size_t)width * 3 * (size_t)height, 0);
    for (int y = 0; y < height; y++) memcpy(&texture[(size_t)y * width * 3], bgr + (size_t)y * pitch, (size_t)width * 3);
    texture_width = width; texture_height = height;
}

bool Material::upload_texture(const std::string& path)
{
    // Material.hpp:29-43 decodes with cv::imread; here: PNG, baseline JPEG or binary PPM by file signature (ImageIO.hpp)
    std::vector<uint8_t> bgr;
    int w = 0, h = 0;
    if (!read_image_bgr(path, bgr, w, h)) return false;
    texture.swap(bgr);
    texture_width = w; texture_height = h;
    return true;
}

// --------------------------------------------------------------------------------- Scene

Scene::Scene() {}
Scene::~Scene() { if (d_scene) rt_scene_destroy(d_scene); }
void Scene::add_material(Material material) { materials.push_back(std::move(material)); }
void Scene::add_mesh(MeshPrimitive mesh) { meshes.push_back(std::move(mesh)); }
void Scene::add_mesh_instance(MeshInstance mesh_instance) { mesh_instances.push_back(mesh_instance); }

static RtInstanceDesc to_desc(const MeshInstance& in)
{
    RtInstanceDesc d;
    static_assert(sizeof(RtInstanceDesc) == sizeof(MeshInstance), "layouts must agree");
    memcpy(&d, &in, sizeof d);
    return d;
}

// Flatten `meshes_now` (the scene's meshes, one of them possibly replaced by a candidate) with the scene's materials and instances
// and upload them as a NEW device scene.  Nothing the Scene owns is touched: the caller swaps on success.
int Scene::upload_as(const std::vector<MeshPrimitive*>& meshes_now, RtScene** out)
{
    struct Flat { std::vector<float> v, n, uv; BVHTree::DeviceCompatible tree; };
    std::vector<Flat> flat(meshes_now.size());
    std::vector<RtMeshDesc> md(meshes_now.size());
    for (size_t i = 0; i < meshes_now.size(); i++) {
        const bool device_build = meshes_now[i]->builds_at_upload();  // no host tree wanted: the GPU builds it inside the scene's arrays
        if (!device_build) meshes_now[i]->sync_tree();                // (a mesh refitted on the device only: its host tree catches up now)
        const MeshPrimitive& m = *meshes_now[i];
        Flat& f = flat[i];
        const auto& tris = m.triangle_array();
        f.v.resize(tris.size() * 9); f.n.resize(tris.size() * 3); f.uv.resize(tris.size() * 6);
        for (size_t t = 0; t < tris.size(); t++) {
            for (int k = 0; k < 3; k++) {
                f.v[9 * t + 3 * k] = tris[t].vertices[k].x; f.v[9 * t + 3 * k + 1] = tris[t].vertices[k].y; f.v[9 * t + 3 * k + 2] = tris[t].vertices[k].z;
                f.uv[6 * t + 2 * k] = tris[t].uv_coords[k].x; f.uv[6 * t + 2 * k + 1] = tris[t].uv_coords[k].y;
            }
            f.n[3 * t] = tris[t].normal.x; f.n[3 * t + 1] = tris[t].normal.y; f.n[3 * t + 2] = tris[t].normal.z;
        }
        RtMeshDesc& d = md[i];
        memset(&d, 0, sizeof d);
        d.num_triangles = m.num_triangles;
        d.vertices = f.v.data(); d.normals = f.n.data(); d.uvs = f.uv.data();
        if (device_build) continue;                              // num_nodes = 0
        f.tree = m.bvh_top.to_device_compatible();                // BVHTree.hpp:364-383
        d.num_nodes = (int32_t)m.bvh_top.nodes.size();
        d.node_bounds = f.tree.node_bounds.data(); d.node_children = f.tree.node_children.data();
        d.node_leaf_first = f.tree.node_leaf_first.data(); d.node_leaf_count = f.tree.node_leaf_count.data();
        d.num_leaf_indices = (int32_t)f.tree.leaf_indices.size(); d.leaf_indices = f.tree.leaf_indices.data();
    }
    std::vector<RtMaterialDesc> mat(materials.size());
    for (size_t i = 0; i < materials.size(); i++) {
        const Material& m = materials[i];
        RtMaterialDesc& d = mat[i];
        memset(&d, 0, sizeof d);
        d.roughness = m.roughness; d.albedo[0] = m.albedo.x; d.albedo[1] = m.albedo.y; d.albedo[2] = m.albedo.z;
        d.metallic = m.metallic; d.illumination = m.illumination;
        if (m.texture_width > 0) {
            d.texture = m.texture.data(); d.texture_width = m.texture_width; d.texture_height = m.texture_height;
            d.texture_pitch = (size_t)m.texture_width * 3;
        }
    }
    std::vector<RtInstanceDesc> inst(mesh_instances.size());
    for (size_t i = 0; i < mesh_instances.size(); i++) {
        mesh_instances[i].build_inv();                                 // Scene.cpp:59
        inst[i] = to_desc(mesh_instances[i]);
    }
    RtSceneDesc sd;
    sd.num_meshes = (int32_t)md.size(); sd.meshes = md.data();
    sd.num_materials = (int32_t)mat.size(); sd.materials = mat.data();
    sd.num_instances = (int32_t)inst.size(); sd.instances = inst.data();
    return rt_scene_upload(&sd, out);
}

void Scene::upload_to_device()
{
    if (d_scene) { rt_scene_destroy(d_scene); d_scene = nullptr; }     // Scene.cpp:28-39
    std::vector<MeshPrimitive*> all;
    for (auto& m : meshes) all.push_back(&m);
    last_error = upload_as(all, &d_scene);
    num_mesh_instances = (int)mesh_instances.size();
    if (last_error) std::cerr << "Scene::upload_to_device: " << rt_error_string(last_error) << std::endl;
}

bool MeshPrimitive::refit(std::vector<TrianglePrimitive> moved, bool defer_tree)
{
    if ((int)moved.size() != num_triangles) return false;
    // A refit moves vertices and normals; texture coordinates stay (rt_scene_refit_mesh takes none -- "uvs and the tree's topology
    // stay", rt_hip.h): whatever uv_coords the caller's triangles carry, the host copy keeps the ones it has, so that a later
    // upload_to_device() sends what the refitted device copy renders with.
    for (size_t i = 0; i < moved.size(); i++)
        for (int k = 0; k < 3; k++) moved[i].uv_coords[k] = triangles[i].uv_coords[k];
    // A mesh whose tree is still to be built on the host (rebuilt on the device, or made for a device build): the tree the device copy
    // has was built from the triangles as they are NOW, and this refit keeps its topology -- remember them, so that the host tree, when
    // it is built at last, is that tree refitted and not a new one over the moved triangles (an upload must not change what is rendered:
    // among exactly coincident triangles the one reported depends on the topology).
    if (tree_needs_rebuild && built_from.empty()) built_from = triangles;
    triangles = std::move(moved);
    tree_stale = true;
    if (!defer_tree) sync_tree();
    return true;
}

void MeshPrimitive::replace(std::vector<TrianglePrimitive> tris, bool defer_tree)
{
    triangles = std::move(tris);
    num_triangles = (int)triangles.size();
    built_from.clear();
    tree_stale = tree_needs_rebuild = true;
    if (!defer_tree) sync_tree();
}

void MeshPrimitive::sync_tree()
{
    if (!tree_stale) return;
    if (tree_needs_rebuild && !built_from.empty()) {
        // built on the device from `built_from`, refitted since: the same tree here (the GPU builder when there is a device -- the two
        // builders give the same tree --, the host builder otherwise), then the bounds of the triangles as they are now
        if (bvh_top.build_on_device(built_from.data(), num_triangles, 32) != RT_OK) bvh_top.build(built_from.data(), num_triangles);
        bvh_top.refit(triangles.data(), num_triangles);
        std::vector<TrianglePrimitive>().swap(built_from);
    }
    else if (tree_needs_rebuild) bvh_top.build(triangles.data(), num_triangles);  // MeshPrimitive.cpp:38-56
    else bvh_top.refit(triangles.data(), num_triangles);
    tree_stale = tree_needs_rebuild = false;
}

void Scene::rebuild_mesh(int mesh_index, std::vector<TrianglePrimitive> tris, void* stream)
{
    if (mesh_index < 0 || mesh_index >= (int)meshes.size()) { last_error = RT_E_INVALID; return; }
    MeshPrimitive& m = meshes[(size_t)mesh_index];
    if (!d_scene) {                                             // not uploaded yet: upload_to_device() will send the new mesh
        m.replace(std::move(tris), false);
        last_error = RT_OK;
        return;
    }
    // The device goes first and the host mesh follows only when the device has the new tree: host and device must never
    // describe different meshes (a later refit_mesh would be refused for its triangle count with nothing to explain it).
    const size_t n = tris.size();
    int32_t capacity = 0;
    last_error = rt_scene_mesh_capacity(d_scene, mesh_index, &capacity);
    if (last_error) return;
    if (n > (size_t)capacity) {
        // More triangles than the mesh's part of the device arrays has room for: the whole scene is uploaded again -- as a NEW device
        // scene built from a candidate mesh, while host and device still describe the old one.  Only when that upload has succeeded
        // is the old device scene released and the host mesh replaced; on an error both stay as they were (last_error says why).
        // The candidate has no host tree: the GPU builds the tree during the upload (byte-identical to the host builder's, and
        // without its seconds for a large mesh); the host copy is built when something next reads it (sync_tree).
        // `stream`: this path replaces every device array of the scene, so it waits for the whole device first (frames in flight on
        // any stream still read the old arrays) -- the one case in which this call is not merely ordered on `stream`.
        MeshPrimitive candidate = MeshPrimitive::for_device_build(std::move(tris));
        std::vector<MeshPrimitive*> all;
        for (auto& each : meshes) all.push_back(&each == &m ? &candidate : &each);
        RtScene* fresh = nullptr;
        last_error = upload_as(all, &fresh);
        if (last_error) return;                                 // (rt_scene_upload releases what it had allocated)
        last_error = rt_device_synchronize();
        if (last_error) { (void)rt_scene_destroy(fresh); return; }
        (void)rt_scene_destroy(d_scene);
        d_scene = fresh;
        m = std::move(candidate);
        num_mesh_instances = (int)mesh_instances.size();
        return;
    }
    std::vector<float> host(n * 18);                            // vertices [n][9], normals [n][3], uvs [n][6]
    float *v = host.data(), *nn = v + n * 9, *uv = nn + n * 3;
    for (size_t t = 0; t < n; t++) {
        const TrianglePrimitive& tr = tris[t];
        for (int k = 0; k < 3; k++) {
            v[9 * t + 3 * k] = tr.vertices[k].x; v[9 * t + 3 * k + 1] = tr.vertices[k].y; v[9 * t + 3 * k + 2] = tr.vertices[k].z;
            uv[6 * t + 2 * k] = tr.uv_coords[k].x; uv[6 * t + 2 * k + 1] = tr.uv_coords[k].y;
        }
        nn[3 * t] = tr.normal.x; nn[3 * t + 1] = tr.normal.y; nn[3 * t + 2] = tr.normal.z;
    }
    void* d = nullptr;
    last_error = rt_malloc(&d, host.size() * sizeof(float));
    if (last_error) return;
    last_error = rt_memcpy_h2d(d, host.data(), host.size() * sizeof(float), stream);
    const float* dv = (const float*)d;
    if (last_error == RT_OK) last_error = rt_scene_rebuild_mesh_device(d_scene, mesh_index, dv, dv + n * 9, dv + n * 12, (int32_t)n, stream);
    (void)rt_free(d);
    if (last_error == RT_OK) m.replace(std::move(tris), true);  // (the host tree is rebuilt when it is next needed)
}

void Scene::refit_mesh(int mesh_index, std::vector<TrianglePrimitive> moved, void* stream)
{
    // (an uploaded scene's device copy is refitted below; the host tree catches up when it is next needed)
    if (mesh_index < 0 || mesh_index >= (int)meshes.size() || !meshes[(size_t)mesh_index].refit(std::move(moved), d_scene != nullptr)) { last_error = RT_E_INVALID; return; }
    if (!d_scene) { last_error = RT_OK; return; }               // not uploaded yet: upload_to_device() will send the moved mesh
    const MeshPrimitive& m = meshes[(size_t)mesh_index];
    std::vector<float> v((size_t)m.num_triangles * 9), n((size_t)m.num_triangles * 3);
    for (int t = 0; t < m.num_triangles; t++) {
        const TrianglePrimitive& tr = m.triangle_array()[(size_t)t];
        for (int k = 0; k < 3; k++) { v[9 * (size_t)t + 3 * k] = tr.vertices[k].x; v[9 * (size_t)t + 3 * k + 1] = tr.vertices[k].y; v[9 * (size_t)t + 3 * k + 2] = tr.vertices[k].z; }
        n[3 * (size_t)t] = tr.normal.x; n[3 * (size_t)t + 1] = tr.normal.y; n[3 * (size_t)t + 2] = tr.normal.z;
    }
    last_error = rt_scene_refit_mesh(d_scene, mesh_index, v.data(), n.data(), m.num_triangles, stream);
    if (last_error == RT_OK) last_error = rt_stream_synchronize(stream);   // v and n die with this call
}

void Scene::update_mesh_instance(int index, MeshInstance mesh_instance)
{
    if (index < 0 || index >= (int)mesh_instances.size()) { last_error = RT_E_INVALID; return; }
    mesh_instances[index] = mesh_instance;
    mesh_instances[index].build_inv();                                 // Scene.cpp:71
    RtInstanceDesc d = to_desc(mesh_instances[index]);
    last_error = d_scene ? rt_scene_update_instance(d_scene, index, &d) : RT_E_INVALID;
}

void Scene::update_mesh_instance(int index, MeshInstance mesh_instance, void* stream)
{
    if (index < 0 || index >= (int)mesh_instances.size()) { last_error = RT_E_INVALID; return; }
    mesh_instances[index] = mesh_instance;
    mesh_instances[index].build_inv();
    RtInstanceDesc d = to_desc(mesh_instances[index]);
    last_error = d_scene ? rt_scene_update_instance_async(d_scene, index, &d, stream) : RT_E_INVALID;
}

// -------------------------------------------------------------------------------- Camera

Camera::Camera(int width, int height, float3x3 K, float4 D) : width(width), height(height), K(K), D(D)
{
    K_inv = invert_intrinsic(K);                                       // Camera.cu:12
    pose = lre();
}

static RtCameraParams camera_params(const Camera& c, const lre& pose)
{
    RtCameraParams p;
    p.width = c.width; p.height = c.height;
    memcpy(p.K_inv, &c.K_inv, sizeof p.K_inv);
    p.D[0] = c.D.x; p.D[1] = c.D.y; p.D[2] = c.D.z; p.D[3] = c.D.w;
    lre inv = invert_lre(pose);                                        // Camera.cu:21
    memcpy(p.camera_pose, &pose, sizeof p.camera_pose);
    memcpy(p.inv_camera_pose, &inv, sizeof p.inv_camera_pose);
    return p;
}

void Camera::render_scene(Scene& scene, uchar3* img_ptr, size_t pitch, bool synchronize)
{
    if (spp != 1 || bounces != 0 || lighting) { render_scene_ex(scene, img_ptr, pitch, nullptr, synchronize); return; }
    RtCameraParams p = camera_params(*this, pose);
    // The reference's call shape -- asynchronous, default stream, two images per synchronise (kernel.cu:277-279) -- keeps its
    // default-stream ordering and lets frames into different images overlap (rt_render_overlapped, rt_hip.h)
    if (stream == nullptr && !synchronize) last_error = rt_render_overlapped(scene.d_scene, &p, (uint8_t*)img_ptr, pitch);
    else last_error = rt_render(scene.d_scene, &p, (uint8_t*)img_ptr, pitch, stream, synchronize ? 1 : 0);
}

void Camera::render_scene_ex(Scene& scene, uchar3* img_ptr, size_t pitch, int* d_total_pops, bool synchronize)
{
    RtCameraParams p = camera_params(*this, pose);
    RtRenderOptions o;
    o.spp = spp; o.bounces = bounces; o.lighting = lighting ? 1 : 0;
    last_error = rt_render_ex(scene.d_scene, &p, &o, (uint8_t*)img_ptr, pitch, d_total_pops, stream, synchronize ? 1 : 0);
}

void Camera::render_scene_stripes(Scene& scene, uchar3* local_ptr, size_t local_pitch, int stripe_rows, int rank, int num_ranks,
                                  bool synchronize)
{
    RtCameraParams p = camera_params(*this, pose);
    if (spp != 1 || bounces != 0 || lighting) {
        RtRenderOptions o;
        o.spp = spp; o.bounces = bounces; o.lighting = lighting ? 1 : 0;
        last_error = rt_render_ex_stripes(scene.d_scene, &p, &o, (uint8_t*)local_ptr, local_pitch, stripe_rows, rank, num_ranks, stream,
                                          synchronize ? 1 : 0);
        return;
    }
    last_error = rt_render_stripes(scene.d_scene, &p, (uint8_t*)local_ptr, local_pitch, stripe_rows, rank, num_ranks, stream,
                                   synchronize ? 1 : 0);
}

void Camera::render_scene_tiled(Scene& scene, RtComm* comm, uchar3* img_ptr, size_t pitch, bool synchronize, int stripe_rows, int root)
{
    RtCameraParams p = camera_params(*this, pose);
    RtRenderOptions o;
    o.spp = spp; o.bounces = bounces; o.lighting = lighting ? 1 : 0;
    last_error = rt_render_tiled(scene.d_scene, comm, &p, &o, (uint8_t*)img_ptr, pitch, stripe_rows, root, stream, synchronize ? 1 : 0);
}

void Camera::render_scene_batch(Scene& scene, const lre* poses, int count, uchar3* const* img_ptrs, size_t pitch, bool synchronize)
{
    if (count < 1 || count > RT_MAX_BATCH || !poses || !img_ptrs) { last_error = RT_E_INVALID; return; }
    RtCameraParams p[RT_MAX_BATCH];
    for (int i = 0; i < count; i++) p[i] = camera_params(*this, poses[i]);
    last_error = rt_render_batch(scene.d_scene, p, (uint8_t* const*)img_ptrs, pitch, count, stream, synchronize ? 1 : 0);
}

void Camera::render_scene_stripes_batch(Scene& scene, const lre* poses, int count, uchar3* const* local_ptrs, size_t local_pitch,
                                        int stripe_rows, int rank, int num_ranks, bool synchronize, int rotate_first)
{
    if (count < 1 || count > RT_MAX_BATCH || !poses || !local_ptrs) { last_error = RT_E_INVALID; return; }
    RtCameraParams p[RT_MAX_BATCH];
    for (int i = 0; i < count; i++) p[i] = camera_params(*this, poses[i]);
    if (rotate_first >= 0)
        last_error = rt_render_stripes_batch_rotating(scene.d_scene, p, (uint8_t* const*)local_ptrs, local_pitch, count, stripe_rows, rank,
                                                      num_ranks, rotate_first, stream, synchronize ? 1 : 0);
    else
        last_error = rt_render_stripes_batch(scene.d_scene, p, (uint8_t* const*)local_ptrs, local_pitch, count, stripe_rows, rank,
                                             num_ranks, stream, synchronize ? 1 : 0);
}

// ------------------------------------------------------------------------------- PNG out

namespace {
uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void png_chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data)
{
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xFFFFFFFFu, &out[start], out.size() - start) ^ 0xFFFFFFFFu);
}
}  // namespace

int write_png_bgr(const char* path, const unsigned char* bgr, int width, int height, size_t pitch)
{
    if (!path || !bgr || width <= 0 || height <= 0 || pitch < (size_t)width * 3) return RT_E_INVALID;
    // raw scanlines: filter byte 0 + RGB
    std::vector<unsigned char> raw((size_t)height * (1 + (size_t)width * 3));
    for (int y = 0; y < height; y++) {
        unsigned char* d = &raw[(size_t)y * (1 + (size_t)width * 3)];
        const unsigned char* s = bgr + (size_t)y * pitch;
        *d++ = 0;
        for (int x = 0; x < width; x++) { d[3 * x] = s[3 * x + 2]; d[3 * x + 1] = s[3 * x + 1]; d[3 * x + 2] = s[3 * x]; }   // B,G,R -> R,G,B (H12)
    }
    // zlib stream of stored blocks
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", std::vector<unsigned char>());
    FILE* f = fopen(path, "wb");
    if (!f) return RT_E_INVALID;
    size_t w = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return w == out.size() ? RT_OK : RT_E_INVALID;
}

int save_png(const char* path, const uchar3* d_img, int width, int height, size_t pitch, void* stream)
{
    if (!d_img || width <= 0 || height <= 0) return RT_E_INVALID;
    std::vector<unsigned char> host((size_t)width * 3 * (size_t)height);
    int rc = rt_memcpy2d_d2h(host.data(), (size_t)width * 3, d_img, pitch, (size_t)width * 3, (size_t)height, stream);
    if (rc) return rc;
    return write_png_bgr(path, host.data(), width, height, (size_t)width * 3);
}

// ----------------------------------------------------------------------------- OBJLoader

namespace {
// ---- scanner over the memory-mapped file: no stream, no per-token std::string, no strtof in the common case ----
inline bool is_space(unsigned char c) { return c == ' ' || (unsigned)(c - '\t') <= (unsigned)('\r' - '\t'); }     // isspace() in the "C" locale

// std::stoi semantics on [p, end): optional sign + digits, anything after is ignored; false if there is no digit or the
// value does not fit an int (stoi throws in both cases)
inline bool scan_int(const char* p, const char* end, int& out)
{
    bool neg = false;
    if (p < end && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    if (p >= end || *p < '0' || *p > '9') return false;
    long long v = 0;
    while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > (1LL << 31)) return false; p++; }
    if (neg) v = -v;
    if (v > 2147483647LL || v < -2147483648LL) return false;
    out = (int)v;
    return true;
}

// the same, moving p past the sign and digits it consumed
inline bool scan_int_prefix(const char*& p, const char* end, int& out)
{
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '+' || *q == '-')) { neg = *q == '-'; q++; }
    if (q >= end || (unsigned)(*q - '0') > 9u) return false;
    long long v = 0;
    while (q < end && (unsigned)(*q - '0') <= 9u) { v = v * 10 + (*q - '0'); if (v > (1LL << 31)) return false; q++; }
    if (neg) v = -v;
    if (v > 2147483647LL || v < -2147483648LL) return false;
    out = (int)v;
    p = q;
    return true;
}

const float kPow10f[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
const double kPow10d[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// std::stof semantics on the token [p, end): the float nearest to the decimal value of the longest numeric prefix (what
// glibc's strtof returns: correctly rounded, ties to even).  Decimal digits are gathered exactly; then
//   * at most 7 significant digits and |exponent| <= 10 (every number a "%.6f" file holds): float(m) and 10^k are exact in
//     fp32, so ONE fp32 multiplication or division rounds the exact value once -- the correctly rounded result;
//   * up to 19 digits and |exponent| <= 22: the same in double (exact operands, one rounding), then double -> float.  Two
//     roundings could differ from one only if the double landed exactly on the midpoint of two floats; that case falls
//     through;
//   * everything else (longer mantissas, huge exponents, inf / nan / hex) goes to strtof on a bounded copy of the token.
// false = no conversion (stof would throw).
bool scan_float(const char* p, const char* end, float& out)
{
    const char* const tok = p;
    bool neg = false;
    if (p < end && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    unsigned long long m = 0;
    int digits = 0, sig = 0, e10 = 0;
    bool any = false, exact = true;
    if (p + 1 < end && p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) exact = false;    // hexadecimal float: strtof's business
    while (p < end && *p >= '0' && *p <= '9') {
        any = true;
        if (sig < 19) { m = m * 10 + (unsigned)(*p - '0'); if (m) sig++; } else { e10++; if (*p != '0') exact = false; }
        p++; digits++;
    }
    if (p < end && *p == '.') {
        p++;
        while (p < end && *p >= '0' && *p <= '9') {
            any = true;
            if (sig < 19) { m = m * 10 + (unsigned)(*p - '0'); if (m) sig++; e10--; } else if (*p != '0') exact = false;
            p++; digits++;
        }
    }
    bool simple = any && exact;
    if (any && p < end && (*p == 'e' || *p == 'E')) {
        const char* q = p + 1;
        bool eneg = false;
        if (q < end && (*q == '+' || *q == '-')) { eneg = *q == '-'; q++; }
        if (q < end && *q >= '0' && *q <= '9') {
            int ev = 0;
            while (q < end && *q >= '0' && *q <= '9') { if (ev < 100000) ev = ev * 10 + (*q - '0'); q++; }
            e10 += eneg ? -ev : ev;
            p = q;
        }
    }
    if (simple) {
        if (m == 0) { out = neg ? -0.0f : 0.0f; return true; }
        if (m < (1ull << 24) && e10 >= -10 && e10 <= 10) {
            const float f = e10 < 0 ? (float)m / kPow10f[-e10] : (float)m * kPow10f[e10];
            out = neg ? -f : f;
            return true;
        }
        if (m < (1ull << 53) && e10 >= -22 && e10 <= 22) {
            const double d = e10 < 0 ? (double)m / kPow10d[-e10] : (double)m * kPow10d[e10];
            uint64_t bits;
            memcpy(&bits, &d, 8);
            const bool normal_float_range = d >= 1.1754943508222875e-38 && d <= 3.4028234663852886e38;
            if (normal_float_range && (bits & 0x1fffffffull) != 0x10000000ull) {     // not a float midpoint: rounding once more is safe
                const float f = (float)d;
                out = neg ? -f : f;
                return true;
            }
        }
    }
    // the general case: strtof on a NUL-terminated copy (the mapping has no terminator, and strtof would run on into the next line)
    char buf[128];
    size_t n = (size_t)(end - tok);
    std::string big;
    const char* src;
    if (n < sizeof buf) { memcpy(buf, tok, n); buf[n] = 0; src = buf; } else { big.assign(tok, n); src = big.c_str(); }
    if (is_space((unsigned char)src[0])) return false;
    char* e = nullptr;
    const float f = strtof(src, &e);
    if (e == src) return false;
    out = f;
    return true;
}

// next whitespace-separated token of [p, end): false at the end of the line
inline bool next_token(const char*& p, const char* end, const char*& tb, const char*& te)
{
    while (p < end && is_space((unsigned char)*p)) p++;
    if (p >= end) return false;
    tb = p;
    while (p < end && !is_space((unsigned char)*p)) p++;
    te = p;
    return true;
}

struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    bool ok = false;
    explicit MappedFile(const char* path)
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            size = (size_t)st.st_size;
            if (size == 0) ok = true;
            else {
                void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
                if (m != MAP_FAILED) { data = (const char*)m; ok = true; (void)madvise(m, size, MADV_SEQUENTIAL); }
            }
        } else if (fstat(fd, &st) == 0) {
            // not a regular file (a pipe, /dev/stdin): read it
            std::string* s = new std::string;
            char buf[1 << 16];
            ssize_t n;
            while ((n = read(fd, buf, sizeof buf)) > 0) s->append(buf, (size_t)n);
            owned = s; data = s->data(); size = s->size(); ok = true;
        }
        close(fd);
    }
    ~MappedFile() { if (owned) delete owned; else if (data) munmap((void*)data, size); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
private:
    std::string* owned = nullptr;
};
}  // namespace

bool OBJLoader::scan_float_token(const char* begin, const char* end, float& out) { return scan_float(begin, end, out); }

// One pass over the mapped file gathers the `v` / `vt` records and the extents of the `f` lines, a second pass over those
// lines builds the fan triangles -- the reference reads the whole file twice for the same reason (OBJLoader.hpp:36-88 then
// :90-171): a face may name a vertex that is defined after it.  Files of more than a megabyte are cut at line ends into
// one piece per thread for both passes (records and triangles are concatenated in file order, so the result -- and the
// first error reported -- is the sequential one).
namespace {
struct ObjFace { const char* begin; const char* end; int nv, nt; };     // the line after "f", and the #v / #vt records before it
struct ObjPiece {
    const char *begin = nullptr, *end = nullptr;
    std::vector<float3> vertices;
    std::vector<float2> tex_coords;
    std::vector<ObjFace> faces;                    // nv / nt count this piece's records only until the pieces are joined
    std::string error;
};

// the next `n` whitespace-separated tokens of the line as floats; p moves past them.  false: a token is missing or not a number
inline bool line_floats(const char*& p, const char* end, float* out, int n)
{
    for (int k = 0; k < n; k++) {
        while (p < end && *p != '\n' && is_space((unsigned char)*p)) p++;
        if (p >= end || *p == '\n') return false;
        const char* tb = p;
        while (p < end && !is_space((unsigned char)*p)) p++;
        if (!scan_float(tb, p, out[k])) return false;
    }
    return true;
}

void obj_pass1(ObjPiece& pc)
{
    const size_t bytes = (size_t)(pc.end - pc.begin);
    pc.vertices.reserve(bytes / 96); pc.tex_coords.reserve(bytes / 96); pc.faces.reserve(bytes / 56);
    const char* const end = pc.end;
    for (const char* p = pc.begin; p < end;) {
        while (p < end && *p != '\n' && is_space((unsigned char)*p)) p++;          // leading blanks of the line
        // the record type is the first token: "v", "vt", "f" (anything else, "vn" included, is skipped: OBJLoader.hpp:55-62)
        if (p + 1 < end && p[0] == 'v' && is_space((unsigned char)p[1])) {
            p += 1;
            float c[3];
            if (!line_floats(p, end, c, 3)) { pc.error = "malformed v record"; return; }
            pc.vertices.push_back(make_float3(c[0], c[1], c[2]));
        } else if (p + 2 < end && p[0] == 'v' && p[1] == 't' && is_space((unsigned char)p[2])) {
            p += 2;
            float c[2];
            if (!line_floats(p, end, c, 2)) { pc.error = "malformed vt record"; return; }
            pc.tex_coords.push_back(make_float2(c[0], c[1]));
        } else if (p < end && p[0] == 'f' && (p + 1 == end || is_space((unsigned char)p[1]))) {
            const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!eol) eol = end;
            pc.faces.push_back(ObjFace{p + 1, eol, (int)pc.vertices.size(), (int)pc.tex_coords.size()});
            p = eol;
        } else if (p + 1 == end && p[0] == 'v') { pc.error = "malformed v record"; return; }
        else if (p + 2 == end && p[0] == 'v' && p[1] == 't') { pc.error = "malformed vt record"; return; }
        const char* eol = p < end ? (const char*)memchr(p, '\n', (size_t)(end - p)) : nullptr;
        p = eol ? eol + 1 : end;
    }
}

// A vector of tens of megabytes that is about to be filled: ask for transparent huge pages for its (page-aligned) inside, so that
// filling it takes a handful of page faults instead of one per 4 KB -- the first touch of fresh memory was half of the second
// pass's time, and page faults of several threads of one process queue on the same lock.  A hint: ignored where THP is off.
template <class T>
void advise_huge_pages(std::vector<T>& v)
{
    const size_t bytes = v.capacity() * sizeof(T);
    if (bytes < ((size_t)4 << 20)) return;
    const uintptr_t a = ((uintptr_t)v.data() + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)v.data() + bytes) & ~(uintptr_t)4095;
    if (b > a) (void)madvise((void*)a, (size_t)(b - a), MADV_HUGEPAGE);
}

// faces [f0, f1) -> fan triangles (OBJLoader.hpp:90-171)
void obj_pass2(const std::vector<ObjFace>& faces, size_t f0, size_t f1, const std::vector<float3>& vertices,
               const std::vector<float2>& tex_coords, bool lenient, std::vector<TrianglePrimitive>& triangles, std::string& error)
{
    triangles.reserve(triangles.size() + (f1 - f0) + (f1 - f0) / 8);
    advise_huge_pages(triangles);
    std::vector<int> vi, ti;
    const int nv = (int)vertices.size(), nt = (int)tex_coords.size();
    for (size_t fi = f0; fi < f1; fi++) {
        const ObjFace& fl = faces[fi];
        vi.clear(); ti.clear();
        const char* p = fl.begin;
        const char* const lend = fl.end;
        for (;;) {
            while (p < lend && is_space((unsigned char)*p)) p++;
            if (p >= lend) break;
            // one token v, v/vt or v/vt/vn, every character visited once: each number is read with std::stoi's prefix rule
            // (digits, then anything up to the next '/' or the end of the token is ignored), the slashes found as
            // string::find finds them (OBJLoader.hpp:104-122)
            int v;
            const char* d = p;
            if (!scan_int_prefix(p, lend, v) || p == d) { error = "malformed face token"; return; }
            vi.push_back(lenient && v < 0 ? fl.nv + v : v - 1);
            while (p < lend && *p != '/' && !is_space((unsigned char)*p)) p++;
            if (p < lend && *p == '/') {
                p++;
                const bool no_tex = lenient && p < lend && *p == '/';                   // v//vn
                if (!no_tex) {
                    if (!scan_int_prefix(p, lend, v)) { error = "malformed face token (v//vn is not supported)"; return; }
                    ti.push_back(lenient && v < 0 ? fl.nt + v : v - 1);
                }
                while (p < lend && *p != '/' && !is_space((unsigned char)*p)) p++;
                if (p < lend && *p == '/') {
                    p++;
                    if (!scan_int_prefix(p, lend, v)) { error = "malformed face token"; return; }
                    while (p < lend && !is_space((unsigned char)*p)) p++;
                }
            }
        }
        for (size_t i = 1; i + 1 < vi.size(); i++) {
            const int ia = vi[0], ib = vi[i], ic = vi[i + 1];
            if (ia < 0 || ib < 0 || ic < 0 || ia >= nv || ib >= nv || ic >= nv) { error = "face vertex index out of range"; return; }
            const float3 &A = vertices[ia], &B = vertices[ib], &C = vertices[ic];
            float3 normal = normalize(cross(TrianglePrimitive::sub(B, A), TrianglePrimitive::sub(C, A)));   // :141-143
            if (!ti.empty()) {
                if (ti.size() <= i + 1) { error = "face mixes v and v/vt tokens"; return; }
                const int ta = ti[0], tb2 = ti[i], tc = ti[i + 1];
                if (ta < 0 || tb2 < 0 || tc < 0 || ta >= nt || tb2 >= nt || tc >= nt) { error = "face texture index out of range"; return; }
                triangles.push_back(TrianglePrimitive(A, B, C, normal, tex_coords[ta], tex_coords[tb2], tex_coords[tc]));
            } else {
                triangles.push_back(TrianglePrimitive(A, B, C, normal));
            }
        }
    }
}

template <class F>
void run_pieces(int threads, int n, F&& body)       // body(k) for k in [0, n): `threads` workers (the caller is one) take the next k each
{
    std::atomic<int> next(0);
    auto work = [&] { for (int k; (k = next.fetch_add(1)) < n;) body(k); };
    std::vector<std::thread> th;
    for (int t = 1; t < threads && t < n; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}
}  // namespace

bool OBJLoader::parse(const std::string& fp, std::vector<TrianglePrimitive>& triangles, std::string* error, bool lenient)
{
    auto fail = [&](const std::string& msg) { if (error) *error = msg; return false; };
    const bool dbg = getenv("RT_OBJ_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_last = t_begin;
    long f_last = 0;
    auto faults = [] { struct rusage ru; getrusage(RUSAGE_SELF, &ru); return ru.ru_minflt; };
    if (dbg) f_last = faults();
    auto lap = [&](const char* what) { if (dbg) { const double t = now(); const long f = faults(); fprintf(stderr, "  obj parse: %-28s %7.2f ms  %6ld page faults\n", what, t - t_last, f - f_last); t_last = t; f_last = f; } };
    MappedFile file(fp.c_str());
    if (!file.ok) return fail("Could not open file " + fp);
    lap("map");
    const char* const data = file.data;
    const char* const fend = data + file.size;
    int threads = 1;
    if (file.size > ((size_t)1 << 20)) {
        threads = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)8, file.size >> 19});
        if (const char* e = getenv("RT_OBJ_THREADS")) threads = std::max(1, std::min(64, atoi(e)));
    }
    // pieces end at line ends; four per thread, handed out as threads finish (the `v` half of a file costs more per byte
    // than the `f` half in the first pass)
    const int npieces = threads == 1 ? 1 : threads * 4;
    std::vector<ObjPiece> pieces((size_t)npieces);
    {
        const char* at = data;
        for (int k = 0; k < npieces; k++) {
            pieces[(size_t)k].begin = at;
            const char* cut = k + 1 == npieces ? fend : data + file.size * (size_t)(k + 1) / (size_t)npieces;
            if (cut < at) cut = at;
            if (cut < fend) { const char* nl = (const char*)memchr(cut, '\n', (size_t)(fend - cut)); cut = nl ? nl + 1 : fend; }
            pieces[(size_t)k].end = at = cut;
        }
    }
    // pass 1: v / vt records (vn is accepted and unused, OBJLoader.hpp:55-62); remember face lines
    // (an exception must not leave a worker thread: out of memory becomes this piece's error)
    // (a worker fills a piece of its OWN and moves it into the shared array at the end: the headers of neighbouring pieces'
    // vectors -- whose end pointers every push_back moves -- share cache lines, and with two threads on neighbouring pieces
    // that false sharing made the passes SLOWER than on one thread)
    run_pieces(threads, npieces, [&](int k) {
        ObjPiece mine;
        mine.begin = pieces[(size_t)k].begin; mine.end = pieces[(size_t)k].end;
        try { obj_pass1(mine); } catch (const std::exception&) { mine.error = "out of memory while reading the file"; }
        pieces[(size_t)k] = std::move(mine);
    });
    for (const ObjPiece& pc : pieces) if (!pc.error.empty()) return fail(pc.error);
    lap("pass 1 (records)");
    std::vector<float3> vertices;
    std::vector<float2> tex_coords;
    std::vector<ObjFace> faces;
    if (threads == 1) {
        vertices.swap(pieces[0].vertices); tex_coords.swap(pieces[0].tex_coords); faces.swap(pieces[0].faces);
    } else {
        // the pieces' records joined in file order -- by all threads: every piece knows where its records start once the counts
        // are summed (the join and the concatenation of the triangles below were the serial third of a large file's parse)
        std::vector<size_t> bv((size_t)npieces + 1, 0), bt((size_t)npieces + 1, 0), bf((size_t)npieces + 1, 0);
        for (int k = 0; k < npieces; k++) {
            bv[(size_t)k + 1] = bv[(size_t)k] + pieces[(size_t)k].vertices.size();
            bt[(size_t)k + 1] = bt[(size_t)k] + pieces[(size_t)k].tex_coords.size();
            bf[(size_t)k + 1] = bf[(size_t)k] + pieces[(size_t)k].faces.size();
        }
        vertices.resize(bv.back()); tex_coords.resize(bt.back()); faces.resize(bf.back());
        run_pieces(threads, npieces, [&](int k) {
            ObjPiece& pc = pieces[(size_t)k];
            for (ObjFace& f : pc.faces) { f.nv += (int)bv[(size_t)k]; f.nt += (int)bt[(size_t)k]; }
            if (!pc.vertices.empty()) memcpy(&vertices[bv[(size_t)k]], pc.vertices.data(), pc.vertices.size() * sizeof(float3));
            if (!pc.tex_coords.empty()) memcpy(&tex_coords[bt[(size_t)k]], pc.tex_coords.data(), pc.tex_coords.size() * sizeof(float2));
            if (!pc.faces.empty()) memcpy(&faces[bf[(size_t)k]], pc.faces.data(), pc.faces.size() * sizeof(ObjFace));
            std::vector<float3>().swap(pc.vertices); std::vector<float2>().swap(pc.tex_coords); std::vector<ObjFace>().swap(pc.faces);
        });
    }
    lap("join records");
    // pass 2
    if (threads == 1) {
        std::string err;
        obj_pass2(faces, 0, faces.size(), vertices, tex_coords, lenient, triangles, err);
        return err.empty() ? true : fail(err);
    }
    std::vector<std::vector<TrianglePrimitive>> part((size_t)npieces);
    std::vector<std::string> errs((size_t)npieces);
    run_pieces(threads, npieces, [&](int k) {
        std::vector<TrianglePrimitive> mine;
        std::string err;
        try {
            obj_pass2(faces, faces.size() * (size_t)k / (size_t)npieces, faces.size() * (size_t)(k + 1) / (size_t)npieces, vertices, tex_coords,
                      lenient, mine, err);
        } catch (const std::exception&) { err = "out of memory while building the triangles"; }
        part[(size_t)k] = std::move(mine);
        errs[(size_t)k] = std::move(err);
    });
    for (const std::string& e : errs) if (!e.empty()) return fail(e);
    lap("pass 2 (triangles)");
    std::vector<size_t> at((size_t)npieces + 1, triangles.size());
    for (int k = 0; k < npieces; k++) at[(size_t)k + 1] = at[(size_t)k] + part[(size_t)k].size();
    triangles.reserve(at.back());
    advise_huge_pages(triangles);
    triangles.resize(at.back());
    static_assert(std::is_trivially_copyable<TrianglePrimitive>::value, "the parts are placed with memcpy");
    run_pieces(threads, npieces, [&](int k) {
        if (!part[(size_t)k].empty()) memcpy((void*)&triangles[at[(size_t)k]], (const void*)part[(size_t)k].data(), part[(size_t)k].size() * sizeof(TrianglePrimitive));
        std::vector<TrianglePrimitive>().swap(part[(size_t)k]);
    });
    lap("concatenate");
    return true;
}

MeshPrimitive OBJLoader::load_for_device(std::string fp, bool lenient)
{
    std::vector<TrianglePrimitive> triangles;
    std::string err;
    if (!parse(fp, triangles, &err, lenient)) throw std::runtime_error("OBJLoader: " + err);
    return MeshPrimitive::for_device_build(std::move(triangles));
}

MeshPrimitive OBJLoader::load_lenient(std::string fp)
{
    std::vector<TrianglePrimitive> triangles;
    std::string err;
    if (!parse(fp, triangles, &err, true)) throw std::runtime_error("OBJLoader: " + err);
    return MeshPrimitive(std::move(triangles));
}

MeshPrimitive OBJLoader::load(std::string fp)
{
    std::cout << "Loading OBJ file: " << fp << std::endl;
    std::vector<TrianglePrimitive> triangles;
    std::string err;
    if (!parse(fp, triangles, &err)) {
        if (err.rfind("Could not open file", 0) == 0) { std::cout << err << std::endl; exit(1); }
        throw std::runtime_error("OBJLoader: " + err);
    }
    std::cout << "OBJ File: " << fp << std::endl;
    std::cout << "Loaded " << triangles.size() << " triangles" << std::endl;
    return MeshPrimitive(std::move(triangles));
}
