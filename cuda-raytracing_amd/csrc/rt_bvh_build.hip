// rt_bvh_build.hip -- GPU build of the reference's BVH.
//
// Produces, node for node, the tree of BVHTree::fill (BVHTree.hpp:203-292): 5 candidate planes per axis at
// (s+1)/6 of the node extent, centroid partition, cost = half-area x count, strict-less axis choice with ties to z,
// "no split if not cheaper" / "no split if a side is empty" / depth and count limits, and the reference's pre-order
// node numbering -- so a mesh built here is interchangeable with one built by the host builder (csrc/host).
//
// The reference recurses node by node on the CPU (1.7 s for 70 k triangles); here the tree grows one LEVEL per step,
// all nodes of the level in parallel:
//   bounds   : one thread per triangle, float atomic min/max into its node's box
//   bins     : one thread per triangle and axis, atomics into 6 bins x 3 axes (box + count) -- a plane's left side is
//              the union of the bins below it, exactly the partition `centroid <= pos` of BVHTree.hpp:339
//   decide   : one thread per node evaluates the 15 costs with the host builder's fp32 operations, picks axis/plane,
//              and creates the two children (their sizes follow from the bin counts)
//   partition: one device-wide exclusive scan of the "goes left" flags gives every triangle its stable rank
// and a final pass converts the breadth-first node order into the reference's depth-first numbering.
// min/max are exact and order-independent, so the result does not depend on scheduling.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <cstdint>
#include <vector>

#include "../../include/rt_hip.h"

#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = (int)e_; goto done; } } while (0)

namespace {

struct BuildNode {
    float mn[3], mx[3];
    int32_t first, count, depth;
    int32_t child_a, child_b;       // breadth-first indices, -1 = leaf
    int32_t axis;                   // split axis (valid while splitting)
    float split_pos;
    int32_t nl;                     // triangles going left
    int32_t size;                   // subtree size (nodes), for the pre-order numbering
    int32_t pre;                    // pre-order index
};

struct Bins {                       // per node: 3 axes x 6 bins
    float mn[3][6][3], mx[3][6][3];
    int32_t cnt[3][6];
};

// Float min/max through integer atomics: non-negative floats order like ints, negative ones like reversed unsigned
// ints.  -0.0 has the bit pattern of INT_MIN and would break both orders, so zeros are canonicalised to +0.0 first
// (x + 0.0f); a box may therefore hold +0.0 where a sequential fminf/fmaxf fold would hold -0.0 -- equal by value,
// and no consumer of the boxes (costs, planes, slab tests) can tell the two apart.
__device__ __forceinline__ void atomic_min_f(float* a, float v)
{
    v = v + 0.0f;
    if (v >= 0.0f) atomicMin((int*)a, __float_as_int(v)); else atomicMax((unsigned int*)a, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* a, float v)
{
    v = v + 0.0f;
    if (v >= 0.0f) atomicMax((int*)a, __float_as_int(v)); else atomicMin((unsigned int*)a, __float_as_uint(v));
}

// per-triangle centroid (TrianglePrimitive::center, TrianglePrimitive.hpp:81-83) and box
__global__ void prep_kernel(const float* __restrict__ v, int n, float* __restrict__ centroid, float* __restrict__ tbox,
                            int32_t* __restrict__ order, int32_t* __restrict__ node_of)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* t = v + 9 * (size_t)i;
    for (int k = 0; k < 3; k++) {
        centroid[3 * (size_t)i + k] = ((t[k] + t[3 + k]) + t[6 + k]) / 3.0f;
        tbox[6 * (size_t)i + k] = fminf(fminf(fminf(FLT_MAX, t[k]), t[3 + k]), t[6 + k]);
        tbox[6 * (size_t)i + 3 + k] = fmaxf(fmaxf(fmaxf(-FLT_MAX, t[k]), t[3 + k]), t[6 + k]);
    }
    order[i] = i;
    node_of[i] = 0;
}

__global__ void init_level_kernel(BuildNode* nodes, Bins* bins, int level_begin, int level_end)
{
    int k = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= level_end) return;
    BuildNode& nd = nodes[k];
    for (int c = 0; c < 3; c++) { nd.mn[c] = FLT_MAX; nd.mx[c] = -FLT_MAX; }
    nd.child_a = nd.child_b = -1;
    Bins& b = bins[k - level_begin];
    for (int a = 0; a < 3; a++)
        for (int s = 0; s < 6; s++) {
            for (int c = 0; c < 3; c++) { b.mn[a][s][c] = FLT_MAX; b.mx[a][s][c] = -FLT_MAX; }
            b.cnt[a][s] = 0;
        }
}

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// BVHTree.hpp:206-209: grow the node's box over its triangles.  Near the root every lane of a wave belongs to the same
// node (ranges are contiguous), so the wave reduces first and issues 6 atomics instead of 384 on one address.
__global__ void bounds_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                              const float* __restrict__ tbox, BuildNode* nodes, int level_begin)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    int k = p < n ? node_of[p] : -1;
    const bool valid = k >= level_begin;                         // else: past the end, or its node was finished earlier
    const unsigned long long vm = __ballot(valid);
    if (vm == 0) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (valid) {
        const float* tb = tbox + 6 * (size_t)order[p];
        for (int c = 0; c < 3; c++) { lo[c] = tb[c]; hi[c] = tb[3 + c]; }
    }
    const int first = __ffsll((long long)vm) - 1;
    const int k0 = __shfl(k, first);
    if (__ballot(valid && k != k0) == 0) {                       // one node for the whole wave
        for (int c = 0; c < 3; c++) { lo[c] = wave_min(lo[c]); hi[c] = wave_max(hi[c]); }
        if ((int)(threadIdx.x & 63) == first) {
            BuildNode& nd = nodes[k0];
            for (int c = 0; c < 3; c++) { atomic_min_f(&nd.mn[c], lo[c]); atomic_max_f(&nd.mx[c], hi[c]); }
        }
    } else if (valid) {
        BuildNode& nd = nodes[k];
        for (int c = 0; c < 3; c++) { atomic_min_f(&nd.mn[c], lo[c]); atomic_max_f(&nd.mx[c], hi[c]); }
    }
}

__device__ __forceinline__ float plane_pos(float mn, float mx, int s)
{
    float split_t = ((float)s + 1) / (5.0f + 1);                 // BVHTree.hpp:303
    return mn + (mx - mn) * (split_t);                           // BVHTree.hpp:318
}

// evaluate_split's partition (BVHTree.hpp:324-348), binned: bin = number of planes the centroid lies beyond.
// Same wave-level pre-reduction as bounds_kernel when the whole wave works on one node.
__global__ void bins_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                            const float* __restrict__ centroid, const float* __restrict__ tbox,
                            const BuildNode* __restrict__ nodes, Bins* bins, int level_begin, int max_depth)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    int k = p < n ? node_of[p] : -1;
    bool valid = k >= level_begin;
    if (valid) {
        const BuildNode& nd = nodes[k];
        valid = !(nd.depth >= max_depth || nd.count <= 1);       // BVHTree.hpp:211-215: no split evaluated
    }
    const unsigned long long vm = __ballot(valid);
    if (vm == 0) return;
    const int first = __ffsll((long long)vm) - 1;
    const int k0 = __shfl(k, first);
    const bool uniform = __ballot(valid && k != k0) == 0;
    const int t = valid ? order[p] : 0;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (valid) { const float* tb = tbox + 6 * (size_t)t; for (int q = 0; q < 3; q++) { lo[q] = tb[q]; hi[q] = tb[3 + q]; } }
    const BuildNode& nd = nodes[valid ? k : k0];
    Bins& b = bins[(valid ? k : k0) - level_begin];
    for (int a = 0; a < 3; a++) {
        int s = 0;
        if (valid) {
            const float c = centroid[3 * (size_t)t + a];
            while (s < 5 && !(c <= plane_pos(nd.mn[a], nd.mx[a], s))) s++;
        }
        if (uniform) {
            for (int bin = 0; bin < 6; bin++) {
                const unsigned long long m = __ballot(valid && s == bin);
                if (m == 0) continue;
                const bool mine = valid && s == bin;
                float rl[3], rh[3];
                for (int q = 0; q < 3; q++) { rl[q] = wave_min(mine ? lo[q] : FLT_MAX); rh[q] = wave_max(mine ? hi[q] : -FLT_MAX); }
                if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
                    for (int q = 0; q < 3; q++) { atomic_min_f(&b.mn[a][bin][q], rl[q]); atomic_max_f(&b.mx[a][bin][q], rh[q]); }
                    atomicAdd(&b.cnt[a][bin], __popcll(m));
                }
            }
        } else if (valid) {
            for (int q = 0; q < 3; q++) { atomic_min_f(&b.mn[a][s][q], lo[q]); atomic_max_f(&b.mx[a][s][q], hi[q]); }
            atomicAdd(&b.cnt[a][s], 1);
        }
    }
}

// BVHTree::cost, BVHTree.hpp:192-201
__device__ __forceinline__ float box_cost(const float* mn, const float* mx, int count)
{
    if (count == 0) return FLT_MAX;
    float sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
    float half_area = sx * (sy + sz) + sy * sz;
    return half_area * (float)count;
}

// BVHTree.hpp:218-289 for every node of the level; children are appended to the node array
__global__ void decide_kernel(BuildNode* nodes, const Bins* bins, int level_begin, int level_end, int max_depth,
                              int32_t* node_counter)
{
    int k = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= level_end) return;
    BuildNode& nd = nodes[k];
    if (nd.depth >= max_depth || nd.count <= 1) return;
    const Bins& b = bins[k - level_begin];
    float eval_cost[3], eval_split[3];
    int eval_nl[3];
    for (int a = 0; a < 3; a++) {
        float best_cost = FLT_MAX, best_split = 0.0f;
        int best_nl = 0;
        for (int s = 0; s < 5; s++) {
            float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            float rmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            int ln = 0, rn = 0;
            for (int q = 0; q < 6; q++) {
                float* mn = q <= s ? lmn : rmn;
                float* mx = q <= s ? lmx : rmx;
                for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], b.mn[a][q][c]); mx[c] = fmaxf(mx[c], b.mx[a][q][c]); }
                if (q <= s) ln += b.cnt[a][q]; else rn += b.cnt[a][q];
            }
            float cost = box_cost(lmn, lmx, ln) + box_cost(rmn, rmx, rn);          // BVHTree.hpp:351
            if (cost < best_cost) { best_cost = cost; best_split = plane_pos(nd.mn[a], nd.mx[a], s); best_nl = ln; }
        }
        eval_cost[a] = best_cost; eval_split[a] = best_split; eval_nl[a] = best_nl;
    }
    int axis;                                                    // BVHTree.hpp:229-243
    if (eval_cost[0] < eval_cost[1] && eval_cost[0] < eval_cost[2]) axis = 0;
    else if (eval_cost[1] < eval_cost[0] && eval_cost[1] < eval_cost[2]) axis = 1;
    else axis = 2;
    if (eval_cost[axis] >= box_cost(nd.mn, nd.mx, nd.count)) return;                // BVHTree.hpp:246
    const int nl = eval_nl[axis], nr = nd.count - nl;
    if (nl == 0 || nr == 0) return;                              // BVHTree.hpp:279
    const int a = atomicAdd(node_counter, 2);
    nd.axis = axis; nd.split_pos = eval_split[axis]; nd.nl = nl;
    nd.child_a = a; nd.child_b = a + 1;
    BuildNode& ca = nodes[a];
    BuildNode& cb = nodes[a + 1];
    ca.first = nd.first; ca.count = nl; ca.depth = nd.depth + 1;
    cb.first = nd.first + nl; cb.count = nr; cb.depth = nd.depth + 1;
}

// "goes left" flag of every position whose node splits (BVHTree.hpp:253-277)
__global__ void flags_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                             const float* __restrict__ centroid, const BuildNode* __restrict__ nodes, int level_begin,
                             int32_t* __restrict__ flags)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int f = 0;
    const int k = node_of[p];
    if (k >= level_begin) {
        const BuildNode& nd = nodes[k];
        if (nd.child_a >= 0) f = centroid[3 * (size_t)order[p] + nd.axis] <= nd.split_pos ? 1 : 0;
    }
    flags[p] = f;
}

// stable partition: left triangles keep their order at the front of the node's range, right ones behind them
__global__ void scatter_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                               const BuildNode* __restrict__ nodes, int level_begin, const int32_t* __restrict__ flags,
                               const int32_t* __restrict__ scan, int32_t* __restrict__ order_out, int32_t* __restrict__ node_of_out)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int k = node_of[p];
    int q = p, child = k;
    if (k >= level_begin) {
        const BuildNode& nd = nodes[k];
        if (nd.child_a >= 0) {
            const int lrank = scan[p] - scan[nd.first];          // left-going triangles before p inside the node
            if (flags[p]) { q = nd.first + lrank; child = nd.child_a; }
            else { q = nd.first + nd.nl + (p - nd.first - lrank); child = nd.child_b; }
        } else {
            child = -1 - k;                                      // finished leaf: never looked at again (negative < level_begin)
        }
    }
    order_out[q] = order[p];
    node_of_out[q] = child;
}

__global__ void size_kernel(BuildNode* nodes, int level_begin, int level_end)
{
    int k = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= level_end) return;
    BuildNode& nd = nodes[k];
    nd.size = nd.child_a >= 0 ? 1 + nodes[nd.child_a].size + nodes[nd.child_b].size : 1;
}

// BVHTree.hpp:283-289: child a is numbered right after its parent, child b after a's whole subtree
__global__ void preorder_kernel(BuildNode* nodes, int level_begin, int level_end)
{
    int k = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= level_end) return;
    BuildNode& nd = nodes[k];
    if (nd.child_a >= 0) {
        nodes[nd.child_a].pre = nd.pre + 1;
        nodes[nd.child_b].pre = nd.pre + 1 + nodes[nd.child_a].size;
    }
}

__global__ void emit_kernel(const BuildNode* __restrict__ nodes, int num_nodes, float* __restrict__ bounds,
                            int32_t* __restrict__ children, int32_t* __restrict__ leaf_first, int32_t* __restrict__ leaf_count)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_nodes) return;
    const BuildNode& nd = nodes[k];
    const size_t o = (size_t)nd.pre;
    for (int c = 0; c < 3; c++) { bounds[6 * o + c] = nd.mn[c]; bounds[6 * o + 3 + c] = nd.mx[c]; }
    const bool leaf = nd.child_a < 0;
    children[2 * o] = leaf ? -1 : nodes[nd.child_a].pre;
    children[2 * o + 1] = leaf ? -1 : nodes[nd.child_b].pre;
    leaf_first[o] = nd.first;
    leaf_count[o] = leaf ? nd.count : 0;
}

}  // namespace

extern "C" int rt_bvh_build(const float* vertices, int32_t n, int32_t max_depth, float* node_bounds, int32_t* node_children,
                            int32_t* node_leaf_first, int32_t* node_leaf_count, int32_t* leaf_indices, int32_t* num_nodes,
                            int32_t* num_levels)
{
    if (n < 0 || max_depth < 1 || (n > 0 && !vertices) || !node_bounds || !node_children || !node_leaf_first ||
        !node_leaf_count || !num_nodes || (n > 0 && !leaf_indices)) return RT_E_INVALID;
    int rc = RT_OK;
    const int cap = n > 0 ? 2 * n : 1;                           // <= 2n - 1 nodes
    const int T = 256;
    const int gridN = (n + T - 1) / T;
    float *d_v = nullptr, *d_centroid = nullptr, *d_tbox = nullptr, *d_bounds = nullptr;
    int32_t *d_order[2] = {nullptr, nullptr}, *d_nodeof[2] = {nullptr, nullptr}, *d_flags = nullptr, *d_scan = nullptr;
    int32_t *d_counter = nullptr, *d_children = nullptr, *d_lfirst = nullptr, *d_lcount = nullptr;
    BuildNode* d_nodes = nullptr;
    Bins* d_bins = nullptr;
    void* d_tmp = nullptr;
    char* arena = nullptr;
    size_t tmp_bytes = 0;
    std::vector<int> level_begin;
    int cur = 0, total = 1, levels = 0;
    BuildNode root;

    // one device allocation for everything (a dozen hipMalloc / hipFree pairs cost more than the build kernels of a
    // 70 k-triangle mesh): sizes are known up front because a tree over n triangles has at most 2n - 1 nodes
    {
        RT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flags, d_scan, n > 0 ? n : 1));
        const size_t n1 = n > 0 ? (size_t)n : 1;
        size_t off = 0;
        auto take = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_v = take(n1 * 9 * 4), o_cen = take(n1 * 3 * 4), o_tbox = take(n1 * 6 * 4), o_ord0 = take(n1 * 4), o_ord1 = take(n1 * 4),
                     o_nof0 = take(n1 * 4), o_nof1 = take(n1 * 4), o_flags = take(n1 * 4), o_scan = take(n1 * 4), o_counter = take(4),
                     o_nodes = take(((size_t)cap + 2) * sizeof(BuildNode)), o_bins = take(n1 * sizeof(Bins)), o_tmp = take(tmp_bytes ? tmp_bytes : 1),
                     o_bounds = take((size_t)cap * 6 * 4), o_children = take((size_t)cap * 2 * 4), o_lfirst = take((size_t)cap * 4),
                     o_lcount = take((size_t)cap * 4);
        RT_HIP(hipMalloc((void**)&arena, off));
        d_v = (float*)(arena + o_v); d_centroid = (float*)(arena + o_cen); d_tbox = (float*)(arena + o_tbox);
        d_order[0] = (int32_t*)(arena + o_ord0); d_order[1] = (int32_t*)(arena + o_ord1);
        d_nodeof[0] = (int32_t*)(arena + o_nof0); d_nodeof[1] = (int32_t*)(arena + o_nof1);
        d_flags = (int32_t*)(arena + o_flags); d_scan = (int32_t*)(arena + o_scan); d_counter = (int32_t*)(arena + o_counter);
        d_nodes = (BuildNode*)(arena + o_nodes); d_bins = (Bins*)(arena + o_bins); d_tmp = arena + o_tmp;
        d_bounds = (float*)(arena + o_bounds); d_children = (int32_t*)(arena + o_children);
        d_lfirst = (int32_t*)(arena + o_lfirst); d_lcount = (int32_t*)(arena + o_lcount);
    }

    if (n > 0) {
        RT_HIP(hipMemcpy(d_v, vertices, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(prep_kernel, dim3(gridN), dim3(T), 0, 0, d_v, n, d_centroid, d_tbox, d_order[0], d_nodeof[0]);
    }
    memset(&root, 0, sizeof root);
    root.first = 0; root.count = n; root.depth = 1; root.child_a = root.child_b = -1;      // fill(1, max_depth), MeshPrimitive.cpp:54
    RT_HIP(hipMemcpy(d_nodes, &root, sizeof root, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(d_counter, &total, sizeof(int), hipMemcpyHostToDevice));

    // ---- one level per iteration ----
    {
        int lb = 0, le = 1;
        while (lb < le) {
            level_begin.push_back(lb);
            levels++;
            const int gridL = (le - lb + T - 1) / T;
            hipLaunchKernelGGL(init_level_kernel, dim3(gridL), dim3(T), 0, 0, d_nodes, d_bins, lb, le);
            if (n > 0) {
                hipLaunchKernelGGL(bounds_kernel, dim3(gridN), dim3(T), 0, 0, d_order[cur], d_nodeof[cur], n, d_tbox, d_nodes, lb);
                hipLaunchKernelGGL(bins_kernel, dim3(gridN), dim3(T), 0, 0, d_order[cur], d_nodeof[cur], n, d_centroid, d_tbox,
                                   d_nodes, d_bins, lb, max_depth);
            }
            hipLaunchKernelGGL(decide_kernel, dim3(gridL), dim3(T), 0, 0, d_nodes, d_bins, lb, le, max_depth, d_counter);
            int new_total = 0;
            RT_HIP(hipMemcpy(&new_total, d_counter, sizeof(int), hipMemcpyDeviceToHost));
            if (new_total > cap) { rc = RT_E_INVALID; goto done; }
            if (new_total > total && n > 0) {
                hipLaunchKernelGGL(flags_kernel, dim3(gridN), dim3(T), 0, 0, d_order[cur], d_nodeof[cur], n, d_centroid, d_nodes, lb, d_flags);
                RT_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flags, d_scan, n));
                hipLaunchKernelGGL(scatter_kernel, dim3(gridN), dim3(T), 0, 0, d_order[cur], d_nodeof[cur], n, d_nodes, lb, d_flags,
                                   d_scan, d_order[cur ^ 1], d_nodeof[cur ^ 1]);
                cur ^= 1;
            }
            lb = le; le = new_total; total = new_total;
        }
    }
    level_begin.push_back(total);
    // ---- breadth-first -> the reference's depth-first numbering ----
    for (int l = levels - 1; l >= 0; l--) {
        const int lb = level_begin[l], le = level_begin[l + 1];
        hipLaunchKernelGGL(size_kernel, dim3((le - lb + T - 1) / T), dim3(T), 0, 0, d_nodes, lb, le);
    }
    for (int l = 0; l < levels; l++) {                            // root.pre = 0 from the memset
        const int lb = level_begin[l], le = level_begin[l + 1];
        hipLaunchKernelGGL(preorder_kernel, dim3((le - lb + T - 1) / T), dim3(T), 0, 0, d_nodes, lb, le);
    }
    hipLaunchKernelGGL(emit_kernel, dim3((total + T - 1) / T), dim3(T), 0, 0, d_nodes, total, d_bounds, d_children, d_lfirst, d_lcount);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpy(node_bounds, d_bounds, (size_t)total * 6 * sizeof(float), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(node_children, d_children, (size_t)total * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(node_leaf_first, d_lfirst, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(node_leaf_count, d_lcount, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n > 0) RT_HIP(hipMemcpy(leaf_indices, d_order[cur], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    *num_nodes = total;
    if (num_levels) *num_levels = levels;

done:
    (void)hipFree(arena);
    return rc;
}
