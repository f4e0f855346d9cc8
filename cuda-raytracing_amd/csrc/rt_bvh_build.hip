// rt_bvh_build.hip -- GPU build of the reference's BVH.
//
// Produces, node for node, the tree of BVHTree::fill (BVHTree.hpp:203-292): 5 candidate planes per axis at
// (s+1)/6 of the node extent, centroid partition, cost = half-area x count, strict-less axis choice with ties to z,
// "no split if not cheaper" / "no split if a side is empty" / depth and count limits, and the reference's pre-order
// node numbering -- so a mesh built here is interchangeable with one built by the host builder (csrc/host).
//
// The reference recurses node by node on the CPU (1.7 s for 70 k triangles); here the tree grows one LEVEL per step,
// all nodes of the level in parallel, while nodes hold more than a wave of triangles; every subtree of at most 64
// triangles is then finished by one wave on its own (small_subtree_kernel), all of them in one launch.  Where a level
// starts and ends is device state, so the host enqueues levels without waiting and asks only from time to time whether any
// large node is left.  Per level (four launches up to 1 M triangles):
//   bins     : one thread per triangle, 6 bins x 3 axes (box + count) per node, collected in LDS per block and added to the
//              node's bins once -- a plane's left side is the union of the bins below it, exactly the partition
//              `centroid <= pos` of BVHTree.hpp:339
//   decide   : one thread per node evaluates the 15 costs with the host builder's fp32 operations, picks axis / plane and
//              creates the two children: sizes from the bin counts, BOXES from the bin boxes (the union of the boxes of
//              a set of triangles is the same whichever way it is folded, so no per-level bounds pass is needed)
//   partition: the "goes left" flags with their exclusive scan per 256-position block and the block totals in one launch,
//              then the scatter (each workgroup turns the totals into block offsets itself): every triangle's stable rank
// and one final pass turns breadth-first order into the reference's depth-first numbering without walking the levels:
// in pre-order a node is preceded by its ancestors and by every node whose triangle range lies entirely to its left, so
// pre(Y) = depth(Y) - 1 + #{X : end(X) <= first(Y)} -- a histogram of range ends and one more scan.
// min/max are exact and order-independent, so the result does not depend on scheduling.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_scene_internal.h"

#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = (int)e_; goto done; } } while (0)

namespace {

constexpr int kMaxLevels = 64;                                   // level table size (the depth limit of the reference is 32)

struct BuildNode {
    float mn[3], mx[3];
    int32_t first, count, depth;
    int32_t child_a, child_b;       // breadth-first indices, -1 = leaf
    int32_t axis;                   // split axis (valid while splitting)
    float split_pos;
    int32_t nl;                     // triangles going left
    int32_t bin;                    // index of this node's Bins in the level's bin array, -1 = the node evaluates no split
};

struct Bins {                       // per node: 3 axes x 6 bins
    float mn[3][6][3], mx[3][6][3];
    int32_t cnt[3][6];
};

// Everything the level loop needs to know lives on the device: level l is nodes [begin[l], begin[l + 1]).
struct BuildState {
    int32_t begin[kMaxLevels + 2];
    int32_t total;                  // nodes created so far (children are appended by decide_kernel)
    int32_t bins_used[2];           // bin slots handed out for the level being created (parity of that level)
    int32_t levels;                 // non-empty levels
    int32_t overflow;               // set when the node array or the level table would overflow (cannot happen for a valid cap)
    int32_t num_small;              // nodes of at most kSmall triangles handed to small_subtree_kernel (listed in small_roots[])
};

constexpr int kSmall = 64;                                       // a subtree of up to one wave of triangles is finished by one wave

// Float min/max through integer atomics: non-negative floats order like ints, negative ones like reversed unsigned
// ints.  -0.0 has the bit pattern of INT_MIN and would break both orders, so zeros are canonicalised to +0.0 first
// (x + 0.0f); a box may therefore hold +0.0 where a sequential fminf/fmaxf fold would hold -0.0 -- equal by value,
// and no consumer of the boxes (costs, planes, slab tests) can tell the two apart.  NaN is skipped, as fminf / fmaxf
// skip it in the host builder's fold (which starts from +-FLT_MAX, so a NaN never enters a box).
__device__ __forceinline__ void atomic_min_f(float* a, float v)
{
    if (!(v == v)) return;
    v = v + 0.0f;
    if (v >= 0.0f) atomicMin((int*)a, __float_as_int(v)); else atomicMax((unsigned int*)a, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* a, float v)
{
    if (!(v == v)) return;
    v = v + 0.0f;
    if (v >= 0.0f) atomicMax((int*)a, __float_as_int(v)); else atomicMin((unsigned int*)a, __float_as_uint(v));
}

__device__ __forceinline__ void clear_bins(Bins& b)
{
    for (int a = 0; a < 3; a++)
        for (int s = 0; s < 6; s++) {
            for (int c = 0; c < 3; c++) { b.mn[a][s][c] = FLT_MAX; b.mx[a][s][c] = -FLT_MAX; }
            b.cnt[a][s] = 0;
        }
}

// per-triangle centroid (TrianglePrimitive::center, TrianglePrimitive.hpp:81-83) and box; block 0 also sets up the root
__global__ void prep_kernel(const float* __restrict__ v, int n, int max_depth, float* __restrict__ centroid, float* __restrict__ tbox,
                            int32_t* __restrict__ order, int32_t* __restrict__ node_of, BuildNode* nodes, Bins* bins, BuildState* st,
                            int32_t* __restrict__ end_hist, int32_t* __restrict__ small_roots, int small_limit)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        BuildNode& r = nodes[0];                                 // fill(1, max_depth), MeshPrimitive.cpp:54
        for (int c = 0; c < 3; c++) { r.mn[c] = FLT_MAX; r.mx[c] = -FLT_MAX; }
        r.first = 0; r.count = n; r.depth = 1; r.child_a = r.child_b = -1; r.axis = 0; r.split_pos = 0.0f; r.nl = 0;
        r.bin = (1 >= max_depth || n <= 1) ? -1 : 0;             // BVHTree.hpp:211-215
        st->num_small = 0;
        if (r.bin == 0 && n <= small_limit) { r.bin = -1; small_roots[0] = 0; st->num_small = 1; }     // the whole mesh fits one wave
        if (r.bin == 0) clear_bins(bins[0]);
        for (int l = 0; l < kMaxLevels + 2; l++) st->begin[l] = l == 0 ? 0 : 1;
        st->total = 1; st->bins_used[0] = r.bin == 0 ? 1 : 0; st->bins_used[1] = 0; st->levels = 1; st->overflow = 0;
    }
    if (i <= n) end_hist[i] = 0;
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

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// BVHTree.hpp:206-209 for the root: grow its box over all triangles.  Few blocks with a grid-stride loop, a reduction
// per block, six atomics per block (one atomic per wave put a thousand waves in line for the same six words).
__global__ __launch_bounds__(1024) void root_bounds_kernel(int n, const float* __restrict__ tbox, BuildNode* nodes)
{
    __shared__ float red[16][6];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const float* tb = tbox + 6 * (size_t)p;
        for (int c = 0; c < 3; c++) { lo[c] = fminf(lo[c], tb[c]); hi[c] = fmaxf(hi[c], tb[3 + c]); }
    }
    for (int c = 0; c < 3; c++) { lo[c] = wave_min(lo[c]); hi[c] = wave_max(hi[c]); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int c = 0; c < 3; c++) { red[wave][c] = lo[c]; red[wave][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomic_min_f(&nodes[0].mn[threadIdx.x], v); else atomic_max_f(&nodes[0].mx[threadIdx.x - 3], v);
    }
}

__device__ __forceinline__ float plane_pos(float mn, float mx, int s)
{
    float split_t = ((float)s + 1) / (5.0f + 1);                 // BVHTree.hpp:303
    return mn + (mx - mn) * (split_t);                           // BVHTree.hpp:318
}

// order-preserving integer image of a float (what integer min / max atomics in LDS work on)
__device__ __forceinline__ int ordered_int(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float ordered_float(int v) { return __int_as_float(v >= 0 ? v : v ^ 0x7fffffff); }
constexpr int kOrderedMax = 0x7f7fffff, kOrderedLowest = (int)0x80800000;      // ordered_int(FLT_MAX), ordered_int(-FLT_MAX)

// evaluate_split's partition (BVHTree.hpp:324-348), binned: bin = number of planes the centroid lies beyond.
// A block of 256 positions covers a handful of splitting nodes at most (ranges are contiguous, and nodes of up to kSmall
// triangles have left the loop), so it bins into LDS -- one set of bins per node it touches -- and adds the non-empty bins
// to the nodes' bins in global memory once: a tenth of the global atomics of one-atomic-per-triangle, which is what bounded
// the middle levels (all of a node's triangles queue on the same 126 words).  A block that touches more than kBlockNodes
// splitting nodes (possible only with RT_BVH_SMALL below 32) falls back to one global atomic per triangle and word.
constexpr int kBlockNodes = 8;
__global__ __launch_bounds__(256) void bins_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                                                   const float* __restrict__ centroid, const float* __restrict__ tbox,
                                                   const BuildNode* __restrict__ nodes, Bins* bins, const BuildState* __restrict__ st, int level)
{
    const int level_begin = st->begin[level];
    if (level_begin >= st->begin[level + 1]) return;             // past the last level
    __shared__ int sbin[kBlockNodes][126];                       // per node, axis and bin: min xyz, max xyz (ordered-int images), count
    __shared__ int slot_bin[kBlockNodes];                        // the node's Bins index
    __shared__ int wave_heads[4];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = p < n ? node_of[p] : -1;
    bool valid = k >= level_begin;                               // else: past the end, or its node was finished earlier
    int bin_index = -1;
    if (valid) { bin_index = nodes[k].bin; valid = bin_index >= 0; }
    // a position starts a new node of the block when its node differs from its left neighbour's (or it is the block's first)
    const int k_left = (threadIdx.x > 0 && p < n) ? node_of[p - 1] : -2;
    const bool head = valid && k != k_left;
    const unsigned long long m_head = __ballot(head);
    if (lane == 0) wave_heads[wave] = __popcll(m_head);
    __syncthreads();
    int heads_before = 0, nslots = 0;
    for (int w = 0; w < 4; w++) { const int c = wave_heads[w]; if (w < wave) heads_before += c; nslots += c; }
    if (nslots == 0) return;
    const int slot = heads_before + __popcll(m_head & ((2ull << lane) - 1ull)) - 1;   // heads at or before me: my node's slot (for a valid lane)
    const int t = valid ? order[p] : 0;
    if (nslots <= kBlockNodes) {
        if (head) slot_bin[slot] = bin_index;
        for (int i = threadIdx.x; i < nslots * 126; i += blockDim.x) { const int w = i % 7; (&sbin[0][0])[i] = w < 3 ? kOrderedMax : (w < 6 ? kOrderedLowest : 0); }
        __syncthreads();
        if (valid) {
            const BuildNode& nd = nodes[k];
            const float* tb = tbox + 6 * (size_t)t;
            int* b = sbin[slot];
            for (int a = 0; a < 3; a++) {
                const float c = centroid[3 * (size_t)t + a];
                int s = 0;
                while (s < 5 && !(c <= plane_pos(nd.mn[a], nd.mx[a], s))) s++;
                int* w = b + (a * 6 + s) * 7;
                for (int q = 0; q < 3; q++) {
                    const float lo_v = tb[q] + 0.0f, hi_v = tb[3 + q] + 0.0f;       // NaN skipped, zeros canonicalised, as atomic_min_f / atomic_max_f
                    if (lo_v == lo_v) atomicMin(&w[q], ordered_int(lo_v));
                    if (hi_v == hi_v) atomicMax(&w[3 + q], ordered_int(hi_v));
                }
                atomicAdd(&w[6], 1);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nslots * 126; i += blockDim.x) {
            const int j = i / 126, r = i - 126 * j, a = r / 42, s = (r / 7) % 6, w = r % 7;
            if (sbin[j][(a * 6 + s) * 7 + 6] == 0) continue;    // empty bin: nothing to add
            Bins& b = bins[slot_bin[j]];
            const int v = sbin[j][r];
            if (w == 6) atomicAdd(&b.cnt[a][s], v);
            else if (w < 3) atomic_min_f(&b.mn[a][s][w], ordered_float(v));
            else atomic_max_f(&b.mx[a][s][w - 3], ordered_float(v));
        }
        return;
    }
    if (!valid) return;
    const float* tb = tbox + 6 * (size_t)t;
    const BuildNode& nd = nodes[k];
    Bins& b = bins[bin_index];
    for (int a = 0; a < 3; a++) {
        const float c = centroid[3 * (size_t)t + a];
        int s = 0;
        while (s < 5 && !(c <= plane_pos(nd.mn[a], nd.mx[a], s))) s++;
        for (int q = 0; q < 3; q++) { atomic_min_f(&b.mn[a][s][q], tb[q]); atomic_max_f(&b.mx[a][s][q], tb[3 + q]); }
        atomicAdd(&b.cnt[a][s], 1);
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

// The bins of one node, as decide reads them: in global memory as floats (the level kernels) or in LDS as the
// order-preserving integer images the LDS atomics work on (small_subtree_kernel).
struct GlobalBinsView {
    const Bins& b;
    __device__ __forceinline__ float mn(int a, int q, int c) const { return b.mn[a][q][c]; }
    __device__ __forceinline__ float mx(int a, int q, int c) const { return b.mx[a][q][c]; }
    __device__ __forceinline__ int cnt(int a, int q) const { return b.cnt[a][q]; }
};
struct LdsBinsView {
    const int* w;                                                // [3][6][7]: min xyz, max xyz, count
    __device__ __forceinline__ float mn(int a, int q, int c) const { return ordered_float(w[(a * 6 + q) * 7 + c]); }
    __device__ __forceinline__ float mx(int a, int q, int c) const { return ordered_float(w[(a * 6 + q) * 7 + 3 + c]); }
    __device__ __forceinline__ int cnt(int a, int q) const { return w[(a * 6 + q) * 7 + 6]; }
};

struct SplitDecision {
    bool split;
    int axis, nl;
    float split_pos;
    float cmn[2][3], cmx[2][3];                                  // the children's boxes
};

struct AxisEval { float cost, split; int nl, s; };             // s = -1: no plane of the axis was accepted

// evaluate_split (BVHTree.hpp:294-361) for one axis of a node whose bins are complete
template <class BinsView>
__device__ __forceinline__ AxisEval evaluate_axis(const BinsView& b, int a, float nmn_a, float nmx_a)
{
    AxisEval e;
    e.cost = FLT_MAX; e.split = 0.0f; e.nl = 0; e.s = -1;
    // right side of plane s = bins s+1..5, accumulated from the far end; the left side grows with s (each bin is read
    // twice instead of five times; unions of boxes are exact, so the order of the folds does not matter)
    float rmn[5][3], rmx[5][3];
    int rcount[5];
    {
        float amn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, amx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        int n = 0;
        for (int q = 5; q >= 1; q--) {
            for (int c = 0; c < 3; c++) { amn[c] = fminf(amn[c], b.mn(a, q, c)); amx[c] = fmaxf(amx[c], b.mx(a, q, c)); }
            n += b.cnt(a, q);
            for (int c = 0; c < 3; c++) { rmn[q - 1][c] = amn[c]; rmx[q - 1][c] = amx[c]; }
            rcount[q - 1] = n;
        }
    }
    float lmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int ln = 0;
    for (int s5 = 0; s5 < 5; s5++) {
        for (int c = 0; c < 3; c++) { lmn[c] = fminf(lmn[c], b.mn(a, s5, c)); lmx[c] = fmaxf(lmx[c], b.mx(a, s5, c)); }
        ln += b.cnt(a, s5);
        const float cost = box_cost(lmn, lmx, ln) + box_cost(rmn[s5], rmx[s5], rcount[s5]);         // BVHTree.hpp:351
        if (cost < e.cost) { e.cost = cost; e.split = plane_pos(nmn_a, nmx_a, s5); e.nl = ln; e.s = s5; }
    }
    return e;
}

// BVHTree.hpp:229-289 once the three axes are evaluated: the axis, "no split if not cheaper", the children's boxes and sizes.
// tri_at(i) = triangle at position i of the node's range (only the degenerate-input path below looks at triangles).
template <class BinsView, class TriAt>
__device__ __forceinline__ void choose_split(const BinsView& b, const AxisEval e0, const AxisEval e1, const AxisEval e2, const float* nmn, const float* nmx, int count,
                                             TriAt tri_at, const float* __restrict__ centroid, const float* __restrict__ tbox, SplitDecision& d)
{
    d.split = true; d.nl = 0;
    for (int side = 0; side < 2; side++)
        for (int c = 0; c < 3; c++) { d.cmn[side][c] = FLT_MAX; d.cmx[side][c] = -FLT_MAX; }
    int axis;
    if (e0.cost < e1.cost && e0.cost < e2.cost) axis = 0;       // BVHTree.hpp:229-243
    else if (e1.cost < e0.cost && e1.cost < e2.cost) axis = 1;
    else axis = 2;
    const AxisEval e = axis == 0 ? e0 : (axis == 1 ? e1 : e2);
    d.axis = axis;
    d.split_pos = e.split;
    if (e.cost >= box_cost(nmn, nmx, count)) { d.split = false; return; }       // BVHTree.hpp:246
    d.nl = e.nl;
    if (e.s >= 0) {
        for (int q = 0; q < 6; q++) {
            const int side = q <= e.s ? 0 : 1;
            for (int c = 0; c < 3; c++) { d.cmn[side][c] = fminf(d.cmn[side][c], b.mn(axis, q, c)); d.cmx[side][c] = fmaxf(d.cmx[side][c], b.mx(axis, q, c)); }
        }
    } else {
        // No plane was cheaper than FLT_MAX, and yet the test above let the node through: its own cost is infinite or
        // NaN (infinite or NaN coordinates).  The reference then partitions at the initial split position 0
        // (BVHTree.hpp:253); the bins say nothing about that plane, so this one thread walks the node's triangles.
        // Degenerate inputs only.
        d.nl = 0;
        for (int i = 0; i < count; i++) {
            const int t = tri_at(i);
            const int side = centroid[3 * (size_t)t + axis] <= d.split_pos ? 0 : 1;
            d.nl += side == 0 ? 1 : 0;
            for (int c = 0; c < 3; c++) { d.cmn[side][c] = fminf(d.cmn[side][c], tbox[6 * (size_t)t + c]); d.cmx[side][c] = fmaxf(d.cmx[side][c], tbox[6 * (size_t)t + 3 + c]); }
        }
        for (int side = 0; side < 2; side++)                     // (the atomics of the binned path canonicalise zeros to +0.0)
            for (int c = 0; c < 3; c++) { d.cmn[side][c] = d.cmn[side][c] + 0.0f; d.cmx[side][c] = d.cmx[side][c] + 0.0f; }
    }
    if (d.nl == 0 || count - d.nl == 0) d.split = false;         // BVHTree.hpp:279
}

// BVHTree.hpp:218-289 for one node whose bins are complete: the host builder's fp32 operations in the same order.
template <class BinsView, class TriAt>
__device__ __forceinline__ void decide_split(const BinsView& b, const float* nmn, const float* nmx, int count, TriAt tri_at,
                                             const float* __restrict__ centroid, const float* __restrict__ tbox, SplitDecision& d)
{
    const AxisEval e0 = evaluate_axis(b, 0, nmn[0], nmx[0]), e1 = evaluate_axis(b, 1, nmn[1], nmx[1]), e2 = evaluate_axis(b, 2, nmn[2], nmx[2]);
    choose_split(b, e0, e1, e2, nmn, nmx, count, tri_at, centroid, tbox, d);
}

// decide_split for every node of the level; children are appended to the node array with their boxes (the union
// of the bin boxes on their side of the plane = BVHTree.hpp:206-209 over their triangles) and, if they will evaluate a
// split themselves, a cleared Bins slot of the other parity -- or, when they hold at most small_limit triangles, a place in
// the list of subtrees small_subtree_kernel finishes after the level loop.
//
// Node pairs, bin slots and list places are handed out per WAVE (one atomic on the shared counter per wave, the lanes take
// consecutive pieces): with one atomic per node the deep levels -- tens of thousands of nodes -- queued on a single word.
__global__ void decide_kernel(BuildNode* nodes, Bins* bins, int bins_per_level, BuildState* st, int level, int max_depth, int cap,
                              const int32_t* __restrict__ order, const float* __restrict__ centroid, const float* __restrict__ tbox,
                              int32_t* __restrict__ small_roots, int small_limit)
{
    const int level_begin = st->begin[level], level_end = st->begin[level + 1];
    const int k = level_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    bool split = k < level_end && nodes[k < level_end ? k : level_begin].bin >= 0;      // depth or count limit: BVHTree.hpp:211-215
    BuildNode& nd = nodes[k < level_end ? k : level_begin];
    SplitDecision d;
    d.nl = 0; d.axis = 2; d.split_pos = 0.0f;
    if (split) {
        const int first = nd.first;
        decide_split(GlobalBinsView{bins[nd.bin]}, nd.mn, nd.mx, nd.count, [&](int i) { return order[first + i]; }, centroid, tbox, d);
        split = d.split;
    }
    const int nl = split ? d.nl : 0;
    // ---- one allocation per wave: 2 nodes per splitting lane, one bin slot per child that will evaluate a split itself
    const int nr = nd.count - nl;
    const bool go_a = split && !(nd.depth + 1 >= max_depth || nl <= 1), go_b = split && !(nd.depth + 1 >= max_depth || nr <= 1);   // BVHTree.hpp:211-215 for the children
    const bool small_a = go_a && nl <= small_limit, small_b = go_b && nr <= small_limit;
    const bool bin_a = go_a && !small_a, bin_b = go_b && !small_b;
    const unsigned long long m_split = __ballot(split), m_a = __ballot(bin_a), m_b = __ballot(bin_b), m_sa = __ballot(small_a), m_sb = __ballot(small_b);
    if (m_split == 0ull) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int leader = __ffsll((long long)m_split) - 1;
    int node_base = 0, bin_base = 0, small_base = 0;
    if (lane == leader) {
        node_base = atomicAdd(&st->total, 2 * __popcll(m_split));
        if ((m_a | m_b) != 0ull) bin_base = atomicAdd(&st->bins_used[(level + 1) & 1], __popcll(m_a) + __popcll(m_b));
        if ((m_sa | m_sb) != 0ull) small_base = atomicAdd(&st->num_small, __popcll(m_sa) + __popcll(m_sb));
    }
    node_base = __shfl(node_base, leader);
    bin_base = __shfl(bin_base, leader);
    small_base = __shfl(small_base, leader);
    if (!split) return;
    const int a = node_base + 2 * __popcll(m_split & below);
    if (a + 2 > cap || level + 2 > kMaxLevels) { st->overflow = 1; return; }
    nd.axis = d.axis; nd.split_pos = d.split_pos; nd.nl = nl;
    nd.child_a = a; nd.child_b = a + 1;
    const int other = ((level + 1) & 1) * bins_per_level;        // children use the other half of the bin array
    const int slot_a = bin_base + __popcll(m_a & below) + __popcll(m_b & below);       // (<= n / 2 such nodes per level)
    const int place_a = small_base + __popcll(m_sa & below) + __popcll(m_sb & below);
    for (int side = 0; side < 2; side++) {
        BuildNode& ch = nodes[a + side];
        for (int c = 0; c < 3; c++) { ch.mn[c] = d.cmn[side][c]; ch.mx[c] = d.cmx[side][c]; }
        ch.first = side == 0 ? nd.first : nd.first + nl;
        ch.count = side == 0 ? nl : nr;
        ch.depth = nd.depth + 1;
        ch.child_a = ch.child_b = -1; ch.axis = 0; ch.split_pos = 0.0f; ch.nl = 0;
        ch.bin = -1;                                             // a leaf for the level kernels (also when a wave will finish it later)
        if (side == 0 ? bin_a : bin_b) {
            ch.bin = other + slot_a + (side == 1 && bin_a ? 1 : 0);
            clear_bins(bins[ch.bin]);
        }
        if (side == 0 ? small_a : small_b) small_roots[place_a + (side == 1 && small_a ? 1 : 0)] = a + side;
    }
}

// A subtree of at most 64 triangles is finished by ONE wave, lane = triangle, instead of travelling through fifteen more
// levels of the loop above (where such nodes are the bulk of the work: tens of thousands of nodes of a few triangles each,
// 21 global atomics per triangle and level).  Everything stays in the wave: the triangles' centroids and boxes in
// registers, the bins of all nodes of the subtree's current level in LDS (LDS atomics on the ordered-integer images), one
// lane per node runs decide_split -- the same function the level kernels use -- and the stable partition is a permutation
// of lanes (ranks from ballots, data through LDS).  The subtree's nodes collect in LDS and are appended to the node array
// in one piece at the end; the numbering pass does not care in which order nodes were created.
struct SmallLds {
    int bins[32][126];                                           // a node that evaluates a split holds >= 2 triangles: <= 32 per level
    float nbox[64][6];                                           // node records, keyed by the lane where the node's range starts
    int ncount[64], ndepth[64], nid[64];
    int slot_head[32];
    int r_split[32], r_axis[32], r_nl[32], r_go_a[32], r_go_b[32];
    float r_pos[32];
    float e_cost[96], e_split[96];                               // per (node, axis) evaluation, item 3 j + a
    int e_nl[96], e_s[96];
    // the subtree's nodes stay here until its last level is done (children as indices into this array) and then take one
    // contiguous piece of the node array: one atomic on the shared node counter per subtree instead of one per level
    float l_mn[126][3], l_mx[126][3];
    int l_first[126], l_count[126], l_depth[126], l_child[126], l_axis[126], l_nl[126];
    float l_pos[126];
    int root_child, root_axis, root_nl;
    float root_pos;
    int tri[64];                                                 // triangle at every position of the subtree's range
    int p_t[64], p_seg[64], p_cnt[64];                           // staging of the lane permutation
    float p_c[64][3], p_b[64][6];
};

__global__ __launch_bounds__(64) void small_subtree_kernel(BuildNode* nodes, BuildState* st, const int32_t* __restrict__ small_roots,
                                                           int32_t* order, const float* __restrict__ centroid, const float* __restrict__ tbox,
                                                           int max_depth, int cap)
{
    __shared__ SmallLds L;
    if ((int)blockIdx.x >= st->num_small) return;
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int root = small_roots[blockIdx.x];
    const int first = nodes[root].first, total = nodes[root].count;
    if (lane == 0) {
        for (int c = 0; c < 3; c++) { L.nbox[0][c] = nodes[root].mn[c]; L.nbox[0][3 + c] = nodes[root].mx[c]; }
        L.ncount[0] = total; L.ndepth[0] = nodes[root].depth; L.nid[0] = -1;      // (-1: the subtree's root, already in the node array)
        L.root_child = -1;
    }
    int created = 0;                                             // nodes of the subtree so far (wave-uniform)
    int seg = lane < total ? 0 : -1, cnt = total;                // my node: the lane its range starts at (-1: finished), its size
    int deepest = 0;                                             // depth of the deepest node this lane created
    int t = 0;
    float cen[3] = {0.0f, 0.0f, 0.0f}, box[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (lane < total) {
        t = order[first + lane];
        for (int c = 0; c < 3; c++) cen[c] = centroid[3 * (size_t)t + c];
        for (int c = 0; c < 6; c++) box[c] = tbox[6 * (size_t)t + c];
    }
    L.tri[lane] = t;
    __syncthreads();
    for (int iteration = 0; iteration < kMaxLevels; iteration++) {
        const bool active = seg >= 0;
        const unsigned long long m_head = __ballot(active && lane == seg);
        if (m_head == 0ull) break;
        const int nslots = __popcll(m_head);
        const int slot = __popcll(m_head & (below | (1ull << lane))) - 1;     // the nearest head at or below an active lane is its own
        if (active && lane == seg) L.slot_head[slot] = lane;
        for (int i = lane; i < nslots * 126; i += 64) { const int w = i % 7; (&L.bins[0][0])[i] = w < 3 ? kOrderedMax : (w < 6 ? kOrderedLowest : 0); }
        __syncthreads();
        // ---- bins (bins_kernel for one wave)
        if (active) {
            int* b = L.bins[slot];
            for (int a = 0; a < 3; a++) {
                const float nmn = L.nbox[seg][a], nmx = L.nbox[seg][3 + a];
                int s = 0;
                while (s < 5 && !(cen[a] <= plane_pos(nmn, nmx, s))) s++;
                int* w = b + (a * 6 + s) * 7;
                for (int q = 0; q < 3; q++) {
                    const float lo_v = box[q] + 0.0f, hi_v = box[3 + q] + 0.0f;          // NaN skipped, zeros canonicalised (atomic_min_f / atomic_max_f)
                    if (lo_v == lo_v) atomicMin(&w[q], ordered_int(lo_v));
                    if (hi_v == hi_v) atomicMax(&w[3 + q], ordered_int(hi_v));
                }
                atomicAdd(&w[6], 1);
            }
        }
        __syncthreads();
        // ---- decide (decide_kernel for the <= 32 nodes of this level of the subtree): lane j owns slot j
        {
            const bool owner = lane < nslots;
            const int h = owner ? L.slot_head[lane] : 0;
            SplitDecision d;
            d.split = false; d.nl = 0; d.axis = 2; d.split_pos = 0.0f;
            float nmn[3], nmx[3];
            for (int c = 0; c < 3; c++) { nmn[c] = L.nbox[h][c]; nmx[c] = L.nbox[h][3 + c]; }
            const int count = L.ncount[h], depth = L.ndepth[h], id = L.nid[h];
            // the three axes of a node on three lanes (item 3 j + a), then the node's lane j collects them: a third of the
            // longest dependent chain while few nodes are alive, which is most of a subtree's life
            for (int item = lane; item < nslots * 3; item += 64) {
                const int j = item / 3, a = item - 3 * j, hj = L.slot_head[j];
                const AxisEval e = evaluate_axis(LdsBinsView{L.bins[j]}, a, L.nbox[hj][a], L.nbox[hj][3 + a]);
                L.e_cost[item] = e.cost; L.e_split[item] = e.split; L.e_nl[item] = e.nl; L.e_s[item] = e.s;
            }
            __syncthreads();
            if (owner) {
                auto stored = [&](int a) { AxisEval e; e.cost = L.e_cost[3 * lane + a]; e.split = L.e_split[3 * lane + a]; e.nl = L.e_nl[3 * lane + a]; e.s = L.e_s[3 * lane + a]; return e; };
                choose_split(LdsBinsView{L.bins[lane]}, stored(0), stored(1), stored(2), nmn, nmx, count, [&](int i) { return L.tri[h + i]; }, centroid, tbox, d);
            }
            const bool split = owner && d.split;
            const unsigned long long m_split = __ballot(split);
            const int a = created + 2 * __popcll(m_split & below);
            created += 2 * __popcll(m_split);
            if (owner) {
                const int nl = d.nl, nr = count - d.nl;
                const bool go_a = split && !(depth + 1 >= max_depth || nl <= 1), go_b = split && !(depth + 1 >= max_depth || nr <= 1);
                L.r_split[lane] = split ? 1 : 0; L.r_axis[lane] = d.axis; L.r_pos[lane] = d.split_pos; L.r_nl[lane] = nl;
                L.r_go_a[lane] = go_a ? 1 : 0; L.r_go_b[lane] = go_b ? 1 : 0;
                if (split) {
                    if (id < 0) { L.root_child = a; L.root_axis = d.axis; L.root_pos = d.split_pos; L.root_nl = nl; }
                    else { L.l_child[id] = a; L.l_axis[id] = d.axis; L.l_pos[id] = d.split_pos; L.l_nl[id] = nl; }
                    for (int side = 0; side < 2; side++) {
                        const int ch = a + side;
                        for (int c = 0; c < 3; c++) { L.l_mn[ch][c] = d.cmn[side][c]; L.l_mx[ch][c] = d.cmx[side][c]; }
                        L.l_first[ch] = side == 0 ? first + h : first + h + nl;
                        L.l_count[ch] = side == 0 ? nl : nr;
                        L.l_depth[ch] = depth + 1;
                        L.l_child[ch] = -1; L.l_axis[ch] = 0; L.l_pos[ch] = 0.0f; L.l_nl[ch] = 0;
                        const int key = side == 0 ? h : h + nl;          // (side 0 reuses the parent's entry: this lane is its only reader)
                        for (int c = 0; c < 3; c++) { L.nbox[key][c] = d.cmn[side][c]; L.nbox[key][3 + c] = d.cmx[side][c]; }
                        L.ncount[key] = side == 0 ? nl : nr; L.ndepth[key] = depth + 1; L.nid[key] = ch;
                    }
                    deepest = depth + 1;
                }
            }
        }
        __syncthreads();
        // ---- stable partition (flags_scan_kernel + scatter_kernel for one wave): a permutation of lanes
        int q = lane, new_seg = -1, new_cnt = 0;
        const bool splits = active && L.r_split[slot] != 0;
        const int split_axis = splits ? L.r_axis[slot] : 0;
        const bool left = splits && (split_axis == 0 ? cen[0] : (split_axis == 1 ? cen[1] : cen[2])) <= L.r_pos[slot];
        const unsigned long long m_left = __ballot(left);
        if (splits) {
            const unsigned long long range = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << seg;
            const int nl = L.r_nl[slot];
            const int lrank = __popcll(m_left & range & below);       // left-going triangles of my node before me
            if (left) { q = seg + lrank; new_seg = L.r_go_a[slot] ? seg : -1; new_cnt = nl; }
            else { q = seg + nl + (lane - seg - lrank); new_seg = L.r_go_b[slot] ? seg + nl : -1; new_cnt = cnt - nl; }
        }
        L.p_t[q] = t; L.p_seg[q] = new_seg; L.p_cnt[q] = new_cnt;
        for (int c = 0; c < 3; c++) L.p_c[q][c] = cen[c];
        for (int c = 0; c < 6; c++) L.p_b[q][c] = box[c];
        __syncthreads();
        t = L.p_t[lane]; seg = L.p_seg[lane]; cnt = L.p_cnt[lane];
        for (int c = 0; c < 3; c++) cen[c] = L.p_c[lane][c];
        for (int c = 0; c < 6; c++) box[c] = L.p_b[lane][c];
        L.tri[lane] = t;
        __syncthreads();
    }
    if (lane < total) order[first + lane] = t;
    // ---- the subtree's nodes into the node array: one piece, one atomic (and one for the depth: every lane of every level
    //      adding its own put a million atomics in line for that one word)
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(deepest, o); deepest = v > deepest ? v : deepest; }
    if (created == 0) return;
    int base = 0;
    if (lane == 0) { base = atomicAdd(&st->total, created); atomicMax(&st->levels, deepest); }
    base = __shfl(base, 0);
    if (base + created > cap) { if (lane == 0) st->overflow = 1; return; }
    for (int i = lane; i < created; i += 64) {
        BuildNode& g = nodes[base + i];
        for (int c = 0; c < 3; c++) { g.mn[c] = L.l_mn[i][c]; g.mx[c] = L.l_mx[i][c]; }
        g.first = L.l_first[i]; g.count = L.l_count[i]; g.depth = L.l_depth[i];
        const int child = L.l_child[i];
        g.child_a = child < 0 ? -1 : base + child; g.child_b = child < 0 ? -1 : base + child + 1;
        g.axis = L.l_axis[i]; g.split_pos = L.l_pos[i]; g.nl = L.l_nl[i]; g.bin = -1;
    }
    if (lane == 0) {
        BuildNode& g = nodes[root];
        g.child_a = base + L.root_child; g.child_b = base + L.root_child + 1;
        g.axis = L.root_axis; g.split_pos = L.root_pos; g.nl = L.root_nl;
    }
}

// "goes left" flag of every position whose node splits (BVHTree.hpp:253-277)
__global__ void flags_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                             const float* __restrict__ centroid, const BuildNode* __restrict__ nodes, const BuildState* __restrict__ st,
                             int level, int32_t* __restrict__ flags)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int f = 0;
    const int k = node_of[p];
    if (k >= st->begin[level]) {
        const BuildNode& nd = nodes[k];
        if (nd.child_a >= 0) f = centroid[3 * (size_t)order[p] + nd.axis] <= nd.split_pos ? 1 : 0;
    }
    flags[p] = f;
}

// The usual case (up to kScanBlocks x 256 = 1 M triangles): flags, their exclusive scan inside each 256-position block and
// the block totals in ONE launch; the scatter kernel turns the block totals into block offsets itself (every workgroup
// scans the short totals array into LDS).  Two launches per level instead of flags + a library scan (two more) + scatter.
constexpr int kScanBlocks = 4096;
__global__ __launch_bounds__(256) void flags_scan_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                                                         const float* __restrict__ centroid, const BuildNode* __restrict__ nodes,
                                                         const BuildState* __restrict__ st, int level, int32_t* __restrict__ flags,
                                                         int32_t* __restrict__ local, int32_t* __restrict__ block_total)
{
    typedef hipcub::BlockScan<int, 256> BlockScan;
    __shared__ typename BlockScan::TempStorage tmp;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int f = 0;
    if (p < n) {
        const int k = node_of[p];
        if (k >= st->begin[level]) {
            const BuildNode& nd = nodes[k];
            if (nd.child_a >= 0) f = centroid[3 * (size_t)order[p] + nd.axis] <= nd.split_pos ? 1 : 0;
        }
    }
    int before = 0, total = 0;
    BlockScan(tmp).ExclusiveSum(f, before, total);
    if (p < n) { flags[p] = f; local[p] = before; }
    if (threadIdx.x == 0) block_total[blockIdx.x] = total;
}

// stable partition: left triangles keep their order at the front of the node's range, right ones behind them.
// Thread 0 also closes the level: the next level ends where the node array now ends.
__global__ void scatter_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ node_of, int n,
                               const BuildNode* __restrict__ nodes, BuildState* st, int level, const int32_t* __restrict__ flags,
                               const int32_t* __restrict__ scan, const int32_t* __restrict__ block_total, int num_blocks,
                               int32_t* __restrict__ order_out, int32_t* __restrict__ node_of_out)
{
    // scan[] holds device-wide exclusive sums (block_total == nullptr: the hipcub path) or sums inside each 256-position
    // block; in the second case the offset of block j is the sum of the totals of the blocks before it
    __shared__ int block_offset[kScanBlocks];
    if (block_total) {
        typedef hipcub::BlockScan<int, 256> BlockScan;
        __shared__ typename BlockScan::TempStorage tmp;
        const int per_thread = (num_blocks + 255) / 256, j0 = threadIdx.x * per_thread;
        int sum = 0;
        for (int j = j0; j < j0 + per_thread && j < num_blocks; j++) sum += block_total[j];
        int before = 0;
        BlockScan(tmp).ExclusiveSum(sum, before);
        for (int j = j0; j < j0 + per_thread && j < num_blocks; j++) { block_offset[j] = before; before += block_total[j]; }
        __syncthreads();
    }
    auto scan_at = [&](int q) { return block_total ? block_offset[q >> 8] + scan[q] : scan[q]; };
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int level_begin = st->begin[level];
    if (p == 0) {
        const int total = st->total < 0x7fffffff ? st->total : 0x7fffffff;
        if (level + 2 <= kMaxLevels + 1) st->begin[level + 2] = total;
        if (total > st->begin[level + 1]) st->levels = level + 2;
        st->bins_used[level & 1] = 0;                            // free for the level after next
    }
    if (p >= n) return;
    const int k = node_of[p];
    int q = p, child = k;
    if (k >= level_begin) {
        const BuildNode& nd = nodes[k];
        if (nd.child_a >= 0) {
            const int lrank = scan_at(p) - scan_at(nd.first);    // left-going triangles before p inside the node
            if (flags[p]) { q = nd.first + lrank; child = nd.child_a; }
            else { q = nd.first + nd.nl + (p - nd.first - lrank); child = nd.child_b; }
        } else {
            child = -1 - k;                                      // finished leaf: never looked at again (negative < level_begin)
        }
    }
    order_out[q] = order[p];
    node_of_out[q] = child;
}

// histogram of range ends (for the pre-order numbering)
__global__ void ends_kernel(const BuildNode* __restrict__ nodes, const BuildState* __restrict__ st, int32_t* __restrict__ end_hist)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= st->total) return;
    atomicAdd(&end_hist[nodes[k].first + nodes[k].count], 1);
}

// BVHTree.hpp:283-289 (child a right after its parent, child b after a's whole subtree) in closed form, and the output arrays
__device__ __forceinline__ int preorder_of(const BuildNode& nd, const int32_t* ends_before)
{
    // ends_before[p] = nodes whose range ends at or before p = exclusive scan of the histogram, taken at p + 1
    return nd.depth - 1 + ends_before[nd.first + 1];
}
__global__ void emit_kernel(const BuildNode* __restrict__ nodes, const BuildState* __restrict__ st, const int32_t* __restrict__ ends_before,
                            float* __restrict__ bounds, int32_t* __restrict__ children, int32_t* __restrict__ leaf_first, int32_t* __restrict__ leaf_count)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= st->total) return;
    const BuildNode& nd = nodes[k];
    // (a node with an empty range -- only the root of an empty mesh -- has nothing to its left)
    const size_t o = nd.count > 0 ? (size_t)preorder_of(nd, ends_before) : 0;
    for (int c = 0; c < 3; c++) { bounds[6 * o + c] = nd.mn[c]; bounds[6 * o + 3 + c] = nd.mx[c]; }
    const bool leaf = nd.child_a < 0;
    children[2 * o] = leaf ? -1 : preorder_of(nodes[nd.child_a], ends_before);
    children[2 * o + 1] = leaf ? -1 : preorder_of(nodes[nd.child_b], ends_before);
    leaf_first[o] = nd.first;
    leaf_count[o] = leaf ? nd.count : 0;
}

// one cached device arena per process (grow-only): a build of a 70 k-triangle mesh is shorter than a hipMalloc / hipFree pair
std::mutex g_arena_mutex;
char* g_arena = nullptr;
size_t g_arena_bytes = 0;
int g_arena_device = -1;

constexpr size_t kExtraBytes = 2048;

// What a finished build leaves on the device (in the cached arena: valid until the next build of the process; the arena
// mutex is held by whoever uses it).
struct DeviceBuild {
    BuildNode* nodes = nullptr;         // breadth-first
    BuildState* state = nullptr;
    int32_t* ends_before = nullptr;     // exclusive scan of the range-end histogram (pre-order numbering, see preorder_of)
    int32_t* order = nullptr;           // triangle indices in leaf order
    int32_t *hist = nullptr, *hist_scan = nullptr;      // n + 2 ints each: free for the caller after the build
    char* extra = nullptr;                              // kExtraBytes of scratch for the caller
    void* tmp = nullptr; size_t tmp_bytes = 0;          // scan scratch (n + 2 items)
    float *bounds = nullptr; int32_t *children = nullptr, *leaf_first = nullptr, *leaf_count = nullptr;   // emit_kernel's outputs
    BuildState st;                      // host copy after the last kernel
    int cap = 0;
};

// The build itself: `vertices` is a host array (copied into the arena) or, with on_device, a device array used where it lies.
// Kernels run on `stream`; the few state read-backs wait for it.  Caller holds g_arena_mutex.
int build_core(const float* vertices, bool on_device, int32_t n, int32_t max_depth, hipStream_t stream, bool emit_host_layout, DeviceBuild& out)
{
    if (max_depth > kMaxLevels) max_depth = kMaxLevels;          // (levels beyond the table cannot be represented; the reference uses 32)
    int rc = RT_OK;
    const int cap = n > 0 ? 2 * n : 1;                           // <= 2n - 1 nodes
    const int T = 256;
    const int gridN = (n + T - 1) / T;
    const int bins_per_level = n / 2 + 1;                        // nodes that evaluate a split hold >= 2 triangles each
    float *d_v, *d_centroid, *d_tbox, *d_bounds;
    int32_t *d_order[2], *d_nodeof[2], *d_flags, *d_scan, *d_btot, *d_hist, *d_hscan, *d_children, *d_lfirst, *d_lcount, *d_small;
    BuildNode* d_nodes;
    Bins* d_bins;
    BuildState* d_state;
    void* d_tmp;
    size_t tmp_bytes = 0, tmp2 = 0;
    int cur = 0;
    BuildState& st = out.st;

    {
        hipError_t e1 = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (int32_t*)nullptr, (int32_t*)nullptr, n > 0 ? n : 1);
        hipError_t e2 = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, (int32_t*)nullptr, (int32_t*)nullptr, n + 2);
        if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
        tmp_bytes = tmp_bytes > tmp2 ? tmp_bytes : tmp2;
        const size_t n1 = n > 0 ? (size_t)n : 1;
        size_t off = 0;
        auto take = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_v = take(on_device ? 4 : n1 * 9 * 4), o_cen = take(n1 * 3 * 4), o_tbox = take(n1 * 6 * 4), o_ord0 = take(n1 * 4), o_ord1 = take(n1 * 4),
                     o_nof0 = take(n1 * 4), o_nof1 = take(n1 * 4), o_flags = take(n1 * 4), o_scan = take(n1 * 4), o_btot = take(((n1 + 255) / 256 + 1) * 4), o_hist = take((n1 + 2) * 4), o_hscan = take((n1 + 2) * 4),
                     o_state = take(sizeof(BuildState)), o_nodes = take(((size_t)cap + 2) * sizeof(BuildNode)),
                     o_bins = take(2 * (size_t)bins_per_level * sizeof(Bins)), o_tmp = take(tmp_bytes ? tmp_bytes : 1),
                     o_bounds = take((size_t)cap * 6 * 4), o_children = take((size_t)cap * 2 * 4), o_lfirst = take((size_t)cap * 4),
                     o_lcount = take((size_t)cap * 4), o_small = take((n1 / 2 + 2) * 4),
                     o_hscan2 = take((n1 + 2) * 4), o_extra = take(kExtraBytes);      // (used by the device-resident rebuild)
        int device = 0;
        hipError_t he = hipGetDevice(&device);
        if (he != hipSuccess) return he == hipErrorNoDevice ? RT_E_NODEVICE : (int)he;
        if (g_arena_bytes < off || g_arena_device != device) {
            (void)hipFree(g_arena);
            g_arena = nullptr; g_arena_bytes = 0;
            he = hipMalloc((void**)&g_arena, off + off / 4);     // some slack: meshes of similar size reuse it
            if (he != hipSuccess) return (int)he;
            g_arena_bytes = off + off / 4; g_arena_device = device;
        }
        char* arena = g_arena;
        d_v = on_device ? const_cast<float*>(vertices) : (float*)(arena + o_v);
        d_centroid = (float*)(arena + o_cen); d_tbox = (float*)(arena + o_tbox);
        d_order[0] = (int32_t*)(arena + o_ord0); d_order[1] = (int32_t*)(arena + o_ord1);
        d_nodeof[0] = (int32_t*)(arena + o_nof0); d_nodeof[1] = (int32_t*)(arena + o_nof1);
        d_flags = (int32_t*)(arena + o_flags); d_scan = (int32_t*)(arena + o_scan); d_btot = (int32_t*)(arena + o_btot); d_hist = (int32_t*)(arena + o_hist); d_hscan = (int32_t*)(arena + o_hscan);
        d_state = (BuildState*)(arena + o_state);
        d_nodes = (BuildNode*)(arena + o_nodes); d_bins = (Bins*)(arena + o_bins); d_tmp = arena + o_tmp;
        d_bounds = (float*)(arena + o_bounds); d_children = (int32_t*)(arena + o_children);
        d_lfirst = (int32_t*)(arena + o_lfirst); d_lcount = (int32_t*)(arena + o_lcount); d_small = (int32_t*)(arena + o_small);
        out.hist_scan = (int32_t*)(arena + o_hscan2); out.extra = arena + o_extra;
    }
    // subtrees of at most this many triangles leave the level loop and are finished by one wave each (RT_BVH_SMALL=0: everything
    // goes through the level loop, the tests' way of keeping that path covered)
    int small_limit = kSmall;
    if (const char* e = getenv("RT_BVH_SMALL")) { const int v = atoi(e); small_limit = v < 0 ? 0 : (v > kSmall ? kSmall : v); }

    const bool debug = getenv("RT_BVH_DEBUG") != nullptr;           // diagnostics: phase timings (with extra synchronisation) and a state dump
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto read_state = [&]() -> hipError_t {                         // the level loop's question to the device: one small read-back
        hipError_t e = hipMemcpyAsync(&st, d_state, sizeof st, hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipStreamSynchronize(stream);
    };
    double t_start = now(), t_in = 0, t_kernels = 0;
    if (n > 0 && !on_device) RT_HIP(hipMemcpyAsync(d_v, vertices, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice, stream));
    if (debug) { (void)hipDeviceSynchronize(); t_in = now(); }
    hipLaunchKernelGGL(prep_kernel, dim3((n + 1 + T - 1) / T), dim3(T), 0, stream, d_v, n, max_depth, d_centroid, d_tbox, d_order[0], d_nodeof[0],
                       d_nodes, d_bins, d_state, d_hist, d_small, small_limit);
    if (n > 0) hipLaunchKernelGGL(root_bounds_kernel, dim3(std::min(64, (n + 1023) / 1024)), dim3(1024), 0, stream, n, d_tbox, d_nodes);

    // ---- the level loop: four launches per level.  Where a level starts and ends is device state, so the host enqueues
    //      levels without waiting; it only has to learn when to stop.  Nodes of more than small_limit triangles are gone
    //      after about log2(n / small_limit) levels, so that is where the host first asks (one 4-byte-class read-back; a
    //      level that is not needed costs four empty launches, about as much), then every other level.  A tree over n
    //      triangles has at most n levels, the depth limit caps it at max_depth. ----
    st.num_small = 0;
    {
        bool have_state = false;
        if (n > 1 && n > small_limit) {
            const bool library_scan = getenv("RT_BVH_LIBRARY_SCAN") != nullptr;     // tests: force the path meshes above 1 M triangles take
            const int levels_to_run = max_depth < n ? max_depth : n;
            int check_at = 2;
            for (long long m = small_limit > 0 ? small_limit : 1; m < n; m *= 2) check_at++;
            for (int l = 0; l < levels_to_run; l++) {
                const long long width = l < 30 ? (1ll << l) : (1ll << 30);
                const int gridL = (int)(((width < n ? width : (long long)n) + T - 1) / T);
                hipLaunchKernelGGL(bins_kernel, dim3(gridN), dim3(T), 0, stream, d_order[cur], d_nodeof[cur], n, d_centroid, d_tbox, d_nodes, d_bins, d_state, l);
                hipLaunchKernelGGL(decide_kernel, dim3(gridL), dim3(T), 0, stream, d_nodes, d_bins, bins_per_level, d_state, l, max_depth, cap,
                                   d_order[cur], d_centroid, d_tbox, d_small, small_limit);
                if (gridN <= kScanBlocks && !library_scan) {
                    hipLaunchKernelGGL(flags_scan_kernel, dim3(gridN), dim3(T), 0, stream, d_order[cur], d_nodeof[cur], n, d_centroid, d_nodes, d_state, l,
                                       d_flags, d_scan, d_btot);
                    hipLaunchKernelGGL(scatter_kernel, dim3(gridN), dim3(T), 0, stream, d_order[cur], d_nodeof[cur], n, d_nodes, d_state, l, d_flags, d_scan,
                                       d_btot, gridN, d_order[cur ^ 1], d_nodeof[cur ^ 1]);
                } else {                                         // very large meshes: library scan over the whole array
                    hipLaunchKernelGGL(flags_kernel, dim3(gridN), dim3(T), 0, stream, d_order[cur], d_nodeof[cur], n, d_centroid, d_nodes, d_state, l, d_flags);
                    RT_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flags, d_scan, n, stream));
                    hipLaunchKernelGGL(scatter_kernel, dim3(gridN), dim3(T), 0, stream, d_order[cur], d_nodeof[cur], n, d_nodes, d_state, l, d_flags, d_scan,
                                       (const int32_t*)nullptr, 0, d_order[cur ^ 1], d_nodeof[cur ^ 1]);
                }
                cur ^= 1;
                have_state = false;
                if (l + 1 >= check_at && l + 1 < levels_to_run) {
                    RT_HIP(read_state());
                    have_state = true;
                    if (st.bins_used[(l + 1) & 1] == 0) break;   // no node of the next level evaluates a split
                    check_at = l + 3;
                }
            }
        }
        if (small_limit > 0 && n > 1) {
            if (!have_state) RT_HIP(read_state());
            if (st.num_small > 0)
                hipLaunchKernelGGL(small_subtree_kernel, dim3(st.num_small), dim3(64), 0, stream, d_nodes, d_state, d_small, d_order[cur], d_centroid, d_tbox,
                                   max_depth, cap);
        }
    }
    // ---- breadth-first -> the reference's depth-first numbering ----
    hipLaunchKernelGGL(ends_kernel, dim3((cap + T - 1) / T), dim3(T), 0, stream, d_nodes, d_state, d_hist);
    RT_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_hist, d_hscan, n + 2, stream));
    if (emit_host_layout)
        hipLaunchKernelGGL(emit_kernel, dim3((cap + T - 1) / T), dim3(T), 0, stream, d_nodes, d_state, d_hscan, d_bounds, d_children, d_lfirst, d_lcount);
    RT_HIP(hipGetLastError());
    if (debug) { (void)hipDeviceSynchronize(); t_kernels = now(); }
    RT_HIP(read_state());
    if (debug) {
        const int total = st.total;
        fprintf(stderr, "bvh: n %d total %d levels %d overflow %d bins_used %d %d small subtrees %d (limit %d) begin", n, st.total, st.levels, st.overflow,
                st.bins_used[0], st.bins_used[1], st.num_small, small_limit);
        for (int l = 0; l < 8; l++) fprintf(stderr, " %d", st.begin[l]);
        fprintf(stderr, "\n");
        std::vector<BuildNode> hn((size_t)(total < 7 ? total : 7));
        (void)hipMemcpy(hn.data(), d_nodes, hn.size() * sizeof(BuildNode), hipMemcpyDeviceToHost);
        for (size_t k = 0; k < hn.size(); k++)
            fprintf(stderr, "  node %zu: [%g %g %g]-[%g %g %g] first %d count %d depth %d children %d %d axis %d split %g nl %d bin %d\n", k, hn[k].mn[0], hn[k].mn[1], hn[k].mn[2],
                    hn[k].mx[0], hn[k].mx[1], hn[k].mx[2], hn[k].first, hn[k].count, hn[k].depth, hn[k].child_a, hn[k].child_b, hn[k].axis, hn[k].split_pos, hn[k].nl, hn[k].bin);
        for (int k = 0; k < 3 && k < total; k++) {
            if (hn[k].bin < 0) continue;
            Bins hb;
            (void)hipMemcpy(&hb, d_bins + hn[k].bin, sizeof hb, hipMemcpyDeviceToHost);
            fprintf(stderr, "  bins of node %d:", k);
            for (int a = 0; a < 3; a++) { fprintf(stderr, " |"); for (int q = 0; q < 6; q++) fprintf(stderr, " %d", hb.cnt[a][q]); }
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "bvh timing: copy in %.3f ms, kernels %.3f ms\n", t_in - t_start, t_kernels - t_in);
    }
    if (st.overflow || st.total < 1 || st.total > cap) { rc = RT_E_INVALID; goto done; }
    out.nodes = d_nodes; out.state = d_state; out.ends_before = d_hscan; out.order = d_order[cur]; out.hist = d_hist;
    out.tmp = d_tmp; out.tmp_bytes = tmp_bytes;
    out.bounds = d_bounds; out.children = d_children; out.leaf_first = d_lfirst; out.leaf_count = d_lcount; out.cap = cap;
done:
    return rc;
}

}  // namespace

extern "C" int rt_bvh_build(const float* vertices, int32_t n, int32_t max_depth, float* node_bounds, int32_t* node_children,
                            int32_t* node_leaf_first, int32_t* node_leaf_count, int32_t* leaf_indices, int32_t* num_nodes,
                            int32_t* num_levels)
{
    if (n < 0 || max_depth < 1 || (n > 0 && !vertices) || !node_bounds || !node_children || !node_leaf_first ||
        !node_leaf_count || !num_nodes || (n > 0 && !leaf_indices)) return RT_E_INVALID;
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    DeviceBuild b;
    int rc = build_core(vertices, false, n, max_depth, nullptr, true, b);
    if (rc) return rc;
    const int total = b.st.total;
    RT_HIP(hipMemcpy(node_bounds, b.bounds, (size_t)total * 6 * sizeof(float), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(node_children, b.children, (size_t)total * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(node_leaf_first, b.leaf_first, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(node_leaf_count, b.leaf_count, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n > 0) RT_HIP(hipMemcpy(leaf_indices, b.order, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    *num_nodes = total;
    if (num_levels) *num_levels = b.st.levels;
done:
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Device-resident rebuild: a new tree over a mesh whose triangles have moved (or been replaced by at most as many), written
// straight into the scene's record arrays -- no host copy of the vertices, of the tree, or of the records.  The host path for
// the same job is rt_bvh_build (9 MB in, 12 MB out for 70 k triangles), MeshPrimitive / Scene::upload_to_device (flatten
// again) and rt_scene_upload (re-lay out on the host, copy 13 MB in): 7 ms around 0.6 ms of kernels.  Reference hooks:
// MeshPrimitive::build_bvh (MeshPrimitive.cpp:38-56) and Scene::upload_to_device (Scene.cpp:25-65).
//
// The records a mesh owns are laid out by rt_scene_upload as [int_cap interior records][slot_cap triangle records], the
// interior nodes in pre-order; an interior node's record index is node_base + (its pre-order number - the leaves before it),
// and both counts come from range-end histograms (a node is preceded, in pre-order, by its ancestors and by every node whose
// triangle range ends at or before its own first triangle -- see preorder_of).  Triangle slot = slot_base + position in leaf
// order.  The kernels below write exactly what rt_scene_upload writes for the same tree.
namespace {

// hist[e] += 1 for every LEAF whose range ends at e; depth_hist[d] += 1 for every interior node of depth d (counted per
// block in LDS first: tens of thousands of nodes share a dozen depths, and a returning atomic on one word is slow)
__global__ __launch_bounds__(256) void leaf_ends_kernel(const BuildNode* __restrict__ nodes, const BuildState* __restrict__ st,
                                                        int32_t* __restrict__ leaf_hist, int32_t* __restrict__ depth_hist)
{
    __shared__ int local[kMaxLevels + 2];
    for (int d = threadIdx.x; d < kMaxLevels + 2; d += blockDim.x) local[d] = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < st->total) {
        const BuildNode& nd = nodes[k];
        if (nd.child_a < 0) atomicAdd(&leaf_hist[nd.first + nd.count], 1);
        else atomicAdd(&local[nd.depth], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kMaxLevels + 2; d += blockDim.x) if (local[d]) atomicAdd(&depth_hist[d], local[d]);
}

// several small clears in one launch (each hipMemsetAsync is a launch of its own): range r = words[r] 32-bit words of value[r]
struct ClearRanges { uint32_t* ptr[7]; unsigned long long words[7]; uint32_t value[7]; int count; };
__global__ void clear_kernel(const ClearRanges c)
{
    for (int r = 0; r < c.count; r++)
        for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < c.words[r]; i += (unsigned long long)gridDim.x * blockDim.x)
            c.ptr[r][i] = c.value[r];
}

// traversal entry of a node (rt_device_types.h): interior -> its record index, leaf -> flag | count | first slot
__device__ __forceinline__ int32_t entry_of(const BuildNode& nd, const int32_t* ends_before, const int32_t* leaf_ends_before, int32_t node_base,
                                            int32_t slot_base)
{
    if (nd.child_a < 0) return rt::kLeafFlag | ((nd.count <= 30 ? nd.count : 31) << rt::kSlotBits) | (slot_base + nd.first);
    const int pre = nd.count > 0 ? preorder_of(nd, ends_before) : 0;
    return node_base + (pre - leaf_ends_before[nd.first + 1]);
}

// interior records (both children's boxes + both child entries), leaf counts, the refit schedule (interior records grouped
// by level, deepest first: level_start[d] = first schedule position of depth d), and the root entry
__global__ __launch_bounds__(256) void emit_nodes_kernel(const BuildNode* __restrict__ nodes, const BuildState* __restrict__ st,
                                  const int32_t* __restrict__ ends_before, const int32_t* __restrict__ leaf_ends_before, int32_t node_base,
                                  int32_t slot_base, float4* __restrict__ records, int32_t* __restrict__ leaf_count,
                                  int32_t* __restrict__ level_cursor, int32_t* __restrict__ sched, int32_t* __restrict__ root_entry,
                                  int32_t* __restrict__ mesh_flag)
{
    __shared__ int local[kMaxLevels + 2], base[kMaxLevels + 2];  // this block's interior nodes per depth -> one cursor bump per depth
    for (int d = threadIdx.x; d < kMaxLevels + 2; d += blockDim.x) local[d] = 0;
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t entry = 0, depth = -1, rank = 0;
    if (k < st->total) {
        const BuildNode& nd = nodes[k];
        entry = entry_of(nd, ends_before, leaf_ends_before, node_base, slot_base);
        if (k == 0) *root_entry = entry;
        if (nd.child_a < 0) {
            if (nd.count > 0) leaf_count[slot_base + nd.first] = nd.count;
        } else {
            const BuildNode &a = nodes[nd.child_a], &b = nodes[nd.child_b];
            float4* q = records + (size_t)entry * 4;
            q[0] = make_float4(a.mn[0], a.mn[1], a.mn[2], a.mx[0]);
            q[1] = make_float4(a.mx[1], a.mx[2], b.mn[0], b.mn[1]);
            q[2] = make_float4(b.mn[2], b.mx[0], b.mx[1], b.mx[2]);
            q[3] = make_float4(__int_as_float(entry_of(a, ends_before, leaf_ends_before, node_base, slot_base)),
                               __int_as_float(entry_of(b, ends_before, leaf_ends_before, node_base, slot_base)), 0.0f, 0.0f);
            const float w[12] = {a.mn[0], a.mn[1], a.mn[2], a.mx[0], a.mx[1], a.mx[2], b.mn[0], b.mn[1], b.mn[2], b.mx[0], b.mx[1], b.mx[2]};
            if (!rt::boxes_ordered(w)) *mesh_flag = rt::kBoxUnordered;     // (the octant-specialised slab test is not for this mesh)
            depth = nd.depth;
            rank = atomicAdd(&local[depth], 1);
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kMaxLevels + 2; d += blockDim.x) if (local[d]) base[d] = atomicAdd(&level_cursor[d], local[d]);
    __syncthreads();
    if (depth >= 0) sched[base[depth] + rank] = entry;
}

// every instance of the rebuilt mesh gets the new root entry and uv mode (what rt_scene_upload puts into its record)
__global__ void patch_instances_kernel(rt::DevInstance* instances, int count, int32_t mesh_index, const int32_t* __restrict__ root_entry,
                                       const int32_t* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || instances[i].mesh_index != mesh_index) return;
    instances[i].root_ref = *root_entry;
    instances[i].exact_uv = *flags & 1;
}

// triangle records, uvs and ids in leaf order (the arithmetic of rt_scene_upload / refit_triangles_kernel); flags[0] |= 1 when a
// uv value is not an ordinary number (the mesh then interpolates uv per candidate, raycast.cu:96)
__global__ void emit_triangles_kernel(const int32_t* __restrict__ order, int n, int32_t slot_base, const float* __restrict__ vertices,
                                      const float* __restrict__ normals, const float* __restrict__ uvs, float4* __restrict__ records,
                                      float* __restrict__ tri_uv, int32_t* __restrict__ tri_id, int32_t* __restrict__ flags)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int t = order[p];
    const size_t slot = (size_t)slot_base + (size_t)p;
    const float* v = vertices + 9 * (size_t)t;
    const float* nn = normals + 3 * (size_t)t;
    const rt::V3 v0 = rt::v3(v[0], v[1], v[2]), v1 = rt::v3(v[3], v[4], v[5]), v2 = rt::v3(v[6], v[7], v[8]);
    const rt::V3 e0 = v2 - v0, e1 = v1 - v0;                    // TrianglePrimitive.hpp:154-155
    const float d00 = rt::dot(e0, e0), d01 = rt::dot(e0, e1), d11 = rt::dot(e1, e1);
    const float inv = 1.0f / (d00 * d11 - d01 * d01);          // TrianglePrimitive.hpp:164
    float4* q = records + slot * 4;
    q[0] = make_float4(v0.x, v0.y, v0.z, nn[0]);
    q[1] = make_float4(nn[1], nn[2], e0.x, e0.y);
    q[2] = make_float4(e0.z, e1.x, e1.y, e1.z);
    q[3] = make_float4(d00, d01, d11, inv);
    bool odd = false;
    for (int c = 0; c < 6; c++) {
        const float u = uvs ? uvs[6 * (size_t)t + c] : 0.0f;
        tri_uv[slot * 6 + c] = u;
        if (!(fabsf(u) < 1e37f)) odd = true;
    }
    tri_id[slot] = t;
    if (odd) atomicOr(&flags[0], 1);
}

// level_cursor[d] <- first schedule position of depth d (deepest level first), from the per-depth counts; one thread
__global__ void level_starts_kernel(const int32_t* __restrict__ depth_hist, int32_t* __restrict__ level_cursor, int32_t* __restrict__ level_count_out)
{
    int at = 0;
    for (int d = kMaxLevels + 1; d >= 0; d--) { level_cursor[d] = at; at += depth_hist[d]; level_count_out[d] = depth_hist[d]; }
}

struct RebuildResult {              // what the host needs to know afterwards: one small read-back
    int32_t root_entry, flags;
    int32_t level_count[kMaxLevels + 2];
};

}  // namespace

extern "C" int rt_scene_rebuild_mesh_device(RtScene* s, int32_t mesh_index, const float* d_vertices, const float* d_normals, const float* d_uvs,
                                            int32_t n, void* stream_)
{
    if (!s) return RT_E_INVALID;
    std::lock_guard<std::recursive_mutex> scene_call_guard(s->call_mu);      // (rt_scene_internal.h: the scene's arrays do not move under this call)
    if (mesh_index < 0 || mesh_index >= (int)s->mesh_refit.size() || n < 0 || (n > 0 && (!d_vertices || !d_normals))) return RT_E_INVALID;
    RtScene::MeshRefit& rf = s->mesh_refit[(size_t)mesh_index];
    if (n > rf.slot_cap || (n > 0 && n - 1 > rf.int_cap)) return RT_E_INVALID;     // more triangles than the mesh was uploaded with
    hipStream_t stream = (hipStream_t)stream_;
    int rc = RT_OK;
    const int T = 256;
    std::lock_guard<std::mutex> lock(g_arena_mutex);
    const bool debug = getenv("RT_BVH_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    DeviceBuild b;
    if ((rc = build_core(d_vertices, true, n, 32, stream, false, b))) return rc;    // fill(1, 32), MeshPrimitive.cpp:54
    const double t1 = now();
    const int cap = b.cap;
    // scratch inside the arena: the range-end histogram (n + 2 ints, consumed by its scan) becomes the LEAF-end histogram, its
    // scan goes to hist_scan; the per-depth counters and the result record live in `extra`
    int32_t* leaf_hist = b.hist;
    int32_t* leaf_before = b.hist_scan;
    int32_t* depth_hist = (int32_t*)b.extra;                        // kMaxLevels + 2 ints
    int32_t* level_cursor = depth_hist + (kMaxLevels + 2);
    RebuildResult* d_result = (RebuildResult*)(level_cursor + (kMaxLevels + 2));
    static_assert(sizeof(RebuildResult) + 2 * (kMaxLevels + 2) * sizeof(int32_t) <= kExtraBytes, "scratch of the rebuild");
    RebuildResult res;
    // From here on kernels that use the shared build arena and write the scene's arrays are in flight on `stream`: an error
    // return must not release the arena (g_arena_mutex) to another build, nor hand the scene back, before they have finished.
    struct DrainOnError { hipStream_t s; bool armed; ~DrainOnError() { if (armed) (void)hipStreamSynchronize(s); } } drain{stream, true};
    {
        // the mesh's part of the scene arrays starts from zero (what rt_scene_upload leaves in unused records), tri_id from -1
        ClearRanges c;
        c.count = 7;
        c.ptr[0] = (uint32_t*)leaf_hist; c.words[0] = (unsigned long long)n + 2; c.value[0] = 0;
        c.ptr[1] = (uint32_t*)depth_hist; c.words[1] = (2 * (kMaxLevels + 2) * sizeof(int32_t) + sizeof(RebuildResult)) / 4; c.value[1] = 0;
        c.ptr[2] = (uint32_t*)(s->d_records + (size_t)rf.node_base * 4); c.words[2] = ((unsigned long long)rf.int_cap + rf.slot_cap) * 16; c.value[2] = 0;
        c.ptr[3] = (uint32_t*)(s->d_tri_uv + (size_t)rf.slot_base * 6); c.words[3] = (unsigned long long)rf.slot_cap * 6; c.value[3] = 0;
        c.ptr[4] = (uint32_t*)(s->d_tri_id + rf.slot_base); c.words[4] = (unsigned long long)rf.slot_cap; c.value[4] = 0xffffffffu;
        c.ptr[5] = (uint32_t*)(s->d_leaf_count + rf.slot_base); c.words[5] = (unsigned long long)rf.slot_cap; c.value[5] = 0;
        c.ptr[6] = (uint32_t*)(s->d_mesh_flags + mesh_index); c.words[6] = 1; c.value[6] = 0;
        hipLaunchKernelGGL(clear_kernel, dim3(2048), dim3(T), 0, stream, c);
    }
    hipLaunchKernelGGL(leaf_ends_kernel, dim3((cap + T - 1) / T), dim3(T), 0, stream, b.nodes, b.state, leaf_hist, depth_hist);
    RT_HIP(hipcub::DeviceScan::ExclusiveSum(b.tmp, b.tmp_bytes, leaf_hist, leaf_before, n + 2, stream));
    hipLaunchKernelGGL(level_starts_kernel, dim3(1), dim3(1), 0, stream, depth_hist, level_cursor, d_result->level_count);
    hipLaunchKernelGGL(emit_nodes_kernel, dim3((cap + T - 1) / T), dim3(T), 0, stream, b.nodes, b.state, b.ends_before, leaf_before, rf.node_base, rf.slot_base,
                       s->d_records, s->d_leaf_count, level_cursor, rf.d_sched, &d_result->root_entry, s->d_mesh_flags + mesh_index);
    if (n > 0)
        hipLaunchKernelGGL(emit_triangles_kernel, dim3((n + T - 1) / T), dim3(T), 0, stream, b.order, n, rf.slot_base, d_vertices, d_normals, d_uvs,
                           s->d_records, s->d_tri_uv, s->d_tri_id, &d_result->flags);
    if (!s->instances.empty())
        hipLaunchKernelGGL(patch_instances_kernel, dim3((unsigned)((s->instances.size() + T - 1) / T)), dim3(T), 0, stream, s->d_instances,
                           (int)s->instances.size(), mesh_index, &d_result->root_entry, &d_result->flags);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(&res, d_result, sizeof res, hipMemcpyDeviceToHost, stream));
    RT_HIP(hipStreamSynchronize(stream));
    drain.armed = false;
    if (debug) fprintf(stderr, "rebuild timing: build %.3f ms, emit into the scene %.3f ms\n", t1 - t0, now() - t1);
    // ---- host bookkeeping: what rt_scene_upload records for a mesh ----
    rf.num_triangles = n; rf.num_slots = n; rf.levels = b.st.levels;
    rf.level_end.clear();
    {
        int32_t at = 0;
        for (int d = kMaxLevels + 1; d >= 0; d--) if (res.level_count[d] > 0) { at += res.level_count[d]; rf.level_end.push_back(at); }
        rf.sched.clear();                                           // (the schedule now lives on the device only)
    }
    s->mesh_exact_uv[(size_t)mesh_index] = res.flags & 1;
    s->mesh_root_ref[(size_t)mesh_index] = res.root_entry;
    s->max_stack = 1;
    for (const auto& m : s->mesh_refit) s->max_stack = std::max(s->max_stack, m.levels);
    if (s->max_stack > rt::kMaxStack) { rc = RT_E_DEPTH; goto done; }
    for (DevInstance& in : s->instances)                            // the host mirror of what patch_instances_kernel wrote
        if (in.mesh_index == mesh_index) { in.root_ref = res.root_entry; in.exact_uv = res.flags & 1; }
done:
    return rc;
}
