// rt_kernels.hip -- the raycast hot path for MI355X (gfx950, wave64) and its C-ABI (include/rt_hip.h).
//
// Path replaced: render<<<>>> / cast_ray (raycast.cu:146-297 / :21-142) with everything they
// call (d_BVHTree::ray_intersects BVHTree.hpp:40-54, TrianglePrimitive::ray_intersect /
// point_inside TrianglePrimitive.hpp:62-79,151-185, the L0 math in utils.hpp / transforms.hpp).
// Results are bit-identical to the reference arithmetic (see rt_math.h); the memory layout and
// the execution mapping are not the reference's -- see DESIGN.md.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (mandatory: SURVEY.md H3) -fno-slp-vectorize (see _build.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_device_types.h"
#include "rt_math.h"
#include "rt_scene_internal.h"

using namespace rt;

// build-time switches of the traversal loop (A/B experiments only: RT_HIPCC_EXTRA=-DRT_OCTANTS=0 ...; the defaults are what ships)
#ifndef RT_OCTANTS
#define RT_OCTANTS 1            // octant-specialised loops, see trace_instance
#endif
#ifndef RT_WAIT_AT_FETCH
#define RT_WAIT_AT_FETCH 0      // 1: the vector fetch waits for its four loads where it issues them (measured +0.5 %), see trace_loop
#endif
#ifndef RT_SENTINEL
#define RT_SENTINEL 1           // the traversal stack starts with a sentinel entry: "stack empty" is a value popped, see trace_loop
#endif
#ifndef RT_STACK_WAVE_CHECK
#define RT_STACK_WAVE_CHECK 0   // 1: push / pop ask once per wave whether any lane is beyond the LDS part of the stack (measured +0.8 %), see StackT
#endif
#ifndef RT_LANE_MORTON
#define RT_LANE_MORTON 0        // 1: lanes of a wave cover their 8x8 pixels in Z-order (a quad of lanes = 2x2 pixels; measured +-0.3 %), see pixel_of
#endif
#ifndef RT_LEAF_FLAT
#define RT_LEAF_FLAT 1          // leaf step without the empty-leaf region; the wave's "any accept" asked inside the leaf region
#endif
#ifndef RT_EX_RECOMPUTE
#define RT_EX_RECOMPUTE 0       // 1: render_ex_kernel derives base colour, normal and cosine of a shaded hit again after its shadow cast instead of keeping
                                // them across it (measured +1.1 % on c3: profiles/r05_experiments/ex_spill_variants.md)
#endif
#ifndef RT_EX_PEEL
#define RT_EX_PEEL 1            // render_ex_kernel: the primary ray's depth written out before the bounce loop (measured -0.7 % on c3, same log)
#endif
#ifndef RT_OPTIMISTIC_STACK
#define RT_OPTIMISTIC_STACK 1   // deep trees: the timed kernels run the LDS-only stack and a lane whose stack would outgrow it starts again on the
                                // general stack afterwards (render_pixel), instead of every push and pop asking "which memory"
#endif
// Measured and off (round 6, profiles/r06_experiments/ex_secondary_asm.md): the bounce kernel's secondary rays through the hand-written loop
// (bit 0: shadow rays, bit 1: bounce rays; trace_instance's SEC) and the kernel compiled for fewer waves per SIMD (8: 64 registers,
// 7: 72, 6: 80).  The loop saves instructions (-4.5 % vector, -16 % scalar on c3 with bit 1) and 5 % of the frame at EQUAL occupancy,
// but at eight waves its 53 registers push the path state into scratch (writes x 2.5, fetches x 10: c3 +5 %), and every wave given up
// for registers costs more than the loop returns (c3: 8 waves 19.4 ms, 7: 19.8, 6: 20.9; 6 waves with the loop: 19.8).
#ifndef RT_EX_BOUNCE_WAVES
#define RT_EX_BOUNCE_WAVES 8
#endif
#ifndef RT_EX_SECONDARY_ASM
#define RT_EX_SECONDARY_ASM 0
#endif
#ifndef RT_EX_PRIMARY_ASM
#define RT_EX_PRIMARY_ASM 1     // the bounce kernel's camera ray through the hand-written loop and the view records
#endif
#ifndef RT_NEED_POP_VALUE
#define RT_NEED_POP_VALUE 1     // "this lane must pop" is a value of `cur` (kNeedPop), not a flag merged across the loop's branches
#endif

// =====================================================================================
//                                      device code
// =====================================================================================

namespace {

constexpr int kBlock = 256;     // the extension kernels' workgroup: one 16x16-pixel tile = 2x2 wave tiles of 8x8 pixels
constexpr int kTile = 16;
// The primary kernels' workgroup is ONE wave = one 8x8-pixel tile (round 5).  A 256-thread workgroup keeps its LDS block until its
// last wave has finished, and no new workgroup fits a CU whose LDS is taken by eight of them: the slots of waves that finished
// early stood empty -- SQ_WAVE_CYCLES put the average residency at 6.7 of 8 waves per SIMD.  With one wave per workgroup a slot is
// refilled the moment its wave ends: c2 far / mid / near -7.7 % / -3.4 % / -4.2 % (profiles/r05_experiments/asm_loop_ab.log).
// (Round 2 measured this form at +-2 %: the kernel of that round was bound by its instruction count, not by waves waiting.)
constexpr int kPrimBlock = 64;
constexpr int kPrimTile = 8;

struct Hit {
    float min;                  // HitInfo::min, raycast.cu:12
    int32_t slot;               // triangle slot of the accepted hit
    int32_t instance;
    float u, v;                 // barycentrics of the accepted hit (uv is interpolated once, at shade time) -- or, for a mesh in the
                                // exact-uv mode, the interpolated uv itself (the two never coexist: two registers, not four)
    V3 loc;                     // world-space location of the accepted hit (extension kernel only)
};

// d_BVHTree::ray_intersects, BVHTree.hpp:40-54, from the six differences (box min - origin, box max - origin):
// the subtraction is done where the record is fetched, see trace_instance.
__device__ __forceinline__ float slab(float dnx, float dny, float dnz, float dxx, float dxy, float dxz, V3 dinv)
{
    float tminx = dnx * dinv.x, tminy = dny * dinv.y, tminz = dnz * dinv.z;
    float tmaxx = dxx * dinv.x, tmaxy = dxy * dinv.y, tmaxz = dxz * dinv.z;
    float t1x = fminf(tminx, tmaxx), t1y = fminf(tminy, tmaxy), t1z = fminf(tminz, tmaxz);
    float t2x = fmaxf(tminx, tmaxx), t2y = fmaxf(tminy, tmaxy), t2z = fmaxf(tminz, tmaxz);
    float dst_far = fminf(fminf(t2x, t2y), t2z);
    float dst_near = fmaxf(fmaxf(t1x, t1y), t1z);
    bool hit = dst_far >= dst_near && dst_far > 0.0f;
    return hit ? dst_near : FLT_MAX;
}

// The same test for a ray whose sign octant is known at compile time (OCT bit k set = dinv component k negative): with
// box.min <= box.max, a finite origin and a finite non-zero dinv, rounding is monotone, so fl((min-o)*dinv) <= fl((max-o)*dinv)
// for dinv > 0 and the reverse for dinv < 0 -- which of the two products is the near one and which the far one is a choice
// of operand per axis, not an instruction.  The values equal fminf / fmaxf of the pair except for the sign of a zero, which
// no compare below can see.  (12 VALU fewer per interior node; trace_instance checks the preconditions per wave.)
template <int OCT>
__device__ __forceinline__ float slab_oct(float dnx, float dny, float dnz, float dxx, float dxy, float dxz, V3 dinv)
{
    float t1x = ((OCT & 1) ? dxx : dnx) * dinv.x, t2x = ((OCT & 1) ? dnx : dxx) * dinv.x;
    float t1y = ((OCT & 2) ? dxy : dny) * dinv.y, t2y = ((OCT & 2) ? dny : dxy) * dinv.y;
    float t1z = ((OCT & 4) ? dxz : dnz) * dinv.z, t2z = ((OCT & 4) ? dnz : dxz) * dinv.z;
    float dst_far = fminf(fminf(t2x, t2y), t2z);
    float dst_near = fmaxf(fmaxf(t1x, t1y), t1z);
    bool hit = dst_far >= dst_near && dst_far > 0.0f;
    return hit ? dst_near : FLT_MAX;
}

// The first twelve words of an interior record are the two child boxes (min xyz, max xyz each); this turns them into
// box - origin, written over the record's registers q0..q2.  W = the record itself, or the 16-word scalar-register
// vector of a wave-uniform fetch: the subtraction then takes its box operand from the scalar register directly and the
// record never has to be copied into vector registers.
template <class W>
__device__ __forceinline__ void box_differences(const W& w, V3 o, float4& q0, float4& q1, float4& q2)
{
    q0 = make_float4(w[0] - o.x, w[1] - o.y, w[2] - o.z, w[3] - o.x);
    q1 = make_float4(w[4] - o.y, w[5] - o.z, w[6] - o.x, w[7] - o.y);
    q2 = make_float4(w[8] - o.z, w[9] - o.x, w[10] - o.y, w[11] - o.z);
}

// Primary ray direction of pixel (x, y): raycast.cu:159-188
__device__ __forceinline__ V3 camera_direction(const FrameParams& p, float fx, float fy)
{
    // apply_matrix(K_inv, (x, y, 1)), utils.hpp:134-140
    float a = p.kinv[0] * fx + p.kinv[1] * fy + p.kinv[2] * 1.0f;
    float b = p.kinv[3] * fx + p.kinv[4] * fy + p.kinv[5] * 1.0f;
    float c = p.kinv[6] * fx + p.kinv[7] * fy + p.kinv[8] * 1.0f;
    float radius = sqrtf(a * a + b * b);
    float theta = atanf_fdlibm(radius);
    // raycast.cu:172 -- float products, double sum, double outer product, narrowed to float
    float thetad = (float)((double)theta * (1.0 + (double)(p.D[0] * theta) + (double)(p.D[1] * theta * theta)
                   + (double)(p.D[2] * theta * theta * theta) + (double)(p.D[3] * theta * theta * theta * theta)));
    float scale = thetad / radius;
    V3 d = normalize(v3(scale * a, scale * b, c));
    d = v3(d.x, d.z, -d.y);                                     // raycast.cu:182
    d = apply_quat(p.q_cam, d);                                 // raycast.cu:185
    return normalize(d);                                        // raycast.cu:188
}

template <bool DEBUG>
struct Counters {
    int pops = 0, aabb = 0, tris = 0, inside = 0;
};
template <>
struct Counters<false> {};

// Per-lane traversal stack (raycast.cu:54-61).  The first `lds_depth` entries live in LDS, one column per lane
// ([entry][kBlock] ints: a wave's accesses are conflict-free); deeper entries -- rare: the tree may be 28+ levels
// deep but rays seldom hold more than a dozen postponed nodes -- spill to a private (scratch) array.  Keeping the
// LDS part at 16 entries (+ the sentinel's row) lets 8 waves/SIMD stay resident (17 KB per 256-thread workgroup, 136 of 160 KB).
constexpr int kLdsStack = 16;
// Two entries no tree contains: leaf references whose slot field is beyond every slot a scene can hold (rt_scene_upload and the
// rebuild keep slot_base + n + 1 <= kSlotMask, so the largest first-slot of a leaf is kSlotMask - 2).  Chosen among the
// hardware's inline integer constants (-16 .. 64): compares and selects against them need no register.
// kSentinel: with RT_SENTINEL the first entry of every traversal; popping it ends the loop.
// kNeedPop: what interior_apply / the leaf step leave in `cur` when the lane has to pop.
constexpr int32_t kSentinel = -2;               // = leaf flag | count 31 | slot kSlotMask - 1
constexpr int32_t kNeedPop = -1;                // = leaf flag | count 31 | slot kSlotMask
static_assert((kSentinel & kSlotMask) == kSlotMask - 1 && (kNeedPop & kSlotMask) == kSlotMask, "beyond the slot space");
// rows of a workgroup's LDS stack block ([row][thread] ints): the postponed nodes kept in LDS, plus the sentinel's row
__host__ __device__ inline int lds_rows(int stack_depth) { return (stack_depth < kLdsStack ? stack_depth : kLdsStack) + RT_SENTINEL; }
typedef __attribute__((address_space(3))) int lds_int;      // typed LDS pointer: keeps stack traffic on ds_read/ds_write
// SPILL = false: the tree is shallow enough for the LDS part alone (a tree of L levels never holds more than L - 1 postponed
// nodes: the entries of a stack sit at strictly increasing levels below the root) -- no private array, and neither push
// nor pop carries the "which memory" branch (5 + 4 scalar instructions of exec-mask bookkeeping per iteration).
// OPTIMISTIC (with SPILL = false, for trees that ARE deeper than the LDS part): the block has one more row, which takes the
// push that does not fit; interior_apply then hands the lane the sentinel, the lane leaves its loop with sp != 0, and the
// caller traces that ray again from the start on a SPILL stack.  Which rays need more than 16 postponed nodes depends on the
// view (none of the three c2 cameras has one in a 28-level tree), so the common case pays two vector instructions per push
// instead of nine scalar ones per iteration.
template <int STRIDE, bool SPILL = true, bool OPTIMISTIC = false>   // STRIDE = threads per workgroup (ints between two entries of a lane)
struct StackT {
    static constexpr bool kOptimistic = OPTIMISTIC;
    static constexpr bool kSpill = SPILL;
    static constexpr int kStride = STRIDE;
    static_assert(!(SPILL && OPTIMISTIC), "the optimistic stack is the LDS-only one");
    lds_int* lds;               // this lane's LDS column
    int* spill;                 // this lane's private overflow, kMaxStack - kLdsStack entries
    int lds_depth;              // entries kept in LDS: lds_rows(stack_depth)
    int sp;
    // (RT_STACK_WAVE_CHECK: whether ANY lane of the wave is beyond the LDS part is asked once per wave -- a scalar branch; the
    // per-lane "which memory" regions then only run when one is.  Deep entries are rare (the c2 tree has 28 levels and no
    // ray of its three cameras holds more than 16 postponed nodes), but the ballot costs what the regions cost: measured
    // +0.8 % on c2, off.)
    __device__ __forceinline__ void push(int32_t v)
    {
        if constexpr (SPILL && RT_STACK_WAVE_CHECK) {
            if (__builtin_amdgcn_ballot_w64(sp >= lds_depth) == 0ull) lds[sp * STRIDE] = v;
            else if (sp < lds_depth) lds[sp * STRIDE] = v; else spill[sp - lds_depth] = v;
        } else if constexpr (SPILL) { if (sp < lds_depth) lds[sp * STRIDE] = v; else spill[sp - lds_depth] = v; }
        else lds[sp * STRIDE] = v;
        sp++;
    }
    __device__ __forceinline__ int32_t pop()
    {
        --sp;
        if constexpr (!SPILL) return lds[sp * STRIDE];
        if constexpr (RT_STACK_WAVE_CHECK) { if (__builtin_amdgcn_ballot_w64(sp >= lds_depth) == 0ull) return lds[sp * STRIDE]; }
        // always an LDS read (index clamped) and, rarely, a private read on top: a select between the two
        // address spaces would turn into one slow flat_load
        int32_t v = lds[(sp < lds_depth ? sp : lds_depth - 1) * STRIDE];
        if (sp >= lds_depth) v = spill[sp - lds_depth];
        return v;
    }
};

typedef StackT<kBlock> Stack;

// Ray in mesh space (raycast.cu:33-51) plus what the leaf code needs of the instance.
struct MeshRay {
    V3 ro, rd, dinv;
};

__device__ __forceinline__ MeshRay to_mesh_space(const DevInstance& in, V3 org, V3 dir)
{
    MeshRay r;
    r.rd = apply_quat(in.q_rot, dir);
    r.rd.x *= in.inv_scale[0]; r.rd.y *= in.inv_scale[1]; r.rd.z *= in.inv_scale[2];
    r.ro = apply_quat(in.q_pose, v3(org.x - in.pose_xyz[0], org.y - in.pose_xyz[1], org.z - in.pose_xyz[2]));
    r.ro.x *= in.inv_scale[0]; r.ro.y *= in.inv_scale[1]; r.ro.z *= in.inv_scale[2];
    r.dinv = v3(1.0f / r.rd.x, 1.0f / r.rd.y, 1.0f / r.rd.z);  // Ray.hpp:21
    return r;
}

// One interior node (raycast.cu:66-79) from its already fetched 64-B record: tests both children, pushes the
// far one if it passes `dist < hit.min`, and leaves in `cur` the entry the reference would pop next (the entry
// pushed last never goes through the stack).  Returns false when nothing was pushed.
// W: the record's first fourteen words, box words already as box - origin: an array of the lane's registers, or the scalar-register
// vector of a wave-uniform fetch of a VIEW record (the products of the slab test then take their box operand from the scalar
// register: no copy, no subtraction).
template <bool DEBUG, class STK, int OCT = -1, bool NEED_POP = false, class W>
__device__ __forceinline__ bool interior_apply_words(const W& w, const MeshRay& r, float hit_min,
                                                     int32_t& cur, STK& stack, Counters<DEBUG>& cnt)
{
    int32_t ra = __float_as_int(w[12]), rb = __float_as_int(w[13]);
    if constexpr (DEBUG) cnt.aabb += 2;
    float da, db;                                                       // w[0..11]: box - origin (box_differences, or a view record)
    if constexpr (OCT < 0) {
        da = slab(w[0], w[1], w[2], w[3], w[4], w[5], r.dinv);
        db = slab(w[6], w[7], w[8], w[9], w[10], w[11], r.dinv);
    } else {
        da = slab_oct<OCT>(w[0], w[1], w[2], w[3], w[4], w[5], r.dinv);
        db = slab_oct<OCT>(w[6], w[7], w[8], w[9], w[10], w[11], r.dinv);
    }
    // push order of raycast.cu:72-79: the farther child is pushed first, the nearer one last (= popped next); each only if
    // its distance passes `dist < hit.min`.  With pa / pb = "child a / b passes": both pass -> push the far one (b when
    // da < db, else a -- a tie takes the else branch) and go on with the near one; one passes -> go on with that one (it is
    // the nearer: the other failed the same bound); none -> the caller pops.  (da, db are never NaN: slab returns a
    // distance or FLT_MAX; a NaN hit_min fails every compare, as in the reference.)
    const bool a_near = da < db;
    const bool pa = da < hit_min, pb = db < hit_min;
    int32_t next = pa ? ra : rb;
    if (pa && pb) {
        stack.push(a_near ? rb : ra);                           // the only entry that really goes through the stack
        next = a_near ? ra : rb;
        // (optimistic stack: that push went to the spare row -- this lane is done here and will be traced again, see StackT)
        if constexpr (STK::kOptimistic) next = stack.sp > stack.lds_depth ? kSentinel : next;
    }
    cur = NEED_POP ? ((pa | pb) ? next : kNeedPop) : next;      // (without NEED_POP: not used when neither passed, the caller pops)
    return pa || pb;
}

template <bool DEBUG, class STK, int OCT = -1, bool NEED_POP = false>
__device__ __forceinline__ bool interior_apply(float4 q0, float4 q1, float4 q2, float4 q3, const MeshRay& r, float hit_min,
                                               int32_t& cur, STK& stack, Counters<DEBUG>& cnt)
{
    const float w[14] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y};
    return interior_apply_words<DEBUG, STK, OCT, NEED_POP>(w, r, hit_min, cur, stack, cnt);
}

// What a triangle test proposes as the new closest hit.  Deliberately left uninitialised by the callers: it is only
// read where triangle_test returned true.
struct Candidate {
    float dist, u, v;
    float2 uv;
    V3 loc;
};

// One triangle of a leaf (one iteration of raycast.cu:85-136) from its already fetched 64-B record t0..t3.
// Returns whether the reference would accept it as the new closest hit (raycast.cu:109) and fills `c` in that case;
// the caller applies the update with selects at the top level of its loop, which keeps the hit record out of the
// control flow (as branch-merged values it cost two register copies per field per loop iteration).
template <bool DEBUG, bool EX>
__device__ __forceinline__ bool triangle_test(const RenderParams& p, const DevInstance& in, bool exact_uv, bool identity_inv,
                                              const MeshRay& r, V3 org, int slot, float hit_min, Counters<DEBUG>& cnt,
                                              float4 t0, float4 t1, float4 t2, float4 t3, Candidate& c)
{
    if constexpr (DEBUG) cnt.tris++;
    V3 v0 = v3(t0.x, t0.y, t0.z), nrm = v3(t0.w, t1.x, t1.y);
    // The reference's chain of early returns (TrianglePrimitive.hpp:66,72, raycast.cu:91,96) is evaluated as
    // one predicate over straight-line code: a wave almost always has some lane that passes each test, so
    // branching per test only adds exec-mask bookkeeping.  Lanes whose predicate is already false compute
    // on garbage that is never used.
    // TrianglePrimitive::ray_intersect, TrianglePrimitive.hpp:62-79
    float denom = dot(r.rd, nrm);
    // `abs(denom) < 1e-6` compares in double; 0x358637be is the smallest float whose double value is >= 1e-6,
    // so this float compare selects exactly the same floats (tests/test_host_logic.py checks the boundary).
    bool ok = !(fabsf(denom) < __int_as_float(0x358637be));
    // A candidate is only ever accepted when same_dir = denom < 0 (raycast.cu:107-109); for denom >= 0 (or NaN)
    // the rest of the test has no observable effect, so the production kernel drops it here.  The debug kernel
    // goes on because the inside-hit count of raycast.cu:96 is one of the parity planes.
    if constexpr (!DEBUG) ok = ok && (denom < 0.0f);
    float tt = dot(v0 - r.ro, nrm) / denom;
    ok = ok && !(tt < 0.0f);
    V3 pt = r.ro + tt * r.rd;
    ok = ok && !(pt.x == FLT_MAX);                          // raycast.cu:91
    // TrianglePrimitive::point_inside, TrianglePrimitive.hpp:151-185
    V3 e0 = v3(t1.z, t1.w, t2.x), e1 = v3(t2.y, t2.z, t2.w);
    V3 e2 = pt - v0;
    float dot02 = dot(e0, e2), dot12 = dot(e1, e2);
    float u = (t3.z * dot02 - t3.y * dot12) * t3.w;
    float v = (t3.x * dot12 - t3.y * dot02) * t3.w;
    ok = ok && (u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f);
    c.u = u; c.v = v;
    if (exact_uv) {
        c.uv = make_float2(0.0f, 0.0f);
        if (ok) {                                           // raycast.cu:96 can only fail for absurd uv data
            const float* q = p.tri_uv + (size_t)slot * 6;
            float w = 1.0f - u - v;
            c.uv.x = (w * q[0] + v * q[2]) + u * q[4];
            c.uv.y = (w * q[1] + v * q[3]) + u * q[5];
            ok = c.uv.x != FLT_MAX;
        }
    }
    bool accept = false;
    if (ok) {
        if constexpr (DEBUG) cnt.inside++;
        // raycast.cu:98-104.  For an instance whose mesh -> world transform is exactly the identity (scale 1,
        // translation 0, quaternion (1,0,0,0): the common case) the scale / translate / rotate sequence returns
        // pt itself up to the sign of zero components, which the squares in magnitude() cannot see -- so the
        // production kernel skips it.  (The extension kernel keeps it: it stores `loc`.)
        V3 loc = pt;
        if (EX || !identity_inv) {
            loc = v3(pt.x * in.scale[0], pt.y * in.scale[1], pt.z * in.scale[2]);
            loc = apply_quat(in.q_inv_pose, v3(loc.x - in.inv_pose_xyz[0], loc.y - in.inv_pose_xyz[1], loc.z - in.inv_pose_xyz[2]));
        }
        c.dist = magnitude(loc - org);
        if constexpr (EX) c.loc = loc;
        // raycast.cu:107-109: same_dir = dot(r_ray.direction, normal) is `denom`
        // (bitwise: as `&&` / `||` the compiler builds a chain of exec-mask regions for the three compares)
        accept = (denom < 0) & ((hit_min == FLT_MAX) | (c.dist < hit_min));
    }
    return accept;
}

// One instance of raycast.cu:26-139.
//
// Interior nodes and triangles are both 64-B records of ONE array, so every lane issues the same four 16-B loads at
// `records + (entry << 6)` ("unified fetch") and the wave waits for memory once per iteration, whatever mix of
// interior and leaf entries its lanes hold.  Each lane still visits exactly the reference's sequence of nodes.
// PROF = diagnostic copy with s_memtime stamps per phase (RT_TRACE_FILE); its frames are never timed.
// COUNT: *iters counts this lane's loop iterations (the cost measure behind the heavy-first dispatch order).
// POPS (the extension kernel): *pops counts the lane's node pops, the one visit count that kernel reports.
// OCT >= 0: every lane of the wave is known to hold a ray of sign octant OCT that meets slab_oct's preconditions.
// ANYHIT (the extension's shadow rays): raycast.cu:129-133 restored -- cast_ray(..., lighting_pass = true, light_distance =
// FLT_MAX) returns at the first accepted hit whose distance is below light_distance.
// VIEW (primary rays of render_kernel<.., VIEW>): interior records are read from the frame's view records, `vdelta` bytes behind the
// record itself, whose box words already are box - r.ro (view_records_kernel: the same subtraction, done once per frame and
// instance instead of per visit and lane); an iteration in which the whole wave holds the same interior node takes them as
// scalar operands and skips the leaf half of the loop altogether.
template <bool DEBUG, bool PROF, bool EX, bool COUNT, class STK, bool POPS, int OCT, bool ANYHIT = false, bool VIEW = false>
__device__ __forceinline__ void trace_loop(const RenderParams& p, const DevInstance& in, int inst_index, const MeshRay& r,
                                           V3 org, STK& stack, Hit& hit, Counters<DEBUG>& cnt, int* iters, int* pops, uint32_t vdelta = 0)
{
    static_assert(!VIEW || (!DEBUG && !PROF && !EX && RT_SENTINEL && RT_LEAF_FLAT), "view records: the timed primary kernels only");
    stack.sp = 0;
#if RT_SENTINEL
    // The stack starts with a sentinel (raycast.cu:58 starts it with the root, which here stays in a register): "pop from an
    // empty stack" (raycast.cu:60) is then an ordinary pop that returns the sentinel, and the loop ends on the VALUE popped --
    // one compare for all lanes at the bottom of the loop instead of an `if (sp == 0) break` nested in the pop branch, which
    // cost ten scalar instructions of exec-mask bookkeeping per iteration.
    stack.push(kSentinel);
#endif
    int32_t cur = in.root_ref;                                  // raycast.cu:58 (kept in a register)
    bool have = true;
    // "this lane must pop" as a value of `cur` instead of the flag `have`: -1.5 % on the primary kernel, +0.8 % on the bounce
    // casts of the extension kernel (EX: they keep the flag) -- profiles/r04_experiments/sentinel_loop_ab.log
    constexpr bool kNeedPopValue = RT_SENTINEL && RT_NEED_POP_VALUE && !EX;
    int rem = -1;                                               // triangles left in the leaf being walked, -1 = not in a leaf
    unsigned long long c_pop = 0, c_mem = 0, c_int = 0, c_leaf = 0, n_it = 0, n_int = 0, n_leaf = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    unsigned long long n_g1 = 0, n_g2 = 0, n_g34 = 0;
    // One iteration = one interior node or ONE triangle of a leaf (a leaf with k triangles takes k iterations and
    // keeps `cur` pointing at its next slot), so the loop has no inner loop and every record -- node or triangle,
    // first or later -- comes through the same fetch below.
    // The loop is bottom-tested (the pop and the stack-empty exit of raycast.cu:60-61 close the iteration) so that the
    // hit record leaves the loop after its update: with the exit at the top it was live across the back edge in two
    // copies.
    const bool exact_uv = in.exact_uv != 0, identity_inv = in.identity_inv != 0;    // (read once, not per triangle)
    do {
        if constexpr (PROF) t1 = __builtin_amdgcn_s_memtime();
        if constexpr (COUNT) (*iters)++;
        const bool interior = cur >= 0;
        if constexpr (DEBUG) cnt.pops += (interior || rem < 0) ? 1 : 0;
        if constexpr (POPS) *pops += (interior || rem < 0) ? 1 : 0;
        // About half of all wave iterations (three quarters for close-up views) find every active lane holding
        // the SAME entry -- coherent rays walk the top of the tree in lockstep.  Those iterations fetch the record
        // once per wave through the scalar cache (s_load_dwordx16) instead of 64 x 64 B through the vector memory
        // path, which is otherwise the busiest unit of the kernel (-9..-13 % frame time).
        float4 r0, r1, r2, r3;          // the record; for interior lanes r0..r2 become child boxes - ray origin
        const int32_t cur0 = __builtin_amdgcn_readfirstlane(cur);
        typedef float f16v __attribute__((ext_vector_type(16)));
        if constexpr (VIEW) {
            const bool one_entry = __ballot(cur != cur0) == 0ull;   // (wave-uniform) every lane holds the same entry
            if (one_entry && cur0 >= 0) {
                // The whole wave at one interior node: its VIEW record (box words = box - r.ro already) through the scalar cache,
                // used where it arrives -- as scalar operands of the slab products.  The iteration closes itself: no lane is at a leaf.
                f16v w;
                const uint32_t off = ((uint32_t)cur0 << 6) + vdelta;
                asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p.records), "s"(off));
                have = interior_apply_words<DEBUG, STK, OCT, kNeedPopValue>(w, r, hit.min, cur, stack, cnt);
                if (kNeedPopValue ? cur == kNeedPop : !have) cur = stack.pop();
                __builtin_amdgcn_wave_barrier();
                continue;                                       // (to the bottom test: the sentinel may have been popped)
            }
            if (one_entry) {
                // the whole wave at one triangle record: copied for the code the per-lane fetch shares (through an OR with a zero
                // the optimiser cannot see through: plain copies are hoisted above the branch and run in every iteration)
                f16v w;
                const uint32_t off = (uint32_t)cur0 << 6;
                asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p.records), "s"(off));
                int z;
                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                auto keep = [z](float f) { return __int_as_float(__float_as_int(f) | z); };
                r0 = make_float4(keep(w[0]), keep(w[1]), keep(w[2]), keep(w[3]));
                r1 = make_float4(keep(w[4]), keep(w[5]), keep(w[6]), keep(w[7]));
                r2 = make_float4(keep(w[8]), keep(w[9]), keep(w[10]), keep(w[11]));
                r3 = make_float4(keep(w[12]), keep(w[13]), keep(w[14]), keep(w[15]));
            } else {
                const uint32_t boff = ((uint32_t)cur << 6) + (interior ? vdelta : 0u);
                const float4* rec = (const float4*)((const char*)p.records + boff);
                r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3];
            }
        } else
        if (!PROF && __ballot(cur != cur0) == 0ull) {
            f16v w;
            // inline asm: hipcc would otherwise merge this load with the per-lane one below into a single vector load.
            // No "memory" clobber: the records are read-only, and a clobber makes every other load in the loop
            // (instance fields, ...) repeat each iteration.
            // The record's byte offset goes into the instruction's scalar offset operand (no 64-bit address arithmetic):
            // entry << 6 drops the flag and count bits of a leaf reference and stays below 4 GiB (at most 2^26 - 2 records).
            const uint32_t off = (uint32_t)cur0 << 6;
            asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(p.records), "s"(off));
            if (cur0 >= 0) {                                    // (a scalar branch: the whole wave holds this interior node)
                box_differences(w, r.ro, r0, r1, r2);
            } else {
                // (a triangle record is used as it is.  The copies go through an OR with a zero the optimiser cannot
                // see through: as plain copies they are hoisted above the branch and run for interior nodes too.)
                int z;
                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                auto keep = [z](float f) { return __int_as_float(__float_as_int(f) | z); };
                r0 = make_float4(keep(w[0]), keep(w[1]), keep(w[2]), keep(w[3]));
                r1 = make_float4(keep(w[4]), keep(w[5]), keep(w[6]), keep(w[7]));
                r2 = make_float4(keep(w[8]), keep(w[9]), keep(w[10]), keep(w[11]));
            }
            r3 = make_float4(w[12], w[13], w[14], w[15]);
        } else {
            const float4* rec = p.records + (size_t)(cur & kSlotMask) * 4;     // node or triangle: one array, one index space
            r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3];
#if RT_WAIT_AT_FETCH
            // All four loads are waited for HERE (the empty asm uses a word of each): otherwise every later use of r0..r3 -- in
            // code that the wave-uniform path shares -- gets its own s_waitcnt vmcnt(n), ten no-ops per iteration on the
            // path that issued no vector load at all.  Measured: ten fewer instructions per iteration and 0.5 % MORE time (the
            // subtractions of box_differences no longer overlap the later loads): off, kept as a switch for the record.
            asm volatile("" : "+v"(r0.x), "+v"(r1.x), "+v"(r2.x), "+v"(r3.x));
#endif
            if (interior) {
                const float w[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                box_differences(w, r.ro, r0, r1, r2);
            }
        }
        if constexpr (PROF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t2 = __builtin_amdgcn_s_memtime();
            n_it++; n_int += __ballot(interior) != 0; n_leaf += __ballot(!interior) != 0;
            // how many DIFFERENT entries the wave's lanes hold in this iteration: one (the scalar-fetch case), two, three or four, more
            {
                unsigned long long left = __ballot(true);
                int groups = 0;
                while (left != 0ull && groups < 5) {
                    const int32_t c = __shfl(cur, __ffsll((long long)left) - 1);
                    left &= ~__ballot(cur == c);
                    groups++;
                }
                n_g1 += groups == 1; n_g2 += groups == 2; n_g34 += groups == 3 || groups == 4;
            }
        }
        if (interior) have = interior_apply<DEBUG, STK, OCT, kNeedPopValue>(r0, r1, r2, r3, r, hit.min, cur, stack, cnt);
        if constexpr (PROF) { __builtin_amdgcn_s_waitcnt(0); t3 = __builtin_amdgcn_s_memtime(); }
        bool accept = false;
        unsigned long long any_accept = 0ull;
        Candidate c;                                            // uninitialised on purpose, see Candidate
        const int slot = cur & kSlotMask;
        if (!interior) {                                        // leaf: contiguous triangle slots, raycast.cu:83-137
            const bool first = rem < 0;                         // first visit: decode the triangle count
            const int coded = (cur >> kSlotBits) & 31;
            rem = first ? coded : rem;
            if (first & (coded == 31)) rem = p.leaf_count[slot];    // (leaves of more than 30 triangles: rare)
            if constexpr (DEBUG || !RT_LEAF_FLAT) {             // (the instrumented kernel counts triangle tests: none for an empty leaf)
                if (rem > 0) accept = triangle_test<DEBUG, EX>(p, in, exact_uv, identity_inv, r, org, slot, hit.min, cnt, r0, r1, r2, r3, c);
            } else {
                // An empty leaf (degenerate splits only) holds no triangle to test: its "test" runs on whatever record sits at
                // the slot (always readable: the arrays are padded) and the result is masked -- one exec-mask region less
                // in every leaf iteration of every other scene.
                accept = triangle_test<DEBUG, EX>(p, in, exact_uv, identity_inv, r, org, slot, hit.min, cnt, r0, r1, r2, r3, c) & (rem > 0);
            }
            if constexpr (RT_LEAF_FLAT) any_accept = __builtin_amdgcn_ballot_w64(accept);   // (asked here, where `accept` is a compare result, not a merged flag)
            rem--;
            have = rem > 0;
            cur = have ? cur + 1 : (kNeedPopValue ? kNeedPop : cur);    // next slot of the same leaf (the slot field never overflows)
            rem = have ? rem : -1;
        }
        if constexpr (!RT_LEAF_FLAT) any_accept = __builtin_amdgcn_ballot_w64(accept);
        if (any_accept != 0ull) {                               // (wave-level: most iterations accept nothing)
            hit.min = accept ? c.dist : hit.min;
            hit.slot = accept ? slot : hit.slot;
            hit.instance = accept ? inst_index : hit.instance;
            hit.u = accept ? (exact_uv ? c.uv.x : c.u) : hit.u;
            hit.v = accept ? (exact_uv ? c.uv.y : c.v) : hit.v;
            if constexpr (EX) { hit.loc.x = accept ? c.loc.x : hit.loc.x; hit.loc.y = accept ? c.loc.y : hit.loc.y; hit.loc.z = accept ? c.loc.z : hit.loc.z; }
            // raycast.cu:129-133: `if (lighting_pass && distance < light_distance) return hit_info;` -- the lane is done with
            // this cast: it takes the sentinel, which ends its loop at the bottom test (and cast_ray_ex skips its other instances)
            if constexpr (ANYHIT) {
                const bool done = accept & (c.dist < FLT_MAX);
                if constexpr (kNeedPopValue) cur = done ? kSentinel : cur;
                else {
                    // (the flag forms of the A/B switches RT_SENTINEL=0 / RT_NEED_POP_VALUE=0: the lane leaves its leaf and finds
                    // nothing left but the sentinel -- or nothing at all -- when it pops next)
                    have = done ? false : have;
                    rem = done ? -1 : rem;
                    stack.sp = done ? RT_SENTINEL : stack.sp;
                }
            }
        }
        if constexpr (PROF) { __builtin_amdgcn_s_waitcnt(0); t0 = __builtin_amdgcn_s_memtime(); }
#if RT_SENTINEL
        // (kNeedPopValue: `have` is not read -- as a flag it is a lane mask that every branch of the loop has to merge into,
        // eight scalar instructions per iteration; as a value of `cur` it costs one select)
        if (kNeedPopValue ? cur == kNeedPop : !have) cur = stack.pop();     // raycast.cu:60-61 (the sentinel when nothing is left)
#else
        if (!have) {
            if (stack.sp == 0) break;
            cur = stack.pop();                                  // raycast.cu:61
        }
#endif
        // One latch for both ways round (popped / kept going): a convergent no-op that the optimiser may not clone.
        // Without it the two back edges are split into nested loops ("iterate while nobody pops" inside "pop"),
        // i.e. lanes that need a pop wait for every lane that does not: +50 % time on views with sky.
        __builtin_amdgcn_wave_barrier();
        if constexpr (PROF) {
            __builtin_amdgcn_s_waitcnt(0);
            c_mem += t2 - t1; c_int += t3 - t2; c_leaf += t0 - t3; c_pop += __builtin_amdgcn_s_memtime() - t0;
        }
    } while (!RT_SENTINEL || cur != kSentinel);
    if constexpr (PROF) {
        if (p.trace) {                                          // the longest-lived lane's view of the wave
            unsigned long long v[10] = {c_pop, c_mem, c_int, c_leaf, n_it, n_int, n_leaf, n_g1, n_g2, n_g34};
            unsigned long long best = n_it;
            for (int o = 32; o > 0; o >>= 1) { unsigned long long x = __shfl_xor(best, o); best = x > best ? x : best; }
            const unsigned long long owner = __ballot(n_it == best);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)owner) - 1) {
                unsigned long long* t = p.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kPrimBlock / 64) + (threadIdx.x >> 6)) * 16 + 4;
                for (int k = 0; k < 10; k++) t[k] += v[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The traversal loop of the timed primary kernels, written in gfx950 assembly (RT_ASM_LOOP; the C++ loop above stays the
// reference form, the instrumented kernels' loop and the loop of every case this one does not take).
//
// Why: on this kernel every instruction costs the same, whatever unit executes it -- measured over four rounds of variants
// the frame time follows the TOTAL of instructions issued (vector + scalar + branch) at 0.67 us per million and frame
// (profiles/r05_experiments/instruction_issue_model.md) -- and a third of the compiler's loop is control flow: its structurizer
// wraps every condition in save / restore / skip sequences and keeps wave-uniform decisions in lane masks.  Written by hand an
// iteration needs two exec regions (the lanes at an interior node, the lanes at a triangle) with the push and the accepted
// hit as sub-regions, scalar branches for what is wave-uniform, and no flags.
// Scope: production primary rays (no counters), the LDS-only stack (plain or optimistic, see StackT), an instance without
// exact-uv mode (any pose: round 6 -- the accepted candidate's way back to world space is part of the loop, see the candidate
// block), a wave with a known sign octant (trace_instance's conditions for slab_oct).  Same arithmetic, instruction for instruction, as the C++ loop compiles to: the operations, their order and
// the IEEE division / square-root expansions are taken from the compiler's own output, so every bit of every result is
// the same (the parity tests compare this loop's frames and hit ids with the instrumented kernel's and the oracle's).
//
// Registers: v0..v15 the record, v16..v27 temporaries, s[48:63] the record of a wave-uniform fetch, s[30:45] and s[64:69] masks and
// scalars, s[46:47] the exec mask the loop was entered with; the ray, the hit and the stack state are operands.
// Hazards (gfx940 family): a VALU-written SGPR / VCC needs two wait states before a VALU reads it (none before a SALU read),
// four before v_div_fmas reads VCC; a transcendental result one before a non-transcendental VALU uses it.
#ifndef RT_ASM_LOOP
#define RT_ASM_LOOP 1
#endif
#ifndef RT_ASM_GUARD
#define RT_ASM_GUARD 0
#endif
// RT_ASM_V2 (round 6, second session; 0 = the loop as round 5 wrote it, kept for A/B): the same decisions from fewer vector
// instructions -- masks that only the per-lane path reads are made there (and one of them by the scalar unit), "which child passes"
// is taken from the slab distances themselves instead of from FLT_MAX-patched copies of them, "the nearer child is next" is folded
// into one select, and the stack's top is kept as an LDS ADDRESS (push and pop add a constant to it instead of shifting an index).
// No floating-point operation changes or moves; profiles/r06_experiments/asm_loop_v2.md.
#ifndef RT_ASM_V2
#define RT_ASM_V2 1
#endif

// RT_ASM_BACKFACE: when no lane of the wave holds a triangle that faces its ray (denom < 0, raycast.cu:107-109) the rest of
// the triangle test -- the division, the point on the plane, the barycentrics -- is skipped for the whole wave.
#ifndef RT_ASM_BACKFACE
#define RT_ASM_BACKFACE 1
#endif
#if RT_ASM_BACKFACE
#define RT_ASM_BACKFACE_EXIT \
    "s_and_b64 s[40:41], s[40:41], s[42:43]\n\t" \
    "s_cbranch_scc0 .Lrt_next_triangle%=\n\t"
#define RT_ASM_BACKFACE_AND ""
#else
#define RT_ASM_BACKFACE_EXIT ""
#define RT_ASM_BACKFACE_AND "s_and_b64 s[40:41], s[40:41], s[42:43]\n\t"
#endif

// one interior node: v0..v11 hold box - origin, v12 / v13 the two child entries; exec = the lanes at this node.
// N* / F* = the registers holding the near / far plane of the axis for this octant (slab_oct's operand choice).
#if !RT_ASM_V2
#define RT_ASM_INTERIOR(AXN, AXF, AYN, AYF, AZN, AZF, BXN, BXF, BYN, BYF, BZN, BZF) \
    "v_mul_f32 v16, " AXN ", %[dix]\n\t"  "v_mul_f32 v17, " AYN ", %[diy]\n\t"  "v_mul_f32 v18, " AZN ", %[diz]\n\t" \
    "v_mul_f32 v19, " AXF ", %[dix]\n\t"  "v_mul_f32 v20, " AYF ", %[diy]\n\t"  "v_mul_f32 v21, " AZF ", %[diz]\n\t" \
    "v_max3_f32 v16, v16, v17, v18\n\t"                 /* near of a */ \
    "v_min3_f32 v19, v19, v20, v21\n\t"                 /* far of a */ \
    "v_mul_f32 v22, " BXN ", %[dix]\n\t"  "v_mul_f32 v23, " BYN ", %[diy]\n\t"  "v_mul_f32 v24, " BZN ", %[diz]\n\t" \
    "v_mul_f32 v25, " BXF ", %[dix]\n\t"  "v_mul_f32 v26, " BYF ", %[diy]\n\t"  "v_mul_f32 v27, " BZF ", %[diz]\n\t" \
    "v_max3_f32 v22, v22, v23, v24\n\t"                 /* near of b */ \
    "v_min3_f32 v25, v25, v26, v27\n\t"                 /* far of b */ \
    "v_cmp_ge_f32_e32 vcc, v19, v16\n\t" \
    "v_cmp_lt_f32_e64 s[38:39], 0, v19\n\t" \
    "v_cmp_ge_f32_e64 s[40:41], v25, v22\n\t" \
    "v_cmp_lt_f32_e64 s[42:43], 0, v25\n\t" \
    "v_mov_b32_e32 v17, 0x7f7fffff\n\t" \
    "s_and_b64 vcc, vcc, s[38:39]\n\t" \
    "s_and_b64 s[40:41], s[40:41], s[42:43]\n\t" \
    "v_cndmask_b32_e32 v16, v17, v16, vcc\n\t"          /* da = hit ? near : FLT_MAX */ \
    "v_cndmask_b32_e64 v22, v17, v22, s[40:41]\n\t"     /* db */ \
    "v_cmp_lt_f32_e64 s[38:39], v16, %[hmin]\n\t"       /* pa */ \
    "v_cmp_lt_f32_e64 s[40:41], v22, %[hmin]\n\t"       /* pb */ \
    "v_cmp_lt_f32_e64 s[42:43], v16, v22\n\t"           /* a is the nearer */ \
    "s_and_b64 s[44:45], s[38:39], s[40:41]\n\t"        /* both pass: one is pushed */ \
    "s_or_b64 vcc, s[38:39], s[40:41]\n\t"              /* any passes */ \
    "v_cndmask_b32_e64 v18, v13, v12, s[38:39]\n\t"     /* next = pa ? a : b */ \
    "s_and_saveexec_b64 s[38:39], s[44:45]\n\t" \
    "v_lshl_add_u32 v20, %[sp], %[shift], %[col]\n\t" \
    "v_cmp_gt_i32_e64 s[44:45], %[depth], %[sp]\n\t"    /* the push fits (else it lands in the spare row and the lane stops, see StackT) */ \
    "v_cndmask_b32_e64 v18, v13, v12, s[42:43]\n\t"     /* the nearer child is next */ \
    "ds_write_b32 v20, %[tos]\n\t"                     /* the stack's top lives in a register: the one below it goes to LDS ... */ \
    "v_add_u32_e32 %[sp], 1, %[sp]\n\t" \
    "v_cndmask_b32_e64 %[tos], v12, v13, s[42:43]\n\t"  /* ... and the farther child becomes the top */ \
    "v_cndmask_b32_e64 v18, -2, v18, s[44:45]\n\t" \
    "s_or_b64 exec, exec, s[38:39]\n\t" \
    "v_cndmask_b32_e32 %[cur], -1, v18, vcc\n\t"        /* nothing passes: this lane pops */
#else
// (v2) pa = hit(a) and near(a) < hit.min, pb likewise: the reference's `dist < hit.min` on a distance that is FLT_MAX for a miss
// (BVHTree.hpp:40-54, raycast.cu:72-79) -- FLT_MAX < hit.min is false for every hit.min, so the mask is the same without the select;
// "a is the nearer" is only read where both pass, i.e. where both distances are the slab's own.  s[38:39] = pa, s[40:41] = pb,
// s[68:69] = near(a) < near(b), s[42:43] = the lanes that go on with a, vcc = the lanes that go on at all.
#define RT_ASM_INTERIOR(AXN, AXF, AYN, AYF, AZN, AZF, BXN, BXF, BYN, BYF, BZN, BZF) \
    "v_mul_f32 v16, " AXN ", %[dix]\n\t"  "v_mul_f32 v17, " AYN ", %[diy]\n\t"  "v_mul_f32 v18, " AZN ", %[diz]\n\t" \
    "v_mul_f32 v19, " AXF ", %[dix]\n\t"  "v_mul_f32 v20, " AYF ", %[diy]\n\t"  "v_mul_f32 v21, " AZF ", %[diz]\n\t" \
    "v_max3_f32 v16, v16, v17, v18\n\t"                 /* near of a */ \
    "v_min3_f32 v19, v19, v20, v21\n\t"                 /* far of a */ \
    "v_mul_f32 v22, " BXN ", %[dix]\n\t"  "v_mul_f32 v23, " BYN ", %[diy]\n\t"  "v_mul_f32 v24, " BZN ", %[diz]\n\t" \
    "v_mul_f32 v25, " BXF ", %[dix]\n\t"  "v_mul_f32 v26, " BYF ", %[diy]\n\t"  "v_mul_f32 v27, " BZF ", %[diz]\n\t" \
    "v_max3_f32 v22, v22, v23, v24\n\t"                 /* near of b */ \
    "v_min3_f32 v25, v25, v26, v27\n\t"                 /* far of b */ \
    "v_cmp_ge_f32_e32 vcc, v19, v16\n\t" \
    "v_cmp_lt_f32_e64 s[38:39], 0, v19\n\t" \
    "v_cmp_lt_f32_e64 s[44:45], v16, %[hmin]\n\t" \
    "v_cmp_ge_f32_e64 s[40:41], v25, v22\n\t" \
    "v_cmp_lt_f32_e64 s[42:43], 0, v25\n\t" \
    "v_cmp_lt_f32_e64 s[66:67], v22, %[hmin]\n\t" \
    "v_cmp_lt_f32_e64 s[68:69], v16, v22\n\t"           /* a is the nearer */ \
    "s_and_b64 vcc, vcc, s[38:39]\n\t" \
    "s_and_b64 s[40:41], s[40:41], s[42:43]\n\t" \
    "s_and_b64 s[38:39], vcc, s[44:45]\n\t"             /* pa */ \
    "s_and_b64 s[40:41], s[40:41], s[66:67]\n\t"        /* pb */ \
    "s_orn2_b64 s[42:43], s[68:69], s[40:41]\n\t" \
    "s_and_b64 s[44:45], s[38:39], s[40:41]\n\t"        /* both pass: one is pushed */ \
    "s_and_b64 s[42:43], s[42:43], s[38:39]\n\t"        /* a is next: it passes, and b does not or is the farther */ \
    "s_or_b64 vcc, s[38:39], s[40:41]\n\t"              /* any passes */ \
    "v_cndmask_b32_e64 v18, v13, v12, s[42:43]\n\t" \
    "s_and_saveexec_b64 s[38:39], s[44:45]\n\t" \
    "ds_write_b32 %[sa], %[tos]\n\t"                   /* the stack's top lives in a register: the one below it goes to LDS ... */ \
    "v_cmp_gt_i32_e64 s[44:45], %[lim], %[sa]\n\t"     /* the push fits (else it landed in the spare row and the lane stops, see StackT) */ \
    "v_add_u32_e32 %[sa], %[stride], %[sa]\n\t" \
    "v_cndmask_b32_e64 %[tos], v12, v13, s[68:69]\n\t" /* ... and the farther child becomes the top */ \
    "v_cndmask_b32_e64 v18, -2, v18, s[44:45]\n\t" \
    "s_or_b64 exec, exec, s[38:39]\n\t" \
    "v_cndmask_b32_e32 %[cur], -1, v18, vcc\n\t"        /* nothing passes: this lane pops */
#endif

// The pieces that differ between the plain loop and the VIEW loop (RtScene::ViewPool: interior records whose box words already are
// box - origin, `vdelta` bytes behind the records themselves):
//   UNI_OFFSET    s31 = byte offset of the wave's one entry            VEC_ADDR     v16 = byte offset of the lane's entry
//   UNI_INTERIOR  the interior step of a whole wave at one node         VEC_SUBS     box - origin of the lanes at interior nodes
#define RT_ASM_UNI_OFFSET_PLAIN \
    "s_lshl_b32 s31, s30, 6\n\t"
#define RT_ASM_UNI_OFFSET_VIEW \
    "s_cmp_lt_i32 s30, 0\n\t" \
    "s_cselect_b32 s38, 0, %[vdelta]\n\t"              /* (a triangle record is read where it always is) */ \
    "s_lshl_b32 s31, s30, 6\n\t" \
    "s_add_u32 s31, s31, s38\n\t"
#define RT_ASM_UNI_SUBS \
    "v_sub_f32_e32 v0, s48, %[rox]\n\t"  "v_sub_f32_e32 v1, s49, %[roy]\n\t"  "v_sub_f32_e32 v2, s50, %[roz]\n\t" \
    "v_sub_f32_e32 v3, s51, %[rox]\n\t"  "v_sub_f32_e32 v4, s52, %[roy]\n\t"  "v_sub_f32_e32 v5, s53, %[roz]\n\t" \
    "v_sub_f32_e32 v6, s54, %[rox]\n\t"  "v_sub_f32_e32 v7, s55, %[roy]\n\t"  "v_sub_f32_e32 v8, s56, %[roz]\n\t" \
    "v_sub_f32_e32 v9, s57, %[rox]\n\t"  "v_sub_f32_e32 v10, s58, %[roy]\n\t" "v_sub_f32_e32 v11, s59, %[roz]\n\t"
#define RT_ASM_VEC_ADDR_PLAIN \
    "v_lshlrev_b32_e32 v16, 6, %[cur]\n\t"
#define RT_ASM_VEC_ADDR_VIEW \
    "v_mov_b32_e32 v17, %[vdelta]\n\t" \
    "v_cndmask_b32_e64 v17, 0, v17, s[64:65]\n\t"      /* lanes at an interior node read the frame's view of it */ \
    "v_lshl_add_u32 v16, %[cur], 6, v17\n\t"
#define RT_ASM_VEC_SUBS \
    "s_waitcnt vmcnt(3)\n\t" \
    "v_sub_f32_e32 v0, v0, %[rox]\n\t"  "v_sub_f32_e32 v1, v1, %[roy]\n\t"  "v_sub_f32_e32 v2, v2, %[roz]\n\t" \
    "v_sub_f32_e32 v3, v3, %[rox]\n\t" \
    "s_waitcnt vmcnt(2)\n\t" \
    "v_sub_f32_e32 v4, v4, %[roy]\n\t"  "v_sub_f32_e32 v5, v5, %[roz]\n\t" \
    "v_sub_f32_e32 v6, v6, %[rox]\n\t"  "v_sub_f32_e32 v7, v7, %[roy]\n\t" \
    "s_waitcnt vmcnt(1)\n\t" \
    "v_sub_f32_e32 v8, v8, %[roz]\n\t" \
    "v_sub_f32_e32 v9, v9, %[rox]\n\t"  "v_sub_f32_e32 v10, v10, %[roy]\n\t" "v_sub_f32_e32 v11, v11, %[roz]\n\t"

// POPS (the samples-only extension kernel counts node pops per lane): one more per interior node, one per leaf arrived at
#define RT_ASM_POPS_INTERIOR "v_add_u32_e32 %[pops], 1, %[pops]\n\t"
#define RT_ASM_POPS_LEAF \
    "v_cndmask_b32_e64 v17, 0, 1, vcc\n\t"             /* (vcc = first triangle of this leaf, two instructions old) */ \
    "v_add_u32_e32 %[pops], %[pops], v17\n\t"

// LOC (the bounce kernel's primary ray): the accepted candidate's point on its plane, in mesh space -- the caller turns it into the
// world-space hit location with the reference's own sequence (raycast.cu:98-104), once, for the hit that was kept
// ANYHIT (a shadow ray of the extension kernel, raycast.cu:129-133: the cast returns at its first accepted hit): s[66:67] = the lanes
// that accepted a hit in this triangle step (cleared at the step's start, set in the accepted-candidate block where exec = those
// lanes and v21 = the distance); they take the sentinel after the step's own bookkeeping and leave the loop at the latch
#define RT_ASM_ANYHIT_CLEAR "s_mov_b64 s[66:67], 0\n\t"
#define RT_ASM_ANYHIT_MARK "v_cmp_lt_f32_e64 s[66:67], v21, %[fmax]\n\t"
#define RT_ASM_ANYHIT_LEAVE "v_cndmask_b32_e64 %[cur], %[cur], -2, s[66:67]\n\t"
#define RT_ASM_LOC "v_mov_b32_e32 %[px], v18\n\tv_mov_b32_e32 %[py], v19\n\tv_mov_b32_e32 %[pz], v20\n\t"

// RT_ASM_LAYOUT (round 6): fewer TAKEN branches on the common paths -- a taken branch restarts the wave's instruction fetch.  The
// count lookup of a leaf above 30 triangles moves out of line (the branch over it was taken at almost every triangle), and the
// wave-uniform interior step ends in its own copy of the pop / latch block instead of a branch to the shared one.
#ifndef RT_ASM_LAYOUT
#define RT_ASM_LAYOUT RT_ASM_V2
#endif
#define RT_ASM_LONG_LEAF_BODY \
    "s_and_saveexec_b64 s[40:41], s[38:39]\n\t" \
    "v_and_b32_e32 v17, 0x3ffffff, %[cur]\n\t" \
    "v_lshlrev_b32_e32 v17, 2, v17\n\t" \
    "global_load_dword %[rem], v17, %[lc]\n\t" \
    "s_waitcnt vmcnt(0)\n\t" \
    "s_or_b64 exec, exec, s[40:41]\n\t"
#if RT_ASM_LAYOUT
#define RT_ASM_LONG_LEAF_INLINE "s_cbranch_scc1 .Lrt_long%=\n\t"
#define RT_ASM_LONG_LEAF_OUT_OF_LINE ".Lrt_long%=:\n\t" RT_ASM_LONG_LEAF_BODY "s_branch .Lrt_short%=\n\t"
#define RT_ASM_UNI_TAIL \
    "v_cmp_eq_u32_e32 vcc, -1, %[cur]\n\t" \
    "s_and_saveexec_b64 s[36:37], vcc\n\t" \
    "s_cbranch_execz .Lrt_latch_u%=\n\t" \
    RT_ASM_POP \
    ".Lrt_latch_u%=:\n\t" \
    "s_or_b64 exec, exec, s[36:37]\n\t" \
    "v_cmp_ne_u32_e32 vcc, -2, %[cur]\n\t" \
    "s_and_b64 exec, exec, vcc\n\t" \
    "s_cbranch_execnz .Lrt_top%=\n\t" \
    "s_branch .Lrt_exit%=\n\t"
#else
#define RT_ASM_LONG_LEAF_INLINE "s_cbranch_scc0 .Lrt_short%=\n\t" RT_ASM_LONG_LEAF_BODY
#define RT_ASM_LONG_LEAF_OUT_OF_LINE ""
#define RT_ASM_UNI_TAIL "s_branch .Lrt_pop%=\n\t"
#endif
#if RT_ASM_V2
#define RT_ASM_POP \
    "v_subrev_u32_e32 %[sa], %[stride], %[sa]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"                         /* (the top's reload of an earlier pop: long done) */ \
    "v_mov_b32_e32 %[cur], %[tos]\n\t"                 /* the popped entry is in a register: the lane goes on at once ... */ \
    "ds_read_b32 %[tos], %[sa]\n\t"                    /* ... and the new top arrives while it works on it */
#else
#define RT_ASM_POP \
    "v_add_u32_e32 %[sp], -1, %[sp]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"                         /* (the top's reload of an earlier pop: long done) */ \
    "v_lshl_add_u32 v16, %[sp], %[shift], %[col]\n\t" \
    "v_mov_b32_e32 %[cur], %[tos]\n\t"                 /* the popped entry is in a register: the lane goes on at once ... */ \
    "ds_read_b32 %[tos], v16\n\t"                      /* ... and the new top arrives while it works on it */
#endif
// s[64:65] = the lanes at an interior node, s[34:35] = the lanes at a triangle: read by the per-lane path only.  v2 makes them there
// (the second one is exec without the first: an entry of a live lane is a node or a triangle); the two wait states a vector read of
// s30 needs after v_readfirstlane are an s_nop then (v1: the two compares).  VEC_ADDR_VIEW reads s[64:65] two instructions later.
#if RT_ASM_V2
#define RT_ASM_MASKS_AT_TOP "s_nop 1\n\t"
#define RT_ASM_MASKS_AT_VECTOR \
    "v_cmp_lt_i32_e64 s[64:65], -1, %[cur]\n\t" \
    "s_andn2_b64 s[34:35], exec, s[64:65]\n\t"
#else
#define RT_ASM_MASKS_AT_TOP \
    "v_cmp_lt_i32_e64 s[64:65], -1, %[cur]\n\t"         /* lanes at an interior node */ \
    "v_cmp_gt_i32_e64 s[34:35], 0, %[cur]\n\t"          /* lanes at a triangle */
#define RT_ASM_MASKS_AT_VECTOR ""
#endif
#define RT_ASM_LOOP_TEXT(COUNT_TEXT, POPS_INT, POPS_LEAF, LOC_TEXT, LEAF_TAIL, UNI_OFFSET, UNI_INTERIOR, VEC_ADDR, VEC_SUBS, AXN, AXF, AYN, AYF, AZN, AZF, BXN, BXF, BYN, BYF, BZN, BZF) \
    "s_mov_b64 s[46:47], exec\n\t" \
    ".Lrt_top%=:\n\t" \
    COUNT_TEXT \
    "v_readfirstlane_b32 s30, %[cur]\n\t" \
    RT_ASM_MASKS_AT_TOP \
    "v_cmp_ne_u32_e32 vcc, s30, %[cur]\n\t" \
    "s_cbranch_vccnz .Lrt_vector%=\n\t" \
    /* ---- every lane holds the same entry: the record comes through the scalar cache */ \
    UNI_OFFSET \
    "s_load_dwordx16 s[48:63], %[rec], s31\n\t" \
    "s_cmp_lt_i32 s30, 0\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_cbranch_scc1 .Lrt_one_leaf%=\n\t" \
    "v_mov_b32_e32 v12, s60\n\t" \
    "v_mov_b32_e32 v13, s61\n\t" \
    POPS_INT \
    UNI_INTERIOR \
    RT_ASM_UNI_TAIL \
    ".Lrt_one_leaf%=:\n\t" \
    "v_mov_b32_e32 v0, s48\n\t"  "v_mov_b32_e32 v1, s49\n\t"  "v_mov_b32_e32 v2, s50\n\t"  "v_mov_b32_e32 v3, s51\n\t" \
    "v_mov_b32_e32 v4, s52\n\t"  "v_mov_b32_e32 v5, s53\n\t"  "v_mov_b32_e32 v6, s54\n\t"  "v_mov_b32_e32 v7, s55\n\t" \
    "v_mov_b32_e32 v8, s56\n\t"  "v_mov_b32_e32 v9, s57\n\t"  "v_mov_b32_e32 v10, s58\n\t" "v_mov_b32_e32 v11, s59\n\t" \
    "v_mov_b32_e32 v12, s60\n\t" "v_mov_b32_e32 v13, s61\n\t" "v_mov_b32_e32 v14, s62\n\t" "v_mov_b32_e32 v15, s63\n\t" \
    "s_mov_b64 s[36:37], exec\n\t" \
    "s_branch .Lrt_leaf%=\n\t" \
    /* ---- lanes hold different entries: one 64-byte record per lane, node or triangle, from one array */ \
    ".Lrt_vector%=:\n\t" \
    RT_ASM_MASKS_AT_VECTOR \
    VEC_ADDR \
    "global_load_dwordx4 v[0:3], v16, %[rec]\n\t" \
    "global_load_dwordx4 v[4:7], v16, %[rec] offset:16\n\t" \
    "global_load_dwordx4 v[8:11], v16, %[rec] offset:32\n\t" \
    "global_load_dwordx4 v[12:15], v16, %[rec] offset:48\n\t" \
    "s_and_saveexec_b64 s[36:37], s[64:65]\n\t" \
    "s_cbranch_execz .Lrt_leaf_lanes%=\n\t" \
    POPS_INT \
    VEC_SUBS \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
    RT_ASM_INTERIOR(AXN, AXF, AYN, AYF, AZN, AZF, BXN, BXF, BYN, BYF, BZN, BZF) \
    ".Lrt_leaf_lanes%=:\n\t" \
    "s_or_b64 exec, exec, s[36:37]\n\t" \
    "s_and_saveexec_b64 s[36:37], s[34:35]\n\t" \
    "s_cbranch_execz .Lrt_leaf_done%=\n\t" \
    /* ---- one triangle of a leaf (exec = the lanes at one; s[36:37] = the mask to return to) */ \
    ".Lrt_leaf%=:\n\t" \
    "v_bfe_u32 v16, %[cur], 26, 5\n\t"                  /* the count coded in the entry */ \
    "v_cmp_gt_i32_e32 vcc, 0, %[rem]\n\t"               /* first triangle of this leaf */ \
    "v_cmp_eq_u32_e64 s[38:39], 31, v16\n\t" \
    "s_waitcnt vmcnt(0)\n\t" \
    "v_cndmask_b32_e32 %[rem], %[rem], v16, vcc\n\t" \
    POPS_LEAF \
    "s_and_b64 s[38:39], s[38:39], vcc\n\t"             /* a leaf of more than 30 triangles: its count is in leaf_count */ \
    RT_ASM_LONG_LEAF_INLINE \
    ".Lrt_short%=:\n\t" \
    /* TrianglePrimitive::ray_intersect: denom = rd . n, tt = ((v0 - ro) . n) / denom */ \
    "v_mul_f32_e32 v16, %[rdx], v3\n\t" \
    "v_mul_f32_e32 v17, %[rdy], v4\n\t" \
    "v_add_f32_e32 v16, v16, v17\n\t" \
    "v_mul_f32_e32 v17, %[rdz], v5\n\t" \
    "v_add_f32_e32 v16, v16, v17\n\t"                   /* denom */ \
    "v_sub_f32_e32 v17, v0, %[rox]\n\t" \
    "v_sub_f32_e32 v18, v1, %[roy]\n\t" \
    "v_sub_f32_e32 v19, v2, %[roz]\n\t" \
    "v_mul_f32_e32 v17, v17, v3\n\t" \
    "v_mul_f32_e32 v18, v18, v4\n\t" \
    "v_add_f32_e32 v17, v17, v18\n\t" \
    "v_mul_f32_e32 v18, v19, v5\n\t" \
    "v_add_f32_e32 v17, v17, v18\n\t"                   /* numerator */ \
    "v_div_scale_f32 v18, s[38:39], v16, v16, v17\n\t" \
    "v_rcp_f32_e32 v19, v18\n\t" \
    "v_cmp_nlt_f32_e64 s[40:41], |v16|, %[eps]\n\t"     /* !(|denom| < 1e-6) */ \
    "v_cmp_gt_f32_e64 s[42:43], 0, v16\n\t"             /* denom < 0: the only candidates that can be accepted */ \
    RT_ASM_BACKFACE_EXIT \
    "v_fma_f32 v20, -v18, v19, 1.0\n\t" \
    "v_fmac_f32_e32 v19, v20, v19\n\t" \
    "v_div_scale_f32 v20, vcc, v17, v16, v17\n\t" \
    "v_mul_f32_e32 v21, v20, v19\n\t" \
    "v_fma_f32 v22, -v18, v21, v20\n\t" \
    "v_fmac_f32_e32 v21, v22, v19\n\t" \
    "v_fma_f32 v18, -v18, v21, v20\n\t" \
    "v_div_fmas_f32 v18, v18, v19, v21\n\t" \
    "v_div_fixup_f32 v17, v18, v16, v17\n\t"            /* tt */ \
    RT_ASM_BACKFACE_AND \
    "v_mul_f32_e32 v18, %[rdx], v17\n\t" \
    "v_mul_f32_e32 v19, %[rdy], v17\n\t" \
    "v_mul_f32_e32 v20, %[rdz], v17\n\t" \
    "v_cmp_nlt_f32_e64 s[42:43], v17, 0\n\t"            /* !(tt < 0) */ \
    "v_add_f32_e32 v18, %[rox], v18\n\t"                /* the point on the plane */ \
    "v_add_f32_e32 v19, %[roy], v19\n\t" \
    "v_add_f32_e32 v20, %[roz], v20\n\t" \
    "s_and_b64 s[40:41], s[40:41], s[42:43]\n\t" \
    "v_cmp_neq_f32_e32 vcc, 0x7f7fffff, v18\n\t"        /* raycast.cu:91 */ \
    /* TrianglePrimitive::point_inside */ \
    "v_sub_f32_e32 v21, v18, v0\n\t" \
    "v_sub_f32_e32 v22, v19, v1\n\t" \
    "v_sub_f32_e32 v23, v20, v2\n\t" \
    "s_and_b64 s[40:41], s[40:41], vcc\n\t" \
    "v_mul_f32_e32 v24, v6, v21\n\t" \
    "v_mul_f32_e32 v25, v7, v22\n\t" \
    "v_add_f32_e32 v24, v24, v25\n\t" \
    "v_mul_f32_e32 v25, v8, v23\n\t" \
    "v_add_f32_e32 v24, v24, v25\n\t"                   /* dot02 */ \
    "v_mul_f32_e32 v25, v9, v21\n\t" \
    "v_mul_f32_e32 v26, v10, v22\n\t" \
    "v_add_f32_e32 v25, v25, v26\n\t" \
    "v_mul_f32_e32 v26, v11, v23\n\t" \
    "v_add_f32_e32 v25, v25, v26\n\t"                   /* dot12 */ \
    "v_mul_f32_e32 v26, v14, v24\n\t" \
    "v_mul_f32_e32 v27, v13, v25\n\t" \
    "v_sub_f32_e32 v26, v26, v27\n\t" \
    "v_mul_f32_e32 v26, v26, v15\n\t"                   /* u */ \
    "v_mul_f32_e32 v27, v12, v25\n\t" \
    "v_mul_f32_e32 v21, v13, v24\n\t" \
    "v_sub_f32_e32 v27, v27, v21\n\t" \
    "v_mul_f32_e32 v27, v27, v15\n\t"                   /* v */ \
    "v_cmp_le_f32_e32 vcc, 0, v26\n\t" \
    "v_cmp_le_f32_e64 s[42:43], 0, v27\n\t" \
    "v_add_f32_e32 v21, v26, v27\n\t" \
    "s_and_b64 s[40:41], s[40:41], vcc\n\t" \
    "v_cmp_ge_f32_e32 vcc, 1.0, v21\n\t" \
    "s_and_b64 s[40:41], s[40:41], s[42:43]\n\t" \
    "s_and_b64 s[40:41], s[40:41], vcc\n\t"             /* a candidate inside its triangle */ \
    "s_and_saveexec_b64 s[44:45], s[40:41]\n\t" \
    "s_cbranch_execz .Lrt_no_candidate%=\n\t" \
    /* its world-space location, raycast.cu:98-104: (pt * scale - inv_pose.xyz) rotated by q_inv_pose.  For an instance with scale 1 and the \
       identity quaternion -- translated or not: the common case, and the reference's own scene (kernel.cu:209-240) -- the product and the rotation \
       return their operand (up to the sign of a zero, which the squares below cannot see) and only the subtraction is left; any other \
       instance takes the whole sequence, out of line behind a scalar branch (.Lrt_general, after the loop) */ \
    "s_cmp_lg_u64 %[ip], 0\n\t" \
    "s_cbranch_scc1 .Lrt_general%=\n\t" \
    "v_subrev_f32_e32 v21, %[tx], v18\n\t" \
    "v_subrev_f32_e32 v22, %[ty], v19\n\t" \
    "v_subrev_f32_e32 v23, %[tz], v20\n\t" \
    ".Lrt_have_loc%=:\n\t" \
    /* its distance from the ray's world origin */ \
    "v_subrev_f32_e32 v21, %[orgx], v21\n\t" \
    "v_subrev_f32_e32 v22, %[orgy], v22\n\t" \
    "v_subrev_f32_e32 v23, %[orgz], v23\n\t" \
    "v_mul_f32_e32 v21, v21, v21\n\t" \
    "v_mul_f32_e32 v22, v22, v22\n\t" \
    "v_add_f32_e32 v21, v21, v22\n\t" \
    "v_mul_f32_e32 v22, v23, v23\n\t" \
    "v_add_f32_e32 v21, v21, v22\n\t" \
    "s_mov_b32 s31, 0xf800000\n\t"                      /* sqrtf, as the compiler expands it */ \
    "s_movk_i32 s30, 0x260\n\t" \
    "v_mul_f32_e32 v22, 0x4f800000, v21\n\t" \
    "v_cmp_gt_f32_e32 vcc, s31, v21\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_e32 v21, v21, v22, vcc\n\t" \
    "v_sqrt_f32_e32 v22, v21\n\t" \
    "s_nop 0\n\t" \
    "v_add_u32_e32 v23, -1, v22\n\t" \
    "v_fma_f32 v24, -v23, v22, v21\n\t" \
    "v_cmp_ge_f32_e64 s[38:39], 0, v24\n\t" \
    "v_add_u32_e32 v24, 1, v22\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32_e64 v23, v22, v23, s[38:39]\n\t" \
    "v_fma_f32 v22, -v24, v22, v21\n\t" \
    "v_cmp_lt_f32_e64 s[38:39], 0, v22\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_e64 v22, v23, v24, s[38:39]\n\t" \
    "v_mul_f32_e32 v23, 0x37800000, v22\n\t" \
    "v_cndmask_b32_e32 v22, v22, v23, vcc\n\t" \
    "v_cmp_class_f32_e64 vcc, v21, s30\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_e32 v21, v22, v21, vcc\n\t"          /* distance */ \
    /* raycast.cu:107-109: accepted when nothing was hit yet or this is closer (denom < 0 is part of the candidate mask) */ \
    "v_cmp_eq_f32_e32 vcc, 0x7f7fffff, %[hmin]\n\t" \
    "v_cmp_lt_f32_e64 s[38:39], v21, %[hmin]\n\t" \
    "v_cmp_lt_i32_e64 s[42:43], 0, %[rem]\n\t"          /* (an empty leaf holds no triangle) */ \
    "s_or_b64 vcc, vcc, s[38:39]\n\t" \
    "s_and_b64 vcc, vcc, s[42:43]\n\t" \
    "s_and_b64 exec, exec, vcc\n\t" \
    "v_mov_b32_e32 %[hmin], v21\n\t" \
    "v_and_b32_e32 %[hslot], 0x3ffffff, %[cur]\n\t" \
    "v_mov_b32_e32 %[hinst], %[inst]\n\t" \
    "v_mov_b32_e32 %[hu], v26\n\t" \
    "v_mov_b32_e32 %[hv], v27\n\t" \
    LOC_TEXT \
    ".Lrt_no_candidate%=:\n\t" \
    "s_or_b64 exec, exec, s[44:45]\n\t" \
    ".Lrt_next_triangle%=:\n\t" \
    /* next triangle of the leaf, or a pop */ \
    "v_add_u32_e32 %[rem], -1, %[rem]\n\t" \
    "v_add_u32_e32 v16, 1, %[cur]\n\t" \
    "v_cmp_lt_i32_e32 vcc, 0, %[rem]\n\t" \
    "s_nop 1\n\t" \
    "v_cndmask_b32_e32 %[cur], -1, v16, vcc\n\t" \
    "v_cndmask_b32_e32 %[rem], -1, %[rem], vcc\n\t" \
    LEAF_TAIL \
    ".Lrt_leaf_done%=:\n\t" \
    "s_or_b64 exec, exec, s[36:37]\n\t" \
    /* ---- raycast.cu:60-61: lanes with nothing to go on with pop (the sentinel when nothing is left) */ \
    ".Lrt_pop%=:\n\t" \
    "v_cmp_eq_u32_e32 vcc, -1, %[cur]\n\t" \
    "s_and_saveexec_b64 s[36:37], vcc\n\t" \
    "s_cbranch_execz .Lrt_latch%=\n\t" \
    RT_ASM_POP \
    ".Lrt_latch%=:\n\t" \
    "s_or_b64 exec, exec, s[36:37]\n\t" \
    "v_cmp_ne_u32_e32 vcc, -2, %[cur]\n\t" \
    "s_and_b64 exec, exec, vcc\n\t"                     /* lanes that popped the sentinel are done */ \
    "s_cbranch_execnz .Lrt_top%=\n\t" \
    ".Lrt_exit%=:\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"                         /* (the last pop's reload of the top) */ \
    "s_mov_b64 exec, s[46:47]\n\t" \
    "s_branch .Lrt_end%=\n\t" \
    /* ---- out of line: mesh -> world of an accepted candidate for an instance that is scaled or rotated (raycast.cu:98-104 with \
       apply_quat of transforms.hpp:165-176 written out: the same products and sums in the same association, no contraction).  The \
       record's registers v0..v7 are free here (the triangle test is over), and so are s[48:63]: the instance's inverse pose comes \
       through the scalar cache where it is needed instead of sitting in ten scalar registers around the loop */ \
    ".Lrt_general%=:\n\t" \
    "s_load_dwordx4 s[48:51], %[ip], 0x20\n\t"         /* DevInstance::q_inv_pose */ \
    "s_load_dwordx2 s[52:53], %[ip], 0x58\n\t"         /* DevInstance::scale x y */ \
    "s_load_dword s54, %[ip], 0x60\n\t"                /* DevInstance::scale z */ \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_mul_f32_e32 v0, s52, v18\n\t" \
    "v_mul_f32_e32 v1, s53, v19\n\t" \
    "v_mul_f32_e32 v2, s54, v20\n\t" \
    "v_subrev_f32_e32 v0, %[tx], v0\n\t" \
    "v_subrev_f32_e32 v1, %[ty], v1\n\t" \
    "v_subrev_f32_e32 v2, %[tz], v2\n\t" \
    "v_mul_f32_e64 v3, -v0, s49\n\t"                   /* a = -v.x * q.y - v.y * q.z - v.z * q.w */ \
    "v_mul_f32_e32 v4, s50, v1\n\t" \
    "v_sub_f32_e32 v3, v3, v4\n\t" \
    "v_mul_f32_e32 v4, s51, v2\n\t" \
    "v_sub_f32_e32 v3, v3, v4\n\t" \
    "v_mul_f32_e32 v4, s48, v0\n\t"                    /* b = v.x * q.x + v.y * q.w - v.z * q.z */ \
    "v_mul_f32_e32 v5, s51, v1\n\t" \
    "v_add_f32_e32 v4, v4, v5\n\t" \
    "v_mul_f32_e32 v5, s50, v2\n\t" \
    "v_sub_f32_e32 v4, v4, v5\n\t" \
    "v_mul_f32_e32 v5, s48, v1\n\t"                    /* c = v.y * q.x + v.z * q.y - v.x * q.w */ \
    "v_mul_f32_e32 v6, s49, v2\n\t" \
    "v_add_f32_e32 v5, v5, v6\n\t" \
    "v_mul_f32_e32 v6, s51, v0\n\t" \
    "v_sub_f32_e32 v5, v5, v6\n\t" \
    "v_mul_f32_e32 v6, s48, v2\n\t"                    /* d = v.z * q.x + v.x * q.z - v.y * q.y */ \
    "v_mul_f32_e32 v7, s50, v0\n\t" \
    "v_add_f32_e32 v6, v6, v7\n\t" \
    "v_mul_f32_e32 v7, s49, v1\n\t" \
    "v_sub_f32_e32 v6, v6, v7\n\t" \
    "v_mul_f32_e32 v21, s48, v4\n\t"                   /* x = q.x * b - q.y * a - q.z * d + q.w * c */ \
    "v_mul_f32_e32 v7, s49, v3\n\t" \
    "v_sub_f32_e32 v21, v21, v7\n\t" \
    "v_mul_f32_e32 v7, s50, v6\n\t" \
    "v_sub_f32_e32 v21, v21, v7\n\t" \
    "v_mul_f32_e32 v7, s51, v5\n\t" \
    "v_add_f32_e32 v21, v21, v7\n\t" \
    "v_mul_f32_e32 v22, s48, v5\n\t"                   /* y = q.x * c - q.z * a - q.w * b + q.y * d */ \
    "v_mul_f32_e32 v7, s50, v3\n\t" \
    "v_sub_f32_e32 v22, v22, v7\n\t" \
    "v_mul_f32_e32 v7, s51, v4\n\t" \
    "v_sub_f32_e32 v22, v22, v7\n\t" \
    "v_mul_f32_e32 v7, s49, v6\n\t" \
    "v_add_f32_e32 v22, v22, v7\n\t" \
    "v_mul_f32_e32 v23, s48, v6\n\t"                   /* z = q.x * d - q.w * a - q.y * c + q.z * b */ \
    "v_mul_f32_e32 v7, s51, v3\n\t" \
    "v_sub_f32_e32 v23, v23, v7\n\t" \
    "v_mul_f32_e32 v7, s49, v5\n\t" \
    "v_sub_f32_e32 v23, v23, v7\n\t" \
    "v_mul_f32_e32 v7, s50, v4\n\t" \
    "v_add_f32_e32 v23, v23, v7\n\t" \
    "s_branch .Lrt_have_loc%=\n\t" \
    RT_ASM_LONG_LEAF_OUT_OF_LINE \
    ".Lrt_end%=:\n\t"

// (experiments: -DRT_ASM_PAD_KIND=1|2|3 adds eight scalar / vector / no-op instructions to every iteration, to price an instruction of each kind)
#define RT_ASM_X8(t) t t t t t t t t
#if RT_ASM_PAD_KIND == 1
#define RT_ASM_PAD RT_ASM_X8("s_mov_b32 s31, s30\n\t")
#elif RT_ASM_PAD_KIND == 2
#define RT_ASM_PAD RT_ASM_X8("v_mov_b32 v27, v26\n\t")
#elif RT_ASM_PAD_KIND == 3
#define RT_ASM_PAD RT_ASM_X8("s_nop 0\n\t")
#else
#define RT_ASM_PAD ""
#endif

// (the byte offsets .Lrt_general reads the instance record at)
static_assert(offsetof(DevInstance, q_inv_pose) == 0x20 && offsetof(DevInstance, scale) == 0x58 && sizeof(DevInstance) % 16 == 0, "DevInstance layout");

// ORGV: the ray's world origin differs per lane (secondary rays of the extension kernel): vector operands, and never a VIEW
template <int OCT, bool COUNT, int ROW_SHIFT, bool VIEW, bool POPS, bool LOC, bool ORGV = false, bool ANYHIT = false>  // ROW_SHIFT = log2 of the bytes between two entries of a lane's LDS stack column
__device__ __forceinline__ void trace_loop_asm(const RenderParams& p, int inst_index, const MeshRay& r, V3 org, lds_int* column, int lds_depth,
                                               int32_t& cur, int32_t& sp, Hit& hit, int& wave_iters, uint32_t vdelta, int& pops, V3& point,
                                               V3 back, const DevInstance* general)
{
    // back = DevInstance::inv_pose_xyz (wave-uniform: scalar operands); general = the instance when its mesh -> world transform scales or
    // rotates (the candidate block then reads scale and q_inv_pose through the scalar cache), null when it only translates
    static_assert(!LOC || POPS, "the hit point is kept for the extension kernel, which counts pops");
    static_assert(!(ORGV && VIEW) && !(ANYHIT && (LOC || !POPS || !ORGV)), "secondary rays: no view; a shadow ray keeps no location");
    const float fmax = FLT_MAX;
    int32_t rem = -1;
#if RT_ASM_V2
    // the stack pointer as the LDS address of the next free row of the lane's column, and the address of the first row that is not there
    // -- as a SCALAR: a column starts less than one row pitch behind its wave's first column (4 bytes per lane), so `row < depth` is
    // `address < first column of the wave + depth * pitch` for every lane
    lds_int* sa = column + (sp << (ROW_SHIFT - 2));
    // (scalar arithmetic on the first active lane's column: a per-lane `column - lane` would be one more value to keep across the loop)
    const int32_t limit = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(size_t)column) - 4 * (int32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)) +
                          (lds_depth << ROW_SHIFT);
#define RT_ASM_STACK_OUT [sa] "+v"(sa)
#define RT_ASM_STACK_IN [lim] "s"(limit), [stride] "n"(1 << ROW_SHIFT)
#else
#define RT_ASM_STACK_OUT [sp] "+v"(sp)
#define RT_ASM_STACK_IN [col] "v"(column), [depth] "s"(lds_depth), [shift] "n"(ROW_SHIFT)
#endif
    int32_t tos = kSentinel;                                    // the stack's top entry (row 0 of the LDS column holds a second sentinel, so
                                                                // that the last pop's reload reads a row that exists and leaves sp == 0)
    const float eps = __int_as_float(0x358637be);
#define RT_ASM_OUT_PLAIN
#define RT_ASM_OUT_POPS , [pops] "+v"(pops)           /* (only the variants that count pops hold a register for them) */
#define RT_ASM_OUT_LOC , [pops] "+v"(pops), [px] "+v"(point.x), [py] "+v"(point.y), [pz] "+v"(point.z)
#define RT_ASM_GO(TEXT) RT_ASM_GO2(TEXT, RT_ASM_OUT_PLAIN)
#define RT_ASM_GO2(TEXT, ...)        /* (... = more output operands, with their leading comma, or nothing) */ \
    asm volatile(TEXT \
                 : [cur] "+v"(cur), RT_ASM_STACK_OUT, [rem] "+v"(rem), [tos] "+v"(tos), [hmin] "+v"(hit.min), [hslot] "+v"(hit.slot), [hinst] "+v"(hit.instance), \
                   [hu] "+v"(hit.u), [hv] "+v"(hit.v), [iters] "+s"(wave_iters) __VA_ARGS__ \
                 : [rox] "v"(r.ro.x), [roy] "v"(r.ro.y), [roz] "v"(r.ro.z), [rdx] "v"(r.rd.x), [rdy] "v"(r.rd.y), [rdz] "v"(r.rd.z), \
                   [dix] "v"(r.dinv.x), [diy] "v"(r.dinv.y), [diz] "v"(r.dinv.z), RT_ASM_STACK_IN, \
                   [rec] "s"(p.records), [lc] "s"(p.leaf_count), [orgx] RT_ASM_ORG_C(org.x), [orgy] RT_ASM_ORG_C(org.y), [orgz] RT_ASM_ORG_C(org.z) RT_ASM_XIN, \
                   [inst] "s"(inst_index), [eps] "s"(eps), [vdelta] "s"(vdelta), \
                   [tx] "s"(back.x), [ty] "s"(back.y), [tz] "s"(back.z), [ip] "s"(general) \
                 : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                   "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
                   "s30", "s31", "s64", "s65", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", \
                   "s66", "s67", "s68", "s69", "vcc", "scc", "memory")
#if RT_ASM_GUARD    // (bring-up only: a loop that does not end leaves after a million iterations instead of hanging the GPU)
#define RT_ASM_COUNT "s_add_u32 %[iters], %[iters], 1\n\ts_cmp_gt_u32 %[iters], 0x100000\n\ts_cbranch_scc1 .Lrt_exit%=\n\t"
#define RT_ASM_NOCOUNT RT_ASM_COUNT
#else
#define RT_ASM_COUNT "s_add_u32 %[iters], %[iters], 1\n\t" RT_ASM_PAD
#define RT_ASM_NOCOUNT RT_ASM_PAD
#endif
#define RT_ASM_ARGS(...) __VA_ARGS__
    // VN = the registers of the record's planes as the per-lane fetch leaves them (x: v0 / v3 for box a, v6 / v9 for box b; y: v1 / v4,
    // v7 / v10; z: v2 / v5, v8 / v11 -- min, max), SN = the scalar registers a wave-uniform fetch leaves them in (s48 ..): per axis
    // (near, far) for this octant -- bit k of OCT set = direction component k negative = the max plane is the near one (slab_oct).
#define RT_ASM_VARIANT(CT, PI, PL, LT, TL, OUT, VN, SN) \
    if constexpr (VIEW) RT_ASM_GO2(RT_ASM_LOOP_TEXT(CT, PI, PL, LT, TL, RT_ASM_UNI_OFFSET_VIEW, RT_ASM_INTERIOR(SN), RT_ASM_VEC_ADDR_VIEW, "", VN), OUT); \
    else RT_ASM_GO2(RT_ASM_LOOP_TEXT(CT, PI, PL, LT, TL, RT_ASM_UNI_OFFSET_PLAIN, RT_ASM_UNI_SUBS RT_ASM_INTERIOR(VN), RT_ASM_VEC_ADDR_PLAIN, RT_ASM_VEC_SUBS, VN), OUT)
    // (COUNT -- the tile cost of single-frame primary launches -- and POPS -- the extension kernel's pop plane -- never meet)
#define RT_ASM_CASE(N, VN, SN) \
    if constexpr (OCT == N) { if constexpr (LOC) { RT_ASM_VARIANT(RT_ASM_NOCOUNT, RT_ASM_POPS_INTERIOR, RT_ASM_POPS_LEAF, RT_ASM_LOC, "", RT_ASM_OUT_LOC, RT_ASM_ARGS(VN), RT_ASM_ARGS(SN)); } \
                              RT_ASM_CASE_ANYHIT(RT_ASM_ARGS(VN), RT_ASM_ARGS(SN)) \
                              else if constexpr (POPS) { RT_ASM_VARIANT(RT_ASM_NOCOUNT, RT_ASM_POPS_INTERIOR, RT_ASM_POPS_LEAF, "", "", RT_ASM_OUT_POPS, RT_ASM_ARGS(VN), RT_ASM_ARGS(SN)); } \
                              else if constexpr (COUNT) { RT_ASM_VARIANT(RT_ASM_COUNT, "", "", "", "", RT_ASM_OUT_PLAIN, RT_ASM_ARGS(VN), RT_ASM_ARGS(SN)); } \
                              else { RT_ASM_VARIANT(RT_ASM_NOCOUNT, "", "", "", "", RT_ASM_OUT_PLAIN, RT_ASM_ARGS(VN), RT_ASM_ARGS(SN)); } }
#define RT_ASM_ALL_CASES \
    RT_ASM_CASE(0, RT_ASM_ARGS("v0", "v3", "v1", "v4", "v2", "v5", "v6", "v9", "v7", "v10", "v8", "v11"), RT_ASM_ARGS("s48", "s51", "s49", "s52", "s50", "s53", "s54", "s57", "s55", "s58", "s56", "s59")) \
    RT_ASM_CASE(1, RT_ASM_ARGS("v3", "v0", "v1", "v4", "v2", "v5", "v9", "v6", "v7", "v10", "v8", "v11"), RT_ASM_ARGS("s51", "s48", "s49", "s52", "s50", "s53", "s57", "s54", "s55", "s58", "s56", "s59")) \
    RT_ASM_CASE(2, RT_ASM_ARGS("v0", "v3", "v4", "v1", "v2", "v5", "v6", "v9", "v10", "v7", "v8", "v11"), RT_ASM_ARGS("s48", "s51", "s52", "s49", "s50", "s53", "s54", "s57", "s58", "s55", "s56", "s59")) \
    RT_ASM_CASE(3, RT_ASM_ARGS("v3", "v0", "v4", "v1", "v2", "v5", "v9", "v6", "v10", "v7", "v8", "v11"), RT_ASM_ARGS("s51", "s48", "s52", "s49", "s50", "s53", "s57", "s54", "s58", "s55", "s56", "s59")) \
    RT_ASM_CASE(4, RT_ASM_ARGS("v0", "v3", "v1", "v4", "v5", "v2", "v6", "v9", "v7", "v10", "v11", "v8"), RT_ASM_ARGS("s48", "s51", "s49", "s52", "s53", "s50", "s54", "s57", "s55", "s58", "s59", "s56")) \
    RT_ASM_CASE(5, RT_ASM_ARGS("v3", "v0", "v1", "v4", "v5", "v2", "v9", "v6", "v7", "v10", "v11", "v8"), RT_ASM_ARGS("s51", "s48", "s49", "s52", "s53", "s50", "s57", "s54", "s55", "s58", "s59", "s56")) \
    RT_ASM_CASE(6, RT_ASM_ARGS("v0", "v3", "v4", "v1", "v5", "v2", "v6", "v9", "v10", "v7", "v11", "v8"), RT_ASM_ARGS("s48", "s51", "s52", "s49", "s53", "s50", "s54", "s57", "s58", "s55", "s59", "s56")) \
    RT_ASM_CASE(7, RT_ASM_ARGS("v3", "v0", "v4", "v1", "v5", "v2", "v9", "v6", "v10", "v7", "v11", "v8"), RT_ASM_ARGS("s51", "s48", "s52", "s49", "s53", "s50", "s57", "s54", "s58", "s55", "s59", "s56"))
    // (the same eight loops with the ray's world origin in vector registers and FLT_MAX at hand: secondary rays)
    if constexpr (ORGV) {
#define RT_ASM_ORG_C "v"
#define RT_ASM_XIN , [fmax] "s"(fmax)
#define RT_ASM_CASE_ANYHIT(VN, SN) else if constexpr (ANYHIT) { RT_ASM_VARIANT(RT_ASM_NOCOUNT, RT_ASM_POPS_INTERIOR, RT_ASM_POPS_LEAF RT_ASM_ANYHIT_CLEAR, RT_ASM_ANYHIT_MARK, RT_ASM_ANYHIT_LEAVE, RT_ASM_OUT_POPS, RT_ASM_ARGS(VN), RT_ASM_ARGS(SN)); }
        RT_ASM_ALL_CASES
#undef RT_ASM_ORG_C
#undef RT_ASM_XIN
#undef RT_ASM_CASE_ANYHIT
    } else {
#define RT_ASM_ORG_C "s"
#define RT_ASM_XIN
#define RT_ASM_CASE_ANYHIT(VN, SN)
        RT_ASM_ALL_CASES
#undef RT_ASM_ORG_C
#undef RT_ASM_XIN
#undef RT_ASM_CASE_ANYHIT
    }
#undef RT_ASM_ALL_CASES
#undef RT_ASM_CASE
#undef RT_ASM_VARIANT
#undef RT_ASM_ARGS
#undef RT_ASM_COUNT
#undef RT_ASM_NOCOUNT
#undef RT_ASM_GO
#undef RT_ASM_GO2
#undef RT_ASM_OUT_PLAIN
#undef RT_ASM_OUT_POPS
#undef RT_ASM_OUT_LOC
#undef RT_ASM_STACK_OUT
#undef RT_ASM_STACK_IN
#if RT_ASM_V2
    sp = (int32_t)(sa - column) >> (ROW_SHIFT - 2);
#endif
}

// Octant-specialised loops (RT_OCTANTS=0 at compile time keeps only the generic one).  The rays of a wave -- an 8x8-pixel
// tile, or the samples of a few pixels -- almost always share the sign octant of their mesh-space direction: one ballot per
// instance decides whether the wave runs the loop instantiated for that octant (slab_oct: the min / max pairs of the slab test
// become operand choices) or the generic loop.  A wave qualifies when every active lane has a finite origin and finite,
// non-zero direction inverses of the same signs, and the mesh has no interior record with an unordered or NaN box
// (mesh_flags bit 0, kept by every writer of interior records: upload, device rebuild, refit).
// OCTANTS = false keeps one loop: the bounce / shadow kernels of the extension renderer carry their path state across every
// cast in registers they do not have (spilled to scratch); nine loops per cast site made that worse (c3 24.9 ms against
// 23.7 ms with the generic loop alone, profiles/r04_experiments/octants_in_extension_kernels.log), while the samples-only
// kernel, which carries nothing, gains like the primary kernel (c4 at 16 spp: 8.27 against 8.50 ms).
// VIEW: `view_off` = byte offset of the frame's view records from p.records (see trace_loop).
// UNIFORM_ORG: every lane's ray starts at the same point (primary rays; the hand-written loop takes the origin as scalars).
// STATS (rt_scene_loop_stats: an instrumented copy of the production kernels, never a timed launch): which loop this wave ran for
// this instance goes to p.loop_stats (RT_LOOP_*), one count per wave and instance.
__device__ __forceinline__ void loop_stat(const RenderParams& p, int which, unsigned long long n = 1ull)
{
    const unsigned long long active = __ballot(true);
    if ((int)(__lane_id()) == __ffsll((long long)active) - 1) atomicAdd(&p.loop_stats[which], n);
}

template <bool DEBUG, bool PROF, bool EX = false, bool COUNT = false, class STK = Stack, bool POPS = false, bool OCTANTS = true, bool ANYHIT = false,
          bool VIEW = false, bool UNIFORM_ORG = !EX, bool STATS = false, bool SEC = false>
__device__ __forceinline__ void trace_instance(const RenderParams& p, const DevInstance& in, int inst_index,
                                               V3 org, V3 dir, STK& stack, Hit& hit, Counters<DEBUG>& cnt, int* iters = nullptr,
                                               int* pops = nullptr, uint32_t view_off = 0)
{
    const MeshRay r = to_mesh_space(in, org, dir);
    const uint32_t vdelta = VIEW ? view_off + (uint32_t)p.view_inst_off[inst_index] : 0u;
    int oct = -1;
    if constexpr (RT_OCTANTS && !PROF && OCTANTS) {
        const float inf = __int_as_float(0x7f800000);
        const bool usable = fabsf(r.dinv.x) < inf && fabsf(r.dinv.y) < inf && fabsf(r.dinv.z) < inf &&
                            r.dinv.x != 0.0f && r.dinv.y != 0.0f && r.dinv.z != 0.0f &&
                            fabsf(r.ro.x) < inf && fabsf(r.ro.y) < inf && fabsf(r.ro.z) < inf;
        const int mine = (r.dinv.x < 0.0f ? 1 : 0) | (r.dinv.y < 0.0f ? 2 : 0) | (r.dinv.z < 0.0f ? 4 : 0);
        const int first = __builtin_amdgcn_readfirstlane(mine);
        if ((p.mesh_flags[in.mesh_index] & 1) == 0 && __ballot(!usable || mine != first) == 0ull) oct = first;
    }
    // SEC (round 6: a secondary ray of the extension kernel -- origin per lane, a shadow ray done at its first hit -- on the optimistic
    // stack): the hand-written loop or nothing.  A wave it does not cover (mixed sign octants, an exact-uv mesh) leaves with sp != 0,
    // which is what an outgrown stack looks like: cast_ray_ex then casts the ray again on the general stack and the compiler's loop, so
    // a cast site holds the eight hand-written loops and ONE compiled loop.
    static_assert(!SEC || (RT_ASM_LOOP && RT_SENTINEL && RT_OCTANTS && OCTANTS && POPS && !VIEW && !UNIFORM_ORG && !STK::kSpill && !DEBUG && !PROF && !COUNT), "SEC");
    if constexpr (SEC) { if (!(oct >= 0 && in.exact_uv == 0)) { stack.sp = 1; return; } }
    if constexpr (RT_ASM_LOOP && RT_SENTINEL && RT_OCTANTS && OCTANTS && !DEBUG && !PROF && (UNIFORM_ORG || SEC) && (!EX || POPS) && !(POPS && COUNT) && (!ANYHIT || SEC) && !STK::kSpill) {
        // the hand-written loop (trace_loop_asm) for what it covers; everything else takes the C++ loops below
        if (SEC || (oct >= 0 && in.exact_uv == 0)) {
            if constexpr (STATS) { loop_stat(p, RT_LOOP_ASM); if (in.unit_inv == 0) loop_stat(p, RT_LOOP_ASM_POSED); }
            stack.sp = 0;
            stack.push(kSentinel);
            int32_t cur = in.root_ref, sp = stack.sp;
            int wave_iters = 0;
            static_assert(STK::kStride == 64 || STK::kStride == 256, "the stack column's row pitch as a shift");
            int no_pops = 0;
            // EX (the bounce kernel's primary ray keeps the hit's world location): the loop hands back the accepted candidate's point in
            // mesh space; whether it accepted anything shows in the instance field, which it overwrites
            V3 point = v3(0.0f, 0.0f, 0.0f);
            const int32_t instance_before = hit.instance;
            if constexpr (EX) hit.instance = -7;
            // (wave-uniform values read through the scalar cache: the instance record's address is uniform)
            const V3 back = v3(in.inv_pose_xyz[0], in.inv_pose_xyz[1], in.inv_pose_xyz[2]);
            const uintptr_t ia = in.unit_inv != 0 ? (uintptr_t)0 : (uintptr_t)&in;
            const DevInstance* general = (const DevInstance*)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ia >> 32)) << 32) |
                                                              (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ia));
#define RT_TRACE_ASM(O) trace_loop_asm<O, COUNT, (STK::kStride == 64 ? 8 : 10), VIEW, POPS, EX, SEC, ANYHIT>(p, inst_index, r, org, stack.lds, stack.lds_depth, cur, sp, hit, wave_iters, vdelta, \
                                                                                              POPS ? *pops : no_pops, point, back, general)
            switch (oct) {
            case 0: RT_TRACE_ASM(0); break;
            case 1: RT_TRACE_ASM(1); break;
            case 2: RT_TRACE_ASM(2); break;
            case 3: RT_TRACE_ASM(3); break;
            case 4: RT_TRACE_ASM(4); break;
            case 5: RT_TRACE_ASM(5); break;
            case 6: RT_TRACE_ASM(6); break;
            default: RT_TRACE_ASM(7); break;
            }
#undef RT_TRACE_ASM
            stack.sp = sp;
            if constexpr (COUNT) *iters += wave_iters;          // (the wave's iterations = those of its longest lane, which is what the tile cost is)
            if constexpr (EX) {
                if (hit.instance == -7) hit.instance = instance_before;
                else {                                          // raycast.cu:98-104, as triangle_test<.., EX> does it for every candidate
                    V3 loc = v3(point.x * in.scale[0], point.y * in.scale[1], point.z * in.scale[2]);
                    hit.loc = apply_quat(in.q_inv_pose, v3(loc.x - in.inv_pose_xyz[0], loc.y - in.inv_pose_xyz[1], loc.z - in.inv_pose_xyz[2]));
                }
            }
            return;
        }
    }
    if constexpr (SEC) return;                                  // (never reached: both ways out are above)
    else {
    // (the C++ loop reads view records for the primary kernels' rays only; an extension ray that does not qualify for the hand-written
    // loop reads the records themselves)
#define RT_TRACE_LOOP(O) trace_loop<DEBUG, PROF, EX, COUNT, STK, POPS, O, ANYHIT, (VIEW && !EX)>(p, in, inst_index, r, org, stack, hit, cnt, iters, pops, vdelta)
    if constexpr (STATS) loop_stat(p, STK::kSpill ? RT_LOOP_DEEP : (oct >= 0 ? RT_LOOP_CPP_OCTANT : RT_LOOP_CPP_GENERIC));
    switch (oct) {                                              // (wave-uniform: a scalar branch)
    case 0: RT_TRACE_LOOP(0); break;
    case 1: RT_TRACE_LOOP(1); break;
    case 2: RT_TRACE_LOOP(2); break;
    case 3: RT_TRACE_LOOP(3); break;
    case 4: RT_TRACE_LOOP(4); break;
    case 5: RT_TRACE_LOOP(5); break;
    case 6: RT_TRACE_LOOP(6); break;
    case 7: RT_TRACE_LOOP(7); break;
    default: RT_TRACE_LOOP(-1); break;
    }
#undef RT_TRACE_LOOP
    }
}

__device__ __forceinline__ uint8_t to_u8(float f) { return (uint8_t)(int)f; }

// texture / albedo colour of a hit, raycast.cu:224-245 (ray.color starts at 1,1,1: Ray.hpp:22)
__device__ __forceinline__ V3 base_colour(const RenderParams& p, const Hit& hit)
{
    const DevInstance& in = p.instances[hit.instance];
    const DevMaterial& m = p.materials[in.material_index];
    if (m.texture_width > 0) {                                  // raycast.cu:224-240
        float2 uv = make_float2(hit.u, hit.v);                  // exact-uv meshes: the hit carries uv itself
        if (!in.exact_uv) {
            const float* q = p.tri_uv + (size_t)hit.slot * 6;
            float w = 1.0f - hit.u - hit.v;
            uv.x = (w * q[0] + hit.v * q[2]) + hit.u * q[4];
            uv.y = (w * q[1] + hit.v * q[3]) + hit.u * q[5];
        }
        int tex_x = (int)(uv.x * (float)m.texture_width);
        int tex_y = (int)((1.0 - (double)uv.y) * (double)(float)m.texture_height);
        tex_x = (int)fmaxf((float)(tex_x % m.texture_width), 0.0f);
        tex_y = (int)fmaxf((float)(tex_y % m.texture_height), 0.0f);
        const uint8_t* tc = m.texture + (size_t)tex_y * m.texture_pitch + 3 * (size_t)tex_x;
        return v3(1.0f * ((float)tc[0] * 0.0039215f), 1.0f * ((float)tc[1] * 0.0039215f), 1.0f * ((float)tc[2] * 0.0039215f));
    }
    return v3(1.0f * m.albedo[0], 1.0f * m.albedo[1], 1.0f * m.albedo[2]);     // raycast.cu:241-245
}

// raycast.cu:207-294 -> the pixel's three bytes packed as x | y << 8 | z << 16
__device__ __forceinline__ uint32_t shade(const RenderParams& p, const Hit& hit)
{
    if (hit.min == FLT_MAX) return 255u | (204u << 8) | (153u << 16);      // sky, raycast.cu:208-216
    const V3 c = base_colour(p, hit);
    const float illumination = 1.0f;                            // raycast.cu:282-290
    return (uint32_t)to_u8(illumination * c.x * 255.0f) |       // raycast.cu:292-294
           ((uint32_t)to_u8(illumination * c.y * 255.0f) << 8) |
           ((uint32_t)to_u8(illumination * c.z * 255.0f) << 16);
}

// pixel of thread `tid` in the 16x16 tile at (x0, y0): a wave64 is an 8x8-pixel block (neighbouring rays walk the same nodes:
// L1 hits, little divergence).  (x, ly) = column and LOCAL row; y = frame row (they differ only when rendering stripes).
__device__ __forceinline__ void pixel_of(const RenderParams& p, const FrameParams& f, int tid, int x0, int y0, int& x, int& ly, int& y)
{
    const int wave = tid >> 6, lane = tid & 63;
#if RT_LANE_MORTON
    // Z-order inside the wave's 8x8 block: four consecutive lanes are a 2x2 block of pixels, sixteen a 4x4 block (the idea: the
    // vector memory path coalesces the lanes of a quad that read the same line, and rays of a 2x2 block hold the same entry
    // more often than four rays in a row).  Measured: no difference on any camera (profiles/r04_experiments/sentinel_loop_ab.log);
    // off.  (Which lane renders which pixel changes nothing a pixel computes.)
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), lyy = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
#else
    const int lx = lane & 7, lyy = lane >> 3;
#endif
    x = x0 + (wave & 1) * 8 + lx;
    ly = y0 + (wave >> 1) * 8 + lyy;
    y = ly;                                                     // stripes: local row -> frame row (the identity for one rank)
    if (p.num_ranks != 1) y = ((ly / p.stripe_rows) * p.num_ranks + f.rank) * p.stripe_rows + ly % p.stripe_rows;
}

// whether render_kernel<DEBUG, PROF, ., SPILL> runs the optimistic stack (StackT), and the rows of its LDS block
template <bool DEBUG, bool PROF, bool SPILL>
__host__ __device__ constexpr bool optimistic_stack() { return RT_OPTIMISTIC_STACK && RT_SENTINEL && SPILL && !DEBUG && !PROF; }
template <bool DEBUG, bool PROF, bool SPILL>
__host__ __device__ inline int lds_block_rows(int stack_depth) { return lds_rows(stack_depth) + (optimistic_stack<DEBUG, PROF, SPILL>() ? 1 : 0); }

// One pixel: camera ray -> cast_ray over all instances -> flat shade -> store (raycast.cu:146-297).
template <bool DEBUG, bool PROF, bool COUNT = false, bool SPILL = true, bool VIEW = false, bool STATS = false>
__device__ __forceinline__ void render_pixel(const RenderParams& p, const FrameParams& f, int x0, int y0, lds_int* lds_base, lds_int*& lds_column,
                                             int* iters = nullptr, uint32_t view_off = 0)
{
    int x, ly, y;
    pixel_of(p, f, (int)(lds_column - lds_base), x0, y0, x, ly, y);
    const V3 org = v3(f.origin[0], f.origin[1], f.origin[2]);
    const V3 dir = camera_direction(f, (float)x, (float)y);

    Hit hit;
    hit.min = FLT_MAX; hit.slot = -1; hit.instance = -1; hit.u = 0.0f; hit.v = 0.0f;
    Counters<DEBUG> cnt;
    if constexpr (optimistic_stack<DEBUG, PROF, SPILL>()) {
        // A tree deeper than the LDS part of the stack, a ray that (almost always) is not: the LDS-only loops, and the general
        // stack only for the lanes that turn out to need it -- from the start of the ray, so the hit is the one the general kernel
        // finds (the instrumented kernel keeps the general stack throughout: its counts are per ray, not per attempt).
        StackT<kPrimBlock, false, true> stack;
        stack.lds = lds_column; stack.spill = nullptr; stack.lds_depth = lds_rows(p.stack_depth); stack.sp = 0;
        int outgrown = 0;                                       // a loop left through the spare row ends with sp != 0
        for (int i = 0; i < p.num_instances; i++) {             // raycast.cu:26
            trace_instance<DEBUG, PROF, false, COUNT, StackT<kPrimBlock, false, true>, false, true, false, VIEW, true, STATS>(p, p.instances[i], i, org, dir, stack, hit, cnt,
                                                                                                                         iters, nullptr, view_off);
            outgrown |= stack.sp;
        }
        if constexpr (STATS) { loop_stat(p, RT_LOOP_WAVES); if (__ballot(outgrown != 0) != 0ull) loop_stat(p, RT_LOOP_RETRACED_LANES, (unsigned long long)__popcll(__ballot(outgrown != 0))); }
        if (outgrown != 0) {
            int spill[kMaxStack - kLdsStack];
            StackT<kPrimBlock, true> deep;
            deep.lds = lds_column; deep.spill = spill; deep.lds_depth = lds_rows(p.stack_depth); deep.sp = 0;
            hit.min = FLT_MAX; hit.slot = -1; hit.instance = -1; hit.u = 0.0f; hit.v = 0.0f;
            for (int i = 0; i < p.num_instances; i++)
                trace_instance<DEBUG, PROF, false, COUNT, StackT<kPrimBlock, true>, false, false, false, false, true, STATS>(p, p.instances[i], i, org, dir, deep, hit, cnt, iters);
        }
    } else {
        int spill[SPILL ? kMaxStack - kLdsStack : 1];
        StackT<kPrimBlock, SPILL> stack;
        stack.lds = lds_column; stack.spill = spill; stack.lds_depth = lds_rows(p.stack_depth); stack.sp = 0;
        if constexpr (STATS) loop_stat(p, RT_LOOP_WAVES);
        for (int i = 0; i < p.num_instances; i++)               // raycast.cu:26
            trace_instance<DEBUG, PROF, false, COUNT, StackT<kPrimBlock, SPILL>, false, true, false, VIEW, true, STATS>(p, p.instances[i], i, org, dir, stack, hit, cnt, iters,
                                                                                                               nullptr, view_off);
    }

    // The pixel's coordinates are not kept across the traversal (three registers in a kernel that has none to spare: they
    // were spilled to scratch): they are derived again from the one per-thread value the loop keeps anyway, the address of
    // the lane's LDS stack column.  (The empty asm hides that address's origin from the optimiser, which would otherwise
    // recognise the recomputation and keep the first copies alive.)
    asm volatile("" : "+v"(lds_column));
    pixel_of(p, f, (int)(lds_column - lds_base), x0, y0, x, ly, y);
    const uint32_t px = shade(p, hit);
    uint8_t* out = f.img + (size_t)ly * p.pitch + 3 * (size_t)x;
    out[0] = (uint8_t)px; out[1] = (uint8_t)(px >> 8); out[2] = (uint8_t)(px >> 16);

    // hit-id planes: also available from the production kernel (rt_render_ids), so that the kernel that is timed is the
    // kernel whose hit ids are checked; a wave-uniform branch on a kernel argument, outside the traversal loop
    if (p.hit_instance || p.hit_triangle) {
        const size_t o = (size_t)y * p.width + x;
        if (p.hit_instance) p.hit_instance[o] = hit.instance;
        if (p.hit_triangle) p.hit_triangle[o] = hit.slot >= 0 ? p.tri_id[hit.slot] : -1;
    }
    if constexpr (DEBUG) {
        size_t o = (size_t)y * p.width + x;
        if (p.node_pops) p.node_pops[o] = cnt.pops;
        if (p.aabb_tests) p.aabb_tests[o] = cnt.aabb;
        if (p.tri_tests) p.tri_tests[o] = cnt.tris;
        if (p.inside_hits) p.inside_hits[o] = cnt.inside;
    }
}

// ORDERED (single-frame launches, see launch()): workgroup b renders tile tile_order[b] --
// the tiles that took longest in the previous frame first -- and the last wave of every workgroup to finish records the workgroup's cost for the
// next frame's order (cost = loop iterations of its longest lane).  A frame rendered alone ends in a tail of a few long waves (rays grazing the silhouette) on an
// otherwise idle chip; started first, those waves run beside the bulk of the frame instead of after it
// (measured: -22 % / -12 % / 0 % frame time for the far / mid / near camera).  Which tile a workgroup renders does not
// change what a pixel computes.
template <bool DEBUG, bool PROF, bool ORDERED = false, bool SPILL = true, bool VIEW = false, bool STATS = false>
__global__ __launch_bounds__(kPrimBlock, 8) void render_kernel(const RenderParams p)
{
    extern __shared__ int lds_stack[];                          // [lds_block_rows(stack_depth)][kPrimBlock] (+ 2 ints when ORDERED)

    // Workgroup b renders tile b (row-major).  Consecutive workgroups are dealt round-robin to the 8 XCDs, so every
    // XCD sees tiles from the whole frame: measured faster than giving each XCD one contiguous band (better load
    // balance; the working set is L1/L2 resident either way).
    int tile = (int)blockIdx.x, frame = (int)blockIdx.y;
    int iters = 0;
    if constexpr (ORDERED) {
        // a one-dimensional grid: workgroup b renders frame b % F of the tile with rank b / F, so the costly tiles of ALL
        // frames of a batch come first and the launch ends with cheap ones
        frame = tile % p.num_frames;
        tile = tile / p.num_frames;
        if (p.tile_order) tile = p.tile_order[tile];
    }
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;

    unsigned long long t_start = 0;
    if (p.trace) t_start = wall_clock64();
    // the thread's index lives on as the address of its LDS stack column only (see render_pixel)
    lds_int* column = (lds_int*)lds_stack + threadIdx.x;
    {
        int x, ly, y;
        pixel_of(p, p.frames[frame], (int)threadIdx.x, tx * kPrimTile, ty * kPrimTile, x, ly, y);
        if (x < p.width && ly < p.frames[frame].local_rows)
            render_pixel<DEBUG, PROF, ORDERED, SPILL, VIEW, STATS>(p, p.frames[frame], tx * kPrimTile, ty * kPrimTile, (lds_int*)lds_stack, column, &iters,
                                                            VIEW ? p.view_base + (uint32_t)frame * p.view_frame_stride : 0u);
    }
    asm volatile("" : "+v"(column));
    const int tid = (int)(column - (lds_int*)lds_stack), wave = tid >> 6, lane = tid & 63;
    if constexpr (ORDERED) {
        // cost of the tile = loop iterations of its longest lane: deterministic, unlike a lifetime, which also measures
        // how full the chip was while the workgroup ran
        static_assert(!ORDERED || kPrimBlock == 64, "one wave per workgroup: the wave's longest lane is the tile's");
        for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(iters, o); iters = v > iters ? v : iters; }
        // Round 6: the cost goes to the tile AND ITS EIGHT NEIGHBOURS, as a maximum (nine lanes, one atomic each; the sort clears what
        // it has read).  The costly tiles are the ones whose rays graze a silhouette, and which tiles those are changes with the
        // smallest camera motion: a frame a few millimetres along finds last frame's order one tile off exactly where it matters
        // (single frames along bench.py's 4 mm camera loop 0.134 ms, the same pose repeated 0.127: single_frame_gap.md).  With the
        // neighbourhood's maximum the tiles NEXT to a costly one start early too; several frames per launch (a rank's stripes of a
        // few frames, which differ in pose and, with a rotating owner, in the rows a tile stands for) add up the same way.
        if (p.tile_cost && lane < 9) {
            const int nx = tile % p.tiles_x + lane % 3 - 1, ny = tile / p.tiles_x + lane / 3 - 1;
            if (nx >= 0 && nx < p.tiles_x && ny >= 0 && ny < p.tiles_y) atomicMax(&p.tile_cost[ny * p.tiles_x + nx], iters);
        }
    }
    if (p.trace && lane == 0) {                                 // diagnostic: per-wave lifetime (RT_TRACE_FILE)
        unsigned long long* t = p.trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (kPrimBlock / 64) + wave) * 16;
        t[0] = t_start; t[1] = wall_clock64();
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        t[2] = ((unsigned long long)xcc << 32) | hw; t[3] = (unsigned long long)tile;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Extension kernel (rt_render_ex): samples per pixel, specular bounces, and the sun + shadow pass that the reference
// carries as commented-out code (raycast.cu:249-290).  The reference has no implementation of these, so the
// semantics are defined in DESIGN.md section 7 (and restated by the test oracle); with spp = 1, bounces = 0, lighting = 0 the
// result equals render_kernel's bit for bit.
// ---------------------------------------------------------------------------------------------------------
// LOC: keep the accepted hit's world location (secondary rays start there).  ANYHIT: a shadow ray, cast_ray(..., true, FLT_MAX)
// of raycast.cu:272 -- it returns at its first accepted hit (:129-133), i.e. a lane that has one (then, and only then, its
// hit.min is below FLT_MAX) takes no part in the remaining instances.
// OPTIMISTIC: only the samples-only kernel asks for it -- in the bounce kernel, which carries its path state across the casts in
// registers it does not have, the second loop costs more in spills than the branches it saves: c3 +8 %
// (profiles/r05_experiments/ex_one_wave_workgroups.log).
// VIEW (the samples-only kernel's primary rays: they share the frame's origin): the cast reads view records, see trace_loop.
// PRIMARY: the ray starts at the camera in every lane (trace_instance's UNIFORM_ORG).
// SEC: a secondary ray through the hand-written loop (trace_instance); with OCTANTS and OPTIMISTIC.
template <bool LOC = true, bool OCTANTS = false, bool ANYHIT = false, class STK = Stack, bool OPTIMISTIC = false, bool VIEW = false, bool PRIMARY = false, bool SEC = false>
__device__ __forceinline__ Hit cast_ray_ex(const RenderParams& p, V3 org, V3 dir, STK& stack, int& pops)
{
    static_assert(!VIEW || OPTIMISTIC, "view records are read by the LDS-only loops");
    static_assert(!SEC || (OCTANTS && OPTIMISTIC && !VIEW && !PRIMARY), "SEC");
    Hit hit;
    hit.min = FLT_MAX; hit.slot = -1; hit.instance = -1; hit.u = 0.0f; hit.v = 0.0f;
    hit.loc = v3(0.0f, 0.0f, 0.0f);
    Counters<false> none;
    if constexpr (OPTIMISTIC && RT_OPTIMISTIC_STACK && RT_SENTINEL && STK::kSpill) {
        // the optimistic stack (StackT, render_pixel): the cast on the LDS part alone, and again from its start on the general stack
        // for the lanes whose stack outgrew it (their pop count starts again too)
        StackT<STK::kStride, false, true> fast;
        fast.lds = stack.lds; fast.spill = nullptr; fast.lds_depth = stack.lds_depth; fast.sp = 0;
        const int pops_before = pops;
        int outgrown = 0;
        for (int i = 0; i < p.num_instances; i++) {
            if constexpr (ANYHIT) { if (hit.min < FLT_MAX) continue; }
            trace_instance<false, false, LOC, false, StackT<STK::kStride, false, true>, true, OCTANTS, ANYHIT, VIEW, PRIMARY, false, SEC>(p, p.instances[i], i, org, dir, fast, hit, none,
                                                                                                                                        nullptr, &pops, p.view_base);
            // (a shadow ray that found its hit left the loop with entries on its stack: it is done, not outgrown)
            outgrown |= (ANYHIT && SEC && hit.min < FLT_MAX) ? 0 : fast.sp;
        }
        if (outgrown == 0) return hit;
        pops = pops_before;
        hit.min = FLT_MAX; hit.slot = -1; hit.instance = -1; hit.u = 0.0f; hit.v = 0.0f;
        hit.loc = v3(0.0f, 0.0f, 0.0f);
    }
    for (int i = 0; i < p.num_instances; i++) {
        if constexpr (ANYHIT) { if (hit.min < FLT_MAX) continue; }
        trace_instance<false, false, LOC, false, STK, true, OCTANTS, ANYHIT>(p, p.instances[i], i, org, dir, stack, hit, none, nullptr, &pops);
    }
    return hit;
}

// world normal of the accepted hit, raycast.cu:115-122
__device__ __forceinline__ V3 hit_normal(const RenderParams& p, const Hit& hit)
{
    const DevInstance& in = p.instances[hit.instance];
    const float4* t = p.records + (size_t)hit.slot * 4;
    float4 t0 = t[0], t1 = t[1];
    V3 n = apply_quat(in.q_inv_rot, v3(t0.w, t1.x, t1.y));
    n.x *= in.scale[0]; n.y *= in.scale[1]; n.z *= in.scale[2];
    return normalize(n);
}

// One workgroup = one 16x16 tile of ONE sample index (blockIdx.y): every (pixel, sample) pair has its own random
// stream, so the samples of a pixel are independent work items and a frame of spp samples is dispatched like a batch
// of spp frames.  (A pixel's samples used to run as a loop inside its lane: the waves on the silhouette then lived
// 64 x as long as their neighbours and the kernel spent most of its time waiting for a handful of them.)
// The sample's radiance and node pops go to ex_samples[sample][local pixel]; resolve_ex_kernel sums them in order.
// (8 waves per SIMD at 32 spilled registers beats 7 / 6 / 5 waves with 18 / 4 / 0 spills: +2 % / +8 % / +21 % time.)
// SIMPLE = no bounces and no lighting (samples per pixel only, BASELINE configs[3]): the path is one primary ray, so
// nothing but the hit has to survive the cast -- no path state spilled around the traversal loop, no hit location.
// PX = the samples of a pixel share a wave (the default for 4 and more samples per pixel): a wave is px_pw x px_ph pixels
// times px_n sample slots -- 2 x 2 pixels x 16 samples, or 1 pixel x 64 samples -- instead of 8 x 8 pixels of one sample
// index.  The jittered rays of one pixel visit almost the same nodes in almost the same order, so the lanes of a wave stay
// together through the loop (most iterations find every lane holding the same entry and take the scalar fetch) and finish
// together; and because a pixel's samples now sit in one wave, they are added up -- in sample order, by one lane per channel,
// through the wave's own LDS columns -- right here: no sample planes in HBM, no resolve pass.  Frames of more than 64 samples
// take several launches, the running sums wait in ex_acc in between.  What a (pixel, sample) pair computes does not depend on
// the lane it runs in, so the frame is the same bit for bit (tests/test_gpu_parity.py compares the two mappings).
// (8 waves per SIMD stays the best residency with this mapping too: 7 / 6 / 5 waves +1.6 % / +6 % / +17 % on c3, +4 % on c4)
// Which (pixel, sample) pair thread `tid` of workgroup (tile, row) works on.  A function of its own so that the kernel can call
// it again after the casts instead of keeping the results (see render_pixel: registers the traversal loop has no room for).
template <bool PX>
__device__ __forceinline__ void ex_work_item(const RenderParams& p, int tile, int row, int tid, int& x, int& ly, int& y, int& s, bool& valid)
{
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int wave = tid >> 6, lane = tid & 63;
    if constexpr (PX) {
        const int pix = lane / p.px_n, sl = lane % p.px_n;
        x = tx * 2 * p.px_pw + (wave & 1) * p.px_pw + pix % p.px_pw;
        ly = ty * 2 * p.px_ph + (wave >> 1) * p.px_ph + pix / p.px_pw;
        s = p.sample_base + sl;
        valid = x < p.width && ly < p.local_rows && sl < p.px_count;
    } else {
        x = tx * kTile + (wave & 1) * 8 + (lane & 7);
        ly = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
        s = p.sample_base + row;
        valid = x < p.width && ly < p.local_rows;
    }
    y = ly;                                                     // stripes: local -> frame row (the identity for one rank)
    if (p.num_ranks != 1) y = ((ly / p.stripe_rows) * p.num_ranks + p.rank) * p.stripe_rows + ly % p.stripe_rows;
}

// The random stream of (pixel, sample) after `drawn` values: the reference's per-pixel seed (raycast.cu:190: int idx * 1000)
// plus the sample index.  A path does not carry the six words of its stream across its casts; it carries how many values
// it has drawn and steps a fresh copy forward when it next needs one (a few dozen integer operations against the hundreds
// of a cast; the sequence of values is the same).
__device__ __forceinline__ Xorwow ex_stream(const RenderParams& p, int x, int y, int s, int drawn)
{
    Xorwow rng;
    xorwow_init(rng, (unsigned long long)((long long)(int32_t)((uint32_t)(y * p.width + x) * 1000u) + (long long)s));
    for (int k = 0; k < drawn; k++) (void)xorwow_next(rng);
    return rng;
}

// One wave per workgroup (round 5, like the primary kernels: a wave's slot is refilled when the wave ends, not when the last of
// four does): workgroups 4t .. 4t + 3 are the four waves of tile t, and `quarter` below is what the wave index within a 256-thread
// workgroup used to be.
constexpr int kExBlock = 64;
typedef StackT<kExBlock> ExStack;
// PHASE (round 6; bounces / lighting, PX mapping): 0 = the whole path in one kernel, as above -- the default.  1 + 2 (RT_EX_SPLIT=1, opt-in)
// = the same path in TWO launches with the same grid: phase 1 casts the camera ray -- the hand-written loop, the frame's view records,
// no path state to carry, like the samples-only kernel -- and stores what survives the cast per lane, (slot, instance, u, v) and
// (location, pops): 32 B in two planes, the lanes of a wave side by side; phase 2 picks the record up and runs the rest of the path
// (shade, shadow ray, bounces) with the compiler's loops only.  The idea: the camera ray is 38 % of c3's node pops,
// and inside the one-kernel form it takes the hand-written loop at the price of 45 spilled registers around EVERY cast of the path (25
// without it).  Measured: SLOWER -- c3 19.5 -> 20.6 ms, c3 from the far camera 7.66 -> 9.14, c5 292 -> 310: the records of a c3 frame
// are 4.2 GB written and 4.2 GB read, 1.0-1.5 ms at the rate HBM gives them, and that is more than the spills cost
// (profiles/r06_experiments/ex_split_two_launches.md, with both forms' kernel times and counters).  The casts, their order and every
// operation on their results are the same, so the frame is the same bit for bit (tests: both forms against each other and the oracle).
// The launch is cut into chunks of workgroups when the records of all of it would not fit the scratch budget (wg_base).
template <bool SIMPLE, bool PX = false, bool VIEW = false, int PHASE = 0>
__global__ __launch_bounds__(kExBlock, (SIMPLE ? 8 : RT_EX_BOUNCE_WAVES)) void render_ex_kernel(const RenderParams p)
{
    static_assert(PHASE == 0 || (PX && !SIMPLE), "the two-launch form is the bounce kernel's, in the pixel-wave mapping");
    static_assert(PHASE != 2 || !VIEW, "phase 2 casts no camera ray");
    extern __shared__ int lds_stack[];
    const FrameParams& f = p.frames[0];
    const unsigned bx = blockIdx.x + (PHASE != 0 ? (unsigned)p.wg_base : 0u);
    const int tile = (int)(bx >> 2), quarter = (int)(bx & 3);
    int x, ly, y, s;
    bool valid;
    ex_work_item<PX>(p, tile, (int)blockIdx.y, quarter * 64 + (int)threadIdx.x, x, ly, y, s, valid);
    if constexpr (!PX) { if (!valid) return; }
    unsigned long long t_start = 0;
    if (p.trace) t_start = wall_clock64();

    int spill[kMaxStack - kLdsStack];
    ExStack stack;
    // (the thread's index lives on as the address of its LDS stack column only)
    stack.lds = (lds_int*)lds_stack + threadIdx.x; stack.spill = spill; stack.lds_depth = lds_rows(p.stack_depth); stack.sp = 0;
    auto where = [&](int& x_, int& ly_, int& y_, int& s_, bool& valid_) {
        asm volatile("" : "+v"(stack.lds));                     // (hides the address's origin: the optimiser would keep the first results alive)
        ex_work_item<PX>(p, tile, (int)blockIdx.y, quarter * 64 + (int)(stack.lds - (lds_int*)lds_stack), x_, ly_, y_, s_, valid_);
    };
    int pops = 0;
    V3 sample = v3(0.0f, 0.0f, 0.0f);
    if (valid) {

    const V3 sun = normalize(v3(-0.2f, 0.0f, 1.0f));                                                    // raycast.cu:249-250
    float px = (float)x, py = (float)y;
    int drawn = 0;                                              // values taken from the (pixel, sample) stream so far
    if (s > 0) {
        Xorwow rng = ex_stream(p, x, y, s, 0);
        px = px + (xorwow_uniform(rng) - 0.5f); py = py + (xorwow_uniform(rng) - 0.5f);
        drawn = 2;
    }
    V3 org = v3(f.origin[0], f.origin[1], f.origin[2]);
    V3 dir = camera_direction(f, px, py);
    V3 weight = v3(1.0f, 1.0f, 1.0f);
    if constexpr (PHASE == 1) {                                 // the camera ray alone: what survives the cast goes to the lane's record
        const Hit hit = cast_ray_ex<true, true, false, ExStack, true, VIEW, true>(p, org, dir, stack, pops);
        asm volatile("" : "+v"(stack.lds));
        const size_t li = (size_t)blockIdx.x * kExBlock + (size_t)(stack.lds - (lds_int*)lds_stack);
        p.ex_rec[li] = make_float4(__int_as_float(hit.slot), __int_as_float(hit.instance), hit.u, hit.v);
        p.ex_rec[(size_t)p.ex_rec_lanes + li] = make_float4(hit.loc.x, hit.loc.y, hit.loc.z, __int_as_float(pops));
    } else
    if constexpr (SIMPLE) {                                     // the loop below for bounces = 0, lighting = 0, written out
        const Hit hit = cast_ray_ex<false, true, false, ExStack, true, VIEW, true>(p, org, dir, stack, pops);
        if (hit.min == FLT_MAX) sample = sample + weight * v3(1.0f, 0.8f, 0.6f);
        else {
            const V3 base = base_colour(p, hit);
            float illum = 1.0f;
            illum = fminf(1.0f, illum);
            illum = fmaxf(0.4f, illum);
            const V3 local = v3(illum * base.x, illum * base.y, illum * base.z);
            sample = sample + weight * (local * (1.0f - 0.0f));
        }
    } else
    {
    // one depth of the path: cast, shade, reflect; false = the path has ended
    // (`primary` = std::true_type for the camera ray.  In the VIEW form of this kernel -- launched when the frame qualifies for view
    // records, i.e. when the tree is small beside the frame -- it takes the hand-written loop and the frame's view: c3 / c5 -4.5 %.
    // The other form keeps the compiler's loop for it: without a view the camera ray's gain is smaller than what 18 more spilled
    // registers cost the bounce casts -- c6 with mirror walls, whose 4 M-node tree gets no view, lost 9 % with it.)
    // (after_cast: everything a depth does with the hit of its cast -- phase 2 enters here with the hit phase 1 stored)
    auto after_cast = [&](Hit hit, const int depth) __attribute__((always_inline)) -> bool {
        if (hit.min == FLT_MAX) { sample = sample + weight * v3(1.0f, 0.8f, 0.6f); return false; }
        float illum = 1.0f;
#if RT_EX_RECOMPUTE
        // What is shaded is a function of the accepted hit -- (slot, instance, u, v) -- and the scene: base colour, normal and
        // cosine are NOT carried across the shadow cast (seven registers of a kernel that spills), they are derived again from
        // those four words after it: the same operations on the same inputs, so the same bits.  (The empty asm makes the slot
        // opaque: the optimiser would otherwise recognise the second derivation and keep the first results alive.)
        if (p.lighting) {                                   // raycast.cu:249-287 with the commented lines active
            const float cos_illum = dot(hit_normal(p, hit), sun);
            illum = (float)(0.4 * (double)cos_illum);
            if (cos_illum > 0) {
                // (only hit-or-miss survives a shadow cast: no hit location to keep.  Octant loops at either cast of this
                // kernel: within +-0.6 %, profiles/r04_experiments/octants_in_extension_kernels.log)
                const Hit sh = cast_ray_ex<false, false, true, ExStack>(p, hit.loc + sun * (float)1e-4, sun, stack, pops);
                asm volatile("" : "+v"(hit.slot));
                if (sh.min == FLT_MAX) illum = (float)(1.0 * (double)dot(hit_normal(p, hit), sun));
            }
        }
        const V3 base = base_colour(p, hit);
        const V3 n = hit_normal(p, hit);
#else
        const V3 base = base_colour(p, hit);
        const V3 n = hit_normal(p, hit);
        if (p.lighting) {                                   // raycast.cu:249-287 with the commented lines active
            const float cos_illum = dot(n, sun);
            illum = (float)(0.4 * (double)cos_illum);
            if (dot(n, sun) > 0) {
                const Hit sh = cast_ray_ex<false, (RT_EX_SECONDARY_ASM & 1) != 0, true, ExStack, (RT_EX_SECONDARY_ASM & 1) != 0, false, false, (RT_EX_SECONDARY_ASM & 1) != 0>(
                    p, hit.loc + sun * (float)1e-4, sun, stack, pops);
                if (sh.min == FLT_MAX) illum = (float)(1.0 * (double)cos_illum);
            }
        }
#endif
        illum = fminf(1.0f, illum);                         // raycast.cu:289-290
        illum = fmaxf(0.4f, illum);
        const V3 local = v3(illum * base.x, illum * base.y, illum * base.z);
        const DevMaterial& mat = p.materials[p.instances[hit.instance].material_index];
        const float m = depth < p.bounces ? mat.metallic : 0.0f;
        sample = sample + weight * (local * (1.0f - m));
        if (!(m > 0.0f)) return false;
        weight = weight * (base * m);
        const float k = 2.0f * dot(dir, n);
        V3 r = dir - n * k;
        if (mat.roughness > 0.0f) {
            int x_, ly_, y_, s_; bool v_;
            where(x_, ly_, y_, s_, v_);
            Xorwow rng = ex_stream(p, x_, y_, s_, drawn);
            drawn += 3;
            float rx = 2.0f * xorwow_uniform(rng) - 1.0f, ry = 2.0f * xorwow_uniform(rng) - 1.0f, rz = 2.0f * xorwow_uniform(rng) - 1.0f;
            r = r + v3(rx, ry, rz) * mat.roughness;
        }
        r = normalize(r);
        org = hit.loc + r * (float)1e-4;
        dir = r;
        return true;
    };
    auto step = [&](auto primary, const int depth) __attribute__((always_inline)) -> bool {
        constexpr bool kPrimary = RT_EX_PRIMARY_ASM && VIEW && decltype(primary)::value;
        constexpr bool kSec = (RT_EX_SECONDARY_ASM & 2) != 0 && !decltype(primary)::value;       // (a bounce ray)
        Hit hit = cast_ray_ex<true, kPrimary || kSec, false, ExStack, kPrimary || kSec, kPrimary, kPrimary, kSec>(p, org, dir, stack, pops);
        return after_cast(hit, depth);
    };
    if constexpr (PHASE == 2) {
        // the camera ray was cast by phase 1: its record (the same lane of the same workgroup of the same grid)
        const size_t li = (size_t)blockIdx.x * kExBlock + (size_t)threadIdx.x;
        const float4 ra = p.ex_rec[li], rb = p.ex_rec[(size_t)p.ex_rec_lanes + li];
        Hit hit;
        hit.slot = __float_as_int(ra.x); hit.instance = __float_as_int(ra.y); hit.u = ra.z; hit.v = ra.w;
        hit.loc = v3(rb.x, rb.y, rb.z);
        hit.min = hit.instance < 0 ? FLT_MAX : 0.0f;            // (only "was anything hit" is asked of it from here on)
        pops = __float_as_int(rb.w);
        if (after_cast(hit, 0))
            for (int depth = 1; depth <= p.bounces; depth++) if (!step(std::false_type{}, depth)) break;
    } else {
#if RT_EX_PEEL
    // the primary ray's depth written out: weight = 1 and sample = 0 are constants across its cast, not registers to keep
    if (step(std::true_type{}, 0))
        for (int depth = 1; depth <= p.bounces; depth++) if (!step(std::false_type{}, depth)) break;
#else
    for (int depth = 0; depth <= p.bounces; depth++) if (!step(std::false_type{}, depth)) break;
#endif
    }
    }
    }
    if constexpr (PHASE == 1) return;                           // (nothing to add up yet: phase 2 finishes the path)
    where(x, ly, y, s, valid);
    const int lane = (int)(stack.lds - (lds_int*)lds_stack);
    if constexpr (PX) {
        // the wave's samples -> its LDS columns (the traversal stacks are idle now), four rows of 64 words: r g b pops
        lds_int* mine = (lds_int*)lds_stack;
        mine[0 * kExBlock + lane] = __float_as_int(sample.x);
        mine[1 * kExBlock + lane] = __float_as_int(sample.y);
        mine[2 * kExBlock + lane] = __float_as_int(sample.z);
        mine[3 * kExBlock + lane] = pops;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the first four lanes of a pixel add up one channel each, in sample order (what resolve_ex_kernel does per pixel)
        const int c = lane % p.px_n;
        if (c < 4 && x < p.width && ly < p.local_rows) {
            const lds_int* col = mine + c * kExBlock + (lane - c);
            const size_t pixel = (size_t)ly * p.width + x;
            if (c < 3) {
                float acc = 0.0f;
                if (!p.px_first) acc = ((const float*)&p.ex_acc[pixel])[c];
                for (int k = 0; k < p.px_count; k++) acc = acc + __int_as_float(col[k]);
                if (!p.px_last) ((float*)&p.ex_acc[pixel])[c] = acc;
                else (f.img + (size_t)ly * p.pitch + 3 * (size_t)x)[c] = to_u8(acc / (float)p.spp * 255.0f);
            } else {
                int acc = 0;
                if (!p.px_first) acc = __float_as_int(p.ex_acc[pixel].w);
                for (int k = 0; k < p.px_count; k++) acc += col[k];
                if (!p.px_last) ((int*)&p.ex_acc[pixel])[3] = acc;
                else if (p.total_pops) p.total_pops[(size_t)y * p.width + x] = acc;
            }
        }
    } else
    p.ex_samples[(size_t)blockIdx.y * ((size_t)p.local_rows * p.width) + (size_t)ly * p.width + x] =
        make_float4(sample.x, sample.y, sample.z, __int_as_float(pops));
    if (p.trace) {                                              // diagnostic: per-wave lifetime (RT_TRACE_FILE)
        const unsigned long long active = __ballot(true);
        if (lane == __ffsll((long long)active) - 1) {
            unsigned long long* t = p.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
            t[0] = t_start; t[1] = wall_clock64(); t[3] = (unsigned long long)tile;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Wavefront form of the extension renderer (opt-in: RT_EX_WAVEFRONT=1, bounces > 0 or lighting): ONE CAST PER LAUNCH, live paths
// compacted in between (the hook the reference leaves for this is its commented-out shadow / light pass, raycast.cu:262-287;
// BASELINE north_star: "wavefront ballot/popc used to compact active rays after the hit test").
//
// render_ex_kernel<false> keeps a path in its lane from the primary ray to its last bounce: the lanes of a wave whose
// paths have ended (sky), or that need no shadow ray, wait for the others at every cast -- 36 of 64 lanes active on the
// blob with sun + 8 bounces.  Here a path is a queue item between two casts:
//
//   depth 0:  ex_wave_kernel<kExGen>      camera ray -> cast -> shade  -+-> sample plane (path ended)
//   depth d:  ex_wave_kernel<kExShadow>   S(d) item  -> shadow cast -> finish the shade   +-> S(d): needs a shadow ray
//             ex_wave_kernel<kExBounce>   A(d) item  -> cast -> shade                   -+-> A(d + 1): next bounce
//
// Queues without atomics and without losing ray coherence: a wave pushes what its lanes produce, compacted with a
// ballot / popc prefix, into the 64-slot SEGMENT that belongs to it, and stores the segment's item count.  exq_k
// consecutive segments form a GROUP; the primary launch orders its waves so that a group is one 8x8-pixel quad (or two
// neighbouring ones) times many samples.  A queue launch gives every group a workgroup: each wave scans the group's
// segment counts (one count per lane, a wave prefix sum), then takes every fourth "pass" of 64 consecutive items of
// the group's concatenated segments -- full waves of rays from the same few pixels, as coherent as primary rays --
// and pushes to the segment (group, pass).  A group's paths never leave the group, so it never holds more than its
// exq_k x 64 primary paths: pass < exq_k, slots are fixed by construction, nothing is counted with atomics and the
// layout of every queue is deterministic.  (A first version with sharded atomic counters mixed rays from unrelated tiles in
// one wave: full waves, but each piece traversed alone -- 55 ms against the 33 ms of the per-lane kernel.)
// A(d + 1) is pushed by two launches (the cast launch of depth d directly, its shadow launch for the shaded rest): two
// segment arrays, concatenated by the reader; the three A arrays rotate (launch_ex).
// An item is the path state between casts as float4 planes (SoA: a wave's store of one plane is 1 KB contiguous).  The
// arithmetic of a path is render_ex_kernel's, operation for operation -- per-(pixel, sample) random streams and the
// index-ordered resolve make the result independent of which lane of which launch ran a cast -- so both forms are
// bit-identical (tests/test_gpu_parity.py runs them against each other and the oracle).  The state a lane would have to
// keep across the traversal loop (32 spilled VGPRs in render_ex_kernel) is re-read from the item after the cast instead.
//
// Item planes (float4 each; `rng` = the six words of the path's XORWOW stream):
//   A: 0 = origin, id      1 = direction, pops      2 = weight, rng.d     3 = sample, rng.v0     4 = rng.v1..v4
//   S: 0 = hit location, id   1 = incoming direction, pops   2..4 as A   5 = base colour, material index   6 = normal
enum { kExGen = 0, kExBounce = 1, kExShadow = 2 };

__device__ __forceinline__ float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ float4 f4(V3 v, int w) { return make_float4(v.x, v.y, v.z, __int_as_float(w)); }
__device__ __forceinline__ float4 f4(V3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }

template <int MODE>
__global__ __launch_bounds__(kBlock, 8) void ex_wave_kernel(const RenderParams p)
{
    extern __shared__ int lds_stack[];
    const FrameParams& f = p.frames[0];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t cap = (size_t)p.exq_nseg * 64;                 // slots (float4) per plane
    const V3 sun = normalize(v3(-0.2f, 0.0f, 1.0f));                                                    // raycast.cu:249-250
    const V3 sky = v3(1.0f, 0.8f, 0.6f);
    const int K = p.exq_k;

    int spill[kMaxStack - kLdsStack];
    Stack stack;
    stack.lds = (lds_int*)lds_stack + tid; stack.spill = spill; stack.lds_depth = lds_rows(p.stack_depth); stack.sp = 0;

    // ---- what this wave reads: GEN = its pixel quad and sample; queue launches = every fourth pass of the group's items ----
    int pass = MODE == kExGen ? 0 : wave, npass = 1, total = 0, incl = 0, count = 0;
    const int group = (int)blockIdx.x;
    const float4* qin = nullptr;
    if constexpr (MODE != kExGen) {
        // one segment count per lane: lanes 0..K-1 the group's segments of the first array, lanes 32..32+K-1 of the second
        const int j = lane & 31, seg = group * K + j;
        if (j < K && seg < p.exq_nseg) {
            if constexpr (MODE == kExShadow) { if (lane < 32) count = p.exq_cnt_s[seg]; }
            else if (lane < 32) count = p.exq_cnt_a[p.exq_in1][seg];
            else if (p.exq_in2 >= 0) count = p.exq_cnt_a[p.exq_in2][seg];
        }
        incl = count;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        total = __builtin_amdgcn_readlane(incl, 63);
        npass = (total + 63) >> 6;
        if (npass > K) npass = K;                               // (cannot happen: a group holds at most K x 64 paths)
        // segments this group will not push to in this launch hold nothing
        if (wave == 0 && lane >= npass && lane < K && group * K + lane < p.exq_nseg) {
            if (MODE == kExBounce && p.lighting) p.exq_cnt_s[group * K + lane] = 0;
            if (p.depth < p.bounces) p.exq_cnt_a[p.exq_out][group * K + lane] = 0;
        }
    }

    for (; pass < npass; pass += kBlock / 64) {
        bool valid;
        size_t in_slot = 0;
        int id = 0, pops = 0, out_seg;
        V3 org = v3(0.0f, 0.0f, 0.0f), dir = v3(0.0f, 0.0f, 0.0f);
        int x = 0, y = 0, s = 0;
        if constexpr (MODE == kExGen) {
            // workgroup = one 8x8-pixel quad x 4 sample indices (wave = sample), or 2 quads x 2 / 4 quads x 1 for chunks of
            // fewer samples; quads of a 16x16 tile are consecutive, and so are the sample indices of a quad: consecutive
            // waves -- the groups of the queue launches -- trace nearly the same rays
            const int spw = p.gen_spw, qpw = 4 / spw;           // samples, quads per workgroup
            const int per_qg = (p.gen_samples + spw - 1) / spw; // workgroups per set of qpw quads
            const int quad = ((int)blockIdx.x / per_qg) * qpw + wave / spw, sl = ((int)blockIdx.x % per_qg) * spw + wave % spw;
            const int tile = quad >> 2, sub = quad & 3;
            const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
            x = tx * kTile + (sub & 1) * 8 + (lane & 7);
            const int ly = ty * kTile + (sub >> 1) * 8 + (lane >> 3);
            valid = x < p.width && ly < p.local_rows && sl < p.gen_samples;
            y = ((ly / p.stripe_rows) * p.num_ranks + p.rank) * p.stripe_rows + ly % p.stripe_rows;    // stripes: local -> frame row
            s = p.sample_base + sl;
            id = (int)((size_t)sl * ((size_t)p.local_rows * p.width) + (size_t)ly * p.width + x);
            out_seg = (int)blockIdx.x * (kBlock / 64) + wave;
            if (valid) {
                // stream of (pixel, sample): the reference's per-pixel seed (raycast.cu:190: int idx * 1000) plus the sample index
                Xorwow rng;
                xorwow_init(rng, (unsigned long long)((long long)(int32_t)((uint32_t)(y * p.width + x) * 1000u) + (long long)s));
                float px = (float)x, py = (float)y;
                if (s > 0) { px = px + (xorwow_uniform(rng) - 0.5f); py = py + (xorwow_uniform(rng) - 0.5f); }
                org = v3(f.origin[0], f.origin[1], f.origin[2]);
                dir = camera_direction(f, px, py);
            }
        } else {
            out_seg = group * K + pass;
            const int i = pass * 64 + lane;                     // item i of the group's concatenated segments
            valid = i < total;
            // the segment holding item i = the first lane whose inclusive count exceeds i
            int lo = 0;
            for (int step = 32; step > 0; step >>= 1) {
                const int v = __shfl(incl, lo + step - 1);
                if (v <= i) lo += step;
            }
            lo = lo > 63 ? 63 : lo;
            const int rank = i - (__shfl(incl, lo) - __shfl(count, lo));
            const int arr = lo < 32 ? p.exq_in1 : p.exq_in2;
            qin = MODE == kExShadow ? p.exq_s : p.exq_a[arr < 0 ? 0 : arr];
            in_slot = ((size_t)(group * K + (lo & 31)) << 6) + (size_t)(valid ? rank : 0);
            if (valid) {
                const float4 q0 = qin[in_slot];
                id = __float_as_int(q0.w);
                org = v3(q0.x, q0.y, q0.z);
                if constexpr (MODE == kExBounce) {
                    const float4 q1 = qin[cap + in_slot];
                    dir = v3(q1.x, q1.y, q1.z);
                    pops = __float_as_int(q1.w);
                } else {
                    org = org + sun * (float)1e-4;                                  // shadow ray from the hit location, raycast.cu:262-266
                    dir = sun;
                }
            }
        }

        Hit hit;
        hit.min = FLT_MAX; hit.slot = -1; hit.instance = -1; hit.u = 0.0f; hit.v = 0.0f;
        hit.loc = v3(0.0f, 0.0f, 0.0f);
        if (valid) {
            if constexpr (MODE == kExShadow) hit = cast_ray_ex<false, false, true>(p, org, dir, stack, pops);   // only hit-or-miss survives
            else hit = cast_ray_ex(p, org, dir, stack, pops);
        }

        // ---- after the cast: the rest of the path state comes from the item (GEN: from the pixel), not across the loop ----
        int action = 0;                                         // 1 = path ended, 2 = push S(depth), 3 = push A(depth + 1)
        V3 weight = v3(1.0f, 1.0f, 1.0f), sample = v3(0.0f, 0.0f, 0.0f);
        V3 base = v3(0.0f, 0.0f, 0.0f), n = v3(0.0f, 0.0f, 0.0f), loc = v3(0.0f, 0.0f, 0.0f);
        int mat_index = 0;
        Xorwow rng;
        rng.d = 0; rng.v[0] = rng.v[1] = rng.v[2] = rng.v[3] = rng.v[4] = 0;
        if (valid) {
            if constexpr (MODE == kExGen) {
                xorwow_init(rng, (unsigned long long)((long long)(int32_t)((uint32_t)(y * p.width + x) * 1000u) + (long long)s));
                if (s > 0) { (void)xorwow_next(rng); (void)xorwow_next(rng); }     // the two jitter values drawn above
            } else {
                const float4 q2 = qin[2 * cap + in_slot], q3 = qin[3 * cap + in_slot], q4 = qin[4 * cap + in_slot];
                weight = v3(q2.x, q2.y, q2.z); sample = v3(q3.x, q3.y, q3.z);
                rng.d = __float_as_uint(q2.w); rng.v[0] = __float_as_uint(q3.w);
                rng.v[1] = __float_as_uint(q4.x); rng.v[2] = __float_as_uint(q4.y); rng.v[3] = __float_as_uint(q4.z); rng.v[4] = __float_as_uint(q4.w);
            }
            float illum = 1.0f;
            bool shade = false;
            if constexpr (MODE != kExShadow) {
                if (hit.min == FLT_MAX) { sample = sample + weight * sky; action = 1; }
                else {
                    base = base_colour(p, hit);
                    n = hit_normal(p, hit);
                    loc = hit.loc;
                    mat_index = p.instances[hit.instance].material_index;
                    shade = true;
                    if (p.lighting) {                               // raycast.cu:249-287 with the commented lines active
                        const float cos_illum = dot(n, sun);
                        illum = (float)(0.4 * (double)cos_illum);
                        if (dot(n, sun) > 0) { action = 2; shade = false; }
                    }
                }
            } else {
                const float4 q1 = qin[cap + in_slot], q5 = qin[5 * cap + in_slot], q6 = qin[6 * cap + in_slot];
                dir = v3(q1.x, q1.y, q1.z);                         // the direction the path arrived with
                pops += __float_as_int(q1.w);
                base = v3(q5.x, q5.y, q5.z); mat_index = __float_as_int(q5.w);
                n = v3(q6.x, q6.y, q6.z);
                const float4 q0 = qin[in_slot];                     // (org was advanced along the sun: the location comes from the item again)
                loc = v3(q0.x, q0.y, q0.z);
                const float cos_illum = dot(n, sun);
                illum = (float)(0.4 * (double)cos_illum);
                if (hit.min == FLT_MAX) illum = (float)(1.0 * (double)cos_illum);
                shade = true;
            }
            if (shade) {
                illum = fminf(1.0f, illum);                         // raycast.cu:289-290
                illum = fmaxf(0.4f, illum);
                const V3 local = v3(illum * base.x, illum * base.y, illum * base.z);
                const DevMaterial& mat = p.materials[mat_index];
                const float m = p.depth < p.bounces ? mat.metallic : 0.0f;
                sample = sample + weight * (local * (1.0f - m));
                action = 1;
                if (m > 0.0f) {
                    weight = weight * (base * m);
                    const float k = 2.0f * dot(dir, n);
                    V3 r = dir - n * k;
                    if (mat.roughness > 0.0f) {
                        float rx = 2.0f * xorwow_uniform(rng) - 1.0f, ry = 2.0f * xorwow_uniform(rng) - 1.0f, rz = 2.0f * xorwow_uniform(rng) - 1.0f;
                        r = r + v3(rx, ry, rz) * mat.roughness;
                    }
                    r = normalize(r);
                    org = loc + r * (float)1e-4;
                    dir = r;
                    action = 3;
                }
            }
        }

        // ---- push (ballot / popc compaction into this wave's segment) and store: the whole wave arrives here together ----
        const unsigned long long below = (1ull << lane) - 1ull;
        if (MODE != kExShadow && p.lighting) {
            const unsigned long long mask = __ballot(action == 2);
            if (lane == 0) p.exq_cnt_s[out_seg] = __popcll(mask);
            if (action == 2) {
                float4* q = p.exq_s + ((size_t)out_seg << 6) + __popcll(mask & below);
                q[0] = f4(loc, id);
                q[cap] = f4(dir, pops);
                q[2 * cap] = f4(weight, rng.d);
                q[3 * cap] = f4(sample, rng.v[0]);
                q[4 * cap] = make_float4(__uint_as_float(rng.v[1]), __uint_as_float(rng.v[2]), __uint_as_float(rng.v[3]), __uint_as_float(rng.v[4]));
                q[5 * cap] = f4(base, mat_index);
                q[6 * cap] = f4(n, 0.0f);
            }
        }
        if (p.depth < p.bounces) {
            const unsigned long long mask = __ballot(action == 3);
            if (lane == 0) p.exq_cnt_a[p.exq_out][out_seg] = __popcll(mask);
            if (action == 3) {
                float4* q = p.exq_a[p.exq_out] + ((size_t)out_seg << 6) + __popcll(mask & below);
                q[0] = f4(org, id);
                q[cap] = f4(dir, pops);
                q[2 * cap] = f4(weight, rng.d);
                q[3 * cap] = f4(sample, rng.v[0]);
                q[4 * cap] = make_float4(__uint_as_float(rng.v[1]), __uint_as_float(rng.v[2]), __uint_as_float(rng.v[3]), __uint_as_float(rng.v[4]));
            }
        }
        if (action == 1) p.ex_samples[id] = make_float4(sample.x, sample.y, sample.z, __int_as_float(pops));
    }
}

// pixel = u8(sum of its samples in index order / spp * 255); a frame with many samples arrives in several chunks of
// `count` samples, the running sum (and node-pop total) waits in ex_acc in between.
__global__ void resolve_ex_kernel(const RenderParams p, int count, int first, int last)
{
    const size_t npix = (size_t)p.local_rows * p.width;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    V3 acc = v3(0.0f, 0.0f, 0.0f);
    int pops = 0;
    if (!first) { const float4 a = p.ex_acc[i]; acc = v3(a.x, a.y, a.z); pops = __float_as_int(a.w); }
    for (int k = 0; k < count; k++) {
        const float4 q = p.ex_samples[(size_t)k * npix + i];
        acc = acc + v3(q.x, q.y, q.z);
        pops += __float_as_int(q.w);
    }
    if (!last) { p.ex_acc[i] = make_float4(acc.x, acc.y, acc.z, __int_as_float(pops)); return; }
    const int ly = (int)(i / p.width), x = (int)(i % p.width);
    const int y = ((ly / p.stripe_rows) * p.num_ranks + p.rank) * p.stripe_rows + ly % p.stripe_rows;
    const float nspp = (float)p.spp;
    uint8_t* out = p.frames[0].img + (size_t)ly * p.pitch + 3 * (size_t)x;
    out[0] = to_u8(acc.x / nspp * 255.0f);
    out[1] = to_u8(acc.y / nspp * 255.0f);
    out[2] = to_u8(acc.z / nspp * 255.0f);
    if (p.total_pops) p.total_pops[(size_t)y * p.width + x] = pops;
}

// rows of a rank-major gathered buffer back into frame order (rt_unstripe): a plain row copy, T = uint4 when every
// address involved is 16-byte aligned (the usual case: tight or hipMallocPitch pitches), bytes otherwise.
template <class T>
__global__ void unstripe_kernel(const uint8_t* __restrict__ src, size_t local_pitch, size_t rank_stride, size_t src_frame_stride,
                                uint8_t* __restrict__ dst, size_t pitch, size_t dst_frame_stride, int row_units, int height,
                                int stripe_rows, int num_ranks, int rotate_first)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // unit i of the frame: row i / row_units
    const int y = (int)(i / (size_t)row_units), u = (int)(i % (size_t)row_units);
    if (y >= height) return;
    src += (size_t)blockIdx.z * src_frame_stride;               // blockIdx.z = frame of the batch
    dst += (size_t)blockIdx.z * dst_frame_stride;
    const int stripe = y / stripe_rows;
    int rank = stripe % num_ranks;                              // the stripe's owner ...
    // ... which, when ownership rotates with the frame index (rt_render_stripes_batch_rotating), rank r plays for frame index fi
    // iff (r + fi) % num_ranks == owner
    if (rotate_first >= 0) rank = (rank + num_ranks - (rotate_first + (int)blockIdx.z) % num_ranks) % num_ranks;
    const int ly = (stripe / num_ranks) * stripe_rows + y % stripe_rows;
    const T* s = (const T*)(src + (size_t)rank * rank_stride + (size_t)ly * local_pitch);
    T* d = (T*)(dst + (size_t)y * pitch);
    d[u] = s[u];
}

// Heavy-first dispatch order of the next single-frame (or thin striped) launch: a counting sort of the tiles by cost -- the loop
// iterations of the longest lane among the tile and its neighbours in the launches since the last sort, longest first.  A tile's key is
// computed once and kept (a render on another stream may be rewriting the costs): whatever the values, the result is a permutation
// of the tiles.  The sort runs on the scene's side stream, but a hipDeviceSynchronize waits for it like for everything else -- the
// reference's loop synchronises the device every two frames (kernel.cu:279) -- so it has to be short AND has to get going: every
// thread loads eight costs before it touches any of them (one memory latency per 2048 tiles).
constexpr int kSortKeys = 1024, kSortThreads = 256, kSortBatch = 8;
// Round 6: ONE workgroup of 256 threads (was 1 024).  Primary launches are one-wave workgroups that refill a wave slot the moment it
// falls free; a 1 024-thread workgroup needs sixteen free slots on ONE compute unit at the same moment, which a saturated chip offers
// when the frames END -- the sort then ran after them, and a device synchronise (the reference's loop: one every two frames) waited
// for it.  Four waves find room within microseconds.  The costs arrive as neighbourhood maxima (render_kernel<.., ORDERED>), so the
// sort reads one value per tile; it clears what it has read (the launches accumulate maxima).
__global__ __launch_bounds__(kSortThreads) void tile_sort_kernel(int32_t* cost, int ntiles, int32_t* __restrict__ keys, int32_t* __restrict__ order)
{
    constexpr int kPer = kSortKeys / kSortThreads;              // classes per thread in the scan
    static_assert(kSortKeys % kSortThreads == 0, "whole classes per thread");
    __shared__ int count[kSortKeys], scan[kSortThreads];
    const int t = threadIdx.x;
    for (int c = t; c < kSortKeys; c += kSortThreads) count[c] = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += kSortThreads * kSortBatch) {
        int k[kSortBatch];
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = base + j * kSortThreads + t;
            k[j] = i < ntiles ? cost[i] : -1;                   // iterations of the longest lane (a few hundred at most) of the 3 x 3 tiles around i
            if (i < ntiles) cost[i] = 0;                        // (a launch running beside this sort may lose a cost it had just written: that
                                                                // tile is ordered by its neighbours' and earlier frames' costs, or late, once)
        }
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = base + j * kSortThreads + t;
            if (i < ntiles) {
                int c = k[j] < 0 ? 0 : (k[j] > kSortKeys - 1 ? kSortKeys - 1 : k[j]);
                c = kSortKeys - 1 - c;                          // longest first
                keys[i] = c;
                atomicAdd(&count[c], 1);
            }
        }
    }
    __syncthreads();
    // exclusive prefix over the classes (each thread its kPer consecutive classes, Hillis-Steele over the threads' sums): count[]
    // becomes the running cursor of each class
    int own[kPer], mine = 0;
#pragma unroll
    for (int c = 0; c < kPer; c++) { own[c] = count[t * kPer + c]; mine += own[c]; }
    scan[t] = mine;
    __syncthreads();
    for (int o = 1; o < kSortThreads; o <<= 1) {
        const int v = t >= o ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    int at = scan[t] - mine;
#pragma unroll
    for (int c = 0; c < kPer; c++) { count[t * kPer + c] = at; at += own[c]; }
    __syncthreads();
    for (int base = 0; base < ntiles; base += kSortThreads * kSortBatch) {
        int k[kSortBatch];
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = base + j * kSortThreads + t;
            k[j] = i < ntiles ? keys[i] : 0;                    // (this thread's own stores of the first pass)
        }
#pragma unroll
        for (int j = 0; j < kSortBatch; j++) {
            const int i = base + j * kSortThreads + t;
            if (i < ntiles) order[atomicAdd(&count[k[j]], 1)] = i;
        }
    }
}

// ---- refit of a deforming mesh (rt_scene_refit_mesh): same topology, new vertex positions ----------------------------
// The reference has no refit (it can only re-pose instances, Scene.cpp:67-74); SURVEY.md 8(f) item 2 lists it next to the
// build.  Result = the tree BVHTree::fill's bounds pass (BVHTree.hpp:206-209) would give every node of the SAME tree over
// the moved triangles: a node's box is the exact min / max over its triangles' vertices, so folding children's boxes
// bottom-up gives the same values as folding the triangles.
__global__ void refit_triangles_kernel(float4* __restrict__ records, const int32_t* __restrict__ tri_id, int slot_base, int num_slots,
                                       const float* __restrict__ vertices, const float* __restrict__ normals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_slots) return;
    const int slot = slot_base + i;
    const int t = tri_id[slot];
    const float* v = vertices + 9 * (size_t)t;
    const float* nn = normals + 3 * (size_t)t;
    const V3 v0 = v3(v[0], v[1], v[2]), v1 = v3(v[3], v[4], v[5]), v2 = v3(v[6], v[7], v[8]);
    const V3 e0 = v2 - v0, e1 = v1 - v0;                        // TrianglePrimitive.hpp:154-155 (as rt_scene_upload)
    const float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1);
    const float inv = 1.0f / (d00 * d11 - d01 * d01);          // TrianglePrimitive.hpp:164
    float4* q = records + (size_t)slot * 4;
    q[0] = make_float4(v0.x, v0.y, v0.z, nn[0]);
    q[1] = make_float4(nn[1], nn[2], e0.x, e0.y);
    q[2] = make_float4(e0.z, e1.x, e1.y, e1.z);
    q[3] = make_float4(d00, d01, d11, inv);
}

// box of the subtree behind `ref`: a leaf's triangles' vertices, or the union of the two boxes an interior record holds
__device__ __forceinline__ void refit_child_box(const float4* records, const int32_t* tri_id, const int32_t* leaf_count, const float* vertices,
                                                int32_t ref, float* mn, float* mx)
{
    for (int c = 0; c < 3; c++) { mn[c] = FLT_MAX; mx[c] = -FLT_MAX; }
    if (ref >= 0) {
        const float4* q = records + (size_t)ref * 4;
        const float4 q0 = q[0], q1 = q[1], q2 = q[2];
        const float a[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y}, b[6] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        for (int c = 0; c < 3; c++) { mn[c] = fminf(fminf(mn[c], a[c]), b[c]); mx[c] = fmaxf(fmaxf(mx[c], a[3 + c]), b[3 + c]); }
    } else {
        const int slot = ref & kSlotMask;
        int count = (ref >> kSlotBits) & 31;
        if (count == 31) count = leaf_count[slot];
        for (int k = 0; k < count; k++) {
            const float* v = vertices + 9 * (size_t)tri_id[slot + k];
            for (int j = 0; j < 3; j++)
                for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], v[3 * j + c]); mx[c] = fmaxf(mx[c], v[3 * j + c]); }
        }
    }
}

// one level of interior nodes, deepest level first: sched[begin .. end) are the record indices of the level
__global__ void refit_level_kernel(float4* records, const int32_t* __restrict__ tri_id, const int32_t* __restrict__ leaf_count,
                                   const float* __restrict__ vertices, const int32_t* __restrict__ sched, int begin, int end, int32_t* mesh_flag)
{
    const int i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    float4* q = records + (size_t)sched[i] * 4;
    const float4 q3 = q[3];
    float amn[3], amx[3], bmn[3], bmx[3];
    refit_child_box(records, tri_id, leaf_count, vertices, __float_as_int(q3.x), amn, amx);
    refit_child_box(records, tri_id, leaf_count, vertices, __float_as_int(q3.y), bmn, bmx);
    q[0] = make_float4(amn[0], amn[1], amn[2], amx[0]);
    q[1] = make_float4(amx[1], amx[2], bmn[0], bmn[1]);
    q[2] = make_float4(bmn[2], bmx[0], bmx[1], bmx[2]);
    const float w[12] = {amn[0], amn[1], amn[2], amx[0], amx[1], amx[2], bmn[0], bmn[1], bmn[2], bmx[0], bmx[1], bmx[2]};
    if (!boxes_ordered(w)) *mesh_flag = kBoxUnordered;          // (cleared by refit_mesh before the first level: a refit rewrites every interior record)
}

// A run of consecutive NARROW levels (the top of every tree, and the thin bottom of a deep one: each a launch of a few dozen
// threads otherwise) in one launch of one workgroup: level after level with a barrier in between.  Children written by
// this workgroup are read by this workgroup only, so the barrier orders them.
constexpr int kRefitRunLevels = 32, kRefitRunThreads = 256;
struct RefitRun { int32_t levels; int32_t bound[kRefitRunLevels + 1]; };       // level k of the run = sched[bound[k] .. bound[k + 1])
__global__ __launch_bounds__(kRefitRunThreads) void refit_run_kernel(float4* records, const int32_t* __restrict__ tri_id,
                                                                   const int32_t* __restrict__ leaf_count, const float* __restrict__ vertices,
                                                                   const int32_t* __restrict__ sched, const RefitRun run, int32_t* mesh_flag)
{
    for (int l = 0; l < run.levels; l++) {
        for (int i = run.bound[l] + (int)threadIdx.x; i < run.bound[l + 1]; i += kRefitRunThreads) {
            float4* q = records + (size_t)sched[i] * 4;
            const float4 q3 = q[3];
            float amn[3], amx[3], bmn[3], bmx[3];
            refit_child_box(records, tri_id, leaf_count, vertices, __float_as_int(q3.x), amn, amx);
            refit_child_box(records, tri_id, leaf_count, vertices, __float_as_int(q3.y), bmn, bmx);
            q[0] = make_float4(amn[0], amn[1], amn[2], amx[0]);
            q[1] = make_float4(amx[1], amx[2], bmn[0], bmn[1]);
            q[2] = make_float4(bmn[2], bmx[0], bmx[1], bmx[2]);
            const float w[12] = {amn[0], amn[1], amn[2], amx[0], amx[1], amx[2], bmn[0], bmn[1], bmn[2], bmx[0], bmx[1], bmx[2]};
            if (!boxes_ordered(w)) *mesh_flag = kBoxUnordered;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// rt_scene_update_instance_async: the new record travels as a kernel argument, so the update is ordered on the stream
// like any launch and needs no host buffer that outlives the call
__global__ void set_instance_kernel(DevInstance* dst, const DevInstance value) { *dst = value; }

// View records of the frames of one launch (RtScene::ViewPool, trace_loop<.., VIEW>): for every frame (blockIdx.y) and every
// instance, the instance's interior records with `box - origin` in place of the twelve box words -- origin = the frame's camera
// position in the instance's mesh space, computed by the function the traversal uses (to_mesh_space), the subtraction being
// the one box_differences does per visit: the same fp32 operation on the same operands, done once.  One thread per 16 bytes.
struct ViewJob {
    int32_t n;                                  // instances
    int32_t first[kMaxViewInstances];           // first record of the instance's part of a frame's view
    int32_t node_base[kMaxViewInstances];       // first interior record of the instance's mesh
    int32_t count[kMaxViewInstances];           // interior-record capacity of that mesh
    int32_t total;                              // records to write per frame (sum of count)
};

__global__ __launch_bounds__(256) void view_records_kernel(const RenderParams p, const ViewJob job)
{
    const int t = (int)(blockIdx.x * 256 + threadIdx.x), frame = (int)blockIdx.y;
    int rec = t >> 2;
    const int quarter = t & 3;
    if (rec >= job.total) return;
    int i = 0;
    while (i + 1 < job.n && rec >= job.count[i]) { rec -= job.count[i]; i++; }
    const FrameParams& f = p.frames[frame];
    const V3 o = to_mesh_space(p.instances[i], v3(f.origin[0], f.origin[1], f.origin[2]), v3(0.0f, 0.0f, 1.0f)).ro;
    float4 v = p.records[(size_t)(job.node_base[i] + rec) * 4 + quarter];
    if (quarter == 0) v = make_float4(v.x - o.x, v.y - o.y, v.z - o.z, v.w - o.x);          // the operand order of box_differences
    else if (quarter == 1) v = make_float4(v.x - o.y, v.y - o.z, v.z - o.x, v.w - o.y);
    else if (quarter == 2) v = make_float4(v.x - o.z, v.y - o.x, v.z - o.y, v.w - o.z);
    char* view = (char*)p.records + p.view_base + (size_t)frame * p.view_frame_stride;
    ((float4*)view)[(size_t)(job.first[i] + rec) * 4 + quarter] = v;
}

}  // namespace

// =====================================================================================
//                                  host side of the C-ABI
// =====================================================================================

struct RtTimer { hipEvent_t start, stop; };

#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

namespace {

DevInstance make_dev_instance(const RtInstanceDesc& d, const RtScene& s)
{
    DevInstance o;
    memset(&o, 0, sizeof o);
    o.q_rot = euler2quat(v3(d.rotation[0], d.rotation[1], d.rotation[2]));
    o.q_pose = euler2quat(v3(d.pose[3], d.pose[4], d.pose[5]));
    o.q_inv_pose = euler2quat(v3(d.inv_pose[3], d.inv_pose[4], d.inv_pose[5]));
    o.q_inv_rot = euler2quat(v3(d.inv_rotation[0], d.inv_rotation[1], d.inv_rotation[2]));
    for (int k = 0; k < 3; k++) {
        o.pose_xyz[k] = d.pose[k]; o.inv_pose_xyz[k] = d.inv_pose[k];
        o.scale[k] = d.scale[k]; o.inv_scale[k] = d.inv_scale[k];
    }
    o.identity_inv = 1;
    for (int k = 0; k < 3; k++) if (!(d.scale[k] == 1.0f && d.inv_pose[k] == 0.0f)) o.identity_inv = 0;
    if (!(o.q_inv_pose.x == 1.0f && o.q_inv_pose.y == 0.0f && o.q_inv_pose.z == 0.0f && o.q_inv_pose.w == 0.0f)) o.identity_inv = 0;
    o.unit_inv = (d.scale[0] == 1.0f && d.scale[1] == 1.0f && d.scale[2] == 1.0f &&
                  o.q_inv_pose.x == 1.0f && o.q_inv_pose.y == 0.0f && o.q_inv_pose.z == 0.0f && o.q_inv_pose.w == 0.0f) ? 1 : 0;
    o.root_ref = s.mesh_root_ref[d.mesh_index];
    o.exact_uv = s.mesh_exact_uv[d.mesh_index];
    o.material_index = d.material_index;
    o.mesh_index = d.mesh_index;
    return o;
}

bool instance_ok(const RtInstanceDesc& d, const RtScene& s)
{
    return d.mesh_index >= 0 && d.mesh_index < (int)s.mesh_root_ref.size() &&
           d.material_index >= 0 && d.material_index < s.num_materials;
}

template <class T>
int upload(T** dptr, const std::vector<T>& h, size_t& total)
{
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    RT_HIP(hipMalloc((void**)dptr, bytes));
    if (!h.empty()) RT_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    total += bytes;
    return 0;
}

int32_t leaf_ref(int64_t slot, int count) { return kLeafFlag | ((count <= 30 ? count : 31) << kSlotBits) | (int32_t)slot; }

bool camera_ok(const RtCameraParams* cam) { return cam && cam->width > 0 && cam->height > 0; }

// count cameras (same width/height) -> the per-frame part of the launch parameters
int fill_params(RenderParams& p, const RtScene* s, const RtCameraParams* cams, uint8_t* const* d_imgs, int count, size_t pitch)
{
    if (!s || !cams || !d_imgs || count < 1 || count > kMaxBatch || !camera_ok(&cams[0]) || pitch < (size_t)cams[0].width * 3) return RT_E_INVALID;
    memset(&p, 0, sizeof p);
    p.width = cams[0].width; p.height = cams[0].height;
    p.num_frames = count;
    for (int i = 0; i < count; i++) {
        const RtCameraParams* cam = &cams[i];
        if (cam->width != p.width || cam->height != p.height || !d_imgs[i]) return RT_E_INVALID;
        FrameParams& f = p.frames[i];
        memcpy(f.kinv, cam->K_inv, sizeof f.kinv);
        memcpy(f.D, cam->D, sizeof f.D);
        f.origin[0] = cam->camera_pose[0]; f.origin[1] = cam->camera_pose[1]; f.origin[2] = cam->camera_pose[2];
        f.q_cam = euler2quat(v3(cam->inv_camera_pose[3], cam->inv_camera_pose[4], cam->inv_camera_pose[5]));
        f.img = d_imgs[i];
        f.rank = 0; f.local_rows = p.height;
    }
    p.records = s->d_records; p.tri_uv = s->d_tri_uv; p.tri_id = s->d_tri_id;
    p.leaf_count = s->d_leaf_count; p.mesh_flags = s->d_mesh_flags;
    p.instances = s->d_instances; p.materials = s->d_materials;
    p.num_instances = (int32_t)s->instances.size();
    p.stack_depth = s->max_stack;
    p.pitch = pitch;
    p.local_rows = p.height; p.stripe_rows = p.height; p.rank = 0; p.num_ranks = 1;
    return RT_OK;
}

// RT_TRACE_FILE diagnostics: a zeroed [waves][16] u64 buffer for the kernel's stamps, written to the file afterwards
hipError_t trace_begin(RenderParams& p, size_t n)
{
    hipError_t e = hipMalloc((void**)&p.trace, n * 8);
    return e != hipSuccess ? e : hipMemset(p.trace, 0, n * 8);
}
hipError_t trace_end(RenderParams& p, size_t n, const char* path, hipStream_t stream)
{
    std::vector<unsigned long long> h(n);
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipMemcpy(h.data(), p.trace, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(p.trace);
    p.trace = nullptr;
    if (e == hipSuccess)
        if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, n, f); fclose(f); }
    return e;
}

// Launch with heavy-first tile order.  The order a launch reads was sorted from the costs of an earlier
// frame by tile_sort_kernel on the scene's own side stream, so sorting never sits between two frames on the caller's
// stream; the host switches to a new order when it finds its sort finished (an event query, no wait) and keeps at most one
// sort in flight per frame size: a sort writes the buffer no queued or running launch reads (every launch issued before the
// sort used the other buffer or has finished -- the sort waits for the last ordered launch of every stream that renders this
// size, up to four at a time; a stream slot whose launch has finished is handed to the next new stream -- and every launch
// issued while the sort is pending still reads the other buffer).  Nothing here waits on the host or synchronises the
// device, except when a FIFTH frame size evicts an idle state (its arrays are freed); a launch that finds no state it
// may use renders in natural order.
namespace {
// whether every ordered launch issued on the entry's stream has finished.  Launches since the stream's last recorded event (dirty):
// an idle stream has finished them all; on a busy one the event is recorded now, behind them, and tells a later call.
bool seen_finished(RtScene::TileOrder::Seen& e)
{
    if (e.dirty) {
        const hipError_t q = hipStreamQuery(e.stream);
        if (q == hipSuccess) { e.dirty = e.recorded = false; return true; }
        (void)hipGetLastError();
        if (q != hipErrorNotReady) { e.dirty = e.recorded = false; return true; }   // (a stream the application has destroyed: its work is over)
        e.dirty = false;
        if (hipEventRecord(e.done, e.stream) != hipSuccess) { (void)hipGetLastError(); return true; }
        e.recorded = true;
        return false;
    }
    if (!e.recorded) return true;                               // (never launched, or everything it launched was seen finished)
    const bool done = hipEventQuery(e.done) == hipSuccess;
    if (!done) (void)hipGetLastError();
    return done;
}

bool order_state_idle(RtScene::TileOrder& o)
{
    if (o.pending) { if (hipEventQuery(o.sort_done) != hipSuccess) return false; o.cur = o.target; o.pending = false; }
    for (auto& e : o.seen) if (e.used && !seen_finished(e)) return false;
    return true;
}
}  // namespace

// a tree of L levels never holds more than L - 1 postponed nodes (see StackT): the kernels without the private overflow.
// RT_STACK_SPILL=1 forces the general kernels (read once per process: tests/test_gpu_parity.py runs a parity scene in a child
// process with it; test_stack_depth_at_the_lds_boundary renders chains of 15..20 levels through whichever form their depth selects)
bool lds_stack_suffices(const RenderParams& p)
{
    static const bool forced = [] { const char* e = getenv("RT_STACK_SPILL"); return e && e[0] == '1'; }();
    return !forced && p.stack_depth - 1 <= kLdsStack;
}

// ---- view records (RtScene::ViewPool; view_records_kernel; render_kernel<.., VIEW>) -------------------------------------------
// RT_VIEW_RECORDS=0: never (tests run parity scenes both ways).  RT_VIEW_MIN_RAYS=k: a launch qualifies when a frame brings at
// least k rays per view record it costs to write (default 8: a record is written once and read by tens of rays; below that --
// a rank's thin stripes of a frame, a huge tree under a small frame -- the pre-pass would cost more than the subtractions).
// RT_VIEW_MIN_FRAMES=f: ... and when it carries at least f frames (default 4): the pre-pass is a launch of its own in front of
// the render kernel, a few microseconds that a batch shares and a single frame pays alone (measured on c2: 32 frames per
// launch -3 % / -6 % for the mid / near camera, one frame per launch +4 % / +7 %; profiles/r05_experiments/view_records_asm_ab.log).
constexpr size_t kViewMaxBytes = (size_t)3 << 30;            // records + pool stay below 4 GiB: offsets are 32 bits

int view_mode()
{
    static const int mode = [] { const char* e = getenv("RT_VIEW_RECORDS"); return e && e[0] == '0' ? 0 : 1; }();
    return mode;
}

// RT_VIEW_MAX_BYTES: what the pool of ONE scene may take (default 1 GiB; never more than keeps records + pool below 4 GiB)
size_t view_budget()
{
    static const size_t b = [] {
        const char* e = getenv("RT_VIEW_MAX_BYTES");
        size_t v = e && *e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 30;
        return v > kViewMaxBytes ? kViewMaxBytes : v;
    }();
    return b;
}

// static eligibility of the scene (decided once; the caller holds call_mu)
void view_decide(RtScene* s)
{
    RtScene::ViewPool& v = s->view;
    if (v.decided) return;
    v.decided = true;
    int32_t cap = 0;
    for (const auto& rf : s->mesh_refit) cap = std::max(cap, rf.int_cap);
    // (an instance may be given another mesh later, rt_scene_update_instance: every instance gets room for the largest)
    v.frame_records = cap * (int32_t)s->instances.size();
    v.inst_first.resize(s->instances.size());
    for (size_t i = 0; i < s->instances.size(); i++) v.inst_first[i] = (int32_t)i * cap;
    v.usable = cap > 0 && s->records_bytes > 0 && !s->instances.empty() && s->instances.size() <= (size_t)kMaxViewInstances &&
               s->records_bytes + (size_t)v.frame_records * 64 <= kViewMaxBytes;
}

// The record array moves into a new allocation with a pool of `slots` x `frames` views as its tail (slots: three, or as many as the
// budget allows).  The caller holds call_mu -- no other launch of this scene can be prepared meanwhile -- and the device is drained
// first: nothing reads the old block afterwards, so it is freed here.  RT_OK, RT_E_NOMEM (budget or allocation), or a HIP error.
int view_resize(RtScene* s, int frames)
{
    RtScene::ViewPool& v = s->view;
    const size_t frame_bytes = (size_t)v.frame_records * 64, base = (s->records_bytes + 255) & ~(size_t)255;
    const size_t room = std::min(view_budget(), kViewMaxBytes > base ? kViewMaxBytes - base : 0);
    int slots = 3;
    while (slots > 1 && frame_bytes * (size_t)slots * (size_t)frames > room) slots--;
    if (frames < 1 || frame_bytes * (size_t)slots * (size_t)frames > room) return RT_E_NOMEM;
    const size_t total = base + frame_bytes * (size_t)slots * (size_t)frames;
    float4* block = nullptr;
    RT_HIP(hipDeviceSynchronize());
    if (hipMalloc((void**)&block, total) != hipSuccess) { (void)hipGetLastError(); return RT_E_NOMEM; }
    hipError_t e = hipMemcpy(block, s->d_records, s->records_bytes, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { (void)hipFree(block); return (int)e; }
    (void)hipFree(s->d_records);
    s->device_bytes += total;
    s->device_bytes -= std::min(s->device_bytes, s->records_alloc_bytes);
    s->d_records = block;
    s->records_alloc_bytes = total;
    v.base_bytes = base; v.slot_frames = frames; v.slots = slots; v.grows++;
    for (auto& sl : v.slot) { sl.used = false; sl.stream = nullptr; }
    return RT_OK;
}

// Decides whether this launch renders through view records; if so: a slot of the pool (grown when needed and allowed), the launch's
// view parameters in `p`, the pre-pass queued on `stream`.  Returns the slot (view_done records its event after the render kernel) or
// -1: the launch then runs the kernels without views -- same pixels.  The caller holds call_mu from here to view_done.
// `samples`: primary rays per pixel (the samples-only extension kernel: its jittered rays share the frame's origin too).
int view_prepare(RtScene* s, RenderParams& p, hipStream_t stream, int samples = 1)
{
    if (!view_mode() || !s || p.num_instances < 1 || p.num_instances > kMaxViewInstances || p.num_ranks < 1) return -1;
    RtScene::ViewPool& v = s->view;
    view_decide(s);
    if (!v.usable || (size_t)p.num_instances != s->instances.size()) return -1;
    ViewJob job;
    job.n = p.num_instances; job.total = 0;
    for (int i = 0; i < job.n; i++) {
        const RtScene::MeshRefit& rf = s->mesh_refit[(size_t)s->instances[(size_t)i].mesh_index];
        job.first[i] = v.inst_first[(size_t)i]; job.node_base[i] = rf.node_base; job.count[i] = rf.int_cap;
        job.total += rf.int_cap;
    }
    static const long min_rays = [] { const char* e = getenv("RT_VIEW_MIN_RAYS"); long k = e ? atol(e) : 8; return k < 0 ? 0 : k; }();
    static const int min_frames = [] { const char* e = getenv("RT_VIEW_MIN_FRAMES"); int k = e ? atoi(e) : 4; return k < 1 ? 1 : k; }();
    if (job.total <= 0 || (long long)p.num_frames * samples < min_frames || (long long)job.total * min_rays > (long long)p.width * p.local_rows * samples) {
        if (getenv("RT_VIEW_DEBUG")) fprintf(stderr, "rt view: not for this launch (records %d, frames %d of %d, rays per frame %lld, rays per record wanted %ld)\n",
                                             job.total, p.num_frames, min_frames, (long long)p.width * p.local_rows, min_rays);
        return -1;
    }
    auto count = [&](uint64_t RtScene::ViewPool::*field) { std::lock_guard<std::mutex> lock(v.m); (v.*field)++; };
    count(&RtScene::ViewPool::launches);
    const size_t frame_bytes = (size_t)v.frame_records * 64;
    if (p.num_frames > v.slot_frames) {
        // the pool is too small for this launch.  Reserved by the application: it stays as it is.  Else it grows to the next of
        // 1 / 4 / 8 / 16 / 32 frames per slot -- or to exactly this launch's frames where the budget allows no more -- in a call
        // that blocks until the device is idle (at most five times in a scene's life; see rt_render_batch in rt_hip.h)
        if (v.reserved) { count(&RtScene::ViewPool::fallbacks); return -1; }
        int want = 1;
        while (want < p.num_frames) want = want == 1 ? 4 : want * 2;
        want = std::min(want, (int)kMaxBatch);
        int rc = view_resize(s, want);
        if (rc == RT_E_NOMEM && want != p.num_frames) rc = view_resize(s, p.num_frames);
        if (rc != RT_OK) {
            if (rc != RT_E_NOMEM) v.usable = false;             // (a HIP error: this scene renders without views from now on)
            count(&RtScene::ViewPool::fallbacks);
            return -1;
        }
    }
    int use = -1;
    for (int k = 0; k < v.slots && use < 0; k++) if (v.slot[k].used && v.slot[k].stream == stream) use = k;   // stream order protects it
    for (int k = 0; k < v.slots && use < 0; k++) if (!v.slot[k].used) use = k;
    for (int k = 0; k < v.slots && use < 0; k++) {                                                            // its last launch has finished
        if (hipEventQuery(v.slot[k].done) == hipSuccess) use = k;
        else (void)hipGetLastError();
    }
    if (use < 0) { count(&RtScene::ViewPool::fallbacks); return -1; }                                        // (all busy on other streams)
    if (!v.slot[use].done && hipEventCreateWithFlags(&v.slot[use].done, hipEventDisableTiming) != hipSuccess) { count(&RtScene::ViewPool::fallbacks); return -1; }
    v.slot[use].used = true; v.slot[use].stream = stream;
    p.records = s->d_records;                                   // (fill_params read it before a possible move)
    p.view_base = (uint32_t)(v.base_bytes + (size_t)use * v.slot_frames * frame_bytes);
    p.view_frame_stride = (uint32_t)frame_bytes;
    for (int i = 0; i < job.n; i++) p.view_inst_off[i] = (int32_t)(((int64_t)job.first[i] - job.node_base[i]) * 64);
    hipLaunchKernelGGL(view_records_kernel, dim3((unsigned)(((size_t)job.total * 4 + 255) / 256), (unsigned)p.num_frames), dim3(256), 0, stream, p, job);
    if (hipGetLastError() != hipSuccess) { count(&RtScene::ViewPool::fallbacks); return -1; }
    return use;
}

void view_done(RtScene* s, int slot, hipStream_t stream)
{
    if (slot < 0) return;
    (void)hipEventRecord(s->view.slot[slot].done, stream);     // (still under the call_mu view_prepare was called under)
}

int launch_ordered(RtScene* s, RenderParams& p, hipStream_t stream, int synchronize, bool view)
{
    RtScene::TileOrderCache& cache = s->order;
    std::lock_guard<std::mutex> lock(cache.m);
    const int ntiles = p.tiles_x * p.tiles_y;
    // (a high-priority side stream was tried: frames alternating between two streams went from 0.111 to 0.126 ms with it)
    if (!cache.sort_stream) RT_HIP(hipStreamCreateWithFlags(&cache.sort_stream, hipStreamNonBlocking));
    RtScene::TileOrder* o = nullptr;
    for (auto& e : cache.entry) if (e.ntiles && e.tiles_x == p.tiles_x && e.tiles_y == p.tiles_y) o = &e;
    if (!o) {                                                   // a size not seen before: a free state, else the least recently used idle one
        for (auto& e : cache.entry) if (!e.ntiles && !o) o = &e;
        if (!o) {
            RtScene::TileOrder* lru = nullptr;
            for (auto& e : cache.entry) if (!lru || e.last_used < lru->last_used) lru = &e;
            if (order_state_idle(*lru)) {
                (void)hipFree(lru->d_cost);
                lru->d_cost = lru->d_keys = lru->d_order[0] = lru->d_order[1] = nullptr;
                lru->tiles_x = lru->tiles_y = lru->ntiles = 0; lru->cur = -1; lru->pending = false; lru->launches = lru->sorted_at = 0;
                for (auto& e : lru->seen) e.used = e.dirty = e.recorded = false;
                o = lru;
            }
        }
        if (o) {
            if (!o->sort_done) {
                RT_HIP(hipEventCreateWithFlags(&o->sort_done, hipEventDisableTiming));
                for (auto& e : o->seen) RT_HIP(hipEventCreateWithFlags(&e.done, hipEventDisableTiming));
            }
            RT_HIP(hipMalloc((void**)&o->d_cost, (size_t)ntiles * 4 * sizeof(int32_t)));
            RT_HIP(hipMemsetAsync(o->d_cost, 0, (size_t)ntiles * sizeof(int32_t), stream));
            o->d_keys = o->d_cost + ntiles; o->d_order[0] = o->d_keys + ntiles; o->d_order[1] = o->d_order[0] + ntiles;
            o->tiles_x = p.tiles_x; o->tiles_y = p.tiles_y; o->ntiles = ntiles;
        }
    }
    RtScene::TileOrder::Seen* mine = nullptr;
    if (o) {
        o->last_used = ++cache.tick;
        if (o->pending && hipEventQuery(o->sort_done) == hipSuccess) { o->cur = o->target; o->pending = false; }
        for (auto& e : o->seen) if (e.used && e.stream == stream) mine = &e;
        if (!mine) for (auto& e : o->seen) if (!e.used) { mine = &e; break; }
        if (!mine) {                                            // all four slots taken: the one whose launches finished longest ago
            for (auto& e : o->seen)
                if (seen_finished(e) && (!mine || e.tick < mine->tick)) mine = &e;
        }
        if (mine) { mine->used = true; mine->stream = stream; mine->tick = cache.tick; }
    }
    p.tile_order = (mine && o->cur >= 0) ? o->d_order[o->cur] : nullptr;
    p.tile_cost = mine ? o->d_cost : nullptr;
    // A new order is sorted every RT_TILE_SORT_INTERVAL-th launch (default 4), on the side stream, from the costs the launches so far
    // have left.  It must not start before every launch that may still read the buffer it writes has finished -- every ordered launch
    // issued BEFORE this one, on any stream (this one reads d_order[cur], the sort writes the other buffer) -- so it runs WHILE this
    // frame renders: a device-wide synchronise after the frame (the reference's loop synchronises every two frames) finds it done.
    // Round 6: the events that tell it are recorded WHEN A SORT IS ISSUED, on every stream that has launched since its last one --
    // not after every launch as before: a recorded event is a marker packet between two kernels of the stream, and back-to-back
    // launches with a marker in between start 7 us apart where launches without start 0 us apart
    // (profiles/r06_experiments/single_frame_gap.md: rocprofv3 kernel-trace timestamps of 60 back-to-back launches, three settings).
    // (This launch rewrites the cost array while the sort reads it: a tile's key is computed once and kept, and whatever the
    // values, the result is a permutation.)
    static const int interval = [] { const char* e = getenv("RT_TILE_SORT_INTERVAL"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
    if (mine && !o->pending && o->launches >= 1 && (o->cur < 0 || o->launches + 1 - o->sorted_at >= (uint64_t)interval)) {
        for (auto& e : o->seen) {
            if (!e.used || !e.dirty) continue;
            e.dirty = false;
            if (hipEventRecord(e.done, e.stream) != hipSuccess) {           // (a stream the application has destroyed since: its work is over)
                (void)hipGetLastError();
                if (&e != mine) e.used = false;
                e.recorded = false;
                continue;
            }
            e.recorded = true;
        }
        // ... and the sort is QUEUED before the frame (after a synchronise it finds an empty chip); its workgroup is four waves, which
        // find room on a saturated chip too (see tile_sort_kernel)
        o->sorted_at = o->launches + 1;
        o->target = o->cur < 0 ? 0 : o->cur ^ 1;
        for (auto& e : o->seen) if (e.used && e.recorded) RT_HIP(hipStreamWaitEvent(cache.sort_stream, e.done, 0));
        hipLaunchKernelGGL(tile_sort_kernel, dim3(1), dim3(kSortThreads), 0, cache.sort_stream, o->d_cost, ntiles, o->d_keys, o->d_order[o->target]);
        RT_HIP(hipGetLastError());
        RT_HIP(hipEventRecord(o->sort_done, cache.sort_stream));
        o->pending = true;
    }
    const size_t lds = (size_t)(lds_stack_suffices(p) ? lds_block_rows<false, false, false>(p.stack_depth) : lds_block_rows<false, false, true>(p.stack_depth)) *
                       kPrimBlock * sizeof(int) + 2 * sizeof(int);
    const dim3 grid((unsigned)ntiles * (unsigned)p.num_frames);
    if (lds_stack_suffices(p)) {
        if (view) hipLaunchKernelGGL((render_kernel<false, false, true, false, true>), grid, dim3(kPrimBlock), lds, stream, p);
        else hipLaunchKernelGGL((render_kernel<false, false, true, false>), grid, dim3(kPrimBlock), lds, stream, p);
    } else {
        if (view) hipLaunchKernelGGL((render_kernel<false, false, true, true, true>), grid, dim3(kPrimBlock), lds, stream, p);
        else hipLaunchKernelGGL((render_kernel<false, false, true>), grid, dim3(kPrimBlock), lds, stream, p);
    }
    RT_HIP(hipGetLastError());
    if (o) o->launches++;
    if (mine) mine->dirty = true;
    if (synchronize) RT_HIP(hipStreamSynchronize(stream));
    return RT_OK;
}

int launch(RenderParams& p, bool debug, hipStream_t stream, int synchronize, RtScene* scene = nullptr, bool stats = false)
{
    if (p.width <= 0 || p.local_rows < 0) return RT_E_INVALID;
    if (p.local_rows == 0) return RT_OK;
    p.tiles_x = (p.width + kPrimTile - 1) / kPrimTile;
    p.tiles_y = (p.local_rows + kPrimTile - 1) / kPrimTile;
    // (one size for whichever kernel is launched below: the spare row of the optimistic stack costs the instrumented kernels nothing)
    const size_t lds = (size_t)(lds_stack_suffices(p) ? lds_rows(p.stack_depth) : lds_block_rows<false, false, true>(p.stack_depth)) * kPrimBlock * sizeof(int);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y), (unsigned)p.num_frames), block(kPrimBlock);
    const char* trace_file = getenv("RT_TRACE_FILE");                               // diagnostics only
    // Heavy-first dispatch (RT_TILE_ORDER=0 turns it off): for one frame per launch with enough tiles to have a tail worth
    // hiding.  Batches do not use it: the tail of one frame already overlaps the bulk of the next, and measured with the
    // order applied across all frames of a batch (workgroup b -> frame b % F of the tile with rank b / F) whole-frame batches ran -2 % (far) to +4 %
    // (near), a rank's stripes of 32 frames -6 % on one stream but no better than the two alternating streams bench.py uses.
    // (RT_TRACE_ORDERED=1 with RT_TRACE_FILE: the stamps of the heavy-first launch itself, for tools/trace_one.py)
    const bool trace_ordered = trace_file && getenv("RT_TRACE_ORDERED");
    const int view_slot = (scene && !debug && !trace_file) ? view_prepare(scene, p, stream) : -1;
    struct ViewDone {                                           // (whichever way this function returns after the render kernel was queued)
        RtScene* s; int slot; hipStream_t stream;
        ~ViewDone() { view_done(s, slot, stream); }
    } view_guard{scene, view_slot, stream};
    if (scene && !debug && (!trace_file || trace_ordered) && !p.hit_instance && !p.hit_triangle && grid.x >= 1024 && (size_t)grid.x * grid.y <= ((size_t)1 << 30)) {
        static const bool enabled = [] { const char* e = getenv("RT_TILE_ORDER"); return !(e && e[0] == '0'); }();
        // ... and (round 6) a rank's STRIPES of a few frames: at eight ranks a group of 20 frames is two and a half frames' worth of
        // work per launch and ends, like a single frame, in a tail of a few long waves (RT_TILE_ORDER_STRIPES=0 turns this part off)
        static const bool stripes_too = [] { const char* e = getenv("RT_TILE_ORDER_STRIPES"); return !(e && e[0] == '0'); }();
        const bool thin_stripes = stripes_too && p.num_ranks > 1 && p.num_frames > 1 && (size_t)grid.x * grid.y < (size_t)4 * 32768;
        if (enabled && !stats && ((p.num_frames == 1 && grid.x >= 8192) || thin_stripes)) {   // (8x8-pixel tiles: half a million pixels)
            if (!trace_ordered) return launch_ordered(scene, p, stream, synchronize, view_slot >= 0);
            const size_t n = (size_t)grid.x * (kPrimBlock / 64) * 16;
            RT_HIP(trace_begin(p, n));
            const int rc = launch_ordered(scene, p, stream, 1, false);
            const hipError_t e = trace_end(p, n, trace_file, stream);
            return rc ? rc : (int)e;
        }
    }
    const size_t trace_n = (size_t)grid.x * grid.y * (kPrimBlock / 64) * 16;
    if (trace_file) RT_HIP(trace_begin(p, trace_n));
    if (trace_file && getenv("RT_TRACE_PROF")) hipLaunchKernelGGL((render_kernel<false, true>), grid, block, lds, stream, p);
    else if (debug) hipLaunchKernelGGL((render_kernel<true, false>), grid, block, lds, stream, p);
    else if (stats) {                                           // rt_scene_loop_stats: the same choices as below, the instrumented copies
        if (lds_stack_suffices(p)) {
            if (view_slot >= 0) hipLaunchKernelGGL((render_kernel<false, false, false, false, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((render_kernel<false, false, false, false, false, true>), grid, block, lds, stream, p);
        } else {
            if (view_slot >= 0) hipLaunchKernelGGL((render_kernel<false, false, false, true, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((render_kernel<false, false, false, true, false, true>), grid, block, lds, stream, p);
        }
    }
    else if (lds_stack_suffices(p)) {
        if (view_slot >= 0) hipLaunchKernelGGL((render_kernel<false, false, false, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((render_kernel<false, false, false, false>), grid, block, lds, stream, p);
    } else {
        if (view_slot >= 0) hipLaunchKernelGGL((render_kernel<false, false, false, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((render_kernel<false, false>), grid, block, lds, stream, p);
    }
    RT_HIP(hipGetLastError());
    if (trace_file) RT_HIP(trace_end(p, trace_n, trace_file, stream));
    if (synchronize) RT_HIP(hipStreamSynchronize(stream));
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

// the hash of the kernel sources and build flags this binary was compiled from (cuda-raytracing_amd/_build.py passes it; a
// build by other means says so).  The text is also what _build.library_code_hash() finds in the file without loading it.
#ifndef RT_CODE_HASH
#define RT_CODE_HASH "built-without-it"
#endif
const char* rt_build_info(void)
{
    static const char info[] = "RT_CODE_HASH=" RT_CODE_HASH;
    return info;
}

int rt_device_count(int* count)
{
    if (!count) return RT_E_INVALID;
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) { *count = 0; return RT_OK; }
    return (int)e;
}
int rt_set_device(int device) { RT_HIP(hipSetDevice(device)); return RT_OK; }
int rt_malloc(void** dptr, size_t bytes) { if (!dptr) return RT_E_INVALID; RT_HIP(hipMalloc(dptr, bytes ? bytes : 1)); return RT_OK; }
int rt_malloc_pitch(void** dptr, size_t* pitch, size_t width_bytes, size_t height)
{
    if (!dptr || !pitch) return RT_E_INVALID;
    RT_HIP(hipMallocPitch(dptr, pitch, width_bytes, height));
    return RT_OK;
}
int rt_free(void* dptr) { RT_HIP(hipFree(dptr)); return RT_OK; }
int rt_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    RT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    RT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
}
int rt_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    RT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    RT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
}
int rt_memcpy2d_d2h(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height, void* stream)
{
    RT_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDeviceToHost, (hipStream_t)stream));
    RT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
}
int rt_stream_synchronize(void* stream) { RT_HIP(hipStreamSynchronize((hipStream_t)stream)); return RT_OK; }
int rt_device_synchronize(void) { RT_HIP(hipDeviceSynchronize()); return RT_OK; }

const char* rt_error_string(int code)
{
    switch (code) {
    case RT_OK: return "ok";
    case RT_E_INVALID: return "rt: invalid argument";
    case RT_E_NOMEM: return "rt: out of host memory";
    case RT_E_DEPTH: return "rt: BVH deeper than the traversal stack";
    case RT_E_NODEVICE: return "rt: no HIP device";
    case RT_E_COMM: return "rt: RCCL unavailable or failed (rt_comm_last_error)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "rt: unknown error";
    }
}

int rt_scene_upload(const RtSceneDesc* desc, RtScene** out)
{
    // tests of the callers' error paths: while RT_TEST_FAIL_UPLOAD=1 is in the environment every upload fails before it touches anything
    if (const char* e = getenv("RT_TEST_FAIL_UPLOAD")) if (e[0] == '1') return RT_E_NOMEM;
    if (!desc || !out || desc->num_meshes < 0 || desc->num_materials < 0 || desc->num_instances < 0) return RT_E_INVALID;
    if ((desc->num_meshes && !desc->meshes) || (desc->num_materials && !desc->materials) ||
        (desc->num_instances && !desc->instances)) return RT_E_INVALID;
    *out = nullptr;
    RtScene* s = new (std::nothrow) RtScene;
    if (!s) return RT_E_NOMEM;
    int rc = RT_OK;
    // One record array and one index space for the whole scene: per mesh its interior nodes, then its triangles in
    // leaf order.  The per-triangle side arrays use the same index (their entries at interior records are unused).
    // The host vectors hold only the parts of meshes that arrive WITH a tree; the part of a mesh the device builds (num_nodes == 0) is
    // a hole of `skipped` records in them: it is never materialised on the host nor copied (the device arrays start from a memset,
    // and the build clears and fills the mesh's part itself) -- for the 260 k-triangle atrium that was 50 MB of zeros written,
    // then copied from pageable memory: a third of the time from file to renderable scene.  A record's index in the device
    // arrays = its index in the vectors + the holes before it; `segments` says which vector ranges go where.
    std::vector<float4> records;
    std::vector<float> tri_uv;
    std::vector<int32_t> tri_id, leaf_count, mesh_flags;
    std::vector<int> build_on_device;               // meshes that arrive without a tree (num_nodes == 0)
    struct Segment { size_t dev_first, vec_first, count; };        // in records
    std::vector<Segment> segments;
    size_t skipped = 0;                             // records of device-built meshes so far
    try {
        for (int mi = 0; mi < desc->num_meshes && rc == RT_OK; mi++) {
            const RtMeshDesc& m = desc->meshes[mi];
            if (m.num_triangles < 0 || (m.num_triangles && (!m.vertices || !m.normals || !m.uvs))) { rc = RT_E_INVALID; break; }
            if (m.num_nodes == 0) {
                // no tree given: the mesh's part of the arrays is reserved here and the tree is built on the device, in place,
                // once the arrays are uploaded (the kernels of rt_scene_rebuild_mesh_device) -- no host tree, no host re-layout
                const int64_t n = m.num_triangles, int_cap = std::max<int64_t>(n - 1, 0);
                const int64_t node_base = (int64_t)(records.size() / 4 + skipped), slot_base = node_base + int_cap;
                if (slot_base + n + 1 > kSlotMask) { rc = RT_E_INVALID; break; }
                skipped += (size_t)(int_cap + n);                // (a hole in the host vectors: see above)
                s->mesh_root_ref.push_back(leaf_ref(slot_base, 0));
                s->mesh_exact_uv.push_back(0);
                RtScene::MeshRefit rf;
                rf.node_base = (int32_t)node_base; rf.int_cap = (int32_t)int_cap; rf.slot_cap = (int32_t)n; rf.levels = 1;
                rf.slot_base = (int32_t)slot_base; rf.num_slots = 0; rf.num_triangles = 0;
                s->mesh_refit.push_back(std::move(rf));
                build_on_device.push_back(mi);
                mesh_flags.push_back(0);                         // (the rebuild on the device sets it)
                continue;
            }
            if (m.num_nodes < 1 || m.num_leaf_indices < 0 || !m.node_bounds || !m.node_children || !m.node_leaf_first || !m.node_leaf_count ||
                (m.num_leaf_indices && !m.leaf_indices)) {
                rc = RT_E_INVALID; break;
            }
            int64_t mesh_interior = 0;
            for (int i = 0; i < m.num_nodes; i++) mesh_interior += m.node_children[2 * i] > 0 ? 1 : 0;
            // (room for any tree over these triangles: rt_scene_rebuild_mesh_device writes a new one in place)
            const int64_t int_cap = std::max<int64_t>(mesh_interior, (int64_t)m.num_triangles - 1);
            const size_t vec_first = records.size() / 4;         // where this mesh's part starts in the host vectors ...
            const int64_t node_base = (int64_t)(vec_first + skipped), slot_base = node_base + int_cap;      // ... and in the device arrays
            if (slot_base + std::max<int64_t>(m.num_leaf_indices, m.num_triangles) + 1 > kSlotMask) { rc = RT_E_INVALID; break; }
            // pass 1: entry of every node (interior -> running interior index, leaf -> slot range), levels
            std::vector<int32_t> entry((size_t)m.num_nodes), level((size_t)m.num_nodes, 0);
            int64_t n_int = 0, n_slot = 0;
            int max_level = 1;
            level[0] = 1;
            for (int i = 0; i < m.num_nodes && rc == RT_OK; i++) {
                const int a = m.node_children[2 * i], b = m.node_children[2 * i + 1];
                if (level[i] == 0) { rc = RT_E_INVALID; break; }          // unreachable or child before parent
                if (a > 0) {                                               // interior test of raycast.cu:66
                    if (a <= i || b <= i || a >= m.num_nodes || b >= m.num_nodes || level[a] || level[b] || a == b) { rc = RT_E_INVALID; break; }
                    level[a] = level[b] = level[i] + 1;
                    max_level = std::max(max_level, level[i] + 1);
                    entry[i] = (int32_t)(node_base + n_int++);
                } else {
                    const int first = m.node_leaf_first[i], count = m.node_leaf_count[i];
                    if (count < 0 || first < 0 || (int64_t)first + count > m.num_leaf_indices) { rc = RT_E_INVALID; break; }
                    entry[i] = leaf_ref(slot_base + n_slot, count);
                    n_slot += count;
                }
            }
            if (rc != RT_OK) break;
            // leaf ranges may overlap in a caller-supplied tree, so the slots actually emitted (the sum of the leaf counts)
            // can exceed num_leaf_indices: the slot field of a ref must still hold them
            if (slot_base + n_slot + 1 > kSlotMask) { rc = RT_E_INVALID; break; }
            if (max_level > kMaxStack) { rc = RT_E_DEPTH; break; }
            s->max_stack = std::max(s->max_stack, max_level);
            // pass 2: emit records
            bool exact_uv = false, unordered = false;
            const int64_t slot_cap = std::max<int64_t>(n_slot, m.num_triangles);
            records.reserve(records.size() + (size_t)(int_cap + slot_cap) * 4 + 4);
            tri_uv.reserve(tri_uv.size() + (size_t)(int_cap + slot_cap) * 6);
            tri_id.reserve(tri_id.size() + (size_t)(int_cap + slot_cap));
            leaf_count.reserve(leaf_count.size() + (size_t)(int_cap + slot_cap));
            records.resize(records.size() + (size_t)int_cap * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            tri_uv.resize(tri_uv.size() + (size_t)int_cap * 6, 0.0f);
            tri_id.resize(tri_id.size() + (size_t)int_cap, -1);
            leaf_count.resize(leaf_count.size() + (size_t)int_cap, 0);
            for (int i = 0; i < m.num_nodes && rc == RT_OK; i++) {
                const int a = m.node_children[2 * i], b = m.node_children[2 * i + 1];
                if (a > 0) {
                    const float* A = m.node_bounds + 6 * (size_t)a;
                    const float* B = m.node_bounds + 6 * (size_t)b;
                    float4* q = &records[((size_t)entry[i] - skipped) * 4];
                    q[0] = make_float4(A[0], A[1], A[2], A[3]);
                    q[1] = make_float4(A[4], A[5], B[0], B[1]);
                    q[2] = make_float4(B[2], B[3], B[4], B[5]);
                    q[3] = make_float4(i2f(entry[a]), i2f(entry[b]), 0.0f, 0.0f);
                    const float w[12] = {A[0], A[1], A[2], A[3], A[4], A[5], B[0], B[1], B[2], B[3], B[4], B[5]};
                    if (!boxes_ordered(w)) unordered = true;
                } else {
                    const int first = m.node_leaf_first[i], count = m.node_leaf_count[i];
                    for (int k = 0; k < count; k++) {
                        const int t = m.leaf_indices[first + k];
                        if (t < 0 || t >= m.num_triangles) { rc = RT_E_INVALID; break; }
                        const float* v = m.vertices + 9 * (size_t)t;
                        const float* nn = m.normals + 3 * (size_t)t;
                        const float* uv = m.uvs + 6 * (size_t)t;
                        V3 v0 = v3(v[0], v[1], v[2]), v1 = v3(v[3], v[4], v[5]), v2 = v3(v[6], v[7], v[8]);
                        V3 e0 = v2 - v0, e1 = v1 - v0;                      // TrianglePrimitive.hpp:154-155
                        float d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1);
                        float inv = 1.0f / (d00 * d11 - d01 * d01);        // TrianglePrimitive.hpp:164
                        records.push_back(make_float4(v0.x, v0.y, v0.z, nn[0]));
                        records.push_back(make_float4(nn[1], nn[2], e0.x, e0.y));
                        records.push_back(make_float4(e0.z, e1.x, e1.y, e1.z));
                        records.push_back(make_float4(d00, d01, d11, inv));
                        for (int c = 0; c < 6; c++) {
                            tri_uv.push_back(uv[c]);
                            if (!(fabsf(uv[c]) < 1e37f)) exact_uv = true;  // also catches NaN / inf
                        }
                        tri_id.push_back(t);
                        leaf_count.push_back(k == 0 ? count : 0);
                    }
                }
            }
            if (rc != RT_OK) break;
            // unused slots of the mesh's part
            const size_t vec_end = (size_t)slot_base + (size_t)slot_cap - skipped;
            records.resize(vec_end * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            tri_uv.resize(vec_end * 6, 0.0f);
            tri_id.resize(vec_end, -1);
            leaf_count.resize(vec_end, 0);
            segments.push_back({(size_t)node_base, vec_first, vec_end - vec_first});
            s->mesh_root_ref.push_back(entry[0]);
            s->mesh_exact_uv.push_back(exact_uv ? 1 : 0);
            mesh_flags.push_back(unordered ? kBoxUnordered : 0);
            {
                RtScene::MeshRefit rf;
                rf.node_base = (int32_t)node_base; rf.int_cap = (int32_t)int_cap; rf.slot_cap = (int32_t)slot_cap; rf.levels = max_level;
                rf.slot_base = (int32_t)slot_base; rf.num_slots = (int32_t)n_slot; rf.num_triangles = m.num_triangles;
                std::vector<std::vector<int32_t>> by_level((size_t)max_level + 1);
                for (int i = 0; i < m.num_nodes; i++)
                    if (m.node_children[2 * i] > 0) by_level[(size_t)level[i]].push_back(entry[i]);
                for (int l = max_level; l >= 1; l--) {
                    if (by_level[(size_t)l].empty()) continue;
                    rf.sched.insert(rf.sched.end(), by_level[(size_t)l].begin(), by_level[(size_t)l].end());
                    rf.level_end.push_back((int32_t)rf.sched.size());
                }
                s->mesh_refit.push_back(std::move(rf));
            }
        }
        if (rc == RT_OK) {
            s->num_materials = desc->num_materials;
            for (int i = 0; i < desc->num_instances; i++)
                if (!instance_ok(desc->instances[i], *s)) { rc = RT_E_INVALID; break; }
            for (int i = 0; i < desc->num_materials; i++) {
                const RtMaterialDesc& m = desc->materials[i];
                if (m.texture && (m.texture_width <= 0 || m.texture_height <= 0 || m.texture_pitch < (size_t)m.texture_width * 3)) { rc = RT_E_INVALID; break; }
            }
        }
    } catch (const std::bad_alloc&) {
        rc = RT_E_NOMEM;
    }
    if (rc != RT_OK) { delete s; return rc; }

    // ---- device upload ----
    auto fail = [&](int code) { rt_scene_destroy(s); return code; };
    hipError_t he = hipGetDevice(&s->device);
    if (he != hipSuccess) return fail(he == hipErrorNoDevice ? RT_E_NODEVICE : (int)he);
    {
        // the arrays start as what unused records hold -- zeros, triangle id -1 -- plus four padding records (a leaf's last iteration
        // looks one record ahead); the parts of meshes that came with a tree are copied over that
        const size_t total = records.size() / 4 + skipped;
        auto make = [&](void** d, size_t bytes, int byte_value) -> hipError_t {
            hipError_t e = hipMalloc(d, std::max<size_t>(bytes, 1));
            if (e == hipSuccess && bytes) e = hipMemset(*d, byte_value, bytes);
            if (e == hipSuccess) s->device_bytes += std::max<size_t>(bytes, 1);
            return e;
        };
        if ((he = make((void**)&s->d_records, (total + 1) * 4 * sizeof(float4), 0)) != hipSuccess) return fail((int)he);
        s->records_bytes = s->records_alloc_bytes = (total + 1) * 4 * sizeof(float4);
        if ((he = make((void**)&s->d_tri_uv, total * 6 * sizeof(float), 0)) != hipSuccess) return fail((int)he);
        if ((he = make((void**)&s->d_tri_id, total * sizeof(int32_t), 0xff)) != hipSuccess) return fail((int)he);
        if ((he = make((void**)&s->d_leaf_count, total * sizeof(int32_t), 0)) != hipSuccess) return fail((int)he);
        for (const Segment& g : segments) {
            if (g.count == 0) continue;
            he = hipMemcpy(s->d_records + g.dev_first * 4, records.data() + g.vec_first * 4, g.count * 4 * sizeof(float4), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(s->d_tri_uv + g.dev_first * 6, tri_uv.data() + g.vec_first * 6, g.count * 6 * sizeof(float), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(s->d_tri_id + g.dev_first, tri_id.data() + g.vec_first, g.count * sizeof(int32_t), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(s->d_leaf_count + g.dev_first, leaf_count.data() + g.vec_first, g.count * sizeof(int32_t), hipMemcpyHostToDevice);
            if (he != hipSuccess) return fail((int)he);
        }
    }
    if ((rc = upload(&s->d_mesh_flags, mesh_flags, s->device_bytes))) return fail(rc);
    for (auto& rf : s->mesh_refit) {
        std::vector<int32_t> sched(rf.sched);
        sched.resize((size_t)std::max(rf.int_cap, 1), 0);            // (capacity for the schedule of any tree over the mesh)
        if ((rc = upload(&rf.d_sched, sched, s->device_bytes))) return fail(rc);
    }
    for (int mi : build_on_device) {                             // trees the device builds in place (before the instances take their root entries)
        const RtMeshDesc& m = desc->meshes[mi];
        const size_t n = (size_t)m.num_triangles;
        float* d = nullptr;
        he = hipMalloc((void**)&d, std::max<size_t>(n, 1) * 18 * sizeof(float));
        if (he != hipSuccess) return fail((int)he);
        if (n) {
            he = hipMemcpy(d, m.vertices, n * 9 * sizeof(float), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(d + n * 9, m.normals, n * 3 * sizeof(float), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(d + n * 12, m.uvs, n * 6 * sizeof(float), hipMemcpyHostToDevice);
        }
        rc = he != hipSuccess ? (int)he : rt_scene_rebuild_mesh_device(s, mi, d, d + n * 9, d + n * 12, (int32_t)n, nullptr);
        (void)hipFree(d);
        if (rc) return fail(rc);
    }
    std::vector<DevMaterial> mats((size_t)desc->num_materials);
    for (int i = 0; i < desc->num_materials; i++) {
        const RtMaterialDesc& m = desc->materials[i];
        DevMaterial& d = mats[i];
        memset(&d, 0, sizeof d);
        d.albedo[0] = m.albedo[0]; d.albedo[1] = m.albedo[1]; d.albedo[2] = m.albedo[2];
        d.roughness = m.roughness; d.metallic = m.metallic;
        if (m.texture && m.texture_width > 0) {
            // tight device copy (the reference uses cudaMallocPitch + GpuMat::upload, Material.hpp:35-41)
            const size_t row = (size_t)m.texture_width * 3, bytes = row * (size_t)m.texture_height;
            uint8_t* dt = nullptr;
            he = hipMalloc((void**)&dt, bytes);
            if (he != hipSuccess) return fail((int)he);
            s->d_textures.push_back(dt);
            he = hipMemcpy2D(dt, row, m.texture, m.texture_pitch, row, (size_t)m.texture_height, hipMemcpyHostToDevice);
            if (he != hipSuccess) return fail((int)he);
            s->device_bytes += bytes;
            d.texture = dt; d.texture_width = m.texture_width; d.texture_height = m.texture_height; d.texture_pitch = (uint32_t)row;
        }
    }
    if ((rc = upload(&s->d_materials, mats, s->device_bytes))) return fail(rc);
    for (int i = 0; i < desc->num_instances; i++) s->instances.push_back(make_dev_instance(desc->instances[i], *s));
    if ((rc = upload(&s->d_instances, s->instances, s->device_bytes))) return fail(rc);
    *out = s;
    return RT_OK;
}

#define RT_SCENE_CALL(s) if (!(s)) return RT_E_INVALID; std::lock_guard<std::recursive_mutex> scene_call_guard_((s)->call_mu)

int rt_scene_update_instance(RtScene* s, int32_t index, const RtInstanceDesc* instance)
{
    RT_SCENE_CALL(s);
    if (!s || !instance || index < 0 || index >= (int)s->instances.size() || !instance_ok(*instance, *s)) return RT_E_INVALID;
    s->instances[index] = make_dev_instance(*instance, *s);
    RT_HIP(hipMemcpy(s->d_instances + index, &s->instances[index], sizeof(DevInstance), hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_scene_update_instance_async(RtScene* s, int32_t index, const RtInstanceDesc* instance, void* stream)
{
    RT_SCENE_CALL(s);
    if (!s || !instance || index < 0 || index >= (int)s->instances.size() || !instance_ok(*instance, *s)) return RT_E_INVALID;
    s->instances[index] = make_dev_instance(*instance, *s);
    hipLaunchKernelGGL(set_instance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, s->d_instances + index, s->instances[index]);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

namespace {
int refit_mesh(RtScene* s, int32_t mesh_index, const float* vertices, const float* normals, int32_t num_triangles, void* stream, bool on_device)
{
    RT_SCENE_CALL(s);
    if (!vertices || !normals || mesh_index < 0 || mesh_index >= (int)s->mesh_refit.size()) return RT_E_INVALID;
    const RtScene::MeshRefit& rf = s->mesh_refit[(size_t)mesh_index];
    if (num_triangles != rf.num_triangles) return RT_E_INVALID;
    if (rf.num_triangles == 0) return RT_OK;
    hipStream_t st = (hipStream_t)stream;
    const float* d_v = vertices;
    const float* d_n = normals;
    if (!on_device) {
        const size_t nv = (size_t)rf.num_triangles * 9, nn = (size_t)rf.num_triangles * 3, need = (nv + nn) * sizeof(float);
        if (s->refit_scratch_bytes < need) {
            (void)hipFree(s->d_refit_scratch);                  // (synchronises with a refit still in flight)
            s->d_refit_scratch = nullptr; s->refit_scratch_bytes = 0;
            RT_HIP(hipMalloc((void**)&s->d_refit_scratch, need));
            s->refit_scratch_bytes = need;
        }
        // (the host arrays may be pageable: the copies return once the bytes are staged, and stay ordered on the stream)
        RT_HIP(hipMemcpyAsync(s->d_refit_scratch, vertices, nv * sizeof(float), hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(s->d_refit_scratch + nv, normals, nn * sizeof(float), hipMemcpyHostToDevice, st));
        d_v = s->d_refit_scratch;
        d_n = s->d_refit_scratch + nv;
    }
    hipLaunchKernelGGL(refit_triangles_kernel, dim3((unsigned)((rf.num_slots + 255) / 256)), dim3(256), 0, st, s->d_records, s->d_tri_id,
                       rf.slot_base, rf.num_slots, d_v, d_n);
    // deepest level first: children before parents.  A wide level is a launch of its own; consecutive levels of at most
    // kRefitRunThreads nodes share one single-workgroup launch (a 28-level tree: 10 launches instead of 28 -- with the arrays
    // already on the device the call is launch-bound)
    int begin = 0;
    int32_t* mesh_flag = s->d_mesh_flags + mesh_index;
    // every interior record of the mesh is rewritten below, so the flag is decided anew: one frame with a NaN vertex in a deforming
    // mesh no longer keeps the mesh on the generic slab loop (4-7 % slower on c2) until its next rebuild
    RT_HIP(hipMemsetAsync(mesh_flag, 0, sizeof(int32_t), st));
    RefitRun run;
    run.levels = 0;
    auto flush = [&] {
        if (run.levels == 0) return;
        if (run.levels == 1)
            hipLaunchKernelGGL(refit_level_kernel, dim3((unsigned)((run.bound[1] - run.bound[0] + 255) / 256)), dim3(256), 0, st, s->d_records, s->d_tri_id,
                               s->d_leaf_count, d_v, rf.d_sched, run.bound[0], run.bound[1], mesh_flag);
        else
            hipLaunchKernelGGL(refit_run_kernel, dim3(1), dim3(kRefitRunThreads), 0, st, s->d_records, s->d_tri_id, s->d_leaf_count, d_v, rf.d_sched, run, mesh_flag);
        run.levels = 0;
    };
    for (int32_t end : rf.level_end) {
        if (end - begin > kRefitRunThreads) {
            flush();
            hipLaunchKernelGGL(refit_level_kernel, dim3((unsigned)((end - begin + 255) / 256)), dim3(256), 0, st, s->d_records, s->d_tri_id,
                               s->d_leaf_count, d_v, rf.d_sched, begin, end, mesh_flag);
        } else {
            if (run.levels == kRefitRunLevels) flush();
            if (run.levels == 0) run.bound[0] = begin;
            run.bound[++run.levels] = end;
        }
        begin = end;
    }
    flush();
    RT_HIP(hipGetLastError());
    return RT_OK;
}
}  // namespace

int rt_scene_refit_mesh(RtScene* s, int32_t mesh_index, const float* vertices, const float* normals, int32_t num_triangles, void* stream)
{
    return refit_mesh(s, mesh_index, vertices, normals, num_triangles, stream, false);
}

int rt_scene_refit_mesh_device(RtScene* s, int32_t mesh_index, const float* d_vertices, const float* d_normals, int32_t num_triangles, void* stream)
{
    return refit_mesh(s, mesh_index, d_vertices, d_normals, num_triangles, stream, true);
}

int rt_scene_debug_read(RtScene* s, int32_t which, void* host_dst, size_t capacity, size_t* bytes)
{
    RT_SCENE_CALL(s);
    if (!bytes || which < 0 || which > 4) return RT_E_INVALID;
    size_t recs = 4;                                            // the padding records at the end
    for (const auto& rf : s->mesh_refit) recs = std::max(recs, (size_t)rf.slot_base + (size_t)rf.slot_cap + 4);
    recs -= 4;
    const void* src = nullptr;
    size_t n = 0;
    switch (which) {
    case 0: src = s->d_records; n = (recs + 1) * 4 * sizeof(float4); break;       // (+ the four padding float4 = one record)
    case 1: src = s->d_tri_uv; n = recs * 6 * sizeof(float); break;
    case 2: src = s->d_tri_id; n = recs * sizeof(int32_t); break;
    case 3: src = s->d_leaf_count; n = recs * sizeof(int32_t); break;
    default: src = s->d_instances; n = s->instances.size() * sizeof(DevInstance); break;
    }
    *bytes = n;
    if (!host_dst || capacity < n) return RT_OK;
    RT_HIP(hipDeviceSynchronize());
    if (n) RT_HIP(hipMemcpy(host_dst, src, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_destroy(RtScene* s)
{
    if (!s) return RT_OK;
    for (int k = 0; k < 2; k++) {
        if (s->overlap.stream[k]) { (void)hipStreamSynchronize(s->overlap.stream[k]); (void)hipStreamDestroy(s->overlap.stream[k]); }
        if (s->overlap.done[k]) (void)hipEventDestroy(s->overlap.done[k]);
    }
    if (s->order.sort_stream) (void)hipStreamSynchronize(s->order.sort_stream);
    for (auto& o : s->order.entry) {
        if (o.sort_done) { (void)hipEventDestroy(o.sort_done); for (auto& e : o.seen) (void)hipEventDestroy(e.done); }
        (void)hipFree(o.d_cost);
    }
    if (s->order.sort_stream) (void)hipStreamDestroy(s->order.sort_stream);
    for (auto& rf : s->mesh_refit) (void)hipFree(rf.d_sched);
    (void)hipFree(s->d_refit_scratch);
    (void)hipFree(s->d_ex_scratch);
    for (auto& sl : s->view.slot) if (sl.done) (void)hipEventDestroy(sl.done);
    (void)hipFree(s->d_records); (void)hipFree(s->d_tri_uv); (void)hipFree(s->d_tri_id);
    (void)hipFree(s->d_leaf_count); (void)hipFree(s->d_mesh_flags); (void)hipFree(s->d_instances); (void)hipFree(s->d_materials);
    for (uint8_t* t : s->d_textures) (void)hipFree(t);
    delete s;
    return RT_OK;
}

int rt_scene_info(const RtScene* s, size_t* device_bytes, int32_t* max_stack)
{
    if (!s) return RT_E_INVALID;
    if (device_bytes) *device_bytes = s->device_bytes;
    if (max_stack) *max_stack = s->max_stack;
    return RT_OK;
}

int rt_scene_view_stats(RtScene* s, uint64_t* launches, uint64_t* fallbacks, uint64_t* grows, int32_t* slot_frames)
{
    RT_SCENE_CALL(s);
    std::lock_guard<std::mutex> lock(s->view.m);
    if (launches) *launches = s->view.launches;
    if (fallbacks) *fallbacks = s->view.fallbacks;
    if (grows) *grows = s->view.grows;
    if (slot_frames) *slot_frames = s->view.slot_frames;
    return RT_OK;
}

int rt_scene_reserve_views(RtScene* s, int32_t frames_per_launch)
{
    RT_SCENE_CALL(s);
    if (frames_per_launch < 0 || frames_per_launch > kMaxBatch) return RT_E_INVALID;
    RtScene::ViewPool& v = s->view;
    view_decide(s);
    if (!view_mode() || !v.usable) return RT_E_INVALID;         // (more than eight instances, or no interior records: this scene renders without views)
    if (frames_per_launch == 0) { v.reserved = false; return RT_OK; }      // back to growing on demand (what is there stays)
    if (frames_per_launch != v.slot_frames) {
        const int rc = view_resize(s, frames_per_launch);
        if (rc) return rc;
    }
    v.reserved = true;
    return RT_OK;
}

int rt_scene_memory(RtScene* s, size_t* records_bytes, size_t* view_pool_bytes, size_t* device_bytes, int32_t* view_slots, int32_t* view_slot_frames)
{
    RT_SCENE_CALL(s);
    if (records_bytes) *records_bytes = s->records_bytes;
    if (view_pool_bytes) *view_pool_bytes = s->view.slot_frames > 0 ? (size_t)s->view.frame_records * 64 * (size_t)s->view.slots * (size_t)s->view.slot_frames : 0;
    if (device_bytes) *device_bytes = s->device_bytes;
    if (view_slots) *view_slots = s->view.slots;
    if (view_slot_frames) *view_slot_frames = s->view.slot_frames;
    return RT_OK;
}

int rt_scene_mesh_flags(RtScene* s, int32_t mesh_index, int32_t* flags)
{
    if (!s || !flags || mesh_index < 0 || mesh_index >= (int)s->mesh_refit.size()) return RT_E_INVALID;
    RT_HIP(hipMemcpy(flags, s->d_mesh_flags + mesh_index, sizeof(int32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_mesh_capacity(const RtScene* s, int32_t mesh_index, int32_t* max_triangles)
{
    if (!s || !max_triangles || mesh_index < 0 || mesh_index >= (int)s->mesh_refit.size()) return RT_E_INVALID;
    const RtScene::MeshRefit& rf = s->mesh_refit[(size_t)mesh_index];
    *max_triangles = std::min(rf.slot_cap, rf.int_cap + 1);
    return RT_OK;
}

// (every render entry point: the scene's call_mu is held while the launch is prepared and queued; a wait the caller asked for
// happens after it has been released)
#define RT_WAIT_IF(synchronize, stream) do { if (synchronize) RT_HIP(hipStreamSynchronize((hipStream_t)(stream))); } while (0)

int rt_render_batch(RtScene* s, const RtCameraParams* cams, uint8_t* const* d_imgs, size_t pitch, int32_t count,
                    void* stream, int synchronize)
{
    {
        RT_SCENE_CALL(s);
        RenderParams p;
        int rc = fill_params(p, s, cams, d_imgs, count, pitch);
        if (rc) return rc;
        if ((rc = launch(p, false, (hipStream_t)stream, 0, s))) return rc;
    }
    RT_WAIT_IF(synchronize, stream);
    return RT_OK;
}

int rt_scene_loop_stats(RtScene* s, const RtCameraParams* cams, uint8_t* const* d_imgs, size_t pitch, int32_t count, void* stream, uint64_t* stats)
{
    if (!stats) return RT_E_INVALID;
    RT_SCENE_CALL(s);
    RenderParams p;
    int rc = fill_params(p, s, cams, d_imgs, count, pitch);
    if (rc) return rc;
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    unsigned long long* d = nullptr;
    RT_HIP(hipMalloc((void**)&d, RT_LOOP_WORDS * sizeof(uint64_t)));
    hipError_t e = hipMemsetAsync(d, 0, RT_LOOP_WORDS * sizeof(uint64_t), (hipStream_t)stream);
    p.loop_stats = d;
    if (e == hipSuccess) rc = launch(p, false, (hipStream_t)stream, 1, s, true);
    if (e == hipSuccess && rc == RT_OK) e = hipMemcpy(stats, d, RT_LOOP_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e != hipSuccess ? (int)e : rc;
}

int rt_render(RtScene* s, const RtCameraParams* cam, uint8_t* d_img, size_t pitch, void* stream, int synchronize)
{
    return rt_render_batch(s, cam, &d_img, pitch, 1, stream, synchronize);
}

// Camera::render_scene(scene, img, pitch, synchronize = false) on the default stream (Camera.cu:18-41) -- see rt_hip.h.
int rt_render_overlapped(RtScene* s, const RtCameraParams* cam, uint8_t* d_img, size_t pitch)
{
    RT_SCENE_CALL(s);
    RenderParams p;
    int rc = fill_params(p, s, cam, &d_img, 1, pitch);
    if (rc) return rc;
    static const bool enabled = [] { const char* e = getenv("RT_RENDER_OVERLAP"); return !(e && e[0] == '0'); }();
    if (!enabled) return launch(p, false, nullptr, 0, s);
    RtScene::Overlap& ov = s->overlap;
    std::lock_guard<std::mutex> lock(ov.m);
    if (!ov.stream[0]) {
        // hipStreamDefault = a BLOCKING stream: work on the null stream waits for everything issued here before it, and
        // everything issued here waits for earlier null-stream work -- the ordering a default-stream launch has with the
        // caller's copies, memsets, instance updates and hipDeviceSynchronize
        // Round 6: stream 0 has the HIGHER priority.  The reference's loop issues two frames and then synchronises the device: two
        // equal streams share the chip, both frames run at half speed and END TOGETHER -- their tails coincide and nothing is hidden
        // (two overlapped frames 245 us, each alone 125: rocprofv3 timeline in profiles/r06_experiments/single_frame_gap.md).  With
        // priorities the first frame of a pair takes the chip, the second fills the slots its tail leaves free and then runs alone:
        // one tail exposed instead of two.  (RT_OVERLAP_PRIORITY=0: two equal streams, for the A/B.)
        static const bool prio = [] { const char* e = getenv("RT_OVERLAP_PRIORITY"); return !(e && e[0] == '0'); }();
        int least = 0, greatest = 0;
        if (!prio || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        hipStream_t st[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipError_t e = hipStreamCreateWithPriority(&st[0], hipStreamDefault, greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&st[1], hipStreamDefault, least);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int k = 0; k < 2; k++) { if (st[k]) (void)hipStreamDestroy(st[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
            return (int)e;
        }
        for (int k = 0; k < 2; k++) { ov.stream[k] = st[k]; ov.done[k] = ev[k]; }
    }
    const RtScene::Overlap::Range r{(uintptr_t)d_img, (uintptr_t)d_img + (size_t)(p.height - 1) * pitch + (size_t)p.width * 3};
    auto meets = [&](const std::vector<RtScene::Overlap::Range>& v) {
        for (const auto& w : v) if (w.lo < r.hi && r.lo < w.hi) return true;
        return false;
    };
    // (the high-priority stream takes the frame whenever it is idle -- after every synchronise, so the first frame of a pair)
    int use = ov.next;
    if (use == 1) { if (hipStreamQuery(ov.stream[0]) == hipSuccess) use = 0; else (void)hipGetLastError(); }
    int other = use ^ 1;
    const bool hit_other = meets(ov.written[other]);
    if (hit_other && !meets(ov.written[use])) {
        std::swap(use, other);                                  // same image as the frame in flight over there: stream order does it
    } else if (hit_other) {
        RT_HIP(hipEventRecord(ov.done[other], ov.stream[other]));          // (recorded when it is needed: after everything issued there so far)
        RT_HIP(hipStreamWaitEvent(ov.stream[use], ov.done[other], 0));
        ov.written[other].clear();                              // everything issued there so far is now ordered before this stream's next work
        ov.waits++;
    }
    rc = launch(p, false, ov.stream[use], 0, s);
    if (rc) return rc;
    std::vector<RtScene::Overlap::Range>& mine = ov.written[use];
    bool known = false;
    for (const auto& w : mine) known = known || (w.lo == r.lo && w.hi == r.hi);
    if (!known) {
        if (mine.size() >= 8) {                                 // an application cycling through many images: one hull (may over-order, never under)
            RtScene::Overlap::Range hull = r;
            for (const auto& w : mine) { hull.lo = std::min(hull.lo, w.lo); hull.hi = std::max(hull.hi, w.hi); }
            mine.assign(1, hull);
        } else {
            mine.push_back(r);
        }
    }
    ov.launches++;
    ov.next = use ^ 1;
    return RT_OK;
}

int rt_render_overlapped_stats(const RtScene* s, uint64_t* launches, uint64_t* cross_stream_waits)
{
    if (!s) return RT_E_INVALID;
    if (launches) *launches = s->overlap.launches;
    if (cross_stream_waits) *cross_stream_waits = s->overlap.waits;
    return RT_OK;
}

int rt_render_ids(RtScene* s, const RtCameraParams* cam, uint8_t* d_img, size_t pitch, int32_t* d_hit_instance,
                  int32_t* d_hit_triangle, void* stream, int synchronize)
{
    {
        RT_SCENE_CALL(s);
        RenderParams p;
        int rc = fill_params(p, s, cam, &d_img, 1, pitch);
        if (rc) return rc;
        p.hit_instance = d_hit_instance; p.hit_triangle = d_hit_triangle;
        if ((rc = launch(p, false, (hipStream_t)stream, 0, s))) return rc;   // (with the scene: the kernel whose ids are checked is the one that is timed, views included)
    }
    RT_WAIT_IF(synchronize, stream);
    return RT_OK;
}

int rt_render_debug(RtScene* s, const RtCameraParams* cam, uint8_t* d_img, size_t pitch,
                    const RtDebugPlanes* planes, void* stream, int synchronize)
{
    if (!planes) return RT_E_INVALID;
    {
        RT_SCENE_CALL(s);
        RenderParams p;
        int rc = fill_params(p, s, cam, &d_img, 1, pitch);
        if (rc) return rc;
        p.hit_instance = planes->hit_instance; p.hit_triangle = planes->hit_triangle; p.node_pops = planes->node_pops;
        p.aabb_tests = planes->aabb_tests; p.tri_tests = planes->tri_tests; p.inside_hits = planes->inside_hits;
        if ((rc = launch(p, true, (hipStream_t)stream, 0))) return rc;
    }
    RT_WAIT_IF(synchronize, stream);
    return RT_OK;
}

// Samples are rendered in chunks of as many sample indices as fit the scratch budget: per chunk either one
// render_ex_kernel launch with grid.y = chunk (samples only, or RT_EX_WAVEFRONT=0), or the wavefront sequence of
// ex_wave_kernel launches; then one resolve_ex_kernel.
constexpr size_t kExScratchBudget = (size_t)2 << 30;           // per-lane form: 16 B per path
constexpr size_t kExSplitBudget = (size_t)5 << 30;             // two-launch bounce form: 32 B per (pixel, sample) of a chunk of workgroups
constexpr size_t kExWaveScratchBudget = (size_t)16 << 30;      // wavefront form: 16 B + (7 + 3 x 5) x 16 B of queue room per path

static int ex_env_int(const char* name, int fallback)
{
    const char* e = getenv(name);
    return e && *e ? atoi(e) : fallback;
}

static int launch_ex(RtScene* s, RenderParams& p, const RtRenderOptions* opts, int32_t* d_total_pops, hipStream_t stream, int synchronize)
{
    if (!opts || opts->spp < 1 || opts->bounces < 0) return RT_E_INVALID;
    if (p.local_rows == 0) return RT_OK;
    p.spp = opts->spp; p.bounces = opts->bounces; p.lighting = opts->lighting ? 1 : 0; p.total_pops = d_total_pops;
    p.tiles_x = (p.width + kTile - 1) / kTile;
    p.tiles_y = (p.local_rows + kTile - 1) / kTile;
    const size_t npix = (size_t)p.local_rows * p.width;
    const size_t ntiles = (size_t)p.tiles_x * p.tiles_y, nquads = ntiles * 4;
    const char* trace_file = getenv("RT_TRACE_FILE");                               // diagnostics only: first chunk, per-lane form
    const bool simple = p.bounces == 0 && !p.lighting;
    // (read per call, not cached: the parity tests switch between the two forms)
    // The wavefront form is opt-in (RT_EX_WAVEFRONT=1): measured 37-52 ms against 32.8 ms on the blob with sun + 8 bounces, see
    // the comment above ex_wave_kernel and profiles/r03_experiments/ex_wavefront.md
    const bool wavefront = !simple && !trace_file && ex_env_int("RT_EX_WAVEFRONT", 0) != 0;
    // The samples of a pixel in one wave (render_ex_kernel<.., PX>): the default from four samples per pixel on.
    // RT_EX_PIXEL_WAVES=0 selects the older mapping (one sample index per launch row, sample planes + resolve pass).
    const bool pixel_waves = !wavefront && !trace_file && p.spp >= 4 && ex_env_int("RT_EX_PIXEL_WAVES", 1) != 0;
    if (pixel_waves) {
        const bool many = p.spp > 64;                                               // several launches: running sums in ex_acc
        // bounces / lighting, RT_EX_SPLIT=1: the camera ray in a launch of its own (render_ex_kernel<.., PHASE>; measured slower, see there),
        // in chunks of workgroups whose records -- 2 KB per one-wave workgroup -- fit the scratch budget (c3: all 2 073 600 waves at once, 4.2 GB)
        const bool split = !simple && ex_env_int("RT_EX_SPLIT", 0) != 0;
        const size_t per_wg = (size_t)kExBlock * 2 * sizeof(float4);
        size_t split_chunk = 0;
        if (split) {
            size_t budget = kExSplitBudget;
            if (const char* e = getenv("RT_EX_SPLIT_BYTES")) budget = (size_t)strtoull(e, nullptr, 10);           // tests: force several chunks
            int slots0 = 4;
            while (slots0 < std::min(64, p.spp)) slots0 <<= 1;
            const int ppw0 = 64 / slots0, pw0 = ppw0 >= 16 ? 4 : (ppw0 >= 4 ? 2 : (ppw0 >= 2 ? 2 : 1)), ph0 = ppw0 / pw0;
            const size_t wgs0 = (size_t)((p.width + 2 * pw0 - 1) / (2 * pw0)) * (size_t)((p.local_rows + 2 * ph0 - 1) / (2 * ph0)) * 4;   // the first (largest) launch
            split_chunk = std::max<size_t>(4, std::min(wgs0, budget / per_wg) & ~(size_t)3);
        }
        const size_t need = (many ? npix * sizeof(float4) : 0) + split_chunk * per_wg;
        if (s->ex_scratch_bytes < need) {
            (void)hipFree(s->d_ex_scratch);                                         // (synchronises with renders in flight)
            s->d_ex_scratch = nullptr; s->ex_scratch_bytes = 0;
            RT_HIP(hipMalloc((void**)&s->d_ex_scratch, need));
            s->ex_scratch_bytes = need;
        }
        p.ex_acc = s->d_ex_scratch;
        // (four LDS rows per lane hold a wave's samples for the in-wave sum, whatever the depth of the stack)
        const size_t lds = (size_t)std::max(lds_rows(p.stack_depth) + 1, 4) * kExBlock * sizeof(int);       // (+ the optimistic stack's spare row)
        // the samples-only kernel's rays all start at the camera: one view of the tree serves every sample of the frame
        const int view_slot = (simple || RT_EX_PRIMARY_ASM) ? view_prepare(s, p, stream, p.spp) : -1;
        struct ViewDone {
            RtScene* s; int slot; hipStream_t stream;
            ~ViewDone() { view_done(s, slot, stream); }
        } view_guard{s, view_slot, stream};
        for (int base = 0; base < p.spp; base += 64) {
            const int n = std::min(64, p.spp - base);
            int slots = 4;
            while (slots < n) slots <<= 1;
            const int ppw = 64 / slots;                                             // pixels of a wave: 16, 8, 4, 2 or 1
            p.px_n = slots; p.px_count = n;
            p.px_pw = ppw >= 16 ? 4 : (ppw >= 4 ? 2 : (ppw >= 2 ? 2 : 1));
            p.px_ph = ppw / p.px_pw;
            p.px_first = base == 0 ? 1 : 0; p.px_last = base + n == p.spp ? 1 : 0;
            p.sample_base = base;
            p.tiles_x = (p.width + 2 * p.px_pw - 1) / (2 * p.px_pw);
            p.tiles_y = (p.local_rows + 2 * p.px_ph - 1) / (2 * p.px_ph);
            const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * 4));         // four one-wave workgroups per tile
            if (split) {
                const size_t chunk = split_chunk;
                p.ex_rec = s->d_ex_scratch + (many ? npix : 0);
                p.ex_rec_lanes = (uint32_t)(chunk * kExBlock);
                for (size_t wg0 = 0; wg0 < grid.x; wg0 += chunk) {
                    const dim3 part((unsigned)std::min(chunk, (size_t)grid.x - wg0));
                    p.wg_base = (int32_t)wg0;
                    if (view_slot >= 0) hipLaunchKernelGGL((render_ex_kernel<false, true, true, 1>), part, dim3(kExBlock), lds, stream, p);
                    else hipLaunchKernelGGL((render_ex_kernel<false, true, false, 1>), part, dim3(kExBlock), lds, stream, p);
                    hipLaunchKernelGGL((render_ex_kernel<false, true, false, 2>), part, dim3(kExBlock), lds, stream, p);
                }
                p.wg_base = 0;
            }
            else if (simple && view_slot >= 0) hipLaunchKernelGGL((render_ex_kernel<true, true, true>), grid, dim3(kExBlock), lds, stream, p);
            else if (simple) hipLaunchKernelGGL((render_ex_kernel<true, true>), grid, dim3(kExBlock), lds, stream, p);
            else if (view_slot >= 0) hipLaunchKernelGGL((render_ex_kernel<false, true, true>), grid, dim3(kExBlock), lds, stream, p);
            else hipLaunchKernelGGL((render_ex_kernel<false, true>), grid, dim3(kExBlock), lds, stream, p);
            RT_HIP(hipGetLastError());
        }
        if (synchronize) RT_HIP(hipStreamSynchronize(stream));
        return RT_OK;
    }
    size_t budget = wavefront ? kExWaveScratchBudget : kExScratchBudget;
    if (const char* e = getenv("RT_EX_SCRATCH_BYTES")) budget = (size_t)strtoull(e, nullptr, 10);   // tests: force several chunks
    // bytes per sample index: its plane of samples, and for the wavefront form a slot for every path of the plane (tiles
    // padded to whole quads) in the S queue and in the three A arrays (worst case: no path ends)
    constexpr size_t kSlotBytes = (kExPlanesS + 3 * kExPlanesA) * sizeof(float4) + 4 * sizeof(int32_t) / 64 + 1;
    const size_t per_sample = npix * sizeof(float4) + (wavefront ? nquads * 64 * kSlotBytes : 0);
    // (a chunk's path ids and slot numbers are 32-bit)
    int chunk = (int)std::max<size_t>(1, std::min<size_t>({(size_t)p.spp, budget / per_sample, (size_t)65535, (((size_t)1 << 31) - 1) / (nquads * 64)}));
    if (wavefront && chunk >= 4) chunk &= ~3;                   // the primary launch takes sample indices four at a time
    {                                                           // chunks of equal size
        const int nchunks = (p.spp + chunk - 1) / chunk;
        int even = (p.spp + nchunks - 1) / nchunks;
        if (wavefront && even >= 4) even = (even + 3) & ~3;
        chunk = std::min(chunk, even);
    }
    // the primary launch's workgroup: 4 / 2 / 1 sample indices of 1 / 2 / 4 quads; segments = its waves
    auto gen_shape = [&](int n, int& spw, size_t& groups) { spw = n >= 4 ? 4 : (n >= 2 ? 2 : 1); groups = nquads / (size_t)(4 / spw) * (size_t)((n + spw - 1) / spw); };
    int spw_full; size_t gen_full;
    gen_shape(chunk, spw_full, gen_full);
    // the queue geometry of a full chunk (the last chunk of a frame may use fewer segments of the same arrays)
    const size_t max_seg = gen_full * 4, cap = wavefront ? max_seg * 64 : 0;
    const size_t need = (size_t)(chunk + 1) * npix * sizeof(float4) + cap * (kExPlanesS + 3 * kExPlanesA) * sizeof(float4) + 4 * max_seg * sizeof(int32_t);
    if (s->ex_scratch_bytes < need) {
        (void)hipFree(s->d_ex_scratch);                                             // (synchronises with renders in flight)
        s->d_ex_scratch = nullptr; s->ex_scratch_bytes = 0;
        RT_HIP(hipMalloc((void**)&s->d_ex_scratch, need));
        s->ex_scratch_bytes = need;
    }
    p.ex_acc = s->d_ex_scratch;
    p.ex_samples = s->d_ex_scratch + npix;
    const size_t lds = (size_t)(lds_rows(p.stack_depth) + 1) * kBlock * sizeof(int);                // (+ the optimistic stack's spare row)
    for (int base = 0; base < p.spp; base += chunk) {
        const int n = std::min(chunk, p.spp - base);
        p.sample_base = base;
        if (wavefront) {
            int spw; size_t gen_groups;
            gen_shape(n, spw, gen_groups);
            const size_t nseg = gen_groups * 4, slots = nseg * 64;
            if (nseg > max_seg) return RT_E_INVALID;            // (cannot happen: n <= chunk)
            p.gen_spw = spw;
            float4* q = p.ex_samples + (size_t)chunk * npix;
            p.exq_s = q; q += slots * kExPlanesS;
            for (int k = 0; k < 3; k++) { p.exq_a[k] = q; q += slots * kExPlanesA; }
            int32_t* c = (int32_t*)(p.ex_samples + (size_t)chunk * npix + cap * (kExPlanesS + 3 * kExPlanesA));
            p.exq_cnt_s = c; c += nseg;
            for (int k = 0; k < 3; k++) { p.exq_cnt_a[k] = c; c += nseg; }
            p.exq_nseg = (int32_t)nseg;
            p.gen_samples = n;
            // segments per group: 32 (one quad x 32 samples) unless that leaves the chip short of workgroups
            int K = std::min(kExMaxGroup, std::max(1, ex_env_int("RT_EX_GROUP", kExMaxGroup)));
            while (K > 4 && nseg / (size_t)K < 4096) K >>= 1;
            p.exq_k = K;
            const dim3 qgrid((unsigned)((nseg + K - 1) / K));
            // A(d) = segments of a[x] (from the cast launch of d - 1) then of a[y] (from its shadow launch); a[z] is free
            int ax = 0, ay = 1, az = 2;
            p.depth = 0; p.exq_in1 = p.exq_in2 = -1; p.exq_out = ax;
            hipLaunchKernelGGL(ex_wave_kernel<kExGen>, dim3((unsigned)gen_groups), dim3(kBlock), lds, stream, p);
            RT_HIP(hipGetLastError());
            if (p.lighting) {
                p.exq_out = ay;
                hipLaunchKernelGGL(ex_wave_kernel<kExShadow>, qgrid, dim3(kBlock), lds, stream, p);
            }
            for (int d = 1; d <= p.bounces; d++) {
                p.depth = d;
                p.exq_in1 = ax; p.exq_in2 = p.lighting ? ay : -1; p.exq_out = az;
                hipLaunchKernelGGL(ex_wave_kernel<kExBounce>, qgrid, dim3(kBlock), lds, stream, p);
                if (p.lighting) {
                    p.exq_out = ax;                             // A(d) has been read: its first array takes the shadow launch's pushes
                    hipLaunchKernelGGL(ex_wave_kernel<kExShadow>, qgrid, dim3(kBlock), lds, stream, p);
                }
                const int nx = az, ny = ax, nz = ay;
                ax = nx; ay = ny; az = nz;
            }
            RT_HIP(hipGetLastError());
        } else {
            const dim3 grid((unsigned)ntiles * 4, (unsigned)n);                    // four one-wave workgroups per tile
            const size_t trace_n = (size_t)grid.x * grid.y * 16;
            const bool tracing = trace_file && base == 0;
            if (tracing) RT_HIP(trace_begin(p, trace_n));
            const size_t lds_wave = (size_t)(lds_rows(p.stack_depth) + 1) * kExBlock * sizeof(int);
            if (simple) hipLaunchKernelGGL(render_ex_kernel<true>, grid, dim3(kExBlock), lds_wave, stream, p);
            else hipLaunchKernelGGL(render_ex_kernel<false>, grid, dim3(kExBlock), lds_wave, stream, p);
            RT_HIP(hipGetLastError());
            if (tracing) RT_HIP(trace_end(p, trace_n, trace_file, stream));
        }
        hipLaunchKernelGGL(resolve_ex_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, p, n, base == 0 ? 1 : 0,
                           base + n == p.spp ? 1 : 0);
        RT_HIP(hipGetLastError());
    }
    if (synchronize) RT_HIP(hipStreamSynchronize(stream));
    return RT_OK;
}

int rt_render_ex(RtScene* s, const RtCameraParams* cam, const RtRenderOptions* opts, uint8_t* d_img, size_t pitch,
                 int32_t* d_total_pops, void* stream, int synchronize)
{
    {
        RT_SCENE_CALL(s);
        RenderParams p;
        int rc = fill_params(p, s, cam, &d_img, 1, pitch);
        if (rc) return rc;
        if ((rc = launch_ex(s, p, opts, d_total_pops, (hipStream_t)stream, 0))) return rc;
    }
    RT_WAIT_IF(synchronize, stream);
    return RT_OK;
}

int rt_render_ex_stripes(RtScene* s, const RtCameraParams* cam, const RtRenderOptions* opts, uint8_t* d_local, size_t local_pitch,
                         int32_t stripe_rows, int32_t rank, int32_t num_ranks, void* stream, int synchronize)
{
    {
        RT_SCENE_CALL(s);
        RenderParams p;
        int rc = fill_params(p, s, cam, &d_local, 1, local_pitch);
        if (rc) return rc;
        int32_t rows = 0;
        if ((rc = rt_stripe_rows(p.height, stripe_rows, rank, num_ranks, &rows))) return rc;
        p.local_rows = rows; p.stripe_rows = stripe_rows; p.rank = rank; p.num_ranks = num_ranks;
        p.frames[0].rank = rank; p.frames[0].local_rows = rows;
        if ((rc = launch_ex(s, p, opts, nullptr, (hipStream_t)stream, 0))) return rc;
    }
    RT_WAIT_IF(synchronize, stream);
    return RT_OK;
}

int rt_stripe_rows(int32_t height, int32_t stripe_rows, int32_t rank, int32_t num_ranks, int32_t* rows)
{
    if (!rows || height < 0 || stripe_rows <= 0 || num_ranks <= 0 || rank < 0 || rank >= num_ranks) return RT_E_INVALID;
    int32_t n = 0;
    for (int32_t s0 = rank; (int64_t)s0 * stripe_rows < height; s0 += num_ranks)
        n += std::min(stripe_rows, height - s0 * stripe_rows);
    *rows = n;
    return RT_OK;
}

// frame i of the launch renders the stripes of owner (rank + first_frame + i) % num_ranks when first_frame >= 0, of `rank` otherwise
static int render_stripes_batch(RtScene* s, const RtCameraParams* cams, uint8_t* const* d_locals, size_t local_pitch, int32_t count,
                                int32_t stripe_rows, int32_t rank, int32_t num_ranks, int32_t first_frame, void* stream, int synchronize)
{
    {
    RT_SCENE_CALL(s);
    RenderParams p;
    int rc = fill_params(p, s, cams, d_locals, count, local_pitch);
    if (rc) return rc;
    int32_t rows = 0;
    if ((rc = rt_stripe_rows(p.height, stripe_rows, rank, num_ranks, &rows))) return rc;
    p.stripe_rows = stripe_rows; p.rank = rank; p.num_ranks = num_ranks;
    int32_t most = 0;
    for (int i = 0; i < count; i++) {
        FrameParams& f = p.frames[i];
        f.rank = first_frame >= 0 ? (int32_t)(((int64_t)rank + first_frame + i) % num_ranks) : rank;
        if (f.rank != rank) { if ((rc = rt_stripe_rows(p.height, stripe_rows, f.rank, num_ranks, &f.local_rows))) return rc; }
        else f.local_rows = rows;
        most = std::max(most, f.local_rows);
    }
    p.local_rows = most;                                        // the grid covers the tallest frame; shorter ones leave their last tiles empty
    if ((rc = launch(p, false, (hipStream_t)stream, 0, s))) return rc;
    }
    RT_WAIT_IF(synchronize, stream);
    return RT_OK;
}

int rt_render_stripes_batch(RtScene* s, const RtCameraParams* cams, uint8_t* const* d_locals, size_t local_pitch, int32_t count,
                            int32_t stripe_rows, int32_t rank, int32_t num_ranks, void* stream, int synchronize)
{
    return render_stripes_batch(s, cams, d_locals, local_pitch, count, stripe_rows, rank, num_ranks, -1, stream, synchronize);
}

int rt_render_stripes_batch_rotating(RtScene* s, const RtCameraParams* cams, uint8_t* const* d_locals, size_t local_pitch, int32_t count,
                                     int32_t stripe_rows, int32_t rank, int32_t num_ranks, int32_t first_frame, void* stream, int synchronize)
{
    if (first_frame < 0) return RT_E_INVALID;
    return render_stripes_batch(s, cams, d_locals, local_pitch, count, stripe_rows, rank, num_ranks, first_frame, stream, synchronize);
}

int rt_render_stripes(RtScene* s, const RtCameraParams* cam, uint8_t* d_local, size_t local_pitch,
                      int32_t stripe_rows, int32_t rank, int32_t num_ranks, void* stream, int synchronize)
{
    return rt_render_stripes_batch(s, cam, &d_local, local_pitch, 1, stripe_rows, rank, num_ranks, stream, synchronize);
}

static int unstripe_batch(const uint8_t* d_gathered, size_t local_pitch, size_t rank_stride, size_t src_frame_stride,
                          uint8_t* d_imgs, size_t pitch, size_t dst_frame_stride, int32_t count,
                          int32_t width, int32_t height, int32_t stripe_rows, int32_t num_ranks, int32_t rotate_first, void* stream)
{
    if (!d_gathered || !d_imgs || count < 1 || width <= 0 || height <= 0 || stripe_rows <= 0 || num_ranks <= 0 ||
        local_pitch < (size_t)width * 3 || pitch < (size_t)width * 3) return RT_E_INVALID;
    const size_t row_bytes = (size_t)width * 3;
    const bool vec = ((uintptr_t)d_gathered | (uintptr_t)d_imgs | local_pitch | rank_stride | src_frame_stride | pitch | dst_frame_stride | row_bytes) % 16 == 0;
    const size_t row_units = vec ? row_bytes / 16 : row_bytes;
    dim3 grid((unsigned)((row_units * (size_t)height + 255) / 256), 1, (unsigned)count), block(256);
    if (vec)
        hipLaunchKernelGGL(unstripe_kernel<uint4>, grid, block, 0, (hipStream_t)stream, d_gathered, local_pitch, rank_stride, src_frame_stride,
                           d_imgs, pitch, dst_frame_stride, (int)row_units, height, stripe_rows, num_ranks, rotate_first);
    else
        hipLaunchKernelGGL(unstripe_kernel<uint8_t>, grid, block, 0, (hipStream_t)stream, d_gathered, local_pitch, rank_stride, src_frame_stride,
                           d_imgs, pitch, dst_frame_stride, (int)row_units, height, stripe_rows, num_ranks, rotate_first);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_unstripe_batch(const uint8_t* d_gathered, size_t local_pitch, size_t rank_stride, size_t src_frame_stride,
                      uint8_t* d_imgs, size_t pitch, size_t dst_frame_stride, int32_t count,
                      int32_t width, int32_t height, int32_t stripe_rows, int32_t num_ranks, void* stream)
{
    return unstripe_batch(d_gathered, local_pitch, rank_stride, src_frame_stride, d_imgs, pitch, dst_frame_stride, count, width, height,
                          stripe_rows, num_ranks, -1, stream);
}

int rt_unstripe_batch_rotating(const uint8_t* d_gathered, size_t local_pitch, size_t rank_stride, size_t src_frame_stride,
                               uint8_t* d_imgs, size_t pitch, size_t dst_frame_stride, int32_t count,
                               int32_t width, int32_t height, int32_t stripe_rows, int32_t num_ranks, int32_t first_frame, void* stream)
{
    if (first_frame < 0) return RT_E_INVALID;
    return unstripe_batch(d_gathered, local_pitch, rank_stride, src_frame_stride, d_imgs, pitch, dst_frame_stride, count, width, height,
                          stripe_rows, num_ranks, first_frame, stream);
}

int rt_unstripe(const uint8_t* d_gathered, size_t local_pitch, size_t rank_stride, uint8_t* d_img, size_t pitch,
                int32_t width, int32_t height, int32_t stripe_rows, int32_t num_ranks, void* stream)
{
    return rt_unstripe_batch(d_gathered, local_pitch, rank_stride, 0, d_img, pitch, 0, 1, width, height, stripe_rows, num_ranks, stream);
}

int rt_timer_create(RtTimer** t)
{
    if (!t) return RT_E_INVALID;
    RtTimer* r = new (std::nothrow) RtTimer;
    if (!r) return RT_E_NOMEM;
    hipError_t e = hipEventCreate(&r->start);
    if (e == hipSuccess) e = hipEventCreate(&r->stop);
    if (e != hipSuccess) { delete r; return (int)e; }
    *t = r;
    return RT_OK;
}
int rt_timer_start(RtTimer* t, void* stream) { if (!t) return RT_E_INVALID; RT_HIP(hipEventRecord(t->start, (hipStream_t)stream)); return RT_OK; }
int rt_timer_stop(RtTimer* t, void* stream) { if (!t) return RT_E_INVALID; RT_HIP(hipEventRecord(t->stop, (hipStream_t)stream)); return RT_OK; }
int rt_timer_elapsed_ms(RtTimer* t, float* ms)
{
    if (!t || !ms) return RT_E_INVALID;
    RT_HIP(hipEventSynchronize(t->stop));
    RT_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return RT_OK;
}
int rt_timer_destroy(RtTimer* t)
{
    if (!t) return RT_OK;
    (void)hipEventDestroy(t->start); (void)hipEventDestroy(t->stop);
    delete t;
    return RT_OK;
}

}  // extern "C"
