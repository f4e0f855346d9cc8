/*
 * rt_oracle.c -- CPU ORACLE for the per-pixel raycast hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load it, and
 * only as the checker.  The product (cuda-raytracing_amd/) never links,
 * imports or falls back to anything in this directory.
 *
 * It is a plain-C restatement of the reference's algorithm (AFIDclan/cuda-raytracing
 * @ 2024_10_08, paths relative to /root/reference/CudaRaytracer/), written from the
 * reference's behaviour, one function per reference function, each citing the lines
 * it follows.  Arithmetic is fp32 in the written association order, fp64 where the
 * reference's expressions promote to double; build with -ffp-contract=off.
 *
 * Pinning (see DESIGN.md "Oracle"): the leaf functions are checked bit-for-bit
 * against the reference's own headers compiled by oracle/Makefile into
 * oracle/_ref/ref_probe (utils.hpp, transforms.hpp, TrianglePrimitive.hpp,
 * BVHTree.hpp, MeshPrimitive.cpp, OBJLoader.hpp build against the CUDA headers this
 * image ships); raycast.cu itself needs curand_kernel.h, which the image lacks, so
 * cast_ray/render are restated only and pinned end-to-end by the frame hashes
 * recorded in SURVEY.md section 4 / 8(d) from the reference's own render().
 *
 * Deliberate, documented deviations from undefined behaviour in the reference:
 *   - Q_rsqrt uses int32 (utils.hpp:14-22 uses `long`: 4 bytes on the MSVC target of
 *     the .vcxproj, an out-of-bounds 8-byte read on LP64).  SURVEY.md H1.
 *   - cast_ray returns by value (raycast.cu:21,141 returns a reference to a local). H2.
 *   - uninitialised uv_coords / texture_* fields are zero (TrianglePrimitive.hpp:15-35,
 *     Material.hpp:13-19).  H6.
 *   - float -> uchar stores go through int (x86 cvttss2si then low byte), which is what
 *     g++ emits for raycast.cu:292-294.
 */
#define _POSIX_C_SOURCE 200809L   /* getline */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;
typedef struct { float x, y, z, w; } f4;
typedef struct { float x, y, z, yaw, pitch, roll; } lre_t;   /* transforms.hpp:10-14 */

/* TrianglePrimitive.hpp:8-11 */
typedef struct { f3 v[3]; f3 normal; f2 uv[3]; } tri_t;

/* ------------------------------------------------------------------ L0 math */

/* utils.hpp:12-27, int32 semantics */
float orc_q_rsqrt(float number)
{
    int32_t i;
    float x2, y;
    x2 = number * 0.5F;
    y = number;
    memcpy(&i, &y, 4);
    i = 0x5f3759df - (i >> 1);
    memcpy(&y, &i, 4);
    y = y * (1.5F - (x2 * y * y));
    return y;
}

static f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
static f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* utils.hpp:65 */
static f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* utils.hpp:57 */
static f3 mul3s(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }        /* utils.hpp:73,77 */
static f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }      /* utils.hpp:69 */
static float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      /* utils.hpp:53 */
static f3 cross3(f3 a, f3 b)                                                      /* utils.hpp:49 */
{ return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static float magnitude3(f3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); } /* utils.hpp:29 */
static f3 normalize3(f3 v)                                                        /* utils.hpp:37-47 */
{
    float inv_mag = orc_q_rsqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk3(v.x * inv_mag, v.y * inv_mag, v.z * inv_mag);
}

/* transforms.hpp:148-163 ; the half angle is a double product narrowed to float by sinf/cosf */
static f4 euler2quat(f3 e)
{
    float sy = sinf((float)(e.x * 0.5));
    float cy = cosf((float)(e.x * 0.5));
    float sp = sinf((float)(e.y * 0.5));
    float cp = cosf((float)(e.y * 0.5));
    float sr = sinf((float)(e.z * 0.5));
    float cr = cosf((float)(e.z * 0.5));
    f4 q;
    q.x = sy * sp * sr + cy * cp * cr;
    q.y = cy * sp * cr + sy * cp * sr;
    q.z = -sy * sp * cr + cy * cp * sr;
    q.w = cy * sp * sr - sy * cp * cr;
    return q;
}

/* transforms.hpp:165-176 */
static f3 apply_quat(f4 q, f3 v)
{
    float a = -v.x * q.y - v.y * q.z - v.z * q.w;
    float b = v.x * q.x + v.y * q.w - v.z * q.z;
    float c = v.y * q.x + v.z * q.y - v.x * q.w;
    float d = v.z * q.x + v.x * q.z - v.y * q.y;
    return mk3(q.x * b - q.y * a - q.z * d + q.w * c,
               q.x * c - q.z * a - q.w * b + q.y * d,
               q.x * d - q.w * a - q.y * c + q.z * b);
}

static f3 apply_euler(f3 e, f3 v) { return apply_quat(euler2quat(e), v); }         /* transforms.hpp:219 */
static f3 apply_lre(lre_t l, f3 v)                                                 /* transforms.hpp:223-226 */
{
    f3 s = mk3(v.x - l.x, v.y - l.y, v.z - l.z);
    return apply_euler(mk3(l.yaw, l.pitch, l.roll), s);
}

typedef struct { float m[3][3]; } m33;
typedef struct { float m[4][4]; } m44;

static f3 apply_rotmat(const m33 *r, f3 v)                                         /* transforms.hpp:63-69 */
{
    f3 o;
    o.x = r->m[0][0] * v.x + r->m[0][1] * v.y + r->m[0][2] * v.z;
    o.y = r->m[1][0] * v.x + r->m[1][1] * v.y + r->m[1][2] * v.z;
    o.z = r->m[2][0] * v.x + r->m[2][1] * v.y + r->m[2][2] * v.z;
    return o;
}
static m33 invert_rotmat(const m33 *r)                                             /* transforms.hpp:55-61 */
{
    m33 o; int i, j;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) o.m[i][j] = r->m[j][i];
    return o;
}
static m33 euler2rotmat(f3 e)                                                      /* transforms.hpp:129-144 */
{
    float sy = sinf(e.x), cy = cosf(e.x), sp = sinf(e.y), cp = cosf(e.y), sr = sinf(e.z), cr = cosf(e.z);
    m33 o;
    o.m[0][0] = cr * cy + sr * sp * sy; o.m[0][1] = -cr * sy + sr * sp * cy; o.m[0][2] = -sr * cp;
    o.m[1][0] = cp * sy;                o.m[1][1] = cp * cy;                 o.m[1][2] = sp;
    o.m[2][0] = sr * cy - cr * sp * sy; o.m[2][1] = -sr * sy - cr * sp * cy; o.m[2][2] = cr * cp;
    return o;
}
static f3 rotmat2euler(const m33 *r)                                               /* transforms.hpp:119-126 */
{
    float a = r->m[1][2];
    if (a > 1) a = 1; else if (a < -1) a = -1;
    return mk3(atan2f(r->m[1][0], r->m[1][1]), asinf(a), atan2f(-r->m[0][2], r->m[2][2]));
}
static m44 lre2homo(lre_t v)                                                       /* transforms.hpp:178-193 */
{
    f3 shift = mk3(-v.x, -v.y, -v.z);
    m33 R = euler2rotmat(mk3(v.yaw, v.pitch, v.roll));
    f3 rs = apply_rotmat(&R, shift);
    m44 o; int i, j;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) o.m[i][j] = R.m[i][j];
    o.m[0][3] = rs.x; o.m[1][3] = rs.y; o.m[2][3] = rs.z;
    o.m[3][0] = 0.0f; o.m[3][1] = 0.0f; o.m[3][2] = 0.0f; o.m[3][3] = 1.0f;
    return o;
}
static m44 invert_homo(const m44 *H)                                               /* transforms.hpp:72-96 */
{
    m33 R, Ri; m44 o; int i, j; f3 t, ti;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) R.m[i][j] = H->m[i][j];
    Ri = invert_rotmat(&R);
    t = mk3(-H->m[0][3], -H->m[1][3], -H->m[2][3]);
    ti = apply_rotmat(&Ri, t);
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) o.m[i][j] = Ri.m[i][j];
    o.m[0][3] = ti.x; o.m[1][3] = ti.y; o.m[2][3] = ti.z;
    o.m[3][0] = 0.0f; o.m[3][1] = 0.0f; o.m[3][2] = 0.0f; o.m[3][3] = 1.0f;
    return o;
}
static lre_t homo2lre(const m44 *H)                                                /* transforms.hpp:195-216 */
{
    m33 R, Ri; int i, j; f3 e, s; lre_t o;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) R.m[i][j] = H->m[i][j];
    e = rotmat2euler(&R);
    s = mk3(H->m[0][3], H->m[1][3], H->m[2][3]);
    Ri = invert_rotmat(&R);
    s = apply_rotmat(&Ri, s);
    o.x = -s.x; o.y = -s.y; o.z = -s.z; o.yaw = e.x; o.pitch = e.y; o.roll = e.z;
    return o;
}
static lre_t invert_lre(lre_t l)                                                   /* transforms.hpp:232-235 */
{
    m44 h = lre2homo(l);
    m44 hi = invert_homo(&h);
    return homo2lre(&hi);
}
static m33 invert_intrinsic(const m33 *K)                                          /* utils.hpp:142-160 */
{
    float fx_inv = 1.0f / K->m[0][0];
    float fy_inv = 1.0f / K->m[1][1];
    float cx = K->m[0][2], cy = K->m[1][2];
    m33 o;
    o.m[0][0] = fx_inv; o.m[0][1] = 0.0f;   o.m[0][2] = -cx * fx_inv;
    o.m[1][0] = 0.0f;   o.m[1][1] = fy_inv; o.m[1][2] = -cy * fy_inv;
    o.m[2][0] = 0.0f;   o.m[2][1] = 0.0f;   o.m[2][2] = 1.0f;
    return o;
}
static f3 apply_matrix33(const m33 *m, f3 v)                                       /* utils.hpp:134-140 */
{
    f3 o;
    o.x = m->m[0][0] * v.x + m->m[0][1] * v.y + m->m[0][2] * v.z;
    o.y = m->m[1][0] * v.x + m->m[1][1] * v.y + m->m[1][2] * v.z;
    o.z = m->m[2][0] * v.x + m->m[2][1] * v.y + m->m[2][2] * v.z;
    return o;
}

/* Ray.hpp:5-24 (fields the path reads) */
typedef struct { f3 origin, direction, direction_inv; } ray_t;
static ray_t make_ray(f3 o, f3 d)
{
    ray_t r; r.origin = o; r.direction = d;
    r.direction_inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}

/* BVHTree.hpp:40-54 */
static float aabb_ray_intersects(f3 bmin, f3 bmax, const ray_t *ray)
{
    f3 tmin = mul3(sub3(bmin, ray->origin), ray->direction_inv);
    f3 tmax = mul3(sub3(bmax, ray->origin), ray->direction_inv);
    f3 t1 = mk3(fminf(tmin.x, tmax.x), fminf(tmin.y, tmax.y), fminf(tmin.z, tmax.z));
    f3 t2 = mk3(fmaxf(tmin.x, tmax.x), fmaxf(tmin.y, tmax.y), fmaxf(tmin.z, tmax.z));
    float dst_far = fminf(fminf(t2.x, t2.y), t2.z);
    float dst_near = fmaxf(fmaxf(t1.x, t1.y), t1.z);
    int hit = dst_far >= dst_near && dst_far > 0.0f;
    return hit ? dst_near : FLT_MAX;
}

/* TrianglePrimitive.hpp:62-79 */
static f3 tri_ray_intersect(const tri_t *t, const ray_t *ray)
{
    float denom = dot3(ray->direction, t->normal);
    float tt;
    if ((double)fabsf(denom) < 1e-6) return mk3(FLT_MAX, FLT_MAX, FLT_MAX);
    tt = dot3(sub3(t->v[0], ray->origin), t->normal) / denom;
    if (tt < 0.0f) return mk3(FLT_MAX, FLT_MAX, FLT_MAX);
    return add3(ray->origin, mul3s(ray->direction, tt));
}

/* TrianglePrimitive.hpp:151-185 */
static f2 tri_point_inside(const tri_t *t, f3 point)
{
    f3 v0 = sub3(t->v[2], t->v[0]);
    f3 v1 = sub3(t->v[1], t->v[0]);
    f3 v2 = sub3(point, t->v[0]);
    float dot00 = dot3(v0, v0), dot01 = dot3(v0, v1), dot02 = dot3(v0, v2);
    float dot11 = dot3(v1, v1), dot12 = dot3(v1, v2);
    float invDenom = 1.0f / (dot00 * dot11 - dot01 * dot01);
    float u = (dot11 * dot02 - dot01 * dot12) * invDenom;
    float v = (dot00 * dot12 - dot01 * dot02) * invDenom;
    f2 r;
    if ((u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f)) {
        float w = 1.0f - u - v;
        /* (w*uv0) + (v*uv1) + (u*uv2), utils.hpp float2 operators :97-111 */
        r.x = (w * t->uv[0].x + v * t->uv[1].x) + u * t->uv[2].x;
        r.y = (w * t->uv[0].y + v * t->uv[1].y) + u * t->uv[2].y;
        return r;
    }
    r.x = FLT_MAX; r.y = FLT_MAX;
    return r;
}

static f3 tri_center(const tri_t *t)                                               /* TrianglePrimitive.hpp:81-83 */
{
    f3 s = add3(add3(t->v[0], t->v[1]), t->v[2]);
    return mk3(s.x / 3.0f, s.y / 3.0f, s.z / 3.0f);
}

/* TrianglePrimitive.hpp:15-23: 3-vertex constructor computes the normal; uv zeroed (H6) */
static tri_t tri_from_vertices(f3 a, f3 b, f3 c)
{
    tri_t t; memset(&t, 0, sizeof t);
    t.v[0] = a; t.v[1] = b; t.v[2] = c;
    t.normal = normalize3(cross3(sub3(b, a), sub3(c, a)));
    return t;
}

/* ------------------------------------------------------------ BVH (host build) */

typedef struct {
    f3 bmin, bmax;
    int child_a, child_b;      /* -1 = leaf (BVHTree.hpp:66-67) */
    int *idx; int count;       /* triangle_indices (kept on every node, like the reference) */
} node_t;

typedef struct OrcMesh {
    tri_t *tris; int ntris;
    node_t *nodes; int nnodes, cap;
} OrcMesh;

static void box_init(f3 *mn, f3 *mx)                                               /* BVHTree.hpp:73-81 */
{ *mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX); *mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX); }
static void box_grow_v(f3 *mn, f3 *mx, f3 v)                                       /* BVHTree.hpp:182-190 */
{
    mn->x = fminf(mn->x, v.x); mn->y = fminf(mn->y, v.y); mn->z = fminf(mn->z, v.z);
    mx->x = fmaxf(mx->x, v.x); mx->y = fmaxf(mx->y, v.y); mx->z = fmaxf(mx->z, v.z);
}
static void box_grow_t(f3 *mn, f3 *mx, const tri_t *t)                             /* BVHTree.hpp:175-180 */
{ int i; for (i = 0; i < 3; i++) box_grow_v(mn, mx, t->v[i]); }
static float box_cost(f3 mn, f3 mx, size_t count)                                  /* BVHTree.hpp:192-201 */
{
    f3 size; float half_area;
    if (count == 0) return FLT_MAX;
    size = sub3(mx, mn);
    half_area = size.x * (size.y + size.z) + size.y * size.z;
    return half_area * (float)count;
}
static float axis_of(f3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

/* BVHTree.hpp:294-361 ; returns best_cost, writes best_split */
static float evaluate_split(const OrcMesh *m, const node_t *n, int axis, float *best_split_out)
{
    float tests_per_axis = 5;
    float best_cost = FLT_MAX, best_split = 0.0f;
    int s, i;
    for (s = 0; s < tests_per_axis; s++) {
        float split_t = ((float)s + 1) / (tests_per_axis + 1);
        float min_check = axis_of(n->bmin, axis), max_check = axis_of(n->bmax, axis);
        float pos = min_check + (max_check - min_check) * (split_t);
        f3 lmin, lmax, rmin, rmax; size_t lc = 0, rc = 0; float cost;
        box_init(&lmin, &lmax); box_init(&rmin, &rmax);
        for (i = 0; i < n->count; i++) {
            const tri_t *t = &m->tris[n->idx[i]];
            float tri_check = axis_of(tri_center(t), axis);
            if (tri_check <= pos) { box_grow_t(&lmin, &lmax, t); lc++; }
            else { box_grow_t(&rmin, &rmax, t); rc++; }
        }
        cost = box_cost(lmin, lmax, lc) + box_cost(rmin, rmax, rc);
        if (cost < best_cost) { best_cost = cost; best_split = pos; }
    }
    *best_split_out = best_split;
    return best_cost;
}

static int push_node(OrcMesh *m, int *idx, int count)
{
    node_t *n;
    if (m->nnodes == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 64;
        m->nodes = (node_t *)realloc(m->nodes, (size_t)m->cap * sizeof(node_t));
    }
    n = &m->nodes[m->nnodes];
    box_init(&n->bmin, &n->bmax);
    n->child_a = -1; n->child_b = -1; n->idx = idx; n->count = count;
    return m->nnodes++;
}

/* BVHTree.hpp:203-292 */
static void bvh_fill(OrcMesh *m, int self, int depth, int max_depth)
{
    int i, axis, nl = 0, nr = 0, *left, *right, a, b;
    float ex, ey, ez, sx, sy, sz, split_pos, best_cost;
    node_t n;
    for (i = 0; i < m->nodes[self].count; i++)
        box_grow_t(&m->nodes[self].bmin, &m->nodes[self].bmax, &m->tris[m->nodes[self].idx[i]]);
    if (depth >= max_depth) return;
    if (m->nodes[self].count <= 1) return;
    n = m->nodes[self];
    ex = evaluate_split(m, &n, 0, &sx);
    ey = evaluate_split(m, &n, 1, &sy);
    ez = evaluate_split(m, &n, 2, &sz);
    if (ex < ey && ex < ez) { axis = 0; split_pos = sx; best_cost = ex; }
    else if (ey < ex && ey < ez) { axis = 1; split_pos = sy; best_cost = ey; }
    else { axis = 2; split_pos = sz; best_cost = ez; }
    if (best_cost >= box_cost(n.bmin, n.bmax, (size_t)n.count)) return;
    left = (int *)malloc(sizeof(int) * (size_t)n.count);
    right = (int *)malloc(sizeof(int) * (size_t)n.count);
    for (i = 0; i < n.count; i++) {
        int idx = n.idx[i];
        float tri_check = axis_of(tri_center(&m->tris[idx]), axis);
        if (tri_check <= split_pos) left[nl++] = idx; else right[nr++] = idx;
    }
    if (nl == 0 || nr == 0) { free(left); free(right); return; }
    a = push_node(m, left, nl);
    m->nodes[self].child_a = a;
    bvh_fill(m, a, depth + 1, max_depth);
    b = push_node(m, right, nr);
    m->nodes[self].child_b = b;
    bvh_fill(m, b, depth + 1, max_depth);
}

/* MeshPrimitive.cpp:5-16,38-56 ; tris18 = n x {v0 v1 v2 normal uv0 uv1 uv2} */
OrcMesh *orc_mesh_from_triangles(const float *tris18, int n)
{
    OrcMesh *m = (OrcMesh *)calloc(1, sizeof(OrcMesh));
    int i, *all;
    m->tris = (tri_t *)malloc(sizeof(tri_t) * (size_t)(n > 0 ? n : 1));
    m->ntris = n;
    memcpy(m->tris, tris18, sizeof(tri_t) * (size_t)n);
    all = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (i = 0; i < n; i++) all[i] = i;
    push_node(m, all, n);
    bvh_fill(m, 0, 1, 32);
    return m;
}

/* Refit (no counterpart in the reference, which can only re-pose instances: Scene.cpp:67-74): the same tree over moved
 * triangles.  Every node keeps its triangle list (like the reference's nodes) and gets the bounds BVHTree::fill's bounds
 * pass (BVHTree.hpp:206-209) would compute over that list; topology and lists are untouched. */
int orc_mesh_refit(OrcMesh *m, const float *tris18, int n)
{
    int k, i;
    if (!m || n != m->ntris) return -1;
    memcpy(m->tris, tris18, sizeof(tri_t) * (size_t)n);
    for (k = 0; k < m->nnodes; k++) {
        node_t *nd = &m->nodes[k];
        box_init(&nd->bmin, &nd->bmax);
        for (i = 0; i < nd->count; i++) box_grow_t(&nd->bmin, &nd->bmax, &m->tris[nd->idx[i]]);
    }
    return 0;
}

/* one triangle from 3 vertices via the 3-vertex constructor (config C1) */
OrcMesh *orc_mesh_single_triangle(const float *abc9)
{
    tri_t t = tri_from_vertices(mk3(abc9[0], abc9[1], abc9[2]), mk3(abc9[3], abc9[4], abc9[5]),
                                mk3(abc9[6], abc9[7], abc9[8]));
    return orc_mesh_from_triangles((const float *)&t, 1);
}

void orc_mesh_free(OrcMesh *m)
{
    int i;
    if (!m) return;
    for (i = 0; i < m->nnodes; i++) free(m->nodes[i].idx);
    free(m->nodes); free(m->tris); free(m);
}
int orc_mesh_num_triangles(const OrcMesh *m) { return m->ntris; }
int orc_mesh_num_nodes(const OrcMesh *m) { return m->nnodes; }
void orc_mesh_get_triangles(const OrcMesh *m, float *out18) { memcpy(out18, m->tris, sizeof(tri_t) * (size_t)m->ntris); }
/* boxes: n x 6, child: n x 2, leaf_count: n (0 for interior), returns total leaf indices */
int orc_mesh_get_nodes(const OrcMesh *m, float *boxes, int *child, int *leaf_count)
{
    int i, total = 0;
    for (i = 0; i < m->nnodes; i++) {
        const node_t *n = &m->nodes[i];
        int leaf = (n->child_a == -1 && n->child_b == -1);
        if (boxes) { boxes[i*6+0] = n->bmin.x; boxes[i*6+1] = n->bmin.y; boxes[i*6+2] = n->bmin.z;
                     boxes[i*6+3] = n->bmax.x; boxes[i*6+4] = n->bmax.y; boxes[i*6+5] = n->bmax.z; }
        if (child) { child[i*2] = n->child_a; child[i*2+1] = n->child_b; }
        if (leaf_count) leaf_count[i] = leaf ? n->count : 0;
        if (leaf) total += n->count;
    }
    return total;
}
/* concatenated leaf lists in node order */
void orc_mesh_get_leaf_indices(const OrcMesh *m, int *out)
{
    int i, k = 0;
    for (i = 0; i < m->nnodes; i++) {
        const node_t *n = &m->nodes[i];
        if (n->child_a == -1 && n->child_b == -1) { memcpy(out + k, n->idx, sizeof(int) * (size_t)n->count); k += n->count; }
    }
}
/* BVHTree.hpp:117-172 print_stats fields: nodes, max/min tris per leaf, max depth (stack size metric), leaves */
void orc_mesh_stats(const OrcMesh *m, int *out5)
{
    int count_nodes = 0, max_t = 0, min_t = 1000000, max_depth = 0, leaves = 0;
    int *stack = (int *)malloc(sizeof(int) * (size_t)(m->nnodes + 2)), sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const node_t *n = &m->nodes[stack[--sp]];
        max_depth = (int)fmaxf((float)max_depth, (float)sp);
        count_nodes++;
        if (n->child_a == -1) {
            if (n->count > max_t) max_t = n->count;
            if (n->count < min_t) min_t = n->count;
            leaves++;
        }
        if (n->child_a != -1) { stack[sp++] = n->child_a; stack[sp++] = n->child_b; }
    }
    free(stack);
    out5[0] = count_nodes; out5[1] = max_t; out5[2] = min_t; out5[3] = max_depth; out5[4] = leaves;
}

/* ---------------------------------------------------------------- OBJ loader */

/* OBJLoader.hpp:15-179.  Tokens split on whitespace (istream_iterator<string>), stof/stoi
 * semantics = strtof / leading-integer parse.  Returns NULL when the file cannot be opened
 * (the reference prints and exit(1)s, :23-27) or on a malformed token (the reference throws). */
typedef struct { char **tok; int n, cap; } toks_t;
static void split_ws(char *line, toks_t *t)
{
    char *p = line;
    t->n = 0;
    while (*p) {
        while (*p && isspace((unsigned char)*p)) p++;
        if (!*p) break;
        if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 16; t->tok = (char **)realloc(t->tok, sizeof(char *) * (size_t)t->cap); }
        t->tok[t->n++] = p;
        while (*p && !isspace((unsigned char)*p)) p++;
        if (*p) *p++ = 0;
    }
}
static int parse_stoi(const char *s, int *out)
{
    char *end; long v = strtol(s, &end, 10);
    if (end == s) return -1;
    *out = (int)v; return 0;
}

OrcMesh *orc_obj_load(const char *path)
{
    FILE *f = fopen(path, "rb");
    f3 *verts = NULL; f2 *tcs = NULL; tri_t *tris = NULL;
    size_t nv = 0, cv = 0, nt = 0, ct = 0, ntri = 0, ctri = 0;
    char *line = NULL; size_t lcap = 0; ssize_t len;
    toks_t t = {0, 0, 0};
    int pass, bad = 0;
    OrcMesh *m;
    if (!f) return NULL;
    for (pass = 0; pass < 2 && !bad; pass++) {
        rewind(f);
        while ((len = getline(&line, &lcap, f)) >= 0) {
            split_ws(line, &t);
            if (t.n == 0) continue;
            if (pass == 0) {
                if (strcmp(t.tok[0], "v") == 0) {                       /* :46-53 */
                    if (t.n < 4) { bad = 1; break; }
                    if (nv == cv) { cv = cv ? cv * 2 : 1024; verts = (f3 *)realloc(verts, cv * sizeof(f3)); }
                    verts[nv++] = mk3(strtof(t.tok[1], NULL), strtof(t.tok[2], NULL), strtof(t.tok[3], NULL));
                }
                /* "vn" parsed but never used (:55-62) */
                if (strcmp(t.tok[0], "vt") == 0) {                      /* :64-70 */
                    if (t.n < 3) { bad = 1; break; }
                    if (nt == ct) { ct = ct ? ct * 2 : 1024; tcs = (f2 *)realloc(tcs, ct * sizeof(f2)); }
                    tcs[nt].x = strtof(t.tok[1], NULL); tcs[nt].y = strtof(t.tok[2], NULL); nt++;
                }
            } else if (strcmp(t.tok[0], "f") == 0) {                    /* :90-171 */
                int nvi = 0, nti = 0, i;
                int *vi = (int *)malloc(sizeof(int) * (size_t)t.n), *ti = (int *)malloc(sizeof(int) * (size_t)t.n);
                for (i = 1; i < t.n && !bad; i++) {
                    char *s1 = strchr(t.tok[i], '/');
                    char *s2 = s1 ? strchr(s1 + 1, '/') : NULL;
                    int v;
                    if (s1) *s1 = 0;
                    if (parse_stoi(t.tok[i], &v)) { bad = 1; break; }
                    vi[nvi++] = v - 1;
                    if (s1) { if (parse_stoi(s1 + 1, &v)) { bad = 1; break; } ti[nti++] = v - 1; }
                    if (s2) { if (parse_stoi(s2 + 1, &v)) { bad = 1; break; } }
                }
                for (i = 1; !bad && i + 1 < nvi; i++) {                  /* fan 0,i,i+1 :139-169 */
                    tri_t tr; memset(&tr, 0, sizeof tr);
                    if (vi[0] < 0 || vi[i] < 0 || vi[i+1] < 0 || (size_t)vi[0] >= nv || (size_t)vi[i] >= nv || (size_t)vi[i+1] >= nv) { bad = 1; break; }
                    tr.v[0] = verts[vi[0]]; tr.v[1] = verts[vi[i]]; tr.v[2] = verts[vi[i+1]];
                    tr.normal = normalize3(cross3(sub3(tr.v[1], tr.v[0]), sub3(tr.v[2], tr.v[0])));   /* :141-143 */
                    if (nti > 0) {
                        if (nti <= i + 1 || ti[0] < 0 || ti[i] < 0 || ti[i+1] < 0 || (size_t)ti[0] >= nt || (size_t)ti[i] >= nt || (size_t)ti[i+1] >= nt) { bad = 1; break; }
                        tr.uv[0] = tcs[ti[0]]; tr.uv[1] = tcs[ti[i]]; tr.uv[2] = tcs[ti[i+1]];
                    }
                    if (ntri == ctri) { ctri = ctri ? ctri * 2 : 1024; tris = (tri_t *)realloc(tris, ctri * sizeof(tri_t)); }
                    tris[ntri++] = tr;
                }
                free(vi); free(ti);
                if (bad) break;
            }
        }
    }
    fclose(f); free(line); free(t.tok);
    if (bad) { free(verts); free(tcs); free(tris); return NULL; }
    m = orc_mesh_from_triangles((const float *)tris, (int)ntri);
    free(verts); free(tcs); free(tris);
    return m;
}

/* --------------------------------------------------------------------- scene */

typedef struct {                                    /* Material.hpp:6-16 */
    f3 albedo;
    const uint8_t *texture; int texture_width, texture_height; size_t texture_pitch;
    float roughness, metallic, illumination;        /* dead in the reference; read only by the extension (orc_render_ex) */
} material_t;

typedef struct {                                    /* MeshInstance.hpp:6-18 */
    int mesh_index, material_index;
    lre_t pose, inv_pose;
    f3 rotation, inv_rotation, scale, inv_scale;
} instance_t;

typedef struct OrcScene {
    material_t *materials; int nmat;
    OrcMesh **meshes; int nmesh;
    instance_t *instances; int ninst;
} OrcScene;

OrcScene *orc_scene_create(void) { return (OrcScene *)calloc(1, sizeof(OrcScene)); }
void orc_scene_free(OrcScene *s) { if (!s) return; free(s->materials); free(s->meshes); free(s->instances); free(s); }

/* texture: BGR bytes, caller keeps it alive; NULL/0 = albedo path (raycast.cu:224) */
int orc_scene_add_material(OrcScene *s, const float *albedo3, const uint8_t *tex, int w, int h, size_t pitch)
{
    material_t *m;
    s->materials = (material_t *)realloc(s->materials, sizeof(material_t) * (size_t)(s->nmat + 1));
    m = &s->materials[s->nmat];
    m->albedo = mk3(albedo3[0], albedo3[1], albedo3[2]);
    m->texture = tex; m->texture_width = tex ? w : 0; m->texture_height = tex ? h : 0; m->texture_pitch = pitch;
    m->roughness = 0.0f; m->metallic = 0.0f; m->illumination = 0.0f;                 /* Material.hpp:19 */
    return s->nmat++;
}
int orc_scene_set_material_params(OrcScene *s, int index, float roughness, float metallic, float illumination)
{
    if (index < 0 || index >= s->nmat) return -1;
    s->materials[index].roughness = roughness; s->materials[index].metallic = metallic; s->materials[index].illumination = illumination;
    return 0;
}
int orc_scene_add_mesh(OrcScene *s, OrcMesh *m)
{
    s->meshes = (OrcMesh **)realloc(s->meshes, sizeof(OrcMesh *) * (size_t)(s->nmesh + 1));
    s->meshes[s->nmesh] = m;
    return s->nmesh++;
}
/* MeshInstance.hpp:39-46 build_inv */
static void instance_build_inv(instance_t *in)
{
    in->inv_pose = invert_lre(in->pose);
    in->inv_scale = mk3(1 / in->scale.x, 1 / in->scale.y, 1 / in->scale.z);
    in->rotation = mk3(in->pose.yaw, in->pose.pitch, in->pose.roll);
    in->inv_rotation = mk3(in->inv_pose.yaw, in->inv_pose.pitch, in->inv_pose.roll);
}
int orc_scene_add_instance(OrcScene *s, int mesh_index, int material_index, const float *pose6, const float *scale3)
{
    instance_t *in;
    s->instances = (instance_t *)realloc(s->instances, sizeof(instance_t) * (size_t)(s->ninst + 1));
    in = &s->instances[s->ninst];
    in->mesh_index = mesh_index; in->material_index = material_index;
    memcpy(&in->pose, pose6, sizeof(lre_t));
    in->scale = mk3(scale3[0], scale3[1], scale3[2]);
    instance_build_inv(in);                         /* Scene.cpp:59 */
    return s->ninst++;
}
/* Scene.cpp:67-74 */
int orc_scene_update_instance(OrcScene *s, int index, int mesh_index, int material_index, const float *pose6, const float *scale3)
{
    instance_t *in;
    if (index < 0 || index >= s->ninst) return -1;
    in = &s->instances[index];
    in->mesh_index = mesh_index; in->material_index = material_index;
    memcpy(&in->pose, pose6, sizeof(lre_t));
    in->scale = mk3(scale3[0], scale3[1], scale3[2]);
    instance_build_inv(in);
    return 0;
}

/* -------------------------------------------------------------------- render */

typedef struct {                                    /* raycast.cu:10-18 + bookkeeping for parity */
    float min;
    f3 location, normal;
    material_t material;
    f2 uv;
    int hit_instance, hit_triangle;
    f3 hit_location;                                /* location of the ACCEPTED hit (`location` above is overwritten by every
                                                       inside candidate, accepted or not: raycast.cu:98-102 precede :109) */
    int pops, aabb_tests, tri_tests, inside_hits, max_stack;
} hit_t;

/* optional per-ray step recorder (analysis tooling): one byte per node pop, 0 = interior,
 * k >= 1 = leaf with k-1 triangle tests (capped at 254) */
static __thread uint8_t *g_rec = NULL;
static __thread int g_rec_n = 0, g_rec_cap = 0;
static __thread int32_t *g_rec_nodes = NULL;
static void rec_step(int v) { if (g_rec && g_rec_n < g_rec_cap) g_rec[g_rec_n] = (uint8_t)(v > 255 ? 255 : v); if (g_rec) g_rec_n++; }
static void rec_node(int node) { if (g_rec_nodes && g_rec_n < g_rec_cap) g_rec_nodes[g_rec_n] = node; }

/* raycast.cu:21-142.  lighting_pass / light_distance: the reference's last two parameters (:21); they are read only by the
 * early return of :129-133, which the snapshot carries commented out -- the extension restores it for its shadow rays
 * (orc_render_ex), the reference path (orc_render) never sets lighting_pass. */
static hit_t cast_ray_lp(const ray_t *ray, const OrcScene *sc, int lighting_pass, float light_distance)
{
    hit_t hit; int mesh_idx;
    memset(&hit, 0, sizeof hit);
    hit.min = FLT_MAX; hit.hit_instance = -1; hit.hit_triangle = -1;

    for (mesh_idx = 0; mesh_idx < sc->ninst; mesh_idx++) {
        instance_t inst = sc->instances[mesh_idx];
        const OrcMesh *mesh = sc->meshes[inst.mesh_index];
        material_t material = sc->materials[inst.material_index];
        f3 r_direction, r_origin; ray_t r_ray;
        int stack[32]; int stack_index = 0;

        r_direction = apply_euler(inst.rotation, ray->direction);        /* :33-37 */
        r_direction.x *= inst.inv_scale.x; r_direction.y *= inst.inv_scale.y; r_direction.z *= inst.inv_scale.z;
        r_origin = apply_lre(inst.pose, ray->origin);                    /* :40-45 */
        r_origin.x *= inst.inv_scale.x; r_origin.y *= inst.inv_scale.y; r_origin.z *= inst.inv_scale.z;
        r_ray = make_ray(r_origin, r_direction);                         /* :47-51 */

        stack[stack_index++] = 0;                                        /* :58 */
        while (stack_index > 0) {
            int node_index;
            const node_t *cur;
            if (stack_index > hit.max_stack) hit.max_stack = stack_index;
            node_index = stack[--stack_index];                           /* :61 */
            rec_node(node_index);
            cur = &mesh->nodes[node_index];
            hit.pops++;
            if (cur->child_a > 0) {                                      /* :66 */
                const node_t *na = &mesh->nodes[cur->child_a], *nb = &mesh->nodes[cur->child_b];
                float dist_a = aabb_ray_intersects(na->bmin, na->bmax, &r_ray);
                float dist_b = aabb_ray_intersects(nb->bmin, nb->bmax, &r_ray);
                hit.aabb_tests += 2;
                rec_step(0);
                if (dist_a < dist_b) {                                   /* :72-79 */
                    if (dist_b < hit.min) stack[stack_index++] = cur->child_b;
                    if (dist_a < hit.min) stack[stack_index++] = cur->child_a;
                } else {
                    if (dist_a < hit.min) stack[stack_index++] = cur->child_a;
                    if (dist_b < hit.min) stack[stack_index++] = cur->child_b;
                }
            } else {
                int i;
                rec_step(1 + cur->count);
                for (i = 0; i < cur->count; i++) {                       /* :85-136 */
                    int index = cur->idx[i];
                    const tri_t *tri = &mesh->tris[index];
                    f3 isect; f2 uv;
                    hit.tri_tests++;
                    isect = tri_ray_intersect(tri, &r_ray);
                    if (isect.x == FLT_MAX) continue;
                    uv = tri_point_inside(tri, isect);
                    if (uv.x != FLT_MAX) {
                        float distance, same_dir;
                        hit.inside_hits++;
                        hit.location.x = isect.x * inst.scale.x;         /* :98-102 */
                        hit.location.y = isect.y * inst.scale.y;
                        hit.location.z = isect.z * inst.scale.z;
                        hit.location = apply_lre(inst.inv_pose, hit.location);
                        distance = magnitude3(sub3(hit.location, ray->origin));
                        same_dir = dot3(r_ray.direction, tri->normal);
                        if (same_dir < 0 && (hit.min == FLT_MAX || distance < hit.min)) {
                            hit.min = distance;
                            hit.normal = apply_euler(inst.inv_rotation, tri->normal);    /* :115-122 */
                            hit.normal.x *= inst.scale.x; hit.normal.y *= inst.scale.y; hit.normal.z *= inst.scale.z;
                            hit.normal = normalize3(hit.normal);
                            hit.uv = uv;
                            hit.material = material;
                            hit.hit_instance = mesh_idx; hit.hit_triangle = index;
                            hit.hit_location = hit.location;
                            if (lighting_pass && distance < light_distance) return hit;  /* :129-133 (commented out in the snapshot) */
                        }
                    }
                }
            }
        }
    }
    return hit;
}

static hit_t cast_ray(const ray_t *ray, const OrcScene *sc) { return cast_ray_lp(ray, sc, 0, FLT_MAX); }    /* the defaults of :21 */

typedef struct {
    int width, height;
    m33 K_inv; f4 D; lre_t camera_pose, inv_camera_pose;
} camera_t;

/* raycast.cu:156-188: primary ray of pixel (x, y) */
static ray_t camera_ray(const camera_t *cam, int x, int y)
{
    f3 origin = mk3(cam->camera_pose.x, cam->camera_pose.y, cam->camera_pose.z);
    f3 ph = mk3((float)x, (float)y, 1.0f);
    f3 direction = apply_matrix33(&cam->K_inv, ph);
    float a = direction.x, b = direction.y;
    float radius = sqrtf(a * a + b * b);
    float theta = atanf(radius);
    f4 D = cam->D;
    /* :172 -- products in float, the sum and the outer product in double */
    float thetad = (float)((double)theta * (1.0 + (double)(D.x * theta) + (double)(D.y * theta * theta)
                   + (double)(D.z * theta * theta * theta) + (double)(D.w * theta * theta * theta * theta)));
    float scale = thetad / radius;
    ray_t r;
    direction.x = scale * a;
    direction.y = scale * b;
    direction = normalize3(direction);
    direction = mk3(direction.x, direction.z, -direction.y);                       /* :182 */
    direction = apply_euler(mk3(cam->inv_camera_pose.yaw, cam->inv_camera_pose.pitch, cam->inv_camera_pose.roll), direction);
    direction = normalize3(direction);
    r = make_ray(origin, direction);
    return r;
}

static uint8_t f2u8(float f) { return (uint8_t)(int)f; }
static uint8_t d2u8(double f) { return (uint8_t)(int)f; }

/* raycast.cu:207-294 ; writes 3 bytes (uchar3 .x .y .z) */
static void shade(const hit_t *hit, uint8_t *px)
{
    f3 color = mk3(1.0f, 1.0f, 1.0f);                                              /* Ray.hpp:22 */
    float illumination;
    if (hit->min == FLT_MAX) {                                                     /* :208-216 */
        px[0] = d2u8(1.0 * 255); px[1] = d2u8(0.8 * 255); px[2] = d2u8(0.6 * 255);
        return;
    }
    if (hit->material.texture_width > 0) {                                         /* :224-240 */
        int tw = hit->material.texture_width, th = hit->material.texture_height;
        int tex_x = (int)(hit->uv.x * (float)tw);
        int tex_y = (int)((1.0 - (double)hit->uv.y) * (double)(float)th);
        const uint8_t *row, *tc;
        tex_x = (int)fmaxf((float)(tex_x % tw), 0);
        tex_y = (int)fmaxf((float)(tex_y % th), 0);
        row = hit->material.texture + (size_t)tex_y * hit->material.texture_pitch;
        tc = row + 3 * (size_t)tex_x;
        color.x *= (float)tc[0] * 0.0039215f;
        color.y *= (float)tc[1] * 0.0039215f;
        color.z *= (float)tc[2] * 0.0039215f;
    } else {                                                                       /* :241-245 */
        color.x *= hit->material.albedo.x; color.y *= hit->material.albedo.y; color.z *= hit->material.albedo.z;
    }
    illumination = 1.0f;                                                           /* :282 */
    illumination = fminf(1.0f, illumination);                                      /* :289-290 */
    illumination = fmaxf(0.4f, illumination);
    px[0] = f2u8(illumination * color.x * 255);                                    /* :292-294 */
    px[1] = f2u8(illumination * color.y * 255);
    px[2] = f2u8(illumination * color.z * 255);
}

/*
 * Camera.cu:6-41 + raycast.cu:146-297 for rows [y0, y1).  K9 row-major, D4, cam_pose6.
 * Optional per-pixel planes (tight, width*height int32 each, may be NULL):
 *   hit_inst, hit_tri (-1 = miss), pops, aabb, tris, inside.
 * stats8 (may be NULL) accumulates: rays, pops, aabb_tests, tri_tests, inside_hits, hits, max_stack, 0.
 */
int orc_render(const OrcScene *sc, int width, int height, const float *K9, const float *D4, const float *cam_pose6,
               uint8_t *img, size_t pitch, int y0, int y1,
               int32_t *hit_inst, int32_t *hit_tri, int32_t *pops, int32_t *aabb, int32_t *tris, int32_t *inside,
               int64_t *stats8)
{
    camera_t cam; m33 K; int x, y;
    int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!sc || !img || width <= 0 || height <= 0) return -1;
    memcpy(&K, K9, sizeof K);
    cam.width = width; cam.height = height;
    cam.K_inv = invert_intrinsic(&K);                                              /* Camera.cu:12 */
    cam.D.x = D4[0]; cam.D.y = D4[1]; cam.D.z = D4[2]; cam.D.w = D4[3];
    memcpy(&cam.camera_pose, cam_pose6, sizeof(lre_t));
    cam.inv_camera_pose = invert_lre(cam.camera_pose);                             /* Camera.cu:21 */
    if (y0 < 0) y0 = 0;
    if (y1 > height) y1 = height;
    for (y = y0; y < y1; y++) {
        uint8_t *row = img + (size_t)y * pitch;
        for (x = 0; x < width; x++) {
            ray_t ray = camera_ray(&cam, x, y);
            hit_t hit = cast_ray(&ray, sc);
            size_t p = (size_t)y * (size_t)width + (size_t)x;
            shade(&hit, row + 3 * (size_t)x);
            if (hit_inst) hit_inst[p] = hit.hit_instance;
            if (hit_tri) hit_tri[p] = hit.hit_triangle;
            if (pops) pops[p] = hit.pops;
            if (aabb) aabb[p] = hit.aabb_tests;
            if (tris) tris[p] = hit.tri_tests;
            if (inside) inside[p] = hit.inside_hits;
            st[0]++; st[1] += hit.pops; st[2] += hit.aabb_tests; st[3] += hit.tri_tests; st[4] += hit.inside_hits;
            st[5] += (hit.min != FLT_MAX);
            if (hit.max_stack > st[6]) st[6] = hit.max_stack;
        }
    }
    if (stats8) { int i; for (i = 0; i < 6; i++) stats8[i] += st[i]; if (st[6] > stats8[6]) stats8[6] = st[6]; }
    return 0;
}

/* step sequence of the ray of pixel (x, y): returns the number of node pops, writes min(n, cap) bytes */
void orc_trace_set_node_buffer(int32_t *nodes) { g_rec_nodes = nodes; }
int orc_trace_steps(const OrcScene *sc, int width, int height, const float *K9, const float *D4, const float *cam_pose6,
                    int x, int y, uint8_t *out, int cap)
{
    camera_t cam; m33 K; ray_t r;
    memcpy(&K, K9, sizeof K);
    cam.width = width; cam.height = height;
    cam.K_inv = invert_intrinsic(&K);
    cam.D.x = D4[0]; cam.D.y = D4[1]; cam.D.z = D4[2]; cam.D.w = D4[3];
    memcpy(&cam.camera_pose, cam_pose6, sizeof(lre_t));
    cam.inv_camera_pose = invert_lre(cam.camera_pose);
    r = camera_ray(&cam, x, y);
    g_rec = out; g_rec_n = 0; g_rec_cap = cap;
    (void)cast_ray(&r, sc);
    g_rec = NULL;
    return g_rec_n;
}

/* camera-space ray direction of one pixel, for ray-generation parity tests */
void orc_camera_ray(int width, int height, const float *K9, const float *D4, const float *cam_pose6, int x, int y, float *dir3)
{
    camera_t cam; m33 K; ray_t r;
    memcpy(&K, K9, sizeof K);
    cam.width = width; cam.height = height;
    cam.K_inv = invert_intrinsic(&K);
    cam.D.x = D4[0]; cam.D.y = D4[1]; cam.D.z = D4[2]; cam.D.w = D4[3];
    memcpy(&cam.camera_pose, cam_pose6, sizeof(lre_t));
    cam.inv_camera_pose = invert_lre(cam.camera_pose);
    r = camera_ray(&cam, x, y);
    dir3[0] = r.direction.x; dir3[1] = r.direction.y; dir3[2] = r.direction.z;
}


/* ======================================================================================================
 * EXTENSION (BASELINE.json configs[2], [4]): samples per pixel, specular bounces and
 * the sun + shadow pass the reference carries as commented-out code.  The reference snapshot has NO implementation
 * of any of this (one primary ray, illumination hard-wired to 1.0, cuRAND state initialised and never sampled), so
 * these semantics are DEFINED HERE and parity for them is unpinned; they are chosen so that
 * (spp = 1, bounces = 0, lighting = 0) reproduces the reference frame bit for bit.
 *
 *   seed    = (int32)((y * width + x) * 1000), sign-extended            raycast.cu:190
 *   rng     = XORWOW seeded like curand_init(seed, 0, 0) (NVIDIA cuRAND, a third-party dependency that is not in
 *             /root/reference; restated from its published algorithm: Marsaglia xorwow + cuRAND's seed scramble)
 *   sample 0 uses the reference's un-jittered pixel (x, y, 1); sample s > 0 uses (x + u - .5, y + u - .5, 1)
 *   lighting = 1: raycast.cu:249-290 with the commented lines restored: sun direction normalize(-0.2, 0, 1),
 *             illum = 0.4 * cos; if the surface faces the sun, a shadow ray -- cast_ray(..., lighting_pass = true, FLT_MAX)
 *             as written at :272, with the early return of :129-133 restored too: the cast ends at the first accepted hit
 *             whose distance is below light_distance (any occluder will do) -- and illum = 1.0 * cos when it escapes;
 *             then the [0.4, 1] clamp of :289-290
 *   bounces: Material::metallic is the mirror weight, Material::roughness perturbs the mirror direction
 *   secondary rays start at the ACCEPTED hit's location (the reference's hit_info.location may hold a later, rejected
 *   candidate's point because raycast.cu:98-102 run before the acceptance test at :109)
 * ====================================================================================================== */
typedef struct { uint32_t v[5]; uint32_t d; } xorwow_t;
static void xorwow_init(xorwow_t *st, unsigned long long seed)
{
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st->d = 6615241u + t1 + t0;
    st->v[0] = 123456789u + t0;
    st->v[1] = 362436069u ^ t0;
    st->v[2] = 521288629u + t1;
    st->v[3] = 88675123u ^ t1;
    st->v[4] = 5783321u + t0;
}
static uint32_t xorwow_next(xorwow_t *st)
{
    uint32_t t = st->v[0] ^ (st->v[0] >> 2);
    st->v[0] = st->v[1]; st->v[1] = st->v[2]; st->v[2] = st->v[3]; st->v[3] = st->v[4];
    st->v[4] = (st->v[4] ^ (st->v[4] << 4)) ^ (t ^ (t << 1));
    st->d += 362437u;
    return st->v[4] + st->d;
}
/* curand_uniform: (0, 1] */
static float xorwow_uniform(xorwow_t *st) { return (float)xorwow_next(st) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }

/* primary ray through the (possibly jittered) pixel position: raycast.cu:159-188 with ph = (px, py, 1) */
static ray_t camera_ray_at(const camera_t *cam, float px, float py)
{
    f3 origin = mk3(cam->camera_pose.x, cam->camera_pose.y, cam->camera_pose.z);
    f3 direction = apply_matrix33(&cam->K_inv, mk3(px, py, 1.0f));
    float a = direction.x, b = direction.y;
    float radius = sqrtf(a * a + b * b);
    float theta = atanf(radius);
    f4 D = cam->D;
    float thetad = (float)((double)theta * (1.0 + (double)(D.x * theta) + (double)(D.y * theta * theta)
                   + (double)(D.z * theta * theta * theta) + (double)(D.w * theta * theta * theta * theta)));
    float scale = thetad / radius;
    direction.x = scale * a;
    direction.y = scale * b;
    direction = normalize3(direction);
    direction = mk3(direction.x, direction.z, -direction.y);
    direction = apply_euler(mk3(cam->inv_camera_pose.yaw, cam->inv_camera_pose.pitch, cam->inv_camera_pose.roll), direction);
    direction = normalize3(direction);
    return make_ray(origin, direction);
}

/* texture / albedo colour of a hit: raycast.cu:224-245 (ray.color starts at 1,1,1) */
static f3 base_colour(const hit_t *hit)
{
    f3 color = mk3(1.0f, 1.0f, 1.0f);
    if (hit->material.texture_width > 0) {
        int tw = hit->material.texture_width, th = hit->material.texture_height;
        int tex_x = (int)(hit->uv.x * (float)tw);
        int tex_y = (int)((1.0 - (double)hit->uv.y) * (double)(float)th);
        const uint8_t *tc;
        tex_x = (int)fmaxf((float)(tex_x % tw), 0);
        tex_y = (int)fmaxf((float)(tex_y % th), 0);
        tc = hit->material.texture + (size_t)tex_y * hit->material.texture_pitch + 3 * (size_t)tex_x;
        color.x *= (float)tc[0] * 0.0039215f; color.y *= (float)tc[1] * 0.0039215f; color.z *= (float)tc[2] * 0.0039215f;
    } else {
        color.x *= hit->material.albedo.x; color.y *= hit->material.albedo.y; color.z *= hit->material.albedo.z;
    }
    return color;
}

/* raycast.cu:249-290 with the commented lines active; *pops accumulates the shadow ray's node pops */
static float sun_illumination(const hit_t *hit, const OrcScene *sc, int64_t *pops, int64_t *rays)
{
    f3 light_direction = normalize3(mk3(-0.2f, 0.0f, 1.0f));                      /* :249-250 */
    ray_t sray = make_ray(add3(hit->hit_location, mul3s(light_direction, (float)1e-4)), light_direction);   /* :254-259 */
    float cos_illum = dot3(hit->normal, light_direction);                          /* :263 */
    float illum = (float)(0.4 * (double)cos_illum);                                /* :266 */
    if (dot3(hit->normal, light_direction) > 0) {                                  /* :268 */
        hit_t sh = cast_ray_lp(&sray, sc, 1, FLT_MAX);                             /* :272: cast_ray(..., true, FLT_MAX) */
        *pops += sh.pops; *rays += 1;
        if (sh.min == FLT_MAX) illum = (float)(1.0 * (double)cos_illum);           /* :276-279 */
    }
    illum = fminf(1.0f, illum);                                                    /* :289-290 */
    illum = fmaxf(0.4f, illum);
    return illum;
}

/* rows [y0, y1); total_pops (may be NULL): tight [height][width] int32, node pops of every ray of the pixel;
 * stats4 (may be NULL) accumulates rays cast, node pops, primary hits, 0 */
int orc_render_ex(const OrcScene *sc, int width, int height, const float *K9, const float *D4, const float *cam_pose6,
                  int spp, int bounces, int lighting, uint8_t *img, size_t pitch, int y0, int y1,
                  int32_t *total_pops, int64_t *stats4)
{
    camera_t cam; m33 K; int x, y;
    int64_t st_rays = 0, st_pops = 0, st_hits = 0;
    if (!sc || !img || width <= 0 || height <= 0 || spp < 1 || bounces < 0) return -1;
    memcpy(&K, K9, sizeof K);
    cam.width = width; cam.height = height;
    cam.K_inv = invert_intrinsic(&K);
    cam.D.x = D4[0]; cam.D.y = D4[1]; cam.D.z = D4[2]; cam.D.w = D4[3];
    memcpy(&cam.camera_pose, cam_pose6, sizeof(lre_t));
    cam.inv_camera_pose = invert_lre(cam.camera_pose);
    if (y0 < 0) y0 = 0;
    if (y1 > height) y1 = height;
    for (y = y0; y < y1; y++) {
        uint8_t *row = img + (size_t)y * pitch;
        for (x = 0; x < width; x++) {
            xorwow_t rng; f3 acc = mk3(0.0f, 0.0f, 0.0f); int s; int64_t pops = 0, rays = 0;
            float inv_spp;
            for (s = 0; s < spp; s++) {
                float px = (float)x, py = (float)y;
                ray_t ray; f3 weight = mk3(1.0f, 1.0f, 1.0f), sample = mk3(0.0f, 0.0f, 0.0f); int depth;
                /* one stream per (pixel, sample): the reference's per-pixel seed (raycast.cu:190, int idx * 1000) + s */
                xorwow_init(&rng, (unsigned long long)((long long)(int32_t)((uint32_t)(y * width + x) * 1000u) + (long long)s));
                if (s > 0) { px = px + (xorwow_uniform(&rng) - 0.5f); py = py + (xorwow_uniform(&rng) - 0.5f); }
                ray = camera_ray_at(&cam, px, py);
                for (depth = 0; depth <= bounces; depth++) {
                    hit_t hit = cast_ray(&ray, sc);
                    f3 base, local, r; float illum, m, k;
                    pops += hit.pops; rays++;
                    if (hit.min == FLT_MAX) {
                        sample = add3(sample, mul3(weight, mk3(1.0f, 0.8f, 0.6f)));
                        break;
                    }
                    if (depth == 0 && s == 0) st_hits++;
                    base = base_colour(&hit);
                    illum = lighting ? sun_illumination(&hit, sc, &pops, &rays) : fmaxf(0.4f, fminf(1.0f, 1.0f));
                    local = mk3(illum * base.x, illum * base.y, illum * base.z);
                    m = depth < bounces ? hit.material.metallic : 0.0f;
                    sample = add3(sample, mul3(weight, mul3s(local, 1.0f - m)));
                    if (!(m > 0.0f)) break;
                    weight = mul3(weight, mul3s(base, m));
                    k = 2.0f * dot3(ray.direction, hit.normal);
                    r = sub3(ray.direction, mul3s(hit.normal, k));
                    if (hit.material.roughness > 0.0f) {
                        float rx = 2.0f * xorwow_uniform(&rng) - 1.0f, ry = 2.0f * xorwow_uniform(&rng) - 1.0f, rz = 2.0f * xorwow_uniform(&rng) - 1.0f;
                        r = add3(r, mul3s(mk3(rx, ry, rz), hit.material.roughness));
                    }
                    r = normalize3(r);
                    ray = make_ray(add3(hit.hit_location, mul3s(r, (float)1e-4)), r);
                }
                acc = add3(acc, sample);
            }
            inv_spp = (float)spp;
            row[3 * x + 0] = f2u8(acc.x / inv_spp * 255);
            row[3 * x + 1] = f2u8(acc.y / inv_spp * 255);
            row[3 * x + 2] = f2u8(acc.z / inv_spp * 255);
            if (total_pops) total_pops[(size_t)y * (size_t)width + (size_t)x] = (int32_t)pops;
            st_rays += rays; st_pops += pops;
        }
    }
    if (stats4) { stats4[0] += st_rays; stats4[1] += st_pops; stats4[2] += st_hits; }
    return 0;
}

/* KAT access to the generator */
void orc_xorwow(unsigned long long seed, int n, uint32_t *out_bits, float *out_uniform)
{
    xorwow_t a, b; int i;
    xorwow_init(&a, seed); b = a;
    for (i = 0; i < n; i++) { if (out_bits) out_bits[i] = xorwow_next(&a); if (out_uniform) out_uniform[i] = xorwow_uniform(&b); }
}

/* ------------------------------------------------- KAT entry points (L0 math) */

void orc_normalize(const float *v, float *o) { f3 r = normalize3(mk3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
float orc_magnitude(const float *v) { return magnitude3(mk3(v[0], v[1], v[2])); }
void orc_euler2quat(const float *e, float *q) { f4 r = euler2quat(mk3(e[0], e[1], e[2])); q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w; }
void orc_apply_quat(const float *q, const float *v, float *o)
{ f4 qq; f3 r; qq.x = q[0]; qq.y = q[1]; qq.z = q[2]; qq.w = q[3]; r = apply_quat(qq, mk3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void orc_apply_euler(const float *e, const float *v, float *o) { f3 r = apply_euler(mk3(e[0], e[1], e[2]), mk3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void orc_invert_lre(const float *l, float *o) { lre_t a, r; memcpy(&a, l, sizeof a); r = invert_lre(a); memcpy(o, &r, sizeof r); }
void orc_apply_lre(const float *l, const float *v, float *o) { lre_t a; f3 r; memcpy(&a, l, sizeof a); r = apply_lre(a, mk3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
/* on_mouse, kernel.cu:112-139: state4 = {last_x, last_y, has_last, is_down}; cv::EVENT_MOUSEMOVE 0, LBUTTONDOWN 1, LBUTTONUP 4 */
void orc_on_mouse(float *pose6, int32_t *state4, int event, int x, int y)
{
    if (event == 1) state4[3] = 1;
    else if (event == 4) state4[3] = 0;
    else if (event == 0) {
        if (state4[2] && state4[3]) {
            int dx = x - state4[0], dy = y - state4[1];
            pose6[3] = (float)((double)pose6[3] + dx * 0.001);          /* pose->yaw += dx * 0.001 */
            pose6[4] = (float)((double)pose6[4] + dy * -0.001);         /* pose->pitch += dy * -0.001 */
        }
        state4[0] = x; state4[1] = y; state4[2] = 1;
    }
}
/* the key handling of kernel.cu:51-103 (commented out in the snapshot): returns 0 for 'q', 1 otherwise */
int orc_on_key(float *pose6, int key)
{
    f3 step, np; lre_t pose, inv;
    if (key == 'w') step = mk3(0.0f, 0.1f, 0.0f);
    else if (key == 's') step = mk3(0.0f, -0.1f, 0.0f);
    else if (key == 'a') step = mk3(-0.1f, 0.0f, 0.0f);
    else if (key == 'd') step = mk3(0.1f, 0.0f, 0.0f);
    else return key == 'q' ? 0 : 1;
    memcpy(&pose, pose6, sizeof pose);
    inv = invert_lre(pose);
    np = apply_lre(inv, step);
    pose6[0] = np.x; pose6[1] = np.y; pose6[2] = np.z;
    return 1;
}
void orc_lre2homo(const float *l, float *o16) { lre_t a; m44 h; memcpy(&a, l, sizeof a); h = lre2homo(a); memcpy(o16, &h, sizeof h); }
void orc_invert_intrinsic(const float *K9, float *o9) { m33 k, r; memcpy(&k, K9, sizeof k); r = invert_intrinsic(&k); memcpy(o9, &r, sizeof r); }
void orc_apply_matrix33(const float *K9, const float *v, float *o)                  /* utils.hpp:134-140 */
{ m33 k; f3 r; memcpy(&k, K9, sizeof k); r = apply_matrix33(&k, mk3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
/* Ray::Ray, Ray.hpp:17-23: what make_ray keeps (origin, direction, direction_inv) and the two constants the shading code
   assumes of a fresh ray (color 1,1,1 -- raycast.cu:226-244 multiplies by it -- and illumination 0) */
void orc_ray_ctor(const float *o, const float *d, float *out13)
{
    ray_t r = make_ray(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]));
    out13[0] = r.origin.x; out13[1] = r.origin.y; out13[2] = r.origin.z;
    out13[3] = r.direction.x; out13[4] = r.direction.y; out13[5] = r.direction.z;
    out13[6] = r.direction_inv.x; out13[7] = r.direction_inv.y; out13[8] = r.direction_inv.z;
    out13[9] = 1.0f; out13[10] = 1.0f; out13[11] = 1.0f; out13[12] = 0.0f;
}
float orc_aabb_ray_intersects(const float *bmin, const float *bmax, const float *o, const float *d)
{ ray_t r = make_ray(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2])); return aabb_ray_intersects(mk3(bmin[0], bmin[1], bmin[2]), mk3(bmax[0], bmax[1], bmax[2]), &r); }
/* tri18 layout as orc_mesh_from_triangles; out: isect xyz, uv xy */
void orc_tri_test(const float *tri18, const float *o, const float *d, float *out5)
{
    tri_t t; ray_t r = make_ray(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2])); f3 p; f2 uv;
    memcpy(&t, tri18, sizeof t);
    p = tri_ray_intersect(&t, &r);
    out5[0] = p.x; out5[1] = p.y; out5[2] = p.z;
    if (p.x == FLT_MAX) { out5[3] = FLT_MAX; out5[4] = FLT_MAX; return; }
    uv = tri_point_inside(&t, p);
    out5[3] = uv.x; out5[4] = uv.y;
}
void orc_tri_from_vertices(const float *abc9, float *tri18)
{ tri_t t = tri_from_vertices(mk3(abc9[0], abc9[1], abc9[2]), mk3(abc9[3], abc9[4], abc9[5]), mk3(abc9[6], abc9[7], abc9[8])); memcpy(tri18, &t, sizeof t); }
void orc_tri_center(const float *tri18, float *o) { tri_t t; f3 c; memcpy(&t, tri18, sizeof t); c = tri_center(&t); o[0] = c.x; o[1] = c.y; o[2] = c.z; }

/* FNV-1a-64 over the tight RGB bytes of a pitched image (SURVEY.md section 4 frame hashes) */
uint64_t orc_fnv1a64_image(const uint8_t *img, size_t pitch, int width, int height)
{
    uint64_t h = 1469598103934665603ULL; int x, y;
    for (y = 0; y < height; y++) {
        const uint8_t *row = img + (size_t)y * pitch;
        for (x = 0; x < width * 3; x++) { h ^= row[x]; h *= 1099511628211ULL; }
    }
    return h;
}
