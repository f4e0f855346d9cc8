// rt_math.h -- L0 math of the raycast path, shared by the HIP kernels and the host library.
//
// Every function restates one reference function bit-for-bit (fp32, written association
// order, no contraction: all translation units that include this are built with
// -ffp-contract=off).  Reference lines are relative to /root/reference/CudaRaytracer/.
// Parity is tested against the oracle (tests/test_host_logic.py::test_host_math_golden and ::test_atanf_restatement_matches_libm, tests/test_gpu_parity.py).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

namespace rt {

struct V3 { float x, y, z; };
struct V2 { float x, y; };
struct Q4 { float x, y, z, w; };            // quaternion stored (w, x, y, z) in .x .y .z .w like transforms.hpp:157-162
struct Pose { float x, y, z, yaw, pitch, roll; };   // `lre`, transforms.hpp:10-14
struct M33 { float m[3][3]; };
struct M44 { float m[4][4]; };

RT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }      // utils.hpp:57
RT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }      // utils.hpp:65
RT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }      // utils.hpp:69
RT_HD V3 operator*(V3 a, float b) { return v3(a.x * b, a.y * b, a.z * b); }         // utils.hpp:73
RT_HD V3 operator*(float b, V3 a) { return v3(a.x * b, a.y * b, a.z * b); }         // utils.hpp:77
RT_HD V3 operator/(V3 a, float b) { return v3(a.x / b, a.y / b, a.z / b); }         // utils.hpp:81
RT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           // utils.hpp:53
RT_HD V3 cross(V3 a, V3 b)                                                          // utils.hpp:49
{ return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

RT_HD int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
RT_HD float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

// utils.hpp:12-27 with a 32-bit integer
RT_HD float q_rsqrt(float number)
{
    float x2 = number * 0.5F;
    int32_t i = f2i(number);
    i = 0x5f3759df - (i >> 1);
    float y = i2f(i);
    y = y * (1.5F - (x2 * y * y));
    return y;
}
RT_HD float magnitude(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }    // utils.hpp:29
RT_HD V3 normalize(V3 v)                                                            // utils.hpp:37-47
{
    float inv_mag = q_rsqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return v3(v.x * inv_mag, v.y * inv_mag, v.z * inv_mag);
}

// transforms.hpp:165-176 (no normalisation of q)
RT_HD V3 apply_quat(Q4 q, V3 v)
{
    float a = -v.x * q.y - v.y * q.z - v.z * q.w;
    float b = v.x * q.x + v.y * q.w - v.z * q.z;
    float c = v.y * q.x + v.z * q.y - v.x * q.w;
    float d = v.z * q.x + v.x * q.z - v.y * q.y;
    return v3(q.x * b - q.y * a - q.z * d + q.w * c,
              q.x * c - q.z * a - q.w * b + q.y * d,
              q.x * d - q.w * a - q.y * c + q.z * b);
}

// atanf as glibc 2.35 computes it (the fdlibm algorithm: reduction to |x| < 7/16 and an
// odd/even split degree-11 polynomial, all in fp32 with IEEE + - * /).  The reference calls
// libm's atan(float) per pixel (raycast.cu:170); ROCm's device atanf differs from glibc by
// ULPs, so the device evaluates this restatement instead.  tests/test_host_logic.py checks
// it against libm atanf over every float in the range ray generation can produce.
RT_HD float atanf_fdlibm(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                          9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                          4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    int32_t hx = f2i(x), ix = hx & 0x7fffffff, id;
    if (ix >= 0x4c000000) {                         // |x| >= 2^25, inf, NaN
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                          // |x| < 0.4375
        if (ix < 0x31000000) return x;              // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                      // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x, w = z * z;
    float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}

// XORWOW generator seeded like cuRAND's curand_init(seed, 0, 0) (the reference initialises one per pixel,
// raycast.cu:190-193, and never samples it; the extension modes do).  cuRAND is a third-party dependency that is
// not part of the reference tree: restated from its published algorithm (Marsaglia xorwow + cuRAND's seed scramble).
struct Xorwow { uint32_t v[5]; uint32_t d; };
RT_HD void xorwow_init(Xorwow& st, unsigned long long seed)
{
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st.d = 6615241u + t1 + t0;
    st.v[0] = 123456789u + t0;
    st.v[1] = 362436069u ^ t0;
    st.v[2] = 521288629u + t1;
    st.v[3] = 88675123u ^ t1;
    st.v[4] = 5783321u + t0;
}
RT_HD uint32_t xorwow_next(Xorwow& st)
{
    uint32_t t = st.v[0] ^ (st.v[0] >> 2);
    st.v[0] = st.v[1]; st.v[1] = st.v[2]; st.v[2] = st.v[3]; st.v[3] = st.v[4];
    st.v[4] = (st.v[4] ^ (st.v[4] << 4)) ^ (t ^ (t << 1));
    st.d += 362437u;
    return st.v[4] + st.d;
}
RT_HD float xorwow_uniform(Xorwow& st) { return (float)xorwow_next(st) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }   // (0, 1]

// ---- host-only pieces (libm sinf/cosf/atan2f/asinf: evaluated once per pose on the host,
//      never per ray; the kernels receive the resulting quaternions) -----------------------
inline Q4 euler2quat(V3 e)                                                          // transforms.hpp:148-163
{
    float sy = sinf((float)(e.x * 0.5)), cy = cosf((float)(e.x * 0.5));
    float sp = sinf((float)(e.y * 0.5)), cp = cosf((float)(e.y * 0.5));
    float sr = sinf((float)(e.z * 0.5)), cr = cosf((float)(e.z * 0.5));
    Q4 q;
    q.x = sy * sp * sr + cy * cp * cr;
    q.y = cy * sp * cr + sy * cp * sr;
    q.z = -sy * sp * cr + cy * cp * sr;
    q.w = cy * sp * sr - sy * cp * cr;
    return q;
}
inline V3 apply_euler(V3 e, V3 v) { return apply_quat(euler2quat(e), v); }          // transforms.hpp:219
inline V3 apply_lre(Pose l, V3 v)                                                   // transforms.hpp:223-226
{ return apply_euler(v3(l.yaw, l.pitch, l.roll), v3(v.x - l.x, v.y - l.y, v.z - l.z)); }

inline V3 apply_rotmat(const M33& r, V3 v)                                          // transforms.hpp:63-69
{
    return v3(r.m[0][0] * v.x + r.m[0][1] * v.y + r.m[0][2] * v.z,
              r.m[1][0] * v.x + r.m[1][1] * v.y + r.m[1][2] * v.z,
              r.m[2][0] * v.x + r.m[2][1] * v.y + r.m[2][2] * v.z);
}
inline M33 invert_rotmat(const M33& r)                                              // transforms.hpp:55-61
{ M33 o; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[i][j] = r.m[j][i]; return o; }
inline M33 euler2rotmat(V3 e)                                                       // transforms.hpp:129-144
{
    float sy = sinf(e.x), cy = cosf(e.x), sp = sinf(e.y), cp = cosf(e.y), sr = sinf(e.z), cr = cosf(e.z);
    M33 o;
    o.m[0][0] = cr * cy + sr * sp * sy; o.m[0][1] = -cr * sy + sr * sp * cy; o.m[0][2] = -sr * cp;
    o.m[1][0] = cp * sy;                o.m[1][1] = cp * cy;                 o.m[1][2] = sp;
    o.m[2][0] = sr * cy - cr * sp * sy; o.m[2][1] = -sr * sy - cr * sp * cy; o.m[2][2] = cr * cp;
    return o;
}
inline V3 rotmat2euler(const M33& r)                                                // transforms.hpp:119-126
{
    float a = r.m[1][2];
    if (a > 1) a = 1; else if (a < -1) a = -1;
    return v3(atan2f(r.m[1][0], r.m[1][1]), asinf(a), atan2f(-r.m[0][2], r.m[2][2]));
}
inline M44 lre2homo(Pose p)                                                         // transforms.hpp:178-193
{
    M33 R = euler2rotmat(v3(p.yaw, p.pitch, p.roll));
    V3 rs = apply_rotmat(R, v3(-p.x, -p.y, -p.z));
    M44 o;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[i][j] = R.m[i][j];
    o.m[0][3] = rs.x; o.m[1][3] = rs.y; o.m[2][3] = rs.z;
    o.m[3][0] = o.m[3][1] = o.m[3][2] = 0.0f; o.m[3][3] = 1.0f;
    return o;
}
inline M44 invert_homo(const M44& H)                                                // transforms.hpp:72-96
{
    M33 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = H.m[i][j];
    M33 Ri = invert_rotmat(R);
    V3 ti = apply_rotmat(Ri, v3(-H.m[0][3], -H.m[1][3], -H.m[2][3]));
    M44 o;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[i][j] = Ri.m[i][j];
    o.m[0][3] = ti.x; o.m[1][3] = ti.y; o.m[2][3] = ti.z;
    o.m[3][0] = o.m[3][1] = o.m[3][2] = 0.0f; o.m[3][3] = 1.0f;
    return o;
}
inline Pose homo2lre(const M44& H)                                                  // transforms.hpp:195-216
{
    M33 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.m[i][j] = H.m[i][j];
    V3 e = rotmat2euler(R);
    V3 s = apply_rotmat(invert_rotmat(R), v3(H.m[0][3], H.m[1][3], H.m[2][3]));
    Pose o; o.x = -s.x; o.y = -s.y; o.z = -s.z; o.yaw = e.x; o.pitch = e.y; o.roll = e.z;
    return o;
}
inline M44 matmul(const M44& a, const M44& b)                                       // transforms.hpp:98-111
{
    M44 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        r.m[i][j] = 0.0f;
        for (int k = 0; k < 4; k++) r.m[i][j] += a.m[i][k] * b.m[k][j];
    }
    return r;
}
inline Pose invert_lre(Pose p) { return homo2lre(invert_homo(lre2homo(p))); }       // transforms.hpp:232-235
inline Pose compose_lre(Pose a, Pose b) { return homo2lre(matmul(lre2homo(b), lre2homo(a))); }  // transforms.hpp:113-116,228-230
inline M33 invert_intrinsic(const M33& K)                                           // utils.hpp:142-160
{
    float fx_inv = 1.0f / K.m[0][0], fy_inv = 1.0f / K.m[1][1];
    float cx = K.m[0][2], cy = K.m[1][2];
    M33 o;
    o.m[0][0] = fx_inv; o.m[0][1] = 0.0f;   o.m[0][2] = -cx * fx_inv;
    o.m[1][0] = 0.0f;   o.m[1][1] = fy_inv; o.m[1][2] = -cy * fy_inv;
    o.m[2][0] = 0.0f;   o.m[2][1] = 0.0f;   o.m[2][2] = 1.0f;
    return o;
}

}  // namespace rt
