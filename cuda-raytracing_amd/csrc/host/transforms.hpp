// transforms.hpp -- host pose algebra with the reference's names (transforms.hpp:8-235 of
// AFIDclan/cuda-raytracing), implemented as thin adapters over ../rt_math.h.  Host-only: the
// kernels never evaluate sinf/cosf; they receive quaternions computed here once per pose.
#pragma once
#include <hip/hip_vector_types.h>
#include <iostream>
#include "../rt_math.h"

namespace transforms {

struct lre {                                       // transforms.hpp:10-14
    float x, y, z, yaw, pitch, roll;
    lre() : x(0.0f), y(0.0f), z(0.0f), yaw(0.0f), pitch(0.0f), roll(0.0f) {}
};
struct float4x4 { float m[4][4]; };
struct float3x3 { float m[3][3]; };

namespace detail {
inline rt::V3 in(float3 v) { return rt::v3(v.x, v.y, v.z); }
inline float3 out(rt::V3 v) { return make_float3(v.x, v.y, v.z); }
inline rt::Pose in(const lre& l) { rt::Pose p; p.x = l.x; p.y = l.y; p.z = l.z; p.yaw = l.yaw; p.pitch = l.pitch; p.roll = l.roll; return p; }
inline lre out(const rt::Pose& p) { lre l; l.x = p.x; l.y = p.y; l.z = p.z; l.yaw = p.yaw; l.pitch = p.pitch; l.roll = p.roll; return l; }
inline rt::M33 in(const float3x3& a) { rt::M33 o; memcpy(&o, &a, sizeof o); return o; }
inline float3x3 out(const rt::M33& a) { float3x3 o; memcpy(&o, &a, sizeof o); return o; }
inline rt::M44 in(const float4x4& a) { rt::M44 o; memcpy(&o, &a, sizeof o); return o; }
inline float4x4 out(const rt::M44& a) { float4x4 o; memcpy(&o, &a, sizeof o); return o; }
inline rt::Q4 in(float4 q) { rt::Q4 o; o.x = q.x; o.y = q.y; o.z = q.z; o.w = q.w; return o; }
}  // namespace detail

inline float3x3 invert_rotmat(const float3x3& r) { return detail::out(rt::invert_rotmat(detail::in(r))); }
inline float3 apply_rotmat(const float3x3& r, const float3& v) { return detail::out(rt::apply_rotmat(detail::in(r), detail::in(v))); }
inline float4x4 invert_homo(const float4x4& H) { return detail::out(rt::invert_homo(detail::in(H))); }
inline float4x4 matmul(float4x4 a, float4x4 b) { return detail::out(rt::matmul(detail::in(a), detail::in(b))); }
inline float4x4 compose_homo(float4x4 H1, float4x4 H2) { return matmul(H2, H1); }
inline float3 rotmat2euler(float3x3 r) { return detail::out(rt::rotmat2euler(detail::in(r))); }
inline float3x3 euler2rotmat(float3 e) { return detail::out(rt::euler2rotmat(detail::in(e))); }
inline float4 euler2quat(float3 e) { rt::Q4 q = rt::euler2quat(detail::in(e)); return make_float4(q.x, q.y, q.z, q.w); }
inline float3 apply_quat(float4 q, float3 v) { return detail::out(rt::apply_quat(detail::in(q), detail::in(v))); }
inline float4x4 lre2homo(lre v) { return detail::out(rt::lre2homo(detail::in(v))); }
inline lre homo2lre(float4x4 H) { return detail::out(rt::homo2lre(detail::in(H))); }
inline float3 apply_euler(float3 e, float3 v) { return detail::out(rt::apply_euler(detail::in(e), detail::in(v))); }
inline float3 apply_lre(lre l, float3 v) { return detail::out(rt::apply_lre(detail::in(l), detail::in(v))); }
inline lre compose_lre(lre a, lre b) { return detail::out(rt::compose_lre(detail::in(a), detail::in(b))); }
inline lre invert_lre(lre l) { return detail::out(rt::invert_lre(detail::in(l))); }

inline void print(const float4x4& m) { for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) std::cout << m.m[i][j] << ' '; std::cout << std::endl; } }
inline void print(const float3x3& m) { for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) std::cout << m.m[i][j] << ' '; std::cout << std::endl; } }
inline void print(const lre& l) { std::cout << l.x << ", " << l.y << ", " << l.z << ", " << l.yaw << ", " << l.pitch << ", " << l.roll << std::endl; }
inline void print(const float3 v) { std::cout << v.x << ", " << v.y << ", " << v.z << std::endl; }
inline void print(const float4 v) { std::cout << v.x << ", " << v.y << ", " << v.z << ", " << v.w << std::endl; }

// transforms.hpp:238-290: the reference's self-check of the pose algebra (commented out at the top of its main(),
// kernel.cu:142): a pose, its homogeneous matrix, the inverse, the inverse pose, and a vector taken through them
inline void test_all()
{
    float3 v = make_float3(6, -2, 5);
    lre l = lre();
    l.y = 10;
    l.pitch = 0.5;
    std::cout << "lre in: \n";
    print(l);
    float4x4 homo = lre2homo(l);
    std::cout << "homo: \n";
    print(homo);
    float4x4 homo_inv = invert_homo(homo);
    std::cout << "inverted: \n";
    print(homo_inv);
    lre l_inv = homo2lre(homo_inv);
    std::cout << "lre inv: \n";
    print(l_inv);
    float3 subtracted = make_float3(v.x - l_inv.x, v.y - l_inv.y, v.z - l_inv.z);
    std::cout << "subtracted: \n";
    print(subtracted);
    float4 quat = euler2quat(make_float3(l_inv.yaw, l_inv.pitch, l_inv.roll));
    std::cout << "Quat: \n";
    print(quat);
    float3 v_appl = apply_quat(quat, subtracted);
    std::cout << "vec out: " << v_appl.x << ", " << v_appl.y << ", " << v_appl.z << std::endl;
}

}  // namespace transforms
