// utils.hpp -- float3/float2 helpers with the reference's names (utils.hpp:12-160), adapters
// over ../rt_math.h so host code and kernels share one bit-exact implementation.
#pragma once
#include <hip/hip_vector_types.h>
#include <cmath>
#include "transforms.hpp"

using namespace transforms;

inline float Q_rsqrt(float number) { return rt::q_rsqrt(number); }
inline float magnitude(float3 v) { return rt::magnitude(rt::v3(v.x, v.y, v.z)); }
inline float magnitude(float2 v) { return sqrtf(v.x * v.x + v.y * v.y); }
inline float inv_magnitude(float3 v) { return rt::q_rsqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline float3 normalize(float3 v) { rt::V3 r = rt::normalize(rt::v3(v.x, v.y, v.z)); return make_float3(r.x, r.y, r.z); }
inline float3 cross(float3 a, float3 b) { rt::V3 r = rt::cross(rt::v3(a.x, a.y, a.z), rt::v3(b.x, b.y, b.z)); return make_float3(r.x, r.y, r.z); }
inline float dot(float3 a, float3 b) { return rt::dot(rt::v3(a.x, a.y, a.z), rt::v3(b.x, b.y, b.z)); }
inline float3 f3_min(float3 a, float3 b) { return make_float3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
inline float3 f3_max(float3 a, float3 b) { return make_float3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
// float3 / float2 arithmetic operators: HIP's vector types already provide the component-wise
// + - * / of utils.hpp:57-115 on the host, with the same single-rounding fp32 semantics.

template <typename T> void cu_swap(T& a, T& b) { T t = a; a = b; b = t; }

inline float4 apply_matrix(const float4x4& m, const float4& v)
{
    float4 r;
    r.x = m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z + m.m[0][3] * v.w;
    r.y = m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z + m.m[1][3] * v.w;
    r.z = m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z + m.m[2][3] * v.w;
    r.w = m.m[3][0] * v.x + m.m[3][1] * v.y + m.m[3][2] * v.z + m.m[3][3] * v.w;
    return r;
}
inline float3 apply_matrix(const float3x3& m, const float3& v) { return apply_rotmat(m, v); }
inline float3x3 invert_intrinsic(const float3x3& K) { return detail::out(rt::invert_intrinsic(detail::in(K))); }
