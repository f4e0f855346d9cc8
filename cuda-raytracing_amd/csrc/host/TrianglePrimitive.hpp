// TrianglePrimitive.hpp -- host triangle record with the reference's layout and constructors
// (TrianglePrimitive.hpp:8-60).  The per-ray tests live in the HIP kernels; the two host
// methods below exist for API compatibility and unit tests.
#pragma once
#include <cfloat>
#include "utils.hpp"

struct Ray {                                       // Ray.hpp:5-24 (fields the path reads)
    float3 origin, direction, direction_inv;
    Ray(float3 o, float3 d) : origin(o), direction(d) { direction_inv = make_float3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
};

struct TrianglePrimitive {
    float3 vertices[3];
    float3 normal;
    float2 uv_coords[3];                           // zero unless given (the reference leaves them uninitialised)

    TrianglePrimitive() : normal(make_float3(0.0f, 0.0f, 0.0f)) { zero(); }
    // normal from the winding, through the fast inverse square root (TrianglePrimitive.hpp:15-23)
    TrianglePrimitive(float3 a, float3 b, float3 c) { zero(); set(a, b, c); normal = normalize(cross(sub(b, a), sub(c, a))); }
    TrianglePrimitive(float3 a, float3 b, float3 c, float3 n) : normal(n) { zero(); set(a, b, c); }
    TrianglePrimitive(float3 a, float3 b, float3 c, float3 n, float2 uv_a, float2 uv_b, float2 uv_c) : normal(n)
    { set(a, b, c); uv_coords[0] = uv_a; uv_coords[1] = uv_b; uv_coords[2] = uv_c; }

    float3 center() const                                                                  // :81-83
    {
        const float3 &a = vertices[0], &b = vertices[1], &c = vertices[2];
        return make_float3(((a.x + b.x) + c.x) / 3.0f, ((a.y + b.y) + c.y) / 3.0f, ((a.z + b.z) + c.z) / 3.0f);
    }

    // plane hit point or (FLT_MAX)^3, TrianglePrimitive.hpp:62-79
    float3 ray_intersect(const Ray& ray) const
    {
        const float3 miss = make_float3(FLT_MAX, FLT_MAX, FLT_MAX);
        float denom = dot(ray.direction, normal);
        if ((double)fabsf(denom) < 1e-6) return miss;
        float t = dot(sub(vertices[0], ray.origin), normal) / denom;
        if (t < 0.0f) return miss;
        return make_float3(ray.origin.x + t * ray.direction.x, ray.origin.y + t * ray.direction.y, ray.origin.z + t * ray.direction.z);
    }
    // interpolated uv or (FLT_MAX)^2, TrianglePrimitive.hpp:151-185
    float2 point_inside(const float3& point) const
    {
        float3 e0 = sub(vertices[2], vertices[0]), e1 = sub(vertices[1], vertices[0]), e2 = sub(point, vertices[0]);
        float d00 = dot(e0, e0), d01 = dot(e0, e1), d02 = dot(e0, e2), d11 = dot(e1, e1), d12 = dot(e1, e2);
        float inv = 1.0f / (d00 * d11 - d01 * d01);
        float u = (d11 * d02 - d01 * d12) * inv, v = (d00 * d12 - d01 * d02) * inv;
        if ((u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f)) {
            float w = 1.0f - u - v;
            return make_float2((w * uv_coords[0].x + v * uv_coords[1].x) + u * uv_coords[2].x,
                               (w * uv_coords[0].y + v * uv_coords[1].y) + u * uv_coords[2].y);
        }
        return make_float2(FLT_MAX, FLT_MAX);
    }

    static float3 sub(const float3& a, const float3& b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }

private:
    void zero() { for (int i = 0; i < 3; i++) uv_coords[i] = make_float2(0.0f, 0.0f); }
    void set(float3 a, float3 b, float3 c) { vertices[0] = a; vertices[1] = b; vertices[2] = c; }
};
static_assert(sizeof(TrianglePrimitive) == 72, "TrianglePrimitive must stay 72 bytes (3 verts, normal, 3 uv)");
