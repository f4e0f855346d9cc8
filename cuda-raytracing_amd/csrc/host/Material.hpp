// Material.hpp -- surface description (Material.hpp:6-43).  Only albedo and the texture are read
// by the path today (roughness / metallic / illumination are dead in the reference as well).
// The texture lives on the host as tight BGR bytes until Scene::upload_to_device copies it;
// upload_texture() decodes PNG, baseline JPEG and binary PPM itself (ImageIO.hpp) because the image has no OpenCV.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <hip/hip_vector_types.h>

struct Material {
    float roughness;
    float3 albedo;
    float metallic;
    float illumination;
    std::vector<uint8_t> texture;      // BGR, row-major, pitch = texture_width * 3
    int texture_width = 0;
    int texture_height = 0;

    Material() : roughness(0.0f), albedo(make_float3(1.0f, 1.0f, 1.0f)), metallic(0.0f), illumination(0.0f) {}
    // Material.hpp:21-27: a device copy of this material alone (rt_hip.h RtMaterialDesc without its texture: the texture
    // travels with Scene::upload_to_device).  The returned pointer is DEVICE memory behind the reference's return type: an
    // opaque handle for rt_free(), never to be dereferenced on the host -- exactly what the reference's pointer is.
    Material* to_device() const;
    bool upload_texture(const std::string& path);                               // PNG / JPEG / P6 PPM -> BGR
    void set_texture_bgr(const uint8_t* bgr, int width, int height, size_t pitch);
};
