// OBJLoader.hpp -- Wavefront OBJ -> MeshPrimitive with the reference's semantics
// (OBJLoader.hpp:15-179): `v` and `vt` records, `f` with v, v/vt or v/vt/vn tokens, fan
// triangulation (0, i, i+1), face normal = normalize(cross(v_i - v_0, v_{i+1} - v_0)) through the
// fast inverse square root; `vn` is parsed and ignored.
#pragma once
#include <string>
#include <vector>
#include "MeshPrimitive.h"

namespace OBJLoader {
// Triangles of the file; returns false (and fills *error) when the file cannot be opened or a
// face is malformed / out of range.
// lenient = false reproduces the reference (a `v//vn` token or a negative index is an error, H11);
// lenient = true additionally accepts `v//vn` (no texture index) and negative (relative) indices as the OBJ
// format defines them: -1 is the most recent `v` / `vt` record before the face line.
bool parse(const std::string& fp, std::vector<TrianglePrimitive>& triangles, std::string* error, bool lenient = false);
// The float scanner of parse() on one token (std::stof semantics, correctly rounded like strtof; exposed for tests)
bool scan_float_token(const char* begin, const char* end, float& out);
// parse + MeshPrimitive::for_device_build: no host tree, the GPU builds it at Scene::upload_to_device; throws like load_lenient
MeshPrimitive load_for_device(std::string fp, bool lenient = false);
// parse(lenient = true) + MeshPrimitive; throws std::runtime_error on failure
MeshPrimitive load_lenient(std::string fp);
// Reference behaviour: prints the progress lines, and on an unreadable file prints
// "Could not open file" and exit(1)s (OBJLoader.hpp:23-27); malformed faces throw std::runtime_error
// (the reference throws std::invalid_argument from stoi).
MeshPrimitive load(std::string fp);
}
