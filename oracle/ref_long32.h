// ref_long32.h -- TEST INFRASTRUCTURE (ours).  Force-included in front of every reference
// translation unit built by oracle/Makefile so that utils.hpp's Q_rsqrt (utils.hpp:12-27)
// sees a 32-bit `long`, as on the MSVC/LLP64 target of CudaRaytracer.vcxproj.  utils.hpp is
// `#pragma once`, so later includes of it are no-ops.  No reference source is edited.
#pragma once
#include <cuda_runtime.h>
#include <cassert>
#include <iostream>
#include <math.h>
#include "transforms.hpp"
#define long int
#include "utils.hpp"
#undef long
