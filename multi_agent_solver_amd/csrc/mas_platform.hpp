// mas_platform.hpp -- function qualifiers for the lane-level math.
//
// The product is built by hipcc for gfx950 (MAS_HD = __host__ __device__).  The same lane
// functions are also compiled by g++ into tests/hostcheck (MAS_HOST_EMULATION), a TEST-ONLY
// artefact that replays the kernels' bodies lane by lane on the CPU so the launch logic and
// the arithmetic can be checked against the oracle without a GPU.  libmas_hip.so never
// contains that path: without a HIP device it reports MAS_HIP_ERR_NO_DEVICE.
#pragma once

#if defined(MAS_HOST_EMULATION)
#define MAS_HD inline
#define MAS_D inline
#define MAS_UNROLL
#else
#include <hip/hip_runtime.h>
#define MAS_HD __host__ __device__ __forceinline__
#define MAS_D __device__ __forceinline__
#define MAS_UNROLL _Pragma("unroll")
#endif
