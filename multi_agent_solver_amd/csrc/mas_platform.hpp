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

// MAS_OPAQUE32(v): hides a loop-invariant 32-bit byte offset from the optimiser inside a loop.  Without it
// LICM hoists the zero-extension to 64 bits out of the loop and instruction selection no longer sees
// "uniform base + zext(32-bit lane offset)", so every access pays a v_lshl_add_u64 (and a hazard nop)
// instead of using the scalar-base addressing form  global_load v, v_off32, s[base:base+1].
// MAS_VGPR64(v): pins a loop-invariant double (a kernel argument) in a vector register pair before a loop, so a
// select that uses it needs no per-use copy from scalar registers.
#if defined(__HIP_DEVICE_COMPILE__)
#define MAS_OPAQUE32(v) asm volatile("" : "+v"(v))
#define MAS_VGPR64(v) asm volatile("" : "+v"(v))
#else
#define MAS_OPAQUE32(v) ((void)0)
#define MAS_VGPR64(v) ((void)0)
#endif
