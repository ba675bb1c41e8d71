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

// mas_now_ticks(): free-running clock in 10-ns ticks (100-MHz REFCLK on the device, steady_clock in the emulation); the
// per-solve wall-clock budget `max_ms` (ilqr.hpp:84-90) is measured with it on the device, one start stamp per agent
#if defined(__HIP_DEVICE_COMPILE__)
#define MAS_NOW_TICKS() static_cast<unsigned long long>(__builtin_amdgcn_s_memrealtime())
#elif defined(MAS_HOST_EMULATION)
#include <chrono>
#define MAS_NOW_TICKS() static_cast<unsigned long long>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10)
#else
#define MAS_NOW_TICKS() 0ull
#endif
constexpr double kTicksPerMs = 1e5;

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

// Division by a value that stays the same for a whole kernel (a finite-difference step 2 eps, eps^2, 4 eps^2; a wheelbase).
// The compiler's IEEE division is eleven dependent instructions (scale, reciprocal seed, two Newton steps, quotient, residual,
// correction, fix-up); with the divisor's correctly rounded reciprocal y = RN(1 / b) formed once -- at compile time for the
// finite-difference constants -- the quotient needs only its last three steps,
//     q = RN(p y),   r = p - b q  (exact in one fma),   RN(q + r y) = RN(p / b)
// (Markstein's correction step: q is within one ulp of p / b, so r is exact and the fused correction rounds the true quotient
// perturbed by less than 2^-53 ulp -- the same last step v_div_fmas_f64 takes in the compiler's own expansion), followed by the
// hardware's v_div_fixup_f64 for the special operands (zero, infinity, NaN).  Bit-identical to `/` wherever the quotient is normal
// and the residual exactly representable: |p| >= 2^-969 (r is a multiple of 2^(e_p - 104)) and |p / b| >= 2^-1022, or p == 0 --
// checked against the device's own division on 10^9 random operands and at that edge by tests/hip/div_by_check.hip, and by every
// parity test against the CPU oracle, which divides with `/`.  Not guarded at run time: include/mas_hip.h states the limit.
// The CPU builds (oracle, host emulation) use `/`.
struct InvDiv { double b, y; };
MAS_HD constexpr InvDiv make_inv_div(double b) { return InvDiv{b, 1.0 / b}; }
MAS_HD double div_by(double p, const InvDiv& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double q = p * d.y;
  const double r = __builtin_fma(-q, d.b, p);
  return __builtin_amdgcn_div_fixup(__builtin_fma(r, d.y, q), d.b, p);
#else
  return p / d.b;
#endif
}

