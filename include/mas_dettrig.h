/*
 * mas_dettrig.h -- portable, bit-reproducible sin / cos / tan for FP64.
 *
 * WHY.  The reference's iLQR is numerically chaotic on its own benchmark workload: a 1-ulp
 * change of one input moves converged single-track costs by 3 % .. 170 % at H=100
 * (tests/test_oracle_sensitivity.py reproduces this on the CPU oracle alone), because the
 * second-difference Hessians amplify rounding noise by 1/eps^2 = 1e10 and the x10
 * regularisation ladder is a discontinuous function of its input.  glibc and ROCm's ocml
 * disagree in the last bit of sin/cos/tan now and then, so no GPU implementation can agree
 * with a glibc-linked CPU run to any useful tolerance -- unless both sides evaluate the
 * transcendental functions with the SAME sequence of IEEE-754 operations.  This header is
 * that sequence: only + - * / , fma, rint and integer masks -- every one of them a correctly
 * rounded IEEE-754-2008 operation -- no tables, no libm transcendental.  The fused multiply-adds
 * are EXPLICIT (v_fma_f64 on gfx950; C99 fma() on the CPU, which glibc evaluates exactly, in
 * hardware where the CPU has it); compiled with -ffp-contract=off, so that the compiler fuses
 * nothing on its own, the header returns identical bits under gcc on x86-64 and hipcc on gfx950.
 * (Round 1 of this header used no fma at all; the Horner steps then cost two dependent
 * instructions each, and sin/cos/tan were two thirds of the line-search rollout's instructions.
 * The reference calls glibc's sin/cos/tan, which use fma internally where the CPU has it: how a
 * libm rounds inside is not part of the reference's arithmetic, only its <= 1 ulp result is.)
 *
 * WHAT.  Three-stage Cody-Waite reduction by pi/2 (33-bit pieces, exact products for
 * |n| < 2^20, i.e. |x| < 1.6e6), then degree-17 / degree-16 Taylor polynomials on
 * [-pi/4, pi/4] with a first-order correction for the reduction tail; tan = sin / cos.
 * Accuracy against glibc (tests/test_dettrig.py): sin, cos <= 1 ulp, tan <= 3 ulp.
 * Outside the reduction range, and for NaN/Inf, the result is NaN (no workload in this
 * repository gets there; a silent switch to the platform libm would break reproducibility).
 *
 * Both the product (multi_agent_solver_amd/csrc) and the CPU oracle (oracle/, in its
 * "det" trig mode) include this file, exactly as both would link one libm.
 * Constants: tools/gen_dettrig_constants.py (exact rational / 120-digit decimal arithmetic).
 */
#ifndef MAS_DETTRIG_H
#define MAS_DETTRIG_H

#include <math.h>

#ifndef MAS_DT_FN
#if defined(__HIPCC__)
#define MAS_DT_FN static __host__ __device__ __forceinline__
#else
#define MAS_DT_FN static inline
#endif
#endif

#define MAS_DT_MAX_ARG 1.6e6
#define MAS_DT_FMA(a, b, c) __builtin_fma((a), (b), (c))   /* one rounding: RN(a*b + c) */

/* reduce x to y0 + y1 in [-pi/4, pi/4] (y1 = tail), return the quadrant n mod 4 */
/* quadrant = (int)fn & 3.  In range (|x| <= MAS_DT_MAX_ARG) fn is an integer below 2^21.  Out of range the result is NaN whatever the
 * quadrant, but the quadrant still picks that NaN's sign: the device's conversion (v_cvt_i32_f64) SATURATES and maps NaN to 0,
 * while the C conversion is undefined there (x86 yields INT_MIN) -- so the CPU builds spell the saturation out and both sides
 * return the same bits for every input.  No instruction is added on the device. */
MAS_DT_FN int mas_dt_quadrant(double fn) {
#if defined(__HIP_DEVICE_COMPILE__)
  return ((int)fn) & 3;
#else
  int n;
  if (fn != fn) n = 0;
  else if (fn >= 2147483647.0) n = 2147483647;
  else if (fn <= -2147483648.0) n = (-2147483647 - 1);
  else n = (int)fn;
  return n & 3;
#endif
}

MAS_DT_FN int mas_dt_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 0x1.45f306dc9c883p-1;
  const double pio2_1 = 0x1.921fb54400000p+0, pio2_1t = 0x1.0b4611a626331p-34;
  const double pio2_2 = 0x1.0b4611a600000p-34, pio2_2t = 0x1.3198a2e037073p-69;
  const double pio2_3 = 0x1.3198a2e000000p-69, pio2_3t = 0x1.b839a252049c1p-104;
  const double fn = rint(x * invpio2);
  double r = MAS_DT_FMA(-fn, pio2_1, x);          /* fn * pio2_1 is exact (33-bit piece): same bits as x - fn*pio2_1 */
  double w, t = r;
  (void)pio2_1t;
  w = fn * pio2_2; r = t - w; w = MAS_DT_FMA(fn, pio2_2t, -((t - r) - w));
  t = r;
  w = fn * pio2_3; r = t - w; w = MAS_DT_FMA(fn, pio2_3t, -((t - r) - w));
  *y0 = r - w;
  *y1 = (r - *y0) - w;
  return mas_dt_quadrant(fn);
}

/* sin(x + y) for |x| <= pi/4, |y| << |x| */
MAS_DT_FN double mas_dt_ksin(double x, double y) {
  const double S3 = -0x1.5555555555555p-3, S5 = 0x1.1111111111111p-7, S7 = -0x1.a01a01a01a01ap-13,
               S9 = 0x1.71de3a556c734p-19, S11 = -0x1.ae64567f544e4p-26, S13 = 0x1.6124613a86d09p-33,
               S15 = -0x1.ae7f3e733b81fp-41, S17 = 0x1.952c77030ad4ap-49;
  const double z = x * x;
  double p = S17;
  p = MAS_DT_FMA(p, z, S15); p = MAS_DT_FMA(p, z, S13); p = MAS_DT_FMA(p, z, S11); p = MAS_DT_FMA(p, z, S9);
  p = MAS_DT_FMA(p, z, S7); p = MAS_DT_FMA(p, z, S5);
  const double v = z * x;
  /* x + (v*(S3 + z*p) + y*(1 - z/2)) */
  return x + MAS_DT_FMA(v, MAS_DT_FMA(z, p, S3), y * (1.0 - 0.5 * z));
}

/* cos(x + y) for |x| <= pi/4 */
MAS_DT_FN double mas_dt_kcos(double x, double y) {
  const double C4 = 0x1.5555555555555p-5, C6 = -0x1.6c16c16c16c17p-10, C8 = 0x1.a01a01a01a01ap-16,
               C10 = -0x1.27e4fb7789f5cp-22, C12 = 0x1.1eed8eff8d898p-29, C14 = -0x1.93974a8c07c9dp-37,
               C16 = 0x1.ae7f3e733b81fp-45;
  const double z = x * x;
  double q = C16;
  q = MAS_DT_FMA(q, z, C14); q = MAS_DT_FMA(q, z, C12); q = MAS_DT_FMA(q, z, C10); q = MAS_DT_FMA(q, z, C8);
  q = MAS_DT_FMA(q, z, C6); q = MAS_DT_FMA(q, z, C4);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + MAS_DT_FMA(z * z, q, -(x * y)));
}

/* Arguments outside the reduction range (|x| > 1.6e6), NaN and Inf give NaN: a loud, and still
 * reproducible, answer.  The functions are branch-free so that a compiler can merge identical
 * sub-expressions across the many perturbed evaluations of a finite-difference sweep, and they are written
 * for a machine on which every instruction of a dependent chain costs the same (a lone GPU wavefront):
 *  - an out-of-range argument turns the reduced argument into NaN by ADDING a poison term, -0.0 for an argument in
 *    range -- y + (-0.0) is y bit for bit, for every y including both zeros -- so no result needs a select of its
 *    own (its quadrant, taken from a conversion that saturates, only picks the sign and payload of that NaN);
 *  - a quadrant's sign change is an exclusive-or on the sign bit (what negation is), driven by integer
 *    arithmetic on n instead of compare-and-select. */
MAS_DT_FN int mas_dt_in_range(double x) { return fabs(x) <= MAS_DT_MAX_ARG; } /* false for NaN */

MAS_DT_FN double mas_dt_flip_sign(double v, unsigned flip /* bit 0 */) {
  unsigned long long u;
  __builtin_memcpy(&u, &v, sizeof u);
  u ^= (unsigned long long)(flip & 1u) << 63;
  __builtin_memcpy(&v, &u, sizeof v);
  return v;
}

/* reduction shared by all entry points: quadrant n, reduced argument y0 (NaN when x is out of range), tail y1 */
MAS_DT_FN unsigned mas_dt_reduce(double x, double* y0, double* y1) {
  const int ok = mas_dt_in_range(x);
  const unsigned n = (unsigned)mas_dt_rem_pio2(x, y0, y1);   /* out of range: a meaningless reduction (and quadrant) of a result that is NaN anyway */
  *y0 += ok ? -0.0 : (double)NAN;
  return n;
}

MAS_DT_FN double mas_dt_sin(double x) {
  double y0, y1;
  const unsigned n = mas_dt_reduce(x, &y0, &y1);
  const double r = (n & 1u) ? mas_dt_kcos(y0, y1) : mas_dt_ksin(y0, y1);
  return mas_dt_flip_sign(r, n >> 1);
}

MAS_DT_FN double mas_dt_cos(double x) {
  double y0, y1;
  const unsigned n = mas_dt_reduce(x, &y0, &y1);
  const double r = (n & 1u) ? mas_dt_ksin(y0, y1) : mas_dt_kcos(y0, y1);
  return mas_dt_flip_sign(r, (n + 1u) >> 1);
}

/* both at once (one reduction); bitwise equal to mas_dt_sin / mas_dt_cos */
MAS_DT_FN void mas_dt_sincos(double x, double* sn, double* cs) {
  double y0, y1;
  const unsigned n = mas_dt_reduce(x, &y0, &y1);
  const double s = mas_dt_ksin(y0, y1), c = mas_dt_kcos(y0, y1);
  const double rs = (n & 1u) ? c : s, rc = (n & 1u) ? s : c;
  *sn = mas_dt_flip_sign(rs, n >> 1);
  *cs = mas_dt_flip_sign(rc, (n + 1u) >> 1);
}

MAS_DT_FN double mas_dt_tan(double x) {
  double y0, y1;
  const unsigned n = mas_dt_reduce(x, &y0, &y1);
  const double s = mas_dt_ksin(y0, y1), c = mas_dt_kcos(y0, y1);
  return (n & 1u) ? -(c / s) : s / c;
}

/* The same function with ONE division of selected operands: in an odd quadrant -(c / s) is (-c) / s bit for bit (division is
 * symmetric in sign), and a choice of operands is a select where a choice between two divisions becomes control flow.  Control
 * flow is not merged across the perturbed evaluations of a finite-difference sweep and splits a rollout step into several
 * scheduling blocks: the six-state bicycle, which takes tan of a STATE in every dynamics evaluation, has 33 divisions and 4 300
 * instructions in its sweep with mas_dt_tan and 4 / 3 000 with this form (and a 8 % shorter rollout step); the single-track
 * models, which take tan of the control once per step, are 3 % faster with the branch (one division executed, no selects).
 * tests/test_dettrig.py holds the two forms against each other bitwise. */
MAS_DT_FN double mas_dt_tan_onediv(double x) {
  double y0, y1;
  const unsigned n = mas_dt_reduce(x, &y0, &y1);
  const double s = mas_dt_ksin(y0, y1), c = mas_dt_kcos(y0, y1);
  const double num = (n & 1u) ? -c : s, den = (n & 1u) ? s : c;
  return num / den;
}

#endif /* MAS_DETTRIG_H */
