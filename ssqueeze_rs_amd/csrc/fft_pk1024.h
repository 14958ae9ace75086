// fft_pk1024.h -- the 1024-point wave FFT's arithmetic on PACKED fp32 (v_pk_add/mul/fma_f32), gfx950.
//
// Why: a wave issues one vector instruction per 4 cycles; two waves of a SIMD can share its 32 lanes only while both
// have an fp32 add/mul/fma ready, and the counters of the 16-wave STFT kernel show that this rarely happens
// (0.87 vector instructions per 4 cycles and SIMD, profiles/r02_final_pmc_summary.txt).  A packed instruction does the
// two halves of a complex value in ONE issue slot, whatever the other waves do.  The swaps and sign flips complex
// arithmetic needs ride on the VOP3P operand selectors (op_sel / op_sel_hi: which half of a source feeds the low /
// high result; neg_lo / neg_hi: negate the source of the low / high result) -- the compiler folds whole-vector
// negations and swizzles but not a per-half negation, so those forms are written as (non-volatile) inline asm.
#pragma once
#include "ssq_common.h"

namespace ssq {
namespace pk {

typedef float v2f __attribute__((ext_vector_type(2)));   // (re, im) in an even-aligned VGPR pair

#define SSQ_PK_DI __device__ __forceinline__

// a * w
SSQ_PK_DI v2f cmul(v2f a, v2f w) {
  const v2f t = a * w.xx;                         // v_pk_mul_f32 ... op_sel_hi:[1,0]
  v2f r;                                          // lo = -a.y w.y + t.x ; hi = a.x w.y + t.y
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// a * w for a wave-uniform w kept in an SGPR pair (compile-time twiddles)
SSQ_PK_DI v2f cmulc(v2f a, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
// a + (-i) b = (a.x + b.y, a.y - b.x)
SSQ_PK_DI v2f addmi(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
SSQ_PK_DI v2f submi(v2f a, v2f b) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a * s for a wave-uniform real s (low half of an SGPR pair)
SSQ_PK_DI v2f scalec(v2f a, v2f s) {
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(s));
  return r;
}

// forward radix-4 butterfly, natural-order output (ssq_common.h::dft4)
SSQ_PK_DI void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = addmi(t1, d);
  a3 = submi(t1, d);
}
// the same with the input a2 still to be multiplied by -i (W16^4): the factor rides on the first two adds
SSQ_PK_DI void dft4_a2mi(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = addmi(a0, a2), t1 = submi(a0, a2), t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = addmi(t1, d);
  a3 = submi(t1, d);
}

constexpr float kC = 0.70710678118654752440f, kC1 = 0.92387953251128675613f, kS1 = 0.38268343236508977173f;

// forward 16-point DFT in registers, natural order (ssq_common.h::dft16): 64 packed adds + 16 packed twiddle ops
SSQ_PK_DI void dft16(v2f (&v)[16]) {
  dft4(v[0], v[4], v[8], v[12]);
  dft4(v[1], v[5], v[9], v[13]);
  dft4(v[2], v[6], v[10], v[14]);
  dft4(v[3], v[7], v[11], v[15]);
  const v2f w1 = {kC1, -kS1}, w3 = {kS1, -kC1}, w9 = {-kC1, kS1}, cp = {kC, kC}, cm = {-kC, -kC};
  v[5] = cmulc(v[5], w1);                         // W16^1
  v[6] = scalec(addmi(v[6], v[6]), cp);           // W16^2 = (1 - i)/sqrt 2
  v[7] = cmulc(v[7], w3);                         // W16^3
  v[9] = scalec(addmi(v[9], v[9]), cp);           // W16^2
  /* v[10]: W16^4 = -i, folded into its butterfly */
  v[11] = scalec(submi(v[11], v[11]), cm);        // W16^6 = (-1 - i)/sqrt 2
  v[13] = cmulc(v[13], w3);                       // W16^3
  v[14] = scalec(submi(v[14], v[14]), cm);        // W16^6
  v[15] = cmulc(v[15], w9);                       // W16^9
  dft4(v[0], v[1], v[2], v[3]);
  dft4(v[4], v[5], v[6], v[7]);
  dft4_a2mi(v[8], v[9], v[10], v[11]);
  dft4(v[12], v[13], v[14], v[15]);
#define SSQ_PK_SWAP(i, j) \
  {                       \
    const v2f t_ = v[i];  \
    v[i] = v[j];          \
    v[j] = t_;            \
  }
  SSQ_PK_SWAP(1, 4) SSQ_PK_SWAP(2, 8) SSQ_PK_SWAP(3, 12) SSQ_PK_SWAP(6, 9) SSQ_PK_SWAP(7, 13) SSQ_PK_SWAP(11, 14)
#undef SSQ_PK_SWAP
}

// pass 1 of the 1024-point wave transform: twiddle W_256^(k m) from the compact [m][k] table, radix 16
SSQ_PK_DI void pass1(v2f (&v)[16], const v2f* __restrict__ tw1, int t) {
  const int k = t & 15;
#pragma unroll
  for (int m = 1; m < 16; ++m) v[m] = cmul(v[m], tw1[m * 16 + k]);
  dft16(v);
}
// pass 2: twiddle W_1024^((t + 64 b) m) from the compact table (row m - 1 of [3][256]), four radix-4 butterflies
SSQ_PK_DI void pass2(v2f (&v)[16], const v2f* __restrict__ tw2, int t) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int m = 1; m < 4; ++m) v[b + 4 * m] = cmul(v[b + 4 * m], tw2[(m - 1) * 256 + t + 64 * b]);
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) dft4(v[b], v[b + 4], v[b + 8], v[b + 12]);
}

}  // namespace pk
}  // namespace ssq
