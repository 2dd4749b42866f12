// txfm_device.h — VP9 inverse 1-D transforms as register-resident device functions.
//
// One lane owns one whole row (or column) of a transform block and runs the 1-D
// transform on N values held in VGPRs; every array index below is a compile-time
// constant after unrolling, so nothing spills to scratch.  The arithmetic contract
// (what is rounded, what wraps and at which width) is libvpx's
// vpx_dsp/inv_txfm.c: idct4_c :133, idct8_c :271, idct16_c :557, idct32_c :813,
// iadst4_c :96, iadst8_c :196, iadst16_c :389 and their vpx_highbd_* twins
// (:1373-2170).  HBD=false reproduces the 8-bit functions (int16 step arrays, `int`
// products), HBD=true the highbd ones (int32 steps, 64-bit products).
#ifndef VP9HIP_TXFM_DEVICE_H_
#define VP9HIP_TXFM_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace txfm {

// Rows of an N x N coefficient block that can be non-zero (the reference's clearing rule, vp9_decodeframe.c:960-967):
// the only rows the kernels read — and the only ones a compact slot holds.  Same as vp9hip_coeff_rows (vp9hip_pack.h).
__device__ __forceinline__ int coeff_rows(int eob, int tx_type, int n) {
  if (eob == 1) return 1;
  if (tx_type == 0 && n <= 16 && eob <= 10) return 4;
  if (n == 32 && eob <= 34) return 8;
  return n;
}

// The coefficient buffer of a launch: int32 slots (tran_low_t of the reference's build), or int16 ones when the
// caller narrowed them (vp9hip_set_coeff_bits: a frame whose coefficients all fit travels at half the bytes).
struct Coefs {
  const void *p;
  int c16;
};
struct CoefAt {
  const void *p;
  unsigned off;
  int c16;
  __device__ __forceinline__ int operator[](int i) const {
    return c16 ? (int)((const short *)p)[off + (unsigned)i] : ((const int *)p)[off + (unsigned)i];
  }
};
__device__ __forceinline__ CoefAt at(const Coefs &c, unsigned off) {
  CoefAt a;
  a.p = c.p;
  a.off = off;
  a.c16 = c.c16;
  return a;
}

// round(16384 * cos(k*pi/64)), vpx_dsp/txfm_common.h:28-58
#define CK(k) (txfm::kCos[k])
__device__ constexpr int kCos[33] = { 16384, 16364, 16305, 16207, 16069, 15893, 15679, 15426, 15137,
                                      14811, 14449, 14053, 13623, 13160, 12665, 12140, 11585, 11003,
                                      10394, 9760,  9102,  8423,  7723,  7005,  6270,  5520,  4756,
                                      3981,  3196,  2404,  1606,  804,   0 };
// txfm_common.h:61-64
__device__ constexpr int kSin[5] = { 0, 5283, 9929, 13377, 15212 };

typedef long long i64;

__device__ __forceinline__ int add32(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int sub32(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int neg32(int a) { return (int)(0u - (unsigned)a); }
__device__ __forceinline__ int mul32(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// width of the reference's step[] arrays
template <bool HBD>
__device__ __forceinline__ int wstep(int v) {
  if constexpr (HBD)
    return v;
  else
    return (int)(short)v;
}

// exact ((int64)t + 8192) >> 14 for a 32-bit t, without 64-bit ops
__device__ __forceinline__ int rs14_i32(int t) { return (t >> 14) + (((t & 0x3fff) + 8192) >> 14); }
__device__ __forceinline__ int rs14_i64(i64 t) { return (int)((t + 8192) >> 14); }

// rs14(a*ca + b*cb), result narrowed to step width.  8-bit: a,b are int16-range so the sum
// fits 32 bits; highbd: 64-bit.
template <bool HBD>
__device__ __forceinline__ int mac2(int a, int ca, int b, int cb) {
  if constexpr (HBD)
    return rs14_i64((i64)a * ca + (i64)b * cb);
  else
    return wstep<false>((a * ca + b * cb + 8192) >> 14);
}
// rs14(v * c) where v = a +/- b computed at the reference's width
template <bool HBD>
__device__ __forceinline__ int mul1(int v, int c) {
  if constexpr (HBD)
    return rs14_i64((i64)v * c);
  else
    return wstep<false>((v * c + 8192) >> 14);
}
template <bool HBD>
__device__ __forceinline__ int sadd(int a, int b) { return wstep<HBD>(add32(a, b)); }
template <bool HBD>
__device__ __forceinline__ int ssub(int a, int b) { return wstep<HBD>(sub32(a, b)); }

// ---- DCT ---------------------------------------------------------------------------------
template <bool HBD>
__device__ __forceinline__ void idct4_core(int x0, int x1, int x2, int x3, int *y) {
  const int s0 = mul1<HBD>(add32(x0, x2), CK(16));
  const int s1 = mul1<HBD>(sub32(x0, x2), CK(16));
  const int s2 = mac2<HBD>(x1, CK(24), x3, -CK(8));
  const int s3 = mac2<HBD>(x1, CK(8), x3, CK(24));
  y[0] = add32(s0, s3);
  y[1] = add32(s1, s2);
  y[2] = sub32(s1, s2);
  y[3] = sub32(s0, s3);
}

template <bool HBD>
__device__ __forceinline__ void idct8_odd(int x1, int x3, int x5, int x7, int *o) {
  const int a4 = mac2<HBD>(x1, CK(28), x7, -CK(4)), a7 = mac2<HBD>(x1, CK(4), x7, CK(28));
  const int a5 = mac2<HBD>(x5, CK(12), x3, -CK(20)), a6 = mac2<HBD>(x5, CK(20), x3, CK(12));
  const int b4 = sadd<HBD>(a4, a5), b5 = ssub<HBD>(a4, a5);
  const int b6 = ssub<HBD>(a7, a6), b7 = sadd<HBD>(a6, a7);
  o[0] = b4;
  o[1] = mul1<HBD>(sub32(b6, b5), CK(16));
  o[2] = mul1<HBD>(add32(b5, b6), CK(16));
  o[3] = b7;
}

template <bool HBD>
__device__ __forceinline__ void idct16_odd(const int *x /* x1,x3,..,x15 */, int *o) {
  int a[8], b[8], c[8];
  a[0] = mac2<HBD>(x[0], CK(30), x[7], -CK(2));
  a[7] = mac2<HBD>(x[0], CK(2), x[7], CK(30));
  a[1] = mac2<HBD>(x[4], CK(14), x[3], -CK(18));
  a[6] = mac2<HBD>(x[4], CK(18), x[3], CK(14));
  a[2] = mac2<HBD>(x[2], CK(22), x[5], -CK(10));
  a[5] = mac2<HBD>(x[2], CK(10), x[5], CK(22));
  a[3] = mac2<HBD>(x[6], CK(6), x[1], -CK(26));
  a[4] = mac2<HBD>(x[6], CK(26), x[1], CK(6));
  b[0] = sadd<HBD>(a[0], a[1]);
  b[1] = ssub<HBD>(a[0], a[1]);
  b[2] = ssub<HBD>(a[3], a[2]);
  b[3] = sadd<HBD>(a[2], a[3]);
  b[4] = sadd<HBD>(a[4], a[5]);
  b[5] = ssub<HBD>(a[4], a[5]);
  b[6] = ssub<HBD>(a[7], a[6]);
  b[7] = sadd<HBD>(a[6], a[7]);
  c[0] = b[0];
  c[7] = b[7];
  c[1] = mac2<HBD>(b[1], -CK(8), b[6], CK(24));
  c[6] = mac2<HBD>(b[1], CK(24), b[6], CK(8));
  c[2] = mac2<HBD>(b[2], -CK(24), b[5], -CK(8));
  c[5] = mac2<HBD>(b[2], -CK(8), b[5], CK(24));
  c[3] = b[3];
  c[4] = b[4];
  a[0] = sadd<HBD>(c[0], c[3]);
  a[1] = sadd<HBD>(c[1], c[2]);
  a[2] = ssub<HBD>(c[1], c[2]);
  a[3] = ssub<HBD>(c[0], c[3]);
  a[4] = ssub<HBD>(c[7], c[4]);
  a[5] = ssub<HBD>(c[6], c[5]);
  a[6] = sadd<HBD>(c[5], c[6]);
  a[7] = sadd<HBD>(c[4], c[7]);
  o[0] = a[0];
  o[1] = a[1];
  o[2] = mul1<HBD>(sub32(a[5], a[2]), CK(16));
  o[5] = mul1<HBD>(add32(a[2], a[5]), CK(16));
  o[3] = mul1<HBD>(sub32(a[4], a[3]), CK(16));
  o[4] = mul1<HBD>(add32(a[3], a[4]), CK(16));
  o[6] = a[6];
  o[7] = a[7];
}

template <bool HBD>
__device__ __forceinline__ void idct32_odd(const int *x /* x1,x3,..,x31 */, int *o) {
  int a[16], b[16];
  a[0] = mac2<HBD>(x[0], CK(31), x[15], -CK(1));
  a[15] = mac2<HBD>(x[0], CK(1), x[15], CK(31));
  a[1] = mac2<HBD>(x[8], CK(15), x[7], -CK(17));
  a[14] = mac2<HBD>(x[8], CK(17), x[7], CK(15));
  a[2] = mac2<HBD>(x[4], CK(23), x[11], -CK(9));
  a[13] = mac2<HBD>(x[4], CK(9), x[11], CK(23));
  a[3] = mac2<HBD>(x[12], CK(7), x[3], -CK(25));
  a[12] = mac2<HBD>(x[12], CK(25), x[3], CK(7));
  a[4] = mac2<HBD>(x[2], CK(27), x[13], -CK(5));
  a[11] = mac2<HBD>(x[2], CK(5), x[13], CK(27));
  a[5] = mac2<HBD>(x[10], CK(11), x[5], -CK(21));
  a[10] = mac2<HBD>(x[10], CK(21), x[5], CK(11));
  a[6] = mac2<HBD>(x[6], CK(19), x[9], -CK(13));
  a[9] = mac2<HBD>(x[6], CK(13), x[9], CK(19));
  a[7] = mac2<HBD>(x[14], CK(3), x[1], -CK(29));
  a[8] = mac2<HBD>(x[14], CK(29), x[1], CK(3));
#pragma unroll
  for (int g = 0; g < 16; g += 4) {
    b[g + 0] = sadd<HBD>(a[g + 0], a[g + 1]);
    b[g + 1] = ssub<HBD>(a[g + 0], a[g + 1]);
    b[g + 2] = ssub<HBD>(a[g + 3], a[g + 2]);
    b[g + 3] = sadd<HBD>(a[g + 2], a[g + 3]);
  }
  a[0] = b[0];
  a[15] = b[15];
  a[1] = mac2<HBD>(b[1], -CK(4), b[14], CK(28));
  a[14] = mac2<HBD>(b[1], CK(28), b[14], CK(4));
  a[2] = mac2<HBD>(b[2], -CK(28), b[13], -CK(4));
  a[13] = mac2<HBD>(b[2], -CK(4), b[13], CK(28));
  a[3] = b[3];
  a[4] = b[4];
  a[5] = mac2<HBD>(b[5], -CK(20), b[10], CK(12));
  a[10] = mac2<HBD>(b[5], CK(12), b[10], CK(20));
  a[6] = mac2<HBD>(b[6], -CK(12), b[9], -CK(20));
  a[9] = mac2<HBD>(b[6], -CK(20), b[9], CK(12));
  a[7] = b[7];
  a[8] = b[8];
  a[11] = b[11];
  a[12] = b[12];
  b[0] = sadd<HBD>(a[0], a[3]);
  b[1] = sadd<HBD>(a[1], a[2]);
  b[2] = ssub<HBD>(a[1], a[2]);
  b[3] = ssub<HBD>(a[0], a[3]);
  b[4] = ssub<HBD>(a[7], a[4]);
  b[5] = ssub<HBD>(a[6], a[5]);
  b[6] = sadd<HBD>(a[5], a[6]);
  b[7] = sadd<HBD>(a[4], a[7]);
  b[8] = sadd<HBD>(a[8], a[11]);
  b[9] = sadd<HBD>(a[9], a[10]);
  b[10] = ssub<HBD>(a[9], a[10]);
  b[11] = ssub<HBD>(a[8], a[11]);
  b[12] = ssub<HBD>(a[15], a[12]);
  b[13] = ssub<HBD>(a[14], a[13]);
  b[14] = sadd<HBD>(a[13], a[14]);
  b[15] = sadd<HBD>(a[12], a[15]);
  a[0] = b[0];
  a[1] = b[1];
  a[2] = mac2<HBD>(b[2], -CK(8), b[13], CK(24));
  a[13] = mac2<HBD>(b[2], CK(24), b[13], CK(8));
  a[3] = mac2<HBD>(b[3], -CK(8), b[12], CK(24));
  a[12] = mac2<HBD>(b[3], CK(24), b[12], CK(8));
  a[4] = mac2<HBD>(b[4], -CK(24), b[11], -CK(8));
  a[11] = mac2<HBD>(b[4], -CK(8), b[11], CK(24));
  a[5] = mac2<HBD>(b[5], -CK(24), b[10], -CK(8));
  a[10] = mac2<HBD>(b[5], -CK(8), b[10], CK(24));
  a[6] = b[6];
  a[7] = b[7];
  a[8] = b[8];
  a[9] = b[9];
  a[14] = b[14];
  a[15] = b[15];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b[i] = sadd<HBD>(a[i], a[7 - i]);
    b[7 - i] = ssub<HBD>(a[i], a[7 - i]);
    b[8 + i] = ssub<HBD>(a[15 - i], a[8 + i]);
    b[15 - i] = sadd<HBD>(a[8 + i], a[15 - i]);
  }
  o[0] = b[0];
  o[1] = b[1];
  o[2] = b[2];
  o[3] = b[3];
#pragma unroll
  for (int i = 4; i < 8; ++i) {
    o[i] = mul1<HBD>(sub32(b[15 - i], b[i]), CK(16));
    o[15 - i] = mul1<HBD>(add32(b[i], b[15 - i]), CK(16));
  }
  o[12] = b[12];
  o[13] = b[13];
  o[14] = b[14];
  o[15] = b[15];
}

// y = idctN(x); x already narrowed to step width.  y is 32-bit (top-level) — the caller of
// an embedded half applies wstep.
template <int N, bool HBD>
__device__ __forceinline__ void idct_core(const int *x, int *y) {
  if constexpr (N == 4) {
    idct4_core<HBD>(x[0], x[1], x[2], x[3], y);
  } else {
    constexpr int H = N / 2;
    int ev[H], od[H], e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      ev[i] = x[2 * i];
      od[i] = x[2 * i + 1];
    }
    idct_core<H, HBD>(ev, e);
#pragma unroll
    for (int i = 0; i < H; ++i) e[i] = wstep<HBD>(e[i]);
    if constexpr (N == 8)
      idct8_odd<HBD>(od[0], od[1], od[2], od[3], o);
    else if constexpr (N == 16)
      idct16_odd<HBD>(od, o);
    else
      idct32_odd<HBD>(od, o);
#pragma unroll
    for (int i = 0; i < H; ++i) {
      y[i] = add32(e[i], o[H - 1 - i]);
      y[N - 1 - i] = sub32(e[i], o[H - 1 - i]);
    }
  }
}

// highbd 1-D transforms zero their output when any |input| >= 2^25 (inv_txfm.c:1278-1290)
template <int N>
__device__ __forceinline__ bool hbd_invalid(const int *v) {
  bool bad = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int a = v[i] < 0 ? neg32(v[i]) : v[i];
    bad |= ((unsigned)a >= (1u << 25));
  }
  return bad;
}

template <int N, bool HBD>
__device__ __forceinline__ void idct1d(int *v) {
  int x[N], y[N];
  if constexpr (HBD) {
    const bool bad = hbd_invalid<N>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = bad ? 0 : v[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = (int)(short)v[i];
  }
  idct_core<N, HBD>(x, y);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = y[i];
}

// ---- ADST --------------------------------------------------------------------------------
template <bool HBD>
__device__ __forceinline__ void iadst4(int *v) {
  const int x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  // (all-zero input gives all-zero output through the arithmetic as well)
  i64 s0, s1, s2, s3, s4, s5, s6;
  if constexpr (HBD) {
    s0 = (i64)kSin[1] * x0;
    s1 = (i64)kSin[2] * x0;
    s2 = (i64)kSin[3] * x1;
    s3 = (i64)kSin[4] * x2;
    s4 = (i64)kSin[1] * x2;
    s5 = (i64)kSin[2] * x3;
    s6 = (i64)kSin[4] * x3;
  } else {  // int * int32 products
    s0 = mul32(kSin[1], x0);
    s1 = mul32(kSin[2], x0);
    s2 = mul32(kSin[3], x1);
    s3 = mul32(kSin[4], x2);
    s4 = mul32(kSin[1], x2);
    s5 = mul32(kSin[2], x3);
    s6 = mul32(kSin[4], x3);
  }
  const i64 s7 = add32(sub32(x0, x2), x3);
  s0 = s0 + s3 + s5;
  s1 = s1 - s4 - s6;
  s3 = s2;
  s2 = (i64)kSin[3] * s7;
  v[0] = rs14_i64(s0 + s3);
  v[1] = rs14_i64(s1 + s3);
  v[2] = rs14_i64(s2);
  v[3] = rs14_i64(s0 + s1 - s3);
}

// 8-bit iadst8_c keeps its s-terms in `int` (wrap at 32 bits), highbd in int64.
template <bool HBD>
struct A8 {
  typedef typename std::conditional<HBD, i64, int>::type S;
  static __device__ __forceinline__ S mac(int ca, int a, int cb, int b) {
    if constexpr (HBD)
      return (i64)ca * a + (i64)cb * b;
    else
      return add32(mul32(ca, a), mul32(cb, b));
  }
  static __device__ __forceinline__ S add(S a, S b) {
    if constexpr (HBD)
      return a + b;
    else
      return add32(a, b);
  }
  static __device__ __forceinline__ S sub(S a, S b) {
    if constexpr (HBD)
      return a - b;
    else
      return sub32(a, b);
  }
  static __device__ __forceinline__ int rs(S s) {
    if constexpr (HBD)
      return rs14_i64(s);
    else
      return rs14_i32(s);
  }
};

template <bool HBD>
__device__ __forceinline__ void iadst8(int *v) {
  typedef A8<HBD> A;
  typedef typename A::S S;
  int x0 = v[7], x1 = v[0], x2 = v[5], x3 = v[2], x4 = v[3], x5 = v[4], x6 = v[1], x7 = v[6];
  S s0 = A::mac(CK(2), x0, CK(30), x1), s1 = A::mac(CK(30), x0, -CK(2), x1);
  S s2 = A::mac(CK(10), x2, CK(22), x3), s3 = A::mac(CK(22), x2, -CK(10), x3);
  S s4 = A::mac(CK(18), x4, CK(14), x5), s5 = A::mac(CK(14), x4, -CK(18), x5);
  S s6 = A::mac(CK(26), x6, CK(6), x7), s7 = A::mac(CK(6), x6, -CK(26), x7);
  x0 = A::rs(A::add(s0, s4));
  x1 = A::rs(A::add(s1, s5));
  x2 = A::rs(A::add(s2, s6));
  x3 = A::rs(A::add(s3, s7));
  x4 = A::rs(A::sub(s0, s4));
  x5 = A::rs(A::sub(s1, s5));
  x6 = A::rs(A::sub(s2, s6));
  x7 = A::rs(A::sub(s3, s7));
  s4 = A::mac(CK(8), x4, CK(24), x5);
  s5 = A::mac(CK(24), x4, -CK(8), x5);
  s6 = A::mac(-CK(24), x6, CK(8), x7);
  s7 = A::mac(CK(8), x6, CK(24), x7);
  {
    const int t0 = x0, t1 = x1, t2 = x2, t3 = x3;
    x0 = add32(t0, t2);
    x1 = add32(t1, t3);
    x2 = sub32(t0, t2);
    x3 = sub32(t1, t3);
  }
  x4 = A::rs(A::add(s4, s6));
  x5 = A::rs(A::add(s5, s7));
  x6 = A::rs(A::sub(s4, s6));
  x7 = A::rs(A::sub(s5, s7));
  // stage 3: 8-bit multiplies the 64-bit sum and truncates to int == 32-bit wrapping product;
  // highbd adds in int32 then multiplies in 64 bit.
  S t2, t3, t6, t7;
  if constexpr (HBD) {
    t2 = (i64)CK(16) * add32(x2, x3);
    t3 = (i64)CK(16) * sub32(x2, x3);
    t6 = (i64)CK(16) * add32(x6, x7);
    t7 = (i64)CK(16) * sub32(x6, x7);
  } else {
    t2 = mul32(CK(16), add32(x2, x3));
    t3 = mul32(CK(16), sub32(x2, x3));
    t6 = mul32(CK(16), add32(x6, x7));
    t7 = mul32(CK(16), sub32(x6, x7));
  }
  x2 = A::rs(t2);
  x3 = A::rs(t3);
  x6 = A::rs(t6);
  x7 = A::rs(t7);
  v[0] = x0;
  v[1] = neg32(x4);
  v[2] = x6;
  v[3] = neg32(x2);
  v[4] = x3;
  v[5] = neg32(x7);
  v[6] = x5;
  v[7] = neg32(x1);
}

// iadst16: s-terms are 64-bit in both variants.  The 8-bit function holds x in int64 (sums /
// negations exact), the highbd one in int32 (they wrap).
template <bool HBD>
__device__ __forceinline__ i64 xsum(int a, int b) {
  if constexpr (HBD)
    return add32(a, b);
  else
    return (i64)a + b;
}
template <bool HBD>
__device__ __forceinline__ i64 xdif(int a, int b) {
  if constexpr (HBD)
    return sub32(a, b);
  else
    return (i64)a - b;
}
template <bool HBD>
__device__ __forceinline__ i64 xneg(int a) {
  if constexpr (HBD)
    return neg32(a);
  else
    return -(i64)a;
}

template <bool HBD>
__device__ __forceinline__ void iadst16(int *v) {
  int x[16];
  i64 s[16];
  {
    constexpr int perm[16] = { 15, 0, 13, 2, 11, 4, 9, 6, 7, 8, 5, 10, 3, 12, 1, 14 };
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = v[perm[i]];
  }
  {
    constexpr int c1[8] = { 1, 5, 9, 13, 17, 21, 25, 29 };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s[2 * i] = (i64)x[2 * i] * CK(c1[i]) + (i64)x[2 * i + 1] * CK(32 - c1[i]);
      s[2 * i + 1] = (i64)x[2 * i] * CK(32 - c1[i]) - (i64)x[2 * i + 1] * CK(c1[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x[i] = rs14_i64(s[i] + s[i + 8]);
    x[i + 8] = rs14_i64(s[i] - s[i + 8]);
  }
  // stage 2
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = x[i];
  s[8] = (i64)x[8] * CK(4) + (i64)x[9] * CK(28);
  s[9] = (i64)x[8] * CK(28) - (i64)x[9] * CK(4);
  s[10] = (i64)x[10] * CK(20) + (i64)x[11] * CK(12);
  s[11] = (i64)x[10] * CK(12) - (i64)x[11] * CK(20);
  s[12] = xneg<HBD>(x[12]) * CK(28) + (i64)x[13] * CK(4);
  s[13] = (i64)x[12] * CK(4) + (i64)x[13] * CK(28);
  s[14] = xneg<HBD>(x[14]) * CK(12) + (i64)x[15] * CK(20);
  s[15] = (i64)x[14] * CK(20) + (i64)x[15] * CK(12);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[i] = (int)(s[i] + s[i + 4]);
    x[i + 4] = (int)(s[i] - s[i + 4]);
    x[i + 8] = rs14_i64(s[i + 8] + s[i + 12]);
    x[i + 12] = rs14_i64(s[i + 8] - s[i + 12]);
  }
  // stage 3
#pragma unroll
  for (int g = 0; g < 16; g += 8) {
    s[g + 0] = x[g + 0];
    s[g + 1] = x[g + 1];
    s[g + 2] = x[g + 2];
    s[g + 3] = x[g + 3];
    s[g + 4] = (i64)x[g + 4] * CK(8) + (i64)x[g + 5] * CK(24);
    s[g + 5] = (i64)x[g + 4] * CK(24) - (i64)x[g + 5] * CK(8);
    s[g + 6] = xneg<HBD>(x[g + 6]) * CK(24) + (i64)x[g + 7] * CK(8);
    s[g + 7] = (i64)x[g + 6] * CK(8) + (i64)x[g + 7] * CK(24);
    x[g + 0] = (int)(s[g + 0] + s[g + 2]);
    x[g + 1] = (int)(s[g + 1] + s[g + 3]);
    x[g + 2] = (int)(s[g + 0] - s[g + 2]);
    x[g + 3] = (int)(s[g + 1] - s[g + 3]);
    x[g + 4] = rs14_i64(s[g + 4] + s[g + 6]);
    x[g + 5] = rs14_i64(s[g + 5] + s[g + 7]);
    x[g + 6] = rs14_i64(s[g + 4] - s[g + 6]);
    x[g + 7] = rs14_i64(s[g + 5] - s[g + 7]);
  }
  // stage 4
  const int r2 = rs14_i64((i64)(-CK(16)) * xsum<HBD>(x[2], x[3]));
  const int r3 = rs14_i64((i64)CK(16) * xdif<HBD>(x[2], x[3]));
  const int r6 = rs14_i64((i64)CK(16) * xsum<HBD>(x[6], x[7]));
  const int r7 = rs14_i64((i64)CK(16) * (HBD ? (i64)add32(neg32(x[6]), x[7]) : (i64)x[7] - x[6]));
  const int r10 = rs14_i64((i64)CK(16) * xsum<HBD>(x[10], x[11]));
  const int r11 = rs14_i64((i64)CK(16) * (HBD ? (i64)add32(neg32(x[10]), x[11]) : (i64)x[11] - x[10]));
  const int r14 = rs14_i64((i64)(-CK(16)) * xsum<HBD>(x[14], x[15]));
  const int r15 = rs14_i64((i64)CK(16) * xdif<HBD>(x[14], x[15]));
  v[0] = x[0];
  v[1] = neg32(x[8]);
  v[2] = x[12];
  v[3] = neg32(x[4]);
  v[4] = r6;
  v[5] = r14;
  v[6] = r10;
  v[7] = r2;
  v[8] = r3;
  v[9] = r11;
  v[10] = r15;
  v[11] = r7;
  v[12] = x[5];
  v[13] = neg32(x[13]);
  v[14] = x[9];
  v[15] = neg32(x[1]);
}

template <int N, bool HBD>
__device__ __forceinline__ void iadst1d(int *v) {
  if constexpr (HBD) {
    const bool bad = hbd_invalid<N>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = bad ? 0 : v[i];
  }
  if constexpr (N == 4)
    iadst4<HBD>(v);
  else if constexpr (N == 8)
    iadst8<HBD>(v);
  else if constexpr (N == 16)
    iadst16<HBD>(v);
}

// 4-point Walsh-Hadamard pass (inv_txfm.c:18-69); `first` applies the >> UNIT_QUANT_SHIFT
__device__ __forceinline__ void iwht4(int *v, bool first) {
  i64 a = first ? (v[0] >> 2) : v[0], c = first ? (v[1] >> 2) : v[1];
  i64 d = first ? (v[2] >> 2) : v[2], b = first ? (v[3] >> 2) : v[3];
  a += c;
  d -= b;
  const i64 e = (a - d) >> 1;
  b = e - b;
  c = e - c;
  a -= b;
  d += c;
  v[0] = (int)a;
  v[1] = (int)b;
  v[2] = (int)c;
  v[3] = (int)d;
}

// DC-only value: vpx_idctNxN_1_add_c (inv_txfm.c:178-194 ...), highbd (:1476-1494 ...)
template <int N, bool HBD>
__device__ __forceinline__ int dc_only(int dc) {
  constexpr int shift = N == 4 ? 4 : (N == 8 ? 5 : 6);
  int out;
  if constexpr (HBD) {
    out = rs14_i64((i64)dc * kCos[16]);
    out = rs14_i64((i64)out * kCos[16]);
  } else {
    out = rs14_i32((int)(short)dc * kCos[16]);
    out = rs14_i32(mul32(out, kCos[16]));
  }
  return add32(out, 1 << (shift - 1)) >> shift;
}


}  // namespace txfm
#endif
