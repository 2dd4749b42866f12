// inter_kernels.hip — batched inter prediction: 8-tap sub-pel convolve (SURVEY §8 a5–a7).
//
// One wavefront (= one 64-thread workgroup) per prediction task.  The wave stages the
// clamped (w+7)x(h+7) reference window in LDS (coordinate clamping == libvpx's decoder border
// emulation, vp9_decodeframe.c:432-505 build_mc_border), filters rows into a clipped LDS
// intermediate (vpx_convolve.c:156-188: the clip BETWEEN the passes is normative) and then
// filters columns.  Phase-0 kernels are the identity ({0,0,0,128,0,0,0,0}: (128*p+64)>>7 = p),
// so always running both passes equals libvpx's copy / horiz-only / vert-only dispatch
// (vp9_scale.c:79-130).  Compound prediction = second reference averaged into the first
// ((a+b+1)>>1, vpx_convolve_avg_c :226-240).
//
// Algorithmic bytes per task: w*h*bps read (reference) + w*h*bps written (+ w*h*bps more
// reference for compound) + 32 (descriptor).
#include "vp9hip_internal.h"
#include <stdlib.h>

namespace {

// vp9/common/vp9_filter.c:14-82, in INTERP_FILTER order (vp9_filter.h:23-28):
// EIGHTTAP, EIGHTTAP_SMOOTH, EIGHTTAP_SHARP, BILINEAR, FOURTAP.
#include "filter_table.inc"
__device__ const int16_t kFilters[5][16][8] = VP9HIP_FILTER_TABLE;
static const int16_t kFiltersHost[5][16][8] = VP9HIP_FILTER_TABLE;


struct RefSet {
  FrameDev f[VP9HIP_MAX_REFS];
};

constexpr int WIN = 72;  // max window extent (64 + 7, or 2*31 + 1 + 8 for 2:1 scaled 32x32 tiles)

template <typename Pix>
__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

// Predict one tile (tw x th at tile offset tx,ty inside the task) from one reference.
// Writes the result into dst (or averages into it when avg).
template <typename Pix>
__device__ void predict_tile(Pix *win, Pix *tmp, const FrameDev &rf, int plane, int px_q4, int py_q4,
                             int xs, int ys, int filt, int tw, int th, Pix *dst, int dstride, int vis_w,
                             int vis_h, bool avg, int maxv) {
  const int lane = threadIdx.x;
  const Pix *src = (const Pix *)rf.plane[plane];
  const int sstride = rf.stride[plane];
  const int fw = rf.width[plane], fh = rf.height[plane];
  const int x0 = px_q4 >> 4, y0 = py_q4 >> 4;
  const int subx = px_q4 & 15, suby = py_q4 & 15;
  const int ww = (((tw - 1) * xs + subx) >> 4) + 8;  // window columns  (x0-3 .. )
  const int wh = (((th - 1) * ys + suby) >> 4) + 8;  // window rows     (y0-3 .. ) == intermediate height
  // 1. stage the clamped window
  for (int i = lane; i < ww * wh; i += 64) {
    const int r = i / ww, c = i - r * ww;
    const int sx = clampi<Pix>(x0 - 3 + c, 0, fw - 1);
    const int sy = clampi<Pix>(y0 - 3 + r, 0, fh - 1);
    win[r * WIN + c] = src[(size_t)sy * sstride + sx];
  }
  __syncthreads();
  // 2. rows -> clipped intermediate tmp[wh][tw]
  for (int i = lane; i < tw * wh; i += 64) {
    const int r = i / tw, c = i - r * tw;
    const int pos = subx + c * xs;
    const int16_t *f = kFilters[filt][pos & 15];
    const Pix *s = win + r * WIN + (pos >> 4);
    int sum = 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)s[k] * f[k];
    tmp[r * 64 + c] = (Pix)clampi<Pix>(sum >> 7, 0, maxv);
  }
  __syncthreads();
  // 3. columns -> destination
  for (int i = lane; i < tw * th; i += 64) {
    const int r = i / tw, c = i - r * tw;
    const int pos = suby + r * ys;
    const int16_t *f = kFilters[filt][pos & 15];
    const Pix *s = tmp + (pos >> 4) * 64 + c;
    int sum = 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)s[k * 64] * f[k];
    int v = clampi<Pix>(sum >> 7, 0, maxv);
    if (r < vis_h && c < vis_w) {
      Pix *d = dst + (size_t)r * dstride + c;
      if (avg) v = ((int)*d + v + 1) >> 1;
      *d = (Pix)v;
    }
  }
  __syncthreads();
}

template <typename Pix>
__global__ __launch_bounds__(64) void inter_pred_kernel(const vp9hip_inter_task *__restrict__ tasks, int n_tasks,
                                                        RefSet refs, FrameDev dstf) {
  __shared__ Pix win[WIN * WIN];
  __shared__ Pix tmp[WIN * 64];
  const int ti = blockIdx.x;
  if (ti >= n_tasks) return;
  const vp9hip_inter_task t = tasks[ti];
  const int plane = t.plane;
  const int filt = (t.flags >> 1) & 7;
  const int nref = (t.flags & 1) ? 2 : 1;
  const int maxv = (1 << dstf.bit_depth) - 1;
  Pix *dplane = (Pix *)dstf.plane[plane];
  const int dstride = dstf.stride[plane];
  for (int r = 0; r < nref; ++r) {
    const FrameDev &rf = refs.f[t.ref[r]];
    const int xs = t.step_x[r], ys = t.step_y[r];
    // unscaled: one 64x64 tile; scaled: 32x32 tiles keep the window inside WIN
    const int tile = (xs == 16 && ys == 16) ? 64 : 32;
    for (int ty = 0; ty < t.h; ty += tile) {
      for (int tx = 0; tx < t.w; tx += tile) {
        const int tw = min(tile, t.w - tx), th = min(tile, t.h - ty);
        const int dx = t.dst_x + tx, dy = t.dst_y + ty;
        // visible part of the tile (blocks may overhang the aligned frame)
        const int vis_w = min(tw, dstf.awidth[plane] - dx), vis_h = min(th, dstf.aheight[plane] - dy);
        if (vis_w <= 0 || vis_h <= 0) continue;
        predict_tile<Pix>(win, tmp, rf, plane, t.pos_x[r] + tx * xs, t.pos_y[r] + ty * ys, xs, ys, filt, tw, th,
                          dplane + (size_t)dy * dstride + dx, dstride, vis_w, vis_h, r == 1, maxv);
      }
    }
  }
}


typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot8_16(unsigned s0, unsigned s1, unsigned s2, unsigned s3, const uint4 &f, int maxv) {
  short2v a, b;
  int acc = 64;
  __builtin_memcpy(&a, &s0, 4); __builtin_memcpy(&b, &f.x, 4); acc = __builtin_amdgcn_sdot2(a, b, acc, false);
  __builtin_memcpy(&a, &s1, 4); __builtin_memcpy(&b, &f.y, 4); acc = __builtin_amdgcn_sdot2(a, b, acc, false);
  __builtin_memcpy(&a, &s2, 4); __builtin_memcpy(&b, &f.z, 4); acc = __builtin_amdgcn_sdot2(a, b, acc, false);
  __builtin_memcpy(&a, &s3, 4); __builtin_memcpy(&b, &f.w, 4); acc = __builtin_amdgcn_sdot2(a, b, acc, false);
  acc >>= 7;
  return acc < 0 ? 0 : (acc > maxv ? maxv : acc);
}

// XCD-aware order inside a class.  Workgroup b runs on XCD b % 8 (round-robin dispatch; an affinity
// assumption used for speed only) and each XCD has its own L2.  A class's tasks are in decode
// (superblock-raster) order, so handing XCD x the x-th contiguous eighth of every class makes
// each XCD work on one horizontal band of the frame across all classes: overlapping windows of
// neighbouring tiles and the band of each reference it reads stay in that XCD's 4 MiB L2 instead
// of being fetched once per XCD (PMC, 1440p frame: FETCH_SIZE 99 MB -> 25 MB, L2 hit rate
// 37 % -> 86 %; profiles/r01_pmc_s2a.json vs r01_pmc_s2b.json).
__device__ __forceinline__ int xcd_order(int b, int s0, int s1) {
  const int x = b & 7;
  int prefix = 0, first_x = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int first = s0 + ((j - s0) & 7);  // first workgroup of [s0, s1) on XCD j
    const int cnt = first < s1 ? (s1 - first + 7) >> 3 : 0;
    if (j < x) prefix += cnt;
    if (j == x) first_x = first;
  }
  return prefix + ((b - first_x) >> 3);
}

// ---------------------------------------------------------------------------------------------
// Register path: 8-bit samples, unscaled references.  No LDS and no cross-lane traffic at all: a
// lane owns FOUR adjacent output columns (one destination dword per row) of an 8-row STRIP of its
// task; a task of width W occupies W/4 neighbouring lanes per strip and HMAX/8 strips.
//  load    15 window rows (y0-3 .. y0+11, row index clamped to the plane = border emulation) of 12
//          bytes from column x0-4+4j: one global_load_dwordx3 each, all in flight together.  Lanes
//          whose 12 bytes would leave the plane take three dword loads from clamped columns and
//          replicate the edge sample with v_perm_b32 instead (wave-uniform branch).
//  rows    per window row: 4 outputs from the 3 dwords with v_alignbyte_b32 + 2 x v_dot4_i32_i8
//          (samples biased by -128), clipped and re-packed into one dword -> 16 dwords hr[]
//  4x4     transposes of hr[] with v_perm_b32 give, per column, the intermediate rows packed four
//          to a dword — the shape the row pass consumed — so the column pass is the same code
//  cols    4 columns x 8 rows, re-packed by row, one dword store per row (16 lanes = 64 bytes)
// ≈ 0.35 instructions per output sample against ≈ 1.7 for an LDS-staged 16x16 tile per wave.
struct RefPlane {
  const unsigned char *p;
  int stride, w, h, pad;
};
struct RefTable {
  RefPlane d[VP9HIP_MAX_REFS][3];
};

// A task of shape W x H: W / 4 lanes per strip, strips of SH rows (4 for the ..x4 shapes, else 8), H / SH strips —
// exactly the lanes its shape needs (with one allocation per WIDTH, sized for the tallest shape, 40 % of the lanes of a
// frame with a VP9 partition had no work, and the 4x4 tasks — 70 % of all — filtered 15 window rows for 4).
template <int W, int H>
struct RegCfg {
  static constexpr int L = W / 4;            // lanes per strip
  static constexpr int SH = H == 4 ? 4 : 8;  // rows per strip
  static constexpr int SPT = H / SH;         // strips per task
  static constexpr int NR = SH + 7;          // window rows of a strip
  static constexpr int NG = (NR + 4) / 4;    // groups of four window rows (the last one padded)
};
constexpr int REG_THREADS = 256;
#ifndef REG_WAVES
#define REG_WAVES 3
#endif

__device__ __forceinline__ unsigned pack4(unsigned o0, unsigned o1, unsigned o2, unsigned o3) {
  return o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
}
// Four outputs = 8 x v_dot4_i32_i8 + 2 x v_ashr_pk_u8_i32, as one hand-scheduled block:
//  * the accumulator seed (128 * sum(taps) + 64: un-bias + rounding) comes from an SGPR in the VOP3P
//    form; the builtin selects the VOP2 v_dot4c form, which needs a v_mov of the seed per output;
//  * v_ashr_pk_u8_i32 (new in gfx950) shifts, saturates to u8 and packs two results into ONE half of
//    the destination and preserves the other half (tools/ashr_pk_probe.hip): low half, then op_sel
//    high half = four clipped samples in two instructions.  (hipcc 7.2 selects the instruction for a
//    clamp-shift-pack of two values by itself but then treats the untouched half as zero — wrong.)
//  * the compiler cannot see into the block, so the gfx90a+ hazard "a dot result read by a different
//    VALU opcode needs 3 wait states" (and "overwritten by one: 4") is met by the order + s_nop 1.
// Inputs come from ordinary VALU instructions and the packed result is written by a non-dot.
__device__ __forceinline__ unsigned dot8x4_clip(unsigned lo0, unsigned lo1, unsigned lo2, unsigned lo3, unsigned hi0,
                                                unsigned hi1, unsigned hi2, unsigned hi3, unsigned flo, unsigned fhi) {
  unsigned r;
  int s0, s1, s2, s3;
  const int seed = 16384 + 64;
  asm("v_dot4_i32_i8 %1, %5, %13, %15\n\t"
      "v_dot4_i32_i8 %2, %6, %13, %15\n\t"
      "v_dot4_i32_i8 %3, %7, %13, %15\n\t"
      "v_dot4_i32_i8 %4, %8, %13, %15\n\t"
      "v_dot4_i32_i8 %1, %9, %14, %1\n\t"
      "v_dot4_i32_i8 %2, %10, %14, %2\n\t"
      "v_dot4_i32_i8 %3, %11, %14, %3\n\t"
      "v_dot4_i32_i8 %4, %12, %14, %4\n\t"
      "s_nop 1\n\t"
      "v_ashr_pk_u8_i32 %0, %1, %2, 7\n\t"
      "v_ashr_pk_u8_i32 %0, %3, %4, 7 op_sel:[0,0,0,1]"
      : "=&v"(r), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
      : "v"(lo0), "v"(lo1), "v"(lo2), "v"(lo3), "v"(hi0), "v"(hi1), "v"(hi2), "v"(hi3), "v"(flo), "v"(fhi), "s"(seed));
  return r;
}
// outputs i = 0..3 from bytes i+S .. i+S+7 of the twelve (biased) bytes {d0, d1, d2}
template <int S>
__device__ __forceinline__ unsigned filt4(unsigned d0, unsigned d1, unsigned d2, unsigned flo, unsigned fhi) {
  unsigned lo[4], hi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int sh = i + S;  // 0..4
    lo[i] = sh == 0 ? d0 : (sh == 4 ? d1 : __builtin_amdgcn_alignbyte(d1, d0, sh & 3));
    hi[i] = sh == 0 ? d1 : (sh == 4 ? d2 : __builtin_amdgcn_alignbyte(d2, d1, sh & 3));
  }
  return dot8x4_clip(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3], flo, fhi);
}
// rows a, b, c, e (four samples each) -> columns c0..c3 (samples of rows a, b, c, e)
__device__ __forceinline__ void transpose4(unsigned a, unsigned b, unsigned c, unsigned e, unsigned &c0, unsigned &c1,
                                           unsigned &c2, unsigned &c3) {
  const unsigned t0 = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
  const unsigned t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
  const unsigned u0 = __builtin_amdgcn_perm(e, c, 0x05010400u);
  const unsigned u1 = __builtin_amdgcn_perm(e, c, 0x07030602u);
  c0 = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
  c1 = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
  c2 = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
  c3 = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
}
// byte k of the result = byte clamp(k + s, 0, 3) of the source dword, s in -3..3
__device__ const unsigned kEdgeSel[7] = { 0x00000000u, 0x01000000u, 0x02010000u, 0x03020100u,
                                          0x03030201u, 0x03030302u, 0x03030303u };

// Probe builds (-DVP9HIP_STAMPS, tools/ only): per wave, the 100 MHz clock at entry and exit and the shader clock at the
// stages of inter_reg_body (window loads issued / arrived / rows done / columns done / stores issued)
#ifdef VP9HIP_STAMPS
__device__ long long g_conv_stamps[8 * 16384];
#define CONV_STAMP(k, v)                                                                                  \
  do {                                                                                                    \
    const int w_ = (int)blockIdx.x * (REG_THREADS / 64) + (int)(threadIdx.x >> 6);                        \
    if ((threadIdx.x & 63) == 0 && w_ < 16384) g_conv_stamps[w_ * 8 + (k)] = (long long)(v);             \
  } while (0)
#define CONV_DRAIN() __builtin_amdgcn_s_waitcnt(0)
extern "C" int vp9hip_debug_conv_stamps(long long *out, int n_waves) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), sizeof(long long) * 8 * (size_t)n_waves) == hipSuccess ? 0 : -1;
}
#else
#define CONV_STAMP(k, v) do { } while (0)
#define CONV_DRAIN() do { } while (0)
#endif

template <int W, int H>
__device__ __forceinline__ void inter_reg_body(int wg, const vp9hip_inter_task *__restrict__ tasks, int n_tasks,
                                               const RefTable &refs, const FrameDev &dstf,
                                               const unsigned *__restrict__ taps) {
  typedef RegCfg<W, H> C;
  constexpr int SH = C::SH, NR = C::NR, NG = C::NG;
  const int gl = wg * REG_THREADS + threadIdx.x;
  const int strip = gl / C::L, j = gl % C::L;
  const int ti = strip / C::SPT, sub = strip % C::SPT;
  const bool active = ti < n_tasks;
  CONV_STAMP(0, wall_clock64());
  CONV_STAMP(1, clock64());
  vp9hip_inter_task t;
  if (active) t = tasks[ti];
  const int plane = active ? t.plane : 0;
  const int filt = active ? (t.flags >> 1) & 7 : 0;
  const int nref = active ? ((t.flags & 1) ? 2 : 1) : 0;
  const int dx = active ? t.dst_x + 4 * j : 0, dy = active ? t.dst_y + sub * SH : 0;
  const int dstride = dstf.stride[plane];
  // awidth is a multiple of 8 and dx of 4: a lane's four columns are visible together or not at all
  const int vis_h = (active && dx < dstf.awidth[plane]) ? min(SH, dstf.aheight[plane] - dy) : 0;
  unsigned char *dst = (unsigned char *)dstf.plane[plane] + (size_t)dy * dstride + dx;
  unsigned k[SH];  // first prediction of a compound strip
#pragma unroll
  for (int y = 0; y < SH; ++y) k[y] = 0;

#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const bool on = r < nref && vis_h > 0;
    if (__builtin_amdgcn_ballot_w64(on) == 0) break;
    if (on) {
      const int px = r ? t.pos_x[1] : t.pos_x[0], py = r ? t.pos_y[1] : t.pos_y[0];
      const int x0 = (px >> 4) + 4 * j, y0 = (py >> 4) + sub * SH;
      const int subx = px & 15, suby = py & 15;
      const RefPlane rp = refs.d[r ? t.ref[1] : t.ref[0]][plane];
      const int xs = x0 - 4;
      unsigned d[NR][3];
      const bool col_ok = xs >= 0 && xs + 12 <= rp.w;
      if (__builtin_amdgcn_ballot_w64(!col_ok) == 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int sy = min(max(y0 - 3 + i, 0), rp.h - 1);
          __builtin_memcpy(d[i], rp.p + (size_t)sy * rp.stride + xs, 12);
        }
      } else {
        int xc[3];
        unsigned sel[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int xq = xs + 4 * q;
          xc[q] = min(max(xq, 0), rp.w - 4);
          sel[q] = kEdgeSel[min(max(xq - xc[q], -3), 3) + 3];
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int sy = min(max(y0 - 3 + i, 0), rp.h - 1);
          const unsigned char *rowp = rp.p + (size_t)sy * rp.stride;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            unsigned v;
            __builtin_memcpy(&v, rowp + xc[q], 4);
            d[i][q] = __builtin_amdgcn_perm(v, v, sel[q]);
          }
        }
      }
      if (r == 0) {
        CONV_STAMP(2, clock64());  // window loads issued (the task record has arrived)
        CONV_DRAIN();
        CONV_STAMP(3, clock64());  // window arrived
      }
      // rows
      unsigned hr[4 * NG];
      {
        const unsigned flo = taps[(filt * 16 + subx) * 2], fhi = taps[(filt * 16 + subx) * 2 + 1];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const unsigned f = filt4<1>(d[i][0] ^ 0x80808080u, d[i][1] ^ 0x80808080u, d[i][2] ^ 0x80808080u, flo, fhi);
          hr[i] = subx == 0 ? d[i][1] : f;
        }
#pragma unroll
        for (int i = NR; i < 4 * NG; ++i) hr[i] = 0;
      }
      if (r == 0) CONV_STAMP(4, clock64());  // rows done
      // columns
      unsigned col[4][NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        transpose4(hr[4 * g], hr[4 * g + 1], hr[4 * g + 2], hr[4 * g + 3], col[0][g], col[1][g], col[2][g], col[3][g]);
#pragma unroll
        for (int c = 0; c < 4; ++c) col[c][g] ^= 0x80808080u;
      }
      const unsigned flo = taps[(filt * 16 + suby) * 2], fhi = taps[(filt * 16 + suby) * 2 + 1];
      unsigned out[SH];
#pragma unroll
      for (int m = 0; m < SH / 4; ++m) {
        unsigned v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = filt4<0>(col[c][m], col[c][m + 1], col[c][m + 2], flo, fhi);
        // v[c] = column c, rows 4m..4m+3 -> back to rows
        transpose4(v[0], v[1], v[2], v[3], out[4 * m], out[4 * m + 1], out[4 * m + 2], out[4 * m + 3]);
      }
#pragma unroll
      for (int y = 0; y < SH; ++y) {
        const unsigned o = suby == 0 ? hr[y + 3] : out[y];
        // second reference of a compound strip: per byte (a + b + 1) >> 1 (vpx_convolve_avg_c)
        const unsigned res = r == 1 ? (k[y] | o) - (((k[y] ^ o) >> 1) & 0x7f7f7f7fu) : o;
        const bool first = nref == 2 && r == 0;
        k[y] = o;
        if (!first && y < vis_h) *(unsigned *)(dst + (size_t)y * dstride) = res;
      }
      if (r == 0) CONV_STAMP(5, clock64());  // first reference done, stores issued
    }
  }
  CONV_STAMP(6, clock64());
  CONV_STAMP(7, wall_clock64());
}

// The thirteen shapes of vp9hip_inter_class in one launch: workgroups [wg_start[k], wg_start[k+1]) serve shape k.
constexpr int REG_SHAPES = VP9HIP_INTER_CLASSES - 1;
struct RegPlan {
  int wg_start[REG_SHAPES + 1];
  int task_start[REG_SHAPES];
  int task_count[REG_SHAPES];
};

__global__ __launch_bounds__(REG_THREADS) __attribute__((amdgpu_waves_per_eu(REG_WAVES, REG_WAVES))) void inter_reg_kernel(const vp9hip_inter_task *__restrict__ tasks,
                                                                RegPlan plan, RefTable refs, FrameDev dstf,
                                                                const unsigned *__restrict__ taps) {
  const int b = blockIdx.x;
  int k = 0;
  while (k < REG_SHAPES - 1 && b >= plan.wg_start[k + 1]) ++k;
  const int wg = xcd_order(b, plan.wg_start[k], plan.wg_start[k + 1]);
  const vp9hip_inter_task *tk = tasks + plan.task_start[k];
  const int n = plan.task_count[k];
  switch (k) {
    case 0: inter_reg_body<4, 4>(wg, tk, n, refs, dstf, taps); break;
    case 1: inter_reg_body<4, 8>(wg, tk, n, refs, dstf, taps); break;
    case 2: inter_reg_body<8, 4>(wg, tk, n, refs, dstf, taps); break;
    case 3: inter_reg_body<8, 8>(wg, tk, n, refs, dstf, taps); break;
    case 4: inter_reg_body<8, 16>(wg, tk, n, refs, dstf, taps); break;
    case 5: inter_reg_body<16, 8>(wg, tk, n, refs, dstf, taps); break;
    case 6: inter_reg_body<16, 16>(wg, tk, n, refs, dstf, taps); break;
    case 7: inter_reg_body<16, 32>(wg, tk, n, refs, dstf, taps); break;
    case 8: inter_reg_body<32, 16>(wg, tk, n, refs, dstf, taps); break;
    case 9: inter_reg_body<32, 32>(wg, tk, n, refs, dstf, taps); break;
    case 10: inter_reg_body<32, 64>(wg, tk, n, refs, dstf, taps); break;
    case 11: inter_reg_body<64, 32>(wg, tk, n, refs, dstf, taps); break;
    default: inter_reg_body<64, 64>(wg, tk, n, refs, dstf, taps); break;
  }
}

template <int W, int H>
int reg_wgs(int n) {
  constexpr int per_wg = REG_THREADS / RegCfg<W, H>::L;  // strips per workgroup
  return (int)(((long long)n * RegCfg<W, H>::SPT + per_wg - 1) / per_wg);
}

// ---------------------------------------------------------------------------------------------
// Register path for 16-bit samples (bit depth 8 / 10 / 12 in uint16 frames), unscaled references: the form of
// inter_reg_body with two samples per dword.  A lane owns TWO adjacent output columns (one destination dword per
// row) of a strip of 4 or 8 rows; a task of width W takes W / 2 neighbouring lanes per strip.
//  load    NR = strip + 7 window rows of ten samples (x0-3 .. x0+6) = five dwords each, row index clamped to the plane
//  rows    output 0 from dwords 0..3, output 1 from the same ten samples shifted by one (v_alignbit_b32 x 4):
//          2 x 4 v_dot2_i32_i16, clip, pack -> one dword per window row.  The phase-0 kernel {0,0,0,128,0,0,0,0}
//          fits i16, so the identity needs no special case; the clip BETWEEN the passes is normative
//          (vpx_convolve.c:355-375, highbd_convolve).
//  cols    per column the window rows pair up two to a dword (v_perm_b32; even and odd starts), 4 dots per output
//  out     (a + b + 1) >> 1 with the first prediction for a compound strip, one dword store per row
// No LDS at all.  K-conv 2160p 10-bit 64x64: 36.3 us against 59.3 us for the LDS-staged tile form it replaced
// (window staging, transposed intermediate, fences per stage), 16x16: 45.8 against 70.1 us.
template <int W, int H>
struct Reg16Cfg {
  static constexpr int L = W / 2;            // lanes per strip
  static constexpr int SH = H == 4 ? 4 : 8;  // rows per strip
  static constexpr int SPT = H / SH;         // strips per task
  static constexpr int NR = SH + 7;          // window rows of a strip
};

template <int W, int H>
__device__ __forceinline__ void inter_reg16_body(int wg, const vp9hip_inter_task *__restrict__ tasks, int n_tasks,
                                                 const RefSet &refs, const FrameDev &dstf) {
  typedef Reg16Cfg<W, H> C;
  constexpr int SH = C::SH, NR = C::NR;
  const int gl = wg * REG_THREADS + threadIdx.x;
  const int strip = gl / C::L, j = gl % C::L;
  const int ti = strip / C::SPT, sub = strip % C::SPT;
  const bool active = ti < n_tasks;
  CONV_STAMP(0, wall_clock64());
  CONV_STAMP(1, clock64());
  vp9hip_inter_task t;
  if (active) t = tasks[ti];
  const int plane = active ? t.plane : 0;
  const int filt = active ? (t.flags >> 1) & 7 : 0;
  const int nref = active ? ((t.flags & 1) ? 2 : 1) : 0;
  const int dx = active ? t.dst_x + 2 * j : 0, dy = active ? t.dst_y + sub * SH : 0;
  const int dstride = dstf.stride[plane];
  const int maxv = (1 << dstf.bit_depth) - 1;
  // awidth is a multiple of 8 and dx of 2: a lane's two columns are visible together or not at all
  const int vis_h = (active && dx < dstf.awidth[plane]) ? min(SH, dstf.aheight[plane] - dy) : 0;
  uint16_t *dst = (uint16_t *)dstf.plane[plane] + (size_t)dy * dstride + dx;
  const uint4 *taps = (const uint4 *)&kFilters[0][0][0];  // (f0,f1) (f2,f3) (f4,f5) (f6,f7) per (filter, phase)
  unsigned k[SH];  // first prediction of a compound strip
#pragma unroll
  for (int y = 0; y < SH; ++y) k[y] = 0;

#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
    const bool on = r < nref && vis_h > 0;
    if (__builtin_amdgcn_ballot_w64(on) == 0) break;
    if (on) {
      const int px = r ? t.pos_x[1] : t.pos_x[0], py = r ? t.pos_y[1] : t.pos_y[0];
      const int x0 = (px >> 4) + 2 * j, y0 = (py >> 4) + sub * SH;
      const FrameDev &rf = refs.f[r ? t.ref[1] : t.ref[0]];
      const uint16_t *src = (const uint16_t *)rf.plane[plane];
      const int sstride = rf.stride[plane], fw = rf.width[plane], fh = rf.height[plane];
      const uint4 fx = taps[filt * 16 + (px & 15)], fy = taps[filt * 16 + (py & 15)];
      const int xs = x0 - 3;
      unsigned hr[NR + 1];
      const bool col_ok = xs >= 0 && xs + 10 <= fw;
      if (__builtin_amdgcn_ballot_w64(!col_ok) == 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int sy = min(max(y0 - 3 + i, 0), fh - 1);
          unsigned d[5];
          __builtin_memcpy(d, src + (size_t)sy * sstride + xs, 20);
          const int o0 = dot8_16(d[0], d[1], d[2], d[3], fx, maxv);
          const int o1 = dot8_16(__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                                 __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d[4], d[3], 16), fx, maxv);
          hr[i] = (unsigned)o0 | ((unsigned)o1 << 16);
        }
      } else {  // a frame edge inside the window: every sample from its clamped column (libvpx's border emulation)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int sy = min(max(y0 - 3 + i, 0), fh - 1);
          const uint16_t *rowp = src + (size_t)sy * sstride;
          unsigned d[5];
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const unsigned a = rowp[min(max(xs + 2 * q, 0), fw - 1)], b = rowp[min(max(xs + 2 * q + 1, 0), fw - 1)];
            d[q] = a | (b << 16);
          }
          const int o0 = dot8_16(d[0], d[1], d[2], d[3], fx, maxv);
          const int o1 = dot8_16(__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                                 __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(d[4], d[3], 16), fx, maxv);
          hr[i] = (unsigned)o0 | ((unsigned)o1 << 16);
        }
      }
      hr[NR] = 0;
      // columns: P[c][m] = rows (2m, 2m+1) of column c, Q[c][m] = rows (2m+1, 2m+2)
      unsigned out[SH];
#pragma unroll
      for (int y = 0; y < SH; ++y) out[y] = 0;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned sel = c ? 0x07060302u : 0x05040100u;
        unsigned P[(NR + 1) / 2], Q[NR / 2];
#pragma unroll
        for (int m = 0; m < (NR + 1) / 2; ++m) P[m] = __builtin_amdgcn_perm(hr[2 * m + 1], hr[2 * m], sel);
#pragma unroll
        for (int m = 0; m < NR / 2; ++m) Q[m] = __builtin_amdgcn_perm(hr[2 * m + 2], hr[2 * m + 1], sel);
#pragma unroll
        for (int y = 0; y < SH; ++y) {
          const int m = y >> 1;
          const int o = (y & 1) ? dot8_16(Q[m], Q[m + 1], Q[m + 2], Q[m + 3], fy, maxv) : dot8_16(P[m], P[m + 1], P[m + 2], P[m + 3], fy, maxv);
          out[y] |= (unsigned)o << (16 * c);
        }
      }
#pragma unroll
      for (int y = 0; y < SH; ++y) {
        // second reference of a compound strip: per sample (a + b + 1) >> 1 (vpx_highbd_convolve_avg_c); samples < 2^15,
        // so the two halves of the dword do not carry into each other
        const unsigned res = r == 1 ? ((k[y] + out[y] + 0x00010001u) >> 1) & 0x7fff7fffu : out[y];
        const bool first = nref == 2 && r == 0;
        k[y] = out[y];
        if (!first && y < vis_h) *(unsigned *)(dst + (size_t)y * dstride) = res;
      }
    }
  }
}

#ifndef REG16_WAVES
#define REG16_WAVES 3  // (4 fits 128 registers only with 45 spills: 37.7 against 36.3 us at K-conv 2160p 10-bit 64x64)
#endif
__global__ __launch_bounds__(REG_THREADS) __attribute__((amdgpu_waves_per_eu(REG16_WAVES, REG16_WAVES))) void inter_reg16_kernel(const vp9hip_inter_task *__restrict__ tasks,
                                                                  RegPlan plan, RefSet refs, FrameDev dstf) {
  const int b = blockIdx.x;
  int k = 0;
  while (k < REG_SHAPES - 1 && b >= plan.wg_start[k + 1]) ++k;
  const int wg = xcd_order(b, plan.wg_start[k], plan.wg_start[k + 1]);
  const vp9hip_inter_task *tk = tasks + plan.task_start[k];
  const int n = plan.task_count[k];
  switch (k) {
    case 0: inter_reg16_body<4, 4>(wg, tk, n, refs, dstf); break;
    case 1: inter_reg16_body<4, 8>(wg, tk, n, refs, dstf); break;
    case 2: inter_reg16_body<8, 4>(wg, tk, n, refs, dstf); break;
    case 3: inter_reg16_body<8, 8>(wg, tk, n, refs, dstf); break;
    case 4: inter_reg16_body<8, 16>(wg, tk, n, refs, dstf); break;
    case 5: inter_reg16_body<16, 8>(wg, tk, n, refs, dstf); break;
    case 6: inter_reg16_body<16, 16>(wg, tk, n, refs, dstf); break;
    case 7: inter_reg16_body<16, 32>(wg, tk, n, refs, dstf); break;
    case 8: inter_reg16_body<32, 16>(wg, tk, n, refs, dstf); break;
    case 9: inter_reg16_body<32, 32>(wg, tk, n, refs, dstf); break;
    case 10: inter_reg16_body<32, 64>(wg, tk, n, refs, dstf); break;
    case 11: inter_reg16_body<64, 32>(wg, tk, n, refs, dstf); break;
    default: inter_reg16_body<64, 64>(wg, tk, n, refs, dstf); break;
  }
}

template <int W, int H>
int reg16_wgs(int n) {
  constexpr int per_wg = REG_THREADS / Reg16Cfg<W, H>::L;  // strips per workgroup
  return (int)(((long long)n * Reg16Cfg<W, H>::SPT + per_wg - 1) / per_wg);
}

int upload_taps(vp9hip_ctx *ctx) {
  if (ctx->d_taps) return VP9HIP_OK;
  unsigned packed[5][16][2];
  for (int f = 0; f < 5; ++f)
    for (int p = 0; p < 16; ++p)
      for (int half = 0; half < 2; ++half) {
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) v |= (unsigned)((uint8_t)(int8_t)kFiltersHost[f][p][half * 4 + k]) << (8 * k);
        packed[f][p][half] = v;
      }
  VP9HIP_CHECK(ctx, hipMalloc(&ctx->d_taps, sizeof(packed)));
  VP9HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_taps, packed, sizeof(packed), hipMemcpyHostToDevice, ctx->stream));
  VP9HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return VP9HIP_OK;
}

}  // namespace

extern "C" int vp9hip_inter_pred_batch(vp9hip_ctx *ctx, const vp9hip_inter_task *d_tasks,
                                       const int32_t class_count[VP9HIP_INTER_CLASSES], const vp9hip_frame *refs, int n_refs,
                                       const vp9hip_frame *dst) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  if (!d_tasks || !class_count || !refs || n_refs <= 0 || n_refs > VP9HIP_MAX_REFS || !frame_ok(dst))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_inter_pred_batch: bad argument");
  RefSet rs;
  memset(&rs, 0, sizeof(rs));
  for (int i = 0; i < n_refs; ++i) {
    if (!frame_ok(&refs[i]) || refs[i].hbd != dst->hbd || refs[i].bit_depth != dst->bit_depth)
      VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_inter_pred_batch: reference %d does not match the destination format", i);
    rs.f[i] = to_dev(&refs[i]);
  }
  int fast_total = 0;
  for (int i = 0; i < VP9HIP_INTER_CLASSES; ++i) {
    if (class_count[i] < 0) VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_inter_pred_batch: negative class count");
    if (i < VP9HIP_INTER_CLASSES - 1) fast_total += class_count[i];
  }
  const FrameDev d = to_dev(dst);
  int rc;
  if (fast_total && !dst->hbd && (rc = upload_taps(ctx))) return rc;
  const vp9hip_inter_task *p = d_tasks + fast_total;
  // the register path replicates edge samples from whole dwords: a reference plane narrower than one
  // dword (frames under 8 samples wide) sends every task through the generic kernel instead
  bool tiny_ref = false;
  for (int i = 0; i < n_refs; ++i)
    for (int pl = 0; pl < 3; ++pl)
      if (refs[i].plane[pl] && refs[i].width[pl] < 4) tiny_ref = true;
  if (fast_total && !dst->hbd && tiny_ref) {
    hipLaunchKernelGGL(inter_pred_kernel<uint8_t>, dim3(fast_total), dim3(64), 0, ctx->stream, d_tasks, fast_total, rs, d);
    VP9HIP_CHECK(ctx, hipGetLastError());
  } else if (fast_total && dst->hbd) {
    RegPlan plan;
    const int wgs[REG_SHAPES] = { reg16_wgs<4, 4>(class_count[0]),    reg16_wgs<4, 8>(class_count[1]),   reg16_wgs<8, 4>(class_count[2]),
                                  reg16_wgs<8, 8>(class_count[3]),    reg16_wgs<8, 16>(class_count[4]),  reg16_wgs<16, 8>(class_count[5]),
                                  reg16_wgs<16, 16>(class_count[6]),  reg16_wgs<16, 32>(class_count[7]), reg16_wgs<32, 16>(class_count[8]),
                                  reg16_wgs<32, 32>(class_count[9]),  reg16_wgs<32, 64>(class_count[10]), reg16_wgs<64, 32>(class_count[11]),
                                  reg16_wgs<64, 64>(class_count[12]) };
    int acc_w = 0, acc_t = 0;
    for (int k = 0; k < REG_SHAPES; ++k) {
      plan.wg_start[k] = acc_w;
      plan.task_start[k] = acc_t;
      plan.task_count[k] = class_count[k];
      acc_w += wgs[k];
      acc_t += class_count[k];
    }
    plan.wg_start[REG_SHAPES] = acc_w;
    hipLaunchKernelGGL(inter_reg16_kernel, dim3(acc_w), dim3(REG_THREADS), 0, ctx->stream, d_tasks, plan, rs, d);
    VP9HIP_CHECK(ctx, hipGetLastError());
  } else if (fast_total) {
    RegPlan plan;
    const int wgs8[REG_SHAPES] = { reg_wgs<4, 4>(class_count[0]),    reg_wgs<4, 8>(class_count[1]),   reg_wgs<8, 4>(class_count[2]),
                                   reg_wgs<8, 8>(class_count[3]),    reg_wgs<8, 16>(class_count[4]),  reg_wgs<16, 8>(class_count[5]),
                                   reg_wgs<16, 16>(class_count[6]),  reg_wgs<16, 32>(class_count[7]), reg_wgs<32, 16>(class_count[8]),
                                   reg_wgs<32, 32>(class_count[9]),  reg_wgs<32, 64>(class_count[10]), reg_wgs<64, 32>(class_count[11]),
                                   reg_wgs<64, 64>(class_count[12]) };
    int acc_w = 0, acc_t = 0;
    for (int k = 0; k < REG_SHAPES; ++k) {
      plan.wg_start[k] = acc_w;
      plan.task_start[k] = acc_t;
      plan.task_count[k] = class_count[k];
      acc_w += wgs8[k];
      acc_t += class_count[k];
    }
    plan.wg_start[REG_SHAPES] = acc_w;
    RefTable rt;
    memset(&rt, 0, sizeof(rt));
    for (int i = 0; i < n_refs; ++i)
      for (int pl = 0; pl < 3; ++pl) {
        rt.d[i][pl].p = (const unsigned char *)refs[i].plane[pl];
        rt.d[i][pl].stride = refs[i].stride[pl];
        rt.d[i][pl].w = refs[i].width[pl];
        rt.d[i][pl].h = refs[i].height[pl];
      }
    hipLaunchKernelGGL(inter_reg_kernel, dim3(acc_w), dim3(REG_THREADS), 0, ctx->stream, d_tasks, plan, rt, d,
                       (const unsigned *)ctx->d_taps);
    VP9HIP_CHECK(ctx, hipGetLastError());
  }
  const int n_gen = class_count[VP9HIP_INTER_CLASSES - 1];
  if (n_gen > 0) {
    if (dst->hbd)
      hipLaunchKernelGGL(inter_pred_kernel<uint16_t>, dim3(n_gen), dim3(64), 0, ctx->stream, p, n_gen, rs, d);
    else
      hipLaunchKernelGGL(inter_pred_kernel<uint8_t>, dim3(n_gen), dim3(64), 0, ctx->stream, p, n_gen, rs, d);
    VP9HIP_CHECK(ctx, hipGetLastError());
  }
  return VP9HIP_OK;
}
