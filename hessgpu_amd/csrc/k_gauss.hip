// k_gauss.hip -- Gaussian scale-space kernels for gfx950 (MI355X).
//
// Replaces FilterH<FW>/FilterV<FW> (two global passes through a scratch buffer, ProgramCU.cu:117-231,
// 455-512), DownsampleKernel (ProgramCU.cu:312-326) and the host-side pixel conversion
// (GLTexImage.cpp:802-916).  By bytes HBM-bound (4 B read + 4 B written per pixel and level, + 4/12 B for
// the fused planes); measured, vector-ALU issue is the contended resource.
//
// gauss_kernel (the shipped form): a workgroup (256 threads = 4 wavefronts) produces one 64x32 tile of one level:
//   stage 1  source rows [y0-R, y0+32+R) x cols [x0-R4, x0+64+R4) -> LDS `s`, 16-byte global loads,
//            borders replicated exactly as the reference clamps its fetch index;
//   stage 1b (HESS) det-Hessian*sigma^4 and (gradient/2, theta) of the SOURCE level for the tile from the staged
//            window, in the same launch (ComputeHessian_Kernel, ProgramCU.cu:523-595): the level is not re-read;
//   stage 2  horizontal pass LDS->LDS: a thread produces 8 adjacent outputs from a register window
//            (ds_read_b128, row stride = 4 mod 8 dwords: conflict-free);
//   stage 3  vertical pass LDS->HBM: a thread produces 4 rows x 2 columns (ds_read_b64, 8-byte
//            coalesced stores, 512 contiguous bytes per wavefront); the launch that produces the down-sampling level
//            also stores its even rows and columns as level 0 of the next octave.
// The column-strip march of round 3 (every source row filtered horizontally once; bit-identical, 21 % fewer vector
// instructions, 50 % slower) is kept as a patch against this file: profiles/r03_experiments/gauss_march.diff.
// Per output the taps are accumulated in the reference's order: v = 0; v = fma(x_i, k_i, v), i=0..FW-1.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "hess_dev.h"
#include "hess_devmath.h"
#include "hess_planes.h"

namespace hess {

namespace {

constexpr int TW = 64, TH = 32, NT = 256;
struct GaussArgs {
  const float* src;
  const uint8_t* src_u8;
  long long src_pitch;       // elements (float) or bytes (u8) between source rows
  long long src_img_stride;  // same unit, between images
  float* dst;                // [batch][h][w]
  int w, h;
  int tiles_x, tiles_y, batch;
  // fused det-Hessian / gradient of the SOURCE level (HESS variant): [batch][h][w] planes
  float* deth_src;
  float2* got_src;   // may be null (levels 0 and dog+1 have no gradient plane)
  float norm_src;    // sigma^4 of the source level
  // level 0 of the next octave = the produced level point-sampled at the even rows and columns (DownsampleKernel,
  // ProgramCU.cu:312-326: dst(x, y) = src(min(2x, w-1), 2y)); null unless this launch produces the down-sampling level
  float* decim_dst;  // [batch][dh][dw]
  int dw, dh;
  // TOP tiles (the octave's top level, nobody's source inside the pyramid): det-Hessian of the PRODUCED level, computed
  // from the output tile (+ a one-pixel halo) while it is in LDS -- the level itself is never written to HBM unless
  // `dst` is given (hess_debug_keep_levels); norm_dst = sigma^4 of the produced level
  float* deth_dst;
  float norm_dst;
  // ... and, on the side, the buffers the detection stages expect zeroed (no fill launch in the chain): 16-byte words
  uint4* zero;
  long long zero_n16;
  Taps taps;
};
// FIRST tiles (octave 0, u8 pixels): `taps0` produces level 0 from the pixels, GaussArgs::taps level 1 from level 0; level 0
// lives in LDS only (dst0: written as well on request), det-H of level 0 goes to deth_src as for any source level
struct FirstArgs {
  Taps taps0;
  float* dst0;
};

// A 16-byte LDS read the compiler keeps whole.  The horizontal passes read a row window of 16-byte groups of which the
// first and last are only partly used; left to itself the compiler narrows the loads to what the packed FMAs take (8-byte
// pairs at odd dword offsets: ds_read2_b32 / ds_read2_b64, 128 B per clock and banked modulo 32 dwords, where this kernel's
// 32-byte lane pitch collides three and four ways).  The empty asm makes the whole group a used value: ds_read_b128,
// 256 B per clock, banked modulo 64 (profiles/r06_experiments/gauss_lds.txt).
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NGRP>
__device__ __forceinline__ void lds_read_groups(const float* p, float (&win)[NGRP * 4]) {
  v4f q[NGRP];
#pragma unroll
  for (int i = 0; i < NGRP; i++) q[i] = *reinterpret_cast<const v4f*>(p + 4 * i);
#pragma unroll
  for (int i = 0; i < NGRP; i++) asm volatile("" : "+v"(q[i]));
#pragma unroll
  for (int i = 0; i < NGRP; i++) { win[4 * i] = q[i].x; win[4 * i + 1] = q[i].y; win[4 * i + 2] = q[i].z; win[4 * i + 3] = q[i].w; }
}

__device__ __forceinline__ float gtex1(const float* p, int n, int i) { return (i < 0 || i >= n) ? 0.0f : p[i]; }

// LDS floats of one tile: (32 + 2R) staged rows of 64 + 2*R4 (+4 pad) columns; a TOP tile stages one more row above
// and below and a column halo of R + 1 (its output tile has a one-pixel halo), and keeps the wrap-around columns of
// the produced level behind the staged rows (2 x (34 + 2R) horizontally filtered values, 2 x 34 results)
// A FIRST tile (R0 > 0: level 0 from u8 pixels in LDS, then level 1 from it) has the pixel window instead -- (32 + 2R + 2R0)
// rows of 64 + 2 (R4 + R0 rounded up to 4) (+4) columns, inside which the level-0 window takes shape in place -- and the wrap
// columns of level 0.
#ifndef HESS_GAUSS_PAD
#define HESS_GAUSS_PAD 4
#endif
#ifndef HESS_FIRST_PAD
#define HESS_FIRST_PAD 4
#endif
constexpr int kRowPad = HESS_GAUSS_PAD;        // dwords between the staged rows of a tile (a multiple of 4: 16-byte accesses)
constexpr int kFirstRowPad = HESS_FIRST_PAD;   // ... of a FIRST tile's pixel window
template <int R, bool TOP = false, int R0 = 0>
constexpr int gauss_tile_lds() {
  return TOP ? (TH + 2 * R + 2) * (TW + 2 * ((R + 1 + 3) & ~3) + kRowPad) + ((2 * (TH + 2 + 2 * R) + 2 * (TH + 2) + 3) & ~3)
             : (R0 > 0 ? (TH + 2 * R + 2 * R0) * (TW + 2 * (((R + 3) & ~3) + ((R0 + 3) & ~3)) + kFirstRowPad) + ((2 * (TH + 2 + 2 * R0) + 2 * (TH + 2) + 3) & ~3)
                       : (TH + 2 * R) * (TW + 2 * ((R + 3) & ~3) + kRowPad));
}

// One 64x32 tile of one level: the body of gauss_kernel, and of either half of gauss_pair_kernel.  `block` = the
// workgroup's index among the launch's (or the half's) workgroups, `s` = gauss_tile_lds<R>() floats of LDS.
//
// TOP (the octave's top level: nobody's source inside the pyramid, read again only to make its own det-Hessian): the
// tile is produced with a one-pixel halo (66 x 34), kept in LDS, det-Hessian*sigma^4 of the PRODUCED level is computed
// from it there and stored; the level itself goes to HBM only when a.dst is given (parity tests).  8 bytes per
// octave-pixel of HBM traffic and the standalone det-H launch are gone.  The reference addresses det-H neighbours by
// 1-D index (ProgramCU.cu:523-595): column -1 of a row is the LAST column of the row above, column w the FIRST of the
// row below -- values of the far side of the image, which tiles at the left / right image border recompute for their
// 34 rows with the same tap chains (a one-column horizontal + vertical pass from HBM: `wrap` below).
//
// FIRST (R0 > 0; octave 0 of a u8 image): level 0 -- nobody's input but level 1's -- is produced from the pixels INSIDE this
// launch, on the tile grown by level 1's radius, and never written to HBM (a.dst0 only on request): the launch that made
// level 0 (4 B written per pixel, read back here) is gone.  Borders exactly as two launches have them: the pixel window is
// staged with clamped rows and replicated edge columns; level 0 at a window position outside the image is level 0 at the
// clamped position (what level 1's clamped fetch would read), i.e. the horizontally filtered values of outside columns and
// the level-0 values of outside rows are copies of the edge ones; det-H of level 0 takes its 1-D wrap columns from a
// one-column pass over the pixels (as TOP tiles do for theirs).
template <int R, bool U8, bool HESS, bool TOP = false, int R0 = 0>
__device__ __forceinline__ void gauss_tile(const GaussArgs& a, float* __restrict__ s, const int block, const int nblocks = 0,
                                           const FirstArgs* fa = nullptr) {
  constexpr bool FIRST = R0 > 0;
  static_assert(!FIRST || (U8 && HESS && !TOP), "a FIRST tile: u8 pixels in, level 1 + det-H of level 0 out");
  constexpr int FW = 2 * R + 1;
  constexpr int RTOP = TOP ? 1 : 0;    // halo of the output tile
  constexpr int R4 = (R + RTOP + 3) & ~3;
  constexpr int OFF = R4 - R;
  constexpr int SW = TW + 2 * R4;
  constexpr int SWP = SW + kRowPad;    // SW % 8 == 0 -> row stride = 4 (mod 8) dwords
  constexpr int ROWS = TH + 2 * R + 2 * RTOP;
  constexpr int NG = SW / 4;           // 16-byte groups per staged row
  constexpr int NV = (OFF + 8 + 2 * R + 3) / 4;
  constexpr int NWH = TH + 2 + 2 * R;  // TOP: horizontally filtered values per wrap column
  // FIRST: the pixel window A behind the level-0 window s
  constexpr int R0P = (R0 + 3) & ~3;           // column halo of level 0's horizontal pass, in whole 16-byte groups
  constexpr int AROWS = ROWS + 2 * R0, ASW = SW + 2 * R0P, ASWP = ASW + kFirstRowPad, ANG = ASW / 4;
  constexpr int NWH0 = TH + 2 + 2 * R0;
  static_assert((FIRST ? AROWS * ASWP + ((2 * NWH0 + 2 * (TH + 2) + 3) & ~3)
                       : ROWS * SWP + (TOP ? ((2 * NWH + 2 * (TH + 2) + 3) & ~3) : 0)) == gauss_tile_lds<R, TOP, R0>(), "LDS size");
  // The window of the SOURCE level the passes below work on: row stride LSWP, first column at ws.  FIRST: the level-0 window
  // is computed in place inside the pixel window (its rows are the pixel rows, its columns start R0P in; stride = 4 mod 8 too).
  constexpr int LSWP = FIRST ? ASWP : SWP;
  float* const ws = FIRST ? s + R0P : s;
  static_assert(TW - 8 + 4 * NV <= SW, "register windows of the horizontal pass stay inside the staged row");
  static_assert(!TOP || (!U8 && HESS), "a top level has a float source with fused planes");

  const int tid = threadIdx.x;
  const int w = a.w, h = a.h;
  if (TOP && a.zero) {  // (block-uniform) the detection stages' zeroed buffers, a slice per workgroup; stores only
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = (long long)block * NT + tid; i < a.zero_n16; i += (long long)nblocks * NT) a.zero[i] = z;
  }
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), so
  // XCD k takes the k-th contiguous eighth of the tile sequence (x fastest, then y, then image) and
  // neighbouring tiles -- which share halo rows/columns -- are served by the same 4 MiB L2.
  const int ntile = a.tiles_x * a.tiles_y * a.batch;
  const int per_xcd = (ntile + 7) >> 3;
  const int tile = (block & 7) * per_xcd + (block >> 3);
  if (tile >= ntile) return;
  const int tz = tile / (a.tiles_x * a.tiles_y);
  const int trem = tile - tz * (a.tiles_x * a.tiles_y);
  const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
  const int x0 = txi * TW, y0 = tyi * TH;
  const long long img = tz;

  float* const wrap0H = s + AROWS * ASWP;  // FIRST: [2][NWH0] horizontally filtered pixels of the wrap columns
  float* const wrap0V = wrap0H + 2 * NWH0;               //        [2][TH + 2] level 0 at column w-1 (rows y0-2 ..) / column 0 (rows y0 ..)
  if (FIRST) {
    float* const A = s;
    // ---- stage 0a: pixels -> A: rows y0-R-R0 .. (clamped), columns x0-R4-R0P .. (edge columns replicated) ----
    {
      constexpr int ANIT = (AROWS * ANG + NT - 1) / NT;
      float4 st0[ANIT];
#pragma unroll
      for (int it = 0; it < ANIT; it++) {
        int g = it * NT + tid;
        g = g < AROWS * ANG ? g : AROWS * ANG - 1;
        const int r = g / ANG, gq = g - r * ANG;
        int y = y0 - R - R0 + r;
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        const int x = x0 - R4 - R0P + gq * 4;
        const int xs = x < 0 ? 0 : (x >= w ? w - 4 : x);
        const uint8_t* row = a.src_u8 + img * a.src_img_stride + (long long)y * a.src_pitch;
        const uchar4 b = *reinterpret_cast<const uchar4*>(row + xs);
        st0[it] = make_float4(dm_u8_unit((float)b.x), dm_u8_unit((float)b.y), dm_u8_unit((float)b.z), dm_u8_unit((float)b.w));
      }
#pragma unroll
      for (int it = 0; it < ANIT; it++) {
        int g = it * NT + tid;
        g = g < AROWS * ANG ? g : AROWS * ANG - 1;
        const int r = g / ANG, gq = g - r * ANG;
        *reinterpret_cast<float4*>(&A[r * ASWP + gq * 4]) = st0[it];
      }
      if (x0 - R4 - R0P < 0 || x0 + TW + R4 + R0P > w) {  // block-uniform
        __syncthreads();
        for (int g = tid; g < AROWS * ANG; g += NT) {
          const int r = g / ANG, gq = g - r * ANG;
          const int x = x0 - R4 - R0P + gq * 4;
          if (x < 0 || x >= w) {
            const float e = A[r * ASWP + gq * 4 + (x < 0 ? 0 : 3)];
            *reinterpret_cast<float4*>(&A[r * ASWP + gq * 4]) = make_float4(e, e, e, e);
          }
        }
      }
    }
    __syncthreads();
    // ---- level 0 at the far side of the image for det-H's 1-D neighbour addressing (tiles at the left / right border):
    // one column, from the pixels, with level 0's own chains ----
    const bool w0_left = x0 == 0, w0_right = x0 + TW >= w;  // block-uniform
    if (w0_left || w0_right) {
      const int side = tid >> 7, t = tid & 127;
      const bool mine = side == 0 ? w0_left : w0_right;
      static_assert(NWH0 <= 128, "one thread per filtered row and side");
      if (mine && t < NWH0) {
        int yy = (side == 0 ? y0 - 2 : y0) - R0 + t;
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
        const int c = side == 0 ? w - 1 : 0;
        const uint8_t* row = a.src_u8 + img * a.src_img_stride + (long long)yy * a.src_pitch;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 2 * R0 + 1; i++) {
          int xx = c - R0 + i;
          xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
          v = fmaf(dm_u8_unit((float)row[xx]), fa->taps0.k[i], v);
        }
        wrap0H[side * NWH0 + t] = v;
      }
      __syncthreads();
      if (mine && t < TH + 2) {
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 2 * R0 + 1; i++) v = fmaf(wrap0H[side * NWH0 + t + i], fa->taps0.k[i], v);
        wrap0V[side * (TH + 2) + t] = v;
      }
    }
    // ---- stage 0b: horizontal pass of level 0, in place: window columns (A columns R0P .. R0P+SW) <- taps0 ----
    {
      constexpr int FW0 = 2 * R0 + 1, OFF0 = R0P - R0, NV0 = (OFF0 + 8 + 2 * R0 + 3) / 4, TPR = SW / 8;
      constexpr int NT0 = (AROWS * TPR + NT - 1) / NT;
      static_assert(SW % 8 == 0 && SW - 8 + 4 * NV0 <= ASW, "register windows stay inside the pixel rows");
      float acc0[NT0][8];
#pragma unroll
      for (int k = 0; k < NT0; k++) {
        const int task = tid + k * NT;
        if (task < AROWS * TPR) {
          const int r = task / TPR, xb = (task - r * TPR) * 8;
          float win[NV0 * 4];
          lds_read_groups<NV0>(&A[r * ASWP + xb], win);
#pragma unroll
          for (int j = 0; j < 8; j++) acc0[k][j] = 0.0f;
#pragma unroll
          for (int i = 0; i < FW0; i++) {
            const float ki = fa->taps0.k[i];
#pragma unroll
            for (int j = 0; j < 8; j++) acc0[k][j] = fmaf(win[OFF0 + j + i], ki, acc0[k][j]);  // ProgramCU.cu:152
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NT0; k++) {
        const int task = tid + k * NT;
        if (task < AROWS * TPR) {
          const int r = task / TPR, xb = (task - r * TPR) * 8;
          *reinterpret_cast<float4*>(&A[r * ASWP + R0P + xb]) = make_float4(acc0[k][0], acc0[k][1], acc0[k][2], acc0[k][3]);
          *reinterpret_cast<float4*>(&A[r * ASWP + R0P + xb + 4]) = make_float4(acc0[k][4], acc0[k][5], acc0[k][6], acc0[k][7]);
        }
      }
      __syncthreads();
      if (x0 - R4 < 0 || x0 + TW + R4 > w) {  // block-uniform: window columns outside the image take the edge column's value
        const int c_lo = R0P + (0 - (x0 - R4)), c_hi = R0P + (w - 1 - (x0 - R4));  // A columns of image columns 0 and w-1
        for (int idx = tid; idx < AROWS * SW; idx += NT) {
          const int r = idx / SW, c = R0P + (idx - r * SW);
          if (c < c_lo) A[r * ASWP + c] = A[r * ASWP + c_lo];
          else if (c > c_hi) A[r * ASWP + c] = A[r * ASWP + c_hi];
        }
        __syncthreads();
      }
    }
    // ---- stage 0c: vertical pass of level 0, in place (window row r <- pixel-window rows r .. r+2 R0; all columns are read
    // into registers before the first row is overwritten): the level-0 window, as stage 1 stages it for any other level ----
    {
      constexpr int FW0 = 2 * R0 + 1, NRG0 = (ROWS + 3) / 4, NCP0 = SW / 2, NTV0 = NRG0 * NCP0, KV0 = (NTV0 + NT - 1) / NT;
      float2 out[KV0][4];
#pragma unroll
      for (int k = 0; k < KV0; k++) {
        const int t = tid + k * NT;
        if (t < NTV0) {
          const int rg = t / NCP0, c = 2 * (t - rg * NCP0);
          const int r0 = min(4 * rg, ROWS - 4);  // (the last group is moved up onto the window's end: same values)
          float2 col[4 + 2 * R0];
#pragma unroll
          for (int i = 0; i < 4 + 2 * R0; i++) col[i] = *reinterpret_cast<const float2*>(&ws[(r0 + i) * LSWP + c]);
#pragma unroll
          for (int j = 0; j < 4; j++) out[k][j] = make_float2(0.0f, 0.0f);
#pragma unroll
          for (int i = 0; i < FW0; i++) {
            const float ki = fa->taps0.k[i];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              out[k][j].x = fmaf(col[j + i].x, ki, out[k][j].x);  // ProgramCU.cu:226
              out[k][j].y = fmaf(col[j + i].y, ki, out[k][j].y);
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < KV0; k++) {
        const int t = tid + k * NT;
        if (t < NTV0) {
          const int rg = t / NCP0, c = 2 * (t - rg * NCP0);
          const int r0 = min(4 * rg, ROWS - 4);
#pragma unroll
          for (int j = 0; j < 4; j++) *reinterpret_cast<float2*>(&ws[(r0 + j) * LSWP + c]) = out[k][j];
        }
      }
      __syncthreads();
      if (y0 - R < 0 || y0 + TH + R > h) {  // block-uniform: window rows outside the image are copies of the edge rows
        const int r_lo = 0 - (y0 - R), r_hi = h - 1 - (y0 - R);  // window rows of image rows 0 and h-1
        for (int idx = tid; idx < ROWS * SW; idx += NT) {
          const int r = idx / SW, c = idx - r * SW;
          if (r < r_lo) ws[r * LSWP + c] = ws[r_lo * LSWP + c];
          else if (r > r_hi) ws[r * LSWP + c] = ws[r_hi * LSWP + c];
        }
        __syncthreads();
      }
      if (fa->dst0) {  // (block-uniform pointer) level 0 itself, only on request
        float* d0 = fa->dst0 + img * (long long)w * h;
        for (int idx = tid; idx < TH * (TW / 4); idx += NT) {
          const int ty = idx / (TW / 4), tx = (idx - ty * (TW / 4)) * 4;
          if (y0 + ty < h && x0 + tx < w)
            *reinterpret_cast<float4*>(&d0[(long long)(y0 + ty) * w + x0 + tx]) = *reinterpret_cast<const float4*>(&ws[(R + ty) * LSWP + R4 + tx]);
        }
      }
    }
  } else {
  // ---- stage 1: global -> LDS, replicate borders (ProgramCU.cu:138, :201) ----
  // All of a thread's loads are issued before its first LDS store, so their HBM latencies overlap.
  constexpr int NIT = (ROWS * NG + NT - 1) / NT;
  // Straight-line, branch-free loads (clamped row, clamped 16-byte-aligned column group) stored raw;
  // groups that lie left/right of the image are patched afterwards (border tiles only).
  float4 stage[NIT];
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int g = it * NT + tid;
    g = g < ROWS * NG ? g : ROWS * NG - 1;  // surplus threads repeat the last group
    const int r = g / NG, gx = g - r * NG;
    int y = y0 - R - RTOP + r;
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    const int x = x0 - R4 + gx * 4;
    const int xs = x < 0 ? 0 : (x >= w ? w - 4 : x);  // w is a multiple of 4
    if (U8) {
      const uint8_t* row = a.src_u8 + img * a.src_img_stride + (long long)y * a.src_pitch;
      const uchar4 b = *reinterpret_cast<const uchar4*>(row + xs);
      stage[it] = make_float4(dm_u8_unit((float)b.x), dm_u8_unit((float)b.y),   // p / 255.0f, GLTexImage.cpp:828
                              dm_u8_unit((float)b.z), dm_u8_unit((float)b.w));
    } else {
      const float* row = a.src + img * a.src_img_stride + (long long)y * a.src_pitch;
      stage[it] = *reinterpret_cast<const float4*>(row + xs);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int g = it * NT + tid;
    g = g < ROWS * NG ? g : ROWS * NG - 1;
    const int r = g / NG, gx = g - r * NG;
    *reinterpret_cast<float4*>(&s[r * SWP + gx * 4]) = stage[it];
  }
  if (x0 - R4 < 0 || x0 + TW + R4 > w) {  // block-uniform: replicate the edge pixel into outside groups
    __syncthreads();
    for (int g = tid; g < ROWS * NG; g += NT) {
      const int r = g / NG, gx = g - r * NG;
      const int x = x0 - R4 + gx * 4;
      if (x < 0 || x >= w) {
        // the group holds source columns 0..3 (left) or w-4..w-1 (right): take the edge one
        const float e = s[r * SWP + gx * 4 + (x < 0 ? 0 : 3)];
        *reinterpret_cast<float4*>(&s[r * SWP + gx * 4]) = make_float4(e, e, e, e);
      }
    }
  }
  __syncthreads();
  }

  // ---- TOP, tiles at the left / right image border: the produced level at the far side of the image, for the 1-D
  // neighbour addressing of det-H.  side 0 (left tiles): column w-1, rows y0-2 .. y0+31 (pixel (0, y) reads rows y-2,
  // y-1, y of it); side 1 (right tiles): column 0, rows y0 .. y0+33 (pixel (w-1, y) reads rows y, y+1, y+2).  Same
  // chains as the tile: v = fma(src(clamp(x - R + i)), k_i, v) along the row, then along the clamped rows. ----
  float* const wrapH = s + ROWS * SWP;      // [2][NWH]
  float* const wrapV = wrapH + 2 * NWH;     // [2][TH + 2]
  const bool wrap_left = TOP && x0 == 0, wrap_right = TOP && x0 + TW >= w;  // block-uniform
  if (TOP && (wrap_left || wrap_right)) {
    const int side = tid >> 7, t = tid & 127;
    const bool mine = side == 0 ? wrap_left : wrap_right;
    static_assert(NWH <= 128 && NT == 256, "one thread per filtered row and side");
    if (mine && t < NWH) {
      int yy = (side == 0 ? y0 - 2 : y0) - R + t;
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      const int c = side == 0 ? w - 1 : 0;
      const float* row = a.src + img * a.src_img_stride + (long long)yy * a.src_pitch;
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < FW; i++) {
        int xx = c - R + i;
        xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
        v = fmaf(row[xx], a.taps.k[i], v);  // ProgramCU.cu:152
      }
      wrapH[side * NWH + t] = v;
    }
    __syncthreads();
    if (mine && t < TH + 2) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < FW; i++) v = fmaf(wrapH[side * NWH + t + i], a.taps.k[i], v);  // ProgramCU.cu:226
      wrapV[side * (TH + 2) + t] = v;
    }
  }

  // ---- stage 1b (HESS): det-Hessian*sigma^4 and (gradient, theta) of the SOURCE level for this tile,
  // straight from the staged source window: the level is never re-read from HBM for it
  // (ComputeHessian_Kernel, ProgramCU.cu:523-595).  A thread does 4 adjacent pixels in each of two rows 16 apart:
  // its det-H store is 16 bytes next to its neighbours' (256 contiguous bytes per row and instruction) and its
  // gradient/theta stores are two 16-byte pieces at a 32-byte lane pitch.  (8 pixels of one row per thread meant
  // four 16-byte pieces at a 64-byte pitch, a shape that stores at half the rate: tools/micro/store_rate.hip.) ----
  if (HESS) {
    const int hx = (tid & 15) * 4;
    const int gx = x0 + hx;
    const bool want_got = a.got_src != nullptr;  // block-uniform
    const float* plane = a.src + img * a.src_img_stride;
    const int n = w * h;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int hr = (tid >> 4) + 16 * half;
      const int gy = y0 + hr;
      if (gy < h && gx < w) {
        float U[6], M[6], D[6];  // columns gx-1 .. gx+4 of rows gy-1, gy, gy+1
        {
          const float* base = &ws[(hr + R + RTOP - 1) * LSWP + hx + R4];
#pragma unroll
          for (int rr = 0; rr < 3; rr++) {
            float* dst = rr == 0 ? U : (rr == 1 ? M : D);
            const float4 v = *reinterpret_cast<const float4*>(base + rr * LSWP);
            dst[0] = base[rr * LSWP - 1];
            dst[1] = v.x; dst[2] = v.y; dst[3] = v.z; dst[4] = v.w;
            dst[5] = base[rr * LSWP + 4];
          }
        }
        // The staged window replicates the image border; the reference addresses neighbours by 1-D
        // index instead: rows outside the plane read 0, column -1 / w wraps to the adjacent row.
        const int idx = gy * w + gx;
        if (gy == 0) {
#pragma unroll
          for (int j = 0; j < 6; j++) U[j] = 0.0f;
        }
        if (gy == h - 1) {
#pragma unroll
          for (int j = 0; j < 6; j++) D[j] = 0.0f;
        }
        if (FIRST) {  // (the source level is not in HBM: its far-side columns were recomputed above)
          if (gx == 0) {
            U[0] = gy >= 2 ? wrap0V[hr] : 0.0f; M[0] = gy >= 1 ? wrap0V[hr + 1] : 0.0f; D[0] = wrap0V[hr + 2];
          }
          if (gx + 4 == w) {
            U[5] = wrap0V[TH + 2 + hr]; M[5] = gy + 1 <= h - 1 ? wrap0V[TH + 2 + hr + 1] : 0.0f; D[5] = gy + 2 <= h - 1 ? wrap0V[TH + 2 + hr + 2] : 0.0f;
          }
        } else {
          if (gx == 0) {
            U[0] = gtex1(plane, n, idx - w - 1); M[0] = gtex1(plane, n, idx - 1); D[0] = gtex1(plane, n, idx + w - 1);
          }
          if (gx + 4 == w) {  // this thread owns the row's last pixel (w is a multiple of 4)
            const int il = idx + 3;
            U[5] = gtex1(plane, n, il - w + 1); M[5] = gtex1(plane, n, il + 1); D[5] = gtex1(plane, n, il + w + 1);
          }
        }
        float hv[4];
        float2 gv[4];
        // two pixels at a time on 2-vectors, no per-pixel branches
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
#define HESS_V2(A, K) ((v2f){A[(K)], A[(K) + 1]})
          const v2f v11 = HESS_V2(U, j), v12 = HESS_V2(U, j + 1), v13 = HESS_V2(U, j + 2);
          const v2f v21 = HESS_V2(M, j), v22 = HESS_V2(M, j + 1), v23 = HESS_V2(M, j + 2);
          const v2f v31 = HESS_V2(D, j), v32 = HESS_V2(D, j + 1), v33 = HESS_V2(D, j + 2);
#undef HESS_V2
          const v2f Lxx = v2_fma(v2_splat(-2.0f), v22, v21) + v23;   // ProgramCU.cu:536
          const v2f Lyy = v2_fma(v2_splat(-2.0f), v22, v12) + v32;   // :537
          const v2f Lxy = (v13 - v11 + v31 - v33) * v2_splat(0.25f);  // :538
          const v2f dh = v2_fma(Lxx, Lyy, -(Lxy * Lxy)) * v2_splat(a.norm_src);  // :553
          hv[j] = dh.x; hv[j + 1] = dh.y;
          if (want_got) {
            const v2f dx = v23 - v21, dy = v32 - v12;                 // :556-557
            const v2f gradient = v2_splat(0.5f) * __builtin_elementwise_sqrt(v2_fma(dx, dx, dy * dy));
            const v2f th = dm_atan2f_x2(dy, dx);
            gv[j].x = gradient.x;     gv[j].y = (gradient.x == 0.0f) ? 0.0f : th.x;
            gv[j + 1].x = gradient.y; gv[j + 1].y = (gradient.y == 0.0f) ? 0.0f : th.y;
          }
        }
        const long long o = img * (long long)w * h + idx;
        store_stream_f4(a.deth_src + o, hv[0], hv[1], hv[2], hv[3]);
        if (want_got) {
          float* gp = reinterpret_cast<float*>(a.got_src + o);
          store_stream_f4(gp, gv[0].x, gv[0].y, gv[1].x, gv[1].y);
          store_stream_f4(gp + 4, gv[2].x, gv[2].y, gv[3].x, gv[3].y);
        }
      }
    }
  }

  // ---- stage 2: horizontal pass, LDS -> LDS in place ----
  // Every thread first computes its (at most two) 8-output tasks from the source window into registers;
  // after a barrier (all windows read) the results overwrite columns 0..63 of their row.  One LDS array
  // instead of two: 15-26 KB per workgroup, so registers (7 wavefronts per SIMD), not LDS (5), bound occupancy.
  {
    constexpr int NTASK = (ROWS * (TW / 8) + NT - 1) / NT;
    float acc[NTASK][8];
#pragma unroll
    for (int k = 0; k < NTASK; k++) {
      const int task = tid + k * NT;
      if (task < ROWS * (TW / 8)) {
        const int r = task >> 3, xb = (task & 7) * 8;
        float win[NV * 4];
        lds_read_groups<NV>(&ws[r * LSWP + xb], win);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[k][j] = 0.0f;
#pragma unroll
        for (int i = 0; i < FW; i++) {
          const float ki = a.taps.k[i];
#pragma unroll
          for (int j = 0; j < 8; j++) acc[k][j] = fmaf(win[OFF + j + i], ki, acc[k][j]);  // ProgramCU.cu:152
        }
      }
    }
    // TOP: the two halo columns of the output tile (image columns x0-1 and x0+64), one output per task
    float hacc = 0.0f;
    static_assert(2 * ROWS <= NT, "one halo task per thread");
    if (TOP && tid < 2 * ROWS) {
      const float* p = &ws[(tid >> 1) * LSWP + ((tid & 1) ? OFF + TW : OFF - 1)];
#pragma unroll
      for (int i = 0; i < FW; i++) hacc = fmaf(p[i], a.taps.k[i], hacc);  // ProgramCU.cu:152
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NTASK; k++) {
      const int task = tid + k * NT;
      if (task < ROWS * (TW / 8)) {
        const int r = task >> 3, xb = (task & 7) * 8;
        *reinterpret_cast<float4*>(&ws[r * LSWP + xb]) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        *reinterpret_cast<float4*>(&ws[r * LSWP + xb + 4]) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
      }
    }
    if (TOP && tid < 2 * ROWS) ws[(tid >> 1) * LSWP + TW + (tid & 1)] = hacc;  // columns 64 (left halo), 65 (right halo) of the row
  }
  __syncthreads();

  // ---- stage 3: vertical pass, LDS -> HBM ----
  if (!TOP) {
    const int cg = tid & 31, rg = tid >> 5;
    float2 col[4 + 2 * R];
#pragma unroll
    for (int i = 0; i < 4 + 2 * R; i++)
      col[i] = *reinterpret_cast<const float2*>(&ws[(rg * 4 + i) * LSWP + cg * 2]);
    float2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < FW; i++) {
      const float ki = a.taps.k[i];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[j].x = fmaf(col[j + i].x, ki, acc[j].x);  // ProgramCU.cu:226
        acc[j].y = fmaf(col[j + i].y, ki, acc[j].y);
      }
    }
    const int x = x0 + cg * 2;
    if (x < w) {
      float* d = a.dst + img * (long long)w * h;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int y = y0 + rg * 4 + j;
        if (y < h) *reinterpret_cast<float2*>(&d[(long long)y * w + x]) = acc[j];
      }
      if (a.decim_dst) {  // block-uniform.  x and y0 + 4*rg are even: rows j = 0, 2 and column .x are the sampled ones
        float* dd = a.decim_dst + img * (long long)a.dw * a.dh;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const int y = y0 + rg * 4 + j;
          if (y < h && (y >> 1) < a.dh) {
            float* row = dd + (long long)(y >> 1) * a.dw;
            if ((x >> 1) < a.dw) row[x >> 1] = acc[j].x;  // (the next octave may be narrower than w/2: widths halve unaligned)
            // columns of the next octave beyond w/2 (its width is aligned up to 4) repeat the source's last column
            if (x == w - 2) for (int xx = w >> 1; xx < a.dw; xx++) row[xx] = acc[j].y;
          }
        }
      }
    }
  }
  if (TOP) {
    // ---- stage 3 (TOP): vertical pass over the tile and its halo, LDS -> registers -> LDS.  Column pairs 0..31 are the
    // tile's, pair 32 = the two halo columns (row entries 64, 65); row groups of four start at tile rows -1, 3, .. 27
    // and 29 (the last one recomputes rows 29, 30: same values). ----
    constexpr int NCP = TW / 2 + 1, NRG = TH / 4 + 1, NTV = NCP * NRG, KV = (NTV + NT - 1) / NT;
    constexpr int TS = TW + 8;  // row pitch of the output tile in LDS: image column x0 - 4 + c, tile row -1 + r
    static_assert((TH + 2) * TS <= ROWS * SWP, "the output tile reuses the staged rows' LDS");
    float2 vout[KV][4];
    int vr0[KV], vcp[KV];
#pragma unroll
    for (int k = 0; k < KV; k++) {
      const int t = tid + k * NT;
      const int rgi = t / NCP;
      vcp[k] = t - rgi * NCP;
      vr0[k] = t < NTV ? (rgi < TH / 4 ? 4 * rgi - 1 : TH - 3) : -100;
      if (t < NTV) {
        float2 col[4 + 2 * R];
#pragma unroll
        for (int i = 0; i < 4 + 2 * R; i++)
          col[i] = *reinterpret_cast<const float2*>(&s[(vr0[k] + 1 + i) * SWP + vcp[k] * 2]);
#pragma unroll
        for (int j = 0; j < 4; j++) vout[k][j] = make_float2(0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < FW; i++) {
          const float ki = a.taps.k[i];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            vout[k][j].x = fmaf(col[j + i].x, ki, vout[k][j].x);  // ProgramCU.cu:226
            vout[k][j].y = fmaf(col[j + i].y, ki, vout[k][j].y);
          }
        }
      }
    }
    __syncthreads();  // every column of horizontal results has been read: the output tile takes their place
#pragma unroll
    for (int k = 0; k < KV; k++) {
      if (vr0[k] > -100) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float* trow = &s[(vr0[k] + 1 + j) * TS];
          if (vcp[k] < TW / 2) *reinterpret_cast<float2*>(&trow[4 + 2 * vcp[k]]) = vout[k][j];
          else { trow[3] = vout[k][j].x; trow[4 + TW] = vout[k][j].y; }
        }
        if (a.dst && vcp[k] < TW / 2) {  // (block-uniform pointer) the level itself, only on request
          const int x = x0 + 2 * vcp[k];
          float* d = a.dst + img * (long long)w * h;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int ty = vr0[k] + j, y = y0 + ty;
            if (ty >= 0 && ty < TH && y < h && x < w) *reinterpret_cast<float2*>(&d[(long long)y * w + x]) = vout[k][j];
          }
        }
      }
    }
    __syncthreads();
    // ---- stage 4 (TOP): det-Hessian*sigma^4 of the produced level from the output tile (ComputeHessian_Kernel,
    // ProgramCU.cu:523-553; thread -> pixels as stage 1b) ----
    {
      const int hx = (tid & 15) * 4;
      const int gx = x0 + hx;
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int hr = (tid >> 4) + 16 * half;
        const int gy = y0 + hr;
        if (gy < h && gx < w) {
          float U[6], M[6], D[6];  // columns gx-1 .. gx+4 of rows gy-1, gy, gy+1
          {
            const float* base = &s[hr * TS + hx + 4];  // tile row hr-1, image column gx
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
              float* dst = rr == 0 ? U : (rr == 1 ? M : D);
              const float4 v = *reinterpret_cast<const float4*>(base + rr * TS);
              dst[0] = base[rr * TS - 1];
              dst[1] = v.x; dst[2] = v.y; dst[3] = v.z; dst[4] = v.w;
              dst[5] = base[rr * TS + 4];
            }
          }
          // 1-D neighbour addressing: rows outside the plane read 0, column -1 / w is the adjacent row's far end
          if (gy == 0) {
#pragma unroll
            for (int j = 0; j < 6; j++) U[j] = 0.0f;
          }
          if (gy == h - 1) {
#pragma unroll
            for (int j = 0; j < 6; j++) D[j] = 0.0f;
          }
          if (gx == 0) {  // index - w - 1, index - 1, index + w - 1: column w-1 of rows gy-2, gy-1, gy
            U[0] = gy >= 2 ? wrapV[hr] : 0.0f; M[0] = gy >= 1 ? wrapV[hr + 1] : 0.0f; D[0] = wrapV[hr + 2];
          }
          if (gx + 4 == w) {  // index - w + 1, index + 1, index + w + 1 of the row's last pixel: column 0 of rows gy, gy+1, gy+2
            U[5] = wrapV[TH + 2 + hr]; M[5] = gy + 1 <= h - 1 ? wrapV[TH + 2 + hr + 1] : 0.0f; D[5] = gy + 2 <= h - 1 ? wrapV[TH + 2 + hr + 2] : 0.0f;
          }
          float hv[4];
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
#define HESS_V2(A, K) ((v2f){A[(K)], A[(K) + 1]})
            const v2f v11 = HESS_V2(U, j), v12 = HESS_V2(U, j + 1), v13 = HESS_V2(U, j + 2);
            const v2f v21 = HESS_V2(M, j), v22 = HESS_V2(M, j + 1), v23 = HESS_V2(M, j + 2);
            const v2f v31 = HESS_V2(D, j), v32 = HESS_V2(D, j + 1), v33 = HESS_V2(D, j + 2);
#undef HESS_V2
            const v2f Lxx = v2_fma(v2_splat(-2.0f), v22, v21) + v23;   // ProgramCU.cu:536
            const v2f Lyy = v2_fma(v2_splat(-2.0f), v22, v12) + v32;   // :537
            const v2f Lxy = (v13 - v11 + v31 - v33) * v2_splat(0.25f);  // :538
            const v2f dh = v2_fma(Lxx, Lyy, -(Lxy * Lxy)) * v2_splat(a.norm_dst);  // :553
            hv[j] = dh.x; hv[j + 1] = dh.y;
          }
          store_stream_f4(a.deth_dst + img * (long long)w * h + (long long)gy * w + gx, hv[0], hv[1], hv[2], hv[3]);
        }
      }
    }
  }
}

template <int R, bool U8, bool HESS>
__global__ __launch_bounds__(NT) void gauss_kernel(GaussArgs a) {
  __shared__ __attribute__((aligned(16))) float s[gauss_tile_lds<R>()];
  gauss_tile<R, U8, HESS>(a, s, (int)blockIdx.x);
}

// Octave 0 of a u8 image: level 0 in LDS only, level 1 + det-H of level 0 out (gauss_tile, FIRST).
template <int R0, int R>
__global__ __launch_bounds__(NT) void gauss_first_kernel(GaussArgs a, FirstArgs fa) {
  __shared__ __attribute__((aligned(16))) float s[gauss_tile_lds<R, false, R0>()];
  gauss_tile<R, true, true, false, R0>(a, s, (int)blockIdx.x, 0, &fa);
}

// The octave's top level: det-H of the produced level from the output tile, the level itself not stored (gauss_tile, TOP).
template <int R>
__global__ __launch_bounds__(NT) void gauss_top_kernel(GaussArgs a) {
  __shared__ __attribute__((aligned(16))) float s[gauss_tile_lds<R, true>()];
  gauss_tile<R, false, true, true>(a, s, (int)blockIdx.x, (int)gridDim.x);
}

// Two level launches that do not depend on each other in one grid: the top level of octave o (taps RA, its source level
// has a gradient plane) and level 1 of octave o+1 (taps RB) -- T(o, l) = 3o + l is the earliest step of level l of
// octave o, so level 4 of one octave and level 1 of the next are due together.  The small half hides behind the large
// one: six launches fewer in the dependent chain of a 1080p pyramid.  The first blocks_a workgroups do half a.
// Half a is a TOP tile (gauss_tile): det-H of the top level from its output tile, the level itself not stored.
template <int RA, int RB>
__global__ __launch_bounds__(NT) void gauss_pair_kernel(GaussArgs a, GaussArgs b, int blocks_a) {
  constexpr int LDS = gauss_tile_lds<RA, true>() > gauss_tile_lds<RB>() ? gauss_tile_lds<RA, true>() : gauss_tile_lds<RB>();
  __shared__ __attribute__((aligned(16))) float s[LDS];
  if ((int)blockIdx.x < blocks_a) gauss_tile<RA, false, true, true>(a, s, (int)blockIdx.x, blocks_a);  // (workgroup-uniform)
  else gauss_tile<RB, false, true>(b, s, (int)blockIdx.x - blocks_a);
}

// ------------------------------------------------------------------------------------------------
// Level chain: levels 1 .. level_ds of one octave in ONE launch -- the levels the NEXT octave waits for.
//
// The pyramid's dependency graph is a chain: level l needs level l-1, and level 0 of octave o+1 is the decimated level
// level_ds (3) of octave o.  Below 960x540 a level launch is a few dozen workgroups that spend their 4 - 6 us mostly on
// latency, and the fifteen launches of octaves 2 - 6 of a 1080p image took 100 of the 190 us of its pyramid (42 % of the
// Gaussian stage's time for 21 % of its bytes at batches of 8).  What the next octave needs of this one is only level
// level_ds; the top level, the det-Hessian and the gradient planes are nobody's input inside the pyramid.  So:
//   gauss_chain_kernel  one workgroup produces a 32x32 tile of levels 1, 2 and 3 from level 0, in LDS, trading the
//                       dependent launches for redundant arithmetic on a shrinking halo (level l is computed on the
//                       tile grown by the radii of the levels still to come: 14, 8, 0 pixels), and stores level 3's
//                       even rows / columns as level 0 of the next octave (DownsampleKernel, ProgramCU.cu:312-326): one
//                       launch per octave on the critical path;
//   off the critical path, after the last octave: level 4 of all octaves in one launch (gauss_multi_kernel below: the
//                       tile kernel, which also emits det-H / gradient of its source level 3) and det-H / gradient of
//                       levels 0 - 2 of the chained octaves from HBM (hessian_rows in k_detect.hip, same launch as the
//                       top levels' det-H).
// Inside the workgroup, per level: horizontal pass in place (4 outputs per task from a register window of ds_read_b128,
// all windows read before the first result is written), vertical pass into the other LDS region (4 rows x 2 columns per
// task, ds_read_b64), then the tile goes to HBM.  Borders as the reference clamps its fetch index (ProgramCU.cu:138,201):
// columns outside the image hold the value of the clamped column (stage 0 / a fix-up after each level, border tiles
// only), rows outside the image are never computed -- the vertical pass of a border tile clamps its row index instead.
// Every value is the tap chain `v = 0; v = fma(x_i, k_i, v)` of the tile kernel on the same inputs, so the planes are
// bit-identical to the level-by-level launches whatever the tiling.  Instantiated for the radii of the reference's
// default schedule (11, 13, 17 taps for levels 1 - 3); other schedules keep the level-by-level launches.
constexpr int CT = 32;     // tile side
constexpr int CNT = 1024;  // threads per workgroup
constexpr int CNL = 3;     // levels per chain launch (= level_ds of the default schedule)

struct ChainArgs {
  const float* src0;     // level 0 of the octave, [batch][h][w]
  float* dst[CNL + 1];   // levels 1..3 ([0] unused)
  int w, h, tiles_x, tiles_y, batch;
  float* decim_dst;      // level 0 of the next octave, [batch][dh][dw] (null: last octave)
  int dw, dh;
  Taps taps[CNL + 1];    // taps[l] produces level l from level l-1
};

template <int R1, int R2, int R3>
struct ChainShape {
  static constexpr int R(int l) { return l == 1 ? R1 : l == 2 ? R2 : R3; }
  static constexpr int R4up(int l) { return (R(l) + 3) & ~3; }
  static constexpr int H(int l) { return l >= CNL ? 0 : H(l + 1) + R(l + 1); }  // halo of level l around the tile
  static constexpr int max2(int a, int b) { return a > b ? a : b; }
  // column origin of the tile inside the LDS rows: every 16-byte aligned register window of the horizontal passes
  // starts at a column >= 0
  static constexpr int orgx() {
    int m = H(0);
    for (int l = 1; l <= CNL; l++) m = max2(m, H(l) + 3 + R4up(l));
    return (m + 3) & ~3;
  }
  static constexpr int ORGX = orgx();
  static constexpr int ORGY = H(0);
  static constexpr int ncol() {
    int m = ORGX + CT + H(0);
    for (int l = 1; l <= CNL; l++) m = max2(m, ((ORGX + CT + H(l) - 1) & ~3) + 4 + R4up(l));
    return (m + 3) & ~3;
  }
  static constexpr int NCOL = ncol();
  static constexpr int STRIDE = NCOL + 4;
  static constexpr int NROW = CT + 2 * H(0);
  static constexpr int REGION = NROW * STRIDE;  // floats per LDS region; two regions
};

template <int R1, int R2, int R3>
__global__ __launch_bounds__(CNT) void gauss_chain_kernel(ChainArgs a) {
  using S = ChainShape<R1, R2, R3>;
  constexpr int ORGX = S::ORGX, ORGY = S::ORGY, STRIDE = S::STRIDE;
  __shared__ __attribute__((aligned(16))) float lds[2 * S::REGION];
  float* const X = lds;
  float* const Y = lds + S::REGION;
  const int tid = threadIdx.x;
  const int w = a.w, h = a.h;
  const int per_img = a.tiles_x * a.tiles_y;
  const int img = (int)blockIdx.x / per_img;
  const int trem = (int)blockIdx.x - img * per_img;
  const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
  const int x0 = txi * CT, y0 = tyi * CT;
  const long long ioff = (long long)img * w * h;
  const int tx = tid & 31, ty = tid >> 5;  // this thread's pixel of the tile
  const int gx = x0 + tx, gy = y0 + ty;
  const bool px_in = gx < w && gy < h;
  // LDS row / column of image row 0 / column 0, of the last image row / column (may lie outside the arrays)
  const int row_lo = ORGY - y0, row_hi = ORGY + (h - 1 - y0);
  const int col_lo = ORGX - x0, col_hi = ORGX + (w - 1 - x0);

  // ---- stage 0: level 0 window -> X (rows inside the image only), columns outside the image replicated ----
  {
    constexpr int NG = S::NCOL / 4, ROWS = S::NROW;
    constexpr int NIT = (ROWS * NG + CNT - 1) / CNT;
    const float* plane = a.src0 + ioff;
    float4 v[NIT];
    int at[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int g = it * CNT + tid;
      const int r = g / NG, gq = g - r * NG;
      const int y = y0 - ORGY + r;
      const int x = x0 - ORGX + gq * 4;
      const bool ok = g < ROWS * NG && y >= 0 && y < h;
      const int xs = x < 0 ? 0 : (x >= w ? w - 4 : x);  // w is a multiple of 4
      float4 q = *reinterpret_cast<const float4*>(plane + (long long)(ok ? y : 0) * w + xs);
      if (x < 0) q = make_float4(q.x, q.x, q.x, q.x);
      if (x >= w) q = make_float4(q.w, q.w, q.w, q.w);
      v[it] = q;
      at[it] = ok ? r * STRIDE + gq * 4 : -1;
    }
#pragma unroll
    for (int it = 0; it < NIT; it++)
      if (at[it] >= 0) *reinterpret_cast<float4*>(&X[at[it]]) = v[it];
  }
  __syncthreads();

  auto level = [&](auto ltag, float* Sg, float* Dg) {
    constexpr int L = decltype(ltag)::value;
    constexpr int R = S::R(L), FW = 2 * R + 1, RU = S::R4up(L);
    constexpr int HP = S::H(L - 1), HL = S::H(L);  // halo of the source level, of this level
    // ---- horizontal pass, in place: rows of the source window (inside the image) x aligned 4-column runs that
    // cover this level's columns (inside the image) ----
    constexpr int RB = ORGY - HP, NRH = CT + 2 * HP;
    constexpr int CB = (ORGX - HL) & ~3, CE = ORGX + CT + HL, NRUN = (CE - CB + 3) / 4;
    constexpr int NTASK = NRH * NRUN, KH = (NTASK + CNT - 1) / CNT;
    constexpr int NV = (4 + 2 * RU) / 4, OFF = RU - R;
    static_assert(CB - RU >= 0 && CB + 4 * NRUN + RU <= S::NCOL, "register windows stay inside the LDS rows");
    float acc[KH][4];
    bool live[KH];
#pragma unroll
    for (int k = 0; k < KH; k++) {
      const int t = tid + k * CNT;
      const int r = RB + t / NRUN, c0 = CB + 4 * (t % NRUN);
      live[k] = t < NTASK && r >= row_lo && r <= row_hi && c0 + 3 >= col_lo && c0 <= col_hi;
      if (live[k]) {
        float win[NV * 4];
        lds_read_groups<NV>(&Sg[r * STRIDE + c0 - RU], win);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[k][j] = 0.0f;
#pragma unroll
        for (int i = 0; i < FW; i++) {
          const float ki = a.taps[L].k[i];
#pragma unroll
          for (int j = 0; j < 4; j++) acc[k][j] = fmaf(win[OFF + j + i], ki, acc[k][j]);  // ProgramCU.cu:152
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KH; k++) {
      if (live[k]) {
        const int t = tid + k * CNT;
        const int r = RB + t / NRUN, c0 = CB + 4 * (t % NRUN);
        *reinterpret_cast<float4*>(&Sg[r * STRIDE + c0]) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
      }
    }
    __syncthreads();
    // ---- vertical pass into the other region: 4 rows x 2 columns per task; the last run of rows is moved up onto
    // the window's end (it recomputes up to three rows of the run before it: same values).  A tile whose window
    // reaches beyond the first or last image row clamps the row index of its reads (the reference's clamped fetch). ----
    constexpr int RB2 = ORGY - HL, NRV = CT + 2 * HL, NRR = (NRV + 3) / 4, NPAIR = 2 * NRUN;
    constexpr int NTV = NRR * NPAIR, KV = (NTV + CNT - 1) / CNT;
    auto vpass = [&](auto clamp_tag) {
      constexpr bool CLAMP = decltype(clamp_tag)::value;
#pragma unroll
      for (int k = 0; k < KV; k++) {
        const int t = tid + k * CNT;
        const int rr = t / NPAIR, cp = t - rr * NPAIR;
        const int r0 = RB2 + min(4 * rr, NRV - 4), c = CB + 2 * cp;
        if (t < NTV && r0 + 3 >= row_lo && r0 <= row_hi && c + 1 >= col_lo && c <= col_hi) {
          float2 col[4 + 2 * R];
#pragma unroll
          for (int i = 0; i < 4 + 2 * R; i++) {
            int r = r0 - R + i;
            if (CLAMP) r = min(max(r, row_lo), row_hi);
            col[i] = *reinterpret_cast<const float2*>(&Sg[r * STRIDE + c]);
          }
          float2 out[4];
#pragma unroll
          for (int j = 0; j < 4; j++) out[j] = make_float2(0.0f, 0.0f);
#pragma unroll
          for (int i = 0; i < FW; i++) {
            const float ki = a.taps[L].k[i];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              out[j].x = fmaf(col[j + i].x, ki, out[j].x);  // ProgramCU.cu:226
              out[j].y = fmaf(col[j + i].y, ki, out[j].y);
            }
          }
#pragma unroll
          for (int j = 0; j < 4; j++) *reinterpret_cast<float2*>(&Dg[(r0 + j) * STRIDE + c]) = out[j];
        }
      }
    };
    if (y0 - HP < 0 || y0 + CT + HP > h) vpass(std::true_type{});  // (workgroup-uniform)
    else vpass(std::false_type{});
    __syncthreads();
    // ---- columns of this level's window outside the image: the value of the clamped column (rows inside the image) ----
    if (HL > 0 && (x0 - HL < 0 || x0 + CT + HL > w)) {  // (workgroup-uniform)
      constexpr int NCW = CT + 2 * HL;
      for (int idx = tid; idx < NRV * NCW; idx += CNT) {
        const int rr = idx / NCW, cc = idx - rr * NCW;
        const int r = RB2 + rr, c = ORGX - HL + cc;
        if (r >= row_lo && r <= row_hi && (c < col_lo || c > col_hi)) Dg[r * STRIDE + c] = Dg[r * STRIDE + (c < col_lo ? col_lo : col_hi)];
      }
      __syncthreads();
    }
    // ---- the tile of this level -> HBM ----
    if (px_in) {
      const float v = Dg[(ORGY + ty) * STRIDE + ORGX + tx];
      a.dst[L][ioff + (long long)gy * w + gx] = v;
      if (L == CNL && a.decim_dst && !(gy & 1) && (gy >> 1) < a.dh) {  // dst(x, y) = src(min(2x, w-1), 2y)
        float* row = a.decim_dst + (long long)img * a.dw * a.dh + (long long)(gy >> 1) * a.dw;
        if (!(gx & 1) && (gx >> 1) < a.dw) row[gx >> 1] = v;
        if (gx == w - 1) for (int xx = w >> 1; xx < a.dw; xx++) row[xx] = v;  // columns beyond w/2 repeat the last one
      }
    }
  };
  level(std::integral_constant<int, 1>{}, X, Y);
  level(std::integral_constant<int, 2>{}, Y, X);
  level(std::integral_constant<int, 3>{}, X, Y);
}

// ---- several independent level launches of the same tap count in one grid (the top levels of all octaves: their
// sources, level 3 of every octave, are complete once the chain above has run): the tile kernel's body per job ----
// The workgroups after the last job's do det-H / gradient of the chained octaves' low levels (hessian_low_levels):
// they, too, only need planes the chain has completed.
constexpr int kMultiJobs = 8;
struct MultiArgs {
  GaussArgs j[kMultiJobs];
  int first_block[kMultiJobs + 1];  // job k owns workgroups [first_block[k], first_block[k+1])
  int njobs, batch;
  // low levels: blocks [first_block[njobs], +low_blocks * batch), image-major
  int low_first, low_nlv, low_blocks;
  const float* gauss;
  float* deth;
  float2* got;
  LevelNorms nm;
};
struct MultiGeom { Geom g; };
template <int R>
__global__ __launch_bounds__(NT) void gauss_multi_kernel(MultiArgs m, MultiGeom mg) {
  __shared__ __attribute__((aligned(16))) float s[gauss_tile_lds<R, true>()];
  const int blk = (int)blockIdx.x;
  if (blk >= m.first_block[m.njobs]) {  // (workgroup-uniform)
    const int lb = blk - m.first_block[m.njobs];
    const int b = lb / m.low_blocks;
    hessian_low_levels(mg.g, m.gauss, m.deth, m.got, m.nm, m.low_first, m.low_nlv, lb - b * m.low_blocks, b);
    return;
  }
  int k = 0;
  for (int q = 1; q < m.njobs; q++) if (blk >= m.first_block[q]) k = q;  // (uniform scalar walk)
  gauss_tile<R, false, true, true>(m.j[k], s, blk - m.first_block[k], m.first_block[k + 1] - m.first_block[k]);  // top levels: TOP tiles
}

template <int R>
void launch_r(hipStream_t st, GaussArgs a, int batch) {
  a.tiles_x = (a.w + TW - 1) / TW;
  a.tiles_y = (a.h + TH - 1) / TH;
  a.batch = batch;
  const int ntile = a.tiles_x * a.tiles_y * batch;
  dim3 grid(((ntile + 7) / 8) * 8);
  if (a.deth_dst)
    hipLaunchKernelGGL((gauss_top_kernel<R>), grid, dim3(NT), 0, st, a);
  else if (a.src_u8)
    hipLaunchKernelGGL((gauss_kernel<R, true, false>), grid, dim3(NT), 0, st, a);
  else if (a.deth_src)
    hipLaunchKernelGGL((gauss_kernel<R, false, true>), grid, dim3(NT), 0, st, a);
  else
    hipLaunchKernelGGL((gauss_kernel<R, false, false>), grid, dim3(NT), 0, st, a);
}

// ---- input conversion (GLTexImage.cpp:802-916): any format/type -> float luminance ----
struct ConvArgs {
  const uint8_t* src;
  long long pitch, img_stride;  // bytes
  int format, pixtype, ds;
  float* dst;
  int w, h;
};

__device__ __forceinline__ float conv_pixel(const uint8_t* p, int format, int pixtype) {
  const bool lum = (format == 1 || format == 2);
  if (pixtype == 3) {
    const float* f = reinterpret_cast<const float*>(p);
    if (lum) return f[0];
    // host arithmetic in the reference: separate multiplies and adds, left to right
    if (format == 3 || format == 4) return __fadd_rn(__fadd_rn(__fmul_rn(0.299f, f[0]), __fmul_rn(0.587f, f[1])), __fmul_rn(0.114f, f[2]));
    return __fadd_rn(__fadd_rn(__fmul_rn(0.114f, f[0]), __fmul_rn(0.587f, f[1])), __fmul_rn(0.299f, f[2]));
  }
  unsigned v0, v1 = 0, v2 = 0;
  float factor;
  if (pixtype == 1) {
    v0 = p[0]; if (!lum) { v1 = p[1]; v2 = p[2]; }
    factor = 255.0f;
  } else {
    const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
    v0 = q[0]; if (!lum) { v1 = q[1]; v2 = q[2]; }
    factor = 65535.0f;
  }
  if (lum) return (float)(int)v0 / factor;
  if (format == 3 || format == 4) return (float)(int32_t)(19595u * v0 + 38470u * v1 + 7471u * v2) / (65535.0f * factor);
  return (float)(int32_t)(7471u * v0 + 38470u * v1 + 19595u * v2) / (65535.0f * factor);
}

__global__ __launch_bounds__(256) void convert_kernel(ConvArgs a) {
  int x = blockIdx.x * 256 + threadIdx.x;
  int y = blockIdx.y;
  if (x >= a.w) return;
  int nch = (a.format == 1) ? 1 : (a.format == 2 ? 2 : ((a.format == 3 || a.format == 5) ? 3 : 4));
  int bpc = a.pixtype == 1 ? 1 : (a.pixtype == 2 ? 2 : 4);
  int step = 1 << a.ds;
  const uint8_t* p = a.src + (long long)blockIdx.z * a.img_stride + (long long)(y * step) * a.pitch +
                     (long long)(x * step) * nch * bpc;
  a.dst[((long long)blockIdx.z * a.h + y) * a.w + x] = conv_pixel(p, a.format, a.pixtype);
}

// ---- UpsampleKernel<LOG_SCALE>, ProgramCU.cu:233-285: linear interpolation by 2^k; the source is addressed
// by 1-D index (index+1 at a row end is the next row's first pixel, past the plane reads 0).  One thread per
// (destination row, source column) writes 2^k adjacent outputs. ----
__global__ __launch_bounds__(256) void upsample_kernel(const float* src, int width, int height, int log_scale,
                                                       float* dst) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= width) return;
  const int SCALE = 1 << log_scale;
  const float INV_SCALE = 1.0f / (float)SCALE;
  const int dst_row = blockIdx.y;
  const int row = dst_row >> log_scale, helper = dst_row & (SCALE - 1);
  const int n = width * height;
  const float* plane = src + (long long)blockIdx.z * n;
  const int index = row * width + col;
  float v1, v2;
  if (helper) {
    const float v11 = gtex1(plane, n, index), v12 = gtex1(plane, n, index + 1);
    const float v21 = gtex1(plane, n, index + width), v22 = gtex1(plane, n, index + width + 1);
    const float w1 = INV_SCALE * helper, w2 = (float)(1.0 - w1);
    v1 = fmaf(v21, w1, w2 * v11);  // :257
    v2 = fmaf(v22, w1, w2 * v12);  // :258
  } else {
    v1 = gtex1(plane, n, index);
    v2 = gtex1(plane, n, index + 1);
  }
  float* d = dst + ((long long)blockIdx.z * (height << log_scale) + dst_row) * ((long long)width << log_scale) +
             ((long long)col << log_scale);
  d[0] = v1;
  for (int i = 1; i < SCALE; ++i) {
    const float r2 = i * INV_SCALE;
    const float r1 = 1.0f - r2;
    d[i] = fmaf(v1, r1, v2 * r2);  // :267
  }
}

// ---- DownsampleKernel<1>, ProgramCU.cu:312-326: dst(x,y) = src(min(2x, sw-1), 2y) ----
__global__ __launch_bounds__(256) void downsample_kernel(const float* src, int sw, int splane, float* dst,
                                                         int dw, int dh) {
  int x = blockIdx.x * 256 + threadIdx.x;
  int y = blockIdx.y;
  if (x >= dw) return;
  int sc = min(x << 1, sw - 1);
  dst[((long long)blockIdx.z * dh + y) * dw + x] = src[(long long)blockIdx.z * splane + (long long)(y << 1) * sw + sc];
}

}  // namespace

namespace { void launch_by_radius(hipStream_t st, const GaussArgs& a, int batch); }

void launch_gauss(hipStream_t st, const float* src, const uint8_t* src_u8, long long src_pitch,
                  long long src_img_stride, float* dst, int wa, int h, int batch, const Taps& taps,
                  float* deth_src, float* got_src, float norm_src, float* decim_dst, int decim_w, int decim_h) {
  GaussArgs a;
  a.decim_dst = decim_dst; a.dw = decim_w; a.dh = decim_h;
  a.src = src; a.src_u8 = src_u8; a.src_pitch = src_pitch; a.src_img_stride = src_img_stride;
  a.dst = dst; a.w = wa; a.h = h; a.taps = taps;
  a.deth_src = deth_src; a.got_src = reinterpret_cast<float2*>(got_src); a.norm_src = norm_src;
  a.deth_dst = nullptr; a.norm_dst = 0.0f; a.zero = nullptr; a.zero_n16 = 0;
  launch_by_radius(st, a, batch);
}

namespace {
void launch_by_radius(hipStream_t st, const GaussArgs& a, int batch) {
  switch (a.taps.fw >> 1) {
    case 2: launch_r<2>(st, a, batch); break;
    case 3: launch_r<3>(st, a, batch); break;
    case 4: launch_r<4>(st, a, batch); break;
    case 5: launch_r<5>(st, a, batch); break;
    case 6: launch_r<6>(st, a, batch); break;
    case 7: launch_r<7>(st, a, batch); break;
    case 8: launch_r<8>(st, a, batch); break;
    case 9: launch_r<9>(st, a, batch); break;
    case 10: launch_r<10>(st, a, batch); break;
    case 11: launch_r<11>(st, a, batch); break;
    case 12: launch_r<12>(st, a, batch); break;
    case 13: launch_r<13>(st, a, batch); break;
    case 14: launch_r<14>(st, a, batch); break;
    case 15: launch_r<15>(st, a, batch); break;
    case 16: launch_r<16>(st, a, batch); break;
    default: break;
  }
}
}  // namespace

namespace {
GaussArgs job_args(const GaussJob& j, int batch) {
  GaussArgs a;
  a.decim_dst = j.decim_dst; a.dw = j.decim_w; a.dh = j.decim_h;
  a.src = j.src; a.src_u8 = nullptr; a.src_pitch = j.wa; a.src_img_stride = (long long)j.wa * j.h;
  a.dst = j.dst; a.w = j.wa; a.h = j.h; a.taps = j.taps;
  a.deth_src = j.deth_src; a.got_src = reinterpret_cast<float2*>(j.got_src); a.norm_src = j.norm_src;
  a.deth_dst = j.deth_dst; a.norm_dst = j.norm_dst;
  a.zero = reinterpret_cast<uint4*>(j.zero); a.zero_n16 = (long long)(j.zero_bytes / 16);
  a.tiles_x = (j.wa + TW - 1) / TW; a.tiles_y = (j.h + TH - 1) / TH; a.batch = batch;
  return a;
}
template <int RA, int RB>
void launch_pair(hipStream_t st, const GaussArgs& a, const GaussArgs& b) {
  const int na = ((a.tiles_x * a.tiles_y * a.batch + 7) / 8) * 8, nb = ((b.tiles_x * b.tiles_y * b.batch + 7) / 8) * 8;
  hipLaunchKernelGGL((gauss_pair_kernel<RA, RB>), dim3(na + nb), dim3(NT), 0, st, a, b, na);
}
template <int RA>
bool launch_pair_b(hipStream_t st, const GaussArgs& a, const GaussArgs& b, int rb) {
  switch (rb) {
    case 4: launch_pair<RA, 4>(st, a, b); return true;
    case 5: launch_pair<RA, 5>(st, a, b); return true;
    case 6: launch_pair<RA, 6>(st, a, b); return true;
    default: return false;
  }
}
}  // namespace

// Octave 0 of u8 pixels: level 0 (taps0, in LDS only unless dst0 is given) and level 1 (the job: its dst, taps, det-H
// plane of its source = level 0) in one launch.  Instantiated for the reference's default schedule (13 and 11 taps);
// false: not this pair of tap counts -- launch level 0 and level 1 one after the other.
bool gauss_first_available(const Taps& taps0, const Taps& taps1) { return (taps0.fw >> 1) == 6 && (taps1.fw >> 1) == 5; }

bool launch_gauss_first(hipStream_t st, const uint8_t* pixels, long long pitch, long long img_stride, const Taps& taps0,
                        const GaussJob& level1, float* dst0, int batch) {
  if (!gauss_first_available(taps0, level1.taps) || !level1.deth_src || level1.got_src || level1.deth_dst ||
      level1.decim_dst || (pitch % 4) != 0 || (img_stride % 4) != 0)
    return false;
  GaussArgs a = job_args(level1, batch);
  a.src = nullptr; a.src_u8 = pixels; a.src_pitch = pitch; a.src_img_stride = img_stride;
  FirstArgs fa;
  fa.taps0 = taps0; fa.dst0 = dst0;
  const int ntile = a.tiles_x * a.tiles_y * batch;
  hipLaunchKernelGGL((gauss_first_kernel<6, 5>), dim3(((ntile + 7) / 8) * 8), dim3(NT), 0, st, a, fa);
  return true;
}

// One level launch described by a job (any level; a job with deth_dst is a top level: TOP tiles).
void launch_gauss_job(hipStream_t st, const GaussJob& j, int batch) {
  launch_by_radius(st, job_args(j, batch), batch);
}

// Level launches a (the larger: top level of an octave) and b (level 1 of the next octave) in one grid.  Instantiated
// for the tap counts around the reference's default schedule (a: 17-25 taps, b: 9-13); false = not this pair, launch
// them one after the other.
bool launch_gauss_pair(hipStream_t st, const GaussJob& ja, const GaussJob& jb, int batch) {
  if (!ja.deth_src || !jb.deth_src || !ja.deth_dst || jb.deth_dst) return false;  // a: a top level (TOP tile), b: not
  const GaussArgs a = job_args(ja, batch), b = job_args(jb, batch);
  const int rb = jb.taps.fw >> 1;
  switch (ja.taps.fw >> 1) {
    case 8: return launch_pair_b<8>(st, a, b, rb);
    case 9: return launch_pair_b<9>(st, a, b, rb);
    case 10: return launch_pair_b<10>(st, a, b, rb);
    case 11: return launch_pair_b<11>(st, a, b, rb);
    case 12: return launch_pair_b<12>(st, a, b, rb);
    default: return false;
  }
}

// Levels 1..3 of one octave in one launch (+ level 0 of the next octave); false: not this schedule (launch the levels
// one by one).
bool gauss_chain_available(const Taps* taps, int level_ds) {
  return level_ds == CNL && taps[1].fw == 11 && taps[2].fw == 13 && taps[3].fw == 17;
}

bool launch_gauss_chain(hipStream_t st, const ChainJob& j, int batch) {
  if (!gauss_chain_available(j.taps, j.nlevels)) return false;
  ChainArgs a;
  a.src0 = j.src0;
  for (int l = 0; l <= CNL; l++) { a.dst[l] = l ? j.dst[l] : nullptr; a.taps[l] = j.taps[l]; }
  a.w = j.wa; a.h = j.h; a.tiles_x = (j.wa + CT - 1) / CT; a.tiles_y = (j.h + CT - 1) / CT; a.batch = batch;
  a.decim_dst = j.decim_dst; a.dw = j.decim_w; a.dh = j.decim_h;
  hipLaunchKernelGGL((gauss_chain_kernel<5, 6, 8>), dim3(a.tiles_x * a.tiles_y * batch), dim3(CNT), 0, st, a);
  return true;
}

// Level launches with the same tap count and a det-H / gradient plane of their source, all in one grid.
namespace {
template <int R>
void launch_multi_r(hipStream_t st, const MultiArgs& m, const MultiGeom& mg, int blocks) {
  static_assert(sizeof(MultiArgs) + sizeof(MultiGeom) <= 4000, "kernel argument segment");
  hipLaunchKernelGGL((gauss_multi_kernel<R>), dim3(blocks), dim3(NT), 0, st, m, mg);
}
}  // namespace
bool launch_gauss_multi(hipStream_t st, const GaussJob* jobs, int njobs, int batch, const LowLevels* low) {
  if (njobs < 1) return true;
  const int r = jobs[0].taps.fw >> 1;
  if (r < 8 || r > 12) return false;
  for (int k = 0; k < njobs; k++)
    if (!jobs[k].deth_src || !jobs[k].got_src || !jobs[k].deth_dst || (jobs[k].taps.fw >> 1) != r) return false;
  for (int k0 = 0; k0 < njobs; k0 += kMultiJobs) {
    MultiArgs m;
    MultiGeom mg;
    memset(&mg, 0, sizeof(mg));
    m.njobs = njobs - k0 < kMultiJobs ? njobs - k0 : kMultiJobs;
    m.batch = batch;
    int blocks = 0;
    for (int k = 0; k < m.njobs; k++) {
      m.j[k] = job_args(jobs[k0 + k], batch);
      m.first_block[k] = blocks;
      blocks += ((m.j[k].tiles_x * m.j[k].tiles_y * batch + 7) / 8) * 8;
    }
    for (int k = m.njobs; k < kMultiJobs; k++) { m.j[k] = m.j[0]; m.first_block[k] = blocks; }
    m.first_block[kMultiJobs] = blocks;
    m.first_block[m.njobs] = blocks;
    m.low_first = 0; m.low_nlv = 0; m.low_blocks = 1; m.gauss = nullptr; m.deth = nullptr; m.got = nullptr;
    for (int l = 0; l < kMaxLev; l++) m.nm.v[l] = 0.0f;
    if (low && low->nlv > 0 && k0 + kMultiJobs >= njobs) {  // with the last group of jobs
      mg.g = *low->g;
      m.low_first = low->first_oct; m.low_nlv = low->nlv;
      m.gauss = low->gauss; m.deth = low->deth; m.got = reinterpret_cast<float2*>(low->got);
      for (int l = 0; l < kMaxLev && l < low->g->dog + 2; l++) m.nm.v[l] = low->norms[l];
      int lb = 0;
      for (int o = low->first_oct; o < low->g->noct; o++) lb += low->nlv * (((low->g->o[o].wa >> 2) * low->g->o[o].h + 255) >> 8);
      if (lb > 0) { m.low_blocks = lb; blocks += lb * batch; }
    }
    switch (r) {
      case 8: launch_multi_r<8>(st, m, mg, blocks); break;
      case 9: launch_multi_r<9>(st, m, mg, blocks); break;
      case 10: launch_multi_r<10>(st, m, mg, blocks); break;
      case 11: launch_multi_r<11>(st, m, mg, blocks); break;
      default: launch_multi_r<12>(st, m, mg, blocks); break;
    }
  }
  return true;
}

void launch_convert(hipStream_t st, const void* src, int format, int pixtype, long long pitch,
                    long long img_stride, int ds, float* dst, int w, int h, int batch) {
  ConvArgs a;
  a.src = (const uint8_t*)src; a.pitch = pitch; a.img_stride = img_stride;
  a.format = format; a.pixtype = pixtype; a.ds = ds; a.dst = dst; a.w = w; a.h = h;
  hipLaunchKernelGGL(convert_kernel, dim3((w + 255) / 256, h, batch), dim3(256), 0, st, a);
}

void launch_upsample(hipStream_t st, const float* src, int w, int h, int log_scale, float* dst, int batch) {
  hipLaunchKernelGGL(upsample_kernel, dim3((w + 255) / 256, h << log_scale, batch), dim3(256), 0, st, src, w, h,
                     log_scale, dst);
}

void launch_downsample(hipStream_t st, const float* src, int sw, int splane, float* dst, int dw, int dh,
                       int batch) {
  hipLaunchKernelGGL(downsample_kernel, dim3((dw + 255) / 256, dh, batch), dim3(256), 0, st, src, sw, splane,
                     dst, dw, dh);
}

}  // namespace hess
