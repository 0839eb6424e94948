// k_gauss.hip -- Gaussian scale-space kernels for gfx950 (MI355X).
//
// Replaces FilterH<FW>/FilterV<FW> (two global passes through a scratch buffer, ProgramCU.cu:117-231,
// 455-512), DownsampleKernel (ProgramCU.cu:312-326) and the host-side pixel conversion
// (GLTexImage.cpp:802-916).  By bytes HBM-bound (4 B read + 4 B written per pixel and level, + 4/12 B for
// the fused planes); measured, vector-ALU issue is the contended resource.
//
// gauss_march_kernel (the shipped form): a workgroup (256 threads = 4 wavefronts) owns a strip of 64 columns and
// marches down a run of its rows, 32 source rows per step; every source row is staged and filtered horizontally ONCE
// (the 64x32 tile form below re-filters its 2R halo rows per tile: 1.31x (R = 5) to 1.63x (R = 10) the needed work):
//   stage    source rows [a-2, a+32) x cols [x0-R4, x0+64+R4) -> LDS `raw`, 16-byte global loads issued one step ahead
//            (they are in flight while the previous step computes), borders replicated as the reference clamps its
//            fetch index; the per-thread load slots keep their column/LDS addresses from step to step;
//   stage 1b (HESS) det-Hessian*sigma^4 and (gradient/2, theta) of the SOURCE level for rows [a-1, a+31) from the
//            staged window, in the same launch (ComputeHessian_Kernel, ProgramCU.cu:523-595): the level is not re-read;
//   H pass   one 8-output task per thread (32 rows x 8 groups) from a register window (ds_read_b128), results into a
//            ring of 32 + 2R rows kept twice (slot t and t + C) so that the vertical pass reads 4 + 2R consecutive rows
//            from a per-step base with immediate offsets, whatever the ring position;
//   V pass   output rows [a-R, a+32-R): a thread produces 4 rows x 2 columns (ds_read_b64, 8-byte coalesced stores,
//            512 contiguous bytes per wavefront); the launch that produces the down-sampling level also stores its even
//            rows and columns as level 0 of the next octave.
// Work is cut into equal runs of rows over the (image, strip, row) order, one run per workgroup and about as many
// workgroups as the chip holds at once (a run may end one strip and begin the next): no tail of half-empty rounds.
// gauss_kernel: the 64x32 tile form of rounds 1-2 (build with -DHESS_GAUSS_TILES=1 for A/B runs):
//   stage 1  source rows [y0-R, y0+32+R) x cols [x0-R4, x0+64+R4) -> LDS `s`, 16-byte global loads,
//            borders replicated exactly as the reference clamps its fetch index;
//   stage 1b (HESS) as above;
//   stage 2  horizontal pass LDS->LDS: a thread produces 8 adjacent outputs from a register window
//            (ds_read_b128, row stride = 4 mod 8 dwords: conflict-free);
//   stage 3  vertical pass LDS->HBM: a thread produces 4 rows x 2 columns (ds_read_b64, 8-byte
//            coalesced stores, 512 contiguous bytes per wavefront).
// Per output the taps are accumulated in the reference's order: v = 0; v = fma(x_i, k_i, v), i=0..FW-1.
#include <cstdlib>
#include <cstring>

#include "hess_dev.h"
#include "hess_devmath.h"

namespace hess {

namespace {

constexpr int TW = 64, TH = 32, NT = 256;
#ifndef HESS_GAUSS_TILES
#define HESS_GAUSS_TILES 1      // 0: the column-strip march instead of the 64x32 tile kernel (A/B builds; measured slower)
#endif
// Diagnostic build (-DHESS_GAUSS_STAMPS=1, tools/gauss_stamps.py): every wavefront adds the shader cycles it spends in
// each phase to a device-side table (s_memtime stamps; phase boundaries force the waits the real kernel leaves to the
// hardware, so the build is for SHARES, not for its run time).  Never in the product build.
#ifndef HESS_GAUSS_STAMPS
#define HESS_GAUSS_STAMPS 0
#endif
#if HESS_GAUSS_STAMPS
constexpr int kStampWaves = 1 << 16;
__device__ unsigned long long g_gstamp[kStampWaves * 8];  // per wavefront slot (private: no atomics), 7 phases + a visit count
__device__ __forceinline__ unsigned long long gstamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define GSTAMP_BEGIN unsigned long long st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 1}; unsigned long long st_prev_ = gstamp_now();
#define GSTAMP(k) do { const unsigned long long t_ = gstamp_now(); st_acc_[k] += t_ - st_prev_; st_prev_ = gstamp_now(); } while (0)
#define GSTAMP_END do { if ((threadIdx.x & 63) == 0) { unsigned long long* p_ = g_gstamp + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kStampWaves - 1)) * 8; \
    for (int q_ = 0; q_ < 8; q_++) p_[q_] += st_acc_[q_]; } } while (0)
#define GSTAMP_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define GSTAMP_BEGIN
#define GSTAMP(k)
#define GSTAMP_END
#define GSTAMP_VMWAIT()
#endif
#ifndef HESS_GAUSS_WG_PER_CU
#define HESS_GAUSS_WG_PER_CU 3  // march form: workgroups per CU the run length is sized for
#endif

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
  int rows_per_wg, nwg;  // march form: rows of the (image, strip, row) order per workgroup, workgroups with work
  Taps taps;
};

__device__ __forceinline__ float gtex1(const float* p, int n, int i) { return (i < 0 || i >= n) ? 0.0f : p[i]; }

// LDS floats of one tile: (32 + 2R) staged rows of 64 + 2*R4 (+4 pad) columns
template <int R>
constexpr int gauss_tile_lds() { return (TH + 2 * R) * (TW + 2 * ((R + 3) & ~3) + 4); }

// One 64x32 tile of one level: the body of gauss_kernel, and of either half of gauss_pair_kernel.  `block` = the
// workgroup's index among the launch's (or the half's) workgroups, `s` = gauss_tile_lds<R>() floats of LDS.
template <int R, bool U8, bool HESS>
__device__ __forceinline__ void gauss_tile(const GaussArgs& a, float* __restrict__ s, const int block) {
  constexpr int FW = 2 * R + 1;
  constexpr int R4 = (R + 3) & ~3;
  constexpr int OFF = R4 - R;
  constexpr int SW = TW + 2 * R4;
  constexpr int SWP = SW + 4;          // SW % 8 == 0 -> row stride = 4 (mod 8) dwords
  constexpr int ROWS = TH + 2 * R;
  constexpr int NG = SW / 4;           // 16-byte groups per staged row
  constexpr int NV = (OFF + 8 + 2 * R + 3) / 4;
  static_assert(ROWS * SWP == gauss_tile_lds<R>(), "LDS size");

  const int tid = threadIdx.x;
  const int w = a.w, h = a.h;
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

  GSTAMP_BEGIN
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
    int y = y0 - R + r;
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
  GSTAMP(0);  // index arithmetic + load issue
  GSTAMP_VMWAIT();
  GSTAMP(1);  // waiting for the source loads
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
  GSTAMP(2);  // LDS stores + barrier

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
          const float* base = &s[(hr + R - 1) * SWP + hx + R4];
#pragma unroll
          for (int rr = 0; rr < 3; rr++) {
            float* dst = rr == 0 ? U : (rr == 1 ? M : D);
            const float4 v = *reinterpret_cast<const float4*>(base + rr * SWP);
            dst[0] = base[rr * SWP - 1];
            dst[1] = v.x; dst[2] = v.y; dst[3] = v.z; dst[4] = v.w;
            dst[5] = base[rr * SWP + 4];
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
        if (gx == 0) {
          U[0] = gtex1(plane, n, idx - w - 1); M[0] = gtex1(plane, n, idx - 1); D[0] = gtex1(plane, n, idx + w - 1);
        }
        if (gx + 4 == w) {  // this thread owns the row's last pixel (w is a multiple of 4)
          const int il = idx + 3;
          U[5] = gtex1(plane, n, il - w + 1); M[5] = gtex1(plane, n, il + 1); D[5] = gtex1(plane, n, il + w + 1);
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

  GSTAMP(3);  // fused det-H / gradient stage (compute + store issue)
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
#pragma unroll
        for (int i = 0; i < NV; i++) {
          float4 q = *reinterpret_cast<const float4*>(&s[r * SWP + xb + 4 * i]);
          win[4 * i] = q.x; win[4 * i + 1] = q.y; win[4 * i + 2] = q.z; win[4 * i + 3] = q.w;
        }
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
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NTASK; k++) {
      const int task = tid + k * NT;
      if (task < ROWS * (TW / 8)) {
        const int r = task >> 3, xb = (task & 7) * 8;
        *reinterpret_cast<float4*>(&s[r * SWP + xb]) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        *reinterpret_cast<float4*>(&s[r * SWP + xb + 4]) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
      }
    }
  }
  __syncthreads();
  GSTAMP(4);  // horizontal pass + its two barriers

  // ---- stage 3: vertical pass, LDS -> HBM ----
  {
    const int cg = tid & 31, rg = tid >> 5;
    float2 col[4 + 2 * R];
#pragma unroll
    for (int i = 0; i < 4 + 2 * R; i++)
      col[i] = *reinterpret_cast<const float2*>(&s[(rg * 4 + i) * SWP + cg * 2]);
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
  GSTAMP(5);  // vertical pass (compute + store issue)
  GSTAMP_VMWAIT();
  GSTAMP(6);  // stores draining
  GSTAMP_END;
}

template <int R, bool U8, bool HESS>
__global__ __launch_bounds__(NT) void gauss_kernel(GaussArgs a) {
  __shared__ __attribute__((aligned(16))) float s[gauss_tile_lds<R>()];
  gauss_tile<R, U8, HESS>(a, s, (int)blockIdx.x);
}

// Two level launches that do not depend on each other in one grid: the top level of octave o (taps RA, its source level
// has a gradient plane) and level 1 of octave o+1 (taps RB) -- T(o, l) = 3o + l is the earliest step of level l of
// octave o, so level 4 of one octave and level 1 of the next are due together.  The small half hides behind the large
// one: six launches fewer in the dependent chain of a 1080p pyramid.  The first blocks_a workgroups do half a.
template <int RA, int RB>
__global__ __launch_bounds__(NT) void gauss_pair_kernel(GaussArgs a, GaussArgs b, int blocks_a) {
  constexpr int LDS = gauss_tile_lds<RA>() > gauss_tile_lds<RB>() ? gauss_tile_lds<RA>() : gauss_tile_lds<RB>();
  __shared__ __attribute__((aligned(16))) float s[LDS];
  if ((int)blockIdx.x < blocks_a) gauss_tile<RA, false, true>(a, s, (int)blockIdx.x);  // (workgroup-uniform)
  else gauss_tile<RB, false, true>(b, s, (int)blockIdx.x - blocks_a);
}

// ------------------------------------------------------------------------------------------------
// Column-strip march (see the file header).  RS source rows per step; ring of C = RS + R2 horizontally filtered rows
// (R2 = 2R rounded up to a multiple of 4), every row stored at slot t and t + C.
constexpr int RS = 32;

template <int R, bool U8, bool HESS>
__global__ __launch_bounds__(NT) void gauss_march_kernel(GaussArgs a) {
  constexpr int FW = 2 * R + 1;
  constexpr int R4 = (R + 3) & ~3;
  constexpr int OFF = R4 - R;
  constexpr int SW = TW + 2 * R4;       // staged columns per row
  constexpr int RWP = SW + 4;           // SW % 8 == 0 -> raw row pitch = 4 (mod 8) dwords
  constexpr int RROWS = RS + 2;         // source rows a-2 .. a+RS-1: the two rows above feed the fused det-H stage
  constexpr int NG = SW / 4;            // 16-byte groups per staged row
  constexpr int NIT = (RROWS * NG + NT - 1) / NT;
  constexpr int R2 = (2 * R + 3) & ~3;
  constexpr int C = RS + R2;            // ring rows
  constexpr int HP = TW + 4;            // ring row pitch
  constexpr int NV = (OFF + 8 + 2 * R + 3) / 4;
  static_assert(RS * (TW / 8) == NT, "one horizontal task per thread");
  static_assert(2 * R <= RS, "ring geometry");

  __shared__ __attribute__((aligned(16))) float raw[RROWS * RWP];
  __shared__ __attribute__((aligned(16))) float ring[2 * C * HP];

  const int tid = threadIdx.x;
  const int w = a.w, h = a.h;
  // XCD-aware order of the runs: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), so XCD k
  // takes the k-th contiguous eighth of the run sequence: runs that share halo columns / boundary rows -- neighbours
  // in the (image, strip, row) order, a few runs apart -- are served by the same 4 MiB L2 at about the same time.
  const int per_xcd = (a.nwg + 7) >> 3;
  const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (wid >= a.nwg) return;
  GSTAMP_BEGIN
  const long long total = (long long)a.batch * a.tiles_x * h;
  long long g = (long long)wid * a.rows_per_wg;
  const long long gend = min(total, g + (long long)a.rows_per_wg);

  // per-thread constants of the passes
  const int hrow = tid >> 3, hxb = (tid & 7) * 8;  // horizontal task: source row a + hrow, outputs hxb .. hxb+7
  const int cg = tid & 31, rg = tid >> 5;          // vertical task: columns 2cg, 2cg+1, rows 4rg .. 4rg+3 of the step

  while (g < gend) {  // one piece = rows [ys, ye) of one strip of one image (a run may end a strip and begin the next)
    const int colid = (int)(g / h);
    const int ys = (int)(g - (long long)colid * h);
    const int ye = (int)min((long long)h, (long long)ys + (gend - g));
    const int img = colid / a.tiles_x;
    const int x0 = (colid - img * a.tiles_x) * TW;
    g += ye - ys;
    const int nsteps = (ye - ys + 2 * R + RS - 1) / RS;
    const bool border = (x0 - R4 < 0) || (x0 + TW + R4 > w);  // strip-uniform

    // load slots of this thread: fixed LDS destination and source column, row advancing by RS per step
    int ldst[NIT], lrow[NIT], lcol[NIT], lflag[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      int gi = it * NT + tid;
      gi = gi < RROWS * NG ? gi : RROWS * NG - 1;  // surplus threads repeat the last group
      const int r = gi / NG, gx = gi - r * NG;
      const int x = x0 - R4 + gx * 4;
      ldst[it] = r * RWP + gx * 4;
      lrow[it] = r;
      lcol[it] = x < 0 ? 0 : (x >= w ? w - 4 : x);  // w is a multiple of 4
      lflag[it] = x < 0 ? 1 : (x >= w ? 2 : 0);     // group left / right of the image: replicate the edge pixel
    }
    float4 stage[NIT];
    uint32_t stage8[NIT];
    auto issue_loads = [&](int a0) {  // source rows a0-2 .. a0+RS-1, clamped (ProgramCU.cu:138, :201)
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        int y = a0 - 2 + lrow[it];
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        if (U8) {
          const uint8_t* row = a.src_u8 + (long long)img * a.src_img_stride + (long long)y * a.src_pitch;
          stage8[it] = *reinterpret_cast<const uint32_t*>(row + lcol[it]);
        } else {
          const float* row = a.src + (long long)img * a.src_img_stride + (long long)y * a.src_pitch;
          stage[it] = *reinterpret_cast<const float4*>(row + lcol[it]);
        }
      }
    };
    issue_loads(ys - R);
    int ws = 0;  // ring slot of source row a - R2 (the top of the window the vertical pass may read)

    for (int s = 0; s < nsteps; s++) {
      const int a0 = ys - R + s * RS;  // first new source row of this step
      GSTAMP(0);  // loop / piece bookkeeping
      GSTAMP_VMWAIT();
      GSTAMP(1);  // waiting for the loads issued one step ahead
      // ---- stage: registers -> LDS (border groups patched on the way), then the next step's loads ----
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        float4 v;
        if (U8) {
          const uint32_t b = stage8[it];
          v = make_float4(dm_u8_unit((float)(b & 0xFFu)), dm_u8_unit((float)((b >> 8) & 0xFFu)),   // p / 255.0f, GLTexImage.cpp:828
                          dm_u8_unit((float)((b >> 16) & 0xFFu)), dm_u8_unit((float)(b >> 24)));
        } else {
          v = stage[it];
        }
        if (border) {
          const float e = lflag[it] == 1 ? v.x : v.w;
          if (lflag[it]) v = make_float4(e, e, e, e);
        }
        *reinterpret_cast<float4*>(&raw[ldst[it]]) = v;
      }
      if (s + 1 < nsteps) issue_loads(a0 + RS);  // in flight while this step computes
      __syncthreads();
      GSTAMP(2);  // LDS stores, next loads issued, barrier

      // ---- stage 1b (HESS): det-Hessian*sigma^4 and (gradient, theta) of source rows [a0-1, a0+RS-1) from the staged
      // window (ComputeHessian_Kernel, ProgramCU.cu:523-595): raw row j holds source row a0-2+j.  A thread does 4
      // adjacent pixels in each of two rows 16 apart (store shapes: see gauss_kernel below). ----
      if (HESS) {
        const int hx = (tid & 15) * 4;
        const int gx = x0 + hx;
        const bool want_got = a.got_src != nullptr;  // block-uniform
        const float* plane = a.src + (long long)img * a.src_img_stride;
        const int n = w * h;
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const int hr = (tid >> 4) + 16 * half;
          const int gy = a0 - 1 + hr;
          if (gy >= ys && gy < ye && gx < w) {
            float U[6], M[6], D[6];  // columns gx-1 .. gx+4 of rows gy-1, gy, gy+1
            {
              const float* base = &raw[hr * RWP + hx + R4];
#pragma unroll
              for (int rr = 0; rr < 3; rr++) {
                float* dst = rr == 0 ? U : (rr == 1 ? M : D);
                const float4 v = *reinterpret_cast<const float4*>(base + rr * RWP);
                dst[0] = base[rr * RWP - 1];
                dst[1] = v.x; dst[2] = v.y; dst[3] = v.z; dst[4] = v.w;
                dst[5] = base[rr * RWP + 4];
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
            if (gx == 0) {
              U[0] = gtex1(plane, n, idx - w - 1); M[0] = gtex1(plane, n, idx - 1); D[0] = gtex1(plane, n, idx + w - 1);
            }
            if (gx + 4 == w) {  // this thread owns the row's last pixel (w is a multiple of 4)
              const int il = idx + 3;
              U[5] = gtex1(plane, n, il - w + 1); M[5] = gtex1(plane, n, il + 1); D[5] = gtex1(plane, n, il + w + 1);
            }
            float hv[4];
            float2 gv[4];
#pragma unroll
            for (int j = 0; j < 4; j += 2) {  // two pixels at a time on 2-vectors, no per-pixel branches
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
            const long long o = (long long)img * w * h + idx;
            *reinterpret_cast<float4*>(a.deth_src + o) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            if (want_got) {
              float2* gp = a.got_src + o;
              *reinterpret_cast<float4*>(gp) = make_float4(gv[0].x, gv[0].y, gv[1].x, gv[1].y);
              *reinterpret_cast<float4*>(gp + 2) = make_float4(gv[2].x, gv[2].y, gv[3].x, gv[3].y);
            }
          }
        }
      }

      GSTAMP(3);  // fused det-H / gradient stage (compute + store issue)
      // ---- horizontal pass: source row a0 + hrow (raw row hrow + 2), 8 outputs, into the ring (both copies) ----
      {
        float win[NV * 4];
#pragma unroll
        for (int i = 0; i < NV; i++) {
          const float4 q = *reinterpret_cast<const float4*>(&raw[(hrow + 2) * RWP + hxb + 4 * i]);
          win[4 * i] = q.x; win[4 * i + 1] = q.y; win[4 * i + 2] = q.z; win[4 * i + 3] = q.w;
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < FW; i++) {
          const float ki = a.taps.k[i];
#pragma unroll
          for (int j = 0; j < 8; j++) acc[j] = fmaf(win[OFF + j + i], ki, acc[j]);  // ProgramCU.cu:152
        }
        int t = ws + R2 + hrow;  // ring slot of source row a0 + hrow
        t = t >= C ? t - C : t;
        float* d = &ring[t * HP + hxb];
        const float4 lo = make_float4(acc[0], acc[1], acc[2], acc[3]), hi = make_float4(acc[4], acc[5], acc[6], acc[7]);
        *reinterpret_cast<float4*>(d) = lo;
        *reinterpret_cast<float4*>(d + 4) = hi;
        *reinterpret_cast<float4*>(d + C * HP) = lo;
        *reinterpret_cast<float4*>(d + C * HP + 4) = hi;
      }
      __syncthreads();
      GSTAMP(4);  // horizontal pass + barrier

      // ---- vertical pass: output rows a0 - R + 4rg + j from ring rows (window-relative) R2 - 2R + 4rg + j + i ----
      {
        const int y0 = a0 - R + 4 * rg;
        if (y0 + 3 >= ys && y0 < ye) {
          const float* cbase = &ring[(ws + (R2 - 2 * R) + 4 * rg) * HP + cg * 2];
          float2 col[4 + 2 * R];
#pragma unroll
          for (int i = 0; i < 4 + 2 * R; i++) col[i] = *reinterpret_cast<const float2*>(cbase + i * HP);
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
            float* d = a.dst + (long long)img * w * h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int y = y0 + j;
              if (y >= ys && y < ye) *reinterpret_cast<float2*>(&d[(long long)y * w + x]) = acc[j];
            }
            if (a.decim_dst) {  // block-uniform.  x is even; the even rows and column .x are the sampled ones
              float* dd = a.decim_dst + (long long)img * a.dw * a.dh;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const int y = y0 + j;
                if (!(y & 1) && y >= ys && y < ye && (y >> 1) < a.dh) {
                  float* row = dd + (long long)(y >> 1) * a.dw;
                  if ((x >> 1) < a.dw) row[x >> 1] = acc[j].x;  // (the next octave may be narrower than w/2: widths halve unaligned)
                  // columns of the next octave beyond w/2 (its width is aligned up to 4) repeat the source's last column
                  if (x == w - 2) for (int xx = w >> 1; xx < a.dw; xx++) row[xx] = acc[j].y;
                }
              }
            }
          }
        }
      }
      GSTAMP(5);  // vertical pass (compute + store issue)
      ws += RS;
      ws = ws >= C ? ws - C : ws;
    }
    // (the next piece's first stage store follows this piece's last barrier in program order for every thread, and
    // its first ring store follows its own stage barrier: no barrier needed between pieces)
  }
  GSTAMP_END;
}

template <int R>
void launch_r(hipStream_t st, GaussArgs a, int batch) {
  a.tiles_x = (a.w + TW - 1) / TW;
  a.tiles_y = (a.h + TH - 1) / TH;
  a.batch = batch;
#if HESS_GAUSS_TILES
  const int ntile = a.tiles_x * a.tiles_y * batch;
  dim3 grid(((ntile + 7) / 8) * 8);
  if (a.src_u8)
    hipLaunchKernelGGL((gauss_kernel<R, true, false>), grid, dim3(NT), 0, st, a);
  else if (a.deth_src)
    hipLaunchKernelGGL((gauss_kernel<R, false, true>), grid, dim3(NT), 0, st, a);
  else
    hipLaunchKernelGGL((gauss_kernel<R, false, false>), grid, dim3(NT), 0, st, a);
#else
  // Runs of rows_per_wg rows of the (image, strip, row) order, one per workgroup; about as many workgroups as the chip
  // holds at once (256 CUs x HESS_GAUSS_WG_PER_CU), each run at least 64 rows, and (run + 2R) a multiple of the step so
  // that whole runs waste no step slots.
  const long long total = (long long)batch * a.tiles_x * a.h;
  long long rows = (total + 256 * HESS_GAUSS_WG_PER_CU - 1) / (256 * HESS_GAUSS_WG_PER_CU);
  if (rows < 64) rows = 64;
  rows = ((rows + 2 * R + RS - 1) / RS) * RS - 2 * R;
  if (rows < RS) rows = RS;
  a.rows_per_wg = (int)rows;
  a.nwg = (int)((total + rows - 1) / rows);
  dim3 grid(((a.nwg + 7) / 8) * 8);
  if (a.src_u8)
    hipLaunchKernelGGL((gauss_march_kernel<R, true, false>), grid, dim3(NT), 0, st, a);
  else if (a.deth_src)
    hipLaunchKernelGGL((gauss_march_kernel<R, false, true>), grid, dim3(NT), 0, st, a);
  else
    hipLaunchKernelGGL((gauss_march_kernel<R, false, false>), grid, dim3(NT), 0, st, a);
#endif
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

void launch_gauss(hipStream_t st, const float* src, const uint8_t* src_u8, long long src_pitch,
                  long long src_img_stride, float* dst, int wa, int h, int batch, const Taps& taps,
                  float* deth_src, float* got_src, float norm_src, float* decim_dst, int decim_w, int decim_h) {
  GaussArgs a;
  a.decim_dst = decim_dst; a.dw = decim_w; a.dh = decim_h;
  a.src = src; a.src_u8 = src_u8; a.src_pitch = src_pitch; a.src_img_stride = src_img_stride;
  a.dst = dst; a.w = wa; a.h = h; a.taps = taps;
  a.deth_src = deth_src; a.got_src = reinterpret_cast<float2*>(got_src); a.norm_src = norm_src;
  switch (taps.fw >> 1) {
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

#if HESS_GAUSS_STAMPS
}  // namespace hess
// diagnostic build only: read and clear the phase table (tools/gauss_stamps.py)
extern "C" int hess_debug_gauss_stamps(unsigned long long* out) {  // out[8]: sums over the wavefront slots
  static unsigned long long host[hess::kStampWaves * 8];
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(hess::g_gstamp), sizeof(host)) != hipSuccess) return -1;
  for (int q = 0; q < 8; q++) out[q] = 0;
  for (int w = 0; w < hess::kStampWaves; w++) for (int q = 0; q < 8; q++) out[q] += host[(size_t)w * 8 + q];
  memset(host, 0, sizeof(host));
  return hipMemcpyToSymbol(HIP_SYMBOL(hess::g_gstamp), host, sizeof(host)) == hipSuccess ? 0 : -1;
}
namespace hess {
#endif

namespace {
GaussArgs job_args(const GaussJob& j, int batch) {
  GaussArgs a;
  a.decim_dst = j.decim_dst; a.dw = j.decim_w; a.dh = j.decim_h;
  a.src = j.src; a.src_u8 = nullptr; a.src_pitch = j.wa; a.src_img_stride = (long long)j.wa * j.h;
  a.dst = j.dst; a.w = j.wa; a.h = j.h; a.taps = j.taps;
  a.deth_src = j.deth_src; a.got_src = reinterpret_cast<float2*>(j.got_src); a.norm_src = j.norm_src;
  a.tiles_x = (j.wa + TW - 1) / TW; a.tiles_y = (j.h + TH - 1) / TH; a.batch = batch;
  a.rows_per_wg = 0; a.nwg = 0;
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

// Level launches a (the larger: top level of an octave) and b (level 1 of the next octave) in one grid.  Instantiated
// for the tap counts around the reference's default schedule (a: 17-25 taps, b: 9-13); false = not this pair, launch
// them one after the other.
bool launch_gauss_pair(hipStream_t st, const GaussJob& ja, const GaussJob& jb, int batch) {
#if HESS_GAUSS_TILES
  if (!ja.deth_src || !jb.deth_src) return false;
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
#else
  (void)st; (void)ja; (void)jb; (void)batch;
  return false;
#endif
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
