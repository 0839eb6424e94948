// k_detect.hip -- det-Hessian / gradient planes, 3x3x3 extrema scan with sub-pixel refinement,
// ordered feature-list generation and top-K selection for gfx950 (MI355X).
//
// Replaces ComputeHessian_Kernel (ProgramCU.cu:523-595), ComputeKEY_Kernel (:657-882), the 16 B/px
// key map + ListGen_Kernel second pass (:924-1051), DetectionData* host round trips (:3057-3079)
// and the bitonic-sort/Blelloch top-K chain (:2205-3051).
//
// Structure (no key map, no host round trip, every candidate evaluated ONCE):
//   hessian_rows4  det-H of the octave's top level (the other levels come fused out of the Gaussian
//                  launches, k_gauss.hip); hessian_kernel is the general one-row form (det-H + gradient);
//   extrema_stream the scan (dog <= 5): register-streaming 26-neighbour filter, candidates queued for the
//                  reference's exact test; an accepted pixel is written -- complete, as a RawKey -- to the image's
//                  UNORDERED list of detections (one wave-aggregated atomic per batch of candidates), sets its
//                  positional bit in the row's mask words, counts in its row and in the top-K key histogram
//                  (extrema_mark: LDS-tiled variant for larger level counts / planes);
//   extrema_place  list order: the first workgroup of an image to arrive scans the row counts in list order (level,
//                  row) -> row offsets, level totals, -tc level truncation; every detection's position is its row's
//                  offset + the number of mask bits below its column (computed while the scan is awaited): the
//                  (level, row, col) order of the raw list, deterministic, with one 32-byte copy per detection;
//   topk           15-bit histogram of abs(half(response)) -> exact cut, then one ordered
//                  compaction pass (ties at the cut resolved towards the lower list index).
// (Rounds 1-5 marked the pixels in the scan and evaluated every detection a second time in an ordered scatter pass:
//  row_scan_kernel + extrema_scatter_kernel, 63 us and 128 MB of det-H re-reads per step of eight 1080p images.)
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include "hess_dev.h"
#include "hess_devmath.h"
#include "hess_planes.h"

namespace hess {

namespace {

// =============================== det-Hessian + gradient ======================================

struct HessArgs {
  const float* gauss;
  float* deth;
  float2* got;
  int wa, h, plane, B, dog, level_first, batch;
  float inv_groups;  // 1 / (wa/4)
  long long lvl_off, got_off;
  float norm[kMaxLev];  // sigma_l^4 (host passes sigma^2, wrapper squares it: ProgramCU.cu:592)
};

__global__ __launch_bounds__(256) void hessian_kernel(HessArgs a) {
  const int z = blockIdx.y;  // l * batch + b
  const int l = a.level_first + z / a.batch, b = z % a.batch;
  const long long poff = a.lvl_off + ((long long)l * a.B + b) * a.plane;
  const bool want_got = (l >= 1 && l <= a.dog);
  hessian_rows_body(a.gauss + poff, a.deth + poff, want_got ? a.got + a.got_off + ((long long)(l - 1) * a.B + b) * a.plane : nullptr,
                    a.wa, a.h, a.norm[l], blockIdx.x * 256 + threadIdx.x);
}

// det-Hessian only (levels without a gradient plane: the top level of an octave), 4 px x 4 rows per
// thread: six 16-byte row loads give four output rows, so every source row is fetched 1.5 times instead of
// 3 (the one-row kernel above measured 2x its algorithmic bytes at HBM: the three readers of a row land
// on different XCDs).  Same 1-D neighbour semantics: the columns left/right of the group come from the
// adjacent lanes, and from explicit 1-D-indexed loads at row ends and wavefront edges.
// det-H of one level of every octave in a single launch (the octaves' top levels: nobody's source level, so no
// Gaussian launch computes it on the side).  1 thread = 4 px x 4 rows; neighbour columns from the adjacent lanes.
__global__ __launch_bounds__(256) void hessian_rows4_kernel(Geom g, const float* gauss, float* deth, int level, float norm,
                                                            uint4* zero, long long zero_n) {
  // On the side: clear what the detection stages expect zeroed (overflow words, row counts, top-K histogram, extrema
  // masks: one allocation, hess_plan.hip) -- a grid-stride fill by this launch's threads instead of a fill
  // launch of its own in the dependent chain.
  if (zero) {
    const long long nthr = (long long)gridDim.x * gridDim.y * 256;
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_n; i += nthr)
      zero[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  int blk = blockIdx.x, o = 0;  // block -> octave: octaves back to back, whole blocks each (uniform scalar walk)
  for (; o < g.noct - 1; o++) {
    const int nb = ((g.o[o].wa >> 2) * ((g.o[o].h + 3) >> 2) + 255) >> 8;
    if (blk < nb) break;
    blk -= nb;
  }
  const int wa = g.o[o].wa, h = g.o[o].h, n = g.o[o].plane;
  const int groups_per_row = wa >> 2;
  const int bands = (h + 3) >> 2;
  const int gid = blk * 256 + threadIdx.x;
  if (gid >= groups_per_row * bands) return;
  int band = (int)(((float)gid + 0.5f) * (1.0f / (float)groups_per_row));
  int rem = gid - band * groups_per_row;
  if (rem < 0) { band--; rem += groups_per_row; }
  else if (rem >= groups_per_row) { band++; rem -= groups_per_row; }
  const int x = rem << 2, r0 = band << 2;
  const int b = blockIdx.y;
  const long long poff = g.o[o].lvl_off + ((long long)level * g.B + b) * n;
  const float* src = gauss + poff;
  const int lane = threadIdx.x & 63;
  const bool edge_l = (x == 0) || (lane == 0);
  const bool edge_r = (x + 4 == wa) || (lane == 63) || (gid == groups_per_row * bands - 1);

  float R[6][6];  // rows r0-1 .. r0+4, columns x-1 .. x+4
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int row = r0 - 1 + k;
    const bool in = row >= 0 && row < h;
    const float4 q = *reinterpret_cast<const float4*>(src + (in ? row : 0) * wa + x);
    R[k][1] = in ? q.x : 0.0f; R[k][2] = in ? q.y : 0.0f; R[k][3] = in ? q.z : 0.0f; R[k][4] = in ? q.w : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    R[k][0] = lane_prev(R[k][4]);
    R[k][5] = lane_next(R[k][1]);
  }
  if (edge_l) {
#pragma unroll
    for (int k = 0; k < 6; k++) R[k][0] = tex1(src, n, (r0 - 1 + k) * wa + x - 1);
  }
  if (edge_r) {
#pragma unroll
    for (int k = 0; k < 6; k++) R[k][5] = tex1(src, n, (r0 - 1 + k) * wa + x + 4);
  }
#pragma unroll
  for (int k = 1; k <= 4; k++) {
    const int row = r0 + k - 1;
    if (row >= h) break;
    float hv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float v11 = R[k - 1][j], v12 = R[k - 1][j + 1], v13 = R[k - 1][j + 2];
      const float v21 = R[k][j], v22 = R[k][j + 1], v23 = R[k][j + 2];
      const float v31 = R[k + 1][j], v32 = R[k + 1][j + 1], v33 = R[k + 1][j + 2];
      hv[j] = dm_deth(v11, v12, v13, v21, v22, v23, v31, v32, v33, norm);  // ProgramCU.cu:536-553
    }
    store_stream_f4(deth + poff + row * wa + x, hv[0], hv[1], hv[2], hv[3]);
  }
}

// =============================== extrema test ================================================

struct KeyVal {
  uint32_t packed;
  float dx, dy, ds;
};

#define HESS_READ_CMP(d0, d1, d2, tex, i)                                        \
  d0 = tex[(i) - 1]; d1 = tex[(i)]; d2 = tex[(i) + 1];                           \
  if (response > nmax) {                                                         \
    nmax = fmaxf(nmax, d0); nmax = fmaxf(nmax, d1); nmax = fmaxf(nmax, d2);      \
    if ((response < nmax) || (response < 0)) return false;                       \
  } else {                                                                       \
    nmin = fminf(nmin, d0); nmin = fminf(nmin, d1); nmin = fminf(nmin, d2);      \
    if ((response > nmin) || (response > 0)) return false;                       \
  }

// ComputeKEY_Kernel body for one interior pixel (ProgramCU.cu:725-857).  The comparison macro
// re-selects its branch per triple with the running nmax exactly as READ_CMP_DOG_DATA does.
// PRE = true stops after the 26-neighbour and edge tests (everything before the sub-pixel solve) and
// reports whether the pixel is still a candidate; the arithmetic is the same code either way.
template <bool PRE = false>
__device__ __forceinline__ bool key_eval(const float* texC, const float* texP, const float* texN,
                                         const float* texG, int width, int index, const DetectParams& dp,
                                         KeyVal* out, long long gindex = -1) {
  float d00, d01, d02, d10, d11, d12, d20, d21, d22;
  float p00, p01, p02, p10, p11, p12, p20, p21, p22;
  float n00, n01, n02, n10, n11, n12, n20, n21, n22;
  float response, nmax, nmin;
  float dx = 0, dy = 0, ds = 0;
  bool offset_test_passed = true;
  const int i0 = index - width, i1 = index, i2 = index + width;

  d11 = response = texC[i1];
  if (fabsf(response) <= dp.thr0) return false;
  d10 = texC[i1 - 1];
  d12 = texC[i1 + 1];
  nmax = fmaxf(d10, d12);
  nmin = fminf(d10, d12);
  if ((response <= nmax) && (response >= nmin)) return false;
  HESS_READ_CMP(d00, d01, d02, texC, i0);
  HESS_READ_CMP(d20, d21, d22, texC, i2);

  const float vx2 = response * 2.0f;
  const float fxx = d10 + d12 - vx2;
  const float fyy = d01 + d21 - vx2;
  const float fxy = 0.25f * (d22 + d00 - d20 - d02);
  const float temp1 = fmaf(fxx, fyy, -(fxy * fxy));
  const float temp2 = (fxx + fyy) * (fxx + fyy);
  if ((temp1 <= 0) || (temp2 > dp.edge * temp1)) return false;

  HESS_READ_CMP(p00, p01, p02, texP, i0);
  HESS_READ_CMP(p10, p11, p12, texP, i1);
  HESS_READ_CMP(p20, p21, p22, texP, i2);
  HESS_READ_CMP(n00, n01, n02, texN, i0);
  HESS_READ_CMP(n10, n11, n12, texN, i1);
  HESS_READ_CMP(n20, n21, n22, texN, i2);
  (void)p00; (void)p02; (void)p20; (void)p22; (void)n00; (void)n02; (void)n20; (void)n22;
  if (PRE) return true;

  if (dp.subpixel) {  // ProgramCU.cu:769-825
    const float fx = 0.5f * (d12 - d10);
    const float fy = 0.5f * (d21 - d01);
    const float fs = 0.5f * (n11 - p11);
    const float fss = (n11 + p11 - vx2);
    const float fxs = 0.25f * (n12 + p10 - n10 - p12);
    const float fys = 0.25f * (n21 + p01 - n01 - p21);
    float4 A0 = (fxx > 0) ? make_float4(fxx, fxy, fxs, -fx) : make_float4(-fxx, -fxy, -fxs, fx);
    float4 A1 = (fxy > 0) ? make_float4(fxy, fyy, fys, -fy) : make_float4(-fxy, -fyy, -fys, fy);
    float4 A2 = (fxs > 0) ? make_float4(fxs, fys, fss, -fs) : make_float4(-fxs, -fys, -fss, fs);
    const float maxa = fmaxf(fmaxf(A0.x, A1.x), A2.x);
    if (maxa >= 1e-10) {
      if (maxa == A1.x) { float4 T = A1; A1 = A0; A0 = T; }
      else if (maxa == A2.x) { float4 T = A2; A2 = A0; A0 = T; }
      A0.y /= A0.x; A0.z /= A0.x; A0.w /= A0.x;
      A1.y = fmaf(-A1.x, A0.y, A1.y); A1.z = fmaf(-A1.x, A0.z, A1.z); A1.w = fmaf(-A1.x, A0.w, A1.w);
      A2.y = fmaf(-A2.x, A0.y, A2.y); A2.z = fmaf(-A2.x, A0.z, A2.z); A2.w = fmaf(-A2.x, A0.w, A2.w);
      if (fabsf(A2.y) > fabsf(A1.y)) { float4 T = A2; A2 = A1; A1 = T; }
      if (fabsf(A1.y) >= 1e-10) {
        A1.z /= A1.y; A1.w /= A1.y;
        A2.z = fmaf(-A2.y, A1.z, A2.z); A2.w = fmaf(-A2.y, A1.w, A2.w);
        if (fabsf(A2.z) >= 1e-10) {
          ds = A2.w / A2.z;
          dy = fmaf(-ds, A1.z, A1.w);
          dx = fmaf(-dy, A0.y, fmaf(-ds, A0.z, A0.w));
          response = fmaf(0.5f, fmaf(ds, fs, fmaf(dx, fx, dy * fy)), d11);
          offset_test_passed = (fabsf(response) > dp.thr) && (fabsf(ds) < 1.0f) && (fabsf(dx) < 1.0f) &&
                               (fabsf(dy) < 1.0f);
        }
      }
    }
  }
  if (!offset_test_passed) return false;
  if (out) {
    uint32_t type;  // ProgramCU.cu:828-851
    if (response < 0) type = 2u;
    else {
      const long long gi = gindex >= 0 ? gindex : (long long)i1;  // (texG may have its own pitch: the LDS-tiled scan)
      const float g0 = texG[gi - 1], g1 = texG[gi], g2 = texG[gi + 1];
      const float Lxx = fmaf(-2.0f, g1, g0) + g2;
      type = (Lxx > 0) ? 0u : 1u;
    }
    out->packed = (dm_f2h(response) << 16) | 0x4u | type;  // ProgramCU.cu:865
    out->dx = dx; out->dy = dy; out->ds = ds;
  }
  return true;
}

// Where an image's detections go.  Every scan task (a wavefront's strip segment, or a tile of the LDS-tiled scan) owns
// DT_SLOTS slots of the image's UNORDERED detection store -- no atomic to claim them, the task counts for itself; what a
// task finds beyond its slots goes to the image's shared spill list (one atomic per wavefront call, rare).  List order is
// rebuilt from the row's positional mask bits and count (extrema_place_kernel).
constexpr int DT_SLOTS = kDetectSlots;
struct DetectSink {
  RawKey* slots;              // this task's DT_SLOTS slots
  RawKey* spill;              // [cap_spill] of this image
  int* spill_count;           // of this image (arrives zeroed)
  int cap_spill;
  unsigned long long* mask;   // this image's mask words
  int* rowcnt;                // this image's row counts
  unsigned* hist;             // this image's histogram of abs(half(response)) (null: no top-K)
};

// One accepted detection per lane (`ok`, m = its ballot), called by the whole wavefront; `before` = detections of the
// task before this call (wavefront-uniform).  Every lane writes its own complete record.
__device__ __forceinline__ void post_detections(const DetectSink& sk, int before, bool ok, uint64_t m, const KeyVal& kv,
                                                int level_index, int row, int col, long long mask_word, int row_index) {
  if (!m) return;
  const int idx = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  const int nspill = before + __popcll(m) - max(before, DT_SLOTS);  // (uniform) of this call's detections, those past the slots
  int sbase = 0;
  if (nspill > 0) {
    const int lane = threadIdx.x & 63, first = __builtin_ctzll(m);
    if (lane == first) sbase = atomicAdd(sk.spill_count, nspill);
    sbase = __shfl(sbase, first);
  }
  if (ok) {
    RawKey* dstk = nullptr;
    if (idx < DT_SLOTS) dstk = sk.slots + idx;
    else {
      const int j = sbase + idx - max(before, DT_SLOTS);
      // (more than cap: the row counts say so, the host grows the lists and runs the batch again)
      if (j < sk.cap_spill) dstk = sk.spill + j;
    }
    if (dstk) {
      uint4* const dst = reinterpret_cast<uint4*>(dstk);
      dst[0] = make_uint4((uint32_t)level_index, (uint32_t)col, (uint32_t)row, kv.packed);
      dst[1] = make_uint4(__float_as_uint(kv.dx), __float_as_uint(kv.dy), __float_as_uint(kv.ds), 0u);
    }
    atomicOr(&sk.mask[mask_word], 1ull << (col & 63));
    atomicAdd(&sk.rowcnt[row_index], 1);
    // histogram of the top-K selection key abs(half(response)), 15 bits (topk_select_kernel): counted here, where the
    // response has just been computed
    if (sk.hist) atomicAdd(&sk.hist[(kv.packed >> 16) & 0x7fffu], 1u);
  }
}

struct RowTask {
  int b, o, l, row, li;
  bool valid;
};

__device__ __forceinline__ RowTask decode_row(const Geom& g, int wave, int batch) {
  RowTask t;
  t.valid = wave < batch * g.NR;
  t.b = wave / g.NR;
  int ri = wave - t.b * g.NR;
  int o = 0;
  for (int k = 1; k < g.noct; k++)
    if (g.o[k].row_base <= ri) o = k;
  t.o = o;
  int rel = ri - g.o[o].row_base;
  int lm1 = rel / g.o[o].h;
  t.l = lm1 + 1;
  t.row = rel - lm1 * g.o[o].h;
  t.li = o * g.dog + lm1;
  return t;
}

// Extrema scan pass 1, LDS-tiled: one workgroup stages a (8+2) x (256+8) window of ALL dog+2 det-H
// levels of one octave (each level-pixel fetched from HBM once, 16-byte loads).  Per detection level:
//   stage 1  every pixel of the tile runs the cheap part of the test (thresholds, 26 neighbours, edge
//            ratio) from LDS, 64 pixels per wavefront step; survivors are appended to an LDS queue;
//   stage 2  the queue is processed one candidate per lane, so the expensive sub-pixel solve
//            (9 IEEE divisions) runs with full lanes instead of 1-2 live lanes per wavefront;
//            accepted pixels set their bit in the tile's mask words (LDS atomicOr);
// then the mask words and per-row counts go to HBM.  Bits are positional, so the list order produced
// by the scatter pass is independent of the order candidates were queued in.
constexpr int EX_TR = 4, EX_TC = 128, EX_STRIDE = EX_TC + 8 + 4;  // cols x0-4 .. x0+260, +4 pad

__global__ __launch_bounds__(256) void extrema_mark_kernel(Geom g, DetectParams dp, const float* gauss, const float* deth,
                                                           uint64_t* rowmask, int* rowcnt, DetectStore ds) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [dog+2][EX_TR+2][EX_STRIDE]
  __shared__ int nfound;  // detections of this tile so far (all levels)
  __shared__ unsigned short cand[EX_TR * EX_TC];
  __shared__ int ncand;
  const int b = blockIdx.y;
  int o = 0;
  for (int k = 1; k < g.noct; k++)
    if (g.o[k].tile_base <= (int)blockIdx.x) o = k;
  const OctGeom& og = g.o[o];
  const int trel = blockIdx.x - og.tile_base;
  const int ty = trel / og.tiles_x, tx = trel - ty * og.tiles_x;
  const int x0 = tx * EX_TC, y0 = ty * EX_TR;
  const int nlv = g.dog + 2;
  constexpr int ROWS = EX_TR + 2, NG = (EX_TC + 8) / 4;
  // ---- stage 0: global -> LDS (zeros outside the plane; those cells are never used by a valid test).
  // Eight independent 16-byte loads are issued before the first LDS store so their HBM latencies overlap
  // (a load->store loop with a run-time trip count serialises one memory round trip per iteration).
  const int ngroups = nlv * ROWS * NG;
  for (int g0 = 0; g0 < ngroups; g0 += 8 * 256) {
    float4 v[8];
    int dst[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int gi = g0 + u * 256 + threadIdx.x;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      dst[u] = -1;
      if (gi < ngroups) {
        const int l = gi / (ROWS * NG);
        const int rem = gi - l * (ROWS * NG);
        const int r = rem / NG, gx = rem - r * NG;
        const int y = y0 - 1 + r, x = x0 - 4 + gx * 4;
        dst[u] = (l * ROWS + r) * EX_STRIDE + gx * 4;
        const bool ok = (y >= 0 && y < og.h && x >= 0 && x < og.wa);
        const int yc = ok ? y : 0, xc = ok ? x : 0;  // branch-free: always load, select afterwards
        const float4 q = *reinterpret_cast<const float4*>(deth + og.lvl_off + ((long long)l * g.B + b) * og.plane + (long long)yc * og.wa + xc);
        if (ok) v[u] = q;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (dst[u] >= 0) *reinterpret_cast<float4*>(&tile[dst[u]]) = v[u];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) nfound = 0;
  const long long fbase = (long long)b * ds.stride;
  const DetectSink sk{ds.found + fbase + (long long)blockIdx.x * DT_SLOTS, ds.found + fbase + (long long)ds.ntask * DT_SLOTS,
                      ds.spill_count + b, ds.cap_spill, reinterpret_cast<unsigned long long*>(rowmask) + (long long)b * g.NM,
                      rowcnt + (long long)b * g.NR, ds.hist ? ds.hist + (long long)b * kHistBins : nullptr};
  for (int l = 1; l <= g.dog; l++) {
    __syncthreads();  // the previous level's candidates have been processed
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();  // also orders stage 0 (first level)
    const float* C = tile + (l * ROWS) * EX_STRIDE;
    const float* P = C - ROWS * EX_STRIDE;
    const float* N = C + ROWS * EX_STRIDE;
    // ---- stage 1: wave w owns tile rows 2w, 2w+1.  Branch-free necessary condition with 27
    // independent LDS reads: a keypoint's response is beyond the first threshold and is >= all 26
    // neighbours or <= all of them (the reference's running nmax/nmin chain implies it).  The exact,
    // order-dependent test runs in stage 2 on the survivors only. ----
#pragma unroll 1
    for (int it = 0; it < (EX_TR / 4) * (EX_TC / 64); it++) {
      const int rl_ = wv * (EX_TR / 4) + it / (EX_TC / 64), blk = it % (EX_TC / 64);
      const int row = y0 + rl_, col = x0 + blk * 64 + lane;
      const int ci = (rl_ + 1) * EX_STRIDE + (blk * 64 + lane + 4);
      const float r = C[ci];
      float mx = -3.402823466e38f, mn = 3.402823466e38f;
#pragma unroll
      for (int dy = -1; dy <= 1; dy++) {
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
          const int k = ci + dy * EX_STRIDE + dx;
          const float a = P[k], c2 = N[k];
          mx = fmaxf(mx, fmaxf(a, c2));
          mn = fminf(mn, fminf(a, c2));
          if (dy != 0 || dx != 0) {
            const float q = C[k];
            mx = fmaxf(mx, q);
            mn = fminf(mn, q);
          }
        }
      }
      const bool flag = (row > 0 && row < og.h - 1 && col > 0 && col < og.wa - 1) && (fabsf(r) > dp.thr0) &&
                        ((r >= mx) || (r <= mn));
      const uint64_t m = __ballot(flag);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&ncand, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (flag) cand[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)((rl_ << 8) | (blk * 64 + lane));
      }
    }
    __syncthreads();
    // ---- stage 2: one candidate per lane; accepted ones go to the image's list, mask words and row counts ----
    const int nc = ncand;
    const float* G = gauss + og.lvl_off + ((long long)l * g.B + b) * og.plane;
    for (int i0 = 0; i0 < nc; i0 += 256) {  // (whole wavefronts go round: post_detections is a wavefront call)
      const int i = i0 + threadIdx.x;
      const int code = cand[i < nc ? i : 0];
      const int rl_ = code >> 8, cl = code & 255;
      const int row = y0 + rl_, col = x0 + cl;
      KeyVal kv;
      const bool ok = i < nc && key_eval<false>(C, P, N, G, EX_STRIDE, (rl_ + 1) * EX_STRIDE + (cl + 4), dp, &kv,
                                                (long long)row * og.wa + col);
      const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
      int before = 0;
      if (m) {  // (wavefront-uniform) the tile's four wavefronts count in LDS
        if (lane == __builtin_ctzll(m)) before = atomicAdd(&nfound, __popcll(m));
        before = __shfl(before, __builtin_ctzll(m));
      }
      post_detections(sk, before, ok, m, kv, o * g.dog + (l - 1), row, col,
                      og.mask_base + ((l - 1) * og.h + row) * og.w64 + (col >> 6), og.row_base + (l - 1) * og.h + row);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) ds.task_count[(long long)b * ds.ntask + blockIdx.x] = min(nfound, DT_SLOTS);
}

// ---- streaming extrema scan (pass 1, default for dog <= 5): registers instead of an LDS tile ----
// One wavefront marches down a strip of 124 owned columns (lane j holds columns x0-2+2j, x0-1+2j of
// every det-H level; lanes 0 and 63 only supply the halo column) over Geom::stream_rows rows (24; 12 for batches of one or two images).  (Columns per lane =
// kStreamCols, hess_dev.h: with one column per lane the kernel needs 116 instead of 168 registers and runs four
// instead of three wavefronts per SIMD, but spends 43 % more instructions per pixel on the neighbour exchange:
// same time)  Per new row and
// level it forms the horizontal 3-max / 3-min (neighbours from the adjacent lanes by DPP wave shifts)
// and keeps them for the last three rows, so the 26-neighbour maximum of a pixel is
//   max3( max over 3 rows of the level below, same of the level above,
//         max3(row above, row below, left/right) of its own level )
// -- about 65 VALU operations per pixel for all levels together, every level-pixel loaded from HBM
// once per strip segment, no LDS traffic and no barriers.  The necessary condition "beyond the first
// threshold and >= all 26 neighbours or <= all of them" is the same superset filter as in
// extrema_mark_kernel; survivors are queued (64 per batch, all lanes busy) for the exact
// order-dependent test key_eval, which sets positional mask bits and row counts with atomics (the
// masks are zeroed before the launch).
constexpr int SX_PITCH = kStreamPitch, SX_QCAP = 128;  // (rows per segment: Geom::stream_rows)

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }

template <int DOG>
__global__ __launch_bounds__(256) void extrema_stream_kernel(Geom g, DetectParams dp, const float* gauss, const float* deth,
                                                             unsigned long long* rowmask, int* rowcnt, DetectStore ds) {
  constexpr int NLV = DOG + 2;
  constexpr int NC = kStreamCols;  // columns per lane
  __shared__ uint32_t queue[4][SX_QCAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int blk = (int)blockIdx.x;
  int o = 0;
  for (int k = 1; k < g.noct; k++)
    if (g.o[k].stream_base <= blk) o = k;
  const OctGeom& og = g.o[o];
  const int task = (blk - og.stream_base) * 4 + wv;
  const int seg = task / og.strips, strip = task - seg * og.strips;
  const int ys = seg * g.stream_rows;
  int* const my_count = ds.task_count + (long long)b * ds.ntask + blk * 4 + wv;  // every task posts its count, also an idle one
  if (ys >= og.h) {  // wavefront-uniform; the kernel has no workgroup barrier
    if (lane == 0) *my_count = 0;
    return;
  }
  const int ye = min(ys + g.stream_rows, og.h);
  const int cx = strip * SX_PITCH - NC + NC * lane;  // this lane's first column
  const bool col_in = cx >= 0 && cx < og.wa;         // (wa is a multiple of 4: a lane's columns are all in or all out)
  const int wa = og.wa, h = og.h;
  const long long lstep = (long long)g.B * og.plane;
  const float* base = deth + og.lvl_off + (long long)b * og.plane;  // level l at base + l*lstep
  // a pixel is tested if it is interior (ProgramCU.cu:700-705) and owned by this lane
  const bool own = lane >= 1 && lane <= 62;
  bool cv[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) cv[c] = own && cx + c > 0 && cx + c < wa - 1;
  uint32_t* q = queue[wv];
  int qn = 0;

  struct Row { float v[NC]; };
  auto load_row = [&](int yy, Row (&dst)[NLV]) {
    // rows/columns outside the plane are only ever neighbours of pixels that are not tested: any
    // finite value will do, so the address is clamped instead of the value being selected
    const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
    const long long off = (long long)yc * wa + (col_in ? cx : 0);
#pragma unroll
    for (int l = 0; l < NLV; l++) {
      if (NC == 2) {
        const float2 t = *reinterpret_cast<const float2*>(base + l * lstep + off);
        dst[l].v[0] = t.x; dst[l].v[NC - 1] = t.y;
      } else {
#pragma unroll
        for (int c = 0; c < NC; c++) dst[l].v[c] = base[l * lstep + off + c];
      }
    }
  };
  const long long fbase = (long long)b * ds.stride;
  const DetectSink sk{ds.found + fbase + (long long)(blk * 4 + wv) * DT_SLOTS, ds.found + fbase + (long long)ds.ntask * DT_SLOTS,
                      ds.spill_count + b, ds.cap_spill, rowmask + (long long)b * g.NM, rowcnt + (long long)b * g.NR,
                      ds.hist ? ds.hist + (long long)b * kHistBins : nullptr};
  const float* gbase = gauss + og.lvl_off + (long long)b * og.plane;
  int nfound = 0;  // detections of this task so far (wavefront-uniform)
  auto process = [&](uint32_t e, bool active) {  // (called by the whole wavefront)
    const int l = e >> 28, row = (e >> 14) & 0x3FFF, col = e & 0x3FFF;
    const float* C = base + l * lstep;
    KeyVal kv;
    const bool ok = active && key_eval<false>(C, C - lstep, C + lstep, gbase + l * lstep, wa, row * wa + col, dp, &kv);
    const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
    post_detections(sk, nfound, ok, m, kv, o * g.dog + (l - 1), row, col, og.mask_base + ((l - 1) * h + row) * og.w64 + (col >> 6),
                    og.row_base + (l - 1) * h + row);
    nfound += __popcll(m);
  };

  // ring of the last three rows: horizontal 3-max/3-min of every level (centre included) and the raw centre values of
  // the detection levels.  The 27-point maximum INCLUDING the pixel itself is all the test needs: r >= max(26
  // neighbours) <=> r >= max(27 values), the pixel being one of them -- so no left/right or above/below maxima that leave
  // the centre out (rounds 1-3 kept those: 24 more instructions per row and 36 more registers).
  float hmx[NLV][3][NC], hmn[NLV][3][NC];
  float rc[DOG][3][NC];
  auto ingest = [&](const Row (&cur)[NLV], int slot) {
#pragma unroll
    for (int l = 0; l < NLV; l++) {
      const float L = lane_prev(cur[l].v[NC - 1]), R = lane_next(cur[l].v[0]);
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const float left = c == 0 ? L : cur[l].v[c > 0 ? c - 1 : 0];
        const float right = c == NC - 1 ? R : cur[l].v[c < NC - 1 ? c + 1 : c];
        const float a = cur[l].v[c];
        hmx[l][slot][c] = max3f(left, a, right);
        hmn[l][slot][c] = min3f(left, a, right);
        if (l >= 1 && l <= DOG) rc[l - 1][slot][c] = a;
      }
    }
  };

  Row cur[NLV], nxt[NLV];
  load_row(ys - 1, cur);
  load_row(ys, nxt);
  ingest(cur, 0);
#pragma unroll
  for (int l = 0; l < NLV; l++) cur[l] = nxt[l];
  load_row(ys + 1, nxt);
  ingest(cur, 1);

  for (int y0 = ys; y0 < ye; y0 += 3) {
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const int y = y0 + s;
      if (y < ye) {  // wavefront-uniform
        // ring slots: row y-1 -> s, row y -> s+1, row y+1 (arriving now) -> s+2 (mod 3)
        constexpr int kRing[5] = {0, 1, 2, 0, 1};
        const int sa = kRing[s], sc = kRing[s + 1], sb = kRing[s + 2];
#pragma unroll
        for (int l = 0; l < NLV; l++) cur[l] = nxt[l];
        if (y + 2 <= ye) load_row(y + 2, nxt);  // prefetch: consumed in the next iteration (row ye is the last one any owned row needs)
        ingest(cur, sb);
        if (y > 0 && y < h - 1) {  // wavefront-uniform
          float m9x[NLV][NC], m9n[NLV][NC];
#pragma unroll
          for (int l = 0; l < NLV; l++)
#pragma unroll
            for (int c = 0; c < NC; c++) {
              m9x[l][c] = max3f(hmx[l][sa][c], hmx[l][sc][c], hmx[l][sb][c]);
              m9n[l][c] = min3f(hmn[l][sa][c], hmn[l][sc][c], hmn[l][sb][c]);
            }
          uint32_t cand = 0;
#pragma unroll
          for (int li = 0; li < DOG; li++)
#pragma unroll
            for (int c = 0; c < NC; c++) {
              const int l = li + 1;
              const float r = rc[li][sc][c];
              const float nx = max3f(m9x[l - 1][c], m9x[l][c], m9x[l + 1][c]);  // over all 27, r itself among them
              const float nn = min3f(m9n[l - 1][c], m9n[l][c], m9n[l + 1][c]);
              const bool f = cv[c] & (fabsf(r) > dp.thr0) & ((r >= nx) | (r <= nn));
              cand |= f ? (1u << (li * NC + c)) : 0u;
            }
          // queue the candidates: per trip every lane that still has one appends its lowest (a lane rarely has
          // more than one per row, so this is one trip, not one per level and column; the order inside the queue
          // does not matter -- accepted pixels set positional mask bits)
          for (uint64_t m = __builtin_amdgcn_ballot_w64(cand != 0); m != 0; m = __builtin_amdgcn_ballot_w64(cand != 0)) {
            // (cross-lane exchange through LDS inside one wavefront: DS operations of a wavefront execute in order; the
            // wave barriers only pin the compiler's ordering of the may-alias accesses, they emit no instruction)
            if (cand != 0) {
              const int k = __builtin_ctz(cand);
              q[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                  ((uint32_t)(k / NC + 1) << 28) | ((uint32_t)y << 14) | (uint32_t)(cx + k % NC);
              cand &= cand - 1;
            }
            __builtin_amdgcn_wave_barrier();
            qn += __popcll(m);
            if (qn >= 64) {
              const uint32_t head = q[lane], tail = q[64 + lane];
              __builtin_amdgcn_wave_barrier();
              process(head, true);
              qn -= 64;
              if (lane < qn) q[lane] = tail;
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (qn > 0) process(q[lane < qn ? lane : 0], lane < qn);
  if (lane == 0) *my_count = min(nfound, DT_SLOTS);
}

__device__ int apply_level_limits(int* lc, int nlev, const LimitParams& lp, bool generation_stage);

// =============================== block scan helpers ==========================================

// Exclusive scan of (a, b) over a 1024-thread workgroup; returns totals through ta/tb.
__device__ __forceinline__ void block_scan2(int a, int b, int* ea, int* eb, int* ta, int* tb, int* lds /*64 ints*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int na = __shfl_up(ia, d), nb = __shfl_up(ib, d);
    if (lane >= d) { ia += na; ib += nb; }
  }
  __syncthreads();
  if (lane == 63) { lds[wv] = ia; lds[16 + wv] = ib; }
  __syncthreads();
  int oa = 0, ob = 0, sa = 0, sb = 0;
  for (int k = 0; k < 16; k++) {
    if (k < wv) { oa += lds[k]; ob += lds[16 + k]; }
    sa += lds[k]; sb += lds[16 + k];
  }
  *ea = oa + ia - a; *eb = ob + ib - b; *ta = sa; *tb = sb;
}

// Level truncation of GenerateFeatureList (-tc2/-tc3 stop adding levels, PyramidCU.cpp:1311-1319)
// and LimitFeatureCount (SiftPyramid.cpp:201-278).  lc[] is updated in place; returns the total.
__device__ int apply_level_limits(int* lc, int nlev, const LimitParams& lp, bool generation_stage) {
  int total = 0;
  const int thr = lp.threshold;
  if (generation_stage) {
    const bool reverse = (lp.method == 1);
    for (int k = 0; k < nlev; k++) {
      int li = reverse ? nlev - 1 - k : k;
      if ((lp.method == 1 || lp.method == 2) && thr > 0 && total > thr) { lc[li] = 0; continue; }
      total += lc[li];
    }
  } else {
    for (int k = 0; k < nlev; k++) total += lc[k];
  }
  if (thr > 0 && lp.method != 3) {
    if (lp.method == 2) {
      int i = 0, nf = 0;
      for (; (nf < thr) && (i < nlev); ++i) nf += lc[i];
      for (; i < nlev; ++i) lc[i] = 0;
      if (nf < total) total = nf;
    } else {
      int i = 0;
      while (i < nlev && (total - lc[i]) > thr) { total -= lc[i]; lc[i++] = 0; }
    }
  }
  return total;
}

// Row counts of image b -> exclusive offsets in list order (level, row), level totals, -tc level truncation
// (GenerateFeatureList / LimitFeatureCount, PyramidCU.cpp:1283-1368, SiftPyramid.cpp:201-278), by ONE workgroup of 1024
// threads (every thread of it calls).
__device__ void row_scan_block(const Geom& g, const LimitParams& lp, const int* rowcnt, int* rowoff, int* level_count,
                               int* raw_total, int cap_raw, int* overflow, int b) {
  __shared__ int lc[kMaxOct * kMaxDog];
  __shared__ int keep[kMaxOct * kMaxDog];
  __shared__ int lds[64];
  const int tid = threadIdx.x;
  const int* cnt = rowcnt + (long long)b * g.NR;
  int* off = rowoff + (long long)b * g.NR;
  for (int i = tid; i < g.nlev; i += 1024) lc[i] = 0;
  __syncthreads();
  // A thread owns `per` consecutive rows of the list order: level totals first, then (after the -tc rules have
  // decided which levels stay) one workgroup scan over the threads' sums and a serial walk over the own rows.
  const int per = (g.NR + 1023) >> 10;
  const int r0 = tid * per, r1 = min(g.NR, r0 + per);
  // Up to RC rows per thread (32 768 rows per image: a 4096^2 image has 24 552) are read once, with independent loads,
  // and kept in registers; the level of a row follows from walking the level boundaries, not from a division per row.
  constexpr int RC = 32;
  __shared__ int oct_h[kMaxOct], oct_base[kMaxOct];
  for (int i = tid; i < g.noct; i += 1024) { oct_h[i] = g.o[i].h; oct_base[i] = g.o[i].row_base; }
  __syncthreads();
  if (per <= RC) {  // uniform
    int cc[RC];
#pragma unroll
    for (int u = 0; u < RC; u++) cc[u] = (u < per && r0 + u < r1) ? cnt[r0 + u] : 0;
    int o = 0;
    for (int k = 1; k < g.noct; k++) if (oct_base[k] <= r0) o = k;
    const int rel = r0 - oct_base[o];
    const int lm0 = rel / oct_h[o];
    const int li0 = o * g.dog + lm0, left0 = oct_h[o] - (rel - lm0 * oct_h[o]);  // rows left in the level, this one included
    int lis[RC];  // level of every own row (levels are consecutive in list order)
    {
      int li = li0, lm = lm0, left = left0, oo = o;
#pragma unroll
      for (int u = 0; u < RC; u++) {
        lis[u] = li;
        if (--left == 0) {
          li++;
          if (++lm == g.dog) { lm = 0; oo++; }
          left = oo < g.noct ? oct_h[oo] : 0x7fffffff;
        }
      }
    }
    {
      int run_level = -1, run = 0;
#pragma unroll
      for (int u = 0; u < RC; u++) {
        if (cc[u]) {
          if (lis[u] != run_level) {
            if (run) atomicAdd(&lc[run_level], run);
            run_level = lis[u]; run = 0;
          }
          run += cc[u];
        }
      }
      if (run) atomicAdd(&lc[run_level], run);
    }
    __syncthreads();
    if (tid == 0) {
      int before[kMaxOct * kMaxDog];
      for (int i = 0; i < g.nlev; i++) before[i] = lc[i];
      int total = apply_level_limits(lc, g.nlev, lp, true);
      for (int i = 0; i < g.nlev; i++) {
        keep[i] = (lc[i] == before[i]);  // a level is either kept whole or dropped
        level_count[b * g.nlev + i] = lc[i];
      }
      raw_total[b] = total < cap_raw ? total : cap_raw;
      if (total > cap_raw) atomicMax(overflow, total);
    }
    __syncthreads();
    // ordered exclusive scan of the kept rows (dropped rows count 0: offsets stay monotone)
    int mine = 0;
#pragma unroll
    for (int u = 0; u < RC; u++) {
      cc[u] = (cc[u] && keep[min(lis[u], g.nlev - 1)]) ? cc[u] : 0;
      mine += cc[u];
    }
    int e, e2, tot, tot2;
    block_scan2(mine, 0, &e, &e2, &tot, &tot2, lds);
#pragma unroll
    for (int u = 0; u < RC; u++) {
      if (u < per && r0 + u < r1) off[r0 + u] = e;
      e += cc[u];
    }
    return;
  }
  // more than RC rows per thread: the same steps with the counts re-read from memory
  auto level_of_row = [&](int i) {
    int o = 0;
    for (int k = 1; k < g.noct; k++) if (oct_base[k] <= i) o = k;
    return o * g.dog + (i - oct_base[o]) / oct_h[o];
  };
  {
    int run_level = -1, run = 0;
    for (int i = r0; i < r1; i++) {
      const int c = cnt[i];
      if (!c) continue;
      const int li = level_of_row(i);
      if (li != run_level) {
        if (run) atomicAdd(&lc[run_level], run);
        run_level = li; run = 0;
      }
      run += c;
    }
    if (run) atomicAdd(&lc[run_level], run);
  }
  __syncthreads();
  if (tid == 0) {
    int before[kMaxOct * kMaxDog];
    for (int i = 0; i < g.nlev; i++) before[i] = lc[i];
    int total = apply_level_limits(lc, g.nlev, lp, true);
    for (int i = 0; i < g.nlev; i++) {
      keep[i] = (lc[i] == before[i]);  // a level is either kept whole or dropped
      level_count[b * g.nlev + i] = lc[i];
    }
    raw_total[b] = total < cap_raw ? total : cap_raw;
    if (total > cap_raw) atomicMax(overflow, total);
  }
  __syncthreads();
  int mine = 0;
  for (int i = r0; i < r1; i++) {
    const int c = cnt[i];
    if (c && keep[level_of_row(i)]) mine += c;
  }
  int e, e2, tot, tot2;
  block_scan2(mine, 0, &e, &e2, &tot, &tot2, lds);
  for (int i = r0; i < r1; i++) {
    off[i] = e;
    const int c = cnt[i];
    if (c && keep[level_of_row(i)]) e += c;
  }
}

// List order.  The unordered detections of image b -- the tasks' slots (task_count[t] of DT_SLOTS in use) and the spill
// list -- are copied to their places in the raw list: position = exclusive offset of the detection's row in list order +
// number of detections of that row to its left (mask bits below its column).  Workgroups take chunks of PL_CHUNK (2048) store
// positions in ARRIVAL order (ticket); the first to arrive scans the image's row counts (row_scan_block) and raises the
// image's flag, the others load their records and count their mask bits meanwhile and then wait for the flag -- a
// bounded, sleeping wait on a workgroup that is running by construction (it drew its ticket first); a flag that never
// comes ends as a device-side error word the host reports (overflow[2], as topk_select_kernel), not as a hung stream.
// ticket / flag arrive zeroed.
constexpr int PL_SPIN_LIMIT = 1 << 21;
#ifndef HESS_PL_PER
#define HESS_PL_PER 2
#endif
// (1 / 2 / 4 / 8 positions per thread, same call: launch 72.8 / 62.7 / 63.6 / 87.8 us, pipelined line 22.37 / 22.57 / 22.27 / 21.72
// Gpix/s -- with one, the waiting workgroups of eight images fill every wavefront slot of the chip while the first ones scan)
constexpr int PL_PER = HESS_PL_PER;        // store positions per thread
constexpr int PL_CHUNK = 1024 * PL_PER;  // ... per workgroup

__global__ __launch_bounds__(1024) void extrema_place_kernel(Geom g, LimitParams lp, DetectStore ds, const uint64_t* rowmask,
                                                             const int* rowcnt, int* rowoff, int* level_count, int* raw_total,
                                                             int* overflow, int* ticket, int* flag, RawKey* raw, int cap_raw) {
  __shared__ int s_ck;
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid == 0) s_ck = atomicAdd(&ticket[b], 1);
  __syncthreads();
  const int ck = s_ck;
  const int nslot = ds.ntask * DT_SLOTS;
  const int total = nslot + min(ds.spill_count[b], ds.cap_spill);
  if (ck != 0 && ck * PL_CHUNK >= total) return;  // (workgroup-uniform) nothing in this chunk; nobody waits for it
  if (ck == 0) {
    row_scan_block(g, lp, rowcnt, rowoff, level_count, raw_total, cap_raw, overflow, b);
    __threadfence();  // the offsets and level totals, before the flag
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flag[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  // A store position's record and the number of detections of its row to its left (mask bits below its column):
  // independent of the scan, so the first trip's are in registers before the flag is awaited.
  struct Pending { uint4 ra, rb; int li, ri, rank; bool valid; };
  auto fetch = [&](int i) {
    Pending p;
    p.valid = i < total && !(i < nslot && (i & (DT_SLOTS - 1)) >= ds.task_count[(long long)b * ds.ntask + i / DT_SLOTS]);
    p.ra = p.rb = make_uint4(0u, 0u, 0u, 0u);
    p.li = p.ri = p.rank = 0;
    if (!p.valid) return p;
    const uint4* const src = reinterpret_cast<const uint4*>(ds.found + (long long)b * ds.stride + i);
    p.ra = src[0]; p.rb = src[1];
    p.li = (int)p.ra.x;
    const int col = (int)p.ra.y, row = (int)p.ra.z;
    const int oct = p.li / g.dog, lm1 = p.li - oct * g.dog;
    const OctGeom& og = g.o[oct];
    p.ri = og.row_base + lm1 * og.h + row;
    const uint64_t* const mrow = rowmask + (long long)b * g.NM + og.mask_base + (lm1 * og.h + row) * og.w64;
    const int wcol = col >> 6;
    p.rank = __popcll(mrow[wcol] & ((1ull << (col & 63)) - 1ull));
    for (int w0 = 0; w0 < wcol; w0 += 8) {  // eight loads in flight per trip
      uint64_t mw[8];
#pragma unroll
      for (int k = 0; k < 8; k++) mw[k] = (w0 + k < wcol) ? mrow[w0 + k] : 0ull;
#pragma unroll
      for (int k = 0; k < 8; k++) p.rank += __popcll(mw[k]);
    }
    return p;
  };
  auto place = [&](const Pending& p) {
    if (!p.valid) return;
    // (written by another workgroup of this launch, possibly on another XCD: loads that go to the coherent level)
    if (__hip_atomic_load(&level_count[b * g.nlev + p.li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;  // level dropped by a -tc rule
    const int pos = __hip_atomic_load(&rowoff[(long long)b * g.NR + p.ri], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + p.rank;
    if (pos < cap_raw) {
      uint4* const dst = reinterpret_cast<uint4*>(raw + (long long)b * cap_raw + pos);
      dst[0] = p.ra;
      dst[1] = p.rb;
    }
  };
  Pending first[PL_PER];
#pragma unroll
  for (int u = 0; u < PL_PER; u++) first[u] = fetch(ck * PL_CHUNK + u * 1024 + tid);
  if (ck != 0) {
    // ONE thread of the workgroup polls, with plain coherent loads (an ACQUIRE load per poll invalidates the cache the
    // scanning workgroup is working from, and sixteen polling wavefronts per workgroup crowd the path the scan's own
    // loads take: 0.29 / 0.068 ms per step in the first forms of this kernel); what is read after the flag is read with
    // coherent loads too
    __shared__ int s_ok;
    if (tid == 0) {
      int v = __hip_atomic_load(&flag[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int spin = 0; !v && spin < PL_SPIN_LIMIT; spin++) {
        __builtin_amdgcn_s_sleep(32);
        v = __hip_atomic_load(&flag[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (!v) atomicMax(overflow + 2, 1);
      s_ok = v;
    }
    __syncthreads();
    if (!s_ok) return;
  }
#pragma unroll
  for (int u = 0; u < PL_PER; u++) place(first[u]);
  for (int i0 = (ck + (int)gridDim.x) * PL_CHUNK; i0 < total; i0 += gridDim.x * PL_CHUNK)  // (a spill list that outgrows the grid: rare)
    for (int u = 0; u < PL_PER; u++) place(fetch(i0 + u * 1024 + tid));
}

// =============================== top-K =======================================================

// Top-K in ONE wide launch (rounds 1-2: one 1024-thread workgroup per image did the whole selection -- 33 us on the
// critical path of a single 1080p image, 0.24 ms for a 4096^2 image with 2.6e5 detections; round 3: a counting and a
// copying launch).  The list is cut into chunks of TK_CHUNK entries, one workgroup each, chunk numbers handed out by a
// ticket counter in ARRIVAL order (so every workgroup with a lower chunk number is already running: the look-back
// below cannot wait for a workgroup that has not started, whatever the order the hardware dispatches them in):
//   1. cut bin and number of tied entries to keep, from the 15-bit histogram (every workgroup for itself: 128 KB, L2);
//   2. the chunk's entries are classified from their keys alone; (sure keeps, ties) of the chunk are PUBLISHED as one
//      64-bit word before the workgroup waits for anything;
//   3. look-back: the words of the chunks before this one are awaited and summed (a predecessor publishes after step
//      2, which depends on nothing: bounded wait);
//   4. one workgroup scan gives every thread its tie rank and output position -- of the first T ties min(T, need) are
//      kept -- and the kept entries are copied in list order; the last chunk in use posts the kept total.
// Same result as before: the K largest abs(half(response)), ties at the cut to the lower list index, list order kept.
// ticket / state / sel_level_count arrive zeroed (they live in the batch's cleared block, hess_plan.hip).
constexpr int TK_PER = 4, TK_CHUNK = 1024 * TK_PER;


// cut bin and number of ties to keep, by all 1024 threads of a workgroup; n >= K.  Result in s_cut / s_need (LDS).
__device__ __forceinline__ void topk_find_cut(const unsigned* h, int K, int* lds, int* s_cut, int* s_need) {
  const int tid = threadIdx.x;
  // thread t owns bins [32t, 32t+32); threads are scanned from the high end: thread r = 1023 - tid loads the bins of scan
  // position tid, so that the thread which finds the cut inside its position walks values it already holds
  const int owner = 1023 - tid;
  int vals[32];
  {
    const uint4* hv = reinterpret_cast<const uint4*>(h + owner * 32);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint4 v = hv[q];
      vals[4 * q] = (int)v.x; vals[4 * q + 1] = (int)v.y; vals[4 * q + 2] = (int)v.z; vals[4 * q + 3] = (int)v.w;
    }
  }
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 32; k++) mine += vals[k];
  int e, e2, tot, tot2;  // suffix sum over the bins = exclusive prefix over the reversed order
  block_scan2(mine, 0, &e, &e2, &tot, &tot2, lds);
  const int above = e, incl = e + mine;  // e = count in strictly higher bins
  if (above < K && incl >= K) {
    int acc = above;
    bool found = false;
#pragma unroll
    for (int k = 31; k >= 0; k--) {
      const int c = vals[k];
      if (!found && acc + c >= K) { *s_cut = owner * 32 + k; *s_need = K - acc; found = true; }
      acc += c;
    }
  }
  __syncthreads();
}

// keys of this thread's TK_PER consecutive entries -> bit masks (sure keep, tie at the cut)
__device__ __forceinline__ void topk_classify(const RawKey* in, int i0, int n, int cut, uint32_t* surem, uint32_t* tiem) {
  uint32_t pk[TK_PER];
#pragma unroll
  for (int u = 0; u < TK_PER; u++) pk[u] = in[max(min(i0 + u, n - 1), 0)].packed;  // (an empty list reads entry 0: never used)
  uint32_t sm = 0, tm = 0;
#pragma unroll
  for (int u = 0; u < TK_PER; u++) {
    if (i0 + u < n) {
      const int key = (int)((pk[u] >> 16) & 0x7fffu);
      if (cut < 0 || key > cut) sm |= 1u << u;
      else if (key == cut) tm |= 1u << u;
    }
  }
  *surem = sm; *tiem = tm;
}

constexpr unsigned long long TK_FLAG = 1ull << 40;
constexpr int TK_SPIN_LIMIT = 1 << 21;  // polls of a predecessor's word, about half a microsecond apart: a second

__global__ __launch_bounds__(1024) void topk_select_kernel(Geom g, int K, const RawKey* raw, const int* raw_total, int cap_raw,
                                                           const unsigned* hist, int* ticket, unsigned long long* state,
                                                           RawKey* sel, int* sel_total, int* sel_level_count, int cap_sel,
                                                           int nchunk, int* overflow) {
  __shared__ int lds[64];
  __shared__ int lc[kMaxOct * kMaxDog];
  __shared__ int s_cut, s_need, s_ck, s_sure0, s_ties0;
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid == 0) { s_ck = atomicAdd(&ticket[b], 1); s_cut = -1; s_need = 0; }
  __syncthreads();
  const int ck = s_ck;
  const int n = raw_total[b];
  const int used = (n + TK_CHUNK - 1) / TK_CHUNK;
  if (ck >= used && ck != 0) return;  // (workgroup-uniform) nothing in this chunk; nobody waits for it
  const RawKey* in = raw + (long long)b * cap_raw;
  RawKey* out = sel + (long long)b * cap_sel;
  for (int i = tid; i < g.nlev; i += 1024) lc[i] = 0;
  if (n >= K) topk_find_cut(hist + (long long)b * kHistBins, K, lds, &s_cut, &s_need);  // SelectTopK is skipped below K detections
  __syncthreads();
  const int cut = s_cut, need = s_need;
  const int i0 = ck * TK_CHUNK + tid * TK_PER;
  uint32_t surem, tiem;
  topk_classify(in, i0, n, cut, &surem, &tiem);
  int etie, esure, ttie, tsure;
  block_scan2(__popc(tiem), __popc(surem), &etie, &esure, &ttie, &tsure, lds);
  unsigned long long* st = state + (long long)b * nchunk;
  if (tid == 0)
    __hip_atomic_store(&st[ck], TK_FLAG | ((unsigned long long)tsure << 20) | (unsigned long long)ttie, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
  // look-back over the chunks before this one
  int ps = 0, pt = 0;
  // (a predecessor publishes before it waits for anything, and ticket order means it is already running: the wait is
  // short.  It is bounded all the same -- a word that never gets its flag, e.g. scratch that was not cleared, must end
  // as an error the host reports (word 2 of the overflow block -> HESS_ERR_DEVICE, as the reference returns 0 on device
  // errors, SiftPyramid.h:162-163), not as a hung stream; the poll sleeps between loads so that 1 024 spinning lanes
  // leave the memory path to the workgroups they wait for.)
  for (int k = tid; k < ck; k += 1024) {
    unsigned long long v = __hip_atomic_load(&st[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; !(v & TK_FLAG) && spin < TK_SPIN_LIMIT; spin++) {
      __builtin_amdgcn_s_sleep(8);
      v = __hip_atomic_load(&st[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!(v & TK_FLAG)) { atomicMax(overflow + 2, 1); v = 0; }
    ps += (int)((v >> 20) & 0xFFFFFu);
    pt += (int)(v & 0xFFFFFu);
  }
  {
    int e0, e1, t0, t1;
    __syncthreads();
    block_scan2(ps, pt, &e0, &e1, &t0, &t1, lds);
    if (tid == 0) { s_sure0 = t0; s_ties0 = t1; }
    __syncthreads();
  }
  const int ties0 = s_ties0, kept0 = s_sure0 + min(ties0, need);
  if (tid == 0 && (ck == used - 1 || used == 0)) {  // the last chunk in use knows the kept total
    const int kept = s_sure0 + tsure + min(ties0 + ttie, need);
    sel_total[b] = kept < cap_sel ? kept : cap_sel;
  }
  int tseen = ties0 + etie;                                  // ties before this thread's entries
  int pos = kept0 + esure + (min(tseen, need) - min(ties0, need));
  int run_level = -1, run = 0;  // kept entries per level: one LDS atomic per run of equal levels, not per entry
  // the kept entries are loaded with independent loads before any of them is stored (one memory round trip)
#define HESS_TK_LOAD(U)                                                                                    \
  const bool tie##U = (tiem >> U) & 1u;                                                                    \
  const bool kp##U = ((surem >> U) & 1u) || (tie##U && tseen < need);                                      \
  tseen += tie##U ? 1 : 0;                                                                                 \
  const uint4* src##U = reinterpret_cast<const uint4*>(in + (kp##U ? i0 + U : 0)); /* always load */       \
  const uint4 ra##U = src##U[0], rb##U = src##U[1]; /* a RawKey is two 16-byte pieces */
#define HESS_TK_STORE(U)                                                     \
  if (kp##U) {                                                               \
    if (pos < cap_sel) {                                                     \
      uint4* dst = reinterpret_cast<uint4*>(out + pos);                      \
      dst[0] = ra##U;                                                        \
      dst[1] = rb##U;                                                        \
    }                                                                        \
    const int lvl = (int)ra##U.x; /* RawKey::level_index */                  \
    if (lvl != run_level) {                                                  \
      if (run) atomicAdd(&lc[run_level], run);                               \
      run_level = lvl; run = 0;                                              \
    }                                                                        \
    run++;                                                                   \
    pos++;                                                                   \
  }
  static_assert(TK_PER == 4 && sizeof(RawKey) == 32 && offsetof(RawKey, level_index) == 0, "RawKey as two uint4, four per thread");
  static_assert(TK_CHUNK < (1 << 20), "chunk counts fit the published word");
  HESS_TK_LOAD(0) HESS_TK_LOAD(1) HESS_TK_LOAD(2) HESS_TK_LOAD(3)
  HESS_TK_STORE(0) HESS_TK_STORE(1) HESS_TK_STORE(2) HESS_TK_STORE(3)
#undef HESS_TK_LOAD
#undef HESS_TK_STORE
  if (run) atomicAdd(&lc[run_level], run);
  __syncthreads();
  for (int i = tid; i < g.nlev; i += 1024)
    if (lc[i]) atomicAdd(&sel_level_count[b * g.nlev + i], lc[i]);
}

// =============================== math probe (parity tests) ===================================
__global__ void math_probe_kernel(int which, const float* a, const float* b, float* out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = 0.0f;
  switch (which) {
    case 0: r = dm_expf(a[i]); break;
    case 1: r = dm_atan2f(a[i], b[i]); break;
    case 2: { float s, c; dm_sincosf(a[i], &s, &c); r = s; break; }
    case 3: { float s, c; dm_sincosf(a[i], &s, &c); r = c; break; }
    case 4: r = (float)dm_f2h(a[i]); break;
    case 5: r = dm_h2f((uint32_t)a[i]); break;
    case 6: r = a[i] / b[i]; break;
    case 7: r = sqrtf(a[i]); break;
    case 8: r = dm_u8_unit(a[i]); break;
    default: break;
  }
  out[i] = r;
}

}  // namespace

void launch_hessian(hipStream_t st, const Geom& g, int octave, const float* gauss, float* deth, float* got,
                    const float* norms, int batch, int level_first, int level_last) {
  const OctGeom& og = g.o[octave];
  HessArgs a;
  a.gauss = gauss; a.deth = deth; a.got = reinterpret_cast<float2*>(got);
  a.wa = og.wa; a.h = og.h; a.plane = og.plane; a.B = g.B; a.dog = g.dog; a.level_first = level_first;
  a.batch = batch; a.lvl_off = og.lvl_off; a.got_off = og.got_off;
  a.inv_groups = 1.0f / (float)(og.wa >> 2);
  for (int l = 0; l < g.dog + 2; l++) a.norm[l] = norms[l];
  const int groups = (og.wa >> 2) * og.h;
  hipLaunchKernelGGL(hessian_kernel, dim3((groups + 255) / 256, (level_last - level_first + 1) * batch), dim3(256),
                     0, st, a);
}

void launch_hessian_level(hipStream_t st, const Geom& g, const float* gauss, float* deth, int level, float norm,
                          int batch, void* zero, size_t zero_bytes) {
  int blocks = 0;
  for (int o = 0; o < g.noct; o++) blocks += ((g.o[o].wa >> 2) * ((g.o[o].h + 3) >> 2) + 255) >> 8;
  hipLaunchKernelGGL(hessian_rows4_kernel, dim3(blocks, batch), dim3(256), 0, st, g, gauss, deth, level, norm,
                     reinterpret_cast<uint4*>(zero), (long long)(zero_bytes / 16));
}

bool extrema_streams(const Geom& g) { return g.dog <= 5 && g.o[0].wa < (1 << 14) && g.o[0].h < (1 << 14); }

int extrema_tasks(const Geom& g) { return extrema_streams(g) ? g.nstream * 4 : g.ntiles; }

void launch_extrema_mark(hipStream_t st, const Geom& g, const DetectParams& dp, const float* gauss,
                         const float* deth, uint64_t* rowmask, int* rowcnt, const DetectStore& ds, int batch) {
  // rowcnt, rowmask, ds.spill_count and ds.hist arrive zeroed (one fill per batch in enqueue(), hess_schedule.hip)
  // streaming scan; its candidate queue packs row and column in 14 bits each
  if (extrema_streams(g)) {
    unsigned long long* rm = reinterpret_cast<unsigned long long*>(rowmask);
    const dim3 grid(g.nstream, batch), blk(256);
#define HESS_STREAM_LAUNCH(D) hipLaunchKernelGGL(extrema_stream_kernel<D>, grid, blk, 0, st, g, dp, gauss, deth, rm, rowcnt, ds)
    switch (g.dog) {
      case 1: HESS_STREAM_LAUNCH(1); break;
      case 2: HESS_STREAM_LAUNCH(2); break;
      case 3: HESS_STREAM_LAUNCH(3); break;
      case 4: HESS_STREAM_LAUNCH(4); break;
      default: HESS_STREAM_LAUNCH(5); break;
    }
#undef HESS_STREAM_LAUNCH
    return;
  }
  // more than 5 detection levels per octave, or planes of 16384 px and more: LDS-tiled scan (level count
  // and plane size are run-time values there)
  const size_t lds = (size_t)(g.dog + 2) * (EX_TR + 2) * EX_STRIDE * sizeof(float);
  static size_t lds_allowed = 0;  // dog >= 4 needs more than the default 64 KB of dynamic LDS
  if (lds > lds_allowed) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(extrema_mark_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_allowed = lds;
  }
  hipLaunchKernelGGL(extrema_mark_kernel, dim3(g.ntiles, batch), dim3(256), lds, st, g, dp, gauss, deth, rowmask, rowcnt, ds);
}

void launch_extrema_place(hipStream_t st, const Geom& g, const LimitParams& lp, const DetectStore& ds, const uint64_t* rowmask,
                          const int* rowcnt, int* rowoff, int* level_count, int* raw_total, int* overflow, int* ticket,
                          int* flag, RawKey* raw, int cap_raw, int batch) {
  // the tasks' slots + a spill list of up to 16 chunks at a time (longer spill lists take more trips: they are rare)
  const int chunks = (ds.ntask * DT_SLOTS + std::min(ds.cap_spill, 16 * 1024) + PL_CHUNK - 1) / PL_CHUNK;
  hipLaunchKernelGGL(extrema_place_kernel, dim3(chunks, batch), dim3(1024), 0, st, g, lp, ds, rowmask, rowcnt, rowoff,
                     level_count, raw_total, overflow, ticket, flag, raw, cap_raw);
}

int topk_chunks(int cap_raw) { return (cap_raw + TK_CHUNK - 1) / TK_CHUNK; }

// scratch (topk_scratch_bytes; must arrive zeroed): [ticket: batch ints, padded to 8 bytes][state: batch x nchunk
// 64-bit words][sel_level_count: batch x nlev ints]
void launch_topk(hipStream_t st, const Geom& g, int K, const RawKey* raw, const int* raw_total, int cap_raw,
                 unsigned* hist, RawKey* sel, int* sel_total, int cap_sel, int batch, void* scratch, int* overflow) {
  const int nchunk = topk_chunks(cap_raw);
  int* ticket = reinterpret_cast<int*>(scratch);
  unsigned long long* state = reinterpret_cast<unsigned long long*>(ticket + ((batch + 1) & ~1));
  int* sel_level_count = reinterpret_cast<int*>(state + (size_t)batch * nchunk);
  hipLaunchKernelGGL(topk_select_kernel, dim3(nchunk, batch), dim3(1024), 0, st, g, K, raw, raw_total, cap_raw, hist,
                     ticket, state, sel, sel_total, sel_level_count, cap_sel, nchunk, overflow);
}

size_t topk_scratch_bytes(int cap_raw, int batch, int nlev) {
  return (size_t)((batch + 1) & ~1) * sizeof(int) + (size_t)batch * topk_chunks(cap_raw) * sizeof(unsigned long long) +
         (size_t)batch * nlev * sizeof(int);
}

void launch_math_probe(hipStream_t st, int which, const float* a, const float* b, float* out, int n) {
  hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, st, which, a, b, out, n);
}

}  // namespace hess
