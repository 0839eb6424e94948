// hess_planes.h -- det-Hessian / gradient of whole plane rows from HBM, shared by k_detect.hip (hessian_kernel) and
// k_gauss.hip (the launch that finishes the chained octaves): ComputeHessian_Kernel, ProgramCU.cu:523-595, with the
// reference's 1-D neighbour addressing.
#pragma once
#include "hess_dev.h"
#include "hess_devmath.h"

namespace hess {
namespace {

__device__ __forceinline__ float tex1(const float* p, int n, int i) { return (i < 0 || i >= n) ? 0.0f : p[i]; }

// neighbour lanes by DPP wave shifts (one VALU move each; no LDS crossbar round trip)
__device__ __forceinline__ float lane_prev(float v) {  // lane i <- lane i-1 (lane 0: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {  // lane i <- lane i+1 (lane 63: 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}

// 4 pixels per thread, 16-byte loads/stores.  Neighbour addressing follows the reference's 1-D
// linear texture: index +-1 wraps across row ends, anything outside [0, wa*h) reads 0.
// One plane (src = the Gaussian level, dh = its det-H plane, gt = its gradient plane or null); gid = this thread's
// 4-pixel group of the plane.
__device__ __forceinline__ void hessian_rows_body(const float* src, float* dh, float2* gt, int wa, int h, float norm,
                                                  int gid) {
  const int groups_per_row = wa >> 2;
  const int nthreads = groups_per_row * h;
  if (gid >= nthreads) return;
  // row = gid / groups_per_row without an integer division: float estimate, then one exact correction step
  int row = (int)(((float)gid + 0.5f) * (1.0f / (float)groups_per_row));
  int rem = gid - row * groups_per_row;
  if (rem < 0) { row--; rem += groups_per_row; }
  else if (rem >= groups_per_row) { row++; rem -= groups_per_row; }
  const int x = rem << 2;
  const int n = wa * h;
  const int idx = row * wa + x;

  // Rows idx-wa, idx, idx+wa as 16-byte loads.  The +-1 neighbours are, in the reference's 1-D
  // addressing, simply the adjacent thread's outer elements (also across a row end), so they come
  // from the neighbouring lanes; only the first/last lane of a wavefront (or of the plane) loads them.
  float U[6], M[6], D[6];
  {
    const float4 m = *reinterpret_cast<const float4*>(src + idx);
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(0.f, 0.f, 0.f, 0.f);
    const int iu = row >= 1 ? idx - wa : idx, id = row + 1 < h ? idx + wa : idx;
    const float4 uq = *reinterpret_cast<const float4*>(src + iu);
    const float4 dq = *reinterpret_cast<const float4*>(src + id);
    if (row >= 1) u = uq;
    if (row + 1 < h) d = dq;
    M[1] = m.x; M[2] = m.y; M[3] = m.z; M[4] = m.w;
    U[1] = u.x; U[2] = u.y; U[3] = u.z; U[4] = u.w;
    D[1] = d.x; D[2] = d.y; D[3] = d.z; D[4] = d.w;
  }
  {
    const int lane = threadIdx.x & 63;
    const float ul = lane_prev(U[4]), ml = lane_prev(M[4]), dl = lane_prev(D[4]);
    const float ur = lane_next(U[1]), mr = lane_next(M[1]), dr = lane_next(D[1]);
    U[0] = ul; M[0] = ml; D[0] = dl;
    U[5] = ur; M[5] = mr; D[5] = dr;
    // (gid & 63 == lane here: the callers hand consecutive gids to consecutive lanes of whole wavefronts)
    if (lane == 0) {
      U[0] = tex1(src, n, idx - wa - 1); M[0] = tex1(src, n, idx - 1); D[0] = tex1(src, n, idx + wa - 1);
    }
    if (lane == 63 || gid == nthreads - 1) {
      U[5] = tex1(src, n, idx - wa + 4); M[5] = tex1(src, n, idx + 4); D[5] = tex1(src, n, idx + wa + 4);
    }
  }
  float hv[4];
  float2 gv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float v11 = U[j], v12 = U[j + 1], v13 = U[j + 2];
    const float v21 = M[j], v22 = M[j + 1], v23 = M[j + 2];
    const float v31 = D[j], v32 = D[j + 1], v33 = D[j + 2];
    hv[j] = dm_deth(v11, v12, v13, v21, v22, v23, v31, v32, v33, norm);  // ProgramCU.cu:536-553
    if (gt) gv[j] = dm_grad_theta(v12, v21, v23, v32);                   // :556-559
  }
  *reinterpret_cast<float4*>(dh + idx) = make_float4(hv[0], hv[1], hv[2], hv[3]);
  if (gt) {
    float2* g = gt + idx;
    *reinterpret_cast<float4*>(g) = make_float4(gv[0].x, gv[0].y, gv[1].x, gv[1].y);
    *reinterpret_cast<float4*>(g + 2) = make_float4(gv[2].x, gv[2].y, gv[3].x, gv[3].y);
  }
}

struct LevelNorms { float v[kMaxLev]; };

// det-H (+ gradient/theta) of levels 0 .. nlv-1 of octaves >= first_oct from HBM: the levels the level-chain launches
// (gauss_chain_kernel, k_gauss.hip) produce without their det-H / gradient planes.  blk -> (octave, level, 256 groups).
__device__ __forceinline__ void hessian_low_levels(const Geom& g, const float* gauss, float* deth, float2* got,
                                                   const LevelNorms& nm, int first_oct, int nlv, int blk, int b) {
  int o = first_oct, nb = 0;
  for (; o < g.noct; o++) {
    nb = ((g.o[o].wa >> 2) * g.o[o].h + 255) >> 8;
    if (blk < nb * nlv) break;
    blk -= nb * nlv;
  }
  if (o >= g.noct) return;
  const int lvl = blk / nb, pb = blk - lvl * nb;
  const int n = g.o[o].plane;
  const long long poff = g.o[o].lvl_off + ((long long)lvl * g.B + b) * n;
  float2* gt = (lvl >= 1 && lvl <= g.dog) ? got + g.o[o].got_off + ((long long)(lvl - 1) * g.B + b) * n : nullptr;
  hessian_rows_body(gauss + poff, deth + poff, gt, g.o[o].wa, g.o[o].h, nm.v[lvl], pb * 256 + (int)threadIdx.x);
}

}  // namespace
}  // namespace hess
