// k_feature.hip -- orientation histograms, multi-orientation expansion and 128-d / 64-d SIFT
// descriptors with normalisation for gfx950 (MI355X).
//
// Replaces ComputeOrientation_Kernel (one thread per keypoint, ProgramCU.cu:1221-1605),
// ReshapeFeatureListCPU (host round trip per level, PyramidCU.cpp:720-924),
// ComputeDescriptor_Kernel (16 threads per keypoint, ProgramCU.cu:1650-1804) and
// NormalizeDescriptor_Kernel (ProgramCU.cu:1950-2054).
//
// One 64-lane wavefront per keypoint (orientation) / per feature (descriptor).  Samples are
// evaluated 64 at a time (one per lane: address, window test, expf weight); the histogram bins
// live one per lane, and each bin receives its contributions in the reference's sample order
// (row-major over the window), so sums are bit-identical to a sequential scan: the contributing
// lanes are visited in ascending order through v_readlane broadcasts and only the lane that owns
// the addressed bin performs the fmaf.
#include "hess_dev.h"
#include "hess_devmath.h"

namespace hess {

namespace {

constexpr double kPI = 3.14159265358979323846;  // config.h:33

__device__ __forceinline__ float rl(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
__device__ __forceinline__ int rli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

__device__ __forceinline__ int float_to_fixed(float v, int n) {  // FLOAT_TO_FIXED_POINT, config.h:73-74
  return (int)((double)(v * (float)(1 << n)) + ((v >= 0.0) ? 0.5 : -0.5));
}

__device__ __forceinline__ void level_of(const Geom& g, int li, int* o, int* l) {
  *o = li / g.dog;
  *l = li - (*o) * g.dog + 1;
}

// ================================= orientation ===============================================

__global__ __launch_bounds__(256) void orientation_kernel(Geom g, OrientParams op, const RawKey* list,
                                                          const int* list_total, int cap_list, const float* got,
                                                          FRec* recs, int* ocount) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int n = list_total[b];
  const int nwaves = gridDim.x * 4;
  const float ten_degree_per_radius = 5.7295779513082320876798154814105f;
  const float radius_per_ten_degrees = (float)(1.0 / 5.7295779513082320876798154814105);
  const float one_third = (float)(1.0 / 3.0);

  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += nwaves) {
    const RawKey rk = list[(long long)b * cap_list + i];
    int o, l;
    level_of(g, rk.level_index, &o, &l);
    const OctGeom& og = g.o[o];
    const float2* gp = reinterpret_cast<const float2*>(got) + og.got_off + ((long long)(l - 1) * g.B + b) * og.plane;
    const int width = og.wa, height = og.h;

    float kx = rk.col + 0.5f, ky = rk.row + 0.5f, kz = op.level_sigma[l];
    if (op.subpixel) {  // ProgramCU.cu:1293-1298
      kx += rk.dx;
      ky += rk.dy;
      kz *= dm_powf_ln(op.ln_sigma_step, rk.ds);
    }
    uint32_t kw_bits = 0;
    int ocnt = 0;

    if (op.num_orientation != 0) {
      const float gsigma = kz * op.gaussian_factor;
      const float win = fabsf(kz) * op.sample_factor;
      const float dist_threshold = win * win + 0.5f;
      const float factor = -0.5f / (gsigma * gsigma);
      const float xmin = fmaxf(1.5f, floorf(kx - win) + 0.5f);
      const float ymin = fmaxf(1.5f, floorf(ky - win) + 0.5f);
      const float xmax = fminf(width - 1.5f, floorf(kx + win) + 0.5f);
      const float ymax = fminf(height - 1.5f, floorf(ky + win) + 0.5f);
      const int nxs = (xmax >= xmin) ? (int)(xmax - xmin) + 1 : 0;
      const int nys = (ymax >= ymin) ? (int)(ymax - ymin) + 1 : 0;
      const int total = nxs * nys;

      float vote = 0.0f;  // lane j < 36 owns vote[j]
      for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        bool inside = false;
        int bin = 0;
        float gx = 0.0f, e = 0.0f;
        if (t < total) {
          const int iy = t / nxs, ix = t - iy * nxs;
          const float x = xmin + (float)ix, y = ymin + (float)iy;
          float dy = y - ky;
          dy *= dy;
          const float dx = x - kx;
          const float sq_dist = fmaf(dx, dx, dy);
          if (!(sq_dist >= dist_threshold)) {
            inside = true;
            const float2 gv = gp[(int)y * width + (int)x];  // tex2D point fetch, ProgramCU.cu:1351
            bin = (int)floorf(gv.y * ten_degree_per_radius);
            if (bin < 0) bin += 36;
            gx = gv.x;
            e = dm_expf(sq_dist * factor);
          }
        }
        uint64_t m = __ballot(inside);
        while (m) {  // ascending lane order = the reference's sample order
          const int j = __builtin_ctzll(m);
          m &= m - 1;
          const int bj = rli(bin, j);
          const float gj = rl(gx, j), ej = rl(e, j);
          if (lane == bj) vote = fmaf(gj, ej, vote);  // ProgramCU.cu:1359
        }
      }
      // six circular 3-tap box passes (ProgramCU.cu:1364-1379): each pass reads only old values
      const int lp = (lane == 0) ? 35 : lane - 1;
      const int ln = (lane >= 35) ? 0 : lane + 1;
#pragma unroll
      for (int p = 0; p < 6; p++) {
        const float pre = __shfl(vote, lp), nxt = __shfl(vote, ln);
        vote = one_third * (pre + vote + nxt);
      }
      const float vote36 = rl(vote, 0);  // vote[36] = vote[0] (ProgramCU.cu:1381), kept across the fold
      if (op.half_sift) {                // ProgramCU.cu:1384-1392
        const float hi = __shfl(vote, (lane + 18) & 63);
        vote = (lane < 18) ? vote + hi : 0.0f;
      }
      if (lane >= 36) vote = -1.0f;  // never a maximum (votes are >= 0)
      float mx = vote;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
      // shuffles stay outside any lane-dependent condition: ds_bpermute returns 0 for a source lane
      // that is masked off at the time it executes
      const float pre = __shfl(vote, lp);
      const float nxt_raw = __shfl(vote, ln);
      const float nxt = (lane == 35) ? vote36 : nxt_raw;

      if (op.num_orientation == 1) {  // ProgramCU.cu:1398-1420
        const uint64_t mm = __ballot(lane < 36 && vote == mx);
        const int index_max = __builtin_ctzll(mm);  // first index reaching the maximum
        const float p0 = rl(pre, index_max), n0 = rl(nxt, index_max), weight = mx;
        const float off = 0.5f * ((n0 - p0) / (weight + weight - n0 - p0));
        const float kw = radius_per_ten_degrees * ((float)index_max + 0.5f + off);
        kw_bits = __float_as_uint(kw);
      } else {  // ProgramCU.cu:1424-1489
        const float vote_threshold = mx * 0.8f;
        const bool peak = (lane < 36) && (vote > vote_threshold) && (vote > pre) && (vote > nxt);
        const float di = 0.5f * ((nxt - pre) / (vote + vote - nxt - pre));
        const float rot = (float)lane + di + 0.5f;
        uint64_t pm = __ballot(peak);
        float mv0 = 0, mv1 = 0, mv2 = 0, mv3 = 0, mr0 = 0, mr1 = 0, mr2 = 0, mr3 = 0;
        while (pm) {
          const int j = __builtin_ctzll(pm);
          pm &= pm - 1;
          float cw = rl(vote, j), cr = rl(rot, j);
          // sorted insertion with strict "<" (equal weights keep the earlier bin first); once the
          // new entry is placed the rest shifts down; a fifth entry falls off (ProgramCU.cu:1454-1469)
          bool placed = false;
          float tw, tr;
#define HESS_INS(S, MV, MR)                                                     \
          if (S < ocnt) {                                                       \
            if (placed || MV < cw) { tw = MV; tr = MR; MV = cw; MR = cr; cw = tw; cr = tr; placed = true; } \
          } else if (S == ocnt) { MV = cw; MR = cr; }
          HESS_INS(0, mv0, mr0)
          HESS_INS(1, mv1, mr1)
          HESS_INS(2, mv2, mr2)
          HESS_INS(3, mv3, mr3)
#undef HESS_INS
          if (ocnt < 4) ocnt++;
        }
        const float mr[4] = {mr0, mr1, mr2, mr3};
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (k < ocnt) {
            float orientation = mr[k] / 36.0f;
            if (orientation < 0) orientation += 1.0f;
            const uint32_t ui = (uint32_t)floorf(orientation * 255.0f);
            packed |= (ui << (8 * k));
          }
        }
        kw_bits = packed;
      }
    } else {
      kw_bits = __float_as_uint(0.0f);
    }
    if (lane == 0) {  // key_store_finish, ProgramCU.cu:1563-1596
      uint32_t posX = (uint32_t)float_to_fixed(kx, 10) & 0x00FFFFFFu;
      uint32_t posY = (uint32_t)float_to_fixed(ky, 10) & 0x00FFFFFFu;
      posX |= (rk.packed & 0xFF000000u);
      posY |= ((rk.packed << 8) & 0xFF000000u);
      uint32_t scale = (uint32_t)float_to_fixed(kz, 8) & 0x0000FFFFu;
      scale |= ((rk.packed & 0x3u) << 30) | (((uint32_t)ocnt & 0x7u) << 27);
      FRec r;
      r.x = posX; r.y = posY; r.z = scale; r.w = kw_bits;
      recs[(long long)b * cap_list + i] = r;
      ocount[(long long)b * cap_list + i] = ocnt;
    }
  }
}

// ================================= feature scan ==============================================

__device__ __forceinline__ int block_scan1(int a, int* total, int* lds) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ia = a;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int na = __shfl_up(ia, d);
    if (lane >= d) ia += na;
  }
  __syncthreads();
  if (lane == 63) lds[wv] = ia;
  __syncthreads();
  int oa = 0, sa = 0;
  for (int k = 0; k < 16; k++) {
    if (k < wv) oa += lds[k];
    sa += lds[k];
  }
  *total = sa;
  return oa + ia - a;
}

__global__ __launch_bounds__(1024) void feature_scan_kernel(Geom g, LimitParams lp, int multi, const RawKey* list,
                                                            const int* list_total, int cap_list, const int* ocount,
                                                            int* foffset, int* feat_total, int* feat_first,
                                                            int cap_feat, int* overflow) {
  __shared__ int lds[64];
  __shared__ int lc[kMaxOct * kMaxDog];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = list_total[b];
  for (int i = tid; i < g.nlev; i += 1024) lc[i] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    int c = 0;
    if (i < n) {
      c = multi ? ocount[(long long)b * cap_list + i] : 1;
      if (c) atomicAdd(&lc[list[(long long)b * cap_list + i].level_index], c);
    }
    int tot;
    const int e = block_scan1(c, &tot, lds);
    const int cb = carry;
    if (i < n) foffset[(long long)b * cap_list + i] = cb + e;
    __syncthreads();
    if (tid == 0) carry = cb + tot;
    __syncthreads();
  }
  if (tid == 0) {
    int total = carry, first = 0;
    // LimitFeatureCount(1) (SiftPyramid.cpp:143,201-278): only after the multi-orientation reshape
    if (multi && lp.threshold > 0 && lp.method != 3) {
      if (lp.method == 2) {
        int i = 0, nf = 0;
        for (; (nf < lp.threshold) && (i < g.nlev); ++i) nf += lc[i];
        if (nf < total) total = nf;
      } else {
        int i = 0;
        while (i < g.nlev && (total - lc[i]) > lp.threshold) { total -= lc[i]; first += lc[i]; i++; }
      }
    }
    if (total > cap_feat) { atomicMax(overflow, total); total = cap_feat; }
    feat_total[b] = total;
    feat_first[b] = first;
  }
}

// ================================= descriptor ================================================

__global__ __launch_bounds__(256) void descriptor_kernel(Geom g, DescParams dp, const RawKey* list,
                                                         const int* list_total, int cap_list, const FRec* recs,
                                                         const int* ocount, const int* foffset,
                                                         const int* feat_total, const int* feat_first,
                                                         const float* got, HostKeypoint* keys, float* desc,
                                                         int cap_feat) {
  __shared__ __attribute__((aligned(16))) float dl[4][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int n = list_total[b];
  const int ftotal = feat_total[b], ffirst = feat_first[b];
  const int nwaves = gridDim.x * 4;
  const int per_kp = dp.multi ? 4 : 1;
  const float rpi = (float)(4.0 / kPI);
  const int dim = dp.half_sift ? 64 : 128;

  for (int wid = blockIdx.x * 4 + wv; wid < n * per_kp; wid += nwaves) {
    const int i = dp.multi ? (wid >> 2) : wid;
    const int k = dp.multi ? (wid & 3) : 0;
    const int cnt = dp.multi ? ocount[(long long)b * cap_list + i] : 1;
    if (k >= cnt) continue;
    const int m = foffset[(long long)b * cap_list + i] + k;
    if (m < ffirst || m >= ffirst + ftotal) continue;
    const int oidx = m - ffirst;
    const FRec rec = recs[(long long)b * cap_list + i];
    const int li = list[(long long)b * cap_list + i].level_index;
    int o, l;
    level_of(g, li, &o, &l);
    const OctGeom& og = g.o[o];
    const float2* gp = reinterpret_cast<const float2*>(got) + og.got_off + ((long long)(l - 1) * g.B + b) * og.plane;
    const int width = og.wa, height = og.h;

    // un-mirrored orientation handed to the kernel (PyramidCU.cpp:764,791; A.1 of SURVEY)
    const float kw = dp.multi ? (float)((2.0 * kPI / 255.0) * (double)((rec.w >> (8 * k)) & 0xFFu))
                              : __uint_as_float(rec.w);
    const float kx = (float)(rec.x & 0x00FFFFFFu) / 1024.0f;
    const float ky = (float)(rec.y & 0x00FFFFFFu) / 1024.0f;
    const float kz = (float)(rec.z & 0x0000FFFFu) / 256.0f;

    if (lane == 0) {  // host keypoint record, PyramidCU.cpp:866-906 / :1097-1137 (host arithmetic)
      const float oss = dp.octave_sigma * (float)(1 << (li / dp.dog));
      const float offset = dp.lowe_origin ? 0.0f : 0.5f;
      HostKeypoint hk;
      hk.x = __fadd_rn(__fmul_rn(oss, kx - 0.5f), offset);
      hk.y = __fadd_rn(__fmul_rn(oss, ky - 0.5f), offset);
      hk.s = oss * kz;
      hk.o = (float)fmod(2.0 * kPI - (double)kw, 2.0 * kPI);
      hk.response = dm_h2f(((rec.x & 0xFF000000u) >> 16) | ((rec.y & 0xFF000000u) >> 24));
      hk.level = (uint16_t)li;
      hk.type = (uint16_t)((rec.z & 0xC0000000u) >> 30);
      keys[(long long)b * cap_feat + oidx] = hk;
    }
    if (!desc) continue;

    const float spt = fabsf(kz * dp.window_factor);
    float s, c;
    dm_sincosf(kw, &s, &c);  // __sincosf, ProgramCU.cu:1698
    const float anglef = (kw > kPI) ? (float)(kw - (2.0 * kPI)) : kw;
    const float cspt = c * spt, sspt = s * spt;
    const float crspt = c / spt, srspt = s / spt;
    const float bsz = fabsf(cspt) + fabsf(sspt);

    for (int cell = 0; cell < 16; cell++) {
      const int ix = cell & 3, iy = cell >> 2;
      const float offx = ix - 1.5f, offy = iy - 1.5f;
      const float ptx = fmaf(cspt, offx, -(sspt * offy)) + kx;
      const float pty = fmaf(cspt, offy, sspt * offx) + ky;
      const float xmin = fmaxf(1.5f, floorf(ptx - bsz) + 0.5f);
      const float ymin = fmaxf(1.5f, floorf(pty - bsz) + 0.5f);
      const float xmax = fminf(width - 1.5f, floorf(ptx + bsz) + 0.5f);
      const float ymax = fminf(height - 1.5f, floorf(pty + bsz) + 0.5f);
      const int nxs = (xmax >= xmin) ? (int)(xmax - xmin) + 1 : 0;
      const int nys = (ymax >= ymin) ? (int)(ymax - ymin) + 1 : 0;
      const int total = nxs * nys;
      float des = 0.0f;  // lane j < 9 owns des[j]
      for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        bool hit = false;
        int fidx = 0;
        float w1 = 0, w2 = 0, wt = 0;
        if (t < total) {
          const int sy = t / nxs, sx = t - sy * nxs;
          const float x = xmin + (float)sx, y = ymin + (float)sy;
          const float dx = x - ptx, dy = y - pty;
          const float nx = fmaf(crspt, dx, srspt * dy);
          const float ny = fmaf(crspt, dy, -(srspt * dx));
          const float nxn = fabsf(nx), nyn = fabsf(ny);
          if ((nxn < 1.0f) && (nyn < 1.0f)) {
            const float2 cc = gp[(int)y * width + (int)x];
            const float dnx = nx + offx, dny = ny + offy;
            const float ww = dm_expf(-0.125f * fmaf(dnx, dnx, dny * dny));
            const float wx = 1.0f - nxn, wy = 1.0f - nyn;
            wt = ww * wx * wy * cc.x;
            float theta = (anglef - cc.y) * rpi;
            if (theta < 0) theta += 8.0f;
            const float fo = floorf(theta);
            fidx = (int)fo;
            w1 = fo + 1.0f - theta;
            w2 = theta - fo;
            hit = (fidx >= 0) && (fidx < 8);  // DYNAMIC_INDEXING=false: only k==fidx, k<8 (ProgramCU.cu:1763-1771)
          }
        }
        uint64_t mk = __ballot(hit);
        while (mk) {
          const int j = __builtin_ctzll(mk);
          mk &= mk - 1;
          const int fj = rli(fidx, j);
          const float w1j = rl(w1, j), w2j = rl(w2, j), wj = rl(wt, j);
          if (lane == fj) des = fmaf(w1j, wj, des);
          if (lane == fj + 1) des = fmaf(w2j, wj, des);
        }
      }
      const float d8 = rl(des, 8);
      if (lane == 0) des += d8;  // des[0] += des[8], ProgramCU.cu:1776
      if (dp.half_sift) {
        const float hi = __shfl(des, (lane + 4) & 63);
        if (lane < 4) { des += hi; dl[wv][cell * 4 + lane] = des; }
      } else {
        if (lane < 8) dl[wv][cell * 8 + lane] = des;
      }
    }
    // same wavefront wrote dl[wv]; LDS operations of one wavefront complete in order
    float* dout = desc + ((long long)b * cap_feat + oidx) * dim;
    if (dp.half_sift) {
      float2 v = make_float2(0, 0);
      if (lane < 32) v = *reinterpret_cast<const float2*>(&dl[wv][lane * 2]);
      if (dp.normalize) {
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          float part = fmaf(v.y, v.y, v.x * v.x);
#pragma unroll
          for (int d = 16; d >= 1; d >>= 1) part += __shfl_down(part, d);
          const float nrm = 1.0f / sqrtf(rl(part, 0));
          if (pass == 0) { v.x = fminf(0.2f, v.x * nrm); v.y = fminf(0.2f, v.y * nrm); }
          else { v.x *= nrm; v.y *= nrm; }
        }
      }
      if (lane < 32) *reinterpret_cast<float2*>(dout + lane * 2) = v;
    } else {
      float4 v = make_float4(0, 0, 0, 0);
      if (lane < 32) v = *reinterpret_cast<const float4*>(&dl[wv][lane * 4]);
      if (dp.normalize) {
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
          float part = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
#pragma unroll
          for (int d = 16; d >= 1; d >>= 1) part += __shfl_down(part, d);
          const float nrm = 1.0f / sqrtf(rl(part, 0));
          if (pass == 0) {
            v.x = fminf(0.2f, v.x * nrm); v.y = fminf(0.2f, v.y * nrm);
            v.z = fminf(0.2f, v.z * nrm); v.w = fminf(0.2f, v.w * nrm);
          } else { v.x *= nrm; v.y *= nrm; v.z *= nrm; v.w *= nrm; }
        }
      }
      if (lane < 32) *reinterpret_cast<float4*>(dout + lane * 4) = v;
    }
  }
}

}  // namespace

void launch_orientation(hipStream_t st, const Geom& g, const OrientParams& op, const RawKey* list,
                        const int* list_total, int cap_list, const float* got, FRec* recs, int* ocount,
                        int batch) {
  int blocks = (cap_list + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(orientation_kernel, dim3(blocks, batch), dim3(256), 0, st, g, op, list, list_total, cap_list,
                     got, recs, ocount);
}

void launch_feature_scan(hipStream_t st, const Geom& g, const LimitParams& lp, int multi, const RawKey* list,
                         const int* list_total, int cap_list, const int* ocount, int* foffset, int* feat_total,
                         int* feat_first, int cap_feat, int* overflow, int batch) {
  hipLaunchKernelGGL(feature_scan_kernel, dim3(batch), dim3(1024), 0, st, g, lp, multi, list, list_total, cap_list,
                     ocount, foffset, feat_total, feat_first, cap_feat, overflow);
}

void launch_descriptor(hipStream_t st, const Geom& g, const DescParams& dp, const RawKey* list,
                       const int* list_total, int cap_list, const FRec* recs, const int* ocount,
                       const int* foffset, const int* feat_total, const int* feat_first, const float* got,
                       HostKeypoint* keys, float* desc, int cap_feat, int batch) {
  long long waves = (long long)cap_list * (dp.multi ? 4 : 1);
  int blocks = (int)((waves + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(descriptor_kernel, dim3(blocks, batch), dim3(256), 0, st, g, dp, list, list_total, cap_list,
                     recs, ocount, foffset, feat_total, feat_first, got, keys, desc, cap_feat);
}

}  // namespace hess
