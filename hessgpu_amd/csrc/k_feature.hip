// k_feature.hip -- orientation histograms, multi-orientation expansion and 128-d / 64-d SIFT
// descriptors with normalisation for gfx950 (MI355X).
//
// Replaces ComputeOrientation_Kernel (one thread per keypoint, ProgramCU.cu:1221-1605),
// ReshapeFeatureListCPU (host round trip per level, PyramidCU.cpp:720-924),
// ComputeDescriptor_Kernel (16 threads per keypoint, ProgramCU.cu:1650-1804) and
// NormalizeDescriptor_Kernel (ProgramCU.cu:1950-2054).
//
// One 64-lane wavefront per keypoint (orientation) / per feature (descriptor); every histogram bin
// receives its contributions in the reference's sample order (row-major over the window), so sums are
// bit-identical to a sequential scan.
//   orientation  64 window samples per step, one per lane, branch-free; the samples inside the disc are
//                compacted in order into an LDS record list; lane j (bin j) reads every record by LDS
//                broadcast and adds it with coefficient (bin == j ? gradient : 0);
//   descriptor   lane = cell*4 + q: the four lanes of a cell take four consecutive samples of the cell's
//                box per iteration, hits are appended in order to the cell's LDS list, and lane (cell, q)
//                owns bins q, q+4 (q = 0 also 8) of its cell and walks the list (see descriptor_kernel).
// Built without the SLP vectoriser (hessgpu_amd/build.py): packed FP32 issues at half rate on gfx950.
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
  __shared__ __attribute__((aligned(16))) float4 orec[4][64];  // per wavefront: (bin, gradient, weight) records
  const int lane = threadIdx.x & 63;
  float4* const myrec = orec[threadIdx.x >> 6];
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
    FRec urec = {0u, 0u, 0u, 0u};
    if (op.existing) {  // ProgramCU.cu:1246-1278: unpack position and scale from the uploaded record
      urec = recs[(long long)b * cap_list + i];
      kx = (float)(urec.x & 0x00FFFFFFu) / 1024.0f;
      ky = (float)(urec.y & 0x00FFFFFFu) / 1024.0f;
      kz = (float)(urec.z & 0x0000FFFFu) / 256.0f;
    } else if (op.subpixel) {  // ProgramCU.cu:1293-1298
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
      const float inv = 1.0f / (float)(nxs > 0 ? nxs : 1);
      const int base = (int)ymin * width + (int)xmin;  // (int)y * width + (int)x of sample (0, 0)
      for (int t0 = 0; t0 < total; t0 += 64) {
        // 64 samples of the window in scan order, branch-free; the ones inside the circle are
        // compacted, in order, into the wavefront's record list
        const int t = t0 + lane;
        const int iy = (int)(((float)t + 0.5f) * inv);  // = t / nxs (exact: |error| << 0.5/nxs)
        const int ix = t - __mul24(iy, nxs);
        const float x = xmin + (float)ix, y = ymin + (float)iy;
        float dy = y - ky;
        dy *= dy;
        const float dx = x - kx;
        const float sq_dist = fmaf(dx, dx, dy);
        const bool inside = (t < total) & !(sq_dist >= dist_threshold);
        const float2 gv = gp[inside ? base + __mul24(iy, width) + ix : 0];  // tex2D point fetch, ProgramCU.cu:1351
        int bin = (int)floorf(gv.y * ten_degree_per_radius);
        bin = (bin < 0) ? bin + 36 : bin;
        const float e = dm_expf(sq_dist * factor);
        const uint64_t mk = __builtin_amdgcn_ballot_w64(inside);
        const int nin = __popcll(mk);
        if (inside) myrec[__popcll(mk & ((1ull << lane) - 1ull))] = make_float4(__int_as_float(bin), gv.x, e, 0.0f);
        // lane j adds the records of bin j in list order = the reference's sample order
        // (ProgramCU.cu:1359: vote[bin] += gradient * weight); for the other bins the coefficient is 0
        // and fmaf(0, e, vote) == vote (votes and weights are non-negative and finite)
        for (int k0 = 0; k0 < nin; k0 += 4) {
          float4 r[4];
#pragma unroll
          for (int u = 0; u < 4; u++) r[u] = myrec[min(k0 + u, 63)];  // same address in all lanes: broadcast
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (k0 + u < nin) {  // wavefront-uniform
              const float cgx = (__float_as_int(r[u].x) == lane) ? r[u].y : 0.0f;
              vote = fmaf(cgx, r[u].z, vote);
            }
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

      if (op.num_orientation == 1 || op.existing) {  // ProgramCU.cu:1398-1420
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
    if (lane == 0 && op.existing) {  // ProgramCU.cu:1597-1602: only the orientation is written back
      urec.w = kw_bits;
      recs[(long long)b * cap_list + i] = urec;
      ocount[(long long)b * cap_list + i] = 0;
    } else if (lane == 0) {  // key_store_finish, ProgramCU.cu:1563-1596
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
                                                            int* foffset, int* fsrc, int* feat_total,
                                                            int* feat_first, int cap_feat, int* overflow) {
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
    if (i < n) {
      foffset[(long long)b * cap_list + i] = cb + e;
      // feature m -> (keypoint i, orientation rank k): lets the descriptor stage give every
      // wavefront real work (ReshapeFeatureListCPU's expansion, PyramidCU.cpp:780-796)
      for (int k = 0; k < c; k++)
        if (cb + e + k < cap_feat) fsrc[(long long)b * cap_feat + cb + e + k] = i * 4 + k;
    }
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

// Records a cell's list holds between two drains.  32 (17 KB of lists per workgroup, six wavefronts per
// SIMD by registers) measured 6 % faster than 64 (four wavefronts per SIMD by LDS); 16-24 drain too often.
constexpr int DL_CAP = 32;
constexpr int DL_STRIDE = DL_CAP + 2;  // list pitch in records: 68 dwords = 4 (mod 64) -> the 16 lists of a
                               // wavefront start in different LDS banks (conflict-free b128 reads)

// One wavefront per feature.  Lane = cell*4 + q.
//   phase 1  the four lanes of a cell take four consecutive samples of the cell's box per iteration
//            (scan order of ProgramCU.cu:1723-1774: y outer, x inner), so the 16 cells advance together
//            with no cross-lane broadcast; hits are appended, in scan order, to the cell's list in LDS;
//   phase 2  ("drain") lane (cell, q) owns the accumulators des[q], des[q+4] (q = 0 also des[8]) of
//            its cell and walks the list in order: one fmaf per record and bin, exactly the reference's
//            `des[fidx] += w1*weight; des[fidx+1] += w2*weight` sequence for every bin.
// The coefficient of bin j for a record is written as med3(0, (j+1)-theta, theta-(j-1)): for
// floor(theta) == j that is w1 = fo+1-theta, for floor(theta) == j-1 it is w2 = theta-fo (same single
// subtraction as the reference), otherwise 0 -- and fmaf(0, weight, acc) == acc because weights and
// accumulators are non-negative and finite.
__global__ __launch_bounds__(256) void descriptor_kernel(Geom g, DescParams dp, const RawKey* list,
                                                         int cap_list, const FRec* recs,
                                                         const int* fsrc, const int* feat_total,
                                                         const int* feat_first, const int* img_base,
                                                         const float* got, HostKeypoint* keys, float* desc,
                                                         int cap_feat) {
  __shared__ __attribute__((aligned(16))) float dl[4][128];
  __shared__ __attribute__((aligned(16))) float2 rec_lds[4][16 * DL_STRIDE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int ftotal = feat_total[b], ffirst = feat_first[b];
  const long long obase = img_base[b];  // packed output: images of the batch back to back
  const int nwaves = gridDim.x * 4;
  const float rpi = (float)(4.0 / kPI);
  const int dim = dp.half_sift ? 64 : 128;
  const int mycell = lane >> 2, sub = lane & 3;
  float2* const rlist = &rec_lds[wv][0];
  float2* const mylist = rlist + mycell * DL_STRIDE;
  // stale list entries are read (with weight forced to 0) when lists have different lengths: make
  // sure they are finite from the start
  for (int i = lane; i < 16 * DL_STRIDE; i += 64) rlist[i] = make_float2(0.0f, 0.0f);
  // bin constants of this lane: des[sub] and des[sub+4]; des[8] only gets w2 of floor(theta) == 7
  const float jlo1 = (float)(sub + 1), jlom = (float)(sub - 1);
  const float jhi1 = (float)(sub + 5), jhim = (float)(sub + 3);
  const float j8m = (sub == 0) ? 7.0f : 1.0e6f;
  const float theta_end = dp.dynamic_indexing ? 8.00000095f : 8.0f;  // next float after 8: admits theta == 8 only

  for (int m = ffirst + blockIdx.x * 4 + wv; m < ffirst + ftotal; m += nwaves) {
    const int src = fsrc[(long long)b * cap_feat + m];
    const int i = src >> 2, k = src & 3;
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
      keys[obase + oidx] = hk;
      if (dp.hkeys) dp.hkeys[obase + oidx] = hk;
    }
    if (!desc) continue;

    const float spt = fabsf(kz * dp.window_factor);
    float s, c;
    dm_sincosf(kw, &s, &c);  // __sincosf, ProgramCU.cu:1698
    const float anglef = (kw > kPI) ? (float)(kw - (2.0 * kPI)) : kw;
    const float cspt = c * spt, sspt = s * spt;
    const float crspt = c / spt, srspt = s / spt;
    const float bsz = fabsf(cspt) + fabsf(sspt);

    // Cell geometry (ProgramCU.cu:1692-1716), every lane for its own cell.
    const float offx = (mycell & 3) - 1.5f, offy = (mycell >> 2) - 1.5f;
    const float ptx = fmaf(cspt, offx, -(sspt * offy)) + kx;
    const float pty = fmaf(cspt, offy, sspt * offx) + ky;
    const float xmin = fmaxf(1.5f, floorf(ptx - bsz) + 0.5f);
    const float ymin = fmaxf(1.5f, floorf(pty - bsz) + 0.5f);
    const float xmax = fminf(width - 1.5f, floorf(ptx + bsz) + 0.5f);
    const float ymax = fminf(height - 1.5f, floorf(pty + bsz) + 0.5f);
    const int nxs = (xmax >= xmin) ? (int)(xmax - xmin) + 1 : 0;
    const int nys = (ymax >= ymin) ? (int)(ymax - ymin) + 1 : 0;
    const int total = nxs * nys;
    const float inv = 1.0f / (float)(nxs > 0 ? nxs : 1);
    // sample (sx, sy) of the box is pixel (xmin+sx, ymin+sy): (int)y * width + (int)x = base + sy*width + sx
    const int base = (int)ymin * width + (int)xmin;
    int maxtotal = total;
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) maxtotal = max(maxtotal, __shfl_xor(maxtotal, d));
    maxtotal = rli(maxtotal, 0);
    const int nit = (maxtotal + 3) >> 2;

    float acc_lo = 0.0f, acc_hi = 0.0f, acc_8 = 0.0f;
    int cnt = 0;  // records in this cell's list (same value in the four lanes of the cell)

    auto update = [&](float theta, float w) {
      const float c0 = __builtin_amdgcn_fmed3f(0.0f, jlo1 - theta, theta - jlom);  // ProgramCU.cu:1752-1753
      acc_lo = fmaf(c0, w, acc_lo);
      const float c1 = __builtin_amdgcn_fmed3f(0.0f, jhi1 - theta, theta - jhim);
      acc_hi = fmaf(c1, w, acc_hi);
      const float c8 = fmaxf(0.0f, theta - j8m);
      acc_8 = fmaf(c8, w, acc_8);
    };
    auto drain = [&]() {
      for (int s0 = 0; __any(s0 < cnt); s0 += 4) {
        const float4 ra = *reinterpret_cast<const float4*>(mylist + s0);
        const float4 rb = *reinterpret_cast<const float4*>(mylist + s0 + 2);
        update(ra.x, (s0 < cnt) ? ra.y : 0.0f);
        update(ra.z, (s0 + 1 < cnt) ? ra.w : 0.0f);
        update(rb.x, (s0 + 2 < cnt) ? rb.y : 0.0f);
        update(rb.z, (s0 + 3 < cnt) ? rb.w : 0.0f);
      }
      cnt = 0;
    };

    for (int it0 = 0; it0 < nit; it0 += 4) {
      if (__any(cnt > DL_CAP - 16)) drain();
      // stage A (4 iterations): geometry, window test, and the gradient gathers issued back to back
      bool in[4];
      float nxa[4], nya[4];
      float2 cca[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = (it0 + u) * 4 + sub;
        const int sy = (int)(((float)t + 0.5f) * inv);  // = t / nxs (exact: |error| << 0.5/nxs)
        const int sx = t - __mul24(sy, nxs);  // box sides are far below 2^23: 24-bit multiplies are exact
        const float x = xmin + (float)sx, y = ymin + (float)sy;
        const float dx = x - ptx, dy = y - pty;
        nxa[u] = fmaf(crspt, dx, srspt * dy);
        nya[u] = fmaf(crspt, dy, -(srspt * dx));
        in[u] = (t < total) & (fabsf(nxa[u]) < 1.0f) & (fabsf(nya[u]) < 1.0f);
        cca[u] = gp[in[u] ? base + __mul24(sy, width) + sx : 0];
      }
      // stage B: weights, ordered append to the cell's list
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float nxn = fabsf(nxa[u]), nyn = fabsf(nya[u]);
        const float dnx = nxa[u] + offx, dny = nya[u] + offy;
        const float ww = dm_expf(-0.125f * fmaf(dnx, dnx, dny * dny));
        const float wx = 1.0f - nxn, wy = 1.0f - nyn;
        const float wt = ww * wx * wy * cca[u].x;
        float theta = (anglef - cca[u].y) * rpi;
        theta = (theta < 0) ? theta + 8.0f : theta;
        // DYNAMIC_INDEXING=false: a sample with floor(theta) == 8 adds nothing (ProgramCU.cu:1763-1771);
        // with -di it adds w1*weight = weight to des[8] (:1755-1759; the write to des[9] adds 0) -- which is
        // what the bin-8 coefficient max(0, theta - 7) gives once the record is let through
        const bool hit = in[u] & (theta >= 0.0f) & (theta < theta_end);
        const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
        const uint32_t nib = (uint32_t)(mk >> (lane & 60)) & 15u;  // hits of this cell's four lanes
        // misses go to the padding slot of the list (never read), so the chunk stays branch-free
        mylist[hit ? cnt + __popc(nib & ((1u << sub) - 1u)) : DL_STRIDE - 1] = make_float2(theta, wt);
        cnt += __popc(nib);
      }
    }
    drain();
    if (sub == 0) acc_lo += acc_8;  // des[0] += des[8], ProgramCU.cu:1776
    if (dp.half_sift) {
      dl[wv][mycell * 4 + sub] = acc_lo + acc_hi;  // des[k] += des[k+4], ProgramCU.cu:1782-1785
    } else {
      dl[wv][mycell * 8 + sub] = acc_lo;
      dl[wv][mycell * 8 + 4 + sub] = acc_hi;
    }
    // same wavefront wrote dl[wv]; LDS operations of one wavefront complete in order
    float* dout = desc + (obase + oidx) * dim;
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
      if (dp.hdesc && lane < 32) *reinterpret_cast<float2*>(dp.hdesc + (obase + oidx) * dim + lane * 2) = v;
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
      if (dp.hdesc && lane < 32) *reinterpret_cast<float4*>(dp.hdesc + (obase + oidx) * dim + lane * 4) = v;
    }
  }
}

__global__ void image_base_kernel(const int* feat_total, int* img_base, int batch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < batch; b++) { img_base[b] = acc; acc += feat_total[b]; }
    img_base[batch] = acc;
  }
}

}  // namespace

void launch_image_base(hipStream_t st, const int* feat_total, int* img_base, int batch) {
  hipLaunchKernelGGL(image_base_kernel, dim3(1), dim3(64), 0, st, feat_total, img_base, batch);
}

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
                         const int* list_total, int cap_list, const int* ocount, int* foffset, int* fsrc,
                         int* feat_total, int* feat_first, int cap_feat, int* overflow, int batch) {
  hipLaunchKernelGGL(feature_scan_kernel, dim3(batch), dim3(1024), 0, st, g, lp, multi, list, list_total, cap_list,
                     ocount, foffset, fsrc, feat_total, feat_first, cap_feat, overflow);
}

void launch_descriptor(hipStream_t st, const Geom& g, const DescParams& dp, const RawKey* list,
                       int cap_list, const FRec* recs, const int* fsrc, const int* feat_total,
                       const int* feat_first, const int* img_base, const float* got, HostKeypoint* keys,
                       float* desc, int cap_feat, int batch) {
  int blocks = (cap_feat + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(descriptor_kernel, dim3(blocks, batch), dim3(256), 0, st, g, dp, list, cap_list, recs, fsrc,
                     feat_total, feat_first, img_base, got, keys, desc, cap_feat);
}

}  // namespace hess
