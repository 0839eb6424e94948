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
//   orientation  64 window samples per step, one per lane, branch-free; the samples inside the disc are grouped
//                by bin (rank inside the group from ballots of the bin bits, groups placed back to back in an
//                LDS list by a DPP scan of the group sizes), each group in lane order = the reference's sample
//                order; lane j (bin j) reads and adds only its own group;
//   descriptor   lane = cell*4 + q: the four lanes of a cell take four consecutive samples of the cell's
//                box per iteration (the reference's scan order); lane (cell, q) owns bins q, q+4 (q = 0 also 8)
//                of its cell and adds the iteration's samples to them through a per-wavefront LDS coefficient
//                table (see descriptor_kernel).
// Built without the SLP vectoriser (hessgpu_amd/build.py): packed FP32 issues at half rate on gfx950.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

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

// ================================= feature scan ==============================================

// exclusive scan over the workgroup (any number of whole wavefronts up to 16); lds: 16 ints
__device__ __forceinline__ int block_scan1(int a, int* total, int* lds) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
  for (int k = 0; k < nw; k++) {
    if (k < wv) oa += lds[k];
    sa += lds[k];
  }
  *total = sa;
  return oa + ia - a;
}

// Multi-orientation expansion of ONE image by the calling workgroup (all its threads): prefix sum of the orientation
// counts, feature -> (keypoint, rank) table, -tc rule on the expanded counts, and -- by the image that finishes last --
// the packed output layout of the batch.  (Run by the orientation launch's last workgroup per image instead of a launch
// of its own, it cost more than it saved: the scan's registers took the orientation kernel from eight to five
// wavefronts per SIMD and a 256-thread workgroup scans slower than a 1024-thread one -- 37 us against 16 + 13.)
struct FeatScan {
  LimitParams lp;
  int multi;
  int* foffset;
  int* fsrc;
  int* feat_total;
  int* feat_first;
  int cap_feat;
  int* overflow;    // the batch's four overflow words (the scan raises word 1); word 8 of the same zeroed block counts the
                    // images that have finished, so that the last one can lay out the packed output of the whole batch
  int* img_base;
  int* host_small;
};

// Inclusive prefix sum over the 64 lanes by DPP row shifts and row broadcasts (no LDS round trips).
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
  return v;
}

__device__ __forceinline__ void feature_scan_image(const Geom& g, const FeatScan& fs, const RawKey* list, const int* list_total,
                                                   int cap_list, const int* ocount, int b, int nimg, int chunk, int nchunk) {
  const LimitParams& lp = fs.lp;
  const int multi = fs.multi, cap_feat = fs.cap_feat;
  int* const foffset = fs.foffset; int* const fsrc = fs.fsrc; int* const feat_total = fs.feat_total;
  int* const feat_first = fs.feat_first; int* const overflow = fs.overflow; int* const img_base = fs.img_base;
  int* const host_small = fs.host_small;
  __shared__ int lds[64];
  __shared__ int lc[kMaxOct * kMaxDog];
  __shared__ int carry;
  const int tid = threadIdx.x, NTH = blockDim.x;
  const int n = list_total[b];
  // The level totals are only read by the -tc1 / -tc2 rules on the expanded counts (below): without one, no keypoint's
  // level is loaded (4 bytes of a 32-byte record each: 2 MB of sectors through one CU for the 65 536 keypoints of a
  // 4096^2 image) and the image's list is split over the launch's `nchunk` workgroups.  A workgroup needs the number of
  // features before its chunk: it adds up the counts of the keypoints before it itself (coalesced, <= 256 KB) -- no
  // workgroup waits for another, and the last chunk's workgroup knows the image's total.  With a rule: one workgroup.
  const bool need_levels = multi && lp.threshold > 0 && lp.method != 3;
  if (need_levels) nchunk = 1;
  if (chunk >= nchunk) return;
  const int clen = (((n + nchunk - 1) / nchunk) + 1023) & ~1023;
  const int c0 = min(n, chunk * clen), c1 = min(n, c0 + clen);
  const int lane = tid & 63, wv = tid >> 6, nwv = NTH >> 6;
  for (int i = tid; i < g.nlev; i += NTH) lc[i] = 0;
  {
    int before = multi ? 0 : c0;
    if (multi && c0 > 0) {  // (workgroup-uniform)
      int part = 0;
      for (int k = tid; k < c0; k += NTH) part += ocount[(long long)b * cap_list + k];
      part = wave_inclusive_scan(part);
      if (lane == 63) lds[wv] = part;
      __syncthreads();
      for (int w = 0; w < nwv; w++) before += lds[w];
      __syncthreads();
    }
    if (tid == 0) carry = before;
  }
  __syncthreads();
  // A pass covers NTH * FC keypoints (16 384 with 1024 threads): wavefront w takes the 64 * FC consecutive keypoints
  // [w * 64 * FC, ...) of the pass in FC steps of 64 -- lane = keypoint, so every load and store of a step is coalesced
  // (the first form gave a THREAD 16 consecutive keypoints: its stores of a step went to 64 different cache lines, and its
  // per-keypoint LDS atomics on the few level totals serialised lane by lane: 0.159 ms for the 65 536 keypoints of a 4096^2
  // image on the one CU this workgroup runs on).  Counts are scanned inside the wavefront (DPP), the wavefronts' totals
  // through LDS, and the level totals take one LDS atomic per step unless the step straddles a level boundary.
  constexpr int FC = 16;
  for (int base = c0; base < c1; base += NTH * FC) {
    const int npass = min(c1 - base, NTH * FC);
    const int fc = (npass + NTH - 1) / NTH;  // steps per wavefront in this pass (a short list is spread over all wavefronts)
    int cc[FC], ex[FC], tot[FC], wsum = 0;
#pragma unroll
    for (int u = 0; u < FC; u++) {
      const int k = (wv * fc + u) * 64 + lane;
      const bool in = u < fc && k < npass;
      const long long at = (long long)b * cap_list + base + (in ? k : 0);
      cc[u] = in ? (multi ? ocount[at] : 1) : 0;
      const int inc = wave_inclusive_scan(cc[u]);
      ex[u] = inc - cc[u];
      tot[u] = __builtin_amdgcn_readlane(inc, 63);
      wsum += tot[u];
      if (need_levels && tot[u]) {  // (wavefront-uniform)
        const int lv = in ? list[at].level_index : 0;
        const int lv0 = __builtin_amdgcn_readfirstlane(lv);
        if (__builtin_amdgcn_ballot_w64(cc[u] != 0 && lv != lv0) == 0) {
          if (lane == 0) atomicAdd(&lc[lv0], tot[u]);
        } else if (cc[u]) {
          atomicAdd(&lc[lv], cc[u]);
        }
      }
    }
    if (lane == 0) lds[wv] = wsum;
    __syncthreads();
    int run = carry, ptot = 0;
    for (int w = 0; w < nwv; w++) {
      const int t = lds[w];
      run += w < wv ? t : 0;
      ptot += t;
    }
#pragma unroll
    for (int u = 0; u < FC; u++) {
      const int k = (wv * fc + u) * 64 + lane;
      if (u < fc && k < npass) {
        const int off = run + ex[u];
        foffset[(long long)b * cap_list + base + k] = off;
        // feature m -> (keypoint i, orientation rank k): lets the descriptor stage give every
        // wavefront real work (ReshapeFeatureListCPU's expansion, PyramidCU.cpp:780-796)
        for (int q = 0; q < cc[u]; q++)
          if (off + q < cap_feat) fsrc[(long long)b * cap_feat + off + q] = (base + k) * 4 + q;
      }
      run += tot[u];
    }
    __syncthreads();
    if (tid == 0) carry += ptot;
    __syncthreads();
  }
  if (chunk != nchunk - 1) return;  // (the last chunk's workgroup: `carry` is the image's total)
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
    if (total > cap_feat) { atomicMax(overflow + 1, total); total = cap_feat; }
    feat_total[b] = total;
    feat_first[b] = first;
    // Packed output offsets of the batch (images back to back), by whichever workgroup finishes last.  With
    // host-direct delivery the offsets and the overflow words also go to the pinned host block the caller reads once
    // the stream has drained (no device->host copy commands).
    __threadfence();
    if (atomicAdd(overflow + 8, 1) == nimg - 1) {
      __threadfence();
      int acc = 0;
      for (int i = 0; i < nimg; i++) {
        img_base[i] = acc;
        if (host_small) host_small[i] = acc;
        acc += __hip_atomic_load(feat_total + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      img_base[nimg] = acc;
      if (host_small) {
        host_small[nimg] = acc;
        for (int i = 0; i < 4; i++)
          host_small[nimg + 1 + i] = __hip_atomic_load(overflow + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

__global__ __launch_bounds__(1024) void feature_scan_kernel(Geom g, FeatScan fs, const RawKey* list, const int* list_total,
                                                            int cap_list, const int* ocount) {
  feature_scan_image(g, fs, list, list_total, cap_list, ocount, (int)blockIdx.y, (int)gridDim.y, (int)blockIdx.x, (int)gridDim.x);
}

// ================================= orientation ===============================================

// e^x for the Gaussian windows of the orientation and descriptor stages: same operations and results as dm_expf() on -87 <= x <= 88 (there
// its two range clamps select nothing, and p * 2^n by v_ldexp is the same single rounding as the multiplication
// by the constructed power of two); arguments here lie in [-1.5625, 0] for every sample that is used.
__device__ __forceinline__ float dm_expf_inrange(float x) {
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float z = r * r;
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  p = fmaf(p, z, r);
  p = p + 1.0f;
  return __builtin_amdgcn_ldexpf(p, (int)n);
}


__global__ __launch_bounds__(256) void orientation_kernel(Geom g, OrientParams op, const RawKey* list,
                                                          const int* list_total, int cap_list, const float* got,
                                                          FRec* recs, int* ocount) {
  __shared__ __attribute__((aligned(8))) float2 orec[4][66];  // per wavefront: (gradient, weight) records of a step, by bin (+ spare)
  const int lane = threadIdx.x & 63;
  float2* const myrec = orec[threadIdx.x >> 6];
  __shared__ int osize[4][36];  // per wavefront: samples per bin in the current step
  int* const mysize = osize[threadIdx.x >> 6];
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
        const float e = dm_expf_inrange(sq_dist * factor);  // in [-2.1, 0] for every sample that is used
        // The step's samples inside the disc are grouped by bin, each group in lane order = the reference's sample
        // order (ProgramCU.cu:1359: vote[bin] += gradient * weight): a sample's rank inside its group is the number of
        // lower lanes with the same bin, lane j (bin j) counts its group; an exclusive scan of the sizes places the groups
        // back to back in the wavefront's LDS list.  Lane j then reads and adds only its own group -- 8 bytes per
        // sample and lane instead of a 16-byte broadcast of every sample to all 64 lanes, which kept LDS, not
        // the vector ALUs, busy.
        const bool counted = inside & ((unsigned)bin < 36u);  // (a bin outside the histogram has no owner lane)
        // mask of the lanes whose key equals this lane's key, from the ballots of the six key bits (x == y bit by
        // bit: and of xnor); lanes without a sample carry key 63, which no bin owner has
        const int key = counted ? bin : 63;
        uint32_t same_lo = ~0u, same_hi = ~0u;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const uint64_t bk = __builtin_amdgcn_ballot_w64((key >> k) & 1);
          const uint32_t km = (uint32_t)-((key >> k) & 1);
          same_lo &= ~((uint32_t)bk ^ km); same_hi &= ~((uint32_t)(bk >> 32) ^ km);
        }
        const int myrank = __builtin_amdgcn_mbcnt_hi(same_hi, __builtin_amdgcn_mbcnt_lo(same_lo, 0u));
        // group sizes reach the bin owners through LDS: the last sample of a group knows its size
        const int gsize = __popc(same_lo) + __popc(same_hi);
        // (cross-lane exchanges through LDS inside one wavefront: DS operations of a wavefront execute in order; the
        // wave barriers only pin the compiler's ordering of the may-alias accesses, they emit no instruction)
        if (lane < 36) mysize[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (counted & (myrank + 1 == gsize)) mysize[bin] = gsize;
        __builtin_amdgcn_wave_barrier();
        const int mycnt = (lane < 36) ? mysize[lane] : 0;
        const int mystart = wave_inclusive_scan(mycnt) - mycnt;
        const int pos = __shfl(mystart, counted ? bin : 0) + myrank;
        __builtin_amdgcn_wave_barrier();  // the previous step's group reads precede this step's record stores
        if (counted) myrec[pos] = make_float2(gv.x, e);
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; __builtin_amdgcn_ballot_w64(k < mycnt) != 0; k += 2) {
          if (k < mycnt) {  // two records per trip (a group's records are adjacent; the list has a spare slot)
            const float2 r0 = myrec[mystart + k], r1 = myrec[mystart + k + 1];  // one ds_read2_b64
            vote = fmaf(r0.x, r0.y, vote);
            if (k + 1 < mycnt) vote = fmaf(r1.x, r1.y, vote);
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

// ================================= descriptor ================================================

// LDS coefficient table of the descriptor kernel: [bin 0..11][cell 0..15 (+pad)][sample slot s 0..3] floats per
// wavefront.  Column (cell, s) is all zeros except, while the samples of the current iteration are being
// accumulated, w1 at bin floor(theta_s) and w2 at the next bin.  The four slots of a (bin, cell) are adjacent, so a
// lane fetches the four coefficients of one of its bins with one 16-byte read.  Bin pitch 80 floats (64 + 16):
//   stores  a lane's bank is 16*(bin & 1) + 4*(cell mod 8) + s -- whatever bins the 32 lanes of a store group
//           hit, at most two of them share a bank (free for 4-byte stores); with the bins of a cell adjacent (the
//           first layout tried) lanes of equal bin collided 4-way and conflicts took half of all LDS cycles;
//   loads   the 16 lanes of a 16-byte read group (cells {0,3,5,6}, {1,2,4,7}, ... x q) fall on 16 different 4-bank
//           groups: 4*q + cell is distinct modulo 16 within every group.
constexpr int DC_BINS = 12;
constexpr int DC_BIN_PITCH = 80;
constexpr int DC_ROWS = DC_BINS * DC_BIN_PITCH;
// Dynamic LDS requested at launch on top of the static 17 KB.  Rounds 1-2 padded a workgroup to 36 KB so that at most
// four fit a CU (the kernel then measured 5-30 % slower at 5-8 wavefronts per SIMD).  With the all-miss skip and the
// features taken largest first that no longer holds: without padding (five workgroups per CU, the register limit)
// the launch takes 0.372 ms per 8 images against 0.405 ms with it, same box (profiles/r03_desc_pad.txt), and the
// unpadded workgroups leave LDS to the kernels of the other contexts.
constexpr int DC_LDS_PAD_BYTES = 0;

typedef const __attribute__((address_space(1))) char* GlobalBytes;  // byte pointer into global memory (HBM)
typedef float dfloat2 __attribute__((ext_vector_type(2)));

// N consecutive iterations of a lane of the descriptor kernel: window coordinates, window test, gathered (gradient, theta)
template <int N_>
struct DescChunk {
  static constexpr int N = N_;
  float nx[N_], ny[N_];
  float2 cc[N_];
  bool in[N_];
};

// acc += coef * (w of lane 4*(lane/4)+S of the caller's quad): quad broadcast by DPP, then one fused multiply-add.
// (The broadcast stays a compiler-visible v_mov_b32_dpp on purpose: folded into the multiply-add by inline
// assembly it saved nothing measurable, and the compiler cannot see the wait states a DPP read of a freshly
// written register needs inside an asm statement.)
template <int S>
__device__ __forceinline__ void quad_fma(float& acc, float coef, float w) {
  static_assert(S >= 0 && S < 4, "quad lane");
  const float wb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w), S * 0x55, 0xF, 0xF, true));
  acc = fmaf(coef, wb, acc);
}

// A launch that does only a part of one image's features (a single large image delivered in parts: the copier's DMA copy
// of part k crosses the host link while part k + 1 is computed): the image's list narrowed to features
// [n part / den, n (part + 1) / den) -- in list order, which is the order of the packed output, so a part is one contiguous
// range of keypoint records and descriptors (the copier thread forms the same bounds from the count it reads).
__device__ __forceinline__ void feature_part(const DescParams& dp, int* ftotal, int* ffirst, long long* obase) {
  if (dp.part_den > 1) {
    const int n = *ftotal;
    const int lo = (int)((long long)n * dp.part / dp.part_den), hi = (int)((long long)n * (dp.part + 1) / dp.part_den);
    *ffirst += lo; *obase += lo; *ftotal = hi - lo;
  }
}

// Host keypoint records (PyramidCU.cpp:866-906 / :1097-1137, host arithmetic) of one image, by the whole workgroup: one
// thread per feature, 256 per pass, staged in LDS (`kst`: 256 x 24 bytes, not otherwise in use yet) and stored as
// contiguous 8-byte pieces -- 24-byte records stored one by one from a lane of every wavefront reached the pinned host
// mirror as as many small PCIe writes and cost the kernel as much as the 512-byte descriptors did.
// (by the launch's LAST workgroups: the first ones hold the largest features and are the launch's critical path)
template <bool HOST_MIRROR>
__device__ __forceinline__ void keypoint_records(const DescParams& dp, uint32_t* const kst, const RawKey* list, int cap_list,
                                                 const FRec* recs, const int* fsrc, int cap_feat, int b, int ftotal, int ffirst,
                                                 long long obase, HostKeypoint* keys) {
  for (int f0 = (gridDim.x - 1 - blockIdx.x) * 256; f0 < ftotal; f0 += gridDim.x * 256) {  // (uniform over the workgroup)
    const int nrec = min(256, ftotal - f0);
    if ((int)threadIdx.x < nrec) {
      const int m = ffirst + f0 + threadIdx.x;
      const int src = fsrc[(long long)b * cap_feat + m];
      const int i = src >> 2, k = src & 3;
      const FRec rec = recs[(long long)b * cap_list + i];
      const int li = list[(long long)b * cap_list + i].level_index;
      const float kw = dp.multi ? (float)((2.0 * kPI / 255.0) * (double)((rec.w >> (8 * k)) & 0xFFu))
                                : __uint_as_float(rec.w);
      const float kx = (float)(rec.x & 0x00FFFFFFu) / 1024.0f;
      const float ky = (float)(rec.y & 0x00FFFFFFu) / 1024.0f;
      const float kz = (float)(rec.z & 0x0000FFFFu) / 256.0f;
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
      *reinterpret_cast<HostKeypoint*>(kst + 6 * threadIdx.x) = hk;
    }
    __syncthreads();
    const uint2* const k2 = reinterpret_cast<const uint2*>(kst);
    uint2* const out = reinterpret_cast<uint2*>(keys + obase + f0);
    uint2* const hout = (HOST_MIRROR && dp.hkeys) ? reinterpret_cast<uint2*>(dp.hkeys + obase + f0) : nullptr;
    for (int j = threadIdx.x; j < 3 * nrec; j += 256) {
      const uint2 v = k2[j];
      out[j] = v;
      if (hout) hout[j] = v;
    }
    __syncthreads();
  }
}

// Normalisation with the reference's 32-lane tree (NormalizeDescriptor_Kernel + ND_WarpReduction, ProgramCU.cu:1950-2054)
// and the 512-byte (256-byte) coalesced store of one descriptor; lanes 0..31 hold four (two) consecutive values each.
template <bool HOST_MIRROR>
__device__ __forceinline__ void finish_descriptor128(const DescParams& dp, int lane, float4 v, float* dout, float* hout) {
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
  // (streaming stores for the mirror were measured: 0.341 - 0.346 against 0.333 - 0.339 ms per single image, same call)
  if (HOST_MIRROR && hout && lane < 32) *reinterpret_cast<float4*>(hout + lane * 4) = v;
}
template <bool HOST_MIRROR>
__device__ __forceinline__ void finish_descriptor64(const DescParams& dp, int lane, float2 v, float* dout, float* hout) {
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
  if (HOST_MIRROR && hout && lane < 32) *reinterpret_cast<float2*>(hout + lane * 2) = v;
}

// One wavefront per feature.  Lane = cell*4 + sub.
//   scan   the four lanes of a cell take four consecutive samples of the cell's box per iteration (scan order of
//          ProgramCU.cu:1723-1774: y outer, x inner), so the 16 cells advance together; a lane steps its own
//          sample four places along the scan by exact float increments (no index division), evaluates the
//          reference's window test, weights and bin coordinate for it (a sample outside the window or past the box
//          gets weight 0);
//   add    lane (cell, q) owns the accumulators des[q], des[q+4], des[q+8] of its cell (des[8] is the reference's
//          ninth bin, des[9..11] stay unused).  The samples of an iteration are added in scan order s = 0..3: the
//          lane that evaluated sample s has put its two bin coefficients (w1 at floor(theta), w2 at the next bin)
//          into an otherwise zero LDS row; every lane of the cell reads the three coefficients of its own bins and
//          does `des += coefficient * weight` -- the reference's `des[fidx] += w1*weight; des[fidx+1] += w2*weight`
//          for the two bins that are touched and `+= 0 * weight` (no change: weights and sums are finite and
//          non-negative) for the others.  Per bin the additions therefore happen in the reference's order.
// No sample lists, no compaction: per iteration a lane does two 2-dword LDS stores (set, clear) and three 16-byte loads.
// Occupancy: five workgroups (= five wavefronts per SIMD) per CU by the register count (see DC_LDS_PAD_BYTES for the
// history: rounds 1-2 held it at four through the LDS footprint).
// HOST_MIRROR: the packed results are also stored into their pinned host mirrors (dp.hkeys / dp.hdesc); a template
// parameter so that the two forms carry different names in profiles (alone on the device the mirroring form waits
// for PCIe).
template <bool HOST_MIRROR, bool SEQ>
__global__ __launch_bounds__(256) void descriptor_kernel(Geom g, DescParams dp, const RawKey* list,
                                                         int cap_list, const FRec* recs,
                                                         const int* fsrc, const int* feat_total,
                                                         const int* feat_first, const int* img_base,
                                                         const float* got, HostKeypoint* keys, float* desc,
                                                         int cap_feat) {
  __shared__ __attribute__((aligned(16))) float dl[4][128];
  __shared__ __attribute__((aligned(16))) float crow[4][DC_ROWS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y + dp.first_image;  // (a batch's descriptors may be launched in two halves, see enqueue())
  int ftotal = feat_total[b], ffirst = feat_first[b];
  long long obase = img_base[b];  // packed output: images of the batch back to back
  feature_part(dp, &ftotal, &ffirst, &obase);
  const int nwaves = gridDim.x * 4;
  const float rpi = (float)(4.0 / kPI);
  const int dim = dp.half_sift ? 64 : 128;
  const int mycell = lane >> 2, sub = lane & 3;
  float* const rows = &crow[wv][0];

  static_assert(sizeof(HostKeypoint) == 24 && 4 * DC_ROWS * 4 >= 256 * 24, "record staging fits the table");
  keypoint_records<HOST_MIRROR>(dp, reinterpret_cast<uint32_t*>(&crow[0][0]), list, cap_list, recs, fsrc, cap_feat, b, ftotal, ffirst, obase, keys);
  if (!desc) return;
  for (int i = lane; i < DC_ROWS; i += 64) rows[i] = 0.0f;
  float* const mycol = rows + mycell * 4 + sub;  // + DC_BIN_PITCH*bin: the column this lane fills (cell, slot `sub`)
  const float* const rdbin = rows + sub * DC_BIN_PITCH + mycell * 4;  // + 4*DC_BIN_PITCH*k: bins sub, sub+4, sub+8
  const uint32_t theta_end_bits = dp.dynamic_indexing ? 0x41000001u : 0x41000000u;  // 8.0f, or the next float (admits theta == 8)

  // Features are taken from the END of the list backwards: the list is ordered by (octave, level), most features
  // belong to octave 0, and within an octave the footprint grows with the level, so the launch's last wavefronts --
  // its tail -- get the smallest features (octave 0, level 1) instead of a mixture.
  // ... and in blocks of 64 consecutive features per XCD: workgroups are dealt round-robin over the eight XCDs, each with
  // its own L2, while features that are neighbours in the list are neighbours in the image (raster order within a
  // level) and read overlapping footprints -- so a block of the list goes to the workgroups of ONE XCD instead of
  // being spread over all eight L2s (dp.xcd_block: 64, or 0 for the plain order when the grid does not divide).
  int mw0 = blockIdx.x * 4 + wv;
  if (dp.xcd_block) {
    const int xcd = blockIdx.x & 7, wx = (int)(blockIdx.x >> 3) * 4 + wv;  // wavefront index inside the XCD
    mw0 = ((wx / dp.xcd_block) * 8 + xcd) * dp.xcd_block + wx % dp.xcd_block;
  }
  for (int mw = mw0; mw < ftotal; mw += nwaves) {
    const int m = ffirst + ftotal - 1 - mw;
    const int src = fsrc[(long long)b * cap_feat + m];
    const int i = src >> 2, k = src & 3;
    const int oidx = m - ffirst;
    const FRec rec = recs[(long long)b * cap_list + i];
    const int li = list[(long long)b * cap_list + i].level_index;
    int o, l;
    level_of(g, li, &o, &l);
    const OctGeom& og = g.o[o];
    // the feature, and with it the gradient plane, is the same in all lanes: keep the plane's address in scalar
    // registers so that a gather is `scalar base + 32-bit lane offset`
    const unsigned long long gpa = (unsigned long long)(reinterpret_cast<const float2*>(got) + og.got_off +
                                                        ((long long)(l - 1) * g.B + b) * og.plane);
    // and gather through a buffer resource over the plane: `resource + 32-bit byte offset`, no 64-bit address
    // arithmetic per sample, and an offset outside the plane reads 0 instead of faulting.  (A generic pointer here
    // made the compiler emit flat_load, which also counts as an LDS operation: every wait for the coefficient
    // table then waited for the gathers too.)
    const GlobalBytes gp = (GlobalBytes)(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gpa >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)(gpa & 0xFFFFFFFFull)));
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)gp, 0, __builtin_amdgcn_readfirstlane(og.plane * 8), 0x00020000 /* raw 32-bit data, gfx9 family */);
    const int width = og.wa, height = og.h;

    // un-mirrored orientation handed to the kernel (PyramidCU.cpp:764,791; A.1 of SURVEY)
    const float kw = dp.multi ? (float)((2.0 * kPI / 255.0) * (double)((rec.w >> (8 * k)) & 0xFFu))
                              : __uint_as_float(rec.w);
    const float kx = (float)(rec.x & 0x00FFFFFFu) / 1024.0f;
    const float ky = (float)(rec.y & 0x00FFFFFFu) / 1024.0f;
    const float kz = (float)(rec.z & 0x0000FFFFu) / 256.0f;


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
    int maxtotal = total;
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) maxtotal = max(maxtotal, __shfl_xor(maxtotal, d));
    maxtotal = rli(maxtotal, 0);
    const int nit = (maxtotal + 3) >> 2;
    // A lane's sample t = 4*iteration + sub sits at pixel centre (xf, yf) = (xmin + t % nxs, ymin + t / nxs), byte
    // offset `goff` in the gradient plane.  Rows of four or more samples (every box of a real keypoint) are
    // walked by increments: four places along x, and back by a row's length onto the next row when that passes
    // xmax -- all values are integers + 0.5 far below 2^23, so the float steps are exact.  Boxes narrower than
    // four samples (degenerate scales) take the division form; the choice is uniform over the wavefront.
    const bool narrow = __any((nxs > 0) & (nxs < 4));
    const float fnxs = (float)nxs;
    const float inv = 1.0f / (float)(nxs > 0 ? nxs : 1);
    const int base = (int)ymin * width + (int)xmin;  // (int)y * width + (int)x of sample (0, 0)
    float xf = xmin + (float)sub;
    float yf = (total > 0) ? ymin : 3.0e38f;  // an empty box never yields a valid sample
    unsigned goff = (unsigned)(base + sub) * 8u;
    const float xwrap = xmax - 4.0f;           // stepping from beyond this lands past the row's end
    const int rowskip = (width - nxs + 4) * 8;  // byte step onto the next row instead of +32

    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;  // des[sub], des[sub+4], des[sub+8]

    // stage A: window test and the gradient gathers of N iterations, issued back to back.  Iterations past the end
    // of the scan are harmless: their samples fail the window test (yf > ymax) and gather from offset 0.
    auto stage_a = [&](auto narrow_tag, int it0, auto& ck) {
      constexpr bool NARROW = decltype(narrow_tag)::value;
      constexpr int N = std::remove_reference_t<decltype(ck)>::N;
#pragma unroll
      for (int u = 0; u < N; u++) {
        if (NARROW) {
          const int t = (it0 + u) * 4 + sub;
          const int sy = (int)(((float)t + 0.5f) * inv);  // = t / nxs (exact: |error| << 0.5/nxs)
          const int sx = t - __mul24(sy, nxs);
          xf = xmin + (float)sx;
          yf = (t < total) ? ymin + (float)sy : 3.0e38f;
          goff = (unsigned)(base + __mul24(sy, width) + sx) * 8u;
        }
        const float dx = xf - ptx, dy = yf - pty;
        ck.nx[u] = fmaf(crspt, dx, srspt * dy);
        ck.ny[u] = fmaf(crspt, dy, -(srspt * dx));
        ck.in[u] = (yf <= ymax) & (fabsf(ck.nx[u]) < 1.0f) & (fabsf(ck.ny[u]) < 1.0f);
        const dfloat2 gv = __builtin_amdgcn_raw_buffer_load_b64(grsrc, (int)(ck.in[u] ? goff : 0u), 0, 0);
        ck.cc[u] = make_float2(gv.x, gv.y);
        if (!NARROW) {
          const bool wrap = xf > xwrap;
          xf += wrap ? 4.0f - fnxs : 4.0f;
          yf += wrap ? 1.0f : 0.0f;
          goff += wrap ? (unsigned)rowskip : 32u;
        }
      }
    };
    // stage B: weight and bin coordinate of the lane's own sample, then the ordered accumulation of the
    // iteration's four samples into every lane's bins
    // the four samples of one iteration, in scan order, into this lane's three bins
    auto accumulate = [&](const float4& c0, const float4& c1, const float4& c2, float wt) {
      quad_fma<0>(acc0, c0.x, wt); quad_fma<0>(acc1, c1.x, wt); quad_fma<0>(acc2, c2.x, wt);
      quad_fma<1>(acc0, c0.y, wt); quad_fma<1>(acc1, c1.y, wt); quad_fma<1>(acc2, c2.y, wt);
      quad_fma<2>(acc0, c0.z, wt); quad_fma<2>(acc1, c1.z, wt); quad_fma<2>(acc2, c2.z, wt);
      quad_fma<3>(acc0, c0.w, wt); quad_fma<3>(acc1, c1.w, wt); quad_fma<3>(acc2, c2.w, wt);
    };
    auto stage_b = [&](const auto& ck) {
      constexpr int N = std::remove_reference_t<decltype(ck)>::N;
      auto one = [&](int u) {
        const float nxn = fabsf(ck.nx[u]), nyn = fabsf(ck.ny[u]);
        const float dnx = ck.nx[u] + offx, dny = ck.ny[u] + offy;
        const float ww = dm_expf_inrange(-0.125f * fmaf(dnx, dnx, dny * dny));
        const float wx = 1.0f - nxn, wy = 1.0f - nyn;
        float wt = ww * wx * wy * ck.cc[u].x;
        float theta = (anglef - ck.cc[u].y) * rpi;
        theta = (theta < 0) ? theta + 8.0f : theta;
        // DYNAMIC_INDEXING=false: a sample with floor(theta) == 8 adds nothing (ProgramCU.cu:1763-1771);
        // with -di it adds w1*weight = weight to des[8] (:1755-1759; the write to des[9] adds 0)
        // 0 <= theta < theta_end as ONE unsigned compare of the bit patterns (negative values and NaNs have larger
        // patterns than any non-negative bound); theta is never -0: the angles subtracted above are not
        const bool hit = ck.in[u] & (__float_as_uint(theta) < theta_end_bits);
        wt = hit ? wt : 0.0f;
        const float fo = floorf(theta);
        const float w1 = fo + 1.0f - theta, w2 = theta - fo;  // ProgramCU.cu:1752-1753
        const int fidx = min(max((int)fo, 0), DC_BINS - 2);  // 0..8 for every finite theta; never outside the table
        if (!SEQ) {
          // HESS_DESC_ORDER_INTERLEAVED: the lane adds its own sample into its own 12-bin histogram in LDS
          // ([bin][lane]: conflict-free); the four lanes of a cell are summed in a fixed order after the scan.  No
          // coefficient table, no quad broadcasts: 57 instead of 71 vector instructions per iteration.  A sample
          // that is not `hit` has weight 0 and leaves both sums as they are.
          float* const hp = rows + fidx * 64 + lane;
          const float a = hp[0], b2 = hp[64];
          hp[0] = fmaf(w1, wt, a);
          hp[64] = fmaf(w2, wt, b2);
          return;
        }
        mycol[fidx * DC_BIN_PITCH] = w1;
        mycol[fidx * DC_BIN_PITCH + DC_BIN_PITCH] = w2;
        __builtin_amdgcn_wave_barrier();  // cross-lane through LDS inside the wavefront: pins the compiler's order only
        const float4 c0 = *reinterpret_cast<const float4*>(rdbin);
        const float4 c1 = *reinterpret_cast<const float4*>(rdbin + 4 * DC_BIN_PITCH);
        const float4 c2 = *reinterpret_cast<const float4*>(rdbin + 8 * DC_BIN_PITCH);
        __builtin_amdgcn_wave_barrier();
        accumulate(c0, c1, c2, wt);
        mycol[fidx * DC_BIN_PITCH] = 0.0f;
        mycol[fidx * DC_BIN_PITCH + DC_BIN_PITCH] = 0.0f;
        __builtin_amdgcn_wave_barrier();
      };
#pragma unroll
      for (int u = 0; u < N; u++) {
        // The sixteen cells walk their boxes in step and have the same shape, so their misses coincide: in about
        // one iteration of six no lane of the wavefront has a sample inside its window.  Such an iteration adds
        // coefficient * 0 everywhere; skipping it changes no bit.
        if (!__any(ck.in[u])) continue;
        one(u);
      }
    };
    if (narrow) {  // degenerate scales only: one iteration at a time
      for (int it0 = 0; it0 < nit; it0++) {
        DescChunk<1> ck;
        stage_a(std::true_type{}, it0, ck);
        stage_b(ck);
      }
    } else {
      constexpr int UN = 4;
      // software pipeline: the gathers of the next chunk are in flight while the current chunk is accumulated
      DescChunk<UN> ca, cb;
      stage_a(std::false_type{}, 0, ca);
      for (int it0 = 0; it0 < nit; it0 += 2 * UN) {
        stage_a(std::false_type{}, it0 + UN, cb);
        stage_b(ca);
        if (it0 + UN >= nit) break;
        stage_a(std::false_type{}, it0 + 2 * UN, ca);
        stage_b(cb);
      }
    }
    if (!SEQ) {  // des[bin] = (p0 + p1) + (p2 + p3) over the cell's four lanes; then the histograms are cleared again
      __builtin_amdgcn_wave_barrier();
      const float* const hb = rows + mycell * 4;
      const float4 h0 = *reinterpret_cast<const float4*>(hb + sub * 64);
      const float4 h1 = *reinterpret_cast<const float4*>(hb + (sub + 4) * 64);
      const float4 h2 = *reinterpret_cast<const float4*>(hb + (sub + 8) * 64);
      acc0 = (h0.x + h0.y) + (h0.z + h0.w);
      acc1 = (h1.x + h1.y) + (h1.z + h1.w);
      acc2 = (h2.x + h2.y) + (h2.z + h2.w);
      __builtin_amdgcn_wave_barrier();
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(rows + mycell * 4 + sub * 64) = z4;
      *reinterpret_cast<float4*>(rows + mycell * 4 + (sub + 4) * 64) = z4;
      *reinterpret_cast<float4*>(rows + mycell * 4 + (sub + 8) * 64) = z4;
      __builtin_amdgcn_wave_barrier();
    }
    if (sub == 0) acc0 += acc2;  // des[0] += des[8], ProgramCU.cu:1776
    if (dp.half_sift) {
      dl[wv][mycell * 4 + sub] = acc0 + acc1;  // des[k] += des[k+4], ProgramCU.cu:1782-1785
    } else {
      dl[wv][mycell * 8 + sub] = acc0;
      dl[wv][mycell * 8 + 4 + sub] = acc1;
    }
    // same wavefront wrote dl[wv]; LDS operations of one wavefront complete in order
    __builtin_amdgcn_wave_barrier();
    float* dout = desc + (obase + oidx) * dim;
    float* hout = (HOST_MIRROR && dp.hdesc) ? dp.hdesc + (obase + oidx) * dim : nullptr;
    if (dp.half_sift) {
      float2 v = make_float2(0, 0);
      if (lane < 32) v = *reinterpret_cast<const float2*>(&dl[wv][lane * 2]);
      finish_descriptor64<HOST_MIRROR>(dp, lane, v, dout, hout);
    } else {
      float4 v = make_float4(0, 0, 0, 0);
      if (lane < 32) v = *reinterpret_cast<const float4*>(&dl[wv][lane * 4]);
      finish_descriptor128<HOST_MIRROR>(dp, lane, v, dout, hout);
    }
  }
}

// ================================= descriptor, pixel raster ===================================
//
// HESS_DESC_ORDER_PIXEL (include/hess_abi.h; restated by the test oracle, compute_descriptor_pixel): one wavefront
// per feature rasters the rotated 5 x 5-cell footprint ONCE, 64 pixels per step, one per lane (the pixels of the
// footprint's rows inside its bounding box: "row spans" in the kernel).
// In the keypoint frame (u, v) = R(-angle)(pixel - keypoint) / spt everything the reference recomputes per (pixel,
// cell) pair is a per-pixel quantity -- the gather, the Gaussian weight exp(-(u^2 + v^2)/8), the bin coordinate theta
// and its split, and the bilinear cell weights (the split of u + 1.5, v + 1.5 between the two nearest cell indices) --
// so a pixel costs one evaluation instead of up to four (2.56 on average), and then adds its weight to <= 2 x 2 cells x
// 2 bins.  descriptor_kernel above spends 5.6 k vector instructions per feature (80 live iterations of 57).
// The scatter is deterministic without any ordering: the sums are 32-bit FIXED POINT with a per-feature power-of-two
// scale, every product is rounded to an integer before it is added (uint32(fma(b, w, 0.5))), and integer addition is
// associative -- so LDS integer atomics give the same bits for every schedule, and the oracle's plain loop over the
// pixels gives them too.  (ds_add_u64: 5.6 LDS cycles per wave-instruction whatever the lanes, tools/micro/
// lds_atomic_int.hip; LDS FLOAT atomics take 170 - 225 cycles, lds_atomic.hip.)  The two bins a pixel touches in a
// cell are neighbours (b0 = floor(theta) and the next, modulo 8): a cell keeps eight 64-bit words, word b0 = [bin b0 |
// bin b0 + 1 mod 8], so ONE ds_add_u64 adds both (the low half cannot carry into the high one: the scale keeps every
// sum below 2^32, see the oracle); bin k is the low half of word k plus the high half of word k - 1.  Four atomics per
// pixel.
// Lanes that hit the same word serialise (about 3.3 cycles per extra lane), and neighbouring pixels do (coherent
// gradients): a wavefront keeps PX_COPIES copies of the 128 words and a lane adds into copy (lane mod PX_COPIES); the
// copies are summed when the raster is done.
// Layout per wavefront: [copy][cell 0..15][word 0..7] 64-bit words, copies PX_COPY_U64 words apart (1 KB + 32 bytes:
// the copies of one word fall on different banks).
#ifndef HESS_PX_COPIES
#define HESS_PX_COPIES 4
#endif
#ifndef HESS_PX_UNROLL
#define HESS_PX_UNROLL 2
#endif
constexpr int PX_COPIES = HESS_PX_COPIES;  // (A/B builds: -DHESS_PX_COPIES=2|8, -DHESS_PX_UNROLL=1|3; same call, descriptor ms per step of 8: 2 copies 0.230, 4: 0.226, 8: 0.269 -- with eight the LDS footprint holds the kernel at four wavefronts per SIMD -- 16: 0.50)
constexpr int PX_COPY_U64 = 128 + 4;
constexpr int PX_WAVE_U64 = PX_COPIES * PX_COPY_U64;
#ifndef HESS_PX_WAVES
#define HESS_PX_WAVES 7
#endif
constexpr int PX_WAVES = HESS_PX_WAVES;  // wavefronts per SIMD the register allocation aims at (= workgroups per CU the LDS admits)
constexpr float PX_SPAN_EPS = 0.02f;
constexpr int PX_BAND_MAX = 4096;  // pixels of a raster band (dp.px_band) at most: 64 steps, one word of row-start bits per lane

// N consecutive steps of a lane: keypoint-frame coordinates (u = 3 marks a step outside the window or past the box),
// gathered (gradient, theta)
template <int N_>
struct PixChunk {
  static constexpr int N = N_;
  float u[N_], v[N_];
  float2 cc[N_];
};

typedef __attribute__((address_space(3))) unsigned long long lds_u64;

// (amdgpu_waves_per_eu(7, 7): 23 040 bytes of LDS per workgroup admit seven workgroups per CU; left alone the register
// allocator takes 77 registers = six wavefronts per SIMD.  71 registers, nothing spilled, launch 86.6 -> 85.1 us.)
template <bool HOST_MIRROR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PX_WAVES, PX_WAVES))) void descriptor_pixel_kernel(Geom g, DescParams dp, const RawKey* list,
                                                               int cap_list, const FRec* recs,
                                                               const int* fsrc, const int* feat_total,
                                                               const int* feat_first, const int* img_base,
                                                               const float* got, HostKeypoint* keys, float* desc,
                                                               int cap_feat) {
  __shared__ __attribute__((aligned(16))) float dl[4][128];
  __shared__ __attribute__((aligned(16))) unsigned long long hist[4][PX_WAVE_U64];
  __shared__ __attribute__((aligned(16))) uint4 rowtab[4][64];  // per wavefront: the rows of the current raster band
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y + dp.first_image;
  int ftotal = feat_total[b], ffirst = feat_first[b];
  long long obase = img_base[b];
  feature_part(dp, &ftotal, &ffirst, &obase);
  const int nwaves = gridDim.x * 4;
  const float rpi = (float)(4.0 / kPI);
  const int dim = dp.half_sift ? 64 : 128;

  static_assert(sizeof(HostKeypoint) == 24 && sizeof(hist) >= 256 * 24, "record staging fits the sums");
  keypoint_records<HOST_MIRROR>(dp, reinterpret_cast<uint32_t*>(&hist[0][0]), list, cap_list, recs, fsrc, cap_feat, b, ftotal, ffirst, obase, keys);
  if (!desc) return;
  unsigned long long* const sums = &hist[wv][0];
  {
    uint4* const z = reinterpret_cast<uint4*>(sums);
    for (int i = lane; i < PX_WAVE_U64 / 2; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  // LDS byte address of the lane's copy, as a float (stage B adds the word's offset in floating point)
  const float mycopy_f = (float)(unsigned)(unsigned long long)(lds_u64*)(sums + (lane % PX_COPIES) * PX_COPY_U64);
  const uint32_t theta_end_bits = dp.dynamic_indexing ? 0x41000001u : 0x41000000u;  // 8.0f, or the next float (admits theta == 8)

  // feature order: as descriptor_kernel (largest footprints first, blocks of consecutive features per XCD)
  int mw0 = blockIdx.x * 4 + wv;
  if (dp.xcd_block) {
    const int xcd = blockIdx.x & 7, wx = (int)(blockIdx.x >> 3) * 4 + wv;
    mw0 = ((wx / dp.xcd_block) * 8 + xcd) * dp.xcd_block + wx % dp.xcd_block;
  }
  for (int mw = mw0; mw < ftotal; mw += nwaves) {
    const int m = ffirst + ftotal - 1 - mw;
    const int src = fsrc[(long long)b * cap_feat + m];
    const int i = src >> 2, k = src & 3;
    const int oidx = m - ffirst;
    const FRec rec = recs[(long long)b * cap_list + i];
    const int li = list[(long long)b * cap_list + i].level_index;
    int o, l;
    level_of(g, li, &o, &l);
    const OctGeom& og = g.o[o];
    const unsigned long long gpa = (unsigned long long)(reinterpret_cast<const float2*>(got) + og.got_off +
                                                        ((long long)(l - 1) * g.B + b) * og.plane);
    const GlobalBytes gp = (GlobalBytes)(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gpa >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)(gpa & 0xFFFFFFFFull)));
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)gp, 0, __builtin_amdgcn_readfirstlane(og.plane * 8), 0x00020000 /* raw 32-bit data, gfx9 family */);
    const int width = og.wa, height = og.h;

    const float kw = dp.multi ? (float)((2.0 * kPI / 255.0) * (double)((rec.w >> (8 * k)) & 0xFFu))
                              : __uint_as_float(rec.w);
    const float kx = (float)(rec.x & 0x00FFFFFFu) / 1024.0f;
    const float ky = (float)(rec.y & 0x00FFFFFFu) / 1024.0f;
    const float kz = (float)(rec.z & 0x0000FFFFu) / 256.0f;
    const float spt = fabsf(kz * dp.window_factor);
    float s, c;
    dm_sincosf(kw, &s, &c);
    const float anglef = (kw > kPI) ? (float)(kw - (2.0 * kPI)) : kw;
    const float cspt = c * spt, sspt = s * spt;
    const float crspt = c / spt, srspt = s / spt;
    const float bsz = fabsf(cspt) + fabsf(sspt);
    const float ext = 2.5f * bsz;  // half extent of the footprint's bounding box
    const float xmin = fmaxf(1.5f, floorf(kx - ext) + 0.5f);
    const float ymin = fmaxf(1.5f, floorf(ky - ext) + 0.5f);
    const float xmax = fminf(width - 1.5f, floorf(kx + ext) + 0.5f);
    const float ymax = fminf(height - 1.5f, floorf(ky + ext) + 0.5f);
    // fixed-point scale 2^sh: 0.75 (spt + 1)^2 < 2^e bounds every sum, sh = 32 - e (the oracle's frexpf)
    const float bound = 0.75f * (spt + 1.0f) * (spt + 1.0f);
    const int sh = min(max(32 - ((int)((__float_as_uint(bound) >> 23) & 0xFFu) - 126), 0), 30);
    const float scale = __uint_as_float((uint32_t)(127 + sh) << 23), rscale = __uint_as_float((uint32_t)(127 - sh) << 23);
    // (the same in every lane: the box is the feature's)
    const int nxs = __builtin_amdgcn_readfirstlane((xmax >= xmin) ? (int)(xmax - xmin) + 1 : 0);
    const int nys = __builtin_amdgcn_readfirstlane((ymax >= ymin) ? (int)(ymax - ymin) + 1 : 0);
    // Row spans.  The window |u| < 2.5, |v| < 2.5 is a rotated square: of the box's pixels 1 / (|c| + |s|)^2 lie inside
    // (0.61 on average over the angles), so the raster runs over the window's own rows instead: row y of the box keeps
    // the pixels x_lo(y) .. x_hi(y), the real-arithmetic solution of the two inequalities for x, slightly widened (a
    // SUPERSET of the pixels that pass the test: every pixel is still tested with the floats the oracle uses, so which
    // pixels count does not depend on the spans; integer sums do not depend on the order either).
    // Bands of <= 64 rows (lane = row) and <= dp.px_band pixels: prefix sums of the span lengths give every row its first
    // place S in the band's sequence; a 64-bit word per step holds the places where rows start (bit S mod 64 of word
    // S / 64, lane w keeps word w), so the lane of place t = 64 step + lane finds its row with two mbcnt and reads the
    // row's entry (first pixel's index - S, x - S, y) from LDS.
    // (a coefficient below 1e-3 per pixel: that inequality is left out -- it cuts the corners of the box only -- so that the
    // rounding of u, v, 1e-6 at most, stays below 1e-3 pixel in x; the spans are widened by PX_SPAN_EPS = 0.02)
    const float rA = (fabsf(crspt) > 1.0e-3f) ? 1.0f / crspt : 0.0f, rB = (fabsf(srspt) > 1.0e-3f) ? 1.0f / srspt : 0.0f;
    const int px_band = __builtin_amdgcn_readfirstlane(dp.px_band);
    for (int ib = 0; ib < nxs; ib += px_band) {  // (column bands: a box wider than a band -- no detected feature's is)
    const int ncol = min(nxs - ib, px_band);
    const int band_rows = min(64, px_band / ncol);
    for (int jb = 0; jb < nys; jb += band_rows) {
    const int nrow = min(band_rows, nys - jb);
    int T;
    uint32_t mword_lo, mword_hi;
    {
      const float yrow = ymin + (float)(jb + lane), dyr = yrow - ky;
      float lo = -3.0e38f, hi = 3.0e38f;
      if (rA != 0.0f) {  // |crspt dx + srspt dy| < 2.5
        const float t1 = (-2.5f - srspt * dyr) * rA, t2 = (2.5f - srspt * dyr) * rA;
        lo = fminf(t1, t2); hi = fmaxf(t1, t2);
      }
      if (rB != 0.0f) {  // |crspt dy - srspt dx| < 2.5
        const float t1 = (crspt * dyr - 2.5f) * rB, t2 = (crspt * dyr + 2.5f) * rB;
        lo = fmaxf(lo, fminf(t1, t2)); hi = fminf(hi, fmaxf(t1, t2));
      }
      const float off = kx - xmin;  // pixel i of the row: x = xmin + i, dx = i - off
      const float flo = fmaxf(ceilf(lo + off - PX_SPAN_EPS), (float)ib), fhi = fminf(floorf(hi + off + PX_SPAN_EPS), (float)(ib + ncol - 1));
      const int len = (lane < nrow && fhi >= flo) ? (int)(fhi - flo) + 1 : 0;
      const int ilo = (int)flo;
      const int incl = wave_inclusive_scan(len);
      const int S = incl - len;
      T = __builtin_amdgcn_readlane(incl, 63);
      const uint64_t ne = __builtin_amdgcn_ballot_w64(len > 0);
      const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ne >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ne, 0u));
      // (dl[wv] is also the float staging area of the feature's end: wavefront fences keep the compiler from moving the
      // 64-bit accesses across the float ones, which type-based alias analysis would allow; they emit no instruction)
      unsigned long long* const starts = reinterpret_cast<unsigned long long*>(&dl[wv][0]);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      starts[lane] = 0ull;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (len > 0) {
        (void)__hip_atomic_fetch_or(starts + (S >> 6), 1ull << (S & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        rowtab[wv][r] = make_uint4((uint32_t)(((int)ymin + jb + lane) * width + (int)xmin + ilo - S),
                                   __float_as_uint(xmin + (float)(ilo - S)), __float_as_uint(yrow), 0u);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const unsigned long long mw = starts[lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      mword_lo = (uint32_t)mw; mword_hi = (uint32_t)(mw >> 32);
    }
    const int nit = (T + 63) >> 6;
    int step = 0, rows_before = -1;  // (rows that started before this step's 64 places) - 1
    int tl = lane;
    float tf = (float)lane;

    // stage A: the lane's pixel of N steps (row by the start bits, place in the row), keypoint-frame coordinates, window
    // test and gather, issued back to back
    auto stage_a = [&](auto& ck) {
      constexpr int N = std::remove_reference_t<decltype(ck)>::N;
#pragma unroll
      for (int q = 0; q < N; q++) {
        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)mword_lo, step), mhi = (uint32_t)__builtin_amdgcn_readlane((int)mword_hi, step);
        const uint64_t m64 = ((uint64_t)mhi << 32) | mlo;
        int rr = rows_before + (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) + (__builtin_amdgcn_inverse_ballot_w64(m64) ? 1 : 0);
        rows_before += __builtin_popcountll(m64);
        rr = (int)min((unsigned)rr, 63u);  // (steps past the band's end: any entry, the lane is switched off below)
        const uint4 e = rowtab[wv][rr];
        const float xf = __uint_as_float(e.y) + tf, yf = __uint_as_float(e.z);
        const unsigned goff = (e.x + (unsigned)tl) * 8u;
        const float dx = xf - kx, dy = yf - ky;
        const float u = fmaf(crspt, dx, srspt * dy);
        ck.v[q] = fmaf(crspt, dy, -(srspt * dx));
        const bool in = (tl < T) & (fabsf(u) < 2.5f) & (fabsf(ck.v[q]) < 2.5f);
        ck.u[q] = in ? u : 3.0f;  // (outside the window)
        const dfloat2 gv = __builtin_amdgcn_raw_buffer_load_b64(grsrc, (int)(in ? goff : 0u), 0, 0);
        ck.cc[q] = make_float2(gv.x, gv.y);
        tl += 64; tf += 64.0f; step++;
      }
    };
    // stage B: the pixel's weight, bin and cell split; four 64-bit additions of two fixed-point values each.
    // (The word's LDS address is formed in floating point from the three floors -- small integers, exact -- and
    // converted once; the four cells' validity masks are combined as wave masks on the scalar unit.)
    auto stage_b = [&](const auto& ck) {
      constexpr int N = std::remove_reference_t<decltype(ck)>::N;
#pragma unroll
      for (int q = 0; q < N; q++) {
        const float u = ck.u[q], v = ck.v[q];
        if (!__any(u < 2.5f)) continue;  // (wave-uniform) a step wholly outside the window: the box's corners
        const float ww = dm_expf_inrange(-0.125f * fmaf(u, u, v * v));
        float theta = (anglef - ck.cc[q].y) * rpi;
        theta = (theta < 0) ? theta + 8.0f : theta;
        // 0 <= theta < theta_end as ONE unsigned compare of the bit patterns (see descriptor_kernel)
        const uint64_t m_hit = __builtin_amdgcn_ballot_w64((u < 2.5f) & (__float_as_uint(theta) < theta_end_bits));
        // b0 = floor(theta), 0..7; theta == 8 (-di only) counts as b0 = 7 with weights (0, 1): the same sums, since word
        // 7 = [bin 7 | bin 0] (the oracle says bin 0 += weight, bin 1 += 0)
        const float fo = fminf(floorf(theta), 7.0f);
        const float wb1 = theta - fo, wb0 = 1.0f - wb1;
        const float au = u + 1.5f, av = v + 1.5f;
        const float fu = floorf(au), fv = floorf(av);  // -1 .. 3: cells fu, fu + 1 / fv, fv + 1 where they exist
        const float wx1 = au - fu, wx0 = 1.0f - wx1;
        const float wy1 = av - fv, wy0 = 1.0f - wy1;
        const float wt = (ww * ck.cc[q].x) * scale;
        const float a0 = wt * wy0, a1 = wt * wy1;
        const float b00 = a0 * wx0, b01 = a0 * wx1, b10 = a1 * wx0, b11 = a1 * wx1;
        const unsigned addr = (unsigned)(int)fmaf(fv, 256.0f, fmaf(fu, 64.0f, fmaf(fo, 8.0f, mycopy_f)));  // byte address in LDS
        lds_u64* const p = (lds_u64*)(unsigned long long)addr;
        const uint64_t mx0 = __builtin_amdgcn_ballot_w64(fu >= 0.0f), mx1 = __builtin_amdgcn_ballot_w64(fu <= 2.0f);
        const uint64_t my0 = m_hit & __builtin_amdgcn_ballot_w64(fv >= 0.0f), my1 = m_hit & __builtin_amdgcn_ballot_w64(fv <= 2.0f);
#define HESS_PX_ADD(P, B)                                                                                          \
  (void)__hip_atomic_fetch_add((P), (unsigned long long)__float2uint_rz(fmaf((B), wb0, 0.5f)) |                    \
                                       ((unsigned long long)__float2uint_rz(fmaf((B), wb1, 0.5f)) << 32),           \
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
        if (__builtin_amdgcn_inverse_ballot_w64(my0 & mx0)) HESS_PX_ADD(p, b00);
        if (__builtin_amdgcn_inverse_ballot_w64(my0 & mx1)) HESS_PX_ADD(p + 8, b01);
        if (__builtin_amdgcn_inverse_ballot_w64(my1 & mx0)) HESS_PX_ADD(p + 32, b10);
        if (__builtin_amdgcn_inverse_ballot_w64(my1 & mx1)) HESS_PX_ADD(p + 40, b11);
#undef HESS_PX_ADD
      }
    };
    {
      constexpr int UN = HESS_PX_UNROLL;
      // software pipeline: the gathers of the next chunk are in flight while the current chunk is accumulated
      PixChunk<UN> ca, cb;
      stage_a(ca);
      for (int it0 = 0; it0 < nit; it0 += 2 * UN) {
        stage_a(cb);
        stage_b(ca);
        if (it0 + UN >= nit) break;
        stage_a(ca);
        stage_b(cb);
      }
    }
    }  // row bands
    }  // column bands
    // The copies' sums: lane (cell, q) reads words 2q, 2q+1 of its cell in every copy (one 16-byte read each), adds the
    // four 32-bit halves apart and clears the words for the next feature.  It owns bins 2q, 2q+1:
    //   bin 2q   = low half of word 2q   + high half of word 2q-1 (lane q-1 of the quad, q = 0: word 7, lane q = 3)
    //   bin 2q+1 = low half of word 2q+1 + high half of word 2q
    __builtin_amdgcn_wave_barrier();
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int cpy = 0; cpy < PX_COPIES; cpy++) {
      uint4* const w = reinterpret_cast<uint4*>(sums + cpy * PX_COPY_U64 + 2 * lane);
      const uint4 x = *w;
      t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
      *w = make_uint4(0u, 0u, 0u, 0u);
    }
    {
      const uint32_t prev_hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)t.w, 0x93 /* quad_perm:[3,0,1,2] */, 0xF, 0xF, true);
      const float f0 = (float)(t.x + prev_hi) * rscale, f1 = (float)(t.z + t.y) * rscale;
      *reinterpret_cast<float2*>(&dl[wv][2 * lane]) = make_float2(f0, f1);
    }
    __builtin_amdgcn_wave_barrier();
    float* dout = desc + (obase + oidx) * dim;
    float* hout = (HOST_MIRROR && dp.hdesc) ? dp.hdesc + (obase + oidx) * dim : nullptr;
    if (dp.half_sift) {  // des[k] += des[k+4], ProgramCU.cu:1782-1785: lane < 32 -> cell lane/2, k = 2 (lane & 1) + {0, 1}
      float2 v = make_float2(0, 0);
      if (lane < 32) {
        const float* cellp = &dl[wv][(lane >> 1) * 8 + (lane & 1) * 2];
        const float2 lo = *reinterpret_cast<const float2*>(cellp), hi = *reinterpret_cast<const float2*>(cellp + 4);
        v = make_float2(lo.x + hi.x, lo.y + hi.y);
      }
      finish_descriptor64<HOST_MIRROR>(dp, lane, v, dout, hout);
    } else {
      float4 v = make_float4(0, 0, 0, 0);
      if (lane < 32) v = *reinterpret_cast<const float4*>(&dl[wv][lane * 4]);
      finish_descriptor128<HOST_MIRROR>(dp, lane, v, dout, hout);
    }
    __builtin_amdgcn_wave_barrier();  // (dl and the sums are rewritten by the next feature)
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
                         const int* list_total, int cap_list, const int* ocount, int* foffset, int* fsrc,
                         int* feat_total, int* feat_first, int cap_feat, int* overflow, int* img_base, int* host_small,
                         int batch) {
  FeatScan fs;
  fs.lp = lp; fs.multi = multi; fs.foffset = foffset; fs.fsrc = fsrc; fs.feat_total = feat_total; fs.feat_first = feat_first;
  fs.cap_feat = cap_feat; fs.overflow = overflow; fs.img_base = img_base; fs.host_small = host_small;
  // chunks of >= 4096 keypoints of an image's list, one workgroup each (see feature_scan_image)
  const int nchunk = std::min(16, std::max(1, cap_list / 4096));
  hipLaunchKernelGGL(feature_scan_kernel, dim3(nchunk, batch), dim3(1024), 0, st, g, fs, list, list_total, cap_list, ocount);
}

void launch_descriptor(hipStream_t st, const Geom& g, const DescParams& dp, const RawKey* list,
                       int cap_list, const FRec* recs, const int* fsrc, const int* feat_total,
                       const int* feat_first, const int* img_base, const float* got, HostKeypoint* keys,
                       float* desc, int cap_feat, int batch, int seen_features) {
  // (dp.first_image: first image of this launch; `batch` images from there)
  // Grid: one wavefront per feature where that is known to fit -- the workgroup dispatcher then hands the features out
  // largest first as wavefront slots free up, which a fixed stride over a smaller grid does not (configs[4], 25 k features
  // per launch over 8192 wavefronts of which 5120 are resident: 0.549 -> 0.473 ms per image with 16384, same call,
  // profiles/r06_experiments/descriptor_grid.txt).  The count of this batch is not known on the host; the largest count per
  // image of the context's last batch (+ 25 %) stands in for it, and the stride loop covers what exceeds the grid.
  // Multiples of 128 workgroups keep whole XCD blocks (see dpx.xcd_block below).
  const int den = dp.part_den > 1 ? dp.part_den : 1;
  int blocks = 2048;
  if (seen_features > 0) {
    const long long want = ((long long)seen_features * 5 / 4 / den + 3) / 4;
    blocks = (int)std::min<long long>(16384, std::max<long long>(256, (want + 127) / 128 * 128));
  }
  const int cap_blocks = (cap_feat / den + 3) / 4;
  if (blocks > cap_blocks) blocks = cap_blocks;
  if (blocks < 1) blocks = 1;
  const int lds_pad = DC_LDS_PAD_BYTES;
  DescParams dpx = dp;
  dpx.px_band = std::min(PX_BAND_MAX, std::max(64, dp.px_band > 0 ? dp.px_band : PX_BAND_MAX));
  // the block order needs whole blocks per XCD, and a grid of whole rounds over the eight XCDs (else the map from
  // (XCD, wavefront in the XCD) to list blocks is not onto: features would be skipped and others computed twice)
  if (dpx.xcd_block && ((blocks * 4) % (8 * dpx.xcd_block) != 0 || blocks % 8 != 0)) dpx.xcd_block = 0;
#define HESS_DESC_LAUNCH(MIRROR, SEQ)                                                                                  \
  hipLaunchKernelGGL((descriptor_kernel<MIRROR, SEQ>), dim3(blocks, batch), dim3(256), lds_pad, st, g, dpx, list, cap_list, \
                     recs, fsrc, feat_total, feat_first, img_base, got, keys, desc, cap_feat)
  const bool mirror = dp.hkeys || dp.hdesc;
  if (dp.pixel) {
    if (mirror) hipLaunchKernelGGL((descriptor_pixel_kernel<true>), dim3(blocks, batch), dim3(256), 0, st, g, dpx, list, cap_list, recs, fsrc, feat_total, feat_first, img_base, got, keys, desc, cap_feat);
    else hipLaunchKernelGGL((descriptor_pixel_kernel<false>), dim3(blocks, batch), dim3(256), 0, st, g, dpx, list, cap_list, recs, fsrc, feat_total, feat_first, img_base, got, keys, desc, cap_feat);
  } else if (dp.sequential) { if (mirror) HESS_DESC_LAUNCH(true, true); else HESS_DESC_LAUNCH(false, true); }
  else { if (mirror) HESS_DESC_LAUNCH(true, false); else HESS_DESC_LAUNCH(false, false); }
#undef HESS_DESC_LAUNCH
}

}  // namespace hess
