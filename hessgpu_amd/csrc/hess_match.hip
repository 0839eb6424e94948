// hess_match.hip -- descriptor matcher for gfx950: the step after the hot path.
//
// Replaces SiftMatchCU (SiftMatchCU.cpp:71-176) and its kernels MultiplyDescriptor(_G)_Kernel,
// RowMatch_Kernel, ColMatch_Kernel (ProgramCU.cu:3455-3843).  Integer work: results are bit-exact.
//
// Unguided match (GetSiftMatch): matrix cores, no score matrix in memory --
//   match_mfma_kernel      one workgroup per (256-row block, column segment): the segment's descriptors of set 2 pass
//                          through LDS once for its four wavefronts, each of which holds 64 rows of set 1 in registers and
//                          folds the v_mfma_i32_32x32x32_i8 tiles into RowMatch_Kernel's per-thread states and the column
//                          partials as packed 32-bit keys; see the comment at the kernel;
//   match_finish_kernel    merges a row's per-segment states (largest score, then the reference's tie order: lower
//                          thread class, lower column), acos distance + ratio;
//                          and, in the same launch, the per-row-block (max, index, second) column partials in ascending
//                          row order (match_col_kernel: the same for the small / guided path).
// Guided match (GetGuidedSiftMatch: per-pair homography / fundamental-matrix gates, per-8-row-block rule) --
//   match_dot_kernel       64x64 tile of the dot-product matrix per workgroup, descriptor panels in LDS,
//                          v_dot4_u32_u8, gates per pair, score matrix written for
//   match_row_kernel       one wavefront per row over the matrix (same tie order), and match_col_kernel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hess_abi.h"

namespace {

constexpr int TM = 64, TN = 64, KD = 128;

struct GeoParams {
  int guided;
  float H[9], F[9];
  float hdistmax, fdistmax;
};

// dotm[i][j] = what RowMatch reads (clamped at 0 in guided mode).  cpart[tile_row][j] = (max, index,
// second) of the reference's unclamped `results` over the tile's 64 rows in ascending order -- the
// d_temp partials of MultiplyDescriptor_Kernel (ProgramCU.cu:3510-3524), 64 rows at a time instead of 8.
__global__ __launch_bounds__(256) void match_dot_kernel(const uint8_t* des1, int num1, const uint8_t* des2, int num2,
                                                        const float2* loc1, const float2* loc2, GeoParams gp,
                                                        int3* cpart, int* dotm) {
  __shared__ uint32_t a[TM][KD / 4 + 1];  // +1 dword: conflict-free column-of-rows reads
  __shared__ uint32_t b[TN][KD / 4 + 1];
  __shared__ int good_blk[TM / 8][TN];
  __shared__ int3 cp[TM / 4][TN];
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN, tid = threadIdx.x;
  for (int g = tid; g < TM * (KD / 16); g += 256) {  // 16-byte loads
    const int r = g >> 3, q = g & 7;
    uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
    if (i0 + r < num1) va = *reinterpret_cast<const uint4*>(des1 + (size_t)(i0 + r) * KD + q * 16);
    if (j0 + r < num2) vb = *reinterpret_cast<const uint4*>(des2 + (size_t)(j0 + r) * KD + q * 16);
    a[r][q * 4] = va.x; a[r][q * 4 + 1] = va.y; a[r][q * 4 + 2] = va.z; a[r][q * 4 + 3] = va.w;
    b[r][q * 4] = vb.x; b[r][q * 4 + 1] = vb.y; b[r][q * 4 + 2] = vb.z; b[r][q * 4 + 3] = vb.w;
  }
  if (tid < (TM / 8) * TN) (&good_blk[0][0])[tid] = 0;
  for (int g = tid + 256; g < (TM / 8) * TN; g += 256) (&good_blk[0][0])[g] = 0;
  __syncthreads();
  const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;  // this thread: rows ti..ti+3, cols tj..tj+3 of the tile
  int acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[r][c] = 0;
#pragma unroll 4
  for (int k = 0; k < KD / 4; k++) {
    uint32_t av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; r++) av[r] = a[ti + r][k];
#pragma unroll
    for (int c = 0; c < 4; c++) bv[c] = b[tj + c][k];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[r][c] = (int)__builtin_amdgcn_udot4(av[r], bv[c], (uint32_t)acc[r][c], false);
  }
  int base[4][4];
  if (gp.guided) {  // ProgramCU.cu:3597-3635
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = i0 + ti + r;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int j = j0 + tj + c;
        int v = -262144;
        if (i < num1 && j < num2) {
          const float2 l1 = loc1[i], l2 = loc2[j];
          const float x0 = fmaf(gp.H[0], l1.x, gp.H[1] * l1.y) + gp.H[2];
          const float x1 = fmaf(gp.H[3], l1.x, gp.H[4] * l1.y) + gp.H[5];
          const float x2 = fmaf(gp.H[6], l1.x, gp.H[7] * l1.y) + gp.H[8];
          const float d0 = fabsf(x0 / x2 - l2.x), d1 = fabsf(x1 / x2 - l2.y);
          if (d0 < gp.hdistmax && d1 < gp.hdistmax) {
            const float fx0 = fmaf(gp.F[0], l1.x, gp.F[1] * l1.y) + gp.F[2];
            const float fx1 = fmaf(gp.F[3], l1.x, gp.F[4] * l1.y) + gp.F[5];
            const float fx2 = fmaf(gp.F[6], l1.x, gp.F[7] * l1.y) + gp.F[8];
            const float ft0 = fmaf(gp.F[0], l2.x, gp.F[3] * l2.y) + gp.F[6];
            const float ft1 = fmaf(gp.F[1], l2.x, gp.F[4] * l2.y) + gp.F[7];
            const float x2fx1 = fmaf(l2.x, fx0, l2.y * fx1) + fx2;
            const float se = (x2fx1 * x2fx1) / fmaf(ft1, ft1, fmaf(ft0, ft0, fmaf(fx0, fx0, fx1 * fx1)));
            v = se < gp.fdistmax ? 0 : -262144;
          }
        }
        base[r][c] = v;
        if (v >= 0) atomicAdd(&good_blk[(ti + r) >> 3][tj + c], 1);  // `good_count` of the 8-row block
      }
    }
  }
  __syncthreads();
  int3 loc_c[4];
#pragma unroll
  for (int c = 0; c < 4; c++) loc_c[c] = make_int3(0, -1, 0);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = i0 + ti + r;
    if (i >= num1) continue;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int j = j0 + tj + c;
      if (j >= num2) continue;
      int res = acc[r][c];
      if (gp.guided) res = base[r][c] + (good_blk[(ti + r) >> 3][tj + c] > 0 ? acc[r][c] : 0);
      dotm[(size_t)i * num2 + j] = gp.guided ? max(res, 0) : res;  // ProgramCU.cu:3684
      if (cpart) {  // strict '>' in ascending row order: the lowest row keeps a tie (ProgramCU.cu:3516-3519)
        if (res > loc_c[c].x) loc_c[c] = make_int3(res, i, loc_c[c].x);
        else loc_c[c].z = max(loc_c[c].z, res);
      }
    }
  }
  if (cpart) {
#pragma unroll
    for (int c = 0; c < 4; c++) cp[tid >> 4][tj + c] = loc_c[c];
    __syncthreads();
    if (tid < TN && j0 + tid < num2) {
      int3 t = cp[0][tid];
      for (int q = 1; q < TM / 4; q++) {  // merge the 16 four-row partials in row order (ColMatch_Kernel's rule)
        const int3 u = cp[q][tid];
        if (t.x < u.x) t = make_int3(u.x, u.y, max(t.x, u.z));
        else t.z = max(t.z, u.x);
      }
      cpart[(size_t)blockIdx.y * num2 + j0 + tid] = t;
    }
  }
}

__device__ __forceinline__ int decide(int best, int second, int idx, float distmax, float ratiomax) {
  const float dist = (float)acos(fmin((double)(best * 0.000003814697265625f), 1.0));     // ProgramCU.cu:3785
  const float distn = (float)acos(fmin((double)(second * 0.000003814697265625f), 1.0));
  return (dist < distmax) && (dist < distn * ratiomax) ? idx : -1;
}

// RowMatch_Kernel semantics: lane = (class c = j mod 32, half); strict '>' per lane keeps its first
// maximum; the two lanes of a class merge towards the lower j, then the classes merge with the
// reference's own tree (partner 16, 8, 4, 2, 1 away; a tie keeps the lower class of the pair).
__global__ __launch_bounds__(256) void match_row_kernel(const int* dotm, int num1, int num2, float distmax,
                                                        float ratiomax, int* rowm) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= num1) return;
  const int* p = dotm + (size_t)row * num2;
  int mx = 0, nx = 0, ix = -1;
  for (int j = lane; j < num2; j += 64) {  // lane covers j = lane, lane+64, ...: class lane&31
    const int v = p[j];
    const bool t = v > mx;
    nx = t ? mx : max(nx, v);
    ix = t ? j : ix;
    mx = t ? v : mx;
  }
  // merge: first the two lanes of a class (lower j wins ties), then classes 16,8,4,2,1 apart (lower class wins)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int omx = __shfl_down(mx, d), onx = __shfl_down(nx, d), oix = __shfl_down(ix, d);
    bool take;  // take the other lane's candidate?
    if (d == 32) take = (omx > mx) || (omx == mx && oix >= 0 && (ix < 0 || oix < ix));
    else take = omx > mx;
    const int nmx = take ? omx : mx;
    const int nnx = take ? max(mx, onx) : max(nx, omx);
    ix = take ? oix : ix;
    nx = nnx;
    mx = nmx;
  }
  if (lane == 0) rowm[row] = decide(mx, nx, ix, distmax, ratiomax);
}

// ColMatch_Kernel (ProgramCU.cu:3808-3827): merge the row-block partials of a column in ascending row order.
// The merge is associative (ties go to the earlier rows), so a column's partials are split into 8 consecutive
// chunks folded by 8 threads and combined in chunk order: 32 columns x 8 chunks per workgroup.
__device__ __forceinline__ void match_col_block(int block, const int3* cpart, int ntile, int num2, float distmax,
                                                float ratiomax, int* colm) {
  __shared__ int3 part[8][32];
  const int c = threadIdx.x & 31, ch = threadIdx.x >> 5;
  const int j = block * 32 + c;
  const int per = (ntile + 7) >> 3;
  const int q0 = ch * per, q1 = min(q0 + per, ntile);
  int3 t = make_int3(0, -1, 0);  // neutral: the per-block partials start from the same state
  if (j < num2)
    for (int q = q0; q < q1; q++) {
      const int3 u = cpart[(size_t)q * num2 + j];
      if (t.x < u.x) t = make_int3(u.x, u.y, max(t.x, u.z));
      else t.z = max(t.z, u.x);
    }
  part[ch][c] = t;
  __syncthreads();
  if (ch == 0 && j < num2) {
    for (int k = 1; k < 8; k++) {
      const int3 u = part[k][c];
      if (t.x < u.x) t = make_int3(u.x, u.y, max(t.x, u.z));
      else t.z = max(t.z, u.x);
    }
    colm[j] = decide(t.x, t.z, t.y, distmax, ratiomax);
  }
}

__global__ __launch_bounds__(256) void match_col_kernel(const int3* cpart, int ntile, int num2, float distmax,
                                                        float ratiomax, int* colm) {
  match_col_block(blockIdx.x, cpart, ntile, num2, distmax, ratiomax, colm);
}

// ---- unguided match on the matrix cores -----------------------------------------------------------------
// The num1 x num2 dot products are never written out.
//
// Work split.  A workgroup (four wavefronts) owns 256 rows of set 1 and a segment of the columns (set 2); wavefront w
// holds rows 64 w .. 64 w + 63 as the A fragments of two 32-row blocks, resident in registers.  The segment is walked
// in SUPER TILES of 128 columns: the workgroup copies the 16 KB of descriptors (biased, see below) and the 128 column
// offsets into LDS -- one coalesced 16-byte load per thread and quarter, issued a whole super tile ahead, double
// buffered, one barrier per super tile -- and every wavefront reads its B fragments from there: set 2 crosses L2 once
// per 256 rows (round 1-5's kernel: once per 32 rows, straight from L2, 16 bytes per lane at a 32-byte stride).
// Per 32-column tile a wavefront issues 2 x 4 v_mfma_i32_32x32x32_i8.
//
// Scores.  Descriptors are unsigned bytes, the instruction multiplies signed ones: bytes are biased by -128 (xor 0x80)
// and the exact dot product is dot_s + 128 (sum a + sum b) - 128^2 * 128.  The row part (rfix) rides in as the MFMA's
// C operand, the column part (cfix) is added while the key is formed.  Both sets are padded with zero descriptors to
// whole blocks (rows: 256, columns: 128), whose exact score is 0 -- never a maximum (the reference's scans start at 0
// and replace on '>' only), so there is no validity test anywhere in the loop.
//
// Folds.  The C layout puts column j0 + (lane & 31) on the lane and 16 rows of a block in its registers, and a tile
// aligned to 32 columns holds exactly one element of each of RowMatch_Kernel's 32 strided threads (class = j mod 32 =
// lane & 31): the per-thread scan of the reference (strict '>' keeps the first maximum, second = second largest,
// ProgramCU.cu:3745-3760) is a per-lane fold over the tiles with no cross-lane traffic.  State per (lane, register): two
// packed keys, key = score << 6 | (62 - tile index in the segment): `best = max(best, key)` keeps the largest score and,
// among equal scores, the earliest tile; `second = med3(best, key, second)` is the second largest key, whose score is the
// second largest score counting duplicates -- three vector instructions per element (shift-add, med3, max) where
// the (max, second, index) triple took seven.  63 in the low bits = "none yet" (a score of 0 forms a key below it).
// Column partials (max, row, second over the wavefront's rows in ascending order, ColMatch's rule, ProgramCU.cu:3510-3519)
// the same way with key = score << 6 | (62 - row in the lane's half), formed from the row key by one add of a scalar.
// At the end of the segment the 32 classes of a row are merged through LDS (one lane per row walks them in ascending
// class order: the reference's tree keeps the lower thread on ties, ProgramCU.cu:3766-3780) and ONE (best, second,
// column) per row and segment is stored.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MM_ROWS = 256;            // rows of set 1 per workgroup (64 per wavefront)
constexpr int MM_SUPER = 128;           // columns per super tile
constexpr int MM_PITCH = KD + 16;       // LDS bytes per staged descriptor: 36 dwords, 16-byte reads of 32 columns spread over the banks
constexpr int MM_MAX_TILES = 60;        // tiles per segment: the tile index shares 6 key bits with "none"
constexpr int MM_RS_PITCH = 33;         // row-state exchange: 8-byte entries per row (32 classes + 1 pad)

__device__ __forceinline__ void col_merge(int3& a, const int3& b) {  // a: earlier rows, b: later rows
  if (a.x < b.x) a = make_int3(b.x, b.y, max(a.x, b.z));
  else a.z = max(a.z, b.x);
}

// median of three (v_med3_i32: the compiler does not form it from min/max of three variables)
__device__ __forceinline__ int med3i(int a, int b, int c) {
  int d;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <bool COLS>
__global__ __launch_bounds__(256) void match_mfma_kernel(const uint8_t* des1, int num1, const uint8_t* des2, int num2,
                                                         const int* rfix, const int* cfix, int nseg, int supers_per_seg,
                                                         int nsuper, int3* cpart, int3* rstate) {
  __shared__ __attribute__((aligned(16))) uint8_t bufB[2][MM_SUPER * MM_PITCH];
  __shared__ int bufC[2][MM_SUPER];
  __shared__ int3 cp[2][4][MM_SUPER];  // the four wavefronts' column partials of a super tile, merged after its barrier
  static_assert(sizeof(bufB) >= 4 * 32 * MM_RS_PITCH * 8, "the row-state exchange reuses the descriptor buffers");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int sg = blockIdx.y;
  const int i0 = blockIdx.x * MM_ROWS + wv * 64;  // this wavefront's first row
  const int s0 = sg * supers_per_seg, s1 = min(s0 + supers_per_seg, nsuper);
  v4i a[2][4];
  v16i ra[2];
  {
    // hardware row r of a tile carries descriptor row perm(r) of the block, chosen so that the 16 accumulator registers
    // of a lane are 16 CONSECUTIVE descriptor rows (reg + 16 h): a lane's column partial is then a plain in-order fold
    const int prow = (r & 3) + 4 * (r >> 3) + 16 * ((r >> 2) & 1);
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
      const uint8_t* pa = des1 + (size_t)(i0 + 32 * blk + prow) * KD + 16 * h;  // (set 1 is padded to whole workgroups)
#pragma unroll
      for (int kk = 0; kk < 4; kk++) a[blk][kk] = *reinterpret_cast<const v4i*>(pa + kk * 32) ^ (int)0x80808080;
      const v4i* const pr = reinterpret_cast<const v4i*>(rfix + i0 + 32 * blk + 16 * h);  // (16 consecutive rows: 64-byte aligned)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const v4i t = pr[q];
        ra[blk][4 * q] = t.x; ra[blk][4 * q + 1] = t.y; ra[blk][4 * q + 2] = t.z; ra[blk][4 * q + 3] = t.w;
      }
    }
  }
  int rmx[2][16], rnx[2][16];
#pragma unroll
  for (int blk = 0; blk < 2; blk++)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) { rmx[blk][reg] = 63; rnx[blk][reg] = 0; }

  // staging: thread t copies 16-byte pieces t, t + 256, t + 512, t + 768 of a super tile (piece = 8 x column + part)
  v4i st[4];
  int stc = 0;
  auto stage_load = [&](int sup) {
    const uint8_t* src = des2 + (size_t)sup * MM_SUPER * KD + (size_t)tid * 16;
#pragma unroll
    for (int q = 0; q < 4; q++) st[q] = *reinterpret_cast<const v4i*>(src + q * 4096);
    if (tid < MM_SUPER) stc = cfix[sup * MM_SUPER + tid];
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int piece = tid + 256 * q;
      *reinterpret_cast<v4i*>(&bufB[buf][(piece >> 3) * MM_PITCH + (piece & 7) * 16]) = st[q] ^ (int)0x80808080;
    }
    if (tid < MM_SUPER) bufC[buf][tid] = stc;
  };
  stage_load(s0);
  stage_store(0);
  __syncthreads();
  for (int sup = s0; sup < s1; sup++) {
    const int cur = (sup - s0) & 1;
    if (sup + 1 < s1) stage_load(sup + 1);  // in flight while this super tile is multiplied
    // (one tile at a time: a software pipeline over the tiles -- the next tile's fragments read and its MFMAs issued between
    // this tile's folds -- was measured twice and lost both times: unrolled it needs 260 registers and spills (- 9 %),
    // rolled it fits 191 and runs 3 % slower than this loop, whose waits the SIMD's other wavefront fills:
    // profiles/r06_experiments/matcher.txt)
#pragma unroll 1
    for (int tt = 0; tt < 4; tt++) {
      const int tl = (sup - s0) * 4 + tt;     // tile index in the segment
      const int ct = 62 - tl;
      const uint8_t* pb = &bufB[cur][(tt * 32 + r) * MM_PITCH + 16 * h];
      v4i b[4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) b[kk] = *reinterpret_cast<const v4i*>(pb + kk * 32);
      const int cbk = (bufC[cur][tt * 32 + r] << 6) | ct;
      int cx = 63, cz = 0;  // column partial over this lane's 32 rows
#pragma unroll
      for (int blk = 0; blk < 2; blk++) {
        v16i c = ra[blk];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[blk][kk], b[kk], c, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
          const int k = (int)(((unsigned)c[reg] << 6) + (unsigned)cbk);  // (score << 6) | ct: score = c + column offset >= 0
          rnx[blk][reg] = med3i(rmx[blk][reg], k, rnx[blk][reg]);
          rmx[blk][reg] = max(rmx[blk][reg], k);
          if (COLS) {
            const int kc = k + ((62 - (reg + 16 * blk)) - ct);            // low bits: 62 - row in the lane's half
            cz = med3i(cx, kc, cz);
            cx = max(cx, kc);
          }
        }
      }
      if (COLS) {
        // the lane's rows are (reg, 32 + reg) + 16 h of the wavefront's 64: decode, then merge the two lane halves
        // (largest score; equal scores: the lower row, ColMatch's ascending order)
        const int low = cx & 63, rl = 62 - low;
        int bx = cx >> 6, by = low == 63 ? -1 : i0 + 32 * (rl >> 4) + (rl & 15) + 16 * h, bz = cz >> 6;
        const int ox = __shfl_xor(bx, 32), oy = __shfl_xor(by, 32), oz = __shfl_xor(bz, 32);
        const bool take = ox > bx || (ox == bx && oy >= 0 && (by < 0 || oy < by));
        const int nz = max(max(bz, oz), min(bx, ox));
        bx = take ? ox : bx;
        by = take ? oy : by;
        if (h == 0) cp[cur][wv][tt * 32 + r] = make_int3(bx, by, nz);
      }
    }
    if (sup + 1 < s1) stage_store(cur ^ 1);  // (read last during super tile sup - 1: every wavefront is past that barrier)
    __syncthreads();
    if (COLS && tid < MM_SUPER) {
      // one partial per column and WORKGROUP (256 rows): the wavefronts' partials in row order (ColMatch's merge rule);
      // cp[cur] is next written during super tile sup + 2, i.e. after the barrier that ends sup + 1
      int3 t = cp[cur][0][tid];
#pragma unroll
      for (int w = 1; w < 4; w++) col_merge(t, cp[cur][w][tid]);
      const int j = sup * MM_SUPER + tid;
      if (j < num2) cpart[(size_t)blockIdx.x * num2 + j] = t;
    }
  }
  // ---- the 32 classes of every row -> one state per row and segment, through LDS (the descriptor buffers are free) ----
  int2* const rs = reinterpret_cast<int2*>(&bufB[0][0]) + wv * 32 * MM_RS_PITCH;
#pragma unroll
  for (int blk = 0; blk < 2; blk++) {
#pragma unroll
    for (int reg = 0; reg < 16; reg++) rs[(reg + 16 * h) * MM_RS_PITCH + r] = make_int2(rmx[blk][reg], rnx[blk][reg]);
    __builtin_amdgcn_wave_barrier();  // (one wavefront's LDS operations complete in order)
    {
      // lane (row r, half hh) walks the 16 classes 16 hh .. 16 hh + 15 of its row in ascending order ('>' on the score: the
      // lower class keeps a tie); then the two halves of a row are combined the same way (the lower classes' half wins a tie)
      const int2* const row = rs + r * MM_RS_PITCH + 16 * h;
      int2 s = row[0];
      int cls = 16 * h;
#pragma unroll 5
      for (int c2 = 1; c2 < 16; c2++) {
        const int2 u = row[c2];
        const bool take = (u.x >> 6) > (s.x >> 6);
        s.y = take ? max(s.x, u.y) : max(s.y, u.x);
        s.x = take ? u.x : s.x;
        cls = take ? 16 * h + c2 : cls;
      }
      const int ox = __shfl_xor(s.x, 32), oy = __shfl_xor(s.y, 32), ocls = __shfl_xor(cls, 32);
      const bool take = (ox >> 6) > (s.x >> 6);   // (meaningful in the lanes of half 0, which store)
      s.y = take ? max(s.x, oy) : max(s.y, ox);
      s.x = take ? ox : s.x;
      cls = take ? ocls : cls;
      const int low = s.x & 63;
      const int grow = i0 + 32 * blk + r;
      if (h == 0 && grow < num1)
        rstate[(size_t)grow * nseg + sg] = make_int3(s.x >> 6, s.y >> 6, low == 63 ? -1 : (s0 * 4 + 62 - low) * 32 + cls);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Rows: merge the per-segment states in the reference's order -- largest score; equal scores: the lower thread class
// (column mod 32: the tree keeps the lower thread), then the lower column (a thread keeps its first maximum).
// ... and, in the same launch, the columns (match_col_block) -- workgroups [0, row_blocks) take rows, the rest columns
// (a launch of its own for either costs more than its work: 5 us each at 8192 x 8192).
__global__ __launch_bounds__(256) void match_finish_kernel(const int3* rstate, int num1, int nseg, int row_blocks,
                                                           const int3* cpart, int ntile, int num2, float distmax,
                                                           float ratiomax, int* rowm, int* colm) {
  if ((int)blockIdx.x >= row_blocks) {  // (workgroup-uniform)
    match_col_block(blockIdx.x - row_blocks, cpart, ntile, num2, distmax, ratiomax, colm);
    return;
  }
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= num1) return;
  const int3* p = rstate + (size_t)row * nseg;
  int3 s = p[0];
  for (int q = 1; q < nseg; q++) {
    const int3 u = p[q];
    const bool take = u.x > s.x || (u.x == s.x && u.z >= 0 &&
                                    (s.z < 0 || (u.z & 31) < (s.z & 31) || ((u.z & 31) == (s.z & 31) && u.z < s.z)));
    s.y = u.x > s.x ? max(s.x, u.y) : max(s.y, u.x);
    if (take) { s.x = u.x; s.z = u.z; }
  }
  rowm[row] = decide(s.x, s.y, s.z, distmax, ratiomax);
}

}  // namespace

struct hess_matcher {
  int device = 0, max_sift = 4096;
  hipStream_t st = nullptr;
  uint8_t* des[2] = {nullptr, nullptr};
  float2* loc[2] = {nullptr, nullptr};
  int num[2] = {0, 0}, have_loc[2] = {0, 0};
  int3* cpart = nullptr;
  int *dotm = nullptr, *rowm = nullptr, *colm = nullptr;
  size_t mat_cap = 0;
  // matrix-core path (unguided): per-descriptor byte sums turned into score offsets, per-segment row states
  int* fix[2] = {nullptr, nullptr};
  int3 *cpart2 = nullptr, *rstate = nullptr;
  size_t cpart2_cap = 0, rstate_cap = 0;
  int rc_cap = 0;
  std::vector<int> hrow, hcol;
  std::string err;
  float last_ms = 0.0f;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

#define M_TRY(m, expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (m)->err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                   \
      return HESS_ERR_DEVICE;                                                                \
    }                                                                                        \
  } while (0)

extern "C" {

hess_matcher* hess_matcher_create(int device, int max_sift) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    fprintf(stderr, "hessgpu: no usable HIP device %d (found %d)\n", device, ndev);
    return nullptr;
  }
  hess_matcher* m = new (std::nothrow) hess_matcher();
  if (!m) return nullptr;
  m->device = device;
  m->max_sift = max_sift > 0 ? max_sift : 4096;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking) != hipSuccess) {
    delete m;
    return nullptr;
  }
  (void)hipEventCreate(&m->e0);
  (void)hipEventCreate(&m->e1);
  return m;
}

void hess_matcher_destroy(hess_matcher* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  for (int k = 0; k < 2; k++) { (void)hipFree(m->des[k]); (void)hipFree(m->loc[k]); }
  (void)hipFree(m->cpart); (void)hipFree(m->dotm); (void)hipFree(m->rowm); (void)hipFree(m->colm);
  (void)hipFree(m->fix[0]); (void)hipFree(m->fix[1]); (void)hipFree(m->cpart2); (void)hipFree(m->rstate);
  if (m->e0) (void)hipEventDestroy(m->e0);
  if (m->e1) (void)hipEventDestroy(m->e1);
  if (m->st) (void)hipStreamDestroy(m->st);
  delete m;
}

int hess_matcher_set_max(hess_matcher* m, int max_sift) {
  if (!m || max_sift <= 0) return HESS_ERR_ARG;
  m->max_sift = max_sift;
  return 0;
}

// SiftMatchCU::SetDescriptors(index, num, const unsigned char*), SiftMatchCU.cpp:71-85.
int hess_matcher_set_descriptors(hess_matcher* m, int index, int num, const unsigned char* des) {
  if (!m || !des || num < 0) return HESS_ERR_ARG;
  index = index > 1 ? 1 : (index < 0 ? 0 : index);
  M_TRY(m, hipSetDevice(m->device));
  if (num > m->max_sift) num = m->max_sift;
  m->have_loc[index] = 0;
  (void)hipFree(m->des[index]);
  m->des[index] = nullptr;
  m->num[index] = num;
  (void)hipFree(m->fix[index]);
  m->fix[index] = nullptr;
  if (num) {
    // padded with zero descriptors to whole blocks of the matrix-core path (256 rows / 128 columns, match_mfma_kernel):
    // their exact score is 0, which is never a maximum
    const size_t padded = ((size_t)num + MM_ROWS - 1) / MM_ROWS * MM_ROWS;
    M_TRY(m, hipMalloc(&m->des[index], padded * KD));
    M_TRY(m, hipMemcpy(m->des[index], des, (size_t)num * KD, hipMemcpyHostToDevice));
    if (padded > (size_t)num) M_TRY(m, hipMemset(m->des[index] + (size_t)num * KD, 0, (padded - num) * KD));
    // score offsets of the matrix-core path: rows 128*sum - 128^2*128, columns 128*sum (see match_mfma_kernel)
    std::vector<int> f(padded);
    for (size_t i = 0; i < padded; i++) {
      int sum = 0;
      if (i < (size_t)num)
        for (int k = 0; k < KD; k++) sum += des[i * KD + k];
      f[i] = 128 * sum - (index == 0 ? 128 * 128 * KD : 0);
    }
    M_TRY(m, hipMalloc(&m->fix[index], padded * sizeof(int)));
    M_TRY(m, hipMemcpy(m->fix[index], f.data(), padded * sizeof(int), hipMemcpyHostToDevice));
  }
  return 0;
}

// Float descriptors are quantised as the reference does: int(512*d + 0.5) into a byte (SiftMatchCU.cpp:88-100).
int hess_matcher_set_descriptors_f32(hess_matcher* m, int index, int num, const float* des) {
  if (!m || !des || num < 0) return HESS_ERR_ARG;
  if (num > m->max_sift) num = m->max_sift;
  std::vector<unsigned char> q((size_t)num * KD);
  for (size_t i = 0; i < q.size(); ++i) q[i] = (unsigned char)(int)(512 * des[i] + 0.5);
  return hess_matcher_set_descriptors(m, index, num, q.data());
}

// SiftMatchCU::SetFeautreLocation, SiftMatchCU.cpp:103-123: (x, y) pairs, `gap` floats skipped after each.
int hess_matcher_set_locations(hess_matcher* m, int index, const float* locations, int gap) {
  if (!m || !locations || index < 0 || index > 1) return HESS_ERR_ARG;
  const int n = m->num[index];
  if (n <= 0) return 0;
  M_TRY(m, hipSetDevice(m->device));
  std::vector<float2> h((size_t)n);
  for (int i = 0; i < n; i++) { h[i].x = locations[0]; h[i].y = locations[1]; locations += 2 + gap; }
  (void)hipFree(m->loc[index]);
  M_TRY(m, hipMalloc(&m->loc[index], (size_t)n * sizeof(float2)));
  M_TRY(m, hipMemcpy(m->loc[index], h.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice));
  m->have_loc[index] = 1;
  return 0;
}

// SiftMatchCU::GetSiftMatch / GetGuidedSiftMatch + GetBestMatch (SiftMatchCU.cpp:125-173).
// H, F: 3x3 row-major, both NULL for the unguided match.  Returns the number of matches (>= 0) or a
// negative hess_status.
int hess_matcher_match(hess_matcher* m, int max_match, int* pairs, const float* H, const float* F, float distmax,
                       float ratiomax, float hdistmax, float fdistmax, int mutual_best) {
  if (!m || !pairs || max_match < 0) return HESS_ERR_ARG;
  const int n1 = m->num[0], n2 = m->num[1];
  if (n1 <= 0 || n2 <= 0) return 0;
  const bool guided = (H != nullptr) || (F != nullptr);
  if (guided && (!H || !F)) { m->err = "guided matching needs both H and F"; return HESS_ERR_ARG; }
  if (guided && (!m->have_loc[0] || !m->have_loc[1])) return 0;  // SiftMatchCU.cpp:131
  M_TRY(m, hipSetDevice(m->device));
  if (m->rc_cap < m->max_sift) {
    (void)hipFree(m->rowm); (void)hipFree(m->colm);
    m->rowm = m->colm = nullptr;
    M_TRY(m, hipMalloc(&m->rowm, (size_t)m->max_sift * sizeof(int) + 4));
    M_TRY(m, hipMalloc(&m->colm, (size_t)m->max_sift * sizeof(int) + 4));
    m->rc_cap = m->max_sift;
  }
  // small problems (three launches of latency) stay on the one-pass dot kernel: 1024 x 1024 0.026 vs 0.033 ms
  if (!guided && (size_t)n1 * n2 > ((size_t)3 << 20)) {
    // matrix-core path: (256-row block, column segment) workgroups, at least two per CU where the problem allows;
    // a segment is whole super tiles of 128 columns, at most 15 of them (the tile index shares six key bits)
    const int nrb = (n1 + MM_ROWS - 1) / MM_ROWS, nsuper = (n2 + MM_SUPER - 1) / MM_SUPER;
    // Workgroups: whole rounds over the 256 CUs -- two per CU (what the registers allow) when that leaves a workgroup at
    // least four super tiles, else one per CU with twice the tiles (its prologue and the merge of the row states at its
    // end cost about as much as two super tiles).  Same call, 4096^2 / 8192^2, TMAC/s: 256 workgroups 91 / 193, 384:
    // 91 / 176, 512: 81 / 211, 768: 81 / 180 (profiles/r06_experiments/matcher.txt).
    const int target_wgs = (long long)nrb * nsuper >= 512 * 4 ? 512 : 256;
    int nseg = (target_wgs + nrb - 1) / nrb;
    nseg = nseg < 1 ? 1 : (nseg > nsuper ? nsuper : nseg);
    int sps = (nsuper + nseg - 1) / nseg;
    if (sps > MM_MAX_TILES / 4) sps = MM_MAX_TILES / 4;
    nseg = (nsuper + sps - 1) / sps;
    const size_t need_rs = (size_t)n1 * nseg, need_cp = (size_t)nrb * n2;  // column partials per 256-row workgroup
    if (need_rs > m->rstate_cap) {
      (void)hipFree(m->rstate); m->rstate = nullptr;
      M_TRY(m, hipMalloc(&m->rstate, need_rs * sizeof(int3)));
      m->rstate_cap = need_rs;
    }
    if (mutual_best && need_cp > m->cpart2_cap) {
      (void)hipFree(m->cpart2); m->cpart2 = nullptr;
      M_TRY(m, hipMalloc(&m->cpart2, need_cp * sizeof(int3)));
      m->cpart2_cap = need_cp;
    }
    (void)hipEventRecord(m->e0, m->st);
    if (mutual_best)
      hipLaunchKernelGGL(match_mfma_kernel<true>, dim3(nrb, nseg), dim3(256), 0, m->st, m->des[0], n1, m->des[1], n2, m->fix[0],
                         m->fix[1], nseg, sps, nsuper, m->cpart2, m->rstate);
    else
      hipLaunchKernelGGL(match_mfma_kernel<false>, dim3(nrb, nseg), dim3(256), 0, m->st, m->des[0], n1, m->des[1], n2, m->fix[0],
                         m->fix[1], nseg, sps, nsuper, nullptr, m->rstate);
    // rows and -- for mutual best -- columns in one launch (the partials of row blocks past num1 hold the neutral
    // (0, -1, 0) of their zero rows)
    const int row_blocks = (n1 + 255) / 256;
    hipLaunchKernelGGL(match_finish_kernel, dim3(row_blocks + (mutual_best ? (n2 + 31) / 32 : 0)), dim3(256), 0, m->st, m->rstate,
                       n1, nseg, row_blocks, m->cpart2, nrb, n2, distmax, ratiomax, m->rowm, m->colm);
  } else {
  const size_t need = (size_t)n1 * n2;
  if (need > m->mat_cap) {
    (void)hipFree(m->cpart); (void)hipFree(m->dotm);
    m->cpart = nullptr;
    m->dotm = nullptr;
    M_TRY(m, hipMalloc(&m->cpart, (size_t)((m->max_sift + TM - 1) / TM + 1) * m->max_sift * sizeof(int3)));
    M_TRY(m, hipMalloc(&m->dotm, need * sizeof(int)));
    m->mat_cap = need;
  }
  GeoParams gp;
  memset(&gp, 0, sizeof(gp));
  gp.guided = guided ? 1 : 0;
  if (guided) { memcpy(gp.H, H, 36); memcpy(gp.F, F, 36); gp.hdistmax = hdistmax; gp.fdistmax = fdistmax; }
  (void)hipEventRecord(m->e0, m->st);
  hipLaunchKernelGGL(match_dot_kernel, dim3((n2 + TN - 1) / TN, (n1 + TM - 1) / TM), dim3(256), 0, m->st, m->des[0], n1,
                     m->des[1], n2, m->loc[0], m->loc[1], gp, mutual_best ? m->cpart : nullptr, m->dotm);
  hipLaunchKernelGGL(match_row_kernel, dim3((n1 + 3) / 4), dim3(256), 0, m->st, m->dotm, n1, n2, distmax, ratiomax,
                     m->rowm);
  if (mutual_best)
    hipLaunchKernelGGL(match_col_kernel, dim3((n2 + 31) / 32), dim3(256), 0, m->st, m->cpart, (n1 + TM - 1) / TM, n2,
                       distmax, ratiomax, m->colm);
  }
  (void)hipEventRecord(m->e1, m->st);
  m->hrow.resize(n1);
  m->hcol.resize(n2);
  M_TRY(m, hipMemcpyAsync(m->hrow.data(), m->rowm, (size_t)n1 * 4, hipMemcpyDeviceToHost, m->st));
  if (mutual_best) M_TRY(m, hipMemcpyAsync(m->hcol.data(), m->colm, (size_t)n2 * 4, hipMemcpyDeviceToHost, m->st));
  M_TRY(m, hipStreamSynchronize(m->st));
  (void)hipEventElapsedTime(&m->last_ms, m->e0, m->e1);
  int nmatch = 0;
  for (int i = 0; i < n1 && nmatch < max_match; ++i) {
    const int j = m->hrow[i];
    if (j >= 0 && (!mutual_best || m->hcol[j] == i)) {
      pairs[2 * nmatch] = i;
      pairs[2 * nmatch + 1] = j;
      nmatch++;
    }
  }
  return nmatch;
}

float hess_matcher_last_ms(hess_matcher* m) { return m ? m->last_ms : 0.0f; }
const char* hess_matcher_last_error(hess_matcher* m) { return m ? m->err.c_str() : "null matcher"; }

}  // extern "C"
