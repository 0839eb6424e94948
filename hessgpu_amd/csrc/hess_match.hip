// hess_match.hip -- descriptor matcher for gfx950: the step after the hot path.
//
// Replaces SiftMatchCU (SiftMatchCU.cpp:71-176) and its kernels MultiplyDescriptor(_G)_Kernel,
// RowMatch_Kernel, ColMatch_Kernel (ProgramCU.cu:3455-3843).  Integer work: results are bit-exact.
//
// Unguided match (GetSiftMatch): matrix cores, no score matrix in memory --
//   match_mfma_kernel      one wavefront per (32-row block, column segment): v_mfma_i32_32x32x32_i8 tiles whose
//                          C layout (column on the lane, 16 rows in registers) makes RowMatch_Kernel's 32 strided
//                          thread scans and the column partials per-lane folds; see the comment at the kernel;
//   match_rowmerge_kernel  merges the per-segment thread states in column order, then the reference's 32-thread
//                          tree (partners 16, 8, 4, 2, 1 apart, ties keep the lower thread), acos distance + ratio;
//   match_col_kernel       merges the per-row-block (max, index, second) column partials in ascending row order.
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
__global__ __launch_bounds__(256) void match_col_kernel(const int3* cpart, int ntile, int num2, float distmax,
                                                        float ratiomax, int* colm) {
  __shared__ int3 part[8][32];
  const int c = threadIdx.x & 31, ch = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
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

// ---- unguided match on the matrix cores -----------------------------------------------------------------
// The num1 x num2 dot products are never written out.  One wavefront owns a block of 32 rows and a segment
// of the columns and walks the segment in 32-column tiles: 4 x v_mfma_i32_32x32x32_i8 per tile (A fragments =
// the block's 32 descriptors, resident in registers; B fragments = 32 descriptors of set 2, 16-byte loads
// straight from L2, no LDS).  The C layout puts column j0 + (lane & 31) on the lane and 16 rows in its
// registers, and a tile aligned to 32 columns holds exactly one element of each of RowMatch_Kernel's 32
// strided threads (class = j mod 32 = lane & 31): the per-thread scan of the reference (strict '>' keeps
// the first maximum, second = second largest) is a per-lane fold over the tiles with no cross-lane traffic.
// Column partials (max, index, second over the block's rows in ascending order, ColMatch's merge rule) are
// folded per lane over the registers and merged once across the two lane halves.
// Descriptors are unsigned bytes, the instruction multiplies signed ones: bytes are biased by -128 (xor 0x80)
// and the exact dot product is restored as dot_s + 128 (sum a + sum b) - 128^2 * 128 from per-descriptor sums.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void col_merge(int3& a, const int3& b) {  // a: earlier rows, b: later rows
  if (a.x < b.x) a = make_int3(b.x, b.y, max(a.x, b.z));
  else a.z = max(a.z, b.x);
}

__global__ __launch_bounds__(256) void match_mfma_kernel(const uint8_t* des1, int num1, const uint8_t* des2, int num2,
                                                         const int* rfix, const int* cfix, int nseg, int tiles_per_seg,
                                                         int3* cpart, int3* rstate) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rb = blockIdx.x, sg = blockIdx.y * 4 + wv;
  if (sg >= nseg) return;  // wavefront-uniform; no workgroup barrier in this kernel
  const int i0 = rb * 32;
  const int ntile2 = (num2 + 31) >> 5;
  v4i a[4];
  {
    // hardware row r of the tile carries descriptor row i0 + perm(r), chosen so that the 16 accumulator
    // registers of a lane are 16 CONSECUTIVE descriptor rows (i0 + reg + 16 h): the column partial of a
    // lane is then a plain in-order fold and the two lane halves merge once per tile
    const int prow = (r & 3) + 4 * (r >> 3) + 16 * ((r >> 2) & 1);
    const uint8_t* pa = des1 + (size_t)min(i0 + prow, num1 - 1) * KD + 16 * h;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) a[kk] = *reinterpret_cast<const v4i*>(pa + kk * 32) ^ (int)0x80808080;
  }
  int ra[16], mx[16], nx[16], ix[16];
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    const int row = i0 + reg + 16 * h;
    ra[reg] = row < num1 ? rfix[row] : -(1 << 29);  // rows past the end: far below every real score, no overflow
    mx[reg] = 0; nx[reg] = 0; ix[reg] = -1;          // RowMatch_Kernel's initial state, ProgramCU.cu:3745-3747
  }
  const int t1 = min((sg + 1) * tiles_per_seg, ntile2);
  auto load_b = [&](int t, v4i (&b)[4], int& cbv) {  // B fragments + column offset of tile t (clamped past the end)
    const int jj = t * 32 + r;
    const int jc = min(jj, num2 - 1);
    const uint8_t* pb = des2 + (size_t)jc * KD + 16 * h;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) b[kk] = *reinterpret_cast<const v4i*>(pb + kk * 32);
    cbv = jj < num2 ? cfix[jc] : -(1 << 29);  // columns past the end: far below every real score
  };
  v4i bn[4];
  int cbn;
  load_b(sg * tiles_per_seg, bn, cbn);
  for (int t = sg * tiles_per_seg; t < t1; t++) {
    const int j = t * 32 + r;
    v4i b[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) b[kk] = bn[kk] ^ (int)0x80808080;
    const int cb = cbn;
    load_b(min(t + 1, ntile2 - 1), bn, cbn);  // next tile's loads fly while this tile is reduced
    v16i c = {0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk], b[kk], c, 0, 0, 0);
    int cx = 0, cy = -1, cz = 0;  // column partial over this lane's 16 rows (ProgramCU.cu:3510-3519)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int v = c[reg] + ra[reg] + cb;                     // invalid row/column: far below 0
      // RowMatch_Kernel's thread update (ProgramCU.cu:3756-3760); second = median(max, v, second)
      nx[reg] = max(min(mx[reg], v), nx[reg]);
      ix[reg] = (v > mx[reg]) ? j : ix[reg];
      mx[reg] = max(mx[reg], v);
      if (cpart) {
        cz = max(min(cx, v), cz);
        cy = (v > cx) ? i0 + reg + 16 * h : cy;
        cx = max(cx, v);
      }
    }
    if (cpart) {  // rows of lane half 0 precede those of half 1
      const int ox = __shfl_xor(cx, 32), oy = __shfl_xor(cy, 32), oz = __shfl_xor(cz, 32);
      int3 lo = h ? make_int3(ox, oy, oz) : make_int3(cx, cy, cz);
      const int3 hi = h ? make_int3(cx, cy, cz) : make_int3(ox, oy, oz);
      col_merge(lo, hi);
      if (h == 0 && j < num2) cpart[(size_t)rb * num2 + j] = lo;
    }
  }
#pragma unroll
  for (int reg = 0; reg < 16; reg++) {
    const int row = i0 + reg + 16 * h;
    if (row < num1) rstate[((size_t)row * nseg + sg) * 32 + r] = make_int3(mx[reg], nx[reg], ix[reg]);
  }
}

// Rows: merge the per-segment thread states of each of the 32 strided threads in column order (a later
// segment only wins with a strictly larger maximum), then the reference's tree over the 32 threads.
__global__ __launch_bounds__(256) void match_rowmerge_kernel(const int3* rstate, int num1, int nseg, float distmax,
                                                             float ratiomax, int* rowm) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5), t = threadIdx.x & 31;
  const int rowc = min(row, num1 - 1);  // keep every lane in the shuffles below
  const int3* p = rstate + (size_t)rowc * nseg * 32 + t;
  int3 s = p[0];
  for (int q = 1; q < nseg; q++) {
    const int3 u = p[(size_t)q * 32];
    if (u.x > s.x) s = make_int3(u.x, max(s.x, u.y), u.z);
    else s.y = max(s.y, u.x);
  }
  int mx = s.x, nx = s.y, ix = s.z;
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {  // ProgramCU.cu:3766-3780: partner d away, a tie keeps the lower thread
    const int omx = __shfl_down(mx, d, 32), onx = __shfl_down(nx, d, 32), oix = __shfl_down(ix, d, 32);
    const bool take = omx > mx;
    const int nnx = take ? max(mx, onx) : max(nx, omx);
    ix = take ? oix : ix;
    mx = take ? omx : mx;
    nx = nnx;
  }
  if (t == 0 && row < num1) rowm[row] = decide(mx, nx, ix, distmax, ratiomax);
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
    M_TRY(m, hipMalloc(&m->des[index], (size_t)num * KD));
    M_TRY(m, hipMemcpy(m->des[index], des, (size_t)num * KD, hipMemcpyHostToDevice));
    // score offsets of the matrix-core path: rows 128*sum - 128^2*128, columns 128*sum (see match_mfma_kernel)
    std::vector<int> f((size_t)num);
    for (int i = 0; i < num; i++) {
      int sum = 0;
      for (int k = 0; k < KD; k++) sum += des[(size_t)i * KD + k];
      f[i] = 128 * sum - (index == 0 ? 128 * 128 * KD : 0);
    }
    M_TRY(m, hipMalloc(&m->fix[index], (size_t)num * sizeof(int)));
    M_TRY(m, hipMemcpy(m->fix[index], f.data(), (size_t)num * sizeof(int), hipMemcpyHostToDevice));
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
    // matrix-core path: enough (row block, column segment) wavefronts to fill the chip
    const int nrb = (n1 + 31) / 32, ntile2 = (n2 + 31) / 32;
    static const int target_waves = getenv("HESS_MATCH_WAVES") ? atoi(getenv("HESS_MATCH_WAVES")) : 2048;
    int nseg = (target_waves + nrb - 1) / nrb;
    nseg = nseg < 1 ? 1 : (nseg > ntile2 ? ntile2 : nseg);
    const int tiles_per_seg = (ntile2 + nseg - 1) / nseg;
    nseg = (ntile2 + tiles_per_seg - 1) / tiles_per_seg;
    const size_t need_rs = (size_t)n1 * nseg * 32, need_cp = (size_t)nrb * n2;
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
    hipLaunchKernelGGL(match_mfma_kernel, dim3(nrb, (nseg + 3) / 4), dim3(256), 0, m->st, m->des[0], n1, m->des[1], n2,
                       m->fix[0], m->fix[1], nseg, tiles_per_seg, mutual_best ? m->cpart2 : nullptr, m->rstate);
    hipLaunchKernelGGL(match_rowmerge_kernel, dim3((n1 + 7) / 8), dim3(256), 0, m->st, m->rstate, n1, nseg, distmax,
                       ratiomax, m->rowm);
    if (mutual_best)
      hipLaunchKernelGGL(match_col_kernel, dim3((n2 + 31) / 32), dim3(256), 0, m->st, m->cpart2, nrb, n2, distmax,
                         ratiomax, m->colm);
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
