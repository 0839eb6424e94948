// hess_match.hip -- descriptor matcher for gfx950: the step after the hot path.
//
// Replaces SiftMatchCU (SiftMatchCU.cpp:71-176) and its kernels MultiplyDescriptor(_G)_Kernel,
// RowMatch_Kernel, ColMatch_Kernel (ProgramCU.cu:3455-3843).  Integer work: results are bit-exact.
//
//   match_dot_kernel   64x64 tile of the num1 x num2 dot-product matrix per workgroup; both 8 KB
//                      descriptor panels staged in LDS, 4x4 outputs per thread, v_dot4_u32_u8 over the
//                      128-byte descriptors (the only dense contraction in the tree; at 4096 x 4096 it is
//                      2.1 GMAC, far below what would need MFMA); guided mode applies the reference's
//                      homography / fundamental-matrix gates per pair and its per-8-row-block rule;
//   match_row_kernel   one wavefront per row: best / second best with the reference's tie order (its
//                      32-thread tree: partners 16, 8, 4, 2, 1 apart, ties keep the lower thread), acos
//                      distance + ratio test;
//   match_col_kernel   one thread per column merges the per-tile (max, index, second) partials that the
//                      dot kernel's epilogue produced, in ascending row order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
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

// ColMatch_Kernel (ProgramCU.cu:3808-3827): merge the row-tile partials of a column in ascending order.
__global__ __launch_bounds__(256) void match_col_kernel(const int3* cpart, int ntile, int num2, float distmax,
                                                        float ratiomax, int* colm) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= num2) return;
  int3 t = cpart[j];
  for (int q = 1; q < ntile; q++) {
    const int3 u = cpart[(size_t)q * num2 + j];
    if (t.x < u.x) t = make_int3(u.x, u.y, max(t.x, u.z));
    else t.z = max(t.z, u.x);
  }
  colm[j] = decide(t.x, t.z, t.y, distmax, ratiomax);
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
  if (num) {
    M_TRY(m, hipMalloc(&m->des[index], (size_t)num * KD));
    M_TRY(m, hipMemcpy(m->des[index], des, (size_t)num * KD, hipMemcpyHostToDevice));
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
  const size_t need = (size_t)n1 * n2;
  if (need > m->mat_cap) {
    (void)hipFree(m->cpart); (void)hipFree(m->dotm); (void)hipFree(m->rowm); (void)hipFree(m->colm);
    m->cpart = nullptr;
    m->dotm = m->rowm = m->colm = nullptr;
    M_TRY(m, hipMalloc(&m->cpart, (size_t)((m->max_sift + TM - 1) / TM + 1) * m->max_sift * sizeof(int3)));
    M_TRY(m, hipMalloc(&m->dotm, need * sizeof(int)));
    M_TRY(m, hipMalloc(&m->rowm, (size_t)m->max_sift * sizeof(int) + 4));
    M_TRY(m, hipMalloc(&m->colm, (size_t)m->max_sift * sizeof(int) + 4));
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
    hipLaunchKernelGGL(match_col_kernel, dim3((n2 + 255) / 256), dim3(256), 0, m->st, m->cpart, (n1 + TM - 1) / TM, n2,
                       distmax, ratiomax, m->colm);
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
