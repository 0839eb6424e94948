/*
 * hess_oracle.c -- CPU ORACLE: plain-C restatement of the reference's Hessian + SIFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see hess_oracle.h).  PARITY: detector UNPINNED (the reference has no golden
 * vectors for the Hessian path and cannot be built here); orientation + descriptor stages pinned by the
 * reference's doc/evaluation/box.siftgpu; see hess_oracle.h and DESIGN.md.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/src/SiftGPU/).  Floating-point model: IEEE binary32, round-to-nearest;
 * this file must be compiled with -ffp-contract=off.  Device code of the reference is compiled
 * by nvcc whose default (-fmad=true) contracts a product feeding an add/sub into one FMA; this
 * is modelled with explicit fmaf() by one rule: "x*y + z", "z + x*y", "x*y - z", "z - x*y" with
 * the product a direct operand become a single fmaf; when both operands are products the LEFT
 * one is fused; association is left to right.  Host code of the reference (g++, x86-64) is
 * modelled without contraction.  Elementary functions: hess_math_ref.h.
 *
 * Deterministic ordering defined by this build (the reference's list order depends on
 * atomicAdd arrival, ProgramCU.cu:1044): within a level, detections are in row-major order
 * (row, then col); top-K ties at the cut are resolved towards the lower (level,row,col).
 */
#include "hess_oracle.h"
#include "hess_math_ref.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_OCT 32
#define MAX_LEV 16      /* dog_level_num <= 10 -> level_num <= 12 */
#define KMAXW 33        /* KERNEL_MAX_WIDTH, ProgramCU.cu:42 */
#define KMINW 5         /* KERNEL_MIN_WIDTH, ProgramCU.cu:43 */
#define PI_D 3.14159265358979323846 /* config.h:33 */

typedef struct {
  int w, wa, h; /* unaligned width, 4-aligned width, height (PyramidCU.cpp:274-309) */
} ogeom;

typedef struct {
  float* gauss[MAX_OCT][MAX_LEV];
  float* deth[MAX_OCT][MAX_LEV];
  float* got[MAX_OCT][MAX_LEV]; /* interleaved (grad, theta); levels 1..dog only */
} pyramid;

typedef struct {
  int n, nraw;
  hess_keypoint* keys;
  float* desc;
  hess_rawkey* raw;
  pyramid pyr;
  int have_pyr;
} image_result;

typedef struct { /* 16-byte device feature record, Appendix A.1 of SURVEY / ProgramCU.cu:1563-1596 */
  uint32_t x, y, z, w;
} frec;

struct hess_cpu_ctx {
  hess_params p;
  /* schedule (SiftGPU.cpp:491-563) */
  int level_num, level_max, level_ds;
  float sigma[MAX_LEV];        /* inter-level blur, _sigma[i] */
  float level_sigma[MAX_LEV];  /* GetLevelSigma(l) */
  float sigma_step, ln_sigma_step;
  /* geometry of the last run */
  int noct, ds;                /* octaves, input down-sample factor */
  int img_w, img_h;            /* after decimation and width truncation */
  ogeom g[MAX_OCT];
  int batch;
  image_result* res;
  int desc_dim;
  int threads, keep;
  /* user-supplied keypoint list for the next run (SiftPyramid::SetKeypointList) */
  hess_keypoint* user_keys;
  int user_num, user_have_orientation;
  int* user_levels;            /* analysis hook: explicit level index per user keypoint (hess_cpu_debug_key_levels) */
  float timing[HESS_T_COUNT];
  char err[256];
};

static double now_ms(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return tv.tv_sec * 1000.0 + tv.tv_usec / 1000.0;
}

/* ------------------------------------------------------------------------------------------ */
/* Parameters: GlobalUtil.cpp:51-144 defaults, SiftGPU.cpp:466-563 schedule.                  */

void hess_cpu_default_params(hess_params* p) {
  memset(p, 0, sizeof(*p));
  p->abi_version = HESS_ABI_VERSION;
  p->dog_level_num = 3;
  p->sigma0 = 1.6f;
  p->sigman = 0.5f;
  p->dog_threshold = 0.02f / 3;
  p->edge_threshold = 10.0f;
  p->filter_width_factor = 4.0f;
  p->orient_window_factor = 2.0f;
  p->orient_gaussian_factor = 1.5f;
  p->desc_window_factor = 3.0f;
  p->first_octave = 0;
  p->octave_num = -1;
  p->subpixel = 1;
  p->max_orientation = 2;
  p->compute_descriptors = 1;
  p->descriptor_order = HESS_DESC_ORDER_PIXEL;
  p->normalize = 1;
  p->truncate_method = HESS_TRUNC_HIGHEST_0;
  p->feature_count_threshold = -1;
  p->tex_max_dim = 3200;
}

static void resolve_params(hess_cpu_ctx* c) {
  hess_params* p = &c->p;
  /* SiftParam::ParseSiftParam, SiftGPU.cpp:491-563 (GPU_HESSIAN branch, _level_min = 0) */
  if (p->dog_level_num == 0) p->dog_level_num = 3;
  if (p->sigma0 == 0.0f) p->sigma0 = 1.6f;
  if (p->sigman == 0.0f) p->sigman = 0.5f;
  if (p->filter_width_factor == 0.0f) p->filter_width_factor = 4.0f;
  if (p->orient_window_factor == 0.0f) p->orient_window_factor = 2.0f;
  if (p->orient_gaussian_factor == 0.0f) p->orient_gaussian_factor = 1.5f;
  if (p->desc_window_factor == 0.0f) p->desc_window_factor = 3.0f;
  if (p->tex_max_dim == 0) p->tex_max_dim = 3200;
  if (p->max_orientation < 1) p->max_orientation = 1; /* SiftGPU.cpp:1047 clamps to 1..4 */
  if (p->max_orientation > 4) p->max_orientation = 4;
  float sigmak = powf(2.0f, 1.0f / p->dog_level_num);
  if (HESS_ORACLE_DETECTOR(p) == 0) {
    c->level_max = p->dog_level_num + 1;
    c->level_num = c->level_max + 1;
    c->level_ds = p->dog_level_num; /* _level_min + _dog_level_num, <= _level_max */
    float dsigma0 = p->sigma0 * sqrtf(sigmak * sigmak - 1.0f); /* SiftGPU.cpp:516 */
    for (int i = 1; i <= c->level_max; i++)
      c->sigma[i - 1] = dsigma0 * powf(sigmak, (float)(i - 1)); /* SiftGPU.cpp:547-552 */
    for (int l = 0; l <= c->level_max; l++) /* GetLevelSigma, SiftGPU.cpp:1422-1425 */
      c->level_sigma[l] = p->sigma0 * powf(2.0f, (float)l / (float)p->dog_level_num);
  } else {
    /* The build WITHOUT GPU_HESSIAN (difference of Gaussians; SiftGPU.cpp:466-556, #else branches): levels
     * _level_min = -1 .. _level_max = dog + 1 with _sigma0 = 1.6 * 2^(1/dog); this oracle's level j is that build's
     * level j - 1, so its levels 0 .. dog + 2 carry the same sigmas as the Hessian build's plus one more on top.
     * p->sigma0 keeps the Hessian convention (sigma of level 0). */
    c->level_max = p->dog_level_num + 2;
    c->level_num = c->level_max + 1;
    c->level_ds = p->dog_level_num; /* _level_min + _dog_level_num = dog - 1 there, + 1 here */
    float sigma0 = p->sigma0 * powf(2.0f, 1.0f / p->dog_level_num);                /* :503 */
    float dsigma0 = sigma0 * sqrtf(1.0f - 1.0f / (sigmak * sigmak));               /* :531 */
    for (int i = -1 + 1; i <= c->level_max - 1; i++)                               /* :544-555: _sigma[i+1-1] = dsigma0 * k^i */
      c->sigma[i] = dsigma0 * powf(sigmak, (float)i);
    for (int l = 0; l <= c->level_max; l++) /* GetLevelSigma(level = l - 1), :1422-1425 */
      c->level_sigma[l] = sigma0 * powf(2.0f, (float)(l - 1) / (float)p->dog_level_num);
    if (HESS_ORACLE_DETECTOR(p) == 2) {
      /* The keypoint scales of doc/evaluation/box.siftgpu are sigma0 * 2^(level / (2 dog)) * step^ds: the file was
       * written before the "bug fix 9/12/2007" that GetLevelSigma's comment records (measured on the file: the
       * ratio to today's formula is 2^(level/6) to four digits at each of the three levels).  Only the sigma handed
       * to the orientation stage is affected; the pyramid is not. */
      for (int l = 0; l <= c->level_max; l++)
        c->level_sigma[l] = sigma0 * powf(2.0f, (float)(l - 1) / (float)(2 * p->dog_level_num));
    }
  }
  if (p->dog_threshold == 0.0f) p->dog_threshold = 0.02f / p->dog_level_num; /* :558-559 */
  if (p->edge_threshold == 0.0f) p->edge_threshold = 10.0f;                   /* :561-562 */
  (void)sigmak;
  c->sigma_step = powf(2.0f, 1.0f / p->dog_level_num); /* PyramidCU.cpp:1821 */
  c->ln_sigma_step = (float)log((double)c->sigma_step);
}

/* SiftParam::GetInitialSmoothSigma, SiftGPU.cpp:482-489 (_level_min = 0). */
static float initial_smooth_sigma(const hess_cpu_ctx* c, int octave_min) {
  /* sa = _sigma0 * 2^(_level_min / dog): level_min = 0 (Hessian) or _sigma0 = 1.6 * 2^(1/dog), level_min = -1 (DoG) */
  float sa = HESS_ORACLE_DETECTOR(&c->p) == 0
                 ? c->p.sigma0 * powf(2.0f, 0.0f / (float)c->p.dog_level_num)
                 : (c->p.sigma0 * powf(2.0f, 1.0f / c->p.dog_level_num)) * powf(2.0f, -1.0f / (float)c->p.dog_level_num);
  float sb = c->p.sigman / powf(2.0f, (float)octave_min);
  return (sa > sb + 0.001) ? sqrtf(sa * sa - sb * sb) : 0.0f;
}

/* ProgramCU::CreateFilterKernel, ProgramCU.cu:423-453 (host code; libm expf). */
/* Sample-window clamps at the image border.  The CUDA path (the product's rule, and the unpacked GLSL shaders' too:
 * ProgramCU.cu:1324-1332,1723-1731, ProgramGLSL.cpp:1242-1244,1617-1619) keeps sample centres in [1.5, dim-1.5].
 * HESS_ORACLE_BORDER(p) == 1 (analysis only, tests/test_reference_fixture.py) is the rule of the PACKED GLSL shaders
 * (ProgramGLSL.cpp:1883-1885,2579-2581: the box is clamped to [2, dim-3] and then widened to whole 2x2 texels), the
 * default backend of the SiftGPU versions that could have written doc/evaluation/box.siftgpu. */
static inline float border_lo(const hess_cpu_ctx* c, float v) {
  if (HESS_ORACLE_BORDER(&c->p) == 1) return 2.0f * floorf(fmaxf(v, 2.0f) * 0.5f) + 0.5f;
  return fmaxf(1.5f, floorf(v) + 0.5f);
}
static inline float border_hi(const hess_cpu_ctx* c, float v, int dim) {
  if (HESS_ORACLE_BORDER(&c->p) == 1) return 2.0f * floorf(fminf(v, (float)dim - 3.0f) * 0.5f) + 1.5f;
  return fminf(dim - 1.5f, floorf(v) + 0.5f);
}

static int create_filter_kernel(const hess_cpu_ctx* c, float sigma, float* kernel) {
  int i, sz = (int)ceil(c->p.filter_width_factor * sigma - 0.5);
  int width = 2 * sz + 1;
  if (width > KMAXW) { sz = KMAXW >> 1; width = KMAXW; }
  else if (width < KMINW) { sz = KMINW >> 1; width = KMINW; }
  float rv = 1.0f / (sigma * sigma), v, ksum = 0;
  for (i = -sz; i <= sz; ++i) {
    kernel[i + sz] = v = expf(-0.5f * i * i * rv);
    ksum += v;
  }
  rv = 1.0f / ksum;
  for (i = 0; i < width; i++) kernel[i] *= rv;
  return width;
}

int hess_cpu_filter_taps(hess_cpu_ctx* c, int level, float* taps) {
  float s = (level == 0) ? initial_smooth_sigma(c, c->ds) : c->sigma[level - 1];
  if (level < 0 || level > c->level_max) return 0;
  if (s <= 0.0f) return 0;
  return create_filter_kernel(c, s, taps);
}
float hess_cpu_level_sigma(hess_cpu_ctx* c, int level) { return c->level_sigma[level]; }

/* ------------------------------------------------------------------------------------------ */
/* Input stage: GLTexInput::SetImageData CUDA branch + DownSamplePixelData*, GLTexImage.cpp:802-1036 */

static int fmt_channels(int format) {
  switch (format) {
    case HESS_FMT_LUM: return 1;
    case HESS_FMT_LUM_ALPHA: return 2;
    case HESS_FMT_RGB: case HESS_FMT_BGR: return 3;
    case HESS_FMT_RGBA: case HESS_FMT_BGRA: return 4;
  }
  return 0;
}

/* One converted pixel; p points at the first channel. */
static inline float convert_pixel(const void* p, int format, int pixtype) {
  if (pixtype == HESS_PIX_F32) {
    const float* f = (const float*)p;
    switch (format) { /* DownSamplePixelDataF, GLTexImage.cpp:864-916 (host: no contraction) */
      case HESS_FMT_LUM: case HESS_FMT_LUM_ALPHA: return f[0];
      case HESS_FMT_RGB: case HESS_FMT_RGBA: return (0.299f * f[0] + 0.587f * f[1] + 0.114f * f[2]);
      default: return (0.114f * f[0] + 0.587f * f[1] + 0.299f * f[2]);
    }
  }
  unsigned v0, v1 = 0, v2 = 0;
  float factor;
  int lum = (format == HESS_FMT_LUM || format == HESS_FMT_LUM_ALPHA);
  if (pixtype == HESS_PIX_U8) {
    const unsigned char* u = (const unsigned char*)p;
    v0 = u[0]; if (!lum) { v1 = u[1]; v2 = u[2]; }
    factor = 255.0f;
  } else {
    const unsigned short* u = (const unsigned short*)p;
    v0 = u[0]; if (!lum) { v1 = u[1]; v2 = u[2]; }
    factor = 65535.0f;
  }
  /* DownSamplePixelDataI2F, GLTexImage.cpp:802-862: integer numerator (int arithmetic, as the
   * promoted unsigned char/short operands are), float divide. */
  if (lum) return (float)(int)v0 / factor;
  /* (int arithmetic wraps for bright 16-bit RGB in the reference too: 65536*65535 > INT_MAX) */
  if (format == HESS_FMT_RGB || format == HESS_FMT_RGBA)
    return (float)(int32_t)(19595u * v0 + 38470u * v1 + 7471u * v2) / (65535.0f * factor);
  return (float)(int32_t)(7471u * v0 + 38470u * v1 + 19595u * v2) / (65535.0f * factor);
}

/* Scale of the first octave relative to the input image: 2^_octave_min (PyramidCU.cpp:746-748,
 * 1054-1057, 566-569): 1 << ds for a decimated input, 1 / (1 << -ds) for an up-sampled one. */
static float first_octave_sigma(const hess_cpu_ctx* c) {
  if (c->ds > 0) return (float)(1 << c->ds);
  if (c->ds < 0) return 1.0f / (float)(1 << (-c->ds));
  return 1.0f;
}

/* UpsampleKernel<LOG_SCALE>, ProgramCU.cu:233-285 (SampleImageU :288-310): linear interpolation by
 * 2^log_scale in both directions; the source is fetched by 1-D index (index+1 at a row end is the next
 * row's first pixel, an index past the plane reads 0).  Only reachable with -fo < 0, which the
 * reference's Hessian build refuses at the option parser (SiftGPU.cpp:1166-1167) although the pyramid
 * code below it keeps the path (PyramidCU.cpp:1517-1525): the oracle has it so that the reference's own
 * feature file (doc/evaluation/box.siftgpu, made with -fo -1) can be reproduced stage by stage. */
static inline float tex1(const float* t, long n, long i);
static void upsample_image(const float* src, int width, int height, int log_scale, float* dst) {
  const int SCALE = 1 << log_scale, SCALE_MASK = SCALE - 1;
  const float INV_SCALE = 1.0f / (float)SCALE;
  const long n = (long)width * height;
#pragma omp parallel for schedule(static)
  for (int dst_row = 0; dst_row < (height << log_scale); dst_row++) {
    int row = dst_row >> log_scale;
    int helper = dst_row & SCALE_MASK;
    for (int col = 0; col < width; col++) {
      long index = (long)row * width + col;
      long dst_idx = ((long)width * dst_row + col) * SCALE;
      float v1, v2;
      if (helper) {
        float v11 = tex1(src, n, index), v12 = tex1(src, n, index + 1);
        float v21 = tex1(src, n, index + width), v22 = tex1(src, n, index + width + 1);
        float w1 = INV_SCALE * helper, w2 = (float)(1.0 - w1);
        v1 = fmaf(v21, w1, w2 * v11);
        v2 = fmaf(v22, w1, w2 * v12);
      } else {
        v1 = tex1(src, n, index);
        v2 = tex1(src, n, index + 1);
      }
      dst[dst_idx] = v1;
      for (int i = 1; i < SCALE; ++i) {
        const float r2 = i * INV_SCALE;
        const float r1 = 1.0f - r2;
        dst[dst_idx + i] = fmaf(v1, r1, v2 * r2);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Pyramid: FilterH/FilterV ProgramCU.cu:117-231, DownsampleKernel :312-326, BuildPyramid
 * PyramidCU.cpp:1486-1558.                                                                    */

static void filter_image(const hess_cpu_ctx* c, float* dst, const float* src, float* buf, int w, int h,
                         const float* k, int fw) {
  int half = fw >> 1;
  (void)c;
  /* Per output pixel the taps are accumulated in the order i = 0..fw-1 starting from 0
   * (ProgramCU.cu:150-153, :224-227); the loops below keep that order per pixel and only
   * interchange it with the pixel loop so that the compiler can vectorise across pixels. */
#pragma omp parallel
  {
    float* pad = (float*)malloc((size_t)(w + 2 * half) * sizeof(float));
    float* acc = (float*)malloc((size_t)w * sizeof(float));
#pragma omp for schedule(static)
    for (int r = 0; r < h; r++) {
      const float* s = src + (size_t)r * w;
      float* b = buf + (size_t)r * w;
      for (int x = 0; x < half; x++) pad[x] = s[0];               /* replicate, ProgramCU.cu:138 */
      memcpy(pad + half, s, (size_t)w * sizeof(float));
      for (int x = 0; x < half; x++) pad[half + w + x] = s[w - 1];
      for (int x = 0; x < w; x++) acc[x] = 0.0f;
      for (int i = 0; i < fw; i++) {
        const float ki = k[i];
        const float* pi = pad + i;
        for (int x = 0; x < w; x++) acc[x] = fmaf(pi[x], ki, acc[x]); /* ProgramCU.cu:152 */
      }
      memcpy(b, acc, (size_t)w * sizeof(float));
    }
#pragma omp for schedule(static)
    for (int r = 0; r < h; r++) {
      float* d = dst + (size_t)r * w;
      for (int x = 0; x < w; x++) acc[x] = 0.0f;
      for (int i = 0; i < fw; i++) {
        int rr = r - half + i;
        rr = rr < 0 ? 0 : (rr > h - 1 ? h - 1 : rr);              /* ProgramCU.cu:201 */
        const float ki = k[i];
        const float* br = buf + (size_t)rr * w;
        for (int x = 0; x < w; x++) acc[x] = fmaf(br[x], ki, acc[x]); /* ProgramCU.cu:226 */
      }
      memcpy(d, acc, (size_t)w * sizeof(float));
    }
    free(pad);
    free(acc);
  }
}

/* ComputeHessian_Kernel, ProgramCU.cu:523-595.  Neighbours are fetched by 1-D index from a
 * linear texture bound to the level: an index outside [0, w*h) reads 0, index +-1 at a row end
 * wraps to the neighbouring row (SURVEY 7.2 "Border semantics"). */
static inline float tex1(const float* t, long n, long i) { return (i < 0 || i >= n) ? 0.0f : t[i]; }

static void compute_hessian(const float* gus, float* deth, float* got, int w, int h, float norm2) {
  long n = (long)w * h;
  float norm = norm2 * norm2; /* ProgramCU.cu:592: kernel receives norm*norm, host passes sigma^2 */
#pragma omp parallel for schedule(static)
  for (int row = 0; row < h; row++) {
    for (int col = 0; col < w; col++) {
      long idx = (long)row * w + col;
      float v11 = tex1(gus, n, idx - w - 1), v12 = tex1(gus, n, idx - w), v13 = tex1(gus, n, idx - w + 1);
      float v21 = tex1(gus, n, idx - 1), v22 = tex1(gus, n, idx), v23 = tex1(gus, n, idx + 1);
      float v31 = tex1(gus, n, idx + w - 1), v32 = tex1(gus, n, idx + w), v33 = tex1(gus, n, idx + w + 1);
      float Lxx = (fmaf(-2.0f, v22, v21) + v23);
      float Lyy = (fmaf(-2.0f, v22, v12) + v32);
      float Lxy = (v13 - v11 + v31 - v33) * 0.25f;
      deth[idx] = fmaf(Lxx, Lyy, -(Lxy * Lxy)) * norm; /* ProgramCU.cu:553 */
      if (got) {
        float dx = v23 - v21;
        float dy = v32 - v12;
        float gradient = 0.5f * sqrtf(fmaf(dx, dx, dy * dy));
        float rot = (gradient == 0.0f) ? 0.0f : om_atan2f(dy, dx);
        got[2 * idx] = gradient;
        got[2 * idx + 1] = rot;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* ComputeKEY_Kernel, ProgramCU.cu:657-882.  Returns 1 and fills *out when (col,row) is a
 * keypoint.  texC/texP/texN = det-H of the level, previous, next; texG = Gaussian level. */
typedef struct { uint32_t packed; float dx, dy, ds; } keyval;

#define READ_CMP(datai, tex, idx)                                      \
  datai[0] = tex[(idx) - 1]; datai[1] = tex[(idx)]; datai[2] = tex[(idx) + 1]; \
  if (response > nmax) {                                               \
    nmax = fmaxf(nmax, datai[0]); nmax = fmaxf(nmax, datai[1]); nmax = fmaxf(nmax, datai[2]); \
    if ((response < nmax) || (response < 0)) return 0;                 \
  } else {                                                             \
    nmin = fminf(nmin, datai[0]); nmin = fminf(nmin, datai[1]); nmin = fminf(nmin, datai[2]); \
    if ((response > nmin) || (response > 0)) return 0;                 \
  }

/* READ_CMP_DOG_DATA without GPU_HESSIAN (ProgramCU.cu:680-699): no sign tests. */
#define READ_CMP_DOG(datai, tex, idx)                                  \
  datai[0] = tex[(idx) - 1]; datai[1] = tex[(idx)]; datai[2] = tex[(idx) + 1]; \
  if (response > nmax) {                                               \
    nmax = fmaxf(nmax, datai[0]); nmax = fmaxf(nmax, datai[1]); nmax = fmaxf(nmax, datai[2]); \
    if (response < nmax) return 0;                                     \
  } else {                                                             \
    nmin = fminf(nmin, datai[0]); nmin = fminf(nmin, datai[1]); nmin = fminf(nmin, datai[2]); \
    if (response > nmin) return 0;                                     \
  }
#define READ_ANY(datai, tex, idx) \
  if (dogmode) { READ_CMP_DOG(datai, tex, idx) } else { READ_CMP(datai, tex, idx) }

/* dogmode: the kernel as compiled without GPU_HESSIAN (the planes are differences of Gaussians, the extremum test
 * has no sign condition, the "type" is the sign of the extremum, :853-854). */
static int compute_key(const float* texC, const float* texP, const float* texN, const float* texG,
                       int width, int row, int col, float thr0, float thr, float edge_thr,
                       int subpixel, int dogmode, keyval* out) {
  float data[3][3], datap[3][3], datan[3][3];
  float response, nmax, nmin;
  float dx = 0, dy = 0, ds = 0;
  int offset_test_passed = 1;
  long index = (long)row * width + col;
  long idx[3] = {index - width, index, index + width};

  data[1][1] = response = texC[idx[1]];
  if (fabsf(response) <= thr0) return 0;
  data[1][0] = texC[idx[1] - 1];
  data[1][2] = texC[idx[1] + 1];
  nmax = fmaxf(data[1][0], data[1][2]);
  nmin = fminf(data[1][0], data[1][2]);
  if ((response <= nmax) && (response >= nmin)) return 0;
  READ_ANY(data[0], texC, idx[0]);
  READ_ANY(data[2], texC, idx[2]);

  /* edge suppression, ProgramCU.cu:748-757 */
  float vx2 = response * 2.0f;
  float fxx = data[1][0] + data[1][2] - vx2;
  float fyy = data[0][1] + data[2][1] - vx2;
  float fxy = 0.25f * (data[2][2] + data[0][0] - data[2][0] - data[0][2]);
  float temp1 = fmaf(fxx, fyy, -(fxy * fxy));
  float temp2 = (fxx + fyy) * (fxx + fyy);
  if ((temp1 <= 0) || (temp2 > edge_thr * temp1)) return 0;

  READ_ANY(datap[0], texP, idx[0]);
  READ_ANY(datap[1], texP, idx[1]);
  READ_ANY(datap[2], texP, idx[2]);
  READ_ANY(datan[0], texN, idx[0]);
  READ_ANY(datan[1], texN, idx[1]);
  READ_ANY(datan[2], texN, idx[2]);

  if (subpixel) { /* ProgramCU.cu:769-825 */
    float fx = 0.5f * (data[1][2] - data[1][0]);
    float fy = 0.5f * (data[2][1] - data[0][1]);
    float fs = 0.5f * (datan[1][1] - datap[1][1]);
    float fss = (datan[1][1] + datap[1][1] - vx2);
    float fxs = 0.25f * (datan[1][2] + datap[1][0] - datan[1][0] - datap[1][2]);
    float fys = 0.25f * (datan[2][1] + datap[0][1] - datan[0][1] - datap[2][1]);
    float A0[4], A1[4], A2[4], T[4];
    if (fxx > 0) { A0[0] = fxx; A0[1] = fxy; A0[2] = fxs; A0[3] = -fx; }
    else { A0[0] = -fxx; A0[1] = -fxy; A0[2] = -fxs; A0[3] = fx; }
    if (fxy > 0) { A1[0] = fxy; A1[1] = fyy; A1[2] = fys; A1[3] = -fy; }
    else { A1[0] = -fxy; A1[1] = -fyy; A1[2] = -fys; A1[3] = fy; }
    if (fxs > 0) { A2[0] = fxs; A2[1] = fys; A2[2] = fss; A2[3] = -fs; }
    else { A2[0] = -fxs; A2[1] = -fys; A2[2] = -fss; A2[3] = fs; }
    float maxa = fmaxf(fmaxf(A0[0], A1[0]), A2[0]);
    if (maxa >= 1e-10) {
      if (maxa == A1[0]) { memcpy(T, A1, 16); memcpy(A1, A0, 16); memcpy(A0, T, 16); }
      else if (maxa == A2[0]) { memcpy(T, A2, 16); memcpy(A2, A0, 16); memcpy(A0, T, 16); }
      A0[1] /= A0[0]; A0[2] /= A0[0]; A0[3] /= A0[0];
      A1[1] = fmaf(-A1[0], A0[1], A1[1]); A1[2] = fmaf(-A1[0], A0[2], A1[2]); A1[3] = fmaf(-A1[0], A0[3], A1[3]);
      A2[1] = fmaf(-A2[0], A0[1], A2[1]); A2[2] = fmaf(-A2[0], A0[2], A2[2]); A2[3] = fmaf(-A2[0], A0[3], A2[3]);
      if (fabsf(A2[1]) > fabsf(A1[1])) { memcpy(T, A2, 16); memcpy(A2, A1, 16); memcpy(A1, T, 16); }
      if (fabsf(A1[1]) >= 1e-10) {
        A1[2] /= A1[1]; A1[3] /= A1[1];
        A2[2] = fmaf(-A2[1], A1[2], A2[2]); A2[3] = fmaf(-A2[1], A1[3], A2[3]);
        if (fabsf(A2[2]) >= 1e-10) {
          ds = A2[3] / A2[2];
          dy = fmaf(-ds, A1[2], A1[3]);
          dx = fmaf(-dy, A0[1], fmaf(-ds, A0[2], A0[3]));
          response = fmaf(0.5f, fmaf(ds, fs, fmaf(dx, fx, dy * fy)), data[1][1]);
          offset_test_passed = (fabsf(response) > thr) && (fabsf(ds) < 1.0f) && (fabsf(dx) < 1.0f) &&
                               (fabsf(dy) < 1.0f);
        }
      }
    }
  }
  if (!offset_test_passed) return 0;

  unsigned type; /* ProgramCU.cu:828-851 */
  if (dogmode) type = (response > nmax) ? HESS_TYPE_BRIGHT_BLOB : HESS_TYPE_DARK_BLOB; /* result = +-1, :853-854 */
  else if (response < 0) type = HESS_TYPE_SADDLE;
  else {
    float g0 = texG[idx[1] - 1], g1 = texG[idx[1]], g2 = texG[idx[1] + 1];
    float Lxx = fmaf(-2.0f, g1, g0) + g2;
    type = (Lxx > 0) ? HESS_TYPE_DARK_BLOB : HESS_TYPE_BRIGHT_BLOB;
  }
  out->packed = (((uint32_t)om_f2h(response)) << 16) | 0x4u | type; /* ProgramCU.cu:865 */
  out->dx = dx; out->dy = dy; out->ds = ds;
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* ComputeOrientation_Kernel, ProgramCU.cu:1221-1605 (detection mode, existing_keypoint = 0). */

#define FLOAT_TO_FIXED(v, n) ((int)((double)((v) * (float)(1 << (n))) + (((v) >= 0.0) ? 0.5 : -0.5)))
#define FIXED_TO_FLOAT(v, n) ((float)(v) / (1 << (n)))

static void compute_orientation(const hess_cpu_ctx* c, const hess_rawkey* rk, const float* got, int width,
                                int height, float sigma, frec* out) {
  const float ten_degree_per_radius = 5.7295779513082320876798154814105f;
  const float radius_per_ten_degrees = (float)(1.0 / 5.7295779513082320876798154814105);
  const hess_params* p = &c->p;
  float gaussian_factor = p->orient_gaussian_factor;
  float sample_factor = p->orient_gaussian_factor * p->orient_window_factor; /* ProgramCU.cu:1638 */
  int num_orientation = p->fixed_orientation ? 0 : p->max_orientation;
  float kx = rk->col + 0.5f, ky = rk->row + 0.5f, kz = sigma, kw = 0.0f;
  uint32_t kw_bits = 0;
  int orientationsCount = 0;
  if (p->subpixel) { /* ProgramCU.cu:1293-1298 */
    kx += rk->dx;
    ky += rk->dy;
    kz *= om_powf_ln(c->ln_sigma_step, rk->ds);
  }
  uint32_t additional = rk->packed;

  if (num_orientation != 0) {
    float vote[37];
    float gsigma = kz * gaussian_factor;
    float win = fabsf(kz) * sample_factor;
    float dist_threshold = win * win + 0.5f;
    float factor = -0.5f / (gsigma * gsigma);
    float xmin = border_lo(c, kx - win);
    float ymin = border_lo(c, ky - win);
    float xmax = border_hi(c, kx + win, width);
    float ymax = border_hi(c, ky + win, height);
    for (int i = 0; i < 36; ++i) vote[i] = 0.0f;
    for (float y = ymin; y <= ymax; y += 1.0f) {
      float dy = y - ky;
      dy *= dy;
      for (float x = xmin; x <= xmax; x += 1.0f) {
        float dx = x - kx;
        float sq_dist = fmaf(dx, dx, dy);
        if (sq_dist >= dist_threshold) continue;
        const float* g = got + 2 * ((long)(int)y * width + (int)x); /* tex2D point fetch */
        int oidx = (int)floorf(g[1] * ten_degree_per_radius);
        if (oidx < 0) oidx += 36;
        vote[oidx] = fmaf(g[0], om_expf(sq_dist * factor), vote[oidx]);
      }
    }
    const float one_third = (float)(1.0 / 3.0);
    for (int i = 0; i < 6; ++i) { /* ProgramCU.cu:1364-1379 */
      vote[36] = vote[0];
      float pre = vote[35];
      for (int j = 0; j < 36; ++j) {
        float temp = one_third * (pre + vote[j] + vote[j + 1]);
        pre = vote[j];
        vote[j] = temp;
      }
    }
    vote[36] = vote[0];
    if (p->half_sift) { /* ProgramCU.cu:1384-1392; note vote[36] keeps the pre-fold vote[0] */
      for (int i = 0; i < 18; i++) { vote[i] += vote[i + 18]; vote[i + 18] = 0; }
    }
    if (num_orientation == 1) { /* ProgramCU.cu:1398-1420 */
      int index_max = 0;
      float max_vote = vote[0];
      for (int i = 1; i < 36; ++i) {
        index_max = (vote[i] > max_vote) ? i : index_max;
        max_vote = fmaxf(max_vote, vote[i]);
      }
      float pre = vote[(index_max == 0) ? 35 : index_max - 1];
      float next = vote[index_max + 1];
      float weight = max_vote;
      float off = 0.5f * ((next - pre) / (weight + weight - next - pre));
      kw = radius_per_ten_degrees * (index_max + 0.5f + off);
      kw_bits = om_f2u(kw);
    } else if (HESS_ORACLE_DETECTOR(p) != 0) {
      /* build without GPU_HESSIAN, ProgramCU.cu:1493-1548: the two strongest peaks, 16-bit angles, 65535 = none */
      float max_vote = vote[0];
      for (int i = 1; i < 36; ++i) max_vote = fmaxf(max_vote, vote[i]);
      float vote_threshold = max_vote * 0.8f;
      float pre = vote[35];
      float max_rot[2] = {0, 0}, max_vot[2] = {0, 0};
      int ocount = 0;
      for (int i = 0; i < 36; ++i) {
        float next = vote[i + 1];
        if ((vote[i] > vote_threshold) && (vote[i] > pre) && (vote[i] > next)) {
          float di = 0.5f * ((next - pre) / (vote[i] + vote[i] - next - pre));
          float rot = i + di + 0.5f;
          float weight = vote[i];
          if (weight > max_vot[1]) {
            if (weight > max_vot[0]) {
              max_vot[1] = max_vot[0]; max_rot[1] = max_rot[0];
              max_vot[0] = weight; max_rot[0] = rot;
            } else {
              max_vot[1] = weight; max_rot[1] = rot;
            }
            ocount++;
          }
        }
        pre = vote[i];
      }
      float fr1 = max_rot[0] / 36.0f;
      if (fr1 < 0) fr1 += 1.0f;
      uint32_t us1 = (ocount == 0) ? 65535u : (uint32_t)(unsigned short)floor(fr1 * 65535.0f);
      uint32_t us2 = 65535u;
      if (ocount > 1) {
        float fr2 = max_rot[1] / 36.0f;
        if (fr2 < 0) fr2 += 1.0f;
        us2 = (uint32_t)(unsigned short)floor(fr2 * 65535.0f);
      }
      kw_bits = (us2 << 16) | us1;
      orientationsCount = (us1 != 65535u) + (us2 != 65535u); /* what ReshapeFeatureListCPU expands, PyramidCU.cpp:800-820 */
    } else { /* ProgramCU.cu:1424-1489 */
      float max_vote = vote[0];
      for (int i = 1; i < 36; ++i) max_vote = fmaxf(max_vote, vote[i]);
      float vote_threshold = max_vote * 0.8f;
      float pre = vote[35];
      float max_vot[5], max_rot[5];
      for (int i = 0; i < 36; ++i) {
        float next = vote[i + 1];
        if ((vote[i] > vote_threshold) && (vote[i] > pre) && (vote[i] > next)) {
          float di = 0.5f * ((next - pre) / (vote[i] + vote[i] - next - pre));
          float rot = i + di + 0.5f;
          float weight = vote[i];
          int idx = orientationsCount;
          if (orientationsCount > 0) {
            while ((idx > 0) && (max_vot[idx - 1] < weight)) {
              max_vot[idx] = max_vot[idx - 1];
              max_rot[idx] = max_rot[idx - 1];
              idx--;
            }
          }
          max_vot[idx] = weight;
          max_rot[idx] = rot;
          if (orientationsCount < 4) orientationsCount++;
        }
        pre = vote[i];
      }
      uint32_t packed = 0;
      int maxCount = orientationsCount < 4 ? orientationsCount : 4;
      for (int idx = 0; idx < maxCount; idx++) {
        float orientation = max_rot[idx] / 36.0f;
        if (orientation < 0) orientation += 1.0f;
        uint32_t ui = (uint32_t)floorf(orientation * 255.0f);
        packed = packed | (ui << 8 * idx);
      }
      kw_bits = packed;
    }
  } else {
    kw_bits = om_f2u(0.0f);
  }
  (void)kw;
  /* key_store_finish, ProgramCU.cu:1563-1596 */
  uint32_t posX = (uint32_t)FLOAT_TO_FIXED(kx, 10) & 0x00FFFFFFu;
  uint32_t posY = (uint32_t)FLOAT_TO_FIXED(ky, 10) & 0x00FFFFFFu;
  posX |= (additional & 0xFF000000u);
  posY |= ((additional << 8) & 0xFF000000u);
  uint32_t scale = (uint32_t)FLOAT_TO_FIXED(kz, 8) & 0x0000FFFFu;
  scale |= ((additional & 0x3u) << 30) | (((uint32_t)orientationsCount & 0x7u) << 27);
  out->x = posX; out->y = posY; out->z = scale; out->w = kw_bits;
}

/* ComputeOrientation_Kernel with existing_keypoint = 1 (ProgramCU.cu:1246-1278,1398-1420,1597-1602):
 * position and scale come from the packed record, sub-pixel offsets are not applied, only the
 * strongest orientation is kept as a float angle in record.w; x, y, z stay as uploaded. */
static void compute_orientation_existing(const hess_cpu_ctx* c, frec* rec, const float* got, int width,
                                         int height) {
  const float ten_degree_per_radius = 5.7295779513082320876798154814105f;
  const float radius_per_ten_degrees = (float)(1.0 / 5.7295779513082320876798154814105);
  const hess_params* p = &c->p;
  int num_orientation = p->fixed_orientation ? 0 : p->max_orientation;
  float kx = FIXED_TO_FLOAT(rec->x & 0x00FFFFFFu, 10);
  float ky = FIXED_TO_FLOAT(rec->y & 0x00FFFFFFu, 10);
  float kz = FIXED_TO_FLOAT(rec->z & 0x0000FFFFu, 8);
  if (num_orientation == 0) { rec->w = om_f2u(0.0f); return; }
  float vote[37];
  float gsigma = kz * p->orient_gaussian_factor;
  float win = fabsf(kz) * (p->orient_gaussian_factor * p->orient_window_factor);
  float dist_threshold = win * win + 0.5f;
  float factor = -0.5f / (gsigma * gsigma);
  float xmin = border_lo(c, kx - win);
  float ymin = border_lo(c, ky - win);
  float xmax = border_hi(c, kx + win, width);
  float ymax = border_hi(c, ky + win, height);
  for (int i = 0; i < 36; ++i) vote[i] = 0.0f;
  for (float y = ymin; y <= ymax; y += 1.0f) {
    float dy = y - ky;
    dy *= dy;
    for (float x = xmin; x <= xmax; x += 1.0f) {
      float dx = x - kx;
      float sq_dist = fmaf(dx, dx, dy);
      if (sq_dist >= dist_threshold) continue;
      const float* g = got + 2 * ((long)(int)y * width + (int)x);
      int oidx = (int)floorf(g[1] * ten_degree_per_radius);
      if (oidx < 0) oidx += 36;
      vote[oidx] = fmaf(g[0], om_expf(sq_dist * factor), vote[oidx]);
    }
  }
  const float one_third = (float)(1.0 / 3.0);
  for (int i = 0; i < 6; ++i) {
    vote[36] = vote[0];
    float pre = vote[35];
    for (int j = 0; j < 36; ++j) {
      float temp = one_third * (pre + vote[j] + vote[j + 1]);
      pre = vote[j];
      vote[j] = temp;
    }
  }
  vote[36] = vote[0];
  if (p->half_sift)
    for (int i = 0; i < 18; i++) { vote[i] += vote[i + 18]; vote[i + 18] = 0; }
  int index_max = 0;
  float max_vote = vote[0];
  for (int i = 1; i < 36; ++i) {
    index_max = (vote[i] > max_vote) ? i : index_max;
    max_vote = fmaxf(max_vote, vote[i]);
  }
  float pre = vote[(index_max == 0) ? 35 : index_max - 1];
  float next = vote[index_max + 1];
  float weight = max_vote;
  float off = 0.5f * ((next - pre) / (weight + weight - next - pre));
  rec->w = om_f2u(radius_per_ten_degrees * (index_max + 0.5f + off));
}

/* ------------------------------------------------------------------------------------------ */
/* ComputeDescriptor_Kernel<-di,HALF>, ProgramCU.cu:1650-1804 + NormalizeDescriptor_Kernel
 * :1950-2054.  `angle` is the un-mirrored float orientation handed to the kernel (key.w).    */

static void normalize_descriptor(const hess_cpu_ctx* c, float* d);

/* HESS_DESC_ORDER_PIXEL (include/hess_abi.h; this build's order, NOT the reference's): every pixel of the keypoint's
 * footprint is visited ONCE.  In the keypoint frame -- (u, v) = R(-angle) (pixel - keypoint) / spt, the 4 x 4 cell
 * centres at -1.5 .. 1.5 -- the quantities the reference recomputes per (pixel, cell) pair from the cell's own rounded
 * centre are per-pixel: nx + ix - 1.5 = u, so the Gaussian weight exp(-(u^2 + v^2) / 8) (ProgramCU.cu:1745-1748), the
 * bin coordinate theta (:1750-1753) and the bilinear cell weights 1 - |nx|, 1 - |ny| = the split of u + 1.5, v + 1.5
 * between the two nearest cell indices.  A pixel then adds weight * wy_j * wx_i * wbin_k to (at most) 2 x 2 cells x 2
 * bins.  The sums are kept in 32-bit FIXED POINT with a per-keypoint power-of-two scale 2^sh (integer addition is
 * associative: the result does not depend on the order of the pixels, so any parallel schedule gives these bits); each
 * of the eight products is rounded to an integer by (uint32)(fma(b, w, 0.5)).  2^sh: a bin cannot exceed 0.7072 (the
 * largest gradient/2 of luminance in [0, 1]) x the lattice sum of the cell's bilinear window (spt^2 to a fraction of
 * a percent, < (spt + 1)^2: tools/r05/lattice_sum.py); with 0.75 (spt + 1)^2 < 2^e and sh = 32 - e the sums stay
 * below 2^32.  The quantisation costs <= 2.2e-7 on unit-norm descriptors against sums in units of 2^-32 (measured).
 * Against the reference's order (sequential float sums) the results differ by rounding -- of the per-pixel quantities
 * (u instead of nx + offx) and of that order's own float arithmetic: for the worst feature of 640-2.jpg a float64
 * evaluation of the reference's formula is 7.9e-6 from the sequential float order and 3.6e-7 from this one.  Tests
 * bound the distance between the two by 1e-5 on unit-norm descriptors (north star 1e-4). */
static inline uint32_t f2u_sat(float v) { /* v_cvt_u32_f32: truncation, saturating, NaN -> 0 */
  if (!(v > 0.0f)) return 0u;
  return v >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)v;
}

static void compute_descriptor_pixel(const hess_cpu_ctx* c, const frec* rec, float angle, const float* got,
                                     int width, int height, float* d /* 128 or 64 */) {
  const float rpi = (float)(4.0 / PI_D);
  int half = c->p.half_sift;
  float kx = FIXED_TO_FLOAT(rec->x & 0x00FFFFFFu, 10);
  float ky = FIXED_TO_FLOAT(rec->y & 0x00FFFFFFu, 10);
  float kz = FIXED_TO_FLOAT(rec->z & 0x0000FFFFu, 8);
  float kw = angle;
  float spt = fabsf(kz * c->p.desc_window_factor);
  float s, co;
  om_sincosf(kw, &s, &co);
  float anglef = (kw > PI_D) ? (float)(kw - (2.0 * PI_D)) : kw;
  float cspt = co * spt, sspt = s * spt;
  float crspt = co / spt, srspt = s / spt;
  float bsz = fabsf(cspt) + fabsf(sspt);
  float ext = 2.5f * bsz; /* half extent of the rotated 5 x 5-cell square's bounding box */
  float xmin = border_lo(c, kx - ext);
  float ymin = border_lo(c, ky - ext);
  float xmax = border_hi(c, kx + ext, width);
  float ymax = border_hi(c, ky + ext, height);
  int e;
  (void)frexpf(0.75f * (spt + 1.0f) * (spt + 1.0f), &e);
  int sh = 32 - e;
  if (sh > 30) sh = 30;
  if (sh < 0) sh = 0;
  const float scale = ldexpf(1.0f, sh), rscale = ldexpf(1.0f, -sh);
  uint32_t bins[16][8];
  memset(bins, 0, sizeof(bins));
  for (float y = ymin; y <= ymax; y += 1.0f) {
    for (float x = xmin; x <= xmax; x += 1.0f) {
      float dx = x - kx;
      float dy = y - ky;
      float u = fmaf(crspt, dx, srspt * dy);
      float v = fmaf(crspt, dy, -(srspt * dx));
      if (!((fabsf(u) < 2.5f) && (fabsf(v) < 2.5f))) continue;
      const float* cc = got + 2 * ((long)(int)y * width + (int)x);
      float ww = om_expf(-0.125f * fmaf(u, u, v * v));
      float theta = (anglef - cc[1]) * rpi;
      if (theta < 0) theta += 8.0f;
      /* DYNAMIC_INDEXING = false drops a sample whose floor(theta) is 8 (:1763-1771); -di adds it to des[8] = des[0] */
      if (!(theta >= 0.0f && (theta < 8.0f || (theta == 8.0f && c->p.dynamic_indexing)))) continue;
      float fo = floorf(theta);
      int b0 = (int)fo & 7, b1 = (b0 + 1) & 7;
      float wb1 = theta - fo, wb0 = 1.0f - wb1;
      float au = u + 1.5f, av = v + 1.5f;
      float fu = floorf(au), fv = floorf(av);
      int ix0 = (int)fu, iy0 = (int)fv;
      float wx1 = au - fu, wx0 = 1.0f - wx1;
      float wy1 = av - fv, wy0 = 1.0f - wy1;
      float wt = (ww * cc[0]) * scale;
      for (int j = 0; j < 2; j++) {
        int iy = iy0 + j;
        if (iy < 0 || iy > 3) continue;
        float a = wt * (j ? wy1 : wy0);
        for (int i = 0; i < 2; i++) {
          int ix = ix0 + i;
          if (ix < 0 || ix > 3) continue;
          float b = a * (i ? wx1 : wx0);
          bins[iy * 4 + ix][b0] += f2u_sat(fmaf(b, wb0, 0.5f));
          bins[iy * 4 + ix][b1] += f2u_sat(fmaf(b, wb1, 0.5f));
        }
      }
    }
  }
  for (int bidx = 0; bidx < 16; bidx++) {
    float des[8];
    for (int k = 0; k < 8; k++) des[k] = (float)bins[bidx][k] * rscale;
    if (half) {
      for (int k = 0; k < 4; k++) d[bidx * 4 + k] = des[k] + des[k + 4]; /* des[k] += des[k+4], ProgramCU.cu:1782-1785 */
    } else {
      for (int k = 0; k < 8; k++) d[bidx * 8 + k] = des[k];
    }
  }
  normalize_descriptor(c, d);
}

/* HESS_ORACLE_DESC_EXACT (oracle/hess_oracle.h): the reference's formula in double precision.  Same pixels, same
 * window test, same weights as ComputeDescriptor_Kernel (ProgramCU.cu:1690-1790) cell by cell, and the normalisation
 * of NormalizeDescriptor_Kernel (:1972-2054) -- every intermediate a double; inputs are the float record fields and
 * the float (gradient, theta) plane. */
static void compute_descriptor_exact(const hess_cpu_ctx* c, const frec* rec, float angle, const float* got,
                                     int width, int height, float* d /* 128 or 64 */) {
  const int half = c->p.half_sift;
  const double kx = FIXED_TO_FLOAT(rec->x & 0x00FFFFFFu, 10);
  const double ky = FIXED_TO_FLOAT(rec->y & 0x00FFFFFFu, 10);
  const double kz = FIXED_TO_FLOAT(rec->z & 0x0000FFFFu, 8);
  const double kw = angle;
  const double spt = fabs(kz * (double)c->p.desc_window_factor);
  const double s = sin(kw), co = cos(kw);
  const float anglef_f = (angle > PI_D) ? (float)(angle - (2.0 * PI_D)) : angle, rpi_f = (float)(4.0 / PI_D);
  const double cspt = co * spt, sspt = s * spt, crspt = co / spt, srspt = s / spt;
  const double bsz = fabs(cspt) + fabs(sspt);
  double out[128];
  for (int bidx = 0; bidx < 16; bidx++) {
    const int ix = bidx & 3, iy = bidx >> 2;
    const double offx = ix - 1.5, offy = iy - 1.5;
    const double ptx = cspt * offx - sspt * offy + kx, pty = cspt * offy + sspt * offx + ky;
    const double xmin = fmax(1.5, floor(ptx - bsz) + 0.5), ymin = fmax(1.5, floor(pty - bsz) + 0.5);
    const double xmax = fmin(width - 1.5, floor(ptx + bsz) + 0.5), ymax = fmin(height - 1.5, floor(pty + bsz) + 0.5);
    double des[9];
    for (int i = 0; i < 9; ++i) des[i] = 0.0;
    for (double y = ymin; y <= ymax; y += 1.0) {
      for (double x = xmin; x <= xmax; x += 1.0) {
        const double dx = x - ptx, dy = y - pty;
        const double nx = crspt * dx + srspt * dy, ny = crspt * dy - srspt * dx;
        if (!(fabs(nx) < 1.0 && fabs(ny) < 1.0)) continue;
        const float* cc = got + 2 * ((long)(int)y * width + (int)x);
        const double dnx = nx + offx, dny = ny + offy;
        const double weight = exp(-0.125 * (dnx * dnx + dny * dny)) * (1.0 - fabs(nx)) * (1.0 - fabs(ny)) * (double)cc[0];
        /* the bin coordinate as the kernels form it, in float: which bin a sample falls into -- and whether it is the
         * dropped floor(theta) == 8 case -- is a discrete decision every float order takes the same way */
        float theta_f = (anglef_f - cc[1]) * rpi_f;
        if (theta_f < 0) theta_f += 8.0f;
        const double theta = theta_f;
        const double fo = floor(theta);
        const int fidx = (int)fo;
        if (fidx >= 0 && fidx < 8) {
          des[fidx] += (fo + 1.0 - theta) * weight;
          des[fidx + 1] += (theta - fo) * weight;
        } else if (fidx == 8 && c->p.dynamic_indexing) {
          des[8] += (fo + 1.0 - theta) * weight;
        }
      }
    }
    des[0] += des[8];
    if (half) for (int k = 0; k < 4; k++) out[bidx * 4 + k] = des[k] + des[k + 4];
    else for (int k = 0; k < 8; k++) out[bidx * 8 + k] = des[k];
  }
  const int n = half ? 64 : 128;
  if (c->p.normalize) {
    for (int pass = 0; pass < 2; pass++) {
      double sum = 0.0;
      for (int j = 0; j < n; j++) sum += out[j] * out[j];
      const double nrm = 1.0 / sqrt(sum);
      for (int j = 0; j < n; j++) out[j] = pass == 0 ? fmin(0.2, out[j] * nrm) : out[j] * nrm;
    }
  }
  for (int j = 0; j < n; j++) d[j] = (float)out[j];
}

static void compute_descriptor(const hess_cpu_ctx* c, int order, const frec* rec, float angle, const float* got,
                               int width, int height, float* d /* 128 or 64 */) {
  if (order == HESS_DESC_ORDER_PIXEL) { compute_descriptor_pixel(c, rec, angle, got, width, height, d); return; }
  if (order == HESS_ORACLE_DESC_EXACT) { compute_descriptor_exact(c, rec, angle, got, width, height, d); return; }
  const float rpi = (float)(4.0 / PI_D);
  int half = c->p.half_sift;
  float kx = FIXED_TO_FLOAT(rec->x & 0x00FFFFFFu, 10);
  float ky = FIXED_TO_FLOAT(rec->y & 0x00FFFFFFu, 10);
  float kz = FIXED_TO_FLOAT(rec->z & 0x0000FFFFu, 8);
  float kw = angle;
  float spt = fabsf(kz * c->p.desc_window_factor);
  float s, co;
  om_sincosf(kw, &s, &co);
  float anglef = (kw > PI_D) ? (float)(kw - (2.0 * PI_D)) : kw;
  float cspt = co * spt, sspt = s * spt;
  float crspt = co / spt, srspt = s / spt;
  for (int bidx = 0; bidx < 16; bidx++) {
    int ix = bidx & 3, iy = bidx >> 2;
    float offx = ix - 1.5f, offy = iy - 1.5f;
    float ptx = fmaf(cspt, offx, -(sspt * offy)) + kx;
    float pty = fmaf(cspt, offy, sspt * offx) + ky;
    float bsz = fabsf(cspt) + fabsf(sspt);
    float xmin = border_lo(c, ptx - bsz);
    float ymin = border_lo(c, pty - bsz);
    float xmax = border_hi(c, ptx + bsz, width);
    float ymax = border_hi(c, pty + bsz, height);
    float des[9];
    for (int i = 0; i < 9; ++i) des[i] = 0.0f;
    /* HESS_DESC_ORDER_INTERLEAVED (include/hess_abi.h; NOT the reference's order, which is the sequential one below):
     * the samples at positions 0, 1, 2, 3 modulo 4 of the scan over the cell's box are summed apart (part[]) and the
     * four sums added as (p0 + p1) + (p2 + p3) -- the product's default summation order, restated here so that the
     * comparison with the HIP path stays bitwise in that mode too.  tests/test_descriptor_order.py bounds the
     * difference between the two orders. */
    const int interleaved = order == HESS_DESC_ORDER_INTERLEAVED;
    float part[4][9];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 9; ++i) part[q][i] = 0.0f;
    unsigned t = 0; /* position in the scan of the box, outside-the-window samples included */
    for (float y = ymin; y <= ymax; y += 1.0f) {
      for (float x = xmin; x <= xmax; x += 1.0f, ++t) {
        float* const acc = interleaved ? part[t & 3u] : des;
        float dx = x - ptx;
        float dy = y - pty;
        float nx = fmaf(crspt, dx, srspt * dy);
        float ny = fmaf(crspt, dy, -(srspt * dx));
        float nxn = fabsf(nx), nyn = fabsf(ny);
        if ((nxn < 1.0f) && (nyn < 1.0f)) {
          const float* cc = got + 2 * ((long)(int)y * width + (int)x);
          float dnx = nx + offx;
          float dny = ny + offy;
          float ww = om_expf(-0.125f * fmaf(dnx, dnx, dny * dny));
          float wx = 1.0f - nxn;
          float wy = 1.0f - nyn;
          float weight = ww * wx * wy * cc[0];
          float theta = (anglef - cc[1]) * rpi;
          if (theta < 0) theta += 8.0f;
          float fo = floorf(theta);
          int fidx = (int)fo;
          float weight1 = fo + 1.0f - theta;
          float weight2 = theta - fo;
          if (fidx >= 0 && fidx < 8) { /* DYNAMIC_INDEXING = false: k==fidx for k<8 only (:1763-1771) */
            acc[fidx] = fmaf(weight1, weight, acc[fidx]);
            acc[fidx + 1] = fmaf(weight2, weight, acc[fidx + 1]);
          } else if (fidx == 8 && c->p.dynamic_indexing) {
            /* -di, DYNAMIC_INDEXING = true (:1755-1759): des[8] += weight1*weight; the reference also
             * writes des[9] (one past the array) += weight2*weight = +0: not restated */
            acc[8] = fmaf(weight1, weight, acc[8]);
          }
        }
      }
    }
    if (interleaved)
      for (int i = 0; i < 9; ++i) des[i] = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    des[0] += des[8];
    if (half) {
      des[0] += des[4]; des[1] += des[5]; des[2] += des[6]; des[3] += des[7];
      for (int k = 0; k < 4; k++) d[bidx * 4 + k] = des[k];
    } else {
      for (int k = 0; k < 8; k++) d[bidx * 8 + k] = des[k];
    }
  }
  normalize_descriptor(c, d);
}

static void normalize_descriptor(const hess_cpu_ctx* c, float* d) {
  if (!c->p.normalize) return;
  /* NormalizeDescriptor_Kernel: 32 lanes, lane j owns 4 (2 in half mode) consecutive floats;
   * tree reduction of ND_WarpReduction (ProgramCU.cu:1954-1969); rsqrt modelled as 1/sqrtf. */
  const int half = c->p.half_sift;
  int per = half ? 2 : 4;
  float part[32];
  for (int pass = 0; pass < 2; pass++) {
    for (int j = 0; j < 32; j++) {
      const float* t = d + j * per;
      part[j] = half ? fmaf(t[1], t[1], t[0] * t[0])
                     : fmaf(t[3], t[3], fmaf(t[2], t[2], fmaf(t[1], t[1], t[0] * t[0])));
    }
    for (int st = 16; st >= 1; st >>= 1)
      for (int j = 0; j < st; j++) part[j] += part[j + st];
    float nrm = 1.0f / sqrtf(part[0]);
    if (pass == 0) for (int j = 0; j < 32 * per; j++) d[j] = fminf(0.2f, d[j] * nrm);
    else for (int j = 0; j < 32 * per; j++) d[j] *= nrm;
  }
}

/* ------------------------------------------------------------------------------------------ */

static void free_pyramid(hess_cpu_ctx* c, pyramid* py) {
  for (int o = 0; o < MAX_OCT; o++)
    for (int l = 0; l < MAX_LEV; l++) {
      free(py->gauss[o][l]); free(py->deth[o][l]); free(py->got[o][l]);
      py->gauss[o][l] = py->deth[o][l] = py->got[o][l] = NULL;
    }
  (void)c;
}

static void free_results(hess_cpu_ctx* c) {
  if (!c->res) return;
  for (int i = 0; i < c->batch; i++) {
    free(c->res[i].keys); free(c->res[i].desc); free(c->res[i].raw);
    if (c->res[i].have_pyr) free_pyramid(c, &c->res[i].pyr);
  }
  free(c->res);
  c->res = NULL;
  c->batch = 0;
}

hess_cpu_ctx* hess_cpu_create(const hess_params* params) {
  hess_cpu_ctx* c = (hess_cpu_ctx*)calloc(1, sizeof(*c));
  if (!c) return NULL;
  if (params) c->p = *params; else hess_cpu_default_params(&c->p);
  if (c->p.abi_version != HESS_ABI_VERSION || c->p.dog_level_num < 0 || c->p.dog_level_num > 10 ||
      HESS_ORACLE_DETECTOR(&c->p) < 0 || HESS_ORACLE_DETECTOR(&c->p) > 2) {
    free(c);
    return NULL;
  }
  if (c->p.first_octave < -3) c->p.first_octave = -3; /* "can't upsample by more than 8": clamped, PyramidCU.cpp:131-132 */
  resolve_params(c);
  c->threads = 1;
  c->keep = 1;
  return c;
}

void hess_cpu_destroy(hess_cpu_ctx* c) {
  if (!c) return;
  free_results(c);
  free(c->user_keys);
  free(c->user_levels);
  free(c);
}
/* Analysis hook (tests/golden/analyze_box_fixture.py): describe user keypoint k at level index
 * levels[k] = octave * dog + (level - 1) instead of the level GenerateFeatureListTex's scale rule picks
 * (-1 keeps the rule).  A detected keypoint is described at its DETECTION level, whose sigma is within one
 * scale step of the keypoint's scale, not within half a step: the hook lets a test ask which of the two
 * admissible levels the reference used.  Applies to the following hess_cpu_set/run_keypoints calls. */
int hess_cpu_debug_key_levels(hess_cpu_ctx* c, const int* levels, int num) {
  if (!c || num < 0) return HESS_ERR_ARG;
  free(c->user_levels);
  c->user_levels = NULL;
  if (levels && num > 0) {
    c->user_levels = (int*)malloc((size_t)num * sizeof(int));
    if (!c->user_levels) return HESS_ERR_NOMEM;
    memcpy(c->user_levels, levels, (size_t)num * sizeof(int));
  }
  return 0;
}

void hess_cpu_set_threads(hess_cpu_ctx* c, int t) { c->threads = t < 1 ? 1 : t; }
void hess_cpu_keep_levels(hess_cpu_ctx* c, int on) { c->keep = on; }

/* Octave geometry: SetImageData (GLTexImage.cpp:936-1033), InitPyramid / ResizePyramid
 * (PyramidCU.cpp:113-310).  Returns 0 or a negative status. */
static int plan_geometry(hess_cpu_ctx* c, int width, int height) {
  const hess_params* p = &c->p;
  int ds = 0, ws = width, hs = height;
  if (p->first_octave > 0) { /* _PreProcessOnCPU = 1: decimate on input, GLTexImage.cpp:932-939 */
    ds = p->first_octave;
    ws = width >> ds;
    hs = height >> ds;
  } else if (p->first_octave < 0) { /* InitPyramid, PyramidCU.cpp:120-138: truncate, then up-sample */
    ds = p->first_octave;
    ws = (width & ~3) << (-ds);
    hs = height << (-ds);
  }
  if (ws > p->tex_max_dim || hs > p->tex_max_dim) {
    if (!p->auto_downscale) {
      snprintf(c->err, sizeof(c->err), "image %dx%d exceeds max dimension %d (use -ads or -maxd)", ws, hs,
               p->tex_max_dim);
      return HESS_ERR_TOO_BIG;
    }
    /* _octave_min++ until it fits, whatever its sign (PyramidCU.cpp:154-166) */
    do { ds++; ws >>= 1; hs >>= 1; } while (ws > p->tex_max_dim || hs > p->tex_max_dim);
  }
  ws &= ~3; /* TruncateWidthCU, GLTexImage.h:127 */
  if (ws < 4 || hs < 1) { snprintf(c->err, sizeof(c->err), "image too small"); return HESS_ERR_ARG; }
  c->ds = ds;
  c->img_w = ws;
  c->img_h = hs;
  int input_sz = ws < hs ? ws : hs;
  int nmax = (int)floor(log((double)input_sz) / log(2.0)) - 3; /* PyramidCU.cpp:242 */
  if (nmax < 1) nmax = 1;
  /* -no N larger than the image supports is capped (the reference would allocate 0-sized
   * octaves in ResizePyramid and caps only in FitPyramid, PyramidCU.cpp:330-335). */
  c->noct = (p->octave_num >= 1 && p->octave_num < nmax) ? p->octave_num : nmax;
  int w = ws, h = hs;
  for (int o = 0; o < c->noct; o++) {
    c->g[o].w = w;
    c->g[o].wa = ((w + 3) / 4) * 4;
    c->g[o].h = h;
    w >>= 1;
    h >>= 1;
  }
  return 0;
}

typedef struct { hess_rawkey* v; int n, cap; } rawvec;
static int raw_push(rawvec* r, const hess_rawkey* k) {
  if (r->n == r->cap) {
    int nc = r->cap ? r->cap * 2 : 1024;
    hess_rawkey* nv = (hess_rawkey*)realloc(r->v, (size_t)nc * sizeof(*nv));
    if (!nv) return -1;
    r->v = nv; r->cap = nc;
  }
  r->v[r->n++] = *k;
  return 0;
}

/* Top-K selection: SelectTopK PyramidCU.cpp:1881-1987 + TopK* ProgramCU.cu:2205-3051.  Keeps the
 * K largest abs(half->float(response)); ties resolved towards the lower list index (our rule;
 * the reference's bitonic network is not stable).  Stable compaction per level. */
typedef struct { float key; int idx; } tk;
static int tk_cmp(const void* a, const void* b) {
  const tk* x = (const tk*)a; const tk* y = (const tk*)b;
  if (x->key > y->key) return -1;
  if (x->key < y->key) return 1;
  return x->idx - y->idx;
}

/* User-supplied keypoints: PyramidCU::GenerateFeatureListTex (PyramidCU.cpp:555-718) bins the keys to
 * levels by scale (half-step bounds, first/last level catch the rest) and packs fixed-point records in
 * input order per level; orientation (strongest only) unless the keys carry one; descriptors; results
 * put back in input order through _keypoint_index (PyramidCU.cpp:537-549,1157-1168).  The pyramid
 * (gradient planes) of R must exist. */
static int user_keypoint_path(hess_cpu_ctx* c, image_result* R) {
  const hess_params* p = &c->p;
  const int dog = p->dog_level_num, num = c->user_num;
  const hess_keypoint* uk = c->user_keys;
  pyramid* py = &R->pyr;
  const double twopi = 2.0 * PI_D;
  float sigma_half_step = powf(2.0f, 0.5f / dog);
  float octave_sigma = first_octave_sigma(c);
  float offset = p->lowe_origin ? 0.0f : 0.5f;
  int cap = 2 * num + 8;
  frec* recs = (frec*)malloc((size_t)cap * sizeof(frec));
  int* rlevel = (int*)malloc((size_t)cap * sizeof(int));
  int* kindex = (int*)malloc((size_t)cap * sizeof(int));
  if (!recs || !rlevel || !kindex) return HESS_ERR_NOMEM;
  int n = 0;
  for (int octave = 0; octave < c->noct; octave++, octave_sigma *= 2.0f) {
    for (int level = 1; level <= dog; level++) {
      float level_sigma = c->level_sigma[level] * octave_sigma;
      float sigma_min = level_sigma / sigma_half_step;
      float sigma_max = level_sigma * sigma_half_step;
      for (int k = 0; k < num && n < cap; k++) {
        float sigmak = uk[k].s;
        if (c->user_levels && c->user_levels[k] >= 0) { /* analysis hook: level given, not derived */
          sigmak = (c->user_levels[k] == octave * dog + (level - 1)) ? level_sigma : -1.0f;
          if (sigmak < 0) continue;
        }
        if (((sigmak >= sigma_min) && (sigmak < sigma_max)) || ((sigmak < sigma_min) && (octave == 0) && (level == 1)) ||
            ((sigmak > sigma_max) && (octave == c->noct - 1) && (level == dog))) {
          float fX = (uk[k].x - offset) / octave_sigma + 0.5f;
          float fY = (uk[k].y - offset) / octave_sigma + 0.5f;
          float fScale = uk[k].s / octave_sigma;
          float fOrientation = (float)fmod(twopi - uk[k].o, twopi);
          recs[n].x = (uint32_t)FLOAT_TO_FIXED(fX, 10) & 0x00FFFFFFu;
          recs[n].y = (uint32_t)FLOAT_TO_FIXED(fY, 10) & 0x00FFFFFFu;
          recs[n].z = (uint32_t)FLOAT_TO_FIXED(fScale, 8) & 0x0000FFFFu;
          recs[n].w = om_f2u(fOrientation);
          rlevel[n] = octave * dog + (level - 1);
          kindex[n] = k;
          n++;
        }
      }
    }
  }
  /* orientation: SiftPyramid.cpp:128-137 (skipped when the keys have one; _MaxOrientation > 0 always) */
  if (!c->user_have_orientation) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
      int o = rlevel[i] / dog, l = rlevel[i] % dog + 1;
      compute_orientation_existing(c, &recs[i], py->got[o][l], c->g[o].wa, c->g[o].h);
    }
  }
  /* keypoints: the caller's records unless DownloadKeypoints runs (SiftPyramid.cpp:160-171) */
  R->n = num;
  R->keys = (hess_keypoint*)malloc((size_t)(num ? num : 1) * sizeof(hess_keypoint));
  memcpy(R->keys, uk, (size_t)num * sizeof(hess_keypoint));
  int download = !c->user_have_orientation && ((p->max_orientation < 2) || p->fixed_orientation);
  int listed = n < num ? n : num;
  if (download) {
    float os = first_octave_sigma(c);
    for (int i = 0; i < listed; i++) {
      int li = rlevel[i];
      float oss = os * (float)(1 << (li / dog));
      hess_keypoint* d = &R->keys[kindex[i]];
      float posX = FIXED_TO_FLOAT(recs[i].x & 0x00FFFFFFu, 10);
      float posY = FIXED_TO_FLOAT(recs[i].y & 0x00FFFFFFu, 10);
      float scale = FIXED_TO_FLOAT(recs[i].z & 0x0000FFFFu, 8);
      d->x = oss * (posX - 0.5f) + offset;
      d->y = oss * (posY - 0.5f) + offset;
      d->s = oss * scale;
      d->o = (float)fmod(twopi - om_u2f(recs[i].w), twopi);
      d->response = om_h2f(0);
      d->level = (uint16_t)li;
      d->type = 0;
    }
  }
  int dim = p->compute_descriptors ? (p->half_sift ? 64 : 128) : 0;
  c->desc_dim = dim;
  R->desc = NULL;
  if (dim) {
    R->desc = (float*)calloc((size_t)(num ? num : 1) * dim, sizeof(float));
    float* tmp = (float*)malloc((size_t)(n ? n : 1) * dim * sizeof(float));
    if (!R->desc || !tmp) return HESS_ERR_NOMEM;
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; i++) {
      int o = rlevel[i] / dog, l = rlevel[i] % dog + 1;
      /* (a keypoint list is described in a floating-point order: the pixel order's fixed-point scale assumes the
       * contrast of a detected keypoint) */
      compute_descriptor(c, p->descriptor_order == HESS_DESC_ORDER_PIXEL ? HESS_DESC_ORDER_INTERLEAVED : p->descriptor_order,
                         &recs[i], om_u2f(recs[i].w), py->got[o][l], c->g[o].wa, c->g[o].h, tmp + (size_t)i * dim);
    }
    for (int i = 0; i < listed; i++) memcpy(R->desc + (size_t)kindex[i] * dim, tmp + (size_t)i * dim, (size_t)dim * 4);
    free(tmp);
  }
  R->nraw = 0;
  free(R->raw);
  R->raw = (hess_rawkey*)malloc(sizeof(hess_rawkey));
  free(recs); free(rlevel); free(kindex);
  return 0;
}

static int process_image(hess_cpu_ctx* c, const unsigned char* pix, int width, int height, int pitch,
                         int format, int pixtype, image_result* R) {
  const hess_params* p = &c->p;
  int nch = fmt_channels(format);
  int bpc = pixtype == HESS_PIX_U8 ? 1 : (pixtype == HESS_PIX_U16 ? 2 : 4);
  int dog = p->dog_level_num;
  pyramid* py = &R->pyr;
  memset(py, 0, sizeof(*py));
  R->have_pyr = 1;
  double t0 = now_ms(), t1;

  /* --- input: decimate by 2^ds, convert, drop W mod 4 columns (GLTexImage.cpp:993-1011) --- */
  const int up = c->ds < 0 ? -c->ds : 0; /* up-sampled first octave: convert at full size, SampleImageU below */
  int W = c->img_w >> up, H = c->img_h >> up, step = 1 << (c->ds > 0 ? c->ds : 0);
  float* input = (float*)malloc((size_t)W * H * sizeof(float));
  if (!input) return HESS_ERR_NOMEM;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < H; r++)
    for (int x = 0; x < W; x++)
      input[(size_t)r * W + x] =
          convert_pixel(pix + (size_t)(r * step) * pitch + (size_t)(x * step) * nch * bpc, format, pixtype);
  if (up) { /* PyramidCU.cpp:1521-1522 */
    float* big = (float*)malloc((size_t)c->img_w * c->img_h * sizeof(float));
    if (!big) { free(input); return HESS_ERR_NOMEM; }
    upsample_image(input, W, H, up, big);
    free(input);
    input = big;
  }
  (void)width; (void)height;
  t1 = now_ms(); c->timing[HESS_T_LOAD] += (float)(t1 - t0); t0 = t1;

  /* --- allocation --- */
  for (int o = 0; o < c->noct; o++) {
    size_t n = (size_t)c->g[o].wa * c->g[o].h;
    for (int l = 0; l < c->level_num; l++) {
      py->gauss[o][l] = (float*)malloc(n * sizeof(float));
      py->deth[o][l] = (float*)malloc(n * sizeof(float));
      if (l >= 1 && l <= dog) py->got[o][l] = (float*)malloc(2 * n * sizeof(float));
      if (!py->gauss[o][l] || !py->deth[o][l] || (l >= 1 && l <= dog && !py->got[o][l])) {
        free(input);
        return HESS_ERR_NOMEM;
      }
    }
  }
  float* buf = (float*)malloc((size_t)c->g[0].wa * c->g[0].h * sizeof(float));
  if (!buf) { free(input); return HESS_ERR_NOMEM; }
  t1 = now_ms(); c->timing[HESS_T_ALLOC] += (float)(t1 - t0); t0 = t1;

  /* --- BuildPyramid, PyramidCU.cpp:1486-1558 --- */
  float taps[KMAXW];
  for (int o = 0; o < c->noct; o++) {
    int wa = c->g[o].wa, h = c->g[o].h;
    if (o == 0) {
      float s0 = initial_smooth_sigma(c, c->ds); /* _octave_min(0) + _down_sample_factor */
      if (s0 > 0.0f) {
        int fw = create_filter_kernel(c, s0, taps);
        filter_image(c, py->gauss[0][0], input, buf, wa, h, taps, fw);
      } else {
        /* ProgramCU::FilterImage with sigma 0: width clamps to 5 taps of exp(-inf)... the reference
         * would divide by zero; sigma0 > sigman always holds for valid parameters. */
        memcpy(py->gauss[0][0], input, (size_t)wa * h * sizeof(float));
      }
    } else { /* SampleImageD from level_ds of the previous octave, ProgramCU.cu:312-326 */
      const float* src = py->gauss[o - 1][c->level_ds];
      int sw = c->g[o - 1].wa;
      float* dst = py->gauss[o][0];
#pragma omp parallel for schedule(static)
      for (int r = 0; r < h; r++)
        for (int x = 0; x < wa; x++) {
          int sc = (x << 1) < (sw - 1) ? (x << 1) : (sw - 1);
          dst[(size_t)r * wa + x] = src[(size_t)(r << 1) * sw + sc];
        }
    }
    for (int l = 1; l <= c->level_max; l++) {
      int fw = create_filter_kernel(c, c->sigma[l - 1], taps);
      filter_image(c, py->gauss[o][l], py->gauss[o][l - 1], buf, wa, h, taps, fw);
    }
  }
  free(buf);
  free(input);
  t1 = now_ms(); c->timing[HESS_T_PYRAMID] += (float)(t1 - t0); t0 = t1;

  /* --- DetectKeypointsEX, PyramidCU.cpp:1560-1699 --- */
  for (int o = 0; o < c->noct; o++)
    for (int l = 0; l <= c->level_max; l++) {
      float ls = c->level_sigma[l] * 1.0f; /* octaveSigma = 1, PyramidCU.cpp:1574-1585 */
      compute_hessian(py->gauss[o][l], py->deth[o][l], py->got[o][l], c->g[o].wa, c->g[o].h, ls * ls);
      if (HESS_ORACLE_DETECTOR(p) != 0 && l >= 1) { /* ComputeDOG_Kernel, ProgramCU.cu:598-637: the plane the extrema are sought in */
        const float* a = py->gauss[o][l];
        const float* b = py->gauss[o][l - 1];
        float* d = py->deth[o][l];
        size_t n = (size_t)c->g[o].wa * c->g[o].h;
        for (size_t i = 0; i < n; i++) d[i] = a[i] - b[i];
      }
    }
  if (c->user_num > 0) { /* SIFT_SKIP_DETECTION: ComputeGradient + GenerateFeatureListTex */
    int rc = user_keypoint_path(c, R);
    if (!c->keep) { free_pyramid(c, py); R->have_pyr = 0; }
    return rc;
  }
  float Tdog = p->dog_threshold;
  float Tdog1 = (p->subpixel ? 0.8f : 1.0f) * Tdog;                         /* ProgramCU.cu:897 */
  float Tedge = (p->edge_threshold + 1) * (p->edge_threshold + 1) / p->edge_threshold; /* :913 */
  int nlev = c->noct * dog;
  rawvec* lists = (rawvec*)calloc((size_t)nlev, sizeof(rawvec));
  int* level_num = (int*)calloc((size_t)nlev + 1, sizeof(int));
  if (!lists || !level_num) return HESS_ERR_NOMEM;
  /* Detection runs for every level (ComputeKEY is launched for all of them); list GENERATION is
   * what -tc2/-tc3 skip (PyramidCU.cpp:1283-1344). */
#pragma omp parallel for schedule(dynamic, 1)
  for (int li = 0; li < nlev; li++) {
    int o = li / dog, l = li % dog + 1;
    int wa = c->g[o].wa, h = c->g[o].h;
    keyval kv;
    /* without GPU_HESSIAN the list `level` (0..dog-1) is sought in DoG plane level + 2 of that build = l + 1 here
     * (between Gaussian levels l and l + 1) and described from Gaussian level l (PyramidCU.cpp:1655-1670, 1825-1846) */
    const int dm = HESS_ORACLE_DETECTOR(p) != 0, pl = dm ? l + 1 : l;
    for (int row = 1; row < h - 1; row++)
      for (int col = 1; col < wa - 1; col++)
        if (compute_key(py->deth[o][pl], py->deth[o][pl - 1], py->deth[o][pl + 1], py->gauss[o][l], wa, row, col,
                        Tdog1, Tdog, Tedge, p->subpixel, dm, &kv)) {
          hess_rawkey rk;
          rk.level_index = li; rk.col = col; rk.row = row; rk.packed = kv.packed;
          rk.dx = kv.dx; rk.dy = kv.dy; rk.ds = kv.ds; rk.pad = 0;
          raw_push(&lists[li], &rk);
        }
  }
  t1 = now_ms(); c->timing[HESS_T_DETECT] += (float)(t1 - t0); t0 = t1;

  /* --- GenerateFeatureList, PyramidCU.cpp:1283-1368: -tc2 (method 1) walks octaves and levels in
   * reverse, and methods 1 and 2 stop adding levels once the count exceeds the threshold. --- */
  int thr = p->feature_count_threshold;
  int feature_num = 0;
  {
    int reverse = (p->truncate_method == HESS_TRUNC_HIGHEST_1);
    for (int k = 0; k < nlev; k++) {
      int li = reverse ? nlev - 1 - k : k;
      if ((p->truncate_method == HESS_TRUNC_HIGHEST_1 || p->truncate_method == HESS_TRUNC_LOWEST) && thr > 0 &&
          feature_num > thr) {
        lists[li].n = 0;
        continue;
      }
      feature_num += lists[li].n;
    }
  }
  for (int li = 0; li < nlev; li++) level_num[li] = lists[li].n;
  t1 = now_ms(); c->timing[HESS_T_LIST] += (float)(t1 - t0); t0 = t1;

  /* --- LimitFeatureCount(0), SiftPyramid.cpp:201-278 --- */
  if (thr > 0) {
    if (p->truncate_method == HESS_TRUNC_TOPK) {
      if (feature_num >= thr) { /* SelectTopK, PyramidCU.cpp:1886-1887 */
        tk* a = (tk*)malloc((size_t)feature_num * sizeof(tk));
        char* keepf = (char*)calloc((size_t)feature_num, 1);
        if (!a || !keepf) return HESS_ERR_NOMEM;
        int n = 0;
        for (int li = 0; li < nlev; li++)
          for (int j = 0; j < lists[li].n; j++, n++) {
            a[n].key = fabsf(om_h2f((uint16_t)(lists[li].v[j].packed >> 16))); /* ProgramCU.cu:2266-2273 */
            a[n].idx = n;
          }
        qsort(a, (size_t)n, sizeof(tk), tk_cmp);
        for (int t = 0; t < thr; t++) keepf[a[t].idx] = 1;
        n = 0;
        feature_num = 0;
        for (int li = 0; li < nlev; li++) {
          int m = 0;
          for (int j = 0; j < lists[li].n; j++, n++)
            if (keepf[n]) lists[li].v[m++] = lists[li].v[j];
          lists[li].n = level_num[li] = m;
          feature_num += m;
        }
        free(a);
        free(keepf);
      }
    } else if (p->truncate_method == HESS_TRUNC_LOWEST) {
      int i = 0, nf = 0;
      for (; (nf < thr) && (i < nlev); ++i) nf += level_num[i];
      for (; i < nlev; ++i) { level_num[i] = 0; lists[i].n = 0; }
      if (nf < feature_num) feature_num = nf;
    } else {
      int i = 0;
      while (i < nlev && (feature_num - level_num[i]) > thr) {
        feature_num -= level_num[i];
        lists[i].n = 0;
        level_num[i++] = 0;
      }
    }
  }
  t1 = now_ms(); c->timing[HESS_T_REDUCTION] += (float)(t1 - t0); t0 = t1;

  /* raw list for parity dumps (after reduction) */
  R->nraw = feature_num;
  R->raw = (hess_rawkey*)malloc((size_t)(feature_num ? feature_num : 1) * sizeof(hess_rawkey));
  {
    int n = 0;
    for (int li = 0; li < nlev; li++)
      for (int j = 0; j < lists[li].n; j++) R->raw[n++] = lists[li].v[j];
  }

  /* --- GetFeatureOrientations, PyramidCU.cpp:1815-1857 --- */
  frec* recs = (frec*)malloc((size_t)(feature_num ? feature_num : 1) * sizeof(frec));
  if (!recs) return HESS_ERR_NOMEM;
#pragma omp parallel for schedule(dynamic, 16)
  for (int n = 0; n < feature_num; n++) {
    const hess_rawkey* rk = &R->raw[n];
    int o = rk->level_index / dog, l = rk->level_index % dog + 1;
    compute_orientation(c, rk, py->got[o][l], c->g[o].wa, c->g[o].h, c->level_sigma[l], &recs[n]);
  }
  t1 = now_ms(); c->timing[HESS_T_ORIENT] += (float)(t1 - t0); t0 = t1;

  /* --- ReshapeFeatureListCPU (PyramidCU.cpp:720-924) or DownloadKeypoints (:1029-1169) --- */
  int multi = (p->max_orientation > 1) && !p->fixed_orientation; /* SiftPyramid.cpp:140 */
  int total = 0;
  for (int n = 0; n < feature_num; n++) total += multi ? (int)((recs[n].z >> 27) & 7u) : 1;
  frec* frecs = (frec*)malloc((size_t)(total ? total : 1) * sizeof(frec));
  float* angles = (float*)malloc((size_t)(total ? total : 1) * sizeof(float));
  int* flevel = (int*)malloc((size_t)(total ? total : 1) * sizeof(int));
  hess_keypoint* keys = (hess_keypoint*)malloc((size_t)(total ? total : 1) * sizeof(hess_keypoint));
  if (!frecs || !angles || !flevel || !keys) return HESS_ERR_NOMEM;
  {
    const double twopi = 2.0 * PI_D;
    const double factor = HESS_ORACLE_DETECTOR(p) != 0 ? 2.0 * PI_D / 65535.0 : 2.0 * PI_D / 255.0; /* PyramidCU.cpp:763-767 */
    float octave_sigma = first_octave_sigma(c); /* 2^_octave_min */
    float offset = p->lowe_origin ? 0.0f : 0.5f;
    int m = 0;
    for (int n = 0; n < feature_num; n++) {
      int li = R->raw[n].level_index;
      int cnt = multi ? (int)((recs[n].z >> 27) & 7u) : 1;
      for (int k = 0; k < cnt; k++, m++) {
        frecs[m] = recs[n];
        flevel[m] = li;
        angles[m] = !multi ? om_u2f(recs[n].w)
                    : (HESS_ORACLE_DETECTOR(p) != 0 ? (float)(factor * ((recs[n].w >> (16 * k)) & 0xFFFFu))
                                        : (float)(factor * ((recs[n].w >> (8 * k)) & 0xFFu)));
        float oss = octave_sigma * (float)(1 << (li / dog));
        float posX = FIXED_TO_FLOAT(recs[n].x & 0x00FFFFFFu, 10);
        float posY = FIXED_TO_FLOAT(recs[n].y & 0x00FFFFFFu, 10);
        uint16_t hr = (uint16_t)(((recs[n].x & 0xFF000000u) >> 16) | ((recs[n].y & 0xFF000000u) >> 24));
        float scale = FIXED_TO_FLOAT(recs[n].z & 0x0000FFFFu, 8);
        keys[m].x = oss * (posX - 0.5f) + offset;
        keys[m].y = oss * (posY - 0.5f) + offset;
        keys[m].s = oss * scale;
        keys[m].o = (float)fmod(twopi - angles[m], twopi);
        keys[m].response = om_h2f(hr);
        keys[m].level = (uint16_t)li;
        keys[m].type = (uint16_t)((recs[n].z & 0xC0000000u) >> 30);
      }
    }
  }
  free(recs);
  /* LimitFeatureCount(1), SiftPyramid.cpp:143: no-op for top-K; for -tc* it re-applies the level
   * truncation on the multi-orientation counts. */
  int first = 0;
  if (multi && thr > 0 && p->truncate_method != HESS_TRUNC_TOPK) {
    int* cnt = (int*)calloc((size_t)nlev, sizeof(int));
    for (int m = 0; m < total; m++) cnt[flevel[m]]++;
    if (p->truncate_method == HESS_TRUNC_LOWEST) {
      int i = 0, nf = 0;
      for (; (nf < thr) && (i < nlev); ++i) nf += cnt[i];
      if (nf < total) total = nf; /* keypoint buffer is not modified, only the count */
    } else {
      int i = 0, tot = total;
      while (i < nlev && (tot - cnt[i]) > thr) { tot -= cnt[i]; first += cnt[i]; i++; }
      total = tot;
    }
    free(cnt);
  }
  t1 = now_ms(); c->timing[HESS_T_MULTI_ORIENT] += (float)(t1 - t0); t0 = t1;

  /* --- GetFeatureDescriptors, PyramidCU.cpp:491-553 --- */
  int dim = p->compute_descriptors ? (p->half_sift ? 64 : 128) : 0;
  c->desc_dim = dim;
  /* the pixel order's fixed-point bound assumes luminance in [0, 1] (8- and 16-bit inputs): float pixels are taken as
   * they are and keep the interleaved order (hess_schedule.hip: the same rule) */
  const int desc_order = (p->descriptor_order == HESS_DESC_ORDER_PIXEL && pixtype == HESS_PIX_F32) ? HESS_DESC_ORDER_INTERLEAVED
                                                                                                   : p->descriptor_order;
  float* desc = NULL;
  if (dim) {
    desc = (float*)malloc((size_t)(total ? total : 1) * dim * sizeof(float));
    if (!desc) return HESS_ERR_NOMEM;
#pragma omp parallel for schedule(dynamic, 8)
    for (int m = 0; m < total; m++) {
      int k = first + m;
      int o = flevel[k] / dog, l = flevel[k] % dog + 1;
      compute_descriptor(c, desc_order, &frecs[k], angles[k], py->got[o][l], c->g[o].wa, c->g[o].h, desc + (size_t)m * dim);
    }
  }
  t1 = now_ms(); c->timing[HESS_T_DESCRIPTOR] += (float)(t1 - t0); t0 = t1;

  R->n = total;
  R->keys = (hess_keypoint*)malloc((size_t)(total ? total : 1) * sizeof(hess_keypoint));
  memcpy(R->keys, keys + first, (size_t)total * sizeof(hess_keypoint));
  R->desc = desc;
  free(keys); free(frecs); free(angles); free(flevel);
  for (int li = 0; li < nlev; li++) free(lists[li].v);
  free(lists); free(level_num);
  if (!c->keep) { free_pyramid(c, py); R->have_pyr = 0; }
  return 0;
}

int hess_cpu_run_host(hess_cpu_ctx* c, const void* pixels, int width, int height, int pitch, size_t image_stride,
                      int batch, int format, int pixtype) {
  if (!c) return HESS_ERR_ARG;
  if (!pixels || width <= 0 || height <= 0 || batch <= 0 || !fmt_channels(format) || pixtype < HESS_PIX_U8 ||
      pixtype > HESS_PIX_F32) {
    snprintf(c->err, sizeof(c->err), "bad argument");
    return HESS_ERR_ARG;
  }
  free_results(c);
  memset(c->timing, 0, sizeof(c->timing));
  int rc = plan_geometry(c, width, height);
  if (rc) return rc;
#ifdef _OPENMP
  omp_set_num_threads(c->threads);
#endif
  c->res = (image_result*)calloc((size_t)batch, sizeof(image_result));
  if (!c->res) return HESS_ERR_NOMEM;
  c->batch = batch;
  double t0 = now_ms();
  for (int b = 0; b < batch; b++) {
    rc = process_image(c, (const unsigned char*)pixels + (size_t)b * image_stride, width, height, pitch, format,
                       pixtype, &c->res[b]);
    if (rc) { snprintf(c->err, sizeof(c->err), "image %d failed (%d)", b, rc); return rc; }
  }
  c->timing[HESS_T_TOTAL] = (float)(now_ms() - t0);
  if (c->user_num > 0) hess_cpu_debug_key_levels(c, NULL, 0); /* the hook covers one keypoint-list run */
  hess_cpu_set_keypoints(c, NULL, 0, 0); /* _existing_keypoints = 0 after RunSIFT, SiftPyramid.cpp:182-184 */
  return 0;
}

int hess_cpu_set_keypoints(hess_cpu_ctx* c, const hess_keypoint* keys, int num, int keys_have_orientation) {
  if (!c || num < 0 || (num > 0 && !keys)) return HESS_ERR_ARG;
  free(c->user_keys);
  c->user_keys = NULL;
  c->user_num = 0;
  if (num > 0) {
    c->user_keys = (hess_keypoint*)malloc((size_t)num * sizeof(hess_keypoint));
    if (!c->user_keys) return HESS_ERR_NOMEM;
    memcpy(c->user_keys, keys, (size_t)num * sizeof(hess_keypoint));
    c->user_num = num;
    c->user_have_orientation = keys_have_orientation != 0;
  }
  return 0;
}

/* SiftGPU::RunSIFT(num, keys, flag) (SiftGPU.cpp:307-315): same image, pyramid not rebuilt. */
int hess_cpu_run_keypoints(hess_cpu_ctx* c, const hess_keypoint* keys, int num, int keys_have_orientation) {
  if (!c || num <= 0 || !keys) return HESS_ERR_ARG;
  if (!c->res || c->batch < 1 || !c->res[0].have_pyr) { snprintf(c->err, sizeof(c->err), "no current image"); return HESS_ERR_STATE; }
  int rc = hess_cpu_set_keypoints(c, keys, num, keys_have_orientation);
  if (rc) return rc;
  image_result* R = &c->res[0];
  free(R->keys); free(R->desc);
  R->keys = NULL; R->desc = NULL;
  rc = user_keypoint_path(c, R);
  hess_cpu_debug_key_levels(c, NULL, 0); /* the hook covers one keypoint-list run */
  hess_cpu_set_keypoints(c, NULL, 0, 0);
  return rc;
}

int hess_cpu_count(hess_cpu_ctx* c, int img) {
  if (!c || !c->res || img < 0 || img >= c->batch) return HESS_ERR_ARG;
  return c->res[img].n;
}
int hess_cpu_desc_dim(hess_cpu_ctx* c) { return c ? c->desc_dim : HESS_ERR_ARG; }

int hess_cpu_fetch(hess_cpu_ctx* c, int img, hess_keypoint* keys, float* desc) {
  if (!c || !c->res || img < 0 || img >= c->batch) return HESS_ERR_ARG;
  image_result* R = &c->res[img];
  if (keys) memcpy(keys, R->keys, (size_t)R->n * sizeof(hess_keypoint));
  if (desc && R->desc) memcpy(desc, R->desc, (size_t)R->n * c->desc_dim * sizeof(float));
  return 0;
}

int hess_cpu_geometry(hess_cpu_ctx* c, int* widths, int* heights) {
  if (!c) return HESS_ERR_ARG;
  for (int o = 0; o < c->noct; o++) { if (widths) widths[o] = c->g[o].wa; if (heights) heights[o] = c->g[o].h; }
  return c->noct;
}

int hess_cpu_debug_level(hess_cpu_ctx* c, int img, int octave, int level, int what, float* out) {
  if (!c || !c->res || img < 0 || img >= c->batch || octave < 0 || octave >= c->noct || level < 0 ||
      level > c->level_max || !out)
    return HESS_ERR_ARG;
  image_result* R = &c->res[img];
  if (!R->have_pyr) return HESS_ERR_STATE;
  size_t n = (size_t)c->g[octave].wa * c->g[octave].h;
  const float* src = NULL;
  if (what == HESS_DBG_GAUSS) src = R->pyr.gauss[octave][level];
  else if (what == HESS_DBG_DETH) src = R->pyr.deth[octave][level];
  else if (what == HESS_DBG_GOT) { src = R->pyr.got[octave][level]; n *= 2; }
  if (!src) return HESS_ERR_ARG;
  memcpy(out, src, n * sizeof(float));
  return 0;
}

int hess_cpu_debug_list(hess_cpu_ctx* c, int img, hess_rawkey* out, int cap) {
  if (!c || !c->res || img < 0 || img >= c->batch) return HESS_ERR_ARG;
  image_result* R = &c->res[img];
  int n = R->nraw < cap ? R->nraw : cap;
  if (out && n > 0) memcpy(out, R->raw, (size_t)n * sizeof(hess_rawkey));
  return R->nraw;
}

const float* hess_cpu_timing(hess_cpu_ctx* c) { return c ? c->timing : NULL; }
const char* hess_cpu_last_error(hess_cpu_ctx* c) { return c ? c->err : "null context"; }

float hess_cpu_expf(float x) { return om_expf(x); }
float hess_cpu_atan2f(float y, float x) { return om_atan2f(y, x); }
void hess_cpu_sincosf(float a, float* s, float* c) { om_sincosf(a, s, c); }
unsigned short hess_cpu_f2h(float f) { return om_f2h(f); }
float hess_cpu_h2f(unsigned short h) { return om_h2f(h); }

/* ========================================================================================== */
/* Descriptor matcher: restatement of MultiplyDescriptor(_G)_Kernel,
 * RowMatch_Kernel, ColMatch_Kernel (ProgramCU.cu:3455-3843) and SiftMatchCU::GetBestMatch
 * (SiftMatchCU.cpp:148-173).  Descriptors are u8 (512*d rounded, SiftMatchCU.cpp:94-99). */

void hess_cpu_match_quantize(const float* desc, int count, unsigned char* out) {
  for (int i = 0; i < count; ++i) out[i] = (unsigned char)(int)(512 * desc[i] + 0.5); /* SiftMatchCU.cpp:97 */
}

int hess_cpu_match(const unsigned char* des1, int num1, const unsigned char* des2, int num2, const float* loc1,
                   const float* loc2, const float* H, const float* F, float distmax, float ratiomax,
                   float hdistmax, float fdistmax, int mutual_best, int max_match, int* pairs) {
  if (!des1 || !des2 || num1 <= 0 || num2 <= 0 || !pairs) return 0;
  const int guided = (loc1 && loc2 && H && F);
  int* dot = (int*)malloc((size_t)num1 * num2 * sizeof(int));   /* d_result */
  int* raw = (int*)malloc((size_t)num1 * num2 * sizeof(int));   /* `results` before the clamp */
  int* rowm = (int*)malloc((size_t)num1 * sizeof(int));
  int* colm = (int*)malloc((size_t)num2 * sizeof(int));
  if (!dot || !raw || !rowm || !colm) return 0;
#pragma omp parallel for schedule(static)
  for (int blk = 0; blk < (num1 + 7) / 8; blk++) { /* MULT_BLOCK_DIMY = 8 rows per block */
    for (int j = 0; j < num2; j++) {
      int res[8], good = 0, rows = 0;
      for (int k = 0; k < 8; k++) {
        int i = blk * 8 + k;
        res[k] = 0;
        if (guided) { /* ProgramCU.cu:3597-3635 */
          if (i < num1) {
            float locx = loc1[2 * i], locy = loc1[2 * i + 1], l2x = loc2[2 * j], l2y = loc2[2 * j + 1];
            float x0 = fmaf(H[0], locx, H[1] * locy) + H[2];
            float x1 = fmaf(H[3], locx, H[4] * locy) + H[5];
            float x2 = fmaf(H[6], locx, H[7] * locy) + H[8];
            float d0 = fabsf(x0 / x2 - l2x), d1 = fabsf(x1 / x2 - l2y);
            if (d0 < hdistmax && d1 < hdistmax) {
              float fx0 = fmaf(F[0], locx, F[1] * locy) + F[2];
              float fx1 = fmaf(F[3], locx, F[4] * locy) + F[5];
              float fx2 = fmaf(F[6], locx, F[7] * locy) + F[8];
              float ft0 = fmaf(F[0], l2x, F[3] * l2y) + F[6];
              float ft1 = fmaf(F[1], l2x, F[4] * l2y) + F[7];
              float x2fx1 = fmaf(l2x, fx0, l2y * fx1) + fx2;
              float se = (x2fx1 * x2fx1) / fmaf(ft1, ft1, fmaf(ft0, ft0, fmaf(fx0, fx0, fx1 * fx1)));
              res[k] = se < fdistmax ? 0 : -262144;
            } else res[k] = -262144;
          } else res[k] = -262144;
          good += (res[k] >= 0);
        }
        if (i < num1) rows++;
      }
      if (!guided || good > 0) {
        for (int k = 0; k < 8; k++) {
          int i = blk * 8 + k;
          if (i >= num1) continue; /* rows past the end read unspecified texels in the reference; never stored */
          const unsigned char *p1 = des1 + (size_t)i * 128, *p2 = des2 + (size_t)j * 128;
          int acc = 0;
          for (int t = 0; t < 128; t++) acc += (int)p1[t] * (int)p2[t];
          res[k] += acc;
        }
      }
      for (int k = 0; k < rows; k++) {
        int i = blk * 8 + k;
        raw[(size_t)i * num2 + j] = res[k];
        dot[(size_t)i * num2 + j] = guided ? (res[k] > 0 ? res[k] : 0) : res[k]; /* :3684 max(results,0) */
      }
    }
  }
  /* RowMatch_Kernel, ProgramCU.cu:3737-3791: 32 threads stride the row, strict '>' keeps the first
   * maximum per thread, the tree keeps the lower thread on ties; second best = second largest value. */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < num1; i++) {
    int tmax[32], tnxt[32], tidx[32];
    for (int t = 0; t < 32; t++) { tmax[t] = 0; tnxt[t] = 0; tidx[t] = -1; }
    for (int j = 0; j < num2; j++) {
      int t = j & 31, v = dot[(size_t)i * num2 + j];
      int test = v > tmax[t];
      tnxt[t] = test ? tmax[t] : (tnxt[t] > v ? tnxt[t] : v);
      tidx[t] = test ? j : tidx[t];
      tmax[t] = test ? v : tmax[t];
    }
    for (int step = 16; step > 0; step /= 2)
      for (int t = 0; t < step; t++) {
        int v1 = tmax[t], v2 = tmax[t + step];
        int test = v2 > v1;
        tnxt[t] = test ? (v1 > tnxt[t + step] ? v1 : tnxt[t + step]) : (tnxt[t] > v2 ? tnxt[t] : v2);
        tidx[t] = test ? tidx[t + step] : tidx[t];
        tmax[t] = test ? v2 : v1;
      }
    float dist = (float)acos(fmin((double)(tmax[0] * 0.000003814697265625f), 1.0));
    float distn = (float)acos(fmin((double)(tnxt[0] * 0.000003814697265625f), 1.0));
    rowm[i] = (dist < distmax) && (dist < distn * ratiomax) ? tidx[0] : -1;
  }
  if (mutual_best) { /* per 8-row block (max, index, second) on the unclamped results, then ColMatch_Kernel */
#pragma omp parallel for schedule(static)
    for (int j = 0; j < num2; j++) {
      int rx = 0, ry = -1, rz = 0, first = 1;
      for (int blk = 0; blk < (num1 + 7) / 8; blk++) {
        int cx = 0, cy = -1, cz = 0;
        for (int k = 0; k < 8 && blk * 8 + k < num1; k++) {
          int v = raw[(size_t)(blk * 8 + k) * num2 + j];
          if (v > cx) { cz = cx; cx = v; cy = blk * 8 + k; }
          else cz = cz > v ? cz : v;
        }
        if (first) { rx = cx; ry = cy; rz = cz; first = 0; }
        else if (rx < cx) { rz = rx > cz ? rx : cz; rx = cx; ry = cy; }
        else rz = rz > cx ? rz : cx;
      }
      float dist = (float)acos(fmin((double)(rx * 0.000003814697265625f), 1.0));
      float distn = (float)acos(fmin((double)(rz * 0.000003814697265625f), 1.0));
      colm[j] = (dist < distmax) && (dist < distn * ratiomax) ? ry : -1;
    }
  }
  int nmatch = 0; /* SiftMatchCU.cpp:161-171 */
  for (int i = 0; i < num1 && nmatch < max_match; ++i) {
    int j = rowm[i];
    if (j >= 0 && (!mutual_best || colm[j] == i)) {
      pairs[2 * nmatch] = i;
      pairs[2 * nmatch + 1] = j;
      nmatch++;
    }
  }
  free(dot); free(raw); free(rowm); free(colm);
  return nmatch;
}
