/*
 * hess_oracle.h -- CPU ORACLE for the Hessian + SIFT-descriptor hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this.  The product (hessgpu_amd/, libhessgpu.so,
 * libsiftgpu.so) never does, and fails loudly when its HIP library is missing.
 *
 * PARITY: the DETECTOR (det-Hessian extrema, sub-pixel refinement, top-K) is UNPINNED -- the reference
 * (sloup/hessgpu) ships no golden vectors, known-answer tests or fixtures for the Hessian path (SURVEY.md
 * section 8c) and cannot be built here (needs nvcc + CUDA runtime + GLEW/GLUT/DevIL).  Everything AFTER
 * detection is pinned by the one feature file the reference ships, doc/evaluation/box.siftgpu (DoG build of
 * the family, which shares the pyramid definition and the orientation / descriptor / normalisation /
 * SaveSIFT code): with the first octave up-sampled as the file's `-fo -1` asks, every one of the 581
 * keypoints whose descriptor footprint lies inside the image is reproduced to <= 1 count of 512 in all 128
 * values, and the computed orientations fall within half an 8-bit step of the file's for 99.7 % of them
 * (tests/test_reference_fixture.py, tests/box_fixture.py).  Run as the build that wrote the file (detector = 2:
 * DoG planes, no sign conditions, two-peak orientation rule -- the reference's #ifndef GPU_HESSIAN lines -- on otherwise
 * the same code) the oracle also re-detects the file's features from the pixels: 671 features against 673, 664
 * matched one to one within the file's rounding, descriptors of all interior ones within 1 count.  Without
 * reference-made evidence remain only the Hessian-specific lines (det-H formula, sign conditions, type) and the
 * top-K tie rule.  The rest of the oracle is checked against (a) IEEE
 * facts it must satisfy (half conversions vs numpy.float16, math functions vs libm) and (b) an independently
 * written NumPy restatement (tests/np_restatement.py).
 *
 * Same entry points as include/hess_abi.h with the prefix hess_cpu_.
 */
#ifndef HESS_ORACLE_H
#define HESS_ORACLE_H

#include "../include/hess_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hess_cpu_ctx hess_cpu_ctx;

/* The oracle's one extension of hess_params: which detector it runs.  It lives in reserved word 0, which the
 * product requires to be zero (hess_create refuses the struct otherwise), so it is not an option of the product.
 *   0 = determinant of Hessian (GPU_HESSIAN: what the product computes)
 *   1 = difference of Gaussians as the reference compiles without GPU_HESSIAN (config.h:36)
 *   2 = the same with the level sigmas of the version that wrote doc/evaluation/box.siftgpu */
#define HESS_ORACLE_DETECTOR(p) ((p)->reserved[0])
/* Analysis switch, reserved word 1 (zero for the product): sample-window clamps at the image border.
 *   0 = the CUDA path's rule (sample centres in [1.5, dim-1.5]; also the unpacked GLSL shaders')
 *   1 = the packed GLSL shaders' rule (box clamped to [2, dim-3], widened to whole 2x2 texels) */
#define HESS_ORACLE_BORDER(p) ((p)->reserved[1])
/* Analysis value of hess_params.descriptor_order, oracle only (hess_create refuses it): the reference's descriptor
 * formula (ProgramCU.cu:1690-1790, per cell, + normalisation :1950-2054) evaluated in DOUBLE precision from the same
 * float (gradient, theta) planes -- what every float summation order approximates.  The tests use it to say how far
 * each declared order is from the formula itself: the reference's sequential float order carries its own rounding
 * (the cell centres are rounded at the magnitude of the image coordinate: 1e-5 relative at x = 4000). */
#define HESS_ORACLE_DESC_EXACT 3

void hess_cpu_default_params(hess_params* p);
hess_cpu_ctx* hess_cpu_create(const hess_params* params);
void hess_cpu_destroy(hess_cpu_ctx* ctx);
/* Worker threads for the data-parallel loops (OpenMP); 1 = scalar port. */
void hess_cpu_set_threads(hess_cpu_ctx* ctx, int threads);
/* Keep every image's pyramid after a run (needed by hess_cpu_debug_level); default on. */
void hess_cpu_keep_levels(hess_cpu_ctx* ctx, int on);

int hess_cpu_run_host(hess_cpu_ctx* ctx, const void* pixels, int width, int height, int pitch,
                      size_t image_stride, int batch, int format, int pixtype);
int hess_cpu_set_keypoints(hess_cpu_ctx* ctx, const hess_keypoint* keys, int num, int keys_have_orientation);
int hess_cpu_run_keypoints(hess_cpu_ctx* ctx, const hess_keypoint* keys, int num, int keys_have_orientation);
/* Analysis hook: explicit level index (octave * dog + level - 1, or -1 = scale rule) per user keypoint. */
int hess_cpu_debug_key_levels(hess_cpu_ctx* ctx, const int* levels, int num);
int hess_cpu_count(hess_cpu_ctx* ctx, int img);
int hess_cpu_desc_dim(hess_cpu_ctx* ctx);
int hess_cpu_fetch(hess_cpu_ctx* ctx, int img, hess_keypoint* keys, float* desc);
int hess_cpu_geometry(hess_cpu_ctx* ctx, int* widths, int* heights);
int hess_cpu_debug_level(hess_cpu_ctx* ctx, int img, int octave, int level, int what, float* out);
int hess_cpu_debug_list(hess_cpu_ctx* ctx, int img, hess_rawkey* out, int cap);
const float* hess_cpu_timing(hess_cpu_ctx* ctx);
const char* hess_cpu_last_error(hess_cpu_ctx* ctx);

/* Schedule inspection for tests: taps of the blur that produces `level` (level 0 = initial
 * smoothing) into taps[33]; returns the tap count (0 = no filtering). */
int hess_cpu_filter_taps(hess_cpu_ctx* ctx, int level, float* taps);
float hess_cpu_level_sigma(hess_cpu_ctx* ctx, int level);

/* Descriptor matcher (SiftMatchGPU, CUDA flavour: SiftMatchCU.cpp:71-176, ProgramCU.cu:3455-3843).
 * des: num x 128 unsigned bytes; loc: num x (x,y) floats; H, F: 3x3 row-major, all four NULL for the
 * unguided match; pairs: max_match x 2 ints.  Returns the number of matches. */
void hess_cpu_match_quantize(const float* desc, int count, unsigned char* out);
int hess_cpu_match(const unsigned char* des1, int num1, const unsigned char* des2, int num2, const float* loc1,
                   const float* loc2, const float* H, const float* F, float distmax, float ratiomax,
                   float hdistmax, float fdistmax, int mutual_best, int max_match, int* pairs);

/* Elementary-function probes for tests/test_oracle_math.py. */
float hess_cpu_expf(float x);
float hess_cpu_atan2f(float y, float x);
void hess_cpu_sincosf(float a, float* s, float* c);
unsigned short hess_cpu_f2h(float f);
float hess_cpu_h2f(unsigned short h);

#ifdef __cplusplus
}
#endif
#endif
