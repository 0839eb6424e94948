/*
 * hess_abi.h -- C ABI of the MI355X-native Hessian + SIFT-descriptor hot path.
 *
 * This is the drop-in boundary of the build: plain C types, pointers and sizes only.
 * It is what the host-side `SiftGPU` class (include/SiftGPU.h, libsiftgpu.so) binds in
 * place of the reference's CUDA backend objects.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference tree, src/SiftGPU/...).
 *
 * The same signatures, prefixed `hess_cpu_`, are implemented by the CPU oracle
 * (oracle/hess_oracle.c) so that the parity tests are backend-agnostic.  The oracle is
 * test infrastructure only; nothing in the product links it.
 *
 * Error convention: functions returning int return 0 (HESS_OK) on success and a negative
 * hess_status on failure; no exceptions cross this boundary; hess_last_error() gives text.
 * One context per host thread / per device (reference: one SiftGPU instance per thread per
 * device, TestWin/MultiThreadSIFT.cpp:90-133).
 */
#ifndef HESS_ABI_H
#define HESS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: hess_params.reserved[] must be zero (the oracle-only `detector` word of version 1 is gone from the product's
 *    struct); hess_submit_host, hess_device_count, hess_debug_key_levels (now covering one run), hess_debug_regrown.
 * 3: hess_share_results / hess_shared_results_info (added during version 2 without a bump: round 3);
 *    hess_params.descriptor_order (the first of the reserved words: a version-2 struct, all zero there, asks for
 *    the default); hess_count / hess_fetch / hess_device_results refuse (HESS_ERR_ARG / _STATE) after a failed run
 *    instead of handing out the results of the run before.
 * 4: HESS_DESC_ORDER_PIXEL, the order hess_default_params now chooses (a version-3 struct keeps what it asks for: 0 is
 *    the interleaved order); hess_debug_keep_levels (the top Gaussian level of an octave is no longer written to HBM
 *    unless asked for); a context whose DMA copy was lost refuses further runs (HESS_ERR_DEVICE, "poisoned"). */
#define HESS_ABI_VERSION 4

typedef enum hess_status {
  HESS_OK = 0,
  HESS_ERR_ARG = -1,      /* bad argument (null pointer, non-positive size, bad index)   */
  HESS_ERR_TOO_BIG = -2,  /* image exceeds tex_max_dim and auto_downscale is off
                             (reference: exit() in PyramidCU.cpp:170-174 -> error return)  */
  HESS_ERR_DEVICE = -3,   /* HIP runtime error (reference: CheckErrorCUDA -> RunSIFT 0)  */
  HESS_ERR_NOMEM = -4,    /* allocation failed                                           */
  HESS_ERR_STATE = -5,    /* call out of order (fetch before run, ...)                   */
  HESS_ERR_UNSUPPORTED = -6
} hess_status;

/* Feature types, reference config.h:45-50. */
enum { HESS_TYPE_DARK_BLOB = 0, HESS_TYPE_BRIGHT_BLOB = 1, HESS_TYPE_SADDLE = 2, HESS_TYPE_NONE = 3 };

/* Truncation methods, reference SiftPyramid.h:73-77 (-tc/-tc1, -tc2, -tc3, -topk). */
enum { HESS_TRUNC_HIGHEST_0 = 0, HESS_TRUNC_HIGHEST_1 = 1, HESS_TRUNC_LOWEST = 2, HESS_TRUNC_TOPK = 3 };

/* Order in which the samples of a descriptor cell are added into its orientation bins (ComputeDescriptor_Kernel,
 * ProgramCU.cu:1723-1774: one thread per cell walks the cell's box row by row and adds as it goes).
 *   SEQUENTIAL             the reference's own order, sample after sample: bit-identical to a sequential scan
 *   INTERLEAVED            four partial sums per bin -- the samples at positions 0, 1, 2, 3 modulo 4 of that walk --
 *                          added as (p0 + p1) + (p2 + p3): every lane of the kernel keeps its own samples' sums, no
 *                          cross-lane exchange per sample; differs from the sequential sum by rounding only (<= 3e-7
 *                          on unit-norm descriptors, measured; the tests bound it by 1e-6; north star: 1e-4);
 *                          descriptor kernel 16 % faster than SEQUENTIAL
 *   PIXEL (default)        every pixel of the footprint is evaluated ONCE (gather, Gaussian weight, bin split in the
 *                          keypoint's frame: the reference re-evaluates it for each of the up to four cells it belongs
 *                          to) and adds its weight to <= 2 x 2 cells x 2 bins in 32-bit fixed point with a per-keypoint
 *                          power-of-two scale -- integer sums, so the result does not depend on the order of the
 *                          additions (any parallel schedule gives the same bits; the oracle's plain loop does).
 *                          Distance to SEQUENTIAL on unit-norm descriptors, measured per BASELINE config
 *                          (tests/test_gpu_parity.py): 640x480 (configs[2]) 2.5e-6, 1920x1080 (configs[1]) 6e-6, 4096x4096
 *                          (configs[4]) 1.6e-5 -- ONE bound for all of them in the tests, TOL_ORDER = 3e-5 (bench.py checks
 *                          its 1080p image against 1e-5); north star 1e-4.  Most of the distance is SEQUENTIAL's own float
 *                          rounding (it rounds every cell centre at the magnitude of the image coordinate: 1e-5 relative
 *                          at x = 4000): a float64 evaluation of the reference's formula is within 2.5e-7 of PIXEL on all
 *                          three (TOL_EXACT = 1e-6).  Descriptor kernel a further 21 % faster, whole path + 9 %.
 *                          Needs luminance in [0, 1] for its overflow bound, so two inputs keep the INTERLEAVED order --
 *                          permanently, not as a transition: float pixels (HESS_PIX_F32, taken as they are) and user
 *                          keypoint lists (hess_set_keypoints: any scale at any level).  All three orders have a bitwise
 *                          oracle restatement and stay in tools/fuzz_parity.py and the robustness matrix. */
enum { HESS_DESC_ORDER_INTERLEAVED = 0, HESS_DESC_ORDER_SEQUENTIAL = 1, HESS_DESC_ORDER_PIXEL = 2 };

/* Pixel formats accepted by hess_run_* (the GL enums of SiftGPU::RunSIFT(w,h,data,fmt,type)
 * are mapped onto these by the C++ class; reference GLTexImage.cpp:918-1036). */
enum { HESS_FMT_LUM = 1, HESS_FMT_LUM_ALPHA = 2, HESS_FMT_RGB = 3, HESS_FMT_RGBA = 4,
       HESS_FMT_BGR = 5, HESS_FMT_BGRA = 6 };
enum { HESS_PIX_U8 = 1, HESS_PIX_U16 = 2, HESS_PIX_F32 = 3 };

/*
 * Tunables.  Replaces the process-global GlobalParam statics (GlobalUtil.cpp:51-144) and
 * the SiftParam members (SiftGPU.h:59-86) that the hot path reads.  A zero in a field
 * marked (0=default) selects the reference default, as SiftParam::ParseSiftParam
 * (SiftGPU.cpp:491-563) does.
 */
typedef struct hess_params {
  int32_t abi_version;          /* must be HESS_ABI_VERSION                                   */
  int32_t dog_level_num;        /* -d   (0=default 3)      scales per octave                  */
  float sigma0;                 /*      (0=default 1.6)                                       */
  float sigman;                 /*      (0=default 0.5)                                       */
  float dog_threshold;          /* -t   (0=default 0.02/dog_level_num)                        */
  float edge_threshold;         /* -e   (0=default 10)                                        */
  float filter_width_factor;    /* -f   (0=default 4.0)   GlobalUtil.cpp:62                   */
  float orient_window_factor;   /* -w   (0=default 2.0)   GlobalUtil.cpp:134                  */
  float orient_gaussian_factor; /*      (0=default 1.5)   GlobalUtil.cpp:135                  */
  float desc_window_factor;     /* -dw  (0=default 3.0)   GlobalUtil.cpp:63                   */
  int32_t first_octave;         /* -fo  (default 0)  SiftGPU.cpp:1166-1175: the reference's Hessian build
                                   takes only >= 0 at its option parser (and so does this build's
                                   SiftGPU::ParseParam); -1..-3 = first octave up-sampled by 2^-fo
                                   (PyramidCU.cpp:120-138,1517-1525, ProgramCU.cu:233-310), kept below the
                                   parser in the reference and reachable here through this struct only */
  int32_t octave_num;           /* -no  (<=0 = no limit)  GlobalUtil.cpp:123                  */
  int32_t subpixel;             /* -s   (default 1)                                           */
  int32_t max_orientation;      /* -m   (default 2; clamped to 1..4 by ParseParam)            */
  int32_t fixed_orientation;    /* -ofix                                                      */
  int32_t lowe_origin;          /* -loweo                                                     */
  int32_t half_sift;            /* -half  64-d descriptor, unsigned gradient                  */
  int32_t compute_descriptors;  /* 0 with -sd                                                 */
  int32_t normalize;            /* 1 (GlobalUtil::_NormalizedSIFT)                            */
  int32_t truncate_method;      /* HESS_TRUNC_*                                               */
  int32_t feature_count_threshold; /* -tc* / -topk value (<=0 = off)                          */
  int32_t tex_max_dim;          /* -maxd (0=default 3200)                                     */
  int32_t auto_downscale;       /* -ads                                                       */
  int32_t verbose;              /* bit 0: messages on stderr; bit 1: stage timers (hess_timing entries 2..10; the
                                   reference's _timingS, SiftGPU.cpp:433-464).  0 = silent, total time only: the
                                   events between stages cost about 6 us each on the device                  */
  int32_t dynamic_indexing;     /* -di  descriptor bins indexed dynamically (GlobalUtil.cpp:108,
                                   ProgramCU.cu:1755-1771): a sample whose bin coordinate rounds up to
                                   exactly 8.0 is then added to bin 8 (folded into bin 0), not dropped */
  int32_t descriptor_order;     /* HESS_DESC_ORDER_*: how a descriptor bin's samples are added up (see the enum)  */
  int32_t reserved[6];          /* must be zero: hess_create refuses anything else (word 0 is where the test
                                   oracle keeps its detector switch, oracle/hess_oracle.h -- not a product option) */
} hess_params;

/* Binary-identical to SiftGPU::SiftKeypoint (SiftGPU.h:108-116): 24 bytes. */
typedef struct hess_keypoint {
  float x, y, s, o;
  float response;
  uint16_t level;
  uint16_t type;
} hess_keypoint;

/* Stage timers in the order of SiftGPU::_timing[12] (config.h:17-31), milliseconds. */
enum { HESS_T_LOAD = 0, HESS_T_ALLOC, HESS_T_PYRAMID, HESS_T_DETECT, HESS_T_LIST, HESS_T_ORIENT,
       HESS_T_MULTI_ORIENT, HESS_T_DOWNLOAD, HESS_T_DESCRIPTOR, HESS_T_VBO, HESS_T_REDUCTION,
       HESS_T_TOTAL, HESS_T_COUNT };

/* Stage dumps for the parity tests (hess_debug_level `what`). */
enum { HESS_DBG_GAUSS = 0,  /* Gaussian level, wa*h floats                                   */
       HESS_DBG_DETH = 1,   /* det-Hessian * sigma^4, wa*h floats                            */
       HESS_DBG_GOT = 2     /* (|grad|/2, theta) interleaved, 2*wa*h floats, levels 1..dog   */ };

/* One raw detection (before top-K / orientation), for hess_debug_list. 32 bytes. */
typedef struct hess_rawkey {
  int32_t level_index;  /* octave*dog_level_num + (level-1)                                  */
  int32_t col, row;
  uint32_t packed;      /* half(response)<<16 | 0x4 | type  (key-map pixel .x, ProgramCU.cu:865) */
  float dx, dy, ds;     /* sub-pixel offsets (ProgramCU.cu:868)                              */
  uint32_t pad;
} hess_rawkey;

typedef struct hess_ctx hess_ctx; /* opaque */

/* Fill *p with the reference defaults (all "0=default" fields resolved). */
void hess_default_params(hess_params* p);

/* Number of usable HIP devices (0 without a GPU): what the reference's callers hard-code as "device 0 and 1"
 * (TestWin/MultiThreadSIFT.cpp:233-234; ProgramCU::CheckCudaDevice, ProgramCU.cu:3386-3440, rejects the rest). */
int hess_device_count(void);

/* 1 when the loaded library is the DEVELOPER build (-DHESS_DEV_SWITCHES: schedule A/B switches, fault injection and the
 * other test hooks read from the environment, hessgpu_amd/csrc/hess_ctx.h), 0 for the shipped library, which reads
 * HESS_SHARE_DIR / TMPDIR, HESS_COPY_TIMEOUT_S and HESS_DELIVERY only.  No reference counterpart.  (Added in version 4.) */
int hess_dev_switches(void);

/* Replaces SiftGPU::CreateContextGL/VerifyContextGL -> InitSiftGPU -> new PyramidCU
 * (SiftGPU.cpp:149-227,1516-1539) and ProgramCU::CheckCudaDevice (ProgramCU.cu:3386-3440).
 * `device` is the HIP device ordinal.  Returns NULL on failure. */
hess_ctx* hess_create(int device, const hess_params* params);
void hess_destroy(hess_ctx* ctx);

/* Replaces PyramidCU::InitPyramid/ResizePyramid/ResizeFeatureStorage (PyramidCU.cpp:113-489)
 * and SiftGPU::AllocatePyramid: pre-allocates pyramids for `batch` images of w*h so that
 * hess_run_* does no allocation (grow-only, like the reference).  It also creates the runtime objects a batch of that
 * size uses (result copier thread and its DMA queue, the stream's hardware queue): the first batch does not pay for them. */
int hess_reserve(hess_ctx* ctx, int width, int height, int batch);

/* Replaces SiftGPU::RunSIFT(w,h,data,fmt,type) -> GLTexInput::SetImageData (CUDA branch,
 * GLTexImage.cpp:918-1036) -> SiftPyramid::RunSIFT (SiftPyramid.cpp:53-198) for `batch`
 * independent images of identical size laid out back to back (`image_stride` bytes apart,
 * rows `pitch` bytes apart).  Host pointer version: pixels are copied to the device. */
int hess_run_host(hess_ctx* ctx, const void* pixels, int width, int height, int pitch,
                  size_t image_stride, int batch, int format, int pixtype);
/* Asynchronous pair: hess_submit_device enqueues the whole path on the context's stream and returns;
 * hess_wait blocks until the results are in host memory.  Two contexts used alternately overlap the
 * result transfer of one batch with the kernels of the next.  One submitted batch per context. */
int hess_submit_device(hess_ctx* ctx, const void* dev_pixels, int width, int height, int pitch,
                       size_t image_stride, int batch, int format, int pixtype);
/* The same from host memory: the pixels cross with one asynchronous transfer on the context's stream (straight
 * from the caller's buffer when it is pinned, through the context's pinned staging buffer otherwise; the buffer
 * may be reused as soon as the call returns in the second case, after hess_wait in the first). */
int hess_submit_host(hess_ctx* ctx, const void* pixels, int width, int height, int pitch,
                     size_t image_stride, int batch, int format, int pixtype);
int hess_wait(hess_ctx* ctx);
/* Same, pixels already resident in device memory (HBM) of ctx's device. */
int hess_run_device(hess_ctx* ctx, const void* dev_pixels, int width, int height, int pitch,
                    size_t image_stride, int batch, int format, int pixtype);

/* The pixels of the last batch handed over with hess_submit_host / hess_run_host, as they were handed over (the
 * context keeps them in its staging area until the next batch).  Replaces GLTexInput keeping the converted image in
 * _pixel_data between calls (GLTexImage.cpp:918-1036, PyramidCU.cpp:1458-1462), which is what lets the reference's
 * SiftGPU::RunSIFT() run the current image again: this build's SiftGPU class borrows the caller's pointer for the call
 * and only comes back for the pixels if the image is run again without being handed over again. */
int hess_last_input(hess_ctx* ctx, void* out, size_t bytes);

/* Replaces SiftGPU::SetKeypointList -> SiftPyramid::SetKeypointList (SiftPyramid.cpp:326-355): the
 * NEXT hess_run_* call (one image) skips detection and computes orientation (unless
 * keys_have_orientation) and descriptors for these keypoints (PyramidCU::GenerateFeatureListTex,
 * PyramidCU.cpp:555-718); results come back in input order; the list is cleared afterwards. */
int hess_set_keypoints(hess_ctx* ctx, const hess_keypoint* keys, int num, int keys_have_orientation);
/* Replaces SiftGPU::RunSIFT(num, keys, keys_have_orientation) (SiftGPU.cpp:307-315): the same on the
 * CURRENT image (first image of the last run) without rebuilding the pyramid. */
int hess_run_keypoints(hess_ctx* ctx, const hess_keypoint* keys, int num, int keys_have_orientation);

/* Replaces SiftGPU::GetFeatureNum (SiftGPU.cpp:1551-1554) for image `img` of the last batch. */
int hess_count(hess_ctx* ctx, int img);
/* Descriptor length of the last run: 128, 64 (-half) or 0 (-sd). */
int hess_desc_dim(hess_ctx* ctx);
/* Replaces SiftGPU::GetFeatureVector -> SiftPyramid::CopyFeatureVector (SiftPyramid.cpp:313-324).
 * Either output may be NULL.  keys: hess_count() records; desc: hess_count()*hess_desc_dim() floats. */
int hess_fetch(hess_ctx* ctx, int img, hess_keypoint* keys, float* desc);

/* Device-resident results of the last run, for consumers that stay on the GPU (the multi-GPU
 * gather over RCCL): keys = [total] hess_keypoint, desc = [total][dim] float, the images of the
 * batch back to back (image b starts at hess_count(0)+..+hess_count(b-1)); *total = records in use.
 * Pointers stay valid until the next run. */
int hess_device_results(hess_ctx* ctx, const void** keys, const void** desc, int* total);

/* Pyramid geometry of the last run (PyramidCU.cpp:238-245,274-309): number of octaves, and per
 * octave the aligned width / height. Arrays must hold >= 32 entries. Returns octave count. */
int hess_geometry(hess_ctx* ctx, int* widths, int* heights);

/* Parity hooks (no reference counterpart; reference has only the GL viewer's level display). */
int hess_debug_level(hess_ctx* ctx, int img, int octave, int level, int what, float* out);
/* Parity hook for the reference's feature file (tests/test_reference_fixture.py): describe user keypoint k
 * at level index levels[k] = octave*dog_level_num + (level-1) instead of the level the scale rule of
 * GenerateFeatureListTex picks (-1 keeps the rule; NULL/0 clears).  Applies to the NEXT keypoint-list run only
 * (hess_set_keypoints + hess_run_*, or hess_run_keypoints) and is cleared by it. */
int hess_debug_key_levels(hess_ctx* ctx, const int* levels, int num);
/* Robustness hook: how many times this context has grown its feature storage after an overflow and run a batch
 * again (the reference grows its lists per image, PyramidCU.cpp:393-397).  In the developer build the environment
 * variable HESS_INITIAL_CAP=<n> makes a new context start with room for n detections per image so that tests can force it. */
int hess_debug_regrown(hess_ctx* ctx);
/* Parity hook: the top Gaussian level of every octave (level dog+1) is nobody's input -- the launch that produces it
 * computes its det-Hessian from the output tile and does NOT write the level to HBM (the reference materialises it,
 * PyramidCU.cpp:1486-1558, and reads it back once, :1576-1591).  on != 0: the following runs of this context store it
 * as well, so that hess_debug_level(HESS_DBG_GAUSS, level dog+1) can return it (HESS_ERR_STATE otherwise). */
int hess_debug_keep_levels(hess_ctx* ctx, int on);

/* Multi-process jobs on one node (one process per GPU): keep this context's pinned host result buffers in
 * POSIX shared memory so that another process of the node -- the rank that collects the global batch -- reads them in
 * place.  Every GPU then delivers over its own host link; nothing is funnelled through the collecting rank's.
 *   "/<name>.h"      4096-byte directory: { u32 magic 'HESS', u32 gen_keys, u32 gen_desc, u32 pad, u64 keys_bytes, u64 desc_bytes }
 *   "/<name>.k<gen>" keypoint records of the last batch, images back to back (hess_keypoint[total])
 *   "/<name>.d<gen>" descriptors of the last batch (float[total][dim])
 * A generation number grows when a buffer is reallocated (a reader re-maps when it changes; the old object is
 * unlinked).  The per-image counts travel by the job's own control channel (hess_count); the data of a batch is
 * complete when hess_wait / hess_run_* has returned and stays until the context's next batch is submitted.
 * Call after hess_create, before the first batch; the objects are unlinked by hess_destroy.  name: no '/'. */
int hess_share_results(hess_ctx* ctx, const char* name);
int hess_shared_results_info(hess_ctx* ctx, unsigned* gen_keys, unsigned* gen_desc, size_t* keys_bytes, size_t* desc_bytes);
/* Raw detections of image `img` in list order; returns the count (<= cap written). */
int hess_debug_list(hess_ctx* ctx, int img, hess_rawkey* out, int cap);

/* Stage times of the last hess_run_* in ms, HESS_T_COUNT floats (SiftGPU::_timing, config.h:17-31). */
const float* hess_timing(hess_ctx* ctx);
const char* hess_last_error(hess_ctx* ctx);

/* Per-kernel device-time accounting (hipEvents on the context's stream) for bench.py's
 * roofline leg.  Off by default.  Kernel ids: HESS_K_*.  HESS_K_GAUSS_OCT0 repeats the Gaussian launches that work on
 * octave 0 (they are counted in HESS_K_GAUSS as well): the launches large enough to be bound by memory bandwidth
 * rather than by the latency of the dependent chain. */
enum { HESS_K_GAUSS = 0, HESS_K_DOWNSAMPLE, HESS_K_HESSIAN, HESS_K_EXTREMA, HESS_K_TOPK,
       HESS_K_ORIENT, HESS_K_DESCRIPTOR, HESS_K_INPUT, HESS_K_GAUSS_OCT0, HESS_K_COUNT };
int hess_profile_enable(hess_ctx* ctx, int on);
/* Accumulated since the last hess_profile_reset: total ms, launches, algorithmic bytes. */
int hess_profile_get(hess_ctx* ctx, int kernel, double* ms, long long* launches, double* bytes);
int hess_profile_reset(hess_ctx* ctx);
/* Bytes of the REFERENCE's array layout (every array written once and read once) that the launches counted
 * under `kernel` neither wrote nor read because the array never left LDS: the octave's top Gaussian level (its det-H
 * comes out of the launch that produces it) and level 0 of octave 0 of a u8 image -- 8 bytes per pixel each.
 * hess_profile_get's bytes are what the launches have to move; bytes + this = the layout's figure. */
int hess_profile_get_in_lds(hess_ctx* ctx, int kernel, double* bytes);

/* ---- Descriptor matcher: replaces SiftMatchGPU's CUDA flavour, SiftMatchCU
 * (SiftMatchCU.cpp:71-176) and MultiplyDescriptor(_G)_Kernel / RowMatch_Kernel / ColMatch_Kernel
 * (ProgramCU.cu:3455-3843).  Descriptors: num x 128 unsigned bytes (512*d rounded); locations: (x,y). */
typedef struct hess_matcher hess_matcher;
hess_matcher* hess_matcher_create(int device, int max_sift);          /* SiftMatchGPU(max_sift) */
void hess_matcher_destroy(hess_matcher* m);
int hess_matcher_set_max(hess_matcher* m, int max_sift);              /* SetMaxSift */
int hess_matcher_set_descriptors(hess_matcher* m, int index, int num, const unsigned char* des);
int hess_matcher_set_descriptors_f32(hess_matcher* m, int index, int num, const float* des);
int hess_matcher_set_locations(hess_matcher* m, int index, const float* locations, int gap);
/* GetSiftMatch (H = F = NULL) / GetGuidedSiftMatch; pairs: max_match x 2 ints; returns #matches. */
int hess_matcher_match(hess_matcher* m, int max_match, int* pairs, const float* H, const float* F,
                       float distmax, float ratiomax, float hdistmax, float fdistmax, int mutual_best);
float hess_matcher_last_ms(hess_matcher* m);   /* device time of the last match's kernels */
const char* hess_matcher_last_error(hess_matcher* m);

/* Test hook: evaluate one of the device's elementary functions (hess_devmath.h) on n inputs.
 * which: 0 exp(a) 1 atan2(a,b) 2 sin(a) 3 cos(a) 4 float->half bits 5 half bits->float 6 a/b 7 sqrt(a)
 * 8 the u8 -> [0,1] conversion a/255 for integer a in 0..255. */
int hess_math_probe(hess_ctx* ctx, int which, const float* a, const float* b, float* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* HESS_ABI_H */
