// hess_dev.h -- structures shared by the host pipeline and the HIP kernels, and the kernel
// launcher prototypes.  Vocabulary follows the reference: octave, level, Gaussian (gus),
// det-Hessian (the reference's "dog" slot), got = (gradient, theta), key, feature list.
//
// HBM layout (one context, batch capacity B, SURVEY.md section 8 "Array sizes"):
//   gauss, deth : float planes  [octave][level 0..dog+1][image][h_o][wa_o]
//   got         : float2 planes [octave][level 1..dog  ][image][h_o][wa_o]
//   per image   : row bit-masks / row counts of the extrema scan, raw detection list,
//                 selected list, feature records, output keypoints + descriptors.
// All images of a batch have the same size, so one launch covers the batch (blockIdx.z or a
// flattened row index) and small octaves still fill the chip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hess {

constexpr int kMaxOct = 16;
constexpr int kMaxDog = 10;
constexpr int kMaxLev = kMaxDog + 2;
constexpr int kMaxTaps = 33;  // KERNEL_MAX_WIDTH, ProgramCU.cu:42
// streaming extrema scan (k_detect.hip): owned columns per strip, rows per wavefront segment
#ifndef HESS_STREAM_NC
#define HESS_STREAM_NC 2
#endif
constexpr int kStreamCols = HESS_STREAM_NC;        // columns per lane
constexpr int kStreamPitch = 62 * kStreamCols;     // lanes 1..62 own columns, lanes 0 and 63 supply the halo
#ifndef HESS_STREAM_ROWS
#define HESS_STREAM_ROWS 24
#endif
constexpr int kStreamRows = HESS_STREAM_ROWS;
constexpr int kHistBins = 32768;  // abs(half) keys of the top-K selection

// Streaming ("non-temporal") stores for the planes one stage writes and a much later one reads (det-H: the extrema
// scan after the whole pyramid; gradient/theta: orientation and descriptors): they should not displace the Gaussian
// level the NEXT launch is about to read from the last-level cache.  Same-call A/B, batches of 8: Gaussian stage
// 0.546 -> 0.514 ms per step, headline + 2.7 % (profiles/r03_experiments/streaming_stores.txt).
typedef float hess_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream_f4(float* p, float x, float y, float z, float w) {
#ifdef HESS_NO_STREAM_STORES  // A/B build: plain stores
  *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w);
#else
  __builtin_nontemporal_store((hess_v4f){x, y, z, w}, reinterpret_cast<hess_v4f*>(p));
#endif
}

struct OctGeom {
  int wa, h;            // 4-aligned width, height (PyramidCU.cpp:274-309)
  int plane;            // wa*h
  int w64;              // mask words per row = ceil(wa/64)
  long long lvl_off;    // element offset of [octave][0][0] in gauss/deth (units: floats)
  long long got_off;    // element offset of [octave][1][0] in got (units: float2)
  int row_base;         // first row index of this octave in the per-image row order
  int mask_base;        // first mask word of this octave in the per-image mask array
  int tiles_x;          // LDS-tiled extrema scan: 128 x 4 px tiles per row of tiles
  int tile_base;        // first extrema tile of this octave in the per-image tile order
  int strips;           // streaming extrema scan: 124-column strips per row (k_detect.hip)
  int stream_base;      // first streaming workgroup of this octave in the per-image order
};

struct Geom {
  int noct, dog, nlev;  // octaves, detection levels per octave, noct*dog
  int B;                // batch capacity the planes are laid out for
  int NR;               // rows per image in list order: dog * sum_o h_o
  int NM;               // mask words per image
  int ntiles;           // extrema tiles per image
  int nstream;          // streaming extrema workgroups per image
  int stream_rows;      // rows per wavefront segment of the streaming extrema scan (a multiple of 3)
  OctGeom o[kMaxOct];
};

// Segment length of the streaming extrema scan and the workgroup layout that follows from it (four (strip, segment)
// tasks per workgroup, octaves back to back).  Batches of one or two images take short segments: twice the
// wavefronts, each half as long -- the scan of a single image is a few hundred wavefronts, latency-bound.
inline void set_stream_rows(Geom& g, int rows) {
  g.stream_rows = rows;
  g.nstream = 0;
  for (int o = 0; o < g.noct; o++) {
    g.o[o].stream_base = g.nstream;
    g.nstream += (g.o[o].strips * ((g.o[o].h + rows - 1) / rows) + 3) / 4;
  }
}

struct Taps {
  int fw;               // number of taps (odd, 5..33)
  float k[kMaxTaps];
};

// One raw detection, list order = (level_index, row, col).  Same layout as hess_rawkey.
struct RawKey {
  int level_index;
  int col, row;
  uint32_t packed;      // half(response)<<16 | 0x4 | type  (ProgramCU.cu:865)
  float dx, dy, ds;
  uint32_t pad;
};

// 16-byte feature record after orientation (ProgramCU.cu:1563-1596).
struct FRec {
  uint32_t x, y, z, w;
};

struct HostKeypoint {   // = SiftGPU::SiftKeypoint (SiftGPU.h:108-116)
  float x, y, s, o, response;
  uint16_t level, type;
};

struct DetectParams {
  float thr0, thr, edge;  // 0.8*T (or T), T, (e+1)^2/e   (ProgramCU.cu:897,913)
  int subpixel;
};

struct LimitParams {
  int method;     // HESS_TRUNC_*
  int threshold;  // <=0: off
};

struct OrientParams {
  float gaussian_factor, sample_factor;  // 1.5, 1.5*2.0 (ProgramCU.cu:1637-1638)
  float ln_sigma_step;
  int num_orientation;  // 0 (-ofix), 1 (-m 1), >1 multi
  int subpixel, half_sift;
  int existing;         // 1: user keypoints -- position/scale from the packed record, only .w is written
  float level_sigma[kMaxLev];
};

struct DescParams {
  float window_factor;  // 3.0
  int half_sift, normalize, multi;
  int lowe_origin;
  float octave_sigma;   // 2^ds (PyramidCU.cpp:746-748)
  int dog;
  int dynamic_indexing;  // -di: theta == 8.0 goes to des[8] (ProgramCU.cu:1755-1759) instead of being dropped
  HostKeypoint* hkeys;   // optional pinned-host mirrors of the packed results (same indexing as keys/desc)
  float* hdesc;
  int first_image;       // the launch covers images first_image .. first_image + gridDim.y - 1 of the batch
  int xcd_block;         // features per block of the list handed to one XCD's workgroups (0: plain order)
  int sequential;        // HESS_DESC_ORDER_SEQUENTIAL: bins summed in the reference's sample order (else four interleaved partial sums)
  int pixel;             // HESS_DESC_ORDER_PIXEL: one raster over the footprint, fixed-point sums (descriptor_pixel_kernel); wins over `sequential`
  int px_band;           // pixel raster: pixels per band of rows, 64 .. 4096 (4096 unless HESS_PX_BAND of the developer build says otherwise)
  int part, part_den;    // part_den > 1 (one image per launch): the launch does features [n part / part_den, n (part + 1) / part_den) of the image's n
};

// ---- launchers (each enqueues on `st`, no host synchronisation) ----------------------------

// Separable Gaussian, one level for the whole batch: dst = G(taps) * src with replicated
// borders, tap order and FMA chain of FilterH/FilterV (ProgramCU.cu:117-231).
// src_u8 != nullptr: source is u8 luminance (value/255.0f, GLTexImage.cpp:828), pitch in bytes.
// deth_src != nullptr (float source with src_pitch == wa, src_img_stride == wa*h only): the kernel
// also emits det-Hessian*sigma^4 (and, if got_src != nullptr, gradient/theta) of the SOURCE level
// from the window it has staged anyway (ComputeHessian_Kernel, ProgramCU.cu:523-595).
void launch_gauss(hipStream_t st, const float* src, const uint8_t* src_u8, long long src_pitch,
                  long long src_img_stride, float* dst, int wa, int h, int batch, const Taps& taps,
                  float* deth_src = nullptr, float* got_src = nullptr, float norm_src = 0.0f,
                  float* decim_dst = nullptr, int decim_w = 0, int decim_h = 0);

// One level launch as data (float source with pitch wa), for launch_gauss_pair.
struct GaussJob {
  const float* src;
  float* dst;
  int wa, h;
  Taps taps;
  float* deth_src;
  float* got_src;
  float norm_src;
  float* decim_dst;
  int decim_w, decim_h;
  // a TOP level (the octave's top level): det-H * sigma^4 of the PRODUCED level from the output tile, `dst` may be null
  // (the level is then never written to HBM); zero / zero_bytes: buffers to clear on the side (multiple of 16 bytes)
  float* deth_dst = nullptr;
  float norm_dst = 0.0f;
  void* zero = nullptr;
  size_t zero_bytes = 0;
};
void launch_gauss_job(hipStream_t st, const GaussJob& j, int batch);
// Levels 0 and 1 of octave 0 from u8 pixels in one launch (level 0 stays in LDS; dst0: stored as well); false: tap counts
// not instantiated.
bool gauss_first_available(const Taps& taps0, const Taps& taps1);  // the tap counts launch_gauss_first is instantiated for
bool launch_gauss_first(hipStream_t st, const uint8_t* pixels, long long pitch, long long img_stride, const Taps& taps0,
                        const GaussJob& level1, float* dst0, int batch);
// Two independent level launches in one grid (the top level of an octave and level 1 of the next); false if the pair of
// tap counts is not instantiated: launch them separately then.
bool launch_gauss_pair(hipStream_t st, const GaussJob& a, const GaussJob& b, int batch);

// Levels 1..nlevels (= level_ds) of one octave in one launch (gauss_chain_kernel, k_gauss.hip), and level 0 of the
// next octave: the part of an octave the next one waits for.  The octave's top level, det-H and gradient planes follow
// off the critical path: launch_gauss_multi, launch_hessian_level.
struct ChainJob {
  const float* src0;
  float* dst[4];
  Taps taps[4];
  int nlevels, wa, h;
  float* decim_dst;
  int decim_w, decim_h;
};
bool launch_gauss_chain(hipStream_t st, const ChainJob& j, int batch);
bool gauss_chain_available(const Taps* taps /* [0..level_ds] */, int level_ds);
// Several level launches (same tap count, each with det-H and gradient of its source level) in one grid; false: not
// instantiated for these jobs.  low: the same launch also computes det-H + gradient/theta of levels 0 .. low_nlv-1 of
// octaves >= low_first from HBM (the levels launch_gauss_chain produced without them).
struct LowLevels {
  const Geom* g;
  const float* gauss;
  float* deth;
  float* got;
  const float* norms;  // sigma^4 per level (host)
  int first_oct, nlv;
};
bool launch_gauss_multi(hipStream_t st, const GaussJob* jobs, int njobs, int batch, const LowLevels* low = nullptr);

// Input conversion to float luminance with 2^ds decimation (GLTexImage.cpp:802-916).
void launch_convert(hipStream_t st, const void* src, int format, int pixtype, long long pitch,
                    long long img_stride, int ds, float* dst, int w, int h, int batch);

// Linear up-sampling by 2^log_scale of the converted input (UpsampleKernel, ProgramCU.cu:233-310); only for
// an up-sampled first octave (first_octave < 0 through the C ABI).  src: [batch][h][w], dst: [batch][h<<k][w<<k].
void launch_upsample(hipStream_t st, const float* src, int w, int h, int log_scale, float* dst, int batch);

// Nearest decimation to the next octave (DownsampleKernel, ProgramCU.cu:312-326).
void launch_downsample(hipStream_t st, const float* src, int sw, int splane, float* dst, int dw,
                       int dh, int batch);

// det-Hessian * sigma^4 for levels 0..dog+1 of one octave, and (gradient, theta) for levels
// 1..dog (ComputeHessian_Kernel, ProgramCU.cu:523-595).
// Levels [level_first, level_last] only (the fused Gaussian kernel covers every level that is the
// source of a blur; the top level of an octave has no successor and is done here).
void launch_hessian(hipStream_t st, const Geom& g, int octave, const float* gauss, float* deth,
                    float* got, const float* norms /* host: sigma^4 per level */, int batch,
                    int level_first, int level_last);
// det-H of level `level` of every octave, one launch (no gradient plane); also clears `zero_bytes` (a multiple of 16)
// at `zero` if given: the buffers the detection stages expect zeroed
void launch_hessian_level(hipStream_t st, const Geom& g, const float* gauss, float* deth, int level, float norm,
                          int batch, void* zero = nullptr, size_t zero_bytes = 0);

// The scan's UNORDERED store of detections, per image `stride` records: [ntask tasks x kDetectSlots slots][cap_spill spill
// list].  A scan task (a wavefront's strip segment; a tile of the LDS-tiled scan) fills its own slots and posts how many it
// used in task_count (every task does, so the array needs no clearing); what a task finds beyond its slots goes to the
// image's spill list under spill_count (arrives zeroed).
constexpr int kDetectSlots = 64;
struct DetectStore {
  RawKey* found;
  long long stride;     // records per image
  int ntask;
  int* task_count;      // [batch][ntask]
  int* spill_count;     // [batch]
  int cap_spill;
  unsigned* hist;       // [batch][kHistBins] top-K key histogram, or null
};
int extrema_tasks(const Geom& g);  // scan tasks per image for this geometry (depends on Geom::stream_rows)
// Extrema scan (ComputeKEY_Kernel, ProgramCU.cu:657-882): every accepted pixel as a complete RawKey in the image's
// unordered store, its bit in the row's mask words, its row's count, and -- ds.hist != null -- the top-K key histogram.
// rowmask, rowcnt, ds.spill_count, ds.hist arrive zeroed.
void launch_extrema_mark(hipStream_t st, const Geom& g, const DetectParams& dp, const float* gauss,
                         const float* deth, uint64_t* rowmask, int* rowcnt, const DetectStore& ds, int batch);
bool extrema_streams(const Geom& g);  // the streaming scan applies to this geometry
// List order (ListGen_Kernel, ProgramCU.cu:924-1051, made deterministic): row counts -> exclusive offsets in list order
// (level totals, -tc level truncation: GenerateFeatureList / LimitFeatureCount, PyramidCU.cpp:1283-1368,
// SiftPyramid.cpp:201-278) by the image's first workgroup, and every detection copied to offset(row) + mask bits to its
// left: the raw list in (level, row, col) order.  ticket, flag: one int per image each, arrive zeroed.
void launch_extrema_place(hipStream_t st, const Geom& g, const LimitParams& lp, const DetectStore& ds, const uint64_t* rowmask,
                          const int* rowcnt, int* rowoff, int* level_count, int* raw_total, int* overflow, int* ticket,
                          int* flag, RawKey* raw, int cap_raw, int batch);

// Top-K (SelectTopK, PyramidCU.cpp:1881-1987): keeps the K largest abs(half(response)), ties to
// the lower list index, order preserved; when total < K the list is copied.  One launch (ticketed chunks with a
// look-back over the chunk counts).  scratch: topk_scratch_bytes(cap_raw, batch, nlev) bytes of device memory that
// arrive ZEROED (tickets, chunk words, kept entries per level).
void launch_topk(hipStream_t st, const Geom& g, int K, const RawKey* raw, const int* raw_total,
                 int cap_raw, unsigned* hist, RawKey* sel, int* sel_total, int cap_sel, int batch, void* scratch,
                 int* overflow);  // overflow[2]: raised when the look-back over the chunk words does not complete
size_t topk_scratch_bytes(int cap_raw, int batch, int nlev);

// Orientation (ComputeOrientation_Kernel, ProgramCU.cu:1221-1605): one wavefront per keypoint.
void launch_orientation(hipStream_t st, const Geom& g, const OrientParams& op, const RawKey* list,
                        const int* list_total, int cap_list, const float* got, FRec* recs,
                        int* ocount, int batch);
// Exclusive scan of the per-keypoint orientation counts -> output offsets and feature totals
// (ReshapeFeatureListCPU, PyramidCU.cpp:720-924; LimitFeatureCount(1)).
void launch_feature_scan(hipStream_t st, const Geom& g, const LimitParams& lp, int multi, const RawKey* list,
                         const int* list_total, int cap_list, const int* ocount, int* foffset, int* fsrc,
                         int* feat_total, int* feat_first, int cap_feat, int* overflow, int* img_base, int* host_small,
                         int batch);
// Descriptor + normalisation + host keypoint record (ComputeDescriptor_Kernel /
// NormalizeDescriptor_Kernel, ProgramCU.cu:1650-2054; keypoint unpack PyramidCU.cpp:866-906).
void launch_descriptor(hipStream_t st, const Geom& g, const DescParams& dp, const RawKey* list,
                       int cap_list, const FRec* recs, const int* fsrc, const int* feat_total,
                       const int* feat_first, const int* img_base, const float* got, HostKeypoint* keys,
                       float* desc, int cap_feat, int batch, int seen_features = 0);
// Exclusive prefix of the per-image feature totals: img_base[0..batch] (packed output layout).
// Device evaluation of the elementary functions for the parity tests.
void launch_math_probe(hipStream_t st, int which, const float* a, const float* b, float* out, int n);

}  // namespace hess
