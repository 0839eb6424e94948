// hess_ctx.h -- the context behind the C ABI (include/hess_abi.h) and what its host-side parts share.
//
// Host side of the hot path, one file per concern:
//   hess_plan.hip      parameters -> sigma schedule and taps (SiftParam::ParseSiftParam, SiftGPU.cpp:491-563), octave geometry
//                      and the grow-only HBM / pinned buffers of a batch shape (PyramidCU::InitPyramid / ResizePyramid /
//                      FitPyramid, PyramidCU.cpp:113-489; CuTexImage::InitTexture)
//   hess_schedule.hip  the launch order of one batch on the context's stream (SiftPyramid::RunSIFT, SiftPyramid.cpp:53-198, and
//                      the PyramidCU stages under it), the table of its thresholds, per-kernel hipEvent profiling
//   hess_copier.hip    pixels in / results out: stager threads, the copier thread + SDMA delivery, submit / wait
//   hess_shared.hip    result buffers in node-shared memory (hess_share_results)
//   hess_abi.hip       the extern "C" entry points
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <limits.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>

#include "../../include/hess_abi.h"
#include "hess_dev.h"

namespace hess {


struct Schedule {
  int dog, level_max, level_num, level_ds;
  float sigma[kMaxLev];        // inter-level blur (SiftGPU.cpp:547-552)
  float level_sigma[kMaxLev];  // GetLevelSigma (SiftGPU.cpp:1422-1425)
  float norm[kMaxLev];         // level_sigma^4 as the ComputeHessian wrapper forms it
  float sigma_step, ln_sigma_step;
  Taps taps[kMaxLev];          // taps[l] produces level l from level l-1 (l >= 1)
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  std::string shm;  // non-empty: p is a registered mapping of this POSIX shared memory object ("/name") or, when /dev/shm
                    // had no room, of this file (an absolute path: it has further slashes) -- hess_share_results
};

struct EventPair {
  hipEvent_t a, b;
  int kernel;
  double bytes;
  int kernel2 = -1;  // a second accumulator for the same launch (HESS_K_GAUSS_OCT0), or -1
  double in_lds = 0.0;  // bytes of the reference's array layout this launch neither writes nor reads: the array lives in LDS only
  int count = 1;     // launches between the two events (ProfRun: a run of consecutive launches of one kernel family)
};


// The schedule's thresholds, in one place.  Each was set by a same-call A/B on MI355X; the measurement is named beside it
// (files under profiles/, sections of DESIGN.md).  The developer build can override the ones marked (dev).
namespace policy {
// Batches up to this many images handed over by a caller who WAITS (hess_run_*) are latency cases: they take the chain
// launches for the small octaves, the short scan segments and the descriptor kernel's own host stores.  A pair SUBMITTED
// asynchronously (hess_submit_*) counts as a throughput batch: 17.0 - 17.3 against 12.3 - 12.6 Gpix/s for six pipelined
// contexts (round 5, DESIGN.md "Latency and throughput settings").
constexpr int kLatencyBatch = 2;            // (dev: HESS_MIRROR_MAX_BATCH)
// ... and only while the results of the context's last batch stayed within this: beyond it the kernel's stores wait for the
// host link (a 4096^2 image's 28 MB: 0.77 ms for the mirroring launch against 0.47 for the plain one + DMA, round 6).
constexpr size_t kMirrorMaxBytes = (size_t)16 << 20;   // (dev: HESS_MIRROR_MAX_MB)
// Octaves whose planes of the whole batch are at most this many pixels get levels 1..3 from ONE chain launch (latency
// batches only): below two 960 x 540 planes a level launch is a few dozen workgroups that mostly wait -- one 1080p image 0.400
// -> 0.334 ms; for larger batches the chain loses 2 - 4 % pipelined (profiles/r05_experiments/chain_stamps_pipelined.txt).
constexpr long long kChainMaxPixels = 2LL * 960 * 540;   // (dev: HESS_CHAIN_FROM forces the first chained octave)
// Rows per wavefront segment of the streaming extrema scan: kStreamRows (24, hess_dev.h) for throughput batches, half of it
// for latency batches (twice the wavefronts, each half as long: the scan of one image is a few hundred wavefronts).
constexpr int kLatencyStreamRowsDiv = 2;    // (dev: HESS_STREAM_ROWS)
// A batch of at least this many images delivered by the copier gets its descriptors in two launches over halves of the
// images: the first half's results cross the host link under the second half's kernel (0.53 ms of transfer for eight
// 1080p images leaves the critical path; four groups: -1 % pipelined).  profiles/r03_experiments, DESIGN.md "Result delivery".
constexpr int kSplitDescriptorsFrom = 4;    // (dev: HESS_DESC_PARTS)
// ONE image delivered by the copier (a large one) gets four launches over quarters of its feature list, for the same
// reason: 2.09 -> 1.7 ms per 4096^2 image (profiles/r05_experiments/large_image_delivery.txt).
constexpr int kLargeImageParts = 4;
// Pinned result buffers hold the worst case up front while that stays below this; beyond it they grow by need.
constexpr size_t kHostWorstCaseMax = (size_t)512 << 20;
// Pageable input of at least this size is staged into pinned memory with the helper threads (hess_submit_host): below it
// starting the helpers costs more than the copy (profiles/r03_f_host_path.json).
constexpr size_t kStagerHelpFrom = (size_t)8 << 20;
}  // namespace policy

}  // namespace hess

using namespace hess;


struct PendingRun {
  const void* dev;
  int width, height, pitch, batch, format, pixtype;
  size_t image_stride;
  double t_load_ms;
  bool active;
  bool timed_load;  // ev_load[] bracket a host->device transfer of this batch
};

// Result delivery ("Result delivery and PCIe").  Three ways for the packed keypoints and
// descriptors of a batch to reach pinned host memory:
//   kDeliverMirror  the descriptor kernel stores them into the pinned buffers as well (posted PCIe writes out of the
//                   kernel): no command after the kernels, the shortest path for one image -- but a kernel that waits
//                   for the link holds up the memory path of whatever runs beside it;
//   kDeliverDma     a per-context copier thread waits on the host for the event behind the descriptor kernel, reads
//                   the exact byte count from the pinned count block and hands the copy to an SDMA engine through
//                   ROCr itself (hsa_amd_memory_async_copy_on_engine), then waits for its completion signal; hess_wait
//                   waits for the thread.  Not hipMemcpyAsync: HIP streams share four hardware queues, so a "copy-only"
//                   stream lands on the queue of some context's kernels, where this runtime executes the copy as a
//                   blit kernel behind them (profiles/r03_*: __amd_rocclr_copyBuffer from the copier's stream) -- the
//                   same PCIe-bound shader copy as the mirror.  hipMemcpyAsync on a copy stream remains the fallback
//                   when ROCr refuses (HESS_COPIER=hip selects it for A/B runs);
//   kDeliverBlit    hipMemcpyAsync on the context's stream after hess_wait has read the counts (the fallback, and
//                   what the reference does per level, PyramidCU.cpp:509-532).
enum { kDeliverMirror = 0, kDeliverDma = 1, kDeliverBlit = 2 };

struct Copier {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool started = false, stop = false, has_job = false, done = true;
  int batch = 0;
  int rc = 0;            // result of the last job (hess_status)
  bool overflow = false; // the batch overflowed its feature storage: nothing was copied
  char err[320] = "";    // message of the last failed job (a fixed array: the copier thread must not throw)
  hipStream_t cs = nullptr;     // copy-only stream (fallback path)
  hipEvent_t ev_done = nullptr; // recorded on the context's stream behind the last kernel of a batch
  // A batch's descriptors may be launched in up to kMaxParts groups of images; ev_part[k] is recorded behind group k
  // (the last group's event is ev_done), part_end[k] = first image after group k.  nparts <= 1: one launch.
  static constexpr int kMaxParts = 4;
  hipEvent_t ev_part[kMaxParts - 1] = {nullptr, nullptr, nullptr};
  int nparts = 1, part_end[kMaxParts] = {0, 0, 0, 0};
  bool part_features = false;  // the parts are ranges of ONE image's features (part k ends at feature n (k + 1) / nparts), not groups of images
  // ROCr side (SDMA): agents owning the device / pinned host buffers, engine, completion signal
  bool hsa_ready = false, hsa_failed = false;
  hsa_agent_t gpu_agent{}, cpu_agent{};
  uint32_t engine = 0;          // hsa_amd_sdma_engine_id_t bit, 0 = let ROCr choose
  uint32_t engine_in = 0;       // the same for the host->device upload of pinned pixels
  hsa_signal_t sig{}, sig2{};   // one completion signal per copy in flight (keypoints, descriptors), each armed with 1: tools that
                                // interpose on ROCr (rocprofv3 --memory-copy-trace) expect exactly that of a copy's signal
  // A job that begins with the batch's pixels still on their way (hess_submit_host, pinned input): the upload is an
  // SDMA copy started by the submitting thread with sig_in as its completion signal; the copier thread waits for it ON
  // THE HOST and only then enqueues the kernels -- no command that waits for the transfer ever sits in a hardware
  // queue, which the context's stream shares with other contexts.
  bool upload_first = false;
  bool have_sig_in = false;
  hsa_signal_t sig_in{};
  PendingRun* run = nullptr;
};

// Persistent helper threads that copy pageable input pixels into the context's pinned staging buffer
// (hess_submit_host).  The calling thread walks the chunks in ascending order -- staging the ones nobody has claimed,
// enqueueing every chunk's transfer as soon as it is staged -- while the helpers claim chunks from the END, so the
// early chunks are ready first.  Started at the first pageable submission, reused afterwards: starting threads per
// call cost more than staging one image.
struct Stager {
  static constexpr int kHelpers = 3;
  std::thread th[kHelpers];
  int nth = 0;
  bool tried = false;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  bool stop = false;
  unsigned long long gen = 0;
  const char* src = nullptr;
  char* dst = nullptr;
  size_t bytes = 0, chunk = 0;
  int nchunk = 0, active = 0;
  std::vector<std::atomic<int>> state;  // per chunk: 0 free, 1 claimed, 2 staged
  std::atomic<int> next_hi{-1};
};

struct hess_ctx {
  int device = 0;
  hipStream_t st = nullptr;
  hess_params p;
  Schedule sch;
  // geometry of the current plan
  bool planned = false;
  int in_w = 0, in_h = 0;   // caller's image size
  int ds = 0;               // input decimation (first_octave / auto down-scaling)
  int img_w = 0, img_h = 0; // after decimation and width truncation
  Geom g;
  Taps taps0;               // initial smoothing
  bool has_taps0 = false;
  int cap_raw = 0, cap_sel = 0, cap_feat = 0;
  bool use_topk = false, multi = false;
  int dim = 0;
  // device buffers (grow-only, like CuTexImage::InitTexture)
  int found_tasks = 0;  // scan tasks per image the detection store is laid out for (plan)
  DevBuf gauss, deth, got, input_f32, upsampled, stage, rowoff, level_count, raw_total, found, task_count, raw, sel,
      sel_total, recs, ocount, foffset, fsrc, feat_total, feat_first, img_base, keys, desc;
  // Everything the detection stages expect zeroed lives in one allocation and is cleared by one fill per batch:
  // overflow flags, detection counters, per-row counts, the top-K histogram, the extrema bit masks (views into `zeroed`).
  DevBuf zeroed;
  size_t zeroed_used = 0;
  bool zero_filled = false;  // the running batch's det-H launch has cleared `zeroed`
  struct View { void* p = nullptr; } rowmask, rowcnt, overflow, hist, tk,  // tk: tickets, chunk words, per-level counts of the top-K launch
      found_count, place_ticket, place_flag;                                // detections found per image; extrema_place_kernel's ticket and flag per image
  // host results
  int batch = 0;          // images whose results the context holds (0 after a failed or while a pending run: hess_count /
                          // hess_fetch / hess_device_results refuse instead of handing out the run before)
  int pyramid_batch = 0;  // images whose pyramid is resident (hess_run_keypoints on the current image)
  std::vector<int> counts;
  std::vector<size_t> offs;
  DevBuf h_keys, h_desc, h_small;  // pinned
  // hess_share_results: the two result buffers live in shared memory objects "/<share>.k<n>" / "/<share>.d<n>" that
  // another process of the node can map; a 4 KB directory object "/<share>.h" says which ones are current
  std::string share;
  // (directory layout = hessgpu_amd/dist.py SharedResultsReader._HDR: generations, sizes and the absolute paths of the
  // current buffers -- under /dev/shm, or under HESS_SHARE_DIR / TMPDIR when /dev/shm has no room for them)
  struct ShareDir {
    uint32_t magic, gen_keys, gen_desc, pad;
    uint64_t keys_bytes, desc_bytes;
    char keys_path[1024], desc_path[1024];
  }* share_dir = nullptr;
  bool share_by_need = false;      // the shared result buffers are sized by the batches seen, not for the worst case
  DevBuf h_stage;                  // pinned staging of pageable input pixels (hess_submit_host)
  double stamp_submit0 = 0.0, stamp_submit1 = 0.0;  // HESS_CHAIN_STAMPS
  bool level0_in_lds = false;      // the last run's level 0 of octave 0 was not written to HBM (FIRST tiles)
  size_t last_input_bytes = 0;     // bytes of the last batch handed over by hess_submit_host (still in `stage`)
  hipEvent_t ev_load[2];           // around the host->device transfer of the pixels
  // results written by the descriptor kernel straight into the pinned host buffers (no D2H pass after it)
  bool host_direct = false;        // delivery == kDeliverMirror for the submitted batch
  bool host_fits = false;          // the pinned result buffers hold the worst case of the current plan
  int delivery = kDeliverMirror;   // of the submitted batch (choose_delivery)
  int nparts = 1, part_end[Copier::kMaxParts] = {0, 0, 0, 0};  // the submitted batch's descriptor launches (groups of images)
  bool part_features = false;      // ... or, for one large image, ranges of its features (DescParams::part)
  size_t mirror_max_bytes = policy::kMirrorMaxBytes;   // (dev: HESS_MIRROR_MAX_MB)
  size_t last_result_bytes = 0;    // keypoints + descriptors the last batch delivered, and its size
  int last_result_batch = 0;
  std::atomic<bool> caller_waits{false};  // inside hess_run_* (submit + wait in one call; read by the copier thread, too)
  int delivery_pref = -1;          // HESS_DELIVERY=mirror|dma|blit (-1 = by batch size, see plan())
  int mirror_max_batch = policy::kLatencyBatch;        // (dev: HESS_MIRROR_MAX_BATCH)
  int regrown = 0;                 // times the feature storage was grown after an overflow (hess_debug_regrown)
  int seen_features = 0;           // largest per-image feature count of the last finished batch (0: none yet): sizes the descriptor grid
  int cap_init = 0;                // HESS_INITIAL_CAP: initial raw/feature capacity (developer switch for the grow path)
  bool no_pair = false;            // HESS_NO_PAIR: one launch per pyramid level (A/B switch)
  bool no_first_fusion = false;    // HESS_NO_FIRST_FUSION: level 0 of octave 0 from a launch of its own, written to HBM (A/B switch)
  bool no_top_fusion = false;      // HESS_NO_TOP_FUSION: the top level is stored and its det-H made by a launch of its own (A/B switch)
  bool keep_levels = false;        // hess_debug_keep_levels: the top Gaussian level of every octave is written to HBM as well
  int chain_from = 0;              // HESS_CHAIN_FROM: first octave produced by one level-chain launch (0: by batch size; 99: none)
  bool no_host_upload = false;     // HESS_NO_SIDE_UPLOAD: pinned input is uploaded by a copy on the context's stream (A/B switch)
  int desc_parts = 0;              // HESS_DESC_PARTS: descriptor launches / result transfers per batch (0: default)
  int stream_rows = 0;             // HESS_STREAM_ROWS: rows per wavefront segment of the extrema scan (0: by batch size; A/B switch)
  int desc_px_band = 4096;         // HESS_PX_BAND: pixels per raster band of descriptor_pixel_kernel (test hook: small bands at ordinary footprints)
  int desc_xcd_block = 64;         // HESS_DESC_XCD: features per XCD block of the descriptor launch (0: plain order; A/B switch)
  Copier cp;
  Stager sg;
  // A DMA copy that did not complete in time (or that ROCr reported as failed) may still be in flight, or land later:
  // its targets -- the pinned result buffers, the pixel staging area -- must neither be reused nor freed.  The context
  // refuses every further run (HESS_ERR_DEVICE) and hess_destroy leaves those buffers and the signals alone.
  std::atomic<bool> poisoned{false};
  long long primed_shapes[4] = {-1, -1, -1, -1};  // shapes (width, height, batch) hess_reserve has run a dry batch for
  unsigned primed_next = 0;
  DevBuf prime_px;                 // the dry batch's scratch image (zero pixels)
  PendingRun* pend = nullptr;      // batch submitted with hess_submit_device and not yet waited for
  // user-supplied keypoint list (SiftPyramid::SetKeypointList): used by the next run, then cleared
  std::vector<hess_keypoint> user_keys;
  bool user_have_orientation = false;
  bool user_on_current = false;     // RunSIFT(num, keys, flag): skip filtering, reuse the resident pyramid
  std::vector<int> user_kindex;     // list position -> input index (_keypoint_index)
  std::vector<int> user_levels;     // parity hook: explicit level index per user keypoint (hess_debug_key_levels)
  bool user_result = false;         // last results are in u_keys / u_desc (input order)
  std::vector<hess_keypoint> u_keys;
  std::vector<float> u_desc;
  const RawKey* d_list = nullptr;  // list fed to the orientation stage in the last run
  const int* d_list_total = nullptr;
  int cap_list = 0;
  float timing[HESS_T_COUNT];
  hipEvent_t ev[8];
  bool stage_events = false;  // events between the stages of the running batch (each costs a ~6 us bubble on the stream)
  bool have_ev = false;
  std::string err;
  // profiling
  bool prof = false;
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> pool;
  double k_ms[HESS_K_COUNT];
  long long k_n[HESS_K_COUNT];
  double k_bytes[HESS_K_COUNT];
  double k_in_lds[HESS_K_COUNT];   // hess_profile_get_in_lds
};

// ---- shared by the host-side files (namespace hess) ----
namespace hess {


void set_err(hess_ctx* c, const char* fmt, ...);

// Environment switches.  The shipped library reads four: HESS_SHARE_DIR (and TMPDIR) -- where node-shared result buffers go
// when /dev/shm has no room --, HESS_COPY_TIMEOUT_S -- how long a result copy may take before the context is poisoned --
// and HESS_DELIVERY (mirror | dma | blit: how results reach the host, hess_copier.hip).  Everything else -- schedule A/B
// switches (HESS_NO_PAIR, HESS_NO_TOP_FUSION, HESS_NO_FIRST_FUSION, HESS_CHAIN_FROM, HESS_STREAM_ROWS, HESS_DESC_PARTS,
// HESS_DESC_XCD, HESS_MIRROR_MAX_BATCH, HESS_MIRROR_MAX_MB, HESS_NO_SIDE_UPLOAD, HESS_NO_PRIME_BATCH), test hooks
// (HESS_INITIAL_CAP, HESS_COPIER_FAULT, HESS_SHARE_FORCE_FILE), engine choices (HESS_COPIER, HESS_COPIER_ENGINE,
// HESS_UPLOAD_ENGINE) and HESS_CHAIN_STAMPS -- exists only in the DEVELOPER build (-DHESS_DEV_SWITCHES:
// hessgpu_amd/dev/libhessgpu.so, built beside the product by hessgpu_amd/build.py; hess_dev_switches() tells which one
// is loaded).  The tests that need a switch and tools/robustness.sh load that build.
#ifdef HESS_DEV_SWITCHES
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

#define HIP_TRY(c, expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      set_err(c, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? HESS_ERR_NOMEM : HESS_ERR_DEVICE;                \
    }                                                                                     \
  } while (0)


// hess_plan.hip
int ensure(hess_ctx* c, DevBuf& b, size_t bytes, bool pinned_host = false);
void release(DevBuf& b, bool pinned_host = false);
void default_params(hess_params* p);
void make_taps(const hess_params& p, float sigma, Taps* t);
void resolve(hess_ctx* c);
int fmt_channels(int format);
int plan(hess_ctx* c, int width, int height, int batch);
// hess_shared.hip
int ensure_shared(hess_ctx* c, DevBuf& b, size_t bytes, char which);
// hess_schedule.hip
void drain_profile(hess_ctx* c);
int enqueue(hess_ctx* c, const void* dev, int pitch, size_t image_stride, int batch, int format, int pixtype);
// hess_copier.hip
void stager_start(Stager& sg);
void stager_copy(Stager& sg, int k);
void stager_stop(Stager& sg);
bool copier_hsa_setup(hess_ctx* c);
int wait_copy_signal(hsa_signal_t sig, hsa_signal_value_t below, hsa_signal_value_t* last, bool injectable = true, bool* real = nullptr);
int copier_start(hess_ctx* c);
void copier_stop(hess_ctx* c);
void choose_delivery(hess_ctx* c, int batch);
int submit_impl(hess_ctx* c, const PendingRun& r);
int wait_impl(hess_ctx* c, const PendingRun& r);

}  // namespace hess
