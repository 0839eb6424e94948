// hess_pipeline.hip -- host side of the C ABI (include/hess_abi.h): context, HBM buffers,
// sigma schedule, octave geometry and the enqueue order of the hot path on one HIP stream.
//
// Replaces the per-image host orchestration of the reference: SiftPyramid::RunSIFT
// (SiftPyramid.cpp:53-198), PyramidCU::BuildPyramid / DetectKeypointsEX / GenerateFeatureList /
// SelectTopK / GetFeatureOrientations / ReshapeFeatureListCPU / GetFeatureDescriptors
// (PyramidCU.cpp:491-553,720-924,1283-1368,1486-1699,1815-1987) and the buffer management of
// CuTexImage (CuTexImage.cpp).  Differences by design: one batch of equally sized images per
// call, every stage enqueued without host synchronisation, a single device->host transfer of the
// counts followed by one of the results; feature-list order is deterministic (level, row, col).
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

using namespace hess;

namespace {

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
};

}  // namespace

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
  size_t mirror_max_bytes = (size_t)16 << 20;  // HESS_MIRROR_MAX_MB: result bytes (of the context's last batch) up to which a small batch uses the in-kernel mirror
  size_t last_result_bytes = 0;    // keypoints + descriptors the last batch delivered, and its size
  int last_result_batch = 0;
  std::atomic<bool> caller_waits{false};  // inside hess_run_* (submit + wait in one call; read by the copier thread, too)
  int delivery_pref = -1;          // HESS_DELIVERY=mirror|dma|blit (-1 = by batch size, see plan())
  int mirror_max_batch = 2;        // HESS_MIRROR_MAX_BATCH: batches up to this size use the in-kernel mirror
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
  int desc_xcd_block = 64;         // HESS_DESC_XCD: features per XCD block of the descriptor launch (0: plain order; A/B switch)
  Copier cp;
  Stager sg;
  // A DMA copy that did not complete in time (or that ROCr reported as failed) may still be in flight, or land later:
  // its targets -- the pinned result buffers, the pixel staging area -- must neither be reused nor freed.  The context
  // refuses every further run (HESS_ERR_DEVICE) and hess_destroy leaves those buffers and the signals alone.
  std::atomic<bool> poisoned{false};
  long long primed_shape = -1;     // (width, height, batch) of the dry batch hess_reserve has run (prime())
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

namespace {

void set_err(hess_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  try { c->err = buf; } catch (...) {}  // (nothing thrown crosses the C ABI; the message is then the previous one)
  if (c->p.verbose & 1) fprintf(stderr, "hessgpu: %s\n", buf);
}

#define HIP_TRY(c, expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      set_err(c, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? HESS_ERR_NOMEM : HESS_ERR_DEVICE;                \
    }                                                                                     \
  } while (0)

int ensure_shared(hess_ctx* c, DevBuf& b, size_t bytes, char which);

int ensure(hess_ctx* c, DevBuf& b, size_t bytes, bool pinned_host = false) {
  if (bytes <= b.bytes) return 0;
  if (pinned_host && c->share_dir && (&b == &c->h_keys || &b == &c->h_desc)) return ensure_shared(c, b, bytes, &b == &c->h_keys ? 'k' : 'd');
  // the new buffer first: when the allocation fails the old one is still there (a context survives a refused
  // hess_reserve)
  const size_t want = bytes + bytes / 8;  // slack so slightly larger inputs do not reallocate
  void* np = nullptr;
  if (pinned_host) HIP_TRY(c, hipHostMalloc(&np, want, hipHostMallocDefault));
  else HIP_TRY(c, hipMalloc(&np, want));
  if (b.p) { if (pinned_host) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
  b.p = np;
  b.bytes = want;
  return 0;
}

void release(DevBuf& b, bool pinned_host = false) {
  if (b.p && !b.shm.empty()) {
    (void)hipHostUnregister(b.p);
    (void)munmap(b.p, b.bytes);
    if (strchr(b.shm.c_str() + 1, '/')) (void)unlink(b.shm.c_str());  // a file (the fallback), else a shared memory object
    else (void)shm_unlink(b.shm.c_str());
    b.shm.clear();
  } else if (b.p) {
    if (pinned_host) (void)hipHostFree(b.p); else (void)hipFree(b.p);
  }
  b.p = nullptr;
  b.bytes = 0;
}

// A pinned result buffer of a context whose results are shared with other processes of the node (hess_share_results):
// a POSIX shared memory object, mapped and registered with the runtime, so that the copier's DMA copy (or the
// descriptor kernel's own stores) lands in memory the consumer process has mapped as well -- every GPU of a node
// delivers over its own host link and nothing is funnelled through one rank's.  `which` is 'k' or 'd'.
// Where /dev/shm has no room (containers often give it 64 MB) the buffer becomes a file under HESS_SHARE_DIR / TMPDIR /
// /tmp instead, mapped MAP_SHARED and registered the same way: page-cache pages, pinned by the registration -- the
// consumer maps the same pages.  Slower to set up, the same to use.  HESS_SHARE_FORCE_FILE=1 skips /dev/shm (tests).
int ensure_shared(hess_ctx* c, DevBuf& b, size_t bytes, char which) {
  if (bytes <= b.bytes) return 0;
  const long page = sysconf(_SC_PAGESIZE);
  size_t want = bytes + bytes / 4;  // grown by need: a quarter of slack so that batches of similar size do not reallocate
  want = (want + (size_t)page - 1) / (size_t)page * (size_t)page;
  uint32_t& gen = which == 'k' ? c->share_dir->gen_keys : c->share_dir->gen_desc;
  char name[256], path[1024];
  snprintf(name, sizeof(name), "/%s.%c%u", c->share.c_str(), which, gen + 1);
  // (posix_fallocate, not ftruncate: a full file system must fail here, not as a bus error at the first store)
  auto size_fd = [&](int fd) {
    int fe = posix_fallocate(fd, 0, (off_t)want);
    if (fe == EOPNOTSUPP || fe == EINVAL) fe = ftruncate(fd, (off_t)want) == 0 ? 0 : errno;
    return fe;
  };
  int fd = -1, why = 0;
  bool is_file = false;
  if (!getenv("HESS_SHARE_FORCE_FILE")) {
    (void)shm_unlink(name);  // a stale object of a dead job with the same name
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) why = errno;
    else if ((why = size_fd(fd)) != 0) { close(fd); shm_unlink(name); fd = -1; }
    if (fd >= 0) snprintf(path, sizeof(path), "/dev/shm%s", name);
  } else {
    why = ENOSPC;
  }
  if (fd < 0) {  // the fallback: a file
    const char* dir = getenv("HESS_SHARE_DIR");
    if (!dir || !dir[0]) dir = getenv("TMPDIR");
    if (!dir || !dir[0]) dir = "/tmp";
    // the directory as an absolute path (a reader process may have another working directory), and a word of warning
    // when it is not memory-backed: the DMA copies of every batch then dirty page-cache pages the kernel writes to disk
    char absdir[PATH_MAX];
    if (realpath(dir, absdir)) dir = absdir;
    struct statfs sfs;
    if (statfs(dir, &sfs) == 0 && sfs.f_type != 0x01021994 /* TMPFS_MAGIC */ && sfs.f_type != 0x858458f6 /* RAMFS_MAGIC */ &&
        (c->p.verbose & 1))
      fprintf(stderr, "hessgpu: shared result buffer %s goes to %s, which is not a tmpfs: expect disk write-back per batch\n", name + 1, dir);
    snprintf(path, sizeof(path), "%s%s", dir, name);
    (void)unlink(path);
    fd = open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
    int fe = fd < 0 ? errno : size_fd(fd);
    if (fe != 0) {
      if (fd >= 0) { close(fd); unlink(path); }
      set_err(c, "cannot place the shared result buffer %s (%zu bytes): /dev/shm: %s; %s: %s", name + 1, want, strerror(why), path, strerror(fe));
      return HESS_ERR_NOMEM;
    }
    is_file = true;
  }
  auto drop = [&]() { if (is_file) unlink(path); else shm_unlink(name); };
  void* np = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
  close(fd);
  if (np == MAP_FAILED) { set_err(c, "mmap(%s) failed: %s", path, strerror(errno)); drop(); return HESS_ERR_NOMEM; }
  void* dp = nullptr;
  hipError_t e = hipHostRegister(np, want, hipHostRegisterPortable | hipHostRegisterMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, np, 0);
  if (e != hipSuccess || dp != np) {  // (the kernels and the copier address the buffer by its host pointer)
    if (e == hipSuccess) (void)hipHostUnregister(np);
    else (void)hipGetLastError();
    set_err(c, "cannot register the shared result buffer %s with the runtime: %s", path,
            e != hipSuccess ? hipGetErrorString(e) : "device alias differs from the host address");
    munmap(np, want); drop();
    return e == hipErrorOutOfMemory ? HESS_ERR_NOMEM : HESS_ERR_DEVICE;
  }
  release(b, true);
  try {
    b.shm = is_file ? path : name;
  } catch (...) {  // (nothing thrown crosses the C ABI)
    (void)hipHostUnregister(np);
    munmap(np, want); drop();
    set_err(c, "out of memory");
    return HESS_ERR_NOMEM;
  }
  b.p = np; b.bytes = want;
  // size and path first, the generation number last: a reader that sees the new generation sees its buffer
  (which == 'k' ? c->share_dir->keys_bytes : c->share_dir->desc_bytes) = want;
  snprintf(which == 'k' ? c->share_dir->keys_path : c->share_dir->desc_path, sizeof(c->share_dir->keys_path), "%s", path);
  __sync_synchronize();
  gen++;
  __sync_synchronize();
  return 0;
}

// ---- parameters: GlobalUtil.cpp:51-144 defaults, SiftParam::ParseSiftParam SiftGPU.cpp:491-563 ----

void default_params(hess_params* p) {
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
  p->normalize = 1;
  p->truncate_method = HESS_TRUNC_HIGHEST_0;
  p->feature_count_threshold = -1;
  p->tex_max_dim = 3200;
  p->descriptor_order = HESS_DESC_ORDER_PIXEL;
}

// ProgramCU::CreateFilterKernel, ProgramCU.cu:423-453 (host arithmetic, libm expf).
void make_taps(const hess_params& p, float sigma, Taps* t) {
  int sz = (int)ceil(p.filter_width_factor * sigma - 0.5);
  int width = 2 * sz + 1;
  if (width > kMaxTaps) { sz = kMaxTaps >> 1; width = kMaxTaps; }
  else if (width < 5) { sz = 2; width = 5; }
  float rv = 1.0f / (sigma * sigma), v, ksum = 0;
  for (int i = -sz; i <= sz; ++i) {
    t->k[i + sz] = v = expf(-0.5f * i * i * rv);
    ksum += v;
  }
  rv = 1.0f / ksum;
  for (int i = 0; i < width; i++) t->k[i] *= rv;
  for (int i = width; i < kMaxTaps; i++) t->k[i] = 0.0f;
  t->fw = width;
}

void resolve(hess_ctx* c) {
  hess_params& p = c->p;
  if (p.dog_level_num == 0) p.dog_level_num = 3;
  if (p.sigma0 == 0.0f) p.sigma0 = 1.6f;
  if (p.sigman == 0.0f) p.sigman = 0.5f;
  if (p.filter_width_factor == 0.0f) p.filter_width_factor = 4.0f;
  if (p.orient_window_factor == 0.0f) p.orient_window_factor = 2.0f;
  if (p.orient_gaussian_factor == 0.0f) p.orient_gaussian_factor = 1.5f;
  if (p.desc_window_factor == 0.0f) p.desc_window_factor = 3.0f;
  if (p.tex_max_dim == 0) p.tex_max_dim = 3200;
  if (p.max_orientation < 1) p.max_orientation = 1;  // SiftGPU.cpp:1047
  if (p.max_orientation > 4) p.max_orientation = 4;
  Schedule& s = c->sch;
  s.dog = p.dog_level_num;
  s.level_max = s.dog + 1;
  s.level_num = s.level_max + 1;
  s.level_ds = s.dog;
  const float sigmak = powf(2.0f, 1.0f / p.dog_level_num);
  const float dsigma0 = p.sigma0 * sqrtf(sigmak * sigmak - 1.0f);
  for (int i = 1; i <= s.level_max; i++) {
    s.sigma[i - 1] = dsigma0 * powf(sigmak, (float)(i - 1));
    make_taps(p, s.sigma[i - 1], &s.taps[i]);
  }
  for (int l = 0; l <= s.level_max; l++) {
    s.level_sigma[l] = p.sigma0 * powf(2.0f, (float)l / (float)p.dog_level_num);
    const float ls = s.level_sigma[l] * 1.0f;  // octaveSigma = 1 (PyramidCU.cpp:1574-1585)
    const float n2 = ls * ls;                  // passed by DetectKeypointsEX
    s.norm[l] = n2 * n2;                       // squared again by ProgramCU::ComputeHessian (:592)
  }
  if (p.dog_threshold == 0.0f) p.dog_threshold = 0.02f / p.dog_level_num;
  if (p.edge_threshold == 0.0f) p.edge_threshold = 10.0f;
  s.sigma_step = powf(2.0f, 1.0f / p.dog_level_num);
  s.ln_sigma_step = (float)log((double)s.sigma_step);
}

float initial_smooth_sigma(const hess_ctx* c, int octave_min) {  // SiftGPU.cpp:482-489
  const float sa = c->p.sigma0 * powf(2.0f, 0.0f / (float)c->p.dog_level_num);
  const float sb = c->p.sigman / powf(2.0f, (float)octave_min);
  return (sa > sb + 0.001) ? sqrtf(sa * sa - sb * sb) : 0.0f;
}

int fmt_channels(int format) {
  switch (format) {
    case HESS_FMT_LUM: return 1;
    case HESS_FMT_LUM_ALPHA: return 2;
    case HESS_FMT_RGB: case HESS_FMT_BGR: return 3;
    case HESS_FMT_RGBA: case HESS_FMT_BGRA: return 4;
  }
  return 0;
}

// Geometry: SetImageData (GLTexImage.cpp:932-1033) + InitPyramid/ResizePyramid (PyramidCU.cpp:113-310).
int plan_inner(hess_ctx* c, int width, int height, int batch) {
  const hess_params& p = c->p;
  int ds = 0, ws = width, hs = height;
  if (p.first_octave > 0) { ds = p.first_octave; ws = width >> ds; hs = height >> ds; }
  else if (p.first_octave < 0) { ds = p.first_octave; ws = (width & ~3) << (-ds); hs = height << (-ds); }  // PyramidCU.cpp:120-138
  if (ws > p.tex_max_dim || hs > p.tex_max_dim) {
    if (!p.auto_downscale) {
      set_err(c, "image %dx%d exceeds max dimension %d (use -ads or -maxd)", ws, hs, p.tex_max_dim);
      return HESS_ERR_TOO_BIG;
    }
    // _octave_min++ until it fits (PyramidCU.cpp:154-166): an up-sampled first octave is up-sampled less, then not at
    // all, then decimated -- the same loop whatever the sign of the first octave
    do { ds++; ws >>= 1; hs >>= 1; } while (ws > p.tex_max_dim || hs > p.tex_max_dim);
  }
  ws &= ~3;  // TruncateWidthCU
  if (ws < 4 || hs < 1) { set_err(c, "image too small"); return HESS_ERR_ARG; }
  const bool same = c->planned && c->in_w == width && c->in_h == height && batch <= c->g.B &&
                    (int)(2 * c->user_keys.size() + 8) <= c->cap_sel && (int)(2 * c->user_keys.size() + 8) <= c->cap_feat;
  if (same) return 0;
  const int B = (c->planned && c->g.B > batch) ? c->g.B : batch;

  Geom g;
  memset(&g, 0, sizeof(g));
  const int input_sz = ws < hs ? ws : hs;
  int nmax = (int)floor(log((double)input_sz) / log(2.0)) - 3;  // PyramidCU.cpp:242
  if (nmax < 1) nmax = 1;
  if (nmax > kMaxOct) nmax = kMaxOct;
  g.noct = (p.octave_num >= 1 && p.octave_num < nmax) ? p.octave_num : nmax;
  g.dog = c->sch.dog;
  g.nlev = g.noct * g.dog;
  g.B = B;
  long long lvl = 0, gt = 0;
  int rows = 0, mw = 0;
  int w = ws, h = hs;
  for (int o = 0; o < g.noct; o++) {
    OctGeom& og = g.o[o];
    og.wa = ((w + 3) / 4) * 4;
    og.h = h;
    og.plane = og.wa * og.h;
    og.w64 = (og.wa + 63) / 64;
    og.lvl_off = lvl;
    og.got_off = gt;
    og.row_base = rows;
    og.mask_base = mw;
    og.tiles_x = (og.wa + 127) / 128;  // EX_TC columns per extrema tile (k_detect.hip)
    og.tile_base = g.ntiles;
    g.ntiles += og.tiles_x * ((og.h + 3) / 4);  // EX_TR rows per extrema tile (k_detect.hip)
    og.strips = (og.wa + kStreamPitch - 1) / kStreamPitch;
    lvl += (long long)c->sch.level_num * B * og.plane;
    gt += (long long)g.dog * B * og.plane;
    rows += g.dog * og.h;
    mw += g.dog * og.h * og.w64;
    w >>= 1;
    h >>= 1;
  }
  g.NR = rows;
  g.NM = mw;
  set_stream_rows(g, kStreamRows);

  c->use_topk = (p.truncate_method == HESS_TRUNC_TOPK && p.feature_count_threshold > 0);
  c->multi = (p.max_orientation > 1) && !p.fixed_orientation;  // SiftPyramid.cpp:140
  c->dim = p.compute_descriptors ? (p.half_sift ? 64 : 128) : 0;
  long long det_px = gt / B;  // detection pixels per image
  int cap_raw = (int)(det_px / 32 < 16384 ? 16384 : det_px / 32);
  if (c->cap_init > 0) cap_raw = c->cap_init;  // developer switch: start small so that the grow-and-re-run path is taken
  if (cap_raw < c->cap_raw) cap_raw = c->cap_raw;
  if (cap_raw < (int)(2 * c->user_keys.size() + 8)) cap_raw = (int)(2 * c->user_keys.size() + 8);
  int cap_sel = c->use_topk ? p.feature_count_threshold : cap_raw;
  if (cap_sel > cap_raw) cap_sel = cap_raw;
  int cap_feat = c->multi ? (c->use_topk ? 4 * cap_sel : cap_sel) : cap_sel;
  if (cap_feat < c->cap_feat) cap_feat = c->cap_feat;
  if (!c->user_keys.empty()) {  // a keypoint list bypasses top-K: every stage must hold 2*num+8 records
    const int need = (int)(2 * c->user_keys.size() + 8);
    if (cap_sel < need) cap_sel = need;
    if (cap_feat < need) cap_feat = need;
  }

  int rc;
  if ((rc = ensure(c, c->gauss, (size_t)lvl * 4))) return rc;
  if ((rc = ensure(c, c->deth, (size_t)lvl * 4))) return rc;
  if ((rc = ensure(c, c->got, (size_t)gt * 8))) return rc;
  {
    const int up = ds < 0 ? -ds : 0;  // the converted input is held at its own size; the up-sampled copy in `upsampled`
    if ((rc = ensure(c, c->input_f32, (size_t)B * (ws >> up) * (hs >> up) * 4))) return rc;
  }
  if (ds < 0 && (rc = ensure(c, c->upsampled, (size_t)B * ws * hs * 4))) return rc;
  {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_found = 256, o_cnt = o_found + up((size_t)3 * B * 4), o_hist = o_cnt + up((size_t)B * g.NR * 4);
    const size_t o_mask = o_hist + (c->use_topk ? up((size_t)B * kHistBins * 4) : 0);
    const size_t o_tk = o_mask + up((size_t)B * g.NM * 8);
    c->zeroed_used = o_tk + (c->use_topk ? up(topk_scratch_bytes(cap_raw, B, g.nlev)) : 0);
    if ((rc = ensure(c, c->zeroed, c->zeroed_used))) return rc;
    char* z = (char*)c->zeroed.p;
    c->overflow.p = z; c->rowcnt.p = z + o_cnt; c->hist.p = z + o_hist; c->rowmask.p = z + o_mask; c->tk.p = z + o_tk;
    c->found_count.p = z + o_found; c->place_ticket.p = z + o_found + (size_t)B * 4; c->place_flag.p = z + o_found + (size_t)2 * B * 4;
  }
  if ((rc = ensure(c, c->rowoff, (size_t)B * g.NR * 4))) return rc;
  if ((rc = ensure(c, c->level_count, (size_t)B * g.nlev * 4))) return rc;
  if ((rc = ensure(c, c->raw_total, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->sel_total, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->feat_total, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->feat_first, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->img_base, (size_t)(B + 1) * 4))) return rc;
  if ((rc = ensure(c, c->raw, (size_t)B * cap_raw * sizeof(RawKey)))) return rc;
  {  // the scan's unordered detections: every scan task's slots + the image's spill list (hess_dev.h, DetectStore);
     // tasks for the shortest segments a batch may be scanned with (enqueue(): batches of one or two images, HESS_STREAM_ROWS)
    int ntask = extrema_tasks(g);
    for (int rows : {kStreamRows / 2, c->stream_rows}) {
      if (rows <= 0) continue;
      Geom gr = g;
      set_stream_rows(gr, rows);
      ntask = std::max(ntask, extrema_tasks(gr));
    }
    c->found_tasks = ntask;
    if ((rc = ensure(c, c->found, (size_t)B * ((size_t)ntask * kDetectSlots + cap_raw) * sizeof(RawKey)))) return rc;
    if ((rc = ensure(c, c->task_count, (size_t)B * ntask * 4))) return rc;
  }
  if (c->use_topk) {
    if ((rc = ensure(c, c->sel, (size_t)B * cap_sel * sizeof(RawKey)))) return rc;
  }
  if ((rc = ensure(c, c->recs, (size_t)B * cap_sel * sizeof(FRec)))) return rc;
  if ((rc = ensure(c, c->ocount, (size_t)B * cap_sel * 4))) return rc;
  if ((rc = ensure(c, c->foffset, (size_t)B * cap_sel * 4))) return rc;
  if ((rc = ensure(c, c->fsrc, (size_t)B * cap_feat * 4))) return rc;
  if ((rc = ensure(c, c->keys, (size_t)B * cap_feat * sizeof(HostKeypoint)))) return rc;
  if (c->dim && (rc = ensure(c, c->desc, (size_t)B * cap_feat * c->dim * 4))) return rc;
  if ((rc = ensure(c, c->h_small, (size_t)(3 * B + 8) * 4, true))) return rc;
  {
    // The pinned result buffers hold the worst case B * cap_feat records up front while that stays moderate; beyond
    // it they grow on demand once the counts are known (wait_impl / the copier), and the in-kernel mirror is not used.
    const size_t host_bytes = (size_t)B * cap_feat * (sizeof(HostKeypoint) + (size_t)c->dim * 4);
    // Node-shared result buffers of a batch the copier delivers are sized by the batches seen (+ 25 %), not for the
    // worst case B * cap_feat: 79 MB per context, 3.8 - 4.4 GB of /dev/shm for a node's six or seven contexts x eight
    // ranks, where the results are 24 MB per context; the copier (or hess_wait) grows them under a new generation when a batch needs more.
    c->share_by_need = c->share_dir && B > c->mirror_max_batch && c->delivery_pref != kDeliverMirror;
    c->host_fits = !c->share_by_need && host_bytes <= ((size_t)512 << 20);
    if (c->host_fits) {
      if ((rc = ensure(c, c->h_keys, (size_t)B * cap_feat * sizeof(HostKeypoint), true))) return rc;
      if (c->dim && (rc = ensure(c, c->h_desc, (size_t)B * cap_feat * c->dim * 4, true))) return rc;
    }
  }

  c->g = g;
  c->ds = ds;
  c->img_w = ws;
  c->img_h = hs;
  c->in_w = width;
  c->in_h = height;
  c->cap_raw = cap_raw;
  c->cap_sel = cap_sel;
  c->cap_feat = cap_feat;
  c->planned = true;
  const float s0 = initial_smooth_sigma(c, ds);
  c->has_taps0 = s0 > 0.0f;
  if (c->has_taps0) make_taps(p, s0, &c->taps0);
  return 0;
}

// A plan that fails half way (an allocation was refused) leaves buffers of mixed sizes behind: the next run plans
// again from scratch (buffers that are large enough are kept), so the context stays usable.
int plan(hess_ctx* c, int width, int height, int batch) {
  const int rc = plan_inner(c, width, height, batch);
  if (rc) {
    c->planned = false;
    (void)hipGetLastError();  // the refused allocation must not be reported by the next call's error check
  }
  return rc;
}

// ---- profiling helpers ----
hipEvent_t get_event(hess_ctx* c) {
  if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) { set_err(c, "hipEventCreate failed: profiling disabled"); c->prof = false; return nullptr; }
  return e;
}
struct ProfScope {
  hess_ctx* c;
  EventPair ep;
  bool on;
  ProfScope(hess_ctx* ctx, int kernel, double bytes, int kernel2 = -1, double in_lds = 0.0) : c(ctx), on(ctx->prof) {
    if (!on) return;
    ep.a = get_event(c); ep.b = get_event(c); ep.kernel = kernel; ep.bytes = bytes; ep.kernel2 = kernel2; ep.in_lds = in_lds;
    if (!ep.a || !ep.b) {  // event creation failed: no record for this launch
      if (ep.a) c->pool.push_back(ep.a);
      if (ep.b) c->pool.push_back(ep.b);
      on = false;
      return;
    }
    if (hipEventRecord(ep.a, c->st) != hipSuccess) { c->pool.push_back(ep.a); c->pool.push_back(ep.b); on = false; }
  }
  ~ProfScope() {
    if (!on) return;
    if (hipEventRecord(ep.b, c->st) != hipSuccess) { c->pool.push_back(ep.a); c->pool.push_back(ep.b); return; }
    c->pending.push_back(ep);
  }
};
void drain_profile(hess_ctx* c) {
  for (auto& ep : c->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
      c->k_ms[ep.kernel] += ms;
      c->k_n[ep.kernel] += 1;
      c->k_bytes[ep.kernel] += ep.bytes;
      c->k_in_lds[ep.kernel] += ep.in_lds;
      if (ep.kernel2 >= 0) {
        c->k_ms[ep.kernel2] += ms; c->k_n[ep.kernel2] += 1; c->k_bytes[ep.kernel2] += ep.bytes; c->k_in_lds[ep.kernel2] += ep.in_lds;
      }
    }
    c->pool.push_back(ep.a);
    c->pool.push_back(ep.b);
  }
  c->pending.clear();
}

// 2^_octave_min: scale of the first octave relative to the input (PyramidCU.cpp:566-569,746-748,1054-1057).
static inline float first_octave_sigma(const hess_ctx* c) {
  return c->ds > 0 ? (float)(1 << c->ds) : (c->ds < 0 ? 1.0f / (float)(1 << (-c->ds)) : 1.0f);
}

int enqueue_user(hess_ctx* c);

// Enqueue the whole path for `batch` images whose pixels are at device address `dev`.
int enqueue(hess_ctx* c, const void* dev, int pitch, size_t image_stride, int batch, int format, int pixtype) {
  const hess_params& p = c->p;
  const Schedule& s = c->sch;
  const Geom& g = c->g;
  hipStream_t st = c->st;
  float* gauss = (float*)c->gauss.p;
  float* deth = (float*)c->deth.p;
  float* got = (float*)c->got.p;
  auto plane_ptr = [&](float* base, int o, int l) { return base + g.o[o].lvl_off + (long long)l * g.B * g.o[o].plane; };

  // Stage timers (SiftGPU::_timing[2..10]) only when asked for: hess_params.verbose bit 1 (the reference's _timingS,
  // SiftGPU.cpp:433-464: its stage times, too, are only meaningful when it synchronises at stage ends) or the bench's
  // per-kernel profile.  An event record between two kernels leaves the stream idle for about 6 us.
  c->stage_events = (c->p.verbose & 2) != 0;
  HIP_TRY(c, hipEventRecord(c->ev[0], st));
  const bool user_mode = !c->user_keys.empty();
  DetectParams dp;
  dp.thr = p.dog_threshold;
  dp.thr0 = (p.subpixel ? 0.8f : 1.0f) * p.dog_threshold;                            // ProgramCU.cu:897
  dp.edge = (p.edge_threshold + 1) * (p.edge_threshold + 1) / p.edge_threshold;      // ProgramCU.cu:913
  dp.subpixel = p.subpixel;
  Geom gx = g;  // the extrema scan's segment length
  if (c->stream_rows > 0) set_stream_rows(gx, c->stream_rows);      // HESS_STREAM_ROWS (A/B switch; a multiple of 3)
  else if (batch <= 2) set_stream_rows(gx, kStreamRows / 2);        // one or two images: shorter segments, twice the wavefronts
  if (!(user_mode && c->user_on_current)) {  // SIFT_SKIP_FILTERING: the resident pyramid is reused
  // ---- input + pyramid (BuildPyramid, PyramidCU.cpp:1486-1558) ----
  const bool direct_u8 = (format == HESS_FMT_LUM && pixtype == HESS_PIX_U8 && c->ds == 0 && c->has_taps0 &&
                          (pitch % 4) == 0 && (image_stride % 4) == 0 && ((uintptr_t)dev % 4) == 0);
  const float* src_f = nullptr;
  if (!direct_u8) {
    ProfScope ps(c, HESS_K_INPUT, (double)batch * c->img_w * c->img_h * (4.0 + fmt_channels(format)));
    const int up = c->ds < 0 ? -c->ds : 0;  // up-sampled first octave: convert at full size, then SampleImageU
    launch_convert(st, dev, format, pixtype, pitch, (long long)image_stride, up ? 0 : c->ds, (float*)c->input_f32.p,
                   c->img_w >> up, c->img_h >> up, batch);
    src_f = (const float*)c->input_f32.p;
    if (up) {  // PyramidCU.cpp:1521-1522
      launch_upsample(st, src_f, c->img_w >> up, c->img_h >> up, up, (float*)c->upsampled.p, batch);
      src_f = (const float*)c->upsampled.p;
    }
  }
  // The launch that produces the down-sampling level also writes level 0 of the next octave (its even rows and
  // columns): no decimation launches.  (A down-sampling level 0 is nobody's product: separate kernel then.)
  const bool fused_decim = s.level_ds >= 1 && s.level_ds <= s.level_max;
  // det-H of the top level by the launch that produces it (k_gauss.hip, TOP tiles); HESS_NO_TOP_FUSION=1: the round-4
  // form (the level is stored, hessian_rows4_kernel reads it back) for A/B runs
  const bool top_fused = s.level_max == g.dog + 1 && s.level_max >= 1 && !c->no_top_fusion;
  c->zero_filled = false;
  // Level l of octave o from level l-1; the same launch emits det-H (+ gradient/theta) of level l-1 from the source
  // window it stages: 8 B R+W for the blur, 4 B (+8 B) W for the fused planes (+ 4 B per pixel of the next octave's
  // level 0 when it is the down-sampling level).
  auto level_job = [&](int o, int l) {
    const OctGeom& og = g.o[o];
    const bool src_got = (l - 1 >= 1 && l - 1 <= g.dog);
    const bool decim = fused_decim && l == s.level_ds && o + 1 < g.noct;
    GaussJob j;
    j.src = plane_ptr(gauss, o, l - 1); j.dst = plane_ptr(gauss, o, l); j.wa = og.wa; j.h = og.h; j.taps = s.taps[l];
    j.deth_src = plane_ptr(deth, o, l - 1);
    j.got_src = src_got ? got + 2 * (og.got_off + (long long)(l - 2) * g.B * og.plane) : nullptr;
    j.norm_src = s.norm[l - 1];
    j.decim_dst = decim ? plane_ptr(gauss, o + 1, 0) : nullptr;
    j.decim_w = decim ? g.o[o + 1].wa : 0; j.decim_h = decim ? g.o[o + 1].h : 0;
    if (top_fused && l == s.level_max) {
      // The octave's top level is nobody's source: its det-H comes out of the launch that produces it (from the output
      // tile in LDS) and the level itself is not written to HBM -- unless the parity tests ask (hess_debug_keep_levels).
      j.deth_dst = plane_ptr(deth, o, l);
      j.norm_dst = s.norm[l];
      if (!c->keep_levels) j.dst = nullptr;
      if (o == 0 && !user_mode) {  // octave 0's launch also clears what the detection stages expect zeroed
        j.zero = c->zeroed.p;
        j.zero_bytes = c->zeroed_used;
      }
    }
    return j;
  };
  // (a fused top level reads its source and writes its own det-H instead of the level: the same 8 bytes)
  auto level_bytes = [&](int o, int l) {
    const OctGeom& og = g.o[o];
    const bool src_got = (l - 1 >= 1 && l - 1 <= g.dog);
    const bool decim = fused_decim && l == s.level_ds && o + 1 < g.noct;
    return (double)batch * og.plane * (8.0 + 4.0 + (src_got ? 8.0 : 0.0)) + (decim ? (double)batch * g.o[o + 1].plane * 4.0 : 0.0);
  };
  // (SURVEY 8d books every array of the reference's layout written once and read once; a level this build keeps in LDS
  // -- the octave's top level, level 0 of octave 0 -- is 8 bytes per pixel of that layout which no launch here moves)
  auto level_in_lds = [&](int o, int l) {
    return (top_fused && l == s.level_max && !c->keep_levels) ? (double)batch * g.o[o].plane * 8.0 : 0.0;
  };
  auto launch_level = [&](const GaussJob& j) {
    if (j.zero) c->zero_filled = true;
    launch_gauss_job(st, j, batch);
  };
  // T(o, l) = 3o + l is the earliest step of level l of octave o (level 0 of octave o+1 is the decimated level_ds of
  // octave o): the top level of an octave and level 1 of the next are due together and independent, so they share a
  // launch (launch_gauss_pair) -- one launch fewer per octave in the dependent chain.
  const bool pair_levels = fused_decim && s.level_ds < s.level_max && s.level_max >= 2 && !c->no_pair;
  // A single image (or two): octaves from 1 on get levels 1..level_ds -- what the next octave waits for -- from ONE
  // launch each (gauss_chain_kernel: 32x32 tiles computed in LDS on a shrinking halo): below 960x540 a level launch is a
  // few dozen workgroups that mostly wait, and the eighteen of them for octaves 1-6 of a 1080p image were two thirds of
  // its pyramid's time.  The top levels (nobody's input) follow in one launch for all octaves, together with det-H /
  // gradient of the chained octaves' levels 0..level_ds-1.  Same box, one 1080p image, device-resident: 0.400 -> 0.334 ms.
  // NOT for larger batches: a batch of 8 is 2 % faster on one stream with octaves >= 2 chained, but six pipelined
  // contexts lose 2 - 4 % (15.5 - 15.6 against 16.1 - 16.2 Gpix/s, same call; 15.8 - 16.0 with octaves >= 3) -- the chain
  // trades dependent launches for redundant arithmetic in 1024-thread workgroups that wait at barriers, which is what
  // an idle device wants and a saturated one does not; and not for the large octaves of a large image (a 4096^2 image's
  // octave 1 is 4 096 such workgroups: configs[4] 2.19 against 2.12 ms).  Needs the default schedule's tap counts.
  // HESS_CHAIN_FROM=n forces the first chained octave (99: never).
  int chain_from = g.noct;
  if (fused_decim && s.level_max == s.level_ds + 1 && gauss_chain_available(s.taps, s.level_ds)) {
    chain_from = c->chain_from;
    if (chain_from <= 0) {  // by size: the first octave (>= 1) whose planes of the whole batch are at most two 960x540 planes
      chain_from = g.noct;
      // (a PAIR of images handed over by hess_submit_* -- a caller who pipelines -- gets the level-by-level launches and
      //  the copier's delivery like a larger batch: 17.0 - 17.3 against 12.3 - 12.6 Gpix/s for six pipelined contexts)
      if (batch == 1 || (batch == 2 && c->caller_waits))
        for (int o = g.noct - 1; o >= 1 && (long long)batch * g.o[o].plane <= 2LL * 960 * 540; o--) chain_from = o;
    }
    if (chain_from > g.noct) chain_from = g.noct;
  }
  const bool chained = chain_from < g.noct;
  GaussJob top_jobs[kMaxOct];  // the top levels of the octaves that do not ride with the next octave's level 1
  int ntop = 0;
  double top_bytes = 0.0, top_in_lds = 0.0;
  int deferred_o = -1;
  for (int o = 0; o < g.noct; o++) {
    const OctGeom& og = g.o[o];
    if (o >= chain_from && o >= 1) {  // (its level 0 is the decimated level_ds of octave o-1, written by that launch)
      ChainJob cj;
      double bytes = 0.0;
      cj.src0 = plane_ptr(gauss, o, 0);
      for (int l = 0; l <= s.level_ds; l++) {
        cj.dst[l] = plane_ptr(gauss, o, l);
        cj.taps[l] = s.taps[l];
        // (the bytes of the fused planes are booked here although hessian_low_levels writes them: per step the sums agree)
        if (l >= 1) bytes += level_bytes(o, l);
      }
      cj.nlevels = s.level_ds; cj.wa = og.wa; cj.h = og.h;
      const bool decim = o + 1 < g.noct;
      cj.decim_dst = decim ? plane_ptr(gauss, o + 1, 0) : nullptr;
      cj.decim_w = decim ? g.o[o + 1].wa : 0; cj.decim_h = decim ? g.o[o + 1].h : 0;
      {
        ProfScope ps(c, HESS_K_GAUSS, bytes);
        if (!launch_gauss_chain(st, cj, batch)) { set_err(c, "level-chain launch refused"); return HESS_ERR_DEVICE; }
      }
      top_jobs[ntop++] = level_job(o, s.level_max);
      top_bytes += level_bytes(o, s.level_max);
      top_in_lds += level_in_lds(o, s.level_max);
      continue;
    }
    bool first_fused = false;  // levels 0 and 1 of octave 0 came out of one launch (level 0 never written)
    if (o == 0 && direct_u8 && c->has_taps0 && !c->no_first_fusion && s.level_max >= 2 && s.level_ds != 1 && chain_from != 0) {
      // u8 pixels -> level 0 (LDS) -> level 1, det-H of level 0: the level-0 plane is nobody's input but level 1's
      const GaussJob j1 = level_job(0, 1);
      ProfScope ps(c, HESS_K_GAUSS, (double)batch * og.plane * (1.0 + 4.0 + 4.0), HESS_K_GAUSS_OCT0,
                   c->keep_levels ? 0.0 : (double)batch * og.plane * 8.0);
      first_fused = launch_gauss_first(st, (const uint8_t*)dev, pitch, (long long)image_stride, c->taps0, j1,
                                       c->keep_levels ? plane_ptr(gauss, 0, 0) : nullptr, batch);
    }
    if (o == 0) c->level0_in_lds = first_fused && !c->keep_levels;
    if (o == 0 && first_fused) {
      // (nothing: level 1 exists, the loop below starts at level 2)
    } else if (o == 0) {
      if (c->has_taps0) {
        ProfScope ps(c, HESS_K_GAUSS, (double)batch * og.plane * (direct_u8 ? 5.0 : 8.0), HESS_K_GAUSS_OCT0);
        if (direct_u8)
          launch_gauss(st, nullptr, (const uint8_t*)dev, pitch, (long long)image_stride, plane_ptr(gauss, 0, 0),
                       og.wa, og.h, batch, c->taps0);
        else
          launch_gauss(st, src_f, nullptr, og.wa, og.plane, plane_ptr(gauss, 0, 0), og.wa, og.h, batch, c->taps0);
      } else {
        HIP_TRY(c, hipMemcpyAsync(plane_ptr(gauss, 0, 0), src_f, (size_t)batch * og.plane * 4, hipMemcpyDeviceToDevice, st));
      }
    } else if (!fused_decim) {
      ProfScope ps(c, HESS_K_DOWNSAMPLE, (double)batch * og.plane * 8.0);
      launch_downsample(st, plane_ptr(gauss, o - 1, s.level_ds), g.o[o - 1].wa, g.o[o - 1].plane,
                        plane_ptr(gauss, o, 0), og.wa, og.h, batch);
    }
    for (int l = first_fused ? 2 : 1; l <= s.level_max; l++) {
      if (l == 1 && deferred_o >= 0) {  // the previous octave's top level rides with this octave's level 1
        const int top_o = deferred_o;
        {
          const GaussJob ja = level_job(top_o, s.level_max), jb = level_job(o, 1);
          ProfScope ps(c, HESS_K_GAUSS, level_bytes(top_o, s.level_max) + level_bytes(o, 1), top_o == 0 ? HESS_K_GAUSS_OCT0 : -1,
                       level_in_lds(top_o, s.level_max));
          if (launch_gauss_pair(st, ja, jb, batch)) c->zero_filled = c->zero_filled || ja.zero != nullptr;
          else { launch_level(ja); launch_level(jb); }
        }
        deferred_o = -1;
        continue;
      }
      if (l == s.level_max && chained && o + 1 >= chain_from) {  // with the chained octaves' top levels, after the chain
        top_jobs[ntop++] = level_job(o, l);
        top_bytes += level_bytes(o, l);
        top_in_lds += level_in_lds(o, l);
        continue;
      }
      if (l == s.level_max && pair_levels && o + 1 < g.noct) { deferred_o = o; continue; }
      ProfScope ps(c, HESS_K_GAUSS, level_bytes(o, l), o == 0 ? HESS_K_GAUSS_OCT0 : -1, level_in_lds(o, l));
      launch_level(level_job(o, l));
    }
  }
  if (deferred_o >= 0) {  // (cannot happen: the last octave never defers)
    ProfScope ps(c, HESS_K_GAUSS, level_bytes(deferred_o, s.level_max), -1, level_in_lds(deferred_o, s.level_max));
    launch_level(level_job(deferred_o, s.level_max));
  }
  if (ntop) {  // the top levels left over by the chain, one launch
    ProfScope ps(c, HESS_K_GAUSS, top_bytes, -1, top_in_lds);
    // (+ det-H / gradient of levels 0..level_ds-1 of the chain-launched octaves, from HBM: hessian_low_levels)
    LowLevels low{&g, gauss, deth, got, s.norm, chain_from, chained ? s.level_ds : 0};
    if (launch_gauss_multi(st, top_jobs, ntop, batch, &low)) {
      for (int k = 0; k < ntop; k++) c->zero_filled = c->zero_filled || top_jobs[k].zero != nullptr;
    } else {
      for (int k = 0; k < ntop; k++) launch_level(top_jobs[k]);
      if (chained)  // (not reached with the schedules the chain is instantiated for)
        for (int o = chain_from; o < g.noct; o++) launch_hessian(st, g, o, gauss, deth, got, s.norm, batch, 0, s.level_ds - 1);
    }
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[1], st));
  // ---- det-Hessian + gradient (DetectKeypointsEX part 1, PyramidCU.cpp:1576-1591) ----
  // (levels 0 .. level_max-1: by the launch that reads the level as its source; the top level: by the launch that
  // produces it -- no launch here with the reference's level layout)
  if (!top_fused) {  // the octaves' top levels from HBM, one launch for all of them
    double px = 0;
    for (int o = 0; o < g.noct; o++) px += g.o[o].plane;
    ProfScope ps(c, HESS_K_HESSIAN, (double)batch * px * 8.0);
    if (s.level_max >= 1 && s.level_max <= g.dog) {  // (never with the reference's level layout: level_max = dog + 1)
      for (int o = 0; o < g.noct; o++) launch_hessian(st, g, o, gauss, deth, got, s.norm, batch, s.level_max, s.level_max);
    } else {
      // this launch also clears the buffers of the detection stages (no fill launch of its own in the chain)
      launch_hessian_level(st, g, gauss, deth, s.level_max, s.norm[s.level_max], batch, user_mode ? nullptr : c->zeroed.p,
                           c->zeroed_used);
      c->zero_filled = !user_mode;
    }
  }
  }  // !(user_mode && on_current)
  if (user_mode) return enqueue_user(c);
  // ---- extrema + ordered list (DetectKeypointsEX part 2 + GenerateFeatureList) ----
  LimitParams lp;
  lp.method = p.truncate_method;
  lp.threshold = p.feature_count_threshold;
  if (!c->zero_filled)  // overflow flags, row counts, histogram, masks (normally cleared by octave 0's top-level launch)
    HIP_TRY(c, hipMemsetAsync(c->zeroed.p, 0, c->zeroed_used, st));
  DetectStore dstore;
  dstore.found = (RawKey*)c->found.p;
  dstore.ntask = extrema_tasks(gx);
  dstore.stride = (long long)c->found_tasks * kDetectSlots + c->cap_raw;
  dstore.task_count = (int*)c->task_count.p;
  dstore.spill_count = (int*)c->found_count.p;
  dstore.cap_spill = c->cap_raw;
  dstore.hist = c->use_topk ? (unsigned*)c->hist.p : nullptr;
  if (dstore.ntask > c->found_tasks) { set_err(c, "detection store laid out for %d scan tasks, the batch has %d", c->found_tasks, dstore.ntask); return HESS_ERR_ARG; }
  {
    // algorithmic bytes: every det-H level of every octave is read once (4 B R per level-pixel)
    double det_bytes = 0;
    for (int o = 0; o < g.noct; o++) det_bytes += 4.0 * s.level_num * g.o[o].plane;
    ProfScope ps(c, HESS_K_EXTREMA, det_bytes * batch);
    launch_extrema_mark(st, gx, dp, gauss, deth, (uint64_t*)c->rowmask.p, (int*)c->rowcnt.p, dstore, batch);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[2], st));
  {
    ProfScope ps(c, HESS_K_EXTREMA, 0.0);
    launch_extrema_place(st, g, lp, dstore, (const uint64_t*)c->rowmask.p, (const int*)c->rowcnt.p, (int*)c->rowoff.p,
                         (int*)c->level_count.p, (int*)c->raw_total.p, (int*)c->overflow.p, (int*)c->place_ticket.p,
                         (int*)c->place_flag.p, (RawKey*)c->raw.p, c->cap_raw, batch);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[3], st));
  // ---- top-K (LimitFeatureCount(0) -> SelectTopK) ----
  const RawKey* list = (const RawKey*)c->raw.p;
  const int* list_total = (const int*)c->raw_total.p;
  int cap_list = c->cap_raw;
  if (c->use_topk) {
    ProfScope ps(c, HESS_K_TOPK, 0.0);
    launch_topk(st, g, p.feature_count_threshold, (const RawKey*)c->raw.p, (const int*)c->raw_total.p, c->cap_raw,
                (unsigned*)c->hist.p, (RawKey*)c->sel.p, (int*)c->sel_total.p, c->cap_sel, batch, c->tk.p, (int*)c->overflow.p);
    list = (const RawKey*)c->sel.p;
    list_total = (const int*)c->sel_total.p;
    cap_list = c->cap_sel;
  }
  c->d_list = list;
  c->d_list_total = list_total;
  c->cap_list = cap_list;
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[4], st));
  // ---- orientation (GetFeatureOrientations) ----
  OrientParams op;
  op.gaussian_factor = p.orient_gaussian_factor;
  op.sample_factor = p.orient_gaussian_factor * p.orient_window_factor;  // ProgramCU.cu:1638
  op.ln_sigma_step = s.ln_sigma_step;
  op.num_orientation = p.fixed_orientation ? 0 : p.max_orientation;      // ProgramCU.cu:1639
  op.subpixel = p.subpixel;
  op.half_sift = p.half_sift;
  op.existing = 0;
  for (int l = 0; l < kMaxLev; l++) op.level_sigma[l] = l <= s.level_max ? s.level_sigma[l] : 0.0f;
  {
    ProfScope ps(c, HESS_K_ORIENT, 0.0);
    launch_orientation(st, g, op, list, list_total, cap_list, got, (FRec*)c->recs.p, (int*)c->ocount.p, batch);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[5], st));
  // ---- multi-orientation expansion (ReshapeFeatureListCPU) ----
  launch_feature_scan(st, g, lp, c->multi ? 1 : 0, list, list_total, cap_list, (const int*)c->ocount.p,
                      (int*)c->foffset.p, (int*)c->fsrc.p, (int*)c->feat_total.p, (int*)c->feat_first.p, c->cap_feat,
                      (int*)c->overflow.p, (int*)c->img_base.p, (int*)c->h_small.p, batch);
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[6], st));
  // ---- descriptors (GetFeatureDescriptors) ----
  DescParams dsp;
  dsp.window_factor = p.desc_window_factor;
  dsp.half_sift = p.half_sift;
  dsp.normalize = p.normalize;
  dsp.multi = c->multi ? 1 : 0;
  dsp.lowe_origin = p.lowe_origin;
  dsp.octave_sigma = first_octave_sigma(c);  // PyramidCU.cpp:746-748
  dsp.dog = g.dog;
  dsp.dynamic_indexing = p.dynamic_indexing ? 1 : 0;
  dsp.hkeys = c->host_direct ? (HostKeypoint*)c->h_keys.p : nullptr;
  dsp.hdesc = (c->host_direct && c->dim) ? (float*)c->h_desc.p : nullptr;
  dsp.first_image = 0;
  dsp.part = 0; dsp.part_den = 1;
  dsp.xcd_block = c->desc_xcd_block;
  dsp.sequential = p.descriptor_order == HESS_DESC_ORDER_SEQUENTIAL;
  // the pixel order's fixed point assumes luminance in [0, 1] (8- and 16-bit inputs); float pixels are taken as they are
  // and keep the interleaved order (the test oracle applies the same rule)
  dsp.pixel = p.descriptor_order == HESS_DESC_ORDER_PIXEL && pixtype != HESS_PIX_F32;
  // Delivered by the copier thread, a batch of four or more images gets its descriptors in two launches (the images
  // are independent and packed back to back): the first half's results cross the host link while the second half is
  // computed -- half of the transfer (0.53 ms for eight 1080p images) leaves the batch's critical path.  Four groups
  // shorten a lone batch a little more (1.75 / 1.58 / 1.53 ms for 1 / 2 / 4) but cost the pipelined rate 1 %:
  // HESS_DESC_PARTS=n overrides (1: one launch, up to kMaxParts).
  // ONE image delivered by the copier thread (a large one: choose_delivery) gets its descriptors in four launches over
  // quarters of its feature list, for the same reason (a 4096^2 image with 102 k half descriptors: 28 MB, 0.58 ms on the
  // link; 2.09 -> 1.7 ms per image on one context).
  {
    int want = batch >= 4 ? 2 : 1;
    c->part_features = false;
    if (batch == 1 && c->delivery == kDeliverDma) { want = Copier::kMaxParts; c->part_features = true; }
    if (c->desc_parts > 0) want = std::max(1, std::min<int>(Copier::kMaxParts, c->part_features ? c->desc_parts : std::min(batch, c->desc_parts)));
    if (c->delivery != kDeliverDma || !c->cp.ev_part[0]) want = 1;
    if (want == 1) c->part_features = false;
    c->nparts = want;
    for (int k = 0; k < want; k++) c->part_end[k] = c->part_features ? 1 : (int)((long long)batch * (k + 1) / want);
  }
  {
    int first = 0;
    for (int k = 0; k < c->nparts; k++) {
      ProfScope ps(c, HESS_K_DESCRIPTOR, 0.0);  // (per launch, so that the counts agree with a kernel trace)
      dsp.first_image = first;
      dsp.part = c->part_features ? k : 0;
      dsp.part_den = c->part_features ? c->nparts : 1;
      launch_descriptor(st, g, dsp, list, cap_list, (const FRec*)c->recs.p, (const int*)c->fsrc.p,
                        (const int*)c->feat_total.p, (const int*)c->feat_first.p, (const int*)c->img_base.p, got,
                        (HostKeypoint*)c->keys.p, c->dim ? (float*)c->desc.p : nullptr, c->cap_feat, c->part_end[k] - first,
                        c->seen_features);
      if (k < c->nparts - 1) HIP_TRY(c, hipEventRecord(c->cp.ev_part[k], st));
      if (!c->part_features) first = c->part_end[k];
    }
  }
  HIP_TRY(c, hipEventRecord(c->ev[7], st));
  return 0;
}

// FLOAT_TO_FIXED_POINT (config.h:73-74), host version.
static inline int float_to_fixed_host(float v, int n) {
  return (int)((double)(v * (float)(1 << n)) + ((v >= 0.0) ? 0.5 : -0.5));
}

// User-supplied keypoints (PyramidCU::GenerateFeatureListTex, PyramidCU.cpp:555-718): bin the keys to
// levels by scale, pack fixed-point records on the host, upload, strongest orientation on the device
// unless supplied, descriptors.  One image.
int enqueue_user(hess_ctx* c) {
  const hess_params& p = c->p;
  const Schedule& s = c->sch;
  const Geom& g = c->g;
  hipStream_t st = c->st;
  const int num = (int)c->user_keys.size();
  const double twopi = 2.0 * 3.14159265358979323846;
  const float sigma_half_step = powf(2.0f, 0.5f / g.dog);
  float octave_sigma = first_octave_sigma(c);
  const float offset = p.lowe_origin ? 0.0f : 0.5f;
  std::vector<RawKey> hl;
  std::vector<FRec> hr;
  c->user_kindex.clear();
  const size_t cap = 2 * (size_t)num + 8;
  for (int octave = 0; octave < g.noct; octave++, octave_sigma *= 2.0f) {
    for (int level = 1; level <= g.dog; level++) {
      const float level_sigma = s.level_sigma[level] * octave_sigma;
      const float sigma_min = level_sigma / sigma_half_step;
      const float sigma_max = level_sigma * sigma_half_step;
      for (int k = 0; k < num && hl.size() < cap; k++) {
        const hess_keypoint& key = c->user_keys[k];
        float sigmak = key.s;
        if ((int)c->user_levels.size() == num && c->user_levels[k] >= 0) {  // parity hook: level given, not derived
          if (c->user_levels[k] != octave * g.dog + (level - 1)) continue;
          sigmak = level_sigma;
        }
        if (((sigmak >= sigma_min) && (sigmak < sigma_max)) || ((sigmak < sigma_min) && (octave == 0) && (level == 1)) ||
            ((sigmak > sigma_max) && (octave == g.noct - 1) && (level == g.dog))) {
          const float fX = (key.x - offset) / octave_sigma + 0.5f;
          const float fY = (key.y - offset) / octave_sigma + 0.5f;
          const float fScale = key.s / octave_sigma;
          const float fOrientation = (float)fmod(twopi - key.o, twopi);
          FRec r;
          r.x = (uint32_t)float_to_fixed_host(fX, 10) & 0x00FFFFFFu;
          r.y = (uint32_t)float_to_fixed_host(fY, 10) & 0x00FFFFFFu;
          r.z = (uint32_t)float_to_fixed_host(fScale, 8) & 0x0000FFFFu;
          memcpy(&r.w, &fOrientation, 4);
          RawKey rk;
          memset(&rk, 0, sizeof(rk));
          rk.level_index = octave * g.dog + (level - 1);
          hl.push_back(rk);
          hr.push_back(r);
          c->user_kindex.push_back(k);
        }
      }
    }
  }
  const int n = (int)hl.size();
  if (n > c->cap_raw || n > c->cap_sel || n > c->cap_feat) {
    set_err(c, "keypoint list (%d) exceeds the reserved feature storage", n);
    return HESS_ERR_NOMEM;  // plan() sizes storage for 2*num+8 when a list is set
  }
  int* hs = (int*)c->h_small.p;
  hs[3 * g.B + 4] = n;
  HIP_TRY(c, hipMemsetAsync(c->overflow.p, 0, 64, st));  // overflow words + feature_scan_kernel's arrival counter
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->raw.p, hl.data(), (size_t)n * sizeof(RawKey), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->recs.p, hr.data(), (size_t)n * sizeof(FRec), hipMemcpyHostToDevice, st));
  }
  HIP_TRY(c, hipMemcpyAsync(c->raw_total.p, hs + 3 * g.B + 4, 4, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipStreamSynchronize(st));  // hl/hr are pageable host vectors: finish before they go away
  const RawKey* list = (const RawKey*)c->raw.p;
  const int* list_total = (const int*)c->raw_total.p;
  c->d_list = list;
  c->d_list_total = list_total;
  c->cap_list = c->cap_raw;
  if (c->stage_events) for (int e = 1; e <= 4; e++) HIP_TRY(c, hipEventRecord(c->ev[e], st));
  float* got = (float*)c->got.p;
  if (!c->user_have_orientation) {
    OrientParams op;
    op.gaussian_factor = p.orient_gaussian_factor;
    op.sample_factor = p.orient_gaussian_factor * p.orient_window_factor;
    op.ln_sigma_step = s.ln_sigma_step;
    op.num_orientation = p.fixed_orientation ? 0 : p.max_orientation;
    op.subpixel = 0;
    op.half_sift = p.half_sift;
    op.existing = 1;
    for (int l = 0; l < kMaxLev; l++) op.level_sigma[l] = l <= s.level_max ? s.level_sigma[l] : 0.0f;
    launch_orientation(st, g, op, list, list_total, c->cap_raw, got, (FRec*)c->recs.p, (int*)c->ocount.p, 1);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[5], st));
  LimitParams lp;
  lp.method = 0;
  lp.threshold = -1;  // LimitFeatureCount returns at once for existing keypoints (SiftPyramid.cpp:203)
  launch_feature_scan(st, g, lp, 0, list, list_total, c->cap_raw, (const int*)c->ocount.p, (int*)c->foffset.p,
                      (int*)c->fsrc.p, (int*)c->feat_total.p, (int*)c->feat_first.p, c->cap_feat,
                      (int*)c->overflow.p, (int*)c->img_base.p, (int*)c->h_small.p, 1);
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[6], st));
  DescParams dsp;
  dsp.window_factor = p.desc_window_factor;
  dsp.half_sift = p.half_sift;
  dsp.normalize = p.normalize;
  dsp.multi = 0;
  dsp.lowe_origin = p.lowe_origin;
  dsp.octave_sigma = first_octave_sigma(c);
  dsp.dog = g.dog;
  dsp.dynamic_indexing = p.dynamic_indexing ? 1 : 0;
  dsp.hkeys = c->host_direct ? (HostKeypoint*)c->h_keys.p : nullptr;
  dsp.hdesc = (c->host_direct && c->dim) ? (float*)c->h_desc.p : nullptr;
  dsp.first_image = 0;
  dsp.part = 0; dsp.part_den = 1;
  dsp.xcd_block = c->desc_xcd_block;
  dsp.sequential = p.descriptor_order == HESS_DESC_ORDER_SEQUENTIAL;
  dsp.pixel = 0;  // a keypoint list is described in a floating-point order (interleaved unless the sequential one is asked for)
  c->nparts = 1;
  c->part_features = false;
  launch_descriptor(st, g, dsp, list, c->cap_raw, (const FRec*)c->recs.p, (const int*)c->fsrc.p,
                    (const int*)c->feat_total.p, (const int*)c->feat_first.p, (const int*)c->img_base.p, got,
                    (HostKeypoint*)c->keys.p, c->dim ? (float*)c->desc.p : nullptr, c->cap_feat, 1);
  HIP_TRY(c, hipEventRecord(c->ev[7], st));
  return 0;
}

// ---- staging helpers (hess_submit_host, pageable input) ----
void stager_copy(Stager& sg, int k) {
  const size_t off = (size_t)k * sg.chunk, len = std::min(sg.chunk, sg.bytes - off);
  memcpy(sg.dst + off, sg.src + off, len);
  { std::lock_guard<std::mutex> lk(sg.mu); sg.state[k].store(2, std::memory_order_release); }
  sg.cv_done.notify_all();
}

void stager_main(Stager* sgp) {
  Stager& sg = *sgp;
  unsigned long long seen = 0;
  std::unique_lock<std::mutex> lk(sg.mu);
  for (;;) {
    sg.cv_job.wait(lk, [&] { return sg.stop || sg.gen != seen; });
    if (sg.stop) return;
    seen = sg.gen;
    lk.unlock();
    for (;;) {
      const int k = sg.next_hi.fetch_sub(1, std::memory_order_acq_rel);
      if (k < 0) break;
      int expect = 0;
      if (sg.state[k].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) stager_copy(sg, k);
      else break;  // met the calling thread coming up: everything is claimed
    }
    lk.lock();
    sg.active--;
    sg.cv_done.notify_all();
  }
}

void stager_start(Stager& sg) {
  if (sg.tried) return;
  sg.tried = true;
  for (int t = 0; t < Stager::kHelpers; t++) {
    try { sg.th[sg.nth] = std::thread(stager_main, &sg); sg.nth++; } catch (...) { break; }  // fewer helpers: the caller copies more
  }
}

void stager_stop(Stager& sg) {
  if (!sg.nth) return;
  { std::lock_guard<std::mutex> lk(sg.mu); sg.stop = true; }
  sg.cv_job.notify_all();
  for (int t = 0; t < sg.nth; t++) sg.th[t].join();
  sg.nth = 0;
}

// ---- copier thread (kDeliverDma) ----
// One job at a time: wait on the host for the event behind the batch's last kernel, read the packed total from the
// pinned count block (stored there by feature_scan_kernel), copy exactly that many records on the copy-only stream.
std::atomic<int> g_copier_count{0};  // contexts with a copier, for spreading them over the preferred engines

// Bind the copier to ROCr: the agents that own the result buffers (from the pointers themselves), an SDMA engine of
// the set ROCr recommends for device->host on this pair of agents (contexts take turns), a completion signal.
bool copier_hsa_setup(hess_ctx* c) {
  Copier& cp = c->cp;
  if (cp.hsa_ready) return true;
  if (cp.hsa_failed) return false;
  cp.hsa_failed = true;
  if (const char* m = getenv("HESS_COPIER")) if (!strcmp(m, "hip")) return false;
  if (hsa_init() != HSA_STATUS_SUCCESS) return false;  // reference-counted: HIP has initialised ROCr already
  hsa_amd_pointer_info_t pi;
  memset(&pi, 0, sizeof(pi));
  pi.size = sizeof(pi);
  if (hsa_amd_pointer_info(c->keys.p, &pi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pi.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
  cp.gpu_agent = pi.agentOwner;
  memset(&pi, 0, sizeof(pi));
  pi.size = sizeof(pi);
  // (the host side from the count block: always the runtime's own pinned allocation, also when the result buffers
  // are registered shared memory, whose owner ROCr reports differently)
  if (hsa_amd_pointer_info(c->h_small.p, &pi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pi.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
  cp.cpu_agent = pi.agentOwner;
  if (hsa_signal_create(0, 0, nullptr, &cp.sig) != HSA_STATUS_SUCCESS) return false;
  if (hsa_signal_create(0, 0, nullptr, &cp.sig2) != HSA_STATUS_SUCCESS) { (void)hsa_signal_destroy(cp.sig); return false; }
  uint32_t pref = 0;
  if (hsa_amd_memory_get_preferred_copy_engine(cp.cpu_agent, cp.gpu_agent, &pref) == HSA_STATUS_SUCCESS && pref) {
    const int n = __builtin_popcount(pref), k = g_copier_count.fetch_add(1) % n;
    uint32_t m = pref;
    for (int i = 0; i < k; i++) m &= m - 1;
    cp.engine = m & (~m + 1);
  }
  // ... and an engine of the host->device set for the pixel uploads (left to ROCr, an upload sometimes lands on an engine
  // that copies at a quarter of the rate: the host-to-host figure varied 15 - 21 Gpix/s from run to run)
  uint32_t pref_in = 0;
  if (hsa_amd_memory_get_preferred_copy_engine(cp.gpu_agent, cp.cpu_agent, &pref_in) == HSA_STATUS_SUCCESS && pref_in) {
    const int n = __builtin_popcount(pref_in), k = g_copier_count.load() % n;
    uint32_t m = pref_in;
    for (int i = 0; i < k; i++) m &= m - 1;
    cp.engine_in = m & (~m + 1);
  }
  if (const char* e = getenv("HESS_COPIER_ENGINE")) cp.engine = (uint32_t)strtoul(e, nullptr, 0);
  if (const char* e = getenv("HESS_UPLOAD_ENGINE")) cp.engine_in = (uint32_t)strtoul(e, nullptr, 0);
  cp.hsa_failed = false;
  cp.hsa_ready = true;
  return true;
}

// Host wait for a copy's completion signal to drop below `below`, in slices of a second up to HESS_COPY_TIMEOUT_S
// (default 10): a lost completion must not hang hess_wait / hess_destroy (the reference returns 0 on device errors,
// SiftPyramid.h:162-163, it never hangs).  Returns 0 when the copies completed, 1 when the limit expired, 2 when ROCr
// reported a failed copy (it then leaves a NEGATIVE value in the signal); *last = the value seen.
// HESS_COPIER_FAULT=timeout|error makes the next wait of the process report that outcome (fault injection for the
// tests; the real signal is still waited for, so nothing is left in flight).  *real (if given) says whether the outcome
// is the signal's own: then the copy may still be in flight and the context is poisoned (HESS_COPIER_FAULT=poisoned
// reports a timeout as if it were real, after the copy has in fact completed).
std::atomic<int> g_copier_fault{-1};  // -1: not read yet, 0: none, 1: timeout, 2: error, 3: poisoned (consumed by the first wait)
int wait_copy_signal(hsa_signal_t sig, hsa_signal_value_t below, hsa_signal_value_t* last, bool injectable = true, bool* real = nullptr) {
  if (real) *real = false;
  static const double limit_s = [] { const char* e = getenv("HESS_COPY_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 10.0; }();
  static const uint64_t ticks_per_s = [] {
    uint64_t f = 0;
    return (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &f) == HSA_STATUS_SUCCESS && f) ? f : (uint64_t)100000000;
  }();
  int inject = injectable ? g_copier_fault.load() : 0;
  if (injectable && inject < 0) {
    const char* e = getenv("HESS_COPIER_FAULT");
    int want = !e ? 0 : (!strcmp(e, "timeout") ? 1 : (!strcmp(e, "error") ? 2 : (!strcmp(e, "poisoned") ? 3 : 0)));
    int expect = -1;
    if (!g_copier_fault.compare_exchange_strong(expect, want)) want = expect;
    inject = want;
  }
  const auto t0 = std::chrono::steady_clock::now();
  hsa_signal_value_t v;
  for (;;) {
    v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, below, ticks_per_s, HSA_WAIT_STATE_BLOCKED);
    if (v < below) break;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
      if (last) *last = v;
      if (real) *real = true;
      return 1;
    }
  }
  if (last) *last = v;
  if (inject > 0) {
    int expect = inject;
    if (g_copier_fault.compare_exchange_strong(expect, 0)) {
      if (inject == 3 && real) *real = true;
      return inject == 3 ? 1 : inject;
    }
  }
  if (v < 0 && real) *real = true;
  return v < 0 ? 2 : 0;
}

// keys + descriptors of features [first, first + total) to the pinned host buffers by SDMA.  0: done; 1: ROCr refused
// to take the copy, use the fallback; 2: a copy was taken and did not complete (timeout or error, `why`): the batch fails.
int copier_hsa_copy(hess_ctx* c, size_t first, size_t total, char* why, size_t why_len) {
  Copier& cp = c->cp;
  hsa_signal_value_t seen = 0;
  bool real = false;
  auto lost = [&](int w) {
    snprintf(why, why_len, "device->host copy of the results %s (signal value %lld, engine 0x%x)",
             w == 1 ? "did not complete in time" : "failed", (long long)seen, cp.engine);
    cp.hsa_ready = false; cp.hsa_failed = true;  // later batches take the stream copy; the signals are not reused
    if (real) c->poisoned.store(true);           // the copy may still land: see hess_ctx::poisoned
    return 2;
  };
  auto one = [&](hsa_signal_t sig, void* dst, const void* src, size_t bytes) {
    hsa_signal_store_relaxed(sig, 1);
    hsa_status_t st = cp.engine
        ? hsa_amd_memory_async_copy_on_engine(dst, cp.cpu_agent, src, cp.gpu_agent, bytes, 0, nullptr, sig,
                                              (hsa_amd_sdma_engine_id_t)cp.engine, false)
        : hsa_amd_memory_async_copy(dst, cp.cpu_agent, src, cp.gpu_agent, bytes, 0, nullptr, sig);
    if (st != HSA_STATUS_SUCCESS && cp.engine)  // engine busy or not available: let ROCr choose
      st = hsa_amd_memory_async_copy(dst, cp.cpu_agent, src, cp.gpu_agent, bytes, 0, nullptr, sig);
    return st == HSA_STATUS_SUCCESS;
  };
  if (!one(cp.sig, (char*)c->h_keys.p + first * sizeof(HostKeypoint), (const char*)c->keys.p + first * sizeof(HostKeypoint),
           total * sizeof(HostKeypoint)))
    return 1;
  const bool second = c->dim && one(cp.sig2, (char*)c->h_desc.p + first * c->dim * 4, (const char*)c->desc.p + first * c->dim * 4, total * c->dim * 4);
  // (both copies are in flight on the same engine; each has its own signal)
  if (const int w = wait_copy_signal(cp.sig, 1, &seen, true, &real)) {
    if (second) { bool r2 = false; (void)wait_copy_signal(cp.sig2, 1, nullptr, false, &r2); real = real || r2; }
    return lost(w);
  }
  if (c->dim && !second) return 1;  // ROCr took the first copy (done by now) and refused the second: the fallback copies both
  if (second)
    if (const int w = wait_copy_signal(cp.sig2, 1, &seen, false, &real)) return lost(w);
  return 0;
}

void copier_main(hess_ctx* c) {
  Copier& cp = c->cp;
  (void)hipSetDevice(c->device);
  std::unique_lock<std::mutex> lk(cp.mu);
  for (;;) {
    cp.cv.wait(lk, [&] { return cp.stop || cp.has_job; });
    if (cp.stop) return;
    const int batch = cp.batch;
    lk.unlock();
    int rc = 0;
    bool overflow = false;
    char msg[256] = "";
    auto fail = [&](const char* what, hipError_t e) {
      snprintf(msg, sizeof(msg), "%s failed: %s (copier)", what, hipGetErrorString(e));
      rc = e == hipErrorOutOfMemory ? HESS_ERR_NOMEM : HESS_ERR_DEVICE;
    };
    hipError_t e = hipSuccess;
    if (cp.upload_first) {  // wait for the pixels on the host, then enqueue the batch
      const auto t0 = std::chrono::steady_clock::now();
      hsa_signal_value_t seen = 0;
      bool real = false;
      if (const int w = wait_copy_signal(cp.sig_in, 1, &seen, true, &real)) {
        if (real) c->poisoned.store(true);  // the upload may still write the staging area: see hess_ctx::poisoned
        // the pixels never arrived (or arrived wrong): the kernels are NOT run on them
        snprintf(msg, sizeof(msg), "host->device upload of the pixels %s (signal value %lld) (copier)",
                 w == 1 ? "did not complete in time" : "failed", (long long)seen);
        rc = HESS_ERR_DEVICE;
        cp.have_sig_in = false;  // (a signal that may still be written is left alone, not reused)
      }
      PendingRun& r = *cp.run;
      r.t_load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      cp.nparts = 1;
      cp.part_features = false;
      if (!rc) {
        try {
          rc = enqueue(c, r.dev, r.pitch, r.image_stride, r.batch, r.format, r.pixtype);
        } catch (...) { rc = HESS_ERR_NOMEM; snprintf(msg, sizeof(msg), "out of host memory (copier)"); }
        if (!rc && (e = hipEventRecord(cp.ev_done, c->st)) != hipSuccess) fail("hipEventRecord", e);
        if (rc && !msg[0]) snprintf(msg, sizeof(msg), "%s", c->err.c_str());
        if (!rc) {  // (a half-run enqueue leaves no parts to wait for)
          cp.nparts = c->nparts;
          cp.part_features = c->part_features;
          for (int k = 0; k < Copier::kMaxParts; k++) cp.part_end[k] = c->part_end[k];
        }
      }
    }
    // Several parts when the batch's descriptors were launched in groups of images (nparts > 1): a group's results
    // cross while the next group is computed; else one part behind the last kernel.  The counts (and the overflow
    // words) are in the pinned count block since feature_scan_kernel, i.e. before any of the events.
    const int nparts = cp.nparts > 1 ? cp.nparts : 1;
    if (!rc && (e = hipEventSynchronize(nparts > 1 ? cp.ev_part[0] : cp.ev_done)) != hipSuccess) fail("hipEventSynchronize", e);
    if (!rc) {
      const int* hs = (const int*)c->h_small.p;
      overflow = hs[batch + 1] != 0 || hs[batch + 2] != 0 || hs[batch + 3] != 0;  // (word 3: a device-side error, nothing to copy)
      const size_t total = overflow ? 0 : (size_t)hs[batch];
      DevBuf *hk = &c->h_keys, *hd = &c->h_desc;
      if (total) {
        // (the pinned buffers hold the worst case unless that exceeds 512 MB: then they grow here, rarely)
        if (hk->bytes < total * sizeof(HostKeypoint) || (c->dim && hd->bytes < total * c->dim * 4)) {
          if (ensure(c, *hk, total * sizeof(HostKeypoint), true) || (c->dim && ensure(c, *hd, total * c->dim * 4, true))) {
            snprintf(msg, sizeof(msg), "pinned result buffers: %s (copier)", c->err.empty() ? "allocation failed" : c->err.c_str());
            rc = HESS_ERR_NOMEM;
          }
        }
      }
      auto copy_part = [&](size_t first, size_t n) {
        if (rc || !n) return;
        if (copier_hsa_setup(c)) {
          const int hr = copier_hsa_copy(c, first, n, msg, sizeof(msg));
          if (hr == 0) return;  // both blocks are in host memory
          if (hr == 2) { rc = HESS_ERR_DEVICE; return; }
        }
        const size_t kb = sizeof(HostKeypoint), db = (size_t)c->dim * 4;
        if ((e = hipMemcpyAsync((char*)hk->p + first * kb, (const char*)c->keys.p + first * kb, n * kb, hipMemcpyDeviceToHost, cp.cs)) != hipSuccess)
          fail("hipMemcpyAsync(keys)", e);
        if (!rc && c->dim &&
            (e = hipMemcpyAsync((char*)hd->p + first * db, (const char*)c->desc.p + first * db, n * db, hipMemcpyDeviceToHost, cp.cs)) != hipSuccess)
          fail("hipMemcpyAsync(desc)", e);
        if (!rc && (e = hipStreamSynchronize(cp.cs)) != hipSuccess) fail("hipStreamSynchronize(copy stream)", e);
      };
      size_t done_feats = 0;
      for (int k = 0; k < nparts; k++) {
        if (k > 0 && (e = hipEventSynchronize(k < nparts - 1 ? cp.ev_part[k] : cp.ev_done)) != hipSuccess) fail("hipEventSynchronize", e);
        // features of the images so far -- or, of one image's list, the bound the part's launch formed (k_feature.hip, feature_part)
        const size_t upto = overflow ? 0 : (k == nparts - 1 ? total : cp.part_features ? (size_t)((long long)total * (k + 1) / nparts)
                                                                                     : (size_t)hs[cp.part_end[k]]);
        copy_part(done_feats, upto - done_feats);
        done_feats = upto;
      }
    }
    lk.lock();
    cp.rc = rc;
    cp.overflow = overflow;
    snprintf(cp.err, sizeof(cp.err), "%s", msg);
    cp.has_job = false;
    cp.done = true;
    cp.cv.notify_all();
  }
}

int copier_start(hess_ctx* c) {
  Copier& cp = c->cp;
  if (cp.started) return 0;
  HIP_TRY(c, hipStreamCreateWithFlags(&cp.cs, hipStreamNonBlocking));
  HIP_TRY(c, hipEventCreateWithFlags(&cp.ev_done, hipEventDisableTiming));
  for (hipEvent_t& ev : cp.ev_part) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  try {
    cp.th = std::thread(copier_main, c);
  } catch (...) {
    set_err(c, "cannot start the copier thread");
    return HESS_ERR_NOMEM;
  }
  cp.started = true;
  return 0;
}

void copier_stop(hess_ctx* c) {
  Copier& cp = c->cp;
  if (cp.started) {
    {
      std::unique_lock<std::mutex> lk(cp.mu);
      cp.cv.wait(lk, [&] { return cp.done; });
      cp.stop = true;
      cp.cv.notify_all();
    }
    cp.th.join();
    cp.started = false;
  }
  // (a poisoned context's signals may still be written by a late copy: left alone, like the buffers)
  if (cp.hsa_ready) { (void)hsa_signal_destroy(cp.sig); (void)hsa_signal_destroy(cp.sig2); cp.hsa_ready = false; }
  if (cp.have_sig_in && !c->poisoned.load()) { (void)hsa_signal_destroy(cp.sig_in); cp.have_sig_in = false; }
  if (cp.cs) { (void)hipStreamDestroy(cp.cs); cp.cs = nullptr; }
  if (cp.ev_done) { (void)hipEventDestroy(cp.ev_done); cp.ev_done = nullptr; }
  for (hipEvent_t& ev : cp.ev_part) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
}

// How the results of a batch of `batch` images reach the host (see the kDeliver* comment): small batches through the
// descriptor kernel's own stores (lowest latency), larger ones by the copier thread's DMA copy (no kernel waits for
// PCIe).  HESS_DELIVERY overrides.
void choose_delivery(hess_ctx* c, int batch) {
  // Small batches keep the in-kernel mirror (latency: no event wake-up, no copy behind the last kernel) -- unless their
  // results are large: a kernel that stores tens of megabytes into host memory waits for the link (one 4096^2 image with
  // 102 k half descriptors, 28 MB: 0.76 ms against 0.47), while the copier's DMA copy of a part runs beside the next
  // part's launch -- 15 % more images per second on three pipelined contexts.  That needs a caller who overlaps: a batch
  // handed over by hess_submit_* (hess_run_*, i.e. submit + wait in one call, keeps the mirror: nothing to overlap with,
  // and through the class with pageable pixels the copier's route is 15 % slower for a 4096^2 image).  "Large" is judged by
  // what the context's last batch of this size delivered (the capacity is a worst case many times the typical count): the
  // first batch of a context uses the mirror.  HESS_MIRROR_MAX_MB (16) is the limit.
  const size_t expect = (c->last_result_batch == batch && !c->caller_waits) ? c->last_result_bytes : 0;
  const bool small = batch <= c->mirror_max_batch && !(batch >= 2 && !c->caller_waits);  // (a submitted pair: see enqueue())
  int d = c->delivery_pref >= 0 ? c->delivery_pref : (small && expect <= c->mirror_max_bytes ? kDeliverMirror : kDeliverDma);
  if (d == kDeliverMirror && !c->host_fits) d = kDeliverDma;  // the mirror needs the worst case pinned up front
  if (d == kDeliverDma && copier_start(c) != 0) d = kDeliverBlit;
  c->delivery = d;
  c->host_direct = d == kDeliverMirror;
}

// Enqueue the whole path (the per-image counts reach the pinned count block by feature_scan_kernel's own stores);
// returns without waiting.
int submit_inner(hess_ctx* c, const PendingRun& r) {
  if (!c->user_keys.empty() && r.batch != 1) {
    set_err(c, "a keypoint list applies to a single image");
    return HESS_ERR_ARG;
  }
  int rc = plan(c, r.width, r.height, r.batch);
  if (rc) return rc;
  choose_delivery(c, r.batch);
  HIP_TRY(c, hipGetLastError());
  rc = enqueue(c, r.dev, r.pitch, r.image_stride, r.batch, r.format, r.pixtype);
  if (rc) return rc;
  HIP_TRY(c, hipGetLastError());
  if (c->delivery == kDeliverDma) {
    Copier& cp = c->cp;
    HIP_TRY(c, hipEventRecord(cp.ev_done, c->st));
    std::lock_guard<std::mutex> lk(cp.mu);
    cp.batch = r.batch;
    cp.upload_first = false;
    cp.nparts = c->nparts;
    cp.part_features = c->part_features;
    for (int k = 0; k < Copier::kMaxParts; k++) cp.part_end[k] = c->part_end[k];
    cp.done = false;
    cp.has_job = true;
    cp.cv.notify_all();
  }
  return 0;
}

// (nothing thrown crosses the C ABI: enqueue_user builds host vectors)
int submit_impl(hess_ctx* c, const PendingRun& r) {
  try {
    return submit_inner(c, r);
  } catch (...) {
    set_err(c, "out of host memory while preparing the batch");
    return HESS_ERR_NOMEM;
  }
}

// Wait for the submitted batch, grow storage and re-run if a list overflowed; with kDeliverBlit bring the
// keypoints and descriptors of the whole batch to the host with one transfer each (the other modes have
// delivered them by now).
int wait_inner(hess_ctx* c, const PendingRun& r) {
  int rc;
  int* hs = (int*)c->h_small.p;
  const int batch = r.batch;
  for (int attempt = 0;; attempt++) {
    if (c->delivery == kDeliverDma) {
      Copier& cp = c->cp;
      std::unique_lock<std::mutex> lk(cp.mu);
      cp.cv.wait(lk, [&] { return cp.done; });
      if (cp.rc) { set_err(c, "%s", cp.err); return cp.rc; }
    } else {
      HIP_TRY(c, hipStreamSynchronize(c->st));
    }
    const int of_raw = hs[batch + 1], of_feat = hs[batch + 2];
    if (hs[batch + 3]) {  // raised by a kernel that gave up a bounded wait (topk_select_kernel's look-back): no results
      set_err(c, "device-side wait did not complete (top-K look-back); the batch has no results");
      return HESS_ERR_DEVICE;
    }
    if (!of_raw && !of_feat) break;
    if (attempt >= 8) { set_err(c, "feature storage keeps overflowing"); return HESS_ERR_NOMEM; }
    // grow-only reallocation, then run the batch again (reference: SetLevelFeatureNum grows on demand,
    // PyramidCU.cpp:393-397)
    if (of_raw) c->cap_raw = of_raw + of_raw / 4;
    if (of_feat) c->cap_feat = of_feat + of_feat / 4;
    c->planned = false;
    c->regrown++;
    if (c->p.verbose & 1) fprintf(stderr, "hessgpu: feature storage grown (raw %d, features %d)\n", c->cap_raw, c->cap_feat);
    if ((rc = submit_impl(c, r))) return rc;
  }
  drain_profile(c);
  c->counts.resize(batch);
  c->offs.assign(batch + 1, 0);
  for (int b = 0; b < batch; b++) {
    c->counts[b] = hs[b + 1] - hs[b];
    c->offs[b + 1] = (size_t)hs[b + 1];
  }
  const size_t total = c->offs[batch];
  c->seen_features = batch ? *std::max_element(c->counts.begin(), c->counts.end()) : 0;
  c->last_result_bytes = total * (sizeof(HostKeypoint) + (size_t)c->dim * 4);
  c->last_result_batch = batch;
  if ((rc = ensure(c, c->h_keys, (total ? total : 1) * sizeof(HostKeypoint), true))) return rc;
  if (c->dim && (rc = ensure(c, c->h_desc, (total ? total : 1) * c->dim * 4, true))) return rc;
  if (total && c->delivery == kDeliverBlit) {
    HIP_TRY(c, hipMemcpyAsync(c->h_keys.p, c->keys.p, total * sizeof(HostKeypoint), hipMemcpyDeviceToHost, c->st));
    if (c->dim)
      HIP_TRY(c, hipMemcpyAsync(c->h_desc.p, c->desc.p, total * c->dim * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
  }
  c->user_result = false;
  if (!c->user_keys.empty()) {
    // back to input order (PyramidCU.cpp:537-549,1157-1168); the caller's keypoints are returned
    // unchanged unless DownloadKeypoints would run (-m 1 / -ofix, SiftPyramid.cpp:160-171)
    const int num = (int)c->user_keys.size();
    const int listed = (int)std::min<size_t>(total, (size_t)num);
    const bool download = !c->user_have_orientation && ((c->p.max_orientation < 2) || c->p.fixed_orientation);
    c->u_keys = c->user_keys;
    c->u_desc.assign((size_t)num * (c->dim ? c->dim : 1), 0.0f);
    for (int i = 0; i < listed; i++) {
      const int k = c->user_kindex[i];
      if (download) memcpy(&c->u_keys[k], (HostKeypoint*)c->h_keys.p + i, sizeof(hess_keypoint));
      if (c->dim) memcpy(&c->u_desc[(size_t)k * c->dim], (float*)c->h_desc.p + (size_t)i * c->dim, (size_t)c->dim * 4);
    }
    c->counts[0] = num;
    c->offs[1] = (size_t)num;
    c->user_result = true;
    c->user_keys.clear();  // _existing_keypoints = 0 after RunSIFT (SiftPyramid.cpp:182-184)
    c->user_levels.clear();  // the parity hook covers one keypoint-list run
    c->user_on_current = false;
  }
  // stage times from the events of the last enqueue (config.h:17-31 order)
  memset(c->timing, 0, sizeof(c->timing));
  auto el = [&](int i, int j) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[j]); return ms; };
  c->timing[HESS_T_LOAD] = (float)r.t_load_ms;
  if (r.timed_load) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev_load[0], c->ev_load[1]); c->timing[HESS_T_LOAD] = ms; }
  if (c->stage_events) {
    c->timing[HESS_T_PYRAMID] = el(0, 1);
    c->timing[HESS_T_DETECT] = el(1, 2);
    c->timing[HESS_T_LIST] = el(2, 3);
    c->timing[HESS_T_REDUCTION] = el(3, 4);
    c->timing[HESS_T_ORIENT] = el(4, 5);
    c->timing[HESS_T_MULTI_ORIENT] = el(5, 6);
    c->timing[HESS_T_DESCRIPTOR] = el(6, 7);
  }
  c->timing[HESS_T_TOTAL] = el(0, 7) + c->timing[HESS_T_LOAD];
  c->batch = c->pyramid_batch = batch;  // only now: every error return above leaves the context without results
  return 0;
}

int wait_impl(hess_ctx* c, const PendingRun& r) {
  try {
    return wait_inner(c, r);
  } catch (...) {  // the host-side count / keypoint-list vectors
    c->user_keys.clear();
    set_err(c, "out of host memory while collecting the results");
    return HESS_ERR_NOMEM;
  }
}

}  // namespace

// ================================== C ABI ====================================================

extern "C" {

void hess_default_params(hess_params* p) { if (p) default_params(p); }

int hess_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

hess_ctx* hess_create(int device, const hess_params* params) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    fprintf(stderr, "hessgpu: no usable HIP device %d (found %d)\n", device, ndev);
    return nullptr;
  }
  hess_ctx* c = new (std::nothrow) hess_ctx();
  if (!c) return nullptr;
  if (params) c->p = *params; else default_params(&c->p);
  bool reserved_nonzero = false;
  for (int r : c->p.reserved) reserved_nonzero = reserved_nonzero || r != 0;
  // version-2 / -3 structs: same layout; 0 in the order word is what they ask for (the interleaved order, their default)
  if ((c->p.abi_version == 2 && c->p.descriptor_order == 0) || (c->p.abi_version == 3 && c->p.descriptor_order <= HESS_DESC_ORDER_SEQUENTIAL))
    c->p.abi_version = HESS_ABI_VERSION;
  if (c->p.abi_version != HESS_ABI_VERSION || c->p.dog_level_num < 0 || c->p.dog_level_num > kMaxDog ||
      c->p.descriptor_order < 0 || c->p.descriptor_order > HESS_DESC_ORDER_PIXEL || reserved_nonzero) {        // reserved words must be zero (word 0 is the test oracle's detector switch: not a product option)
    fprintf(stderr, "hessgpu: bad hess_params (abi_version %d)\n", c->p.abi_version);
    delete c;
    return nullptr;
  }
  if (c->p.first_octave < -3) c->p.first_octave = -3;  // "can't upsample by more than 8": clamped, PyramidCU.cpp:131-132
  c->device = device;
  resolve(c);
  memset(c->timing, 0, sizeof(c->timing));
  memset(c->k_ms, 0, sizeof(c->k_ms));
  memset(c->k_n, 0, sizeof(c->k_n));
  memset(c->k_bytes, 0, sizeof(c->k_bytes));
  memset(c->k_in_lds, 0, sizeof(c->k_in_lds));
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) {
    fprintf(stderr, "hessgpu: cannot create stream on device %d\n", device);
    delete c;
    return nullptr;
  }
  bool ev_ok = true;
  for (int i = 0; i < 8; i++) { c->ev[i] = nullptr; ev_ok = ev_ok && hipEventCreate(&c->ev[i]) == hipSuccess; }
  for (int i = 0; i < 2; i++) { c->ev_load[i] = nullptr; ev_ok = ev_ok && hipEventCreate(&c->ev_load[i]) == hipSuccess; }
  c->have_ev = true;
  if (!ev_ok) {
    fprintf(stderr, "hessgpu: cannot create events on device %d\n", device);
    hess_destroy(c);
    return nullptr;
  }
  if (const char* d = getenv("HESS_DELIVERY")) {
    if (!strcmp(d, "mirror")) c->delivery_pref = kDeliverMirror;
    else if (!strcmp(d, "dma")) c->delivery_pref = kDeliverDma;
    else if (!strcmp(d, "blit")) c->delivery_pref = kDeliverBlit;
  }
  if (const char* m = getenv("HESS_MIRROR_MAX_BATCH")) c->mirror_max_batch = atoi(m);
  if (const char* ci = getenv("HESS_INITIAL_CAP")) c->cap_init = atoi(ci) > 0 ? atoi(ci) : 0;
  c->no_pair = getenv("HESS_NO_PAIR") != nullptr;
  c->no_top_fusion = getenv("HESS_NO_TOP_FUSION") != nullptr;
  c->no_first_fusion = getenv("HESS_NO_FIRST_FUSION") != nullptr;
  if (const char* e = getenv("HESS_MIRROR_MAX_MB")) c->mirror_max_bytes = (size_t)std::max(0, atoi(e)) << 20;
  if (const char* cf = getenv("HESS_CHAIN_FROM")) c->chain_from = atoi(cf);
  c->no_host_upload = getenv("HESS_NO_SIDE_UPLOAD") != nullptr;
  if (const char* dpn = getenv("HESS_DESC_PARTS")) c->desc_parts = atoi(dpn);
  if (const char* sr = getenv("HESS_STREAM_ROWS")) c->stream_rows = atoi(sr) > 0 ? (atoi(sr) / 3) * 3 : 0;
  if (const char* dx = getenv("HESS_DESC_XCD")) c->desc_xcd_block = atoi(dx) > 0 ? atoi(dx) : 0;
  return c;
}

void hess_destroy(hess_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->st) (void)hipStreamSynchronize(c->st);
  copier_stop(c);
  stager_stop(c->sg);
  DevBuf* bufs[] = {&c->gauss, &c->deth, &c->got, &c->input_f32, &c->upsampled, &c->stage, &c->zeroed, &c->rowoff,
                    &c->level_count, &c->raw_total, &c->found, &c->task_count, &c->raw, &c->sel, &c->sel_total,
                    &c->recs, &c->ocount, &c->foffset, &c->fsrc, &c->feat_total, &c->feat_first, &c->img_base,
                    &c->keys, &c->desc};
  // A poisoned context (a DMA copy that was lost may still be in flight or land late) deliberately leaks the copy's
  // sources and targets -- result buffers on both sides and the pixel staging area -- rather than hand memory that may
  // still be written back to the allocator; a shared result buffer stays mapped for the same reason.
  const bool leak = c->poisoned.load();
  for (DevBuf* b : bufs)
    if (!(leak && (b == &c->keys || b == &c->desc || b == &c->stage))) release(*b);
  if (!leak) {
    release(c->h_keys, true);
    release(c->h_desc, true);
  }
  release(c->h_small, true);
  if (c->share_dir) {
    (void)munmap(c->share_dir, 4096);
    char dir[256];
    snprintf(dir, sizeof(dir), "/%s.h", c->share.c_str());
    (void)shm_unlink(dir);
    c->share_dir = nullptr;
  }
  release(c->h_stage, true);
  if (c->have_ev) {
    for (int i = 0; i < 8; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 2; i++) if (c->ev_load[i]) (void)hipEventDestroy(c->ev_load[i]);
  }
  for (auto& ep : c->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c->pend;
  delete c;
}

static int refuse_poisoned(hess_ctx* c) {
  if (!c->poisoned.load()) return 0;
  set_err(c, "the context is poisoned: a DMA copy did not complete and may still write its buffers; destroy the context");
  return HESS_ERR_DEVICE;
}

// The runtime objects a batch of this size will need are created by hess_reserve, not by the first batch: the copier
// thread and its binding to ROCr incl. the SDMA engine's queue (a 64-byte copy into each result buffer, only while the
// context holds no results), the hardware queue behind the context's stream -- and, once per reserved shape, ONE DRY
// BATCH of that shape on zeroed pixels, so that the first real batch finds the context in the state its second batch
// would (round 4's driver run: two of seven contexts ran their first batch inside a 20-step timed region; a first batch
// took 1.95 ms against 0.9).  The reference keeps allocation out of its steady-state numbers the same way
// (hessgpucmd.cpp:137-138,172: the first run is the allocating one).  The dry batch runs on a scratch image of its
// own (the staging area may hold the caller's last input, hess_last_input) and leaves no results behind.
// HESS_NO_PRIME_BATCH=1 switches it off (A/B).
static int prime(hess_ctx* c, int width, int height, int batch) {
  if (c->pend && c->pend->active) return 0;
  choose_delivery(c, batch);
  if (c->delivery == kDeliverDma && c->batch == 0 && c->keys.p && c->h_keys.p && c->h_keys.bytes >= 64 && !c->cp.has_job) {
    Copier& cp = c->cp;
    if (copier_hsa_setup(c)) {
      auto tiny = [&](hsa_signal_t sig, void* dst, const void* src) {
        hsa_signal_store_relaxed(sig, 1);
        hsa_status_t st = cp.engine
            ? hsa_amd_memory_async_copy_on_engine(dst, cp.cpu_agent, src, cp.gpu_agent, 64, 0, nullptr, sig,
                                                  (hsa_amd_sdma_engine_id_t)cp.engine, false)
            : hsa_amd_memory_async_copy(dst, cp.cpu_agent, src, cp.gpu_agent, 64, 0, nullptr, sig);
        if (st == HSA_STATUS_SUCCESS && wait_copy_signal(sig, 1, nullptr, false) != 0) { cp.hsa_ready = false; cp.hsa_failed = true; }
      };
      tiny(cp.sig, c->h_keys.p, c->keys.p);
      if (cp.hsa_ready && c->dim && c->desc.p && c->h_desc.p && c->h_desc.bytes >= 64) tiny(cp.sig2, c->h_desc.p, c->desc.p);
    }
  }
  if (c->zeroed.p && c->zeroed.bytes >= 64) {
    HIP_TRY(c, hipMemsetAsync(c->zeroed.p, 0, 64, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
  }
  static const bool no_dry = getenv("HESS_NO_PRIME_BATCH") != nullptr;
  const long long shape = ((long long)width << 40) ^ ((long long)height << 16) ^ batch;
  if (!no_dry && c->batch == 0 && c->user_keys.empty() && c->primed_shape != shape) {
    const size_t bytes = (size_t)batch * width * height;
    void* px = nullptr;
    if (hipMalloc(&px, bytes + 16) != hipSuccess) { (void)hipGetLastError(); return 0; }  // no room for the scratch image: no dry batch
    int rc = 0;
    if (hipMemsetAsync(px, 0, bytes, c->st) != hipSuccess) rc = HESS_ERR_DEVICE;
    const PendingRun r{px, width, height, width, batch, HESS_FMT_LUM, HESS_PIX_U8, (size_t)width * height, 0.0, false, false};
    if (!rc) rc = submit_impl(c, r);
    if (!rc) rc = wait_impl(c, r);
    (void)hipStreamSynchronize(c->st);
    (void)hipFree(px);
    c->batch = c->pyramid_batch = 0;  // a dry batch leaves neither results nor a current image
    memset(c->timing, 0, sizeof(c->timing));
    if (rc) return rc;
    c->primed_shape = shape;
  }
  return 0;
}

int hess_reserve(hess_ctx* c, int width, int height, int batch) {
  if (!c || width <= 0 || height <= 0 || batch <= 0) return HESS_ERR_ARG;
  if (refuse_poisoned(c)) return HESS_ERR_DEVICE;
  HIP_TRY(c, hipSetDevice(c->device));
  const int rc = plan(c, width, height, batch);
  if (rc) return rc;
  return prime(c, width, height, batch);
}

static int check_run_args(hess_ctx* c, const void* pixels, int width, int height, int pitch, int batch, int format,
                          int pixtype) {
  if (!c) return HESS_ERR_ARG;
  if (refuse_poisoned(c)) return HESS_ERR_DEVICE;
  if (!pixels || width <= 0 || height <= 0 || batch <= 0 || pitch <= 0 || !fmt_channels(format) ||
      pixtype < HESS_PIX_U8 || pixtype > HESS_PIX_F32) {
    set_err(c, "bad argument");
    return HESS_ERR_ARG;
  }
  return 0;
}

// HESS_CHAIN_STAMPS=1 (diagnostics, stderr): per finished batch the device interval of its launch chain (first event to
// last event of the context's stream) and the host times of submit / wait return, all in ms since one base that is taken
// on both clocks when the first batch is submitted -- where do the contexts of a pipelined loop spend their time?
namespace {
struct ChainStamps {
  bool on = getenv("HESS_CHAIN_STAMPS") && atoi(getenv("HESS_CHAIN_STAMPS")) != 0;
  std::mutex mu;
  hipEvent_t base = nullptr;
  std::chrono::steady_clock::time_point host0;
  double now() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host0).count(); }
} g_stamps;
void chain_stamp_submit(hess_ctx* c, bool before) {
  if (!g_stamps.on) return;
  std::lock_guard<std::mutex> lk(g_stamps.mu);
  if (!g_stamps.base) {
    if (hipEventCreate(&g_stamps.base) != hipSuccess || hipEventRecord(g_stamps.base, c->st) != hipSuccess ||
        hipEventSynchronize(g_stamps.base) != hipSuccess) { g_stamps.on = false; return; }
    g_stamps.host0 = std::chrono::steady_clock::now();
  }
  (before ? c->stamp_submit0 : c->stamp_submit1) = g_stamps.now();
}
void chain_stamp_done(hess_ctx* c, double wait0) {
  if (!g_stamps.on || !g_stamps.base) return;
  float a = 0.0f, b = 0.0f;
  if (hipEventElapsedTime(&a, g_stamps.base, c->ev[0]) != hipSuccess || hipEventElapsedTime(&b, g_stamps.base, c->ev[7]) != hipSuccess) return;
  fprintf(stderr, "hess chain ctx %p: host submit %.3f - %.3f  device %.3f - %.3f  host wait %.3f - %.3f\n", (void*)c,
          c->stamp_submit0, c->stamp_submit1, (double)a, (double)b, wait0, g_stamps.now());
}
}  // namespace

int hess_submit_device(hess_ctx* c, const void* dev_pixels, int width, int height, int pitch, size_t image_stride,
                       int batch, int format, int pixtype) {
  int rc = check_run_args(c, dev_pixels, width, height, pitch, batch, format, pixtype);
  if (rc) return rc;
  if (c->pend && c->pend->active) { set_err(c, "a submitted batch is still pending: call hess_wait first"); return HESS_ERR_STATE; }
  HIP_TRY(c, hipSetDevice(c->device));
  c->batch = c->pyramid_batch = 0;  // the results and the pyramid of the run before are gone from here on
  if (!c->pend && !(c->pend = new (std::nothrow) PendingRun())) { set_err(c, "out of memory"); return HESS_ERR_NOMEM; }
  *c->pend = PendingRun{dev_pixels, width, height, pitch, batch, format, pixtype, image_stride, 0.0, false, false};
  chain_stamp_submit(c, true);
  rc = submit_impl(c, *c->pend);
  chain_stamp_submit(c, false);
  if (rc) return rc;
  c->pend->active = true;
  return 0;
}

int hess_wait(hess_ctx* c) {
  if (!c) return HESS_ERR_ARG;
  if (!c->pend || !c->pend->active) { set_err(c, "nothing submitted"); return HESS_ERR_STATE; }
  HIP_TRY(c, hipSetDevice(c->device));
  c->pend->active = false;
  const double wait0 = g_stamps.on && g_stamps.base ? g_stamps.now() : 0.0;
  const int rc = wait_impl(c, *c->pend);
  if (!rc) chain_stamp_done(c, wait0);
  return rc;
}

int hess_run_device(hess_ctx* c, const void* dev_pixels, int width, int height, int pitch, size_t image_stride,
                    int batch, int format, int pixtype) {
  if (c) c->caller_waits = true;
  int rc = hess_submit_device(c, dev_pixels, width, height, pitch, image_stride, batch, format, pixtype);
  if (!rc) rc = hess_wait(c);
  if (c) c->caller_waits = false;
  return rc;
}

// Host pixels: one asynchronous host->device transfer on the context's stream, then the path.  Pinned caller
// memory (hipHostMalloc / hipHostRegister) is read by the copy engine directly; pageable memory is first copied
// into the context's pinned staging buffer by the calling thread -- while the device still works on the batches
// of other contexts -- so that the transfer itself never blocks the host or the other streams of the device
// (a hipMemcpyAsync from pageable memory does both).
int hess_submit_host(hess_ctx* c, const void* pixels, int width, int height, int pitch, size_t image_stride, int batch,
                     int format, int pixtype) {
  int rc = check_run_args(c, pixels, width, height, pitch, batch, format, pixtype);
  if (rc) return rc;
  if (c->pend && c->pend->active) { set_err(c, "a submitted batch is still pending: call hess_wait first"); return HESS_ERR_STATE; }
  HIP_TRY(c, hipSetDevice(c->device));
  c->batch = c->pyramid_batch = 0;  // the results and the pyramid of the run before are gone from here on
  if (!c->pend && !(c->pend = new (std::nothrow) PendingRun())) { set_err(c, "out of memory"); return HESS_ERR_NOMEM; }
  const size_t bytes = (size_t)(batch - 1) * image_stride + (size_t)height * pitch;
  rc = ensure(c, c->stage, bytes + 16);
  if (rc) return rc;
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, pixels) == hipSuccess && at.type == hipMemoryTypeHost;
  if (pinned && c->user_keys.empty() && !c->no_host_upload) {
    // Pinned pixels of a batch that the copier thread will deliver: the upload goes to an SDMA engine directly and
    // the copier thread enqueues the kernels once it has landed (Copier::upload_first).  A copy command on the
    // context's stream would hold its hardware queue -- shared with other contexts -- for the length of the
    // transfer: 15.9 - 16.2 against 17.0 Gpix/s for six pipelined contexts, while the same bytes uploaded on the side
    // cost nothing (tools/r03/r03_h2d_bg.py).
    if ((rc = plan(c, width, height, batch))) return rc;
    choose_delivery(c, batch);
    Copier& cp = c->cp;
    hsa_amd_pointer_info_t pi;
    memset(&pi, 0, sizeof(pi));
    pi.size = sizeof(pi);
    if (c->delivery == kDeliverDma && copier_hsa_setup(c) &&
        hsa_amd_pointer_info(const_cast<void*>(pixels), &pi, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS &&
        pi.type != HSA_EXT_POINTER_TYPE_UNKNOWN &&
        (cp.have_sig_in || hsa_signal_create(1, 0, nullptr, &cp.sig_in) == HSA_STATUS_SUCCESS)) {
      cp.have_sig_in = true;
      hsa_signal_store_relaxed(cp.sig_in, 1);
      hsa_status_t up = cp.engine_in
          ? hsa_amd_memory_async_copy_on_engine(c->stage.p, cp.gpu_agent, pixels, pi.agentOwner, bytes, 0, nullptr, cp.sig_in,
                                                (hsa_amd_sdma_engine_id_t)cp.engine_in, false)
          : HSA_STATUS_ERROR;
      if (up != HSA_STATUS_SUCCESS)  // no engine chosen, or busy / not available: let ROCr choose
        up = hsa_amd_memory_async_copy(c->stage.p, cp.gpu_agent, pixels, pi.agentOwner, bytes, 0, nullptr, cp.sig_in);
      if (up == HSA_STATUS_SUCCESS) {
        c->last_input_bytes = bytes;
        *c->pend = PendingRun{c->stage.p, width, height, pitch, batch, format, pixtype, image_stride, 0.0, false, false};
        {
          std::lock_guard<std::mutex> lk(cp.mu);
          cp.batch = batch;
          cp.upload_first = true;
          cp.run = c->pend;
          cp.nparts = 1;
          cp.part_features = false;
          cp.done = false;
          cp.has_job = true;
          cp.cv.notify_all();
        }
        c->pend->active = true;
        return 0;
      }
    }
  }
  HIP_TRY(c, hipEventRecord(c->ev_load[0], c->st));
  if (pinned) {
    HIP_TRY(c, hipMemcpyAsync(c->stage.p, pixels, bytes, hipMemcpyHostToDevice, c->st));
  } else {
    (void)hipGetLastError();  // an unregistered pointer is reported as an error: not one
    if ((rc = ensure(c, c->h_stage, bytes, true))) return rc;
    // Pageable memory: copied into the pinned staging buffer in chunks, each chunk's transfer enqueued as soon as
    // it is staged, so the copy engine works while the next chunks are being copied (one core copies at about
    // 17 GB/s, a third of what the link takes: the context's helper threads share the work, see Stager).
    // 4 MB chunks; a small input (one image) is cut in four so that its transfer, too, overlaps its staging, and is
    // staged by the calling thread alone: waking the helpers costs more than they save below about 8 MB (one 1080p
    // image: 0.544 ms per call alone, 0.58 ms with helpers; profiles/r03_host_path.json).
    // Nothing here allocates or throws once the helpers exist (nothing thrown crosses the C ABI).
    Stager& sg = c->sg;
    const size_t chunk = std::min<size_t>((size_t)4 << 20, std::max<size_t>((size_t)256 << 10, ((bytes / 4 + 65535) >> 16) << 16));
    const int nchunk = (int)((bytes + chunk - 1) / chunk);
    const bool helped = bytes >= ((size_t)8 << 20) && nchunk > 1;
    if (helped) stager_start(sg);
    try {
      if ((int)sg.state.size() < nchunk) { std::vector<std::atomic<int>> grown(nchunk); sg.state.swap(grown); }
    } catch (...) { set_err(c, "out of memory"); return HESS_ERR_NOMEM; }
    {
      std::lock_guard<std::mutex> lk(sg.mu);
      for (int k = 0; k < nchunk; k++) sg.state[k].store(0, std::memory_order_relaxed);
      sg.src = (const char*)pixels; sg.dst = (char*)c->h_stage.p; sg.bytes = bytes; sg.chunk = chunk; sg.nchunk = nchunk;
      sg.next_hi.store(helped ? nchunk - 1 : -1, std::memory_order_release);
      sg.active = helped ? sg.nth : 0;
      if (helped && sg.nth) sg.gen++;
    }
    if (helped && sg.nth) sg.cv_job.notify_all();
    hipError_t cerr = hipSuccess;
    for (int k = 0; k < nchunk; k++) {
      int expect = 0;
      if (sg.state[k].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) stager_copy(sg, k);
      else if (sg.state[k].load(std::memory_order_acquire) != 2) {
        std::unique_lock<std::mutex> lk(sg.mu);
        sg.cv_done.wait(lk, [&] { return sg.state[k].load(std::memory_order_acquire) == 2; });
      }
      const size_t off = (size_t)k * chunk, len = std::min(chunk, bytes - off);
      if (cerr == hipSuccess)
        cerr = hipMemcpyAsync((char*)c->stage.p + off, (const char*)c->h_stage.p + off, len, hipMemcpyHostToDevice, c->st);
    }
    {  // the helpers are done with this job's bookkeeping before the next one rewrites it
      std::unique_lock<std::mutex> lk(sg.mu);
      sg.cv_done.wait(lk, [&] { return sg.active == 0; });
    }
    HIP_TRY(c, cerr);
  }
  HIP_TRY(c, hipEventRecord(c->ev_load[1], c->st));
  c->last_input_bytes = bytes;
  *c->pend = PendingRun{c->stage.p, width, height, pitch, batch, format, pixtype, image_stride, 0.0, false, true};
  rc = submit_impl(c, *c->pend);
  if (rc) return rc;
  c->pend->active = true;
  return 0;
}

int hess_run_host(hess_ctx* c, const void* pixels, int width, int height, int pitch, size_t image_stride, int batch,
                  int format, int pixtype) {
  if (c) c->caller_waits = true;   // (a synchronous call has nothing to overlap the delivery with: choose_delivery, enqueue)
  int rc = hess_submit_host(c, pixels, width, height, pitch, image_stride, batch, format, pixtype);
  if (!rc) rc = hess_wait(c);      // (the flag stays up: pinned pixels are enqueued by the copier thread, during the wait)
  if (c) c->caller_waits = false;
  return rc;
}

int hess_last_input(hess_ctx* c, void* out, size_t bytes) {
  if (!c || !out) return HESS_ERR_ARG;
  if (c->pend && c->pend->active) { set_err(c, "a submitted batch is still pending: call hess_wait first"); return HESS_ERR_STATE; }
  if (!c->last_input_bytes || bytes > c->last_input_bytes) { set_err(c, "no host input of that size is retained"); return HESS_ERR_STATE; }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(out, c->stage.p, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int hess_set_keypoints(hess_ctx* c, const hess_keypoint* keys, int num, int keys_have_orientation) {
  if (!c || num < 0 || (num > 0 && !keys)) return HESS_ERR_ARG;
  try {
    c->user_keys.assign(keys, keys + num);
  } catch (...) { c->user_keys.clear(); set_err(c, "out of host memory"); return HESS_ERR_NOMEM; }
  c->user_have_orientation = keys_have_orientation != 0;
  c->user_on_current = false;
  return 0;
}

int hess_run_keypoints(hess_ctx* c, const hess_keypoint* keys, int num, int keys_have_orientation) {
  if (!c || num <= 0 || !keys) return HESS_ERR_ARG;
  if (refuse_poisoned(c)) return HESS_ERR_DEVICE;
  if (!c->planned || c->pyramid_batch < 1) { set_err(c, "no current image: run an image first"); return HESS_ERR_STATE; }
  if (c->pend && c->pend->active) { set_err(c, "a submitted batch is still pending: call hess_wait first"); return HESS_ERR_STATE; }
  HIP_TRY(c, hipSetDevice(c->device));
  try {
    c->user_keys.assign(keys, keys + num);
  } catch (...) { c->user_keys.clear(); set_err(c, "out of host memory"); return HESS_ERR_NOMEM; }
  c->user_have_orientation = keys_have_orientation != 0;
  c->user_on_current = true;
  if (!c->pend) { c->user_keys.clear(); set_err(c, "no current image"); return HESS_ERR_STATE; }
  PendingRun r = *c->pend;  // geometry of the current image
  if (r.width <= 0) { c->user_keys.clear(); set_err(c, "no current image"); return HESS_ERR_STATE; }
  r.batch = 1;
  r.t_load_ms = 0.0;
  const int keep_pyramid = c->pyramid_batch;
  c->batch = 0;  // results of the run before: gone; the pyramid stays (that is the point of this entry)
  int rc = submit_impl(c, r);
  if (rc) { c->user_keys.clear(); return rc; }
  rc = wait_impl(c, r);
  c->pyramid_batch = keep_pyramid;
  if (rc) c->user_keys.clear();
  return rc;
}

int hess_debug_key_levels(hess_ctx* c, const int* levels, int num) {
  if (!c || num < 0) return HESS_ERR_ARG;
  c->user_levels.clear();
  try {
    if (levels && num > 0) c->user_levels.assign(levels, levels + num);
  } catch (...) { c->user_levels.clear(); set_err(c, "out of host memory"); return HESS_ERR_NOMEM; }
  return 0;
}

int hess_count(hess_ctx* c, int img) {
  if (!c || img < 0 || img >= c->batch) return HESS_ERR_ARG;
  return c->counts[img];
}

int hess_desc_dim(hess_ctx* c) { return c ? c->dim : HESS_ERR_ARG; }

int hess_fetch(hess_ctx* c, int img, hess_keypoint* keys, float* desc) {
  if (!c || img < 0 || img >= c->batch) return HESS_ERR_ARG;
  const size_t n = (size_t)c->counts[img];
  if (c->user_result) {
    if (keys && n) memcpy(keys, c->u_keys.data(), n * sizeof(hess_keypoint));
    if (desc && c->dim && n) memcpy(desc, c->u_desc.data(), n * c->dim * 4);
    return 0;
  }
  if (keys && n) memcpy(keys, (HostKeypoint*)c->h_keys.p + c->offs[img], n * sizeof(HostKeypoint));
  if (desc && c->dim && n) memcpy(desc, (float*)c->h_desc.p + c->offs[img] * c->dim, n * c->dim * 4);
  return 0;
}

int hess_device_results(hess_ctx* c, const void** keys, const void** desc, int* capacity) {
  if (!c || !c->batch) return HESS_ERR_STATE;
  if (keys) *keys = c->keys.p;
  if (desc) *desc = c->dim ? c->desc.p : nullptr;
  if (capacity) *capacity = (int)c->offs[c->batch];  // records in use; image b starts at sum of counts < b
  return 0;
}

int hess_geometry(hess_ctx* c, int* widths, int* heights) {
  if (!c || !c->planned) return HESS_ERR_STATE;
  for (int o = 0; o < c->g.noct; o++) {
    if (widths) widths[o] = c->g.o[o].wa;
    if (heights) heights[o] = c->g.o[o].h;
  }
  return c->g.noct;
}

int hess_debug_level(hess_ctx* c, int img, int octave, int level, int what, float* out) {
  if (!c || !c->planned || !out || img < 0 || img >= c->batch || octave < 0 || octave >= c->g.noct || level < 0 ||
      level > c->sch.level_max)
    return HESS_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  const OctGeom& og = c->g.o[octave];
  if (what == HESS_DBG_GAUSS && !c->keep_levels &&
      ((level == c->sch.level_max && !c->no_top_fusion) || (level == 0 && octave == 0 && c->level0_in_lds))) {
    set_err(c, "this Gaussian level is not materialised (the octave's top level; level 0 of octave 0): call hess_debug_keep_levels before the run");
    return HESS_ERR_STATE;
  }
  if (what == HESS_DBG_GAUSS || what == HESS_DBG_DETH) {
    const float* base = (const float*)(what == HESS_DBG_GAUSS ? c->gauss.p : c->deth.p);
    const float* src = base + og.lvl_off + ((long long)level * c->g.B + img) * og.plane;
    HIP_TRY(c, hipMemcpy(out, src, (size_t)og.plane * 4, hipMemcpyDeviceToHost));
    return 0;
  }
  if (what == HESS_DBG_GOT) {
    if (level < 1 || level > c->g.dog) return HESS_ERR_ARG;
    const float* src = (const float*)c->got.p + 2 * (og.got_off + ((long long)(level - 1) * c->g.B + img) * og.plane);
    HIP_TRY(c, hipMemcpy(out, src, (size_t)og.plane * 8, hipMemcpyDeviceToHost));
    return 0;
  }
  return HESS_ERR_ARG;
}

int hess_debug_regrown(hess_ctx* c) { return c ? c->regrown : HESS_ERR_ARG; }

int hess_debug_keep_levels(hess_ctx* c, int on) {
  if (!c) return HESS_ERR_ARG;
  c->keep_levels = on != 0;
  return 0;
}

int hess_share_results(hess_ctx* c, const char* name) {
  if (!c) return HESS_ERR_ARG;
  if (!name || !name[0] || strlen(name) > 200 || strchr(name, '/')) {
    set_err(c, "hess_share_results: the name must be 1..200 characters without '/'");
    return HESS_ERR_ARG;
  }
  if (c->pend) { set_err(c, "a batch is in flight"); return HESS_ERR_ARG; }
  if (c->share_dir) { set_err(c, "the results of this context are shared already (as %s)", c->share.c_str()); return HESS_ERR_ARG; }
  HIP_TRY(c, hipSetDevice(c->device));
  char dir[256];
  snprintf(dir, sizeof(dir), "/%s.h", name);
  try {
    c->share = name;  // (nothing thrown crosses the C ABI)
  } catch (...) { set_err(c, "out of memory"); return HESS_ERR_NOMEM; }
  (void)shm_unlink(dir);
  for (unsigned gen = 1; gen <= 64; gen++) {  // buffers a crashed job of the same name left behind (the generations start at 1)
    char stale[256];
    snprintf(stale, sizeof(stale), "/%s.k%u", name, gen); (void)shm_unlink(stale);
    snprintf(stale, sizeof(stale), "/%s.d%u", name, gen); (void)shm_unlink(stale);
  }
  const int fd = shm_open(dir, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) { set_err(c, "shm_open(%s) failed: %s", dir, strerror(errno)); c->share.clear(); return HESS_ERR_NOMEM; }
  void* m = ftruncate(fd, 4096) == 0 ? mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
  close(fd);
  if (m == MAP_FAILED) { set_err(c, "cannot map %s: %s", dir, strerror(errno)); shm_unlink(dir); c->share.clear(); return HESS_ERR_NOMEM; }
  memset(m, 0, 4096);
  c->share_dir = static_cast<hess_ctx::ShareDir*>(m);
  c->share_dir->magic = 0x48455353u;  // "HESS"
  // results of an earlier run stay readable through hess_fetch only until the next run: the buffers move now
  if (c->st) HIP_TRY(c, hipStreamSynchronize(c->st));
  release(c->h_keys, true);
  release(c->h_desc, true);
  c->planned = false;
  c->batch = 0;
  return 0;
}

int hess_shared_results_info(hess_ctx* c, unsigned* gen_keys, unsigned* gen_desc, size_t* keys_bytes, size_t* desc_bytes) {
  if (!c || !c->share_dir) return HESS_ERR_ARG;
  if (gen_keys) *gen_keys = c->share_dir->gen_keys;
  if (gen_desc) *gen_desc = c->share_dir->gen_desc;
  if (keys_bytes) *keys_bytes = (size_t)c->share_dir->keys_bytes;
  if (desc_bytes) *desc_bytes = (size_t)c->share_dir->desc_bytes;
  return 0;
}

int hess_debug_list(hess_ctx* c, int img, hess_rawkey* out, int cap) {
  if (!c || !c->d_list || img < 0 || img >= c->batch) return HESS_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  int n = 0;
  HIP_TRY(c, hipMemcpy(&n, c->d_list_total + img, 4, hipMemcpyDeviceToHost));
  const int m = n < cap ? n : cap;
  static_assert(sizeof(hess_rawkey) == sizeof(RawKey), "raw key layout");
  if (out && m > 0)
    HIP_TRY(c, hipMemcpy(out, c->d_list + (size_t)img * c->cap_list, (size_t)m * sizeof(RawKey), hipMemcpyDeviceToHost));
  return n;
}

const float* hess_timing(hess_ctx* c) { return c ? c->timing : nullptr; }
const char* hess_last_error(hess_ctx* c) { return c ? c->err.c_str() : "null context"; }

int hess_profile_enable(hess_ctx* c, int on) { if (!c) return HESS_ERR_ARG; c->prof = on != 0; return 0; }
int hess_profile_reset(hess_ctx* c) {
  if (!c) return HESS_ERR_ARG;
  memset(c->k_ms, 0, sizeof(c->k_ms));
  memset(c->k_n, 0, sizeof(c->k_n));
  memset(c->k_bytes, 0, sizeof(c->k_bytes));
  memset(c->k_in_lds, 0, sizeof(c->k_in_lds));
  return 0;
}
int hess_profile_get(hess_ctx* c, int kernel, double* ms, long long* launches, double* bytes) {
  if (!c || kernel < 0 || kernel >= HESS_K_COUNT) return HESS_ERR_ARG;
  if (ms) *ms = c->k_ms[kernel];
  if (launches) *launches = c->k_n[kernel];
  if (bytes) *bytes = c->k_bytes[kernel];
  return 0;
}

int hess_profile_get_in_lds(hess_ctx* c, int kernel, double* bytes) {
  if (!c || !bytes || kernel < 0 || kernel >= HESS_K_COUNT) return HESS_ERR_ARG;
  *bytes = c->k_in_lds[kernel];
  return 0;
}

// Device evaluation of the elementary functions (tests only; see hess_devmath.h).
int hess_math_probe(hess_ctx* c, int which, const float* a, const float* b, float* out, int n) {
  if (!c || !a || !out || n <= 0) return HESS_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  HIP_TRY(c, hipMalloc(&da, (size_t)n * 4));
  HIP_TRY(c, hipMalloc(&db, (size_t)n * 4));
  HIP_TRY(c, hipMalloc(&dout, (size_t)n * 4));
  HIP_TRY(c, hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(db, b ? b : a, (size_t)n * 4, hipMemcpyHostToDevice));
  launch_math_probe(c->st, which, da, db, dout, n);
  HIP_TRY(c, hipStreamSynchronize(c->st));
  HIP_TRY(c, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return 0;
}

}  // extern "C"
