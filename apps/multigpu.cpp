// multigpu -- the C++ host path of a whole node: one host thread per GPU, every GPU runs its share of a batch of images
// through the C ABI (include/hess_abi.h), and the feature lists are gathered onto device 0 with RCCL -- exact-size
// ncclSend / ncclRecv pairs over xGMI straight out of the contexts' packed device-resident result buffers
// (hess_device_results): the only exchange of the path, as BASELINE.json's north star asks ("host code stays C++ ...
// RCCL over xGMI only to gather the final feature list").  The reference's own multi-GPU sample stops before that
// step: it runs one instance per device thread and leaves every list where it is (src/TestWin/MultiThreadSIFT.cpp:
// 83-156,231-244; apps/multithread.cpp is that sample).  The Python harness does the same gather with
// torch.distributed (hessgpu_amd/dist.py); this is it for a C++ caller.
//
//   multigpu -i a.pgm [-i b.pgm ...] [-devices N] [-batch B] [-n steps] [-topk K]
//
// Device d works on images d*B .. d*B+B-1 of the global batch (the given files, cycled).  Per step: every device runs
// its batch (hess_run_device), posts its per-image counts to the host table (a few integers: the counts are in host
// memory anyway), and inside one ncclGroup device 0 receives `n_r x 24` bytes of keypoints and `n_r x dim` floats of
// descriptors from every other device r while those send exactly that.  After the last step device 0's gathered copy
// is compared with what every device delivered to its own host (hess_fetch).  With one device there is nothing to
// send; the table, the group and the check still run (the only form testable on a one-GPU box).
// Exit code 0 only if every device ran and the gathered lists equal the devices' own.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hess_abi.h"

namespace {

struct Barrier {  // all device threads, reusable
  std::mutex mu;
  std::condition_variable cv;
  int n, waiting = 0;
  unsigned long long gen = 0;
  explicit Barrier(int count) : n(count) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long long g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

bool read_pgm(const std::string& path, std::vector<unsigned char>& px, int& w, int& h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  int maxv = 0;
  bool ok = fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && !strcmp(magic, "P5") && w > 0 && h > 0 && maxv == 255;
  if (ok) {
    fgetc(f);  // the single whitespace byte before the raster
    px.resize((size_t)w * h);
    ok = fread(px.data(), 1, px.size(), f) == px.size();
  }
  fclose(f);
  return ok;
}

struct Shared {
  int ndev = 0, batch = 0, steps = 0, w = 0, h = 0, dim = 0, topk = 0;
  std::vector<std::vector<unsigned char>> images;  // the global batch's distinct images
  std::vector<ncclComm_t> comms;
  std::vector<int> totals;                          // features per device in the current step
  std::vector<std::vector<int>> counts;             // per device, per image
  std::vector<std::vector<hess_keypoint>> host_keys;  // per device: its own host results of the last step (the check)
  std::vector<std::vector<float>> host_desc;
  // failure flags, read only after the barrier that follows their writes: `failed` before the gather (set before the
  // counts barrier), `failed_after` by the gather itself (set before the end-of-step barrier) -- two arrays, so that a
  // slow thread still reading one never sees a fast thread's later write
  std::vector<int> failed, failed_after;
  std::vector<int> result;  // per device, written once when its thread is done; read after the joins
  double seconds = 0.0;
  long long gathered_features = 0;
  bool gather_ok = true;
  Barrier* bar = nullptr;
};

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "multigpu: device %d: %s failed: %s\n", dev, #expr, hipGetErrorString(e_)); fail = true; } } while (0)
#define NCCL_OK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { fprintf(stderr, "multigpu: device %d: %s failed: %s\n", dev, #expr, ncclGetErrorString(r_)); fail = true; } } while (0)

void device_thread(Shared* S, int dev) {
  bool fail = false;
  HIP_OK(hipSetDevice(dev));
  hipStream_t st = nullptr;
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hess_params p;
  hess_default_params(&p);
  if (S->topk > 0) { p.truncate_method = HESS_TRUNC_TOPK; p.feature_count_threshold = S->topk; }
  hess_ctx* ctx = fail ? nullptr : hess_create(dev, &p);
  if (!ctx) fail = true;
  const int B = S->batch, w = S->w, h = S->h;
  const size_t img_bytes = (size_t)w * h;
  unsigned char* d_px = nullptr;
  if (!fail) HIP_OK(hipMalloc(&d_px, img_bytes * B));
  for (int b = 0; b < B && !fail; b++) {
    const auto& img = S->images[(size_t)(dev * B + b) % S->images.size()];
    HIP_OK(hipMemcpy(d_px + b * img_bytes, img.data(), img_bytes, hipMemcpyHostToDevice));
  }
  if (!fail && hess_reserve(ctx, w, h, B) != 0) { fprintf(stderr, "multigpu: device %d: %s\n", dev, hess_last_error(ctx)); fail = true; }
  // device 0: landing buffers for the other devices' lists, grown on demand
  std::vector<void*> land_keys(S->ndev, nullptr), land_desc(S->ndev, nullptr);
  std::vector<size_t> land_cap(S->ndev, 0);
  S->failed_after[dev] = fail;  // (set-up counts as the end of a step before the first)
  S->bar->wait();
  bool any_failed = false;
  for (int f : S->failed_after) any_failed = any_failed || f;
  const auto t0 = std::chrono::steady_clock::now();
  for (int step = 0; step < S->steps && !any_failed; step++) {
    const void *dk = nullptr, *dd = nullptr;
    int total = 0;
    if (hess_run_device(ctx, d_px, w, h, w, img_bytes, B, HESS_FMT_LUM, HESS_PIX_U8) != 0 ||
        hess_device_results(ctx, &dk, &dd, &total) != 0) {
      fprintf(stderr, "multigpu: device %d: %s\n", dev, hess_last_error(ctx));
      fail = true;
    }
    for (int b = 0; b < B; b++) S->counts[dev][b] = fail ? 0 : hess_count(ctx, b);
    S->totals[dev] = fail ? 0 : total;
    if (dev == 0) S->dim = hess_desc_dim(ctx);  // (same parameters on every device: written once)
    S->failed[dev] = fail;
    S->bar->wait();  // every device's counts are in the table
    for (int f : S->failed) any_failed = any_failed || f;
    if (any_failed) break;
    const int dim = S->dim;
    if (dev == 0) {
      for (int r = 1; r < S->ndev; r++) {
        const size_t n = (size_t)S->totals[r];
        if (n > land_cap[r]) {
          if (land_keys[r]) { HIP_OK(hipFree(land_keys[r])); HIP_OK(hipFree(land_desc[r])); }
          land_cap[r] = n + n / 4 + 64;
          HIP_OK(hipMalloc(&land_keys[r], land_cap[r] * sizeof(hess_keypoint)));
          HIP_OK(hipMalloc(&land_desc[r], land_cap[r] * (size_t)(dim ? dim : 1) * sizeof(float)));
        }
      }
    }
    // the gather: exact sizes, nothing for an empty list, device 0's own block stays where it is
    NCCL_OK(ncclGroupStart());
    if (dev == 0) {
      for (int r = 1; r < S->ndev; r++) {
        const size_t n = (size_t)S->totals[r];
        if (!n) continue;
        NCCL_OK(ncclRecv(land_keys[r], n * sizeof(hess_keypoint), ncclUint8, r, S->comms[0], st));
        if (dim) NCCL_OK(ncclRecv(land_desc[r], n * (size_t)dim, ncclFloat, r, S->comms[0], st));
      }
    } else if (total > 0) {
      NCCL_OK(ncclSend(dk, (size_t)total * sizeof(hess_keypoint), ncclUint8, 0, S->comms[dev], st));
      if (dim) NCCL_OK(ncclSend(dd, (size_t)total * (size_t)dim, ncclFloat, 0, S->comms[dev], st));
    }
    NCCL_OK(ncclGroupEnd());
    HIP_OK(hipStreamSynchronize(st));
    S->failed_after[dev] = fail;
    if (step == S->steps - 1 && !fail) {  // the last step's own host results, for the check below
      S->host_keys[dev].resize((size_t)total + 1);
      S->host_desc[dev].resize(((size_t)total + 1) * (size_t)(dim ? dim : 1));
      size_t at = 0;
      for (int b = 0; b < B; b++) {
        hess_fetch(ctx, b, S->host_keys[dev].data() + at, dim ? S->host_desc[dev].data() + at * dim : nullptr);
        at += (size_t)S->counts[dev][b];
      }
    }
    S->bar->wait();  // the step is over on every device (and the contexts may overwrite their result buffers)
    for (int f : S->failed_after) any_failed = any_failed || f;  // every thread sees the same flags: all leave together
  }
  if (dev == 0 && !any_failed) {
    S->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // what device 0 holds against what every device delivered to its own host
    long long sum = S->totals[0];
    for (int r = 1; r < S->ndev; r++) {
      const size_t n = (size_t)S->totals[r];
      sum += (long long)n;
      if (!n) continue;
      std::vector<hess_keypoint> k(n);
      std::vector<float> d(n * (size_t)(S->dim ? S->dim : 1));
      HIP_OK(hipMemcpy(k.data(), land_keys[r], n * sizeof(hess_keypoint), hipMemcpyDeviceToHost));
      if (S->dim) HIP_OK(hipMemcpy(d.data(), land_desc[r], n * (size_t)S->dim * sizeof(float), hipMemcpyDeviceToHost));
      if (memcmp(k.data(), S->host_keys[r].data(), n * sizeof(hess_keypoint)) != 0 ||
          (S->dim && memcmp(d.data(), S->host_desc[r].data(), n * (size_t)S->dim * sizeof(float)) != 0))
        S->gather_ok = false;
    }
    S->gathered_features = sum;
    if (fail) S->gather_ok = false;
  }
  for (int r = 0; r < S->ndev; r++) if (land_keys[r]) { (void)hipFree(land_keys[r]); (void)hipFree(land_desc[r]); }
  if (d_px) (void)hipFree(d_px);
  if (ctx) hess_destroy(ctx);
  if (st) (void)hipStreamDestroy(st);
  S->result[dev] = fail;
}

}  // namespace

int main(int argc, char** argv) {
  Shared S;
  S.ndev = hess_device_count();
  S.batch = 8; S.steps = 10; S.topk = 4096;
  std::vector<std::string> files;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-i") && i + 1 < argc) files.push_back(argv[++i]);
    else if (!strcmp(argv[i], "-devices") && i + 1 < argc) S.ndev = std::min(S.ndev, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-batch") && i + 1 < argc) S.batch = std::max(1, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) S.steps = std::max(1, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-topk") && i + 1 < argc) S.topk = atoi(argv[++i]);
  }
  if (files.empty() || S.ndev < 1) {
    fprintf(stderr, S.ndev < 1 ? "multigpu: no HIP device\n" : "multigpu -i a.pgm [-i b.pgm ...] [-devices N] [-batch B] [-n steps] [-topk K]\n");
    return EXIT_FAILURE;
  }
  for (const std::string& f : files) {
    std::vector<unsigned char> px;
    int w = 0, h = 0;
    if (!read_pgm(f, px, w, h) || (S.w && (w != S.w || h != S.h))) {
      fprintf(stderr, "multigpu: %s: need binary 8-bit PGMs of one size\n", f.c_str());
      return EXIT_FAILURE;
    }
    S.w = w; S.h = h;
    S.images.push_back(std::move(px));
  }
  std::vector<int> devs(S.ndev);
  for (int d = 0; d < S.ndev; d++) devs[d] = d;
  S.comms.resize(S.ndev);
  if (ncclCommInitAll(S.comms.data(), S.ndev, devs.data()) != ncclSuccess) {
    fprintf(stderr, "multigpu: ncclCommInitAll failed\n");
    return EXIT_FAILURE;
  }
  S.totals.assign(S.ndev, 0);
  S.counts.assign(S.ndev, std::vector<int>(S.batch, 0));
  S.host_keys.resize(S.ndev); S.host_desc.resize(S.ndev);
  S.failed.assign(S.ndev, 0);
  S.failed_after.assign(S.ndev, 0);
  S.result.assign(S.ndev, 0);
  Barrier bar(S.ndev);
  S.bar = &bar;
  printf("multigpu: %d device(s), %d image(s) of %dx%d per device and step, %d steps, top-K %d\n", S.ndev, S.batch, S.w, S.h,
         S.steps, S.topk);
  std::vector<std::thread> threads;
  for (int d = 0; d < S.ndev; d++) threads.emplace_back(device_thread, &S, d);
  for (auto& t : threads) t.join();
  for (int d = 0; d < S.ndev; d++) ncclCommDestroy(S.comms[d]);
  bool ok = S.gather_ok;
  for (int f : S.result) ok = ok && !f;
  if (ok) {
    for (int d = 0; d < S.ndev; d++) {
      printf("#%d:", d);
      for (int b = 0; b < S.batch; b++) printf(" %d", S.counts[d][b]);
      printf(" features\n");
    }
    const double mpix = (double)S.ndev * S.batch * S.steps * S.w * S.h / 1e6;
    printf("GATHER OK: %lld features of %d images on device 0 per step; %.1f Mpixel/s over %d device(s) (%.3f s, synchronous steps)\n",
           S.gathered_features, S.ndev * S.batch, S.seconds > 0 ? mpix / S.seconds : 0.0, S.ndev, S.seconds);
  } else {
    printf("FAILED\n");
  }
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
