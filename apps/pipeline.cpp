// pipeline -- bench.py's step loop without Python: C contexts of one device used round-robin through the C ABI
// (hess_submit_device / hess_wait, include/hess_abi.h), batches of B images resident in HBM, results delivered to host
// memory.  It answers one question: are the gaps in the pipelined device's timeline the submitting
// host loop's?  Same work per step as `bench.py --gpus 1` (the reference's callers are C++: hessgpucmd.cpp:130-173 times
// its loop the same way -- steady state, allocation excluded).
//
//   pipeline -i a.pgm [-i b.pgm ...] [-batch B] [-contexts C] [-n steps] [-topk K] [-device d]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hess_abi.h"

static bool read_pgm(const std::string& path, std::vector<unsigned char>& px, int& w, int& h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  int maxv = 0;
  bool ok = fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && !strcmp(magic, "P5") && w > 0 && h > 0 && maxv == 255;
  if (ok) {
    fgetc(f);
    px.resize((size_t)w * h);
    ok = fread(px.data(), 1, px.size(), f) == px.size();
  }
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  int batch = 8, nctx = 7, steps = 200, topk = 4096, dev = 0;
  std::vector<std::string> files;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-i") && i + 1 < argc) files.push_back(argv[++i]);
    else if (!strcmp(argv[i], "-batch") && i + 1 < argc) batch = std::max(1, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-contexts") && i + 1 < argc) nctx = std::max(1, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) steps = std::max(1, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-topk") && i + 1 < argc) topk = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-device") && i + 1 < argc) dev = atoi(argv[++i]);
  }
  if (files.empty()) { fprintf(stderr, "pipeline -i a.pgm [-i b.pgm ...] [-batch B] [-contexts C] [-n steps] [-topk K] [-device d]\n"); return EXIT_FAILURE; }
  int w = 0, h = 0;
  std::vector<std::vector<unsigned char>> images;
  for (const std::string& f : files) {
    std::vector<unsigned char> px;
    int iw = 0, ih = 0;
    if (!read_pgm(f, px, iw, ih) || (w && (iw != w || ih != h))) { fprintf(stderr, "pipeline: %s: need binary 8-bit PGMs of one size\n", f.c_str()); return EXIT_FAILURE; }
    w = iw; h = ih;
    images.push_back(std::move(px));
  }
  if (hipSetDevice(dev) != hipSuccess) { fprintf(stderr, "pipeline: no HIP device %d\n", dev); return EXIT_FAILURE; }
  const size_t img_bytes = (size_t)w * h;
  unsigned char* d_px = nullptr;
  if (hipMalloc(&d_px, img_bytes * batch) != hipSuccess) return EXIT_FAILURE;
  for (int b = 0; b < batch; b++)
    if (hipMemcpy(d_px + b * img_bytes, images[(size_t)b % images.size()].data(), img_bytes, hipMemcpyHostToDevice) != hipSuccess) return EXIT_FAILURE;
  hess_params p;
  hess_default_params(&p);
  if (topk > 0) { p.truncate_method = HESS_TRUNC_TOPK; p.feature_count_threshold = topk; }
  std::vector<hess_ctx*> ctx(nctx, nullptr);
  for (int c = 0; c < nctx; c++) {
    ctx[c] = hess_create(dev, &p);
    if (!ctx[c] || hess_reserve(ctx[c], w, h, batch) != 0 ||
        hess_run_device(ctx[c], d_px, w, h, w, img_bytes, batch, HESS_FMT_LUM, HESS_PIX_U8) != 0) {
      fprintf(stderr, "pipeline: context %d: %s\n", c, ctx[c] ? hess_last_error(ctx[c]) : "hess_create failed");
      return EXIT_FAILURE;
    }
  }
  auto run = [&](int n) {
    std::vector<int> inflight;
    size_t head = 0;
    for (int i = 0; i < n; i++) {
      const int c = i % nctx;
      if ((int)(inflight.size() - head) == nctx) { if (hess_wait(ctx[inflight[head++]]) != 0) return false; }
      if (hess_submit_device(ctx[c], d_px, w, h, w, img_bytes, batch, HESS_FMT_LUM, HESS_PIX_U8) != 0) return false;
      inflight.push_back(c);
    }
    while (head < inflight.size()) if (hess_wait(ctx[inflight[head++]]) != 0) return false;
    return true;
  };
  bool ok = run(2 * nctx);
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  ok = ok && run(steps);
  (void)hipDeviceSynchronize();
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long long feats = 0;
  for (int b = 0; b < batch; b++) feats += hess_count(ctx[(steps - 1) % nctx], b);
  if (!ok) { fprintf(stderr, "pipeline: %s\n", hess_last_error(ctx[0])); return EXIT_FAILURE; }
  printf("PIPELINE: %d steps of %d images %dx%d, %d contexts: %.3f ms per step, MPIX: %.1f (%lld features in the last batch)\n", steps, batch, w, h,
         nctx, s * 1e3 / steps, (double)steps * batch * w * h / s / 1e6, feats);
  for (hess_ctx* c : ctx) hess_destroy(c);
  (void)hipFree(d_px);
  return EXIT_SUCCESS;
}
