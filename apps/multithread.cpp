// multithread -- one SiftGPU instance per host thread per device, the reference's multi-GPU usage pattern
// (src/TestWin/MultiThreadSIFT.cpp:83-156,231-244): every thread constructs and initialises its own instance under
// one process-wide mutex (the reference's options are process-global statics; this build keeps them per instance,
// the mutex is kept because that is what existing callers do), loads its image once, then repeats RunSIFT() on it
// without any lock and reports the rate.  Unlike the sample it runs on every visible device (or -devices N), can put
// several instances on one device (-per-device K) and checks that the feature count is the same in every repetition
// and in every thread that was given the same image.
//   multithread -i a.pgm [-i b.pgm ...] [-n reps] [-devices N] [-per-device K] [-mem] [SiftGPU options]
// Thread t works on device t / K with image t % (number of images).  Exit code 0 only if every thread succeeded.
// -mem: the in-memory calling pattern of the reference's harnesses (speed.cpp:107-124 with the image handed over as
// RunSIFT(w, h, data, GL_LUMINANCE, GL_UNSIGNED_BYTE), SiftGPU.cpp:248-290): every repetition hands the (binary PGM)
// pixels over again and copies the results out with GetFeatureVector; the rate is also printed in Mpixel/s.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "SiftGPU.h"
#include "hess_abi.h"

static std::mutex g_init_mutex;  // "siftgpu_initialize", MultiThreadSIFT.cpp:90
// The timed loops start together, once every thread is initialised (the reference starts each thread's clock as it
// gets there; a common start makes the summed rate the rate of the concurrent phase).
static std::atomic<int> g_ready{0};
static int g_threads = 0;
static std::chrono::steady_clock::time_point g_start;
static std::mutex g_end_mutex;
static std::chrono::steady_clock::time_point g_end;

// binary PGM (P5, 8 bit) into memory, for -mem
static bool read_pgm(const std::string& path, std::vector<unsigned char>& px, int& w, int& h) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0, 0, 0};
  int vals[3], got = 0;
  bool ok = fscanf(f, "%2s", magic) == 1 && !strcmp(magic, "P5");
  while (ok && got < 3) {
    int ch = fgetc(f);
    if (ch == '#') { while (ch != '\n' && ch != EOF) ch = fgetc(f); continue; }
    if (ch == EOF) { ok = false; break; }
    if (ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t') continue;
    ungetc(ch, f);
    ok = fscanf(f, "%d", &vals[got]) == 1;
    got++;
  }
  if (ok) {
    fgetc(f);  // the single whitespace byte before the raster
    w = vals[0]; h = vals[1];
    ok = w > 0 && h > 0 && vals[2] == 255;
    if (ok) { px.resize((size_t)w * h); ok = fread(px.data(), 1, px.size(), f) == px.size(); }
  }
  fclose(f);
  return ok;
}

struct Worker {
  int id = 0, device = 0, reps = 0;
  bool mem = false;
  double mpix = 0.0;
  std::string image;
  std::vector<char*> args;  // caller's SiftGPU options
  int features = -1;
  double hz = 0.0;
  bool ok = false;

  void run() {
    SiftGPU* sift = nullptr;
    std::vector<unsigned char> px;
    int w = 0, h = 0;
    if (mem && !read_pgm(image, px, w, h)) {
      fprintf(stderr, "#%d: -mem needs a binary 8-bit PGM: %s\n", id, image.c_str());
      g_ready.fetch_add(1);
      return;
    }
    const unsigned GL_LUMINANCE_ = 0x1909, GL_UNSIGNED_BYTE_ = 0x1401;
    {
      std::lock_guard<std::mutex> lock(g_init_mutex);
      sift = new SiftGPU;
      if (!args.empty()) sift->ParseParam((int)args.size(), args.data());
      std::string dev = std::to_string(device);
      char v0[] = "-v", v1[] = "0", c0[] = "-cuda";
      char* own[] = {v0, v1, c0, &dev[0]};
      sift->ParseParam(4, own);
      if (sift->CreateContextGL() != SiftGPU::SIFTGPU_FULL_SUPPORTED ||
          !(mem ? sift->RunSIFT(w, h, px.data(), GL_LUMINANCE_, GL_UNSIGNED_BYTE_) : sift->RunSIFT(image.c_str()))) {
        fprintf(stderr, "#%d: cannot initialise on device %d with %s\n", id, device, image.c_str());
        delete sift;
        g_ready.fetch_add(1);
        return;
      }
      features = sift->GetFeatureNum();
    }
    std::vector<SiftGPU::SiftKeypoint> keys(mem ? features + 1 : 0);
    std::vector<float> desc(mem ? (size_t)(features + 1) * 128 : 0);
    if (g_ready.fetch_add(1) + 1 == g_threads) g_start = std::chrono::steady_clock::now();
    while (g_ready.load() < g_threads) std::this_thread::yield();  // (a thread that failed above has counted itself too)
    const auto t0 = std::chrono::steady_clock::now();
    bool stable = true;
    for (int i = 0; i < reps; ++i) {
      if (mem) {
        stable = sift->RunSIFT(w, h, px.data(), GL_LUMINANCE_, GL_UNSIGNED_BYTE_) && sift->GetFeatureNum() == features && stable;
        if (stable) sift->GetFeatureVector(keys.data(), desc.data());
      } else {
        stable = sift->RunSIFT() && sift->GetFeatureNum() == features && stable;
      }
    }
    const auto t1 = std::chrono::steady_clock::now();
    { std::lock_guard<std::mutex> lock(g_end_mutex); if (t1 > g_end) g_end = t1; }
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    hz = reps / (sec > 0 ? sec : 1e-9);
    mpix = mem ? hz * (double)w * h / 1e6 : 0.0;
    ok = stable;
    {
      std::lock_guard<std::mutex> lock(g_init_mutex);
      printf("#%d: device %d, %s: %d features, %.1f Hz%s\n", id, device, image.c_str(), features, hz,
             stable ? "" : "  FEATURE COUNT CHANGED");
      delete sift;
    }
  }
};

int main(int argc, char** argv) {
  int reps = 100, devices = hess_device_count(), per_device = 1;
  bool mem = false;
  std::vector<std::string> images;
  std::vector<char*> pass;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-i") && i + 1 < argc) images.push_back(argv[++i]);
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-devices") && i + 1 < argc) devices = std::min(devices, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-per-device") && i + 1 < argc) per_device = std::max(1, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-mem")) mem = true;
    else pass.push_back(argv[i]);
  }
  if (images.empty() || devices < 1) {
    fprintf(stderr, devices < 1 ? "multithread: no HIP device\n"
                                : "multithread -i a.pgm [-i b.pgm ...] [-n reps] [-devices N] [-per-device K] [sift params]\n");
    return EXIT_FAILURE;
  }
  const int nthreads = devices * per_device;
  printf("Starting %d thread(s) on %d device(s)...\n", nthreads, devices);
  g_threads = nthreads;
  std::vector<Worker> workers(nthreads);
  std::vector<std::thread> threads;
  for (int t = 0; t < nthreads; t++) {
    Worker& w = workers[t];
    w.id = t; w.device = t / per_device; w.reps = reps; w.image = images[t % images.size()]; w.args = pass; w.mem = mem;
    threads.emplace_back(&Worker::run, &w);
  }
  for (auto& th : threads) th.join();
  bool ok = true;
  double total_hz = 0, total_mpix = 0;
  for (const Worker& w : workers) {
    ok = ok && w.ok;
    total_hz += w.hz;
    total_mpix += w.mpix;
    for (const Worker& v : workers) ok = ok && (v.image != w.image || v.features == w.features);
  }
  printf("%s: %.1f images/s over all threads\n", ok ? "OK" : "FAILED", total_hz);
  if (mem) {
    // all repetitions of all threads over the wall time from the common start to the last thread's finish
    const double wall = std::chrono::duration<double>(g_end - g_start).count();
    const double per_image = total_hz > 0 ? total_mpix / total_hz : 0.0;  // Mpixel per image
    printf("MPIX: %.1f Mpixel/s over all threads (RunSIFT(w,h,data) + GetFeatureVector, %d thread(s), %d repetitions each, %.3f s)\n",
           wall > 0 ? per_image * (double)reps * nthreads / wall : 0.0, nthreads, reps, wall);
  }
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
