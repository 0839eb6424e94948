// multithread -- one SiftGPU instance per host thread per device, the reference's multi-GPU usage pattern
// (src/TestWin/MultiThreadSIFT.cpp:83-156,231-244): every thread constructs and initialises its own instance under
// one process-wide mutex (the reference's options are process-global statics; this build keeps them per instance,
// the mutex is kept because that is what existing callers do), loads its image once, then repeats RunSIFT() on it
// without any lock and reports the rate.  Unlike the sample it runs on every visible device (or -devices N), can put
// several instances on one device (-per-device K) and checks that the feature count is the same in every repetition
// and in every thread that was given the same image.
//   multithread -i a.pgm [-i b.pgm ...] [-n reps] [-devices N] [-per-device K] [SiftGPU options]
// Thread t works on device t / K with image t % (number of images).  Exit code 0 only if every thread succeeded.
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

struct Worker {
  int id = 0, device = 0, reps = 0;
  std::string image;
  std::vector<char*> args;  // caller's SiftGPU options
  int features = -1;
  double hz = 0.0;
  bool ok = false;

  void run() {
    SiftGPU* sift = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_init_mutex);
      sift = new SiftGPU;
      if (!args.empty()) sift->ParseParam((int)args.size(), args.data());
      std::string dev = std::to_string(device);
      char v0[] = "-v", v1[] = "0", c0[] = "-cuda";
      char* own[] = {v0, v1, c0, &dev[0]};
      sift->ParseParam(4, own);
      if (sift->CreateContextGL() != SiftGPU::SIFTGPU_FULL_SUPPORTED || !sift->RunSIFT(image.c_str())) {
        fprintf(stderr, "#%d: cannot initialise on device %d with %s\n", id, device, image.c_str());
        delete sift;
        return;
      }
      features = sift->GetFeatureNum();
    }
    const auto t0 = std::chrono::steady_clock::now();
    bool stable = true;
    for (int i = 0; i < reps; ++i) stable = sift->RunSIFT() && sift->GetFeatureNum() == features && stable;
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hz = reps / (sec > 0 ? sec : 1e-9);
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
  std::vector<std::string> images;
  std::vector<char*> pass;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-i") && i + 1 < argc) images.push_back(argv[++i]);
    else if (!strcmp(argv[i], "-n") && i + 1 < argc) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-devices") && i + 1 < argc) devices = std::min(devices, atoi(argv[++i]));
    else if (!strcmp(argv[i], "-per-device") && i + 1 < argc) per_device = std::max(1, atoi(argv[++i]));
    else pass.push_back(argv[i]);
  }
  if (images.empty() || devices < 1) {
    fprintf(stderr, devices < 1 ? "multithread: no HIP device\n"
                                : "multithread -i a.pgm [-i b.pgm ...] [-n reps] [-devices N] [-per-device K] [sift params]\n");
    return EXIT_FAILURE;
  }
  const int nthreads = devices * per_device;
  printf("Starting %d thread(s) on %d device(s)...\n", nthreads, devices);
  std::vector<Worker> workers(nthreads);
  std::vector<std::thread> threads;
  for (int t = 0; t < nthreads; t++) {
    Worker& w = workers[t];
    w.id = t; w.device = t / per_device; w.reps = reps; w.image = images[t % images.size()]; w.args = pass;
    threads.emplace_back(&Worker::run, &w);
  }
  for (auto& th : threads) th.join();
  bool ok = true;
  double total_hz = 0;
  for (const Worker& w : workers) {
    ok = ok && w.ok;
    total_hz += w.hz;
    for (const Worker& v : workers) ok = ok && (v.image != w.image || v.features == w.features);
  }
  printf("%s: %.1f images/s over all threads\n", ok ? "OK" : "FAILED", total_hz);
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
