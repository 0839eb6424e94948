// speed -- repetition benchmark on the SiftGPU plugin surface, after the reference's harness
// (src/TestWin/speed.cpp:68-184): one warm-up run, N timed repetitions of RunSIFT on the same image,
// a '+' per repetition whose feature count equals the first run's and an 'e' otherwise (the only
// stability check the reference has), then the rate in Hz and Mpixel/s and the per-stage averages.
//   speed -i image.pgm [-n reps] [SiftGPU options]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "SiftGPU.h"

int main(int argc, char** argv) {
  int reps = 30;
  for (int i = 1; i + 1 < argc; i++)
    if (!strcmp(argv[i], "-n")) reps = atoi(argv[i + 1]);
  SiftGPU* sift = CreateNewSiftGPU(1);
  sift->ParseParam(argc - 1, argv + 1);
  char v0[] = "-v", v1[] = "0";
  char* quiet[] = {v0, v1};
  sift->ParseParam(2, quiet);
  if (sift->GetImageCount() < 1 || sift->CreateContextGL() != SiftGPU::SIFTGPU_FULL_SUPPORTED) {
    std::cerr << "speed -i image.pgm [-n reps] [sift params]\n";
    return EXIT_FAILURE;
  }
  if (!sift->RunSIFT(0)) return EXIT_FAILURE;  // warm-up: load, allocation
  const int n0 = sift->GetFeatureNum();
  double stage[TIMINGS_COUNT] = {0};
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) {
    sift->RunSIFT(0);
    std::cout << (sift->GetFeatureNum() == n0 ? '+' : 'e') << std::flush;
    for (int i = 0; i < TIMINGS_COUNT; i++) stage[i] += sift->_timing[i];
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << "\n" << n0 << " features, " << reps / sec << " Hz\n";
  static const char* names[TIMINGS_COUNT] = {"load", "allocate", "pyramid", "detect", "list", "orientation",
                                             "multi-orientation", "download", "descriptor", "vbo", "reduction", "total"};
  for (int i = 0; i < TIMINGS_COUNT; i++)
    if (stage[i] > 0) std::cout << "  " << names[i] << ":\t" << stage[i] / reps << " ms\n";
  delete sift;
  return EXIT_SUCCESS;
}
