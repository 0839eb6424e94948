// speed -- repetition benchmark on the SiftGPU plugin surface, after the reference's harness
// (src/TestWin/speed.cpp:68-184): warm-up runs, then TWO passes of N repetitions of RunSIFT() on the same image:
//   pass 1  SetVerbose(0): no messages, no stage timers -- the rate.  A '+' per repetition whose feature count
//           equals the warm-up run's, an 'e' otherwise (the only stability check the reference has);
//   pass 2  SetVerbose(-2) (speed.cpp:128: "disable all output but keep the timing"): stage timers on -- here they
//           are events between the stages, about 6 us each on the device, as the reference's own comment says of its
//           pass ("the overall speed will be decreased") -- a '#' per repetition, then the per-stage averages.
//   speed -i image.pgm [-n reps] [SiftGPU options]
// (Units: this build's _timing[] is in milliseconds; the reference keeps seconds and multiplies by 1000 when printing.)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "SiftGPU.h"

int main(int argc, char** argv) {
  int reps = 30;  // SIFTGPU_REPEAT, speed.cpp:60
  for (int i = 1; i + 1 < argc; i++)
    if (!strcmp(argv[i], "-n")) reps = atoi(argv[i + 1]);
  if (reps < 1) reps = 1;
  SiftGPU* sift = CreateNewSiftGPU(1);
  sift->ParseParam(argc - 1, argv + 1);
  sift->SetVerbose(0);
  std::cout << "Initialize and warm up...\n";
  if (sift->GetImageCount() < 1 || sift->CreateContextGL() != SiftGPU::SIFTGPU_FULL_SUPPORTED) {
    std::cerr << "speed -i image.pgm [-n reps] [sift params]\n";
    return EXIT_FAILURE;
  }
  if (!sift->RunSIFT(0)) return EXIT_FAILURE;  // loads the image (once for this experiment), allocates
  std::cout << "Loading image: " << sift->_timing[0] << "ms, Tex initialization: " << sift->_timing[1] << "ms\n\n"
            << "Start 2x" << reps << " repetitions...\n";
  sift->RunSIFT();  // "run one more time to get all texture allocated"
  const int num = sift->GetFeatureNum();

  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) {
    sift->RunSIFT();
    std::cout << (sift->GetFeatureNum() == num ? '+' : 'e') << std::flush;
  }
  const double time_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
  std::cout << "\n";

  // stage timers on: more accurate per-step times, lower overall speed
  sift->SetVerbose(-2);
  double timing[TIMINGS_COUNT] = {0};
  t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < reps; k++) {
    sift->RunSIFT();
    for (int j = 0; j < TIMINGS_COUNT; j++) timing[j] += sift->_timing[j];
    std::cout << (sift->GetFeatureNum() == num ? '#' : 'e') << std::flush;
  }
  const double time_staged = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
  for (int j = 0; j < TIMINGS_COUNT; j++) timing[j] /= reps;

  std::cout << "\n\n****************************************\n"
            << "[Feature Count]:\t" << num << "\n"
            << "[Average Time]:\t\t" << time_all * 1000.0 << "ms\n"
            << "[Average Speed]:\t" << 1.0 / time_all << "hz\n"
            << "[Build Pyramid]:\t" << timing[TIMINGS_BUILD_PYRAMID] << "ms\n"
            << "[Detection]:\t\t" << timing[TIMINGS_DETECT_KEYPOINTS] << "ms\n"
            << "[Feature List]:\t\t" << timing[TIMINGS_GENERATE_FEATURE_LIST] << "ms\n"
            << "[Orientation]:\t\t" << timing[TIMINGS_COMPUTE_ORIENTATIONS] << "ms\n"
            << "[MO Feature List]:\t" << timing[TIMINGS_MULTI_ORIENTATIONS] << "ms\n"
            << "[Download Keys]:\t" << timing[TIMINGS_DOWNLOAD_KEYPOINTS] << "ms\n"
            << "[Descriptor]:\t\t" << timing[TIMINGS_COMPUTE_DESCRIPTORS] << "ms\n"
            << "[Top-K Reduction]:\t" << timing[TIMINGS_FEATURES_REDUCTION] << "ms\n"
            << "[With stage timers]:\t" << time_staged * 1000.0 << "ms per image\n"
            << "****************************************\n";
  delete sift;
  return EXIT_SUCCESS;
}
