// hess -- batch command-line driver on the SiftGPU plugin surface (MI355X build).
//
// Same observable behaviour as the reference's `hess` tool (src/HessGPU/hessgpucmd.cpp:24-305):
//   hess -i <images..> | -il <listfile> [-time] [-speed] [SiftGPU options]
// For every image: RunSIFT(index), write <image>.sift (text unless -b / -bvlf); with -time write
// <image>.timings (11 comma-separated stage times in ms, the order of hessgpucmd.cpp:246-300) and
// silence stdout; with -speed repeat each image 10 times and report averages (load and allocation
// counted once).  Images: PGM / PPM built in, PNG and JPEG through libpng16 / libjpeg looked up at run time (this build
// has no DevIL).  The files of the next list entries are decoded on host threads while an image runs
// (hessgpu_amd/csrc/siftgpu_api.cpp, decode_ahead).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>

#include "SiftGPU.h"

static const int kSpeedIterations = 10;  // SPEED_TEST_NUM_ITERATIONS

int main(int argc, char** argv) {
  SiftGPU sift;
  sift.ParseParam(argc, argv);
  char a0[] = "-cuda", a1[] = "0", a2[] = "-nogl", a3[] = "-v", a4[] = "1";
  char* local[] = {a0, a1, a2, a3, a4};  // as the reference: device 0, -nogl is ignored, brief output
  sift.ParseParam(5, local);

  if (sift.GetImageCount() < 1) {
    std::cout << "hess -i <list of image names> | -il <file with image names> [-time] [-speed] [sift params list]\n\n"
                 "-time                write <image>.timings and suppress output (except warnings and errors)\n"
                 "-speed               speed test - average of 10 runs\n"
                 "[sift params list]   use option -h to get a list of these params\n";
    return EXIT_FAILURE;
  }
  bool export_timings = false, save_output = true, speed_test = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-time")) export_timings = true;
    else if (!strcmp(argv[i], "-speed")) speed_test = true;
    else if (!strcmp(argv[i], "-o")) save_output = sift.GetImageCount() > 1;
  }
  if (sift.CreateContextGL() != SiftGPU::SIFTGPU_FULL_SUPPORTED) return EXIT_FAILURE;
  if (export_timings) sift.SetVerbose(-2);

  static const int order[11] = {TIMINGS_LOAD_IMAGE, TIMINGS_ALLOCATE_PYRAMID, TIMINGS_BUILD_PYRAMID,
                                TIMINGS_DETECT_KEYPOINTS, TIMINGS_GENERATE_FEATURE_LIST, TIMINGS_FEATURES_REDUCTION,
                                TIMINGS_COMPUTE_ORIENTATIONS, TIMINGS_MULTI_ORIENTATIONS, TIMINGS_DOWNLOAD_KEYPOINTS,
                                TIMINGS_COMPUTE_DESCRIPTORS, TIMINGS_TOTAL};
  for (int idx = 0; idx < sift.GetImageCount(); idx++) {
    if (!sift.RunSIFT(idx)) continue;
    double acc[TIMINGS_COUNT];
    for (int i = 0; i < TIMINGS_COUNT; i++) acc[i] = sift._timing[i];
    if (speed_test) {
      for (int it = 1; it < kSpeedIterations; it++) {
        sift.RunSIFT(idx);
        for (int i = 0; i < 11; i++) acc[i] += sift._timing[i];
        acc[TIMINGS_TOTAL] += sift._timing[TIMINGS_TOTAL] + acc[TIMINGS_LOAD_IMAGE] + acc[TIMINGS_ALLOCATE_PYRAMID];
      }
      for (int i = 2; i < 12; i++) acc[i] /= kSpeedIterations;  // load and allocation happen once
    }
    const std::string img = sift.GetCurrentImagePath();
    if (save_output) sift.SaveSIFT((img + ".sift").c_str());
    if (export_timings) {
      std::ofstream out((img + ".timings").c_str());
      out.flags(std::ios::fixed);
      for (int k = 0; k < 11; k++) {
        if (speed_test) out << std::setprecision(2) << acc[order[k]];
        else out << sift._timing[order[k]];
        out << (k < 10 ? ", " : "");
      }
      out << std::endl;
    }
  }
  return EXIT_SUCCESS;
}
