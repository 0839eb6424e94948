// Caller of the SiftGPU plugin surface used by tests/test_abi_vs_reference_header.py.  It is compiled twice --
// once against the reference's own header (-I /root/reference/src/SiftGPU) and once against include/SiftGPU.h --
// and both binaries are linked to this build's libsiftgpu.so.  Every call below goes through what the header it
// was compiled with says about object layout, vtable order, default arguments and mangled names, so the binary
// built from the reference's header is "an existing caller, unchanged".  Usage pattern of the callers in the
// reference: stack object (HessGPU/hessgpucmd.cpp:27), factory (TestWin/SimpleSIFT.cpp:88-202), direct reads of
// _timing (hessgpucmd.cpp:96-97, TestWin/speed.cpp:98).
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "SiftGPU.h"

struct Probe : public SiftGPU {  // protected members are part of the layout a stack-allocating caller relies on
  static void layout() {
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
    printf("offsetof _sigma %zu\n", offsetof(Probe, _sigma));
    printf("offsetof _sigma0 %zu\n", offsetof(Probe, _sigma0));
    printf("offsetof _dog_level_num %zu\n", offsetof(Probe, _dog_level_num));
    printf("offsetof _level_ds %zu\n", offsetof(Probe, _level_ds));
    printf("offsetof _edge_threshold %zu\n", offsetof(Probe, _edge_threshold));
    printf("offsetof _current %zu\n", offsetof(Probe, _current));
    printf("offsetof _initialized %zu\n", offsetof(Probe, _initialized));
    printf("offsetof _image_loaded %zu\n", offsetof(Probe, _image_loaded));
    printf("offsetof _imgpath %zu\n", offsetof(Probe, _imgpath));
    printf("offsetof _outpath %zu\n", offsetof(Probe, _outpath));
    printf("offsetof _list %zu\n", offsetof(Probe, _list));
    printf("offsetof _texImage %zu\n", offsetof(Probe, _texImage));
    printf("offsetof _pyramid %zu\n", offsetof(Probe, _pyramid));
    printf("offsetof _timing %zu\n", offsetof(Probe, _timing));
#pragma GCC diagnostic pop
  }
};

int main() {
  printf("sizeof SiftParam %zu\n", sizeof(SiftParam));
  printf("sizeof SiftGPU %zu\n", sizeof(SiftGPU));
  printf("sizeof SiftKeypoint %zu\n", sizeof(SiftGPU::SiftKeypoint));
  printf("sizeof SiftMatchGPU %zu\n", sizeof(SiftMatchGPU));
  printf("sizeof ComboSiftGPU %zu\n", sizeof(ComboSiftGPU));
  printf("sizeof timing %zu\n", sizeof(((SiftGPU*)0)->_timing));
  printf("SIFT_KEYPOINT_ITEMS %d\n", (int)SIFT_KEYPOINT_ITEMS);
  printf("enum support %d %d %d\n", (int)SiftGPU::SIFTGPU_NOT_SUPPORTED, (int)SiftGPU::SIFTGPU_PARTIAL_SUPPORTED,
         (int)SiftGPU::SIFTGPU_FULL_SUPPORTED);
  printf("enum timing %d %d %d %d %d\n", (int)TIMINGS_LOAD_IMAGE, (int)TIMINGS_BUILD_PYRAMID, (int)TIMINGS_COMPUTE_DESCRIPTORS,
         (int)TIMINGS_FEATURES_REDUCTION, (int)TIMINGS_TOTAL);
  printf("enum match %d %d\n", (int)SiftMatchGPU::SIFTMATCH_SAME_AS_SIFTGPU, (int)SiftMatchGPU::SIFTMATCH_CUDA);
  Probe::layout();

  // 1. object from the library's factory, every call through the vtable (SimpleSIFT.cpp:100-202)
  SiftGPU* s = CreateNewSiftGPU(1);
  printf("factory %s\n", s ? "ok" : "null");
  const char* argv1[] = {"-v", "0", "-t", "0.01", "-d", "4", "-topk", "100"};
  s->ParseParam(8, (char**)argv1);
  printf("after ParseParam: _dog_level_num %d _dog_threshold %.4f\n", s->_dog_level_num, s->_dog_threshold);
  const char* files[] = {"a.pgm", "b.pgm", "c.pgm"};
  s->SetImageList(3, files);
  printf("GetImageCount %d\n", s->GetImageCount());
  printf("GetFeatureNum %d\n", s->GetFeatureNum());
  s->SetVerbose(0);
  s->SetTightPyramid(1);
  s->SetMaxDimension(2048);
  s->SetActivePyramid(0);
  int ctx = s->CreateContextGL();
  printf("CreateContextGL %s\n", ctx == SiftGPU::SIFTGPU_FULL_SUPPORTED ? "full" : (ctx == 0 ? "none" : "other"));
  printf("IsFullSupported consistent %d\n", (int)((s->IsFullSupported() != 0) == (ctx == SiftGPU::SIFTGPU_FULL_SUPPORTED)));
  printf("RunSIFT(missing file) %d\n", s->RunSIFT("/nonexistent/image.pgm"));
  printf("timing[total] finite %d\n", (int)(s->_timing[TIMINGS_TOTAL] == s->_timing[TIMINGS_TOTAL]));
  delete s;  // virtual destructor through the vtable, memory from the library's operator new (SiftGPU.cpp:116-125)

  // 2. stack object, as hessgpucmd.cpp:27 does: constructor/destructor symbols and sizeof(SiftGPU) of THIS header
  {
    SiftGPU sift;
    const char* argv2[] = {"-i", "x.pgm", "-fo", "-1", "-w", "2.5"};
    sift.ParseParam(6, (char**)argv2);
    printf("stack GetImageCount %d\n", sift.GetImageCount());
    unsigned char px[64 * 48];
    memset(px, 7, sizeof(px));
    int rc = sift.RunSIFT(64, 48, px, 0x1909 /* GL_LUMINANCE */, 0x1401 /* GL_UNSIGNED_BYTE */);
    printf("stack RunSIFT(pixels) %s\n", rc == 0 || rc == 1 ? "returned" : "bad");
  }
  // 3. the other factories of SiftGPU.h:364-379
  SiftMatchGPU* m = CreateNewSiftMatchGPU(128);
  printf("match factory %s\n", m ? "ok" : "null");
  if (m) delete m;
  printf("done\n");
  return 0;
}
