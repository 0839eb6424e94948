// SiftGPU.h -- plugin surface of the MI355X-native HessGPU build (libsiftgpu.so).
//
// Source- and binary-compatible with the class interface of the reference
// (src/SiftGPU/SiftGPU.h, GPU_HESSIAN build): same public data of SiftParam, same object layout of
// SiftGPU (callers stack-allocate it and read _timing[] directly: hessgpucmd.cpp:27,96-97), same
// virtual-function order (dlopen users call through the vtable: SimpleSIFT.cpp:88-202), same
// extern "C" factories.  Everything behind it is new: the methods drive the C ABI of
// include/hess_abi.h (hand-written HIP kernels for gfx950); there is no OpenGL, CUDA or DevIL.
//
// Not provided in this build: CreateRemoteSiftGPU (TCP server mode; returns NULL), SiftGPUEX (viewer),
// the rectangle-descriptor hack of SetKeypointList (keys_have_orientation == -1).
#ifndef GPU_SIFT_H
#define GPU_SIFT_H

#include <stddef.h>

// The reference's config.h values that callers index with (config.h:17-31, 45-50).
#ifndef _CONFIG_H
#define _CONFIG_H
#define GPU_HESSIAN
#define TOP_K_SELECTION
enum {
  TIMINGS_LOAD_IMAGE = 0, TIMINGS_ALLOCATE_PYRAMID, TIMINGS_BUILD_PYRAMID, TIMINGS_DETECT_KEYPOINTS,
  TIMINGS_GENERATE_FEATURE_LIST, TIMINGS_COMPUTE_ORIENTATIONS, TIMINGS_MULTI_ORIENTATIONS,
  TIMINGS_DOWNLOAD_KEYPOINTS, TIMINGS_COMPUTE_DESCRIPTORS, TIMINGS_GENERATE_VBO,
  TIMINGS_FEATURES_REDUCTION, TIMINGS_TOTAL, TIMINGS_COUNT
};
enum { FEATURE_TYPE_DARK_BLOB = 0, FEATURE_TYPE_BRIGHT_BLOB = 1, FEATURE_TYPE_SADDLE_POINT = 2, FEATURE_TYPE_NONE = 3 };
#endif

#define SIFTGPU_EXPORT
#define SIFTGPU_EXPORT_EXTERN extern "C"
#define SIFT_KEYPOINT_ITEMS 6

class SiftParam {
 public:
  float* _sigma;        // inter-level blur sigmas, _sigma_num entries
  float _sigma_skip0;
  float _sigma_skip1;
  float _sigma0;        // sigma of the first level (1.6)
  float _sigman;        // nominal sigma of the input (0.5)
  int _sigma_num;
  int _dog_level_num;   // scales per octave
  int _level_num;       // _dog_level_num + 2
  int _level_min;       // 0 in the Hessian build
  int _level_max;
  int _level_ds;
  float _dog_threshold;
  float _edge_threshold;
  void ParseSiftParam();

 public:
  float GetLevelSigma(int lev);
  float GetInitialSmoothSigma(int octave_min);
  SiftParam();
};

class LiteWindow;
class GLTexInput;
class ShaderMan;
class SiftPyramid;
class ImageList;

class SiftGPU : public SiftParam {
 public:
  enum { SIFTGPU_NOT_SUPPORTED = 0, SIFTGPU_PARTIAL_SUPPORTED = 1, SIFTGPU_FULL_SUPPORTED = 2 };
  typedef struct SiftKeypoint {
    float x, y, s, o;  // position, scale, orientation (mirrored angle)
    float response;    // det-Hessian response after sub-pixel refinement (through a half float)
    unsigned short level;
    unsigned short type;
  } SiftKeypoint;

 protected:
  int _current;
  int _initialized;
  int _image_loaded;
  char* _imgpath;
  char* _outpath;
  ImageList* _list;       // opaque in this build
  GLTexInput* _texImage;  // opaque in this build
  SiftPyramid* _pyramid;  // opaque in this build (owns the hess_ctx)
  static void PrintUsage();
  void InitSiftGPU();
  void LoadImageList(const char* imlist);

 public:
  float _timing[12];  // ms, indexed by TIMINGS_*
  inline const char* GetCurrentImagePath() { return _imgpath; }

 public:
  // -- virtual interface: order is ABI --
  virtual void SetImageList(int nimage, const char** filelist);
  virtual int GetFeatureNum();
  virtual void SaveSIFT(const char* szFileName);
  virtual void GetFeatureVector(SiftKeypoint* keys, float* descriptors);
  virtual void SetKeypointList(int num, const SiftKeypoint* keys, int keys_have_orientation = 1);
  virtual int CreateContextGL();   // creates the HIP context; returns SIFTGPU_FULL_SUPPORTED or 0
  virtual int VerifyContextGL();
  virtual int IsFullSupported();
  virtual void SetVerbose(int verbose = 4);
  inline void SetVerboseBrief() { SetVerbose(2); }
  virtual void ParseParam(int argc, char** argv);
  virtual int RunSIFT(const char* imgpath);
  virtual int RunSIFT(int index);
  virtual int RunSIFT(int width, int height, const void* data, unsigned int gl_format, unsigned int gl_type);
  virtual int RunSIFT();
  virtual int RunSIFT(int num, const SiftKeypoint* keys, int keys_have_orientation = 1);
  SiftGPU(int np = 1);
  virtual ~SiftGPU();
  virtual void SetActivePyramid(int index) { (void)index; }
  virtual int GetImageCount();
  virtual void SetTightPyramid(int tight = 1);
  virtual int AllocatePyramid(int width, int height);
  virtual void SetMaxDimension(int sz);

 public:
  void* operator new(size_t size);
};

// Descriptor matcher (SiftMatchGPU, CUDA flavour of the reference) on the hess_matcher_* C ABI.
class SiftMatchGPU {
 public:
  enum SIFTMATCH_LANGUAGE { SIFTMATCH_SAME_AS_SIFTGPU = 0, SIFTMATCH_GLSL = 2, SIFTMATCH_CUDA = 3, SIFTMATCH_CUDA_DEVICE0 = 3 };

 private:
  int __max_sift;
  int __language;
  SiftMatchGPU* __matcher;
  virtual void InitSiftMatch() {}

 protected:
  virtual int _CreateContextGL();
  virtual int _VerifyContextGL();

 public:
  inline int CreateContextGL() { return _CreateContextGL(); }
  inline int VerifyContextGL() { return _VerifyContextGL(); }
  SiftMatchGPU(int max_sift = 4096);
  virtual void SetLanguage(int gpu_language);
  virtual void SetDeviceParam(int argc, char** argv);
  virtual void SetMaxSift(int max_sift);
  virtual ~SiftMatchGPU();
  virtual void SetDescriptors(int index, int num, const float* descriptors, int id = -1);
  virtual void SetDescriptors(int index, int num, const unsigned char* descriptors, int id = -1);
  virtual int GetSiftMatch(int max_match, int match_buffer[][2], float distmax = 0.7, float ratiomax = 0.8,
                           int mutual_best_match = 1);
  virtual void SetFeautreLocation(int index, const float* locations, int gap = 0);
  inline void SetFeatureLocation(int index, const SiftGPU::SiftKeypoint* keys) {
    SetFeautreLocation(index, (const float*)keys, 2);
  }
  virtual int GetGuidedSiftMatch(int max_match, int match_buffer[][2], float H[3][3], float F[3][3],
                                 float distmax = 0.7, float ratiomax = 0.8, float hdistmax = 32,
                                 float fdistmax = 16, int mutual_best_match = 1);
  void* operator new(size_t size);
};

typedef SiftGPU::SiftKeypoint SiftKeypoint;

class ComboSiftGPU : public SiftGPU, public SiftMatchGPU {
 public:
  void* operator new(size_t size);
};

SIFTGPU_EXPORT_EXTERN SiftGPU* CreateNewSiftGPU(int np = 1);
SIFTGPU_EXPORT_EXTERN SiftMatchGPU* CreateNewSiftMatchGPU(int max_sift = 4096);
SIFTGPU_EXPORT_EXTERN ComboSiftGPU* CreateComboSiftGPU();
SIFTGPU_EXPORT_EXTERN ComboSiftGPU* CreateRemoteSiftGPU(int port = 7777, char* remote_server = NULL);  // NULL

// Flat C mirror of the class for FFI users and tests (ctypes / cgo / JNI cannot call a vtable).
extern "C" {
void siftgpu_destroy(SiftGPU* s);
void siftgpu_parse_param(SiftGPU* s, int argc, char** argv);
int siftgpu_create_context(SiftGPU* s);
int siftgpu_run_data(SiftGPU* s, int w, int h, const void* data, unsigned gl_format, unsigned gl_type);
int siftgpu_run_file(SiftGPU* s, const char* path);
int siftgpu_run_index(SiftGPU* s, int index);
int siftgpu_feature_num(SiftGPU* s);
void siftgpu_feature_vector(SiftGPU* s, SiftGPU::SiftKeypoint* keys, float* desc);
void siftgpu_save(SiftGPU* s, const char* path);
const float* siftgpu_timing(SiftGPU* s);
void siftgpu_set_verbose(SiftGPU* s, int v);
int siftgpu_image_count(SiftGPU* s);
SiftMatchGPU* siftmatch_create(int max_sift);
void siftmatch_destroy(SiftMatchGPU* m);
void siftmatch_set_descriptors_f32(SiftMatchGPU* m, int index, int num, const float* d);
int siftmatch_get_match(SiftMatchGPU* m, int max_match, int* buf, float distmax, float ratiomax, int mbm);
int siftgpu_run_keys(SiftGPU* s, int num, const SiftGPU::SiftKeypoint* keys, int have_orientation);
void siftgpu_set_keys(SiftGPU* s, int num, const SiftGPU::SiftKeypoint* keys, int have_orientation);
// The resolved hess_params of the instance (what ParseParam did), for tests.
int siftgpu_get_params(SiftGPU* s, void* hess_params_out);
int siftgpu_descriptor_dim(SiftGPU* s);
}

#endif  // GPU_SIFT_H
