// siftgpu_api.cpp -- the SiftGPU C++ plugin surface (include/SiftGPU.h) implemented on the C ABI
// of include/hess_abi.h.  Host C++ only: no OpenGL, CUDA or DevIL.  Behaviour follows the
// reference's SiftGPU.cpp (argv-style ParseParam with its first-four-characters option matching,
// RunSIFT overloads, GetFeatureVector, SaveSIFT formats, _timing[] indices) and
// GLTexImage.cpp's built-in PNM loader; citations per function.
#include "../../include/SiftGPU.h"

#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <fstream>
#include <future>
#include <map>
#include <iomanip>
#include <iostream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/hess_abi.h"

namespace {

// OpenGL enumerants accepted by RunSIFT(w,h,data,format,type) (GLTexImage.cpp IsSimpleGlFormat).
enum : unsigned {
  kGL_LUMINANCE = 0x1909, kGL_LUMINANCE_ALPHA = 0x190A, kGL_RGB = 0x1907, kGL_RGBA = 0x1908,
  kGL_BGR = 0x80E0, kGL_BGRA = 0x80E1, kGL_UNSIGNED_BYTE = 0x1401, kGL_UNSIGNED_SHORT = 0x1403,
  kGL_FLOAT = 0x1406
};

// Everything the reference keeps in process-global statics (GlobalUtil.h:35-111) lives per instance.
// A decoded image file, 8 bits per channel.  status: 1 ok; 0 cannot be opened / not a format this build reads; -1 a PNG
// or JPEG file that cannot be decoded (the loader has said why on stderr).
struct DecodedImage {
  std::vector<unsigned char> px;
  int w = 0, h = 0, fmt = 0, status = 0;
};

struct Impl {
  hess_params p;
  hess_ctx* ctx = nullptr;
  bool dirty = true;          // parameters changed since the context was created
  int device = 0;
  int verbose = 1, timingS = 1, timingO = 0, timingL = 0;  // GlobalUtil.cpp:51-54
  int binary_sift = 0;        // 0 text, 1 binary, 2 vlfeat (-b, -bvlf)
  int init_w = 0, init_h = 0; // -p WxH
  int tight = 0;
  float mr_size = 3.0f;       // GlobalUtil.cpp:93
  std::vector<std::string> list;
  // current image.  RunSIFT(w,h,data,..) borrows the caller's pointer for the call, as the reference does; the copy
  // the reference keeps for a later RunSIFT() (GLTexInput::_pixel_data) stays in the context's staging area and is
  // only brought back (hess_last_input) if the image is in fact run again without being handed over again.
  std::vector<unsigned char> pixels;
  const void* borrowed = nullptr;   // caller's pixels, during RunSIFT(w,h,data,..) only
  bool pixels_in_ctx = false;       // `pixels` is stale: the current image's pixels are the context's last input
  int w = 0, h = 0, fmt = 0, pix = 0;
  // results of the last run: they stay in the context's pinned host buffers (GetFeatureVector copies from there
  // straight into the caller's arrays) and are only copied here when something else needs them (SaveSIFT, or the
  // context being rebuilt after a parameter change)
  std::vector<hess_keypoint> keys;
  std::vector<float> desc;
  bool results_in_ctx = false;
  int nfeat = 0, dim = 0;
  std::vector<hess_keypoint> pending_keys;  // SetKeypointList before the context existed / was rebuilt
  int pending_keys_orient = 1;
  // An image LIST (SetImageList / -i / -il) tells which files come next: while image i is computed and saved, the files
  // of list entries i + 1 .. i + kAhead are read and decoded on host threads (`ahead`: list index -> decode in flight),
  // so that a caller walking the list -- hess -il (hessgpucmd.cpp:61-89), speed -- waits for the device, not for a
  // single-threaded JPEG decode.  list_index: the list entry the current _imgpath came from (-1: not from the list).
  // Twelve entries ahead, not four: a list of mixed sizes needs LEAD TIME, not only threads -- a 2048 x 1536 JPEG decodes in
  // 25 ms, thirty times what an image of the list takes to run, so with four entries of lead the walk stalled 20 ms at
  // every large file (2.7 ms per image over the reference's data/ mix; profiles/r06_experiments/list_callers.txt).
  static constexpr int kAhead = 12;
  int list_index = -1;
  std::map<int, std::future<DecodedImage>> ahead;
};

inline Impl* I(SiftPyramid* p) { return reinterpret_cast<Impl*>(p); }

int gl_to_hess(unsigned gl_format, unsigned gl_type, int* fmt, int* pix) {
  switch (gl_format) {
    case kGL_LUMINANCE: *fmt = HESS_FMT_LUM; break;
    case kGL_LUMINANCE_ALPHA: *fmt = HESS_FMT_LUM_ALPHA; break;
    case kGL_RGB: *fmt = HESS_FMT_RGB; break;
    case kGL_RGBA: *fmt = HESS_FMT_RGBA; break;
    case kGL_BGR: *fmt = HESS_FMT_BGR; break;
    case kGL_BGRA: *fmt = HESS_FMT_BGRA; break;
    default: return 0;
  }
  switch (gl_type) {
    case kGL_UNSIGNED_BYTE: *pix = HESS_PIX_U8; break;
    case kGL_UNSIGNED_SHORT: *pix = HESS_PIX_U16; break;
    case kGL_FLOAT: *pix = HESS_PIX_F32; break;
    default: return 0;
  }
  return 1;
}

int channels(int fmt) { return fmt == HESS_FMT_LUM ? 1 : fmt == HESS_FMT_LUM_ALPHA ? 2 : (fmt == HESS_FMT_RGB || fmt == HESS_FMT_BGR) ? 3 : 4; }
int pix_bytes(int pix) { return pix == HESS_PIX_U8 ? 1 : pix == HESS_PIX_U16 ? 2 : 4; }

// Option key: the first up to four characters, lower-cased (STRING_TO_INT, SiftGPU.cpp:857-860).
std::string opt_key(const char* opt) {
  std::string k;
  for (int i = 0; i < 4 && opt[i]; i++) k.push_back((char)tolower((unsigned char)opt[i]));
  return k;
}

// Built-in PNM loader (the reference's SIFTGPU_NO_DEVIL path, GLTexImage.cpp:1159-1220): P2/P3/P5/P6,
// colour files reduced with its PNM-specific weights int(0.10454 B + 0.60581 G + 0.28965 R).
bool load_pnm(const char* path, std::vector<unsigned char>& out, int& w, int& h) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  char magic[8] = {0};
  int cn = 0;
  if (fscanf(f, "%7s %d %d %d", magic, &w, &h, &cn) < 4 || cn > 255 || w <= 0 || h <= 0) { fclose(f); return false; }
  out.assign((size_t)w * h, 0);
  bool ok = true;
  if (!strcmp(magic, "P5")) {
    fgetc(f);
    ok = fread(out.data(), 1, out.size(), f) == out.size();
  } else if (!strcmp(magic, "P2")) {
    for (size_t i = 0; i < out.size() && ok; i++) { int g; ok = fscanf(f, "%d", &g) == 1; out[i] = (unsigned char)g; }
  } else if (!strcmp(magic, "P6")) {
    fgetc(f);
    unsigned char rgb[3];
    for (size_t i = 0; i < out.size() && ok; i++) {
      ok = fread(rgb, 1, 3, f) == 3;
      out[i] = (unsigned char)int(0.10454f * rgb[2] + 0.60581f * rgb[1] + 0.28965f * rgb[0]);
    }
  } else if (!strcmp(magic, "P3")) {
    for (size_t i = 0; i < out.size() && ok; i++) {
      int r, g, b;
      ok = fscanf(f, "%d %d %d", &r, &g, &b) == 3;
      out[i] = (unsigned char)int(0.10454f * b + 0.60581f * g + 0.28965f * r);
    }
  } else {
    ok = false;
  }
  fclose(f);
  return ok;
}

// PNG files through libpng's "simplified API" (png.h of libpng 1.6: png_image_begin_read_from_file / _finish_read /
// _free), looked up at RUN time: the reference decodes files with DevIL (GLTexImage.cpp:1117-1158) and hands the decoded
// pixels in the file's own layout to SetImageData, which is what happens here -- grey, grey + alpha, RGB or RGBA, 8 bits
// (16-bit and palette files are converted to those by libpng).  No build dependency: the image this was written on has
// libpng16.so.16 and no headers; png_image is the documented, versioned public struct of that API.  Returns 0: not a PNG
// file; -1: a PNG file that cannot be read (library missing or decode error, message on stderr); else the HESS_FMT_*.
struct PngImage {
  void* opaque;
  uint32_t version, width, height, format, flags, colormap_entries;
  uint32_t warning_or_error;
  char message[64];
};
int load_png(const char* path, std::vector<unsigned char>& out, int& w, int& h) {
  unsigned char sig[8] = {0};
  FILE* f = fopen(path, "rb");
  if (!f) return 0;
  const bool is_png = fread(sig, 1, 8, f) == 8 && !memcmp(sig, "\x89PNG\r\n\x1a\n", 8);
  fclose(f);
  if (!is_png) return 0;
  typedef int (*begin_fn)(PngImage*, const char*);
  typedef int (*finish_fn)(PngImage*, const void*, void*, int32_t, void*);
  typedef void (*free_fn)(PngImage*);
  static void* lib = dlopen("libpng16.so.16", RTLD_LAZY | RTLD_LOCAL);
  static begin_fn begin = lib ? (begin_fn)dlsym(lib, "png_image_begin_read_from_file") : nullptr;
  static finish_fn finish = lib ? (finish_fn)dlsym(lib, "png_image_finish_read") : nullptr;
  static free_fn release = lib ? (free_fn)dlsym(lib, "png_image_free") : nullptr;
  if (!begin || !finish || !release) {
    std::cerr << "PNG file, but libpng16.so.16 is not available at run time: " << path << "\n";
    return -1;
  }
  PngImage img;
  memset(&img, 0, sizeof(img));
  img.version = 1;  // PNG_IMAGE_VERSION
  if (!begin(&img, path)) { std::cerr << "libpng: " << img.message << ": " << path << "\n"; return -1; }
  const bool colour = (img.format & 0x02u) != 0, alpha = (img.format & 0x01u) != 0;  // PNG_FORMAT_FLAG_COLOR / _ALPHA
  img.format = (colour ? 0x02u : 0u) | (alpha ? 0x01u : 0u);  // PNG_FORMAT_GRAY / GA / RGB / RGBA: 8-bit, no colour map
  const int ch = (colour ? 3 : 1) + (alpha ? 1 : 0);
  w = (int)img.width; h = (int)img.height;
  try {
    out.assign((size_t)w * h * ch, 0);
  } catch (...) { release(&img); return -1; }
  if (!finish(&img, nullptr, out.data(), 0, nullptr)) {
    std::cerr << "libpng: " << img.message << ": " << path << "\n";
    release(&img);
    return -1;
  }
  return ch == 1 ? HESS_FMT_LUM : ch == 2 ? HESS_FMT_LUM_ALPHA : ch == 3 ? HESS_FMT_RGB : HESS_FMT_RGBA;
}

// JPEG files (the reference's data/*.jpg, read there by DevIL: GLTexImage.cpp:1117-1158) through libjpeg, looked up at
// RUN time like libpng above.  libjpeg's decompress object is a large struct whose layout depends on the library version,
// so this part is only compiled where the library's own header is on the include path (the build adds the directory of
// one whose shared library it finds: hessgpu_amd/build.py), and it asks for the library OF THAT VERSION by name; the
// library checks version and struct size itself (jpeg_CreateDecompress) and every error comes back through error_exit ->
// longjmp.  Grey files are handed on as luminance, everything else as RGB.  Returns 0: not a JPEG file; -1: a JPEG file
// that cannot be read (library missing or decode error, message on stderr); else the HESS_FMT_*.
#if defined(__has_include)
#if __has_include(<jpeglib.h>)
#define HESS_HAVE_JPEGLIB 1
#endif
#endif
#ifdef HESS_HAVE_JPEGLIB
}  // namespace
#include <csetjmp>
extern "C" {
#include <jpeglib.h>
}
namespace {
struct JpegErr {
  jpeg_error_mgr pub;
  jmp_buf jb;
  char msg[JMSG_LENGTH_MAX];
};
void jpeg_fail(j_common_ptr cinfo) {
  JpegErr* e = reinterpret_cast<JpegErr*>(cinfo->err);
  (*cinfo->err->format_message)(cinfo, e->msg);
  longjmp(e->jb, 1);
}
#define HESS_STR2(x) #x
#define HESS_STR(x) HESS_STR2(x)
#endif
int load_jpeg(const char* path, std::vector<unsigned char>& out, int& w, int& h) {
  unsigned char sig[3] = {0};
  FILE* f = fopen(path, "rb");
  if (!f) return 0;
  const bool is_jpeg = fread(sig, 1, 3, f) == 3 && sig[0] == 0xFF && sig[1] == 0xD8 && sig[2] == 0xFF;
  if (!is_jpeg) { fclose(f); return 0; }
#ifndef HESS_HAVE_JPEGLIB
  fclose(f);
  std::cerr << "JPEG file, but this build was made without libjpeg's header: " << path << "\n";
  return -1;
#else
  rewind(f);
  // the library whose header this was compiled against: libjpeg.so.<major> (IJG 9: libjpeg.so.9, libjpeg-turbo's
  // version-8 emulation: libjpeg.so.8), also beside the header's own installation
  static void* lib = [] {
    const char* names[] = {"libjpeg.so." HESS_STR(JPEG_LIB_VERSION_MAJOR),
#ifdef HESS_JPEG_LIBDIR
                           HESS_JPEG_LIBDIR "/libjpeg.so." HESS_STR(JPEG_LIB_VERSION_MAJOR),
#endif
                           nullptr};
    for (const char** n = names; *n; n++)
      if (void* l = dlopen(*n, RTLD_LAZY | RTLD_LOCAL)) return l;
    return (void*)nullptr;
  }();
#define HESS_JSYM(name) static decltype(&::name) p_##name = lib ? (decltype(&::name))dlsym(lib, #name) : nullptr
  HESS_JSYM(jpeg_std_error); HESS_JSYM(jpeg_CreateDecompress); HESS_JSYM(jpeg_stdio_src); HESS_JSYM(jpeg_read_header);
  HESS_JSYM(jpeg_start_decompress); HESS_JSYM(jpeg_read_scanlines); HESS_JSYM(jpeg_finish_decompress);
  HESS_JSYM(jpeg_destroy_decompress);
#undef HESS_JSYM
  if (!p_jpeg_std_error || !p_jpeg_CreateDecompress || !p_jpeg_stdio_src || !p_jpeg_read_header || !p_jpeg_start_decompress ||
      !p_jpeg_read_scanlines || !p_jpeg_finish_decompress || !p_jpeg_destroy_decompress) {
    fclose(f);
    std::cerr << "JPEG file, but libjpeg.so." HESS_STR(JPEG_LIB_VERSION_MAJOR) " is not available at run time: " << path << "\n";
    return -1;
  }
  jpeg_decompress_struct cinfo;
  JpegErr err;
  memset(&cinfo, 0, sizeof(cinfo));
  cinfo.err = p_jpeg_std_error(&err.pub);
  err.pub.error_exit = jpeg_fail;
  volatile int fmt = -1;
  if (setjmp(err.jb)) {  // every libjpeg error ends here (incl. a version / struct-size mismatch)
    std::cerr << "libjpeg: " << err.msg << ": " << path << "\n";
    p_jpeg_destroy_decompress(&cinfo);
    fclose(f);
    return -1;
  }
  p_jpeg_CreateDecompress(&cinfo, JPEG_LIB_VERSION, sizeof(cinfo));
  p_jpeg_stdio_src(&cinfo, f);
  p_jpeg_read_header(&cinfo, TRUE);
  const bool grey = cinfo.jpeg_color_space == JCS_GRAYSCALE;
  cinfo.out_color_space = grey ? JCS_GRAYSCALE : JCS_RGB;
  p_jpeg_start_decompress(&cinfo);
  const int ch = (int)cinfo.output_components;
  if (ch != (grey ? 1 : 3)) { strcpy(err.msg, "unexpected number of output components"); longjmp(err.jb, 1); }
  w = (int)cinfo.output_width; h = (int)cinfo.output_height;
  out.assign((size_t)w * h * ch, 0);  // (bad_alloc: the class has no exception boundary above RunSIFT either)
  while (cinfo.output_scanline < cinfo.output_height) {
    JSAMPROW row = out.data() + (size_t)cinfo.output_scanline * w * ch;
    p_jpeg_read_scanlines(&cinfo, &row, 1);
  }
  p_jpeg_finish_decompress(&cinfo);
  p_jpeg_destroy_decompress(&cinfo);
  fclose(f);
  fmt = grey ? HESS_FMT_LUM : HESS_FMT_RGB;
  return fmt;
#endif
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// SiftParam: SiftGPU.cpp:466-563, 1422-1425

SiftParam::SiftParam() {
  _sigma = nullptr;
  _sigma_skip0 = _sigma_skip1 = 0;
  _sigma_num = 0;
  _level_min = 0;
  _dog_level_num = 3;
  _level_num = 0;
  _level_max = 0;
  _level_ds = 0;
  _sigma0 = 0;
  _sigman = 0;
  _edge_threshold = 0;
  _dog_threshold = 0;
}

float SiftParam::GetInitialSmoothSigma(int octave_min) {
  float sa = _sigma0 * powf(2.0f, float(_level_min) / float(_dog_level_num));
  float sb = _sigman / powf(2.0f, float(octave_min));
  return (sa > sb + 0.001) ? sqrtf(sa * sa - sb * sb) : 0.0f;
}

float SiftParam::GetLevelSigma(int lev) { return _sigma0 * powf(2.0f, float(lev) / float(_dog_level_num)); }

void SiftParam::ParseSiftParam() {
  if (_dog_level_num == 0) _dog_level_num = 3;
  if (_level_max == 0) _level_max = _dog_level_num + 1;
  if (_sigma0 == 0.0f) _sigma0 = 1.6f;
  if (_sigman == 0.0f) _sigman = 0.5f;
  _level_num = _level_max - _level_min + 1;
  _level_ds = _level_min + _dog_level_num;
  if (_level_ds > _level_max) _level_ds = _level_max;
  const float sigmak = powf(2.0f, 1.0f / _dog_level_num);
  const float dsigma0 = _sigma0 * sqrtf(sigmak * sigmak - 1.0f);
  float sa = _sigma0 * powf(sigmak, (float)_level_min);
  float sb = _sigman;  // first octave 0
  _sigma_skip0 = (sa > sb + 0.001) ? sqrtf(sa * sa - sb * sb) : 0.0f;
  sb = _sigma0 * powf(sigmak, float(_level_ds - _dog_level_num));
  _sigma_skip1 = (sa > sb + 0.001) ? sqrtf(sa * sa - sb * sb) : 0.0f;
  _sigma_num = _level_max - _level_min;
  delete[] _sigma;
  _sigma = new float[_sigma_num];
  for (int i = _level_min + 1; i <= _level_max; i++)
    _sigma[i - (_level_min + 1)] = dsigma0 * powf(sigmak, float(i - (_level_min + 1)));
  if (_dog_threshold == 0) _dog_threshold = 0.02f / _dog_level_num;
  if (_edge_threshold == 0) _edge_threshold = 10.0f;
}

// ------------------------------------------------------------------------------------------------
// SiftGPU

namespace {
struct Peek0 : SiftGPU {
  static Impl* impl(SiftGPU* s) { return I(static_cast<Peek0*>(s)->_pyramid); }
};
// Counts and stage times of the last run; the keypoints and descriptors stay in the context (see Impl).
void collect_results(SiftGPU* self) {
  Impl* im = Peek0::impl(self);
  im->nfeat = hess_count(im->ctx, 0);
  im->dim = hess_desc_dim(im->ctx);
  im->results_in_ctx = true;
  const float* t = hess_timing(im->ctx);
  for (int k = 0; k < 12; k++) self->_timing[k] = t[k];
}
// Bring the results of the last run into the instance's own arrays (before the context goes away, or for SaveSIFT).
void materialize_results(Impl* im) {
  if (!im->results_in_ctx || !im->ctx) return;
  im->results_in_ctx = false;
  // The arrays are sized from what the context holds NOW, never from the instance's bookkeeping: hess_fetch copies
  // hess_count() records, so a count that disagrees (a failed run in between) means there is nothing to bring over.
  const int held = hess_count(im->ctx, 0);
  if (held != im->nfeat || held <= 0) {
    if (held != im->nfeat) im->nfeat = 0;
    im->keys.clear();
    im->desc.clear();
    return;
  }
  im->keys.resize((size_t)held);
  im->desc.resize((size_t)held * (im->dim ? im->dim : 1));
  hess_fetch(im->ctx, 0, im->keys.data(), im->dim ? im->desc.data() : nullptr);
}
// Same for the pixels of the current image.
bool materialize_pixels(Impl* im) {
  if (!im->pixels_in_ctx) return true;
  const size_t bytes = (size_t)im->w * im->h * channels(im->fmt) * pix_bytes(im->pix);
  im->pixels.resize(bytes);
  im->pixels_in_ctx = false;
  return im->ctx && hess_last_input(im->ctx, im->pixels.data(), bytes) == 0;
}
void drop_context(Impl* im) {
  if (!im->ctx) return;
  materialize_results(im);
  materialize_pixels(im);
  hess_destroy(im->ctx);
  im->ctx = nullptr;
}
// What RunSIFT(path) does with a file: PNG, JPEG (run-time libraries), then the built-in PNM reader.
DecodedImage decode_file(const std::string& path) {
  DecodedImage d;
  const int png = load_png(path.c_str(), d.px, d.w, d.h);
  if (png < 0) { d.status = -1; return d; }
  const int jpg = png == 0 ? load_jpeg(path.c_str(), d.px, d.w, d.h) : 0;
  if (jpg < 0) { d.status = -1; return d; }
  if (png > 0 || jpg > 0) { d.fmt = png > 0 ? png : jpg; d.status = 1; }
  else if (load_pnm(path.c_str(), d.px, d.w, d.h)) { d.fmt = HESS_FMT_LUM; d.status = 1; }
  return d;
}

// Start decoding the list entries after `index` that are not in flight yet; forget the ones that fell out of the window
// (a forgotten future waits for its thread: a decode is milliseconds).
void decode_ahead(Impl* im, int index) {
  const int n = (int)im->list.size();
  if (n < 2) { im->ahead.clear(); return; }
  std::map<int, std::future<DecodedImage>> keep;
  for (int k = 1; k <= Impl::kAhead && k < n; k++) {
    const int j = (index + k) % n;
    if (keep.count(j)) continue;
    auto it = im->ahead.find(j);
    if (it != im->ahead.end()) { keep.emplace(j, std::move(it->second)); im->ahead.erase(it); }
    else keep.emplace(j, std::async(std::launch::async, decode_file, im->list[j]));
  }
  im->ahead.swap(keep);
}
}  // namespace

SiftGPU::SiftGPU(int np) {
  (void)np;
  _texImage = nullptr;
  _list = nullptr;
  _imgpath = new char[4096];
  _outpath = new char[4096];
  _imgpath[0] = _outpath[0] = 0;
  _initialized = 0;
  _image_loaded = 0;
  _current = 0;
  Impl* im = new Impl();
  hess_default_params(&im->p);
  _pyramid = reinterpret_cast<SiftPyramid*>(im);
  memset(_timing, 0, sizeof(_timing));
}

SiftGPU::~SiftGPU() {
  Impl* im = I(_pyramid);
  if (im) {
    if (im->ctx) hess_destroy(im->ctx);
    delete im;
  }
  delete[] _imgpath;
  delete[] _outpath;
  delete[] _sigma;
}

void* SiftGPU::operator new(size_t size) {  // SiftGPU.cpp:116-125: heap allocation inside the library
  void* p = malloc(size);
  if (!p) throw std::bad_alloc();
  return p;
}

void SiftGPU::PrintUsage() {
  std::cout << "SiftGPU (MI355X HessGPU build) options:\n"
               "-i <files..> -il <listfile> -o <out>     input images / image list / output file\n"
               "-t <f> -e <f> -d <n> -fo <n> -no <n>     threshold, edge threshold, scales per octave, first octave, octaves\n"
               "-f <f> -w <f> -dw <f>                    filter width, orientation window, descriptor window factors\n"
               "-m [n] -s [n] -ofix -ofix-not -loweo     orientations, sub-pixel, fixed orientation, Lowe origin\n"
               "-topk <n> -tc/-tc1/-tc2/-tc3 <n>         limit the number of features\n"
               "-half -sd -b -bvlf -ads -maxd <n> -p WxH -tight -cuda <dev> -v <0..4>\n"
               "-dseq                                    descriptor bins summed in the reference's sequential order\n"
               "-dint                                    ... as four interleaved partial sums (equal to -dseq within 1e-6)\n"
               "                                         default: one pass over the pixels, fixed-point sums -- within 3e-5 of -dseq\n"
               "                                         (1.6e-5 measured at 4096^2, 6e-6 at 1080p; include/hess_abi.h, HESS_DESC_ORDER_*)\n"
               "Image files: PGM / PPM (P2 P3 P5 P6); PNG when libpng16.so.16, JPEG when libjpeg is present at run time -- this build has no\n"
               "DevIL; decode JPEG in the caller and hand the pixels to RunSIFT(width, height, data, gl_format, gl_type).\n";
}

void SiftGPU::SetVerbose(int verbose) {  // SiftGPU.cpp:433-464
  Impl* im = I(_pyramid);
  const int before = (im->verbose ? 1 : 0) | (im->timingS ? 2 : 0);
  im->timingO = (verbose > 2);
  im->timingL = (verbose > 3);
  if (verbose == -1) {
    if (im->verbose) { im->verbose = im->timingS; im->timingS = 0; }
    else { im->verbose = 1; im->timingS = 1; }
  } else if (verbose == -2) {
    im->verbose = 0;
    im->timingS = 1;
  } else {
    im->verbose = (verbose > 0);
    im->timingS = (verbose > 1);
  }
  // messages / stage timers are properties of the device context (hess_params.verbose): rebuilt on the next run
  if (((im->verbose ? 1 : 0) | (im->timingS ? 2 : 0)) != before) im->dirty = true;
}

void SiftGPU::ParseParam(int argc, char** argv) {  // SiftGPU.cpp:855-1380
  Impl* im = I(_pyramid);
  hess_params& p = im->p;
  for (int i = 0; i < argc; i++) {
    const char* arg = argv[i];
    if (!arg || arg[0] != '-' || !arg[1]) continue;
    const char* opt = arg + 1;
    const std::string k = opt_key(opt);
    const char* param = (i + 1 < argc) ? argv[i + 1] : nullptr;
    im->dirty = true;
    // ---- options without a mandatory value ----
    if (k == "h" || k == "help") { PrintUsage(); continue; }
    if (k == "cuda") {
      if (!_initialized) {
        int device = -1;
        if (param && sscanf(param, "%d", &device) && device >= 0) { im->device = device; i++; }
      }
      continue;
    }
    if (k == "lcpu" || k == "lc" || k == "prep" || k == "nopr" || k == "exit" || k == "debu" ||
        k == "k0" || k == "kx" || k == "da" || k == "fmc" || k == "nomc")
      continue;  // accepted, no effect on this backend
    if (k == "di") { p.dynamic_indexing = 1; continue; }  // SiftGPU.cpp:1030-1032
    if (k == "dseq") { p.descriptor_order = HESS_DESC_ORDER_SEQUENTIAL; continue; }  // this build only: hess_abi.h
    if (k == "dint") { p.descriptor_order = HESS_DESC_ORDER_INTERLEAVED; continue; }
    if (k == "sd") { if (!_initialized) p.compute_descriptors = 0; continue; }
    if (k == "b") { im->binary_sift = 1; continue; }
    if (k == "ads") { p.auto_downscale = 1; continue; }
    if (k == "bvlf") { im->binary_sift = 2; continue; }
    if (k == "half") { p.half_sift = 1; continue; }
    if (k == "tigh") { im->tight = 1; continue; }
    if (k == "m" || k == "mo") {
      if (!_initialized) {
        int mo = 2;
        if (param) sscanf(param, "%d", &mo);
        p.max_orientation = mo < 1 ? 1 : (mo > 4 ? 4 : mo);
      }
      continue;  // the value is read but not consumed (SiftGPU.cpp:1039-1049)
    }
    if (k == "s") {
      if (!_initialized) {
        int sp = 1;
        if (param) sscanf(param, "%d", &sp);
        p.subpixel = sp < 0 ? 0 : (sp > 5 ? 5 : sp);
      }
      continue;
    }
    if (k == "ofix") { p.fixed_orientation = (strcasecmp(opt, "ofix") == 0); continue; }
    if (k == "lowe") { p.lowe_origin = 1; continue; }
    // ---- options that need a value ----
    if (!param) continue;
    if (k == "i") {
      strcpy(_imgpath, param);
      i++;
      im->ahead.clear();
      im->list.push_back(param);
      while (i + 1 < argc && argv[i + 1][0] != '-') im->list.push_back(argv[++i]);
    } else if (k == "il") {
      LoadImageList(param);
      i++;
    } else if (k == "o") {
      strcpy(_outpath, param);
      i++;
    } else if (k == "f") {
      float v = 0; if (sscanf(param, "%f", &v) && v > 0) { p.filter_width_factor = v; i++; }
    } else if (k == "ot") {
      float v = 0; if (sscanf(param, "%f", &v) && v > 0) i++;  // parsed; the kernel constant is 0.8
    } else if (k == "w") {
      float v = 0; if (sscanf(param, "%f", &v) && v > 0) { p.orient_window_factor = v; i++; }
    } else if (k == "dw") {
      float v = 0; if (sscanf(param, "%f", &v) && v > 0) { p.desc_window_factor = v; i++; }
    } else if (k == "fo") {
      int v = -3; if (sscanf(param, "%d", &v) && v >= 0) { p.first_octave = v; i++; }
    } else if (k == "no") {
      if (!_initialized) {
        int v = -1;
        if (sscanf(param, "%d", &v)) { if (v < -1) v = -1; if (v == -1 || v >= 1) { p.octave_num = v; i++; } }
      }
    } else if (k == "t") {
      float v = 0; if (sscanf(param, "%f", &v) && v > 0 && v < 0.5f) { _dog_threshold = v; i++; }
    } else if (k == "e") {
      float v = 0; if (sscanf(param, "%f", &v) && v > 0) { _edge_threshold = v; i++; }
    } else if (k == "d") {
      int v = 0; if (sscanf(param, "%d", &v) && v >= 1 && v <= 10) { _dog_level_num = v; i++; }
    } else if (k == "fs" || k == "lm" || k == "lmp" || k == "winp" || k == "disp") {
      i++;  // storage-block / window options of other backends
    } else if (k == "p") {
      int w = 0, h = 0;
      if (sscanf(param, "%dx%d", &w, &h) == 2 && w > 0 && h > 0) { im->init_w = w; im->init_h = h; i++; }
    } else if (k == "tc" || k == "tc1" || k == "tc2" || k == "tc3" || k == "topk") {
      p.truncate_method = (k == "tc2") ? HESS_TRUNC_HIGHEST_1 : (k == "tc3") ? HESS_TRUNC_LOWEST
                          : (k == "topk") ? HESS_TRUNC_TOPK : HESS_TRUNC_HIGHEST_0;
      int v = -1;
      if (sscanf(param, "%d", &v) && v > 0) { p.feature_count_threshold = v; i++; }
    } else if (k == "v") {
      int v = 0; if (sscanf(param, "%d", &v) && v >= 0 && v <= 4) SetVerbose(v);
    } else if (k == "maxd") {
      int v = 0; if (sscanf(param, "%d", &v) && v > 0) p.tex_max_dim = v;
    } else if (k == "mind") {
      // parsed; the CUDA backend of the reference never reads _texMinDim
    }
  }
  if (_outpath[0] && im->list.size() > 1) _outpath[0] = 0;  // SiftGPU.cpp:1377-1379
}

void SiftGPU::SetImageList(int nimage, const char** filelist) {
  Impl* im = I(_pyramid);
  im->ahead.clear();
  im->list.clear();
  for (int i = 0; i < nimage; i++) im->list.push_back(filelist[i]);
  _current = 0;
}

void SiftGPU::LoadImageList(const char* imlist) {  // SiftGPU.cpp:1394-1420
  Impl* im = I(_pyramid);
  std::ifstream in(imlist);
  std::string name;
  im->ahead.clear();  // (the working directory changes below: nothing decoded against the old one stays)
  while (in >> name) im->list.push_back(name);
  if (!im->list.empty()) {
    strcpy(_imgpath, im->list[0].c_str());
    std::string dir(imlist);
    size_t slash = dir.find_last_of("\\/");
    if (slash != std::string::npos) {
      dir.resize(slash + 1);
      if (chdir(dir.c_str()) != 0 && im->verbose) std::cerr << "cannot chdir to " << dir << "\n";
    }
  }
  _image_loaded = 0;
}

int SiftGPU::GetImageCount() { return (int)I(_pyramid)->list.size(); }
void SiftGPU::SetTightPyramid(int tight) { I(_pyramid)->tight = tight; }
void SiftGPU::SetMaxDimension(int sz) { if (sz > 0) { I(_pyramid)->p.tex_max_dim = sz; I(_pyramid)->dirty = true; } }

// InitSiftGPU (SiftGPU.cpp:149-227): resolve the schedule and (re)create the device context.
void SiftGPU::InitSiftGPU() {
  Impl* im = I(_pyramid);
  if (_initialized && !im->dirty && im->ctx) return;
  ParseSiftParam();
  hess_params p = im->p;
  p.dog_level_num = _dog_level_num;
  p.sigma0 = _sigma0;
  p.sigman = _sigman;
  p.dog_threshold = _dog_threshold;
  p.edge_threshold = _edge_threshold;
  p.verbose = (im->verbose ? 1 : 0) | (im->timingS ? 2 : 0);  // messages | stage timers (hess_abi.h)
  drop_context(im);  // (results and pixels of the last run are kept in the instance)
  im->ctx = hess_create(im->device, &p);
  im->p = p;
  im->dirty = false;
  _initialized = 1;
  if (im->ctx && im->init_w > 0 && im->init_h > 0) hess_reserve(im->ctx, im->init_w, im->init_h, 1);
}

int SiftGPU::CreateContextGL() { return VerifyContextGL(); }  // SiftGPU.cpp:1516-1539

int SiftGPU::VerifyContextGL() {
  InitSiftGPU();
  return I(_pyramid)->ctx ? SIFTGPU_FULL_SUPPORTED : SIFTGPU_NOT_SUPPORTED;
}

int SiftGPU::IsFullSupported() { return I(_pyramid)->ctx != nullptr; }

int SiftGPU::AllocatePyramid(int width, int height) {
  InitSiftGPU();
  Impl* im = I(_pyramid);
  return im->ctx && hess_reserve(im->ctx, width, height, 1) == 0;
}

int SiftGPU::RunSIFT(int index) {  // SiftGPU.cpp:229-246
  Impl* im = I(_pyramid);
  if (im->list.empty()) return 0;
  index = index % (int)im->list.size();
  if (strcmp(_imgpath, im->list[index].c_str())) {
    strcpy(_imgpath, im->list[index].c_str());
    _image_loaded = 0;
    _current = index;
  }
  im->list_index = index;
  const int ok = RunSIFT();
  im->list_index = -1;
  return ok;
}

int SiftGPU::RunSIFT(const char* imgpath) {  // SiftGPU.cpp:292-305
  if (!imgpath || !imgpath[0]) return 0;
  strcpy(_imgpath, imgpath);
  _image_loaded = 0;
  return RunSIFT();
}

int SiftGPU::RunSIFT(int width, int height, const void* data, unsigned int gl_format, unsigned int gl_type) {
  // SiftGPU.cpp:248-290 -> GLTexInput::SetImageData
  Impl* im = I(_pyramid);
  int fmt, pix;
  if (width <= 0 || height <= 0 || !data) return 0;
  if (!gl_to_hess(gl_format, gl_type, &fmt, &pix)) {
    std::cerr << "Input format not supported under current settings.\n";
    return 0;
  }
  im->w = width; im->h = height; im->fmt = fmt; im->pix = pix;
  _imgpath[0] = 0;
  _image_loaded = 2;
  im->borrowed = data;  // no copy here: the context stages the pixels and keeps them until the next image
  im->pixels_in_ctx = false;
  const int ok = RunSIFT();
  im->borrowed = nullptr;
  im->pixels_in_ctx = ok != 0;
  if (!ok) { _image_loaded = 0; im->pixels.clear(); }
  return ok;
}

int SiftGPU::RunSIFT(int num, const SiftKeypoint* keys, int keys_have_orientation) {  // SiftGPU.cpp:307-315
  Impl* im = I(_pyramid);
  if (num <= 0 || !keys || !im->ctx) return 0;
  const int rc = hess_run_keypoints(im->ctx, reinterpret_cast<const hess_keypoint*>(keys), num, keys_have_orientation);
  if (rc != 0) {
    std::cerr << "SiftGPU: " << hess_last_error(im->ctx) << "\n";
    im->results_in_ctx = false;  // a failed run leaves no results
    im->nfeat = 0;
    return 0;
  }
  collect_results(this);
  return 1;
}

void SiftGPU::SetKeypointList(int num, const SiftKeypoint* keys, int keys_have_orientation) {
  Impl* im = I(_pyramid);
  InitSiftGPU();
  if (im->ctx) hess_set_keypoints(im->ctx, reinterpret_cast<const hess_keypoint*>(keys), num, keys_have_orientation);
  im->pending_keys.assign(reinterpret_cast<const hess_keypoint*>(keys), reinterpret_cast<const hess_keypoint*>(keys) + (num > 0 ? num : 0));
  im->pending_keys_orient = keys_have_orientation;
}

int SiftGPU::RunSIFT() {  // SiftGPU.cpp:317-415
  Impl* im = I(_pyramid);
  if (_imgpath[0] == 0 && _image_loaded == 0) return 0;
  InitSiftGPU();
  if (!im->ctx) return 0;
  memset(_timing, 0, sizeof(_timing));
  // whatever the last run left is about to be replaced -- or, when this run fails at any point below, to become
  // stale: a failed run leaves no results (SiftPyramid.h:162-163)
  im->results_in_ctx = false;
  im->nfeat = 0;
  im->keys.clear();
  im->desc.clear();
  if (_image_loaded == 0) {
    DecodedImage d;
    auto ahead = im->list_index >= 0 ? im->ahead.find(im->list_index) : im->ahead.end();
    if (ahead != im->ahead.end()) {  // decoded while the images before it ran
      d = ahead->second.get();
      im->ahead.erase(ahead);
    } else {
      d = decode_file(_imgpath);
    }
    if (im->list_index >= 0) decode_ahead(im, im->list_index);  // the next entries' files, while this image runs
    if (d.status < 0) return 0;
    if (d.status > 0) {
      im->pixels.swap(d.px);
      im->w = d.w; im->h = d.h; im->fmt = d.fmt;
    } else {
      std::cerr << "Unable to open image (this build reads PGM / PPM and, with libpng16 / libjpeg present at run time, PNG "
                   "and JPEG; other formats: decode in the caller and use RunSIFT(width, height, data, gl_format, gl_type)): "
                << _imgpath << "\n";
      return 0;
    }
    im->pix = HESS_PIX_U8;
    im->pixels_in_ctx = false;
    if (im->verbose) std::cout << "Image loaded :\t" << _imgpath << "\n";
  }
  _image_loaded = 1;
  if (!im->pending_keys.empty())
    hess_set_keypoints(im->ctx, im->pending_keys.data(), (int)im->pending_keys.size(), im->pending_keys_orient);
  const int pitch = im->w * channels(im->fmt) * pix_bytes(im->pix);
  const void* px = im->borrowed;
  if (!px) {  // RunSIFT() on the current image again: its pixels may still be with the context only
    if (!materialize_pixels(im)) { std::cerr << "SiftGPU: the current image's pixels are gone\n"; return 0; }
    px = im->pixels.data();
  }
  const int rc = hess_run_host(im->ctx, px, im->w, im->h, pitch, (size_t)pitch * im->h, 1, im->fmt, im->pix);
  if (rc != 0) {
    std::cerr << "SiftGPU: " << hess_last_error(im->ctx) << "\n";
    im->nfeat = 0;
    im->keys.clear();
    im->desc.clear();
    return 0;  // device errors -> 0 (SiftPyramid.h:162-163); oversize image is an error return here, not exit()
  }
  collect_results(this);
  if (im->verbose) {
    std::cout << "#Features:\t" << im->nfeat << "\n";
    if (im->timingS) std::cout << "RUN SIFT:\t" << _timing[TIMINGS_TOTAL] << "ms\n";
    std::cout << std::endl;
  }
  im->pending_keys.clear();
  if (_outpath[0]) { SaveSIFT(_outpath); _outpath[0] = 0; }
  return 1;
}

int SiftGPU::GetFeatureNum() { return I(_pyramid)->nfeat; }

void SiftGPU::GetFeatureVector(SiftKeypoint* keys, float* descriptors) {  // SiftPyramid.cpp:313-324
  Impl* im = I(_pyramid);
  static_assert(sizeof(SiftKeypoint) == sizeof(hess_keypoint), "keypoint layout");
  if (im->results_in_ctx && im->ctx) {  // straight from the context's pinned result buffers into the caller's arrays
    if (im->nfeat) hess_fetch(im->ctx, 0, reinterpret_cast<hess_keypoint*>(keys), im->dim ? descriptors : nullptr);
    return;
  }
  const size_t n = (size_t)(im->nfeat > 0 ? im->nfeat : 0);
  if (keys && n && im->keys.size() >= n) memcpy(keys, im->keys.data(), n * sizeof(SiftKeypoint));
  // The reference always copies 128*n floats, over-reading its 64*n buffer in -half mode; here dim*n.
  if (descriptors && im->dim && n && im->desc.size() >= n * im->dim) memcpy(descriptors, im->desc.data(), n * im->dim * sizeof(float));
}

void SiftGPU::SaveSIFT(const char* szFileName) {  // SiftPyramid::SaveSIFT, SiftPyramid.cpp:357-571
  Impl* im = I(_pyramid);
  if (im->nfeat <= 0) return;
  materialize_results(im);
  const int n = im->nfeat, dim = im->dim;
  const hess_keypoint* pk = im->keys.data();
  const float* pd = im->desc.data();
  if (im->binary_sift == 2) {  // vlfeat-style binary
    std::ofstream out(szFileName, std::ios::binary);
    out.write("aff\1", 4);
    out.write((const char*)&n, sizeof(int));
    out.write((const char*)&dim, sizeof(int));
    int iw = im->w >> 0, ih = im->h;
    out.write((const char*)&iw, sizeof(int));
    out.write((const char*)&ih, sizeof(int));
    for (int i = 0; i < n; i++, pk++) {
      const unsigned int lt = ((unsigned)pk->level << 2) | pk->type;
      const float scale = pk->s * im->mr_size;
      const float a11 = cosf(pk->o), a12 = -sinf(pk->o), a21 = sinf(pk->o), a22 = cosf(pk->o);
      out.write((const char*)&pk->x, 4); out.write((const char*)&pk->y, 4); out.write((const char*)&scale, 4);
      out.write((const char*)&a11, 4); out.write((const char*)&a12, 4); out.write((const char*)&a21, 4); out.write((const char*)&a22, 4);
      out.write((const char*)&lt, 4);
      out.write((const char*)&pk->response, 4);
      for (int k = 0; k < dim; k++, pd++) {
        const unsigned char v = (unsigned char)floor(0.5f + 255.0f * (*pd));
        out.write((const char*)&v, 1);
      }
    }
  } else if (im->binary_sift) {
    std::ofstream out(szFileName, std::ios::binary);
    out.write((const char*)&n, sizeof(int));
    out.write((const char*)&dim, sizeof(int));
    for (int i = 0; i < n; i++, pk++) {
      out.write((const char*)&pk->y, 4); out.write((const char*)&pk->x, 4);
      out.write((const char*)&pk->s, 4); out.write((const char*)&pk->o, 4);
      out.write((const char*)&pk->response, 4);
      out.write((const char*)&pk->type, 2); out.write((const char*)&pk->level, 2);
      if (dim) { out.write((const char*)pd, dim * sizeof(float)); pd += dim; }
    }
  } else {
    // (formatted in memory and written once: the reference's std::endl after every line is a write system call per line --
    //  33 k of them for 4096 features, 6 ms per 1080p image -- the bytes are the same)
    std::ostringstream out;
    out.flags(std::ios::fixed);
    out << n << " " << dim << std::endl;
    for (int i = 0; i < n; i++, pk++) {
      out << std::setprecision(2) << pk->y << " " << std::setprecision(2) << pk->x << " "
          << std::setprecision(3) << pk->s << " " << std::setprecision(3) << pk->o;
      out << " " << std::setprecision(8) << pk->response;
      out << " " << pk->type << " " << pk->level;
      out << std::endl;
      if (dim && im->p.normalize) {
        // the 128 integers of a descriptor: decimal digits written by hand (the stream's integer formatting of half a
        // million numbers per 1080p image took 4 ms; the characters are the same: an unsigned value, one blank, a
        // newline after every 20th)
        char line[128 * 12 + 16];
        char* q = line;
        for (int k = 0; k < dim; k++, pd++) {
          unsigned int v = (unsigned int)floor(0.5 + 512.0f * (*pd));
          char digits[12];
          int nd = 0;
          do { digits[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
          while (nd) *q++ = digits[--nd];
          *q++ = ' ';
          if ((k + 1) % 20 == 0) *q++ = '\n';
        }
        *q++ = '\n';
        out.write(line, q - line);
      } else if (dim) {
        for (int k = 0; k < dim; k++, pd++) {
          out << std::setprecision(8) << pd[0] << " ";
          if ((k + 1) % 20 == 0) out << std::endl;
        }
        out << std::endl;
      }
    }
    const std::string text = out.str();
    std::ofstream file(szFileName);
    file.write(text.data(), (std::streamsize)text.size());
  }
}

// ------------------------------------------------------------------------------------------------
// SiftMatchGPU on the matcher entry points of the C ABI (reference: SiftMatch.cpp's dispatcher +
// SiftMatchCU.cpp).  __matcher holds the hess_matcher handle.

static hess_matcher* MH(SiftMatchGPU* m_as_ptr) { return reinterpret_cast<hess_matcher*>(m_as_ptr); }

SiftMatchGPU::SiftMatchGPU(int max_sift) : __max_sift(max_sift), __language(0), __matcher(nullptr) {}
SiftMatchGPU::~SiftMatchGPU() {
  if (__matcher) hess_matcher_destroy(MH(__matcher));
}
int SiftMatchGPU::_CreateContextGL() { return _VerifyContextGL(); }
int SiftMatchGPU::_VerifyContextGL() {
  if (!__matcher) __matcher = reinterpret_cast<SiftMatchGPU*>(hess_matcher_create(__language >= 3 ? __language - 3 : 0, __max_sift));
  return __matcher ? 1 : 0;
}
void SiftMatchGPU::SetLanguage(int language) { __language = language; }  // SIFTMATCH_CUDA_DEVICE0 + i selects device i
void SiftMatchGPU::SetDeviceParam(int argc, char** argv) {
  for (int i = 0; i + 1 < argc; i++)
    if (!strcasecmp(argv[i], "-cuda")) { int d = 0; if (sscanf(argv[i + 1], "%d", &d) == 1 && d >= 0) __language = 3 + d; }
}
void SiftMatchGPU::SetMaxSift(int max_sift) {
  __max_sift = max_sift;
  if (__matcher) hess_matcher_set_max(MH(__matcher), max_sift);
}
void SiftMatchGPU::SetDescriptors(int index, int num, const float* descriptors, int id) {
  (void)id;
  if (_VerifyContextGL()) hess_matcher_set_descriptors_f32(MH(__matcher), index, num, descriptors);
}
void SiftMatchGPU::SetDescriptors(int index, int num, const unsigned char* descriptors, int id) {
  (void)id;
  if (_VerifyContextGL()) hess_matcher_set_descriptors(MH(__matcher), index, num, descriptors);
}
int SiftMatchGPU::GetSiftMatch(int max_match, int match_buffer[][2], float distmax, float ratiomax, int mbm) {
  if (!__matcher) return 0;
  const int n = hess_matcher_match(MH(__matcher), max_match, &match_buffer[0][0], nullptr, nullptr, distmax, ratiomax, 0, 0, mbm);
  return n < 0 ? 0 : n;
}
void SiftMatchGPU::SetFeautreLocation(int index, const float* locations, int gap) {
  if (__matcher) hess_matcher_set_locations(MH(__matcher), index, locations, gap);
}
int SiftMatchGPU::GetGuidedSiftMatch(int max_match, int match_buffer[][2], float H[3][3], float F[3][3], float distmax,
                                     float ratiomax, float hdistmax, float fdistmax, int mbm) {
  if (!__matcher) return 0;
  // SiftMatchGPU::GetGuidedSiftMatch (SiftMatch.cpp:663-676): no matrix at all -> plain match; a missing
  // one is replaced by the identity with a distance bound of 1e20
  if (!H && !F) return GetSiftMatch(max_match, match_buffer, distmax, ratiomax, mbm);
  static const float Z[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const float ti = 1.0e+20F;
  const int n = hess_matcher_match(MH(__matcher), max_match, &match_buffer[0][0], H ? &H[0][0] : Z, F ? &F[0][0] : Z,
                                   distmax, ratiomax, H ? hdistmax : ti, F ? fdistmax : ti, mbm);
  return n < 0 ? 0 : n;
}
void* SiftMatchGPU::operator new(size_t size) {
  void* p = malloc(size);
  if (!p) throw std::bad_alloc();
  return p;
}
void* ComboSiftGPU::operator new(size_t size) {
  void* p = malloc(size);
  if (!p) throw std::bad_alloc();
  return p;
}

// ------------------------------------------------------------------------------------------------
// Factories (SiftGPU.h:364-379) and the flat C mirror.

extern "C" {

SiftGPU* CreateNewSiftGPU(int np) { return new SiftGPU(np); }
SiftMatchGPU* CreateNewSiftMatchGPU(int max_sift) { return new SiftMatchGPU(max_sift); }
ComboSiftGPU* CreateComboSiftGPU() { return new ComboSiftGPU(); }
ComboSiftGPU* CreateRemoteSiftGPU(int, char*) { return nullptr; }

void siftgpu_destroy(SiftGPU* s) { delete s; }
void siftgpu_parse_param(SiftGPU* s, int argc, char** argv) { s->ParseParam(argc, argv); }
int siftgpu_create_context(SiftGPU* s) { return s->CreateContextGL(); }
int siftgpu_run_data(SiftGPU* s, int w, int h, const void* data, unsigned f, unsigned t) { return s->RunSIFT(w, h, data, f, t); }
int siftgpu_run_file(SiftGPU* s, const char* path) { return s->RunSIFT(path); }
int siftgpu_run_index(SiftGPU* s, int index) { return s->RunSIFT(index); }
int siftgpu_feature_num(SiftGPU* s) { return s->GetFeatureNum(); }
void siftgpu_feature_vector(SiftGPU* s, SiftGPU::SiftKeypoint* keys, float* desc) { s->GetFeatureVector(keys, desc); }
void siftgpu_save(SiftGPU* s, const char* path) { s->SaveSIFT(path); }
const float* siftgpu_timing(SiftGPU* s) { return s->_timing; }
void siftgpu_set_verbose(SiftGPU* s, int v) { s->SetVerbose(v); }
int siftgpu_image_count(SiftGPU* s) { return s->GetImageCount(); }
SiftMatchGPU* siftmatch_create(int max_sift) {
  SiftMatchGPU* m = new SiftMatchGPU(max_sift);
  m->VerifyContextGL();
  return m;
}
void siftmatch_destroy(SiftMatchGPU* m) { delete m; }
void siftmatch_set_descriptors_f32(SiftMatchGPU* m, int index, int num, const float* d) { m->SetDescriptors(index, num, d, -1); }
int siftmatch_get_match(SiftMatchGPU* m, int max_match, int* buf, float distmax, float ratiomax, int mbm) {
  return m->GetSiftMatch(max_match, reinterpret_cast<int(*)[2]>(buf), distmax, ratiomax, mbm);
}
int siftgpu_run_keys(SiftGPU* s, int num, const SiftGPU::SiftKeypoint* keys, int have_orientation) { return s->RunSIFT(num, keys, have_orientation); }
void siftgpu_set_keys(SiftGPU* s, int num, const SiftGPU::SiftKeypoint* keys, int have_orientation) { s->SetKeypointList(num, keys, have_orientation); }

}  // extern "C"

// siftgpu_get_params / siftgpu_descriptor_dim need the protected _pyramid: a friend-free accessor.
namespace {
struct Peek : SiftGPU {
  static Impl* impl(SiftGPU* s) { return I(static_cast<Peek*>(s)->_pyramid); }
  static void sync(SiftGPU* s) {
    Peek* p = static_cast<Peek*>(s);
    Impl* im = I(p->_pyramid);
    SiftParam tmp;
    tmp._dog_level_num = p->_dog_level_num; tmp._sigma0 = p->_sigma0; tmp._sigman = p->_sigman;
    tmp._dog_threshold = p->_dog_threshold; tmp._edge_threshold = p->_edge_threshold;
    tmp.ParseSiftParam();
    im->p.dog_level_num = tmp._dog_level_num; im->p.sigma0 = tmp._sigma0; im->p.sigman = tmp._sigman;
    im->p.dog_threshold = tmp._dog_threshold; im->p.edge_threshold = tmp._edge_threshold;
    delete[] tmp._sigma;
    tmp._sigma = nullptr;
  }
};
}  // namespace

extern "C" int siftgpu_get_params(SiftGPU* s, void* out) {
  if (!s || !out) return -1;
  Peek::sync(s);
  memcpy(out, &Peek::impl(s)->p, sizeof(hess_params));
  return 0;
}
extern "C" int siftgpu_descriptor_dim(SiftGPU* s) { return s ? Peek::impl(s)->dim : -1; }

// Test hook: decode an image file as RunSIFT(path) would (PNG, JPEG, PNM), without a device.  Returns the HESS_FMT_* of
// the decoded pixels (8 bits per channel) or <= 0; *w, *h are set; the pixels are copied when they fit `cap` bytes.
extern "C" int siftgpu_debug_load_image(const char* path, unsigned char* out, size_t cap, int* w, int* h) {
  std::vector<unsigned char> px;
  int iw = 0, ih = 0, fmt = load_png(path, px, iw, ih);
  if (fmt == 0) fmt = load_jpeg(path, px, iw, ih);
  if (fmt == 0 && load_pnm(path, px, iw, ih)) fmt = HESS_FMT_LUM;
  if (fmt <= 0) return fmt;
  if (w) *w = iw;
  if (h) *h = ih;
  if (out && px.size() <= cap) memcpy(out, px.data(), px.size());
  return fmt;
}
