// hess_plan.hip -- parameters, sigma schedule, octave geometry and the buffers of a batch shape (see hess_ctx.h).
#include "hess_ctx.h"

namespace hess {

void set_err(hess_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  try { c->err = buf; } catch (...) {}  // (nothing thrown crosses the C ABI; the message is then the previous one)
  if (c->p.verbose & 1) fprintf(stderr, "hessgpu: %s\n", buf);
}


int ensure(hess_ctx* c, DevBuf& b, size_t bytes, bool pinned_host) {
  if (bytes <= b.bytes) return 0;
  if (pinned_host && c->share_dir && (&b == &c->h_keys || &b == &c->h_desc)) return ensure_shared(c, b, bytes, &b == &c->h_keys ? 'k' : 'd');
  // the new buffer first: when the allocation fails the old one is still there (a context survives a refused
  // hess_reserve)
  const size_t want = bytes + bytes / 8;  // slack so slightly larger inputs do not reallocate
  void* np = nullptr;
  if (pinned_host) HIP_TRY(c, hipHostMalloc(&np, want, hipHostMallocDefault));
  else HIP_TRY(c, hipMalloc(&np, want));
  if (b.p) { if (pinned_host) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
  b.p = np;
  b.bytes = want;
  return 0;
}

void release(DevBuf& b, bool pinned_host) {
  if (b.p && !b.shm.empty()) {
    (void)hipHostUnregister(b.p);
    (void)munmap(b.p, b.bytes);
    if (strchr(b.shm.c_str() + 1, '/')) (void)unlink(b.shm.c_str());  // a file (the fallback), else a shared memory object
    else (void)shm_unlink(b.shm.c_str());
    b.shm.clear();
  } else if (b.p) {
    if (pinned_host) (void)hipHostFree(b.p); else (void)hipFree(b.p);
  }
  b.p = nullptr;
  b.bytes = 0;
}

// ---- parameters: GlobalUtil.cpp:51-144 defaults, SiftParam::ParseSiftParam SiftGPU.cpp:491-563 ----

void default_params(hess_params* p) {
  memset(p, 0, sizeof(*p));
  p->abi_version = HESS_ABI_VERSION;
  p->dog_level_num = 3;
  p->sigma0 = 1.6f;
  p->sigman = 0.5f;
  p->dog_threshold = 0.02f / 3;
  p->edge_threshold = 10.0f;
  p->filter_width_factor = 4.0f;
  p->orient_window_factor = 2.0f;
  p->orient_gaussian_factor = 1.5f;
  p->desc_window_factor = 3.0f;
  p->first_octave = 0;
  p->octave_num = -1;
  p->subpixel = 1;
  p->max_orientation = 2;
  p->compute_descriptors = 1;
  p->normalize = 1;
  p->truncate_method = HESS_TRUNC_HIGHEST_0;
  p->feature_count_threshold = -1;
  p->tex_max_dim = 3200;
  p->descriptor_order = HESS_DESC_ORDER_PIXEL;
}

// ProgramCU::CreateFilterKernel, ProgramCU.cu:423-453 (host arithmetic, libm expf).
void make_taps(const hess_params& p, float sigma, Taps* t) {
  int sz = (int)ceil(p.filter_width_factor * sigma - 0.5);
  int width = 2 * sz + 1;
  if (width > kMaxTaps) { sz = kMaxTaps >> 1; width = kMaxTaps; }
  else if (width < 5) { sz = 2; width = 5; }
  float rv = 1.0f / (sigma * sigma), v, ksum = 0;
  for (int i = -sz; i <= sz; ++i) {
    t->k[i + sz] = v = expf(-0.5f * i * i * rv);
    ksum += v;
  }
  rv = 1.0f / ksum;
  for (int i = 0; i < width; i++) t->k[i] *= rv;
  for (int i = width; i < kMaxTaps; i++) t->k[i] = 0.0f;
  t->fw = width;
}

void resolve(hess_ctx* c) {
  hess_params& p = c->p;
  if (p.dog_level_num == 0) p.dog_level_num = 3;
  if (p.sigma0 == 0.0f) p.sigma0 = 1.6f;
  if (p.sigman == 0.0f) p.sigman = 0.5f;
  if (p.filter_width_factor == 0.0f) p.filter_width_factor = 4.0f;
  if (p.orient_window_factor == 0.0f) p.orient_window_factor = 2.0f;
  if (p.orient_gaussian_factor == 0.0f) p.orient_gaussian_factor = 1.5f;
  if (p.desc_window_factor == 0.0f) p.desc_window_factor = 3.0f;
  if (p.tex_max_dim == 0) p.tex_max_dim = 3200;
  if (p.max_orientation < 1) p.max_orientation = 1;  // SiftGPU.cpp:1047
  if (p.max_orientation > 4) p.max_orientation = 4;
  Schedule& s = c->sch;
  s.dog = p.dog_level_num;
  s.level_max = s.dog + 1;
  s.level_num = s.level_max + 1;
  s.level_ds = s.dog;
  const float sigmak = powf(2.0f, 1.0f / p.dog_level_num);
  const float dsigma0 = p.sigma0 * sqrtf(sigmak * sigmak - 1.0f);
  for (int i = 1; i <= s.level_max; i++) {
    s.sigma[i - 1] = dsigma0 * powf(sigmak, (float)(i - 1));
    make_taps(p, s.sigma[i - 1], &s.taps[i]);
  }
  for (int l = 0; l <= s.level_max; l++) {
    s.level_sigma[l] = p.sigma0 * powf(2.0f, (float)l / (float)p.dog_level_num);
    const float ls = s.level_sigma[l] * 1.0f;  // octaveSigma = 1 (PyramidCU.cpp:1574-1585)
    const float n2 = ls * ls;                  // passed by DetectKeypointsEX
    s.norm[l] = n2 * n2;                       // squared again by ProgramCU::ComputeHessian (:592)
  }
  if (p.dog_threshold == 0.0f) p.dog_threshold = 0.02f / p.dog_level_num;
  if (p.edge_threshold == 0.0f) p.edge_threshold = 10.0f;
  s.sigma_step = powf(2.0f, 1.0f / p.dog_level_num);
  s.ln_sigma_step = (float)log((double)s.sigma_step);
}

float initial_smooth_sigma(const hess_ctx* c, int octave_min) {  // SiftGPU.cpp:482-489
  const float sa = c->p.sigma0 * powf(2.0f, 0.0f / (float)c->p.dog_level_num);
  const float sb = c->p.sigman / powf(2.0f, (float)octave_min);
  return (sa > sb + 0.001) ? sqrtf(sa * sa - sb * sb) : 0.0f;
}

int fmt_channels(int format) {
  switch (format) {
    case HESS_FMT_LUM: return 1;
    case HESS_FMT_LUM_ALPHA: return 2;
    case HESS_FMT_RGB: case HESS_FMT_BGR: return 3;
    case HESS_FMT_RGBA: case HESS_FMT_BGRA: return 4;
  }
  return 0;
}

// Geometry: SetImageData (GLTexImage.cpp:932-1033) + InitPyramid/ResizePyramid (PyramidCU.cpp:113-310).
int plan_inner(hess_ctx* c, int width, int height, int batch) {
  const hess_params& p = c->p;
  int ds = 0, ws = width, hs = height;
  if (p.first_octave > 0) { ds = p.first_octave; ws = width >> ds; hs = height >> ds; }
  else if (p.first_octave < 0) { ds = p.first_octave; ws = (width & ~3) << (-ds); hs = height << (-ds); }  // PyramidCU.cpp:120-138
  if (ws > p.tex_max_dim || hs > p.tex_max_dim) {
    if (!p.auto_downscale) {
      set_err(c, "image %dx%d exceeds max dimension %d (use -ads or -maxd)", ws, hs, p.tex_max_dim);
      return HESS_ERR_TOO_BIG;
    }
    // _octave_min++ until it fits (PyramidCU.cpp:154-166): an up-sampled first octave is up-sampled less, then not at
    // all, then decimated -- the same loop whatever the sign of the first octave
    do { ds++; ws >>= 1; hs >>= 1; } while (ws > p.tex_max_dim || hs > p.tex_max_dim);
  }
  ws &= ~3;  // TruncateWidthCU
  if (ws < 4 || hs < 1) { set_err(c, "image too small"); return HESS_ERR_ARG; }
  const bool same = c->planned && c->in_w == width && c->in_h == height && batch <= c->g.B &&
                    (int)(2 * c->user_keys.size() + 8) <= c->cap_sel && (int)(2 * c->user_keys.size() + 8) <= c->cap_feat;
  if (same) return 0;
  const int B = (c->planned && c->g.B > batch) ? c->g.B : batch;

  Geom g;
  memset(&g, 0, sizeof(g));
  const int input_sz = ws < hs ? ws : hs;
  int nmax = (int)floor(log((double)input_sz) / log(2.0)) - 3;  // PyramidCU.cpp:242
  if (nmax < 1) nmax = 1;
  if (nmax > kMaxOct) nmax = kMaxOct;
  g.noct = (p.octave_num >= 1 && p.octave_num < nmax) ? p.octave_num : nmax;
  g.dog = c->sch.dog;
  g.nlev = g.noct * g.dog;
  g.B = B;
  long long lvl = 0, gt = 0;
  int rows = 0, mw = 0;
  int w = ws, h = hs;
  for (int o = 0; o < g.noct; o++) {
    OctGeom& og = g.o[o];
    og.wa = ((w + 3) / 4) * 4;
    og.h = h;
    og.plane = og.wa * og.h;
    og.w64 = (og.wa + 63) / 64;
    og.lvl_off = lvl;
    og.got_off = gt;
    og.row_base = rows;
    og.mask_base = mw;
    og.tiles_x = (og.wa + 127) / 128;  // EX_TC columns per extrema tile (k_detect.hip)
    og.tile_base = g.ntiles;
    g.ntiles += og.tiles_x * ((og.h + 3) / 4);  // EX_TR rows per extrema tile (k_detect.hip)
    og.strips = (og.wa + kStreamPitch - 1) / kStreamPitch;
    lvl += (long long)c->sch.level_num * B * og.plane;
    gt += (long long)g.dog * B * og.plane;
    rows += g.dog * og.h;
    mw += g.dog * og.h * og.w64;
    w >>= 1;
    h >>= 1;
  }
  g.NR = rows;
  g.NM = mw;
  set_stream_rows(g, kStreamRows);

  c->use_topk = (p.truncate_method == HESS_TRUNC_TOPK && p.feature_count_threshold > 0);
  c->multi = (p.max_orientation > 1) && !p.fixed_orientation;  // SiftPyramid.cpp:140
  c->dim = p.compute_descriptors ? (p.half_sift ? 64 : 128) : 0;
  long long det_px = gt / B;  // detection pixels per image
  int cap_raw = (int)(det_px / 32 < 16384 ? 16384 : det_px / 32);
  if (c->cap_init > 0) cap_raw = c->cap_init;  // developer switch: start small so that the grow-and-re-run path is taken
  if (cap_raw < c->cap_raw) cap_raw = c->cap_raw;
  if (cap_raw < (int)(2 * c->user_keys.size() + 8)) cap_raw = (int)(2 * c->user_keys.size() + 8);
  int cap_sel = c->use_topk ? p.feature_count_threshold : cap_raw;
  if (cap_sel > cap_raw) cap_sel = cap_raw;
  int cap_feat = c->multi ? (c->use_topk ? 4 * cap_sel : cap_sel) : cap_sel;
  if (cap_feat < c->cap_feat) cap_feat = c->cap_feat;
  if (!c->user_keys.empty()) {  // a keypoint list bypasses top-K: every stage must hold 2*num+8 records
    const int need = (int)(2 * c->user_keys.size() + 8);
    if (cap_sel < need) cap_sel = need;
    if (cap_feat < need) cap_feat = need;
  }

  int rc;
  if ((rc = ensure(c, c->gauss, (size_t)lvl * 4))) return rc;
  if ((rc = ensure(c, c->deth, (size_t)lvl * 4))) return rc;
  if ((rc = ensure(c, c->got, (size_t)gt * 8))) return rc;
  {
    const int up = ds < 0 ? -ds : 0;  // the converted input is held at its own size; the up-sampled copy in `upsampled`
    if ((rc = ensure(c, c->input_f32, (size_t)B * (ws >> up) * (hs >> up) * 4))) return rc;
  }
  if (ds < 0 && (rc = ensure(c, c->upsampled, (size_t)B * ws * hs * 4))) return rc;
  {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_found = 256, o_cnt = o_found + up((size_t)3 * B * 4), o_hist = o_cnt + up((size_t)B * g.NR * 4);
    const size_t o_mask = o_hist + (c->use_topk ? up((size_t)B * kHistBins * 4) : 0);
    const size_t o_tk = o_mask + up((size_t)B * g.NM * 8);
    c->zeroed_used = o_tk + (c->use_topk ? up(topk_scratch_bytes(cap_raw, B, g.nlev)) : 0);
    if ((rc = ensure(c, c->zeroed, c->zeroed_used))) return rc;
    char* z = (char*)c->zeroed.p;
    c->overflow.p = z; c->rowcnt.p = z + o_cnt; c->hist.p = z + o_hist; c->rowmask.p = z + o_mask; c->tk.p = z + o_tk;
    c->found_count.p = z + o_found; c->place_ticket.p = z + o_found + (size_t)B * 4; c->place_flag.p = z + o_found + (size_t)2 * B * 4;
  }
  if ((rc = ensure(c, c->rowoff, (size_t)B * g.NR * 4))) return rc;
  if ((rc = ensure(c, c->level_count, (size_t)B * g.nlev * 4))) return rc;
  if ((rc = ensure(c, c->raw_total, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->sel_total, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->feat_total, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->feat_first, (size_t)B * 4))) return rc;
  if ((rc = ensure(c, c->img_base, (size_t)(B + 1) * 4))) return rc;
  if ((rc = ensure(c, c->raw, (size_t)B * cap_raw * sizeof(RawKey)))) return rc;
  {  // the scan's unordered detections: every scan task's slots + the image's spill list (hess_dev.h, DetectStore);
     // tasks for the shortest segments a batch may be scanned with (enqueue(): batches of one or two images, HESS_STREAM_ROWS)
    int ntask = extrema_tasks(g);
    for (int rows : {kStreamRows / policy::kLatencyStreamRowsDiv, c->stream_rows}) {
      if (rows <= 0) continue;
      Geom gr = g;
      set_stream_rows(gr, rows);
      ntask = std::max(ntask, extrema_tasks(gr));
    }
    c->found_tasks = ntask;
    if ((rc = ensure(c, c->found, (size_t)B * ((size_t)ntask * kDetectSlots + cap_raw) * sizeof(RawKey)))) return rc;
    if ((rc = ensure(c, c->task_count, (size_t)B * ntask * 4))) return rc;
  }
  if (c->use_topk) {
    if ((rc = ensure(c, c->sel, (size_t)B * cap_sel * sizeof(RawKey)))) return rc;
  }
  if ((rc = ensure(c, c->recs, (size_t)B * cap_sel * sizeof(FRec)))) return rc;
  if ((rc = ensure(c, c->ocount, (size_t)B * cap_sel * 4))) return rc;
  if ((rc = ensure(c, c->foffset, (size_t)B * cap_sel * 4))) return rc;
  if ((rc = ensure(c, c->fsrc, (size_t)B * cap_feat * 4))) return rc;
  if ((rc = ensure(c, c->keys, (size_t)B * cap_feat * sizeof(HostKeypoint)))) return rc;
  if (c->dim && (rc = ensure(c, c->desc, (size_t)B * cap_feat * c->dim * 4))) return rc;
  if ((rc = ensure(c, c->h_small, (size_t)(3 * B + 8) * 4, true))) return rc;
  {
    // The pinned result buffers hold the worst case B * cap_feat records up front while that stays moderate; beyond
    // it they grow on demand once the counts are known (wait_impl / the copier), and the in-kernel mirror is not used.
    const size_t host_bytes = (size_t)B * cap_feat * (sizeof(HostKeypoint) + (size_t)c->dim * 4);
    // Node-shared result buffers of a batch the copier delivers are sized by the batches seen (+ 25 %), not for the
    // worst case B * cap_feat: 79 MB per context, 3.8 - 4.4 GB of /dev/shm for a node's six or seven contexts x eight
    // ranks, where the results are 24 MB per context; the copier (or hess_wait) grows them under a new generation when a batch needs more.
    c->share_by_need = c->share_dir && B > c->mirror_max_batch && c->delivery_pref != kDeliverMirror;
    c->host_fits = !c->share_by_need && host_bytes <= policy::kHostWorstCaseMax;
    if (c->host_fits) {
      if ((rc = ensure(c, c->h_keys, (size_t)B * cap_feat * sizeof(HostKeypoint), true))) return rc;
      if (c->dim && (rc = ensure(c, c->h_desc, (size_t)B * cap_feat * c->dim * 4, true))) return rc;
    }
  }

  c->g = g;
  c->ds = ds;
  c->img_w = ws;
  c->img_h = hs;
  c->in_w = width;
  c->in_h = height;
  c->cap_raw = cap_raw;
  c->cap_sel = cap_sel;
  c->cap_feat = cap_feat;
  c->planned = true;
  const float s0 = initial_smooth_sigma(c, ds);
  c->has_taps0 = s0 > 0.0f;
  if (c->has_taps0) make_taps(p, s0, &c->taps0);
  return 0;
}

// A plan that fails half way (an allocation was refused) leaves buffers of mixed sizes behind: the next run plans
// again from scratch (buffers that are large enough are kept), so the context stays usable.
int plan(hess_ctx* c, int width, int height, int batch) {
  const int rc = plan_inner(c, width, height, batch);
  if (rc) {
    c->planned = false;
    (void)hipGetLastError();  // the refused allocation must not be reported by the next call's error check
  }
  return rc;
}


}  // namespace hess
