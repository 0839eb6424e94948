// hess_schedule.hip -- the launch order of one batch on the context's stream (see hess_ctx.h).
#include <optional>

#include "hess_ctx.h"

namespace hess {

// ---- profiling helpers ----
hipEvent_t get_event(hess_ctx* c) {
  if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) { set_err(c, "hipEventCreate failed: profiling disabled"); c->prof = false; return nullptr; }
  return e;
}
struct ProfScope {
  hess_ctx* c;
  EventPair ep;
  bool on;
  ProfScope(hess_ctx* ctx, int kernel, double bytes, int kernel2 = -1, double in_lds = 0.0) : c(ctx), on(ctx->prof) {
    if (!on) return;
    ep.a = get_event(c); ep.b = get_event(c); ep.kernel = kernel; ep.bytes = bytes; ep.kernel2 = kernel2; ep.in_lds = in_lds;
    if (!ep.a || !ep.b) {  // event creation failed: no record for this launch
      if (ep.a) c->pool.push_back(ep.a);
      if (ep.b) c->pool.push_back(ep.b);
      on = false;
      return;
    }
    if (hipEventRecord(ep.a, c->st) != hipSuccess) { c->pool.push_back(ep.a); c->pool.push_back(ep.b); on = false; }
  }
  ~ProfScope() {
    if (!on) return;
    if (hipEventRecord(ep.b, c->st) != hipSuccess) { c->pool.push_back(ep.a); c->pool.push_back(ep.b); return; }
    c->pending.push_back(ep);
  }
};
// A RUN of consecutive launches of one kernel family between ONE pair of events: an event record between two kernels leaves
// the stream idle for a few microseconds, which per-launch pairs book on 22 pyramid launches a step (the hipEvent figure of
// the Gaussian stage read 0.53 ms where the kernel trace sums 0.47).  The launches of octave 0 keep a pair each (their
// share of the stage is reported apart); the small octaves' launches, which follow them back to back, share one.
struct ProfRun {
  hess_ctx* c;
  int kernel;
  EventPair ep;
  bool open = false;
  ProfRun(hess_ctx* ctx, int k) : c(ctx), kernel(k) {}
  void add(double bytes, double in_lds = 0.0) {
    if (!c->prof) return;
    if (!open) {
      ep = EventPair{};
      ep.a = get_event(c); ep.b = get_event(c); ep.kernel = kernel; ep.bytes = 0.0; ep.count = 0;
      if (!ep.a || !ep.b || hipEventRecord(ep.a, c->st) != hipSuccess) {
        if (ep.a) c->pool.push_back(ep.a);
        if (ep.b) c->pool.push_back(ep.b);
        return;
      }
      open = true;
    }
    ep.bytes += bytes; ep.in_lds += in_lds; ep.count++;
  }
  void close() {
    if (!open) return;
    open = false;
    if (hipEventRecord(ep.b, c->st) != hipSuccess) { c->pool.push_back(ep.a); c->pool.push_back(ep.b); return; }
    c->pending.push_back(ep);
  }
  ~ProfRun() { close(); }
};
void drain_profile(hess_ctx* c) {
  for (auto& ep : c->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
      c->k_ms[ep.kernel] += ms;
      c->k_n[ep.kernel] += ep.count;
      c->k_bytes[ep.kernel] += ep.bytes;
      c->k_in_lds[ep.kernel] += ep.in_lds;
      if (ep.kernel2 >= 0) {
        c->k_ms[ep.kernel2] += ms; c->k_n[ep.kernel2] += ep.count; c->k_bytes[ep.kernel2] += ep.bytes; c->k_in_lds[ep.kernel2] += ep.in_lds;
      }
    }
    c->pool.push_back(ep.a);
    c->pool.push_back(ep.b);
  }
  c->pending.clear();
}

// 2^_octave_min: scale of the first octave relative to the input (PyramidCU.cpp:566-569,746-748,1054-1057).
static inline float first_octave_sigma(const hess_ctx* c) {
  return c->ds > 0 ? (float)(1 << c->ds) : (c->ds < 0 ? 1.0f / (float)(1 << (-c->ds)) : 1.0f);
}

int enqueue_user(hess_ctx* c);

// Enqueue the whole path for `batch` images whose pixels are at device address `dev`.
int enqueue(hess_ctx* c, const void* dev, int pitch, size_t image_stride, int batch, int format, int pixtype) {
  const hess_params& p = c->p;
  const Schedule& s = c->sch;
  const Geom& g = c->g;
  hipStream_t st = c->st;
  float* gauss = (float*)c->gauss.p;
  float* deth = (float*)c->deth.p;
  float* got = (float*)c->got.p;
  auto plane_ptr = [&](float* base, int o, int l) { return base + g.o[o].lvl_off + (long long)l * g.B * g.o[o].plane; };

  // Stage timers (SiftGPU::_timing[2..10]) only when asked for: hess_params.verbose bit 1 (the reference's _timingS,
  // SiftGPU.cpp:433-464: its stage times, too, are only meaningful when it synchronises at stage ends) or the bench's
  // per-kernel profile.  An event record between two kernels leaves the stream idle for about 6 us.
  c->stage_events = (c->p.verbose & 2) != 0;
  HIP_TRY(c, hipEventRecord(c->ev[0], st));
  const bool user_mode = !c->user_keys.empty();
  DetectParams dp;
  dp.thr = p.dog_threshold;
  dp.thr0 = (p.subpixel ? 0.8f : 1.0f) * p.dog_threshold;                            // ProgramCU.cu:897
  dp.edge = (p.edge_threshold + 1) * (p.edge_threshold + 1) / p.edge_threshold;      // ProgramCU.cu:913
  dp.subpixel = p.subpixel;
  Geom gx = g;  // the extrema scan's segment length
  if (c->stream_rows > 0) set_stream_rows(gx, c->stream_rows);      // HESS_STREAM_ROWS (A/B switch; a multiple of 3)
  else if (batch <= policy::kLatencyBatch) set_stream_rows(gx, kStreamRows / policy::kLatencyStreamRowsDiv);  // shorter segments, twice the wavefronts
  if (!(user_mode && c->user_on_current)) {  // SIFT_SKIP_FILTERING: the resident pyramid is reused
  // ---- input + pyramid (BuildPyramid, PyramidCU.cpp:1486-1558) ----
  const bool direct_u8 = (format == HESS_FMT_LUM && pixtype == HESS_PIX_U8 && c->ds == 0 && c->has_taps0 &&
                          (pitch % 4) == 0 && (image_stride % 4) == 0 && ((uintptr_t)dev % 4) == 0);
  const float* src_f = nullptr;
  if (!direct_u8) {
    ProfScope ps(c, HESS_K_INPUT, (double)batch * c->img_w * c->img_h * (4.0 + fmt_channels(format)));
    const int up = c->ds < 0 ? -c->ds : 0;  // up-sampled first octave: convert at full size, then SampleImageU
    launch_convert(st, dev, format, pixtype, pitch, (long long)image_stride, up ? 0 : c->ds, (float*)c->input_f32.p,
                   c->img_w >> up, c->img_h >> up, batch);
    src_f = (const float*)c->input_f32.p;
    if (up) {  // PyramidCU.cpp:1521-1522
      launch_upsample(st, src_f, c->img_w >> up, c->img_h >> up, up, (float*)c->upsampled.p, batch);
      src_f = (const float*)c->upsampled.p;
    }
  }
  // The launch that produces the down-sampling level also writes level 0 of the next octave (its even rows and
  // columns): no decimation launches.  (A down-sampling level 0 is nobody's product: separate kernel then.)
  const bool fused_decim = s.level_ds >= 1 && s.level_ds <= s.level_max;
  // det-H of the top level by the launch that produces it (k_gauss.hip, TOP tiles); HESS_NO_TOP_FUSION=1: the round-4
  // form (the level is stored, hessian_rows4_kernel reads it back) for A/B runs
  const bool top_fused = s.level_max == g.dog + 1 && s.level_max >= 1 && !c->no_top_fusion;
  c->zero_filled = false;
  // Level l of octave o from level l-1; the same launch emits det-H (+ gradient/theta) of level l-1 from the source
  // window it stages: 8 B R+W for the blur, 4 B (+8 B) W for the fused planes (+ 4 B per pixel of the next octave's
  // level 0 when it is the down-sampling level).
  auto level_job = [&](int o, int l) {
    const OctGeom& og = g.o[o];
    const bool src_got = (l - 1 >= 1 && l - 1 <= g.dog);
    const bool decim = fused_decim && l == s.level_ds && o + 1 < g.noct;
    GaussJob j;
    j.src = plane_ptr(gauss, o, l - 1); j.dst = plane_ptr(gauss, o, l); j.wa = og.wa; j.h = og.h; j.taps = s.taps[l];
    j.deth_src = plane_ptr(deth, o, l - 1);
    j.got_src = src_got ? got + 2 * (og.got_off + (long long)(l - 2) * g.B * og.plane) : nullptr;
    j.norm_src = s.norm[l - 1];
    j.decim_dst = decim ? plane_ptr(gauss, o + 1, 0) : nullptr;
    j.decim_w = decim ? g.o[o + 1].wa : 0; j.decim_h = decim ? g.o[o + 1].h : 0;
    if (top_fused && l == s.level_max) {
      // The octave's top level is nobody's source: its det-H comes out of the launch that produces it (from the output
      // tile in LDS) and the level itself is not written to HBM -- unless the parity tests ask (hess_debug_keep_levels).
      j.deth_dst = plane_ptr(deth, o, l);
      j.norm_dst = s.norm[l];
      if (!c->keep_levels) j.dst = nullptr;
      if (o == 0 && !user_mode) {  // octave 0's launch also clears what the detection stages expect zeroed
        j.zero = c->zeroed.p;
        j.zero_bytes = c->zeroed_used;
      }
    }
    return j;
  };
  // (a fused top level reads its source and writes its own det-H instead of the level: the same 8 bytes)
  auto level_bytes = [&](int o, int l) {
    const OctGeom& og = g.o[o];
    const bool src_got = (l - 1 >= 1 && l - 1 <= g.dog);
    const bool decim = fused_decim && l == s.level_ds && o + 1 < g.noct;
    return (double)batch * og.plane * (8.0 + 4.0 + (src_got ? 8.0 : 0.0)) + (decim ? (double)batch * g.o[o + 1].plane * 4.0 : 0.0);
  };
  // (SURVEY 8d books every array of the reference's layout written once and read once; a level this build keeps in LDS
  // -- the octave's top level, level 0 of octave 0 -- is 8 bytes per pixel of that layout which no launch here moves)
  auto level_in_lds = [&](int o, int l) {
    return (top_fused && l == s.level_max && !c->keep_levels) ? (double)batch * g.o[o].plane * 8.0 : 0.0;
  };
  auto launch_level = [&](const GaussJob& j) {
    if (j.zero) c->zero_filled = true;
    launch_gauss_job(st, j, batch);
  };
  // T(o, l) = 3o + l is the earliest step of level l of octave o (level 0 of octave o+1 is the decimated level_ds of
  // octave o): the top level of an octave and level 1 of the next are due together and independent, so they share a
  // launch (launch_gauss_pair) -- one launch fewer per octave in the dependent chain.
  const bool pair_levels = fused_decim && s.level_ds < s.level_max && s.level_max >= 2 && !c->no_pair;
  // A single image (or two): octaves from 1 on get levels 1..level_ds -- what the next octave waits for -- from ONE
  // launch each (gauss_chain_kernel: 32x32 tiles computed in LDS on a shrinking halo): below 960x540 a level launch is a
  // few dozen workgroups that mostly wait, and the eighteen of them for octaves 1-6 of a 1080p image were two thirds of
  // its pyramid's time.  The top levels (nobody's input) follow in one launch for all octaves, together with det-H /
  // gradient of the chained octaves' levels 0..level_ds-1.  Same box, one 1080p image, device-resident: 0.400 -> 0.334 ms.
  // NOT for larger batches: a batch of 8 is 2 % faster on one stream with octaves >= 2 chained, but six pipelined
  // contexts lose 2 - 4 % (15.5 - 15.6 against 16.1 - 16.2 Gpix/s, same call; 15.8 - 16.0 with octaves >= 3) -- the chain
  // trades dependent launches for redundant arithmetic in 1024-thread workgroups that wait at barriers, which is what
  // an idle device wants and a saturated one does not; and not for the large octaves of a large image (a 4096^2 image's
  // octave 1 is 4 096 such workgroups: configs[4] 2.19 against 2.12 ms).  Needs the default schedule's tap counts.
  // HESS_CHAIN_FROM=n forces the first chained octave (99: never).
  int chain_from = g.noct;
  if (fused_decim && s.level_max == s.level_ds + 1 && gauss_chain_available(s.taps, s.level_ds)) {
    chain_from = c->chain_from;
    if (chain_from <= 0) {  // by size: the first octave (>= 1) whose planes of the whole batch are at most two 960x540 planes
      chain_from = g.noct;
      // (a PAIR of images handed over by hess_submit_* -- a caller who pipelines -- gets the level-by-level launches and
      //  the copier's delivery like a larger batch: 17.0 - 17.3 against 12.3 - 12.6 Gpix/s for six pipelined contexts)
      if (batch == 1 || (batch <= policy::kLatencyBatch && c->caller_waits))
        for (int o = g.noct - 1; o >= 1 && (long long)batch * g.o[o].plane <= policy::kChainMaxPixels; o--) chain_from = o;
    }
    if (chain_from > g.noct) chain_from = g.noct;
  }
  const bool chained = chain_from < g.noct;
  ProfRun small_octaves(c, HESS_K_GAUSS);  // one event pair for the consecutive launches of octaves >= 1 (see ProfRun)
  GaussJob top_jobs[kMaxOct];  // the top levels of the octaves that do not ride with the next octave's level 1
  int ntop = 0;
  double top_bytes = 0.0, top_in_lds = 0.0;
  int deferred_o = -1;
  for (int o = 0; o < g.noct; o++) {
    const OctGeom& og = g.o[o];
    if (o >= chain_from && o >= 1) {  // (its level 0 is the decimated level_ds of octave o-1, written by that launch)
      ChainJob cj;
      double bytes = 0.0;
      cj.src0 = plane_ptr(gauss, o, 0);
      for (int l = 0; l <= s.level_ds; l++) {
        cj.dst[l] = plane_ptr(gauss, o, l);
        cj.taps[l] = s.taps[l];
        // (the bytes of the fused planes are booked here although hessian_low_levels writes them: per step the sums agree)
        if (l >= 1) bytes += level_bytes(o, l);
      }
      cj.nlevels = s.level_ds; cj.wa = og.wa; cj.h = og.h;
      const bool decim = o + 1 < g.noct;
      cj.decim_dst = decim ? plane_ptr(gauss, o + 1, 0) : nullptr;
      cj.decim_w = decim ? g.o[o + 1].wa : 0; cj.decim_h = decim ? g.o[o + 1].h : 0;
      {
        small_octaves.add(bytes);
        if (!launch_gauss_chain(st, cj, batch)) { set_err(c, "level-chain launch refused"); return HESS_ERR_DEVICE; }
      }
      top_jobs[ntop++] = level_job(o, s.level_max);
      top_bytes += level_bytes(o, s.level_max);
      top_in_lds += level_in_lds(o, s.level_max);
      continue;
    }
    bool first_fused = false;  // levels 0 and 1 of octave 0 came out of one launch (level 0 never written)
    if (o == 0 && direct_u8 && c->has_taps0 && !c->no_first_fusion && s.level_max >= 2 && s.level_ds != 1 && chain_from != 0 &&
        gauss_first_available(c->taps0, s.taps[1])) {  // (decided BEFORE the profile scope: a refused launch must not book bytes)
      // u8 pixels -> level 0 (LDS) -> level 1, det-H of level 0: the level-0 plane is nobody's input but level 1's
      const GaussJob j1 = level_job(0, 1);
      ProfScope ps(c, HESS_K_GAUSS, (double)batch * og.plane * (1.0 + 4.0 + 4.0), HESS_K_GAUSS_OCT0,
                   c->keep_levels ? 0.0 : (double)batch * og.plane * 8.0);
      first_fused = launch_gauss_first(st, (const uint8_t*)dev, pitch, (long long)image_stride, c->taps0, j1,
                                       c->keep_levels ? plane_ptr(gauss, 0, 0) : nullptr, batch);
    }
    if (o == 0) c->level0_in_lds = first_fused && !c->keep_levels;
    if (o == 0 && first_fused) {
      // (nothing: level 1 exists, the loop below starts at level 2)
    } else if (o == 0) {
      if (c->has_taps0) {
        ProfScope ps(c, HESS_K_GAUSS, (double)batch * og.plane * (direct_u8 ? 5.0 : 8.0), HESS_K_GAUSS_OCT0);
        if (direct_u8)
          launch_gauss(st, nullptr, (const uint8_t*)dev, pitch, (long long)image_stride, plane_ptr(gauss, 0, 0),
                       og.wa, og.h, batch, c->taps0);
        else
          launch_gauss(st, src_f, nullptr, og.wa, og.plane, plane_ptr(gauss, 0, 0), og.wa, og.h, batch, c->taps0);
      } else {
        HIP_TRY(c, hipMemcpyAsync(plane_ptr(gauss, 0, 0), src_f, (size_t)batch * og.plane * 4, hipMemcpyDeviceToDevice, st));
      }
    } else if (!fused_decim) {
      ProfScope ps(c, HESS_K_DOWNSAMPLE, (double)batch * og.plane * 8.0);
      launch_downsample(st, plane_ptr(gauss, o - 1, s.level_ds), g.o[o - 1].wa, g.o[o - 1].plane,
                        plane_ptr(gauss, o, 0), og.wa, og.h, batch);
    }
    for (int l = first_fused ? 2 : 1; l <= s.level_max; l++) {
      if (l == 1 && deferred_o >= 0) {  // the previous octave's top level rides with this octave's level 1
        const int top_o = deferred_o;
        {
          const GaussJob ja = level_job(top_o, s.level_max), jb = level_job(o, 1);
          std::optional<ProfScope> ps;
          if (top_o == 0) {
            small_octaves.close();
            ps.emplace(c, HESS_K_GAUSS, level_bytes(top_o, s.level_max) + level_bytes(o, 1), HESS_K_GAUSS_OCT0, level_in_lds(top_o, s.level_max));
          } else {
            small_octaves.add(level_bytes(top_o, s.level_max) + level_bytes(o, 1), level_in_lds(top_o, s.level_max));
          }
          if (launch_gauss_pair(st, ja, jb, batch)) c->zero_filled = c->zero_filled || ja.zero != nullptr;
          else { launch_level(ja); launch_level(jb); }
        }
        deferred_o = -1;
        continue;
      }
      if (l == s.level_max && chained && o + 1 >= chain_from) {  // with the chained octaves' top levels, after the chain
        top_jobs[ntop++] = level_job(o, l);
        top_bytes += level_bytes(o, l);
        top_in_lds += level_in_lds(o, l);
        continue;
      }
      if (l == s.level_max && pair_levels && o + 1 < g.noct) { deferred_o = o; continue; }
      std::optional<ProfScope> ps;
      if (o == 0) { small_octaves.close(); ps.emplace(c, HESS_K_GAUSS, level_bytes(o, l), HESS_K_GAUSS_OCT0, level_in_lds(o, l)); }
      else small_octaves.add(level_bytes(o, l), level_in_lds(o, l));
      launch_level(level_job(o, l));
    }
  }
  if (deferred_o >= 0) {  // (cannot happen: the last octave never defers)
    small_octaves.add(level_bytes(deferred_o, s.level_max), level_in_lds(deferred_o, s.level_max));
    launch_level(level_job(deferred_o, s.level_max));
  }
  if (ntop) {  // the top levels left over by the chain, one launch
    small_octaves.add(top_bytes, top_in_lds);
    // (+ det-H / gradient of levels 0..level_ds-1 of the chain-launched octaves, from HBM: hessian_low_levels)
    LowLevels low{&g, gauss, deth, got, s.norm, chain_from, chained ? s.level_ds : 0};
    if (launch_gauss_multi(st, top_jobs, ntop, batch, &low)) {
      for (int k = 0; k < ntop; k++) c->zero_filled = c->zero_filled || top_jobs[k].zero != nullptr;
    } else {
      for (int k = 0; k < ntop; k++) launch_level(top_jobs[k]);
      if (chained)  // (not reached with the schedules the chain is instantiated for)
        for (int o = chain_from; o < g.noct; o++) launch_hessian(st, g, o, gauss, deth, got, s.norm, batch, 0, s.level_ds - 1);
    }
  }
  small_octaves.close();
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[1], st));
  // ---- det-Hessian + gradient (DetectKeypointsEX part 1, PyramidCU.cpp:1576-1591) ----
  // (levels 0 .. level_max-1: by the launch that reads the level as its source; the top level: by the launch that
  // produces it -- no launch here with the reference's level layout)
  if (!top_fused) {  // the octaves' top levels from HBM, one launch for all of them
    double px = 0;
    for (int o = 0; o < g.noct; o++) px += g.o[o].plane;
    ProfScope ps(c, HESS_K_HESSIAN, (double)batch * px * 8.0);
    if (s.level_max >= 1 && s.level_max <= g.dog) {  // (never with the reference's level layout: level_max = dog + 1)
      for (int o = 0; o < g.noct; o++) launch_hessian(st, g, o, gauss, deth, got, s.norm, batch, s.level_max, s.level_max);
    } else {
      // this launch also clears the buffers of the detection stages (no fill launch of its own in the chain)
      launch_hessian_level(st, g, gauss, deth, s.level_max, s.norm[s.level_max], batch, user_mode ? nullptr : c->zeroed.p,
                           c->zeroed_used);
      c->zero_filled = !user_mode;
    }
  }
  }  // !(user_mode && on_current)
  if (user_mode) return enqueue_user(c);
  // ---- extrema + ordered list (DetectKeypointsEX part 2 + GenerateFeatureList) ----
  LimitParams lp;
  lp.method = p.truncate_method;
  lp.threshold = p.feature_count_threshold;
  if (!c->zero_filled)  // overflow flags, row counts, histogram, masks (normally cleared by octave 0's top-level launch)
    HIP_TRY(c, hipMemsetAsync(c->zeroed.p, 0, c->zeroed_used, st));
  DetectStore dstore;
  dstore.found = (RawKey*)c->found.p;
  dstore.ntask = extrema_tasks(gx);
  dstore.stride = (long long)c->found_tasks * kDetectSlots + c->cap_raw;
  dstore.task_count = (int*)c->task_count.p;
  dstore.spill_count = (int*)c->found_count.p;
  dstore.cap_spill = c->cap_raw;
  dstore.hist = c->use_topk ? (unsigned*)c->hist.p : nullptr;
  if (dstore.ntask > c->found_tasks) { set_err(c, "detection store laid out for %d scan tasks, the batch has %d", c->found_tasks, dstore.ntask); return HESS_ERR_ARG; }
  {
    // algorithmic bytes: every det-H level of every octave is read once (4 B R per level-pixel)
    double det_bytes = 0;
    for (int o = 0; o < g.noct; o++) det_bytes += 4.0 * s.level_num * g.o[o].plane;
    ProfScope ps(c, HESS_K_EXTREMA, det_bytes * batch);
    launch_extrema_mark(st, gx, dp, gauss, deth, (uint64_t*)c->rowmask.p, (int*)c->rowcnt.p, dstore, batch);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[2], st));
  {
    ProfScope ps(c, HESS_K_EXTREMA, 0.0);
    launch_extrema_place(st, g, lp, dstore, (const uint64_t*)c->rowmask.p, (const int*)c->rowcnt.p, (int*)c->rowoff.p,
                         (int*)c->level_count.p, (int*)c->raw_total.p, (int*)c->overflow.p, (int*)c->place_ticket.p,
                         (int*)c->place_flag.p, (RawKey*)c->raw.p, c->cap_raw, batch);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[3], st));
  // ---- top-K (LimitFeatureCount(0) -> SelectTopK) ----
  const RawKey* list = (const RawKey*)c->raw.p;
  const int* list_total = (const int*)c->raw_total.p;
  int cap_list = c->cap_raw;
  if (c->use_topk) {
    ProfScope ps(c, HESS_K_TOPK, 0.0);
    launch_topk(st, g, p.feature_count_threshold, (const RawKey*)c->raw.p, (const int*)c->raw_total.p, c->cap_raw,
                (unsigned*)c->hist.p, (RawKey*)c->sel.p, (int*)c->sel_total.p, c->cap_sel, batch, c->tk.p, (int*)c->overflow.p);
    list = (const RawKey*)c->sel.p;
    list_total = (const int*)c->sel_total.p;
    cap_list = c->cap_sel;
  }
  c->d_list = list;
  c->d_list_total = list_total;
  c->cap_list = cap_list;
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[4], st));
  // ---- orientation (GetFeatureOrientations) ----
  OrientParams op;
  op.gaussian_factor = p.orient_gaussian_factor;
  op.sample_factor = p.orient_gaussian_factor * p.orient_window_factor;  // ProgramCU.cu:1638
  op.ln_sigma_step = s.ln_sigma_step;
  op.num_orientation = p.fixed_orientation ? 0 : p.max_orientation;      // ProgramCU.cu:1639
  op.subpixel = p.subpixel;
  op.half_sift = p.half_sift;
  op.existing = 0;
  for (int l = 0; l < kMaxLev; l++) op.level_sigma[l] = l <= s.level_max ? s.level_sigma[l] : 0.0f;
  {
    ProfScope ps(c, HESS_K_ORIENT, 0.0);
    launch_orientation(st, g, op, list, list_total, cap_list, got, (FRec*)c->recs.p, (int*)c->ocount.p, batch);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[5], st));
  // ---- multi-orientation expansion (ReshapeFeatureListCPU) ----
  launch_feature_scan(st, g, lp, c->multi ? 1 : 0, list, list_total, cap_list, (const int*)c->ocount.p,
                      (int*)c->foffset.p, (int*)c->fsrc.p, (int*)c->feat_total.p, (int*)c->feat_first.p, c->cap_feat,
                      (int*)c->overflow.p, (int*)c->img_base.p, (int*)c->h_small.p, batch);
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[6], st));
  // ---- descriptors (GetFeatureDescriptors) ----
  DescParams dsp;
  dsp.window_factor = p.desc_window_factor;
  dsp.half_sift = p.half_sift;
  dsp.normalize = p.normalize;
  dsp.multi = c->multi ? 1 : 0;
  dsp.lowe_origin = p.lowe_origin;
  dsp.octave_sigma = first_octave_sigma(c);  // PyramidCU.cpp:746-748
  dsp.dog = g.dog;
  dsp.dynamic_indexing = p.dynamic_indexing ? 1 : 0;
  dsp.hkeys = c->host_direct ? (HostKeypoint*)c->h_keys.p : nullptr;
  dsp.hdesc = (c->host_direct && c->dim) ? (float*)c->h_desc.p : nullptr;
  dsp.first_image = 0;
  dsp.part = 0; dsp.part_den = 1;
  dsp.xcd_block = c->desc_xcd_block;
  dsp.px_band = c->desc_px_band;
  dsp.sequential = p.descriptor_order == HESS_DESC_ORDER_SEQUENTIAL;
  // the pixel order's fixed point assumes luminance in [0, 1] (8- and 16-bit inputs); float pixels are taken as they are
  // and keep the interleaved order (the test oracle applies the same rule)
  dsp.pixel = p.descriptor_order == HESS_DESC_ORDER_PIXEL && pixtype != HESS_PIX_F32;
  // Delivered by the copier thread, a batch of four or more images gets its descriptors in two launches (the images
  // are independent and packed back to back): the first half's results cross the host link while the second half is
  // computed -- half of the transfer (0.53 ms for eight 1080p images) leaves the batch's critical path.  Four groups
  // shorten a lone batch a little more (1.75 / 1.58 / 1.53 ms for 1 / 2 / 4) but cost the pipelined rate 1 %:
  // HESS_DESC_PARTS=n overrides (1: one launch, up to kMaxParts).
  // ONE image delivered by the copier thread (a large one: choose_delivery) gets its descriptors in four launches over
  // quarters of its feature list, for the same reason (a 4096^2 image with 102 k half descriptors: 28 MB, 0.58 ms on the
  // link; 2.09 -> 1.7 ms per image on one context).
  {
    int want = batch >= policy::kSplitDescriptorsFrom ? 2 : 1;
    c->part_features = false;
    static_assert(policy::kLargeImageParts <= Copier::kMaxParts, "parts of one image");
    if (batch == 1 && c->delivery == kDeliverDma) { want = policy::kLargeImageParts; c->part_features = true; }
    if (c->desc_parts > 0) want = std::max(1, std::min<int>(Copier::kMaxParts, c->part_features ? c->desc_parts : std::min(batch, c->desc_parts)));
    if (c->delivery != kDeliverDma || !c->cp.ev_part[0]) want = 1;
    if (want == 1) c->part_features = false;
    c->nparts = want;
    for (int k = 0; k < want; k++) c->part_end[k] = c->part_features ? 1 : (int)((long long)batch * (k + 1) / want);
  }
  {
    int first = 0;
    for (int k = 0; k < c->nparts; k++) {
      ProfScope ps(c, HESS_K_DESCRIPTOR, 0.0);  // (per launch, so that the counts agree with a kernel trace)
      dsp.first_image = first;
      dsp.part = c->part_features ? k : 0;
      dsp.part_den = c->part_features ? c->nparts : 1;
      launch_descriptor(st, g, dsp, list, cap_list, (const FRec*)c->recs.p, (const int*)c->fsrc.p,
                        (const int*)c->feat_total.p, (const int*)c->feat_first.p, (const int*)c->img_base.p, got,
                        (HostKeypoint*)c->keys.p, c->dim ? (float*)c->desc.p : nullptr, c->cap_feat, c->part_end[k] - first,
                        c->seen_features);
      if (k < c->nparts - 1) HIP_TRY(c, hipEventRecord(c->cp.ev_part[k], st));
      if (!c->part_features) first = c->part_end[k];
    }
  }
  HIP_TRY(c, hipEventRecord(c->ev[7], st));
  return 0;
}

// FLOAT_TO_FIXED_POINT (config.h:73-74), host version.
static inline int float_to_fixed_host(float v, int n) {
  return (int)((double)(v * (float)(1 << n)) + ((v >= 0.0) ? 0.5 : -0.5));
}

// User-supplied keypoints (PyramidCU::GenerateFeatureListTex, PyramidCU.cpp:555-718): bin the keys to
// levels by scale, pack fixed-point records on the host, upload, strongest orientation on the device
// unless supplied, descriptors.  One image.
int enqueue_user(hess_ctx* c) {
  const hess_params& p = c->p;
  const Schedule& s = c->sch;
  const Geom& g = c->g;
  hipStream_t st = c->st;
  const int num = (int)c->user_keys.size();
  const double twopi = 2.0 * 3.14159265358979323846;
  const float sigma_half_step = powf(2.0f, 0.5f / g.dog);
  float octave_sigma = first_octave_sigma(c);
  const float offset = p.lowe_origin ? 0.0f : 0.5f;
  std::vector<RawKey> hl;
  std::vector<FRec> hr;
  c->user_kindex.clear();
  const size_t cap = 2 * (size_t)num + 8;
  for (int octave = 0; octave < g.noct; octave++, octave_sigma *= 2.0f) {
    for (int level = 1; level <= g.dog; level++) {
      const float level_sigma = s.level_sigma[level] * octave_sigma;
      const float sigma_min = level_sigma / sigma_half_step;
      const float sigma_max = level_sigma * sigma_half_step;
      for (int k = 0; k < num && hl.size() < cap; k++) {
        const hess_keypoint& key = c->user_keys[k];
        float sigmak = key.s;
        if ((int)c->user_levels.size() == num && c->user_levels[k] >= 0) {  // parity hook: level given, not derived
          if (c->user_levels[k] != octave * g.dog + (level - 1)) continue;
          sigmak = level_sigma;
        }
        if (((sigmak >= sigma_min) && (sigmak < sigma_max)) || ((sigmak < sigma_min) && (octave == 0) && (level == 1)) ||
            ((sigmak > sigma_max) && (octave == g.noct - 1) && (level == g.dog))) {
          const float fX = (key.x - offset) / octave_sigma + 0.5f;
          const float fY = (key.y - offset) / octave_sigma + 0.5f;
          const float fScale = key.s / octave_sigma;
          const float fOrientation = (float)fmod(twopi - key.o, twopi);
          FRec r;
          r.x = (uint32_t)float_to_fixed_host(fX, 10) & 0x00FFFFFFu;
          r.y = (uint32_t)float_to_fixed_host(fY, 10) & 0x00FFFFFFu;
          r.z = (uint32_t)float_to_fixed_host(fScale, 8) & 0x0000FFFFu;
          memcpy(&r.w, &fOrientation, 4);
          RawKey rk;
          memset(&rk, 0, sizeof(rk));
          rk.level_index = octave * g.dog + (level - 1);
          hl.push_back(rk);
          hr.push_back(r);
          c->user_kindex.push_back(k);
        }
      }
    }
  }
  const int n = (int)hl.size();
  if (n > c->cap_raw || n > c->cap_sel || n > c->cap_feat) {
    set_err(c, "keypoint list (%d) exceeds the reserved feature storage", n);
    return HESS_ERR_NOMEM;  // plan() sizes storage for 2*num+8 when a list is set
  }
  int* hs = (int*)c->h_small.p;
  hs[3 * g.B + 4] = n;
  HIP_TRY(c, hipMemsetAsync(c->overflow.p, 0, 64, st));  // overflow words + feature_scan_kernel's arrival counter
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(c->raw.p, hl.data(), (size_t)n * sizeof(RawKey), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->recs.p, hr.data(), (size_t)n * sizeof(FRec), hipMemcpyHostToDevice, st));
  }
  HIP_TRY(c, hipMemcpyAsync(c->raw_total.p, hs + 3 * g.B + 4, 4, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipStreamSynchronize(st));  // hl/hr are pageable host vectors: finish before they go away
  const RawKey* list = (const RawKey*)c->raw.p;
  const int* list_total = (const int*)c->raw_total.p;
  c->d_list = list;
  c->d_list_total = list_total;
  c->cap_list = c->cap_raw;
  if (c->stage_events) for (int e = 1; e <= 4; e++) HIP_TRY(c, hipEventRecord(c->ev[e], st));
  float* got = (float*)c->got.p;
  if (!c->user_have_orientation) {
    OrientParams op;
    op.gaussian_factor = p.orient_gaussian_factor;
    op.sample_factor = p.orient_gaussian_factor * p.orient_window_factor;
    op.ln_sigma_step = s.ln_sigma_step;
    op.num_orientation = p.fixed_orientation ? 0 : p.max_orientation;
    op.subpixel = 0;
    op.half_sift = p.half_sift;
    op.existing = 1;
    for (int l = 0; l < kMaxLev; l++) op.level_sigma[l] = l <= s.level_max ? s.level_sigma[l] : 0.0f;
    launch_orientation(st, g, op, list, list_total, c->cap_raw, got, (FRec*)c->recs.p, (int*)c->ocount.p, 1);
  }
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[5], st));
  LimitParams lp;
  lp.method = 0;
  lp.threshold = -1;  // LimitFeatureCount returns at once for existing keypoints (SiftPyramid.cpp:203)
  launch_feature_scan(st, g, lp, 0, list, list_total, c->cap_raw, (const int*)c->ocount.p, (int*)c->foffset.p,
                      (int*)c->fsrc.p, (int*)c->feat_total.p, (int*)c->feat_first.p, c->cap_feat,
                      (int*)c->overflow.p, (int*)c->img_base.p, (int*)c->h_small.p, 1);
  if (c->stage_events) HIP_TRY(c, hipEventRecord(c->ev[6], st));
  DescParams dsp;
  dsp.window_factor = p.desc_window_factor;
  dsp.half_sift = p.half_sift;
  dsp.normalize = p.normalize;
  dsp.multi = 0;
  dsp.lowe_origin = p.lowe_origin;
  dsp.octave_sigma = first_octave_sigma(c);
  dsp.dog = g.dog;
  dsp.dynamic_indexing = p.dynamic_indexing ? 1 : 0;
  dsp.hkeys = c->host_direct ? (HostKeypoint*)c->h_keys.p : nullptr;
  dsp.hdesc = (c->host_direct && c->dim) ? (float*)c->h_desc.p : nullptr;
  dsp.first_image = 0;
  dsp.part = 0; dsp.part_den = 1;
  dsp.xcd_block = c->desc_xcd_block;
  dsp.px_band = c->desc_px_band;
  dsp.sequential = p.descriptor_order == HESS_DESC_ORDER_SEQUENTIAL;
  dsp.pixel = 0;  // a keypoint list is described in a floating-point order (interleaved unless the sequential one is asked for)
  c->nparts = 1;
  c->part_features = false;
  launch_descriptor(st, g, dsp, list, c->cap_raw, (const FRec*)c->recs.p, (const int*)c->fsrc.p,
                    (const int*)c->feat_total.p, (const int*)c->feat_first.p, (const int*)c->img_base.p, got,
                    (HostKeypoint*)c->keys.p, c->dim ? (float*)c->desc.p : nullptr, c->cap_feat, 1);
  HIP_TRY(c, hipEventRecord(c->ev[7], st));
  return 0;
}


}  // namespace hess
