// hess_copier.hip -- pixels in, results out: stager threads, the copier thread, submit / wait (see hess_ctx.h).
#include "hess_ctx.h"

namespace hess {

// ---- staging helpers (hess_submit_host, pageable input) ----
void stager_copy(Stager& sg, int k) {
  const size_t off = (size_t)k * sg.chunk, len = std::min(sg.chunk, sg.bytes - off);
  memcpy(sg.dst + off, sg.src + off, len);
  { std::lock_guard<std::mutex> lk(sg.mu); sg.state[k].store(2, std::memory_order_release); }
  sg.cv_done.notify_all();
}

void stager_main(Stager* sgp) {
  Stager& sg = *sgp;
  unsigned long long seen = 0;
  std::unique_lock<std::mutex> lk(sg.mu);
  for (;;) {
    sg.cv_job.wait(lk, [&] { return sg.stop || sg.gen != seen; });
    if (sg.stop) return;
    seen = sg.gen;
    lk.unlock();
    for (;;) {
      const int k = sg.next_hi.fetch_sub(1, std::memory_order_acq_rel);
      if (k < 0) break;
      int expect = 0;
      if (sg.state[k].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) stager_copy(sg, k);
      else break;  // met the calling thread coming up: everything is claimed
    }
    lk.lock();
    sg.active--;
    sg.cv_done.notify_all();
  }
}

void stager_start(Stager& sg) {
  if (sg.tried) return;
  sg.tried = true;
  for (int t = 0; t < Stager::kHelpers; t++) {
    try { sg.th[sg.nth] = std::thread(stager_main, &sg); sg.nth++; } catch (...) { break; }  // fewer helpers: the caller copies more
  }
}

void stager_stop(Stager& sg) {
  if (!sg.nth) return;
  { std::lock_guard<std::mutex> lk(sg.mu); sg.stop = true; }
  sg.cv_job.notify_all();
  for (int t = 0; t < sg.nth; t++) sg.th[t].join();
  sg.nth = 0;
}

// ---- copier thread (kDeliverDma) ----
// One job at a time: wait on the host for the event behind the batch's last kernel, read the packed total from the
// pinned count block (stored there by feature_scan_kernel), copy exactly that many records on the copy-only stream.
std::atomic<int> g_copier_count{0};  // contexts with a copier, for spreading them over the preferred engines

// Bind the copier to ROCr: the agents that own the result buffers (from the pointers themselves), an SDMA engine of
// the set ROCr recommends for device->host on this pair of agents (contexts take turns), a completion signal.
bool copier_hsa_setup(hess_ctx* c) {
  Copier& cp = c->cp;
  if (cp.hsa_ready) return true;
  if (cp.hsa_failed) return false;
  cp.hsa_failed = true;
  if (const char* m = dev_env("HESS_COPIER")) if (!strcmp(m, "hip")) return false;
  if (hsa_init() != HSA_STATUS_SUCCESS) return false;  // reference-counted: HIP has initialised ROCr already
  hsa_amd_pointer_info_t pi;
  memset(&pi, 0, sizeof(pi));
  pi.size = sizeof(pi);
  if (hsa_amd_pointer_info(c->keys.p, &pi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pi.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
  cp.gpu_agent = pi.agentOwner;
  memset(&pi, 0, sizeof(pi));
  pi.size = sizeof(pi);
  // (the host side from the count block: always the runtime's own pinned allocation, also when the result buffers
  // are registered shared memory, whose owner ROCr reports differently)
  if (hsa_amd_pointer_info(c->h_small.p, &pi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pi.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
  cp.cpu_agent = pi.agentOwner;
  if (hsa_signal_create(0, 0, nullptr, &cp.sig) != HSA_STATUS_SUCCESS) return false;
  if (hsa_signal_create(0, 0, nullptr, &cp.sig2) != HSA_STATUS_SUCCESS) { (void)hsa_signal_destroy(cp.sig); return false; }
  uint32_t pref = 0;
  if (hsa_amd_memory_get_preferred_copy_engine(cp.cpu_agent, cp.gpu_agent, &pref) == HSA_STATUS_SUCCESS && pref) {
    const int n = __builtin_popcount(pref), k = g_copier_count.fetch_add(1) % n;
    uint32_t m = pref;
    for (int i = 0; i < k; i++) m &= m - 1;
    cp.engine = m & (~m + 1);
  }
  // ... and an engine of the host->device set for the pixel uploads (left to ROCr, an upload sometimes lands on an engine
  // that copies at a quarter of the rate: the host-to-host figure varied 15 - 21 Gpix/s from run to run)
  uint32_t pref_in = 0;
  if (hsa_amd_memory_get_preferred_copy_engine(cp.gpu_agent, cp.cpu_agent, &pref_in) == HSA_STATUS_SUCCESS && pref_in) {
    const int n = __builtin_popcount(pref_in), k = g_copier_count.load() % n;
    uint32_t m = pref_in;
    for (int i = 0; i < k; i++) m &= m - 1;
    cp.engine_in = m & (~m + 1);
  }
  if (const char* e = dev_env("HESS_COPIER_ENGINE")) cp.engine = (uint32_t)strtoul(e, nullptr, 0);
  if (const char* e = dev_env("HESS_UPLOAD_ENGINE")) cp.engine_in = (uint32_t)strtoul(e, nullptr, 0);
  cp.hsa_failed = false;
  cp.hsa_ready = true;
  return true;
}

// Host wait for a copy's completion signal to drop below `below`, in slices of a second up to HESS_COPY_TIMEOUT_S
// (default 10): a lost completion must not hang hess_wait / hess_destroy (the reference returns 0 on device errors,
// SiftPyramid.h:162-163, it never hangs).  Returns 0 when the copies completed, 1 when the limit expired, 2 when ROCr
// reported a failed copy (it then leaves a NEGATIVE value in the signal); *last = the value seen.
// HESS_COPIER_FAULT=timeout|error makes the next wait of the process report that outcome (fault injection for the
// tests; the real signal is still waited for, so nothing is left in flight).  *real (if given) says whether the outcome
// is the signal's own: then the copy may still be in flight and the context is poisoned (HESS_COPIER_FAULT=poisoned
// reports a timeout as if it were real, after the copy has in fact completed).
std::atomic<int> g_copier_fault{-1};  // -1: not read yet, 0: none, 1: timeout, 2: error, 3: poisoned (consumed by the first wait)
int wait_copy_signal(hsa_signal_t sig, hsa_signal_value_t below, hsa_signal_value_t* last, bool injectable, bool* real) {
  if (real) *real = false;
  static const double limit_s = [] { const char* e = getenv("HESS_COPY_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 10.0; }();
  static const uint64_t ticks_per_s = [] {
    uint64_t f = 0;
    return (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &f) == HSA_STATUS_SUCCESS && f) ? f : (uint64_t)100000000;
  }();
  int inject = injectable ? g_copier_fault.load() : 0;
  if (injectable && inject < 0) {
    const char* e = dev_env("HESS_COPIER_FAULT");
    int want = !e ? 0 : (!strcmp(e, "timeout") ? 1 : (!strcmp(e, "error") ? 2 : (!strcmp(e, "poisoned") ? 3 : 0)));
    int expect = -1;
    if (!g_copier_fault.compare_exchange_strong(expect, want)) want = expect;
    inject = want;
  }
  const auto t0 = std::chrono::steady_clock::now();
  hsa_signal_value_t v;
  for (;;) {
    v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, below, ticks_per_s, HSA_WAIT_STATE_BLOCKED);
    if (v < below) break;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
      if (last) *last = v;
      if (real) *real = true;
      return 1;
    }
  }
  if (last) *last = v;
  if (inject > 0) {
    int expect = inject;
    if (g_copier_fault.compare_exchange_strong(expect, 0)) {
      if (inject == 3 && real) *real = true;
      return inject == 3 ? 1 : inject;
    }
  }
  if (v < 0 && real) *real = true;
  return v < 0 ? 2 : 0;
}

// keys + descriptors of features [first, first + total) to the pinned host buffers by SDMA.  0: done; 1: ROCr refused
// to take the copy, use the fallback; 2: a copy was taken and did not complete (timeout or error, `why`): the batch fails.
int copier_hsa_copy(hess_ctx* c, size_t first, size_t total, char* why, size_t why_len) {
  Copier& cp = c->cp;
  hsa_signal_value_t seen = 0;
  bool real = false;
  auto lost = [&](int w) {
    snprintf(why, why_len, "device->host copy of the results %s (signal value %lld, engine 0x%x)",
             w == 1 ? "did not complete in time" : "failed", (long long)seen, cp.engine);
    cp.hsa_ready = false; cp.hsa_failed = true;  // later batches take the stream copy; the signals are not reused
    if (real) c->poisoned.store(true);           // the copy may still land: see hess_ctx::poisoned
    return 2;
  };
  auto one = [&](hsa_signal_t sig, void* dst, const void* src, size_t bytes) {
    hsa_signal_store_relaxed(sig, 1);
    hsa_status_t st = cp.engine
        ? hsa_amd_memory_async_copy_on_engine(dst, cp.cpu_agent, src, cp.gpu_agent, bytes, 0, nullptr, sig,
                                              (hsa_amd_sdma_engine_id_t)cp.engine, false)
        : hsa_amd_memory_async_copy(dst, cp.cpu_agent, src, cp.gpu_agent, bytes, 0, nullptr, sig);
    if (st != HSA_STATUS_SUCCESS && cp.engine)  // engine busy or not available: let ROCr choose
      st = hsa_amd_memory_async_copy(dst, cp.cpu_agent, src, cp.gpu_agent, bytes, 0, nullptr, sig);
    return st == HSA_STATUS_SUCCESS;
  };
  if (!one(cp.sig, (char*)c->h_keys.p + first * sizeof(HostKeypoint), (const char*)c->keys.p + first * sizeof(HostKeypoint),
           total * sizeof(HostKeypoint)))
    return 1;
  const bool second = c->dim && one(cp.sig2, (char*)c->h_desc.p + first * c->dim * 4, (const char*)c->desc.p + first * c->dim * 4, total * c->dim * 4);
  // (both copies are in flight on the same engine; each has its own signal)
  if (const int w = wait_copy_signal(cp.sig, 1, &seen, true, &real)) {
    if (second) { bool r2 = false; (void)wait_copy_signal(cp.sig2, 1, nullptr, false, &r2); real = real || r2; }
    return lost(w);
  }
  if (c->dim && !second) return 1;  // ROCr took the first copy (done by now) and refused the second: the fallback copies both
  if (second)
    if (const int w = wait_copy_signal(cp.sig2, 1, &seen, false, &real)) return lost(w);
  return 0;
}

void copier_main(hess_ctx* c) {
  Copier& cp = c->cp;
  (void)hipSetDevice(c->device);
  std::unique_lock<std::mutex> lk(cp.mu);
  for (;;) {
    cp.cv.wait(lk, [&] { return cp.stop || cp.has_job; });
    if (cp.stop) return;
    const int batch = cp.batch;
    lk.unlock();
    int rc = 0;
    bool overflow = false;
    char msg[256] = "";
    auto fail = [&](const char* what, hipError_t e) {
      snprintf(msg, sizeof(msg), "%s failed: %s (copier)", what, hipGetErrorString(e));
      rc = e == hipErrorOutOfMemory ? HESS_ERR_NOMEM : HESS_ERR_DEVICE;
    };
    hipError_t e = hipSuccess;
    if (cp.upload_first) {  // wait for the pixels on the host, then enqueue the batch
      const auto t0 = std::chrono::steady_clock::now();
      hsa_signal_value_t seen = 0;
      bool real = false;
      if (const int w = wait_copy_signal(cp.sig_in, 1, &seen, true, &real)) {
        if (real) c->poisoned.store(true);  // the upload may still write the staging area: see hess_ctx::poisoned
        // the pixels never arrived (or arrived wrong): the kernels are NOT run on them
        snprintf(msg, sizeof(msg), "host->device upload of the pixels %s (signal value %lld) (copier)",
                 w == 1 ? "did not complete in time" : "failed", (long long)seen);
        rc = HESS_ERR_DEVICE;
        cp.have_sig_in = false;  // (a signal that may still be written is left alone, not reused)
      }
      PendingRun& r = *cp.run;
      r.t_load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      cp.nparts = 1;
      cp.part_features = false;
      if (!rc) {
        try {
          rc = enqueue(c, r.dev, r.pitch, r.image_stride, r.batch, r.format, r.pixtype);
        } catch (...) { rc = HESS_ERR_NOMEM; snprintf(msg, sizeof(msg), "out of host memory (copier)"); }
        if (!rc && (e = hipEventRecord(cp.ev_done, c->st)) != hipSuccess) fail("hipEventRecord", e);
        if (rc && !msg[0]) snprintf(msg, sizeof(msg), "%s", c->err.c_str());
        if (!rc) {  // (a half-run enqueue leaves no parts to wait for)
          cp.nparts = c->nparts;
          cp.part_features = c->part_features;
          for (int k = 0; k < Copier::kMaxParts; k++) cp.part_end[k] = c->part_end[k];
        }
      }
    }
    // Several parts when the batch's descriptors were launched in groups of images (nparts > 1): a group's results
    // cross while the next group is computed; else one part behind the last kernel.  The counts (and the overflow
    // words) are in the pinned count block since feature_scan_kernel, i.e. before any of the events.
    const int nparts = cp.nparts > 1 ? cp.nparts : 1;
    if (!rc && (e = hipEventSynchronize(nparts > 1 ? cp.ev_part[0] : cp.ev_done)) != hipSuccess) fail("hipEventSynchronize", e);
    if (!rc) {
      const int* hs = (const int*)c->h_small.p;
      overflow = hs[batch + 1] != 0 || hs[batch + 2] != 0 || hs[batch + 3] != 0;  // (word 3: a device-side error, nothing to copy)
      const size_t total = overflow ? 0 : (size_t)hs[batch];
      DevBuf *hk = &c->h_keys, *hd = &c->h_desc;
      if (total) {
        // (the pinned buffers hold the worst case unless that exceeds 512 MB: then they grow here, rarely)
        if (hk->bytes < total * sizeof(HostKeypoint) || (c->dim && hd->bytes < total * c->dim * 4)) {
          if (ensure(c, *hk, total * sizeof(HostKeypoint), true) || (c->dim && ensure(c, *hd, total * c->dim * 4, true))) {
            snprintf(msg, sizeof(msg), "pinned result buffers: %s (copier)", c->err.empty() ? "allocation failed" : c->err.c_str());
            rc = HESS_ERR_NOMEM;
          }
        }
      }
      auto copy_part = [&](size_t first, size_t n) {
        if (rc || !n) return;
        if (copier_hsa_setup(c)) {
          const int hr = copier_hsa_copy(c, first, n, msg, sizeof(msg));
          if (hr == 0) return;  // both blocks are in host memory
          if (hr == 2) { rc = HESS_ERR_DEVICE; return; }
        }
        const size_t kb = sizeof(HostKeypoint), db = (size_t)c->dim * 4;
        if ((e = hipMemcpyAsync((char*)hk->p + first * kb, (const char*)c->keys.p + first * kb, n * kb, hipMemcpyDeviceToHost, cp.cs)) != hipSuccess)
          fail("hipMemcpyAsync(keys)", e);
        if (!rc && c->dim &&
            (e = hipMemcpyAsync((char*)hd->p + first * db, (const char*)c->desc.p + first * db, n * db, hipMemcpyDeviceToHost, cp.cs)) != hipSuccess)
          fail("hipMemcpyAsync(desc)", e);
        if (!rc && (e = hipStreamSynchronize(cp.cs)) != hipSuccess) fail("hipStreamSynchronize(copy stream)", e);
      };
      size_t done_feats = 0;
      for (int k = 0; k < nparts; k++) {
        if (k > 0 && (e = hipEventSynchronize(k < nparts - 1 ? cp.ev_part[k] : cp.ev_done)) != hipSuccess) fail("hipEventSynchronize", e);
        // features of the images so far -- or, of one image's list, the bound the part's launch formed (k_feature.hip, feature_part)
        const size_t upto = overflow ? 0 : (k == nparts - 1 ? total : cp.part_features ? (size_t)((long long)total * (k + 1) / nparts)
                                                                                     : (size_t)hs[cp.part_end[k]]);
        copy_part(done_feats, upto - done_feats);
        done_feats = upto;
      }
    }
    lk.lock();
    cp.rc = rc;
    cp.overflow = overflow;
    snprintf(cp.err, sizeof(cp.err), "%s", msg);
    cp.has_job = false;
    cp.done = true;
    cp.cv.notify_all();
  }
}

int copier_start(hess_ctx* c) {
  Copier& cp = c->cp;
  if (cp.started) return 0;
  HIP_TRY(c, hipStreamCreateWithFlags(&cp.cs, hipStreamNonBlocking));
  HIP_TRY(c, hipEventCreateWithFlags(&cp.ev_done, hipEventDisableTiming));
  for (hipEvent_t& ev : cp.ev_part) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  try {
    cp.th = std::thread(copier_main, c);
  } catch (...) {
    set_err(c, "cannot start the copier thread");
    return HESS_ERR_NOMEM;
  }
  cp.started = true;
  return 0;
}

void copier_stop(hess_ctx* c) {
  Copier& cp = c->cp;
  if (cp.started) {
    {
      std::unique_lock<std::mutex> lk(cp.mu);
      cp.cv.wait(lk, [&] { return cp.done; });
      cp.stop = true;
      cp.cv.notify_all();
    }
    cp.th.join();
    cp.started = false;
  }
  // (a poisoned context's signals may still be written by a late copy: left alone, like the buffers)
  if (cp.hsa_ready) { (void)hsa_signal_destroy(cp.sig); (void)hsa_signal_destroy(cp.sig2); cp.hsa_ready = false; }
  if (cp.have_sig_in && !c->poisoned.load()) { (void)hsa_signal_destroy(cp.sig_in); cp.have_sig_in = false; }
  if (cp.cs) { (void)hipStreamDestroy(cp.cs); cp.cs = nullptr; }
  if (cp.ev_done) { (void)hipEventDestroy(cp.ev_done); cp.ev_done = nullptr; }
  for (hipEvent_t& ev : cp.ev_part) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
}

// How the results of a batch of `batch` images reach the host (see the kDeliver* comment): small batches through the
// descriptor kernel's own stores (lowest latency), larger ones by the copier thread's DMA copy (no kernel waits for
// PCIe).  HESS_DELIVERY overrides.
void choose_delivery(hess_ctx* c, int batch) {
  // Small batches keep the in-kernel mirror (latency: no event wake-up, no copy behind the last kernel) -- unless their
  // results are large: a kernel that stores tens of megabytes into host memory waits for the link (one 4096^2 image with
  // 102 k half descriptors, 28 MB: 0.76 ms against 0.47), while the copier's DMA copy of a part runs beside the next
  // part's launch -- 15 % more images per second on three pipelined contexts.  That needs a caller who overlaps: a batch
  // handed over by hess_submit_* (hess_run_*, i.e. submit + wait in one call, keeps the mirror: nothing to overlap with,
  // and through the class with pageable pixels the copier's route is 15 % slower for a 4096^2 image).  "Large" is judged by
  // what the context's last batch of this size delivered (the capacity is a worst case many times the typical count): the
  // first batch of a context uses the mirror.  HESS_MIRROR_MAX_MB (16) is the limit.
  const size_t expect = (c->last_result_batch == batch && !c->caller_waits) ? c->last_result_bytes : 0;
  const bool small = batch <= c->mirror_max_batch && !(batch >= 2 && !c->caller_waits);  // (a submitted pair: see enqueue())
  int d = c->delivery_pref >= 0 ? c->delivery_pref : (small && expect <= c->mirror_max_bytes ? kDeliverMirror : kDeliverDma);
  if (d == kDeliverMirror && !c->host_fits) d = kDeliverDma;  // the mirror needs the worst case pinned up front
  if (d == kDeliverDma && copier_start(c) != 0) d = kDeliverBlit;
  c->delivery = d;
  c->host_direct = d == kDeliverMirror;
}

// Enqueue the whole path (the per-image counts reach the pinned count block by feature_scan_kernel's own stores);
// returns without waiting.
int submit_inner(hess_ctx* c, const PendingRun& r) {
  if (!c->user_keys.empty() && r.batch != 1) {
    set_err(c, "a keypoint list applies to a single image");
    return HESS_ERR_ARG;
  }
  int rc = plan(c, r.width, r.height, r.batch);
  if (rc) return rc;
  choose_delivery(c, r.batch);
  HIP_TRY(c, hipGetLastError());
  rc = enqueue(c, r.dev, r.pitch, r.image_stride, r.batch, r.format, r.pixtype);
  if (rc) return rc;
  HIP_TRY(c, hipGetLastError());
  if (c->delivery == kDeliverDma) {
    Copier& cp = c->cp;
    HIP_TRY(c, hipEventRecord(cp.ev_done, c->st));
    std::lock_guard<std::mutex> lk(cp.mu);
    cp.batch = r.batch;
    cp.upload_first = false;
    cp.nparts = c->nparts;
    cp.part_features = c->part_features;
    for (int k = 0; k < Copier::kMaxParts; k++) cp.part_end[k] = c->part_end[k];
    cp.done = false;
    cp.has_job = true;
    cp.cv.notify_all();
  }
  return 0;
}

// (nothing thrown crosses the C ABI: enqueue_user builds host vectors)
int submit_impl(hess_ctx* c, const PendingRun& r) {
  try {
    return submit_inner(c, r);
  } catch (...) {
    set_err(c, "out of host memory while preparing the batch");
    return HESS_ERR_NOMEM;
  }
}

// Wait for the submitted batch, grow storage and re-run if a list overflowed; with kDeliverBlit bring the
// keypoints and descriptors of the whole batch to the host with one transfer each (the other modes have
// delivered them by now).
int wait_inner(hess_ctx* c, const PendingRun& r) {
  int rc;
  int* hs = (int*)c->h_small.p;
  const int batch = r.batch;
  for (int attempt = 0;; attempt++) {
    if (c->delivery == kDeliverDma) {
      Copier& cp = c->cp;
      std::unique_lock<std::mutex> lk(cp.mu);
      cp.cv.wait(lk, [&] { return cp.done; });
      if (cp.rc) { set_err(c, "%s", cp.err); return cp.rc; }
    } else {
      HIP_TRY(c, hipStreamSynchronize(c->st));
    }
    const int of_raw = hs[batch + 1], of_feat = hs[batch + 2];
    if (hs[batch + 3]) {  // raised by a kernel that gave up a bounded wait (topk_select_kernel's look-back): no results
      set_err(c, "device-side wait did not complete (top-K look-back); the batch has no results");
      return HESS_ERR_DEVICE;
    }
    if (!of_raw && !of_feat) break;
    if (attempt >= 8) { set_err(c, "feature storage keeps overflowing"); return HESS_ERR_NOMEM; }
    // grow-only reallocation, then run the batch again (reference: SetLevelFeatureNum grows on demand,
    // PyramidCU.cpp:393-397)
    if (of_raw) c->cap_raw = of_raw + of_raw / 4;
    if (of_feat) c->cap_feat = of_feat + of_feat / 4;
    c->planned = false;
    c->regrown++;
    if (c->p.verbose & 1) fprintf(stderr, "hessgpu: feature storage grown (raw %d, features %d)\n", c->cap_raw, c->cap_feat);
    if ((rc = submit_impl(c, r))) return rc;
  }
  drain_profile(c);
  c->counts.resize(batch);
  c->offs.assign(batch + 1, 0);
  for (int b = 0; b < batch; b++) {
    c->counts[b] = hs[b + 1] - hs[b];
    c->offs[b + 1] = (size_t)hs[b + 1];
  }
  const size_t total = c->offs[batch];
  c->seen_features = batch ? *std::max_element(c->counts.begin(), c->counts.end()) : 0;
  c->last_result_bytes = total * (sizeof(HostKeypoint) + (size_t)c->dim * 4);
  c->last_result_batch = batch;
  if ((rc = ensure(c, c->h_keys, (total ? total : 1) * sizeof(HostKeypoint), true))) return rc;
  if (c->dim && (rc = ensure(c, c->h_desc, (total ? total : 1) * c->dim * 4, true))) return rc;
  if (total && c->delivery == kDeliverBlit) {
    HIP_TRY(c, hipMemcpyAsync(c->h_keys.p, c->keys.p, total * sizeof(HostKeypoint), hipMemcpyDeviceToHost, c->st));
    if (c->dim)
      HIP_TRY(c, hipMemcpyAsync(c->h_desc.p, c->desc.p, total * c->dim * 4, hipMemcpyDeviceToHost, c->st));
    HIP_TRY(c, hipStreamSynchronize(c->st));
  }
  c->user_result = false;
  if (!c->user_keys.empty()) {
    // back to input order (PyramidCU.cpp:537-549,1157-1168); the caller's keypoints are returned
    // unchanged unless DownloadKeypoints would run (-m 1 / -ofix, SiftPyramid.cpp:160-171)
    const int num = (int)c->user_keys.size();
    const int listed = (int)std::min<size_t>(total, (size_t)num);
    const bool download = !c->user_have_orientation && ((c->p.max_orientation < 2) || c->p.fixed_orientation);
    c->u_keys = c->user_keys;
    c->u_desc.assign((size_t)num * (c->dim ? c->dim : 1), 0.0f);
    for (int i = 0; i < listed; i++) {
      const int k = c->user_kindex[i];
      if (download) memcpy(&c->u_keys[k], (HostKeypoint*)c->h_keys.p + i, sizeof(hess_keypoint));
      if (c->dim) memcpy(&c->u_desc[(size_t)k * c->dim], (float*)c->h_desc.p + (size_t)i * c->dim, (size_t)c->dim * 4);
    }
    c->counts[0] = num;
    c->offs[1] = (size_t)num;
    c->user_result = true;
    c->user_keys.clear();  // _existing_keypoints = 0 after RunSIFT (SiftPyramid.cpp:182-184)
    c->user_levels.clear();  // the parity hook covers one keypoint-list run
    c->user_on_current = false;
  }
  // stage times from the events of the last enqueue (config.h:17-31 order)
  memset(c->timing, 0, sizeof(c->timing));
  auto el = [&](int i, int j) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[j]); return ms; };
  c->timing[HESS_T_LOAD] = (float)r.t_load_ms;
  if (r.timed_load) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev_load[0], c->ev_load[1]); c->timing[HESS_T_LOAD] = ms; }
  if (c->stage_events) {
    c->timing[HESS_T_PYRAMID] = el(0, 1);
    c->timing[HESS_T_DETECT] = el(1, 2);
    c->timing[HESS_T_LIST] = el(2, 3);
    c->timing[HESS_T_REDUCTION] = el(3, 4);
    c->timing[HESS_T_ORIENT] = el(4, 5);
    c->timing[HESS_T_MULTI_ORIENT] = el(5, 6);
    c->timing[HESS_T_DESCRIPTOR] = el(6, 7);
  }
  c->timing[HESS_T_TOTAL] = el(0, 7) + c->timing[HESS_T_LOAD];
  c->batch = c->pyramid_batch = batch;  // only now: every error return above leaves the context without results
  return 0;
}

int wait_impl(hess_ctx* c, const PendingRun& r) {
  try {
    return wait_inner(c, r);
  } catch (...) {  // the host-side count / keypoint-list vectors
    c->user_keys.clear();
    set_err(c, "out of host memory while collecting the results");
    return HESS_ERR_NOMEM;
  }
}


}  // namespace hess
