// prad_image.h -- the case pipeline's native half (a textual part of prad_api.hip: it uses that unit's internal entry points;
// split off in round 6 so that the dispatch code of the matrix calls and the pipeline's queues / launcher do not share a file):
//   prad_image_enqueue_dev / prad_image_wait          every class of ONE derived image queued by one call on four side streams
//   prad_image_submit / _result / _wait / _release    the same call issued by a launcher thread that belongs to the calling thread
// Included by prad_api.hip only.
#pragma once

// ---- one derived image, every class, one call (the case pipeline's enqueue half in native code) -------------------------
// What pyradiomics_amd/featureextractor.py did with ~25 calls per derived image -- GLCM + GLRLM sweep, their formulas and the MCC
// on side stream 0, GLSZM on side stream 1, first order on side stream 2, GLDM + NGTDM pass and formulas on side stream 3
// (round 6; two alternating sets of these streams, one byte-packed copy of the volume shared in front of the fork), each under its own
// workspace set, a verdict mark and an event behind each -- as one entry point: the per-call cost of the Python / ctypes
// layer (15 - 25 us each) was a fifth of a 256^3 case.  The matrices live in workspace buffers (stream order recycles
// them), the values land in one block of the result arena.
#define PRAD_IMG_TICKETS 4
#define PRAD_IMG_STREAMS 4   // side streams of an image: 0 sweeps (GLCM + GLRLM), 1 GLSZM, 2 first order, 3 neighbourhoods (GLDM + NGTDM)
#define GF_FEATURES 23   // GF_COUNT of kernels_features.h
namespace {
struct ImageQueues {
  hipStream_t sp[2][PRAD_IMG_STREAMS] = {};   // two sets of side streams: consecutive images alternate (round 6)
  hipEvent_t in = nullptr;
  hipEvent_t done[PRAD_IMG_TICKETS][PRAD_IMG_STREAMS] = {};
  int *flag[PRAD_IMG_TICKETS][PRAD_IMG_STREAMS] = {};
  unsigned used[PRAD_IMG_TICKETS] = {};
  unsigned long long seq = 0;
  int device = -1;
};
ImageQueues *image_queue_table() {
  static thread_local ImageQueues q[16];
  return q;
}
ImageQueues &image_queues() { return image_queue_table()[ctx().device & 15]; }
void release_image_queues() {
  ImageQueues *tab = image_queue_table();
  for (int d = 0; d < 16; d++) {
    ImageQueues &q = tab[d];
    if (q.device < 0) continue;
    for (int k = 0; k < PRAD_IMG_STREAMS; k++)
      for (int h = 0; h < 2; h++)
        if (q.sp[h][k]) (void)hipStreamDestroy(q.sp[h][k]);
    if (q.in) (void)hipEventDestroy(q.in);
    for (int t = 0; t < PRAD_IMG_TICKETS; t++)
      for (int k = 0; k < PRAD_IMG_STREAMS; k++)
        if (q.done[t][k]) (void)hipEventDestroy(q.done[t][k]);
    q = ImageQueues();
  }
}
}  // namespace
extern "C" {

int prad_image_enqueue_dev(const int32_t *levels, const uint8_t *mask, const void *raw, int raw_dtype, const int *size,
                           int Nd, int Ng, long long Ns, int classes, int symmetric, int alpha, int force2Ddim,
                           double voxelArrayShift, double **results, int *layout, int *ticket, void *stream) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!levels || !mask || !size || !results || !layout || !ticket || Nd < 1 || Nd > PRAD_MAX_ND || Ng < 1)
    return fail(PRAD_E_ARG, "image_enqueue: bad arguments");
  if ((classes & PRAD_IMG_FIRSTORDER) && !raw) return fail(PRAD_E_ARG, "image_enqueue: first order needs the undiscretised image");
  ImageQueues &q = image_queues();
  if (q.device != c.device) {
    for (int h = 0; h < 2; h++)
      for (int k = 0; k < PRAD_IMG_STREAMS; k++) PRAD_HIP(hipStreamCreateWithFlags(&q.sp[h][k], hipStreamNonBlocking));
    PRAD_HIP(hipEventCreateWithFlags(&q.in, hipEventDisableTiming));
    for (int t = 0; t < PRAD_IMG_TICKETS; t++)
      for (int k = 0; k < PRAD_IMG_STREAMS; k++) PRAD_HIP(hipEventCreateWithFlags(&q.done[t][k], hipEventDisableTiming));
    q.device = c.device;
  }
  if (q.used[q.seq % PRAD_IMG_TICKETS] != 0)
    return fail(PRAD_E_ARG, "image_enqueue: %d images are in flight on this thread; prad_image_wait one first", PRAD_IMG_TICKETS);
  PRAD_TRY(prad_set_deferred(1));     // (creates the sticky word on first use)
  // Consecutive images alternate between two sets of side streams and workspace sets: the tail of image i on a stream -- its
  // chain of small formula kernels, ~0.1 ms in which the GPU is nearly idle -- runs under the first kernels of image i + 1
  // instead of in front of them (one stream per class serialised the images of a case class by class).  PRAD_IMG_ONE_SET=1: as before.
  static const bool one_set = getenv("PRAD_IMG_ONE_SET") != nullptr;
  const int par = one_set ? 0 : (int)(q.seq & 1);
  hipStream_t *qs = q.sp[par];
  const int wso = 4 * par;               // workspace sets 4..7 (even images), 8..11 (odd images)
  struct Restore {
    Context &c;
    ~Restore() {
      c.deferred = false;
      c.workspace = 0;
    }
  } restore{c};
  const int one = 1;
  const int Na = angle_count(size, &one, Nd, 1, 0, force2Ddim), Nab = angle_count(size, &one, Nd, 1, 1, force2Ddim);
  if (Na < 1 || Nab < 1 || Na > PRAD_MAX_SWEEP) return fail(PRAD_E_UNSUPPORTED, "image_enqueue: %d / %d angles", Na, Nab);
  std::vector<int> ang((size_t)Na * Nd), angb((size_t)Nab * Nd);
  if (angle_build(size, &one, Nd, 1, force2Ddim, Na, ang.data()) || angle_build(size, &one, Nd, 1, force2Ddim, Nab, angb.data()))
    return fail(PRAD_E_ARG, "image_enqueue: angles");
  int Nr = 1;
  long long n = 1;
  for (int d = 0; d < Nd; d++) {
    Nr = std::max(Nr, size[d]);
    n *= size[d];
  }
  // result block (doubles): [0] glcm Na x 23, [1] glcm empty (Na ints), [2] mcc Na + 1, [3] glrlm Na x 16, [4] glrlm empty,
  // [5] gldm 16, [6] gldm empty, [7] ngtdm 5, [8] glszm 17, [9] glszm empty, [10] first order 16; layout[k] = offset in
  // doubles or -1 (class not asked for / declined), layout[11] = Na, layout[12] = total
  size_t off = 0;
  auto take = [&](int k, size_t doubles) {
    layout[k] = (int)off;
    off += (doubles + 7) & ~(size_t)7;
  };
  for (int k = 0; k < 16; k++) layout[k] = -1;
  // (values and their flags back to back: the formula calls then copy both with one transfer)
  auto take2 = [&](int kv, size_t values, int kf, size_t flags) {
    layout[kv] = (int)off;
    layout[kf] = (int)(off + values);
    off += (values + (flags + 1) / 2 + 1 + 7) & ~(size_t)7;
  };
  if (classes & PRAD_IMG_GLCM) take2(0, (size_t)Na * GF_FEATURES, 1, (size_t)Na);
  if ((classes & PRAD_IMG_GLCM) && (classes & PRAD_IMG_MCC)) take(2, (size_t)Na + 1);
  if (classes & PRAD_IMG_GLRLM) take2(3, (size_t)Na * ZM_FEATURES, 4, (size_t)Na);
  if (classes & PRAD_IMG_GLDM) take2(5, ZM_FEATURES, 6, 1);
  if (classes & PRAD_IMG_NGTDM) take(7, 5);
  if (classes & PRAD_IMG_GLSZM) { take(8, ZM_FEATURES + 1); take(9, 1); }
  if (classes & PRAD_IMG_FIRSTORDER) take(10, 16);
  layout[11] = Na;
  layout[12] = (int)off;
  void *blk = nullptr;
  PRAD_TRY(c.arena_alloc(sizeof(double) * std::max<size_t>(off, 8), &blk));
  double *res = (double *)blk;
  *results = res;
  auto at = [&](int k) { return res + layout[k]; };
  // ONE byte-packed copy of the volume for the neighbourhood pass and the GLSZM (each of them packed for itself until round 6),
  // made on the caller's stream in front of the fork; a buffer per ticket: up to four images are in flight
  struct ClearShared {
    ~ClearShared() { shared_pack() = SharedPack(); }
  } clear_shared;
  if ((classes & (PRAD_IMG_GLDM | PRAD_IMG_NGTDM | PRAD_IMG_GLSZM)) && Nd <= 3 && Ng <= 255 && n < 0x7fffffffLL &&
      !getenv("PRAD_IMG_NO_SHARED_PACK")) {
    char name[32];
    snprintf(name, sizeof(name), "img_pack%d", (int)(q.seq % PRAD_IMG_TICKETS));
    uint8_t *pk = nullptr;
    int *pf = nullptr;
    c.workspace = 0;
    PRAD_TRY(c.get<uint8_t>(name, (size_t)n + 64 + 16, &pk));
    pf = (int *)(pk + (((size_t)n + 64 + 3) & ~(size_t)3));
    PRAD_HIP(hipMemsetAsync(pf, 0, sizeof(int) * 2, (hipStream_t)stream));
    const int vec_ok = ((((uintptr_t)levels) | ((uintptr_t)mask) | ((uintptr_t)pk)) & 15) == 0;
    const int NX = size[Nd - 1];
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((n / 16 + 255) / 256, 4096));
    hipLaunchKernelGGL(pack_levels_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, levels, mask, n, NX, NX, 0, Ng, pk, pf, vec_ok, 0);
    PRAD_TRY(check_launch("pack_levels_kernel"));
    SharedPack &sp = shared_pack();
    sp.image = levels; sp.mask = mask; sp.n = n; sp.Ng = Ng; sp.levels = pk; sp.flags = pf;
  }
  // the side streams wait for everything queued on the caller's stream (binning produced the levels there)
  PRAD_HIP(hipEventRecord(q.in, (hipStream_t)stream));
  for (int k = 0; k < PRAD_IMG_STREAMS; k++) PRAD_HIP(hipStreamWaitEvent(qs[k], q.in, 0));
  unsigned used = 0;
  // (workspace sets 4, 5, 6: sets 1 - 3 belong to callers that drive side streams of their own, engine.side_queue)
  // ---- side stream 0: GLCM + GLRLM (one sweep) and their formulas ----
  c.workspace = 4 + wso;
  if (classes & (PRAD_IMG_GLCM | PRAD_IMG_GLRLM)) {
    double *gm = nullptr, *rm = nullptr;
    PRAD_TRY(c.get<double>("img_glcm", (size_t)Ng * Ng * Na, &gm));
    PRAD_TRY(c.get<double>("img_glrlm", (size_t)Ng * Nr * Na, &rm));
    int rc = texture_pairs_runs(levels, mask, size, Nd, ang.data(), Na, Ng, Nr, 1, nullptr, 0, force2Ddim, gm, rm, qs[0]);
    if (rc != PRAD_OK) return rc;
    c.workspace = 4 + wso;     // (texture_pairs_runs leaves the lane guard's state)
    c.deferred = true;
    PRAD_TRY(prad_deferred_join(qs[0]));
    if (classes & PRAD_IMG_GLCM) {
      PRAD_TRY(prad_glcm_features_dev(gm, Ng, Na, symmetric, at(0), (int *)at(1), qs[0]));
      if (layout[2] >= 0) {
        rc = prad_glcm_mcc_dev(gm, Ng, Na, symmetric, at(2), qs[0]);
        if (rc == PRAD_E_UNSUPPORTED) layout[2] = -1;       // (too many grey levels for the device MCC: the caller's host route)
        else if (rc != PRAD_OK) return rc;
      }
    }
    if (classes & PRAD_IMG_GLRLM) {
      PRAD_TRY(prad_zone_matrix_features_dev(rm, Ng, Nr, Na, (long long)Nr * Na, (long long)Na, 1LL, nullptr, at(3),
                                             (int *)at(4), qs[0]));     // (size values 1 .. Nr: no table)
    }
    used |= 1u;
  }
  // ---- side stream 3: GLDM + NGTDM (one pass over the neighbourhoods) and their formulas.  Until round 6 they ran behind the
  // sweeps on stream 0, whose chain of small formula kernels (~0.6 ms per 256^3 image) was then the longest of the image: the
  // per-image critical path is the GLSZM's now (profiles/r06_probes.md section 6)
  if (classes & (PRAD_IMG_GLDM | PRAD_IMG_NGTDM)) {
    static const bool own_stream = !getenv("PRAD_IMG_NEIGH_ON_SWEEP_STREAM");
    const int ks = own_stream ? 3 : 0;
    c.workspace = (own_stream ? 7 : 4) + wso;
    const int W = 2 * Nab + 1;
    double *dm = nullptr, *nm = nullptr;
    PRAD_TRY(c.get<double>("img_gldm", (size_t)Ng * W, &dm));
    PRAD_TRY(c.get<double>("img_ngtdm", (size_t)Ng * 3, &nm));
    PRAD_TRY(texture_gldm_ngtdm(levels, mask, size, Nd, angb.data(), Nab, Ng, alpha, dm, nm, qs[ks]));
    c.workspace = (own_stream ? 7 : 4) + wso;
    c.deferred = true;
    if (classes & PRAD_IMG_GLDM) {
      PRAD_TRY(prad_zone_matrix_features_dev(dm, Ng, W, 1, (long long)W, 1LL, 0LL, nullptr, at(5), (int *)at(6), qs[ks]));
    }
    if (classes & PRAD_IMG_NGTDM) PRAD_TRY(prad_ngtdm_features_dev(nm, Ng, at(7), qs[ks]));
    used |= 1u << ks;
  }
  // ---- side stream 1: GLSZM ----
  if (classes & PRAD_IMG_GLSZM) {
    c.workspace = 5 + wso;
    const int rc = prad_glszm_features_dev(levels, mask, size, Nd, angb.data(), Nab, Ng, (int)std::min<long long>(Ns, 2147483647LL),
                                           at(8), (int *)at(9), qs[1]);
    if (rc == PRAD_E_UNSUPPORTED) layout[8] = layout[9] = -1;
    else if (rc != PRAD_OK) return rc;
    else used |= 2u;
    c.deferred = true;
  }
  // ---- side stream 2: first order ----
  if (classes & PRAD_IMG_FIRSTORDER) {
    c.workspace = 6 + wso;
    const int rc = prad_firstorder_queue_dev(raw, raw_dtype, mask, n, Ns, voxelArrayShift, at(10), qs[2]);
    if (rc == PRAD_E_UNSUPPORTED) layout[10] = -1;
    else if (rc != PRAD_OK) return rc;
    else used |= 4u;
    c.deferred = true;
  }
  c.workspace = 0;
  // a verdict mark and an event behind the work of every side stream that got some
  const int t = (int)(q.seq++ % PRAD_IMG_TICKETS);
  for (int k = 0; k < PRAD_IMG_STREAMS; k++) {
    if (!(used & (1u << k))) continue;
    void *f = nullptr;
    PRAD_TRY(c.arena_alloc(sizeof(int), &f));
    q.flag[t][k] = (int *)f;
    *q.flag[t][k] = 0;
    PRAD_TRY(prad_deferred_mark(q.flag[t][k], qs[k]));
    PRAD_HIP(hipEventRecord(q.done[t][k], qs[k]));
  }
  q.used[t] = used | 0x80000000u;     // (in flight, even when every class was declined)
  *ticket = t;
  return PRAD_OK;
}

int prad_image_wait(int ticket) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (ticket < 0 || ticket >= PRAD_IMG_TICKETS) return fail(PRAD_E_ARG, "image_wait: ticket %d", ticket);
  ImageQueues &q = image_queues();
  bool bad = false;
  for (int k = 0; k < PRAD_IMG_STREAMS; k++) {
    if (!(q.used[ticket] & (1u << k))) continue;
    PRAD_HIP(hipEventSynchronize(q.done[ticket][k]));
    bad = bad || *q.flag[ticket][k] != 0;
  }
  q.used[ticket] = 0;
  if (bad) {
    for (int h = 0; h < 2; h++)
      for (int k = 0; k < PRAD_IMG_STREAMS; k++)
        if (q.sp[h][k]) (void)prad_deferred_status(q.sp[h][k]);     // synchronises and clears the sticky word
    return fail(PRAD_E_DEFERRED, "a queued call of the image saw masked levels outside [1, Ng]; repeat it synchronously");
  }
  return PRAD_OK;
}

}  // extern "C"

// ---- the image launcher: prad_image_enqueue_dev issued from a helper thread (round 6) ----------------------------------
// One derived image is ~65 kernel launches, copies and fills behind one C call: 0.25 ms of the calling thread, nine times
// per case, on a path whose bound IS that thread (profiles/r06_probes.md section 6).  prad_image_submit hands the call to a
// launcher thread that belongs to the calling thread (created on first use, with a Context -- workspace, result arena, side
// streams, tickets -- of its own) and returns at once; the caller goes on with its own work (crop + binning of the next image,
// collecting the one before) and asks for the outcome when it needs it.  Jobs of one caller run in the order they were given.
namespace {
struct ImgJob {
  enum Kind { NONE, SUBMIT, WAIT, RELEASE, RETIRE } kind = NONE;
  // SUBMIT arguments
  const int32_t *levels = nullptr;
  const uint8_t *mask = nullptr;
  const void *raw = nullptr;
  int raw_dtype = 0, size[PRAD_MAX_ND] = {0}, Nd = 0, Ng = 0, classes = 0, symmetric = 1, alpha = 0, force2Ddim = -1;
  long long Ns = 0;
  double shift = 0;
  void *stream = nullptr;
  int device = 0;
  // outcome
  int rc = PRAD_OK, wait_rc = PRAD_OK, ticket = -1, layout[16] = {0};
  double *results = nullptr;
  char msg[512] = {0};
  bool submitted = false, waited = false, in_use = false, retiring = false;
  // what the caller needs to wait for the image's GPU work on its own thread: the launcher's events and verdict words
  hipEvent_t ev[PRAD_IMG_STREAMS] = {};
  int *flag[PRAD_IMG_STREAMS] = {};
  unsigned used = 0;
};
constexpr int PRAD_IMG_JOBS = PRAD_IMG_TICKETS;     // (as many as the launcher's context has tickets: a fifth submit fails at once, like prad_image_enqueue_dev)
struct ImgLauncher {
  std::thread th;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::deque<std::pair<int, int>> q;   // (job slot, kind)
  ImgJob jobs[PRAD_IMG_JOBS];
  unsigned seq = 0;
  bool stop = false, started = false;
  void run() {
    for (;;) {
      std::pair<int, int> w;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        w = q.front();
        q.pop_front();
      }
      ImgJob &j = jobs[w.first];
      if (w.second == ImgJob::SUBMIT) {
        int rc = prad_set_device(j.device);
        if (rc == PRAD_OK)
          rc = prad_image_enqueue_dev(j.levels, j.mask, j.raw, j.raw_dtype, j.size, j.Nd, j.Ng, j.Ns, j.classes, j.symmetric, j.alpha,
                                      j.force2Ddim, j.shift, &j.results, j.layout, &j.ticket, j.stream);
        std::lock_guard<std::mutex> lk(m);
        j.rc = rc;
        if (rc != PRAD_OK) snprintf(j.msg, sizeof(j.msg), "%s", err_state().msg);
        if (rc == PRAD_OK) {
          ImageQueues &iq = image_queues();
          j.used = iq.used[j.ticket];
          for (int k = 0; k < PRAD_IMG_STREAMS; k++) {
            j.ev[k] = iq.done[j.ticket][k];
            j.flag[k] = iq.flag[j.ticket][k];
          }
        }
        j.submitted = true;
      } else if (w.second == ImgJob::RETIRE) {      // the caller has waited for the image's events itself: free ticket and slot
        if (prad_set_device(j.device) == PRAD_OK) image_queues().used[j.ticket] = 0;
        std::lock_guard<std::mutex> lk(m);
        j.in_use = false;
        j.retiring = false;
      } else if (w.second == ImgJob::WAIT) {
        const int rc = j.rc == PRAD_OK ? prad_image_wait(j.ticket) : j.rc;
        std::lock_guard<std::mutex> lk(m);
        j.wait_rc = rc;
        if (rc != PRAD_OK && j.rc == PRAD_OK) snprintf(j.msg, sizeof(j.msg), "%s", err_state().msg);
        j.waited = true;
      } else if (w.second == ImgJob::RELEASE) {
        (void)prad_release_workspace();
        std::lock_guard<std::mutex> lk(m);
        j.waited = true;
      }
      cv_done.notify_all();
    }
  }
  ~ImgLauncher() {
    if (started) {
      {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
      }
      cv_work.notify_all();
      if (th.joinable()) th.join();
    }
  }
};
ImgLauncher &img_launcher() {
  static thread_local ImgLauncher L;
  return L;
}
}  // namespace

extern "C" {

int prad_image_submit(const int32_t *levels, const uint8_t *mask, const void *raw, int raw_dtype, const int *size, int Nd, int Ng,
                      long long Ns, int classes, int symmetric, int alpha, int force2Ddim, double voxelArrayShift, void *stream,
                      int *job) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  if (!levels || !mask || !size || !job || Nd < 1 || Nd > PRAD_MAX_ND || Ng < 1) return fail(PRAD_E_ARG, "image_submit: bad arguments");
  ImgLauncher &L = img_launcher();
  std::unique_lock<std::mutex> lk(L.m);
  const int slot = (int)(L.seq % PRAD_IMG_JOBS);
  ImgJob &j = L.jobs[slot];
  if (j.in_use && j.retiring) L.cv_done.wait(lk, [&] { return !j.in_use; });     // (waited for already: the launcher is about to free it)
  if (j.in_use) return fail(PRAD_E_ARG, "image_submit: %d images are in flight on this thread; prad_image_submit_wait one first", PRAD_IMG_JOBS);
  j = ImgJob();
  j.kind = ImgJob::SUBMIT;
  j.levels = levels; j.mask = mask; j.raw = raw; j.raw_dtype = raw_dtype;
  for (int d = 0; d < Nd; d++) j.size[d] = size[d];
  j.Nd = Nd; j.Ng = Ng; j.Ns = Ns; j.classes = classes; j.symmetric = symmetric; j.alpha = alpha; j.force2Ddim = force2Ddim;
  j.shift = voxelArrayShift; j.stream = stream; j.device = c.device;
  j.in_use = true;
  L.seq++;
  L.q.emplace_back(slot, (int)ImgJob::SUBMIT);
  if (!L.started) {
    L.started = true;
    L.th = std::thread([&L] { L.run(); });
  }
  lk.unlock();
  L.cv_work.notify_one();
  *job = slot;
  return PRAD_OK;
}

int prad_image_submit_result(int job, double **results, int *layout) {
  ImgLauncher &L = img_launcher();
  if (job < 0 || job >= PRAD_IMG_JOBS || !results || !layout) return fail(PRAD_E_ARG, "image_submit_result: job %d", job);
  std::unique_lock<std::mutex> lk(L.m);
  ImgJob &j = L.jobs[job];
  if (!j.in_use) return fail(PRAD_E_ARG, "image_submit_result: job %d is not in flight", job);
  L.cv_done.wait(lk, [&] { return j.submitted; });
  if (j.rc != PRAD_OK) return fail(j.rc, "%s", j.msg);
  *results = j.results;
  for (int k = 0; k < 16; k++) layout[k] = j.layout[k];
  return PRAD_OK;
}

int prad_image_submit_wait(int job) {
  ImgLauncher &L = img_launcher();
  if (job < 0 || job >= PRAD_IMG_JOBS) return fail(PRAD_E_ARG, "image_submit_wait: job %d", job);
  std::unique_lock<std::mutex> lk(L.m);
  ImgJob &j = L.jobs[job];
  if (!j.in_use || j.retiring) return fail(PRAD_E_ARG, "image_submit_wait: job %d is not in flight", job);
  // The image's GPU work is waited for HERE, on the calling thread (the launcher may be busy issuing the next image: a wait
  // queued behind that would cost the caller those 0.25 ms per image); the launcher only frees the ticket afterwards.  A
  // failed submit and a voided image (a level outside [1, Ng]: the sticky word has to be cleared on the launcher's
  // streams) take the launcher's own prad_image_wait.
  L.cv_done.wait(lk, [&] { return j.submitted; });
  if (j.rc == PRAD_OK) {
    hipEvent_t ev[PRAD_IMG_STREAMS];
    int *flag[PRAD_IMG_STREAMS];
    const unsigned used = j.used;
    for (int k = 0; k < PRAD_IMG_STREAMS; k++) { ev[k] = j.ev[k]; flag[k] = j.flag[k]; }
    lk.unlock();
    bool bad = false;
    hipError_t herr = hipSuccess;
    for (int k = 0; k < PRAD_IMG_STREAMS && herr == hipSuccess; k++) {
      if (!(used & (1u << k))) continue;
      herr = hipEventSynchronize(ev[k]);
      bad = bad || *flag[k] != 0;
    }
    lk.lock();
    if (herr == hipSuccess && !bad) {
      j.retiring = true;
      L.q.emplace_back(job, (int)ImgJob::RETIRE);
      lk.unlock();
      L.cv_work.notify_one();
      return PRAD_OK;
    }
  }
  L.q.emplace_back(job, (int)ImgJob::WAIT);
  lk.unlock();
  L.cv_work.notify_one();
  lk.lock();
  L.cv_done.wait(lk, [&] { return j.waited; });
  const int rc = j.wait_rc;
  char msg[512];
  snprintf(msg, sizeof(msg), "%s", j.msg);
  j.in_use = false;
  lk.unlock();
  if (rc != PRAD_OK) return fail(rc, "%s", msg);
  return PRAD_OK;
}

// frees the launcher thread's workspace and result arena (outstanding jobs must have been waited for)
int prad_image_submit_release(void) {
  ImgLauncher &L = img_launcher();
  if (!L.started) return PRAD_OK;
  std::unique_lock<std::mutex> lk(L.m);
  for (int k = 0; k < PRAD_IMG_JOBS; k++) {
    if (L.jobs[k].in_use && L.jobs[k].retiring) L.cv_done.wait(lk, [&] { return !L.jobs[k].in_use; });
    if (L.jobs[k].in_use) return fail(PRAD_E_ARG, "image_submit_release: job %d is in flight", k);
  }
  ImgJob &j = L.jobs[L.seq % PRAD_IMG_JOBS];
  j = ImgJob();
  j.in_use = true;
  const int slot = (int)(L.seq % PRAD_IMG_JOBS);
  L.q.emplace_back(slot, (int)ImgJob::RELEASE);
  lk.unlock();
  L.cv_work.notify_one();
  lk.lock();
  L.cv_done.wait(lk, [&] { return j.waited; });
  j.in_use = false;
  return PRAD_OK;
}

}  // extern "C"
