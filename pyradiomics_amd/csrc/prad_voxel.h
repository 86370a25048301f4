// prad_voxel.h -- the fused voxel-based (feature map) entry points of the library (a textual part of prad_api.hip: it uses that
// unit's call setup, workspace and angle code; split off in round 6 so that the matrix dispatch and the voxel routing do not
// share a file):
//   voxslide_schedule            the lane-balanced schedule of the 3-D sliding-window kernel (kernels_voxslide.h)
//   voxel_glcm_features_dev      GLCM feature maps: sliding-window kernel / light window kernel / full window kernel by request
//   voxel_glcm_mcc_dev           MCC maps (per-centre matrices + the eigenvalue kernel)
//   voxel_texture_features_dev   GLRLM / GLSZM / GLDM / NGTDM / first-order maps
// Included by prad_api.hip only, inside its anonymous namespace.
#pragma once

// ------------------------------------------------------------------------------------------------
// fused voxel-based GLCM features
// ------------------------------------------------------------------------------------------------
// The lane-balanced schedule of a 3-D sliding window (kernels_voxslide.h, VoxSlideBal): which pair positions of a plane each of a
// group's sixteen lanes visits.  false: the angles' overflow does not fit the three helper lanes (never for the 13 angles of
// distance 1; the caller then stays on the window kernel).
template <int R>
bool voxslide_schedule_r(const VoxAngles &A, VoxSlideSched *sc) {
  using BL = VoxSlideBal<R>;
  constexpr int D = 2 * R + 1;
  memset(sc, 0, sizeof(*sc));
  memset(sc->rec, 255, sizeof(sc->rec));
  int seg = 0;                                            // helper segments handed out
  for (int a = 0; a < A.na; a++) {
    const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
    int n = 0, nq = 0;
    for (int p = 0; p < D * D; p++) {
      const int pz = p / D, py = p % D;
      if (pz + dz < 0 || pz + dz >= D || py + dy < 0 || py + dy >= D) continue;
      const unsigned d = (unsigned)p | (unsigned)(p + dz * D + dy) << 8 | (unsigned)a << 16 | (unsigned)(dx + 1) << 20 | 1u << 24;
      if (n < BL::NSLOT) {
        sc->slot[a][n] = d;
      } else {
        const int o = n - BL::NSLOT;                      // the o-th overflow position of this angle
        if (o % BL::SEG == 0) {                           // opens a segment
          if (seg >= 3 * BL::NSEG || nq >= BL::MAXQ) return false;
          sc->rec[a][nq++] = (unsigned char)seg++;
        }
        const int sg = seg - 1;
        sc->slot[13 + sg / BL::NSEG][(sg % BL::NSEG) * BL::SEG + o % BL::SEG] = d;
      }
      n++;
    }
  }
  return true;
}
bool voxslide_schedule(const VoxAngles &A, int R, VoxSlideSched *sc) {
  if (A.na != 13) return false;
  return R == 2 ? voxslide_schedule_r<2>(A, sc) : R == 1 ? voxslide_schedule_r<1>(A, sc) : false;
}

int voxel_glcm_features_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                            int Na, int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                            int symmetric, const int *feature_ids, int nfeat, double *out, uint32_t *empty_mask,
                            uint32_t *any_nonempty, hipStream_t s) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!image || !mask || !angles || !voxels || !feature_ids || !out) return fail(PRAD_E_ARG, "voxel_glcm: NULL pointer");
  if (Nvox < 1 || nfeat < 1 || kernelRadius <= 0) return fail(PRAD_E_ARG, "voxel_glcm: Nvox/nfeat/kernelRadius must be >= 1");
  if (Nd > 3 || Ng < 1 || Ng > 64 || Na < 1 || Na > PRAD_VOX_MAX_ANGLES)
    return fail(PRAD_E_UNSUPPORTED, "voxel_glcm: needs Nd <= 3, Ng <= 64, Na <= %d (got Nd=%d Ng=%d Na=%d)",
                PRAD_VOX_MAX_ANGLES, Nd, Ng, Na);
  VoxAngles A;
  A.na = Na;
  for (int a = 0; a < Na; a++) {
    for (int d = 0; d < 4; d++) A.o[a][d] = 0;
    for (int d = 0; d < Nd; d++) {
      const int o = angles[a * Nd + d];
      if (o < -127 || o > 127) return fail(PRAD_E_UNSUPPORTED, "voxel_glcm: angle offset %d", o);
      A.o[a][3 - Nd + d] = (signed char)o;
    }
  }
  unsigned fmask = 0;
  int slot[VF_COUNT];
  for (int f = 0; f < VF_COUNT; f++) slot[f] = 0;
  for (int i = 0; i < nfeat; i++) {
    if (feature_ids[i] < 0 || feature_ids[i] >= VF_COUNT) return fail(PRAD_E_ARG, "voxel_glcm: feature id %d", feature_ids[i]);
    if (fmask & (1u << feature_ids[i])) return fail(PRAD_E_ARG, "voxel_glcm: duplicate feature id %d", feature_ids[i]);
    fmask |= 1u << feature_ids[i];
    slot[feature_ids[i]] = i;
  }
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < Nd; d++) dims[3 - Nd + d] = g.size[d];
  const int f2d3 = force2Ddim >= 0 ? 3 - Nd + force2Ddim : -1;
  PRAD_TRY(c.begin_call(s));
  int *flags = nullptr, *slot_d = nullptr;
  PRAD_TRY(c.get<int>("flags", 4, &flags));
  PRAD_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, s));
  PRAD_TRY(c.get<int>("vox_slots", VF_COUNT, &slot_d));
  PRAD_HIP(hipMemcpyAsync(slot_d, slot, sizeof(int) * VF_COUNT, hipMemcpyHostToDevice, s));
  unsigned *scratch_mask = nullptr;
  PRAD_TRY(c.get<unsigned>("vox_masks", (size_t)Nvox + 4, &scratch_mask));
  unsigned *em = empty_mask ? empty_mask : scratch_mask + 4;
  unsigned *an = any_nonempty ? any_nonempty : scratch_mask;
  PRAD_HIP(hipMemsetAsync(an, 0, sizeof(unsigned), s));
  uint8_t *levels = nullptr;
  PRAD_TRY(neigh_pack(&c, s, g, image, mask, Ng, flags, &levels));
  bool slide_taken = false;
  {
    Timed t(c, "voxel", s);
    const size_t lds = sizeof(u32) * PRAD_VOX_WAVES * ((size_t)Ng * Ng + 5 * (size_t)Ng + 1);
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(((long long)Nvox + PRAD_VOX_WAVES - 1) / PRAD_VOX_WAVES,
                                                                           (long long)cu_count() * 8));
    long long wmax = 1;                       // largest window of the call
    for (int d = 0; d < 3; d++)
      if (d != f2d3 && d >= 3 - Nd) wmax *= std::min(2 * kernelRadius + 1, dims[d]);
    // Sliding-window maps (kernels_voxslide.h): a dense map of the whole volume, the requested centres gathered from it.
    // Taken when the request is what that kernel covers and the centres are dense enough to pay for a whole-volume map.
    const unsigned slide_base = (1u << VF_JointEntropy) | (1u << VF_JointEnergy) | (1u << VF_JointAverage);
    // round 5: fourteen more features with pair-by-pair sums (kernels_voxslide.h WIDE)
    const unsigned slide_wide = (1u << VF_Autocorrelation) | (1u << VF_ClusterProminence) | (1u << VF_ClusterShade) |
                                (1u << VF_ClusterTendency) | (1u << VF_Contrast) | (1u << VF_DifferenceAverage) |
                                (1u << VF_DifferenceVariance) | (1u << VF_Id) | (1u << VF_Idm) | (1u << VF_Idn) | (1u << VF_Idmn) |
                                (1u << VF_InverseVariance) | (1u << VF_SumAverage) | (1u << VF_SumSquares);
    const unsigned slide_feats = slide_base | slide_wide;
    const bool slide_is_wide = (fmask & slide_wide) != 0;
    const bool std13 = Nd == 3 && Na == 13 && f2d3 < 0 && dims[0] > 1 && dims[1] > 1 && dims[2] > 1;
    const bool std4 = Nd == 3 && Na == 4 && (f2d3 == 0 || dims[0] == 1) && dims[1] > 1 && dims[2] > 1;
    bool slide = (std13 || std4) && symmetric && Ng <= 64 && (kernelRadius == 1 || kernelRadius == 2) &&
                 (fmask & ~slide_feats) == 0 && (long long)Nvox * 64 >= g.n && !getenv("PRAD_VOX_NO_SLIDE");
    if (slide) {
      // the kernel's lanes assume the reference's angle order: dx in {-1, 0, 1}, and for the 2-D window no z component
      for (int a = 0; a < Na && slide; a++)
        slide = std::abs(A.o[a][0]) <= 1 && std::abs(A.o[a][1]) <= 1 && std::abs(A.o[a][2]) <= 1 && (std13 || A.o[a][0] == 0);
    }
    VoxSlideSched sched_h;                                 // 3-D windows, the three base features: the lane-balanced schedule
    const bool balanced = slide && std13 && !slide_is_wide;
    if (balanced) slide = voxslide_schedule(A, kernelRadius, &sched_h);
    int z_begin = 0, z_end = -1;
    if (slide) {
      // the slices the centres lie in (one rank of a sharded map owns a z-slab of them: batch.voxel_maps_sharded)
      int *zr_d = nullptr;
      void *zr_h = nullptr;
      PRAD_TRY(c.get<int>("voxslide_zr", 2, &zr_d));
      PRAD_TRY(c.get_pinned("voxslide_zr_h", sizeof(int) * 2, &zr_h));
      ((int *)zr_h)[0] = 0x7fffffff;
      ((int *)zr_h)[1] = -1;
      PRAD_HIP(hipMemcpyAsync(zr_d, zr_h, sizeof(int) * 2, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(voxel_zrange_kernel, dim3((unsigned)std::min<long long>(((long long)Nvox + 255) / 256, 1024)), dim3(256), 0, s,
                         voxels, Nvox, zr_d);
      PRAD_TRY(check_launch("voxel_zrange_kernel"));
      PRAD_HIP(hipMemcpyAsync(zr_h, zr_d, sizeof(int) * 2, hipMemcpyDeviceToHost, s));
      PRAD_HIP(hipStreamSynchronize(s));
      z_begin = std::max(0, ((int *)zr_h)[0]);
      z_end = std::min(dims[0] - 1, ((int *)zr_h)[1]);
      // dense enough inside its slab to pay for a map of the slab?
      if (z_end < z_begin || (long long)Nvox * 8 < (long long)(z_end - z_begin + 1) * dims[1] * dims[2]) slide = false;
    }
    if (slide) {
      slide_taken = true;
      static const VoxSlideLut lut_h = [] {
        VoxSlideLut t;
        auto f = [](int cnt) -> long long {
          if (cnt <= 1) return 0;
          return (long long)std::llround((double)cnt * std::log2((double)cnt) * (double)(1LL << PRAD_VS_FIX));
        };
        for (int k = 0; k < PRAD_VS_LUT; k++) {
          const bool absent = k == PRAD_VS_LUT - 1;            // (a count never gets there: <= 100 pairs per angle)
          const long long nz = k == 0 ? 1LL << PRAD_VS_NNZ_SHIFT : 0;
          t.off[k] = absent ? VoxSlideLutE{0, 0, 0} : VoxSlideLutE{2 * (f(k + 1) - f(k)) + 2 * nz, (1 << 20) | (2 * (2 * k + 1)), 0};
          t.dia[k] = absent ? VoxSlideLutE{0, 0, 0} : VoxSlideLutE{f(2 * k + 2) - f(2 * k) + nz, (1 << 20) | (4 * (2 * k + 1)), 0};
          t.pt[k].lg2T = k ? std::log2(2.0 * k) : 0.0;
          t.pt[k].inv = k ? 1.0 / (2.0 * k) : 0.0;
          t.g_off[k] = t.off[k].g;
          t.g_dia[k] = t.dia[k].g;
        }
        for (int n = 0; n < 16; n++) t.inv_na[n] = n ? 1.0 / (double)n : std::numeric_limits<double>::quiet_NaN();
        return t;
      }();
      VoxSlideLut *lut_dev = nullptr;
      PRAD_TRY(c.get<VoxSlideLut>("voxslide_lut", 1, &lut_dev));
      PRAD_HIP(hipMemcpyAsync(lut_dev, &lut_h, sizeof(lut_h), hipMemcpyHostToDevice, s));
      double *maps = nullptr;
      unsigned *emap = nullptr;
      PRAD_TRY(c.get<double>("voxslide_maps", (size_t)nfeat * g.n, &maps));
      PRAD_TRY(c.get<unsigned>("voxslide_empty", (size_t)g.n, &emap));
      VoxSlideSlots sl;
      for (int f = 0; f < VF_COUNT; f++) sl.s[f] = ((fmask >> f) & 1u) ? slot[f] : -1;
      VoxSlideSched *sched_dev = nullptr;
      if (balanced) {
        void *sc_h = nullptr;
        PRAD_TRY(c.get<VoxSlideSched>("voxslide_sched", 1, &sched_dev));
        PRAD_TRY(c.get_pinned("voxslide_sched_h", sizeof(sched_h), &sc_h));
        memcpy(sc_h, &sched_h, sizeof(sched_h));
        PRAD_HIP(hipMemcpyAsync(sched_dev, sc_h, sizeof(sched_h), hipMemcpyHostToDevice, s));
      }
      VoxSlideLutK *lutk_dev = nullptr;
      VoxSlideLutK lutk_h;
      if (slide_is_wide) {
        const double sc = (double)(1LL << PRAD_VS_FIX), ng = (double)Ng;
        for (int k = 0; k < PRAD_VS_KMAX; k++) {
          const double kd = (double)k;
          lutk_h.g[0][k] = std::llround(sc / (1.0 + kd));                       // Id      (glcm.py:741)
          lutk_h.g[1][k] = std::llround(sc / (1.0 + kd * kd));                  // Idm     (:662)
          lutk_h.g[2][k] = std::llround(sc / (1.0 + kd / ng));                  // Idn     (:759)
          lutk_h.g[3][k] = std::llround(sc / (1.0 + (kd * kd) / (ng * ng)));    // Idmn    (:726)
          lutk_h.g[4][k] = k ? std::llround(sc / (kd * kd)) : 0;                // InverseVariance (:773-776, k = 0 skipped)
        }
        PRAD_TRY(c.get<VoxSlideLutK>("voxslide_lutk", 1, &lutk_dev));
        PRAD_HIP(hipMemcpyAsync(lutk_dev, &lutk_h, sizeof(lutk_h), hipMemcpyHostToDevice, s));
      }
#define PRAD_SLIDE_TW(RR, TWOD, RUNL, TBB, WV, WD, JJ, LT)                                                                         \
  do {                                                                                                                      \
    constexpr size_t lds_s = voxel_glcm_slide_lds<RR, TWOD, RUNL, TBB, WV, WD>();                                           \
    static_assert(lds_s <= 160 * 1024, "LDS");                                                                              \
    PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_glcm_slide_kernel<RR, TWOD, RUNL, TBB, WV, WD, JJ, LT>),     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));                                  \
    const int wv = WV, rows = TWOD ? 16 : 4;                                                                                \
    const int nruns = (dims[2] + RUNL - 1) / RUNL;                                                                          \
    hipLaunchKernelGGL((voxel_glcm_slide_kernel<RR, TWOD, RUNL, TBB, WV, WD, JJ, LT>), dim3((nruns + wv - 1) / wv, (dims[1] + rows - 1) / rows, z_end - z_begin + 1), \
                       dim3(64 * wv), lds_s, s, levels, dims[0], dims[1], dims[2], A, Ng, lut_dev, lutk_dev, sched_dev, sl, maps, emap, flags, z_begin); \
  } while (0)
      // (the WIDE instantiation carries 2.5 KB of g(k) tables in LDS: one wave less where the base shape fills the 160 KB)
#define PRAD_SLIDE_T(RR, TWOD, RUNL, TBB, WV, WVW)                                                                          \
  do {                                                                                                                      \
    if (slide_is_wide) PRAD_SLIDE_TW(RR, TWOD, RUNL, TBB, WVW, true, true, false);                                          \
    else if (sl.s[VF_JointAverage] >= 0) PRAD_SLIDE_TW(RR, TWOD, RUNL, TBB, WV, false, true, false);                        \
    else if (sl.s[VF_JointEnergy] >= 0) PRAD_SLIDE_TW(RR, TWOD, RUNL, TBB, WV, false, false, false);                        \
    else PRAD_SLIDE_TW(RR, TWOD, RUNL, TBB, WV, false, false, true);                                                        \
  } while (0)
      // the lanes' private count tables hold Ng (Ng + 1) / 2 bytes: table size and waves per workgroup by level count
      // (brain1 under exampleVoxel.yaml has 33 levels).  2-D windows at <= 32 levels: runs of 32 centres instead of 64 -- the
      // staged planes of a wave halve, and FOUR waves' tables fit the 160 KB instead of three (a SIMD of every CU sat idle)
#define PRAD_SLIDE(RR, TWOD, RUNL)                                                                                          \
  do {                                                                                                                      \
    if (Ng <= 32) PRAD_SLIDE_T(RR, TWOD, (TWOD ? 32 : RUNL), 532, 4, 4);                                                    \
    else if (Ng <= 40) PRAD_SLIDE_T(RR, TWOD, RUNL, 828, (TWOD ? 2 : 3), 2);                                                \
    else if (Ng <= 48) PRAD_SLIDE_T(RR, TWOD, RUNL, 1180, (TWOD ? 1 : 2), (TWOD ? 1 : 2));                                  \
    else PRAD_SLIDE_T(RR, TWOD, RUNL, 2084, 1, 1);                                                                          \
  } while (0)
      if (std13 && kernelRadius == 2) PRAD_SLIDE(2, false, 64);
      else if (std13) PRAD_SLIDE(1, false, 64);
      else if (kernelRadius == 2) PRAD_SLIDE(2, true, 64);
      else PRAD_SLIDE(1, true, 64);
#undef PRAD_SLIDE
#undef PRAD_SLIDE_T
#undef PRAD_SLIDE_TW
      PRAD_TRY(check_launch("voxel_glcm_slide_kernel"));
      const unsigned allbits = (1u << Na) - 1u;
      const unsigned gb = (unsigned)std::min<long long>(((long long)Nvox + 255) / 256, (long long)cu_count() * 16);
      hipLaunchKernelGGL(voxel_map_gather_kernel, dim3(gb), dim3(256), 0, s, maps, emap, dims[0], dims[1], dims[2], nfeat, Nvox,
                         voxels, allbits, out, em, an);
      PRAD_TRY(check_launch("voxel_map_gather_kernel"));
    } else if ((fmask & ~PRAD_VF_LIGHT) == 0 && wmax <= 64 && !getenv("PRAD_VOX_NO_LIGHT")) {
      const size_t lds2 = sizeof(double) * 2 * 257 + sizeof(u32) * PRAD_VOX_WAVES * (size_t)Ng * Ng;
      PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_glcm_light_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      hipLaunchKernelGGL(voxel_glcm_light_kernel, dim3(gx), dim3(64 * PRAD_VOX_WAVES), lds2, s, levels, dims[0], dims[1],
                         dims[2], A, Ng, Nvox, voxels, Nd, kernelRadius, f2d3, symmetric, fmask, slot_d, out, em, an, flags);
    } else if ((fmask & ~PRAD_VF_LIGHT) == 0) {
      PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_glcm_kernel<PRAD_VF_LIGHT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(voxel_glcm_kernel<PRAD_VF_LIGHT>, dim3(gx), dim3(64 * PRAD_VOX_WAVES), lds, s, levels, dims[0], dims[1],
                         dims[2], A, Ng, Nvox, voxels, Nd, kernelRadius, f2d3, symmetric, fmask, slot_d, out, em, an, flags);
    } else {
      PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_glcm_kernel<0xffffffffu>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(voxel_glcm_kernel<0xffffffffu>, dim3(gx), dim3(64 * PRAD_VOX_WAVES), lds, s, levels, dims[0], dims[1],
                         dims[2], A, Ng, Nvox, voxels, Nd, kernelRadius, f2d3, symmetric, fmask, slot_d, out, em, an, flags);
    }
    PRAD_TRY(check_launch("voxel_glcm_kernel"));
  }
  void *fh = nullptr;
  PRAD_TRY(c.get_pinned("flags_h", sizeof(int) * 4, &fh));
  PRAD_HIP(hipMemcpyAsync(fh, flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  if (((int *)fh)[0]) return fail(PRAD_E_UNSUPPORTED, "voxel_glcm: masked levels outside [1, Ng]; use the matrix path");
  c.last_path = "voxel-fused";
  c.last_variant = slide_taken ? "slide" : "window";
  return PRAD_OK;
}

int voxel_glcm_mcc_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                       int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, int symmetric, double *out,
                       hipStream_t s) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!image || !mask || !angles || !voxels || !out) return fail(PRAD_E_ARG, "voxel_glcm_mcc: NULL pointer");
  if (Nvox < 1 || kernelRadius <= 0) return fail(PRAD_E_ARG, "voxel_glcm_mcc: Nvox/kernelRadius must be >= 1");
  if (Nd > 3 || Ng < 1 || Ng > 64 || Na < 1 || Na > PRAD_VOX_MAX_ANGLES)
    return fail(PRAD_E_UNSUPPORTED, "voxel_glcm_mcc: needs Nd <= 3, Ng <= 64, Na <= %d", PRAD_VOX_MAX_ANGLES);
  VoxAngles A;
  A.na = Na;
  for (int a = 0; a < Na; a++) {
    for (int d = 0; d < 4; d++) A.o[a][d] = 0;
    for (int d = 0; d < Nd; d++) {
      const int o = angles[a * Nd + d];
      if (o < -127 || o > 127) return fail(PRAD_E_UNSUPPORTED, "voxel_glcm_mcc: angle offset %d", o);
      A.o[a][3 - Nd + d] = (signed char)o;
    }
  }
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < Nd; d++) dims[3 - Nd + d] = g.size[d];
  const int f2d3 = force2Ddim >= 0 ? 3 - Nd + force2Ddim : -1;
  PRAD_TRY(c.begin_call(s));
  int *flags = nullptr;
  PRAD_TRY(c.get<int>("flags", 4, &flags));
  PRAD_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, s));
  uint8_t *levels = nullptr;
  PRAD_TRY(neigh_pack(&c, s, g, image, mask, Ng, flags, &levels));
  {
    Timed t(c, "voxel", s);
    const int nmax = std::min(Ng, PRAD_MCC_NMAX);
    const size_t per_wave = (mcc_scratch_bytes(Ng, nmax) + sizeof(u32) * (size_t)Ng * Ng + 15) & ~(size_t)15;
    // two waves per workgroup unless their scratch areas exceed the 160 KiB a workgroup may declare (Ng = 63, 64:
    // 2 x 84 KB) -- then one wave per workgroup, and more workgroups
    constexpr size_t kLdsMax = 160 * 1024;
    if (per_wave > kLdsMax) {
      (void)c.end_call(s);
      return fail(PRAD_E_UNSUPPORTED, "voxel_glcm_mcc: %zu B of LDS per kernel window exceed the device limit; use the matrix path", per_wave);
    }
    const int waves = per_wave * PRAD_MCC_WAVES <= kLdsMax ? PRAD_MCC_WAVES : 1;
    const size_t lds = per_wave * waves;
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(((long long)Nvox + waves - 1) / waves,
                                                                           (long long)cu_count() * 8 * (PRAD_MCC_WAVES / waves)));
    PRAD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&voxel_glcm_mcc_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(voxel_glcm_mcc_kernel, dim3(gx), dim3(64 * waves), lds, s, levels, dims[0], dims[1], dims[2],
                       A, Ng, Nvox, voxels, Nd, kernelRadius, f2d3, symmetric, nmax, out, flags + 3, flags);
    PRAD_TRY(check_launch("voxel_glcm_mcc_kernel"));
  }
  void *fh = nullptr;
  PRAD_TRY(c.get_pinned("flags_h", sizeof(int) * 4, &fh));
  PRAD_HIP(hipMemcpyAsync(fh, flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  if (((int *)fh)[0]) return fail(PRAD_E_UNSUPPORTED, "voxel_glcm_mcc: masked levels outside [1, Ng]; use the matrix path");
  c.last_path = "voxel-fused";
  return PRAD_OK;
}

int voxel_texture_features_dev(int family, const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                               const int *angles, int Na, int Ng, int alpha, int Nvox, const int *voxels, int kernelRadius,
                               int force2Ddim, const int *feature_ids, int nfeat, double *out, hipStream_t s) {
  Context &c = ctx();
  PRAD_TRY(c.ensure_device());
  Geo g;
  PRAD_TRY(make_geo(size, Nd, &g));
  if (!image || !mask || !angles || !voxels || !feature_ids || !out) return fail(PRAD_E_ARG, "voxel_texture: NULL pointer");
  if (Nvox < 1 || nfeat < 1 || kernelRadius <= 0) return fail(PRAD_E_ARG, "voxel_texture: Nvox/nfeat/kernelRadius must be >= 1");
  if (family < PRAD_VT_GLDM || family > PRAD_VT_GLSZM) return fail(PRAD_E_ARG, "voxel_texture: family %d", family);
  long long W = 1;
  for (int d = 0; d < Nd; d++)
    if (d != force2Ddim) W *= std::min(2 * kernelRadius + 1, g.size[d]);
  if (Nd > 3 || Ng < 1 || Ng > 255 || Na < 1 || Na > PRAD_VOX_MAX_ANGLES || W > PRAD_VT_MAXW)
    return fail(PRAD_E_UNSUPPORTED, "voxel_texture: needs Nd <= 3, Ng <= 255, Na <= %d, <= %d voxels per kernel "
                "(got Nd=%d Ng=%d Na=%d W=%lld)", PRAD_VOX_MAX_ANGLES, PRAD_VT_MAXW, Nd, Ng, Na, W);
  const int fcount = family == PRAD_VT_NGTDM ? (int)NF_COUNT : (int)ZF_COUNT;
  for (int i = 0; i < nfeat; i++)
    if (feature_ids[i] < 0 || feature_ids[i] >= fcount) return fail(PRAD_E_ARG, "voxel_texture: feature id %d", feature_ids[i]);
  VoxAngles A;
  A.na = Na;
  for (int a = 0; a < Na; a++) {
    for (int d = 0; d < 4; d++) A.o[a][d] = 0;
    for (int d = 0; d < Nd; d++) {
      const int o = angles[a * Nd + d];
      if (o < -127 || o > 127) return fail(PRAD_E_UNSUPPORTED, "voxel_texture: angle offset %d", o);
      A.o[a][3 - Nd + d] = (signed char)o;
    }
  }
  int dims[3] = {1, 1, 1};
  for (int d = 0; d < Nd; d++) dims[3 - Nd + d] = g.size[d];
  const int f2d3 = force2Ddim >= 0 ? 3 - Nd + force2Ddim : -1;
  PRAD_TRY(c.begin_call(s));
  int *flags = nullptr, *ids_d = nullptr;
  PRAD_TRY(c.get<int>("flags", 4, &flags));
  PRAD_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, s));
  PRAD_TRY(c.get<int>("vt_ids", (size_t)nfeat, &ids_d));
  PRAD_HIP(hipMemcpyAsync(ids_d, feature_ids, sizeof(int) * nfeat, hipMemcpyHostToDevice, s));
  uint8_t *levels = nullptr;
  PRAD_TRY(neigh_pack(&c, s, g, image, mask, Ng, flags, &levels));
  {
    Timed t(c, "voxel", s);
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>(((long long)Nvox + PRAD_VT_WAVES - 1) / PRAD_VT_WAVES,
                                                                           (long long)cu_count() * 16));
    if (family == PRAD_VT_NGTDM) {
      hipLaunchKernelGGL(voxel_ngtdm_kernel, dim3(gx), dim3(64 * PRAD_VT_WAVES), 0, s, levels, dims[0], dims[1], dims[2], A,
                         Ng, Nvox, voxels, Nd, kernelRadius, f2d3, ids_d, nfeat, out, flags);
      PRAD_TRY(check_launch("voxel_ngtdm_kernel"));
    } else {
      double *tabs = nullptr;          // log2(n), 1 / n, 1 / n^2 for n <= PRAD_VT_MAXW (kernels_voxtex.h zl_accumulate)
      const bool fresh = !c.has("vt_tables");
      PRAD_TRY(c.get<double>("vt_tables", 3 * PRAD_VT_TAB, &tabs));
      if (fresh) {
        std::vector<double> h(3 * PRAD_VT_TAB, 0.0);
        for (int i = 1; i < PRAD_VT_TAB; i++) {
          h[i] = log2((double)i);
          h[PRAD_VT_TAB + i] = 1.0 / (double)i;
          h[2 * PRAD_VT_TAB + i] = 1.0 / ((double)i * (double)i);
        }
        PRAD_HIP(hipMemcpy(tabs, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
      }
      hipLaunchKernelGGL(voxel_zonelike_kernel, dim3(gx), dim3(64 * PRAD_VT_WAVES), 0, s, family, levels, dims[0], dims[1],
                         dims[2], A, alpha, Nvox, voxels, Nd, kernelRadius, f2d3, ids_d, nfeat, out, flags, (const double *)tabs);
      PRAD_TRY(check_launch("voxel_zonelike_kernel"));
    }
  }
  void *fh = nullptr;
  PRAD_TRY(c.get_pinned("flags_h", sizeof(int) * 4, &fh));
  PRAD_HIP(hipMemcpyAsync(fh, flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  PRAD_TRY(c.end_call(s));
  PRAD_HIP(hipStreamSynchronize(s));
  if (((int *)fh)[0]) return fail(PRAD_E_UNSUPPORTED, "voxel_texture: masked levels outside [1, Ng]; use the matrix path");
  c.last_path = "voxel-fused";
  return PRAD_OK;
}
