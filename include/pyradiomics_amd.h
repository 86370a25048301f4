/*
 * pyradiomics_amd.h -- C ABI of the MI355X (gfx950) texture-matrix engine.
 *
 * This is the drop-in boundary for pyradiomics' native operator layer.  It replaces
 *   - the eight C prototypes of radiomics/src/cmatrices.h:1-8, and
 *   - the per-voxel driver loops of the CPython wrapper radiomics/src/_cmatrices.c
 *     (set_bb :1120-1147 and the `for v in Nvox` loops :203-222, :355-377, :552-570, :700-718,
 *     :850-868), which are folded into the calls below so that a whole voxel batch is ONE launch.
 * The Python face `radiomics.cMatrices` (radiomics/__init__.py:343-349) is restored on top of this
 * ABI by pyradiomics_amd/cmatrices.py with ctypes; see INTEGRATION.md for the binding a reference
 * maintainer would add.
 *
 * Conventions (all functions):
 *   - plain pointers and ints only; no exceptions cross the boundary; thread-safe per device.
 *   - `image` is int32 [size[0]]...[size[Nd-1]] C-contiguous, `mask` is uint8/bool of the same shape
 *     (what _cmatrices.c:1023-1085 coerces its inputs to).  1 <= Nd <= PRAD_MAX_ND.
 *   - voxel mode: `voxels` is int32 [Nd][Nvox] (np.where layout, _cmatrices.c:1087-1118); kernel v covers
 *     centre +- kernelRadius clamped to the array, collapsed in force2Ddim.  Segment mode: voxels=NULL,
 *     Nvox=1, box = whole array.  force2Ddim = -1 when force2D is off (_cmatrices.c:126).
 *   - outputs are caller-allocated float64 arrays in the reference's layouts; they need NOT be
 *     pre-zeroed (the reference memsets them itself, e.g. _cmatrices.c:185).
 *   - return value: PRAD_OK (1) on success; PRAD_INDEX_ERROR (0) where the reference's core returns 0
 *     ("index out of range", surfaced by the wrapper as IndexError, _cmatrices.c:219,566,714,864);
 *     negative PRAD_E_* for everything else; prad_last_error() gives the message.
 *   - the `_dev` variants take DEVICE pointers for image/mask/voxels/outputs and a hipStream_t (as
 *     void*); they enqueue on that stream and synchronise it before returning the status.
 *     The host variants stage through the library's per-device workspace.
 */
#ifndef PYRADIOMICS_AMD_H
#define PYRADIOMICS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRAD_MAX_ND 8

#define PRAD_OK 1
#define PRAD_INDEX_ERROR 0
#define PRAD_E_ARG (-1)         /* bad argument (NULL pointer, Nd out of range, kernelRadius <= 0 with voxels...) */
#define PRAD_E_HIP (-2)         /* HIP runtime failure; message in prad_last_error() */
#define PRAD_E_NOMEM (-3)       /* device or host allocation failed */
#define PRAD_E_UNSUPPORTED (-4) /* valid for the reference but not implemented here (stated in the message) */
#define PRAD_E_INDEX (-5)       /* GLSZM phase 1 only: the reference's calculate_glszm would return -1 (scratch sized by
                                   Ns exhausted, cmatrices.c:174,226,245,274) -> IndexError in the wrapper */

/* ---- runtime ---------------------------------------------------------------------------------- */
const char *prad_version(void);
const char *prad_last_error(void);       /* thread-local, valid until the next failing call */
int prad_device_count(void);             /* number of visible HIP devices (0 if none) */
int prad_set_device(int device);         /* selects the device used by this thread's later calls */
int prad_get_device(void);
/* The library keeps grow-only scratch buffers per calling thread and device (packed levels, accumulators, union-find
 * labels, staging copies of host inputs ...) so that repeated calls of the same shape allocate nothing.
 * prad_workspace_bytes: device bytes currently held by the calling thread; prad_release_workspace: free them (and
 * the pinned host buffers); any pending two-phase GLSZM state is dropped. */
long long prad_workspace_bytes(void);
int prad_release_workspace(void);
/* name of the code path taken by this thread's last calculate_* call ("sweep", "generic", ...);
 * lets tests assert that the fast kernels (not a fallback) produced a result. */
const char *prad_last_path(void);
/* which sweep kernels the plan of this thread's last GLCM / GLRLM call chose: "fw" (fixed window, fused table: <= 44 grey
 * levels, rows of 65..512 voxels), "fw2" (fixed window, two tables, 16-bit levels: 45+ levels), "lines" (wrapped lines). */
const char *prad_last_variant(void);
/* Total device time (ms, HIP events on the work stream) of this thread's last calculate_* call, and
 * the time of its dominant kernel family; used by bench.py for the roofline figure. */
double prad_last_device_ms(void);
double prad_last_kernel_ms(const char *kernel_family);
/* Accumulated timing over many calls (the per-call figures above need the call to have finished, i.e. one host
 * synchronisation per call).  prad_timing_begin: start keeping the event brackets of every following call of this
 * thread; prad_timing_ms: sum of the brackets of a kernel family (NULL: whole calls) since then, in ms -- synchronises
 * on the last recorded event; prad_timing_calls: calls recorded; prad_timing_end: stop and drop the records.
 * prad_timing_begin_only(family): as prad_timing_begin, but ONLY the launches of that kernel family are bracketed (no call
 * brackets either) -- two event records per call instead of ten: a timed loop that wants the duration of its dominant
 * kernel without paying for the rest (each record costs the stream 3 - 6 us; bench.py).  prad_timing_count(family): how
 * many brackets of the family were recorded (NULL: calls). */
int prad_timing_begin(void);
int prad_timing_begin_only(const char *kernel_family);
int prad_timing_count(const char *kernel_family);
double prad_timing_ms(const char *kernel_family);
int prad_timing_calls(void);
int prad_timing_end(void);
/* Deferred mode for the device-pointer GLCM / GLRLM entry points (segment mode): with on != 0 a call only ENQUEUES its
 * kernels on the caller's stream and returns without synchronising the host, so consecutive volumes pipeline on the
 * GPU.  The one thing the host learns late is whether a volume held masked levels outside [1, Ng] (which the
 * synchronous call answers by re-running on the exact generic kernels): every deferred call latches that into a sticky
 * device flag; prad_deferred_status synchronises the stream, returns PRAD_OK, or PRAD_E_DEFERRED if any deferred call
 * since the last query saw such levels (its outputs are then undefined: repeat that call synchronously), and clears
 * the flag.  Calls the sweep kernels cannot serve (voxel mode, Nd > 3, ...) run synchronously as before.
 * Lanes: deferred whole-volume calls are dealt round-robin onto `n` internal streams (default 2, environment
 * PRAD_LANES; 1 = everything stays on the caller's stream), each with its own workspace and each waiting for the work
 * queued on the caller's stream at the time of the call, so the kernels of consecutive volumes share the GPU.  The
 * caller's stream does NOT wait for a lane: inputs and outputs of a deferred call belong to the library until
 * prad_deferred_status returns (it synchronises the stream and every lane).
 * Pipeline (the default deferred mode; prad_set_deferred_mode(0) or PRAD_DEFERRED_MODE=lanes selects the lanes):
 * deferred whole-volume calls stay on the caller's stream and form a two-stage pipeline -- call N launches the walks of
 * volume N-1 with the PACK of volume N riding in the same launch as a side job of the walking waves (the pack is
 * HBM-bound, the walk issue-bound: the pack's memory time disappears), then finalizes volume N-1.  Volume N is walked by
 * the next deferred call, or by prad_deferred_join / prad_deferred_status, which flush the pipeline.  As with the lanes,
 * inputs and outputs of a deferred call belong to the library until one of those two returns.  The pipeline takes the
 * volumes of the fixed-window kernels -- fused table (up to 44 levels) and, since round 5, two tables (45 .. 160 levels;
 * a volume of the other kind than the one before it packs in a launch of its own) -- every other whole-volume call is
 * dealt onto the lanes. */
#define PRAD_E_DEFERRED (-6)
int prad_set_deferred(int on);
int prad_set_lanes(int n);                 /* 0 = default; returns PRAD_E_ARG outside [0, 4] */
int prad_set_deferred_mode(int mode);      /* 1 pipeline, 0 lanes, -1 environment default (PRAD_DEFERRED_MODE); flushes */
/* makes `stream` wait ON THE DEVICE for every deferred call issued so far (no host synchronisation): afterwards work
 * queued on `stream` may read the outputs of those calls; the levels verdict still needs prad_deferred_status */
int prad_deferred_join(void *stream);
int prad_deferred_status(void *stream);
/* queues a copy of the sticky verdict word into `flag` (an int inside the result arena) on `stream`: a caller that waits
 * on an EVENT recorded after this call -- not on the whole stream, so that work queued later keeps running -- reads
 * *flag != 0 as "a deferred call before the mark saw levels outside [1, Ng]" (then prad_deferred_status clears it) */
int prad_deferred_mark(int *flag, void *stream);
/* Result arena + enqueue-only feature calls (the case pipeline: every matrix and feature kernel of one derived image is
 * queued before the host waits once).  prad_result_alloc hands out pinned host memory from a per-thread ring of 4 MiB
 * (64-byte aligned; at most 1 MiB per allocation; an allocation stays untouched until 4 MiB more have been handed out).
 * In deferred mode (prad_set_deferred(1)):
 *  - prad_calculate_gldm_dev / prad_calculate_ngtdm_dev in segment mode only enqueue their kernels; a level outside
 *    [1, Ng] under the mask is latched like the GLCM / GLRLM verdict (prad_deferred_status -> PRAD_E_DEFERRED);
 *  - prad_glcm_features_dev, prad_zone_matrix_features_dev, prad_ngtdm_features_dev and prad_glcm_mcc_dev whose output
 *    pointers lie INSIDE the arena enqueue their kernels and the device-to-host copies into those pointers and return:
 *    the values are valid once the stream has been synchronised (prad_deferred_status, or an event recorded after the
 *    call).  The arena is mapped host memory: the formula kernels store their few values straight into it (no copy
 *    operation in the stream; PRAD_ZERO_COPY=0 restores device buffer + copy).  prad_glcm_mcc_dev then needs room for Na + 1 doubles: out[Na] != 0 says that more than 64 grey levels
 *    occurred (the synchronous call's PRAD_E_UNSUPPORTED) and out[0..Na) is void.
 * With outputs outside the arena these calls stay synchronous.  All calls of one pipeline go to ONE stream (the
 * workspace buffers are recycled in stream order).  No reference analogue (the reference computes the formulas in numpy
 * from host matrices: glcm.py, glrlm.py, ...). */
int prad_result_alloc(size_t bytes, void **out);
/* Workspace set of the calling thread (0 = default, 1..7): the library recycles named scratch buffers in stream order, so
 * a thread that issues calls on a SECOND stream concurrently selects another set for them (and switches back).  The
 * case pipeline queues its enqueue-only classes on a side stream under set 1 while first order / GLSZM run on the main
 * stream under set 0. */
int prad_set_workspace(int id);
/* One derived image, every class, one call: the enqueue half of the case pipeline in native code.  levels / mask: DEVICE
 * (discretised image, ROI); raw: DEVICE undiscretised image of dtype raw_dtype (first order; NULL without that class);
 * Ns = ROI voxels; classes = sum of PRAD_IMG_* (PRAD_IMG_MCC with PRAD_IMG_GLCM adds the MCC); distance-1 neighbourhoods,
 * force2Ddim as elsewhere.  Queues, without waiting: GLCM + GLRLM (one sweep), their formulas and the MCC, GLDM + NGTDM (one
 * pass over the neighbourhoods) and their formulas on an internal stream, GLSZM on a second, first order on a third -- each
 * stream waits for the work queued on `stream` so far and has its own workspace set -- and a verdict mark + event behind
 * each.  *results: one block of the result arena; layout[k] (int[16]) = offset of part k in doubles, -1 = not asked for or
 * declined by the device route (the caller then computes that class on its own):
 *   0 GLCM values [Na][23], 1 GLCM "angle empty" flags (int [Na]), 2 MCC [Na] + verdict, 3 GLRLM values [Na][16], 4 flags,
 *   5 GLDM values [16], 6 flag, 7 NGTDM values [5], 8 GLSZM values [16] + verdict, 9 flag, 10 first order [15] + verdict;
 *   layout[11] = Na, layout[12] = doubles in the block.
 * prad_image_wait(ticket) waits for the image's work only (later images keep running; up to 4 images may be in flight per
 * thread) and returns PRAD_OK, or PRAD_E_DEFERRED when a queued call saw levels outside [1, Ng] (values void).  Inputs
 * must stay alive until then.  No reference analogue (base.py:181-198 evaluates class after class on the host). */
#define PRAD_IMG_GLCM 1
#define PRAD_IMG_GLRLM 2
#define PRAD_IMG_GLDM 4
#define PRAD_IMG_NGTDM 8
#define PRAD_IMG_GLSZM 16
#define PRAD_IMG_FIRSTORDER 32
#define PRAD_IMG_MCC 64
int prad_image_enqueue_dev(const int32_t *levels, const uint8_t *mask, const void *raw, int raw_dtype, const int *size,
                           int Nd, int Ng, long long Ns, int classes, int symmetric, int alpha, int force2Ddim,
                           double voxelArrayShift, double **results, int *layout, int *ticket, void *stream);
int prad_image_wait(int ticket);
/* prad_image_enqueue_dev issued from a LAUNCHER thread that belongs to the calling thread (created on first use; it has a
 * workspace, result arena, side streams and tickets of its own).  One derived image is ~65 launches, copies and fills: 0.25 ms
 * of host time per image on a path that is bound by its host thread (featureextractor.py:371-395 + base.py:181-198 in the
 * reference are that thread too).  prad_image_submit copies the arguments and returns at once (*job: slot of up to four in
 * flight per thread); the side streams wait for what was queued on `stream` by the time the launcher gets to the job, i.e.
 * at least for everything the caller queued before the call.  prad_image_submit_result blocks until the launcher has ISSUED
 * the job (not until the GPU has run it) and returns prad_image_enqueue_dev's code, result block and layout;
 * prad_image_submit_wait waits for the image's GPU work like prad_image_wait and frees the slot (it must be called exactly
 * once per job, also after a failed submit).  Jobs of one calling thread are issued in order.  prad_image_submit_release:
 * prad_release_workspace on the launcher's context (nothing may be in flight). */
int prad_image_submit(const int32_t *levels, const uint8_t *mask, const void *raw, int raw_dtype, const int *size, int Nd, int Ng,
                      long long Ns, int classes, int symmetric, int alpha, int force2Ddim, double voxelArrayShift, void *stream,
                      int *job);
int prad_image_submit_result(int job, double **results, int *layout);
int prad_image_submit_wait(int job);
int prad_image_submit_release(void);

/* ---- angles: cmatrices.h get_angle_count / build_angles (cmatrices.c:756-892) ------------------- */
/* returns the number of angles, 0 on invalid distance (as the reference) */
int prad_get_angle_count(const int *size, const int *distances, int Nd, int Ndist, int bidirectional,
                         int force2Ddim);
/* returns 0 on success, 1 on invalid distance (as the reference); angles is int32 [Na][Nd] */
int prad_build_angles(const int *size, const int *distances, int Nd, int Ndist, int force2Ddim, int Na,
                      int *angles);

/* ---- GLCM: calculate_glcm (cmatrices.c:4-92) ---------------------------------------------------- */
/* glcm: float64 [Nvox][Ng][Ng][Na] */
int prad_calculate_glcm(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                        const int *angles, int Na, int Ng,
                        int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                        double *glcm);
int prad_calculate_glcm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                            const int *angles, int Na, int Ng,
                            int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                            double *glcm, void *stream);

/* ---- GLRLM: calculate_glrlm (cmatrices.c:299-541) ----------------------------------------------- */
/* glrlm: float64 [Nvox][Ng][Nr][Na] */
int prad_calculate_glrlm(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                         const int *angles, int Na, int Ng, int Nr,
                         int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                         double *glrlm);
int prad_calculate_glrlm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                             const int *angles, int Na, int Ng, int Nr,
                             int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             double *glrlm, void *stream);

/* ---- GLCM + GLRLM in one call (the headline path: one discretised volume -> both matrices).
 * Same results as the two calls above with the same `angles`; the volume is packed once and every
 * angle is swept once for both matrices.  Either output may be NULL to skip it. ------------------- */
int prad_calculate_glcm_glrlm(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                              const int *angles, int Na, int Ng, int Nr,
                              int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                              double *glcm, double *glrlm);
int prad_calculate_glcm_glrlm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                                  const int *angles, int Na, int Ng, int Nr,
                                  int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                                  double *glcm, double *glrlm, void *stream);

/* ---- GLDM: calculate_gldm (cmatrices.c:660-754) -------------------------------------------------- */
/* gldm: float64 [Nvox][Ng][2*Na+1] */
int prad_calculate_gldm(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                        const int *angles, int Na, int Ng, int alpha,
                        int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                        double *gldm);
int prad_calculate_gldm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                            const int *angles, int Na, int Ng, int alpha,
                            int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                            double *gldm, void *stream);

/* ---- NGTDM: calculate_ngtdm (cmatrices.c:543-658) ------------------------------------------------ */
/* ngtdm: float64 [Nvox][Ng][3].  Column 1 (sum of |i - mean(neighbours)|) is evaluated as
 * sum_c (sum_p |c*i - s|)/c with exact integer inner sums in segment mode (correctly rounded to a few ulp; the
 * reference's own raster-order sum of Nv rounded terms drifts from that by ~1e-13 .. 1e-12 relative at 10^6
 * voxels -- tests bound the distance to the exact rational value and to the
 * reference's raster-order float64 sum) and in raster order -- bit-identical -- in voxel mode. */
int prad_calculate_ngtdm(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                         const int *angles, int Na, int Ng,
                         int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                         double *ngtdm);
int prad_calculate_ngtdm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                             const int *angles, int Na, int Ng,
                             int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             double *ngtdm, void *stream);
/* GLDM (cmatrices.c:568-648) and NGTDM (cmatrices.c:650-745) of one segment from ONE pass over the 26 / 8 neighbours
 * (both matrices look at the same neighbourhood; the case pipeline asks for both): device pointers, the bidirectional
 * angle set both classes use, gldm float64 [Ng][2 Na + 1], ngtdm float64 [Ng][3].  Falls back to the two separate calls
 * where the packed-byte kernel does not apply (alpha != 0, rows not a multiple of 4 voxels, Nd > 3).  Deferred mode:
 * enqueue only, as prad_calculate_gldm_dev. */
int prad_calculate_gldm_ngtdm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                                  int Na, int Ng, int alpha, double *gldm, double *ngtdm, void *stream);

/* ---- one large segment over several GPUs: plane-range accumulators of GLDM / NGTDM ------------------
 * (no reference analogue; SURVEY 8e "one exchange step": GLDM / NGTDM histograms are additive over centre
 * voxels.)  acc: DEVICE int64 [Ng][Na+1] for the centre voxels in planes z_lo <= z < z_hi of dim 0 of a 3-D
 * volume, neighbours taken from the whole volume (only planes z_lo-1 .. z_hi are read for distance 1):
 *   family 0, GLDM :  acc[g][k] = voxels of level g+1 with dependence k            (cmatrices.c:737-748)
 *   family 1, NGTDM:  acc[g][0] = voxels of level g+1;  acc[g][c] = sum over those with c valid neighbours
 *                     of |c*level - sum(neighbour levels)|                          (cmatrices.c:637-652)
 * The accumulators of disjoint plane ranges add up (exactly, they are integers) to those of the whole
 * volume; prad_neigh_finalize_dev turns the sum into the matrix prad_calculate_gldm / _ngtdm return
 * (gldm float64 [Ng][2*Na+1], ngtdm float64 [Ng][3]), bit-identical to the single-device call.
 * `alpha` is used by GLDM only.  PRAD_E_UNSUPPORTED when Nd != 3, Ng > 255 or the bins exceed 64 KiB of LDS;
 * PRAD_INDEX_ERROR when a masked level is outside 1..Ng. */
int prad_neigh_accumulate_dev(int family, const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                              const int *angles, int Na, int Ng, int alpha, int z_lo, int z_hi, long long *acc,
                              void *stream);
int prad_neigh_finalize_dev(int family, const long long *acc, int Ng, int Na, double *out, void *stream);

/* ---- GLSZM: calculate_glszm + fill_glszm (cmatrices.c:94-297) ----------------------------------- */
/* Phase 1: labels the zones of every kernel on the device and keeps the (level, size) list in the
 * library's per-thread workspace.  Returns the largest zone size over all Nvox kernels (>= 0), or
 * PRAD_E_* (< 0); *nzones (optional) receives the total number of zones.  `Ns` only sizes scratch in the
 * reference (_cmatrices.c:298-322) but its exhaustion is observable: a kernel with >= 2*Ns zones, or (voxel mode)
 * more than Ns masked voxels, makes the reference fail -- reproduced here as PRAD_E_INDEX (so an empty mask with
 * Ns = 0 raises IndexError exactly like the reference). */
int prad_calculate_glszm(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                         const int *angles, int Na, int Ng, int Ns,
                         int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                         long long *nzones);
int prad_calculate_glszm_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                             const int *angles, int Na, int Ng, int Ns,
                             int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             long long *nzones, void *stream);
/* Phase 2: histograms the zone list of the preceding phase-1 call into glszm float64
 * [Nvox][Ng][maxRegion] (host pointer for prad_fill_glszm, device pointer for _dev).
 * Returns PRAD_OK, or PRAD_INDEX_ERROR when a zone has level <= 0 or falls outside Ng x maxRegion
 * (cmatrices.c:290-291). */
int prad_fill_glszm(double *glszm, int Nvox, int Ng, int maxRegion);
int prad_fill_glszm_dev(double *glszm, int Nvox, int Ng, int maxRegion, void *stream);
/* Optional: copy the zone list of kernel v to the host in the reference's tempData format
 * (level,size pairs in raster order of each zone's first voxel, terminated by -1; cmatrices.c:255-276).
 * tempData must hold 2*nzones_v+1 ints; returns nzones_v or PRAD_E_*. */
long long prad_glszm_zones(int v, int *tempData, long long capacity_pairs);
/* Optional, segment mode: the matrix with its empty size columns already removed -- what glszm.py:118-131 keeps
 * (`jvector`, `np.delete(P_glszm, emptyZoneSizes, 2)`).  The reference layout [Ng][maxRegion] is almost all
 * zeros once zones are large (maxRegion in the millions for a smooth 256^3 volume, a few thousand distinct sizes).
 * prad_glszm_sizes: the distinct zone sizes of the preceding phase-1 call, ascending, into the HOST array `sizes`
 *   (capacity entries; at most min(maxRegion, sqrt(2 Nvoxels) + 1) are needed); returns their number k or PRAD_E_*.
 * prad_fill_glszm_compact_dev: glszm float64 [Ng][k] (DEVICE pointer), column c counting the zones of size
 *   sizes[c]; PRAD_OK / PRAD_INDEX_ERROR as prad_fill_glszm. */
int prad_glszm_sizes(int *sizes, int capacity);
int prad_fill_glszm_compact_dev(double *glszm, int Ng, int nsizes, void *stream);

/* ---- fused voxel-based GLCM feature maps (no reference analogue at this boundary) ---------------------------
 * For every centre voxel the GLCM of its kernel window is built and reduced to the requested features on the
 * device; the P[Nvox][Ng][Ng][Na] intermediate of the reference (glcm.py:145, base.py:200-245) is never
 * materialised.  Semantics follow glcm.py:149-887 with weightingNorm = None: per-angle normalisation, nanmean over
 * the angles that are non-empty for the voxel.  feature_ids: 0 Autocorrelation, 1 JointAverage, 2 ClusterProminence,
 * 3 ClusterShade, 4 ClusterTendency, 5 Contrast, 6 Correlation, 7 DifferenceAverage, 8 DifferenceEntropy,
 * 9 DifferenceVariance, 10 JointEnergy, 11 JointEntropy, 12 Imc1, 13 Imc2, 14 Idm, 15 Idmn, 16 Id, 17 Idn,
 * 18 InverseVariance, 19 MaximumProbability, 20 SumAverage, 21 SumEntropy, 22 SumSquares (MCC: prad_voxel_glcm_mcc).
 *   out         float64 [nfeat][Nvox]
 *   empty_mask  uint32 [Nvox] (optional): bit a set <=> angle a has no voxel pair in kernel v
 *   any_nonempty uint32 [1] (optional): OR of the non-empty angle bits over all kernels (JointAverage, which the
 *               reference averages with a plain mean, is NaN for kernels whose empty angles are not empty
 *               everywhere -- glcm.py:292; the Python layer applies that rule)
 * Requirements: Nd <= 3, Ng <= 64, Na <= 32, masked levels in [1, Ng]; otherwise PRAD_E_UNSUPPORTED and the
 * caller uses prad_calculate_glcm + host features.  All pointers of the _dev variant are device pointers. */
int prad_voxel_glcm_features(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles,
                             int Na, int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim,
                             int symmetric, const int *feature_ids, int nfeat, double *out, uint32_t *empty_mask,
                             uint32_t *any_nonempty);
int prad_voxel_glcm_features_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                                 const int *angles, int Na, int Ng, int Nvox, const int *voxels, int kernelRadius,
                                 int force2Ddim, int symmetric, const int *feature_ids, int nfeat, double *out,
                                 uint32_t *empty_mask, uint32_t *any_nonempty, void *stream);

/* MCC (glcm.py:665-707), the one GLCM feature that is an eigenvalue problem: sqrt of the second largest eigenvalue of
 * Q(i,j) = sum_k p(i,k) p(j,k) / (px(i) py(k) + eps), per angle, mean over the non-empty angles (np.nanmean).  One wave
 * per matrix runs a cyclic Jacobi iteration on the symmetric matrix A A^T that is similar to Q, restricted to the grey
 * levels that occur (at most 64 of them; more -> PRAD_E_UNSUPPORTED and the caller's host route).
 *   prad_voxel_glcm_mcc[_dev]  out float64 [Nvox]: per kernel (NaN when no angle has a voxel pair); arguments as for
 *                              prad_voxel_glcm_features
 *   prad_glcm_mcc_dev          glcm: DEVICE float64 raw counts [Ng][Ng][Na] (reference layout); out: HOST float64 [Na],
 *                              NaN for an angle without pairs
 * The reference's "a matrix of a single grey level -> 1" rule (glcm.py:702-703) depends on the grey levels of the whole
 * ROI and stays with the caller. */
int prad_voxel_glcm_mcc(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                        int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, int symmetric, double *out);
int prad_voxel_glcm_mcc_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                            int Ng, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, int symmetric,
                            double *out, void *stream);
int prad_glcm_mcc_dev(const double *glcm, int Ng, int Na, int symmetric, double *out, void *stream);

/* Fused voxel-based feature maps of the other four texture classes (same idea, same window rule as above).
 * family: 1 = GLDM (alpha = gldm_a; angles bidirectional), 2 = NGTDM (bidirectional), 3 = GLRLM (unidirectional, mean
 * over the non-empty angles as np.nanmean does), 4 = GLSZM (bidirectional).  feature_ids (HOST) follow the lists
 * VOXEL_*_FEATURES of pyradiomics_amd/cmatrices.py; out is DEVICE float64 [nfeat][Nvox]; image / mask / voxels are
 * DEVICE pointers.  PRAD_E_UNSUPPORTED (Nd > 3, Ng > 255, > 32 angles, > 512 voxels per kernel, levels outside
 * [1, Ng]) means: build the matrices with prad_calculate_* and evaluate the formulas on the host. */
int prad_voxel_texture_features_dev(int family, const int32_t *image, const uint8_t *mask, const int *size, int Nd,
                                    const int *angles, int Na, int Ng, int alpha, int Nvox, const int *voxels,
                                    int kernelRadius, int force2Ddim, const int *feature_ids, int nfeat, double *out,
                                    void *stream);

/* ---- segment-mode feature formulas on the device matrices (no reference analogue: numpy in glcm.py:208-887,
 * glrlm.py:174-523, glszm.py:108-434, gldm.py:103-430) ------------------------------------------------------------
 * prad_glcm_features_dev: glcm = DEVICE float64 [Ng][Ng][Na] raw counts as prad_calculate_glcm[_glrlm]_dev leave
 *   them; per angle the matrix is symmetrised (symmetric != 0), normalised, and reduced to the 23 features that are
 *   plain sums (ids = VOXEL_GLCM_FEATURES order; MCC excluded).  out: HOST float64 [Na][23]; empty: HOST int [Na],
 *   1 where the angle holds no pair (its row of `out` is NaN; glcm.py:186-198 drops such angles).
 * prad_zone_matrix_features_dev: P(i, j, a) = P[i*stride_i + j*stride_j + a*stride_a] DEVICE float64 counts with level
 *   value i + 1 and size value jvals[j] (HOST float64 [Nj]: run lengths / zone sizes / dependence counts + 1; NULL: j + 1);
 *   out: HOST float64 [Na][16] in the shared numbering of prad_voxel_texture_features_dev; empty as above. */
int prad_glcm_features_dev(const double *glcm, int Ng, int Na, int symmetric, double *out, int *empty, void *stream);
int prad_zone_matrix_features_dev(const double *P, int Ni, int Nj, int Na, long long stride_i, long long stride_j,
                                  long long stride_a, const double *jvals, double *out, int *empty, void *stream);
/* prad_glszm_features_dev: the GLSZM of a segment (what prad_calculate_glszm_dev + prad_glszm_sizes +
 *   prad_fill_glszm_compact_dev build, cmatrices.c:294-443 / glszm.py:108-131) AND its 16 features, without a host round
 *   trip in between: zones, then the distinct zone sizes are ranked on the device, the compact matrix filled, the
 *   formulas evaluated.  image / mask DEVICE, angles HOST int [Na][Nd] (the bidirectional distance-1 set), Ns = ROI
 *   voxels.  out: float64 [17] -- the 16 features in the shared numbering, then a verdict: 0 = fine, bit 1 = the zone
 *   list would overflow the reference's Ns-sized scratch (cmatrices.c:366-373), other bits = the device-side ranking
 *   declined (levels outside 1..Ng, more than 4096 zones of 8192+ voxels); empty: int [1], 1 = no zone.
 *   Needs the packed-byte tile kernels (Nd <= 3, full 26- / 8-neighbourhood, Ng <= 255): PRAD_E_UNSUPPORTED otherwise,
 *   before anything is launched.  In deferred mode with out / empty inside the result arena: enqueue only, the caller
 *   reads the verdict from out[16] after synchronising; otherwise synchronous, verdict bit 1 -> PRAD_E_INDEX, any other
 *   bit -> PRAD_E_UNSUPPORTED (take the three-call route). */
int prad_glszm_features_dev(const int32_t *image, const uint8_t *mask, const int *size, int Nd, const int *angles, int Na,
                            int Ng, int Ns, double *out, int *empty, void *stream);
/* NGTDM (ngtdm.py:133-287): P = DEVICE float64 [Ng][3] as prad_calculate_ngtdm_dev leaves it;
 * out: HOST float64 [5] = Coarseness, Contrast, Busyness, Complexity, Strength. */
int prad_ngtdm_features_dev(const double *P, int Ng, double *out, void *stream);

/* ---- on-device discretisation (radiomics/imageoperations.py:67-174; device pointers only) ----------------------
 * dtype: 0 = float32, 1 = float64, 2 = int32, 3 = int16 image.
 * prad_roi_minmax_dev: minmax[0..1] (HOST doubles) = min / max of image over mask != 0; PRAD_E_ARG if the ROI is empty.
 * prad_digitize_dev:   levels[i] = number of edges <= image[i] (np.digitize) where mask != 0, else 0;
 *                      `edges` is a HOST array of nedges ascending float64 values (getBinEdges output);
 *                      *max_level (HOST int) receives the largest level written (= Ng of base.py:121-124).
 * prad_level_counts_dev: counts[g] (HOST int64, Ng + 1 entries) = number of ROI voxels with level g for g in 1..Ng
 *                      (which levels are present = `grayLevels` of base.py:120-122, their sum = Ns of glszm.py:84);
 *                      counts[0] = ROI voxels whose level is outside 1..Ng. */
int prad_roi_minmax_dev(const void *image, int dtype, const uint8_t *mask, long long n, double *minmax, void *stream);
int prad_digitize_dev(const void *image, int dtype, const uint8_t *mask, long long n, const double *edges, int nedges,
                      int32_t *levels, int *max_level, void *stream);
int prad_level_counts_dev(const int32_t *levels, const uint8_t *mask, long long n, int Ng, long long *counts,
                          void *stream);
/* prad_digitize_counts_dev: prad_digitize_dev and the level census in ONE pass over the image (base.py:119-125 needs
 *                      both for every derived image): counts (HOST int64, nedges + 1 entries, may be NULL) receives the
 *                      number of ROI voxels per level 0..nedges.  Any number of edges (beyond 7 680 the edge list is
 *                      searched in global memory instead of LDS). */
int prad_digitize_counts_dev(const void *image, int dtype, const uint8_t *mask, long long n, const double *edges,
                             int nedges, int32_t *levels, int *max_level, long long *counts, void *stream);
/* binCount discretisation with ONE host synchronisation: ROI min / max, the np.histogram edges built on the device
 * (np.linspace's arithmetic: i * step + min with two roundings, last edge = max, then + 1 as in imageoperations.py:122-126;
 * in float32 for a float32 image, as numpy does it, in float64 otherwise), levels, largest level and the level census in
 * one queue.  dtype as for prad_digitize_dev.  minmax: HOST double [2]; edges:
 * HOST double [binCount + 1] or NULL (what the device used: bit-identical to getBinEdges on (min, max));
 * counts: HOST int64 [binCount + 2] as prad_digitize_counts_dev, or NULL.  PRAD_E_UNSUPPORTED for a constant or
 * non-finite ROI (np.histogram widens the range then): the two-call route with host-built edges serves those. */
int prad_bincount_dev(const void *image, int dtype, const uint8_t *mask, long long n, int binCount, int32_t *levels,
                      double *minmax, double *edges, int *max_level, long long *counts, void *stream);
/* prad_bincount_dev in two halves, for callers that have other host work between "queued" and "needed" (the case pipeline
 * bins derived image i + 1 while the host still collects image i: imageoperations.py:67-174 + base.py:119-125 are one
 * synchronous step per feature class in the reference).  prad_bincount_enqueue_dev queues the same kernels and the copy of
 * their 8 x (2 binCount + 7) result bytes into a pinned block and returns a ticket (up to four in flight per host thread);
 * prad_bincount_wait waits for THAT work only (an event behind the copy) and fills the host outputs of prad_bincount_dev,
 * with its return codes.  A ticket is waited for exactly once; `levels` must stay allocated until then. */
int prad_bincount_enqueue_dev(const void *image, int dtype, const uint8_t *mask, long long n, int binCount, int32_t *levels,
                              int *ticket, void *stream);
int prad_bincount_wait(int ticket, double *minmax, double *edges, int *max_level, long long *counts);

/* ---- first-order statistics of the ROI intensities (radiomics/firstorder.py:33-474; device pointers) -----------
 * Segment mode.  The reference computes these with numpy on image[mask] (firstorder.py:96-101); here the ROI is
 * compacted, sorted and reduced on the device and `out` (HOST, PRAD_FO_COUNT doubles) receives the statistics every
 * feature of the class derives from:
 *   Np; Energy = sum (x + voxelArrayShift)^2 (:151); Minimum / Maximum (:207,:244); the 10 / 25 / 75 / 90-th
 *   percentiles and the median with numpy's linear interpolation (:218-231,:268,:282); Mean (:256); mean absolute
 *   deviation (:308); robust mean absolute deviation over p10 <= x <= p90 (:323-342); central moments m2, m3, m4
 *   (:136-145, Variance / Skewness / Kurtosis :381-446).
 * dtype as for prad_digitize_dev.  Entropy / Uniformity come from prad_level_counts_dev on the discretised image. */
enum { PRAD_FO_NP = 0, PRAD_FO_ENERGY, PRAD_FO_MINIMUM, PRAD_FO_P10, PRAD_FO_P25, PRAD_FO_MEDIAN, PRAD_FO_P75, PRAD_FO_P90,
       PRAD_FO_MAXIMUM, PRAD_FO_MEAN, PRAD_FO_MAD, PRAD_FO_RMAD, PRAD_FO_M2, PRAD_FO_M3, PRAD_FO_M4, PRAD_FO_COUNT };
int prad_firstorder_dev(const void *image, int dtype, const uint8_t *mask, long long n, double voxelArrayShift,
                        double *out, void *stream);
/* The same statistics without a host round trip between the passes (prad_firstorder_dev has five): what the host does
 * in between -- adding up block partials, placing the quantiles, finding the histogram bins that hold their ranks,
 * interpolating -- runs in single-workgroup kernels on a device record, the same operations in the same order (the same
 * bits).  float32 / float64 images with roi_count >= 2^20 ROI voxels (the caller knows the count from the level census);
 * PRAD_E_UNSUPPORTED otherwise, before anything is launched.  out: 16 doubles -- the PRAD_FO_COUNT statistics, then a
 * verdict (0 = fine; 1: the ROI holds a different number of voxels, 2: constant or non-finite ROI, 8: the selected
 * histogram bins hold more than the gather capacity: half of the ROI, at least 2^18 voxels).  In deferred mode with `out` inside the result arena: enqueue only;
 * otherwise synchronous, a non-zero verdict returns PRAD_E_UNSUPPORTED (call prad_firstorder_dev). */
int prad_firstorder_queue_dev(const void *image, int dtype, const uint8_t *mask, long long n, long long roi_count,
                              double voxelArrayShift, double *out, void *stream);
/* Voxel mode (firstorder.py:37-118,104-118): for each of the Nvox centres (`voxels` DEVICE int32 [Nd][Nvox]) the
 * window centre + {offsets of infinity-norm <= kernelRadius} -- per dimension limited to |offset| < bbsize[d] (HOST,
 * the `boundingBoxSize` of firstorder.py:45-58; NULL = no limit) and 0 in the force2D dimension -- is reduced over
 * its voxels with mask != 0 (the reference's NaN padding / NaN outside the ROI).  levels (discretised image, may
 * be NULL) feeds Entropy / Uniformity.  feature_ids (HOST) index the feature list in the order of
 * pyradiomics_amd.firstorder.FEATURES; out is DEVICE float64 [nfeat][Nvox]. */
int prad_voxel_firstorder_dev(const void *image, int dtype, const uint8_t *mask, const int32_t *levels, const int *size,
                              int Nd, int Nvox, const int *voxels, int kernelRadius, int force2Ddim, const int *bbsize,
                              double voxelArrayShift, double voxelVolume, const int *feature_ids, int nfeat, double *out,
                              void *stream);

/* ---- resampling in front of the path (radiomics/imageoperations.py:448-612 on SimpleITK's ResampleImageFilter) ------
 * Output voxel o along (numpy) axis d samples the input at the continuous index start[d] + o * step[d]; the output
 * grid shares the input's axes (the reference derives start / step / newsize from the ROI bounding box, padDistance
 * and resampledPixelSpacing, imageoperations.py:548-578).  interpolator: 0 nearest neighbour (round half up), 1 linear,
 * 3 cubic B-spline exactly as ITK evaluates it (recursive prefilter with mirror boundaries, mirrored 4^Nd support);
 * samples outside the buffer ([-0.5, N - 0.5) per axis) become 0; integer pixel types are clamped and truncated like
 * ResampleImageFilter::CastPixelWithBoundsChecking.  image / out: DEVICE pointers of the same dtype (codes as for
 * prad_digitize_dev); size / start / step / newsize: HOST arrays of Nd (<= 3) entries. */
int prad_resample_dev(const void *image, int dtype, const int *size, int Nd, const double *start, const double *step,
                      const int *newsize, int interpolator, void *out, void *stream);

/* ---- filter stack in front of the matrices (radiomics/imageoperations.py:756-970) ---------------------------
 * The arithmetic of both filters lives in third-party wheels (PyWavelets, SimpleITK/ITK) that are not part of
 * the reference tree; these entry points implement their published algorithms (see oracle/filters_oracle.py):
 * parity is pinned by the 198 brain1 values the reference recorded in notebooks/helloFeatureClass.ipynb
 * (tests/test_notebook_pin.py).
 *
 * prad_swt_level1: one level of the undecimated (stationary) wavelet transform with periodisation along `axes`
 * (in that order), i.e. pywt.swtn(data, wavelet, level=1, start_level=0, axes) (imageoperations.py:928,935).
 *   in   float64 [size[0]]..[size[Nd-1]], every transformed axis of even length
 *   out  float64 [2^naxes][...]: sub-bands in PyWavelets' key order ('a' = dec_lo before 'd' = dec_hi, the first
 *        axis of `axes` is the most significant letter): aaa, aad, ada, add, daa, dad, dda, ddd for 3 axes.
 * prad_log: ITK LaplacianRecursiveGaussianImageFilter (imageoperations.py:824-830): float32 in / out; per dimension the
 *   second-order pass along it first, then the zero-order passes along the other dimensions in increasing ITK direction
 *   (the order that reproduces the float32 order statistics the reference recorded for brain1 bit for bit),
 *   `spacing` per ARRAY axis (i.e. SimpleITK spacing reversed), sigma in the units of spacing,
 *   normalize != 0 multiplies by sigma^2 (NormalizeAcrossScale).  Every axis needs >= 4 samples. */
int prad_swt_level1(const double *in, const int *size, int Nd, const double *dec_lo, const double *dec_hi, int flen,
                    const int *axes, int naxes, double *out);
int prad_swt_level1_dev(const double *in, const int *size, int Nd, const double *dec_lo, const double *dec_hi,
                        int flen, const int *axes, int naxes, double *out, void *stream);
/* The same with the image in its own element type (dtype codes as prad_roi_minmax_dev: 0 float32, 1 float64, 2 int32,
 * 3 int16): the widening to float64 is folded into the fused 3-D kernel.  (The reference copies and pads the array,
 * imageoperations.py:914-919, and pywt.swtn widens integer images to float64; it keeps FLOAT32 images in float32 arithmetic
 * and returns float32 sub-bands -- here every input type is transformed in float64: for float32 images the sub-bands carry
 * more precision than the reference's, a stated deviation)
 * PRAD_E_UNSUPPORTED when the call is not a 3-D transform over axes (2, 1, 0) with 2 / 4 / 6 taps: convert and use
 * prad_swt_level1_dev. */
int prad_swt_level1_any_dev(const void *in, int dtype, const int *size, int Nd, const double *dec_lo,
                            const double *dec_hi, int flen, const int *axes, int naxes, double *out, void *stream);
int prad_log(const float *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
             float *out);
int prad_log_dev(const float *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
                 float *out, void *stream);
/* nsig (1..8) sigmas of ONE input in the same launches (the reference loops over its sigma list, imageoperations.py:
 * 806-836, one filter run each): a pass of one sigma is bound by the latency of its serial recursion, the waves of the
 * other sigmas fill the gaps.  outs: HOST array of nsig DEVICE pointers; the same arithmetic per sigma as prad_log_dev. */
int prad_log_multi_dev(const float *in, const int *size, int Nd, const double *spacing, const double *sigmas, int nsig,
                       int normalize, float *const *outs, void *stream);
/* The same filter on float64 images (e.g. after `normalize: true`, whose output is float64): the result is float64 like the
 * input (sitk.LaplacianRecursiveGaussianImageFilter keeps the pixel type, imageoperations.py:824-830) and is computed as
 * ITK computes it: the derivative pass of every term reads the float64 input, the images between the passes and the
 * cumulative image are float (itkLaplacianRecursiveGaussianImageFilter.h: InternalRealType = float), the sum is widened at
 * the end.  (Rounds 3-4 kept float64 images between the passes.) */
int prad_log_f64(const double *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
                 double *out);
int prad_log_dev_f64(const double *in, const int *size, int Nd, const double *spacing, double sigma, int normalize,
                     double *out, void *stream);
int prad_log_multi_dev_f64(const double *in, const int *size, int Nd, const double *spacing, const double *sigmas,
                           int nsig, int normalize, double *const *outs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PYRADIOMICS_AMD_H */
