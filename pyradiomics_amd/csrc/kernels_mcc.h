// kernels_mcc.h -- Maximal Correlation Coefficient on the device (glcm.py:665-707): the one GLCM feature that is an
// eigenvalue problem.  MCC = sqrt(second largest eigenvalue of Q), Q(i,j) = sum_k p(i,k) p(j,k) / (px(i) py(k) + eps).
// Q is similar to M = A A^T with A(i,k) = p(i,k) / sqrt(px(i) py(k) + eps), a symmetric positive semi-definite matrix, so a
// cyclic Jacobi iteration gives its whole spectrum to machine precision.  ONE WAVE per matrix:
//   1. marginals px, py; the n grey levels that occur at all (rows / columns of zeros only add zero eigenvalues)
//   2. M restricted to those levels, n x n float64 in LDS (n <= 64; more -> the caller's host route)
//   3. parallel cyclic Jacobi: a round-robin tournament pairs the n indices into n/2 disjoint (p, q) per step, the n/2
//      rotations of a step are applied together (all row pairs, then all column pairs); converged when the off-diagonal
//      mass is below 1e-30 of the diagonal's
//   4. second largest diagonal entry.
// Segment mode: one wave per angle on the raw count matrix [Ng][Ng][Na].  Voxel mode: one wave per kernel centre
// builds the window's counts per angle in LDS (as kernels_voxel.h pass A) and runs the same routine.
#pragma once
#include "prad_runtime.h"
#include "kernels_voxel.h"

namespace prad {

#define PRAD_MCC_NMAX 64

struct MccScratch {
  double *M;    // [nmax * nmax]
  double *A;    // [nmax * (nmax + 1)]  A(i, k) = p(i, k) / sqrt(px(i) py(k) + eps) on the occurring levels (odd row pitch)
  double *px;   // [Ng]
  double *py;   // [Ng]
  double *cs;   // [nmax] (c, s) of the step's rotations
  int *idx;     // [nmax] grey levels that occur
  int *pq;      // [nmax] (p, q) of the step's rotations
};
// (the iteration pads an odd number of levels with one zero row / column: arrays are sized for nmax rounded up to even)
__host__ __device__ inline size_t mcc_scratch_bytes(int Ng, int nmax) {
  nmax += nmax & 1;
  return sizeof(double) * ((size_t)nmax * nmax + (size_t)nmax * (nmax + 1) + 2 * (size_t)Ng + nmax) + sizeof(int) * 2 * (size_t)nmax;
}
__device__ __forceinline__ MccScratch mcc_scratch(void *base, int Ng, int nmax) {
  MccScratch S;
  nmax += nmax & 1;
  S.M = (double *)base;
  S.A = S.M + (size_t)nmax * nmax;
  S.px = S.A + (size_t)nmax * (nmax + 1);
  S.py = S.px + Ng;
  S.cs = S.py + Ng;
  S.idx = (int *)(S.cs + nmax);
  S.pq = S.idx + nmax;
  return S;
}
__device__ __forceinline__ void mcc_wave_sync() {   // LDS written by one lane is read by another lane of the same wave
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// C(i, j): count (already symmetrised if wanted) of the ordered level pair, tot: their sum (> 0).
// Returns sqrt(lambda_2) (0 when fewer than two levels occur); *too_many is set when more than nmax levels occur.
template <class Acc>
__device__ double wave_mcc(Acc C, int Ng, double tot, MccScratch S, int nmax, int *too_many) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const double eps = 2.220446049250313e-16;
  for (int i = lane; i < Ng; i += 64) {
    double r = 0, c = 0;
    for (int j = 0; j < Ng; j++) {
      r += C(i, j) / tot;
      c += C(j, i) / tot;
    }
    S.px[i] = r;
    S.py[i] = c;
  }
  mcc_wave_sync();
  int n = 0;
  for (int base = 0; base < Ng; base += 64) {
    const int i = base + lane;
    const bool pres = i < Ng && (S.px[i] > 0 || S.py[i] > 0);
    const unsigned long long m = __ballot(pres);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (pres && n + before < nmax) S.idx[n + before] = i;
    n += __popcll(m);
  }
  if (n > nmax) {
    if (lane == 0) *too_many = 1;
    return __builtin_nan("");
  }
  if (n < 2) return 0.0;
  mcc_wave_sync();
  const int np = n + (n & 1);
  // A once (n^2 divisions and square roots instead of n^3: the products below are the same values in the same order)
  const int ap = np + 1;
  for (int e = lane; e < n * n; e += 64) {
    const int a = e / n, k = e - a * n;
    const int ia = S.idx[a], ik = S.idx[k];
    const double ca = C(ia, ik);
    S.A[a * ap + k] = ca != 0 ? (ca / tot) / sqrt(S.px[ia] * S.py[ik] + eps) : 0.0;
  }
  mcc_wave_sync();
  for (int e = lane; e < np * np; e += 64) {
    const int a = e / np, b = e - a * np;
    double v = 0;
    if (a < n && b < n) {
      const double *ra = S.A + a * ap, *rb = S.A + b * ap;
      for (int k = 0; k < n; k++) {
        const double x = ra[k], y = rb[k];
        if (x != 0 && y != 0) v += x * y;
      }
    }
    S.M[e] = v;
  }
  mcc_wave_sync();
  const int m1 = np - 1, half = np / 2;
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0, dia = 0;
    for (int e = lane; e < np * np; e += 64) {
      const int a = e / np, b = e - a * np;
      const double v = S.M[e];
      if (a == b) dia += v * v; else off += v * v;
    }
    off = wave_sum_f64(off);
    dia = wave_sum_f64(dia);
    if (off <= 1e-30 * dia) break;
    for (int step = 0; step < m1; step++) {
      if (lane < half) {
        int p, q;
        if (lane == 0) { p = step; q = m1; }
        else { p = (step + lane) % m1; q = (step - lane + m1) % m1; }
        const double app = S.M[p * np + p], aqq = S.M[q * np + q], apq = S.M[p * np + q];
        double c = 1.0, s = 0.0;
        if (apq != 0.0) {
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
        }
        S.pq[2 * lane] = p; S.pq[2 * lane + 1] = q;
        S.cs[2 * lane] = c; S.cs[2 * lane + 1] = s;
      }
      mcc_wave_sync();
      for (int w = lane; w < half * np; w += 64) {      // J^T M: rows p, q of every pair
        const int t = w / np, r = w - t * np;
        const int p = S.pq[2 * t], q = S.pq[2 * t + 1];
        const double c = S.cs[2 * t], s = S.cs[2 * t + 1];
        const double mp = S.M[p * np + r], mq = S.M[q * np + r];
        S.M[p * np + r] = c * mp - s * mq;
        S.M[q * np + r] = s * mp + c * mq;
      }
      mcc_wave_sync();
      for (int w = lane; w < half * np; w += 64) {      // (J^T M) J: columns p, q of every pair
        const int t = w / np, r = w - t * np;
        const int p = S.pq[2 * t], q = S.pq[2 * t + 1];
        const double c = S.cs[2 * t], s = S.cs[2 * t + 1];
        const double mp = S.M[r * np + p], mq = S.M[r * np + q];
        S.M[r * np + p] = c * mp - s * mq;
        S.M[r * np + q] = s * mp + c * mq;
      }
      mcc_wave_sync();
    }
  }
  // second largest eigenvalue: the largest one with its first position masked out
  double d = lane < n ? S.M[lane * np + lane] : -1.0;
  const double top = wave_max_f64(d);
  const unsigned long long at = __ballot(d == top);
  const int first = (int)(__ffsll((long long)at) - 1);
  if (lane == first) d = -1.0;
  const double second = wave_max_f64(d);
  return sqrt(fmax(second, 0.0));
}

// The same routine for a whole workgroup (segment mode: one matrix per angle, 13 matrices in all -- a single wave per
// matrix spent ~1 ms in the latency chain of ~250 Jacobi steps, each two passes of 8 dependent LDS round trips per lane;
// with PRAD_MCC_BT threads a pass is 2 round trips).  Same arithmetic, same order of the rotations: same bits.
#ifndef PRAD_MCC_BT
#define PRAD_MCC_BT 256
#endif
__device__ __forceinline__ double block_sum_f64(double v, double *red) {    // every thread gets the sum
  v = wave_sum_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < PRAD_MCC_BT / 64; w++) t += red[w];
  return t;
}
template <class Acc>
__device__ double block_mcc(Acc C, int Ng, double tot, MccScratch S, int nmax, int *too_many, double *red, int *shn) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x, NT = PRAD_MCC_BT;
  const double eps = 2.220446049250313e-16;
  for (int i = tid; i < Ng; i += NT) {
    double r = 0, c = 0;
    for (int j = 0; j < Ng; j++) {
      r += C(i, j) / tot;
      c += C(j, i) / tot;
    }
    S.px[i] = r;
    S.py[i] = c;
  }
  __syncthreads();
  if (tid < 64) {          // the occurring levels, in order (one wave: ballot ranks)
    int n = 0;
    for (int base = 0; base < Ng; base += 64) {
      const int i = base + tid;
      const bool pres = i < Ng && (S.px[i] > 0 || S.py[i] > 0);
      const unsigned long long m = __ballot(pres);
      const int before = __popcll(m & ((1ull << tid) - 1ull));
      if (pres && n + before < nmax) S.idx[n + before] = i;
      n += __popcll(m);
    }
    if (tid == 0) *shn = n;
  }
  __syncthreads();
  const int n = *shn;
  if (n > nmax) {
    if (tid == 0) *too_many = 1;
    return __builtin_nan("");
  }
  if (n < 2) return 0.0;
  const int np = n + (n & 1), ap = np + 1;
  for (int e = tid; e < n * n; e += NT) {
    const int a = e / n, k = e - a * n;
    const int ia = S.idx[a], ik = S.idx[k];
    const double ca = C(ia, ik);
    S.A[a * ap + k] = ca != 0 ? (ca / tot) / sqrt(S.px[ia] * S.py[ik] + eps) : 0.0;
  }
  __syncthreads();
  for (int e = tid; e < np * np; e += NT) {
    const int a = e / np, b = e - a * np;
    double v = 0;
    if (a < n && b < n) {
      const double *ra = S.A + a * ap, *rb = S.A + b * ap;
      for (int k = 0; k < n; k++) {
        const double x = ra[k], y = rb[k];
        if (x != 0 && y != 0) v += x * y;
      }
    }
    S.M[e] = v;
  }
  __syncthreads();
  // Only the second largest eigenvalue is wanted: Householder reduction of M to tridiagonal form (n - 2 reflections, each
  // a matrix-vector product and a rank-2 update over the workgroup), then its position by Sturm counts -- 64 shifts per
  // round, one per lane, every round narrows the bracket 65-fold (absolute accuracy ~ n eps ||M||, as LAPACK's on the
  // reference side).  ~40 us per matrix at n = 32; the cyclic Jacobi iteration this replaces (kept in wave_mcc for the
  // voxel maps, whose matrices are tiny) needed ~280 rotation steps of 4 barriers each: 0.6 ms.
  double *dg = S.px, *sd = S.py;              // diagonal / sub-diagonal of T (the marginals are not needed any more)
  double *u = S.A, *pv = S.A + nmax + 1;      // scratch of the bracket search, M u / h (A is free: nmax (nmax + 1) >= 2 (nmax + 1))
  // (every wave works out the reflector's scalars for itself -- same instructions, same bits -- and reads the reflector
  // from row i of M, which no later update touches: two barriers per reflection instead of five)
  const int lane = tid & 63;
  for (int i = n - 1; i >= 1; i--) {
    const int l = i - 1;                      // the reflector annihilates M[i][0 .. l-1]
    const double *rowi = S.M + i * np;
    double v = 0;
    if (l >= 1) {
      for (int k = lane; k <= l; k += 64) v += rowi[k] * rowi[k];
      v = wave_sum_f64(v);
    }
    const double f = rowi[l];
    if (l == 0 || v == 0.0 || v == f * f) {          // nothing to annihilate
      if (tid == 0) sd[i] = f;
      continue;
    }
    const double g = f >= 0 ? -sqrt(v) : sqrt(v), h = v - f * g, ul = f - g;
    if (tid == 0) sd[i] = g;
    auto uk = [&](int k) -> double { return k == l ? ul : rowi[k]; };
    {
      // M u over the whole workgroup: LPR lanes per row, every lane a strided part of the row, shuffle tree (a lane per row
      // was a chain of l + 1 dependent LDS reads with 32 of the 256 lanes busy: a quarter of the kernel)
      const int LPR = l < 32 ? 8 : 4, j = tid / LPR, part = tid % LPR;
      double acc = 0;
      if (j <= l)
        for (int k = part; k <= l; k += LPR) acc += S.M[j * np + k] * uk(k);
      for (int o = 1; o < LPR; o <<= 1) acc += __shfl_xor(acc, o);
      if (j <= l && part == 0) pv[j] = acc / h;
    }
    __syncthreads();
    double w = 0;
    for (int k = lane; k <= l; k += 64) w += uk(k) * pv[k];
    const double K = wave_sum_f64(w) / (h + h);
    for (int e = tid; e < (l + 1) * (l + 1); e += NT) {
      const int j = e / (l + 1), k = e - j * (l + 1);
      const double uj = uk(j), ukk = uk(k);
      const double qj = pv[j] - K * uj, qk = pv[k] - K * ukk;
      S.M[j * np + k] -= uj * qk + qj * ukk;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += NT) dg[i] = S.M[i * np + i];
  if (tid == 0) sd[0] = 0.0;
  __syncthreads();
  // Gershgorin bracket, then multisection on N(x) = #{eigenvalues < x}: lambda_(n-1) = sup{x : N(x) <= n - 2}; one shift per
  // thread, every round narrows the bracket (NT + 1)-fold.  The Sturm recurrence is a chain of n divisions: v_rcp_f64 and
  // one Newton step instead of the IEEE division (a count can only differ where q is within an ulp of zero, i.e. the
  // eigenvalue moves by its own rounding error)
  double lo = 1e300, hi = -1e300;
  for (int i = tid; i < n; i += NT) {
    const double r = fabs(sd[i]) + (i + 1 < n ? fabs(sd[i + 1]) : 0.0);
    lo = fmin(lo, dg[i] - r);
    hi = fmax(hi, dg[i] + r);
  }
  lo = -wave_max_f64(-lo);
  hi = wave_max_f64(hi);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = lo; u[tid >> 6] = hi; }     // (u: free again)
  __syncthreads();
  for (int w = 0; w < NT / 64; w++) { lo = fmin(lo, red[w]); hi = fmax(hi, u[w]); }
  __syncthreads();
  for (int i = tid; i < n; i += NT) pv[i] = sd[i] * sd[i];
  __syncthreads();
  const double tiny = 1e-300;
  int *cntw = reinterpret_cast<int *>(red);
  for (int round = 0; round < 12 && hi > lo; round++) {
    const double step = (hi - lo) / (double)(NT + 1);
    const double x = lo + step * (double)(tid + 1);
    double q = dg[0] - x;
    int cnt = q < 0;
    for (int i = 1; i < n; i++) {
      if (q == 0.0) q = tiny;
      double r = __builtin_amdgcn_rcp(q);
      r = r * (2.0 - q * r);
      q = dg[i] - x - pv[i] * r;
      cnt += q < 0;
    }
    const unsigned long long below = __ballot(cnt <= n - 2);      // a prefix of the threads (N is monotone)
    __syncthreads();
    if ((tid & 63) == 0) cntw[tid >> 6] = __popcll(below);
    __syncthreads();
    int nb = 0;
    for (int w = 0; w < NT / 64; w++) nb += cntw[w];
    const double nlo = nb > 0 ? lo + step * (double)nb : lo, nhi = nb < NT ? lo + step * (double)(nb + 1) : hi;
    if (nlo == lo && nhi == hi) break;
    lo = nlo;
    hi = nhi;
  }
  const double second = 0.5 * (lo + hi);
  return sqrt(fmax(second, 0.0));
}

// segment mode: counts [Ng][Ng][Na] float64 (raw, reference layout) -> out[a] = MCC of angle a (NaN: no pair)
__global__ void __launch_bounds__(PRAD_MCC_BT) glcm_matrix_mcc_kernel(const double *__restrict__ counts, int Ng, int Na,
                                                                      int symmetric, int nmax, int staged,
                                                                      double *__restrict__ out, int *__restrict__ too_many) {
  extern __shared__ double mcc_lds[];
  __shared__ double red[PRAD_MCC_BT / 64];
  __shared__ int shn;
  const int a = blockIdx.x, tid = threadIdx.x;
  auto G = [&](int i, int j) -> double {
    const double v = counts[((size_t)i * Ng + j) * Na + a];
    return symmetric ? v + counts[((size_t)j * Ng + i) * Na + a] : v;
  };
  double v = __builtin_nan("");
  if (staged) {
    // the angle's (symmetrised) counts go to LDS once: the routine reads every entry ~Ng times, and in the [Ng][Ng][Na]
    // layout each read is a cache line of its own
    double *Cs = (double *)((char *)mcc_lds + ((mcc_scratch_bytes(Ng, nmax) + 15) & ~(size_t)15));
    double tot = 0;
    for (int e = tid; e < Ng * Ng; e += PRAD_MCC_BT) {
      const double c = G(e / Ng, e % Ng);
      Cs[e] = c;
      tot += c;
    }
    tot = block_sum_f64(tot, red);
    auto C = [&](int i, int j) -> double { return Cs[i * Ng + j]; };
    if (tot > 0) v = block_mcc(C, Ng, tot, mcc_scratch(mcc_lds, Ng, nmax), nmax, too_many, red, &shn);
  } else {
    double tot = 0;
    for (int e = tid; e < Ng * Ng; e += PRAD_MCC_BT) tot += G(e / Ng, e % Ng);
    tot = block_sum_f64(tot, red);
    if (tot > 0) v = block_mcc(G, Ng, tot, mcc_scratch(mcc_lds, Ng, nmax), nmax, too_many, red, &shn);
  }
  if (tid == 0) out[a] = v;
}

#define PRAD_MCC_WAVES 2
// voxel mode: out[v] = mean over the non-empty angles of the per-angle MCC of kernel v (NaN: no angle has a pair)
__global__ void __launch_bounds__(64 * PRAD_MCC_WAVES) voxel_glcm_mcc_kernel(
    const uint8_t *__restrict__ L, int Nz, int Ny, int Nx, VoxAngles A, int Ng, int nvox, const int *__restrict__ voxels,
    int vox_nd, int radius, int f2d3, int symmetric, int nmax, double *__restrict__ out, int *__restrict__ too_many,
    const int *__restrict__ flags) {
  extern __shared__ double mcc_lds[];
  if (flags[0]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wave = (mcc_scratch_bytes(Ng, nmax) + sizeof(u32) * (size_t)Ng * Ng + 15) & ~(size_t)15;
  char *base = (char *)mcc_lds + per_wave * wave;
  MccScratch S = mcc_scratch(base, Ng, nmax);
  u32 *tab = (u32 *)(base + mcc_scratch_bytes(Ng, nmax));
  for (int i = lane; i < Ng * Ng; i += 64) tab[i] = 0;
  mcc_wave_sync();
  const bool sym = symmetric != 0;
  const int nwaves = (int)(blockDim.x >> 6);   // 2, or 1 when two per-wave scratch areas exceed the LDS (Ng 63, 64)
  for (int v = blockIdx.x * nwaves + wave; v < nvox; v += gridDim.x * nwaves) {
    int c[3] = {0, 0, 0};
    for (int d = 0; d < vox_nd; d++) c[3 - vox_nd + d] = voxels[(long long)d * nvox + v];
    const int dims[3] = {Nz, Ny, Nx};
    int lo[3], ext[3];
    for (int d = 0; d < 3; d++) {
      if (d == f2d3 || d < 3 - vox_nd) { lo[d] = c[d]; ext[d] = 1; }
      else {
        lo[d] = max(c[d] - radius, 0);
        ext[d] = min(c[d] + radius, dims[d] - 1) - lo[d] + 1;
      }
    }
    const int W = ext[0] * ext[1] * ext[2];
    double acc = 0;
    int n_angles = 0;
    for (int a = 0; a < A.na; a++) {
      const int dz = A.o[a][0], dy = A.o[a][1], dx = A.o[a][2];
      int np = 0;
      for (int k = lane; k < W; k += 64) {
        const int kx = k % ext[2], kr = k / ext[2];
        const int ky = kr % ext[1], kz = kr / ext[1];
        const int qz = kz + dz, qy = ky + dy, qx = kx + dx;
        if ((unsigned)qz >= (unsigned)ext[0] || (unsigned)qy >= (unsigned)ext[1] || (unsigned)qx >= (unsigned)ext[2]) continue;
        const int li = L[((long long)(lo[0] + kz) * Ny + (lo[1] + ky)) * Nx + lo[2] + kx];
        if (!li) continue;
        const int lj = L[((long long)(lo[0] + qz) * Ny + (lo[1] + qy)) * Nx + lo[2] + qx];
        if (!lj) continue;
        np++;
        atomicAdd(&tab[(li - 1) * Ng + (lj - 1)], 1u);
        if (sym) atomicAdd(&tab[(lj - 1) * Ng + (li - 1)], 1u);
      }
      np = wave_sum_i32(np);
      if (np == 0) continue;
      mcc_wave_sync();
      auto C = [&](int i, int j) -> double { return (double)tab[i * Ng + j]; };
      acc += wave_mcc(C, Ng, (double)((sym ? 2 : 1) * np), S, nmax, too_many);
      n_angles++;
      mcc_wave_sync();
      for (int k = lane; k < W; k += 64) {        // clear what this angle wrote
        const int kx = k % ext[2], kr = k / ext[2];
        const int ky = kr % ext[1], kz = kr / ext[1];
        const int qz = kz + dz, qy = ky + dy, qx = kx + dx;
        if ((unsigned)qz >= (unsigned)ext[0] || (unsigned)qy >= (unsigned)ext[1] || (unsigned)qx >= (unsigned)ext[2]) continue;
        const int li = L[((long long)(lo[0] + kz) * Ny + (lo[1] + ky)) * Nx + lo[2] + kx];
        if (!li) continue;
        const int lj = L[((long long)(lo[0] + qz) * Ny + (lo[1] + qy)) * Nx + lo[2] + qx];
        if (!lj) continue;
        tab[(li - 1) * Ng + (lj - 1)] = 0;
        tab[(lj - 1) * Ng + (li - 1)] = 0;
      }
      mcc_wave_sync();
    }
    if (lane == 0) out[v] = n_angles ? acc / (double)n_angles : __builtin_nan("");
  }
}

}  // namespace prad
