// Dense fallback of the GCR solve on gfx950: scipy.linalg.solve(A, b, assume_a='her') of the
// driver (example.py:944) for the symmetric positive-definite operator, as a blocked Cholesky.
//   * n < CHOL_BLOCKED_MIN_N: 32-wide right-looking steps, tuned for the launch chain at n ~ 2000;
//   * larger n: 256-wide panels with the trailing update on the FP64 matrix cores and
//     multi-workgroup triangular solves (config 5: 100 000 unknowns).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace hydra {

// ---------------------------------------------------------------- blocked Cholesky

constexpr int NB = 32;

__device__ __forceinline__ double readlane_f64(double x, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// Cholesky of one NB x NB block by ONE wave, rows in registers: lane i (< NB) holds row i of the
// lower triangle in v[]; column c's pivot and multipliers travel by v_readlane, so the NB steps
// need no LDS and no barrier (the barrier-per-column form took 16 us per block, this one ~2).
// Entries above the diagonal are never read.  Returns true if a pivot was not positive.
__device__ __forceinline__ bool potrf_wave(double (&v)[NB], int i) {
  bool bad = false;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double d = readlane_f64(v[c], c);
    if (!(d > 0.0)) bad = true;
    // reciprocal square root + one correction instead of sqrt and a divide per column: the two
    // long dependent sequences were most of this routine's time
    double inv = rsqrt(d);
    double sd = d * inv;
    sd = fma(fma(-sd, sd, d), 0.5 * inv, sd);
    v[c] = (i == c) ? sd : v[c] * inv;
#pragma unroll
    for (int j = c + 1; j < NB; ++j) {
      double ljc = readlane_f64(v[c], j);
      v[j] -= v[c] * ljc;
    }
  }
  return bad;
}

// factor the nb x nb diagonal block at (k,k) in place from the tile `t` in LDS (rows/cols >= nb
// are treated as identity); called by the first wave of a workgroup.
__device__ __forceinline__ void potrf_tile(const double (*t)[NB + 1], double* __restrict__ A,
                                           size_t ld, int k, int nb, int32_t* __restrict__ info,
                                           int lane) {
  int i = lane & (NB - 1);
  double v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = (i < nb && j < nb) ? t[i][j] : (i == j ? 1.0 : 0.0);
  bool bad = potrf_wave(v, i);
  if (bad && lane == 0) atomicOr(info, 1);
  if (lane < nb) {
    double* row = A + (size_t)(k + lane) * ld + k;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j <= lane) row[j] = v[j];
  }
}

// first diagonal block (the later ones are factored by the trailing update that completes them).
// Throughout the 32-wide kernels: ld = leading dimension, n = rows (= columns for a whole matrix;
// a standalone panel -- hydra_chol_panel -- has more rows than columns and ld = its width).
__global__ void __launch_bounds__(NB * NB)
potrf_diag_kernel(double* __restrict__ A, size_t ld, int n, int k, int32_t* __restrict__ info) {
  __shared__ double s[NB][NB + 1];
  int i = threadIdx.y, j = threadIdx.x;
  int nb = min(NB, n - k);
  s[i][j] = (i < nb && j < nb) ? A[(size_t)(k + i) * ld + k + j] : (i == j ? 1.0 : 0.0);
  __syncthreads();
  int tid = i * NB + j;
  if (tid < 64) potrf_tile(s, A, ld, k, nb, info, tid);
}

// rows below a full diagonal block: A[r, k:k+NB] <- A[r, k:k+NB] L11^-T, thread <-> row.  One wave
// per workgroup so the panel spreads over many CUs (the substitution reads L11 from LDS once per
// multiply-add); rows travel through LDS so that global accesses run along the rows.
constexpr int TRSM_T = 64;
static_assert(TRSM_T == 2 * NB, "trsm staging assumes two rows per trip");
__global__ void __launch_bounds__(TRSM_T)
trsm_panel_kernel(double* __restrict__ A, size_t ld, int n, int k) {
  __shared__ double l[NB][NB + 1];
  __shared__ double dinv[NB];
  __shared__ double rows[TRSM_T][NB + 1];
  const int tid = threadIdx.x;
  // unconditional loads (in bounds: the block is full, row indices are clamped) so that the
  // compiler can issue them all before the first use
  const int r0 = k + NB + blockIdx.x * TRSM_T;
  {
    const int j = tid % NB, i0 = tid / NB;   // TRSM_T / NB = 2 rows per trip
    double a[NB / 2], rr[NB];
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) a[q] = A[(size_t)(k + i0 + 2 * q) * ld + k + j];
#pragma unroll
    for (int q = 0; q < NB; ++q) rr[q] = A[(size_t)min(r0 + i0 + 2 * q, n - 1) * ld + k + j];
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) {
      int i = i0 + 2 * q;
      l[i][j] = (j <= i) ? a[q] : 0.0;
      if (i == j) dinv[i] = 1.0 / a[q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) rows[i0 + 2 * q][j] = rr[q];
  }
  __syncthreads();
  double v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = rows[tid][j];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double t = v[j];
#pragma unroll
    for (int c = 0; c < j; ++c) t -= v[c] * l[j][c];
    v[j] = t * dinv[j];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) rows[tid][j] = v[j];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    int i = tid / NB + 2 * q, j = tid % NB;
    if (r0 + i < n) A[(size_t)(r0 + i) * ld + k + j] = rows[i][j];
  }
}

// trailing update of the lower triangle: A22 -= L21 L21^T, 64x64 output tiles by 256 threads
// (4x4 entries each, the NB-deep panels in LDS).  The top-left NB x NB corner of tile (0,0) is the
// NEXT diagonal block, complete after this update: its workgroup factors it on the spot, which
// saves one launch per block column.  Only launched for full blocks (nb == NB).
// col_end (exclusive) caps the updated columns: n for the small-matrix path; the end of the current
// 256-wide panel for the large-matrix path, whose remaining columns get one rank-256 update on the
// matrix cores instead (chol_syrk_mfma_kernel).
constexpr int SY_T = 64;
__global__ void __launch_bounds__(256)
syrk_update_kernel(double* __restrict__ A, size_t ld, int n, int k, int col_end,
                   int32_t* __restrict__ info) {
  const int base = k + NB;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  __shared__ double sa[SY_T][NB + 1], sb[SY_T][NB + 1];
  const int tid = threadIdx.x, ty = tid / 16, tx = tid % 16;
  const int i0 = base + bi * SY_T, j0 = base + bj * SY_T;
  {
    const int c = tid % NB, r = tid / NB;   // 8 panel rows per trip, loads along the rows
    double pa[SY_T / 8], pb[SY_T / 8];
#pragma unroll
    for (int q = 0; q < SY_T / 8; ++q) {
      pa[q] = A[(size_t)min(i0 + r + 8 * q, n - 1) * ld + k + c];
      pb[q] = A[(size_t)min(j0 + r + 8 * q, n - 1) * ld + k + c];
    }
#pragma unroll
    for (int q = 0; q < SY_T / 8; ++q) {
      sa[r + 8 * q][c] = pa[q];
      sb[r + 8 * q][c] = pb[q];
    }
  }
  double old[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int gi = i0 + ty + 16 * a, gj = j0 + tx + 16 * b;
      old[a][b] = (gi < n && gj <= gi && gj < col_end) ? A[(size_t)gi * ld + gj] : 0.0;
    }
  __syncthreads();
  double acc[4][4] = {};
#pragma unroll 8
  for (int c = 0; c < NB; ++c) {
    double ra[4], rb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) ra[a] = sa[ty + 16 * a][c];
#pragma unroll
    for (int b = 0; b < 4; ++b) rb[b] = sb[tx + 16 * b][c];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += ra[a] * rb[b];
  }
  const bool corner = bi == 0;   // then bj == 0 too: this tile holds the next diagonal block
  if (corner) __syncthreads();   // sa is free: its first NB rows take the updated corner
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int li = ty + 16 * a, lj = tx + 16 * b;
      int gi = i0 + li, gj = j0 + lj;
      double val = old[a][b] - acc[a][b];
      if (corner && li < NB && lj < NB) sa[li][lj] = val;
      else if (gi < n && gj <= gi && gj < col_end) A[(size_t)gi * ld + gj] = val;
    }
  if (!corner) return;
  __syncthreads();
  if (tid < 64) potrf_tile(sa, A, ld, base, min(NB, n - base), info, tid);
}

// ---- one launch per 32-column step ------------------------------------------------------------
// The pair above is a chain of two dependent launches per block column (panel solve 9 us, update +
// next diagonal block 14 us at n = 2000: 62 x 23 us = the 1.4 ms of config 2's factorisation).
// chol_step_kernel does a whole step: the 64 x 64 tiles of the trailing update as before, and the
// workgroups of the FIRST tile column go on to the next block column -- each forms the next
// diagonal block D' = D - Pd Pd^T (32 x 32, from the panel rows it has staged anyway), factors it
// (one wave, ~2 us, the same in every one of them) and solves its own 64 rows of the next panel
// against it.  No workgroup waits for another.  What must not happen is a write to D while others
// still read it, so the factored diagonal block is NOT stored by this launch: the next launch has
// one extra workgroup (grid row nby) that forms and factors it again from the same inputs -- bit for
// bit the same values -- and stores it, off the chain; chol_store_diag_kernel does that for the last
// block of a panel.
constexpr int STEP_T = 256;

// sd <- D - Pd Pd^T for the nbd x nbd diagonal block at (kd, kd) (identity beyond nbd), Pd = rows
// 0..31 of `pd` (the solved panel rows kd..kd+31 of the previous block column).  All STEP_T threads;
// ends with a barrier.
__device__ __forceinline__ void step_next_diag(const double* __restrict__ A, size_t ld, int n, int kd,
                                               const double (*pd)[NB + 1], double (*sd)[NB + 1],
                                               int tid) {
  const int nbd = min(NB, n - kd);
  double dv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + STEP_T * q, i = e / NB, j = e % NB;
    dv[q] = A[(size_t)min(kd + i, n - 1) * ld + min(kd + j, n - 1)];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + STEP_T * q, i = e / NB, j = e % NB;
    double acc = 0.0;
#pragma unroll 8
    for (int c = 0; c < NB; ++c) acc += pd[i][c] * pd[j][c];
    sd[i][j] = (i < nbd && j < nbd) ? ((j <= i) ? dv[q] - acc : 0.0) : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
}

// factor sd in place (first wave); dinv <- reciprocals of the diagonal.  Returns "a pivot was bad"
// in every lane of the first wave.
__device__ __forceinline__ bool step_potrf(double (*sd)[NB + 1], double* dinv, int lane) {
  const int i = lane & (NB - 1);
  double v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = sd[i][j];
  const bool bad = potrf_wave(v, i);
  if (lane < NB) {
#pragma unroll
    for (int j = 0; j < NB; ++j) sd[lane][j] = (j <= lane) ? v[j] : 0.0;
    dinv[lane] = 1.0 / sd[lane][lane];   // (read back: v[lane] would index the registers by lane)
  }
  return bad;
}

// panel rows kd..kd+31 of the block column at kprev (32 x 32) into pd; all STEP_T threads, no barrier
__device__ __forceinline__ void step_load_pd(const double* __restrict__ A, size_t ld, int n, int kd,
                                             int kprev, double (*pd)[NB + 1], int tid) {
  double pv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + STEP_T * q, i = e / NB, j = e % NB;
    pv[q] = A[(size_t)min(kd + i, n - 1) * ld + kprev + j];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + STEP_T * q, i = e / NB, j = e % NB;
    pd[i][j] = (kd + i < n) ? pv[q] : 0.0;
  }
}

// store the factored diagonal block at (kd, kd): formed from the panel columns kd-32..kd-1 as in
// the step that used it
__device__ __forceinline__ void step_store_diag(double* __restrict__ A, size_t ld, int n, int kd,
                                                double (*pd)[NB + 1], double (*sd)[NB + 1],
                                                double* dinv, int32_t* __restrict__ info, int tid) {
  step_load_pd(A, ld, n, kd, kd - NB, pd, tid);
  __syncthreads();
  step_next_diag(A, ld, n, kd, pd, sd, tid);
  if (tid < 64) {
    const bool bad = step_potrf(sd, dinv, tid);
    if (bad && tid == 0) atomicOr(info, 1);
  }
  __syncthreads();
  const int nbd = min(NB, n - kd);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + STEP_T * q, i = e / NB, j = e % NB;
    if (i < nbd && j <= i) A[(size_t)(kd + i) * ld + kd + j] = sd[i][j];
  }
}

__global__ void __launch_bounds__(STEP_T)
chol_store_diag_kernel(double* __restrict__ A, size_t ld, int n, int kd, int32_t* __restrict__ info) {
  __shared__ double pd[NB][NB + 1], sd[NB][NB + 1];
  __shared__ double dinv[NB];
  step_store_diag(A, ld, n, kd, pd, sd, dinv, info, threadIdx.x);
}

// Step at block column k (full, with rows below): trailing update of the columns [k+32, col_end),
// next diagonal block and next panel.  Grid (column tiles, row tiles + 1); store_prev: the diagonal
// block at k is still to be stored (it was formed by the previous step, not by potrf_diag_kernel).
__global__ void __launch_bounds__(STEP_T)
chol_step_kernel(double* __restrict__ A, size_t ld, int n, int k, int col_end, int store_prev,
                 int32_t* __restrict__ info) {
  __shared__ double sa[SY_T][NB + 1], sb[SY_T][NB + 1];
  __shared__ double sd[NB][NB + 1];
  __shared__ double dinv[NB];
  const int base = k + NB;
  const int bi = blockIdx.y, bj = blockIdx.x, nby = gridDim.y - 1;
  const int tid = threadIdx.x, ty = tid / 16, tx = tid % 16;
  if (bi == nby) {                 // the extra row: its first workgroup stores the diagonal block at k
    if (bj == 0 && store_prev) step_store_diag(A, ld, n, k, sa, sd, dinv, info, tid);
    return;
  }
  if (bj > bi) return;
  const int i0 = base + bi * SY_T, j0 = base + bj * SY_T;
  {
    const int c = tid % NB, r = tid / NB;   // 8 panel rows per trip, loads along the rows
    double pa[SY_T / 8], pb[SY_T / 8];
#pragma unroll
    for (int q = 0; q < SY_T / 8; ++q) {
      pa[q] = A[(size_t)min(i0 + r + 8 * q, n - 1) * ld + k + c];
      pb[q] = A[(size_t)min(j0 + r + 8 * q, n - 1) * ld + k + c];
    }
#pragma unroll
    for (int q = 0; q < SY_T / 8; ++q) {
      sa[r + 8 * q][c] = (i0 + r + 8 * q < n) ? pa[q] : 0.0;
      sb[r + 8 * q][c] = (j0 + r + 8 * q < n) ? pb[q] : 0.0;
    }
  }
  double old[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int gi = i0 + ty + 16 * a, gj = j0 + tx + 16 * b;
      old[a][b] = (gi < n && gj <= gi && gj < col_end) ? A[(size_t)gi * ld + gj] : 0.0;
    }
  __syncthreads();
  double acc[4][4] = {};
#pragma unroll 8
  for (int c = 0; c < NB; ++c) {
    double ra[4], rb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) ra[a] = sa[ty + 16 * a][c];
#pragma unroll
    for (int b = 0; b < 4; ++b) rb[b] = sb[tx + 16 * b][c];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += ra[a] * rb[b];
  }
  const bool first = bj == 0;      // this tile's first 32 columns are the next panel
  const bool solve = first && n - base > NB;   // ... and there are rows below the next diagonal block
  if (first) {
    // the next diagonal block, from the panel rows base..base+31 = the first 32 rows of sb
    step_next_diag(A, ld, n, base, sb, sd, tid);     // (ends with a barrier: sa is free after it)
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int li = ty + 16 * a, lj = tx + 16 * b;
      const int gi = i0 + li, gj = j0 + lj;
      const double val = old[a][b] - acc[a][b];
      if (first && lj < NB) {
        // next panel: rows of the diagonal block are left alone (stored by the next launch),
        // rows below it go through the solve
        if (solve) sa[li][lj] = val;
      } else if (gi < n && gj <= gi && gj < col_end) {
        A[(size_t)gi * ld + gj] = val;
      }
    }
  if (!first) return;
  if (tid < 64) step_potrf(sd, dinv, tid);           // (a bad pivot is reported by the storing workgroup)
  __syncthreads();
  if (!solve) return;
  if (tid < SY_T) {                // thread <-> row: v <- v L11^-T
    double v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = sa[tid][j];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      double t0 = v[j], t1 = 0.0;
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        if (c < j) {               // (j and c are constants once unrolled)
          if (c & 1) t1 -= v[c] * sd[j][c];
          else       t0 -= v[c] * sd[j][c];
        }
      }
      v[j] = (t0 + t1) * dinv[j];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) sa[tid][j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < SY_T / 8; ++q) {
    const int li = tid / NB + 8 * q, j = tid % NB;
    const int gi = i0 + li;
    if (gi >= base + NB && gi < n) A[(size_t)gi * ld + base + j] = sa[li][j];
  }
}

// Both triangular solves of the Cholesky fallback in ONE launch by one workgroup of NB*NB threads
// (a kernel pair per block column needs 4 n / NB launches of 5-18 us each, most of the solve time
// at n ~ 2000).  Left-looking: block k of the solution needs sum_j L[k, j] y_j over the blocks
// already solved, a (NB x k) product whose loads run along the rows of L (forward sweep) or along
// the NB-wide column panel (backward sweep) - 256-byte pieces either way, never one line per lane.
// Thread (a, c) = (tid / NB, tid % NB) leaves a partial sum in part[][]; the first wave adds them
// up and does the NB x NB substitution in registers.  L is read once per sweep.
constexpr int TRSV_T = NB * NB;

// element (tid / NB, tid % NB) of the diagonal block at (k,k), clamped into the matrix (entries
// outside the nb x nb lower triangle are masked when staged)
__device__ __forceinline__ double trsv_diag_load(const double* __restrict__ A, int ld, int n, int k,
                                                 int tid) {
  int i = min(max(k, 0) + tid / NB, n - 1), j = min(max(k, 0) + tid % NB, n - 1);
  return A[(size_t)i * ld + j];
}

// BL: the solution vector lives in dynamic LDS (copied in and out), next to the multiply-adds;
// otherwise it stays in b.  (A template, not a flat pointer: flat loads wait on both memory
// counters and would serialise against the loads of L.)
template <bool BL>
__global__ void __launch_bounds__(TRSV_T)
trsv_fused_kernel(const double* __restrict__ A, int ld, int n, double* __restrict__ b,
                  int sweep_begin, int sweep_end) {
  extern __shared__ double b_lds[];
  __shared__ double l[NB][NB + 1];
  __shared__ double part[NB][NB + 1];
  __shared__ double dinv[NB];
  const int tid = threadIdx.x;
  const int ta = tid / NB, tc = tid % NB;
  if (BL)
    for (int i = tid; i < n; i += TRSV_T) b_lds[i] = b[i];
  auto bv = [&](int i) -> double& { return BL ? b_lds[i] : b[i]; };
  __syncthreads();
  const int last = ((n - 1) / NB) * NB;
  // sweep 0: L y = b left to right, 1: L^T x = y right to left.  A is an n x n block with leading
  // dimension ld (the whole matrix, or one diagonal block of the large-matrix path).
  for (int sweep = sweep_begin; sweep < sweep_end; ++sweep) {
    double a = trsv_diag_load(A, ld, n, sweep ? last : 0, tid);
    for (int k = sweep ? last : 0; sweep ? k >= 0 : k < n; k += sweep ? -NB : NB) {
      const int nb = min(NB, n - k);
      l[ta][tc] = (ta < nb && tc <= ta) ? a : 0.0;
      if (ta == tc) dinv[ta] = (ta < nb) ? 1.0 / a : 1.0;
      a = trsv_diag_load(A, ld, n, sweep ? k - NB : k + NB, tid);   // next block, in flight meanwhile
      double s0 = 0.0, s1 = 0.0;
      if (!sweep) {   // row k + ta of L against y[0:k]; this thread takes columns tc, tc + NB, ...
        const double* row = A + (size_t)min(k + ta, n - 1) * ld;
        for (int j = tc; j < k; j += 8 * NB) {   // 8 loads in flight; the tail is masked, not serial
          double rv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) rv[u] = row[min(j + u * NB, k - NB + tc)];
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            int j0 = j + u * NB, j1 = j0 + NB;
            s0 += (j0 < k) ? rv[u] * bv(min(j0, k - 1)) : 0.0;
            s1 += (j1 < k) ? rv[u + 1] * bv(min(j1, k - 1)) : 0.0;
          }
        }
      } else if (nb == NB) {   // column k + tc of L against x[k+NB:n]; rows ta, ta + NB, ... of it
        const double* col = A + k + tc;
        for (int r = k + NB + ta; r < n; r += 8 * NB) {
          double rv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) rv[u] = col[(size_t)min(r + u * NB, n - 1) * ld];
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            int r0 = r + u * NB, r1 = r0 + NB;
            s0 += (r0 < n) ? rv[u] * bv(min(r0, n - 1)) : 0.0;
            s1 += (r1 < n) ? rv[u + 1] * bv(min(r1, n - 1)) : 0.0;
          }
        }
      }
      if (!sweep) part[tc][ta] = s0 + s1;   // part[a][i]: partial a of solution entry i
      else        part[ta][tc] = s0 + s1;
      __syncthreads();
      if (tid < 64) {   // lane i owns entry i of the block
        const int lane = tid & (NB - 1);
        double yi = (lane < nb) ? bv(k + lane) : 0.0, lr[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) yi -= part[q][lane];
        if (!sweep) {
#pragma unroll
          for (int c = 0; c < NB; ++c) lr[c] = l[lane][c];
#pragma unroll
          for (int c = 0; c < NB; ++c) {
            double yc = readlane_f64(yi, c) * dinv[c];
            yi = (lane == c) ? yc : ((lane > c) ? yi - lr[c] * yc : yi);
          }
        } else {
#pragma unroll
          for (int c = 0; c < NB; ++c) lr[c] = l[c][lane];
#pragma unroll
          for (int c = NB - 1; c >= 0; --c) {
            double yc = readlane_f64(yi, c) * dinv[c];
            yi = (lane == c) ? yc : ((lane < c) ? yi - lr[c] * yc : yi);
          }
        }
        if (tid < nb) bv(k + tid) = yi;
      }
      __syncthreads();
    }
  }
  if (BL)
    for (int i = tid; i < n; i += TRSV_T) b[i] = b_lds[i];
}


// ---------------------------------------------------------------- large matrices
//
// The driver's default for the source head is a dense solve whatever --solver says
// (example.py:940-947); at BASELINE config 5 that is a 100 000 x 100 000 operator (80 GB), N^3/3 =
// 333 TFLOP.  Right-looking blocked Cholesky with 256-wide panels:
//   * the panel (all rows below, 256 columns) is factored by the 32-wide steps above, their
//     updates capped at the panel's last column (HBM-bound, ~2 % of the time);
//   * the rest of the matrix takes ONE rank-256 update per panel on the FP64 matrix cores,
//     C -= P P^T over the lower-triangle tiles, P = the finished panel read in place.
// chol_syrk_mfma_kernel: 128 x 128 tile per 4-wave workgroup (wave tile 64 x 64 = 16 accumulators
// of v_mfma_f64_16x16x4_f64), two workgroups per CU so that one's prologue / epilogue / barriers
// run under the other's matrix work.  K is walked in chunks of 16: both operand strips
// (128 rows x 128 B each) are copied by LDS-DMA into one of two LDS stages one chunk ahead.
// Lane (r = l & 15, g = l >> 4) of an MFMA needs P[row r][k = 4 step + g]: since the product sums
// over k, any assignment of the chunk's 16 k-values to (step, g) works if both operands use the
// same one -- lane group g reads the 16-byte pair (k = 2c, 2c+1), c = 4h + g, with ONE
// ds_read_b128 and feeds steps 2h and 2h+1 from its halves.  A row is 128 B = 8 pairs; pair c of row
// r sits at position c ^ ((r >> 1) & 7) (the per-lane DMA source address does the swizzle), which
// makes every ds_read_b128 group hit 16 distinct 4-bank slots.
// Algorithmic work: rest (rest + 128) * 256 flops per launch (rest = rows below the panel), bound:
// FP64 MFMA; HBM traffic per tile: 128 KB of C in and out (32 flop/B) + the strips from L2.
constexpr int PB = 256;           // panel width = K of one trailing update
constexpr int CT = 128;           // output tile edge
constexpr int KC = 16;            // k-values per LDS stage
constexpr int NKC = PB / KC;
constexpr int STG_BYTES = 2 * CT * KC * 8;   // one stage: A strip + B strip = 32 KiB
constexpr int SUP = 8;            // tiles per edge of the patch one XCD works on at a time

__device__ __forceinline__ void tri_of(int p, int* i, int* j) {   // p = i (i+1)/2 + j, j <= i
  int a = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
  while ((a + 1) * (a + 2) / 2 <= p) ++a;
  while (a * (a + 1) / 2 > p) --a;
  *i = a;
  *j = p - a * (a + 1) / 2;
}

__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// General form: C[i][j] -= sum_k PA[i][k] PB[j][k] (k < 256) for i < mrows, j < ncols and
// j <= i + diag.  The factorisation of one matrix calls it with C, PA, PB inside that matrix
// (triangular tile order below); the row-sharded factorisation (hydra_chol_update_rect) with a
// rank's own rows as C / PA and the gathered panel as PB (plain 2-D tile order, tiles entirely
// above the diagonal leave at once).
__global__ void __launch_bounds__(256, 2)
chol_syrk_mfma_kernel(double* __restrict__ C, long long ldc, const double* __restrict__ PA,
                      long long lda, const double* __restrict__ PB, long long ldb, int mrows,
                      int ncols, int diag, int ntr, int nsup, int ntc, int nkc) {
  __shared__ __align__(16) unsigned char smem[2 * STG_BYTES];
  int ti, tj;
  if (nsup > 0) {
    // block -> tile: blocks with equal id % 8 share an XCD (round-robin dispatch) and walk one
    // 8 x 8 patch of tiles after another, so the ~64 tiles resident on an XCD share 16 row strips
    const int id = blockIdx.x, slot = id >> 3;
    const int sidx = (slot / (SUP * SUP)) * 8 + (id & 7);
    if (sidx >= nsup) return;
    int si, sj;
    tri_of(sidx, &si, &sj);
    const int within = slot % (SUP * SUP);
    ti = si * SUP + within / SUP;
    tj = sj * SUP + within % SUP;
    if (ti >= ntr || tj > ti) return;
  } else if (nsup < 0) {
    // rectangle in 8 x 8 patches, one patch after another per XCD (the row-sharded factorisation's
    // updates: in plain row-major order every XCD met every strip of both operands)
    const int id = blockIdx.x, slot = id >> 3;
    const int npc = (ntc + SUP - 1) / SUP, npr = (ntr + SUP - 1) / SUP;
    const int patch = (slot / (SUP * SUP)) * 8 + (id & 7);
    if (patch >= npr * npc) return;
    const int within = slot % (SUP * SUP);
    ti = (patch / npc) * SUP + within / SUP;
    tj = (patch % npc) * SUP + within % SUP;
    if (ti >= ntr || tj >= ntc || (long long)tj * CT > (long long)ti * CT + CT - 1 + diag) return;
  } else {
    ti = blockIdx.x / ntc;
    tj = blockIdx.x % ntc;
    if (ti >= ntr || (long long)tj * CT > (long long)ti * CT + CT - 1 + diag) return;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, g = lane >> 4;
  const int rowA0 = ti * CT, rowB0 = tj * CT;

  // LDS-DMA: a stage is 32 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w, w+4, ...:
  // the first four belong to the A strip, the last four to the B strip.  Rows past the matrix
  // are clamped (their products land in rows / columns that are not stored).
  const char* baseA = reinterpret_cast<const char*>(PA + (size_t)rowA0 * lda);
  const char* baseB = reinterpret_cast<const char*>(PB + (size_t)rowB0 * ldb);
  unsigned off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = wave + 4 * i;
    const int pr = (q & 15) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((pr >> 1) & 7);
    const int row0 = (i < 4) ? rowA0 : rowB0;
    const int gr = min(row0 + pr, ((i < 4) ? mrows : ncols) - 1) - row0;
    off[i] = (unsigned)gr * (unsigned)((i < 4) ? lda : ldb) * 8u + (unsigned)c * 16u;
  }
  // (LDS-DMA by hand-written asm: through the compiler's builtin every later wait in the wave
  // becomes lgkmcnt(0) / vmcnt(0) while a DMA may be in flight -- see gram_uniform_kernel)
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem + wave * 1024));
  auto issue = [&](int kc, int buf, int i0, int i1) {
    const unsigned dst = lds_wave + (unsigned)(buf * STG_BYTES);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i >= i0 && i < i1) {
        const char* src = ((i < 4) ? baseA : baseB) + kc * (KC * 8);     // uniform
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst + (unsigned)(i * 4096)), "v"(off[i]), "s"(src) : "memory");
      }
    }
  };
  issue(0, 0, 0, 8);

  // accumulators start as -C (negation is exact; the sum of products is added, -acc stored)
  f64x4 acc[4][4];
  const int ci = rowA0 + wr * 64 + g, cj = rowB0 + wc * 64 + lr;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int gi = min(ci + mi * 16 + 4 * e, mrows - 1), gj = min(cj + ni * 16, ncols - 1);
        acc[mi][ni][e] = -C[(size_t)gi * ldc + gj];
      }

  const int swz = (lr >> 1) & 7;
  const unsigned char* rdA = smem + (wr * 64 + lr) * (KC * 8);
  const unsigned char* rdB = smem + CT * KC * 8 + (wc * 64 + lr) * (KC * 8);
  const int p0 = ((0 + g) ^ swz) * 16, p1 = ((4 + g) ^ swz) * 16;

  for (int kc = 0; kc < nkc; ++kc) {
    const int buf = kc & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk kc have landed
    __syncthreads();                                   // ... everyone's; stage buf^1 is free again
    const bool more = kc + 1 < nkc;
    const unsigned char* sa = rdA + buf * STG_BYTES;
    const unsigned char* sb = rdB + buf * STG_BYTES;
    // Reads in the order of use (first half of the chunk's k-values, then the second: the second
    // half arrives under the first 32 matrix instructions, the compiler's waits are counted);
    // the next chunk's staging goes out two pieces at a time between the bursts.
    f64x2 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      a0[m] = *reinterpret_cast<const f64x2*>(sa + m * 16 * KC * 8 + p0);
      b0[m] = *reinterpret_cast<const f64x2*>(sb + m * 16 * KC * 8 + p0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      a1[m] = *reinterpret_cast<const f64x2*>(sa + m * 16 * KC * 8 + p1);
      b1[m] = *reinterpret_cast<const f64x2*>(sb + m * 16 * KC * 8 + p1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma64(a0[mi].x, b0[ni].x, acc[mi][ni]);
      if (more) issue(kc + 1, buf ^ 1, mi, mi + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma64(a0[mi].y, b0[ni].y, acc[mi][ni]);
      if (more) issue(kc + 1, buf ^ 1, 4 + mi, 5 + mi);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma64(a1[mi].x, b1[ni].x, acc[mi][ni]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma64(a1[mi].y, b1[ni].y, acc[mi][ni]);
  }

#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int gi = ci + mi * 16 + 4 * e, gj = cj + ni * 16;
        if (gi < mrows && gj < ncols && gj <= gi + diag) C[(size_t)gi * ldc + gj] = -acc[mi][ni][e];
      }
}

// Triangular solves for large n, one 256-wide panel at a time, ONE launch per panel.  A sweep is a
// dependent chain (n = 100 000: 391 panels), so what a link costs is what matters: with the diagonal
// block's substitution and the update of the rest of the vector as two launches a panel took 52 us
// (30 + 22) at that size, most of the 30 being first-touch latencies of L, one per 32-row step.  Now
// launch p has a SOLVING workgroup -- it takes the previous panel's solution out of panel p's own 256
// entries (a 256 x 256 piece of L) and then solves the diagonal block, the loads of each 32-row step
// issued one step ahead -- beside the workgroups that take the previous panel's solution out of
// everything further on:
//   forward : b[r] -= L[r, k-256:k] . y_prev   for the rows r beyond panel p (one wave per 4 rows,
//             4 columns per lane: a row of the panel is 2 KiB contiguous);
//   backward: b[c] -= sum_r L[k+256+r, c] x_next[r] for the columns c left of panel p (a workgroup per
//             256 columns, its four quarters take 64 rows each: coalesced along the rows of L).
// Right-looking both ways, so no sum crosses workgroups; fixed order.  Each sweep streams L once:
// 4 n^2 bytes, HBM-bound.
constexpr int SWT = 1024;          // threads of a sweep workgroup (the 32 x 32 layout of a substitution step)
static_assert(SWT == NB * NB && PB == 256, "sweep kernels: 32 x 32 threads, 256-wide panels");

struct SweepShared {
  double l[NB][NB + 1];
  double part[NB][NB + 1];
  double dinv[NB];
  double bl[PB + 2];               // the panel's entries of the vector; bl[PB] stays 0.0 (masked operands)
  double red[4][PB];               // backward: the four row quarters' sums per column
};

// One 32-row step of the solve with the nb x nb (nb <= 256) lower-triangular block D (leading
// dimension ld) against sh.bl -- SWEEP 0: L y = b top down, 1: L^T x = y bottom up -- at rows k.
// All SWT threads; thread (ta, tc) = (tid / 32, tid % 32).  `a` is this thread's entry of the step's
// 32 x 32 diagonal block and rv its (at most 7) values of L outside it; the next step's are loaded
// into an / rvn before anything here waits (two register sets that the caller swaps: copying one set
// into the other would wait for the loads at once).
template <int SWEEP>
__device__ __forceinline__ void sweep_step(const double* __restrict__ D, size_t ld, int nb, int k,
                                           bool more, SweepShared& sh, int tid, double a,
                                           const double (&rv)[7], double& an, double (&rvn)[7]) {
  const int ta = tid / NB, tc = tid % NB;
  const int w = min(NB, nb - k);
  // the step's diagonal block without its diagonal (whose reciprocals go to dinv): the substitution
  // factors below are then zero wherever an entry is closed, with nothing to test
  sh.l[ta][tc] = (ta < w && tc < ta) ? a : 0.0;
  if (ta == tc) sh.dinv[ta] = (ta < w) ? 1.0 / a : 1.0;
  if (more) {
    const int kn = SWEEP ? k - NB : k + NB;
    an = D[(size_t)min(kn + ta, nb - 1) * ld + min(kn + tc, nb - 1)];
    if (SWEEP == 0) {              // row kn + ta, columns tc + 32 u (used where < kn)
      const double* row = D + (size_t)min(kn + ta, nb - 1) * ld;
#pragma unroll
      for (int u = 0; u < 7; ++u) rvn[u] = row[min(tc + u * NB, nb - 1)];
    } else {                       // column kn + tc, rows kn + 32 + ta + 32 u (used where < nb)
      const double* col = D + min(kn + tc, nb - 1);
#pragma unroll
      for (int u = 0; u < 7; ++u) rvn[u] = col[(size_t)min(kn + NB + ta + u * NB, nb - 1) * ld];
    }
  }
  // this thread's part of (rows of L) . (entries solved so far); where a product does not count its
  // vector operand is read from the zero slot -- a test on the product became a branch around an
  // LDS read with a wait of its own, seven times
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    const int i = SWEEP ? k + NB + ta + u * NB : tc + u * NB;
    const bool in = SWEEP ? i < nb : i < k;
    const double t = rv[u] * sh.bl[in ? i : PB];
    if (u & 1) s1 += t; else s0 += t;
  }
  if (SWEEP == 0) sh.part[tc][ta] = s0 + s1;   // part[q][i]: partial q of the step's entry i
  else            sh.part[ta][tc] = s0 + s1;
  __syncthreads();
  if (tid < 64) {                  // lane i owns entry i of the step (lanes 32-63 mirror 0-31)
    // The substitution is a chain of 32 dependent steps, so a step is kept to a lane broadcast and
    // one multiply-add: z_i is the running numerator b_i - sum_j l_ij y_j, step c subtracts
    // (l_ic / l_cc) z_c from the entries still open -- the factor is formed ahead of the chain and
    // is zero for the entries that are done -- and y = z / l_ii at the end.
    const int lane = tid & (NB - 1);
    double z = sh.bl[lane < w ? k + lane : PB];
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
    for (int q = 0; q < NB; q += 4) {
      p0 += sh.part[q][lane];
      p1 += sh.part[q + 1][lane];
      p2 += sh.part[q + 2][lane];
      p3 += sh.part[q + 3][lane];
    }
    z -= (p0 + p1) + (p2 + p3);
    // (the factors eight at a time: all 32 of them beside the prefetched values of L spill)
#pragma unroll
    for (int h = 0; h < NB; h += 8) {
      const int c0 = SWEEP ? NB - 8 - h : h;
      double m[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u;
        m[u] = (SWEEP ? sh.l[c][lane] : sh.l[lane][c]) * sh.dinv[c];
      }
      if (SWEEP == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) z -= m[u] * readlane_f64(z, c0 + u);
      } else {
#pragma unroll
        for (int u = 7; u >= 0; --u) z -= m[u] * readlane_f64(z, c0 + u);
      }
    }
    if (tid < w) sh.bl[k + tid] = z * sh.dinv[tid];
  }
  __syncthreads();
}

// The whole block: sh.bl[0:nb] in, solution out; sh.bl[PB] must hold 0.0.
template <int SWEEP>
__device__ __forceinline__ void sweep_block_solve(const double* __restrict__ D, size_t ld, int nb,
                                                  SweepShared& sh, int tid) {
  const int ta = tid / NB, tc = tid % NB;
  const int nsteps = (nb + NB - 1) / NB;
  const int dk = SWEEP ? -NB : NB;
  int k = SWEEP ? (nsteps - 1) * NB : 0;
  double a0 = D[(size_t)min(k + ta, nb - 1) * ld + min(k + tc, nb - 1)], a1 = 0.0, rv0[7], rv1[7];
  if (SWEEP == 0) {                // the first step has nothing outside its diagonal block; the last
#pragma unroll                     // step of the backward sweep neither
    for (int u = 0; u < 7; ++u) rv0[u] = 0.0;
  } else {
#pragma unroll
    for (int u = 0; u < 7; ++u) rv0[u] = 0.0;
  }
#pragma unroll 1
  for (int it = 0; it < nsteps; it += 2) {
    sweep_step<SWEEP>(D, ld, nb, k, it + 1 < nsteps, sh, tid, a0, rv0, a1, rv1);
    k += dk;
    if (it + 1 < nsteps) {
      sweep_step<SWEEP>(D, ld, nb, k, it + 2 < nsteps, sh, tid, a1, rv1, a0, rv0);
      k += dk;
    }
  }
}

// Forward, panel p at rows / columns k = 256 p (pb of them; only the last panel is narrower).
// Workgroup 0 solves, the others update the rows beyond the panel with the previous panel's solution.
__global__ void __launch_bounds__(SWT)
chol_fwd_step_kernel(const double* __restrict__ A, size_t ld, int n, int k, double* __restrict__ b) {
  __shared__ SweepShared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pb = min(PB, n - k);
  f64x2 y0 = {0.0, 0.0}, y1 = {0.0, 0.0};
  if (k > 0) {
    const f64x2* yp = reinterpret_cast<const f64x2*>(b + k - PB);
    y0 = yp[2 * lane];
    y1 = yp[2 * lane + 1];
  }
  // NR rows (from r, clamped below r_end) of the previous panel's 256 columns against its solution:
  // b[row] minus the product, in every lane.  All loads first, then the NR wave sums side by side
  // (one after the other each of them is six dependent lane exchanges).
  auto rows = [&](int r, int r_end, auto& out) {
    constexpr int NR = sizeof(out) / sizeof(double);
    f64x2 v[NR][2];
    double bq[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int rr = min(r + u, r_end - 1);
      const f64x2* row = reinterpret_cast<const f64x2*>(A + (size_t)rr * ld + (k - PB));
      v[u][0] = row[2 * lane];
      v[u][1] = row[2 * lane + 1];
      bq[u] = b[rr];
    }
    // (b rides in lane 0's term of the sum: used only in the store at the end, its load moved
    // there, behind the reductions and under a wait of its own)
    double t[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u)
      t[u] = ((v[u][0].x * y0.x + v[u][0].y * y0.y) + (v[u][1].x * y1.x + v[u][1].y * y1.y)) -
             (lane == 0 ? bq[u] : 0.0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < NR; ++u) t[u] += __shfl_down(t[u], off, 64);
#pragma unroll
    for (int u = 0; u < NR; ++u) out[u] = -t[u];
  };
  if (tid == 0) sh.bl[PB] = 0.0;
  if (blockIdx.x == 0) {
    if (k > 0) {
      for (int r = wave * 8; r < pb; r += 128) {     // 16 waves x 8 rows a trip
        double out[8];
        rows(k + r, k + pb, out);
        if (lane == 0)                               // (rows beyond pb: clamped repeats, never read)
#pragma unroll
          for (int u = 0; u < 8; ++u) sh.bl[r + u] = out[u];
      }
    } else {
      for (int i = tid; i < pb; i += SWT) sh.bl[i] = b[i];
    }
    __syncthreads();
    sweep_block_solve<0>(A + (size_t)k * ld + k, ld, pb, sh, tid);
    for (int i = tid; i < pb; i += SWT) b[k + i] = sh.bl[i];
  } else {
    for (int r = k + PB + ((blockIdx.x - 1) * 16 + wave) * 4; r < n; r += (gridDim.x - 1) * 64) {
      double out[4];
      rows(r, n, out);
      if (lane == 0)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (r + u < n) b[r + u] = out[u];
    }
  }
}

// Backward, panel p at k = 256 p (pb entries); the panel after it (rows k + 256 .., pbn of them) is
// solved.  Workgroup 0 solves panel p, workgroup j > 0 updates the columns 256 (j - 1) .. left of it.
__global__ void __launch_bounds__(SWT)
chol_bwd_step_kernel(const double* __restrict__ A, size_t ld, int n, int k, double* __restrict__ b) {
  __shared__ SweepShared sh;
  __shared__ double xn[PB];
  const int tid = threadIdx.x;
  const int pb = min(PB, n - k);
  const int kn = k + PB, pbn = max(0, min(PB, n - kn));
  for (int i = tid; i < pbn; i += SWT) xn[i] = b[kn + i];
  if (tid == 0) sh.bl[PB] = 0.0;
  __syncthreads();
  // column c0 + (tid % 256) against x_next, the rows split in four quarters (tid / 256); the sum of
  // the quarters lands in sh.red[0][.]
  auto columns = [&](int c0, int ncols) -> double {
    const int cl = tid & (PB - 1), q = tid >> 8;
    const double* col = A + (size_t)kn * ld + c0 + min(cl, ncols - 1);
    const int r_begin = q * 64, r_end = min(pbn, r_begin + 64);
    double s0 = 0.0, s1 = 0.0;
    for (int r = r_begin; r < r_end; r += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)min(r + u, pbn - 1) * ld];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        s0 += (r + u < r_end) ? v[u] * xn[min(r + u, pbn - 1)] : 0.0;
        s1 += (r + u + 1 < r_end) ? v[u + 1] * xn[min(r + u + 1, pbn - 1)] : 0.0;
      }
    }
    sh.red[q][cl] = s0 + s1;
    __syncthreads();
    return (sh.red[0][cl] + sh.red[1][cl]) + (sh.red[2][cl] + sh.red[3][cl]);
  };
  if (blockIdx.x == 0) {
    if (pbn > 0) {
      const double t = columns(k, pb);
      if (tid < pb) sh.bl[tid] = b[k + tid] - t;
    } else {
      if (tid < pb) sh.bl[tid] = b[k + tid];
    }
    __syncthreads();
    sweep_block_solve<1>(A + (size_t)k * ld + k, ld, pb, sh, tid);
    if (tid < pb) b[k + tid] = sh.bl[tid];
  } else {
    const int c0 = (blockIdx.x - 1) * PB;       // < k, a whole 256 columns (k is a multiple of 256)
    const double t = columns(c0, PB);
    if (tid < PB) b[c0 + tid] -= t;
  }
}

// y += A x for a symmetric operator of which only the STRICT UPPER triangle and the diagonal are
// left -- after an in-place factorisation the lower triangle holds L -- for the rows a rank owns
// (local row i = global row rows[i], ascending): first the rows (one wave per row: d_i x_i and the
// entries right of the diagonal), then their mirror images (one thread per column j: the entries
// above the diagonal of column j in this rank's rows).  The residual check of a kept factor.
__global__ void __launch_bounds__(256)
symv_upper_rows_kernel(const double* __restrict__ A, size_t ld, int nloc, int n,
                       const long long* __restrict__ rows, const double* __restrict__ diag,
                       const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < nloc; i += gridDim.x * 4) {
    const int gi = (int)min(max(rows[i], 0LL), (long long)n - 1);   // (a bad index must not fault)
    const double* row = A + (size_t)i * ld;
    double s0 = 0.0, s1 = 0.0;
    int j = gi + 1 + lane;
    for (; j + 64 < n; j += 128) {
      s0 += row[j] * x[j];
      s1 += row[j + 64] * x[j + 64];
    }
    if (j < n) s0 += row[j] * x[j];
    const double t = wave_sum(s0 + s1);
    if (lane == 0) y[gi] += diag[i] * x[gi] + t;
  }
}
__global__ void __launch_bounds__(256)
symv_upper_cols_kernel(const double* __restrict__ A, size_t ld, int nloc, int n,
                       const long long* __restrict__ rows, const double* __restrict__ x,
                       double* __restrict__ y) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jc = min(j, n - 1);
  const double* col = A + jc;
  double s0 = 0.0, s1 = 0.0;
  // rows with global index < the first column of this workgroup count for all of its threads;
  // the few after that are tested one by one
  const int j_first = blockIdx.x * 256;
  int i = 0;
  for (; i + 8 <= nloc && rows[i + 7] < j_first; i += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(i + u) * ld];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0 += v[u] * x[max(rows[i + u], 0LL)];                 // (< j_first <= n here)
      s1 += v[u + 1] * x[max(rows[i + u + 1], 0LL)];
    }
  }
  for (; i < nloc && rows[i] < j_first + 256; ++i)
    if (rows[i] < j) s0 += col[(size_t)i * ld] * x[max(rows[i], 0LL)];
  if (j < n) y[j] += s0 + s1;
}

// X = L^-1, one workgroup per block column J of 32 columns: for the row blocks I = J, J+1, ...
//   X[I,J] = L[I,I]^-1 ( [I == J] 1 - sum_{J <= K < I} L[I,K] X[K,J] ),
// the 32 x 32 products by all 32 x 32 threads through LDS, the diagonal-block solve by 32 threads
// (one column each).  The workgroup reads back the blocks of X it wrote itself (after a barrier:
// a first touch of those lines by this CU).  Set-up work, once per matrix: n = 1681 takes 3.9 ms,
// n = 4096 21 ms (rocBLAS trsm: 0.8 / 3.2 ms; tools/tri_inverse_bench.py).
__global__ void __launch_bounds__(NB * NB)
tri_inverse_kernel(const double* __restrict__ L, size_t ld, int n, double* X, size_t ldx) {
  __shared__ double sL[NB][NB + 1], sX[NB][NB + 1], sR[NB][NB + 1];
  const int tid = threadIdx.x, ta = tid / NB, tc = tid % NB;
  const int J = blockIdx.x, nbk = (n + NB - 1) / NB;
  const int col = NB * J + tc;
  for (int r = ta; r < NB * J; r += NB)                      // above the diagonal block: zeros
    if (col < n) X[(size_t)r * ldx + col] = 0.0;
  for (int I = J; I < nbk; ++I) {
    const int row = NB * I + ta;
    double acc = 0.0;
    for (int K = J; K < I; ++K) {
      // (row < n here only in the last block row; K < I, so the columns 32 K + tc are < n)
      sL[ta][tc] = (row < n) ? L[(size_t)row * ld + NB * K + tc] : 0.0;
      sX[ta][tc] = (col < n) ? X[(size_t)(NB * K + ta) * ldx + col] : 0.0;
      __syncthreads();
#pragma unroll 8
      for (int c = 0; c < NB; ++c) acc += sL[ta][c] * sX[c][tc];
      __syncthreads();
    }
    sR[ta][tc] = ((I == J && ta == tc) ? 1.0 : 0.0) - acc;
    {
      const int dc = NB * I + tc;
      sL[ta][tc] = (row < n && dc < n) ? ((tc <= ta) ? L[(size_t)row * ld + dc] : 0.0)
                                       : (ta == tc ? 1.0 : 0.0);
    }
    __syncthreads();
    if (ta == 0) {                                            // thread tc: column tc of the block
      double y[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        double t0 = sR[r][tc], t1 = 0.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          if (c < r) {
            if (c & 1) t1 -= sL[r][c] * y[c];
            else       t0 -= sL[r][c] * y[c];
          }
        }
        y[r] = (t0 + t1) / sL[r][r];
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) sR[r][tc] = y[r];
    }
    __syncthreads();
    if (row < n && col < n) X[(size_t)row * ldx + col] = sR[ta][tc];
    __syncthreads();
  }
}

// y[i] -= sum_j M[i][j] x[j] for an nrows x ncols piece of a row-major matrix (leading dimension
// ld): one wave per row, the lanes along the row.  The forward sweep of the row-sharded solve: a
// rank's rows below a solved block take that block's part of the solution.
__global__ void __launch_bounds__(256)
rows_update_kernel(const double* __restrict__ M, size_t ld, int nrows, int ncols,
                   const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = blockIdx.x * 4 + wave; i < nrows; i += gridDim.x * 4) {
    const double* row = M + (size_t)i * ld;
    double s0 = 0.0, s1 = 0.0;
    int j = lane;
    for (; j + 64 < ncols; j += 128) {
      s0 += row[j] * x[j];
      s1 += row[j + 64] * x[j + 64];
    }
    if (j < ncols) s0 += row[j] * x[j];
    const double t = wave_sum(s0 + s1);
    if (lane == 0) y[i] -= t;
  }
}

// b[j] -= sum_i M[i][j] x[i], j < ncols: one thread per column, coalesced along the rows of M.  The
// backward sweep: the columns left of a solved block take that block's rows.
__global__ void __launch_bounds__(256)
cols_update_kernel(const double* __restrict__ M, size_t ld, int nrows, int ncols,
                   const double* __restrict__ x, double* __restrict__ b) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int cc = min(c, ncols - 1);
  const double* col = M + cc;
  double s0 = 0.0, s1 = 0.0;
  int r = 0;
  for (; r + 8 <= nrows; r += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(r + u) * ld];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0 += v[u] * x[r + u];
      s1 += v[u + 1] * x[r + u + 1];
    }
  }
  for (; r < nrows; ++r) s0 += col[(size_t)r * ld] * x[r];
  if (c < ncols) b[c] -= s0 + s1;
}
}  // namespace hydra

using namespace hydra;

// Small matrices: below this the 32-wide path (a chain of ~4 n / 32 short launches) wins.
static constexpr int CHOL_BLOCKED_MIN_N = 6144;

static int chol_small_solve(const double* A, double* b, int n, hipStream_t s) {
  size_t b_bytes = (size_t)n * sizeof(double);
  int use_lds = b_bytes <= 96 * 1024;   // b rides in LDS when it fits beside the staging tiles
  if (use_lds) {
    int rc = reserve_lds(reinterpret_cast<const void*>(trsv_fused_kernel<true>), b_bytes);
    if (rc) return rc;
    trsv_fused_kernel<true><<<1, TRSV_T, b_bytes, s>>>(A, n, n, b, 0, 2);   // L y = b, then L^T x = y
  } else {
    trsv_fused_kernel<false><<<1, TRSV_T, 0, s>>>(A, n, n, b, 0, 2);
  }
  HYDRA_CHECK_LAUNCH("hydra_chol_solve(solve)");
  return HYDRA_OK;
}

static int chol_solve_blocked(const double* A, double* b, int n, hipStream_t s);

// The two triangular solves.  One workgroup streams L through one CU (n = 2000: 0.93 ms, 40 % of
// the whole dense solve and 15 % of a sample that reuses the factor); from 1024 unknowns on the
// panel-wise form takes over -- 256 x 256 diagonal blocks by that workgroup, the rest of L by many
// (n = 2000: 32 short launches).  Both callers of a given n take the same route, so a solve from a
// kept factor (hydra_chol_apply) repeats hydra_chol_solve bit for bit.
static constexpr int CHOL_SOLVE_BLOCKED_MIN_N = 1024;
static int chol_solve_any(const double* A, double* b, int n, hipStream_t s) {
  if (n >= CHOL_SOLVE_BLOCKED_MIN_N && n % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(b) % 16 == 0)
    return chol_solve_blocked(A, b, n, s);
  return chol_small_solve(A, b, n, s);
}

// the 32-wide steps of the columns [k, col_end) of an n-row array: one launch per step
// (chol_step_kernel) after the first block column's own two
// (HYDRA_CHOL_FUSED_STEP=0 in the environment selects the two-launch form, for comparison)
#ifndef HYDRA_CHOL_FUSED_STEP_DEFAULT
#define HYDRA_CHOL_FUSED_STEP_DEFAULT 1
#endif
static constexpr int CHOL_FUSED_STEP_MAX_ROWS = 8192;
static bool chol_fused_step() {
  static const bool on = [] {
    const char* e = getenv("HYDRA_CHOL_FUSED_STEP");
    return e ? atoi(e) != 0 : HYDRA_CHOL_FUSED_STEP_DEFAULT != 0;
  }();
  return on;
}
static void steps_fused(double* A, size_t ld, int n, int k, int col_end, int32_t* info_dev,
                        hipStream_t s) {
  dim3 tb(NB, NB);
  potrf_diag_kernel<<<1, tb, 0, s>>>(A, ld, n, k, info_dev);
  if (n - k - NB <= 0) return;                     // nothing below the first block
  trsm_panel_kernel<<<(n - k - NB + TRSM_T - 1) / TRSM_T, TRSM_T, 0, s>>>(A, ld, n, k);
  int last = -1;
  for (int kk = k; col_end - (kk + NB) > 0 && n - kk - NB > 0; kk += NB) {
    const int base = kk + NB;
    const int nct = (col_end - base + SY_T - 1) / SY_T, nrt = (n - base + SY_T - 1) / SY_T;
    chol_step_kernel<<<dim3(nct, nrt + 1), STEP_T, 0, s>>>(A, ld, n, kk, col_end, kk > k, info_dev);
    last = base;
  }
  if (last >= 0) chol_store_diag_kernel<<<1, STEP_T, 0, s>>>(A, ld, n, last, info_dev);
}

static int chol_small(double* A, double* b, int n, int32_t* info_dev, hipStream_t s) {
  if (chol_fused_step()) {
    steps_fused(A, (size_t)n, n, 0, n, info_dev, s);
  } else {
    dim3 tb(NB, NB);
    potrf_diag_kernel<<<1, tb, 0, s>>>(A, (size_t)n, n, 0, info_dev);
    for (int k = 0; k < n; k += NB) {
      int nb = std::min(NB, n - k);
      int rest = n - k - nb;
      if (rest > 0) {   // then the block is full (nb == NB)
        trsm_panel_kernel<<<(rest + TRSM_T - 1) / TRSM_T, TRSM_T, 0, s>>>(A, (size_t)n, n, k);
        int nblk = (rest + SY_T - 1) / SY_T;
        syrk_update_kernel<<<dim3(nblk, nblk), 256, 0, s>>>(A, (size_t)n, n, k, n, info_dev);  // + potrf of block k+NB
      }
    }
  }
  HYDRA_CHECK_LAUNCH("hydra_chol_solve(factor)");
  return b ? chol_solve_any(A, b, n, s) : HYDRA_OK;
}

// Columns are finished 256 at a time (32-wide steps capped at the sub-panel's last column), but the
// rest of the matrix is updated once per SUPER-panel of SPW columns with K = SPW: per 128 x 128 tile
// the 128 KB of C come in and go out once per SPW / 4 matrix instructions per wave instead of once
// per 64.  Timing builds of the update kernel with the C loads or stores removed showed what they
// cost at K = 256: 0.76 -> 0.85 / 0.84 of the matrix peak, 0.90 without both -- each workgroup alone
// nearly fills the matrix pipe, so a second one on the CU cannot hide the other's memory phases.
// Inside a super-panel the columns still to be factored take a rank-256 update from every finished
// sub-panel (the same kernel, a rectangle of at most SPW - 256 columns: under 2 % of the flops).
#ifndef HYDRA_CHOL_SPW
#define HYDRA_CHOL_SPW 2048
#endif
static constexpr int SPW = HYDRA_CHOL_SPW;
static_assert(SPW % PB == 0, "super-panel = whole 256-wide panels");

// Columns [k, col_end) of the n-row array A (leading dimension ld) by the 32-wide steps: the diagonal
// block at (k,k) is factored, the rows below it solved against it, and the columns still inside
// [k, col_end) updated after every step (the update that completes a 32-block factors it).
static void factor_panel(double* A, size_t ld, int n, int k, int col_end, int32_t* info_dev,
                         hipStream_t s, bool allow_fused = true) {
  // (every workgroup of the first tile column factors the next diagonal block for itself: with many
  // rows below that redundant single-wave work outweighs the launch it saves -- n = 100 000: 61 us
  // per step against 28 + 13 -- so tall panels keep the two-launch form)
  if (allow_fused && chol_fused_step() && n - k <= CHOL_FUSED_STEP_MAX_ROWS)
    return steps_fused(A, ld, n, k, col_end, info_dev, s);
  dim3 tb(NB, NB);
  potrf_diag_kernel<<<1, tb, 0, s>>>(A, ld, n, k, info_dev);
  for (int kk = k; kk < col_end; kk += NB) {
    const int nb = std::min(NB, n - kk), rest = n - kk - nb;
    if (rest <= 0) break;
    trsm_panel_kernel<<<(rest + TRSM_T - 1) / TRSM_T, TRSM_T, 0, s>>>(A, ld, n, kk);
    const int ncols = col_end - (kk + NB);
    if (ncols > 0)   // update the panel's remaining columns (+ potrf of the next 32-block)
      syrk_update_kernel<<<dim3((ncols + SY_T - 1) / SY_T, (rest + SY_T - 1) / SY_T), 256, 0, s>>>(
          A, ld, n, kk, col_end, info_dev);
  }
}

static int chol_factor_blocked(double* A, int n, int32_t* info_dev, hipStream_t s) {
  for (int k0 = 0; k0 < n; k0 += SPW) {
    const int sp_end = std::min(k0 + SPW, n);
    for (int k = k0; k < sp_end; k += PB) {
      const int pb = std::min(PB, n - k), col_end = k + pb;
      factor_panel(A, (size_t)n, n, k, col_end, info_dev, s);
      const int wrest = sp_end - col_end;        // columns of the super-panel still to be factored
      if (wrest > 0) {                            // then this sub-panel is full (pb == PB)
        const int mrows = n - col_end;
        const int ntr = (mrows + CT - 1) / CT, ntc = (wrest + CT - 1) / CT;
        const long long grid = (long long)ntr * ntc;
        if (grid > 2147483647LL) return fail(HYDRA_ERR_ARG, "hydra_chol_solve: n=%d is too large", n);
        const double* P = A + (size_t)col_end * n + k;
        chol_syrk_mfma_kernel<<<(unsigned)grid, 256, 0, s>>>(A + (size_t)col_end * n + col_end, n, P, n,
                                                             P, n, mrows, wrest, 0, ntr, 0, ntc, NKC);
      }
    }
    const int rest = n - sp_end;
    if (rest > 0) {   // then the super-panel is full
      const int ntr = (rest + CT - 1) / CT, nsr = (ntr + SUP - 1) / SUP;
      const long long nsup = (long long)nsr * (nsr + 1) / 2;
      const long long grid = (nsup + 7) / 8 * 8 * SUP * SUP;
      if (grid > 2147483647LL) return fail(HYDRA_ERR_ARG, "hydra_chol_solve: n=%d is too large", n);
      const double* P = A + (size_t)sp_end * n + k0;              // the finished super-panel, in place
      chol_syrk_mfma_kernel<<<(unsigned)grid, 256, 0, s>>>(A + (size_t)sp_end * n + sp_end, n, P, n,
                                                           P, n, rest, rest, 0, ntr, (int)nsup, ntr,
                                                           SPW / KC);
    }
  }
  HYDRA_CHECK_LAUNCH("hydra_chol_solve(blocked factor)");
  return HYDRA_OK;
}

// sweeps: bit 0 = L y = b, bit 1 = L^T x = y; A is an n x n lower triangle with leading dimension ld
static int chol_sweeps_blocked(const double* A, size_t ld, double* b, int n, int sweeps, hipStream_t s) {
  const int np = (n + PB - 1) / PB;
  if (sweeps & 1)
    for (int p = 0; p < np; ++p) {
      const int k = p * PB;
      // launch p: solve panel p (after taking panel p-1's solution out of it), and take panel p-1's
      // solution out of the rows beyond panel p
      const int rest = p > 0 ? std::max(0, n - k - PB) : 0;
      chol_fwd_step_kernel<<<1 + std::min((rest + 63) / 64, 2048), SWT, 0, s>>>(A, ld, n, k, b);
    }
  if (sweeps & 2)
    for (int p = np - 1; p >= 0; --p) {
      const int k = p * PB;
      const int left = p < np - 1 ? k / PB : 0;      // 256-column groups left of panel p
      chol_bwd_step_kernel<<<1 + left, SWT, 0, s>>>(A, ld, n, k, b);
    }
  HYDRA_CHECK_LAUNCH("hydra_chol_solve(blocked solve)");
  return HYDRA_OK;
}

static int chol_solve_blocked(const double* A, double* b, int n, hipStream_t s) {
  return chol_sweeps_blocked(A, (size_t)n, b, n, 3, s);
}

static bool chol_takes_blocked(const double* A, const double* b, int n) {
  return n >= CHOL_BLOCKED_MIN_N && n % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(b) % 16 == 0;
}

extern "C" {

int hydra_chol_apply(const double* L, double* b, int n, void* stream) {
  HYDRA_CHECK_ARG(L && b && n > 0, "hydra_chol_apply: bad arguments");
  hipStream_t s = as_stream(stream);
  return chol_solve_any(L, b, n, s);
}

int hydra_symv_upper(const double* A, long long ld, int nloc, int n, const long long* rows,
                     const double* diag, const double* x, double* y, void* stream) {
  HYDRA_CHECK_ARG(A && rows && diag && x && y && nloc >= 0 && n > 0 && ld >= n,
                  "hydra_symv_upper: bad arguments (nloc=%d n=%d)", nloc, n);
  if (nloc == 0) return HYDRA_OK;
  hipStream_t s = as_stream(stream);
  symv_upper_rows_kernel<<<std::min((nloc + 3) / 4, 8192), 256, 0, s>>>(A, (size_t)ld, nloc, n, rows, diag, x, y);
  symv_upper_cols_kernel<<<(n + 255) / 256, 256, 0, s>>>(A, (size_t)ld, nloc, n, rows, x, y);
  HYDRA_CHECK_LAUNCH("hydra_symv_upper");
  return HYDRA_OK;
}

int hydra_tri_inverse(const double* L, long long ld, int n, double* X, long long ldx, void* stream) {
  HYDRA_CHECK_ARG(L && X && n > 0 && ld >= n && ldx >= n, "hydra_tri_inverse: bad arguments (n=%d)", n);
  HYDRA_CHECK_ARG(L != X, "hydra_tri_inverse: X must not be L");
  tri_inverse_kernel<<<(n + NB - 1) / NB, NB * NB, 0, as_stream(stream)>>>(L, (size_t)ld, n, X, (size_t)ldx);
  HYDRA_CHECK_LAUNCH("hydra_tri_inverse");
  return HYDRA_OK;
}

int hydra_chol_solve(double* A, double* b, int n, int32_t* info_dev, void* stream) {
  HYDRA_CHECK_ARG(A && info_dev && n > 0, "hydra_chol_solve: bad arguments");
  // (odd n: rows are not 16-byte aligned for the LDS-DMA of the blocked path -- callers pad, as
  // hydra_amd.linear_solver.solve_dense does)
  if (chol_takes_blocked(A, b, n)) return hydra_chol_solve_blocked(A, b, n, info_dev, stream);
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(info_dev, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_chol_solve: %s", hipGetErrorString(e));
  return chol_small(A, b, n, info_dev, s);
}

int hydra_chol_solve_blocked(double* A, double* b, int n, int32_t* info_dev, void* stream) {
  HYDRA_CHECK_ARG(A && info_dev && n > 0, "hydra_chol_solve_blocked: bad arguments");
  HYDRA_CHECK_ARG(reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(b) % 16 == 0 &&
                      n % 2 == 0,
                  "hydra_chol_solve_blocked: A and b must be 16-byte aligned and n even (n=%d)", n);
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(info_dev, 0, sizeof(int32_t), s);
  if (e != hipSuccess)
    return fail(HYDRA_ERR_HIP, "hydra_chol_solve_blocked: %s", hipGetErrorString(e));
  int rc = chol_factor_blocked(A, n, info_dev, s);
  if (rc || !b) return rc;
  return chol_solve_blocked(A, b, n, s);
}

int hydra_chol_panel(double* P, long long ld, int nrows, int ncols, int32_t* info_dev, void* stream) {
  HYDRA_CHECK_ARG(P && info_dev && ncols > 0 && nrows >= ncols && ld >= ncols,
                  "hydra_chol_panel: bad arguments (nrows=%d ncols=%d ld=%lld)", nrows, ncols, ld);
  HYDRA_CHECK_ARG(nrows == ncols || ncols % NB == 0,
                  "hydra_chol_panel: a panel with rows below it must be whole 32-column steps");
  // (always the two-launch steps: the ranks of a row-sharded factorisation factor the diagonal block
  // redundantly on panels of different heights, and a choice by height would give them factors that
  // differ in the last bit)
  factor_panel(P, (size_t)ld, nrows, 0, ncols, info_dev, as_stream(stream), false);
  HYDRA_CHECK_LAUNCH("hydra_chol_panel");
  return HYDRA_OK;
}

int hydra_trsv_block(const double* L, long long ld, int n, double* b, int transpose, void* stream) {
  HYDRA_CHECK_ARG(L && b && n > 0 && ld >= n, "hydra_trsv_block: bad arguments (n=%d)", n);
  // panel-wise from 1024 unknowns on (the single workgroup streams L through one CU); needs rows and
  // vector aligned for the 16-byte loads of the update kernel
  if (n >= CHOL_SOLVE_BLOCKED_MIN_N && ld % 2 == 0 && reinterpret_cast<uintptr_t>(L) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(b) % 16 == 0)
    return chol_sweeps_blocked(L, (size_t)ld, b, n, transpose ? 2 : 1, as_stream(stream));
  const size_t b_bytes = (size_t)n * sizeof(double);
  HYDRA_CHECK_ARG(b_bytes <= 96 * 1024, "hydra_trsv_block: n=%d exceeds the LDS-resident vector (12288)", n);
  const int sw = transpose ? 1 : 0;
  if (int rc = reserve_lds(reinterpret_cast<const void*>(trsv_fused_kernel<true>), b_bytes)) return rc;
  trsv_fused_kernel<true><<<1, TRSV_T, b_bytes, as_stream(stream)>>>(L, (int)ld, n, b, sw, sw + 1);
  HYDRA_CHECK_LAUNCH("hydra_trsv_block");
  return HYDRA_OK;
}

int hydra_gemv_sub(const double* M, long long ld, int nrows, int ncols, const double* x, double* y,
                   int transpose, void* stream) {
  HYDRA_CHECK_ARG(M && x && y && nrows > 0 && ncols > 0 && ld >= ncols,
                  "hydra_gemv_sub: bad arguments (%d x %d, ld=%lld)", nrows, ncols, ld);
  if (transpose)
    cols_update_kernel<<<(ncols + 255) / 256, 256, 0, as_stream(stream)>>>(M, (size_t)ld, nrows, ncols, x, y);
  else
    rows_update_kernel<<<std::min((nrows + 3) / 4, 8192), 256, 0, as_stream(stream)>>>(M, (size_t)ld, nrows,
                                                                                   ncols, x, y);
  HYDRA_CHECK_LAUNCH("hydra_gemv_sub");
  return HYDRA_OK;
}

// (measured at n = 65 536 on one rank: 1.72 s with the patch order against 1.60 s in plain row-major
// order -- the updates of one column block are 16 tiles wide, two patches, and the padded grids
// cost more than the strips they save; kept behind this switch)
#ifndef HYDRA_CHOL_RECT_PATCHES
#define HYDRA_CHOL_RECT_PATCHES 0
#endif
int hydra_chol_update_rect(double* C, long long ldc, const double* PA, long long lda,
                           const double* PB, long long ldb, int mrows, int ncols, int diag, int kdim,
                           void* stream) {
  HYDRA_CHECK_ARG(C && PA && PB && mrows > 0 && ncols > 0 && kdim >= KC && kdim % KC == 0 &&
                      ldc >= ncols && lda >= kdim && ldb >= kdim,
                  "hydra_chol_update_rect: bad arguments (mrows=%d ncols=%d kdim=%d)", mrows, ncols,
                  kdim);
  HYDRA_CHECK_ARG(reinterpret_cast<uintptr_t>(PA) % 16 == 0 && reinterpret_cast<uintptr_t>(PB) % 16 == 0 &&
                      lda % 2 == 0 && ldb % 2 == 0,
                  "hydra_chol_update_rect: operand rows must be 16-byte aligned");
  HYDRA_CHECK_ARG((unsigned long long)lda * 8ull * (unsigned)CT < 4294967296ull &&
                      (unsigned long long)ldb * 8ull * (unsigned)CT < 4294967296ull,
                  "hydra_chol_update_rect: leading dimension too large");
  const int ntr = (mrows + CT - 1) / CT, ntc = (ncols + CT - 1) / CT;
  long long grid = (long long)ntr * ntc;
  int order = 0;
  if (HYDRA_CHOL_RECT_PATCHES && grid >= 4 * SUP * SUP) {          // enough tiles for patches to pay
    const long long npatch = (long long)((ntr + SUP - 1) / SUP) * ((ntc + SUP - 1) / SUP);
    grid = (npatch + 7) / 8 * 8 * SUP * SUP;
    order = -1;
  }
  HYDRA_CHECK_ARG(grid <= 2147483647LL, "hydra_chol_update_rect: too many tiles");
  chol_syrk_mfma_kernel<<<(unsigned)grid, 256, 0, as_stream(stream)>>>(C, ldc, PA, lda, PB, ldb, mrows,
                                                                       ncols, diag, ntr, order, ntc,
                                                                       kdim / KC);
  HYDRA_CHECK_LAUNCH("hydra_chol_update_rect");
  return HYDRA_OK;
}

}  // extern "C"
