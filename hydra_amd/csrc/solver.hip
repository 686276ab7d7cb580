// Linear solvers for the GCR system on gfx950:
//   * dense symmetric matrix-vector product (HBM/L2 bound: 8 N^2 bytes per product);
//   * the reference's conjugate-gradient iteration (hydra/linear_solver.py:378-509) as one
//     fused step kernel (both dot products, the three vector updates and the convergence
//     test in a single launch, state kept on the device);
//   * a blocked Cholesky solve as the dense fallback (example.py:944).
#include <algorithm>
#include <mutex>

#include "common.h"

namespace hydra {

// ---------------------------------------------------------------- dense gemv

__global__ void __launch_bounds__(256)
gemv_kernel(const double* __restrict__ A, int n, const double* __restrict__ x,
            double* __restrict__ y) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < n; r += gridDim.x * 4) {
    const double* ap = A + (size_t)r * n;
    double s = 0.0;
    // 8 loads of the row in flight per lane (clamped, unconditional): with one per trip an
    // iteration of the SH head's CG (n = 1681, matrix in L2) spent 10.4 us here.  Same order of
    // summation.
    for (int c0 = lane; c0 < n; c0 += 512) {
      double av[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = min(c0 + 64 * u, n - 1);
        av[u] = ap[c];
        xv[u] = x[c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + 64 * u < n) s += av[u] * xv[u];
    }
    s = wave_sum(s);
    if (lane == 0) y[r] = s;
  }
}

// y[q] (+)= A[q] x[q] for a stack of nb independent (nrows,n) matrices (blockIdx.y = q): the
// per-frequency likelihood term of the cosmological-field operator (cosmo_sampler.py:219-221),
// and (nb = 1, nrows < n) a rank's row block of a matrix sharded by rows over GPUs.
// HBM-bound (8 n^2 bytes per matrix): a wave takes two rows at a time, 16-byte loads when the
// rows are 16-byte aligned (n even), eight loads in flight per lane.
template <bool VEC2>
__global__ void __launch_bounds__(256)
gemv_batched_kernel(const double* __restrict__ A, int nrows, int n, const double* __restrict__ x,
                    double* __restrict__ y, int accumulate) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* Aq = A + (size_t)blockIdx.y * nrows * n;
  const double* xq = x + (size_t)blockIdx.y * n;
  double* yq = y + (size_t)blockIdx.y * nrows;
  for (int r = 2 * (blockIdx.x * 4 + wave); r < nrows; r += 2 * gridDim.x * 4) {
    const bool two = r + 1 < nrows;
    const double* a0 = Aq + (size_t)r * n;
    const double* a1 = Aq + (size_t)(two ? r + 1 : r) * n;
    double s0 = 0.0, s1 = 0.0;
    if (VEC2) {
      const f64x2* v0 = reinterpret_cast<const f64x2*>(a0);
      const f64x2* v1 = reinterpret_cast<const f64x2*>(a1);
      const f64x2* xv = reinterpret_cast<const f64x2*>(xq);
      const int nh = n / 2;
      int c = lane;
      for (; c + 192 < nh; c += 256) {
        f64x2 p[4], q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          p[u] = v0[c + 64 * u];
          q[u] = v1[c + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f64x2 xx = xv[c + 64 * u];
          s0 += p[u].x * xx.x + p[u].y * xx.y;
          s1 += q[u].x * xx.x + q[u].y * xx.y;
        }
      }
      for (; c < nh; c += 64) {
        f64x2 xx = xv[c], p = v0[c], q = v1[c];
        s0 += p.x * xx.x + p.y * xx.y;
        s1 += q.x * xx.x + q.y * xx.y;
      }
    } else {
      for (int c = lane; c < n; c += 64) {
        double xx = xq[c];
        s0 += a0[c] * xx;
        s1 += a1[c] * xx;
      }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
      yq[r] = accumulate ? yq[r] + s0 : s0;
      if (two) yq[r + 1] = accumulate ? yq[r + 1] + s1 : s1;
    }
  }
}

// y = A x for SYMMETRIC A stored in full, reading only the blocks on and below the diagonal
// (4 n^2 instead of 8 n^2 bytes): the workgroup of block (bi, bj), bi >= bj, stages the 64 x 64
// block B in LDS and forms both B x_j (rows of block bi) and, off the diagonal, B^T x_i (rows of
// block bj).  Every (block column k, row) pair receives exactly one such piece, stored in
// part[k][row]; a second launch sums the pieces in k order -- no atomics, reproducible sums.
// Loads are unconditional from clamped addresses (a conditional load costs a branch and a wait:
// n = 16 384 went from 4.2 to 5.2 TB/s); rows and columns >= n meet x = 0 or are not stored.
// (A variant walking strips of 16 blocks of one block row, accumulating B x_j in registers, was
// measured at n = 32 768 .. 100 000: 2 % faster at 100 000, 20 % slower at 32 768 -- not kept.)
constexpr int SV_T = 64;
constexpr int SVB_LD = SV_T + 2;   // LDS row stride (doubles): even, so 16-byte stores stay aligned

template <bool VEC2>
__global__ void __launch_bounds__(256)
symv_blocks_kernel(const double* __restrict__ A, int n, int nbk, const double* __restrict__ x,
                   double* __restrict__ part) {
  __shared__ __align__(16) double sBk[SV_T * SVB_LD];
  __shared__ double sx[2][SV_T];        // x_j, x_i
  __shared__ double so[2][2][SV_T];     // [product][half][row or column]
  // linear block index -> (bi, bj), bj <= bi
  int t = blockIdx.x;
  int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  int bj = t - bi * (bi + 1) / 2;
  const int q = blockIdx.y;
  const double* Aq = A + (size_t)q * n * n;
  const double* xq = x + (size_t)q * n;
  double* pq = part + (size_t)q * nbk * n;
  const int tid = threadIdx.x;
  const int r0 = bi * SV_T, c0 = bj * SV_T;
  if (VEC2) {   // n even: 16-byte loads, 32 per row, 8 rows per trip
    const int cp = tid % 32, rr = tid / 32;
    f64x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // unconditional, clamped (rows / columns >= n meet x = 0 or are not stored); n even
      int r = min(r0 + rr + 8 * k, n - 1), c = min(c0 + 2 * cp, n - 2);
      v[k] = *reinterpret_cast<const f64x2*>(Aq + (size_t)r * n + c);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<f64x2*>(&sBk[(rr + 8 * k) * SVB_LD + 2 * cp]) = v[k];
  } else {
    const int cc = tid % 64, rr = tid / 64;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int r = min(r0 + rr + 4 * k, n - 1), c = min(c0 + cc, n - 1);
      v[k] = Aq[(size_t)r * n + c];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) sBk[(rr + 4 * k) * SVB_LD + cc] = v[k];
  }
  if (tid < SV_T) sx[0][tid] = (c0 + tid < n) ? xq[c0 + tid] : 0.0;
  else if (tid < 2 * SV_T) sx[1][tid - SV_T] = (r0 + tid - SV_T < n) ? xq[r0 + tid - SV_T] : 0.0;
  __syncthreads();
  {
    const int which = tid >> 7, half = (tid >> 6) & 1, l = tid & 63;
    double s0 = 0.0, s1 = 0.0;
    if (which == 0) {   // (B x_j)[l] over columns [32 half, 32 half + 32)
      const double* row = sBk + l * SVB_LD + 32 * half;
      const double* xv = sx[0] + 32 * half;
#pragma unroll
      for (int c = 0; c < 32; c += 2) {
        s0 += row[c] * xv[c];
        s1 += row[c + 1] * xv[c + 1];
      }
    } else {            // (B^T x_i)[l] over rows [32 half, 32 half + 32)
      const double* col = sBk + (32 * half) * SVB_LD + l;
      const double* xv = sx[1] + 32 * half;
#pragma unroll
      for (int r = 0; r < 32; r += 2) {
        s0 += col[r * SVB_LD] * xv[r];
        s1 += col[(r + 1) * SVB_LD] * xv[r + 1];
      }
    }
    so[which][half][l] = s0 + s1;
  }
  __syncthreads();
  if (tid < SV_T) {
    if (r0 + tid < n) pq[(size_t)bj * n + r0 + tid] = so[0][0][tid] + so[0][1][tid];
  } else if (tid < 2 * SV_T && bi != bj) {
    int l = tid - SV_T;
    if (c0 + l < n) pq[(size_t)bi * n + c0 + l] = so[1][0][l] + so[1][1][l];
  }
}

// grid (ceil(n / 64), batch): lane <-> row of block b, the 4 waves take the pieces k(0), k(1), ..
// round robin (8 loads in flight each), combined through LDS in wave order
__global__ void __launch_bounds__(256)
symv_reduce_kernel(const double* __restrict__ part, int n, int nbk, double* __restrict__ y,
                   int accumulate) {
  __shared__ double sred[4][SV_T];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = min(b * SV_T + lane, n - 1);
  const double* pq = part + (size_t)blockIdx.y * nbk * n + i;
  const int cnt = nbk;
  auto kof = [&](int t) { return t; };
  double s0 = 0.0, s1 = 0.0;
  int t = wave;
  for (; t + 28 < cnt; t += 32) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = pq[(size_t)kof(t + 4 * u) * n];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0 += a[u];
      s1 += a[u + 1];
    }
  }
  for (; t < cnt; t += 4) s0 += pq[(size_t)kof(t) * n];
  sred[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && b * SV_T + lane < n) {
    double v = (sred[0][lane] + sred[1][lane]) + (sred[2][lane] + sred[3][lane]);
    double* yq = y + (size_t)blockIdx.y * n;
    yq[i] = accumulate ? yq[i] + v : v;
  }
}

// ---------------------------------------------------------------- packed lower triangle
// The Gram of a rank's rows is exactly symmetric, so the sum over GPUs (comm.Reduce of the full
// square at ptsrc_sampler.py:86) only needs the lower triangle on the wire: 4 n (n+1) bytes
// instead of 8 n^2.  packed[i (i+1) / 2 + j] = A[i][j], j <= i.  Unpacking mirrors through LDS
// (64 x 64 tiles) so that both triangles are written along rows; the result is symmetric bit for
// bit whatever order the collective summed in.
__global__ void __launch_bounds__(256)
pack_lower_kernel(const double* __restrict__ A, int n, double* __restrict__ packed) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const double* row = A + (size_t)i * n;
    double* dst = packed + (size_t)i * (i + 1) / 2;
    for (int j = threadIdx.x; j <= i; j += 256) dst[j] = row[j];
  }
}

__global__ void __launch_bounds__(256)
unpack_lower_kernel(const double* __restrict__ packed, int n, double* __restrict__ A) {
  __shared__ double tile[64][65];
  const long long t = blockIdx.x;
  int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((long long)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while ((long long)bi * (bi + 1) / 2 > t) --bi;
  const int bj = (int)(t - (long long)bi * (bi + 1) / 2);
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int gi = bi * 64 + r, gj = bj * 64 + c;
    double v = 0.0;
    if (gi < n && gj <= gi) {
      v = packed[(size_t)gi * (gi + 1) / 2 + gj];
      A[(size_t)gi * n + gj] = v;
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;             // entry (r, c) of the transposed tile
    const int gi = bj * 64 + r, gj = bi * 64 + c;
    if (gi < n && gj < n && gj > gi) A[(size_t)gi * n + gj] = tile[c][r];
  }
}

// ---------------------------------------------------------------- symv without a workspace
// hydra_symv_static: the pieces of symv_blocks_kernel go to a file-scope device array instead of
// a caller workspace, for callers that cannot allocate (HIP graph capture, serving loops).
// Every GPU the library runs on holds its own copy (8 MiB) until the process ends; host code
// never uses the array's name as an address -- only the per-device pointer obtained below.
constexpr int SYMV_STATIC_MAX_N = 8192;
constexpr size_t SYMV_STATIC_DOUBLES = (size_t)(SYMV_STATIC_MAX_N / SV_T) * SYMV_STATIC_MAX_N;
static __device__ __align__(256) double scratch[SYMV_STATIC_DOUBLES];

constexpr int SCRATCH_MAX_DEVICES = 64;
static double* g_scratch_ptr[SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// device address of `scratch` on the CURRENT device (looked up once per device)
static int scratch_ptr_for_current_device(double** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return fail(HYDRA_ERR_HIP, "hydra_symv_static: hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= SCRATCH_MAX_DEVICES)
    return fail(HYDRA_ERR_ARG, "hydra_symv_static: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    double* p = nullptr;
    e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
      return fail(HYDRA_ERR_HIP, "hydra_symv_static: hipGetSymbolAddress: %s",
                  hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return HYDRA_OK;
}

// ---------------------------------------------------------------- CG

constexpr int CG_THREADS = 1024;

// block-wide sum / max over CG_THREADS threads, result broadcast to all threads
__device__ __forceinline__ double block_sum(double v, double* s_red) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < CG_THREADS / 64; ++w) t += s_red[w];
  return t;
}
__device__ __forceinline__ double block_max(double v, double* s_red) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < CG_THREADS / 64; ++w) t = fmax(t, s_red[w]);
  return t;
}

// r = b - Ax ; p = r ; rr = r.r ; convergence test on r (linear_solver.py:446-464)
__global__ void __launch_bounds__(CG_THREADS)
cg_init_kernel(int n, const double* __restrict__ b, const double* __restrict__ Ax,
               double* __restrict__ r, double* __restrict__ p, double* __restrict__ st,
               double tol, int use_norm) {
  __shared__ double s_red[CG_THREADS / 64];
  double rr = 0.0, mx = 0.0;
  for (int i = threadIdx.x; i < n; i += CG_THREADS) {
    double ri = b[i] - Ax[i];
    r[i] = ri;
    p[i] = ri;
    rr += ri * ri;
    mx = fmax(mx, fabs(ri));
  }
  rr = block_sum(rr, s_red);
  mx = block_max(mx, s_red);
  if (threadIdx.x == 0) {
    double res = use_norm ? sqrt(rr) : mx;
    st[0] = rr;
    st[1] = (res < tol) ? 1.0 : 0.0;
    st[2] = 0.0;
    st[3] = res;
  }
}

// One CG iteration given Ap (linear_solver.py:479-492) followed by the convergence test
// the reference performs at the top of the NEXT iteration (:455-464).
__global__ void __launch_bounds__(CG_THREADS)
cg_step_kernel(int n, const double* __restrict__ Ap, double* __restrict__ x,
               double* __restrict__ r, double* __restrict__ p, double* __restrict__ st, double tol,
               int use_norm) {
  __shared__ double s_red[CG_THREADS / 64];
  __shared__ double s_red2[CG_THREADS / 64];
  if (st[1] != 0.0) return;  // converged earlier: this enqueued step is a no-op
  // n < CGL_MIN_N = 8192: every thread keeps its (at most 8) entries of the four vectors in
  // registers across the three phases -- one round trip to memory instead of three (the step is
  // pure latency: 6.1 us per iteration of the SH head's CG with the vectors re-read per phase).
  // Same per-thread order of summation as before.
  constexpr int MAXE = 8;
  double pv[MAXE], av[MAXE], xv[MAXE], rv[MAXE];
  const double rr = st[0];
  const int ne = (n + CG_THREADS - 1) / CG_THREADS;       // uniform: rounds that hold entries
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    pv[e] = av[e] = xv[e] = rv[e] = 0.0;
    if (e < ne) {
      const int ic = min(threadIdx.x + e * CG_THREADS, n - 1);
      pv[e] = p[ic]; av[e] = Ap[ic]; xv[e] = x[ic]; rv[e] = r[ic];
    }
  }
  double pap = 0.0;
#pragma unroll
  for (int e = 0; e < MAXE; ++e)
    if (threadIdx.x + e * CG_THREADS < n) pap += pv[e] * av[e];
  pap = block_sum(pap, s_red);
  const double alpha = rr / pap;
  double rr_new = 0.0, mx = 0.0;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = threadIdx.x + e * CG_THREADS;
    if (i < n) {
      x[i] = xv[e] + alpha * pv[e];
      rv[e] = rv[e] - alpha * av[e];
      r[i] = rv[e];
      rr_new += rv[e] * rv[e];
      mx = fmax(mx, fabs(rv[e]));
    }
  }
  {                                    // both reductions behind one pair of barriers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    rr_new = wave_sum(rr_new);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) { s_red[wave] = rr_new; s_red2[wave] = mx; }
    __syncthreads();
    rr_new = 0.0; mx = 0.0;
    for (int w = 0; w < CG_THREADS / 64; ++w) { rr_new += s_red[w]; mx = fmax(mx, s_red2[w]); }
  }
  const double beta = rr_new / rr;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = threadIdx.x + e * CG_THREADS;
    if (i < n) p[i] = rv[e] + beta * pv[e];
  }
  if (threadIdx.x == 0) {
    double res = use_norm ? sqrt(rr_new) : mx;
    st[0] = rr_new;
    st[1] = (res < tol) ? 1.0 : 0.0;
    st[2] = st[2] + 1.0;
    st[3] = res;
  }
}

// The same iteration for long vectors (the cosmological-field head solves for 10^4-10^5 unknowns
// with a cheap operator): one workgroup streams x, r, p, Ap through a single CU at ~30 GB/s, so the
// step is split over up to CGL_MAXB workgroups and three launches, the dot products passing through
// per-workgroup partial sums in st[8..] (summed in index order by every workgroup: deterministic).
constexpr int CGL_T = 256;
constexpr int CGL_MAXB = 256;
constexpr int CG_STATE_LEN = 8 + 3 * CGL_MAXB;
constexpr int CGL_MIN_N = 8192;

__device__ __forceinline__ double cgl_block_sum(double v, double* s_red) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ double cgl_block_max(double v, double* s_red) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
}
// sum (or max) of the nblk per-workgroup partials, same order in every workgroup
__device__ __forceinline__ double cgl_gather(const double* part, int nblk, bool is_max, double* s_red) {
  double v = (threadIdx.x < nblk) ? part[threadIdx.x] : 0.0;   // nblk <= CGL_MAXB == CGL_T
  return is_max ? cgl_block_max(v, s_red) : cgl_block_sum(v, s_red);
}

__global__ void __launch_bounds__(CGL_T)
cgl_dot_kernel(int n, const double* __restrict__ Ap, const double* __restrict__ p,
               double* __restrict__ st) {
  __shared__ double s_red[4];
  if (st[1] != 0.0) return;
  double pap = 0.0;
  for (int i = blockIdx.x * CGL_T + threadIdx.x; i < n; i += gridDim.x * CGL_T) pap += p[i] * Ap[i];
  pap = cgl_block_sum(pap, s_red);
  if (threadIdx.x == 0) st[8 + blockIdx.x] = pap;
}

__global__ void __launch_bounds__(CGL_T)
cgl_update_kernel(int n, const double* __restrict__ Ap, const double* __restrict__ p,
                  double* __restrict__ x, double* __restrict__ r, double* __restrict__ st) {
  __shared__ double s_red[4];
  const double done = st[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st[4] = st[0];   // rr and the done flag as they were when this iteration started: the last
    st[5] = done;    // launch of the step reads these while workgroup 0 rewrites st[0..3]
  }
  if (done != 0.0) return;
  double pap = cgl_gather(st + 8, gridDim.x, false, s_red);
  double alpha = st[0] / pap;
  double rr_new = 0.0, mx = 0.0;
  for (int i = blockIdx.x * CGL_T + threadIdx.x; i < n; i += gridDim.x * CGL_T) {
    x[i] = x[i] + alpha * p[i];
    double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    rr_new += ri * ri;
    mx = fmax(mx, fabs(ri));
  }
  rr_new = cgl_block_sum(rr_new, s_red);
  mx = cgl_block_max(mx, s_red);
  if (threadIdx.x == 0) {
    st[8 + CGL_MAXB + blockIdx.x] = rr_new;
    st[8 + 2 * CGL_MAXB + blockIdx.x] = mx;
  }
}

__global__ void __launch_bounds__(CGL_T)
cgl_dir_kernel(int n, const double* __restrict__ r, double* __restrict__ p,
               double* __restrict__ st, double tol, int use_norm) {
  __shared__ double s_red[4];
  if (st[5] != 0.0) return;
  double rr_new = cgl_gather(st + 8 + CGL_MAXB, gridDim.x, false, s_red);
  double mx = cgl_gather(st + 8 + 2 * CGL_MAXB, gridDim.x, true, s_red);
  double beta = rr_new / st[4];
  for (int i = blockIdx.x * CGL_T + threadIdx.x; i < n; i += gridDim.x * CGL_T)
    p[i] = r[i] + beta * p[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double res = use_norm ? sqrt(rr_new) : mx;
    st[0] = rr_new;
    st[1] = (res < tol) ? 1.0 : 0.0;
    st[2] = st[2] + 1.0;
    st[3] = res;
  }
}

}  // namespace hydra

using namespace hydra;

extern "C" {

int hydra_gemv(const double* A, int n, const double* x, double* y, void* stream) {
  HYDRA_CHECK_ARG(A && x && y && n > 0, "hydra_gemv: bad arguments (n=%d)", n);
  int grid = std::min((n + 3) / 4, 4096);
  gemv_kernel<<<grid, 256, 0, as_stream(stream)>>>(A, n, x, y);
  HYDRA_CHECK_LAUNCH("hydra_gemv");
  return HYDRA_OK;
}

int hydra_gemv_batched(const double* A, int n, int nb, const double* x, double* y, int accumulate,
                       void* stream) {
  HYDRA_CHECK_ARG(A && x && y && n > 0 && nb > 0 && nb <= 65535,
                  "hydra_gemv_batched: bad arguments (n=%d, nb=%d)", n, nb);
  dim3 grid(std::min((n + 7) / 8, 1024), nb);
  bool vec2 = n % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 &&
              reinterpret_cast<uintptr_t>(x) % 16 == 0;
  if (vec2)
    gemv_batched_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(A, n, n, x, y, accumulate);
  else
    gemv_batched_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(A, n, n, x, y, accumulate);
  HYDRA_CHECK_LAUNCH("hydra_gemv_batched");
  return HYDRA_OK;
}

int hydra_gemv_rows(const double* A, int nrows, int ncols, const double* x, double* y,
                    int accumulate, void* stream) {
  HYDRA_CHECK_ARG(A && x && y && nrows > 0 && ncols > 0, "hydra_gemv_rows: bad arguments (%d x %d)",
                  nrows, ncols);
  bool vec2 = ncols % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 &&
              reinterpret_cast<uintptr_t>(x) % 16 == 0;
  dim3 grid((unsigned)std::min(2048, (nrows + 7) / 8), 1);
  if (vec2)
    gemv_batched_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(A, nrows, ncols, x, y, accumulate);
  else
    gemv_batched_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(A, nrows, ncols, x, y, accumulate);
  HYDRA_CHECK_LAUNCH("hydra_gemv_rows");
  return HYDRA_OK;
}

size_t hydra_symv_workspace_bytes(int n, int nb) {
  if (n <= 0 || nb <= 0) return 0;
  size_t nbk = ((size_t)n + SV_T - 1) / SV_T;
  return (size_t)nb * nbk * n * sizeof(double);
}

int hydra_symv_batched(const double* A, int n, int nb, const double* x, double* y, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream) {
  HYDRA_CHECK_ARG(A && x && y && n > 0 && nb > 0 && nb <= 65535,
                  "hydra_symv_batched: bad arguments (n=%d, nb=%d)", n, nb);
  size_t need = hydra_symv_workspace_bytes(n, nb);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_symv_batched: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int nbk = (n + SV_T - 1) / SV_T;
  // strip length: long for large matrices (page and TLB locality, fewer pieces), short for small
  // ones (enough workgroups)
  long long nwg = (long long)nbk * (nbk + 1) / 2;
  HYDRA_CHECK_ARG(nwg <= 2147483647LL, "hydra_symv_batched: n=%d is too large", n);
  double* part = static_cast<double*>(workspace);
  hipStream_t s = as_stream(stream);
  bool vec2 = n % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0;
  dim3 grid((unsigned)nwg, nb);
  if (vec2) symv_blocks_kernel<true><<<grid, 256, 0, s>>>(A, n, nbk, x, part);
  else      symv_blocks_kernel<false><<<grid, 256, 0, s>>>(A, n, nbk, x, part);
  symv_reduce_kernel<<<dim3(nbk, nb), 256, 0, s>>>(part, n, nbk, y, accumulate);
  HYDRA_CHECK_LAUNCH("hydra_symv_batched");
  return HYDRA_OK;
}

int hydra_pack_lower(const double* A, int n, double* packed, void* stream) {
  HYDRA_CHECK_ARG(A && packed && n > 0, "hydra_pack_lower: bad arguments (n=%d)", n);
  pack_lower_kernel<<<std::min(n, 65535), 256, 0, as_stream(stream)>>>(A, n, packed);
  HYDRA_CHECK_LAUNCH("hydra_pack_lower");
  return HYDRA_OK;
}

int hydra_unpack_lower(const double* packed, int n, double* A, void* stream) {
  HYDRA_CHECK_ARG(A && packed && n > 0, "hydra_unpack_lower: bad arguments (n=%d)", n);
  long long nb = (n + 63) / 64, ntile = nb * (nb + 1) / 2;
  HYDRA_CHECK_ARG(ntile <= 2147483647LL, "hydra_unpack_lower: n=%d is too large", n);
  unpack_lower_kernel<<<(unsigned)ntile, 256, 0, as_stream(stream)>>>(packed, n, A);
  HYDRA_CHECK_LAUNCH("hydra_unpack_lower");
  return HYDRA_OK;
}

// Also the set-up call: looks up the scratch array's address on the CURRENT device (a symbol lookup,
// possibly a module load -- not a stream operation), so that hydra_symv_static itself is nothing
// but two kernel launches afterwards and may run under HIP graph capture.
int hydra_symv_static_max_n(void) {
  double* p = nullptr;
  (void)scratch_ptr_for_current_device(&p);      // a failure is reported by hydra_symv_static
  return SYMV_STATIC_MAX_N;
}

int hydra_symv_static(const double* A, int n, const double* x, double* y, int accumulate,
                      void* stream) {
  HYDRA_CHECK_ARG(A && x && y && n > 0, "hydra_symv_static: bad arguments (n=%d)", n);
  if (n > SYMV_STATIC_MAX_N)
    return fail(HYDRA_ERR_ARG,
                "hydra_symv_static: n=%d exceeds the static scratch (max n=%d); use "
                "hydra_symv_batched with a caller workspace", n, SYMV_STATIC_MAX_N);
  double* part = nullptr;
  int rc = scratch_ptr_for_current_device(&part);
  if (rc) return rc;
  int nbk = (n + SV_T - 1) / SV_T;
  int nwg = nbk * (nbk + 1) / 2;
  hipStream_t s = as_stream(stream);
  bool vec2 = n % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0;
  // every (block column, row < n) piece is written before the second launch reads it, so the
  // contents left by the previous call never reach y and nothing needs clearing
  if (vec2) symv_blocks_kernel<true><<<dim3(nwg, 1), 256, 0, s>>>(A, n, nbk, x, part);
  else      symv_blocks_kernel<false><<<dim3(nwg, 1), 256, 0, s>>>(A, n, nbk, x, part);
  symv_reduce_kernel<<<dim3(nbk, 1), 256, 0, s>>>(part, n, nbk, y, accumulate);
  HYDRA_CHECK_LAUNCH("hydra_symv_static");
  return HYDRA_OK;
}

int hydra_cg_init(int n, const double* b, const double* Ax, double* r, double* p, double* st,
                  double abs_tol, int use_norm_tol, void* stream) {
  HYDRA_CHECK_ARG(n > 0 && b && Ax && r && p && st, "hydra_cg_init: bad arguments");
  cg_init_kernel<<<1, CG_THREADS, 0, as_stream(stream)>>>(n, b, Ax, r, p, st, abs_tol, use_norm_tol);
  HYDRA_CHECK_LAUNCH("hydra_cg_init");
  return HYDRA_OK;
}

int hydra_cg_step(int n, const double* Ap, double* x, double* r, double* p, double* st,
                  double abs_tol, int use_norm_tol, void* stream) {
  HYDRA_CHECK_ARG(n > 0 && Ap && x && r && p && st, "hydra_cg_step: bad arguments");
  if (n >= CGL_MIN_N) {
    int nblk = std::min(CGL_MAXB, (n + 4 * CGL_T - 1) / (4 * CGL_T));
    hipStream_t s = as_stream(stream);
    cgl_dot_kernel<<<nblk, CGL_T, 0, s>>>(n, Ap, p, st);
    cgl_update_kernel<<<nblk, CGL_T, 0, s>>>(n, Ap, p, x, r, st);
    cgl_dir_kernel<<<nblk, CGL_T, 0, s>>>(n, r, p, st, abs_tol, use_norm_tol);
  } else {
    cg_step_kernel<<<1, CG_THREADS, 0, as_stream(stream)>>>(n, Ap, x, r, p, st, abs_tol,
                                                            use_norm_tol);
  }
  HYDRA_CHECK_LAUNCH("hydra_cg_step");
  return HYDRA_OK;
}

int hydra_cg_state_len(void) { return CG_STATE_LEN; }

size_t hydra_cg_workspace_bytes(int n) {
  if (n <= 0) return 0;
  // the half-read symmetric product is only taken from n = 4096 on (below, one launch beats two)
  return ((size_t)3 * n + CG_STATE_LEN) * sizeof(double) +
         (n >= 4096 ? hydra_symv_workspace_bytes(n, 1) : 0);
}

int hydra_cg_solve_dense(const double* A, const double* b, double* x, int n, int maxiters,
                         double abs_tol, int use_norm_tol, int symmetric, int check_every,
                         int* iters_host, void* workspace, size_t workspace_bytes, void* stream) {
  HYDRA_CHECK_ARG(A && b && x && n > 0 && maxiters >= 0, "hydra_cg_solve_dense: bad arguments");
  size_t need = hydra_cg_workspace_bytes(n);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_cg_solve_dense: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  if (check_every <= 0) check_every = 16;
  double* r = static_cast<double*>(workspace);
  double* p = r + n;
  double* ap = p + n;
  double* st = ap + n;
  void* sv_ws = st + CG_STATE_LEN;
  size_t sv_bytes = n >= 4096 ? hydra_symv_workspace_bytes(n, 1) : 0;
  hipStream_t s = as_stream(stream);
  // A p: half the matrix is read when the caller vouches for exact symmetry
  auto matvec = [&](const double* v, double* out) {
    // below a few thousand unknowns the product is launch-bound and one launch beats two
    return (symmetric && n >= 4096) ? hydra_symv_batched(A, n, 1, v, out, 0, sv_ws, sv_bytes, stream)
                                    : hydra_gemv(A, n, v, out, stream);
  };
  int rc = matvec(x, ap);
  if (rc) return rc;
  rc = hydra_cg_init(n, b, ap, r, p, st, abs_tol, use_norm_tol, stream);
  if (rc) return rc;
  double st_host[4] = {0, 0, 0, 0};
  int it = 0;
  for (;;) {
    hipError_t e = hipMemcpyAsync(st_host, st, sizeof(st_host), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess)
      return fail(HYDRA_ERR_HIP, "hydra_cg_solve_dense: %s", hipGetErrorString(e));
    if (st_host[1] != 0.0 || it >= maxiters) break;
    int chunk = std::min(check_every, maxiters - it);
    for (int c = 0; c < chunk; ++c) {
      rc = matvec(p, ap);
      if (rc) return rc;
      rc = hydra_cg_step(n, ap, x, r, p, st, abs_tol, use_norm_tol, stream);
      if (rc) return rc;
    }
    it += chunk;
  }
  if (iters_host) *iters_host = (int)st_host[2];
  return HYDRA_OK;
}

}  // extern "C"
