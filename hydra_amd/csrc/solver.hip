// Linear solvers for the GCR system on gfx950:
//   * dense symmetric matrix-vector product (HBM/L2 bound: 8 N^2 bytes per product);
//   * the reference's conjugate-gradient iteration (hydra/linear_solver.py:378-509) as one
//     fused step kernel (both dot products, the three vector updates and the convergence
//     test in a single launch, state kept on the device);
//   * a blocked Cholesky solve as the dense fallback (example.py:944).
#include <algorithm>

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
    for (int c = lane; c < n; c += 64) s += ap[c] * x[c];
    s = wave_sum(s);
    if (lane == 0) y[r] = s;
  }
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
  if (st[1] != 0.0) return;  // converged earlier: this enqueued step is a no-op
  double rr = st[0];
  double pap = 0.0;
  for (int i = threadIdx.x; i < n; i += CG_THREADS) pap += p[i] * Ap[i];
  pap = block_sum(pap, s_red);
  double alpha = rr / pap;
  double rr_new = 0.0, mx = 0.0;
  for (int i = threadIdx.x; i < n; i += CG_THREADS) {
    double pi = p[i];
    x[i] = x[i] + alpha * pi;
    double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    rr_new += ri * ri;
    mx = fmax(mx, fabs(ri));
  }
  rr_new = block_sum(rr_new, s_red);
  mx = block_max(mx, s_red);
  double beta = rr_new / rr;
  for (int i = threadIdx.x; i < n; i += CG_THREADS) p[i] = r[i] + beta * p[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    double res = use_norm ? sqrt(rr_new) : mx;
    st[0] = rr_new;
    st[1] = (res < tol) ? 1.0 : 0.0;
    st[2] = st[2] + 1.0;
    st[3] = res;
  }
}

// ---------------------------------------------------------------- blocked Cholesky

constexpr int NB = 32;

// factor the NBxNB diagonal block at (k,k) in place (lower), one workgroup of NB*NB threads
__global__ void __launch_bounds__(NB * NB)
potrf_diag_kernel(double* __restrict__ A, int n, int k, int32_t* __restrict__ info) {
  __shared__ double s[NB][NB + 1];
  int i = threadIdx.y, j = threadIdx.x;
  int nb = min(NB, n - k);
  s[i][j] = (i < nb && j < nb) ? A[(size_t)(k + i) * n + k + j] : (i == j ? 1.0 : 0.0);
  __syncthreads();
  for (int c = 0; c < nb; ++c) {
    if (i == c && j == c) {
      double d = s[c][c];
      if (!(d > 0.0)) atomicOr(info, 1);
      s[c][c] = sqrt(d);
    }
    __syncthreads();
    if (j == c && i > c) s[i][c] /= s[c][c];
    __syncthreads();
    if (j > c && i >= j) s[i][j] -= s[i][c] * s[j][c];
    __syncthreads();
  }
  if (i < nb && j < nb && j <= i) A[(size_t)(k + i) * n + k + j] = s[i][j];
}

// rows below the diagonal block: A[r, k:k+nb] <- A[r, k:k+nb] L11^-T, thread <-> row
__global__ void __launch_bounds__(256)
trsm_panel_kernel(double* __restrict__ A, int n, int k) {
  __shared__ double l[NB][NB + 1];
  int nb = min(NB, n - k);
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    int i = e / NB, j = e % NB;
    l[i][j] = (i < nb && j <= i) ? A[(size_t)(k + i) * n + k + j] : 0.0;
  }
  __syncthreads();
  int r = k + nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double* row = A + (size_t)r * n + k;
  double v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = (j < nb) ? row[j] : 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < nb) {
      double t = v[j];
#pragma unroll
      for (int c = 0; c < j; ++c) t -= v[c] * l[j][c];
      v[j] = t / l[j][j];
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (j < nb) row[j] = v[j];
}

// trailing update of the lower triangle: A22 -= L21 L21^T with 32x32 output tiles
__global__ void __launch_bounds__(NB * NB)
syrk_update_kernel(double* __restrict__ A, int n, int k) {
  int nb = min(NB, n - k);
  int base = k + nb;
  int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  __shared__ double sa[NB][NB + 1], sb[NB][NB + 1];
  int ty = threadIdx.y, tx = threadIdx.x;
  int ri = base + bi * NB + ty, rj = base + bj * NB + ty;
  sa[ty][tx] = (ri < n && tx < nb) ? A[(size_t)ri * n + k + tx] : 0.0;
  sb[ty][tx] = (rj < n && tx < nb) ? A[(size_t)rj * n + k + tx] : 0.0;
  __syncthreads();
  int gi = base + bi * NB + ty, gj = base + bj * NB + tx;
  if (gi < n && gj < n && gj <= gi) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) s += sa[ty][c] * sb[tx][c];
    A[(size_t)gi * n + gj] -= s;
  }
}

// forward substitution with the block column k: y_k = L_kk^-1 b_k (one wave), then
// b_rest -= L[rest, k] y_k by the update kernel.
__global__ void __launch_bounds__(64)
trsv_diag_kernel(const double* __restrict__ A, int n, int k, double* __restrict__ b, int upper) {
  __shared__ double l[NB][NB + 1];
  __shared__ double y[NB];
  int nb = min(NB, n - k);
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    int i = e / NB, j = e % NB;
    l[i][j] = (i < nb && j <= i) ? A[(size_t)(k + i) * n + k + j] : 0.0;
  }
  if (threadIdx.x < NB) y[threadIdx.x] = (threadIdx.x < nb) ? b[k + threadIdx.x] : 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!upper) {
      for (int i = 0; i < nb; ++i) {
        double t = y[i];
        for (int c = 0; c < i; ++c) t -= l[i][c] * y[c];
        y[i] = t / l[i][i];
      }
    } else {  // solve L_kk^T x = y
      for (int i = nb - 1; i >= 0; --i) {
        double t = y[i];
        for (int c = i + 1; c < nb; ++c) t -= l[c][i] * y[c];
        y[i] = t / l[i][i];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nb) b[k + threadIdx.x] = y[threadIdx.x];
}

// lower: b[r] -= sum_c L[r, k+c] y[k+c] for r >= k+nb
// upper: b[r] -= sum_c L[k+c, r] x[k+c] for r < k     (L^T above the block)
__global__ void __launch_bounds__(256)
trsv_update_kernel(const double* __restrict__ A, int n, int k, double* __restrict__ b, int upper) {
  __shared__ double y[NB];
  int nb = min(NB, n - k);
  if (threadIdx.x < NB) y[threadIdx.x] = (threadIdx.x < nb) ? b[k + threadIdx.x] : 0.0;
  __syncthreads();
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (!upper) {
    int r = k + nb + idx;
    if (r >= n) return;
    const double* row = A + (size_t)r * n + k;
    double s = 0.0;
    for (int c = 0; c < nb; ++c) s += row[c] * y[c];
    b[r] -= s;
  } else {
    int r = idx;
    if (r >= k) return;
    double s = 0.0;
    for (int c = 0; c < nb; ++c) s += A[(size_t)(k + c) * n + r] * y[c];
    b[r] -= s;
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
  cg_step_kernel<<<1, CG_THREADS, 0, as_stream(stream)>>>(n, Ap, x, r, p, st, abs_tol,
                                                          use_norm_tol);
  HYDRA_CHECK_LAUNCH("hydra_cg_step");
  return HYDRA_OK;
}

size_t hydra_cg_workspace_bytes(int n) {
  if (n <= 0) return 0;
  return ((size_t)3 * n + 8) * sizeof(double);
}

int hydra_cg_solve_dense(const double* A, const double* b, double* x, int n, int maxiters,
                         double abs_tol, int use_norm_tol, int check_every, int* iters_host,
                         void* workspace, size_t workspace_bytes, void* stream) {
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
  hipStream_t s = as_stream(stream);
  int rc = hydra_gemv(A, n, x, ap, stream);
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
      rc = hydra_gemv(A, n, p, ap, stream);
      if (rc) return rc;
      rc = hydra_cg_step(n, ap, x, r, p, st, abs_tol, use_norm_tol, stream);
      if (rc) return rc;
    }
    it += chunk;
  }
  if (iters_host) *iters_host = (int)st_host[2];
  return HYDRA_OK;
}

int hydra_chol_solve(double* A, double* b, int n, int32_t* info_dev, void* stream) {
  HYDRA_CHECK_ARG(A && b && info_dev && n > 0, "hydra_chol_solve: bad arguments");
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(info_dev, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_chol_solve: %s", hipGetErrorString(e));
  dim3 tb(NB, NB);
  for (int k = 0; k < n; k += NB) {
    int nb = std::min(NB, n - k);
    potrf_diag_kernel<<<1, tb, 0, s>>>(A, n, k, info_dev);
    int rest = n - k - nb;
    if (rest > 0) {
      trsm_panel_kernel<<<(rest + 255) / 256, 256, 0, s>>>(A, n, k);
      int nblk = (rest + NB - 1) / NB;
      syrk_update_kernel<<<dim3(nblk, nblk), tb, 0, s>>>(A, n, k);
    }
  }
  HYDRA_CHECK_LAUNCH("hydra_chol_solve(factor)");
  for (int k = 0; k < n; k += NB) {  // L y = b
    int nb = std::min(NB, n - k);
    trsv_diag_kernel<<<1, 64, 0, s>>>(A, n, k, b, 0);
    int rest = n - k - nb;
    if (rest > 0) trsv_update_kernel<<<(rest + 255) / 256, 256, 0, s>>>(A, n, k, b, 0);
  }
  int last = ((n - 1) / NB) * NB;
  for (int k = last; k >= 0; k -= NB) {  // L^T x = y
    trsv_diag_kernel<<<1, 64, 0, s>>>(A, n, k, b, 1);
    if (k > 0) trsv_update_kernel<<<(k + 255) / 256, 256, 0, s>>>(A, n, k, b, 1);
  }
  HYDRA_CHECK_LAUNCH("hydra_chol_solve(solve)");
  return HYDRA_OK;
}

}  // extern "C"
