// Linearised per-antenna gain head (hydra/gain_sampler.py:51-301) on gfx950.
//
//   operator:  x -> x + S^1/2 F^H A^H N^-1 A F S^1/2 x        gain_sampler.py:234-301
//   A x      = (x_i + conj(x_j)) * model_vis                   gain_sampler.py:81-106
//   A^H v    : y_i += v conj(model_vis), y_j += conj(v conj(model_vis))   :109-142
//
// Between A and A^H only the real weight W = |model_vis|^2 N^-1 per (baseline, f, t) survives:
//   (A^H N^-1 A g)_a = g_a sum_{b∋a} W_b + conj( sum_{b∋a} W_b g_other(b) ),
// so one matvec reads W once per antenna end (16 B per visibility sample, HBM bound; the
// reference's unfused chain moves >= 24 B three times) and the antenna fields g (Nant,Nf,Nt)
// stay in L2.  The sparse (Nbl x Nant) projectors of the reference become a CSR adjacency
// antenna -> (baseline, other antenna, role).  Several right-hand sides ("batched CG
// matvecs", BASELINE config 4) share one pass over W.  No atomics: every output element is
// owned by one thread and summed in adjacency order (bitwise reproducible).
#include <algorithm>

#include "common.h"

namespace hydra {

// W[b,ft] = |mv|^2 * w
__global__ void __launch_bounds__(256)
gain_weights_kernel(const f64x2* __restrict__ mv, const double* __restrict__ w, size_t n,
                    double* __restrict__ W) {
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < n;
       o += (size_t)gridDim.x * blockDim.x) {
    f64x2 m = mv[o];
    W[o] = (m.x * m.x + m.y * m.y) * (w ? w[o] : 1.0);
  }
}

// g[v,a,ft] = sum_m F[m,ft] * s_m (xre[v,a,m] + i xim[v,a,m]);  x (nvec, 2*nant*nmodes)
__global__ void __launch_bounds__(256)
gain_basis_fwd_kernel(const double* __restrict__ x, const f64x2* __restrict__ F,
                      const double* __restrict__ s, int nant, int nmodes, int nft,
                      f64x2* __restrict__ g) {
  int ft = blockIdx.x * blockDim.x + threadIdx.x;
  int a = blockIdx.y, v = blockIdx.z;
  if (ft >= nft) return;
  const double* xre = x + (size_t)v * 2 * nant * nmodes + (size_t)a * nmodes;
  const double* xim = xre + (size_t)nant * nmodes;
  f64x2 acc = {0.0, 0.0};
  for (int m = 0; m < nmodes; ++m) {
    double sm = s ? s[m] : 1.0;
    f64x2 z = {sm * xre[m], sm * xim[m]};
    f64x2 r = cmul(F[(size_t)m * nft + ft], z);
    acc.x += r.x;
    acc.y += r.y;
  }
  g[((size_t)v * nant + a) * nft + ft] = acc;
}

// y[v,a,ft] = g[v,a,ft] sum_b W[b,ft] + conj( sum_b W[b,ft] g[v,other(b),ft] ), b in adj(a)
template <int NV>
__global__ void __launch_bounds__(256)
gain_pairs_kernel(const f64x2* __restrict__ g, const double* __restrict__ W,
                  const int32_t* __restrict__ adj_ptr, const int32_t* __restrict__ adj_b,
                  const int32_t* __restrict__ adj_o, int nant, int nft, int v0, int nvec,
                  f64x2* __restrict__ y) {
  int ft = blockIdx.x * blockDim.x + threadIdx.x;
  int a = blockIdx.y;
  if (ft >= nft) return;
  double sw = 0.0;
  f64x2 acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q] = f64x2{0.0, 0.0};
  int e0 = adj_ptr[a], e1 = adj_ptr[a + 1];
  for (int e = e0; e < e1; ++e) {
    double wv = W[(size_t)adj_b[e] * nft + ft];
    size_t go = (size_t)adj_o[e] * nft + ft;
    sw += wv;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      if (v0 + q < nvec) {
        f64x2 go_v = g[(size_t)(v0 + q) * nant * nft + go];
        acc[q].x += wv * go_v.x;
        acc[q].y += wv * go_v.y;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    if (v0 + q < nvec) {
      size_t o = ((size_t)(v0 + q) * nant + a) * nft + ft;
      f64x2 ga = g[o];
      y[o] = f64x2{ga.x * sw + acc[q].x, ga.y * sw - acc[q].y};
    }
  }
}


// Tiled form of gain_pairs_kernel for pair lists without duplicates.  A workgroup owns 16
// antennas x 64 (f,t) samples: lanes <-> (f,t), each wave 4 antennas.  It walks all antennas
// o in blocks of 16 whose fields g_o (NV vectors) are staged in LDS, so one LDS read of g_o
// serves 4 antennas, and the only per-pair global traffic is the weight W itself (8 B, 512-B
// coalesced per wave) -- the CSR kernel re-reads g_o from L2 for every pair and is bound by the
// vector cache.  pairb[a*nant + o] = baseline of the pair {a,o} or -1; an auto-correlation
// (a == o) counts twice, as in the adjacency form.  Antenna blocks are the fast grid axis so the
// two ends of a baseline are read close in time (second read from L2 / MALL).
template <int NV>
__global__ void __launch_bounds__(256)
gain_pairs_tiled_kernel(const f64x2* __restrict__ g, const double* __restrict__ W,
                        const int32_t* __restrict__ pairb, int nant, int nft, int v0, int nvec,
                        f64x2* __restrict__ y) {
  __shared__ f64x2 s_g[16 * NV * 64];   // [o][v][ft lane]
  __shared__ int32_t s_b[16 * 16];      // [a local][o local] baseline index of the block pair
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 1-D grid of 8 * ceil(nchunk / 8) * nblk workgroups.  Consecutive ids go round robin to the 8
  // XCDs: id -> (xcd, k) gives every XCD whole (f,t) chunks, the nblk antenna blocks of a chunk
  // consecutive on it, so the two readers of a weight share an L2.
  const int nblk = (nant + 15) / 16;
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
  const int myblk = kx % nblk, chunk = (kx / nblk) * 8 + xcd;
  if (chunk * 64 >= nft) return;
  const int ablk = myblk * 16;
  const int a0 = ablk + wave * 4;
  const int ft = chunk * 64 + lane;
  const bool ftok = ft < nft;
  const int ftc = ftok ? ft : nft - 1;
  double sw[4] = {0.0, 0.0, 0.0, 0.0};
  f64x2 acc[4][NV];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = f64x2{0.0, 0.0};
  const double* Wft = W + ftc;
  // Partner blocks in round-robin-tournament order: at step s block A meets block (s - A) mod nblk,
  // an involution, so the two workgroups that need the weights of a block pair read them at the
  // same step and the second read comes from the memory-side cache instead of HBM (each weight
  // was fetched from HBM twice: PMC 3.85 GB per application against 1.79 GB of weights).
  for (int step = 0; step < nblk; ++step) {
    int ob = step - myblk;
    if (ob < 0) ob += nblk;
    const int o0 = ob * 16;
    __syncthreads();
    for (int e = wave; e < 16 * NV; e += 4) {     // e = o_local * NV + v
      int ol = e / NV, v = e - ol * NV;
      int o = min(o0 + ol, nant - 1);
      int vv = min(v0 + v, nvec - 1);
      s_g[e * 64 + lane] = g[((size_t)vv * nant + o) * nft + ftc];
    }
    {   // the 16 x 16 baseline indices of this block pair, one per thread (-1: no such pair)
      int al = threadIdx.x >> 4, ol = threadIdx.x & 15;
      int a = ablk + al, o = o0 + ol;
      s_b[threadIdx.x] = (a < nant && o < nant) ? pairb[(size_t)a * nant + o] : -1;
    }
    __syncthreads();
    // 4 other antennas per trip: their 16 weight loads are issued together
#pragma unroll
    for (int ol = 0; ol < 16; ol += 4) {
      double w[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int b = s_b[(wave * 4 + q) * 16 + ol + k];            // wave-uniform
          // unconditional load (a load under a condition becomes a branch with a wait behind
          // it and the 16 loads of a trip would go out one by one); absent pairs read baseline 0
          // and are zeroed by the factor
          double wv = Wft[(size_t)max(b, 0) * nft];
          double fac = (b >= 0) ? ((a0 + q == o0 + ol + k) ? 2.0 : 1.0) : 0.0;
          w[k][q] = wv * fac;
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sw[q] += w[k][q];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          f64x2 go = s_g[((ol + k) * NV + v) * 64 + lane];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[q][v].x += w[k][q] * go.x;
            acc[q][v].y += w[k][q] * go.y;
          }
        }
      }
    }
  }
  if (!ftok) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int a = a0 + q;
    if (a >= nant) continue;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v0 + v < nvec) {
        size_t o = ((size_t)(v0 + v) * nant + a) * nft + ft;
        f64x2 ga = g[o];
        y[o] = f64x2{ga.x * sw[q] + acc[q][v].x, ga.y * sw[q] - acc[q][v].y};
      }
    }
  }
}

// v[b,ft] = (x_i + conj(x_j)) * mv[b,ft]
__global__ void __launch_bounds__(256)
gain_proj_kernel(const f64x2* __restrict__ x, const int32_t* __restrict__ bl_i,
                 const int32_t* __restrict__ bl_j, const f64x2* __restrict__ mv, int nbl, int nft,
                 f64x2* __restrict__ out) {
  for (int b = blockIdx.y; b < nbl; b += gridDim.y) {   // grid.y is capped at 65535
    int i = bl_i[b], j = bl_j[b];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nft; k += gridDim.x * blockDim.x) {
      f64x2 xi = x[(size_t)i * nft + k], xj = x[(size_t)j * nft + k];
      f64x2 c = {1.0 * (xi.x + xj.x), 1.0 * (xi.y - xj.y)};
      size_t o = (size_t)b * nft + k;
      out[o] = cmul(c, mv[o]);
    }
  }
}

// y[a,ft] = sum_{b in adj(a)} role ? conj(v conj(mv)) : v conj(mv)
__global__ void __launch_bounds__(256)
gain_proj_conj_kernel(const f64x2* __restrict__ v, const f64x2* __restrict__ mv,
                      const int32_t* __restrict__ adj_ptr, const int32_t* __restrict__ adj_b,
                      const int32_t* __restrict__ adj_role, int nft, f64x2* __restrict__ y) {
  int ft = blockIdx.x * blockDim.x + threadIdx.x;
  int a = blockIdx.y;
  if (ft >= nft) return;
  f64x2 acc = {0.0, 0.0};
  int e0 = adj_ptr[a], e1 = adj_ptr[a + 1];
  for (int e = e0; e < e1; ++e) {
    size_t o = (size_t)adj_b[e] * nft + ft;
    f64x2 vv = cmul_conj(mv[o], v[o]);
    acc.x += vv.x;
    acc.y += adj_role[e] ? -vv.y : vv.y;
  }
  y[(size_t)a * nft + ft] = acc;
}

// out[v, a*nmodes+m] = Re, out[v, nant*nmodes + a*nmodes+m] = Im of
//   s_m sum_ft conj(F[m,ft]) y[v,a,ft]  (+ add[v, same index] if add != NULL)
// One workgroup per (antenna, vector): the y row is staged in LDS in chunks and every wave
// owns the modes m = wave, wave+4, ...; lanes stride over (f,t); fixed-order wave reduction.
constexpr int GB_CHUNK = 2048;
__global__ void __launch_bounds__(256)
gain_basis_adj_kernel(const f64x2* __restrict__ y, const f64x2* __restrict__ F,
                      const double* __restrict__ s, const double* __restrict__ add, int nant,
                      int nmodes, int nft, double* __restrict__ out) {
  __shared__ f64x2 s_y[GB_CHUNK];
  int a = blockIdx.x, v = blockIdx.y;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f64x2* yrow = y + ((size_t)v * nant + a) * nft;
  size_t obase = (size_t)v * 2 * nant * nmodes + (size_t)a * nmodes;
  for (int m0 = 0; m0 < nmodes; m0 += 4 * 8) {
    f64x2 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f64x2{0.0, 0.0};
    for (int c0 = 0; c0 < nft; c0 += GB_CHUNK) {
      int cn = min(GB_CHUNK, nft - c0);
      __syncthreads();
      for (int k = threadIdx.x; k < cn; k += 256) s_y[k] = yrow[c0 + k];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int m = m0 + q * 4 + wave;
        if (m < nmodes) {
          const f64x2* fr = F + (size_t)m * nft + c0;
          f64x2 t = acc[q];
          // 8 basis loads in flight per lane (unconditional, clamped): one dependent load per
          // trip left this kernel at 0.12 ms per call, latency bound.  Same summation order.
          for (int kb = lane; kb < cn; kb += 512) {
            f64x2 fv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) fv[u] = fr[min(kb + 64 * u, cn - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              int k = kb + 64 * u;
              if (k < cn) {
                f64x2 r = cmul_conj(fv[u], s_y[k]);
                t.x += r.x;
                t.y += r.y;
              }
            }
          }
          acc[q] = t;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int m = m0 + q * 4 + wave;
      if (m < nmodes) {  // uniform per wave
        double re = wave_sum(acc[q].x), im = wave_sum(acc[q].y);
        if (lane == 0) {
          double sm = s ? s[m] : 1.0;
          size_t o = obase + m;
          size_t oi = o + (size_t)nant * nmodes;
          out[o] = sm * re + (add ? add[o] : 0.0);
          out[oi] = sm * im + (add ? add[oi] : 0.0);
        }
      }
    }
  }
}

}  // namespace hydra

using namespace hydra;

static int check_adj(const char* name, const int32_t* adj_ptr_host, int nant, int nadj) {
  HYDRA_CHECK_ARG(adj_ptr_host && adj_ptr_host[0] == 0 && adj_ptr_host[nant] == nadj,
                  "%s: adjacency pointer must run from 0 to nadj=%d", name, nadj);
  for (int a = 0; a < nant; ++a)
    HYDRA_CHECK_ARG(adj_ptr_host[a] <= adj_ptr_host[a + 1], "%s: adjacency pointer not monotone",
                    name);
  return HYDRA_OK;
}

extern "C" {

int hydra_gain_weights(const double* model_vis, const double* inv_noise, long long m, double* W,
                       void* stream) {
  HYDRA_CHECK_ARG(model_vis && W && m > 0, "hydra_gain_weights: bad arguments");
  int grid = (int)std::min<long long>((m + 255) / 256, 8192);
  gain_weights_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(model_vis), inv_noise, (size_t)m, W);
  HYDRA_CHECK_LAUNCH("hydra_gain_weights");
  return HYDRA_OK;
}

int hydra_gain_basis_fwd(const double* x, int nvec, int nant, int nmodes, int nft,
                         const double* Fbasis, const double* pspec_sqrt, double* g, void* stream) {
  HYDRA_CHECK_ARG(x && Fbasis && g && nvec > 0 && nant > 0 && nmodes > 0 && nft > 0,
                  "hydra_gain_basis_fwd: bad arguments");
  HYDRA_CHECK_ARG(nant <= 65535 && nvec <= 65535, "hydra_gain_basis_fwd: nant, nvec <= 65535");
  dim3 grid((nft + 255) / 256, nant, nvec);
  gain_basis_fwd_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      x, reinterpret_cast<const f64x2*>(Fbasis), pspec_sqrt, nant, nmodes, nft,
      reinterpret_cast<f64x2*>(g));
  HYDRA_CHECK_LAUNCH("hydra_gain_basis_fwd");
  return HYDRA_OK;
}

int hydra_gain_pairs(const double* g, int nvec, int nant, int nft, const double* W, int nbl,
                     const int32_t* adj_ptr_host, const int32_t* adj_ptr, const int32_t* adj_b,
                     const int32_t* adj_o, int nadj, double* y, void* stream) {
  HYDRA_CHECK_ARG(g && W && adj_ptr && adj_b && adj_o && y, "hydra_gain_pairs: null pointer");
  HYDRA_CHECK_ARG(nvec > 0 && nant > 0 && nft > 0 && nbl > 0 && nadj >= 0 && nant <= 65535,
                  "hydra_gain_pairs: bad sizes");
  if (int rc = check_adj("hydra_gain_pairs", adj_ptr_host, nant, nadj)) return rc;
  dim3 grid((nft + 255) / 256, nant);
  hipStream_t st = as_stream(stream);
  const f64x2* gp = reinterpret_cast<const f64x2*>(g);
  f64x2* yp = reinterpret_cast<f64x2*>(y);
  for (int v0 = 0; v0 < nvec; v0 += 8) {
    int left = nvec - v0;
    if (left >= 5)
      gain_pairs_kernel<8><<<grid, 256, 0, st>>>(gp, W, adj_ptr, adj_b, adj_o, nant, nft, v0, nvec, yp);
    else if (left >= 3)
      gain_pairs_kernel<4><<<grid, 256, 0, st>>>(gp, W, adj_ptr, adj_b, adj_o, nant, nft, v0, nvec, yp);
    else if (left == 2)
      gain_pairs_kernel<2><<<grid, 256, 0, st>>>(gp, W, adj_ptr, adj_b, adj_o, nant, nft, v0, nvec, yp);
    else
      gain_pairs_kernel<1><<<grid, 256, 0, st>>>(gp, W, adj_ptr, adj_b, adj_o, nant, nft, v0, nvec, yp);
    HYDRA_CHECK_LAUNCH("hydra_gain_pairs");
  }
  return HYDRA_OK;
}

int hydra_gain_proj(const double* x, const int32_t* bl_i, const int32_t* bl_j, int nbl,
                    const double* model_vis, int nft, double* out, void* stream) {
  HYDRA_CHECK_ARG(x && bl_i && bl_j && model_vis && out && nbl > 0 && nft > 0,
                  "hydra_gain_proj: bad arguments");
  dim3 grid(std::min((nft + 255) / 256, 1024), std::min(nbl, 65535));
  gain_proj_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(x), bl_i, bl_j, reinterpret_cast<const f64x2*>(model_vis),
      nbl, nft, reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_gain_proj");
  return HYDRA_OK;
}

int hydra_gain_proj_conj(const double* v, const double* model_vis, int nant, int nft,
                         const int32_t* adj_ptr_host, const int32_t* adj_ptr,
                         const int32_t* adj_b, const int32_t* adj_role, int nadj, double* y,
                         void* stream) {
  HYDRA_CHECK_ARG(v && model_vis && adj_ptr && adj_b && adj_role && y && nant > 0 && nft > 0 &&
                      nant <= 65535,
                  "hydra_gain_proj_conj: bad arguments");
  if (int rc = check_adj("hydra_gain_proj_conj", adj_ptr_host, nant, nadj)) return rc;
  dim3 grid((nft + 255) / 256, nant);
  gain_proj_conj_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(v), reinterpret_cast<const f64x2*>(model_vis), adj_ptr, adj_b,
      adj_role, nft, reinterpret_cast<f64x2*>(y));
  HYDRA_CHECK_LAUNCH("hydra_gain_proj_conj");
  return HYDRA_OK;
}

int hydra_gain_basis_adj(const double* y, int nvec, int nant, int nmodes, int nft,
                         const double* Fbasis, const double* pspec_sqrt, const double* add,
                         double* out, void* stream) {
  HYDRA_CHECK_ARG(y && Fbasis && out && nvec > 0 && nant > 0 && nmodes > 0 && nft > 0,
                  "hydra_gain_basis_adj: bad arguments");
  HYDRA_CHECK_ARG(nvec <= 65535, "hydra_gain_basis_adj: nvec <= 65535");
  dim3 grid(nant, nvec);
  gain_basis_adj_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(y), reinterpret_cast<const f64x2*>(Fbasis), pspec_sqrt, add,
      nant, nmodes, nft, out);
  HYDRA_CHECK_LAUNCH("hydra_gain_basis_adj");
  return HYDRA_OK;
}


int hydra_gain_pairs_tiled(const double* g, int nvec, int nant, int nft, const double* W, int nbl,
                           const int32_t* pairb, double* y, void* stream) {
  HYDRA_CHECK_ARG(g && W && pairb && y, "hydra_gain_pairs_tiled: null pointer");
  HYDRA_CHECK_ARG(nvec > 0 && nant > 0 && nft > 0 && nbl > 0, "hydra_gain_pairs_tiled: bad sizes");
  long long nwg = (long long)(((nft + 63) / 64 + 7) / 8) * 8 * ((nant + 15) / 16);
  HYDRA_CHECK_ARG(nwg <= 2147483647LL, "hydra_gain_pairs_tiled: nft too large");
  unsigned grid = (unsigned)nwg;
  hipStream_t st = as_stream(stream);
  const f64x2* gp = reinterpret_cast<const f64x2*>(g);
  f64x2* yp = reinterpret_cast<f64x2*>(y);
  for (int v0 = 0; v0 < nvec; v0 += 4) {
    int left = nvec - v0;
    if (left >= 3)
      gain_pairs_tiled_kernel<4><<<grid, 256, 0, st>>>(gp, W, pairb, nant, nft, v0, nvec, yp);
    else if (left == 2)
      gain_pairs_tiled_kernel<2><<<grid, 256, 0, st>>>(gp, W, pairb, nant, nft, v0, nvec, yp);
    else
      gain_pairs_tiled_kernel<1><<<grid, 256, 0, st>>>(gp, W, pairb, nant, nft, v0, nvec, yp);
    HYDRA_CHECK_LAUNCH("hydra_gain_pairs_tiled");
  }
  return HYDRA_OK;
}

}  // extern "C"
