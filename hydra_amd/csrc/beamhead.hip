// Consumers of the visibility generator in the per-antenna beam head
// (hydra/per_ant_beam_sampler.py:321-465): the uniform-beam per-source visibilities, made
// Hermitian in the antenna pair, contracted over sources with the Fourier-Bessel design matrices
//
//   C[f,t,a,A,q] = sum_s M_s[a,A] B[t,s,q],   M_s[a,A] = conj(V_a,s) V_A,s ref[f,t,s]        (a < A),
//                                              = conj(V_a,s) V_A,s conj(ref[f,t,s])  (a > A),
//                                              = 2 |V_a,s|^2 Re ref[f,t,s]           (a = A)
//
// (the reference scales the upper-triangular cube by ref and adds it to its own conjugate
// transpose, which doubles the autos, :410-413).  The per-source cube (Nf,Nt,Nant,Nant,Ns) of the
// reference is never formed: the pair products are generated from LDS-staged antenna voltages
// inside the contraction, a complex GEMM with K = sources per (f,t).  FP64 vector work,
// 8 Nant^2 Ns Q flops per (f,t).
#include <algorithm>

#include "common.h"

namespace hydra {

constexpr int BH_TA = 64;   // antennas A per tile
constexpr int BH_TQ = 64;   // basis columns q per tile
constexpr int BH_KS = 16;   // sources per staged chunk

// grid (ft, rows * ceil(nant / 64), ceil(nq / 64)), 256 threads = 16 (A) x 16 (q), 4 x 4 outputs each.
// row >= 0: only the antenna row `row` is formed and conjugated, i.e. the column
//   C[f,t,A,row,q] = sum_s conj(V_A) V_row ref B  of the Hermitian pair matrix, written to an
//   (Nf,Nt,Nant,Q) array -- all the sampled-antenna step needs (:458).
__global__ void __launch_bounds__(256)
bess_contract_kernel(const f64x2* __restrict__ V, int nft, int nt, int nant, int ns,
                     const f64x2* __restrict__ B, int nq, const f64x2* __restrict__ ref, int row,
                     f64x2* __restrict__ out) {
  __shared__ __align__(16) f64x2 sP[BH_KS][BH_TA + 1];
  __shared__ __align__(16) f64x2 sB[BH_KS][BH_TQ];
  const int ft = blockIdx.x, t = ft % nt;
  const int ntile_a = (nant + BH_TA - 1) / BH_TA;
  const int a = (row >= 0) ? row : (int)blockIdx.y / ntile_a;
  const int A0 = ((int)blockIdx.y % ntile_a) * BH_TA;
  const int q0 = blockIdx.z * BH_TQ;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const f64x2* vft = V + (size_t)ft * nant * ns;
  const f64x2* va = vft + (size_t)a * ns;
  const f64x2* bt = B + (size_t)t * ns * nq;
  const f64x2* rf = ref ? ref + (size_t)ft * ns : nullptr;
  f64x2 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x2{0.0, 0.0};

  for (int s0 = 0; s0 < ns; s0 += BH_KS) {
    __syncthreads();
    // pair products of this chunk: consecutive threads <-> consecutive sources of one antenna
#pragma unroll
    for (int e = tid; e < BH_KS * BH_TA; e += 256) {
      int ks = e % BH_KS, Al = e / BH_KS;
      int s = s0 + ks, A = A0 + Al;
      f64x2 p = {0.0, 0.0};
      if (s < ns && A < nant) {
        f64x2 x = va[s], y = vft[(size_t)A * ns + s];
        p = cmul_conj(x, y);                       // conj(V_a) V_A = entry (a, A)
        if (row >= 0) p.y = -p.y;                  // entry (A, row) of the Hermitian matrix
        // the reference scales the upper-triangular cube by ref and THEN adds the conjugate
        // transpose (:410-413): entries above the diagonal carry ref, those below conj(ref),
        // the autos appear twice (with Re ref)
        const int i = (row >= 0) ? A : a, j = (row >= 0) ? a : A;
        f64x2 r = rf ? rf[s] : f64x2{1.0, 0.0};
        if (i > j) r.y = -r.y;
        p = (i == j) ? f64x2{2.0 * p.x * r.x, 0.0} : cmul(p, r);
      }
      sP[ks][Al] = p;
    }
#pragma unroll
    for (int e = tid; e < BH_KS * BH_TQ; e += 256) {
      int ql = e % BH_TQ, ks = e / BH_TQ;
      int s = s0 + ks, q = q0 + ql;
      sB[ks][ql] = (s < ns && q < nq) ? bt[(size_t)s * nq + q] : f64x2{0.0, 0.0};
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < BH_KS; ++ks) {
      f64x2 p[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = sP[ks][ty + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sB[ks][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j].x += p[i].x * b[j].x - p[i].y * b[j].y;
          acc[i][j].y += p[i].x * b[j].y + p[i].y * b[j].x;
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int A = A0 + ty + 16 * i;
    if (A >= nant) continue;
    size_t base = (row >= 0) ? ((size_t)ft * nant + A) * nq
                             : (((size_t)ft * nant + a) * nant + A) * nq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int q = q0 + tx + 16 * j;
      if (q < nq) out[base + q] = acc[i][j];
    }
  }
}

// out[f,t,k,B] = sum_b conj(coef[b,f,A_k]) C[f,t,A_k,(samp,) b,B],  A_k = k-th antenna != samp
// (get_bess_to_vis_from_contraction, :453-463).  C is the full contraction (Nf,Nt,Nant,Nant,Nc,NB)
// (column `samp` is used) or, with col_only, the column alone (Nf,Nt,Nant,Nc,NB).  NB = Nc, or 1
// for a contraction with a reference beam.  One workgroup per (f,t,k): threads over B, the Nc x NB
// block is read once (HBM bound).
__global__ void __launch_bounds__(256)
bess_to_vis_kernel(const f64x2* __restrict__ C, const f64x2* __restrict__ coef, int nf, int nt,
                   int nant, int nc, int nb, int samp, int col_only, f64x2* __restrict__ out) {
  __shared__ f64x2 sc[256];
  const int k = blockIdx.x % (nant - 1);
  const int ft = blockIdx.x / (nant - 1);
  const int f = ft / nt;
  const int A = k + (k >= samp ? 1 : 0);
  const f64x2* blk = col_only ? C + ((size_t)ft * nant + A) * nc * nb
                              : C + (((size_t)ft * nant + A) * nant + samp) * nc * nb;
  f64x2* o = out + (size_t)blockIdx.x * nb;
  for (int b0 = 0; b0 < nc; b0 += 256) {
    __syncthreads();
    int bb = b0 + threadIdx.x;
    if (bb < nc) {
      f64x2 c = coef[((size_t)bb * nf + f) * nant + A];
      sc[threadIdx.x] = f64x2{c.x, -c.y};
    }
    __syncthreads();
    int nbl = min(256, nc - b0);
    for (int B = threadIdx.x; B < nb; B += 256) {
      f64x2 acc = (b0 == 0) ? f64x2{0.0, 0.0} : o[B];
      for (int b = 0; b < nbl; ++b) {
        f64x2 v = blk[(size_t)(b0 + b) * nb + B];
        f64x2 c = sc[b];
        acc.x += c.x * v.x - c.y * v.y;
        acc.y += c.x * v.y + c.y * v.x;
      }
      o[B] = acc;
    }
  }
}

}  // namespace hydra

using namespace hydra;

extern "C" {

int hydra_bess_sky_contraction(const double* V, int nf, int nt, int nant, int ns, const double* B,
                               int nq, const double* ref, int row, double* out, void* stream) {
  HYDRA_CHECK_ARG(V && B && out && nf > 0 && nt > 0 && nant > 0 && ns > 0 && nq > 0,
                  "hydra_bess_sky_contraction: bad arguments");
  HYDRA_CHECK_ARG(row < nant, "hydra_bess_sky_contraction: row %d out of range (Nant = %d)", row,
                  nant);
  int ntile_a = (nant + BH_TA - 1) / BH_TA, ntile_q = (nq + BH_TQ - 1) / BH_TQ;
  long long gy = (long long)(row >= 0 ? 1 : nant) * ntile_a;
  HYDRA_CHECK_ARG(gy <= 65535 && ntile_q <= 65535, "hydra_bess_sky_contraction: grid too large");
  dim3 grid(nf * nt, (unsigned)gy, ntile_q);
  bess_contract_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(V), nf * nt, nt, nant, ns, reinterpret_cast<const f64x2*>(B), nq,
      reinterpret_cast<const f64x2*>(ref), row, reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_bess_sky_contraction");
  return HYDRA_OK;
}

int hydra_bess_to_vis(const double* C, const double* coef, int nf, int nt, int nant, int nc, int nb,
                      int samp, int col_only, double* out, void* stream) {
  HYDRA_CHECK_ARG(C && coef && out && nf > 0 && nt > 0 && nant > 1 && nc > 0 && nb > 0,
                  "hydra_bess_to_vis: bad arguments");
  HYDRA_CHECK_ARG(samp >= 0 && samp < nant, "hydra_bess_to_vis: antenna %d out of range", samp);
  long long nblk = (long long)nf * nt * (nant - 1);
  HYDRA_CHECK_ARG(nblk <= 2147483647LL, "hydra_bess_to_vis: too many (f,t,antenna) blocks");
  bess_to_vis_kernel<<<(unsigned)nblk, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(C), reinterpret_cast<const f64x2*>(coef), nf, nt, nant, nc, nb,
      samp, col_only, reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_bess_to_vis");
  return HYDRA_OK;
}

}  // extern "C"
