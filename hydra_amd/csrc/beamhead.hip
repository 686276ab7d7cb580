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

constexpr int BH_TM = 64;    // antenna pairs (a, A) per tile
constexpr int BH_TQ = 128;   // basis columns q per tile
constexpr int BH_KS = 16;    // sources per staged chunk

// The contraction per (f,t) is one complex GEMM  C[m,q] = sum_s P[s,m] B[s,q]  over the FLATTENED
// pair index m = a Nant + A (the output (Nf,Nt,Nant,Nant,Q) is contiguous in m, and no tile is
// padded from Nant up to a block multiple), K = sources.
// One workgroup per (ft, 64-pair tile, 128-column tile), 512 threads = 8 waves = 2 (m) x 4 (q), wave tile 32 x 32.
// Matrix cores, three real products per complex one:
//   P1 = Px Bx,  P2 = Py By,  P3 = (Px + Py)(Bx + By):  Re = P1 - P2,  Im = P3 - P1 - P2
// accumulated over all sources and combined once at the end (12 accumulator blocks, 48 VGPRs).
// The pair products of a chunk of 16 sources are generated from global voltages into LDS by all
// threads, B is copied beside them; the next chunk's global loads are issued before the matrix
// loop of the current one and land in registers meanwhile (unconditional loads from clamped
// addresses; out-of-range sources get P = 0, out-of-range pairs and columns are not stored).
// row >= 0: only the antenna row `row` is formed and conjugated, i.e. the column
//   C[f,t,A,row,q] = sum_s conj(V_A) V_row ref B  of the Hermitian pair matrix, written to an
//   (Nf,Nt,Nant,Q) array -- all the sampled-antenna step needs (:458); then M = Nant, m = A.
__global__ void __launch_bounds__(512, 2)
bess_contract_kernel(const f64x2* __restrict__ V, int nft, int nt, int nant, int ns,
                     const f64x2* __restrict__ B, int nq, const f64x2* __restrict__ ref, int row,
                     f64x2* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char bh_smem[];
  // rows of sP are padded by one element: the 16 lanes that write 16 sources of one pair would
  // otherwise all hit the same banks
  auto sP = reinterpret_cast<f64x2 (*)[BH_KS][BH_TM + 1]>(bh_smem);                                    // [2]
  auto sB = reinterpret_cast<f64x2 (*)[BH_KS][BH_TQ]>(bh_smem + 2 * sizeof(f64x2[BH_KS][BH_TM + 1]));  // [2]
  // 1-D grid of 8 * per blocks.  Hardware deals consecutive block ids round-robin to the 8 XCDs;
  // id' = (id % 8) * per + id / 8 gives every XCD a contiguous range of tiles, pair tile fastest,
  // then column tile, then (f,t): the 22 pair tiles that share one B tile and one voltage slice
  // run together under one L2.
  const int mtot = (row >= 0) ? nant : nant * nant;
  const int nmt = (mtot + BH_TM - 1) / BH_TM, nqt = (nq + BH_TQ - 1) / BH_TQ;
  const long long per = gridDim.x / 8;
  const long long tile = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
  if (tile >= (long long)nft * nmt * nqt) return;
  const int ft = (int)(tile / ((long long)nmt * nqt)), t = ft % nt;
  const int m0 = (int)(tile % nmt) * BH_TM, q0 = (int)((tile / nmt) % nqt) * BH_TQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wq = wave >> 1;
  const int kr = lane >> 4, lc = lane & 15;
  const f64x2* vft = V + (size_t)ft * nant * ns;
  const f64x2* bt = B + (size_t)t * ns * nq;
  const f64x2* rf = ref ? ref + (size_t)ft * ns : nullptr;

  // staging roles.  P: element (ks, ml0 + 32 u), u < 2 -- 16 consecutive threads <-> 16 consecutive
  // sources of one pair.  B: element (kb0 + 4 u, ql), u < 4 -- consecutive threads <-> columns.
  const int ks = tid & 15, ml0 = tid >> 4;
  const int ql = tid & 127, kb0 = tid >> 7;
  int offa[2], offA[2];      // row offsets of V_a and V_A (elements)
  int kind[2];               // pair (i, j): 0 auto, +1 above the diagonal, -1 below
  bool mval[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int m = m0 + ml0 + 32 * u;
    mval[u] = m < mtot;
    m = min(m, mtot - 1);
    int a = (row >= 0) ? row : m / nant;
    int A = (row >= 0) ? m : m - a * nant;
    offa[u] = a * ns;
    offA[u] = A * ns;
    // entry (i, j) = (a, A), or (A, row) in column mode
    const int i = (row >= 0) ? A : a, j = (row >= 0) ? a : A;
    kind[u] = (i == j) ? 0 : (i > j ? -1 : 1);
  }
  const int qcl = min(q0 + ql, nq - 1);

  f64x2 gx[2], gy[2], gr, gb[4];
  auto fetch = [&](int s0) {
    const int s = min(s0 + ks, ns - 1);
    gr = rf ? rf[s] : f64x2{1.0, 0.0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      gx[u] = vft[offa[u] + s];
      gy[u] = vft[offA[u] + s];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      gb[u] = bt[(size_t)min(s0 + kb0 + 4 * u, ns - 1) * nq + qcl];
  };
  auto stash = [&](int s0, int buf) {
    const bool sval = s0 + ks < ns;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f64x2 p = cmul_conj(gx[u], gy[u]);             // conj(V_a) V_A = entry (a, A)
      if (row >= 0) p.y = -p.y;                      // entry (A, row) of the Hermitian matrix
      // the reference scales the upper-triangular cube by ref and THEN adds the conjugate transpose
      // (:410-413): entries above the diagonal carry ref, those below conj(ref), the autos appear
      // twice (with Re ref)
      const f64x2 r = {gr.x, gr.y * (double)kind[u]};
      p = (kind[u] == 0) ? f64x2{2.0 * p.x * r.x, 0.0} : cmul(p, r);
      sP[buf][ks][ml0 + 32 * u] = (sval && mval[u]) ? p : f64x2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) sB[buf][kb0 + 4 * u][ql] = gb[u];   // finite everywhere; P = 0 masks s >= ns
  };

  f64x4 acc1[2][2], acc2[2][2], acc3[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      acc1[mb][qb] = f64x4{0.0, 0.0, 0.0, 0.0};
      acc2[mb][qb] = f64x4{0.0, 0.0, 0.0, 0.0};
      acc3[mb][qb] = f64x4{0.0, 0.0, 0.0, 0.0};
    }

  auto products = [&](int buf, int kk) {
    f64x2 pa[2], pb[2];
    double psum[2], bsum[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      pa[mb] = sP[buf][4 * kk + kr][(2 * wm + mb) * 16 + lc];
      psum[mb] = pa[mb].x + pa[mb].y;
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      pb[qb] = sB[buf][4 * kk + kr][(2 * wq + qb) * 16 + lc];
      bsum[qb] = pb[qb].x + pb[qb].y;
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        acc1[mb][qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[mb].x, pb[qb].x, acc1[mb][qb], 0, 0, 0);
        acc2[mb][qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[mb].y, pb[qb].y, acc2[mb][qb], 0, 0, 0);
        acc3[mb][qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(psum[mb], bsum[qb], acc3[mb][qb], 0, 0, 0);
      }
  };

  // two LDS buffers, one barrier per chunk: chunk c + 1 is written (from the registers fetched
  // during chunk c - 1) in the middle of chunk c's products, chunk c + 2's loads follow it
  fetch(0);
  stash(0, 0);
  if (BH_KS < ns) fetch(BH_KS);
  int buf = 0;
  for (int s0 = 0; s0 < ns; s0 += BH_KS, buf ^= 1) {
    __syncthreads();   // buffer `buf` is complete; nobody reads the other one any more
#pragma unroll
    for (int kk = 0; kk < BH_KS / 8; ++kk) products(buf, kk);
    if (s0 + BH_KS < ns) {
      stash(s0 + BH_KS, buf ^ 1);
      if (s0 + 2 * BH_KS < ns) fetch(s0 + 2 * BH_KS);
    }
#pragma unroll
    for (int kk = BH_KS / 8; kk < BH_KS / 4; ++kk) products(buf, kk);
  }
  // D layout: lane holds rows kr + 4 g, column lc of each 16 x 16 block
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m = m0 + (2 * wm + mb) * 16 + kr + 4 * g;
      if (m >= mtot) continue;
      f64x2* o = out + ((size_t)ft * mtot + m) * nq;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int q = q0 + (2 * wq + qb) * 16 + lc;
        const double p1 = acc1[mb][qb][g], p2 = acc2[mb][qb][g], p3 = acc3[mb][qb][g];
        if (q < nq) o[q] = f64x2{p1 - p2, p3 - p1 - p2};
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
  // threads <-> (row group g, column B): 256 / BW row groups walk the rows b = g, g + NG, ... of
  // the Nc x NB block with 4 loads in flight each, then the groups are summed through LDS in
  // group order.  (With threads over the columns only, 40 of 256 threads worked at NB = 40.)
  __shared__ f64x2 sc[256];
  __shared__ f64x2 sred[256];
  const int k = blockIdx.x % (nant - 1);
  const int ft = blockIdx.x / (nant - 1);
  const int f = ft / nt;
  const int A = k + (k >= samp ? 1 : 0);
  const f64x2* blk = col_only ? C + ((size_t)ft * nant + A) * nc * nb
                              : C + (((size_t)ft * nant + A) * nant + samp) * nc * nb;
  f64x2* o = out + (size_t)blockIdx.x * nb;
  // column tile width: the smallest power of two >= min(nb, 256)
  int BW = 1;
  while (BW < nb && BW < 256) BW <<= 1;
  const int NG = 256 / BW;
  const int g = threadIdx.x / BW, Bl = threadIdx.x % BW;
  for (int B0 = 0; B0 < nb; B0 += BW) {
    const int B = B0 + Bl;
    const int Bc = min(B, nb - 1);
    f64x2 acc = {0.0, 0.0};
    for (int b0 = 0; b0 < nc; b0 += 256) {
      __syncthreads();
      int bb = b0 + threadIdx.x;
      if (bb < nc) {
        f64x2 c = coef[((size_t)bb * nf + f) * nant + A];
        sc[threadIdx.x] = f64x2{c.x, -c.y};
      }
      __syncthreads();
      const int nbl = min(256, nc - b0);
      for (int b = g; b < nbl; b += 4 * NG) {
        f64x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)      // unconditional, clamped; rows beyond the block are skipped below
          v[u] = blk[(size_t)min(b0 + b + u * NG, nc - 1) * nb + Bc];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (b + u * NG < nbl) {
            f64x2 c = sc[b + u * NG];
            acc.x += c.x * v[u].x - c.y * v[u].y;
            acc.y += c.x * v[u].y + c.y * v[u].x;
          }
        }
      }
    }
    __syncthreads();
    sred[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && B < nb) {
      f64x2 t = sred[Bl];
      for (int q = 1; q < NG; ++q) {
        t.x += sred[q * BW + Bl].x;
        t.y += sred[q * BW + Bl].y;
      }
      o[B] = t;
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
  HYDRA_CHECK_ARG((long long)nant * ns < 2147483647LL && (long long)nant * nant < 2147483647LL,
                  "hydra_bess_sky_contraction: (antenna, source) slice too large for 32-bit offsets");
  const int mtot = row >= 0 ? nant : nant * nant;
  long long ntile_q = (nq + BH_TQ - 1) / BH_TQ;
  long long ntile_m = ((long long)mtot + BH_TM - 1) / BH_TM;
  long long tiles = (long long)nf * nt * ntile_m * ntile_q;
  HYDRA_CHECK_ARG(tiles < 2147483647LL - 8, "hydra_bess_sky_contraction: grid too large");
  unsigned grid = (unsigned)((tiles + 7) / 8 * 8);
  constexpr size_t lds = 2 * (sizeof(f64x2[BH_KS][BH_TM + 1]) + sizeof(f64x2[BH_KS][BH_TQ]));   // 96.5 KiB
  if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&bess_contract_kernel), lds)) return rc_lds;
  bess_contract_kernel<<<grid, 512, lds, as_stream(stream)>>>(
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
