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
// A thread owns one (f,t) sample and GF_AB antennas: the basis values it loads (GF_MC modes at a
// time, in registers) serve all of them; the scaled mode amplitudes of the workgroup's antennas
// are staged in LDS and read as broadcasts.  (One thread per antenna and sample made every
// antenna re-read the whole basis through L2 -- 24 us per application at config 4 -- and with
// the amplitudes fetched by scalar loads inside the loop the kernel waited on those: 38 us.)
// Modes are summed in order.
constexpr int GF_AB = 8, GF_MC = 16;
__global__ void __launch_bounds__(256)
gain_basis_fwd_kernel(const double* __restrict__ x, const f64x2* __restrict__ F,
                      const double* __restrict__ s, int nant, int nmodes, int nft,
                      f64x2* __restrict__ g) {
  __shared__ f64x2 s_z[GF_AB][GF_MC];
  const int ft = blockIdx.x * blockDim.x + threadIdx.x;
  const int ftc = min(ft, nft - 1);
  const int a0 = blockIdx.y * GF_AB, v = blockIdx.z;
  const double* xre = x + (size_t)v * 2 * nant * nmodes;
  const double* xim = xre + (size_t)nant * nmodes;
  f64x2 acc[GF_AB];
#pragma unroll
  for (int q = 0; q < GF_AB; ++q) acc[q] = f64x2{0.0, 0.0};
  for (int m0 = 0; m0 < nmodes; m0 += GF_MC) {
    f64x2 fv[GF_MC];
#pragma unroll
    for (int u = 0; u < GF_MC; ++u) fv[u] = F[(size_t)min(m0 + u, nmodes - 1) * nft + ftc];
    __syncthreads();
    if (threadIdx.x < GF_AB * GF_MC) {
      const int q = threadIdx.x / GF_MC, u = threadIdx.x % GF_MC, m = m0 + u;
      f64x2 z = {0.0, 0.0};
      if (m < nmodes) {
        const size_t o = (size_t)min(a0 + q, nant - 1) * nmodes + m;
        const double sm = s ? s[m] : 1.0;
        z = f64x2{sm * xre[o], sm * xim[o]};
      }
      s_z[q][u] = z;      // modes past the end contribute exact zeros
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GF_MC; ++u)
#pragma unroll
      for (int q = 0; q < GF_AB; ++q) {      // four FMAs per complex multiply-add
        const f64x2 z = s_z[q][u];
        acc[q].x = __builtin_fma(fv[u].x, z.x, acc[q].x);
        acc[q].x = __builtin_fma(-fv[u].y, z.y, acc[q].x);
        acc[q].y = __builtin_fma(fv[u].x, z.y, acc[q].y);
        acc[q].y = __builtin_fma(fv[u].y, z.x, acc[q].y);
      }
  }
  if (ft < nft) {
#pragma unroll
    for (int q = 0; q < GF_AB; ++q)
      if (a0 + q < nant) g[((size_t)v * nant + a0 + q) * nft + ft] = acc[q];
  }
}
static_assert(GF_AB * GF_MC <= 256, "at most one amplitude per thread");

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

// ---------------------------------------------------------------------------------------------
// Streamed form: every weight crosses the fabric ONCE per operator application.
//
// A workgroup owns FT consecutive (f,t) samples and ALL antennas: the fields g (Nant x FT) and the
// accumulators T (Nant x FT) live in LDS, and the weights of its samples arrive as one contiguous
// stream in the order they are used.  That order ("stream layout", written once per operator by
// gain_weights_stream_kernel) is
//   chunk (FT samples) > sweep A (antennas a = 8A .. 8A+7) > sub-tile k > row r > column c' > ft
// where sweep A holds the pairs (a, o) with o = 8A + 1 + c, c = 0 .. L_A - 1, L_A = Nant - 1 - 8A,
// cut into sub-tiles of COLS = 256 / FT columns (the last one shorter, no padding in memory).
// Slots with o <= a, a >= Nant or without a baseline hold 0; every unordered pair has exactly
// one slot (sweep a / 8, row a % 8).  Padding: the 28 slots below the diagonal of each sweep, 2 %.
//
// A sub-tile (16 KiB) is copied to one of NBUF LDS buffers by LDS-DMA, NBUF - 1 sub-tiles ahead;
// the loop holds no other vector-memory operation (the compiler answers a pending LDS-DMA with a
// wait for vmcnt(0) at every dependent use, which would drain the pipeline), so all global reads
// of g / sw and the writes of y sit before and after it.  Per sub-tile:
//   column phase: thread (ft, c) adds sum_r W[r][c] g[8A + r] to T[o]     (one owner per entry)
//   row phase:    wave w, rows 2w and 2w+1, lane (ft, j) sums W[r][c] g[o] over its columns in
//                 registers across the sweep; an xor-shuffle reduction at the end of the sweep
//                 and the owner lanes add the sums to T[a] behind the next barrier.
// Fixed summation order, no atomics.  y = g sw + conj(T) with sw = sum_{b ∋ a} W_b precomputed by
// one application to g = 1 (sw == NULL: y = conj(T)).
template <int FT>
struct GainStream {
  static constexpr int COLS = 256 / FT;          // columns per sub-tile
  static constexpr int TILE = 8 * COLS * FT;     // doubles per full sub-tile (= 2048)
  static constexpr int JW = 64 / FT;             // column lanes per wave in the row phase
};

// doubles of one chunk before sweep A
__host__ __device__ inline long long gs_sweep_base(int A, int nant, int ft) {
  return 8LL * ft * ((long long)A * (nant - 1) - 4LL * A * (A - 1));
}
__host__ __device__ inline int gs_nsweeps(int nant) { return (nant - 1 + 7) / 8; }   // sweeps with L_A > 0
// sub-tiles of one chunk
__host__ __device__ inline int gs_ntiles(int nant, int cols) {
  int n = 0;
  for (int A = 0; A < gs_nsweeps(nant); ++A) n += (nant - 1 - 8 * A + cols - 1) / cols;
  return n;
}
// The stream starts with one descriptor per sub-tile (the same for every chunk), so that the pair
// kernel's loop reads its bookkeeping with one scalar load instead of recomputing it in every
// wave: {byte offset in the chunk, columns | first-of-sweep << 16 | last-of-sweep << 17, first row
// antenna, first column antenna}; header = descriptors padded to 256 B (in doubles below).
__host__ __device__ inline long long gs_header_doubles(int nant, int cols) {
  return (((long long)gs_ntiles(nant, cols) * 16 + 255) / 256) * 32;
}

__global__ void __launch_bounds__(256)
gain_stream_desc_kernel(int nant, int ft, int4* __restrict__ desc) {
  const int cols = 256 / ft;
  int n = 0;
  for (int A = 0; A < gs_nsweeps(nant); ++A) {
    const int L = nant - 1 - 8 * A, nt = (L + cols - 1) / cols;
    for (int k = threadIdx.x; k < nt; k += 256) {
      const int nc = min(cols, L - k * cols);
      const long long off = 8 * (gs_sweep_base(A, nant, ft) + (long long)k * 8 * cols * ft);
      desc[n + k] = int4{(int)off, nc | (k == 0 ? 1 << 16 : 0) | (k == nt - 1 ? 1 << 17 : 0), 8 * A,
                         8 * A + 1 + k * cols};
    }
    n += nt;
  }
}

template <int FT>
__global__ void __launch_bounds__(256)
gain_weights_stream_kernel(const f64x2* __restrict__ mv, const double* __restrict__ w,
                           const int32_t* __restrict__ pairb, int nant, int nft,
                           double* __restrict__ Wt) {
  using G = GainStream<FT>;
  const int A = blockIdx.x, chunk = blockIdx.y;
  const int L = nant - 1 - 8 * A;
  const long long chunk_len = gs_sweep_base(gs_nsweeps(nant), nant, FT);
  double* out = Wt + gs_header_doubles(nant, G::COLS) + (long long)chunk * chunk_len + gs_sweep_base(A, nant, FT);
  const int total = 8 * L * FT;
  for (int e = threadIdx.x; e < total; e += 256) {
    int k = e / G::TILE, e2 = e - k * G::TILE;
    int nc = min(G::COLS, L - k * G::COLS);
    int r = e2 / (nc * FT), rem = e2 - r * nc * FT;
    int c = rem / FT, ft = chunk * FT + (rem - c * FT);
    int a = 8 * A + r, o = 8 * A + 1 + k * G::COLS + c;
    double v = 0.0;
    if (a < o && a < nant && ft < nft) {
      int b = pairb[(size_t)a * nant + o];
      if (b >= 0) {
        f64x2 m = mv[(size_t)b * nft + ft];
        v = (m.x * m.x + m.y * m.y) * (w ? w[(size_t)b * nft + ft] : 1.0);
      }
    }
    out[e] = v;
  }
}

// One pass over the visibilities for everything a gain GCR sample needs from them.  Per baseline
// b = (a, o), a < o, and (f,t) sample, with v = g_a conj(g_o) m_b the gain-applied model
// (utils.py:63-110), r = d_b - v the residual (example.py:690-700) and w the inverse noise
// variance:
//   W_b   = |v|^2 w                                   operator weight, gain_sampler.py:278-284
//   z_b   = r w + omega_b sqrt(w)                     gain_sampler.py:201-204
//   p_b   = conj(v) z_b                               A^H z, gain_sampler.py:109-142:
//   yy[a] += p_b,  yy[o] += conj(p_b),  sw[a] += W_b,  sw[o] += W_b
// W goes straight into the stream layout of the pair kernel (header and slots identical to
// gain_weights_stream_kernel's), yy and sw -- the antenna-space right-hand side before the basis
// projection and the operator's diagonal term -- leave as (Nant, Nft) arrays.  The unfused sequence
// (apply_gains, a subtraction, rhs_rows, proj_conj, weights_stream, one pair-kernel application to
// ones) moves 55 GB at config 4; this pass reads model, data, weights and omega once and writes W
// once: 14.3 GB.
//
// A workgroup owns FT consecutive samples and all antennas, like the pair kernel: gains, yy and sw
// of its samples live in LDS.  Its two halves (256 threads each) take alternate sub-tiles of a
// sweep; thread (ft, c) of a half owns column antenna o of its sub-tile -- it adds the column sums
// of the 8 rows to yy[o], sw[o] (one owner per entry) -- and keeps the 8 row sums in registers
// across the sweep; at the end of the sweep they are reduced over the lanes of equal ft (xor
// shuffles), then over the 8 waves through LDS in wave order.  Fixed summation order, no atomics.
// Every thread has its 8 index loads, then its 32 data loads (one 128-byte line per 8 lanes and
// array at FT = 8) in flight together.
constexpr int GP_THREADS = 512;

template <int FT>
__global__ void __launch_bounds__(GP_THREADS)
gain_prepare_stream_kernel(const f64x2* __restrict__ data, const f64x2* __restrict__ model,
                           const f64x2* __restrict__ gains, const double* __restrict__ w,
                           const f64x2* __restrict__ om, const int32_t* __restrict__ pairb,
                           int nant, int nft, long long header, long long chunk_len,
                           double* __restrict__ Wt, double* __restrict__ sw_out,
                           f64x2* __restrict__ yy) {
  using G = GainStream<FT>;
  constexpr int COLS = G::COLS;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nantp = ((nant + 7) & ~7) + COLS + 8;
  f64x2* s_g = reinterpret_cast<f64x2*>(smem);                 // [nantp][FT] gains
  f64x2* s_t = s_g + (size_t)nantp * FT;                       // [nantp][FT] yy
  double* s_sw = reinterpret_cast<double*>(s_t + (size_t)nantp * FT);   // [nantp][FT]
  double* s_red = s_sw + (size_t)nantp * FT;                   // [8 waves][8 rows][FT][3]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, t = tid & 255;
  const int cft = t % FT, cc = t / FT;
  const int chunk = blockIdx.x, ft0 = chunk * FT, ft = ft0 + cft;
  const bool ftok = ft < nft;

  for (int e = tid; e < nantp * FT; e += GP_THREADS) {
    const int a = e / FT, f = ft0 + (e - a * FT);
    s_g[e] = (a < nant && f < nft) ? gains[(size_t)a * nft + f] : f64x2{0.0, 0.0};
    s_t[e] = f64x2{0.0, 0.0};
    s_sw[e] = 0.0;
  }
  __syncthreads();

  double* wchunk = Wt + header + (long long)chunk * chunk_len;
  const int nsweeps = gs_nsweeps(nant);
  for (int A = 0; A < nsweeps; ++A) {
    const int L = nant - 1 - 8 * A, ntl = (L + COLS - 1) / COLS;
    f64x2 ga[8], racc[8];
    double rsw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      ga[r] = s_g[(size_t)(8 * A + r) * FT + cft];
      racc[r] = f64x2{0.0, 0.0};
      rsw[r] = 0.0;
    }
    for (int k = half; k < ntl; k += 2) {
      const int nc = min(COLS, L - k * COLS);
      if (cc < nc) {
        const int o = 8 * A + 1 + k * COLS + cc;
        int b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int a = 8 * A + r;
          b[r] = (a < o && a < nant) ? pairb[(size_t)a * nant + o] : -1;
        }
        f64x2 m[8], d[8], q[8];
        double ww[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const size_t idx = (b[r] >= 0 && ftok) ? (size_t)b[r] * nft + ft : 0;
          m[r] = model[idx];
          d[r] = data[idx];
          ww[r] = w ? w[idx] : 1.0;
          q[r] = om ? om[idx] : f64x2{0.0, 0.0};
        }
        const f64x2 go = s_g[(size_t)o * FT + cft];
        f64x2 tcol = {0.0, 0.0};
        double swc = 0.0;
        double* wo = wchunk + gs_sweep_base(A, nant, FT) + (long long)k * 8 * COLS * FT + cc * FT + cft;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const bool valid = b[r] >= 0 && ftok;
          const f64x2 v = cmul(cmul_conj(go, ga[r]), m[r]);              // g_a conj(g_o) m
          const double W = valid ? (v.x * v.x + v.y * v.y) * ww[r] : 0.0;
          const double rt = sqrt(ww[r]);
          const f64x2 z = {(d[r].x - v.x) * ww[r] + q[r].x * rt, (d[r].y - v.y) * ww[r] + q[r].y * rt};
          f64x2 p = cmul_conj(v, z);
          if (!valid) p = f64x2{0.0, 0.0};
          racc[r].x += p.x; racc[r].y += p.y; rsw[r] += W;
          tcol.x += p.x; tcol.y -= p.y; swc += W;
          wo[(size_t)r * nc * FT] = W;
        }
        f64x2* tp = s_t + (size_t)o * FT + cft;
        const f64x2 told = *tp;
        *tp = f64x2{told.x + tcol.x, told.y + tcol.y};
        s_sw[(size_t)o * FT + cft] += swc;
      }
    }
    // row sums of the sweep: lanes of equal ft, then the 8 waves in order
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      double x = racc[r].x, y = racc[r].y, u = rsw[r];
#pragma unroll
      for (int off = 32; off >= FT; off >>= 1) {
        x += __shfl_xor(x, off, 64);
        y += __shfl_xor(y, off, 64);
        u += __shfl_xor(u, off, 64);
      }
      if (lane < FT) {
        double* rp = s_red + ((size_t)(wave * 8 + r) * FT + lane) * 3;
        rp[0] = x; rp[1] = y; rp[2] = u;
      }
    }
    __syncthreads();
    if (tid < 8 * FT) {
      const int r = tid / FT, f = tid - r * FT;
      double x = 0.0, y = 0.0, u = 0.0;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) {
        const double* rp = s_red + ((size_t)(wv * 8 + r) * FT + f) * 3;
        x += rp[0]; y += rp[1]; u += rp[2];
      }
      const size_t e = (size_t)(8 * A + r) * FT + f;
      const f64x2 told = s_t[e];
      s_t[e] = f64x2{told.x + x, told.y + y};
      s_sw[e] += u;
    }
    __syncthreads();
  }
  for (int e = tid; e < nant * FT; e += GP_THREADS) {
    const int a = e / FT, f = ft0 + (e - a * FT);
    if (f >= nft) continue;
    yy[(size_t)a * nft + f] = s_t[e];
    sw_out[(size_t)a * nft + f] = s_sw[e];
  }
}

// 8 waves: waves 0-3 run the column phase, waves 4-7 the row phase of the same sub-tile, each in a
// loop of its own (both pass the same barriers).  A SIMD that holds one wave issues at most one
// instruction every four cycles, scalar bookkeeping included: with both phases in the same four
// waves and the tile bookkeeping recomputed per wave the loop body (300 instructions per 16 KiB)
// took longer than the memory system needs to deliver the sub-tile.
constexpr int GS_THREADS = 512;

template <int FT, int NBUF>
__global__ void __launch_bounds__(GS_THREADS)
gain_pairs_stream_kernel(const f64x2* __restrict__ g, const double* __restrict__ Wt,
                         const double* __restrict__ sw, int nant, int nft, int nt_total,
                         long long header, long long chunk_len, f64x2* __restrict__ y) {
  using G = GainStream<FT>;
  extern __shared__ __align__(16) unsigned char smem[];
  const int nantp = ((nant + 7) & ~7) + G::COLS + 8;          // zero pad behind the last antenna
  f64x2* s_g = reinterpret_cast<f64x2*>(smem);                // [nantp][FT]
  f64x2* s_t = s_g + (size_t)nantp * FT;                      // [nantp][FT]
  double* s_w = reinterpret_cast<double*>(s_t + (size_t)nantp * FT);   // [NBUF][8][COLS][FT]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x, ft0 = chunk * FT;
  const int4* __restrict__ desc = reinterpret_cast<const int4*>(Wt);
  const char* wsrc = reinterpret_cast<const char*>(Wt + header + (long long)chunk * chunk_len);

  // sub-tile pn -> LDS buffer pn % NBUF: this wave issues 2 of the 16 DMA instructions (row
  // `wave`, both halves; lanes beyond the row's end stay off -- what they would fetch belongs to
  // later rows -- except lane 0 of the second half, so that every wave issues exactly two
  // instructions per sub-tile and the wait counts below are fixed)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  int pn = 0;
  auto fetch = [&](const i32x4 d) {
    const int rowbytes = (d.y & 0xffff) * FT * 8;
    double* dst = s_w + (size_t)(pn % NBUF) * G::TILE + wave * (G::COLS * FT);
    const char* src = wsrc + d.x + wave * rowbytes + lane * 16;
    if (lane * 16 < rowbytes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    if (lane * 16 + 1024 < rowbytes || lane == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024),
                                       (__attribute__((address_space(3))) void*)(dst + 128), 16, 0, 0);
    ++pn;
  };
  // Descriptors come through scalar LOADS written out by hand (the compiler turns a plain read
  // into a vector load here, and a vector load's wait drains the LDS-DMA queue to zero).
  auto load_desc = [&](int i) {
    i32x4 d;
    const int4* p = desc + i;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(d) : "s"(p) : "memory");
    return d;
  };
  for (int i = 0; i < NBUF - 1 && pn < nt_total; ++i) fetch(load_desc(pn));

  // fields into LDS; T starts at conj(g) sw, so that y = conj(T) at the end.  All global loads
  // of a thread are issued before the first use.
  {
    constexpr int PER = 8;
    for (int e0 = tid; e0 < nantp * FT; e0 += GS_THREADS * PER) {
      f64x2 gv[PER];
      double sv[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = e0 + u * GS_THREADS;
        const int a = e / FT, ft = ft0 + (e - a * FT);
        const bool ok = e < nantp * FT && a < nant && ft < nft;
        const size_t o = ok ? (size_t)a * nft + ft : 0;
        gv[u] = g[o];
        sv[u] = sw ? sw[o] : 0.0;
        if (!ok) { gv[u] = f64x2{0.0, 0.0}; sv[u] = 0.0; }
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = e0 + u * GS_THREADS;
        if (e < nantp * FT) {
          s_g[e] = gv[u];
          s_t[e] = f64x2{gv[u].x * sv[u], -gv[u].y * sv[u]};
        }
      }
    }
  }
  __syncthreads();                 // g, T in place (this also waits for the first fetches: once)

  // sub-tile n has landed once at most (NBUF - 2) later sub-tiles of this wave are in flight; the
  // descriptors of the sub-tile to multiply (returned) and of the one to fetch are loaded under
  // the same wait
  auto arrive = [&](int n) {
    i32x4 dn, dp;
    const int4* pc = desc + n;
    const int4* pp = desc + min(pn, nt_total - 1);
#define HYDRA_GS_ARRIVE(VM)                                                                        \
  asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x0\n\t"                      \
               "s_waitcnt vmcnt(" #VM ") lgkmcnt(0)\n\ts_barrier"                                  \
               : "=&s"(dn), "=&s"(dp) : "s"(pc), "s"(pp) : "memory")
    if (pn - n - 1 >= NBUF - 2) {
      static_assert(NBUF >= 3 && NBUF <= 8, "wait counts below");
      if (NBUF == 3) HYDRA_GS_ARRIVE(2);
      if (NBUF == 4) HYDRA_GS_ARRIVE(4);
      if (NBUF == 5) HYDRA_GS_ARRIVE(6);
      if (NBUF == 6) HYDRA_GS_ARRIVE(8);
      if (NBUF == 7) HYDRA_GS_ARRIVE(10);
      if (NBUF == 8) HYDRA_GS_ARRIVE(12);
    } else {
      HYDRA_GS_ARRIVE(0);
    }
#undef HYDRA_GS_ARRIVE
    if (pn < nt_total) fetch(dp);
    return dn;
  };

  if (wave < 4) {
    // column phase: thread (ft, c) adds sum_r W[r][c] g[A0 + r] to T[O0 + c]
    const int cft = tid % FT, cc = tid / FT;
    f64x2 ga[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) ga[r] = f64x2{0.0, 0.0};
    for (int n = 0; n < nt_total; ++n) {
      const i32x4 d = arrive(n);
      const int nc = d.y & 0xffff;
      if (d.y & (1 << 16)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) ga[r] = s_g[(size_t)(d.z + r) * FT + cft];
      }
      if (cc < nc) {
        const double* wb = s_w + (size_t)(n % NBUF) * G::TILE + tid;
        double wc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) wc[r] = wb[r * (G::COLS * FT)];
        f64x2* tp = s_t + (size_t)(d.w + cc) * FT + cft;
        f64x2 t = *tp;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          t.x += wc[r] * ga[r].x;
          t.y += wc[r] * ga[r].y;
        }
        *tp = t;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  } else {
    // row phase: wave rw owns rows 2 rw, 2 rw + 1 of the sweep; lane (ft, j) sums its columns in
    // registers across the sweep, xor-shuffle reduction at its end, the owner lanes add the sums
    // to T behind the next barrier (no column of a later sweep touches these rows)
    const int rw = wave - 4, rj = lane / FT;
    f64x2 racc[2] = {f64x2{0.0, 0.0}, f64x2{0.0, 0.0}};
    f64x2 rdone[2] = {f64x2{0.0, 0.0}, f64x2{0.0, 0.0}};
    int rdone_a = -1;
    auto flush = [&]() {
      if (rdone_a >= 0 && lane < FT) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          f64x2* t = s_t + (size_t)(rdone_a + 2 * rw + rr) * FT + lane;
          f64x2 v = *t;
          *t = f64x2{v.x + rdone[rr].x, v.y + rdone[rr].y};
        }
      }
      rdone_a = -1;
    };
    for (int n = 0; n < nt_total; ++n) {
      const i32x4 d = arrive(n);
      const int nc = d.y & 0xffff;
      flush();
      const double* wb = s_w + (size_t)(n % NBUF) * G::TILE + (2 * rw) * (G::COLS * FT) + lane;
      const f64x2* gp = s_g + (size_t)d.w * FT + lane;
      double wr[2][4];
      f64x2 go[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        go[k] = gp[k * 64];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) wr[rr][k] = wb[rr * (G::COLS * FT) + k * 64];
      }
      if (nc < G::COLS) {                            // partial sub-tile: columns beyond nc hold junk
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k * G::JW + rj >= nc) wr[0][k] = wr[1][k] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          racc[rr].x += wr[rr][k] * go[k].x;
          racc[rr].y += wr[rr][k] * go[k].y;
        }
      if (d.y & (1 << 17)) {                         // end of the sweep: reduce over the j lanes
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          double x = racc[rr].x, yv = racc[rr].y;
#pragma unroll
          for (int off = 32; off >= FT; off >>= 1) {
            x += __shfl_xor(x, off, 64);
            yv += __shfl_xor(yv, off, 64);
          }
          rdone[rr] = f64x2{x, yv};
          racc[rr] = f64x2{0.0, 0.0};
        }
        rdone_a = d.z;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    flush();
  }
  __syncthreads();
  for (int e = tid; e < nant * FT; e += GS_THREADS) {
    int a = e / FT, ft = ft0 + (e - a * FT);
    if (ft >= nft) continue;
    f64x2 t = s_t[e];
    y[(size_t)a * nft + ft] = f64x2{t.x, -t.y};
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
// One workgroup of 16 waves per pair of antennas (and vector).  A lane owns (f,t) samples -- wave
// w the 64-sample pieces w, w + 16, ... -- and keeps the sums of GA_AB antennas x GA_MC modes in
// registers, so every basis value it loads (coalesced, once per pass) serves both antennas and no
// lane waits on another until the end of the pass; four waves per SIMD cover each other's load
// latency.  Then the lanes' sums are added through LDS, 8 values at a time (each lane writes its
// 8, lane (i, grp) adds 8 rows of column i, three shuffles join the eight groups), and the
// sixteen waves' sums in wave order.  The order is fixed: same bits every run.
// (Before: one workgroup per antenna, a wave per mode, 32 basis loads in flight per lane --
// 54 us per application at config 4, one dependent wave reduction per mode.)
constexpr int GA_AB = 2, GA_MC = 8, GA_WAVES = 16;
constexpr int GA_NVAL = GA_AB * GA_MC * 2;
__global__ void __launch_bounds__(64 * GA_WAVES)
gain_basis_adj_kernel(const f64x2* __restrict__ y, const f64x2* __restrict__ F,
                      const double* __restrict__ s, const double* __restrict__ add, int nant,
                      int nmodes, int nft, double* __restrict__ out) {
  __shared__ double s_tr[GA_WAVES][64][9];
  __shared__ double s_part[GA_WAVES][GA_NVAL];
  const int a0 = blockIdx.x * GA_AB, v = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f64x2* yrow[GA_AB];
#pragma unroll
  for (int q = 0; q < GA_AB; ++q) yrow[q] = y + ((size_t)v * nant + min(a0 + q, nant - 1)) * nft;
  const size_t obase = (size_t)v * 2 * nant * nmodes;
  for (int m0 = 0; m0 < nmodes; m0 += GA_MC) {
    f64x2 acc[GA_AB][GA_MC];
#pragma unroll
    for (int q = 0; q < GA_AB; ++q)
#pragma unroll
      for (int mm = 0; mm < GA_MC; ++mm) acc[q][mm] = f64x2{0.0, 0.0};
    for (int base = wave * 64; base < nft; base += GA_WAVES * 64) {
      const int ft = base + lane, ftc = min(ft, nft - 1);
      f64x2 yv[GA_AB];
#pragma unroll
      for (int q = 0; q < GA_AB; ++q) {
        yv[q] = yrow[q][ftc];
        if (ft >= nft) yv[q] = f64x2{0.0, 0.0};
      }
      f64x2 fv[GA_MC];
#pragma unroll
      for (int mm = 0; mm < GA_MC; ++mm) fv[mm] = F[(size_t)min(m0 + mm, nmodes - 1) * nft + ftc];
#pragma unroll
      for (int mm = 0; mm < GA_MC; ++mm)
#pragma unroll
        for (int q = 0; q < GA_AB; ++q) {    // conj(f) y, four FMAs
          acc[q][mm].x = __builtin_fma(fv[mm].x, yv[q].x, acc[q][mm].x);
          acc[q][mm].x = __builtin_fma(fv[mm].y, yv[q].y, acc[q][mm].x);
          acc[q][mm].y = __builtin_fma(fv[mm].x, yv[q].y, acc[q][mm].y);
          acc[q][mm].y = __builtin_fma(-fv[mm].y, yv[q].x, acc[q][mm].y);
        }
    }
    // lanes -> one sum per wave
#pragma unroll
    for (int b0 = 0; b0 < GA_NVAL; b0 += 8) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = b0 + i;
        const f64x2 c = acc[idx / (2 * GA_MC)][(idx / 2) % GA_MC];
        s_tr[wave][lane][i] = (idx & 1) ? c.y : c.x;
      }
      __syncthreads();
      const int i = lane & 7, grp = lane >> 3;
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += s_tr[wave][grp * 8 + r][i];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (lane < 8) s_part[wave][b0 + lane] = t;
    }
    __syncthreads();
    // waves -> out
    if (tid < GA_AB * GA_MC) {
      const int q = tid / GA_MC, mm = tid - q * GA_MC, m = m0 + mm;
      if (m < nmodes && a0 + q < nant) {
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int w = 0; w < GA_WAVES; ++w) {
          re += s_part[w][(q * GA_MC + mm) * 2];
          im += s_part[w][(q * GA_MC + mm) * 2 + 1];
        }
        const double sm = s ? s[m] : 1.0;
        const size_t o = obase + (size_t)(a0 + q) * nmodes + m;
        const size_t oi = o + (size_t)nant * nmodes;
        out[o] = sm * re + (add ? add[o] : 0.0);
        out[oi] = sm * im + (add ? add[oi] : 0.0);
      }
    }
  }
}
static_assert(GA_NVAL % 8 == 0, "whole batches of 8 values");


// ---------------------------------------------------------------------------------------------
// Batched operator on the FP64 matrix cores (BASELINE config 4: "batched CG matvecs").
//
// For a batch of vectors the pair sums of one (f,t) sample are a real symmetric matrix times a
// block of columns: T[a][j] = sum_o W[{a,o}] G[o][j], j = (vector, re / im) -- 8 vectors are the 16
// columns of one v_mfma_f64_16x16x4_f64.  The weights are stored once more in BLOCK LAYOUT
//   Wb[chunk of GM_FT samples][block (A <= B) of 16 x 16 antennas, row-major upper][f][a][b]
// (zero where there is no pair; diagonal blocks symmetric with a zero diagonal; 8 % more doubles
// than there are pairs at 331 antennas), written once per operator by gain_weights_blocks_kernel.
//
// A workgroup owns GM_FT samples and ALL antennas; the conjugated fields G' = conj(g) of its
// samples sit in LDS as [f][antenna][j] (86 KiB at 331 antennas).  Block ROW X of T belongs to one
// wave (RPW rows per wave), in registers for the whole pass.  Rounds s = 0 .. nb-1 of a round-robin
// tournament: in round s row X meets block P = (s - X) mod nb -- an involution, so the waves that
// own X and P are at the SAME stored block in the same round, one taking it as it is
// (T[X] += W_XP G[P], X <= P), the other transposed (T[P] += W_XP^T G[X]): the block crosses the
// fabric once and the second reader finds it in the CU's vector cache (a barrier per round keeps
// the waves in step).  Either way the MFMA A operand of k-step t is 4 (direct: consecutive, one
// 32-byte piece per lane) or 4 x 1 (transposed: 128-byte rows across 16 lanes) doubles straight
// from global memory, fetched one round ahead; the B operand is G'[16 P + 4 g + t][j] from LDS.
// No atomics, fixed summation order (round order).  y = g sw + conj(T) = sgn_j G' sw + T'.
#ifdef HYDRA_GM_NO_ROUND_SYNC
#define GM_ROUND_SYNC() do {} while (0)
#else
#define GM_ROUND_SYNC() __syncthreads()
#endif
#ifndef HYDRA_GM_FT
#define HYDRA_GM_FT 2
#endif
constexpr int GM_FT = HYDRA_GM_FT;
constexpr int GM_WAVES = 11;
constexpr int GM_THREADS = 64 * GM_WAVES;
constexpr int GM_NV = 8;                      // vectors per pass
constexpr int GM_MAX_RPW = 3;                 // up to 33 blocks = 528 antennas

__host__ __device__ inline long long gm_blk(int A, int B, int nb) {      // A <= B
  return (long long)A * nb - (long long)A * (A - 1) / 2 + (B - A);
}
// entry (a, b) of a 16 x 16 block: sixteen 4 x 4 sub-tiles of one 128-byte line each, so that the 16
// lanes of a quarter-wave touch 4 lines whether they walk a row (b = 4 g + t) or a column of the block
__host__ __device__ inline int gm_elem(int a, int b) {
  return ((((a >> 2) << 2) + (b >> 2)) << 4) + ((a & 3) << 2) + (b & 3);
}

// One workgroup per (block, GM_WB_GROUPS groups of 16 consecutive samples).  The inputs are
// pair-major ((Nbl, Nft): a pair's samples are contiguous), the block layout is sample-major (a
// sample's 256 entries are contiguous), so a group goes through LDS: thread (p, s) = (tid / 16,
// tid % 16) reads sample s of pair slots p, p + 16, ... (256-byte runs of the model, 128-byte runs of
// the noise weights, every line used whole) into tile[s][entry]; then the 16 samples leave as
// 2 KiB rows, 16 bytes per thread.  (The first form had thread <-> pair with 8 samples in a loop:
// 8.4 ms and 56 GB fetched for 5.4 GB of inputs at config 4 -- every 16-byte piece pulled its line
// again.)  gains == NULL: model_vis already carries them.
constexpr int GM_WB_GROUPS = 4;
constexpr int GM_WB_S = 16;                   // samples per group
constexpr int GM_WB_ROW = 256 + 2;            // tile row (doubles): +2 keeps the 16 samples on 16 bank pairs
static_assert(GM_WB_S % GM_FT == 0, "a group holds whole chunks");
__global__ void __launch_bounds__(256)
gain_weights_blocks_kernel(const f64x2* __restrict__ mv, const f64x2* __restrict__ gains,
                           const double* __restrict__ w, const int32_t* __restrict__ pairb, int nant,
                           int nft, int nb, double* __restrict__ Wb) {
  __shared__ __align__(16) double tile[GM_WB_S * GM_WB_ROW];
  __shared__ int s_pair[256];
  const int tid = threadIdx.x;
  int A = 0, B;
  {
    int base = 0;
    const int blk = blockIdx.x;
    while (A + 1 < nb && base + (nb - A) <= blk) { base += nb - A; ++A; }
    B = A + (blk - base);
  }
  const long long nblk = (long long)nb * (nb + 1) / 2;
  {
    const int a = tid >> 4, b = tid & 15;
    const int ag = 16 * A + a, bg = 16 * B + b;
    s_pair[tid] = (ag < nant && bg < nant && ag != bg) ? pairb[(size_t)ag * nant + bg] : -1;
  }
  __syncthreads();
  const int p = tid >> 4, sl = tid & 15;
  const int ngroups = (nft + GM_WB_S - 1) / GM_WB_S;
  for (int grp = blockIdx.y * GM_WB_GROUPS; grp < min((blockIdx.y + 1) * GM_WB_GROUPS, ngroups); ++grp) {
    const int ft0 = grp * GM_WB_S;
    const int ft = min(ft0 + sl, nft - 1);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int slot = it * 16 + p;            // (a, b) = (slot / 16, slot % 16)
      const int pair = s_pair[slot];
      double v = 0.0;
      if (pair >= 0) {
        const f64x2 m = mv[(size_t)pair * nft + ft];
        v = (m.x * m.x + m.y * m.y) * (w ? w[(size_t)pair * nft + ft] : 1.0);
        if (gains) {
          const int ag = 16 * A + (slot >> 4), bg = 16 * B + (slot & 15);
          const f64x2 ga = gains[(size_t)ag * nft + ft], gb = gains[(size_t)bg * nft + ft];
          v *= (ga.x * ga.x + ga.y * ga.y) * (gb.x * gb.x + gb.y * gb.y);
        }
      }
      tile[sl * GM_WB_ROW + gm_elem(slot >> 4, slot & 15)] = (ft0 + sl < nft) ? v : 0.0;
    }
    __syncthreads();
    // out: sample s of the group is entry f = s % GM_FT of chunk (ft0 + s) / GM_FT, 256 doubles
#pragma unroll
    for (int q = 0; q < GM_WB_S * 128 / 256; ++q) {
      const int e = q * 256 + tid;             // 16-byte piece e of the group: sample e / 128, piece e % 128
      const int so = e >> 7, pc = e & 127;
      const long long chunk = (ft0 + so) / GM_FT;
      const int f = (ft0 + so) % GM_FT;
      if (chunk * GM_FT < nft) {               // (the chunk exists; samples beyond nft in it are zeros)
        const f64x2 val = *reinterpret_cast<const f64x2*>(&tile[so * GM_WB_ROW + 2 * pc]);
        *reinterpret_cast<f64x2*>(&Wb[((chunk * nblk + blockIdx.x) * GM_FT + f) * 256 + 2 * pc]) = val;
      }
    }
    __syncthreads();
  }
}

// BLK: the fields come and go in SAMPLE-MAJOR layout Gc / Yc [ft][16 nb antennas][16 columns] (one
// pass of 8 vectors; Gc already conjugated, rows >= Nant and columns >= 2 nvec zero) as
// gain_basis_fwd_blocks_kernel writes and gain_basis_adj_blocks_kernel reads it: a workgroup's
// fields are then one contiguous 84 KiB piece and a lane group's 16 columns one 128-byte line,
// where the (nvec, Nant, Nft) arrays of the other kernels hand a workgroup 32-byte pieces of 2648
// lines.
template <int RPW, bool BLK>
__global__ void __launch_bounds__(GM_THREADS)
gain_pairs_mfma_kernel(const f64x2* __restrict__ g, const double* __restrict__ Wb,
                       const double* __restrict__ sw, int nant, int nft, int nb, int v0, int nvec,
                       int nchunk, f64x2* __restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* s_g = reinterpret_cast<double*>(smem);             // [GM_FT][16 nb][16]
  const int nap = 16 * nb;
  double* s_sw = s_g + (size_t)GM_FT * nap * 16;             // [GM_FT][16 nb]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, gq = lane >> 4;
  // consecutive workgroup ids go round robin to the 8 XCDs: give every XCD a contiguous run of
  // chunks, so that the 4 chunks that share a 128-byte line of g / y meet in one L2
  const int per_x = (nchunk + 7) / 8;
  const int chunk = (blockIdx.x & 7) * per_x + (blockIdx.x >> 3);
  if (chunk >= nchunk) return;
  const int ft0 = chunk * GM_FT;
  const long long nblk = (long long)nb * (nb + 1) / 2;
  const double* wchunk = Wb + (long long)chunk * nblk * GM_FT * 256;

  if (BLK) {
    // one contiguous piece: samples ft0 .. ft0 + GM_FT - 1 (the array is padded to whole chunks)
    const f64x2* src = g + (size_t)ft0 * nap * 8;
    f64x2* dst = reinterpret_cast<f64x2*>(s_g);
    const int total = GM_FT * nap * 8;
    for (int e0 = tid; e0 < total; e0 += 8 * GM_THREADS) {
      f64x2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[min(e0 + u * GM_THREADS, total - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + u * GM_THREADS < total) dst[e0 + u * GM_THREADS] = t[u];
    }
  } else {
    // conj(g) of this chunk -> LDS; antennas >= nant, vectors >= nvec and samples >= nft are zero
    for (int e = tid; e < GM_NV * nap; e += GM_THREADS) {
      const int v = e / nap, a = e - v * nap;
      f64x2 gv[GM_FT];
#pragma unroll
      for (int f = 0; f < GM_FT; ++f) {
        const bool ok = a < nant && v0 + v < nvec && ft0 + f < nft;
        gv[f] = ok ? g[((size_t)(v0 + v) * nant + a) * nft + ft0 + f] : f64x2{0.0, 0.0};
      }
#pragma unroll
      for (int f = 0; f < GM_FT; ++f) {
        double* d = s_g + ((size_t)f * nap + a) * 16 + 2 * v;
        d[0] = gv[f].x;
        d[1] = -gv[f].y;
      }
    }
  }
  // the diagonal term's factors of this chunk
  for (int e = tid; e < GM_FT * nap; e += GM_THREADS) {
    const int f = e / nap, a = e - f * nap;
    s_sw[e] = (sw && a < nant && ft0 + f < nft) ? sw[(size_t)a * nft + ft0 + f] : 0.0;
  }

  // Both ways of taking a block are TWO 16-byte loads per lane and sample, no branch around them
  // (a branch with a different number of loads on either side ended in vmcnt(0) before every
  // batch, and with four 8-byte loads per sample the kernel was bound by the number of vector-memory
  // instructions: 0.53 ms for the loads alone).  As stored, lane (a = lr, g) takes its row of
  // sub-tile (lr / 4, g) -- 32 contiguous bytes.  Transposed, lane (b = lr, g) needs column lr % 4
  // of sub-tile (g, lr / 4): the even lane of a pair loads rows 0, 1 and the odd lane rows 2, 3 of the
  // pair's two columns, and each hands the other its half (two DPP moves per double).
  struct Raw {
    f64x2 v[RPW][GM_FT][2];
  };
  auto load_round = [&](Raw& o, int s) {
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int X = min(wave + GM_WAVES * k, nb - 1);       // (rows past the end repeat the last)
      int P = s - X;
      if (P < 0) P += nb;
      const bool direct = X <= P;
      const int off = direct ? gm_elem(lr, 4 * gq)
                             : ((((gq << 2) + (lr >> 2)) << 4) + ((lr & 1) << 3) + (lr & 2));
      const double* base = wchunk + gm_blk(min(X, P), max(X, P), nb) * GM_FT * 256 + off;
      const int second = direct ? 2 : 4;                      // next pair of the row / next row
#pragma unroll
      for (int f = 0; f < GM_FT; ++f) {
#ifdef HYDRA_GM_NO_LOAD
        o.v[k][f][0] = f64x2{(double)off, 1.0};
        o.v[k][f][1] = f64x2{(double)second, 2.0};
#elif defined(HYDRA_GM_ONE_READER)
        if (direct) {
          o.v[k][f][0] = *reinterpret_cast<const f64x2*>(base + f * 256);
          o.v[k][f][1] = *reinterpret_cast<const f64x2*>(base + f * 256 + second);
        } else {
          o.v[k][f][0] = f64x2{(double)off, 1.0};
          o.v[k][f][1] = f64x2{(double)second, 2.0};
        }
#else
        o.v[k][f][0] = *reinterpret_cast<const f64x2*>(base + f * 256);
        o.v[k][f][1] = *reinterpret_cast<const f64x2*>(base + f * 256 + second);
#endif
      }
    }
  };
  auto swap_pair = [&](double x) {          // the value of lane ^ 1 (quad permutation 1,0,3,2)
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  };
  // the four k-step operands of (row k, sample f) from the raw loads of round s
  auto operands = [&](const Raw& o, int s, int k, int f, double (&w)[4]) {
    const int X = min(wave + GM_WAVES * k, nb - 1);
    int P = s - X;
    if (P < 0) P += nb;
    const f64x2 v0 = o.v[k][f][0], v1 = o.v[k][f][1];
    if (X <= P) {                                            // wave-uniform
      w[0] = v0.x; w[1] = v0.y; w[2] = v1.x; w[3] = v1.y;
    } else {
      const bool odd = lr & 1;
      const double r0 = swap_pair(odd ? v0.x : v0.y), r1 = swap_pair(odd ? v1.x : v1.y);
      w[0] = odd ? r0 : v0.x;
      w[1] = odd ? r1 : v1.x;
      w[2] = odd ? v0.y : r0;
      w[3] = odd ? v1.y : r1;
    }
  };

  f64x4 acc[RPW][GM_FT];
#pragma unroll
  for (int k = 0; k < RPW; ++k)
#pragma unroll
    for (int f = 0; f < GM_FT; ++f) acc[k][f] = f64x4{0.0, 0.0, 0.0, 0.0};
  auto multiply = [&](const Raw& o, int s) {
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int X = wave + GM_WAVES * k;
      if (X >= nb) continue;
      int P = s - X;
      if (P < 0) P += nb;
      const double* gp = s_g + ((size_t)16 * P + 4 * gq) * 16 + lr;
      double w[GM_FT][4];
#pragma unroll
      for (int f = 0; f < GM_FT; ++f) operands(o, s, k, f, w[f]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int f = 0; f < GM_FT; ++f)
#ifdef HYDRA_GM_NO_MFMA
          acc[k][f][0] += w[f][t];
#else
          acc[k][f] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[f][t], gp[((size_t)f * nap + t) * 16],
                                                           acc[k][f], 0, 0, 0);
#endif
    }
  };
  // The fragments of a round are fetched two rounds ahead of their use into three register sets
  // that take turns (a copy from one set to the next would wait for the newest loads: the loop is
  // unrolled by three instead); with three rows per wave three sets do not fit beside the
  // accumulators: two sets, one round ahead.
  Raw fa, fb, fc;
  auto at = [&](int s) { return min(s, nb - 1); };
  load_round(fa, 0);
  if (RPW <= 2) {
    load_round(fb, at(1));
    __syncthreads();                                 // the fields are in place
    for (int s = 0; s < nb; s += 3) {
      load_round(fc, at(s + 2));
      multiply(fa, s);
      GM_ROUND_SYNC();                               // the two readers of a block stay in step
      load_round(fa, at(s + 3));
      if (s + 1 < nb) multiply(fb, s + 1);
      GM_ROUND_SYNC();
      load_round(fb, at(s + 4));
      if (s + 2 < nb) multiply(fc, s + 2);
      GM_ROUND_SYNC();
    }
  } else {
    __syncthreads();
    for (int s = 0; s < nb; s += 2) {
      load_round(fb, at(s + 1));
      multiply(fa, s);
      __syncthreads();
      load_round(fa, at(s + 2));
      if (s + 1 < nb) multiply(fb, s + 1);
      __syncthreads();
    }
  }

  // y[v][a][ft] = g sw + conj(T): lane (j = lr, gq) holds rows a = 16 X + gq + 4 e
  const int v = lr >> 1, ri = lr & 1;
  double* yd = reinterpret_cast<double*>(y);
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int X = wave + GM_WAVES * k;
    if (X >= nb) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int a = 16 * X + gq + 4 * e;
      if (a >= nant) continue;
#pragma unroll
      for (int f = 0; f < GM_FT; ++f) {
        if (ft0 + f >= nft) continue;
        const double gc = s_g[((size_t)f * nap + a) * 16 + lr];          // re, or -im
        const double val = (ri ? -gc : gc) * s_sw[f * nap + a] + acc[k][f][e];
        if (BLK) yd[((size_t)(ft0 + f) * nap + a) * 16 + lr] = val;
        else if (v0 + v < nvec) yd[(((size_t)(v0 + v) * nant + a) * nft + ft0 + f) * 2 + ri] = val;
      }
    }
  }
}

// Gc[ft][a][2 v], [2 v + 1] = Re, -Im of sum_m F[m,ft] s_m (xre[v0+v,a,m] + i xim[v0+v,a,m]) for the
// 8 vectors of a pass, sample-major (rows a >= Nant, vectors >= nvec and the samples that pad the
// last chunk are written as zeros).  A wave owns one antenna and 64 samples: the basis values a
// lane loads serve the 8 vectors, whose scaled amplitudes are LDS broadcasts.
__global__ void __launch_bounds__(256)
gain_basis_fwd_blocks_kernel(const double* __restrict__ x, const f64x2* __restrict__ F,
                             const double* __restrict__ s, int nant, int nmodes, int nft, int nap,
                             int nftp, int v0, int nvec, f64x2* __restrict__ Gc) {
  __shared__ f64x2 s_z[4][GM_NV][GF_MC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ft = blockIdx.x * 64 + lane, ftc = min(ft, nft - 1);
  const int a = blockIdx.y * 4 + wave;
  f64x2 acc[GM_NV];
#pragma unroll
  for (int v = 0; v < GM_NV; ++v) acc[v] = f64x2{0.0, 0.0};
  for (int m0 = 0; m0 < nmodes; m0 += GF_MC) {
    f64x2 fv[GF_MC];
#pragma unroll
    for (int u = 0; u < GF_MC; ++u) fv[u] = F[(size_t)min(m0 + u, nmodes - 1) * nft + ftc];
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * GM_NV * GF_MC; e += 256) {
      const int q = e / (GM_NV * GF_MC), v = (e / GF_MC) % GM_NV, u = e % GF_MC, m = m0 + u;
      const int aa = blockIdx.y * 4 + q;
      f64x2 z = {0.0, 0.0};
      if (m < nmodes && aa < nant && v0 + v < nvec) {
        const double* xre = x + (size_t)(v0 + v) * 2 * nant * nmodes;
        const size_t o = (size_t)aa * nmodes + m;
        const double sm = s ? s[m] : 1.0;
        z = f64x2{sm * xre[o], sm * xre[o + (size_t)nant * nmodes]};
      }
      s_z[q][v][u] = z;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GF_MC; ++u)
#pragma unroll
      for (int v = 0; v < GM_NV; ++v) {
        const f64x2 z = s_z[wave][v][u];
        acc[v].x = __builtin_fma(fv[u].x, z.x, acc[v].x);
        acc[v].x = __builtin_fma(-fv[u].y, z.y, acc[v].x);
        acc[v].y = __builtin_fma(fv[u].x, z.y, acc[v].y);
        acc[v].y = __builtin_fma(fv[u].y, z.x, acc[v].y);
      }
  }
  if (ft < nftp && a < nap) {
    f64x2* row = Gc + ((size_t)ft * nap + a) * 8;
    const bool live = ft < nft;
#pragma unroll
    for (int v = 0; v < GM_NV; ++v) row[v] = live ? f64x2{acc[v].x, -acc[v].y} : f64x2{0.0, 0.0};
  }
}

// The same for up to NM modes with the roles turned round: a thread owns (antenna, vector) and keeps
// its NM scaled amplitudes in registers, the basis values of 64 samples sit in LDS (every lane reads
// the same one: broadcasts), and per sample the 32 antennas x 8 vectors of a workgroup store 4 KiB in
// one piece -- in the kernel above a lane owns a sample, and its eight 16-byte stores of a row of Gc
// land 43 KB from the next lane's (0.10 ms for 176 MB at config 4, most of it the stores).
constexpr int GFB_ANT = 32, GFB_FT = 64;
template <int NM>
__global__ void __launch_bounds__(256)
gain_basis_fwd_rows_kernel(const double* __restrict__ x, const f64x2* __restrict__ F,
                           const double* __restrict__ s, int nant, int nmodes, int nft, int nap,
                           int nftp, int v0, int nvec, f64x2* __restrict__ Gc) {
  __shared__ f64x2 s_f[GFB_FT][NM];
  const int tid = threadIdx.x;
  const int v = tid & 7, a = blockIdx.y * GFB_ANT + (tid >> 3);
  const int ft0 = blockIdx.x * GFB_FT;
  for (int e = tid; e < GFB_FT * NM; e += 256) {
    const int m = e / GFB_FT, fl = e % GFB_FT;
    const f64x2 f = F[(size_t)min(m, nmodes - 1) * nft + min(ft0 + fl, nft - 1)];
    s_f[fl][m] = (m < nmodes) ? f : f64x2{0.0, 0.0};
  }
  f64x2 z[NM];
  {
    const bool live = a < nant && v0 + v < nvec;
    const double* xre = x + (size_t)min(v0 + v, nvec - 1) * 2 * nant * nmodes +
                        (size_t)min(a, nant - 1) * nmodes;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int mc = min(m, nmodes - 1);
      const double sm = s ? s[mc] : 1.0;
      const f64x2 t = f64x2{sm * xre[mc], sm * xre[mc + (size_t)nant * nmodes]};
      z[m] = (live && m < nmodes) ? t : f64x2{0.0, 0.0};
    }
  }
  __syncthreads();
  if (a >= nap) return;
  for (int fl = 0; fl < GFB_FT; ++fl) {
    const int ft = ft0 + fl;
    if (ft >= nftp) break;
    f64x2 acc = {0.0, 0.0};
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const f64x2 f = s_f[fl][m];
      acc.x = __builtin_fma(f.x, z[m].x, acc.x);
      acc.x = __builtin_fma(-f.y, z[m].y, acc.x);
      acc.y = __builtin_fma(f.x, z[m].y, acc.y);
      acc.y = __builtin_fma(f.y, z[m].x, acc.y);
    }
    Gc[((size_t)ft * nap + a) * 8 + v] = (ft < nft) ? f64x2{acc.x, -acc.y} : f64x2{0.0, 0.0};
  }
}

// out[v0+v, a*nmodes+m] = Re, out[v0+v, nant*nmodes + a*nmodes+m] = Im of
//   s_m sum_ft conj(F[m,ft]) y[v,a,ft]  (+ add) with y in the sample-major layout Yc[ft][a][16].
// One workgroup of 16 waves per antenna: wave (half q, mode group mg) sums the samples
// 64 (q + 2 i) + lane for the modes 2 mg, 2 mg + 1 of all 8 vectors (a 128-byte row of Yc per
// sample), the lanes' sums are added through LDS as in gain_basis_adj_kernel and the two halves
// in order: fixed summation order.
constexpr int GAB_MODES = 2, GAB_Q = 2, GAB_MG = 16 / GAB_Q;
__global__ void __launch_bounds__(1024)
gain_basis_adj_blocks_kernel(const f64x2* __restrict__ Yc, const f64x2* __restrict__ F,
                             const double* __restrict__ s, const double* __restrict__ add, int nant,
                             int nmodes, int nft, int nap, int v0, int nvec,
                             double* __restrict__ out) {
  __shared__ double s_tr[16][64][9];
  __shared__ double s_part[16][GAB_MODES * GM_NV * 2];
  const int a = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = wave / GAB_MG, mg = wave % GAB_MG;
  for (int m0 = 0; m0 < nmodes; m0 += GAB_MG * GAB_MODES) {
    const int mb = m0 + GAB_MODES * mg;
    f64x2 acc[GAB_MODES][GM_NV];
#pragma unroll
    for (int mm = 0; mm < GAB_MODES; ++mm)
#pragma unroll
      for (int v = 0; v < GM_NV; ++v) acc[mm][v] = f64x2{0.0, 0.0};
    for (int base = q * 64; base < nft; base += GAB_Q * 64) {
      const int ft = base + lane, ftc = min(ft, nft - 1);
      const f64x2* row = Yc + ((size_t)ftc * nap + a) * 8;
      f64x2 yv[GM_NV], fv[GAB_MODES];
#pragma unroll
      for (int v = 0; v < GM_NV; ++v) yv[v] = row[v];
#pragma unroll
      for (int mm = 0; mm < GAB_MODES; ++mm) fv[mm] = F[(size_t)min(mb + mm, nmodes - 1) * nft + ftc];
      if (ft >= nft) {
#pragma unroll
        for (int v = 0; v < GM_NV; ++v) yv[v] = f64x2{0.0, 0.0};
      }
#pragma unroll
      for (int mm = 0; mm < GAB_MODES; ++mm)
#pragma unroll
        for (int v = 0; v < GM_NV; ++v) {    // conj(f) y, four FMAs
          acc[mm][v].x = __builtin_fma(fv[mm].x, yv[v].x, acc[mm][v].x);
          acc[mm][v].x = __builtin_fma(fv[mm].y, yv[v].y, acc[mm][v].x);
          acc[mm][v].y = __builtin_fma(fv[mm].x, yv[v].y, acc[mm][v].y);
          acc[mm][v].y = __builtin_fma(-fv[mm].y, yv[v].x, acc[mm][v].y);
        }
    }
    // lanes -> one sum per wave, 8 values at a time
#pragma unroll
    for (int b0 = 0; b0 < GAB_MODES * GM_NV * 2; b0 += 8) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = b0 + i;                        // (mm, v, re/im)
        const f64x2 c = acc[idx / (2 * GM_NV)][(idx / 2) % GM_NV];
        s_tr[wave][lane][i] = (idx & 1) ? c.y : c.x;
      }
      __syncthreads();
      const int i = lane & 7, grp = lane >> 3;
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += s_tr[wave][grp * 8 + r][i];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (lane < 8) s_part[wave][b0 + lane] = t;
    }
    __syncthreads();
    // halves -> out: thread (mg, mm, v)
    if (tid < GAB_MG * GAB_MODES * GM_NV) {
      const int g4 = tid / (GAB_MODES * GM_NV), mm = (tid / GM_NV) % GAB_MODES, v = tid % GM_NV;
      const int m = m0 + GAB_MODES * g4 + mm;
      if (m < nmodes && v0 + v < nvec) {
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int qq = 0; qq < GAB_Q; ++qq) {
          re += s_part[qq * GAB_MG + g4][(mm * GM_NV + v) * 2];
          im += s_part[qq * GAB_MG + g4][(mm * GM_NV + v) * 2 + 1];
        }
        const double sm = s ? s[m] : 1.0;
        const size_t o = (size_t)(v0 + v) * 2 * nant * nmodes + (size_t)a * nmodes + m;
        const size_t oi = o + (size_t)nant * nmodes;
        out[o] = sm * re + (add ? add[o] : 0.0);
        out[oi] = sm * im + (add ? add[oi] : 0.0);
      }
    }
    __syncthreads();
  }
}


// Ft[ft][n], n < nmp: Re F[n][ft], nmp <= n < 2 nmp: Im F[n - nmp][ft] (zero for modes >= nmodes and
// samples >= nft): the basis as the B operand of the real product below, a 128-byte line per
// 16 columns.  1 MB at config 4, rebuilt per call.
__global__ void __launch_bounds__(256)
gain_basis_table_kernel(const f64x2* __restrict__ F, int nmodes, int nft, int nmp, int nftp,
                        double* __restrict__ Ft) {
  const size_t total = (size_t)nftp * 2 * nmp;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const int ft = (int)(e / (2 * nmp)), n = (int)(e % (2 * nmp));
    const int m = n < nmp ? n : n - nmp;
    double v = 0.0;
    if (m < nmodes && ft < nft) {
      const f64x2 f = F[(size_t)m * nft + ft];
      v = n < nmp ? f.x : f.y;
    }
    Ft[e] = v;
  }
}

// The adjoint basis transform of a pass of 8 vectors as ONE real product per antenna on the matrix
// cores: D[j][n] = sum_ft Yc[ft][a][j] Ft[ft][n] (16 x 2 nmp, K = Nft), then
//   Re out[v][m] = D[2v][m] + D[2v+1][nmp + m],  Im out[v][m] = D[2v+1][m] - D[2v][nmp + m]
// (conj(F) y summed over the samples).  Both operands are 128-byte lines straight from global
// memory (A: the antenna's row of Yc for each of the 4 samples of a k-step; B: rows of Ft).  A
// workgroup of 8 waves per antenna, wave w the k-steps w, w + 8, ...; the waves' sums are added
// through LDS in wave order.  NCB = 2 nmp / 16 column blocks (<= 4: up to 32 modes).
template <int NCB>
__global__ void __launch_bounds__(512)
gain_basis_adj_mfma_kernel(const double* __restrict__ Yc, const double* __restrict__ Ft,
                           const double* __restrict__ s, const double* __restrict__ add, int nant,
                           int nmodes, int nft, int nap, int v0, int nvec,
                           double* __restrict__ out) {
  __shared__ double s_d[8][NCB][256];
  constexpr int nmp = 8 * NCB;
  const int a = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, gq = lane >> 4;
  f64x4 acc[NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int nks = (nft + 3) / 4;
  constexpr int UN = 4;
  for (int k0 = wave; k0 < nks; k0 += 8 * UN) {
    double av[UN], bv[UN][NCB];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int ft = 4 * (k0 + 8 * u) + gq, ftc = min(ft, nft - 1);
      av[u] = Yc[((size_t)ftc * nap + a) * 16 + lr];
      if (ft >= nft) av[u] = 0.0;
#pragma unroll
      for (int c = 0; c < NCB; ++c) bv[u][c] = Ft[(size_t)ftc * (2 * nmp) + 16 * c + lr];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int c = 0; c < NCB; ++c)
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u][c], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) s_d[wave][c][(gq + 4 * e) * 16 + lr] = acc[c][e];   // [j][n local]
  __syncthreads();
  // thread (v, m): the eight waves in order, then the complex combination
  for (int e = tid; e < GM_NV * nmp; e += 512) {
    const int v = e / nmp, m = e % nmp;
    if (m >= nmodes || v0 + v >= nvec) continue;
    auto D = [&](int j, int n) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += s_d[w][n >> 4][j * 16 + (n & 15)];
      return t;
    };
    const double re = D(2 * v, m) + D(2 * v + 1, nmp + m);
    const double im = D(2 * v + 1, m) - D(2 * v, nmp + m);
    const double sm = s ? s[m] : 1.0;
    const size_t o = (size_t)(v0 + v) * 2 * nant * nmodes + (size_t)a * nmodes + m;
    const size_t oi = o + (size_t)nant * nmodes;
    out[o] = sm * re + (add ? add[o] : 0.0);
    out[oi] = sm * im + (add ? add[oi] : 0.0);
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

template <bool BLK>
static int launch_pairs_mfma(const double* g, int nvec, int nant, int nft, const double* Wb,
                             const double* sw, double* y, int v0, hipStream_t st) {
  const int nb = (nant + 15) / 16;
  const int rpw = (nb + GM_WAVES - 1) / GM_WAVES;
  const int nchunk = (nft + GM_FT - 1) / GM_FT;
  const size_t lds = (size_t)GM_FT * 16 * nb * (16 + 1) * 8;
  const unsigned grid = (unsigned)((nchunk + 7) / 8 * 8);
  const f64x2* gp = reinterpret_cast<const f64x2*>(g);
  f64x2* yp = reinterpret_cast<f64x2*>(y);
#define HYDRA_GM_LAUNCH(R)                                                                        \
  do {                                                                                            \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&gain_pairs_mfma_kernel<R, BLK>), lds))  \
      return rc;                                                                                  \
    gain_pairs_mfma_kernel<R, BLK><<<grid, GM_THREADS, lds, st>>>(gp, Wb, sw, nant, nft, nb, v0,     \
                                                                  nvec, nchunk, yp);               \
  } while (0)
  if (rpw == 1) HYDRA_GM_LAUNCH(1);
  else if (rpw == 2) HYDRA_GM_LAUNCH(2);
  else HYDRA_GM_LAUNCH(3);
#undef HYDRA_GM_LAUNCH
  HYDRA_CHECK_LAUNCH("hydra_gain_pairs_mfma");
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
  HYDRA_CHECK_ARG(nant <= 65535 * GF_AB && nvec <= 65535,
                  "hydra_gain_basis_fwd: nant <= 65535 * 16, nvec <= 65535");
  dim3 grid((nft + 255) / 256, (nant + GF_AB - 1) / GF_AB, nvec);
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
  dim3 grid((nant + GA_AB - 1) / GA_AB, nvec);
  gain_basis_adj_kernel<<<grid, 64 * GA_WAVES, 0, as_stream(stream)>>>(
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

// FT and the LDS buffers of the streamed pair kernel for nant antennas; 0 = does not fit
static int gain_stream_plan(int nant, int* nbuf, size_t* lds) {
  const int fts[3] = {8, 4, 2};
  for (int ft : fts) {
    const int cols = 256 / ft;
    const size_t nantp = (size_t)((nant + 7) & ~7) + cols + 8;
    for (int nb = 4; nb >= 3; --nb) {   // (deeper prefetch measured no faster)
      size_t need = 2 * nantp * ft * 16 + (size_t)nb * 2048 * 8;
      if (need <= 160 * 1024) {
        *nbuf = nb;
        *lds = need;
        return ft;
      }
    }
  }
  return 0;
}

size_t hydra_gain_stream_weight_count(int nant, int nft) {
  int nbuf;
  size_t lds;
  int ft = (nant >= 2 && nft > 0) ? gain_stream_plan(nant, &nbuf, &lds) : 0;
  if (!ft) return 0;
  long long nchunk = (nft + ft - 1) / ft;
  return (size_t)(gs_header_doubles(nant, 256 / ft) +
                  nchunk * gs_sweep_base(gs_nsweeps(nant), nant, ft)) + 512;   // + DMA over-read slack
}

int hydra_gain_weights_stream(const double* model_vis, const double* inv_noise, int nbl, int nant,
                              int nft, const int32_t* pairb, double* Wt, void* stream) {
  HYDRA_CHECK_ARG(model_vis && pairb && Wt && nbl > 0 && nant >= 2 && nft > 0,
                  "hydra_gain_weights_stream: bad arguments");
  int nbuf;
  size_t lds;
  int ft = gain_stream_plan(nant, &nbuf, &lds);
  HYDRA_CHECK_ARG(ft != 0, "hydra_gain_weights_stream: %d antennas do not fit the streamed kernel", nant);
  const int nchunk = (nft + ft - 1) / ft;
  HYDRA_CHECK_ARG(nchunk <= 65535, "hydra_gain_weights_stream: nft too large");
  hipStream_t st = as_stream(stream);
  size_t count = hydra_gain_stream_weight_count(nant, nft);
  hipError_t e = hipMemsetAsync(Wt + (count - 512), 0, 512 * sizeof(double), st);
  if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_gain_weights_stream: %s", hipGetErrorString(e));
  e = hipMemsetAsync(Wt, 0, (size_t)gs_header_doubles(nant, 256 / ft) * sizeof(double), st);
  if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_gain_weights_stream: %s", hipGetErrorString(e));
  gain_stream_desc_kernel<<<1, 256, 0, st>>>(nant, ft, reinterpret_cast<int4*>(Wt));
  dim3 grid(gs_nsweeps(nant), nchunk);
  const f64x2* mv = reinterpret_cast<const f64x2*>(model_vis);
  if (ft == 8) gain_weights_stream_kernel<8><<<grid, 256, 0, st>>>(mv, inv_noise, pairb, nant, nft, Wt);
  else if (ft == 4) gain_weights_stream_kernel<4><<<grid, 256, 0, st>>>(mv, inv_noise, pairb, nant, nft, Wt);
  else gain_weights_stream_kernel<2><<<grid, 256, 0, st>>>(mv, inv_noise, pairb, nant, nft, Wt);
  HYDRA_CHECK_LAUNCH("hydra_gain_weights_stream");
  return HYDRA_OK;
}

size_t hydra_gain_blocks_weight_count(int nant, int nft) {
  if (nant < 2 || nft <= 0) return 0;
  const int nb = (nant + 15) / 16;
  if (nb > GM_WAVES * GM_MAX_RPW || (size_t)GM_FT * 16 * nb * (16 + 1) * 8 > 160 * 1024) return 0;
  const long long nchunk = (nft + GM_FT - 1) / GM_FT;
  return (size_t)(nchunk * ((long long)nb * (nb + 1) / 2) * GM_FT * 256);
}

int hydra_gain_weights_blocks(const double* model_vis, const double* gains, const double* inv_noise,
                              int nbl, int nant, int nft, const int32_t* pairb, double* Wb,
                              void* stream) {
  HYDRA_CHECK_ARG(model_vis && pairb && Wb && nbl > 0 && nant >= 2 && nft > 0,
                  "hydra_gain_weights_blocks: bad arguments");
  HYDRA_CHECK_ARG(hydra_gain_blocks_weight_count(nant, nft) != 0,
                  "hydra_gain_weights_blocks: %d antennas do not fit the block layout", nant);
  const int nb = (nant + 15) / 16;
  const int ngroups = ((nft + GM_WB_S - 1) / GM_WB_S + GM_WB_GROUPS - 1) / GM_WB_GROUPS;
  HYDRA_CHECK_ARG(ngroups <= 65535, "hydra_gain_weights_blocks: nft too large");
  dim3 grid(nb * (nb + 1) / 2, ngroups);
  gain_weights_blocks_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(model_vis), reinterpret_cast<const f64x2*>(gains), inv_noise,
      pairb, nant, nft, nb, Wb);
  HYDRA_CHECK_LAUNCH("hydra_gain_weights_blocks");
  return HYDRA_OK;
}

int hydra_gain_pairs_mfma(const double* g, int nvec, int nant, int nft, const double* Wb,
                          const double* sw, double* y, void* stream) {
  HYDRA_CHECK_ARG(g && Wb && y && nvec > 0 && nant >= 2 && nft > 0,
                  "hydra_gain_pairs_mfma: bad arguments");
  HYDRA_CHECK_ARG(hydra_gain_blocks_weight_count(nant, nft) != 0,
                  "hydra_gain_pairs_mfma: %d antennas do not fit the block layout", nant);
  for (int v0 = 0; v0 < nvec; v0 += GM_NV)
    if (int rc = launch_pairs_mfma<false>(g, nvec, nant, nft, Wb, sw, y, v0, as_stream(stream))) return rc;
  return HYDRA_OK;
}

size_t hydra_gain_operator_blocks_workspace_bytes(int nant, int nft) {
  if (hydra_gain_blocks_weight_count(nant, nft) == 0) return 0;
  const size_t nap = (size_t)(nant + 15) / 16 * 16;
  const size_t nftp = (size_t)(nft + GM_FT - 1) / GM_FT * GM_FT;
  return 2 * nftp * nap * 16 * sizeof(double) +           // Gc and Yc of one pass of 8 vectors
         nftp * 2 * 32 * sizeof(double);                  // the basis as a real (Nft, 2 x <= 32) table
}

int hydra_gain_operator_blocks(const double* x, int nvec, int nant, int nmodes, int nft,
                               const double* Fbasis, const double* pspec_sqrt, const double* Wb,
                               const double* sw, const double* add, double* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  HYDRA_CHECK_ARG(x && Fbasis && Wb && out && nvec > 0 && nant >= 2 && nmodes > 0 && nft > 0,
                  "hydra_gain_operator_blocks: bad arguments");
  const size_t need = hydra_gain_operator_blocks_workspace_bytes(nant, nft);
  HYDRA_CHECK_ARG(need != 0, "hydra_gain_operator_blocks: %d antennas do not fit the block layout", nant);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_gain_operator_blocks: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  const int nap = (nant + 15) / 16 * 16;
  const int nftp = (nft + GM_FT - 1) / GM_FT * GM_FT;
  double* Gc = static_cast<double*>(workspace);
  double* Yc = Gc + (size_t)nftp * nap * 16;
  double* Ft = Yc + (size_t)nftp * nap * 16;
  hipStream_t st = as_stream(stream);
  const f64x2* Fp = reinterpret_cast<const f64x2*>(Fbasis);
  const int nmp = nmodes <= 8 ? 8 : (nmodes <= 16 ? 16 : (nmodes <= 24 ? 24 : 32));
  if (nmodes <= 32) {
    gain_basis_table_kernel<<<512, 256, 0, st>>>(Fp, nmodes, nft, nmp, nftp, Ft);
    HYDRA_CHECK_LAUNCH("hydra_gain_operator_blocks(basis table)");
  }
  for (int v0 = 0; v0 < nvec; v0 += GM_NV) {
    const dim3 grows((nftp + GFB_FT - 1) / GFB_FT, (nap + GFB_ANT - 1) / GFB_ANT);
    if (nmodes <= 16)
      gain_basis_fwd_rows_kernel<16><<<grows, 256, 0, st>>>(
          x, Fp, pspec_sqrt, nant, nmodes, nft, nap, nftp, v0, nvec, reinterpret_cast<f64x2*>(Gc));
    else if (nmodes <= 32)
      gain_basis_fwd_rows_kernel<32><<<grows, 256, 0, st>>>(
          x, Fp, pspec_sqrt, nant, nmodes, nft, nap, nftp, v0, nvec, reinterpret_cast<f64x2*>(Gc));
    else
      gain_basis_fwd_blocks_kernel<<<dim3((nftp + 63) / 64, nap / 4), 256, 0, st>>>(
          x, Fp, pspec_sqrt, nant, nmodes, nft, nap, nftp, v0, nvec, reinterpret_cast<f64x2*>(Gc));
    HYDRA_CHECK_LAUNCH("hydra_gain_operator_blocks(basis)");
    if (int rc = launch_pairs_mfma<true>(Gc, nvec, nant, nft, Wb, sw, Yc, v0, st)) return rc;
    if (nmodes <= 8)
      gain_basis_adj_mfma_kernel<1><<<nant, 512, 0, st>>>(Yc, Ft, pspec_sqrt, add, nant, nmodes, nft, nap, v0, nvec, out);
    else if (nmodes <= 16)
      gain_basis_adj_mfma_kernel<2><<<nant, 512, 0, st>>>(Yc, Ft, pspec_sqrt, add, nant, nmodes, nft, nap, v0, nvec, out);
    else if (nmodes <= 24)
      gain_basis_adj_mfma_kernel<3><<<nant, 512, 0, st>>>(Yc, Ft, pspec_sqrt, add, nant, nmodes, nft, nap, v0, nvec, out);
    else if (nmodes <= 32)
      gain_basis_adj_mfma_kernel<4><<<nant, 512, 0, st>>>(Yc, Ft, pspec_sqrt, add, nant, nmodes, nft, nap, v0, nvec, out);
    else
      gain_basis_adj_blocks_kernel<<<nant, 1024, 0, st>>>(reinterpret_cast<const f64x2*>(Yc), Fp,
                                                          pspec_sqrt, add, nant, nmodes, nft, nap, v0,
                                                          nvec, out);
    HYDRA_CHECK_LAUNCH("hydra_gain_operator_blocks(adjoint basis)");
  }
  return HYDRA_OK;
}

int hydra_gain_prepare_stream(const double* data, const double* model_vis, const double* gains,
                              const double* inv_noise, const double* omega_r, int nbl, int nant,
                              int nft, const int32_t* pairb, double* Wt, double* sw, double* yy,
                              void* stream) {
  HYDRA_CHECK_ARG(data && model_vis && gains && pairb && Wt && sw && yy && nbl > 0 && nant >= 2 && nft > 0,
                  "hydra_gain_prepare_stream: bad arguments");
  int nbuf;
  size_t lds_pair;
  int ft = gain_stream_plan(nant, &nbuf, &lds_pair);
  HYDRA_CHECK_ARG(ft != 0, "hydra_gain_prepare_stream: %d antennas do not fit the streamed kernel", nant);
  const int cols = 256 / ft;
  const size_t nantp = (size_t)((nant + 7) & ~7) + cols + 8;
  const size_t lds = nantp * ft * 40 + (size_t)8 * 8 * ft * 3 * 8;
  HYDRA_CHECK_ARG(lds <= 160 * 1024, "hydra_gain_prepare_stream: %d antennas need %zu bytes of LDS", nant, lds);
  const int nchunk = (nft + ft - 1) / ft;
  hipStream_t st = as_stream(stream);
  size_t count = hydra_gain_stream_weight_count(nant, nft);
  hipError_t e = hipMemsetAsync(Wt + (count - 512), 0, 512 * sizeof(double), st);
  if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_gain_prepare_stream: %s", hipGetErrorString(e));
  e = hipMemsetAsync(Wt, 0, (size_t)gs_header_doubles(nant, cols) * sizeof(double), st);
  if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_gain_prepare_stream: %s", hipGetErrorString(e));
  gain_stream_desc_kernel<<<1, 256, 0, st>>>(nant, ft, reinterpret_cast<int4*>(Wt));
  const long long header = gs_header_doubles(nant, cols);
  const long long chunk_len = gs_sweep_base(gs_nsweeps(nant), nant, ft);
#define HYDRA_GP_LAUNCH(FT_)                                                                       \
  do {                                                                                             \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&gain_prepare_stream_kernel<FT_>), lds)) \
      return rc;                                                                                   \
    gain_prepare_stream_kernel<FT_><<<nchunk, GP_THREADS, lds, st>>>(                              \
        reinterpret_cast<const f64x2*>(data), reinterpret_cast<const f64x2*>(model_vis),           \
        reinterpret_cast<const f64x2*>(gains), inv_noise, reinterpret_cast<const f64x2*>(omega_r), \
        pairb, nant, nft, header, chunk_len, Wt, sw, reinterpret_cast<f64x2*>(yy));                \
  } while (0)
  if (ft == 8) HYDRA_GP_LAUNCH(8);
  else if (ft == 4) HYDRA_GP_LAUNCH(4);
  else HYDRA_GP_LAUNCH(2);
#undef HYDRA_GP_LAUNCH
  HYDRA_CHECK_LAUNCH("hydra_gain_prepare_stream");
  return HYDRA_OK;
}

int hydra_gain_pairs_stream(const double* g, int nant, int nft, const double* Wt, const double* sw,
                            double* y, void* stream) {
  HYDRA_CHECK_ARG(g && Wt && y && nant >= 2 && nft > 0, "hydra_gain_pairs_stream: bad arguments");
  int nbuf;
  size_t lds;
  int ft = gain_stream_plan(nant, &nbuf, &lds);
  HYDRA_CHECK_ARG(ft != 0, "hydra_gain_pairs_stream: %d antennas do not fit the streamed kernel", nant);
  const int nchunk = (nft + ft - 1) / ft;
  const int nt_total = gs_ntiles(nant, 256 / ft);
  const long long header = gs_header_doubles(nant, 256 / ft);
  const long long chunk_len = gs_sweep_base(gs_nsweeps(nant), nant, ft);
  hipStream_t st = as_stream(stream);
  const f64x2* gp = reinterpret_cast<const f64x2*>(g);
  f64x2* yp = reinterpret_cast<f64x2*>(y);
#define HYDRA_GS_LAUNCH(FT_, NB_)                                                                  \
  do {                                                                                             \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&gain_pairs_stream_kernel<FT_, NB_>), lds)) \
      return rc;                                                                                   \
    gain_pairs_stream_kernel<FT_, NB_><<<nchunk, GS_THREADS, lds, st>>>(                          \
        gp, Wt, sw, nant, nft, nt_total, header, chunk_len, yp);        \
  } while (0)
#define HYDRA_GS_CASE(FT_)                                                                         \
  switch (nbuf) {                                                                                  \
    case 3: HYDRA_GS_LAUNCH(FT_, 3); break;                                                        \
    case 4: HYDRA_GS_LAUNCH(FT_, 4); break;                                                        \
    case 5: HYDRA_GS_LAUNCH(FT_, 5); break;                                                        \
    case 6: HYDRA_GS_LAUNCH(FT_, 6); break;                                                        \
    case 7: HYDRA_GS_LAUNCH(FT_, 7); break;                                                        \
    default: HYDRA_GS_LAUNCH(FT_, 8); break;                                                       \
  }
  if (ft == 8) { HYDRA_GS_CASE(8) }
  else if (ft == 4) { HYDRA_GS_CASE(4) }
  else { HYDRA_GS_CASE(2) }
#undef HYDRA_GS_CASE
#undef HYDRA_GS_LAUNCH
  HYDRA_CHECK_LAUNCH("hydra_gain_pairs_stream");
  return HYDRA_OK;
}

}  // extern "C"
