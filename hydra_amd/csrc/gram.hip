// Gram matrix Re(R^H W R) on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), gfx950.
//
// Real SYRK over the stacked real matrix [Re(R'); Im(R')] with R' = diag(rows) R
// (hydra/ptsrc_sampler.py:62-75, hydra/sh_sampler.py:393-398): K = 2M real rows, N = Ns.
// Algorithmic work: 2 * M * Ns * (Ns+1) flops (lower triangle, both real and imaginary
// parts).  Bound: FP64 MFMA.
//
// Layout of one MFMA: D(16x16) += A(16x4) B(4x16); lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15] and holds D[(l>>4) + 4*reg][l&15].  With A[i][k] = X[k][s0+i] and
// B[k][j] = X[k][s0'+j] (X = stacked real operand) both operands have the SAME lane map:
// lane -> (row k0 + (l>>4), column base + (l&15)).  So each lane produces its own operand
// values directly in registers -- from one 16-byte complex load (materialised R) or from
// two LDS reads of antenna voltages (fused form) -- and the real and imaginary parts of the
// one complex value feed two consecutive MFMA k-steps.  No LDS transpose is needed.
//
// Workgroup: 256 threads = 4 waves as 2x2, output tile 128x128, wave tile 64x64
// (4x4 MFMA tiles, 16 accumulators of 4 f64).  Only tiles of the lower triangle are
// computed.  The row (K) range is split over workgroups; every workgroup writes its partial
// tile to a slab and a second kernel sums the slabs in a fixed order (bitwise reproducible,
// no atomics), mirrors the triangle and applies the prior epilogue.
#include <algorithm>
#include "common.h"

namespace hydra {

#ifdef HYDRA_GRAM_STAMPS
// Diagnostic build only (make stamps): cycle shares of the fused kernel's phases, summed by
// wave 0 of workgroup 0.  [0] segment change (barrier wait + LDS fill), [5] shader cycles total,
// [6] 100 MHz ticks total; with -DHYDRA_GRAM_STAMPS_INNER also [1] load issue, [2] MFMA burst,
// [3] VALU batch, [4] k-steps -- those stamps wait for the LDS reads in flight and so change
// what they measure; the per-wave timeline (g_trace) does not.  Never built into the product.
__device__ unsigned long long g_stamps[8];
// per-wave timeline of workgroup 0's third (f,t): [wave][barrier-1 arrival, after barrier 1,
// DMA landed, after barrier 2, loop end]
__device__ unsigned long long g_trace[8][8];
#ifndef HYDRA_STAMP_BLOCK
#define HYDRA_STAMP_BLOCK 0   // the fused kernel's stamped workgroup (config 2: 0 = a diagonal tile, 8 = a full one)
#endif
#define STAMP(var)                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
  __builtin_amdgcn_sched_barrier(0);
#endif

// Optional HIP events recorded on the launch stream immediately before / after the main Gram
// kernel (hydra_profile_events): lets a caller time that kernel alone, without the small
// helper kernels of the same entry point.
static thread_local hipEvent_t g_prof_ev[2] = {nullptr, nullptr};
static inline void profile_record(int which, hipStream_t st) {
  if (g_prof_ev[which]) (void)hipEventRecord(g_prof_ev[which], st);
}

constexpr int TS = 128;            // workgroup tile edge
constexpr int TILE_ELEMS = TS * TS;

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// p -> (ti, tj) with ti >= tj, p = ti (ti+1)/2 + tj
__device__ __forceinline__ void tile_of(int p, int* ti, int* tj) {
  int i = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
  while ((i + 1) * (i + 2) / 2 <= p) ++i;
  while (i * (i + 1) / 2 > p) --i;
  *ti = i;
  *tj = p - i * (i + 1) / 2;
}

// Block id -> (tile, split) such that blocks with equal id % 8 (which share an XCD and its
// L2 under round-robin dispatch) work on the same row range one tile after another.
__device__ __forceinline__ void block_work(int id, int ntiles, int* tile, int* split) {
  if (gridDim.x == (unsigned)ntiles) {  // one split (many tiles): nothing to interleave
    *tile = id;
    *split = 0;
    return;
  }
  int xcd = id & 7, q = id >> 3;
  *tile = q % ntiles;
  *split = (q / ntiles) * 8 + xcd;
}

// Wave tile of MI x NI blocks of 16 x 16 whose first entry is (row0, col0) of the workgroup's tile.
template <int MI, int NI>
__device__ __forceinline__ void store_blocks(double* __restrict__ slab, const f64x4 (&acc)[MI][NI],
                                             int row0, int col0, int lane) {
  int r0 = row0 + (lane >> 4), c0 = col0 + (lane & 15);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        slab[(size_t)(r0 + mi * 16 + 4 * g) * TS + c0 + ni * 16] = acc[mi][ni][g];
}

__device__ __forceinline__ void store_tile(double* __restrict__ slab, const f64x4 (&acc)[4][4],
                                           int wr, int wc, int lane) {
  store_blocks<4, 4>(slab, acc, wr * 64, wc * 64, lane);
}

// ------------------------------------------------------------------ materialised R

// Operands of one k-step (4 rows x this wave's MI row blocks and NI column blocks), real and
// imaginary parts.
template <int MI, int NI>
struct Frag {
  double are[MI], aim[MI], bre[NI], bim[NI];
};

// Software pipeline of one wave.  FP64 VALU work does not hide under an f64 MFMA on gfx950
// (both run on the SIMD's FP64 lanes: tools/mfma_peak.hip measures +18 cycles for the first
// v_fma_f64 placed after an MFMA and ~4.5 for each further one), and with one wave per SIMD a
// wait on LDS / memory idles the matrix pipe.  So each k-step is cut into three phases kept
// apart with sched_barrier: (1) issue the loads of step k+1, (2) the 2 MI NI back-to-back MFMAs of step
// k, which cover the load latency, (3) one batch of VALU turning the loaded values into the
// operands of step k+1.
// DIAG: a tile on the Gram's diagonal; its `lower` waves skip the blocks above their own diagonal.
template <int MI, int NI, bool DIAG, class P>
__device__ __forceinline__ void mfma_pipeline(P& prod, int nsteps, f64x4 (&acc)[MI][NI],
                                              bool lower = false) {
  Frag<MI, NI> cur;
  prod.issue();
  prod.finish(cur);
#ifdef HYDRA_GRAM_STAMPS_INNER
  unsigned long long t0, t1, t2, t3, s_issue = 0, s_mfma = 0, s_valu = 0, s_steps = 0;
#endif
  for (int n = 0; n < nsteps; ++n) {
#ifdef HYDRA_GRAM_STAMPS_INNER
    STAMP(t0)
#endif
    prod.issue();  // past the end: the producer clamps the row and yields zeros
#ifdef HYDRA_GRAM_STAMPS_INNER
    STAMP(t1)
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < MI; ++g) {
      if (DIAG) break;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[g][ni] = mfma(cur.are[g], cur.bre[ni], acc[g][ni]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[g][ni] = mfma(cur.aim[g], cur.bim[ni], acc[g][ni]);
    }
    if (DIAG) {
      // all real-part products, then all imaginary-part ones (the same order per accumulator):
      // with a row of one or two blocks the two would otherwise follow each other directly and
      // wait out the instruction's latency; the blocks above the diagonal under one uniform
      // branch each (two pipelines, one per kind of wave, spilled 345 registers)
#pragma unroll
      for (int g = 0; g < MI; ++g)
#pragma unroll
        for (int ni = 0; ni <= g && ni < NI; ++ni)
          acc[g][ni] = mfma(cur.are[g], cur.bre[ni], acc[g][ni]);
      if (!lower) {
#pragma unroll
        for (int g = 0; g < MI; ++g)
#pragma unroll
          for (int ni = g + 1; ni < NI; ++ni) acc[g][ni] = mfma(cur.are[g], cur.bre[ni], acc[g][ni]);
      }
#pragma unroll
      for (int g = 0; g < MI; ++g)
#pragma unroll
        for (int ni = 0; ni <= g && ni < NI; ++ni)
          acc[g][ni] = mfma(cur.aim[g], cur.bim[ni], acc[g][ni]);
      if (!lower) {
#pragma unroll
        for (int g = 0; g < MI; ++g)
#pragma unroll
          for (int ni = g + 1; ni < NI; ++ni) acc[g][ni] = mfma(cur.aim[g], cur.bim[ni], acc[g][ni]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef HYDRA_GRAM_STAMPS_INNER
    STAMP(t2)
#endif
    prod.finish(cur);
    __builtin_amdgcn_sched_barrier(0);
#ifdef HYDRA_GRAM_STAMPS_INNER
    STAMP(t3)
    s_issue += t1 - t0; s_mfma += t2 - t1; s_valu += t3 - t2; s_steps += 1;
#endif
  }
#ifdef HYDRA_GRAM_STAMPS_INNER
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_stamps[1] += s_issue; g_stamps[2] += s_mfma; g_stamps[3] += s_valu; g_stamps[4] += s_steps;
  }
#endif
}

// Gram of a materialised operator: G = Re((c R)^H (c R)) = sum_rows |c_row|^2 Re(conj(R_row,i) R_row,j),
// a real rank-2M update.  128 x 128 output tile per 4-wave workgroup (wave tile 64 x 64), two
// workgroups per CU; the rows are walked in chunks of 8: both operand strips (8 rows x 128 columns
// x 16 B) and the chunk's row factors are copied by LDS-DMA into one of two LDS stages one chunk
// ahead, so that the four waves share every fetched byte (the first version let every wave load
// its own operands from global memory, 8 loads per lane and k-step: 47-52 % of the matrix peak).
// Lane (i = l & 15, g = l >> 4) of an MFMA needs X[k = 4 step + g][column i]: the complex element
// (row 4 s + g, column i) read with ONE ds_read_b128 feeds two steps, k = (row, re) and (row, im).
// 16 lanes read 256 contiguous bytes (all banks once), the four row groups are separate passes:
// no conflict, no swizzle.  Only |c|^2 survives in the product, so it multiplies the A operands
// alone.  Rows past the split's end are clamped and get factor 0, columns past ns are clamped
// (their products land in entries that the slab sum never reads).
constexpr int GL_KC = 8;                                  // rows per stage
constexpr int GL_STRIP = GL_KC * TS * 16;                 // one strip: 16 KiB
constexpr int GL_STAGE = 2 * GL_STRIP + 256;              // A strip, B strip, row factors

__global__ void __launch_bounds__(256, 2)
gram_kernel(const f64x2* __restrict__ R, const f64x2* __restrict__ rows, long long m, int ns,
            int ntiles, long long rows_per_split, double* __restrict__ slabs) {
  __shared__ __align__(16) unsigned char smem[2 * GL_STAGE];
  int tile, split;
  block_work(blockIdx.x, ntiles, &tile, &split);
  int ti, tj;
  tile_of(tile, &ti, &tj);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, g = lane >> 4;
  const long long k_begin = (long long)split * rows_per_split;
  const long long k_end = min(k_begin + rows_per_split, m);

  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  if (k_begin < k_end) {
    const int nchunk = (int)((k_end - k_begin + GL_KC - 1) / GL_KC);
    // LDS-DMA (hand-written: see gram_uniform_kernel): a strip is 16 pieces of 1 KiB = (row r,
    // 64-column half h); wave w issues pieces w, w + 4, ... of the 32: the first four belong to the
    // A strip, the last four to the B strip.  The per-lane byte offset of a piece (without the
    // row, which is added as a uniform) fits 32 bits for one chunk.
    unsigned off[8];
    int prow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = (wave + 4 * i) & 15;             // piece inside its strip
      const int col0 = (i < 4 ? ti : tj) * TS + (q & 1) * 64 + lane;
      prow[i] = q >> 1;
      off[i] = (unsigned)min(col0, ns - 1) * 16u;
    }
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
    auto issue = [&](int ch, int buf, int i0, int i1) {
      const long long r0 = k_begin + (long long)ch * GL_KC;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i >= i0 && i < i1) {
          const long long row = min(r0 + prow[i], k_end - 1);
          const char* src = reinterpret_cast<const char*>(R + (size_t)row * ns);   // uniform
          const unsigned dst = lds_base + (unsigned)(buf * GL_STAGE + (i < 4 ? 0 : GL_STRIP)) +
                               (unsigned)(((wave + 4 * i) & 15) * 1024);
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                       :: "s"(dst), "v"(off[i]), "s"(src) : "memory");
        }
      }
      if (rows && wave == 0 && i1 == 8) {            // the chunk's 8 row factors: 8 lanes
        const long long row = min(r0 + (lane & 7), k_end - 1);
        const unsigned woff = (unsigned)((row - r0) * 16);
        const char* src = reinterpret_cast<const char*>(rows + r0);
        const unsigned dst = lds_base + (unsigned)(buf * GL_STAGE + 2 * GL_STRIP);
        if (lane < 8)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                       :: "s"(dst), "v"(woff), "s"(src) : "memory");
      }
    };
    issue(0, 0, 0, 8);

    const unsigned char* rdA = smem + (wr * 64 + lr) * 16 + g * (TS * 16);
    const unsigned char* rdB = smem + GL_STRIP + (wc * 64 + lr) * 16 + g * (TS * 16);
    for (int ch = 0; ch < nchunk; ++ch) {
      const int buf = ch & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk ch have landed
      __syncthreads();                                   // ... everyone's; the other stage is free
      const bool more = ch + 1 < nchunk;
      const unsigned char* sa = rdA + buf * GL_STAGE;
      const unsigned char* sb = rdB + buf * GL_STAGE;
      const long long r0 = k_begin + (long long)ch * GL_KC;
      f64x2 a[2][4], b[2][4];
      double w[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          a[h][mi] = *reinterpret_cast<const f64x2*>(sa + h * 4 * (TS * 16) + mi * 256);
          b[h][mi] = *reinterpret_cast<const f64x2*>(sb + h * 4 * (TS * 16) + mi * 256);
        }
        double wv = 1.0;
        if (rows) {
          f64x2 c = *reinterpret_cast<const f64x2*>(smem + buf * GL_STAGE + 2 * GL_STRIP +
                                                    (h * 4 + g) * 16);
          wv = c.x * c.x + c.y * c.y;
        }
        w[h] = (r0 + h * 4 + g < k_end) ? wv : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const double ax = a[h][mi].x * w[h];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma(ax, b[h][ni].x, acc[mi][ni]);
          if (more && h == 0) issue(ch + 1, buf ^ 1, mi, mi + 1);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const double ay = a[h][mi].y * w[h];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma(ay, b[h][ni].y, acc[mi][ni]);
          if (more && h == 0) issue(ch + 1, buf ^ 1, 4 + mi, 5 + mi);
        }
      }
    }
  }
  store_tile(slabs + ((size_t)split * ntiles + tile) * TILE_ELEMS, acc, wr, wc, lane);
}

// Sum the per-split slabs in order, mirror the lower triangle, apply the epilogue:
//   prior_std: A = I + s_i G_ij s_j ; add_diag: A = G + diag(add) ; else G.
__global__ void __launch_bounds__(256)
gram_finish_kernel(const double* __restrict__ slabs, int nsplit, int ntiles, int ns,
                   const double* __restrict__ prior_std, const double* __restrict__ add_diag,
                   double* __restrict__ G, double scale = 1.0) {
  int tile = blockIdx.x;
  int ti, tj;
  tile_of(tile, &ti, &tj);
  for (int e = threadIdx.x; e < TILE_ELEMS; e += blockDim.x) {
    int r = e / TS, c = e % TS;
    int gi = ti * TS + r, gj = tj * TS + c;
    if (gi >= ns || gj >= ns) continue;
    if (ti == tj && c > r) continue;
    double sum = 0.0;
    for (int s = 0; s < nsplit; ++s) sum += slabs[((size_t)s * ntiles + tile) * TILE_ELEMS + e];
    sum *= scale;
    if (prior_std) {
      sum = prior_std[gi] * sum * prior_std[gj];
      if (gi == gj) sum = 1.0 + sum;
    } else if (add_diag && gi == gj) {
      sum += add_diag[gi];
    }
    G[(size_t)gi * ns + gj] = sum;
    if (gi != gj) G[(size_t)gj * ns + gi] = sum;
  }
}

// Split plan shared by the launch and the workspace query.  `nunits` rows (or (f,t) slices)
// are cut into `nsplit` (multiple of 8, see block_work) ranges of `per_split` units each,
// aiming at ~4096 workgroups and at least `min_units` units per range.
// in-place epilogue on a full (ns,ns) matrix (used after a cross-GPU sum of raw Grams)
__global__ void __launch_bounds__(256)
prior_epilogue_kernel(double* __restrict__ G, int ns, const double* __restrict__ prior_std,
                      const double* __restrict__ add_diag) {
  size_t n = (size_t)ns * ns;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    int i = (int)(e / ns), j = (int)(e % ns);
    double v = G[e];
    if (prior_std) {
      // the same operation order on (i,j) and (j,i), and the one gram_finish_kernel uses on the
      // lower triangle: a symmetric G stays symmetric bit for bit
      v = prior_std[i > j ? i : j] * v * prior_std[i > j ? j : i];
      if (i == j) v = 1.0 + v;
    } else if (add_diag && i == j) {
      v += add_diag[i];
    }
    G[e] = v;
  }
}

// The fused kernel's workgroups are long (config 2 with 4096 of them: 67 ms each, 17 rounds of
// 256), so the last round's stragglers are a visible tail: 16 384 workgroups took 1124 ms against
// 1143 (12 288: 1126, 8192: 1133) at the price of 2.1 GB of slabs instead of 0.6.
constexpr long long FUSED_WORKGROUPS = 16384;

static void gram_plan(long long nunits, int ns, int* ntiles, int* nsplit, long long* per_split,
                      long long unit_align, long long min_units, long long target = 4096) {
  int nt1 = (ns + TS - 1) / TS;
  *ntiles = nt1 * (nt1 + 1) / 2;
  if (*ntiles >= 16384) {  // enough tiles to fill the chip evenly: no K split, slabs = the tiles
    *nsplit = 1;
    *per_split = (nunits + unit_align - 1) / unit_align * unit_align;
    return;
  }
  long long want = std::max<long long>(8, target / *ntiles);
  want = (want + 7) / 8 * 8;
  long long cap = std::max<long long>(8, (nunits / min_units) / 8 * 8);
  want = std::min<long long>(std::min<long long>(want, cap), 512);
  long long per = (nunits + want - 1) / want;
  per = (per + unit_align - 1) / unit_align * unit_align;
  *nsplit = (int)want;
  *per_split = per;
}

// ------------------------------------------------------------------ fused (factored) form

// Row factor of every baseline row and (f,t) sample, in the row order of the segment table:
//   out[ft][row] = inv_noise[b][ft] |g_i|^2 |g_j|^2 .
// Only |c_r|^2 of the reference's complex row factor c_r = g_i conj(g_j) sqrt(w) survives in
// Re(conj(R'_r,s) R'_r,s'), so the gains never touch the voltages: the Gram kernel copies raw V
// into LDS by DMA and multiplies its A operands by this one real number per row.
__global__ void __launch_bounds__(256)
row_weights_kernel(const double* __restrict__ w, const f64x2* __restrict__ gains,
                   const int32_t* __restrict__ seg, const int32_t* __restrict__ rowtab, int nbl,
                   int nft, double* __restrict__ out) {
  const int32_t* sd = seg + 6 * blockIdx.y;
  const int i0 = sd[0], ni = sd[1], j0 = sd[2], row0 = sd[4], nrows = sd[5];
  size_t n = (size_t)nrows * nft;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < n;
       o += (size_t)gridDim.x * blockDim.x) {
    int ft = (int)(o / nrows), r = (int)(o % nrows);
    const int32_t* rt = rowtab + 3 * (size_t)(row0 + r);
    double v = w ? w[(size_t)rt[2] * nft + ft] : 1.0;
    if (gains) {
      int ai = rt[0] < ni ? i0 + rt[0] : j0 + (rt[0] - ni);
      int aj = rt[1] < ni ? i0 + rt[1] : j0 + (rt[1] - ni);
      f64x2 gi = gains[(size_t)ai * nft + ft], gj = gains[(size_t)aj * nft + ft];
      v *= (gi.x * gi.x + gi.y * gi.y) * (gj.x * gj.x + gj.y * gj.y);
    }
    out[(size_t)ft * nbl + row0 + r] = v;
  }
}

// Which k-steps (groups of 4 baseline rows) a wave group takes.  One group: all of them.  Two
// groups share a SIMD pairwise, and the matrix pipe serves the older wave (group 0) first: the
// younger only gets the gaps the older leaves while it reads LDS and forms operands.  With
// alternate k-steps group 0 finished a segment 20 % of the segment's time before group 1 and
// idled at the barrier (per-wave timeline of tools/gram_stamps.py, config 2: loops of 320 k and
// 400 k cycles).  The split follows the measured rates instead: of every 5 k-steps group 0 takes
// numbers 0, 2, 4 and group 1 takes 1, 3.  The assignment is static, so the summation order --
// and with it every bit of the result -- stays the same from run to run.
template <int NG, int M>
struct KSplit {
  static constexpr int P = 2 * M + 1;   // M = 0: plain alternation
  __device__ static __forceinline__ int step(int grp, int n) {          // n-th k-step of a group
    if (NG == 1) return n;
    if (M == 0) return 2 * n + grp;
    return grp == 0 ? P * (n / (M + 1)) + 2 * (n % (M + 1)) : P * (n / M) + 1 + 2 * (n % M);
  }
  __device__ static __forceinline__ int count(int grp, int nk) {        // of nk k-steps in all
    if (NG == 1) return nk;
    if (M == 0) return (nk + 1 - grp) / 2;
    int rem = nk % P;
    return grp == 0 ? (M + 1) * (nk / P) + (rem + 1) / 2 : M * (nk / P) + rem / 2;
  }
};
// (The 5 x 2 wave tiles of a ragged last band were timed with M = 0, 1, 2, 3, 4: no difference.)
constexpr int KSPLIT_M = 2;

template <int NG, int MI, int NI>
struct FusedProducer {
  const f64x2* s_u;
  const double* s_w;
  const uint32_t* s_info;
  int nrows, kr, offA, offB, grp;
  int ksub, kpar;       // this wave takes k-steps ksub * (its group's n-th) + kpar  (1, 0: all)
  int n_next;           // ordinal (within the group) of the k-step whose row info is fetched next
  uint32_t info_next;   // packed LDS rows of the step after the one being issued
  double w_next, w;
  f64x2 ai[MI], aj[MI], bi[NI], bj[NI];

  __device__ __forceinline__ void prefetch_info() {
    int r = 4 * (ksub * KSplit<NG, KSPLIT_M>::step(grp, n_next) + kpar) + kr;
    ++n_next;
    bool valid = r < nrows;
    info_next = s_info[valid ? r : 0];
    w_next = valid ? s_w[r] : 0.0;
  }
  __device__ __forceinline__ void start() { n_next = 0; prefetch_info(); }
  // LDS reads of the k-step whose row info was prefetched; then prefetch the next step's info
  __device__ __forceinline__ void issue() {
    const f64x2* pi = s_u + (info_next & 0xffffu) * 256;
    const f64x2* pj = s_u + (info_next >> 16) * 256;
    w = w_next;
#pragma unroll
    for (int g = 0; g < MI; ++g) { ai[g] = pi[offA + g * 16]; aj[g] = pj[offA + g * 16]; }
#pragma unroll
    for (int g = 0; g < NI; ++g) { bi[g] = pi[offB + g * 16]; bj[g] = pj[offB + g * 16]; }
    prefetch_info();
  }
  // P[r,s] = conj(v_i,s) v_j,s from the voltages staged in LDS; the real row factor
  // w_r |g_i g_j|^2 goes wholly on the A side (sum_r c_r P_r[s] P_r[s'] needs it once)
  __device__ __forceinline__ void finish(Frag<MI, NI>& f) const {
#pragma unroll
    for (int g = 0; g < MI; ++g) {
      f64x2 ui = ai[g];
      ui.x *= w; ui.y *= w;
      f64x2 y = cmul_conj(ui, aj[g]);
      f.are[g] = y.x; f.aim[g] = y.y;
    }
#pragma unroll
    for (int g = 0; g < NI; ++g) {
      f64x2 z = cmul_conj(bi[g], bj[g]);
      f.bre[g] = z.x; f.bim[g] = z.y;
    }
  }
};

// LDS: v[nrows_lds][256] complex (cols 0..127: A side = row-tile sources; 128..255: B side)
//      info[max_rows] uint32 (ldsrow_i | ldsrow_j << 16), sw[max_rows] f64
// The register cap of 256 (2 waves per SIMD) makes hipcc keep the MFMA accumulators in
// ArchVGPRs: on gfx950 an f64 MFMA whose C/D operand sits in AGPRs issues at about half rate
// (tools/mfma_peak.hip: 138 vs 64 cycles per v_mfma_f64_16x16x4_f64).
//
// NG = number of wave groups of 4 waves.  With many antennas the LDS staging leaves room for
// one workgroup per CU only; NG = 2 then puts two waves on every SIMD that share the staged
// voltages and take alternate k-steps of the same 128x128 tile, so one wave's LDS reads,
// waits and VALU batch run under the other's MFMA burst.  Group 1 hands its accumulators to
// group 0 through LDS at the end (fixed order: group 0 + group 1).
//
// Staging.  Between two segments the matrix pipe has nothing to do (the LDS rows are being
// replaced), so the fill must be ONE memory round trip: every wave copies its quarter of the
// 256 columns (64 lanes x 16 B = 1 KiB per instruction) of its antennas straight from V by
// LDS-DMA, all instructions in flight together, no registers, no VALU; the row factors come in
// row order from row_weights_kernel, the packed LDS rows do not depend on (f,t) and are written
// once when there is a single segment.  (The first version multiplied the gains in while
// staging through registers, 8 loads at a time, and chased rowtab -> baseline -> weight: 5-7
// dependent round trips, 19.5 % of the kernel at config 2 by tools/gram_stamps.py.)
// Columns past ns are clamped: their products land in entries the slab sum never reads.
// One workgroup's tile.  MI x NI = blocks of 16 x 16 per wave: 4 x 4 with the four waves of a
// group as 2 x 2 (a 128 x 128 tile), or 5 x 2 with the waves side by side (80 x 128) for the last
// row band when at most 80 of its 128 rows exist (config 2: 2000 = 15 x 128 + 80 -- 16 of the 136
// tiles then cost 20 instead of 32 matrix instructions per k-step).
// DIAG (a 128 x 128 tile on the Gram's diagonal): the upper right wave tile is never read, so its
// wave (1) joins wave 2 on the lower left one -- odd and even k-steps, summed through LDS at the
// end -- and waves 0 and 3 compute only the 10 blocks on and below their own diagonal: 20, 16, 16
// and 20 matrix instructions per k-step on the four SIMDs instead of 32 on three and none on one.
template <int NG, int MI, int NI, bool DIAG>
__device__ __forceinline__ void
fused_tile(const f64x2* __restrict__ V, int nft, int nant, int ns,
           const double* __restrict__ wrow, int nbl,
           const int32_t* __restrict__ seg, int nseg, const int32_t* __restrict__ rowtab,
           int max_lds_rows, int max_seg_rows, int ntiles, int ft_per_split,
           double* __restrict__ slabs, int tile, int split, int ti, int tj) {
  extern __shared__ __align__(16) unsigned char smem[];
  f64x2* s_u = reinterpret_cast<f64x2*>(smem);
  double* s_sw = reinterpret_cast<double*>(smem + (size_t)max_lds_rows * 256 * 16);
  uint32_t* s_info = reinterpret_cast<uint32_t*>(s_sw + max_seg_rows);
  constexpr int NTHREADS = 256 * NG;
  static_assert((MI == 4 && NI == 4) || (MI == 5 && NI == 2 && !DIAG), "wave tile shapes");

  int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;
  const int row0w = DIAG ? (wave == 0 ? 0 : 64) : MI == 4 ? (wave >> 1) * 64 : 0;
  const int col0w = DIAG ? (wave == 3 ? 64 : 0) : MI == 4 ? (wave & 1) * 64 : wave * 32;
  int ft_begin = split * ft_per_split;
  int ft_end = ft_begin + ft_per_split;
  if (ft_end > nft) ft_end = nft;
#ifdef HYDRA_GRAM_STAMPS
  unsigned long long k_c0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  f64x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  // this wave's quarter of the staged columns, and where it lands in an LDS row
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0 .. 4 NG - 1
  const int sq = swave & 3;
  const int scol = (sq < 2 ? ti * TS + sq * 64 : tj * TS + (sq - 2) * 64) + lane;
  const unsigned svoff = (unsigned)min(scol, ns - 1) * 16u;
  const unsigned lds_q = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_u + sq * 64));
  const int scid = tid & 255;
  const int kr = lane >> 4;
  const int offA = row0w + (lane & 15);
  const int offB = 128 + col0w + (lane & 15);
  const bool lower = DIAG && (sq == 0 || sq == 3);     // wave-uniform
  const int ksub = (DIAG && (sq == 1 || sq == 2)) ? 2 : 1;
  const int kpar = (DIAG && sq == 1) ? 1 : 0;

  for (int ft = ft_begin; ft < ft_end; ++ft) {
    const f64x2* vft = V + (size_t)ft * nant * ns;
    for (int sg = 0; sg < nseg; ++sg) {
      const int32_t* sd = seg + 6 * sg;
      int i0 = sd[0], ni = sd[1], j0 = sd[2], nj = sd[3], row0 = sd[4], nrows = sd[5];
      int nload = (j0 == i0) ? ni : ni + nj;
#ifdef HYDRA_GRAM_STAMPS
      unsigned long long f0, f1, e1, e2, e5;
      STAMP(f0)
#endif
      __syncthreads();  // previous segment fully consumed
#ifdef HYDRA_GRAM_STAMPS
      STAMP(e1)
#endif
      for (int a = swave >> 2; a < nload; a += NG) {
        int ant = (a < ni) ? i0 + a : j0 + (a - ni);
        const char* src = reinterpret_cast<const char*>(vft + (size_t)ant * ns);   // uniform
        const unsigned dst = lds_q + (unsigned)a * 4096u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst), "v"(svoff), "s"(src) : "memory");
      }
      if (nseg > 1 || ft == ft_begin) {
        for (int r = tid; r < nrows; r += NTHREADS) {
          const int32_t* rt = rowtab + 3 * (size_t)(row0 + r);
          s_info[r] = (uint32_t)rt[0] | ((uint32_t)rt[1] << 16);
        }
      }
      const double* wft = wrow + (size_t)ft * nbl + row0;
      for (int r = tid; r < nrows; r += NTHREADS) s_sw[r] = wft[r];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA is invisible to the compiler
#ifdef HYDRA_GRAM_STAMPS
      STAMP(e2)
#endif
      __syncthreads();
#ifdef HYDRA_GRAM_STAMPS
      STAMP(f1)
      if (blockIdx.x == HYDRA_STAMP_BLOCK && threadIdx.x == 0) g_stamps[0] += f1 - f0;
#endif

      FusedProducer<NG, MI, NI> prod;
      prod.s_u = s_u; prod.s_w = s_sw; prod.s_info = s_info;
      prod.nrows = nrows; prod.kr = kr; prod.offA = offA; prod.offB = offB;
      prod.grp = swave >> 2;
      prod.ksub = ksub; prod.kpar = kpar;
      prod.start();
      const int nk = ((nrows + 3) / 4 - kpar + ksub - 1) / ksub;   // k-steps of this wave's tile share
      mfma_pipeline<MI, NI, DIAG>(prod, KSplit<NG, KSPLIT_M>::count(swave >> 2, nk), acc, lower);
#ifdef HYDRA_GRAM_STAMPS
      STAMP(e5)
      if (blockIdx.x == HYDRA_STAMP_BLOCK && lane == 0 && ft == ft_begin + 2 && sg == 0) {
        g_trace[swave][0] = f0; g_trace[swave][1] = e1; g_trace[swave][2] = e2;
        g_trace[swave][3] = f1; g_trace[swave][4] = e5;
      }
#endif
    }
  }
  if (NG == 2) {
    // group 1 -> LDS -> group 0; slot index keeps the 256 lanes of a group on consecutive words
    double* s_red = reinterpret_cast<double*>(smem);
    __syncthreads();
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) s_red[((i * NI + j) * 4 + g) * 256 + scid] = acc[i][j][g];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][j][g] += s_red[((i * NI + j) * 4 + g) * 256 + scid];
    }
  }
  if (DIAG) {
    // wave 1 -> LDS -> wave 2 (group 0 holds the groups' sum by now)
    double* s_red = reinterpret_cast<double*>(smem);
    __syncthreads();
    if (grp == 0 && wave == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) s_red[((i * NI + j) * 4 + g) * 64 + lane] = acc[i][j][g];
    }
    __syncthreads();
    if (grp == 0 && wave == 2) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[i][j][g] += s_red[((i * NI + j) * 4 + g) * 64 + lane];
    }
  }
  if (grp == 0 && !(DIAG && wave == 1))
    store_blocks<MI, NI>(slabs + ((size_t)split * ntiles + tile) * TILE_ELEMS, acc, row0w, col0w,
                         lane);
#ifdef HYDRA_GRAM_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_stamps[5] += __builtin_amdgcn_s_memtime() - k_c0;
    g_stamps[6] += __builtin_amdgcn_s_memrealtime() - k_r0;
  }
#endif
}

template <int NG>
__global__ void __launch_bounds__(256 * NG, 2 / NG)
gram_fused_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns,
                  const double* __restrict__ wrow, int nbl,
                  const int32_t* __restrict__ seg, int nseg, const int32_t* __restrict__ rowtab,
                  int max_lds_rows, int max_seg_rows, int ntiles, int ft_per_split,
                  double* __restrict__ slabs) {
  // Workgroup -> (tile, split).  The three kinds of tile differ in cost (measured per (f,t) at
  // config 2: full 1, diagonal 0.68, short last band 0.67), and workgroups are handed out in index
  // order, round robin over the XCDs: with the kinds interleaved a CU that finishes a cheap tile
  // waits for the XCD whose turn it is.  So the kinds are queued one after the other, the dearest
  // first (which also shortens the tail); inside a kind the order is block_work's -- an XCD walks
  // the tiles of one (f,t) range after the other.
  const int nt1 = (ns + TS - 1) / TS;
  const int last_rows = ns % TS;
  const bool short_band = last_rows != 0 && last_rows <= 80;
  int tile, split, ti, tj, kind;   // kind 0 full, 1 diagonal, 2 short last band
  if (gridDim.x == (unsigned)ntiles) {
    tile = blockIdx.x; split = 0;
    tile_of(tile, &ti, &tj);
    kind = (short_band && ti == nt1 - 1) ? 2 : (ti == tj ? 1 : 0);
  } else {
    const int xcd = blockIdx.x & 7, spx = gridDim.x / (8 * ntiles);   // splits per XCD
    int q = blockIdx.x >> 3;
    const int nsq = short_band ? nt1 - 1 : nt1;        // rows of the square-tile part
    const int n_full = nsq * (nsq - 1) / 2, n_diag = nsq;
    int c;
    if (q < n_full * spx) {
      kind = 0; c = q % n_full; split = (q / n_full) * 8 + xcd;
      tile_of(c, &ti, &tj);                            // strictly lower: (a, b), b <= a  ->  (a + 1, b)
      ti += 1;
    } else if ((q -= n_full * spx) < n_diag * spx) {
      kind = 1; c = q % n_diag; split = (q / n_diag) * 8 + xcd;
      ti = tj = c;
    } else {
      q -= n_diag * spx;
      kind = 2; c = q % nt1; split = (q / nt1) * 8 + xcd;
      ti = nt1 - 1; tj = c;
    }
    tile = ti * (ti + 1) / 2 + tj;
  }
  if (kind == 2)   // uniform over the workgroup
    fused_tile<NG, 5, 2, false>(V, nft, nant, ns, wrow, nbl, seg, nseg, rowtab, max_lds_rows,
                                max_seg_rows, ntiles, ft_per_split, slabs, tile, split, ti, tj);
  else if (kind == 1)
    fused_tile<NG, 4, 4, true>(V, nft, nant, ns, wrow, nbl, seg, nseg, rowtab, max_lds_rows,
                               max_seg_rows, ntiles, ft_per_split, slabs, tile, split, ti, tj);
  else
    fused_tile<NG, 4, 4, false>(V, nft, nant, ns, wrow, nbl, seg, nseg, rowtab, max_lds_rows,
                                max_seg_rows, ntiles, ft_per_split, slabs, tile, split, ti, tj);
}


// ------------------------------------------------------------------ uniform-weight form
//
// When every antenna pair i<j is present exactly once and the noise weight of an (f,t) sample is
// the same for all baselines, the Gram has an exact closed form in the antenna voltages
// (R'_ij,s = conj(u_i,s) u_j,s, so conj(R'_ij,s) R'_ij,s' = P_i conj(P_j), P_a = u_a,s conj(u_a,s')):
//
//   G[s,s'] = sum_ft 1/2 ( |T_ft[s,s']|^2 - sum_a q_a,s q_a,s' ),
//   T_ft = sum_a u_a,s conj(u_a,s'),  q_a,s = |u_a,s|^2,  u = w_ft^(1/4) |g_a| v_a
//
// i.e. per (f,t) a Hermitian rank-Nant update on the matrix cores (K = antennas, not baselines)
// followed by an element-wise square-accumulate.  T is kept in registers and never stored.
// The scaled voltages of 20 antennas are one LDS buffer; two buffers are filled alternately by
// LDS-DMA (global_load_lds_dwordx4, no registers, no VALU) one phase ahead of the MFMA loop
// that reads the other, so staging hides under the matrix work.
// Tile shape knobs (defaults measured best on config 2: 8 waves, 96 x 128, 20 antennas per buffer,
// one workgroup per CU = 135 ms; the 4-wave 96 x 64 tile with two workgroups per CU and 12
// antennas per buffer, -DHYDRA_K3_WC=2 -DHYDRA_K3_UCH=12, takes 151 ms: more barriers per (f,t)
// and 43 % more staging traffic outweigh the overlap of one workgroup's barriers with the other).
#ifndef HYDRA_K3_UCH
#define HYDRA_K3_UCH 20
#endif
#ifndef HYDRA_K3_WC
#define HYDRA_K3_WC 4
#endif
constexpr int UCH = HYDRA_K3_UCH;  // antennas per LDS buffer (k-steps of 4)

// U[ft][a][s] = V[ft][a][s] * w_ft^(1/4) * |g_a,ft| for a < nant, 0 for the pad rows a >= nant.
// same_beam: every antenna has the same |v_a,s| (one beam model), so
//   sum_a q_a,s q_a,s' = (sum_a f_a^4) p_s p_s',  p_s = |v_0,s|^2,
// and row `nant` carries (sqrt(sum_a f_a^4) p_s, 0): the kernel subtracts that rank-1 term with
// one MFMA per block and (f,t) instead of one per k-step.
// The factors are formed once per (f,t) and antenna (ufactor_kernel: two square roots each) and
// the scaling pass only streams: with the square roots inside it, recomputed for every source,
// the pass took 6.4 ms at config 2 -- 5 % of a sample -- for 3.1 GB of traffic.
__global__ void __launch_bounds__(256)
ufactor_kernel(const f64x2* __restrict__ gains, const double* __restrict__ wft, int nft, int nant,
               int fstride, double* __restrict__ f) {
  int ft = blockIdx.x * blockDim.x + threadIdx.x;
  if (ft >= nft) return;
  double w4 = wft ? sqrt(sqrt(wft[ft])) : 1.0;
  double c4 = 0.0;
  for (int a = 0; a < nant; ++a) {
    double fa = w4;
    if (gains) {
      f64x2 g = gains[(size_t)a * nft + ft];
      fa *= sqrt(g.x * g.x + g.y * g.y);
    }
    f[(size_t)ft * fstride + a] = fa;
    double f2 = fa * fa;
    c4 += f2 * f2;
  }
  f[(size_t)ft * fstride + nant] = sqrt(c4);
}

// One workgroup streams one antenna row of one (f,t) -- Ns contiguous complex values in, the same
// out -- eight 16-byte loads in flight per lane.  (With a workgroup per 256 sources walking the
// antennas, 37 separate 4-KiB streams each, the pass ran at 4.9 TB/s.)
__global__ void __launch_bounds__(256)
uscale_kernel(const f64x2* __restrict__ V, const double* __restrict__ f, int fstride, int nft,
              int nant, int nantp, int ns, int same_beam, f64x2* __restrict__ U) {
  const int a = blockIdx.x;                                  // row of U
  for (int ft = blockIdx.y; ft < nft; ft += gridDim.y) {     // grid.y is capped at 65535
    f64x2* up = U + ((size_t)ft * nantp + a) * ns;
    const int kind = a < nant ? 1 : (same_beam && a == nant ? 2 : 0);
    if (kind == 0) {
      for (int s = threadIdx.x; s < ns; s += 256) up[s] = f64x2{0.0, 0.0};
      continue;
    }
    const double fa = f[(size_t)ft * fstride + a];           // uniform: scalar load
    const f64x2* vp = V + ((size_t)ft * nant + (kind == 1 ? a : 0)) * ns;
    for (int s0 = threadIdx.x; s0 < ns; s0 += 8 * 256) {
      f64x2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = vp[min(s0 + q * 256, ns - 1)];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int s = s0 + q * 256;
        if (s < ns)
          up[s] = kind == 1 ? f64x2{v[q].x * fa, v[q].y * fa}
                            : f64x2{fa * (v[q].x * v[q].x + v[q].y * v[q].y), 0.0};
      }
    }
  }
}

// wft[ft] = w[0][ft]; *flag |= 1 if any baseline's weight differs from it
__global__ void __launch_bounds__(256)
noise_uniform_kernel(const double* __restrict__ w, int nbl, int nft, double* __restrict__ wft,
                     int32_t* __restrict__ flag) {
  // one entry of w per thread and trip: compare with baseline 0 of the same (f,t)
  const size_t total = (size_t)nbl * nft;
  bool bad = false;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    int ft = (int)(e % nft);
    double w0 = w[ft], we = w[e];
    if (e < (size_t)nft) wft[ft] = w0;
    bad |= (we != w0);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---- closed-form Gram with flagged / down-weighted samples -------------------------------
// Real data are flagged: a few (b,f,t) samples carry weight 0 (or less than their neighbours).
// With w_ref[ft] = max_b w[b,ft] the Gram splits exactly into
//   G = G_closed(all pairs, w_ref) - sum_{(b,ft): w < w_ref} (w_ref[ft] - w[b,ft]) Re(conj(R'_b,ft) R'_b,ft^T),
// the closed form plus a Gram over the MATERIALISED rows of the deviating samples only
// (gram_kernel; 1 % flags at config 2 = 85 000 of 8.5 M rows).  Replaces ptsrc_sampler.py:62-75
// for arbitrary inv_noise_var as long as the deviating samples are few; otherwise the general
// kernel runs.
__global__ void __launch_bounds__(256)
noise_ref_kernel(const double* __restrict__ w, int nbl, int nft, double* __restrict__ wref) {
  int ft = blockIdx.x * blockDim.x + threadIdx.x;
  if (ft >= nft) return;
  double m = w[ft];
  for (int b = 1; b < nbl; ++b) m = fmax(m, w[(size_t)b * nft + ft]);
  wref[ft] = m;
}

// row indices e = b * nft + ft of the samples with w[e] != wref[ft], appended in arbitrary order
// (the caller sorts them); *count receives their number even beyond `capacity`
__global__ void __launch_bounds__(256)
noise_dev_kernel(const double* __restrict__ w, const double* __restrict__ wref, int nbl, int nft,
                 long long capacity, long long* __restrict__ idx,
                 unsigned long long* __restrict__ count) {
  const size_t total = (size_t)nbl * nft;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    if (w[e] != wref[e % nft]) {
      unsigned long long pos = atomicAdd(count, 1ULL);
      if ((long long)pos < capacity) idx[pos] = (long long)e;
    }
  }
}

// R[r,:] = conj(V[ft,i_b,:]) V[ft,j_b,:] and rows[r] = g_i conj(g_j) sqrt(wref[ft] - w[b,ft]) for
// the listed samples r -> e = idx[r] = b * nft + ft
__global__ void __launch_bounds__(256)
flagged_rows_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns,
                    const f64x2* __restrict__ gains, const double* __restrict__ w,
                    const double* __restrict__ wref, const int32_t* __restrict__ bl_i,
                    const int32_t* __restrict__ bl_j, const long long* __restrict__ idx,
                    long long nrows, f64x2* __restrict__ R, f64x2* __restrict__ rows) {
  for (long long r = blockIdx.y; r < nrows; r += gridDim.y) {
    const long long e = idx[r];
    const int b = (int)(e / nft), ft = (int)(e % nft);
    const int i = bl_i[b], j = bl_j[b];
    const f64x2* vi = V + ((size_t)ft * nant + i) * ns;
    const f64x2* vj = V + ((size_t)ft * nant + j) * ns;
    f64x2* out = R + (size_t)r * ns;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x)
      out[s] = cmul_conj(vi[s], vj[s]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      f64x2 fac = {1.0, 0.0};
      if (gains) {
        f64x2 gi = gains[(size_t)i * nft + ft], gj = gains[(size_t)j * nft + ft];
        gj.y = -gj.y;
        fac = cmul(gi, gj);
      }
      double d = sqrt(fmax(wref[ft] - w[e], 0.0));
      rows[r] = f64x2{fac.x * d, fac.y * d};
    }
  }
}

// G <- epilogue(G - C) on full symmetric matrices (same operation order on (i,j) and (j,i))
__global__ void __launch_bounds__(256)
gram_subtract_kernel(double* __restrict__ G, const double* __restrict__ C, int ns,
                     const double* __restrict__ prior_std, const double* __restrict__ add_diag) {
  size_t n = (size_t)ns * ns;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    int i = (int)(e / ns), j = (int)(e % ns);
    double v = G[e] - C[e];
    if (prior_std) {
      v = prior_std[i > j ? i : j] * v * prior_std[i > j ? j : i];
      if (i == j) v = 1.0 + v;
    } else if (add_diag && i == j) {
      v += add_diag[i];
    }
    G[e] = v;
  }
}

// ---- 3-product form of the closed-form Gram ---------------------------------------------
// T = sum_a u_s conj(u_s') needs Re T = X^T X' + Y^T Y' and Im T = Y^T X' - X^T Y' (u = x + i y).
// With P1 = X^T X', P2 = Y^T Y', P3 = (X+Y)^T (X'-Y') = P1 - P2 + Im T:
//   Re T = P1 + P2,   Im T = P3 - P1 + P2,
// three real matrix products instead of four (25 % fewer MFMAs); the operands x+y and x'-y' cost
// one FP64 add each.  Three per-(f,t) accumulators + the persistent G per 16x16 block: output
// tile 96 (rows) x 128 (columns) per 8-wave workgroup, wave tile 48 x 32 (3 x 2 blocks,
// 24 accumulators = 192 VGPRs).  Tiles form a 96 x 128 grid over the lower triangle; the tile
// (ti, tj) is needed iff 96 ti + 95 >= 128 tj.
constexpr int K3_WC = HYDRA_K3_WC;                 // waves along the columns (2 along the rows)
constexpr int K3_THREADS = 2 * K3_WC * 64;
constexpr int K3_TR = 96, K3_TC = 32 * K3_WC, K3_W = K3_TR + K3_TC;   // LDS row (complex)
constexpr int K3_PIECES = UCH * K3_W / 64;         // 64-element DMA pieces per buffer
static_assert(UCH % 4 == 0 && (UCH * K3_W) % 64 == 0, "LDS buffer = whole k-steps and DMA pieces");
constexpr int K3_TILE = K3_TR * K3_TC;

__host__ __device__ inline int k3_ncols(int ti, int nct) {   // column tiles needed in row tile ti
  int c = (K3_TR * ti + K3_TR - 1) / K3_TC + 1;
  return c < nct ? c : nct;
}
static int k3_ntiles(int ns) {
  int nrt = (ns + K3_TR - 1) / K3_TR, nct = (ns + K3_TC - 1) / K3_TC, n = 0;
  for (int ti = 0; ti < nrt; ++ti) n += k3_ncols(ti, nct);
  return n;
}
// Compact tile index -> (ti, tj) in closed form: a workgroup of config 5 (408 000 tiles, 1042 row
// tiles) spent ~40 000 cycles walking the rows one by one, 10 % of the main kernel's workgroup and
// most of the slab-sum kernel's (159 -> 37 ms).
__host__ __device__ inline long long k3_floor_sum(long long n, long long m, long long a, long long b) {
  // sum_{i=0}^{n-1} floor((a i + b) / m), Euclid-like, O(log m)
  long long ans = 0;
  for (;;) {
    if (a >= m) { ans += (n - 1) * n / 2 * (a / m); a %= m; }
    if (b >= m) { ans += n * (b / m); b %= m; }
    long long ymax = a * n + b;
    if (ymax < m) break;
    n = ymax / m;
    b = ymax % m;
    long long t = m; m = a; a = t;
  }
  return ans;
}
// number of needed tiles in the row tiles 0 .. ti-1 (= compact index of tile (ti, 0))
__host__ __device__ inline long long k3_prefix(int ti, int nct) {
  // ncols(i) = min(nct, floor((TR i + TR - 1) / TC) + 1); the clamp holds from row tile i0 on
  long long i0 = ((long long)(nct - 1) * K3_TC - (K3_TR - 1) + K3_TR - 1) / K3_TR;
  if (i0 < 0) i0 = 0;
  long long lo = ti < i0 ? ti : i0;
  return k3_floor_sum(lo, K3_TC, K3_TR, K3_TR - 1) + lo + (ti - lo) * (long long)nct;
}
// Compact index p -> (ti, tj), bands of K3_BAND row tiles, column-major inside a band, so that the
// ~32 tiles an XCD works on at a time form a compact patch (6 row tiles x 5 column tiles share
// 1216 of the source columns; 32 tiles of one or two row tiles need nearly all of them).
constexpr int K3_BAND = 6;
__host__ __device__ inline void k3_tile_of(int p, int nct, int nrt, int* ti, int* tj) {
  // band: prefix(b0) <= p < prefix(b0 + K3_BAND); prefix(i) ~ TR i^2 / (2 TC) below the clamp
  double est = sqrt(2.0 * K3_TC * (double)p / K3_TR);
  int b0 = ((int)est / K3_BAND) * K3_BAND;
  if (b0 > nrt - 1) b0 = (nrt - 1) / K3_BAND * K3_BAND;
  while (b0 > 0 && k3_prefix(b0, nct) > p) b0 -= K3_BAND;
  while (b0 + K3_BAND < nrt && k3_prefix(b0 + K3_BAND, nct) <= p) b0 += K3_BAND;
  int q = (int)(p - k3_prefix(b0, nct));
  const int rows = (nrt - b0 < K3_BAND) ? nrt - b0 : K3_BAND;
  const int c0 = k3_ncols(b0, nct);          // columns present in every row tile of the band
  if (q < rows * c0) {                       // full columns, row tile fastest
    *tj = q / rows;
    *ti = b0 + q % rows;
    return;
  }
  q -= rows * c0;                            // ragged rest: row by row
  for (int i = b0; i < b0 + rows; ++i) {
    int extra = k3_ncols(i, nct) - c0;
    if (q < extra) { *ti = i; *tj = c0 + q; return; }
    q -= extra;
  }
  *ti = b0 + rows - 1; *tj = 0;              // not reached
}

template <bool SAMEBEAM>
__global__ void __launch_bounds__(K3_THREADS, K3_WC == 4 ? 1 : 2)
gram_uniform_kernel(const f64x2* __restrict__ U, int nft, int nantp, int ns, int ntiles,
                       int ft_per_split, int qrow, double* __restrict__ slabs) {
  extern __shared__ __align__(16) unsigned char smem[];
  f64x2* s_u = reinterpret_cast<f64x2*>(smem);  // [2][UCH][K3_W]

  int tile, split;
  block_work(blockIdx.x, ntiles, &tile, &split);
  int ti, tj;
  k3_tile_of(tile, (ns + K3_TC - 1) / K3_TC, (ns + K3_TR - 1) / K3_TR, &ti, &tj);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / K3_WC, wc = wave % K3_WC;
  const int kr = lane >> 4, lc = lane & 15;
  int ft_begin = split * ft_per_split;
  int ft_end = ft_begin + ft_per_split;
  if (ft_end > nft) ft_end = nft;
  const int nc = (nantp + UCH - 1) / UCH;
  const int nphase = (ft_end > ft_begin) ? (ft_end - ft_begin) * nc : 0;

  // LDS-DMA: one buffer = UCH rows x K3_W columns = K3_PIECES pieces of 64 consecutive elements; a
  // piece may straddle two rows (the per-lane SOURCE address makes that free); wave w issues pieces
  // w, w + nwaves, ...  The per-lane source offsets inside a buffer (row r, source column sc) do
  // not depend on the phase, so they are computed once: recomputing them per phase (a division by
  // the row length and the tile-column selects, ~35 VALU instructions per piece, issued by both
  // waves of a SIMD right after the barrier) kept the MFMA pipe idle for 16 % of the kernel.
  // Rows beyond nantp in the last buffer read the scratch rows behind U (never multiplied).
  constexpr int NPW = (K3_PIECES + 2 * K3_WC - 1) / (2 * K3_WC);   // pieces per wave
  unsigned off[NPW];   // byte offsets: 32 bits, added to a wave-uniform 64-bit base (saddr form)
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    int q = wave + 2 * K3_WC * i;
    int L = q * 64 + lane;
    int r = L / K3_W, col = L - r * K3_W;
    int sc = (col < K3_TR) ? ti * K3_TR + col : tj * K3_TC + (col - K3_TR);
    sc = min(sc, ns - 1);
    off[i] = (unsigned)(r * ns + sc) * 16u;
  }
  // Pieces i0 .. i1-1 of phase p (this wave's share).  The LDS-DMA instruction is written out by
  // hand: issued through the compiler's builtin, a FLAT-encoded operation that touches LDS puts
  // the compiler's wait-count bookkeeping into a mode where EVERY later wait in the wave is for
  // vmcnt(0) / lgkmcnt(0) as long as the DMA may be in flight -- here the whole phase -- so no
  // LDS read could stay in flight across a matrix instruction.  What the compiler does not see it
  // does not wait for; the one wait the DMA needs (all of it landed before the barrier) is
  // explicit below.
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s_u + wave * 64));
  auto issue = [&](int p, int i0, int i1) {
    int ft = ft_begin + p / nc, c = p % nc, buf = p & 1;
    const char* src = reinterpret_cast<const char*>(U + ((size_t)ft * nantp + c * UCH) * ns);   // uniform
    const unsigned dst = lds_wave + (unsigned)(buf * UCH * K3_W) * 16u;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (i >= i0 && i < i1 && wave + 2 * K3_WC * i < K3_PIECES) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst + (unsigned)(2 * K3_WC * 64 * i) * 16u), "v"(off[i]), "s"(src)
                     : "memory");
      }
    }
  };

  f64x4 G[3][2], P1[3][2], P2[3][2], P3[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) G[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  if (nphase > 0) issue(0, 0, NPW);
  const int offA = wr * 48 + lc;
  const int offB = K3_TR + wc * 32 + lc;
#ifdef HYDRA_GRAM_STAMPS
  unsigned long long u0, u1, u2, u3, u4, s_bar = 0, s_dma = 0, s_ks = 0, s_ep = 0;
  unsigned long long k_c0 = __builtin_amdgcn_s_memtime();
#endif
  for (int p = 0; p < nphase; ++p) {
    int c = p % nc, buf = p & 1;
#ifdef HYDRA_GRAM_STAMPS
    STAMP(u0)
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of phase p have landed
    __syncthreads();  // everyone's have; the other buffer is no longer being read
#ifdef HYDRA_GRAM_STAMPS
    STAMP(u1)
#endif
    // (the next phase's staging is issued piece by piece between the matrix instructions of
    // the first k-step: issued in one go after the barrier, the nine LDS-DMA instructions of a
    // wave -- and of its SIMD sibling at the same time -- kept the matrix pipe idle for ~800
    // cycles per phase, 5 % of the kernel)
    const bool stage_next = p + 1 < nphase;
#ifdef HYDRA_GRAM_STAMPS
    STAMP(u2)
#endif
    int ksteps = min(UCH, nantp - c * UCH) >> 2;
    const f64x2* ub = s_u + (size_t)(buf * UCH + kr) * K3_W;
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          P1[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
          P2[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
          P3[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
        }
    }
    // K-steps of 4 antennas.  The operands of k-step kk+1 are read INTO THE REGISTERS OF kk as
    // those fall free, in the order of their use: the first column block's B after the first
    // burst of nine matrix instructions, each row block's A after its last use in the second
    // burst, the second column block's B at the end -- every read is at least six matrix
    // instructions (384 cycles) ahead of its first use, no register is added, and the waits the
    // compiler places are counted (lgkmcnt(n)), not drains.  Before, a k-step began with LDS round
    // trips (three at first, then one) that left the matrix pipe idle: the SIMD's other wave,
    // which arbitration lets run only in those gaps, met the same gaps later with nobody to fill
    // them.  Only in the k-step that holds the q row (SAMEBEAM) the q lane is masked out of the
    // products and the rank-one term applied; every other k-step uses the operands as read (the
    // masks used to run in all of them: 20 VALU ops and 18 registers).
    auto ld = [&](int kk, int which) -> f64x2 {      // which: 0..2 = A row block, 3..4 = B column block
      const f64x2* up = ub + kk * 4 * K3_W;
      return which < 3 ? up[offA + which * 16] : up[offB + (which - 3) * 16];
    };
    auto m3 = [&](const f64x2& av, const f64x2& bv, int rb, int cb) {
      P1[rb][cb] = mfma(av.x, bv.x, P1[rb][cb]);
      P2[rb][cb] = mfma(av.y, bv.y, P2[rb][cb]);
      P3[rb][cb] = mfma(av.x + av.y, bv.x - bv.y, P3[rb][cb]);
      if (!SAMEBEAM)
        G[rb][cb] = mfma(-(av.x * av.x + av.y * av.y), bv.x * bv.x + bv.y * bv.y, G[rb][cb]);
    };
    f64x2 a0, a1, a2, b0, b1;
    if (ksteps > 0) {
      b0 = ld(0, 3); a0 = ld(0, 0); a1 = ld(0, 1); a2 = ld(0, 2); b1 = ld(0, 4);   // (steady-state order)
    }
#pragma unroll
    for (int kk = 0; kk < UCH / 4; ++kk) {
      if (kk < ksteps) {
        const int kn = min(kk + 1, ksteps - 1);      // (the last k-step re-reads itself: no branch)
        if (SAMEBEAM && (c * UCH + kk * 4 == (qrow & ~3))) {
          // a real branch (the empty asm keeps it from being turned into selects)
          asm volatile("" ::: "memory");
          const bool qlane = kr == (qrow & 3);
          f64x2* av[3] = {&a0, &a1, &a2};
#pragma unroll
          for (int rb = 0; rb < 3; ++rb) {
            const double nqv = qlane ? -av[rb]->x : 0.0;
            G[rb][0] = mfma(nqv, b0.x, G[rb][0]);
            G[rb][1] = mfma(nqv, b1.x, G[rb][1]);
            if (qlane) *av[rb] = f64x2{0.0, 0.0};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        m3(a0, b0, 0, 0);
        if (kk == 0 && stage_next) issue(p + 1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);   // (else the three a.x + a.y are hoisted to the top,
        m3(a1, b0, 1, 0);                    //  and with them the waits for all three reads)
        if (kk == 0 && stage_next) issue(p + 1, 2, 4);
        __builtin_amdgcn_sched_barrier(0);
        m3(a2, b0, 2, 0);
        if (kk == 0 && stage_next) issue(p + 1, 4, 6);
        __builtin_amdgcn_sched_barrier(0);
        b0 = ld(kn, 3);
        __builtin_amdgcn_sched_barrier(0);
        m3(a0, b1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        a0 = ld(kn, 0);
        if (kk == 0 && stage_next) issue(p + 1, 6, 8);
        __builtin_amdgcn_sched_barrier(0);
        m3(a1, b1, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        a1 = ld(kn, 1);
        if (kk == 0 && stage_next) issue(p + 1, 8, NPW);
        __builtin_amdgcn_sched_barrier(0);
        m3(a2, b1, 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        a2 = ld(kn, 2);
        b1 = ld(kn, 4);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef HYDRA_GRAM_STAMPS
    STAMP(u3)
#endif
    if (c == nc - 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            double tre = P1[i][j][g] + P2[i][j][g];
            double tim = P3[i][j][g] - (P1[i][j][g] - P2[i][j][g]);
            G[i][j][g] = fma(tim, tim, fma(tre, tre, G[i][j][g]));
          }
    }
#ifdef HYDRA_GRAM_STAMPS
    STAMP(u4)
    s_bar += u1 - u0; s_dma += u2 - u1; s_ks += u3 - u2; s_ep += u4 - u3;
#endif
  }
#ifdef HYDRA_GRAM_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_stamps[0] += s_bar; g_stamps[1] += s_dma; g_stamps[2] += s_ks; g_stamps[3] += s_ep;
    g_stamps[4] += nphase; g_stamps[5] += __builtin_amdgcn_s_memtime() - k_c0;
  }
#endif
  double* slab = slabs + ((size_t)split * ntiles + tile) * K3_TILE;
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        slab[(size_t)(wr * 48 + rb * 16 + kr + 4 * g) * K3_TC + wc * 32 + cb * 16 + lc] = G[rb][cb][g];
}

// Sum the slabs of the 96 x 128 tiles in split order, scale, mirror, apply the epilogue.
// grid (ntiles, K3_FIN_CHUNKS): a workgroup owns 1024 entries of one tile, 4 per thread, and
// keeps 16 slab loads in flight per thread (the sum over splits is the only long loop here).
constexpr int K3_FIN_CHUNKS = K3_TILE / 1024;
static_assert(K3_TILE % 1024 == 0, "finish kernel: 4 entries per thread, 256 threads");
__global__ void __launch_bounds__(256)
gram_finish_k3_kernel(const double* __restrict__ slabs, int nsplit, int ntiles, int ns,
                      const double* __restrict__ prior_std, const double* __restrict__ add_diag,
                      double* __restrict__ G, double scale) {
  int tile = blockIdx.x;
  int ti, tj;
  k3_tile_of(tile, (ns + K3_TC - 1) / K3_TC, (ns + K3_TR - 1) / K3_TR, &ti, &tj);
  const size_t stride = (size_t)ntiles * K3_TILE;
  const double* base = slabs + (size_t)tile * K3_TILE + blockIdx.y * 1024 + threadIdx.x;
  double sum[4] = {0.0, 0.0, 0.0, 0.0};
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    double v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[u][q] = base[(size_t)(s + u) * stride + q * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u)      // split order, as a serial sum would do
#pragma unroll
      for (int q = 0; q < 4; ++q) sum[q] += v[u][q];
  }
  for (; s < nsplit; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) sum[q] += base[(size_t)s * stride + q * 256];
  // the chunk is 1024 / K3_TC consecutive tile rows: written as it is, and mirrored across the
  // diagonal through LDS so that the mirror is stored in runs of (1024 / K3_TC) consecutive
  // doubles instead of single ones (the 80 GB matrix of config 5 spent 158 ms here)
  constexpr int CH_ROWS = 1024 / K3_TC;
  __shared__ double s_t[CH_ROWS][K3_TC + 1];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int e = blockIdx.y * 1024 + q * 256 + threadIdx.x;
    int r = e / K3_TC, c = e % K3_TC;
    int gi = ti * K3_TR + r, gj = tj * K3_TC + c;
    double val = sum[q] * scale;
    if (gi < ns && gj < ns && gj <= gi) {
      if (prior_std) {
        val = prior_std[gi] * val * prior_std[gj];
        if (gi == gj) val = 1.0 + val;
      } else if (add_diag && gi == gj) {
        val += add_diag[gi];
      }
      G[(size_t)gi * ns + gj] = val;
    }
    s_t[r - blockIdx.y * CH_ROWS][c] = val;
  }
  __syncthreads();
  const int rl = threadIdx.x % CH_ROWS;
  const int gi = ti * K3_TR + blockIdx.y * CH_ROWS + rl;
  for (int c = threadIdx.x / CH_ROWS; c < K3_TC; c += 256 / CH_ROWS) {
    int gj = tj * K3_TC + c;
    if (gi < ns && gj < gi) G[(size_t)gj * ns + gi] = s_t[rl][c];
  }
}

static void k3_plan(long long nft, int ns, int* ntiles, int* nsplit, long long* per) {
  *ntiles = k3_ntiles(ns);
  if (*ntiles >= 16384) {
    *nsplit = 1;
    *per = nft;
    return;
  }
  // (finer cuts measured no faster here: 6144 .. 16 384 workgroups, 115-117 ms against 114.5)
  long long want = std::max<long long>(8, 4096 / *ntiles);
  want = (want + 7) / 8 * 8;
  // at least ~8 (f,t) samples per workgroup: each one writes a whole slab of its tile at the end
  long long cap = std::max<long long>(8, nft / 64 * 8);
  want = std::min<long long>(std::min<long long>(want, cap), 512);
  *per = (nft + want - 1) / want;
  *nsplit = (int)want;
}

}  // namespace hydra

using namespace hydra;

#ifdef HYDRA_GRAM_STAMPS
extern "C" int hydra_debug_trace(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_trace), sizeof(g_trace)) == hipSuccess ? 0 : -3;
}
extern "C" int hydra_debug_stamps(unsigned long long* out_host, int reset) {
  if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -3;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#endif

extern "C" {

size_t hydra_gram_workspace_bytes(long long m, int ns) {
  if (m <= 0 || ns <= 0) return 0;
  int ntiles, nsplit;
  long long per;
  gram_plan(m, ns, &ntiles, &nsplit, &per, 4, 64);
  return (size_t)nsplit * ntiles * TILE_ELEMS * sizeof(double);
}

int hydra_gram(const double* R, const double* rows, long long m, int ns, const double* prior_std,
               const double* add_diag, double* G, void* workspace, size_t workspace_bytes,
               void* stream) {
  HYDRA_CHECK_ARG(R && G && m > 0 && ns > 0, "hydra_gram: bad arguments (m=%lld ns=%d)", m, ns);
  HYDRA_CHECK_ARG(!(prior_std && add_diag), "hydra_gram: give at most one of prior_std, add_diag");
  size_t need = hydra_gram_workspace_bytes(m, ns);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_gram: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int ntiles, nsplit;
  long long per;
  gram_plan(m, ns, &ntiles, &nsplit, &per, 4, 64);
  double* slabs = static_cast<double*>(workspace);
  profile_record(0, as_stream(stream));
  gram_kernel<<<ntiles * nsplit, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(R), reinterpret_cast<const f64x2*>(rows), m, ns, ntiles, per,
      slabs);
  profile_record(1, as_stream(stream));
  HYDRA_CHECK_LAUNCH("hydra_gram");
  gram_finish_kernel<<<ntiles, 256, 0, as_stream(stream)>>>(slabs, nsplit, ntiles, ns, prior_std,
                                                            add_diag, G);
  HYDRA_CHECK_LAUNCH("hydra_gram(finish)");
  return HYDRA_OK;
}

int hydra_noise_deviations(const double* inv_noise, int nbl, int nf, int nt, double* w_ref,
                           long long* idx, long long capacity, unsigned long long* count_dev,
                           void* stream) {
  HYDRA_CHECK_ARG(inv_noise && w_ref && count_dev && nbl > 0 && nf > 0 && nt > 0 && capacity >= 0 &&
                      (idx || capacity == 0),
                  "hydra_noise_deviations: bad arguments");
  hipStream_t st = as_stream(stream);
  int nft = nf * nt;
  hipError_t e = hipMemsetAsync(count_dev, 0, sizeof(unsigned long long), st);
  if (e != hipSuccess)
    return fail(HYDRA_ERR_HIP, "hydra_noise_deviations: %s", hipGetErrorString(e));
  noise_ref_kernel<<<(nft + 255) / 256, 256, 0, st>>>(inv_noise, nbl, nft, w_ref);
  size_t total = (size_t)nbl * nft;
  int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
  noise_dev_kernel<<<grid, 256, 0, st>>>(inv_noise, w_ref, nbl, nft, capacity, idx, count_dev);
  HYDRA_CHECK_LAUNCH("hydra_noise_deviations");
  return HYDRA_OK;
}

int hydra_flagged_rows(const double* V, int nf, int nt, int nant, int ns, const double* gains,
                       const double* inv_noise, const double* w_ref, const int32_t* bl_i,
                       const int32_t* bl_j, int nbl, const long long* idx, long long nrows,
                       double* R, double* rows, void* stream) {
  HYDRA_CHECK_ARG(V && inv_noise && w_ref && bl_i && bl_j && idx && R && rows && nf > 0 && nt > 0 &&
                      nant > 0 && ns > 0 && nbl > 0 && nrows > 0,
                  "hydra_flagged_rows: bad arguments");
  dim3 grid(std::min((ns + 255) / 256, 64), (unsigned)std::min<long long>(nrows, 65535));
  flagged_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(V), nf * nt, nant, ns, reinterpret_cast<const f64x2*>(gains),
      inv_noise, w_ref, bl_i, bl_j, idx, nrows, reinterpret_cast<f64x2*>(R),
      reinterpret_cast<f64x2*>(rows));
  HYDRA_CHECK_LAUNCH("hydra_flagged_rows");
  return HYDRA_OK;
}

int hydra_gram_subtract(double* G, const double* C, int ns, const double* prior_std,
                        const double* add_diag, void* stream) {
  HYDRA_CHECK_ARG(G && C && ns > 0 && !(prior_std && add_diag), "hydra_gram_subtract: bad arguments");
  size_t n = (size_t)ns * ns;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  gram_subtract_kernel<<<grid, 256, 0, as_stream(stream)>>>(G, C, ns, prior_std, add_diag);
  HYDRA_CHECK_LAUNCH("hydra_gram_subtract");
  return HYDRA_OK;
}

int hydra_profile_events(void* start_event, void* stop_event) {
  g_prof_ev[0] = static_cast<hipEvent_t>(start_event);
  g_prof_ev[1] = static_cast<hipEvent_t>(stop_event);
  return HYDRA_OK;
}

int hydra_prior_epilogue(double* G, int ns, const double* prior_std, const double* add_diag,
                         void* stream) {
  HYDRA_CHECK_ARG(G && ns > 0 && !(prior_std && add_diag), "hydra_prior_epilogue: bad arguments");
  size_t n = (size_t)ns * ns;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  prior_epilogue_kernel<<<grid, 256, 0, as_stream(stream)>>>(G, ns, prior_std, add_diag);
  HYDRA_CHECK_LAUNCH("hydra_prior_epilogue");
  return HYDRA_OK;
}

size_t hydra_gram_fused_workspace_bytes(int nf, int nt, int nbl, int ns) {
  if (nf <= 0 || nt <= 0 || nbl <= 0 || ns <= 0) return 0;
  int ntiles, nsplit;
  long long per;
  gram_plan((long long)nf * nt, ns, &ntiles, &nsplit, &per, 1, 1, FUSED_WORKGROUPS);
  size_t slabs = (size_t)nsplit * ntiles * TILE_ELEMS * sizeof(double);
  return align_up(slabs, 256) + (size_t)nf * nt * nbl * sizeof(double);
}

int hydra_gram_fused(const double* V, int nf, int nt, int nant, int ns, const double* gains,
                     const double* inv_noise, int nbl, const int32_t* seg_host,
                     const int32_t* rowtab_host, const int32_t* seg, const int32_t* rowtab,
                     int nseg, int nrows, const double* prior_std,
                     const double* add_diag, double* G, void* workspace, size_t workspace_bytes,
                     void* stream) {
  HYDRA_CHECK_ARG(V && G && seg_host && rowtab_host && seg && rowtab,
                  "hydra_gram_fused: null pointer argument");
  HYDRA_CHECK_ARG(nf > 0 && nt > 0 && nant > 0 && ns > 0 && nbl > 0 && nseg > 0 && nrows > 0,
                  "hydra_gram_fused: sizes must be positive");
  HYDRA_CHECK_ARG(!(prior_std && add_diag),
                  "hydra_gram_fused: give at most one of prior_std, add_diag");
  // validate the segment table on the host: it drives LDS indexing in the kernel
  int max_lds_rows = 0, max_seg_rows = 0;
  long long total = 0;
  for (int s = 0; s < nseg; ++s) {
    const int32_t* d = seg_host + 6 * s;
    HYDRA_CHECK_ARG(d[0] >= 0 && d[1] > 0 && d[0] + d[1] <= nant && d[2] >= 0 && d[3] > 0 &&
                        d[2] + d[3] <= nant && d[4] == total && d[5] > 0,
                    "hydra_gram_fused: malformed segment %d", s);
    int nload = (d[2] == d[0]) ? d[1] : d[1] + d[3];
    HYDRA_CHECK_ARG(d[2] != d[0] || d[3] == d[1], "hydra_gram_fused: segment %d aliases unequal blocks", s);
    for (int r = 0; r < d[5]; ++r) {
      const int32_t* rt = rowtab_host + 3 * (total + r);
      HYDRA_CHECK_ARG(rt[0] >= 0 && rt[0] < nload && rt[1] >= 0 && rt[1] < nload && rt[2] >= 0 &&
                          rt[2] < nbl,
                      "hydra_gram_fused: row %lld of segment %d out of range", total + r, s);
    }
    max_lds_rows = std::max(max_lds_rows, nload);
    max_seg_rows = std::max(max_seg_rows, (int)d[5]);
    total += d[5];
  }
  HYDRA_CHECK_ARG(total == nrows, "hydra_gram_fused: segments cover %lld rows, table has %d", total,
                  nrows);
  size_t lds = (size_t)max_lds_rows * 256 * 16 + (size_t)max_seg_rows * 12 + 16;
  HYDRA_CHECK_ARG(lds <= 160 * 1024, "hydra_gram_fused: segment needs %zu bytes of LDS (> 160 KiB)",
                  lds);
  size_t need = hydra_gram_fused_workspace_bytes(nf, nt, nbl, ns);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_gram_fused: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int nft = nf * nt;
  int ntiles, nsplit;
  long long per;
  gram_plan(nft, ns, &ntiles, &nsplit, &per, 1, 1, FUSED_WORKGROUPS);
  double* slabs = static_cast<double*>(workspace);
  size_t slab_bytes = align_up((size_t)nsplit * ntiles * TILE_ELEMS * sizeof(double), 256);
  double* sqrtw_t = reinterpret_cast<double*>(static_cast<unsigned char*>(workspace) + slab_bytes);
  hipStream_t st = as_stream(stream);
  {
    int gx = (int)std::min<long long>(((long long)max_seg_rows * nft + 255) / 256, 2048);
    row_weights_kernel<<<dim3(gx, nseg), 256, 0, st>>>(
        inv_noise, reinterpret_cast<const f64x2*>(gains), seg, rowtab, nbl, nft, sqrtw_t);
  }
  HYDRA_CHECK_LAUNCH("hydra_gram_fused(row weights)");
  // two wave groups per workgroup when LDS leaves room for a single workgroup per CU
  bool two_groups = lds > 80 * 1024;
  if (two_groups) lds = std::max<size_t>(lds, (size_t)TILE_ELEMS * sizeof(double));
  lds = std::max<size_t>(lds, 64 * 64 * sizeof(double));   // diagonal tiles: wave 1 -> wave 2
  const void* kfn = two_groups ? reinterpret_cast<const void*>(&gram_fused_kernel<2>)
                               : reinterpret_cast<const void*>(&gram_fused_kernel<1>);
  hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess)
    return fail(HYDRA_ERR_HIP, "hydra_gram_fused: cannot reserve %zu bytes of LDS: %s", lds,
                hipGetErrorString(e));
  profile_record(0, st);
  if (two_groups)
    gram_fused_kernel<2><<<ntiles * nsplit, 512, lds, st>>>(
        reinterpret_cast<const f64x2*>(V), nft, nant, ns, sqrtw_t, nbl, seg, nseg, rowtab,
        max_lds_rows, max_seg_rows, ntiles, (int)per, slabs);
  else
    gram_fused_kernel<1><<<ntiles * nsplit, 256, lds, st>>>(
        reinterpret_cast<const f64x2*>(V), nft, nant, ns, sqrtw_t, nbl, seg, nseg, rowtab,
        max_lds_rows, max_seg_rows, ntiles, (int)per, slabs);
  profile_record(1, st);
  HYDRA_CHECK_LAUNCH("hydra_gram_fused");
  gram_finish_kernel<<<ntiles, 256, 0, st>>>(slabs, nsplit, ntiles, ns, prior_std, add_diag, G);
  HYDRA_CHECK_LAUNCH("hydra_gram_fused(finish)");
  return HYDRA_OK;
}


size_t hydra_gram_uniform_workspace_bytes(int nf, int nt, int nant, int ns) {
  if (nf <= 0 || nt <= 0 || nant <= 0 || ns <= 0) return 0;
  int ntiles, nsplit;
  long long per;
  k3_plan((long long)nf * nt, ns, &ntiles, &nsplit, &per);
  size_t slabs = align_up((size_t)nsplit * ntiles * K3_TILE * sizeof(double), 256);
  int nantp = (nant + 1 + 3) / 4 * 4;   // room for the same-beam row
  size_t ftab = align_up((size_t)nf * nt * (nant + 1) * sizeof(double), 256);   // scale factors
  // + UCH scratch rows: the staging of the last (partial) antenna buffer reads a whole buffer
  return slabs + ftab + ((size_t)nf * nt * nantp + UCH) * ns * 16;
}

int hydra_noise_uniformity(const double* inv_noise, int nbl, int nf, int nt, double* w_ft,
                           int32_t* flag, void* stream) {
  HYDRA_CHECK_ARG(inv_noise && w_ft && flag && nbl > 0 && nf > 0 && nt > 0,
                  "hydra_noise_uniformity: bad arguments");
  int nft = nf * nt;
  size_t total = (size_t)nbl * nft;
  int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
  noise_uniform_kernel<<<grid, 256, 0, as_stream(stream)>>>(inv_noise, nbl, nft, w_ft, flag);
  HYDRA_CHECK_LAUNCH("hydra_noise_uniformity");
  return HYDRA_OK;
}

int hydra_gram_uniform(const double* V, int nf, int nt, int nant, int ns, const double* gains,
                       const double* w_ft, int same_beam, const double* prior_std,
                       const double* add_diag, double* G, void* workspace, size_t workspace_bytes,
                       void* stream) {
  HYDRA_CHECK_ARG(V && G, "hydra_gram_uniform: null pointer argument");
  HYDRA_CHECK_ARG(nf > 0 && nt > 0 && nant > 1 && ns > 0, "hydra_gram_uniform: bad sizes");
  HYDRA_CHECK_ARG(!(prior_std && add_diag),
                  "hydra_gram_uniform: give at most one of prior_std, add_diag");
  HYDRA_CHECK_ARG((long long)nf * nt <= 2147483647LL / 8, "hydra_gram_uniform: nf*nt too large");
  size_t need = hydra_gram_uniform_workspace_bytes(nf, nt, nant, ns);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_gram_uniform: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int nft = nf * nt;
  same_beam = same_beam ? 1 : 0;
  int nantp = (nant + same_beam + 3) / 4 * 4;
  int ntiles, nsplit;
  long long per;
  k3_plan(nft, ns, &ntiles, &nsplit, &per);
  double* slabs = static_cast<double*>(workspace);
  size_t slab_bytes = align_up((size_t)nsplit * ntiles * K3_TILE * sizeof(double), 256);
  double* ftab = reinterpret_cast<double*>(static_cast<unsigned char*>(workspace) + slab_bytes);
  size_t ftab_bytes = align_up((size_t)nft * (nant + 1) * sizeof(double), 256);
  f64x2* U = reinterpret_cast<f64x2*>(static_cast<unsigned char*>(workspace) + slab_bytes + ftab_bytes);
  hipStream_t st = as_stream(stream);
  ufactor_kernel<<<(nft + 255) / 256, 256, 0, st>>>(reinterpret_cast<const f64x2*>(gains), w_ft, nft,
                                                    nant, nant + 1, ftab);
  HYDRA_CHECK_LAUNCH("hydra_gram_uniform(factors)");
  dim3 ugrid(nantp, std::min(nft, 65535));
  uscale_kernel<<<ugrid, 256, 0, st>>>(reinterpret_cast<const f64x2*>(V), ftab, nant + 1, nft, nant,
                                       nantp, ns, same_beam, U);
  HYDRA_CHECK_LAUNCH("hydra_gram_uniform(scale)");
  size_t lds = (size_t)2 * UCH * K3_W * 16;
  profile_record(0, st);
  if (same_beam) {
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&gram_uniform_kernel<true>), lds)) return rc;
    gram_uniform_kernel<true><<<ntiles * nsplit, K3_THREADS, lds, st>>>(U, nft, nantp, ns, ntiles,
                                                                    (int)per, nant, slabs);
  } else {
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&gram_uniform_kernel<false>), lds)) return rc;
    gram_uniform_kernel<false><<<ntiles * nsplit, K3_THREADS, lds, st>>>(U, nft, nantp, ns, ntiles,
                                                                     (int)per, -1, slabs);
  }
  profile_record(1, st);
  HYDRA_CHECK_LAUNCH("hydra_gram_uniform");
  gram_finish_k3_kernel<<<dim3(ntiles, K3_FIN_CHUNKS), 256, 0, st>>>(slabs, nsplit, ntiles, ns, prior_std, add_diag, G, 0.5);
  HYDRA_CHECK_LAUNCH("hydra_gram_uniform(finish)");
  return HYDRA_OK;
}

}  // extern "C"
