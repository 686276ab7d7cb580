// Operator application kernels (HBM-bandwidth bound): R x, Re(R^H z) and the row-wise
// GCR right-hand-side terms, for a materialised R (M,Ns) c128 and for the factored form
// R[(b,f,t),s] = conj(V[f,t,i_b,s]) V[f,t,j_b,s].
//
// Algorithmic bytes: apply_R / adjoint_R read R once = 16*M*Ns B; the factored forms read V
// once = 16*Nf*Nt*Nant*Ns B.  All reductions run in a fixed order (no atomics) so results
// are bitwise reproducible.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace hydra {

// ---------------------------------------------------------------- materialised R

// out[r] = rowmul[r] * sum_s R[r,s] x[s]; one wave per row, 16-B loads contiguous over s.
__global__ void __launch_bounds__(256)
apply_R_kernel(const f64x2* __restrict__ R, long long m, int ns, const double* __restrict__ x,
               const f64x2* __restrict__ rowmul, f64x2* __restrict__ out) {
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < m; r += (long long)gridDim.x * 4) {
    const f64x2* rp = R + (size_t)r * ns;
    double sx = 0.0, sy = 0.0;
    for (int s = lane; s < ns; s += 64) {
      f64x2 v = rp[s];
      double xs = x[s];
      sx += v.x * xs;
      sy += v.y * xs;
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    if (lane == 0) {
      f64x2 o = {sx, sy};
      if (rowmul) o = cmul(rowmul[r], o);
      out[r] = o;
    }
  }
}

// partial[c][s] = sum over rows of chunk c of Re(conj(R[r,s]) z[r]); threads own columns.
constexpr int ADJ_COLS = 4;  // columns per thread -> 1024 columns per workgroup pass
__global__ void __launch_bounds__(256)
adjoint_R_kernel(const f64x2* __restrict__ R, long long m, int ns, const f64x2* __restrict__ z,
                 long long rows_per_chunk, double* __restrict__ partial) {
  long long r0 = (long long)blockIdx.x * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  if (r1 > m) r1 = m;
  int cbase = blockIdx.y * (256 * ADJ_COLS) + threadIdx.x;
  double acc[ADJ_COLS];
#pragma unroll
  for (int c = 0; c < ADJ_COLS; ++c) acc[c] = 0.0;
  for (long long r = r0; r < r1; ++r) {
    f64x2 zr = z[r];
    const f64x2* rp = R + (size_t)r * ns;
#pragma unroll
    for (int c = 0; c < ADJ_COLS; ++c) {
      int s = cbase + c * 256;
      if (s < ns) {
        f64x2 v = rp[s];
        acc[c] += v.x * zr.x + v.y * zr.y;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < ADJ_COLS; ++c) {
    int s = cbase + c * 256;
    if (s < ns) partial[(size_t)blockIdx.x * ns + s] = acc[c];
  }
}

// out[s] = scale[s] * sum_c partial[c][s] + add[s]
__global__ void __launch_bounds__(256)
adjoint_reduce_kernel(const double* __restrict__ partial, int nchunk, int ns,
                      const double* __restrict__ scale, const double* __restrict__ add,
                      double* __restrict__ out) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  double sum = 0.0;
  for (int c = 0; c < nchunk; ++c) sum += partial[(size_t)c * ns + s];
  if (scale) sum *= scale[s];
  if (add) sum += add[s];
  out[s] = sum;
}

// z[r] = conj(gg[r]) * ((data[r] - model[r]) * w[r] + omega[r] * sqrt(w[r]))
__global__ void __launch_bounds__(256)
rhs_rows_kernel(const f64x2* __restrict__ data, const f64x2* __restrict__ model,
                const double* __restrict__ w, const f64x2* __restrict__ omega,
                const f64x2* __restrict__ gg, long long m, f64x2* __restrict__ z) {
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < m;
       r += (long long)gridDim.x * blockDim.x) {
    f64x2 d = data[r];
    if (model) {
      f64x2 mo = model[r];
      d.x -= mo.x;
      d.y -= mo.y;
    }
    double wr = w ? w[r] : 1.0;
    f64x2 y = {d.x * wr, d.y * wr};
    if (omega) {
      double rt = sqrt(wr);
      f64x2 o = omega[r];
      y.x += o.x * rt;
      y.y += o.y * rt;
    }
    if (gg) y = cmul_conj(gg[r], y);
    z[r] = y;
  }
}

__global__ void __launch_bounds__(256)
scale_rows_kernel(const f64x2* y, const double* __restrict__ w, long long m, f64x2* z) {
  // z may alias y (in-place scaling)
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < m;
       r += (long long)gridDim.x * blockDim.x) {
    f64x2 v = y[r];
    double wr = w[r];
    z[r] = f64x2{v.x * wr, v.y * wr};
  }
}

// ---------------------------------------------------------------- factored form

// Sources staged per pass: 64 normally, 16 when the array is so large that all antennas x 64
// sources would not fit in the 160 KiB of LDS (HERA-350).  Rows are padded by one 16-B element
// so that different antennas land on different banks.

// out[b,ft] = rowmul[b,ft] * sum_s conj(V[ft,i_b,s]) V[ft,j_b,s] x[s]
// One workgroup per (f,t) and baseline chunk (blockIdx.z: chunks of 2048 baselines, dealt round
// robin -- only when there are too few (f,t) to fill the chip, since every chunk restages the
// voltages): voltages staged 64 sources at a time, thread <-> baseline.
// With seg != NULL (region sums): blockIdx.y = region r, the sum runs over the sources
// seg[r] <= s < seg[r+1] only and lands in out[(b,ft), r]; x may be NULL (= 1).
template <int VCH>
__global__ void __launch_bounds__(256)
apply_V_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns,
               const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
               const double* __restrict__ x, const f64x2* __restrict__ rowmul,
               f64x2* __restrict__ out, const int32_t* __restrict__ seg, int nreg) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int VPAD = VCH + 1;
  constexpr int NSUB = 64 / VCH;  // antennas staged per wave pass
  f64x2* s_v = reinterpret_cast<f64x2*>(smem);                       // [nant][VPAD]
  double* s_x = reinterpret_cast<double*>(s_v + (size_t)nant * VPAD);  // [VCH]
  int ft = blockIdx.x;
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int slane = lane % VCH, sub = lane / VCH;
  const f64x2* vft = V + (size_t)ft * nant * ns;
  const int s_lo = seg ? seg[blockIdx.y] : 0, s_hi = seg ? seg[blockIdx.y + 1] : ns;
  // up to 8 baselines per thread are accumulated in registers per pass over the sources
  constexpr int MAXB = 8;
  for (int bbase = blockIdx.z * 256 * MAXB; bbase < nbl; bbase += gridDim.z * 256 * MAXB) {
    f64x2 acc[MAXB];
    int bi[MAXB], bj[MAXB];
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
      acc[q] = f64x2{0.0, 0.0};
      int b = bbase + q * 256 + tid;
      bool ok = b < nbl;
      bi[q] = ok ? bl_i[b] : 0;
      bj[q] = ok ? bl_j[b] : 0;
    }
    for (int s0 = s_lo; s0 < s_hi; s0 += VCH) {
      __syncthreads();
      int s = s0 + slane;
      for (int a = wave * NSUB + sub; a < nant; a += 4 * NSUB)
        s_v[a * VPAD + slane] = (s < s_hi) ? vft[(size_t)a * ns + s] : f64x2{0.0, 0.0};
      if (tid < VCH) s_x[tid] = (s0 + tid < s_hi) ? (x ? x[s0 + tid] : 1.0) : 0.0;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < MAXB; ++q) {
        if (bbase + q * 256 < nbl) {  // uniform across the workgroup
          const f64x2* pi = s_v + bi[q] * VPAD;
          const f64x2* pj = s_v + bj[q] * VPAD;
          f64x2 a = acc[q];
          for (int k = 0; k < VCH; ++k) {
            f64x2 r = cmul_conj(pi[k], pj[k]);
            double xs = s_x[k];
            a.x += r.x * xs;
            a.y += r.y * xs;
          }
          acc[q] = a;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
      int b = bbase + q * 256 + tid;
      if (b < nbl) {
        size_t o = (size_t)b * nft + ft;
        f64x2 v = acc[q];
        if (rowmul) v = cmul(rowmul[o], v);
        out[seg ? o * nreg + blockIdx.y : o] = v;
      }
    }
  }
}

// ---- R x in antenna space on the matrix cores -----------------------------------------------------
// (R x)[b = (i,j)] = sum_s conj(V_i,s) V_j,s x_s = C[i][j],  C = conj(V) diag(x) V^T per (f,t): a
// Hermitian (Nant x Nant) product with K = sources.  Both MFMA operands have the antenna on the
// 16-lane axis and 4 consecutive sources on the k axis, so they are loaded straight from global
// memory (64-byte pieces of 16 antenna rows per instruction; the next k-step uses the other half
// of each 128-byte line) -- no LDS, no barrier.  One WAVE owns one (f,t) (or one of `nsplit` source
// ranges of it when there are few (f,t)) and keeps all upper-triangular 16 x 16 blocks of C in
// registers; C goes to a scratch array from which a gather kernel forms out[b] (any pair list,
// repeated and reversed pairs included).  Nant <= 16 NR <= 64.
template <int NR>
__global__ void __launch_bounds__(256, NR <= 3 ? 2 : 1)
apply_V_mfma_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns,
                    const double* __restrict__ x, int nsplit, int k_per_split,
                    f64x2* __restrict__ Cmat) {
  constexpr int NP = NR * (NR + 1) / 2;
  constexpr int NA = 16 * NR;
  const int lane = threadIdx.x & 63;
  const int kr = lane >> 4, lc = lane & 15;
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long long)nft * nsplit) return;
  const int ft = (int)(w / nsplit), sp = (int)(w % nsplit);
  // the waves of a workgroup share an (f,t) and take INTERLEAVED k-steps (wave sp: k-steps sp,
  // sp + nsplit, ...), so that together they walk each antenna row front to back -- contiguous
  // ranges per wave meant 4 x 48 scattered DRAM streams per workgroup and 2.1 TB/s
  const int k0 = 0, k1 = k_per_split;
  const f64x2* vft = V + (size_t)ft * nant * ns;
  const f64x2* rowp[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) rowp[r] = vft + (size_t)min(r * 16 + lc, nant - 1) * ns;
  // three real products per complex one: with a = conj(v_i) x = (ax, -ay), b = v_j = (bx, by)
  //   P1 = ax bx,  P2 = ay by,  P3 = (ax - ay)(bx + by):   Re C = P1 + P2,  Im C = P3 - P1 + P2
  // (NR = 4 keeps the four-product form, p1 = Re C, p2 = Im C: a third accumulator set would not
  // fit its 256 architectural registers)
  constexpr bool THREE = NR <= 3;
  f64x4 p1[NP], p2[NP], p3[THREE ? NP : 1];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    p1[p] = f64x4{0.0, 0.0, 0.0, 0.0};
    p2[p] = f64x4{0.0, 0.0, 0.0, 0.0};
    if (THREE) p3[p] = f64x4{0.0, 0.0, 0.0, 0.0};
  }
  constexpr int DEPTH = 4;   // k-steps of operands in flight
  // Loads are unconditional (clamped, in-bounds addresses) and masked only through the weight xs
  // of the A operand: a load under a condition is compiled into a branch with a wait right behind
  // it, which serialised the whole stream (7.4 ms instead of 4.7).  Antenna rows beyond Nant repeat
  // row Nant-1; their rows/columns of C are never gathered.
  auto load = [&](int kk, f64x2 (&v)[NR], double& xs) {
    int s = 4 * (kk * nsplit + sp) + kr;
    int sc = min(s, ns - 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = rowp[r][sc];
    double xv = x ? x[sc] : 1.0;
    xs = (kk < k1 && s < ns) ? xv : 0.0;
  };
  f64x2 vb[DEPTH][NR];
  double xb[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(k0 + d, vb[d], xb[d]);
  for (int kk = k0; kk < k1; kk += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      f64x2 v[NR];
      double ax[NR], ay[NR], ad[NR], bs[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        v[r] = vb[d][r];
        ax[r] = v[r].x * xb[d];
        ay[r] = v[r].y * xb[d];
        if (THREE) {
          ad[r] = ax[r] - ay[r];
          bs[r] = v[r].x + v[r].y;
        }
      }
      load(kk + d + DEPTH, vb[d], xb[d]);   // refill this slot (masked beyond k1)
      // three sweeps over the block pairs, one product each: consecutive MFMAs never write the
      // accumulator the previous one wrote
#pragma unroll
      for (int which = 0; which < (THREE ? 3 : 4); ++which) {
        int p = 0;
#pragma unroll
        for (int rb = 0; rb < NR; ++rb)
#pragma unroll
          for (int cb = rb; cb < NR; ++cb, ++p) {
            if constexpr (THREE) {
              if (which == 0) p1[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[rb], v[cb].x, p1[p], 0, 0, 0);
              if (which == 1) p2[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay[rb], v[cb].y, p2[p], 0, 0, 0);
              if (which == 2) p3[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[rb], bs[cb], p3[p], 0, 0, 0);
            } else {
              if (which == 0) p1[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[rb], v[cb].x, p1[p], 0, 0, 0);
              if (which == 1) p2[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[rb], v[cb].y, p2[p], 0, 0, 0);
              if (which == 2) p1[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay[rb], v[cb].y, p1[p], 0, 0, 0);
              if (which == 3) p2[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ay[rb], v[cb].x, p2[p], 0, 0, 0);
            }
          }
      }
    }
  }
  f64x2* cm = Cmat + ((size_t)sp * nft + ft) * NA * NA;
  int p = 0;
#pragma unroll
  for (int rb = 0; rb < NR; ++rb)
#pragma unroll
    for (int cb = rb; cb < NR; ++cb, ++p)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        cm[(size_t)(rb * 16 + kr + 4 * g) * NA + cb * 16 + lc] =
            THREE ? f64x2{p1[p][g] + p2[p][g], p3[THREE ? p : 0][g] - p1[p][g] + p2[p][g]}
                  : f64x2{p1[p][g], p2[p][g]};
}

// ---- the same product for 65 .. 144 antennas -----------------------------------------------------
// The upper-triangular 16 x 16 blocks of C no longer fit one wave's registers (36 blocks x 3
// products at 127 antennas), so a WORKGROUP of 8 waves owns an (f,t) (or one source range of it) and
// splits the blocks, row-major, in runs of MAXB; the voltages of 32 sources x all antennas are
// staged in LDS (two buffers, LDS-DMA written by hand as in gram_uniform_kernel so that the
// compiler's waits stay counted), and every wave reads the A (row block) and B (column block)
// operand of each of its blocks from there: element (antenna a, source s) sits at slot
// s ^ (a & 15) of its row (the per-lane DMA source address does the swizzle), which spreads the 16
// lanes of a read over all banks.  Three real products per complex one, as above.  Runs shorter
// than MAXB are padded with block (0,0), whose second result is dropped.  Two 32-source tiles of
// up to 144 antennas fit the 160 KiB of LDS; larger arrays keep the pair loop.
constexpr int AVB_SRC = 32;
constexpr int AVB_THREADS = 512;

template <int MAXB>
__global__ void __launch_bounds__(AVB_THREADS)
apply_V_big_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns, int nr,
                   const double* __restrict__ x, int nsplit, int tiles_per_split,
                   f64x2* __restrict__ Cmat) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nap = 16 * nr, rows = nap;
  const int tile_bytes = rows * AVB_SRC * 16 + AVB_SRC * 16;    // voltages, then x (padded to 16 B)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int ft = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int ntile = (ns + AVB_SRC - 1) / AVB_SRC;
  const int t0 = sp * tiles_per_split, t1 = min(t0 + tiles_per_split, ntile);

  // this wave's blocks: entries q0 .. q0 + MAXB - 1 of the row-major list of upper blocks
  const int nblk = nr * (nr + 1) / 2;
  int brow[MAXB], bcol[MAXB];
  {
    int q0 = wave * MAXB, rb = 0, base = 0;
    while (rb < nr && base + (nr - rb) <= q0) { base += nr - rb; ++rb; }
    int cb = rb + (q0 - base);
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
      const bool ok = q0 + q < nblk;
      brow[q] = ok ? rb : -1;              // padding: multiplies block (0,0) again, never stored
      bcol[q] = ok ? cb : 0;
      if (++cb >= nr) { ++rb; cb = rb; }
    }
  }

  // staging: a piece = 2 antenna rows x 32 sources = 1 KiB; wave w issues pieces w, w + 8, ...;
  // rows beyond Nant repeat row Nant-1 (never gathered), sources beyond Ns repeat source Ns-1
  // (their x is 0).  x: one 256-byte piece by wave 0.
  const f64x2* vft = V + (size_t)ft * nant * ns;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int npiece = nap / 2;
  auto stage = [&](int t, int buf) {
    const int s0 = t * AVB_SRC;
    for (int q = wave; q < npiece; q += 8) {
      const int a = min(2 * q + (lane >> 5), nant - 1);
      const int sl = (lane & 31) ^ ((2 * q + (lane >> 5)) & 15);
      const unsigned off = (unsigned)min(s0 + sl, ns - 1) * 16u;
      const char* src = reinterpret_cast<const char*>(vft);
      const unsigned long long rowoff = (unsigned long long)a * ns * 16ull + off;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * tile_bytes + q * 1024));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                   :: "s"(dst), "v"(src + rowoff) : "memory");
    }
    if (wave == 0 && x) {                 // 32 doubles of x as 64 dwords, one per lane
      const long long dw = min((long long)s0 * 2 + lane, 2LL * ns - 1);
      const char* src = reinterpret_cast<const char*>(x) + dw * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(
          lds_base + (unsigned)(buf * tile_bytes + rows * AVB_SRC * 16));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                   :: "s"(dst), "v"(src) : "memory");
    }
  };

  f64x4 p1[MAXB], p2[MAXB], p3[MAXB];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    p1[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    p2[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    p3[q] = f64x4{0.0, 0.0, 0.0, 0.0};
  }
  if (t0 < t1) stage(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // tile t has landed; the other buffer is free
    if (t + 1 < t1) stage(t + 1, buf ^ 1);
    const unsigned char* tb = smem + buf * tile_bytes;
    const double* xs = reinterpret_cast<const double*>(tb + rows * AVB_SRC * 16);
    const int s0 = t * AVB_SRC;
#pragma unroll 2
    for (int k = 0; k < AVB_SRC / 4; ++k) {
      const int sl = 4 * k + g;                                 // source within the tile
      double xv = x ? xs[sl] : 1.0;
      if (s0 + sl >= ns) xv = 0.0;
#pragma unroll
      for (int q = 0; q < MAXB; ++q) {
        const int ra = max(brow[q], 0) * 16 + lr, rc = bcol[q] * 16 + lr;
        const f64x2 av = *reinterpret_cast<const f64x2*>(tb + (ra * AVB_SRC + (sl ^ lr)) * 16);
        const f64x2 bv = *reinterpret_cast<const f64x2*>(tb + (rc * AVB_SRC + (sl ^ lr)) * 16);
        const double ax = av.x * xv, ay = av.y * xv;
        p1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bv.x, p1[q], 0, 0, 0);
        p2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, bv.y, p2[q], 0, 0, 0);
        p3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax - ay, bv.x + bv.y, p3[q], 0, 0, 0);
      }
    }
  }
  f64x2* cm = Cmat + ((size_t)sp * nft + ft) * nap * nap;
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    if (brow[q] >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        cm[(size_t)(brow[q] * 16 + g + 4 * e) * nap + bcol[q] * 16 + lr] =
            f64x2{p1[q][e] + p2[q][e], p3[q][e] - p1[q][e] + p2[q][e]};
    }
  }
}

// ---- the same product for MORE than 144 antennas (HERA-350: 22 block rows, 253 upper blocks) ----
// Neither the blocks of C (3 accumulators each) fit the registers of one workgroup nor 32 sources of
// all antennas two LDS buffers, so the grid gets a third dimension: a workgroup owns one SUPER-BLOCK
// of C -- 4 x 4 blocks = 64 row antennas x 64 column antennas, on or above the super-block diagonal
// -- of one (f,t) and source range, and stages the voltages of those 128 antennas only (rows 0-63 of
// the LDS tile: the row antennas, 64-127: the column antennas; two buffers of 64 KiB + x).  Each of
// the 8 waves takes two of the (up to) 16 blocks.  Staging, swizzle, k-step and the 3-product complex
// multiply are apply_V_big_kernel's.  The super-block index is the fastest grid axis, so the 21
// workgroups that read the same voltages run at the same time and all but the first find them in
// the memory-side cache.
constexpr int AVS_SB = 4;                      // blocks per super-block edge
constexpr int AVS_ROWS = 2 * 16 * AVS_SB;      // rows of an LDS tile

__global__ void __launch_bounds__(AVB_THREADS)
apply_V_sb_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns, int nr,
                  const double* __restrict__ x, int nsplit, int tiles_per_split, int nsr,
                  f64x2* __restrict__ Cmat) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int tile_bytes = AVS_ROWS * AVB_SRC * 16 + AVB_SRC * 16;   // voltages, then x
  const int nap = 16 * nr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int nsb = nsr * (nsr + 1) / 2;
  const int sb = blockIdx.x % nsb, rest = blockIdx.x / nsb;
  const int sp = rest % nsplit, ft = rest / nsplit;
  int sI = 0, sJ;
  {
    int base = 0;
    while (sI + 1 < nsr && base + (nsr - sI) <= sb) { base += nsr - sI; ++sI; }
    sJ = sI + (sb - base);
  }
  const int ntile = (ns + AVB_SRC - 1) / AVB_SRC;
  const int t0 = sp * tiles_per_split, t1 = min(t0 + tiles_per_split, ntile);

  // this wave's two blocks: entries 2 wave, 2 wave + 1 of the super-block's list of blocks that
  // exist (inside the array, on or above the block diagonal)
  int brl[2] = {-1, -1}, bcl[2] = {0, 0};
  {
    int cnt = 0;
    for (int c = 0; c < AVS_SB * AVS_SB; ++c) {
      const int rbl = c / AVS_SB, cbl = c % AVS_SB;
      const int rb = AVS_SB * sI + rbl, cb = AVS_SB * sJ + cbl;
      if (rb < nr && cb < nr && cb >= rb) {
        if (cnt == 2 * wave) { brl[0] = rbl; bcl[0] = cbl; }
        if (cnt == 2 * wave + 1) { brl[1] = rbl; bcl[1] = cbl; }
        ++cnt;
      }
    }
  }

  const f64x2* vft = V + (size_t)ft * nant * ns;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  auto stage = [&](int t, int buf) {
    const int s0 = t * AVB_SRC;
    for (int q = wave; q < AVS_ROWS / 2; q += 8) {
      const int row = 2 * q + (lane >> 5);
      const int a = min((row < 16 * AVS_SB ? 16 * AVS_SB * sI + row : 16 * AVS_SB * (sJ - 1) + row), nant - 1);
      const int sl = (lane & 31) ^ (row & 15);
      const unsigned off = (unsigned)min(s0 + sl, ns - 1) * 16u;
      const char* src = reinterpret_cast<const char*>(vft);
      const unsigned long long rowoff = (unsigned long long)a * ns * 16ull + off;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * tile_bytes + q * 1024));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                   :: "s"(dst), "v"(src + rowoff) : "memory");
    }
    if (wave == 0 && x) {
      const long long dw = min((long long)s0 * 2 + lane, 2LL * ns - 1);
      const char* src = reinterpret_cast<const char*>(x) + dw * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(
          lds_base + (unsigned)(buf * tile_bytes + AVS_ROWS * AVB_SRC * 16));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off"
                   :: "s"(dst), "v"(src) : "memory");
    }
  };

  f64x4 p1[2], p2[2], p3[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    p1[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    p2[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    p3[q] = f64x4{0.0, 0.0, 0.0, 0.0};
  }
  int ra[2], rc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    ra[q] = max(brl[q], 0) * 16 + lr;
    rc[q] = 16 * AVS_SB + bcl[q] * 16 + lr;
  }
  if (t0 < t1) stage(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // tile t has landed; the other buffer is free
    if (t + 1 < t1) stage(t + 1, buf ^ 1);
    const unsigned char* tb = smem + buf * tile_bytes;
    const double* xs = reinterpret_cast<const double*>(tb + AVS_ROWS * AVB_SRC * 16);
    const int s0 = t * AVB_SRC;
#pragma unroll 4
    for (int k = 0; k < AVB_SRC / 4; ++k) {
      const int sl = 4 * k + g;
      double xv = x ? xs[sl] : 1.0;
      if (s0 + sl >= ns) xv = 0.0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f64x2 av = *reinterpret_cast<const f64x2*>(tb + (ra[q] * AVB_SRC + (sl ^ lr)) * 16);
        const f64x2 bv = *reinterpret_cast<const f64x2*>(tb + (rc[q] * AVB_SRC + (sl ^ lr)) * 16);
        const double ax = av.x * xv, ay = av.y * xv;
        p1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bv.x, p1[q], 0, 0, 0);
        p2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, bv.y, p2[q], 0, 0, 0);
        p3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax - ay, bv.x + bv.y, p3[q], 0, 0, 0);
      }
    }
  }
  f64x2* cm = Cmat + ((size_t)sp * nft + ft) * nap * nap;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (brl[q] >= 0) {
      const int rb = AVS_SB * sI + brl[q], cb = AVS_SB * sJ + bcl[q];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        cm[(size_t)(rb * 16 + g + 4 * e) * nap + cb * 16 + lr] =
            f64x2{p1[q][e] + p2[q][e], p3[q][e] - p1[q][e] + p2[q][e]};
    }
  }
}

// out[b,ft] = rowmul[b,ft] * sum_split C[split][ft][i_b][j_b]; entries below the block diagonal
// are the conjugates of the stored upper blocks
__global__ void __launch_bounds__(256)
apply_V_gather_kernel(const f64x2* __restrict__ Cmat, int nft, int na, int nsplit,
                      const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                      const f64x2* __restrict__ rowmul, f64x2* __restrict__ out) {
  size_t n = (size_t)nbl * nft;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < n;
       o += (size_t)gridDim.x * blockDim.x) {
    int b = (int)(o / nft), ft = (int)(o % nft);
    int i = bl_i[b], j = bl_j[b];
    bool flip = (i >> 4) > (j >> 4);
    int r = flip ? j : i, c = flip ? i : j;
    f64x2 acc = {0.0, 0.0};
    for (int sp = 0; sp < nsplit; ++sp) {
      f64x2 v = Cmat[(((size_t)sp * nft + ft) * na + r) * na + c];
      acc.x += v.x;
      acc.y += v.y;
    }
    if (flip) acc.y = -acc.y;
    if (rowmul) acc = cmul(rowmul[o], acc);
    out[o] = acc;
  }
}

// ---- R^H z in antenna space on the matrix cores -------------------------------------------------
// sum_b Re(conj(conj(V_i) V_j) z_b) = Re sum_i V_i conj(W_i),  W = Zc V,  Zc[i][j] = sum of conj(z_b)
// over the baselines b = (i, j): per (f,t) a (Nant x Nant) x (Nant x Ns) complex product instead of
// Nbl pair products per source -- 4 real MFMAs per 16 x 4 block of Zc (blocks without a baseline
// are skipped: the upper-triangular pair list of an array leaves ~60 %), then a dot product with
// the voltages the lane already holds as MFMA B operands (the D-layout row 16 rb + kr + 4 g of a lane
// is exactly the antenna of its B operand of k-step 4 rb + g).  Zc is scattered into LDS from the
// transposed z, double buffered, one barrier per (f,t).  Nant <= 16 NR <= 64.
#ifndef HYDRA_ADJ_OCC1_FROM
#define HYDRA_ADJ_OCC1_FROM 4   // NR from which one workgroup per CU (512 registers a lane) is asked for
#endif
#ifndef HYDRA_ADJ_THREE_MAX
#define HYDRA_ADJ_THREE_MAX 4   // largest NR that uses the 3-product complex multiply
#endif
#ifndef HYDRA_ADJ_NOPF_FROM
#define HYDRA_ADJ_NOPF_FROM 99  // NR from which the next sample's voltages are NOT prefetched into registers
#endif
template <int NR>
__global__ void __launch_bounds__(256, NR >= HYDRA_ADJ_OCC1_FROM ? 1 : 2)
adjoint_V_mfma_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns,
                      const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                      const f64x2* __restrict__ zt, int ft_per_split, double* __restrict__ partial) {
  constexpr int NK = 4 * NR;            // k-steps of 4 antennas
  constexpr int S = 4 * NK + 1;         // LDS row stride of Zc (complex elements), odd
  constexpr int MAXB = (16 * NR * (16 * NR + 1) / 2 + 255) / 256;   // baselines per thread
  constexpr bool PF = NR < HYDRA_ADJ_NOPF_FROM;
  extern __shared__ __align__(16) unsigned char smem[];
  f64x2* s_z = reinterpret_cast<f64x2*>(smem);                       // [2][16 NR][S]
  int* s_mask = reinterpret_cast<int*>(s_z + 2 * 16 * NR * S);       // [NR] bit kk set: block used
  unsigned* s_off = reinterpret_cast<unsigned*>(s_mask + 4);         // [4][NK] byte offsets of V rows
  int* s_cnt = reinterpret_cast<int*>(s_z);   // [16 NR][16 NR] pair counts, prologue only (aliases Zc)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kr = lane >> 4, lc = lane & 15;
  const int s = (blockIdx.x * 4 + wave) * 16 + lc;
  const int ft0 = blockIdx.y * ft_per_split;
  const int ft1 = min(ft0 + ft_per_split, nft);

  // ---- prologue: zero both Zc buffers, note which blocks hold a baseline, look for duplicates
  for (int e = tid; e < 16 * NR * 16 * NR; e += 256) s_cnt[e] = 0;
  if (tid < NR) s_mask[tid] = 0;
  // row offsets of the voltages a lane loads at k-step kk (rows beyond Nant clamped), kept in LDS
  // and read back per sample: as loop invariants they would sit in 2 NK registers a lane
  if (tid < 4 * NK)
    s_off[tid] = (unsigned)min(4 * (tid % NK) + tid / NK, nant - 1) * (unsigned)ns * 16u;
  __syncthreads();
  int pos[MAXB];
  bool dup = false;
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    int b = tid + 256 * q;
    pos[q] = -1;
    if (b < nbl) {
      int i = bl_i[b], j = bl_j[b];
      pos[q] = i * S + j;
      atomicOr(&s_mask[i >> 4], 1 << (j >> 2));
      if (atomicAdd(&s_cnt[i * 16 * NR + j], 1) > 0) dup = true;
    }
  }
  const bool any_dup = __syncthreads_or(dup);
  int mask[NR];
#pragma unroll
  for (int rb = 0; rb < NR; ++rb) mask[rb] = s_mask[rb];
  for (int e = tid; e < 2 * 16 * NR * S; e += 256) s_z[e] = f64x2{0.0, 0.0};
  __syncthreads();

  auto load_z = [&](int ft, f64x2 (&zr)[MAXB]) {
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
      unsigned b = (unsigned)min(tid + 256 * q, nbl - 1);   // unconditional, clamped: only stored where pos >= 0
      zr[q] = *reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(zt + (size_t)ft * nbl) + b * 16u);
    }
  };
  auto store_z = [&](int buf, const f64x2 (&zr)[MAXB]) {
    f64x2* zb = s_z + buf * 16 * NR * S;
    if (!any_dup) {
#pragma unroll
      for (int q = 0; q < MAXB; ++q)
        if (pos[q] >= 0) zb[pos[q]] = f64x2{zr[q].x, -zr[q].y};
    } else {   // repeated pairs: clear the used entries, then add (two extra barriers)
#pragma unroll
      for (int q = 0; q < MAXB; ++q)
        if (pos[q] >= 0) zb[pos[q]] = f64x2{0.0, 0.0};
      __syncthreads();
#pragma unroll
      for (int q = 0; q < MAXB; ++q)
        if (pos[q] >= 0) {
          double* zd = reinterpret_cast<double*>(zb + pos[q]);
          atomicAdd(zd, zr[q].x);
          atomicAdd(zd + 1, -zr[q].y);
        }
    }
  };
  auto load_v = [&](int ft, f64x2 (&v)[NK]) {
    // unconditional loads from clamped addresses (a conditional load compiles to a branch with a
    // wait behind it): rows beyond Nant meet zero columns of Zc and give zero rows of W, lanes
    // beyond Ns are never written out
    const char* vft = reinterpret_cast<const char*>(V + (size_t)ft * nant * ns);   // uniform
    const unsigned s16 = (unsigned)min(s, ns - 1) * 16u;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
      v[kk] = *reinterpret_cast<const f64x2*>(vft + (s_off[kr * NK + kk] + s16));
  };

  // The block pattern is fixed before the (f,t) loop, and the loop exists twice, so that its body
  // has no branch around the matrix instructions (a uniform branch per block makes the compiler
  // copy every accumulator at every join): pair lists that leave all blocks below the block
  // diagonal empty -- any i <= j list, the usual case -- skip those blocks; other lists take all.
  bool upper = true;
#pragma unroll
  for (int rb = 0; rb < NR; ++rb) upper = upper && (mask[rb] & ((1 << (4 * rb)) - 1)) == 0;

  double acc = 0.0;
  auto run = [&](auto upper_tag) {
    constexpr bool UPPER = decltype(upper_tag)::value;
    f64x2 zr[MAXB], v[NK], vn[PF ? NK : 1];
    load_z(ft0, zr);
    load_v(ft0, v);
    store_z(0, zr);
    // one (f,t) sample: Zc[buf] x voltages vc; the next sample's z and voltages (into vnx) travel
    // meanwhile.  "Next" is clamped to the last sample instead of tested, and the two register
    // sets swap roles from call to call: a conditional copy vn -> v would cost a third set.
    auto step = [&](int ft, f64x2 (&vc)[NK], f64x2 (&vnx)[PF ? NK : 1]) {
      const int buf = (ft - ft0) & 1;
      const int nxt = min(ft + 1, ft1 - 1);
      __syncthreads();   // Zc[buf] is complete; nobody reads Zc[buf ^ 1] any more
      load_z(nxt, zr);
      if constexpr (PF) load_v(nxt, reinterpret_cast<f64x2 (&)[NK]>(vnx));
      const f64x2* zb = s_z + buf * 16 * NR * S;
      // W = Zc V with three real products per complex one:
      //   P1 = Zr Vr,  P2 = Zi Vi,  P3 = (Zr + Zi)(Vr + Vi):  Re W = P1 - P2,  Im W = P3 - P1 - P2
      constexpr bool THREE = NR <= HYDRA_ADJ_THREE_MAX;
      f64x4 w1[NR], w2[NR], w3[THREE ? NR : 1];
#pragma unroll
      for (int rb = 0; rb < NR; ++rb) {
        w1[rb] = f64x4{0.0, 0.0, 0.0, 0.0};
        w2[rb] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (THREE) w3[rb] = f64x4{0.0, 0.0, 0.0, 0.0};
      }
      // k-steps, the Zc operands of step kk + 1 read from LDS before the products of step kk are
      // issued; the scheduling barriers keep the compiler from hoisting ALL the reads to the top
      // (which costs more registers than there are)
      f64x2 zc[NR], zn[NR];
#pragma unroll
      for (int rb = 0; rb < NR; ++rb) zc[rb] = zb[(rb * 16 + lc) * S + kr];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        if (kk + 1 < NK) {
#pragma unroll
          for (int rb = 0; rb < NR; ++rb)
            if (!(UPPER && kk + 1 < 4 * rb)) zn[rb] = zb[(rb * 16 + lc) * S + 4 * (kk + 1) + kr];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double vs = vc[kk].x + vc[kk].y;
#pragma unroll
        for (int rb = 0; rb < NR; ++rb) {
          if (UPPER && kk < 4 * rb) continue;   // compile time
          if constexpr (THREE) {
            w1[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[rb].x, vc[kk].x, w1[rb], 0, 0, 0);
            w2[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[rb].y, vc[kk].y, w2[rb], 0, 0, 0);
            w3[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[rb].x + zc[rb].y, vs, w3[rb], 0, 0, 0);
          } else {
            w1[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[rb].x, vc[kk].x, w1[rb], 0, 0, 0);
            w1[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-zc[rb].y, vc[kk].y, w1[rb], 0, 0, 0);
            w2[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[rb].y, vc[kk].x, w2[rb], 0, 0, 0);
            w2[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[rb].x, vc[kk].y, w2[rb], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < NK) {
#pragma unroll
          for (int rb = 0; rb < NR; ++rb)
            if (!(UPPER && kk + 1 < 4 * rb)) zc[rb] = zn[rb];
        }
      }
#pragma unroll
      for (int rb = 0; rb < NR; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double wre = THREE ? w1[rb][g] - w2[rb][g] : w1[rb][g];
          const double wim = THREE ? w3[THREE ? rb : 0][g] - w1[rb][g] - w2[rb][g] : w2[rb][g];
          acc += wre * vc[4 * rb + g].x + wim * vc[4 * rb + g].y;
        }
      store_z(buf ^ 1, zr);   // (after the last sample: a copy of its own z, never read)
      if constexpr (!PF) load_v(nxt, vc);
    };
    for (int ft = ft0; ft < ft1; ft += 2) {
      step(ft, v, vn);
      if (ft + 1 >= ft1) break;
      if constexpr (PF)
        step(ft + 1, reinterpret_cast<f64x2 (&)[NK]>(vn), reinterpret_cast<f64x2 (&)[PF ? NK : 1]>(v));
      else
        step(ft + 1, v, vn);
    }
  };
  if (ft0 < ft1) {
    if (upper) run(std::true_type{});
    else run(std::false_type{});
  }
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (kr == 0 && s < ns) partial[(size_t)blockIdx.y * ns + s] = acc;
}

// (nbl, nft) -> (nft, nbl) complex transpose so a workgroup reads one (f,t) slice contiguously
__global__ void __launch_bounds__(256)
ctranspose_kernel(const f64x2* __restrict__ in, int nbl, int nft, f64x2* __restrict__ out) {
  size_t n = (size_t)nbl * nft;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < n;
       o += (size_t)gridDim.x * blockDim.x) {
    size_t ft = o / nbl, b = o % nbl;
    out[o] = in[b * (size_t)nft + ft];
  }
}

// partial[split][s] = sum_{ft in split} sum_b Re( V_i,s conj(V_j,s) z[b,ft] )
// Workgroup <-> (CW-source chunk, ft range); the 256 threads are CW sources x (256/CW) baseline
// groups; fixed-order reduction over the groups through LDS.  CW = 64 normally, 16 for arrays
// too large to stage all antennas x 64 sources.
template <int CW>
__global__ void __launch_bounds__(256)
adjoint_V_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns,
                 const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                 const f64x2* __restrict__ zt, int ft_per_split, double* __restrict__ partial) {
  constexpr int NG = 256 / CW;
  extern __shared__ __align__(16) unsigned char smem[];
  f64x2* s_v = reinterpret_cast<f64x2*>(smem);  // [nant][CW]
  double* s_red = reinterpret_cast<double*>(s_v + (size_t)nant * CW);  // [NG][CW]
  int tid = threadIdx.x, sl = tid % CW, grp = tid / CW;
  int s = blockIdx.x * CW + sl;
  int ft0 = blockIdx.y * ft_per_split;
  int ft1 = ft0 + ft_per_split;
  if (ft1 > nft) ft1 = nft;
  double acc = 0.0, acc2 = 0.0;
  for (int ft = ft0; ft < ft1; ++ft) {
    const f64x2* vft = V + (size_t)ft * nant * ns;
    const f64x2* zft = zt + (size_t)ft * nbl;
    __syncthreads();
    for (int a = grp; a < nant; a += NG)
      s_v[a * CW + sl] = (s < ns) ? vft[(size_t)a * ns + s] : f64x2{0.0, 0.0};
    __syncthreads();
    // 4 baselines per trip: the (wave-uniform) index and z loads of all four are issued before
    // the LDS reads, so their latency overlaps instead of serialising per baseline
    int b = grp;
    for (; b + 3 * NG < nbl; b += 4 * NG) {
      int i0 = bl_i[b], i1 = bl_i[b + NG], i2 = bl_i[b + 2 * NG], i3 = bl_i[b + 3 * NG];
      int j0 = bl_j[b], j1 = bl_j[b + NG], j2 = bl_j[b + 2 * NG], j3 = bl_j[b + 3 * NG];
      f64x2 z0 = zft[b], z1 = zft[b + NG], z2 = zft[b + 2 * NG], z3 = zft[b + 3 * NG];
      f64x2 r0 = cmul_conj(s_v[i0 * CW + sl], s_v[j0 * CW + sl]);  // R[b,s]
      f64x2 r1 = cmul_conj(s_v[i1 * CW + sl], s_v[j1 * CW + sl]);
      f64x2 r2 = cmul_conj(s_v[i2 * CW + sl], s_v[j2 * CW + sl]);
      f64x2 r3 = cmul_conj(s_v[i3 * CW + sl], s_v[j3 * CW + sl]);
      acc += (r0.x * z0.x + r0.y * z0.y) + (r1.x * z1.x + r1.y * z1.y);  // Re(conj(R) z)
      acc2 += (r2.x * z2.x + r2.y * z2.y) + (r3.x * z3.x + r3.y * z3.y);
    }
    for (; b < nbl; b += NG) {
      f64x2 r = cmul_conj(s_v[bl_i[b] * CW + sl], s_v[bl_j[b] * CW + sl]);
      f64x2 zb = zft[b];
      acc += r.x * zb.x + r.y * zb.y;
    }
  }
  acc += acc2;
  __syncthreads();
  s_red[grp * CW + sl] = acc;
  __syncthreads();
  if (grp == 0 && s < ns) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) t += s_red[g * CW + sl];
    partial[(size_t)blockIdx.y * ns + s] = t;
  }
}

static void adjoint_plan(long long m, int ns, int* nchunk, long long* rows_per_chunk) {
  long long want = std::min<long long>(1024, std::max<long long>(1, m / 32));
  long long per = (m + want - 1) / want;
  *nchunk = (int)((m + per - 1) / per);
  *rows_per_chunk = per;
  (void)ns;
}

// test hook: 0 forces the pair-loop kernels (hydra_debug_adjoint_mfma)
static int g_adjoint_mfma = 1;

// ---- R^H z in antenna space for arrays of MORE than 64 antennas ------------------------------------
// The same identity as adjoint_V_mfma_kernel -- W = Zc V per (f,t), Zc[i][j] = conj(z_ij) for the
// listed pairs i <= j, then out[s] += Re sum_i V_i,s conj(W_i,s) -- but neither Zc (127 antennas:
// 258 KB) nor a lane's voltages (Nant / 4 registers pairs) fit on chip whole:
//   * the voltages of a 32-source tile, all antennas, are staged in LDS by LDS-DMA, double buffered
//     over (f,t) (rows of 512 B: the four antenna rows x 16 sources of a B-operand read fall on
//     disjoint banks);
//   * Zc is laid out beforehand in MFMA-fragment order (adjbig_frag_kernel): the A operand of
//     row block rb and k-step kk is 64 consecutive entries, one per lane, read straight from
//     global memory / L2 (every source tile of an (f,t) reads the same fragments);
//   * only k-steps on and right of the block diagonal exist (Zc is upper triangular); the four
//     waves of a workgroup share the source tile and take row blocks in pairs (k, nr-1-k), which
//     have the same number of k-steps together.
// Three real products per complex one: P1 = Zr Vr, P2 = Zi Vi, P3 = (Zr + Zi)(Vr + Vi).
// Work: 6 Nant_p^2/2 Ns flops per (f,t) executed (8 M Ns in the baseline-form count); V is read
// from HBM once.  The pair loop on the vector units it replaces is LDS-bound at ~5 TFLOP/s.
constexpr int AB_SRC = 32;            // sources per workgroup tile (two 16-wide column blocks)

__host__ __device__ inline int adjbig_prefix(int rb, int nk) { return rb * nk - 2 * rb * (rb - 1); }

// tab[i * nap + j] = baseline index of the pair (i, j), -1 if absent (nap = 16 nr)
__global__ void __launch_bounds__(256)
adjbig_table_kernel(const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                    int nap, int32_t* __restrict__ tab) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nbl; b += gridDim.x * blockDim.x)
    tab[bl_i[b] * nap + bl_j[b]] = b;
}

// Fragments of Zc for every (f,t): zri[(ft * nsteps + step) * 64 + lane] = conj(z[b]) with
// i = 16 rb + (lane & 15), j = 4 kk + (lane >> 4), step = prefix(rb) + kk - 4 rb; zs = re + im.
__global__ void __launch_bounds__(256)
adjbig_frag_kernel(const f64x2* __restrict__ zt, const int32_t* __restrict__ tab, int nbl, int nft,
                   int nr, int nsteps, f64x2* __restrict__ zri, double* __restrict__ zs) {
  const int nk = 4 * nr, nap = 16 * nr;
  const int lane = threadIdx.x & 63;
  const int step = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (step >= nsteps) return;
  int rb = 0;
  while (rb + 1 < nr && adjbig_prefix(rb + 1, nk) <= step) ++rb;
  const int kk = 4 * rb + step - adjbig_prefix(rb, nk);
  const int i = 16 * rb + (lane & 15), j = 4 * kk + (lane >> 4);
  const int b = (j >= i) ? tab[i * nap + j] : -1;
  for (int ft = blockIdx.y; ft < nft; ft += gridDim.y) {
    f64x2 v = {0.0, 0.0};
    if (b >= 0) {
      f64x2 zv = zt[(size_t)ft * nbl + b];
      v = f64x2{zv.x, -zv.y};
    }
    size_t o = ((size_t)ft * nsteps + step) * 64 + lane;
    zri[o] = v;
    zs[o] = v.x + v.y;
  }
}

// Five waves: waves 0-3 multiply, wave 4 only stages the next voltage tile.  The split is what
// lets the multiplying waves pipeline their operand loads: an LDS-DMA operation counts on vmcnt and,
// being a FLAT-encoded instruction that touches LDS, makes the compiler drain vmcnt AND lgkmcnt to
// zero at the next dependent use in the wave that issued it -- one full memory latency per group of
// MFMAs.  In a wave of its own it delays nobody.
constexpr int AB_THREADS = 320;
#ifdef HYDRA_GRAM_STAMPS
// diagnostic build only: per-wave timeline of workgroup 0's third (f,t): [wave][arrival at the
// barrier, past it, first row block done, all row blocks done]
__device__ unsigned long long g_adj_trace[8][8];
#define ADJ_STAMP(var)                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                     \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
  __builtin_amdgcn_sched_barrier(0);
#endif

__global__ void __launch_bounds__(AB_THREADS)
adjoint_V_big_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns, int nr,
                     const f64x2* __restrict__ zri, const double* __restrict__ zs, int ft_per_split,
                     double* __restrict__ partial) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nk = 4 * nr, nap = 16 * nr, nsteps = adjbig_prefix(nr, nk);
  f64x2* s_v = reinterpret_cast<f64x2*>(smem);                 // [2][nap][AB_SRC]
  double* s_red = reinterpret_cast<double*>(smem);             // [16][AB_SRC], after the loop
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the loops branch, not mask
  const int kr = lane >> 4, lc = lane & 15;
  const int s0 = blockIdx.x * AB_SRC;
  const int ft0 = blockIdx.y * ft_per_split, ft1 = min(ft0 + ft_per_split, nft);

  // LDS-DMA of one (f,t) tile: a piece = 2 antenna rows x 32 sources = 1 KiB.  Rows beyond Nant
  // repeat row Nant-1 (they only meet zero entries of Zc), sources beyond Ns repeat source Ns-1
  // (their sums are not stored).
  // The stager's loop is scalar instructions plus the DMA itself: while a SIMD executes FP64
  // matrix instructions back to back, another wave's VECTOR instructions get in about once per
  // matrix instruction (measured: 14 instructions per piece, 9 of them vector address arithmetic,
  // took 320 cycles per piece -- the 64 pieces of a tile the whole compute phase, and the barrier
  // waited for them; raising the wave's priority changed nothing).  So the per-lane part of the
  // address is one constant offset, the rest is a scalar base.  The DMA is issued by inline asm
  // (see gram_uniform_kernel in gram.hip).
  const unsigned lane_off = (unsigned)min(s0 + (lane & 31), ns - 1) * 16u;
  const unsigned voff2 = lane_off + (unsigned)(lane >> 5) * (unsigned)ns * 16u;   // two rows
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  auto stage = [&](int ft, int buf) {
    const char* vft = reinterpret_cast<const char*>(V + (size_t)ft * nant * ns);
    const size_t row_bytes = (size_t)ns * 16;
    unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((size_t)buf * nap * AB_SRC * 16));
    const int nfull = nant / 2;                      // pieces whose two rows both exist
    const char* src = vft;
    for (int q = 0; q < nfull; ++q) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(dst), "v"(voff2), "s"(src) : "memory");
      src += 2 * row_bytes;
      dst += 2 * AB_SRC * 16;
    }
    // rows beyond Nant repeat row Nant-1 (both halves of the piece read the same row)
    const char* last = vft + (size_t)(nant - 1) * row_bytes;
    for (int q = nfull; q < nap / 2; ++q) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(dst), "v"(lane_off), "s"(last) : "memory");
      dst += 2 * AB_SRC * 16;
    }
  };

  // operands of one group = four k-steps = one 16-antenna column block of Zc
  struct Group {
    f64x2 z[4];
    double s[4];
    f64x2 v0[4], v1[4];
  };

  double acc[2] = {0.0, 0.0};
  if (wave == 4 && ft0 < ft1) stage(ft0, 0);
  for (int ft = ft0; ft < ft1; ++ft) {
    const int buf = (ft - ft0) & 1;
#ifdef HYDRA_GRAM_STAMPS
    unsigned long long a0, a1, a2 = 0, a3;
    ADJ_STAMP(a0)
#endif
    if (wave == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                      // tile ft has landed; the other buffer is free
#ifdef HYDRA_GRAM_STAMPS
    ADJ_STAMP(a1)
#endif
    if (wave == 4) {
      if (ft + 1 < ft1) stage(ft + 1, buf ^ 1);
#ifdef HYDRA_GRAM_STAMPS
      ADJ_STAMP(a3)
      if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && ft == ft0 + 2) {
        g_adj_trace[4][0] = a0; g_adj_trace[4][1] = a1; g_adj_trace[4][2] = a3; g_adj_trace[4][3] = a3;
      }
#endif
      continue;
    }
    const f64x2* vb = s_v + (size_t)buf * nap * AB_SRC;
    const f64x2* zr_ft = zri + (size_t)ft * nsteps * 64 + lane;
    const double* zs_ft = zs + (size_t)ft * nsteps * 64 + lane;
    auto row_block = [&](int rb) {
      f64x4 p1[2], p2[2], p3[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        p1[c] = f64x4{0.0, 0.0, 0.0, 0.0};
        p2[c] = f64x4{0.0, 0.0, 0.0, 0.0};
        p3[c] = f64x4{0.0, 0.0, 0.0, 0.0};
      }
      const int ng = nr - rb;                      // groups of this row block
      const f64x2* zp = zr_ft + (size_t)adjbig_prefix(rb, nk) * 64;
      const double* zq = zs_ft + (size_t)adjbig_prefix(rb, nk) * 64;
      const f64x2* vp = vb + (size_t)(16 * rb + kr) * AB_SRC + lc;
      auto load_step = [&](Group& o, int g, int u) {
        o.z[u] = zp[(size_t)(4 * g + u) * 64];
        o.s[u] = zq[(size_t)(4 * g + u) * 64];
        o.v0[u] = vp[(size_t)(4 * g + u) * 4 * AB_SRC];
        o.v1[u] = vp[(size_t)(4 * g + u) * 4 * AB_SRC + 16];
      };
      auto load = [&](Group& o, int g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) load_step(o, g, u);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mma_step = [&](const Group& o, int u) {
        const double s0v = o.v0[u].x + o.v0[u].y, s1v = o.v1[u].x + o.v1[u].y;
        p1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.z[u].x, o.v0[u].x, p1[0], 0, 0, 0);
        p1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.z[u].x, o.v1[u].x, p1[1], 0, 0, 0);
        p2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.z[u].y, o.v0[u].y, p2[0], 0, 0, 0);
        p2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.z[u].y, o.v1[u].y, p2[1], 0, 0, 0);
        p3[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.s[u], s0v, p3[0], 0, 0, 0);
        p3[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.s[u], s1v, p3[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      // One group multiplies while the operands of the group after next are read INTO ITS
      // REGISTERS, k-step by k-step as they fall free: every read is 1.75 groups (42 matrix
      // instructions, 2700 cycles) ahead of its use with two operand sets and no register moves
      // (the Zc fragments come from L2, ~1500 cycles away).  No branch inside the loop body (a
      // conditional use lets the compiler sink the reads next to it): reads past the row block's
      // end repeat its last group and are not used.  What this does not cover is the start of a
      // row block (two per (f,t) and wave: ~2000 cycles each until the first operands are there,
      // and an epilogue of 8 LDS reads) -- carrying the two sets on into the next row block was
      // tried: the operand sets then live through the epilogue, the compiler serialises its reads
      // for want of registers (256: the stager shares a SIMD) and answers the joins with
      // vmcnt(0): 2.6 ms against 2.1; prefetching only the next row block's first Zc fragments
      // into 24 spare registers (across (f,t) too): 2.18 ms, no gain either.
      auto mma_reload = [&](Group& o, int gnext) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          mma_step(o, u);
          load_step(o, gnext, u);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto mma = [&](const Group& o) {
#pragma unroll
        for (int u = 0; u < 4; ++u) mma_step(o, u);
      };
      Group ga, gb;
      load(ga, 0);
      load(gb, min(1, ng - 1));
      for (int g = 0; g + 1 < ng; g += 2) {
        mma_reload(ga, min(g + 2, ng - 1));
        mma_reload(gb, min(g + 3, ng - 1));
      }
      if (ng & 1) mma(ga);
      // Re(V_i conj(W_i)) for the rows this lane holds: antenna 16 rb + kr + 4 g
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f64x2 vv = vb[(size_t)(16 * rb + kr + 4 * g) * AB_SRC + 16 * c + lc];
          const double wre = p1[c][g] - p2[c][g];
          const double wim = p3[c][g] - p1[c][g] - p2[c][g];
          acc[c] += vv.x * wre + vv.y * wim;
        }
    };
    for (int k = wave; 2 * k < nr; k += 4) {
      row_block(k);
#ifdef HYDRA_GRAM_STAMPS
      if (k == wave) { ADJ_STAMP(a2) }
#endif
      if (nr - 1 - k != k) row_block(nr - 1 - k);
    }
#ifdef HYDRA_GRAM_STAMPS
    ADJ_STAMP(a3)
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && ft == ft0 + 2) {
      g_adj_trace[wave][0] = a0; g_adj_trace[wave][1] = a1; g_adj_trace[wave][2] = a2; g_adj_trace[wave][3] = a3;
    }
#endif
  }
  __syncthreads();                         // everyone is done with the voltage tiles
  if (wave < 4) {
    s_red[(wave * 4 + kr) * AB_SRC + lc] = acc[0];
    s_red[(wave * 4 + kr) * AB_SRC + 16 + lc] = acc[1];
  }
  __syncthreads();
  if (tid < AB_SRC && s0 + tid < ns) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s_red[q * AB_SRC + tid];
    partial[(size_t)blockIdx.y * ns + s0 + tid] = t;
  }
}

// ---- R^H z in antenna space for arrays of more than 160 antennas (HERA-350) ---------------------
// Two 32-source tiles of ALL antennas no longer fit the LDS (350 antennas: 2 x 176 KiB), so the grid
// gets a third dimension here too: a workgroup owns a SUPER-ROW of Zc -- 4 row blocks = 64 antennas,
// one per multiplying wave -- for its 32 sources and (f,t) range, and walks the super-columns
// K = I .. nsr-1 on and right of the diagonal; the voltages of the 64 antennas of super-column K
// (32 KiB) are double-buffered in LDS by the staging wave, one barrier per tile.  W_i = sum_k Zc[i][k]
// V_k accumulates in the wave's registers over the tiles of an (f,t); then Re(V_i conj(W_i)) with V_i
// re-read from global memory (8 loads per lane and (f,t): the tile that held them is gone).  The Zc
// fragments are adjbig_frag_kernel's; the fragments of the NEXT group of four k-steps are fetched
// (global memory / L2) while the current group multiplies, across tile and (f,t) boundaries.
// Super-column K's voltages are read once per super-row I <= K: (nsr + 1) / 2 times on average,
// mostly from the memory-side cache (workgroups of equal source tile and (f,t) range differ in
// blockIdx.z only).
constexpr int AS_SB = 4;                       // row blocks per super-row = multiplying waves

__global__ void __launch_bounds__(AB_THREADS)
adjoint_V_sb_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns, int nr,
                    const f64x2* __restrict__ zri, const double* __restrict__ zs, int ft_per_split,
                    int nsr, double* __restrict__ partial) {
  __shared__ __align__(16) f64x2 s_v[2][16 * AS_SB * AB_SRC];          // 2 x 32 KiB
  double* s_red = reinterpret_cast<double*>(&s_v[0][0]);               // [16][AB_SRC], after the loop
  const int nk = 4 * nr, nsteps = adjbig_prefix(nr, nk);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kr = lane >> 4, lc = lane & 15;
  const int s0 = blockIdx.x * AB_SRC;
  const int ft0 = blockIdx.y * ft_per_split, ft1 = min(ft0 + ft_per_split, nft);
  const int sI = nsr - 1 - (int)blockIdx.z;      // the long super-rows are handed out first
  const int rb = AS_SB * sI + wave;              // this wave's row block (waves 0-3)
  const bool active = wave < AS_SB && rb < nr;

  const unsigned lane_off = (unsigned)min(s0 + (lane & 31), ns - 1) * 16u;
  const unsigned voff2 = lane_off + (unsigned)(lane >> 5) * (unsigned)ns * 16u;   // two rows
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) void*)&s_v[0][0]);
  const size_t row_bytes = (size_t)ns * 16;
  auto stage = [&](int ft, int K, int buf) {
    const char* vft = reinterpret_cast<const char*>(V + (size_t)ft * nant * ns);
    unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * 16 * AS_SB * AB_SRC * 16));
    const int a0 = 16 * AS_SB * K;
    for (int q = 0; q < 8 * AS_SB; ++q) {          // 32 pieces of 2 rows x 32 sources
      const int a = a0 + 2 * q;
      if (a + 1 < nant) {
        const char* src = vft + (size_t)a * row_bytes;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst), "v"(voff2), "s"(src) : "memory");
      } else {          // rows beyond Nant repeat an existing row: they only meet zeros of Zc
        const char* src = vft + (size_t)min(a, nant - 1) * row_bytes;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst), "v"(lane_off), "s"(src) : "memory");
      }
      dst += 2 * AB_SRC * 16;
    }
  };

  struct ZS {
    f64x2 z[4];
    double s[4];
  };
  // fragments of the group (= column block kb) of row block rb at (f,t) ft
  auto load_zs = [&](ZS& o, int ft, int kb) {
    const size_t step0 = (size_t)ft * nsteps + adjbig_prefix(rb, nk) + 4 * (kb - rb);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      o.z[u] = zri[(step0 + u) * 64 + lane];
      o.s[u] = zs[(step0 + u) * 64 + lane];
    }
  };

  double acc[2] = {0.0, 0.0};
  ZS cur, nxt;
  if (active && ft0 < ft1) load_zs(cur, ft0, rb);
  if (wave == AS_SB && ft0 < ft1) stage(ft0, sI, 0);
  int tile = 0;
  for (int ft = ft0; ft < ft1; ++ft) {
    f64x4 p1[2], p2[2], p3[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      p1[c] = f64x4{0.0, 0.0, 0.0, 0.0};
      p2[c] = f64x4{0.0, 0.0, 0.0, 0.0};
      p3[c] = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    for (int K = sI; K < nsr; ++K, ++tile) {
      const int buf = tile & 1;
      if (wave == AS_SB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                    // this tile has landed; the other buffer is free
      if (wave == AS_SB) {
        if (K + 1 < nsr) stage(ft, K + 1, buf ^ 1);
        else if (ft + 1 < ft1) stage(ft + 1, sI, buf ^ 1);
        continue;
      }
      if (!active) continue;
      const f64x2* vb = &s_v[buf][0];
      const int kb0 = max(AS_SB * K, rb), kb1 = min(AS_SB * K + AS_SB, nr);
      for (int kb = kb0; kb < kb1; ++kb) {
        // the group after this one: next column block, next tile, or next (f,t)
        int nft_ = ft, nkb = kb + 1;
        if (nkb >= nr) { nkb = rb; nft_ = (ft + 1 < ft1) ? ft + 1 : ft; }
        load_zs(nxt, nft_, nkb);
        const f64x2* vp = vb + (size_t)(16 * (kb - AS_SB * K) + kr) * AB_SRC + lc;
        f64x2 v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v0[u] = vp[(size_t)u * 4 * AB_SRC];
          v1[u] = vp[(size_t)u * 4 * AB_SRC + 16];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double s0v = v0[u].x + v0[u].y, s1v = v1[u].x + v1[u].y;
          p1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.z[u].x, v0[u].x, p1[0], 0, 0, 0);
          p1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.z[u].x, v1[u].x, p1[1], 0, 0, 0);
          p2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.z[u].y, v0[u].y, p2[0], 0, 0, 0);
          p2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.z[u].y, v1[u].y, p2[1], 0, 0, 0);
          p3[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.s[u], s0v, p3[0], 0, 0, 0);
          p3[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.s[u], s1v, p3[1], 0, 0, 0);
        }
        cur = nxt;
      }
    }
    if (active) {
      // Re(V_i conj(W_i)) for the rows this lane holds: antenna 16 rb + kr + 4 g
      const f64x2* vft = V + (size_t)ft * nant * ns;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int a = 16 * rb + kr + 4 * g;
          const f64x2 vv = vft[(size_t)min(a, nant - 1) * ns + min(s0 + 16 * c + lc, ns - 1)];
          const double wre = p1[c][g] - p2[c][g];
          const double wim = p3[c][g] - p1[c][g] - p2[c][g];
          if (a < nant) acc[c] += vv.x * wre + vv.y * wim;
        }
    }
  }
  __syncthreads();                         // everyone is done with the voltage tiles
  if (wave < AS_SB) {
    s_red[(wave * 4 + kr) * AB_SRC + lc] = acc[0];
    s_red[(wave * 4 + kr) * AB_SRC + 16 + lc] = acc[1];
  }
  __syncthreads();
  if (tid < AB_SRC && s0 + tid < ns) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s_red[q * AB_SRC + tid];
    partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * ns + s0 + tid] = t;
  }
}

static void adjoint_V_plan(int nft, int ns, int* nsplit, int* per) {
  int chunks = (ns + 63) / 64;   // (sized for the 64-wide kernel; the 16-wide one just has more chunks)
  int want = std::max(1, std::min(nft, 2048 / chunks));
  *per = (nft + want - 1) / want;
  *nsplit = (nft + *per - 1) / *per;
}

}  // namespace hydra

using namespace hydra;

#ifdef HYDRA_GRAM_STAMPS
extern "C" int hydra_debug_adj_trace(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_adj_trace), sizeof(g_adj_trace)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" {

int hydra_apply_R(const double* R, long long m, int ns, const double* x, const double* rowmul,
                  double* out, void* stream) {
  HYDRA_CHECK_ARG(R && x && out && m > 0 && ns > 0, "hydra_apply_R: bad arguments (m=%lld ns=%d)", m,
                  ns);
  int grid = (int)std::min<long long>((m + 3) / 4, 8192);
  apply_R_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const f64x2*>(R), m, ns, x,
                                                      reinterpret_cast<const f64x2*>(rowmul),
                                                      reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_apply_R");
  return HYDRA_OK;
}

size_t hydra_adjoint_workspace_bytes(long long m, int ns) {
  if (m <= 0 || ns <= 0) return 0;
  int nchunk;
  long long per;
  adjoint_plan(m, ns, &nchunk, &per);
  return (size_t)nchunk * ns * sizeof(double);
}

int hydra_adjoint_R(const double* R, long long m, int ns, const double* z, const double* scale,
                    const double* add, double* out, void* workspace, size_t workspace_bytes,
                    void* stream) {
  HYDRA_CHECK_ARG(R && z && out && m > 0 && ns > 0, "hydra_adjoint_R: bad arguments (m=%lld ns=%d)",
                  m, ns);
  size_t need = hydra_adjoint_workspace_bytes(m, ns);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_adjoint_R: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int nchunk;
  long long per;
  adjoint_plan(m, ns, &nchunk, &per);
  double* partial = static_cast<double*>(workspace);
  dim3 grid(nchunk, (ns + 256 * ADJ_COLS - 1) / (256 * ADJ_COLS));
  adjoint_R_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const f64x2*>(R), m, ns,
                                                        reinterpret_cast<const f64x2*>(z), per,
                                                        partial);
  HYDRA_CHECK_LAUNCH("hydra_adjoint_R");
  adjoint_reduce_kernel<<<(ns + 255) / 256, 256, 0, as_stream(stream)>>>(partial, nchunk, ns, scale,
                                                                         add, out);
  HYDRA_CHECK_LAUNCH("hydra_adjoint_R(reduce)");
  return HYDRA_OK;
}

int hydra_rhs_rows(const double* data, const double* model, const double* w,
                   const double* omega_n, const double* gg, long long m, double* z, void* stream) {
  HYDRA_CHECK_ARG(data && z && m > 0, "hydra_rhs_rows: bad arguments");
  int grid = (int)std::min<long long>((m + 255) / 256, 2048);
  rhs_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(data), reinterpret_cast<const f64x2*>(model), w,
      reinterpret_cast<const f64x2*>(omega_n), reinterpret_cast<const f64x2*>(gg), m,
      reinterpret_cast<f64x2*>(z));
  HYDRA_CHECK_LAUNCH("hydra_rhs_rows");
  return HYDRA_OK;
}

int hydra_scale_rows(const double* y, const double* w, long long m, double* z, void* stream) {
  HYDRA_CHECK_ARG(y && w && z && m > 0, "hydra_scale_rows: bad arguments");
  int grid = (int)std::min<long long>((m + 255) / 256, 2048);
  scale_rows_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const f64x2*>(y), w, m,
                                                         reinterpret_cast<f64x2*>(z));
  HYDRA_CHECK_LAUNCH("hydra_scale_rows");
  return HYDRA_OK;
}

static int launch_apply_V(const char* who, const double* V, int nf, int nt, int nant, int ns,
                          const int32_t* bl_i, const int32_t* bl_j, int nbl, const double* x,
                          const double* rowmul, double* out, const int32_t* seg, int nreg,
                          void* stream) {
  const f64x2* Vp = reinterpret_cast<const f64x2*>(V);
  const f64x2* rm = reinterpret_cast<const f64x2*>(rowmul);
  f64x2* op = reinterpret_cast<f64x2*>(out);
  size_t lds64 = ((size_t)nant * 65) * 16 + 64 * 8, lds16 = ((size_t)nant * 17) * 16 + 16 * 8;
  // few (f,t) (config 5: 2 per step, 61 075 baselines): spread the baseline chunks over the chip
  const long long base_wgs = (long long)nf * nt * (seg ? nreg : 1);
  const long long chunks = ((long long)nbl + 2047) / 2048;
  const unsigned nz = (unsigned)std::max<long long>(1, std::min<long long>({chunks, 1024 / base_wgs, 65535}));
  dim3 grid(nf * nt, seg ? nreg : 1, nz);
  if (lds64 <= 160 * 1024) {
    if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&apply_V_kernel<64>), lds64)) return rc_lds;
    apply_V_kernel<64><<<grid, 256, lds64, as_stream(stream)>>>(Vp, nf * nt, nant, ns, bl_i, bl_j,
                                                                nbl, x, rm, op, seg, nreg);
  } else {
    HYDRA_CHECK_ARG(lds16 <= 160 * 1024, "%s: nant=%d exceeds the LDS staging limit (600)", who,
                    nant);
    if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&apply_V_kernel<16>), lds16)) return rc_lds;
    apply_V_kernel<16><<<grid, 256, lds16, as_stream(stream)>>>(Vp, nf * nt, nant, ns, bl_i, bl_j,
                                                                nbl, x, rm, op, seg, nreg);
  }
  HYDRA_CHECK_LAUNCH(who);
  return HYDRA_OK;
}

static void apply_V_mfma_plan(int nft, int ns, int* nsplit, int* k_per_split) {
  int nk = (ns + 3) / 4;
  int want = (nk >= 64) ? 4 : 1;            // the 4 waves of a workgroup share one (f,t)
  while (want < 16 && (long long)nft * want < 4096 && nk / (2 * want) >= 16) want *= 2;
  *nsplit = want;
  *k_per_split = ((nk + want - 1) / want + 3) / 4 * 4;   // k-steps per wave, whole groups of 4
}

// source tiles per workgroup of apply_V_big_kernel: whole (f,t) unless there are few of them
static void apply_V_big_plan(int nft, int ns, int* nsplit, int* tiles_per_split) {
  const int ntile = (ns + AVB_SRC - 1) / AVB_SRC;
  int want = 1;
  while (want < 64 && (long long)nft * want < 1024 && ntile / (2 * want) >= 8) want *= 2;
  *tiles_per_split = (ntile + want - 1) / want;
  *nsplit = (ntile + *tiles_per_split - 1) / *tiles_per_split;
}

// the same for apply_V_sb_kernel, whose grid also runs over the super-blocks
static void apply_V_sb_plan(int nft, int ns, int nsb, int* nsplit, int* tiles_per_split) {
  const int ntile = (ns + AVB_SRC - 1) / AVB_SRC;
  int want = 1;
  while (want < 64 && (long long)nft * nsb * want < 2048 && ntile / (2 * want) >= 8) want *= 2;
  *tiles_per_split = (ntile + want - 1) / want;
  *nsplit = (ntile + *tiles_per_split - 1) / *tiles_per_split;
}

static constexpr int APPLY_V_BIG_MAX_NANT = 144;     // two LDS tiles of all antennas x 32 sources
static constexpr int APPLY_V_SB_MAX_NANT = 1024;     // super-block form: 32-bit offsets, table sizes

size_t hydra_apply_V_workspace_bytes(int nf, int nt, int nant, int ns) {
  if (nf <= 0 || nt <= 0 || nant <= 0 || ns <= 0 || nant > APPLY_V_SB_MAX_NANT) return 0;
  if ((size_t)nant * ns * 16 >= ((size_t)1 << 32)) return 0;   // 32-bit offsets inside one (f,t)
  if (nant > APPLY_V_BIG_MAX_NANT) {
    const int nr = (nant + 15) / 16, nsr = (nr + AVS_SB - 1) / AVS_SB;
    int nsplit, tps;
    apply_V_sb_plan(nf * nt, ns, nsr * (nsr + 1) / 2, &nsplit, &tps);
    return (size_t)nsplit * nf * nt * (16 * nr) * (16 * nr) * 16;
  }
  if (nant > 64) {
    int nsplit, tps;
    apply_V_big_plan(nf * nt, ns, &nsplit, &tps);
    size_t na = (size_t)(nant + 15) / 16 * 16;
    return (size_t)nsplit * nf * nt * na * na * 16;
  }
  int nsplit, kps;
  apply_V_mfma_plan(nf * nt, ns, &nsplit, &kps);
  size_t na = (size_t)(nant + 15) / 16 * 16;
  return (size_t)nsplit * nf * nt * na * na * 16;
}

int hydra_apply_V(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                  const int32_t* bl_j, int nbl, const double* x, const double* rowmul,
                  double* out, void* workspace, size_t workspace_bytes, void* stream) {
  HYDRA_CHECK_ARG(V && bl_i && bl_j && out && nf > 0 && nt > 0 && nant > 0 && ns > 0 && nbl > 0,
                  "hydra_apply_V: bad arguments");
  size_t need = hydra_apply_V_workspace_bytes(nf, nt, nant, ns);
  if (!workspace || need == 0 || !g_adjoint_mfma)   // pair-loop form (any Nant, no workspace)
    return launch_apply_V("hydra_apply_V", V, nf, nt, nant, ns, bl_i, bl_j, nbl, x, rowmul, out,
                          nullptr, 1, stream);
  if (workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_apply_V: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int nft = nf * nt, nsplit, kps;
  int nr = (nant + 15) / 16;
  if (nant > APPLY_V_BIG_MAX_NANT) {
    const int nsr = (nr + AVS_SB - 1) / AVS_SB, nsb = nsr * (nsr + 1) / 2;
    int tps;
    apply_V_sb_plan(nft, ns, nsb, &nsplit, &tps);
    HYDRA_CHECK_ARG((long long)nft * nsplit * nsb <= 2147483647LL, "hydra_apply_V: too many (f,t)");
    const size_t lds = 2 * ((size_t)AVS_ROWS * AVB_SRC * 16 + AVB_SRC * 16);
    hipStream_t st = as_stream(stream);
    f64x2* cm = static_cast<f64x2*>(workspace);
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&apply_V_sb_kernel), lds)) return rc;
    apply_V_sb_kernel<<<(unsigned)(nft * nsplit * nsb), AVB_THREADS, lds, st>>>(
        reinterpret_cast<const f64x2*>(V), nft, nant, ns, nr, x, nsplit, tps, nsr, cm);
    HYDRA_CHECK_LAUNCH("hydra_apply_V(antenna space, super-blocks)");
    size_t n = (size_t)nbl * nft;
    apply_V_gather_kernel<<<(unsigned)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(
        cm, nft, nr * 16, nsplit, bl_i, bl_j, nbl, reinterpret_cast<const f64x2*>(rowmul),
        reinterpret_cast<f64x2*>(out));
    HYDRA_CHECK_LAUNCH("hydra_apply_V(gather)");
    return HYDRA_OK;
  }
  if (nant > 64) {
    int tps;
    apply_V_big_plan(nft, ns, &nsplit, &tps);
    HYDRA_CHECK_ARG((long long)nft * nsplit <= 2147483647LL, "hydra_apply_V: too many (f,t)");
    const int nblk = nr * (nr + 1) / 2, maxb = (nblk + 7) / 8;
    const size_t lds = 2 * ((size_t)(16 * nr) * AVB_SRC * 16 + AVB_SRC * 16);
    hipStream_t st = as_stream(stream);
    const f64x2* Vp = reinterpret_cast<const f64x2*>(V);
    f64x2* cm = static_cast<f64x2*>(workspace);
    unsigned grid = (unsigned)(nft * nsplit);
#define HYDRA_AVB_LAUNCH(MB)                                                                       \
  do {                                                                                             \
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&apply_V_big_kernel<MB>), lds)) return rc; \
    apply_V_big_kernel<MB><<<grid, AVB_THREADS, lds, st>>>(Vp, nft, nant, ns, nr, x, nsplit, tps, cm); \
  } while (0)
    switch (maxb) {
      case 2: HYDRA_AVB_LAUNCH(2); break;
      case 3: HYDRA_AVB_LAUNCH(3); break;
      case 4: HYDRA_AVB_LAUNCH(4); break;
      case 5: HYDRA_AVB_LAUNCH(5); break;
      case 6: HYDRA_AVB_LAUNCH(6); break;
      default: HYDRA_AVB_LAUNCH(7); break;
    }
#undef HYDRA_AVB_LAUNCH
    HYDRA_CHECK_LAUNCH("hydra_apply_V(antenna space)");
    size_t n = (size_t)nbl * nft;
    apply_V_gather_kernel<<<(unsigned)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(
        cm, nft, nr * 16, nsplit, bl_i, bl_j, nbl, reinterpret_cast<const f64x2*>(rowmul),
        reinterpret_cast<f64x2*>(out));
    HYDRA_CHECK_LAUNCH("hydra_apply_V(gather)");
    return HYDRA_OK;
  }
  apply_V_mfma_plan(nft, ns, &nsplit, &kps);
  long long nwaves = (long long)nft * nsplit;
  unsigned grid = (unsigned)((nwaves + 3) / 4);
  hipStream_t st = as_stream(stream);
  const f64x2* Vp = reinterpret_cast<const f64x2*>(V);
  f64x2* cm = static_cast<f64x2*>(workspace);
  if (nr == 1) apply_V_mfma_kernel<1><<<grid, 256, 0, st>>>(Vp, nft, nant, ns, x, nsplit, kps, cm);
  else if (nr == 2) apply_V_mfma_kernel<2><<<grid, 256, 0, st>>>(Vp, nft, nant, ns, x, nsplit, kps, cm);
  else if (nr == 3) apply_V_mfma_kernel<3><<<grid, 256, 0, st>>>(Vp, nft, nant, ns, x, nsplit, kps, cm);
  else apply_V_mfma_kernel<4><<<grid, 256, 0, st>>>(Vp, nft, nant, ns, x, nsplit, kps, cm);
  HYDRA_CHECK_LAUNCH("hydra_apply_V(mfma)");
  size_t n = (size_t)nbl * nft;
  apply_V_gather_kernel<<<(unsigned)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(
      cm, nft, nr * 16, nsplit, bl_i, bl_j, nbl, reinterpret_cast<const f64x2*>(rowmul),
      reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_apply_V(gather)");
  return HYDRA_OK;
}

int hydra_vis_regions(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                      const int32_t* bl_j, int nbl, const int32_t* seg_host, const int32_t* seg,
                      int nreg, double* out, void* stream) {
  HYDRA_CHECK_ARG(V && bl_i && bl_j && seg_host && seg && out && nf > 0 && nt > 0 && nant > 0 &&
                      ns > 0 && nbl > 0 && nreg > 0 && nreg <= 65535,
                  "hydra_vis_regions: bad arguments");
  HYDRA_CHECK_ARG(seg_host[0] == 0 && seg_host[nreg] == ns,
                  "hydra_vis_regions: segments must cover sources 0..%d (got %d..%d)", ns,
                  seg_host[0], seg_host[nreg]);
  for (int r = 0; r < nreg; ++r)
    HYDRA_CHECK_ARG(seg_host[r] <= seg_host[r + 1], "hydra_vis_regions: segment %d is reversed", r);
  return launch_apply_V("hydra_vis_regions", V, nf, nt, nant, ns, bl_i, bl_j, nbl, nullptr, nullptr,
                        out, seg, nreg, stream);
}

static constexpr int ADJ_BIG_MAX_NAP = 160;     // adjoint_V_big_kernel: two tiles of all antennas in LDS

static void adjbig_plan(int nft, int ns, int* nsplit, int* per) {
  int tiles = (ns + AB_SRC - 1) / AB_SRC;
  int want = std::max(1, std::min(nft, 2048 / std::max(tiles, 1)));
  *per = (nft + want - 1) / want;
  *nsplit = (nft + *per - 1) / *per;
}

size_t hydra_adjoint_V_antspace_workspace_bytes(int nf, int nt, int nant, int nbl, int ns) {
  if (nf <= 0 || nt <= 0 || nant <= 0 || nbl <= 0 || ns <= 0) return 0;
  int nft = nf * nt, nsplit, per;
  adjbig_plan(nft, ns, &nsplit, &per);
  const int nr = (nant + 15) / 16, nk = 4 * nr, nap = 16 * nr;
  const size_t nsteps = (size_t)adjbig_prefix(nr, nk);
  const size_t nsr = (nr + AS_SB - 1) / AS_SB;      // partial sums per super-row above 160 antennas
  const size_t nparts = (size_t)nsplit * (nap > ADJ_BIG_MAX_NAP ? nsr : 1);
  return align_up(nparts * ns * sizeof(double), 256) +                  // partial sums
         align_up((size_t)nft * nbl * 16, 256) +                        // z transposed
         align_up((size_t)nap * nap * sizeof(int32_t), 256) +           // pair table
         align_up((size_t)nft * nsteps * 64 * 16, 256) +                // Zc fragments (re, im)
         align_up((size_t)nft * nsteps * 64 * 8, 256);                  // Zc fragments (re + im)
}

int hydra_adjoint_V_antspace(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                             const int32_t* bl_j, int nbl, const double* z, const double* scale,
                             const double* add, double* out, void* workspace,
                             size_t workspace_bytes, void* stream) {
  HYDRA_CHECK_ARG(V && bl_i && bl_j && z && out && nf > 0 && nt > 0 && nant > 0 && ns > 0 && nbl > 0,
                  "hydra_adjoint_V_antspace: bad arguments");
  const int nr = (nant + 15) / 16, nk = 4 * nr, nap = 16 * nr;
  const bool super_rows = nap > ADJ_BIG_MAX_NAP;
  const size_t lds = std::max<size_t>((size_t)2 * nap * AB_SRC * 16, (size_t)16 * AB_SRC * 8);
  HYDRA_CHECK_ARG(nant <= 1024, "hydra_adjoint_V_antspace: nant=%d exceeds the table limit (1024)", nant);
  HYDRA_CHECK_ARG((size_t)nant * ns * 16 < ((size_t)1 << 32),
                  "hydra_adjoint_V_antspace: one (f,t) slice of V must be below 4 GiB");
  size_t need = hydra_adjoint_V_antspace_workspace_bytes(nf, nt, nant, nbl, ns);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_adjoint_V_antspace: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  int nft = nf * nt, nsplit, per;
  adjbig_plan(nft, ns, &nsplit, &per);
  HYDRA_CHECK_ARG(nsplit <= 65535, "hydra_adjoint_V_antspace: too many splits");
  const int nsteps = adjbig_prefix(nr, nk);
  unsigned char* w = static_cast<unsigned char*>(workspace);
  const int nsr = (nr + AS_SB - 1) / AS_SB;
  const size_t nparts = (size_t)nsplit * (super_rows ? nsr : 1);
  double* partial = reinterpret_cast<double*>(w);
  w += align_up(nparts * ns * sizeof(double), 256);
  f64x2* zt = reinterpret_cast<f64x2*>(w);
  w += align_up((size_t)nft * nbl * 16, 256);
  int32_t* tab = reinterpret_cast<int32_t*>(w);
  w += align_up((size_t)nap * nap * sizeof(int32_t), 256);
  f64x2* zri = reinterpret_cast<f64x2*>(w);
  w += align_up((size_t)nft * nsteps * 64 * 16, 256);
  double* zs = reinterpret_cast<double*>(w);
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(tab, 0xff, (size_t)nap * nap * sizeof(int32_t), st);   // -1
  if (e != hipSuccess)
    return fail(HYDRA_ERR_HIP, "hydra_adjoint_V_antspace: %s", hipGetErrorString(e));
  adjbig_table_kernel<<<std::min((nbl + 255) / 256, 1024), 256, 0, st>>>(bl_i, bl_j, nbl, nap, tab);
  ctranspose_kernel<<<2048, 256, 0, st>>>(reinterpret_cast<const f64x2*>(z), nbl, nft, zt);
  adjbig_frag_kernel<<<dim3((nsteps + 3) / 4, std::min(nft, 4096)), 256, 0, st>>>(
      zt, tab, nbl, nft, nr, nsteps, zri, zs);
  HYDRA_CHECK_LAUNCH("hydra_adjoint_V_antspace(fragments)");
  if (super_rows) {
    dim3 grid((ns + AB_SRC - 1) / AB_SRC, nsplit, nsr);
    adjoint_V_sb_kernel<<<grid, AB_THREADS, 0, st>>>(reinterpret_cast<const f64x2*>(V), nft, nant, ns,
                                                     nr, zri, zs, per, nsr, partial);
  } else {
    if (int rc = reserve_lds(reinterpret_cast<const void*>(&adjoint_V_big_kernel), lds)) return rc;
    dim3 grid((ns + AB_SRC - 1) / AB_SRC, nsplit);
    adjoint_V_big_kernel<<<grid, AB_THREADS, lds, st>>>(reinterpret_cast<const f64x2*>(V), nft, nant, ns,
                                                        nr, zri, zs, per, partial);
  }
  HYDRA_CHECK_LAUNCH("hydra_adjoint_V_antspace");
  adjoint_reduce_kernel<<<(ns + 255) / 256, 256, 0, st>>>(partial, (int)nparts, ns, scale, add, out);
  HYDRA_CHECK_LAUNCH("hydra_adjoint_V_antspace(reduce)");
  return HYDRA_OK;
}

int hydra_debug_adjoint_mfma(int enable) {
  int old = g_adjoint_mfma;
  g_adjoint_mfma = enable ? 1 : 0;
  return old;
}

size_t hydra_adjoint_V_workspace_bytes(int nf, int nt, int nbl, int ns) {
  if (nf <= 0 || nt <= 0 || nbl <= 0 || ns <= 0) return 0;
  int nsplit, per;
  adjoint_V_plan(nf * nt, ns, &nsplit, &per);
  return align_up((size_t)nsplit * ns * sizeof(double), 256) + (size_t)nf * nt * nbl * 16;
}

int hydra_adjoint_V(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                    const int32_t* bl_j, int nbl, const double* z, const double* scale,
                    const double* add, double* out, void* workspace, size_t workspace_bytes,
                    void* stream) {
  HYDRA_CHECK_ARG(V && bl_i && bl_j && z && out && nf > 0 && nt > 0 && nant > 0 && ns > 0 && nbl > 0,
                  "hydra_adjoint_V: bad arguments");
  size_t need = hydra_adjoint_V_workspace_bytes(nf, nt, nbl, ns);
  if (!workspace || workspace_bytes < need)
    return fail(HYDRA_ERR_WORKSPACE, "hydra_adjoint_V: workspace too small (%zu < %zu bytes)",
                workspace_bytes, need);
  size_t lds64 = (size_t)nant * 64 * 16 + 256 * 8, lds16 = (size_t)nant * 16 * 16 + 256 * 8;
  bool wide = lds64 <= 160 * 1024;
  HYDRA_CHECK_ARG(wide || lds16 <= 160 * 1024,
                  "hydra_adjoint_V: nant=%d exceeds the LDS staging limit (630)", nant);
  int nft = nf * nt, nsplit, per;
  adjoint_V_plan(nft, ns, &nsplit, &per);
  HYDRA_CHECK_ARG(nsplit <= 65535, "hydra_adjoint_V: too many splits");
  double* partial = static_cast<double*>(workspace);
  f64x2* zt = reinterpret_cast<f64x2*>(static_cast<unsigned char*>(workspace) +
                                       align_up((size_t)nsplit * ns * sizeof(double), 256));
  hipStream_t st = as_stream(stream);
  ctranspose_kernel<<<2048, 256, 0, st>>>(reinterpret_cast<const f64x2*>(z), nbl, nft, zt);
  HYDRA_CHECK_LAUNCH("hydra_adjoint_V(transpose)");
  const f64x2* Vp = reinterpret_cast<const f64x2*>(V);
  // antenna-space product on the matrix cores (32-bit byte offsets within one (f,t) slice)
  if (nant <= 64 && g_adjoint_mfma && (size_t)nant * ns * 16 < ((size_t)1 << 32)) {
    int nr = (nant + 15) / 16;
    dim3 grid((ns + 63) / 64, nsplit);
#define HYDRA_ADJ_MFMA(NR_)                                                                      \
    {                                                                                            \
      size_t lds = (size_t)2 * 16 * NR_ * (16 * NR_ + 1) * 16 + 16 + 64 * NR_;                    \
      if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&adjoint_V_mfma_kernel<NR_>), lds)) \
        return rc_lds;                                                                           \
      adjoint_V_mfma_kernel<NR_><<<grid, 256, lds, st>>>(Vp, nft, nant, ns, bl_i, bl_j, nbl, zt,   \
                                                         per, partial);                          \
    }
    if (nr == 1) HYDRA_ADJ_MFMA(1)
    else if (nr == 2) HYDRA_ADJ_MFMA(2)
    else if (nr == 3) HYDRA_ADJ_MFMA(3)
    else HYDRA_ADJ_MFMA(4)
#undef HYDRA_ADJ_MFMA
  } else if (wide) {
    if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&adjoint_V_kernel<64>), lds64)) return rc_lds;
    dim3 grid((ns + 63) / 64, nsplit);
    adjoint_V_kernel<64><<<grid, 256, lds64, st>>>(Vp, nft, nant, ns, bl_i, bl_j, nbl, zt, per, partial);
  } else {
    if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&adjoint_V_kernel<16>), lds16)) return rc_lds;
    dim3 grid((ns + 15) / 16, nsplit);
    adjoint_V_kernel<16><<<grid, 256, lds16, st>>>(Vp, nft, nant, ns, bl_i, bl_j, nbl, zt, per, partial);
  }
  HYDRA_CHECK_LAUNCH("hydra_adjoint_V");
  adjoint_reduce_kernel<<<(ns + 255) / 256, 256, 0, st>>>(partial, nsplit, ns, scale, add, out);
  HYDRA_CHECK_LAUNCH("hydra_adjoint_V(reduce)");
  return HYDRA_OK;
}

}  // extern "C"
