// Spherical-harmonic response (hydra/vis_simulator.py:571-736, hydra/sh_sampler.py:117-174)
// on gfx950:
//   * HEALPix RING pixel centres and the real-field mode maps alm2map(e_n * {1, i}) * pixel
//     area, generated on the device (normalised Legendre recurrence per pixel);
//   * the pixel -> mode projection  out[b,ft,n] = sum_p conj(V_i,p) V_j,p Y[p,n]  as a real
//     GEMM on the FP64 matrix cores with the left operand generated from the antenna
//     voltages staged in LDS (the per-pixel visibility cube is never materialised).
//     Algorithmic work: 4 * Nbl*Nf*Nt * Npix * Nm flops.  Bound: FP64 MFMA.
#include <algorithm>

#include "common.h"

namespace hydra {

constexpr double kPi = 3.141592653589793;

__device__ __forceinline__ long long isqrt_ll(long long v) {
  long long r = (long long)sqrt((double)v);
  while (r * r > v) --r;
  while ((r + 1) * (r + 1) <= v) ++r;
  return r;
}

// HEALPix RING pix2ang (Gorski et al. 2005; same branches as healpix_cxx pix2loc)
__device__ __forceinline__ void pix2ang_ring(long long nside, long long pix, double* theta,
                                             double* phi) {
  long long npix = 12 * nside * nside;
  long long ncap = 2 * nside * (nside - 1);
  double fact2 = 4.0 / (double)npix;
  double fact1 = (double)(2 * nside) * fact2;
  if (pix < ncap) {
    long long iring = (1 + isqrt_ll(1 + 2 * pix)) >> 1;
    long long iphi = (pix + 1) - 2 * iring * (iring - 1);
    double tmp = (double)(iring * iring) * fact2;
    double z = 1.0 - tmp;
    *theta = (z > 0.99) ? atan2(sqrt(tmp * (2.0 - tmp)), z) : acos(z);
    *phi = ((double)iphi - 0.5) * (0.5 * kPi) / (double)iring;
  } else if (pix < npix - ncap) {
    long long ip = pix - ncap;
    long long nl4 = 4 * nside;
    long long tmp = ip / nl4;
    long long iring = tmp + nside;
    long long iphi = ip - nl4 * tmp + 1;
    double fodd = ((iring + nside) & 1) ? 1.0 : 0.5;
    double z = (double)(2 * nside - iring) * fact1;
    *theta = acos(z);
    *phi = ((double)iphi - fodd) * kPi * 0.75 * fact1;
  } else {
    long long ip = npix - pix;
    long long iring = (1 + isqrt_ll(2 * ip - 1)) >> 1;
    long long iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
    double tmp = (double)(iring * iring) * fact2;
    double z = tmp - 1.0;
    *theta = (z < -0.99) ? atan2(sqrt(tmp * (2.0 - tmp)), z) : acos(z);
    *phi = ((double)iphi - 0.5) * (0.5 * kPi) / (double)iring;
  }
}

__global__ void __launch_bounds__(256)
pix2ang_kernel(int nside, double* __restrict__ theta, double* __restrict__ phi) {
  long long npix = 12LL * nside * nside;
  long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= npix) return;
  double t, f;
  pix2ang_ring(nside, p, &t, &f);
  theta[p] = t;
  phi[p] = f;
}

// Y[p, col]: thread <-> pixel, loops over m then l with the normalised recurrence
//   lam_mm = -lam_{m-1,m-1} sin(th) sqrt((2m+1)/(2m)),  lam_00 = sqrt(1/4pi)
//   lam_{m+1,m} = cos(th) sqrt(2m+3) lam_mm
//   lam_lm = sqrt((4l^2-1)/(l^2-m^2)) (cos(th) lam_{l-1,m} - sqrt(((l-1)^2-m^2)/(4(l-1)^2-1)) lam_{l-2,m})
// healpy index of (l,m): m(2 lmax+1-m)/2 + l.  Real-part column n holds lam (m=0) or
// 2 lam cos(m phi); imaginary-part column holds -2 lam sin(m phi) (m>0; zero for m=0).
__global__ void __launch_bounds__(256)
mode_maps_kernel(int lmax, int nside, double scale, int packed, int ncols,
                 double* __restrict__ ymap) {
  long long npix = 12LL * nside * nside;
  long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= npix) return;
  double theta, phi;
  pix2ang_ring(nside, p, &theta, &phi);
  double x = cos(theta), s = sin(theta);
  int nalm = (lmax + 1) * (lmax + 2) / 2;
  double* row = ymap + (size_t)p * ncols;
  int im_off = packed ? nalm - (lmax + 1) : nalm;  // column of imag part = idx + im_off
  double pmm = sqrt(1.0 / (4.0 * kPi));
  for (int m = 0; m <= lmax; ++m) {
    if (m > 0) pmm = -pmm * s * sqrt((2.0 * m + 1.0) / (2.0 * m));
    double cm = 1.0, sm = 0.0;
    if (m > 0) sincos((double)m * phi, &sm, &cm);
    double fre = (m == 0) ? 1.0 : 2.0 * cm;
    double fim = (m == 0) ? 0.0 : -2.0 * sm;
    int idx = m * (2 * lmax + 1 - m) / 2 + m;
    double p0 = 0.0, p1 = pmm;
    for (int l = m; l <= lmax; ++l) {
      double val;
      if (l == m) {
        val = pmm;
      } else if (l == m + 1) {
        val = x * sqrt(2.0 * m + 3.0) * pmm;
      } else {
        double a = sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
        double b = sqrt((((double)l - 1.0) * (l - 1.0) - (double)m * m) /
                        (4.0 * ((double)l - 1.0) * (l - 1.0) - 1.0));
        val = a * (x * p1 - b * p0);
      }
      p0 = p1;
      p1 = val;
      row[idx] = (val * fre) * scale;
      if (m > 0 || !packed) row[idx + im_off] = (val * fim) * scale;
      ++idx;
    }
  }
}

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- pixel -> mode projection -----------------------------------------------------------
// Per (f,t): C (Nbl x Nm complex) = P (Nbl x Npix complex, generated) * Y (Npix x Nm real).
// MFMA k index = pixel.  A-operand lane map: (row = l&15 -> baseline, k = l>>4 -> pixel);
// B-operand: (k = l>>4 -> pixel, col = l&15 -> mode) read from Y[p][n] (128-B segments).
// Real and imaginary parts of a baseline's row go to two accumulators held by the SAME lane,
// so the epilogue stores complex (re, im) pairs of 16 B, 256 B contiguous per 16 lanes.
// Workgroup: 4 waves as 2 (baselines) x 2 (modes); wave tile 32 baselines x 64 modes.
constexpr int SH_PC = 64;         // pixels staged per pass
constexpr int SH_PAD = SH_PC + 1;  // padded LDS row in 16-B units
constexpr int SH_BT = 64;         // baselines per workgroup
constexpr int SH_NT = 128;        // modes per workgroup

__global__ void __launch_bounds__(256, 2)
sh_project_kernel(const f64x2* __restrict__ V, int nft, int nant, int npix,
                  const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                  const double* __restrict__ ymap, int nm, const double* __restrict__ scale_f,
                  int nt, f64x2* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  f64x2* s_v = reinterpret_cast<f64x2*>(smem);  // [nant][SH_PAD]
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int wb = wave >> 1, wn = wave & 1;
  int b0 = blockIdx.x * SH_BT + wb * 32;
  int n0 = blockIdx.y * SH_NT + wn * 64;
  int ft = blockIdx.z;
  const f64x2* vft = V + (size_t)ft * nant * npix;
  int kp = lane >> 4, lr = lane & 15;

  // this lane's two baselines (A-operand rows) and four mode columns (B-operand)
  int ai[2], aj[2];
  bool aok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int b = b0 + q * 16 + lr;
    aok[q] = b < nbl;
    ai[q] = aok[q] ? bl_i[b] : 0;
    aj[q] = aok[q] ? bl_j[b] : 0;
  }
  int ncol[4];
  double nmask[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int n = n0 + q * 16 + lr;
    nmask[q] = (n < nm) ? 1.0 : 0.0;
    ncol[q] = min(n, nm - 1);
  }
  f64x4 acc_re[2][4], acc_im[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc_re[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
      acc_im[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    }

  for (int p0 = 0; p0 < npix; p0 += SH_PC) {
    __syncthreads();
    int p = p0 + lane;
    for (int a = wave; a < nant; a += 4)
      s_v[a * SH_PAD + lane] = (p < npix) ? vft[(size_t)a * npix + p] : f64x2{0.0, 0.0};
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < SH_PC; k += 4) {
      int pk = p0 + k + kp;
      int pk_c = min(pk, npix - 1);
      double are[2], aim[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f64x2 r = cmul_conj(s_v[ai[q] * SH_PAD + k + kp], s_v[aj[q] * SH_PAD + k + kp]);
        are[q] = aok[q] ? r.x : 0.0;
        aim[q] = aok[q] ? r.y : 0.0;
      }
      const double* yrow = ymap + (size_t)pk_c * nm;
      double bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = yrow[ncol[q]] * nmask[q];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc_re[i][j] = mfma(are[i], bv[j], acc_re[i][j]);
          acc_im[i][j] = mfma(aim[i], bv[j], acc_im[i][j]);
        }
    }
  }
  // D[(l>>4) + 4g][l&15]: row -> baseline, col -> mode
  double sc = scale_f ? scale_f[ft / nt] : 1.0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int b = b0 + i * 16 + kp + 4 * g;
      if (b >= nbl) continue;
      f64x2* orow = out + ((size_t)b * nft + ft) * nm;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = n0 + j * 16 + lr;
        if (n < nm) orow[n] = f64x2{acc_re[i][j][g] * sc, acc_im[i][j][g] * sc};
      }
    }
}


// ---- pixel -> mode projection, tiled form --------------------------------------------------
// Same contraction as sh_project_kernel, restructured for the matrix pipe:
//   * a workgroup owns an ANTENNA-BLOCK PAIR (8 antennas i x 16 antennas j = 128 baseline
//     slots) x 128 modes, so per staged pixel it needs 24 voltages (384 B) instead of all Nant,
//     and the left operand of a 16-row MFMA block is one broadcast V_i times 16 different V_j;
//   * pixels below the horizon at time t have zero voltage (vis_simulator.py:237,272): only the
//     compacted list of pixels above it is walked (half the sky), an exact saving;
//   * V and Y tiles of 32 pixels are double-buffered in LDS and filled by LDS-DMA
//     (global_load_lds_dwordx4) one chunk ahead; the per-lane SOURCE address does the pixel
//     gather and the bank swizzles (V: pixel k of antenna row r sits at slot (k + r) & 31;
//     Y: odd pixel rows are rotated by 16 modes), so every LDS read is conflict free.
// 8 waves = 4 (pairs of i) x 2 (64 modes); wave tile 32 slots x 64 modes, Re and Im
// accumulated by the same lane (16-B complex stores in the epilogue).
constexpr int SP_PC = 32;                    // pixels per chunk
constexpr int SP_NI = 8, SP_NJ = 16;         // antenna block sizes
constexpr int SP_ROWS = SP_NI + SP_NJ;       // staged antenna rows
constexpr int SP_MT = 128;                   // modes per workgroup
constexpr int SP_VBUF = SP_ROWS * SP_PC;     // complex elements per V buffer
constexpr int SP_YBUF = SP_PC * SP_MT;       // doubles per Y buffer

__global__ void __launch_bounds__(512, 1)
sh_project_tiled_kernel(const f64x2* __restrict__ V, int nant, int npix,
                        const int32_t* __restrict__ pixidx, const int32_t* __restrict__ npix_t,
                        int npix_max, const int32_t* __restrict__ tiles,
                        const int32_t* __restrict__ bmap, const double* __restrict__ ymap_p, int nmp,
                        int nm, const double* __restrict__ scale_f, int nt, int f0, int nf_total,
                        f64x2* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  f64x2* s_v = reinterpret_cast<f64x2*>(smem);                                  // [2][24][32]
  double* s_y = reinterpret_cast<double*>(smem + (size_t)2 * SP_VBUF * 16);     // [2][32][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wb = wave >> 1, wn = wave & 1;
  const int kp = lane >> 4, lc = lane & 15;
  const int tile = blockIdx.y, n0 = blockIdx.x * SP_MT, ft = blockIdx.z;
  const int t = ft % nt;
  const int i0 = tiles[2 * tile], j0 = tiles[2 * tile + 1];
  const f64x2* vft = V + (size_t)ft * nant * npix;
  const int32_t* pix = pixidx + (size_t)t * npix_max;
  const int count = npix_t[t];
  const int nchunk = (count + SP_PC - 1) / SP_PC;

  // DMA of chunk c into buffer c & 1.  Pieces: 12 for V (two antenna rows each), 32 for Y (one
  // pixel row each); wave w takes pieces w, w+8, ...
  auto issue = [&](int c) {
    int buf = c & 1;
    int kidx = c * SP_PC + (lane & 31);
    int mypix = pix[min(kidx, count - 1)];      // lanes l and l+32 hold the same entry
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      int piece = wave + 8 * it;                // 0..47, 44 used
      if (piece < 12) {
        int row = 2 * piece + (lane >> 5);
        int ant = (row < SP_NI) ? i0 + row : j0 + (row - SP_NI);
        ant = min(ant, nant - 1);
        int k = ((lane & 31) - row) & 31;        // slot q holds pixel (q - row) & 31
        int p = __shfl(mypix, k, 64);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(vft + (size_t)ant * npix + p),
            (__attribute__((address_space(3))) void*)(s_v + buf * SP_VBUF + piece * 64), 16, 0, 0);
      } else if (piece < 12 + SP_PC) {
        int r = piece - 12;
        int p = __shfl(mypix, r, 64);
        int src = 2 * ((lane - 8 * (r & 1)) & 63);   // odd rows rotated by 16 modes
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ymap_p + (size_t)p * nmp + n0 + src),
            (__attribute__((address_space(3))) void*)(s_y + buf * SP_YBUF + r * SP_MT), 16, 0, 0);
      }
    }
  };

  f64x4 acc_re[2][4], acc_im[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc_re[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
      acc_im[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    }

  if (nchunk > 0) issue(0);
  const int rj = SP_NI + lc;                      // this lane's V_j row
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                              // chunk c landed; the other buffer is free
    if (c + 1 < nchunk) issue(c + 1);
    const f64x2* bv = s_v + (c & 1) * SP_VBUF;
    const double* by = s_y + (c & 1) * SP_YBUF;
    const int left = count - c * SP_PC;           // valid pixels in this chunk
#pragma unroll
    for (int k = 0; k < SP_PC; k += 4) {
      int kk = k + kp;
      double m = (kk < left) ? 1.0 : 0.0;         // tail of the pixel list
      f64x2 vj = bv[rj * SP_PC + ((kk + rj) & 31)];
      double are[2], aim[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int ri = wb * 2 + q;
        f64x2 vi = bv[ri * SP_PC + ((kk + ri) & 31)];
        f64x2 r = cmul_conj(vi, vj);
        are[q] = r.x * m;
        aim[q] = r.y * m;
      }
      const double* yrow = by + kk * SP_MT;
      int rot = 16 * (kk & 1);
      double bvv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bvv[j] = yrow[(wn * 64 + j * 16 + lc + rot) & 127];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc_re[q][j] = mfma(are[q], bvv[j], acc_re[q][j]);
          acc_im[q][j] = mfma(aim[q], bvv[j], acc_im[q][j]);
        }
    }
  }
  // D[(l>>4) + 4g][l&15]: row -> j slot, col -> mode
  const int f = ft / nt;
  double sc = scale_f ? scale_f[f] : 1.0;
  const size_t nft_total = (size_t)nf_total * nt;
  const size_t ft_out = (size_t)(f0 + f) * nt + t;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int slot = (wb * 2 + q) * SP_NJ + kp + 4 * g;
      int b = bmap[(size_t)tile * (SP_NI * SP_NJ) + slot];
      if (b < 0) continue;
      f64x2* orow = out + ((size_t)b * nft_total + ft_out) * nm;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int n = n0 + wn * 64 + j * 16 + lc;
        if (n < nm) orow[n] = f64x2{acc_re[q][j][g] * sc, acc_im[q][j][g] * sc};
      }
    }
}

}  // namespace hydra

using namespace hydra;

extern "C" {

int hydra_healpix_pix2ang(int nside, double* theta, double* phi, void* stream) {
  HYDRA_CHECK_ARG(nside > 0 && nside <= 8192 && theta && phi, "hydra_healpix_pix2ang: bad arguments");
  long long npix = 12LL * nside * nside;
  pix2ang_kernel<<<(unsigned)((npix + 255) / 256), 256, 0, as_stream(stream)>>>(nside, theta, phi);
  HYDRA_CHECK_LAUNCH("hydra_healpix_pix2ang");
  return HYDRA_OK;
}

int hydra_sh_mode_maps(int lmax, int nside, double amplitude, int packed, double* ymap,
                       void* stream) {
  HYDRA_CHECK_ARG(lmax >= 0 && nside > 0 && nside <= 8192 && ymap, "hydra_sh_mode_maps: bad arguments");
  long long npix = 12LL * nside * nside;
  int nalm = (lmax + 1) * (lmax + 2) / 2;
  int ncols = packed ? 2 * nalm - (lmax + 1) : 2 * nalm;
  double scale = (4.0 * kPi / (double)npix) * amplitude;
  hipStream_t s = as_stream(stream);
  if (!packed) {
    hipError_t e = hipMemsetAsync(ymap, 0, (size_t)npix * ncols * sizeof(double), s);
    if (e != hipSuccess) return fail(HYDRA_ERR_HIP, "hydra_sh_mode_maps: %s", hipGetErrorString(e));
  }
  mode_maps_kernel<<<(unsigned)((npix + 255) / 256), 256, 0, s>>>(lmax, nside, scale, packed, ncols,
                                                                  ymap);
  HYDRA_CHECK_LAUNCH("hydra_sh_mode_maps");
  return HYDRA_OK;
}

int hydra_sh_project(const double* V, int nf, int nt, int nant, int npix, const int32_t* bl_i,
                     const int32_t* bl_j, int nbl, const double* ymap, int nm,
                     const double* scale_f, double* out, void* stream) {
  HYDRA_CHECK_ARG(V && bl_i && bl_j && ymap && out, "hydra_sh_project: null pointer argument");
  HYDRA_CHECK_ARG(nf > 0 && nt > 0 && nant > 0 && npix > 0 && nbl > 0 && nm > 0,
                  "hydra_sh_project: sizes must be positive");
  HYDRA_CHECK_ARG((long long)nf * nt <= 65535, "hydra_sh_project: nf*nt must be <= 65535 per call");
  size_t lds = (size_t)nant * SH_PAD * 16;
  HYDRA_CHECK_ARG(lds <= 160 * 1024, "hydra_sh_project: nant=%d exceeds the LDS staging limit (157)",
                  nant);
  dim3 grid((nbl + SH_BT - 1) / SH_BT, (nm + SH_NT - 1) / SH_NT, nf * nt);
  HYDRA_CHECK_ARG(grid.y <= 65535, "hydra_sh_project: too many modes");
  if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&sh_project_kernel), lds)) return rc_lds;
  sh_project_kernel<<<grid, 256, lds, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(V), nf * nt, nant, npix, bl_i, bl_j, nbl, ymap, nm, scale_f, nt,
      reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_sh_project");
  return HYDRA_OK;
}


int hydra_sh_project_tiled(const double* V, int nf, int nt, int nant, int npix,
                           const int32_t* pixidx, const int32_t* npix_t, int npix_max,
                           const int32_t* tiles, const int32_t* bmap, int ntile,
                           const double* ymap_padded, int nm_padded, int nm, const double* scale_f,
                           int f0, int nf_total, double* out, void* stream) {
  HYDRA_CHECK_ARG(V && pixidx && npix_t && tiles && bmap && ymap_padded && out,
                  "hydra_sh_project_tiled: null pointer argument");
  HYDRA_CHECK_ARG(nf > 0 && nt > 0 && nant > 0 && npix > 0 && npix_max > 0 && ntile > 0 && nm > 0,
                  "hydra_sh_project_tiled: sizes must be positive");
  HYDRA_CHECK_ARG(nm_padded % SP_MT == 0 && nm_padded >= nm,
                  "hydra_sh_project_tiled: nm_padded must be a multiple of %d and >= nm", SP_MT);
  HYDRA_CHECK_ARG(f0 >= 0 && f0 + nf <= nf_total, "hydra_sh_project_tiled: channel block out of range");
  HYDRA_CHECK_ARG((long long)nf * nt <= 65535 && ntile <= 65535,
                  "hydra_sh_project_tiled: nf*nt and ntile must be <= 65535 per call");
  size_t lds = (size_t)2 * SP_VBUF * 16 + (size_t)2 * SP_YBUF * 8;
  if (int rc = reserve_lds(reinterpret_cast<const void*>(&sh_project_tiled_kernel), lds)) return rc;
  dim3 grid(nm_padded / SP_MT, ntile, nf * nt);
  sh_project_tiled_kernel<<<grid, 512, lds, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(V), nant, npix, pixidx, npix_t, npix_max, tiles, bmap,
      ymap_padded, nm_padded, nm, scale_f, nt, f0, nf_total, reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_sh_project_tiled");
  return HYDRA_OK;
}

}  // extern "C"
