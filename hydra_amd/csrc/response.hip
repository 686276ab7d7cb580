// Visibility response generator for gfx950: geometry, antenna voltages, and the
// per-source baseline products.  Follows the operation order of
// hydra/vis_simulator.py:214-289 (see oracle/response.py).
//
// Roofline: every kernel here is HBM-write bound (16 B per complex element
// produced); the sincos/exp work is O(Nant*Ns) per (f,t) against O(Nbl*Ns)
// elements written.
#include <stdarg.h>

#include "common.h"

namespace hydra {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

constexpr double kC = 299792458.0;  // astropy.constants.c.value (vis_simulator.py:262)
constexpr double kPi = 3.141592653589793;

// ---------------------------------------------------------------- geometry

__global__ void crd_eq_kernel(const double* __restrict__ ra, const double* __restrict__ dec,
                              int ns, double* __restrict__ out) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  double sr, cr, sd, cd;
  sincos(ra[s], &sr, &cr);
  sincos(dec[s], &sd, &cd);
  out[s] = cr * cd;
  out[ns + s] = sr * cd;
  out[2 * (size_t)ns + s] = sd;
}

__global__ void eq2tops_kernel(const double* __restrict__ lsts, int nt, double lat,
                               double* __restrict__ out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  double sh, ch, sl, cl;
  sincos(lsts[t], &sh, &ch);
  sincos(lat, &sl, &cl);
  double* m = out + 9 * (size_t)t;
  m[0] = -sh;      m[1] = ch;       m[2] = 0.0;
  m[3] = -sl * ch; m[4] = -sl * sh; m[5] = cl;
  m[6] = cl * ch;  m[7] = cl * sh;  m[8] = sl;
}

struct Top {
  double tx, ty, tz;
};
__device__ __forceinline__ Top rotate(const double* __restrict__ m, double cx, double cy,
                                      double cz) {
  Top o;
  o.tx = m[0] * cx + m[1] * cy + m[2] * cz;
  o.ty = m[3] * cx + m[4] * cy + m[5] * cz;
  o.tz = m[6] * cx + m[7] * cy + m[8] * cz;
  return o;
}
// matvis.coordinates.enu_to_az_za(..., orientation="uvbeam")
__device__ __forceinline__ void az_za_uvbeam(double e, double n, double* az, double* za) {
  double lsqr = n * n + e * e;
  double zeta = lsqr < 1.0 ? sqrt(1.0 - lsqr) : 0.0;
  *za = 0.5 * kPi - asin(zeta);
  double a = 0.5 * kPi - atan2(e, n);
  a = fmod(a, 2.0 * kPi);
  if (a < 0.0) a += 2.0 * kPi;
  *az = a;
}

__global__ void az_za_kernel(const double* __restrict__ eq2tops, int nt,
                             const double* __restrict__ crd_eq, int ns, double* __restrict__ az,
                             double* __restrict__ za, uint8_t* __restrict__ above) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int t = blockIdx.y;
  if (s >= ns) return;
  Top c = rotate(eq2tops + 9 * (size_t)t, crd_eq[s], crd_eq[ns + s], crd_eq[2 * (size_t)ns + s]);
  double a, z;
  az_za_uvbeam(c.tx, c.ty, &a, &z);
  size_t o = (size_t)t * ns + s;
  az[o] = a;
  za[o] = z;
  above[o] = c.tz > 0.0 ? 1 : 0;
}

// ---------------------------------------------------------------- voltages

// One thread per (source, time, frequency); loops over beams then antennas.  Stores are
// 16 B per lane, contiguous over sources (1 KiB per wave instruction).
__global__ void __launch_bounds__(256)
voltages_kernel(const double* __restrict__ antpos, int nant, const double* __restrict__ freqs,
                const double* __restrict__ eq2tops, int nt, const double* __restrict__ crd_eq,
                int ns, const double* __restrict__ fluxes, int nf, int beam_kind,
                const double* __restrict__ beam_params, int nbeam,
                const int32_t* __restrict__ beam_idx, const double* __restrict__ beam_vals,
                int no_sqrt, f64x2* __restrict__ V, int32_t* __restrict__ bad_flag) {
  extern __shared__ double s_beam[];  // [nbeam][blockDim.x] sqrt-power per beam for this thread
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  int t = blockIdx.y;
  int f = blockIdx.z;
  if (s >= ns) return;
  double freq = freqs[f];
  Top c = rotate(eq2tops + 9 * (size_t)t, crd_eq[s], crd_eq[ns + s], crd_eq[2 * (size_t)ns + s]);
  bool up = c.tz > 0.0;
  double az, za;
  az_za_uvbeam(c.tx, c.ty, &az, &za);
  (void)az;

  bool bad = false;
  for (int b = 0; b < nbeam; ++b) {
    const double* prm = beam_params + HYDRA_BEAM_NPARAMS * b;
    double power;
    if (beam_kind == HYDRA_BEAM_GAUSS_DIAMETER) {
      double wavelength = kC / freq;
      double sigma = asin(2.2150894 * wavelength / (kPi * prm[0])) * 2 / 2.355;
      power = exp(-(za * za) / (sigma * sigma));
    } else if (beam_kind == HYDRA_BEAM_GAUSS_SIGMA) {
      double sigma = prm[0];
      if (prm[1] > 0.0) sigma = sigma * pow(freq / prm[1], prm[2]);
      power = exp(-(za * za) / (sigma * sigma));
    } else if (beam_kind == HYDRA_BEAM_UNIFORM) {
      power = 1.0;
    } else if (beam_kind == HYDRA_BEAM_POLY) {
      // E = T(x) / T(-1), x = 2 sin(za / (nu/nu_ref)^index) - 1, T a Chebyshev series summed by
      // Clenshaw's recurrence (numpy.polynomial.chebyshev.chebval); power = E^2
      int nco = (int)prm[2];
      const double* cf = prm + 3;
      double fscale = pow(freq / prm[0], prm[1]);
      double x = 2.0 * sin(za / fscale) - 1.0;
      double ev[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        double xx = w ? -1.0 : x;
        double v;
        if (nco == 1) {
          v = cf[0];
        } else {
          double c0 = cf[nco - 2], c1 = cf[nco - 1], x2 = 2.0 * xx;
          for (int i = 3; i <= nco; ++i) {
            double tmp = c0;
            c0 = cf[nco - i] - c1;
            c1 = tmp + c1 * x2;
          }
          v = c0 + c1 * xx;
        }
        ev[w] = v;
      }
      double e = ev[0] / ev[1];
      power = e * e;
    } else {
      power = beam_vals[(((size_t)b * nf + f) * nt + t) * ns + s];
    }
    double a = no_sqrt ? power : sqrt(power);
    if (isnan(a) || isinf(a)) bad = true;
    s_beam[b * blockDim.x + threadIdx.x] = a;
  }
  if (bad) atomicOr(bad_flag, 1);

  double isqrt = sqrt(0.5 * fluxes[(size_t)s * nf + f]);
  double ang_freq = 2.0 * kPi * freq;
  f64x2* out = V + ((size_t)f * nt + t) * nant * (size_t)ns + s;
  for (int a = 0; a < nant; ++a) {
    double tau = antpos[3 * a] * c.tx + antpos[3 * a + 1] * c.ty + antpos[3 * a + 2] * c.tz;
    tau /= kC;
    double sn, cs;
    sincos(ang_freq * tau, &sn, &cs);
    double amp = up ? isqrt : 0.0;
    double bm = s_beam[beam_idx[a] * blockDim.x + threadIdx.x];
    f64x2 v;
    v.x = bm * (cs * amp);
    v.y = bm * (sn * amp);
    out[(size_t)a * ns] = v;
  }
}


// ---------------------------------------------------------------- gridded (spline) beams

// Cubic B-spline basis at x for the knot span l (t[l] <= x < t[l+1]): FITPACK fpbspl, k = 3.
__device__ __forceinline__ void bspl3(const double* __restrict__ t, int l, double x, double h[4]) {
  double hh[3];
  h[0] = 1.0;
#pragma unroll
  for (int j = 1; j <= 3; ++j) {
#pragma unroll
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
#pragma unroll
    for (int i = 1; i <= j; ++i) {
      double tli = t[l + i], tlj = t[l + i - j];
      double f = hh[i - 1] / (tli - tlj);
      h[i - 1] += f * (tli - x);
      h[i] = f * (x - tlj);
    }
  }
}
// span index as in FITPACK bispeu: largest l in [3, n-5] with t[l] <= x (x clamped to the domain)
__device__ __forceinline__ int bspl_span(const double* __restrict__ t, int n, double* x) {
  double tb = t[3], te = t[n - 4];
  if (*x < tb) *x = tb;
  if (*x > te) *x = te;
  int lo = 3, hi = n - 5;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (t[mid] <= *x) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// out[f, p] = sum_ab Bx_a(x_p) By_b(y_p) coef[f][(lx-3+a) * (ny-4) + (ly-3+b)]
__global__ void __launch_bounds__(256)
bspline2d_kernel(const double* __restrict__ tx, int nx, const double* __restrict__ ty, int ny,
                 const double* __restrict__ coef, int nf, const double* __restrict__ x,
                 const double* __restrict__ y, size_t npts, double* __restrict__ out) {
  size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= npts) return;
  double xv = x[p], yv = y[p];
  int lx = bspl_span(tx, nx, &xv), ly = bspl_span(ty, ny, &yv);
  double hx[4], hy[4];
  bspl3(tx, lx, xv, hx);
  bspl3(ty, ly, yv, hy);
  int ncy = ny - 4;
  size_t ncoef = (size_t)(nx - 4) * ncy;
  for (int f = 0; f < nf; ++f) {
    const double* c = coef + f * ncoef + (size_t)(lx - 3) * ncy + (ly - 3);
    double sum = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double row = 0.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) row += c[(size_t)a * ncy + b] * hy[b];
      sum += hx[a] * row;
    }
    out[f * npts + p] = sum;
  }
}

// HEALPix (RING) bilinear interpolation, Healpix_Base::get_interpol: the two iso-latitude rings
// around theta, two pixels on each, weights linear in phi along the rings and in theta between
// them; in the polar caps the four pixels of the first / last ring stand in for the missing ring.
struct HpRing {
  long long start;
  int npix;
  double theta;
  bool shifted;
};
__device__ inline HpRing hp_ring_info(int nside, int ring) {
  const long long npix = 12LL * nside * nside, ncap = 2LL * nside * (nside - 1);
  const double fact2 = 4.0 / (double)npix, fact1 = (2 * nside) * fact2;
  const int north = ring > 2 * nside ? 4 * nside - ring : ring;
  HpRing r;
  double costh;
  if (north < nside) {
    double tmp = (double)north * north * fact2;
    costh = 1.0 - tmp;
    double sinth = sqrt(fmax((1.0 - costh) * (1.0 + costh), 0.0));
    r.theta = atan2(sinth, costh);
    r.npix = 4 * north;
    r.shifted = true;
    r.start = 2LL * north * (north - 1);
  } else {
    costh = (2 * nside - north) * fact1;
    r.theta = acos(fmin(fmax(costh, -1.0), 1.0));
    r.npix = 4 * nside;
    r.shifted = ((north - nside) & 1) == 0;
    r.start = ncap + (long long)(north - nside) * r.npix;
  }
  if (north != ring) {
    r.theta = M_PI - r.theta;
    r.start = npix - r.start - r.npix;
  }
  return r;
}

// out[f, p] = (1 - wf) sum_k wgt_k map[f0][pix_k] + wf sum_k wgt_k map[f1][pix_k]
__global__ void __launch_bounds__(256)
beam_healpix_kernel(const double* __restrict__ maps, int nside, const int32_t* __restrict__ f0,
                    const int32_t* __restrict__ f1, const double* __restrict__ wf, int nf,
                    const double* __restrict__ za, const double* __restrict__ az, size_t npts,
                    double* __restrict__ out) {
  size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= npts) return;
  const long long npix = 12LL * nside * nside;
  const double theta = za[p];
  double phi = fmod(az[p], 2.0 * M_PI);
  if (phi < 0.0) phi += 2.0 * M_PI;
  const double z = cos(theta), azz = fabs(z);
  int ir1;
  if (azz <= 2.0 / 3.0) {
    ir1 = (int)(nside * (2.0 - 1.5 * z));
  } else {
    int ic = (int)(nside * sqrt(3.0 * (1.0 - azz)));
    ir1 = z > 0 ? ic : 4 * nside - ic - 1;
  }
  const int ir2 = ir1 + 1;
  long long pix[4] = {0, 0, 0, 0};
  double wgt[4] = {0.0, 0.0, 0.0, 0.0}, th[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ir = k == 0 ? ir1 : ir2;
    const bool ok = ir >= 1 && ir <= 4 * nside - 1;
    HpRing r = hp_ring_info(nside, min(max(ir, 1), 4 * nside - 1));
    const double dphi = 2.0 * M_PI / r.npix, sh = r.shifted ? 0.5 : 0.0;
    const double tmp = phi / dphi - sh;
    long long i1 = tmp < 0 ? (long long)tmp - 1 : (long long)tmp;
    const double w1 = (phi - (i1 + sh) * dphi) / dphi;
    long long i2 = i1 + 1;
    if (i1 < 0) i1 += r.npix;
    if (i2 >= r.npix) i2 -= r.npix;
    if (ok) {
      pix[2 * k] = r.start + i1;
      pix[2 * k + 1] = r.start + i2;
      wgt[2 * k] = 1.0 - w1;
      wgt[2 * k + 1] = w1;
    }
    th[k] = r.theta;
  }
  if (ir1 == 0) {
    const double wn = theta / th[1], fac = (1.0 - wn) * 0.25;
    wgt[2] = wgt[2] * wn + fac;
    wgt[3] = wgt[3] * wn + fac;
    wgt[0] = wgt[1] = fac;
    pix[0] = (pix[2] + 2) & 3;
    pix[1] = (pix[3] + 2) & 3;
  } else if (ir2 == 4 * nside) {
    const double ws = (theta - th[0]) / (M_PI - th[0]), fac = ws * 0.25;
    wgt[0] = wgt[0] * (1.0 - ws) + fac;
    wgt[1] = wgt[1] * (1.0 - ws) + fac;
    wgt[2] = wgt[3] = fac;
    pix[2] = ((pix[0] + 2) & 3) + npix - 4;
    pix[3] = ((pix[1] + 2) & 3) + npix - 4;
  } else {
    const double wt = (theta - th[0]) / (th[1] - th[0]);
    wgt[0] *= 1.0 - wt;
    wgt[1] *= 1.0 - wt;
    wgt[2] *= wt;
    wgt[3] *= wt;
  }
  for (int f = 0; f < nf; ++f) {
    const double* m0 = maps + (size_t)f0[f] * npix;
    const double* m1 = maps + (size_t)f1[f] * npix;
    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v0 += wgt[k] * m0[pix[k]];
      v1 += wgt[k] * m1[pix[k]];
    }
    out[f * npts + p] = (1.0 - wf[f]) * v0 + wf[f] * v1;
  }
}

// ---------------------------------------------------------------- baseline products

// R[b,f,t,s] = conj(V[f,t,i_b,s]) V[f,t,j_b,s].  A workgroup stages the voltages of all
// antennas for 64 sources of one (f,t) in LDS (Nant*1 KiB), then its 4 waves stream
// baselines: each output element costs two ds_read_b128 and one 16-B global store.
template <int MODE>  // 0: baseline list -> (Nbl,Nf,Nt,Ns); 1: full cube upper triangle
__global__ void __launch_bounds__(256)
response_kernel(const f64x2* __restrict__ V, int nf, int nt, int nant, int ns,
                const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                f64x2* __restrict__ R) {
  extern __shared__ f64x2 s_v[];  // [nant][64]
  int lane = threadIdx.x & 63;
  int wave = threadIdx.x >> 6;
  int s0 = blockIdx.x * 64;
  int t = blockIdx.y, f = blockIdx.z;
  int s = s0 + lane;
  bool ok = s < ns;
  const f64x2* vin = V + ((size_t)f * nt + t) * nant * (size_t)ns;
  for (int a = wave; a < nant; a += 4) {
    f64x2 v = {0.0, 0.0};
    if (ok) v = vin[(size_t)a * ns + s];
    s_v[a * 64 + lane] = v;
  }
  __syncthreads();
  if (!ok) return;
  if (MODE == 0) {
    size_t stride_b = (size_t)nf * nt * ns;
    f64x2* out = R + ((size_t)f * nt + t) * ns + s;
    for (int b = wave; b < nbl; b += 4) {
      f64x2 r = cmul_conj(s_v[bl_i[b] * 64 + lane], s_v[bl_j[b] * 64 + lane]);
      out[(size_t)b * stride_b] = r;
    }
  } else {
    f64x2* out = R + ((size_t)f * nt + t) * nant * nant * (size_t)ns + s;
    int npair = nant * nant;
    for (int p = wave; p < npair; p += 4) {
      int i = p / nant, j = p - i * nant;
      f64x2 r = {0.0, 0.0};
      if (j >= i) r = cmul_conj(s_v[i * 64 + lane], s_v[j * 64 + lane]);
      out[(size_t)p * ns] = r;
    }
  }
}

__global__ void __launch_bounds__(256)
subarr_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns, int sub,
              f64x2* __restrict__ out) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  for (size_t ft = blockIdx.y; ft < (size_t)nft; ft += gridDim.y) {   // grid.y is capped at 65535
    const f64x2* vin = V + ft * nant * (size_t)ns + s;
    f64x2 vs = vin[(size_t)sub * ns];
    f64x2* o = out + ft * nant * (size_t)ns + s;
    for (int a = 0; a < nant; ++a) o[(size_t)a * ns] = cmul_conj(vin[(size_t)a * ns], vs);
  }
}

// vis[f,t,i,j] = sum_s conj(V_i) V_j for j >= i: one wave per (i,j) pair, lanes stride over
// sources, wave-level shuffle reduction (fixed order -> bitwise reproducible).
__global__ void __launch_bounds__(256)
vis_sum_kernel(const f64x2* __restrict__ V, int nft, int nant, int ns, f64x2* __restrict__ vis) {
  int lane = threadIdx.x & 63;
  int wave = threadIdx.x >> 6;
  int p = blockIdx.x * 4 + wave;
  if (p >= nant * nant) return;
  int i = p / nant, j = p - i * nant;
  for (size_t ft = blockIdx.y; ft < (size_t)nft; ft += gridDim.y) {   // grid.y is capped at 65535
    f64x2 acc = {0.0, 0.0};
    if (j >= i) {
      const f64x2* vi = V + (ft * nant + i) * (size_t)ns;
      const f64x2* vj = V + (ft * nant + j) * (size_t)ns;
      for (int s = lane; s < ns; s += 64) {
        f64x2 r = cmul_conj(vi[s], vj[s]);
        acc.x += r.x;
        acc.y += r.y;
      }
      acc.x = wave_sum(acc.x);
      acc.y = wave_sum(acc.y);
    }
    if (lane == 0) vis[ft * nant * nant + p] = acc;
  }
}

// ---------------------------------------------------------------- gains / row factors

__global__ void __launch_bounds__(256)
apply_gains_kernel(f64x2* __restrict__ v, const f64x2* __restrict__ gains,
                   const f64x2* __restrict__ pert, const int32_t* __restrict__ bl_i,
                   const int32_t* __restrict__ bl_j, int nbl, int nft) {
  for (int b = blockIdx.y; b < nbl; b += gridDim.y) {   // grid.y is capped at 65535
  int i = bl_i[b], j = bl_j[b];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nft; k += gridDim.x * blockDim.x) {
    f64x2 gi = gains[(size_t)i * nft + k];
    f64x2 gj = gains[(size_t)j * nft + k];
    gj.y = -gj.y;
    f64x2 fac = cmul(gi, gj);
    if (pert) {
      f64x2 xi = pert[(size_t)i * nft + k];
      f64x2 xj = pert[(size_t)j * nft + k];
      f64x2 one = {1.0 + xi.x + xj.x, xi.y - xj.y};
      fac = cmul(fac, one);
    }
    size_t o = (size_t)b * nft + k;
    v[o] = cmul(v[o], fac);
  }
  }
}

__global__ void __launch_bounds__(256)
row_factors_kernel(const f64x2* __restrict__ gains, const double* __restrict__ w,
                   const int32_t* __restrict__ bl_i, const int32_t* __restrict__ bl_j, int nbl,
                   int nft, f64x2* __restrict__ gg, f64x2* __restrict__ rows) {
  for (int b = blockIdx.y; b < nbl; b += gridDim.y) {   // grid.y is capped at 65535
  int i = bl_i[b], j = bl_j[b];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nft; k += gridDim.x * blockDim.x) {
    f64x2 fac = {1.0, 0.0};
    if (gains) {
      f64x2 gi = gains[(size_t)i * nft + k];
      f64x2 gj = gains[(size_t)j * nft + k];
      gj.y = -gj.y;
      fac = cmul(gi, gj);
    }
    size_t o = (size_t)b * nft + k;
    if (gg) gg[o] = fac;
    if (rows) {
      double rt = w ? sqrt(w[o]) : 1.0;
      f64x2 r = {fac.x * rt, fac.y * rt};
      rows[o] = r;
    }
  }
  }
}

}  // namespace hydra

using namespace hydra;

extern "C" {

const char* hydra_last_error(void) { return err_buf(); }
int hydra_abi_version(void) { return 1; }

int hydra_crd_eq(const double* ra, const double* dec, int ns, double* crd_eq, void* stream) {
  HYDRA_CHECK_ARG(ra && dec && crd_eq && ns > 0, "hydra_crd_eq: bad arguments (ns=%d)", ns);
  crd_eq_kernel<<<(ns + 255) / 256, 256, 0, as_stream(stream)>>>(ra, dec, ns, crd_eq);
  HYDRA_CHECK_LAUNCH("hydra_crd_eq");
  return HYDRA_OK;
}

int hydra_eq2tops(const double* lsts, int nt, double latitude, double* eq2tops, void* stream) {
  HYDRA_CHECK_ARG(lsts && eq2tops && nt > 0, "hydra_eq2tops: bad arguments (nt=%d)", nt);
  eq2tops_kernel<<<(nt + 255) / 256, 256, 0, as_stream(stream)>>>(lsts, nt, latitude, eq2tops);
  HYDRA_CHECK_LAUNCH("hydra_eq2tops");
  return HYDRA_OK;
}

int hydra_az_za(const double* eq2tops, int nt, const double* crd_eq, int ns, double* az,
                double* za, uint8_t* above, void* stream) {
  HYDRA_CHECK_ARG(eq2tops && crd_eq && az && za && above && nt > 0 && ns > 0 && nt <= 65535,
                  "hydra_az_za: bad arguments (nt=%d ns=%d)", nt, ns);
  dim3 grid((ns + 255) / 256, nt);
  az_za_kernel<<<grid, 256, 0, as_stream(stream)>>>(eq2tops, nt, crd_eq, ns, az, za, above);
  HYDRA_CHECK_LAUNCH("hydra_az_za");
  return HYDRA_OK;
}

int hydra_voltages(const double* antpos, int nant, const double* freqs_hz, int nf,
                   const double* eq2tops, int nt, const double* crd_eq, int ns,
                   const double* fluxes, int beam_kind, const double* beam_params, int nbeam,
                   const int32_t* beam_idx, const double* beam_vals, int force_no_beam_sqrt,
                   double* V, int32_t* bad_flag, void* stream) {
  HYDRA_CHECK_ARG(antpos && freqs_hz && eq2tops && crd_eq && fluxes && beam_idx && V && bad_flag,
                  "hydra_voltages: null pointer argument");
  HYDRA_CHECK_ARG(nant > 0 && nf > 0 && nt > 0 && ns > 0 && nbeam > 0,
                  "hydra_voltages: sizes must be positive (nant=%d nf=%d nt=%d ns=%d nbeam=%d)",
                  nant, nf, nt, ns, nbeam);
  HYDRA_CHECK_ARG(nt <= 65535 && nf <= 65535, "hydra_voltages: nt, nf must be <= 65535");
  HYDRA_CHECK_ARG(beam_kind >= 0 && beam_kind <= HYDRA_BEAM_POLY,
                  "hydra_voltages: unknown beam_kind %d", beam_kind);
  HYDRA_CHECK_ARG(beam_kind != HYDRA_BEAM_VALUES || beam_vals,
                  "hydra_voltages: beam_vals required for HYDRA_BEAM_VALUES");
  HYDRA_CHECK_ARG(beam_kind == HYDRA_BEAM_VALUES || beam_kind == HYDRA_BEAM_UNIFORM || beam_params,
                  "hydra_voltages: beam_params required for analytic beams");
  int block = 256;
  size_t lds = (size_t)nbeam * block * sizeof(double);
  while (lds > 64 * 1024 && block > 64) {
    block >>= 1;
    lds = (size_t)nbeam * block * sizeof(double);
  }
  HYDRA_CHECK_ARG(lds <= 64 * 1024, "hydra_voltages: too many distinct beams (%d)", nbeam);
  dim3 grid((ns + block - 1) / block, nt, nf);
  voltages_kernel<<<grid, block, lds, as_stream(stream)>>>(
      antpos, nant, freqs_hz, eq2tops, nt, crd_eq, ns, fluxes, nf, beam_kind, beam_params, nbeam,
      beam_idx, beam_vals, force_no_beam_sqrt, reinterpret_cast<f64x2*>(V), bad_flag);
  HYDRA_CHECK_LAUNCH("hydra_voltages");
  return HYDRA_OK;
}

static int check_v(const char* name, const double* V, int nf, int nt, int nant, int ns) {
  HYDRA_CHECK_ARG(V && nf > 0 && nt > 0 && nant > 0 && ns > 0 && nt <= 65535 && nf <= 65535,
                  "%s: bad voltage cube arguments (nf=%d nt=%d nant=%d ns=%d)", name, nf, nt, nant,
                  ns);
  HYDRA_CHECK_ARG((size_t)nant * 64 * 16 <= 160 * 1024,
                  "%s: nant=%d exceeds the LDS staging limit (160)", name, nant);
  return HYDRA_OK;
}

int hydra_response(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                   const int32_t* bl_j, int nbl, double* R, void* stream) {
  int rc = check_v("hydra_response", V, nf, nt, nant, ns);
  if (rc) return rc;
  HYDRA_CHECK_ARG(bl_i && bl_j && R && nbl > 0, "hydra_response: bad baseline arguments");
  dim3 grid((ns + 63) / 64, nt, nf);
  size_t lds = (size_t)nant * 64 * 16;
  if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&response_kernel<0>), lds)) return rc_lds;
  response_kernel<0><<<grid, 256, lds, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(V), nf, nt, nant, ns, bl_i, bl_j, nbl,
      reinterpret_cast<f64x2*>(R));
  HYDRA_CHECK_LAUNCH("hydra_response");
  return HYDRA_OK;
}

int hydra_vis_cube(const double* V, int nf, int nt, int nant, int ns, double* cube, void* stream) {
  int rc = check_v("hydra_vis_cube", V, nf, nt, nant, ns);
  if (rc) return rc;
  HYDRA_CHECK_ARG(cube, "hydra_vis_cube: null output");
  dim3 grid((ns + 63) / 64, nt, nf);
  size_t lds = (size_t)nant * 64 * 16;
  if (int rc_lds = reserve_lds(reinterpret_cast<const void*>(&response_kernel<1>), lds)) return rc_lds;
  response_kernel<1><<<grid, 256, lds, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(V), nf, nt, nant, ns, nullptr, nullptr, 0,
      reinterpret_cast<f64x2*>(cube));
  HYDRA_CHECK_LAUNCH("hydra_vis_cube");
  return HYDRA_OK;
}

int hydra_vis_subarr(const double* V, int nf, int nt, int nant, int ns, int subarr_ant,
                     double* out, void* stream) {
  HYDRA_CHECK_ARG(V && out && nf > 0 && nt > 0 && nant > 0 && ns > 0,
                  "hydra_vis_subarr: bad arguments");
  HYDRA_CHECK_ARG(subarr_ant >= 0 && subarr_ant < nant, "hydra_vis_subarr: subarr_ant=%d out of range",
                  subarr_ant);
  HYDRA_CHECK_ARG((long long)nf * nt <= 2147483647LL, "hydra_vis_subarr: nf*nt too large");
  dim3 grid((ns + 255) / 256, std::min(nf * nt, 65535));
  subarr_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const f64x2*>(V), nf * nt, nant,
                                                     ns, subarr_ant, reinterpret_cast<f64x2*>(out));
  HYDRA_CHECK_LAUNCH("hydra_vis_subarr");
  return HYDRA_OK;
}

int hydra_vis_sum(const double* V, int nf, int nt, int nant, int ns, double* vis, void* stream) {
  HYDRA_CHECK_ARG(V && vis && nf > 0 && nt > 0 && nant > 0 && ns > 0, "hydra_vis_sum: bad arguments");
  HYDRA_CHECK_ARG((long long)nf * nt <= 2147483647LL, "hydra_vis_sum: nf*nt too large");
  dim3 grid((nant * nant + 3) / 4, std::min(nf * nt, 65535));
  vis_sum_kernel<<<grid, 256, 0, as_stream(stream)>>>(reinterpret_cast<const f64x2*>(V), nf * nt, nant,
                                                      ns, reinterpret_cast<f64x2*>(vis));
  HYDRA_CHECK_LAUNCH("hydra_vis_sum");
  return HYDRA_OK;
}

int hydra_apply_gains(double* v, const double* gains, const double* pert, const int32_t* bl_i,
                      const int32_t* bl_j, int nbl, int nf, int nt, void* stream) {
  HYDRA_CHECK_ARG(v && gains && bl_i && bl_j && nbl > 0 && nf > 0 && nt > 0,
                  "hydra_apply_gains: bad arguments");
  int nft = nf * nt;
  dim3 grid(std::min((nft + 255) / 256, 1024), std::min(nbl, 65535));
  apply_gains_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<f64x2*>(v), reinterpret_cast<const f64x2*>(gains),
      reinterpret_cast<const f64x2*>(pert), bl_i, bl_j, nbl, nft);
  HYDRA_CHECK_LAUNCH("hydra_apply_gains");
  return HYDRA_OK;
}

int hydra_row_factors(const double* gains, const double* inv_noise, const int32_t* bl_i,
                      const int32_t* bl_j, int nbl, int nf, int nt, double* gg, double* rows,
                      void* stream) {
  HYDRA_CHECK_ARG(bl_i && bl_j && nbl > 0 && nf > 0 && nt > 0 && (gg || rows),
                  "hydra_row_factors: bad arguments");
  int nft = nf * nt;
  dim3 grid(std::min((nft + 255) / 256, 1024), std::min(nbl, 65535));
  row_factors_kernel<<<grid, 256, 0, as_stream(stream)>>>(
      reinterpret_cast<const f64x2*>(gains), inv_noise, bl_i, bl_j, nbl, nft,
      reinterpret_cast<f64x2*>(gg), reinterpret_cast<f64x2*>(rows));
  HYDRA_CHECK_LAUNCH("hydra_row_factors");
  return HYDRA_OK;
}


int hydra_beam_bspline(const double* tx, int nx, const double* ty, int ny, const double* coef,
                       int nf, const double* x, const double* y, long long npts, double* out,
                       void* stream) {
  HYDRA_CHECK_ARG(tx && ty && coef && x && y && out, "hydra_beam_bspline: null pointer argument");
  HYDRA_CHECK_ARG(nx >= 8 && ny >= 8 && nf > 0 && npts > 0,
                  "hydra_beam_bspline: need nx, ny >= 8 knots (cubic), nf, npts > 0");
  unsigned grid = (unsigned)((npts + 255) / 256);
  bspline2d_kernel<<<grid, 256, 0, as_stream(stream)>>>(tx, nx, ty, ny, coef, nf, x, y,
                                                        (size_t)npts, out);
  HYDRA_CHECK_LAUNCH("hydra_beam_bspline");
  return HYDRA_OK;
}

int hydra_beam_healpix(const double* maps, int nmaps, int nside, const int32_t* f0,
                       const int32_t* f1, const double* wf, int nf, const double* za,
                       const double* az, long long npts, double* out, void* stream) {
  HYDRA_CHECK_ARG(maps && f0 && f1 && wf && za && az && out,
                  "hydra_beam_healpix: null pointer argument");
  HYDRA_CHECK_ARG(nmaps > 0 && nside > 0 && nside <= 8192 && nf > 0 && npts > 0,
                  "hydra_beam_healpix: bad sizes (nmaps=%d nside=%d nf=%d)", nmaps, nside, nf);
  unsigned grid = (unsigned)((npts + 255) / 256);
  beam_healpix_kernel<<<grid, 256, 0, as_stream(stream)>>>(maps, nside, f0, f1, wf, nf, za, az,
                                                           (size_t)npts, out);
  HYDRA_CHECK_LAUNCH("hydra_beam_healpix");
  return HYDRA_OK;
}

}  // extern "C"
