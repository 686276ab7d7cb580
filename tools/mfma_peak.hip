// Microbenchmark: FP64 MFMA (v_mfma_f64_16x16x4_f64) and FP64 VALU FMA issue rates on gfx950,
// alone and interleaved, to pin the real FP64 roofline used in DESIGN.md / bench.py.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_cycles[2];  // [0] shader cycles, [1] 100 MHz ticks (block 0, wave 0)

template <int NACC, int NFMA>
__global__ void __launch_bounds__(256) k_mix(double* out, int iters, double a0, double b0) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  double v[NFMA > 0 ? NFMA : 1];
  for (int i = 0; i < (NFMA > 0 ? NFMA : 1); ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NFMA; ++j) v[j] = __builtin_fma(v[j], a, b);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NFMA > 0 ? NFMA : 1); ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 17 && threadIdx.x == 0) { g_cycles[0] = c1 - c0; g_cycles[1] = r1 - r0; }
}

// 4x4 register tile like the Gram kernel: acc[m][n] += a[m] * b[n] with distinct operands
__global__ void __launch_bounds__(256) k_tile(double* out, int iters, double a0, double b0) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f64x4 acc[4][4];
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = a0 + threadIdx.x * 1e-3 + i;
    b[i] = b0 * (1 + i) + threadIdx.x * 1e-5;
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0, 0, 0, 0};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 17 && threadIdx.x == 0) { g_cycles[0] = c1 - c0; g_cycles[1] = r1 - r0; }
}

// 2x4 register tile (8 accumulators -> 2 blocks per CU), operands constant or per-lane random
__global__ void __launch_bounds__(256, 2) k_tile2(double* out, int iters, double a0, double b0, int rnd) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f64x4 acc[2][4];
  double a[2], b[4];
  for (int i = 0; i < 4; ++i) {
    b[i] = rnd ? b0 * (1 + i) + threadIdx.x * 1.3e-5 : b0;
    for (int j = 0; j < 2; ++j) acc[j][i] = f64x4{0, 0, 0, 0};
  }
  for (int i = 0; i < 2; ++i) a[i] = rnd ? a0 + threadIdx.x * 1.7e-3 + i : a0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 17 && threadIdx.x == 0) { g_cycles[0] = c1 - c0; g_cycles[1] = r1 - r0; }
}

// 2x4 VGPR-accumulator tile with NF fp64 VALU fmas and NL 16-byte LDS reads issued after each MFMA:
// does the wave's own VALU / LDS work hide in the shadow of the 64-cycle matrix op?
template <int NF, int NL>
__global__ void __launch_bounds__(256, 2) k_fill(double* out, int iters, double a0, double b0) {
  __shared__ f64x4 lds[512];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f64x4 acc[2][4];
  double a[2], b[4];
  lds[threadIdx.x] = f64x4{a0, b0, a0, b0};
  lds[threadIdx.x + 256] = f64x4{b0, a0, b0, a0};
  __syncthreads();
  for (int i = 0; i < 4; ++i) {
    b[i] = b0 * (1 + i) + threadIdx.x * 1.3e-5;
    for (int j = 0; j < 2; ++j) acc[j][i] = f64x4{0, 0, 0, 0};
  }
  for (int i = 0; i < 2; ++i) a[i] = a0 + threadIdx.x * 1.7e-3 + i;
  double v[NF > 0 ? NF : 1];
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) v[i] = a0 + i;
  f64x4 l = {0, 0, 0, 0};
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NF; ++j) v[j] = __builtin_fma(v[j], a[0], b[0]);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          f64x4 t = lds[(idx + 17 * (m * 4 + n) + j * 64) & 511];
          l += t;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  double s = l[0] + l[1] + l[2] + l[3];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 17 && threadIdx.x == 0) { g_cycles[0] = c1 - c0; g_cycles[1] = r1 - r0; }
}

// NACC independent accumulators, dependent distance NACC, optional integer VALU fillers
template <int NACC, int NINT>
__global__ void __launch_bounds__(256) k_dist(double* out, int iters, double a0, double b0) {
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0;
  unsigned u[NINT > 0 ? NINT : 1];
  for (int i = 0; i < (NINT > 0 ? NINT : 1); ++i) u[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NINT; ++j) u[j] = u[j] * 1664525u + 1013904223u;
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NINT > 0 ? NINT : 1); ++i) s += u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 17 && threadIdx.x == 0) { g_cycles[0] = c1 - c0; g_cycles[1] = r1 - r0; }
}

template <int NFMA>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0) {
  double v[NFMA];
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int i = 0; i < NFMA; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < NFMA; ++j) v[j] = __builtin_fma(v[j], a, b);
  double s = 0;
  for (int i = 0; i < NFMA; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
double time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main(int argc, char** argv) {
  double* out; hipMalloc(&out, sizeof(double) * 256 * 256 * 16);
  const int iters = 20000;
  if (argc > 1) {
    // sustained clock: the same pure-MFMA kernel for ~argv[1] x 17 ms in one launch (the short
    // sweeps below end before the power management settles)
    int mult = atoi(argv[1]);
    for (int rep = 0; rep < 3; ++rep) {
      int grid = 256 * 2;
      double ms = time_ms([&] { k_tile2<<<grid, 256>>>(out, iters * 2 * mult, 1.2345678901234567, 0.3712345678901234, 1); });
      double fl = (double)grid * 4 * iters * 16 * 2048.0 * mult;
      unsigned long long cyc[2];
      hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
      printf("sustained (2 blocks/CU, 2x4 tile, random data) x%d: %.1f ms %.1f TFLOP/s | %.1f cyc/MFMA per wave, %.2f GHz\n",
             mult, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0 * mult), (double)cyc[0] / cyc[1] * 0.1);
    }
    return 0;
  }
  // occupancy sweep with a small accumulator set (nacc=2: <=64 VGPR -> up to 8 waves/SIMD)
  for (int wpc = 1; wpc <= 8; ++wpc) {
    int grid = 256 * wpc;
    double ms = time_ms([&] { k_dist<2, 0><<<grid, 256>>>(out, iters * 8, 1.0, 0.37); });
    double fl = (double)grid * 4 * iters * 16 * 2048.0;
    unsigned long long cyc[2];
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
    printf("occupancy sweep nacc=2 blocks/CU=%d: %.2f ms %.1f TFLOP/s | %.1f cyc/MFMA per wave, %.2f GHz\n",
           wpc, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1);
  }
  for (int z = 0; z < 3; ++z) {
    int grid = 256 * 2;
    double av = z == 0 ? 0.0 : (z == 1 ? 1.0 : 1.2345678901234567);
    double bv = z == 0 ? 0.0 : (z == 1 ? 1.0 : 0.3712345678901234);
    double ms = time_ms([&] { k_tile2<<<grid, 256>>>(out, iters * 2, av, bv, z == 2); });
    double fl = (double)grid * 4 * iters * 16 * 2048.0;
    unsigned long long cyc[2];
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
    printf("data sweep (2 blocks/CU, 2x4 tile) a=%.3f b=%.3f: %.2f ms %.1f TFLOP/s | %.1f cyc/MFMA per wave, %.2f GHz\n",
           av, bv, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1);
  }
#define RUN_FILL(NF, NL)                                                                        \
  {                                                                                             \
    int grid = 256;                                                                             \
    double ms = time_ms([&] { k_fill<NF, NL><<<grid, 256>>>(out, iters * 2, 1.2345678, 0.3712345); }); \
    double fl = (double)grid * 4 * iters * 16 * 2048.0;                                         \
    unsigned long long cyc[2];                                                                  \
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));                                \
    printf("fill sweep (1 block/CU, VGPR acc) fma/mfma=%d ldsread/mfma=%d: %.2f ms %.1f TFLOP/s | %.1f cyc/MFMA, %.2f GHz\n", \
           NF, NL, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1); \
  }
  RUN_FILL(0, 0) RUN_FILL(1, 0) RUN_FILL(2, 0) RUN_FILL(4, 0) RUN_FILL(8, 0) RUN_FILL(0, 1) RUN_FILL(0, 2) RUN_FILL(2, 1) RUN_FILL(4, 1)
  for (int wpc = 1; wpc <= 4; ++wpc) {
    int grid = 256 * wpc;
    double ms = time_ms([&] { k_dist<4, 0><<<grid, 256>>>(out, iters * 4, 1.0, 0.37); });
    double fl = (double)grid * 4 * iters * 16 * 2048.0;
    unsigned long long cyc[2];
    hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
    printf("occupancy sweep nacc=4 blocks/CU=%d: %.2f ms %.1f TFLOP/s | %.1f cyc/MFMA per wave, %.2f GHz\n",
           wpc, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1);
  }
  for (int wg_per_cu = 1; wg_per_cu <= 1; ++wg_per_cu) {
    int grid = 256 * wg_per_cu;
    {
      double ms = time_ms([&] { k_mix<16, 0><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
      double fl = (double)grid * 4 * iters * 16 * 2048.0;
      unsigned long long cyc[2];
      hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
      printf("mfma only       waves/SIMD=%d: %.2f ms  %.1f TFLOP/s  | %.1f shader cycles per MFMA, clock %.2f GHz\n",
             wg_per_cu, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1);
    }
    {
      double ms = time_ms([&] { k_tile<<<grid, 256>>>(out, iters, 1.0, 0.37); });
      double fl = (double)grid * 4 * iters * 16 * 2048.0;
      unsigned long long cyc[2];
      hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
      printf("mfma 4x4 tile   waves/SIMD=%d: %.2f ms  %.1f TFLOP/s  | %.1f shader cycles per MFMA, clock %.2f GHz\n",
             wg_per_cu, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0), (double)cyc[0] / cyc[1] * 0.1);
    }
#define RUN_DIST(NACC, NINT)                                                                    \
    {                                                                                           \
      double ms = time_ms([&] { k_dist<NACC, NINT><<<grid, 256>>>(out, iters * 16 / NACC, 1.0, 0.37); }); \
      double fl = (double)grid * 4 * iters * 16 * 2048.0;                                       \
      unsigned long long cyc[2];                                                                \
      hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));                              \
      printf("mfma nacc=%2d nint=%d waves/SIMD=%d: %.2f ms %.1f TFLOP/s | %.1f cyc/MFMA, %.2f GHz\n", NACC, \
             NINT, wg_per_cu, ms, fl / ms / 1e9, (double)cyc[0] / (iters * 16.0),                \
             (double)cyc[0] / cyc[1] * 0.1);                                                    \
    }
    RUN_DIST(1, 0) RUN_DIST(2, 0) RUN_DIST(4, 0) RUN_DIST(8, 0) RUN_DIST(16, 0) RUN_DIST(4, 4) RUN_DIST(4, 16)
    {
      double ms = time_ms([&] { k_mix<16, 2><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
      double fl = (double)grid * 4 * iters * 16 * 2048.0;
      printf("mfma + 2 fma/mfma waves/SIMD=%d: %.2f ms  %.1f TFLOP/s (mfma flops only)\n", wg_per_cu, ms, fl / ms / 1e9);
    }
    {
      double ms = time_ms([&] { k_mix<16, 4><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
      double fl = (double)grid * 4 * iters * 16 * 2048.0;
      printf("mfma + 4 fma/mfma waves/SIMD=%d: %.2f ms  %.1f TFLOP/s (mfma flops only)\n", wg_per_cu, ms, fl / ms / 1e9);
    }
    {
      double ms = time_ms([&] { k_mix<16, 8><<<grid, 256>>>(out, iters, 1.0, 1e-9); });
      double fl = (double)grid * 4 * iters * 16 * 2048.0;
      printf("mfma + 8 fma/mfma waves/SIMD=%d: %.2f ms  %.1f TFLOP/s (mfma flops only)\n", wg_per_cu, ms, fl / ms / 1e9);
    }
    {
      double ms = time_ms([&] { k_fma<16><<<grid, 256>>>(out, iters * 16, 1.0, 1e-9); });
      double fl = (double)grid * 256 * (double)iters * 16 * 16 * 2.0;
      printf("valu fma only   waves/SIMD=%d: %.2f ms  %.1f TFLOP/s\n", wg_per_cu, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
