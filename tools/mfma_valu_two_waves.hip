// Microbenchmark: what a second wave on the same SIMD costs an FP64 MFMA stream on gfx950.
// One 512-thread workgroup per CU; waves 0-3 run back-to-back v_mfma_f64_16x16x4_f64, waves 4-7
// (one per SIMD beside them) run nothing / FP64 FMAs / integer VALU / LDS reads / MFMAs.
// Then the Gram kernel's own shape: both waves alternate a burst of 32 MFMAs with a batch of
// NV FP64 VALU ops, with NV = 0, 20, 40.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_two_waves.hip -o /tmp/two && /tmp/two
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_cyc[8];

template <int MODE>
__global__ void __launch_bounds__(512) k_pair(double* out, int iters, double a0, double b0) {
  __shared__ f64x4 lds[1024];
  lds[threadIdx.x] = f64x4{a0, b0, a0, b0};
  lds[threadIdx.x + 512] = f64x4{b0, a0, b0, a0};
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double s = 0;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  if (wave < 4 || MODE == 4) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (MODE == 1) {
    double v[16], a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_fma(v[j], a, b);
    }
    for (int i = 0; i < 16; ++i) s += v[i];
  } else if (MODE == 2) {
    unsigned u[16];
    for (int i = 0; i < 16; ++i) u[i] = threadIdx.x + i;
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) u[j] = u[j] * 1664525u + 1013904223u;
    }
    for (int i = 0; i < 16; ++i) s += u[i];
  } else if (MODE == 3) {
    f64x4 l = {0, 0, 0, 0};
    int idx = threadIdx.x;
    for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) l += lds[(idx + 64 * j) & 1023];
    }
    s = l[0] + l[1] + l[2] + l[3];
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 17 && (threadIdx.x & 63) == 0) g_cyc[wave] = c1 - c0;
}

// the Gram kernel's shape: burst of 32 MFMAs, then NV FP64 VALU ops, on every wave
template <int NV, int NWAVES>
__global__ void __launch_bounds__(64 * NWAVES) k_burst(double* out, int iters, double a0, double b0) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f64x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0;
  double v[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = a + i;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = __builtin_fma(v[j], a, b);
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 17 && (threadIdx.x & 63) == 0) g_cyc[wave] = c1 - c0;
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 512 * 256);
  const int iters = 4000;
  unsigned long long c[8];
  const char* names[5] = {"nothing", "FP64 FMA stream", "integer VALU stream", "LDS b128 reads", "MFMA stream"};
#define RUN(M)                                                                                   \
  k_pair<M><<<256, 512>>>(out, iters, 1.234, 0.371);                                             \
  hipDeviceSynchronize();                                                                        \
  hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cyc), sizeof(c));                                          \
  printf("beside %-20s: MFMA wave %.1f cycles per MFMA; other wave %.0f cycles in all (%.2f per op)\n", names[M], \
         (double)c[0] / (iters * 16.0), (double)c[4], (double)c[4] / (iters * 8 * 16.0));
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
#define RUNB(NV, NW)                                                                             \
  k_burst<NV, NW><<<256, 64 * NW>>>(out, iters, 1.234, 0.371);                                   \
  hipDeviceSynchronize();                                                                        \
  hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cyc), sizeof(c));                                          \
  printf("burst 32 MFMA + %2d FMA, %d waves per SIMD: wave 0 %.0f cycles per step, last wave %.0f; per SIMD %.0f cycles per burst\n", \
         NV, NW / 4, (double)c[0] / iters, (double)c[NW - 1] / iters,                            \
         (double)(c[0] > c[NW - 1] ? c[0] : c[NW - 1]) / iters / (NW / 4));
  RUNB(0, 4) RUNB(20, 4) RUNB(40, 4) RUNB(0, 8) RUNB(20, 8) RUNB(40, 8)
  return 0;
}
