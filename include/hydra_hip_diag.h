/*
 * hydra_hip_diag.h -- measurement and test hooks of libhydra_hip.so.
 *
 * NOT part of the product boundary (include/hydra_hip.h): nothing a Hydra maintainer binds to
 * replace a reference function is declared here, and no entry point of the product ABI needs
 * these calls.  bench.py uses the first to time one kernel inside an entry point, the GPU tests
 * use the second to run the same product through both of its kernels.  Neither has a
 * counterpart in the reference.
 */
#ifndef HYDRA_HIP_DIAG_H
#define HYDRA_HIP_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Profiling hook: HIP events (hipEvent_t, NULL to clear) recorded by this thread's next calls of
 * hydra_gram_fused / hydra_gram_uniform on their stream immediately before and after the main
 * Gram kernel.  No counterpart in the reference.                                            */
int hydra_profile_events(void* start_event, void* stop_event);

/* Test hook: hydra_adjoint_V forms R^H z in antenna space on the matrix cores when Nant <= 64
 * (default) and with the pair-loop kernel otherwise; enable = 0 forces the pair loop.  Returns the
 * previous setting.  No counterpart in the reference.                                        */
int hydra_debug_adjoint_mfma(int enable);

#ifdef __cplusplus
}
#endif

#endif /* HYDRA_HIP_DIAG_H */
