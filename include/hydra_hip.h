/*
 * hydra_hip.h -- C ABI of libhydra_hip.so: the MI355X (gfx950) implementation of
 * Hydra's Gaussian-constrained-realisation hot path.
 *
 * The reference (HydraRadio/Hydra, pure Python) has no FFI: its boundary for this
 * path is the module-level Python API of hydra.vis_simulator / ptsrc_sampler /
 * sh_sampler / linear_solver.  Each entry point below names the reference
 * expression(s) it replaces (file:line relative to the reference root).  The
 * Python mirror of that API lives in hydra_amd/ and binds these symbols with
 * ctypes (hydra_amd/_lib.py); INTEGRATION.md shows the stub a Hydra maintainer
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - arrays are C-contiguous; complex128 is interleaved (re, im) doubles;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls are
 *     asynchronous on it and allocate nothing: scratch comes in through
 *     (workspace, workspace_bytes), sized by the matching *_workspace_bytes();
 *   - return value: 0 on success, <0 on error; hydra_last_error() returns a
 *     thread-local message for the last failure.
 */
#ifndef HYDRA_HIP_H
#define HYDRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYDRA_OK 0
#define HYDRA_ERR_ARG (-1)
#define HYDRA_ERR_WORKSPACE (-2)
#define HYDRA_ERR_HIP (-3)

/* beam kinds for hydra_voltages() */
#define HYDRA_BEAM_GAUSS_DIAMETER 0 /* params: [diameter_m]                             */
#define HYDRA_BEAM_GAUSS_SIGMA 1    /* params: [sigma_rad, ref_freq_hz (0=none), index]  */
#define HYDRA_BEAM_UNIFORM 2        /* params: unused                                    */
#define HYDRA_BEAM_VALUES 3         /* beam_vals (Nbeam,Nf,Nt,Ns) power values given     */
#define HYDRA_BEAM_POLY 4           /* Chebyshev polynomial beam (hera_sim PolyBeam, hydra/example.py:
                                       143-168): params [ref_freq_hz, spectral_index, ncoef, c_0 .. ] */
#define HYDRA_BEAM_NPARAMS 32       /* doubles per beam in beam_params                   */
#define HYDRA_BEAM_POLY_MAXCOEF 29  /* = HYDRA_BEAM_NPARAMS - 3                          */

const char* hydra_last_error(void);
int hydra_abi_version(void);

/* ---- geometry (matvis.coordinates pieces; vis_simulator.py:467,470,247) ---------- */

/* crd_eq (3,Ns) = (cos ra cos dec, sin ra cos dec, sin dec)          vis_simulator.py:467 */
int hydra_crd_eq(const double* ra, const double* dec, int ns, double* crd_eq, void* stream);

/* eq2tops (Nt,3,3) = eci_to_enu_matrix(lst_t, latitude)              vis_simulator.py:470 */
int hydra_eq2tops(const double* lsts, int nt, double latitude, double* eq2tops, void* stream);

/* az, za (Nt,Ns) ["uvbeam" convention] and horizon flag tz>0 (uint8) for host-evaluated
 * beams                                                              vis_simulator.py:233-247 */
int hydra_az_za(const double* eq2tops, int nt, const double* crd_eq, int ns, double* az,
                double* za, uint8_t* above, void* stream);

/* Gridded (UVBeam-style) power beam: bicubic tensor-product B-spline evaluated at npts points,
 * for nf coefficient sets that share the knots (one set per frequency channel):
 *   out[f,p] = sum_ab B_a(x_p; tx) B_b(y_p; ty) coef[f][(lx-3+a)(ny-4) + (ly-3+b)]
 * -- the evaluation half of scipy.interpolate.RectBivariateSpline.ev (FITPACK bispeu), which is
 * what pyuvdata's az_za_simple beam interpolation calls with reuse_spline=True
 * (vis_simulator.py:340-349).  x = za, y = az; arguments are clamped to the knot domain.
 * tx (nx), ty (ny) knots and coef (nf, (nx-4)(ny-4)) come from the host-side fit.          */
int hydra_beam_bspline(const double* tx, int nx, const double* ty, int ny, const double* coef,
                       int nf, const double* x, const double* y, long long npts, double* out,
                       void* stream);

/* HEALPix-tabulated power beam (RING maps, one per tabulated channel): pyuvdata's
 * "healpix_simple" interpolation -- bilinear over the four nearest pixels at (theta = za,
 * phi = az) -- with linear interpolation between channels, as the reference's SH response
 * script configures its UVBeam (scripts/precompute_response_sh_singlefreq_general.py:106-110;
 * consumed at vis_simulator.py:340-349):
 *   out[f,p] = (1 - wf[f]) I(maps[f0[f]], p) + wf[f] I(maps[f1[f]], p),  I = 4-pixel sum.
 * maps (nmaps, 12 nside^2); f0, f1 (nf) int32 map indices and wf (nf) from the host.        */
int hydra_beam_healpix(const double* maps, int nmaps, int nside, const int32_t* f0,
                       const int32_t* f1, const double* wf, int nf, const double* za,
                       const double* az, long long npts, double* out, void* stream);

/* ---- response generator ---------------------------------------------------------- */

/* Antenna voltages V (Nf,Nt,Nant,Ns) c128:
 *   V = sqrt(beam_power(az,za,nu)) * sqrt(0.5 flux) * exp(i 2 pi nu (antpos . s_top) / c),
 *   zero below the horizon.                          vis_simulator.py:214,233-276,347-359
 * fluxes (Ns,Nf); beam_idx (Nant) int32 into the Nbeam beams; beam_vals only for
 * HYDRA_BEAM_VALUES (else NULL).  *bad_flag (device int32, caller-zeroed) is set to 1 if any
 * beam value is NaN/Inf (the reference raises ValueError, vis_simulator.py:256-257).      */
int hydra_voltages(const double* antpos, int nant, const double* freqs_hz, int nf,
                   const double* eq2tops, int nt, const double* crd_eq, int ns,
                   const double* fluxes, int beam_kind, const double* beam_params, int nbeam,
                   const int32_t* beam_idx, const double* beam_vals, int force_no_beam_sqrt,
                   double* V, int32_t* bad_flag, void* stream);

/* Per-source visibilities for a list of antenna pairs:
 *   R[b,f,t,s] = conj(V[f,t,bl_i[b],s]) * V[f,t,bl_j[b],s]   (Nbl,Nf,Nt,Ns) c128
 * = vis_sim_per_source's outer product + calc_proj_operator's baseline gather
 *                                        vis_simulator.py:281-289; ptsrc_sampler.py:209-213 */
int hydra_response(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                   const int32_t* bl_j, int nbl, double* R, void* stream);

/* Full antenna-antenna cube (Nf,Nt,Nant,Nant,Ns) c128, upper triangle incl. autos, lower
 * triangle zero -- simulate_vis_per_source's return value       vis_simulator.py:281-289,515 */
int hydra_vis_cube(const double* V, int nf, int nt, int nant, int ns, double* cube, void* stream);

/* subarr_ant variant: out (Nf,Nt,Nant,Ns) = conj(V[..,a,:]) * V[..,subarr_ant,:]
 *                                                               vis_simulator.py:290-297 */
int hydra_vis_subarr(const double* V, int nf, int nt, int nant, int ns, int subarr_ant,
                     double* out, void* stream);

/* Sum over sources: vis (Nf,Nt,Nant,Nant) c128 upper triangle = sum_s conj(V_i) V_j
 * (simulate_vis)                                                vis_simulator.py:565-568 */
int hydra_vis_sum(const double* V, int nf, int nt, int nant, int ns, double* vis, void* stream);

/* Sum over the sources of each region: out (Nbl,Nf,Nt,Nreg) c128,
 *   out[b,f,t,r] = sum_{seg[r] <= s < seg[r+1]} conj(V[f,t,i_b,s]) V[f,t,j_b,s]
 * (one simulate_vis per region + baseline gather, region_sampler.py:198-223).  The sources must be
 * ordered by region; seg (Nreg+1) int32 with seg[0] = 0, seg[Nreg] = Ns is passed twice: a host
 * copy (validated before launch) and the device copy.                                         */
int hydra_vis_regions(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                      const int32_t* bl_j, int nbl, const int32_t* seg_host, const int32_t* seg,
                      int nreg, double* out, void* stream);

/* Beam-head consumer of the generator: per-source visibilities of a uniform-beam array made
 * Hermitian in the antenna pair (autos doubled) and contracted over sources with the
 * Fourier-Bessel matrices, without forming the per-source cube:
 *   out[f,t,a,A,q] = sum_s M_s[a,A] B[t,s,q],  M_s[a,A] = conj(V_a,s) V_A,s r_s with r = ref[f,t,s] for
 *   a < A, conj(ref) for a > A, and M_s[a,a] = 2|V_a,s|^2 Re ref (ref scales the upper triangle first)
 * V (Nf,Nt,Nant,Ns) c128 from hydra_voltages with a uniform beam; B (Nt,Ns,Q) c128 (Q = Nc^2 for
 * the outer product of the design matrix, Nc for a power-beam/reference contraction); ref
 * (Nf,Nt,Ns) c128 or NULL; out (Nf,Nt,Nant,Nant,Q) c128.   per_ant_beam_sampler.py:385-422
 * row >= 0: only out[f,t,A,row,q] (the column of the sampled antenna) is formed, into an
 * (Nf,Nt,Nant,Q) array -- what get_bess_to_vis needs (per_ant_beam_sampler.py:468-583).       */
int hydra_bess_sky_contraction(const double* V, int nf, int nt, int nant, int ns, const double* B,
                               int nq, const double* ref, int row, double* out, void* stream);

/* out[f,t,k,B] = sum_b conj(coef[b,f,A_k]) C[f,t,A_k,samp,b,B], A_k the k-th antenna != samp:
 * the operator from the sampled antenna's beam coefficients to visibilities
 * (get_bess_to_vis_from_contraction, per_ant_beam_sampler.py:453-463).  C (Nf,Nt,Nant,Nant,Nc,NB)
 * c128, or with col_only != 0 the column alone (Nf,Nt,Nant,Nc,NB); coef (Nc,Nf,Nant) c128; NB = Nc,
 * or 1 for a reference-beam contraction; out (Nf,Nt,Nant-1,NB) c128.                          */
int hydra_bess_to_vis(const double* C, const double* coef, int nf, int nt, int nant, int nc, int nb,
                      int samp, int col_only, double* out, void* stream);

/* ---- GCR operator ---------------------------------------------------------------- */

/* Per-row factors for rows r = (b,f,t) in (Nbl,Nf,Nt) order:
 *   gg[r]   = g_i conj(g_j)                 (c128)           ptsrc_sampler.py:53-59
 *   rows[r] = gg[r] * sqrt(inv_noise[r])    (c128)           ptsrc_sampler.py:66-71
 * gains (Nant,Nf,Nt) c128 or NULL (= 1); inv_noise (Nbl,Nf,Nt) f64 or NULL (= 1).
 * gg and/or rows may be NULL.                                                          */
int hydra_row_factors(const double* gains, const double* inv_noise, const int32_t* bl_i,
                      const int32_t* bl_j, int nbl, int nf, int nt, double* gg, double* rows,
                      void* stream);

/* Gram of a materialised operator on the FP64 matrix cores:
 *   G = Re( (diag(rows) R)^H (diag(rows) R) )   (Ns,Ns) f64, full symmetric matrix
 * R (M,Ns) c128, rows (M) c128 or NULL.  = v_re.T @ v_re + v_im.T @ v_im
 *                                         ptsrc_sampler.py:62-75; sh_sampler.py:393-398
 * If prior_std != NULL the epilogue also forms A = I + diag(s) G diag(s)
 *                                                             ptsrc_sampler.py:91-94
 * and if add_diag != NULL, A = G + diag(add_diag)             sh_sampler.py:399-401    */
size_t hydra_gram_workspace_bytes(long long m, int ns);
int hydra_gram(const double* R, const double* rows, long long m, int ns, const double* prior_std,
               const double* add_diag, double* G, void* workspace, size_t workspace_bytes,
               void* stream);

/* The two epilogues of hydra_gram applied in place to a full (Ns,Ns) matrix -- used when raw
 * Grams of several GPUs are summed first (comm.Reduce, ptsrc_sampler.py:86) and the prior is
 * applied to the total (ptsrc_sampler.py:91-94).                                          */
int hydra_prior_epilogue(double* G, int ns, const double* prior_std, const double* add_diag,
                         void* stream);

/* Same Gram without materialising R (R would be 273 GB at BASELINE config 2), i.e.
 * ptsrc_sampler.py:48-94 on the output of vis_simulator.py:281-289 / ptsrc_sampler.py:209-213:
 *   R'[(b,f,t),s] = g_i conj(g_j) sqrt(w_bft) conj(V[f,t,i_b,s]) V[f,t,j_b,s].
 * gains (Nant,Nf,Nt) c128 or NULL; inv_noise (Nbl,Nf,Nt) f64 or NULL.
 * Tables (built by hydra_amd/_tables.py) group the baselines by antenna block so that one
 * block pair's voltages fit in LDS: seg (nseg,6) int32 = [i0, ni, j0, nj, row0, nrows];
 * rowtab (nrows,3) int32 = [LDS row of antenna i, LDS row of antenna j, baseline index b].
 * Both are passed twice: a host copy (validated before launch) and the device copy.      */
size_t hydra_gram_fused_workspace_bytes(int nf, int nt, int nbl, int ns);
int hydra_gram_fused(const double* V, int nf, int nt, int nant, int ns, const double* gains,
                     const double* inv_noise, int nbl, const int32_t* seg_host,
                     const int32_t* rowtab_host, const int32_t* seg, const int32_t* rowtab,
                     int nseg, int nrows, const double* prior_std, const double* add_diag,
                     double* G, void* workspace, size_t workspace_bytes, void* stream);

/* The same Gram for the common case where (a) every antenna pair i<j appears exactly once and
 * (b) the noise weight of an (f,t) sample is the same for all baselines: exact closed form in
 * the antenna voltages,
 *   G[s,s'] = sum_ft 1/2 ( |sum_a u_a,s conj(u_a,s')|^2 - sum_a |u_a,s|^2 |u_a,s'|^2 ),
 *   u_a = w_ft^(1/4) |g_a| V[f,t,a,:],
 * a Hermitian rank-Nant update per (f,t) on the matrix cores instead of a rank-Nbl one
 * (same result as hydra_gram_fused up to rounding; ptsrc_sampler.py:48-94).
 * w_ft (Nf*Nt) f64 or NULL (= 1); gains (Nant,Nf,Nt) c128 or NULL.  Epilogues as hydra_gram.
 * same_beam != 0 asserts that |V[f,t,a,s]| does not depend on the antenna a (one beam model for
 * the whole array, as in hydra/example.py:143-172): the term sum_a |u_a,s|^2 |u_a,s'|^2 is then
 * rank one per (f,t) and costs one MFMA per output block and (f,t) instead of one per k-step.
 * hydra_noise_uniformity fills w_ft[f,t] = inv_noise[0,f,t] and sets *flag (device int32,
 * caller-zeroed) to 1 if any baseline's weight differs, i.e. if (b) does not hold.          */
int hydra_noise_uniformity(const double* inv_noise, int nbl, int nf, int nt, double* w_ft,
                           int32_t* flag, void* stream);
size_t hydra_gram_uniform_workspace_bytes(int nf, int nt, int nant, int ns);
int hydra_gram_uniform(const double* V, int nf, int nt, int nant, int ns, const double* gains,
                       const double* w_ft, int same_beam, const double* prior_std,
                       const double* add_diag, double* G, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Closed-form Gram for FLAGGED data (arbitrary inv_noise_var of ptsrc_sampler.py:62-75 as long as few
 * samples deviate): with w_ref[f,t] = max_b inv_noise[b,f,t],
 *   G = G_closed(all pairs, w_ref) - sum_{(b,f,t): w < w_ref} (w_ref - w) Re(conj(R'_bft) R'_bft^T),
 * i.e. hydra_gram_uniform(w_ft = w_ref, no epilogue) minus hydra_gram over the materialised rows of
 * the deviating samples, combined by hydra_gram_subtract (which applies the epilogue).
 *   hydra_noise_deviations: fills w_ref (Nf*Nt) and appends the flat indices e = b*Nf*Nt + ft of the
 *     samples with inv_noise[e] != w_ref[ft] to idx (int64, arbitrary order, at most `capacity`);
 *     *count_dev (device uint64) receives their total number, which may exceed capacity.
 *   hydra_flagged_rows: for the listed samples, R (nrows,Ns) c128 = conj(V_i) V_j and
 *     rows (nrows) c128 = g_i conj(g_j) sqrt(w_ref - w) -- the operands of hydra_gram.
 *   hydra_gram_subtract: G <- epilogue(G - C), epilogues as hydra_gram (ptsrc_sampler.py:91-94).   */
int hydra_noise_deviations(const double* inv_noise, int nbl, int nf, int nt, double* w_ref,
                           long long* idx, long long capacity, unsigned long long* count_dev,
                           void* stream);
int hydra_flagged_rows(const double* V, int nf, int nt, int nant, int ns, const double* gains,
                       const double* inv_noise, const double* w_ref, const int32_t* bl_i,
                       const int32_t* bl_j, int nbl, const long long* idx, long long nrows,
                       double* R, double* rows, void* stream);
int hydra_gram_subtract(double* G, const double* C, int ns, const double* prior_std,
                        const double* add_diag, void* stream);

/* ---- operator application (RHS, matrix-free LHS, model update) ------------------- */

/* out[r] = rowmul[r] * sum_s R[r,s] x[s]   (M) c128; x (Ns) f64; rowmul (M) c128 or NULL
 *                       ptsrc_sampler.py:104-106; example.py:974-976,1065; sh_sampler.py:394 */
int hydra_apply_R(const double* R, long long m, int ns, const double* x, const double* rowmul,
                  double* out, void* stream);

/* out[s] = scale[s] * sum_r Re( conj(R[r,s]) z[r] ) + add[s]   (Ns) f64
 * z (M) c128; scale, add (Ns) f64 or NULL.
 *                       ptsrc_sampler.py:124,138; sh_sampler.py:376-384,393-398           */
size_t hydra_adjoint_workspace_bytes(long long m, int ns);
int hydra_adjoint_R(const double* R, long long m, int ns, const double* z, const double* scale,
                    const double* add, double* out, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Factored forms of the two products above (operator given by V, never materialised;
 * ptsrc_sampler.py:104-106,124,138; example.py:974-976):
 *   out[b,f,t] = gg[b,f,t] * sum_s conj(V[f,t,i_b,s]) V[f,t,j_b,s] x[s]     (x NULL = ones)
 *   out[s]     = scale[s] * sum_{b,f,t} Re( conj(conj(V_i) V_j) z[b,f,t] ) + add[s]
 * For Nant <= 64 both run in antenna space on the matrix cores (C = conj(V) diag(x) V^T and
 * W = Zc V per (f,t)); hydra_apply_V does so up to 144 antennas (from 65 on a workgroup shares
 * the blocks of C and stages the voltages through LDS) and then needs
 * hydra_apply_V_workspace_bytes() of workspace (with workspace == NULL, or more antennas, it
 * loops over the baselines instead).  The adjoint for 65 .. 160 antennas:
 * hydra_adjoint_V_antspace below.                                                           */
size_t hydra_apply_V_workspace_bytes(int nf, int nt, int nant, int ns);
int hydra_apply_V(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                  const int32_t* bl_j, int nbl, const double* x, const double* rowmul,
                  double* out, void* workspace, size_t workspace_bytes, void* stream);
size_t hydra_adjoint_V_workspace_bytes(int nf, int nt, int nbl, int ns);
int hydra_adjoint_V(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                    const int32_t* bl_j, int nbl, const double* z, const double* scale,
                    const double* add, double* out, void* workspace, size_t workspace_bytes,
                    void* stream);
/* hydra_adjoint_V for arrays of ANY size in antenna space on the matrix cores (hydra_adjoint_V
 * itself does so up to 64 antennas and loops over the baselines on the vector units beyond):
 * W = Zc V per (f,t) with Zc[i][j] = conj(z_ij), then out[s] = scale[s] Re sum_i V_i,s conj(W_i,s)
 * + add[s]; the voltages of a 32-source tile are staged in LDS, Zc is read in MFMA-fragment order.
 * Precondition: every pair of (bl_i, bl_j) has i <= j and occurs once (a complete array, or any
 * duplicate-free list in antenna order); Nant <= 160.   ptsrc_sampler.py:124,138; sh_sampler.py:376-384
 * through the factored per-pixel operator.                                                        */
size_t hydra_adjoint_V_antspace_workspace_bytes(int nf, int nt, int nant, int nbl, int ns);
int hydra_adjoint_V_antspace(const double* V, int nf, int nt, int nant, int ns, const int32_t* bl_i,
                             const int32_t* bl_j, int nbl, const double* z, const double* scale,
                             const double* add, double* out, void* workspace,
                             size_t workspace_bytes, void* stream);
/* GCR right-hand-side row term:
 *   z[r] = conj(gg[r]) * ( (data[r] - model[r]) * w[r] + omega_n[r] * sqrt(w[r]) )
 * model may be NULL (=0), gg NULL (=1), omega_n NULL (=0).
 *                                      ptsrc_sampler.py:104-123; sh_sampler.py:423-429 */
int hydra_rhs_rows(const double* data, const double* model, const double* w,
                   const double* omega_n, const double* gg, long long m, double* z, void* stream);

/* z[r] = w[r] * y[r]   (complex y, real w) -- the N^-1 in R^H N^-1 R a  sh_sampler.py:470 */
int hydra_scale_rows(const double* y, const double* w, long long m, double* z, void* stream);

/* ---- solver ---------------------------------------------------------------------- */

/* y = A x, A (N,N) f64 row-major                                  linear_solver.py:431 */
int hydra_gemv(const double* A, int n, const double* x, double* y, void* stream);

/* y[q] = A[q] x[q] (accumulate == 0) or y[q] += A[q] x[q], q < nb: A (nb,N,N), x, y (nb,N) f64 --
 * the per-frequency likelihood term of the cosmological-field operator, added to the prior term
 * already in y                                                   cosmo_sampler.py:219-221 */
int hydra_gemv_batched(const double* A, int n, int nb, const double* x, double* y, int accumulate,
                       void* stream);

/* y (+)= A x for a ROW BLOCK A (nrows, ncols) of a larger matrix: the product each GPU forms when
 * the GCR operator is sharded by rows over the GPUs of a node (the matrix of 100 000 sources is
 * 80 GB; linear_solver.py:11-49 does the same per 2-D block over MPI ranks).                   */
int hydra_gemv_rows(const double* A, int nrows, int ncols, const double* x, double* y,
                    int accumulate, void* stream);

/* The same products for SYMMETRIC matrices (A[q] == A[q]^T exactly, stored in full): only the
 * 64 x 64 blocks on and below the diagonal are read -- half the HBM traffic.  workspace >=
 * hydra_symv_workspace_bytes(n, nb) holds the per-block-column pieces that a second launch sums
 * in a fixed order (no atomics).  The GCR operators are symmetric by construction
 * (ptsrc_sampler.py:62-94, cosmo_sampler.py:284).                                          */
size_t hydra_symv_workspace_bytes(int n, int nb);
int hydra_symv_batched(const double* A, int n, int nb, const double* x, double* y, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Lower triangle of a symmetric matrix in packed row order, packed[i (i+1) / 2 + j] = A[i][j] for
 * j <= i, and back (both triangles written, exactly symmetric).  The sum of the ranks' Grams
 * (comm.Reduce of the full (Ns,Ns) square, ptsrc_sampler.py:86; sh_sampler.py:485 for the SH
 * left-hand side) then moves 4 n (n+1) instead of 8 n^2 bytes over xGMI.  packed: n (n+1) / 2 f64. */
int hydra_pack_lower(const double* A, int n, double* packed, void* stream);
int hydra_unpack_lower(const double* packed, int n, double* A, void* stream);

/* hydra_symv_batched for one matrix WITHOUT a workspace argument and without any allocation: the
 * per-block-column pieces go to a device array inside the library (8 MiB on every GPU the library
 * runs on, held until the process ends).  No counterpart in the reference: it is for callers that
 * cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  n <= hydra_symv_static_max_n() (8192), larger n is rejected with HYDRA_ERR_ARG.
 * Same kernels, same summation order and bit-identical y as hydra_symv_batched(A, n, 1, ...).
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different
 * streams or threads) race; calls on one stream are ordered and safe.
 * hydra_symv_static_max_n() is also the SET-UP call: it looks up the array's address on the current
 * device (a symbol lookup, not a stream operation).  Call it once per device before the first
 * hydra_symv_static that runs under graph capture; after it hydra_symv_static is two kernel
 * launches and nothing else (without it the first call on a device does the lookup itself).  */
int hydra_symv_static_max_n(void);
int hydra_symv_static(const double* A, int n, const double* x, double* y, int accumulate,
                      void* stream);

/* Conjugate gradients exactly as linear_solver.cg (linear_solver.py:378-509):
 * convergence tested BEFORE each iteration, element-wise |r_i| < tol or ||r||_2 < tol.
 * State (device): x, r, p (N) f64 and `st` = hydra_cg_state_len() doubles (776):
 *   st[0]=rr  st[1]=done(0/1)  st[2]=iterations done  st[3]=max|r| or ||r||  st[4..] scratch
 *   (per-workgroup partial sums of the multi-workgroup step used for long vectors).
 * hydra_cg_init: given x and r0 = b - A x (caller computes A x): p=r, rr=r.r, test.
 * hydra_cg_step: given Ap = A p, one fused update (both dot products, x/r/p updates, test).
 *   Once done==1 a step is a no-op, so the host may enqueue steps ahead of the test.     */
int hydra_cg_init(int n, const double* b, const double* Ax, double* r, double* p, double* st,
                  double abs_tol, int use_norm_tol, void* stream);
int hydra_cg_step(int n, const double* Ap, double* x, double* r, double* p, double* st,
                  double abs_tol, int use_norm_tol, void* stream);
int hydra_cg_state_len(void);

/* Whole dense solve on the device: x (in: x0, out: solution), A (N,N), b (N).
 * workspace >= hydra_cg_workspace_bytes(n).  Runs at most maxiters iterations, enqueuing
 * `check_every` iterations between host reads of the done flag.  *iters_host receives the
 * iteration count.  symmetric != 0: the caller vouches that A == A^T exactly and the products
 * A p read half the matrix (hydra_symv_batched).    linear_solver.py:378-509 with Amat given */
size_t hydra_cg_workspace_bytes(int n);
int hydra_cg_solve_dense(const double* A, const double* b, double* x, int n, int maxiters,
                         double abs_tol, int use_norm_tol, int symmetric, int check_every,
                         int* iters_host, void* workspace, size_t workspace_bytes, void* stream);

/* Dense fallback: solve A x = b for symmetric positive definite A by blocked Cholesky
 * (stands for scipy.linalg.solve(A, b, assume_a='her'), example.py:944).  A is overwritten
 * by its factor (lower triangle); b by x; b == NULL: factor only.  *info_dev = 1 if A is not
 * positive definite.                                                                      */
int hydra_chol_solve(double* A, double* b, int n, int32_t* info_dev, void* stream);

/* b <- A^-1 b from the factor hydra_chol_solve left in A (the two triangular solves alone): for
 * systems whose matrix stays the same from sample to sample -- the SH head's left-hand side
 * R^T N^-1 R + S^-1 (sh_sampler.py:389-402) changes only with the noise and prior variances, so
 * one factorisation serves every sample of a run where the reference runs CG anew each time
 * (example.py:1014-1052).  Same alignment rules as hydra_chol_solve.                        */
int hydra_chol_apply(const double* L, double* b, int n, void* stream);

/* X <- L^-1 for the lower-triangular factor hydra_chol_solve left in A (row stride ld; what lies
 * above the diagonal of L is not read), X (n, n) with row stride ldx, zero above the diagonal.
 * For the same constant systems (sh_sampler.py:389-402, example.py:1014-1052) at small n, where
 * two matrix-vector products with L^-1 and L^-T beat the two substitution chains of
 * hydra_chol_apply (hydra_amd.linear_solver.cho_factor(inverse_factor=True)): done once per
 * matrix, column block by column block, one workgroup per 32 columns.                      */
int hydra_tri_inverse(const double* L, long long ld, int n, double* X, long long ldx, void* stream);

/* y += A x for the symmetric operator of the source head (ptsrc_sampler.py:62-94) of which an
 * in-place factorisation has left only the strict upper triangle (the lower one holds L) and the
 * saved diagonal: the driver's residual check |A x - b| (example.py:974-976 style diagnostics) on a
 * kept factor without a second copy of the matrix.  A (nloc, >= n) with row stride ld: the rows a
 * rank owns, local row i = global row rows[i] (int64, ascending; all of 0..n-1 on one GPU);
 * diag (nloc); x, y (n).  A rank adds its rows' part; the caller sums y over ranks.            */
int hydra_symv_upper(const double* A, long long ld, int nloc, int n, const long long* rows,
                     const double* diag, const double* x, double* y, void* stream);

/* The large-matrix algorithm behind hydra_chol_solve (taken automatically for even n >= 6144):
 * 256-wide panels factored by 32-wide steps, then ONE rank-256 update of the rest of the lower
 * triangle per panel on the FP64 matrix cores (C -= P P^T, P = the finished panel read in place),
 * and panel-wise multi-workgroup triangular solves.  N^3/3 flops, FP64-MFMA-bound: the 100 000-source
 * operator of BASELINE config 5 (80 GB) is the case it exists for -- the reference's driver solves
 * the source head densely whatever --solver says (example.py:940-947).  A (N,N) f64, 16-byte
 * aligned, N even; the lower triangle is overwritten by L (the strict upper triangle is left
 * stale).  b == NULL: factor only.  *info_dev = 1 if A is not positive definite.              */
int hydra_chol_solve_blocked(double* A, double* b, int n, int32_t* info_dev, void* stream);

/* Pieces of the same dense solve (example.py:940-947) for an operator sharded by rows over GPUs
 * (hydra_amd.linear_solver.factor_dense_rows), each on arrays with an explicit leading dimension:
 *   hydra_chol_panel : P (nrows x ncols, ld) = a diagonal block (its lower triangle) on top of the
 *                      rows below it: L11 <- chol(block), rows <- rows L11^-T, by the 32-wide steps
 *                      of hydra_chol_solve (ncols <= 256 in practice; a multiple of 32 unless nrows
 *                      == ncols)
 *   hydra_trsv_block : b <- L^-1 b (transpose = 0) or L^-T b (1) with the n x n lower triangle L
 *                      (n <= 12288), one diagonal block of the sharded factor
 *   hydra_gemv_sub   : y[nrows] -= M x[ncols] (transpose = 0) or y[ncols] -= M^T x[nrows] (1) for
 *                      an nrows x ncols piece M of the factor: the sweeps' off-diagonal part    */
int hydra_chol_panel(double* P, long long ld, int nrows, int ncols, int32_t* info_dev, void* stream);
int hydra_trsv_block(const double* L, long long ld, int n, double* b, int transpose, void* stream);
int hydra_gemv_sub(const double* M, long long ld, int nrows, int ncols, const double* x, double* y,
                   int transpose, void* stream);

/* The trailing update of that factorisation as a building block for matrices sharded by rows over
 * GPUs (hydra_amd.linear_solver.solve_dense_rows; the same dense solve, example.py:940-947):
 *   C[i][j] -= sum_{k < kdim} PA[i][k] PB[j][k]   for i < mrows, j < ncols, j <= i + diag
 * on the FP64 matrix cores.  C (mrows, >= ncols) with row stride ldc: the rank's rows of the
 * matrix right of the panel; PA (mrows, kdim), row stride lda: the same rows of the finished
 * panel(s); PB (ncols, kdim), row stride ldb: the panel rows that belong to C's columns (gathered
 * from all ranks); diag = (global row of C's first row) - (global column of C's first column);
 * kdim a multiple of 16 (256 per panel, 2048 per super-panel).  PA, PB 16-byte aligned with even
 * strides.                                                                                     */
int hydra_chol_update_rect(double* C, long long ldc, const double* PA, long long lda,
                           const double* PB, long long ldb, int mrows, int ncols, int diag, int kdim,
                           void* stream);

/* ---- spherical-harmonic response (simulate_vis_per_alm) -------------------------- */

/* RING pixel centres: theta, phi (Npix = 12 nside^2)                vis_simulator.py:659 */
int hydra_healpix_pix2ang(int nside, double* theta, double* phi, void* stream);

/* Ymap (Npix, Ncols) f64, pixel-major: column n holds alm2map(e_n * {1, i}) * (4 pi / Npix)
 * * amplitude; Ncols = 2 nalm - (lmax+1) in the hydra mode order [Re (nalm); Im without m=0]
 * when packed != 0, or the reference's raw 2*nalm order when packed == 0.
 *                                              vis_simulator.py:702-730; sh_sampler.py:127-131 */
int hydra_sh_mode_maps(int lmax, int nside, double amplitude, int packed, double* ymap,
                       void* stream);

/* SH response operator on the FP64 matrix cores:
 *   out[b,f,t,n] = scale_f[f] * sum_p conj(V[f,t,i_b,p]) V[f,t,j_b,p] Ymap[p,n]
 * (Nbl,Nf,Nt,Nm) c128; V from hydra_voltages with the pixels as unit-flux sources.
 *                                     vis_simulator.py:702-730; sh_sampler.py:152-168 */
int hydra_sh_project(const double* V, int nf, int nt, int nant, int npix, const int32_t* bl_i,
                     const int32_t* bl_j, int nbl, const double* ymap, int nm,
                     const double* scale_f, double* out, void* stream);

/* The same projection, tiled for the matrix pipe (see csrc/sh.hip): a workgroup owns an
 * antenna-block pair (8 x 16 antennas = 128 baseline slots) x 128 modes and walks only the
 * pixels above the horizon at time t.
 *   pixidx (Nt, npix_max) int32: indices of the pixels with non-zero voltage at time t, first
 *     npix_t[t] entries valid (vis_simulator.py:237,272 zeroes the others, so skipping them is
 *     exact); tiles (ntile,2) int32 = first antenna (i0, j0) of the two blocks;
 *     bmap (ntile,128) int32 = output baseline index of slot (i-i0)*16 + (j-j0), or -1;
 *   ymap_padded (Npix, nm_padded) f64 with nm_padded a multiple of 128, zero beyond nm;
 *   out (Nbl, nf_total, Nt, Nm) c128: the channels [f0, f0+nf) are written (V holds nf channels),
 *   so a long frequency axis can be projected block by block.  Same result as hydra_sh_project
 *   up to the order of the pixel sum.                vis_simulator.py:702-730; sh_sampler.py:152-168 */
int hydra_sh_project_tiled(const double* V, int nf, int nt, int nant, int npix,
                           const int32_t* pixidx, const int32_t* npix_t, int npix_max,
                           const int32_t* tiles, const int32_t* bmap, int ntile,
                           const double* ymap_padded, int nm_padded, int nm, const double* scale_f,
                           int f0, int nf_total, double* out, void* stream);

/* ---- gains ------------------------------------------------------------------------ */

/* v[b,f,t] *= g_i conj(g_j) (1 + x_i + conj(x_j)) in place; pert may be NULL
 *                                                                   utils.py:63-110 */
int hydra_apply_gains(double* v, const double* gains, const double* pert, const int32_t* bl_i,
                      const int32_t* bl_j, int nbl, int nf, int nt, void* stream);

/* ---- linearised gain head (hydra/gain_sampler.py; BASELINE config 4) ---------------
 * The reference's sparse (Nbl x Nant) projectors A_real (+1,+1) / A_imag (+1,-1)
 * (gain_sampler.py:51-78) are passed as a CSR adjacency antenna -> baselines:
 *   adj_ptr (Nant+1), adj_b (nadj) baseline index, adj_o (nadj) the baseline's other antenna,
 *   adj_role (nadj) 0 if the antenna is the first of the pair, 1 if the second.
 * adj_ptr is passed twice (host copy validated before launch, device copy used by the kernel).
 * Gain-parameter vectors x are (nvec, 2*Nant*Nmodes) f64: real parts then imaginary parts,
 * each (Nant, Nmodes) -- the layout of gain_sampler.py:257-258,294.  nft = Nf*Nt.          */

/* W[b,f,t] = |model_vis|^2 * inv_noise (inv_noise NULL = 1): all that survives between A and
 * A^H in apply_operator_mpi                                      gain_sampler.py:278-284 */
int hydra_gain_weights(const double* model_vis, const double* inv_noise, long long m, double* W,
                       void* stream);

/* g[v,a,f,t] = sum_m Fbasis[m,f,t] * pspec_sqrt[m] (xre[v,a,m] + i xim[v,a,m])  (c128)
 *                                                                gain_sampler.py:257-272 */
int hydra_gain_basis_fwd(const double* x, int nvec, int nant, int nmodes, int nft,
                         const double* Fbasis, const double* pspec_sqrt, double* g, void* stream);

/* y = A^H N^-1 A g for nvec antenna fields at once (one pass over W per 8 vectors):
 *   y[v,a] = g[v,a] sum_{b in adj(a)} W[b] + conj(sum_{b in adj(a)} W[b] g[v,other(b)])
 *                                                                gain_sampler.py:278-284 */
int hydra_gain_pairs(const double* g, int nvec, int nant, int nft, const double* W, int nbl,
                     const int32_t* adj_ptr_host, const int32_t* adj_ptr, const int32_t* adj_b,
                     const int32_t* adj_o, int nadj, double* y, void* stream);

/* Same operator (gain_sampler.py:278-284) for pair lists without duplicates, tiled through LDS
 * (16 antennas x 64 (f,t) per workgroup; see csrc/gain.hip): pairb (Nant,Nant) int32 = baseline index of the unordered
 * pair {a,o} (symmetric), -1 if absent; an auto-correlation counts twice.                   */
int hydra_gain_pairs_tiled(const double* g, int nvec, int nant, int nft, const double* W, int nbl,
                           const int32_t* pairb, double* y, void* stream);

/* The same operator (gain_sampler.py:278-284) with every weight read ONCE per application:
 * a workgroup owns a few (f,t) samples and all antennas (fields and sums in LDS) and streams its
 * weights, stored in the order of use, through LDS-DMA buffers (csrc/gain.hip, "streamed form").
 * For pair lists without duplicates and without auto-correlations, up to about 1600 antennas.
 *   hydra_gain_stream_weight_count : doubles of the stream layout Wt (0: Nant does not fit)
 *   hydra_gain_weights_stream      : Wt <- |model_vis|^2 inv_noise (inv_noise NULL: 1), the
 *                                    fused weight of gain_sampler.py:278-284, in stream layout;
 *                                    pairb as for hydra_gain_pairs_tiled
 *   hydra_gain_pairs_stream        : y[a] = g[a] sw[a] + conj(sum_o W[{a,o}] g[o]) for ONE field
 *                                    g (Nant,Nft) c128; sw (Nant,Nft) f64 = sum_{b in adj(a)} W[b]
 *                                    (what an application to g = 1 with sw = NULL returns in
 *                                    Re y); sw NULL: the second term alone                       */
size_t hydra_gain_stream_weight_count(int nant, int nft);
int hydra_gain_weights_stream(const double* model_vis, const double* inv_noise, int nbl, int nant,
                              int nft, const int32_t* pairb, double* Wt, void* stream);
int hydra_gain_pairs_stream(const double* g, int nant, int nft, const double* Wt, const double* sw,
                            double* y, void* stream);

/* The same operator for a BATCH of fields on the FP64 matrix cores (BASELINE config 4, "batched CG
 * matvecs"; gain_sampler.py:278-284 applied to several vectors at once): per (f,t) sample the pair
 * sums are the real symmetric matrix W[{a,o}] times the 2 nvec real columns of the fields.
 *   hydra_gain_blocks_weight_count : doubles of the block layout Wb (0: Nant does not fit)
 *   hydra_gain_weights_blocks      : Wb <- |g_a conj(g_o) model_vis|^2 inv_noise in 16 x 16 antenna
 *                                    blocks (gains NULL: model_vis carries them; inv_noise NULL: 1);
 *                                    pairb as for hydra_gain_pairs_tiled, no duplicate pairs
 *   hydra_gain_pairs_mfma          : y[v][a] = g[v][a] sw[a] + conj(sum_o W[{a,o}] g[v][o]) for
 *                                    nvec fields g (nvec,Nant,Nft) c128, every weight read once per
 *                                    8 vectors; sw (Nant,Nft) = sum of W over the pairs of an
 *                                    antenna, or NULL for 0                                      */
size_t hydra_gain_blocks_weight_count(int nant, int nft);
int hydra_gain_weights_blocks(const double* model_vis, const double* gains, const double* inv_noise,
                              int nbl, int nant, int nft, const int32_t* pairb, double* Wb,
                              void* stream);
int hydra_gain_pairs_mfma(const double* g, int nvec, int nant, int nft, const double* Wb,
                          const double* sw, double* y, void* stream);
/* The whole data term of the operator for a batch, out = S^1/2 F^H [pair sums] F S^1/2 x (+ add)
 * (gain_sampler.py:256-301): hydra_gain_basis_fwd, hydra_gain_pairs_mfma and hydra_gain_basis_adj
 * in one call with the fields of a pass of 8 vectors kept in a sample-major layout
 * ([ft][antenna][2 x 8 real columns]) in the workspace, so that the pair kernel reads and writes
 * whole lines.  x, add, out (nvec, 2 Nant Nmodes) f64 as for hydra_gain_basis_fwd / _adj.      */
size_t hydra_gain_operator_blocks_workspace_bytes(int nant, int nft);
int hydra_gain_operator_blocks(const double* x, int nvec, int nant, int nmodes, int nft,
                               const double* Fbasis, const double* pspec_sqrt, const double* Wb,
                               const double* sw, const double* add, double* out, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Everything a gain GCR sample needs from the visibilities, in ONE pass over them -- the
 * driver's sequence example.py:690-745: ggv = apply_gains(model) (utils.py:63-110), resid = data -
 * ggv, the visibility half of construct_rhs_mpi (gain_sampler.py:201-213: z = resid N^-1 +
 * omega_r N^-1/2, A^H z with model_vis = ggv) and the operator weights |ggv|^2 N^-1
 * (gain_sampler.py:278-284).  For pair lists as for hydra_gain_pairs_stream whose baselines are
 * all listed as (lower antenna index, higher antenna index).
 *   data, model_vis, omega_r (Nbl,Nft) c128 (omega_r NULL: 0), inv_noise (Nbl,Nft) f64 (NULL: 1),
 *   gains (Nant,Nft) c128, pairb as for hydra_gain_pairs_tiled;
 *   Wt: stream layout, hydra_gain_stream_weight_count doubles (same content as
 *       hydra_gain_weights_stream on ggv);
 *   sw (Nant,Nft) f64 = sum_{b in adj(a)} W[b];
 *   yy (Nant,Nft) c128 = apply_proj_conj(z) -- feed to hydra_gain_basis_adj for the RHS.          */
int hydra_gain_prepare_stream(const double* data, const double* model_vis, const double* gains,
                              const double* inv_noise, const double* omega_r, int nbl, int nant,
                              int nft, const int32_t* pairb, double* Wt, double* sw, double* yy,
                              void* stream);

/* apply_proj: out[b,f,t] = (x_i + conj(x_j)) model_vis[b,f,t]    gain_sampler.py:81-106 */
int hydra_gain_proj(const double* x, const int32_t* bl_i, const int32_t* bl_j, int nbl,
                    const double* model_vis, int nft, double* out, void* stream);

/* apply_proj_conj: y[a] = sum_{b: a first} v conj(mv) + sum_{b: a second} conj(v conj(mv))
 *                                                                gain_sampler.py:109-142 */
int hydra_gain_proj_conj(const double* v, const double* model_vis, int nant, int nft,
                         const int32_t* adj_ptr_host, const int32_t* adj_ptr,
                         const int32_t* adj_b, const int32_t* adj_role, int nadj, double* y,
                         void* stream);

/* out[v] = [Re; Im] of pspec_sqrt[m] sum_{f,t} conj(Fbasis[m,f,t]) y[v,a,f,t]  (+ add[v] if
 * add != NULL; add has the layout of out)            gain_sampler.py:213-220,286-301 */
int hydra_gain_basis_adj(const double* y, int nvec, int nant, int nmodes, int nft,
                         const double* Fbasis, const double* pspec_sqrt, const double* add,
                         double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_HIP_H */
