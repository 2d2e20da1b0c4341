/* tike_amd.h -- C ABI of the MI355X-native ptychography hot path.
 *
 * Drop-in boundary for the tike.operators / tike.ptycho.solvers hot path
 * (reference: AdvancedPhotonSource/tike @ 2024_10_08; file:line citations are
 * relative to the reference tree).  The reference has no FFI for this path:
 * its operators are Python classes over CuPy, one JIT-compiled CUDA source
 * (src/tike/operators/cupy/convolution.cu) and cuFFT.  These entry points are
 * what a ctypes binding of those operators calls instead.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the calling thread's current HIP
 *     device; complex64 arrays are interleaved (re, im) float32, C-contiguous;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     calls are asynchronous on that stream, allocate nothing and never
 *     synchronise (tike_init, called once per device, is the exception);
 *   - return value: 0 on success, a hipError_t (> 0) from the runtime, or
 *     TIKE_ERR_ARG / TIKE_ERR_UNSUPPORTED for violated shape relations -- the
 *     Python layer raises ValueError / RuntimeError from these, mirroring the
 *     reference's assert / ValueError behaviour;
 *   - scan positions are float32 (y, x) of the patch's minimum corner
 *     (reference patch.py:102-106).
 */
#ifndef TIKE_AMD_H
#define TIKE_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define TIKE_ERR_ARG 1000001
#define TIKE_ERR_UNSUPPORTED 1000002
#define TIKE_ERR_COMM 2000000 /* + ncclResult_t of a failed RCCL call */
#define TIKE_COMM_ID_BYTES 128

/* Version of THIS header.  Bumped whenever an entry point is added, removed or
 * changes its argument list (entries take up to 31 positional arguments, so a
 * binding built against another header would pass garbage without noticing).
 * A binding compares tike_abi_version() of the loaded library with the
 * TIKE_ABI_VERSION it was written against before its first call
 * (tike_amd/_lib.py does; INTEGRATION.md shows the check). */
#define TIKE_ABI_VERSION 11

/* sha256 (64 hex digits) of the sources the library was built from: the PMC
 * traffic files under profiles/ carry it, and bench.py withholds a traffic
 * figure whose kernels are no longer the ones loaded.  Host only. */
const char* tike_build_id(void);

/* The TIKE_ABI_VERSION the library was built from.  No device, no allocation. */
int tike_abi_version(void);

/* Deterministic mode (process-wide switch; off by default).  Off: sums that
 * several workgroups contribute to -- the object gradient, the probe gradient,
 * per-pattern costs, the sums of the step-size statistics -- are float atomics,
 * as in the reference (operators/cupy/convolution.cu:51-66): their order and
 * with it the last bits of every result change from run to run.  On: every
 * such sum of the lstsq_grad / rpie path (gaussian model, one object slice)
 * has one contributor per address, or its partial sums are written to
 * `scratch` and added in a fixed order -- two runs give bit-identical
 * iterates.  scratch: caller-owned DEVICE memory of `bytes` bytes that the
 * library may use from then on (one user at a time: launches on one stream);
 * 256 MiB serve a 256 x 256 x 8-mode minibatch; an entry that needs more
 * returns TIKE_ERR_ARG.  Covered since round 6: the sums of cgrad's direction
 * and of its all-steps-at-once line search (tike_cgrad_direction,
 * tike_cgrad_line_search_linear).  The per-mode poisson step lengths formed
 * from the forward hand-off (tike_poisson_steps_handoff,
 * tike_poisson_steps_grad_ifft2_pass1) keep their atomics: under the switch
 * the Python layer routes the poisson model through the stored-far-plane
 * entries (tike_poisson_steps: one workgroup per pattern, no atomics). */
int tike_set_deterministic(int on, void* scratch, long bytes);

/* Create the per-device constant tables (FFT twiddles).  Allocates; call once
 * per device before capturing graphs.  Every other call does it lazily. */
int tike_init(void);

/* ---- Patch: replaces fwd_patch / adj_patch<float2,float2,float>
 * (convolution.cu:146-165 as launched by operators/cupy/patch.py:79-188).
 * images (nimage,H,W) c64; positions (nimage,nscan,2) f32;
 * fwd: patches (nimage, nscan*nrepeat, padded, padded) -- only the centred
 * patch_width window is written; adj: images += scatter(patches), patches
 * (nimage, npatch, padded, padded) with (nscan*nrepeat) % npatch == 0 and
 * npatch >= nrepeat (broadcast, patch.py:155). */
int tike_patch_fwd(const void* images, void* patches, const float* positions, int nimage, int H,
                   int W, int nscan, int nrepeat, int patch_width, int padded_width,
                   void* stream);
int tike_patch_adj(void* images, const void* patches, const float* positions, int nimage, int H,
                   int W, int nscan, int nrepeat, int patch_width, int padded_width, int npatch,
                   void* stream);

/* ---- Convolution (operators/cupy/convolution.py:58-154), single image.
 * psi (H,W); scan (nscan,2); probe (1|nscan, S, pw, pw) selected by
 * probe_per_scan; nearplane (nscan, S, det, det).
 * conv_adj accumulates into psi; conv_adj_probe writes (nscan,S,pw,pw). */
int tike_conv_fwd(const void* psi, const float* scan, const void* probe, int probe_per_scan,
                  void* nearplane, int nscan, int S, int pw, int det, int H, int W, void* stream);
int tike_conv_adj(const void* nearplane, const float* scan, const void* probe,
                  int probe_per_scan, void* psi, int nscan, int S, int pw, int det, int H, int W,
                  void* stream);
int tike_conv_adj_probe(const void* nearplane, const float* scan, const void* psi,
                        void* probe_adj, int nscan, int S, int pw, int det, int H, int W,
                        void* stream);

/* ---- Propagation: replaces cuFFT behind CachedFFT._fft2/_ifft2
 * (operators/cupy/propagation.py:43-73, cache.py:66-82).
 * ntile tiles of n x n c64; out may alias in (overwrite); every element is
 * multiplied by `scale` (norm='ortho' -> 1/n both ways).
 * Any n cuFFT would be handed by a detector: powers of two 32..1024 on the
 * register engines; every other n = 2^a 3^b 5^c 7^d 11^e 13^f <= 4096 (96,
 * 192, 320, 384, 640, 768, 2048 ...) by mixed-radix lines in LDS; any other
 * n <= 2048 (45 * 23, 127, primes) by Bluestein's chirp-z over that engine.
 * TIKE_ERR_UNSUPPORTED only beyond those bounds (tike_fft2_supported). */
int tike_fft2(const void* in, void* out, long ntile, int n, int inverse, float scale,
              void* stream);

/* 1 when tike_fft2 takes tiles of n x n, else 0 (host only; the plan tables
 * of a size -- twiddles, chirp -- are built at its first transform and kept
 * per (n, device), as the reference keeps its cuFFT plans, cache.py:32-46). */
int tike_fft2_supported(int n);

/* The shape-general engine of tike_fft2 called directly, for EVERY n it
 * supports (powers of two included: the cross-check of the two engines), with
 * its grouping as arguments: lines (rows / columns) a workgroup transforms
 * together, rounded down to a power of two that fits LDS; 0 = the planner's
 * choice.  Same contract as tike_fft2 otherwise. */
int tike_fft2_general(const void* in, void* out, long ntile, int n, int inverse, float scale,
                      int lines_per_group_rows, int lines_per_group_cols, void* stream);

/* ---- Ptycho.fwd fused (operators/cupy/ptycho.py:114-129):
 * farplane[n][s] = scale * FFT2( pad( patch_n(psi) * probe_n[s] ) ).
 * The probe at position n is probe[n|0][s] or, when eigen_weights != NULL,
 * weights[n][0][s]*probe[0][s] + sum_c weights[n][c+1][s]*eigen[c][s]
 * (ptycho/probe.py:272-303); eigen_probe (num_eigen, eigen_modes, pw, pw),
 * eigen_weights (nscan, num_eigen+1, S) f32. farplane (nscan,S,det,det).
 * sub_batch: det = 256 / 512 run as two streaming kernels per sub-batch of
 * that many positions, so that the hand-off between them stays in the
 * Infinity Cache (0 = the library's default, 256 MiB of far plane; < 0 = one
 * batch); results do not depend on it. */
int tike_ptycho_fwd(const void* psi, const float* scan, const void* probe, int probe_per_scan,
                    const void* eigen_probe, const float* eigen_weights, int num_eigen,
                    int eigen_modes, void* farplane, int nscan, int S, int pw, int det, int H,
                    int W, float scale, int sub_batch, void* stream);

/* ---- Ptycho.adj fused (operators/cupy/ptycho.py:148-176 = propagation.py:59-73
 * IFFT2, then convolution.py:103-127 `adj` and :129-154 `adj_probe`):
 *   chi[n][s]       = scale * IFFT2( farplane[n][s] )
 *   probe_adj[n][s] = conj( patch_n(psi) ) * chi[n][s]          (nscan,S,pw,pw)
 *   psi_adj         = sum_n scatter_n( sum_s conj(probe[n|0][s]) * chi[n][s] )   (H,W), OVERWRITTEN
 * Probe window = detector, det in {128, 256, 512}, S <= 8 (TIKE_ERR_UNSUPPORTED
 * otherwise: use tike_ifft2_crop + tike_conv_adj + tike_conv_adj_probe).
 * Scan positions must keep the patch and its +1 taps inside the image
 * (ptycho/position.py:600-628 check_allowed_positions, which the reference
 * relies on too, convolution.cu:113-133); pixels falling outside are dropped.
 * farplane is read only.  probe_adj doubles as the workspace of the two-pass
 * inverse transform (it must not alias farplane).  Caller-owned scratch:
 * objproj_work (nscan,pw,pw) c64, acc_work (2,H,W) f32; nothing is allocated.
 * sub_batch > 0: positions per sub-batch as in tike_ptycho_fwd; 0 or < 0: one
 * batch, the default (measured faster for the adjoint: 0.391 -> 0.438 of the
 * roofline at 256^2 x 1 mode). */
int tike_ptycho_adj(const void* farplane, const void* probe, int probe_per_scan,
                    const float* scan, const void* psi, void* psi_adj, void* probe_adj,
                    void* objproj_work, float* acc_work, int nscan, int S, int pw, int det,
                    int H, int W, float scale, int sub_batch, void* stream);

/* ---- Ptycho.fwd + intensity, position-major (one workgroup per position
 * walks all S modes): same far-plane as tike_ptycho_fwd plus
 * intensity[n] = sum_s |farplane[n][s]|^2 (ptycho.py:18-23) accumulated in
 * registers, so the far-plane is not re-read to form it.  det in {128, 256, 512}
 * (TIKE_ERR_UNSUPPORTED otherwise: use tike_ptycho_fwd + tike_intensity).
 * intensity (nscan,det,det) f32 may be NULL.  With eigen_weights the probe of
 * mode s at position n is unique_probe[n][s] for s < eigen_modes (from
 * tike_varying_probe) and eigen_weights[n][0][s] * probe[s] otherwise.
 * patches (nscan,pw,pw) c64, may be NULL: the object patches O_n = Patch.fwd(psi)
 * (lstsq.py:524-531), stored while they are in registers. */
int tike_ptycho_fwd_intensity(const void* psi, const float* scan, const void* probe,
                              int probe_per_scan, const void* unique_probe,
                              const float* eigen_weights, int num_eigen, int eigen_modes,
                              void* farplane, float* intensity, void* patches, int nscan, int S,
                              int pw, int det, int H, int W, float scale, void* stream);

/* ---- the same forward model for the intensity ONLY (det = 256;
 * TIKE_ERR_UNSUPPORTED otherwise): the far-plane waves are formed in registers
 * and never stored; `scratch` (nscan,S,det,det) c64 receives instead the input
 * of the column pass of every tile, the operand of tike_grad_ifft2_crop. */
int tike_ptycho_fwd_intensity_only(const void* psi, const float* scan, const void* probe,
                                   int probe_per_scan, const void* unique_probe,
                                   const float* eigen_weights, int num_eigen, int eigen_modes,
                                   void* scratch, float* intensity, int nscan, int S, int pw,
                                   int det, int H, int W, float scale, void* stream);

/* tike_ptycho_fwd_intensity_only followed by tike_gradient_scale, in one
 * launch: gscale and costs are formed from the intensity while it is still in
 * registers (objective.py:31-44,47-70,90-125; lstsq.py:444-502).  intensity may
 * be NULL (not stored); costs may be NULL.  det = 256. */
int tike_ptycho_fwd_gradient_scale(const void* psi, const float* scan, const void* probe,
                                   int probe_per_scan, const void* unique_probe,
                                   const float* eigen_weights, int num_eigen, int eigen_modes,
                                   void* scratch, float* intensity, void* patches,
                                   const float* data, const unsigned char* measured,
                                   float* gscale, float* costs, int nscan, int S, int pw, int det,
                                   int H, int W, float scale, int model, float unmeasured_scaling,
                                   long num_measured, void* stream);

/* ---- the same, split in two launches (det = 256 or 512; tike_fwd_pass1 also
 * det = 128, for the multislice chain below).  tike_fwd_pass1: bilinear
 * gather * probe -> row transforms -> radix-16 column stage; scratch
 * (nscan,S,det,det) receives the UNSCALED input of the column pass of every
 * tile, patches (nscan,pw,pw, may be NULL) the object patches O_n.  The varying
 * probe of the first eigen_modes modes is read from unique_probe when given,
 * otherwise formed on the fly from eigen_probe (probe.py:272-303).
 * tike_fwd_gradient_scale: streams that scratch once, forms F = scale *
 * (column pass) in registers, I = sum_s |F_s|^2, and emits gscale, the costs
 * (either may be NULL, not both: a cost-only call is a line-search probe of
 * cgrad) and optionally the intensity (may be NULL) -- operands as
 * tike_ptycho_fwd_gradient_scale; data is float32, or uint16 when data_u16 != 0
 * (detector counts that arrived as <= 16-bit integers stay 16-bit in HBM,
 * ptycho.py:383-390).  farplane (nscan,S,det,det), if not NULL, also receives
 * the far-plane waves F themselves (for the pipelines that keep them: poisson
 * per-mode steps, the 512^2 inverse); it must not alias scratch. */
int tike_fwd_pass1(const void* psi, const float* scan, const void* probe, int probe_per_scan,
                   const void* unique_probe, const void* eigen_probe,
                   const float* eigen_weights, int num_eigen, int eigen_modes, void* scratch,
                   void* patches, int nscan, int S, int pw, int det, int H, int W, void* stream);
int tike_fwd_gradient_scale(const void* scratch, const void* data, int data_u16,
                            const unsigned char* measured, float* gscale, float* intensity,
                            float* costs, void* farplane, int nscan, int S, int det, float scale,
                            int model, float unmeasured_scaling, long num_measured,
                            void* stream);

/* ---- far-plane gradient + IFFT2 + crop from that scratch (lstsq.py:491-507):
 * chi = crop(IFFT2(F * gscale [* mode_scale on measured pixels])) * inv_scale
 * with F = fwd_scale * (column pass of `colin`) re-formed in registers, so the
 * far plane is neither written nor read.  mode_scale / measured as in
 * tike_ifft2_crop_scaled_modes (both may be NULL).  work (ntile,det,det) must
 * not alias colin; chi may alias work only when pw == det.  det = 256. */
int tike_grad_ifft2_crop(const void* colin, const float* gscale, const float* mode_scale,
                         const unsigned char* measured, int S, void* work, void* chi,
                         long ntile, int det, int pw, float fwd_scale, float inv_scale,
                         void* stream);

/* ---- the inverse transform split for the gradient pass (lstsq.py:504-539).
 * Pass 1 only: `work` (ntile,det,det) receives the INPUT of the inverse column
 * pass of every tile (rows 16 k + ya of fft_engine2.h) -- from the forward
 * kernel's scratch (tike_grad_ifft2_pass1, det = 256 or 512; operands as
 * tike_grad_ifft2_crop) or from a stored far plane times gscale
 * [* mode_scale on measured pixels] (tike_ifft2_pass1_scaled, det in
 * {128,256,512}; operands as tike_ifft2_crop_scaled_modes, mode_scale /
 * measured may be NULL).  work must not alias the input. */
/* The column pass + gradient factor + inverse pass 1 in ONE launch (det = 256
 * or 512; gaussian, or poisson without per-mode step lengths): what
 * tike_fwd_gradient_scale followed by tike_grad_ifft2_pass1 compute, without
 * the factor going through memory (a work item = (position, k1) sweeps the
 * hand-off rows of all S modes twice: for the intensity, then -- newest first --
 * for the gradient and the inverse's pass 1).  scratch (nscan,S,det,det) from
 * tike_fwd_pass1; data f32 or uint16; measured may be NULL; costs (nscan, may be
 * NULL) overwritten; work (nscan,S,det,det) != scratch receives the input of
 * tike_ifft2_pass2_gradients. */
int tike_fwd_grad_ifft2_pass1(const void* scratch, const void* data, int data_u16,
                              const unsigned char* measured, float* costs, void* work,
                              int nscan, int S, int det, float fwd_scale, int model,
                              float unmeasured_scaling, long num_measured, void* stream);
/* The same for the LAST slice of a multislice object (ptycho/solvers/rpie.py:
 * 444-472; det = 256, probe window = detector): the loop there hands
 * diff = FresnelSpectProp.adj(diff) (fresnelspectprop.py:100-113) to the slice
 * in front, b times for slice nslices - 1 - b.  With chi = IFFT2(G) that is
 * IFFT2(conj(H) FFT2(IFFT2(G))) = IFFT2(conj(H)^b G): the step's forward
 * transform cancels.  work (nslices,nscan,S,det,det): work[b] = the inverse's
 * pass 1 of conj(propagator)^b x G, each finished by tike_ifft2_pass2_products
 * of its slice; propagator (det,det) c64 in FFT order. */
int tike_fwd_grad_ifft2_pass1_slices(const void* scratch, const void* data, int data_u16,
                                     const unsigned char* measured, float* costs, void* work,
                                     int nscan, int S, int det, float fwd_scale, int model,
                                     float unmeasured_scaling, long num_measured,
                                     const void* propagator, int nslices, void* stream);
int tike_grad_ifft2_pass1(const void* colin, const float* gscale, const float* mode_scale,
                          const unsigned char* measured, int S, void* work, long ntile, int det,
                          float fwd_scale, void* stream);
int tike_ifft2_pass1_scaled(const void* farplane, const float* gscale, const float* mode_scale,
                            const unsigned char* measured, int S, void* work, long ntile, int det,
                            void* stream);

/* Pass 2 fused with both gradients, chi never stored (lstsq.py:504-539):
 *   chi_n,s  = inv_scale * (column pass of work[n][s])         (registers only)
 *   objproj[n]        = sum_s conj(P_n,s) chi_n,s     (lstsq.py:510-513)
 *   m_probe_update[s] += mpu_scale * sum_n conj(patches[n]) chi_n,s   (:531-539,
 *                        mpu_scale = 1 / num_batch is the division of :596)
 *   chi0[n]           = chi_n,0                       (:507; step sizes etc.)
 * P_n,s = the shared probe scaled by eigen_weights[n][0][s] plus, for the first
 * eigen_modes modes, the eigen probes (probe.py:272-303, applied on the fly from
 * LDS-resident slices: num_eigen * eigen_modes * det * 32 bytes <= 32 KiB, else
 * TIKE_ERR_UNSUPPORTED).  patches (nscan,det,det)
 * = O_n as stored by the forward kernels.  objproj / chi0 / m_probe_update may
 * each be NULL.  Probe window = detector (pw == det); det in {128,256,512};
 * S <= 8 (<= 4 at 512): TIKE_ERR_UNSUPPORTED otherwise -- use tike_ifft2_crop*
 * followed by tike_lstsq_gradients. */
int tike_ifft2_pass2_gradients(const void* work, const void* patches, const void* probe,
                               const void* eigen_probe, const float* eigen_weights,
                               int num_eigen, int eigen_modes, void* objproj, void* chi0,
                               void* m_probe_update, float mpu_scale, int nscan, int S, int det,
                               float inv_scale, void* stream);
/* ... with chi_n,s also times mode_scale[n][s] (nscan,S) f32: per-mode factors
 * that became known only after pass 1 of the inverse was written (the poisson
 * step lengths of tike_poisson_steps_grad_ifft2_pass1). */
int tike_ifft2_pass2_gradients_scaled(const void* work, const void* patches, const void* probe,
                                      const void* eigen_probe, const float* eigen_weights,
                                      int num_eigen, int eigen_modes, void* objproj, void* chi0,
                                      void* m_probe_update, float mpu_scale, int nscan, int S,
                                      int det, float inv_scale, const float* mode_scale,
                                      void* stream);
/* 1 where the eigen probes of a problem fit the LDS slices
 * tike_ifft2_pass2_gradients[_scaled|_modes] keep of them (num_eigen x
 * eigen_modes x det / 16 rows x 64 columns of complex64 in 32 KiB: 8 probe-mode
 * pairs at 128^2, 4 at 256^2, 2 at 512^2); 0: those entries return
 * TIKE_ERR_UNSUPPORTED and the caller keeps chi (tike_ifft2_crop* +
 * tike_lstsq_gradients, probe.py:272-303 on the fly).  No device work. */
int tike_ifft2_pass2_eigen_fits(int det, int num_eigen, int eigen_modes);
/* ... for MORE modes than one launch holds in registers (more than 8 at 128^2 /
 * 256^2, more than 4 at 512^2; the reference's cuFFT path, lstsq.py:504-539,
 * takes any number): the modes [mode0, mode0 + nmodes) of an S-mode problem,
 * 2 <= nmodes <= 8 (512^2: 4 without register spills) --
 * their probe gradients, mode 0 of chi when mode0 == 0, and their share of
 * objproj = sum_s conj(P_n,s) chi_n,s stored (accumulate == 0) or added to what
 * the launch of the modes in front left there.  Arguments as
 * tike_ifft2_pass2_gradients with S the mode count of the problem (objproj
 * NULL: the probe gradients and chi0 only); the eigen probes must all belong to
 * the modes of the first group. */
int tike_ifft2_pass2_gradients_modes(const void* work, const void* patches, const void* probe,
                                     const void* eigen_probe, const float* eigen_weights,
                                     int num_eigen, int eigen_modes, void* objproj, void* chi0,
                                     void* m_probe_update, float mpu_scale, int nscan, int S,
                                     int det, float inv_scale, int mode0, int nmodes,
                                     int accumulate, void* stream);

/* ---- far-plane gradient factor from the intensity (objective.py:31-44,97-109;
 * lstsq.py:491-502): gscale[n][p] = -(1 - sqrt(d)/(sqrt(I)+1e-9)) (gaussian) or
 * -(1 - d/(I+1e-9)) (poisson) on measured pixels, (unmeasured_scaling - 1)
 * elsewhere; costs[n] (optional) = per-pattern cost over measured pixels. */
int tike_gradient_scale(const float* intensity, const float* data, const unsigned char* measured,
                        float* gscale, float* costs, int nscan, int det, int model,
                        float unmeasured_scaling, long num_measured, void* stream);

/* ---- IFFT2 + crop of (farplane * gscale): the far-plane gradient is applied
 * while the rows are loaded (no separate read-modify-write pass).  gscale
 * (ntile / S, det, det) f32 is shared by the S modes of a position; work must
 * not alias farplane; det in {128, 256, 512}. */
int tike_ifft2_crop_scaled(const void* farplane, const float* gscale, int S, void* work,
                           void* chi, long ntile, int det, int pw, float scale, void* stream);

/* ---- poisson noise model (lstsq.py:454-489): per-(position, mode) step
 * lengths of exitwave.py:122-184 (all modes) or :187-234 (dominant_mode != 0:
 * one step per position, written to all S entries).  steps (nscan, S) f32;
 * farplane (nscan,S,det,det) scaled far-plane waves (unused for dominant mode);
 * intensity, data (nscan,det,det) f32; measured (det,det) u8 or NULL. */
int tike_poisson_steps(const void* farplane, const float* intensity, const float* data,
                       const unsigned char* measured, float* steps, int nscan, int S, int det,
                       float step_start, float weight, int dominant_mode, void* stream);

/* The same per-mode step lengths (all modes, exitwave.py:122-184) WITHOUT a
 * stored far plane, for the far-plane-free sizes (det 256 with S <= 8, 512 with
 * S <= 4; TIKE_ERR_UNSUPPORTED otherwise): two column passes over the forward
 * hand-off `scratch` of tike_fwd_pass1 with |F_s|^2 of all modes in registers.
 * The first pass also leaves what tike_fwd_gradient_scale(model = 1) leaves:
 * gscale (nscan,det,det) the poisson gradient factor, costs (nscan) or NULL.
 * steps (nscan,S) out; sums (nscan,S,2) f32 workspace.  The inverse that
 * follows is tike_grad_ifft2_pass1(scratch, gscale, steps, measured, ...). */
int tike_poisson_steps_handoff(const void* scratch, const void* data, int data_u16,
                               const unsigned char* measured, float* gscale, float* costs,
                               float* steps, float* sums, int nscan, int S, int det, float scale,
                               float unmeasured_scaling, long num_measured, float step_start,
                               float weight, void* stream);

/* The same step lengths AND the gradient pass, det = 256, S <= 8, for data
 * whose unmeasured pixels carry no gradient: `measured` NULL (every pixel
 * measured) or unmeasured_scaling == 1, the reference's default
 * (TIKE_ERR_UNSUPPORTED otherwise: tike_poisson_steps_handoff +
 * tike_grad_ifft2_pass1).  The far-plane gradient of mode s (exitwave.py:
 * 122-184, lstsq.py:454-502) is then steps[n][s] x (F_s x poisson factor) --
 * linear in the step length -- so the second sweep and the gradient pass are
 * ONE launch: work (nscan,S,det,det) != scratch receives pass 1 of the inverse
 * of F_s x factor x scale WITHOUT the step length, which
 * tike_ifft2_pass2_gradients_scaled applies (mode_scale = steps).  With 6-8
 * modes both sweeps run inside the resident gradient kernel (the hand-off is
 * read once per sweep).  costs (nscan) or NULL, steps (nscan,S) out, sums
 * (nscan,S,2) workspace; counts at unmeasured pixels may be NaN. */
int tike_poisson_steps_grad_ifft2_pass1(const void* scratch, const void* data, int data_u16,
                                        const unsigned char* measured, float* costs,
                                        float* steps, float* sums, void* work, int nscan, int S,
                                        int det, float scale, float unmeasured_scaling,
                                        long num_measured, float step_start, float weight,
                                        void* stream);

/* tike_ifft2_crop_scaled with the factor of mode s multiplied by
 * mode_scale[n][s] on measured pixels (lstsq.py:487-489: farplane[measured] =
 * -step_length * grad_cost). */
int tike_ifft2_crop_scaled_modes(const void* farplane, const float* gscale,
                                 const float* mode_scale, const unsigned char* measured, int S,
                                 void* work, void* chi, long ntile, int det, int pw, float scale,
                                 void* stream);

/* farplane[tile][p] *= mode_scale[tile] on measured pixels: the same factor
 * for detector sizes the fused inverse does not cover. */
int tike_scale_modes(void* farplane, const float* mode_scale, const unsigned char* measured,
                     long ntile, int det, void* stream);

/* ---- IFFT2 + crop to the probe window (propagation.py:59-73 followed by
 * lstsq.py:506-507 / convolution.py:108-110 crop).  work (ntile,det,det) holds
 * the intermediate and may alias farplane; chi (ntile,pw,pw) may alias work
 * only when pw == det. */
int tike_ifft2_crop(const void* farplane, void* work, void* chi, long ntile, int det, int pw,
                    float scale, void* stream);

/* ---- intensity, per-pattern cost and far-plane gradient in one pass
 * (ptycho.py:18-23; objective.py:11-124; lstsq.py:444-502).
 * model 0 = gaussian, 1 = poisson.  measured (det,det) uint8 mask or NULL
 * (all measured); num_measured = number of non-zero mask pixels.
 * intensity (nscan,det,det) and costs (nscan) are optional outputs (NULL to
 * skip).  With apply_gradient the farplane is overwritten by
 *   -grad on measured pixels, (unmeasured_scaling-1)*farplane elsewhere. */
int tike_farplane_gradient(void* farplane, const float* data, const unsigned char* measured,
                           float* intensity, float* costs, int nscan, int S, int det, int model,
                           int apply_gradient, float unmeasured_scaling, long num_measured,
                           void* stream);

/* ---- stand-alone objective helpers (operators/cupy/objective.py:18-124,
 * ptycho.py:18-23 _intensity_from_farplane).  farplane (nscan,S,npix) c64,
 * data / intensity (nscan,npix) f32, costs (nscan) = mean over npix. */
int tike_intensity(const void* farplane, float* intensity, long nscan, int S, long npix,
                   void* stream);
int tike_cost_each_pattern(const float* data, const float* intensity, float* costs, long nscan,
                           long npix, int model, void* stream);
int tike_objective_grad(const float* data, const void* farplane, const float* intensity,
                        void* out, long nscan, int S, long npix, int model, void* stream);

/* ==== lstsq_grad update loop (ptycho/solvers/lstsq.py), one minibatch ====
 * chi (nscan,S,pw,pw): exit-wave update = tike_ifft2_crop of the far-plane
 * gradient.  The probe arguments select P_n,s as in tike_ptycho_fwd. */

/* One pass over chi for both gradients (lstsq.py:506-539):
 *   m_probe_update (S,pw,pw) += sum_n conj(patch_n(psi)) * chi_n,s
 *   objproj (nscan,pw,pw)     = sum_s conj(P_n,s) * chi_n,s
 *   patches (nscan,pw,pw)     = patch_n(psi)            (the reference's bpatches)
 * any of the three outputs may be NULL.  S <= 16.  unique_probe (nscan,
 * eigen_modes,pw,pw), if not NULL, is the varying probe of the first
 * eigen_modes modes from tike_varying_probe (used instead of re-synthesising
 * it from eigen_probe / eigen_weights). */
int tike_lstsq_gradients(const void* chi, const float* scan, const void* psi, const void* probe,
                         const void* eigen_probe, const float* eigen_weights, int num_eigen,
                         int eigen_modes, const void* unique_probe, void* patches,
                         void* m_probe_update, void* objproj, int nscan, int S, int pw, int H,
                         int W, void* stream);

/* acc (2,H,W) f32, PLANAR (real plane, imaginary plane) += scatter_n( objproj_n ):
 * the adjoint of the bilinear patch gather (Patch.adj, patch.py:132-188) with
 * one atomic per object pixel and position.  Positions must satisfy
 * check_allowed_positions (position.py:600-628). 
 * Consecutive positions that are spatial neighbours (spread <= 112 px) are
 * summed on chip first, 8 at a time; any order gives the same sums. */
int tike_scatter_patches(const void* objproj, const float* scan, float* acc, int nscan, int pw,
                         int H, int W, void* stream);

/* m_probe_update (S,pw,pw) += sum_n conj(patch_n(psi)) * chi_n,s
 * (lstsq.py:524-539); patches (nscan,pw,pw), if not NULL, receives
 * patch_n(psi) (the reference's bpatches).  S <= 16. */
int tike_probe_grad(const void* chi, const float* scan, const void* psi, void* patches,
                    void* m_probe_update, int nscan, int S, int pw, int H, int W, void* stream);

/* out (H,W) f32 += scatter_n( probe_amp ), the (real-valued) object
 * preconditioner (solvers/_preconditioner.py:48-104 = Patch.adj of one
 * broadcast patch); probe_amp (pw,pw) f32 = sum_s |probe_s|^2
 * (_preconditioner.py:40-45).  Positions must satisfy
 * check_allowed_positions (position.py:600-628). */
int tike_psi_preconditioner(const float* probe_amp, const float* scan, void* out, int nscan,
                            int pw, int H, int W, void* stream);

/* the same with one real patch PER POSITION: out (H,W) f32 +=
 * scatter_n( amp[n] ), amp (nscan,pw,pw) f32 -- the illumination of a slice
 * behind the first one of a multislice object (_preconditioner.py:82-95). */
int tike_scatter_amplitudes(const float* amp, const float* scan, float* out, int nscan, int pw,
                            int H, int W, void* stream);

/* out (pw,pw) c64: real part += sum_n |patch_n(psi)|^2
 * (solvers/_preconditioner.py:116-167). */
int tike_probe_preconditioner(const float* scan, const void* psi, void* out, int nscan, int pw,
                              int H, int W, void* stream);

/* Per-position sums of the 2x2 step-size normal equations for mode 0
 * (lstsq.py:619-718) and of the eigen-probe intensity coefficients
 * (lstsq.py:721-738): stats (nscan, 8) f32 =
 * { sum|dOP|^2, sum|dPO|^2, Re/Im sum dOP conj(dPO), sum Re(conj(dOP) chi0),
 *   sum Re(conj(dPO) chi0), sum Re(conj(O P_0) chi0), sum|O P_0|^2 },
 * dOP = patch_n(object_update_precond) * P_n,0, dPO = m_probe_update[0] * O_n,
 * O_n = patch_n(psi), P_0 = shared probe mode 0.  object_update_precond and
 * m_probe_update may be NULL (that direction is then zero).  chi is laid out
 * (nscan, chi_modes, pw, pw) and only its mode 0 is read (chi_modes = 1 when
 * the caller kept just that mode). 
 * patches (nscan,pw,pw), if not NULL, holds patch_n(psi) as stored by
 * tike_lstsq_gradients and replaces the bilinear gather of psi.
 * eigen_proj (nscan) f32, if not NULL, receives sum Re(conj(R_n) E) with
 * R_n = conj(O_n) chi_n,0 - m_probe_update[0] and E = eigen0 (pw,pw) c64: the
 * first sum of tike_eigen_position_sums for the first eigen probe, for free. */
int tike_lstsq_step_stats(const void* chi, const float* scan, const void* psi,
                          const void* object_update_precond, const void* probe,
                          const void* eigen_probe, const float* eigen_weights, int num_eigen,
                          int eigen_modes, const void* unique_probe, const void* m_probe_update,
                          const void* patches, float* stats, int nscan, int S, int chi_modes,
                          int pw, int H, int W, const void* eigen0, float* eigen_proj,
                          void* stream);

/* out (nscan, eigen_modes, pw, pw) = weights[n][0][s]*probe[s] +
 * sum_c weights[n][c+1][s]*eigen[c][s]: the varying probe of the modes that
 * own eigen probes (ptycho/probe.py:272-303), synthesised once per chunk. */
int tike_varying_probe(const void* probe, const void* eigen_probe, const float* eigen_weights,
                       int num_eigen, int eigen_modes, void* out, int nscan, int S, int pw,
                       void* stream);

/* ==== eigen-probe ("OPR") update for mode 0 (ptycho/probe.py:362-476,
 * solvers/lstsq.py:297-364,740-761).  The residual
 *   R_n = conj(patches[n]) * chi0[n] - mpu0 - sum_{c'<c} coefs[n][c'] * eigen[c'][0]
 * is recomputed on the fly by both kernels.  patches, chi0 (nscan,pw,pw) c64;
 * mpu0 (pw,pw) c64 = m_probe_update mode 0; eigen_probe (C,Sm,pw,pw) c64;
 * coefs (nscan,C) c64 (unused entries >= c ignored; may be NULL when c == 0).
 * chi0 of position n starts at chi0 + n*chi_modes*pw*pw (chi_modes = 1 for a
 * packed mode-0 array, S to read mode 0 out of the full chi). */

/* sums (nscan,5) f32 = { sum Re(conj(R) E_c), sum Re(chi0 conj(O E_c)),
 * sum |O E_c|^2, Re sum R conj(E_c), Im sum R conj(E_c) } with E_c = eigen[c][0]. */
int tike_eigen_position_sums(const void* patches, const void* chi0, const void* mpu0,
                             const void* eigen_probe, const void* coefs, int num_eigen,
                             int eigen_modes, int c, float* sums, int nscan, int pw,
                             int chi_modes, void* stream);

/* update (pw,pw) c64 += sum_n R_n * pm[n]   (pm (nscan) f32). */
int tike_eigen_pixel_update(const void* patches, const void* chi0, const void* mpu0,
                            const void* eigen_probe, const void* coefs, int num_eigen,
                            int eigen_modes, int c, const float* pm, void* update, int nscan,
                            int pw, int chi_modes, void* stream);

/* ---- position correction (lstsq.py:545-579): per position, over the central
 * half of the probe window and for mode 0,
 *   numerator[n]   = ( sum Re(conj(gx P) chi), sum Re(conj(gy P) chi) )
 *   denominator[n] = ( sum |gx P|^2, sum |gy P|^2 )
 * with gx / gy the first-order Gaussian derivatives of the object patch along
 * rows / columns (position.py:779-810; `taps` = 2*radius+1 HOST floats t[d],
 * g[i] = sum_d t[d] x[i+d], edge mode 'nearest') and P the probe of that
 * position (shared, or varying as in tike_ptycho_fwd).  patches (nscan,pw,pw),
 * chi (nscan,chi_modes,pw,pw) c64; numerator, denominator (nscan,2) f32. */
int tike_position_sums(const void* patches, const void* chi, int chi_modes, const void* probe,
                       const void* eigen_probe, const float* eigen_weights, int num_eigen,
                       int eigen_modes, const float* taps, int radius, float* numerator,
                       float* denominator, int nscan, int S, int pw, void* stream);

/* ---- the chunk body of _get_nearplane_gradients for ANY shape (round 6;
 * ptycho/solvers/lstsq.py:422-579 = operators/cupy/ptycho.py:114-176 around
 * objective.py:31-44): probe window pw <= det, any number of modes S, any
 * detector size with a mixed-radix plan (det = 2^a 3^b 5^c 7^d 11^e 13^f <=
 * 4096).  Three launches on the LDS line engine; the zero padding, the far
 * plane and chi never exist in memory.  tike_gen_supported(S, pw, det) = 1
 * where the three entries run (lines of all S modes fit LDS), else 0 and they
 * return TIKE_ERR_UNSUPPORTED.
 *   hand1, hand2 (nscan, S, pw, det) c64: the rows of the probe window after
 *   the forward row transforms / before the inverse row transforms (unscaled).
 * The probe at position n as in tike_ptycho_fwd (shared; one per position;
 * shared + eigen probes x weights; `unique` (nscan, eigen_modes, pw, pw): the
 * varying modes already synthesised by tike_varying_probe). */
int tike_gen_supported(int S, int pw, int det);

/* K1 (convolution.py:58-101 + the row half of propagation.py:43-57):
 * hand1[n][s][y][:] = FFT_row( pad( patch_n(psi)[y][:] * probe_n[s][y][:] ) ),
 * patches (nscan, pw, pw) = patch_n(psi) if not NULL (gathered once per
 * position for all modes; pixels outside the image are zero). */
int tike_gen_fwd_rows(const void* psi, const float* scan, const void* probe, int probe_per_scan,
                      const void* unique, const void* eigen_probe, const float* eigen_weights,
                      int num_eigen, int eigen_modes, void* hand1, void* patches, int nscan, int S,
                      int pw, int det, int H, int W, void* stream);

/* K2 (the column half of propagation.py:43-73 around objective.py:11-124 and
 * lstsq.py:444-502): far plane F = fwd_scale * FFT_col(hand1 zero-padded),
 * intensity = sum_s |F_s|^2, costs[n] = mean over measured pixels of the
 * per-pixel cost (model 0 gaussian, 1 poisson), gradient factor g (measured
 * pixels; unmeasured_scaling - 1 elsewhere; counts at unmeasured pixels may be
 * NaN), hand2 = rows [pad, pad + pw) of IFFT_col(F * g), unscaled.  data
 * (nscan, det, det) f32; measured (det, det) u8 or NULL; hand2 NULL = costs
 * only. */
int tike_gen_cols_gradient(const void* hand1, const float* data, const unsigned char* measured,
                           float* costs, void* hand2, int nscan, int S, int pw, int det,
                           float fwd_scale, int model, float unmeasured_scaling,
                           long num_measured, void* stream);

/* K3 (the row half of propagation.py:59-73, convolution.py:103-154,
 * lstsq.py:504-539): chi[n][s] = inv_scale * crop(IFFT_row(hand2)) is formed
 * in LDS; objproj[n] = sum_s conj(probe_n[s]) chi[n][s] (input of
 * tike_scatter_patches), chi0[n] = chi[n][0], m_probe_update[s] +=
 * probe_update_scale * sum_n conj(patches[n]) chi[n][s] -- each NULL to skip.
 * objproj, chi0, patches (nscan, pw, pw) c64; m_probe_update (S, pw, pw) c64,
 * accumulated (one float atomic per pixel, mode and chunk of positions; under
 * tike_set_deterministic per-chunk partial sums added in order). */
int tike_gen_inv_rows_gradients(const void* hand2, const void* patches, const void* probe,
                                int probe_per_scan, const void* unique, const void* eigen_probe,
                                const float* eigen_weights, int num_eigen, int eigen_modes,
                                void* objproj, void* chi0, void* m_probe_update,
                                float probe_update_scale, int nscan, int S, int pw, int det,
                                float inv_scale, void* stream);

/* ---- the same chunk body for detector sizes det = p * M, p in {3, 5, 7}, M a
 * power of two in 32 .. 512 (96, 160, 192, 224, 320, 384, 448, 640, 768, 896,
 * 1536 ...) -- and for 1024 = 4 x 256 and 2048 = 4 x 512, where the step is a
 * Cooley-Tukey one (twiddles w_det^(n1 k2) applied by the combine entry) -- by the
 * prime-factor decomposition: p and M are coprime, so the det x det transform
 * is p x p sub-tiles of M x M through the power-of-two register engine plus a
 * pointwise p x p DFT across the sub-tiles, no twiddles between them
 * (csrc/pfa.hip).  tike_pfa_supported(S, pw, det) = 1 where these entries run.
 *   subtiles (nscan, S, p, p, M, M) c64: tile row y and column x live in
 *   sub-tile (y qM mod p, x qM mod p) at element (y qp mod M, x qp mod M),
 *   qM = M^-1 mod p, qp = p^-1 mod M; the far plane in sub-tile (k1y, k1x) at
 *   (k2y, k2x) is frequency (M qM k1 + p qp k2) mod det.
 * A chunk: tike_pfa_fwd_gather -> tike_pfa_fft2(forward) ->
 * tike_pfa_combine_gradient -> tike_pfa_fft2(inverse) -> tike_pfa_inv_products
 * -> tike_scatter_patches; probe arguments as in tike_gen_fwd_rows. */
int tike_pfa_supported(int S, int pw, int det);

/* convolution.py:58-101: subtiles = pad(patch_n(psi) * probe_n[s]) in the
 * sub-tile layout; patches (nscan, pw, pw) = patch_n(psi) if not NULL. */
int tike_pfa_fwd_gather(const void* psi, const float* scan, const void* probe,
                        int probe_per_scan, const void* unique, const void* eigen_probe,
                        const float* eigen_weights, int num_eigen, int eigen_modes,
                        void* subtiles, void* patches, int nscan, int S, int pw, int det, int H,
                        int W, void* stream);

/* propagation.py:43-73, the power-of-two part: the M x M transform of every
 * sub-tile of ntile tiles (unscaled); out must not alias in. */
int tike_pfa_fft2(const void* in, void* out, long ntile, int det, int inverse, void* stream);

/* tike_pfa_fwd_gather + the forward tike_pfa_fft2 in ONE launch where the
 * sub-tiles are 128 x 128 (det = 384, 640, 896; a shared probe with or without
 * eigen probes): a sub-tile is gathered, multiplied by the probe and
 * transformed inside the LDS of a CU and written once (the forward operator's
 * 128^2 kernel with the prime-factor map in front; lstsq.py:441-452 +
 * propagation.py:43-57).  subtiles (nscan,S,p,p,128,128) receives what the two
 * launches leave in their output; probe_scratch ((S + num_eigen x
 * eigen_modes) det^2 c64) is rewritten by every call (the probe and the eigen
 * probes in the sub-tile layout).  tike_pfa_fwd_subtiles_supported: 1 where it
 * serves, no device work. */
int tike_pfa_fwd_subtiles_supported(int S, int pw, int det);
int tike_pfa_fwd_subtiles(const void* psi, const float* scan, const void* probe,
                          const void* eigen_probe, const float* eigen_weights, int num_eigen,
                          int eigen_modes, void* probe_scratch, void* subtiles, void* patches,
                          int nscan, int S, int pw, int det, int H, int W, void* stream);

/* The p x p DFT across the sub-tiles completes the far plane F (x fwd_scale);
 * objective.py:11-124 + lstsq.py:444-502 on it (costs, gradient factor, as
 * tike_gen_cols_gradient); apply_gradient: F x factor goes back through the
 * inverse p x p DFT, in place (input of tike_pfa_fft2(inverse)). */
int tike_pfa_combine_gradient(void* subtiles, const float* data, const unsigned char* measured,
                              float* costs, int nscan, int S, int det, float fwd_scale, int model,
                              float unmeasured_scaling, long num_measured, int apply_gradient,
                              void* stream);

/* convolution.py:103-154 + lstsq.py:504-539 on chi = inv_scale * (inverse
 * transform, cropped), read through the sub-tile map: outputs as
 * tike_gen_inv_rows_gradients. */
int tike_pfa_inv_products(const void* subtiles, const void* patches, const void* probe,
                          int probe_per_scan, const void* unique, const void* eigen_probe,
                          const float* eigen_weights, int num_eigen, int eigen_modes,
                          void* objproj, void* chi0, void* m_probe_update,
                          float probe_update_scale, int nscan, int S, int pw, int det,
                          float inv_scale, void* stream);

/* ---- the stages of a multislice object, fused (operators/cupy/multislice.py:
 * 69-92,144-194 = Convolution + FresnelSpectProp slice by slice;
 * fresnelspectprop.py:52-113; ptycho/solvers/rpie.py:367-495).  A Fresnel
 * step FFT2 -> x propagator -> IFFT2 runs as
 *   pass 1 (tike_fwd_pass1 with the incident probes as per-position probes:
 *   patch x probe formed on the fly; or tike_fft2_pass1 on a stored wave)
 *   -> tike_fresnel_colpass -> inverse pass 2 (tike_fft2_pass2_inplace, or
 *   tike_ifft2_pass2_products on the way back),
 * three launches with two hand-offs instead of four transforms and a multiply.
 * Tiles are det x det c64, det in {128, 256, 512} unless noted. */

/* pass 1 of the two-pass transform on plain tiles: rows + radix-16 column
 * stage; out (ntile,det,det) must not alias in. */
int tike_fft2_pass1(const void* in, void* out, long ntile, int det, int inverse, void* stream);

/* pass 2 alone, in place, every element times `scale`. */
int tike_fft2_pass2_inplace(void* tiles, long ntile, int det, int inverse, float scale,
                            void* stream);

/* The last pass of a Fresnel step fused with the first pass of the next
 * slice's transform (operators/cupy/multislice.py:86-91, then :79-85 of the
 * slice behind; probe window = detector): wave (nscan,S,det,det)
 * in = tike_fresnel_colpass's output, out = the probe incident on the slice
 * (x scale), in place; farplane1 (nscan,S,det,det) != wave receives pass 1 of
 * FFT2(wave x patch_n(psi)) -- the input of the column pass kernels
 * (tike_fresnel_colpass, tike_fwd_grad_ifft2_pass1[_slices]).  psi (H,W). */
int tike_slice_step(void* wave, const void* psi, const float* scan, void* farplane1, int nscan,
                    int S, int det, int H, int W, float scale, void* stream);

/* pass 2 of the S modes of every position with the illumination
 * amplitude[n] = sum_s |wave[n][s] * scale|^2 formed from the registers
 * (ptycho/solvers/_preconditioner.py:40-45,86-95: _probe_amp_sum of the probe
 * propagated to a slice).  tiles (nscan,S,det,det) c64: the finished wave is
 * written back in place iff `keep`, otherwise tiles is left half-transformed
 * (scratch); amplitude (nscan,det,det) f32. */
int tike_fft2_pass2_intensity(void* tiles, float* amplitude, long nscan, int S, int det,
                              int inverse, float scale, int keep, void* stream);

/* forward column pass -> x propagator (conj when `adjoint`) x scale -> inverse
 * pass 1 (fresnelspectprop.py:86-113): colin = output of a FORWARD pass 1,
 * work = input of an INVERSE pass 2; propagator (det,det) c64 in FFT order
 * (fresnelspectprop.py:115-137).  det in {128, 256, 512} (TIKE_ERR_UNSUPPORTED
 * otherwise: use tike_fresnel_spect_prop). */
int tike_fresnel_colpass(const void* colin, const void* propagator, int adjoint, void* work,
                         long ntile, int det, float scale, void* stream);

/* inverse pass 2 in place fused with the numerators of one slice
 * (rpie.py:444-472): with chi = inv_scale * (pass 2 of work),
 *   objproj[n]       = sum_s conj(probe[n|0][s]) * chi[n][s]      (nscan,det,det)
 *   probe_numerator += numerator_scale * sum_n conj(patch_n(psi)) * chi[n][s]
 *                                                   (S,det,det) c64, may be NULL
 *   keep_chi != 0: work <- chi (all modes: the wave the Fresnel step back to
 *                  the slice in front transforms); else chi0[n] <- chi[n][0]
 *                  (nscan,det,det; may be NULL) and work is left as it was.
 * psi (H,W) is the slice; scan as in tike_ptycho_adj.  S <= 8. */
int tike_ifft2_pass2_products(void* work, const void* psi, const float* scan, const void* probe,
                              int probe_per_scan, void* objproj, void* probe_numerator,
                              float numerator_scale, void* chi0, int keep_chi, int nscan, int S,
                              int det, int H, int W, float inv_scale, void* stream);

/* ---- FresnelSpectProp.fwd / .adj (operators/cupy/fresnelspectprop.py:52-113):
 * out = IFFT2(FFT2(in) * propagator) (adjoint != 0: conj(propagator)) for ntile
 * n x n tiles sharing one (n,n) c64 propagator; fwd_scale / inv_scale are the
 * two transforms' normalisations (1/n each for 'ortho').  in may equal out. */
int tike_fresnel_spect_prop(const void* in, void* out, const void* propagator, long ntile, int n,
                            int adjoint, float fwd_scale, float inv_scale, void* stream);

/* ---- small fused kernels of the lstsq_grad host loop (psi-, probe- or
 * (positions,)-sized work between the heavy kernels; sums that span all ranks
 * stay in small device buffers the caller all-reduces between two phases) ---- */

/* _precondition_object_update (lstsq.py:605-616) straight from the scatter
 * accumulator: g = acc (2,npix planar re/im f32);
 * upd_precond = g / sqrt(((1-alpha) Re precond)^2 + (alpha pmax[0])^2);
 * upd_sum = g as complex64 (may be NULL); combined (2,npix planar, may be NULL)
 * += g (the 'compact' epoch sum, lstsq.py:174).  upd_precond may be NULL. */
int tike_object_update_precond(const float* acc, const void* precond, const float* pmax,
                               float alpha, void* upd_sum, void* upd_precond, float* combined,
                               long npix, void* stream);

/* 2x2 step-length systems of a minibatch (lstsq.py:641-718) from the (B,8)
 * table of tike_lstsq_step_stats.  tike_lstsq_step_sums: sums[0..2] = { sum(A1 +
 * eps), sum(A4 + eps), sum(costs) } over the B local positions (costs may be
 * NULL).  tike_lstsq_step_solve, given the sums and count over ALL ranks:
 * out[0..1] = { sum 0.9 max(0, Re x1), sum 0.9 max(0, Re x2) } over the local
 * positions and out[2..4] = { out0/count, out1/count, sums2/count } (beta_object,
 * beta_probe, mean cost when one rank holds the whole minibatch). */
int tike_lstsq_step_sums(const float* stats, const float* costs, int B, float eps, float* sums,
                         void* stream);
int tike_lstsq_step_solve(const float* stats, int B, float eps, const float* sums, double count,
                          int recover_psi, int recover_probe, float* out, void* stream);

/* probe += beta[0] * mpu; combined += beta[0] * mpu * inv_num_batch (combined may
 * be NULL) over n complex elements (lstsq.py:177-181); beta is a device scalar. */
int tike_probe_update(void* probe, void* combined, const void* mpu, const float* beta,
                      float inv_num_batch, long n, void* stream);

/* Eigen-probe ("OPR") bookkeeping of a minibatch, mode m (lstsq.py:297-364,721-761;
 * probe.py:362-476).  weights: the minibatch's rows of eigen_weights (B, C+1, S).
 * tike_eigen_weights0: weights[n][0][m] += 0.1 stats[n][6] / stats[n][7];
 *   norms[c-1] = sum_n weights[n][c][m]^2 for c = 1..C (local; all-reduce).
 * tike_eigen_proj_mean: pm[n] = (first[n*first_stride] / P + weights_c[n*row]) / norm[0].
 * tike_eigen_normalise: E <- E + beta u / mnorm(u), u = update / count, then
 *   E <- E / mnorm(E); esum[0] = sum |E|^2 (may be NULL); work: 4 floats of device
 *   scratch (the sums both norms follow from).
 * tike_eigen_dsum: dsum[0] = sum_n sums[n][2] / P (local; all-reduce).
 * tike_eigen_weights: weights_c[n*row] += (sums[n][1]/P) / (sums[n][2]/P + 0.1
 *   dsum[0] / count); coefs_c[n*coef_stride] = (sums[n][3] + i sums[n][4]) / esum[0]
 *   (coefs_c may be NULL).  sums: (B,5) from tike_eigen_position_sums. */
int tike_eigen_weights0(float* weights, const float* stats, int B, int C, int S, int m,
                        float* norms, void* stream);
int tike_eigen_proj_mean(const float* first, int first_stride, const float* weights_c,
                         long weights_row, const float* norm, long P, int B, float* pm,
                         void* stream);
int tike_eigen_normalise(void* eigen, const void* update, double count, float beta, int npix,
                         float* esum, float* work, void* stream);
int tike_eigen_dsum(const float* sums, int B, long P, float* dsum, void* stream);
int tike_eigen_weights(const float* sums, int B, long P, const float* dsum, double count,
                       float* weights_c, long weights_row, void* coefs_c, int coef_stride,
                       const float* esum, void* stream);

/* ---- conjugate direction of the conjugate-gradient solver on the device
 * (reference opt.py:281-301 `direction_dy`, Dai-Yuan, as solvers/cgrad.py
 * composes it with the gradient buffers of tike_lstsq_chunk_gradients):
 *   g1 = -update        update = the accumulated descent direction of the
 *                       minibatch, either planar (2, n) float32 (the object
 *                       accumulator) or n interleaved complex64 (the probe's
 *                       m_probe_update) -- exactly one of the two is given
 *   first != 0:  direction = -g1
 *   otherwise:   direction = -g1 + direction |g1|^2 / (sum conj(direction)
 *                (g1 - gradient) + 1e-32)     (gradient = g1 of the last call)
 *   gradient <- g1
 *   first != 0 and costs given: state[0] = sum(costs[0..ncost)) / count -- the
 *                mean cost at x that tike_cgrad_line_search starts from.
 * gradient, direction: n complex64, updated in place; sums: 4 doubles of
 * scratch (zeroed by the call).  Two kernels, no host synchronisation. */
int tike_cgrad_direction(const float* update_planar, const void* update_complex, void* gradient,
                         void* direction, long n, int first, const float* costs, int ncost,
                         double count, double* state, double* sums, void* stream);

/* ---- backtracking line search decided on the device (the conjugate-gradient
 * solver; reference opt.py:216-278 `line_search` over the gaussian cost
 * `Ptycho.cost`, ptycho.py:193-204, as solvers/cgrad.py composes them).
 * Tries x + step d, x + step/2 d, ... for at most nslots step lengths; every
 * trial is a cost-only forward pass (tike_fwd_pass1 + tike_fwd_gradient_scale
 * over the minibatch in chunks of `chunk` positions), all enqueued at once: a
 * trial whose predecessor was accepted returns immediately, so the host reads
 * nothing back between trials.  det in {128, 256, 512} (128: float32 data; the
 * trial is tike_ptycho_fwd + the cost of tike_farplane_gradient), probe window =
 * detector, every pixel measured.
 *   variable  0: x, d, xs are the object (H,W); other = probe (S,det,det)
 *             1: x, d, xs are the probe (S,det,det); other = object (H,W)
 *   state     device double[5] = { fx, step, done, trials, failures }: on entry
 *             the mean cost at x and the first step length; on return,
 *             accepted: the mean cost and step length accepted, done = 1 (xs =
 *             the new iterate); not accepted: done = 0, step = the next length
 *             to try, failures += 1; trials counts the cost evaluations made
 *   count     positions over all ranks (the mean's denominator x det^2 is the
 *             kernels'); skip: one device int of scratch; costs (nscan) f32 and
 *             scratch (chunk,S,det,det) c64 workspaces. */
int tike_cgrad_line_search(int variable, const void* x, const void* d, void* xs,
                           const void* other, const float* scan, const void* data,
                           int data_u16, void* scratch, float* costs, int nscan, int chunk,
                           int S, int det, int H, int W, float fwd_scale, double count,
                           double* state, int* skip, int nslots, void* stream);

/* ---- the same search with every step length evaluated in one pass.  The far
 * plane is linear in the variable the search moves along: F(x + s d) = F(x) +
 * s F(d), F(d) being the forward model with the direction in place of the object
 * (variable 0) or of the probe (variable 1).  One forward pass 1 of the
 * direction and one column pass over the two hand-offs give the gaussian costs
 * of x and of x + step d, x + step/2 d, ... (16 step lengths, 8 per pass) from the per-pixel
 * quadratic sum_m |A_m|^2 + 2 s sum_m Re(conj(A_m) B_m) + s^2 sum_m |B_m|^2; one
 * small kernel then takes the decision of opt.py:216-278 (the first of those
 * lengths whose cost is no larger than the cost at x) and xs = x + step d is
 * formed with the accepted step (xs = x when none was).  Candidates, rule and
 * state as tike_cgrad_line_search, results equal up to float32 rounding; the
 * cost at x that decides is the one formed here (state[0] on entry is ignored).
 *   far_a   (chunk,S,det,det) c64: with a_valid != 0 and nscan <= chunk it holds
 *           what the gradient pass at x left in its `scratch` (the forward
 *           hand-off of x; at 128^2 the far plane of x) and is read as it is;
 *           otherwise workspace (x's forward pass is made here).
 *   far_b   (chunk,S,det,det) c64 workspace; costs_k 17 * nscan + 1 f32 of workspace.
 * Two passes of 8 step lengths are enqueued; the second returns at once when the
 * first has accepted a step.
 *   stage 0: the whole search (one rank; sums unused).  Several ranks -- the cost
 *   sums must be all-reduced between a cost pass and its decision; count is the
 *   number of positions over all ranks --: stage 1 = first cost pass, leaving this
 *   rank's row sums in sums (17 doubles); [all-reduce sums]; 2 = first decision
 *   from sums; 3 = second cost pass -> sums; [all-reduce]; 4 = second decision,
 *   then xs.  The same arguments in every stage. */
int tike_cgrad_line_search_linear(int variable, const void* x, const void* d, void* xs,
                                  const void* other, const float* scan, const void* data,
                                  int data_u16, void* far_a, int a_valid, void* far_b,
                                  float* costs_k, int nscan, int chunk, int S, int det, int H,
                                  int W, float fwd_scale, double count, double* state,
                                  int stage, double* sums, void* stream);

/* ---- the packed minibatch tail: the arithmetic of tike_lstsq_step_sums / _solve,
 * tike_probe_update and the tike_eigen_* entries above for the common case of ONE
 * eigen probe (or none), in four launches after the step statistics and with two
 * small all-reduces between them when several ranks share a minibatch
 * (lstsq.py:136-205,297-364,641-761; probe.py:362-476):
 *
 *   tike_lstsq_step_stats                         stats, eigen_proj
 *   tike_eigen_pixel_update1                      update += sum_n R_n pm[n]; sums3
 *     [all-reduce { sums3 ; update }]
 *   tike_lstsq_tail_mid                           tail3[0..1]; eigen0 <- E'
 *   tike_eigen_position_sums1                     sums5, tail3[2] (dsum)
 *     [all-reduce tail3]
 *   tike_lstsq_tail_finish                        weights, probe, steps
 * (no eigen probe: tike_lstsq_step_sums in place of tike_eigen_pixel_update1, no
 * tike_eigen_position_sums1).
 *
 * tike_eigen_pixel_update1: as tike_eigen_pixel_update for the first eigen probe
 *   with pm[n] = (eigen_proj[n] / P + weights_c[n*weights_row]) / norm[0] formed on
 *   the fly (eigen_proj from tike_lstsq_step_stats; norm = sum over ALL ranks of the
 *   minibatch's weights_c^2); eigen0 (pw,pw) c64: that eigen probe.  With sums3 not
 *   NULL one extra workgroup also leaves sums3[0..2] = tike_lstsq_step_sums(stats,
 *   costs, nscan, eps): the two share one all-reduce.
 *   Both eigen entries take psi (H,W) c64 and scan (nscan,2), or NULL: given, the
 *   object patches O_n of interior positions are recomputed from psi (the taps and
 *   arithmetic of tike_fwd_pass1 -- psi must still be the array the stored patches
 *   were gathered from) instead of being streamed from `patches`: half the HBM
 *   bytes of the pass.
 * tike_lstsq_tail_mid: the 2x2 solves of the B local positions with sums3 and
 *   count over all ranks: tail3[0..1] = { sum 0.9 max(0, Re x1), sum 0.9 max(0, Re
 *   x2) }; and, eigen0 not NULL, nacc[0..2] += { sum |update|^2, sum |eigen0|^2,
 *   sum Re(conj(eigen0) update) } over npix pixels (nacc zero on entry), then
 *   eigen0 <- normalise(eigen0 + beta_eigen u / mnorm(u)), u = update / count
 *   (tike_eigen_normalise, probe.py:440-448).
 * tike_eigen_position_sums1: sums5 (nscan,5) as tike_eigen_position_sums for the
 *   first eigen probe, and dsum[0] += sum_n sums5[n][2] / P (zero on entry).
 * tike_lstsq_tail_finish, with tail3 = { sum step_o, sum step_p, dsum } over all
 *   ranks: steps[0..4] = { tail3[0], tail3[1], tail3[0]/count, tail3[1]/count,
 *   sums3[2]/count }; probe += beta_probe mpu and combined += beta_probe mpu *
 *   inv_num_batch over nprobe elements (probe NULL: skipped; combined may be NULL);
 *   weights (B,C+1,S): [n][0][m] += 0.1 stats[n][6] / stats[n][7] and, with sums5
 *   (npix = P), [n][1][m] += (s1/P) / (s2/P + 0.1 dsum / count) (weights NULL:
 *   skipped). */
int tike_eigen_pixel_update1(const void* patches, const void* chi0, const void* mpu0,
                             const void* eigen0, const float* eigen_proj,
                             const float* weights_c, long weights_row, const float* norm,
                             void* update, int nscan, int pw, int chi_modes, const float* stats,
                             const float* costs, float eps, float* sums3, const void* psi,
                             const float* scan, int H, int W, void* stream);
int tike_lstsq_tail_mid(void* eigen0, const void* update, int npix, float* nacc,
                        float beta_eigen, const float* stats, int B, float eps,
                        const float* sums3, double count, int recover_psi, int recover_probe,
                        float* tail3, void* stream);
int tike_eigen_position_sums1(const void* patches, const void* chi0, const void* mpu0,
                              const void* eigen0, float* sums5, float* dsum, int nscan, int pw,
                              int chi_modes, const void* psi, const float* scan, int H, int W,
                              void* stream);
int tike_lstsq_tail_finish(const float* tail3, const float* sums3, double count, float* steps,
                           void* probe, void* combined, const void* mpu, float inv_num_batch,
                           long nprobe, float* weights, long weights_row, int S, int m,
                           const float* stats, const float* sums5, int B, int npix,
                           void* stream);

/* ---- the chunk body of _get_nearplane_gradients (lstsq.py:422-579) in ONE call:
 * the far-plane-free pipeline for a chunk of nscan positions,
 *   tike_fwd_pass1 -> tike_fwd_grad_ifft2_pass1 (256^2; at 512^2
 *   tike_fwd_gradient_scale -> tike_grad_ifft2_pass1) ->
 *   tike_ifft2_pass2_gradients -> tike_scatter_patches,
 * for callers that do not need the stages separately (a level-B binding of
 * lstsq.py's chunk loop).  Inputs as in those entries (psi, scan, probe, the
 * eigen probes applied on the fly, data f32 or uint16, optional mask);
 * workspaces scratch and work (nscan,S,det,det) c64 each and gscale
 * (nscan,det,det) f32 (512^2 only; may be NULL at 256^2, where the factor
 * stays in registers) are the caller's, must not alias and hold no result
 * afterwards.  Outputs: patches, chi0 (nscan,det,det) c64 and costs (nscan)
 * are overwritten; m_probe_update (S,det,det) c64 += mpu_scale * sum_n ...,
 * object_acc (2,H,W) planar f32 += scatter_n(sum_s conj(P_n,s) chi_n,s) via
 * objproj (nscan,det,det) c64 (overwritten); object_acc / objproj may both be
 * NULL (probe gradient only), m_probe_update and chi0 may be NULL.
 * Probe window = detector; det = 256 with S <= 8, or det = 512 with S <= 4;
 * model 0 gaussian / 1 poisson (without per-mode step lengths).
 * det = 128 with S <= 8 (float32 data, no eigen probes) runs the pipeline that
 * keeps the far plane at that size (tike_ptycho_fwd_intensity ->
 * tike_gradient_scale -> tike_ifft2_pass1_scaled -> tike_ifft2_pass2_gradients ->
 * tike_scatter_patches); gscale must then hold 2 x (nscan,det,det) f32.
 * TIKE_ERR_UNSUPPORTED otherwise -- compose the stages yourself. */
int tike_lstsq_chunk_gradients(const void* psi, const float* scan, const void* probe,
                               const void* eigen_probe, const float* eigen_weights,
                               int num_eigen, int eigen_modes, const void* data, int data_u16,
                               const unsigned char* measured, int model,
                               float unmeasured_scaling, long num_measured, void* scratch,
                               void* work, float* gscale, void* patches, float* costs,
                               void* objproj, void* chi0, void* m_probe_update,
                               float mpu_scale, float* object_acc, int nscan, int S, int det,
                               int H, int W, float fwd_scale, float inv_scale, void* stream);

/* ---- collectives: the per-minibatch gradient all-reduce over RCCL / xGMI,
 * one process (or thread) per GPU.  Replaces the serial peer-copy reduction of
 * communicators/pool.py:300-395 (reduce_gpu / allreduce) as composed by
 * communicators/comm.py:96-136.  librccl.so.1 is bound at the first call;
 * TIKE_ERR_UNSUPPORTED if it cannot be loaded; a failed RCCL call returns
 * TIKE_ERR_COMM + its ncclResult_t.
 * tike_comm_unique_id: one rank fills a HOST buffer of TIKE_COMM_ID_BYTES and
 *   hands it to the others out of band (file, socket, MPI, torch store).
 * tike_comm_create: collective over the nranks callers, each on its own device
 *   (blocks until all have joined); *comm is the opaque handle of the others.
 * tike_comm_allreduce_sum: buf[0..count) of float32 (f64 = 0) or float64
 *   (f64 = 1) summed over the ranks IN PLACE, asynchronously on `stream`;
 *   complex64 arrays are reduced as 2*numel float32.
 * tike_comm_broadcast: nbytes of rank `root`'s buf to every rank's buf. */
int tike_comm_unique_id(void* id);
int tike_comm_create(const void* id, int nranks, int rank, void** comm);
int tike_comm_destroy(void* comm);
int tike_comm_allreduce_sum(void* comm, void* buf, long count, int f64, void* stream);
int tike_comm_broadcast(void* comm, void* buf, long nbytes, int root, void* stream);

/* ---- host side: the inner loops of the minibatch selectors (no GPU).
 * tike_cluster_farthest_fill: the round robin of cluster.py:352-377 and
 *   :445-462 (wobbly_center, wobbly_center_random_bootstrap).  points (n,2)
 *   f32; owner (n) i64 in/out, -1 = free.  For turn = 0 .. turns-1 the cluster
 *   turn % num_cluster claims the free point farthest from the float32 mean
 *   of its members (first index among equals), exactly as the NumPy
 *   expressions of the reference evaluate it.
 * tike_cluster_swap_sweep: one sweep of the swap refinement of cluster.py:
 *   568-626 (compact).  dist (n,k) f64 distances to the centroids; label (n)
 *   i64 in/out; best (n) i64 nearest centroid; order (n) i64 the sweep order
 *   (ascending regret); regret (n) f64 in/out; *moved = 1 if any pair was
 *   exchanged. */
int tike_cluster_farthest_fill(const float* points, long n, long* owner, int num_cluster,
                               long turns);
int tike_cluster_swap_sweep(const double* dist, long n, int k, long* label, const long* best,
                            const long* order, double* regret, int* moved);

#ifdef __cplusplus
}
#endif
#endif /* TIKE_AMD_H */
