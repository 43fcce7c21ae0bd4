/* videoitg_hip_profiling.h - entry points that exist ONLY in libvideoitg_hip_prof.so (videoitg_amd/csrc/build.sh --profiling:
 * -DVITG_PROFILING_BUILD=1 + gemm_w4.hip, gemm_wf.hip, gemm_pe.hip).  Research scaffolding of round 1, kept for the
 * measurements in profiles/r01_gemm_shapes.md; the product library (include/videoitg_hip.h) does not contain them and the
 * product path never loads this library.
 *
 * Additional vitg_set_option keys in the profiling library:
 *   gemm_cfg    1 256x256x32 ring-4 | 3 256x256x64 ring-2 | 4 persistent | 5 asm LDS reads | 7 phased, 2 k-steps per interval |
 *               8 four waves x 128x128 (AGPR accumulators) | 11 register-staged operands | 13 persistent blocks, deferred epilogue
 *   gemm_flags  bit0 non-temporal stores, bit5 N-split launch, bit7 5-stage ring for cfg 8, bit20/21 auto mode uses cfg 13;
 *               bits 1-4, 6, 9, 15-19, 22-23: ablations that produce WRONG results (additionally need VITG_PROFILING=1)
 *   gemm_stagger  persistent kernel (cfg 4): de-phase the XCDs
 */
#ifndef VIDEOITG_HIP_PROFILING_H
#define VIDEOITG_HIP_PROFILING_H
#include "videoitg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Static-weight fast path of vitg_gemm.  Weights never change between forwards, so they can be stored a second time in the order
 * the MFMA consumes them: vitg_pack_weight_fragments writes, for every (32-row weight tile, 16-deep k-step), one contiguous 1-KiB
 * block whose 16-byte chunk l is W[tile*32 + (l & 31)][kstep*16 + 8*(l >> 5) .. +7] (rows >= N zero; N rounded up to 64).
 * vitg_gemm_wf then fetches whole fragments with coalesced global loads and keeps the weight operand out of LDS
 * (videoitg_amd/csrc/gemm_wf.hip).  Same contract, same results as vitg_gemm (same MFMA, same accumulation order); shapes the
 * fast path does not cover (M or N < 256) run vitg_gemm on W.  Wf may be NULL (= vitg_gemm).  Same reference call sites as
 * vitg_gemm: the weights are the nn.Linear parameters of [3P] modeling_siglip / modeling_qwen2 and mlp_proj.py:31-35. */
int vitg_weight_fragments_bytes(int N, int K, long* bytes);
int vitg_pack_weight_fragments(const void* W, int ldw, int N, int K, void* out, void* stream);
int vitg_gemm_wf(const void* A, int lda, const void* W, int ldw, const void* Wf, int M, int N, int K, int dtype, int mode,
                 const float* bias, int act, void* out, int ldo, const float* res, int ldr, int res_mod, void* stream);

/* (round 2; measured SLOWER than the stand-alone norm kernels on every shape of the step - profiles/r02_fused_norm.md - and therefore not
 * in the product library)  LayerNorm / RMSNorm folded into the GEMMs on either side of it: the stand-alone norm pass (read the fp32 residual stream, write a
 * 16-bit copy) disappears.  With y = norm(x) = (x - mean) * rstd * gamma + beta feeding a Linear W (+ bias b):
 *     W y + b  =  rstd * ( W (gamma * x)  -  mean * s )  +  c,      s[n] = sum_k gamma[k] W[n,k],   c = W beta + b   (both fp32, load time)
 * PRODUCER (vitg_gemm_norm_producer) = vitg_gemm(mode F32, act none, res given) that additionally writes xh = round16(out * next_gamma)
 * and, per row and per group of `cols` output columns (vitg_gemm_norm_plan; 0 = shape not covered, use the stand-alone norm),
 * the partial statistics (mean, M2) of out: partials[m][N / cols][2].  vitg_norm_stats_finalize combines them (Chan's formula) into
 * stats[m] = (mean, rstd) - for rms != 0: (0, rsqrt(mean(x^2) + eps)).  CONSUMER (vitg_gemm_norm_consumer) = vitg_gemm(mode STORE16 /
 * SWIGLU) on A = xh with out = epilogue(rstd[m] * (acc - mean[m] * colsum[n]) + bias[n]); colsum = s (LayerNorm) or NULL (RMSNorm),
 * bias = c.  Replaces [3P] SigLIP layer_norm1/2 and Qwen2RMSNorm together with the Linear layers around them
 * (clip_encoder.py:146, grounding_qwen2.py:135). */
int vitg_gemm_norm_plan(int M, int N, int K, int* cols_per_partial);
int vitg_gemm_norm_producer(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias, float* out,
                            int ldo, const float* res, int ldr, int res_mod, const float* next_gamma, void* xh, int ldxh,
                            float* partials, void* stream);
int vitg_norm_stats_finalize(const float* partials, long M, int P, int cols, float eps, int rms, float* stats, void* stream);
int vitg_gemm_norm_consumer(const void* XH, int lda, const void* W, int ldw, int M, int N, int K, int dtype, int mode, const float* bias,
                            int act, void* out, int ldo, const float* stats, const float* colsum, void* stream);

/* Timeline of the 256-tile GEMM kernel.  While a buffer is set, waves 0 and 4 (one per wave group) of the first `blocks` blocks of
 * every vitg_gemm launch on that kernel write the 100 MHz real-time counter into buf[((block*2 + group)*4 + tile)*32 + slot] for their
 * first four tiles: slot 0 tile start, 1 first K-tile landed, 2+kt end of K-tile kt (kt < 24), 28 epilogue begins, 29 epilogue stores
 * issued, 30 tile done, 31 shader-clock counter at tile start.  buf = NULL switches it off.  (tools/gemm_timeline.py) */
int vitg_gemm_timeline_buffer(void* buf, int blocks);

#ifdef __cplusplus
}
#endif
#endif
