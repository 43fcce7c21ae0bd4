/* videoitg_hip_profiling.h - entry points that exist ONLY in libvideoitg_hip_prof.so (videoitg_amd/csrc/build.sh --profiling:
 * -DVITG_PROFILING_BUILD=1 + gemm_w4.hip, gemm_wf.hip, gemm_pe.hip).  Research scaffolding of round 1, kept for the
 * measurements in profiles/r01_gemm_shapes.md; the product library (include/videoitg_hip.h) does not contain them and the
 * product path never loads this library.
 *
 * Additional vitg_set_option keys / values in the profiling library (environment: VITG_<KEY IN CAPS>):
 *   gemm_cfg    1 256x256x32 ring-4 | 3 256x256x64 ring-2 | 4 persistent | 5 asm LDS reads | 7 phased, 2 k-steps per interval |
 *               8 four waves x 128x128 (AGPR accumulators) | 11 register-staged operands | 13 persistent blocks, deferred epilogue
 *   gemm_flags  bit0 non-temporal stores, bit5 N-split launch, bit7 5-stage ring for cfg 8, bit20/21 auto mode uses cfg 13;
 *               bits 1-4, 6, 9, 15-19, 22-23: ablations that produce WRONG results (additionally need VITG_PROFILING=1);
 *               bit8 wide tiles off, bits 11-14 cost-model A/B switches, bit24 one hardware block per tile instead of one block per CU walking the
 *               tile list, bit26 XCD rendezvous (gemm256.hip) / residual-form bias per quad (gemm_mi16.hip), bits 27-30 epilogue-form A/B switches of the
 *               16x16x32 kernels (27 bias in the epilogue instead of as the accumulators' start value, 28 activation tested per quad + scalar GELU / SiLU,
 *               29 no interior-tile forms, 30 bias tested per quad: same results, profiles/r03_gemm_epilogue.md)
 *   gemm_stagger  persistent kernel (cfg 4) / tile loop of the fp32-residual shapes: de-phase the XCDs / tile groups
 *   gemm_group  M-tiles per group of the XCD-aware tile order (default 4)
 *   gemm_mi     32 = the 256x256 / 256x384 / 384x256 tiles on v_mfma_f32_32x32x16 (gemm256.hip; the product runs them on 16x16x32, gemm_mi16.hip);
 *               gemm_mi_modes bit mask of epilogue modes on the 16x16x32 kernels;  gemm_pref forced tile shape per mode
 *   gemm_wm4    wave grid of the 384x256 tile on the 16x16x32 kernels: bit mask of epilogue modes (1 16-bit output, 2 fp32 residual, 4 SwiGLU, 8 16-bit
 *               residual) that run 4 x 2 waves (96 x 128 wave tiles) instead of 2 x 4 (192 x 64); -1 none
 *   attn_waves  4 | 8 waves per block;  attn_dma 1 LDS-DMA | 2 register K/V staging;  attn_pipe 1 cross-tile pipelined hd-72 loop
 *   attn_ab     1 forced on | 3 16 waves per block | 4 PV on v_mfma_f32_32x32x16 (head dim padded to 96) | 5 PV on 16x16x32 with the softmax row sums on
 *               the VALU | 6 pre-scaled Q with the multiply and the max subtraction kept on the VALU
 *   attn_sp     1 software-pipelined kernel for both head sizes | 4 hd 128 on the one-wave-per-SIMD kernel (attention_w4.hip; product key until round 5) | 3 hd 128 with PV on v_mfma_f32_16x16x32 (3-4 % slower than the default form) | 5 attention_w4.hip with in-place tt = 1 score tiles | 6 / 7 its staging pieces from step 0 / 16 | 100 + bits its timing ablations
 */
#ifndef VIDEOITG_HIP_PROFILING_H
#define VIDEOITG_HIP_PROFILING_H
#include "videoitg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* (round 3; LOST its A/B against split-K x2 on the per-rank shapes - 447 / 694 against 682 / 911 TF/s, profiles/r03_context_parallel.md - and
 * therefore not in the product library since round 4)  Stream-K form of vitg_gemm_splitk's GEMM for FEWER output tiles than CUs and a long K - the same shapes one rank of an 8-GPU run has
 * (o: 98 tiles x 56 K-tiles, down: 98 x 296 on 256 CUs), where split-K x2 leaves 60 CUs idle: the K-tiles of all output tiles are laid end to end
 * and cut into equal runs, one per block (a block covers the tail of one tile and the head of the next); each segment's raw fp32 partial tile goes
 * to its own workspace slot and a second kernel adds a tile's segments in block order, the bias and the residual (no atomics: bit-reproducible).
 * vitg_gemm_streamk_plan: blocks == 0 means "does not apply" (use vitg_gemm_splitk_plan / vitg_gemm); otherwise pass blocks and a workspace of
 * workspace_bytes to vitg_gemm_streamk.  out = A W^T + bias + res (fp32). */
int vitg_gemm_streamk_plan(int M, int N, int K, int* blocks, long* workspace_bytes);
int vitg_gemm_streamk(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias, float* out,
                      int ldo, const float* res, int ldr, float* workspace, int blocks, void* stream);

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
