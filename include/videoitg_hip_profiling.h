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

#ifdef __cplusplus
}
#endif
#endif
