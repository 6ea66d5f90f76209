/* matpbr_experimental.h -- measurement and A/B entry points of libmatpbr.so.  NOT part of the drop-in boundary (include/matpbr.h): nothing a
 * product path needs is declared here, and anything here may go away.  Exported by the same library so that the tools/ scripts and bench.py's
 * `roofline` can use them without a second build. */
#ifndef MATPBR_EXPERIMENTAL_H
#define MATPBR_EXPERIMENTAL_H
#include "matpbr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement only (bench.py's `roofline`): matpbr_brdf_phase_stages whose folded step launch (MATPBR_STAGE_BACKWARD of a phase with `lazy_fold`)
 * records the kernel's own begin and end into two caller-owned hipEvent_t created with timing enabled (hipExtLaunchKernelGGL) -- events recorded
 * around a launch include the stream's dispatch latency on both sides, 2-5 us on a 50 us kernel.  MATPBR_ERR_UNSUPPORTED for other phases. */
int matpbr_brdf_phase_stages_timed(const MatpbrBrdfPhase* phase, int t, float lr, uint32_t stages, void* start_event, void* stop_event,
                                   void* stream);

/* The main loop of the split-operand layer kernels (matpbr_mlp_layer_fwd_bx[_tail|_head], matpbr_mlp_layer_bwd_input_bx[_sgn],
 * matpbr_mlp_first_layer_bwd_bx) with 256 output columns and a reduction that is a multiple of 32:
 *   2 (default)  operands by LDS-DMA (global_load_lds_dwordx4), two 256-thread workgroups per CU with 64 x 128 wave tiles
 *                (mlp_nt_gx; the first-layer form and the head form with stored cosines run as mode 1)
 *   1            operands by LDS-DMA, one 512-thread workgroup per CU (mlp_nt_bx<.., GL>)
 *   0            register-staged operands (mlp_nt_bx)
 *   3            as 2 with the first-layer form on mlp_nt_gx as well (23 spilled registers; measured +0.2 %: not the default)
 * The three form the same products in the same order: outputs are the same bits (the bias-gradient column sums are grouped per
 * workgroup and agree to rounding).  A measurement switch, process-wide; returns the previous setting. */
int matpbr_mlp_set_lds_dma(int mode);

#ifdef __cplusplus
}
#endif
#endif
