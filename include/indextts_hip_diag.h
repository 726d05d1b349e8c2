/*
 * indextts_hip_diag.h -- entry points that exist ONLY in the diagnostic build of the kernels
 * (index-tts-lora_amd/indextts/_lib/libindextts_hip_diag.so, `make -C index-tts-lora_amd/csrc diag`, -DITTS_DIAG=1
 * -DITTS_STAMPS=1).  The product library libindextts_hip.so exports none of them and keeps no mutable globals.
 * Used by tools/ (sweeps, timeline_skinny.py) through ITTS_HIP_LIB=<path to the diag library>; never by the package.
 */
#ifndef INDEXTTS_HIP_DIAG_H
#define INDEXTTS_HIP_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

/* tuning overrides: key 1 = column tiles per skinny-GEMM workgroup, key 2 = waves per skinny-GEMM workgroup, key 3 =
 * plain-GEMM kernel override (0 restores the built-in heuristic), key 4 = waves per decode-attention workgroup (4 or 8),
 * key 5 = ablation BIT MASK of the tiled convolution kernel for timing experiments (results are WRONG when non-zero):
 * 1 = no weight-fragment loads, 2 = no MFMA, 4 = no LDS fragment reads, 8 = no activation prefetch after the first chunk,
 * 16 = no epilogue */
int itts_debug_set(int key, int value);

/* every later itts_gemm_skinny launch writes 16 x u64 per workgroup (linear id = blockIdx.y * gridDim.x + blockIdx.x) to
 * buf: [0..9] s_memtime stamps (0 entry, 1 loads issued, 2 operands landed, 3 MFMAs done, 4 cross-wave barrier passed,
 * 5 epilogue stores issued, 6 stores drained + barrier, 7 ticket drawn, 8 all tickets seen, 9 row reduced), [10] exit
 * s_memtime, [11] / [12] s_memrealtime at entry / exit (100 MHz, comparable across workgroups and launches), [13] XCC id.
 * NULL switches the stamps off. */
int itts_debug_stamps(void* buf);

/* the same for gemm_conv_kernel (the tiled convolution): 16 x u64 per workgroup: s_memtime at 0 entry, 1 first activation
 * prefetch issued, 2 first chunk staged in LDS, 3 all MFMA steps done, 4 epilogue issued, 5 stores drained; [12] HW_ID,
 * [13] XCC id, [14] / [15] s_memrealtime at entry / exit */
int itts_debug_stamps_conv(void* buf);

/* the same for itts_sample: 16 x u64 per batch row: s_memtime at 0 entry, 1 logits + bitmap done, 2 processed scores in LDS,
 * 3 threshold known, 4 candidates compacted, 5 rank sort done, 6 token drawn, 7 bookkeeping done; [14] / [15] s_memrealtime
 * at entry / exit */
int itts_debug_stamps_sample(void* buf);

#ifdef __cplusplus
}
#endif
#endif
