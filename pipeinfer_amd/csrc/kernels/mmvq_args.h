// Launch arguments shared by the quantised mat-vec kernels (mmvq.hip) and the multi-token MFMA kernel (mmq_mfma.hip).
#pragma once
#include "ops_common.h"

// up to three weight matrices of one type and width share a launch (wq/wk/wv, ffn_up/ffn_gate): one activation, more bytes
// per kernel — at decode sizes a launch costs ~3.5 us before HBM is streaming, so fewer, larger launches matter.
#define MMVQ_MAX_SEG 3
struct mmvq_seg {
    const uint8_t * p0, * p1, * p2;
    const uint8_t * mq;         // tile-major copy (MFMA kernel), may be null
    float *         out;        // [n_tok][M]
    const float *   res;        // nullable
    int             M;
    int             wave0;      // first wave (K-quant kernel) of this segment
    // K-quant kernel, attention projections: where the rows go instead of `out` (pi_op_mul_mat_qkv_rope)
    //   1  RoPE (pairs 2i, 2i+1), f16, query buffer  dst16[(tok0 + t) * stride16 + row]
    //   2  RoPE, f16, K cache cell                   dst16[(kv_head + tok0 + t) * stride16 + row]
    //   3  f16, V cache cell                         dst16[(kv_head + tok0 + t) * stride16 + row]
    int             store;
    uint16_t *      dst16;
    int             stride16;
};
struct mmvq_args {
    mmvq_seg seg[MMVQ_MAX_SEG];
    const uint8_t * aq;         // activation planes (pi_q8_layout)
    const uint8_t * ad;
    const uint8_t * abs_;
    int K, nb;                  // nb: super-blocks (K-quants) or groups of 4 blocks (Q4_0/Q8_0) per row
    // RoPE + KV-store epilogue (seg.store != 0): positions of the launch's tokens, (cos, sin) table [pos][hd/2][2], {kv_head, n_kv} in device memory
    const int32_t * pos; const float * rope_cs; const int32_t * dyn;
    int kv_head, hd, tok0;
    // multi-token MFMA kernel (mmq_mfma.hip): operand planes of this launch's chunk of <= 16 tokens (pi_q8_frag_layout), tokens in the chunk
    const uint8_t * frag; const uint8_t * aux; int nt;
    int type_b_mask;            // k_mmvq_kq2 (two weight types in one launch): bit i set <=> segment i is of the second type
    const int32_t * abort_w;    // abort word of the run (pi_device.h), nullable: a kernel whose run has been cancelled returns on entry
    int dbg;                    // tuning only (PI_MMQ_DBG): 1 = skip the arithmetic, 2 = skip the activation staging and its barriers (results are then garbage)
};


// host-side description of one launch: segments + bytes for the roofline report
struct mmvq_launch { mmvq_args g; int nseg; int64_t m_total, m_min; double wbytes; bool any_res; bool rope = false; };

// mmq_mfma.hip: 1 when the MFMA kernel handles this weight type
bool pi_mmq_supported(int type);
bool pi_mmq_wants(int type, int n_tok, int64_t rows, int64_t k);      // k: row length (only Q6_K's threshold depends on it; 0: not known)
// launches k_mmq_kq for L (segments filled in, g.frag / g.aux / g.nt set); ea / eb: optional events bracketing the kernel (sampled timing)
int  pi_mmq_launch(int type, mmvq_launch & L, hipStream_t stream, hipEvent_t ea, hipEvent_t eb);
// two weight types in one matrix-core launch (k_mmq_kq2t): n_a Q4_K segments followed by segments of type_b (Q5_K / Q6_K), 3..8 tokens
bool pi_mmq2_applies(int type_b, int n_tok, int64_t rows_a, int64_t rows_b);
int  pi_mmq_launch2(int type_b, int n_a, bool four_chunks, mmvq_launch & L, hipStream_t stream, hipEvent_t ea, hipEvent_t eb);
