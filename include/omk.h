/* omk.h -- C ABI of libomnimamba_hip.so: the MI355X (gfx950) kernels behind OmniMamba's Mamba-2 hot path.
 *
 * The reference has NO native boundary of its own: every hot op is a Python import from the third-party
 * packages mamba_ssm==2.2.2 / causal-conv1d==1.4.0 (/root/reference/requirements.txt:12-13), whose pybind11
 * Torch extensions take at::Tensor.  This header is the drop-in boundary this repo defines instead: plain
 * pointers, sizes and element strides, no torch types, no exceptions.  Each entry point names the reference
 * call site it serves (paths relative to /root/reference) and the upstream Python symbol it stands behind.
 *
 * Conventions
 *   - every function returns 0 (OMK_OK) or a negative omk_status; omk_last_error() gives thread-local text.
 *   - all launches are asynchronous on the hipStream_t passed as `stream` (0 = default stream); the library
 *     never allocates or frees device memory, never synchronises, keeps no mutable global state -> safe
 *     under hipGraph capture and one-process-per-GPU data parallelism.
 *   - tensors are described by OmkTensor (device pointer + dtype + shape + ELEMENT strides); data == NULL
 *     marks an absent optional tensor.  Scratch is caller-owned: ask omk_*_workspace_bytes first.
 *   - all arithmetic is fp32 internally; dtype says how a tensor is stored.
 */
#ifndef OMK_H
#define OMK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMK_ABI_VERSION 7
#define OMK_MAX_DIMS 5

typedef enum { OMK_OK = 0, OMK_EINVAL = -1, OMK_EARCH = -2, OMK_ELAUNCH = -3, OMK_EUNSUPPORTED = -4 } omk_status;
typedef enum { OMK_F32 = 0, OMK_BF16 = 1, OMK_F16 = 2, OMK_U8 = 3 /* masks only */ } omk_dtype;
typedef void* omk_stream; /* hipStream_t */

typedef struct {
  void* data;
  int32_t dtype; /* omk_dtype */
  int32_t ndim;
  int64_t shape[OMK_MAX_DIMS];
  int64_t stride[OMK_MAX_DIMS]; /* elements */
} OmkTensor;

int omk_abi_version(void);
const char* omk_last_error(void);
/* 1 when the library was built for the emulator (tests only), 0 for the real gfx950 build */
int omk_is_emulated(void);
/* sizeof() of the named parameter struct ("OmkTensor", "OmkSsdFwd", ...), 0 when unknown: lets a foreign-language
 * binding check its own struct layout against the library at load time */
size_t omk_sizeof(const char* struct_name);

/* ---- fused residual-add + RMSNorm/LayerNorm -------------------------------------------------------------
 * upstream mamba_ssm.ops.triton.layer_norm.layer_norm_fn / rms_norm_fn / RMSNorm
 * reference call sites: models/stage2/block.py:10,86-95 ; models/stage2/mixer_seq_simple.py:30,428-437      */
typedef struct {
  OmkTensor x;            /* (rows, cols) activation */
  OmkTensor residual;     /* optional (rows, cols) */
  OmkTensor weight;       /* (cols) */
  OmkTensor bias;         /* optional (cols) */
  OmkTensor y;            /* out (rows, cols), dtype of x */
  OmkTensor residual_out; /* optional out (rows, cols): x + residual, its own dtype (fp32 when residual_in_fp32) */
  OmkTensor rstd;         /* optional out (rows) f32 */
  OmkTensor mean;         /* optional out (rows) f32, LayerNorm only */
  float eps;
  int32_t is_rms_norm;
} OmkAddNormFwd;
int omk_add_norm_fwd(const OmkAddNormFwd* p, omk_stream stream);

typedef struct {
  OmkTensor dy;            /* (rows, cols) */
  OmkTensor dresidual_out; /* optional incoming grad of the prenorm residual output */
  OmkTensor xsum;          /* (rows, cols) the pre-norm sum saved by forward (residual_out, or x when no residual) */
  OmkTensor weight;
  OmkTensor rstd;          /* (rows) f32 */
  OmkTensor mean;          /* optional (rows) f32 */
  OmkTensor dx;            /* out (rows, cols) */
  OmkTensor dresidual_in;  /* optional out (rows, cols): same value as dx in the residual's dtype */
  OmkTensor dweight;       /* optional out (cols) f32; absent (frozen weight): no partial sums, no reduction launch */
  OmkTensor dbias;         /* optional out (cols) f32 */
  void* workspace;
  size_t workspace_bytes;
  int32_t is_rms_norm;
  int32_t has_bias;
} OmkAddNormBwd;
size_t omk_add_norm_bwd_workspace_bytes(const OmkAddNormBwd* p);
int omk_add_norm_bwd(const OmkAddNormBwd* p, omk_stream stream);

/* ---- gated RMSNorm -----------------------------------------------------------------------------------------
 * upstream mamba_ssm.ops.triton.layernorm_gated.{RMSNorm, rmsnorm_fn}; used as Mamba2.norm(y, z)
 * reference reach: models/stage2/block.py:117 -> Mamba2.forward / Mamba2.step                                 */
typedef struct {
  OmkTensor x;      /* (rows, cols) */
  OmkTensor z;      /* optional gate (rows, cols) */
  OmkTensor weight; /* (cols) */
  OmkTensor bias;   /* optional */
  OmkTensor y;      /* out (rows, cols) */
  OmkTensor rstd;   /* optional out (rows, ngroups) f32 */
  int64_t group_size;
  float eps;
  int32_t norm_before_gate;
} OmkNormGatedFwd;
int omk_norm_gated_fwd(const OmkNormGatedFwd* p, omk_stream stream);

typedef struct {
  OmkTensor dy, x, z, weight; /* z optional */
  OmkTensor dx, dz;           /* out; dz optional */
  OmkTensor dweight;          /* optional out (cols) f32; absent (frozen weight): no partial sums, no reduction launch */
  void* workspace;
  size_t workspace_bytes;
  int64_t group_size;
  float eps;
  int32_t norm_before_gate;
} OmkNormGatedBwd;
size_t omk_norm_gated_bwd_workspace_bytes(const OmkNormGatedBwd* p);
int omk_norm_gated_bwd(const OmkNormGatedBwd* p, omk_stream stream);

/* ---- causal depthwise conv1d ---------------------------------------------------------------------------
 * upstream causal_conv1d.{causal_conv1d_fn, causal_conv1d_update}
 * reference reach: models/stage2/mixer_seq_simple.py:17,200-205 (Mamba2) ; decode via
 * models/stage2/generation.py:195-211,412-431.  Logical shape (batch, channels, seqlen); strides make both the
 * channel-last (B, L, C) storage the Mamba-2 block uses and channel-first storage legal.                    */
typedef struct {
  OmkTensor x;              /* (B, C, L) logical */
  OmkTensor weight;         /* (C, W), W in 2..4 */
  OmkTensor bias;           /* optional (C) */
  OmkTensor initial_states; /* optional (B, C, W-1) */
  OmkTensor out;            /* (B, C, L) logical */
  OmkTensor final_states;   /* optional out (B, C, W-1) */
  int32_t silu;
} OmkConv1dFwd;
int omk_causal_conv1d_fwd(const OmkConv1dFwd* p, omk_stream stream);

typedef struct {
  OmkTensor x, weight, bias, initial_states; /* bias / initial_states optional */
  OmkTensor dout;                            /* (B, C, L) */
  OmkTensor dx;                              /* out (B, C, L) */
  OmkTensor dweight;                         /* out (C, W) f32, ACCUMULATED into (caller zeroes) */
  OmkTensor dbias;                           /* optional out (C) f32, accumulated */
  OmkTensor dinitial_states;                 /* optional out (B, C, W-1) */
  int32_t silu;
  /* ABI 7, optional: with a workspace of omk_causal_conv1d_bwd_workspace_bytes() the long channel-last 16-bit kernel leaves one partial
   * dweight / dbias row per (batch, 256-token tile) and a second launch adds them up in a fixed order -- the same gradients on every run
   * and ~ 12 us less at the 1.3B slice than the fp32 atomics taken without one (NULL / too small: atomics; other layouts: unused) */
  void* workspace;
  size_t workspace_bytes;
} OmkConv1dBwd;
size_t omk_causal_conv1d_bwd_workspace_bytes(const OmkConv1dBwd* p);
int omk_causal_conv1d_bwd(const OmkConv1dBwd* p, omk_stream stream);

typedef struct {
  OmkTensor x;          /* (B, C, T), T small (1 for decode) */
  OmkTensor conv_state; /* (B, C, S), S >= W-1, updated in place */
  OmkTensor weight, bias;
  OmkTensor out;        /* (B, C, T) */
  int32_t silu;
} OmkConv1dUpdate;
int omk_causal_conv1d_update(const OmkConv1dUpdate* p, omk_stream stream);

/* ---- single-token SSM state update ---------------------------------------------------------------------
 * upstream mamba_ssm.ops.triton.selective_state_update.selective_state_update (Mamba2.step)
 * reference reach: models/stage2/generation.py:195-211,412-424 -> MixerModel.forward -> Block -> Mamba2.step */
typedef struct {
  OmkTensor state;   /* (B, H, P, N) in place */
  OmkTensor x;       /* (B, H, P) */
  OmkTensor dt;      /* (B, H, P) stride 0 on P allowed */
  OmkTensor A;       /* (H, P, N) stride 0 on P,N allowed */
  OmkTensor Bm, Cm;  /* (B, G, N) */
  OmkTensor D;       /* optional (H, P) */
  OmkTensor z;       /* optional (B, H, P) */
  OmkTensor dt_bias; /* optional (H, P) */
  OmkTensor out;     /* (B, H, P) */
  int32_t dt_softplus;
} OmkStateUpdate;
int omk_selective_state_update(const OmkStateUpdate* p, omk_stream stream);

/* ---- Mamba-1 selective scan ----------------------------------------------------------------------------
 * upstream mamba_ssm.ops.selective_scan_interface.selective_scan_fn (BASELINE.json configs[0] signature)
 * reference reach: models/stage2/mixer_seq_simple.py:16,197-201 (ssm_cfg.layer == "Mamba1")                  */
typedef struct {
  OmkTensor u, delta;   /* (B, D, L) logical, any strides */
  OmkTensor A;          /* (D, N) f32 */
  OmkTensor Bm, Cm;     /* (B, G, N, L) logical (G=1 for the 3-d form) or (D, N) constant (ndim == 2) */
  OmkTensor D;          /* optional (D) */
  OmkTensor z;          /* optional (B, D, L) */
  OmkTensor delta_bias; /* optional (D) */
  OmkTensor out;        /* (B, D, L) */
  OmkTensor last_state; /* optional out (B, D, N) f32 */
  OmkTensor pass_states;/* optional out (B, D, ceil(L / 512), N) f32, contiguous: the state in front of every 512-token pass of the
                           chunked scan -- hand it to omk_selective_scan_bwd and the backward skips its state-only forward pass.
                           Needs L-contiguous u / delta / z / out / B / C and L >= 64, or form 2 below (OMK_EUNSUPPORTED otherwise). */
  int32_t delta_softplus;
} OmkSelScanFwd;
int omk_selective_scan_fwd(const OmkSelScanFwd* p, omk_stream stream);
/* Which form omk_selective_scan_fwd takes for these tensors (nothing is launched): 2 = lanes-are-channels sweep (channel-last (B, L, D)
 * or L-contiguous storage, d_state <= 16, variable B / C of u's dtype, and enough (batch, 64-channel tile) waves to fill the chip --
 * or L < 64), 1 = chunked associative scan (L-contiguous rows, lanes = time), 0 = per-channel sequential kernel; < 0: omk_status.
 * A host that holds channel-last views asks before it decides to make L-contiguous copies (omnimamba_amd/selective_scan.py). */
int omk_selective_scan_fwd_form(const OmkSelScanFwd* p);

typedef struct {
  OmkTensor u, delta, A, Bm, Cm, D, z, delta_bias; /* as forward */
  OmkTensor dout;                                  /* (B, D, L) */
  OmkTensor du, ddelta;                            /* out (B, D, L) */
  OmkTensor dA;                                    /* out (D, N) f32 accumulated */
  OmkTensor dB, dC;                                /* out f32 accumulated: (B, G, N, L) for variable B / C, (D, N) for constant */
  OmkTensor dD;                                    /* optional out (D) f32 accumulated */
  OmkTensor dz;                                    /* optional out (B, D, L) */
  OmkTensor ddelta_bias;                           /* optional out (D) f32 accumulated */
  OmkTensor pass_states;                           /* optional in: what omk_selective_scan_fwd left (chunked form only) */
  void* workspace;                                 /* state checkpoints of the recomputed forward */
  size_t workspace_bytes;
  int32_t delta_softplus;
} OmkSelScanBwd;
/* ABI 6: which form omk_selective_scan_bwd takes on these views: 2 = the lanes-are-channels reverse sweep on channel-last (B, L, D) views
 * as they lie (input-dependent B / C of u's dtype, d_state <= 16, enough sequences to fill the chip), 1 = the chunked scan (L-contiguous rows),
 * 0 = the per-channel kernel.  The host mirror asks before it decides about L-contiguous copies (omk_selective_scan_fwd_form's twin). */
int omk_selective_scan_bwd_form(const OmkSelScanBwd* p);
size_t omk_selective_scan_bwd_workspace_bytes(const OmkSelScanBwd* p);
int omk_selective_scan_bwd(const OmkSelScanBwd* p, omk_stream stream);   /* L-contiguous rows, variable B / C, L >= 64, 8 | channels
                                                                             per group: d_state <= 64; other layouts: d_state <= 16 */

/* ---- decode-step projection fused with the normalisation in front of it ----------------------------------
 * one launch for: reference block.py:86-95 (fused add + RMSNorm) -> lora.py:185-279 (base + task LoRA) at one token
 * per sequence, and for Mamba2.step's gated RMSNorm -> out_proj (upstream mamba2.py step()).  One sequence: any dtype mix.  Two to eight
 * sequences: one dtype for activations / weights / norm weight / LoRA (fp32 or bf16), in_features 1024 / 2048 / 4096.
 * Anything else returns OMK_EUNSUPPORTED: use the separate ops.                                                     */
typedef struct {
  OmkTensor x;             /* (B, in) */
  OmkTensor residual;      /* optional (B, in): added before the norm */
  OmkTensor z;             /* optional (B, in): gate of the gated norm (dtype of x) */
  OmkTensor norm_weight;   /* optional (in): RMSNorm weight; absent = no normalisation */
  OmkTensor weight;        /* (out, in) f32 / bf16 / f16, 16-byte aligned rows */
  OmkTensor bias;          /* optional (out) */
  OmkTensor lora_a;        /* optional (r, in), r <= 16 */
  OmkTensor lora_b;        /* optional (out, r) */
  OmkTensor residual_out;  /* optional out (B, in): x + residual */
  OmkTensor out;           /* out (B, out) */
  /* optional tail for the in_proj call of Mamba2.step: output rows [conv_offset, conv_offset + C) are the new xBC inputs;
   * they are pushed through causal_conv1d_update (+ SiLU) right where they are produced -- out holds the convolved
   * values, conv_state (B, C, S) is rolled in place -- so the step needs no separate convolution launch.  Only taken by
   * the uniform-dtype kernel (all of x / weight / conv tensors one dtype); otherwise OMK_EUNSUPPORTED. */
  OmkTensor conv_state;    /* optional (B, C, S), S = W-1 .. 4, in place */
  OmkTensor conv_weight;   /* (C, W), W = 2 .. 4 */
  OmkTensor conv_bias;     /* optional (C) */
  int64_t group_size;      /* norm group size (0 = in) */
  int64_t conv_offset;
  float eps;
  float lora_scale;
  int32_t norm_before_gate;
  int32_t conv_silu;
} OmkNormLinear;
int omk_norm_linear(const OmkNormLinear* p, omk_stream stream);

/* ---- task LoRA of a projection: out += scale * h lora_b^T, in place ------------------------------------------
 * reference models/stage2/lora.py:263-279 (result += lora_B(lora_A(dropout(x))) * scaling) at training / prefill token counts.
 * Rank 8 or 16, 16-byte aligned rows; otherwise OMK_EUNSUPPORTED (callers use a GEMM).                              */
typedef struct {
  OmkTensor out;     /* (T, N) in place */
  OmkTensor h;       /* (T, r) = lora_A(dropout(x)), dtype of out */
  OmkTensor lora_b;  /* (N, r) */
  OmkTensor mask;    /* optional (T, N) u8, rows contiguous: only elements with a non-zero mask byte receive the update (the
                        dropout in front of lora_A, seen from its backward: dx += mask / (1 - p) * (dh A)) */
  float scale;
} OmkLoraAdd;
int omk_lora_add(const OmkLoraAdd* p, omk_stream stream);

/* ---- backward of the rank-r up-projection, both products in ONE pass over dy --------------------------------------------
 * dh (T, r) = dy (T, N) lora_b (N, r)   and   dlora_b (N, r) = dy^T h (T, r)   (reference lora.py:263-279, backward of
 * lora_B(.)).  As two library GEMMs each reads the (tokens, 8512) gradient once (279 MB at 16 k tokens: 69 + 59 us).
 * bf16 dy / h, rank 8, 16-byte aligned rows.  The kernel leaves PARTIAL sums -- dh per 256-column block of dy, dlora_b per token
 * chunk -- that the caller adds up (two small reductions: deterministic, no atomics); omk_lora_up_bwd_parts gives the counts.   */
typedef struct {
  OmkTensor dy;       /* (T, N) bf16 */
  OmkTensor lora_b;   /* (N, r) f32 / bf16 / f16 */
  OmkTensor h;        /* (T, r) bf16 */
  OmkTensor dh;       /* out (column_blocks, T, r) f32, dense: every element is written */
  OmkTensor dlora_b;  /* out (token_chunks, N, r) f32, dense: every element is written */
} OmkLoraUpBwd;
/* column_blocks -> parts[0], token_chunks -> parts[1] for a (T, N) gradient */
int omk_lora_up_bwd_parts(int64_t T, int64_t N, int32_t* parts);
int omk_lora_up_bwd(const OmkLoraUpBwd* p, omk_stream stream);

/* ---- Mamba-2 SSD chunked scan ----------------------------------------------------------------------------
 * upstream mamba_ssm.ops.triton.ssd_combined.mamba_chunk_scan_combined (+ the scan stage of
 * mamba_split_conv1d_scan_combined); reference reach: models/stage2/block.py:117 -> Mamba2.forward              */
typedef struct {
  OmkTensor x;              /* (B, L, H, P) */
  OmkTensor dt;             /* (B, L, H) raw dt (before bias/softplus/clamp) */
  OmkTensor A;              /* (H) f32, negative */
  OmkTensor Bm, Cm;         /* (B, L, G, N) */
  OmkTensor D;              /* optional (H) or (H, P) */
  OmkTensor z;              /* optional (B, L, H, P): out = y * silu(z) */
  OmkTensor dt_bias;        /* optional (H) */
  OmkTensor initial_states; /* optional (B, H, P, N) */
  OmkTensor out;            /* (B, L, H, P); absent = the STATE-ONLY pass: only final_states is produced (one shard of a
                             * sequence cut over several GPUs, omnimamba_amd/context_parallel.py) -- MFMA shape only, else
                             * OMK_EUNSUPPORTED; Cm, D are ignored */
  OmkTensor out_x;          /* optional out (B, L, H, P): the pre-gate y (only meaningful with z; saved for backward) */
  OmkTensor final_states;   /* optional out (B, H, P, N) f32 */
  OmkTensor window_states;  /* optional out, bf16, contiguous, omk_ssd_scan_fwd_window_states_bytes(p) bytes: the carried state in
                             * front of every 128-token window, (B, ceil(L / 128), H) images of 16 KB in the kernel's own operand
                             * layout.  A training forward saves it and hands it to omk_ssd_scan_bwd, which then skips its own
                             * state pass over x (upstream's backward recomputes the chunk states the same way, ssd_combined.py
                             * _mamba_chunk_scan_combined_bwd).  Opaque: only omk_ssd_scan_bwd reads it */
  void* workspace;
  size_t workspace_bytes;
  float dt_min, dt_max;     /* clamp (dt_limit); (0, +inf) = none */
  int32_t dt_softplus;
  int32_t chunk_size;       /* API parity only: the result does not depend on it */
  int32_t force_generic;    /* 1 = use the shape-generic fp32 VALU kernel even when the MFMA kernel applies */
  int32_t flags;            /* ABI 6: OMK_SSD_* bits below, 0 = the default kernels.  Per call -- the library reads no environment
                             * variable that changes which scan kernel runs or what it computes */
  OmkTensor conv_weight;    /* ABI 6, optional (H * P, width <= 4): K2 fusion of the forward-only path (prefill / inference).  When present,
                             * x is the PRE-conv input of upstream's causal_conv1d_fn(..., activation="silu") for the x channels of xBC, and
                             * the scan applies out[t] = silu(bias + sum_k w[k] x[t - width + 1 + k]) (zeros in front of the sequence) while
                             * it stages x -- same fp32 arithmetic and the same single bf16 rounding as omk_causal_conv1d_fwd, so the result
                             * equals conv kernel + scan bit for bit.  B / C stay the conv OUTPUT (a 2 G N-channel conv launch of the caller).
                             * Plain bf16 forward on the specialised-wave kernel only (no z / out_x / window_states / PRECISE, sequence not
                             * split): otherwise OMK_EUNSUPPORTED and the caller runs conv + scan separately.
                             * reference: models/stage2/generation.py:195-211 prefill, scripts/inference_mmu.py:137-147 */
  OmkTensor conv_bias;      /* optional (H * P) */
} OmkSsdFwd;
/* OmkSsdFwd::flags / OmkSsdBwd::flags */
#define OMK_SSD_PRECISE      1  /* forward, bf16 MFMA scan: the carried state meets C as a bf16 hi + lo pair and the state-update operand is
                                 * hi + lo: y within 1e-3 (arithmetic, rel-L2) of the fp32 recurrence on EVERY head, slow-decay heads
                                 * included (the default rounds both to bf16 once, as the reference pipeline does: 1.3e-3 .. 2.3e-3 there).
                                 * Price: profiles/r06_precise.txt.  Window states cannot be saved by a PRECISE forward */
#define OMK_SSD_KHILO        2  /* forward: only the state-update operand as hi + lo (the carried state / final_states exact to fp32
                                 * accumulation); implied whenever final_states is asked for */
#define OMK_SSD_EVERY_CHUNK  4  /* specialised-wave kernel: rescale the carried state at every chunk end instead of only where its basis
                                 * drifts by 2^60 (the arithmetic of the column-slice kernel, bit for bit; tests) */
#define OMK_SSD_NO_SPLIT     8  /* never cut the sequence into segments (few (batch, head) sequences normally are: csrc/ssd_scan.h) */
#define OMK_SSD_COLUMN_SLICE 16 /* class A scans on the column-slice kernel (ssd_a6.hip) instead of the specialised-wave kernel */
/* Measurement aid (ABI 6): the scan kernels the LAST omk_ssd_scan_fwd / omk_ssd_scan_bwd call of this thread launched, ';'-separated,
 * template arguments included, e.g. "ssd_dt_prep;ssd_a8<mode=0,dump=1,khilo=0,precise=0>".  bench.py puts it next to its HIP-event
 * time and refuses a PMC traffic file (profiles/ssd_*_traffic.json) recorded for another kernel. */
const char* omk_ssd_last_kernels(void);
size_t omk_ssd_scan_fwd_workspace_bytes(const OmkSsdFwd* p);
/* bytes of window_states for these arguments (window_states itself is not looked at); 0 = this forward cannot save them (shape
 * outside the MFMA kernel, gate / out_x requested, fp32 activations): pass none */
size_t omk_ssd_scan_fwd_window_states_bytes(const OmkSsdFwd* p);
int omk_ssd_scan_fwd(const OmkSsdFwd* p, omk_stream stream);

typedef struct {
  OmkTensor x, dt, A, Bm, Cm, D, dt_bias, initial_states; /* as forward.  z gating is NOT part of this call: the
                                caller passes dout * silu(z) and forms dz = dout * out_x * silu'(z) itself (out_x = the
                                forward's pre-gate output) */
  OmkTensor y;               /* optional (B, L, H, P): the forward's pre-gate output.  Accepted for signature parity with
                                upstream's backward; no kernel reads it (the scans recompute what they need) */
  OmkTensor dout;            /* (B, L, H, P) grad of the (pre-gate) output */
  OmkTensor dfinal_states;   /* optional (B, H, P, N) f32 */
  OmkTensor dx;              /* out (B, L, H, P) */
  OmkTensor ddt;             /* out (B, L, H) f32: grad wrt RAW dt */
  OmkTensor dA;              /* out (H) f32 */
  OmkTensor dB, dC;          /* out (B, L, G, N) */
  OmkTensor dD;              /* optional out (H) or (H, P) f32 */
  OmkTensor ddt_bias;        /* optional out (H) f32 */
  OmkTensor dinitial_states; /* optional out (B, H, P, N) f32 */
  OmkTensor window_states;   /* optional in: what omk_ssd_scan_fwd left in its window_states for the SAME x, dt, A, Bm, dt_bias,
                                initial_states and clamp; ignored by the paths that do not use window states */
  void* workspace;
  size_t workspace_bytes;
  float dt_min, dt_max;
  int32_t dt_softplus;
  int32_t chunk_size;
  int32_t force_generic;
  int32_t flags;             /* ABI 6: OMK_SSD_NO_SPLIT, OMK_SSD_COLUMN_SLICE, OMK_SSD_EVERY_CHUNK, OMK_SSD_SEQUENTIAL_BWD */
} OmkSsdBwd;
#define OMK_SSD_SEQUENTIAL_BWD 32 /* backward: the three sequential MFMA scans of rounds 1 - 2 instead of the chunk-parallel dB / dC */
size_t omk_ssd_scan_bwd_workspace_bytes(const OmkSsdBwd* p);
int omk_ssd_scan_bwd(const OmkSsdBwd* p, omk_stream stream);

/* ---- cross entropy over one block of logits: loss per row + the gradient written over the logits ----------------
 * The loss the reference forms from the heads' outputs (models/mamba_vlm.py:88-102 shift + flatten; models/omnimamba.py:63,
 * 276-279,305-306: torch.nn.CrossEntropyLoss(), mean over the labels that are not -100).  Used by the chunked fused
 * linear + cross-entropy of omnimamba_amd/fused_ce.py, which never holds more than one token block of logits
 * (SURVEY.md section 8 row f1).  Per row r (label y):  loss[r] = logsumexp(logits[r]) - logits[r][y];
 * logits[r][v] <- (softmax(logits[r])[v] - [v == y]) * grad_scale[0]   (rows with label == ignore_index: loss 0, gradient 0). */
typedef struct {
  OmkTensor logits;        /* (T, V) in / out (the gradient), unit stride on V */
  const int64_t* labels;   /* (T) */
  OmkTensor losses;        /* out (T) f32 */
  const float* grad_scale; /* device scalar the gradient is multiplied with (1 / number of counted labels); NULL = 1 */
  int64_t ignore_index;
  int32_t write_grad;      /* 0: only the losses */
} OmkCrossEntropy;
int omk_cross_entropy(const OmkCrossEntropy* p, omk_stream stream);

/* ---- token sampling inside the decode step --------------------------------------------------------------
 * reference models/stage2/generation.py:87-121 `sample(logits, top_k, top_p, min_p, temperature)`: the top_k == 1 short cut
 * (argmax) and the top_k > 0 branch (top-k -> / temperature -> top-p filter of :64-76 -> multinomial), for 1 <= top_k <= 64.
 * One uniform per row from Philox4x32-10 keyed by (seed, row, *step_counter + offset): the reference draws with
 * torch.multinomial, so the ids agree in distribution (and exactly for top_k == 1), not stream for stream.  step_counter is a
 * device int64 the caller advances (inside its captured graph), NULL = 0.  top_k == 0 is the whole-vocabulary branch (:107-119):
 * the plain multinomial of softmax(logits / temperature) (the default arguments of t2i_generate), behind the reference's top-p cut
 * when 0 < top_p < 1 (ascending cumulative probability <= 1 - top_p is cut, ties at the boundary in index order) or, ABI 7, behind its
 * min_p filter when min_p > 0 (a raw logit below min_p x the largest softmax(logits) probability is cut -- the reference's own
 * comparison, :43; top_p is ignored then, as there; a row with nothing left returns its arg max where the reference raises).      */
typedef struct {
  OmkTensor logits;        /* (batch, vocab) f32 / bf16 / f16, unit last stride */
  OmkTensor out_ids;       /* out (batch) int64, dense (dtype field ignored) */
  const void* step_counter; /* optional device int64 */
  uint64_t seed, offset;
  int32_t top_k;           /* 0 (whole vocabulary) or 1 .. 64 */
  float top_p, temperature;
  float min_p;             /* ABI 7 (occupies the former tail padding): 0 = off; > 0 only with top_k == 0 */
} OmkSample;
int omk_sample(const OmkSample* p, omk_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OMK_H */
