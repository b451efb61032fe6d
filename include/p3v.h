/* p3v.h -- C ABI of the MI355X (gfx950) Phi-3-Vision hot path.
 *
 * The reference (JosefAlbers/Phi-3-Vision-MLX) has no native layer: every
 * tensor op on its hot path is a call into the third-party `mlx` wheel.  Each
 * entry point below therefore replaces one MLX call site (or a fused group of
 * them) in the reference's `phi.py` / `phi_3_vision_mlx.py`; the call site is
 * cited as file:line next to each declaration.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless
 *     marked `host`; bf16 travels as uint16_t;
 *   - `stream` is a hipStream_t passed as void*; launches are stream-ordered,
 *     never synchronise and never allocate (callers pass workspaces);
 *   - return 0 on success, a negative P3V_ERR_* code otherwise; nothing throws;
 *   - re-entrant across streams.  Weights are [out_features, in_features]
 *     row-major (the HF layout the reference loads).
 */
#ifndef P3V_H
#define P3V_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3V_VERSION 600
#define P3V_OK 0
#define P3V_ERR_ARG (-22)         /* bad shape / null pointer / unsupported size */
#define P3V_ERR_LAUNCH (-5)       /* hipGetLastError() != hipSuccess after launch */
#define P3V_ERR_UNSUPPORTED (-95) /* combination not implemented */
#define P3V_ERR_HIP (-14)         /* a HIP runtime call failed */

typedef struct {
  int cu_count, lds_per_cu, wave_size, clock_khz, mem_clock_khz, mem_bus_bits;
  int64_t hbm_bytes;
  char arch[32];
} p3v_props_t;

int p3v_version(void);
int p3v_device_props(int device, p3v_props_t* out /* host */);
const char* p3v_strerror(int code);

/* Launch-policy knobs ("gemm_big_rows", "gemm_no_splitk", "gemv_wpc", ... -- the table in csrc/p3v_runtime.hip).  Each is
 * read ONCE, on first use, from the environment variable P3V_<NAME IN UPPER CASE>; afterwards only these calls change
 * it (kernel tests pin a kernel variant with them).  No launch reads the environment.  Process-wide, not thread-safe
 * against concurrent launches: set before launching.  P3V_ERR_ARG for an unknown name. */
int p3v_set_tuning(const char* name /* host */, int value);
int p3v_get_tuning(const char* name /* host */, int* value /* host */);

/* ---- embedding gather: nn.Embedding, phi.py:568,577.  Negative ids (image
 * slots, phi.py:270) are clamped to 0; their rows are overwritten later. */
int p3v_embed_gather(const int32_t* ids, const uint16_t* table, uint16_t* out,
                     int n_tok, int hidden, int vocab, void* stream);

/* ---- nn.RMSNorm (= mx.fast.rms_norm), phi.py:478-479,571: y = bf16(bf16(x*rsqrt(mean(x^2)+eps)) * w) -- fp32 statistics, the
 *      normalised row is rounded to bf16 before the weight multiply (pinned by tests/golden/ref_model_tiny.npz) */
int p3v_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int hidden, float eps, void* stream);

/* ---- nn.LayerNorm, phi.py:165,167,212: fp32 in (ViT residual stream); bf16 out (feeds a
 * projection) or fp32 out (pre_layrnorm, which IS the residual stream; may be in place) */
int p3v_layernorm(const float* x, const uint16_t* w, const uint16_t* b, void* y, int out_f32,
                  int rows, int hidden, float eps, void* stream);

/* ---- dense projection  C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), bf16 MFMA, fp32 accumulate.
 * Replaces every nn.Linear on the path (phi.py:140-149,154-159,391,437-438,
 * 465-466,604) and the k=14,s=14 patch conv (phi.py:186-192,199). */
enum {
  P3V_EPI_NONE = 0,        /* out bf16 = acc                                              */
  P3V_EPI_BIAS = 1,        /* out bf16 = acc + bias                                       */
  P3V_EPI_BIAS_QGELU = 2,  /* out bf16 = g(acc+bias), g(x)=x*sigmoid(1.702x)  phi.py:154  */
  P3V_EPI_BIAS_GELU = 3,   /* out bf16 = gelu_erf(acc+bias)                   phi.py:391  */
  P3V_EPI_BIAS_RESID_F32 = 4, /* out f32  = resid_f32 + acc + bias            phi.py:170-171 */
  P3V_EPI_RESID_BF16 = 5,  /* out bf16 = resid + bf16(acc)                    phi.py:460,483-485 */
  P3V_EPI_SILU_MUL = 6,    /* W = [gate;up] (2N rows): out bf16[M,N] = silu(g)*u  phi.py:469-471 */
  P3V_EPI_PATCH = 7,       /* out f32 row (m/P)*(P+1)+1+m%P = acc + pos[1+m%P]   phi.py:199-205 */
  P3V_EPI_F32 = 8          /* out f32 = acc (+ bias if given)                               */
};
typedef struct {
  const uint16_t* A;   /* [M, lda] bf16 */
  const uint16_t* W;   /* [N or 2N, ldw] bf16 */
  void* out;           /* bf16 or f32, row stride ldo elements */
  const uint16_t* bias;  /* [N] bf16 or null */
  const void* resid;   /* bf16 or f32 [M, ldo] or null; may alias out */
  const uint16_t* pos; /* P3V_EPI_PATCH: position embedding [(P+1), N] bf16 */
  int M, N, K;         /* K % 64 == 0 */
  int lda, ldw, ldo;
  int epilogue;
  int patches_per_img; /* P3V_EPI_PATCH: P (=576) */
  void* ws;            /* caller-owned device workspace (16-byte aligned) or null */
  int64_t ws_bytes;    /* its size; see p3v_gemm_ws_bytes */
} p3v_gemm_args_t;
/* Operands are addressed with 32-bit byte offsets: M * lda * 2 and rows(W) * ldw * 2 must stay below 4 GiB
 * (P3V_ERR_UNSUPPORTED otherwise).  Shapes with 17 <= M <= 1024 and fewer than 256 output tiles (short prompts) run split
 * over K through fp32 partials in `ws`: p3v_gemm_ws_bytes(M, N, K, epilogue) is the size that path needs (0 = the shape
 * does not use a workspace).  The library never allocates, frees or synchronises: with ws == null or too small the same
 * shape runs on the one-pass kernel (slower for short prompts, same results up to fp32 summation order).  The workspace
 * is only live between the two launches of one call, so calls on ONE stream may share it; concurrent streams need their
 * own.  Graph-capturable. */
int p3v_gemm(const p3v_gemm_args_t* args /* host */, void* stream);
int64_t p3v_gemm_ws_bytes(int M, int N, int K, int epilogue);
/* Round 6: 9 .. 32 rows of A on bf16 weights with K = 3072 or 8192 (a 9 .. 32-sequence decode batch; the reference's batched benchmark
 * runs 15, phi_3_vision_mlx.py:1226-1243) take a register-streaming kernel inside p3v_gemm / p3v_gemm_resid_norm (weights HBM ->
 * registers in whole lines, A resident as MFMA fragments): this query says how -- 0 = not that kernel's shape, 1 = one pass,
 * S > 1 = S K slices through the `ws` partials + the reduction launch.  Host-only. */
int p3v_gemm_rows_slices(int M, int N, int K, int epilogue);

/* ---- skinny projection for decode: y[M,N] = x[M,K] * W[N,K]^T, M <= 16, weight-streaming
 * (M == 1: VALU dot products; 2 <= M <= 16: the weight rows go straight from HBM into MFMA fragments).
 * Same call sites as p3v_gemm when L*B is small.  Optional fused RMSNorm of x
 * (norm_w != null: x is normalised with norm_w/eps before use, phi.py:482,484).
 * Epilogues: NONE, RESID_BF16, SILU_MUL (W = [gate;up]), F32. */
typedef struct {
  const uint16_t* x;      /* [M, K] bf16 */
  const uint16_t* W;      /* [N or 2N, K] bf16 */
  void* out;              /* [M, N] bf16 (or f32 for P3V_EPI_F32) */
  const uint16_t* resid;  /* [M, N] bf16 or null; may alias out */
  const uint16_t* norm_w; /* [K] bf16 or null */
  float norm_eps;
  int M, N, K;            /* M <= 16, K % 8 == 0 (M > 8 needs K % 512 == 0) */
  int epilogue;
} p3v_gemv_args_t;
int p3v_gemv(const p3v_gemv_args_t* args /* host */, void* stream);

/* ---- fp8 (OCP e4m3fn) weight-only projections (quantize_model=True; replaces the int4 `nn.quantize` of
 * phi_3_vision_mlx.py:264,296): W is u8 [N or 2N, K] bit patterns, w = fp8 * w_scale[row]; x, accumulation
 * and outputs as in p3v_gemv.  K in {3072, 8192}, M <= 16.  p3v_dequant_fp8 expands rows to bf16 (prefill). */
typedef struct {
  const uint16_t* x; const uint8_t* W; const float* w_scale; void* out;
  const uint16_t* resid; const uint16_t* norm_w;
  float norm_eps;
  int M, N, K;
  int epilogue;
} p3v_gemv_fp8_args_t;
int p3v_gemv_fp8(const p3v_gemv_fp8_args_t* args /* host */, void* stream);
int p3v_dequant_fp8(const uint8_t* w8, const float* w_scale, uint16_t* out_bf16, int rows, int K, void* stream);

/* ---- W8A8 projection on the fp8 matrix cores (BASELINE config 5, prompt-sized inputs; the same call sites:
 * QuantizedLinear, phi_3_vision_mlx.py:264,291-305, under `quantize_model=True`).
 *   out[M, N] = epilogue((a_scale[m] * w_scale[n]) * sum_k A8[m,k] * W8[n,k])     A8 [M, lda], W8 [N or 2N, ldw]: e4m3 bytes
 * on v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 1.0).  epilogue: P3V_EPI_NONE, P3V_EPI_RESID_BF16 (resid bf16 [M, ldo]),
 * P3V_EPI_SILU_MUL (W8 / w_scale hold the N gate rows then the N up rows; out [M, N]).  K % 128 == 0, N % 256 == 0
 * (128 for SILU_MUL), 16-byte aligned pointers and row strides.
 * p3v_quant_fp8_rows makes A8 / a_scale from bf16 rows: one scale s per row = max|h| / 448, codes e4m3(h * (1 / s)),
 * h = x or, with norm_w,
 * h = bf16(x * rsqrt(mean x^2 + eps) * norm_w) (nn.RMSNorm, phi.py:478-479), so the norm costs no extra pass. */
typedef struct {
  const uint8_t* A; const float* a_scale; const uint8_t* W; const float* w_scale;
  uint16_t* out; const uint16_t* resid;
  int M, N, K, lda, ldw, ldo, epilogue;
} p3v_gemm_fp8_args_t;
int p3v_gemm_fp8(const p3v_gemm_fp8_args_t* args /* host */, void* stream);
int p3v_quant_fp8_rows(const uint16_t* x, const uint16_t* norm_w /* optional */, float eps, uint8_t* q, float* scale,
                       int rows, int K, void* stream);

/* ---- SuRoPE tables, phi.py:487-504: cos/sin[n_pos, half] = {cos,sin}(pos*inv_freq)*scale (fp32) */
int p3v_rope_table(const float* pos, const float* inv_freq, float scale, float* cos_out, float* sin_out,
                   int n_pos, int half_dim, void* stream);

/* ---- split + RoPE + KV append, phi.py:443-452 and 542-548.
 * qkv [B*L, (nh+2*nkv)*hd] bf16 -> q_out [B, nh, L, hd] bf16 (rotated),
 * K (rotated) -> k_dst[b, h, dst_off + l, :]   (K   layout [B, nkv, dst_t, hd]),
 * V           -> v_dst[b, h, :, dst_off + l]   (V^T layout [B, nkv, hd, dst_t]: the PV product
 * then reads 8 consecutive keys per lane with one 16-byte load, like QK^T does for K).
 * cos/sin tables are [B/tab_div, tab_t, hd/2]; position of (b,l) is past+l.
 * cos_t == sin_t == NULL: no rotation, plain head split (CLIP q/k/v, phi.py:147).
 * `d_past` (device int32, may be null) overrides `past` -- used under graph replay.
 * q_scale: rotated queries are multiplied by it BEFORE their one rounding to bf16 (phi.py:454 scales q before the
 * product too); 1.0f = plain.  With q_scale = scale * log2(e) the attention call takes q_prescaled = 1 and its softmax
 * needs no multiply per score.  Applies to the plain head split as well (q = bf16(q * q_scale), bit for bit; the ViT path
 * does not use it: tests only).  Prompt-sized calls (L >= 32) run the split / rotation and the V transpose as ONE launch. */
int p3v_rope_kv_append(const uint16_t* qkv, const float* cos_t, const float* sin_t,
                       uint16_t* q_out, uint16_t* k_dst, uint16_t* v_dst,
                       int B, int L, int n_heads, int n_kv, int hd,
                       int past, const int32_t* d_past, int dst_t, int dst_off_is_past,
                       int tab_t, int tab_div, float q_scale, void* stream);

/* ---- a residual projection whose consumer is an RMSNorm, as one launch sequence (round 5; phi.py:478-484: `r + o_proj(...)` followed
 *      by `post_attention_layernorm`, `r + down_proj(...)` followed by the next layer's `input_layernorm`):
 *      p3v_gemm_resid_norm(g, w, eps, normed) == p3v_gemm(g) with P3V_EPI_RESID_BF16 + p3v_rmsnorm(g->out, w, normed, M, N, eps), bit
 *      for bit, where the projection runs as K slices (9 .. 256 rows and N small enough that the split pays: the fp32 partials are
 *      summed, the residual added, the row normalised by the SAME launch).  P3V_ERR_UNSUPPORTED -- nothing launched, the caller runs the
 *      two calls -- for every other shape, without a workspace of p3v_gemm_ws_bytes(), or N > 3072.  normed: bf16 [M, N] contiguous. */
int p3v_gemm_resid_norm(const p3v_gemm_args_t* gemm /* host */, const uint16_t* norm_w, float eps, uint16_t* normed, void* stream);

/* ---- the qkv projection with the split, the rotation and the KV append in its EPILOGUE (round 5):
 *      p3v_gemm_qkv(g, s) == p3v_gemm(g) into a scratch [M, N] + p3v_rope_kv_append(scratch, ...) with s's arguments, bit for bit, in the
 *      GEMM's launches alone (phi.py:437-452 / 140-147: `qkv_proj` + `split` + `_rotate_half` + KVCache append; CLIP: q/k/v projections
 *      + head split).  g: A [B*L, K], W [(nh + 2 nkv) * hd, K] (q rows, k rows, v rows), epilogue P3V_EPI_NONE or P3V_EPI_BIAS; out /
 *      resid / ws unused.  P3V_ERR_UNSUPPORTED (nothing launched) unless hd % 32 == 0, (hd / 2) % 16 == 0, an 8-aligned append
 *      offset and dst_t (round 6: batch rows of any length L -- CLIP's 577-token crops), and either M >= 1024 with whole 128-column tiles per region (the 256 x 256 /
 *      128 x 128-tile kernels) or 9 <= M <= 256 without bias, whole 64-column Q / K tiles and 128-row V tiles (the 128 x 64-tile
 *      weight-streaming kernel) -- callers then run the two calls. */
typedef struct {
  const float* cos_t; const float* sin_t;          /* as p3v_rope_kv_append; both null: plain head split */
  uint16_t* q_out; uint16_t* k_dst; uint16_t* v_dst;
  int B, L, n_heads, n_kv, hd;
  int past, dst_t, dst_off_is_past, tab_t, tab_div;
  float q_scale;
} p3v_qkv_split_t;
int p3v_gemm_qkv(const p3v_gemm_args_t* gemm /* host */, const p3v_qkv_split_t* split /* host */, void* stream);

/* ---- attention, phi.py:454-457 (decoder, causal + left-pad, Mask4D phi.py:550-563)
 * and phi.py:148 (CLIP, no mask).  q [B, nh, L, hd]; keys/values come from two
 * segments: positions [0,past) from (k_past,v_past) batch row b/past_div
 * (K [.., past_t, hd], V^T [.., hd, past_t]), positions [past,past+L) from
 * (k_new,v_new) batch row b (K [.., new_t, hd], V^T [.., hd, new_t]) -- or, with
 * new_is_cache, from (k_past,v_past) as well (the normal case: rows were just
 * appended to the cache).  V^T columns beyond the valid keys must hold finite values.
 * out [B, L, nh*hd] bf16.
 * Query i sees key t iff (!causal || t <= past+i) && t >= pad_len[b]; a query
 * that is itself padding outputs 0.  hd in {64, 96}. */
typedef struct {
  const uint16_t* q;
  const uint16_t* k_past; const uint16_t* v_past;
  const uint16_t* k_new;  const uint16_t* v_new;
  uint16_t* out;
  const int32_t* pad_len;   /* [B/past_div... indexed by b/pad_div] or null */
  const int32_t* d_past;    /* device override of `past` or null */
  float* ws;                /* split-KV workspace (decode path) or null */
  int B, L, n_heads, n_kv, hd;
  int past, past_t, past_div, new_t, pad_div;
  int causal;
  float scale;
  int n_split;              /* decode path: KV splits (<=1: no split) */
  int new_is_cache;         /* 1: (k_new,v_new) ignored, rows [past,past+L) are read from (k_past,v_past) too --
                               used with d_past, when the append offset is only known on the device */
  int q_prescaled;          /* 1: q already carries scale * log2(e) (p3v_rope_kv_append's q_scale); `scale` is then unused */
} p3v_attn_args_t;
int p3v_attention(const p3v_attn_args_t* args /* host */, void* stream);
/* bytes of workspace p3v_attention needs for the decode (L<=P3V_DECODE_MAX_L) path */
int64_t p3v_attention_ws_bytes(int B, int L, int n_heads, int hd, int n_split);
#define P3V_DECODE_MAX_L 16

/* ---- fused decode-step attention (L <= P3V_DECODE_MAX_L new tokens, no beams): head split +
 * _rotate_half + KVCache append + split-KV attention + merge in one call (phi.py:443-457, 542-548).
 * qkv [B*L, (nh+2*nkv)*hd] is the raw projection; positions [past, past+L) of the caches
 * (K [B, nkv, cache_t, hd], V^T [B, nkv, hd, cache_t], cache_t % 64 == 0) are written,
 * positions [0, past) read.  cos_t/sin_t point at the rows of the NEW positions: row
 * (b, r) = b*rope_bstride + r, r in [0, L) (either a view into the prompt tables at `past`, or
 * the compact buffers p3v_stage_rope fills when `past` only lives on the device);
 * ws >= p3v_attention_ws_bytes(B, L, nh, hd, n_split) even when n_split == 1.  hd == 96.
 * Key ranges of the splits are a static function of cache_t, so loads start before d_past arrives.
 * `past` with d_past != NULL: a LOWER BOUND of *d_past known when the launch is recorded (a captured decode step: the prompt
 * length), or negative for "none".  The 128-key kernel requests tiles below the bound at once; tiles at or beyond it wait
 * for *d_past and fetch nothing when they lie wholly past the live keys (the capacity is prompt + max_tokens). */
typedef struct {
  const uint16_t* qkv; const float* cos_t; const float* sin_t;
  uint16_t* k_cache; uint16_t* v_cache; uint16_t* out;
  const int32_t* pad_len; const int32_t* d_past; float* ws;
  int B, L, n_heads, n_kv, hd, past, cache_t, rope_bstride, n_split;
  float scale;
  int merge_in_launch; /* nonzero: the split-KV partials are merged by the last split of each (b, head) INSIDE the attention
                          launch; 0 = a separate merge launch.  Either way `ws` must hold 0xFF in every byte before its first use
                          (hipMemset) and every launch leaves it so: a partial word is valid when it is not the all-ones pattern */
  /* optional (round 4): the layer's o_proj + residual in the SAME launch (phi.py:460, 483), for shapes
   * p3v_attention_decode_can_fuse_oproj() accepts.  o_proj_w [o_n, n_heads * hd] bf16; o_proj_x [o_n] bf16 is the residual row,
   * updated in place: x += bf16(W_o . attention output), bit-identical to p3v_gemv with P3V_EPI_RESID_BF16 on `out`.  `out` must
   * hold 0xFF in every byte when the launch starts (the projecting workgroups poll its words); `o_rearm` (n_heads * hd bf16) is
   * set to 0xFF by the launch: callers alternate two output buffers between consecutive layers.  NULL o_proj_w = plain attention. */
  const void* o_proj_w; uint16_t* o_proj_x; uint16_t* o_rearm; int o_n;
  /* ... or on MLX 4-bit group-64 weights (the device layout p3v_gemv_q4 takes): o_proj_w = W4 [o_n, n_heads * hd / 8] u32 and
   * o_proj_sb != NULL = its scale | bias words [o_n, n_heads * hd / 64]; bit-identical to p3v_gemv_q4 with P3V_EPI_RESID_BF16. */
  const uint32_t* o_proj_sb;
} p3v_attn_decode_args_t;
int p3v_attention_decode(const p3v_attn_decode_args_t* args /* host */, void* stream);
/* Does p3v_attention_decode take `o_proj_w` for this shape on this device (host query, no launch)?  0 = no; 1 = attention + o_proj +
 * residual in ONE launch (merge_in_launch = 1, the 128-key one-tile plan); 2 (round 6) = merge_in_launch = 0: the launch that merges
 * the split-KV partials also carries the o_proj + residual (long contexts, 64-key plans) -- same arguments, same bits. */
int p3v_attention_decode_can_fuse_oproj(int B, int L, int n_heads, int hd, int n_split, int cache_t, int o_n, int merge_in_launch);
/* Which role workgroup `wg` of the fused launch's 1-D grid (n_heads * n_split workgroups) plays under placement `map` (the
 * attn_fo_map tuning knob: 2 = default, 1 = the round's first form): out[0] = key split, out[1] = head, out[2] / out[3] = its
 * projection units (rows [8 u, 8 u + 8) of o_proj; -1 = none).  Split n_split - 1 of a head is its merging workgroup.  Workgroups
 * L and L +- 256 share a CU on this part (round-robin dispatch), which is what the placement is built on: mergers sit alone with
 * one tile-only workgroup, projection units ride on the last workgroup of every other CU.  Returns P3V_ERR_UNSUPPORTED for shapes
 * the placement does not cover (the launch then uses the (split, head) grid).  Host-only, no GPU needed: tests/test_abi.py. */
int p3v_attention_decode_fused_role(int wg, int n_heads, int n_split, int o_n, int map, int* out4);
/* cos/sin rows of positions [past, past+L) of each batch row ([B, tab_t, half] tables) -> [B, L, half] */
int p3v_stage_rope(const float* cos_t, const float* sin_t, int past, const int32_t* d_past,
                   float* cos_out, float* sin_out, int B, int L, int tab_t, int half_dim, void* stream);

/* ---- int8 KV cache (quantize_cache=True; replaces the 4-bit prompt cache of phi.py:528-540).
 * Bytes are offset-binary u = round(x/s)+128 with one fp32 scale per (batch row, kv head, token):
 * K8 [B*nkv, dst_t, hd], V8^T [B*nkv, hd, dst_t], scales [B*nkv, dst_t].
 * p3v_kv_quantize converts rows [t0, t0+n_tok) of a bf16 K [BH, src_t, hd] / V^T [BH, hd, src_t] pair. */
int p3v_kv_quantize(const uint16_t* k, const uint16_t* vt, uint8_t* k8, uint8_t* v8t, float* k_scale, float* v_scale,
                    int BH, int hd, int src_t, int dst_t, int t0, int n_tok, void* stream);
/* p3v_attention_decode on the int8 cache: same contract; the step's own new rows are quantised,
 * appended, and attended in their quantised form (one representation per key, whenever it is read). */
/* the inverse, tokens [0, n_tok) of every (batch row, kv head): bf16 K [BH, dst_t, hd] / V^T [BH, hd, dst_t] = (code - 128) * scale.
 * For cached calls with more than 16 new tokens on the int8 cache (they attend through p3v_attention on this copy). */
int p3v_kv_dequantize(const uint8_t* k8, const uint8_t* v8t, const float* k_scale, const float* v_scale, uint16_t* k,
                      uint16_t* vt, int BH, int hd, int src_t, int dst_t, int n_tok, void* stream);

/* ---- the reference's OWN quantised cache, as an opt-in (load(..., quantize_cache=True, cache_format="mlx4")): phi.py:528-540 -- on the
 * first call mx.quantize(keys.reshape(B*N, -1), group_size=32) (4 bits): every token's hd values are hd / 32 affine groups; later
 * calls attend on mx.dequantize of them, later tokens stay unquantised.  Tokens [0, n_tok) of a bf16 K [BH, cache_t, hd] /
 * V^T [BH, hd, cache_t] pair -> codes k4 / v4 [BH, n_tok, hd / 32, 4] (uint32, MLX's packing: code k of a word at bits [4k, 4k+4)),
 * k_sb / v_sb [BH, n_tok, hd / 32, 2] fp32 (scale, bias), and the rows REWRITTEN IN PLACE with scale * q + bias (one rounding to
 * bf16): what every later call of the reference attends on.  hd % 32 == 0.
 * qkv != null: the keys are quantised from their EXACT fp32 values, as the reference's are (RoPE promotes k to fp32, phi.py:451) --
 * recomputed from the layer's projection output qkv [B * n_tok, (n_heads + 2 n_kv) * hd] (the first call: L = n_tok) and the rotation
 * tables (positions past + t), with p3v_rope_kv_append's arithmetic; null: from the cache rows (already rounded to bf16). */
int p3v_kv_quantize_mlx4(uint16_t* k, uint16_t* vt, uint32_t* k4, uint32_t* v4, float* k_sb, float* v_sb, int BH, int hd,
                         int cache_t, int n_tok, const uint16_t* qkv, const float* cos_t, const float* sin_t, int n_heads, int n_kv,
                         int past, int tab_t, int tab_div, void* stream);

typedef struct {
  const uint16_t* qkv; const float* cos_t; const float* sin_t;
  uint8_t* k8; uint8_t* v8t; float* k_scale; float* v_scale; uint16_t* out;
  const int32_t* pad_len; const int32_t* d_past; float* ws;
  int B, L, n_heads, n_kv, hd, past, cache_t, rope_bstride, n_split;
  float scale;
  int merge_in_launch;    /* as p3v_attn_decode_args_t.merge_in_launch (honoured with one tile per split and n_split <= 16 on
                             the 64-key plan, always on the 128-key plan; otherwise the merge launch runs) */
  /* optional (round 4), as p3v_attn_decode_args_t's o_proj_* fields but for e4m3 weights: o_proj_w8 [o_n, n_heads * hd] bytes with
   * one fp32 scale per row; x += bf16(scale * (W8 . attention output)), bit-identical to p3v_gemv_fp8 with P3V_EPI_RESID_BF16.
   * For shapes p3v_attention_decode_q8_can_fuse_oproj() accepts; `out` all 0xFF on entry, `o_rearm` set to 0xFF. */
  const uint8_t* o_proj_w8; const float* o_proj_scale; uint16_t* o_proj_x; uint16_t* o_rearm; int o_n;
} p3v_attn_decode_q8_args_t;
int p3v_attention_decode_q8(const p3v_attn_decode_q8_args_t* args /* host */, void* stream);
int p3v_attention_decode_q8_can_fuse_oproj(int B, int L, int n_heads, int hd, int n_split, int cache_t, int o_n, int merge_in_launch);

/* ---- CLIP patch unfold: pixel_values [N,3,S,S] f32 -> patches [N*P, kpad] bf16, (c,ky,kx) order, zero padded */
int p3v_im2col_patches(const float* pix, uint16_t* patches, int n_img, int img, int patch, int kpad, void* stream);
/* ---- CLS rows: x[n, 0, :] = class_emb + pos[0]  (phi.py:202-205); x [N, P+1, D] f32 */
int p3v_clip_cls_rows(float* x, const uint16_t* cls, const uint16_t* pos, int n_img, int tokens, int dim, void* stream);

/* ---- HD merge, phi.py:403-407: ViT features [n_crops, P+1, C] f32 (CLS at row 0 skipped)
 * -> rows [n_out, 4C] bf16 in the reference's order: sub crops (h*12 rows of
 * w*12 tokens, each row followed by sub_GN), glb_GN, global crop (12 rows of 12 + sub_GN). */
int p3v_hd_merge(const float* feats, const uint16_t* sub_gn, const uint16_t* glb_gn, uint16_t* out,
                 int h_crops, int w_crops, int grid /* 24 */, int C, void* stream);

/* ---- argmax over the last axis of bf16 logits (first maximum wins), phi_3_vision_mlx.py:386,392 */
int p3v_argmax(const uint16_t* logits, int32_t* out, int rows, int n, int64_t row_stride, void* stream);
/* ---- nn.log_softmax = x - logsumexp(x), phi_3_vision_mlx.py:92,476,511,541: bf16 io; the fp32 log-sum-exp is rounded to bf16
 *      (it is an array of its own in the reference) and the subtraction rounds once more */
int p3v_log_softmax(const uint16_t* x, uint16_t* y, int rows, int n, void* stream);
/* ---- top-k (k<=8) by (-value, index), replaces mx.argpartition phi_3_vision_mlx.py:507 */
int p3v_topk(const uint16_t* x, int32_t* idx_out, int rows, int n, int k, int64_t row_stride, void* stream);

/* ---- 4-bit group-64 affine weights = the reference's `quantize_model=True` (nn.quantize(model, 64, 4),
 * phi_3_vision_mlx.py:264,297-305; w = scale * q + bias per 64 input columns, mx.quantized_matmul).
 * W [N or 2N, K/8] u32 in the DEVICE nibble order (weights 8d..8d+7 at bits 0,16,4,20,8,24,12,28 of dword d;
 * weights.q4_repack converts MLX's sequential order), sb [N or 2N, K/64] u32 = scale bf16 | bias bf16 << 16.
 * p3v_gemv_q4: decode projection, K = 3072 or 8192.  M = 1: same norm / epilogue options as p3v_gemv.  2 <= M <= 16 (round 6: a
 *   decode batch, a constrained-decoding step; N % 16 == 0 (SiLU: % 8)): the packed weights are dequantised in registers (fp16
 *   scale * q + bias, two weights per instruction) in front of the fp16 MFMA -- 0.5 byte per weight from HBM at every batch size, as
 *   the reference's QuantizedLinear (phi_3_vision_mlx.py:296).  norm_w: up to 8 rows the input RMSNorm may ride along (the rows are
 *   normalised to the bf16 values p3v_rmsnorm writes while the first weights are in flight); 9 .. 16 rows: must be NULL
 *   (P3V_ERR_UNSUPPORTED otherwise -- the caller runs p3v_rmsnorm first).
 * p3v_dequant_q4: -> bf16 [rows, K] (prefill / batched decode run the bf16 kernels on a dequantised scratch). */
typedef struct {
  const uint16_t* x; const uint32_t* W; const uint32_t* sb; void* out;
  const uint16_t* resid; const uint16_t* norm_w; float norm_eps;
  int M, N, K;
  int epilogue;
} p3v_gemv_q4_args_t;
int p3v_gemv_q4(const p3v_gemv_q4_args_t* args /* host */, void* stream);
int p3v_dequant_q4(const uint32_t* w4, const uint32_t* sb, uint16_t* out_bf16, int rows, int K, void* stream);

/* ---- on-device image preprocessing (Phi3VImageProcessor, phi.py:283-372), bit-compatible with the host path.
 * p3v_resample_u8: one pass of Pillow's 8-bit ImagingResample (what `img.resize(..., Image.BILINEAR)` phi.py:301 runs):
 *   in [outer, in_len, inner] u8 -> out [outer, out_len, inner]; coeffs [out_len, ksize] 22-bit fixed point and
 *   bounds [out_len, 2] = (first input index, tap count) are computed on the host (processor.pil_bilinear_coeffs).
 *   Horizontal pass: outer = rows, inner = 3; vertical pass: outer = 1, inner = 3 * width.
 * p3v_hd_preprocess: resized image [rh, rw, 3] u8 -> pixel_values [n_slots, 3, 336, 336] f32: white padding of `top` rows
 *   above / (hp - rh - top) below (phi.py:302-306), transpose back if `portrait` (:307-308), normalisation through
 *   lut [3][256] f64 = (v / 255 - mean) / std (:309), crop grid into slots 1.. (:313-314), degenerate-bicubic global view
 *   into slot 0 with taps hw/ww [336][2] f32, hi/wi [336][2] i32 (:331-372), remaining slots zero (:315-316). */
int p3v_resample_u8(const uint8_t* in, uint8_t* out, int outer, int in_len, int out_len, int inner,
                    const int32_t* coeffs, const int32_t* bounds, int ksize, void* stream);
int p3v_hd_preprocess(const uint8_t* resized, int rh, int rw, int top, int hp, int portrait, const double* lut,
                      const float* hw, const int32_t* hi, const float* ww, const int32_t* wi, float* pixel_values,
                      int n_slots, void* stream);

/* ---- LoRA adapter inference, LoRALinear.__call__ (phi.py:129-133):
 *   y = linear(x); z = (x @ lora_a) @ lora_b; out = (y + scale*z).astype(bf16),  scale = cfg.scale * alpha / rank (phi.py:120)
 * lora_a [K, r] and lora_b [r, N] are the fp32 tensors of adapters.safetensors, r <= 64.
 * p3v_lora_down: t[M, r] (f32) = x[M, K] (bf16) @ lora_a.
 * p3v_lora_up:   v = bf16(y + scale * (t @ lora_b)) with y[M, N] the frozen projection's plain bf16 output, then the
 *   epilogue that projection would have carried: P3V_EPI_NONE out[M,N] = v; P3V_EPI_RESID_BF16 out = resid + v;
 *   P3V_EPI_SILU_MUL (N = 2*I, gate rows first) out[M, I] = silu(v[:, :I]) * v[:, I:]. */
int p3v_lora_down(const uint16_t* x, const float* lora_a, float* t, int M, int K, int r, void* stream);
int p3v_lora_up(const uint16_t* y, const float* t, const float* lora_b, float scale, int epilogue,
                const uint16_t* resid, uint16_t* out, int M, int N, int r, void* stream);

/* ---- decode-step helpers (device-resident loop state for graph replay) */
int p3v_add_i32(int32_t* x, int n, int delta, void* stream);
/* history[b, *d_step] = tok[b]; if tok_next != NULL also tok_next[b] = tok[b] (feeds the next replayed step) */
int p3v_store_token(const int32_t* tok, int32_t* history, const int32_t* d_step, int32_t* tok_next,
                    int B, int max_steps, void* stream);
/* fused head of a replayed greedy step: x_out[b] = table[tok[b]] (p3v_embed_gather, phi.py:597) and the rotation
 * rows of position *d_past -> cos_out/sin_out [B, 1, half] (p3v_stage_rope with L = 1) */
int p3v_step_begin(const int32_t* tok, const uint16_t* table, uint16_t* x_out, const float* cos_t, const float* sin_t,
                   const int32_t* d_past, float* cos_out, float* sin_out, int B, int hidden, int vocab, int tab_t,
                   int half_dim, int32_t* zero_buf /* optional: n_zero int32 cleared (in-launch flags of the step) */,
                   int n_zero, void* stream);
/* fused tail: next_tok[b] = tok[b] = argmax(logits[b]) (phi_3_vision_mlx.py:392), history[b, *d_step] = it,
 * then *d_step += 1 and *d_past += 1 (done once, by the last workgroup; `ticket` is a zero-initialised int32) */
int p3v_step_end(const uint16_t* logits, int32_t* next_tok, int32_t* tok, int32_t* history, int32_t* d_step,
                 int32_t* d_past, int32_t* ticket, int B, int n, int max_steps, void* stream);

/* Round 6: the same two stages WITHOUT launches of their own -- folded into the step's first projection (layer 0's RMSNorm + qkv)
 * and its last (final norm + lm_head, phi.py:597-608, phi_3_vision_mlx.py:386-393): p3v_gemv with
 *   begin (tok != NULL):      x row b = embed_table[clamp(tok[b])] instead of args->x; the rows are also written to x_out (the residual
 *                             stream) and the rotation rows of position *d_past staged into cos_out / sin_out, as p3v_step_begin does;
 *   end (next_tok != NULL):   next_tok[b] = tok_out[b] = argmax(out[b]) (first maximum of the bf16 values; a NaN row reports -1),
 *                             history[b, *d_step] = it, then *d_step += 1, *d_past += 1 -- by the last workgroup to finish (every
 *                             workgroup publishes its candidates into amax_ws and counts itself in: one arrival counter per XCD, then
 *                             one for the eight group-last workgroups -- a thousand workgroups finishing together on ONE counter
 *                             serialise for ~10 us; amax_ws: P3V_GEMV_STEP_WS_BYTES bytes, 8-byte aligned: P3V_GEMV_STEP_MAX_WG
 *                             candidate records (contents irrelevant) + nine counters, 128 bytes apart, that must be ZERO before
 *                             the first launch and are left zero by every launch; `ticket` is not touched).
 * Exactly one of the two.  M = 1 row (the kernel p3v_gemv runs there), bf16 weights, K = 3072 or 8192, P3V_EPI_NONE: otherwise
 * P3V_ERR_UNSUPPORTED and the caller
 * keeps p3v_step_begin / p3v_step_end.  Bit-identical to the separate launches. */
#define P3V_GEMV_STEP_MAX_WG 1024
#define P3V_GEMV_STEP_WS_BYTES (P3V_GEMV_STEP_MAX_WG * 8 + 9 * 128)
typedef struct {
  const int32_t* tok; const uint16_t* embed_table; int vocab; uint16_t* x_out;
  const float* cos_t; const float* sin_t; float* cos_out; float* sin_out; int tab_t, half_dim;
  int32_t* next_tok; int32_t* tok_out; int32_t* history; int32_t* d_step; int32_t* ticket; float* amax_ws; int max_steps;
  int32_t* d_past;            /* begin: read (position of the rows to stage); end: incremented */
} p3v_gemv_step_t;
int p3v_gemv_step(const p3v_gemv_args_t* args /* host */, const p3v_gemv_step_t* step /* host */, void* stream);
/* the same on e4m3 weights (config 5: `args` as for p3v_gemv_fp8, one row, no epilogue) */
int p3v_gemv_fp8_step(const p3v_gemv_fp8_args_t* args /* host */, const p3v_gemv_step_t* step /* host */, void* stream);
/* and on MLX 4-bit group-64 weights (`args` as for p3v_gemv_q4, one row, no epilogue) */
int p3v_gemv_q4_step(const p3v_gemv_q4_args_t* args /* host */, const p3v_gemv_step_t* step /* host */, void* stream);

/* ---- hipGraph helpers: capture a sequence of the launches above and replay it */
int p3v_graph_begin(void* stream);
int p3v_graph_end(void* stream, void** graph_exec_out /* host */);
int p3v_graph_launch(void* graph_exec, void* stream);
int p3v_graph_destroy(void* graph_exec);

/* ---- timing with HIP events on a given stream (bench.py roofline leg) */
int p3v_event_create(void** ev /* host */);
int p3v_event_record(void* ev, void* stream);
int p3v_event_elapsed_ms(void* start, void* stop, float* ms /* host */);
int p3v_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* P3V_H */
