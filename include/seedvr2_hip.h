/* libseedvr2_hip.so - C ABI of the MI355X (gfx950) SeedVR2 hot path.
 *
 * The reference (disk02/ComfyUI-SeedVR2 v2.5.18) is 100 % Python and has no FFI: every
 * entry point below replaces an ATen call (or a Python loop over ATen calls) that the
 * reference issues on its hot path; the reference call site each one replaces is cited
 * as  [ref: file:line]  relative to /root/reference/src.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr());
 *     the library never allocates, frees or synchronises; scratch is passed in explicitly.
 *   - every call is asynchronous on `stream` (a hipStream_t, pass torch's current stream).
 *   - returns 0 on success, a negative SVR2_E_* code otherwise; svr2_last_error() returns
 *     a thread-local message for the last failing call of the calling thread.
 *   - activations are bf16, row-major, channels-last; accumulation is fp32;
 *     per-channel vectors (bias, modulation, norm weights) are fp32.
 *   - safe to call from several host threads / streams and on several devices of one process (kernel
 *     attributes are applied per device; launches go to the device `stream` belongs to, which must be the
 *     calling thread's current device).  The only process-wide state is the table of tuning switches below,
 *     resolved once (svr2_set_option, else the environment variable of the same name at first use).
 */
#ifndef SEEDVR2_HIP_H
#define SEEDVR2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVR2_ABI_VERSION 2

enum {
    SVR2_OK = 0,
    SVR2_E_SHAPE = -1,  /* inconsistent sizes / unsupported geometry */
    SVR2_E_ALIGN = -2,  /* pointer or leading dimension not 16-byte aligned */
    SVR2_E_ARG = -3,    /* null pointer / bad enum */
    SVR2_E_ARCH = -4,   /* device is not gfx950 */
    SVR2_E_LAUNCH = -5  /* HIP launch error */
};

int svr2_abi_version(void);
const char* svr2_last_error(void);
/* Checks that device `dev` is gfx950 and returns its CU count (>0) or an error code. */
int svr2_device_check(int dev);
/* Tuning switches for A/B measurements and tests (never needed in production): name is one of SVR2_HALO_WR,
 * SVR2_GEMM_V, SVR2_GEMM4, SVR2_GEMM_BM, SVR2_GEMM2_NARROW, SVR2_CONV_LINEAR_TILES, SVR2_ATTN_V (1 = the plain window-attention kernel, bit-identical to the default), SVR2_ATTN5_V (0 = attn_d512 without the
 * software-pipelined fragment schedule, bit-identical),
 * SVR2_HALO_PREF (0 = halo conv without the fragment prefetch, bit-identical), SVR2_CONV_TFAST (frame-fastest conv tile order), SVR2_CONV_SPLIT (0: never split a conv along K); unset != 0 returns
 * the switch to its built-in default.  Switches never change results beyond fp32 summation order. */
int svr2_set_option(const char* name, int value, int unset);

/* ------------------------------------------------------------------ GEMM (+ fused epilogues)
 * C[M,N] = epilogue( A[M,K] * W[N,K]^T ),  bf16 in, fp32 accumulate, bf16 out.
 * Replaces nn.Linear / F.linear on the DiT path
 *   [ref: models/dit_3b/nablocks/attention/mmattn.py:56-59,173,269; mlp.py:56-61;
 *         patch/patch_v1.py:96,114; nadit.py:114-118]
 * and the 1x1x1 convolutions of the VAE [ref: models/video_vae_v3/modules/attn_video_vae.py:96-98].
 *   W must be allocated with n_pad = round_up(N,128) rows (rows >= N zero), K % 64 == 0.
 *   a_rows (optional): A row used for output row m is a_rows[m] (fused window_reverse gather,
 *   [ref: models/dit_3b/na.py:637-638]).
 * Epilogues:
 *   SVR2_EPI_BIAS       C = acc + bias                     (bias may be NULL)
 *   SVR2_EPI_GATE_RES   C = resid + gate * (acc + bias)    AdaSingle 'out' gate + residual add fused
 *                        [ref: modulation.py:112-116, nablocks/mmsr_block.py:112-113,125-126]
 *   SVR2_EPI_SWIGLU     W rows interleaved [8 gate | 8 in] per 16; C[M,N/2] = silu(g) * i
 *                        [ref: models/dit_3b/mlp.py:60-61]
 *   SVR2_EPI_GELU       C = gelu_tanh(acc + bias)          7B MLP [ref: models/dit_7b/mlp.py:28-43]
 *   SVR2_EPI_PIXSHUF3D  Upsample3D: 1x1x1 conv C->C*r then 'b (x y z c) f h w -> b c (f z)(h x)(w y)',
 *                        optional remove_head [ref: attn_video_vae.py:135-143,152-153;
 *                        causal_inflation_lib.py:412-419]; A rows are (f,h,w) raster, C is [F',2H,2W,C].
 */
enum { SVR2_EPI_BIAS = 0, SVR2_EPI_GATE_RES = 1, SVR2_EPI_SWIGLU = 2, SVR2_EPI_PIXSHUF3D = 3, SVR2_EPI_GELU = 4 };

typedef struct svr2_gemm_desc {
    int64_t M, N, K;
    const void* A;         /* bf16 [*, lda] */
    int64_t lda;
    const int32_t* a_rows; /* optional [M] */
    const void* W;         /* bf16 [n_pad, K] */
    void* C;               /* bf16 */
    int64_t ldc;
    int32_t epilogue;
    const float* bias;     /* [N] or NULL */
    const float* gate;     /* [N]      (GATE_RES) */
    const void* resid;     /* bf16 [M, ldr] (GATE_RES) */
    int64_t ldr;
    /* PIXSHUF3D geometry: A rows = f*(ph*pw) + h*pw + w */
    int32_t ps_frames, ps_h, ps_w, ps_c, ps_tz, ps_drop_head;
    const void* zero_page; /* >= 256 bytes of zeros (device) */
} svr2_gemm_desc;

int svr2_gemm(const svr2_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------ causal 3-D convolution
 * Implicit-GEMM NDHWC causal conv3d of the video VAE.
 *   [ref: models/video_vae_v3/modules/causal_inflation_lib.py:62-305 (InflatedCausalConv3d:
 *    temporal padding replaced by causal head padding / carried frames), :84-113 (Conv3d)]
 *   x      bf16 [T, Hi, Wi, Cin]  current temporal chunk
 *   carry  bf16 [npad, Hi, Wi, Cin] previous chunk's last (kt - st) frames, or NULL:
 *          NULL => the head is padded with `npad` copies of frame 0 (first chunk: 2*pad_t)
 *   w      bf16 [cout_pad, kt, Cin/64, kh, kw, 64]  (cout_pad = round_up(Cout,128), extra rows zero): 64-channel
 *          slices innermost, spatial taps next, so one slice's taps hit the same cache lines back to back
 *   y      bf16 [To, Ho, Wo, Cout],  To = (T + npad - kt)/st + 1, Ho = (Hi + pad_h0 + pad_h1 - kh)/sh + 1 ...
 *   resid  optional bf16 [To,Ho,Wo,Cout] added after bias (ResnetBlock3D skip, attn_video_vae.py:360)
 *   Cin % 64 == 0 (small-Cin layers go through svr2_im2col3d + svr2_gemm).
 *
 *   w_layout = SVR2_WL_TILED_128x32 selects the halo-resident kernel (input halo of a pixel patch kept in LDS,
 *   the nine spatial taps are shifted LDS reads) for the kh = kw = 3, stride-1, pad-1 convs; it needs
 *   Cin % 32 == 0 and w as bf16 [cout_pad/128][kt][Cin/32][kw][kh][128][32] (kx outer, ky inner: the three ky
 *   tiles of one kx column are 24 contiguous KiB) where each [128][32] tile is stored in LDS-image order: the
 *   16-byte chunk at (row r, slot s) holds channels 8*(s ^ f(r)) .. +8, f(r) = {0,2,3,1}[(r >> 2) & 3]
 *   (every weight DMA is then one linear KiB).
 *
 *   w_layout = SVR2_WL_CIN3 selects the fused kernel of the encoder's conv_in (conv_in.hip): Cin = 3, Cout = 128,
 *   3x3x3, stride 1, zero padding 1, npad = 2, no residual; x and carry are bf16 [.., Hi, Wi, 3] (6 bytes per pixel),
 *   w is bf16 [128][128] with k = (kt*3 + ky)*12 + kx*4 + c (c = 3 and k >= 108 zero).  gn_partial as for the tiled layout
 *   (same tile geometry).  Replaces svr2_im2col3d + svr2_gemm for this layer: no column matrix is written.
 */
enum { SVR2_WL_ROWS = 0, SVR2_WL_TILED_128x32 = 1, SVR2_WL_CIN3 = 2 };

typedef struct svr2_conv3d_desc {
    const void* x;
    const void* carry;
    const void* w;
    const float* bias;
    const void* resid;
    void* y;
    int32_t T, Hi, Wi, Cin, Cout;
    int32_t kt, kh, kw, st, sh, sw;
    int32_t npad;             /* causal head frames (carry frames or replicas of frame 0) */
    int32_t pad_h0, pad_w0;   /* zero padding before; padding after is implied by Ho/Wo */
    int32_t To, Ho, Wo;
    const void* zero_page;
    int32_t w_layout;         /* SVR2_WL_ROWS (0) or SVR2_WL_TILED_128x32 (1) */
    float* gn_partial;        /* optional (SVR2_WL_TILED_128x32 only): fp32 [To][tiles][2][Cout] per-tile per-channel
                                 (sum, sum of squares) of the bf16 outputs, tiles = svr2_conv3d_gn_tiles(Ho, Wo, Cout);
                                 feeds svr2_groupnorm_finalize so the following GroupNorm needs no statistics pass */
    void* split_ws;           /* optional (SVR2_WL_TILED_128x32 only): >= svr2_conv3d_split_ws_bytes(d) bytes of scratch, 16-byte */
    int64_t split_ws_bytes;   /* aligned.  With it, a conv whose grid would leave most of the chip idle (latent-resolution
                                 512-channel convs of a small image: 16-128 tiles, 48 channel slices) is split 2-8 ways along
                                 K; a second launch adds the fp32 partial tiles in a fixed order and finishes them.  NULL or
                                 too small: never split.  Split and unsplit results differ by fp32 summation order only. */
    int32_t carry_frames;     /* frames in `carry`: 0 or npad = all npad head frames are given; 1 <= carry_frames < npad =
                                 carry holds only the LAST carry_frames of the head and its first frame also stands for the
                                 missing earlier ones - the state after a first chunk shorter than the head (one frame:
                                 the reference's memory is then [f0, f0], causal_inflation_lib.py:266-278, 422-437), passed
                                 as a view instead of a materialised copy */
} svr2_conv3d_desc;

int svr2_conv3d(const svr2_conv3d_desc* d, void* stream);
/* Scratch bytes svr2_conv3d would use for this descriptor (pointers are not read); 0 = this shape is not split. */
int64_t svr2_conv3d_split_ws_bytes(const svr2_conv3d_desc* d);
int32_t svr2_conv3d_gn_tiles(int32_t Ho, int32_t Wo, int32_t Cout);

/* Explicit im2col for the small-Cin causal convs (Cin = 3 or 16): col[M, kpad] with
 * k = ((kt*kh+ky)*kw+kx)*Cin + c, zero padded to kpad (multiple of 64). */
int svr2_im2col3d(const svr2_conv3d_desc* d, void* col, int64_t kpad, void* stream);

/* ------------------------------------------------------------------ norms / modulation
 * y = x * rsqrt(mean(x^2) + eps) * scale + shift   per row; scale/shift fp32 [dim] (shift may be NULL).
 * Fuses CustomRMSNorm (no affine) with AdaSingle mode 'in' whose per-channel factors are constants at
 * inference [ref: models/dit_3b/normalization.py:88-109, modulation.py:110-111,
 *            nablocks/mmsr_block.py:108-110,115-123, nadit.py:234-244]. dim % 8 == 0, dim <= 8192. */
int svr2_rmsnorm_mod(const void* x, void* y, const float* scale, const float* shift,
                     int64_t rows, int32_t dim, float eps, void* stream);

/* ------------------------------------------------------------------ attention preparation
 * Builds window-ordered Q,K,V for the joint window attention from the fused qkv projections:
 *   - gathers the video rows of every window (window partition) and appends the text rows
 *     to every window          [ref: models/dit_3b/na.py:228-229,386-393; mmattn.py:195-217]
 *   - per-head RMSNorm of q,k with weight [ref: mmattn.py:207-208]
 *   - rotary embedding in fp32, window-local video coordinates (Lt+t, h, w), text token p at (p,p,p)
 *     [ref: models/dit_3b/rope.py:118-126,171-173]  (rope_mode 1 = 3B 'lang' mm-rope, 2 = 7B 'pixel'
 *     video-only rope with per-position angles, 0 = none)
 *   qkv_vid [L, 3*H*D], qkv_txt [Lt, 3*H*D]  (layout (o h d), mmattn.py:201-202)
 *   src_row [S]: >= 0 video row, < 0 text row (-1 - j);   pos [S*3] int32 (t,h,w) rope coordinates
 *   q,k,v out: [S, H, D].  D must be 128.
 *   rope_cos/rope_sin: fp32 [max_pos, nfreq] tables (3B: nfreq = 21), or per-position [S, 3*nfreq] (7B)
 */
typedef struct svr2_qkv_prep_desc {
    const void* qkv_vid;
    const void* qkv_txt;
    const int32_t* src_row;
    const int32_t* pos;
    const float* qn_vid; const float* kn_vid; /* [D] */
    const float* qn_txt; const float* kn_txt;
    const float* rope_cos; const float* rope_sin;
    void* q; void* k; void* v;
    int64_t S;
    int32_t H, D, nfreq, max_pos, rope_mode;
    float eps;
} svr2_qkv_prep_desc;

int svr2_qkv_prep(const svr2_qkv_prep_desc* d, void* stream);

/* ------------------------------------------------------------------ varlen window attention
 * out[s, h, :] = softmax(q[s,h,:] . k[seq(s),h,:]^T / sqrt(D)) v[seq(s),h,:], non-causal, per sequence.
 * Replaces the per-window SDPA loop / flash_attn_varlen_func
 *   [ref: models/dit_3b/attention.py:24-61,65-92,138-157].
 *   q,k,v,out bf16 [S, H, 128];  cu_seqlens int32 [nseq+1];
 *   tile_seq/tile_q0 int32 [ntiles]: sequence id and first query row (multiple of 128 within the
 *   sequence) of every 128-query tile (host-built work list).
 */
int svr2_window_attn(const void* q, const void* k, const void* v, void* out,
                     const int32_t* cu_seqlens, const int32_t* tile_seq, const int32_t* tile_q0,
                     int32_t ntiles, int32_t H, int32_t D, float scale, void* stream);
/* Same, with the result of position s written to row out_rows[s] of `out` (int32 [S], a permutation of 0..S-1): lets
 * the caller get the video rows back in natural token order (window_reverse, mmattn.py:260-264) so that the
 * following projection is a dense GEMM instead of a row gather. */
int svr2_window_attn_scatter(const void* q, const void* k, const void* v, void* out, const int32_t* out_rows,
                             const int32_t* cu_seqlens, const int32_t* tile_seq, const int32_t* tile_q0,
                             int32_t ntiles, int32_t H, int32_t D, float scale, void* stream);

/* Mean over windows of the replicated text rows [ref: models/dit_3b/na.py:396-417]:
 * out[j, :] = mean_w attn[txt_row0[w] + j, :],  attn bf16 [S, dim]. */
int svr2_txt_mean(const void* attn, const int32_t* txt_row0, int32_t nwin, int32_t Lt, int32_t dim,
                  void* out, void* stream);

/* ------------------------------------------------------------------ patch in / one-step endpoint
 * rows[L, kpad] = patchify(cat[noise, latent, 1]) with patch (1,2,2), feature order (h w c), zero padded
 * to kpad [ref: core/infer.py:54-78,340; models/dit_3b/patch/patch_v1.py:86-93].
 * noise, latent: bf16 [T, H, W, C]. */
int svr2_patch_in(const void* noise, const void* latent, void* rows, int32_t T, int32_t H, int32_t W,
                  int32_t C, int32_t kpad, void* stream);
/* x0[T,H,W,C] = noise - unpatchify(pred[L, 4*C])
 * [ref: patch_v1.py:119-124; common/diffusion/schedules/base.py:108-110 (v_lerp endpoint at t=T)]. */
int svr2_patch_out_x0(const void* pred, int64_t ldp, const void* noise, void* x0, int32_t T, int32_t H,
                      int32_t W, int32_t C, void* stream);

/* ------------------------------------------------------------------ VAE: per-frame GroupNorm (+SiLU)
 * GroupNorm over (H*W, C/G) of every frame, channels-last x [F, P, C] (P = H*W pixels)
 *   [ref: models/video_vae_v3/modules/causal_inflation_lib.py:354-409 (causal_norm_wrapper:
 *    'b c t h w -> (b t) c h w' then F.group_norm, eps 1e-6), attn_video_vae.py:316-318,344,350]
 * svr2_groupnorm_stats: two launches (fp32 partial sums per strip, fp64 fold) -> coef [F, C, 2] with
 *   y = x * coef[..,0] + coef[..,1];  workspace: svr2_groupnorm_ws_floats(F, P, C) floats.
 * svr2_groupnorm_apply: y = x*a + b, optionally followed by SiLU (the nonlinearity that follows
 *   every norm in ResnetBlock3D / conv_norm_out). */
int64_t svr2_groupnorm_ws_floats(int32_t F, int64_t P, int32_t C);
int svr2_groupnorm_stats(const void* x, int32_t F, int64_t P, int32_t C, int32_t G, float eps,
                         const float* gamma, const float* beta, float* workspace, float* coef, void* stream);
/* svr2_groupnorm_finalize: the second launch of svr2_groupnorm_stats alone, on partial sums [F][nblk][2][C] that a
 *   producer already wrote (svr2_conv3d's gn_partial). */
int svr2_groupnorm_finalize(const float* partial, int32_t nblk, int32_t F, int64_t P, int32_t C, int32_t G, float eps,
                            const float* gamma, const float* beta, float* coef, void* stream);
/* carry_out (optional): the last carry_frames frames of y are also stored to carry_out [carry_frames, P, C] - the
 *   causal-conv state of the conv that consumes y [ref: causal_inflation_lib.py:266-278], saving a copy pass. */
int svr2_groupnorm_apply(const void* x, const float* coef, void* y, int32_t F, int64_t P, int32_t C,
                         int32_t silu, void* carry_out, int32_t carry_frames, void* stream);

/* ------------------------------------------------------------------ VAE: mid-block spatial attention
 * Single-head attention with head dim 512 over the P = h*w tokens of every latent frame
 *   [ref: attn_video_vae.py:615-631,661-665 (diffusers Attention, heads=1, dim_head=512, SDPA)].
 *   q,k,v: bf16 rows of 512 with row stride ld (elements; a fused [S,1536] qkv buffer works),
 *   frame f owns rows [f*P, (f+1)*P);  out bf16 [frames*P, ldo]. */
int svr2_attn_d512(const void* q, const void* k, const void* v, int64_t ld, void* out, int64_t ldo,
                   int32_t frames, int32_t P, float scale, void* stream);


/* ------------------------------------------------------------------ pixel-space stages
 * Antialiased bicubic resize (separable taps computed by the host exactly as ATen's
 * _upsample_bicubic2d_aa does), clamp to [0,1], zero-pad bottom/right to (Hp, Wp), (x-0.5)/0.5
 *   [ref: core/generation_utils.py:47-84; data/image/transforms/side_resize.py:36-75,
 *         divisible_crop.py:43-80].  in bf16 [T,Hin,Win,3] -> out bf16 [T,Hp,Wp,3] (channels-last).
 *   wy [Hout,KY], ymin/ycnt [Hout]; wx [Wout,KX], xmin/xcnt [Wout]. */
int svr2_resize_pad_normalize(const void* in, void* out, const float* wy, const int32_t* ymin,
                              const int32_t* ycnt, const float* wx, const int32_t* xmin, const int32_t* xcnt,
                              int32_t T, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t Hp,
                              int32_t Wp, int32_t KY, int32_t KX, void* stream);
/* The resize alone (result rounded to bf16, no clamp / pad / normalise): first pass of the reference's two-pass resize when
 * `max_resolution` caps the long edge -- it resizes to the short-edge size, then resizes THAT bf16 image again
 *   [ref: data/image/transforms/side_resize.py:61-73].  out bf16 [T,Hout,Wout,3]. */
int svr2_resize_bicubic_aa(const void* in, void* out, const float* wy, const int32_t* ymin, const int32_t* ycnt,
                           const float* wx, const int32_t* xmin, const int32_t* xcnt, int32_t T, int32_t Hin, int32_t Win,
                           int32_t Hout, int32_t Wout, int32_t KY, int32_t KX, void* stream);
/* out = prev_tail * w_prev[t] + cur_head * (1 - w_prev[t]) per frame t of the temporal overlap
 *   [ref: core/generation_utils.py:284-312; generation_phases.py:968-986]. */
int svr2_blend_overlap(const void* prev_tail, const void* cur_head, void* out, const float* w_prev,
                       int64_t frame_elems, int32_t overlap, void* stream);
/* Crop [T,Hp,Wp,3] -> [T,H,W,3]; to_unit != 0 also applies clamp(-1,1)*0.5+0.5
 *   [ref: core/generation_phases.py:942-958, 1337]. */
int svr2_crop_to_unit(const void* in, void* out, int32_t T, int32_t Hp, int32_t Wp, int32_t H, int32_t W,
                      int32_t to_unit, void* stream);

/* ------------------------------------------------------------------ phase-4 colour correction
 * Channels-last bf16 frames [F, H, W, 3] in [-1, 1]; content = decoded sample, style = transformed input of the same
 * frames [ref: src/core/generation_phases.py:1290-1304, src/utils/color_fix.py].  The reference runs these in the
 * compute dtype (bf16): the kernels round to bf16 after the same ops.
 *
 * svr2_wavelet_level: one level of the dilated-Gaussian pyramid [ref: color_fix.py:122-184]:
 *   low = conv3x3(img; 1/16 [1 2 1; 2 4 2; 1 2 1], dilation = radius, replicate padding);
 *   high = (first ? 0 : high) + img - low   (high may be NULL).  radius = min(2^level, max(1, min(H,W)/8)).
 * svr2_wavelet_merge: out = clamp(high + low, -1, 1)                         [ref: color_fix.py:241-246]
 * svr2_adain: per (frame, channel) (x - mean_c) / std_c * std_s + mean_s      [ref: color_fix.py:72-119]
 * svr2_rgb_to_lab: planar fp32 lab[3][n] of clamp((x+1)/2, 0, 1), D65         [ref: color_fix.py:322-329,368-411]
 * svr2_hist_match: out[rank-i element of source] = i-th smallest of reference [ref: color_fix.py:473-513]
 *   (device radix sort; equal source values receive consecutive reference quantiles in index order - the
 *    reference leaves that order unspecified)
 * svr2_lab_to_rgb: L = Lc*lw + Lm*(1-lw) (Lm may be NULL), lab -> rgb -> bf16 [-1,1]   [ref: color_fix.py:335-361,414-470]
 */
int svr2_wavelet_level(const void* img, void* low, void* high, int32_t F, int32_t H, int32_t W, int32_t radius,
                       int32_t first, int32_t is_f32, void* stream);
int svr2_wavelet_merge(const void* high, const void* low, void* out, int64_t n, int32_t is_f32, void* stream);
int64_t svr2_adain_ws_bytes(int32_t F, int64_t P);
int svr2_adain(const void* content, const void* style, void* out, int32_t F, int64_t P, void* workspace, void* stream);
int svr2_rgb_to_lab(const void* x, float* lab, int64_t n, void* stream);
int64_t svr2_hist_match_ws_bytes(int64_t n);
int svr2_hist_match(const float* source, const float* reference, float* out, int64_t n, void* workspace,
                    int64_t ws_bytes, void* stream);
int svr2_lab_to_rgb(const float* L_content, const float* L_matched, const float* a, const float* b,
                    float luminance_weight, void* y, int64_t n, void* stream);
/* 'hsv' and 'wavelet_adaptive' [ref: color_fix.py:516-860] (is_f32 above: the adaptive mode runs its wavelet in fp32).
 * svr2_rgb_to_hsv: planar fp32 hsv[3][n] of clamp((x+1)/2, 0, 1); equal maxima resolve b over g over r like the
 *   reference's masked assignments.
 * svr2_hue_sat_match: content saturation with, per 30-degree hue bin holding more than min_pixels pixels in both
 *   images, the style saturation distribution moved onto the content ranks (unequal counts: reference quantile
 *   long(linspace(0,1,n_src)[rank] * (n_ref - 1))); bin 0 also takes hues >= 11/12, which bin 11 then overwrites.
 *   thresholds (host): fp32(k/12) for k = 0..12, then fp32(1 - 1/12).
 * svr2_hsv_to_rgb: (h, s, v) -> rgb, clamp, *2-1, stored as bf16 or fp32.
 * svr2_sat_blend: out = wav (1 - w) + hsv w, w = sigmoid(5 ((sat(content) - sat(style)) - 0.15)) where
 *   sat(wav) - sat(style) > 0.075, else 0. */
int svr2_rgb_to_hsv(const void* x, float* hsv, int64_t n, void* stream);
int64_t svr2_hue_sat_match_ws_bytes(int64_t n);
int svr2_hue_sat_match(const float* hsv_c, const float* hsv_s, float* out, int64_t n, const float* thresholds,
                       int32_t min_pixels, void* workspace, int64_t ws_bytes, void* stream);
int svr2_hsv_to_rgb(const float* h, const float* s, const float* v, void* y, int64_t n, int32_t out_f32, void* stream);
int svr2_sat_blend(const void* content, const void* style, const float* wav, const float* hsv, void* out, int64_t n,
                   void* stream);

/* ------------------------------------------------------------------ streaming frame I/O
 * The bytes that cross PCIe are the container's 8-bit pixels; both conversions run on the device and replace the
 * reference's host-side cv2/numpy chains.
 * svr2_u8_to_frames: packed 8-bit rows -> bf16 [T,H,W,3] in [0,1] = bf16(fp16(fp32(u8) / 255))
 *   [ref: inference_cli.py:527-548 (_read_frames_from_cap: BGR->RGB, /255), :627 (.to(float16));
 *    generation_phases.py:376-384 (-> compute dtype)].  Source pixel (t, y, x) at
 *    src + t*frame_stride + (bottom_up ? H-1-y : y)*row_stride + 3*x.
 * svr2_frames_to_u8: bf16 [T,H,W,3] in [0,1] -> packed rows, u8 = trunc(x * 255) [ref: inference_cli.py:520, 691, 734
 *   ((frames * 255.0).astype(np.uint8)), :661-664 (RGB->BGR)]; row padding (row_stride - 3W <= 3 bytes) is zeroed.
 * svr2_frames_to_yuv: the same quantisation, then the 8-bit integer BT.601 studio-range matrix
 *   Y = ((66R+129G+25B+128)>>8)+16, U = ((-38R-74G+112B+128)>>8)+128, V = ((112R-94G-18B+128)>>8)+128; planar frames
 *   [Y: H*W][U][V] with chroma subsampling sub = 1 (4:4:4) or 2 (4:2:0, chroma of the rounded 2x2 mean of RGB).  This is
 *   the picture a BGR24 -> yuv420p scaler hands the reference's mp4v encoder (cv2.VideoWriter, inference_cli.py:693-706);
 *   the arithmetic of that scaler lives in a dependency absent from the reference tree: parity unpinned, defined here.
 * svr2_yuv_to_frames: inverse matrix R = clip((298(Y-16)+409(V-128)+128)>>8) ..., chroma replicated, then the uint8
 *   chain of svr2_u8_to_frames. */
#define SVR2_PIX_SWAP_RB 1   /* memory order B,G,R */
#define SVR2_PIX_BOTTOM_UP 2 /* first row in memory is the bottom image row (AVI DIB) */
#define SVR2_PIX_ALPHA 4     /* four channels per pixel on both sides (RGBA / BGRA <-> [T,H,W,4]); alpha is converted
                                like a colour channel [ref: inference_cli.py:262-270, 656-664] */
int svr2_u8_to_frames(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int64_t row_stride,
                      int64_t frame_stride, int32_t flags, void* stream);
int svr2_frames_to_u8(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int64_t row_stride,
                      int64_t frame_stride, int32_t flags, void* stream);
int svr2_frames_to_yuv(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int32_t sub, int64_t frame_stride,
                       void* stream);
int svr2_yuv_to_frames(const void* src, void* dst, int32_t T, int32_t H, int32_t W, int32_t sub, int64_t frame_stride,
                       void* stream);

/* ------------------------------------------------------------------ spatial VAE tiling
 * Blend of overlapping tile results with the roundings of the reference's in-place bf16 torch ops
 * [ref: src/models/video_vae_v3/modules/attn_video_vae.py:1403-1452 (tiled_encode), :1565-1614 (tiled_decode)].
 * svr2_tile_accumulate: tile bf16 [T,th,tw,Ct] (channels 0..C-1 and rows/columns 0..eh-1 / 0..ew-1 used), result bf16
 *   [T,H,W,C], count bf16 [H,W], fade weights wh bf16 [eh], ww bf16 [ew] (device):
 *     result[t,y0+y,x0+x,c] += bf16(bf16(tile * wh[y]) * ww[x]);   count[y0+y,x0+x] = bf16(count + wh[y]*ww[x]).
 * svr2_tile_normalize: result = bf16(result / max(count, bf16(1e-6)))  (result.div_(count.clamp(min=1e-6))). */
int svr2_tile_accumulate(const void* tile, int32_t T, int32_t th, int32_t tw, int32_t Ct, void* result, void* count,
                         int32_t H, int32_t W, int32_t C, int32_t y0, int32_t x0, int32_t eh, int32_t ew, const void* wh,
                         const void* ww, void* stream);
int svr2_tile_normalize(void* result, const void* count, int32_t T, int32_t H, int32_t W, int32_t C, void* stream);

/* ------------------------------------------------------------------ RGBA inputs: edge-guided alpha upscaling (next-4)
 * [ref: src/core/alpha_upscaling.py:124-430 (detect_edges_batch, guided_filter_pytorch, edge_guided_alpha_upscale);
 *  call site src/core/generation_phases.py:1132-1190].  Planes are fp32 [T,H,W]; rgb is the blended upscaled video,
 * channels-last bf16 [T,H,W,3] in [-1,1] (before colour correction).  hip.alpha_upscale() chains them.
 * svr2_alpha_stats: counts[0] = |{alpha < 0.1}|, counts[1] = |{alpha > 0.9}| (binary-mask test, :325-330) and
 *   rgb_min[0] = min(rgb) (the two "if x.min() < 0: x = (x+1)/2" branches, :346-348 and :147-149); workspace >= 4 KiB.
 * svr2_alpha_guide: guide = mean over channels of the (optionally normalised) rgb in fp32 (:217-218) and the uint8 grey
 *   image of cv2.cvtColor(RGB2GRAY) on uint8(clip(x*255)): (4899 R + 9617 G + 1868 B + 8192) >> 14.
 * svr2_alpha_sobel: mag = sqrt(sx^2 + sy^2) in float64, 3x3 Sobel with BORDER_REFLECT_101 (cv2.Sobel(CV_64F, ksize=3)),
 *   frame_max[t] (8 bytes each) = max over the frame.  svr2_alpha_edges: uint8(mag / max * 255) / 255 (:167-177).
 * svr2_resize_plane: F.interpolate(bicubic, antialias=True) of a fp32 plane, tap tables as svr2_resize_pad_normalize.
 * svr2_guided_ab / svr2_guided_out: _apply_guided_filter (:233-290), box filter = avg_pool2d(2r+1, 1, r).
 * svr2_alpha_refine: binary != 0: steps 3-8 of the binary-mask branch (:372-410); then clamp(0,1) (:427). */
int svr2_alpha_stats(const void* alpha, int64_t n_alpha, const void* rgb, int64_t n_rgb, int32_t* counts, float* rgb_min,
                     void* workspace, void* stream);
int svr2_alpha_guide(const void* rgb, void* gray, float* guide, int64_t npix, int32_t norm_guide, int32_t norm_edges,
                     void* stream);
int svr2_alpha_sobel(const void* gray, double* mag, void* frame_max, int32_t T, int32_t H, int32_t W, void* stream);
int svr2_alpha_edges(const double* mag, const void* frame_max, float* edges, int32_t T, int64_t HW, void* stream);
int svr2_resize_plane(const float* in, float* out, const float* wy, const int32_t* ymin, const int32_t* ycnt,
                      const float* wx, const int32_t* xmin, const int32_t* xcnt, int32_t T, int32_t Hin, int32_t Win,
                      int32_t Hout, int32_t Wout, int32_t KY, int32_t KX, int32_t clamp01, void* stream);
int svr2_guided_ab(const float* guide, const float* src, float* a, float* b, int32_t T, int32_t H, int32_t W,
                   int32_t radius, float eps, void* stream);
int svr2_guided_out(const float* a, const float* b, const float* guide, float* out, int32_t T, int32_t H, int32_t W,
                    int32_t radius, void* stream);
int svr2_alpha_refine(const float* refined, const float* edges, float* out, int32_t T, int32_t H, int32_t W,
                      int32_t binary, void* stream);

/* ------------------------------------------------------------------ baseline JPEG for Motion-JPEG output (next-3)
 * The reference's writer is cv2's mp4v encoder [ref: inference_cli.py:666-706]; without a codec library the compressed
 * container offered here is MJPEG, encoded on the device (arithmetic defined by oracle/seedvr2_ref/jpeg.py, ITU-T T.81).
 * frames bf16 [T,H,W,3] in [0,1]; div int32 [2][64] = 8*Q in zig-zag order (luma, chroma); huff uint32 [2][12+256] =
 * length<<16 | code for the DC categories then the AC run/size symbols (tables: comfyui-seedvr2_amd/jpeg.py).
 * packed uint8 [T][frame_cap] receives each frame's entropy-coded scan (MCU rows separated by RSTn, one restart interval
 * per MCU row; header and EOI are the caller's), frame_len int32 [T] its length or -1 if it did not fit. */
int64_t svr2_jpeg_ws_bytes(int32_t T, int32_t H, int32_t W);
int svr2_jpeg_encode(const void* frames, int32_t T, int32_t H, int32_t W, const int32_t* div, const uint32_t* huff,
                     void* packed, int64_t frame_cap, int32_t* frame_len, void* workspace, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ optional noise injection (node inputs / CLI flags)
 * x, y, out bf16 [n]; scalars as torch applies python floats / fp32 scalar tensors to bf16 tensors.
 *   mode 0: input_noise_scale [ref: src/core/generation_phases.py:409-425]: n = y*0.05 ; out = x*a + (x+n)*b with a = 1-s/2,
 *           b = s/2, every op rounded to bf16 (x = transformed video, y = randn_like(x)).
 *   mode 1: aug noise [ref: :679]: out = x*a + y*b rounded per op (a = 0.1, b = 0.05; x = base noise, y = second randn).
 *   mode 2: latent_noise_scale, schedule.forward with an fp32 timestep [ref: :684-693; schedules/base.py:82-87]:
 *           out = bf16(fp32(a*x) + fp32(b*y)), a = 1 - t/T, b = t/T. */
int svr2_noise_mix(const void* x, const void* y, void* out, int64_t n, float a, float b, int32_t mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEDVR2_HIP_H */
