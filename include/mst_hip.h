/* mst_hip.h - C ABI of libmst_hip.so: the MI355X (gfx950) hot path of the music-mixing-style-transfer
 * inference pipeline (FXencoder -> mean FX embedding -> FiLM-conditioned TCN "MixFXcloner", plus the
 * FX-manipulator processors).
 *
 * The reference (jhtonyKoo/music_mixing_style_transfer) is pure Python and has no native boundary; its
 * boundary for this path is the torch module API `from networks import FXencoder, TCNModel`
 * (inference/style_transfer.py:22,47-57,149,161).  The entry points below are what sits directly under
 * those modules' forward() in this build - each one names the reference code it replaces.  A binding
 * only needs raw pointers, sizes and a hipStream_t: there are no torch types in any signature.
 *
 * Conventions
 *   - every function returns MST_OK (0) or a negative MstStatus; nothing throws or exits;
 *     mst_last_error() returns a thread-local human readable message for the last failure.
 *   - "host" pointers are CPU memory in the REFERENCE's own tensor layouts (state_dict layouts);
 *     "dev" pointers are device memory owned by the caller.  The library owns only the opaque handle,
 *     its packed/folded weights and the FiLM factor table; forward() allocates nothing - the caller
 *     passes a workspace of mst_*_workspace_bytes().
 *   - all kernels are enqueued on the caller's stream and return immediately.
 *   - a handle belongs to one device and is not thread-safe (one stream at a time).
 */
#ifndef MST_HIP_H
#define MST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MST_OK = 0,
    MST_ERR_ARG = -1,          /* bad argument (null pointer, non-positive size, ...) */
    MST_ERR_UNSUPPORTED = -2,  /* configuration outside what the gfx950 kernels implement */
    MST_ERR_HIP = -3,          /* HIP runtime error (see mst_last_error) */
    MST_ERR_STATE = -4,        /* weights / condition not loaded before forward */
    MST_ERR_WORKSPACE = -5     /* workspace pointer null or too small */
} MstStatus;

/* arithmetic mode of the dense convolutions (TCN blocks 1..n-1, all FXencoder convs) */
typedef enum {
    MST_PREC_F32 = 0,   /* v_mfma_f32_32x32x2_f32, fp32 activations in HBM: the parity mode */
    MST_PREC_BF16 = 1,  /* bf16 operands / fp32 accumulate (TCN blocks: v_mfma_f32_16x16x32_bf16, FXencoder: v_mfma_f32_32x32x16_bf16), bf16 activations in HBM */
    MST_PREC_BF16X3 = 2 /* every fp32 operand split x = hi + lo into two bf16 values, three bf16 MFMAs per product
                         * (hi*hi + hi*lo + lo*hi): fp32-class accuracy at a third of the bf16 rate.  TCN: fp32 activations in HBM,
                         * <= 1e-4 on the waveform (measured 4e-6).  FXencoder: the channel-minor bf16 pipeline in the same split
                         * arithmetic (two bf16 planes per activation), <= 1e-4 * max|embedding| (measured 6e-6) - NOT the exact
                         * MST_PREC_F32 kernels: use MST_PREC_F32 where bit-level parity with the fp32 reference order matters. */
} MstPrecision;

#define MST_MAX_BLOCKS 32

int mst_version(void);
const char *mst_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * MixFXcloner: TCNModel (networks/architectures.py:76-174) / TCNBlock (:177-234) / FiLM
 * (networks/network_utils.py:156-182).
 * ---------------------------------------------------------------------------------------------- */
typedef struct MstTcn MstTcn;

typedef struct {
    int nblocks;                   /* TCNModel(nblocks)                      default cfg: 14  */
    int ninputs;                   /* input channels of block 0              default cfg: 2   */
    int noutputs;                  /* channels of the 1x1 output conv        default cfg: 2   */
    int channels;                  /* channel_width                          default cfg: 128 */
    int kernel_size;               /*                                        default cfg: 15  */
    int cond_dim;                  /* FiLM condition length                  default cfg: 2048 */
    int dilations[MST_MAX_BLOCKS]; /* dilation_growth ** (n % stack_size), architectures.py:122 */
    int causal;                    /* TCNBlock(causal=True): zero padding (k-1)*d on the left only - the reference pads both
                                    * sides by (k-1)*d and drops the last (k-1)*d outputs (architectures.py:199,230-231) */
} MstTcnDesc;

/* replaces TCNModel.__init__ (architectures.py:93-133): conditional blocks of dense convolutions (a grouped conv1 is handed
 * over as its block-diagonal dense weight).  The configs.yaml shape (channel_width 128, kernel_size 15, ninputs 2,
 * noutputs <= 2, non-causal - what inference/style_transfer.py:48-57 constructs) runs on the specialised kernels in every
 * precision; any other channel width / kernel size / input width / dilation / a causal net runs on a generic exact-fp32
 * implicit-GEMM path (the precision argument is then ignored).  A single TCNBlock is a net of one block run through
 * mst_tcn_forward_blocks. */
int mst_tcn_create(const MstTcnDesc *desc, MstTcn **out);
int mst_tcn_destroy(MstTcn *tcn);

/* replaces load_state_dict for blocks.{n}.* (style_transfer.py:94-108).  Host fp32, reference layouts:
 *   conv_w  [channels, cin, k]      blocks.n.conv1.weight (bias=False, architectures.py:201-207)
 *   bn_*    [channels]              blocks.n.bn.{weight,bias,running_mean,running_var}; eval-mode BN is
 *                                   folded into the conv weight + a per-channel shift
 *   film_w  [2*channels, cond_dim]  blocks.n.film.film_fc.weight ; film_b [2*channels]
 *   res_w   [channels]              blocks.n.res.weight ([channels,1,1]: grouped 1x1, groups=cin,
 *                                   architectures.py:216-220; block 0: out-channel o reads in-channel
 *                                   o / (channels/ninputs)) */
int mst_tcn_load_block(MstTcn *tcn, int n, const float *conv_w, const float *bn_weight, const float *bn_bias,
                       const float *bn_mean, const float *bn_var, float bn_eps, const float *film_w,
                       const float *film_b, const float *res_w, void *stream);
/* output.weight [noutputs, channels(,1)] , output.bias [noutputs]  (architectures.py:133) */
int mst_tcn_load_output(MstTcn *tcn, const float *w, const float *b, void *stream);

/* replaces FiLM.forward's film_fc(condition) for all blocks at once (network_utils.py:180-181): computes
 * the (r, b) factor table.  cond_dev: device fp32; n_rows = 1 (one embedding broadcast over the batch,
 * what style_transfer.py:161 passes) or B (one row per batch item).  block_stride (in floats) = 0 when
 * every block sees the same condition, else block n reads cond_dev + n*block_stride (the reference's
 * list-of-conditions branch, architectures.py:139-140). */
int mst_tcn_set_cond(MstTcn *tcn, const float *cond_dev, int n_rows, long block_stride, void *stream);

size_t mst_tcn_workspace_bytes(const MstTcn *tcn, int B, int L, int precision);

/* replaces TCNModel.forward (architectures.py:135-147): x_dev fp32 [B, ninputs, L] (NCL, contiguous) ->
 * y_dev fp32 [B, noutputs, L], clamped to [-1, 1].  Zero padding ((k-1)*d)//2 both sides per block. */
int mst_tcn_forward(MstTcn *tcn, const float *x_dev, float *y_dev, int B, int L, int precision,
                    void *workspace, size_t workspace_bytes, void *stream);
/* parity probe: run only the first n_run blocks and return that block's output activations as fp32
 * [B, channels, L] (NCL, the reference's layout) - the per-block hook the parity tests compare. */
int mst_tcn_forward_blocks(MstTcn *tcn, const float *x_dev, float *act_dev, int B, int L, int precision, int n_run,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---- Kernel-form switches at a glance (round 6) ------------------------------------------------------------------------------------------
 * None of them changes WHAT is computed; "=" bit-identical to the default, "~" equal up to fp32 summation order.  The library keeps no
 * process-wide switch: every one lives in a handle or in the arguments of one call.  Defaults are the measured-fastest forms.
 *
 *   where                          bit / field            default  selects                                                  result  why it is still here
 *   mst_tcn_set_tuning (handle)    bit 0                  1        bf16x3: 128-time tiles of <= 2 phases, 2 workgroups / CU   ~       256-time form needed where < 64 steps per phase
 *                                  bits 1-2 = 2           2        bf16: 256-time class-major tiles - the one-tile kernel     ~/=     form 0 (tap-major one-tile kernel) runs what neither takes (odd d, short segments) and is
 *                                                                  with bit 7, else the persistent duo kernel                         the other side of GPU / emulator tests
 *                                  bit 3, form 1          -        (removed kernels: rejected with MST_ERR_ARG)
 *                                  bit 4                  1        bf16 duo: class-major main loop                           ~       tap-major loop = the P = 1 path and the GPU test's other side
 *                                  bit 5                  1        bf16: block 0 inside the d = 2 block's launch             =       separate block-0 kernel = probes, other precisions, short segments
 *                                  bit 6                  1        bf16x3: class-major loop in the eight-phase half kernel    ~       other side of a GPU test
 *                                  bit 7                  1        bf16: two- / four-phase class-major blocks, one 256-time   =/~     the duo kernel is the other side of the bit-identity tests (emulator + GPU
 *                                                                  tile per workgroup, two workgroups per CU (round 6)                forms 181 / 53) and what bit 7 off selects
 *   mst_enc_set_tuning (handle)    rows_min_tiles         512      bf16: rows-resident conv kernel from this many tiles on    =       small layers run the im2col kernel
 *   mst_enc_set_schedule (handle)  bit 0                  1        weight-major workgroup order of weight-heavy layers        =
 *                                  bit 1                  0        2 x 2 wave tiling of the 128-channel kernel (slower)       ~       A/B record only
 *                                  bit 2                  0        fp32: 64-bit gather addresses                              =       the path > 4 GiB activations take anyway (test hook)
 *                                  bit 3                  0        stereo block as two direct launches                        =       reference form of the fused kernel's bit-identity test
 *                                  bit 4                  0        blocks 1 / 2 as two launches each                          ~       reference form of the fused kernel's test
 *                                  bit 5                  0        128-channel layers on the four-wave im2col kernel          ~       reference form of the raw-rows kernel's test; layers the raw-rows kernel cannot take
 *   MstFxFuse.forms (per call)     EQ_LANE_APPLY          0        stereo equaliser apply pass, one lane per chunk            =       reference form of a GPU / emulator test; non-stereo path
 *                                  EQ_VALU_ENDS           0        stereo equaliser state pass on VALU dot products           =       reference form; chunk lengths that are no multiple of 16
 *                                  COMP_SLICE_SMALL       0        compressor: three time slices whatever the size            =       lets small emulated problems take the pipelined path
 * (mst_fx_set_tuning - a process-wide switch - left the ABI in round 6.)
 * ------------------------------------------------------------------------------------------------------------------------------------------ */
/* tuning flags (choose between forms of the block kernels; flags = x3_small_tiles | bf16_form << 1 | bf16_reuse << 4 | bf16_fuse0 << 5 |
 * x3_half_cm << 6 | bf16_onetile << 7, default 245; bit 3 and form 1 named kernels that were measured slower and left the library in round 5 - they are rejected):
 * bit 0 (bf16x3 mode; default 1, measured 5.13 instead of 5.45 ms per launch at 32 x 131072): the split-bf16 block kernel on 128-time
 *   tiles of <= 2 phases (two workgroups per CU) wherever the segment has at least 64 steps per phase, 0 = 256-time tiles (one
 *   workgroup per CU); the two-phase 128-time tiles run the class-major loop (B fragment pairs reused by the two taps of a class: 4.62 ->
 *   4.09 ms per launch, round 4), the 256-time tiles the tap-major one: results agree to fp32 accumulation rounding (~1e-6).
 * bits 1-2 (bf16 mode), form of the dense block kernel - measured at 32 x 131072, profiles/r03_tcn_block_forms_summary.md:
 *   0 tcn_block_bf16_kernel: one tile per workgroup, two workgroups per CU                                              1.49-1.51 ms
 *   2 (default) the blocks with 256-time tiles of <= 4 phases on the class-major family: with bit 7 (default since round 6) the one-tile kernel at
 *     two workgroups per CU (1.28-1.33 ms, see bit 7), else tcn_block_bf16_duo_kernel (form 0 for the others and for the last
 *     block): persistent, one workgroup of 4 matrix waves + 4 loader waves per CU, two tile buffers, the next tile by LDS-DMA and
 *     the previous tile's row stores during the main loop; with bit 4 off bit-identical to form 0                        1.47-1.48 ms (1.40 with bit 4)
 *   (1 was tcn_block_bf16_stream_kernel, 1.67 ms; bit 3 tcn_block_bf16x3_duo_kernel, 5.45-5.6 ms per launch against 4.09: EXPERIMENTS.md)
 * bit 4 (bf16 mode, form 2; default 1): the duo kernel's main loop runs class-major - taps grouped by j mod (16 / phases), every B
 *   fragment read from LDS once per class and k-step and fed to up to eight MFMAs (304 instead of 960 LDS reads per tile at four phases).
 *   Same products, another fp32 summation order: agrees with bit 4 off to accumulation rounding (not bit-identical).  Measured at
 *   32 x 131072, same box, d = 4 ... 2048: 1.40 ms per launch against 1.46 (profiles/r04_tcn_forms_reuse.log).
 * bit 5 (bf16 mode, with form 2 and bit 4; default 1 since round 5): block 0 (2 -> 128 channels) is not launched - the loader waves of the d = 2 block's
 *   duo kernel compute its outputs straight into the LDS image (same arithmetic as tcn_block0_mfma_kernel: bit-identical results, checked
 *   on the MI355X at 32 x 131072; no 1.07 GB store and re-read).  Measured, same box, alternating (profiles/r04_tcn_forms_fuse0.log):
 *   the fused launch 1.58 ms against 0.31 + 1.44 for the two kernels, -0.2 ms per forward (1 % of the step).  Applies only when block 1 is
 *   the d = 2 block on two-phase class-major tiles (>= 128 steps per phase) and is not the last block; otherwise the separate block-0
 *   kernel runs - mst_tcn_get_tuning reports which happened.  Emulator tests (bit identity over random shapes) + tests/test_gpu_parity.py form 53.
 * bit 6 (bf16x3 mode; default 1 since round 5): the eight-phase half-tile kernel (d >= 4096 at L = 131072: 2 of that mode's 13 launches) runs a
 *   class-major loop too (pseudo-classes of two taps of one parity); results agree with bit 6 off to accumulation rounding (GPU test:
 *   <= 1e-5 on the waveform, both within 1e-4 of the oracle).  Measured, same box, alternating: 571.7 / 572.1 against 566.2 segments/s for the
 *   whole bf16x3 step at 32 x 131072 (profiles/r05_x3_ab_bit6_53_117.jsonl).
 * bit 7 (bf16 mode, with form 2 and bit 4; default 1 since round 6): the two- and four-phase blocks (d = 2 ... 2048 at L = 131072: 11 of the 13 dense launches)
 *   run tcn_block_bf16_kernel<P, false, 8, 2> - ONE 256-time tile per workgroup of four waves, TWO workgroups per CU, the duo kernel's
 *   class-major loop (same products, same order: bit-identical to bit 7 off) - instead of the persistent duo kernel: two matrix waves per SIMD
 *   cover each other's staging and epilogue.  With bit 5 the d = 2 block's workgroups compute block 0 in their staging phase (<2, false, 8, 2, true>:
 *   the duo loader's arithmetic, bit for bit).  Same box, alternating (profiles/r06_tcn_forms_onetile_ab.txt, r06_tcn_forms_onetile_fuse0_ab.txt):
 *   1.312-1.318 ms per launch against 1.404-1.409 for the duo kernel, the d = 2 launch with block 0 inside 1.47 against 1.56 (a 128-time form at
 *   three workgroups per CU: 1.336-1.338 - it streams every weight fragment twice as often, and under the chip's power limit a tile's energy is
 *   what counts; EXPERIMENTS.md E.6).  One case is NOT bit-identical to bit 7 off: a LAST block with two / four phases (segments of >= 2^19 samples: d = 8192
 *   has 64 steps per phase) carries the fused output head and ran the one-tile kernel's tap-major loop; with bit 7 it runs the class-major loop
 *   (<4, true, 8, 2>: 1.32 against 1.36 ms) - its activation differs by fp32 summation order, the waveform by <= 2e-3 after the bf16 re-rounding
 *   (profiles/r06_tcn_forms_2p19_segments.txt; both within the bf16 tolerance of the oracle and of the reference's real-audio goldens).
 *   Bit 7 also selects, for a block whose phase sequences are EXACTLY one 256-time tile (L = 64 d, 32 d or 16 d: d = 2048 / 4096 / 8192 at L = 131072), the
 *   unrolled forms <4 | 8 | 16, ., 8, 1>: no all-padding (column tile, tap) pair exists, the LDS image keeps only the halo steps a row window can straddle
 *   into (280 / 272 / 256 rows), the sixteen-phase form carries the fused head.  The four-phase form sums in the duo kernel's order (bit-identical); the
 *   other two in another fp32 order than round 5's 128-time forms (one bf16 ulp on the activation).  1.32 -> 1.26, 1.25 -> 1.19, 1.21 -> 1.03 ms
 *   (profiles/r06_tcn_whole_sequence_256_tiles_ab.txt). */
int mst_tcn_set_tuning(MstTcn *tcn, int flags);
/* the flags in force and whether the handle's LAST forward ran block 0 inside block 1's launch (bit 5 is a request: see its conditions above);
 * either pointer may be null. */
int mst_tcn_get_tuning(const MstTcn *tcn, int *flags, int *last_forward_fused_block0);

/* measurement hook (bench.py's roofline leg): between _begin and _end every mst_tcn_forward records HIP events
 * on its stream around each kernel; _end synchronises, writes the AVERAGE milliseconds per forward of
 * block kernels 0..nblocks-1 followed by the output kernel into ms_out[nblocks+1], and the number of
 * forwards seen into *n_forwards. */
int mst_tcn_timing_begin(MstTcn *tcn, int max_forwards);
int mst_tcn_timing_end(MstTcn *tcn, float *ms_out, int *n_forwards);

/* Calibration of the box (measurement infrastructure, bench.py "roofline.calib_ms"): runs the BARE main loop of the bf16 TCN block
 * kernel (csrc/tcn_kernels.h::tcn_calib_mainloop_kernel: the MFMAs, weight stream and LDS reads of one dense 32 x 131072 block launch =
 * 2.06 TFLOP; no staging, no epilogue, no store) `launches` times back to back on `stream` on synthetic operands with realistic
 * statistics; the first half settles the power controller, the second half is timed with HIP events.  *ms_per_launch = average duration
 * of a timed launch, *sclk_mhz = the shader clock the last launch ran at (s_memtime / s_memrealtime inside the kernel).  Allocates and
 * frees 1.1 MB of device memory; blocks until done.  No reference counterpart. */
int mst_calib_mainloop(int launches, float *ms_per_launch, float *sclk_mhz, void *stream);

/* ------------------------------------------------------------------------------------------------
 * FXencoder (networks/architectures.py:26-70) = Res_ConvBlock x N (network_utils.py:96-119), each two
 * Conv1d_layer (network_utils.py:15-89: ReflectionPad1d -> Conv1d -> BatchNorm1d(eval) -> ReLU), then
 * AdaptiveAvgPool1d(1).
 * ---------------------------------------------------------------------------------------------- */
typedef struct MstEnc MstEnc;

typedef struct {
    int nblocks;                       /* len(config["kernels"])                                   */
    int channels[MST_MAX_BLOCKS + 1];  /* [2] + config["channels"]  (architectures.py:30)           */
    int kernels[MST_MAX_BLOCKS];
    int strides[MST_MAX_BLOCKS];
    int dilations[MST_MAX_BLOCKS];
    int valid_padding;                 /* Conv1d_layer(padding="VALID"): no reflection padding, L_out = (L - (k-1)d - 1)/s + 1   */
    float act_slope;                   /* activation of every layer (network_utils.py:76-80) as the slope for negative values:
                                        * 0 = ReLU (config "relu", the zero-initialised default), 0.01 = LeakyReLU ("lrelu"), 1 = none */
} MstEncDesc;

int mst_enc_create(const MstEncDesc *desc, MstEnc **out);
int mst_enc_destroy(MstEnc *enc);
/* which = 0: encoder.{block}.conv1 (cin->cin, stride 1), 1: encoder.{block}.conv2 (cin->cout, stride s).
 * w [cout, cin, k]; bias [cout] or NULL; BN arrays [cout].  Host side: BN folding and the MFMA-fragment images of the layer (fp32, bf16, bf16 hi / lo, raw-rows),
 * packed on up to 16 short-lived host threads (one per 128-channel tile) before the synchronous upload; the default encoder's 81 M weights load in a few hundred ms. */
int mst_enc_load_conv(MstEnc *enc, int block, int which, const float *w, const float *bias,
                      const float *bn_weight, const float *bn_bias, const float *bn_mean, const float *bn_var,
                      float bn_eps, void *stream);
/* tuning (bf16 / bf16x3 modes): a conv layer (bf16 mode: with at most 64 input channels) whose launch has at least rows_min_tiles tiles keeps its
 * input rows resident in LDS (enc_conv_rows_kernel) instead of gathering an im2col slice per k-chunk; default 512, 0 = whenever a
 * layer qualifies (any channel count), negative = never.  Both forms produce identical bits. */
int mst_enc_set_tuning(MstEnc *enc, long rows_min_tiles);
/* workgroup order of the channel-minor conv kernel (bf16 / bf16x3 modes).  bit 0 (default on): layers with more weight bytes than
 * activation bytes run their workgroups in weight-major order - all column tiles of one (channel tile, k-slice) on one XCD, so a
 * weight slice crosses the fabric once instead of once per XCD.  bit 1 (default off: measured 7-12 % slower): the 128-channel x
 * 128-column tile with its waves 2 x 2 (enc_conv_nlc22_kernel: two MFMAs per LDS read, every weight fragment fetched by two waves).
 * bit 2 (exact-fp32 mode, a test hook): gather through 64-bit addresses - the path that activations beyond the 32-bit offset range take
 * by themselves - instead of buffer loads.  Same bits either way.  (Round 5 built and measured an in-kernel split-K finalize - tickets, last
 * workgroup of a tile sums the partial tiles - as bit 3: correct, and 5 x SLOWER per layer (the device-scope release fence writes the
 * whole L2 back on this part; EXPERIMENTS.md D.3): not in the library.)
 * bit 3 (default off, the reference form of a GPU / emulator test): the default encoder's stereo block (2 -> 2, k = 25 with skip; 2 -> 16, k = 25,
 * stride 4) as two direct-kernel launches with the intermediate in HBM instead of the fused enc_stereo_block_kernel.  Same bits either way.
 * bit 4 (default off, bf16 mode): blocks 1 and 2 of the default encoder (16 -> 16, k = 25 with skip; 16 -> 32, k = 25, stride 4 / 32 -> 32, k = 15;
 * 32 -> 64, k = 15, stride 2) as their two conv launches each instead of the fused enc_block1_fused_kernel (intermediate in LDS, weights resident in registers); same operands, another fp32 summation
 * order: the two forms agree to accumulation rounding (one bf16 ulp on isolated elements).
 * bit 5 (default off, bf16 mode): the 128-channel layers (kernel 5 / 10, stride 1 / 2, input channels a multiple of 64, output length a multiple
 * of 32) on the four-wave im2col kernel (enc_conv_nlc_kernel<4>) instead of enc_conv_taps_kernel (raw input rows staged once per 64-channel
 * block by LDS-DMA, loader + matrix waves, 256-column tiles, split-K over channel blocks); same operands, another fp32 summation order. */
int mst_enc_set_schedule(MstEnc *enc, int flags);
/* nn.AdaptiveAvgPool1d(1) on its own (architectures.py:63,67; FXencoder(conv_block='conv') runs its ConvBlocks one by one through
 * mst_enc_forward_conv and pools here): x_dev fp32 [rows, L] -> y_dev[rows] = mean over L. */
int mst_global_avgpool(const float *x_dev, float *y_dev, long rows, int L, void *stream);
/* Conv1d_layer(mode="deconv") (network_utils.py:24-26,38-42, nn.ConvTranspose1d): the zero-stuffed input of the equivalent stride-1
 * convolution - y_dev[row][pad_left + i * stride] = x_dev[row][i], zeros elsewhere; x_dev fp32 [rows, L] -> y_dev fp32 [rows, Lu],
 * Lu >= pad_left + (L - 1) * stride + 1.  The convolution itself is mst_enc_forward_conv of a handle loaded with the tap-reversed,
 * channel-transposed weights and VALID padding (networks/network_utils.py does both). */
int mst_enc_zero_stuff(const float *x_dev, float *y_dev, long rows, long L, int stride, long pad_left, long Lu, void *stream);
size_t mst_enc_workspace_bytes(const MstEnc *enc, int B, int L);
/* replaces FXencoder.forward (architectures.py:65-70): x_dev fp32 [B, 2, L] -> emb_dev fp32 [B, C_last].
 * precision: MST_PREC_F32 (exact fp32 MFMA, parity mode) or MST_PREC_BF16 (bf16 operands, fp32 accumulate;
 * activations stay fp32 in HBM). */
int mst_enc_forward(MstEnc *enc, const float *x_dev, float *emb_dev, int B, int L, int precision, void *workspace,
                    size_t workspace_bytes, void *stream);
/* parity probe: run only the first n_run Res_ConvBlocks; out_dev fp32 [B, channels[n_run], L_out(n_run)] */
int mst_enc_forward_blocks(MstEnc *enc, const float *x_dev, float *out_dev, int B, int L, int precision, int n_run,
                           void *workspace, size_t workspace_bytes, void *stream);
/* Conv1d_layer.forward on its own (network_utils.py:86-89): ONE conv of the handle - which = 0: encoder.{block}.conv1
 * (cin->cin, stride 1), 1: conv2 (cin->cout, stride s) - ReflectionPad1d -> Conv1d -> BatchNorm1d(eval) -> ReLU, exact fp32;
 * x_dev [B, cin, L] -> y_dev [B, cout, mst_enc_conv_length(...)].  Only that conv needs to be loaded. */
int mst_enc_conv_length(const MstEnc *enc, int block, int which, int L);
int mst_enc_forward_conv(MstEnc *enc, int block, int which, const float *x_dev, float *y_dev, int B, int L, void *stream);
/* output length of Res_ConvBlock `block` for input length L ("SAME" padding ignores the stride:
 * L_out = floor((L-1)/s)+1, network_utils.py:30-34,48-51) */
int mst_enc_block_length(const MstEnc *enc, int block, int L);

/* FiLM.forward on its own (network_utils.py:163-182): f = film_fc(cond) [rows, 2C]; y = f[:, :C] * x + f[:, C:] over x_dev
 * [B, C, L] (NCL); rows = 1 (broadcast) or B.  w_dev [2C, cond_dim], b_dev [2C], cond_dev [rows, cond_dim] are device fp32
 * (the module's parameters); table_dev: >= rows * 2C floats of caller scratch. */
int mst_film_forward(const float *w_dev, const float *b_dev, const float *cond_dev, int rows, int cond_dim, int C,
                     const float *x_dev, float *y_dev, int B, long L, float *table_dev, void *stream);

/* replaces torch.stack/reshape/mean(axis=0) over segment embeddings (style_transfer.py:152-153):
 * out[d] = mean over rows of emb[n_rows, dim], summed in row order (so the result does not depend on
 * how rows were sharded across GPUs before the all-gather). */
int mst_embedding_mean(const float *emb_dev, int n_rows, int dim, float *out_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * FX-manipulator processors (mixing_manipulator/common_audioeffects.py).  Audio layout as in the
 * reference processors: [n_items][L][C] time-major, interleaved channels, fp32.  One (item, channel)
 * sequence per lane for the serial recursions; float64 internal arithmetic like the reference.
 * ---------------------------------------------------------------------------------------------- */
/* Chain fusion (AugmentationChain.apply_processor :115-148 without its extra passes).  A processor entry point that takes an
 * MstFxFuse reads its input as x * (float)in_scale_dev[item] - the PENDING factor of the previous processor's rms-normalise,
 * applied in float32 exactly like the separate scale pass would - and leaves sum(y^2) of every item of its raw output in
 * out_sumsq_dev[item] (float64; the first launch of the call clears the slots, the caller need not; the imager writes its closed
 * form), so that an rms-normalise step costs
 * one tiny mst_fx_rms_pending launch instead of two energy passes and a scale pass over the audio.  fuse = NULL or both
 * members NULL: the plain processor.  mst_fx_scale_items applies a pending factor when a chain ends with one. */
#define MST_SUMSQ_SLOTS 64     /* an energy sum is kept as 64 partial sums per item (producers spread their atomics over them) */
#define MST_FX_FORM_EQ_LANE_APPLY 1   /* MstFxFuse.forms */
#define MST_FX_FORM_EQ_VALU_ENDS 2
#define MST_FX_FORM_COMP_SLICE_SMALL 4
typedef struct {
    /* = sizeof(MstFxFuse) of the header the CALLER was built against: an entry point given another value returns MST_ERR_ARG instead of
     * reading a struct of another layout (the struct has grown twice) */
    unsigned struct_size;
    /* kernel forms of this call - per call, not per process (round 6: the library keeps no mutable global state); results do not depend on
     * them.  mst_fx_biquad_cascade, stereo: MST_FX_FORM_EQ_LANE_APPLY = the apply pass one lane per chunk straight from global memory
     * (fx_biquad_chunk_kernel<true>) instead of 16-frame slabs through LDS (fx_biquad_stereo_apply_kernel), identical bits;
     * MST_FX_FORM_EQ_VALU_ENDS = the state pass as float64 VALU dot products with the impulse-state table in LDS
     * (fx_biquad_stereo_ends_kernel) instead of v_mfma_f64_16x16x4_f64 with the table as A fragments (fx_biquad_stereo_ends_mfma_kernel),
     * the same products added in the same (sample) order.  Both are the reference forms of a GPU test.  mst_fx_compressor:
     * MST_FX_FORM_COMP_SLICE_SMALL = cut the time-parallel compressor into its three time slices (map / apply kernels of neighbouring slices
     * on an internal low-priority side stream beside the chain kernel, events order map_i -> chain_i -> apply_i, the caller's stream joins
     * the side stream before the call returns) whatever the size of the batch (>= 8 chain batches) - large batches (>= 32 chain batches of
     * 1024 samples and >= 4e6 samples in all) always are; the hook lets small emulated problems take that path.  0: the defaults. */
    int forms;
    const double *in_scale_dev;   /* [n_items] or NULL */
    double *out_sumsq_dev;        /* [n_items][MST_SUMSQ_SLOTS] or NULL; overwritten */
    /* tail folding (mst_fx_midside_imager only; the other entry points refuse post_rms != 0): the rms-normalise that FOLLOWS the
     * processor and a Gain behind it happen in the processor's own last pass, y_out = (y * s) * post_gain in float32 - the values the
     * separate mst_fx_rms_pending + mst_fx_gain launches produce - with s from in_sumsq_dev = sum(x_raw^2) of the processor's raw input
     * ([n_items][MST_SUMSQ_SLOTS]) and the closed-form energy of y */
    const double *in_sumsq_dev;
    int post_rms;
    float post_gain;
    /* mst_fx_biquad_cascade only (the others refuse it): also leave sum(x_raw^2) of the call's raw input here
     * ([n_items][MST_SUMSQ_SLOTS], the arithmetic of mst_fx_sumsq) - the first rms-normalise of a chain then needs no energy pass over x */
    double *out_in_sumsq_dev;
    /* mid / side energies handed from mst_fx_compressor (stereo; the others refuse it) to mst_fx_midside_imager: the compressor leaves
     * sum((l + r)^2), sum((l - r)^2) of its raw output - float32 sums and squares per frame, float64 accumulation, the arithmetic of the
     * imager's own energy pass - as MST_SUMSQ_SLOTS pairs per item in out_ms_dev ([n_items][MST_SUMSQ_SLOTS][2]: slot s covers a 64th of the
     * frames, summed in a fixed order - deterministic since round 6), and an imager given the same array as in_ms_dev skips its energy pass
     * over the audio.  NULL: off. */
    double *out_ms_dev;
    const double *in_ms_dev;
} MstFxFuse;
int mst_fx_sumsq(const float *x_dev, int n_items, long per_item, double *out_dev, void *stream);   /* out[item][slot]: partial sums of x^2 */
/* scale_out[item] = float32(sqrt(mean(x_true^2) / max(1e-7, mean(y^2)))), mean(x_true^2) = scale_x^2 sum_slots(sumsq_x) / per_x
 * (scale_x NULL = 1); sumsq_x / sumsq_y in the [n_items][MST_SUMSQ_SLOTS] form */
int mst_fx_rms_pending(const double *scale_x_dev, const double *sumsq_x_dev, long per_x, const double *sumsq_y_dev, long per_y,
                       double *scale_out_dev, int n_items, void *stream);
int mst_fx_scale_items(const float *x_dev, float *y_dev, int n_items, long per_item, const double *scale_dev, void *stream);

/* Equaliser.process (:500-525): cascade of n_bands biquads, zero initial state per band; coef host
 * float64 [n_bands][6] = (b0,b1,b2,a0,a1,a2) shared by all items.  With a scratch buffer of
 * mst_fx_biquad_scratch_bytes() the cascade runs parallel in time (chunked state-space scan, same float64
 * per-sample recursion); with scratch_dev = NULL it runs one lane per (item, channel) sequence. */
size_t mst_fx_biquad_scratch_bytes(int n_items, long L, int C, int n_bands);
int mst_fx_biquad_cascade(const float *x_dev, float *y_dev, int n_items, long L, int C, const double *coef_host,
                          int n_bands, double *scratch_dev, size_t scratch_bytes, const MstFxFuse *fuse, void *stream);

/* Compressor.process / compressor_process (:529-587, :637-649), makeup gain 0.  With a scratch buffer of
 * mst_fx_compressor_scratch_bytes() (about 9 bytes per sample) the gain computer and the gain application run over all samples
 * in parallel and the attack/release smoother runs parallel in time (per-chunk convex piecewise-linear maps + one walk over the
 * chunk summaries of each sequence); scratch_dev = NULL runs the serial form, one wave per sequence. */
size_t mst_fx_compressor_scratch_bytes(int n_items, long L, int C);
int mst_fx_compressor(const float *x_dev, float *y_dev, int n_items, long L, int C, double threshold_db,
                      double attack_ms, double release_ms, double ratio, double sample_rate, double *scratch_dev,
                      size_t scratch_bytes, const MstFxFuse *fuse, void *stream);
/* MidSideImager.process (:964-1007), stereo only; scratch_dev: >= n_items * 2 * MST_SUMSQ_SLOTS doubles (partial energy sums) */
int mst_fx_midside_imager(const float *x_dev, float *y_dev, int n_items, long L, double bal, double *scratch_dev,
                          const MstFxFuse *fuse, void *stream);
/* Haas.process / haas_process (:768-786, :826-843): y = x, y[:, wet] += feedback * np.roll(x[:, wet], delay) - the roll is
 * circular, delay may be negative; wet_channel 0 = 'left', 1 = 'right'.  c_in = 1 (mono, repeated to stereo) or 2;
 * y is always [n_items, L, 2]. */
int mst_fx_haas(const float *x_dev, float *y_dev, int n_items, long L, int c_in, long delay, double feedback,
                int wet_channel, void *stream);
/* Panner.process (:927-943): x * gains with the two gains of Panner._calculate_pan_coefficents (:882-912) computed by
 * the caller (float32, like the reference's self.gains); c_in = 1 or 2, y is [n_items, L, 2]. */
int mst_fx_panner(const float *x_dev, float *y_dev, int n_items, long L, int c_in, float gain_left, float gain_right,
                  void *stream);
/* ConvolutionalReverb.process (:727-764): y = dry * x + wet * (x (*) h)[offset : offset + L] per channel, x (*) h the FULL
 * linear convolution (scipy.signal.oaconvolve(x, h, mode='full', axes=0) in the reference).  Computed as one FFT convolution of
 * n_fft = next power of two >= L + Lh_max - 1 per (item, channel): the library's own FFT kernels (csrc/fft_kernels.h) + three HIP kernels.
 * The caller resolves what the reference does on the host: IR choice / decay fade (:704-725), mono <-> stereo IR (:739-742),
 * offset = argmax_t max_c |h| + pre-delay samples, clipped to [0, Lh-1] (:757-761).  x/y: [n_items, L, C]; h: [Lh, C] device
 * float32, one impulse response for all items (apply_same_processor, :150-154).  All L + Lh - 1 samples are exact linear
 * convolution (no circular wrap).  A convolver owns its FFT plans; workspace is caller memory. */
typedef struct MstConvolver MstConvolver;
int mst_fx_convolver_create(long L, long Lh_max, int n_items, int C, MstConvolver **out);
void mst_fx_convolver_destroy(MstConvolver *cv);
size_t mst_fx_convolver_workspace_bytes(const MstConvolver *cv);
int mst_fx_convolve(MstConvolver *cv, const float *x_dev, const float *h_dev, long Lh, float *y_dev, long offset, double dry,
                    double wet, void *workspace_dev, size_t workspace_bytes, void *stream);
/* Gain.process (:1041-1051) */
int mst_fx_gain(const float *x_dev, float *y_dev, int n_items, long L, int C, double gain_db, int invert, const MstFxFuse *fuse,
                void *stream);
/* AugmentationChain.apply_processor rms_normalize branch (:143-146): y *= sqrt(mean(x^2)/max(1e-7, mean(y^2)))
 * per item; per_x / per_y = samples per item of x and of y (L * channels each: the means are scalars over each array, and a
 * processor such as Panner / Haas may turn mono into stereo); scratch_dev: >= n_items*4 doubles */
int mst_fx_rms_normalize(const float *x_dev, float *y_dev, int n_items, long per_x, long per_y, double *scratch_dev, void *stream);


/* ------------------------------------------------------------------------------------------------
 * Kernels under the input normaliser Audio_Effects_Normalizer (mixing_manipulator/data_normalization.py:76-155), which
 * inference/style_transfer.py applies to the input stems by default (data_loader.py:586-587).  The normaliser's control
 * flow (feature tables, gating, FIR design, the search loop) is host code in the package; these entry points carry its
 * sample-rate work.
 * ---------------------------------------------------------------------------------------------- */
/* The threshold x ratio search of get_comp_matching (utils_data_normalization.py:384-398): n_items candidate settings of
 * compressor_process applied to ONE input signal x_dev [L, C]; item i uses (threshold_db_dev[i], ratio_dev[i]) (device
 * float64 arrays) and writes y_dev[i] [L, C].  scratch as for mst_fx_compressor with n_items items.  peak_dev (n_items * 64
 * doubles, may be NULL): when given, a candidate whose peak reaches 1.0 is clipped to [-1, 1] like `compress` does (:352-353). */
int mst_fx_compressor_grid(const float *x_dev, float *y_dev, int n_items, long L, int C, const double *threshold_db_dev,
                           const double *ratio_dev, double attack_ms, double release_ms, double sample_rate,
                           double *scratch_dev, size_t scratch_bytes, double *peak_dev, void *stream);
/* out_dev[r] = sum of squares (mode 0) or max |x| (mode 1), float64, over x_dev[item_dev[r]][lo_dev[r] : hi_dev[r]][channel]
 * of a [n_items, L, C] batch.  Mode 0 gives the BS.1770 gating-block energies of lufs_normalize (fx_utils.py:220-238;
 * pyloudnorm Meter.integrated_loudness), mode 1 the peak of every inter-onset interval of get_mean_peak
 * (utils_data_normalization.py:316-321). */
int mst_fx_range_reduce(const float *x_dev, long L, int C, int channel, const int *item_dev, const long *lo_dev,
                        const long *hi_dev, int n_ranges, int mode, double *out_dev, void *stream);
/* Onset-detection function of aubio.onset('hfc', buf_size = hop_size = win) as driven by get_mean_peak
 * (utils_data_normalization.py:304-314), for every whole frame of `win` samples of channel `channel` of each item:
 * out_dev[item][frame] = (hfc, mean square of the frame) as float pairs; hfc = sum_k (k+1) log(|X_k| + 1) over the
 * hanningz-windowed frame's spectrum.  win in {256, 512, 1024, 2048}.  The peak picking over this short sequence is host code. */
int mst_fx_onset_hfc(const float *x_dev, int n_items, long L, int C, int channel, int win, float *out_dev, void *stream);
/* Mean STFT magnitude of get_eq_matching (utils_data_normalization.py:74-79: librosa.stft(center=False) with the given window,
 * |.|, mean over frames): mean_dev[k], k = 0 .. n_fft/2, of channel `channel` of x_dev [L, C].  Transforms in batches
 * of at most max_batch frames (csrc/fft_kernels.h: n_fft must be a power of two >= 4 - the reference's FFT_SIZE is 65536 - otherwise
 * MST_ERR_UNSUPPORTED); the analysis window is host float32 [n_fft]. */
typedef struct MstStft MstStft;
int mst_fx_stft_create(long n_fft, long hop, const float *window_host, int max_batch, MstStft **out);
void mst_fx_stft_destroy(MstStft *st);
size_t mst_fx_stft_workspace_bytes(const MstStft *st);
int mst_fx_stft_mean_magnitude(MstStft *st, const float *x_dev, long L, int C, int channel, float *mean_dev, void *workspace_dev,
                               size_t workspace_bytes, void *stream);
/* normalize_imager / process_balance (normalization_imager.py:22-99): out_dev[item] = (sum L^2, sum R^2, sum L*R) in float64 of a
 * stereo batch [n_items, L, 2] - the mid / side / left / right energies of every step of the balancing follow from these -
 * and y = M x per stereo sample, the composed re-mix (l', r') = (m00 l + m01 r, m10 l + m11 r). */
int mst_fx_stereo_moments(const float *x_dev, int n_items, long L, double *out_dev, void *stream);
int mst_fx_stereo_mix(const float *x_dev, float *y_dev, int n_items, long L, float m00, float m01, float m10, float m11, void *stream);
/* AlgorithmicReverb.process (common_audioeffects.py:1447-1495): per side (left / right, the right delays longer by
 * stereo_spread samples) the SUM of n_combs damped feedback comb filters on in_gain * x, then n_allpass all-pass sections in
 * series, then out_L = wet1 xL + wet2 xR + dry x_L, out_R = wet1 xR + wet2 xL + dry x_R.  comb_delays [n_combs] and
 * allpass_delays [n_allpass][2] (left, right) are host arrays; feedback of both filter kinds = room_size.  x_dev [n_items, L, C]
 * (C = 1 or 2) -> y_dev [n_items, L, 2].  The caller resolves the reference's quirks (only combs 5..8 reach the output, the
 * 255-sample right delay of the last all-pass).  Comb / all-pass arithmetic restated from the published Schroeder / Freeverb
 * structure (pymixconsole.components, not vendored): parity unpinned. */
size_t mst_fx_algorithmic_reverb_scratch_bytes(int n_items, long L, int n_combs);
int mst_fx_algorithmic_reverb(const float *x_dev, float *y_dev, int n_items, long L, int C, const int *comb_delays, int n_combs,
                              const int *allpass_delays, int n_allpass, int stereo_spread, double damping, double room_size,
                              double in_gain, double wet1, double wet2, double dry, double *scratch_dev, size_t scratch_bytes,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MST_HIP_H */
