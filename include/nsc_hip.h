/*
 * nsc_hip.h - C ABI of libnsc_hip.so: the MI355X (gfx950) kernels behind the NSC/CMRL hot path.
 *
 * The reference (cocosci/NSC) has no FFI: its boundary is the Python module of free functions
 * nn_core_operator.py (+ the loss functions of loss_terms_and_measures.py), every one of which
 * bottoms out in a TensorFlow op.  Each entry point below replaces the TF op(s) behind one of those
 * functions; the Python side (nsc_amd/nn_core_operator.py, nsc_amd/loss_terms_and_measures.py)
 * keeps the reference's names/arguments and binds these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; every buffer is caller-owned DEVICE memory (fp32 unless stated);
 *  - activations are time-contiguous  [B, C, T]  ("bct"); the reference's channels_last [B, T, C]
 *    is converted at the op surface with nsc_transpose_last2 (for C == 1 both layouts coincide);
 *  - conv kernels keep the TF layout  [K, Cin, Cout];
 *  - every call takes an explicit hipStream_t (as void*), allocates nothing, never synchronises,
 *    and is safe to capture into a hipGraph;
 *  - return value: 0 on success, negative nsc_status on error (never throws); nsc_last_error()
 *    returns a thread-local message for the last failure.
 *  - "accumulate" outputs (dw, db, hist, dalpha, dbins) are atomically ADDED to; the caller zeroes
 *    them (one memset of the flat gradient buffer per step).
 */
#ifndef NSC_HIP_H
#define NSC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum nsc_status {
  NSC_OK = 0,
  NSC_ERR_BAD_ARG = -1,     /* null pointer / non-positive size */
  NSC_ERR_UNSUPPORTED = -2, /* shape outside what the kernels are built for */
  NSC_ERR_LAUNCH = -3       /* HIP launch failure (message has hipGetErrorString) */
};

enum nsc_act { NSC_ACT_NONE = 0, NSC_ACT_TANH = 1, NSC_ACT_LRELU = 2 }; /* lrelu alpha = 0.2 */

int nsc_version(void);
const char* nsc_last_error(void);

/* ---- conv1d (replaces tf.compat.v1.layers.conv1d behind nn_core_operator.py:6-14 `conv1d`,
 *      :45-54 `change_channel`, and neural_speech_coding_module.py:152-156 `_down_sampling_mod`) ----
 * y[b,o,t] = epilogue( sum_{k,i} xin[b,i, t*stride + k*dil - padL] * w[k,i,o] )
 * epilogue(v): v += bias[o]; v += res (res_mode); v = act(v); v *= act'(aux) (mul_mode);
 *              store (plain | sub-pixel-shuffled | accumulate).
 * TF 'SAME' padding is expressed through padL (zero fill outside [0,Tin)); Tout is explicit.
 * in_up=1 reads a virtual zero-upsampled-by-2 input (transposed conv = dgrad of a stride-2 conv).
 */
typedef struct nsc_conv_desc {
  int B, Cin, Cout, Tin, Tout, K, dil, stride, padL;
  int act;        /* nsc_act */
  int res_mode;   /* 0 none, 1 res[B,Cout,Tout], 2 broadcast res[B,1,Tout] */
  int mul_mode;   /* 0 none, 1 v *= lrelu'(aux), 2 v *= tanh'(aux) = 1-aux^2 ; aux[B,Cout,Tout] is an activation OUTPUT */
  int out_mode;   /* 0 plain y[B,Cout,Tout]; 1 sub-pixel shuffle: y[B,Cout/2,2*Tout], y[b,o>>1,2t+(o&1)] (nsc_module:158-167);
                     with out_mode 1, res / aux are laid out like the OUTPUT */
  int in_up;      /* 0 plain; 1 virtual input u: x[u/2] if u even else 0, virtual length 2*Tin */
  int accumulate; /* 0 y = v ; 1 y += v */
} nsc_conv_desc;

/* MFMA (v_mfma_f32_16x16x4_f32) implicit-GEMM forward; Cout >= 1 (small Cout wastes tiles: use _cout1). */
int nsc_conv1d_fwd(const nsc_conv_desc* d, const float* x, const float* w, const float* bias,
                   const float* res, const float* aux, float* y, void* stream);
/* Cout == 1 special case (k55 C->1 convs, nsc_module:236,255-259): VALU dot products. Same epilogue. */
int nsc_conv1d_cout1_fwd(const nsc_conv_desc* d, const float* x, const float* w, const float* bias,
                         const float* res, const float* aux, float* y, void* stream);
/* the same conv followed by an elementwise chain on its [B,1,Tout] result v (each step was a launch of its own: nsc_cascade_step
 * behind a codec's output conv - cmrl.py:49-94: decoded (+)= dec / rs, next input = rs (x - decoded) -, nsc_axpby behind the first
 * conv's data gradient):   out2 = pa * p_in + pb * v   (p_in nullable: pb * v);   out3 = qa * q_in + qb * out2   (out3 nullable).
 * out2 may alias p_in, out3 must not alias an input.  chain nullable = nsc_conv1d_cout1_fwd. */
typedef struct nsc_cout1_chain {
  const float* p_in;
  float* out2;
  float pa, pb;
  const float* q_in;
  float* out3;
  float qa, qb;
} nsc_cout1_chain;
int nsc_conv1d_cout1_fwd_chain(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                               const float* aux, float* y, const nsc_cout1_chain* chain, void* stream);
/* the encoder's output conv (`change_channel` + tanh, neural_speech_coding_module.py:236) followed IN THE SAME LAUNCH by the soft-to-hard
 * quantizer of the training step (nn_core_operator.py:140-164) on the codes it produced: y = code [B,1,T] (d->act = tanh), qcode = the
 * quantised code, quan[b] += mean_l sum_k sqrt(p + 1e-20) (ACCUMULATED: the caller zeroes quan), hist[k] += sum p (both nullable); p is
 * not materialised.  32 bins and the k55 C -> 1 shapes only (NSC_ERR_UNSUPPORTED otherwise: launch nsc_conv1d_cout1_fwd and
 * nsc_quantize_fwd separately - same results up to summation order). */
typedef struct nsc_cout1_quant {
  const float *alpha, *bins;
  float is_quan_on;
  int soft, nb;
  float *qcode, *quan, *hist;
} nsc_cout1_quant;
int nsc_conv1d_cout1_fwd_quant(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                               const nsc_cout1_quant* quant, void* stream);
/* weight gradient: dw[k,i,o] += sum_{b,t} xin[b,i,t*stride+k*dil-padL] * dz[b,o,t];  db[o] += sum dz (db nullable).
 * flip_taps=1 writes tap k to row K-1-k (used when x/dz roles are swapped for Cout==1 convs). */
int nsc_conv1d_wgrad(const nsc_conv_desc* d, const float* x, const float* dz, float* dw, float* db,
                     int flip_taps, void* stream);
/* Same, with caller-owned scratch (floats; size from nsc_conv1d_wgrad_workspace): the partial sums of the (b,t) splits
 * are stored to private slabs and summed by a second launch instead of being added with same-address float atomics.
 * workspace NULL or too small -> the atomic path of nsc_conv1d_wgrad. */
int nsc_conv1d_wgrad_ws(const nsc_conv_desc* d, const float* x, const float* dz, float* dw, float* db,
                        int flip_taps, float* workspace, long workspace_floats, void* stream);
long nsc_conv1d_wgrad_workspace(const nsc_conv_desc* d);
/* Batched form: the weight (+bias) gradients of njobs convs in a few launches (one per kernel class + one slab
 * reduction each).  Like the block form, it exists because nothing but the optimizer reads these gradients, so a host can
 * defer them to the end of the backward pass.  Gradients are ACCUMULATED into dw / db (db nullable). */
typedef struct nsc_conv_wgrad_job {
  nsc_conv_desc d;
  const float *x, *dz;
  float *dw, *db;
  int flip_taps;
} nsc_conv_wgrad_job;
long nsc_conv1d_wgrad_batch_workspace(const nsc_conv_wgrad_job* jobs, int njobs);
int nsc_conv1d_wgrad_batch(const nsc_conv_wgrad_job* jobs, int njobs, float* workspace, long workspace_floats,
                           void* stream);
/* The stride-2 down-sampling conv of the encoder (conv1d(h, 100, 9, strides=2) + leaky-relu, neural_speech_coding_module.py:229-233)
 * and its data gradient (tf.gradients through it) with the contraction on the bf16 matrix cores: every fp32 operand split into three
 * bf16 pieces, six products, fp32 accumulation - fp32-class results (see nsc_gated_block_fwd_simg).  d = the FORWARD conv's
 * descriptor in both calls; served: Cin = Cout = 100, K 9, stride 2, dilation 1, padL 3, Tout % 64 == 0, 16-byte aligned tensors,
 * no residual / multiply / shuffle epilogue (NSC_ERR_UNSUPPORTED otherwise: nsc_conv1d_fwd serves every shape).
 *   nsc_conv1d_simage_words(which, d)              words (4 bytes) of the kernel-ready weight image; which 0 forward, 1 data gradient; 0 = not served
 *   nsc_conv1d_simage_index(which, d, w_off, idx)  nsc_gather index of the image from a buffer that holds the kernel [9][100][100] at w_off
 *   nsc_conv1d_fwd_simg:    y [B,100,Tout] = act(conv(x [B,100,Tin]) + bias)
 *   nsc_conv1d_dgrad_simg:  dx [B,100,Tin] = conv^T(dy [B,100,Tout])   (polyphase form: no zero-upsampled input) */
long nsc_conv1d_simage_words(int which, const nsc_conv_desc* d);
int nsc_conv1d_simage_index(int which, const nsc_conv_desc* d, long w_off, int* idx);
int nsc_conv1d_fwd_simg(const nsc_conv_desc* d, const float* x, const void* image, const float* bias, float* y, void* stream);
int nsc_conv1d_dgrad_simg(const nsc_conv_desc* d, const float* dy, const void* image, float* dx, void* stream);
/* ... and its WEIGHT gradient (tf.gradients w.r.t. the conv's kernel and bias) on split operands: the jobs of nsc_conv1d_wgrad_batch
 * that have this shape (x = the conv's input [B,100,Tin], dz = dy [B,100,Tout], flip_taps 0), up to 8 per call, one persistent launch
 * + one slab reduction; dw [9][100][100] / db [100] (nullable) are ACCUMULATED into.  workspace: nsc_conv1d_wgrad_split_workspace()
 * floats, caller-owned, 16-byte aligned.  Other shapes: NSC_ERR_UNSUPPORTED (nsc_conv1d_wgrad_batch serves them). */
long nsc_conv1d_wgrad_split_workspace(void);
int nsc_conv1d_wgrad_split(const nsc_conv_wgrad_job* jobs, int njobs, float* workspace, long workspace_floats, void* stream);
/* wt[K-1-k, o, i] = w[k, i, o] : weights of the data-gradient conv (dgrad == nsc_conv1d_fwd on wt). */
int nsc_weight_flip_transpose(const float* w, float* wt, int K, int Cin, int Cout, void* stream);
/* The four flipped / transposed kernels nsc_gated_block_dgrad[_cin1] reads, in ONE launch: wt = wt1 [1][narrow][Cin] | wtl [15][narrow][narrow]
 * | wtr | wt9 [k9][C][narrow], contiguous (Cin*narrow + 2*15*narrow*narrow + k9*narrow*C floats). */
int nsc_gated_block_flip_weights(const float* w1, const float* wl, const float* wr, const float* w9, float* wt, int C, int Cin,
                                 int narrow, int k9, void* stream);

/* ---- fused gated bottleneck block (replaces the whole of nn_core_operator.py:82-112 `gated_bottleneck` for
 *      Cin == wide_layer > 1, narrow_layer == 20, non_dilated_neck_kernel_size == 9, "gln" blocks) ----
 * out = [lrelu]( conv9( conv15_d(h; wl) * tanh(conv15_d(h; wr)) ) + x ),  h = lrelu(conv1(x)).  x,out [B,C,T].
 * h_out / lin_out / th_out / g_out (nullable) [B,20,T] save the intermediates for the unfused backward. */
int nsc_gated_block_fwd(const float* x, const float* w1, const float* b1, const float* wl, const float* bl,
                        const float* wr, const float* br, const float* w9, const float* b9, float* out,
                        float* h_out, float* lin_out, float* th_out, float* g_out, int B, int C, int T,
                        int narrow, int k9, int dil, int flat, void* stream);
/* The same block with ONE input channel (first block of a decoder stage, `_the_decoder_in_each_module`
 * neural_speech_coding_module.py:246-251 -> gated_bottleneck on the [B,T,1] code): x [B,1,T], w1 [1,1,20]; the residual add
 * broadcasts x over the C output channels.  C in {100, 50, 25}, dil in {1, 2}; other shapes: NSC_ERR_UNSUPPORTED (per-conv path). */
int nsc_gated_block_fwd_cin1(const float* x, const float* w1, const float* b1, const float* wl, const float* bl,
                        const float* wr, const float* br, const float* w9, const float* b9, float* out,
                        float* h_out, float* lin_out, float* th_out, float* g_out, int B, int C, int T,
                        int narrow, int k9, int dil, int flat, void* stream);

/* Fused DATA-PATH backward of the same block (8 waves, two workgroups per CU): from the saved h, lin, th (tanh branch)
 * [B,20,T], x and dy [B,C,T] it writes dx = (conv1^T(dz1) + dy) * act'(x), da [B,40,T] (= dlin | dgate) and
 * dz1 [B,20,T] - exactly the inputs of nsc_gated_block_wgrad.  dil in {1,2}, narrow 20, k9 9, C <= 112. */
int nsc_gated_block_dgrad(const float* x, const float* h, const float* lin, const float* th, const float* dy,
                          const float* wt1, const float* wtl, const float* wtr, const float* wt9, float* dx, float* da,
                          float* dz1, int B, int C, int T, int narrow, int k9, int dil, int in_act, void* stream);
/* The same for the block with ONE input channel (see nsc_gated_block_fwd_cin1): dx [B,1,T] = sum_c w1[c] dz1[c] + sum_o dy[o]
 * (the forward broadcast x over the C output channels; the producer of x is the quantizer, so no activation gradient);
 * wt1 [20]; da_rows = channel rows per frame of the tensor(s) dlin / dgate point into: 20 = two separate [B,20,T] tensors,
 * 40 = the two halves of one [B,40,T] tensor (dgate = dlin + 20 T; the form nsc_gated_block_wgrad_batch reads).
 * C in {100, 50, 25}, dil in {1, 2}. */
int nsc_gated_block_dgrad_cin1(const float* h, const float* lin, const float* th, const float* dy,
                               const float* wt1, const float* wtl, const float* wtr, const float* wt9, float* dx,
                               float* dlin, float* dgate, float* dz1, int B, int C, int T, int narrow, int k9,
                               int dil, int da_rows, void* stream);

/* Kernel-ready parameter IMAGES of a gated block: the fast prologue of the two persistent kernels above (no counterpart in
 * the reference: TensorFlow keeps its own packed filter copies).  A caller that runs the same block many times keeps one
 * image per block and direction on the device and rebuilds it with ONE nsc_gather launch whenever the parameters change:
 *   nsc_gated_block_image_floats(which, C, Cin, dil)   size of the image in floats (0: no image kernel for this shape)
 *   nsc_gated_block_image_index(which, ..., offs, idx) host index map for nsc_gather: image[i] = src[idx[i]] (-1: pad).
 *       which = 0 (forward):       offs = offsets of w1, b1, wl, bl, wr, br, w9, b9 in the gathered buffer
 *       which = 1 (data gradient): offs = offsets of wt1, wtl, wtr, wt9 (the flipped / transposed kernels)
 *   nsc_gated_block_fwd_img / _dgrad_img: nsc_gated_block_fwd[_cin1] / nsc_gated_block_dgrad[_cin1] on an image (16-byte aligned);
 *       Cin = C or 1; same outputs bit for bit.  C in {100, 50, 25}, dil in {1, 2}. */
long nsc_gated_block_image_floats(int which, int C, int Cin, int dil);
int nsc_gated_block_image_index(int which, int C, int Cin, int dil, const long* offs, int* idx);
int nsc_gated_block_fwd_img(const float* img, const float* x, float* out, float* h_out, float* lin_out, float* th_out,
                            float* g_out, int B, int C, int Cin, int T, int dil, int flat, void* stream);
int nsc_gated_block_dgrad_img(const float* img, const float* x, const float* h, const float* lin, const float* th,
                              const float* dy, float* dx, float* dlin, float* dgate, float* dz1, int B, int C, int Cin,
                              int T, int dil, int in_act, int da_rows, void* stream);

/* ---- PAIR launches: the two blocks of a stack (`_stack_bottleneck_blocks`, neural_speech_coding_module.py:183-217: dilation 1 then 2,
 *      both C -> C) in ONE launch.  The second block's edge tiles read what neighbouring workgroups wrote in the first block: every
 *      workgroup publishes a flag when its first-block tiles are in memory (agent-scope release) and waits for its two neighbours'
 *      flags before the second block's first load (acquire) - no kernel boundary, no second dispatch, the second block's weight
 *      prologue runs while the neighbours finish.  Requires every workgroup of the launch to be resident (grid = min(tiles, 256) <=
 *      CUs: checked, NSC_ERR_UNSUPPORTED otherwise - launch the blocks one by one then).
 *      flags: nsc_gated_block_pair_flag_ints() ints (one per workgroup), ZEROED by the caller before each launch.
 *      timeouts: ONE caller-owned int the library only ever ADDS to: neighbour waits that gave up (must stay 0; the results of that
 *      launch are undefined otherwise).  It lives outside whatever the caller zeroes per launch / per step, so a count survives.
 *   fwd:   x -> block 0 (dil 1, lrelu out) -> out0 -> block 1 (dil 2, flat1) -> out1; saved activations h / lin / th / g of each
 *          block nullable (together).  Cin0 = C, or 1: block 0 is the first block of a decoder stage (x [B,1,T], image of Cin = 1).
 *   dgrad: block 1 first (dy1 -> dx1, da1 [B,40,T], dz1_1), then block 0 on dy = dx1 (-> dx0, da0, dz1_0); in_act0 = activation
 *          that produced block 0's input (Cin0 = 1: none; x0 unused, dx0 [B,1,T], da0 = dlin | dgate as one [B,40,T] tensor). */
int nsc_gated_block_pair_flag_ints(void);
int nsc_gated_block_pair_fwd_img(const float* img0, const float* img1, const float* x, float* out0, float* h0, float* lin0,
                                 float* th0, float* g0, float* out1, float* h1, float* lin1, float* th1, float* g1, int B, int C,
                                 int Cin0, int T, int flat1, int* flags, int* timeouts, void* stream);
int nsc_gated_block_pair_dgrad_img(const float* img1, const float* x1, const float* h1, const float* lin1, const float* th1,
                                   const float* dy1, float* dx1, float* da1, float* dz1_1, const float* img0, const float* x0,
                                   const float* h0, const float* lin0, const float* th0, float* dx0, float* da0, float* dz1_0,
                                   int B, int C, int Cin0, int T, int in_act0, int* flags, int* timeouts, void* stream);

/* ---- The same block kernels on the bf16 MATRIX CORES with SPLIT operands (csrc/block_split.hip; replaces the same reference
 *      function, nn_core_operator.py:82-112).  Every fp32 operand of the two long contractions (both k15 gate convs, the k9 conv:
 *      94 % of the block's MACs) is split exactly into three bf16 pieces (hi + lo + lo2 = x), six of the nine piece products are
 *      formed by v_mfma_f32_16x16x32_bf16 and accumulated in fp32: the error class of the fp32 matrix instruction (dropped terms
 *      <= 2^-24 |a||b|), at 2.7x its rate, with the vector ALU free for the elementwise phases.  The 1x1 conv stays on the exact
 *      fp32 instruction.  Results agree with nsc_gated_block_fwd_img to fp32 rounding (not bit for bit).
 *   nsc_gated_block_simage_words(which, C, Cin, dil)   size of the split image in 32-bit words (0: no split kernel for this shape;
 *                                                      which = 0: forward, 2: the three-launch data gradient below)
 *   nsc_gated_block_simage_index(which, ..., offs, idx) index map for nsc_gather / nsc_step_begin, offs as for
 *      nsc_gated_block_image_index.  Entries carry a MODE in bits 26..29 (so source offsets must stay below 2^26 - 2^20):
 *      0 = the fp32 value; m > 0 = two bf16 pieces packed in one word, low half from src[i], high half from src[i + stride],
 *      plane (m - 1) / 5 (0 hi, 1 lo, 2 lo2), stride {20, 25, 50, 100, 1}[(m - 1) % 5].  nsc_gather understands them.
 *   nsc_gated_block_fwd_simg / nsc_gated_block_pair_fwd_simg: arguments of nsc_gated_block_fwd_img / _pair_fwd_img; T % 4 == 0 and
 *      16-byte aligned tensors (NSC_ERR_UNSUPPORTED otherwise: use the exact kernels). */
long nsc_gated_block_simage_words(int which, int C, int Cin, int dil);
int nsc_gated_block_simage_index(int which, int C, int Cin, int dil, const long* offs, int* idx);
int nsc_gated_block_fwd_simg(const float* img, const float* x, float* out, float* h_out, float* lin_out, float* th_out,
                             float* g_out, int B, int C, int Cin, int T, int dil, int flat, void* stream);
/* The same data-path backward as THREE launches on the which = 2 image (csrc/block_bwd_split.hip): the two long contractions as
 * polyphase GEMMs with 80 = 4 x 20 rows (no padded row tiles, no halo recompute, no partial sums across waves), then the HBM-bound
 * 1x1 gradient + residual.  which = 2 image: the three bf16 pieces of W9 [9][20][C -> multiple of 8] and of Wl | Wr [15][20][40] in
 * the kernels' own layout (offs as for which = 0; C in {100, 50}).  w1: the 1x1 kernel in the PARAMETER layout [Cin][20];
 * da = the joint [B,40,T] tensor (dlin | dgate); dx nullable; Cin = C or 1 (in_act none); T % 4 == 0, 16-byte aligned tensors.
 * Replaces (on the bf16 matrix cores) the tf.gradients of nn_core_operator.py:82-112 with respect to the block's input. */
int nsc_gated_block_dgrad_simg2(const void* img, const float* w1, const float* x, const float* h, const float* lin, const float* th,
                                const float* dy, float* dx, float* da, float* dz1, int B, int C, int Cin, int T, int dil, int in_act,
                                void* stream);
int nsc_gated_block_pair_fwd_simg(const float* img0, const float* img1, const float* x, float* out0, float* h0, float* lin0,
                                  float* th0, float* g0, float* out1, float* h1, float* lin1, float* th1, float* g1, int B, int C,
                                  int Cin0, int T, int flat1, int* flags, int* timeouts, void* stream);

/* Persistent weight-gradient kernel of one gated block: all eight parameter gradients (accumulated) from the saved
 * activations x [B,C,T], h, g [B,20,T] and the data-path gradients dy [B,C,T], da [B,40,T] (= dlin | dgate, the
 * nsc_glu_bwd_cat output), dz1 [B,20,T] (= dL/d(pre-activation of h)).  Optionally (dx != NULL) it also produces the
 * block's 1x1 data gradient dx = (conv1^T(dz1) + dy) * act'(x) from the tiles it has staged anyway (wt1 = the
 * flipped/transposed 1x1 kernel [20][C]).  workspace (nullable): nsc_gated_block_wgrad_workspace(C) floats; enables the
 * store+reduce flush when the eight gradients are one contiguous range in creation order. */
int nsc_gated_block_wgrad(const float* x, const float* h, const float* g, const float* dy, const float* da,
                          const float* dz1, float* dw1, float* db1, float* dwl, float* dbl, float* dwr, float* dbr,
                          float* dw9, float* db9, const float* wt1, float* dx, int in_act, int B, int C, int T,
                          int narrow, int k9, int dil, int waves /*8: fastest alone; 4: leaves half the CU's registers and
                          LDS to kernels running concurrently on another stream*/,
                          int part /*0 all gradients; 1 dW9/db9 only; 2 dWl/dWr/dW1 + biases only (light launches that
                          share a CU with other kernels)*/, float* workspace, void* stream);
long nsc_gated_block_wgrad_workspace(int C);
/* Batched form: the parameter gradients of njobs gated blocks in ONE persistent launch (+ one slab reduction).  The
 * reference's tf.gradients has no ordering constraint between a block's weight gradients and the rest of the backward
 * pass (cmrl.py:106-113 / :369-372 only hand them to the optimizer), so a host can defer all of them to the end.
 * Each job's eight gradients must be contiguous in creation order starting at `grads`
 * (dw1 | db1 | dwl | dbl | dwr | dbr | dw9 | db9); they are ACCUMULATED into.  workspace: >=
 * nsc_gated_block_wgrad_batch_workspace(max C) floats, caller-owned; with twice that, the launches of the two block widths
 * (C <= 64 | C > 64) keep their slabs side by side and ONE reduce launch sums both. */
typedef struct nsc_block_wgrad_job {
  const float *x, *h, *g, *dy, *da, *dz1;
  float* grads;
  int C, T, dil;
  int Cin;   /* channels of x: 0 or C for a C -> C block, 1 for the one-input-channel decoder block (grads then starts
              * with dW1 [1,20]) */
} nsc_block_wgrad_job;
long nsc_gated_block_wgrad_batch_workspace(int Cmax);
int nsc_gated_block_wgrad_batch(const nsc_block_wgrad_job* jobs, int njobs, int B, int narrow, int k9,
                                float* workspace, long workspace_floats, void* stream);
/* nsc_gated_block_wgrad_batch on the bf16 matrix cores with split operands (see nsc_gated_block_fwd_simg): same arguments, same
 * results to fp32 rounding; jobs it does not serve (T % 4 != 0, tensors not 16-byte aligned) run the exact kernel. */
int nsc_gated_block_wgrad_batch_split(const nsc_block_wgrad_job* jobs, int njobs, int B, int narrow, int k9,
                                float* workspace, long workspace_floats, void* stream);

/* ---- separable conv pieces (replaces tf.keras.layers.SeparableConv1D behind nn_core_operator.py:17-21) ---- */
int nsc_depthwise_fwd(const float* x, const float* wd /*[K,C]*/, float* y, int B, int C, int T, int K, void* stream);
int nsc_depthwise_bwd(const float* x, const float* wd, const float* dy, float* dx, float* dwd /*accumulate*/,
                      int B, int C, int T, int K, void* stream);

/* ---- the decoder's whole up-sampling stage in one kernel per direction (replaces conv1d_depth = SeparableConv1D,
 *      nn_core_operator.py:17-21, + activation + sub-pixel shuffle, neural_speech_coding_module.py:168-181) ----
 * fwd: y [B,C/2,2T] = shuffle(act(wp . depthwise9(x) + bias)); x [B,C,T], wd [9,C] depthwise taps, wp [C,C] pointwise
 *      kernel ([1,Cin,Cout] as stored), dwo (nullable) [B,C,T] receives the depthwise output the backward pass needs.
 * bwd: from dz [B,C/2,2T] = dL/d(pre-activation of y): dzp [B,C,T] (dz un-shuffled: input of the pointwise weight
 *      gradient), ddw [B,C,T] (gradient at the depthwise output: input of the depthwise weight gradient) and
 *      dx [B,C,T].  K = 9, C in {100, 50}; other shapes: compose nsc_depthwise_* / nsc_conv1d_* / nsc_unshuffle2. */
int nsc_upsample_fwd(const float* x, const float* wd, const float* wp, const float* bias, float* dwo, float* y,
                     int B, int C, int T, int K, int act, void* stream);
int nsc_upsample_bwd(const float* dz, const float* wd, const float* wp, float* dzp, float* ddw, float* dx,
                     int B, int C, int T, int K, void* stream);

/* ---- gate (tf.multiply(left, tanh-right), nn_core_operator.py:100) on a fused [B,2n,T] pre-activation ----
 * fwd: a[:, n:] <- tanh(a[:, n:]) in place; g = a[:, :n] * a[:, n:].   bwd: da from dg and the saved a. */
int nsc_gate_fwd(float* a, float* g, int B, int n, int T, void* stream);
int nsc_gate_bwd(const float* a, const float* dg, float* da, int B, int n, int T, void* stream);

/* separate-branch form used by the unfused path: g = lin * th (th = tanh branch output), and its backward
 * dlin = dg*th, dgate_pre = dg*lin*(1-th^2). */
int nsc_mul(const float* a, const float* b, float* out, long n, void* stream);
int nsc_glu_bwd(const float* lin, const float* th, const float* dg, float* dlin, float* dgate, long n, void* stream);
int nsc_glu_bwd_cat(const float* lin, const float* th, const float* dg, float* da /*[B,2n,T] = dlin | dgate*/, int B, int n, int T, void* stream);

/* stand-alone activation (nn_core_operator.py:24-31 `activation_func` = leaky_relu 0.2; tanh for conv epilogues) */
int nsc_act_fwd(const float* x, float* y, long n, int act, void* stream);
int nsc_act_bwd(const float* dy, const float* y, float* dx, long n, int act, void* stream); /* dx = dy * act'(.) from the OUTPUT y */
/* quan_loss / entropy histogram of a materialised soft assignment p[B,L,nb] (op-surface form of
 * loss_terms_and_measures.py:257-267) and the matching elementwise backward dp = gq[b]/L*0.5/sqrt(p+1e-20) + gh[k]. */
int nsc_p_stats(const float* p, int B, int L, int nb, float* quan /*[B] nullable*/, float* hist /*[nb] accumulate, nullable*/, void* stream);
int nsc_p_stats_bwd(const float* p, const float* gq, const float* gh, float* dp, int B, int L, int nb, void* stream);

/* ---- small glue ---- */
int nsc_gather(const float* src, const int* idx, float* dst, long n, void* stream); /* dst[e] = src[idx[e]] */
int nsc_axpby(const float* x, const float* y, float* out, float a, float b, long n, void* stream); /* out = a*x + b*y (y nullable) */
int nsc_channel_sum(const float* x, float* out, int B, int C, int T, int accumulate, void* stream); /* out[b,0,t] (+)= sum_c x[b,c,t] */
int nsc_unshuffle2(const float* ys /*[B,C/2,2T]*/, float* y /*[B,C,T]*/, int B, int C, int T, void* stream);
/* the sub-pixel shuffle itself (neural_speech_coding_module.py:158-167) on [B,C,T] tensors: ys[b, c>>1, 2t + (c&1)] = y[b,c,t] */
int nsc_shuffle2(const float* y /*[B,C,T]*/, float* ys /*[B,C/2,2T]*/, int B, int C, int T, void* stream);
int nsc_transpose_last2(const float* x /*[B,R,Cc]*/, float* y /*[B,Cc,R]*/, int B, int R, int Cc, void* stream);
/* cascade step between two codecs (cmrl.py:49-94): decoded = sc*dec (+ decoded if accumulate); xin (nullable: last codec)
 * = rs*x - rs*decoded, i.e. the next codec's input res_scalar * (x - sum of the outputs so far) */
int nsc_cascade_step(const float* dec, float* decoded, int accumulate, const float* x, float* xin, float sc, float rs,
                     long n, void* stream);
int nsc_sum_all(const float* x, float* out /*accumulate [1]*/, long n, void* stream);
/* up to NSC_SUM_MAXJ such sums in one launch (the bias gradients of the Cout = 1 convs of a step) */
#define NSC_SUM_MAXJ 8
typedef struct nsc_sum_job {
  const float* x;
  float* out;
  long n;
} nsc_sum_job;
int nsc_sum_all_batch(const nsc_sum_job* jobs, int njobs, void* stream);

/* ---- soft-to-hard scalar quantizer (replaces nn_core_operator.py:140-164 `scalar_softmax_quantization`,
 *      fused with loss_terms_and_measures.py:257-259 `quan_loss` and :262-267 `entropy_coding_loss` partials) ----
 * code[B*L]; p = softmax_k(alpha*|c-bins_k|); q = sum_k sel_k*bins_k (sel = p if soft else one_hot(argmax p,
 * lowest index on ties)); out = (1-is_quan_on)*c + is_quan_on*q.
 * p_out (nullable) [B*L,nb];  quan_out (nullable) [B] = mean_l sum_k sqrt(p+1e-20);
 * hist (nullable, accumulate) [nb] += sum_{b,l} p. */
int nsc_quantize_fwd(const float* code, const float* alpha, const float* bins, float is_quan_on, int soft,
                     int B, int L, int nb, float* p_out, float* out, float* quan_out, float* hist, void* stream);
/* entropy from a (possibly all-reduced) histogram: ent[0] = -sum h log2(h+1e-7), h = hist/sum(hist);
 * ghist[k] = d ent / d hist[k]. */
int nsc_entropy_from_hist(const float* hist, int nb, float* ent, float* ghist, void* stream);
/* the same for up to NSC_ENT_MAXJ histograms in one launch (one per codec + the LSF quantizer's) */
#define NSC_ENT_MAXJ 8
typedef struct nsc_entropy_job {
  const float* hist;
  float *ent, *ghist;   /* both nullable */
  int nb;
} nsc_entropy_job;
int nsc_entropy_from_hist_batch(const nsc_entropy_job* jobs, int njobs, void* stream);
/* entropy_coding_loss evaluated one frame at a time (the reference's validation loop feeds batches of 1,
 * neural_speech_coding_module.py:685-722): ent[b] = entropy in bits of frame b's own soft histogram, p [B, L, nb]. */
int nsc_frame_entropy(const float* p, int B, int L, int nb, float* ent, void* stream);
/* backward.  Upstream gradients: dout[B*L] (of `out`), dp (nullable, explicit dL/dp [B*L,nb]),
 * plus the analytically fused loss terms  c_quan*quan_loss(p)[b] summed over b  and  ent_scale*entropy(p):
 *   dL/dp += c_quan/L * 0.5/sqrt(p+1e-20) + ent_scale*ghist[k].
 * pre_tanh=1 additionally multiplies dcode by (1-code^2) (code = tanh(z): returns dL/dz).
 * dalpha[1], dbins[nb] are accumulated (nullable = not trainable). */
int nsc_quantize_bwd(const float* code, const float* alpha, const float* bins, float is_quan_on, int soft,
                     int B, int L, int nb, const float* dout, const float* dp, float c_quan,
                     const float* ghist, float ent_scale, int pre_tanh,
                     float* dcode, float* dalpha, float* dbins, void* stream);

/* ---- reconstruction losses (replaces loss_terms_and_measures.py:77-79 `mse_loss`, :178-183 `tf_stft`,
 *      :130-148 `mfcc_transform`, :151-175 `mfcc_loss`): one workgroup per 512-sample frame, radix-2 FFT in LDS ----
 * time_out[b] = sqrt(mean_t (d-o)^2 + 1e-7);  freq_out[b] = mean over 4 mel banks of sqrt(mean_m (dlogmel)^2 + 1e-7).
 * grad (nullable) [B,512] = d( sum_b gt[b]*time[b] + gf[b]*freq[b] ) / d decoded, where gt/gf are per-frame
 * upstream weights (nullable -> the scalars ct / cf).  mel [257,184] row-major, melT [184,257]. */
int nsc_recon_loss(const float* decoded, const float* target, int B, float ct, float cf,
                   const float* gt, const float* gf, const float* mel, const float* melT,
                   float* time_out, float* freq_out, float* grad, void* stream);
/* The same with the band structure of the mel matrix handed in (the four banks are triangular filters: a column is
 * non-zero on one run of bins, a bin belongs to at most a few neighbouring filters of each bank), so only those terms
 * are visited - identical results, the matrix products shrink ~20x.  ranges: int32 [184][2] = [k_lo, k_hi) of every
 * mel column, followed by [257][4][2] = [j_lo, j_hi) of every bin within each bank (columns 0:8, 8:24, 24:56, 56:184). */
int nsc_recon_loss_banded(const float* decoded, const float* target, int B, float ct, float cf,
                          const float* gt, const float* gf, const float* mel, const float* melT, const int* ranges,
                          float* time_out, float* freq_out, float* grad, void* stream);
/* Backward of the pair (mse_loss, mfcc_loss) from the forward launch's leftovers (op surface): gfreq [B,512] = the `grad` of an
 * nsc_recon_loss call with ct = 0, cf = 1 (d mfcc_loss[b] / d decoded[b,:]), time [B] its time_out;
 * grad[b,:] = gf[b] gfreq[b,:] + gt[b] (decoded - target)[b,:] / (512 time[b]); gt / gf nullable (that loss is unused).
 * Replaces the second traversal tf.gradients makes of loss_terms_and_measures.py:77-79, 151-175. */
int nsc_recon_loss_combine(const float* decoded, const float* target, const float* time, const float* gt, const float* gf,
                           const float* gfreq, int B, float* grad, void* stream);
/* bare rFFT-512 magnitude (tf_stft): re/im/mag [B,257] (any nullable). */
int nsc_rfft512(const float* sig, int B, float* re, float* im, float* mag, void* stream);

/* ---- optimizer (replaces tf.compat.v1.train.AdamOptimizer, nsc_module:922-925): TF1 Adam on flat buffers ----
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).  t is read from t_dev[0] if non-null. */
int nsc_adam_tf1_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                      float eps, int t, const int* t_dev, void* stream);
int nsc_increment(int* counter, void* stream);
/* measurement aid (bench.py): 256 one-wave workgroups run a dependent chain of `iters` FMAs; a launch whose duration no cache can
 * change, bracketed once and twice by events to find what a bracket adds to the launch inside it. */
int nsc_spin(float* sink, int iters, void* stream);
/* opening of a training step in one launch (the engine's sess.run(trainop) prologue, neural_speech_coding_module.py:455-458 has
 * no counterpart - TF zeroes nothing and keeps no flipped kernels): dst[e] = idx[e] >= 0 ? src[idx[e]] : 0 for e < n (nsc_gather:
 * the data-gradient kernels and parameter images); zero[0, zero_n) = 0 (gradients + histograms; 16-byte aligned, zero_n % 4 == 0);
 * counter[0] += 1 when non-null (the Adam step counter nsc_adam_tf1_step reads at the end of the step). */
int nsc_step_begin(const float* src, const int* idx, float* dst, long n, float* zero, long zero_n, int* counter, void* stream);
/* ... gathering only the chunks[c] = {first word, words (<= 1024)} of dst / idx that the step's kernels read (int pairs, device memory) */
int nsc_step_begin_chunks(const float* src, const int* idx, float* dst, const int* chunks, int nchunks, float* zero, long zero_n,
                          int* counter, void* stream);

/* ---- LPC front / back end of the collaborative-quantisation path (replaces the tf.py_func bodies of
 *      lpc_utilities.py: `lsf2poly_after_quan` :28-33, `lpc_analysis_get_residual` :37-77, `lpc_synthesizer_tr` :137-156;
 *      call sites nsc_module:1012-1013, 1029, 1100-1101, cmrl.py:161-162, 239, 414-415, 451).  float32 in / out like the
 *      reference's arrays, double-precision accumulation like its Python floats. ----
 * lsf [B,order] (radians, ascending) -> poly [B,order+1] (a[0] = 1): spectrum.lsf2poly per frame. */
int nsc_lsf2poly(const float* lsf, float* poly, int B, int order, void* stream);
/* x [B,512], poly [B,order+1] -> res [B,512]: seven 128-sample sub-frames at hop 64, each filtered from rest by A(z)
 * and cross-faded with hanning(128) (first sub-frame: flat then falling half, last: rising half then flat). */
int nsc_lpc_residual(const float* x, const float* poly, float* res, int B, int order, void* stream);
/* out = res / A(z) per frame, from rest (order 16). */
int nsc_lpc_synthesis(const float* poly, const float* res, float* out, int B, int order, void* stream);
/* p[0..n) = 0 on the stream (a memset node under hipGraph capture). */
int nsc_zero(float* p, long n, void* stream);
/* *id = identity of the hipGraph capture `stream` is recording into, 0 when it is not capturing (host bookkeeping of the op surface:
 * buffers prepared outside a capture must be prepared again inside it; no counterpart in the reference - TF1 sessions have no capture). */
int nsc_stream_capture_id(void* stream, unsigned long long* id);

/* ---- framing (utilities.py:7-39): frames[i,:] = utt[480 i : 480 i + 512] * window; overlap-add back ---- */
int nsc_frame_utterance(const float* utt, long n, const float* window /*nullable [512]*/, float* frames, int nframes, void* stream);
int nsc_overlap_add(const float* frames, int nframes, const float* win3 /*[3,512]: first, middle, last Hann variants*/,
                    float* out /*[480(n-1)+512]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSC_HIP_H */
