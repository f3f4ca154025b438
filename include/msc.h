/* msc.h -- C ABI of libmsc_hip.so: the MI355X (gfx950) kernels behind the U-Net segmentation hot
 * path of neptune-ai/open-solution-mapping-challenge.
 *
 * The reference is pure Python and has no FFI of its own; every entry point below replaces a
 * call the reference makes into a third-party native library (cuDNN/MKL-DNN through torch,
 * scipy.ndimage, scikit-image, pydensecrf).  The reference call site each one stands in for is
 * cited as file:line under /root/reference.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; msc_last_error() gives a thread-local text
 *   - all tensors are raw DEVICE pointers owned by the caller (torch allocates them); no hidden
 *     global state; `stream` is a hipStream_t passed as void* (0 = default stream)
 *   - activations are NHWC: element (n,y,x,c) at ((n*H+y)*W+x)*ld + c, where `ld` (elements) >= C
 *     lets a tensor be a channel slice of a wider buffer (skip-concats are never materialised)
 *   - dtype: MSC_DTYPE_F32 (exact-fp32 parity mode, v_mfma_f32_16x16x4_f32), MSC_DTYPE_BF16 (throughput mode,
 *     v_mfma_f32_16x16x32_bf16) or MSC_DTYPE_F16 (v_mfma_f32_16x16x32_f16; BASELINE.json configs[4]); accumulation,
 *     BatchNorm statistics, losses and master weights are fp32 in every mode; 16-bit values are raw uint16 bit patterns
 *   - re-entrant per stream; one host thread per GPU/process, no internal threads
 */
#ifndef MSC_H
#define MSC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSC_DTYPE_F32 0
#define MSC_DTYPE_BF16 1
#define MSC_DTYPE_F16 2

const char* msc_last_error(void);
int msc_abi_version(void);

/* ---------------------------------------------------------------- convolutions (network) ------
 * nn.Conv2d / nn.ConvTranspose2d forward and data-gradient:
 *   conv3x3+bias+ReLU  src/unet_models.py:21-34      ConvTranspose2d(k4,s2,p1)+ReLU  :138-140
 *   final 1x1          src/unet_models.py:383,403    ResNet convs (torchvision)      :345-371
 * mode 0 (gather):      out[q] = sum_t in[q*stride + off(t)] * W[.][t][.]   off = t-pad, or pad-t if flip
 * mode 1 (transposed):  out[q] = sum_{t:(q+pad-t) even} in[(q+pad-t)/2] * W[.][t][.]   (stride must be 2)
 * epilogue: v = acc*scale[c] + shift[c] (+ res) ; ReLU ; store.  scale/shift/res may be NULL.
 * stats (mode 0 only, may be NULL): per-channel sum / sum of squares of the raw accumulators, ADDED (fp32 atomics) into
 *   [MSC_BN_SLOTS][Cout][2] doubles -- one slot per XCD, so that every address is only ever touched from one L2; the
 *   caller zeroes the slots (msc_memset_zero), msc_bn_apply sums them in its prologue (BatchNorm2d training mode).
 *   stats_kind 1 (data-gradient convs): the launch also reduces what msc_bn_bwd_reduce would read back -- the output
 *   is the gradient w.r.t. a BatchNorm+ReLU layer's activation, stats_y that layer's pre-BN tensor (same shape as
 *   out), scale/shift its forward coefficients (used for the ReLU mask only, NULL = no ReLU; no affine is applied):
 *   stats[slot][c] += (sum dh, sum dh*y), dh = acc*[scale*y+shift > 0].  Same layout, consumed by msc_bn_bwd_apply.
 *   ABI v6: with stats_z set the mask is [stats_z > 0] instead (the layer's OUTPUT: a residual block's ReLU sits after the add, so the
 *   pre-BN tensor alone does not give it) and a residual is allowed: the conv that ACCUMULATES the last addend of a residual join's
 *   gradient (out = acc + res) reduces dh = out*[stats_z > 0] -- the msc_bn_bwd_reduce launch of the join's BatchNorm is not needed.
 *   stats_kind 2 (data-gradient convs; no scale/shift/res/relu): the output is the gradient w.r.t. the activation of a
 *   bias+ReLU layer (ConvRelu, ConvTranspose2d+ReLU: src/unet_models.py:25-34,138-140), stats_y that activation: the
 *   launch STORES out = acc*[stats_y > 0] (the ReLU backward) and adds stats[slot][c][0] += sum_p out[p][c] (the layer's
 *   bias gradient, folded into db by msc_bias_slots_finalize) -- what a separate msc_relu_bias_grad pass would do.
 * weights: dtype [Cout][KH][KW][Cin].  Cin*sizeof(dtype) % 64 == 0, Cout % 32 == 0. */
typedef struct msc_conv_desc {
    const void* in;
    const void* wt;
    void* out;
    const void* res;
    const float* scale;
    const float* shift;
    double* stats;
    int64_t in_ld, out_ld, res_ld;
    int32_t dtype, mode;
    int32_t N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad, flip, relu;
    int32_t cfg;   /* 0 = heuristic kernel configuration, 1..msc_conv_num_cfgs() = explicit (tile, K-step, ring depth) */
    int32_t stats_kind;
    const void* stats_y;
    int64_t stats_y_ld;
    /* ABI v5, eval mode: the network's last two layers in one launch.  When final_w is set (only with the 32-channel 3x3 halo kernel,
     * cfg 27: Cin = Cout = 32, 3x3, stride 1) the epilogue also applies the final 1x1 convolution 32 -> 2 + bias to the stored
     * (16-bit rounded) ReLU output and writes logits / softmax probabilities as f32 NCHW planes [N][2][Ho][Wo] -- what msc_final_fwd
     * would read the tensor back for (Conv2d(32, 2, 1) + the host softmax, src/unet_models.py:383,403, src/models.py:88-92);
     * final_logits or final_probs may be NULL.  final_skip_store: do not write `out` (nothing else reads dec0's output in eval). */
    const float* final_w;
    const float* final_b;
    float* final_logits;
    float* final_probs;
    int32_t final_skip_store;
    /* ABI v5, split-K (mode 0, no statistics): splitk > 1 runs the reduction (filter taps x channel chunks) as `splitk` slices, each
     * block storing its fp32 partial tile into its plane of splitk_ws (f32 [splitk][pixels][Cout], scratch); a finishing pass adds
     * the planes in slice order, applies the epilogue and stores `out`.  For layers with few output tiles and a long reduction -- the decoder's
     * centre / dec5 ConvRelu on 4x4 and 8x8 maps (src/unet_models.py:373-374): 512 pixels x 512 channels x 18432 is 32 tiles. */
    int32_t splitk;
    float* splitk_ws;
    /* ABI v6: stats_kind 1 with the ReLU mask taken from a stored activation (see above); same shape as out, stats_z_ld elements per pixel */
    const void* stats_z;
    int64_t stats_z_ld;
    /* ABI v7: 1 = stats_z is the ReLU byte mask msc_bn_apply wrote (relu_mask: bit e of byte k = channel (16 / sizeof(dtype)) * k + e is
     * positive), stats_z_ld BYTES per pixel -- the join's gradient is masked without re-reading the 16-bit activation */
    int32_t stats_z_bits;
    int32_t reserved0;
    /* ABI v9: `in` is the RAW output y of a training-mode BatchNorm'd conv (see msc_bn_input below); NULL: `in` is used as it is */
    const struct msc_bn_input* in_bn;
} msc_conv_desc;
/* BatchNorm + ReLU applied to a convolution's input ON LOAD (ABI v9; training; torchvision Bottleneck: bn2 + relu in front of conv3,
 * src/unet_models.py:345-351,365-371): instead of a msc_bn_apply launch that reads y and writes relu(bn(y)) for the next conv to read,
 * that conv fetches y, finalises the layer's coefficients from its statistics slots (what msc_bn_apply's prologue does, same arithmetic)
 * and rewrites every landed operand stage in LDS.  The fields are msc_bn_apply's; `out` (may be NULL) receives the activation relu(scale*y +
 * shift), stored once per pixel by the blocks of the first channel tile -- the weight gradient of the consuming conv reads it.
 * Supported: mode 0, stride 1, 1x1 (cfg 33 / 1, or 0 = the first of the two that takes the layer) or 3x3 with pad 1 (the halo-tile kernel: cfg 42 / 51 / 53, or 0 = the first of them that takes the layer; the zero padding is of the
 * activation, not of y), 16-bit dtype, Cin % 64 == 0, Cin <= 512, one image range per launch; msc_conv_cfg_ok tells. */
typedef struct msc_bn_input {
    const double* slots;      /* [MSC_BN_SLOTS][Cin][2] partial (sum, sum of squares) of y, as msc_conv_igemm's `stats` leaves them */
    int64_t count;            /* pixels the statistics are over */
    const float* gamma; const float* beta;
    float eps, momentum;
    float* running_mean; float* running_var;            /* updated (unbiased variance), may be NULL */
    float* scale; float* shift;                         /* f32[Cin] out: the coefficients, for the backward */
    float* save_mean; float* save_invstd;               /* f32[Cin] out, may be NULL */
    void* out; int64_t out_ld;                          /* the activation (dtype, NHWC, out_ld elements per pixel), may be NULL */
} msc_bn_input;
int msc_conv_igemm(const msc_conv_desc* d, void* stream);
int msc_conv_stats_slices(const msc_conv_desc* d);   /* depends on d->cfg */
/* configurations of the conv kernel that are valid for a descriptor (for per-layer timing by the caller) */
int msc_conv_num_cfgs(void);
int msc_conv_cfg_ok(const msc_conv_desc* d, int cfg);

/* weight gradient (autograd of the same modules; reference: loss.backward(), src/steps/pytorch/models.py:110)
 *   dw[a][kh][kw][b] += sum_m p[m][a] * q[(y*stride-pad+kh, x*stride-pad+kw)][b]    (fp32 atomics, dw pre-zeroed)
 *   conv: p = dY (a = Cout, Hp x Wp = output grid), q = X (b = Cin);  convT: p = X (a = Cin), q = dOut (b = Cout), stride 2 */
typedef struct msc_wgrad_desc {
    const void* p;
    const void* q;
    float* dw;
    int64_t p_ld, q_ld;
    int32_t dtype;
    int32_t N, Hp, Wp, A, Hq, Wq, B, KH, KW, stride, pad;
    int32_t cfg;   /* 0 = heuristic; 1..msc_conv_wgrad_num_cfgs(): 1 + tile*5 + split, tile 0 = 128x128 when divisible /
                      1 = 64x64 at most / 2 = 32x64 at most, split = index of the target block count
                      {256, 512, 1024, 2048, none} the pixel dimension is split for */
} msc_wgrad_desc;
int msc_conv_wgrad(const msc_wgrad_desc* d, void* stream);
int msc_conv_wgrad_num_cfgs(void);

/* Several weight gradients in one launch per tile shape.  The layers of a ResNet stage (src/unet_models.py:365-368:
 * torchvision layer1..4) are individually too small to fill 256 CUs; their weight gradients depend only on saved
 * activations and on gradients that stay live until the end of backward, so the caller may defer and batch them.
 * steps_per_block > 0: every block runs about that many k-steps (32 bf16 / 16 f32 pixels each) and tiles are capped
 * at tile_cap (128/64/32) -- the policy for grouped launches; steps_per_block == 0: each descriptor's own cfg.
 * The descriptors are copied to the device at creation; the group stays valid until destroyed and may be run any
 * number of times (also inside a hipGraph capture).
 * flags (ABI v8): MSC_WGRAD_ORDERED -- the blocks that share a gradient tile (one per range of pixels) store their partial tile into a
 * plane of their own (device memory owned by the group: sum over layers of splits * sizeof(dw)) and one more launch adds the planes
 * in pixel-range order, instead of fp32 atomics in the order the blocks happen to finish: the gradients, and with msc_final_bwd's
 * ordered_ws a whole training step, are reproducible bit for bit (torch.use_deterministic_algorithms' role for the cuDNN backward the
 * reference runs, src/steps/pytorch/models.py:110).  One descriptor per gradient buffer. */
enum { MSC_WGRAD_ORDERED = 1 };
typedef struct msc_wgrad_group msc_wgrad_group;
int msc_wgrad_group_create(const msc_wgrad_desc* descs, int n, int steps_per_block, int tile_cap, int flags, msc_wgrad_group** out);
int msc_wgrad_group_run(const msc_wgrad_group* g, void* stream);
int msc_wgrad_group_launches(const msc_wgrad_group* g);
/* ABI v11: launch `part` of the group alone (0 .. msc_wgrad_group_launches() - 1; the launches of a group -- one per tile shape -- write
 * disjoint gradient buffers, so a caller may put them on different streams / graph branches; with MSC_WGRAD_ORDERED the LAST part is the
 * pass that adds the planes and must follow all others). */
int msc_wgrad_group_run_part(const msc_wgrad_group* g, int part, void* stream);
void msc_wgrad_group_destroy(msc_wgrad_group* g);

/* fp32 master weight -> compute copy.  msc_pack_cast: same layout.  msc_pack_transpose: [A][T][B] -> [B][T][A]
 * (data-gradient / ConvTranspose2d operand).  msc_stem_pack: conv1 weight [64][3][7][7] (torch layout,
 * src/unet_models.py:360) -> [64][7][8][4] (kh, kw padded to 8, ci padded to 4); msc_stem_unpack_grad is its adjoint. */
int msc_pack_cast(const float* src, void* dst, int dtype, int64_t n, void* stream);
int msc_pack_transpose(const float* src, void* dst, int dtype, int A, int T, int B, void* stream);
/* the same for MANY tensors in one launch: block b handles items[block_item[b]], piece block_local[b]
 * (kind 0: 2048-element piece of a cast; kind 1: one 64x32 (A x B) tile of tap t of a transpose, local index
 * = (t*ceil(A/64) + a_tile)*ceil(B/32) + b_tile).  All three tables live in device memory. */
typedef struct msc_pack_item {
    const float* src;
    void* dst;
    int32_t kind, A, T, B;
    int64_t n;
} msc_pack_item;
int msc_pack_multi(const msc_pack_item* items, const int32_t* block_item, const int32_t* block_local, int nblocks,
                   int dtype, void* stream);
int msc_stem_pack(const float* w, void* dst, int dtype, int cout, void* stream);
/* wire format of the data-parallel gradient exchange (replaces nn.DataParallel's reduce-add, src/models.py:65): the fp32
 * gradient range is cast to `dtype` (msc_pack_cast), all-to-all'ed into recv[world][shard]; msc_grad_reduce sums the
 * `world` shards in fp32 and rounds once: out[i] = dtype(sum_w recv[w][i]); msc_grad_unpack widens the all-gathered
 * result back: g[i] = float(in[i]). */
int msc_grad_reduce(const void* recv, void* out, int dtype, int world, int64_t shard, void* stream);
int msc_grad_unpack(const void* in, float* g, int dtype, int64_t n, void* stream);
int msc_stem_unpack_grad(const float* dpacked, float* dw, int cout, void* stream);

/* network input: x f32 NCHW [N,3,H,W] (what the loaders hand to model(X), src/steps/pytorch/models.py:92)
 * -> zero-padded NHWC4 image [N][H+6][W+8][4] (3 px top/left) so the 7x7/2 stem becomes an implicit GEMM. */
int msc_stem_prepare(const float* x, void* xp, int dtype, int N, int H, int W, void* stream);

/* MaxPool2d(2,2) (src/unet_models.py:356,363,392) forward / backward (first maximum wins, as torch) */
int msc_maxpool2_fwd(const void* in, int64_t in_ld, void* out, int64_t out_ld, int dtype, int N, int Ho, int Wo, int C, void* stream);
int msc_maxpool2_bwd(const void* dout, int64_t dout_ld, const void* in, int64_t in_ld, void* din, int64_t din_ld,
                     int dtype, int N, int Ho, int Wo, int C, int accumulate, void* stream);

/* BatchNorm2d, training mode (torchvision ResNet BNs; eps 1e-5, momentum 0.1).  The statistics arrive as per-XCD partial
 * sums, double slots[MSC_BN_SLOTS][C][2] (from msc_conv_igemm's `stats` or msc_bn_bwd_reduce), and are finalised in the prologue
 * of the kernel that needs them -- there is no separate finalize launch.
 * apply:  (sum y, sum y^2) -> batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale;
 *         out = act(y*scale + shift (+ res)); also writes scale / shift / save_mean / save_invstd and updates the running
 *         statistics (unbiased variance).  slots == NULL: scale / shift are inputs (no statistics involved). */
#define MSC_BN_SLOTS 8
/* stream-ordered fill / device-to-device copy: with these the whole training step is a list of C-ABI launches (capturable
 * as hipGraph nodes, interpretable on the host by the tests) */
/* (round 5: both are KERNEL launches -- captured into a hipGraph, hipMemsetAsync / hipMemcpyAsync become memset / memcpy nodes, and replayed training
 * steps holding such nodes were seen to run with garbage gradients; MSC_MEMOPS_KERNEL=0 restores the runtime calls for A/B) */
int msc_memset_zero(void* ptr, int64_t bytes, void* stream);
int msc_copy(void* dst, const void* src, int64_t bytes, void* stream);
int msc_bn_apply(const void* y, int64_t y_ld, const void* res, int64_t res_ld, void* out, int64_t out_ld,
                 const double* slots, int64_t count, const float* gamma, const float* beta, float eps, float momentum,
                 float* running_mean, float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                 uint8_t* relu_mask, int64_t relu_mask_ld, int relu, int dtype, int64_t pixels, int C, void* stream);
/* eval mode (model.eval(), src/steps/pytorch/models.py:116): scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale */
int msc_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                float eps, float* scale, float* shift, int C, void* stream);
/* Eval-mode fused Bottleneck (ABI v5): one launch for the torchvision ResNet bottleneck block in its identity form (stride 1,
 * no downsample branch) -- conv1x1 -> bn -> ReLU -> conv3x3 -> bn -> ReLU -> conv1x1 -> bn -> (+x) -> ReLU -- that the
 * reference's encoder stages layer1..layer3 are made of (`self.encoder.layer1..3` used at src/unet_models.py:345-351,365-371:
 * 27 of ResNet101's 33 blocks, 44 of ResNet152's 50).  BatchNorm is folded (msc_bn_fold); the two Cmid-channel intermediates
 * stay in LDS.  x: NHWC [N][H][W][x_ld], 4*Cmid channels (also the residual); out likewise; Cmid in {64, 128, 256};
 * W % 16 == 0, H % (patch rows) == 0; 16-bit dtypes.  wpk: the three weight tensors ([Cmid][4Cmid], [Cmid][3][3][Cmid],
 * [4Cmid][Cmid] in dtype, as msc_conv_igemm takes them) re-ordered by msc_bottleneck_pack into the fragment-major stream the
 * kernel reads (msc_bottleneck_pack_bytes bytes).  cfg: patch rows (0 = chosen by shape).  msc_bottleneck_ok: 1 if the fused
 * kernel takes the descriptor (otherwise the caller runs the three msc_conv_igemm launches). */
typedef struct msc_bneck_desc {
    const void* x;
    void* out;
    const void* wpk;
    const float* scale1; const float* shift1;
    const float* scale2; const float* shift2;
    const float* scale3; const float* shift3;
    int64_t x_ld, out_ld;
    int32_t dtype, N, H, W, Cmid, cfg;
} msc_bneck_desc;
int msc_bottleneck_ok(const msc_bneck_desc* d);
int64_t msc_bottleneck_pack_bytes(int Cmid);
int msc_bottleneck_pack(const void* w1, const void* w2, const void* w3, void* wpk, int Cmid, int dtype, void* stream);
int msc_bottleneck_fused(const msc_bneck_desc* d, void* stream);
/* backward of out = relu?(bn(y) (+res)):  dh = dout * [out>0] (relu 1) or dout * [scale*y+shift > 0] (relu 2);
 * reduce: slots[xcd][c] += (sum dh, sum dh*y)   (the data-gradient conv that writes dout can do this in its epilogue instead);
 * apply:  prologue: dbeta += sum dh, dgamma += sum dh*xhat (into the fp32 gradients), dy = a*dh + b*y + k per channel;
 *         dres (optional) = dh (or += if dres_acc). */
int msc_bn_bwd_reduce(const void* dout, int64_t dout_ld, const void* out, int64_t out_ld, const void* y, int64_t y_ld,
                      int relu, const float* scale, const float* shift, double* slots, int dtype, int64_t pixels, int C,
                      void* stream);
/* msc_bn_apply relu_mask (ABI v7, may be NULL; needs relu): one byte per 16-byte channel vector and pixel, relu_mask_ld bytes per pixel, bit e
 * = [out channel (16 / sizeof(dtype)) * k + e > 0].  msc_bn_bwd_apply relu 3: `out` points at such a mask (out_ld bytes per pixel) instead
 * of the activation -- the residual joins' backward reads 1/16 of the bytes for its ReLU mask. */
int msc_bn_bwd_apply(const void* dout, int64_t dout_ld, const void* out, int64_t out_ld, const void* y, int64_t y_ld,
                     int relu, const float* scale, const float* shift, const double* slots, int64_t count, const float* gamma,
                     const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* dy, int64_t dy_ld,
                     void* dres, int64_t dres_ld, int dres_acc, const void* res_y, int64_t res_y_ld, double* res_slots, int dtype,
                     int64_t pixels, int C, void* stream);
/* res_y / res_slots (ABI v7, may be NULL): the launch also adds (sum dh, sum dh*res_y) to res_slots -- the BatchNorm-backward sums of the
 * layer that produced the residual (the downsample branch of a stage's first block, whose output gradient is the dh written to dres and
 * whose pre-BN tensor is res_y): no msc_bn_bwd_reduce launch for that layer. */

/* The stem's BatchNorm2d (training) + ReLU + MaxPool2d(2,2) (self.conv1 = Sequential(encoder.conv1, encoder.bn1, encoder.relu, self.pool),
 * src/unet_models.py:360-363) without materialising the full-resolution activation (ABI v7): nothing but the pool reads it.
 *   msc_bn_apply_pool:      y [N, 2Ho, 2Wo, C] (raw conv output, statistics in `slots` as for msc_bn_apply) -> out [N, Ho, Wo, C] =
 *                           max over the 2x2 window of relu(scale*y + shift); publishes scale / shift / mean / invstd / running statistics
 *   msc_bn_pool_bwd_reduce: slots += (sum dh, sum dh*y), dh = the pooled gradient at the window's FIRST maximum of relu(scale*y + shift)
 *                           (torch's tie rule) if that maximum is positive, zero elsewhere -- MaxPool backward + ReLU backward + the sums
 *                           of msc_bn_bwd_reduce from (dpool, y) alone
 *   msc_bn_pool_bwd_apply:  y <- dy = g*invstd*(dh - mean(dh) - xhat*mean(dh*xhat)) in place; dgamma / dbeta += as msc_bn_bwd_apply */
int msc_bn_apply_pool(const void* y, int64_t y_ld, void* out, int64_t out_ld, const double* slots, int64_t count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                      float* save_mean, float* save_invstd, int dtype, int N, int Ho, int Wo, int C, void* stream);
int msc_bn_pool_bwd_reduce(const void* dpool, int64_t dpool_ld, const void* y, int64_t y_ld, const float* scale, const float* shift,
                           double* slots, int dtype, int N, int Ho, int Wo, int C, void* stream);
int msc_bn_pool_bwd_apply(const void* dpool, int64_t dpool_ld, void* y, int64_t y_ld, const float* scale, const float* shift,
                          const double* slots, int64_t count, const float* gamma, const float* save_mean, const float* save_invstd,
                          float* dgamma, float* dbeta, int dtype, int N, int Ho, int Wo, int C, void* stream);

/* ReLU backward for the decoder (ConvRelu / deconv+ReLU): dx = dy*[y>0], optionally dx += */
int msc_relu_bwd(const void* dy, int64_t dy_ld, const void* y, int64_t y_ld, void* dx, int64_t dx_ld,
                 int accumulate, int dtype, int64_t pixels, int C, void* stream);
/* per-channel bias gradient: db[c] += sum_pixels dy[p][c]  (conv bias of ConvRelu / ConvTranspose2d);
 * workspace: msc_bias_grad_workspace_bytes() bytes */
int64_t msc_bias_grad_workspace_bytes(int64_t pixels, int C, int dtype);
int msc_bias_grad(const void* dy, int64_t dy_ld, float* db, void* workspace, int dtype, int64_t pixels, int C, void* stream);
/* both in one pass over the tensors: dx = dy*[y>0] (dx may alias dy) and db[c] += sum_pixels dx[p][c]; same workspace */
int msc_relu_bias_grad(const void* dy, int64_t dy_ld, const void* y, int64_t y_ld, void* dx, int64_t dx_ld, float* db,
                       void* workspace, int dtype, int64_t pixels, int C, void* stream);

/* bias gradient out of the slots a stats_kind-2 conv filled ([MSC_BN_SLOTS][Cs][2] doubles): db[c] += sum_slots [.][c][0], c < C <= Cs */
int msc_bias_slots_finalize(const double* slots, int Cs, float* db, int C, void* stream);
/* several layers in one launch; `items` is a HOST array of n <= MSC_BIAS_SLOTS_MAX entries, copied into the launch */
#define MSC_BIAS_SLOTS_MAX 16
typedef struct msc_bias_slots_item {
    const double* slots;
    float* db;
    int32_t Cs, C;
} msc_bias_slots_item;
int msc_bias_slots_finalize_multi(const msc_bias_slots_item* items, int n, void* stream);

/* final 1x1 conv 32 -> 2 with bias (src/unet_models.py:383,403; dropout p = 0) fused with the channel softmax
 * the reference applies on the host afterwards (src/models.py:88-92, src/utils.py:231-273).
 * in: NHWC dtype [pixels][C]; w f32 [2][C]; logits / probs: f32 NCHW [N,2,H,W] (either may be NULL). */
int msc_final_fwd(const void* in, int64_t in_ld, const float* w, const float* b, float* logits, float* probs,
                  int dtype, int N, int H, int W, int C, void* stream);
/* backward: dlogits f32 NCHW -> din[p][c] = sum_k dlogits[k][p]*w[k][c] (masked by in>0: the ReLU of dec0),
 * dw[k][c] += sum_p dlogits[k][p]*in[p][c], db[k] += sum_p dlogits[k][p];
 * dbias_in (f32[C], may be NULL): dbias_in[c] += sum_p din[p][c] -- the bias gradient of the layer that produced `in`
 * (dec0's conv bias), so no separate msc_bias_grad pass reads din back.  C*sizeof(dtype) must be a multiple of 16, C <= 64.
 * ordered_ws (ABI v8, may be NULL: fp32 atomics, the order of the additions varies from run to run): MSC_FINAL_BWD_WS_ROWS * (3*C + 2)
 * floats of scratch.  The per-block sums go there and a second launch adds them up in a fixed order: bit-for-bit reproducible. */
#define MSC_FINAL_BWD_WS_ROWS 1024
int msc_final_bwd(const float* dlogits, const void* in, int64_t in_ld, const float* w, void* din, int64_t din_ld,
                  float* dw, float* db, float* dbias_in, float* ordered_ws, int dtype, int N, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------- losses / optimizer ----------
 * mixed distance-weighted cross entropy + soft Dice (src/models.py:310-454, validation.py:8-16) or plain CE
 * (validation.py:25-28).  logits f32 NCHW [N,2,H,W]; target f32 NCHW [N,tc,H,W] (tc = 1: class only, tc = 3:
 * class, distance, sqrt(size)).  Two phases so the three Dice sums can be all-reduced across ranks between them:
 *   msc_loss_sums:  sums[0..3] = (sum_p w*ce, sum p1*t, sum p1, sum t)   (double accumulation)
 *   msc_loss_grad:  loss[0] = ce_weight*sums0/total_pixels + dice_weight*(1-(2*s1+smooth)/(s2+s3+smooth+eps));
 *                   dlogits = d loss / d logits (NCHW f32), scaled by grad_scale */
typedef struct msc_loss_cfg {
    float w0, sigma, size_c;     /* get_weights(): 1 + w0*exp(-d^2/sigma^2), C = sqrt(h*w)/2 */
    float dice_weight, ce_weight, smooth, eps;
    int32_t weighted;            /* 0: plain CE (target channel 0 only), 1: distance/size weighted */
    int32_t dice_sigmoid;        /* ABI v10: Dice activation, 0 = Softmax2d, 1 = Sigmoid of the class-1 logit (src/models.py:437-442) */
} msc_loss_cfg;
int msc_loss_sums(const float* logits, const float* target, int tc, const msc_loss_cfg* cfg, double* sums,
                  int N, int H, int W, void* stream);
int msc_loss_grad(const float* logits, const float* target, int tc, const msc_loss_cfg* cfg, const double* sums,
                  double total_pixels, float grad_scale, const float* scale_state, float* loss, float* dlogits, int N, int H, int W,
                  void* stream);
/* scale_state (device, may be NULL; ABI v7): the optimizer state below -- dlogits are additionally multiplied by its dynamic loss
 * scale state[MSC_OPT_SCALE] (> 0), read on the device so that a captured hipGraph replays with the current scale. */

/* Optimizer state in device memory, f32[MSC_OPT_STATE] (ABI v7; was {step, lr}): what a captured step must read at REPLAY time.
 *   STEP      Adam's step count (bias corrections)            LR       learning rate
 *   OVERFLOW  raised by msc_grad_check when a gradient element is not finite
 *   SKIP      set by msc_adam_tick for the current step: msc_adam_step / msc_adam_pack leave p, m, v untouched
 *   SCALE     loss scale: msc_loss_grad multiplies dlogits by it (0: no scaling)
 *   GOOD / GROWTH  clean steps since the last change of SCALE / clean steps after which SCALE doubles (0: static scale)
 *   SKIPPED   number of skipped steps so far
 *   UNSCALE   1 / SCALE as it was when this step's loss gradient was scaled (written by msc_adam_tick BEFORE it changes SCALE): the
 *             factor the Adam kernels apply to the gradient (0: none -- a state nobody ticked yet)
 * msc_adam_tick: OVERFLOW set -> SKIP = 1, SCALE halves (not below 1), STEP unchanged; else SKIP = 0, STEP += 1, GOOD counted.
 * Replaces the reference's plain optimizer.step() (src/steps/pytorch/models.py:111), which has no 16-bit mode to protect. */
enum { MSC_OPT_STEP = 0, MSC_OPT_LR = 1, MSC_OPT_OVERFLOW = 2, MSC_OPT_SKIP = 3, MSC_OPT_SCALE = 4, MSC_OPT_GOOD = 5, MSC_OPT_GROWTH = 6,
       MSC_OPT_SKIPPED = 7, MSC_OPT_UNSCALE = 8, MSC_OPT_STATE = 12 };

/* Adam with L2 folded into the gradient (torch.optim.Adam(weight_decay), src/models.py:57,287-292) over one flat
 * fp32 parameter buffer.  `state` (device f32[MSC_OPT_STATE], may be NULL): when given it overrides `step` and `lr`, divides
 * grad_scale by its loss scale and makes a skipped step a no-op. */
int msc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, const float* state, void* stream);
int msc_adam_tick(float* state, void* stream);
/* state[MSC_OPT_OVERFLOW] = 1 if any of g[0..n) is NaN or +-inf (launched between the backward / gradient exchange and
 * msc_adam_tick in the fp16 mode) */
int msc_grad_check(const float* g, int64_t n, float* state, void* stream);
/* The same update over a TABLE of tensors inside the flat buffers that ALSO writes the compute copies of the weights (ABI v7): the
 * conv weights' 16-bit copy in their own layout (`direct`) and the [B][T][A] transpose of the [A][T][B] master (`trans`; the
 * data-gradient / ConvTranspose2d operand) leave the kernel that has the updated fp32 values in registers -- msc_pack_multi after
 * the optimizer re-read every master (602 MB per ResNet101 step).  Block b handles items[block_item[b]], piece block_local[b]:
 * without `trans` 2048 consecutive elements, with it one 64x32 (A x B) tile of tap t, local index = (t*ceil(A/64) + a_tile)*
 * ceil(B/32) + b_tile, B % 4 == 0.  `off` (elements, multiple of 4) locates the tensor in p / g / m / v; tables in device memory. */
typedef struct msc_adam_item {
    int64_t off, n;
    void* direct;
    void* trans;
    int32_t A, T, B, reserved;
} msc_adam_item;
int msc_adam_pack(float* p, const float* g, float* m, float* v, const msc_adam_item* items, const int32_t* block_item,
                  const int32_t* block_local, int nblocks, int dtype, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, const float* state, void* stream);

/* ---------------------------------------------------------------- mask post-processing --------
 * Batched over B images; each replaces a per-image Python/scipy/skimage loop of src/postprocessing.py. */
/* resize_image (src/postprocessing.py:48-61 = skimage.transform.resize(order 1, mode 'constant') -> scipy map_coordinates): bilinear with
 * half-pixel centres, scipy's 'constant' edge rule (no interpolation beyond the edges: cval 0), skimage's clip to the input range joined with 0.
 * The interpolant is evaluated in double with numpy's roundings (no fused multiply-add); f32 [B,C,h,w] -> [B,C,H,W], stored as float64
 * (`out_f64` = 1: what the reference function returns) or rounded to float32.  `minmax_ws`: float[2*B] scratch (ABI v10). */
int msc_resize_bilinear(const float* in, void* out, int out_f64, float* minmax_ws, int B, int C, int h, int w, int H, int W, void* stream);
/* ABI v10: resize_image + categorize_multilayer_image in one pass (src/postprocessing.py:48-61,77-84; Steps 'mask_resize' -> 'category_mapper',
 * src/pipelines.py:249-268): the reference thresholds the FLOAT64 map the resize returns, so the layers are compared in double against the
 * double interpolant; `out` gets the map rounded to float32 (what the scoring reads). */
int msc_resize_threshold(const float* in, float* out, uint8_t* layers, float* minmax_ws, int B, int C, int h, int w, int H, int W,
                         const int32_t* layer_class, const double* layer_thr, int L, void* stream);
/* crop_image_center_per_class (src/postprocessing.py:239-258) */
int msc_crop_center(const float* in, float* out, int B, int C, int h, int w, int hc, int wc, void* stream);
/* categorize_multilayer_image (src/postprocessing.py:77-84): layers[b][l] = probs[b][cls(l)] > thr(l); u8.  ABI v10: thresholds are the
 * reference's float64 values (np.arange(1/(L+1), 1, 1/(L+1))) and the comparison is in double (numpy: array > float64 scalar); `probs` is
 * float32 or, with `probs_f64` = 1, the float64 map msc_resize_bilinear(out_f64 = 1) wrote */
int msc_threshold_layers(const void* probs, int probs_f64, uint8_t* layers, int B, int C, int H, int W,
                         const int32_t* layer_class, const double* layer_thr, int L, void* stream);
/* categorize_image (src/postprocessing.py:64-74): argmax over channels (first maximum), int32 [B,H,W] */
int msc_argmax_channels(const float* probs, int32_t* out, int B, int C, int H, int W, void* stream);
/* erosion / dilation with a k x k rectangle, skimage origin convention (src/postprocessing.py:148-154,172-179):
 * window offsets -((k-1)/2) .. -((k-1)/2)+k-1 on both axes, clamped border; u8 or int32 images [B,H,W] */
int msc_erode_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int k, void* stream);
int msc_dilate_i32(const int32_t* in, int32_t* out, int B, int H, int W, int k, void* stream);
/* min / max over the window offsets lo..hi (lo <= 0 <= hi) on both axes, reflected border; u8 images [B,H,W] */
int msc_rect_filter_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int lo, int hi, int is_max, void* stream);
/* label (src/utils.py:328-330 = scipy.ndimage.label, 4-connectivity, labels 1..n in raster order of each
 * component's first pixel): mask u8 [B,H,W] (nonzero = foreground) -> int32 labels; counts[b] = n.
 * workspace: msc_label_workspace_bytes(B,H,W) bytes. */
int64_t msc_label_workspace_bytes(int B, int H, int W);
int msc_label4(const uint8_t* mask, int32_t* labels, int32_t* counts, void* workspace, int B, int H, int W, void* stream);
/* add_dropped_objects (src/utils.py:333-339): out = processed + [component of `original` with no surviving pixel];
 * labels_orig = msc_label4(original); u8 result.  As in the reference, which tests np.any(np.where(overlap)) -- the index
 * arrays -- a component whose only surviving pixel is (0,0) counts as dropped.  bool_sum: `+=` on the bool masks the
 * reference feeds it is a logical or (values stay 0/1); 0 = integer sum (the surviving (0,0) pixel becomes 2). */
int msc_add_dropped(const uint8_t* processed, const int32_t* labels_orig, uint8_t* out, void* workspace,
                    int B, int H, int W, int bool_sum, void* stream);
/* EXTENSION, no reference function (WATERSHED.md): marker-controlled watershed.  labels int32 [B,H,W] holds the markers
 * (> 0) on entry and the flooded labels on return; mask u8 [B,H,W] bounds the flood; prob f32 [B,H,W] gives the relief
 * h = clamp(floor((1-prob)*255), 0, 255); synchronous immersion, ties to the smaller label.  One workgroup per [H,W] plane.
 * workspace: msc_watershed_workspace_bytes(B,H,W) bytes, 4-byte aligned. */
int64_t msc_watershed_workspace_bytes(int B, int H, int W);
int msc_watershed(const float* prob, const uint8_t* mask, int32_t* labels, void* workspace, int B, int H, int W, void* stream);
/* build_score (src/postprocessing.py:228-236): per label, mean(prob over label) * sqrt(area).
 * labels int32 [B,H,W], probs f32 [B,H,W] (the matching channel); sums f64 [B][max_labels], areas i32 [B][max_labels]
 * are zeroed by the call; score[b][l-1] = sums/areas*sqrt(areas) (f64). */
int msc_build_score(const int32_t* labels, const float* probs, double* sums, int32_t* areas, double* score,
                    int B, int H, int W, int max_labels, void* stream);
/* dense_crf (src/postprocessing.py:183-225), exact windowed mean field (see oracle/crf_ref.py: PARITY UNPINNED):
 * probs f32 [B,2,H,W], rgb u8 [B,H,W,3] -> out f32 [B,2,H,W]; workspace msc_crf_workspace_bytes() */
int64_t msc_crf_workspace_bytes(int B, int H, int W, int radius);
int msc_dense_crf(const float* probs, const uint8_t* rgb, float* out, void* workspace, int B, int H, int W,
                  float sxy_g, float compat_g, float sxy_b, float srgb, float compat_b, int iterations, void* stream);

/* ---------------------------------------------------------------- test-time augmentation --------
 * src/loaders.py:401-517 on the device.  specs[v]: bit 0 ud flip, bit 1 lr flip (ud wins, like the reference's elif
 * chain), bits 2-3 rotation/90 counter-clockwise.  transform: x f32 [N,C,H,W] -> out [V,N,C,H,W] (variant v of every
 * image); aggregate: preds [V,N,C,H,W] -> out [N,C,H,W], inverse transform then method 0 mean / 1 gmean / 2 max / 3 min
 * (TestTimeAugmentationAggregator.agg_method).  specs is a device array; any_quarter_turn tells whether H == W is needed. */
int msc_tta_transform(const float* x, float* out, const int32_t* specs, int N, int C, int H, int W, int V, int any_quarter_turn, void* stream);
int msc_tta_aggregate(const float* preds, float* out, const int32_t* specs, int N, int C, int H, int W, int V, int method,
                      int any_quarter_turn, void* stream);

/* ---------------------------------------------------------------- annotation encoding ------------
 * src/utils.py:61-127 (decompose -> rle_from_binary -> bounding_box_from_rle, i.e. pycocotools 2.0.0 maskApi.c
 * rleEncode / rleToString / rleToBbox) for every instance of every layer in two calls, both synchronous on `stream`
 * (they return counts the caller sizes the next buffer with):
 *   msc_rle_segments: labels i32 [layers,H,W] (0 = background, instance ids < 2^24) -> column-major runs of equal
 *     label, kept in `ws` (msc_rle_segments_workspace bytes); *nseg = number of runs.
 *   msc_rle_encode: runs -> instances sorted by (layer, label).  *table -> i32 [n_inst][8] = layer, label, string
 *     begin, string end (byte offsets into *chars), xs, ys, xe, ye (inclusive box; COCO bbox = [xs, ys, xe-xs+1,
 *     ye-ys+1]); *chars -> the concatenated COCO count strings.  Both point into `ws` (device memory,
 *     msc_rle_encode_workspace(nseg) bytes).  Instance ids with no pixel do not appear. */
int64_t msc_rle_segments_workspace(int layers, int H, int W);
int msc_rle_segments(const int32_t* labels, int layers, int H, int W, void* ws, int64_t ws_bytes, int32_t* nseg, void* stream);
int64_t msc_rle_encode_workspace(int nseg);
int msc_rle_encode(const void* seg_ws, int layers, int H, int W, int nseg, void* ws, int64_t ws_bytes, int32_t* n_inst,
                   int64_t* n_chars, const int32_t** table, const char** chars, void* stream);

/* The annotation list as JSON text -- what create_annotations (src/utils.py:76-115) passes to json.dumps for submission.json --
 * written on the HOST from the encoder's table (copied from the device): table i32 [n_inst][8] and chars as msc_rle_encode
 * returns them; per encoded layer k: image_ids[k], category_ids[k], counts[k] scores at scores[score_off[k] ...] (instance id - 1
 * indexes them).  Instances 1 .. min(largest id present, counts[k]) per layer, ids without pixels as empty masks (decompose(),
 * src/utils.py:61-73).  Returns the number of bytes of the document; it is written to `out` when that is <= cap. */
int64_t msc_annotations_json(const int32_t* table, int n_inst, const char* chars, int layers, const int64_t* image_ids,
                             const int32_t* category_ids, const int32_t* counts, const double* scores, const int64_t* score_off,
                             int H, int W, char* out, int64_t cap);

/* ---------------------------------------------------------------- target preparation -------------
 * overlay_mask_one_image (src/preparation.py:44-84) for erode = dilate = 0 (neptune.yaml:69-70) from the decoded
 * instance masks of ONE image, masks u8 [n,H,W] in annotation order, category_nr i32 [n] (NULL: all 1):
 *   mask_overlayed u8 [H,W]; distances_f16 [H,W] = float16 bits of (nearest + second nearest instance distance);
 *   second_nearest f64 [H,W] (may be NULL); kept i32 [n] (may be NULL): 0 skipped by is_on_border(mask, 2),
 *   1 used, 2 overlayed but discarded by update_distances' `dist.sum() == 0` rule.  border_masks (may be NULL = masks):
 *   the tensors is_on_border() is applied to.  All pointers device memory.
 * msc_prep_border: the optional border class (:73-76), in place; synchronous (needs mask.max()).
 * msc_size_matrix: get_size_matrix (:181-187) from msc_label4 labels: sizes[p] = area of p's component, 1 on
 *   background; areas = scratch i32 [B][max_labels+1]. */
int64_t msc_prep_workspace_bytes(int n, int H, int W);
int msc_prep_targets(const uint8_t* masks, const uint8_t* border_masks, const int32_t* category_nr, int n, int H, int W,
                     uint8_t* mask_overlayed, uint16_t* distances_f16, double* second_nearest, int32_t* kept, void* workspace,
                     void* stream);
/* the eroded / eroded+dilated variants (src/preparation.py:61-77,121-143,166-178): chosen[i] = binary_erosion(mask_i,
 * rectangle(erode, erode)) if mask_i has more than small_annotations_size^2 pixels, else mask_i (dilate == 0) or
 * binary_dilation(mask_i, rectangle(dilate, dilate)).  msc_prep_targets then takes `chosen` as masks and the
 * annotations themselves as border_masks (is_on_border looks at the annotation).  msc_prep_paint: overlay[p] = value
 * where mask[p] (np.where(mask, category_nr, mask_overlayed), :79). */
int64_t msc_prep_morph_workspace_bytes(int n, int H, int W);
int msc_prep_morph(const uint8_t* masks, int n, int H, int W, int erode, int dilate, int small_annotations_size, uint8_t* chosen,
                   void* workspace, void* stream);
int msc_prep_paint(uint8_t* mask_overlayed, const uint8_t* mask, int value, int H, int W, void* stream);
int msc_prep_border(uint8_t* mask_overlayed, const double* second_nearest, double border_width, int32_t* scratch, int H, int W,
                    void* stream);
int msc_size_matrix(const int32_t* labels, int32_t* sizes, int32_t* areas, int B, int H, int W, int max_labels, void* stream);

#ifdef __cplusplus
}
#endif
#endif
