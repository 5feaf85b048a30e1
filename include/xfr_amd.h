/*
 * xfr_amd.h -- C ABI of the MI355X-native excitation-backprop (EBP) saliency engine.
 *
 * Drop-in boundary for the hot path of stresearch/xfr:
 *     python/xfr/models/whitebox.py:482-558   Whitebox.ebp / contrastive_ebp / truncated_contrastive_ebp
 *     python/xfr/models/whitebox.py:742-785   Whitebox.encode / embeddings
 * The reference has NO native layer (it is PyTorch hooks + autograd); this library replaces the three
 * hooked forwards and the autograd sweep of whitebox.py:490-498 with a static layer program executed by
 * hand-written HIP kernels for gfx950.  Each entry point below cites the reference code it stands in for.
 *
 * Conventions
 *   - plain C types only; all device pointers are raw HIP device pointers (float32, contiguous);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); no entry point synchronises
 *     the stream unless documented;
 *   - every function returns an xfr_status; on failure xfr_last_error() holds a thread-local message;
 *   - one engine per (process, device); calls on one engine must be serialised by the caller
 *     (the reference is equally non-re-entrant: whitebox.py:291-296 mutable hook state);
 *   - the engine never mutates caller buffers other than the documented outputs (the reference leaves
 *     W+ installed in the caller's modules after ebp(): whitebox.py:371-377);
 *   - a call is bit-reproducible for a given batch size and settings.  ACROSS batch sizes a sample's map is the same
 *     arithmetic in a different summation order, not the same bits: three defaults look at the batch -- the lean schedule
 *     (batch % 4 == 0, xfr_engine_set_lean), the bf16x6 kernel (grids of >= 128 tiles, xfr_engine_set_split_gemm) and tail
 *     balancing (xfr_engine_set_tail_balance).  With all three off every launch of a layer runs one kernel in one K order.
 */
#ifndef XFR_AMD_H
#define XFR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XFR_AMD_ABI_VERSION 6

typedef enum {
    XFR_OK = 0,
    XFR_INVALID_ARG = 1,        /* -> ValueError / AssertionError in the Python mirror            */
    XFR_UNSUPPORTED_LAYER = 2,  /* -> ValueError (whitebox.py:403 Sigmoid/ELU/Tanh, unknown kinds) */
    XFR_OOM = 3,
    XFR_HIP_ERROR = 4,
    XFR_STATE_ERROR = 5,        /* e.g. weights not loaded */
    XFR_RCCL_ERROR = 6          /* librccl missing, or a collective failed */
} xfr_status;

/* Layer program op kinds.  "Hooked" kinds are leaf nn.Module calls of the reference (they receive the
 * pre-forward/forward hooks of whitebox.py:306-437); G_* kinds are functional glue between modules
 * (torch.add, torch.max, F.normalize) that the reference's hooks never see. */
typedef enum {
    XFR_OP_CONV = 1,        /* nn.Conv2d                      resnet.py:116-122,177; lightcnn.py:53        */
    XFR_OP_BATCHNORM = 2,   /* nn.BatchNorm2d (eval)          resnet.py:118,179                             */
    XFR_OP_RELU = 3,        /* nn.ReLU(inplace=True)          resnet.py:124                                 */
    XFR_OP_MAXPOOL = 4,     /* nn.MaxPool2d                   resnet.py:181; resnet50_128.py:16 (ceil_mode)  */
    XFR_OP_AVGPOOL = 5,     /* nn.AvgPool2d                   resnet.py:186,211; lightcnn.py:237            */
    XFR_OP_ADD = 6,         /* Add module (2 inputs)          resnet.py:104-108; lightcnn.py:33-37          */
    XFR_OP_CONCAT = 7,      /* ConcatChannels (zero pad)      resnet.py:152-157                             */
    XFR_OP_MULTIPLY = 8,    /* Multiply(n)                    resnet.py:160-165                             */
    XFR_OP_LINEAR = 9,      /* nn.Linear (on flattened CHW)   resnet.py:187-189; lightcnn.py:228-229        */
    XFR_OP_SPLIT = 10,      /* Split module (identity view)   lightcnn.py:39-45                             */
    XFR_OP_G_ADD = 11,      /* functional a+b                 resnet50_128.py:187; lightcnn.py:252          */
    XFR_OP_G_MAXHALVES = 12,/* torch.max(split[0], split[1])  lightcnn.py:62                                */
    XFR_OP_G_NORMALIZE = 13 /* F.normalize(p=2, dim=1)        resnet.py:250                                 */
} xfr_op_kind;

/* One record of the static layer program (call order == the reference's forward call order).
 * Tensor id 0 is the network input; every op defines exactly one new tensor id `out`
 * (ids must be 1,2,3,... in op order).  All tensors are logically N x C x H x W. */
typedef struct {
    int32_t kind;        /* xfr_op_kind */
    int32_t in0;         /* first input tensor id  */
    int32_t in1;         /* second input tensor id (ADD, G_ADD) or -1 */
    int32_t out;         /* output tensor id */
    int32_t cout;        /* CONV/LINEAR: output channels; CONCAT: number of zero copies appended (resnet.py:157) */
    int32_t kh, kw;      /* CONV/MAXPOOL/AVGPOOL kernel; LINEAR: must equal the input H, W */
    int32_t stride;
    int32_t pad;
    int32_t ceil_mode;   /* MAXPOOL */
    int32_t inplace;     /* RELU: 1 = nn.ReLU(inplace=True) (hook lands on the output tensor) */
    float   fparam;      /* MULTIPLY: n;  BATCHNORM: eps */
    int32_t w_weight;    /* index into the weight table, or -1 */
    int32_t w_bias;      /* index into the weight table, or -1 */
    int32_t w_mean;      /* BATCHNORM running_mean */
    int32_t w_var;       /* BATCHNORM running_var  */
} xfr_op_desc;

/* A host-memory view of one parameter tensor (float32, contiguous, PyTorch layout:
 * Conv [Cout][Cin][kh][kw], Linear [out][in], BatchNorm vectors [C]). */
typedef struct {
    const float* data;
    int64_t      numel;
} xfr_tensor_view;

/* ebp_subtree_mode of whitebox.py:262-263,396-430 */
typedef enum {
    XFR_MODE_AFFINEONLY = 0,
    XFR_MODE_AFFINEONLY_WITH_PRIOR = 1,
    XFR_MODE_NORELU = 2,
    XFR_MODE_ALL = 3
} xfr_subtree_mode;

typedef struct xfr_engine xfr_engine;

/* Version of this ABI (XFR_AMD_ABI_VERSION of the library that was loaded). */
int32_t xfr_abi_version(void);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* xfr_last_error(void);

/* Build an engine for one backbone on HIP device `device`.
 * Stands in for: Whitebox.__init__ (whitebox.py:261-304: hook installation over every leaf module) plus the
 * backbone constructors (resnet.py:168-198, resnet50_128.py:6-170, lightcnn.py:216-247).
 * `ops`/`n_ops`: the layer program; `n_weights`: size of the weight table the ops index into;
 * (in_c,in_h,in_w): input image shape; `max_batch`: largest N accepted by the run calls.
 * Workspace for activations (A), positive activations (X) and two gradient streams is allocated here. */
xfr_status xfr_engine_create(const xfr_op_desc* ops, int32_t n_ops, int32_t n_weights,
                             int32_t in_c, int32_t in_h, int32_t in_w,
                             int32_t max_batch, int32_t device, xfr_engine** out);

xfr_status xfr_engine_destroy(xfr_engine* e);

/* Upload and pack the parameters (W, W+ = relu(W), transposed/flipped W+ for the backward-data GEMMs,
 * folded eval-mode BatchNorm scale/shift for gamma and relu(gamma)).
 * Stands in for: load_state_dict (resnet.py:277-279) and for the per-forward weight save / relu / restore
 * dance of whitebox.py:309-324,333-346,371-377, which becomes a one-off pack. */
xfr_status xfr_engine_load_weights(xfr_engine* e, const xfr_tensor_view* weights, int32_t n_weights);

/* Device address and size of the packed parameter arena, so that a multi-GPU launcher can broadcast it
 * (rank 0 loads, the others receive: replaces the per-job torch.load of
 * eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:74).  After EVERY write into the arena -- of an
 * engine that never called xfr_engine_load_weights, or through a pointer kept from an earlier call -- call
 * xfr_engine_mark_weights_loaded: it waits for the device and rebuilds what the engine derives from the arena
 * (the bf16 planes of xfr_engine_set_split_gemm); until then the bf16x6 kernel would run on the old weights. */
xfr_status xfr_engine_weight_arena(xfr_engine* e, void** dev_ptr, size_t* bytes);
xfr_status xfr_engine_mark_weights_loaded(xfr_engine* e);

/* Multi-GPU for hosts that are not Python (the Python mirror uses torch.distributed for the same broadcast): one process
 * per GPU, one RCCL communicator, ONE collective -- the broadcast of the packed parameter arena from `root`, which replaces
 * the per-job torch.load of the 298 MB checkpoint in every worker (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:74,
 * 193-216).  The steady state has no exchange: triplets are independent.
 *   xfr_comm_unique_id   rank 0 fills 128 bytes (an ncclUniqueId) and hands them to the other ranks by any out-of-band means
 *   xfr_comm_init        collective over all `world` ranks; `device` is this rank's HIP device
 *   xfr_broadcast_weights  collective; the root must have called xfr_engine_load_weights, the others need not: their engines
 *                        are marked loaded on return.  Synchronises `stream`.
 * librccl is bound on first use; without it these four calls fail with XFR_RCCL_ERROR and nothing else is affected. */
#define XFR_COMM_ID_BYTES 128
typedef struct xfr_comm xfr_comm;
xfr_status xfr_comm_unique_id(void* id_out);
xfr_status xfr_comm_init(int32_t rank, int32_t world, const void* unique_id, int32_t device, xfr_comm** out);
xfr_status xfr_broadcast_weights(xfr_engine* e, xfr_comm* comm, int32_t root, void* stream);
xfr_status xfr_comm_destroy(xfr_comm* comm);

/* Whitebox(..., with_bias, eps, ebp_subtree_mode) -- whitebox.py:262-304. */
xfr_status xfr_engine_set_mode(xfr_engine* e, int32_t subtree_mode, float eps, int32_t with_bias);

/* Number of tensors / shape of a tensor of the program (C,H,W), for wrappers and tests. */
xfr_status xfr_engine_tensor_shape(xfr_engine* e, int32_t tensor_id, int32_t* c, int32_t* h, int32_t* w);

/* Forward only.  Stands in for WhiteboxNetwork.encode / classify (whitebox.py:58-64,98-103,126-133,222-230).
 * x_dev: N x in_c x in_h x in_w (NCHW).  The true values of tensor `tensor_id` are written to out_dev as
 * N x C x H x W (NCHW).  */
xfr_status xfr_forward(xfr_engine* e, const float* x_dev, int32_t n, int32_t tensor_id,
                       float* out_dev, void* stream);

/* Excitation backprop.  Stands in for Whitebox.ebp(x, Pn, mwp=True) (whitebox.py:482-504): the 'activation',
 * 'positive_activation' and 'ebp' forwards (:490-497) and Xn.backward(Pn) with the _backward_ebp tensor hooks
 * (:381-430, :498).
 *   x_dev        N x in_c x in_h x in_w
 *   n_streams    S = 1 (ebp) or 2 (the mate and non-mate sweeps of contrastive_ebp :514,:520 share one forward)
 *   seed_tensor  tensor id at which the gradient seed is injected: the classify() output when the classifier is
 *                hooked (seed = Pn), or the encode() tensor when the classifier is the un-hooked triplet layer
 *                of set_triplet_classifier (whitebox.py:93-96; seed = Pn @ W_cls, computed by the caller)
 *   seed_dev     S x N x D  (D = C*H*W of seed_tensor)
 * Outputs (either may be NULL):
 *   mwp_dev      S x N x C1 x H1 x W1 : P[-2], the MWP at the first conv's output (whitebox.py:499 before pooling)
 *   pooled_dev   S x N x H1 x W1      : sum over channels of P[-2]          (whitebox.py:499)
 */
xfr_status xfr_ebp(xfr_engine* e, const float* x_dev, int32_t n, int32_t n_streams,
                   int32_t seed_tensor, const float* seed_dev,
                   float* mwp_dev, float* pooled_dev, void* stream);

/* Contrastive / truncated contrastive EBP for a batch of independent probes.
 * Stands in for Whitebox.contrastive_ebp (whitebox.py:506-527) when percentile < 0 and
 * Whitebox.truncated_contrastive_ebp (whitebox.py:529-558) when 0 <= percentile <= 100, applied per sample,
 * followed by _mwp_to_saliency (ebp_ver 6 branch, whitebox.py:456-459: gaussian sigma=2 'nearest', clamp, /sum).
 *   seed_dev  2 x N x D : stream 0 = mate (k_poschannel), stream 1 = non-mate (k_negchannel)
 *   sal_dev   N x H1 x W1 saliency maps (each sums to 1) */
xfr_status xfr_contrastive(xfr_engine* e, const float* x_dev, int32_t n,
                           int32_t seed_tensor, const float* seed_dev, float percentile,
                           float* sal_dev, void* stream);

/* As xfr_contrastive, but stops before _mwp_to_saliency: contrast_dev (N x H1 x W1) receives `mwp_contrastive` /
 * `mwp_truncated_contrastive` (whitebox.py:526 / :557), for callers that apply the uint8 + PIL blur of ebp_version != 6
 * (whitebox.py:451-454) on the host. */
xfr_status xfr_contrastive_raw(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                               float percentile, float* contrast_dev, void* stream);

/* One whole triplet step for N independent (mate, non-mate, probe) triplets -- demo/test_whitebox.py:124-133:
 *     x_mate = encode(mate); x_nonmate = encode(nonmate)                         (:71-72)
 *     set_triplet_classifier(scale * x_mate, scale * x_nonmate)                  (:129, scale = 1/2500)
 *     contrastive_ebp(probe, 0, 1)  |  truncated_contrastive_ebp(probe, 0, 1, percentile)   (:130)
 *   probes_dev   N  x in_c x in_h x in_w
 *   gallery_dev  2N x in_c x in_h x in_w : the N mates followed by the N non-mates
 *   encode_tensor  tensor id of encode()'s output; requires max_batch >= 2N
 * The two encode forwards run as one 2N-image batch, concurrently (internal streams, joined before the backward
 * sweep) with the probe forward.  sal_dev: N x H1 x W1. */
xfr_status xfr_triplet_contrastive(xfr_engine* e, const float* probes_dev, const float* gallery_dev, int32_t n,
                                   int32_t encode_tensor, float scale, float percentile, float* sal_dev, void* stream,
                                   int32_t inputs_ready);

/* Cross-call pipelining of xfr_triplet_contrastive (off by default).  When enabled, the engine keeps two forward slots
 * and the forwards of call i+1 start as soon as the slot they overwrite is free, i.e. they overlap the backward
 * sweep of call i (for calls made with inputs_ready = 1); results still appear in order on `stream`.  This is the steady state of the reference's job loop
 * (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:200-215: independent jobs, inputs loaded ahead).
 * enable = 2 additionally pipelines xfr_ebp / xfr_contrastive / xfr_contrastive_raw: their forward overlaps the previous
 * call's sweep, and their x_dev must satisfy the inputs_ready contract of xfr_triplet_contrastive.
 * enable | 4: THREE forward slots instead of two -- the forwards may run two calls ahead of the sweep, which decouples a schedule whose forward is one
 * stream from its sweep (Light-CNN-29v2 EBP, 128 images per call: +1.8 %; level for the three-stream triplet step of the ResNets).
 * Costs one extra copy of the forward workspace per slot beyond the first.  Synchronises the device. */
xfr_status xfr_engine_set_pipeline(xfr_engine* e, int32_t enable);

/* Pipeline level 2 only.  ready = 0 (default): the internal forward stream of xfr_ebp / xfr_contrastive /
 * xfr_contrastive_raw first waits for everything already enqueued on the caller's `stream`, so an x_dev that is still
 * being produced there (a cast, a host-to-device copy) is safe -- at the price of the cross-call overlap.  ready = 1: the
 * caller promises that x_dev of the NEXT run call is already valid on the device and stays untouched until the result
 * has been consumed (the inputs_ready contract of xfr_triplet_contrastive); that call's forward then only waits for the slot
 * it overwrites.  The promise is consumed by the call it covers (one-shot): every later call is back to ready = 0 until the
 * caller renews it, so a stale promise cannot leak into a call that never made one (xfr_ebp_capture, xfr_ebp_store_firing and
 * every other entry point that sweeps go through the same core).  The reference has no counterpart: its inputs are host
 * tensors moved with .to(device) on one stream. */
xfr_status xfr_engine_set_inputs_ready(xfr_engine* e, int32_t ready);

/* Epilogue fusion (default: enable = 3).  Bit 0: the hook chain that follows a backward-data GEMM, BatchNorm / residual add /
 * ReLU after a convolution of a forward-only run (encode, the gallery of a triplet step) and MaxFeatureMap after a Light-CNN
 * convolution execute in the GEMM's epilogue on the LDS-transposed accumulator tile instead of in their own launches -- same
 * arithmetic in the same order, bit-identical maps, about 8 % more triplet maps per second.  Bit 1: also BatchNorm / ReLU after a
 * convolution of the PROBE forward, which additionally keeps the raw output and, where a hook needs it, the positive-pass BatchNorm
 * output (bit-identical; +0.6 % on ResNet-101, +2.2 % on ResNet-50-128d).  Bit 2 (tests): fused chains run through the
 * INTERPRETED epilogue -- the path a network outside the compiled signature table takes; the probe-forward and MaxFeatureMap
 * fusions, which only exist compiled, are then off.  enable = 0 gives every elementwise segment its own kernel again (the GEMM
 * launches then contain convolution work only, which is what one wants when profiling the MFMA kernel by itself).
 * Light-CNN's pooling stages (lightcnn.py:252, MaxPool2d(2)(x) + AvgPool2d(2)(x)) run as one forward kernel (sum, argmax bytes, positive-pass
 * sum) and their two VJPs as the head of the hook chain that follows them (EW_POOL2_IN) whenever bit 0 is set; bit 3 (tests) keeps the
 * separate kernels / launches: bit-identical either way (round 4).  Light-CNN's first layer (one input channel, 5x5, MaxFeatureMap) runs as a
 * direct convolution in the GEMM's K order unless bit 4 (tests) is set.  Backward chain GEMMs that cover the two streams of a contrastive sweep
 * walk their column tiles stream-interleaved (ConvParams::pair_m: the chain's forward-side operands are fetched once for both streams, -13 % GEMM
 * FETCH_SIZE on ResNet-101; which workgroup computes which tile is all that changes) unless bit 5 (A/B measurements) is set.  The block-input
 * gradient of a down-sampling residual block (shortcut AvgPool2d(2) [+ ConcatChannels], main path through a 1x1 / stride 2 convolution:
 * resnet.py:111-149) is built per pixel in the head of the hook chain that consumes it (EW_AVGUP_IN) from the pooled gradient and the
 * compact result of the strided GEMM, instead of by a slice copy, a hook launch, the pool's VJP and a read-modify-write scatter: same
 * operands, same operations, same bits; bit 6 (tests) keeps the separate launches.  Where the shortcut is a projection (resnet50_128.py), the
 * main path's hook chain of that block runs as a side branch of the Add-output GEMM's epilogue (the gradient is saved, the branch stored, the saved
 * value restored for the shortcut's chain) instead of as a launch of its own: same operations on the same operands; bit 7 (tests) keeps the launch.
 * Forward: the reference evaluates a down-sampling block's shortcut behind the main path (resnet.py:144-146); the engine computes it in front of
 * the main path's last convolution, so that the residual add [+ ReLU] joins that convolution's epilogue like in every other block, and keeps the
 * pooled shortcut inside its zero-padded form (a channel prefix is a storage prefix in CNHW: the padding is one fill, no copy).  The order of two
 * independent branches changes no value; bit 8 (tests, A/B) keeps program order and the separate add. */
xfr_status xfr_engine_set_epilogue_fusion(xfr_engine* e, int32_t enable);

/* Share one forward pass between consecutive calls on the same input (off by default).  While hold = 1, a run call
 * (xfr_subtree_weights, xfr_ebp_capture, xfr_layerwise_ebp, xfr_ebp, ...) whose x_dev, n, seed tensor and stream equal those
 * of the previous call skips its forward and sweeps over the activations that are still in the workspace -- the three phases
 * of weighted_subtree_ebp (whitebox.py:647-737) run the same image through the network four times.  The caller promises
 * that the content of x_dev did not change in between.  Loading weights, changing the mode, or any forward on other
 * inputs drops the held state; hold = 0 ends the group. */
xfr_status xfr_engine_hold_forward(xfr_engine* e, int32_t hold);

/* Tail balancing of the convolution GEMMs (on by default).  A GEMM whose tile count is not a multiple of the CU count
 * has its last tiles cut along K so that every CU gets an equal share of the final round (+7 % GEMM rate on
 * ResNet-101 at 32-64 images).  The cut tiles add their K-parts in a fixed order: results are deterministic for a
 * given batch size, but which tiles are cut depends on the batch size, so a sample's map can differ in the last fp32
 * bits between batch sizes / positions.  enable = 0 restores batch-invariant arithmetic (every output element is
 * accumulated in one K order regardless of the batch). */
xfr_status xfr_engine_set_tail_balance(xfr_engine* e, int32_t enable);

/* uint8 entry points (ABI version 4).  The reference's callers hold decoded uint8 H x W x C crops and turn every one into a float tensor on the
 * host (resnet.py:25-37 convert_resnet101v4_image, whitebox.py:235-258 Whitebox_resnet50_128.preprocess, lightcnn.py:19-25
 * prepare_lightCNN_image) before it is copied to the device as fp32.  Here the uint8 images go to the device as they are -- a quarter of the bytes --
 * and the layout kernel in front of the first convolution does that arithmetic, in float64 like numpy does, bit for bit the reference's tensor
 * (tests/test_gpu_entry.py on the four bundled JPEGs): kind SUB_MEAN: out[c] = (float)((double)u8[c] - mean[c]); kind LUMINANCE (3-channel image,
 * 1-channel network): (float)((r / 255) w[0] + (g / 255) w[1] + (b / 255) w[2]).  Resizing / cropping stays with the caller, as in the reference.
 * x_u8_dev: n images, each in_h x in_w x channels, contiguous.  Everything else is xfr_forward / xfr_triplet_contrastive. */
enum { XFR_U8_SUB_MEAN = 0, XFR_U8_LUMINANCE = 1 };
typedef struct xfr_u8_preprocess {
    int32_t kind;       /* XFR_U8_* */
    int32_t channels;   /* of the uint8 images */
    double mean[4];     /* SUB_MEAN: per image channel */
    double weight[4];   /* LUMINANCE: per image channel */
} xfr_u8_preprocess;
xfr_status xfr_engine_set_u8_preprocess(xfr_engine* e, const xfr_u8_preprocess* p);
xfr_status xfr_forward_u8(xfr_engine* e, const uint8_t* x_u8_dev, int32_t n, int32_t tensor_id, float* out_dev, void* stream);
xfr_status xfr_triplet_contrastive_u8(xfr_engine* e, const uint8_t* probes_u8_dev, const uint8_t* gallery_u8_dev, int32_t n, int32_t encode_tensor,
                                      float scale, float percentile, float* sal_dev, void* stream, int32_t inputs_ready);
/* The same call for a caller whose images are fresh every time and live in HOST memory (ABI version 6): n probe and 2n gallery images, uint8 H x W x C,
 * pinned memory preferably.  The engine copies them itself -- its own copy stream, one device staging buffer per forward slot -- and orders the
 * forwards behind that copy, not behind `stream`: with xfr_engine_set_pipeline on, the copy, the preprocessing and the forwards of call i + 1 overlap
 * the sweep of call i, and nothing is promised about residency (xfr_triplet_contrastive_u8 with inputs_ready = 0 orders them behind everything on
 * `stream`, i.e. behind the previous sweep).  The copy is asynchronous: the host buffers must stay unchanged until it has run --
 * xfr_engine_wait_inputs_copied blocks the calling thread until the copies of the LAST such call are done (a caller that alternates two host buffers
 * calls it before refilling the one it used last; the call before that is then done too, copies run in order). */
xfr_status xfr_triplet_contrastive_u8_host(xfr_engine* e, const uint8_t* probes_u8_host, const uint8_t* gallery_u8_host, int32_t n, int32_t encode_tensor,
                                           float scale, float percentile, float* sal_dev, void* stream);
xfr_status xfr_engine_wait_inputs_copied(xfr_engine* e);
/* parity hook: the fp32 network input (n x C x H x W) the uint8 path builds from x_u8_dev */
xfr_status xfr_debug_u8_preprocess(xfr_engine* e, const uint8_t* x_u8_dev, int32_t n, float* out_nchw_dev, void* stream);

/* The lean schedule (on by default; ABI version 4).  A sweep nobody observes (no trace, prior, capture or stored firing; batch % 4 == 0) does not
 * need the literal operands a and x of whitebox.py:388-428 at every hook: the probe forward's W / relu(W) convolution forms the BatchNorm hook's
 * a / (x + eps) in its epilogue (both accumulators in one workgroup) and stores that ONE tensor instead of the two, with the sign bit recording
 * where the ReLU output behind it is zero; hooks whose x is their a (every Conv / Linear / pool / Add hook) become that one-bit gate.  Per hook the
 * result differs from the literal expression by at most one ulp (for a >= 1.7e-9; identical at a = 0): plain-EBP maps agree with the literal path to
 * ~1e-6 of their maximum; a contrastive map is a difference of two nearly equal sweeps and moves by what its conditioning makes of that -- measured up to
 * 1e-3 of the maximum on the ill-conditioned demo triplets (tests/test_gpu_parity.py::test_lean_schedule_equals_literal holds them to 2e-3), inside the
 * stated tolerance against the reference (5e-3 there) but not bit for bit.  enable = 0: the literal operands everywhere
 * (what every observing call -- traces, Whitebox.P[k], layerwise / weighted-subtree EBP -- runs regardless).  Batches that are not a multiple of four
 * always run literal, so a sample's map can differ in its last digits between a batch of 32 and a batch of 1: callers that need batch-invariant
 * arithmetic switch this off together with xfr_engine_set_tail_balance. */
xfr_status xfr_engine_set_lean(xfr_engine* e, int32_t enable);
/* bf16x6 GEMMs (ABI version 5).  The deep-K stride-1 convolutions of 14 x 14 and larger maps (K >= 256 for 1x1, K >= 1152 for KxK, 128 | Cout) can run
 * on the bf16 matrix pipe: every fp32 operand is the exact sum of three bf16 pieces, the six piece products of order <= 2 are exact and are accumulated in
 * fp32 (conv_gemm_split.hip K17).  Since round 6 no sum stays in the matrix pipe for more than three K-steps (48 of the K terms): the partial sums are
 * added to fp32 registers with round-to-nearest adds and alternate in sign, and a launch's error against float64 is BELOW the fp32 MFMA kernels'
 * (rms 5-9e-9 of the sum of magnitudes against 1.1-2.0e-8, no offset: profiles/r6/conv_error_probe.txt).
 * mode 3 (the default since round 6): the forward convolutions AND the sweep's backward-data GEMMs of those layers.  mode 1: the forward convolutions
 * only (the round-5 default).  mode 2: backward only (tuning).  mode 0: fp32 MFMA kernels everywhere.
 * Which launches: a covered layer's launch takes the kernel when its grid has at least 128 tiles of 128 x 128 (half the CUs: about 25 images of a
 * 14 x 14 layer, 6 of a 28 x 28 one); smaller launches -- Whitebox.contrastive_ebp on one image, the weighted-subtree probes -- run the fp32 kernels,
 * which fill the chip with 64 x 64 tiles where this one would leave it idle.  So a sample's map is bit-reproducible for a given batch size and agrees
 * across batch sizes to the kernels' summation-order difference (~1e-6 of its maximum; tests/test_gpu_parity.py::test_split_gemm_equals_fp32_kernels),
 * like K1's tail balancing (xfr_engine_set_tail_balance) and the lean schedule above.  Callers that need one arithmetic for every batch size: mode 0,
 * or mode + 4 (modes 5 .. 7): the covered layers take the kernel whatever the grid (tests and tuning: small launches are slow on it).
 * The weight packs of the covered layers carry bf16 planes (+1.5x their size), built when the weights arrive (xfr_engine_load_weights,
 * xfr_engine_mark_weights_loaded, xfr_broadcast_weights, or this call); every entry point that changes the weights drops the old ones.  After writing
 * through a pointer from xfr_engine_weight_arena, call xfr_engine_mark_weights_loaded again: it rebuilds them.
 * XFR_SPLIT_GEMM=<mode> in the environment sets the mode of new engines. */
xfr_status xfr_engine_set_split_gemm(xfr_engine* e, int32_t mode);
/* Launches of the bf16x6 kernel so far, process-wide. */
xfr_status xfr_engine_split_gemm_stats(xfr_engine* e, int64_t* launches);

/* Convolution launches that took the lean form so far (W and relu(W) accumulated by one workgroup, quotient stored): 0 on an engine whose calls
 * all ran the literal schedule. */
xfr_status xfr_engine_lean_stats(xfr_engine* e, int64_t* dual_launches);

/* xfr_forward on batches of >= 32 images runs as two half batches on the engine's two internal streams and joins them on the caller's stream
 * (on by default; enable = 0: one forward on the caller's stream).  Images are independent; a sample's values can differ in the last fp32 bits
 * from the unsplit run exactly as they do between two batch sizes (tail balancing, see xfr_engine_set_tail_balance).  ABI version 3. */
xfr_status xfr_engine_set_forward_split(xfr_engine* e, int32_t enable);

/* _mwp_to_saliency (whitebox.py:448-460, ebp_ver 6) on N pooled maps: in N x H x W -> out N x H x W. */
xfr_status xfr_mwp_to_saliency(xfr_engine* e, const float* pooled_dev, int32_t n, int32_t h, int32_t w,
                               float* sal_dev, void* stream);

/* ---- "next" row: layerwise_ebp / weighted_subtree_ebp (whitebox.py:561-581, 647-737) ------------------------------------
 * Firing indices are positions in Whitebox.P (reference order); the last one computed here is P[-2]. */

/* Number of hook firings of a sweep seeded at `seed_tensor` (= len(Whitebox.P) - 1: the image hook is not computed). */
xfr_status xfr_firing_count(xfr_engine* e, int32_t seed_tensor, int32_t* n_firings);

/* xfr_op_kind of the hooked module of every firing, reference order: what Whitebox.P_layername (whitebox.py:393) lists,
 * without running anything. */
xfr_status xfr_firing_kinds(xfr_engine* e, int32_t seed_tensor, int32_t* kinds, int32_t capacity);

/* whitebox.py:652-697: true-weight gradients of the two classifier outputs at every hooked module input (the `dA`
 * lists of the 'activation'-mode `_savegrad` hooks, :355-358), reduced per firing k to
 *     w[k] = max((g0[k] >= 0) * (-g1[k])),  idx[k] = argmax (flattened c,h,w index)          (gate_ge0 = 1, :689-690)
 *     w[k] = max((g0[k] <  0) * (-g1[k]))   with g0 = gradient of the cross-entropy loss          (gate_ge0 = 0, :693-694)
 * seed_dev: 2 x N x D gradient seeds at seed_tensor (stream 0 -> g0, stream 1 -> g1 = y[0][1]); w_host / idx_host:
 * n_firings x N host arrays.  Synchronises `stream`. */
xfr_status xfr_subtree_weights(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                               int32_t gate_ge0, float* w_host, int32_t* idx_host, int32_t capacity, void* stream);

/* One standard EBP sweep (whitebox.py:567) over N independent images that returns, for image b and firing k,
 * P[k][b].flatten()[elem[k][b]] (elem < 0: skip) -- the values layerwise_ebp(mode='elementwise') turns into priors
 * (:575-577).  seed_dev: 1 x N x D; elem_host / p_host: n_firings x N host arrays.  Synchronises. */
xfr_status xfr_ebp_capture(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                           const int32_t* elem_host, float* p_host, int32_t n_firings, void* stream);

/* whitebox.py:570-581 for a BATCH of layers of N independent images: sweep j of image b re-runs ebp(img_b, 0*P0) with
 * P_prior[firing[j][b]] set to a tensor that is zero except element elem[j][b] = val[j][b] (mode 'elementwise'), or -- one
 * sweep of one image -- to dense_prior_dev (mode 'argmax').  The N forwards run once; the n_sweeps x N sweeps form the
 * gradient batch (n_sweeps * N <= 2 * max_batch).  firing / elem / val: n_sweeps x N host arrays, firing < 0 marks an idle
 * sweep (its map is zero).  Handing the sweeps over in ascending firing order per image lets sweep j join the backward
 * pass only where its first prior fires.  pooled_dev: n_sweeps x N x H1 x W1 channel-pooled P[-2]. */
xfr_status xfr_layerwise_ebp(xfr_engine* e, const float* x_dev, int32_t n, int32_t n_sweeps, int32_t seed_tensor,
                             const int32_t* firing_host, const int32_t* elem_host, const float* val_host,
                             const float* dense_prior_dev, float* pooled_dev, void* stream);

/* Whitebox.P[firing] of a standard EBP sweep (whitebox.py:394): out_dev receives N x C x H x W; (c,h,w) receive its shape
 * (out_dev == NULL: shape query only, nothing is run).  firing == xfr_firing_count: the hook on the first convolution's INPUT (the
 * reference's P[-1], the MWP at the image): relu(image) * relu(backward-data of that convolution with relu(W)), a gather kernel run on
 * demand -- nothing on the hot path reads it (round 4). */
xfr_status xfr_ebp_store_firing(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                                int32_t firing, float* out_dev, int32_t* c, int32_t* h, int32_t* w, void* stream);

/* Debug / parity: after an xfr_ebp call made while tracing is enabled, the per-firing trace
 * sum(P[i]) (what the golden fixtures store for every entry of Whitebox.P, whitebox.py:394).
 * xfr_engine_set_trace(e, 1) makes xfr_ebp record it (slower).  `sums` receives n_firings x S x N doubles
 * in the reference's firing order; `kinds` (may be NULL) receives the xfr_op_kind of the hooked module. */
xfr_status xfr_engine_set_trace(xfr_engine* e, int32_t enable);
xfr_status xfr_engine_trace_size(xfr_engine* e, int32_t* n_firings);
xfr_status xfr_engine_get_trace(xfr_engine* e, double* sums, int32_t* kinds, int32_t capacity);

/* Test / tuning hook: one forward convolution through the engine's implicit-GEMM kernel, outside any engine.
 * in_dev/out_dev are CNHW device tensors ([C][NB][H][W]); w_host/bias_host are PyTorch-layout host arrays.
 * cfg % 100: 0 lets the launcher pick the tile configuration, 4 / 5 force the 16- / 32-deep 64x64 configuration;
 * (cfg / 10000) % 100: tail balancing 0 = heuristic, 1 = off, S >= 2 = S parts per tail tile; cfg / 1000000 = n in 2..4: the
 * launches go to n streams at once (aggregate rate of co-running launches; *ms_out is then the time per launch).  The kernel is run
 * `reps` times after one untimed launch; *ms_out receives the average duration (HIP events on the null stream). */
xfr_status xfr_debug_conv(const float* in_dev, const float* w_host, const float* bias_host, float* out_dev, int32_t cin,
                          int32_t h, int32_t w, int32_t nb, int32_t cout, int32_t kh, int32_t kw, int32_t stride,
                          int32_t pad, int32_t relu_in, int32_t cfg, int32_t reps, float* ms_out);

/* Tuning hook: while stamps_dev is non-NULL, every GEMM launch of this process records, per wave, 8 64-bit words at
 * stamps_dev[(block * 4 + wave) * 8]: s_memrealtime (100 MHz) at kernel entry / first operands landed / K loop done / epilogue entered / exit,
 * then HW_ID, XCC_ID, and the wave's life in shader-clock cycles (s_memtime; with the 100 MHz stamps: the effective clock) (tools/conv_sweep.py --stamps draws a launch's timeline from them).  The buffer holds
 * 32 words for each of capacity_workgroups workgroups; workgroups beyond that do not record.  NULL switches it off (the default).
 * capacity_workgroups < 0: sampled mode for whole steps (tools/phase_probe.py) -- the buffer holds -capacity_workgroups records in regions
 * of 256; the n-th launch writes up to 256 evenly spaced workgroups into region n % regions, word 6 also carries K, M and Cout of the launch. */
xfr_status xfr_debug_conv_stamps(void* stamps_dev, int32_t capacity_workgroups);

/* Tuning hook: a timeline of the GEMM launches as the device ran them, streams overlapped.  While log_dev is non-NULL (zero-filled
 * device memory, 64 bytes per launch, `capacity` launches), every GEMM launch of this process records when its first workgroup
 * started and its last one ended (s_memrealtime, 10 ns ticks) and the library notes its shape, stream and tile configuration.  A
 * call with dump_path != NULL first writes what has been recorded so far as CSV (synchronise the device before); log_dev = NULL
 * stops recording.  tools/gemm_timeline.py reads the file.
 * xfr_debug_conv_stamps and xfr_debug_conv_log are PROCESS-GLOBAL (every engine of the process records into them) and mutex-guarded: engines
 * may launch from other host threads while they are set, dumped or cleared; launches only touch the lock while a hook is on.  A dump after
 * log_dev = NULL writes nothing and returns XFR_OK. */
xfr_status xfr_debug_conv_log(void* log_dev, int32_t capacity, const char* dump_path);

/* Bytes of device memory held by the engine (weights + workspace). */
xfr_status xfr_engine_memory(xfr_engine* e, size_t* weight_bytes, size_t* workspace_bytes);

/* Duration in ms of the conv/linear GEMM launches of the most recent run call (sum of HIP-event timings on the
 * run's stream) and their count and algorithmic FLOPs; enabled by xfr_engine_set_profile(e, 1).  Used by
 * bench.py for the live roofline figure. */
xfr_status xfr_engine_set_profile(xfr_engine* e, int32_t enable);
xfr_status xfr_engine_get_profile(xfr_engine* e, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);
/* The same three figures split by the kernel family that really ran each launch (ABI version 6): index 0 the fp32 MFMA kernels, index 1 the bf16x6
 * kernel -- each family has its own ceiling (157.3 / 419.4 fp32-equivalent TFLOP/s), and bench.py reports a roofline fraction per family.  Arrays of 2. */
xfr_status xfr_engine_get_profile_by_kernel(xfr_engine* e, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);
/* While profiling is on, also append one CSV record per GEMM launch to `path` (NULL: stop):
 * Cout,nhalves,K,M,kh,stride,out_stride,relu_in,accumulate,ms,TFLOP/s,cfg  (profiles/layer_table.py reads it; cfg = the configuration that ran, 9 = bf16x6). */
xfr_status xfr_engine_profile_csv(xfr_engine* e, const char* path);

/* Process-wide counts of GEMM launches that carried a fused elementwise chain: those whose epilogue was one of the
 * compile-time specialised ones (xfr_amd/csrc/chain_sigs.inc) and those that fell back to the interpreter; n_signatures =
 * size of the compiled table.  bench.py snapshots the counters right after its timed loop and fails `outputs_ok` if an interpreted
 * launch ran there; tests/test_gpu_entry.py asserts interpreted == 0 for the benchmarked entry point on the BASELINE backbones. */
xfr_status xfr_chain_epilogue_stats(int64_t* compiled_launches, int64_t* interpreted_launches, int32_t* n_signatures);

/* The planner without a device: the fused forward-only and backward schedules of a layer program for one subtree mode and
 * seed tensor, as text (one launch per line; GEMM lines carry the signature of their fused chain and the index of its
 * compiled epilogue, -1 if it would be interpreted).  Makes no HIP call, so it also runs where no GPU is visible:
 * tools/gen_chain_sigs.py builds chain_sigs.inc from it and the CPU test-suite checks the table against it.
 * `needed` (may be NULL) receives the full length including the terminator; `buf` gets at most `capacity` bytes. */
xfr_status xfr_plan_describe(const xfr_op_desc* ops, int32_t n_ops, int32_t n_weights, int32_t in_c, int32_t in_h, int32_t in_w,
                             int32_t batch, int32_t subtree_mode, int32_t seed_tensor, char* buf, size_t capacity, size_t* needed);

#ifdef __cplusplus
}
#endif
#endif /* XFR_AMD_H */
