// engine.hip -- the C ABI (include/xfr_amd.h) and the layer-program executor of the EBP engine.
//
// What the reference does with forward hooks, pre-forward hooks, tensor hooks and a freshly recorded autograd
// graph on every call (whitebox.py:306-437, :482-504) is done here once, at engine creation:
//   * shape inference over the static layer program;
//   * the hook table: which (module call, input) hooks sit on which tensor, in registration order, with the
//     in-place-ReLU placement and the late-binding (a, x) of two-input Add modules;
//   * a static analysis of the 'positive_activation' pass (whitebox.py:315-330): for every tensor whether its
//     positive-pass value equals the true value (EQ), equals relu(true value) (RELU) or has to be computed
//     (OTHER), so that X is only materialised where it differs from A;
//   * the backward schedule: GEMMs for conv/linear VJPs with relu(W), and every elementwise step between two
//     GEMMs (tensor hooks, ReLU masks, BatchNorm / Multiply VJPs) fused into one EwChain launch.
#include "../../include/xfr_amd.h"
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <dlfcn.h>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

xfr_status fail(xfr_status st, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(_e == hipErrorOutOfMemory ? XFR_OOM : XFR_HIP_ERROR, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                       \
    } while (0)

enum PState { PS_EQ = 0, PS_RELU = 1, PS_OTHER = 2 };

struct Hook {
    int op;        // hooked module call
    int j;         // which input of that call
    int a_tensor;  // tensor providing a (and x): the LAST input of the call (whitebox.py:379-381 late binding)
};

struct Tensor {
    int C = 0, H = 0, W = 0;
    int producer = -1;
    std::vector<int> consumers;
    bool nonneg = false;
    int pstate = PS_OTHER;
    int alias = -1;          // shares T storage with this tensor (in-place ReLU, Split)
    int prefix_of = -1;      // T storage = the leading channels of this tensor's (the pooled shortcut inside its zero-padded form; layout_workspace)
    size_t t_off = 0, pv_off = 0, g_off = 0;   // offsets (floats) into the workspace
    bool need_pv = false;
    std::vector<Hook> hooks;
    long per_n() const { return (long)C * H * W; }
    int HW() const { return H * W; }
};

struct OpRec {
    xfr_op_desc d;
    // packed parameter offsets (floats) into the arena; -1 if absent
    long w_true = -1, w_pos = -1, w_bwd = -1, w_bwd_true = -1;   // w_bwd_true: true-weight backward pack (plain gradients)
    long b_true = -1, b_pos = -1;            // conv/linear bias and relu(bias)
    long bn_alpha_t = -1, bn_beta_t = -1, bn_alpha_p = -1, bn_beta_p = -1, bn_beta_pb = -1;
    int ldw = 0, ldb = 0;
    bool tap_fwd = false, tap_bwd = false;   // K packed tap-major (kh,kw,ci) for the forward / backward-data GEMM
    bool tap4_fwd = false;                   // image stems (Cin 3 or 4): K packed (kh,kw,4 channel slots); Kf = rows of the forward pack
    int Kf = 0;
    int Cin = 0, K = 0, Kb = 0;
    size_t idx_off = 0;                      // maxpool argmax (bytes into idx workspace)
    size_t norm_off = 0;                     // normalize: norms (floats into misc workspace)
    int pair = 0;                            // MaxFeatureMap convolution (Conv -> Split -> max of halves, lightcnn.py:48-62): Co = cout / 2; its forward
                                             // pack (and bias) holds the output channels interleaved (column 2c = channel c, 2c+1 = channel c + Co)
    int pair_split = -1, pair_max = -1;      // the Split and G_MAXHALVES ops behind it
    bool fuse_relu = false;                  // forward: the following in-place ReLU is applied in this op's kernel
    bool relu_fused_away = false;            // forward: this ReLU is executed by its producer
};

enum StepKind { ST_EW, ST_CONV_BWD, ST_MAXPOOL_BWD, ST_AVGPOOL_BWD, ST_COPY, ST_MAXHALVES_BWD, ST_NORMALIZE_BWD, ST_ZERO };

struct HookRef { int tensor; int hook; int slot; };

struct BwdStep {
    int kind;
    int op = -1;
    int src_t = -1, dst_t = -1;
    int accumulate = 0;
    long copy_elems_per_sb = 0;   // ST_COPY: channels to copy (prefix), dst/src channel counts differ for concat
    // ST_EW: symbolic chain (resolved to pointers at run time)
    struct Sym { int type; int action; int t0; int x_t; float f; int op; int slot; bool tap; };
    std::vector<Sym> chain;   // ST_EW: the chain; ST_CONV_BWD: epilogue chain fused into the GEMM (may be empty)
    int ew_t = -1;     // tensor whose shape the chain runs over
    bool compact = false;   // ST_CONV_BWD of a 1x1 / stride 2 convolution: the result stays on the sampled grid, dense, at the start of dst_t's gradient
                            // region (the chain head EW_AVGUP_IN of the launch that follows puts it in place)
};

struct BwdPlan {
    int seed_tensor = -1;
    int mode = -1;
    bool plain = false;              // true-weight gradients without hooks (whitebox.py:652-676 dA lists)
    std::vector<int> firing_tensor;  // tensor whose gradient each firing sees
    std::vector<BwdStep> steps;      // one launch per step, no cross-kernel fusion (used when tracing)
    std::vector<BwdStep> fused;      // after copy forwarding and chain -> chain merging
    std::vector<BwdStep> fused_gemm; // ... and with the chains that follow a backward GEMM run in its epilogue
    std::vector<BwdStep> fused_gemm_nofan; // the same without the MaxFeatureMap fan-out (a compiled-only epilogue step): what the interpreted epilogues run
    std::vector<int> firing_kinds;   // xfr_op_kind per firing, reference order
    std::vector<int> firing_ops;     // hooked module call (op index) per firing: Whitebox.P_layername is str(module) of these (whitebox.py:393)
    int n_firings = 0;
    int fan_ok = -1;                 // does every fan-out epilogue of fused_gemm have a compiled signature (-1: not checked yet; fanout_compiled)
    // The lean schedule (xfr_engine_set_lean): the probe forward stores quotients instead of hook operands, the sweep reads them.
    int lean_state = -1;             // -1 not prepared, 0 does not apply to this plan, 1 ready (lean_prepare)
    std::vector<char> lean_q;        // per tensor: 1 = its T storage holds a / (x + eps) of the BatchNorm hook on it (sign bit: lean_final <= 0),
                                     // 2 = its Pv storage holds a / (x + eps) of the in-place ReLU hook behind it
    std::vector<int> lean_final;     // per tensor with lean_q 1: root of the tensor whose positivity the sign bit records (-1: none)
    std::vector<BwdStep> fused_gemm_lean;
};

}  // namespace

struct xfr_engine {
    int device = 0;
    int max_batch = 0;
    int in_c = 0, in_h = 0, in_w = 0;
    std::vector<OpRec> ops;
    std::vector<Tensor> tens;
    int n_weights = 0;
    // parameters
    float* arena = nullptr;
    size_t arena_floats = 0;
    bool weights_loaded = false;
    // workspace
    float* ws = nullptr;
    size_t ws_floats = 0;
    uint8_t* idx_ws = nullptr;
    size_t idx_bytes = 0;
    size_t x_off = 0, seed_off = 0, tap_off = 0, pooled_off = 0, blur_a_off = 0, blur_b_off = 0, misc_off = 0, thr_off = 0;
    double* dbl_ws = nullptr;      // sums [2*maxB] + trace
    size_t trace_cap = 0;          // firings capacity
    void* trunc_ws = nullptr;
    // mode
    int mode = XFR_MODE_AFFINEONLY_WITH_PRIOR;
    float eps = 1e-16f;
    int with_bias = 0;
    bool need_dirty = true;
    // plans
    std::deque<BwdPlan> plans;         // deque: get_plan() hands out pointers that must survive later insertions
    // trace / profile
    int trace_on = 0;
    // per-call context of the backward sweep ("next" rows: layerwise / weighted-subtree EBP)
    // xfr_engine_hold_forward: the forward state of (held_x, held_B, held_last) is still in slot 0
    bool hold_forward = false, held_pos = false;
    const float* held_x = nullptr;
    int held_B = 0, held_last = -1;
    hipStream_t held_stream = nullptr;
    bool lazy_zero = false;                           // prefix sweeps: run_backward zeroes un-written gradient rows on demand (xfr_layerwise_ebp)
    std::vector<int> rc_active;                       // layerwise sweeps in ascending firing order: stream j (all its samples) is identically zero before firing rc_active[j]
    int rc_n = 1;                                     // samples per stream of the current layerwise batch
    size_t g_begin = 0, g_end = 0;                    // the gradient region of the workspace (floats)
    // priors / captures of the current sweep: tables [n_firings][tab_sb] over the gradient rows sb (stream * n + sample),
    // staged in pinned host memory and copied once per call (common.h: EwStep::prior_elem / cap_elem)
    bool rc_priors = false, rc_caps = false;
    int tab_sb = 0;                                   // row length of the tables of the current call
    std::vector<char> rc_prior_row, rc_cap_row;       // per firing: does the row hold any entry?
    int rc_dense_slot = -1;                           // firing that carries the dense prior (-1: none)
    const float* rc_prior_dense = nullptr;            // dense prior tensor (single sweep of one image)
    int *tab_elem_h = nullptr, *tab_elem_d = nullptr; // prior element (or capture element) per (firing, row); -1: none
    float *tab_val_h = nullptr, *tab_val_d = nullptr; // prior value per (firing, row)
    size_t tab_cap = 0;                               // entries allocated
    hipEvent_t ev_tab = nullptr;                      // the last host-to-device table copy
    float* cap_dev = nullptr;                         // [n_firings][tab_sb] captured values
    float* stat_v = nullptr;                          // [n_firings][max_batch]
    int* stat_i = nullptr;
    void* stat_scratch = nullptr;
    StatDesc* stat_desc = nullptr;                    // [n_firings] tensor descriptors + firing -> tensor map of the last plan used
    int* stat_f2u = nullptr;
    const void* stat_plan = nullptr;
    int stat_nu = 0;
    int store_slot = -1;                              // firing whose full P tensor is kept in store_dev
    float* store_dev = nullptr;
    int store_tensor = -1, store_sb = 0;
    std::vector<char> is_hook_a;       // per tensor: some hook takes its a (and x) from this tensor's forward values
    std::vector<char> fwd_done;        // per forward pass: ops whose work was folded into an earlier GEMM epilogue
    std::vector<char> pos_done;        // ... and whose positive-pass output was produced there too
    int fwd_last_op = 0;
    // Measured on MI355X (bench.py, B=32): folding BatchNorm/ReLU(/residual) into the forward GEMM epilogue is 2-4 % SLOWER
    // than separate streaming kernels -- the HBM-bound elementwise kernels of one stream overlap the MFMA-bound GEMMs of
    // the other for free, while extra epilogue stores stretch every workgroup of a lock-step grid.  Kept for experiments.
    // tail-balancing scratch (conv_gemm.hip), one per stream that launches GEMMs; kernels on one stream serialise,
    // so consecutive launches share it
    struct TailWs { hipStream_t s; float* ws; unsigned* cnt; };
    TailWs tail_ws[8];
    int n_tail_ws = 0;
    bool tail_balance = true;          // xfr_engine_set_tail_balance
    bool split_forward = true;         // xfr_engine_set_forward_split: forward-only batches of >= 32 images as two halves on the internal streams
    bool interpret_chains = false;     // xfr_engine_set_epilogue_fusion bit 2: fused chains run through the interpreted epilogue (tests)
    bool planning_only = false;        // xfr_plan_describe: list what the planner WOULD fuse, whatever the signature table holds
    bool fuse_probe_fwd = true;        // probe forward (with the positive pass): BatchNorm / ReLU in the (dual) GEMM's epilogue (STORE raw, [FORK positive
                                       // BatchNorm], affine, clamp).  Round 3, MI355X: +0.6 % maps/s on ResNet-101, +2.2 % on ResNet-50-128d, bit-identical
    bool fuse_fwd_only = true;         // forward-only runs: BatchNorm / residual add / ReLU in the GEMM epilogue
    bool hoist_shortcut = true;        // down-sampling blocks: the shortcut (average pool, channel padding) is computed BEFORE the main path's last
                                       // convolution, so that the residual add joins its epilogue like in every other block (bit 8 of the fusion mask)
    bool direct_stem = true;           // Light-CNN's 1-channel 5x5 first layer as a direct convolution (xfr_engine_set_epilogue_fusion bit 4 clear; tests set it)
    bool fuse_avgup = true;            // down-sampling blocks: slice copy + pooled hook + average-pool VJP + strided GEMM's read-modify-write as the head of the
                                       // hook chain that follows (EW_AVGUP_IN; xfr_engine_set_epilogue_fusion bit 6 clear)
    bool fuse_branch = true;           // projection-shortcut blocks: the main path's hook chain as a side branch of the Add-output GEMM's epilogue (EW_STORE actions
                                       // 1 / 2; xfr_engine_set_epilogue_fusion bit 7 clear)
    // uint8 entry points (xfr_forward_u8 / xfr_triplet_contrastive_u8): the image pointer handed to the forward is uint8 H x W x C and the layout
    // kernel in front of the first convolution does the reference's preprocessing arithmetic (xfr_engine_set_u8_preprocess)
    bool u8_on = false;
    bool u8_set = false;
    U8Pre u8_pre;
    bool split_any_grid = false;       // xfr_engine_set_split_gemm mode + 4: covered layers take the bf16x6 kernel whatever the launch's grid (tests, tuning)
    int split_mask = 3;                // xfr_engine_set_split_gemm: which covered layers run the bf16x6 kernel (conv_gemm_split.hip K17) -- bit 0 the forward
                                       // convolutions, bit 1 the sweep's backward-data GEMMs; both by default since round 6 (short in-pipe sums)
    bool lean = true;                  // xfr_engine_set_lean: plain sweeps (no trace / prior / capture / stored firing, batch % 4 == 0) take the lean schedule
    const BwdPlan* lean_cur = nullptr; // the plan whose lean tables the running probe forward / sweep follow (null: literal)
    bool lean_decide = false;          // lean_prepare's dry run of the probe forward: decide per convolution, record in lean_q_run / lean_final_run
    bool dry_run = false;              // ... which launches nothing
    bool lean_missing_sig = false;     // ... and found a lean epilogue without a compiled signature
    long lean_launches = 0;            // dual-accumulator launches so far (xfr_engine_lean_stats)
    std::vector<char> lean_q_run;
    std::vector<int> lean_final_run;
    bool pair_tiles = true;            // backward chain GEMMs over two streams walk their m-tiles stream-interleaved (xfr_engine_set_epilogue_fusion bit 5 clear)
    bool fuse_pools = true;            // Light-CNN's maxpool + avgpool pair: one forward kernel (xfr_engine_set_epilogue_fusion bit 0 switches it with the rest)
    bool fuse_gemm_epilogue = true;    // hook chains that follow a backward GEMM run in its (vector) epilogue
                                       // (both: xfr_engine_set_epilogue_fusion; DESIGN.md section 6 has the measurements)
    int last_trace_firings = 0, last_trace_sb = 0;
    std::vector<int> last_trace_kinds;
    int profile_on = 0;
    std::string profile_csv;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<ConvParams> ev_params;
    std::vector<int> ev_cfg;           // the configuration each profiled launch really ran
    double fam_ms[2] = {0.0, 0.0}, fam_flops[2] = {0.0, 0.0};      // last profiled run, by kernel family: [0] fp32 MFMA, [1] bf16x6
    long fam_launches[2] = {0, 0};
    size_t ev_used = 0;
    double prof_flops = 0.0;
    double last_gemm_ms = 0.0;
    long last_gemm_launches = 0;
    double last_gemm_flops = 0.0;

    float* t_bank = nullptr;       // when set, true activations live in this bank (gallery forward of a triplet step)
    float* ws_enc = nullptr;       // second bank of true activations (T region only), allocated on first use
    size_t t_region_floats = 0;
    hipStream_t s_a = nullptr, s_b = nullptr;
    // xfr_triplet_contrastive_u8_host: the engine's own copy stream and one uint8 staging buffer per forward slot -- fresh inputs keep the cross-call overlap
    hipStream_t s_copy = nullptr;
    uint8_t* u8_stage[3] = {nullptr, nullptr, nullptr};
    size_t u8_stage_bytes = 0;
    hipEvent_t ev_copied[3] = {nullptr, nullptr, nullptr}, ev_stage_a[3] = {nullptr, nullptr, nullptr}, ev_stage_b[3] = {nullptr, nullptr, nullptr};
    bool stage_busy[3] = {false, false, false};
    hipEvent_t inputs_event = nullptr;   // one-shot, set by the _host entry point: the inputs of THIS call are complete when it fires (instead of the caller's stream order)
    int stage_slot = -1;                 // ... and the staging slot its forwards read
    hipEvent_t last_copied = nullptr;    // xfr_engine_wait_inputs_copied
    hipEvent_t ev_fork = nullptr, ev_a = nullptr, ev_b = nullptr;
    // cross-step pipelining (xfr_engine_set_pipeline): two forward slots (T, Pv, norms, argmax) so that the forward of
    // triplet call i+1 may run while the backward sweep of call i still reads slot i%2
    bool pipeline = false;
    bool pipeline_all = false;     // level 2: xfr_ebp / xfr_contrastive calls are pipelined too
    bool inputs_ready = false;     // xfr_engine_set_inputs_ready: the NEXT level-2 call may read x_dev without waiting for the caller's stream (one-shot)
    int cur_slot = 0;
    long seq = 0;
    float* ws2 = nullptr, *ws3 = nullptr;       // forward workspaces of pipeline slots 1 and 2
    uint8_t* idx_ws2 = nullptr, *idx_ws3 = nullptr;
    int n_slots = 2;               // xfr_engine_set_pipeline bit 2: three forward slots (the forwards may run two calls ahead of the sweep)
    size_t fwd_region_floats = 0;
    float* seedbuf[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_slot_done[3] = {nullptr, nullptr, nullptr};
    bool slot_pending[3] = {false, false, false};
    float* fwd_base() { return cur_slot == 0 ? ws : (cur_slot == 1 ? ws2 : ws3); }
    uint8_t* idx_base() { return cur_slot == 0 ? idx_ws : (cur_slot == 1 ? idx_ws2 : idx_ws3); }
    float* T(int t) { const Tensor& x = tens[t]; return (t_bank ? t_bank : fwd_base()) + tens[x.alias >= 0 ? root(t) : t].t_off; }
    float* Pv(int t) { return fwd_base() + tens[t].pv_off; }
    float* misc() { return fwd_base() + misc_off; }
    float* G(int t) { return ws + tens[t].g_off; }
    int root(int t) const { while (tens[t].alias >= 0) t = tens[t].alias; return t; }
};

namespace {

bool is_hooked(int kind) { return kind >= XFR_OP_CONV && kind <= XFR_OP_SPLIT; }
bool is_affine_name(int kind)
{   // whitebox.py:399/:409: 'Conv' | 'Linear' | 'AvgPool' | 'BatchNorm' in str(module)
    return kind == XFR_OP_CONV || kind == XFR_OP_LINEAR || kind == XFR_OP_AVGPOOL || kind == XFR_OP_BATCHNORM;
}

int hook_action(int mode, int kind)
{
    switch (mode) {
        case XFR_MODE_AFFINEONLY: return is_affine_name(kind) ? HOOK_DIV : HOOK_PASS;
        case XFR_MODE_AFFINEONLY_WITH_PRIOR: return is_affine_name(kind) ? HOOK_DIV : HOOK_RELU;
        default: return HOOK_DIV;   // 'norelu' without priors and 'all' (whitebox.py:416-428)
    }
}

int pool_out(int in, int k, int s, int p, bool ceil_mode)
{
    int num = in + 2 * p - k;
    int o = (ceil_mode ? (num + s - 1) / s : num / s) + 1;
    if (ceil_mode && (o - 1) * s >= in + p) --o;   // last window must start inside the (left-padded) input
    return o;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------------------------
xfr_status build(xfr_engine* e, const xfr_op_desc* ops, int n_ops)
{
    e->tens.resize(n_ops + 1);
    e->ops.resize(n_ops);
    Tensor& in = e->tens[0];
    in.C = e->in_c; in.H = e->in_h; in.W = e->in_w;
    in.pstate = PS_EQ;
    for (int k = 0; k < n_ops; ++k) {
        OpRec& o = e->ops[k];
        o.d = ops[k];
        const xfr_op_desc& d = o.d;
        if (d.out != k + 1) return fail(XFR_INVALID_ARG, "op %d: out tensor id must be %d (got %d)", k, k + 1, d.out);
        if (d.in0 < 0 || d.in0 > k) return fail(XFR_INVALID_ARG, "op %d: bad in0 %d", k, d.in0);
        const bool two = (d.kind == XFR_OP_ADD || d.kind == XFR_OP_G_ADD);
        if (two && (d.in1 < 0 || d.in1 > k)) return fail(XFR_INVALID_ARG, "op %d: bad in1 %d", k, d.in1);
        auto chkw = [&](int w) { return w >= -1 && w < e->n_weights; };
        if (!chkw(d.w_weight) || !chkw(d.w_bias) || !chkw(d.w_mean) || !chkw(d.w_var))
            return fail(XFR_INVALID_ARG, "op %d: weight index out of range", k);
        const Tensor& a = e->tens[d.in0];
        Tensor& t = e->tens[d.out];
        t.producer = k;
        e->tens[d.in0].consumers.push_back(k);
        if (two) e->tens[d.in1].consumers.push_back(k);
        switch (d.kind) {
            case XFR_OP_CONV:
            case XFR_OP_LINEAR: {
                if (d.cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride <= 0 || d.pad < 0 || d.w_weight < 0)
                    return fail(XFR_INVALID_ARG, "op %d: bad conv/linear geometry", k);
                if (d.kind == XFR_OP_LINEAR && (d.kh != a.H || d.kw != a.W || d.pad != 0))
                    return fail(XFR_INVALID_ARG, "op %d: linear kernel must equal the input extent %dx%d", k, a.H, a.W);
                t.C = d.cout;
                t.H = (a.H + 2 * d.pad - d.kh) / d.stride + 1;
                t.W = (a.W + 2 * d.pad - d.kw) / d.stride + 1;
                if (t.H <= 0 || t.W <= 0) return fail(XFR_INVALID_ARG, "op %d: empty conv output", k);
                if (d.stride > 1 && !(d.kh == 1 && d.kw == 1) && k != 0)
                    return fail(XFR_UNSUPPORTED_LAYER, "op %d: strided %dx%d convolution is only supported as the first layer "
                                "(its backward-data pass is not needed for P[-2])", k, d.kh, d.kw);
                o.Cin = a.C; o.K = a.C * d.kh * d.kw; o.Kb = d.cout * d.kh * d.kw;
                t.nonneg = false; t.pstate = PS_OTHER;
                break;
            }
            case XFR_OP_BATCHNORM:
                if (d.w_weight < 0 || d.w_bias < 0 || d.w_mean < 0 || d.w_var < 0)
                    return fail(XFR_INVALID_ARG, "op %d: batchnorm needs weight, bias, running_mean, running_var", k);
                t.C = a.C; t.H = a.H; t.W = a.W; t.nonneg = false; t.pstate = PS_OTHER;
                break;
            case XFR_OP_RELU:
                t.C = a.C; t.H = a.H; t.W = a.W; t.nonneg = true; t.pstate = PS_EQ;
                if (d.inplace) {
                    if (e->tens[d.in0].consumers.size() != 1)
                        return fail(XFR_UNSUPPORTED_LAYER, "op %d: in-place ReLU on a tensor with other consumers", k);
                    t.alias = d.in0;
                }
                break;
            case XFR_OP_MAXPOOL:
                if (d.kh != d.kw || d.kh <= 0 || d.kh > 15 || d.stride <= 0) return fail(XFR_INVALID_ARG, "op %d: bad maxpool", k);
                t.C = a.C; t.H = pool_out(a.H, d.kh, d.stride, d.pad, d.ceil_mode != 0);
                t.W = pool_out(a.W, d.kw, d.stride, d.pad, d.ceil_mode != 0);
                t.nonneg = a.nonneg; t.pstate = a.nonneg ? PS_EQ : PS_RELU;
                break;
            case XFR_OP_AVGPOOL:
                if (d.kh != d.kw || d.kh <= 0 || d.stride <= 0 || d.pad != 0) return fail(XFR_INVALID_ARG, "op %d: bad avgpool", k);
                t.C = a.C; t.H = (a.H - d.kh) / d.stride + 1; t.W = (a.W - d.kw) / d.stride + 1;
                t.nonneg = a.nonneg; t.pstate = a.nonneg ? PS_EQ : PS_OTHER;
                if (d.kh == 1 && d.stride == 1) t.alias = d.in0;       // AvgPool2d(1, 1) (resnet.py:210): the identity -- same storage, no launch
                break;
            case XFR_OP_ADD:
            case XFR_OP_G_ADD: {
                const Tensor& b = e->tens[d.in1];
                if (a.C != b.C || a.H != b.H || a.W != b.W) return fail(XFR_INVALID_ARG, "op %d: add shape mismatch", k);
                t.C = a.C; t.H = a.H; t.W = a.W; t.nonneg = a.nonneg && b.nonneg;
                if (d.kind == XFR_OP_ADD) t.pstate = (a.nonneg && b.nonneg) ? PS_EQ : PS_OTHER;
                else t.pstate = (a.pstate == PS_EQ && b.pstate == PS_EQ) ? PS_EQ : PS_OTHER;
                break;
            }
            case XFR_OP_CONCAT:
                if (d.cout < 0) return fail(XFR_INVALID_ARG, "op %d: bad concat", k);
                t.C = a.C * (1 + d.cout); t.H = a.H; t.W = a.W; t.nonneg = a.nonneg; t.pstate = a.nonneg ? PS_EQ : PS_RELU;
                break;
            case XFR_OP_MULTIPLY:
                if (!(d.fparam > 0.f)) return fail(XFR_UNSUPPORTED_LAYER, "op %d: Multiply(n) needs n > 0", k);
                t.C = a.C; t.H = a.H; t.W = a.W; t.nonneg = a.nonneg; t.pstate = a.nonneg ? PS_EQ : PS_RELU;
                break;
            case XFR_OP_SPLIT:
                t.C = a.C; t.H = a.H; t.W = a.W; t.nonneg = a.nonneg; t.pstate = a.nonneg ? PS_EQ : PS_RELU;
                t.alias = d.in0;
                break;
            case XFR_OP_G_MAXHALVES:
                if (a.C % 2) return fail(XFR_INVALID_ARG, "op %d: max-of-halves needs an even channel count", k);
                t.C = a.C / 2; t.H = a.H; t.W = a.W; t.nonneg = a.nonneg; t.pstate = a.pstate;
                break;
            case XFR_OP_G_NORMALIZE:
                if (a.H != 1 || a.W != 1) return fail(XFR_UNSUPPORTED_LAYER, "op %d: normalize is only supported on N x C vectors", k);
                t.C = a.C; t.H = 1; t.W = 1; t.nonneg = false; t.pstate = (a.pstate == PS_EQ) ? PS_EQ : PS_OTHER;
                break;
            default:
                return fail(XFR_UNSUPPORTED_LAYER, "op %d: unsupported layer kind %d (Sigmoid/ELU/Tanh and friends are not "
                            "supported, see whitebox.py:403)", k, d.kind);
        }
        if (t.nonneg && t.pstate == PS_RELU) t.pstate = PS_EQ;
    }
    // an in-place ReLU overwrites its input: nothing else may read that tensor, before or after the ReLU in call order
    for (int k = 0; k < n_ops; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (d.kind == XFR_OP_RELU && d.inplace && e->tens[d.in0].consumers.size() != 1)
            return fail(XFR_UNSUPPORTED_LAYER, "op %d: in-place ReLU on tensor %d, which op %d also reads", k, d.in0,
                        e->tens[d.in0].consumers[e->tens[d.in0].consumers[0] == k ? 1 : 0]);
    }
    // hook table (registration order == call order)
    for (int k = 0; k < n_ops; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (!is_hooked(d.kind)) continue;
        const int nin = (d.kind == XFR_OP_ADD) ? 2 : 1;
        const int last_in = (nin == 2) ? d.in1 : d.in0;
        for (int j = 0; j < nin; ++j) {
            const int tin = (j == 0) ? d.in0 : d.in1;
            const int ht = (d.kind == XFR_OP_RELU && d.inplace) ? d.out : tin;
            Hook h; h.op = k; h.j = j; h.a_tensor = (d.kind == XFR_OP_RELU && d.inplace) ? d.out : last_in;
            e->tens[ht].hooks.push_back(h);
        }
    }
    e->is_hook_a.assign(e->tens.size(), 0);
    for (auto& x : e->tens)
        for (const Hook& h : x.hooks) e->is_hook_a[h.a_tensor] = 1;
    // forward fusion: <BatchNorm | Add | functional add> followed by an in-place ReLU on its output
    for (int k = 0; k + 1 < n_ops; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        const xfr_op_desc& nx = e->ops[k + 1].d;
        if ((d.kind == XFR_OP_BATCHNORM || d.kind == XFR_OP_ADD || d.kind == XFR_OP_G_ADD) && nx.kind == XFR_OP_RELU &&
            nx.inplace && nx.in0 == d.out) {
            e->ops[k].fuse_relu = true;
            e->ops[k + 1].relu_fused_away = true;
        }
    }
    // MaxFeatureMap: Conv -> Split -> torch.max(halves) with single consumers all the way
    for (int k = 0; k + 2 < n_ops; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (d.kind != XFR_OP_CONV || (d.cout & 1) || e->tens[d.out].consumers.size() != 1) continue;
        const int k1 = e->tens[d.out].consumers[0];
        if (e->ops[k1].d.kind != XFR_OP_SPLIT || e->tens[e->ops[k1].d.out].consumers.size() != 1) continue;
        const int k2 = e->tens[e->ops[k1].d.out].consumers[0];
        if (e->ops[k2].d.kind != XFR_OP_G_MAXHALVES) continue;
        e->ops[k].pair = d.cout / 2;
        e->ops[k].pair_split = k1;
        e->ops[k].pair_max = k2;
    }
    return XFR_OK;
}

// x source of a hook / positive-pass value of a tensor, as (pointer, relu-on-load)
struct Src { const float* p; int relu; };

Src pv_src(xfr_engine* e, int t)
{
    const Tensor& x = e->tens[t];
    if (x.pstate == PS_EQ) return {e->T(t), 0};
    if (x.pstate == PS_RELU) return {e->T(t), 1};
    return {e->Pv(t), 0};
}

void mark_need(xfr_engine* e, int t)
{
    Tensor& x = e->tens[t];
    if (x.pstate != PS_OTHER || x.need_pv) return;
    x.need_pv = true;
    if (x.producer < 0) return;
    const xfr_op_desc& d = e->ops[x.producer].d;
    if (!is_hooked(d.kind)) {   // glue consumes positive-pass values of its inputs
        mark_need(e, d.in0);
        if (d.kind == XFR_OP_G_ADD) mark_need(e, d.in1);
    }
}

void compute_need(xfr_engine* e)
{
    for (auto& t : e->tens) t.need_pv = false;
    for (size_t t = 0; t < e->tens.size(); ++t)
        for (const Hook& h : e->tens[t].hooks)
            if (hook_action(e->mode, e->ops[h.op].d.kind) == HOOK_DIV) {
                // x of the hook = relu(positive-pass value of the call's LAST input); for an in-place ReLU the call's input
                const xfr_op_desc& d = e->ops[h.op].d;
                const int xt = (d.kind == XFR_OP_ADD) ? d.in1 : d.in0;
                mark_need(e, xt);
            }
    e->need_dirty = false;
    e->plans.clear();
    e->stat_plan = nullptr;       // the cached descriptor table belonged to one of those plans
}

// ---------------------------------------------------------------------------------------------------------------
xfr_status layout_workspace(xfr_engine* e)
{
    const size_t B = (size_t)e->max_batch;
    for (auto& x : e->tens)
        if (2 * B * (size_t)x.per_n() * sizeof(float) >= (1ull << 31))
            return fail(XFR_INVALID_ARG, "max_batch %d makes a tensor exceed 2 GiB (32-bit buffer offsets)", e->max_batch);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return o; };
    e->x_off = take(B * e->tens[0].per_n());
    // ConcatChannels (resnet.py:210-213) pads the pooled shortcut with zero channels.  In CNHW a channel prefix is a storage prefix for every
    // batch size, so the pooled tensor lives INSIDE the padded one: the average pool writes it there and the padding is one fill, no copy.
    for (auto& o : e->ops) {
        if (o.d.kind != XFR_OP_CONCAT) continue;
        Tensor& in = e->tens[o.d.in0];
        if (in.alias >= 0 || e->tens[o.d.out].alias >= 0 || in.consumers.size() != 1 || in.producer < 0 ||
            e->ops[in.producer].d.kind != XFR_OP_AVGPOOL)
            continue;
        in.prefix_of = o.d.out;
    }
    for (size_t t = 0; t < e->tens.size(); ++t) {
        Tensor& x = e->tens[t];
        if (x.alias < 0 && x.prefix_of < 0) x.t_off = take(B * x.per_n());
    }
    for (auto& x : e->tens)
        if (x.prefix_of >= 0) x.t_off = e->tens[x.prefix_of].t_off;
    e->t_region_floats = off;
    for (size_t t = 0; t < e->tens.size(); ++t) {
        Tensor& x = e->tens[t];
        if (x.pstate == PS_OTHER) x.pv_off = take(B * x.per_n());
    }
    // normalize norms live in the forward region too (written by the forward, read by the backward)
    size_t misc = 0;
    size_t idxb = 0;
    for (auto& o : e->ops) {
        if (o.d.kind == XFR_OP_G_NORMALIZE) { o.norm_off = misc; misc += align_up(B, 64); }
        if (o.d.kind == XFR_OP_MAXPOOL) { o.idx_off = idxb; idxb += align_up(B * e->tens[o.d.out].per_n(), 256); }
    }
    e->misc_off = take(std::max<size_t>(misc, 64));
    take(4096);
    e->fwd_region_floats = off;
    e->g_begin = off;
    for (size_t t = 1; t < e->tens.size(); ++t) e->tens[t].g_off = take(2 * B * e->tens[t].per_n());
    e->g_end = off;
    size_t max_per_n = 0;
    for (auto& x : e->tens) max_per_n = std::max(max_per_n, (size_t)x.per_n());
    e->seed_off = take(2 * B * max_per_n);
    const Tensor& t1 = e->tens[1];
    e->tap_off = take(2 * B * t1.per_n());
    e->pooled_off = take(2 * B * t1.HW());
    e->blur_a_off = take(2 * B * std::max(t1.HW(), 1));
    e->blur_b_off = take(2 * B * std::max(t1.HW(), 1));
    e->thr_off = take(B);
    take(4096);   // slack: vector loads of a tile's dead columns may run past the last tensor
    e->ws_floats = off;
    e->idx_bytes = std::max<size_t>(idxb, 256);
    return XFR_OK;
}

xfr_status allocate(xfr_engine* e)
{
    xfr_status st = layout_workspace(e);
    if (st != XFR_OK) return st;
    const size_t B = (size_t)e->max_batch;
    HIP_TRY(hipMalloc(&e->ws, e->ws_floats * sizeof(float)));
    HIP_TRY(hipMalloc(&e->idx_ws, e->idx_bytes));
    size_t hooks = 0;
    for (auto& x : e->tens) hooks += x.hooks.size();
    e->trace_cap = hooks;
    HIP_TRY(hipMalloc(&e->dbl_ws, sizeof(double) * (2 * B + hooks * 2 * B + 64)));
    HIP_TRY(hipMalloc(&e->trunc_ws, truncation_scratch_bytes((int)B)));
    return XFR_OK;
}

// parameter arena layout
xfr_status layout_arena(xfr_engine* e, bool device = true)
{
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return (long)o; };
    for (size_t k = 0; k < e->ops.size(); ++k) {
        OpRec& o = e->ops[k];
        const xfr_op_desc& d = o.d;
        if (d.kind == XFR_OP_CONV || d.kind == XFR_OP_LINEAR) {
            o.ldw = (int)align_up(d.cout, 128);
            o.tap_fwd = (d.kh * d.kw > 1) && (o.Cin % 16 == 0) && (d.kh * d.kw <= 64);
            o.tap_bwd = (d.kh * d.kw > 1) && (d.cout % 16 == 0) && (d.kh * d.kw <= 64);
            o.tap4_fwd = (d.kh * d.kw > 1) && (o.Cin == 3 || o.Cin == 4) && (d.kh * d.kw <= 60);
            o.Kf = o.tap4_fwd ? 4 * d.kh * d.kw : o.K;
            o.w_true = take(align_up(o.Kf, 32) * o.ldw);
            o.w_pos = take(align_up(o.Kf, 32) * o.ldw);
            if (k != 0) {
                o.ldb = (int)align_up(o.Cin, 128);
                o.w_bwd = take(align_up(o.Kb, 32) * o.ldb);
                o.w_bwd_true = take(align_up(o.Kb, 32) * o.ldb);
            }
            if (d.w_bias >= 0) { o.b_true = take(d.cout); o.b_pos = take(d.cout); }
        } else if (d.kind == XFR_OP_BATCHNORM) {
            const int C = e->tens[d.out].C;
            o.bn_alpha_t = take(C); o.bn_beta_t = take(C); o.bn_alpha_p = take(C); o.bn_beta_p = take(C); o.bn_beta_pb = take(C);
        }
    }
    e->arena_floats = std::max<size_t>(off, 64);
    if (device) HIP_TRY(hipMalloc(&e->arena, e->arena_floats * sizeof(float)));
    return XFR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
xfr_status run_conv(xfr_engine* e, const ConvParams& p_in, hipStream_t s)
{
    ConvParams p = p_in;
    p.chain_interpret = e->interpret_chains ? 1 : 0;
    p.split_ok = (e->split_mask & (p.bwd ? 2 : 1)) ? (e->split_any_grid ? 2 : 1) : 0;
    p.tail_force = 1;
    if (e->tail_balance) {
        p.tail_force = 0;
        int k = 0;
        while (k < e->n_tail_ws && e->tail_ws[k].s != s) ++k;
        if (k == e->n_tail_ws && k < 8) {
            float* ws = nullptr;
            HIP_TRY(hipMalloc(&ws, XFR_TAIL_WS_BYTES + XFR_TAIL_MAX_TILES * sizeof(unsigned)));
            unsigned* cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + XFR_TAIL_WS_BYTES);
            HIP_TRY(hipMemset(cnt, 0, XFR_TAIL_MAX_TILES * sizeof(unsigned)));
            HIP_TRY(hipDeviceSynchronize());
            e->tail_ws[k] = {s, ws, cnt};
            e->n_tail_ws = k + 1;
        }
        if (k < e->n_tail_ws) {
            p.tail_ws = e->tail_ws[k].ws;
            p.tail_cnt = e->tail_ws[k].cnt;
            p.tail_ws_bytes = XFR_TAIL_WS_BYTES;
        }
    }
    if (e->profile_on) {
        if (e->ev_used == e->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            e->ev_pool.emplace_back(a, b);
        }
        if (e->ev_params.size() < e->ev_pool.size()) { e->ev_params.resize(e->ev_pool.size()); e->ev_cfg.resize(e->ev_pool.size()); }
        const int why = conv_gemm_cannot_launch(p);
        if (why) return fail(XFR_STATE_ERROR, "%s", conv_gemm_refusal(why));          // nothing launched: no event pair, no record
        // (HIP events misread the FIRST GEMM of a profiled run -- 0.87 ms for a 0.37 ms stem in round 3, 1.03 ms with a stream synchronise in
        // front of it in round 4: the start event is stamped on a queue that has just been idle.  The per-shape tables of profiles/ therefore
        // also come from the kernels' own stamps: bench.py --serial --launch-log-csv, profiles/layer_table.py.)
        e->ev_params[e->ev_used] = p;
        auto& ev = e->ev_pool[e->ev_used++];
        HIP_TRY(hipEventRecord(ev.first, s));
        launch_conv_gemm(p, s);
        e->ev_cfg[e->ev_used - 1] = conv_gemm_last_cfg();
        HIP_TRY(hipEventRecord(ev.second, s));
        e->prof_flops += 2.0 * (double)(p.K_logical ? p.K_logical : p.K) * (double)p.M * (double)p.CoutTot * (double)(p.dualacc ? 2 : p.nhalves);
    } else if (!launch_conv_gemm(p, s)) {
        return fail(XFR_STATE_ERROR, "%s", conv_gemm_refusal(conv_gemm_cannot_launch(p)));
    }
    return XFR_OK;
}

void conv_geometry(xfr_engine* e, int k, int NB, ConvParams& p)
{
    const OpRec& o = e->ops[k];
    const xfr_op_desc& d = o.d;
    const Tensor& a = e->tens[d.in0];
    const Tensor& t = e->tens[d.out];
    memset(&p, 0, sizeof(p));
    p.Cin = a.C; p.H = a.H; p.W = a.W; p.NB = NB;
    p.kh = d.kh; p.kw = d.kw; p.stride = d.stride; p.pad = d.pad;
    p.OH = t.H; p.OW = t.W;
    p.K = o.Kf; p.K_logical = o.K; p.M = NB * t.H * t.W;
    p.ldw = o.ldw;
    p.out_H = t.H; p.out_W = t.W; p.out_stride = 1;
    p.in_nb = NB; p.out_nb = NB;
    p.in_bytes = (unsigned)((size_t)NB * a.per_n() * sizeof(float));
    p.tap_major = o.tap4_fwd ? 2 : (o.tap_fwd ? 1 : 0);
    p.co_pair = o.pair;
}

// Forward-only runs (encode, embeddings, the gallery of a triplet step) never need the raw convolution output:
xfr_status fwd_op(xfr_engine* e, int k, int B, bool want_pos, hipStream_t s);
xfr_status pos_op(xfr_engine* e, int k, int B, hipStream_t s);

// The reference evaluates a down-sampling block's shortcut BEHIND the main path (resnet.py:144-146: `residual = self.downsample(x)` after
// bn3), so in program order the residual operand does not exist yet when the block's last convolution is launched and the add kept its own
// kernel.  Nothing orders the two branches: when the operand is the end of a short chain of pooling / padding ops over tensors that exist,
// run that chain now (true values and, in a probe forward, its positive values) and mark it done.  Returns true when tensor `t` exists afterwards.
bool operand_ready(xfr_engine* e, int t, int k, int B, bool with_pos, hipStream_t s)
{
    if (e->tens[t].producer < k) return true;
    if (!e->hoist_shortcut) return false;
    int chain[4], n = 0;
    for (int u = t; e->tens[u].producer >= k; u = e->ops[e->tens[u].producer].d.in0) {
        const int kp = e->tens[u].producer;
        if (e->fwd_done[kp]) break;                       // enqueued already (stream order makes it exist)
        const xfr_op_desc& d = e->ops[kp].d;
        if (kp > e->fwd_last_op || n == 4 || (d.kind != XFR_OP_AVGPOOL && d.kind != XFR_OP_CONCAT)) return false;
        if (with_pos && e->tens[d.out].need_pv && d.kind != XFR_OP_AVGPOOL) return false;     // pos_op computes no padded positive value
        chain[n++] = kp;
    }
    for (int i = n - 1; i >= 0; --i) {
        const int kp = chain[i];
        if (!e->planning_only && !e->dry_run) {
            if (!e->fwd_done[kp] && fwd_op(e, kp, B, with_pos, s) != XFR_OK) return false;
            if (with_pos && e->tens[e->ops[kp].d.out].need_pv && !e->pos_done[kp] && pos_op(e, kp, B, s) != XFR_OK) return false;
        }
        e->fwd_done[kp] = 1;
        e->pos_done[kp] = 1;
    }
    return true;
}

// Conv -> BatchNorm [-> Add with an already computed operand] [-> in-place ReLU] runs in the GEMM's chain epilogue
// (per-channel affine, residual read as 16-byte pieces, clamp), same arithmetic in the same order as the stand-alone
// kernels.
void fuse_forward_only(xfr_engine* e, int k, int B, ConvParams& p, hipStream_t s)
{
    const xfr_op_desc& d = e->ops[k].d;
    const Tensor& c = e->tens[d.out];
    if (c.consumers.size() != 1) return;
    const int k1 = c.consumers[0];
    if (k1 > e->fwd_last_op) return;
    const OpRec& bn = e->ops[k1];
    if (bn.d.kind != XFR_OP_BATCHNORM) return;
    const int bn_out = bn.d.out;
    EwChain& ch = p.chain;
    ch.n = 0;
    auto push = [&](int type) -> EwStep& { EwStep& q = ch.s[ch.n++]; memset(&q, 0, sizeof(q)); q.type = type; q.prior_sb = -1; return q; };
    {
        EwStep& q = push(EW_AFFINE_C);
        q.p0 = e->arena + bn.bn_alpha_t;
        q.p1 = e->arena + bn.bn_beta_t;
    }
    int final_t = bn_out, k2 = -1;
    bool fused_add = false;
    if (!bn.fuse_relu && e->tens[bn_out].consumers.size() == 1) {
        k2 = e->tens[bn_out].consumers[0];
        const OpRec& ad = e->ops[k2];
        if (k2 <= e->fwd_last_op && (ad.d.kind == XFR_OP_ADD || ad.d.kind == XFR_OP_G_ADD)) {
            const int other = (ad.d.in0 == bn_out) ? ad.d.in1 : ad.d.in0;
            if (other != bn_out && operand_ready(e, other, k, B, false, s)) {          // the other operand is already computed (or is now)
                push(EW_ADDP).p0 = e->T(other);
                if (ad.fuse_relu) push(EW_RELU);
                final_t = ad.d.out;
                fused_add = true;
            }
        }
    }
    if (!fused_add && bn.fuse_relu) push(EW_RELU);
    p.out0 = e->T(final_t);
    p.chain_B = B;
    p.chain_eps = e->eps;
    e->fwd_done[k1] = 1;
    if (fused_add) e->fwd_done[k2] = 1;
}

// The probe forward (with the positive pass) needs more than the gallery forward: the RAW convolution output stays (the
// BatchNorm hook's a is relu(conv output)), and in the modes that divide by a ReLU / Add input's X the BatchNorm's positive
// output is needed too.  Conv -> BatchNorm [-> in-place ReLU] then runs as: STORE raw, [FORK positive BatchNorm], affine, [clamp].
// The residual add is left to its own kernel (its pre-add operand is hook state as well).  Returns false if nothing was fused.
bool can_fuse_probe(xfr_engine* e, int k, int* k1_out)
{
    const xfr_op_desc& d = e->ops[k].d;
    const Tensor& c = e->tens[d.out];
    if (c.consumers.size() != 1) return false;
    const int k1 = c.consumers[0];
    if (k1 > e->fwd_last_op || e->ops[k1].d.kind != XFR_OP_BATCHNORM) return false;
    *k1_out = k1;
    return true;
}

// Lean variant (`dual` launches of a lean call, xfr_engine_set_lean): the W and relu(W) accumulators meet in ONE workgroup (ConvParams::dualacc), so
// the BatchNorm hook's a / (x + eps) is formed there and stored in place of the raw output (EW_LEAN_Q ... EW_LEAN_STORE); where the in-place ReLU
// behind the BatchNorm [+ functional add] has a dividing hook too, its quotient replaces the positive BatchNorm output (EW_LEAN_XR).  Two
// tensors written per convolution instead of three or four, and the sweep reads one or two instead of three or four.
void fuse_probe_forward(xfr_engine* e, int k, int B, ConvParams& p, hipStream_t s, bool dual = false, bool lean_try = true)
{
    int k1 = -1;
    if (!can_fuse_probe(e, k, &k1)) return;
    const xfr_op_desc& d = e->ops[k].d;
    const OpRec& bn = e->ops[k1];
    const int bn_out = bn.d.out;
    EwChain& ch = p.chain;
    ch.n = 0;
    auto push = [&](int type) -> EwStep& { EwStep& q = ch.s[ch.n++]; memset(&q, 0, sizeof(q)); q.type = type; q.prior_sb = -1; return q; };
    // lean: every hook on the raw output is the BatchNorm's (its a and x are the two accumulators), and the call asked for it
    // (tuning: XFR_LEAN_MAX_K -- only convolutions with at most that many K rows go lean)
    const int lean_max_k = [] { const char* v = getenv("XFR_LEAN_MAX_K"); return v ? atoi(v) : 512; }();
    // ... and for KxK convolutions: their two-accumulator form ties with the dual launch up to K = 1152 (measured, profiles/r5/experiments/lean_k_threshold.txt)
    const int lean_max_k3 = [&] { const char* v = getenv("XFR_LEAN_MAX_K3"); return v ? atoi(v) : std::max(lean_max_k, 1152); }();
    bool lean = lean_try && dual && d.out != 1 && e->ops[k].Kf <= (d.kh * d.kw > 1 ? lean_max_k3 : lean_max_k) && e->tens[d.out].need_pv &&
                (e->lean_decide || (e->lean_cur && e->lean_cur->lean_q[d.out] == 1));
    if (lean)
        for (const Hook& h : e->tens[d.out].hooks)
            if (h.op != k1 || h.a_tensor != d.out) lean = false;
    if (lean) push(EW_LEAN_Q);
    else push(EW_STORE).pstore = e->T(d.out);
    int fork_at = -1;
    if (e->tens[bn_out].need_pv) {
        fork_at = ch.n;
        EwStep& q = push(EW_FORK_POSBN);
        q.p0 = e->arena + bn.bn_alpha_p;
        q.p1 = e->arena + (e->with_bias ? bn.bn_beta_pb : bn.bn_beta_p);
        q.pstore = e->Pv(bn_out);
        e->pos_done[k1] = 1;
    }
    {
        EwStep& q = push(EW_AFFINE_C);
        q.p0 = e->arena + bn.bn_alpha_t;
        q.p1 = e->arena + bn.bn_beta_t;
    }
    // The residual add behind the BatchNorm joins the chain where the pre-add tensor is nobody's business afterwards: no hook takes its
    // (a, x) from it (the reference's Add hooks both use the LAST input, the residual: whitebox.py:379-381), and the positive pass
    // does not need the sum's inputs (modes that divide by a ReLU input's X compute it from them: then the add keeps its kernel).
    int final_t = bn_out, k2 = -1, pos_add = -1;
    bool fused_add = false;
    if (!bn.fuse_relu && e->tens[bn_out].consumers.size() == 1 && !e->is_hook_a[bn_out]) {
        k2 = e->tens[bn_out].consumers[0];
        const OpRec& ad = e->ops[k2];
        // (the positive pass of a functional add reads its inputs' POSITIVE values, never the true ones: only the Add module's does)
        if (k2 <= e->fwd_last_op && (ad.d.kind == XFR_OP_G_ADD || (ad.d.kind == XFR_OP_ADD && !e->tens[ad.d.out].need_pv))) {
            const int other = (ad.d.in0 == bn_out) ? ad.d.in1 : ad.d.in0;
            if (other != bn_out && operand_ready(e, other, k, B, true, s)) {
                push(EW_ADDP).p0 = e->T(other);
                if (ad.fuse_relu) push(EW_RELU);
                final_t = ad.d.out;
                fused_add = true;
                // The FUNCTIONAL add's positive-pass output (resnet50_128.py: torch.add(shortcut, 1, bn); 'norelu' / 'all' divide by it at the ReLU
                // behind it) is pv(shortcut) + positive BatchNorm: the fork adds the other operand -- already computed -- and stores the SUM; the
                // BatchNorm's own positive output has no other reader (single consumer, no hook takes its x from it).  One add2 launch per block less.
                if (e->fuse_pools && fork_at >= 0 && ad.d.kind == XFR_OP_G_ADD && e->tens[ad.d.out].need_pv) {
                    const Src o2 = pv_src(e, other);
                    EwStep& q = ch.s[fork_at];
                    q.p2 = o2.p;
                    q.action = o2.relu ? 1 : 0;
                    q.pstore = e->Pv(ad.d.out);
                    pos_add = k2;
                }
            }
        }
    }
    if (!fused_add && bn.fuse_relu) push(EW_RELU);
    int lean_tq = -1;
    if (lean) {
        const bool ends_relu = fused_add ? e->ops[k2].fuse_relu : bn.fuse_relu;
        // the ReLU hook's quotient: the fork's value has one reader, the x of the hook of the in-place ReLU that ends this chain
        if (fork_at >= 0 && ends_relu) {
            const int cand = pos_add >= 0 ? e->ops[pos_add].d.out : (fused_add ? -1 : bn_out);
            if (cand >= 0 && e->tens[cand].consumers.size() == 1 && e->ops[e->tens[cand].consumers[0]].d.kind == XFR_OP_RELU &&
                e->ops[e->tens[cand].consumers[0]].relu_fused_away && e->root(e->ops[e->tens[cand].consumers[0]].d.out) == e->root(final_t))
                lean_tq = cand;
        }
        if (lean_tq >= 0) { ch.s[fork_at].type = EW_LEAN_XR; ch.s[fork_at].pstore = nullptr; }
        { EwStep& q = push(EW_LEAN_STORE); q.action = 0; q.pstore = e->T(d.out); }
        if (lean_tq >= 0) { EwStep& q = push(EW_LEAN_STORE); q.action = 1; q.pstore = e->Pv(lean_tq); }
        if (e->lean_decide) {
            e->lean_q_run[d.out] = 1;
            e->lean_final_run[d.out] = ends_relu ? e->root(final_t) : -1;
            if (lean_tq >= 0) e->lean_q_run[lean_tq] = 2;
        }
    }
    // a dual launch can only carry a chain through the compiled float4 epilogue (conv_gemm.hip): rows that are a multiple of 4
    // long and a signature that is in the table; otherwise the BatchNorm keeps its own kernel
    {
        const Tensor& t = e->tens[d.out];
        EwChain probe = ch;
        EwLoads ld;
        ew_plan_loads(probe, e->T(final_t), ld, EW_FWD_SLOTS_WIDE);
        if (!e->planning_only && ((((long)B * t.HW()) & 3) != 0 || conv_gemm_chain_sig(probe) < 0)) {
            ch.n = 0;
            e->pos_done[k1] = 0;
            if (pos_add >= 0) e->pos_done[pos_add] = 0;
            if (lean) {     // no compiled lean epilogue: the plan as a whole stays literal (lean_prepare), this launch too
                e->lean_missing_sig = true;
                if (e->lean_decide) { e->lean_q_run[d.out] = 0; e->lean_final_run[d.out] = -1; if (lean_tq >= 0) e->lean_q_run[lean_tq] = 0; }
                fuse_probe_forward(e, k, B, p, s, dual, false);
            }
            return;
        }
    }
    p.out0 = e->T(final_t);
    p.chain_B = B;
    p.chain_eps = e->eps;
    e->fwd_done[k1] = 1;
    if (fused_add) e->fwd_done[k2] = 1;
    if (pos_add >= 0) e->pos_done[pos_add] = 1;
}

// MaxFeatureMap in the convolution's epilogue (lightcnn.py:48-62: Conv -> Split -> torch.max of the halves).  The forward pack holds
// the two halves interleaved, so a channel and its partner are neighbouring rows of one accumulator tile: the epilogue stores the raw
// rows where the Split hook and the VJP expect them (keep_raw; a forward-only run needs neither) and the even rows store the maximum.
// Returns false -- nothing fused, the three ops run as before -- where the float4 epilogue does not apply.
bool fuse_mfm_forward(xfr_engine* e, int k, int B, bool keep_raw, ConvParams& p)
{
    const OpRec& o = e->ops[k];
    if (!o.pair || o.pair_max > e->fwd_last_op || e->interpret_chains) return false;      // the interpreter has no EW_MAXPAIR
    const Tensor& t = e->tens[o.d.out];
    EwChain ch;
    ch.n = 0;
    auto push = [&](int type) -> EwStep& { EwStep& q = ch.s[ch.n++]; memset(&q, 0, sizeof(q)); q.type = type; q.prior_sb = -1; return q; };
    if (keep_raw) push(EW_STORE).pstore = e->T(o.d.out);
    push(EW_MAXPAIR);
    float* dst = e->T(e->ops[o.pair_max].d.out);
    // The resblock's Add (lightcnn.py:88: out = mfm(mfm(x)) + x) behind the pair maximum: nobody else reads the maximum (the Add hooks take their
    // (a, x) from the LAST input, the residual), so the even rows store the sum -- and, where a hook divides by it, the Add's positive-pass output
    // relu(max) + relu(residual) -- instead of the maximum; the add2 launches (true and positive) go away.
    int k3 = -1;
    {
        const int tmax = e->ops[o.pair_max].d.out;
        const Tensor& tm = e->tens[tmax];
        if (e->fuse_pools && tm.consumers.size() == 1 && !e->is_hook_a[tmax] && !tm.need_pv) {
            const int kc = tm.consumers[0];
            const OpRec& ad = e->ops[kc];
            const int other = ad.d.in0 == tmax ? ad.d.in1 : ad.d.in0;
            if (kc <= e->fwd_last_op && ad.d.kind == XFR_OP_ADD && !ad.fuse_relu && other != tmax && e->tens[other].producer < k &&
                e->tens[other].alias < 0) {
                if (keep_raw && e->tens[ad.d.out].need_pv) {
                    EwStep& q = push(EW_FORK_POSADD);
                    q.p0 = e->T(other);
                    q.pstore = e->Pv(ad.d.out);
                    // pos_op: relu on an input unless it is provably >= 0; bit 0 = the maximum, bit 1 = the residual
                    q.action = (tm.nonneg ? 0 : 1) | (e->tens[other].nonneg ? 0 : 2);
                }
                push(EW_ADDP_CO).p0 = e->T(other);
                dst = e->T(ad.d.out);
                k3 = kc;
            }
        }
    }
    if (!e->planning_only) {
        if ((((long)B * t.HW()) & 3) != 0) return false;
        auto compiled = [&](const EwChain& c, const float* d) { EwChain probe = c; EwLoads ld; ew_plan_loads(probe, d, ld, EW_FWD_SLOTS_WIDE); return conv_gemm_chain_sig(probe) >= 0; };
        if (!compiled(ch, dst)) {
            // a network outside the signature table: without the resblock's Add the chain is [STORE raw,] MAXPAIR again (the Add keeps its kernel)
            if (k3 < 0) return false;
            ch.n = 0;
            if (keep_raw) push(EW_STORE).pstore = e->T(o.d.out);
            push(EW_MAXPAIR);
            dst = e->T(e->ops[o.pair_max].d.out);
            k3 = -1;
            if (!compiled(ch, dst)) return false;
        }
    }
    p.chain = ch;
    p.out0 = dst;
    p.chain_B = B;
    p.chain_eps = e->eps;
    e->fwd_done[o.pair_split] = 1;
    e->fwd_done[o.pair_max] = 1;
    if (k3 >= 0) { e->fwd_done[k3] = 1; e->pos_done[k3] = 1; }
    return true;
}

// lightcnn.py:252: `pool = MaxPool2d(2)(x) + AvgPool2d(2)(x)` -- MAXPOOL(k), AVGPOOL(k+1) on the same x, G_ADD(k+2) of the two, nobody else
// reading the pools' outputs: one pass over x writes the sum, the argmax bytes and (if the consumer's hook divides by it) the positive-pass sum,
// instead of max-pool, average pool (twice with the positive pass) and two adds.  Bit-identical (pool2_fwd_kernel).  false: nothing was launched.
bool fuse_pool2_forward(xfr_engine* e, int k, int B, bool want_pos, hipStream_t s)
{
    if (!e->fuse_pools || k + 2 > e->fwd_last_op || k + 2 >= (int)e->ops.size()) return false;
    const xfr_op_desc& dm = e->ops[k].d;
    const xfr_op_desc& da = e->ops[k + 1].d;
    const xfr_op_desc& dd = e->ops[k + 2].d;
    if (da.kind != XFR_OP_AVGPOOL || dd.kind != XFR_OP_G_ADD || da.in0 != dm.in0) return false;
    if (!((dd.in0 == dm.out && dd.in1 == da.out) || (dd.in0 == da.out && dd.in1 == dm.out))) return false;
    if (dm.kh != 2 || dm.kw != 2 || dm.stride != 2 || dm.pad != 0 || da.kh != 2 || da.kw != 2 || da.stride != 2) return false;
    const Tensor& x = e->tens[dm.in0];
    const Tensor& tm = e->tens[dm.out];
    const Tensor& ta = e->tens[da.out];
    const Tensor& ts = e->tens[dd.out];
    if (tm.consumers.size() != 1 || ta.consumers.size() != 1 || ta.alias >= 0 || tm.need_pv) return false;
    uint8_t* idx = e->t_bank ? nullptr : e->idx_base() + e->ops[k].idx_off;
    if (!pool2_fwd_ok(e->T(dm.in0), idx, x.C * B, x.H, x.W, tm.H, tm.W)) return false;
    float* pos = nullptr;
    int relu_max = 0, avg_mode = 0;
    if (want_pos && ts.need_pv) {
        if (tm.pstate == PS_OTHER) return false;
        relu_max = tm.pstate == PS_RELU ? 1 : 0;
        avg_mode = ta.pstate == PS_EQ ? 0 : (ta.pstate == PS_RELU ? 1 : 2);
        if (avg_mode == 2 && x.nonneg) avg_mode = 0;          // the positive average pool clamps its input only where it is signed (pos_op)
        pos = e->Pv(dd.out);
    }
    launch_pool2_fwd(e->T(dm.in0), e->T(dd.out), idx, pos, x.C * B, x.H, x.W, tm.H, tm.W, relu_max, avg_mode, s);
    e->fwd_done[k + 1] = 1;
    e->fwd_done[k + 2] = 1;
    e->pos_done[k + 1] = 1;
    e->pos_done[k + 2] = 1;
    return true;
}

// forward of op k on true values (and, for "dual" convolutions, the positive output in the same launch)
xfr_status fwd_op(xfr_engine* e, int k, int B, bool want_pos, hipStream_t s)
{
    OpRec& o = e->ops[k];
    const xfr_op_desc& d = o.d;
    const Tensor& a = e->tens[d.in0];
    const Tensor& t = e->tens[d.out];
    const long n_in = (long)B * a.per_n(), n_out = (long)B * t.per_n();
    switch (d.kind) {
        case XFR_OP_CONV:
        case XFR_OP_LINEAR: {
            ConvParams p;
            conv_geometry(e, k, B, p);
            p.in = e->T(d.in0);
            p.w = e->arena + o.w_true;
            p.bias = o.b_true >= 0 ? e->arena + o.b_true : nullptr;
            p.out0 = e->T(d.out);
            p.out1 = nullptr;
            p.CoutTot = d.cout;
            // dual launch: positive activations X = relu(W)*A + b from the same staged input tile.  Valid when the true
            // input is already A (provably >= 0).
            const bool dual = want_pos && t.need_pv && a.nonneg;
            if (dual) {
                p.w_pos = e->arena + o.w_pos;
                p.bias_pos = o.b_true >= 0 ? e->arena + (e->with_bias ? o.b_pos : o.b_true) : nullptr;
                p.out1 = e->Pv(d.out);
                p.nhalves = 2;
            } else p.nhalves = 1;
            // Light-CNN's first layer (one input channel, 5x5, MaxFeatureMap): a direct convolution instead of a 25-deep GEMM
            if (o.pair && e->fuse_fwd_only && e->direct_stem && !dual && !p.relu_in && a.C == 1 && d.kh == 5 && d.kw == 5 && d.stride == 1 && d.pad == 2 &&
                !o.tap_fwd && !o.tap4_fwd && o.pair_max <= e->fwd_last_op && !e->interpret_chains && stem5_mfm_ok(e->T(d.in0), B, a.H, a.W)) {
                if (!e->dry_run) launch_stem5_mfm(e->T(d.in0), p.w, o.ldw, p.bias, want_pos ? e->T(d.out) : nullptr, e->T(e->ops[o.pair_max].d.out), o.pair, B, a.H, a.W, s);
                e->fwd_done[o.pair_split] = 1;
                e->fwd_done[o.pair_max] = 1;
                return XFR_OK;
            }
            if (o.pair && e->fuse_fwd_only && !dual && !p.relu_in && fuse_mfm_forward(e, k, B, want_pos, p)) { }
            else if (!want_pos && e->fuse_fwd_only && !p.relu_in) fuse_forward_only(e, k, B, p, s);
            else if (want_pos && e->fuse_probe_fwd && !p.relu_in) fuse_probe_forward(e, k, B, p, s, dual);
            if (p.chain.n > 0 && p.chain.s[0].type == EW_LEAN_Q) {
                // one workgroup per tile accumulates W and relu(W) (the latter from the clamped W fragment): no second pack, no second output
                p.nhalves = 1;
                p.dualacc = 1;
                if (!e->dry_run) e->lean_launches++;
                p.w_pos = nullptr;
                p.out1 = nullptr;
            } else if (e->lean_cur && !e->lean_decide && e->lean_cur->lean_q[d.out] == 1) {
                return fail(XFR_STATE_ERROR, "lean schedule: convolution %d was planned with a lean epilogue and ran without one", k);
            }
            if (e->dry_run) return XFR_OK;
            return run_conv(e, p, s);
        }
        case XFR_OP_BATCHNORM:
            launch_affine_c(e->T(d.in0), e->T(d.out), e->arena + o.bn_alpha_t, e->arena + o.bn_beta_t, t.C, (long)B * t.HW(), 0,
                            o.fuse_relu ? 1 : 0, s);
            return XFR_OK;
        case XFR_OP_RELU:
            if (o.relu_fused_away) return XFR_OK;
            launch_relu(e->T(d.in0), e->T(d.out), n_in, s);
            return XFR_OK;
        case XFR_OP_MAXPOOL:
            if (fuse_pool2_forward(e, k, B, want_pos, s)) return XFR_OK;
            launch_maxpool_fwd(e->T(d.in0), e->T(d.out), e->t_bank ? nullptr : e->idx_base() + o.idx_off, a.C * B, a.H, a.W, t.H, t.W, d.kh, d.stride, d.pad, s);
            return XFR_OK;
        case XFR_OP_AVGPOOL: {
            if (t.alias >= 0) return XFR_OK;
            // the pooled shortcut lives inside its zero-padded form (layout_workspace): the pool writes the padding planes as well
            int zero_planes = 0;
            if (t.prefix_of >= 0 && e->hoist_shortcut && t.consumers[0] <= e->fwd_last_op) {
                zero_planes = (e->tens[t.prefix_of].C - t.C) * B;
                e->fwd_done[t.consumers[0]] = 1;
            }
            launch_avgpool_fwd(e->T(d.in0), e->T(d.out), a.C * B, a.H, a.W, t.H, t.W, d.kh, d.stride, 0, s, zero_planes);
            return XFR_OK;
        }
        case XFR_OP_ADD:
        case XFR_OP_G_ADD:
            launch_add2(e->T(d.in0), e->T(d.in1), e->T(d.out), n_out, 0, 0, o.fuse_relu ? 1 : 0, s);
            return XFR_OK;
        case XFR_OP_CONCAT:
            if (e->T(d.in0) != e->T(d.out)) launch_copy_acc(e->T(d.in0), e->T(d.out), n_in, 0, s);
            if (n_out > n_in) launch_fill(e->T(d.out) + n_in, n_out - n_in, 0.f, s);
            return XFR_OK;
        case XFR_OP_MULTIPLY:
            launch_scale(e->T(d.in0), e->T(d.out), n_in, d.fparam, 0, s);
            return XFR_OK;
        case XFR_OP_SPLIT:
            return XFR_OK;
        case XFR_OP_G_MAXHALVES:
            launch_maxhalves_fwd(e->T(d.in0), e->T(d.out), t.C, (long)B * t.HW(), 0, s);
            return XFR_OK;
        case XFR_OP_G_NORMALIZE:
            launch_normalize_fwd(e->T(d.in0), e->T(d.out), e->t_bank ? nullptr : e->misc() + o.norm_off, t.C, B, 0, s);
            return XFR_OK;
    }
    return fail(XFR_UNSUPPORTED_LAYER, "forward: unsupported kind %d", d.kind);
}

// positive pass for tensor out(k) (only called when need_pv and not produced by a dual launch)
xfr_status pos_op(xfr_engine* e, int k, int B, hipStream_t s)
{
    OpRec& o = e->ops[k];
    const xfr_op_desc& d = o.d;
    const Tensor& a = e->tens[d.in0];
    const Tensor& t = e->tens[d.out];
    const long n_out = (long)B * t.per_n();
    switch (d.kind) {
        case XFR_OP_CONV:
        case XFR_OP_LINEAR: {
            ConvParams p;
            conv_geometry(e, k, B, p);
            p.in = e->T(d.in0);
            p.relu_in = a.nonneg ? 0 : 1;
            p.w = e->arena + o.w_pos;
            p.bias = o.b_true >= 0 ? e->arena + (e->with_bias ? o.b_pos : o.b_true) : nullptr;
            p.out0 = e->Pv(d.out);
            p.CoutTot = d.cout; p.nhalves = 1;
            return run_conv(e, p, s);
        }
        case XFR_OP_BATCHNORM:
            launch_affine_c(e->T(d.in0), e->Pv(d.out), e->arena + o.bn_alpha_p, e->arena + (e->with_bias ? o.bn_beta_pb : o.bn_beta_p),
                            t.C, (long)B * t.HW(), a.nonneg ? 0 : 1, 0, s);
            return XFR_OK;
        case XFR_OP_AVGPOOL:
            launch_avgpool_fwd(e->T(d.in0), e->Pv(d.out), a.C * B, a.H, a.W, t.H, t.W, d.kh, d.stride, a.nonneg ? 0 : 1, s);
            return XFR_OK;
        case XFR_OP_ADD: {
            const Tensor& b = e->tens[d.in1];
            launch_add2(e->T(d.in0), e->T(d.in1), e->Pv(d.out), n_out, a.nonneg ? 0 : 1, b.nonneg ? 0 : 1, 0, s);
            return XFR_OK;
        }
        case XFR_OP_G_ADD: {
            const Src x = pv_src(e, d.in0), y = pv_src(e, d.in1);
            launch_add2(x.p, y.p, e->Pv(d.out), n_out, x.relu, y.relu, 0, s);
            return XFR_OK;
        }
        case XFR_OP_G_MAXHALVES: {
            const Src x = pv_src(e, d.in0);
            launch_maxhalves_fwd(x.p, e->Pv(d.out), t.C, (long)B * t.HW(), x.relu, s);
            return XFR_OK;
        }
        case XFR_OP_G_NORMALIZE: {
            const Src x = pv_src(e, d.in0);
            launch_normalize_fwd(x.p, e->Pv(d.out), nullptr, t.C, B, x.relu, s);
            return XFR_OK;
        }
    }
    return fail(XFR_UNSUPPORTED_LAYER, "positive pass: kind %d cannot have a computed positive value", d.kind);
}

xfr_status forward_all(xfr_engine* e, const float* x_dev, int B, int last_tensor, bool with_pos, hipStream_t s)
{
    // xfr_engine_hold_forward: consecutive calls on the same input share one forward (slot 0, main bank only)
    const bool holdable = e->hold_forward && e->cur_slot == 0 && !e->t_bank;
    if (holdable && e->held_x == x_dev && e->held_B == B && e->held_last == last_tensor && e->held_stream == s &&
        (e->held_pos || !with_pos))
        return XFR_OK;
    if (holdable) with_pos = true;           // later calls of the group may need the positive pass
    e->held_x = nullptr;
    const Tensor& in = e->tens[0];
    if (e->dry_run) { }
    else if (e->u8_on) launch_u8hwc_to_cnhw(reinterpret_cast<const uint8_t*>(x_dev), e->T(0), B, in.C, in.HW(), e->u8_pre, s);
    else launch_nchw_to_cnhw(x_dev, e->T(0), B, in.C, in.HW(), s);
    const int last_op = e->tens[last_tensor].producer;
    e->fwd_done.assign(e->ops.size(), 0);
    e->pos_done.assign(e->ops.size(), 0);
    e->fwd_last_op = last_op;
    for (int k = 0; k <= last_op; ++k) {
        xfr_status st = XFR_OK;
        if (!e->fwd_done[k]) st = fwd_op(e, k, B, with_pos, s);
        if (st != XFR_OK) return st;
        if (with_pos) {
            const Tensor& t = e->tens[e->ops[k].d.out];
            const xfr_op_desc& d = e->ops[k].d;
            if (t.need_pv && !e->pos_done[k]) {
                const bool dual_done = (d.kind == XFR_OP_CONV || d.kind == XFR_OP_LINEAR) && e->tens[d.in0].nonneg;
                if (!dual_done) { st = pos_op(e, k, B, s); if (st != XFR_OK) return st; }
            }
        }
    }
    if (holdable) { e->held_x = x_dev; e->held_B = B; e->held_last = last_tensor; e->held_pos = with_pos; e->held_stream = s; }
    return XFR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// backward schedule
bool unary_elementwise(int kind)
{
    return kind == XFR_OP_RELU || kind == XFR_OP_BATCHNORM || kind == XFR_OP_MULTIPLY || kind == XFR_OP_SPLIT;
}

xfr_status make_plan(xfr_engine* e, int seed_tensor, BwdPlan& plan, bool plain)
{
    plan.plain = plain;
    const int nt = (int)e->tens.size();
    plan.seed_tensor = seed_tensor;
    plan.mode = e->mode;
    plan.steps.clear();
    // reachability: every op propagates to all of its inputs
    std::vector<char> reach(nt, 0);
    reach[seed_tensor] = 1;
    for (int k = e->tens[seed_tensor].producer; k >= 0; --k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (!reach[d.out]) continue;
        reach[d.in0] = 1;
        if (d.kind == XFR_OP_ADD || d.kind == XFR_OP_G_ADD) reach[d.in1] = 1;
    }
    // number of gradient contributors per tensor
    std::vector<int> contrib(nt, 0);
    for (int k = 0; k <= e->tens[seed_tensor].producer; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (!reach[d.out]) continue;
        contrib[d.in0]++;
        if (d.kind == XFR_OP_ADD || d.kind == XFR_OP_G_ADD) contrib[d.in1]++;
    }
    // firing order (reference): descending producer index, registration order within a tensor; image last
    std::vector<std::vector<int>> slot(nt);
    plan.firing_kinds.clear();
    plan.firing_ops.clear();
    plan.firing_tensor.clear();
    for (int k = e->tens[seed_tensor].producer; k >= 0; --k) {
        const int t = e->ops[k].d.out;
        if (!reach[t]) continue;
        for (const Hook& h : e->tens[t].hooks) {
            if (e->ops[h.op].d.out > seed_tensor) { slot[t].push_back(-1); continue; }   // call beyond the seed
            slot[t].push_back((int)plan.firing_kinds.size());
            plan.firing_kinds.push_back(e->ops[h.op].d.kind);
            plan.firing_ops.push_back(h.op);
            plan.firing_tensor.push_back(t);
        }
    }
    plan.n_firings = (int)plan.firing_kinds.size();   // (+1 for the image hook of op 0, which is not computed)

    std::vector<char> written(nt, 0), hooks_done(nt, 0), op_done(e->ops.size(), 0);
    written[seed_tensor] = 1;

    auto append_hooks = [&](BwdStep& st, int t) -> bool {
        const Tensor& x = e->tens[t];
        for (size_t i = 0; i < x.hooks.size(); ++i) {
            const Hook& h = x.hooks[i];
            if (slot[t][i] < 0) continue;   // hook of a call that lies beyond the seed tensor
            if (plain) continue;            // plain gradients: the _savegrad hooks only record
            const xfr_op_desc& hd = e->ops[h.op].d;
            BwdStep::Sym sy;
            sy.type = EW_HOOK;
            sy.action = hook_action(e->mode, hd.kind);
            sy.t0 = h.a_tensor;
            const int xt = (hd.kind == XFR_OP_ADD) ? hd.in1 : hd.in0;
            sy.x_t = (e->tens[xt].pstate == PS_OTHER) ? xt : -1;   // -1: x == a
            sy.f = 0.f; sy.op = h.op; sy.slot = slot[t][i]; sy.tap = false;
            st.chain.push_back(sy);
        }
        hooks_done[t] = 1;
        return true;
    };

    for (int k = e->tens[seed_tensor].producer; k >= 1; --k) {
        if (op_done[k]) continue;
        const int t0 = e->ops[k].d.out;
        if (!reach[t0]) continue;
        BwdStep ew;
        ew.kind = ST_EW;
        ew.src_t = t0;
        ew.ew_t = t0;
        if (!hooks_done[t0]) append_hooks(ew, t0);
        int cur_op = k;
        int cur_t = t0;
        bool emitted = false;
        while (true) {
            const xfr_op_desc& d = e->ops[cur_op].d;
            if (!unary_elementwise(d.kind)) break;
            // VJP of the elementwise op
            BwdStep::Sym sy;
            sy.action = 0; sy.x_t = -1; sy.f = 0.f; sy.op = cur_op; sy.slot = -1; sy.tap = false; sy.t0 = -1;
            bool has = true;
            if (d.kind == XFR_OP_RELU) { sy.type = EW_MASK; sy.t0 = d.out; }
            else if (d.kind == XFR_OP_BATCHNORM) { sy.type = EW_SCALE_C; }
            else if (d.kind == XFR_OP_MULTIPLY) { sy.type = EW_SCALE; sy.f = d.fparam; }
            else has = false;
            if (has) ew.chain.push_back(sy);
            op_done[cur_op] = 1;
            const int ti = d.in0;
            if (ti == 0) {   // reached the image: nothing below
                emitted = true;   // nothing to store
                ew.chain.clear();
                break;
            }
            const bool single = (contrib[ti] == 1);
            const bool room = (ew.chain.size() + e->tens[ti].hooks.size() + 2 <= XFR_MAX_EW_STEPS);
            if (single && room && ti != 1) {
                append_hooks(ew, ti);
                cur_t = ti;
                cur_op = e->tens[ti].producer;
                if (op_done[cur_op]) break;
                continue;
            }
            if (single && room && ti == 1) {
                // tensor 1 = output of the first layer: its last hook is P[-2] (whitebox.py:499); stop here
                append_hooks(ew, ti);
                for (int q = (int)ew.chain.size() - 1; q >= 0; --q)
                    if (ew.chain[q].type == EW_HOOK) { ew.chain[q].tap = true; break; }
                ew.dst_t = 1; ew.accumulate = 0;
                plan.steps.push_back(ew);
                return XFR_OK;
            }
            // store into G[ti] (possibly accumulating); its hooks fire later when its producer is visited
            ew.dst_t = ti; ew.accumulate = written[ti] ? 1 : 0;
            written[ti] = 1;
            plan.steps.push_back(ew);
            emitted = true;
            break;
        }
        if (emitted) continue;
        // cur_t's producer (cur_op) is not elementwise (or already done): flush the chain in place, then its VJP
        if (cur_t == 1) {
            // hooks of tensor 1 were appended by a chain that started above; mark the tap
            for (int q = (int)ew.chain.size() - 1; q >= 0; --q)
                if (ew.chain[q].type == EW_HOOK) { ew.chain[q].tap = true; break; }
            ew.dst_t = 1; ew.accumulate = 0;
            plan.steps.push_back(ew);
            return XFR_OK;
        }
        if (!ew.chain.empty() || cur_t != t0) {
            ew.dst_t = cur_t; ew.accumulate = 0;
            plan.steps.push_back(ew);
            written[cur_t] = 1;
        }
        if (op_done[cur_op]) continue;
        op_done[cur_op] = 1;
        const xfr_op_desc& d = e->ops[cur_op].d;
        BwdStep st;
        st.op = cur_op; st.src_t = cur_t;
        auto target = [&](int ti, BwdStep s2) {
            if (ti == 0) return;   // no gradient wrt the image is needed for P[-2]
            s2.dst_t = ti; s2.accumulate = written[ti] ? 1 : 0; written[ti] = 1;
            plan.steps.push_back(s2);
        };
        switch (d.kind) {
            case XFR_OP_CONV:
            case XFR_OP_LINEAR:
                if (d.stride > 1 && d.in0 != 0 && !written[d.in0]) {
                    BwdStep z; z.kind = ST_ZERO; z.dst_t = d.in0; plan.steps.push_back(z); written[d.in0] = 1;
                }
                st.kind = ST_CONV_BWD; target(d.in0, st); break;
            case XFR_OP_MAXPOOL: st.kind = ST_MAXPOOL_BWD; target(d.in0, st); break;
            case XFR_OP_AVGPOOL:
                if (d.kh == 1 && d.stride == 1) { st.kind = ST_COPY; st.copy_elems_per_sb = e->tens[cur_t].C; }    // identity: a gradient copy (often forwarded away)
                else st.kind = ST_AVGPOOL_BWD;
                target(d.in0, st);
                break;
            case XFR_OP_ADD:
            case XFR_OP_G_ADD:
                st.kind = ST_COPY; st.copy_elems_per_sb = e->tens[cur_t].C;
                target(d.in0, st); target(d.in1, st); break;
            case XFR_OP_CONCAT:
                st.kind = ST_COPY; st.copy_elems_per_sb = e->tens[d.in0].C; target(d.in0, st); break;
            case XFR_OP_G_MAXHALVES: {
                // the VJP of max(split[0], split[1]) as the HEAD of an elementwise chain over the 2*Co-channel Split tensor: it then
                // merges with the hook chain that follows (fuse_plan) instead of writing the routed gradient out and reading it back
                st.kind = ST_EW;
                st.ew_t = d.in0;
                BwdStep::Sym sy;
                sy.type = EW_MAXHALF_IN; sy.action = e->tens[d.out].C; sy.t0 = d.in0; sy.x_t = -1; sy.f = 0.f; sy.op = cur_op; sy.slot = -1; sy.tap = false;
                st.chain.push_back(sy);
                target(d.in0, st);
                break;
            }
            case XFR_OP_G_NORMALIZE: st.kind = ST_NORMALIZE_BWD; target(d.in0, st); break;
            default:
                return fail(XFR_UNSUPPORTED_LAYER, "backward: unsupported kind %d", d.kind);
        }
    }
    return fail(XFR_STATE_ERROR, "backward schedule never reached the first layer's output");
}


// ---------------------------------------------------------------------------------------------------------------
// Cross-kernel fusion of the backward schedule.
//   1. copy forwarding: a full-tensor gradient copy (Add / functional-add VJP) becomes an alias; the first later
//      writer that accumulated into the copy's destination instead adds the alias source in its chain (EW_ADDP).
//   2. chain -> chain: EW(a->b) followed by EW(b->c) becomes one launch (with an EW_STORE of b if b has other readers).
//   3. GEMM -> chain: a non-scattering backward-data GEMM whose output only feeds a chain runs that chain in its
//      epilogue, so the gradient between two GEMMs is never written to HBM un-hooked.
// ---- helpers shared by the fusion passes
BwdStep::Sym fuse_mk(int type, int t0)
{
    BwdStep::Sym s; s.type = type; s.action = 0; s.t0 = t0; s.x_t = -1; s.f = 0.f; s.op = -1; s.slot = -1; s.tap = false;
    return s;
}
bool fuse_reads(const BwdStep& b, int t)
{
    if (b.kind != ST_ZERO && b.src_t == t) return true;
    for (const BwdStep::Sym& y : b.chain) if (y.type == EW_ADDP && y.t0 == t) return true;
    for (const BwdStep::Sym& y : b.chain) if (y.type == EW_AVGUP_IN && y.slot == t) return true;      // the compact GEMM result in t's gradient region
    if (b.accumulate && b.dst_t == t) return true;
    return false;
}
bool fuse_writes(const BwdStep& b, int t)
{
    if (b.dst_t == t) return true;
    for (const BwdStep::Sym& y : b.chain) if (y.type == EW_STORE && y.t0 == t) return true;
    return false;
}

// Pass 1: copy forwarding.
void fuse_copy_forwarding(xfr_engine* e, std::vector<BwdStep>& st)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
    {
        std::vector<BwdStep> out;
        std::vector<int> alias(nt, -1);
        for (size_t i = 0; i < st.size(); ++i) {
            BwdStep b = st[i];
            // readers use the alias
            if (b.kind != ST_ZERO && b.src_t >= 0 && alias[b.src_t] >= 0) b.src_t = alias[b.src_t];
            const int d = b.dst_t;
            // ... or a channel-prefix slice (ConcatChannels VJP: rows [0, C_d) of the source, same row stride) whose first toucher is the in-place hook
            // flush of d: that launch then reads the source's rows directly
            const bool full_copy = b.kind == ST_COPY && d >= 0 && e->tens[d].C == e->tens[b.src_t].C && b.copy_elems_per_sb == e->tens[d].C;
            bool prefix_copy = false;
            if (b.kind == ST_COPY && !b.accumulate && d >= 0 && !full_copy && e->fuse_avgup && e->tens[b.src_t].C > e->tens[d].C &&
                b.copy_elems_per_sb == e->tens[d].C && e->tens[b.src_t].HW() == e->tens[d].HW()) {
                for (size_t j = i + 1; j < st.size(); ++j) {
                    const BwdStep& c = st[j];
                    if (c.dst_t == d || (c.kind != ST_ZERO && c.src_t == d) || writes(c, b.src_t)) {
                        prefix_copy = c.kind == ST_EW && c.src_t == d && c.dst_t == d && !c.accumulate && !writes(c, b.src_t);
                        break;
                    }
                }
            }
            if (b.kind == ST_COPY && !b.accumulate && d >= 0 && (full_copy || prefix_copy)) {
                // forward only if every later accumulating writer of d can take an addend in its chain
                bool ok = true;
                bool later_writer = false, later_reader = false;
                for (size_t j = i + 1; j < st.size() && ok; ++j) {
                    const BwdStep& c = st[j];
                    if (c.dst_t == d) {
                        later_writer = true;
                        // a chain flushed IN PLACE on d (hooks of d where its producer is glue) reads the copy's source instead
                        if (!c.accumulate && c.kind == ST_EW && c.src_t == d) break;
                        if (!c.accumulate) { ok = false; break; }
                        if (!(c.kind == ST_EW || (c.kind == ST_CONV_BWD && !scatter_conv(c)))) ok = false;
                        break;   // after the first physical writer the tensor is real again
                    }
                    if (c.kind != ST_ZERO && c.src_t == d) later_reader = true;
                }
                (void)later_writer; (void)later_reader;
                if (ok) { alias[d] = b.src_t; continue; }
            }
            if (d >= 0 && alias[d] >= 0) {
                // first physical writer of an aliased tensor: it was an accumulate; turn it into "+ alias source"
                if (b.accumulate) {
                    b.accumulate = 0;
                    // (behind a MaxFeatureMap head: that step defines the gradient the chain starts from)
                    b.chain.insert(b.chain.begin() + ((!b.chain.empty() && b.chain[0].type == EW_MAXHALF_IN) ? 1 : 0), mk(EW_ADDP, alias[d]));
                    if (b.kind == ST_CONV_BWD) b.ew_t = d;
                }
                alias[d] = -1;
            }
            out.push_back(b);
        }
        st.swap(out);
    }
}

// Pass 1b: Light-CNN's pool pair.
void fuse_pool_pair(xfr_engine* e, std::vector<BwdStep>& st)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
    // ---- 1b. pool pair (lightcnn.py:252: maxpool(x) + avgpool(x), both 2x2 / 2 on the same x).  AVGPOOL_BWD(S -> D), accumulating
    // MAXPOOL_BWD(S -> D) and the in-place hook chain of D (the two pools' tensor hooks on the accumulated gradient) become ONE chain launch
    // whose head (EW_POOL2_IN) builds the summed gradient of a pixel from its window's gradient and argmax byte: D is written once
    // instead of written, read-modified twice and read again (13 -> 9.3 tensor passes per pooling stage with the expanding chain behind it).
    for (size_t i = 0; e->fuse_pools && i + 1 < st.size(); ++i) {
        const BwdStep av = st[i], mx = st[i + 1];
        if (av.kind != ST_AVGPOOL_BWD || mx.kind != ST_MAXPOOL_BWD || av.accumulate || !mx.accumulate) continue;
        if (av.src_t != mx.src_t || av.dst_t != mx.dst_t || av.dst_t < 0) continue;
        const xfr_op_desc& da = e->ops[av.op].d;
        const xfr_op_desc& dm = e->ops[mx.op].d;
        const Tensor& x = e->tens[av.dst_t];
        const Tensor& y = e->tens[dm.out];
        if (da.in0 != av.dst_t || dm.in0 != av.dst_t || da.kh != 2 || da.kw != 2 || da.stride != 2 || dm.kh != 2 || dm.kw != 2 || dm.stride != 2 || dm.pad != 0)
            continue;
        if ((x.W & 3) != 0 || (x.H & 1) != 0 || y.H * 2 != x.H || y.W * 2 != x.W || (e->ops[mx.op].idx_off & 3) != 0) continue;
        BwdStep f;
        f.kind = ST_EW;
        f.src_t = av.src_t;
        f.dst_t = av.dst_t;
        f.ew_t = av.dst_t;
        f.accumulate = 0;
        Sym h = mk(EW_POOL2_IN, -1);
        h.op = mx.op;
        h.action = x.W;
        f.chain.push_back(h);
        size_t drop = 1;
        if (i + 2 < st.size()) {
            const BwdStep& c = st[i + 2];
            const bool headless = c.chain.empty() || (c.chain[0].type != EW_MAXHALF_IN && c.chain[0].type != EW_POOL2_IN);
            if (c.kind == ST_EW && c.src_t == av.dst_t && c.dst_t == av.dst_t && !c.accumulate && c.ew_t == av.dst_t && headless &&
                c.chain.size() + 1 <= XFR_MAX_EW_STEPS) {
                f.chain.insert(f.chain.end(), c.chain.begin(), c.chain.end());
                drop = 2;
            }
        }
        st[i] = f;
        st.erase(st.begin() + i + 1, st.begin() + i + 1 + drop);
    }
}

// Pass 2b: down-sampling residual block with an average-pool shortcut.
void fuse_downsample_avgpool(xfr_engine* e, std::vector<BwdStep>& st)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
        // ---- 2b (GEMM-fused schedules only: no traces, priors or stores there).  Down-sampling residual block, shortcut = AvgPool2d(2) [+ ConcatChannels],
        // main path entered through a 1x1 / stride 2 convolution (resnet.py:111-149).  Its block-input gradient D was built by five launches:
        //   COPY S -> P (channel prefix), EW P (the pooled tensor's hook, in place), AVGPOOL_BWD P -> D, ..., CONV_BWD -> D (scatter, read-modify-write),
        //   EW D -> E (the block input's hook chain).
        // Now the GEMM leaves its result compact and the last launch builds D's value per pixel in its head (EW_AVGUP_IN): D is never written,
        // three launches are gone and the GEMM stores rows instead of scattering dwords.
        for (size_t i0 = 0; i0 < st.size(); ++i0) {
            const BwdStep cp = st[i0];
            // (the slice copy may already have been forwarded into the pooled tensor's hook launch: EW S -> P, one hook)
            const bool fwd_hook = cp.kind == ST_EW && cp.src_t != cp.dst_t && cp.chain.size() == 1 && cp.chain[0].type == EW_HOOK && !cp.chain[0].tap &&
                                  cp.ew_t == cp.dst_t;
            if ((cp.kind != ST_COPY && !fwd_hook) || cp.accumulate || cp.dst_t < 0 || cp.src_t < 0) continue;
            const int S = cp.src_t, P = cp.dst_t;
            const Tensor& tp = e->tens[P];
            if ((cp.kind == ST_COPY && cp.copy_elems_per_sb != tp.C) || e->tens[S].C < tp.C || e->tens[S].H != tp.H || e->tens[S].W != tp.W) continue;
            auto next_touch = [&](size_t from, int t) {
                size_t k = from;
                for (; k < st.size(); ++k)
                    if (reads(st[k], t) || writes(st[k], t)) break;
                return k;
            };
            size_t i1 = next_touch(i0 + 1, P);
            if (i1 >= st.size()) continue;
            Sym hook = mk(EW_AVGUP_IN, -1);
            hook.action = -1;
            size_t i2 = i1;
            if (fwd_hook) {
                hook.action = cp.chain[0].action;
                if (hook.action == HOOK_DIV) { hook.t0 = cp.chain[0].t0; hook.x_t = cp.chain[0].x_t; }
            } else if (st[i1].kind == ST_EW) {          // the pooled tensor's hook, in place
                const BwdStep& h = st[i1];
                if (h.src_t != P || h.dst_t != P || h.accumulate || h.chain.size() != 1 || h.chain[0].type != EW_HOOK || h.chain[0].tap) continue;
                hook.action = h.chain[0].action;
                if (hook.action == HOOK_DIV) { hook.t0 = h.chain[0].t0; hook.x_t = h.chain[0].x_t; }     // otherwise p is not observed in this schedule
                i2 = next_touch(i1 + 1, P);
                if (i2 >= st.size()) continue;
            }
            const BwdStep av = st[i2];
            if (av.kind != ST_AVGPOOL_BWD || av.src_t != P || av.accumulate || av.dst_t < 0) continue;
            const xfr_op_desc& da = e->ops[av.op].d;
            const int D = av.dst_t;
            const Tensor& td = e->tens[D];
            if (da.kh != 2 || da.kw != 2 || da.stride != 2 || da.pad != 0 || td.H != 2 * tp.H || td.W != 2 * tp.W || td.C != tp.C) continue;
            if (next_touch(i2 + 1, P) < st.size()) continue;          // nobody else wants the pooled gradient
            const size_t i3 = next_touch(i2 + 1, D);
            if (i3 >= st.size()) continue;
            const BwdStep& cv = st[i3];
            if (cv.kind != ST_CONV_BWD || cv.dst_t != D || !cv.accumulate || !cv.chain.empty()) continue;
            const xfr_op_desc& dc = e->ops[cv.op].d;
            if (dc.kh != 1 || dc.kw != 1 || dc.stride != 2 || dc.pad != 0 || dc.in0 != D || e->tens[dc.out].H != tp.H || e->tens[dc.out].W != tp.W) continue;
            const size_t i4 = next_touch(i3 + 1, D);
            if (i4 >= st.size()) continue;
            const BwdStep& ew = st[i4];
            if (ew.kind != ST_EW || ew.src_t != D || ew.dst_t == D || ew.accumulate || ew.ew_t != D || ew.chain.empty()) continue;
            if (ew.chain[0].type == EW_MAXHALF_IN || ew.chain[0].type == EW_POOL2_IN || ew.chain[0].type == EW_AVGUP_IN) continue;
            if ((int)ew.chain.size() + 1 > XFR_MAX_EW_STEPS) continue;
            bool bad = false;
            for (const Sym& y : ew.chain)
                if ((y.type == EW_STORE || y.type == EW_ADDP) && (y.t0 == D || y.t0 == S)) bad = true;
            if (next_touch(i4 + 1, D) < st.size()) {       // a later reader of D would want the tensor that is no longer written
                size_t k = next_touch(i4 + 1, D);
                if (reads(st[k], D)) bad = true;
            }
            for (size_t k = i0 + 1; k <= i4 && !bad; ++k)
                if (writes(st[k], S)) bad = true;            // S is now read where the chain runs
            if (bad) continue;
            hook.op = td.W;
            hook.slot = D;
            BwdStep f = ew;
            f.src_t = S;
            f.chain.insert(f.chain.begin(), hook);
            st[i4] = f;
            st[i3].compact = true;
            st[i3].accumulate = 0;
            // erase back to front
            st.erase(st.begin() + i2);
            if (i1 != i2) st.erase(st.begin() + i1);
            st.erase(st.begin() + i0);
            --i0;
        }
}

// Pass 2b': down-sampling residual block with a projection shortcut.
void fuse_downsample_projection(xfr_engine* e, std::vector<BwdStep>& st)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
        // ---- 2b'.  The same block input where the shortcut is a strided 1x1 projection (resnet50_128.py): ZERO D, CONV_BWD -> D (scatter), ...,
        // CONV_BWD -> D (scatter), EW D -> E.  Both GEMMs now work on the compact grid (the second accumulates there: dense rows), the zero fill
        // is gone and the chain's head puts the sum on the even pixels (EW_AVGUP_IN without a pooled source).
        for (size_t i0 = 0; i0 < st.size(); ++i0) {
            if (st[i0].kind != ST_ZERO || st[i0].dst_t < 0) continue;
            const int D = st[i0].dst_t;
            const Tensor& td = e->tens[D];
            auto next_touch = [&](size_t from, int t) {
                size_t k = from;
                for (; k < st.size(); ++k)
                    if (reads(st[k], t) || writes(st[k], t)) break;
                return k;
            };
            std::vector<size_t> gemms;
            size_t k = next_touch(i0 + 1, D);
            bool ok = true;
            int gh = -1, gw = -1;
            while (k < st.size() && st[k].kind == ST_CONV_BWD) {
                const BwdStep& cv = st[k];
                const xfr_op_desc& dc = e->ops[cv.op].d;
                const Tensor& tg = e->tens[dc.out];
                if (cv.dst_t != D || !cv.accumulate || !cv.chain.empty() || dc.kh != 1 || dc.kw != 1 || dc.stride != 2 || dc.pad != 0 || dc.in0 != D ||
                    td.H != 2 * tg.H || td.W != 2 * tg.W || (gh >= 0 && (gh != tg.H || gw != tg.W))) { ok = false; break; }
                gh = tg.H; gw = tg.W;
                gemms.push_back(k);
                k = next_touch(k + 1, D);
            }
            if (!ok || gemms.empty() || k >= st.size()) continue;
            const BwdStep& ew = st[k];
            if (ew.kind != ST_EW || ew.src_t != D || ew.dst_t == D || ew.accumulate || ew.ew_t != D || ew.chain.empty()) continue;
            if (ew.chain[0].type == EW_MAXHALF_IN || ew.chain[0].type == EW_POOL2_IN || ew.chain[0].type == EW_AVGUP_IN) continue;
            if ((int)ew.chain.size() + 1 > XFR_MAX_EW_STEPS) continue;
            bool bad = false;
            for (const Sym& y : ew.chain)
                if ((y.type == EW_STORE || y.type == EW_ADDP) && y.t0 == D) bad = true;
            {
                const size_t k2 = next_touch(k + 1, D);
                if (k2 < st.size() && reads(st[k2], D)) bad = true;
            }
            if (bad) continue;
            Sym head = mk(EW_AVGUP_IN, -1);
            head.action = -2;
            head.op = td.W;
            head.slot = D;
            st[k].chain.insert(st[k].chain.begin(), head);
            for (size_t q = 0; q < gemms.size(); ++q) {
                st[gemms[q]].compact = true;
                st[gemms[q]].accumulate = q == 0 ? 0 : 1;
            }
            st.erase(st.begin() + i0);
            --i0;
        }
}

// Pass 3b: first block of a stage, both Add operands continue from the same clamp.
void fuse_stage_head_relu(xfr_engine* e, std::vector<BwdStep>& st)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
        // ---- 3b (after the GEMM -> chain merges of pass 1).  First block of a stage: the GEMM that produces the gradient of the block's Add output ends [.., STORE(t), relu] -> D, where D (the
        // shortcut operand's gradient) and t (the main-path operand's) have different readers, and the main path's chain EW(t -> u) starts with the
        // same relu.  Both then continue from relu(v): the chain runs on in the GEMM's epilogue as [.., relu, STORE(D), rest] -> u -- the signature of
        // every other block's epilogue -- and the stand-alone launch is gone.
        for (size_t i = 0; i < st.size(); ++i) {
            BwdStep& a = st[i];
            if (a.kind != ST_CONV_BWD || scatter_conv(a) || a.compact || a.accumulate || a.chain.size() < 2 || a.dst_t < 0) continue;
            const size_t n = a.chain.size();
            const Sym r1 = a.chain[n - 1], s1 = a.chain[n - 2];
            auto plain_relu = [](const Sym& y) { return y.type == EW_HOOK && y.action == HOOK_RELU && !y.tap; };
            if (!plain_relu(r1) || s1.type != EW_STORE) continue;
            const int t = s1.t0, D = a.dst_t;
            if (t == D || t < 0) continue;
            size_t j = i + 1;
            for (; j < st.size(); ++j)
                if (reads(st[j], t) || writes(st[j], t)) break;
            if (j >= st.size()) continue;
            const BwdStep c = st[j];
            if (c.kind != ST_EW || c.src_t != t || c.accumulate || c.dst_t == t || c.dst_t == D || c.chain.empty() || !plain_relu(c.chain[0])) continue;
            if (e->tens[c.ew_t].C != e->tens[D].C || e->tens[c.ew_t].HW() != e->tens[D].HW()) continue;
            bool bad = false, other_readers = false;
            for (const Sym& y : c.chain)
                if ((y.type == EW_STORE || y.type == EW_ADDP) && (y.t0 == D || y.t0 == t)) bad = true;
            for (size_t k = j + 1; k < st.size(); ++k) {
                if (reads(st[k], t)) { other_readers = true; break; }
                if (writes(st[k], t)) break;
            }
            const int u = c.dst_t;
            for (size_t k = i + 1; k < j && !bad; ++k) {
                if (writes(st[k], u) || reads(st[k], u)) bad = true;
                for (const Sym& y : c.chain)
                    if (y.type == EW_ADDP && writes(st[k], y.t0)) bad = true;
            }
            if (bad || n - 2 + (other_readers ? 1 : 0) + 1 + c.chain.size() > XFR_MAX_EW_STEPS) continue;
            std::vector<Sym> merged(a.chain.begin(), a.chain.begin() + (n - 2));
            if (other_readers) merged.push_back(s1);
            merged.push_back(c.chain[0]);
            merged.push_back(mk(EW_STORE, D));
            merged.insert(merged.end(), c.chain.begin() + 1, c.chain.end());
            a.chain = merged;
            a.ew_t = D;
            a.dst_t = u;
            st.erase(st.begin() + j);
        }
}

// Pass 3c: first block of a stage with a projection shortcut, the main path's chain as a side branch.
void fuse_stage_head_branch(xfr_engine* e, std::vector<BwdStep>& st)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
        // ---- 3c.  First block of a stage with a PROJECTION shortcut (resnet50_128.py): the GEMM that produces the gradient of the block's Add output
        // ends [.., mask, STORE(t), rest_s] -> D: D, the shortcut branch's gradient, continues in the epilogue, and t, the Add output's gradient, is
        // stored for the main path, whose own hook chain EW(t -> u) = [rest_m] was a launch of its own.  Both chains start from the same value:
        // [.., mask, SAVE, rest_m, STORE(u) + RESTORE, rest_s] -> D runs the main path's chain as a side branch on the saved value -- the same
        // operations on the same operands, t is never written, the launch is gone.
        for (size_t i = 0; i < st.size(); ++i) {
            BwdStep& a = st[i];
            if (a.kind != ST_CONV_BWD || scatter_conv(a) || a.compact || a.accumulate || a.dst_t < 0 || a.chain.empty()) continue;
            int k = -1;
            bool plain = true;
            for (size_t q = 0; q < a.chain.size(); ++q) {
                const Sym& y = a.chain[q];
                if (y.type == EW_STORE && y.action == 0 && k < 0) k = (int)q;
                else if (y.type == EW_STORE && y.action != 0) plain = false;                 // one branch per chain
                if (y.type == EW_MAXHALF_OUT || y.type == EW_MAXPAIR || y.type == EW_ADDP_CO || y.type == EW_FORK_POSADD) plain = false;
            }
            if (k < 0 || !plain) continue;
            const int t = a.chain[k].t0, D = a.dst_t;
            if (t < 0 || t == D) continue;
            size_t j = i + 1;
            for (; j < st.size(); ++j)
                if (reads(st[j], t) || writes(st[j], t)) break;
            if (j >= st.size()) continue;
            const BwdStep c = st[j];
            if (c.kind != ST_EW || c.src_t != t || c.accumulate || c.dst_t < 0 || c.dst_t == t || c.dst_t == D || c.chain.empty()) continue;
            if (e->tens[c.ew_t].C != e->tens[a.ew_t >= 0 ? a.ew_t : D].C || e->tens[c.ew_t].HW() != e->tens[a.ew_t >= 0 ? a.ew_t : D].HW()) continue;
            const int u = c.dst_t;
            bool bad = false;
            for (const Sym& y : c.chain) {
                if (y.type != EW_HOOK && y.type != EW_MASK && y.type != EW_SCALE_C && y.type != EW_SCALE && y.type != EW_RELU) bad = true;   // plain per-element steps only
                if (y.type == EW_HOOK && y.tap) bad = true;
            }
            for (size_t q = j + 1; q < st.size() && !bad; ++q) {        // nobody else reads t before it is rewritten
                if (reads(st[q], t)) bad = true;
                if (writes(st[q], t)) break;
            }
            for (size_t q = i + 1; q < j && !bad; ++q)
                if (writes(st[q], u) || reads(st[q], u)) bad = true;
            for (const Sym& y : a.chain)
                if ((y.type == EW_STORE || y.type == EW_ADDP) && y.t0 == u) bad = true;
            if (bad || a.chain.size() + 1 + c.chain.size() > (size_t)XFR_MAX_EW_STEPS) continue;
            std::vector<Sym> merged(a.chain.begin(), a.chain.begin() + k);
            Sym save = mk(EW_STORE, -1);
            save.action = 1;
            merged.push_back(save);
            merged.insert(merged.end(), c.chain.begin(), c.chain.end());
            Sym back = mk(EW_STORE, u);
            back.action = 2;
            merged.push_back(back);
            merged.insert(merged.end(), a.chain.begin() + k + 1, a.chain.end());
            a.chain = merged;
            st.erase(st.begin() + j);
        }
}

// Passes 2 + 3 to a fixed point: chain -> chain merges (pass 0), GEMM -> chain merges (pass >= 1), the MaxFeatureMap fan-out (pass 2).
void fuse_merge_to_fixed_point(xfr_engine* e, std::vector<BwdStep>& st, int pass)
{
    typedef BwdStep::Sym Sym;
    const int nt = (int)e->tens.size();
    (void)nt; (void)sizeof(Sym);
    auto mk = fuse_mk;
    auto reads = fuse_reads;
    auto writes = fuse_writes;
    auto scatter_conv = [&](const BwdStep& b) { return b.kind == ST_CONV_BWD && e->ops[b.op].d.stride != 1; };
    (void)mk; (void)reads; (void)writes; (void)scatter_conv;
    bool changed = true;
    while (changed) {
        changed = false;
        for (size_t i = 0; i < st.size() && !changed; ++i) {
            BwdStep& a = st[i];
            const bool a_ew = a.kind == ST_EW;
            const bool a_conv = pass >= 1 && a.kind == ST_CONV_BWD && !scatter_conv(a);
            if (!a_ew && !a_conv) continue;
            const int b_t = a.dst_t;
            if (b_t < 0) continue;
            bool tap_inside = false;
            for (const Sym& y : a.chain) if (y.tap) tap_inside = true;
            if (tap_inside) continue;                       // the tap launch is the last one
            // next step that touches b_t
            size_t j = i + 1;
            for (; j < st.size(); ++j)
                if (reads(st[j], b_t) || writes(st[j], b_t)) break;
            if (j >= st.size()) continue;
            BwdStep& c = st[j];
            // an IN-PLACE chain on b_t (hooks flushed where the producer is glue) merges too: the merged launch simply ends in b_t
            bool inplace = c.kind == ST_EW && c.dst_t == b_t && !c.accumulate;
            // ... and so does a chain that stores its own intermediate value back into b_t on the way (a chain -> chain merge of an
            // in-place flush with its reader): that store is then the one b_t's later readers see
            bool restores = false;
            for (const Sym& y : c.chain) if (y.type == EW_STORE && y.t0 == b_t) { inplace = false; restores = true; }
            if (c.dst_t == b_t) restores = false;
            if (c.kind != ST_EW || c.src_t != b_t || (writes(c, b_t) && !inplace && !restores)) continue;
            // fan-out: GEMM (-> Co channels) followed by the chain whose head is the MaxFeatureMap VJP (over 2 * Co channels)
            bool fan = false;
            // ... or behind the chain launch whose head is the pool pair's VJP (1b): the Co-channel gradient between them never reaches HBM
            const bool a_pool = a_ew && !a.chain.empty() && a.chain[0].type == EW_POOL2_IN && !a.accumulate;
            if (pass == 2 && (a_conv || a_pool) && !c.chain.empty() && c.chain[0].type == EW_MAXHALF_IN && e->tens[c.ew_t].C == 2 * e->tens[b_t].C &&
                e->tens[c.ew_t].HW() == e->tens[b_t].HW() && (e->tens[b_t].HW() & 3) == 0) {
                fan = true;
                for (const Sym& y : a.chain) if (y.type == EW_MAXHALF_OUT) fan = false;
                for (size_t q = 1; q < c.chain.size(); ++q)
                    if (c.chain[q].type == EW_SCALE_C || c.chain[q].type == EW_AFFINE_C || c.chain[q].type == EW_FORK_POSBN) fan = false;   // per-channel
            }                                                                                                                        // parameters of row c
            if (!fan && (e->tens[c.ew_t].C != e->tens[b_t].C || e->tens[c.ew_t].HW() != e->tens[b_t].HW())) continue;
            // behind a fan-out the chain runs per half at channel c + h * Co, but the epilogue loads per-channel parameters at GEMM row
            // c: a chain with per-channel steps must not follow EW_MAXHALF_OUT (at the merge that creates the fan-out, above, or later)
            {
                bool a_fanned = false, c_perchan = false;
                for (const Sym& y : a.chain) if (y.type == EW_MAXHALF_OUT) a_fanned = true;
                for (const Sym& y : c.chain) if (y.type == EW_SCALE_C || y.type == EW_AFFINE_C || y.type == EW_FORK_POSBN) c_perchan = true;
                if (a_fanned && c_perchan) continue;
            }
            // does anything after j still read b_t?
            bool other_readers = false;
            for (size_t k = j + 1; k < st.size() && !inplace; ++k) {      // (in place: later readers want the chain's result, which is what stays)
                if (reads(st[k], b_t)) { other_readers = true; break; }
                if (writes(st[k], b_t)) break;
            }
            // the merged launch runs at position i: nothing in (i, j) may write c's destination or read/write what the
            // merged chain stores
            const int u = c.dst_t;
            bool blocked = false;
            for (size_t k = i + 1; k < j; ++k)
                if (writes(st[k], u) || reads(st[k], u)) blocked = true;
            // ADDP sources of c must be final before position i
            for (const Sym& y : c.chain)
                if (y.type == EW_ADDP)
                    for (size_t k = i; k < j; ++k)
                        if (writes(st[k], y.t0)) blocked = true;
            if (blocked) continue;
            std::vector<Sym> merged = a.chain;
            if (a.accumulate) {
                // a accumulates into b_t (partial sums already there): fold as an addend, then continue
                merged.push_back(mk(EW_ADDP, b_t));
            }
            if ((other_readers || a.accumulate) && !inplace && !restores) merged.push_back(mk(EW_STORE, b_t));
            if (fan) {
                Sym f = c.chain[0];
                f.type = EW_MAXHALF_OUT;
                merged.push_back(f);
                merged.insert(merged.end(), c.chain.begin() + 1, c.chain.end());
            } else {
                merged.insert(merged.end(), c.chain.begin(), c.chain.end());
            }
            if (c.accumulate) merged.push_back(mk(EW_ADDP, u));
            if ((int)merged.size() > XFR_MAX_EW_STEPS) continue;
            a.chain = merged;
            a.dst_t = u;
            a.accumulate = 0;
            if (a.kind == ST_CONV_BWD) a.ew_t = b_t;
            st.erase(st.begin() + j);
            changed = true;
        }
    }
}

// The fused schedules of a plan, built by the passes above in this order.  plan.fused: copy forwarding, the pool pair and chain -> chain merges only (what
// the observing sweeps run); plan.fused_gemm_nofan: + the down-sampling block rewrites and GEMM -> chain merges; plan.fused_gemm: + the MaxFeatureMap fan-out.
void fuse_plan(xfr_engine* e, BwdPlan& plan)
{
    std::vector<BwdStep> st = plan.steps;
    fuse_copy_forwarding(e, st);
    fuse_pool_pair(e, st);
    for (int pass = 0; pass < 3; ++pass) {
        if (pass == 1) plan.fused = st;
        if (pass == 1 && e->fuse_avgup) {
            fuse_downsample_avgpool(e, st);
            fuse_downsample_projection(e, st);
        }
        if (pass == 2 && e->fuse_avgup) fuse_stage_head_relu(e, st);
        if (pass == 2 && e->fuse_branch) fuse_stage_head_branch(e, st);
        if (pass == 2) plan.fused_gemm_nofan = st;
        fuse_merge_to_fixed_point(e, st, pass);
    }
    plan.fused_gemm.swap(st);
}

xfr_status get_plan(xfr_engine* e, int seed_tensor, BwdPlan** out, bool plain = false)
{
    for (auto& p : e->plans)
        if (p.seed_tensor == seed_tensor && p.mode == e->mode && p.plain == plain) { *out = &p; return XFR_OK; }
    e->plans.emplace_back();
    xfr_status st = make_plan(e, seed_tensor, e->plans.back(), plain);
    if (st != XFR_OK) { e->plans.pop_back(); return st; }
    if (!plain) fuse_plan(e, e->plans.back());
    *out = &e->plans.back();
    return XFR_OK;
}

// ---- the lean schedule ------------------------------------------------------------------------------------------------------------
// Every hook of a plain sweep (nothing observed: no trace, prior, capture or stored firing) needs less than its literal operands:
//   * a hook whose x IS its a (every Conv / Linear / pool / Concat / Add hook): a * relu(g) / (a + eps) is relu(g) where a > 0
//     and 0 where a = 0 -- one bit per element.  Where the tensor is the in-place ReLU output behind a lean BatchNorm, that bit is the sign bit of
//     the BatchNorm hook's stored quotient (HOOK_GATE_SIGN), otherwise the tensor itself is compared with 0 (HOOK_GATE);
//   * the BatchNorm hook (a = relu(W x + b), x = relu(relu(W) x + b)) and, in the modes that divide there, the in-place ReLU hook behind it: the
//     probe forward stored a / (x + eps) (fuse_probe_forward), the hook is relu(g) * q (HOOK_Q);
//   * ReLU masks and RELU-action hooks that the steps in front of them already imply are dropped.
// lean_prepare decides per plan (dry run of the probe forward, then a rewrite of plan.fused_gemm); the literal schedules stay what every
// observing call runs.
void lean_rewrite_chain(xfr_engine* e, const BwdPlan& plan, std::vector<BwdStep::Sym>& chain)
{
    typedef BwdStep::Sym Sym;
    std::vector<Sym> out;
    // ReLU-output roots whose positivity some lean BatchNorm quotient read by THIS chain carries in its sign bit
    auto sign_source = [&](int root) -> int {
        for (const Sym& y : chain)
            if (y.type == EW_HOOK && y.action == HOOK_DIV && !y.tap && y.x_t == y.t0 && y.t0 >= 0 && plan.lean_q[y.t0] == 1 && plan.lean_final[y.t0] == root) return y.t0;
        return -1;
    };
    bool nonneg = false;           // g >= 0 is known here
    int gated = -1;                // root r: g == 0 wherever T(r) <= 0 is known here
    // a lean hook clamps g itself: a plain clamp right in front of it is dropped
    auto drop_clamp = [&]() {
        if (!out.empty() && (out.back().type == EW_RELU || (out.back().type == EW_HOOK && out.back().action == HOOK_RELU && !out.back().tap))) out.pop_back();
    };
    for (const Sym& y : chain) {
        Sym z = y;
        switch (y.type) {
            case EW_HOOK: {
                if (y.tap) { out.push_back(z); nonneg = false; gated = -1; break; }       // P[-2]: p is stored, literal
                if (y.action == HOOK_DIV && y.x_t >= 0 && plan.lean_q[y.x_t] == 1 && y.x_t == y.t0) {
                    z.action = HOOK_Q; z.x_t = -1;                                        // the quotient sits in T(t0)
                    drop_clamp(); out.push_back(z); nonneg = true;
                } else if (y.action == HOOK_DIV && y.x_t >= 0 && plan.lean_q[y.x_t] == 2 && e->root(y.x_t) == e->root(y.t0)) {
                    z.action = HOOK_Q; z.t0 = y.x_t; z.x_t = y.x_t;                       // the quotient sits in Pv(x_t); zero exactly where the ReLU output is
                    drop_clamp(); out.push_back(z); nonneg = true; gated = e->root(y.t0);
                } else if (y.action == HOOK_DIV && y.x_t < 0) {
                    const int r = e->root(y.t0);
                    if (gated == r) { if (!nonneg) { z.type = EW_RELU; z.t0 = -1; out.push_back(z); nonneg = true; } break; }
                    const int c = sign_source(r);
                    if (c >= 0) { z.action = HOOK_GATE_SIGN; z.t0 = c; } else z.action = HOOK_GATE;
                    drop_clamp(); out.push_back(z); nonneg = true; gated = r;
                } else if (y.action == HOOK_RELU) {
                    if (!nonneg) { out.push_back(z); nonneg = true; }
                } else if (y.action == HOOK_PASS) {
                    // nothing observed, nothing returned: no step
                } else {
                    out.push_back(z); nonneg = (y.action == HOOK_DIV); gated = -1;        // a literal dividing hook (x from another tensor): p / (x + eps) >= 0
                }
                break;
            }
            case EW_MASK: {
                const int r = e->root(y.t0);
                if (gated == r) break;
                const int c = sign_source(r);
                if (c >= 0) { z.action = 1; z.t0 = c; }
                out.push_back(z); gated = r;
                break;
            }
            case EW_RELU: if (!nonneg) { out.push_back(z); nonneg = true; } break;
            case EW_SCALE_C: out.push_back(z); break;                                     // relu(gamma) * invstd >= 0: signs and zeros stay
            case EW_SCALE: out.push_back(z); if (!(y.f > 0.f)) { nonneg = false; gated = -1; } break;
            case EW_STORE: out.push_back(z); if (y.action == 2) { nonneg = false; gated = -1; } break;      // the restored value is the branch point's
            default: out.push_back(z); nonneg = false; gated = -1; break;                 // ADDP, chain heads, fan-outs: anything may follow
        }
    }
    chain.swap(out);
}

xfr_status fwd_op(xfr_engine* e, int k, int B, bool want_pos, hipStream_t s);
void lean_prepare(xfr_engine* e, BwdPlan& plan, int B)
{
    if (plan.lean_state >= 0) return;
    plan.lean_state = 0;
    if (plan.plain || plan.fused_gemm.empty() || !e->fuse_probe_fwd || !e->fuse_gemm_epilogue || e->interpret_chains) return;
    const int nt = (int)e->tens.size();
    const int last_op = e->tens[plan.seed_tensor].producer;
    // dry run of the probe forward: the same decisions the real one takes, nothing launched
    e->lean_q_run.assign(nt, 0);
    e->lean_final_run.assign(nt, -1);
    e->lean_decide = true;
    e->dry_run = true;
    e->lean_missing_sig = false;
    e->fwd_done.assign(e->ops.size(), 0);
    e->pos_done.assign(e->ops.size(), 0);
    e->fwd_last_op = last_op;
    for (int k = 0; k <= last_op; ++k) {
        const int kind = e->ops[k].d.kind;
        if (e->fwd_done[k] || (kind != XFR_OP_CONV && kind != XFR_OP_LINEAR)) continue;
        if (fwd_op(e, k, B, true, nullptr) != XFR_OK) e->lean_missing_sig = true;
    }
    e->lean_decide = false;
    e->dry_run = false;
    bool any = false;
    for (int t = 0; t < nt; ++t) any = any || e->lean_q_run[t] == 1;
    if (!any || e->lean_missing_sig) return;
    plan.lean_q = e->lean_q_run;
    plan.lean_final = e->lean_final_run;
    plan.fused_gemm_lean = plan.fused_gemm;
    for (BwdStep& b : plan.fused_gemm_lean)
        if (!b.chain.empty()) lean_rewrite_chain(e, plan, b.chain);
    plan.lean_state = 1;
}

// may this call take the lean schedule?  (B % 4: every lean epilogue is a float4 epilogue)
bool lean_applies(xfr_engine* e, BwdPlan& plan, int B)
{
    if (!e->lean || (B & 3) != 0 || e->trace_on || e->rc_priors || e->rc_caps || e->store_slot >= 0 || e->hold_forward || plan.plain) return false;
    if (!e->fuse_probe_fwd || !e->fuse_gemm_epilogue || e->interpret_chains) return false;
    lean_prepare(e, plan, B);
    return plan.lean_state == 1;
}

int prior_action_for(int mode, int kind)
{   // what the hook returns for a sample whose p was overridden by a prior (whitebox.py:396-428 with p_prior set)
    switch (mode) {
        case XFR_MODE_AFFINEONLY: return is_affine_name(kind) ? PRIOR_DIV : PRIOR_PASS;
        case XFR_MODE_AFFINEONLY_WITH_PRIOR: return is_affine_name(kind) ? PRIOR_DIV : PRIOR_GATEZ;
        case XFR_MODE_NORELU: return (kind == XFR_OP_MAXPOOL || kind == XFR_OP_RELU) ? PRIOR_PASS : PRIOR_DIV;
        default: return PRIOR_DIV;
    }
}

void resolve_chain(xfr_engine* e, const std::vector<BwdStep::Sym>& syms, EwChain& ch, double* trace, int SB, bool plain = false)
{
    ch.n = 0;
    for (const auto& sy : syms) {
        EwStep& q = ch.s[ch.n++];
        memset(&q, 0, sizeof(q));
        q.type = sy.type;
        q.action = sy.action;
        q.f = sy.f;
        q.prior_sb = -1;
        switch (sy.type) {
            case EW_HOOK:
                if (sy.action >= HOOK_Q) {        // lean hooks (lean_rewrite_chain): one source, nothing observed
                    q.p0 = (sy.action == HOOK_Q && sy.x_t >= 0) ? e->Pv(sy.x_t) : e->T(sy.t0);
                    break;
                }
                q.p0 = e->T(sy.t0);
                q.p1 = sy.x_t >= 0 ? e->Pv(sy.x_t) : nullptr;
                if (sy.tap) q.pstore = e->ws + e->tap_off;
                if (e->trace_on && sy.slot >= 0 && trace) q.trace = trace + (size_t)sy.slot * SB;
                if (sy.slot >= 0) {
                    if (sy.slot == e->store_slot && !sy.tap) q.pstore = e->store_dev;
                    if (e->rc_priors && sy.slot == e->rc_dense_slot) {
                        q.prior_sb = 0;
                        q.prior_dense = e->rc_prior_dense;
                        q.prior_action = prior_action_for(e->mode, e->ops[sy.op].d.kind);
                    } else if (e->rc_priors && sy.slot < (int)e->rc_prior_row.size() && e->rc_prior_row[sy.slot]) {
                        q.prior_elem = e->tab_elem_d + (size_t)sy.slot * e->tab_sb;
                        q.prior_val = e->tab_val_d + (size_t)sy.slot * e->tab_sb;
                        q.prior_action = prior_action_for(e->mode, e->ops[sy.op].d.kind);
                    }
                    if (e->rc_caps && sy.slot < (int)e->rc_cap_row.size() && e->rc_cap_row[sy.slot]) {
                        q.cap_elem = e->tab_elem_d + (size_t)sy.slot * e->tab_sb;
                        q.cap_dst = e->cap_dev + (size_t)sy.slot * e->tab_sb;
                    }
                }
                break;
            case EW_MASK: q.p0 = e->T(sy.t0); break;
            case EW_MAXHALF_IN: q.p0 = e->T(sy.t0); break;
            case EW_POOL2_IN: q.p0 = reinterpret_cast<const float*>(e->idx_base() + e->ops[sy.op].idx_off); break;      // the max-pool's argmax bytes
            case EW_AVGUP_IN:
                // sy.action: the pooled tensor's hook (or -1), sy.t0 / sy.x_t: its a / x tensors (-1: not observed / x == a), sy.op: full-res width,
                // sy.slot: the tensor whose gradient region holds the compact GEMM result (-1: none)
                q.p0 = sy.t0 >= 0 ? e->T(sy.t0) : nullptr;
                q.p1 = sy.x_t >= 0 ? e->Pv(sy.x_t) : nullptr;
                q.p2 = sy.slot >= 0 ? e->G(sy.slot) : nullptr;
                q.prior_sb = sy.op;
                break;
            case EW_MAXHALF_OUT: q.p0 = e->T(sy.t0); break;
            case EW_SCALE_C: q.p0 = e->arena + (plain ? e->ops[sy.op].bn_alpha_t : e->ops[sy.op].bn_alpha_p); break;
            case EW_STORE: q.pstore = sy.t0 >= 0 ? e->G(sy.t0) : nullptr; break;      // action 1 (save) has no destination
            case EW_ADDP: q.p0 = e->G(sy.t0); break;
            default: break;
        }
    }
}

// launch parameters of a backward-data GEMM step (with its fused chain resolved against the workspace)
void bwd_conv_params(xfr_engine* e, const BwdPlan& plan, const BwdStep& st, int B, int SB, int SBa, ConvParams& p)
{
    const OpRec& o = e->ops[st.op];
    const xfr_op_desc& d = o.d;
    const Tensor& a = e->tens[d.in0];
    const Tensor& t = e->tens[d.out];
    memset(&p, 0, sizeof(p));
    p.in = e->G(st.src_t);
    p.w = e->arena + (plan.plain ? o.w_bwd_true : o.w_bwd);
    p.out0 = e->G(st.dst_t);
    p.Cin = t.C; p.H = t.H; p.W = t.W; p.NB = SBa; p.in_nb = SB; p.out_nb = SB;
    p.tap_major = (d.stride == 1 && o.tap_bwd) ? 1 : 0;
    p.in_bytes = (unsigned)((size_t)SB * t.per_n() * sizeof(float));
    p.CoutTot = a.C; p.nhalves = 1; p.ldw = o.ldb;
    p.K = o.Kb;
    p.accumulate = st.accumulate;
    if (st.compact) {
        // 1x1 stride-s, result left on the sampled grid (dense rows of t.H x t.W per sample): EW_AVGUP_IN places it
        p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0;
        p.OH = t.H; p.OW = t.W;
        p.out_H = t.H; p.out_W = t.W; p.out_stride = 1;
        p.accumulate = st.accumulate;           // a second strided GEMM onto the same tensor adds to the first one's rows
        p.as_strided = 1;
    } else if (d.stride == 1) {
        // backward-data of a stride-1 convolution == convolution with the flipped, transposed kernel and padding k-1-p
        p.kh = d.kh; p.kw = d.kw; p.stride = 1; p.pad = d.kh - 1 - d.pad;
        p.OH = a.H; p.OW = a.W;
        p.out_H = a.H; p.out_W = a.W; p.out_stride = 1;
    } else {
        // 1x1 stride-s: the gradient lands on the sampled grid only
        p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0;
        p.OH = t.H; p.OW = t.W;
        p.out_H = a.H; p.out_W = a.W; p.out_stride = d.stride;
        p.accumulate = 1;   // the target was zero-filled or already holds other contributions
    }
    p.M = SBa * p.OH * p.OW;
    if (!st.chain.empty()) {
        resolve_chain(e, st.chain, p.chain, nullptr, SB);
        p.chain_B = B;
        p.chain_eps = e->eps;
        p.accumulate = 0;
        if (e->pair_tiles && SBa == 2 * B) p.pair_m = B * p.OH * p.OW;     // the two streams' tiles of one position side by side (ConvParams::pair_m)
    }
    p.chain_interpret = e->interpret_chains ? 1 : 0;
    p.bwd = 1;
}

// The fan-out schedule (plan.fused_gemm: MaxFeatureMap VJPs inside GEMM epilogues) only runs where every such epilogue is COMPILED --
// the interpreter has no fan-out step.  Every other fusion falls back to the interpreted epilogue; a network whose merged fan-out
// chain is not in chain_sigs.inc falls back to the schedule without fan-outs (plan.fused_gemm_nofan).  Decided once per plan: whether
// a chain has a compiled signature depends on the layer program and the mode, not on the batch (the fan-out requires HW % 4 == 0).
bool fanout_compiled(xfr_engine* e, BwdPlan& plan, int B, int SB)
{
    if (plan.fan_ok >= 0) return plan.fan_ok != 0;
    plan.fan_ok = 1;
    for (const BwdStep& st : plan.fused_gemm) {
        if (st.kind != ST_CONV_BWD) continue;
        bool fan = false;
        for (const auto& sy : st.chain) if (sy.type == EW_MAXHALF_OUT) fan = true;
        if (!fan) continue;
        ConvParams p;
        bwd_conv_params(e, plan, st, B, SB, SB, p);
        p.chain_interpret = 0;
        if (conv_gemm_cannot_launch(p)) { plan.fan_ok = 0; break; }
    }
    return plan.fan_ok != 0;
}

xfr_status run_backward(xfr_engine* e, BwdPlan& plan, int B, int S, hipStream_t s)
{
    const int SB = S * B;
    double* trace = e->dbl_ws + 2 * e->max_batch;
    if (e->trace_on) {
        HIP_TRY(hipMemsetAsync(trace, 0, sizeof(double) * (size_t)plan.n_firings * SB, s));
        e->last_trace_firings = plan.n_firings;
        e->last_trace_sb = SB;
        e->last_trace_kinds = plan.firing_kinds;
    }
    const bool special = e->rc_priors || e->rc_caps || e->store_slot >= 0;
    const bool use_fused = !e->trace_on && !plan.fused.empty() && !plan.plain;
    // Layerwise sweeps sorted by firing (rc_active): stream j is identically zero until the step that holds its prior
    // hook, so the GEMMs and hook chains before that step leave it out (the gradient region was zero-filled; the small
    // pool / copy kernels still run over all streams and move zeros).  SBa = streams alive at this step.
    const bool prefix = !e->rc_active.empty() && (int)e->rc_active.size() * e->rc_n == SB;
    int run_max = -1;
    const bool use_gemm_fusion = use_fused && e->fuse_gemm_epilogue && !special && !plan.fused_gemm.empty();
    const bool fanout = use_gemm_fusion && !e->interpret_chains && fanout_compiled(e, plan, B, SB);
    const bool lean = e->lean_cur == &plan && use_gemm_fusion && plan.lean_state == 1;
    if (e->lean_cur == &plan && !lean) return fail(XFR_STATE_ERROR, "lean schedule: the probe forward ran lean and the sweep cannot");
    // On-demand zeroing of the prefix sweeps (e->lazy_zero): wr[t] = leading rows of G(t) that hold defined values.  A launch that reads rows
    // [0, r) first gets the rows [wr[t], r) zeroed (one 2-D memset over the channels); launches that walk whole tensors or use another layout
    // (pool / copy VJPs, scattering and compact strided GEMMs, chain heads that expand a pooled gradient) get whole tensors.
    const bool lazy = prefix && e->lazy_zero;
    std::vector<int> wr;
    if (lazy) { wr.assign(e->tens.size(), 0); wr[plan.seed_tensor] = SB; }
    auto need = [&](int t, int rows) -> xfr_status {
        if (!lazy || t < 0 || wr[t] >= rows) return XFR_OK;
        const Tensor& x = e->tens[t];
        const size_t hw = (size_t)x.HW();
        HIP_TRY(hipMemset2DAsync(e->G(t) + (size_t)wr[t] * hw, (size_t)SB * hw * sizeof(float), 0, (size_t)(rows - wr[t]) * hw * sizeof(float), (size_t)x.C, s));
        wr[t] = rows;
        return XFR_OK;
    };
    auto wrote = [&](int t, int rows) { if (lazy && t >= 0 && wr[t] < rows) wr[t] = rows; };
    for (const BwdStep& st : (lean ? plan.fused_gemm_lean : use_gemm_fusion ? (fanout ? plan.fused_gemm : plan.fused_gemm_nofan) : use_fused ? plan.fused : plan.steps)) {
        int SBa = SB;
        if (prefix) {
            for (const auto& sy : st.chain)
                if (sy.type == EW_HOOK && sy.slot > run_max) run_max = sy.slot;
            SBa = (int)(std::upper_bound(e->rc_active.begin(), e->rc_active.end(), run_max) - e->rc_active.begin()) * e->rc_n;
            if (SBa == 0) continue;
        }
        if (lazy) {
            bool irregular = st.compact || !(st.kind == ST_EW || st.kind == ST_CONV_BWD);
            if (st.kind == ST_CONV_BWD && e->ops[st.op].d.stride != 1) irregular = true;
            for (const auto& sy : st.chain)
                if (sy.type == EW_AVGUP_IN || sy.type == EW_POOL2_IN || sy.type == EW_MAXHALF_IN || sy.type == EW_MAXHALF_OUT) irregular = true;
            const int ew_hw = st.kind == ST_EW ? e->tens[st.ew_t].HW() : 4;
            // rows this launch covers: the float4 chain kernel and the GEMMs honour the prefix, the scalar chain kernels walk every row
            const int rows = (irregular || st.kind == ST_ZERO || (st.kind == ST_EW && ((ew_hw & 3) != 0 || st.accumulate))) ? SB : SBa;
            xfr_status zs = XFR_OK;
            if (st.kind != ST_ZERO && zs == XFR_OK) zs = need(st.src_t, rows);
            if ((st.accumulate || irregular) && zs == XFR_OK) zs = need(st.dst_t, rows);
            for (const auto& sy : st.chain) {
                if (zs != XFR_OK) break;
                if (sy.type == EW_ADDP) zs = need(sy.t0, rows);
                else if (sy.type == EW_AVGUP_IN && sy.slot >= 0) zs = need(sy.slot, SB);
            }
            if (zs != XFR_OK) return zs;
            wrote(st.dst_t, rows);
            for (const auto& sy : st.chain)
                if (sy.type == EW_STORE && sy.action != 1) wrote(sy.t0, rows);
        }
        switch (st.kind) {
            case ST_EW: {
                EwChain ch;
                resolve_chain(e, st.chain, ch, trace, SB, plan.plain);
                const Tensor& x = e->tens[st.ew_t];
                launch_ew_chain(e->G(st.src_t), e->G(st.dst_t), st.accumulate, ch, x.C, SB, B, x.HW(), e->eps, s, SBa);
                break;
            }
            case ST_ZERO:
                launch_fill(e->G(st.dst_t), (long)SB * e->tens[st.dst_t].per_n(), 0.f, s);
                break;
            case ST_CONV_BWD: {
                ConvParams p;
                bwd_conv_params(e, plan, st, B, SB, SBa, p);
                xfr_status rs = run_conv(e, p, s);
                if (rs != XFR_OK) return rs;
                break;
            }
            case ST_MAXPOOL_BWD: {
                const OpRec& o = e->ops[st.op];
                const xfr_op_desc& d = o.d;
                const Tensor& a = e->tens[d.in0];
                const Tensor& t = e->tens[d.out];
                launch_maxpool_bwd(e->G(st.src_t), e->idx_base() + o.idx_off, e->G(st.dst_t), st.accumulate, a.C, SB, B, a.H, a.W, t.H,
                                   t.W, d.kh, d.stride, d.pad, s);
                break;
            }
            case ST_AVGPOOL_BWD: {
                const xfr_op_desc& d = e->ops[st.op].d;
                const Tensor& a = e->tens[d.in0];
                const Tensor& t = e->tens[d.out];
                launch_avgpool_bwd(e->G(st.src_t), e->G(st.dst_t), st.accumulate, a.C * SB, a.H, a.W, t.H, t.W, d.kh, d.stride, s);
                break;
            }
            case ST_COPY: {
                const Tensor& dt = e->tens[st.dst_t];
                launch_copy_acc(e->G(st.src_t), e->G(st.dst_t), (long)st.copy_elems_per_sb * SB * dt.HW(), st.accumulate, s);
                break;
            }
            case ST_MAXHALVES_BWD: {
                const xfr_op_desc& d = e->ops[st.op].d;
                const Tensor& t = e->tens[d.out];
                launch_maxhalves_bwd(e->G(st.src_t), e->T(d.in0), e->G(st.dst_t), st.accumulate, t.C, SB, B, t.HW(), s);
                break;
            }
            case ST_NORMALIZE_BWD: {
                const OpRec& o = e->ops[st.op];
                const xfr_op_desc& d = o.d;
                const Tensor& t = e->tens[d.out];
                launch_normalize_bwd(e->G(st.src_t), e->T(d.in0), e->misc() + o.norm_off, e->G(st.dst_t), st.accumulate,
                                     t.C, SB, B, s);
                break;
            }
        }
    }
    return XFR_OK;
}

xfr_status check_run(xfr_engine* e, const void* x, int n)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (!e->weights_loaded) return fail(XFR_STATE_ERROR, "weights not loaded");
    if (!x) return fail(XFR_INVALID_ARG, "null input");
    if (n < 1 || n > e->max_batch) return fail(XFR_INVALID_ARG, "batch %d outside [1, %d]", n, e->max_batch);
    HIP_TRY(hipSetDevice(e->device));
    if (e->need_dirty) compute_need(e);
    return XFR_OK;
}

void prof_begin(xfr_engine* e) { e->ev_used = 0; e->prof_flops = 0.0; }

xfr_status prof_end(xfr_engine* e, hipStream_t s)
{
    if (!e->profile_on) return XFR_OK;
    HIP_TRY(hipStreamSynchronize(s));
    double ms = 0.0;
    FILE* f = e->profile_csv.empty() ? nullptr : fopen(e->profile_csv.c_str(), "a");   // per-launch GEMM records (xfr_engine_profile_csv)
    for (int q = 0; q < 2; ++q) { e->fam_ms[q] = 0.0; e->fam_flops[q] = 0.0; e->fam_launches[q] = 0; }
    for (size_t i = 0; i < e->ev_used; ++i) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e->ev_pool[i].first, e->ev_pool[i].second));
        ms += t;
        const ConvParams& p = e->ev_params[i];
        const int Kl = p.K_logical ? p.K_logical : p.K;
        const double fl = 2.0 * Kl * (double)p.M * p.CoutTot * (p.dualacc ? 2 : p.nhalves);
        const int fam = e->ev_cfg[i] == 9 ? 1 : 0;
        e->fam_ms[fam] += t; e->fam_flops[fam] += fl; e->fam_launches[fam] += 1;
        if (f) fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.2f,%d\n", p.CoutTot, p.nhalves, Kl, p.M, p.kh, p.stride, p.out_stride,
                       p.relu_in, p.accumulate, t, fl / (t * 1e-3) / 1e12, e->ev_cfg[i]);
    }
    if (f) fclose(f);
    e->last_gemm_ms = ms;
    e->last_gemm_launches = (long)e->ev_used;
    e->last_gemm_flops = e->prof_flops;
    return XFR_OK;
}

// Any run that used forward slot 0 outside the pipelined paths must fence it, or a later pipelined forward (internal
// stream) could overwrite activations this run's kernels on `s` are still using.
xfr_status fence_slot0(xfr_engine* e, hipStream_t s)
{
    if (e->pipeline && e->ev_slot_done[0]) {
        HIP_TRY(hipEventRecord(e->ev_slot_done[0], s));
        e->slot_pending[0] = true;
    }
    return XFR_OK;
}

xfr_status ebp_core(xfr_engine* e, const float* x_dev, int n, int S, int seed_tensor, const float* seed_dev, hipStream_t s)
{
    // the one-shot promise of xfr_engine_set_inputs_ready covers THIS call, whatever becomes of it: consumed before the first check that
    // can fail, so an early error never leaves it standing for a later call that declared nothing
    const bool ready = e->inputs_ready;
    e->inputs_ready = false;
    if (seed_tensor < 2 || seed_tensor >= (int)e->tens.size()) return fail(XFR_INVALID_ARG, "bad seed tensor %d", seed_tensor);
    if (!seed_dev) return fail(XFR_INVALID_ARG, "null seed");
    BwdPlan* plan = nullptr;
    xfr_status st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    // pipeline level 2: the forward of this call runs on an internal stream and only waits for the slot it overwrites, so it
    // overlaps the backward sweep of the previous call (which is still reading the other slot)
    const bool pipe = e->pipeline_all && !e->profile_on && e->s_b;
    hipStream_t sf = pipe ? e->s_b : s;
    e->cur_slot = pipe ? (int)(e->seq++ % e->n_slots) : 0;
    const int slot = e->cur_slot;
    if (pipe && e->slot_pending[slot]) HIP_TRY(hipStreamWaitEvent(sf, e->ev_slot_done[slot], 0));
    // (the promise covers ONE call: a caller that forgets to renew it falls back to the safe ordering, never to a stale promise)
    if (pipe && !ready) {
        // x_dev may still be pending on the caller's stream (a cast, a copy): order the internal forward after it.  Callers
        // whose inputs are resident declare it with xfr_engine_set_inputs_ready and keep the cross-call overlap.
        HIP_TRY(hipEventRecord(e->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(sf, e->ev_fork, 0));
    }
    struct LeanGuard { xfr_engine* e; ~LeanGuard() { e->lean_cur = nullptr; } } lean_guard{e};
    e->lean_cur = lean_applies(e, *plan, n) ? plan : nullptr;
    st = forward_all(e, x_dev, n, seed_tensor, true, sf);
    if (st != XFR_OK) { e->cur_slot = 0; return st; }
    if (pipe) {
        HIP_TRY(hipEventRecord(e->ev_b, sf));
        HIP_TRY(hipStreamWaitEvent(s, e->ev_b, 0));
    }
    const Tensor& sd = e->tens[seed_tensor];
    launch_seed_to_cnhw(seed_dev, e->G(seed_tensor), S * n, sd.C, sd.HW(), s);
    st = run_backward(e, *plan, n, S, s);
    if (pipe && st == XFR_OK) {
        HIP_TRY(hipEventRecord(e->ev_slot_done[slot], s));
        e->slot_pending[slot] = true;
    } else if (st == XFR_OK) {
        st = fence_slot0(e, s);
    }
    e->cur_slot = 0;
    return st;
}

}  // namespace

// ===================================================================================================================
extern "C" {

int32_t xfr_abi_version(void) { return XFR_AMD_ABI_VERSION; }

const char* xfr_last_error(void) { return g_err.c_str(); }

xfr_status xfr_engine_create(const xfr_op_desc* ops, int32_t n_ops, int32_t n_weights, int32_t in_c, int32_t in_h, int32_t in_w,
                             int32_t max_batch, int32_t device, xfr_engine** out)
{
    if (!ops || n_ops < 2 || !out || in_c < 1 || in_h < 1 || in_w < 1 || max_batch < 1 || n_weights < 0)
        return fail(XFR_INVALID_ARG, "xfr_engine_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(XFR_HIP_ERROR, "no HIP device visible: the xfr_amd engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(XFR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    if (ops[0].kind != XFR_OP_CONV || ops[0].in0 != 0)
        return fail(XFR_UNSUPPORTED_LAYER, "the first layer must be a convolution on the input image");
    xfr_engine* e = new xfr_engine();
    e->device = device; e->max_batch = max_batch; e->in_c = in_c; e->in_h = in_h; e->in_w = in_w; e->n_weights = n_weights;
    xfr_status st = build(e, ops, n_ops);
    if (st == XFR_OK) st = layout_arena(e);
    if (st == XFR_OK) st = allocate(e);
    if (st != XFR_OK) { xfr_engine_destroy(e); return st; }
    if (const char* v = getenv("XFR_SPLIT_GEMM")) { e->split_mask = atoi(v) & 3; e->split_any_grid = (atoi(v) & 4) != 0; }      // A/B runs: the mode of xfr_engine_set_split_gemm for new engines
    *out = e;
    return XFR_OK;
}

xfr_status xfr_engine_destroy(xfr_engine* e)
{
    if (!e) return XFR_OK;
    (void)hipSetDevice(e->device);
    if (e->ws) (void)hipFree(e->ws);
    if (e->idx_ws) (void)hipFree(e->idx_ws);
    if (e->arena) { conv_gemm_forget_split(e->arena, e->arena_floats * sizeof(float)); (void)hipFree(e->arena); }
    if (e->dbl_ws) (void)hipFree(e->dbl_ws);
    if (e->trunc_ws) (void)hipFree(e->trunc_ws);
    if (e->ws_enc) (void)hipFree(e->ws_enc);
    for (int i = 0; i < e->n_tail_ws; ++i) (void)hipFree(e->tail_ws[i].ws);
    if (e->ws2) (void)hipFree(e->ws2);
    if (e->ws3) (void)hipFree(e->ws3);
    if (e->cap_dev) (void)hipFree(e->cap_dev);
    if (e->tab_elem_d) (void)hipFree(e->tab_elem_d);
    if (e->tab_val_d) (void)hipFree(e->tab_val_d);
    if (e->tab_elem_h) (void)hipHostFree(e->tab_elem_h);
    if (e->tab_val_h) (void)hipHostFree(e->tab_val_h);
    if (e->ev_tab) (void)hipEventDestroy(e->ev_tab);
    if (e->stat_v) (void)hipFree(e->stat_v);
    if (e->stat_i) (void)hipFree(e->stat_i);
    if (e->stat_scratch) (void)hipFree(e->stat_scratch);
    if (e->stat_desc) (void)hipFree(e->stat_desc);
    if (e->stat_f2u) (void)hipFree(e->stat_f2u);
    if (e->store_dev) (void)hipFree(e->store_dev);
    if (e->idx_ws2) (void)hipFree(e->idx_ws2);
    if (e->idx_ws3) (void)hipFree(e->idx_ws3);
    for (int i = 0; i < 3; ++i) { if (e->seedbuf[i]) (void)hipFree(e->seedbuf[i]); if (e->ev_slot_done[i]) (void)hipEventDestroy(e->ev_slot_done[i]); }
    for (int i = 0; i < 3; ++i) {
        if (e->u8_stage[i]) (void)hipFree(e->u8_stage[i]);
        if (e->ev_copied[i]) (void)hipEventDestroy(e->ev_copied[i]);
        if (e->ev_stage_a[i]) (void)hipEventDestroy(e->ev_stage_a[i]);
        if (e->ev_stage_b[i]) (void)hipEventDestroy(e->ev_stage_b[i]);
    }
    if (e->s_copy) (void)hipStreamDestroy(e->s_copy);
    if (e->s_a) (void)hipStreamDestroy(e->s_a);
    if (e->s_b) (void)hipStreamDestroy(e->s_b);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_a) (void)hipEventDestroy(e->ev_a);
    if (e->ev_b) (void)hipEventDestroy(e->ev_b);
    for (auto& ev : e->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete e;
    return XFR_OK;
}

static void presplit_weights(xfr_engine* e);

xfr_status xfr_engine_load_weights(xfr_engine* e, const xfr_tensor_view* w, int32_t n_weights)
{
    if (e) e->held_x = nullptr;
    if (!e || !w) return fail(XFR_INVALID_ARG, "null argument");
    if (n_weights != e->n_weights) return fail(XFR_INVALID_ARG, "expected %d weight views, got %d", e->n_weights, n_weights);
    HIP_TRY(hipSetDevice(e->device));
    std::vector<float> host(e->arena_floats, 0.f);
    for (size_t k = 0; k < e->ops.size(); ++k) {
        const OpRec& o = e->ops[k];
        const xfr_op_desc& d = o.d;
        if (d.kind == XFR_OP_CONV || d.kind == XFR_OP_LINEAR) {
            const xfr_tensor_view& wv = w[d.w_weight];
            const int khw = d.kh * d.kw;
            if (!wv.data || wv.numel != (int64_t)d.cout * o.Cin * khw)
                return fail(XFR_INVALID_ARG, "op %zu: weight has %lld elements, expected %lld", k, (long long)wv.numel,
                            (long long)d.cout * o.Cin * khw);
            float* wt = host.data() + o.w_true;
            float* wp = host.data() + o.w_pos;
            // forward pack: [k][co], k = (ci,kh,kw) or tap-major (kh,kw,ci)
            // column of output channel co in the forward pack: interleaved halves for MaxFeatureMap convolutions (OpRec::pair)
            auto col_of = [&](int co) { return o.pair ? (co % o.pair) * 2 + co / o.pair : co; };
            for (int co = 0; co < d.cout; ++co) {
                const float* src = wv.data + (size_t)co * o.K;
                const int col = col_of(co);
                for (int ci = 0; ci < o.Cin; ++ci)
                    for (int tp = 0; tp < khw; ++tp) {
                        const float v = src[ci * khw + tp];
                        const size_t kk = o.tap4_fwd ? (size_t)tp * 4 + ci : (o.tap_fwd ? (size_t)tp * o.Cin + ci : (size_t)ci * khw + tp);
                        wt[kk * o.ldw + col] = v;
                        wp[kk * o.ldw + col] = v > 0.f ? v : 0.f;   // relu(W): whitebox.py:319
                    }
            }
            if (o.w_bwd >= 0) {
                // backward-data pack of relu(W): [k' = (co, kh', kw')][ci] with the kernel flipped
                float* wb = host.data() + o.w_bwd;
                float* wbt = host.data() + o.w_bwd_true;
                for (int co = 0; co < d.cout; ++co)
                    for (int ci = 0; ci < o.Cin; ++ci)
                        for (int a = 0; a < d.kh; ++a)
                            for (int b = 0; b < d.kw; ++b) {
                                const float v = wv.data[(((size_t)co * o.Cin + ci) * d.kh + a) * d.kw + b];
                                const int a2 = d.kh - 1 - a, b2 = d.kw - 1 - b;
                                const size_t kk = (o.tap_bwd && d.stride == 1) ? (size_t)(a2 * d.kw + b2) * d.cout + co
                                                                               : (size_t)(co * d.kh + a2) * d.kw + b2;
                                wb[kk * o.ldb + ci] = v > 0.f ? v : 0.f;
                                wbt[kk * o.ldb + ci] = v;
                            }
            }
            if (d.w_bias >= 0) {
                const xfr_tensor_view& bv = w[d.w_bias];
                if (!bv.data || bv.numel != d.cout) return fail(XFR_INVALID_ARG, "op %zu: bad bias size", k);
                for (int co = 0; co < d.cout; ++co) {
                    host[o.b_true + col_of(co)] = bv.data[co];
                    host[o.b_pos + col_of(co)] = bv.data[co] > 0.f ? bv.data[co] : 0.f;   // whitebox.py:323 (with_bias)
                }
            }
        } else if (d.kind == XFR_OP_BATCHNORM) {
            const int C = e->tens[d.out].C;
            const xfr_tensor_view &g = w[d.w_weight], &b = w[d.w_bias], &m = w[d.w_mean], &v = w[d.w_var];
            if (!g.data || !b.data || !m.data || !v.data || g.numel != C || b.numel != C || m.numel != C || v.numel != C)
                return fail(XFR_INVALID_ARG, "op %zu: bad batchnorm parameter sizes", k);
            for (int c = 0; c < C; ++c) {
                // at::native inference batch norm: alpha = w * invstd, beta = b - mean * alpha
                const float invstd = 1.0f / sqrtf(v.data[c] + d.fparam);
                const float gp = g.data[c] > 0.f ? g.data[c] : 0.f;            // relu(gamma): whitebox.py:317-320
                const float bp = b.data[c] > 0.f ? b.data[c] : 0.f;
                const float at = g.data[c] * invstd, ap = gp * invstd;
                host[o.bn_alpha_t + c] = at;
                host[o.bn_beta_t + c] = b.data[c] - m.data[c] * at;
                host[o.bn_alpha_p + c] = ap;
                host[o.bn_beta_p + c] = b.data[c] - m.data[c] * ap;
                host[o.bn_beta_pb + c] = bp - m.data[c] * ap;
            }
        }
    }
    conv_gemm_forget_split(e->arena, e->arena_floats * sizeof(float));          // bf16 planes of the old weights (K17)
    HIP_TRY(hipMemcpy(e->arena, host.data(), e->arena_floats * sizeof(float), hipMemcpyHostToDevice));
    e->weights_loaded = true;
    presplit_weights(e);
    return XFR_OK;
}

// bf16 planes (K17) of every pack the bf16x6 kernel may be asked to run, built when the weights arrive instead of at a pack's first launch (round 5:
// the first step after a weight change stalled once per covered layer).  Layers, not launches: the geometry of one image decides.
static void presplit_weights(xfr_engine* e)
{
    if (!e->arena || !e->weights_loaded || !e->split_mask) return;        // (packs that have their planes keep them)
    for (size_t k = 0; k < e->ops.size(); ++k) {
        const OpRec& o = e->ops[k];
        const xfr_op_desc& d = o.d;
        if (d.kind != XFR_OP_CONV && d.kind != XFR_OP_LINEAR) continue;
        ConvParams p;
        conv_geometry(e, (int)k, 1, p);
        p.CoutTot = d.cout; p.nhalves = 1;
        if ((e->split_mask & 1) && conv_gemm_split_covers(p)) {
            (void)conv_gemm_presplit(p, e->arena + o.w_true, 0);
            (void)conv_gemm_presplit(p, e->arena + o.w_pos, 0);
        }
        if ((e->split_mask & 2) && k != 0 && d.stride == 1) {
            // the backward-data GEMM of a stride-1 convolution (bwd_conv_params): a convolution with the flipped, transposed pack
            const Tensor& a = e->tens[d.in0];
            const Tensor& t = e->tens[d.out];
            ConvParams q;
            memset(&q, 0, sizeof(q));
            q.Cin = t.C; q.H = t.H; q.W = t.W;
            q.kh = d.kh; q.kw = d.kw; q.stride = 1; q.pad = d.kh - 1 - d.pad;
            q.OH = a.H; q.OW = a.W; q.out_stride = 1;
            q.tap_major = o.tap_bwd ? 1 : 0;
            q.CoutTot = a.C; q.nhalves = 1; q.ldw = o.ldb; q.K = o.Kb;
            if (conv_gemm_split_covers(q)) {
                (void)conv_gemm_presplit(q, e->arena + o.w_bwd, 0);
                (void)conv_gemm_presplit(q, e->arena + o.w_bwd_true, 0);
            }
        }
    }
}

xfr_status xfr_engine_weight_arena(xfr_engine* e, void** dev_ptr, size_t* bytes)
{
    if (!e || !dev_ptr || !bytes) return fail(XFR_INVALID_ARG, "null argument");
    *dev_ptr = e->arena;              // (a caller that writes through it ends with xfr_engine_mark_weights_loaded, which rebuilds the bf16 planes of K17)
    *bytes = e->arena_floats * sizeof(float);
    return XFR_OK;
}

xfr_status xfr_engine_mark_weights_loaded(xfr_engine* e)
{
    if (e) e->held_x = nullptr;
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->weights_loaded = true;
    // the caller wrote the arena (through a pointer it may have held across forwards): planes built from the old contents are stale
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    conv_gemm_forget_split(e->arena, e->arena_floats * sizeof(float));
    presplit_weights(e);
    return XFR_OK;
}

xfr_status xfr_engine_set_mode(xfr_engine* e, int32_t subtree_mode, float eps, int32_t with_bias)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (subtree_mode < 0 || subtree_mode > 3) return fail(XFR_INVALID_ARG, "Invalid subtree mode %d", subtree_mode);
    if (!(eps >= 0.f)) return fail(XFR_INVALID_ARG, "eps must be >= 0");
    const int wb = with_bias ? 1 : 0;
    if (e->mode == subtree_mode && e->eps == eps && e->with_bias == wb) return XFR_OK;     // nothing to re-plan
    e->mode = subtree_mode; e->eps = eps; e->with_bias = wb;
    e->need_dirty = true;
    e->held_x = nullptr;
    return XFR_OK;
}

xfr_status xfr_engine_tensor_shape(xfr_engine* e, int32_t t, int32_t* c, int32_t* h, int32_t* w)
{
    if (!e || t < 0 || t >= (int)e->tens.size()) return fail(XFR_INVALID_ARG, "bad tensor id");
    if (c) *c = e->tens[t].C;
    if (h) *h = e->tens[t].H;
    if (w) *w = e->tens[t].W;
    return XFR_OK;
}

static xfr_status ensure_streams(xfr_engine* e);

xfr_status xfr_forward(xfr_engine* e, const float* x_dev, int32_t n, int32_t tensor_id, float* out_dev, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (tensor_id < 1 || tensor_id >= (int)e->tens.size() || !out_dev) return fail(XFR_INVALID_ARG, "bad tensor id / null output");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(e);
    // A forward-only batch (whitebox.py:747-785 embeddings; blackbox.py:366-414 scores ~6500 masked copies of a probe with it) as two half
    // batches on the two internal streams, like the gallery / probe pair of a triplet step: a layer's launches of the two halves fill each
    // other's prologues, epilogues and tails (round 3: one stream reached 0.52 of the fp32 MFMA peak).  The first half runs in the second
    // activation region (the one the triplet step's gallery forward uses); images are independent, the halves meet on the caller's stream.
    const int n0 = (n / 2) & ~3;
    if (e->split_forward && !e->profile_on && !e->hold_forward && n >= 32 && n0 >= 8) {
        if (!e->ws_enc) HIP_TRY(hipMalloc(&e->ws_enc, (e->t_region_floats + 4096) * sizeof(float)));
        st = ensure_streams(e);
        if (st != XFR_OK) return st;
        const Tensor& t = e->tens[tensor_id];
        const size_t in_per_n = (size_t)e->in_c * e->in_h * e->in_w;
        struct BankGuard { xfr_engine* e; ~BankGuard() { e->t_bank = nullptr; } } bank_guard{e};
        HIP_TRY(hipEventRecord(e->ev_fork, s));          // after everything already on the caller's stream (inputs, earlier sweeps)
        HIP_TRY(hipStreamWaitEvent(e->s_a, e->ev_fork, 0));
        HIP_TRY(hipStreamWaitEvent(e->s_b, e->ev_fork, 0));
        // whatever was enqueued on the internal streams (also by a half that then failed) is ordered before anything the caller puts on s next
        auto join = [&]() -> xfr_status {
            HIP_TRY(hipEventRecord(e->ev_a, e->s_a));
            HIP_TRY(hipEventRecord(e->ev_b, e->s_b));
            HIP_TRY(hipStreamWaitEvent(s, e->ev_a, 0));
            HIP_TRY(hipStreamWaitEvent(s, e->ev_b, 0));
            return XFR_OK;
        };
        e->t_bank = e->ws_enc;
        st = forward_all(e, x_dev, n0, tensor_id, false, e->s_a);
        if (st != XFR_OK) { const std::string why = g_err; join(); g_err = why; return st; }
        launch_cnhw_to_nchw(e->T(tensor_id), out_dev, n0, t.C, t.HW(), e->s_a);
        e->t_bank = nullptr;
        const float* x_hi = e->u8_on ? reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(x_dev) + (size_t)n0 * e->in_h * e->in_w * e->u8_pre.channels)
                                     : x_dev + (size_t)n0 * in_per_n;
        st = forward_all(e, x_hi, n - n0, tensor_id, false, e->s_b);
        if (st != XFR_OK) { const std::string why = g_err; join(); g_err = why; return st; }
        launch_cnhw_to_nchw(e->T(tensor_id), out_dev + (size_t)n0 * t.per_n(), n - n0, t.C, t.HW(), e->s_b);
        st = join();
        if (st != XFR_OK) return st;
        HIP_TRY(hipGetLastError());
        st = fence_slot0(e, s);
        if (st != XFR_OK) return st;
        return prof_end(e, s);
    }
    st = forward_all(e, x_dev, n, tensor_id, false, s);
    if (st != XFR_OK) return st;
    const Tensor& t = e->tens[tensor_id];
    launch_cnhw_to_nchw(e->T(tensor_id), out_dev, n, t.C, t.HW(), s);
    HIP_TRY(hipGetLastError());
    st = fence_slot0(e, s);
    if (st != XFR_OK) return st;
    return prof_end(e, s);
}

xfr_status xfr_ebp(xfr_engine* e, const float* x_dev, int32_t n, int32_t n_streams, int32_t seed_tensor, const float* seed_dev,
                   float* mwp_dev, float* pooled_dev, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (n_streams < 1 || n_streams > 2) return fail(XFR_INVALID_ARG, "n_streams must be 1 or 2");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(e);
    st = ebp_core(e, x_dev, n, n_streams, seed_tensor, seed_dev, s);
    if (st != XFR_OK) return st;
    const Tensor& t1 = e->tens[1];
    const int SB = n_streams * n;
    if (mwp_dev) launch_cnhw_to_nchw(e->ws + e->tap_off, mwp_dev, SB, t1.C, t1.HW(), s);
    if (pooled_dev) launch_channel_pool(e->ws + e->tap_off, pooled_dev, t1.C, SB, t1.HW(), s);
    HIP_TRY(hipGetLastError());
    return prof_end(e, s);
}

static xfr_status ensure_streams(xfr_engine* e)
{
    if (e->s_a) return XFR_OK;
    {
        // The internal streams run forwards -- in pipelined mode the NEXT call's -- while the caller's stream runs the backward
        // sweep whose maps the caller waits for: the forwards take the lowest priority, so the dispatcher prefers the sweep's
        // workgroups when both have some ready (measured on MI355X: 26.81 -> 26.69 ms per step; forwards at high priority: 27.3)
        int lowest = 0, highest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lowest, &highest));
        HIP_TRY(hipStreamCreateWithPriority(&e->s_a, hipStreamNonBlocking, lowest));
        HIP_TRY(hipStreamCreateWithPriority(&e->s_b, hipStreamNonBlocking, lowest));
    }
    HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_a, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_b, hipEventDisableTiming));
    return XFR_OK;
}

static xfr_status contrastive_tail(xfr_engine* e, int n, float percentile, float* sal_dev, hipStream_t s, bool raw = false)
{
    const Tensor& t1 = e->tens[1];
    const float* P = e->ws + e->tap_off;
    double* sums = e->dbl_ws;
    launch_sample_sums(P, sums, t1.C, 2 * n, t1.HW(), s);
    float* thr = nullptr;
    if (percentile >= 0.f) {
        thr = e->ws + e->thr_off;
        launch_truncation_threshold(P, sums, percentile, thr, e->trunc_ws, t1.C, n, t1.HW(), s);
    }
    float* contrast = raw ? sal_dev : e->ws + e->blur_a_off;
    launch_contrast(P, sums, thr, contrast, t1.C, n, t1.HW(), s);
    if (!raw) launch_saliency_blur(contrast, e->ws + e->blur_b_off, sal_dev, n, t1.H, t1.W, e->eps, s);
    HIP_TRY(hipGetLastError());
    return XFR_OK;
}

xfr_status xfr_contrastive(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                           float percentile, float* sal_dev, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (!sal_dev) return fail(XFR_INVALID_ARG, "null output");
    if (percentile > 100.f) return fail(XFR_INVALID_ARG, "percentile must be <= 100 (or < 0 for plain contrastive)");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(e);
    st = ebp_core(e, x_dev, n, 2, seed_tensor, seed_dev, s);
    if (st != XFR_OK) return st;
    st = contrastive_tail(e, n, percentile, sal_dev, s);
    if (st != XFR_OK) return st;
    return prof_end(e, s);
}

xfr_status xfr_contrastive_raw(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                               float percentile, float* contrast_dev, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (!contrast_dev) return fail(XFR_INVALID_ARG, "null output");
    if (percentile > 100.f) return fail(XFR_INVALID_ARG, "percentile must be <= 100 (or < 0 for plain contrastive)");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(e);
    st = ebp_core(e, x_dev, n, 2, seed_tensor, seed_dev, s);
    if (st != XFR_OK) return st;
    st = contrastive_tail(e, n, percentile, contrast_dev, s, true);
    if (st != XFR_OK) return st;
    return prof_end(e, s);
}

xfr_status xfr_triplet_contrastive(xfr_engine* e, const float* probes_dev, const float* gallery_dev, int32_t n,
                                   int32_t encode_tensor, float scale, float percentile, float* sal_dev, void* stream,
                                   int32_t inputs_ready)
{
    xfr_status st = check_run(e, probes_dev, n);
    if (st != XFR_OK) return st;
    if (!gallery_dev || !sal_dev) return fail(XFR_INVALID_ARG, "null argument");
    if (2 * n > e->max_batch) return fail(XFR_INVALID_ARG, "triplet batch %d needs max_batch >= %d (the gallery forward runs 2n images)", n, 2 * n);
    if (percentile > 100.f) return fail(XFR_INVALID_ARG, "percentile must be <= 100 (or < 0 for plain contrastive)");
    if (encode_tensor < 2 || encode_tensor >= (int)e->tens.size()) return fail(XFR_INVALID_ARG, "bad encode tensor %d", encode_tensor);
    hipStream_t s = (hipStream_t)stream;
    if (!e->ws_enc) HIP_TRY(hipMalloc(&e->ws_enc, (e->t_region_floats + 4096) * sizeof(float)));
    {
        xfr_status es = ensure_streams(e);
        if (es != XFR_OK) return es;
    }
    BwdPlan* plan = nullptr;
    st = get_plan(e, encode_tensor, &plan);
    if (st != XFR_OK) return st;
    prof_begin(e);
    // The gallery encodes (2n images, true weights only) and the probe forward (n images, W and relu(W)) are
    // independent until the backward sweep needs its seeds: run them on two streams so their small layer-3/4 grids
    // fill each other's idle CUs.  With profiling on, everything is serialised on the caller's stream instead.
    // In pipeline mode the forwards do not wait for the caller's stream at all (only for the slot they overwrite), so
    // the forward of call i+1 overlaps the backward sweep of call i.
    const bool fork = !e->profile_on;
    const bool pipe = fork && e->pipeline;
    hipStream_t sa = fork ? e->s_a : s, sb = fork ? e->s_b : s;
    // every exit path (errors included) leaves the engine on slot 0 / the main bank: the non-pipelined entry points assume it
    struct SlotGuard { xfr_engine* e; ~SlotGuard() { e->cur_slot = 0; e->t_bank = nullptr; } } slot_guard{e};
    e->cur_slot = pipe ? (int)(e->seq++ % e->n_slots) : 0;
    const int slot = e->cur_slot;
    // the engine's own staging (xfr_triplet_contrastive_u8_host): the inputs are complete when the copy's event fires -- nothing on the caller's stream
    // concerns them, so the forwards of a pipelined call need not wait for it (one-shot: consumed here, whatever becomes of the call)
    hipEvent_t in_ev = e->inputs_event;
    const int stage_slot = e->stage_slot;
    e->inputs_event = nullptr;
    e->stage_slot = -1;
    if (fork) {
        if (pipe && (inputs_ready || in_ev)) {
            if (e->slot_pending[slot]) {       // the backward that last read this slot must be done
                HIP_TRY(hipStreamWaitEvent(sa, e->ev_slot_done[slot], 0));
                HIP_TRY(hipStreamWaitEvent(sb, e->ev_slot_done[slot], 0));
            }
        } else {                               // order the forwards after everything already on the caller's stream
            HIP_TRY(hipEventRecord(e->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(sa, e->ev_fork, 0));
            HIP_TRY(hipStreamWaitEvent(sb, e->ev_fork, 0));
        }
        if (in_ev) {
            HIP_TRY(hipStreamWaitEvent(sa, in_ev, 0));
            HIP_TRY(hipStreamWaitEvent(sb, in_ev, 0));
        }
    } else if (in_ev) HIP_TRY(hipStreamWaitEvent(s, in_ev, 0));
    const Tensor& sd = e->tens[encode_tensor];
    float* seed_dst = pipe ? e->seedbuf[slot] : e->G(encode_tensor);
    e->t_bank = e->ws_enc;
    st = forward_all(e, gallery_dev, 2 * n, encode_tensor, false, sa);
    // seeds: stream 0 = scale * encode(mate_i), stream 1 = scale * encode(nonmate_i)  (demo/test_whitebox.py:129 with the
    // one-hot priors of whitebox.py:512,518 folded through the un-hooked 2-way classifier).  Layouts coincide:
    // both are [D][2n][1].
    if (st == XFR_OK) launch_scale(e->T(encode_tensor), seed_dst, (long)sd.per_n() * 2 * n, scale, 0, sa);
    e->t_bank = nullptr;
    if (st != XFR_OK) return st;
    struct LeanGuard { xfr_engine* e; ~LeanGuard() { e->lean_cur = nullptr; } } lean_guard{e};
    e->lean_cur = lean_applies(e, *plan, n) ? plan : nullptr;
    st = forward_all(e, probes_dev, n, encode_tensor, true, sb);
    if (st != XFR_OK) return st;
    if (fork) {
        HIP_TRY(hipEventRecord(e->ev_a, sa));
        HIP_TRY(hipEventRecord(e->ev_b, sb));
        HIP_TRY(hipStreamWaitEvent(s, e->ev_a, 0));
        HIP_TRY(hipStreamWaitEvent(s, e->ev_b, 0));
    }
    if (stage_slot >= 0) {                     // the staging buffer may be overwritten once both forwards have read it
        HIP_TRY(hipEventRecord(e->ev_stage_a[stage_slot], sa));
        HIP_TRY(hipEventRecord(e->ev_stage_b[stage_slot], sb));
        e->stage_busy[stage_slot] = true;
    }
    if (pipe) launch_copy_acc(seed_dst, e->G(encode_tensor), (long)sd.per_n() * 2 * n, 0, s);
    st = run_backward(e, *plan, n, 2, s);
    if (st != XFR_OK) return st;
    st = contrastive_tail(e, n, percentile, sal_dev, s);
    if (st != XFR_OK) return st;
    if (pipe) {
        HIP_TRY(hipEventRecord(e->ev_slot_done[slot], s));
        e->slot_pending[slot] = true;
    }
    e->cur_slot = 0;
    return prof_end(e, s);
}

xfr_status xfr_engine_set_u8_preprocess(xfr_engine* e, const xfr_u8_preprocess* p)
{
    if (!e || !p) return fail(XFR_INVALID_ARG, "null argument");
    if (p->kind == XFR_U8_SUB_MEAN) {
        if (p->channels != e->in_c || p->channels > 4) return fail(XFR_INVALID_ARG, "uint8 preprocessing: %d image channels for a %d-channel network input", p->channels, e->in_c);
    } else if (p->kind == XFR_U8_LUMINANCE) {
        if (p->channels != 3 || e->in_c != 1) return fail(XFR_INVALID_ARG, "uint8 luminance preprocessing takes 3-channel images into a 1-channel network input");
    } else return fail(XFR_INVALID_ARG, "unknown uint8 preprocessing kind %d", p->kind);
    e->u8_pre.kind = p->kind;
    e->u8_pre.channels = p->channels;
    for (int i = 0; i < 4; ++i) { e->u8_pre.mean[i] = p->mean[i]; e->u8_pre.weight[i] = p->weight[i]; }
    e->u8_set = true;
    e->held_x = nullptr;
    return XFR_OK;
}

namespace {
struct U8Guard { xfr_engine* e; explicit U8Guard(xfr_engine* e_) : e(e_) { e->u8_on = true; } ~U8Guard() { e->u8_on = false; } };
}

xfr_status xfr_forward_u8(xfr_engine* e, const uint8_t* x_u8_dev, int32_t n, int32_t tensor_id, float* out_dev, void* stream)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (!e->u8_set) return fail(XFR_STATE_ERROR, "xfr_engine_set_u8_preprocess has not been called");
    U8Guard g(e);
    return xfr_forward(e, reinterpret_cast<const float*>(x_u8_dev), n, tensor_id, out_dev, stream);
}

xfr_status xfr_triplet_contrastive_u8(xfr_engine* e, const uint8_t* probes_u8_dev, const uint8_t* gallery_u8_dev, int32_t n, int32_t encode_tensor, float scale,
                                      float percentile, float* sal_dev, void* stream, int32_t inputs_ready)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (!e->u8_set) return fail(XFR_STATE_ERROR, "xfr_engine_set_u8_preprocess has not been called");
    U8Guard g(e);
    return xfr_triplet_contrastive(e, reinterpret_cast<const float*>(probes_u8_dev), reinterpret_cast<const float*>(gallery_u8_dev), n, encode_tensor, scale,
                                   percentile, sal_dev, stream, inputs_ready);
}

// Fresh uint8 images in HOST memory, every call (demo/test_whitebox.py:124-133: every call brings new images): the engine copies them itself -- its own
// copy stream, one device staging buffer per forward slot -- and orders the forwards behind THAT copy instead of behind the caller's stream, so the
// copy, the preprocessing and the forwards of call i + 1 overlap the sweep of call i without the caller promising anything about residency.
xfr_status xfr_triplet_contrastive_u8_host(xfr_engine* e, const uint8_t* probes_u8_host, const uint8_t* gallery_u8_host, int32_t n, int32_t encode_tensor,
                                           float scale, float percentile, float* sal_dev, void* stream)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (!e->u8_set) return fail(XFR_STATE_ERROR, "xfr_engine_set_u8_preprocess has not been called");
    if (!probes_u8_host || !gallery_u8_host) return fail(XFR_INVALID_ARG, "null argument");
    if (n < 1 || 2 * n > e->max_batch) return fail(XFR_INVALID_ARG, "triplet batch %d needs max_batch >= %d (the gallery forward runs 2n images)", n, 2 * n);
    HIP_TRY(hipSetDevice(e->device));
    {
        xfr_status es = ensure_streams(e);
        if (es != XFR_OK) return es;
    }
    const size_t img = (size_t)e->u8_pre.channels * e->tens[0].HW();
    const size_t need = (size_t)(e->max_batch + e->max_batch / 2 + 1) * img;
    if (!e->s_copy) HIP_TRY(hipStreamCreateWithFlags(&e->s_copy, hipStreamNonBlocking));
    // the slot the call below will take (xfr_triplet_contrastive: seq % n_slots when pipelined)
    const int slot = (e->pipeline && !e->profile_on) ? (int)(e->seq % e->n_slots) : 0;
    if (!e->u8_stage[slot] || e->u8_stage_bytes < need) {
        for (int i = 0; i < 3; ++i) {
            if (e->u8_stage[i]) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(e->u8_stage[i]); e->u8_stage[i] = nullptr; e->stage_busy[i] = false; }
        }
        for (int i = 0; i < 3; ++i) {
            HIP_TRY(hipMalloc(&e->u8_stage[i], need));
            if (!e->ev_copied[i]) {
                HIP_TRY(hipEventCreateWithFlags(&e->ev_copied[i], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&e->ev_stage_a[i], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&e->ev_stage_b[i], hipEventDisableTiming));
            }
        }
        e->u8_stage_bytes = need;
    }
    if (e->stage_busy[slot]) {                 // the forwards that last read this staging buffer
        HIP_TRY(hipStreamWaitEvent(e->s_copy, e->ev_stage_a[slot], 0));
        HIP_TRY(hipStreamWaitEvent(e->s_copy, e->ev_stage_b[slot], 0));
    }
    uint8_t* gal = e->u8_stage[slot];
    uint8_t* pro = gal + (size_t)2 * n * img;
    HIP_TRY(hipMemcpyAsync(gal, gallery_u8_host, (size_t)2 * n * img, hipMemcpyHostToDevice, e->s_copy));
    HIP_TRY(hipMemcpyAsync(pro, probes_u8_host, (size_t)n * img, hipMemcpyHostToDevice, e->s_copy));
    HIP_TRY(hipEventRecord(e->ev_copied[slot], e->s_copy));
    e->last_copied = e->ev_copied[slot];
    e->inputs_event = e->ev_copied[slot];
    e->stage_slot = slot;
    U8Guard g(e);
    const xfr_status st = xfr_triplet_contrastive(e, reinterpret_cast<const float*>(pro), reinterpret_cast<const float*>(gal), n, encode_tensor, scale, percentile,
                                                  sal_dev, stream, 0);
    e->inputs_event = nullptr;                 // (an argument error returned before the call consumed them)
    e->stage_slot = -1;
    return st;
}

xfr_status xfr_engine_wait_inputs_copied(xfr_engine* e)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (e->last_copied) HIP_TRY(hipEventSynchronize(e->last_copied));
    return XFR_OK;
}

xfr_status xfr_debug_u8_preprocess(xfr_engine* e, const uint8_t* x_u8_dev, int32_t n, float* out_nchw_dev, void* stream)
{
    if (!e || !x_u8_dev || !out_nchw_dev) return fail(XFR_INVALID_ARG, "null argument");
    if (!e->u8_set) return fail(XFR_STATE_ERROR, "xfr_engine_set_u8_preprocess has not been called");
    if (n < 1 || n > e->max_batch) return fail(XFR_INVALID_ARG, "batch %d outside [1, %d]", n, e->max_batch);
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    const Tensor& in = e->tens[0];
    launch_u8hwc_to_cnhw(x_u8_dev, e->T(0), n, in.C, in.HW(), e->u8_pre, s);
    launch_cnhw_to_nchw(e->T(0), out_nchw_dev, n, in.C, in.HW(), s);
    HIP_TRY(hipGetLastError());
    e->held_x = nullptr;
    return fence_slot0(e, s);
}

xfr_status xfr_engine_set_epilogue_fusion(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->fuse_gemm_epilogue = (enable & 1) != 0;
    e->fuse_fwd_only = (enable & 1) != 0;
    e->fuse_probe_fwd = (enable & 2) != 0 && (enable & 4) == 0;     // a dual launch needs the compiled epilogue
    e->interpret_chains = (enable & 4) != 0;
    // bit 3 (tests): the max-pool + average-pool pair of Light-CNN keeps its separate forward kernels and VJP launches.  The backward
    // schedules are built with or without the pair's chain head: drop the cached ones when the switch moves.
    const bool pools = (enable & 1) != 0 && (enable & 8) == 0;
    if (pools != e->fuse_pools) e->plans.clear();
    e->fuse_pools = pools;
    {
        const bool avgup = (enable & 1) != 0 && (enable & 64) == 0;   // bit 6 (tests): the down-sampling blocks' shortcut VJP as separate launches
        if (avgup != e->fuse_avgup) e->plans.clear();
        e->fuse_avgup = avgup;
    }
    {
        const bool branch = (enable & 1) != 0 && (enable & 128) == 0;   // bit 7 (tests): the main path's chain of a projection-shortcut block as its own launch
        if (branch != e->fuse_branch) e->plans.clear();
        e->fuse_branch = branch;
    }
    e->pair_tiles = (enable & 32) == 0;           // bit 5 (A/B measurements): tile order of the two-stream backward GEMMs as before round 4
    e->hoist_shortcut = (enable & 256) == 0;      // bit 8 (tests, A/B): the down-sampling blocks' shortcut in program order, their residual add as its own launch
    e->direct_stem = (enable & 16) == 0;          // bit 4 (tests): the first layer of Light-CNN through the GEMM like every other convolution
    e->held_x = nullptr;
    for (auto& p : e->plans) p.lean_state = -1;   // the lean tables follow the probe forward's fusion decisions
    return XFR_OK;
}

xfr_status xfr_engine_hold_forward(xfr_engine* e, int32_t hold)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->hold_forward = hold != 0;
    e->held_x = nullptr;
    return XFR_OK;
}

xfr_status xfr_engine_set_tail_balance(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->tail_balance = enable != 0;
    return XFR_OK;
}

xfr_status xfr_engine_set_split_gemm(xfr_engine* e, int32_t mode)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (mode < 0 || mode > 7) return fail(XFR_INVALID_ARG, "xfr_engine_set_split_gemm: mode 0 (off), 1 (forward convolutions), 2 (backward-data GEMMs), 3 (both); + 4: whatever the launch's grid");
    e->split_mask = mode & 3;
    e->split_any_grid = (mode & 4) != 0;
    e->held_x = nullptr;
    presplit_weights(e);               // planes of the packs the new mode adds
    return XFR_OK;
}

xfr_status xfr_engine_split_gemm_stats(xfr_engine* e, int64_t* launches)
{
    if (!e || !launches) return fail(XFR_INVALID_ARG, "null argument");
    *launches = conv_gemm_split_launches();       // process-wide: the kernel's launch counter is not per engine
    return XFR_OK;
}

xfr_status xfr_engine_set_lean(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->lean = enable != 0;
    e->held_x = nullptr;
    return XFR_OK;
}

xfr_status xfr_engine_lean_stats(xfr_engine* e, int64_t* dual_launches)
{
    if (!e || !dual_launches) return fail(XFR_INVALID_ARG, "null argument");
    *dual_launches = e->lean_launches;
    return XFR_OK;
}

xfr_status xfr_engine_set_forward_split(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->split_forward = enable != 0;
    return XFR_OK;
}

xfr_status xfr_engine_set_pipeline(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const bool three = enable > 0 && (enable & 4) != 0;
    size_t max_per_n = 0;
    for (auto& x : e->tens) max_per_n = std::max(max_per_n, (size_t)x.per_n());
    auto slot_state = [&](int i) -> xfr_status {
        if (e->seedbuf[i]) return XFR_OK;
        HIP_TRY(hipMalloc(&e->seedbuf[i], 2 * (size_t)e->max_batch * max_per_n * sizeof(float)));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_slot_done[i], hipEventDisableTiming));
        return XFR_OK;
    };
    if (enable && !e->ws2) {
        HIP_TRY(hipMalloc(&e->ws2, e->fwd_region_floats * sizeof(float)));
        HIP_TRY(hipMalloc(&e->idx_ws2, e->idx_bytes));
    }
    if (three && !e->ws3) {
        HIP_TRY(hipMalloc(&e->ws3, e->fwd_region_floats * sizeof(float)));
        HIP_TRY(hipMalloc(&e->idx_ws3, e->idx_bytes));
    }
    if (enable)
        for (int i = 0; i < (three ? 3 : 2); ++i) { xfr_status ss = slot_state(i); if (ss != XFR_OK) return ss; }
    if (enable) { xfr_status es = ensure_streams(e); if (es != XFR_OK) return es; }
    e->pipeline = enable != 0;
    e->pipeline_all = enable > 0 && (enable & 2) != 0;
    e->n_slots = three ? 3 : 2;
    e->slot_pending[0] = e->slot_pending[1] = e->slot_pending[2] = false;
    e->seq = 0;
    return XFR_OK;
}

xfr_status xfr_mwp_to_saliency(xfr_engine* e, const float* pooled_dev, int32_t n, int32_t h, int32_t w, float* sal_dev, void* stream)
{
    if (!e || !pooled_dev || !sal_dev) return fail(XFR_INVALID_ARG, "null argument");
    if (n < 1 || (size_t)n * h * w > 2 * (size_t)e->max_batch * e->tens[1].HW())
        return fail(XFR_INVALID_ARG, "xfr_mwp_to_saliency: %d maps of %dx%d exceed the engine's scratch", n, h, w);
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    launch_saliency_blur(pooled_dev, e->ws + e->blur_b_off, sal_dev, n, h, w, e->eps, s);
    HIP_TRY(hipGetLastError());
    return XFR_OK;
}

// ---- "next" rows: layerwise / weighted-subtree EBP -----------------------------------------------------------------------
static xfr_status ensure_subtree_scratch(xfr_engine* e)
{
    if (e->cap_dev) return XFR_OK;
    const size_t nf = e->trace_cap + 1;
    size_t max_per_n = 0;
    for (auto& x : e->tens) max_per_n = std::max(max_per_n, (size_t)x.per_n());
    e->tab_cap = nf * 2 * (size_t)e->max_batch;
    HIP_TRY(hipMalloc(&e->cap_dev, e->tab_cap * sizeof(float)));
    HIP_TRY(hipMalloc(&e->tab_elem_d, e->tab_cap * sizeof(int)));
    HIP_TRY(hipMalloc(&e->tab_val_d, e->tab_cap * sizeof(float)));
    HIP_TRY(hipHostMalloc(&e->tab_elem_h, e->tab_cap * sizeof(int)));
    HIP_TRY(hipHostMalloc(&e->tab_val_h, e->tab_cap * sizeof(float)));
    HIP_TRY(hipEventCreateWithFlags(&e->ev_tab, hipEventDisableTiming));
    HIP_TRY(hipMalloc(&e->stat_v, nf * e->max_batch * sizeof(float)));
    HIP_TRY(hipMalloc(&e->stat_i, nf * e->max_batch * sizeof(int)));
    HIP_TRY(hipMalloc(&e->stat_scratch, subtree_stats_scratch_bytes(e->max_batch, (int)nf)));
    HIP_TRY(hipMalloc(&e->stat_desc, nf * sizeof(StatDesc)));
    HIP_TRY(hipMalloc(&e->stat_f2u, nf * sizeof(int)));
    HIP_TRY(hipMalloc(&e->store_dev, 2 * (size_t)e->max_batch * max_per_n * sizeof(float)));
    return XFR_OK;
}

xfr_status xfr_firing_count(xfr_engine* e, int32_t seed_tensor, int32_t* n_firings)
{
    if (!e || !n_firings) return fail(XFR_INVALID_ARG, "null argument");
    if (e->need_dirty) compute_need(e);
    BwdPlan* plan = nullptr;
    xfr_status st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    *n_firings = plan->n_firings;
    return XFR_OK;
}

xfr_status xfr_subtree_weights(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                               int32_t gate_ge0, float* w_host, int32_t* idx_host, int32_t capacity, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (!seed_dev || !w_host || !idx_host) return fail(XFR_INVALID_ARG, "null argument");
    st = ensure_subtree_scratch(e);
    if (st != XFR_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    BwdPlan* plan = nullptr;
    st = get_plan(e, seed_tensor, &plan, true);
    if (st != XFR_OK) return st;
    const int nf = plan->n_firings;
    if (capacity < nf * n) return fail(XFR_INVALID_ARG, "need room for %d x %d values", nf, n);
    st = forward_all(e, x_dev, n, seed_tensor, false, s);
    if (st != XFR_OK) return st;
    const Tensor& sd = e->tens[seed_tensor];
    launch_seed_to_cnhw(seed_dev, e->G(seed_tensor), 2 * n, sd.C, sd.HW(), s);
    e->rc_priors = e->rc_caps = false; e->store_slot = -1;
    st = run_backward(e, *plan, n, 2, s);
    if (st != XFR_OK) return st;
    if (e->stat_plan != plan) {       // descriptor table of this plan: one entry per distinct gradient tensor
        std::vector<StatDesc> desc;
        std::vector<int> f2u(nf);
        int last_t = -1;
        for (int f = 0; f < nf; ++f) {
            const int t = plan->firing_tensor[f];
            if (t != last_t) desc.push_back(StatDesc{e->G(t), e->tens[t].C, e->tens[t].HW()});   // several hooks on one tensor see the same gradient
            f2u[f] = (int)desc.size() - 1;
            last_t = t;
        }
        HIP_TRY(hipMemcpyAsync(e->stat_desc, desc.data(), desc.size() * sizeof(StatDesc), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(e->stat_f2u, f2u.data(), f2u.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));     // the host vectors go out of scope
        e->stat_plan = plan;
        e->stat_nu = (int)desc.size();
    }
    launch_subtree_stats(e->stat_desc, e->stat_nu, e->stat_f2u, nf, e->stat_v, e->stat_i, e->stat_scratch, n, gate_ge0, s);
    HIP_TRY(hipMemcpyAsync(w_host, e->stat_v, (size_t)nf * n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(idx_host, e->stat_i, (size_t)nf * n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return XFR_OK;
}

// stage the element / value tables of a call: the pinned host copies may only be rewritten once the previous call's
// host-to-device copy has completed
static xfr_status tables_begin(xfr_engine* e, int nf, int rows)
{
    if ((size_t)nf * rows > e->tab_cap) return fail(XFR_INVALID_ARG, "%d firings x %d gradient rows exceed the table scratch", nf, rows);
    HIP_TRY(hipEventSynchronize(e->ev_tab));
    e->tab_sb = rows;
    for (size_t i = 0; i < (size_t)nf * rows; ++i) { e->tab_elem_h[i] = -1; e->tab_val_h[i] = 0.f; }
    return XFR_OK;
}

static xfr_status tables_commit(xfr_engine* e, int nf, bool with_vals, hipStream_t s)
{
    const size_t n = (size_t)nf * e->tab_sb;
    HIP_TRY(hipMemcpyAsync(e->tab_elem_d, e->tab_elem_h, n * sizeof(int), hipMemcpyHostToDevice, s));
    if (with_vals) HIP_TRY(hipMemcpyAsync(e->tab_val_d, e->tab_val_h, n * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(e->ev_tab, s));
    return XFR_OK;
}

xfr_status xfr_ebp_capture(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                           const int32_t* elem_host, float* p_host, int32_t n_firings, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (!seed_dev || !elem_host || !p_host) return fail(XFR_INVALID_ARG, "null argument");
    st = ensure_subtree_scratch(e);
    if (st != XFR_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    BwdPlan* plan = nullptr;
    st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    if (n_firings != plan->n_firings) return fail(XFR_INVALID_ARG, "expected %d firings, got %d", plan->n_firings, n_firings);
    st = tables_begin(e, n_firings, n);
    if (st != XFR_OK) return st;
    e->rc_priors = false; e->store_slot = -1;
    e->rc_cap_row.assign(n_firings, 0);
    for (int f = 0; f < n_firings; ++f) {
        const Tensor& x = e->tens[plan->firing_tensor[f]];
        for (int b = 0; b < n; ++b) {
            const int el = elem_host[(size_t)f * n + b];
            if (el >= 0 && el < x.per_n()) { e->tab_elem_h[(size_t)f * n + b] = el; e->rc_cap_row[f] = 1; }
        }
    }
    st = tables_commit(e, n_firings, false, s);
    if (st != XFR_OK) return st;
    HIP_TRY(hipMemsetAsync(e->cap_dev, 0, (size_t)n_firings * n * sizeof(float), s));
    e->rc_caps = true;
    st = ebp_core(e, x_dev, n, 1, seed_tensor, seed_dev, s);
    e->rc_caps = false;
    if (st != XFR_OK) return st;
    HIP_TRY(hipMemcpyAsync(p_host, e->cap_dev, (size_t)n_firings * n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return XFR_OK;
}

xfr_status xfr_layerwise_ebp(xfr_engine* e, const float* x_dev, int32_t n, int32_t n_sweeps, int32_t seed_tensor,
                             const int32_t* firing_host, const int32_t* elem_host, const float* val_host,
                             const float* dense_prior_dev, float* pooled_dev, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (!firing_host || !pooled_dev) return fail(XFR_INVALID_ARG, "null argument");
    const long rows = (long)n_sweeps * n;
    if (n_sweeps < 1 || rows > 2L * e->max_batch)
        return fail(XFR_INVALID_ARG, "%d sweeps x %d images exceed the %d gradient rows of this engine", n_sweeps, n, 2 * e->max_batch);
    if (dense_prior_dev && rows != 1) return fail(XFR_INVALID_ARG, "a dense prior needs one sweep of one image");
    if (!dense_prior_dev && (!elem_host || !val_host)) return fail(XFR_INVALID_ARG, "null prior arrays");
    st = ensure_subtree_scratch(e);
    if (st != XFR_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    BwdPlan* plan = nullptr;
    st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    const int nf = plan->n_firings;
    for (long r = 0; r < rows; ++r)
        if (firing_host[r] >= nf || (firing_host[r] < 0 && dense_prior_dev))
            return fail(XFR_INVALID_ARG, "firing %d outside [0, %d)", firing_host[r], nf);
    st = tables_begin(e, nf, (int)rows);
    if (st != XFR_OK) return st;
    e->rc_caps = false; e->store_slot = -1;
    e->rc_prior_row.assign(nf, 0);
    e->rc_dense_slot = -1;
    e->rc_prior_dense = dense_prior_dev;
    if (dense_prior_dev) {
        e->rc_dense_slot = firing_host[0];
    } else {
        for (long r = 0; r < rows; ++r) {          // row r = sweep j * n + image b; firing < 0: an idle row (stays zero)
            const int f = firing_host[r];
            if (f < 0) continue;
            e->tab_elem_h[(size_t)f * rows + r] = elem_host[r];
            e->tab_val_h[(size_t)f * rows + r] = val_host[r];
            e->rc_prior_row[f] = 1;
        }
        st = tables_commit(e, nf, true, s);
        if (st != XFR_OK) return st;
    }
    // Sweeps handed over in ascending firing order (per image): sweep j -- identically zero above the earliest of its n priors --
    // only joins the GEMMs and hook chains from the step that holds that prior hook (run_backward: the launches cover a prefix
    // of the gradient rows)
    e->rc_active.clear();
    e->rc_n = n;
    std::vector<int> first(n_sweeps, nf);
    for (int j = 0; j < n_sweeps; ++j)
        for (int b = 0; b < n; ++b) { const int f = firing_host[(size_t)j * n + b]; if (f >= 0) first[j] = std::min(first[j], f); }
    bool ascending = n_sweeps > 1;
    for (int j = 1; j < n_sweeps; ++j) ascending = ascending && first[j] >= first[j - 1];
    if (ascending && !e->trace_on) e->rc_active = first;
    // one forward for all sweeps (whitebox.py:581 runs ebp(img, 0*P0) again for every layer); zero seeds: all the
    // gradient enters through the priors
    e->rc_priors = true;
    st = forward_all(e, x_dev, n, seed_tensor, true, s);
    if (st == XFR_OK) {
        const Tensor& sd = e->tens[seed_tensor];
        // Rows of a gradient tensor that no launch has written must read as zero (a sweep joins at its own firing).  The whole gradient region
        // used to be zero-filled here -- 30 GB at 256 rows, a third of a round; now run_backward zeroes exactly the rows a launch is about to
        // read and nobody has written (XFR_EAGER_ZERO=1: the old fill, for A/B runs; XFR_POISON_G=1, tests: NaN-fill first, so that a row the
        // bookkeeping misses shows up in the maps)
        static const bool eager = getenv("XFR_EAGER_ZERO") != nullptr, poison = getenv("XFR_POISON_G") != nullptr;
        e->lazy_zero = !e->rc_active.empty() && !eager;
        if (!e->rc_active.empty() && (eager || poison))
            HIP_TRY(hipMemsetAsync(e->ws + e->g_begin, eager ? 0 : 0xFF, (e->g_end - e->g_begin) * sizeof(float), s));
        launch_fill(e->G(seed_tensor), (long)sd.per_n() * rows, 0.f, s);
        st = run_backward(e, *plan, n, n_sweeps, s);
    }
    e->rc_active.clear();
    e->lazy_zero = false;
    e->rc_priors = false;
    e->rc_prior_dense = nullptr;
    e->rc_dense_slot = -1;
    if (st != XFR_OK) return st;
    const Tensor& t1 = e->tens[1];
    launch_channel_pool(e->ws + e->tap_off, pooled_dev, t1.C, (int)rows, t1.HW(), s);
    HIP_TRY(hipGetLastError());
    return fence_slot0(e, s);
}

xfr_status xfr_ebp_store_firing(xfr_engine* e, const float* x_dev, int32_t n, int32_t seed_tensor, const float* seed_dev,
                                int32_t firing, float* out_dev, int32_t* c, int32_t* h, int32_t* w, void* stream)
{
    xfr_status st = check_run(e, x_dev, n);
    if (st != XFR_OK) return st;
    if (!seed_dev) return fail(XFR_INVALID_ARG, "null seed");
    st = ensure_subtree_scratch(e);
    if (st != XFR_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    BwdPlan* plan = nullptr;
    st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    if (firing < 0 || firing > plan->n_firings) return fail(XFR_INVALID_ARG, "firing %d outside [0, %d]", firing, plan->n_firings);
    if (firing == plan->n_firings) {
        // the image hook, Whitebox.P[-1]: one standard sweep leaves the gradient of the first convolution's output (after its hooks) in G(1);
        // its backward-data pass with relu(W) and the hook p = relu(image) * relu(z) run as one gather kernel (nothing on the path reads this)
        if (c) *c = e->in_c;
        if (h) *h = e->in_h;
        if (w) *w = e->in_w;
        if (!out_dev) return XFR_OK;
        e->rc_priors = e->rc_caps = false;
        e->store_slot = -1;
        {
            // un-pipelined on purpose: the gather below reads the image from forward slot 0 on the caller's stream; a pipelined call would have
            // put it into slot seq % n_slots on an internal stream (and ebp_core resets cur_slot before it returns)
            struct Unpipe { xfr_engine* e; bool was; ~Unpipe() { e->pipeline_all = was; } } unpipe{e, e->pipeline_all};
            e->pipeline_all = false;
            st = ebp_core(e, x_dev, n, 1, seed_tensor, seed_dev, s);
        }
        if (st != XFR_OK) return st;
        const OpRec& o = e->ops[0];
        const xfr_op_desc& d = o.d;
        const Tensor& t1 = e->tens[d.out];
        launch_image_mwp(e->G(d.out), e->arena + o.w_pos, e->T(0), out_dev, e->in_c, n, e->in_h, e->in_w, d.cout, t1.H, t1.W, d.kh, d.kw, d.stride,
                         d.pad, o.ldw, o.tap4_fwd ? 2 : (o.tap_fwd ? 1 : 0), o.pair, s);
        HIP_TRY(hipGetLastError());
        return fence_slot0(e, s);           // the gather still reads slot 0: a later pipelined forward into it waits for THIS point
    }
    const Tensor& x = e->tens[plan->firing_tensor[firing]];
    if (c) *c = x.C;
    if (h) *h = x.H;
    if (w) *w = x.W;
    if (!out_dev) return XFR_OK;          // shape query
    e->rc_priors = e->rc_caps = false;
    const bool is_tap = (firing == plan->n_firings - 1);
    e->store_slot = is_tap ? -1 : firing;
    st = ebp_core(e, x_dev, n, 1, seed_tensor, seed_dev, s);
    e->store_slot = -1;
    if (st != XFR_OK) return st;
    launch_cnhw_to_nchw(is_tap ? e->ws + e->tap_off : e->store_dev, out_dev, n, x.C, x.HW(), s);
    HIP_TRY(hipGetLastError());
    return XFR_OK;
}

xfr_status xfr_firing_kinds(xfr_engine* e, int32_t seed_tensor, int32_t* kinds, int32_t capacity)
{
    if (!e || !kinds) return fail(XFR_INVALID_ARG, "null argument");
    if (e->need_dirty) compute_need(e);
    BwdPlan* plan = nullptr;
    xfr_status st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    if (capacity < plan->n_firings) return fail(XFR_INVALID_ARG, "need room for %d kinds", plan->n_firings);
    for (int i = 0; i < plan->n_firings; ++i) kinds[i] = plan->firing_kinds[i];
    return XFR_OK;
}

xfr_status xfr_engine_set_inputs_ready(xfr_engine* e, int32_t ready)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->inputs_ready = ready != 0;
    return XFR_OK;
}

xfr_status xfr_engine_set_trace(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->trace_on = enable ? 1 : 0;
    return XFR_OK;
}

xfr_status xfr_engine_trace_size(xfr_engine* e, int32_t* n_firings)
{
    if (!e || !n_firings) return fail(XFR_INVALID_ARG, "null argument");
    *n_firings = e->last_trace_firings;
    return XFR_OK;
}

xfr_status xfr_engine_get_trace(xfr_engine* e, double* sums, int32_t* kinds, int32_t capacity)
{
    if (!e || !sums) return fail(XFR_INVALID_ARG, "null argument");
    const int nf = e->last_trace_firings, SB = e->last_trace_sb;
    if (capacity < nf * SB) return fail(XFR_INVALID_ARG, "trace needs %d doubles", nf * SB);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(sums, e->dbl_ws + 2 * e->max_batch, sizeof(double) * (size_t)nf * SB, hipMemcpyDeviceToHost));
    if (kinds) for (int i = 0; i < nf; ++i) kinds[i] = e->last_trace_kinds[i];
    return XFR_OK;
}

xfr_status xfr_debug_conv(const float* in_dev, const float* w_host, const float* bias_host, float* out_dev, int32_t cin, int32_t h,
                          int32_t w, int32_t nb, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t relu_in,
                          int32_t cfg, int32_t reps, float* ms_out)
{
    if (!in_dev || !w_host || !out_dev || cin < 1 || cout < 1 || kh < 1 || kw < 1 || stride < 1 || reps < 1)
        return fail(XFR_INVALID_ARG, "xfr_debug_conv: bad arguments");
    const int khw = kh * kw, K = cin * khw;
    const int ldw = (int)align_up(cout, 128);
    const bool tap = khw > 1 && (cin % 16 == 0) && khw <= 64;
    const bool tap4 = khw > 1 && (cin == 3 || cin == 4) && khw <= 60;
    const int Kf = tap4 ? 4 * khw : K;
    std::vector<float> host(align_up(Kf, 32) * (size_t)ldw, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tp = 0; tp < khw; ++tp) {
                const size_t kk = tap4 ? (size_t)tp * 4 + ci : (tap ? (size_t)tp * cin + ci : (size_t)ci * khw + tp);
                host[kk * ldw + co] = w_host[((size_t)co * cin + ci) * khw + tp];
            }
    float *wd = nullptr, *bd = nullptr;
    HIP_TRY(hipMalloc(&wd, host.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(wd, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    if (bias_host) {
        HIP_TRY(hipMalloc(&bd, cout * sizeof(float)));
        HIP_TRY(hipMemcpy(bd, bias_host, cout * sizeof(float), hipMemcpyHostToDevice));
    }
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = in_dev; p.w = wd; p.bias = bd; p.out0 = out_dev;
    p.Cin = cin; p.H = h; p.W = w; p.NB = nb; p.in_nb = nb; p.out_nb = nb;
    p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad;
    p.OH = (h + 2 * pad - kh) / stride + 1; p.OW = (w + 2 * pad - kw) / stride + 1;
    p.K = Kf; p.K_logical = K; p.M = nb * p.OH * p.OW; p.CoutTot = cout; p.nhalves = 1; p.ldw = ldw;
    p.relu_in = relu_in; p.out_H = p.OH; p.out_W = p.OW; p.out_stride = 1;
    p.in_bytes = (unsigned)((size_t)cin * nb * h * w * sizeof(float));
    p.tap_major = tap4 ? 2 : (tap ? 1 : 0); p.force_cfg = cfg % 100;
    float* tws = nullptr;
    HIP_TRY(hipMalloc(&tws, XFR_TAIL_WS_BYTES + XFR_TAIL_MAX_TILES * sizeof(unsigned)));
    p.tail_ws = tws; p.tail_ws_bytes = XFR_TAIL_WS_BYTES;
    p.tail_cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(tws) + XFR_TAIL_WS_BYTES);
    HIP_TRY(hipMemset(p.tail_cnt, 0, XFR_TAIL_MAX_TILES * sizeof(unsigned)));
    p.tail_force = (cfg / 10000) % 100;  // 0 heuristic, 1 off, S >= 2 forced
    const bool split = (cfg % 100) == 9;  // the bf16x6 kernel (layers it does not cover run the fp32 kernel the rules give, like in the engine)
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    const int nstreams = std::max(1, std::min(4, cfg / 1000000));     // > 1: the same launches on several streams at once
    p.tail_force = (cfg / 10000) % 100;
    // untimed warm-up: as many launches as are timed (at most 200).  The allocations and copies above left the device idle; one launch
    // does not bring the clocks back, and the first configuration of a sweep row then reads 5-12 % low (round 4: the same kernel measured
    // first and third in a row)
    for (int r = 0; r < std::max(1, std::min(reps, 200)); ++r) launch_conv_gemm(p, 0);
    float ms = 0.f;
    if (nstreams == 1) {
        HIP_TRY(hipEventRecord(a, 0));
        for (int r = 0; r < reps; ++r) launch_conv_gemm(p, 0);
        HIP_TRY(hipEventRecord(b, 0));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
    } else {
        // every stream needs its own tail-balancing scratch; the outputs coincide (same values)
        hipStream_t st[4];
        float* tw[4];
        ConvParams q[4];
        for (int i = 0; i < nstreams; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            HIP_TRY(hipMalloc(&tw[i], XFR_TAIL_WS_BYTES + XFR_TAIL_MAX_TILES * sizeof(unsigned)));
            q[i] = p;
            q[i].tail_ws = tw[i];
            q[i].tail_cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(tw[i]) + XFR_TAIL_WS_BYTES);
            HIP_TRY(hipMemset(q[i].tail_cnt, 0, XFR_TAIL_MAX_TILES * sizeof(unsigned)));
        }
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipEventRecord(a, 0));
        for (int i = 0; i < nstreams; ++i) HIP_TRY(hipStreamWaitEvent(st[i], a, 0));
        for (int r = 0; r < reps; ++r)
            for (int i = 0; i < nstreams; ++i) launch_conv_gemm(q[i], st[i]);
        for (int i = 0; i < nstreams; ++i) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(e, st[i]));
            HIP_TRY(hipStreamWaitEvent(0, e, 0));
            (void)hipEventDestroy(e);
        }
        HIP_TRY(hipEventRecord(b, 0));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipEventElapsedTime(&ms, a, b));
        ms /= nstreams;                       // per launch, all streams' launches counted
        for (int i = 0; i < nstreams; ++i) { (void)hipStreamDestroy(st[i]); (void)hipFree(tw[i]); }
    }
    if (ms_out) *ms_out = ms / reps;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (split) conv_gemm_forget_split(wd, host.size() * sizeof(float));
    (void)hipFree(wd);
    (void)hipFree(tws);
    if (bd) (void)hipFree(bd);
    HIP_TRY(hipGetLastError());
    return XFR_OK;
}

xfr_status xfr_debug_conv_stamps(void* stamps_dev, int32_t capacity_workgroups)
{
    if (stamps_dev && (capacity_workgroups == 0 || (capacity_workgroups < 0 && -capacity_workgroups < 256)))
        return fail(XFR_INVALID_ARG, "xfr_debug_conv_stamps: capacity must be positive (or <= -256: sampled mode)");
    conv_gemm_set_stamps(reinterpret_cast<unsigned long long*>(stamps_dev), capacity_workgroups);
    return XFR_OK;
}

xfr_status xfr_debug_conv_log(void* log_dev, int32_t capacity, const char* dump_path)
{
    if (dump_path) {
        const int n = conv_gemm_dump_log(dump_path);
        if (n < 0) return fail(XFR_HIP_ERROR, "xfr_debug_conv_log: cannot write %s", dump_path);
    }
    conv_gemm_set_log(reinterpret_cast<unsigned long long*>(log_dev), capacity);
    return XFR_OK;
}

// ---- multi-GPU: RCCL behind the C ABI -----------------------------------------------------------
// One process per GPU; the only collective of the path is the one-off broadcast of the packed parameter arena.  librccl is
// bound at first use (dlopen), so the library loads -- and every other entry point works -- where RCCL is absent.
namespace {
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void* /* ncclUniqueId by value: 128 bytes, passed in memory */, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
struct UniqueId { char b[128]; };     // layout of ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
Rccl g_rccl;

xfr_status rccl_bind()
{
    if (g_rccl.h) return XFR_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(XFR_RCCL_ERROR, "cannot load librccl: %s", dlerror());
    Rccl r;
    r.h = h;
    r.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(h, "ncclBroadcast"));
    r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.Broadcast || !r.CommDestroy || !r.GetErrorString)
        return fail(XFR_RCCL_ERROR, "librccl lacks an expected symbol");
    g_rccl = r;
    return XFR_OK;
}
#define RCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        int _r = (expr);                                                                                        \
        if (_r != 0) return fail(XFR_RCCL_ERROR, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));           \
    } while (0)
}  // namespace

struct xfr_comm {
    void* comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

xfr_status xfr_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(XFR_INVALID_ARG, "null argument");
    xfr_status st = rccl_bind();
    if (st != XFR_OK) return st;
    RCCL_TRY(g_rccl.GetUniqueId(id_out));
    return XFR_OK;
}

xfr_status xfr_comm_init(int32_t rank, int32_t world, const void* unique_id, int32_t device, xfr_comm** out)
{
    if (!unique_id || !out || world < 1 || rank < 0 || rank >= world) return fail(XFR_INVALID_ARG, "xfr_comm_init: bad arguments");
    xfr_status st = rccl_bind();
    if (st != XFR_OK) return st;
    HIP_TRY(hipSetDevice(device));
    xfr_comm* c = new xfr_comm();
    c->rank = rank; c->world = world; c->device = device;
    UniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    // ncclCommInitRank takes the 128-byte id BY VALUE; on x86-64 a struct of that size is passed in memory, which is what
    // a function pointer declared with the same struct type produces
    typedef int (*init_fn)(void**, int, UniqueId, int);
    const int r = reinterpret_cast<init_fn>(reinterpret_cast<void*>(g_rccl.CommInitRank))(&c->comm, world, id, rank);
    if (r != 0) { delete c; return fail(XFR_RCCL_ERROR, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    *out = c;
    return XFR_OK;
}

xfr_status xfr_broadcast_weights(xfr_engine* e, xfr_comm* c, int32_t root, void* stream)
{
    if (!e || !c) return fail(XFR_INVALID_ARG, "null argument");
    if (root < 0 || root >= c->world) return fail(XFR_INVALID_ARG, "root %d outside [0, %d)", root, c->world);
    if (c->rank == root && !e->weights_loaded) return fail(XFR_STATE_ERROR, "the root rank has no weights loaded");
    HIP_TRY(hipSetDevice(e->device));
    const size_t bytes = e->arena_floats * sizeof(float);
    conv_gemm_forget_split(e->arena, bytes);
    RCCL_TRY(g_rccl.Broadcast(e->arena, e->arena, bytes, /* ncclChar */ 0, root, c->comm, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    e->weights_loaded = true;
    e->held_x = nullptr;
    presplit_weights(e);
    return XFR_OK;
}

xfr_status xfr_comm_destroy(xfr_comm* c)
{
    if (!c) return XFR_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return XFR_OK;
}

xfr_status xfr_engine_profile_csv(xfr_engine* e, const char* path)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->profile_csv = path ? path : "";
    return XFR_OK;
}

xfr_status xfr_chain_epilogue_stats(int64_t* compiled_launches, int64_t* interpreted_launches, int32_t* n_signatures)
{
    long c = 0, i = 0;
    conv_gemm_chain_launch_counts(&c, &i);
    if (compiled_launches) *compiled_launches = c;
    if (interpreted_launches) *interpreted_launches = i;
    if (n_signatures) *n_signatures = conv_gemm_num_chain_sigs();
    return XFR_OK;
}

// Device-free: builds the planner state of an engine (no HIP call, fake base addresses that are never dereferenced) and
// writes the fused schedules as text.
xfr_status xfr_plan_describe(const xfr_op_desc* ops, int32_t n_ops, int32_t n_weights, int32_t in_c, int32_t in_h, int32_t in_w,
                             int32_t batch, int32_t subtree_mode, int32_t seed_tensor, char* buf, size_t capacity, size_t* needed)
{
    if (!ops || n_ops < 2 || in_c < 1 || in_h < 1 || in_w < 1 || batch < 1 || n_weights < 0)
        return fail(XFR_INVALID_ARG, "xfr_plan_describe: bad arguments");
    if (subtree_mode < 0 || subtree_mode > 3) return fail(XFR_INVALID_ARG, "Invalid subtree mode %d", subtree_mode);
    if (ops[0].kind != XFR_OP_CONV || ops[0].in0 != 0)
        return fail(XFR_UNSUPPORTED_LAYER, "the first layer must be a convolution on the input image");
    xfr_engine* e = new xfr_engine();
    struct Del { xfr_engine* e; ~Del() { delete e; } } del{e};
    e->max_batch = batch; e->in_c = in_c; e->in_h = in_h; e->in_w = in_w; e->n_weights = n_weights;
    e->mode = subtree_mode;
    e->planning_only = true;
    // tools/gen_chain_sigs.py lists the chains of the test / A-B fusion levels too (XFR_DESCRIBE_FUSION = an xfr_engine_set_epilogue_fusion value)
    if (const char* f = getenv("XFR_DESCRIBE_FUSION")) xfr_engine_set_epilogue_fusion(e, atoi(f));
    xfr_status st = build(e, ops, n_ops);
    if (st == XFR_OK) st = layout_arena(e, false);
    if (st == XFR_OK) st = layout_workspace(e);
    if (st != XFR_OK) return st;
    if (seed_tensor < 2 || seed_tensor >= (int)e->tens.size()) return fail(XFR_INVALID_ARG, "bad seed tensor %d", seed_tensor);
    e->ws = reinterpret_cast<float*>((uintptr_t)1 << 40);
    e->arena = reinterpret_cast<float*>((uintptr_t)2 << 40);
    compute_need(e);
    BwdPlan* plan = nullptr;
    st = get_plan(e, seed_tensor, &plan);
    if (st != XFR_OK) return st;
    std::string out;
    char line[512];
    auto emit_sig = [&](const EwChain& ch) {
        uint16_t codes[XFR_MAX_EW_STEPS];
        const int n = ew_chain_codes(ch, codes);
        out += " SIG";
        for (int i = 0; i < n; ++i) { snprintf(line, sizeof(line), " %04x", codes[i]); out += line; }
        snprintf(line, sizeof(line), " compiled=%d", n > 0 ? conv_gemm_chain_sig(ch) : -1);
        out += line;
    };
    snprintf(line, sizeof(line), "plan seed_tensor %d mode %d firings %d launches %zu (unfused %zu)\n", seed_tensor, subtree_mode,
             plan->n_firings, plan->fused_gemm.size(), plan->steps.size());
    out += line;
    // the hooked module call behind every firing, reference order (the image hook of op 0, which the engine does not compute, comes last there)
    out += "firing_ops";
    for (int op : plan->firing_ops) { snprintf(line, sizeof(line), " %d", op); out += line; }
    out += "\n";
    // forward-only runs (encode / the gallery of a triplet step): Conv -> BatchNorm [-> Add] [-> ReLU] epilogues
    const int last_op = e->tens[seed_tensor].producer;
    e->fwd_done.assign(e->ops.size(), 0);
    e->pos_done.assign(e->ops.size(), 0);
    e->fwd_last_op = last_op;
    for (int k = 0; k <= last_op; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (e->fwd_done[k] || (d.kind != XFR_OP_CONV && d.kind != XFR_OP_LINEAR)) continue;
        ConvParams p;
        conv_geometry(e, k, batch, p);
        p.out0 = e->T(d.out);
        if (!fuse_mfm_forward(e, k, batch, false, p)) fuse_forward_only(e, k, batch, p, nullptr);
        if (p.chain.n == 0) continue;
        EwLoads ld;
        ew_plan_loads(p.chain, p.out0, ld, EW_FWD_SLOTS_WIDE);
        snprintf(line, sizeof(line), "fwd CONV op %d [%d x %d x %d] K %d", k, e->tens[d.out].C, e->tens[d.out].H, e->tens[d.out].W, e->ops[k].K);
        out += line;
        emit_sig(p.chain);
        out += "\n";
    }
    // the probe forward (positive pass alongside): Conv -> BatchNorm [-> ReLU] with the raw output kept
    e->fwd_done.assign(e->ops.size(), 0);
    e->pos_done.assign(e->ops.size(), 0);
    for (int k = 0; k <= last_op; ++k) {
        const xfr_op_desc& d = e->ops[k].d;
        if (d.kind != XFR_OP_CONV && d.kind != XFR_OP_LINEAR) continue;
        ConvParams p;
        conv_geometry(e, k, batch, p);
        p.out0 = e->T(d.out);
        if (!fuse_mfm_forward(e, k, batch, true, p)) fuse_probe_forward(e, k, batch, p, nullptr);
        if (p.chain.n == 0) continue;
        EwLoads ld;
        ew_plan_loads(p.chain, p.out0, ld, EW_FWD_SLOTS_WIDE);
        snprintf(line, sizeof(line), "probe CONV op %d [%d x %d x %d] K %d", k, e->tens[d.out].C, e->tens[d.out].H, e->tens[d.out].W, e->ops[k].K);
        out += line;
        emit_sig(p.chain);
        out += "\n";
    }
    static const char* kn[] = {"EW", "CONV_BWD", "MAXPOOL_BWD", "AVGPOOL_BWD", "COPY", "MAXHALVES_BWD", "NORMALIZE_BWD", "ZERO"};
    for (const BwdStep& b : plan->fused_gemm) {
        const int tt = b.kind == ST_EW ? b.ew_t : b.dst_t;
        snprintf(line, sizeof(line), "bwd %s src %d dst %d acc %d", kn[b.kind], b.src_t, b.dst_t, b.accumulate);
        out += line;
        if (tt >= 0) { snprintf(line, sizeof(line), " [%d x %d x %d]", e->tens[tt].C, e->tens[tt].H, e->tens[tt].W); out += line; }
        if (b.kind == ST_CONV_BWD) { snprintf(line, sizeof(line), " K %d", e->ops[b.op].Kb); out += line; }
        if (!b.chain.empty()) {
            EwChain ch;
            EwLoads ld;
            resolve_chain(e, b.chain, ch, nullptr, 2 * batch);
            ew_plan_loads(ch, e->G(b.dst_t), ld, b.kind == ST_CONV_BWD ? EW_FWD_SLOTS_WIDE : EW_FWD_SLOTS_BASE);
            if (b.kind == ST_CONV_BWD) emit_sig(ch);
            else { snprintf(line, sizeof(line), " steps %d", ch.n); out += line; }
            if (getenv("XFR_DESCRIBE_TYPES")) {
                out += " types";
                for (const auto& y : b.chain) { snprintf(line, sizeof(line), " %d:%d:%d", y.type, y.action, y.t0); out += line; }
            }
        }
        out += "\n";
    }
    // the schedule of the OBSERVING calls (priors / captures / stored firings: layerwise and weighted-subtree EBP): chains stay in their own launches there,
    // but copy forwarding leaves short hook-free chains (fan-in adds, store-backs) behind some GEMMs -- listed so that they get compiled epilogues too
    for (const BwdStep& b : plan->fused) {
        if (b.kind != ST_CONV_BWD || b.chain.empty()) continue;
        EwChain ch;
        EwLoads ld;
        resolve_chain(e, b.chain, ch, nullptr, 2 * batch);
        ew_plan_loads(ch, e->G(b.dst_t), ld, EW_FWD_SLOTS_WIDE);
        snprintf(line, sizeof(line), "observed-bwd CONV_BWD src %d dst %d acc %d", b.src_t, b.dst_t, b.accumulate);
        out += line;
        emit_sig(ch);
        out += "\n";
    }
    // the lean schedule of the same plan (xfr_engine_set_lean): probe-forward epilogues over two accumulator tiles, sweep chains on stored quotients
    lean_prepare(e, *plan, batch);
    if (plan->lean_state == 1) {
        e->lean_cur = plan;
        e->fwd_done.assign(e->ops.size(), 0);
        e->pos_done.assign(e->ops.size(), 0);
        int n_lean = 0;
        for (int k = 0; k <= last_op; ++k) {
            const xfr_op_desc& d = e->ops[k].d;
            if (d.kind != XFR_OP_CONV && d.kind != XFR_OP_LINEAR) continue;
            ConvParams p;
            conv_geometry(e, k, batch, p);
            p.out0 = e->T(d.out);
            const bool dual = e->tens[d.out].need_pv && e->tens[d.in0].nonneg;
            if (e->ops[k].pair || !dual) continue;
            fuse_probe_forward(e, k, batch, p, nullptr, dual);
            if (p.chain.n == 0 || p.chain.s[0].type != EW_LEAN_Q) continue;
            EwLoads ld;
            ew_plan_loads(p.chain, p.out0, ld, EW_FWD_SLOTS_WIDE);
            snprintf(line, sizeof(line), "lean-probe CONV op %d [%d x %d x %d] K %d", k, e->tens[d.out].C, e->tens[d.out].H, e->tens[d.out].W, e->ops[k].K);
            out += line;
            emit_sig(p.chain);
            out += "\n";
            ++n_lean;
        }
        e->lean_cur = nullptr;
        for (const BwdStep& b : plan->fused_gemm_lean) {
            if (b.chain.empty()) continue;
            EwChain ch;
            EwLoads ld;
            resolve_chain(e, b.chain, ch, nullptr, 2 * batch);
            ew_plan_loads(ch, e->G(b.dst_t), ld, b.kind == ST_CONV_BWD ? EW_FWD_SLOTS_WIDE : EW_FWD_SLOTS_BASE);
            snprintf(line, sizeof(line), "lean-bwd %s src %d dst %d", kn[b.kind], b.src_t, b.dst_t);
            out += line;
            if (b.kind == ST_CONV_BWD) emit_sig(ch);
            else { snprintf(line, sizeof(line), " steps %d", ch.n); out += line; }
            out += "\n";
        }
        snprintf(line, sizeof(line), "lean convolutions %d\n", n_lean);
        out += line;
    }
    if (needed) *needed = out.size() + 1;
    if (buf && capacity > 0) {
        const size_t n = std::min(out.size(), capacity - 1);
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return XFR_OK;
}

xfr_status xfr_engine_memory(xfr_engine* e, size_t* weight_bytes, size_t* workspace_bytes)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (weight_bytes) *weight_bytes = e->arena_floats * sizeof(float);
    if (workspace_bytes) *workspace_bytes = e->ws_floats * sizeof(float) + e->idx_bytes;
    return XFR_OK;
}

xfr_status xfr_engine_set_profile(xfr_engine* e, int32_t enable)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    e->profile_on = enable ? 1 : 0;
    return XFR_OK;
}

xfr_status xfr_engine_get_profile(xfr_engine* e, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops)
{
    if (!e) return fail(XFR_INVALID_ARG, "null engine");
    if (gemm_ms) *gemm_ms = e->last_gemm_ms;
    if (gemm_launches) *gemm_launches = e->last_gemm_launches;
    if (gemm_flops) *gemm_flops = e->last_gemm_flops;
    return XFR_OK;
}

xfr_status xfr_engine_get_profile_by_kernel(xfr_engine* e, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops)
{
    if (!e || !gemm_ms || !gemm_launches || !gemm_flops) return fail(XFR_INVALID_ARG, "null argument");
    for (int q = 0; q < 2; ++q) { gemm_ms[q] = e->fam_ms[q]; gemm_launches[q] = e->fam_launches[q]; gemm_flops[q] = e->fam_flops[q]; }
    return XFR_OK;
}

}  // extern "C"
