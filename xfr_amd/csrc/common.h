// common.h -- shared declarations of the xfr_amd HIP engine (gfx950 only).
//
// Tensor layout in HBM: every activation / gradient tensor is stored CNHW, i.e. [C][NB][H][W] with NB the
// number of images in flight (N for forward tensors, S*N for the S gradient streams).  With the batch folded
// inside the channel, a 1x1 stride-1 convolution over the whole batch is ONE plain row-major GEMM
// Out[Cout][NB*H*W] = W[Cout][Cin] * In[Cin][NB*H*W], tiles may straddle image boundaries, and the spatial
// index -- the contiguous one -- is the MFMA "B" lane index, so both the im2col gather and the epilogue
// stores are coalesced along it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XFR_MAX_EW_STEPS 14

// ---- fused elementwise chain (micro-program) -------------------------------------------------------------------------
// Executed per element either by the stand-alone ew_chain kernels or inside the conv_gemm epilogue.
// Index spaces: "g-index" = position in the tensor being produced ([C][SB][HW]); "a-index" = the same (c, hw) in a
// forward-side tensor with B images per channel row (sample b = sb % B).  For forward launches SB == B.
enum {
    EW_HOOK = 0,      // tensor hook (whitebox.py:388-428): a = relu(p0[a]), x = relu(p1[a]) or a; p = a*relu(g); action
    EW_MASK = 1,      // g = p0[a] > 0 ? g : 0                      (in-place ReLU VJP); action 1: the mask is the sign bit of a stored quotient (HOOK_GATE_SIGN)
    EW_SCALE_C = 2,   // g *= p0[c]                                 (BatchNorm VJP with relu(gamma)*invstd)
    EW_SCALE = 3,     // g *= f                                     (Multiply VJP)
    EW_STORE = 4,     // pstore[g] = g.  action 1: no store -- SAVE g (a branch point); action 2: pstore[g] = g, then g = the saved value: the steps
                      // between a save and its store-restore are a side BRANCH of the chain (GEMM epilogues: the main path's hook chain of a
                      // projection-shortcut block next to the shortcut's, resnet50_128.py; one save per chain)
    EW_ADDP = 5,      // g += p0[g]                                 (gradient fan-in / residual add)
    EW_AFFINE_C = 6,  // g = g*p0[c] + p1[c]                        (eval BatchNorm forward)
    EW_RELU = 7,      // g = max(g, 0)
    EW_FORK_POSBN = 8,// pstore[g] = max(g,0)*p0[c] + p1[c]         (positive-pass BatchNorm output, g unchanged)
                      // with p2 != null: ... + (action & 1 ? relu(p2[g]) : p2[g]) -- the positive-pass output of the functional add behind the BatchNorm
                      // (resnet50_128.py torch.add(shortcut, 1, bn)), whose other operand is already there; the BatchNorm's own positive output is then not stored
    EW_MAXHALF_IN = 9,// chain HEAD only (stand-alone kernels): the VJP of torch.max(split[0], split[1]) (lightcnn.py:62).  The chain runs
                      // over the 2*Co-channel Split tensor, the source gradient has Co channels: g = src[c % Co], routed by the true
                      // forward halves a = p0[c % Co], b = p0[c % Co + Co] (ties split evenly like at::maximum); action = Co
    EW_MAXHALF_OUT = 11,// compiled GEMM epilogue only: the same VJP as a FAN-OUT in the epilogue of the GEMM that produces the Co-channel
                      // gradient: for both halves h the routed gradient runs the REST of the chain as channel c + h*Co of the 2*Co-channel
                      // tensor (operands loaded in place at that channel) and is stored there; p0 = true forward halves, action = Co
    EW_ADDP_CO = 13,  // compiled GEMM epilogue only, behind EW_MAXPAIR: g += p0[channel c of the Co-channel tensor] -- the residual add of a Light-CNN
                      // resblock (lightcnn.py:88) on the pair maximum; the even rows then store the SUM
    EW_FORK_POSADD = 14, // ... and, in front of it: pstore[c of Co] = (action & 1 ? relu(g) : g) + (action & 2 ? relu(p0) : p0) -- the Add module's
                      // positive-pass output (its inputs are overridden by A = relu(true input), whitebox.py:315-330); g unchanged
    EW_POOL2_IN = 12, // chain HEAD only (stand-alone kernels): the VJPs of MaxPool2d(2) and AvgPool2d(2) on the SAME input, summed (lightcnn.py:252:
                      // `maxpool(x) + avgpool(x)`).  The chain runs over the pools' input tensor [C][SB][H][W]; the source gradient is the gradient of
                      // the sum, [C][SB][H/2][W/2]: g = 0.25 * src[window] + (argmax[window] == this pixel ? src[window] : 0) -- what AVGPOOL_BWD
                      // followed by an accumulating MAXPOOL_BWD leave in the tensor, bit for bit.  p0 = the max-pool's argmax bytes ([C][B][H/2][W/2],
                      // window-local index dh * 2 + dw), action = W
    EW_AVGUP_IN = 15, // chain HEAD only (stand-alone kernels): the block-input gradient of a down-sampling residual block whose shortcut is
                      // AvgPool2d(2) (+ ConcatChannels) and whose main path starts with a 1x1 / stride 2 convolution (resnet.py:111-149), built in place of
                      // four launches (slice copy, the pooled tensor's hook, the average pool's VJP, the strided GEMM's read-modify-write):
                      //   g(ih, iw) = (0 + hook(src[ih/2, iw/2])) * 0.25  [+ p2[ih/2, iw/2] on the even (ih, iw): the GEMM's result, kept compact]
                      // src = the pooled tensor's gradient ([C][SB][H/2][W/2]: the first C rows of the concatenated gradient); the hook of the pooled
                      // tensor is EW_HOOK's arithmetic on one value (action = HOOK_* or -1 for none, p0 / p1 its a / x at pooled resolution, null p0:
                      // not observed); prior_sb = W.  Same operands, same operations as the four launches: same bits.
                      // action = -2: no pooled source -- the tensor was a zero fill plus strided 1x1 GEMMs (projection shortcut + main path,
                      // resnet50_128.py): g = p2 on the even pixels, 0 elsewhere; the GEMMs accumulate on the compact grid.
    // Lean probe forward (compiled epilogue of a ConvParams::dualacc launch only; g = W accumulator + bias, gp = relu(W) accumulator + bias):
    EW_LEAN_Q = 16,   // FIRST step: q = relu(g) / (relu(gp) + eps) is kept in registers -- the BatchNorm hook's a / (x + eps) (whitebox.py:388-428); g unchanged
    EW_LEAN_XR = 17,  // xr = relu(relu(g) * p0[c] + p1[c] [+ (action & 1 ? relu(p2[g]) : p2[g])]) is kept in registers: what EW_FORK_POSBN would have
                      // stored, clamped -- the x of the ReLU hook behind the BatchNorm [and the functional add]; g unchanged
    EW_LEAN_STORE = 18, // LAST steps.  action 0: pstore[g] = q with its sign bit set where the chain's final value g is <= 0 (the ReLU mask and every
                      // x == a hook of that tensor read the sign, the BatchNorm hook the magnitude); action 1: pstore[g] = g / (xr + eps), the ReLU hook's quotient
    EW_MAXPAIR = 10   // GEMM epilogue only, last step: g = max(g, value of the partner row c ^ 1) -- MaxFeatureMap of a convolution
                      // whose output channels were packed interleaved (row 2c = channel c, row 2c+1 = channel c + Co); the even
                      // rows then store g as channel c of the Co-channel output
};
enum { HOOK_DIV = 0, HOOK_RELU = 1, HOOK_PASS = 2,
       // The "lean" forms of a hook nobody observes (no P store, trace, prior or capture): the probe forward left, instead of the hook's literal
       // operands a and x, what the sweep needs of them (ConvParams::dualacc; engine.hip lean_rewrite).  Within one ulp per hook of the
       // literal expression for every a a real network produces -- gated by the golden tolerances, not bit for bit.
       HOOK_Q = 3,          // g = relu(g) * |p0[a]|:  p0 holds q = a / (x + eps), computed once in the probe forward (BatchNorm and ReLU hooks)
       HOOK_GATE = 4,       // g = p0[a] > 0 ? relu(g) : 0:  a hook whose x IS its a -- a * relu(g) / (a + eps) = relu(g) wherever a > 1.7e-9, 0 at a = 0
       HOOK_GATE_SIGN = 5   // the same gate read from the SIGN bit of a stored quotient (set where the gated tensor is <= 0)
     };
enum { PRIOR_DIV = 0, PRIOR_PASS = 1, PRIOR_GATEZ = 2 };   // p/(x+eps) | gradient unchanged | (prior>0)*z

struct EwStep {
    int type;
    int action;        // HOOK_*
    const float* p0;
    const float* p1;
    float* pstore;     // HOOK: if non-null, p is stored here (g-index);  STORE / FORK_POSBN: destination
    const float* p2;   // FORK_POSBN: optional addend (g-index), see above
    double* trace;     // HOOK: if non-null, sum(p) per (stream,sample) is accumulated at trace[sb] (stand-alone kernels only)
    float f;           // SCALE: factor
    // HOOK extras for layerwise EBP (whitebox.py:390-392,406-419): at this firing a gradient row sb (stream * B + sample) may
    // have its p OVERRIDDEN by a prior --
    //   prior_elem != null: table over sb; prior_elem[sb] >= 0 means p of row sb is zero except element prior_elem[sb]
    //                       (c*HW+hw within the sample), which is prior_val[sb]   (mode 'elementwise'; many probes x sweeps)
    //   prior_dense != null: row prior_sb gets the dense tensor prior_dense       (mode 'argmax'; one sweep of one image)
    int prior_sb;
    int prior_action;  // PRIOR_* : what the hook returns for such a row
    const int* prior_elem;
    const float* prior_val;
    const float* prior_dense;
    // HOOK: capture p of element cap_elem[sb] (c*HW+hw within the sample; -1: none) of every row sb into cap_dst[sb]
    // (stand-alone kernels only)
    const int* cap_elem;
    float* cap_dst;
    // float4 chain kernel: prefetch slots of p0 / p1 (elementwise.hip, plan_loads); -1: load in place, -2: not needed
    int ls0, ls1;
};

struct EwChain {
    int n;
    EwStep s[XFR_MAX_EW_STEPS];
};

// Operand prefetch plan of a chain.  The per-element operands of all steps -- the forward stashes a / x of the hooks, ReLU
// masks, fan-in gradients -- do not depend on g, and several steps of one chain read the same tensor (in-place ReLU: the
// ReLU hook, the BatchNorm hook and the mask all see the ReLU output).  The float4 chain kernel and the chain epilogue
// of the GEMM issue the distinct loads together before they interpret the steps, instead of one dependent load after
// another.  EwStep::ls0/ls1 hold the slot of p0 / p1.
// Slots 0..2 hold operands indexed like the forward tensors (shared by the gradient streams of one position), slot 3 the one
// gradient-indexed operand (fan-in); slots 5 and 6 are two more forward-indexed ones that only the compiled GEMM epilogues use
// (the 'norelu' / 'all' chains read up to five forward tensors).  4 is not a slot: it encodes "load in place" in a signature.
constexpr int EW_NLOADS = 7;
constexpr int EW_FWD_SLOTS_BASE = 3, EW_FWD_SLOTS_WIDE = 5;
struct EwLoads {
    int nl;                                  // stand-alone kernels: slots 0..nl-1 may be in use (3 may follow a gap)
    const float* lp[EW_NLOADS];
    int lk[EW_NLOADS];                       // 0: indexed like the forward tensors (a-index), 1: like the gradient
};

// Assign prefetch slots.  A load may be hoisted only if nothing in the chain (or the final store to dst) writes the
// buffer it reads: stores land at the same element index in the same thread, after the prefetch.
inline void ew_plan_loads(EwChain& ch, const float* dst, EwLoads& ld, int fwd_slots = EW_FWD_SLOTS_BASE)
{
    static const int fwd_order[EW_FWD_SLOTS_WIDE] = {0, 1, 2, 5, 6};
    ld.nl = 0;
    for (int l = 0; l < EW_NLOADS; ++l) { ld.lp[l] = nullptr; ld.lk[l] = 0; }
    auto written = [&](const float* p) {
        if (p == dst) return true;
        for (int i = 0; i < ch.n; ++i)
            if (ch.s[i].pstore == p) return true;
        return false;
    };
    auto slot_for = [&](const float* p, int kind) -> int {
        if (!p || written(p)) return -1;
        if (kind == 1) {
            if (ld.lp[3] == p) return 3;
            if (ld.lp[3]) return -1;
            ld.lp[3] = p;
            ld.lk[3] = 1;
            return 3;
        }
        for (int i = 0; i < fwd_slots; ++i) {
            const int l = fwd_order[i];
            if (ld.lp[l] == p) return l;
            if (!ld.lp[l]) { ld.lp[l] = p; ld.lk[l] = 0; return l; }
        }
        return -1;
    };
    bool fanned = false;       // behind EW_MAXHALF_OUT the steps run per half at another channel: their operands are loaded in place
    for (int i = 0; i < ch.n; ++i) {
        EwStep& st = ch.s[i];
        st.ls0 = -1;
        st.ls1 = -1;
        if (st.type == EW_MAXHALF_OUT) { fanned = true; continue; }
        if (fanned) {
            if (st.type == EW_HOOK && !st.pstore && !st.trace && (st.action == HOOK_RELU || st.action == HOOK_PASS) && !st.prior_elem && !st.prior_dense && !st.cap_dst) st.ls0 = -2;
            continue;
        }
        if (st.type == EW_HOOK) {
            if (!st.pstore && !st.trace && (st.action == HOOK_RELU || st.action == HOOK_PASS) && !st.prior_elem && !st.prior_dense && !st.cap_dst) { st.ls0 = -2; continue; }
            st.ls0 = slot_for(st.p0, 0);
            if (st.action == HOOK_DIV && st.p1) st.ls1 = slot_for(st.p1, 0);
        } else if (st.type == EW_MASK) {
            st.ls0 = slot_for(st.p0, 0);
        } else if (st.type == EW_ADDP) {
            st.ls0 = slot_for(st.p0, 1);
        }
    }
    ld.nl = ld.lp[3] ? 4 : (ld.lp[2] ? 3 : (ld.lp[1] ? 2 : (ld.lp[0] ? 1 : 0)));   // slots in use form a prefix, except that 3 may follow a gap
}

// Compile-time signature of a planned chain: one 16-bit code per step that does anything.  The GEMM epilogue is
// specialised per signature (conv_gemm.hip: the table chain_sigs.inc lists the signatures the three backbones produce;
// chains outside the table run through the interpreter).  Code = op | s0 << 5 | s1 << 8 | store << 11 | step << 12 with
// s0 / s1 the prefetch slot of p0 / p1 (0..3, 5, 6), 4 = load in place, 7 = none (a hook whose x is its a).
enum { SIG_END = 0, SIG_HOOK_DIV = 1, SIG_HOOK_RELU = 2, SIG_HOOK_PASS = 3, SIG_RELU = 4, SIG_MASK = 5, SIG_SCALE_C = 6, SIG_SCALE = 7,
       SIG_STORE = 8, SIG_ADDP = 9, SIG_AFFINE_C = 10, SIG_FORK_POSBN = 11, SIG_MAXPAIR = 12, SIG_MAXHALF_OUT = 13, SIG_ADDP_CO = 14, SIG_FORK_POSADD = 15,
       SIG_HOOK_Q = 16, SIG_GATE = 17, SIG_GATE_SIGN = 18, SIG_MASK_SIGN = 19, SIG_LEAN_Q = 20, SIG_LEAN_XR = 21, SIG_LEAN_STORE = 22 };
constexpr int sig_op(unsigned c) { return (int)(c & 31u); }
constexpr int sig_s0(unsigned c) { return (int)((c >> 5) & 7u); }
constexpr int sig_s1(unsigned c) { return (int)((c >> 8) & 7u); }
constexpr bool sig_store(unsigned c) { return ((c >> 11) & 1u) != 0; }
constexpr int sig_step(unsigned c) { return (int)((c >> 12) & 15u); }
constexpr bool sig_is_slot(int v) { return v != 4 && v != 7; }

// codes[] of a chain whose prefetch slots are assigned (ew_plan_loads); returns the number of codes, or -1 if the chain
// uses features only the interpreter has (priors, captures, traces)
inline int ew_chain_codes(const EwChain& ch, uint16_t codes[XFR_MAX_EW_STEPS])
{
    int n = 0;
    auto slot = [](int ls) -> unsigned { return ls >= 0 ? (unsigned)ls : 4u; };     // slots 0..3, 5, 6; 4 = in place
    for (int i = 0; i < ch.n; ++i) {
        const EwStep& st = ch.s[i];
        unsigned op = 0, s0 = 7, s1 = 7, store = 0;
        switch (st.type) {
            case EW_HOOK:
                if (st.trace || st.prior_elem || st.prior_dense || st.cap_dst) return -1;
                if (st.ls0 == -2) {                     // p unobserved: relu(g) or the identity
                    if (st.action != HOOK_RELU) continue;
                    op = SIG_RELU;
                    break;
                }
                op = st.action == HOOK_DIV ? SIG_HOOK_DIV : st.action == HOOK_RELU ? SIG_HOOK_RELU : st.action == HOOK_PASS ? SIG_HOOK_PASS
                     : st.action == HOOK_Q ? SIG_HOOK_Q : st.action == HOOK_GATE ? SIG_GATE : SIG_GATE_SIGN;
                s0 = slot(st.ls0);
                if (st.action == HOOK_DIV && st.p1) s1 = slot(st.ls1);
                store = st.pstore ? 1u : 0u;
                break;
            case EW_MASK: op = st.action == 1 ? SIG_MASK_SIGN : SIG_MASK; s0 = slot(st.ls0); break;
            case EW_SCALE_C: op = SIG_SCALE_C; break;
            case EW_SCALE: op = SIG_SCALE; break;
            case EW_STORE: op = SIG_STORE; if (st.action == 1 || st.action == 2) s0 = (unsigned)st.action; break;     // s0: 7 plain, 1 save, 2 store + restore
            case EW_ADDP: op = SIG_ADDP; s0 = slot(st.ls0); break;
            case EW_AFFINE_C: op = SIG_AFFINE_C; break;
            case EW_RELU: op = SIG_RELU; break;
            case EW_FORK_POSBN: op = SIG_FORK_POSBN; if (st.p2) { store = 1; s1 = (unsigned)(st.action & 1); } break;     // store bit: with addend; s1: its clamp flag
            case EW_MAXPAIR: op = SIG_MAXPAIR; break;
            case EW_MAXHALF_OUT: op = SIG_MAXHALF_OUT; break;
            case EW_ADDP_CO: op = SIG_ADDP_CO; break;
            case EW_FORK_POSADD: op = SIG_FORK_POSADD; s0 = (unsigned)(st.action & 3); break;      // the two clamp flags are part of the signature
            case EW_LEAN_Q: op = SIG_LEAN_Q; break;
            case EW_LEAN_XR: op = SIG_LEAN_XR; if (st.p2) { store = 1; s1 = (unsigned)(st.action & 1); } break;     // like SIG_FORK_POSBN
            case EW_LEAN_STORE: op = SIG_LEAN_STORE; s0 = (unsigned)(st.action & 1); break;                          // s0: which quotient
            default: return -1;
        }
        codes[n++] = (uint16_t)(op | (s0 << 5) | (s1 << 8) | (store << 11) | ((unsigned)i << 12));
    }
    return n;
}
// index into the compiled-signature table, -1 if absent (conv_gemm.hip)
int conv_gemm_chain_sig(const EwChain& planned_chain);
int conv_gemm_num_chain_sigs();
void conv_gemm_chain_launch_counts(long* compiled, long* interpreted);

// ---- implicit-GEMM convolution -------------------------------------------------------------------------------
struct ConvParams {
    const float* in;    // [Cin][NB][H][W]
    const float* w;     // packed [K][ldw], K = Cin*kh*kw ordered (ci, kh, kw); column = output channel
    const float* w_pos; // second weight matrix of a dual launch (relu(W)), same packing
    const float* bias;  // [CoutTot] or nullptr
    const float* bias_pos;
    float* out0;        // output of half 0 (w, bias)
    float* out1;        // output of half 1 (w_pos, bias_pos) when nhalves == 2
    int Cin, H, W, NB;  // NB: images covered by this launch (M = NB*OH*OW)
    int in_nb, out_nb;  // images per channel row of the input / output tensors (>= NB: a launch may cover a batch prefix)
    unsigned in_bytes;  // byte size of the input tensor (buffer descriptor range; < 2^31)
    int force_cfg;      // 0 = heuristic tile choice, else a configuration id (tuning / tests)
    int pair_m;         // > 0: the launch covers exactly TWO gradient streams of pair_m = B * OH * OW columns each (M = 2 * pair_m): where pair_m is a
                        // multiple of the tile width the m-tiles are walked stream-interleaved (tile 2j = stream 0's j-th, 2j + 1 = stream 1's j-th),
                        // so the forward-side operands of the epilogue chain -- the same for both streams -- are fetched from HBM once and hit L2 the
                        // second time.  Which workgroup computes which tile changes; no arithmetic does.
    int as_strided;     // a stride-2 1x1 backward-data GEMM whose result stays on its own (compact) grid (EW_AVGUP_IN): tile configuration -- hence K
                        // order -- of the scattering launch it replaces
    // tail balancing (conv_gemm.hip): the last tiles % CUs tiles of a small grid are cut along K into tail_s parts
    // each, so that every CU gets the same share of the final round; parts meet in tail_ws, the last arriver reduces.
    float* tail_ws;     // scratch for the parts' accumulators (nullptr: never balance); one per stream in flight
    unsigned* tail_cnt; // arrival counters, zero between launches (XFR_TAIL_MAX_TILES entries)
    size_t tail_ws_bytes;
    int tail_force;     // 0 = heuristic, 1 = off, S >= 2 = force S parts per tail tile (tuning / tests)
    int tail_q, tail_s; // set by launch_conv_gemm: whole tiles, parts per tail tile (tail_s <= 1: off)
    int tap_major;      // 1: K ordered (kh, kw, ci) instead of (ci, kh, kw); requires Cin % 16 == 0.  2: Cin <= 4 (image stems): K ordered
                        // (kh, kw, 4 channel slots), i.e. K = 4 * kh * kw with zero rows for the missing channels
    int K_logical;      // Cin * kh * kw, for FLOP accounting (== K unless tap_major == 2)
    int kh, kw, stride, pad;
    int OH, OW;
    int K, M;           // M = NB*OH*OW
    int CoutTot, nhalves, ldw;   // CoutTot = output channels per half
    int dualacc;        // 1 (nhalves == 1): every workgroup accumulates W AND relu(W) -- taken from the W fragment in registers, no second pack is read -- over the
                        // same staged input tile; the chain (compiled, first step EW_LEAN_Q) sees both.  The lean probe forward: bias_pos is the second bias
    int relu_in;        // clamp the gathered input at 0 (A = relu(input))
    int bwd;            // 1: a backward-data GEMM of the sweep (bwd_conv_params); 0: a forward convolution
    int split_ok;       // 1: the launch may take the bf16x6 kernel (conv_gemm_split.hip K17) where the layer is one it covers and the grid large enough; 2: whatever the grid (xfr_engine_set_split_gemm)
    int accumulate;     // out += result
    int out_H, out_W, out_stride;  // out_stride > 1: scatter the (OH,OW) grid into an (out_H,out_W) tensor
    int chain_B;        // forward batch for the a-index of the epilogue chain
    float chain_eps;    // eps of the hook divide
    EwChain chain;      // epilogue micro-program applied to half 0 before the final store (n == 0: none)
    EwLoads chain_ld;   // its operand prefetch plan (set by launch_conv_gemm)
    int chain_sig;      // index of the chain's compiled epilogue (chain_sigs.inc), -1: interpreted (set by launch_conv_gemm)
    int chain_interpret; // 1: run the chain through the interpreted epilogue even if a compiled one exists (tests)
    int co_pair;         // Co > 0: MaxFeatureMap convolution whose 2*Co output channels are PACKED interleaved (GEMM row 2c = channel c, row
                         // 2c+1 = channel c + Co): rows are stored at their channel's place, a chain may end in EW_MAXPAIR
    unsigned long long* stamps;   // tuning hook (xfr_debug_conv_stamps): per wave 8 words -- s_memrealtime at kernel entry, first operands landed,
                                  // K loop done, epilogue entered, exit; HW_ID; XCC_ID; life in shader cycles.  nullptr: off
    int stamps_cap;               // workgroups the stamp buffer holds (later ones do not record)
    int stamp_regions, stamp_seq; // sampled mode (regions > 0): launch `seq` records up to 256 evenly spaced workgroups into region seq % regions
    unsigned long long* span;     // tuning hook (xfr_debug_conv_log): eight words of THIS launch -- block 0's entry time, then exit times of a sample of the workgroups
                                  // (s_memrealtime), i.e. when the launch really started and ended next to the other streams' launches
};
void conv_gemm_set_stamps(unsigned long long* dev_ptr, int capacity_workgroups);
// launch log: every later launch_conv_gemm takes the next eight-word record of log_dev (capacity records) and notes its shape and
// stream on the host; conv_gemm_dump_log writes one CSV line per launch (seq, stream, shape, cfg, start and end in 10 ns ticks)
void conv_gemm_set_log(unsigned long long* log_dev, int capacity);
int conv_gemm_dump_log(const char* path);   // every later launch_conv_gemm records into dev_ptr (nullptr: stop)

constexpr int XFR_TAIL_MAX_TILES = 256;
constexpr size_t XFR_TAIL_WS_BYTES = (size_t)16 << 20;
// false: nothing was launched -- a dual (W / relu(W)) launch carries a chain for which no compiled epilogue exists
bool launch_conv_gemm(const ConvParams& p, hipStream_t s);
// 0 if launch_conv_gemm will accept p's fused chain, else why not (text: conv_gemm_refusal): 1 = dual launch without a compiled epilogue,
// 2 = MaxFeatureMap step (EW_MAXPAIR / EW_MAXHALF_OUT) without a compiled epilogue.  No device work.
int conv_gemm_cannot_launch(const ConvParams& p);
const char* conv_gemm_refusal(int why);
int conv_gemm_pick_cfg(const ConvParams& p);
int conv_gemm_last_cfg();       // the configuration the calling thread's last launch_conv_gemm really ran (9: the bf16x6 kernel)
// bf16x6 split GEMM (conv_gemm_split.hip K17, configuration 9; ConvParams::split_ok).  The fp32 pack of a covered layer needs bf16 planes: the engine
// builds them when the weights arrive (conv_gemm_presplit; a pack that has none at a launch gets them there, on that launch's stream, which is then
// drained once).  conv_gemm_forget_split: the packs inside [lo, lo + bytes) changed or go away -- drop their planes.
void conv_gemm_forget_split(const void* lo, size_t bytes);
bool conv_gemm_presplit(const ConvParams& layer, const float* w, hipStream_t s);     // the planes of pack w of that layer (geometry, K, CoutTot, ldw); false: no memory (the pack stays on the fp32 kernels)
int conv_gemm_split_covers(const ConvParams& p);       // a layer xfr_engine_set_split_gemm covers (whatever the launch's grid)
long conv_gemm_split_launches();

// g = src[idx]; run chain; dst[idx] = (accumulate ? dst[idx] : 0) + g.   Tensors are [C][SB][HW] for the gradient
// and [C][B][HW] for the forward-side sources (sample b = sb % B).
// SBa <= SB: only the first SBa gradient streams of every channel row are processed (layerwise sweeps whose later
// streams are still identically zero); the float4 kernel honours it, the scalar kernels process all SB.
void launch_ew_chain(const float* src, float* dst, int accumulate, const EwChain& chain,
                     int C, int SB, int B, int HW, float eps, hipStream_t s, int SBa = -1);

// ---- simple forward / backward kernels -----------------------------------------------------------------------
void launch_nchw_to_cnhw(const float* in, float* out, int N, int C, int HW, hipStream_t s);
void launch_cnhw_to_nchw(const float* in, float* out, int N, int C, int HW, hipStream_t s);
// uint8 H x W x C images (as a decoder hands them over) -> the network's fp32 input in CNHW, preprocessing included, in float64 like the
// reference's numpy code: kind 0 = out[c] = (float)(u8[c] - mean[c]) (resnet.py:25-37, whitebox.py:235-258), kind 1 = luminance
// (float)((r/255) w0 + (g/255) w1 + (b/255) w2) into ONE channel (lightcnn.py:19-25 through skimage's rgb2gray)
struct U8Pre { int kind; int channels; double mean[4]; double weight[4]; };
void launch_u8hwc_to_cnhw(const uint8_t* in, float* out, int N, int Cnet, int HW, const U8Pre& pre, hipStream_t s);
// out = maybe_relu( maybe_relu_in(in) * alpha[c] + beta[c] )
void launch_affine_c(const float* in, float* out, const float* alpha, const float* beta, int C, long per_c,
                     int relu_in, int relu_out, hipStream_t s);
void launch_relu(const float* in, float* out, long n, hipStream_t s);
void launch_scale(const float* in, float* out, long n, float f, int relu_in, hipStream_t s);
// out = maybe_relu_out( maybe_relu_in(a) + maybe_relu_in(b) )
void launch_add2(const float* a, const float* b, float* out, long n, int relu_a, int relu_b, int relu_out, hipStream_t s);
// dst (+)= src
void launch_copy_acc(const float* src, float* dst, long n, int accumulate, hipStream_t s);
void launch_maxpool_fwd(const float* in, float* out, uint8_t* idx, int CN, int H, int W, int OH, int OW,
                        int k, int stride, int pad, hipStream_t s);
// out_sum = maxpool2x2(in) + avgpool2x2(in), idx = the max-pool's argmax bytes (may be null), out_pos (may be null) = the positive-pass sum
// (relu_max_pos ? relu(max) : max) + (pos_avg_mode 0: avg, 1: relu(avg), 2: avgpool(relu(in))); bit-identical to the three separate kernels
bool pool2_fwd_ok(const float* in, const uint8_t* idx, int CN, int H, int W, int OH, int OW);
// 5x5 / stride 1 / pad 2 convolution of a one-channel image with MaxFeatureMap, as a direct convolution reading the GEMM's weight pack (rows
// interleaved: 2c = channel c, 2c + 1 = channel c + Co); raw (may be null) [2 Co][NB][H][W], omax [Co][NB][H][W]; bit-identical to the GEMM path
bool stem5_mfm_ok(const float* in, int NB, int H, int W);
// Whitebox.P[-1]: out[n][ci][ih][iw] = relu(img) * relu(conv_backward_data(g, relu(W))) for the first convolution (on demand only)
void launch_image_mwp(const float* g, const float* wp, const float* img, float* out, int Cin, int N, int H, int W, int Cout, int OH, int OW,
                      int kh, int kw, int stride, int pad, int ldw, int kmode, int co_pair, hipStream_t s);
void launch_stem5_mfm(const float* in, const float* wp, int ldw, const float* bias, float* raw, float* omax, int Co, int NB, int H, int W, hipStream_t s);
void launch_pool2_fwd(const float* in, float* out_sum, uint8_t* idx, float* out_pos, int CN, int H, int W, int OH, int OW, int relu_max_pos,
                      int pos_avg_mode, hipStream_t s);
// gin (+)= scatter of gout through the stored argmax; gradient batch SB vs forward batch B
void launch_maxpool_bwd(const float* gout, const uint8_t* idx, float* gin, int accumulate, int C, int SB, int B,
                        int H, int W, int OH, int OW, int k, int stride, int pad, hipStream_t s);
// zero_planes > 0: that many further planes behind the CN pooled ones are written as zeros (the pooled shortcut inside its channel-padded form)
void launch_avgpool_fwd(const float* in, float* out, int CN, int H, int W, int OH, int OW, int k, int stride,
                        int relu_in, hipStream_t s, int zero_planes = 0);
void launch_avgpool_bwd(const float* gout, float* gin, int accumulate, int CN, int H, int W, int OH, int OW,
                        int k, int stride, hipStream_t s);
// out[c] = max(in[c], in[c+Co]) ; optional relu on load
void launch_maxhalves_fwd(const float* in, float* out, int Co, long per_c, int relu_in, hipStream_t s);
// gin[2Co] from gout[Co] using the TRUE forward input tin (ties split evenly, like at::maximum backward)
void launch_maxhalves_bwd(const float* gout, const float* tin, float* gin, int accumulate, int Co, int SB, int B,
                          int HW, hipStream_t s);
// per-sample L2 normalise over channels: tensor [C][NB] (HW == 1)
void launch_normalize_fwd(const float* in, float* out, float* norms, int C, int NB, int relu_in, hipStream_t s);
void launch_normalize_bwd(const float* gout, const float* tin, const float* norms, float* gin, int accumulate,
                          int C, int SB, int B, hipStream_t s);
// seed [S][N][D] (D = C*HW, NCHW order inside a sample) -> gradient tensor [C][S*N][HW]
void launch_seed_to_cnhw(const float* seed, float* g, int SB, int C, int HW, hipStream_t s);
void launch_fill(float* p, long n, float v, hipStream_t s);
// per sample n: v = (gate_ge0 ? gm >= 0 : gm < 0) * (-gn) over the C*HW elements of gm = G[:, n], gn = G[:, N+n] (two gradient streams);
// vmax[n] = max v, vidx[n] = smallest c*HW+hw attaining it   (whitebox.py:689-690)
// weighted-subtree layer weights for all hooked tensors of a sweep in two launches: tensor u = desc[u] ([C][2N][HW]
// gradients, mate stream then non-mate stream); firing f reads tensor f2u[f]; vmax / vidx are [n_firings][N]
struct StatDesc { const float* G; int C; int HW; };
void launch_subtree_stats(const StatDesc* desc_dev, int n_tensors, const int* f2u_dev, int n_firings, float* vmax, int* vidx,
                          void* scratch, int N, int gate_ge0, hipStream_t s);
size_t subtree_stats_scratch_bytes(int N, int n_tensors);

// ---- saliency post-processing --------------------------------------------------------------------------------
// pooled[sb][hw] = sum_c P[c][sb][hw]
void launch_channel_pool(const float* P, float* pooled, int C, int SB, int HW, hipStream_t s);
// sums[sb] = sum over (c,hw) of P[c][sb][hw]  (double accumulation)
void launch_sample_sums(const float* P, double* sums, int C, int SB, int HW, hipStream_t s);
// contrast[n][hw] = sum_c relu(keep * (P[c][n][hw]/sum_n - P[c][N+n][hw]/sum_{N+n})), keep = P/sum >= thr[n] (thr may be null)
void launch_contrast(const float* P, const double* sums, const float* thr, float* out, int C, int N, int HW, hipStream_t s);
// per-sample truncation threshold of whitebox.py:550-553 on m = P[:, n, :]/sum_n: smallest value v such that the
// ascending cumulative sum reaches percentile% of the total at v.
void launch_truncation_threshold(const float* P, const double* sums, float percentile, float* thr, void* scratch,
                                 int C, int N, int HW, hipStream_t s);
size_t truncation_scratch_bytes(int N);
// gaussian(sigma=2, nearest, truncate 4) -> clamp -> /max(sum,eps)    in/out [N][H][W]; tmp same size
void launch_saliency_blur(const float* in, float* tmp, float* out, int N, int H, int W, float eps, hipStream_t s);
