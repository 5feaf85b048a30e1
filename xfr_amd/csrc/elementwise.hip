// elementwise.hip -- HBM-bound kernels of the EBP engine (gfx950): the fused hook chain of the MWP sweep,
// eval-mode BatchNorm / ReLU / Add / pools / MaxFeatureMap forward and VJPs, layout conversion.
//
// All tensors are CNHW (common.h).  These kernels do no data reuse, so the rules that matter are coalescing
// (consecutive lanes walk the contiguous H*W*NB run of one channel, float4 where alignment allows), enough
// workgroups to cover 256 CUs several times over (grid-stride loops, 256-thread blocks = 4 wave64), and fusing
// every elementwise step that touches the same gradient into ONE pass (EwChain) so that each gradient element is
// read and written once between two GEMMs.
#include "common.h"
#include "ew_interp.h"
#include <cstring>

namespace {

constexpr int NT = 256;

inline int grid_for(long n, int per_thread = 1)
{
    long b = (n + (long)NT * per_thread - 1) / ((long)NT * per_thread);
    if (b < 1) b = 1;
    if (b > 256L * 32) b = 256L * 32;   // 32 workgroups per CU, grid-stride beyond
    return (int)b;
}

// ---------------------------------------------------------------------------------------------------------
// The tensor hooks of whitebox.py:381-430 (+ ReLU / BatchNorm / Multiply VJPs), fused.
//   zh = relu(z); p = a*zh;                                            (:388-389)
//   DIV : z <- p / (x + eps)     RELU: z <- zh     PASS: z unchanged    (:396-430)
// `a` and `x` are clamped at 0 on load (A = relu(input) :359, X = relu(input) :327).
// The division is a true IEEE fp32 divide and ReLU zeros are exact: the eps = 1e-16 corner (x == 0 < a) is
// where the reference's values jump by 1e16, and it must jump identically here.
// ---------------------------------------------------------------------------------------------------------
// Scalar kernel (any HW), with optional P store and trace.
template <bool TRACE>
__global__ __launch_bounds__(NT) void ew_chain_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                     int accumulate, const EwChain ch, int C, int SB, int B, int HW,
                                                     float eps)
{
    const long per_c = (long)SB * HW;
    const long total = (long)C * per_c;
    for (long base = (long)blockIdx.x * NT; base < total; base += (long)gridDim.x * NT) {
        const long idx = base + threadIdx.x;
        const bool ok = idx < total;
        int c = 0, sb = 0;
        long aidx = 0;
        float g = 0.f;
        if (ok) {
            c = (int)(idx / per_c);
            const long r = idx - (long)c * per_c;
            sb = (int)(r / HW);
            const int hw = (int)(r - (long)sb * HW);
            const int b = sb % B;
            aidx = ((long)c * B + b) * HW + hw;
            if (ch.n > 0 && ch.s[0].type == EW_MAXHALF_IN) {
                const int Co = ch.s[0].action, cs = c % Co;
                const long arow = (long)B * HW, apos = (long)b * HW + hw;
                const float a = ch.s[0].p0[(long)cs * arow + apos], bb = ch.s[0].p0[(long)(cs + Co) * arow + apos];
                g = ew_maxhalf_route(src[(long)cs * per_c + r], c < Co ? a : bb, c < Co ? bb : a);
            } else if (ch.n > 0 && ch.s[0].type == EW_AVGUP_IN) {
                const EwStep& h = ch.s[0];
                const int W = h.prior_sb, PW = W >> 1, PHW = HW >> 2;
                const int ih = hw / W, iw = hw - ih * W;
                const long arow = ((long)c * B + b) * PHW;
                g = ew_avgup_pixel(h, src + ((long)c * SB + sb) * PHW, h.p0 ? h.p0 + arow : nullptr, h.p1 ? h.p1 + arow : nullptr,
                                   h.p2 ? h.p2 + ((long)c * SB + sb) * PHW : nullptr, (ih >> 1) * PW + (iw >> 1), !((ih | iw) & 1), eps);
            } else if (ch.n > 0 && ch.s[0].type == EW_POOL2_IN) {
                const int W = ch.s[0].action, OW = W >> 1, OHW = HW >> 2;
                const int ih = hw / W, iw = hw - ih * W;
                const int win = (ih >> 1) * OW + (iw >> 1);
                const uint8_t* ix = reinterpret_cast<const uint8_t*>(ch.s[0].p0) + ((long)c * B + b) * OHW;
                g = ew_pool2_route(src[((long)c * SB + sb) * OHW + win], (int)ix[win], ih & 1, iw & 1);
            } else {
                g = src[idx];
            }
        }
        float sv = 0.f;          // EW_STORE action 1 / 2: the value saved at a branch point
#pragma unroll 1
        for (int i = 0; i < ch.n; ++i) {
            const EwStep& st = ch.s[i];
            if (st.type == EW_HOOK) {
                float p = 0.f;
                if (ok && st.action >= HOOK_Q) {       // lean hooks (common.h): never observed, no priors
                    const float t = st.p0[aidx], z = fmaxf(g, 0.f);
                    g = st.action == HOOK_Q ? z * fabsf(t) : (st.action == HOOK_GATE ? (t > 0.f ? z : 0.f) : ((__float_as_uint(t) >> 31) ? 0.f : z));
                } else if (ok) {
                    const float a = fmaxf(st.p0[aidx], 0.f);
                    const float zh = fmaxf(g, 0.f);
                    p = a * zh;
                    const int hw_ = (int)((idx - (long)c * per_c) - (long)sb * HW);
                    const int el = c * HW + hw_;
                    int pel = -2;               // -2: no prior for this row, -1: the dense prior, >= 0: the one non-zero element
                    if (st.prior_dense) pel = (sb == st.prior_sb) ? -1 : -2;
                    else if (st.prior_elem) { const int pe = st.prior_elem[sb]; pel = pe >= 0 ? pe : -2; }
                    if (pel != -2) {
                        // layerwise EBP: p is overridden by the prior (whitebox.py:390-392)
                        const float pr = pel == -1 ? st.prior_dense[el] : (el == pel ? st.prior_val[sb] : 0.f);
                        p = pr;
                        if (st.pstore) st.pstore[idx] = p;
                        if (st.prior_action == PRIOR_DIV) {
                            const float x = st.p1 ? fmaxf(st.p1[aidx], 0.f) : a;
                            g = __fdiv_rn(p, x + eps);
                        } else if (st.prior_action == PRIOR_GATEZ) {
                            g = pr > 0.f ? g : 0.f;
                        }
                    } else {
                        if (st.pstore) st.pstore[idx] = p;
                        if (st.action == HOOK_DIV) {
                            const float x = st.p1 ? fmaxf(st.p1[aidx], 0.f) : a;
                            g = __fdiv_rn(p, x + eps);
                        } else if (st.action == HOOK_RELU) {
                            g = zh;
                        }
                    }
                    if (st.cap_dst && el == st.cap_elem[sb]) st.cap_dst[sb] = p;
                }
                if (TRACE && st.trace) {
                    // per-(stream,sample) sum of p; a block may straddle samples when HW < NT, so reduce per lane
                    // run only when the whole wave shares sb, else fall back to per-lane atomics.
                    const int sb0 = __shfl(sb, 0);
                    const bool uniform = __all(!ok || sb == sb0);
                    if (uniform) {
                        double v = ok ? (double)p : 0.0;
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
                        if ((threadIdx.x & 63) == 0 && __any(ok)) atomicAdd(&st.trace[sb0], v);
                    } else if (ok) {
                        atomicAdd(&st.trace[sb], (double)p);
                    }
                }
            } else if (ok) {
                if (st.type == EW_MASK) g = (st.action == 1 ? (__float_as_uint(st.p0[aidx]) >> 31) == 0u : st.p0[aidx] > 0.f) ? g : 0.f;
                else if (st.type == EW_SCALE_C) g = g * st.p0[c];
                else if (st.type == EW_SCALE) g = g * st.f;
                else if (st.type == EW_STORE) {
                    if (st.action == 1) sv = g;
                    else { st.pstore[idx] = g; if (st.action == 2) g = sv; }
                }
                else if (st.type == EW_ADDP) g += st.p0[idx];
                else if (st.type == EW_AFFINE_C) g = __fadd_rn(__fmul_rn(g, st.p0[c]), st.p1[c]);
                else if (st.type == EW_RELU) g = fmaxf(g, 0.f);
                else if (st.type == EW_MAXHALF_IN || st.type == EW_MAXPAIR || st.type == EW_MAXHALF_OUT || st.type == EW_POOL2_IN || st.type == EW_AVGUP_IN) { }      // applied at the load / compiled epilogues only
                else {
                    float v = __fadd_rn(__fmul_rn(fmaxf(g, 0.f), st.p0[c]), st.p1[c]);
                    if (st.p2) v = __fadd_rn((st.action & 1) ? fmaxf(st.p2[idx], 0.f) : st.p2[idx], v);
                    st.pstore[idx] = v;
                }
            }
        }
        if (ok) {
            if (accumulate) g += dst[idx];
            dst[idx] = g;
        }
    }
}

// float4 kernel: HW % 4 == 0, no trace.  blockIdx.y = channel, blockIdx.z = group of SG gradient streams; a thread owns one
// (sample, position) of the channel row and runs the chain for the SG streams at that position: the forward-side operands
// (a, x, masks: slots 0..2) are loaded once for all of them, the per-stream ones (g, the fan-in slot 3, dst) once each, all
// in flight before the steps are interpreted.  Contrastive EBP has two streams per sample, the layerwise sweeps up to 64.
template <bool PRIOR, int SG>     // PRIOR: the chain carries layerwise-EBP priors or captures (EwStep.prior_*, cap_*)
__global__ __launch_bounds__(NT) void ew_chain_kernel_v4(const float4* __restrict__ src, float4* __restrict__ dst,
                                                        int accumulate, const EwChain ch, const EwLoads ld, int C, int SB,
                                                        int B, int HW4, float eps, int SBa, int per_ca4)
{
    // per_ca4 = B*HW/4: one stream of the channel row in float4 pieces.  HW4 = HW/4 is only used to split a position into
    // (sample, hw) for priors and for stream prefixes; when HW % 4 != 0 but B*HW % 4 == 0 (7x7 maps at an even batch) a
    // piece may straddle two samples of the same stream, which the chain arithmetic does not care about
    const int c = blockIdx.y;
    const unsigned per_ca = (unsigned)per_ca4;
    const unsigned per_c = (unsigned)(SB / B) * per_ca;        // row stride; < 2^31 (every tensor is < 2^31 bytes)
    const unsigned pos = blockIdx.x * (unsigned)NT + threadIdx.x;
    if (pos >= per_ca) return;
    const long aidx = (long)c * per_ca + pos;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s0 = z4, s1 = z4, s2 = z4;
    if (ld.lp[0]) s0 = reinterpret_cast<const float4*>(ld.lp[0])[aidx];
    if (ld.lp[1]) s1 = reinterpret_cast<const float4*>(ld.lp[1])[aidx];
    if (ld.lp[2]) s2 = reinterpret_cast<const float4*>(ld.lp[2])[aidx];
    // chain head EW_POOL2_IN: this thread's four pixels lie in two 2x2 windows of one output row
    const bool head_pool2 = ch.n > 0 && ch.s[0].type == EW_POOL2_IN;
    // chain head EW_AVGUP_IN: the four pixels (any W: a piece may straddle two rows) take their window's pooled gradient
    const bool head_avgup = ch.n > 0 && ch.s[0].type == EW_AVGUP_IN;
    unsigned b = 0, hw = pos;
    if (PRIOR || SBa < SB || head_pool2 || head_avgup) { b = pos / (unsigned)HW4; hw = pos - b * (unsigned)HW4; }
    unsigned p2_win = 0, p2_ohw = 0;
    int p2_ph = 0, p2_id0 = 0, p2_id1 = 0;
    if (head_pool2) {
        const unsigned W4 = (unsigned)ch.s[0].action >> 2, OW = (unsigned)ch.s[0].action >> 1;
        const unsigned ih = hw / W4, iw4 = hw - ih * W4;
        p2_ohw = (unsigned)HW4;                                   // (H/2) * (W/2) = HW / 4
        p2_win = (ih >> 1) * OW + iw4 * 2u;                        // first of the two windows (even: 2-byte / 8-byte aligned pairs)
        p2_ph = (int)(ih & 1u);
        const uint16_t two = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(ch.s[0].p0) + ((size_t)c * B + b) * p2_ohw + p2_win);
        p2_id0 = two & 255;
        p2_id1 = two >> 8;
    }
    // chain head EW_MAXHALF_IN: the true forward halves of this position, shared by the gradient streams
    const bool head_maxhalf = ch.n > 0 && ch.s[0].type == EW_MAXHALF_IN;
    int cs = c;
    float4 own = z4, oth = z4;
    if (head_maxhalf) {
        const int Co = ch.s[0].action;
        cs = c % Co;
        const float4* tin = reinterpret_cast<const float4*>(ch.s[0].p0);
        const float4 ta = tin[(long)cs * per_ca + pos], tb = tin[(long)(cs + Co) * per_ca + pos];
        own = c < Co ? ta : tb;
        oth = c < Co ? tb : ta;
    }
    float4 g[SG], od[SG], v3[SG];
    long idx[SG];
    int sb[SG];
    bool ok[SG];
#pragma unroll
    for (int u = 0; u < SG; ++u) {
        const unsigned st = blockIdx.z * (unsigned)SG + u;                    // stream
        sb[u] = (int)(st * (unsigned)B + b);
        ok[u] = st * (unsigned)B < (unsigned)SB && sb[u] < SBa;
        idx[u] = (long)c * per_c + (ok[u] ? st * per_ca + pos : pos);
        if (head_maxhalf) {
            const float4 gs = src[idx[u] - (long)(c - cs) * per_c];           // the Co-channel gradient, row c % Co
            g[u] = make_float4(ew_maxhalf_route(gs.x, own.x, oth.x), ew_maxhalf_route(gs.y, own.y, oth.y),
                               ew_maxhalf_route(gs.z, own.z, oth.z), ew_maxhalf_route(gs.w, own.w, oth.w));
        } else if (head_avgup) {
            const EwStep& h = ch.s[0];
            const int W = h.prior_sb, PW = W >> 1, PHW = HW4;                  // (H / 2) * (W / 2) = HW / 4
            const size_t grow = ((size_t)c * SB + (ok[u] ? sb[u] : (int)b)) * PHW, arow = ((size_t)c * B + b) * PHW;
            const float* gp = reinterpret_cast<const float*>(src) + grow;
            const float* ap = h.p0 ? h.p0 + arow : nullptr;
            const float* xp = h.p1 ? h.p1 + arow : nullptr;
            const float* tp = h.p2 ? h.p2 + grow : nullptr;
            float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = (int)hw * 4 + q, ih = px / W, iw = px - ih * W;
                r[q] = ew_avgup_pixel(h, gp, ap, xp, tp, (ih >> 1) * PW + (iw >> 1), !((ih | iw) & 1), eps);
            }
            g[u] = make_float4(r[0], r[1], r[2], r[3]);
        } else if (head_pool2) {
            // the gradient of the pooled sum, [C][SB][H/2][W/2]: two windows = one aligned float2
            const float2 go = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(src) + ((size_t)c * SB + (ok[u] ? sb[u] : (int)b)) * p2_ohw + p2_win);
            g[u] = make_float4(ew_pool2_route(go.x, p2_id0, p2_ph, 0), ew_pool2_route(go.x, p2_id0, p2_ph, 1),
                               ew_pool2_route(go.y, p2_id1, p2_ph, 0), ew_pool2_route(go.y, p2_id1, p2_ph, 1));
        } else {
            g[u] = src[idx[u]];
        }
        v3[u] = z4;
        if (ld.lp[3]) v3[u] = reinterpret_cast<const float4*>(ld.lp[3])[idx[u]];
        od[u] = z4;
        if (accumulate) od[u] = dst[idx[u]];
    }
    const int el0 = (int)(((unsigned)c * (unsigned)HW4 + hw) * 4u);
#pragma unroll
    for (int u = 0; u < SG; ++u)
        ew_interpret<PRIOR>(ok[u], idx[u], aidx, sb[u], el0, g[u], od[u], s0, s1, s2, v3[u], dst, accumulate, ch, c, eps, (long)per_c, (long)per_ca);
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void nchw_to_cnhw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int N, int C, int HW)
{
    const long total = (long)N * C * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        // i indexes the output [C][N][HW]
        const int c = (int)(i / ((long)N * HW));
        const long r = i - (long)c * N * HW;
        const int n = (int)(r / HW);
        const int hw = (int)(r - (long)n * HW);
        out[i] = in[((long)n * C + c) * HW + hw];
    }
}

__global__ __launch_bounds__(NT) void u8hwc_to_cnhw_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int N, int Cnet, int HW, const U8Pre pre)
{
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int n = (int)(i / HW);
        const int hw = (int)(i - (long)n * HW);
        const uint8_t* px = in + i * pre.channels;
        if (pre.kind == 0) {
            for (int c = 0; c < Cnet; ++c) out[((long)c * N + n) * HW + hw] = (float)__dsub_rn((double)px[c], pre.mean[c]);
        } else {
            // (rgb / 255) @ w, left to right, no contraction
            double v = __dmul_rn(__ddiv_rn((double)px[0], 255.0), pre.weight[0]);
            v = __dadd_rn(v, __dmul_rn(__ddiv_rn((double)px[1], 255.0), pre.weight[1]));
            v = __dadd_rn(v, __dmul_rn(__ddiv_rn((double)px[2], 255.0), pre.weight[2]));
            out[(long)n * HW + hw] = (float)v;
        }
    }
}

__global__ __launch_bounds__(NT) void cnhw_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int N, int C, int HW)
{
    const long total = (long)N * C * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        // i indexes the output [N][C][HW]
        const int n = (int)(i / ((long)C * HW));
        const long r = i - (long)n * C * HW;
        const int c = (int)(r / HW);
        const int hw = (int)(r - (long)c * HW);
        out[i] = in[((long)c * N + n) * HW + hw];
    }
}

__global__ __launch_bounds__(NT) void affine_c_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     const float* __restrict__ alpha, const float* __restrict__ beta,
                                                     int C, long per_c, int relu_in, int relu_out)
{
    const long total = (long)C * per_c;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i / per_c);
        float v = in[i];
        if (relu_in) v = fmaxf(v, 0.f);
        // at::native batch_norm inference: out = x * (w * invstd) + (b - mean * w * invstd), no fused multiply-add
        v = __fadd_rn(__fmul_rn(v, alpha[c]), beta[c]);
        if (relu_out) v = fmaxf(v, 0.f);
        out[i] = v;
    }
}

constexpr int ST_U = 4;    // float4 elements per thread of the streaming kernels (all loads issued before the stores)

__global__ __launch_bounds__(NT) void affine_c_kernel_v4(const float4* __restrict__ in, float4* __restrict__ out,
                                                        const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        int C, long per_c4, int relu_in, int relu_out)
{
    const int c = blockIdx.y;
    const float al = alpha[c], be = beta[c];
    const long row = (long)c * per_c4;
    const long r0 = (long)blockIdx.x * (NT * ST_U) + threadIdx.x;
    float4 v[ST_U];
#pragma unroll
    for (int u = 0; u < ST_U; ++u)
        if (r0 + u * NT < per_c4) v[u] = in[row + r0 + u * NT];
#pragma unroll
    for (int u = 0; u < ST_U; ++u) {
        if (r0 + u * NT >= per_c4) continue;
        float q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = q[k];
            if (relu_in) t = fmaxf(t, 0.f);
            t = __fadd_rn(__fmul_rn(t, al), be);
            if (relu_out) t = fmaxf(t, 0.f);
            q[k] = t;
        }
        out[row + r0 + u * NT] = make_float4(q[0], q[1], q[2], q[3]);
    }
}

__global__ __launch_bounds__(NT) void scale_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float f,
                                                  int relu_in)
{
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        float v = in[i];
        if (relu_in) v = fmaxf(v, 0.f);
        out[i] = v * f;
    }
}

__global__ __launch_bounds__(NT) void add2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, long n, int relu_a, int relu_b, int relu_out)
{
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        float x = a[i], y = b[i];
        if (relu_a) x = fmaxf(x, 0.f);
        if (relu_b) y = fmaxf(y, 0.f);
        float v = x + y;
        if (relu_out) v = fmaxf(v, 0.f);
        out[i] = v;
    }
}

__global__ __launch_bounds__(NT) void add2_kernel_v4(const float4* __restrict__ a, const float4* __restrict__ b,
                                                    float4* __restrict__ out, long n4, int relu_a, int relu_b, int relu_out)
{
    const long i0 = (long)blockIdx.x * (NT * ST_U) + threadIdx.x;
    float4 xs[ST_U], ys[ST_U];
#pragma unroll
    for (int u = 0; u < ST_U; ++u)
        if (i0 + u * NT < n4) { xs[u] = a[i0 + u * NT]; ys[u] = b[i0 + u * NT]; }
#pragma unroll
    for (int u = 0; u < ST_U; ++u) {
        if (i0 + u * NT >= n4) continue;
        float4 x = xs[u], y = ys[u];
        if (relu_a) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        if (relu_b) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        out[i0 + u * NT] = v;
    }
}

__global__ __launch_bounds__(NT) void copy_acc_kernel(const float* __restrict__ src, float* __restrict__ dst, long n,
                                                     int accumulate)
{
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        float v = src[i];
        if (accumulate) v += dst[i];
        dst[i] = v;
    }
}

__global__ __launch_bounds__(NT) void copy_acc_kernel_v4(const float4* __restrict__ src, float4* __restrict__ dst, long n4, int accumulate)
{
    const long i0 = (long)blockIdx.x * (NT * ST_U) + threadIdx.x;
    float4 v[ST_U], d[ST_U];
#pragma unroll
    for (int u = 0; u < ST_U; ++u)
        if (i0 + u * NT < n4) { v[u] = src[i0 + u * NT]; if (accumulate) d[u] = dst[i0 + u * NT]; }
#pragma unroll
    for (int u = 0; u < ST_U; ++u) {
        if (i0 + u * NT >= n4) continue;
        float4 o = v[u];
        if (accumulate) { o.x += d[u].x; o.y += d[u].y; o.z += d[u].z; o.w += d[u].w; }
        dst[i0 + u * NT] = o;
    }
}

__global__ __launch_bounds__(NT) void fill_kernel(float* __restrict__ p, long n, float v)
{
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) p[i] = v;
}

// MaxPool2d forward: at::max_pool2d semantics -- the window is clipped to the input (padding = -inf), the first
// maximum in (kh, kw) scan order wins, NaN propagates.  idx stores the window-local offset dh*k+dw of the winner.
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        uint8_t* __restrict__ idx, int CN, int H, int W, int OH, int OW,
                                                        int k, int stride, int pad)
{
    const long total = (long)CN * OH * OW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int cn = (int)(i / ((long)OH * OW));
        const int r = (int)(i - (long)cn * OH * OW);
        const int oh = r / OW, ow = r - oh * OW;
        const float* src = in + (long)cn * H * W;
        float best = -INFINITY;
        int bi = 0;
        bool first = true;
        for (int dh = 0; dh < k; ++dh) {
            const int ih = oh * stride - pad + dh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int dw = 0; dw < k; ++dw) {
                const int iw = ow * stride - pad + dw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float v = src[ih * W + iw];
                if (first || v > best || v != v) { best = v; bi = dh * k + dw; first = false; }
            }
        }
        out[i] = best;
        if (idx) idx[i] = (uint8_t)bi;
    }
}

// OW % 4 == 0: a thread owns four consecutive outputs of one row and writes one float4 and one packed word of argmax bytes
template <int KK, int SS>
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel_v4(const float* __restrict__ in, float4* __restrict__ out, uint32_t* __restrict__ idx,
                                                           int H, int W, int OH, int OW, int k_rt, int stride_rt, int pad)
{
    const int k = KK ? KK : k_rt, stride = SS ? SS : stride_rt;
    const int plane = blockIdx.y;
    const int OW4 = OW >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= OH * OW4) return;
    const int oh = q / OW4, ow0 = (q - oh * OW4) * 4;
    const float* __restrict__ src = in + (size_t)plane * H * W;
    float best[4];
    int bi[4];
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int j = 0; j < 4; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    for (int dh = 0; dh < k; ++dh) {
        const int ih = oh * stride - pad + dh;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int dw = 0; dw < k; ++dw) {
                const int iw = (ow0 + j) * stride - pad + dw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float v = src[ih * W + iw];
                if (first[j] || v > best[j] || v != v) { best[j] = v; bi[j] = dh * k + dw; first[j] = false; }
            }
    }
    out[(size_t)plane * OH * OW4 + q] = make_float4(best[0], best[1], best[2], best[3]);
    if (idx) idx[(size_t)plane * OH * OW4 + q] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
}

// The pools of the backbones (3x3 stride 2 pad 1; 3x3 stride 2 pad 0 with ceil_mode; 2x2 stride 2 pad 0) on planes with W = 2 OW,
// W % 8 == 0: the eight input columns under a thread's four windows are two aligned float4 loads per row (plus the column left or
// right of them for the 3x3 pools) instead of one dword load per window element; candidates are compared in the same (dh, dw)
// order: first maximum wins, NaN sticks.
template <int KK, int PAD>
__global__ __launch_bounds__(NT) void maxpool_fwd_kernel_rows(const float* __restrict__ in, float4* __restrict__ out, uint32_t* __restrict__ idx,
                                                             int H, int W, int OH, int OW)
{
    const int plane = blockIdx.y;
    const int OW4 = OW >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= OH * OW4) return;
    const int oh = q / OW4, ow0 = (q - oh * OW4) * 4;
    const float* __restrict__ src = in + (size_t)plane * H * W;
    float best[4];
    int bi[4];
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int j = 0; j < 4; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    const bool left_ok = PAD == 1 && ow0 > 0;                          // column 2 ow0 - 1
    const bool right_ok = KK == 3 && PAD == 0 && 2 * ow0 + 8 < W;      // column 2 ow0 + 8
#pragma unroll
    for (int dh = 0; dh < KK; ++dh) {
        const int ih = oh * 2 - PAD + dh;
        if ((unsigned)ih >= (unsigned)H) continue;
        const float* row = src + (size_t)ih * W + 2 * ow0;
        const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
        float c[10];      // c[1 + i] = column 2 ow0 + i, i = -1 .. 8
        c[1] = a.x; c[2] = a.y; c[3] = a.z; c[4] = a.w; c[5] = b.x; c[6] = b.y; c[7] = b.z; c[8] = b.w;
        c[0] = left_ok ? row[-1] : 0.f;
        c[9] = right_ok ? row[8] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int dw = 0; dw < KK; ++dw) {
                const int ci = 1 + 2 * j - PAD + dw;
                if (ci == 0 && !left_ok) continue;
                if (ci == 9 && !right_ok) continue;
                const float v = c[ci];
                if (first[j] || v > best[j] || v != v) { best[j] = v; bi[j] = dh * KK + dw; first[j] = false; }
            }
    }
    out[(size_t)plane * OH * OW4 + q] = make_float4(best[0], best[1], best[2], best[3]);
    if (idx) idx[(size_t)plane * OH * OW4 + q] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
}

// lightcnn.py:252 `MaxPool2d(2)(x) + AvgPool2d(2)(x)` in one pass over x (W = 2 OW, W % 8 == 0): the true sum, the max-pool's argmax bytes, and
// -- where the consumer's hook divides by it -- the positive-pass sum relu(max) + avgpool(relu(x)).  A thread owns four windows of one output
// row (two aligned float4 loads per input row).  Same arithmetic, in the same order, as maxpool_fwd_kernel_rows<2, 0> (first maximum wins, NaN
// sticks), avgpool_fwd_kernel_v4<2, 2> (((0 + v00) + v01) + v10) + v11, times 0.25) and add2_kernel_v4 (max + avg): identical bits.
// pos_avg_mode: 0 = the positive average is the true one, 1 = relu(true average), 2 = average of relu(x) (x signed)
__global__ __launch_bounds__(NT) void pool2_fwd_kernel(const float* __restrict__ in, float4* __restrict__ out, uint32_t* __restrict__ idx,
                                                      float4* __restrict__ out_pos, int H, int W, int OH, int OW, int relu_max_pos, int pos_avg_mode)
{
    const int plane = blockIdx.y;
    const int OW4 = OW >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= OH * OW4) return;
    const int oh = q / OW4, ow0 = (q - oh * OW4) * 4;
    const float* __restrict__ row0 = in + (size_t)plane * H * W + (size_t)(2 * oh) * W + 2 * ow0;
    const float* __restrict__ row1 = row0 + W;
    const float4 a0 = *reinterpret_cast<const float4*>(row0), b0 = *reinterpret_cast<const float4*>(row0 + 4);
    const float4 a1 = *reinterpret_cast<const float4*>(row1), b1 = *reinterpret_cast<const float4*>(row1 + 4);
    const float c0[8] = {a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w};
    const float c1[8] = {a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w};
    float sum[4], psum[4];
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v[4] = {c0[2 * j], c0[2 * j + 1], c1[2 * j], c1[2 * j + 1]};
        float best = v[0];
        int bi = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t)
            if (v[t] > best || v[t] != v[t]) { best = v[t]; bi = t; }
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc += v[t];
        const float avg = acc * 0.25f;
        sum[j] = best + avg;
        packed |= (uint32_t)bi << (8 * j);
        if (out_pos) {
            float pavg = avg;
            if (pos_avg_mode == 1) pavg = fmaxf(avg, 0.f);
            else if (pos_avg_mode == 2) {
                float pacc = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) pacc += fmaxf(v[t], 0.f);
                pavg = pacc * 0.25f;
            }
            psum[j] = (relu_max_pos ? fmaxf(best, 0.f) : best) + pavg;
        }
    }
    const size_t o = (size_t)plane * OH * OW4 + q;
    out[o] = make_float4(sum[0], sum[1], sum[2], sum[3]);
    if (idx) idx[o] = packed;
    if (out_pos) out_pos[o] = make_float4(psum[0], psum[1], psum[2], psum[3]);
}

__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const float* __restrict__ gout, const uint8_t* __restrict__ idx,
                                                        float* __restrict__ gin, int accumulate, int C, int SB, int B,
                                                        int H, int W, int OH, int OW, int k, int stride, int pad)
{
    const long total = (long)C * SB * H * W;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int csb = (int)(i / ((long)H * W));
        const int r = (int)(i - (long)csb * H * W);
        const int ih = r / W, iw = r - ih * W;
        const int c = csb / SB, sb = csb - c * SB;
        const int b = sb % B;
        const uint8_t* ix = idx + ((long)c * B + b) * OH * OW;
        const float* go = gout + (long)csb * OH * OW;
        // output windows that contain (ih, iw): oh in [ceil((ih+pad-k+1)/stride), floor((ih+pad)/stride)]
        int oh_lo = ih + pad - k + 1;
        oh_lo = oh_lo > 0 ? (oh_lo + stride - 1) / stride : 0;
        int oh_hi = (ih + pad) / stride;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = iw + pad - k + 1;
        ow_lo = ow_lo > 0 ? (ow_lo + stride - 1) / stride : 0;
        int ow_hi = (iw + pad) / stride;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int dh = ih - (oh * stride - pad), dw = iw - (ow * stride - pad);
                if ((int)ix[oh * OW + ow] == dh * k + dw) acc += go[oh * OW + ow];
            }
        if (accumulate) acc += gin[i];
        gin[i] = acc;
    }
}

// AvgPool2d, no padding (resnet.py:186,211; resnet50_128.py pool5; lightcnn.py:237): mean over k*k
// zero_planes: that many further output planes behind the CN pooled ones are written as zeros (ConcatChannels of a down-sampling shortcut)
__global__ __launch_bounds__(NT) void avgpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int CN,
                                                        int H, int W, int OH, int OW, int k, int stride, int relu_in, int zero_planes)
{
    const long pooled = (long)CN * OH * OW;
    const long total = pooled + (long)zero_planes * OH * OW;
    const float inv = 1.0f / (float)(k * k);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        if (i >= pooled) { out[i] = 0.f; continue; }
        const int cn = (int)(i / ((long)OH * OW));
        const int r = (int)(i - (long)cn * OH * OW);
        const int oh = r / OW, ow = r - oh * OW;
        const float* src = in + (long)cn * H * W + (long)(oh * stride) * W + ow * stride;
        float acc = 0.f;
        for (int dh = 0; dh < k; ++dh)
            for (int dw = 0; dw < k; ++dw) {
                float v = src[dh * W + dw];
                if (relu_in) v = fmaxf(v, 0.f);
                acc += v;
            }
        out[i] = (k == 1) ? acc : acc * inv;
    }
}

__global__ __launch_bounds__(NT) void avgpool_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin,
                                                        int accumulate, int CN, int H, int W, int OH, int OW, int k,
                                                        int stride)
{
    const long total = (long)CN * H * W;
    const float inv = 1.0f / (float)(k * k);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int cn = (int)(i / ((long)H * W));
        const int r = (int)(i - (long)cn * H * W);
        const int ih = r / W, iw = r - ih * W;
        const float* go = gout + (long)cn * OH * OW;
        int oh_lo = ih - k + 1;
        oh_lo = oh_lo > 0 ? (oh_lo + stride - 1) / stride : 0;
        int oh_hi = ih / stride;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = iw - k + 1;
        ow_lo = ow_lo > 0 ? (ow_lo + stride - 1) / stride : 0;
        int ow_hi = iw / stride;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) acc += go[oh * OW + ow];
        acc = (k == 1) ? acc : acc * inv;
        if (accumulate) acc += gin[i];
        gin[i] = acc;
    }
}

// Global average pool (one window = the whole H x W plane, ResNet's 7x7 pool): a block stages GP consecutive planes through LDS with
// coalesced loads, then one thread per plane adds its HW values in the same row-major order as avgpool_fwd_kernel (bit-identical); the
// VJP hands every pixel (0 + g) / (H W) like the generic kernel, without its window arithmetic.
constexpr int GP = 128;
__global__ __launch_bounds__(NT) void avgpool_global_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int CN, int HW, int relu_in)
{
    extern __shared__ float tile[];
    const long p0 = (long)blockIdx.x * GP;
    const int np = (int)((CN - p0) < GP ? (CN - p0) : GP);
    const float* __restrict__ src = in + p0 * HW;
    for (int i = threadIdx.x; i < np * HW; i += NT) {
        const float v = src[i];
        tile[i] = relu_in ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
    if ((int)threadIdx.x < np) {
        const float inv = 1.0f / (float)HW;
        const float* t = tile + threadIdx.x * HW;
        float acc = 0.f;
        for (int i = 0; i < HW; ++i) acc += t[i];
        out[p0 + threadIdx.x] = (HW == 1) ? acc : acc * inv;
    }
}

__global__ __launch_bounds__(NT) void avgpool_global_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, int accumulate, long total, int HW)
{
    const float inv = 1.0f / (float)HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        float acc = 0.f;
        acc += gout[i / HW];
        acc = (HW == 1) ? acc : acc * inv;
        if (accumulate) acc += gin[i];
        gin[i] = acc;
    }
}

// float4 variants of the pool VJPs and the average pool: blockIdx.y = plane (channel x gradient row), a thread owns four consecutive
// pixels of one row, 32-bit index arithmetic only (the scalar kernels above pay a 64-bit division per element: the 3x3/2 max-pool
// VJP of a 32-triplet step -- 205 MB written -- ran at 0.9 TB/s).  W % 4 == 0.
template <int KK, int SS>     // compile-time window / stride (0: run-time values): the window bounds are divisions by the stride
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel_v4(const float* __restrict__ gout, const uint8_t* __restrict__ idx,
                                                           float4* __restrict__ gin, int accumulate, int SB, int B, int H, int W,
                                                           int OH, int OW, int k_rt, int stride_rt, int pad)
{
    const int k = KK ? KK : k_rt, stride = SS ? SS : stride_rt;
    const int plane = blockIdx.y;                       // c * SB + sb
    const int c = plane / SB, sb = plane - c * SB;
    const int W4 = W >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= H * W4) return;
    const int ih = q / W4, iw0 = (q - ih * W4) * 4;
    const uint8_t* __restrict__ ix = idx + (size_t)(c * B + sb % B) * OH * OW;
    const float* __restrict__ go = gout + (size_t)plane * OH * OW;
    int oh_lo = ih + pad - k + 1;
    oh_lo = oh_lo > 0 ? (oh_lo + stride - 1) / stride : 0;
    int oh_hi = (ih + pad) / stride;
    if (oh_hi > OH - 1) oh_hi = OH - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KK > 0 && SS > 0) {
        // the windows that can hold one of the four pixels: ow in [ow_lo(iw0), ow_hi(iw0 + 3)] -- at most NOW of them; their argmax
        // bytes and gradients are loaded once per row and handed to the pixels they point at
        constexpr int NOW = (KK + 2) / SS + 2;
        int ow_b = iw0 + pad - KK + 1;
        ow_b = ow_b > 0 ? (ow_b + SS - 1) / SS : 0;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const int dh = ih - (oh * SS - pad);
            int id[NOW];
            float g[NOW];
#pragma unroll
            for (int t = 0; t < NOW; ++t) {
                const int ow = ow_b + t;
                const bool in = ow < OW;
                id[t] = in ? (int)ix[oh * OW + ow] : -1;
                g[t] = in ? go[oh * OW + ow] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < NOW; ++t) {
                const int dw0 = iw0 - ((ow_b + t) * SS - pad);          // window-local column of pixel 0
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int dw = dw0 + j;
                    if (dw >= 0 && dw < KK && id[t] == dh * KK + dw) acc[j] += g[t];
                }
            }
        }
    } else {
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        const int dh = ih - (oh * stride - pad);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = iw0 + j;
            int ow_lo = iw + pad - k + 1;
            ow_lo = ow_lo > 0 ? (ow_lo + stride - 1) / stride : 0;
            int ow_hi = (iw + pad) / stride;
            if (ow_hi > OW - 1) ow_hi = OW - 1;
            for (int ow = ow_lo; ow <= ow_hi; ++ow)
                if ((int)ix[oh * OW + ow] == dh * k + (iw - (ow * stride - pad))) acc[j] += go[oh * OW + ow];
        }
    }
    }
    const size_t o = (size_t)plane * H * W4 + q;
    float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (accumulate) { const float4 d = gin[o]; v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
    gin[o] = v;
}

template <int KK, int SS>
__global__ __launch_bounds__(NT) void avgpool_bwd_kernel_v4(const float* __restrict__ gout, float4* __restrict__ gin, int accumulate,
                                                           int H, int W, int OH, int OW, int k_rt, int stride_rt)
{
    const int k = KK ? KK : k_rt, stride = SS ? SS : stride_rt;
    const int plane = blockIdx.y;
    const int W4 = W >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= H * W4) return;
    const int ih = q / W4, iw0 = (q - ih * W4) * 4;
    const float* __restrict__ go = gout + (size_t)plane * OH * OW;
    const float inv = 1.0f / (float)(k * k);
    int oh_lo = ih - k + 1;
    oh_lo = oh_lo > 0 ? (oh_lo + stride - 1) / stride : 0;
    int oh_hi = ih / stride;
    if (oh_hi > OH - 1) oh_hi = OH - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = iw0 + j;
            int ow_lo = iw - k + 1;
            ow_lo = ow_lo > 0 ? (ow_lo + stride - 1) / stride : 0;
            int ow_hi = iw / stride;
            if (ow_hi > OW - 1) ow_hi = OW - 1;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) acc[j] += go[oh * OW + ow];
        }
    }
    const size_t o = (size_t)plane * H * W4 + q;
    float4 v = (k == 1) ? make_float4(acc[0], acc[1], acc[2], acc[3]) : make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    if (accumulate) { const float4 d = gin[o]; v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
    gin[o] = v;
}

// OW % 4 == 0: a thread owns four consecutive outputs of one row
template <int KK, int SS>
__global__ __launch_bounds__(NT) void avgpool_fwd_kernel_v4(const float* __restrict__ in, float4* __restrict__ out, int H, int W, int OH,
                                                           int OW, int k_rt, int stride_rt, int relu_in, int CN)
{
    const int k = KK ? KK : k_rt, stride = SS ? SS : stride_rt;
    const int plane = blockIdx.y;
    const int OW4 = OW >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= OH * OW4) return;
    if (plane >= CN) { out[(size_t)plane * OH * OW4 + q] = make_float4(0.f, 0.f, 0.f, 0.f); return; }     // zero planes behind the pooled ones
    const int oh = q / OW4, ow0 = (q - oh * OW4) * 4;
    const float* __restrict__ src = in + (size_t)plane * H * W + (size_t)(oh * stride) * W;
    const float inv = 1.0f / (float)(k * k);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int dh = 0; dh < k; ++dh)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int dw = 0; dw < k; ++dw) {
                float v = src[dh * W + (ow0 + j) * stride + dw];
                if (relu_in) v = fmaxf(v, 0.f);
                acc[j] += v;
            }
    out[(size_t)plane * OH * OW4 + q] = (k == 1) ? make_float4(acc[0], acc[1], acc[2], acc[3])
                                                 : make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
}

// MaxFeatureMap glue: torch.max(split[0], split[1])  (lightcnn.py:62)
__global__ __launch_bounds__(NT) void maxhalves_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int Co,
                                                          long per_c, int relu_in)
{
    const long total = (long)Co * per_c;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        float a = in[i], b = in[i + total];
        if (relu_in) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
        out[i] = (a != a) ? a : ((b != b) ? b : fmaxf(a, b));
    }
}

__global__ __launch_bounds__(NT) void maxhalves_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ tin,
                                                          float* __restrict__ gin, int accumulate, int Co, int SB, int B,
                                                          int HW)
{
    const long per_c = (long)SB * HW;
    const long total = (long)Co * per_c;
    const long tper_c = (long)B * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i / per_c);
        const long r = i - (long)c * per_c;
        const int sb = (int)(r / HW);
        const int hw = (int)(r - (long)sb * HW);
        const long t = ((long)c * B + (sb % B)) * HW + hw;
        const float a = tin[t], b = tin[t + (long)Co * tper_c];
        const float g = gout[i];
        // at::maximum backward: grad/2 to both on ties, else all to the larger
        float ga = (a == b) ? g * 0.5f : (a > b ? g : 0.f);
        float gb = (a == b) ? g * 0.5f : (a < b ? g : 0.f);
        if (accumulate) { ga += gin[i]; gb += gin[i + total]; }
        gin[i] = ga;
        gin[i + total] = gb;
    }
}

// F.normalize(x, p=2, dim=1) on a [C][NB] tensor: x / max(||x||_2, 1e-12)   (resnet.py:250)
__global__ __launch_bounds__(64) void normalize_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          float* __restrict__ norms, int C, int NB, int relu_in)
{
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) { float v = in[(long)c * NB + n]; if (relu_in) v = fmaxf(v, 0.f); acc += v * v; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    acc = __shfl(acc, 0);
    const float nrm = fmaxf(sqrtf(acc), 1e-12f);
    if (lane == 0 && norms) norms[n] = nrm;
    for (int c = lane; c < C; c += 64) {
        float v = in[(long)c * NB + n];
        if (relu_in) v = fmaxf(v, 0.f);
        out[(long)c * NB + n] = v / nrm;
    }
}

// VJP of y = x / ||x||:  gx = (g - y * <g, y>) / ||x||    (norm above the clamp)
__global__ __launch_bounds__(64) void normalize_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ tin,
                                                          const float* __restrict__ norms, float* __restrict__ gin,
                                                          int accumulate, int C, int SB, int B)
{
    const int sb = blockIdx.x;
    const int b = sb % B;
    const int lane = threadIdx.x;
    const float nrm = norms[b];
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += gout[(long)c * SB + sb] * (tin[(long)c * B + b] / nrm);
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
    dot = __shfl(dot, 0);
    for (int c = lane; c < C; c += 64) {
        const float y = tin[(long)c * B + b] / nrm;
        float v = (gout[(long)c * SB + sb] - y * dot) / nrm;
        if (accumulate) v += gin[(long)c * SB + sb];
        gin[(long)c * SB + sb] = v;
    }
}

__global__ __launch_bounds__(NT) void seed_to_cnhw_kernel(const float* __restrict__ seed, float* __restrict__ g, int SB,
                                                         int C, int HW)
{
    const long total = (long)SB * C * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i / ((long)SB * HW));
        const long r = i - (long)c * SB * HW;
        const int sb = (int)(r / HW);
        const int hw = (int)(r - (long)sb * HW);
        g[i] = seed[((long)sb * C + c) * HW + hw];
    }
}

// weighted-subtree layer weights (whitebox.py:689-690): max / first argmax of (gm >= 0) * (-gn) per sample
struct StatPartial { float v; int i; };

__global__ __launch_bounds__(NT) void subtree_stats_kernel(const StatDesc* __restrict__ desc, StatPartial* __restrict__ part,
                                                          int N, int chunks, int gate_ge0)
{
    const int n = blockIdx.y, u = blockIdx.z;
    const float* __restrict__ G = desc[u].G;
    const int C = desc[u].C, HW = desc[u].HW;
    const long total = (long)C * HW;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)chunks * NT) {
        const int c = (int)(e / HW);
        const int hw = (int)(e - (long)c * HW);
        const float gm = G[((long)c * 2 * N + n) * HW + hw];
        const float gn = G[((long)c * 2 * N + N + n) * HW + hw];
        const float v = ((gate_ge0 ? gm >= 0.f : gm < 0.f) ? 1.f : 0.f) * (-gn);
        if (v > best || (v == best && (int)e < bi)) { best = v; bi = (int)e; }
    }
    __shared__ float sv[NT];
    __shared__ int si[NT];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v2 = sv[threadIdx.x + o];
            const int i2 = si[threadIdx.x + o];
            if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        StatPartial& q = part[((long)u * N + n) * chunks + blockIdx.x];
        q.v = sv[0];
        q.i = si[0];
    }
}

// one thread per (firing, sample): several hooks on one tensor see the same gradient
__global__ void subtree_stats_final_kernel(const StatPartial* __restrict__ part, const int* __restrict__ f2u, float* __restrict__ vmax,
                                           int* __restrict__ vidx, int chunks, int N, int n_firings)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_firings * N) return;
    const int f = t / N, n = t - f * N;
    const int u = f2u[f];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = 0; k < chunks; ++k) {
        const StatPartial q = part[((long)u * N + n) * chunks + k];
        if (q.v > best || (q.v == best && q.i < bi)) { best = q.v; bi = q.i; }
    }
    vmax[t] = best;
    vidx[t] = bi;
}

constexpr int STAT_CHUNKS = 16;

}  // namespace

size_t subtree_stats_scratch_bytes(int N, int n_tensors) { return sizeof(StatPartial) * (size_t)N * STAT_CHUNKS * (size_t)n_tensors; }

void launch_subtree_stats(const StatDesc* desc_dev, int n_tensors, const int* f2u_dev, int n_firings, float* vmax, int* vidx,
                          void* scratch, int N, int gate_ge0, hipStream_t s)
{
    StatPartial* part = reinterpret_cast<StatPartial*>(scratch);
    hipLaunchKernelGGL(subtree_stats_kernel, dim3(STAT_CHUNKS, N, n_tensors), dim3(NT), 0, s, desc_dev, part, N, STAT_CHUNKS, gate_ge0);
    const int total = n_firings * N;
    hipLaunchKernelGGL(subtree_stats_final_kernel, dim3((total + 63) / 64), dim3(64), 0, s, part, f2u_dev, vmax, vidx, STAT_CHUNKS, N,
                       n_firings);
}

void launch_ew_chain(const float* src, float* dst, int accumulate, const EwChain& chain, int C, int SB, int B, int HW,
                     float eps, hipStream_t s, int SBa)
{
    if (SBa < 0 || SBa > SB) SBa = SB;
    bool trace = false, special = false, prior = false;
    for (int i = 0; i < chain.n; ++i) {
        if (chain.s[i].type != EW_HOOK) continue;
        if (chain.s[i].trace) trace = true;
        if (chain.s[i].prior_elem || chain.s[i].prior_dense || chain.s[i].cap_dst) prior = true;
    }
    const long total = (long)C * SB * HW;
    for (int i = 0; i < chain.n; ++i)
        if (accumulate && chain.s[i].pstore == dst) special = true;   // the float4 kernel reads dst before the chain runs
    // (a chain with a fan-out -- EW_MAXHALF_OUT behind a pool-pair head -- only exists in the float4 interpreter; the planner creates it
    // for tensors with HW % 4 == 0 in the schedule that carries no traces and no priors, never accumulating)
    // float4 pieces: whole samples when HW % 4 == 0; otherwise (7x7 maps) pieces of the flat [B*HW] stream row, which is
    // fine as long as nothing needs the sample of a piece (no priors, no stream prefix)
    const bool pieces_ok = (HW % 4) == 0 || (((long)B * HW) % 4 == 0 && !prior && SBa == SB);
    if (!trace && !special && pieces_ok && C <= 65535 && SB % B == 0 && SB / B <= 2 * 65535) {
        EwLoads ld;
        EwChain planned = chain;
        ew_plan_loads(planned, dst, ld);
        const int S = SB / B, Sa = (SBa + B - 1) / B;                    // streams, streams with live samples
        const long per_ca4 = (long)B * HW / 4;
        const unsigned gx = (unsigned)((per_ca4 + NT - 1) / NT);
        auto go = [&](auto kern, int sg) {
            hipLaunchKernelGGL(kern, dim3(gx, C, (unsigned)((Sa + sg - 1) / sg)), dim3(NT), 0, s, reinterpret_cast<const float4*>(src),
                               reinterpret_cast<float4*>(dst), accumulate, planned, ld, C, SB, B, HW / 4, eps, SBa, (int)per_ca4);
        };
        if (S == 1) { if (prior) go(ew_chain_kernel_v4<true, 1>, 1); else go(ew_chain_kernel_v4<false, 1>, 1); }
        else { if (prior) go(ew_chain_kernel_v4<true, 2>, 2); else go(ew_chain_kernel_v4<false, 2>, 2); }
    } else if (trace) {
        hipLaunchKernelGGL(ew_chain_kernel<true>, dim3(grid_for(total)), dim3(NT), 0, s, src, dst, accumulate, chain, C, SB,
                           B, HW, eps);
    } else {
        hipLaunchKernelGGL(ew_chain_kernel<false>, dim3(grid_for(total)), dim3(NT), 0, s, src, dst, accumulate, chain, C, SB,
                           B, HW, eps);
    }
}

void launch_nchw_to_cnhw(const float* in, float* out, int N, int C, int HW, hipStream_t s)
{
    hipLaunchKernelGGL(nchw_to_cnhw_kernel, dim3(grid_for((long)N * C * HW)), dim3(NT), 0, s, in, out, N, C, HW);
}
void launch_u8hwc_to_cnhw(const uint8_t* in, float* out, int N, int Cnet, int HW, const U8Pre& pre, hipStream_t s)
{
    hipLaunchKernelGGL(u8hwc_to_cnhw_kernel, dim3(grid_for((long)N * HW)), dim3(NT), 0, s, in, out, N, Cnet, HW, pre);
}
void launch_cnhw_to_nchw(const float* in, float* out, int N, int C, int HW, hipStream_t s)
{
    hipLaunchKernelGGL(cnhw_to_nchw_kernel, dim3(grid_for((long)N * C * HW)), dim3(NT), 0, s, in, out, N, C, HW);
}
void launch_affine_c(const float* in, float* out, const float* alpha, const float* beta, int C, long per_c, int relu_in,
                     int relu_out, hipStream_t s)
{
    if ((per_c % 4) == 0 && C <= 65535)
        hipLaunchKernelGGL(affine_c_kernel_v4, dim3((unsigned)((per_c / 4 + NT * ST_U - 1) / (NT * ST_U)), C), dim3(NT), 0, s,
                           reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), alpha, beta, C, per_c / 4,
                           relu_in, relu_out);
    else
        hipLaunchKernelGGL(affine_c_kernel, dim3(grid_for((long)C * per_c)), dim3(NT), 0, s, in, out, alpha, beta, C, per_c,
                           relu_in, relu_out);
}
void launch_relu(const float* in, float* out, long n, hipStream_t s)
{
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(NT), 0, s, in, out, n, 1.0f, 1);
}
void launch_scale(const float* in, float* out, long n, float f, int relu_in, hipStream_t s)
{
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(NT), 0, s, in, out, n, f, relu_in);
}
void launch_add2(const float* a, const float* b, float* out, long n, int relu_a, int relu_b, int relu_out, hipStream_t s)
{
    if ((n % 4) == 0)
        hipLaunchKernelGGL(add2_kernel_v4, dim3((unsigned)((n / 4 + NT * ST_U - 1) / (NT * ST_U))), dim3(NT), 0, s, reinterpret_cast<const float4*>(a),
                           reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n / 4, relu_a, relu_b, relu_out);
    else
        hipLaunchKernelGGL(add2_kernel, dim3(grid_for(n)), dim3(NT), 0, s, a, b, out, n, relu_a, relu_b, relu_out);
}
void launch_copy_acc(const float* src, float* dst, long n, int accumulate, hipStream_t s)
{
    if ((n & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        hipLaunchKernelGGL(copy_acc_kernel_v4, dim3((unsigned)((n / 4 + NT * ST_U - 1) / (NT * ST_U))), dim3(NT), 0, s, reinterpret_cast<const float4*>(src),
                           reinterpret_cast<float4*>(dst), n / 4, accumulate);
        return;
    }
    hipLaunchKernelGGL(copy_acc_kernel, dim3(grid_for(n)), dim3(NT), 0, s, src, dst, n, accumulate);
}
void launch_fill(float* p, long n, float v, hipStream_t s)
{
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(NT), 0, s, p, n, v);
}
void launch_maxpool_fwd(const float* in, float* out, uint8_t* idx, int CN, int H, int W, int OH, int OW, int k, int stride,
                        int pad, hipStream_t s)
{
    if ((OW & 3) == 0 && CN <= 65535 && (((uintptr_t)idx) & 3) == 0) {
        const dim3 g((OH * (OW / 4) + NT - 1) / NT, CN);
        float4* out4 = reinterpret_cast<float4*>(out);
        uint32_t* idx4 = reinterpret_cast<uint32_t*>(idx);
        const bool rows = stride == 2 && W == 2 * OW && (W & 7) == 0 && (((uintptr_t)in) & 15) == 0 && ((k == 3 && pad <= 1) || (k == 2 && pad == 0));
        if (rows && k == 3 && pad == 1) hipLaunchKernelGGL((maxpool_fwd_kernel_rows<3, 1>), g, dim3(NT), 0, s, in, out4, idx4, H, W, OH, OW);
        else if (rows && k == 3) hipLaunchKernelGGL((maxpool_fwd_kernel_rows<3, 0>), g, dim3(NT), 0, s, in, out4, idx4, H, W, OH, OW);
        else if (rows) hipLaunchKernelGGL((maxpool_fwd_kernel_rows<2, 0>), g, dim3(NT), 0, s, in, out4, idx4, H, W, OH, OW);
        else if (k == 3 && stride == 2) hipLaunchKernelGGL((maxpool_fwd_kernel_v4<3, 2>), g, dim3(NT), 0, s, in, out4, idx4, H, W, OH, OW, k, stride, pad);
        else if (k == 2 && stride == 2) hipLaunchKernelGGL((maxpool_fwd_kernel_v4<2, 2>), g, dim3(NT), 0, s, in, out4, idx4, H, W, OH, OW, k, stride, pad);
        else hipLaunchKernelGGL((maxpool_fwd_kernel_v4<0, 0>), g, dim3(NT), 0, s, in, out4, idx4, H, W, OH, OW, k, stride, pad);
        return;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)CN * OH * OW)), dim3(NT), 0, s, in, out, idx, CN, H, W, OH, OW,
                       k, stride, pad);
}
// ---- Light-CNN's first layer as a direct convolution (lightcnn.py:249: mfm(1, 48, 5, 1, 2)) --------------------------------------------------
// One input channel, K = 25: as an implicit GEMM the layer is all prologue and epilogue (a 64x64 tile holds 16 MFMAs per wave; round 3: 653 us for
// 10 GFLOP where its 1.2 GB of output need ~250 us).  Here a thread owns four consecutive pixels, keeps their 5 x 8 input patch in registers and
// walks the channel PAIRS of the MaxFeatureMap: 25 fused multiply-adds per output in the GEMM's K order (k = dh * 5 + dw, accumulator from 0, then
// the bias) -- v_mfma_f32_32x32x2_f32 is an fmaf chain, so the bits are the GEMM's -- with the pair's weights as scalar operands; it stores the raw
// rows where the Split hook and the VJP expect them (keep_raw) and max(row, partner) like the EW_MAXPAIR epilogue (NaN propagates).  Weights and bias
// are read from the GEMM's own pack: wp[k * ldw + r], row r = 2c + half (channel c + half * Co).
__global__ __launch_bounds__(NT) void stem5_mfm_kernel(const float* __restrict__ in, const float* __restrict__ wp, int ldw, const float* __restrict__ bias,
                                                      float* __restrict__ raw, float* __restrict__ omax, int Co, int NB, int H, int W)
{
    const int n = blockIdx.y;
    const int W4 = W >> 2;
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= H * W4) return;
    const int h = q / W4, w0 = (q - h * W4) * 4;
    const float* __restrict__ img = in + (size_t)n * H * W;
    float x[5][8];                                   // x[dh][j] = pixel (h + dh - 2, w0 - 2 + j), 0 outside the image (padding 2)
#pragma unroll
    for (int dh = 0; dh < 5; ++dh) {
        const int ih = h + dh - 2;
        const bool rok = (unsigned)ih < (unsigned)H;
        const float* row = img + (size_t)(rok ? ih : 0) * W + w0;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 l = (rok && w0 > 0) ? *reinterpret_cast<const float4*>(row - 4) : z;
        const float4 m = rok ? *reinterpret_cast<const float4*>(row) : z;
        const float4 r = (rok && w0 + 4 < W) ? *reinterpret_cast<const float4*>(row + 4) : z;
        x[dh][0] = l.z; x[dh][1] = l.w; x[dh][2] = m.x; x[dh][3] = m.y; x[dh][4] = m.z; x[dh][5] = m.w; x[dh][6] = r.x; x[dh][7] = r.y;
    }
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)n * plane + (size_t)h * W + w0;
#pragma unroll 1
    for (int c = 0; c < Co; ++c) {
        float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dh = 0; dh < 5; ++dh)
#pragma unroll
            for (int dw = 0; dw < 5; ++dw) {
                const float wl = wp[(size_t)(dh * 5 + dw) * ldw + 2 * c], wh = wp[(size_t)(dh * 5 + dw) * ldw + 2 * c + 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = __builtin_fmaf(wl, x[dh][j + dw], lo[j]);
                    hi[j] = __builtin_fmaf(wh, x[dh][j + dw], hi[j]);
                }
            }
        if (bias) {
            const float bl = bias[2 * c], bh = bias[2 * c + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] += bl; hi[j] += bh; }
        }
        if (raw) {
            *reinterpret_cast<float4*>(raw + (size_t)c * NB * plane + pix) = make_float4(lo[0], lo[1], lo[2], lo[3]);
            *reinterpret_cast<float4*>(raw + (size_t)(c + Co) * NB * plane + pix) = make_float4(hi[0], hi[1], hi[2], hi[3]);
        }
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = (lo[j] != lo[j]) ? lo[j] : ((hi[j] != hi[j]) ? hi[j] : fmaxf(lo[j], hi[j]));
        *reinterpret_cast<float4*>(omax + (size_t)c * NB * plane + pix) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// Whitebox.P[-1] (whitebox.py:394, the hook on the FIRST convolution's input): p = relu(image) * relu(z), z = the backward-data pass of that
// convolution with relu(W) on the gradient that leaves its output's hooks.  Nothing on the hot path reads it (the saliency tap is P[-2]), so this is
// a plain gather kernel run on demand: a thread owns one input pixel of one channel and walks the output positions whose window covers it.
// g: [Cout][SB][OH][OW] (SB == N here), wp: the forward pack of relu(W) ([K][ldw], k_of(tap, ci) per kmode: 0 = (ci, kh, kw), 1 = (kh, kw, ci),
// 2 = (kh, kw, 4 channel slots); column co_pair ? interleaved halves : co), img: [Cin][N][H][W], out: NCHW.
__global__ __launch_bounds__(NT) void image_mwp_kernel(const float* __restrict__ g, const float* __restrict__ wp, const float* __restrict__ img,
                                                      float* __restrict__ out, int Cin, int N, int H, int W, int Cout, int OH, int OW, int kh,
                                                      int kw, int stride, int pad, int ldw, int kmode, int co_pair)
{
    const long total = (long)N * Cin * H * W;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int iw = (int)(i % W);
        const int ih = (int)((i / W) % H);
        const int ci = (int)((i / ((long)W * H)) % Cin);
        const int n = (int)(i / ((long)W * H * Cin));
        float z = 0.f;
        for (int dh = 0; dh < kh; ++dh) {
            const int t = ih + pad - dh;
            if (t < 0 || t % stride) continue;
            const int oh = t / stride;
            if (oh >= OH) continue;
            for (int dw = 0; dw < kw; ++dw) {
                const int u = iw + pad - dw;
                if (u < 0 || u % stride) continue;
                const int ow = u / stride;
                if (ow >= OW) continue;
                const int tap = dh * kw + dw;
                const long k = kmode == 2 ? (long)tap * 4 + ci : (kmode == 1 ? (long)tap * Cin + ci : (long)ci * kh * kw + tap);
                const float* wrow = wp + k * ldw;
                const float* gp = g + ((long)n * OH + oh) * OW + ow;
                for (int co = 0; co < Cout; ++co) {
                    const int col = co_pair ? (co % co_pair) * 2 + co / co_pair : co;
                    z = __builtin_fmaf(gp[(long)co * N * OH * OW], wrow[col], z);
                }
            }
        }
        const float a = fmaxf(img[((long)ci * N + n) * H * W + (long)ih * W + iw], 0.f);
        out[i] = a * fmaxf(z, 0.f);
    }
}

bool pool2_fwd_ok(const float* in, const uint8_t* idx, int CN, int H, int W, int OH, int OW)
{
    return W == 2 * OW && H == 2 * OH && (W & 7) == 0 && CN <= 65535 && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)idx) & 3) == 0;
}
void launch_image_mwp(const float* g, const float* wp, const float* img, float* out, int Cin, int N, int H, int W, int Cout, int OH, int OW,
                      int kh, int kw, int stride, int pad, int ldw, int kmode, int co_pair, hipStream_t s)
{
    hipLaunchKernelGGL(image_mwp_kernel, dim3(grid_for((long)N * Cin * H * W)), dim3(NT), 0, s, g, wp, img, out, Cin, N, H, W, Cout, OH, OW, kh,
                       kw, stride, pad, ldw, kmode, co_pair);
}
bool stem5_mfm_ok(const float* in, int NB, int H, int W)
{
    return (W & 3) == 0 && NB <= 65535 && (((uintptr_t)in) & 15) == 0;
}
void launch_stem5_mfm(const float* in, const float* wp, int ldw, const float* bias, float* raw, float* omax, int Co, int NB, int H, int W, hipStream_t s)
{
    hipLaunchKernelGGL(stem5_mfm_kernel, dim3((H * (W / 4) + NT - 1) / NT, NB), dim3(NT), 0, s, in, wp, ldw, bias, raw, omax, Co, NB, H, W);
}
void launch_pool2_fwd(const float* in, float* out_sum, uint8_t* idx, float* out_pos, int CN, int H, int W, int OH, int OW, int relu_max_pos,
                      int pos_avg_mode, hipStream_t s)
{
    const dim3 g((OH * (OW / 4) + NT - 1) / NT, CN);
    hipLaunchKernelGGL(pool2_fwd_kernel, g, dim3(NT), 0, s, in, reinterpret_cast<float4*>(out_sum), reinterpret_cast<uint32_t*>(idx),
                       reinterpret_cast<float4*>(out_pos), H, W, OH, OW, relu_max_pos, pos_avg_mode);
}
void launch_maxpool_bwd(const float* gout, const uint8_t* idx, float* gin, int accumulate, int C, int SB, int B, int H, int W,
                        int OH, int OW, int k, int stride, int pad, hipStream_t s)
{
    if ((W & 3) == 0 && (long)C * SB <= 65535) {
        const dim3 g((H * (W / 4) + NT - 1) / NT, C * SB);
        float4* gin4 = reinterpret_cast<float4*>(gin);
        if (k == 3 && stride == 2) hipLaunchKernelGGL((maxpool_bwd_kernel_v4<3, 2>), g, dim3(NT), 0, s, gout, idx, gin4, accumulate, SB, B, H, W, OH, OW, k, stride, pad);
        else if (k == 2 && stride == 2) hipLaunchKernelGGL((maxpool_bwd_kernel_v4<2, 2>), g, dim3(NT), 0, s, gout, idx, gin4, accumulate, SB, B, H, W, OH, OW, k, stride, pad);
        else hipLaunchKernelGGL((maxpool_bwd_kernel_v4<0, 0>), g, dim3(NT), 0, s, gout, idx, gin4, accumulate, SB, B, H, W, OH, OW, k, stride, pad);
        return;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)C * SB * H * W)), dim3(NT), 0, s, gout, idx, gin, accumulate,
                       C, SB, B, H, W, OH, OW, k, stride, pad);
}
void launch_avgpool_fwd(const float* in, float* out, int CN, int H, int W, int OH, int OW, int k, int stride, int relu_in,
                        hipStream_t s, int zero_planes)
{
    if (zero_planes == 0 && OH == 1 && OW == 1 && k == H && k == W && (size_t)GP * H * W * sizeof(float) <= 64 * 1024) {
        hipLaunchKernelGGL(avgpool_global_fwd_kernel, dim3((CN + GP - 1) / GP), dim3(NT), (size_t)GP * H * W * sizeof(float), s, in, out, CN, H * W, relu_in);
        return;
    }
    if ((OW & 3) == 0 && CN + zero_planes <= 65535) {
        const dim3 g((OH * (OW / 4) + NT - 1) / NT, CN + zero_planes);
        float4* out4 = reinterpret_cast<float4*>(out);
        if (k == 2 && stride == 2) hipLaunchKernelGGL((avgpool_fwd_kernel_v4<2, 2>), g, dim3(NT), 0, s, in, out4, H, W, OH, OW, k, stride, relu_in, CN);
        else hipLaunchKernelGGL((avgpool_fwd_kernel_v4<0, 0>), g, dim3(NT), 0, s, in, out4, H, W, OH, OW, k, stride, relu_in, CN);
        return;
    }
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long)(CN + zero_planes) * OH * OW)), dim3(NT), 0, s, in, out, CN, H, W, OH, OW, k,
                       stride, relu_in, zero_planes);
}
void launch_avgpool_bwd(const float* gout, float* gin, int accumulate, int CN, int H, int W, int OH, int OW, int k, int stride,
                        hipStream_t s)
{
    if (OH == 1 && OW == 1 && k == H && k == W) {
        const long total = (long)CN * H * W;
        hipLaunchKernelGGL(avgpool_global_bwd_kernel, dim3(grid_for(total)), dim3(NT), 0, s, gout, gin, accumulate, total, H * W);
        return;
    }
    if ((W & 3) == 0 && CN <= 65535) {
        const dim3 g((H * (W / 4) + NT - 1) / NT, CN);
        float4* gin4 = reinterpret_cast<float4*>(gin);
        if (k == 2 && stride == 2) hipLaunchKernelGGL((avgpool_bwd_kernel_v4<2, 2>), g, dim3(NT), 0, s, gout, gin4, accumulate, H, W, OH, OW, k, stride);
        else hipLaunchKernelGGL((avgpool_bwd_kernel_v4<0, 0>), g, dim3(NT), 0, s, gout, gin4, accumulate, H, W, OH, OW, k, stride);
        return;
    }
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)CN * H * W)), dim3(NT), 0, s, gout, gin, accumulate, CN, H, W,
                       OH, OW, k, stride);
}
void launch_maxhalves_fwd(const float* in, float* out, int Co, long per_c, int relu_in, hipStream_t s)
{
    hipLaunchKernelGGL(maxhalves_fwd_kernel, dim3(grid_for((long)Co * per_c)), dim3(NT), 0, s, in, out, Co, per_c, relu_in);
}
void launch_maxhalves_bwd(const float* gout, const float* tin, float* gin, int accumulate, int Co, int SB, int B, int HW,
                          hipStream_t s)
{
    hipLaunchKernelGGL(maxhalves_bwd_kernel, dim3(grid_for((long)Co * SB * HW)), dim3(NT), 0, s, gout, tin, gin, accumulate,
                       Co, SB, B, HW);
}
void launch_normalize_fwd(const float* in, float* out, float* norms, int C, int NB, int relu_in, hipStream_t s)
{
    hipLaunchKernelGGL(normalize_fwd_kernel, dim3(NB), dim3(64), 0, s, in, out, norms, C, NB, relu_in);
}
void launch_normalize_bwd(const float* gout, const float* tin, const float* norms, float* gin, int accumulate, int C, int SB,
                          int B, hipStream_t s)
{
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(SB), dim3(64), 0, s, gout, tin, norms, gin, accumulate, C, SB, B);
}
void launch_seed_to_cnhw(const float* seed, float* g, int SB, int C, int HW, hipStream_t s)
{
    hipLaunchKernelGGL(seed_to_cnhw_kernel, dim3(grid_for((long)SB * C * HW)), dim3(NT), 0, s, seed, g, SB, C, HW);
}
