// Shared device code: the float4 interpreter of a hook-chain micro-program (EwChain), used by the stand-alone chain kernel
// (elementwise.hip) and by the chain epilogue of the convolution GEMM (conv_gemm.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ float pick1(float a0, float a1, float a2, float a3, int slot)
{
    float r = a0;
    r = slot == 1 ? a1 : r;
    r = slot == 2 ? a2 : r;
    r = slot == 3 ? a3 : r;
    return r;
}
// component-wise on plain floats: a select between float4 objects is lowered through memory (scratch)
__device__ __forceinline__ float4 pick_slot(float4 v0, float4 v1, float4 v2, float4 v3, int slot)
{
    return make_float4(pick1(v0.x, v1.x, v2.x, v3.x, slot), pick1(v0.y, v1.y, v2.y, v3.y, slot),
                       pick1(v0.z, v1.z, v2.z, v3.z, slot), pick1(v0.w, v1.w, v2.w, v3.w, slot));
}

// chain head EW_POOL2_IN: gradient of pixel (row parity ph, column parity pw) of a 2x2 window whose summed max-pool + average-pool output has
// gradient go and whose argmax byte is id: the average pool's share (0.f + go) * 0.25f plus the max pool's (all of go to the argmax)
__device__ __forceinline__ float ew_pool2_route(float go, int id, int ph, int pw)
{
    return (id == ph * 2 + pw ? go : 0.f) + go * 0.25f;
}

// chain head EW_AVGUP_IN: one pixel.  gp / tp: the pooled gradient and the compact GEMM result at this pixel's window (tp only read on the even
// pixels), ap / xp: the pooled tensor's hook operands at the window (EW_HOOK's arithmetic, one value)
__device__ __forceinline__ float ew_avgup_pixel(const EwStep& st, const float* __restrict__ gp, const float* __restrict__ ap, const float* __restrict__ xp,
                                                const float* __restrict__ tp, int pidx, bool on_grid, float eps)
{
    if (st.action == -2) return on_grid ? tp[pidx] : 0.f;      // no pooled source (both contributions are strided GEMMs): the zero fill + scatters
    float g = gp[pidx];
    if (st.action == HOOK_RELU && !ap) g = fmaxf(g, 0.f);
    else if (st.action >= 0 && ap) {
        const float a = fmaxf(ap[pidx], 0.f);
        const float zh = fmaxf(g, 0.f);
        const float pr = a * zh;
        if (st.action == HOOK_DIV) {
            const float x = xp ? fmaxf(xp[pidx], 0.f) : a;
            g = __fdiv_rn(pr, x + eps);
        } else if (st.action == HOOK_RELU) g = zh;
    }
    g = (0.f + g) * 0.25f;                      // avgpool_bwd_kernel: acc = 0; acc += g; acc * (1 / 4)
    if (on_grid && tp) g = tp[pidx] + g;        // the scattering GEMM's read-modify-write: v = acc + old
    return g;
}

// VJP of max(a, b) for the half that holds `own`: all of g to the larger input, half of it on ties (at::maximum backward)
__device__ __forceinline__ float ew_maxhalf_route(float g, float own, float other)
{
    return own == other ? g * 0.5f : (own > other ? g : 0.f);
}

// steps [i0, n) of a chain on one float4 piece; returns the index of an EW_MAXHALF_OUT step it stopped at (the fan-out: the caller routes the
// gradient to the two halves and runs the rest per half), or n
template <bool PRIOR>
__device__ __forceinline__ int ew_steps(int i0, float (&g)[4], long idx, long aidx, int sb, int el0, float4 v0, float4 v1, float4 v2, float4 v3,
                                        const EwChain& ch, int c, float eps)
{
    float sv[4] = {0.f, 0.f, 0.f, 0.f};          // EW_STORE action 1 / 2: the value saved at a branch point
#pragma unroll 1
    for (int i = i0; i < ch.n; ++i) {
        const EwStep& st = ch.s[i];
        const int s0 = st.ls0, s1 = st.ls1;
        if (st.type == EW_MAXHALF_OUT) return i;
        if (st.type == EW_HOOK) {
            if (s0 == -2) {          // p is not observed: the hook is relu(g) or the identity
                if (st.action == HOOK_RELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] = fmaxf(g[q], 0.f);
                }
                continue;
            }
            const float4 av = s0 >= 0 ? pick_slot(v0, v1, v2, v3, s0) : reinterpret_cast<const float4*>(st.p0)[aidx];
            if (st.action >= HOOK_Q) {      // lean hooks (common.h): a stored quotient / a gate instead of the literal operands; never observed
                const float t[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = fmaxf(g[q], 0.f);
                    g[q] = st.action == HOOK_Q ? z * fabsf(t[q]) : (st.action == HOOK_GATE ? (t[q] > 0.f ? z : 0.f) : ((__float_as_uint(t[q]) >> 31) ? 0.f : z));
                }
                continue;
            }
            const float a[4] = {fmaxf(av.x, 0.f), fmaxf(av.y, 0.f), fmaxf(av.z, 0.f), fmaxf(av.w, 0.f)};
            float p[4], zh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { zh[q] = fmaxf(g[q], 0.f); p[q] = a[q] * zh[q]; }
            int pel = -2;                   // -2: no prior for this row, -1: the dense prior, >= 0: the one non-zero element
            if (PRIOR) {
                if (st.prior_dense) pel = (sb == st.prior_sb) ? -1 : -2;
                else if (st.prior_elem) { const int pe = st.prior_elem[sb]; pel = pe >= 0 ? pe : -2; }
            }
            if (PRIOR && pel != -2) {
                // layerwise EBP: p of this row is overridden by the prior (whitebox.py:390-392)
                float pr[4];
                const float pv = pel >= 0 ? st.prior_val[sb] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) pr[q] = pel == -1 ? st.prior_dense[el0 + q] : ((el0 + q) == pel ? pv : 0.f);
                if (st.pstore) reinterpret_cast<float4*>(st.pstore)[idx] = make_float4(pr[0], pr[1], pr[2], pr[3]);
                if (st.prior_action == PRIOR_DIV) {
                    float x[4] = {a[0], a[1], a[2], a[3]};
                    if (st.p1) {
                        const float4 xv = s1 >= 0 ? pick_slot(v0, v1, v2, v3, s1) : reinterpret_cast<const float4*>(st.p1)[aidx];
                        x[0] = fmaxf(xv.x, 0.f); x[1] = fmaxf(xv.y, 0.f); x[2] = fmaxf(xv.z, 0.f); x[3] = fmaxf(xv.w, 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] = __fdiv_rn(pr[q], x[q] + eps);
                } else if (st.prior_action == PRIOR_GATEZ) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] = pr[q] > 0.f ? g[q] : 0.f;
                }
                if (st.cap_dst) {
                    const int ce = st.cap_elem[sb];
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (el0 + q == ce) st.cap_dst[sb] = pr[q];
                }
                continue;
            }
            if (PRIOR && st.cap_dst) {
                const int ce = st.cap_elem[sb];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (el0 + q == ce) st.cap_dst[sb] = p[q];
            }
            if (st.pstore) reinterpret_cast<float4*>(st.pstore)[idx] = make_float4(p[0], p[1], p[2], p[3]);
            if (st.action == HOOK_DIV) {
                float x[4];
                if (st.p1) {
                    const float4 xv = s1 >= 0 ? pick_slot(v0, v1, v2, v3, s1) : reinterpret_cast<const float4*>(st.p1)[aidx];
                    x[0] = fmaxf(xv.x, 0.f); x[1] = fmaxf(xv.y, 0.f); x[2] = fmaxf(xv.z, 0.f); x[3] = fmaxf(xv.w, 0.f);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = a[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = __fdiv_rn(p[q], x[q] + eps);
            } else if (st.action == HOOK_RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = zh[q];
            }
        } else if (st.type == EW_MASK) {
            const float4 tv = s0 >= 0 ? pick_slot(v0, v1, v2, v3, s0) : reinterpret_cast<const float4*>(st.p0)[aidx];
            if (st.action == 1) {       // the mask is the sign bit of a stored quotient
                g[0] = (__float_as_uint(tv.x) >> 31) ? 0.f : g[0]; g[1] = (__float_as_uint(tv.y) >> 31) ? 0.f : g[1];
                g[2] = (__float_as_uint(tv.z) >> 31) ? 0.f : g[2]; g[3] = (__float_as_uint(tv.w) >> 31) ? 0.f : g[3];
            } else {
                g[0] = tv.x > 0.f ? g[0] : 0.f; g[1] = tv.y > 0.f ? g[1] : 0.f;
                g[2] = tv.z > 0.f ? g[2] : 0.f; g[3] = tv.w > 0.f ? g[3] : 0.f;
            }
        } else if (st.type == EW_SCALE_C) {
            const float sc = st.p0[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] *= sc;
        } else if (st.type == EW_SCALE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] *= st.f;
        } else if (st.type == EW_STORE) {
            if (st.action == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sv[q] = g[q];
            } else {
                reinterpret_cast<float4*>(st.pstore)[idx] = make_float4(g[0], g[1], g[2], g[3]);
                if (st.action == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[q] = sv[q];
                }
            }
        } else if (st.type == EW_ADDP) {
            const float4 d = s0 >= 0 ? pick_slot(v0, v1, v2, v3, s0) : reinterpret_cast<const float4*>(st.p0)[idx];
            g[0] += d.x; g[1] += d.y; g[2] += d.z; g[3] += d.w;
        } else if (st.type == EW_AFFINE_C) {
            const float al = st.p0[c], be = st.p1[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(__fmul_rn(g[q], al), be);
        } else if (st.type == EW_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = fmaxf(g[q], 0.f);
        } else if (st.type == EW_MAXHALF_IN || st.type == EW_MAXPAIR || st.type == EW_POOL2_IN || st.type == EW_AVGUP_IN) {
            // MAXHALF_IN / POOL2_IN were applied where the gradient was loaded; MAXPAIR only exists in compiled epilogues
        } else {
            const float al = st.p0[c], be = st.p1[c];
            float4 v = make_float4(__fadd_rn(__fmul_rn(fmaxf(g[0], 0.f), al), be), __fadd_rn(__fmul_rn(fmaxf(g[1], 0.f), al), be),
                                   __fadd_rn(__fmul_rn(fmaxf(g[2], 0.f), al), be), __fadd_rn(__fmul_rn(fmaxf(g[3], 0.f), al), be));
            if (st.p2) {
                float4 d = reinterpret_cast<const float4*>(st.p2)[idx];
                if (st.action & 1) { d.x = fmaxf(d.x, 0.f); d.y = fmaxf(d.y, 0.f); d.z = fmaxf(d.z, 0.f); d.w = fmaxf(d.w, 0.f); }
                v.x = __fadd_rn(d.x, v.x); v.y = __fadd_rn(d.y, v.y); v.z = __fadd_rn(d.z, v.z); v.w = __fadd_rn(d.w, v.w);
            }
            reinterpret_cast<float4*>(st.pstore)[idx] = v;
        }
    }
    return ch.n;
}

// per_c4 / per_ca4: row strides (in float4 pieces) of the gradient tensors and of the forward-side tensors -- only the fan-out needs them
template <bool PRIOR>
__device__ __forceinline__ void ew_interpret(bool ok, long idx, long aidx, int sb, int el0, float4 gv, float4 od, float4 v0, float4 v1,
                                             float4 v2, float4 v3, float4* __restrict__ dst,
                                             int accumulate, const EwChain& ch, int c, float eps, long per_c4 = 0, long per_ca4 = 0)
{
    if (!ok) return;
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    const int fan = ew_steps<PRIOR>(0, g, idx, aidx, sb, el0, v0, v1, v2, v3, ch, c, eps);
    if (fan < ch.n) {
        // EW_MAXHALF_OUT (the VJP of torch.max(split[0], split[1]) as a fan-out, like the compiled GEMM epilogue's): the chain so far ran over
        // the Co-channel tensor; both halves h route g by the true forward halves, run the REST of the chain as channel c + h * Co of the
        // 2 * Co-channel tensor (operands in place at that channel: ew_plan_loads) and store there.  The launcher excludes accumulate.
        const EwStep& st = ch.s[fan];
        const int Co = st.action;
        const float4* tin = reinterpret_cast<const float4*>(st.p0);
        const float4 ta = tin[aidx], tb = tin[aidx + (long)Co * per_ca4];
        const float av[4] = {ta.x, ta.y, ta.z, ta.w}, bv[4] = {tb.x, tb.y, tb.z, tb.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float gh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gh[q] = ew_maxhalf_route(g[q], h ? bv[q] : av[q], h ? av[q] : bv[q]);
            const long i4 = idx + (long)(h * Co) * per_c4, a4 = aidx + (long)(h * Co) * per_ca4;
            ew_steps<PRIOR>(fan + 1, gh, i4, a4, sb, el0, v0, v1, v2, v3, ch, c + h * Co, eps);
            dst[i4] = make_float4(gh[0], gh[1], gh[2], gh[3]);
        }
        return;
    }
    float4 o = make_float4(g[0], g[1], g[2], g[3]);
    if (accumulate) { o.x += od.x; o.y += od.y; o.z += od.z; o.w += od.w; }
    dst[idx] = o;
}
