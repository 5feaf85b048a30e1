// conv_gemm.hip -- fp32 MFMA implicit-GEMM convolution for gfx950 (MI355X).
//
// Replaces, for the EBP hot path, every mkldnn_convolution / convolution_backward / addmm that the reference's
// hooked forwards and autograd sweep dispatch (whitebox.py:490-498; 600 + 200 conv calls per ResNet-101 map).
//
// GEMM view (per launch):   Out[co][m] = sum_k Wp[k][co] * Im2col[k][m]  (+ bias[co])
//     co : output channel            -> MFMA "A" operand index i (register-mapped rows of the 32x32 D tile)
//     m  : (n, oh, ow) flattened     -> MFMA "B" operand index j (lane-mapped columns of D)  => contiguous in HBM
//     k  : (ci, kh, kw) flattened    -> reduction
// Tensors are CNHW (common.h), so for a fixed k the B row is a shifted, masked, contiguous run of the input and
// the D tile is stored with 128-byte coalesced rows.  Weights are pre-packed K-major ([K][ldw]) at load time, so
// the A tile is read with float4 loads.  A "dual" launch (nhalves == 2) runs W (-> out0 = true activations) and
// relu(W) (-> out1 = positive activations X) as two halves of one grid whose co-tiles of the same m-tile are
// scheduled back-to-back on one XCD, so the activation tile is fetched from HBM once for both.
//
// The same kernel runs the backward-data GEMMs of the MWP sweep: the engine packs relu(W) transposed and
// spatially flipped, which turns conv-backward-data (stride 1) into a forward convolution; 1x1 stride-2
// backward scatters its output grid (out_stride = 2).
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s chip peak).  Block = 256 threads =
// 4 waves in a 2x2 arrangement, each wave owns (TCO/2)x(TM/2) of the block tile as (TCO/64)x(TM/64) MFMA tiles.
// K is consumed in steps of 16 through double-buffered LDS (register-staged prefetch of the next step overlaps
// the 32*MI*NJ/4 MFMAs of the current one); one barrier per step.
#include "common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int NT = 256;

template <int KS>
struct KDecode {
    // k -> (ci, dh, dw)
    __device__ static inline void run(int k, int kh, int kw, int& ci, int& dh, int& dw) {
        if (KS == 1) { ci = k; dh = 0; dw = 0; }
        else if (KS > 1) { ci = k / (KS * KS); int r = k - ci * (KS * KS); dh = r / KS; dw = r - dh * KS; }
        else { int kk = kh * kw; ci = k / kk; int r = k - ci * kk; dh = r / kw; dw = r - dh * kw; }
    }
};

// XCD-aware block -> tile mapping.  The dispatcher places block b on XCD b % 8; remap so that each XCD walks a
// contiguous range of logical tiles, ordered co-fastest: the blocks that share one activation (m) tile run
// back-to-back on the same XCD and hit its private L2.  Bijective for any grid size.
__device__ inline int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

template <int TCO, int TM, int KS, bool VEC>
__global__ __launch_bounds__(NT) void conv_gemm_kernel(const ConvParams p, const int n_co_tiles, const int n_m_tiles)
{
    constexpr int MI = TCO / 64;   // MFMA tiles per wave along co
    constexpr int NJ = TM / 64;    // MFMA tiles per wave along m
    __shared__ float As[2][BK][TCO];
    __shared__ float Bs[2][BK][TM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wrow = wave >> 1, wcol = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // n_co_tiles counts the tiles of BOTH halves of a dual launch (half 1 = relu(W) -> positive activations)
    const int lid = xcd_remap(blockIdx.x, n_co_tiles * n_m_tiles);
    const int tile_m = lid / n_co_tiles;
    const int tile_co_all = lid - tile_m * n_co_tiles;
    const int n_co_half = n_co_tiles / p.nhalves;
    const int half = tile_co_all / n_co_half;
    const int tile_co = tile_co_all - half * n_co_half;
    const int co0 = tile_co * TCO;
    const int m0 = tile_m * TM;
    const float* __restrict__ wsel = half ? p.w_pos : p.w;
    const float* __restrict__ bsel = half ? p.bias_pos : p.bias;
    float* __restrict__ osel = half ? p.out1 : p.out0;

    // ---- A (weights) staging geometry: float4 along co
    constexpr int A_F4_ROW = TCO / 4;
    constexpr int A_ROWS_PASS = NT / A_F4_ROW;
    constexpr int A_PASSES = BK / A_ROWS_PASS;
    const int a_row = tid / A_F4_ROW;
    const int a_c4 = tid - a_row * A_F4_ROW;
    const float* wbase = wsel + co0 + a_c4 * 4;
    float4 areg[A_PASSES];

    // ---- B (im2col) staging geometry
    // generic: one m column per thread, BK*TM/NT rows
    constexpr int B_PER_T = BK * TM / NT;
    constexpr int B_KSTEP = NT / TM;
    const int b_m = tid % TM;
    const int b_k0 = tid / TM;
    // vector (1x1 stride 1, M % 4 == 0): float4 along m
    constexpr int B_F4_ROW = TM / 4;
    constexpr int B_ROWS_PASS = NT / B_F4_ROW;
    constexpr int B_PASSES = BK / B_ROWS_PASS;
    const int bv_row = tid / B_F4_ROW;
    const int bv_c4 = tid - bv_row * B_F4_ROW;

    float breg[VEC ? 1 : B_PER_T];
    float4 bvreg[VEC ? B_PASSES : 1];

    long base_m = 0;
    int ih0 = 0, iw0 = 0;
    bool m_ok = false;
    const long chan_stride = (long)p.NB * p.H * p.W;
    if (!VEC) {
        const int m = m0 + b_m;
        m_ok = m < p.M;
        const int mm = m_ok ? m : 0;
        const int ohw = p.OH * p.OW;
        const int n = mm / ohw;
        const int r = mm - n * ohw;
        const int oh = r / p.OW;
        const int ow = r - oh * p.OW;
        ih0 = oh * p.stride - p.pad;
        iw0 = ow * p.stride - p.pad;
        base_m = (long)n * p.H * p.W + (long)ih0 * p.W + iw0;
    }

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const int k = k0 + a_row + ps * A_ROWS_PASS;
            if (k < p.K) areg[ps] = *reinterpret_cast<const float4*>(wbase + (long)k * p.ldw);
            else areg[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (VEC) {
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps) {
                const int k = k0 + bv_row + ps * B_ROWS_PASS;
                const int m = m0 + bv_c4 * 4;
                if (k < p.K && m < p.M) {
                    float4 v = *reinterpret_cast<const float4*>(p.in + (long)k * p.M + m);
                    if (p.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    bvreg[ps] = v;
                } else bvreg[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PER_T; ++i) {
                const int k = k0 + b_k0 + i * B_KSTEP;
                int ci, dh, dw;
                KDecode<KS>::run(k, p.kh, p.kw, ci, dh, dw);
                const bool ok = m_ok && (k < p.K) && ((unsigned)(ih0 + dh) < (unsigned)p.H) &&
                                ((unsigned)(iw0 + dw) < (unsigned)p.W);
                float v = 0.f;
                if (ok) v = p.in[(long)ci * chan_stride + base_m + dh * p.W + dw];
                if (p.relu_in) v = fmaxf(v, 0.f);
                breg[i] = v;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
            *reinterpret_cast<float4*>(&As[buf][a_row + ps * A_ROWS_PASS][a_c4 * 4]) = areg[ps];
        if (VEC) {
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps)
                *reinterpret_cast<float4*>(&Bs[buf][bv_row + ps * B_ROWS_PASS][bv_c4 * 4]) = bvreg[ps];
        } else {
#pragma unroll
            for (int i = 0; i < B_PER_T; ++i) Bs[buf][b_k0 + i * B_KSTEP][b_m] = breg[i];
        }
    };

    v16f acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int a_off = wrow * (TCO / 2) + l31;
    const int b_off = wcol * (TM / 2) + l31;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = As[buf][kk + lhi][a_off + i * 32];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[buf][kk + lhi][b_off + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[i = (r&3) + 8*(r>>2) + 4*(lane>>5)][j = lane&31]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = m0 + wcol * (TM / 2) + j * 32 + l31;
        if (m >= p.M) continue;
        long col;
        long row_stride;
        if (p.out_stride == 1) {
            col = m;
            row_stride = p.M;
        } else {
            const int ohw = p.OH * p.OW;
            const int n = m / ohw;
            const int r = m - n * ohw;
            const int oh = r / p.OW;
            const int ow = r - oh * p.OW;
            col = ((long)n * p.out_H + (long)oh * p.out_stride) * p.out_W + (long)ow * p.out_stride;
            row_stride = (long)p.NB * p.out_H * p.out_W;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wrow * (TCO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co < p.CoutTot) {
                    float v = acc[i][j][r];
                    if (bsel) v += bsel[co];
                    float* o = osel + (long)co * row_stride;
                    if (p.accumulate) v += o[col];
                    o[col] = v;
                }
            }
        }
    }
}

template <int TCO, int TM>
void launch_cfg(const ConvParams& p, hipStream_t s)
{
    const int n_co = ((p.CoutTot + TCO - 1) / TCO) * p.nhalves;
    const int n_m = (p.M + TM - 1) / TM;
    dim3 grid(n_co * n_m), block(NT);
    const bool vec = (p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && (p.M % 4) == 0 &&
                      p.OH == p.H && p.OW == p.W);
    if (vec) hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, 1, true>), grid, block, 0, s, p, n_co, n_m);
    else if (p.kh == 1 && p.kw == 1) hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, 1, false>), grid, block, 0, s, p, n_co, n_m);
    else if (p.kh == 3 && p.kw == 3) hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, 3, false>), grid, block, 0, s, p, n_co, n_m);
    else hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, 0, false>), grid, block, 0, s, p, n_co, n_m);
}

}  // namespace

void launch_conv_gemm(const ConvParams& p, hipStream_t s)
{
    // Tile choice: biggest tile that still gives every CU >= 2 workgroups (256 CUs); otherwise shrink.
    auto blocks = [&](int tco, int tm) { return (long)((p.CoutTot + tco - 1) / tco) * p.nhalves * ((p.M + tm - 1) / tm); };
    const long want = 512;
    if (p.CoutTot > 64 && blocks(128, 128) >= want) launch_cfg<128, 128>(p, s);
    else if (blocks(64, 128) >= want) launch_cfg<64, 128>(p, s);
    else if (p.CoutTot > 64 && blocks(128, 64) >= want && p.M <= 64 * 1024) launch_cfg<128, 64>(p, s);
    else launch_cfg<64, 64>(p, s);
}
