// conv_gemm.hip -- fp32 MFMA implicit-GEMM convolution for gfx950 (MI355X).
//
// Replaces, for the EBP hot path, every mkldnn_convolution / convolution_backward / addmm that the reference's
// hooked forwards and autograd sweep dispatch (whitebox.py:490-498; 600 + 200 conv calls per ResNet-101 map).
//
// GEMM view (per launch):   Out[co][m] = sum_k Wp[k][co] * Im2col[k][m]  (+ bias[co])
//     co : output channel            -> MFMA "A" operand index i (register-mapped rows of the 32x32 D tile)
//     m  : (n, oh, ow) flattened     -> MFMA "B" operand index j (lane-mapped columns of D)  => contiguous in HBM
//     k  : reduction index, (tap, ci) "tap-major" when Cin % 16 == 0, else (ci, kh, kw)
// Tensors are CNHW (common.h), so for a fixed k the B row is a shifted, masked, contiguous run of the input and
// the D tile is stored with 128-byte coalesced rows.  Weights are pre-packed K-major ([K][ldw]) at load time.
// A "dual" launch (nhalves == 2) runs W (-> out0 = true activations) and relu(W) (-> out1 = positive activations
// X) as two halves of one grid whose co-tiles of the same m-tile are scheduled back-to-back on one XCD, so the
// activation tile is fetched from HBM once for both.
//
// The same kernel runs the backward-data GEMMs of the MWP sweep: the engine packs relu(W) transposed and
// spatially flipped, which turns conv-backward-data (stride 1) into a forward convolution; 1x1 stride-2
// backward scatters its output grid (out_stride = 2).
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD, 157 TFLOP/s chip peak).  Block = 256 threads =
// 4 waves (2x2), each wave owns (TCO/2)x(TM/2) of the block tile as (TCO/64)x(TM/64) MFMA tiles.
//
// Feeding: fp32 MFMA needs few bytes per flop, so the enemy is latency, not bandwidth (the layer-3/4 GEMMs of a
// 32-image batch give each CU only 1-3 workgroups).  Operands therefore go global -> LDS directly
// (global_load_lds, no VGPR round trip) through an NST-deep ring of LDS stages: the loads of K-step kt+NST-1 are
// issued before the MFMAs of step kt, and a counted `s_waitcnt vmcnt((NST-2)*L)` + one raw s_barrier per step is
// the only synchronisation -- loads stay in flight across barriers.  Loads are buffer-addressed
// (buffer_load ... lds): per-lane 32-bit offsets are precomputed, the K walk is a scalar soffset, and masked im2col
// elements (padding, K/M tails) are out-of-range offsets for which the hardware writes 0.0 -- the K-loop carries
// no per-element address arithmetic and no branches.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.h"
#include <cstdio>
#include <cstdlib>

#include "conv_gemm_dev.h"
#include "conv_gemm_host.h"

std::atomic<long> g_conv_chain_launches[2];

namespace {

// CHAIN: 0 = plain epilogue, 1 = compiled chain epilogue (p.chain_sig), 2 = interpreted chain epilogue, 3 = compiled, MaxFeatureMap family,
// 4 = DUAL: compiled lean probe-forward epilogue over two accumulator tiles per wave -- W and relu(W), the latter multiplied from the clamped W
// fragment (ConvParams::dualacc: one pack, one staged input tile, two MFMAs per fragment pair)
// Waves per SIMD the MaxFeatureMap epilogue family (CHAIN 3) is compiled for: 6 (<= 80 registers) spills 8 registers of the epilogue into
// 36 bytes of scratch, 5 (<= 96) does not.  Round 4 measured both on Light-CNN (profiles/r4/experiments/mfm_launch_bounds.txt).
#ifndef XFR_MFM_WAVES
#define XFR_MFM_WAVES 6
#endif
#ifndef XFR_DUAL_WAVES
#define XFR_DUAL_WAVES 4          /* 5 (<= 96 registers) spills 4-8 registers of the lean epilogue */
#endif
template <int TCO, int TM, int BK, int NST, int MODE, bool RELU, int CHAIN>
__global__ __launch_bounds__(NT, CHAIN == 0 ? 1 : (CHAIN == 1 ? 7 : (CHAIN == 4 ? XFR_DUAL_WAVES : XFR_MFM_WAVES))) void conv_gemm_kernel(const ConvParams p, const int n_co_tiles, const int n_m_tiles)
{
    // TCO == 32 (with TM == 128): the four waves side by side along m -- a block tile for layers whose channel count leaves a 64-row tile half
    // empty (Light-CNN: 96 = 3 x 32).  Same K-steps, same accumulator order: same bits as the 64 x 64 tile.
    constexpr bool ROW = TCO == 32;
    static_assert(!ROW || TM == 128, "the 32-row tile is 32 x 128");
    constexpr int MI = ROW ? 1 : TCO / 64;   // MFMA tiles per wave along co
    constexpr int NJ = ROW ? 1 : TM / 64;    // MFMA tiles per wave along m
    constexpr int A_FLOATS = BK * TCO, B_FLOATS = BK * TM;
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_PER_WAVE = ROW ? A_FLOATS / 64 / 4 : A_FLOATS / 256 / 4;   // 4-byte (32-row tile) | 16-byte wave-loads per wave per stage
    constexpr int B_PER_WAVE = (MODE == MODE_VEC) ? B_FLOATS / 256 / 4    // 16-byte wave-loads
                                                  : B_FLOATS / 64 / 4;    // 4-byte wave-loads (one k row x 64 m)
    constexpr int L = A_PER_WAVE + B_PER_WAVE;
    constexpr int MSLOTS = TM / 64;                                       // m columns owned by a lane in the dword modes

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = ROW ? 0 : wave >> 1, wcol = ROW ? wave : wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    stamp(p, wave, lane, 0);

    // n_co_tiles counts the tiles of BOTH halves of a dual launch (half 1 = relu(W) -> positive activations)
    const int n_tiles_all = n_co_tiles * n_m_tiles;
    int part, nparts, lid, tail_t = -1;
    if (p.tail_s > 1) {
        // tail balancing: blocks [0, tail_q) are whole tiles; the rest are K-parts of the last tiles
        if ((int)blockIdx.x < p.tail_q) {
            lid = xcd_remap(blockIdx.x, p.tail_q);
            part = 0;
            nparts = 1;
        } else {
            const int tb = blockIdx.x - p.tail_q;
            tail_t = tb / p.tail_s;
            part = tb - tail_t * p.tail_s;
            nparts = p.tail_s;
            lid = p.tail_q + tail_t;
        }
    } else {
        lid = xcd_remap(blockIdx.x, n_tiles_all);
        part = 0;
        nparts = 1;
    }
    int tile_m = lid / n_co_tiles;
    const int tile_co_all = lid - tile_m * n_co_tiles;
    if (p.pair_m > 0 && (p.pair_m % TM) == 0) tile_m = (tile_m >> 1) + (tile_m & 1) * (p.pair_m / TM);      // ConvParams::pair_m
    const int n_co_half = n_co_tiles / p.nhalves;
    const int half = tile_co_all / n_co_half;
    const int tile_co = tile_co_all - half * n_co_half;
    const int co0 = tile_co * TCO;
    const int m0 = tile_m * TM;
    const float* __restrict__ wsel = half ? p.w_pos : p.w;
    const float* __restrict__ bsel = half ? p.bias_pos : p.bias;
    float* __restrict__ osel = half ? p.out1 : p.out0;

    const int nk_all = (p.K + BK - 1) / BK;   // the packed weights are zero-padded to a multiple of 32 rows
    const int kt_lo = (int)((long)nk_all * part / nparts);
    const int kt_hi = (int)((long)nk_all * (part + 1) / nparts);
    const int nk = kt_hi;                     // K-steps are numbered globally; this block runs [kt_lo, kt_hi)
    const unsigned chan_bytes = (unsigned)p.in_nb * p.H * p.W * 4u;
    // weights are packed with K padded to a multiple of BK (zero rows): no K tail on the A side
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, 0, nk_all * BK * p.ldw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);

    // ---- A side: per-lane byte offsets inside a K-step are constant; soffset walks k0
    unsigned voffA[A_PER_WAVE];
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
        const int f = ROW ? (wave * A_PER_WAVE + i) * 64 + lane : (wave * A_PER_WAVE + i) * 256 + lane * 4;
        const int row = f / TCO, col = f - row * TCO;
        voffA[i] = (unsigned)(row * p.ldw + co0 + col) * 4u;
    }
    const unsigned a_step = (unsigned)BK * p.ldw * 4u;

    // ---- B side
    unsigned voffB[(MODE == MODE_VEC) ? B_PER_WAVE : MSLOTS];
    int base_m[MSLOTS], ih0[MSLOTS], iw0[MSLOTS];
    unsigned long long tapmask[MSLOTS];
    if (MODE == MODE_VEC) {
#pragma unroll
        for (int i = 0; i < B_PER_WAVE; ++i) {
            const int f = (wave * B_PER_WAVE + i) * 256 + lane * 4;
            const int row = f / TM, col = f - row * TM;
            const int m = m0 + col;
            // columns beyond M are never stored; keep them inside the row so the read stays in range
            voffB[i] = (m < p.M) ? (unsigned)row * chan_bytes + (unsigned)m * 4u : OOB;
        }
    } else {
#pragma unroll
        for (int s = 0; s < MSLOTS; ++s) {
            const int m = m0 + s * 64 + lane;
            const bool m_ok = m < p.M;
            const int mm = m_ok ? m : 0;
            const int ohw = p.OH * p.OW;
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oh = r / p.OW;
            const int ow = r - oh * p.OW;
            ih0[s] = m_ok ? oh * p.stride - p.pad : -(1 << 28);
            iw0[s] = ow * p.stride - p.pad;
            base_m[s] = n * p.H * p.W + (oh * p.stride - p.pad) * p.W + iw0[s];
            unsigned long long mk = 0ull;
            if ((MODE == MODE_TAP || MODE == MODE_TAP4) && m_ok) {
                // the window is a rectangle: valid columns once (kw tests), then one shifted copy per valid row (kh tests) instead of kh * kw
                // tests -- the 7x7 stems spent a third of a tile's prologue in this loop
                unsigned long long vw = 0ull;
                for (int dw = 0; dw < p.kw; ++dw)
                    if ((unsigned)(iw0[s] + dw) < (unsigned)p.W) vw |= 1ull << dw;
                for (int dh = 0; dh < p.kh; ++dh)
                    if ((unsigned)(ih0[s] + dh) < (unsigned)p.H) mk |= vw << (dh * p.kw);
            }
            tapmask[s] = mk;
            voffB[s] = OOB;
        }
    }
    // generic gather (stems: Cin = 1 or 3): k -> (channel offset, dh, dw) comes from a table in LDS behind the ring
    // instead of two integer divisions per load (the 7x7 stem was VALU-bound on them)
    constexpr int KTAB_MAX = 512;
    int2* ktab = reinterpret_cast<int2*>(smem + NST * STAGE);
    const bool use_ktab = (MODE == MODE_GEN) && (nk_all * BK <= KTAB_MAX);
    if (use_ktab) {
        const int kk = p.kh * p.kw;
        for (int k = tid; k < nk_all * BK; k += NT) {
            int2 e;
            if (k < p.K) {
                const int ci = k / kk;
                const int r = k - ci * kk;
                const int dh = r / p.kw;
                e.x = (int)((unsigned)ci * chan_bytes);
                e.y = (dh << 8) | (r - dh * p.kw);
            } else {
                e.x = 0;
                e.y = -1;
            }
            ktab[k] = e;
        }
        __syncthreads();
    }
    int iss_tap = 0, iss_ci0 = 0;     // (tap, first input channel) of the next K-step to be issued
    if (MODE == MODE_TAP && kt_lo > 0) { iss_tap = (kt_lo * BK) / p.Cin; iss_ci0 = kt_lo * BK - iss_tap * p.Cin; }
    bool tap_dirty = true;

    // ---- issue the loads of one K-step into LDS stage `st`
    // `part` < 0: all loads of the K-step; 0 .. NP-1: the share that is issued after the part-th quarter of the step's
    // MFMAs (the K loop interleaves the loads of a later step with the MFMAs of the current one: a vector-memory
    // instruction issued while the wave's own MFMA executes costs the MFMA pipe nothing, a block of them between the
    // barrier and the first MFMA does)
    constexpr int NP = 4;
    auto issue = [&](int kt, int st, int part) {
        auto mine = [&](int i, int n) { return part < 0 || (i * NP) / n == part; };
        float* As = smem + st * STAGE;
        float* Bs = As + A_FLOATS;
        const int k0 = kt * BK;
        const bool live = kt < nk;      // steps past the end load nothing real (uniform vmcnt accounting)
#pragma unroll
        for (int i = 0; i < A_PER_WAVE; ++i) {
            if (!mine(i, A_PER_WAVE)) continue;
            if constexpr (ROW) bload4(rW, As + (wave * A_PER_WAVE + i) * 64, live ? voffA[i] : OOB, (unsigned)kt * a_step);
            else bload16(rW, As + (wave * A_PER_WAVE + i) * 256, live ? voffA[i] : OOB, (unsigned)kt * a_step);
        }
        if (MODE == MODE_VEC) {
#pragma unroll
            for (int i = 0; i < B_PER_WAVE; ++i)
                if (mine(i, B_PER_WAVE)) bload16(rIn, Bs + (wave * B_PER_WAVE + i) * 256, live ? voffB[i] : OOB, (unsigned)k0 * chan_bytes);
        } else if (MODE == MODE_TAP) {
            // Cin % 16 == 0: one tap per K-step; K-steps are issued in order, so (tap, ci0) advance incrementally
            const int tap = iss_tap, ci0 = iss_ci0;
            if (part <= 0 && tap_dirty) {             // wave-uniform: new tap => new per-lane shifted offsets
                tap_dirty = false;
                const int dh = tap / p.kw, dw = tap - dh * p.kw;
                const int shift = dh * p.W + dw;
#pragma unroll
                for (int s = 0; s < MSLOTS; ++s)
                    voffB[s] = (live && ((tapmask[s] >> tap) & 1ull)) ? (unsigned)(base_m[s] + shift) * 4u : OOB;
            }
#pragma unroll
            for (int i = 0; i < B_PER_WAVE; ++i) {
                const int q = wave * B_PER_WAVE + i;  // (k row, 64-wide m segment)
                const int row = q / MSLOTS, s = q - row * MSLOTS;
                if (mine(i, B_PER_WAVE)) bload4(rIn, Bs + q * 64, voffB[s], (unsigned)(ci0 + row) * chan_bytes);
            }
            if (part < 0 || part == NP - 1) {
                iss_ci0 += BK;
                if (iss_ci0 >= p.Cin) { iss_ci0 = 0; iss_tap += 1; tap_dirty = true; }
            }
        } else if (MODE == MODE_TAP4) {
            // Image stems (Cin <= 4): a K-step is four taps x four channel slots, and the four k rows a wave gathers are the
            // channels of ONE tap (4 * kt + wave): one shifted, masked per-lane offset per wave and K-step, the channel is the
            // scalar offset -- the gather of the tap mode, without a per-load table look-up
            static_assert(MODE != MODE_TAP4 || (MSLOTS == 1 && B_PER_WAVE == 4), "the 4-channel tap mode is written for 64-wide m tiles");
            if (part <= 0) {
                const int tap = kt * 4 + wave;
                const int dh = tap / p.kw, dw = tap - dh * p.kw;
                const bool tap_ok = live && tap < p.kh * p.kw;
                voffB[0] = (tap_ok && ((tapmask[0] >> (tap & 63)) & 1ull)) ? (unsigned)(base_m[0] + dh * p.W + dw) * 4u : OOB;
            }
#pragma unroll
            for (int i = 0; i < B_PER_WAVE; ++i)
                if (mine(i, B_PER_WAVE)) bload4(rIn, Bs + (wave * B_PER_WAVE + i) * 64, i < p.Cin ? voffB[0] : OOB, (unsigned)i * chan_bytes);
        } else {
#pragma unroll
            for (int i = 0; i < B_PER_WAVE; ++i) {
                if (!mine(i, B_PER_WAVE)) continue;
                const int q = wave * B_PER_WAVE + i;
                const int row = q / MSLOTS, s = q - row * MSLOTS;
                const int k = k0 + row;
                int dh, dw;
                unsigned coff;
                bool kok;
                if (use_ktab) {
                    const int2 e = ktab[live ? k : 0];
                    const int ey = __builtin_amdgcn_readfirstlane(e.y);
                    coff = (unsigned)__builtin_amdgcn_readfirstlane(e.x);
                    kok = live && ey >= 0;
                    dh = ey >> 8;
                    dw = ey & 255;
                } else {
                    const int kk = p.kh * p.kw;
                    const int ci = k / kk;
                    const int r = k - ci * kk;
                    dh = r / p.kw;
                    dw = r - dh * p.kw;
                    coff = (unsigned)ci * chan_bytes;
                    kok = k < p.K;
                }
                const bool ok = kok && ((unsigned)(ih0[s] + dh) < (unsigned)p.H) && ((unsigned)(iw0[s] + dw) < (unsigned)p.W);
                bload4(rIn, Bs + q * 64, ok ? (unsigned)(base_m[s] + dh * p.W + dw) * 4u : OOB, coff);
            }
        }
    };

    constexpr bool DUAL = CHAIN == 4;
    static_assert(!DUAL || (MI == 1 && NJ == 1 && !RELU), "the dual-accumulator loop is written for one quadrant per wave");
    v16f acc[MI][NJ], accp[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accp[i][j][r] = 0.f; }

    // prologue: fill NST-1 stages
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(kt_lo + s, s, -1);

    const int a_off = ROW ? l31 : wrow * (TCO / 2) + l31;
    const int b_off = ROW ? wcol * 32 + l31 : wcol * (TM / 2) + l31;

    int st = 0;
    for (int kt = kt_lo; kt < nk; ++kt) {
        // stage kt has landed once at most (NST-2) younger stages are still in flight; the barrier then also
        // guarantees every wave is done reading the stage we are about to refill.
        wait_vmcnt<(NST - 2) * L>();
        __builtin_amdgcn_s_barrier();
        if (kt == kt_lo) stamp(p, wave, lane, 1);
        int st_fill = st + NST - 1;
        if (st_fill >= NST) st_fill -= NST;
        const float* As = smem + st * STAGE;
        const float* Bs = As + A_FLOATS;
        // software-pipelined LDS reads: the fragments of k-pair kk+2 are in flight while the MFMAs of kk run
        float a_cur[MI], b_cur[NJ], a_nxt[MI], b_nxt[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) a_cur[i] = As[lhi * TCO + a_off + i * 32];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b_cur[j] = Bs[lhi * TM + b_off + j * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a_nxt[i] = As[(kk + 2 + lhi) * TCO + a_off + i * 32];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b_nxt[j] = Bs[(kk + 2 + lhi) * TM + b_off + j * 32];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (RELU) b_cur[j] = fmaxf(b_cur[j], 0.f);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
                    if constexpr (DUAL) accp[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fmaxf(a_cur[i], 0.f), b_cur[j], accp[i][j], 0, 0, 0);
                }
            if ((kk / 2) % (BK / 2 / NP) == 0) issue(kt + NST - 1, st_fill, (kk / 2) / (BK / 2 / NP));
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b_cur[j] = b_nxt[j];
            }
        }
        st = (st + 1 == NST) ? 0 : st + 1;
    }
    wait_vmcnt<0>();   // drain the tail loads before the LDS is released
    stamp(p, wave, lane, 2);
    stamp(p, wave, lane, 3);

    block_epilogue<CHAIN, true, ROW>(p, acc, smem, tid, lane, wave, co0, m0, half, tail_t, part, nparts, osel, bsel, DUAL ? &accp[0][0] : nullptr);      // the launcher allocates at least the four 32 x 36 tiles
    stamp(p, wave, lane, 4);
}

// ---- intra-workgroup split-K kernel ------------------------------------------------------------------------------------
// Same 64x64 block tile, same grid, same epilogues as conv_gemm_kernel -- a different K loop.  Above, the four waves split the
// tile 2x2 and share one LDS ring: one accumulator chain per wave (every MFMA waits for the wave's previous one), one
// workgroup barrier per K-step, and a stall of any wave is a stall of all four.  Here every wave owns the WHOLE 64x64 tile over a
// QUARTER of the K range: four independent accumulators (four MFMAs back to back per k-pair), a wave-private ring that the
// wave fills for itself with the same buffer-addressed global -> LDS loads, and therefore no barrier in the loop at all -- a
// wave waits only for its own loads (counted vmcnt).  Each k row is still fetched exactly once per workgroup, so the L2 -> LDS
// traffic is unchanged.  After the loop the four partial tiles are exchanged through the (now free) rings: wave q ends up with
// quadrant q summed over the waves in K order, i.e. exactly the register layout the shared epilogues expect.
// One exchange round covers the quadrants [Q0, Q0 + NQ): wave W parks those it does not own, consecutively, in its own LDS region.
constexpr int ks_slot(int w, int q0, int q)
{   // slot of quadrant q in wave w's region: the quadrants of the round before q that w parks
    int n = 0;
    for (int x = q0; x < q; ++x)
        if (x != w) ++n;
    return n;
}
template <int W, int Q0, int NQ>
__device__ __forceinline__ void ks_park(float* my, const v16f (&acc)[2][2], int lane)
{
#pragma unroll
    for (int q = Q0; q < Q0 + NQ; ++q) {
        if (q == W) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) my[(ks_slot(W, Q0, q) * 16 + r) * 64 + lane] = acc[q >> 1][q & 1][r];
    }
}
template <int W, int Q0>
__device__ __forceinline__ void ks_gather(const float* smem, int wave_lds, const v16f (&acc)[2][2], int lane, v16f& out)
{   // quadrant W summed over the waves in K order (wave 0 holds the lowest K rows)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const float t = (w2 == W) ? acc[W >> 1][W & 1][r] : smem[w2 * wave_lds + (ks_slot(w2, Q0, W) * 16 + r) * 64 + lane];
            sum = (w2 == 0) ? t : sum + t;
        }
        out[r] = sum;
    }
}
// all four waves call this with their own W: the same number of barriers on every path
template <int W, int WAVE_LDS>
__device__ __forceinline__ void ks_exchange(float* smem, const v16f (&acc)[2][2], int lane, v16f& out)
{
    float* my = smem + W * WAVE_LDS;
    if constexpr (WAVE_LDS >= 3 * 16 * 64) {
        ks_park<W, 0, 4>(my, acc, lane);
        __syncthreads();
        ks_gather<W, 0>(smem, WAVE_LDS, acc, lane, out);
    } else {
        // a region holds two quadrants: first the upper half of the tile (owners: waves 0 and 1), then the lower half
        static_assert(WAVE_LDS >= 2 * 16 * 64, "the exchange needs room for two quadrants per wave");
        ks_park<W, 0, 2>(my, acc, lane);
        __syncthreads();
        if constexpr (W < 2) ks_gather<W, 0>(smem, WAVE_LDS, acc, lane, out);
        __syncthreads();
        ks_park<W, 2, 2>(my, acc, lane);
        __syncthreads();
        if constexpr (W >= 2) ks_gather<W, 2>(smem, WAVE_LDS, acc, lane, out);
    }
}

// CHAIN == 4 (DUAL, ConvParams::dualacc): every wave carries EIGHT accumulators -- its 64x64 tile of W and of relu(W), the second multiplied from
// the clamped W fragments: eight independent MFMAs per fragment pair on one private ring.  Two exchanges (W, then relu(W)) leave quadrant q of both
// tiles with wave q, which is what the lean epilogue expects.
#ifndef XFR_KS_DUAL_WAVES
#define XFR_KS_DUAL_WAVES 2
#endif
template <int BK, int NST, int MODE, bool RELU, int CHAIN>
__global__ __launch_bounds__(NT, CHAIN == 4 ? XFR_KS_DUAL_WAVES : 5) void conv_gemm_ks_kernel(const ConvParams p, const int n_co_tiles, const int n_m_tiles)
{
    static_assert(MODE == MODE_VEC || MODE == MODE_TAP, "the split-K kernel covers the 1x1 float4 path and the tap-major gather");
    constexpr int TCO = 64, TM = 64;
    constexpr int A_FLOATS = BK * TCO, B_FLOATS = BK * TM;
    constexpr int STAGE = A_FLOATS + B_FLOATS;                            // per wave
    constexpr int RING = NST * STAGE;
    constexpr int WAVE_LDS = RING > 2 * 16 * 64 ? RING : 2 * 16 * 64;     // the ring also parks partial quadrants: all three, or two per round
    constexpr int A_LOADS = A_FLOATS / 256;                               // 16-byte wave-loads (4 k rows x 64 co each)
    constexpr int B_LOADS = (MODE == MODE_VEC) ? B_FLOATS / 256 : BK;     // 16-byte wave-loads | one 4-byte wave-load per k row
    constexpr int L = A_LOADS + B_LOADS;
    constexpr int NP = BK / 2 - 1;                                        // shares the loads of a K-step are issued in: one after every k-pair but the last
    static_assert((NST - 2) * L <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    stamp(p, wave, lane, 0, 1);

    const int n_tiles_all = n_co_tiles * n_m_tiles;
    int part, nparts, lid, tail_t = -1;
    if (p.tail_s > 1) {
        if ((int)blockIdx.x < p.tail_q) {
            lid = xcd_remap(blockIdx.x, p.tail_q);
            part = 0;
            nparts = 1;
        } else {
            const int tb = blockIdx.x - p.tail_q;
            tail_t = tb / p.tail_s;
            part = tb - tail_t * p.tail_s;
            nparts = p.tail_s;
            lid = p.tail_q + tail_t;
        }
    } else {
        lid = xcd_remap(blockIdx.x, n_tiles_all);
        part = 0;
        nparts = 1;
    }
    int tile_m = lid / n_co_tiles;
    const int tile_co_all = lid - tile_m * n_co_tiles;
    if (p.pair_m > 0 && (p.pair_m % TM) == 0) tile_m = (tile_m >> 1) + (tile_m & 1) * (p.pair_m / TM);      // ConvParams::pair_m
    const int n_co_half = n_co_tiles / p.nhalves;
    const int half = tile_co_all / n_co_half;
    const int tile_co = tile_co_all - half * n_co_half;
    const int co0 = tile_co * TCO;
    const int m0 = tile_m * TM;
    const float* __restrict__ wsel = half ? p.w_pos : p.w;
    const float* __restrict__ bsel = half ? p.bias_pos : p.bias;
    float* __restrict__ osel = half ? p.out1 : p.out0;

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt_lo = (int)((long)nk_all * part / nparts);
    const int kt_hi = (int)((long)nk_all * (part + 1) / nparts);
    // this wave's quarter of the block's K-steps
    const int w_lo = kt_lo + (int)((long)(kt_hi - kt_lo) * wave / 4);
    const int w_hi = kt_lo + (int)((long)(kt_hi - kt_lo) * (wave + 1) / 4);
    const unsigned chan_bytes = (unsigned)p.in_nb * p.H * p.W * 4u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, 0, ((p.K + 31) / 32) * 32 * p.ldw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);

    // ---- A side: load i covers k rows 4i .. 4i+3 of the step, lane -> (row 4i + lane/16, four co at (lane%16)*4)
    unsigned voffA[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) voffA[i] = (unsigned)((4 * i + (lane >> 4)) * p.ldw + co0 + (lane & 15) * 4) * 4u;
    const unsigned a_step = (unsigned)BK * p.ldw * 4u;

    // ---- B side
    unsigned voffB[(MODE == MODE_VEC) ? B_LOADS : 1];
    int base_m = 0;
    unsigned long long tapmask = 0ull;
    if (MODE == MODE_VEC) {
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int m = m0 + (lane & 15) * 4;
            voffB[i] = (m < p.M) ? (unsigned)(4 * i + (lane >> 4)) * chan_bytes + (unsigned)m * 4u : OOB;
        }
    } else {
        const int m = m0 + lane;
        const bool m_ok = m < p.M;
        const int mm = m_ok ? m : 0;
        const int ohw = p.OH * p.OW;
        const int n = mm / ohw;
        const int r = mm - n * ohw;
        const int oh = r / p.OW;
        const int ow = r - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
        base_m = n * p.H * p.W + ih0 * p.W + iw0;
        if (m_ok) {
            unsigned long long vw = 0ull;
            for (int dw = 0; dw < p.kw; ++dw)
                if ((unsigned)(iw0 + dw) < (unsigned)p.W) vw |= 1ull << dw;
            for (int dh = 0; dh < p.kh; ++dh)
                if ((unsigned)(ih0 + dh) < (unsigned)p.H) tapmask |= vw << (dh * p.kw);
        }
        voffB[0] = OOB;
    }
    int iss_tap = 0, iss_ci0 = 0;     // (tap, first input channel) of the next K-step to be issued
    if (MODE == MODE_TAP) { iss_tap = (w_lo * BK) / p.Cin; iss_ci0 = w_lo * BK - iss_tap * p.Cin; }
    bool tap_dirty = true;

    float* ring = smem + wave * WAVE_LDS;
    // `share` < 0: all loads of the K-step; 0 .. NP-1: the share issued with that k-pair's MFMAs
    auto issue = [&](int kt, int st, int share) {
        auto mine = [&](int i, int n) { return share < 0 || (i * NP) / n == share; };
        float* As = ring + st * STAGE;
        float* Bs = As + A_FLOATS;
        const bool live = kt < w_hi;      // steps past the end load nothing real (uniform vmcnt accounting)
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            if (mine(i, A_LOADS)) bload16(rW, As + i * 256, live ? voffA[i] : OOB, (unsigned)kt * a_step);
        if (MODE == MODE_VEC) {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)
                if (mine(i, B_LOADS)) bload16(rIn, Bs + i * 256, live ? voffB[i] : OOB, (unsigned)(kt * BK) * chan_bytes);
        } else {
            const int tap = iss_tap, ci0 = iss_ci0;
            if (share <= 0 && tap_dirty) {            // wave-uniform: new tap => new per-lane shifted offset
                tap_dirty = false;
                const int dh = tap / p.kw, dw = tap - dh * p.kw;
                voffB[0] = (live && ((tapmask >> tap) & 1ull)) ? (unsigned)(base_m + dh * p.W + dw) * 4u : OOB;
            }
#pragma unroll
            for (int row = 0; row < B_LOADS; ++row)
                if (mine(row, B_LOADS)) bload4(rIn, Bs + row * 64, live ? voffB[0] : OOB, (unsigned)(ci0 + row) * chan_bytes);
            if (share < 0 || share == NP - 1) {
                iss_ci0 += BK;
                if (iss_ci0 >= p.Cin) { iss_ci0 = 0; iss_tap += 1; tap_dirty = true; }
            }
        }
    };

    constexpr bool DUAL = CHAIN == 4;
    static_assert(!DUAL || !RELU, "a dual-accumulator launch reads an input that is already clamped");
    v16f acc[2][2], accp[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accp[i][j][r] = 0.f; }

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(w_lo + s, s, -1);

    // The loop is software-pipelined across K-steps: the loads of step kt+NST-1 are issued during the first NP-1 k-pairs of step
    // kt, then -- before the LAST k-pair -- the wave waits for step kt+1 to have landed ((NST-2)*L younger loads may stay in
    // flight) and reads that step's first fragments, so that neither the wait for the loads nor the LDS latency of a step's
    // first read sits in front of an idle MFMA pipe.  The stage refilled during step kt is the one step kt-1 was read from: all
    // its reads fed MFMAs that have been issued.  No barrier anywhere: the ring is this wave's own.
    static_assert(NST >= 3, "the stage being refilled must not be the one read next");
    int st = 0;
    float a_cur[2], b_cur[2], a_nxt[2], b_nxt[2];
    wait_vmcnt<(NST - 2) * L>();
    stamp(p, wave, lane, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) { a_cur[i] = ring[lhi * TCO + l31 + i * 32]; b_cur[i] = ring[A_FLOATS + lhi * TM + l31 + i * 32]; }
    for (int kt = w_lo; kt < w_hi; ++kt) {
        int st_fill = st + NST - 1;
        if (st_fill >= NST) st_fill -= NST;
        const int st_next = (st + 1 == NST) ? 0 : st + 1;
        const float* As = ring + st * STAGE;
        const float* Bs = As + A_FLOATS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { a_nxt[i] = As[(kk + 2 + lhi) * TCO + l31 + i * 32]; b_nxt[i] = Bs[(kk + 2 + lhi) * TM + l31 + i * 32]; }
            } else {
                wait_vmcnt<(NST - 2) * L>();
                const float* An = ring + st_next * STAGE;
#pragma unroll
                for (int i = 0; i < 2; ++i) { a_nxt[i] = An[lhi * TCO + l31 + i * 32]; b_nxt[i] = An[A_FLOATS + lhi * TM + l31 + i * 32]; }
            }
            if (RELU) { b_cur[0] = fmaxf(b_cur[0], 0.f); b_cur[1] = fmaxf(b_cur[1], 0.f); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
            if constexpr (DUAL) {
                const float ap0 = fmaxf(a_cur[0], 0.f), ap1 = fmaxf(a_cur[1], 0.f);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    accp[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap0, b_cur[j], accp[0][j], 0, 0, 0);
                    accp[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap1, b_cur[j], accp[1][j], 0, 0, 0);
                }
            }
            if (kk + 2 < BK) issue(kt + NST - 1, st_fill, kk / 2);
            // order within the k-pair: the next pair's fragment reads first (they then have four MFMAs to land), the global
            // loads between the MFMAs (issued while the pipe is busy)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, (L + NP - 1) / NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, DUAL ? 7 : 3, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) { a_cur[i] = a_nxt[i]; b_cur[i] = b_nxt[i]; }
        }
        st = st_next;
        asm volatile("" ::: "memory");     // the fragment reads of the next step stay in this iteration
    }
    wait_vmcnt<0>();   // the tail loads (nothing real) have landed: the ring is free
    stamp(p, wave, lane, 2);

    // ---- exchange: park the three foreign quadrants in the own ring, meet, gather the own quadrant in K order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    v16f accq[1][1];
    if (wave == 0) ks_exchange<0, WAVE_LDS>(smem, acc, lane, accq[0][0]);
    else if (wave == 1) ks_exchange<1, WAVE_LDS>(smem, acc, lane, accq[0][0]);
    else if (wave == 2) ks_exchange<2, WAVE_LDS>(smem, acc, lane, accq[0][0]);
    else ks_exchange<3, WAVE_LDS>(smem, acc, lane, accq[0][0]);
    v16f accpq;
    if constexpr (DUAL) {
        __syncthreads();          // every wave has gathered its W quadrant: the rings may be overwritten with the relu(W) tiles
        if (wave == 0) ks_exchange<0, WAVE_LDS>(smem, accp, lane, accpq);
        else if (wave == 1) ks_exchange<1, WAVE_LDS>(smem, accp, lane, accpq);
        else if (wave == 2) ks_exchange<2, WAVE_LDS>(smem, accp, lane, accpq);
        else ks_exchange<3, WAVE_LDS>(smem, accp, lane, accpq);
    }
    // the epilogues start with a workgroup barrier before they reuse the LDS (tail parts: before the arrival flag)
    stamp(p, wave, lane, 3);
    block_epilogue<CHAIN, true>(p, accq, smem, tid, lane, wave, co0, m0, half, tail_t, part, nparts, osel, bsel, DUAL ? &accpq : nullptr);
    stamp(p, wave, lane, 4);
}


int num_cus()
{
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
            cu = 256;
        return cu;
    }();
    return n;
}

// Tail balancing.  A grid of T tiles on C CUs, all co-resident (T < ~6 C), finishes when the CUs that got
// ceil(T / C) tiles finish: T = 784 (ResNet-101 layer 3 at 64 images) runs at 3.06 / 4 = 77 % of the rate of T = 768
// (measured: 88 vs 104-114 TFLOP/s).  The last r = T % C tiles are therefore cut along K into S parts, r * S ~ C
// small blocks: every CU receives floor(T / C) whole tiles plus one part.  Returns S (1 = leave the grid alone).
int pick_tail_split(const ConvParams& p, int tiles, int nk, size_t tile_bytes)
{
    constexpr int qmax = 7;
    if (!p.tail_ws || !p.tail_cnt || p.tail_force == 1) return 1;
    const int C = num_cus();
    const int q = tiles / C, r = tiles % C;
    if (r == 0 || r > XFR_TAIL_MAX_TILES) return 1;
    const long max_blocks = (long)(p.tail_ws_bytes / tile_bytes);
    if (p.tail_force >= 2) {
        const int S = std::min(p.tail_force, nk);
        return ((long)r * S <= max_blocks) ? S : 1;
    }
    if (q == 0 && p.K >= 8192 && p.tail_force == 0) {
        // A few tiles with a very deep K -- the backward GEMM through a hooked classifier (K = 80013 / 65359 classes, 4-8 tiles), a Linear on a flattened
        // map: a handful of workgroups would walk the whole weight matrix.  As many parts as fill the chip once, at least 256 K-rows each (round 6:
        // 534 -> 99 us at one image with 64 parts instead of 8, 204 -> 60 us at eight; the exchange costs 0.6 us per part since it stopped
        // serialising its loads).
        const long S = std::min<long>(std::min<long>(64, p.K / 256), std::min<long>(max_blocks / r, std::max<long>(8, C / r)));
        return (int)std::min<long>(S, nk);
    }
    if (q >= qmax || p.K < 1024) return 1;   // many rounds: blocks are dispatched as slots free up, the last round matters little
    // Final-round cost in units of one whole tile: a CU runs ceil(r*S/C) parts of 1/S tile, each with a fixed ramp
    // (ring fill, partial store, arrival) worth ~192 K-rows.  Measured on MI355X: parts shorter
    // than 256 K-rows cost more than they balance, and nothing is gained below K = 1024.  (Round 6, after the last arriver's reduction stopped
    // serialising its loads -- conv_gemm_dev.h block_epilogue: in isolation K = 512 layers under one round now gain 28 % from four parts, but in
    // the timed step and in a batch-1 call every variant of this rule -- K from 256 / 512, parts from 128 rows, half / twice the per-part
    // penalty -- measured level with it: profiles/r6/experiments/tail_exchange.txt.)
    const double ramp = 192.0 / (double)p.K;
    const double whole = 1.0 + ramp;
    double best = whole;
    int bestS = 1;
    const int Smax = std::min(std::min(8, p.K / 256), nk);
    for (int S = 2; S <= Smax; ++S) {
        if ((long)r * S > max_blocks) break;
        const double per_cu = (double)(((long)r * S + C - 1) / C);
        const double cost = per_cu * (1.0 / S + ramp) + 0.01 * S;
        if (cost < best) { best = cost; bestS = S; }
    }
    // the whole grid takes q + 1 rounds unsplit; ask for a 4 % gain on that before adding the exchange
    if (whole - best < 0.04 * (q + 1) * whole) bestS = 1;
    return bestS;
}

// An interpreted epilogue on the product path is a performance cliff (round 5: a one-step chain without a signature ran its GEMMs at 4 TFLOP/s for a
// whole round before a profile showed it): say so once per process, with the signature the generator should have listed.
void warn_interpreted(const ConvParams& q)
{
    static std::atomic<int> said{0};
    // (a launch whose rows are not float4-aligned -- M, or images x pixels, not a multiple of 4: stage 4 of a batch of one -- runs the interpreter BY DESIGN:
    // the compiled epilogues load and store float4; nothing to regenerate, nothing to say)
    const int ohw = q.OH * q.OW;
    const bool vec_ok = (q.M & 3) == 0 && ((q.chain_B * ohw) & 3) == 0 && ((q.out_nb * ohw) & 3) == 0;
    if (q.chain_interpret || !vec_ok || getenv("XFR_QUIET") || said.exchange(1)) return;
    uint16_t codes[XFR_MAX_EW_STEPS];
    const int n = ew_chain_codes(q.chain, codes);
    fprintf(stderr, "xfr_amd: a GEMM launch (Cout %d, K %d, M %d) runs its %d-step fused chain through the INTERPRETED epilogue: no compiled signature [", q.CoutTot, q.K, q.M, q.chain.n);
    for (int i = 0; i < n; ++i) fprintf(stderr, "%s%04x", i ? " " : "", codes[i]);
    fprintf(stderr, "]%s.  Regenerate the table (python tools/gen_chain_sigs.py && make -C xfr_amd/csrc) if this network is a supported backbone; "
                    "xfr_chain_epilogue_stats counts such launches.  (said once; XFR_QUIET=1 silences it)\n",
            n < 0 ? " (the chain carries a trace / prior / capture: interpreted by design)" : (q.accumulate ? " for an accumulating launch" : ""));
}

// Chain of a launch: operand prefetch plan, compiled signature.  Returns 0, or why the launch cannot carry its chain (conv_gemm_refusal).
int plan_chain(ConvParams& q)
{
    EwChain wide = q.chain;                  // compiled epilogues may use the two extra forward slots
    EwLoads wide_ld;
    ew_plan_loads(wide, q.out0, wide_ld, EW_FWD_SLOTS_WIDE);
    ew_plan_loads(q.chain, q.out0, q.chain_ld);
    const int ohw = q.OH * q.OW;
    const bool vec_ok = (q.M & 3) == 0 && ((q.chain_B * ohw) & 3) == 0 && ((q.out_nb * ohw) & 3) == 0;
    q.chain_sig = (q.accumulate || !vec_ok || q.chain_interpret) ? -1 : conv_gemm_chain_sig(wide);
    if (q.chain_sig >= 0) { q.chain = wide; q.chain_ld = wide_ld; }
    if (q.nhalves == 2 && q.chain_sig < 0) return 1;      // a dual launch can only carry a compiled chain: the caller un-fuses
    if ((q.dualacc != 0) != (q.chain_sig >= 0 && chain_sig_is_dual(q.chain_sig))) return 3;   // lean steps <=> two accumulator tiles
    for (int i = 0; i < q.chain.n && q.chain_sig < 0; ++i)
        if (q.chain.s[i].type == EW_MAXPAIR || q.chain.s[i].type == EW_MAXHALF_OUT) return 2;   // steps the interpreter does not have
    return 0;
}

template <int TCO, int TM, int BK, int NST, int MODE>
bool launch_one(const ConvParams& p, hipStream_t s)
{
    const int n_co = ((p.CoutTot + TCO - 1) / TCO) * p.nhalves;
    const int n_m = (p.M + TM - 1) / TM;
    const size_t lds = std::max((size_t)NST * BK * (TCO + TM) * sizeof(float) + (MODE == MODE_GEN ? 512 * sizeof(int2) : 0), (size_t)4 * 32 * 36 * sizeof(float));
    ConvParams q = p;
    int grid = n_co * n_m;
    q.tail_q = 0;
    q.tail_s = 1;
    {
        const int S = pick_tail_split(p, n_co * n_m, (p.K + BK - 1) / BK, (size_t)TCO * TM * sizeof(float) * (p.dualacc ? 2 : 1));
        if (S > 1) {
            const int r = (n_co * n_m) % num_cus();
            q.tail_q = n_co * n_m - r;
            q.tail_s = S;
            grid = q.tail_q + r * S;
        }
    }
    if (p.dualacc && (q.chain.n == 0 || p.nhalves != 1 || p.relu_in)) return false;
    if constexpr ((TCO == 64 && TM == 64) || (TCO == 32 && TM == 128)) {
        if (q.chain.n > 0) {      // fused micro-program (no relu_in)
            if (plan_chain(q)) return false;
            g_conv_chain_launches[q.chain_sig >= 0 ? 0 : 1]++;
            if (q.chain_sig < 0) warn_interpreted(q);
            if constexpr (TCO == 64 && MODE != MODE_TAP4 && MODE != MODE_GEN) {
                if (q.dualacc) {
                    hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, 4>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
                    return true;
                }
            }
            if (q.dualacc) return false;
            if (q.chain_sig >= 0 && chain_sig_is_mfm(q.chain_sig))
                hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, 3>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
            else if (q.chain_sig >= 0)
                hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, 1>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
            else
                hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, 2>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
            return true;
        }
    }
    if (p.relu_in)
        hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, true, 0>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, 0>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
    return true;
}

// the split-K kernel: same grid, tail split and chain selection as launch_one
template <int BK, int NST, int MODE>
bool launch_one_ks(const ConvParams& p, hipStream_t s)
{
    constexpr int TCO = 64, TM = 64;
    const int n_co = ((p.CoutTot + TCO - 1) / TCO) * p.nhalves;
    const int n_m = (p.M + TM - 1) / TM;
    constexpr int RING = NST * BK * (TCO + TM);
    const size_t lds = (size_t)4 * (RING > 2048 ? RING : 2048) * sizeof(float);
    ConvParams q = p;
    int grid = n_co * n_m;
    q.tail_q = 0;
    q.tail_s = 1;
    {
        const int S = pick_tail_split(p, n_co * n_m, (p.K + BK - 1) / BK, (size_t)TCO * TM * sizeof(float) * (p.dualacc ? 2 : 1));
        if (S > 1) {
            const int r = (n_co * n_m) % num_cus();
            q.tail_q = n_co * n_m - r;
            q.tail_s = S;
            grid = q.tail_q + r * S;
        }
    }
    if (p.dualacc && (q.chain.n == 0 || p.nhalves != 1 || p.relu_in)) return false;
    if (q.chain.n > 0) {
        if (plan_chain(q)) return false;
        g_conv_chain_launches[q.chain_sig >= 0 ? 0 : 1]++;
        if (q.chain_sig < 0) warn_interpreted(q);
        if (q.dualacc) {
            hipLaunchKernelGGL((conv_gemm_ks_kernel<BK, NST, MODE, false, 4>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
            return true;
        }
        if (q.chain_sig >= 0 && chain_sig_is_mfm(q.chain_sig))
            hipLaunchKernelGGL((conv_gemm_ks_kernel<BK, NST, MODE, false, 3>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
        else if (q.chain_sig >= 0)
            hipLaunchKernelGGL((conv_gemm_ks_kernel<BK, NST, MODE, false, 1>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
        else
            hipLaunchKernelGGL((conv_gemm_ks_kernel<BK, NST, MODE, false, 2>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
        return true;
    }
    if (p.relu_in)
        hipLaunchKernelGGL((conv_gemm_ks_kernel<BK, NST, MODE, true, 0>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
    else
        hipLaunchKernelGGL((conv_gemm_ks_kernel<BK, NST, MODE, false, 0>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
    return true;
}

// Layers the split-K kernel covers: stride-1 convolutions whose K-steps lie inside one filter tap -- 1x1 (float4 rows when the launch's
// M allows, else the one-tap gather: same K order, same bits) and the tap-major KxK gather.  A property of the LAYER, not of the batch.
template <int BK>
bool ks_ok(const ConvParams& p)
{
    if ((p.Cin % BK) != 0) return false;
    if (p.kh == 1 && p.kw == 1) return p.pad == 0 && p.stride <= 2;      // strided 1x1 (projection / reduce convolutions): the one-tap gather;
                                                                         // their backward-data GEMM scatters in the epilogue (out_stride)
    if (p.out_stride != 1) return false;
    return p.stride == 1 && p.tap_major == 1 && p.kh * p.kw <= 64;
}

template <int BK, int NST>
bool launch_cfg_ks(const ConvParams& p, hipStream_t s)
{
    const bool vec = (p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && (p.M % 4) == 0 && p.OH == p.H && p.OW == p.W);
    if (vec) return launch_one_ks<BK, NST, MODE_VEC>(p, s);
    return launch_one_ks<BK, NST, MODE_TAP>(p, s);
}


template <int TCO, int TM, int BK, int NST>
bool launch_cfg(const ConvParams& p_in, hipStream_t s)
{
    ConvParams p = p_in;
    // a strided 1x1 convolution on the dual-accumulator loop: the tap-major gather with its single tap (for one tap the two K orders are the same
    // rows of the same pack) instead of the generic table gather, which has no dual-accumulator instantiation
    if (p.dualacc && p.kh == 1 && p.kw == 1 && p.tap_major == 0 && (p.Cin % 16) == 0) p.tap_major = 1;
    const bool vec = (p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && (p.M % 4) == 0 && p.OH == p.H && p.OW == p.W);
    if (vec) return launch_one<TCO, TM, BK, NST, MODE_VEC>(p, s);
    if (p.tap_major == 2) return launch_one<64, 64, 16, 4, MODE_TAP4>(p, s);
    if (p.tap_major && (p.Cin % BK) == 0) return launch_one<TCO, TM, BK, NST, MODE_TAP>(p, s);
    if (p.tap_major) return launch_one<TCO, TM, 16, 4, MODE_TAP>(p, s);
    return launch_one<TCO, TM, 16, 4, MODE_GEN>(p, s);
}

}  // namespace

int conv_gemm_num_cus() { return num_cus(); }
int conv_gemm_plan_chain(ConvParams& q) { return plan_chain(q); }
void conv_gemm_warn_interpreted(const ConvParams& q) { warn_interpreted(q); }

// host side of the signature table: exact match of the code sequence
int conv_gemm_chain_sig(const EwChain& ch)
{
    uint16_t codes[XFR_MAX_EW_STEPS];
    const int n = ew_chain_codes(ch, codes);
    if (n <= 0) return -1;
    for (int sgi = 0; sgi < kNumChainSigs; ++sgi) {
        bool same = true;
        for (int i = 0; i < XFR_MAX_EW_STEPS && same; ++i) same = kChainSigs[sgi][i] == (i < n ? codes[i] : 0);
        if (same) return sgi;
    }
    return -1;
}
int conv_gemm_num_chain_sigs() { return kNumChainSigs; }

int conv_gemm_cannot_launch(const ConvParams& p)
{
    if (p.chain.n <= 0) return 0;
    ConvParams q = p;
    return plan_chain(q);
}
const char* conv_gemm_refusal(int why)
{
    switch (why) {
        case 0: return "the convolution launch was refused without a reason";
        case 1: return "a dual (W / relu(W)) convolution launch carries a fused chain without a compiled epilogue";
        case 3: return "lean probe-forward steps and the dual-accumulator launch (ConvParams::dualacc) only exist together, behind a compiled epilogue";
        case 2: return "a fused chain with a MaxFeatureMap step (pair maximum / fan-out VJP) has no compiled epilogue and the interpreter does not run those steps";
        default: return "unknown convolution launch refusal";
    }
}

void conv_gemm_chain_launch_counts(long* compiled, long* interpreted)
{
    if (compiled) *compiled = g_conv_chain_launches[0].load();
    if (interpreted) *interpreted = g_conv_chain_launches[1].load();
}

static int pick_cfg_impl(const ConvParams& p_in, bool allow_split);
int conv_gemm_pick_cfg(const ConvParams& p_in) { return pick_cfg_impl(p_in, true); }
static int pick_cfg_impl(const ConvParams& p_in, bool allow_split)
{
    // A dual-accumulator launch does the work of the dual (W / relu(W)) launch of the same layer with half the workgroups, twice as long each: it
    // takes the kernel that launch takes (the rules below see the dual launch's tile count), never the 32 x 128 tile.
    ConvParams p = p_in;
    if (p.dualacc) { p.nhalves = 2; p.dualacc = 0; const int c = pick_cfg_impl(p, false); return c == 12 ? 4 : c; }
    // Measured on MI355X over the ResNet-101 / ResNet-50 / Light-CNN GEMM shapes (M = 1.5k..400k, K = 64..4608,
    // Cout = 64..2048): the 64x64 tile wins or ties everywhere -- these grids are small (1-12 workgroups per CU), so
    // finer tiles balance the 256 CUs better and keep more waves per SIMD than 128-wide tiles buy in reuse.
    // K >= 512: the intra-workgroup split-K kernel (tools/conv_sweep.py, round 3: +6..13 % on the 3x3 layers and the K = 512 /
    // 2048 1x1 layers, equal at K = 256 / 1024 with N = 1024 / 256, slower below: two K-steps per wave are all prologue).
    // Channel counts that leave the last 64-row tile at most half full (Light-CNN: 96 = 64 + 32): the 32 x 128 block tile wastes no MFMA row.
    // A property of the layer; K order of the 64 x 64 tile, so the bits do not move where that one ran before.
    // bf16x6 (K17): the layers xfr_engine_set_split_gemm covers, when the launch's grid is at least half the CUs (conv_gemm_split.hip)
    if (allow_split && p.split_ok && conv_gemm_split_wanted(p)) return 9;
    {
        const int rem = p.CoutTot % 64;
#ifndef XFR_NO_ROW_TILE      /* A/B builds only (profiles/r4/experiments/row_tile_ab.txt) */
        // (more than one 64-column tile of positions: a one-image call through an 80013-way classifier has ONE column, and 2501 tiles of 32 x 128 took
        // 183 us where the 64 x 64 tile takes 57 -- same K order, same bits)
        if (rem > 0 && rem <= 32 && p.tap_major != 2 && p.out_stride == 1 && p.CoutTot > 32 && p.M > 64) return 12;
#endif
    }
    if (ks_ok<8>(p)) {
        // The choice depends on the LAYER only, never on the batch: the two kernels sum K in different orders, and a sample's map must
        // not depend on how many samples share its launch.  In-engine serial table (tools/cmp_layers.py): deep-K 3x3 (ResNet layers
        // 3 / 4) -4..-12 %, 1x1 with K = 512 or K >= 2048 -3..-14 %, K = 1024 +4 % (stays on K1); strided 1x1 convolutions (forward:
        // gathered input, backward: scattered output) -35 % against the generic-gather kernel.  Split-K for every K >= 512 layer is
        // 0.4 % slower in the timed three-stream schedule (32 KB rings crowd out the other streams' workgroups).
        if (p.kh == 1 && (p.stride == 2 || p.out_stride == 2 || p.as_strided)) return 7;
        if ((p.kh > 1 && p.K >= 2048) || (p.kh == 1 && (p.K == 512 || p.K >= 2048))) return 7;
    }
    // Deep-K launches of at most two tiles per CU (layer 3/4 of a 32-image batch) prefer 32-deep K-steps: half the barriers
    // per MFMA.  Their 48 KB ring allows three workgroups per CU, so larger grids (the W / relu(W) dual launch of the
    // same layer has twice the tiles) stay on the 24 KB ring where whole tiles and tail parts are all resident.
    const long tiles = (long)((p.CoutTot + 63) / 64) * p.nhalves * ((p.M + 63) / 64);
    if (p.K >= 1024 && tiles <= 512 && (p.tap_major ? (p.Cin % 32 == 0) : true)) return 5;
    return 4;
}

// Tuning state (stamps, launch log): process-global, shared by every engine of the process.  g_tuning says whether any of it is on -- the launch
// path takes g_tune_mu only then, so production launches (tuning off) pay one relaxed atomic load; with tuning on, engines launching from several
// host threads serialise on the mutex while they take their record, and set / dump / clear are safe against them.
static std::mutex g_tune_mu;
static thread_local int g_last_cfg = 0;      // configuration of this thread's last launch_conv_gemm (conv_gemm_last_cfg)
static std::atomic<int> g_tuning{0};
static unsigned long long* g_stamps = nullptr;
static int g_stamps_cap = 0;
static int g_stamp_regions = 0, g_stamp_seq = 0;
namespace {
struct LogRec { void* stream; int cout, nhalves, K, M, kh, chain, cfg; };
unsigned long long* g_log = nullptr;
int g_log_cap = 0;
std::vector<LogRec> g_log_recs;
bool g_log_on() { return g_log != nullptr; }
}
// capacity < 0: sampled mode -- |capacity| / 256 regions of 256 records, launch n writes up to 256 evenly spaced workgroups into region n % regions
void conv_gemm_set_stamps(unsigned long long* dev_ptr, int capacity_workgroups)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_stamps = dev_ptr;
    g_stamps_cap = dev_ptr ? (capacity_workgroups < 0 ? -capacity_workgroups : capacity_workgroups) : 0;
    g_stamp_regions = (dev_ptr && capacity_workgroups < 0) ? (-capacity_workgroups) / 256 : 0;
    g_stamp_seq = 0;
    g_tuning.store((g_stamps || g_log_on()) ? 1 : 0);
}

void conv_gemm_set_log(unsigned long long* log_dev, int capacity)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_log = log_dev;
    g_log_cap = log_dev ? capacity : 0;
    g_log_recs.clear();
    g_tuning.store((g_stamps || g_log) ? 1 : 0);
}
int conv_gemm_dump_log(const char* path)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (!g_log) return 0;                       // logging was stopped: the host records are stale, the device buffer may be gone
    const size_t n = g_log_recs.size();
    std::vector<unsigned long long> h(8 * n + 8);
    if (n && hipMemcpy(h.data(), g_log, 8 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "seq,stream,Cout,nhalves,K,M,kh,chain,cfg,start_10ns,end_10ns\n");
    for (size_t i = 0; i < n; ++i) {
        const LogRec& r = g_log_recs[i];
        fprintf(f, "%zu,%p,%d,%d,%d,%d,%d,%d,%d,%llu,%llu\n", i, r.stream, r.cout, r.nhalves, r.K, r.M, r.kh, r.chain, r.cfg, h[8 * i], *std::max_element(h.begin() + 8 * i + 1, h.begin() + 8 * i + 8));
    }
    fclose(f);
    return (int)n;
}

bool launch_conv_gemm(const ConvParams& p_in, hipStream_t s)
{
    ConvParams p = p_in;
    p.stamps = nullptr;
    p.stamps_cap = 0;
    p.stamp_regions = 0;
    p.stamp_seq = 0;
    p.span = nullptr;
    size_t log_idx = (size_t)-1;
    if (g_tuning.load(std::memory_order_relaxed)) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        p.stamps = g_stamps;
        p.stamps_cap = g_stamps_cap;
        p.stamp_regions = g_stamp_regions;
        p.stamp_seq = g_stamps ? g_stamp_seq++ : 0;
        if (g_log && (int)g_log_recs.size() < g_log_cap) {
            p.span = g_log + 8 * g_log_recs.size();
            log_idx = g_log_recs.size();
            g_log_recs.push_back(LogRec{(void*)s, p.CoutTot, p.dualacc ? 2 : p.nhalves, p.K, p.M, p.kh, p.chain.n, -1});      // cfg: the kernel that really runs (below)
        }
    }
    int cfg = p.force_cfg > 0 ? p.force_cfg : conv_gemm_pick_cfg(p);
    {
        // tuning hook (A/B runs of the timed step on one box, tools/ab_env.sh): XFR_CFG_REMAP="4:5,7:6" sends every launch the rules above give
        // configuration 4 to 5 and 7 to 6.  Read once; unset in production.
        struct Remap { int v[32]; };
        static const Remap remap = [] {                        // function-local static: initialised once, thread-safe
            Remap r;
            for (int i = 0; i < 32; ++i) r.v[i] = i;
            if (const char* e = getenv("XFR_CFG_REMAP")) {
                int a = 0, b = 0, n = 0;
                while (sscanf(e, "%d:%d%n", &a, &b, &n) == 2) {
                    if (a >= 0 && a < 32 && b >= 0 && b < 32) r.v[a] = b;
                    e += n;
                    if (*e == ',') ++e;
                }
            }
            return r;
        }();
        if (p.force_cfg <= 0 && cfg >= 0 && cfg < 32) cfg = remap.v[cfg];
    }
    // cfg 6 / 7: the intra-workgroup split-K kernel, (BK, ring stages) = (8, 3): 48 KB of LDS, three workgroups per CU; (4, 4): 32 KB, five.
    // Round 3 sweep (tools/conv_sweep.py): (4, 5) and (4, 6) tie with (4, 4), (16, 3) -- one workgroup per CU -- loses 15 %.
    // the configuration that really runs: the launch log and the engine's per-launch profile record THIS, not the rules' first answer
    auto ran = [&](int eff, bool ok) {
        g_last_cfg = eff;
        if (log_idx != (size_t)-1) {
            std::lock_guard<std::mutex> lk(g_tune_mu);
            if (log_idx < g_log_recs.size()) g_log_recs[log_idx].cfg = eff;
        }
        return ok;
    };
    if (cfg == 9) {
        if (conv_gemm_launch_split(p, s)) return ran(9, true);
        cfg = pick_cfg_impl(p, false);     // refused (a chain family without a split instantiation, no memory for the planes): the fp32 rules
    }
    if (cfg == 6 && ks_ok<8>(p)) return ran(6, launch_cfg_ks<8, 3>(p, s));
    if (cfg == 7 && ks_ok<4>(p)) return ran(7, launch_cfg_ks<4, 4>(p, s));
    if (cfg == 5) return ran(5, launch_cfg<64, 64, 32, 3>(p, s));
    if (cfg == 12) return ran(12, launch_cfg<32, 128, 16, 3>(p, s));          // the 32 x 128 block tile (four waves side by side along m)
    return ran(4, launch_cfg<64, 64, 16, 3>(p, s));
}

int conv_gemm_last_cfg() { return g_last_cfg; }
