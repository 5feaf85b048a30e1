// conv_gemm_split.hip -- K17: fp32-accurate GEMMs on the bf16 MFMA pipe ("bf16x6") for gfx950, kernel and host side.  Split from conv_gemm.hip in
// round 6 (its own translation unit: the two files compile in parallel).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "conv_gemm_dev.h"
#include "conv_gemm_host.h"

namespace {

// ---- K17: conv_gemm_split_kernel -- fp32-accurate GEMM on the bf16 MFMA pipe ("bf16x6") ---------------------------------------------------------------
// An fp32 value is EXACTLY the sum of three bf16 pieces (8 + 8 + 8 significant bits, each piece the round-to-nearest bf16 of what is left); a bf16 x
// bf16 product is exact in fp32; the six products (i, j) with i + j <= 2, accumulated in fp32 smallest first, reproduce the fp32 product to ~2^-23 -- the
// maps cannot tell it from a re-ordered fp32 sum (tests/precision/split_probe.py, tests/test_split_arith.py).  Six v_mfma_f32_32x32x16_bf16 take
// 6 x 32 cycles where the sixteen fp32 MFMAs of the same K = 16 take 8 x 64.
//   * W: split when the weights arrive (conv_gemm_presplit) into three bf16 planes, tiled [128-row tile][K step][piece][k half][128][8]: a K-step of a row
//     tile is one contiguous 12 KB block that goes global -> LDS without registers (three LDS stages, two steps ahead);
//   * X: fp32 as the producing epilogues left it, split in registers on its way into LDS (ds_write as inline asm: a compiler-visible ds_write after a
//     buffer_load...lds is ordered with s_waitcnt vmcnt(0), which would drain exactly that prefetch), in one of two ways -- below;
//   * 128 x 128 block tile on EIGHT waves, two per SIMD: wave (wr4, wc) owns rows wr4 * 32 .. + 31 and columns wc * 64 .. + 63 -- two 32 x 32
//     accumulator tiles, 12 MFMAs per K-step -- each a quadrant of one of the tile's four 64 x 64 sub-tiles, which are laid out exactly like the block
//     tile of K1 / K2 and leave through their epilogues (block_epilogue) unchanged;
//   * no sum stays in the matrix pipe for more than a pass of three K-steps: the MFMAs of a pass start from C = 0 and the pass is added to the fp32
//     result registers with v_pk_add_f32, every other pass (patch mode: channel block) negated -- see `fold`;
//   * the loop is unrolled by three (static X stage / register-set indices; up to two steps past the end multiply zeros), one counted wait and one raw
//     barrier per step.
// Layers: stride-1 convolutions (1x1 and tap-major KxK) with Cin % 16 == 0, 128 | Cout, no dual-accumulator launch, dense output.
//
// Why eight waves (round 6; profiles/r6/experiments/).  Round 5's kernel ran the tile on four waves, one per SIMD, and that wave issued its DMA (~150
// cycles per instruction in a busy step), waited for its fragment reads and then fed 24 MFMAs, one thing after the other: ~1900 cycles per K-step for 768
// cycles of matrix pipe.  Three ways out were built and measured: (a) K-steps alternating between two four-wave groups in a compute / load ping-pong with
// stream-K grids -- the loading group becomes the bottleneck (2300 cycles per step: DMA and load issue, the split's VALU work under the partner's MFMAs),
// the workgroup owns a whole CU (256 registers x 8 waves) and the timed three-stream step LOST 13 %; (b) the same tile cut among eight waves of ~125
// registers, all in the same instruction order: the deep-K 1x1 layers +21 % in isolation, the 3x3 layers level (the two waves of a SIMD run in
// lock-step between the barriers and wait for LDS together), the step +1 % -- this is the slab mode below; (c) patch mode: the eight waves hold the
// fragments of step kt in registers while they read those of step kt + 1, and the two waves of a SIMD take the two halves of a step in opposite order
// (waves 0-3: MFMAs first; waves 4-7: DMA and fragment reads first): the stage-3 3x3 layer 119 -> 142 TFLOP/s-equivalent in isolation at ~200 registers
// per wave, the timed step level (what one launch gains it takes from the launches that shared its CUs).  The same pipeline in slab mode needs ~250
// registers (two X register sets beside the two fragment sets): 1024 -> 256 at 110 TFLOP/s-equivalent where the lock-step waves hold 119-121, the step
// -3 % (slab_pipelined_step.txt) -- the slab stays in lock-step.
//
// X, mode 1 -- the slab (1x1 and general tap-major layers): per K-step the 512 lanes gather a 16 x 128 slab three steps ahead into registers (the
// tap-major gather of K2: per-lane shifted offset per filter tap, channels as the scalar offset, padding / m tails / steps past the end out of the
// buffer's range = 0), four values per lane.
// X, mode 2 -- the patch ("same" KxK convolutions: 3x3 pad 1, every ResNet bottleneck's middle layer, forward and backward-data).  What limits the slab
// kernel is not the matrix pipe: a 128 x 128 x 16 step moves 20 KB from L2 (12 KB of W planes, 8 KB of fp32 X) and splits 2048 X values; with the MFMAs
// removed the launch is no faster than ~150 TFLOP/s-equivalent (v2_phase_variants.txt).  In such a convolution the X slab of tap (dh, dw) is the slab of
// tap (0, 0) shifted by (dh - pad) * W + (dw - pad) positions -- the same fp32 values, loaded and split kh * kw times.  Here the K order is (channel
// block, tap, channel): per block of 16 channels the workgroup stages ONE patch of the input row -- its 128 positions plus a halo of pad * W + pad on
// both sides -- as bf16 pieces in LDS (one task per lane, spread over the block's first steps), and every tap's B fragments are ds_reads of that patch at
// the tap's offset; positions a tap reaches outside the image read a zero slot.  Per 9 K-steps of a 3x3 layer: 108 + 10 KB from L2 instead of 180, one
// split instead of nine (+14 % in isolation, the ResNet-101 step +3.5 %).  W planes in (channel block, tap) step order (split_pack_kernel, taps > 1).
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SP_T = 128, SP_BK = 16;
constexpr int SP_A_BYTES = 3 * 2 * SP_T * 16, SP_B_BYTES = 3 * 2 * SP_T * 16;
constexpr size_t SP_LDS = 3 * (size_t)(SP_A_BYTES + SP_B_BYTES);

typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)      // v_cvt_pk_bf16_f32: two round-to-nearest-even conversions, lo in the low half
{
    const v2f_t t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, v2bf_t));
}
// Round-to-nearest split of a PAIR (even k in the low half): v = p0 + p1 + p2 exactly, every piece at most half a bf16 ulp of the one before, so the three
// dropped products of order 3 are ~2^-25 of the product and of either sign.  (A truncating split -- one AND per piece -- leaves remainders of up to a
// whole ulp, all of the value's sign: dropped terms ~2^-21 that add up along K instead of averaging out.  Measured in round 5: golden maps 4x further
// from the reference than with the fp32 kernels; with this split they are where the fp32 kernels are.)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);       // exact
    p1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);     // exact, <= 8 significant bits
    p2 = cvt_pk_bf16(sa, sb);
}

// the patch of X mode 2
constexpr int SPP_MAX_HALO = 64, SPP_MAX_TAPS = 25;

__host__ __device__ inline int spp_halo(const ConvParams& p) { return p.pad * p.W + p.pad; }
__host__ __device__ inline int spp_patch_len(const ConvParams& p) { return SP_T + 2 * spp_halo(p); }
inline size_t spp_lds_bytes(const ConvParams& p) { return (size_t)3 * SP_A_BYTES + (size_t)2 * 6 * (spp_patch_len(p) + 1) * 16; }

constexpr int NT8 = 512;

template <int N>
__device__ __forceinline__ void sp8_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <bool RELU, int CHAIN, bool PATCH>
__global__ __launch_bounds__(NT8, 2) void conv_gemm_split_kernel(const ConvParams p, const uint16_t* __restrict__ ws0, const uint16_t* __restrict__ ws1,
                                                                  const int n_co_tiles, const int n_m_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    // W ring: NST stages, DMA issued D = NST - 1 steps ahead (round 6, NST 3 / 4 / 5 on the same box: 1592 / 1550 / 1480 maps/s -- the DMA's latency is not
    // what a step waits for, and a deeper ring costs LDS that co-running workgroups need); X behind it
    constexpr int NST = 3, D = NST - 1, XBASE = NST * SP_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr4 = wave8 >> 1, wc = wave8 & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const bool two_dma = wave8 < 4;               // 12 DMA chunks of 1 KB per K-step on eight waves: waves 0-3 issue two, waves 4-7 one
    if (wave8 < 4) stamp(p, wave8, lane, 0, 4);

    const int lid = xcd_remap(blockIdx.x, n_co_tiles * n_m_tiles);
    const int tile_m = lid / n_co_tiles;
    const int tile_co_all = lid - tile_m * n_co_tiles;
    const int n_co_half = n_co_tiles / p.nhalves;
    const int half = tile_co_all / n_co_half;
    const int tile_co = tile_co_all - half * n_co_half;
    const int co0 = tile_co * SP_T, m0 = tile_m * SP_T;
    const uint16_t* __restrict__ wsel = half ? ws1 : ws0;
    const float* __restrict__ bsel = half ? p.bias_pos : p.bias;
    float* __restrict__ osel = half ? p.out1 : p.out0;
    const int T = p.kh * p.kw, ncb = p.Cin / SP_BK, nk = p.K / SP_BK;
    const int halo = spp_halo(p), PL = spp_patch_len(p);
    const int PS = PATCH ? (PL + 1) * 16 : SP_T * 16;       // bytes of one (piece, k half) plane of X: a patch (+ zero slot), or a step's 128 columns
    const int XB = 6 * PS;
    const unsigned chan_bytes = (unsigned)p.in_nb * p.H * p.W * 4u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, 0, n_co_half * nk * SP_A_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds);

    // ---- X staging role of the lane
    //   PATCH: task t = tid = k half * PL + patch position q (2 PL <= 512)
    //   else : column m0 + (tid & 127), k quarter tid >> 7 (wave-uniform): four of a step's sixteen k
    unsigned t_voff = OOB, t_lds = 0;
    bool t_ok = false;
    int base_m = 0;
    unsigned long long tapmask = 0ull;
    const int sm = tid & 127, skq = wave8 >> 1;
    if constexpr (PATCH) {
        const int kh_ = tid >= PL ? 1 : 0, q = tid - kh_ * PL;
        const long g = (long)m0 - halo + q;
        t_ok = tid < 2 * PL;
        const bool in_row = t_ok && g >= 0 && g < (long)p.in_nb * p.H * p.W;
        t_voff = in_row ? (unsigned)g * 4u + (unsigned)(kh_ * 8) * chan_bytes : OOB;
        t_lds = lds_base + XBASE + kh_ * PS + q * 16;
    } else {
        const int m = m0 + sm;
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
        t_lds = lds_base + XBASE + ((skq >> 1) * SP_T + sm) * 16 + (skq & 1) * 8;
    }
    // ---- fragment columns (PATCH): sub-block j2, column wc * 64 + j2 * 32 + l31; which taps stay inside the image there
    int c_pos[2] = {0, 0};
    unsigned c_mask[2] = {0u, 0u};
    if constexpr (PATCH) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = wc * 64 + j * 32 + l31, m = m0 + c;
            c_pos[j] = c + halo;
            unsigned mk = 0u;
            if (m < p.M) {
                const int ohw = p.OH * p.OW;
                const int r = m % ohw;
                const int oh = r / p.OW, ow = r - oh * p.OW;
                for (int dh = 0; dh < p.kh; ++dh)
                    for (int dw = 0; dw < p.kw; ++dw)
                        if ((unsigned)(oh + dh - p.pad) < (unsigned)p.H && (unsigned)(ow + dw - p.pad) < (unsigned)p.W) mk |= 1u << (dh * p.kw + dw);
            }
            c_mask[j] = mk;
        }
    }
    const unsigned char* a_lane = lds + (lhi * SP_T + wr4 * 32 + l31) * 16;
    const unsigned char* b_plane = lds + XBASE + lhi * PS;

    v16f acc[2], pipe[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; pipe[j][r] = 0.f; }

    auto dma = [&](int kt, int stage, int ch) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(lds + stage * SP_A_BYTES + ch * 1024), 16, (ch * 1024 + lane * 16) | (kt < nk ? 0u : OOB),
                                                 (tile_co * nk + kt) * SP_A_BYTES, 0, 0);
    };
    auto load_w = [&](int kt, int stage) {
        dma(kt, stage, wave8);
        if (two_dma) dma(kt, stage, 8 + wave8);
    };
    // Short in-pipe sums and sign phases (round 6).  The bf16 MFMA does not round its running sum to nearest like the fp32 MFMA (an fmaf chain) does:
    // against float64 a K-long in-pipe sum has 1.5-3x the rms error of the fp32 kernels and sits ~4e-9 of its sum of magnitudes BELOW the exact value, in
    // every element of every layer -- which a contrastive map (a difference of two nearly equal sweeps) amplifies.  So (a) the MFMAs of a pass (three
    // K-steps) start from C = 0 and their sums are added to the fp32 result registers with v_pk_add_f32 (round to nearest): the K-long sum is an ordinary
    // fp32 sum of K / 48 partial sums; (b) every other pass (patch mode: every other channel block) is computed NEGATED -- its X pieces are stored with
    // the sign bits flipped (one XOR per word) and its sums subtracted -- so that the pipe's downward offset changes sign and cancels.  Measured: rms
    // 5-9e-9 of the sum of magnitudes, mean ~1e-11 (fp32 split-K kernel: 1.1e-8; round 5's K-long in-pipe sums: 3.2e-8), profiles/r6/conv_error_probe.txt.
    auto fold = [&](bool negated) {
        if (negated) { acc[0] -= pipe[0]; acc[1] -= pipe[1]; }
        else { acc[0] += pipe[0]; acc[1] += pipe[1]; }
    };
    auto mfmas = [&](const v8bf (&af)[3], const v8bf (&bf)[3][2], auto FIRST) {
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
        v16f zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                pipe[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]], bf[TB[t]][j], (t == 0 && decltype(FIRST)::value) ? zero : pipe[j], 0, 0, 0);
    };
    std::integral_constant<int, 0> S0; std::integral_constant<int, 1> S1; std::integral_constant<int, 2> S2;
    std::integral_constant<bool, true> first; std::integral_constant<bool, false> later;

    if constexpr (PATCH) {
        float xp[8];
        auto load_patch = [&](int cb) {
            const unsigned so = (unsigned)(cb * SP_BK) * chan_bytes;
#pragma unroll
            for (int i = 0; i < 8; ++i) xp[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rIn, t_voff, so + (unsigned)i * chan_bytes, 0));
        };
        auto store_patch = [&](int buf, unsigned flip) {
            v4u q0, q1, q2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = RELU ? fmaxf(xp[2 * i], 0.f) : xp[2 * i], b = RELU ? fmaxf(xp[2 * i + 1], 0.f) : xp[2 * i + 1];
                unsigned h0, h1, h2;
                split_pair(a, b, h0, h1, h2);
                q0[i] = h0 ^ flip; q1[i] = h1 ^ flip; q2[i] = h2 ^ flip;
            }
            if (t_ok) {
                const unsigned at = t_lds + buf * XB;
                asm volatile("ds_write_b128 %0, %1" :: "v"(at), "v"(q0));
                asm volatile("ds_write_b128 %0, %1" :: "v"(at + 2 * PS), "v"(q1));
                asm volatile("ds_write_b128 %0, %1" :: "v"(at + 4 * PS), "v"(q2));
            }
        };
        // Software pipeline, two roles.  A wave holds the fragments of step kt in registers while it reads those of step kt + 1; the two waves that share
        // a SIMD (wave8 and wave8 + 4) run the two halves of a step in opposite order -- waves 0-3: MFMAs, then the load part; waves 4-7: the load part,
        // then MFMAs -- so that on every SIMD one wave's DMA issue and LDS latency sit under the other wave's MFMAs.  (All eight in the same order -- the
        // first cut of this kernel -- run in lock-step between the barriers and wait for LDS together: no faster than four waves.)
        // Position of the step whose fragments are read NEXT (kt + 1): filter tap, channel block, patch buffer, the tap's shift
        int tap = 0, cb = 0, buf = 0, shift = -halo, tdw = 0;
        bool xl_prev = false, xl_now = false;
        const bool mfma_first = wave8 < 4;
        // the load part of step kt: DMA of W(kt + 3), the next block's patch (loaded at tap 0 of the position above, split and stored at tap 2 -- BEFORE
        // the DMA, so that the wait for its X registers leaves at most the previous step's DMA in flight), fragments of step kt + 1
        auto load_part = [&](int kt, v8bf (&af)[3], v8bf (&bf)[3][2]) {
            const int st1 = (kt + 1) % 3, st3 = kt % 3;          // (kt >= -1: the prologue calls this with kt = -1, st3 = 2)
            const bool more = cb + 1 < ncb;
            xl_now = false;
            if (more && tap == 2) store_patch(buf ^ 1, ((cb + 1) & 1) ? 0x80008000u : 0u);
            load_w(kt + 3, kt < 0 ? 2 : st3);
            if (more && tap == 0) { load_patch(cb + 1); xl_now = true; }
            const unsigned char* As = a_lane + (kt < 0 ? 0 : st1) * SP_A_BYTES;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) af[pc] = *(const v8bf*)(As + pc * 2 * SP_T * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pos = ((c_mask[j] >> tap) & 1u) ? c_pos[j] + shift : PL;       // a tap outside the image: the zero slot
                const unsigned char* Bs = b_plane + buf * XB + pos * 16;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) bf[pc][j] = *(const v8bf*)(Bs + pc * 2 * PS);
            }
            tap += 1; tdw += 1; shift += 1;
            if (tdw == p.kw) { tdw = 0; shift += p.W - p.kw; }
            if (tap == T) { tap = 0; shift = -halo; cb += 1; buf ^= 1; }
        };
        // W(kt + 2) -- read during the next step -- has landed when at most this step's DMA and the patch loads of this step or the last one are outstanding
        auto end_step = [&]() {
            if (xl_now || xl_prev) { if (two_dma) sp8_wait_barrier<10>(); else sp8_wait_barrier<9>(); }
            else { if (two_dma) sp8_wait_barrier<2>(); else sp8_wait_barrier<1>(); }
            xl_prev = xl_now;
        };
        auto step = [&](int kt, v8bf (&caf)[3], v8bf (&cbf)[3][2], v8bf (&naf)[3], v8bf (&nbf)[3][2], auto FIRST) {
            // (one copy of the load part, the MFMA block on either side of it: two whole-step branches made the compiler keep three accumulator sets)
            if (mfma_first) mfmas(caf, cbf, FIRST);
            __builtin_amdgcn_sched_barrier(0);
            load_part(kt, naf, nbf);
            __builtin_amdgcn_sched_barrier(0);
            if (!mfma_first) mfmas(caf, cbf, FIRST);
            end_step();
        };
        if (tid < 12) {
            const v4u z = {0u, 0u, 0u, 0u};
            asm volatile("ds_write_b128 %0, %1" :: "v"(lds_base + XBASE + (tid / 6) * XB + (tid % 6) * PS + PL * 16), "v"(z));
        }
        load_w(0, 0);
        load_w(1, 1);
        load_patch(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_patch(0, 0u);
        sp8_wait_barrier<0>();                       // W(0), W(1), the patch of block 0 and the zero slots are in LDS
        v8bf f0a[3], f0b[3][2], f1a[3], f1b[3][2];
        load_part(-1, f0a, f0b);                     // W(2); fragments of step 0
        end_step();
        if (wave8 < 4) stamp(p, wave8, lane, 1);
        int pass_in_blk = 0;
        bool neg = false;                            // sign of the channel block the pass being COMPUTED belongs to
        const int passes_per_blk = T / 3;
        auto pass_done = [&]() {
            fold(neg);
            if (++pass_in_blk == passes_per_blk) { pass_in_blk = 0; neg = !neg; }
        };
        for (int kt = 0; kt < nk; kt += 6) {         // up to five steps past the end multiply zeros (their W stages are zero-filled)
            step(kt, f0a, f0b, f1a, f1b, first);
            step(kt + 1, f1a, f1b, f0a, f0b, later);
            step(kt + 2, f0a, f0b, f1a, f1b, later);
            pass_done();
            step(kt + 3, f1a, f1b, f0a, f0b, first);
            step(kt + 4, f0a, f0b, f1a, f1b, later);
            step(kt + 5, f1a, f1b, f0a, f0b, later);
            pass_done();
        }
    } else {
        int ld_tap = 0, ld_ci0 = 0;
        unsigned ld_voff = (tapmask & 1ull) ? (unsigned)base_m * 4u : OOB;
        auto load_x = [&](int kt, float (&v)[4]) {
            const unsigned voff = ld_voff | (kt < nk ? 0u : OOB);
            const unsigned so = (unsigned)(ld_ci0 + skq * 4) * chan_bytes;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rIn, voff, so + (unsigned)i * chan_bytes, 0));
            ld_ci0 += SP_BK;
            if (ld_ci0 >= p.Cin) {
                ld_ci0 = 0;
                ld_tap += 1;
                const int dh = ld_tap / p.kw, dw = ld_tap - dh * p.kw;
                ld_voff = (ld_tap < p.kh * p.kw && ((tapmask >> ld_tap) & 1ull)) ? (unsigned)(base_m + dh * p.W + dw) * 4u : OOB;
            }
        };
        auto store_x = [&](const float (&v)[4], int stage, unsigned flip) {
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            v2u q0, q1, q2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float a = RELU ? fmaxf(v[2 * i], 0.f) : v[2 * i], b = RELU ? fmaxf(v[2 * i + 1], 0.f) : v[2 * i + 1];
                unsigned h0, h1, h2;
                split_pair(a, b, h0, h1, h2);
                q0[i] = h0 ^ flip; q1[i] = h1 ^ flip; q2[i] = h2 ^ flip;
            }
            const unsigned at = t_lds + stage * XB;
            asm volatile("ds_write_b64 %0, %1" :: "v"(at), "v"(q0));
            asm volatile("ds_write_b64 %0, %1" :: "v"(at + 2 * PS), "v"(q1));
            asm volatile("ds_write_b64 %0, %1" :: "v"(at + 4 * PS), "v"(q2));
        };
        int wst = 0;
        auto step = [&](int kt, auto ST, float (&xcur)[4], float (&xnew)[4], unsigned flip, auto FIRST) {
            constexpr int st = decltype(ST)::value, st1 = (st + 1) % 3;       // X ring: three stages, static (the loop is unrolled by three)
            const int stD = wst + D >= NST ? wst + D - NST : wst + D;
            load_w(kt + D, stD);
            load_x(kt + 3, xnew);            // (AFTER the DMA: the counted wait below relies on this order)
            const unsigned char* As = a_lane + wst * SP_A_BYTES;
            const unsigned char* Bs = b_plane + st * XB + (wc * 64 + l31) * 16;
            v8bf af[3], bf[3][2];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                af[pc] = *(const v8bf*)(As + pc * 2 * SP_T * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[pc][j] = *(const v8bf*)(Bs + pc * 2 * PS + j * 32 * 16);
            }
            mfmas(af, bf, FIRST);
            store_x(xcur, st1, flip);
            // W(kt + 1) has landed when at most X(kt + 2), the DMA of steps kt + 2 .. kt + D, X(kt + 3) are outstanding
            if (two_dma) sp8_wait_barrier<8 + 2 * (D - 1)>(); else sp8_wait_barrier<8 + (D - 1)>();
            wst = wst + 1 == NST ? 0 : wst + 1;
        };
        float x0[4], x1[4], x2[4];
        load_w(0, 0);
        load_x(0, x0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_x(x0, 0, 0u);
#pragma unroll
        for (int d = 1; d < D; ++d) load_w(d, d);
        load_x(1, x0);
        load_x(2, x1);
        if (two_dma) sp8_wait_barrier<8 + 2 * (D - 1)>(); else sp8_wait_barrier<8 + (D - 1)>();
        if (wave8 < 4) stamp(p, wave8, lane, 1);
        unsigned flip = 0u;
        for (int kt = 0; kt < nk; kt += 3) {
            const bool negated = flip != 0u;
            step(kt, S0, x0, x2, flip, first);
            step(kt + 1, S1, x1, x0, flip, later);
            flip ^= 0x80008000u;
            step(kt + 2, S2, x2, x1, flip, later);
            fold(negated);
        }
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wave8 < 4) { stamp(p, wave8, lane, 2); stamp(p, wave8, lane, 3); }

    // ---- epilogues of K1: the wave's two 32 x 32 tiles are quadrants (wr4 & 1, 0) and (wr4 & 1, 1) of the 64 x 64 sub-tile (wr4 >> 1, wc).  block_epilogue
    // indexes its transposition scratch by the quadrant's wave number; the base is shifted so that every one of the eight waves lands on its own 32 x 36 tile.
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
        v16f t[1][1];
        t[0][0] = acc[0];
        const int wq = (wr4 & 1) * 2 + c;
        block_epilogue<CHAIN, true>(p, t, smem + (wave8 - wq) * (32 * 36), tid & 255, lane, wq, co0 + (wr4 >> 1) * 64, m0 + wc * 64, half, -1, 0, 1, osel, bsel, nullptr);
        acc[0] = acc[1];
    }
    if (wave8 < 4) stamp(p, wave8, lane, 4);
}

// W[k][ldw] fp32 (k rows, output channel = column) -> bf16 planes [cout / 128][K step][piece][k half][128][8].  taps == 1: the K steps in the pack's row
// order (X mode 1).  taps > 1 (X mode 2, the patch): the pack's rows are tap-major (k = tap * Cin + ci); step = (ci / 16) * taps + tap -- a channel block's taps side by side.
__global__ void split_pack_kernel(const float* __restrict__ w, uint16_t* __restrict__ planes, int K, int cout, int ldw, int taps)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)K * cout) return;
    const int k = (int)(idx / cout), co = (int)(idx - (long)k * cout);
    unsigned h[3];
    split_pair(w[(long)k * ldw + co], 0.f, h[0], h[1], h[2]);          // the value's pieces in the low halves
    const int cin = K / taps, tap = k / cin, ci = k - tap * cin;
    const int rt = co / SP_T, r = co - rt * SP_T, kt = (ci / SP_BK) * taps + tap, kk = ci % SP_BK, nk = K / SP_BK;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        planes[((((long)(rt * nk + kt) * 3 + q) * 2 + kk / 8) * SP_T + r) * 8 + kk % 8] = (uint16_t)(h[q] & 0xffffu);
}

// ---- the bf16x6 kernel's host side: a registry of split packs (fp32 pack pointer -> bf16 planes), filled by the engine for the layers it covers
struct SplitPack { uint16_t* planes; int K, cout, ldw, taps; };      // taps: 1 = K steps in pack order (slab), kh * kw = channel-block-major (patch)
static std::mutex g_split_mu;
static std::unordered_map<const float*, SplitPack> g_split;

// what the kernel can run at all (xfr_debug_conv with cfg 9 asks for exactly this) ...
bool split_can_run(const ConvParams& p)
{
    if (p.dualacc || p.out_stride != 1 || p.as_strided || p.co_pair > 0 || p.stride != 1) return false;
    if ((p.Cin % SP_BK) != 0 || (p.K % SP_BK) != 0 || (p.CoutTot % SP_T) != 0) return false;
    if (p.kh == 1 && p.kw == 1) return p.pad == 0;
    return p.tap_major == 1 && p.kh * p.kw <= 64;
}
// ... the layers the engine sends to it (their packs get bf16 planes when the weights are loaded) ...
struct SplitPolicy { int min_k1, min_k3, min_tiles; };
const SplitPolicy& split_policy()
{
    // A/B runs only (tools/ab_env.sh): XFR_SPLIT_MIN_K1 / _K3 = the shallowest 1x1 / KxK layer, XFR_SPLIT_MIN_TILES = the smallest grid
    static const SplitPolicy pol = [] {
        SplitPolicy q{256, 1152, 128};     // round 6, same box, alternating: 1x1 from K = 512 +0.5 % (ResNet-101) / +1 % (ResNet-50-128d) over K >= 1024; from
                                           // K = 256 level with that in time (final kernel: 1535 / 2490 against 1532 / 2499 maps/s) and closer to the reference
                                           // (the kernel's sums are more accurate than the fp32 MFMA kernels': row-0 1 - cosine 1.1e-6 against 4.9e-6); from
                                           // K = 128 a loss; KxK from K = 576 and grids from 64 / 196 tiles: level
        if (const char* e = getenv("XFR_SPLIT_MIN_K1")) q.min_k1 = atoi(e);
        if (const char* e = getenv("XFR_SPLIT_MIN_K3")) q.min_k3 = atoi(e);
        if (const char* e = getenv("XFR_SPLIT_MIN_TILES")) q.min_tiles = atoi(e);
        return q;
    }();
    return pol;
}
bool split_layer_ok(const ConvParams& p)
{
    if (!split_can_run(p)) return false;
    if (p.OH * p.OW < 196) return false;       // 7 x 7 maps (round 5: 68 against 111 TFLOP/s on layer 4)
    return p.K >= (p.kh == 1 && p.kw == 1 ? split_policy().min_k1 : split_policy().min_k3);
}
// ... and the launches of such a layer that take it: grids of at least half the CUs.  One wave per SIMD and 128 x 128 tiles: a small grid leaves most of the
// chip idle where the 64 x 64 fp32 tiles still fill it (round 6, M = 1568 / 6272 / 12544 on the stage-3 3x3 layer: 13 / 52 / 108 TFLOP/s-equivalent against
// 49 / 93 / 111 for the fp32 split-K kernel).  A launch's kernel therefore depends on its batch: maps of one image agree across batch sizes to the kernels'
// summation-order difference (~1e-6 of the maximum), not bit for bit -- as with K1's tail balancing.
bool split_grid_ok(const ConvParams& p)
{
    const long tiles = (long)(p.CoutTot / SP_T) * p.nhalves * ((p.M + SP_T - 1) / SP_T);
    return tiles >= split_policy().min_tiles;
}

static std::atomic<long> g_split_launches{0};

// the layers that stage X as a patch: "same" KxK convolutions (output map = input map) with 5 .. 25 taps whose halo fits the patch
bool split_patch_ok(const ConvParams& p)
{
    const int T = p.kh * p.kw;
    if (T < 5 || T > SPP_MAX_TAPS || (T % 3) != 0 || p.stride != 1 || p.tap_major != 1) return false;        // (T % 3: a pass of three K-steps stays inside one channel block)
    if (p.OH != p.H || p.OW != p.W || 2 * p.pad != p.kh - 1 || 2 * p.pad != p.kw - 1) return false;
    static const bool off = getenv("XFR_SPLIT_NO_PATCH") != nullptr;          // A/B runs: the slab for every covered layer
    return spp_halo(p) <= SPP_MAX_HALO && !off;
}

// the planes of pack w, split now if this is its first launch: the split runs on the launch's stream, which is drained before the entry becomes visible
// (another stream's launch of the same layer may follow at once)
const uint16_t* split_planes(const float* w, int K, int cout, int ldw, int taps, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split.find(w);
    if (it != g_split.end())
        return (it->second.K == K && it->second.cout == cout && it->second.ldw == ldw && it->second.taps == taps) ? it->second.planes : nullptr;
    SplitPack sp{nullptr, K, cout, ldw, taps};
    if (hipMalloc(&sp.planes, (size_t)K * cout * 3 * sizeof(uint16_t)) != hipSuccess) {
        (void)hipGetLastError();
        sp.planes = nullptr;
        g_split[w] = sp;                 // no memory for the planes: this pack stays on the fp32 kernels (until conv_gemm_forget_split) instead of retrying per launch
        return nullptr;
    }
    const long n = (long)K * cout;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, sp.planes, K, cout, ldw, taps);
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(sp.planes); return nullptr; }
    g_split[w] = sp;
    return sp.planes;
}

template <bool RELU, int CHAIN>
void launch_split_inst(const ConvParams& q, const uint16_t* w0, const uint16_t* w1, int n_co, int n_m, hipStream_t s, bool patch)
{
    // once per instantiation AND device: the kernels' dynamic LDS (72 KB; patch: 36 KB + two patches) exceeds the default limit
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    constexpr int PATCH_LDS_MAX = 3 * SP_A_BYTES + 2 * 6 * (SP_T + 2 * SPP_MAX_HALO + 1) * 16;
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute((const void*)conv_gemm_split_kernel<RELU, CHAIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_LDS);
        (void)hipFuncSetAttribute((const void*)conv_gemm_split_kernel<RELU, CHAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PATCH_LDS_MAX);
        done.fetch_or(bit);
    }
    const dim3 grid(n_co * n_m);
    if (patch) hipLaunchKernelGGL((conv_gemm_split_kernel<RELU, CHAIN, true>), grid, dim3(NT8), spp_lds_bytes(q), s, q, w0, w1, n_co, n_m);
    else hipLaunchKernelGGL((conv_gemm_split_kernel<RELU, CHAIN, false>), grid, dim3(NT8), SP_LDS, s, q, w0, w1, n_co, n_m);
}

// false: the launch is not one the split kernel covers (or its pack is not registered) -- the caller takes the fp32 kernel the rules give
bool launch_split(const ConvParams& p, hipStream_t s)
{
    if (p.force_cfg == 9 ? !split_can_run(p) : !conv_gemm_split_wanted(p)) return false;
    if (p.chain.n > 0 && p.relu_in) return false;
    ConvParams q = p;
    q.tail_q = 0;
    q.tail_s = 1;
    int family = p.relu_in ? -1 : 0;       // -1: relu on the input, 0: plain epilogue, 1 / 3: compiled chain (3: MaxFeatureMap), 2: interpreted
    if (q.chain.n > 0) {
        if (conv_gemm_plan_chain(q)) return false;
        if (q.chain_sig >= 0 && chain_sig_is_dual(q.chain_sig)) return false;      // (only ever with dualacc, which the layer test excludes)
        family = q.chain_sig < 0 ? 2 : (chain_sig_is_mfm(q.chain_sig) ? 3 : 1);
    }
    const bool patch = split_patch_ok(p);
    const int taps = patch ? p.kh * p.kw : 1;
    const uint16_t* w0 = split_planes(p.w, p.K, p.CoutTot, p.ldw, taps, s);
    const uint16_t* w1 = p.nhalves == 2 ? split_planes(p.w_pos, p.K, p.CoutTot, p.ldw, taps, s) : nullptr;
    if (!w0 || (p.nhalves == 2 && !w1)) return false;
    // counted only now: every refusal above sends the launch to an fp32 kernel
    g_split_launches++;
    if (q.chain.n > 0) {
        g_conv_chain_launches[q.chain_sig >= 0 ? 0 : 1]++;
        if (q.chain_sig < 0) conv_gemm_warn_interpreted(q);
    }
    const int n_co = (p.CoutTot / SP_T) * p.nhalves;
    const int n_m = (p.M + SP_T - 1) / SP_T;
    switch (family) {
        case -1: launch_split_inst<true, 0>(q, w0, w1, n_co, n_m, s, patch); break;
        case 0: launch_split_inst<false, 0>(q, w0, w1, n_co, n_m, s, patch); break;
        case 1: launch_split_inst<false, 1>(q, w0, w1, n_co, n_m, s, patch); break;
        case 2: launch_split_inst<false, 2>(q, w0, w1, n_co, n_m, s, patch); break;
        default: launch_split_inst<false, 3>(q, w0, w1, n_co, n_m, s, patch); break;
    }
    return true;
}

}  // namespace

// bf16x6 split packs (K17): drop the planes of every pack inside [lo, lo + bytes) -- its contents changed, or its memory goes away
void conv_gemm_forget_split(const void* lo, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_split_mu);
    const char* a = static_cast<const char*>(lo);
    for (auto it = g_split.begin(); it != g_split.end();) {
        const char* w = reinterpret_cast<const char*>(it->first);
        if (w >= a && w < a + bytes) {
            if (it->second.planes) (void)hipFree(it->second.planes);     // hipFree waits for the device: no launch still reads them
            it = g_split.erase(it);
        } else ++it;
    }
}
int conv_gemm_split_covers(const ConvParams& p) { return split_layer_ok(p) ? 1 : 0; }
bool conv_gemm_presplit(const ConvParams& p, const float* w, hipStream_t s) { return split_planes(w, p.K, p.CoutTot, p.ldw, split_patch_ok(p) ? p.kh * p.kw : 1, s) != nullptr; }
bool conv_gemm_split_wanted(const ConvParams& p) { return split_layer_ok(p) && (p.split_ok == 2 || split_grid_ok(p)); }
bool conv_gemm_launch_split(const ConvParams& p, hipStream_t s) { return launch_split(p, s); }
long conv_gemm_split_launches() { return g_split_launches.load(); }

