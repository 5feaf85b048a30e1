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
// maps cannot tell it from a re-ordered fp32 sum (tests/precision/split_probe.py, profiles/r5/experiments/bf16_split.txt).  Six
// v_mfma_f32_32x32x16_bf16 take 6 x 32 cycles where the sixteen fp32 MFMAs of the same K = 16 take 8 x 64.
//   * W: split once per pack (conv_gemm_register_split) into three bf16 planes, tiled [128-row tile][K step][piece][k half][128][8]: a K-step of a row
//     tile is one contiguous 12 KB block that goes global -> LDS without registers (three LDS stages, two steps ahead);
//   * X: fp32 as the producing epilogues left it; the workgroup's 256 lanes gather a 16 x 128 slab three steps ahead into registers (the tap-major
//     gather of K2: per-lane shifted offset per filter tap, channels as the scalar offset, padding / m tails / steps past the end out of the buffer's
//     range = 0), split every value once and write the three planes to LDS in fragment order (ds_write as inline asm: a compiler-visible ds_write after a
//     buffer_load...lds is ordered with s_waitcnt vmcnt(0), which would drain exactly that prefetch);
//   * 128 x 128 block tile, wave (wr, wc) holds the quadrants (wr, wc) of the four 64 x 64 sub-tiles -- so each sub-tile is laid out exactly like the
//     block tile of K1 / K2 and runs their epilogues (block_epilogue) unchanged;
//   * the loop is unrolled by three (stage and register-set indices are static; up to two steps past the end multiply zeros), one counted wait and one
//     raw barrier per step.
// Layers: stride-1 convolutions (1x1 and tap-major KxK) with Cin % 16 == 0, 128 | Cout, no dual-accumulator launch, dense output.
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

template <bool RELU, int CHAIN>
__global__ __launch_bounds__(NT, 2) void conv_gemm_split_kernel(const ConvParams p, const uint16_t* __restrict__ ws0, const uint16_t* __restrict__ ws1,
                                                            const int n_co_tiles, const int n_m_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    constexpr int XBASE = 3 * SP_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    stamp(p, wave, lane, 0, 2);       // tuning stamps / the launch log's span record, like K1

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
    const int nk = p.K / SP_BK;
    const unsigned chan_bytes = (unsigned)p.in_nb * p.H * p.W * 4u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, 0, n_co_half * nk * SP_A_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);

    // staging role: column m0 + (tid & 127), k-half wave >> 1 (wave-uniform)
    const int sm = tid & 127, skh = wave >> 1;
    int base_m = 0;
    unsigned long long tapmask = 0ull;
    {
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
    }
    const unsigned xs_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds) + XBASE + (skh * SP_T + sm) * 16;   // this lane's LDS slot

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_w = [&](int kt, int stage) {
#pragma unroll
        for (int b = 0; b < 3; ++b)      // steps past the end: out of range, the hardware writes zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(lds + stage * SP_A_BYTES + (b * 4 + wave) * 1024), 16,
                                                     ((b * 4 + wave) * 1024 + lane * 16) | (kt < nk ? 0u : OOB), (tile_co * nk + kt) * SP_A_BYTES, 0, 0);
    };
    // the X loads walk the K-steps in order: (tap, first channel) of the next step to be loaded, and that tap's per-lane offset
    int ld_tap = 0, ld_ci0 = 0;
    unsigned ld_voff = (tapmask & 1ull) ? (unsigned)base_m * 4u : OOB;
    auto load_x = [&](int kt, float (&v)[8]) {
        const unsigned voff = ld_voff | (kt < nk ? 0u : OOB);
        const unsigned so = (unsigned)(ld_ci0 + skh * 8) * chan_bytes;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rIn, voff, so + (unsigned)i * chan_bytes, 0));
        ld_ci0 += SP_BK;
        if (ld_ci0 >= p.Cin) {          // wave-uniform: next filter tap => new per-lane shifted offset
            ld_ci0 = 0;
            ld_tap += 1;
            const int dh = ld_tap / p.kw, dw = ld_tap - dh * p.kw;
            ld_voff = (ld_tap < p.kh * p.kw && ((tapmask >> ld_tap) & 1ull)) ? (unsigned)(base_m + dh * p.W + dw) * 4u : OOB;
        }
    };
    auto write_piece = [&](v4u q, int stage, int piece) {
        asm volatile("ds_write_b128 %0, %1" :: "v"(xs_addr + stage * SP_B_BYTES + piece * 2 * SP_T * 16), "v"(q));
    };
    // `flip`: 0x80008000 for a K-step of a NEGATED phase (below), else 0 -- the sign bits of both bf16 of a word
    auto store_x = [&](const float (&v)[8], int stage, unsigned flip) {
        v4u p0, p1, p2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = RELU ? fmaxf(v[2 * i], 0.f) : v[2 * i], b = RELU ? fmaxf(v[2 * i + 1], 0.f) : v[2 * i + 1];
            unsigned q0, q1, q2;
            split_pair(a, b, q0, q1, q2);
            p0[i] = q0 ^ flip; p1[i] = q1 ^ flip; p2[i] = q2 ^ flip;
        }
        write_piece(p0, stage, 0); write_piece(p1, stage, 1); write_piece(p2, stage, 2);
    };
    // Sign phases.  The bf16 MFMA does not round its sum to nearest like the fp32 MFMA (an fmaf chain) does: measured against float64, every output of a
    // bf16x6 GEMM sits ~4e-9 of its sum of magnitudes BELOW the exact value (fp32 kernels: 2e-11, either sign) -- 0.1 of the rms error, but of one sign in
    // every element of every layer, which a contrastive map amplifies.  So the sum changes sign with every pass of the loop below (three K-steps): X pieces
    // are stored negated (one XOR per word), the accumulators negated in place (exact); the hardware's downward error then pushes the true sum UP, and the
    // two cancel.  `flip`: the sign-bit mask of the phase the step's X slab (step kt+1) belongs to.
    // one K-step; ST = kt % 3 (static), xcur = X(kt+1) registers (split here), xnew = registers that receive X(kt+3)
    auto step = [&](int kt, auto ST, float (&xcur)[8], float (&xnew)[8], unsigned flip) {
        constexpr int st = decltype(ST)::value, st1 = (st + 1) % 3, st2 = (st + 2) % 3;
        load_w(kt + 2, st2);
        load_x(kt + 3, xnew);
        const unsigned char* As = lds + st * SP_A_BYTES;
        const unsigned char* Bs = lds + XBASE + st * SP_B_BYTES;
        v8bf af[3][2], bf[3][2];
        // quadrant (wr, wc) of sub-tile (i, j): rows i * 64 + wr * 32, columns j * 64 + wc * 32
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[pc][i] = *(const v8bf*)(As + ((pc * 2 + lhi) * SP_T + i * 64 + wr * 32 + l31) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[pc][j] = *(const v8bf*)(Bs + ((pc * 2 + lhi) * SP_T + j * 64 + wc * 32 + l31) * 16);
        }
        // (piece of W, piece of X), smallest products first
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]][i], bf[TB[t]][j], acc[i][j], 0, 0, 0);
        store_x(xcur, st1, flip);
        // W(kt+1) (issued during step kt-1) has landed when at most X(kt+2), W(kt+2), X(kt+3) are outstanding: 8 + 3 + 8
        asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto negate_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = -acc[i][j][r];
    };
    float x0[8], x1[8], x2[8];
    load_w(0, 0);
    load_x(0, x0);
    store_x(x0, 0, 0u);
    load_w(1, 1);
    load_x(1, x0);
    load_x(2, x1);
    asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // W(0) and the split of X(0) are in LDS
    stamp(p, wave, lane, 1);
    std::integral_constant<int, 0> S0; std::integral_constant<int, 1> S1; std::integral_constant<int, 2> S2;
    unsigned flip = 0u;                         // the current pass's phase
    for (int kt = 0; kt < nk; kt += 3) {        // up to two steps past the end multiply zeros: no branch on kt inside the body
        step(kt, S0, x0, x2, flip);
        step(kt + 1, S1, x1, x0, flip);
        flip ^= 0x80008000u;
        step(kt + 2, S2, x2, x1, flip);         // its X slab is the next pass's first
        negate_acc();
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (flip) negate_acc();                     // an odd number of passes
    stamp(p, wave, lane, 2);
    stamp(p, wave, lane, 3);

    // the four 64 x 64 sub-tiles leave through the epilogues of K1 (each starts with a workgroup barrier before it reuses the LDS).  A REAL loop -- one
    // epilogue instance in the code, not four (the compiled-chain family is 86 signatures) -- over a fixed register tile: the other three shift down.
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        v16f t[1][1];
        t[0][0] = acc[0][0];
        block_epilogue<CHAIN, true>(p, t, smem, tid, lane, wave, co0 + (q >> 1) * 64, m0 + (q & 1) * 64, half, -1, 0, 1, osel, bsel, nullptr);
        acc[0][0] = acc[0][1];
        acc[0][1] = acc[1][0];
        acc[1][0] = acc[1][1];
    }
    stamp(p, wave, lane, 4);
}

// ---- K17 v2 (round 6): eight waves, two per SIMD, in a compute / load ping-pong; short in-pipe sums; stream-K grids -------------------------------
// What round 5 measured on v1: (a) one wave per SIMD (196 workgroups of a 14 x 14 layer on 256 CUs) runs its fragment reads, 24 MFMAs, the split's ~50
// VALU instructions and the LDS stores one after the other -- the matrix pipe idles half the step; (b) the bf16 MFMA does not round its running sum to
// nearest, and the noise of a K-long in-pipe chain shows in contrastive maps once the backward-data GEMMs run on it; (c) a grid of T tiles on C CUs
// takes ceil(T / C) tile times (196 / 256: a quarter of the chip idle; M = 1568: 26 workgroups).  This kernel:
//   * 512 threads = two groups of four waves, one wave of each group on every SIMD.  The K-steps alternate between the groups: while a group runs the 24
//     MFMAs of its step from fragments it already holds in registers, the other group -- on the same SIMDs -- folds its last step's sums into the fp32
//     result, reads the fragments of its next step from LDS, splits the next X slab (the other group's) and issues the loads of later steps.  One
//     s_barrier per step; W: three LDS stages (global -> LDS DMA, issued three steps ahead by the group that will read it), X: two.
//   * every step's six products start from C = 0 and are added (v_pk_add_f32, round-to-nearest) to the result registers during the group's next load
//     phase: no in-pipe chain is longer than 6 MFMAs (96 terms), the K-long sum is an ordinary fp32 sum.  The sign phases of v1 are gone with their cause.
//   * stream-K: a launch whose tile count does not fill the CUs evenly runs one workgroup per CU, each taking an equal share of the (tile, K-step) line;
//     a tile cut by a share boundary is summed by its last part to arrive, in part order (parts park their 128 x 128 partial sums write-through;
//     arrival counters per tile, zero between launches -- the mechanism of K1's tail parts).
// The two K parities of a tile meet through LDS after the loop: group g keeps the 64-row half g of the tile (two 64 x 64 sub-tiles, each laid out like
// K1's block tile) and runs K1's epilogues on it with its own transposition scratch.
constexpr int SP2_NT = 512;
constexpr int SP2_PARK = 64 * 1024;                        // the parity exchange: 8 waves x 2 accumulator tiles x 4 KB (the 60 KB ring lives inside it)
constexpr int SP2_SCRATCH = 4 * 32 * 36 * 4;               // one group's four 32 x 36 transposition tiles
constexpr size_t SP2_LDS = (size_t)SP2_PARK + 2 * SP2_SCRATCH;
constexpr size_t SP2_SLAB_FLOATS = (size_t)SP_T * SP_T;    // one parked part: 128 x 128 fp32

struct StreamK { int grid, tiles; };      // grid == tiles: one whole tile per workgroup, nothing parked

// a workgroup barrier that nothing is scheduled across: an asm's "memory" clobber holds back loads and stores only, and the compiler otherwise moves
// MFMAs and VALU work from one phase of the ping-pong into the other (measured on the first build: the step's first four MFMAs sat in the load phase)
#define SP2_BARRIER(waits)                                              \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        asm volatile(waits "s_barrier" ::: "memory");                   \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)

template <bool RELU, int CHAIN>
__global__ __launch_bounds__(SP2_NT, 2) void conv_gemm_split2_kernel(const ConvParams p, const uint16_t* __restrict__ ws0, const uint16_t* __restrict__ ws1,
                                                                    const int n_co_tiles, const int n_m_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    constexpr int XBASE = 3 * SP_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, gtid = tid & 255;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (grp == 0) stamp(p, wave, lane, 0, 2);

    const int nk = p.K / SP_BK;
    const int n_tiles = n_co_tiles * n_m_tiles;
    const int n_co_half = n_co_tiles / p.nhalves;
    const unsigned chan_bytes = (unsigned)p.in_nb * p.H * p.W * 4u;
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds);
    const int sm = gtid & 127, skh = wave >> 1;                    // staging role inside the group: column m0 + sm, k-half skh (wave-uniform)
    const unsigned xs_addr = lds_base + XBASE + (skh * SP_T + sm) * 16;
    const unsigned char* a_lane = lds + (lhi * SP_T + wr * 32 + l31) * 16;
    const unsigned char* b_lane = lds + XBASE + (lhi * SP_T + wc * 32 + l31) * 16;

    // this workgroup's share of the (tile, K-step) line
    const long total = (long)n_tiles * nk;
    const int G = gridDim.x;
    auto share_begin = [&](int c) -> long { return (total * c) / G; };
    auto owner_of = [&](long x) -> int {             // the workgroup whose share holds step x of the line
        int c = (int)((x * G) / total);
        while (c + 1 < G && share_begin(c + 1) <= x) ++c;
        while (c > 0 && share_begin(c) > x) --c;
        return c;
    };
    const long s_begin = share_begin(blockIdx.x), s_end = share_begin(blockIdx.x + 1);

#pragma unroll 1
    for (long s_at = s_begin; s_at < s_end;) {
        // tiles in XCD order: consecutive shares (workgroups c, c + 8, ... run on one XCD) would put a cut tile's parts on different XCDs either way;
        // the remap keeps the co-tiles of one activation tile on one XCD's L2 when workgroups take whole tiles
        const int lin = __builtin_amdgcn_readfirstlane((int)(s_at / nk));
        const int kb = __builtin_amdgcn_readfirstlane((int)(s_at - (long)lin * nk));
        const int ke = __builtin_amdgcn_readfirstlane((s_end - s_at) < (long)(nk - kb) ? kb + (int)(s_end - s_at) : nk);
        const int lid = G == n_tiles ? xcd_remap(lin, n_tiles) : lin;
        const int tile_m = lid / n_co_tiles;
        const int tile_co_all = lid - tile_m * n_co_tiles;
        const int half = tile_co_all / n_co_half;
        const int tile_co = tile_co_all - half * n_co_half;
        const int co0 = tile_co * SP_T, m0 = tile_m * SP_T;
        const float* __restrict__ bsel = half ? p.bias_pos : p.bias;
        float* __restrict__ osel = half ? p.out1 : p.out0;
        const unsigned long long wbits = (unsigned long long)(half ? ws1 : ws0);         // (a scalar in fact: the compiler must know it, or the DMA gets a waterfall loop)
        const uint16_t* wsel = (const uint16_t*)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wbits)
                                                 | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wbits >> 32)) << 32);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, 0, n_co_half * nk * SP_A_BYTES, 0x00020000);

        int base_m = 0;
        unsigned long long tapmask = 0ull;
        {
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
        }

        v16f res[2][2], pipe[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { res[i][j][r] = 0.f; pipe[i][j][r] = 0.f; }
        v8bf af[3][2], bf[3][2];
        float xr[8];
#ifdef SP2_PROF      /* tuning build (tools/sp2_prof.py): shader cycles a wave spends issuing each phase and waiting at its end */
        unsigned long long pf_t = 0, pf_load = 0, pf_load_w = 0, pf_comp = 0, pf_comp_w = 0;
#define SP2_T(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - pf_t; pf_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SP2_T(acc) do { } while (0)
#endif

        const unsigned flip = grp == 0 ? 0x80008000u : 0u;     // sign phases (see fold)
        auto load_w = [&](int kt, int stage) {         // the group's four waves: 12 KB = 12 x 1 KB; steps past the segment: out of range, zeros
#pragma unroll
            for (int b = 0; b < 3; ++b)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(lds + stage * SP_A_BYTES + (b * 4 + wave) * 1024), 16,
                                                         ((b * 4 + wave) * 1024 + lane * 16) | (kt < ke ? 0u : OOB), (tile_co * nk + kt) * SP_A_BYTES, 0, 0);
        };
        // the X walker of this group: every second K-step (the OTHER group's steps), (tap, first channel) and the tap's per-lane offset
        int ld_tap = 0, ld_ci0 = 0;
        unsigned ld_voff = OOB;
        auto tap_voff = [&](int tap) -> unsigned {
            const int dh = tap / p.kw, dw = tap - dh * p.kw;
            return (tap < p.kh * p.kw && ((tapmask >> tap) & 1ull)) ? (unsigned)(base_m + dh * p.W + dw) * 4u : OOB;
        };
        auto set_walker = [&](int kt) {
            const int k = kt * SP_BK;
            ld_tap = k / p.Cin;
            ld_ci0 = k - ld_tap * p.Cin;
            ld_voff = tap_voff(ld_tap);
        };
        auto load_x = [&](int kt) {
            const unsigned voff = ld_voff | (kt < ke ? 0u : OOB);
            const unsigned so = (unsigned)(ld_ci0 + skh * 8) * chan_bytes;
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rIn, voff, so + (unsigned)i * chan_bytes, 0));
            ld_ci0 += 2 * SP_BK;
            if (ld_ci0 >= p.Cin) {                      // wave-uniform: the walk crossed into the next filter tap (two taps when Cin = 16)
                ld_ci0 -= p.Cin; ld_tap += 1;
                if (ld_ci0 >= p.Cin) { ld_ci0 -= p.Cin; ld_tap += 1; }
                ld_voff = tap_voff(ld_tap);
            }
        };
        auto store_x = [&](int stage) {
            v4u q0, q1, q2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = RELU ? fmaxf(xr[2 * i], 0.f) : xr[2 * i], b = RELU ? fmaxf(xr[2 * i + 1], 0.f) : xr[2 * i + 1];
                unsigned h0, h1, h2;
                split_pair(a, b, h0, h1, h2);
                q0[i] = h0 ^ flip; q1[i] = h1 ^ flip; q2[i] = h2 ^ flip;
            }
            const unsigned at = xs_addr + stage * SP_B_BYTES;
            asm volatile("ds_write_b128 %0, %1" :: "v"(at), "v"(q0));
            asm volatile("ds_write_b128 %0, %1" :: "v"(at + 2 * SP_T * 16), "v"(q1));
            asm volatile("ds_write_b128 %0, %1" :: "v"(at + 4 * SP_T * 16), "v"(q2));
        };
        auto read_frags = [&](int wstage, int xstage) {
            const unsigned char* As = a_lane + wstage * SP_A_BYTES;
            const unsigned char* Bs = b_lane + xstage * SP_B_BYTES;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                for (int i = 0; i < 2; ++i) af[pc][i] = *(const v8bf*)(As + (pc * 2 * SP_T + i * 64) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[pc][j] = *(const v8bf*)(Bs + (pc * 2 * SP_T + j * 64) * 16);
            }
        };
        // Sign phases.  The bf16 MFMA does not round its sum to nearest: against float64 every six-product sum sits ~5e-9 of its sum of magnitudes BELOW
        // the exact value (measured, round 6: rms 6.8e-9 of which 5.0e-9 is that offset, the same in every element of every layer -- what a contrastive
        // map amplifies).  So group 1's steps are summed NEGATED -- group 0 stages their X slabs with the sign bits flipped (one XOR per word), group 1
        // subtracts its sums -- and the offsets of the even and the odd steps cancel.
        auto fold = [&]() {
#ifdef SP2_X_NOFOLD
            return;
#endif
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) res[i][j] += pipe[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) res[i][j] -= pipe[i][j];
            }
        };
        auto compute = [&](auto FRESH) {
            // (piece of W, piece of X), smallest products first; FRESH: the first of the six starts the sum (else the step adds to the last one's)
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
            v16f zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#ifdef SP2_X_NOMFMA
            return;
#endif
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        pipe[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]][i], bf[TB[t]][j], (t == 0 && decltype(FRESH)::value) ? zero : pipe[i][j], 0, 0, 0);
        };
        // load phase of interval I (absolute step numbers; the group owns step I + 1): fold the last step's sums, fragments of step I + 1, split the slab
        // of step I + 2 (loaded two intervals ago: the only loads of this wave still in flight), then W of step I + 3 and the X registers of step I + 4
#ifdef SP2_PROF
        unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pfs[7] = {0, 0, 0, 0, 0, 0, 0};
#define SP2_TS(i) do { __builtin_amdgcn_sched_barrier(0); ts[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SP2_TS(i) do { } while (0)
#endif
        auto load_phase = [&](int I, int w_rd, int w_wr, int x_rd, int x_wr, auto DOFOLD) {
#ifdef SP2_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            SP2_TS(0);
#ifdef SP2_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the X registers (everything this wave still has in flight)
#endif
            SP2_TS(1);
            if constexpr (decltype(DOFOLD)::value) fold();
            SP2_TS(2);
#ifndef SP2_X_NOFRAGS
            read_frags(w_rd, x_rd);
#endif
            SP2_TS(3);
#ifndef SP2_X_NOSTOREX
            store_x(x_wr);                       // (the compiler counts the X loads down itself: vmcnt(6), (4), ... as the pairs are split)
#endif
            SP2_TS(4);
#ifndef SP2_X_NODMA
            load_w(I + 3, w_wr);
#endif
            SP2_TS(5);
#ifndef SP2_X_NOLOADX
            load_x(I + 4);
#endif
            SP2_TS(6);
#ifdef SP2_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };

        // ---- fill: group 0 owns the even steps of the segment (kb, kb + 2, ...), group 1 the odd ones
        if (grp == 0) {
            load_w(kb, 0);
            set_walker(kb + 1);
            load_x(kb + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_x(1);
            load_w(kb + 2, 2);
            load_x(kb + 3);
        } else {
            load_w(kb + 1, 1);
            set_walker(kb);
            load_x(kb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_x(0);
            load_x(kb + 2);
        }
        SP2_BARRIER("s_waitcnt lgkmcnt(0)\n\t");        // W(kb), W(kb + 1), the split slabs of both steps are in LDS
        if (grp == 0) read_frags(0, 0);
        SP2_BARRIER("s_waitcnt lgkmcnt(0)\n\t");        // ... and step kb's fragments out of stage 0 before it is written again
        if (grp == 0 && s_at == s_begin) stamp(p, wave, lane, 1);

        // ---- the loop.  Both groups run the same program -- load phase, barrier, compute, barrier -- one interval apart: group 1 from interval kb on,
        // group 0 after the peeled MFMAs of step kb (and with one fold left over at the end).  Interval I: the group that owns step I + 1 loads.
        const int n_int = (ke - kb + 1) & ~1;             // intervals of the segment (an even number: a last odd step multiplies zeros)
        std::integral_constant<bool, true> yes;
        std::integral_constant<bool, false> no;
        if (grp == 0) {
            compute(yes);                                                       // step kb
            SP2_BARRIER("s_waitcnt vmcnt(8)\n\t");     // W(kb + 2) has landed (at most the X registers of step kb + 3 in flight)
        }
        {
            int I = kb + 1 - grp;
            int w_wr = 1 - grp;                          // (I - kb) % 3
            const int x_wr = 1 - grp;                    // (I - kb) % 2
            const int trips = n_int / 2 - (1 - grp);
#ifdef SP2_PROF
            __builtin_amdgcn_sched_barrier(0);
            pf_t = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
#endif
            auto trip = [&](auto DOFOLD, auto FRESH) {
                const int w_rd = w_wr == 2 ? 0 : w_wr + 1;
                load_phase(I, w_rd, w_wr, 1 - x_wr, x_wr, DOFOLD);
                SP2_T(pf_load);
                SP2_BARRIER("s_waitcnt lgkmcnt(0)\n\t");
                SP2_T(pf_load_w);
#ifdef SP2_PROF
                for (int i = 0; i < 6; ++i) pfs[i] += ts[i + 1] - ts[i];
                pfs[6] += pf_t - ts[6];
#endif
                compute(FRESH);                                                 // step I + 1
                SP2_T(pf_comp);
                SP2_BARRIER("s_waitcnt vmcnt(8)\n\t"); // W(I + 3) has landed
                SP2_T(pf_comp_w);
                I += 2;
                w_wr = w_rd == 2 ? 0 : w_rd + 1;
            };
#ifdef SP2_FOLD2
            // the in-pipe sums run over TWO of the group's steps (12 MFMAs, 192 terms) before they are folded: half the fold's VALU work
            int left = trips;
            if (grp == 1 && left > 0) { trip(no, yes); --left; }       // (group 0 enters with step kb in the pipe registers)
#pragma unroll 1
            for (; left >= 2; left -= 2) { trip(no, no); trip(yes, yes); }
            if (left) trip(no, no);
#else
#pragma unroll 1
            for (int n = 0; n < trips; ++n) trip(yes, yes);
#endif
#ifdef SP2_PROF
            if (p.stamps && lane == 0 && (int)blockIdx.x < p.stamps_cap && s_at == s_begin) {
                unsigned long long* q = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 8 + (size_t)grp * 4 * 8 * p.stamps_cap;      // group 1's records behind group 0's
                q[1] = pf_load; q[2] = pf_load_w; q[3] = pf_comp; q[4] = pf_comp_w; q[5] = (unsigned long long)trips;
                unsigned long long* q2 = q + (size_t)8 * 8 * p.stamps_cap;        // the load phase in detail: behind both groups' records
                for (int i = 0; i < 7; ++i) q2[i] = pfs[i];
            }
#endif
        }
        fold();
        if (grp == 0) SP2_BARRIER("");                   // group 1's last interval
        SP2_BARRIER("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t");      // the ring is free (loads past the segment wrote zeros)
        if (grp == 0 && s_at == s_begin) { stamp(p, wave, lane, 2); stamp(p, wave, lane, 3); }

        // ---- the K parities meet: a wave parks the two tiles of the OTHER group's half, takes its partner's (same quadrant, same lane layout)
        v16f mine[2];
        {
            float4* park = reinterpret_cast<float4*>(lds);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const v16f& give = grp ? res[0][j] : res[1][j];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    park[((((1 - grp) * 4 + wave) * 2 + j) * 4 + r4) * 64 + lane] = make_float4(give[4 * r4], give[4 * r4 + 1], give[4 * r4 + 2], give[4 * r4 + 3]);
            }
            SP2_BARRIER("s_waitcnt lgkmcnt(0)\n\t");
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                mine[j] = grp ? res[1][j] : res[0][j];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 t = park[(((grp * 4 + wave) * 2 + j) * 4 + r4) * 64 + lane];
                    // even steps + odd steps, the same order in both groups
                    if (grp == 0) { mine[j][4 * r4] += t.x; mine[j][4 * r4 + 1] += t.y; mine[j][4 * r4 + 2] += t.z; mine[j][4 * r4 + 3] += t.w; }
                    else { mine[j][4 * r4] = t.x + mine[j][4 * r4]; mine[j][4 * r4 + 1] = t.y + mine[j][4 * r4 + 1]; mine[j][4 * r4 + 2] = t.z + mine[j][4 * r4 + 2]; mine[j][4 * r4 + 3] = t.w + mine[j][4 * r4 + 3]; }
                }
            }
        }

        // ---- a tile cut by a share boundary: park the part write-through, count arrivals, the last part sums all of them in part order
        bool run_epilogue = true;
        if (kb != 0 || ke != nk) {
            const int c_first = owner_of((long)lin * nk), c_last = owner_of((long)lin * nk + nk - 1);
            const int nparts = c_last - c_first + 1;
            // a workgroup parks at most two parts: slot 0 for a part of the first tile of its share, slot 1 for a later one
            float* __restrict__ slab = p.tail_ws + ((size_t)blockIdx.x * 2 + (s_at == s_begin ? 0 : 1)) * SP2_SLAB_FLOATS;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) __hip_atomic_store(slab + (j * 16 + r) * SP2_NT + tid, mine[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_vmcnt<0>();
            __syncthreads();
            int* flag = reinterpret_cast<int*>(lds + SP2_PARK);
            if (tid == 0) {
                const unsigned old = atomicAdd(p.tail_cnt + lin, 1u);
                const int last = (old == (unsigned)(nparts - 1));
                if (last) atomicExch(p.tail_cnt + lin, 0u);     // ready for the next launch on this stream
                *flag = last;
            }
            __syncthreads();
            run_epilogue = *flag != 0;
            if (run_epilogue) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[j][r] = 0.f;
                for (int q = 0; q < nparts; ++q) {
                    const int c = c_first + q;
                    const float* __restrict__ src = p.tail_ws + ((size_t)c * 2 + ((int)(share_begin(c) / nk) == lin ? 0 : 1)) * SP2_SLAB_FLOATS;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            mine[j][r] += __hip_atomic_load(src + (j * 16 + r) * SP2_NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();                            // the flag word lies in group 0's epilogue scratch
        }

        // ---- the group's two 64 x 64 sub-tiles leave through K1's epilogues (each starts with a workgroup barrier: both groups take the same path)
        if (run_epilogue) {
            float* scratch = smem + (SP2_PARK + grp * SP2_SCRATCH) / 4;
#pragma unroll 1
            for (int j = 0; j < 2; ++j) {
                v16f t[1][1];
                t[0][0] = mine[0];
                block_epilogue<CHAIN, true>(p, t, scratch, gtid, lane, wave, co0 + grp * 64, m0 + j * 64, half, -1, 0, 1, osel, bsel, nullptr);
                mine[0] = mine[1];
            }
        }
        s_at += ke - kb;
        if (s_at < s_end) __syncthreads();              // the next segment's fill writes the ring
    }
    if (grp == 0) stamp(p, wave, lane, 4);
}

// W[k][ldw] fp32 (k rows, output channel = column) -> bf16 planes [cout / 128][K / 16][piece][k half][128][8]
__global__ void split_pack_kernel(const float* __restrict__ w, uint16_t* __restrict__ planes, int K, int cout, int ldw)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)K * cout) return;
    const int k = (int)(idx / cout), co = (int)(idx - (long)k * cout);
    unsigned h[3];
    split_pair(w[(long)k * ldw + co], 0.f, h[0], h[1], h[2]);          // the value's pieces in the low halves
    const int rt = co / SP_T, r = co - rt * SP_T, kt = k / SP_BK, kk = k - kt * SP_BK, nk = K / SP_BK;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        planes[((((long)(rt * nk + kt) * 3 + q) * 2 + kk / 8) * SP_T + r) * 8 + kk % 8] = (uint16_t)(h[q] & 0xffffu);
}

// ---- the bf16x6 kernel's host side: a registry of split packs (fp32 pack pointer -> bf16 planes), filled by the engine for the layers it covers
struct SplitPack { uint16_t* planes; int K, cout, ldw; };
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
// ... and the layers the engine sends to it: a property of the LAYER and the engine's setting, never of the batch
bool split_layer_ok(const ConvParams& p)
{
    if (!split_can_run(p)) return false;
    if (p.OH * p.OW < 196) return false;       // 7 x 7 maps (round 5, v1: 68 against 111 TFLOP/s on layer 4)
    return p.K >= (p.kh == 1 && p.kw == 1 ? 1024 : 1152);
}

static std::atomic<long> g_split_launches{0};

// the planes of pack w, split now if this is its first launch: the split runs on the launch's stream, which is drained before the entry becomes visible
// (another stream's launch of the same layer may follow at once)
const uint16_t* split_planes(const float* w, int K, int cout, int ldw, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_split_mu);
    auto it = g_split.find(w);
    if (it != g_split.end()) return (it->second.K == K && it->second.cout == cout && it->second.ldw == ldw) ? it->second.planes : nullptr;
    SplitPack sp{nullptr, K, cout, ldw};
    if (hipMalloc(&sp.planes, (size_t)K * cout * 3 * sizeof(uint16_t)) != hipSuccess) {
        (void)hipGetLastError();
        sp.planes = nullptr;
        g_split[w] = sp;                 // no memory for the planes: this pack stays on the fp32 kernels (until conv_gemm_forget_split) instead of retrying per launch
        return nullptr;
    }
    const long n = (long)K * cout;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, sp.planes, K, cout, ldw);
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(sp.planes); return nullptr; }
    g_split[w] = sp;
    return sp.planes;
}

template <bool RELU, int CHAIN>
void launch_split_inst(const ConvParams& q, const uint16_t* w0, const uint16_t* w1, int grid, int n_co, int n_m, hipStream_t s, int version)
{
    // once per instantiation AND device: the kernels' dynamic LDS (72 / 100 KB) exceeds the default limit
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
#ifndef SP2_ONLY_PLAIN
        (void)hipFuncSetAttribute((const void*)conv_gemm_split_kernel<RELU, CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP_LDS);
#endif
        (void)hipFuncSetAttribute((const void*)conv_gemm_split2_kernel<RELU, CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SP2_LDS);
        done.fetch_or(bit);
    }
#ifndef SP2_ONLY_PLAIN
    if (version == 1) { hipLaunchKernelGGL((conv_gemm_split_kernel<RELU, CHAIN>), dim3(n_co * n_m), dim3(NT), SP_LDS, s, q, w0, w1, n_co, n_m); return; }
#endif
    hipLaunchKernelGGL((conv_gemm_split2_kernel<RELU, CHAIN>), dim3(grid), dim3(SP2_NT), SP2_LDS, s, q, w0, w1, n_co, n_m);
}

// Workgroups of a v2 launch: its tiles when they fill the CUs evenly enough (or nothing can be parked), else one workgroup per CU, each with an equal share
// of the (tile, K-step) line -- at least eight K-steps, so that a share is more than the pipeline's fill.
int pick_stream_k(const ConvParams& p, int tiles, int nk)
{
    static const int force = [] { const char* e = getenv("XFR_STREAM_K"); return e ? atoi(e) : -1; }();        // A/B runs: 0 = whole tiles always
    if (!p.tail_ws || !p.tail_cnt || p.tail_force == 1 || force == 0 || tiles > XFR_TAIL_MAX_TILES) return tiles;
    const int C = conv_gemm_num_cus();
    const long slots = (long)(p.tail_ws_bytes / (2 * SP2_SLAB_FLOATS * sizeof(float)));
    const long total = (long)tiles * nk;
    const int G = (int)std::min<long>(std::min<long>(C, slots), std::max<long>(1, total / 8));
    if (G <= 1 || tiles % G == 0) return tiles;
    // T tiles on C CUs take ceil(T / C) tile times as whole tiles and T / C as shares (plus the parts' exchange, ~5 % of a tile at K = 1024):
    // cut when that saves at least a tenth
    const double whole = (double)((tiles + C - 1) / C), shares = (double)tiles / G + 0.05;
    return shares < 0.9 * whole ? G : tiles;
}

// false: the launch is not one the split kernel covers (or its pack is not registered) -- the caller takes the fp32 kernel the rules give
bool launch_split(const ConvParams& p, hipStream_t s)
{
    static const int version = [] { const char* e = getenv("XFR_SPLIT_KERNEL"); return e ? atoi(e) : 2; }();     // A/B runs: 1 = the round-5 kernel
    if (p.force_cfg == 9 ? !split_can_run(p) : !split_layer_ok(p)) return false;
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
    const uint16_t* w0 = split_planes(p.w, p.K, p.CoutTot, p.ldw, s);
    const uint16_t* w1 = p.nhalves == 2 ? split_planes(p.w_pos, p.K, p.CoutTot, p.ldw, s) : nullptr;
    if (!w0 || (p.nhalves == 2 && !w1)) return false;
    // counted only now: every refusal above sends the launch to an fp32 kernel
    g_split_launches++;
    if (q.chain.n > 0) {
        g_conv_chain_launches[q.chain_sig >= 0 ? 0 : 1]++;
        if (q.chain_sig < 0) conv_gemm_warn_interpreted(q);
    }
    const int n_co = (p.CoutTot / SP_T) * p.nhalves;
    const int n_m = (p.M + SP_T - 1) / SP_T;
    const int grid = version == 1 ? n_co * n_m : pick_stream_k(p, n_co * n_m, p.K / SP_BK);
#ifdef SP2_ONLY_PLAIN
    if (family != 0) return false;
    launch_split_inst<false, 0>(q, w0, w1, grid, n_co, n_m, s, 2);
    return true;
#else
    switch (family) {
        case -1: launch_split_inst<true, 0>(q, w0, w1, grid, n_co, n_m, s, version); break;
        case 0: launch_split_inst<false, 0>(q, w0, w1, grid, n_co, n_m, s, version); break;
        case 1: launch_split_inst<false, 1>(q, w0, w1, grid, n_co, n_m, s, version); break;
        case 2: launch_split_inst<false, 2>(q, w0, w1, grid, n_co, n_m, s, version); break;
        default: launch_split_inst<false, 3>(q, w0, w1, grid, n_co, n_m, s, version); break;
    }
    return true;
#endif
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
bool conv_gemm_split_layer_ok(const ConvParams& p) { return split_layer_ok(p); }
bool conv_gemm_launch_split(const ConvParams& p, hipStream_t s) { return launch_split(p, s); }
long conv_gemm_split_launches() { return g_split_launches.load(); }

