// conv_ws.hip -- persistent, wave-specialised fp32-MFMA GEMM for the 1x1 stride-1 convolutions (forward and backward-data) of the EBP hot
// path: the short- and mid-K layers whose launches pay more around their K loops than inside them (whitebox.py:388-428 hook chains behind
// the backward GEMMs, resnet.py:111-149 BatchNorm / add / ReLU behind the forward ones).
//
// Why another kernel.  conv_gemm_kernel gives every 64x64 tile its own workgroup: operands of the first K-steps (4-5 us until they
// land), K loop, epilogue (4-9 us of HBM round trips for a hook chain), exit.  A CU holds six such workgroups and they move in step:
// during prologue and epilogue the MFMA pipes idle.  For K = 256 that is 11 us around a 17 us K loop, twice per launch (DESIGN.md
// section 6: 28 % of the GEMM time of a ResNet-101 step runs at 77 TFLOP/s).
//
// Here a workgroup is PERSISTENT (it walks a list of tiles) and its eight waves have two jobs:
//   * waves 0-3 ("math"): the 2x2 wave grid of conv_gemm_kernel over a 64x64 tile, the same LDS ring filled by buffer_load ... lds --
//     but the ring never drains between tiles: the operands of the next tile's first K-steps are issued during the last K-steps of
//     the current one.  After a tile's last K-step a wave parks its 32x32 accumulators in a hand-off tile in LDS (16 ds_write) and
//     starts the next tile.  No global loads into registers, no stores, no epilogue: nothing but MFMAs, fragment reads and ring loads.
//   * waves 4-7 ("epilogue"): take the hand-off tile of tile i while the math waves run tile i+1 and do everything else -- bias, the
//     compiled hook chain with its operand loads (prefetched one piece ahead), BatchNorm / residual / ReLU, MaxFeatureMap, the stores.
//     Their HBM latencies meet no MFMA.
// The only synchronisation is the per-K-step workgroup barrier the ring needs anyway: the epilogue waves take part in it (s_barrier
// counts every wave of the workgroup) and slice their work between barriers -- piece q of the previous tile after barrier q * nk / 4.
// The K order of every output element is conv_gemm_kernel's (one accumulator, K-steps in order): bit-identical results.
//
// Scope: MODE_VEC of conv_gemm.hip (1x1, stride 1, pad 0, M % 4 == 0), dense float4 output rows, no chain / compiled chain; dual
// (W / relu(W)) launches; `accumulate` without a chain.  Everything else stays on conv_gemm_kernel / conv_gemm_ks_kernel.
#include <algorithm>
#include "conv_epilogue.h"

namespace {

constexpr int WS_NT = 512;          // 4 math + 4 epilogue waves
constexpr int WS_LD = 36;           // row pitch of a hand-off tile (floats): 16-byte aligned rows, no bank clash (conv_gemm.hip dense_epilogue)

// tiles of one XCD: the same contiguous, co-fastest range xcd_remap gives it, walked by the XCD's workgroups in interleaved order
struct WsSeq {
    int base, cnt;      // this XCD's tiles [base, base + cnt)
    int loc, nwg;       // this workgroup's ordinal among the XCD's workgroups, and their number
};
__device__ inline WsSeq ws_seq(int n_tiles)
{
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = n_tiles >> 3, r = n_tiles & 7;
    WsSeq s;
    s.base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    s.cnt = q + (xcd < r ? 1 : 0);
    s.loc = bid >> 3;
    s.nwg = (G - xcd + 7) >> 3;
    return s;
}

struct WsTile { int co0, m0, half; };
__device__ inline WsTile ws_tile(const ConvParams& p, int lid, int n_co_tiles)
{
    const int tile_m = lid / n_co_tiles;
    const int tile_co_all = lid - tile_m * n_co_tiles;
    const int n_co_half = n_co_tiles / p.nhalves;
    WsTile t;
    t.half = tile_co_all / n_co_half;
    t.co0 = (tile_co_all - t.half * n_co_half) * 64;
    t.m0 = tile_m * 64;
    return t;
}

// ---- math waves ---------------------------------------------------------------------------------------------------------
// MODE: WS_VEC = 1x1 stride-1 layers (16-byte loads of contiguous rows); WS_TAP4 = image stems (Cin <= 4, K packed (kh, kw, 4 channel slots):
// conv_gemm.hip MODE_TAP4) -- a K-step is four taps x four channel slots, the four k rows a wave gathers are the channels of ONE tap, one shifted,
// masked per-lane offset per wave and K-step.  RELU: clamp the gathered input at 0 (the positive pass of a signed image).
enum { WS_VEC = 0, WS_TAP4 = 1 };
template <int BK, int NST, int MODE, bool RELU>
__device__ __forceinline__ void ws_math(const ConvParams& p, float* smem, float* handoff, const int wave, const int lane, const int n_co_tiles,
                                        const WsSeq sq)
{
    static_assert(MODE == WS_VEC || BK == 16, "the 4-channel tap gather is written for 16-deep K-steps (four taps)");
    constexpr int A_FLOATS = BK * 64, B_FLOATS = BK * 64, STAGE = A_FLOATS + B_FLOATS;
    constexpr int APW = A_FLOATS / 256 / 4;                                 // 16-byte wave-loads per wave and stage
    constexpr int BPW = (MODE == WS_VEC) ? B_FLOATS / 256 / 4 : B_FLOATS / 64 / 4;      // 16-byte wave-loads | 4-byte wave-loads (one k row x 64 m)
    constexpr int L = APW + BPW;
    constexpr int NP = 4;                                                   // shares the loads of a K-step are issued in (between the MFMAs)
    static_assert((NST - 2) * L <= 63, "vmcnt is a 6-bit counter");
    const int wrow = wave >> 1, wcol = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nk = (p.K + BK - 1) / BK;                                     // the packed weights are zero-padded to a multiple of 32 rows
    const unsigned chan_bytes = (unsigned)p.in_nb * p.H * p.W * 4u;
    const unsigned a_step = (unsigned)BK * p.ldw * 4u;
    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const int w_bytes = ((p.K + 31) / 32) * 32 * p.ldw * 4;

    // ---- issue cursor: (tile ordinal, K-step) of the next stage to be filled; it runs NST-1 K-steps ahead of the MFMAs, across tiles
    int is_j = sq.loc, is_kt = 0;
    unsigned voffA[APW], voffB[(MODE == WS_VEC) ? BPW : 1];
    int base_m = 0;                       // WS_TAP4: this lane's m column of the cursor's tile -- input offset of its window's corner, valid taps
    unsigned long long tapmask = 0ull;
#pragma unroll
    for (int i = 0; i < APW; ++i) voffA[i] = OOB;
#pragma unroll
    for (int i = 0; i < ((MODE == WS_VEC) ? BPW : 1); ++i) voffB[i] = OOB;
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
    auto set_issue_tile = [&](int j) {
        const WsTile t = ws_tile(p, sq.base + j, n_co_tiles);
        rW = __builtin_amdgcn_make_buffer_rsrc((void*)(t.half ? p.w_pos : p.w), 0, w_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int f = (wave * APW + i) * 256 + lane * 4;
            const int row = f >> 6, col = f & 63;
            voffA[i] = (unsigned)(row * p.ldw + t.co0 + col) * 4u;
        }
        if constexpr (MODE == WS_VEC) {
#pragma unroll
            for (int i = 0; i < BPW; ++i) {
                const int f = (wave * BPW + i) * 256 + lane * 4;
                const int row = f >> 6, col = f & 63;
                const int m = t.m0 + col;
                voffB[i] = (m < p.M) ? (unsigned)row * chan_bytes + (unsigned)m * 4u : OOB;      // k rows beyond Cin lie outside the descriptor: 0.0
            }
        } else {
            const int m = t.m0 + lane;
            const bool m_ok = m < p.M;
            const int mm = m_ok ? m : 0;
            const int ohw = p.OH * p.OW;
            const int n = mm / ohw;
            const int r = mm - n * ohw;
            const int oh = r / p.OW;
            const int ow = r - oh * p.OW;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            base_m = n * p.H * p.W + ih0 * p.W + iw0;
            tapmask = 0ull;
            if (m_ok) {
                unsigned long long vw = 0ull;
                for (int dw = 0; dw < p.kw; ++dw)
                    if ((unsigned)(iw0 + dw) < (unsigned)p.W) vw |= 1ull << dw;
                for (int dh = 0; dh < p.kh; ++dh)
                    if ((unsigned)(ih0 + dh) < (unsigned)p.H) tapmask |= vw << (dh * p.kw);
            }
        }
    };
    bool is_live = is_j < sq.cnt;
    if (is_live) set_issue_tile(is_j);
    // share `part` (0 .. NP-1, or < 0: everything) of the loads of the cursor's K-step into stage st; the cursor moves on with the last share
    auto issue = [&](int st, int part) {
        auto mine = [&](int i, int n) { return part < 0 || (i * NP) / n == part; };
        float* As = smem + st * STAGE;
        float* Bs = As + A_FLOATS;
#pragma unroll
        for (int i = 0; i < APW; ++i)
            if (mine(i, APW)) bload16(rW, As + (wave * APW + i) * 256, is_live ? voffA[i] : OOB, (unsigned)is_kt * a_step);
        if constexpr (MODE == WS_VEC) {
#pragma unroll
            for (int i = 0; i < BPW; ++i)
                if (mine(i, BPW)) bload16(rIn, Bs + (wave * BPW + i) * 256, is_live ? voffB[i] : OOB, (unsigned)(is_kt * BK) * chan_bytes);
        } else {
            if (part <= 0) {
                const int tap = is_kt * 4 + wave;
                const int dh = tap / p.kw, dw = tap - dh * p.kw;
                const bool tap_ok = is_live && tap < p.kh * p.kw;
                voffB[0] = (tap_ok && ((tapmask >> (tap & 63)) & 1ull)) ? (unsigned)(base_m + dh * p.W + dw) * 4u : OOB;
            }
#pragma unroll
            for (int i = 0; i < BPW; ++i)
                if (mine(i, BPW)) bload4(rIn, Bs + (wave * BPW + i) * 64, i < p.Cin ? voffB[0] : OOB, (unsigned)i * chan_bytes);
        }
        if (part < 0 || part == NP - 1) {
            if (++is_kt == nk) {
                is_kt = 0;
                is_j += sq.nwg;
                is_live = is_j < sq.cnt;
                if (is_live) set_issue_tile(is_j);
            }
        }
    };

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, -1);

    const int a_off = wrow * 32 + l31, b_off = wcol * 32 + l31;
    float* my_tile = handoff + wave * (32 * WS_LD);
    int st = 0;
    for (int j = sq.loc; j < sq.cnt; j += sq.nwg) {
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            // stage st has landed once at most (NST-2) younger stages are in flight; the barrier also says every wave is done reading the
            // stage about to be refilled -- and (first K-step of a tile) publishes the hand-off tile written after the previous tile
            wait_vmcnt<(NST - 2) * L>();
            __builtin_amdgcn_s_barrier();
            int st_fill = st + NST - 1;
            if (st_fill >= NST) st_fill -= NST;
            const float* As = smem + st * STAGE;
            const float* Bs = As + A_FLOATS;
            float a_cur = As[lhi * 64 + a_off], b_cur = Bs[lhi * 64 + b_off], a_nxt = 0.f, b_nxt = 0.f;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                if (kk + 2 < BK) {
                    a_nxt = As[(kk + 2 + lhi) * 64 + a_off];
                    b_nxt = Bs[(kk + 2 + lhi) * 64 + b_off];
                }
                if (RELU) b_cur = fmaxf(b_cur, 0.f);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur, acc, 0, 0, 0);
                if ((kk / 2) % (BK / 2 / NP) == 0) issue(st_fill, (kk / 2) / (BK / 2 / NP));
                a_cur = a_nxt;
                b_cur = b_nxt;
            }
            st = (st + 1 == NST) ? 0 : st + 1;
        }
        // park the tile for the epilogue waves: D[i = (r&3) + 8*(r>>2) + 4*lhi][j = l31], the layout conv_gemm.hip's epilogues turn through LDS.
        // The epilogue waves emptied the tile right after the barrier that opened this tile's K loop (nk >= 2 barriers ago).
#pragma unroll
        for (int r = 0; r < 16; ++r) my_tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * WS_LD + l31] = acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    wait_vmcnt<0>();                      // the cursor's trailing loads (nothing real) have landed before the LDS is released
    __builtin_amdgcn_s_barrier();         // publishes the last hand-off tile
}

// ---- epilogue waves -----------------------------------------------------------------------------------------------------
// What a lane keeps about the tile it is finishing -- four float4 pieces (channel group hf: row hf*8 + lane/8 of the wave's quadrant, columns
// 4*(lane%8)..+3); the per-piece indices are rebuilt from it (two multiplications), so the state is four registers, not thirty.
struct WsEpi {
    int co_lane;        // GEMM row of piece 0
    unsigned mm4;       // column piece index inside an output row
    unsigned acol4;     // ... inside a forward-side row (sample sb % B, same position); chains only
    int flags;          // bit 0: m in range, bit 1: half
};
struct WsPiece { unsigned idx4, aidx4; int co; bool ok; };

template <bool HAS_CHAIN>
__device__ __forceinline__ WsEpi ws_epi_setup(const ConvParams& p, const WsTile c, int e, int lane)
{
    const int wrow = e >> 1, wcol = e & 1;
    const int m = c.m0 + wcol * 32 + (lane & 7) * 4;
    const bool m_ok = m < p.M;
    const int mm = m_ok ? m : 0;
    WsEpi t;
    t.co_lane = c.co0 + wrow * 32 + (lane >> 3);
    t.mm4 = (unsigned)mm / 4u;
    t.acol4 = 0;
    if constexpr (HAS_CHAIN) t.acol4 = (unsigned)(mm % (p.chain_B * p.OH * p.OW)) / 4u;
    t.flags = (m_ok ? 1 : 0) | (c.half ? 2 : 0);
    return t;
}
__device__ __forceinline__ WsPiece ws_piece(const ConvParams& p, const WsEpi& t, int hf, unsigned row4, unsigned arow4)
{
    WsPiece q;
    q.co = t.co_lane + hf * 8;
    q.ok = q.co < p.CoutTot && (t.flags & 1);
    const int cc = q.ok ? out_row(p, q.co) : 0;
    q.idx4 = (unsigned)cc * row4 + t.mm4;
    q.aidx4 = (unsigned)cc * arow4 + t.acol4;
    return q;
}

// dense rows: bias, optional accumulate, one 16-byte store
__device__ __forceinline__ void ws_dense_piece(const ConvParams& p, const WsPiece& q, int half, float4 v)
{
    if (!q.ok || p.ws_debug == 1) return;          // ws_debug 1 (tuning): the math side alone -- nothing is stored
    const float* __restrict__ bsel = half ? p.bias_pos : p.bias;
    float4* dst = reinterpret_cast<float4*>(half ? p.out1 : p.out0) + q.idx4;
    if (bsel) { const float b = bsel[q.co]; v.x += b; v.y += b; v.z += b; v.w += b; }
    if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *dst = v;
}

template <int SIG>
__device__ __forceinline__ void ws_chain_piece(const ConvParams& p, const WsPiece& q, unsigned mm4, float4 gv, float4 partner, EpiOps& ops, unsigned row4,
                                               unsigned arow4)
{
    if (!q.ok) return;
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    float4* out4 = reinterpret_cast<float4*>(p.out0);
    if (p.bias) { const float b = p.bias[q.co]; g[0] += b; g[1] += b; g[2] += b; g[3] += b; }
    if constexpr (sig_has_maxpair<SIG>()) {
        if (p.bias) { const float b = p.bias[q.co ^ 1]; partner.x += b; partner.y += b; partner.z += b; partner.w += b; }
        ops.partner = partner;
        ops.co_idx4 = (q.co & 1) ? ~0u : (unsigned)(q.co >> 1) * row4 + mm4;
    }
    if constexpr (sig_has_fanout<SIG>()) { ops.out4 = out4; ops.row4 = row4; ops.arow4 = arow4; }
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    epi_steps<SIG, 0>(g, sv, ops, p.chain, q.idx4, q.aidx4, p.chain_eps);
    if constexpr (sig_has_fanout<SIG>()) {
        // stored by the fan-out
    } else if constexpr (sig_has_maxpair<SIG>()) {
        // both rows of a pair hold the maximum now; the even row stores it as channel co / 2 of the Co-channel output
        if ((q.co & 1) == 0) out4[(unsigned)(q.co >> 1) * row4 + mm4] = make_float4(g[0], g[1], g[2], g[3]);
    } else {
        out4[q.idx4] = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// SIG < 0: no chain.  The whole persistent loop is instantiated per signature: the dispatch happens once per workgroup, not per piece.
// While the math waves run tile j (nk barriers) this wave finishes tile j - 1: piece q right after barrier (q * (nk-1)) / 4 + 1 -- every piece
// has left the hand-off tile one barrier before the math waves refill it (after the nk-th), whatever nk >= 2 is.
template <int SIG>
__device__ __forceinline__ void ws_epilogue(const ConvParams& p, float* handoff, const int e, const int lane, const int n_co_tiles, const WsSeq sq,
                                            const int nk)
{
    constexpr bool HAS_CHAIN = SIG >= 0;
    constexpr int S = HAS_CHAIN ? SIG : 0;
    const int ohw = p.OH * p.OW;
    // piece (float4) indices fit 32 bits: every tensor is smaller than 2 GiB (checked when the workspace is laid out)
    const unsigned row4 = (unsigned)(p.out_nb * ohw) / 4u, arow4 = HAS_CHAIN ? (unsigned)(p.chain_B * ohw) / 4u : 0u;
    const float* tile = handoff + e * (32 * WS_LD) + (lane >> 3) * WS_LD + (lane & 7) * 4;
    WsEpi cur;
    EpiOps ops;
    bool have_prev = false;
    for (int j = sq.loc;; j += sq.nwg) {
        const bool have_tile = j < sq.cnt;                 // the math waves run tile j (nk barriers), or are past their last tile (one barrier)
        if (!have_tile && !have_prev) break;
        const int nbar = have_tile ? nk : 1;
        int done = 0;
        if (have_prev) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int target = have_tile ? ((q * (nk - 1)) >> 2) + 1 : 1;
                while (done < target) { __builtin_amdgcn_s_barrier(); ++done; }
                const float4 gv = *reinterpret_cast<const float4*>(tile + q * 8 * WS_LD);
                float4 pv = gv;
                if constexpr (HAS_CHAIN && sig_has_maxpair<S>())
                    pv = *reinterpret_cast<const float4*>(handoff + e * (32 * WS_LD) + ((q * 8 + (lane >> 3)) ^ 1) * WS_LD + (lane & 7) * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // it has left the tile before this wave meets another barrier
                const WsPiece pc = ws_piece(p, cur, q, row4, arow4);
                bool chain_here = false;
                if constexpr (HAS_CHAIN) chain_here = (cur.flags & 2) == 0;    // the relu(W) half of a dual launch leaves as plain rows
                if (chain_here) {
                    if constexpr (HAS_CHAIN) {
                        ws_chain_piece<S>(p, pc, cur.mm4, gv, pv, ops, row4, arow4);
                        // operands of the next piece: one piece ahead (the next tile's first piece is issued below, behind this tile's last)
                        if (q < 3) {
                            const WsPiece nx = ws_piece(p, cur, q + 1, row4, arow4);
                            epi_load<S>(ops, p, nx.idx4, nx.aidx4, nx.ok ? nx.co : 0);
                        }
                    }
                } else {
                    ws_dense_piece(p, pc, (cur.flags >> 1) & 1, gv);
                }
            }
        }
        if (have_tile) {
            cur = ws_epi_setup<HAS_CHAIN>(p, ws_tile(p, sq.base + j, n_co_tiles), e, lane);
            if constexpr (HAS_CHAIN) {
                if ((cur.flags & 2) == 0) {
                    const WsPiece nx = ws_piece(p, cur, 0, row4, arow4);
                    epi_load<S>(ops, p, nx.idx4, nx.aidx4, nx.ok ? nx.co : 0);
                }
            }
        }
        while (done < nbar) { __builtin_amdgcn_s_barrier(); ++done; }
        if (!have_tile) break;
        have_prev = true;
    }
}

template <int SIG, bool MFM>
__device__ __forceinline__ void ws_epilogue_dispatch(int sig, const ConvParams& p, float* handoff, int e, int lane, int n_co_tiles, const WsSeq sq, int nk)
{
    if constexpr (SIG < kNumChainSigs) {
        if constexpr (sig_is_mfm<SIG>() == MFM) {
            if (sig == SIG) { ws_epilogue<SIG>(p, handoff, e, lane, n_co_tiles, sq, nk); return; }
        }
        ws_epilogue_dispatch<SIG + 1, MFM>(sig, p, handoff, e, lane, n_co_tiles, sq, nk);
    }
}

// CHAIN: 0 = no chain, 1 = compiled chain (p.chain_sig), 3 = compiled MaxFeatureMap chain (the two register families of conv_gemm.hip)
template <int BK, int NST, int CHAIN, int MODE, bool RELU>
__global__ __launch_bounds__(WS_NT, 6) void conv_ws_kernel(const ConvParams p, const int n_co_tiles, const int n_m_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RING = NST * BK * 128;
    float* handoff = smem + RING;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WsSeq sq = ws_seq(n_co_tiles * n_m_tiles);
    if (wave == 0) stamp(p, 0, lane, 0, 2);
    if (wave < 4) {
        ws_math<BK, NST, MODE, RELU>(p, smem, handoff, wave, lane, n_co_tiles, sq);
    } else {
        const int nk = (p.K + BK - 1) / BK;
        if constexpr (CHAIN == 0) ws_epilogue<-1>(p, handoff, wave - 4, lane, n_co_tiles, sq, nk);
        else ws_epilogue_dispatch<0, CHAIN == 3>(p.chain_sig, p, handoff, wave - 4, lane, n_co_tiles, sq, nk);
        if (wave == 4) stamp(p, 0, lane, 4, 2);          // the workgroup's life ends with its last store
    }
}

int ws_num_cus()
{
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        return cu;
    }();
    return n;
}

template <int BK, int NST, int CHAIN, int MODE, bool RELU>
void ws_launch_one(const ConvParams& q, hipStream_t s)
{
    const int n_co = ((q.CoutTot + 63) / 64) * q.nhalves;
    const int n_m = (q.M + 63) / 64;
    constexpr size_t lds = ((size_t)NST * BK * 128 + 4 * 32 * WS_LD) * sizeof(float);
    // resident workgroups per CU, asked once per instantiation: a persistent grid must not be larger than what is co-resident
    static const int per_cu = [] {
        int b = 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<BK, NST, CHAIN, MODE, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, conv_ws_kernel<BK, NST, CHAIN, MODE, RELU>, WS_NT, lds) != hipSuccess || b < 1) b = 1;
        return b;
    }();
    const int grid = std::min(n_co * n_m, per_cu * ws_num_cus());
    hipLaunchKernelGGL((conv_ws_kernel<BK, NST, CHAIN, MODE, RELU>), dim3(grid), dim3(WS_NT), lds, s, q, n_co, n_m);
}

template <int BK, int NST>
void ws_launch(const ConvParams& q, int chain_kind, hipStream_t s)
{
    if (q.tap_major == 2) {
        // image stems: 16-deep K-steps only; no MaxFeatureMap stem has 3 or 4 input channels
        if (q.relu_in) ws_launch_one<16, 4, 0, WS_TAP4, true>(q, s);
        else if (chain_kind == 1) ws_launch_one<16, 4, 1, WS_TAP4, false>(q, s);
        else ws_launch_one<16, 4, 0, WS_TAP4, false>(q, s);
        return;
    }
    if (chain_kind == 3) ws_launch_one<BK, NST, 3, WS_VEC, false>(q, s);
    else if (chain_kind == 1) ws_launch_one<BK, NST, 1, WS_VEC, false>(q, s);
    else ws_launch_one<BK, NST, 0, WS_VEC, false>(q, s);
}

}  // namespace

bool conv_ws_ok(const ConvParams& p)
{
    if ((p.M & 3) != 0 || ((p.out_nb * p.OH * p.OW) & 3) != 0 || p.out_stride != 1) return false;      // dense float4 output rows
    if (p.K < 32) return false;                         // at least two K-steps per tile (the hand-off protocol counts on it)
    if (p.tap_major == 2)                               // image stem: (kh, kw, 4 channel slots) pack, at most 64 taps, one half
        return p.Cin <= 4 && p.kh * p.kw <= 64 && p.nhalves == 1 && (!p.relu_in || p.chain.n == 0) && p.co_pair == 0;
    if (!(p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W)) return false;
    return !p.relu_in;
}

// q: chain already planned (conv_gemm.hip plan_chain): chain_sig >= 0 or no chain at all
void launch_conv_ws(const ConvParams& q, int cfg, hipStream_t s)
{
    const int kind = q.chain.n > 0 ? (chain_sig_is_mfm(q.chain_sig) ? 3 : 1) : 0;
    if (cfg == 9 || cfg == 19) ws_launch<16, 4>(q, kind, s);
    else ws_launch<16, 3>(q, kind, s);
}
