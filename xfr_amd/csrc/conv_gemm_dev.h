// conv_gemm_dev.h -- device code shared by the GEMM translation units (conv_gemm.hip: K1 / K1b, conv_gemm_split.hip: K17): the epilogues of a
// 64 x 64 block tile.  Everything sits in an anonymous namespace: each translation unit compiles its own copy into its own kernels.
#pragma once
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int NT = 256;

enum { MODE_VEC = 0, MODE_TAP = 1, MODE_GEN = 2, MODE_TAP4 = 3 };

// Epilogue of one wave's 32x32 accumulator tile for a chain with signature SIG.  A lane owns, for each of the four 8-channel
// groups hf, one float4 piece (channel hf*8 + lane/8, positions 4*(lane%8)..+3).  The operand loads of groups 0 and 1 are
// issued first -- before the accumulators are turned through LDS -- and those of group hf+2 right after group hf's have been
// consumed, so two groups' worth of HBM requests are in flight per lane instead of one dependent round trip per group.
// Lean probe-forward signatures (sig_is_dual): accp is the wave's relu(W) accumulator tile of the same quadrant; it is turned through the same LDS
// tile first and kept as four pieces (+ bias_pos) in registers.
template <int SIG>
__device__ __forceinline__ void chain_epilogue(const ConvParams& p, const v16f& acc, const v16f* accp, float* tile, int lane, int l31, int lhi,
                                               int co_base, int m, float* __restrict__ osel, const float* __restrict__ bsel)
{
    constexpr int LD = 36;
    const int ohw = p.OH * p.OW;
    // piece (float4) indices fit 32 bits: every tensor is smaller than 2 GiB (checked when the workspace is laid out)
    const unsigned row4 = (unsigned)(p.out_nb * ohw) / 4u, arow4 = (unsigned)(p.chain_B * ohw) / 4u;
    const int mq = (lane & 7) * 4;
    const bool m_ok = m < p.M;
    const int mm = m_ok ? m : 0;
    const unsigned acol4 = (unsigned)(mm % (p.chain_B * ohw)) / 4u;      // forward-side column of this piece: sample sb % B, same position
    EpiOps ops[4];
    unsigned idx4[4], aidx4[4];
    int cos[4];
    bool ok[4];
#pragma unroll
    for (int hf = 0; hf < 4; ++hf) {
        cos[hf] = co_base + hf * 8 + (lane >> 3);
        ok[hf] = cos[hf] < p.CoutTot && m_ok;
        const int cc = ok[hf] ? out_row(p, cos[hf]) : 0;
        idx4[hf] = (unsigned)cc * row4 + (unsigned)mm / 4u;
        aidx4[hf] = (unsigned)cc * arow4 + acol4;
    }
    // channel groups in flight: two for chains with at most two operand tensors, one otherwise -- measured on MI355X: 1 and 2
    // groups time the same (26.5 ms per step), 4 cost occupancy (27.4 ms); one group for the wide chains keeps the kernel at
    // 64-65 VGPRs (5-6 waves per SIMD)
    constexpr unsigned live_slots = sig_live_slots<SIG>();
    constexpr int n_live = __builtin_popcount(live_slots);
    constexpr int DEPTH = n_live >= 3 ? 1 : 2;
#pragma unroll
    for (int hf = 0; hf < DEPTH; ++hf) epi_load<SIG>(ops[hf], p, idx4[hf], aidx4[hf], ok[hf] ? cos[hf] : 0);
    // every wave is done reading the ring (its LDS reads fed MFMAs that have retired); a raw barrier, not
    // __syncthreads(): that one would first drain the operand loads just issued
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float4 gpv[4];
    if constexpr (sig_is_dual<SIG>()) {
        // the tile is this wave's own: its LDS operations execute in order, the waits only keep the compiler from moving them
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * LD + l31] = (*accp)[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int hf = 0; hf < 4; ++hf) {
            gpv[hf] = *reinterpret_cast<const float4*>(tile + (hf * 8 + (lane >> 3)) * LD + mq);
            if (p.bias_pos && ok[hf]) { const float b = p.bias_pos[cos[hf]]; gpv[hf].x += b; gpv[hf].y += b; gpv[hf].z += b; gpv[hf].w += b; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * LD + l31] = acc[r];
    float4* out4 = reinterpret_cast<float4*>(osel);
#pragma unroll
    for (int hf = 0; hf < 4; ++hf) {
        const int cl = hf * 8 + (lane >> 3);
        const float4 gv = *reinterpret_cast<const float4*>(tile + cl * LD + mq);
        float g[4] = {gv.x, gv.y, gv.z, gv.w};
        LeanRegs lr;
        if constexpr (sig_is_dual<SIG>()) { lr.gp[0] = gpv[hf].x; lr.gp[1] = gpv[hf].y; lr.gp[2] = gpv[hf].z; lr.gp[3] = gpv[hf].w; }
        if (bsel && ok[hf]) { const float b = bsel[cos[hf]]; g[0] += b; g[1] += b; g[2] += b; g[3] += b; }
        if constexpr (sig_has_maxpair<SIG>()) {
            float4 w = *reinterpret_cast<const float4*>(tile + (cl ^ 1) * LD + mq);
            if (bsel && ok[hf]) { const float b = bsel[cos[hf] ^ 1]; w.x += b; w.y += b; w.z += b; w.w += b; }
            ops[hf].partner = w;
            ops[hf].co_idx4 = (cos[hf] & 1) ? ~0u : (unsigned)(cos[hf] >> 1) * row4 + (unsigned)mm / 4u;
        }
        if constexpr (sig_has_fanout<SIG>()) { ops[hf].out4 = out4; ops[hf].row4 = row4; ops[hf].arow4 = arow4; }
        if (ok[hf]) {
            float sv[4] = {0.f, 0.f, 0.f, 0.f};
            epi_steps<SIG, 0>(g, sv, lr, ops[hf], p.chain, idx4[hf], aidx4[hf], p.chain_eps);
            if constexpr (sig_has_fanout<SIG>()) {
                // stored by the fan-out
            } else if constexpr (sig_has_maxpair<SIG>()) {
                // both rows of a pair hold the maximum now; the even row stores it as channel cos / 2 of the Co-channel output
                if ((cos[hf] & 1) == 0) out4[(unsigned)(cos[hf] >> 1) * row4 + (unsigned)mm / 4u] = make_float4(g[0], g[1], g[2], g[3]);
            } else {
                out4[idx4[hf]] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
        if (hf + DEPTH < 4) epi_load<SIG>(ops[hf + DEPTH], p, idx4[hf + DEPTH], aidx4[hf + DEPTH], ok[hf + DEPTH] ? cos[hf + DEPTH] : 0);
    }
}

// Dense output rows without a chain: turn the wave's 32x32 accumulator tile through LDS (the ring is free) so that a lane holds
// four consecutive m of one channel and the tile leaves in 4 dwordx4 stores per lane instead of 16 dword stores (8 x 128-byte
// rows per store instruction instead of 2).
__device__ __forceinline__ void dense_epilogue(const ConvParams& p, const v16f& acc, float* tile, int lane, int l31, int lhi, int co_base,
                                               int m, float* __restrict__ osel, const float* __restrict__ bsel)
{
    constexpr int LD = 36;                                   // row pitch in floats: 16-byte aligned rows, no bank clash
    __syncthreads();                                         // every wave is done reading the ring
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * LD + l31] = acc[r];
    const long row_stride = (long)p.out_nb * p.OH * p.OW;
    const int mq = (lane & 7) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int cl = it * 8 + (lane >> 3);
        const int co = co_base + cl;
        float4 v = *reinterpret_cast<const float4*>(tile + cl * LD + mq);
        if (co < p.CoutTot && m < p.M) {
            float4* dst = reinterpret_cast<float4*>(osel + (long)out_row(p, co) * row_stride + m);
            if (bsel) { const float b = bsel[co]; v.x += b; v.y += b; v.z += b; v.w += b; }
            if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *dst = v;
        }
    }
}

// The compiled epilogues live in two kernel families: the MaxFeatureMap signatures (pair maximum, fan-out VJP: two more operand
// loads and a chain tail that runs twice) need ~10 registers more than the rest, and a kernel's register count -- hence how many
// workgroups share a CU -- is the maximum over everything it contains.  MFM = false: every other signature (all ResNet chains).

// ... FAM 2: the lean probe-forward signatures, which need the second accumulator tile (kernels compiled with CHAIN == 4)
template <int SIG>
constexpr int sig_family() { return sig_is_dual<SIG>() ? 2 : (sig_is_mfm<SIG>() ? 1 : 0); }

template <int SIG, int FAM>
__device__ __forceinline__ void chain_epilogue_dispatch(int sig, const ConvParams& p, const v16f& acc, const v16f* accp, float* tile, int lane, int l31,
                                                        int lhi, int co_base, int m, float* __restrict__ osel, const float* __restrict__ bsel)
{
    if constexpr (SIG < kNumChainSigs) {
        if constexpr (sig_family<SIG>() == FAM) {
            if (sig == SIG) { chain_epilogue<SIG>(p, acc, accp, tile, lane, l31, lhi, co_base, m, osel, bsel); return; }
        }
        chain_epilogue_dispatch<SIG + 1, FAM>(sig, p, acc, accp, tile, lane, l31, lhi, co_base, m, osel, bsel);
    }
}

// Everything after the K loop of a 64x64 block tile whose wave (wrow, wcol) holds the 32x32 quadrant acc[0][0]: the exchange of a
// tail tile's K-parts, then the epilogue.  Shared by the two kernels below.  CHAIN: 0 = plain epilogue, 1 = compiled chain epilogue
// (p.chain_sig), 2 = interpreted chain epilogue; LDS_OK: the workgroup's LDS holds the four 32 x 36 transposition tiles.
// ROW: the four waves lie side by side along m (a 32 x 128 block tile) instead of 2 x 2 (64 x 64); either way a wave holds one 32 x 32 quadrant.
template <int CHAIN, bool LDS_OK, bool ROW = false>
__device__ __forceinline__ void block_epilogue(const ConvParams& p, v16f (&acc)[1][1], float* smem, const int tid, const int lane, const int wave,
                                               const int co0, const int m0, const int half, const int tail_t, const int part, const int nparts,
                                               float* __restrict__ osel, const float* __restrict__ bsel, v16f* accp = nullptr)
{
    constexpr int MI = 1, NJ = 1, TCO = 64, TM = 64;         // (TCO / 2, TM / 2 below = the 32 rows / columns of a wave's quadrant in both layouts)
    const int wrow = ROW ? 0 : wave >> 1, wcol = ROW ? wave : wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (tail_t >= 0) {
        // K-part of a tail tile: park the accumulators, count arrivals; the last part to arrive sums all parts in
        // part order (deterministic) and runs the normal epilogue.  Agent-scope stores / loads: the parts ran on
        // different XCDs, whose L2s are not coherent for plain accesses.
        constexpr int TILE_FLOATS = TCO * TM * (CHAIN == 4 ? 2 : 1);       // a dual-accumulator launch parks both tiles
        float* __restrict__ slab = p.tail_ws + (long)(tail_t * nparts + part) * TILE_FLOATS;
        if constexpr (CHAIN == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(slab + (16 + r) * NT + tid, (*accp)[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(slab + ((i * NJ + j) * 16 + r) * NT + tid, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // write-through (agent-scope) stores, then only wait for them: a full __threadfence() would write back AND
        // invalidate this XCD's whole L2 under the other resident workgroups (measured: 37 us per launch at 256 parts)
        wait_vmcnt<0>();
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const unsigned old = atomicAdd(p.tail_cnt + tail_t, 1u);
            const int last = (old == (unsigned)(nparts - 1));
            if (last) atomicExch(p.tail_cnt + tail_t, 0u);     // ready for the next launch on this stream
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        const float* base = p.tail_ws + (long)tail_t * nparts * TILE_FLOATS;
        // Part by part, the sixteen registers of a part in flight together (round 6: with the part loop innermost every one of the 16 x nparts
        // agent-scope loads waited for the one before it -- 4.2 us per part, 40 us of a 43 us launch at eight parts).  Same sums in the same
        // order: element r adds part 0, 1, 2, ... to 0.0f.
        static_assert(MI == 1 && NJ == 1, "one quadrant per wave");
        constexpr int NACC = CHAIN == 4 ? 32 : 16;
        float sum[NACC];
#pragma unroll
        for (int r = 0; r < NACC; ++r) sum[r] = 0.f;
        for (int q = 0; q < nparts; ++q) {
            const float* src = base + (long)q * TILE_FLOATS + tid;
            float v[NACC];
#pragma unroll
            for (int r = 0; r < NACC; ++r) v[r] = __hip_atomic_load(src + r * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < NACC; ++r) sum[r] += v[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = sum[r];
        if constexpr (CHAIN == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) (*accp)[r] = sum[16 + r];
        }
    }

    // ---- epilogue: D[i = (r&3) + 8*(r>>2) + 4*(lane>>5)][j = lane&31]
    // Optional fused micro-program (half 0 only): forward = bias -> [store raw] -> BatchNorm -> [+residual] -> ReLU;
    // backward = [+fan-in gradient] -> tensor hooks / ReLU mask / BatchNorm VJP -> next GEMM's input.
    if constexpr (CHAIN == 4) {
        // lean probe forward: the wave holds the W and the relu(W) tile of its quadrant; the compiled chain stores what the sweep needs of them
        if constexpr (MI == 1 && NJ == 1)
            chain_epilogue_dispatch<0, 2>(p.chain_sig, p, acc[0][0], accp, smem + wave * (32 * 36), lane, l31, lhi, co0 + wrow * 32,
                                          m0 + wcol * 32 + (lane & 7) * 4, osel, bsel);
    } else if constexpr (CHAIN == 1 || CHAIN == 3) {
        // compiled chain epilogue (CHAIN 3: the MaxFeatureMap signatures); launch_one only selects this instantiation when the float4 layout conditions hold.  The chain
        // belongs to half 0; the relu(W) half of a dual launch (positive activations) leaves as plain dense rows.
        if constexpr (MI == 1 && NJ == 1) {
            if (half == 0)
                chain_epilogue_dispatch<0, CHAIN == 3 ? 1 : 0>(p.chain_sig, p, acc[0][0], nullptr, smem + wave * (32 * 36), lane, l31, lhi, co0 + wrow * 32,
                                           m0 + wcol * 32 + (lane & 7) * 4, osel, bsel);
            else
                dense_epilogue(p, acc[0][0], smem + wave * (32 * 36), lane, l31, lhi, co0 + wrow * 32, m0 + wcol * 32 + (lane & 7) * 4, osel, bsel);
        }
    } else if constexpr (CHAIN == 2) {
        // Epilogue with a fused micro-program (backward: [+fan-in gradient] -> tensor hooks / ReLU mask / BatchNorm VJP ->
        // the next GEMM's input).  The 16 accumulator registers of a 32x32 tile are 4 groups of 4 consecutive output
        // channels; they are processed group by group, and all per-element operands of a group (plan in
        // p.chain_ld, <= 4 distinct tensors) are in flight together before the steps are interpreted: every workgroup of
        // a launch reaches its epilogue at the same time, so a chain of dependent loads here is paid in full.
        const EwLoads& ld = p.chain_ld;
        // The interpreter indexes the chain's steps with a run-time index: on the by-value kernel argument that index sends the whole ConvParams to
        // scratch (2 KB per lane, every step read back from there: a stage-4 launch of a one-image call took 75 us where the compiled chain takes 23).
        // Read the steps from the kernel-argument segment itself -- ConvParams is the first argument of every GEMM kernel -- through the constant cache.
        const EwChain& kchain = ((const ConvParams*)__builtin_amdgcn_kernarg_segment_ptr())->chain;
        // float4 pieces need rows whose length is a multiple of 4 on both sides (gradient rows of out_nb images, forward
        // rows of chain_B images); a piece may then straddle two samples (7x7 maps) but never a row
        const bool vec_ok = MI == 1 && NJ == 1 && LDS_OK && (p.M & 3) == 0 &&
                            ((p.chain_B * p.OH * p.OW) & 3) == 0 && ((p.out_nb * p.OH * p.OW) & 3) == 0;
        if (vec_ok) {
            // vector path: the tile is turned through LDS like in the plain epilogue, a lane then owns float4 pieces
            // (one channel, four consecutive positions of one sample) and runs the same float4 interpreter as the
            // stand-alone chain kernel, operands fetched as 16-byte loads
            constexpr int LD = 36;
            __syncthreads();
            float* tile = smem + wave * (32 * LD);
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * LD + l31] = acc[0][0][r];
            const int ohw = p.OH * p.OW;
            const long row4 = (long)p.out_nb * ohw / 4, arow4 = (long)p.chain_B * ohw / 4;
            const int mq = (lane & 7) * 4;
            const int m = m0 + wcol * 32 + mq;
            const int mm = m < p.M ? m : 0;
            const int sb = mm / ohw;
            const long acol4 = (mm % (p.chain_B * ohw)) / 4;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4* out4 = reinterpret_cast<float4*>(osel);
#pragma unroll
            for (int hf = 0; hf < 4; ++hf) {
                float4 g[1], od[1], v0[1], v1[1], v2[1], v3[1];
                long idx4[1], aidx4[1];
                bool ok[1];
                int cos[1];
#pragma unroll
                for (int u = 0; u < 1; ++u) {
                    const int cl = (hf + u) * 8 + (lane >> 3);
                    cos[u] = co0 + wrow * 32 + cl;
                    ok[u] = cos[u] < p.CoutTot && m < p.M;
                    const int cc = ok[u] ? out_row(p, cos[u]) : 0;
                    idx4[u] = (long)cc * row4 + mm / 4;
                    aidx4[u] = (long)cc * arow4 + acol4;
                    g[u] = *reinterpret_cast<const float4*>(tile + cl * LD + mq);
                    v0[u] = v1[u] = v2[u] = v3[u] = od[u] = z4;
                    if (ld.lp[0]) v0[u] = reinterpret_cast<const float4*>(ld.lp[0])[aidx4[u]];
                    if (ld.lp[1]) v1[u] = reinterpret_cast<const float4*>(ld.lp[1])[aidx4[u]];
                    if (ld.lp[2]) v2[u] = reinterpret_cast<const float4*>(ld.lp[2])[aidx4[u]];
                    if (ld.lp[3]) v3[u] = reinterpret_cast<const float4*>(ld.lp[3])[idx4[u]];
                    if (p.accumulate) od[u] = out4[idx4[u]];
                    if (bsel && ok[u]) { const float b = bsel[cos[u]]; g[u].x += b; g[u].y += b; g[u].z += b; g[u].w += b; }
                }
#pragma unroll
                for (int u = 0; u < 1; ++u)
                    ew_interpret<false>(ok[u], idx4[u], aidx4[u], sb, 0, g[u], od[u], v0[u], v1[u], v2[u], v3[u], out4, p.accumulate,
                                        kchain, cos[u], p.chain_eps);
            }
        } else
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = m0 + wcol * (TM / 2) + j * 32 + l31;
            if (m >= p.M) continue;
            const int ohw = p.OH * p.OW;
            const long col = m;                                   // chains are only fused into dense (out_stride 1) launches
            const long row_stride = (long)p.out_nb * ohw;
            const int sb = m / ohw;
            const int hw = m - sb * ohw;
            const long acol = (long)(sb % p.chain_B) * ohw + hw;
            const long arow = (long)p.chain_B * ohw;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int hf = 0; hf < 4; ++hf) {
                    float g[4], pv0[4], pv1[4], pv2[4], pv3[4], sv[4] = {0.f, 0.f, 0.f, 0.f};
                    int gi[4], ai[4];
                    bool ok[4];
#pragma unroll
                    for (int e8 = 0; e8 < 4; ++e8) {
                        const int rg = hf, q = e8;
                        const int co = co0 + wrow * (TCO / 2) + i * 32 + 4 * lhi + 8 * rg + q;
                        ok[e8] = co < p.CoutTot;
                        const int cc = ok[e8] ? out_row(p, co) : 0;
                        gi[e8] = (int)((long)cc * row_stride + col);
                        ai[e8] = (int)((long)cc * arow + acol);
                        pv0[e8] = pv1[e8] = pv2[e8] = pv3[e8] = 0.f;
                    }
#pragma unroll
                    for (int e8 = 0; e8 < 4; ++e8) {
                        if (ld.lp[0]) pv0[e8] = ld.lp[0][ld.lk[0] ? gi[e8] : ai[e8]];
                        if (ld.lp[1]) pv1[e8] = ld.lp[1][ld.lk[1] ? gi[e8] : ai[e8]];
                        if (ld.lp[2]) pv2[e8] = ld.lp[2][ld.lk[2] ? gi[e8] : ai[e8]];
                        if (ld.lp[3]) pv3[e8] = ld.lp[3][ld.lk[3] ? gi[e8] : ai[e8]];
                    }
#pragma unroll
                    for (int e8 = 0; e8 < 4; ++e8) {
                        const int rg = hf, q = e8;
                        const int co = co0 + wrow * (TCO / 2) + i * 32 + 4 * lhi + 8 * rg + q;
                        float v = acc[i][j][rg * 4 + q];
                        if (ok[e8]) {
                            if (bsel) v += bsel[co];
                            if (p.accumulate) v += osel[gi[e8]];
                        }
                        g[e8] = v;
                    }
#pragma unroll 1
                    for (int sidx = 0; sidx < kchain.n; ++sidx) {
                        const EwStep& st = kchain.s[sidx];
                        const int type = st.type, s0 = st.ls0, s1 = st.ls1;
                        if (type == EW_HOOK) {
                            if (s0 == -2) {                      // p is not observed: relu(g) or the identity
                                if (st.action == HOOK_RELU) {
#pragma unroll
                                    for (int e8 = 0; e8 < 4; ++e8) g[e8] = fmaxf(g[e8], 0.f);
                                }
                                continue;
                            }
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) {
                                if (!ok[e8]) continue;
                                const float a_raw = s0 >= 0 ? pick4(pv0[e8], pv1[e8], pv2[e8], pv3[e8], s0) : st.p0[ai[e8]];
                                const float a = fmaxf(a_raw, 0.f);
                                const float zh = fmaxf(g[e8], 0.f);
                                if (st.action >= HOOK_Q) {      // lean hooks (common.h)
                                    g[e8] = st.action == HOOK_Q ? zh * fabsf(a_raw)
                                                                : (st.action == HOOK_GATE ? (a_raw > 0.f ? zh : 0.f) : ((__float_as_uint(a_raw) >> 31) ? 0.f : zh));
                                    continue;
                                }
                                const float pp = a * zh;
                                if (st.pstore) st.pstore[gi[e8]] = pp;
                                if (st.action == HOOK_DIV) {
                                    const float x = st.p1 ? fmaxf(s1 >= 0 ? pick4(pv0[e8], pv1[e8], pv2[e8], pv3[e8], s1) : st.p1[ai[e8]], 0.f) : a;
                                    g[e8] = __fdiv_rn(pp, x + p.chain_eps);
                                } else if (st.action == HOOK_RELU) {
                                    g[e8] = zh;
                                }
                            }
                        } else if (type == EW_MASK) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) {
                                if (!ok[e8]) continue;
                                const float t = s0 >= 0 ? pick4(pv0[e8], pv1[e8], pv2[e8], pv3[e8], s0) : st.p0[ai[e8]];
                                g[e8] = (st.action == 1 ? (__float_as_uint(t) >> 31) == 0u : t > 0.f) ? g[e8] : 0.f;
                            }
                        } else if (type == EW_SCALE_C) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8)
                                if (ok[e8]) g[e8] *= st.p0[co0 + wrow * (TCO / 2) + i * 32 + 4 * lhi + 8 * hf + e8];
                        } else if (type == EW_SCALE) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) g[e8] *= st.f;
                        } else if (type == EW_STORE) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) {
                                if (st.action == 1) { sv[e8] = g[e8]; continue; }
                                if (ok[e8]) st.pstore[gi[e8]] = g[e8];
                                if (st.action == 2) g[e8] = sv[e8];
                            }
                        } else if (type == EW_ADDP) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8)
                                if (ok[e8]) g[e8] += s0 >= 0 ? pick4(pv0[e8], pv1[e8], pv2[e8], pv3[e8], s0) : st.p0[gi[e8]];
                        } else if (type == EW_AFFINE_C) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) {
                                const int co = co0 + wrow * (TCO / 2) + i * 32 + 4 * lhi + 8 * hf + e8;
                                if (ok[e8]) g[e8] = __fadd_rn(__fmul_rn(g[e8], st.p0[co]), st.p1[co]);
                            }
                        } else if (type == EW_RELU) {
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) g[e8] = fmaxf(g[e8], 0.f);
                        } else {   // EW_FORK_POSBN
#pragma unroll
                            for (int e8 = 0; e8 < 4; ++e8) {
                                const int co = co0 + wrow * (TCO / 2) + i * 32 + 4 * lhi + 8 * hf + e8;
                                if (!ok[e8]) continue;
                                float v = __fadd_rn(__fmul_rn(fmaxf(g[e8], 0.f), st.p0[co]), st.p1[co]);
                                if (st.p2) v = __fadd_rn((st.action & 1) ? fmaxf(st.p2[gi[e8]], 0.f) : st.p2[gi[e8]], v);
                                st.pstore[gi[e8]] = v;
                            }
                        }
                    }
#pragma unroll
                    for (int e8 = 0; e8 < 4; ++e8)
                        if (ok[e8]) osel[gi[e8]] = g[e8];
                }
            }
        }
    } else if (MI == 1 && NJ == 1 && LDS_OK && p.out_stride == 1 && (p.M & 3) == 0 && ((p.out_nb * p.OH * p.OW) & 3) == 0) {
        if constexpr (MI == 1 && NJ == 1)
            dense_epilogue(p, acc[0][0], smem + wave * (32 * 36), lane, l31, lhi, co0 + wrow * 32, m0 + wcol * 32 + (lane & 7) * 4, osel, bsel);
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = m0 + wcol * (TM / 2) + j * 32 + l31;
            if (m >= p.M) continue;
            long col;
            long row_stride;
            const int ohw = p.OH * p.OW;
            if (p.out_stride == 1) {
                col = m;
                row_stride = (long)p.out_nb * ohw;
            } else {
                const int n = m / ohw;
                const int r = m - n * ohw;
                const int oh = r / p.OW;
                const int ow = r - oh * p.OW;
                col = ((long)n * p.out_H + (long)oh * p.out_stride) * p.out_W + (long)ow * p.out_stride;
                row_stride = (long)p.out_nb * p.out_H * p.out_W;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wrow * (TCO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co >= p.CoutTot) continue;
                    const long gi = (long)out_row(p, co) * row_stride + col;
                    float v = acc[i][j][r];
                    if (bsel) v += bsel[co];
                    if (p.accumulate) v += osel[gi];
                    osel[gi] = v;
                }
        }
    }
}

}  // namespace
