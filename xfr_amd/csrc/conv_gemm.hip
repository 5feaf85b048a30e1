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
#include "common.h"
#include "ew_interp.h"
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int NT = 256;

// XCD-aware block -> tile mapping.  The dispatcher places block b on XCD b % 8; remap so that each XCD walks a
// contiguous range of logical tiles, ordered co-fastest: the blocks that share one activation (m) tile run
// back-to-back on the same XCD and hit its private L2.  Bijective for any grid size.
__device__ inline int xcd_remap(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

typedef __attribute__((address_space(3))) void* lptr_t;

// Buffer-addressed global -> LDS loads.  address = rsrc.base + voffset (per lane) + soffset (wave-uniform); a lane
// whose voffset is >= rsrc.num_records (we use 0x80000000 for masked im2col elements) is out of range and the
// hardware writes 0.0 into its LDS slot -- padding, M tails and K tails cost no branch and no zero page.
__device__ inline void bload16(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)l, 16, (int)voff, (int)soff, 0, 0);
}
__device__ inline void bload4(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)l, 4, (int)voff, (int)soff, 0, 0);
}

constexpr unsigned OOB = 0x80000000u;

enum { MODE_VEC = 0, MODE_TAP = 1, MODE_GEN = 2 };

template <int N>
__device__ inline void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float pick4(float a0, float a1, float a2, float a3, int slot)
{   // slot is wave-uniform
    float r = a0;
    r = slot == 1 ? a1 : r;
    r = slot == 2 ? a2 : r;
    r = slot == 3 ? a3 : r;
    return r;
}

template <int TCO, int TM, int BK, int NST, int MODE, bool RELU, bool CHAIN>
__global__ __launch_bounds__(NT) void conv_gemm_kernel(const ConvParams p, const int n_co_tiles, const int n_m_tiles)
{
    constexpr int MI = TCO / 64;   // MFMA tiles per wave along co
    constexpr int NJ = TM / 64;    // MFMA tiles per wave along m
    constexpr int A_FLOATS = BK * TCO, B_FLOATS = BK * TM;
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_PER_WAVE = A_FLOATS / 256 / 4;                        // 16-byte wave-loads per wave per stage
    constexpr int B_PER_WAVE = (MODE == MODE_VEC) ? B_FLOATS / 256 / 4    // 16-byte wave-loads
                                                  : B_FLOATS / 64 / 4;    // 4-byte wave-loads (one k row x 64 m)
    constexpr int L = A_PER_WAVE + B_PER_WAVE;
    constexpr int MSLOTS = TM / 64;                                       // m columns owned by a lane in the dword modes

    extern __shared__ __attribute__((aligned(16))) float smem[];

    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (p.dbg_ts) ts0 = __builtin_readcyclecounter();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave >> 1, wcol = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // n_co_tiles counts the tiles of BOTH halves of a dual launch (half 1 = relu(W) -> positive activations)
    // split-K: blocks [part * n_tiles, (part+1) * n_tiles) reduce K-steps [kt_lo, kt_hi) of every tile into a scratch slab
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
        const int lid_all = xcd_remap(blockIdx.x, n_tiles_all * p.ksplit);
        part = lid_all / n_tiles_all;
        lid = lid_all - part * n_tiles_all;
        nparts = p.ksplit;
    }
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
        const int f = (wave * A_PER_WAVE + i) * 256 + lane * 4;
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
            if (MODE == MODE_TAP && m_ok) {
                for (int dh = 0; dh < p.kh; ++dh)
                    for (int dw = 0; dw < p.kw; ++dw)
                        if ((unsigned)(ih0[s] + dh) < (unsigned)p.H && (unsigned)(iw0[s] + dw) < (unsigned)p.W)
                            mk |= 1ull << (dh * p.kw + dw);
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
        for (int i = 0; i < A_PER_WAVE; ++i)
            if (mine(i, A_PER_WAVE)) bload16(rW, As + (wave * A_PER_WAVE + i) * 256, live ? voffA[i] : OOB, (unsigned)kt * a_step);
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

    v16f acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: fill NST-1 stages
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(kt_lo + s, s, -1);

    const int a_off = wrow * (TCO / 2) + l31;
    const int b_off = wcol * (TM / 2) + l31;

    int st = 0;
    for (int kt = kt_lo; kt < nk; ++kt) {
        // stage kt has landed once at most (NST-2) younger stages are still in flight; the barrier then also
        // guarantees every wave is done reading the stage we are about to refill.
        wait_vmcnt<(NST - 2) * L>();
        __builtin_amdgcn_s_barrier();
        if (p.dbg_ts && kt == kt_lo) ts1 = __builtin_readcyclecounter();
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
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
    if (p.dbg_ts) ts2 = __builtin_readcyclecounter();

    if (tail_t >= 0) {
        // K-part of a tail tile: park the accumulators, count arrivals; the last part to arrive sums all parts in
        // part order (deterministic) and runs the normal epilogue.  Agent-scope stores / loads: the parts ran on
        // different XCDs, whose L2s are not coherent for plain accesses.
        constexpr int TILE_FLOATS = TCO * TM;
        float* __restrict__ slab = p.tail_ws + (long)(tail_t * nparts + part) * TILE_FLOATS;
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
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sum = 0.f;
                    for (int q = 0; q < nparts; ++q)
                        sum += __hip_atomic_load(base + (long)q * TILE_FLOATS + ((i * NJ + j) * 16 + r) * NT + tid,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acc[i][j][r] = sum;
                }
    }

    // ---- epilogue: D[i = (r&3) + 8*(r>>2) + 4*(lane>>5)][j = lane&31]
    // Optional fused micro-program (half 0 only): forward = bias -> [store raw] -> BatchNorm -> [+residual] -> ReLU;
    // backward = [+fan-in gradient] -> tensor hooks / ReLU mask / BatchNorm VJP -> next GEMM's input.
    if (p.ksplit > 1) {
        // raw partial sums -> scratch[part][co][m]; bias / accumulate / layout are applied by splitk_reduce_kernel
        float* __restrict__ slab = p.splitk_ws + (long)part * p.CoutTot * p.M;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int m = m0 + wcol * (TM / 2) + j * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wrow * (TCO / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co < p.CoutTot) slab[(long)co * p.M + m] = acc[i][j][r];
                }
        }
        return;
    }
    if constexpr (CHAIN) {
        // Epilogue with a fused micro-program (backward: [+fan-in gradient] -> tensor hooks / ReLU mask / BatchNorm VJP ->
        // the next GEMM's input).  The 16 accumulator registers of a 32x32 tile are 4 groups of 4 consecutive output
        // channels; they are processed group by group, and all per-element operands of a group (plan in
        // p.chain_ld, <= 4 distinct tensors) are in flight together before the steps are interpreted: every workgroup of
        // a launch reaches its epilogue at the same time, so a chain of dependent loads here is paid in full.
        const EwLoads& ld = p.chain_ld;
        if (MI == 1 && NJ == 1 && NST * STAGE >= 4 * 32 * 36 && (p.M & 3) == 0 && ((p.OH * p.OW) & 3) == 0) {
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
            const int hw = mm - sb * ohw;
            const long acol4 = ((long)(sb % p.chain_B) * ohw + hw) / 4;
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
                    const int cc = ok[u] ? cos[u] : 0;
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
                                        p.chain, cos[u], p.chain_eps);
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
                    float g[4], pv0[4], pv1[4], pv2[4], pv3[4];
                    int gi[4], ai[4];
                    bool ok[4];
#pragma unroll
                    for (int e8 = 0; e8 < 4; ++e8) {
                        const int rg = hf, q = e8;
                        const int co = co0 + wrow * (TCO / 2) + i * 32 + 4 * lhi + 8 * rg + q;
                        ok[e8] = co < p.CoutTot;
                        const int cc = ok[e8] ? co : 0;
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
                    for (int sidx = 0; sidx < p.chain.n; ++sidx) {
                        const EwStep& st = p.chain.s[sidx];
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
                                const float a = fmaxf(s0 >= 0 ? pick4(pv0[e8], pv1[e8], pv2[e8], pv3[e8], s0) : st.p0[ai[e8]], 0.f);
                                const float zh = fmaxf(g[e8], 0.f);
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
                                g[e8] = t > 0.f ? g[e8] : 0.f;
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
                            for (int e8 = 0; e8 < 4; ++e8)
                                if (ok[e8]) st.pstore[gi[e8]] = g[e8];
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
                                if (ok[e8]) st.pstore[gi[e8]] = __fadd_rn(__fmul_rn(fmaxf(g[e8], 0.f), st.p0[co]), st.p1[co]);
                            }
                        }
                    }
#pragma unroll
                    for (int e8 = 0; e8 < 4; ++e8)
                        if (ok[e8]) osel[gi[e8]] = g[e8];
                }
            }
        }
    } else if (MI == 1 && NJ == 1 && NST * STAGE >= 4 * 32 * 36 && p.out_stride == 1 && (p.M & 3) == 0 && ((p.out_nb * p.OH * p.OW) & 3) == 0) {
        // Dense output rows: turn the wave's 32x32 accumulator tile through LDS (the ring is free) so that a lane holds
        // four consecutive m of one channel and the tile leaves in 4 dwordx4 stores per lane instead of 16 dword stores
        // (8 x 128-byte rows per store instruction instead of 2).
        constexpr int LD = 36;                                   // row pitch in floats: 16-byte aligned rows, no bank clash
        __syncthreads();                                         // every wave is done reading the ring
        float* tile = smem + wave * (32 * LD);
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lhi) * LD + l31] = acc[0][0][r];
        const long row_stride = (long)p.out_nb * p.OH * p.OW;
        const int mq = (lane & 7) * 4;
        const int m = m0 + wcol * 32 + mq;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cl = it * 8 + (lane >> 3);
            const int co = co0 + wrow * 32 + cl;
            float4 v = *reinterpret_cast<const float4*>(tile + cl * LD + mq);
            if (co < p.CoutTot && m < p.M) {
                float4* dst = reinterpret_cast<float4*>(osel + (long)co * row_stride + m);
                if (bsel) { const float b = bsel[co]; v.x += b; v.y += b; v.z += b; v.w += b; }
                if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *dst = v;
            }
        }
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
                    const long gi = (long)co * row_stride + col;
                    float v = acc[i][j][r];
                    if (bsel) v += bsel[co];
                    if (p.accumulate) v += osel[gi];
                    osel[gi] = v;
                }
        }
    }
    if (p.dbg_ts && tid == 0) {
        unsigned long long* d = p.dbg_ts + (size_t)blockIdx.x * 5;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_readcyclecounter();
        d[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
    }
}

// out[co][m] = sum_part slab[part][co][m] + bias[co] (+ out[co][m])
__global__ __launch_bounds__(NT) void splitk_reduce_kernel(const float4* __restrict__ ws, float4* __restrict__ out,
                                                          const float* __restrict__ bias, int parts, int Cout, long M4,
                                                          int accumulate)
{
    const long total = (long)Cout * M4;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        float4 a = ws[i];
        for (int q = 1; q < parts; ++q) {
            const float4 b = ws[(long)q * total + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) { const float bb = bias[(int)(i / M4)]; a.x += bb; a.y += bb; a.z += bb; a.w += bb; }
        if (accumulate) { const float4 o = out[i]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        out[i] = a;
    }
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
    static const int qmax = getenv("XFR_TAIL_QMAX") ? atoi(getenv("XFR_TAIL_QMAX")) : 7;
    static const int off = getenv("XFR_NO_TAIL") != nullptr;
    if (!p.tail_ws || !p.tail_cnt || p.tail_force == 1 || off) return 1;
    const int C = num_cus();
    const int q = tiles / C, r = tiles % C;
    if (r == 0 || r > XFR_TAIL_MAX_TILES) return 1;
    const long max_blocks = (long)(p.tail_ws_bytes / tile_bytes);
    if (p.tail_force >= 2) {
        const int S = std::min(p.tail_force, nk);
        return ((long)r * S <= max_blocks) ? S : 1;
    }
    if (q >= qmax || p.K < 1024) return 1;   // many rounds: blocks are dispatched as slots free up, the last round matters little
    // Final-round cost in units of one whole tile: a CU runs ceil(r*S/C) parts of 1/S tile, each with a fixed ramp
    // (ring fill, partial store, arrival) worth ~192 K-rows.  Measured on MI355X: parts shorter
    // than 256 K-rows cost more than they balance, and nothing is gained below K = 1024.
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

template <int TCO, int TM, int BK, int NST, int MODE>
void launch_one(const ConvParams& p, hipStream_t s)
{
    const int n_co = ((p.CoutTot + TCO - 1) / TCO) * p.nhalves;
    const int n_m = (p.M + TM - 1) / TM;
    const size_t lds = (size_t)NST * BK * (TCO + TM) * sizeof(float) + (MODE == MODE_GEN ? 512 * sizeof(int2) : 0);
    ConvParams q = p;
    int grid = n_co * n_m * p.ksplit;
    q.tail_q = 0;
    q.tail_s = 1;
    if (p.ksplit == 1) {
        const int S = pick_tail_split(p, n_co * n_m, (p.K + BK - 1) / BK, (size_t)TCO * TM * sizeof(float));
        if (S > 1) {
            const int r = (n_co * n_m) % num_cus();
            q.tail_q = n_co * n_m - r;
            q.tail_s = S;
            grid = q.tail_q + r * S;
        }
    }
    if constexpr (TCO == 64 && TM == 64) {
        if (q.chain.n > 0) {      // fused micro-program: backward launches only (no relu_in, one half)
            ew_plan_loads(q.chain, q.out0, q.chain_ld);
            hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, true>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
            return;
        }
    }
    if (p.relu_in)
        hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, true, false>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<TCO, TM, BK, NST, MODE, false, false>), dim3(grid), dim3(NT), lds, s, q, n_co, n_m);
}

template <int TCO, int TM, int BK, int NST>
void launch_cfg(const ConvParams& p, hipStream_t s)
{
    const bool vec = (p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && (p.M % 4) == 0 && p.OH == p.H && p.OW == p.W);
    if (vec) launch_one<TCO, TM, BK, NST, MODE_VEC>(p, s);
    else if (p.tap_major && (p.Cin % BK) == 0) launch_one<TCO, TM, BK, NST, MODE_TAP>(p, s);
    else if (p.tap_major) launch_one<TCO, TM, 16, 4, MODE_TAP>(p, s);
    else launch_one<TCO, TM, 16, 4, MODE_GEN>(p, s);
}

}  // namespace

int conv_gemm_pick_cfg(const ConvParams& p)
{
    // Measured on MI355X over the ResNet-101 / ResNet-50 / Light-CNN GEMM shapes (M = 1.5k..400k, K = 64..4608,
    // Cout = 64..2048): the 64x64 tile wins or ties everywhere -- these grids are small (1-12 workgroups per CU), so
    // finer tiles balance the 256 CUs better and keep more waves per SIMD than 128-wide tiles buy in reuse.  Deep-K
    // launches of at most two tiles per CU (layer 3/4 of a 32-image batch) prefer 32-deep K-steps: half the barriers
    // per MFMA.  Their 48 KB ring allows three workgroups per CU, so larger grids (the W / relu(W) dual launch of the
    // same layer has twice the tiles) stay on the 24 KB ring where whole tiles and tail parts are all resident.
    const long tiles = (long)((p.CoutTot + 63) / 64) * p.nhalves * ((p.M + 63) / 64);
    if (p.K >= 1024 && tiles <= 512 && (p.tap_major ? (p.Cin % 32 == 0) : true)) return 5;
    return 4;
}

void launch_cfg_switch(const ConvParams& p, int cfg, hipStream_t s);

int conv_gemm_pick_ksplit(const ConvParams& p, int cfg)
{
    // Few-tile grids (layer 3/4 of a 32-image batch) leave CUs idle in the last wave of workgroups: splitting K puts
    // P x as many, P x shorter workgroups on the chip.  P minimises the makespan ceil(tiles*P / (256 CUs)) / P; the
    // partial sums go through a scratch slab and one small reduce kernel (deterministic, no atomics).
    if (p.nhalves != 1 || p.out_stride != 1 || p.chain.n > 0 || !p.splitk_ws || p.out_nb != p.NB || (p.M & 3)) return 1;
    const int tile = 64;
    (void)cfg;
    const long tiles = (long)((p.CoutTot + tile - 1) / tile) * ((p.M + tile - 1) / tile);
    const int bk = (cfg >= 5) ? 32 : 16;
    const int nk = (p.K + bk - 1) / bk;
    if (tiles >= 1536) return 1;
    int best = 1;
    double best_cost = 1e30;
    const int cand[6] = {1, 2, 3, 4, 6, 8};
    for (int ci = 0; ci < 6; ++ci) {
        const int P = cand[ci];
        if (P > 1 && nk / P < 12) break;
        if ((size_t)P * p.CoutTot * p.M * sizeof(float) > p.splitk_ws_bytes) break;
        const double rounds = (double)((tiles * P + 255) / 256) / P;              // in units of one full tile
        const double cost = rounds * (1.0 + 0.04 * (P - 1)) + (P > 1 ? 0.08 : 0.0);  // pipeline refill + reduce pass
        if (cost < best_cost - 1e-9) { best_cost = cost; best = P; }
    }
    return best;
}

void launch_conv_gemm(const ConvParams& p, hipStream_t s)
{
    int cfg = p.force_cfg > 0 ? p.force_cfg : conv_gemm_pick_cfg(p);
    static const int remap4 = getenv("XFR_CFG4") ? atoi(getenv("XFR_CFG4")) : 0;   // experiment: replace the default 64x64x16 ring depth
    if (p.force_cfg == 0 && cfg == 4 && remap4 > 0) cfg = remap4;
    static const int remap5 = getenv("XFR_CFG5") ? atoi(getenv("XFR_CFG5")) : 0;
    if (p.force_cfg == 0 && cfg == 5 && remap5 > 0) cfg = remap5;
    if (p.chain.n > 0 && cfg != 4 && cfg != 5 && cfg != 9 && cfg != 10 && cfg != 11 && cfg != 12) cfg = 4;   // the chain epilogue exists for 64x64 tiles
    ConvParams q = p;
    q.ksplit = p.ksplit > 0 ? p.ksplit : conv_gemm_pick_ksplit(p, cfg);
    {   // a requested split is honoured only where the slab layout and the float4 reduce apply
        const int bk = (cfg >= 5) ? 32 : 16;
        const int nk = (p.K + bk - 1) / bk;
        const bool can = p.nhalves == 1 && p.out_stride == 1 && p.chain.n == 0 && p.splitk_ws && p.out_nb == p.NB &&
                         (p.M & 3) == 0 && (size_t)q.ksplit * p.CoutTot * p.M * sizeof(float) <= p.splitk_ws_bytes;
        if (q.ksplit < 1 || !can) q.ksplit = 1;
        if (q.ksplit > nk) q.ksplit = nk;
    }
    launch_cfg_switch(q, cfg, s);
    if (q.ksplit > 1) {
        const long M4 = q.M / 4;
        long blocks = ((long)q.CoutTot * M4 + NT - 1) / NT;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(NT), 0, s, reinterpret_cast<const float4*>(q.splitk_ws),
                           reinterpret_cast<float4*>(q.out0), q.bias, q.ksplit, q.CoutTot, M4, q.accumulate);
    }
}

void launch_cfg_switch(const ConvParams& p, int cfg, hipStream_t s)
{
    switch (cfg) {
        case 1: launch_cfg<128, 128, 16, 3>(p, s); break;
        case 2: launch_cfg<64, 128, 16, 4>(p, s); break;
        case 3: launch_cfg<128, 64, 16, 4>(p, s); break;
        case 4: launch_cfg<64, 64, 16, 3>(p, s); break;     // 24 KB of LDS per workgroup: 6 resident per CU, room for
                                                            // a second stream's workgroups (measured -2.7 % per step vs 4 stages)
        case 5: launch_cfg<64, 64, 32, 3>(p, s); break;
        case 6: launch_cfg<64, 128, 32, 3>(p, s); break;
        case 7: launch_cfg<128, 64, 32, 3>(p, s); break;
        case 9: launch_cfg<64, 64, 16, 4>(p, s); break;
        case 10: launch_cfg<64, 64, 16, 6>(p, s); break;
        case 11: launch_cfg<64, 64, 16, 2>(p, s); break;
        case 12: launch_cfg<64, 64, 32, 2>(p, s); break;
        default: launch_cfg<64, 64, 16, 3>(p, s); break;
    }
}
