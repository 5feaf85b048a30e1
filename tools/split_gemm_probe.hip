// split_gemm_probe.hip -- EXPERIMENT (not part of libxfr_amd.so): how fast is an fp32-accurate GEMM on the bf16 MFMA pipe of gfx950?
//
// The product's GEMMs run v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak, 1/16 of the bf16 rate) at 0.72-0.8 of that peak: the only lever left past it is
// to feed the bf16 pipe with SPLIT operands.  An fp32 value is exactly the sum of three bf16 pieces (8 + 8 + 8 significant bits; the "deep" variant and the
// host-side weight split round every piece to nearest like the product kernel, conv_gemm_split.hip K17; the first-cut and direct-B variants still truncate, which
// leaves one-signed remainders), a bf16 x bf16 product is exact in fp32, and the six products with i + j <= 2 reproduce the fp32 product to ~2^-23
// ("bf16x6": tests/precision/split_probe.py shows the maps cannot tell it from a re-ordered fp32 sum, while the three-product "bf16x3" moves the
// contrastive maps by 2e-2).  Six v_mfma_f32_32x32x16_bf16 do the work of sixteen fp32 MFMAs in 6 x 32 instead of 8 x 64 cycles: a 2.67x ceiling.
//
// This probe measures how much of that survives the splitting work, for C[Cout][M] = W[Cout][K] * X[K][M] (a 1x1 stride-1 convolution in the
// engine's CNHW layout: X rows are channels):
//   * W is split ONCE on the host (weights are static) into three bf16 planes, pre-tiled so that a K-step of a row tile is one contiguous block that
//     goes global -> LDS without touching registers (buffer_load ... lds);
//   * X is fp32 in HBM (what the epilogues of the producing launches write); a workgroup loads a 16 x 128 slab into registers, splits every value
//     once (4 VALU + 1.5 pack instructions per element), and writes three bf16 planes to LDS in fragment order;
//   * each wave holds TR x TC 32x32 accumulator tiles and issues the six products per tile pair, smallest terms first.
// Usage: split_gemm_probe [reps]   (prints one line per shape / tile configuration: us, TFLOP/s-equivalent, max error against a double reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
#include <type_traits>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BK = 16;          // one bf16 MFMA's K
constexpr int TN = 128;         // activation columns per workgroup (256 lanes = 128 columns x 2 k-halves of 8)

struct Split3 { unsigned h0, h1, h2; };      // the three pieces, each in the upper half of a word

typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)      // v_cvt_pk_bf16_f32: two round-to-nearest-even conversions, lo in the low half
{
    const v2f_t t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, v2bf_t));
}
// Round-to-nearest split of a PAIR (even k in the low half): every piece is at most half a bf16 ulp of the one before, so the three dropped products of
// order 3 are ~2^-25 of the product and of either sign.  (A truncating split -- one AND per piece -- leaves remainders of up to a whole ulp, all of the
// value's sign: dropped terms ~2^-21 that add up along K instead of averaging out; measured: maps 4x further from the reference than the fp32 kernels.)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    p0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);       // exact
    p1 = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);     // exact, <= 8 significant bits
    p2 = cvt_pk_bf16(sa, sb);
}

__device__ __forceinline__ Split3 split3(float v)
{
    Split3 s;
    s.h0 = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(s.h0);
    s.h1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(s.h1);
    s.h2 = __float_as_uint(r2);             // <= 8 significant bits left: the lower half is zero
    return s;
}

__device__ __forceinline__ unsigned pack_hi(unsigned odd, unsigned even)     // (odd & 0xffff0000) | (even >> 16)
{
    return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}

template <int TERMS> struct Terms;
// (piece of W, piece of X), smallest first
template <> struct Terms<6> { static constexpr int n = 6; static constexpr int a[6] = {2, 1, 0, 1, 0, 0}; static constexpr int b[6] = {0, 1, 2, 0, 1, 0}; };
template <> struct Terms<3> { static constexpr int n = 3; static constexpr int a[3] = {1, 0, 0}; static constexpr int b[3] = {0, 1, 0}; };
template <> struct Terms<1> { static constexpr int n = 1; static constexpr int a[1] = {0}; static constexpr int b[1] = {0}; };

// WR x WC waves, each TR x TC accumulator tiles: workgroup tile = (WR*TR*32) rows of W  x  (WC*TC*32 = 128) columns of X
template <int WR, int WC, int TR, int TC, int TERMS, int NSPLIT, int PIPE>
__global__ __launch_bounds__(WR * WC * 64) void split_gemm_kernel(const uint16_t* __restrict__ Wt, const float* __restrict__ X, float* __restrict__ C,
                                                                   int Cout, int K, int M)
{
    constexpr int TM = WR * TR * 32;
    static_assert(WC * TC * 32 == TN, "column tile is 128");
    static_assert(WR * WC == 4, "four waves");
    constexpr int A_BYTES = 3 * 2 * TM * 16, B_BYTES = 3 * 2 * TN * 16;
    constexpr int WST = PIPE ? 3 : 2;              // LDS stages of W; X has two
    constexpr int XBASE = WST * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int nrt = Cout / TM;
    const int rt = blockIdx.x % nrt, ct = blockIdx.x / nrt;        // row tiles fastest: the workgroups sharing an X slab run together
    const int m0 = ct * TN;
    const int nk = K / BK;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, nrt * nk * A_BYTES, 0x00020000);

    // staging role: column m0 + (tid & 127), k-half tid >> 7
    const int sm = tid & 127, skh = tid >> 7;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, K * M * 4, 0x00020000);
    const unsigned xoff = m0 + sm < M ? (unsigned)((skh * 8 * M + m0 + sm) * 4) : 0x80000000u;      // out of range: the hardware returns 0

    v16f acc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    auto load_w = [&](int kt, int stage) {
        constexpr int NB = A_BYTES / 1024;            // 1 KB per wave instruction
#pragma unroll
        for (int b = wave; b < NB; b += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(lds + stage * A_BYTES + b * 1024), 16, b * 1024 + lane * 16, (rt * nk + kt) * A_BYTES, 0, 0);
    };
    auto load_x = [&](int kt, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, xoff, (kt * BK + i) * M * 4, 0));
    };
    auto store_x = [&](const float (&v)[8], int stage) {
        unsigned char* base = lds + XBASE + stage * B_BYTES + (skh * TN + sm) * 16;
        if (NSPLIT == 3) {
            Split3 s[8];
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = split3(v[i]);
            uint4 p0, p1, p2;
            p0.x = pack_hi(s[1].h0, s[0].h0); p0.y = pack_hi(s[3].h0, s[2].h0); p0.z = pack_hi(s[5].h0, s[4].h0); p0.w = pack_hi(s[7].h0, s[6].h0);
            p1.x = pack_hi(s[1].h1, s[0].h1); p1.y = pack_hi(s[3].h1, s[2].h1); p1.z = pack_hi(s[5].h1, s[4].h1); p1.w = pack_hi(s[7].h1, s[6].h1);
            p2.x = pack_hi(s[1].h2, s[0].h2); p2.y = pack_hi(s[3].h2, s[2].h2); p2.z = pack_hi(s[5].h2, s[4].h2); p2.w = pack_hi(s[7].h2, s[6].h2);
            *(uint4*)(base) = p0;
            *(uint4*)(base + 2 * TN * 16) = p1;
            *(uint4*)(base + 4 * TN * 16) = p2;
        } else {
            uint4 p0;
            p0.x = pack_hi(__float_as_uint(v[1]), __float_as_uint(v[0])); p0.y = pack_hi(__float_as_uint(v[3]), __float_as_uint(v[2]));
            p0.z = pack_hi(__float_as_uint(v[5]), __float_as_uint(v[4])); p0.w = pack_hi(__float_as_uint(v[7]), __float_as_uint(v[6]));
            *(uint4*)(base) = p0;
        }
    };

    const int l32 = lane & 31, kh = lane >> 5;
    float xa[8];
    using T = Terms<TERMS>;
    auto frags = [&](const unsigned char* As, const unsigned char* Bs, v8bf (&af)[3][TR], v8bf (&bf)[3][TC]) {
#pragma unroll
        for (int p = 0; p < NSPLIT; p++) {
#pragma unroll
            for (int i = 0; i < TR; i++) af[p][i] = *(const v8bf*)(As + ((p * 2 + kh) * TM + (wr * TR + i) * 32 + l32) * 16);
#pragma unroll
            for (int j = 0; j < TC; j++) bf[p][j] = *(const v8bf*)(Bs + ((p * 2 + kh) * TN + (wc * TC + j) * 32 + l32) * 16);
        }
    };
    auto mfmas = [&](const v8bf (&af)[3][TR], const v8bf (&bf)[3][TC], int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; t++)
#pragma unroll
            for (int i = 0; i < TR; i++)
#pragma unroll
                for (int j = 0; j < TC; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[T::a[t] < NSPLIT ? T::a[t] : 0][i], bf[T::b[t] < NSPLIT ? T::b[t] : 0][j], acc[i][j], 0, 0, 0);
    };

    if (PIPE == 0) {
        // Step kt: (1) split the X slab of step kt+1 (loaded during step kt-1) into stage nxt, (2) issue the loads of W(kt+1) -> stage nxt and X(kt+2) ->
        // registers, (3) fragments + MFMAs of stage cur, (4) counted wait for W(kt+1), one raw barrier.  The split comes BEFORE the loads of the step because
        // the compiler orders a ds_write after every outstanding buffer_load...lds (it cannot tell the LDS-DMA's destination from the ds_write's): vmcnt(0).
        load_w(0, 0);
        load_x(0, xa);
        store_x(xa, 0);
        if (nk > 1) load_x(1, xa);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        if (nk <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; kt++) {
            const int cur = kt & 1, nxt = cur ^ 1;
            if (kt + 1 < nk) {
                store_x(xa, nxt);
                load_w(kt + 1, nxt);
            }
            if (kt + 2 < nk) load_x(kt + 2, xa);
            v8bf af[3][TR], bf[3][TC];
            frags(lds + cur * A_BYTES, lds + XBASE + cur * B_BYTES, af, bf);
            mfmas(af, bf, 0, T::n);
            __builtin_amdgcn_sched_barrier(0);                                 // keep the MFMAs above the wait
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     // W(kt+1) has landed; only the 8 X loads of step kt+2 may still be in flight
            if (kt + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // raw: __syncthreads() would also drain the X loads
        }
    } else {
        // Three W stages: the loads of step kt+2 (W -> LDS, X -> registers) are issued in the MIDDLE of step kt, right after the split of X(kt+1) has been
        // written -- so the compiler's vmcnt(0) before that ds_write waits exactly for what the step needs anyway (W(kt+1), X(kt+1), issued a whole step
        // earlier), and the split's VALU work sits between two halves of the step's MFMAs.
        load_w(0, 0);
        load_x(0, xa);
        store_x(xa, 0);
        if (nk > 1) { load_w(1, 1); load_x(1, xa); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + (A_BYTES / 1024 + 3) / 4) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int ws = 0;
        for (int kt = 0; kt < nk; kt++) {
            const int cur = kt & 1, nxt = cur ^ 1;
            v8bf af[3][TR], bf[3][TC];
            frags(lds + ws * A_BYTES, lds + XBASE + cur * B_BYTES, af, bf);
            mfmas(af, bf, 0, T::n / 2);
            if (kt + 1 < nk) store_x(xa, nxt);
            if (kt + 2 < nk) {
                const int w2 = ws == 0 ? 2 : ws - 1;        // (kt + 2) % 3
                load_w(kt + 2, w2);
                load_x(kt + 2, xa);
            }
            mfmas(af, bf, T::n / 2, T::n);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (kt + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // last steps: nothing newer hides W(kt+1)
            __builtin_amdgcn_s_barrier();
            ws = ws == 2 ? 0 : ws + 1;
        }
    }

    // C/D layout of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++) {
            const int col = m0 + (wc * TC + j) * 32 + l32;
            const int row0 = rt * TM + (wr * TR + i) * 32 + 4 * kh;
            if (col < M)
#pragma unroll
                for (int r = 0; r < 16; r++) C[(size_t)(row0 + (r & 3) + 8 * (r >> 2)) * M + col] = acc[i][j][r];
        }
}

// Third variant ("deep"): what the first two leave on the table is LATENCY, not VALU work (the 6-MFMA launch without any split is only 5 % faster).
//   * the X slab of step kt+3 is loaded while step kt runs (three register sets, the loop unrolled by three so that no copy touches a register in flight),
//     W two steps ahead through three LDS stages;
//   * the split's ds_writes are inline asm: a compiler-visible ds_write after a buffer_load...lds is ordered with s_waitcnt vmcnt(0), which would drain
//     exactly that prefetch;
//   * no branch in a step (loads past the end are out of range of their buffer and return zeros), so the step is one basic block and
//     sched_group_barrier can spread the split's VALU instructions between the MFMAs (ILV; measured: no difference).
// Measured and removed: reading the fragments of step kt+1 during step kt (a third fragment set, 197 registers): slower on every shape.
template <int TERMS, int NSPLIT, int ILV>
__global__ __launch_bounds__(256) void split_gemm_deep_kernel(const uint16_t* __restrict__ Wt, const float* __restrict__ X, float* __restrict__ C, int Cout, int K, int M)
{
    constexpr int TR = 2, TC = 2, TM = 128;
    constexpr int A_BYTES = 3 * 2 * TM * 16, B_BYTES = 3 * 2 * TN * 16, XBASE = 3 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nrt = Cout / TM;
    const int rt = blockIdx.x % nrt, ct = blockIdx.x / nrt;
    const int m0 = ct * TN;
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, nrt * nk * A_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, K * M * 4, 0x00020000);
    const int sm = tid & 127, skh = tid >> 7;
    const unsigned xoff = m0 + sm < M ? (unsigned)((skh * 8 * M + m0 + sm) * 4) : 0x80000000u;
    const unsigned xs_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds) + XBASE + (skh * TN + sm) * 16;   // LDS byte address of this lane's slot
    const int l32 = lane & 31, kh = lane >> 5;

    v16f acc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    auto load_w = [&](int kt, int stage) {
#pragma unroll
        for (int b = 0; b < 3; b++)      // steps past the end: out of range, the hardware writes zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(lds + stage * A_BYTES + (b * 4 + wave) * 1024), 16,
                                                     ((b * 4 + wave) * 1024 + lane * 16) | (kt < nk ? 0u : 0x80000000u), (rt * nk + kt) * A_BYTES, 0, 0);
    };
    auto load_x = [&](int kt, float (&v)[8]) {
        const unsigned dead = kt < nk ? 0u : 0x80000000u;     // steps past the end: out of range, zeros
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, xoff | dead, (kt * BK + i) * M * 4, 0));
    };
    auto write_piece = [&](uint4 p, int stage, int piece) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u q = {p.x, p.y, p.z, p.w};
        asm volatile("ds_write_b128 %0, %1" :: "v"(xs_addr + stage * B_BYTES + piece * 2 * TN * 16), "v"(q));
    };
    auto store_x = [&](const float (&v)[8], int stage) {
        if (NSPLIT == 3) {
            uint4 p0, p1, p2;
            split_pair(v[0], v[1], p0.x, p1.x, p2.x);
            split_pair(v[2], v[3], p0.y, p1.y, p2.y);
            split_pair(v[4], v[5], p0.z, p1.z, p2.z);
            split_pair(v[6], v[7], p0.w, p1.w, p2.w);
            write_piece(p0, stage, 0); write_piece(p1, stage, 1); write_piece(p2, stage, 2);
        } else {
            uint4 p0;
            p0.x = pack_hi(__float_as_uint(v[1]), __float_as_uint(v[0])); p0.y = pack_hi(__float_as_uint(v[3]), __float_as_uint(v[2]));
            p0.z = pack_hi(__float_as_uint(v[5]), __float_as_uint(v[4])); p0.w = pack_hi(__float_as_uint(v[7]), __float_as_uint(v[6]));
            write_piece(p0, stage, 0);
        }
    };
    using T = Terms<TERMS>;
    struct Frags { v8bf a[3][TR], b[3][TC]; };
    auto read_frags = [&](int stage, Frags& f) {
        const unsigned char* As = lds + stage * A_BYTES;
        const unsigned char* Bs = lds + XBASE + stage * B_BYTES;
#pragma unroll
        for (int p = 0; p < NSPLIT; p++) {
#pragma unroll
            for (int i = 0; i < TR; i++) f.a[p][i] = *(const v8bf*)(As + ((p * 2 + kh) * TM + (wr * TR + i) * 32 + l32) * 16);
#pragma unroll
            for (int j = 0; j < TC; j++) f.b[p][j] = *(const v8bf*)(Bs + ((p * 2 + kh) * TN + (wc * TC + j) * 32 + l32) * 16);
        }
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int t = 0; t < T::n; t++)
#pragma unroll
            for (int i = 0; i < TR; i++)
#pragma unroll
                for (int j = 0; j < TC; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[T::a[t] < NSPLIT ? T::a[t] : 0][i], f.b[T::b[t] < NSPLIT ? T::b[t] : 0][j], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&]() {
        if (ILV) {
#pragma unroll
            for (int q = 0; q < T::n * TR * TC; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, ILV, 0);    // ILV VALU instructions of the split
            }
        }
    };
    std::integral_constant<int, 0> S0; std::integral_constant<int, 1> S1; std::integral_constant<int, 2> S2;
    float x0[8], x1[8], x2[8];
    {
        // one K-step; ST = kt % 3 (static), xcur = X(kt+1) registers (split here), xnew = registers that receive X(kt+3)
        auto step = [&](int kt, auto ST, float (&xcur)[8], float (&xnew)[8]) {
            constexpr int st = decltype(ST)::value, st1 = (st + 1) % 3, st2 = (st + 2) % 3;
            load_w(kt + 2, st2);
            load_x(kt + 3, xnew);
            Frags f;
            read_frags(st, f);
            mfmas(f);
            store_x(xcur, st1);
            interleave();
            // W(kt+1) (issued during step kt-1) has landed when at most X(kt+2), W(kt+2), X(kt+3) are outstanding: 8 + 3 + 8
            asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        load_w(0, 0);
        load_x(0, x0);
        store_x(x0, 0);
        load_w(1, 1);
        load_x(1, x0);
        load_x(2, x1);
        asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // W(0) and the split of X(0) are in LDS
        for (int kt = 0; kt < nk; kt += 3) {        // up to two steps past the end multiply zeros: no branch inside the body
            step(kt, S0, x0, x2);
            step(kt + 1, S1, x1, x0);
            step(kt + 2, S2, x2, x1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++) {
            const int col = m0 + (wc * TC + j) * 32 + l32;
            const int row0 = rt * TM + (wr * TR + i) * 32 + 4 * kh;
            if (col < M)
#pragma unroll
                for (int r = 0; r < 16; r++) C[(size_t)(row0 + (r & 3) + 8 * (r >> 2)) * M + col] = acc[i][j][r];
        }
}

// Fourth variant ("direct B"): X never touches LDS.  The B operand of v_mfma_f32_32x32x16_bf16 wants, per lane, 8 consecutive k of ONE column -- which is
// what a lane gets when it loads its own column from 8 rows of X (lanes 0-31 / 32-63 read two 128-byte row segments per instruction).  Each wave loads and
// splits the 16 x 64 slab of its own column half (the two waves of a column half do it twice: +VALU, -LDS: no ds_write, half the LDS reads, and the MFMAs of
// a step depend on no barrier but the one that publishes W).  The split of step kt+1 runs during the MFMAs of step kt.
template <int TERMS, int NSPLIT>
__global__ __launch_bounds__(256) void split_gemm_directb_kernel(const uint16_t* __restrict__ Wt, const float* __restrict__ X, float* __restrict__ C, int Cout, int K, int M)
{
    constexpr int TR = 2, TC = 2, TM = 128;
    constexpr int A_BYTES = 3 * 2 * TM * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nrt = Cout / TM;
    const int rt = blockIdx.x % nrt, ct = blockIdx.x / nrt;
    const int m0 = ct * TN;
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, nrt * nk * A_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, K * M * 4, 0x00020000);
    const int l32 = lane & 31, kh = lane >> 5;
    unsigned xoff[TC];
#pragma unroll
    for (int j = 0; j < TC; j++) {
        const int col = m0 + (wc * TC + j) * 32 + l32;
        xoff[j] = col < M ? (unsigned)((kh * 8 * M + col) * 4) : 0x80000000u;
    }
    v16f acc[TR][TC];
#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    struct XRegs { float v[TC][8]; };
    struct BFrags { v8bf b[3][TC]; };
    auto load_w = [&](int kt, int stage) {
#pragma unroll
        for (int b = 0; b < 3; b++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(lds + stage * A_BYTES + (b * 4 + wave) * 1024), 16,
                                                     ((b * 4 + wave) * 1024 + lane * 16) | (kt < nk ? 0u : 0x80000000u), (rt * nk + kt) * A_BYTES, 0, 0);
    };
    auto load_x = [&](int kt, XRegs& x) {
        const unsigned dead = kt < nk ? 0u : 0x80000000u;
#pragma unroll
        for (int j = 0; j < TC; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) x.v[j][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, xoff[j] | dead, (kt * BK + i) * M * 4, 0));
    };
    auto split_x = [&](const XRegs& x, BFrags& f) {
#pragma unroll
        for (int j = 0; j < TC; j++) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            if (NSPLIT == 3) {
                Split3 s[8];
#pragma unroll
                for (int i = 0; i < 8; i++) s[i] = split3(x.v[j][i]);
                const v4u p0 = {pack_hi(s[1].h0, s[0].h0), pack_hi(s[3].h0, s[2].h0), pack_hi(s[5].h0, s[4].h0), pack_hi(s[7].h0, s[6].h0)};
                const v4u p1 = {pack_hi(s[1].h1, s[0].h1), pack_hi(s[3].h1, s[2].h1), pack_hi(s[5].h1, s[4].h1), pack_hi(s[7].h1, s[6].h1)};
                const v4u p2 = {pack_hi(s[1].h2, s[0].h2), pack_hi(s[3].h2, s[2].h2), pack_hi(s[5].h2, s[4].h2), pack_hi(s[7].h2, s[6].h2)};
                f.b[0][j] = __builtin_bit_cast(v8bf, p0); f.b[1][j] = __builtin_bit_cast(v8bf, p1); f.b[2][j] = __builtin_bit_cast(v8bf, p2);
            } else {
                const float* v = x.v[j];
                const v4u p0 = {pack_hi(__float_as_uint(v[1]), __float_as_uint(v[0])), pack_hi(__float_as_uint(v[3]), __float_as_uint(v[2])),
                                pack_hi(__float_as_uint(v[5]), __float_as_uint(v[4])), pack_hi(__float_as_uint(v[7]), __float_as_uint(v[6]))};
                f.b[0][j] = __builtin_bit_cast(v8bf, p0);
            }
        }
    };
    using T = Terms<TERMS>;
    auto step = [&](int kt, auto ST, const XRegs& xsplit, XRegs& xload, const BFrags& bcur, BFrags& bnext) {
        constexpr int st = decltype(ST)::value, st2 = (st + 2) % 3;
        __builtin_amdgcn_sched_barrier(0);
        load_w(kt + 2, st2);
        load_x(kt + 3, xload);
        v8bf af[3][TR];
        const unsigned char* As = lds + st * A_BYTES;
#pragma unroll
        for (int p = 0; p < NSPLIT; p++)
#pragma unroll
            for (int i = 0; i < TR; i++) af[p][i] = *(const v8bf*)(As + ((p * 2 + kh) * TM + (wr * TR + i) * 32 + l32) * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < T::n; t++)
#pragma unroll
            for (int i = 0; i < TR; i++)
#pragma unroll
                for (int j = 0; j < TC; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[T::a[t] < NSPLIT ? T::a[t] : 0][i], bcur.b[T::b[t] < NSPLIT ? T::b[t] : 0][j], acc[i][j], 0, 0, 0);
        split_x(xsplit, bnext);          // X(kt+1), loaded two steps ago
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x8073);      // vmcnt(35) lgkmcnt(0): W(kt+1) has landed; X(kt+2), W(kt+2), X(kt+3) may be in flight (16 + 3 + 16)
        __builtin_amdgcn_s_barrier();
    };
    std::integral_constant<int, 0> S0; std::integral_constant<int, 1> S1; std::integral_constant<int, 2> S2;
    XRegs xa, xb, xc;
    BFrags b0, b1, b2;
    load_w(0, 0);
    load_w(1, 1);
    load_x(0, xa);
    load_x(1, xb);
    load_x(2, xc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x8070);          // vmcnt(32): W(0), W(1), X(0) are there
    split_x(xa, b0);
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; kt += 3) {
        step(kt, S0, xb, xa, b0, b1);
        step(kt + 1, S1, xc, xb, b1, b2);
        step(kt + 2, S2, xa, xc, b2, b0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0)
#pragma unroll
    for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++) {
            const int col = m0 + (wc * TC + j) * 32 + l32;
            const int row0 = rt * TM + (wr * TR + i) * 32 + 4 * kh;
            if (col < M)
#pragma unroll
                for (int r = 0; r < 16; r++) C[(size_t)(row0 + (r & 3) + 8 * (r >> 2)) * M + col] = acc[i][j][r];
        }
}

// host: W[Cout][K] fp32 -> tiles [Cout/TM][K/16][piece][k-half][TM][8] of bf16 pieces (truncation split, exact)
static uint16_t bf16_rne(float v)
{
    uint32_t u; memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split_host(float v, uint16_t out[3])
{
    out[0] = bf16_rne(v);
    const float r1 = v - bf16_f(out[0]);
    out[1] = bf16_rne(r1);
    const float r2 = r1 - bf16_f(out[1]);
    out[2] = bf16_rne(r2);
}

static std::vector<uint16_t> tile_weights(const std::vector<float>& W, int Cout, int K, int TM)
{
    const int nrt = Cout / TM, nk = K / BK;
    std::vector<uint16_t> T((size_t)nrt * nk * 3 * 2 * TM * 8);
    for (int rt = 0; rt < nrt; rt++)
        for (int kt = 0; kt < nk; kt++)
            for (int r = 0; r < TM; r++)
                for (int kk = 0; kk < BK; kk++) {
                    uint16_t p[3];
                    split_host(W[(size_t)(rt * TM + r) * K + kt * BK + kk], p);
                    for (int q = 0; q < 3; q++)
                        T[((((size_t)(rt * nk + kt) * 3 + q) * 2 + kk / 8) * TM + r) * 8 + kk % 8] = p[q];
                }
    return T;
}

struct Shape { const char* what; int Cout, K, M; };

typedef void (*kern_t)(const uint16_t*, const float*, float*, int, int, int);
static void run_kernel(kern_t k, int TM, size_t lds, const char* name, const Shape& s, const std::vector<float>& W, const float* dX, float* dC, const std::vector<float>& X, int reps, bool check);

template <int WR, int WC, int TR, int TC, int TERMS, int NSPLIT, int PIPE>
static void run_cfg(const char* name, const Shape& s, const std::vector<float>& W, const float* dX, float* dC, const std::vector<float>& X, int reps, bool check)
{
    constexpr int TM = WR * TR * 32;
    run_kernel(split_gemm_kernel<WR, WC, TR, TC, TERMS, NSPLIT, PIPE>, TM, (PIPE ? 3 : 2) * (3 * 2 * TM * 16) + 2 * (3 * 2 * TN * 16), name, s, W, dX, dC, X, reps, check);
}

template <int TERMS, int NSPLIT, int ILV>
static void run_deep(const char* name, const Shape& s, const std::vector<float>& W, const float* dX, float* dC, const std::vector<float>& X, int reps, bool check)
{
    run_kernel(split_gemm_deep_kernel<TERMS, NSPLIT, ILV>, 128, 3 * (3 * 2 * 128 * 16) + 3 * (3 * 2 * TN * 16), name, s, W, dX, dC, X, reps, check);
}

template <int TERMS, int NSPLIT>
static void run_directb(const char* name, const Shape& s, const std::vector<float>& W, const float* dX, float* dC, const std::vector<float>& X, int reps, bool check)
{
    run_kernel(split_gemm_directb_kernel<TERMS, NSPLIT>, 128, 3 * (3 * 2 * 128 * 16), name, s, W, dX, dC, X, reps, check);
}

static void run_kernel(kern_t k, int TM, size_t lds, const char* name, const Shape& s, const std::vector<float>& W, const float* dX, float* dC, const std::vector<float>& X, int reps, bool check)
{
    if (s.Cout % TM || s.K % BK) return;
    std::vector<uint16_t> T = tile_weights(W, s.Cout, s.K, TM);
    uint16_t* dW;
    CK(hipMalloc(&dW, T.size() * 2));
    CK(hipMemcpy(dW, T.data(), T.size() * 2, hipMemcpyHostToDevice));
    const int grid = (s.Cout / TM) * ((s.M + TN - 1) / TN);
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, dW, dX, dC, s.Cout, s.K, s.M);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, dW, dX, dC, s.Cout, s.K, s.M);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * s.Cout * s.K * s.M / (us * 1e-6) / 1e12;
    double worst = 0, scale = 0;
    if (check) {
        std::vector<float> Cc((size_t)s.Cout * s.M);
        CK(hipMemcpy(Cc.data(), dC, Cc.size() * 4, hipMemcpyDeviceToHost));
        uint64_t st = 12345;
        for (int n = 0; n < 4000; n++) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const int r = (st >> 33) % s.Cout;
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const int c = (st >> 33) % s.M;
            double ref = 0, mag = 0;
            for (int kk = 0; kk < s.K; kk++) { const double t = (double)W[(size_t)r * s.K + kk] * X[(size_t)kk * s.M + c]; ref += t; mag += fabs(t); }
            worst = fmax(worst, fabs(Cc[(size_t)r * s.M + c] - ref) / mag);     // error relative to the sum of magnitudes (what rounding scales with)
            scale = fmax(scale, mag);
        }
    }
    printf("%-34s %-22s grid %5d  %8.1f us  %7.1f TFLOP/s-eq   max err / sum|terms| %.2e\n", s.what, name, grid, us, tf, worst);
    fflush(stdout);
    CK(hipFree(dW));
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const Shape shapes[] = {
        {"layer3 3x3 as GEMM 256x2304x12544", 256, 2304, 12544},
        {"layer3 1x1 256x1024x12544", 256, 1024, 12544},
        {"layer3 1x1 1024x256x12544", 1024, 256, 12544},
        {"layer2 3x3 as GEMM 128x1152x50176", 128, 1152, 50176},
        {"layer1 1x1 256x64x200704", 256, 64, 200704},
        {"big 1024x1024x50176", 1024, 1024, 50176},
    };
    for (const Shape& s : shapes) {
        std::vector<float> W((size_t)s.Cout * s.K), X((size_t)s.K * s.M);
        uint64_t st = 99;
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
        for (auto& w : W) w = rnd() * 0.05f;
        for (auto& x : X) { float r = rnd(); x = r > 0 ? r * 3.f : 0.f; }       // post-ReLU activations: half zeros
        float *dX, *dC;
        CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dC, (size_t)s.Cout * s.M * 4));
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
        run_cfg<2, 2, 2, 2, 6, 3, 0>("128x128 bf16x6", s, W, dX, dC, X, reps, true);
        run_deep<6, 3, 0>("128x128 bf16x6 deep", s, W, dX, dC, X, reps, true);
        run_directb<6, 3>("128x128 bf16x6 direct B", s, W, dX, dC, X, reps, true);
        run_directb<6, 1>("128x128 6 MFMAs no split direct B", s, W, dX, dC, X, reps, false);
        run_directb<3, 3>("128x128 bf16x3 direct B", s, W, dX, dC, X, reps, true);
        run_directb<1, 1>("128x128 plain bf16 direct B", s, W, dX, dC, X, reps, true);
        run_deep<6, 1, 0>("128x128 6 MFMAs no split deep", s, W, dX, dC, X, reps, false);
        CK(hipFree(dX)); CK(hipFree(dC));
    }
    return 0;
}
