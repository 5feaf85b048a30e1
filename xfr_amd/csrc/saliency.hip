// saliency.hip -- the tail of the hot path: channel pooling, per-sample normalisation, contrastive and truncated
// contrastive combination (whitebox.py:499, :524-526, :547-557) and _mwp_to_saliency (whitebox.py:448-460,
// ebp_ver 6 branch).  Wavefront (64-lane) shuffle reductions; fp64 accumulation for the normalisation sums.
#include "common.h"
#include <math.h>

namespace {

constexpr int NT = 256;

__device__ inline double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// pooled[sb][hw] = sum_c P[c][sb][hw]   (np.sum(axis=1), whitebox.py:499)
__global__ __launch_bounds__(NT) void channel_pool_kernel(const float* __restrict__ P, float* __restrict__ pooled, int C,
                                                         long per_c)
{
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < per_c; i += (long)gridDim.x * NT) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += P[(long)c * per_c + i];
        pooled[i] = acc;
    }
}

// sums[sb] = sum_{c,hw} P[c][sb][hw]   (torch.sum(P[-2]), whitebox.py:524).  grid = (C, SB): a block walks one (channel, row) run of
// HW floats (float4 when HW % 4 == 0), fp64 accumulation, one atomic per block
__global__ __launch_bounds__(NT) void sample_sums_kernel(const float* __restrict__ P, double* __restrict__ sums, int C,
                                                        int SB, int HW)
{
    const int c = blockIdx.x, sb = blockIdx.y;
    const float* __restrict__ row = P + ((size_t)c * SB + sb) * HW;
    double acc = 0.0;
    if ((HW & 3) == 0) {
        const float4* __restrict__ r4 = reinterpret_cast<const float4*>(row);
        for (int i = threadIdx.x; i < (HW >> 2); i += NT) {
            const float4 v = r4[i];
            acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += NT) acc += (double)row[i];
    }
    acc = wave_sum(acc);
    __shared__ double part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[sb], (part[0] + part[1]) + (part[2] + part[3]));
}

// out[n][hw] = sum_c relu(keep*m - keep*q),  m = P[c][n][hw]/S_n, q = P[c][N+n][hw]/S_{N+n},
// keep = (m >= thr[n]) (truncated, whitebox.py:553-556) or 1 (contrastive, :526)
__global__ __launch_bounds__(NT) void contrast_kernel(const float* __restrict__ P, const double* __restrict__ sums,
                                                     const float* __restrict__ thr, float* __restrict__ out, int C, int N,
                                                     int HW)
{
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int n = (int)(i / HW);
        const int hw = (int)(i - (long)n * HW);
        const float sm = (float)sums[n], sq = (float)sums[N + n];
        const float t = thr ? thr[n] : 0.f;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float m = __fdiv_rn(P[((long)c * 2 * N + n) * HW + hw], sm);
            const float q = __fdiv_rn(P[((long)c * 2 * N + N + n) * HW + hw], sq);
            float d = m - q;
            if (thr && !(m >= t)) d = 0.f;     // mask*m - mask*q with mask in {0,1}
            acc += fmaxf(d, 0.f);
        }
        out[i] = acc;
    }
}

// ---- truncation threshold: radix select on the bit pattern of the non-negative floats m = P/S ---------------
// whitebox.py:550-553: ascending sort, cumulative sum, keep where cumsum >= (pct/100)*cumsum[-1].  Because m >= 0 the
// kept set is a suffix of the sorted order; it is found without sorting by descending 8 bits at a time: per pass a
// 256-bin histogram of value sums over the elements matching the prefix fixed so far, then the first bin whose
// running total reaches the target is entered.  After 4 passes the prefix is the exact threshold value v*.
struct TruncState {       // one per sample
    double below;         // sum of all elements strictly below the current prefix range
    double target;        // (pct/100) * total
    unsigned prefix;      // fixed high bits
    int pass;             // number of passes done
    double hist[256];
    unsigned cnt[256];
};

// One pass of the radix select: the histogram (value sums in fp64, counts) of the digit `pass` over the values whose higher digits match the
// prefix.  A block walks whole channel rows of its map with float4 loads; a thread keeps the running (digit, sum, count) of the run of equal
// digits it is in and only touches the LDS histogram when the digit changes -- neighbouring pixels share their leading digit almost always,
// and in the first pass all 256 threads of a block would otherwise queue on a handful of bins.
__global__ __launch_bounds__(NT) void trunc_hist_kernel(const float* __restrict__ P, const double* __restrict__ sums,
                                                       TruncState* __restrict__ st, int C, int N, int HW, int pass)
{
    const int n = blockIdx.y;
    __shared__ double h[256];
    __shared__ unsigned cn[256];
    h[threadIdx.x] = 0.0;
    cn[threadIdx.x] = 0u;
    __syncthreads();
    const float s = (float)sums[n];
    const unsigned prefix = st[n].prefix;
    const int shift = 24 - 8 * pass;
    int cur = -1;
    double acc = 0.0;
    unsigned cnt = 0u;
    auto take = [&](float p) {
        const float m = __fdiv_rn(p, s);
        const unsigned bits = __float_as_uint(m);
        const bool match = (pass == 0) || ((bits >> (shift + 8)) == (prefix >> (shift + 8)));
        if (!match) return;
        const int d = (int)((bits >> shift) & 255u);
        if (d != cur) {
            if (cnt) { atomicAdd(&h[cur], acc); atomicAdd(&cn[cur], cnt); }
            cur = d; acc = 0.0; cnt = 0u;
        }
        acc += (double)m;
        cnt += 1u;
    };
    const bool vec = (HW & 3) == 0 && (((uintptr_t)P) & 15) == 0;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const float* __restrict__ row = P + ((long)c * 2 * N + n) * HW;
        if (vec) {
            const float4* __restrict__ row4 = reinterpret_cast<const float4*>(row);
            for (int i = threadIdx.x; i < (HW >> 2); i += NT) {
                const float4 v = row4[i];
                take(v.x); take(v.y); take(v.z); take(v.w);
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += NT) take(row[i]);
        }
    }
    if (cnt) { atomicAdd(&h[cur], acc); atomicAdd(&cn[cur], cnt); }
    __syncthreads();
    if (cn[threadIdx.x]) {
        atomicAdd(&st[n].hist[threadIdx.x], h[threadIdx.x]);
        atomicAdd(&st[n].cnt[threadIdx.x], cn[threadIdx.x]);
    }
}

__global__ void trunc_select_kernel(TruncState* __restrict__ st, float percentile, float* __restrict__ thr, int pass)
{
    const int n = blockIdx.x;
    if (threadIdx.x != 0) return;
    TruncState& s = st[n];
    if (pass == 0) {
        double tot = 0.0;
        for (int d = 0; d < 256; ++d) tot += s.hist[d];
        s.target = (double)(percentile / 100.0f) * tot;
        s.below = 0.0;
    }
    const int shift = 24 - 8 * pass;
    double run = s.below;
    int pick = -1, last = -1;
    for (int d = 0; d < 256; ++d) {
        if (!s.cnt[d]) continue;
        last = d;
        if (run + s.hist[d] >= s.target) { pick = d; break; }
        run += s.hist[d];
    }
    if (pick < 0) {            // rounding left the total just short of the target: enter the last non-empty bin
        pick = last < 0 ? 0 : last;
        run = s.below;
        for (int d = 0; d < pick; ++d) run += s.hist[d];
    }
    s.below = run;
    s.prefix |= ((unsigned)pick) << shift;
    s.pass = pass + 1;
    for (int d = 0; d < 256; ++d) { s.hist[d] = 0.0; s.cnt[d] = 0u; }
    if (pass == 3) thr[n] = __uint_as_float(s.prefix);
}

__global__ void trunc_init_kernel(TruncState* __restrict__ st)
{
    TruncState& s = st[blockIdx.x];
    for (int d = threadIdx.x; d < 256; d += blockDim.x) { s.hist[d] = 0.0; s.cnt[d] = 0u; }
    if (threadIdx.x == 0) { s.below = 0.0; s.target = 0.0; s.prefix = 0u; s.pass = 0; }
}

// ---- skimage.filters.gaussian(img, 2) == scipy.ndimage.gaussian_filter(sigma=2, mode='nearest', truncate=4) ------
// 17 taps; scipy's correlate1d accumulates in double, symmetric form  tmp = x[0]*w[0] + sum_{j=-8..-1} (x[j]+x[-j])*w[j],
// rounds the line to the array dtype (float32) after each axis; axis 0 first, then axis 1.
struct BlurW { double w[9]; };   // w[0] = centre, w[j] = tap at distance j

__global__ __launch_bounds__(NT) void blur_axis_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H,
                                                      int W, int axis, const BlurW bw)
{
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int n = (int)(i / ((long)H * W));
        const int r = (int)(i - (long)n * H * W);
        const int y = r / W, x = r - y * W;
        const float* img = in + (long)n * H * W;
        double tmp;
        if (axis == 0) {
            tmp = (double)img[y * W + x] * bw.w[0];
            for (int j = 8; j >= 1; --j) {
                int ya = y - j; ya = ya < 0 ? 0 : ya;
                int yb = y + j; yb = yb > H - 1 ? H - 1 : yb;
                tmp += ((double)img[ya * W + x] + (double)img[yb * W + x]) * bw.w[j];
            }
        } else {
            tmp = (double)img[y * W + x] * bw.w[0];
            for (int j = 8; j >= 1; --j) {
                int xa = x - j; xa = xa < 0 ? 0 : xa;
                int xb = x + j; xb = xb > W - 1 ? W - 1 : xb;
                tmp += ((double)img[y * W + xa] + (double)img[y * W + xb]) * bw.w[j];
            }
        }
        out[i] = (float)tmp;
    }
}

// img = max(0, img); img /= max(img.sum(), eps)     (whitebox.py:458-459).  One workgroup per sample.
__global__ __launch_bounds__(NT) void clamp_normalize_kernel(float* __restrict__ img, int HW, float eps)
{
    float* p = img + (long)blockIdx.x * HW;
    __shared__ double part[NT / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < HW; i += NT) acc += (double)fmaxf(p[i], 0.f);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < NT / 64; ++k) tot += part[k];
    const float denom = fmaxf((float)tot, eps);
    for (int i = threadIdx.x; i < HW; i += NT) p[i] = __fdiv_rn(fmaxf(p[i], 0.f), denom);
}

inline int grid_for(long n)
{
    long b = (n + NT - 1) / NT;
    if (b < 1) b = 1;
    if (b > 256L * 32) b = 256L * 32;
    return (int)b;
}

}  // namespace

void launch_channel_pool(const float* P, float* pooled, int C, int SB, int HW, hipStream_t s)
{
    const long per_c = (long)SB * HW;
    hipLaunchKernelGGL(channel_pool_kernel, dim3(grid_for(per_c)), dim3(NT), 0, s, P, pooled, C, per_c);
}

void launch_sample_sums(const float* P, double* sums, int C, int SB, int HW, hipStream_t s)
{
    (void)hipMemsetAsync(sums, 0, sizeof(double) * SB, s);
    hipLaunchKernelGGL(sample_sums_kernel, dim3(C, SB), dim3(NT), 0, s, P, sums, C, SB, HW);
}

void launch_contrast(const float* P, const double* sums, const float* thr, float* out, int C, int N, int HW, hipStream_t s)
{
    hipLaunchKernelGGL(contrast_kernel, dim3(grid_for((long)N * HW)), dim3(NT), 0, s, P, sums, thr, out, C, N, HW);
}

size_t truncation_scratch_bytes(int N) { return sizeof(TruncState) * (size_t)N; }

void launch_truncation_threshold(const float* P, const double* sums, float percentile, float* thr, void* scratch, int C, int N,
                                 int HW, hipStream_t s)
{
    TruncState* st = reinterpret_cast<TruncState*>(scratch);
    hipLaunchKernelGGL(trunc_init_kernel, dim3(N), dim3(NT), 0, s, st);
    const long chunks = std::max(1, std::min(C, 64));      // blocks per map: whole channel rows each
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(trunc_hist_kernel, dim3((int)chunks, N), dim3(NT), 0, s, P, sums, st, C, N, HW, pass);
        hipLaunchKernelGGL(trunc_select_kernel, dim3(N), dim3(64), 0, s, st, percentile, thr, pass);
    }
}

void launch_saliency_blur(const float* in, float* tmp, float* out, int N, int H, int W, float eps, hipStream_t s)
{
    // scipy.ndimage._filters._gaussian_kernel1d(sigma=2, order=0, radius=int(4*2+0.5)=8)
    double phi[17], sum;
    for (int i = 0; i < 17; ++i) { const double x = (double)(i - 8); phi[i] = exp(-0.5 / 4.0 * x * x); }
    {   // numpy pairwise summation of 17 doubles: 8 running partial sums over the first 16, then the tail
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = phi[k];
        for (int k = 0; k < 8; ++k) r[k] += phi[8 + k];
        sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        sum += phi[16];
    }
    BlurW bw;
    for (int j = 0; j <= 8; ++j) bw.w[j] = phi[8 + j] / sum;
    const long total = (long)N * H * W;
    hipLaunchKernelGGL(blur_axis_kernel, dim3(grid_for(total)), dim3(NT), 0, s, in, tmp, N, H, W, 0, bw);
    hipLaunchKernelGGL(blur_axis_kernel, dim3(grid_for(total)), dim3(NT), 0, s, tmp, out, N, H, W, 1, bw);
    hipLaunchKernelGGL(clamp_normalize_kernel, dim3(N), dim3(NT), 0, s, out, H * W, eps);
}
