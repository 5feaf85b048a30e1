/* A host that is not Python: plain C99 against include/xfr_amd.h, nothing else.
 *
 * Builds a three-layer program (Conv 3x3 -> BatchNorm -> in-place ReLU -> Linear over the map -> L2 normalise), loads seeded
 * weights, and -- when a HIP device is visible -- encodes two images and runs one contrastive EBP (what Whitebox.contrastive_ebp,
 * python/xfr/models/whitebox.py:506-527, does through its hooks).  Without a device it prints the planner's schedule
 * (xfr_plan_describe needs none) and the engine's loud refusal to run on the CPU.
 *
 *   gcc -std=c99 -Iinclude -Iexamples examples/c_host.c -Lxfr_amd/csrc -lxfr_amd -Wl,-rpath,$PWD/xfr_amd/csrc -lm -o c_host && ./c_host
 *
 * Device memory is allocated with hipMalloc / hipMemcpy resolved from libamdhip64 at run time (dlopen), so that this file
 * needs no HIP headers: the ABI takes raw device pointers, whoever allocated them.
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xfr_amd.h"
#include "c_host_ref.h"      /* what the real reference computes for this very network and these images (tests/golden/make_golden_chost.py) */

#define IMG 16
#define C1 8
#define D 6

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

int main(void)
{
    /* ---- the layer program: tensor 0 = image, op k defines tensor k + 1 */
    xfr_op_desc ops[5];
    memset(ops, 0, sizeof(ops));
    for (int k = 0; k < 5; ++k) { ops[k].in1 = -1; ops[k].out = k + 1; ops[k].stride = 1; ops[k].w_weight = ops[k].w_bias = ops[k].w_mean = ops[k].w_var = -1; }
    ops[0].kind = XFR_OP_CONV;      ops[0].in0 = 0; ops[0].cout = C1; ops[0].kh = ops[0].kw = 3; ops[0].pad = 1; ops[0].w_weight = 0; ops[0].w_bias = 1;
    ops[1].kind = XFR_OP_BATCHNORM; ops[1].in0 = 1; ops[1].fparam = 1e-5f; ops[1].w_weight = 2; ops[1].w_bias = 3; ops[1].w_mean = 4; ops[1].w_var = 5;
    ops[2].kind = XFR_OP_RELU;      ops[2].in0 = 2; ops[2].inplace = 1;
    ops[3].kind = XFR_OP_LINEAR;    ops[3].in0 = 3; ops[3].cout = D; ops[3].kh = ops[3].kw = IMG; ops[3].w_weight = 6; ops[3].w_bias = 7;
    ops[4].kind = XFR_OP_G_NORMALIZE; ops[4].in0 = 4;
    const int n_ops = 5, n_weights = 8, encode_tensor = 5;

    char text[4096];
    if (xfr_plan_describe(ops, n_ops, n_weights, 1, IMG, IMG, 2, XFR_MODE_AFFINEONLY_WITH_PRIOR, encode_tensor, text, sizeof(text), NULL) != XFR_OK) {
        fprintf(stderr, "plan: %s\n", xfr_last_error());
        return 1;
    }
    printf("-- fused schedule (device-free planner)\n%s", text);

    /* ---- seeded parameters, PyTorch layouts */
    unsigned seed = 12345u;
    static float w0[C1 * 1 * 9], b0[C1], g[C1], be[C1], mu[C1], var[C1], w3[D * C1 * IMG * IMG], b3[D];
    for (int i = 0; i < C1 * 9; ++i) w0[i] = frand(&seed);
    for (int i = 0; i < C1; ++i) { b0[i] = 0.1f * frand(&seed); g[i] = 1.0f + 0.4f * frand(&seed); be[i] = 0.2f * frand(&seed); mu[i] = 0.2f * frand(&seed); var[i] = 1.0f + 0.5f * frand(&seed); }
    for (int i = 0; i < D * C1 * IMG * IMG; ++i) w3[i] = 0.05f * frand(&seed);
    for (int i = 0; i < D; ++i) b3[i] = 0.05f * frand(&seed);
    xfr_tensor_view views[8] = {{w0, C1 * 9}, {b0, C1}, {g, C1}, {be, C1}, {mu, C1}, {var, C1}, {w3, (int64_t)D * C1 * IMG * IMG}, {b3, D}};

    xfr_engine* e = NULL;
    xfr_status st = xfr_engine_create(ops, n_ops, n_weights, 1, IMG, IMG, 4, 0, &e);
    if (st != XFR_OK) {
        printf("-- no engine: %s\n", xfr_last_error());       /* "... the xfr_amd engine has no CPU fallback" */
        return st == XFR_HIP_ERROR ? 0 : 1;
    }
    void* hip = dlopen("libamdhip64.so", RTLD_NOW);
    if (!hip) hip = dlopen("/opt/rocm/lib/libamdhip64.so", RTLD_NOW);
    if (!hip) { fprintf(stderr, "libamdhip64: %s\n", dlerror()); return 1; }
    int (*hipMalloc_)(void**, size_t) = (int (*)(void**, size_t))dlsym(hip, "hipMalloc");
    int (*hipMemcpy_)(void*, const void*, size_t, int) = (int (*)(void*, const void*, size_t, int))dlsym(hip, "hipMemcpy");
    int (*hipDeviceSynchronize_)(void) = (int (*)(void))dlsym(hip, "hipDeviceSynchronize");
    if (!hipMalloc_ || !hipMemcpy_ || !hipDeviceSynchronize_) return 1;

    if (xfr_engine_load_weights(e, views, n_weights) != XFR_OK || xfr_engine_set_mode(e, XFR_MODE_AFFINEONLY_WITH_PRIOR, 1e-16f, 0) != XFR_OK) {
        fprintf(stderr, "%s\n", xfr_last_error());
        return 1;
    }
    /* two "gallery" images -> encodings -> the rows of the un-hooked 2-way triplet classifier; one probe -> contrastive EBP */
    static float imgs[3 * IMG * IMG], enc[2 * D], sal[IMG * IMG];
    for (int i = 0; i < 3 * IMG * IMG; ++i) imgs[i] = 4.0f * frand(&seed);
    float *d_img = NULL, *d_enc = NULL, *d_seed = NULL, *d_sal = NULL;
    hipMalloc_((void**)&d_img, sizeof(imgs)); hipMalloc_((void**)&d_enc, sizeof(enc)); hipMalloc_((void**)&d_seed, sizeof(enc)); hipMalloc_((void**)&d_sal, sizeof(sal));
    hipMemcpy_(d_img, imgs, sizeof(imgs), 1 /* host to device */);
    if (xfr_forward(e, d_img, 2, encode_tensor, d_enc, NULL) != XFR_OK) { fprintf(stderr, "%s\n", xfr_last_error()); return 1; }
    hipDeviceSynchronize_();
    hipMemcpy_(enc, d_enc, sizeof(enc), 2 /* device to host */);
    double enc_err = 0.0;
    for (int i = 0; i < 2 * D; ++i) enc_err = fmax(enc_err, fabs((double)enc[i] - (double)c_host_ref_enc[i]));
    for (int i = 0; i < 2 * D; ++i) enc[i] *= 1.0f / 2500.0f;                      /* demo/test_whitebox.py:129 */
    hipMemcpy_(d_seed, enc, sizeof(enc), 1);                                        /* 2 streams x 1 probe x D: one_hot(k) @ W_cls */
    if (xfr_contrastive(e, d_img + 2 * IMG * IMG, 1, encode_tensor, d_seed, -1.0f, d_sal, NULL) != XFR_OK) { fprintf(stderr, "%s\n", xfr_last_error()); return 1; }
    hipDeviceSynchronize_();
    hipMemcpy_(sal, d_sal, sizeof(sal), 2);
    double sum = 0.0; float mx = 0.f; int arg = 0;
    for (int i = 0; i < IMG * IMG; ++i) { sum += sal[i]; if (sal[i] > mx) { mx = sal[i]; arg = i; } }
    printf("-- contrastive EBP saliency map %dx%d: sum %.6f, max %.5f at (%d, %d)\n", IMG, IMG, sum, mx, arg / IMG, arg % IMG);
    /* against the reference's own map of the same triplet: max|d| / max and cosine (the parity tests' criterion, tests/parity_utils.py) */
    double dmax = 0.0, rmax = 0.0, dot = 0.0, na = 0.0, nb = 0.0;
    for (int i = 0; i < IMG * IMG; ++i) {
        const double a = sal[i], b = c_host_ref_map[i];
        dmax = fmax(dmax, fabs(a - b)); rmax = fmax(rmax, fabs(b));
        dot += a * b; na += a * a; nb += b * b;
    }
    const double rel = dmax / rmax, cosine = dot / sqrt(na * nb);
    printf("-- against the reference: encodings max|d| %.2e, map max|d|/max %.3e, cosine %.8f\n", enc_err, rel, cosine);
    xfr_engine_destroy(e);
    return (fabs(sum - 1.0) < 1e-3 && enc_err < 1e-5 && rel <= 1e-3 && cosine >= 0.99999) ? 0 : 1;
}
