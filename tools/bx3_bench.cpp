// Stand-alone correctness + timing harness for csrc/gemm_bx3.hip (the plane GEMM), torch-free: links libeagcn_hip.so through
// its C ABI.   hipcc -O2 -o tools/bx3_bench tools/bx3_bench.cpp -Leagcn_amd/lib -leagcn_hip -Wl,-rpath,'$ORIGIN/../eagcn_amd/lib'
//   ./tools/bx3_bench check            accuracy of NT / TN / pair at a list of shapes against float64 (and the fp32 MFMA kernel)
//   ./tools/bx3_bench time [iters]     timings at the layer shapes of the BASELINE configs next to the fp32 MFMA kernel (gemm3)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/eagcn_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define RC(x) do { int r_ = (x); if (r_) { printf("eagcn error %d (%s) at line %d\n", r_, eagcn_last_error(), __LINE__); exit(3); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand() {          // uniform [-1, 1), full-range mantissas
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
struct Mat {
    int rows, cols, ld;
    std::vector<float> h;
    float* d = nullptr;
    uint16_t* pl = nullptr;
    size_t pstride = 0;
    int cap = 0;                 // row capacity of the plane images
    Mat(int r, int c, int ld_, float scale = 1.0f, int extra_rows = 0) : rows(r), cols(c), ld(ld_), h((size_t)(r + extra_rows) * ld_, 0.f) {
        for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) h[(size_t)i * ld + j] = scale * frand();
        // rows beyond `rows` (capacity) and columns beyond `cols` hold NaN-free garbage that must never reach a result
        for (int i = r; i < r + extra_rows; ++i) for (int j = 0; j < ld; ++j) h[(size_t)i * ld + j] = 1.0e30f;
        CK(hipMalloc(&d, h.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        cap = r + extra_rows;
        pstride = ((eagcn_bx3_plane_elems(cap, ld) + 63) / 64) * 64;
        CK(hipMalloc(&pl, 3 * pstride * 2));
        CK(hipMemset(pl, 0x7F, 3 * pstride * 2));          // (bf16 NaNs wherever the split does not write)
        RC(eagcn_bx3_split(d, r + extra_rows, ld, pl, pstride, cap, 3, nullptr));
    }
    ~Mat() { (void)hipFree(d); (void)hipFree(pl); }
};

static double check_samples(const char* tag, const std::vector<float>& C, int ldc, int M, int N, int K, const Mat& A, const Mat& B, bool tn,
                            int nsamp, const std::vector<float>* ref32 = nullptr) {
    // error relative to sum |a||b| of the entry (the natural scale of a dot product's rounding error)
    double worst = 0.0, worst32 = 0.0;
    const bool all = (double)M * N <= nsamp;
    const int n = all ? M * N : nsamp;
    for (int t = 0; t < n; ++t) {
        int i, j;
        if (all) { i = t / N; j = t % N; }
        else { i = (int)((frand() * 0.5 + 0.5) * M) % M; j = (int)((frand() * 0.5 + 0.5) * N) % N; }
        double s = 0.0, sa = 0.0;
        for (int k = 0; k < K; ++k) {
            const double a = tn ? A.h[(size_t)k * A.ld + i] : A.h[(size_t)i * A.ld + k];
            const double b = tn ? B.h[(size_t)k * B.ld + j] : B.h[(size_t)j * B.ld + k];
            s += a * b; sa += fabs(a * b);
        }
        const double e = fabs((double)C[(size_t)i * ldc + j] - s) / (sa > 0 ? sa : 1.0);
        if (!(e <= worst)) worst = e;            // (NaN-catching comparison)
        if (ref32) { const double e2 = fabs((double)(*ref32)[(size_t)i * ldc + j] - s) / (sa > 0 ? sa : 1.0); if (!(e2 <= worst32)) worst32 = e2; }
    }
    printf("  %-34s max |err| / sum|a||b| = %.2e (%.1f eps)", tag, worst, worst / 5.96e-8);
    if (ref32) printf("   fp32 MFMA kernel: %.2e (%.1f eps)", worst32, worst32 / 5.96e-8);
    printf("\n");
    return worst;
}

static int fails = 0;
static void expect(bool ok, const char* what) { if (!ok) { printf("  FAIL: %s\n", what); ++fails; } }

static void check_nt(int M, int N, int K, int cap_extra) {
    printf("NT  M=%d N=%d K=%d (capacity rows +%d)\n", M, N, K, cap_extra);
    const int lda = (K + 15) / 16 * 16 + 16, ldb = (K + 15) / 16 * 16, ldc = (N + 3) / 4 * 4 + 4;
    Mat A(M, K, lda, 1.0f, cap_extra), B(N, K, ldb, 0.05f);
    float* dC; CK(hipMalloc(&dC, (size_t)(M + cap_extra) * ldc * 4));
    CK(hipMemset(dC, 0xFF, (size_t)(M + cap_extra) * ldc * 4));
    int* dM; CK(hipMalloc(&dM, 4)); CK(hipMemcpy(dM, &M, 4, hipMemcpyHostToDevice));
    // capacity-sized call with the device-side row count is what the model engine issues; the C entry takes static extents, so
    // run it with the exact M (rows beyond are then never touched) ...
    RC(eagcn_gemm_bx3(0, M, N, K, A.pl, A.pstride, lda, A.cap, B.pl, B.pstride, ldb, B.cap, dC, ldc, 1, 0, 3, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> C((size_t)M * ldc);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    // fp32 MFMA reference kernel
    size_t wsb = eagcn_gemm_sk_workspace_bytes();
    void* ws; CK(hipMalloc(&ws, wsb));
    float* dC2; CK(hipMalloc(&dC2, (size_t)M * ldc * 4));
    std::vector<float> C2((size_t)M * ldc);
    RC(eagcn_gemm_f32_sk(0, 1, M, N, K, A.d, lda, B.d, ldb, dC2, ldc, ws, wsb, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
    const double e = check_samples("bx3 NT vs float64", C, ldc, M, N, K, A, B, false, 20000, &C2);
    expect(e < 8 * 5.96e-8, "NT error above 8 eps of sum|a||b|");
    // padding columns of C untouched
    bool pad_ok = true;
    for (int i = 0; i < M && pad_ok; ++i) for (int j = N; j < ldc; ++j) { uint32_t u; memcpy(&u, &C[(size_t)i * ldc + j], 4); if (u != 0xFFFFFFFFu) { pad_ok = false; break; } }
    expect(pad_ok, "NT wrote beyond column N");
    (void)hipFree(dC); (void)hipFree(dC2); (void)hipFree(ws); (void)hipFree(dM);
}

static void check_tn(int M, int N, int K, int splits, int cap_extra) {
    printf("TN  M=%d N=%d K=%d splits=%d (capacity rows +%d)\n", M, N, K, splits, cap_extra);
    const int lda = (M + 15) / 16 * 16, ldb = (N + 15) / 16 * 16 + 16, ldc = (N + 3) / 4 * 4;
    Mat A(K, M, lda, 1.0f, cap_extra), B(K, N, ldb, 1.0f, cap_extra);
    const size_t slab = (size_t)M * ldc;
    float* dC; CK(hipMalloc(&dC, slab * splits * 4));
    CK(hipMemset(dC, 0, slab * splits * 4));          // (chunks beyond bx3_used_splits are not written)
    RC(eagcn_gemm_bx3(1, M, N, K, A.pl, A.pstride, lda, A.cap, B.pl, B.pstride, ldb, B.cap, dC, ldc, splits, slab, 3, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> Cs(slab * splits), C(slab, 0.f);
    CK(hipMemcpy(Cs.data(), dC, Cs.size() * 4, hipMemcpyDeviceToHost));
    for (int z = 0; z < splits; ++z) for (size_t i = 0; i < slab; ++i) if ((int)(i % ldc) < N) C[i] += Cs[z * slab + i];
    const double e = check_samples("bx3 TN vs float64", C, ldc, M, N, K, A, B, true, 20000);
    expect(e < 8 * 5.96e-8, "TN error above 8 eps of sum|a||b|");
    (void)hipFree(dC);
}



template <typename F>
static float time_fn(int iters, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

static void time_layer(const char* name, int T, int FIN, int FP, int iters) {
    // forward P = X.WcatT^T (NT, M=T, N=FP, K=FIN); dX = dP.Wcat^T (NT, M=T, N=FIN, K=FP); dW = X^T.dP (TN, M=FIN, N=FP, K=T)
    Mat X(T, FIN, FIN), WT(FP, FIN, FIN, 0.05f), W(FIN, FP, FP, 0.05f), dP(T, FP, FP);
    float *P, *dX, *dW; 
    int splits = 64;        // slab CAPACITY, as the layer path passes it: the kernel decides how many chunks it uses (bx3.h)
    if (getenv("BX3_SPLITS")) splits = atoi(getenv("BX3_SPLITS"));
    const size_t slab = (size_t)FIN * FP;
    CK(hipMalloc(&P, (size_t)T * FP * 4)); CK(hipMalloc(&dX, (size_t)T * FIN * 4)); CK(hipMalloc(&dW, slab * splits * 4));
    size_t wsb = eagcn_gemm_sk_workspace_bytes();
    void* ws; CK(hipMalloc(&ws, wsb));
    const double f1 = 2.0 * T * FIN * FP;
    float t;
    printf("%s: T=%d F_in=%d Fp=%d\n", name, T, FIN, FP);
    // both kernels behind the entry points (csrc/bx3.h): 128 x 128 tiles (4 compute + 2 loader waves) | 256 x 128 tiles (8 compute waves)
    float tf[2], tx[2], tw[2], tp[2]; int used[2], usedp[2];
    for (int w = 0; w < 2; ++w) {
        const int old = eagcn_set_bx3_wide(w);
        used[w] = eagcn_bx3_used_splits(splits, FIN, FP, T); usedp[w] = eagcn_bx3_pair_used_splits(splits, FIN, FP, T, T, FIN, FP);
        tf[w] = time_fn(iters, [&] { RC(eagcn_gemm_bx3(0, T, FP, FIN, X.pl, X.pstride, FIN, X.cap, WT.pl, WT.pstride, FIN, WT.cap, P, FP, 1, 0, 3, nullptr)); });
        tx[w] = time_fn(iters, [&] { RC(eagcn_gemm_bx3(0, T, FIN, FP, dP.pl, dP.pstride, FP, dP.cap, W.pl, W.pstride, FP, W.cap, dX, FIN, 1, 0, 3, nullptr)); });
        tw[w] = time_fn(iters, [&] { RC(eagcn_gemm_bx3(1, FIN, FP, T, X.pl, X.pstride, FIN, X.cap, dP.pl, dP.pstride, FP, dP.cap, dW, FP, splits, slab, 3, nullptr)); });
        tp[w] = time_fn(iters, [&] { RC(eagcn_gemm_bx3_pair(T, FIN, FP, dP.pl, dP.pstride, FP, dP.cap, W.pl, W.pstride, FP, W.cap, dX, FIN, FIN, FP, T, X.pl, X.pstride, FIN, X.cap, dP.pl,
                                                            dP.pstride, FP, dP.cap, dW, FP, splits, slab, 3, nullptr)); });
        eagcn_set_bx3_wide(old);
    }
    printf("  (dW k-chunks used: 128-tile kernel %d alone / %d beside dX; 256-tile kernel %d / %d)\n", used[0], usedp[0], used[1], usedp[1]);
    t = time_fn(iters, [&] { RC(eagcn_gemm_f32_sk(0, 1, T, FP, FIN, X.d, FIN, WT.d, FIN, P, FP, ws, wsb, nullptr)); });
    printf("  forward   128x128 %8.1f us %6.1f TF | 256x128 %8.1f us %6.1f TF | fp32 MFMA %8.1f us %6.1f TF\n", tf[0], f1 / tf[0] * 1e-6, tf[1], f1 / tf[1] * 1e-6, t, f1 / t * 1e-6);
    printf("  dX        128x128 %8.1f us %6.1f TF | 256x128 %8.1f us %6.1f TF\n", tx[0], f1 / tx[0] * 1e-6, tx[1], f1 / tx[1] * 1e-6);
    printf("  dW        128x128 %8.1f us %6.1f TF | 256x128 %8.1f us %6.1f TF\n", tw[0], f1 / tw[0] * 1e-6, tw[1], f1 / tw[1] * 1e-6);
    t = time_fn(iters, [&] { RC(eagcn_gemm_pair_sk(T, FIN, FP, dP.d, FP, W.d, FP, dX, FIN, FIN, FP, T, X.d, FIN, dP.d, FP, dW, FP, ws, wsb, nullptr)); });
    printf("  dX + dW   128x128 %8.1f us %6.1f TF | 256x128 %8.1f us %6.1f TF | fp32 MFMA %8.1f us %6.1f TF\n", tp[0], 2 * f1 / tp[0] * 1e-6, tp[1], 2 * f1 / tp[1] * 1e-6, t, 2 * f1 / t * 1e-6);
    fflush(stdout);
    (void)hipFree(P); (void)hipFree(dX); (void)hipFree(dW); (void)hipFree(ws);
}

// signed error of long same-sign reductions: does the accumulation round or truncate?
static void bias_test(int K) {
    const int M = 128, N = 128;
    const int lda = M, ldb = N, ldc = N;
    Mat A(K, M, lda, 1.0f), B(K, N, ldb, 1.0f);
    for (auto& v : A.h) v = fabsf(v);
    for (auto& v : B.h) v = fabsf(v);
    CK(hipMemcpy(A.d, A.h.data(), A.h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.d, B.h.data(), B.h.size() * 4, hipMemcpyHostToDevice));
    RC(eagcn_bx3_split(A.d, K, lda, A.pl, A.pstride, A.cap, 3, nullptr));
    RC(eagcn_bx3_split(B.d, K, ldb, B.pl, B.pstride, B.cap, 3, nullptr));
    float *dC, *dC2; CK(hipMalloc(&dC, M * ldc * 4)); CK(hipMalloc(&dC2, M * ldc * 4));
    size_t wsb = eagcn_gemm_sk_workspace_bytes();
    void* ws; CK(hipMalloc(&ws, wsb));
    CK(hipMemset(dC, 0, M * ldc * 4));
    RC(eagcn_gemm_bx3(1, M, N, K, A.pl, A.pstride, lda, A.cap, B.pl, B.pstride, ldb, B.cap, dC, ldc, 1, (size_t)M * ldc, 3, nullptr));
    RC(eagcn_gemm_f32_sk(1, 0, M, N, K, A.d, lda, B.d, ldb, dC2, ldc, ws, wsb, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> C(M * ldc), C2(M * ldc);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
    double sb = 0, s2 = 0, mb = 0, m2 = 0; int n = 0;
    for (int i = 0; i < M; i += 3) for (int j = 0; j < N; j += 5) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)A.h[(size_t)k * lda + i] * B.h[(size_t)k * ldb + j];
        const double e1 = (C[i * ldc + j] - s) / s, e2 = (C2[i * ldc + j] - s) / s;
        sb += e1; s2 += e2; mb = fmax(mb, fabs(e1)); m2 = fmax(m2, fabs(e2)); ++n;
    }
    printf("same-sign reduction K=%6d: bx3 mean signed rel err %+.2e (max %.2e) | fp32 MFMA %+.2e (max %.2e)   [eps = 6e-8]\n", K, sb / n, mb, s2 / n, m2);
    (void)hipFree(dC); (void)hipFree(dC2); (void)hipFree(ws);
}

// heavy cancellation: nearly constant A columns against zero-mean B columns (what a BatchNorm backward feeds the weight gradient)
static void cancel_test(int K, float spread) {
    const int M = 128, N = 128;
    const int lda = M, ldb = N, ldc = N;
    Mat A(K, M, lda, 1.0f), B(K, N, ldb, 1.0f);
    for (auto& v : A.h) v = 1.0f + spread * v;
    for (int j = 0; j < N; ++j) { double m = 0; for (int k = 0; k < K; ++k) m += B.h[(size_t)k * ldb + j]; m /= K; for (int k = 0; k < K; ++k) B.h[(size_t)k * ldb + j] -= (float)m; }
    CK(hipMemcpy(A.d, A.h.data(), A.h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.d, B.h.data(), B.h.size() * 4, hipMemcpyHostToDevice));
    RC(eagcn_bx3_split(A.d, K, lda, A.pl, A.pstride, A.cap, 3, nullptr));
    RC(eagcn_bx3_split(B.d, K, ldb, B.pl, B.pstride, B.cap, 3, nullptr));
    float *dC, *dC2; CK(hipMalloc(&dC, M * ldc * 4)); CK(hipMalloc(&dC2, M * ldc * 4));
    size_t wsb = eagcn_gemm_sk_workspace_bytes();
    void* ws; CK(hipMalloc(&ws, wsb));
    CK(hipMemset(dC, 0, M * ldc * 4));
    RC(eagcn_gemm_bx3(1, M, N, K, A.pl, A.pstride, lda, A.cap, B.pl, B.pstride, ldb, B.cap, dC, ldc, 1, (size_t)M * ldc, 3, nullptr));
    RC(eagcn_gemm_f32_sk(1, 0, M, N, K, A.d, lda, B.d, ldb, dC2, ldc, ws, wsb, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<float> C(M * ldc), C2(M * ldc);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(C2.data(), dC2, C2.size() * 4, hipMemcpyDeviceToHost));
    double num1 = 0, num2 = 0, den = 0, sab = 0; int n = 0;
    for (int i = 0; i < M; i += 3) for (int j = 0; j < N; j += 5) {
        double s = 0, sa = 0; for (int k = 0; k < K; ++k) { const double t = (double)A.h[(size_t)k * lda + i] * B.h[(size_t)k * ldb + j]; s += t; sa += fabs(t); }
        num1 = fmax(num1, fabs(C[i * ldc + j] - s)); num2 = fmax(num2, fabs(C2[i * ldc + j] - s)); den = fmax(den, fabs(s)); sab = fmax(sab, sa); ++n;
    }
    printf("cancellation K=%6d spread %.0e: max|result| / sum|a||b| = %.1e;  max err / max|result|: bx3 %.2e | fp32 MFMA %.2e\n", K, spread, den / sab, num1 / den, num2 / den);
    (void)hipFree(dC); (void)hipFree(dC2); (void)hipFree(ws);
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "check";
    if (!strcmp(mode, "cancel")) { cancel_test(25000, 1.0f); cancel_test(25000, 1e-1f); cancel_test(25000, 1e-2f); cancel_test(25000, 1e-3f); cancel_test(4096, 1e-2f); return 0; }
    if (!strcmp(mode, "bias")) { bias_test(512); bias_test(4096); bias_test(25000); bias_test(100000); return 0; }
    printf("abi %d\n", eagcn_abi_version());
    if (!strcmp(mode, "check")) {
      for (int w = 0; w < 2; ++w) {
        printf("---- %s kernel\n", w ? "256 x 128 (gemm_bx3w.hip)" : "128 x 128 (gemm_bx3.hip)");
        eagcn_set_bx3_wide(w);
        check_nt(77000, 130, 72, 0);
        check_tn(300, 260, 33000, 40, 0);
        check_nt(128, 128, 32, 0);
        check_nt(128, 128, 64, 0);
        check_nt(100, 90, 48, 7);
        check_nt(300, 200, 400, 3);
        check_nt(4809, 704, 400, 11);
        check_nt(4809, 400, 720, 0);
        check_nt(37, 16, 128, 0);
        check_nt(2500, 1264, 512, 0);
        check_tn(128, 128, 32, 1, 0);
        check_tn(128, 128, 64, 1, 0);
        check_tn(100, 90, 75, 1, 5);
        check_tn(400, 720, 4809, 6, 40);
        check_tn(400, 720, 4809, 1, 0);
        check_tn(512, 1024, 3000, 3, 0);
        check_tn(128, 16, 37, 2, 0);
      }
        eagcn_set_bx3_wide(-1);
        printf(fails ? "BX3_CHECK FAILED (%d)\n" : "BX3_CHECK OK\n", fails);
        return fails ? 1 : 0;
    }
    if (getenv("BX3_WIDE")) eagcn_set_bx3_wide(atoi(getenv("BX3_WIDE")));
    if (!strcmp(mode, "fwd")) {          // fwd <T> <FIN> <FP> <iters> <wide>: the forward product only, one kernel (counter passes)
        const int T = atoi(argv[2]), FIN = atoi(argv[3]), FP = atoi(argv[4]), iters = argc > 5 ? atoi(argv[5]) : 5;
        eagcn_set_bx3_wide(argc > 6 ? atoi(argv[6]) : 1);
        Mat X(T, FIN, FIN), WT(FP, FIN, FIN, 0.05f);
        float* P; CK(hipMalloc(&P, (size_t)T * FP * 4));
        const float t = time_fn(iters, [&] { RC(eagcn_gemm_bx3(0, T, FP, FIN, X.pl, X.pstride, FIN, X.cap, WT.pl, WT.pstride, FIN, WT.cap, P, FP, 1, 0, 3, nullptr)); });
        printf("forward T=%d F_in=%d Fp=%d: %.1f us  %.1f TF\n", T, FIN, FP, t, 2.0 * T * FIN * FP / t * 1e-6);
        return 0;
    }
    if (!strcmp(mode, "one")) {          // one <T> <FIN> <FP> <iters>: a single layer shape (profiling runs)
        time_layer("one", atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 5);
        return 0;
    }
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    time_layer("tox21 c2 B=256 L2", 4809, 400, 720, iters);
    time_layer("tox21 c2 B=1024 L2", 19200, 400, 720, iters);
    time_layer("hiv c3 L2", 25000, 512, 6320, iters / 2 + 1);
    time_layer("lipo c4 L3", 14000, 512, 1040, iters);
    time_layer("c5 synth L2", 262144, 512, 1024, iters / 4 + 1);
    return 0;
}
