// abi_forward.cpp -- a native (non-Python) client of the C ABI in include/qsae.h.
//
// Runs the BinarySAE forward of the hot path (reference sae/binary.py:91-103 + :24-47) the way a host
// program written against the header would: its own hipMalloc'd buffers, plain pointers and sizes, one
// stream -- and checks every output against the CPU oracle (oracle/qsae_oracle.c, test infrastructure,
// linked here as the checker only).  Built and run by tests/test_c_abi_gpu.py; prints "PASS" or the
// first mismatch.  Usage: abi_forward [B D H k n_bits]
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "qsae.h"

extern "C" {
void qsae_oracle_encode(const float* x, const float* W, const float* bias, int B, int D, int H, int act, float* out);
void qsae_oracle_topk(const float* latent, int B, int H, int k, int32_t* idx, float* val);
int qsae_oracle_binary_row_bytes(int D, int n);
void qsae_oracle_pack_binary(const float* logits, int H, int D, int n, uint8_t* packed);
void qsae_oracle_decode_binary(const int32_t* idx, const float* val, int B, int k, const uint8_t* packed, int D, int n,
                               float step, const float* bias, float* recon);
double qsae_oracle_sq_err_sum(const float* recon, const float* x, size_t n);
}

#define HIP_OK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));            \
            return 2;                                                                    \
        }                                                                                \
    } while (0)
#define QSAE_OK_OR_DIE(call)                                                             \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != QSAE_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, qsae_last_error());            \
            return 3;                                                                    \
        }                                                                                \
    } while (0)

static uint64_t g_state = 0x243F6A8885A308D3ull;
static double uniform01() {                       // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);
}
static float normal() {
    const double u1 = uniform01() + 1e-300, u2 = uniform01();
    return static_cast<float>(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
}

template <class T>
static int upload(const std::vector<T>& h, T** d) {
    HIP_OK(hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(T)));
    HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char** argv) {
    int B = 2304, D = 512, H = 8192, k = 16, n_bits = 4;
    if (argc == 6) { B = atoi(argv[1]); D = atoi(argv[2]); H = atoi(argv[3]); k = atoi(argv[4]); n_bits = atoi(argv[5]); }
    const float gamma = 4.0f, step = gamma / static_cast<float>(1 << (n_bits - 1));
    if (qsae_abi_version() != QSAE_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

    // ---- synthetic checkpoint + batch (xavier-uniform encoder, saturated decoder logits) ------------
    std::vector<float> x(static_cast<size_t>(B) * D), W(static_cast<size_t>(H) * D), bias(H), dbias(D);
    std::vector<float> logits(static_cast<size_t>(H) * D * n_bits);
    const float bound = std::sqrt(6.0f / static_cast<float>(D + H));
    for (auto& v : x) v = normal();
    for (auto& v : W) v = static_cast<float>((uniform01() * 2.0 - 1.0) * bound);
    for (auto& v : bias) v = 0.01f * normal();
    for (auto& v : dbias) v = 0.1f * normal();
    for (auto& v : logits) v = uniform01() < 0.5 ? -30.0f : 30.0f;

    // ---- oracle -------------------------------------------------------------------------------------
    std::vector<float> lat(static_cast<size_t>(B) * H), oval(static_cast<size_t>(B) * k), orecon(static_cast<size_t>(B) * D);
    std::vector<int32_t> oidx(static_cast<size_t>(B) * k);
    const int rb = qsae_oracle_binary_row_bytes(D, n_bits);
    std::vector<uint8_t> opacked(static_cast<size_t>(H) * rb);
    qsae_oracle_encode(x.data(), W.data(), bias.data(), B, D, H, 0, lat.data());
    qsae_oracle_topk(lat.data(), B, H, k, oidx.data(), oval.data());
    qsae_oracle_pack_binary(logits.data(), H, D, n_bits, opacked.data());
    qsae_oracle_decode_binary(oidx.data(), oval.data(), B, k, opacked.data(), D, n_bits, step, dbias.data(), orecon.data());
    const double osq = qsae_oracle_sq_err_sum(orecon.data(), x.data(), static_cast<size_t>(B) * D);

    // ---- the library, through the header only -----------------------------------------------------------
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float *dx, *dW, *dbias_enc, *dbias_dec, *dlogits;
    if (upload(x, &dx) || upload(W, &dW) || upload(bias, &dbias_enc) || upload(dbias, &dbias_dec) || upload(logits, &dlogits)) return 2;
    if (qsae_binary_row_bytes(D, n_bits) != rb) { fprintf(stderr, "row bytes differ\n"); return 1; }
    uint8_t* dpacked;
    void *dWq, *dws;
    float *dmeta, *dval, *ddense, *drecon;
    int32_t* didx;
    double* dsq;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dpacked), static_cast<size_t>(H) * rb));
    HIP_OK(hipMalloc(&dWq, qsae_prefilter_w_bytes(H, D)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dmeta), 4 * sizeof(float)));
    HIP_OK(hipMemset(dmeta, 0, 4 * sizeof(float)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&didx), static_cast<size_t>(B) * k * 4));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dval), static_cast<size_t>(B) * k * 4));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&ddense), static_cast<size_t>(B) * H * 4));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&drecon), static_cast<size_t>(B) * D * 4));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dsq), sizeof(double)));
    HIP_OK(hipMemset(dsq, 0, sizeof(double)));
    const size_t ws_bytes = qsae_encode_topk_prefilter_workspace_bytes(B, D, H, k);
    const bool prefilter = ws_bytes != 0;                 // shapes outside the candidate sweep: the exact fp32 entry
    const size_t ws_exact = qsae_encode_topk_workspace_bytes(B, D, H, k);
    HIP_OK(hipMalloc(&dws, prefilter ? ws_bytes : (ws_exact ? ws_exact : 256)));

    float* dgap;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dgap), sizeof(float)));
    QSAE_OK_OR_DIE(qsae_pack_binary(dlogits, H, D, n_bits, dpacked, nullptr, dgap, stream));  // once per checkpoint
    float gap = -1.f;
    HIP_OK(hipMemcpyAsync(&gap, dgap, sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (!(gap >= 0.f && gap < 1e-6f)) { printf("FAIL soft/hard gap %g: the fixture's logits are saturated\n", gap); return 1; }
    if (prefilter) {
        QSAE_OK_OR_DIE(qsae_prefilter_pack_w(dW, dbias_enc, H, D, dWq, dmeta, stream));
        int flagged = -1;
        QSAE_OK_OR_DIE(qsae_encode_topk_prefilter(dx, dW, dbias_enc, dWq, dmeta, B, D, H, k, didx, dval, ddense, H, dws,
                                                  ws_bytes, /*spec_rows=*/0, &flagged, stream));
        if (flagged < 0 || flagged > B) { printf("FAIL flagged-row count %d\n", flagged); return 1; }
    } else {
        QSAE_OK_OR_DIE(qsae_encode_topk_latent(dx, dW, dbias_enc, B, D, H, k, didx, dval, ddense, H, 0, dws, ws_exact, stream));
    }
    QSAE_OK_OR_DIE(qsae_decode_binary_sparse(didx, dval, B, k, dpacked, H, D, n_bits, step, dbias_dec, drecon, stream));
    QSAE_OK_OR_DIE(qsae_sq_err_sum(drecon, dx, static_cast<size_t>(B) * D, dsq, stream));
    HIP_OK(hipStreamSynchronize(stream));
    // the same forward as ONE call (the refinement kernel decodes the rows): must reproduce the two-call outputs
    std::vector<float> frecon;
    std::vector<int32_t> fidx;
    if (prefilter) {
        int32_t* didx2;
        float *dval2, *drecon2;
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&didx2), static_cast<size_t>(B) * k * 4));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&dval2), static_cast<size_t>(B) * k * 4));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&drecon2), static_cast<size_t>(B) * D * 4));
        QSAE_OK_OR_DIE(qsae_binary_forward_prefilter(dx, dW, dbias_enc, dWq, dmeta, B, D, H, k, dpacked, n_bits, step, dbias_dec,
                                                     didx2, dval2, nullptr, 0, drecon2, dws, ws_bytes, /*spec_rows=*/32, nullptr,
                                                     stream));
        HIP_OK(hipStreamSynchronize(stream));
        frecon.resize(static_cast<size_t>(B) * D);
        fidx.resize(static_cast<size_t>(B) * k);
        HIP_OK(hipMemcpy(frecon.data(), drecon2, frecon.size() * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(fidx.data(), didx2, fidx.size() * 4, hipMemcpyDeviceToHost));
        // ... and as submit / finish, the host waiting on its own event instead of inside the library
        int* flagged_host;
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&flagged_host), sizeof(int), hipHostMallocDefault));
        *flagged_host = -1;
        hipEvent_t landed;
        HIP_OK(hipEventCreateWithFlags(&landed, hipEventDisableTiming));
        HIP_OK(hipMemsetAsync(drecon2, 0xFF, static_cast<size_t>(B) * D * 4, stream));
        HIP_OK(hipMemsetAsync(didx2, 0xFF, static_cast<size_t>(B) * k * 4, stream));
        QSAE_OK_OR_DIE(qsae_prefilter_submit(dx, dW, dbias_enc, dWq, dmeta, B, D, H, k, dpacked, n_bits, step, dbias_dec, didx2,
                                             dval2, nullptr, 0, drecon2, dws, ws_bytes, flagged_host, stream));
        HIP_OK(hipEventRecord(landed, stream));
        HIP_OK(hipEventSynchronize(landed));
        if (*flagged_host < 0 || *flagged_host > B) { printf("FAIL submit: flagged-row count %d\n", *flagged_host); return 1; }
        QSAE_OK_OR_DIE(qsae_prefilter_finish(dx, dW, dbias_enc, dWq, dmeta, B, D, H, k, dpacked, n_bits, step, dbias_dec, didx2,
                                             dval2, nullptr, 0, drecon2, dws, ws_bytes, *flagged_host, stream));
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<float> srecon(frecon.size());
        std::vector<int32_t> sidx(fidx.size());
        HIP_OK(hipMemcpy(srecon.data(), drecon2, srecon.size() * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(sidx.data(), didx2, sidx.size() * 4, hipMemcpyDeviceToHost));
        if (memcmp(srecon.data(), frecon.data(), frecon.size() * 4) != 0 || memcmp(sidx.data(), fidx.data(), fidx.size() * 4) != 0) {
            printf("FAIL submit / finish differs from the one-call forward\n");
            return 1;
        }
    }

    std::vector<int32_t> gidx(oidx.size());
    std::vector<float> gval(oval.size()), grecon(orecon.size()), gdense(lat.size());
    std::vector<uint8_t> gpacked(opacked.size());
    double gsq = 0.0;
    HIP_OK(hipMemcpy(gidx.data(), didx, gidx.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gval.data(), dval, gval.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(grecon.data(), drecon, grecon.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gdense.data(), ddense, gdense.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gpacked.data(), dpacked, gpacked.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&gsq, dsq, sizeof(double), hipMemcpyDeviceToHost));

    // ---- comparisons: bit-exact except the fp64 sum (order of the partial sums) ----------------------
    if (memcmp(gpacked.data(), opacked.data(), opacked.size()) != 0) { printf("FAIL packed dictionary\n"); return 1; }
    if (memcmp(gidx.data(), oidx.data(), oidx.size() * 4) != 0) { printf("FAIL top-k indices\n"); return 1; }
    if (memcmp(gval.data(), oval.data(), oval.size() * 4) != 0) { printf("FAIL top-k values\n"); return 1; }
    if (memcmp(grecon.data(), orecon.data(), orecon.size() * 4) != 0) { printf("FAIL reconstruction\n"); return 1; }
    if (prefilter && (memcmp(frecon.data(), orecon.data(), orecon.size() * 4) != 0 ||
                      memcmp(fidx.data(), oidx.data(), oidx.size() * 4) != 0)) { printf("FAIL one-call forward\n"); return 1; }
    size_t nonzero = 0;
    for (int b = 0; b < B; ++b) {
        const float* row = gdense.data() + static_cast<size_t>(b) * H;
        for (int h = 0; h < H; ++h) nonzero += row[h] != 0.0f || std::signbit(row[h]);
        for (int j = 0; j < k; ++j) {
            const int32_t h = oidx[static_cast<size_t>(b) * k + j];
            if (memcmp(&row[h], &oval[static_cast<size_t>(b) * k + j], 4) != 0) { printf("FAIL dense latent entry\n"); return 1; }
        }
    }
    size_t expect_nonzero = 0;
    for (float v : oval) expect_nonzero += v != 0.0f || std::signbit(v);
    if (nonzero != expect_nonzero) { printf("FAIL dense latent has %zu non-zeros, expected %zu\n", nonzero, expect_nonzero); return 1; }
    if (std::fabs(gsq - osq) > 1e-9 * std::fabs(osq)) { printf("FAIL squared error %.17g vs %.17g\n", gsq, osq); return 1; }
    printf("PASS B=%d D=%d H=%d k=%d n_bits=%d path=%s mse=%.9g\n", B, D, H, k, n_bits, prefilter ? "prefilter" : "exact",
           gsq / (static_cast<double>(B) * D));
    return 0;
}
