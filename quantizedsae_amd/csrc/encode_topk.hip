// encode_topk.hip -- encoder + per-row top-k without a [B, H] latent in HBM.
//
// Replaces `latent = encode(x); latent.topk(k)` (reference sae/binary.py:93-94,
// sae/baseline.py:22-35).  Current form: the batch is walked in row chunks whose dense latent
// (chunk x H fp32) stays in a small workspace that fits the 256 MiB Infinity Cache; each
// chunk is contracted by the exact-fp32 MFMA kernel and reduced to (idx, val) by the top-k
// kernel before the next chunk overwrites the workspace.  Results are identical to
// qsae_encode_dense + qsae_topk_rows.
#include "common.h"

namespace qsae {
constexpr int kChunkRows = 1024;   // 1024 x 32768 x 4 B = 128 MiB of latent per chunk
}

using namespace qsae;

extern "C" size_t qsae_encode_topk_workspace_bytes(int B, int D, int H, int k) {
    (void)D; (void)k;
    if (B <= 0 || H <= 0) return 0;
    const size_t rows = static_cast<size_t>(B < kChunkRows ? B : kChunkRows);
    return rows * static_cast<size_t>(H) * sizeof(float);
}

extern "C" int qsae_encode_topk(const float* x, const float* W, const float* bias, int B, int D, int H, int k,
                                int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                                qsae_stream_t stream) {
    QSAE_CHECK_ARG(B >= 0 && D > 0 && H > 0, "B >= 0, D > 0, H > 0 required");
    if (B == 0) return QSAE_OK;
    QSAE_CHECK_ARG(x && W && idx && val && workspace, "null pointer");
    QSAE_CHECK_ARG(k >= 1 && k <= H, "1 <= k <= H required");
    if (workspace_bytes < qsae_encode_topk_workspace_bytes(B, D, H, k))
        return fail(QSAE_ERR_WORKSPACE, "%s: workspace too small", __func__);
    QSAE_CHECK_ARG(aligned16(workspace), "workspace must be 16-byte aligned");
    float* lat = static_cast<float*>(workspace);
    for (int b0 = 0; b0 < B; b0 += kChunkRows) {
        const int rows = (B - b0) < kChunkRows ? (B - b0) : kChunkRows;
        int rc = qsae_encode_dense(x + static_cast<size_t>(b0) * D, W, bias, rows, D, H, QSAE_ACT_NONE, lat, H, stream);
        if (rc != QSAE_OK) return rc;
        rc = qsae_topk_rows(lat, H, rows, H, k, idx + static_cast<size_t>(b0) * k, val + static_cast<size_t>(b0) * k, 0,
                            stream);
        if (rc != QSAE_OK) return rc;
    }
    return QSAE_OK;
}
