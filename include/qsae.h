/*
 * qsae.h -- C ABI of the MI355X (gfx950) quantized-SAE forward backend.
 *
 * Drop-in boundary for the forward hot path of ASSERT-KTH/QuantizedSAE.  The reference
 * is pure Python/PyTorch and has no FFI of its own; each entry
 * point below replaces the ATen call sequence cited next to it (paths relative to the
 * reference root).  Everything here is stateless: plain device pointers and sizes, no
 * torch types, work is enqueued on the given HIP stream and the call returns without
 * synchronising (exceptions are marked "one host round trip").  No call leaves anything
 * behind that a later call reads: what the library keeps is per device ("this kernel's LDS limit
 * is raised on device d") or owned by one host thread for one device (a side stream, three
 * events, one pinned word -- created on first use).  Calls may therefore be made from several
 * host threads and on several devices of one process; the caller makes the device of its
 * pointers current, as for any HIP library, and gives every concurrently running call its own
 * workspace.  (Several host threads: tested on hardware.  Several devices in ONE process: by
 * construction only -- no box with two visible devices has run it yet; one process per GPU is the
 * tested deployment.)  The Python host side in quantizedsae_amd/ binds these with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - all tensors row-major, contiguous, device memory unless stated;
 *   - B rows of activations, D = input_dim, H = hidden_dim, k = top-k;
 *   - return value: QSAE_OK or a negative QSAE_ERR_* code; qsae_last_error() gives text;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - B == 0 is a valid no-op everywhere.
 *
 * Numerical contract (bit-exact against oracle/qsae_oracle.c):
 *   encoder latent = fp32 fmaf chain over k ascending seeded with bias (what
 *   v_mfma_f32_32x32x2_f32 computes when K is walked in order); top-k = k largest by
 *   (value desc, index asc), NaN above +inf; sparse decode = ascending-index fmaf chain,
 *   then separately rounded *step and +bias.
 */
#ifndef QSAE_H
#define QSAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QSAE_ABI_VERSION 4

#define QSAE_OK 0
#define QSAE_ERR_INVALID_ARG (-1)  /* null pointer, non-positive dim, misaligned pointer      */
#define QSAE_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels implement                 */
#define QSAE_ERR_HIP (-3)          /* a HIP runtime call failed (text in qsae_last_error())    */
#define QSAE_ERR_WORKSPACE (-4)    /* workspace too small; see the *_workspace_bytes() helpers */

/* activation applied by qsae_encode_dense */
#define QSAE_ACT_NONE 0    /* BinarySAE / Baseline: sae/binary.py:82-84, sae/baseline.py:8-10 */
#define QSAE_ACT_RELU 1    /* Ternary: sae/ternary.py:95-98                                    */
#define QSAE_ACT_SIGMOID 2 /* Matryoshka: sae/quantized_matryoshka.py:206-209                  */

typedef void* qsae_stream_t;

/* -- library ---------------------------------------------------------------------------- */
int qsae_abi_version(void);
/* Thread-local text of the last failing call ("" if none). */
const char* qsae_last_error(void);
/* Properties of the current device: compute units, name of the gfx target (e.g. "gfx950"). */
int qsae_device_info(int* cu_count, char* arch, int arch_len);

/* -- profiling ---------------------------------------------------------------------------- */
/* One-shot, per host thread: the NEXT fused / prefilter / bits-prefilter call of the calling thread records
 * ev_begin (hipEvent_t) on its stream right before its candidate-sweep launch and ev_end right after it (after the
 * co-resident fill kernel has joined, where there is one), then forgets them.  NULL, NULL clears a pending pair.
 * This is how bench.py times the dominant kernel live on the launch stream. */
int qsae_profile_sweep_events(void* ev_begin, void* ev_end);
/* The events for the call above, created / read / destroyed through the HIP runtime this library is linked against
 * (timing enabled).  elapsed: QSAE_ERR_HIP while either event has not completed. */
int qsae_profile_event_create(void** ev);
int qsae_profile_event_destroy(void* ev);
int qsae_profile_event_elapsed_ms(void* ev_begin, void* ev_end, float* ms);
/* Fraction of the encoder's 2 B D H FLOPs that the profiled launch covers for hidden width H (1.0 when the sweep
 * derives its thresholds itself, else (H - pilot block) / H). */
double qsae_profile_sweep_flop_fraction(int H);

/* -- encoder ---------------------------------------------------------------------------- */
/* K-interleaved operand layout.  dst[r][8g + j/2 + 4*(j&1)] = src[r][8g + j]: every 8 consecutive
 * k of a row stored as [k0 k2 k4 k6 k1 k3 k5 k7] -- the order the fp32 MFMA consumes them from LDS,
 * so that staged 16-byte chunks need no register shuffle.  Purely a storage permutation: the
 * contraction still runs in ascending k and gives bit-identical results.  W_enc is permuted once
 * per checkpoint, x once per batch (a 2 x B x D x 4-byte copy).  K % 8 == 0. */
int qsae_kperm_rows(const float* src, int rows, int K, float* dst, qsae_stream_t stream);

/* out[b][h] = act(bias[h] + sum_k x[b][k] * W[h][k])           (fp32 MFMA, exact fmaf chain)
 * Replaces nn.Linear (+ReLU / +Sigmoid) in SparseAutoencoder.encode, sae/base.py:16-19.
 * x [B][D], W [H][D], bias [H] or NULL, out [B][out_ld] with out_ld >= H.
 * Requires D % 4 == 0 and 16-byte aligned x, W. */
int qsae_encode_dense(const float* x, const float* W, const float* bias, int B, int D, int H,
                      int act, float* out, int64_t out_ld, qsae_stream_t stream);
/* Same with x and W already K-interleaved (qsae_kperm_rows); D % 32 == 0. */
int qsae_encode_dense_kperm(const float* xp, const float* Wp, const float* bias, int B, int D, int H,
                            int act, float* out, int64_t out_ld, qsae_stream_t stream);

/* The same contraction at fp32 ACCURACY, not fp32 bit-exactness, on the fp16 matrix pipe (opt-in; nothing that ranks or
 * thresholds latents uses it): both operands are split into two fp16 terms under power-of-two scales (x per row, W global),
 * the three partial contractions x1.w1 + x1.w2 + x2.w1 (every product exact in fp32) run as one fp16 GEMM over a concatenated
 * K of 3 D with fp32 accumulation.  The result differs from qsae_encode_dense by accumulation-order noise (~1e-6 of a latent's
 * standard deviation -- the size of the reference's own sgemm-vs-chain difference).  Rows or weights that are not finite give
 * NaN outputs for the whole row / everything.  D % 64 == 0.
 *   qsae_emu_pack_w: once per checkpoint, W [H][D] -> Wc (qsae_emu_w_bytes() = 6 H D bytes, opaque), meta2 = 2 device floats;
 *   qsae_encode_dense_emu: workspace from qsae_encode_dense_emu_workspace_bytes(B, D) (the split copy of the batch). */
size_t qsae_emu_w_bytes(int H, int D);
int qsae_emu_pack_w(const float* W, int H, int D, void* Wc, float* meta2, qsae_stream_t stream);
size_t qsae_encode_dense_emu_workspace_bytes(int B, int D);
int qsae_encode_dense_emu(const float* x, const void* Wc, const float* meta2, const float* bias, int B, int D, int H, int act,
                          float* out, int64_t out_ld, void* workspace, size_t workspace_bytes, qsae_stream_t stream);

/* zbits[b][w] bit j = (sigmoid(pre[b][32w+j]) > 0.5) == (pre >= 0x33C00001), pre as above.
 * Replaces encoder(x) followed by `latent > 0.5`, sae/quantized_matryoshka.py:97-99,206-209
 * (also scripts/analysis/dynamic_analysis.py:51).  zbits [B][words_ld] uint32, words_ld >=
 * ceil(H/32); bits beyond H are zero. */
int qsae_encode_bits(const float* x, const float* W, const float* bias, int B, int D, int H,
                     uint32_t* zbits, int64_t words_ld, qsae_stream_t stream);

/* Per-row top-k of a dense latent [B][ld]: idx/val [B][k] ordered by (value desc, index asc).
 * If zero_rest != 0 every non-selected entry of `latent` is overwritten with +0 in place, which
 * yields `latent * mask` of sae/binary.py:94-99 and the scatter of sae/baseline.py:34-40.
 * Replaces torch.topk + zeros_like + scatter_.  1 <= k <= min(H, 256), H <= 32768. */
int qsae_topk_rows(float* latent, int64_t ld, int B, int H, int k, int32_t* idx, float* val,
                   int zero_rest, qsae_stream_t stream);

/* Fused encoder + top-k without materialising the dense latent: same results as
 * qsae_encode_dense(act=NONE) followed by qsae_topk_rows.  Workspace from
 * qsae_encode_topk_workspace_bytes(). */
size_t qsae_encode_topk_workspace_bytes(int B, int D, int H, int k);
int qsae_encode_topk(const float* x, const float* W, const float* bias, int B, int D, int H, int k,
                     int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                     qsae_stream_t stream);
/* Same with x and W already K-interleaved (qsae_kperm_rows); D % 32 == 0. */
int qsae_encode_topk_kperm(const float* xp, const float* Wp, const float* bias, int B, int D, int H, int k,
                           int32_t* idx, float* val, void* workspace, size_t workspace_bytes,
                           qsae_stream_t stream);

/* qsae_encode_topk plus the reference's dense return value: dense[b][h] = latent if h is one of row
 * b's top-k else +0 (`latent * mask`, sae/binary.py:96-99; `zeros_like + scatter_`,
 * sae/baseline.py:38-40).  In the fused form the sweep zero-fills the dense tensor tile by tile
 * underneath its own MFMAs and the k survivors are scattered in at the end -- no separate memset
 * pass, no dense latent read back.  kperm != 0: x and W are K-interleaved.  Same workspace. */
int qsae_encode_topk_latent(const float* x, const float* W, const float* bias, int B, int D, int H, int k,
                            int32_t* idx, float* val, float* dense, int64_t dense_ld, int kperm,
                            void* workspace, size_t workspace_bytes, qsae_stream_t stream);

/* Order-preserving fp16 prefilter for the same operation.  A fp16 MFMA pass (per-row power-of-two
 * scaling, fp32 accumulation) with a rigorous per-row error bound eps_b only decides which hidden units
 * CAN belong to a row's top-k (everything with approximate value >= approximate k-th - 2 eps_b, ~90 of
 * 32768); those survivors are re-evaluated with the exact fp32 fmaf chain and ranked exactly, so idx,
 * val and dense are bit-identical to qsae_encode_topk_latent.  Rows the bound cannot serve (non-finite
 * inputs, overflowing lists) go through the exact kernels.  Wq/meta come from qsae_prefilter_pack_w
 * (once per checkpoint: Wq = H*D fp16, meta = 4 device floats -- weight scale, largest row norm, largest |bias|, largest
 * distance between a row and its fp16 copy; opaque to the caller).  D % 64 == 0, D <= 2048; other shapes
 * return QSAE_ERR_UNSUPPORTED (use qsae_encode_topk_latent).  dense may be NULL.  For D in {128, 256, 512} the
 * candidate pass is one launch (activation rows stationary in registers, fp16 weights streamed once per workgroup)
 * that also derives the row thresholds and writes the zeros of `dense`; the survivors are written by the refinement.
 * With D = 512 and a dense output the zeros are written by a second kernel that runs beside the candidate pass on a
 * side stream owned by the calling thread (one per device), forked from and joined back into `stream` inside the call
 * (the call stays ordered on `stream`).
 * One host round trip per call: the number of rows sent through the exact kernels (normally 0-5 of 65536), which also
 * comes back in *flagged_rows (host int, may be NULL).  spec_rows > 0 lets the device recompute the first spec_rows
 * flagged rows while the host waits for that count (worth it only when the previous batch had flagged rows; <= 1024).
 * qsae_prefilter_submit / _finish below are the same call without the round trip inside. */
size_t qsae_prefilter_w_bytes(int H, int D);
int qsae_prefilter_pack_w(const float* W, const float* bias, int H, int D, void* Wq, float* meta,
                          qsae_stream_t stream);
size_t qsae_encode_topk_prefilter_workspace_bytes(int B, int D, int H, int k);
int qsae_encode_topk_prefilter(const float* x, const float* W, const float* bias, const void* Wq,
                               const float* meta, int B, int D, int H, int k, int32_t* idx, float* val,
                               float* dense, int64_t dense_ld, void* workspace, size_t workspace_bytes,
                               int spec_rows, int* flagged_rows, qsae_stream_t stream);

/* BinarySAE.forward in one call (sae/binary.py:91-103 with binary_decoder.forward, :24-47, on the k kept entries):
 * qsae_encode_topk_prefilter followed by qsae_decode_binary_sparse, with the decode of a row done by the refinement
 * kernel as soon as it has ranked the row (its dictionary gathers and integer converts fill issue slots that the
 * refinement's own gathers leave idle; rows that take the exact fallback are decoded afterwards).  Outputs are
 * bit-identical to the two separate calls.  packed from qsae_pack_binary; same workspace and shape limits as
 * qsae_encode_topk_prefilter; dense may be NULL (compact outputs only). */
int qsae_binary_forward_prefilter(const float* x, const float* W, const float* bias, const void* Wq,
                                  const float* meta, int B, int D, int H, int k, const uint8_t* packed, int n_bits,
                                  float step, const float* dec_bias, int32_t* idx, float* val, float* dense,
                                  int64_t dense_ld, float* recon, void* workspace, size_t workspace_bytes,
                                  int spec_rows, int* flagged_rows, qsae_stream_t stream);

/* The same pipeline in two calls, so that the host never waits inside the library and can queue the next batch behind
 * this one (the blocking forms above are submit + wait + finish):
 *   submit : everything up to the refinement; every row that did not need the exact fallback is final afterwards.  The
 *            number of rows that do need it is copied asynchronously, in stream order, into *flagged_host (a host int,
 *            page-locked if the copy is to be asynchronous); the caller learns when it has landed from an event it
 *            records behind the call (or any later synchronisation of `stream`).
 *   finish : given that number, enqueues the exact fallback for those rows (their idx / val / dense entries and, with a
 *            dictionary, their reconstruction) and, on the paths whose sweep does not write the zeros, the dense latent.
 *            Outputs may be consumed once finish has been enqueued.  Same arguments as submit; workspace untouched in
 *            between (a second batch in flight needs a second workspace).
 * packed == NULL: no reconstruction (qsae_encode_topk_prefilter), n_bits / step / dec_bias / recon ignored. */
int qsae_prefilter_submit(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                          int B, int D, int H, int k, const uint8_t* packed, int n_bits, float step,
                          const float* dec_bias, int32_t* idx, float* val, float* dense, int64_t dense_ld,
                          float* recon, void* workspace, size_t workspace_bytes, int* flagged_host,
                          qsae_stream_t stream);
int qsae_prefilter_finish(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                          int B, int D, int H, int k, const uint8_t* packed, int n_bits, float step,
                          const float* dec_bias, int32_t* idx, float* val, float* dense, int64_t dense_ld,
                          float* recon, void* workspace, size_t workspace_bytes, int flagged,
                          qsae_stream_t stream);

/* The three calls above with an fp32 dictionary [H][D] (row h = the decoder's weights of hidden unit h) in place of the
 * packed n-bit one: recon = scale * sum_j val_j table[idx_j] + dec_bias, the arithmetic of qsae_decode_table_sparse, done
 * by the refinement kernel for every row it ranks.  Two users: BaselineSparseAutoencoder.forward (sae/baseline.py:17-31;
 * table = decoder.weight transposed, scale = 1) and BinarySAE.forward on a checkpoint whose decoder logits are not
 * polarised (sae/binary.py:24-47 with the soft integers of :26-35; table = qsae_binary_soft_table, scale =
 * quantization step).  table and recon 16-byte aligned; otherwise as above. */
int qsae_table_forward_prefilter(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                 int B, int D, int H, int k, const float* table, float scale, const float* dec_bias,
                                 int32_t* idx, float* val, float* dense, int64_t dense_ld, float* recon, void* workspace,
                                 size_t workspace_bytes, int spec_rows, int* flagged_rows, qsae_stream_t stream);
int qsae_prefilter_submit_table(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                int B, int D, int H, int k, const float* table, float scale, const float* dec_bias,
                                int32_t* idx, float* val, float* dense, int64_t dense_ld, float* recon, void* workspace,
                                size_t workspace_bytes, int* flagged_host, qsae_stream_t stream);
int qsae_prefilter_finish_table(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                int B, int D, int H, int k, const float* table, float scale, const float* dec_bias,
                                int32_t* idx, float* val, float* dense, int64_t dense_ld, float* recon, void* workspace,
                                size_t workspace_bytes, int flagged, qsae_stream_t stream);

/* dense[b][h] = val if (b,h) selected else +0; dense [B][ld].  Replaces zeros_like+scatter_. */
int qsae_densify(const int32_t* idx, const float* val, int B, int k, int H, float* dense, int64_t ld,
                 qsae_stream_t stream);

/* -- BinarySAE decoder (sae/binary.py:10-69) ---------------------------------------------- */
/* Bytes per packed dictionary row: D fields of width fw = 1,2,4,8 (smallest power of two >=
 * n_bits), rounded up to a multiple of 4 bytes. */
int qsae_binary_row_bytes(int D, int n_bits);
/* Hard two's-complement packer == binary_decoder.quantized_int_weights(), sae/binary.py:49-58:
 * bit = sigmoid(logit) > 0.5; logits [H][D*n_bits] (column d*n+b = bit b of output d, LSB first,
 * MSB negative) -> packed [H][row_bytes], fields little-endian.  If polarize_sum != NULL the
 * device double receives sum(p(1-p)2^b) (sae/binary.py:42-43; caller divides by H*D*n).
 * If soft_gap != NULL the device float receives max over (h, d) of |soft - hard| with
 * soft = sum_b sigmoid(logit_b) bw_b -- the integer the reference's forward actually multiplies
 * with (sae/binary.py:26-35) -- and hard the packed integer: the distance, in integer steps,
 * between the reference forward and a hard-bit decode of this checkpoint (~1e-12 at +-30
 * logits, ~0.5 for an untrained decoder; +inf for NaN logits).  The host side uses it to pick
 * qsae_decode_binary_sparse (hard) or qsae_decode_table_sparse over qsae_binary_soft_table. */
int qsae_pack_binary(const float* logits, int H, int D, int n_bits, uint8_t* packed,
                     double* polarize_sum, float* soft_gap, qsae_stream_t stream);
/* int_weights[h][d] as fp32 (for decoder_dictionary(), inference/framework.py:114-124). */
int qsae_unpack_binary(const uint8_t* packed, int H, int D, int n_bits, float* int_weights,
                       qsae_stream_t stream);
/* recon[b][d] = step * sum_j val[b][j] * w[idx[b][j]][d] + bias[d]  -- the reference's dense
 * `latent.matmul(int_weights)` (sae/binary.py:38) evaluated on the k non-zeros only. */
int qsae_decode_binary_sparse(const int32_t* idx, const float* val, int B, int k,
                              const uint8_t* packed, int H, int D, int n_bits, float step,
                              const float* bias, float* recon, qsae_stream_t stream);
/* Same with an fp32 table [H][D]: Baseline decoder nn.Linear(H, D) (sae/baseline.py:12,29; the
 * table is decoder.weight transposed) and BinarySAE's "soft" int_weights for unpolarised
 * checkpoints (sae/binary.py:26-38).  scale == 1 skips the multiply. */
int qsae_decode_table_sparse(const int32_t* idx, const float* val, int B, int k, const float* table,
                             int H, int D, float scale, const float* bias, float* recon,
                             qsae_stream_t stream);
/* Soft int_weights table of sae/binary.py:26-35: table[h][d] = sum_b sigmoid(logit)*bw[b]. */
int qsae_binary_soft_table(const float* logits, int H, int D, int n_bits, float* table,
                           qsae_stream_t stream);

/* -- Ternary decoder (sae/ternary.py:41-52) ---------------------------------------------- */
/* codes2 [D][ceil(H/16)] uint32: 2-bit fields, 0 -> 0, 1 -> +1, 3 -> -1 (two's complement),
 * hard = sign(w) * (|w| >= 0.5); w is decoder.weight [D][H]. */
int qsae_pack_ternary(const float* w, int D, int H, uint32_t* codes2, qsae_stream_t stream);
/* recon[b][d] = sum_h h[b][h] * hard[d][h]   (no bias), h [B][ld]. */
int qsae_decode_ternary_dense(const float* h, int64_t ld, int B, int H, const uint32_t* codes2, int D,
                              float* recon, qsae_stream_t stream);

/* -- Matryoshka decoder (sae/quantized_matryoshka.py:10-143) ------------------------------ */
/* level sizes of :25-38; sizes[n_bits]. Host-side helper. */
int qsae_matryoshka_sizes(int H, int n_bits, int32_t* sizes);
/* codes2t [D][ceil(H/16)] uint32 2-bit two's-complement fields of S/2 in {-1,0,+1} where
 * S = sgn(sig(w)>=.5)+sgn(sig(wm)>=.5) (:67-80), transposed so that H is contiguous;
 * scale[j] = reciprocal(||S_j||+1e-8) * 2^(n-i-2) * abs_range/2^(n-1) for j in level i (:82-90).
 * level_sizes: HOST array of n_bits level sizes summing to H, or NULL for qsae_matryoshka_sizes(H)
 * (a caller that pads levels to multiples of 32 passes the padded sizes). */
int qsae_pack_matryoshka(const float* w, const float* wm, int H, int D, int n_bits, float abs_range,
                         const int32_t* level_sizes, uint32_t* codes2t, float* scale,
                         qsae_stream_t stream);
/* levels[i][b][d] cumulative reconstructions (:121-129), l0_counts[i] = number of set z bits in
 * level i over the whole batch (latent_group[i] = l0_counts[i] / B, :128).
 * zbits as produced by qsae_encode_bits.  Level boundaries (and H) must be multiples of 32. */
int qsae_decode_matryoshka(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, int n_bits,
                           const int32_t* level_sizes, const uint32_t* codes2t, const float* scale,
                           const float* bias, int allow_bias, float* levels,
                           unsigned long long* l0_counts, qsae_stream_t stream);
/* The same z bits as qsae_encode_bits (z = sigmoid(x W^T + b) > 0.5, sae/quantized_matryoshka.py:97-99,217-220)
 * without computing every latent in fp32: the fp16 candidate sweep of the prefilter (tau = the fp32 cutoff of
 * sigmoid > 0.5, Wq/meta from qsae_prefilter_pack_w) lists the latents within 2 eps_b of the cutoff or above it;
 * latents above cutoff + eps_b get bit 1 directly, the few inside the band are re-evaluated with the exact fp32
 * chain.  Bit-identical to qsae_encode_bits.  Rows with more than 2048 listed latents (dense activations) and
 * rows with non-finite inputs are recomputed by the exact dense kernel; *flagged_rows (host int, may be NULL)
 * receives their count so that a caller can route a dense-regime model to qsae_encode_bits.  D in {128,256,512},
 * H % 64 == 0, else QSAE_ERR_UNSUPPORTED.  One host round trip per call (that count). */
size_t qsae_encode_bits_prefilter_workspace_bytes(int B, int D, int H);
int qsae_encode_bits_prefilter(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                               int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                               size_t workspace_bytes, int* flagged_rows, qsae_stream_t stream);
/* The same in two calls (see qsae_prefilter_submit / _finish): submit enqueues everything up to the bit resolution and the
 * asynchronous copy of the flagged-row count into *flagged_host; finish, given that count, enqueues the exact dense kernel
 * for those rows.  Bits of unflagged rows are final after submit; workspace untouched in between. */
int qsae_encode_bits_prefilter_submit(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                      int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                      size_t workspace_bytes, int* flagged_host, qsae_stream_t stream);
int qsae_encode_bits_prefilter_finish(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                      int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                      size_t workspace_bytes, int flagged, qsae_stream_t stream);
/* The same bits when the activations are DENSE (an untrained encoder: half of the units fire; the lists of the call above
 * overflow and every row would take the exact fp32 contraction): an fp16 MFMA pass classifies EVERY latent -- bit 1 above
 * cutoff + eps_b, bit 0 below cutoff - eps_b -- and lists only the latents inside the band (~0.7 % of them), which are
 * re-evaluated with the exact fp32 chain.  Bit-identical to qsae_encode_bits.  Rows with more than 1024 band entries or
 * non-finite inputs are recomputed by the exact dense kernel (*flagged_rows).  D % 64 == 0, H % 32 == 0, else
 * QSAE_ERR_UNSUPPORTED.  One host round trip per call. */
size_t qsae_encode_bits_band_workspace_bytes(int B, int D, int H);
int qsae_encode_bits_band(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                          int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                          size_t workspace_bytes, int* flagged_rows, qsae_stream_t stream);
/* ... and in two calls (as qsae_encode_bits_prefilter_submit / _finish). */
int qsae_encode_bits_band_submit(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                 int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                 size_t workspace_bytes, int* flagged_host, qsae_stream_t stream);
int qsae_encode_bits_band_finish(const float* x, const float* W, const float* bias, const void* Wq, const float* meta,
                                 int B, int D, int H, uint32_t* zbits, int64_t words_ld, void* workspace,
                                 size_t workspace_bytes, int flagged, qsae_stream_t stream);
/* Hidden-major dictionary for the sparse decoder: codes_rows[j][ceil(D/8)] uint32, 4-bit two's-complement
 * fields of S_j/2 (same S as qsae_pack_matryoshka; ABI 4: 2-bit fields, ceil(D/16) words per row, until ABI 3). */
int qsae_pack_matryoshka_rows(const float* w, const float* wm, int H, int D, uint32_t* codes_rows,
                              qsae_stream_t stream);
/* qsae_decode_matryoshka evaluated on the active units only (ascending walk over the row's z bits, one fmaf
 * per active unit and output column): the same chain as the dense kernel, whose z = 0 terms leave the
 * accumulator unchanged -- bit-identical levels and l0_counts.  Pays off below ~5 % active units.
 * D in {64,128,256,512,1024}; H % 32 == 0 (level boundaries need no alignment here). */
int qsae_decode_matryoshka_sparse(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, int n_bits,
                                  const int32_t* level_sizes, const uint32_t* codes_rows, const float* scale,
                                  const float* bias, int allow_bias, float* levels,
                                  unsigned long long* l0_counts, qsae_stream_t stream);
/* zbits[b][w] bit j = dense[b][32w+j] > thr -- the `latent > 0.5` binarisation applied to an
 * already materialised sigmoid latent (sae/quantized_matryoshka.py:97-99 when the decoder is
 * called directly, scripts/analysis/dynamic_analysis.py:51,66). */
int qsae_pack_bits_gt(const float* dense, int64_t ld, int B, int H, float thr, uint32_t* zbits,
                      int64_t words_ld, qsae_stream_t stream);

/* -- the dense decoders on the bf16 matrix pipe ------------------------------------------------------ */
/* qsae_decode_ternary_dense and qsae_decode_matryoshka contract a dictionary of {-1, 0, +1} with fp32 MFMA, at 1/16 of the
 * bf16 rate.  The *_split forms give the same sums from v_mfma_f32_32x32x16_bf16: the fp32 operand (the latent h, resp.
 * z_j * 2 scale_j) is split exactly into three bf16 terms (8 + 8 + 8 mantissa bits), the dictionary entries are exact in
 * bf16, so every product is exact and the three passes add into one fp32 accumulator.  What differs from the fp32 kernels
 * is the order of the fp32 accumulation roundings (results agree to ~1e-6 relative; both are graded at 1e-5 against the
 * fp64-accumulating oracle).  D == 512, H % 64 == 0 (matryoshka: level boundaries % 64 == 0); qsae_split_dec_supported() says
 * whether a shape qualifies.
 *   qsae_expand_codes_bf16: once per checkpoint, codes2 [D][ceil(H/16)] (as produced by qsae_pack_ternary or
 *     qsae_pack_matryoshka) -> the bf16 image the kernels stream (qsae_expand_codes_bf16_bytes() = 2 H D bytes, opaque);
 *   qsae_split_scale_bf16: once per checkpoint, scale [H] (qsae_pack_matryoshka) -> the three bf16 terms of 2 scale,
 *     s3 [3][H] bf16 (6 H bytes). */
int qsae_split_dec_supported(int B, int H, int D);
size_t qsae_expand_codes_bf16_bytes(int D, int H);
int qsae_expand_codes_bf16(const uint32_t* codes2, int D, int H, void* tq, qsae_stream_t stream);
/* recon[b][d] = sum_h h[b][h] * hard[d][h] (sae/ternary.py:41-52), h [B][ld] fp32. */
int qsae_decode_ternary_dense_split(const float* h, int64_t ld, int B, int H, const void* tq, int D, float* recon,
                                    qsae_stream_t stream);
int qsae_split_scale_bf16(const float* scale, int H, void* s3, qsae_stream_t stream);
/* levels / l0_counts as qsae_decode_matryoshka (sae/quantized_matryoshka.py:121-129). */
int qsae_decode_matryoshka_split(const uint32_t* zbits, int64_t words_ld, int B, int H, int D, int n_bits,
                                 const int32_t* level_sizes, const void* tq, const void* s3, const float* bias,
                                 int allow_bias, float* levels, unsigned long long* l0_counts, qsae_stream_t stream);

/* -- small elementwise steps (each operation rounded separately, as in the reference's ATen sequence) -------- */
/* out[i] = (residual[i] - recon[i]) * scale: the residual handed to the next stage of ResidualQuantizedSAE
 * (sae/residual_quantized.py:67, scale = 2).  out may alias residual. */
int qsae_residual_update(const float* residual, const float* recon, size_t n, float scale, float* out,
                         qsae_stream_t stream);
/* out[i] = pre[i] >= cutoff ? 1 : 0: the binary latent of BinaryLatentSAE, sigmoid(pre) >= 0.5 <=> pre >= 0xB43FFFFE
 * (sae/binary_latent.py:21-24). */
int qsae_threshold_ge(const float* pre, size_t n, float cutoff, float* out, qsae_stream_t stream);
/* out[b][d] = scale * acc[b][d] + bias[d] (bias may be NULL): the tail of binary_decoder.forward on an arbitrary dense
 * latent, reconstruction = quantization_step * latent.matmul(int_weights) + bias (sae/binary.py:38). */
int qsae_scale_bias_rows(const float* acc, int B, int D, float scale, const float* bias, float* out,
                         qsae_stream_t stream);

/* -- metric ------------------------------------------------------------------------------ */
/* *sum += sum_i (float)((recon[i]-x[i])^2) accumulated in double (device pointer; the caller
 * zeroes it).  scripts/analysis/dynamic_analysis.py:86-100. */
int qsae_sq_err_sum(const float* recon, const float* x, size_t n, double* sum, qsae_stream_t stream);

/* -- consumers of the sparse latent (scripts/analysis/dynamic_analysis.py:255-311, 314-440) ---------------- */
/* counts[idx[b][j]] += 1 for every entry with val[b][j] > 0 (val == NULL: every entry): mask.sum(dim=0) of the
 * reference's activation mask `latent > 0`, accumulated over calls (the caller zeroes counts[H], uint64). */
int qsae_activation_counts(const int32_t* idx, const float* val, int B, int k, int H, unsigned long long* counts,
                           qsae_stream_t stream);
/* The same for bit-packed masks (`latent > 0.5` of the matryoshka / residual encoders): counts[32 w + j] += bit j
 * of zbits[b][w]; nbits a multiple of 32. */
int qsae_activation_counts_bits(const uint32_t* zbits, int64_t words_ld, int B, int nbits, unsigned long long* counts,
                                qsae_stream_t stream);
/* coact[a][c] += 1 for every ordered pair of units active in the same row, diagonal included: the reference's
 * mask_int.t() @ mask_int (dynamic_analysis.py:296, 411), accumulated over calls into an int32 [H][ld] matrix. */
int qsae_coactivation_sparse(const int32_t* idx, const float* val, int B, int k, int H, int32_t* coact, int64_t ld,
                             qsae_stream_t stream);

/* -- activation quantizer of the binary datasets (src/quantized_sae/data/dataset.py:76-102) ----------------- */
/* bits[b][d*n + j] = bit j (LSB first, as 0.0 / 1.0) of the n-bit code of x[b][d]:
 *   is_signed = 0 (quantize):        int(round(clamp((x * sf) * 2 + 2^(n-1), 0, 2^n - 1)))
 *   is_signed = 1 (quantize_signed): int(round(clamp(x * sf, -2^(n-1), 2^(n-1) - 1))) & (2^n - 1)
 * with sf = scale_factor = 2^(n-1) / (gamma + 1e-5) rounded to fp32, round half to even. */
int qsae_quantize_bits(const float* x, int64_t ld, int B, int D, int n_bits, float scale_factor, int is_signed,
                       float* bits, qsae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QSAE_H */
