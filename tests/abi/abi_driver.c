/* Host-side driver of the C ABI's argument-validation layer, built with AddressSanitizer + UndefinedBehaviorSanitizer
 * against a HOST-ONLY build of librnamsm_hip (`make -C rna-msm_amd/csrc check-asan`; SURVEY.md §5: sanitizers run on the
 * CPU build only -- GPU ASan is not available on this pool).  Every call here must be REFUSED by an entry point's checks
 * (negative status + a message) or be a pure host function: nothing reaches a kernel launch, so no GPU is needed.
 * What the sanitizers watch: the error-text formatting (thread-local 512-byte buffer), the workspace / split arithmetic
 * (size_t and int overflows at extreme shapes), the parameter-name parsing, the timing bookkeeping. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rnamsm.h"

static int checks = 0, failures = 0;
#define EXPECT(cond)                                                                      \
    do {                                                                                  \
        ++checks;                                                                         \
        if (!(cond)) {                                                                    \
            ++failures;                                                                   \
            fprintf(stderr, "FAILED %s:%d: %s  [last error: %s]\n", __FILE__, __LINE__, #cond, rnamsm_last_error()); \
        }                                                                                 \
    } while (0)
#define REFUSED(call) EXPECT((call) < 0 && strlen(rnamsm_last_error()) > 0 && strlen(rnamsm_last_error()) < 512)

int main(void) {
    /* "device" buffers: host memory, 256-byte aligned -- never dereferenced, only validated */
    float* buf = (float*)aligned_alloc(256, 1 << 16);
    uint8_t* bytes = (uint8_t*)buf;
    uint16_t* halves = (uint16_t*)buf;
    int64_t* toks = (int64_t*)buf;
    int* ints = (int*)buf;
    float* odd = (float*)((char*)buf + 4);          /* misaligned on purpose */
    rnamsm_model_dims dims = {10, 768, 12, 3072, 12, 1026, 1, 1e-5f};
    const float* weights[7 + 10 * 18] = {0};

    EXPECT(rnamsm_version() == RNAMSM_VERSION);
    EXPECT(rnamsm_device_count() >= 0);

    /* pure host functions, at ordinary and extreme shapes (integer arithmetic under UBSan) */
    EXPECT(rnamsm_row_logits_nsplit(256, 512, 12) == 8);      /* 32-row slabs, accumulated as four 512-term fp32 chains */
    EXPECT(rnamsm_row_logits_nsplit(0, 512, 12) == 0 && rnamsm_row_logits_nsplit(-5, -5, -5) == 0);
    EXPECT(rnamsm_row_logits_nsplit(1024, 1025, 12) >= 1);
    EXPECT(rnamsm_row_logits_workspace_bytes(1024, 1025, 12) >= (size_t)12 * 1025 * 1025 * 4);
    EXPECT(rnamsm_row_logits16_nsplit(1024, 1024, 12, 3) >= 1 && rnamsm_row_logits16_nsplit(1, 2, 1, 1) >= 1);
    EXPECT(rnamsm_row_logits16_workspace_bytes(1024, 1024, 12) >= rnamsm_row_logits_workspace_bytes(1024, 1024, 12) / 64);
    EXPECT(rnamsm_row_chunks(6, 21, 42) == 3 && rnamsm_row_chunks(6, 21, 126) == 0 && rnamsm_row_chunks(6, 21, 1) == 6);
    EXPECT(rnamsm_row_chunks(1024, 1024, 2147483647) == 0 && rnamsm_row_chunks(0, 0, 0) == 0);
    EXPECT(rnamsm_forward_workspace_bytes(&dims, 1024, 1024, 0, 0) > (size_t)1024 * 1024 * 768 * 4 * 6);
    EXPECT(rnamsm_forward_workspace_bytes(&dims, 1024, 1024, 1, 1) > rnamsm_forward_workspace_bytes(&dims, 1024, 1024, 0, 0));
    EXPECT(rnamsm_forward_workspace_bytes(NULL, 4, 4, 0, 0) == 0 && rnamsm_forward_workspace_bytes(&dims, 0, 4, 0, 0) == 0);
    EXPECT(rnamsm_contact_head_workspace_bytes(1025, 120) > 0);
    EXPECT(rnamsm_greedy_select_workspace_bytes(100000, 1024, 1024) >= (size_t)1023 * 100000 * 2);
    EXPECT(rnamsm_greedy_select_workspace_bytes(0, 1, 1) == 0);

    /* every compute entry point must refuse null / misaligned / out-of-range arguments with a message */
    REFUSED(rnamsm_embed_ln(NULL, buf, buf, buf, buf, buf, buf, 4, 4, 768, 12, 1026, 1, 1e-5f, ints, NULL));
    REFUSED(rnamsm_embed_ln(toks, buf, buf, buf, buf, buf, buf, 1025, 4, 768, 12, 1026, 1, 1e-5f, ints, NULL));
    REFUSED(rnamsm_layernorm(NULL, buf, buf, buf, 8, 768, 1e-5f, NULL));
    REFUSED(rnamsm_layernorm(buf, buf, buf, buf, 8, 770, 1e-5f, NULL));
    REFUSED(rnamsm_layernorm(buf, buf, buf, buf, 8, 1 << 20, 1e-5f, NULL));
    /* K1 folded */
    REFUSED(rnamsm_row_partials(NULL, buf, 8, 768, NULL));
    REFUSED(rnamsm_row_partials(buf, buf, 8, 770, NULL));
    REFUSED(rnamsm_row_partials(buf, buf, 0, 768, NULL));
    REFUSED(rnamsm_gemm_residual_stats(buf, 64, buf, buf, NULL, 128, buf, 128, 8, 128, 64, buf, 8, 0, NULL));   /* no residual */
    REFUSED(rnamsm_gemm_residual_stats(buf, 64, buf, buf, buf, 128, buf, 128, 8, 128, 64, NULL, 8, 0, NULL));   /* no partials */
    REFUSED(rnamsm_gemm_residual_stats(buf, 64, buf, buf, buf, 128, buf, 128, 8, 100, 64, buf, 8, 0, NULL));    /* N % 128 */
    REFUSED(rnamsm_gemm_residual_stats(buf, 64, buf, buf, buf, 128, buf, 128, 8, 128, 64, buf, 8, 2, NULL));    /* dtype */
    REFUSED(rnamsm_gemm_residual_stats(buf, 64, buf, buf, buf, 128, buf, 128, 8, 128, 64, buf, 4, 0, NULL));          /* partials_ld < M */
    REFUSED(rnamsm_gemm16_lnfold(NULL, NULL, 768, halves, NULL, buf, buf, buf, halves, NULL, 2304, 4096, 2304, 768, 0, 1.f, 0, 1, 0, NULL));
    REFUSED(rnamsm_gemm16_lnfold(halves, NULL, 768, halves, NULL, buf, buf, buf, halves, NULL, 2304, 100, 2304, 768, 0, 1.f, 0, 1, 0, NULL));    /* M < 2048 */
    REFUSED(rnamsm_gemm16_lnfold(halves, NULL, 768, halves, NULL, buf, buf, buf, halves, NULL, 2304, 4096, 2304, 768, 0, 1.f, 0, 3, 0, NULL));   /* split 3 without lo */
    REFUSED(rnamsm_gemm16_residual_stats(halves, NULL, 768, halves, NULL, buf, NULL, 768, 4096, 768, 768, 1, 0, halves, NULL, 768, buf, 4096, NULL));
    REFUSED(rnamsm_gemm16_residual_stats(halves, NULL, 768, halves, NULL, buf, buf, 768, 4096, 768, 768, 1, 0, halves, NULL, 768, buf, 100, NULL)); /* partials_ld < M */
    REFUSED(rnamsm_row_stats_from_partials(NULL, 8, 8, 768, 1e-5f, buf, NULL, NULL));
    REFUSED(rnamsm_row_stats_from_partials(buf, 4, 8, 768, 1e-5f, buf, NULL, NULL));          /* partials_ld < M */
    REFUSED(rnamsm_row_stats_from_partials(buf, 8, 8, 770, 1e-5f, buf, NULL, NULL));          /* K % 32 */
    REFUSED(rnamsm_ln_fold_weights(NULL, buf, buf, buf, buf, buf, buf, 128, 64, NULL));
    REFUSED(rnamsm_ln_fold_weights(buf, buf, buf, buf, buf, buf, buf, 0, 64, NULL));
    REFUSED(rnamsm_gemm_lnfold(NULL, 64, buf, buf, buf, 1e-5f, NULL, NULL, buf, 128, 8, 128, 64, 0, 1.f, 0, 0, NULL));
    REFUSED(rnamsm_gemm_lnfold(buf, 64, buf, buf, buf, 1e-5f, NULL, NULL, buf, 128, 8, 100, 64, 0, 1.f, 0, 0, NULL));   /* N % 128 */
    REFUSED(rnamsm_gemm_lnfold(buf, 64, buf, buf, buf, 1e-5f, NULL, NULL, buf, 128, 8, 128, 64, 0, 1.f, 6, 0, NULL));    /* scale_cols % 4 */
    REFUSED(rnamsm_gemm_lnfold(buf, 64, buf, buf, buf, 1e-5f, NULL, NULL, buf, 128, 8, 128, 64, 7, 1.f, 0, 0, NULL));    /* activation */
    REFUSED(rnamsm_gemm_lnfold(buf, 64, buf, buf, buf, 1e-5f, NULL, NULL, buf, 128, 8, 128, 64, 0, 1.f, 0, 1, NULL));    /* dtype */
    REFUSED(rnamsm_gemm_lnfold(buf, 64, buf, buf, buf, -1.f, NULL, NULL, buf, 128, 8, 128, 64, 0, 1.f, 0, 0, NULL));     /* eps */
    REFUSED(rnamsm_layernorm_split(buf, buf, buf, NULL, NULL, 8, 768, 1e-5f, 0, NULL));
    REFUSED(rnamsm_gemm_bias_act_res(NULL, 0, NULL, NULL, NULL, 0, NULL, 0, 4, 128, 32, 0, 1.f, 0, NULL, 0, NULL));
    REFUSED(rnamsm_gemm_bias_act_res(buf, 32, buf, NULL, NULL, 0, buf, 100, 4, 100, 32, 0, 1.f, 0, NULL, 0, NULL));
    REFUSED(rnamsm_gemm_bias_act_res(odd, 32, buf, NULL, NULL, 0, buf, 128, 4, 128, 32, 0, 1.f, 0, NULL, 0, NULL));
    REFUSED(rnamsm_gemm_bias_act_res(buf, 32, buf, NULL, NULL, 0, buf, 128, 4, 128, 32, 7, 1.f, 0, NULL, 0, NULL));
    REFUSED(rnamsm_gemm_bias_act_res(buf, 32, buf, NULL, NULL, 0, buf, 128, 4, 128, 32, 0, 1.f, 0, NULL, 1, NULL));
    REFUSED(rnamsm_split_bf16(NULL, halves, halves, 16, 0, NULL));
    REFUSED(rnamsm_split_bf16(buf, halves, halves, 16, 9, NULL));
    REFUSED(rnamsm_gemm_bf16(NULL, 32, NULL, NULL, NULL, NULL, 0, NULL, 0, 4, 128, 32, 0, 1.f, 0, 3, 0, NULL, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_gemm_bf16(buf, 32, halves, halves, NULL, NULL, 0, buf, 128, 4, 128, 32, 0, 1.f, 0, 2, 0, NULL, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_row_logits(NULL, buf, 2304, buf, 4, 4, 12, 64, 0, NULL));
    REFUSED(rnamsm_row_logits(buf, buf, 2304, buf, 4, 4, 12, 32, 0, NULL));
    REFUSED(rnamsm_row_logits(buf, buf, 2304, buf, 2000, 4, 12, 64, 0, NULL));
    REFUSED(rnamsm_row_logits(buf, buf, 2304, buf, 4, 4, 12, 64, 1, NULL));
    REFUSED(rnamsm_row_logits_chunked(buf, buf, 2304, buf, 4, 4, 12, 64, 0, 0, NULL));
    REFUSED(rnamsm_softmax_rows(NULL, 1, buf, 12, 4, NULL, NULL));
    REFUSED(rnamsm_softmax_rows(buf, 0, buf, 12, 4, NULL, NULL));
    REFUSED(rnamsm_softmax_rows(buf, 1, buf, 12, 5000, NULL, NULL));
    REFUSED(rnamsm_softmax_rows_chunked(buf, 2, buf, 12, 4, NULL, 2, NULL));
    REFUSED(rnamsm_softmax_rows_planes(buf, 1, buf, NULL, NULL, 64, 4096.f, 12, 4, NULL, 0, NULL));
    REFUSED(rnamsm_softmax_rows_planes(buf, 1, buf, halves, halves, 65, 4096.f, 12, 4, NULL, 0, NULL));
    REFUSED(rnamsm_softmax_rows_planes(buf, 1, buf, halves, halves, 64, -1.f, 12, 4, NULL, 0, NULL));
    REFUSED(rnamsm_row_apply(NULL, buf, 2304, buf, 768, 4, 4, 12, 64, NULL, NULL, 0, 0, NULL));
    REFUSED(rnamsm_row_apply(buf, buf, 2303, buf, 768, 4, 4, 12, 64, NULL, NULL, 0, 0, NULL));
    REFUSED(rnamsm_col_attn_fused(buf, buf, buf, 64, buf, 64, 4, 4, 1, 32, NULL, NULL, NULL, 0, 0, NULL));
    REFUSED(rnamsm_col_attn_fused(NULL, buf, buf, 2304, buf, 768, 4, 4, 12, 64, NULL, NULL, NULL, 0, 0, NULL));
    REFUSED(rnamsm_col_attn_fused(buf, buf, buf, 2304, buf, 768, 5000, 4, 12, 64, NULL, NULL, NULL, 0, 0, NULL));
    REFUSED(rnamsm_col_attn_fused_queries(buf, buf, buf, 2304, buf, 768, 4, 4, 12, 64, 0, NULL, 0, NULL));
    REFUSED(rnamsm_col_attn_fused_queries(buf, buf, buf, 2304, buf, 768, 4, 4, 12, 64, 5, NULL, 0, NULL));
    REFUSED(rnamsm_col_attn_probs(NULL, buf, 2304, buf, 4, 4, 12, 64, NULL, 1.f, 0, NULL));
    REFUSED(rnamsm_col_attn_probs(buf, buf, 2304, buf, 4, 4, 12, 32, NULL, 1.f, 0, NULL));
    REFUSED(rnamsm_col_attn_probs(buf, buf, 2304, buf, 1025, 4, 12, 64, NULL, 1.f, 0, NULL));
    REFUSED(rnamsm_col_attn_probs(buf, buf, 100, buf, 4, 4, 12, 64, NULL, 1.f, 0, NULL));
    REFUSED(rnamsm_col_attn_probs(buf, buf, 2304, buf, 4, 4, 12, 64, NULL, 1.f, 1, NULL));
    REFUSED(rnamsm_col_attn_probs16(halves, halves, halves, NULL, 2304, buf, 4, 4, 12, 64, NULL, 0, 1.f, NULL));
    REFUSED(rnamsm_col_attn_probs16(halves, NULL, halves, NULL, 2304, buf, 4, 4, 12, 64, NULL, 2, 1.f, NULL));
    REFUSED(rnamsm_row_logits16(NULL, NULL, halves, NULL, 2304, buf, 4, 4, 12, 64, 1.f, 0, NULL));
    REFUSED(rnamsm_row_apply16(NULL, NULL, 64, halves, NULL, 2304, buf, 768, 4, 4, 12, 64, 1.f, NULL, NULL, 0, NULL));
    REFUSED(rnamsm_col_attn16(NULL, NULL, halves, NULL, halves, NULL, 2304, buf, 768, 4, 4, 12, 64, 1.f, NULL, NULL, NULL, 0, NULL));
    REFUSED(rnamsm_zero_plane_rows(NULL, NULL, bytes, 4, 768, 2304, NULL));
    REFUSED(rnamsm_zero_plane_rows(halves, NULL, bytes, 4, 770, 2304, NULL));
    REFUSED(rnamsm_pad_mask(NULL, bytes, 16, 1, NULL));
    REFUSED(rnamsm_head_mean(NULL, buf, 12, 16, NULL));
    REFUSED(rnamsm_head_mean(buf, buf, 0, 16, NULL));
    REFUSED(rnamsm_pack_outputs(NULL, buf, buf, buf, 4, 768, 10, 12, NULL));
    REFUSED(rnamsm_pack_outputs(buf, buf, buf, buf, 4, 6, 10, 12, NULL));                   /* D % 4 != 0: vector copy of the embedding rows */
    REFUSED(rnamsm_pack_outputs(buf + 1, buf, buf, buf, 4, 768, 10, 12, NULL));             /* x_final not 16-byte aligned */
    REFUSED(rnamsm_pack_outputs(buf, buf, buf + 1, buf, 4, 768, 10, 12, NULL));             /* emb not 16-byte aligned */
    REFUSED(rnamsm_contact_head(NULL, buf, buf, buf, buf, 1 << 16, 4, 120, NULL));
    REFUSED(rnamsm_contact_head(buf, buf, buf, buf, buf, 0, 400, 120, NULL));
    REFUSED(rnamsm_greedy_select(NULL, 8, 8, 4, 0, ints, buf, 1 << 16, NULL));
    REFUSED(rnamsm_greedy_select(bytes, 8, 8, 9, 0, ints, buf, 1 << 16, NULL));
    REFUSED(rnamsm_greedy_select(bytes, 8, 8, 4, 0, ints, buf, 8, NULL));
    REFUSED(rnamsm_greedy_select(bytes, 8, 70000, 4, 0, ints, buf, 1 << 16, NULL));
    REFUSED(rnamsm_msa_weights(NULL, 8, 8, 0.2, (double*)buf, NULL));
    REFUSED(rnamsm_msa_weights(bytes, 0, 8, 0.2, (double*)buf, NULL));
    REFUSED(rnamsm_msa_weights(bytes, 8, 40000, 0.2, (double*)buf, NULL));
    REFUSED(rnamsm_forward(NULL, weights, toks, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 0, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_forward(&dims, weights, toks, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 9, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_forward(&dims, weights, toks, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 3, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_forward(&dims, weights, toks, 1025, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 0, NULL, NULL, NULL, NULL));
    EXPECT(strstr(rnamsm_last_error(), "maximum MSA depth of 1024") != NULL);             /* model.py:355-359 */
    REFUSED(rnamsm_forward(&dims, weights, toks, 4, 1, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 0, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_forward(&dims, weights, toks, 4, 4000, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 0, NULL, NULL, NULL, NULL));
    REFUSED(rnamsm_forward(&dims, weights, toks, 64, 128, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 0, NULL, NULL, NULL, NULL));
    EXPECT(strstr(rnamsm_last_error(), "workspace too small") != NULL);
    /* the batched form: same validation, plus the batch count and the 31-bit token range */
    EXPECT(rnamsm_forward_batch_workspace_bytes(&dims, 8, 16, 64) > rnamsm_forward_batch_workspace_bytes(&dims, 1, 16, 64));
    EXPECT(rnamsm_forward_batch_workspace_bytes(NULL, 2, 4, 4) == 0 && rnamsm_forward_batch_workspace_bytes(&dims, 0, 4, 4) == 0);
    REFUSED(rnamsm_forward_batch(NULL, weights, toks, 2, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 0, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 2, 1025, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 2, 4, 1, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 4096, 1024, 1024, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 2, 4, 4, NULL, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 2, 64, 128, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 0, NULL, NULL));
    EXPECT(strstr(rnamsm_last_error(), "forward_batch: workspace too small") != NULL);
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 2, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 9, NULL, NULL));
    REFUSED(rnamsm_forward_batch(&dims, weights, toks, 2, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, NULL, NULL, 1, NULL, NULL));
    EXPECT(strstr(rnamsm_last_error(), "need weight_planes") != NULL);
    {   /* the token-packed driver: refusals of the table itself, all before any launch */
        const int two_classes[4] = {64, 128, 8, 40};          /* 8192 tokens (LayerNorm folded from 4096) next to 320 */
        const int one_class[4] = {8, 40, 6, 33};
        const int too_deep[2] = {1025, 8};
        EXPECT(rnamsm_forward_packed_workspace_bytes(&dims, 2, two_classes) > 0 && rnamsm_forward_packed_workspace_bytes(&dims, 1, too_deep) == 0);
        REFUSED(rnamsm_forward_packed(&dims, weights, toks, 1, too_deep, buf, (size_t)1 << 40, buf, buf, buf, buf, ints, NULL, 0, NULL, NULL));
        REFUSED(rnamsm_forward_packed(&dims, weights, toks, 2, one_class, buf, 1 << 10, buf, buf, buf, buf, ints, NULL, 0, NULL, NULL));   /* workspace too small */
        REFUSED(rnamsm_forward_packed(&dims, weights, toks, 2, one_class, buf, (size_t)1 << 40, buf, buf, buf, buf, ints, NULL, 1, NULL, NULL)); /* 16-bit mode without planes */
        /* a table mixing the two LayerNorm-fold classes (by-shape rule, folded tables given) is refused, not run unfolded */
        REFUSED(rnamsm_forward_packed(&dims, weights, toks, 2, two_classes, buf, (size_t)1 << 40, buf, buf, buf, buf, ints, weights, 0, NULL, NULL));
        EXPECT(strstr(rnamsm_last_error(), "mixes alignments below and from 4096 tokens") != NULL);
    }
    {
        rnamsm_model_dims bad = dims;
        bad.embed_dim = 700;
        REFUSED(rnamsm_forward(&bad, weights, toks, 4, 4, buf, 1 << 16, buf, buf, buf, buf, ints, 0, 0, 1, 0, NULL, NULL, NULL, NULL));
    }

    /* parameter parsing and timing bookkeeping */
    {
        char longname[2048];
        memset(longname, 'x', sizeof(longname) - 1);
        longname[sizeof(longname) - 1] = 0;
        REFUSED(rnamsm_set_param(longname, 1));                     /* 2 KB name into the 512-byte error buffer */
        REFUSED(rnamsm_set_param(NULL, 1));
        REFUSED(rnamsm_set_param("gemm_group", 1000));
        EXPECT(rnamsm_get_param(longname) == -1 && rnamsm_get_param(NULL) == -1);
        EXPECT(rnamsm_set_param("gemm_tile", 2) == RNAMSM_OK && rnamsm_get_param("gemm_tile") == 2);
        EXPECT(rnamsm_set_param("gemm_tile", 0) == RNAMSM_OK);
    }
    {
        const char* name = NULL;
        long long n = -1;
        double ms = -1, fl = -1, by = -1;
        rnamsm_timing_reset();
        EXPECT(rnamsm_timing_enable(0) == RNAMSM_OK);
        int cats = rnamsm_timing_collect();
        EXPECT(cats > 0);
        for (int c = 0; c < cats; ++c) EXPECT(rnamsm_timing_get(c, &name, &n, &ms, &fl, &by) == RNAMSM_OK && name && n == 0);
        REFUSED(rnamsm_timing_get(cats, &name, &n, &ms, &fl, &by));
        REFUSED(rnamsm_timing_get(-1, NULL, NULL, NULL, NULL, NULL));
    }
    free(buf);
    printf("ABI driver: %d checks, %d failed\n", checks, failures);
    return failures ? 1 : 0;
}
