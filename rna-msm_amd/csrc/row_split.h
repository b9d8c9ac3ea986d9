// Row-range split of the tied row-attention logits (shared by the fp32 and the 16-bit row_logits kernels, and by
// rnamsm_row_logits_nsplit / _workspace_bytes, so every path sizes and sums the partial slabs identically).
#pragma once

namespace rnamsm {

struct RowSplit {
    int nsplit, rows_per_split;
};

// Deterministic function of the shape only.  Picks the split count whose block count best fills
// 256 CUs x 2 resident blocks, subject to >= 4 rows (8 K tiles) per split.
// tile / slots: output tile edge and resident block slots of the kernel that consumes the split (128 / 512 for the
// 128x128 kernels with two blocks per CU, 256 / 256 for the 256x256 16-bit kernel with one).
// max_rows (0 = no limit): upper bound on the rows of one split, i.e. on the length max_rows * 64 of an fp32 accumulation
// chain.  The exact-fp32 MFMA adds its products one after the other (an fmaf chain), so a slab of r rows is a chain of
// 64 r terms and its rounding error grows with r, while the slabs themselves are added pairwise-like by K5.  Measured
// against an fp64 truth at M=256 x L=512 (tests/analysis/error_sources.py): chains of 4096 / 2048 / 1024 / 512 / 256 terms leave
// the 10-layer embedding 1.26e-4 / 7.1e-5 / 4.1e-5 / 3.2e-5 / 3.1e-5 from the truth -- the reference's own CPU arithmetic
// (blocked sgemm) sits at 3.2e-5.  The fp32 kernel therefore keeps every chain at 8 rows (512 terms).
// slab_penalty: what one more partial slab costs in units of fill efficiency (each slab is an H*C*C fp32 write here and a
// read in K5, and one more epilogue per output tile): 0.002 breaks ties only; the persistent 256x256 bf16 kernel, whose launch
// is short enough for that traffic to show, passes 0.01 (M=256 x L=512: 5 slabs of 51 rows instead of 16 of 16).
inline RowSplit choose_row_split(int R, int C, int H, int tile = 128, int slots = 512, int max_rows = 0, double slab_penalty = 0.002) {
    const long tiles = (long)((C + tile - 1) / tile) * ((C + tile - 1) / tile) * H;
    const int min_ns = max_rows > 0 ? (R + max_rows - 1) / max_rows : 1;
    int best_ns = min_ns;
    double best_score = -1.0;
    int max_ns = R / 4 > 1 ? (R / 4 < 64 ? R / 4 : 64) : 1;
    if (max_ns < min_ns) max_ns = min_ns;
    for (int ns = min_ns; ns <= max_ns; ++ns) {
        const int rps = (R + ns - 1) / ns;
        const int real_ns = (R + rps - 1) / rps;
        const long blocks = tiles * real_ns;
        const long rounds = (blocks + slots - 1) / slots;
        double score = (double)blocks / (double)(rounds * slots);   // fill efficiency of the last round
        score -= slab_penalty * real_ns;                             // prefer fewer partial slabs
        if (score > best_score + 1e-9) {
            best_score = score;
            best_ns = real_ns;
        }
    }
    RowSplit s;
    s.rows_per_split = (R + best_ns - 1) / best_ns;
    s.nsplit = (R + s.rows_per_split - 1) / s.rows_per_split;
    return s;
}

// fp32 row_logits: the kernel restarts its accumulators every CHAIN_ROWS rows (<= 512-term chains, see above) and adds
// the chain sums in order, so a slab may cover MAX_ROWS rows (four chains) without lengthening any chain.
constexpr int ROW_LOGITS_F32_CHAIN_ROWS = 8;
constexpr int ROW_LOGITS_F32_MAX_ROWS = 32;

}  // namespace rnamsm
