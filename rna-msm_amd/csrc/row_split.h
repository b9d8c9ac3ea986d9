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
inline RowSplit choose_row_split(int R, int C, int H, int tile = 128, int slots = 512) {
    const long tiles = (long)((C + tile - 1) / tile) * ((C + tile - 1) / tile) * H;
    int best_ns = 1;
    double best_score = -1.0;
    const int max_ns = R / 4 > 1 ? (R / 4 < 64 ? R / 4 : 64) : 1;
    for (int ns = 1; ns <= max_ns; ++ns) {
        const int rps = (R + ns - 1) / ns;
        const int real_ns = (R + rps - 1) / rps;
        const long blocks = tiles * real_ns;
        const long rounds = (blocks + slots - 1) / slots;
        double score = (double)blocks / (double)(rounds * slots);   // fill efficiency of the last round
        score -= 0.002 * real_ns;                                    // prefer fewer partial slabs on ties
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

}  // namespace rnamsm
