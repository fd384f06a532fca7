// Engines and launch-size regimes AS DATA: the one place that names which GEMM engine exists under which number and at which padded
// row bound a launch changes its route.  gemm_dispatch.hip picks engines from here, api.hip reads its bounds from here,
// tools/gen_regime_doc.py prints REGIME_TABLE into include/mmscore.h's comment (tests/test_abi.py checks that the header carries exactly
// that text), and the kernel tests name engines by these values through mms_dbg_gemm (lib.py mirrors the enum).
#pragma once
#include <cstdint>

// GemmParams::engine.  ENG_AUTO = the per-shape choice of a forward (gemm_dispatch.hip pick_engine); every other value names ONE engine
// (kernel tests, tools, and the two places in api.hip that must have the skinny kernel).  Values are part of the test hooks' contract.
enum Engine : int {
    ENG_AUTO = 0,
    ENG_TILE_128 = 1,           // gemm_tile.hip: register-staged 128 x 128 tile (the only tile for N % 256 != 0, and the three-pass tile of precision mode 3)
    ENG_TILE_DMA = 3,           // gemm_tile.hip: 128 x 256 tile, LDS-DMA double buffered, one workgroup per CU (launches of no more workgroups than CUs)
    ENG_TILE = 4,               // gemm_tile.hip: register-staged 128 x 256 tile, the default below PP_ROWS
    ENG_SKINNY = 5,             // gemm_skinny.hip: <= 128 rows, one workgroup per 16 output columns, K sliced over its waves (GemmParams::wave_k_slices)
    ENG_TILE_256 = 16,          // gemm_tile.hip: 256 x 256 / 16 waves (kept for the kernel tests; no forward selects it since round 4)
    ENG_PP = 20,                // gemm_pp.hip: 256 x 256 ping-pong phases, one tile per workgroup
    ENG_PP_PERSIST = 26,        // gemm_pp.hip: the same, persistent workgroups (XCD-aware tile walk): launches of >= PP_ROWS rows
    ENG_PPW = 27,               // gemm_ppw.hip: 256 x 128 three-pass ping-pong (precision mode 3, >= PP_ROWS rows)
    ENG_DW = 28,                // lab build: gemm_dw.hip (128 x 256, two 4-wave workgroups per CU) -- measured and shelved
    ENG_SKINNY_K4 = 54,         // kernel tests: the skinny kernel with 4 wave-level K slices
    ENG_SKINNY_PARTS = 55,      // gemm_skinny.hip with the K slices dealt to single-wave WORKGROUPS: GemmParams::k_splits fp32 partials (the split-K contract of the tiles)
    ENG_SKINNY_K8 = 58,         // kernel tests: 8 wave-level K slices
    ENG_DIAG_BASE = 200,        // lab build + MMS_GEMM_DIAG: 201 .. 232, 264 = timing-only instantiations of gemm_pp.hip (WRONG results by design)
};
constexpr bool engine_is_named(int e) {
    return e == ENG_TILE_128 || e == ENG_TILE_DMA || e == ENG_TILE || e == ENG_SKINNY || e == ENG_TILE_256 || e == ENG_PP || e == ENG_PP_PERSIST ||
           e == ENG_PPW || e == ENG_DW || e == ENG_SKINNY_K4 || e == ENG_SKINNY_PARTS || e == ENG_SKINNY_K8;
}

// Padded row bounds (pairs x sequence length of the stream; the live count stays on the device).  The measurements behind each value are cited
// where api.hip / gemm_dispatch.hip use it.
constexpr int64_t SKINNY_ROWS_DEFAULT = 128;
constexpr int64_t TINY_ROWS_DEFAULT = 1024;
constexpr int64_t FUSED_ATTN_ROWS_DEFAULT = 1024;
constexpr int64_t TALL_ROWS = 4096;
constexpr int64_t SPLITK_HALF_ROWS = 4096;
constexpr int64_t PP_WIDE_ROWS_DEFAULT = 5120;
constexpr int64_t SPLITK_ROWS = 8192;
constexpr int64_t SPLITK2_LO_ROWS = 11264;
constexpr int64_t PP_ROWS_DEFAULT = 16384;
constexpr int64_t LNF_ROWS_DEFAULT = 98304;
constexpr int64_t ENS_LANE_ROWS_DEFAULT = 200000;
constexpr int64_t LANE_ROWS_DEFAULT = 400000;
constexpr int KSPLIT_MAX = 8;

struct RegimeBound {
    const char* name;
    const char* cmp;            // how the bound is applied to a launch's padded rows
    int64_t rows;
    bool numerical;             // true: crossing it changes a summation order or a route (results move by fp32 round-off); false: a speed choice, tested bit-identical
    const char* what;
};
// One line per bound, ascending.  include/mmscore.h carries this table verbatim (tools/gen_regime_doc.py).
constexpr RegimeBound REGIME_TABLE[] = {
    {"SKINNY_ROWS", "<=", SKINNY_ROWS_DEFAULT, false,
     "every projection on the skinny kernel (gemm_skinny.hip), K sliced exactly as the tile route of the same projection slices it"},
    {"TINY_ROWS", "<", TINY_ROWS_DEFAULT, true,
     "wide projections (N >= 1536, K = 768: QKV, K | V, FFN-up): K in 4 slices, summed in fixed order (k_splitk_reduce)"},
    {"FUSED_ATTN_ROWS", ">=", FUSED_ATTN_ROWS_DEFAULT, true,
     "mms_config.fuse_attention: a stream's QKV projection + attention in one kernel (qkv_attn.hip); 1 = the two-kernel route's arithmetic (bit-identical to it), "
     "2 = split-bf16 attention over 16-query tiles of a packed sub-tile: a pair's logits depend on its place in the launch by fp32 round-off (<= 1e-4 relative)"},
    {"TALL_ROWS", "<", TALL_ROWS, true,
     "the long-K projections in front of the encoder (K >= 2048, N = 768: zk kdd_conv1 as im2col, kdd_conv2 / visn_fc / featureemb over the box rows): K in 8 slices"},
    {"SPLITK_HALF_ROWS", ">=", SPLITK_HALF_ROWS, true,
     "the LayerNorm-followed K >= 2048 projections (FFN-down) of the split-K regime: 4 K slices instead of 8"},
    {"PP_WIDE_ROWS", ">=", PP_WIDE_ROWS_DEFAULT, false,
     "wide projections (N >= 1536) on the persistent ping-pong engine instead of the 128 x 256 tiles (same contraction order per element)"},
    {"SPLITK_ROWS", "<", SPLITK_ROWS, true,
     "the N = 768 projections that a LayerNorm follows (attention output, FFN-down): K in 4 (K = 768) / 8 or 4 (K >= 2048) slices, summed by the LayerNorm kernel"},
    {"SPLITK2_LO_ROWS", ">=", SPLITK2_LO_ROWS, true,
     "FFN-down (K = 3072) of launches below PP_ROWS: 2 K slices (one pass between SPLITK_ROWS and here)"},
    {"PP_ROWS", ">=", PP_ROWS_DEFAULT, false,
     "persistent ping-pong engines for every projection (same contraction order per element as the tiles: bit-identical, tested)"},
    {"LNF_ROWS", ">=", LNF_ROWS_DEFAULT, true,
     "mms_config.fuse_layernorm: bias + residual + LayerNorm in the GEMM epilogue (one-pass variance) -- not in a call that runs on launch lanes, see LANE_ROWS / ENS_LANE_ROWS"},
    {"ENS_LANE_ROWS", "<", ENS_LANE_ROWS_DEFAULT, true,
     "mms_score_ensemble: the three members of a launch wave (rows of its longest member; ~5000 pairs) side by side on three streams, every LayerNorm by its own kernel"},
    {"LANE_ROWS", "<", LANE_ROWS_DEFAULT, true,
     "lxmert calls (pairs x (text_len + 10) rows; ~12 500 pairs): the two streams' launch chains on two lanes, every LayerNorm by its own kernel (with or without per-launch timing)"},
};
constexpr int REGIME_COUNT = sizeof(REGIME_TABLE) / sizeof(REGIME_TABLE[0]);
