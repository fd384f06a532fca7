// GEMM dispatch: which tile engine runs a given (M, N, K, passes) contraction.
//
// Product library (libmmscore.so): per-shape choice between the kernels a forward can launch -- the 256x256 persistent ping-pong
// engine (gemm_pp.hip) for the big encoder GEMMs, its 256x128 three-pass cut (gemm_ppw.hip) for precision mode 3, and the
// register-staged tiles of gemm_tile.hip for the small ones (CLS-only last block, poolers, heads, label text).  No environment
// variable is read here and there is no process-global state: a caller that wants ONE engine names it per launch (GemmParams::engine, enum Engine of regimes.h;
// the kernel tests do, through mms_dbg_gemm).
//
// Lab library (libmmscore_lab.so, `make lab`, -DMMS_LAB): additionally honours MMS_GEMM_VARIANT, reaches the LDS-DMA double-buffer
// tile of gemm_tile.hip, the measured-and-shelved engines (gemm_dw.hip: ENG_DW; gemm_mx.hip's fp16 + MX-fp8 "1.5 pass" kernel) and
// -- with MMS_GEMM_DIAG -- the timing-only DIAG instantiations of gemm_pp.hip, which compute WRONG results by design.  None of that code is in the product binary.  (The round-1 A/B kernels gemm.hip / gemm_ring.hip were removed in round 3:
// their measurements stay in profiles/r01c_gemm_variants.txt.)
#include <atomic>
#include <cstdlib>

#include "kernels.h"

int device_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 8;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) ? prop.multiProcessorCount / 8 * 8 : 8;
    cache[dev].store(v, std::memory_order_relaxed);      // idempotent: every thread computes the same value
    return v;
}

int pp_rows() {
#ifdef MMS_LAB
    static const int v = getenv("MMS_PP_ROWS") ? atoi(getenv("MMS_PP_ROWS")) : (int)PP_ROWS_DEFAULT;
    return v;
#else
    return (int)PP_ROWS_DEFAULT;
#endif
}

int pp_wide_rows() {
#ifdef MMS_LAB
    static const int v = getenv("MMS_PP_WIDE_ROWS") ? atoi(getenv("MMS_PP_WIDE_ROWS")) : (int)PP_WIDE_ROWS_DEFAULT;
    return v;
#else
    return (int)PP_WIDE_ROWS_DEFAULT;
#endif
}

// ---- regime -> engine: the per-shape choice of a forward (GemmParams::engine == ENG_AUTO), a pure function of the launch's shape --------------------------
// Two passes / one pass (precision modes 2, 1 and the single-plane launches of mode 4's fallbacks), profiles/r01c_gemm_variants.txt:
//   rows >= PP_ROWS, N % 256 == 0                        ENG_PP_PERSIST
//   rows >= PP_WIDE_ROWS, N >= 1536, no split-K          ENG_PP_PERSIST   -- the engines contract K in the same order per element (bit-identical,
//       tests/test_kernels_gpu.py), so this is a speed choice alone.  At 256 zk pairs FFN-up is 372 live 128 x 256 workgroups on 256 CUs (two rounds, 47 us) against
//       192 ping-pong tiles in one round (41 us): zk 2.33 -> 2.10 ms per 256-pair call, lds 2.47 -> 2.26 at 150 pairs, 4.90 -> 4.40 at 350, lxmert 3.63 -> 3.46 at 512;
//       below ~5000 rows the tiles win (zk 140 / 170 pairs +4 % with a bound of 4096), and the N = 768 projections lose at every size below PP_ROWS (profiles/rd5_pp_plain.txt)
//   otherwise, N % 256 == 0                              ENG_TILE; ENG_TILE_DMA when even the PADDED row bound gives no more workgroups than the chip has CUs (calls of
//       up to ~50 .. 120 pairs): the same tile with LDS-DMA double buffering (no VGPR round trip, one barrier per K step; 128 KiB of LDS, so one workgroup per CU -- which
//       is all such a launch has anyway): 10 .. 15 % faster there, slower as soon as a CU would hold two of the register-staged workgroups
//       (profiles/rd4x_tile_engines_midsize.txt).  Same accumulation order: bit-identical to ENG_TILE.
//       (rounds 1-3 sent wide outputs at M >= 8192 to ENG_TILE_256: 82 .. 95 us per launch on lds' 256-pair calls, the 128 x 256 tile is faster there)
//   N % 256 != 0                                         ENG_TILE_128 (run_engine's last resort)
// Three passes (precision mode 3):  rows >= PP_ROWS -> ENG_PPW, else the three-pass 128 x 128 tile.
// ENG_SKINNY / ENG_SKINNY_PARTS are never chosen here: api.hip names them (it knows the row bound of the whole launch wave and owns the partial buffers).
static Engine pick_engine(const GemmParams& p, int nsplit) {
    if (nsplit == 3) return p.M >= pp_rows() ? ENG_PPW : ENG_TILE_128;
    if (p.N % 256 != 0) return ENG_TILE_128;
    if (p.M >= pp_rows()) return ENG_PP_PERSIST;
    if (p.N >= 1536 && p.k_splits <= 1 && p.M >= pp_wide_rows()) return ENG_PP_PERSIST;
    const long long wgs = (long long)((p.M + 127) / 128) * (p.N / 256) * (p.k_splits > 1 ? p.k_splits : 1);
    return wgs <= device_cu_count() ? ENG_TILE_DMA : ENG_TILE;
}

// ---- engine -> launch, with each engine's fall-back when it does not take the shape ---------------------------------------------------------------------
static bool run_engine(Engine e, const GemmParams& p, int nsplit, hipStream_t st) {
    // the wave-level K slicing of the skinny kernel lives in its own field (wave_k_slices): a launch that falls through to a tile engine
    // carries only the split-K contract (k_splits), which that engine honours
    if (nsplit == 3) {   // three passes: ENG_PPW for large M, the 128 x 128 tile otherwise; a handful of rows: skinny kernel
        switch (e) {
            case ENG_SKINNY: if (launch_gemm_skinny(p, 3, st)) return true; break;
            case ENG_SKINNY_PARTS: if (launch_gemm_skinny_parts(p, 3, st)) return true; break;
            case ENG_SKINNY_K4: case ENG_SKINNY_K8: { GemmParams q = p; q.wave_k_slices = e == ENG_SKINNY_K4 ? 4 : 8; if (launch_gemm_skinny(q, 3, st)) return true; } break;
            case ENG_PPW: if (launch_gemm_ppw(p, st)) return true; break;
            default: break;
        }
        return launch_gemm_tile(p, 3, ENG_TILE_128, st);
    }
    Engine tile = ENG_TILE;      // what a non-tile engine falls back to
    switch (e) {
        case ENG_SKINNY:
            if (launch_gemm_skinny(p, nsplit, st)) return true;
            break;
        case ENG_SKINNY_PARTS:      // split-K partials from the skinny kernel (api.hip proj_ln); the tile engine honours the same k_splits contract
            if (launch_gemm_skinny_parts(p, nsplit, st)) return true;
            break;
        case ENG_SKINNY_K4: case ENG_SKINNY_K8: {      // kernel tests: the skinny kernel with 4 / 8 wave-level K slices
            GemmParams q = p; q.wave_k_slices = e == ENG_SKINNY_K4 ? 4 : 8;
            if (launch_gemm_skinny(q, nsplit, st)) return true;
        } break;
#ifdef MMS_LAB
        case ENG_DW:
            if (launch_gemm_dw(p, nsplit, st)) return true;
            if (launch_gemm_pp(p, nsplit, 0, st, true)) return true;
            break;
#endif
        case ENG_PP_PERSIST: if (launch_gemm_pp(p, nsplit, 0, st, true)) return true; break;
        case ENG_PP: if (launch_gemm_pp(p, nsplit, 0, st)) return true; break;
        case ENG_TILE_128: case ENG_TILE_DMA: case ENG_TILE: case ENG_TILE_256: tile = e; break;
        default: break;
    }
    if (launch_gemm_tile(p, nsplit, tile, st)) return true;
    return launch_gemm_tile(p, nsplit, ENG_TILE_128, st);   // N % 256 != 0: 128 x 128 tile
}

// false: no engine took the launch (nothing was enqueued) -- the caller turns that into MMS_ERR_ARG instead of letting the next kernel read an unwritten buffer
bool launch_gemm(const GemmParams& p, int nsplit, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0) return true;
    int e = p.engine;
#ifdef MMS_LAB
    if (e == ENG_AUTO) { static const int env_engine = getenv("MMS_GEMM_VARIANT") ? atoi(getenv("MMS_GEMM_VARIANT")) : ENG_AUTO; e = env_engine; }
    if (e > ENG_DIAG_BASE && nsplit != 3) {   // timing diagnostics (ping-pong): WRONG results on purpose, only with MMS_GEMM_DIAG
        static const bool diag_ok = getenv("MMS_GEMM_DIAG") != nullptr;
        const int d = e - ENG_DIAG_BASE;
        if (diag_ok && (d < 33 || d == 64) && launch_gemm_pp(p, nsplit, d, st)) return true;
        e = ENG_AUTO;
    }
#endif
    if (!engine_is_named(e) || (e == ENG_PPW && nsplit != 3)) e = ENG_AUTO;
#ifndef MMS_LAB
    if (e == ENG_DW) e = ENG_AUTO;
#endif
    return run_engine(e == ENG_AUTO ? pick_engine(p, nsplit) : (Engine)e, p, nsplit, st);
}
