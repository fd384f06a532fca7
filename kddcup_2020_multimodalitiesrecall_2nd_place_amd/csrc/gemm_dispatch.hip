// GEMM dispatch: which tile engine runs a given (M, N, K, passes) contraction.
//
// Product library (libmmscore.so): per-shape choice between the kernels a forward can launch -- the 256x256 persistent ping-pong
// engine (gemm_pp.hip) for the big encoder GEMMs, its 256x128 three-pass cut (gemm_ppw.hip) for precision mode 3, and the
// register-staged tiles of gemm_tile.hip for the small ones (CLS-only last block, poolers, heads, label text).  No environment
// variable is read here and there is no process-global state: a caller that wants ONE engine names it per launch (GemmParams::variant;
// the kernel tests do, through mms_dbg_gemm).
//
// Lab library (libmmscore_lab.so, `make lab`, -DMMS_LAB): additionally honours MMS_GEMM_VARIANT, reaches the LDS-DMA double-buffer
// tile of gemm_tile.hip, the measured-and-shelved engines (gemm_dw.hip: variant 28; gemm_mx.hip's fp16 + MX-fp8 "1.5 pass" kernel) and
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
    static const int v = getenv("MMS_PP_ROWS") ? atoi(getenv("MMS_PP_ROWS")) : 16384;
    return v;
#else
    return 16384;
#endif
}

int pp_wide_rows() {
#ifdef MMS_LAB
    static const int v = getenv("MMS_PP_WIDE_ROWS") ? atoi(getenv("MMS_PP_WIDE_ROWS")) : 5120;
    return v;
#else
    return 5120;
#endif
}

// false: no engine took the launch (nothing was enqueued) -- the caller turns that into MMS_ERR_ARG instead of letting the next kernel read an unwritten buffer
bool launch_gemm(const GemmParams& p, int nsplit, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0) return true;
    int variant = p.variant ? p.variant : 99;
#ifdef MMS_LAB
    if (!p.variant) { static const int env_variant = getenv("MMS_GEMM_VARIANT") ? atoi(getenv("MMS_GEMM_VARIANT")) : 99; variant = env_variant; }
#endif
    if (nsplit == 3) {   // three passes: 256x128 ping-pong phases for large M (variant 27 forces it), 128x128 tile otherwise; a handful of rows: skinny kernel (api.hip names it)
        if (variant == 5) {      // (k_splits is the skinny kernel's wave-level K slicing there: the tile engines below must not read it as their split-K contract)
            if (launch_gemm_skinny(p, 3, st)) return true;
            GemmParams q = p; q.k_splits = 0;
            return launch_gemm_tile(q, 3, 1, st);
        }
        if (variant == 55 && launch_gemm_skinny_parts(p, 3, st)) return true;
        if (variant == 54 || variant == 58) { GemmParams q = p; q.k_splits = variant - 50; if (launch_gemm_skinny(q, 3, st)) return true; }
        if ((variant == 27 || (variant == 99 && p.M >= pp_rows())) && launch_gemm_ppw(p, st)) return true;
        return launch_gemm_tile(p, 3, 1, st);
    }
#ifdef MMS_LAB
    if (variant > 100) {   // timing diagnostics (201-232 ping-pong): WRONG results on purpose, only with MMS_GEMM_DIAG
        static const bool diag_ok = getenv("MMS_GEMM_DIAG") != nullptr;
        if (diag_ok && variant > 200 && (variant < 233 || variant == 264) && launch_gemm_pp(p, nsplit, variant - 200, st)) return true;
        variant = 99;
    }
#endif
    if (variant != 1 && variant != 3 && variant != 4 && variant != 5 && variant != 54 && variant != 55 && variant != 58 && variant != 16 && variant != 20 && variant != 26
#ifdef MMS_LAB
        && variant != 28
#endif
    ) variant = 99;
    if (variant == 99) {  // auto (profiles/r01c_gemm_variants.txt): 256x256 ping-pong phases for large M; for the small GEMMs
                          // (CLS-only last block, poolers) 256x256 / 16 waves on wide outputs, 128x256 / 8 waves otherwise
        if (p.N % 256 == 0 && p.M >= pp_rows()) variant = 26;   // ping-pong phases, persistent workgroups (20 = one tile per workgroup)
        // ... and for the WIDE projections (N >= 1536: FFN-up, an unfused QKV / K | V) already from 5120 padded rows on: the engines contract K in the same order per element
        // (bit-identical, tests/test_kernels_gpu.py), so this is a speed choice alone.  At 256 zk pairs FFN-up is 372 live 128 x 256 workgroups on 256 CUs (two rounds, 47 us)
        // against 192 ping-pong tiles in one round (41 us): zk 2.33 -> 2.10 ms per 256-pair call, lds 2.47 -> 2.26 at 150 pairs, 4.90 -> 4.40 at 350, lxmert 3.63 -> 3.46 at 512;
        // below ~5000 rows the tiles win (zk 140 / 170 pairs +4 % with a bound of 4096), and the N = 768 projections lose at every size below 16 384 (profiles/rd5_pp_plain.txt)
        else if (p.N % 256 == 0 && p.N >= 1536 && p.k_splits <= 1 && p.M >= pp_wide_rows()) variant = 26;
        else variant = 4;   // (rounds 1-3 sent wide outputs at M >= 8192 to the 256x256 / 16-wave tile: 82 .. 95 us per launch on lds' 256-pair calls, the 128x256 tile is faster there)
        // ... and when even the PADDED row bound gives no more workgroups than the chip has CUs (calls of up to ~50 .. 120 pairs): the same tile with LDS-DMA
        // double buffering (no VGPR round trip, one barrier per K step; 128 KiB of LDS, so one workgroup per CU -- which is all such a launch has anyway):
        // 10 .. 15 % faster there, slower as soon as a CU would hold two of the register-staged workgroups (profiles/rd4x_tile_engines_midsize.txt).
        // Same accumulation order: bit-identical to variant 4.
        if (variant == 4 && p.N % 256 == 0) {
            const long long wgs = (long long)((p.M + 127) / 128) * (p.N / 256) * (p.k_splits > 1 ? p.k_splits : 1);
            if (wgs <= device_cu_count()) variant = 3;
        }
    }
#ifdef MMS_LAB
    if (variant == 28) { if (launch_gemm_dw(p, nsplit, st)) return true; variant = 26; }
#endif
    if (variant == 5) {      // a handful of rows (api.hip names it with its K-slice count in k_splits; never the per-shape default here)
        if (launch_gemm_skinny(p, nsplit, st)) return true;
        // not taken: k_splits was the skinny kernel's wave-level K slicing (bias and activation still in the epilogue) -- the tile engine's k_splits means
        // "fp32 partials, no epilogue", so it must not see it
        GemmParams q = p; q.k_splits = 0;
        if (launch_gemm_tile(q, nsplit, 4, st)) return true;
        return launch_gemm_tile(q, nsplit, 1, st);
    }
    if (variant == 55) { if (launch_gemm_skinny_parts(p, nsplit, st)) return true; variant = 4; }      // split-K partials from the skinny kernel (api.hip proj_ln); the tile engine honours the same k_splits contract
    if (variant == 54 || variant == 58) {      // kernel tests: the skinny kernel with 4 / 8 K slices
        GemmParams q = p; q.k_splits = variant - 50;
        if (launch_gemm_skinny(q, nsplit, st)) return true;
        variant = 4;
    }
    if (variant == 26) { if (launch_gemm_pp(p, nsplit, 0, st, true)) return true; variant = 4; }
    if (variant == 20) { if (launch_gemm_pp(p, nsplit, 0, st)) return true; variant = 4; }
    if (launch_gemm_tile(p, nsplit, variant, st)) return true;
    return launch_gemm_tile(p, nsplit, 1, st);   // N % 256 != 0: 128x128 tile
}
