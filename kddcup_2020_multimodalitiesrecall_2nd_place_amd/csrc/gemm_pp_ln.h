// Fused "bias + residual + LayerNorm" epilogue of the ping-pong GEMM for the N = 768 projections (attention output, FFN down:
// pixelbert.py:960-966,980-985; modeling.py:369-378,408-420).  Removes the fp32 round trip of the pre-LayerNorm tensor (GEMM
// writes 3 KB/row, the LayerNorm kernel reads it back with the residual and writes the planes: 12 KB/row/LayerNorm through
// HBM becomes 6) and the LayerNorm launch itself.
//
// A row's 768 outputs are spread over the THREE 256-column tiles of its row panel, i.e. over three workgroups.  Each keeps its
// tile (acc + bias + residual) in the accumulator registers, reduces per-row {sum, sum of squares} over its 256 columns (lane
// shuffles -> LDS across the four column waves), PUBLISHES them as two 8-byte {value, launch tag} granules per row (agent-scope
// relaxed atomic stores = write-through; no fence: a granule is self-validating), POLLS the granules of the other two tiles of
// the panel (agent-scope atomic loads), and normalises its own tile in registers:  y = (v - mean) * rsqrt(var + eps) * gamma + beta,
// var = E[v^2] - mean^2 in fp32 (biased, eps inside the root: pixelbert.py:414-417 / modeling.py:266), stored as split planes.
//
// Waiting on another workgroup is only safe if that workgroup is RUNNING.  The launch is persistent (one workgroup per CU), every
// workgroup checks in at kernel entry, and the launch decides ONCE for the whole grid, before the first tile (a CAS on ln_ctl[1]):
// all gridDim.x workgroups have checked in -> they are all resident, none leaves before its tiles are done, so every wait ends
// (tiles are walked in increasing virtual order and a tile's partners sit within +-16 of it: no cycle) -> mode 1, fused.  Otherwise
// (the device is shared with another process, a CU is unavailable, ...) -> mode 2: this launch writes the plain fp32 tensor and the
// LayerNorm kernel queued behind it does its usual work; in mode 1 that kernel sees ln_ctl[1] == 1 and returns at once.
// The XCD remap assigns WHOLE row panels to an XCD, so the three partner tiles run on neighbouring CUs in the same round.
#pragma once
#include "kernels.h"
#include "gemm_pp_epilogue.h"

__device__ __forceinline__ unsigned long long ln_granule(float v, unsigned tag) {
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// 1 = fused, 2 = plain; decided once per launch (a CAS), identical for every workgroup.  Called by every wave at kernel entry, after
// its workgroup's check-in: fused as soon as all gridDim.x workgroups are in, plain if they are not within ~30 us (wall_clock64 ticks
// at 100 MHz) -- a workgroup that arrives later than that just reads the decision.
__device__ __forceinline__ int ln_decide(int* ctl, int grid, int lane) {
    int m = 0;
    if (lane == 0) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            m = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (m) break;
            const int c = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c >= grid) atomicCAS(&ctl[1], 0, 1);
            else if (wall_clock64() - t0 > 3000ull) atomicCAS(&ctl[1], 0, 2);
            else __builtin_amdgcn_s_sleep(8);
        }
    }
    return __builtin_amdgcn_readfirstlane(m);
}

// Plain route of an LN launch (mode 2): fp32 rows of acc (* weight scale) + bias; in this mode the accumulators started at zero and
// the LayerNorm kernel queued behind the launch adds the residual.
template <int FM, int FN>
__device__ __forceinline__ void pp_epilogue_plain_f32(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0, int lane, int Meff) {
    const int mrow = lane & 15, col = col0 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 s4 = p.col_scale ? *reinterpret_cast<const f32x4*>(p.col_scale + col + 16 * j) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = row0 + 16 * i + mrow;
            if (row < Meff) *reinterpret_cast<f32x4*>(p.c_f32 + (long long)row * p.ldc + col + 16 * j) = acc[i][j] * s4 + b4;
        }
    }
}

// acc[i][j] of lane l = C[row0 + 16 i + (l & 15)][col0 + 16 j + 4 (l >> 4) + 0..3]  (transposed fragments, gemm_pp_epilogue.h).
// lds: >= 14 KiB of LDS nobody else touches during the epilogue (ring slot 2).  All 512 threads call this.
#ifdef MMS_LAB
#define LN_STAMP(k) do { if (p.ln_dbg && tid == 0) p.ln_dbg[(long long)dbg_tile * 6 + (k)] = wall_clock64(); } while (0)
#else
#define LN_STAMP(k) do { } while (0)
#endif
template <int FM, int FN>
__device__ __forceinline__ void pp_epilogue_ln(const GemmParams& p, f32x4 (&acc)[FM][FN], int bm, int bn, int wm, int wn, int lane, int tid,
                                               int Meff, unsigned char* lds, int dbg_tile = 0) {
    // every address of this epilogue is re-derived per tile: hoisted out of the persistent loop they sit in registers (or scratch: the kernel
    // has none to spare) across the whole K loop
    asm volatile("" : "+v"(lane), "+v"(tid));
    LN_STAMP(1);
    const int mrow = lane & 15, nq = lane >> 4;
    const int row0 = bm * 256 + wm * 128, col = bn * 256 + wn * 64 + nq * 4;
    // ---- v = acc + bias: the residual is already in the accumulator (the residual stages of gemm_pp.hip's K loop) ----
    // row sums go to LDS row by row (no 16-register array of partial sums kept across the loop)
    float2* red = reinterpret_cast<float2*>(lds);                 // [4][256]
    float2* part = reinterpret_cast<float2*>(lds + 8192);         // [512]: the two partner tiles' sums per row
    float2* stat = reinterpret_cast<float2*>(lds + 8192 + 4096);  // [256]: mean, rstd
    if (p.bias) {      // column fragment outermost: four bias registers live at a time
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col + 16 * j);
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[i][j] += b4;
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ss += acc[i][j][e]; qq += acc[i][j][e] * acc[i][j][e]; }
        }
        if (row0 + 16 * i + mrow >= Meff) { ss = 0.f; qq = 0.f; }
        ss = rows4_sum(ss); qq = rows4_sum(qq);      // common.h: v_permlane16/32_swap, not ds_bpermute (16 dependent reductions per lane: ~1 us per tile)
        if (nq == 0) red[wn * 256 + wm * 128 + 16 * i + mrow] = float2{ss, qq};
    }
    __syncthreads();
    LN_STAMP(2);
    const int r = tid & 255;
    const int grow = bm * 256 + r;
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(p.ln_stats);
    float2 mine = float2{0.f, 0.f};
    if (tid < 256) {
        mine = red[r];
        const float2 a = red[256 + r], b = red[512 + r], c = red[768 + r];
        mine.x += a.x + b.x + c.x; mine.y += a.y + b.y + c.y;
        if (grow < Meff) {
            unsigned long long* g = gran + ((long long)grow * 3 + bn) * 2;
            __hip_atomic_store(g, ln_granule(mine.x, p.ln_tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, ln_granule(mine.y, p.ln_tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    {   // threads 0..255 take the next tile of the panel, 256..511 the one after; every publisher is running (see the header).
        // ONE lane per partner waits (on the granules of the partner's last live row, with a long sleep between looks: hundreds of
        // lanes spinning on L2 slow every CU's operand stream, MI355X_MICROARCH.md "polling-cost"); the others sleep at the
        // barrier and then find their own granules in place -- each still checks its tag, a granule is self-validating.
        const int which = tid < 256 ? (bn + 1) % 3 : (bn + 2) % 3;
        const int last = (bm * 256 + 255 < Meff ? bm * 256 + 255 : Meff - 1);
        if ((tid & 255) == 0) {
            const unsigned long long* g = gran + ((long long)last * 3 + which) * 2;
            for (;;) {
                const unsigned long long a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(a >> 32) == p.ln_tag && (unsigned)(b >> 32) == p.ln_tag) break;
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
        LN_STAMP(3);
        float2 got = float2{0.f, 0.f};
        if (grow < Meff) {
            const unsigned long long* g = gran + ((long long)grow * 3 + which) * 2;
            unsigned long long a, b;
            for (;;) {
                a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(a >> 32) == p.ln_tag && (unsigned)(b >> 32) == p.ln_tag) break;
                __builtin_amdgcn_s_sleep(4);
            }
            got = float2{__uint_as_float((unsigned)a), __uint_as_float((unsigned)b)};
        }
        part[tid] = got;
    }
    __syncthreads();
    if (tid < 256) {
        const float2 a = part[r], b = part[256 + r];
        const float S = mine.x + a.x + b.x, Q = mine.y + a.y + b.y;
        const float mean = S * (1.0f / MMS_HIDDEN);
        const float var = fmaxf(Q * (1.0f / MMS_HIDDEN) - mean * mean, 0.f);
        stat[r] = float2{mean, 1.0f / sqrtf(var + MMS_LN_EPS)};
    }
    __syncthreads();
    LN_STAMP(4);
    // ---- normalise this tile and store the planes: column fragment PAIRS outermost (gamma / beta cost 16 registers, not 32), one
    // 16-byte store per plane, row and pair (pp_store_plane_pair, gemm_pp_epilogue.h) ----
    const int col0 = bn * 256 + wn * 64;
#pragma unroll
    for (int j = 0; j < FN; j += 2) {
        f32x4 g4[2], b4[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            g4[jj] = *reinterpret_cast<const f32x4*>(p.ln_gamma + col + 16 * (j + jj));
            b4[jj] = *reinterpret_cast<const f32x4*>(p.ln_beta + col + 16 * (j + jj));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = row0 + 16 * i + mrow;
            if (row >= Meff) continue;
            const float2 ms = stat[wm * 128 + 16 * i + mrow];
            bf16x4 h[2], l[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = (acc[i][j + jj][e] - ms.x) * ms.y * g4[jj][e] + b4[jj][e];
                    bf16 a, c2;
                    split_bf16(y[e], a, c2);
                    h[jj][e] = a; l[jj][e] = c2;
                }
                if (p.c_f8) *reinterpret_cast<unsigned*>(p.c_f8 + (long long)row * p.ldf8 + col + 16 * (j + jj)) = pack4_f8(y[0], y[1], y[2], y[3]);
            }
            bf16* dh = pp_plane_lane_row(p.c_hi, (long long)row * p.ldp, col0, nq);
            pp_store_plane_pair(dh, j, h[0], h[1]);
            pp_store_plane_pair(dh + (p.c_lo - p.c_hi), j, l[0], l[1]);
        }
    }
#ifdef MMS_LAB
    if (p.ln_dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); LN_STAMP(5); }
#endif
}
