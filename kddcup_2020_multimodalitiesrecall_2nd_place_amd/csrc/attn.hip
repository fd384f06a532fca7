// Small-sequence multi-head attention: one wavefront per (pair, head), everything in registers.
//
// Reference: pixelbert.py:790-850 / pixelmodel.py:770-831 (zk, lds) and lxrt/modeling.py:326-352
// (lxmert self- and cross-attention): scores = QK^T / sqrt(64) + (1-mask)*-10000, softmax over
// keys, probs @ V.  Sequences are tiny (S = 30 / 40 / 23 / 10, head dim 64) so the whole head lives
// in one wave's VGPRs -- no LDS, no cross-wave traffic.
//
// Exact-fp32 matrix path: v_mfma_f32_16x16x4_f32 (one rounding per product == an fmaf chain), so the
// attention block adds no bf16 rounding to the parity budget (it is < 1 % of the FLOPs).
//
// Register choreography (wave64, 16x16 tiles):
//  * S^T = K Q^T is computed "swapped": A-operand = K rows, B-operand = Q rows, so the C layout
//    (col = lane & 15 -> query i, row = 4*(lane>>4)+r -> key j) leaves each softmax row spread over
//    the 4 lanes {i, i+16, i+32, i+48} x 4 regs x key tiles: the row reduction is 2 shuffles.
//  * The same registers are directly the A-operand of P V (A wants row i = lane & 15, k = lane >> 4),
//    no transpose or LDS round trip.
//  * Q/K fragments are float4 loads (16 B/lane); the MFMA k index is consistently permuted in both
//    operands (d = 16 s + 4 (lane>>4) + t), V fragments are float4 loads with the output column
//    permuted d = 4 (lane&15) + dt so each row's 64 outputs are stored as contiguous 8-B bf16x4.
#include "kernels.h"

template <int QT, int KT>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    const int lane = threadIdx.x & 63;
    int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= p.B * MMS_HEADS) return;
    if (p.reverse) unit = p.B * MMS_HEADS - 1 - unit;
    const int b = unit / MMS_HEADS, h = unit % MMS_HEADS;
    const int fr = lane & 15, fk = lane >> 4;
    // dense: rows b*S .. ; packed (ragged): per-pair row offset / live-token count
    const int q0 = p.q_base + (p.q_off ? p.q_off[b] : b * (p.q_stride ? p.q_stride : p.Sq));
    const int kv0 = p.kv_base + (p.kv_off ? p.kv_off[b] : b * p.Sk);
    const int Sq = p.q_cnt ? p.q_cnt[b] : p.Sq;
    const int Sk = p.kv_cnt ? p.kv_cnt[b] : p.Sk;

    // ---- S^T = K Q^T ----
    float4 kf[KT][4], qf[QT][4];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt) {
        if (jt * 16 >= Sk) continue;   // wave-uniform: this key tile holds no live key (ragged pairs)
        int j = jt * 16 + fr;
        j = j < Sk ? j : Sk - 1;
        const float* kr = p.k + (long long)(kv0 + j) * p.ldkv + h * p.hs_kv + fk * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[jt][s] = *reinterpret_cast<const float4*>(kr + s * 16);
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        if (qt * 16 >= Sq) continue;
        int i = qt * 16 + fr;
        i = i < Sq ? i : Sq - 1;
        const float* qr = p.q + (long long)(q0 + i) * p.ldq + h * p.hs_q + fk * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qt][s] = *reinterpret_cast<const float4*>(qr + s * 16);
    }
    // V fragments are requested up front as well: all of the unit's HBM traffic is in flight before the first MFMA (the loads used
    // to be issued one row group at a time inside the P V loop, each exposing its latency)
    float4 vfr[KT][4];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt) {
        if (jt * 16 >= Sk) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int j = jt * 16 + fk * 4 + r;
            j = j < Sk ? j : Sk - 1;  // P is exactly 0 there; keep the load in bounds and finite
            vfr[jt][r] = *reinterpret_cast<const float4*>(p.v + (long long)(kv0 + j) * p.ldkv + h * p.hs_kv + fr * 4);
        }
    }
    f32x4 sc[KT][QT];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (jt * 16 < Sk && qt * 16 < Sq)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][s].x, qf[qt][s].x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][s].y, qf[qt][s].y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][s].z, qf[qt][s].z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jt][s].w, qf[qt][s].w, a, 0, 0, 0);
            }
            sc[jt][qt] = a;
        }

    // ---- scale, mask, softmax over keys (rows of S^T) ----
    float add[KT][4];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jt * 16 + fk * 4 + r;
            add[jt][r] = j < Sk ? (p.key_add ? p.key_add[kv0 - p.kv_base + j] : 0.f) : -INFINITY;
        }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        if (qt * 16 >= Sq) continue;
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = sc[jt][qt][r] * 0.125f + add[jt][r];
                sc[jt][qt][r] = v;
                m = fmaxf(m, v);
            }
        m = rows4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(sc[jt][qt][r] - m);
                sc[jt][qt][r] = e;
                sum += e;
            }
        sum = rows4_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[jt][qt][r] *= inv;
    }

    // ---- O = P V ----
    f32x4 o[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (jt * 16 >= Sk) continue;   // whole tile has P == 0
            const float4 vf = vfr[jt][r];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (qt * 16 >= Sq) continue;
                const float pv = sc[jt][qt][r];
                o[qt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.x, o[qt][0], 0, 0, 0);
                o[qt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.y, o[qt][1], 0, 0, 0);
                o[qt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.z, o[qt][2], 0, 0, 0);
                o[qt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vf.w, o[qt][3], 0, 0, 0);
            }
        }

    // ---- store: lane holds O[i = qt*16 + 4*fk + r][d = 4*fr + dt], dt = 0..3 ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = qt * 16 + fk * 4 + r;
            if (i >= Sq) continue;
            const long long off = (long long)(p.o_compact ? b : q0 + i) * p.ldo + h * MMS_HEAD_DIM + fr * 4;
            if (p.o_f8) {   // precision mode 4: the context feeds an fp8 GEMM, nothing else reads it
                *reinterpret_cast<unsigned*>(p.o_f8 + off) = pack4_f8(o[qt][0][r], o[qt][1][r], o[qt][2][r], o[qt][3][r]);
                continue;
            }
            bf16x4 hi, lo;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16 a, c;
                split_bf16(o[qt][dt][r], a, c);
                hi[dt] = a;
                lo[dt] = c;
            }
            *reinterpret_cast<bf16x4*>(plane_ptr(p.o_hi, off)) = hi;      // hl32 plane layout (common.h)
            *reinterpret_cast<bf16x4*>(plane_ptr(p.o_lo, off)) = lo;
        }
}

// false: no instantiation covers (Sq, Sk) -- sequences beyond 48 tokens; the caller must turn that into an error (a silent
// no-op would leave stale context rows behind and score garbage)
bool launch_attention(const AttnParams& p, hipStream_t st) {
    const int qt = (p.Sq + 15) / 16, kt = (p.Sk + 15) / 16;
    const dim3 grid((p.B * MMS_HEADS + 3) / 4), block(256);
    if (p.B <= 0) return true;
#define ATTN_CASE(Q, K) \
    if (qt == Q && kt == K) { hipLaunchKernelGGL((attn_kernel<Q, K>), grid, block, 0, st, p); return true; }
    ATTN_CASE(1, 1) ATTN_CASE(1, 2) ATTN_CASE(2, 1) ATTN_CASE(2, 2) ATTN_CASE(3, 3)
    ATTN_CASE(1, 3) ATTN_CASE(3, 1) ATTN_CASE(2, 3) ATTN_CASE(3, 2)
#undef ATTN_CASE
    return false;
}
