/* A plain-C host of libmmscore ("C / C++ host"): no Python, no torch.  Creates an imagebert_lds handle, loads the
 * checkpoint tensors from a flat file, copies one feed to the device with the HIP runtime's C API, scores it and writes the
 * logits.  tests/test_round3_gpu.py compiles this with gcc and compares the logits bitwise with the ctypes route.
 *
 *   weights file:  int32 n; n x { int32 name_len; char name[name_len]; int32 rank; int64 shape[rank]; float data[prod(shape)] }
 *   feed file:     int64 B, T; int64 input_ids[B*T]; int64 segment_ids[B*T]; float features[B*10*2048]; int64 labelfeat[B*10*8]
 *   usage: lds_host <layers> <vocab> <inter> <weights> <feed> <logits out> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mmscore.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define CHECK_MMS(x) do { int r_ = (x); if (r_ != MMS_OK) { fprintf(stderr, "%s: rc %d: %s\n", #x, r_, h ? mms_last_error(h) : mms_global_error()); return 4; } } while (0)

static void* slurp(FILE* f, size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p || fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage\n"); return 1; }
    mms_handle* h = 0;
    if (mms_version() != MMS_ABI_VERSION) { fprintf(stderr, "ABI revision %d, header %d\n", mms_version(), MMS_ABI_VERSION); return 1; }
    mms_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.model = MMS_MODEL_LDS; cfg.layers = atoi(argv[1]); cfg.vocab = atoi(argv[2]); cfg.inter = atoi(argv[3]);
    cfg.max_pos = 512; cfg.type_vocab = 2; cfg.text_len = 20; cfg.precision = 2; cfg.stop_after = -1; cfg.device = 0; cfg.pack_tokens = 1;
    CHECK_MMS(mms_create(&cfg, &h));

    FILE* f = fopen(argv[4], "rb");
    if (!f) { perror(argv[4]); return 2; }
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    for (int32_t i = 0; i < n; ++i) {
        int32_t len = 0, rank = 0;
        int64_t shape[8], numel = 1;
        char name[512];
        if (fread(&len, 4, 1, f) != 1 || len <= 0 || len >= (int32_t)sizeof name) return 2;
        if (fread(name, 1, (size_t)len, f) != (size_t)len) return 2;
        name[len] = 0;
        if (fread(&rank, 4, 1, f) != 1 || rank < 0 || rank > 8) return 2;
        if (rank && fread(shape, 8, (size_t)rank, f) != (size_t)rank) return 2;
        for (int r = 0; r < rank; ++r) numel *= shape[r];
        float* data = (float*)slurp(f, (size_t)numel * 4);
        CHECK_MMS(mms_load_weight(h, name, data, shape, rank));
        free(data);
    }
    fclose(f);
    CHECK_MMS(mms_finalize(h));

    f = fopen(argv[5], "rb");
    if (!f) { perror(argv[5]); return 2; }
    int64_t BT[2];
    if (fread(BT, 8, 2, f) != 2) return 2;
    const int64_t B = BT[0], T = BT[1];
    const size_t n_ids = (size_t)(B * T) * 8, n_feat = (size_t)B * 10 * 2048 * 4, n_lab = (size_t)B * 10 * 8 * 8;
    void *ids = slurp(f, n_ids), *seg = slurp(f, n_ids), *feat = slurp(f, n_feat), *lab = slurp(f, n_lab);
    fclose(f);
    void *d_ids, *d_seg, *d_feat, *d_lab;
    float *d_logits, *d_probs;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMalloc(&d_ids, n_ids)); CHECK_HIP(hipMalloc(&d_seg, n_ids)); CHECK_HIP(hipMalloc(&d_feat, n_feat)); CHECK_HIP(hipMalloc(&d_lab, n_lab));
    CHECK_HIP(hipMalloc((void**)&d_logits, (size_t)B * 8)); CHECK_HIP(hipMalloc((void**)&d_probs, (size_t)B * 8));
    CHECK_HIP(hipMemcpy(d_ids, ids, n_ids, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(d_seg, seg, n_ids, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_feat, feat, n_feat, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(d_lab, lab, n_lab, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    mms_lds_batch b;
    memset(&b, 0, sizeof b);
    b.n_pairs = B; b.input_ids = (const int64_t*)d_ids; b.segment_ids = (const int64_t*)d_seg; b.features = (const float*)d_feat;
    b.labelfeat = (const int64_t*)d_lab;
    CHECK_MMS(mms_score_lds(h, &b, d_logits, d_probs, st));          /* asynchronous on st */
    CHECK_HIP(hipStreamSynchronize(st));
    float* out = (float*)malloc((size_t)B * 8);
    CHECK_HIP(hipMemcpy(out, d_logits, (size_t)B * 8, hipMemcpyDeviceToHost));
    f = fopen(argv[6], "wb");
    if (!f || fwrite(out, 8, (size_t)B, f) != (size_t)B) return 2;
    fclose(f);
    mms_destroy(h);
    printf("scored %lld pairs\n", (long long)B);
    return 0;
}
