// Internal interface of the fused feature-space kNN translation unit (psg_knn.hip; not part of the C ABI).
// Reference: ResGCN/gcn_lib/dense/torch_edge.py:32-59 (pairwise_distance + topk(-dist, k * d)), :19-29 (every d-th neighbour).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace psg {

struct KnnBuffers {
    float *xp = nullptr;              // [rows][64] fp32 operand copy (v_mfma_f32_16x16x4_f32 order), exact path
    void *bp = nullptr;               // [rows / 32][9][64] x 16 bytes: bf16 hi / lo / augmented fragments, prefilter path
    float *sq = nullptr;              // [rows] squared norms in torch.sum's order
    const float *x = nullptr;         // the features themselves, row-major [rows][ld] (16-byte aligned rows): the prefilter path reads
    int ld = 0;                       // its few finalists' 64 floats from here (256 contiguous bytes instead of 16 pieces of xp)
    unsigned long long *stats = nullptr;   // optional device counters of the prefilter kernel ([8], see psg_knn_bf.cuh: KnnBfArgs::stats)
};

enum KnnPath { KNN_PATH_BF16 = 0, KNN_PATH_F32 = 1 };

size_t knn_xp_bytes(size_t rows);
size_t knn_bp_bytes(size_t rows);
// raises the kernels' dynamic-LDS limit once per process; returns hipSuccess when both fused kernels can run here
hipError_t knn_setup();
bool knn_shape_ok(int N, int k, int d, KnnPath path);
// xp, bp, sq of row-major x [rows][ld] (the network's producer kernel writes them itself: psg_knn_ops.cuh)
hipError_t knn_prep_launch(const float *x, int ld, size_t rows, const KnnBuffers &buf, bool want_bp, hipStream_t st);
// out [B * N][k]: ranks 0, d, .., (k - 1) d of every point's neighbours inside its room
hipError_t knn_launch(const KnnBuffers &buf, int B, int N, int k, int d, int32_t *out, KnnPath path, hipStream_t st);

}  // namespace psg
