// Per-operator entry points behind the stand-alone forwards of the reference's public modules:
//   PointNetSetAbstraction(.Msg).forward   PointNet/models/pointnet_util.py:181-207, 229-267
//   PointNetFeaturePropagation.forward     PointNet/models/pointnet_util.py:281-320
// (the ResGCN ones - EdgeConv2d, pairwise_distance - live in psg_resgcn.hip next to the kernels they launch).
// Inside get_model these modules run as ONE fused kernel each (psg_pn2.hip); called on their own they are composed from
// the operators here, every one a thin launch on the caller's tensors: grouping (gather + centre + concat), the shared
// 1x1-conv stack on the fp32-MFMA row GEMM of psg_gemm.cuh (BatchNorm folded by the caller, ReLU bits kept for the
// input-gradient pass), max-pool over the K samples with its arg index, 3-NN interpolation + concat, and the
// input-gradient transposes of each.  Parameter gradients are not produced: the attack path differentiates w.r.t. the
// input only (SURVEY.md 3.1 "Backward dataflow actually needed").
#include <algorithm>

#include "psg_common.h"
#include "psg_gemm.cuh"

using namespace psg;

namespace {

GemmArgs ops_args(const float *in, int ld_in, const float *w, int ld_w, float *out, int ld_out, int rows, int K, int M)
{
    GemmArgs a;
    a.in = in; a.w = w; a.bias = nullptr; a.gbias = nullptr; a.scale = nullptr; a.shift = nullptr; a.sq = nullptr;
    a.out = out; a.mask_out = nullptr; a.mask_in = nullptr;
    a.rows = rows; a.K = K; a.M = M; a.ld_in = ld_in; a.ld_w = ld_w; a.ld_out = ld_out; a.group_rows = 1;
    a.accumulate = 0; a.addend = nullptr; a.ld_add = 0;
    return a;
}

template <int EPI>
int ops_gemm(const GemmArgs &a, hipStream_t st)
{
    dim3 grid(ceil_div(a.rows, 128), ceil_div(a.M, 128));
    if ((size_t)grid.x * grid.y < 128) {   // few 128-wide tiles: 64 x 64 tiles, one MFMA tile per wave
        dim3 small(ceil_div(a.rows, 64), ceil_div(a.M, 64));
        hipLaunchKernelGGL((gemm_rows_kernel<2, 2, EPI, false, 1, 1>), small, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((gemm_rows_kernel<2, 2, EPI, false>), grid, dim3(256), 0, st, a);
    }
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// rows[(b, s, k)][:] = [xyz[b][g] - new_xyz[b][s] (3), feat[b][g][0..D)]  with g = gidx[b][s][k]
// (feat_first: [feat, rel_xyz], the MSG order, pointnet_util.py:251-254; else the SSG order, :137)
__global__ void group_rows_kernel(const float *__restrict__ xyz, const float *__restrict__ feat, const float *__restrict__ new_xyz,
                                  const int32_t *__restrict__ gidx, int N, int S, int K, int D, int feat_first, size_t total,
                                  float *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int C = D + 3;
    const size_t row = t / C;
    const int c = (int)(t % C);
    const size_t bs = row / K;                 // (b, s)
    const size_t b = bs / S;
    const int g = gidx[row];
    const bool is_xyz = feat_first ? c >= D : c < 3;
    const int cc = feat_first ? (is_xyz ? c - D : c) : (is_xyz ? c : c - 3);
    out[t] = is_xyz ? xyz[(b * N + g) * 3 + cc] - new_xyz[bs * 3 + cc] : feat[(b * N + g) * (size_t)D + cc];
}

// transpose of the feature part: dfeat[b][g][c] += drows[row][feature column c]   (dfeat zeroed by the caller)
__global__ void group_rows_bwd_kernel(const float *__restrict__ drows, const int32_t *__restrict__ gidx, int N, int S, int K, int D,
                                      int feat_first, size_t total, float *__restrict__ dfeat)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / D;
    const int c = (int)(t % D);
    const size_t b = row / ((size_t)S * K);
    const float v = drows[row * (D + 3) + (feat_first ? c : 3 + c)];
    if (v != 0.0f) atomicAdd(dfeat + (b * N + gidx[row]) * (size_t)D + c, v);
}

// out[g][c] = max_k in[g*K + k][c], arg = first k that attains it (torch.max: lowest index on ties)
__global__ void group_max_fwd_kernel(const float *__restrict__ in, int K, int C, size_t total, float *__restrict__ out,
                                     uint8_t *__restrict__ arg)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t g = t / C;
    const int c = (int)(t % C);
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        const float v = in[(g * K + k) * (size_t)C + c];
        if (v > best) { best = v; bk = k; }
    }
    out[t] = best;
    arg[t] = (uint8_t)bk;
}

// din[g*K + k][c] = dout[g][c] if k == arg[g][c] and the last layer's ReLU let it through (bit c of its mask row)
__global__ void group_max_bwd_kernel(const float *__restrict__ dout, const uint8_t *__restrict__ arg, const uint32_t *__restrict__ mask,
                                     int K, int C, size_t total, float *__restrict__ din)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // over rows * C
    if (t >= total) return;
    const size_t row = t / C;
    const int c = (int)(t % C);
    const size_t g = row / K;
    const int k = (int)(row % K);
    const bool on = arg[g * C + c] == k && (!mask || ((mask[row * ((C + 31) >> 5) + (c >> 5)] >> (c & 31)) & 1u));
    din[t] = on ? dout[g * C + c] : 0.0f;
}

// out[b][n][:] = [feat1[b][n][0..D1), sum_j w[b][n][j] * feat2[b][idx[b][n][j]][0..D2)]   (pointnet_util.py:301-314)
__global__ void three_interp_fwd_kernel(const float *__restrict__ feat2, const int32_t *__restrict__ idx, const float *__restrict__ w,
                                        const float *__restrict__ feat1, int N, int S, int D1, int D2, size_t total,
                                        float *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int C = D1 + D2;
    const size_t bn = t / C;
    const int c = (int)(t % C);
    if (c < D1) { out[t] = feat1[bn * D1 + c]; return; }
    const size_t b = bn / N;
    const int32_t *ii = idx + bn * 3;
    const float *ww = w + bn * 3;
    const float *f = feat2 + b * S * (size_t)D2 + (c - D1);
    // torch.sum(index_points(points2, idx) * weight.view(B, N, 3, 1), dim=2): ((w0 f0 + w1 f1) + w2 f2), no fma
    out[t] = __fadd_rn(__fadd_rn(__fmul_rn(ww[0], f[(size_t)ii[0] * D2]), __fmul_rn(ww[1], f[(size_t)ii[1] * D2])),
                       __fmul_rn(ww[2], f[(size_t)ii[2] * D2]));
}

__global__ void three_interp_bwd_kernel(const float *__restrict__ dout, int ld, int col0, const int32_t *__restrict__ idx,
                                        const float *__restrict__ w, int N, int S, int D2, size_t total, float *__restrict__ dfeat2)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // over B * N * D2
    if (t >= total) return;
    const size_t bn = t / D2;
    const int c = (int)(t % D2);
    const size_t b = bn / N;
    const float g = dout[bn * ld + col0 + c];
    if (g == 0.0f) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) atomicAdd(dfeat2 + (b * S + idx[bn * 3 + j]) * (size_t)D2 + c, w[bn * 3 + j] * g);
}

// g[row][c] *= bit c of bits[row]  (backward through a layer's own ReLU)
__global__ void apply_relu_bits_kernel(float *__restrict__ g, int ld, const uint32_t *__restrict__ bits, const float *__restrict__ scale,
                                       int M, size_t total)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const size_t row = t / M;
    const int c = (int)(t % M);
    const bool on = (bits[row * ((M + 31) >> 5) + (c >> 5)] >> (c & 31)) & 1u;
    float *p = g + row * ld + c;
    *p = on ? (scale ? *p * scale[c] : *p) : 0.0f;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int psg_group_rows(const float *xyz, const float *feat, const float *new_xyz, const int32_t *gidx, int B, int N, int S,
                              int K, int D, int feat_first, float *rows_out, psg_stream stream)
{
    PSG_REQUIRE(xyz && new_xyz && gidx && rows_out && (feat || D == 0), "psg_group_rows: null argument");
    PSG_REQUIRE(B > 0 && N > 0 && S > 0 && K > 0 && D >= 0, "psg_group_rows: bad sizes");
    const size_t total = (size_t)B * S * K * (D + 3);
    hipLaunchKernelGGL(group_rows_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, xyz, feat, new_xyz, gidx, N, S,
                       K, D, feat_first, total, rows_out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

extern "C" int psg_group_rows_bwd(const float *drows, const int32_t *gidx, int B, int N, int S, int K, int D, int feat_first,
                                  float *dfeat, psg_stream stream)
{
    PSG_REQUIRE(drows && gidx && dfeat && D > 0, "psg_group_rows_bwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    PSG_CHECK_HIP(hipMemsetAsync(dfeat, 0, (size_t)B * N * D * sizeof(float), st));
    const size_t total = (size_t)B * S * K * D;
    hipLaunchKernelGGL(group_rows_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, st, drows, gidx, N, S, K, D, feat_first, total,
                       dfeat);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// one shared 1x1-conv layer over point-major rows: out = [relu](in . w^T + bias) [* scale + shift]; w [M][K].  PointNet++:
// Conv -> BN(eval) -> ReLU with the BatchNorm folded into w / bias by the caller (pointnet_util.py:200-203,317-319); ResGCN's
// BasicConv: Conv -> ReLU -> BN(eval) through scale / shift (torch_nn.py:55-75).  mask_out: ReLU bits [rows][ceil(M/32)]
extern "C" int psg_pw_mlp_fwd(const float *in, int ld_in, int rows, int K, const float *w, const float *bias, int relu, int M,
                              float *out, int ld_out, uint32_t *mask_out, const float *scale, const float *shift, psg_stream stream)
{
    PSG_REQUIRE(in && w && out && rows > 0 && K > 0 && M > 0, "psg_pw_mlp_fwd: bad argument");
    PSG_REQUIRE((!scale && !shift) || (relu && scale && shift), "psg_pw_mlp_fwd: the affine follows the ReLU (needs relu, scale and shift)");
    GemmArgs a = ops_args(in, ld_in, w, K, out, ld_out, rows, K, M);
    a.bias = bias;
    a.scale = scale; a.shift = shift;
    a.mask_out = mask_out;
    return relu ? ops_gemm<EPI_RELU_AFFINE>(a, (hipStream_t)stream) : ops_gemm<EPI_LINEAR>(a, (hipStream_t)stream);
}

// input gradient of that layer: din = (dout . w) * [ReLU bits of the layer BELOW];  wT [K][M] = w transposed
extern "C" int psg_pw_mlp_bwd(const float *dout, int ld_dout, int rows, int M, const float *wT, const uint32_t *mask_below, int K,
                              float *din, int ld_din, psg_stream stream)
{
    PSG_REQUIRE(dout && wT && din && rows > 0 && K > 0 && M > 0, "psg_pw_mlp_bwd: bad argument");
    GemmArgs a = ops_args(dout, ld_dout, wT, M, din, ld_din, rows, M, K);
    a.mask_in = mask_below;
    return ops_gemm<EPI_LINEAR>(a, (hipStream_t)stream);
}

// backward through a layer's OWN ReLU (the top layer of a stack): g[row][c] = 0 where bit c of bits[row] is clear, else
// g * scale[c] when a BatchNorm affine follows the ReLU (scale may be NULL)
extern "C" int psg_apply_relu_bits(float *g, int ld, const uint32_t *bits, const float *scale, int rows, int M, psg_stream stream)
{
    PSG_REQUIRE(g && bits && rows > 0 && M > 0, "psg_apply_relu_bits: bad argument");
    const size_t total = (size_t)rows * M;
    hipLaunchKernelGGL(apply_relu_bits_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, g, ld, bits, scale, M, total);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// PointNetSetAbstraction's arithmetic after grouping (pointnet_util.py:200-205): n_layers shared 1x1-conv layers over the
// grouped rows [n_groups * K][cin], then the max over the K samples of a group.  Ping-pong scratch [rows][max width] x 2,
// ReLU bits of every layer to masks[l] ([rows][ceil(width_l / 32)]) when the backward will be wanted.
extern "C" int psg_sa_mlp_max_fwd(const float *rows_in, int n_groups, int K, int cin, int n_layers, const int *widths,
                                  const float *const *w, const float *const *bias, float *scratch_a, float *scratch_b,
                                  uint32_t *const *masks, float *out, uint8_t *arg, psg_stream stream)
{
    PSG_REQUIRE(rows_in && widths && w && bias && scratch_a && scratch_b && out && arg && n_layers >= 1 && n_layers <= 8 &&
                    K >= 1 && K <= 255, "psg_sa_mlp_max_fwd: bad argument");
    const int rows = n_groups * K;
    const float *cur = rows_in;
    int c = cin;
    for (int l = 0; l < n_layers; ++l) {
        float *dst = (l & 1) ? scratch_b : scratch_a;
        int rc = psg_pw_mlp_fwd(cur, c, rows, c, w[l], bias[l], 1, widths[l], dst, widths[l], masks ? masks[l] : nullptr, nullptr,
                                nullptr, stream);
        if (rc) return rc;
        cur = dst;
        c = widths[l];
    }
    const size_t total = (size_t)n_groups * c;
    hipLaunchKernelGGL(group_max_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, cur, K, c, total, out, arg);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// its input gradient: dout [n_groups][C_last] -> drows_in [rows][cin]; wT[l] = w[l] transposed ([width_{l-1}][width_l])
extern "C" int psg_sa_mlp_max_bwd(const float *dout, const uint8_t *arg, int n_groups, int K, int cin, int n_layers,
                                  const int *widths, const float *const *wT, const uint32_t *const *masks, float *scratch_a,
                                  float *scratch_b, float *drows_in, psg_stream stream)
{
    PSG_REQUIRE(dout && arg && widths && wT && masks && scratch_a && scratch_b && drows_in && n_layers >= 1 && n_layers <= 8,
                "psg_sa_mlp_max_bwd: bad argument");
    const int rows = n_groups * K;
    const int c_last = widths[n_layers - 1];
    const size_t total = (size_t)rows * c_last;
    float *cur = (n_layers & 1) ? scratch_a : scratch_b;
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dout, arg,
                       masks[n_layers - 1], K, c_last, total, cur);
    PSG_LAUNCH_CHECK();
    for (int l = n_layers - 1; l >= 0; --l) {
        const int c_in = l ? widths[l - 1] : cin;
        float *dst = l == 0 ? drows_in : (cur == scratch_a ? scratch_b : scratch_a);
        int rc = psg_pw_mlp_bwd(cur, widths[l], rows, widths[l], wT[l], l ? masks[l - 1] : nullptr, c_in, dst, c_in, stream);
        if (rc) return rc;
        cur = dst;
    }
    return PSG_OK;
}

extern "C" int psg_three_interp_fwd(const float *feat2, const int32_t *idx, const float *w, const float *feat1, int B, int N, int S,
                                    int D1, int D2, float *out, psg_stream stream)
{
    PSG_REQUIRE(feat2 && idx && w && out && (feat1 || D1 == 0) && D2 > 0, "psg_three_interp_fwd: bad argument");
    const size_t total = (size_t)B * N * (D1 + D2);
    hipLaunchKernelGGL(three_interp_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, feat2, idx, w, feat1, N, S,
                       D1, D2, total, out);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}

// transpose of the interpolation: dfeat2[b][idx][c] += w * dout[b][n][col0 + c]   (dfeat2 is zeroed here)
extern "C" int psg_three_interp_bwd(const float *dout, int ld, int col0, const int32_t *idx, const float *w, int B, int N, int S,
                                    int D2, float *dfeat2, psg_stream stream)
{
    PSG_REQUIRE(dout && idx && w && dfeat2 && D2 > 0, "psg_three_interp_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    PSG_CHECK_HIP(hipMemsetAsync(dfeat2, 0, (size_t)B * S * D2 * sizeof(float), st));
    const size_t total = (size_t)B * N * D2;
    hipLaunchKernelGGL(three_interp_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, st, dout, ld, col0, idx, w, N, S, D2, total,
                       dfeat2);
    PSG_LAUNCH_CHECK();
    return PSG_OK;
}
