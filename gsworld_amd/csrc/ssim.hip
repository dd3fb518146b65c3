// ssim.hip -- fused SSIM map and its analytic backward (upstream fused-ssim/ssim.cu fusedssimCUDA /
// fusedssim_backwardCUDA).
//
// 11x11 Gaussian window (sigma 1.5) applied separably with zero "same" padding.  One 16x16 output tile per
// 256-thread workgroup, one (batch, channel) plane per blockIdx.z: the 26x26 halo of both images is staged in
// LDS once, the horizontal pass writes 5 running moments (26 rows x 16 columns) back to LDS, the vertical pass
// finishes them per pixel.  HBM traffic is one read of each image plus one write of each output map.
#include "gsr_internal.h"

namespace {

constexpr int kT = 16;        // tile edge
constexpr int kR = 5;         // window radius
constexpr int kH = kT + 2 * kR;  // 26

struct Gauss11 {
    float w[11];
};

__device__ __forceinline__ float load_px(const float *__restrict__ img, int x, int y, int W, int H) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(GSR_BLOCK) void ssim_forward_kernel(int H, int W, float C1, float C2, Gauss11 g,
                                                                 const float *__restrict__ img1,
                                                                 const float *__restrict__ img2, int train,
                                                                 float *__restrict__ ssim_map,
                                                                 float *__restrict__ dm_dmu1,
                                                                 float *__restrict__ dm_dsigma1_sq,
                                                                 float *__restrict__ dm_dsigma12) {
    __shared__ float s1[kH][kH + 1];
    __shared__ float s2[kH][kH + 1];
    __shared__ float sh[5][kH][kT + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *p1 = img1 + plane, *p2 = img2 + plane;
    const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kH * kH; i += GSR_BLOCK) {
        const int ly = i / kH, lx = i - ly * kH;
        s1[ly][lx] = load_px(p1, x0 + lx - kR, y0 + ly - kR, W, H);
        s2[ly][lx] = load_px(p2, x0 + lx - kR, y0 + ly - kR, W, H);
    }
    __syncthreads();
    // horizontal pass: 26 rows x 16 columns
    for (int i = tid; i < kH * kT; i += GSR_BLOCK) {
        const int ly = i / kT, lx = i - ly * kT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float a = s1[ly][lx + k], b = s2[ly][lx + k], w = g.w[k];
            m1 += w * a;
            m2 += w * b;
            e11 += w * (a * a);
            e22 += w * (b * b);
            e12 += w * (a * b);
        }
        sh[0][ly][lx] = m1; sh[1][ly][lx] = m2; sh[2][ly][lx] = e11; sh[3][ly][lx] = e22; sh[4][ly][lx] = e12;
    }
    __syncthreads();
    const int lx = tid & (kT - 1), ly = tid >> 4;
    const int x = x0 + lx, y = y0 + ly;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = g.w[k];
        mu1 += w * sh[0][ly + k][lx];
        mu2 += w * sh[1][ly + k][lx];
        e11 += w * sh[2][ly + k][lx];
        e22 += w * sh[3][ly + k][lx];
        e12 += w * sh[4][ly + k][lx];
    }
    if (x < W && y < H) {
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sigma1_sq = e11 - mu1_sq, sigma2_sq = e22 - mu2_sq, sigma12 = e12 - mu12;
        const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
        const float Cc = 2.f * mu12 + C1, D = 2.f * sigma12 + C2;
        const size_t o = plane + (size_t)y * W + x;
        ssim_map[o] = (Cc * D) / (A * B);
        if (train) {
            dm_dmu1[o] = (mu2 * 2.f * D) / (A * B) - (mu2 * 2.f * Cc) / (A * B) - (mu1 * 2.f * Cc * D) / (A * A * B) +
                         (mu1 * 2.f * Cc * D) / (A * B * B);
            dm_dsigma1_sq[o] = (-Cc * D) / (A * B * B);
            dm_dsigma12[o] = (2.f * Cc) / (A * B);
        }
    }
}

__global__ __launch_bounds__(GSR_BLOCK) void ssim_backward_kernel(int H, int W, Gauss11 g,
                                                                  const float *__restrict__ img1,
                                                                  const float *__restrict__ img2,
                                                                  const float *__restrict__ dL_dmap,
                                                                  const float *__restrict__ dm_dmu1,
                                                                  const float *__restrict__ dm_dsigma1_sq,
                                                                  const float *__restrict__ dm_dsigma12,
                                                                  float *__restrict__ dL_dimg1) {
    __shared__ float s[3][kH][kH + 1];
    __shared__ float sh[3][kH][kT + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kH * kH; i += GSR_BLOCK) {
        const int ly = i / kH, lx = i - ly * kH;
        const int x = x0 + lx - kR, y = y0 + ly - kR;
        float a = 0.f, b = 0.f, c = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = plane + (size_t)y * W + x;
            const float dl = dL_dmap[o];
            a = dl * dm_dmu1[o];
            b = dl * dm_dsigma1_sq[o];
            c = dl * dm_dsigma12[o];
        }
        s[0][ly][lx] = a; s[1][ly][lx] = b; s[2][ly][lx] = c;
    }
    __syncthreads();
    for (int i = tid; i < kH * kT; i += GSR_BLOCK) {
        const int ly = i / kT, lx = i - ly * kT;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = g.w[k];
            a += w * s[0][ly][lx + k];
            b += w * s[1][ly][lx + k];
            c += w * s[2][ly][lx + k];
        }
        sh[0][ly][lx] = a; sh[1][ly][lx] = b; sh[2][ly][lx] = c;
    }
    __syncthreads();
    const int lx = tid & (kT - 1), ly = tid >> 4;
    const int x = x0 + lx, y = y0 + ly;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = g.w[k];
        a += w * sh[0][ly + k][lx];
        b += w * sh[1][ly + k][lx];
        c += w * sh[2][ly + k][lx];
    }
    if (x < W && y < H) {
        const size_t o = plane + (size_t)y * W + x;
        dL_dimg1[o] = a + 2.f * img1[o] * b + img2[o] * c;
    }
}

Gauss11 make_window() {
    Gauss11 g;
    double w[11], sum = 0.0;
    for (int k = 0; k < 11; k++) {
        w[k] = exp(-((double)(k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5));
        sum += w[k];
    }
    for (int k = 0; k < 11; k++) g.w[k] = (float)(w[k] / sum);
    return g;
}

}  // namespace

extern "C" int gsr_ssim_forward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float *img1,
                                const float *img2, int32_t train, float *ssim_map, float *dm_dmu1,
                                float *dm_dsigma1_sq, float *dm_dsigma12, void *stream) {
    if (B < 0 || CH < 0 || H < 0 || W < 0) {
        gsr_set_error("gsr_ssim_forward: negative size");
        return GSR_E_INVALID;
    }
    if ((size_t)B * CH * H * W == 0) return GSR_OK;
    if (!img1 || !img2 || !ssim_map || (train && (!dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12))) {
        gsr_set_error("gsr_ssim_forward: null pointer");
        return GSR_E_INVALID;
    }
    if ((int64_t)B * CH > 65535) {
        gsr_set_error("gsr_ssim_forward: batch*channels exceeds the grid's z extent");
        return GSR_E_INVALID;
    }
    const dim3 grid(gsr_div_up(W, kT), gsr_div_up(H, kT), B * CH);
    hipLaunchKernelGGL(ssim_forward_kernel, grid, dim3(GSR_BLOCK), 0, (hipStream_t)stream, H, W, C1, C2, make_window(),
                       img1, img2, train, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
    return gsr_check_launch("ssim_forward", false, (hipStream_t)stream);
}

extern "C" int gsr_ssim_backward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float *img1,
                                 const float *img2, const float *dL_dmap, const float *dm_dmu1,
                                 const float *dm_dsigma1_sq, const float *dm_dsigma12, float *dL_dimg1, void *stream) {
    (void)C1;
    (void)C2;
    if (B < 0 || CH < 0 || H < 0 || W < 0) {
        gsr_set_error("gsr_ssim_backward: negative size");
        return GSR_E_INVALID;
    }
    if ((size_t)B * CH * H * W == 0) return GSR_OK;
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) {
        gsr_set_error("gsr_ssim_backward: null pointer");
        return GSR_E_INVALID;
    }
    if ((int64_t)B * CH > 65535) {
        gsr_set_error("gsr_ssim_backward: batch*channels exceeds the grid's z extent");
        return GSR_E_INVALID;
    }
    const dim3 grid(gsr_div_up(W, kT), gsr_div_up(H, kT), B * CH);
    hipLaunchKernelGGL(ssim_backward_kernel, grid, dim3(GSR_BLOCK), 0, (hipStream_t)stream, H, W, make_window(), img1,
                       img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
    return gsr_check_launch("ssim_backward", false, (hipStream_t)stream);
}
