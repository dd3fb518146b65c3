// ssim.hip -- fused SSIM map and its analytic backward (upstream fused-ssim/ssim.cu fusedssimCUDA /
// fusedssim_backwardCUDA).
//
// 11x11 Gaussian window (sigma 1.5) applied separably with zero "same" padding.  One 32x16 output tile per
// 256-thread workgroup, one (batch, channel) plane per blockIdx.z: the 42x26 halo of both images is staged in
// LDS once, the horizontal pass writes 5 running moments (26 rows x 32 columns) back to LDS, the vertical pass
// finishes them per pixel.  HBM traffic is one read of each image plus one write of each output map.
#include "gsr_internal.h"

namespace {

constexpr int kTX = 32, kTY = 16;  // output tile of a workgroup
constexpr int kR = 5;              // window radius
constexpr int kHX = kTX + 2 * kR, kHY = kTY + 2 * kR;  // 42 x 26 halo

struct Gauss11 {
    float w[11];
};

// (the file is compiled without contraction like the rest of the library; the window sums are spelled as fused
//  multiply-adds -- one rounding per tap, as nvcc contracts upstream's -- which also halves their instruction count)
__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float load_px(const float *__restrict__ img, int x, int y, int W, int H) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.f;
}

// LOSS (gsr_photometric_loss): img1 is read through clamp(., 0, 1) when `clamp01`; no ssim map is written -- the
// workgroup's sums of the ssim values and of |img1 - img2| over its 32 x 16 pixels go to partials[2 b], [2 b + 1]
// (b = the workgroup's linear index; summed in a fixed order: the loss is reproducible bit for bit).
//
// Tile of 32 x 16 pixels per 256-thread workgroup.  Both passes are bound by their LDS reads, so each thread makes
// several outputs from one window of loaded values: the horizontal pass 4 adjacent columns from 14 values of a row
// (instead of 4 x 11), the vertical pass 2 adjacent rows from 12 values of a column.  Every output still sums its 11
// taps in the same order, one fused multiply-add per tap: the maps are the same bits whatever the tiling.
template <bool LOSS>
__global__ __launch_bounds__(GSR_BLOCK, 5) void ssim_forward_kernel(int H, int W, float C1, float C2, Gauss11 g,
                                                                 const float *__restrict__ img1,
                                                                 const float *__restrict__ img2, int train,
                                                                 float *__restrict__ ssim_map,
                                                                 float *__restrict__ dm_dmu1,
                                                                 float *__restrict__ dm_dsigma1_sq,
                                                                 float *__restrict__ dm_dsigma12, int clamp01,
                                                                 float *__restrict__ partials) {
    __shared__ float s1[kHY][kHX + 1];
    __shared__ float s2[kHY][kHX + 1];
    __shared__ float sh[5][kHY][kTX + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *p1 = img1 + plane, *p2 = img2 + plane;
    const int x0 = blockIdx.x * kTX, y0 = blockIdx.y * kTY;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kHY * kHX; i += GSR_BLOCK) {
        const int ly = i / kHX, lx = i - ly * kHX;
        float a = load_px(p1, x0 + lx - kR, y0 + ly - kR, W, H);
        if (LOSS && clamp01) a = a != a ? a : fminf(fmaxf(a, 0.f), 1.f);  // (torch.clamp keeps a NaN; max / min drop it)
        s1[ly][lx] = a;
        s2[ly][lx] = load_px(p2, x0 + lx - kR, y0 + ly - kR, W, H);
    }
    __syncthreads();
    // horizontal pass: 26 rows x 8 groups of 4 columns
    if (tid < kHY * (kTX / 4)) {
        const int ly = tid / (kTX / 4), c0 = (tid - ly * (kTX / 4)) * 4;
        float a[14], b[14], aa[14], bb[14], ab[14];
#pragma unroll
        for (int j = 0; j < 14; j++) {
            a[j] = s1[ly][c0 + j];
            b[j] = s2[ly][c0 + j];
        }
        // (the three products once per loaded value, not once per tap that uses it: the same roundings, 42 multiplies for
        //  the four outputs instead of 132)
#pragma unroll
        for (int j = 0; j < 14; j++) {
            aa[j] = a[j] * a[j];
            bb[j] = b[j] * b[j];
            ab[j] = a[j] * b[j];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = g.w[k];
                m1 = fmaf_(w, a[c + k], m1);
                m2 = fmaf_(w, b[c + k], m2);
                e11 = fmaf_(w, aa[c + k], e11);
                e22 = fmaf_(w, bb[c + k], e22);
                e12 = fmaf_(w, ab[c + k], e12);
            }
            sh[0][ly][c0 + c] = m1; sh[1][ly][c0 + c] = m2; sh[2][ly][c0 + c] = e11; sh[3][ly][c0 + c] = e22;
            sh[4][ly][c0 + c] = e12;
        }
    }
    __syncthreads();
    // vertical pass: column lx, rows 2 rp and 2 rp + 1
    const int lx = tid & (kTX - 1), rp = tid >> 5;
    float acc[2][5];
#pragma unroll
    for (int m = 0; m < 5; m++) {  // (one moment at a time: 12 live values, not 60)
        float v[12];
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = sh[m][2 * rp + j][lx];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) t = fmaf_(g.w[k], v[r + k], t);
            acc[r][m] = t;
        }
    }
    float v_ssim = 0.f, v_l1 = 0.f;
    const int x = x0 + lx;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int ly = 2 * rp + r, y = y0 + ly;
        if (x < W && y < H) {
            const float mu1 = acc[r][0], mu2 = acc[r][1], e11 = acc[r][2], e22 = acc[r][3], e12 = acc[r][4];
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float sigma1_sq = e11 - mu1_sq, sigma2_sq = e22 - mu2_sq, sigma12 = e12 - mu12;
            const float A = mu1_sq + mu2_sq + C1, B = sigma1_sq + sigma2_sq + C2;
            const float Cc = 2.f * mu12 + C1, D = 2.f * sigma12 + C2;
            const size_t o = plane + (size_t)y * W + x;
            if (LOSS) {
                // (the loss is compared at 1e-6 and summed over a million pixels: the hardware reciprocals of A and B --
                //  1 ulp each -- replace the seven correctly rounded divisions of upstream's expressions, which the
                //  drop-in maps below keep)
                const float iA = __builtin_amdgcn_rcpf(A), iB = __builtin_amdgcn_rcpf(B), iAB = iA * iB;
                v_ssim += (Cc * D) * iAB;
                v_l1 += fabsf(s1[ly + kR][lx + kR] - s2[ly + kR][lx + kR]);
                if (train) {
                    const float t = 2.f * Cc * D;
                    dm_dmu1[o] = 2.f * mu2 * (D - Cc) * iAB + mu1 * t * iAB * (iB - iA);
                    dm_dsigma1_sq[o] = -(Cc * D) * iAB * iB;
                    dm_dsigma12[o] = (2.f * Cc) * iAB;
                }
            } else {
                ssim_map[o] = (Cc * D) / (A * B);
            }
            if (!LOSS && train) {
                dm_dmu1[o] = (mu2 * 2.f * D) / (A * B) - (mu2 * 2.f * Cc) / (A * B) - (mu1 * 2.f * Cc * D) / (A * A * B) +
                             (mu1 * 2.f * Cc * D) / (A * B * B);
                dm_dsigma1_sq[o] = (-Cc * D) / (A * B * B);
                dm_dsigma12[o] = (2.f * Cc) / (A * B);
            }
        }
    }
    if (LOSS) {
        __shared__ float s_red[2][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            v_ssim += __shfl_xor(v_ssim, o, 64);
            v_l1 += __shfl_xor(v_l1, o, 64);
        }
        if (gsr_lane() == 0) {
            s_red[0][gsr_wave()] = v_ssim;
            s_red[1][gsr_wave()] = v_l1;
        }
        __syncthreads();
        if (tid == 0) {
            const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partials[2 * b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
            partials[2 * b + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
        }
    }
}

// one workgroup: loss = (1 - lambda) sum|.| / n + lambda (1 - sum ssim / n) from the per-workgroup partial sums
constexpr int kFinT = 1024;
__global__ __launch_bounds__(kFinT) void loss_finish_kernel(const float *__restrict__ partials, int nb, double n,
                                                            float lambda, float *__restrict__ loss) {
    __shared__ double s_a[kFinT], s_b[kFinT];
    double a = 0.0, b = 0.0;
    const float2 *pp = reinterpret_cast<const float2 *>(partials);
    for (int i0 = (int)threadIdx.x; i0 < nb; i0 += 4 * kFinT) {  // (four loads in flight per thread and round)
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = i0 + u * kFinT < nb ? pp[i0 + u * kFinT] : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a += (double)v[u].x;
            b += (double)v[u].y;
        }
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
    __syncthreads();
    for (int o = kFinT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_a[threadIdx.x] += s_a[threadIdx.x + o];
            s_b[threadIdx.x] += s_b[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l1 = s_b[0] / n, ssim = s_a[0] / n;
        loss[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ssim));
    }
}

// LOSS (gsr_photometric_loss): dL/dmap is the constant w_ssim (= -lambda / n); the L1 term w_l1 sign(x - y) is added
// (sign(0) = 0, torch's abs backward) and the whole gradient passes through clamp(., 0, 1)'s backward when `clamp01`
// (kept where 0 <= img1 <= 1, bounds included, as torch.clamp).  Same 32 x 16 tiling as the forward.
template <bool LOSS>
__global__ __launch_bounds__(GSR_BLOCK, 6) void ssim_backward_kernel(int H, int W, Gauss11 g,
                                                                  const float *__restrict__ img1,
                                                                  const float *__restrict__ img2,
                                                                  const float *__restrict__ dL_dmap,
                                                                  const float *__restrict__ dm_dmu1,
                                                                  const float *__restrict__ dm_dsigma1_sq,
                                                                  const float *__restrict__ dm_dsigma12,
                                                                  float *__restrict__ dL_dimg1, float w_ssim,
                                                                  float w_l1, int clamp01) {
    __shared__ float s[3][kHY][kHX + 1];
    __shared__ float sh[3][kHY][kTX + 1];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * kTX, y0 = blockIdx.y * kTY;
    const int tid = (int)threadIdx.x;
    if (LOSS) {  // the scalar loss's incoming gradient scales both terms
        const float gl = dL_dmap != nullptr ? dL_dmap[0] : 1.0f;
        w_ssim *= gl;
        w_l1 *= gl;
    }
    for (int i = tid; i < kHY * kHX; i += GSR_BLOCK) {
        const int ly = i / kHX, lx = i - ly * kHX;
        const int x = x0 + lx - kR, y = y0 + ly - kR;
        float a = 0.f, b = 0.f, c = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = plane + (size_t)y * W + x;
            const float dl = LOSS ? w_ssim : dL_dmap[o];
            a = dl * dm_dmu1[o];
            b = dl * dm_dsigma1_sq[o];
            c = dl * dm_dsigma12[o];
        }
        s[0][ly][lx] = a; s[1][ly][lx] = b; s[2][ly][lx] = c;
    }
    __syncthreads();
    if (tid < kHY * (kTX / 4)) {
        const int ly = tid / (kTX / 4), c0 = (tid - ly * (kTX / 4)) * 4;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float v[14];
#pragma unroll
            for (int j = 0; j < 14; j++) v[j] = s[m][ly][c0 + j];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) t = fmaf_(g.w[k], v[c + k], t);
                sh[m][ly][c0 + c] = t;
            }
        }
    }
    __syncthreads();
    const int lx = tid & (kTX - 1), rp = tid >> 5;
    float acc[2][3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        float v[12];
#pragma unroll
        for (int j = 0; j < 12; j++) v[j] = sh[m][2 * rp + j][lx];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) t = fmaf_(g.w[k], v[r + k], t);
            acc[r][m] = t;
        }
    }
    const int x = x0 + lx;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = y0 + 2 * rp + r;
        if (x < W && y < H) {
            const float a = acc[r][0], b = acc[r][1], c = acc[r][2];
            const size_t o = plane + (size_t)y * W + x;
            if (LOSS) {
                const float raw = img1[o], yv = img2[o];
                const float xv = (clamp01 && raw == raw) ? fminf(fmaxf(raw, 0.f), 1.f) : raw;
                const float d = xv - yv;
                float gsum = a + 2.f * xv * b + yv * c;
                gsum += w_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                dL_dimg1[o] = (!clamp01 || (raw >= 0.f && raw <= 1.f)) ? gsum : 0.f;
            } else {
                dL_dimg1[o] = a + 2.f * img1[o] * b + img2[o] * c;
            }
        }
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
    const dim3 grid(gsr_div_up(W, kTX), gsr_div_up(H, kTY), B * CH);
    hipLaunchKernelGGL(ssim_forward_kernel<false>, grid, dim3(GSR_BLOCK), 0, (hipStream_t)stream, H, W, C1, C2,
                       make_window(), img1, img2, train, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, 0,
                       (float *)nullptr);
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
    const dim3 grid(gsr_div_up(W, kTX), gsr_div_up(H, kTY), B * CH);
    hipLaunchKernelGGL(ssim_backward_kernel<false>, grid, dim3(GSR_BLOCK), 0, (hipStream_t)stream, H, W, make_window(),
                       img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, 0.f, 0.f, 0);
    return gsr_check_launch("ssim_backward", false, (hipStream_t)stream);
}

// ---- the photometric loss of the 3DGS training step in two passes --------------------------------------------------
extern "C" size_t gsr_photometric_loss_scratch_floats(int32_t planes, int32_t H, int32_t W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    const size_t nb = (size_t)gsr_div_up(W, kTX) * gsr_div_up(H, kTY) * planes;
    return 3 * (size_t)planes * H * W + 2 * nb;
}

extern "C" int gsr_photometric_loss(int32_t planes, int32_t H, int32_t W, const float *img, const float *target,
                                    float lambda_dssim, int32_t clamp01, float *scratch, float *loss,
                                    int32_t keep_for_backward, void *stream_) {
    if (planes <= 0 || H <= 0 || W <= 0 || planes > 65535) {
        gsr_set_error("gsr_photometric_loss: planes must be in 1 .. 65535, H and W positive");
        return GSR_E_INVALID;
    }
    if (!img || !target || !scratch || !loss) {
        gsr_set_error("gsr_photometric_loss: null pointer");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t n = (size_t)planes * H * W;
    const dim3 grid(gsr_div_up(W, kTX), gsr_div_up(H, kTY), planes);
    const int nb = (int)(grid.x * grid.y * grid.z);
    float *partials = scratch, *maps = scratch + 2 * (size_t)nb;  // (the pairs first: read as float2)
    float *dm_dmu1 = maps, *dm_dsigma1_sq = maps + n, *dm_dsigma12 = maps + 2 * n;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const Gauss11 g = make_window();
    hipLaunchKernelGGL(ssim_forward_kernel<true>, grid, dim3(GSR_BLOCK), 0, stream, H, W, C1, C2, g, img, target,
                       keep_for_backward ? 1 : 0, (float *)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, clamp01, partials);
    if (int e = gsr_check_launch("photometric_loss (forward)", false, stream)) return e;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kFinT), 0, stream, (const float *)partials, nb, (double)n,
                       lambda_dssim, loss);
    return gsr_check_launch("photometric_loss (sum)", false, stream);
}

extern "C" int gsr_photometric_loss_backward(int32_t planes, int32_t H, int32_t W, const float *img, const float *target,
                                             float lambda_dssim, int32_t clamp01, const float *scratch,
                                             const float *grad_loss, float *dL_dimg, void *stream_) {
    if (planes <= 0 || H <= 0 || W <= 0 || planes > 65535) {
        gsr_set_error("gsr_photometric_loss_backward: planes must be in 1 .. 65535, H and W positive");
        return GSR_E_INVALID;
    }
    if (!img || !target || !scratch || !dL_dimg) {
        gsr_set_error("gsr_photometric_loss_backward: null pointer");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const size_t n = (size_t)planes * H * W;
    const dim3 grid(gsr_div_up(W, kTX), gsr_div_up(H, kTY), planes);
    const size_t nb = (size_t)grid.x * grid.y * grid.z;
    const float *maps = scratch + 2 * nb;
    const float w_ssim = (float)(-(double)lambda_dssim / (double)n);
    const float w_l1 = (float)((1.0 - (double)lambda_dssim) / (double)n);
    // (LOSS mode: the dL_dmap slot carries the incoming gradient of the scalar loss, one float on the device, or NULL = 1)
    hipLaunchKernelGGL(ssim_backward_kernel<true>, grid, dim3(GSR_BLOCK), 0, stream, H, W, make_window(), img, target,
                       grad_loss, maps, maps + n, maps + 2 * n, dL_dimg, w_ssim, w_l1, clamp01);
    return gsr_check_launch("photometric_loss (gradient)", false, stream);
}
