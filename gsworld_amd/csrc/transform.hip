// transform.hip -- fused per-step rigid transform of labelled Gaussians.
//
// What GSWorld does per step and per camera-frame (/root/reference/gsworld/mani_skill/utils/wrappers/
// gs_world_wrapper.py:110-162 and :244-265): deep-copy the whole model (352 MB at 1.47 M Gaussians), then for each
// of ~18 movable parts build an isin() mask over all N labels, gather, run transform_gaussians
// (gsworld/utils/gs_utils.py:283-385: scale -> rotate -> translate), and scatter xyz / rotation back through the
// same mask.  Here: ONE pass over N.  label -> transform index through a small LUT, then
//   xyz' = R (s * xyz) + t,   rot' = standardize(q_R (x) rot/|rot|) * |rot|
// written straight into the buffers the rasterizer reads; untouched labels are copied through.
// HBM traffic: 32 B read + 28 B written per Gaussian (88 MB at 1.47 M) -- a pure stream.
#include "gsr_internal.h"

namespace {

constexpr int kXf = 17;  // floats per transform: R (row-major 9), t (3), scale (1), q_R (w,x,y,z)

// blockIdx.y = environment e: the same Gaussians under the e-th pose table, written to the e-th slice of the outputs
// (the (B,N,.) shapes transform_gaussians returns for a batch of poses, gs_utils.py:283-385).  SCALING: also rewrite the
// log-scale parameter of the parts flagged in `rescale` the way the reference does for a per-env scale vector
// (gs_utils.py `scaling = inverse_sigmoid(torch.exp(scaling) * scale)`, log(x / (1 - x)) -- sic), copy it for the rest.
template <bool SCALING>
__global__ __launch_bounds__(GSR_BLOCK) void transform_kernel(int P, const float *__restrict__ xyz,
                                                              const float *__restrict__ rot,
                                                              const float *__restrict__ scaling,
                                                              const float *__restrict__ semantics,
                                                              const int32_t *__restrict__ lut, int lut_size,
                                                              const float *__restrict__ xf_all, int K,
                                                              const uint8_t *__restrict__ rescale,
                                                              float *__restrict__ xyz_out_all,
                                                              float *__restrict__ rot_out_all,
                                                              float *__restrict__ scaling_out_all) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    if (i >= P) return;
    const size_t e = blockIdx.y;
    const float *xf = xf_all + e * (size_t)K * kXf;
    float *xyz_out = xyz_out_all + e * (size_t)P * 3, *rot_out = rot_out_all + e * (size_t)P * 4;
    float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    float4 q = *reinterpret_cast<const float4 *>(rot + 4 * (size_t)i);
    const int label = (int)semantics[i];  // the reference compares labels after .long() (truncation)
    const int k = (label >= 0 && label < lut_size) ? lut[label] : -1;
    if (SCALING) {
        float s0 = scaling[3 * (size_t)i], s1 = scaling[3 * (size_t)i + 1], s2 = scaling[3 * (size_t)i + 2];
        if (k >= 0 && k < K && rescale != nullptr && rescale[k]) {
            const float s = xf[(size_t)k * kXf + 12];
            const float x0 = expf(s0) * s, x1 = expf(s1) * s, x2 = expf(s2) * s;
            s0 = logf(x0 / (1.0f - x0));
            s1 = logf(x1 / (1.0f - x1));
            s2 = logf(x2 / (1.0f - x2));
        }
        float *so = scaling_out_all + e * (size_t)P * 3 + 3 * (size_t)i;
        so[0] = s0; so[1] = s1; so[2] = s2;
    }
    if (k >= 0 && k < K) {
        const float *T = xf + (size_t)k * kXf;
        const float s = T[12];
        px *= s; py *= s; pz *= s;
        const float rx = T[0] * px + T[1] * py + T[2] * pz + T[9];
        const float ry = T[3] * px + T[4] * py + T[5] * pz + T[10];
        const float rz = T[6] * px + T[7] * py + T[8] * pz + T[11];
        px = rx; py = ry; pz = rz;
        // Gaussian quaternion: rotate the normalised quaternion, keep its norm (gs_utils.py:242-249)
        const float norm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        const float bw = q.x / norm, bx = q.y / norm, by = q.z / norm, bz = q.w / norm;
        const float aw = T[13], ax = T[14], ay = T[15], az = T[16];
        float ow = aw * bw - ax * bx - ay * by - az * bz;
        float ox = aw * bx + ax * bw + ay * bz - az * by;
        float oy = aw * by - ax * bz + ay * bw + az * bx;
        float oz = aw * bz + ax * by - ay * bx + az * bw;
        if (ow < 0.f) { ow = -ow; ox = -ox; oy = -oy; oz = -oz; }  // quaternion_multiply standardises the sign
        q = make_float4(ow * norm, ox * norm, oy * norm, oz * norm);
    }
    xyz_out[3 * (size_t)i] = px;
    xyz_out[3 * (size_t)i + 1] = py;
    xyz_out[3 * (size_t)i + 2] = pz;
    *reinterpret_cast<float4 *>(rot_out + 4 * (size_t)i) = q;
}

// (K,4,4) rigid matrices (+ uniform scales) -> the 17-float transform table, on the device, so that a closed loop
// whose link poses already live on the GPU (ManiSkill hands them over as device tensors) never touches the host.
// The quaternion is ManiSkill's / PyTorch3D's matrix_to_quaternion in its exact float32 order (mirror:
// gsworld_amd/transform.py): 4 candidate rows scaled by 1 / (2 max(q_abs, 0.1)), the row of the largest q_abs (first
// on ties), sign standardised to a non-negative real part.
__device__ __forceinline__ void pack_one_transform(const float *__restrict__ M, const float scale, float *__restrict__ T) {
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9],
                m22 = M[10];
    const float t0 = M[3], t1 = M[7], t2 = M[11];
    const float d[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    float qa[4];
    int best = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        qa[r] = d[r] > 0.0f ? sqrtf(d[r]) : 0.0f;
        if (qa[r] > qa[best]) best = r;
    }
    const float c[4][4] = {{qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
                           {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
                           {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
                           {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
    const float denom = 2.0f * fmaxf(qa[best], 0.1f);
    float q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = c[best][j] / denom;
    if (q[0] < 0.0f) {
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = -q[j];
    }
    T[0] = m00; T[1] = m01; T[2] = m02; T[3] = m10; T[4] = m11; T[5] = m12; T[6] = m20; T[7] = m21; T[8] = m22;
    T[9] = t0; T[10] = t1; T[11] = t2;
    T[12] = scale;
    T[13] = q[0]; T[14] = q[1]; T[15] = q[2]; T[16] = q[3];
}

__global__ void pack_transforms_kernel(int K, const float *__restrict__ matrices, const float *__restrict__ scales,
                                       float *__restrict__ table) {
    const int k = blockIdx.x * blockDim.x + (int)threadIdx.x;
    if (k >= K) return;
    pack_one_transform(matrices + 16 * (size_t)k, scales ? scales[k] : 1.0f, table + (size_t)k * kXf);
}

// One step's host values -> the device, in ONE launch: `src` is a pinned host buffer the device reads directly (n floats:
// part matrices, uniform scales, camera matrices -- gsworld_amd/closed_loop.py keeps them in one staging vector), `dst`
// its device-resident twin the step's kernels read; the K part matrices at src + mat_off (scales at src + scale_off, or
// none) are packed into the 17-float pose table on the way.  Replaces an H2D copy + pack_transforms_kernel per
// closed-loop step (4.2 + 4.4 us on the step's stream) by one kernel whose loads cross PCIe.
__global__ __launch_bounds__(GSR_BLOCK) void stage_step_kernel(int n, const float *__restrict__ src,
                                                               float *__restrict__ dst, int K, int mat_off,
                                                               int scale_off, float *__restrict__ table) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    // (both requests go out before either is used)
    const float v = i < n ? src[i] : 0.0f;
    if (i < K) pack_one_transform(src + mat_off + 16 * (size_t)i, scale_off >= 0 ? src[scale_off + i] : 1.0f,
                                  table + (size_t)i * kXf);
    if (i < n) dst[i] = v;
}

}  // namespace

extern "C" int gsr_pack_part_transforms(int32_t K, const float *matrices, const float *scales, float *table,
                                        void *stream) {
    if (K < 0 || (K > 0 && (!matrices || !table))) {
        gsr_set_error("gsr_pack_part_transforms: negative K or null pointer");
        return GSR_E_INVALID;
    }
    if (K == 0) return GSR_OK;
    hipLaunchKernelGGL(pack_transforms_kernel, dim3(gsr_div_up(K, 64)), dim3(64), 0, (hipStream_t)stream, K, matrices,
                       scales, table);
    return gsr_check_launch("pack_part_transforms", false, (hipStream_t)stream);
}

extern "C" int gsr_pinned_device_address(const void *host, void **device) {
    if (!host || !device) {
        gsr_set_error("gsr_pinned_device_address: null argument");
        return GSR_E_INVALID;
    }
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void *>(host), 0) != hipSuccess || d == nullptr) {
        (void)hipGetLastError();
        gsr_set_error("gsr_pinned_device_address: not pinned host memory the device can read");
        return GSR_E_INVALID;
    }
    *device = d;
    return GSR_OK;
}

extern "C" int gsr_stage_step(int32_t n, const float *src, float *dst, int32_t K, int32_t mat_off, int32_t scale_off,
                              float *table, void *stream) {
    if (n < 0 || K < 0 || (n > 0 && (!src || !dst)) || (K > 0 && (!table || mat_off < 0 || mat_off + 16 * (int64_t)K > n ||
                                                                  (scale_off >= 0 && scale_off + (int64_t)K > n)))) {
        gsr_set_error("gsr_stage_step: negative size, null pointer or part matrices outside the staged floats");
        return GSR_E_INVALID;
    }
    const int m = n > K ? n : K;
    if (m == 0) return GSR_OK;
    // (nothing but the launch: the call is capturable into a hipGraph -- the address was resolved by the caller, once)
    hipLaunchKernelGGL(stage_step_kernel, dim3(gsr_div_up(m, GSR_BLOCK)), dim3(GSR_BLOCK), 0, (hipStream_t)stream, n, src, dst, K,
                       mat_off, scale_off, table);
    return gsr_check_launch("stage_step", false, (hipStream_t)stream);
}

extern "C" int gsr_transform_gaussians_batch(int32_t P, int32_t E, const float *xyz, const float *rot,
                                             const float *scaling, const float *semantics, const int32_t *lut,
                                             int32_t lut_size, const float *transforms, int32_t K,
                                             const uint8_t *rescale, float *xyz_out, float *rot_out,
                                             float *scaling_out, void *stream) {
    if (P < 0 || K < 0 || lut_size < 0 || E < 0 || E > 65535) {
        gsr_set_error("gsr_transform_gaussians: negative size (or more than 65535 environments)");
        return GSR_E_INVALID;
    }
    if (P == 0 || E == 0) return GSR_OK;
    if (!xyz || !rot || !semantics || !xyz_out || !rot_out || (lut_size > 0 && !lut) || (K > 0 && !transforms) ||
        ((scaling_out != nullptr) != (scaling != nullptr))) {
        gsr_set_error("gsr_transform_gaussians: null pointer (scaling and scaling_out go together)");
        return GSR_E_INVALID;
    }
    if ((reinterpret_cast<uintptr_t>(rot) | reinterpret_cast<uintptr_t>(rot_out)) & 15u) {
        gsr_set_error("gsr_transform_gaussians: rotation buffers must be 16-byte aligned");
        return GSR_E_INVALID;
    }
    const dim3 grid(gsr_div_up(P, GSR_BLOCK), E);
    if (scaling_out)
        hipLaunchKernelGGL(transform_kernel<true>, grid, dim3(GSR_BLOCK), 0, (hipStream_t)stream, P, xyz, rot, scaling,
                           semantics, lut, lut_size, transforms, K, rescale, xyz_out, rot_out, scaling_out);
    else
        hipLaunchKernelGGL(transform_kernel<false>, grid, dim3(GSR_BLOCK), 0, (hipStream_t)stream, P, xyz, rot, scaling,
                           semantics, lut, lut_size, transforms, K, rescale, xyz_out, rot_out, scaling_out);
    return gsr_check_launch("transform_gaussians", false, (hipStream_t)stream);
}

extern "C" int gsr_transform_gaussians(int32_t P, const float *xyz, const float *rot, const float *semantics,
                                       const int32_t *lut, int32_t lut_size, const float *transforms, int32_t K,
                                       float *xyz_out, float *rot_out, void *stream) {
    return gsr_transform_gaussians_batch(P, 1, xyz, rot, nullptr, semantics, lut, lut_size, transforms, K, nullptr,
                                         xyz_out, rot_out, nullptr, stream);
}
