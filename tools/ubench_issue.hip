// issue-rate microbenchmark for the compositor's instruction mix (gfx950): tools/scratch/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k(float *out, const float *in, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    float a = in[t & 1023], b = in[(t + 1) & 1023], c = in[(t + 2) & 1023], d = in[(t + 3) & 1023];
    v2f p = {a, b}, q = {c, d}, r = {a, c}, s = {b, d};
    v2f Tp = {1.0f, 0.f};
    unsigned long long alive = ~0ull;
    const unsigned long long full = __builtin_amdgcn_read_exec();
    const float al = 0.02f + 0.0001f * (float)(t & 63);   // alpha in [0.02, 0.0264]: every lane touches, nobody stops soon
    v2f ao = {al, 1.0f - al};
    for (int i = 0; i < iters; i++) {
        if (V == 0) {
            asm volatile(REP16("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\t") : "+v"(a), "+v"(b) : "v"(c), "v"(d));  // 32 VALU
        } else if (V == 1) {
            asm volatile(REP16("v_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %1, %2, %3, %1\n\t") : "+v"(p), "+v"(q) : "v"(r), "v"(s));  // 32 pk
        } else if (V == 2) {
            asm volatile(REP4(REP4("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\t") "v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\t") : "+v"(a), "+v"(b) : "v"(c), "v"(d));  // 40 VALU?  (4 x (8 + 4)) = 48
        } else if (V == 3) {
            // 24 fma + 8 exp
            asm volatile(REP4("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_exp_f32 %4, %2\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_exp_f32 %5, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=v"(p.x), "=v"(p.y));
        } else if (V == 4) {
            // 24 fma + 8 v_cmpx (+ 8 s_mov exec)
            asm volatile(REP4("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_cmpx_le_f32 0x3b808081, %4\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_cmpx_le_f32 0x3b808081, %4\n\ts_mov_b64 exec, %5\n\t")
                         : "+v"(a), "+v"(b) : "v"(c), "v"(d), "v"(al), "s"(full) : "vcc");
        } else if (V == 5) {
            // 24 fma + 8 v_cmp (vcc)
            asm volatile(REP4("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_cmp_le_f32 vcc, 0x3b808081, %4\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_cmp_le_f32 vcc, 0x3b808081, %4\n\t")
                         : "+v"(a), "+v"(b) : "v"(c), "v"(d), "v"(al) : "vcc");
        } else if (V == 6) {
            // 32 fma + 32 SALU interleaved
            unsigned long long t0 = alive;
            asm volatile(REP16("v_fma_f32 %0, %1, %2, %0\n\ts_xor_b64 %4, %4, %5\n\tv_fma_f32 %1, %2, %3, %1\n\ts_andn2_b64 %4, %4, %5\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(t0) : "s"(full) : "scc");
            alive ^= t0;
        } else if (V == 7) {
            // the old masked blend x 4 (8 VALU each)
            unsigned long long sv = full;
            asm volatile(REP4(
                "v_cmpx_le_f32 0x3b808081, %[a]\n\t"
                "v_mul_f32 v95, %[T], %[oma]\n\t"
                "v_cmp_gt_f32 vcc, 0x38d1b717, v95\n\t"
                "v_mul_f32 v94, %[a], %[T]\n\t"
                "s_nop 0\n\t"
                "v_cndmask_b32_e64 v94, v94, 0, vcc\n\t"
                "v_cndmask_b32_e64 %[T], v95, -|%[T]|, vcc\n\t"
                "v_pk_fma_f32 %[rg], %[crg], v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[bd], %[cbd], v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"
                "s_mov_b64 exec, %[sv]\n\t")
                : [T] "+v"(a), [rg] "+v"(p), [bd] "+v"(q)
                : [a] "v"(ao.x), [oma] "v"(ao.y), [crg] "v"(r), [cbd] "v"(s), [sv] "s"(sv) : "vcc", "v94", "v95");
            a = 1.0f;
        } else if (V == 8) {
            // the new alive blend x 4 (6 VALU each)
            unsigned long long tmp;
            asm volatile(REP4(
                "s_mov_b64 exec, %[alive]\n\t"
                "v_cmpx_le_f32 0x3b808081, %[a]\n\t"
                "s_mov_b64 %[tmp], vcc\n\t"
                "v_pk_mul_f32 v[94:95], %[ao], v[92:93] op_sel_hi:[1,0]\n\t"
                "v_cmpx_ngt_f32 0x38d1b717, v95\n\t"
                "v_mov_b32 v92, v95\n\t"
                "v_pk_fma_f32 %[rg], %[crg], v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[bd], %[cbd], v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"
                "s_xor_b64 %[tmp], %[tmp], vcc\n\t"
                "s_andn2_b64 %[alive], %[alive], %[tmp]\n\t")
                "s_mov_b64 exec, %[sv]\n\t"
                : "+{v[92:93]}"(Tp), [rg] "+v"(p), [bd] "+v"(q), [alive] "+s"(alive), [tmp] "=&s"(tmp)
                : [a] "v"(ao.x), [ao] "v"(ao), [crg] "v"(r), [cbd] "v"(s), [sv] "s"(full) : "vcc", "scc", "v94", "v95");
            Tp.x = 1.0f; alive = ~0ull;
        } else if (V == 9) {
            // new blend without the scalar bookkeeping (not correct: issue-rate probe only)
            asm volatile(REP4(
                "v_cmpx_le_f32 0x3b808081, %[a]\n\t"
                "v_pk_mul_f32 v[94:95], %[ao], v[92:93] op_sel_hi:[1,0]\n\t"
                "v_cmpx_ngt_f32 0x38d1b717, v95\n\t"
                "v_mov_b32 v92, v95\n\t"
                "v_pk_fma_f32 %[rg], %[crg], v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[bd], %[cbd], v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"
                "s_mov_b64 exec, %[sv]\n\t")
                : "+{v[92:93]}"(Tp), [rg] "+v"(p), [bd] "+v"(q)
                : [a] "v"(ao.x), [ao] "v"(ao), [crg] "v"(r), [cbd] "v"(s), [sv] "s"(full) : "vcc", "v94", "v95");
            Tp.x = 1.0f;
        } else if (V == 11) {
            // the committed blend x 4: v_cmpx, pk_mul, v_cmp, 2 s_andn2, v_mov, 2 pk_fma, s_mov exec
            asm volatile("s_mov_b64 exec, %[alive]\n\t" REP4(
                "v_cmpx_le_f32 0x3b808081, %[a]\n\t"
                "v_pk_mul_f32 v[94:95], %[ao], v[92:93] op_sel_hi:[1,0]\n\t"
                "v_cmp_gt_f32 vcc, 0x38d1b717, v95\n\t"
                "s_andn2_b64 %[alive], %[alive], vcc\n\t"
                "s_andn2_b64 exec, exec, vcc\n\t"
                "v_mov_b32 v92, v95\n\t"
                "v_pk_fma_f32 %[rg], %[crg], v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[bd], %[cbd], v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"
                "s_mov_b64 exec, %[alive]\n\t")
                "s_mov_b64 exec, %[sv]\n\t"
                : "+{v[92:93]}"(Tp), [rg] "+v"(p), [bd] "+v"(q), [alive] "+s"(alive)
                : [a] "v"(ao.x), [ao] "v"(ao), [crg] "v"(r), [cbd] "v"(s), [sv] "s"(full) : "vcc", "scc", "v94", "v95");
            Tp.x = 1.0f; alive = ~0ull;
        } else if (V == 12) {
            // the same with a branch around the two s_andn2 (no lane stops: the common case)
            asm volatile("s_mov_b64 exec, %[alive]\n\t" REP4(
                "v_cmpx_le_f32 0x3b808081, %[a]\n\t"
                "v_pk_mul_f32 v[94:95], %[ao], v[92:93] op_sel_hi:[1,0]\n\t"
                "v_cmp_gt_f32 vcc, 0x38d1b717, v95\n\t"
                "s_cbranch_vccz 1f\n\t"
                "s_andn2_b64 %[alive], %[alive], vcc\n\t"
                "s_andn2_b64 exec, exec, vcc\n\t"
                "1:\n\t"
                "v_mov_b32 v92, v95\n\t"
                "v_pk_fma_f32 %[rg], %[crg], v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[bd], %[cbd], v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"
                "s_mov_b64 exec, %[alive]\n\t")
                "s_mov_b64 exec, %[sv]\n\t"
                : "+{v[92:93]}"(Tp), [rg] "+v"(p), [bd] "+v"(q), [alive] "+s"(alive)
                : [a] "v"(ao.x), [ao] "v"(ao), [crg] "v"(r), [cbd] "v"(s), [sv] "s"(full) : "vcc", "scc", "v94", "v95");
            Tp.x = 1.0f; alive = ~0ull;
        } else if (V == 13) {
            // T updated in place by the packed multiply (T in v93, weight in v92): no v_mov (valid when T's last value is not read)
            asm volatile("s_mov_b64 exec, %[alive]\n\t" REP4(
                "v_cmpx_le_f32 0x3b808081, %[a]\n\t"
                "v_pk_mul_f32 v[92:93], %[ao], v[92:93] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                "v_cmp_gt_f32 vcc, 0x38d1b717, v93\n\t"
                "s_andn2_b64 %[alive], %[alive], vcc\n\t"
                "s_andn2_b64 exec, exec, vcc\n\t"
                "v_pk_fma_f32 %[rg], %[crg], v[92:93], %[rg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[bd], %[cbd], v[92:93], %[bd] op_sel_hi:[1,0,1]\n\t"
                "s_mov_b64 exec, %[alive]\n\t")
                "s_mov_b64 exec, %[sv]\n\t"
                : "+{v[92:93]}"(Tp), [rg] "+v"(p), [bd] "+v"(q), [alive] "+s"(alive)
                : [a] "v"(ao.x), [ao] "v"(ao), [crg] "v"(r), [cbd] "v"(s), [sv] "s"(full) : "vcc", "scc");
            Tp.y = 1.0f; alive = ~0ull;
        } else if (V == 10) {
            // 6 plain VALU x 4 (reference for 8 / 9)
            asm volatile(REP4("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_pk_fma_f32 %4, %5, %6, %4\n\tv_pk_fma_f32 %5, %6, %4, %5\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q) : "v"(r));
        }
    }
    out[t] = a + b + c + d + p.x + p.y + q.x + q.y + Tp.x + (float)(alive & 1);
}

template <int V>
float run(float *out, const float *in, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(1280), dim3(256), 0, 0, out, in, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(1280), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out, *in;
    hipMalloc(&out, 1280 * 256 * 4); hipMalloc(&in, 4096);
    std::vector<float> h(1024, 0.5f);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    const char *names[] = {"32 v_fma", "32 v_pk_fma", "48 v_fma", "24 fma + 8 exp", "24 fma + 8 cmpx + 4 s_mov exec", "24 fma + 8 v_cmp", "32 fma + 32 salu",
                           "old blend x4 (32 VALU)", "new blend x4 (24 VALU + 20 SALU)", "new blend x4, no salu bookkeeping", "24 plain VALU", "committed blend x4 (24 VALU + 13 SALU)", "committed blend + branch around s_andn2", "in-place T (20 VALU + 13 SALU)"};
    float ms[14];
    ms[0] = run<0>(out, in, iters); ms[1] = run<1>(out, in, iters); ms[2] = run<2>(out, in, iters); ms[3] = run<3>(out, in, iters);
    ms[4] = run<4>(out, in, iters); ms[5] = run<5>(out, in, iters); ms[6] = run<6>(out, in, iters); ms[7] = run<7>(out, in, iters);
    ms[8] = run<8>(out, in, iters); ms[9] = run<9>(out, in, iters); ms[10] = run<10>(out, in, iters); ms[11] = run<11>(out, in, iters); ms[12] = run<12>(out, in, iters); ms[13] = run<13>(out, in, iters);
    // 5 waves per SIMD: cycles per iteration per SIMD = ms * clk / iters; per wave-instruction slot: / (5 * n)
    for (int v = 0; v < 14; v++)
        printf("%-40s %8.3f ms  -> %7.1f ns per iteration of 5 waves = %6.1f cycles @2.4GHz per wave-iteration\n", names[v], ms[v],
               ms[v] * 1e6 / iters, ms[v] * 1e6 / iters * 2.4 / 5.0);
    return 0;
}
