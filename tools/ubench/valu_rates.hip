// Issue rates of the vector instructions the fused Mix-FFN kernel is made of, per wave, with 1 / 2 / 3 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float floatx2;
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters, unsigned* lds_seed) {
    __shared__ __attribute__((aligned(16))) char sm[32768];
    floatx2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, b = {1.0001f, 0.9999f}, c = {1e-6f, 1e-6f};
    float s0 = threadIdx.x, s1 = 2.f, s2 = 3.f, s3 = 4.f;
    unsigned u0 = threadIdx.x, u1 = 77, u2 = 99, u3 = 1234;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((unsigned*)sm)[i] = i;
    __syncthreads();
    const unsigned la = (threadIdx.x & 63) * 16, lb = 0;
    uint4 r0, r1, r2, r3;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (MODE == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
        if (MODE == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b[0]), "v"(c[0]));) }
        if (MODE == 3) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(s0) : "v"(b[0]), "v"(c[0]));) }
        if (MODE == 4) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));) }
        if (MODE == 5) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));) }
        if (MODE == 6) { REP16(asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %2, 0xffff0000, %1\n v_lshlrev_b32 %3, 16, %4\n v_and_b32 %5, 0xffff0000, %4" : "=v"(u0), "+v"(u1), "=v"(u2), "=v"(u3), "+v"(s1), "=v"(s2));) }
        if (MODE == 7) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2\n v_cvt_pk_bf16_f32 %0, %2, %1\n v_cvt_pk_bf16_f32 %3, %2, %1" : "=v"(u0), "+v"(s1), "+v"(s2), "=v"(u3));) }
        if (MODE == 8) { REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(la));) }
        if (MODE == 9) { REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(lb));) }
        if (MODE == 10) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    }
    if (MODE >= 8 && MODE <= 9) { u0 = r0.x + r1.y + r2.z + r3.w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[0] + a3[1] + s0 + s1 + s2 + s3 + u0 + u1 + u2 + u3;
}
template <int MODE> void run(const char* name, float* out) {
    for (int wps = 1; wps <= 3; ++wps) {
        const int threads = 256 * wps, iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 10, nullptr);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 64;   // per wave
        printf("%-34s waves/SIMD %d: %7.2f ns per wave-instruction, %6.2f ns per SIMD-instruction\n", name, wps, ms * 1e6 / insts, ms * 1e6 / insts / wps);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 768 * 4);
    run<0>("v_pk_fma_f32 (4 independent)", out);
    run<1>("v_pk_fma_f32 (dependent chain)", out);
    run<2>("v_fma_f32 (4 independent)", out);
    run<3>("v_fma_f32 (dependent chain)", out);
    run<4>("v_exp_f32", out);
    run<5>("v_rcp_f32", out);
    run<6>("v_lshlrev / v_and (bf16 widen)", out);
    run<7>("v_cvt_pk_bf16_f32", out);
    run<10>("v_pk_mul / v_pk_add", out);
    run<8>("ds_read_b128 (lane-linear) x4+wait", out);
    run<9>("ds_read_b128 (broadcast) x4+wait", out);
    return 0;
}
