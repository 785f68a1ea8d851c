// ubench3.hip -- does a 3-VGPR-operand instruction pay for VGPR bank conflicts?  v_bitop3_b32 on eight
// independent chains whose three source registers sit in 1, 2 or 3 different banks (bank = index mod 4),
// one wave per SIMD (the chain kernels' situation) and four.  Prints cycles per instruction per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
#define BODY(S1, S2)                                                                         \
    "v_bitop3_b32 v20, v20, " S1 ", " S2 " bitop3:0xd2\n v_bitop3_b32 v24, v24, " S1 ", " S2 " bitop3:0xd2\n" \
    "v_bitop3_b32 v28, v28, " S1 ", " S2 " bitop3:0xd2\n v_bitop3_b32 v32, v32, " S1 ", " S2 " bitop3:0xd2\n" \
    "v_bitop3_b32 v36, v36, " S1 ", " S2 " bitop3:0xd2\n v_bitop3_b32 v40, v40, " S1 ", " S2 " bitop3:0xd2\n" \
    "v_bitop3_b32 v44, v44, " S1 ", " S2 " bitop3:0xd2\n v_bitop3_b32 v48, v48, " S1 ", " S2 " bitop3:0xd2\n"
#define CLOB "v1", "v2", "v4", "v8", "v20", "v24", "v28", "v32", "v36", "v40", "v44", "v48"
#define DEFK(NAME, S1, S2)                                                                   \
    __global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint64_t *cyc, uint32_t seed) \
    {                                                                                        \
        asm volatile("v_mov_b32 v1, %0\n v_mov_b32 v2, %0\n v_mov_b32 v4, %0\n v_mov_b32 v8, %0\n" \
                     "v_mov_b32 v20, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n v_mov_b32 v32, %0\n" \
                     "v_mov_b32 v36, %0\n v_mov_b32 v40, %0\n v_mov_b32 v44, %0\n v_mov_b32 v48, %0\n" \
                     :: "v"(seed + threadIdx.x) : CLOB);                                      \
        uint64_t t0 = clock64();                                                             \
        for (int i = 0; i < ITER; i++) asm volatile(BODY(S1, S2) BODY(S1, S2) ::: CLOB);      \
        uint64_t t1 = clock64();                                                             \
        uint32_t r;                                                                          \
        asm volatile("v_xor_b32 %0, v20, v24\n v_xor_b32 %0, %0, v28" : "=v"(r) :: CLOB);      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                              \
    }
DEFK(k_b3_3banks, "v1", "v2")
DEFK(k_b3_2banks, "v4", "v2")
DEFK(k_b3_1bank, "v4", "v8")
#define BODYA(S1)                                                                            \
    "v_alignbit_b32 v20, v20, " S1 ", 7\n v_alignbit_b32 v24, v24, " S1 ", 7\n v_alignbit_b32 v28, v28, " S1 ", 7\n" \
    "v_alignbit_b32 v32, v32, " S1 ", 7\n v_alignbit_b32 v36, v36, " S1 ", 7\n v_alignbit_b32 v40, v40, " S1 ", 7\n" \
    "v_alignbit_b32 v44, v44, " S1 ", 7\n v_alignbit_b32 v48, v48, " S1 ", 7\n"
#define BODYX(S1)                                                                            \
    "v_xor_b32 v20, v20, " S1 "\n v_xor_b32 v24, v24, " S1 "\n v_xor_b32 v28, v28, " S1 "\n v_xor_b32 v32, v32, " S1 "\n" \
    "v_xor_b32 v36, v36, " S1 "\n v_xor_b32 v40, v40, " S1 "\n v_xor_b32 v44, v44, " S1 "\n v_xor_b32 v48, v48, " S1 "\n"
#define DEFK1(NAME, B)                                                                       \
    __global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint64_t *cyc, uint32_t seed) \
    {                                                                                        \
        asm volatile("v_mov_b32 v1, %0\n v_mov_b32 v4, %0\n v_mov_b32 v20, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n" \
                     "v_mov_b32 v32, %0\n v_mov_b32 v36, %0\n v_mov_b32 v40, %0\n v_mov_b32 v44, %0\n v_mov_b32 v48, %0\n" \
                     :: "v"(seed + threadIdx.x) : CLOB);                                      \
        uint64_t t0 = clock64();                                                             \
        for (int i = 0; i < ITER; i++) asm volatile(B B ::: CLOB);                            \
        uint64_t t1 = clock64();                                                             \
        uint32_t r;                                                                          \
        asm volatile("v_xor_b32 %0, v20, v24" : "=v"(r) :: CLOB);                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                              \
    }
DEFK1(k_align_2banks, BODYA("v1"))
DEFK1(k_align_1bank, BODYA("v4"))
DEFK1(k_xor_2banks, BODYX("v1"))
DEFK1(k_xor_1bank, BODYX("v4"))

template <typename K>
static void run(const char *name, K k, int waves_per_simd)
{
    uint32_t *out; uint64_t *cyc, h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, 5u);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, 5u);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s %d wave(s)/SIMD: %.2f clk per instruction per wave (clock64 counts at 100 MHz? raw %llu)\n", name, waves_per_simd,
           (double)h / (ITER * 16.0), (unsigned long long)h);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : {1, 2, 4})
    {
        run("bitop3 3 banks", k_b3_3banks, w); run("bitop3 2 banks", k_b3_2banks, w); run("bitop3 1 bank", k_b3_1bank, w);
        run("alignbit 2 banks", k_align_2banks, w); run("alignbit 1 bank", k_align_1bank, w);
        run("xor 2 banks", k_xor_2banks, w); run("xor 1 bank", k_xor_1bank, w);
    }
    return 0;
}
