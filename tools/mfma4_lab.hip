// mfma4_lab.hip -- what v_mfma_f64_4x4x4_4b_f64 is and what it costs on this chip (round 3,
// per-plate inverse of the missing-data block):
//   (1) operand / result lane layout, probed with unit inputs (one non-zero lane in A, one in B);
//   (2) issue cost per wavefront-instruction relative to v_fma_f64 and v_mfma_f64_16x16x4_f64,
//       alone and interleaved (do fp64 vector and 4x4x4 matrix instructions overlap?);
//   (3) cost of v_mov_b64 DPP row_newbcast and of ds_read_b128 broadcasts beside fp64 FMAs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma4_lab.hip -o tools/mfma4_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(64) probe(double *out)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + l] = d;
        }
}

enum { K_FMA = 0, K_M4, K_M16, K_MIX_M4_FMA, K_MIX_M16_FMA, K_DPP_FMA, K_LDS_FMA, K_MIX_M4_DPP,
       K_M4_LDS2, K_M4_LDS1, K_M4_BITS, K_M4_VAR, K_M4_VAR16, K_M16_VAR, K_M4_VGPR, K_M4_VGPR_BITS };

template <int KIND>
__global__ void __launch_bounds__(256) loop(double *out, int iters)
{
    __shared__ __attribute__((aligned(16))) double sh[512];
    const int l = threadIdx.x;
    sh[l] = l * 1e-3;
    sh[256 + l] = l * 2e-3;
    __syncthreads();
    double a = l * 1e-3, b = 1.0 + l * 1e-9;
    double x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    v4f64 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4f64{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
        } else if (KIND == K_M4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[i], 0, 0, 0);
        } else if (KIND == K_M16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        } else if (KIND == K_MIX_M4_FMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[i], 0, 0, 0);
                x[i] = __builtin_fma(x[i], b, a);
            }
        } else if (KIND == K_MIX_M16_FMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                x[2 * i] = __builtin_fma(x[2 * i], b, a);
                x[2 * i + 1] = __builtin_fma(x[2 * i + 1], b, a);
            }
        } else if (KIND == K_DPP_FMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double r = __builtin_amdgcn_update_dpp(0.0, x[(i + 1) & 7], 0x150 + 3, 0xf, 0xf, true);
                x[i] = __builtin_fma(r, b, x[i]);
            }
        } else if (KIND == K_LDS_FMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v2f64 r = *reinterpret_cast<const v2f64 *>(&sh[((it + i) & 63) * 2 + (l >> 4) * 128]);
                x[2 * i] = __builtin_fma(r.x, b, x[2 * i]);
                x[2 * i + 1] = __builtin_fma(r.y, b, x[2 * i + 1]);
            }
        } else if (KIND == K_M4_LDS2) {
            // A operand of two MFMAs from one ds_read_b64 (16 distinct values, each read by 4 lanes)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double av = sh[((it + i) & 7) * 40 + (l >> 4) * 20 + (l & 3) + 4 * (i & 3)];
                m[2 * i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, m[2 * i], 0, 0, 0);
                m[2 * i + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, a, m[2 * i + 1], 0, 0, 0);
            }
        } else if (KIND == K_M4_LDS1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double av = sh[((it + i) & 7) * 40 + (l >> 4) * 20 + (l & 3) + 4 * (i & 3)];
                m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, m[i], 0, 0, 0);
            }
        } else if (KIND == K_M4_BITS) {
            // A operand from a mask bit: bfe + cvt per two MFMAs
            const unsigned mw = (unsigned)(it * 2654435761u) ^ (unsigned)l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double av = (double)((mw >> (4 * i + (it & 3))) & 1u);
                m[2 * i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b, m[2 * i], 0, 0, 0);
                m[2 * i + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, a, m[2 * i + 1], 0, 0, 0);
            }
        } else if (KIND == K_M4_VAR) {
            // distinct A / B registers per instruction (no operand repeats back to back)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[(i * 3) & 7], x[(i * 5 + 1) & 7], m[i], 0, 0, 0);
        } else if (KIND == K_M4_VAR16) {
            // 16 accumulators, A changes every 2 instructions, B alternates (the GEMM stage pattern)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[i], a, m[i], 0, 0, 0);
                acc[i >> 1][2 * (i & 1)] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[i], b, acc[i >> 1][2 * (i & 1)], 0, 0, 0);
            }
        } else if (KIND == K_M16_VAR) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[i + 4], acc[i], 0, 0, 0);
        } else if (KIND == K_M4_VGPR) {
            // accumulators forced into VGPRs (the compiler's choice in the 2-waves-per-SIMD kernels)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[i]) : "v"(x[i]), "v"(b));
        } else if (KIND == K_M4_VGPR_BITS) {
            const unsigned mw = (unsigned)(it * 2654435761u) ^ (unsigned)l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double av = (double)((mw >> (4 * i + (it & 3))) & 1u);
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[2 * i]) : "v"(av), "v"(b));
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(m[2 * i + 1]) : "v"(av), "v"(a));
            }
        } else if (KIND == K_MIX_M4_DPP) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[i], 0, 0, 0);
                x[i] = __builtin_amdgcn_update_dpp(0.0, x[(i + 1) & 7], 0x150 + 3, 0xf, 0xf, true);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + m[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + l] = s;
}

// the inner loop of the GEMM stages: NA x 2 accumulators, A operand from a mask bit per pair of
// instructions, B operands from registers; BITS = 0: A operand from a register instead
template <int NA, int BITS>
__global__ void __launch_bounds__(256, 2) gemm_loop(double *out, int iters)
{
    const int l = threadIdx.x;
    double acc[NA][2];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i][0] = acc[i][1] = 0.0;
    double bx[4], by[4], ar[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bx[k] = 1.0 + l * 1e-3 * (k + 1);
        by[k] = 1.0 - l * 1e-3 * (k + 1);
        ar[k] = (l >> k) & 1;
    }
    for (int it = 0; it < iters; ++it) {
        const unsigned mw = (unsigned)(it * 2654435761u) ^ (unsigned)(l * 40503u);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const double am = BITS ? (double)((mw >> ((i + 7 * k) & 31)) & 1u) : ar[(i + k) & 3];
                acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(am, bx[k], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(am, by[k], acc[i][1], 0, 0, 0);
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1];
    out[(size_t)blockIdx.x * 256 + l] = s;
}

template <int NA, int BITS>
static void run_gemm(const char *name, double *out, int wpsimd)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    gemm_loop<NA, BITS><<<256 * wpsimd, 256>>>(out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    gemm_loop<NA, BITS><<<256 * wpsimd, 256>>>(out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s waves/SIMD %d  %8.3f ms   %7.2f ns per mfma per SIMD\n", name, wpsimd, ms,
           ms * 1e6 / ((double)wpsimd * iters * 8 * NA));
}

template <int KIND>
static void run(const char *name, double *out, int per_iter_main, int per_iter_other, int wpsimd)
{
    const int iters = 20000;
    const int grid = 256 * wpsimd;        // 4 wavefronts per workgroup: one per SIMD of a CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    loop<KIND><<<grid, 256>>>(out, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    loop<KIND><<<grid, 256>>>(out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: wpsimd wavefronts x iters x instructions
    const double n_main = (double)wpsimd * iters * per_iter_main;
    const double n_other = (double)wpsimd * iters * per_iter_other;
    printf("%-34s waves/SIMD %d  %8.3f ms   %7.2f ns per main instr per SIMD (%d main + %d other per iter)\n",
           name, wpsimd, ms, ms * 1e6 / n_main, per_iter_main, per_iter_other);
    (void)n_other;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 64 * 64 * 64 + sizeof(double) * 256 * 256 * 16));
    // ---- (1) layout ---------------------------------------------------------------------------
    probe<<<1, 64>>>(out);
    CK(hipDeviceSynchronize());
    double *h = (double *)malloc(sizeof(double) * 64 * 64 * 64);
    CK(hipMemcpy(h, out, sizeof(double) * 64 * 64 * 64, hipMemcpyDeviceToHost));
    // hypothesis: A[b][i][k] in lane 16 b + 4 k + i, B[b][k][j] in lane 16 b + 4 k + j,
    // D[b][i][j] in lane 16 b + 4 i + j  (and every alternative with i/k, k/j, i/j swapped)
    const char *names[8] = {"A(i+4k) B(j+4k) D(j+4i)", "A(i+4k) B(j+4k) D(i+4j)", "A(i+4k) B(k+4j) D(j+4i)",
                            "A(i+4k) B(k+4j) D(i+4j)", "A(k+4i) B(j+4k) D(j+4i)", "A(k+4i) B(j+4k) D(i+4j)",
                            "A(k+4i) B(k+4j) D(j+4i)", "A(k+4i) B(k+4j) D(i+4j)"};
    for (int hyp = 0; hyp < 8; ++hyp) {
        int ok = 1;
        for (int la = 0; la < 64 && ok; ++la)
            for (int lb = 0; lb < 64 && ok; ++lb) {
                const int ba = la >> 4, bb = lb >> 4;
                int ia, ka, kb, jb;
                if (hyp & 4) { ka = la & 3; ia = (la >> 2) & 3; } else { ia = la & 3; ka = (la >> 2) & 3; }
                if (hyp & 2) { kb = lb & 3; jb = (lb >> 2) & 3; } else { jb = lb & 3; kb = (lb >> 2) & 3; }
                for (int l = 0; l < 64; ++l) {
                    double want = 0.0;
                    if (ba == bb && ka == kb) {
                        const int dl = 16 * ba + ((hyp & 1) ? (ia + 4 * jb) : (jb + 4 * ia));
                        want = (l == dl) ? 1.0 : 0.0;
                    }
                    if (h[(la * 64 + lb) * 64 + l] != want) { ok = 0; break; }
                }
            }
        printf("layout hypothesis %-26s : %s\n", names[hyp], ok ? "MATCHES" : "no");
    }
    // raw dump: every (A lane, B lane) pair with a non-zero result
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            int any = 0;
            for (int l = 0; l < 64; ++l)
                if (h[(la * 64 + lb) * 64 + l] != 0.0) any = 1;
            if (!any) continue;
            printf("P %2d %2d :", la, lb);
            for (int l = 0; l < 64; ++l)
                if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" %d", l);
            printf("\n");
        }
    if (getenv("LAB_PROBE_ONLY")) return 0;
    if (getenv("LAB_GEMM")) {
        double *og = out + 64 * 64 * 64;
        run_gemm<8, 0>("gemm loop 16 acc, A from registers", og, 2);
        run_gemm<8, 1>("gemm loop 16 acc, A from mask bits", og, 2);
        run_gemm<16, 0>("gemm loop 32 acc, A from registers", og, 2);
        run_gemm<16, 1>("gemm loop 32 acc, A from mask bits", og, 2);
        run_gemm<32, 0>("gemm loop 64 acc, A from registers", og, 2);
        run_gemm<32, 1>("gemm loop 64 acc, A from mask bits", og, 2);
        run_gemm<36, 1>("gemm loop 72 acc, A from mask bits", og, 2);
        return 0;
    }
    const int wmin = getenv("LAB_W4") ? 2 : 1;
    // ---- (2), (3) issue costs -----------------------------------------------------------------
    double *o2 = out + 64 * 64 * 64;
    for (int w = wmin; w <= 4; w *= 2) {
        run<K_FMA>("v_fma_f64 x8", o2, 8, 0, w);
        run<K_M4>("mfma_f64_4x4x4_4b x8", o2, 8, 0, w);
        run<K_M16>("mfma_f64_16x16x4 x4", o2, 4, 0, w);
        run<K_MIX_M4_FMA>("8 x (mfma4x4x4 + v_fma_f64)", o2, 8, 8, w);
        run<K_MIX_M16_FMA>("4 x (mfma16x16x4 + 2 v_fma_f64)", o2, 4, 8, w);
        run<K_DPP_FMA>("8 x (v_mov_b64_dpp + v_fma_f64)", o2, 8, 8, w);
        run<K_LDS_FMA>("4 x (ds_read_b128 + 2 v_fma_f64)", o2, 8, 4, w);
        run<K_MIX_M4_DPP>("8 x (mfma4x4x4 + v_mov_b64_dpp)", o2, 8, 8, w);
        run<K_M4_LDS2>("4 x (ds_read_b64 + 2 mfma4x4x4)", o2, 8, 4, w);
        run<K_M4_LDS1>("8 x (ds_read_b64 + mfma4x4x4)", o2, 8, 8, w);
        run<K_M4_BITS>("4 x (bfe + cvt + 2 mfma4x4x4)", o2, 8, 8, w);
        run<K_M4_VAR>("mfma4x4x4 x8, distinct A/B regs", o2, 8, 0, w);
        run<K_M4_VAR16>("mfma4x4x4 x16, A per pair, B alternating", o2, 16, 0, w);
        run<K_M16_VAR>("mfma16x16x4 x4, distinct A/B regs", o2, 4, 0, w);
        run<K_M4_VGPR>("mfma4x4x4 x8, VGPR accumulators", o2, 8, 0, w);
        run<K_M4_VGPR_BITS>("4 x (bfe + cvt + 2 mfma4x4x4), VGPR acc", o2, 8, 8, w);
    }
    return 0;
}
