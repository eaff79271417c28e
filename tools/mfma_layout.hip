// Operand / result layout probe for v_mfma_f64_4x4x4_4b_f64 (gfx950): which lane holds which (block, row, k) of A,
// (block, k, col) of B and (block, row, col) of D.  For every source lane s: A = [lane == s], B = lane + 1; the
// non-zero results name the lanes of s's block and row, their values the B-lanes of s's k and the result's column.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_layout tools/mfma_layout.hip && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    for (int s = 0; s < 64; ++s) {
        const double a = (lane == s) ? 1.0 : 0.0, b = lane + 1.0;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[s * 64 + lane] = d;
    }
}
int main() {
    double* d;
    hipMalloc(&d, sizeof(double) * 4096);
    probe<<<1, 64>>>(d);
    double h[4096];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int s = 0; s < 64; ++s) {
        printf("A-lane %2d ->", s);
        for (int l = 0; l < 64; ++l)
            if (h[s * 64 + l] != 0.0) printf("  D-lane %2d = B-lane %2d", l, (int)h[s * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}
