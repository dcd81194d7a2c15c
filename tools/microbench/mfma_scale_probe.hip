// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950) with e4m3 operands and unit scales: which (lane, byte) of the A / B
// operand is which (row / column, k)?  Fills A[row][k] = small integers that identify k (and B = one-hot columns), runs ONE
// instruction and prints the inferred map.   hipcc --offload-arch=gfx950 -O2 -o probe mfma_scale_probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const unsigned char* a_bytes, const unsigned char* b_bytes, float* out) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = ((const int*)a_bytes)[lane * 8 + i];
        b[i] = ((const int*)b_bytes)[lane * 8 + i];
    }
    f32x16 c = {0};
    // cbsz = 0 (A fp8 e4m3), blgp = 0 (B fp8 e4m3); scales: E8M0 127 = 1.0 in byte 0 (opsel 0)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

// e4m3 encodings of small integers 0..8: 0->0x00, 1->0x38, 2->0x40, 3->0x44, 4->0x48, 5->0x4A, 6->0x4C, 7->0x4E, 8->0x50
static const unsigned char E[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};

int main() {
    std::vector<unsigned char> A(64 * 32), B(64 * 32);
    std::vector<float> out(64 * 16);
    unsigned char *dA, *dB; float* dO;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dO, out.size() * 4);
    // Experiment: for each candidate byte position (lane half h, byte j) set A = 1 at (all lanes with that h, byte j) and B = 1 at
    // (lane with column 0 ... all lanes, h', byte j'): the product is non-zero only if the two bytes are the same k.
    // Build the k-equivalence table T[h][j] x [h'][j'].
    int ok = 1;
    int kmapA[2][32];
    // step 1: find, for A byte (h, j), which B byte (h', j') it pairs with
    for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 32; ++j) {
            int found = -1;
            for (int hb = 0; hb < 2 && found < 0; ++hb)
                for (int jb = 0; jb < 32 && found < 0; ++jb) {
                    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
                    for (int l = 0; l < 64; ++l) { if ((l >> 5) == h) A[l * 32 + j] = E[1]; if ((l >> 5) == hb) B[l * 32 + jb] = E[1]; }
                    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO);
                    hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
                    if (out[0] != 0.f) found = hb * 32 + jb;
                }
            kmapA[h][j] = found;
            if (found != h * 32 + j) ok = 0;
        }
    printf("A byte (h, j) pairs with B byte (h', j'): identity map = %s\n", ok ? "YES" : "NO");
    if (!ok) for (int h = 0; h < 2; ++h) { for (int j = 0; j < 32; ++j) printf("%d ", kmapA[h][j]); printf("\n"); }
    // step 2: rows / columns: A = value (row+1 mod 8 + 1 .. ) at byte (0,0) for every lane of half 0; B one-hot column c
    std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
    for (int l = 0; l < 32; ++l) A[l * 32 + 0] = E[(l % 8) + 1];      // lane l (h = 0) byte 0: value (l%8)+1
    for (int l = 0; l < 32; ++l) B[l * 32 + 0] = E[1];                 // every column gets k0 = 1
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
    // expected C/D map (32x32): lane = column, register i = row (i&3) + 8*(i>>2) + 4*(lane>>5); D[row][col] = (row%8)+1 if A lane = row
    int good = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5); if (out[l * 16 + i] != (float)((row % 8) + 1)) good = 0; }
    printf("A lane (l & 31) = row, B lane (l & 31) = column, standard C/D map: %s\n", good ? "YES" : "NO");
    if (!good) { for (int i = 0; i < 16; ++i) printf("%g ", out[i]); printf("| lane 33: "); for (int i = 0; i < 16; ++i) printf("%g ", out[33 * 16 + i]); printf("\n"); }
    return 0;
}
