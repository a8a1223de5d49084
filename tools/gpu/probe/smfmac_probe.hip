// Probe of the operand layout of v_smfmac_f32_16x16x64_bf16 on gfx950: which dense (k, j) element of B each compressed A
// slot of each lane multiplies for a given index field.  One wave; results printed as tables.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k64(const unsigned short* A, const unsigned short* B, const int* idx, float* C, int abid) {
    const int l = threadIdx.x;
    bf16x8 a = *(const bf16x8*)(A + l * 8);
    bf16x16 b = *(const bf16x16*)(B + l * 16);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (abid == 0) c = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, idx[l], 0, 0);
    else c = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, idx[l], 0, 1);
    *(f32x4*)(C + l * 4) = c;
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    unsigned short *dA, *dB; int* dI; float* dC;
    hipMalloc(&dA, 64 * 8 * 2); hipMalloc(&dB, 64 * 16 * 2); hipMalloc(&dI, 64 * 4); hipMalloc(&dC, 64 * 4 * 4);
    std::vector<unsigned short> A(64 * 8), B(64 * 16);
    std::vector<int> I(64);
    std::vector<float> C(256);
    auto run = [&](int abid) {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dI, I.data(), 64 * 4, hipMemcpyHostToDevice);
        k64<<<1, 64>>>(dA, dB, dI, dC, abid);
        hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    };
    // 1. C layout + which B lane/element: A one-hot at (lane La, slot s), idx = v in every field
    for (int La : {0, 5, 16, 37, 63}) {
        for (int s = 0; s < 8; ++s) {
            for (int v = 0; v < 4; ++v) {
                std::fill(A.begin(), A.end(), 0);
                A[La * 8 + s] = f2bf(1.f);
                int field = 0;
                for (int f = 0; f < 16; ++f) field |= v << (2 * f);
                std::fill(I.begin(), I.end(), field);
                // pass 1: B = lane id; pass 2: B = element id + 1
                int blane = -1, belem = -1, crow_lane = -1, creg = -1, nz = 0;
                for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) B[l * 16 + e] = f2bf((float)(l + 1));
                run(0);
                for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (C[l * 4 + r] != 0.f) {
                    ++nz;
                    if ((l & 15) == 0 && crow_lane < 0) { crow_lane = l; creg = r; blane = (int)C[l * 4 + r] - 1; }
                }
                for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) B[l * 16 + e] = f2bf((float)(e + 1));
                run(0);
                if (crow_lane >= 0) belem = (int)C[crow_lane * 4 + creg] - 1;
                printf("A lane %2d slot %d idx %d -> nonzero C entries %2d; column 0 sits in C lane %2d reg %d; B lane %2d elem %2d\n",
                       La, s, v, nz, crow_lane, creg, blane, belem);
            }
        }
    }
    // 2. which index field belongs to which slot: all fields 0 except field f = 3
    for (int s = 0; s < 8; ++s)
        for (int f = 0; f < 16; ++f) {
            std::fill(A.begin(), A.end(), 0);
            A[0 * 8 + s] = f2bf(1.f);
            std::fill(I.begin(), I.end(), 3 << (2 * f));
            for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) B[l * 16 + e] = f2bf((float)(e + 1));
            run(0);
            float got = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if ((l & 15) == 0 && C[l * 4 + r] != 0.f) got = C[l * 4 + r];
            if ((int)got - 1 != ((s >> 1) * 4)) printf("slot %d: field %2d moves the pick to B elem %d\n", s, f, (int)got - 1);
        }
    // 3. abid = 1: which fields then
    for (int s = 0; s < 8; ++s)
        for (int f = 0; f < 16; ++f) {
            std::fill(A.begin(), A.end(), 0);
            A[0 * 8 + s] = f2bf(1.f);
            std::fill(I.begin(), I.end(), 3 << (2 * f));
            for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) B[l * 16 + e] = f2bf((float)(e + 1));
            run(1);
            float got = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if ((l & 15) == 0 && C[l * 4 + r] != 0.f) got = C[l * 4 + r];
            if ((int)got - 1 != ((s >> 1) * 4)) printf("abid 1 slot %d: field %2d moves the pick to B elem %d\n", s, f, (int)got - 1);
        }
    // 4. random check against the hypothesis: lane (i = l & 15, kq = l >> 4); A slot s -> dense k = 16 kq + 4 (s >> 1) + idx_s,
    //    idx_s = bits [2s+1 : 2s]; B lane (j, kq) elem e -> dense k = 16 kq + e; C lane (j, q) reg r -> C[4 q + r][j]
    srand(1);
    for (auto& a : A) a = f2bf((float)(rand() % 17 - 8));
    for (auto& b : B) b = f2bf((float)(rand() % 17 - 8));
    for (int l = 0; l < 64; ++l) {
        int field = 0;
        for (int g = 0; g < 4; ++g) {
            int p0 = rand() % 3, p1 = p0 + 1 + rand() % (3 - p0);
            field |= (p0 | (p1 << 2)) << (4 * g);
        }
        I[l] = field | (rand() << 16);
    }
    run(0);
    double maxerr = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int kq = 0; kq < 4; ++kq) for (int s = 0; s < 8; ++s) {
            const int la = kq * 16 + i, lb = kq * 16 + j;
            const int ix = (I[la] >> (2 * s)) & 3;
            ref += (double)bf2f(A[la * 8 + s]) * bf2f(B[lb * 16 + 4 * (s >> 1) + ix]);
        }
        const float got = C[((i >> 2) * 16 + j) * 4 + (i & 3)];
        maxerr = fmax(maxerr, fabs(ref - got));
    }
    printf("random check against the hypothesis: max abs error %g\n", maxerr);
    return 0;
}
