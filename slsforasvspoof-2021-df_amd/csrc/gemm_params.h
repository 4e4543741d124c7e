// Kernel-side argument block and staging constants shared by the MFMA GEMM kernels
// (gemm.hip: 256 x 128 and 128 x BN tiles; gemm_wide.hip: 208 x 256 tiles).
#pragma once
#include "common.h"
#include "kernels.h"

namespace slsmi {

constexpr int BK = 64;             // K-tile depth
constexpr int ROW_BYTES = BK * 2;  // one staged row: 64 16-bit values = a whole 128-byte line

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    void* C;
    long lda, ldw, ldc;
    int M, N, K;
    int ka;  // period of A along K (elements, a multiple of the K-tile): K-tile kt reads A at (kt * 64) mod ka; = K normally
    int mode;
    int nz;
    long zA;
    int zWmod;
    long zW, zBias;
    int zCdiv;
    long zC1, zC2;
    int splitk;
    int ktiles_per_split;
    int grp_in, grp_valid, grp_out;
    const float* dot_w;
    float* dot_out;
    bf16_t* pre_out;  // GEMM_BF16_GELU on the 128-row kernel: the value before the GELU goes here as well (same layout as C)
    const float* R;  // residual modes: the old values are read here (same layout as C; R == C: in place)
    // LayerNorm folded into the GEMMs around it (GemmArgs::ln_mode; 208 x 256 kernel only)
    int ln_mode;
    const float* ln_proxy;  // [M][2]
    const float* ln_vec;    // emit: gamma [N]; apply: g [N]
    bf16_t* ln_a;           // emit: the next GEMM's operand [M][N]
    float* ln_slots;        // [16][M][2]: written by emit, read by apply
    float* ln_stat;         // apply: [M][2] out (may be null)
    float ln_eps;
    int ln_cgroups;         // apply: workgroups per tile row (set by the launcher)
    const float* ln_aff;    // apply on the 128-row / 256 x 128 kernels: [M][2] = (alpha, -ubar); then ln_vec = g, ln_c = c
    const float* ln_c;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8 share an L2),
// so a plain (x, y) map spreads every operand panel over all eight 4 MiB L2s.
// Re-number so that each XCD walks ONE contiguous range of a tile order whose
// concurrently resident tiles form a compact patch (8 tile rows x all-needed
// tile columns, rows fastest).  Affinity only: any placement computes the same tiles.
__device__ __forceinline__ void xcd_tile_map(int lin, int tiles_n, int tiles_m, int& tile_m, int& tile_n) {
    const int total = tiles_n * tiles_m;
    const int q = total >> 3, r = total & 7;
    const int xcd = lin & 7, j = lin >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    constexpr int GM = 8;
    const int band = v / (GM * tiles_n);
    const int rem = v - band * (GM * tiles_n);
    const int band_rows = min(GM, tiles_m - band * GM);
    tile_n = rem / band_rows;
    tile_m = band * GM + (rem - tile_n * band_rows);
}

// The 208 x 256 kernel (gemm_wide.hip).  m_step: rows between the first rows of consecutive tile rows
// (<= 208; a tile still loads and multiplies 208 rows, and stores the first m_step of them).
bool gemm_wide_supported(const GemmParams& p);
int gemm_wide_row_step(int M);
long gemm_wide_tiles(int M, int N, int m_step);
// LN_APPLY launches: workgroups per tile row (0: the shape has more tile rows than the device has CUs)
int gemm_wide_apply_cgroups(int M, int N, int m_step, int n_cu);
hipError_t launch_gemm_wide(const GemmParams& p, int m_step, int fmt, int n_cu, hipStream_t s);

}  // namespace slsmi
