// kernels_svr_gemm.hip — libsvm RBF-SVR of a LIST of candidates from their 192 features, on the FP64 matrix cores (gfx950).
//
// Mixed designs scan with the logistic model and re-score the condensed survivors (2 per scan position: 10^7 candidates for 1,000 x 5 kb)
// with the SVR (/root/reference/mipgen.cpp:1523-1527,1873-1877 -> predict_value :1948-2019 -> svm_predict, svm.cpp:2504-2593, 329-368):
//        score_i = sum_j coef_j exp(-gamma |x_i - s_j|^2) - rho.
// Unlike the dense grid (kernels_svr.hip: neighbouring candidates share sequence windows) a survivor list has no structure to exploit
// between candidates - but [candidates x 192] . [192 x support vectors] is a dense contraction:
//        |x - s|^2 = |x - c|^2 + |s - c|^2 - 2 (x - c).(s - c)
// with c the mean support vector (centring keeps the three terms of the size of the distance itself: the cancellation costs ~1e-15
// relative, against libsvm's own summation order ~1e-13).  The products run on v_mfma_f64_16x16x4f64; a wavefront owns 16 candidates
// x 64 support vectors (four accumulator tiles), its A operands (the centred candidate features) stay in registers, the centred model is
// stored TRANSPOSED in HBM ([feature][support vector], 1.5 MB for 1,024 SVs: L2 resident) so that the B operands are coalesced
// rows, staged through LDS once per workgroup; one exponential per (candidate, SV) follows in the epilogue of each SV tile.
//
// Layout of v_mfma_f64_16x16x4f64 (probed on gfx950): A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[i][j] in lane
// 16 (i mod 4) + j, register i / 4.
#include <hip/hip_runtime.h>
#include "common.h"
#include "device_utils.h"
#include "exp2_coef.h"

#define SG_CANDS 64                  // candidates per workgroup (4 wavefronts x 16)
#define SG_SVT 64                    // support vectors per tile (4 MFMA tiles per wavefront)

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

// 2^t, degree-10 polynomial (exp2_coef.h, 6.7e-16); t <= 0 here, clamped at -1000
__device__ __forceinline__ double exp2_poly(double t)
{
    constexpr double C[11] = EXP2_COEF_10;
    t = fmax(t, -1000.0);
    const double MAGIC = 6755399441055744.0;
    const double m = t + MAGIC;
    const double f = t - (m - MAGIC);
    double p = C[10];
#pragma unroll
    for (int k = 9; k >= 0; k--) p = fma(p, f, C[k]);
    return __hiloint2double(__double2hiint(p) + (__double2loint(m) << 20), __double2loint(p));
}

}  // namespace

// feats [n][192]; records [n] (valid flag); model_t [192][n_sv_pad] centred; sv_norm [n_sv_pad] = |s - c|^2 (+ the squares of libsvm
// indices > 192); sv_coef [n_sv_pad] (0 for the padding); center [192]; n_sv_pad a multiple of SG_SVT.
// A lane keeps its 48 A operands (candidate lane & 15, features (lane >> 4) + 4 t) in registers for the whole support-vector loop.  The B
// operands - the same for the four wavefronts of a workgroup - go through LDS in chunks of SG_KC features x 64 support vectors, double
// buffered (the next chunk's global loads are in flight under the matrix instructions of the current one); the row pitch of 80 doubles
// puts the four feature rows a wavefront reads at once into disjoint bank groups.
#define SG_KC 48                     // features per staged chunk (12 matrix instructions per accumulator tile)
#define SG_BP 80                     // doubles per staged row: 640 B = 32 banks (mod 64)

__global__ __launch_bounds__(256) void k_svr_gemm(int n, const double* __restrict__ feats, const uint64_t* __restrict__ records,
                                                  const double* __restrict__ model_t, const double* __restrict__ sv_norm,
                                                  const double* __restrict__ sv_coef, const double* __restrict__ center, int n_sv_pad,
                                                  double gamma_l2e, double rho, double* __restrict__ scores)
{
    __shared__ double sB[2][SG_KC * SG_BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * SG_CANDS + wave * 16;                 // first candidate of this wavefront
    // this lane's operands: A[i][k] with i = lane & 15, k = lane >> 4; B[k][j] with j = lane & 15
    const int li = lane & 15, lk = lane >> 4;
    const bool have = c0 + li < n;
    const double* xrow = feats + (int64_t)(have ? c0 + li : 0) * MIPGEN_N_FEATURES + lk;
    double xa[MIPGEN_N_FEATURES / 4];
    double q = 0.0;
    int special = 0;                                                  // 1 = a feature is +-inf (every kernel value is 0), 2 = a feature is NaN
#pragma unroll
    for (int t = 0; t < MIPGEN_N_FEATURES / 4; t++) {
        double v = have ? xrow[4 * t] : 0.0;
        const double cj = center[lk + 4 * t];
        if (!(fabs(v) <= 1.7976931348623157e308)) { special |= v != v ? 2 : 1; v = cj; }      // log10(copy <= 0): -inf or NaN (SVMipv4.cpp:109-112)
        v = have ? v - cj : 0.0;
        xa[t] = v;
        q = fma(v, v, q);
    }
    // |x - c|^2 and the special flags of candidate li: over the four lanes li, li + 16, li + 32, li + 48
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    special |= __shfl_xor(special, 16, 64); special |= __shfl_xor(special, 32, 64);
    // D rows of this lane: i = (lane >> 4) + 4 r
    double xr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) xr[r] = __shfl(q, lk + 4 * r, 64);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};             // score partial sums of the lane's four candidates over its support-vector columns

    // staging: chunk g = (SV tile g / 4, feature chunk g % 4) is SG_KC rows of 64 doubles; thread tid copies row (tid >> 6) + 4 j, column tid & 63
    constexpr int NCH = MIPGEN_N_FEATURES / SG_KC;                    // chunks per SV tile
    static_assert(NCH % 2 == 0, "the double buffer alternates with the chunk index");
    constexpr int SR = SG_KC / 4;                                     // rows per thread
    const int n_chunks = (n_sv_pad / SG_SVT) * NCH;
    const int s_row = tid >> 6, s_col = tid & 63;
    double stage[SR];
    auto fetch = [&](int g) {
        const int s0 = (g / NCH) * SG_SVT, k0 = (g % NCH) * SG_KC;
        const double* src = model_t + (int64_t)(k0 + s_row) * n_sv_pad + s0 + s_col;
#pragma unroll
        for (int j = 0; j < SR; j++) stage[j] = src[(int64_t)(4 * j) * n_sv_pad];
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < SR; j++) sB[buf][(s_row + 4 * j) * SG_BP + s_col] = stage[j];
    };
    fetch(0);
    store(0);
    __syncthreads();
    for (int tile = 0; tile < n_sv_pad / SG_SVT; tile++) {
        double4_t d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0}, d2 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NCH; c++) {                               // unrolled: the A operands are addressed statically
            const int g = tile * NCH + c, buf = c & 1;                // NCH is even: chunk g sits in buffer g & 1 = c & 1
            if (g + 1 < n_chunks) fetch(g + 1);                       // in flight under the matrix instructions below
            // volatile: four plain ds_read_b64 per step (2 LDS cycles each); merged into two ds_read2_b64 they occupy the LDS array for 8 cycles each
            typedef volatile __attribute__((address_space(3))) const double lds_vcd;
            lds_vcd* bb = (lds_vcd*)&sB[buf][lk * SG_BP + li];
#pragma unroll
            for (int t = 0; t < SG_KC / 4; t++) {
                lds_vcd* b = bb + (4 * t) * SG_BP;
                const double b0 = b[0], b1 = b[16], b2 = b[32], b3 = b[48];
                d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c * (SG_KC / 4) + t], b0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c * (SG_KC / 4) + t], b1, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c * (SG_KC / 4) + t], b2, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c * (SG_KC / 4) + t], b3, d3, 0, 0, 0);
            }
            if (g + 1 < n_chunks) store(buf ^ 1);                     // the buffer the previous chunk read: every wavefront passed the barrier since
            __syncthreads();
        }
        // epilogue of the SV tile: K = 2^(-gamma log2(e) (|x|^2 + |s|^2 - 2 x.s)); column j = s0 + 16 t + li
        const int s0 = tile * SG_SVT;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double sn = sv_norm[s0 + 16 * t + li], cf = sv_coef[s0 + 16 * t + li];
            const double4_t d = t == 0 ? d0 : (t == 1 ? d1 : (t == 2 ? d2 : d3));
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double d2v = fmax(xr[r] + sn - 2.0 * d[r], 0.0);                   // a squared distance: rounding may leave -1e-17
                acc[r] = fma(cf, exp2_poly(-gamma_l2e * d2v), acc[r]);
            }
        }
    }
    // sum over the 16 lanes that share lane >> 4 (the columns), then one lane per row writes
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double v = acc[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        acc[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r;                                    // candidate c0 + row; its flags sit in lane `row`
        const int sp = __shfl(special, row, 64);
        if (li == 0 && c0 + row < n) {
            double s = acc[r] - rho;
            if (sp & 2) s = __longlong_as_double(0x7FF8000000000000LL);              // a NaN feature poisons every kernel value
            else if (sp & 1) s = -rho;                                               // an infinite distance: every kernel value is 0
            if (!(MIPGEN_REC_FLAGS(records[c0 + row]) & MIPGEN_FLAG_VALID)) s = 0.0; // a candidate the bounds skips remove scores 0, as in k_candidates
            scores[c0 + row] = s;
        }
    }
}

extern "C" hipError_t mipgen_launch_svr_gemm(hipStream_t stream, int n, const double* feats, const uint64_t* records, const double* model_t,
                                             const double* sv_norm, const double* sv_coef, const double* center, int n_sv_pad, double gamma,
                                             double rho, double* scores)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_svr_gemm, dim3((n + SG_CANDS - 1) / SG_CANDS), dim3(256), 0, stream, n, feats, records, model_t, sv_norm, sv_coef, center,
                       n_sv_pad, gamma * 1.4426950408889634074, rho, scores);
    return hipGetLastError();
}
