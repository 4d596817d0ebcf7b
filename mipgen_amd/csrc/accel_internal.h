// accel_internal.h — what the host-side translation units of libmipgen_accel.so share: the kernel launchers' prototypes, the error convention,
// device buffers, the handle (struct mipgen_accel) and the helpers that cross files.  accel.hip: lifecycle, model, region batch; accel_tiles.hip:
// tile lists of the scoring kernels; accel_score.hip: scoring / replay / collapse / record text / downloads; accel_kmer.hip: section 8f-3.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "logistic_device.h"

// ---- kernel launchers (other translation units) -------------------------------------------------------
struct LrcMers { int8_t k[MIPGEN_N_LRC]; int8_t code[MIPGEN_N_LRC]; int8_t rc[MIPGEN_N_LRC]; };
struct TileDims;
extern "C" {
size_t mipgen_logistic_lds_bytes(int span);
hipError_t mipgen_launch_records_logistic(hipStream_t, int score, int n_tiles, int span_max, const DevParams*, const DevRegion*,
                                          const LogTile*, const uint8_t*, const int32_t*, const uint8_t*, const HostConsts*,
                                          double*, uint64_t*, int64_t* sat_idx, unsigned int* sat_count, unsigned int sat_cap);
size_t mipgen_svr_lds_bytes_tile(int np, int kc_ss_range, int ssmax, int Lmax, int n_arm, int group, int n_e, int n_l, int n_threads);
int mipgen_svr_scores_fit_lds(int np, int kc, int n_pairs, int ss_range, int ssmax, int Lmax, int n_arm, int group, int n_e, int n_l, int n_threads);
hipError_t mipgen_launch_svr_dense(hipStream_t, int n_tiles, int n_tiles_few, size_t lds_bytes, const DevParams*, const SvrGeom*, const SvrGeom* geom_few,
                                   const DevRegion*, const SvrTile*, const uint8_t*, const int32_t*, const double* log10_tab,
                                   const double* model, int n_sv, double gamma_l2e, double rho, double s_guard,
                                   const uint64_t* records, double* scores, int64_t n_cand, int n_split, double* partials);
hipError_t mipgen_launch_candidates(hipStream_t, int n, const DevParams*, const DevRegion*, const mipgen_candidate*, const uint8_t*,
                                    const int32_t*, const uint8_t*, const HostConsts*, const double* model, int n_sv, double gamma,
                                    double rho, int method, double*, uint64_t*, double*, mipgen_candidate_ints*, int literal, const unsigned int* n_dev);
hipError_t mipgen_launch_print_boundary_scan(hipStream_t, const DevParams*, const DevRegion*, int r0, int r1, const double* scores, const uint64_t* records, int64_t n,
                                             double tol_rel, double tol_abs, mipgen_candidate* out, int64_t* out_idx, unsigned int* count, unsigned int cap, int n_cu);
hipError_t mipgen_launch_print_boundary_scan_list(hipStream_t, const mipgen_candidate* cands, const double* scores, const uint64_t* records, int n, double tol_rel,
                                                  double tol_abs, mipgen_candidate* out, int64_t* out_idx, unsigned int* count, unsigned int cap);
hipError_t mipgen_launch_svr_run_state(hipStream_t s, int64_t n_pos, const DevParams* P, const DevRegion* regions, const int32_t* pos_region, const int32_t* pos_local,
                                       const uint32_t* run_bounds, int max_levels, int level, double margin, const double* scores, const uint64_t* records, double* pbs,
                                       uint8_t* state);
hipError_t mipgen_launch_svr_tile_keep(hipStream_t s, int n_tiles, const SvrTile* tiles, const int64_t* region_pos0, int64_t win_pos0, const uint8_t* state, int64_t* keep);
hipError_t mipgen_launch_svr_tile_compact(hipStream_t s, int n_tiles, const SvrTile* tiles, const int64_t* keep, const int64_t* offs, SvrTile* out, const DevParams* P,
                                          const DevRegion* regions, double* scores, unsigned long long* skipped);
hipError_t mipgen_launch_dense_candidates(hipStream_t, const DevParams* P, const DevRegion* regions, int r0, int r1, int64_t c0, int n, mipgen_candidate* out);
hipError_t mipgen_launch_dense_list_fix(hipStream_t, int n, const uint64_t* records, double rho, double s_guard, double* scores);
hipError_t mipgen_launch_scatter_scores(hipStream_t, const double* src, const int64_t* idx, int cap, const unsigned int* n_dev, double* scores, unsigned int* over);
hipError_t mipgen_launch_print_boundary_scan_surv(hipStream_t, const DevParams*, const DevRegion*, int r0, int r1, const mipgen_survivor* surv, int64_t n, int64_t cand0,
                                                  double tol_rel, double tol_abs, mipgen_candidate* out, int64_t* out_idx, unsigned int* count, unsigned int cap);
hipError_t mipgen_launch_scatter_surv_scores(hipStream_t, const double* src, const int64_t* idx, int cap, const unsigned int* n_dev, mipgen_survivor* surv, unsigned int* over);
hipError_t mipgen_launch_surv_keep(hipStream_t, const mipgen_survivor* surv, int64_t n, int64_t* keep, double* svr);
hipError_t mipgen_launch_surv_candidates(hipStream_t, const DevParams*, const DevRegion*, int r0, int r1, const mipgen_survivor* surv, int64_t n, int64_t cand0,
                                         const int64_t* offs, mipgen_candidate* out, int64_t* out_idx);
hipError_t mipgen_launch_scatter_f64(hipStream_t, const double* src, const int64_t* idx, int64_t n, double* dst);
hipError_t mipgen_launch_long_range(hipStream_t, int n, const char* seqs, const int64_t* offs, const int32_t* lens, const int32_t* denoms,
                                    const LrcMers*, double* out);
hipError_t mipgen_launch_replay_condense(hipStream_t, int n_regions, int total_pos, const DevParams*, int n_pairs, int n_sizes_max, const DevRegion*,
                                         const int32_t* pos_region, const int32_t* pos_local, const double* scores,
                                         const uint64_t* records, const int32_t* copy, int64_t cand_base, uint8_t* emitted,
                                         mipgen_survivor* survivors, unsigned long long* emitted_per_region);
size_t mipgen_logistic_dense_lds_bytes(int np_all, int np, int ssr, int ssmax, int Lmax, int n_up, int n_dn);
hipError_t mipgen_launch_logistic_dense(hipStream_t, int n_tiles, size_t lds_bytes, const DevParams*, const DevRegion*, const SvrTile*, const uint8_t*,
                                        const int32_t*, const uint8_t*, const HostConsts*, double*, uint64_t*, int64_t* sat_idx, unsigned int* sat_count, unsigned int sat_cap);
hipError_t mipgen_launch_index_candidates(hipStream_t, const DevParams*, const DevRegion*, int r0, int r1, const int64_t* idx, const unsigned int* count, unsigned int cap,
                                          mipgen_candidate* out);
struct FmtRegion { int32_t chr_off, chr_len, label_off, label_len, feature_start, feature_stop; int64_t rb0; };
struct FmtConst { char middle[96]; int32_t middle_len; int32_t n_regions; int64_t first_index; };
hipError_t mipgen_launch_fmt_count(hipStream_t, int64_t n_rb, int r0, const FmtConst*, const FmtRegion*, const DevParams*, const DevRegion*, const uint8_t* emitted, int64_t* cnt);
hipError_t mipgen_launch_fmt_records(hipStream_t, int write, int64_t n_rb, int r0, const FmtConst*, const FmtRegion*, const char* pool, const DevParams*, const DevRegion*,
                                     const char* letters, const int32_t* copy, const double* scores, const uint64_t* records, const uint8_t* emitted,
                                     const int64_t* rank0, const int64_t* off, int64_t* len_out, char* text);
hipError_t mipgen_scan_i64(hipStream_t, void* temp, size_t* temp_bytes, const int64_t* in, int64_t* out, int64_t n);
struct KmerParams { int32_t n_k; int32_t k[MIPGEN_MAX_OLIGO]; int32_t kmax; int32_t filter_bits; uint64_t cap_mask; };   // as in kernels_kmer.hip
hipError_t mipgen_launch_features_batch(hipStream_t, int n, const DevParams*, const DevRegion*, const mipgen_candidate*, const uint8_t* bases, const int32_t* copy,
                                        const uint8_t* unmap, const HostConsts*, uint64_t* records, double* features);
hipError_t mipgen_launch_svr_gemm(hipStream_t, int n, const double* feats, const uint64_t* records, const double* model_t, const double* sv_norm,
                                  const double* sv_coef, const double* center, int n_sv_pad, double gamma, double rho, double* scores);
hipError_t mipgen_launch_kmer_insert(hipStream_t, const char* seq, int64_t len, const KmerParams*, uint64_t* keys, uint32_t* filter);
hipError_t mipgen_launch_kmer_fold(hipStream_t, const uint32_t* filter, int filter_bits, uint32_t* folded);
hipError_t mipgen_launch_kmer_count(hipStream_t, const char* genome, int64_t len, const KmerParams*, const uint64_t* keys, const uint32_t* filter,
                                    const uint32_t* folded, unsigned int* counts, int n_cu);
hipError_t mipgen_launch_fill_pos_map(hipStream_t, const int64_t* region_pos0, int n_regions, int64_t total, int32_t* pos_region, int32_t* pos_local);
hipError_t mipgen_launch_kmer_lookup(hipStream_t, const char* seq, int64_t len, const KmerParams*, const uint64_t* keys, const unsigned int* counts, int32_t* out);
hipError_t mipgen_launch_kmer_place(hipStream_t, const int32_t* src, int64_t len, const KmerParams*, const int64_t* roff, int n_regions, int32_t* dst, void* big,
                                    unsigned int* n_big, unsigned int big_cap);
hipError_t mipgen_launch_collapse(hipStream_t, int n_tiles, const CollapseTile* tiles, const DevParams*, const DevRegion*, const int64_t* region_pos0,
                                  const int64_t* region_base0, const mipgen_survivor* survivors, const int32_t* copy, int64_t cand_base, int32_t* collapsed,
                                  int max_scan_all);
}

// ---- errors ----------------------------------------------------------------------------------------------
// (one buffer per calling thread, shared by the translation units of the library: accel.hip defines it)
extern thread_local char g_mipgen_accel_err[512];
#define g_err g_mipgen_accel_err
static inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) return fail(MIPGEN_E_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// consecutive regions whose dense results share the result arrays at one time
struct Window {
    int r0 = 0, r1 = 0;              // regions [r0, r1)
    int64_t cand0 = 0, n_cand = 0;   // batch-wide candidate index of the first candidate; candidates
    int64_t pos0 = 0, n_pos = 0;     // batch-wide scan-position index; positions
    int log_tile0 = 0, n_log_tiles = 0, svr_tile0 = 0, n_svr_tiles = 0, col_tile0 = 0, n_col_tiles = 0, ld_tile0 = 0, n_ld_tiles = 0;
    int n_svr_few = 0;               // the LAST n_svr_few of the window's dense SVR tiles run with the few-sizes thread geometry (regions of one capture size)
    std::vector<int> lvl_tile0;      // dense SVR tiles by capture-size run: run l = svr_tiles_lvl[lvl_tile0[l], lvl_tile0[l + 1])
    int lvl0_few = 0;                // ... the last lvl0_few tiles of run 0 with the few-sizes geometry (such regions have one run)
    int64_t base0 = 0, n_base_entries = 0;   // collapsed entries (2 per base) of the window inside the batch-wide array
};

// Device buffers a handle has let go of, kept for its next allocations.  hipFree of tens of GB returns at once but the release is paid by
// a later hipMalloc (~60 ms per GB, tools/microbench/alloc_cost.hip: seconds for the k-mer tables of an exome); handing the blocks on -
// the counter's tables become the result arrays of the scoring calls - costs nothing.  Whole blocks only, best fit, at most 4x the request.
struct DevPool {
    struct Block { void* p; size_t bytes; };
    std::vector<Block> blocks;
    size_t held() const { size_t n = 0; for (const Block& b : blocks) n += b.bytes; return n; }
    void* take(size_t bytes, size_t* got)
    {
        int best = -1;
        for (int i = 0; i < (int)blocks.size(); i++)
            if (blocks[(size_t)i].bytes >= bytes && blocks[(size_t)i].bytes / 4 <= bytes && (best < 0 || blocks[(size_t)i].bytes < blocks[(size_t)best].bytes)) best = i;
        if (best < 0) return nullptr;
        void* p = blocks[(size_t)best].p;
        *got = blocks[(size_t)best].bytes;
        blocks.erase(blocks.begin() + best);
        return p;
    }
    void give(void* p, size_t bytes) { if (bytes >= ((size_t)1 << 20)) blocks.push_back({p, bytes}); else (void)hipFree(p); }   // small ones are not worth keeping
    void clear() { for (const Block& b : blocks) (void)hipFree(b.p); blocks.clear(); }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevPool* pool = nullptr;                 // where the buffer comes from / goes to (nullptr: hipMalloc / hipFree)
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        release();
        size_t want = n + std::min<size_t>(n / 8, (size_t)1 << 20) + 64;
        if (pool) {
            size_t got = 0;
            if (void* q = pool->take(want * sizeof(T), &got)) { p = (T*)q; cap = got / sizeof(T); return 0; }
        }
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess && pool && !pool->blocks.empty()) {        // out of memory with blocks in hand: give them back and try again
            (void)hipGetLastError();
            pool->clear();
            e = hipMalloc((void**)&p, want * sizeof(T));
        }
        if (e != hipSuccess) { p = nullptr; return fail(MIPGEN_E_NOMEM, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) { if (pool) pool->give(p, cap * sizeof(T)); else (void)hipFree(p); }
        p = nullptr; cap = 0;
    }
};

// two pinned host chunks + their "copy finished" events: large tables cross PCIe as they are packed / unpacked, chunk by chunk
template <typename T>
struct PinnedPair {
    T* buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    hipError_t alloc(size_t n)
    {
        for (int b = 0; b < 2; b++) {
            hipError_t e = hipHostMalloc((void**)&buf[b], std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    hipError_t wait(int b) { hipError_t e = busy[b] ? hipEventSynchronize(done[b]) : hipSuccess; busy[b] = false; return e; }
    ~PinnedPair() { for (int b = 0; b < 2; b++) { if (done[b]) { if (busy[b]) (void)hipEventSynchronize(done[b]); (void)hipEventDestroy(done[b]); } if (buf[b]) (void)hipHostFree(buf[b]); } }
};

struct mipgen_accel {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    mipgen_params params;
    DevParams hp;                    // host copy
    DevParams* dp = nullptr;
    HostConsts* dconsts = nullptr;
    HostConsts hconsts;
    SvrGeom geom;
    SvrGeom geom_few;                // thread geometry of the tiles of regions that keep one capture size: more, shorter arm-pair chunks
    bool have_few = false;           // (twice the chunks, half the lanes per chunk: such a tile fills a quarter of the main geometry's lanes)
    // model
    int n_sv = 0;
    double gamma = 0, rho = 0, s_guard = 0;
    DevBuf<double> model;
    std::string svr_geometry_error;  // why the dense SVR kernel cannot run this parameter set ("" = it can); reported when SVR is requested
    std::string svr_batch_error;     // same, for the resident batch (tile does not fit LDS)
    bool record_tiles_ready = false, logistic_tiles_ready = false, svr_tiles_ready = false;   // tile lists of the resident batch, laid out on first use
    int sv_split = 0;                // 0 = chosen per launch from the tile count; > 0 forced
    int ld_subruns = 0;              // position sub-runs per tile of the dense logistic kernel: 0 = from the batch size; 1..8 forced
    int n_cu = 256;
    // batch
    int n_regions = 0;
    int64_t n_cand = 0;
    int64_t total_pos = 0;
    std::vector<DevRegion> hregions;
    std::vector<mipgen_grid> grids;
    DevBuf<DevRegion> regions;
    DevBuf<uint8_t> bases, unmap;
    DevBuf<char> letters;                     // the region strings as given (record formatting prints them; `bases` keeps only classes)
    // device-side all_mips formatting (section 8f-4)
    DevBuf<FmtRegion> fmt_regions;
    DevBuf<char> fmt_pool, fmt_text, fmt_temp;
    DevBuf<int64_t> fmt_a, fmt_b, fmt_c, fmt_d;
    int64_t fmt_bytes = -1;
    DevPool pool;                             // large buffers the handle let go of (see DevPool)
    DevBuf<int32_t> copy;
    std::vector<int32_t> resident_lens;      // seq_len of the regions whose copy tables mipgen_accel_count_oligo_copies_resident left in `copy`
    std::vector<mipgen_big_copy> big_copies; // ... and their counts >= 65535
    DevBuf<LogTile> log_tiles;
    DevBuf<SvrTile> svr_tiles, ld_tiles;      // dense SVR tiles; tiles of the table-based dense logistic kernel (same shape, own sizes)
    size_t ld_lds = 0;                        // 0: some region does not fit that kernel's LDS -> the per-candidate kernel scores the batch
    int log_span_max = 0;
    size_t svr_lds = 0;
    // result windows: the inputs of every region stay resident; the dense result arrays (16 B per candidate) hold one window of
    // consecutive regions at a time
    int64_t window_cap = 0;          // max candidates per window; 0 = as many as fit in free device memory
    std::vector<int32_t> window_breaks;   // batch indices at which a window must start (mipgen_accel_set_window_breaks), ascending
    std::vector<Window> windows;
    int cur_window = -1;
    DevBuf<double> scores, partials;
    DevBuf<uint64_t> records;
    bool scored = false;
    // replay
    DevBuf<uint8_t> emitted;
    DevBuf<mipgen_survivor> survivors;
    DevBuf<unsigned long long> emitted_per_region;
    DevBuf<int32_t> pos_region, pos_local;
    // collapse: per base and strand the scan-start index of the best survivor covering it
    std::vector<int64_t> h_region_base0;     // first collapsed entry of every region (batch-wide), + total at the end
    DevBuf<int64_t> region_pos0, region_base0;
    DevBuf<CollapseTile> col_tiles;
    DevBuf<int32_t> collapsed;
    // scores on a rounding boundary of the 6 printed digits are re-scored in the reference's operation order (fix_print_boundaries)
    bool print_exact = true;
    double sum_abs_coef = 0.0;
    DevBuf<mipgen_candidate> pb_cands;
    // mixed designs: SVR score of every condensed survivor of the batch (mipgen_accel_rescore_survivors), slot for slot beside `survivors`
    DevBuf<double> surv_svr;
    DevBuf<int64_t> rs_keep, rs_offs, rs_idx;
    DevBuf<int64_t> pb_idx;
    DevBuf<double> pb_scores;
    DevBuf<unsigned int> pb_count;
    // logistic candidates whose b^x lies in [2^53, 2^54) - their score turns on the last bit of the reference's pow (kernels_logistic_dense.hip) - listed by the
    // dense kernel and re-scored in the reference's term order with the correctly rounded power before anything is replayed (rescore_saturated)
    DevBuf<int64_t> sat_idx;
    DevBuf<unsigned int> sat_count;
    DevBuf<mipgen_candidate> sat_cands;
    DevBuf<double> sat_scores;
    // flag image of the last mipgen_accel_window_uniqueness_begin: uint8 [win_sizes][win_total], region r at column win_roff[r]
    DevBuf<uint8_t> win_img;
    std::vector<int64_t> win_roff;
    std::vector<int32_t> win_lens;
    int win_sizes = 0;
    int64_t win_total = 0;
    // dynamic skip between capture-size runs (kernels_skip.hip; mipgen_accel_set_dynamic_skip)
    bool dyn_skip = false;
    int svr_levels = 1;                      // capture-size runs of the region with the most of them (1: nothing to skip between)
    DevBuf<SvrTile> svr_tiles_lvl, svr_tiles_kept;
    DevBuf<uint32_t> run_bounds;             // [region][level]: first size index | sizes << 16
    DevBuf<double> run_pbs;                  // per scan position of the window: previous_best_score after the runs scored so far
    DevBuf<uint8_t> run_state;               // 0 still constructing, 1 stopped (mipgen.cpp:430), 2 too close to the limit to call
    DevBuf<int64_t> run_keep, run_offs;
    DevBuf<unsigned long long> skip_count;   // dense candidates of the tiles skipped since the last read
    unsigned long long skipped_total = 0;
    bool skip_count_valid = false;
    unsigned int* pb_over = nullptr;         // host-mapped word: entries a re-score list could not hold (checked at the next download: pb_check)
    std::vector<uint8_t> win_state;          // per result window: bit 0 = survivors / emitted counts are of the scores it holds now (replayed), bit 1 = collapsed,
                                             // bit 2 = surv_svr holds the SVR scores of its current survivors
    DevBuf<double> model_t, sv_norm, sv_coef, sv_center;   // the model centred and transposed for the survivor-list scorer (kernels_svr_gemm.hip)
    int n_sv_pad = 0;
    double kmer_count_ms = -1.0;             // genome pass of the last mipgen_accel_count_oligo_copies
    double list_feat_ms = -1.0, list_svr_ms = -1.0;   // k_features_batch / k_svr_gemm of the last list call (timing enabled)
    int64_t kmer_genome_bytes = 0;
    bool replayed = false, mask_valid = false;
    // sparse scratch
    DevBuf<mipgen_candidate> cand_in;
    DevBuf<double> cand_scores, cand_feats;
    DevBuf<uint64_t> cand_records;
    DevBuf<mipgen_candidate_ints> cand_ints;
    DevBuf<char> lrc_seq;
    DevBuf<double> lrc_out;
    DevBuf<int64_t> lrc_offs;
    DevBuf<int32_t> lrc_lens, lrc_denoms;
    // timing: four events per window (records | svr | replay), summed over the windows of the last call
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<uint8_t> ev_used;    // per window: bit 0 scored, bit 1 replayed in the last call
};

// ---- small helpers ----------------------------------------------------------------------------------------
static inline int n_sizes_all(const mipgen_params& P)
{
    if (P.max_capture_size < P.min_capture_size) return 0;
    return (P.max_capture_size - P.min_capture_size) / P.capture_increment + 1;
}

static inline void grid_of(const mipgen_params& P, const DevParams& D, const mipgen_region& R, mipgen_grid* g)
{
    // positions: mipgen.cpp:421-425; static size skip: mipgen.cpp:429
    int cur = R.start_flanked - P.max_capture_size + (P.arm_sum_key_max > 0 ? P.arm_sum_key_max : D.max_sum);   // (:421 uses the largest KEY of the arm-sum map)
    if (cur < 0) cur = 0;
    g->first_pos = cur + 1;
    g->n_pos = std::max(0, R.stop_flanked - cur);
    int K = D.n_sizes_all, k0 = 0;
    while (k0 < K) {
        int C = P.max_capture_size - k0 * P.capture_increment;
        if (C > R.stop_flanked - R.start_flanked + P.max_mip_overlap && C - P.capture_increment >= P.min_capture_size) k0++;
        else break;
    }
    g->first_size_index = k0;
    g->n_sizes = K - k0;
    g->count = (int64_t)g->n_pos * g->n_sizes * P.n_arm_pairs * 2;
    g->offset = 0;
}

static inline uint8_t base_code(char c)
{
    switch (c) {
        case 'A': return BASE_A; case 'C': return BASE_C; case 'G': return BASE_G; case 'T': return BASE_T;
        case 'N': return BASE_N; case '-': return BASE_DASH; default: return BASE_OTHER;
    }
}

#ifdef MIPGEN_DIAG
struct DiagClock {                           // host seconds per stage of a call, on stderr (diagnostic builds only)
    const char* what; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit DiagClock(const char* w) : what(w) {}
    void lap(const char* stage) { const auto n_ = std::chrono::steady_clock::now(); fprintf(stderr, "[mipgen_accel] %s: %s %.3f s\n", what, stage, std::chrono::duration<double>(n_ - t).count()); t = n_; }
};
#define DIAG_CLOCK(name) DiagClock diag_clock(name)
#define DIAG_LAP(stage) diag_clock.lap(stage)
#else
#define DIAG_CLOCK(name) do { } while (0)
#define DIAG_LAP(stage) do { } while (0)
#endif

// ---- helpers that cross translation units ----------------------------------------------------------------------
extern "C" {
int mipgen_pick_sv_split(int n_tiles, int n_sv, int n_cu);       // accel_tiles.hip
int mipgen_ensure_events(mipgen_accel* h);
int mipgen_ensure_tiles(mipgen_accel* h, int32_t method);
int mipgen_pb_check(mipgen_accel* h);                            // accel_score.hip
}
