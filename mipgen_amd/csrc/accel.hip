// accel.hip — host side of libmipgen_accel.so: the C-ABI of include/mipgen_accel.h over the gfx950 kernels.
// No CPU scoring path exists in this library: every entry point either drives the GPU or fails.
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
                                          double*, uint64_t*);
size_t mipgen_svr_lds_bytes_tile(int np, int kc_ss_range, int ssmax, int Lmax, int n_arm, int group, int n_e, int n_l, int n_threads);
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
                                        const int32_t*, const uint8_t*, const HostConsts*, double*, uint64_t*);
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
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...)
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
static int n_sizes_all(const mipgen_params& P)
{
    if (P.max_capture_size < P.min_capture_size) return 0;
    return (P.max_capture_size - P.min_capture_size) / P.capture_increment + 1;
}

static void grid_of(const mipgen_params& P, const DevParams& D, const mipgen_region& R, mipgen_grid* g)
{
    // positions: mipgen.cpp:421-425; static size skip: mipgen.cpp:429
    int cur = R.start_flanked - P.max_capture_size + D.max_sum;
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

static uint8_t base_code(char c)
{
    switch (c) {
        case 'A': return BASE_A; case 'C': return BASE_C; case 'G': return BASE_G; case 'T': return BASE_T;
        case 'N': return BASE_N; case '-': return BASE_DASH; default: return BASE_OTHER;
    }
}

// ---- C ABI ---------------------------------------------------------------------------------------------------
extern "C" {

int mipgen_accel_abi_version(void) { return MIPGEN_ACCEL_ABI_VERSION; }
const char* mipgen_accel_last_error(void) { return g_err; }

int mipgen_accel_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mipgen_accel_create(const mipgen_params* params, int device, void* stream, mipgen_accel** out)
{
    if (!params || !out) return fail(MIPGEN_E_INVALID, "null argument");
    *out = nullptr;
    if (params->abi_version != MIPGEN_ACCEL_ABI_VERSION) return fail(MIPGEN_E_INVALID, "abi_version %d != %d", params->abi_version, MIPGEN_ACCEL_ABI_VERSION);
    if (params->n_arm_pairs < 1 || params->n_arm_pairs > MIPGEN_MAX_ARM_PAIRS) return fail(MIPGEN_E_INVALID, "n_arm_pairs out of range");
    if (params->capture_increment <= 0) return fail(MIPGEN_E_INVALID, "capture_increment must be positive (the reference coerces 0 to 1 before this call)");
    if (params->max_capture_size < params->min_capture_size) return fail(MIPGEN_E_INVALID, "max_capture_size < min_capture_size");
    int ndev = mipgen_accel_device_count();
    if (ndev <= 0) return fail(MIPGEN_E_NODEVICE, "no HIP device: libmipgen_accel has no CPU path");
    if (device < 0 || device >= ndev) return fail(MIPGEN_E_INVALID, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    mipgen_accel* h = new mipgen_accel();
    // the buffers that reach gigabytes share the handle's pool
    h->copy.pool = h->scores.pool = h->partials.pool = h->records.pool = h->emitted.pool = h->survivors.pool = h->collapsed.pool = h->pos_region.pool =
        h->pos_local.pool = h->bases.pool = h->letters.pool = h->unmap.pool = h->fmt_pool.pool = h->fmt_text.pool = h->fmt_temp.pool = &h->pool;
    h->device = device;
    h->params = *params;
    DevParams& D = h->hp;
    memset(&D, 0, sizeof D);
    D.min_capture = params->min_capture_size; D.max_capture = params->max_capture_size; D.inc = params->capture_increment;
    D.n_sizes_all = n_sizes_all(*params);
    D.n_pairs = params->n_arm_pairs;
    D.check_copy_number = params->check_copy_number; D.logistic_heuristic = params->logistic_heuristic;
    D.score_method = params->score_method;
    D.masked_arm_threshold = params->masked_arm_threshold; D.upper = params->upper_score_limit; D.lower = params->lower_score_limit;
    D.max_arm_copy_product = params->max_arm_copy_product; D.target_arm_copy = params->target_arm_copy;
    D.min_sum = 1 << 30; D.max_sum = 0; D.e_min = D.l_min = 1 << 30; D.e_max = D.l_max = 0;
    for (int i = 0; i <= MIPGEN_MAX_OLIGO; i++) D.len_slot[i] = -1;
    for (int i = 0; i < D.n_pairs; i++) {
        int e = params->arm_ext[i], l = params->arm_lig[i];
        if (e < 2 || l < 2 || e > MIPGEN_MAX_OLIGO || l > MIPGEN_MAX_OLIGO) { delete h; return fail(MIPGEN_E_INVALID, "arm length out of range [2,%d]", MIPGEN_MAX_OLIGO); }
        D.arm_ext[i] = (uint8_t)e; D.arm_lig[i] = (uint8_t)l;
        D.min_sum = std::min(D.min_sum, e + l); D.max_sum = std::max(D.max_sum, e + l);
        D.e_min = std::min(D.e_min, e); D.e_max = std::max(D.e_max, e);
        D.l_min = std::min(D.l_min, l); D.l_max = std::max(D.l_max, l);
        D.len_slot[e] = 0; D.len_slot[l] = 0;
    }
    if (params->min_capture_size - D.max_sum < 1) { delete h; return fail(MIPGEN_E_INVALID, "min_capture_size leaves no scan target (capture size <= arm length sum)"); }
    // limits of the dense SVR kernel are reported when an SVR launch is requested: logistic designs are not bound by them
    if (params->min_capture_size - D.max_sum < 3) h->svr_geometry_error = "min_capture_size leaves a scan size < 3 (insert 3-mer frequencies are undefined)";
    int slot = 0;
    for (int i = 0; i <= MIPGEN_MAX_OLIGO; i++) if (D.len_slot[i] == 0) D.len_slot[i] = (int8_t)slot++;
    D.n_len_slots = slot;
    for (int i = 0; i < D.n_pairs;) {                         // arm-sum lists are contiguous runs (mipgen.cpp:431-438)
        int j = i, s = D.arm_ext[i] + D.arm_lig[i];
        while (j < D.n_pairs && D.arm_ext[j] + D.arm_lig[j] == s) j++;
        for (int k = i; k < j; k++) D.group_end[k] = (uint16_t)j;
        i = j;
    }
    // SVR thread geometry
    SvrGeom& G = h->geom;
    memset(&G, 0, sizeof G);
    G.nchunk = (D.n_pairs + SVR_MAX_CHUNK - 1) / SVR_MAX_CHUNK;
    G.chunk_len = (D.n_pairs + G.nchunk - 1) / G.nchunk;
    G.n_e = D.e_max - D.e_min + 1; G.n_l = D.l_max - D.l_min + 1;
    G.group = SVR_GROUP;
    G.wpc = 4;                                                   // waves per arm-pair chunk
    while (G.wpc > 1 && G.nchunk * G.wpc * 64 > SVR_MAX_THREADS) G.wpc--;
    if (G.nchunk * G.wpc * 64 > SVR_MAX_THREADS) {
        char msg[128];
        snprintf(msg, sizeof msg, "too many arm pairs for the dense SVR kernel (max %d)", SVR_MAX_CHUNK * (SVR_MAX_THREADS / 64));
        h->svr_geometry_error = msg;
        G.wpc = 1;
    }
    // Regions that keep ONE capture size after the static skip of mipgen.cpp:429 (short exons: half of an exome BED) hold few (position, size)
    // items per tile - the tile's positions are bound by its LDS, not by the lanes -, so with the main geometry half of the wavefronts own no
    // candidate while the others walk fifteen pairs each.  Their tiles run with twice the chunks of half the length on two wavefronts each.
    h->geom_few = G;
    h->have_few = false;
    if (h->svr_geometry_error.empty() && G.wpc == 4 && G.chunk_len > 8) {
        SvrGeom& F = h->geom_few;
        const int max_chunks = SVR_MAX_THREADS / 64 / 2;
        F.chunk_len = (D.n_pairs + max_chunks - 1) / max_chunks;
        F.nchunk = (D.n_pairs + F.chunk_len - 1) / F.chunk_len;
        F.wpc = 2;
        h->have_few = F.chunk_len < G.chunk_len && F.nchunk * F.wpc * 64 <= SVR_MAX_THREADS;
    }
    {
        const int n_arm = std::max(G.n_e, G.n_l) | 1;            // table row pitch in slots (as in the kernel)
        for (int strand = 0; strand < 2; strand++) {
            const int up_min = strand ? D.l_min : D.e_min, dn_min = strand ? D.e_min : D.l_min;
            for (int i = 0; i < MIPGEN_MAX_ARM_PAIRS + 16; i++) {
                const int a = std::min(i, D.n_pairs - 1);
                const int e = D.arm_ext[a], l = D.arm_lig[a];
                const uint32_t uU = (uint32_t)(((strand ? l : e) - up_min) * SVR_GROUP * 8);
                const uint32_t uD = (uint32_t)(((D.max_sum - (e + l)) * n_arm + ((strand ? e : l) - dn_min)) * SVR_GROUP * 8);
                D.pair_prog[strand][i] = uU | (uD << 16);
            }
        }
    }

    if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(MIPGEN_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    for (int i = 0; i < 101; i++) h->hconsts.log10_tab[i] = log10((double)i);    // host libm, as the reference computes it
    h->hconsts.ln_base = log(MIPGEN_LOGISTIC_BASE);
    hipError_t e1 = hipMalloc((void**)&h->dp, sizeof(DevParams));
    hipError_t e2 = hipMalloc((void**)&h->dconsts, sizeof(HostConsts));
    if (e1 != hipSuccess || e2 != hipSuccess) { mipgen_accel_destroy(h); return fail(MIPGEN_E_NOMEM, "hipMalloc failed"); }
    if (hipHostMalloc((void**)&h->pb_over, sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) { h->pb_over = nullptr; mipgen_accel_destroy(h); return fail(MIPGEN_E_NOMEM, "hipHostMalloc failed"); }
    *h->pb_over = 0u;
    hipError_t e3 = hipMemcpy(h->dp, &D, sizeof D, hipMemcpyHostToDevice);
    if (e3 == hipSuccess) e3 = hipMemcpy(h->dconsts, &h->hconsts, sizeof(HostConsts), hipMemcpyHostToDevice);
    if (e3 != hipSuccess) { mipgen_accel_destroy(h); return fail(MIPGEN_E_HIP, "hipMemcpy of the run parameters: %s", hipGetErrorString(e3)); }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cu = cus; }
    *out = h;
    return MIPGEN_OK;
}

void mipgen_accel_destroy(mipgen_accel* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->model.release(); h->model_t.release(); h->sv_norm.release(); h->sv_coef.release(); h->sv_center.release(); h->regions.release(); h->bases.release(); h->unmap.release(); h->copy.release();
    h->log_tiles.release(); h->svr_tiles.release(); h->ld_tiles.release(); h->scores.release(); h->records.release();
    h->emitted.release(); h->survivors.release(); h->emitted_per_region.release(); h->pos_region.release(); h->pos_local.release();
    h->letters.release(); h->fmt_regions.release(); h->fmt_pool.release(); h->fmt_text.release(); h->fmt_temp.release();
    h->fmt_a.release(); h->fmt_b.release(); h->fmt_c.release(); h->fmt_d.release();
    h->region_pos0.release(); h->region_base0.release(); h->col_tiles.release(); h->collapsed.release();
    h->cand_in.release(); h->cand_scores.release(); h->cand_feats.release(); h->cand_records.release(); h->cand_ints.release();
    h->win_img.release();
    h->svr_tiles_lvl.release(); h->svr_tiles_kept.release(); h->run_bounds.release(); h->run_pbs.release(); h->run_state.release(); h->run_keep.release();
    h->run_offs.release(); h->skip_count.release();
    h->lrc_seq.release(); h->lrc_out.release(); h->lrc_offs.release(); h->lrc_lens.release(); h->lrc_denoms.release(); h->partials.release();
    h->pool.clear();
    if (h->dp) (void)hipFree(h->dp);
    if (h->dconsts) (void)hipFree(h->dconsts);
    if (h->pb_over) (void)hipHostFree(h->pb_over);
    for (hipEvent_t e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// ---- model ----------------------------------------------------------------------------------------------------
// The list scorer's view of the model (kernels_svr_gemm.hip): support vectors centred on their mean and transposed to [feature][SV], the
// centred squared norms (+ the squares of libsvm indices > 192, which candidates never have: svm.cpp:359-363), the coefficients; the SV
// count padded to a multiple of 64 with zero-coefficient columns.
static int build_list_model(mipgen_accel* h, const std::vector<double>& rows, int n_sv)
{
    const int pad = std::max(64, (n_sv + 63) / 64 * 64);
    std::vector<double> center(MIPGEN_N_FEATURES, 0.0), mt((size_t)MIPGEN_N_FEATURES * pad, 0.0), sn((size_t)pad, 0.0), cf((size_t)pad, 0.0);
    for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
        double s = 0.0;
        for (int i = 0; i < n_sv; i++) s += rows[(size_t)i * SV_ROW + j];
        center[(size_t)j] = n_sv ? s / n_sv : 0.0;
    }
    for (int i = 0; i < n_sv; i++) {
        double q = 0.0;
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
            const double v = rows[(size_t)i * SV_ROW + j] - center[(size_t)j];
            mt[(size_t)j * pad + i] = v;
            q += v * v;
        }
        sn[(size_t)i] = q + rows[(size_t)i * SV_ROW + SVR_N_EXTRA];
        cf[(size_t)i] = rows[(size_t)i * SV_ROW + SVR_COEF];
    }
    if (h->model_t.reserve(mt.size()) || h->sv_norm.reserve(sn.size()) || h->sv_coef.reserve(cf.size()) || h->sv_center.reserve(center.size())) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpy(h->model_t.p, mt.data(), mt.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sv_norm.p, sn.data(), sn.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sv_coef.p, cf.data(), cf.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sv_center.p, center.data(), center.size() * sizeof(double), hipMemcpyHostToDevice));
    h->n_sv_pad = pad;
    return MIPGEN_OK;
}

int mipgen_accel_set_model(mipgen_accel* h, int32_t n_sv, double gamma, double rho, const double* coef, const double* sv)
{
    if (!h || n_sv < 0 || (n_sv > 0 && (!coef || !sv))) return fail(MIPGEN_E_INVALID, "bad model arguments");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<double> rows((size_t)std::max(n_sv, 1) * SV_ROW, 0.0);
    double s_guard = 0.0;
    for (int i = 0; i < n_sv; i++) {
        double* r = &rows[(size_t)i * SV_ROW];
        const double* x = sv + (size_t)i * MIPGEN_N_FEATURES;
        double ne = 0, ni = 0, nl = 0, nj = 0, tot = 0;
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
            r[j] = x[j];
            double q = x[j] * x[j];
            tot += q;                                     // index order, as the sparse merge walk accumulates (svm.cpp:329-368)
            if (j <= 20) ne += q;
            else if (j >= F_INS && j <= 150) ni += q;
            else if (j >= F_LIG && j <= 172) nl += q;
            else if (j >= F_JUNC && j <= 189) nj += q;
        }
        r[SVR_COEF] = coef[i]; r[SVR_N_EXT] = ne; r[SVR_N_INS] = ni; r[SVR_N_LIG] = nl; r[SVR_N_JUNC] = nj;
        r[SVR_N_TOTAL] = tot; r[SVR_N_EXTRA] = 0.0;
        // what one base adds to a block's 1-mer window sum: its 1-mer value, plus the GC-fraction value for C and G (mer order A, AA, .., C, ..:
        // SVMipv4.cpp:69-70; the GC fraction sits before the "T" entry)
        const int blk_base[3] = {F_EXT, F_INS, F_LIG}, blk_gc[3] = {F_EXT_GC, F_INS_GC, F_LIG_GC}, blk_step[3] = {5, 21, 5}, blk_thr[3] = {15, 63, 15};
        for (int b = 0; b < 3; b++)
            for (int c = 0; c < 4; c++) {
                const int idx = blk_step[b] * c;
                r[SVR_C1 + 4 * b + c] = x[blk_base[b] + idx + (idx >= blk_thr[b])] + ((c == BASE_C || c == BASE_G) ? x[blk_gc[b]] : 0.0);
            }
        s_guard += coef[i] * exp(-gamma * tot);
    }
    if (h->model.reserve(rows.size())) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpy(h->model.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    h->n_sv = n_sv; h->gamma = gamma; h->rho = rho; h->s_guard = s_guard - rho;
    h->sum_abs_coef = 0.0;
    for (int i = 0; i < n_sv; i++) h->sum_abs_coef += fabs(coef[i]);
    return build_list_model(h, rows, n_sv);
}

// libsvm 3.17 text model (grammar: svm.cpp:2779-2962)
int mipgen_accel_load_model_file(mipgen_accel* h, const char* path)
{
    if (!h || !path) return fail(MIPGEN_E_INVALID, "null argument");
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(MIPGEN_E_MODEL, "cannot open model file %s (the reference would dereference NULL here, svm.cpp:2507)", path);
    char cmd[128];
    std::string svm_type, kernel_type;
    double gamma = 0, rho = 0;
    int nr_class = 2, total_sv = -1;
    bool ok = true, seen_sv = false;
    while (ok && fscanf(fp, "%127s", cmd) == 1) {
        std::string c(cmd);
        if (c == "svm_type") { ok = fscanf(fp, "%127s", cmd) == 1; svm_type = cmd; }
        else if (c == "kernel_type") { ok = fscanf(fp, "%127s", cmd) == 1; kernel_type = cmd; }
        else if (c == "degree") { int d; ok = fscanf(fp, "%d", &d) == 1; }
        else if (c == "gamma") ok = fscanf(fp, "%lf", &gamma) == 1;
        else if (c == "coef0") { double d; ok = fscanf(fp, "%lf", &d) == 1; }
        else if (c == "nr_class") ok = fscanf(fp, "%d", &nr_class) == 1;
        else if (c == "total_sv") ok = fscanf(fp, "%d", &total_sv) == 1;
        else if (c == "rho") { int n = nr_class * (nr_class - 1) / 2; for (int i = 0; i < n && ok; i++) { double r; ok = fscanf(fp, "%lf", &r) == 1; if (i == 0) rho = r; } }
        else if (c == "label" || c == "nr_sv") { for (int i = 0; i < nr_class && ok; i++) { int d; ok = fscanf(fp, "%d", &d) == 1; } }
        else if (c == "probA" || c == "probB") { int n = nr_class * (nr_class - 1) / 2; for (int i = 0; i < n && ok; i++) { double d; ok = fscanf(fp, "%lf", &d) == 1; } }
        else if (c == "SV") { int ch; while ((ch = getc(fp)) != EOF && ch != '\n') {} seen_sv = true; break; }
        else { fclose(fp); return fail(MIPGEN_E_MODEL, "unknown text in model file: [%s]", cmd); }
    }
    if (!ok || !seen_sv || total_sv < 0) { fclose(fp); return fail(MIPGEN_E_MODEL, "malformed model header in %s", path); }
    if (svm_type != "epsilon_svr" && svm_type != "nu_svr") { fclose(fp); return fail(MIPGEN_E_MODEL, "svm_type %s unsupported (SVR only)", svm_type.c_str()); }
    if (kernel_type != "rbf") { fclose(fp); return fail(MIPGEN_E_MODEL, "kernel_type %s unsupported (rbf only)", kernel_type.c_str()); }
    std::vector<double> sv((size_t)std::max(total_sv, 1) * MIPGEN_N_FEATURES, 0.0), coef((size_t)std::max(total_sv, 1), 0.0), extra((size_t)std::max(total_sv, 1), 0.0);
    char* line = nullptr; size_t cap = 0;
    int got = 0;
    while (got < total_sv && getline(&line, &cap, fp) >= 0) {
        char* save = nullptr;
        char* p = strtok_r(line, " \t\n", &save);
        if (!p) continue;
        coef[got] = strtod(p, nullptr);
        for (int k = 1; k < nr_class - 1; k++) strtok_r(nullptr, " \t", &save);
        for (;;) {
            char* idx = strtok_r(nullptr, ":", &save);
            char* val = strtok_r(nullptr, " \t\n", &save);
            if (!val) break;
            long j = strtol(idx, nullptr, 10);
            double v = strtod(val, nullptr);
            if (j >= 1 && j <= MIPGEN_N_FEATURES) sv[(size_t)got * MIPGEN_N_FEATURES + (j - 1)] = v;
            else if (j > MIPGEN_N_FEATURES) extra[got] += v * v;   // x has no such index: contributes sv^2 (svm.cpp:359-363)
        }
        got++;
    }
    free(line);
    fclose(fp);
    if (got != total_sv) return fail(MIPGEN_E_MODEL, "model file has %d SV lines, header says %d", got, total_sv);
    int rc = mipgen_accel_set_model(h, total_sv, gamma, rho, coef.data(), sv.data());
    if (rc) return rc;
    bool any_extra = false;
    for (double x : extra) any_extra |= x != 0.0;
    if (any_extra) {
        std::vector<double> rows((size_t)total_sv * SV_ROW);
        HIP_TRY(hipMemcpy(rows.data(), h->model.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
        double sg = 0;
        for (int i = 0; i < total_sv; i++) {
            rows[(size_t)i * SV_ROW + SVR_N_EXTRA] = extra[i];
            rows[(size_t)i * SV_ROW + SVR_N_TOTAL] += extra[i];
            sg += coef[i] * exp(-gamma * rows[(size_t)i * SV_ROW + SVR_N_TOTAL]);
        }
        HIP_TRY(hipMemcpy(h->model.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
        h->s_guard = sg - rho;
        return build_list_model(h, rows, total_sv);
    }
    return MIPGEN_OK;
}

int mipgen_accel_model_info(const mipgen_accel* h, int32_t* n_sv, double* gamma, double* rho)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (n_sv) *n_sv = h->n_sv;
    if (gamma) *gamma = h->gamma;
    if (rho) *rho = h->rho;
    return MIPGEN_OK;
}

// ---- region batch ---------------------------------------------------------------------------------------------
int mipgen_accel_set_window_candidates(mipgen_accel* h, int64_t max_candidates)
{
    if (!h || max_candidates < 0) return fail(MIPGEN_E_INVALID, "bad arguments");
    h->window_cap = max_candidates;
    return MIPGEN_OK;
}

int mipgen_accel_set_print_exact(mipgen_accel* h, int32_t on)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->print_exact = on != 0;
    return MIPGEN_OK;
}

int mipgen_accel_set_logistic_subruns(mipgen_accel* h, int32_t n)
{
    if (!h || n < 0 || n > 8) return fail(MIPGEN_E_INVALID, "bad arguments");
    h->ld_subruns = n;
    h->logistic_tiles_ready = false;                                   // laid out again on the next logistic call
    return MIPGEN_OK;
}

int mipgen_accel_set_sv_split(mipgen_accel* h, int32_t n_split)
{
    if (!h || n_split < 0 || n_split > 64) return fail(MIPGEN_E_INVALID, "n_split must be in [0, 64] (0 = automatic)");
    h->sv_split = n_split;
    return MIPGEN_OK;
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

int mipgen_accel_upload_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, mipgen_grid* grids_out)
{
    if (!h || n < 0 || (n > 0 && !regions)) return fail(MIPGEN_E_INVALID, "bad arguments");
    DIAG_CLOCK("upload");
    HIP_TRY(hipSetDevice(h->device));
    const mipgen_params& P = h->params;
    const DevParams& D = h->hp;
    h->cur_window = -1;
    h->windows.clear();
    h->hregions.assign((size_t)n, DevRegion());
    h->grids.assign((size_t)n, mipgen_grid());
    int64_t seq_total = 0, copy_total = 0, unmap_total = 0, cand_total = 0, pos_total = 0, cand_max = 0;
    // copy tables the handle counted itself (mipgen_accel_count_oligo_copies_resident): the batch must be the one they were counted for
    int n_resident = 0;
    for (int i = 0; i < n; i++) n_resident += regions[i].copy == MIPGEN_COPY_RESIDENT;
    if (n_resident) {
        bool same = n_resident == n && h->resident_lens.size() == (size_t)n;
        for (int i = 0; same && i < n; i++) same = h->resident_lens[(size_t)i] == regions[i].seq_len;
        if (!same) return fail(MIPGEN_E_INVALID, "MIPGEN_COPY_RESIDENT: the batch is not the one mipgen_accel_count_oligo_copies_resident counted (regions, order, seq_len)");
    } else {
        h->resident_lens.clear();                                      // `copy` is about to be overwritten (or unused)
    }
    for (int i = 0; i < n; i++) {
        const mipgen_region& R = regions[i];
        if (!R.seq || R.seq_len <= 0) return fail(MIPGEN_E_INVALID, "region %d has no sequence", i);
        if (R.seq_stop - R.seq_start + 1 != R.seq_len) return fail(MIPGEN_E_INVALID, "region %d: seq_len %d != seq_stop-seq_start+1 = %d", i, R.seq_len, R.seq_stop - R.seq_start + 1);
        DevRegion& d = h->hregions[i];
        mipgen_grid& g = h->grids[i];
        grid_of(P, D, R, &g);
        g.offset = cand_total;
        d.seq_off = seq_total;
        d.seq_len = R.seq_len; d.seq_start = R.seq_start; d.seq_stop = R.seq_stop;
        d.start_fl = R.start_flanked; d.stop_fl = R.stop_flanked;
        d.first_pos = g.first_pos; d.n_pos = g.n_pos; d.k0 = g.first_size_index; d.n_sizes = g.n_sizes;
        d.copy_off = R.copy ? copy_total : -1;
        d.unmap_off = R.unmappable ? unmap_total : -1;
        memcpy(d.lrc, R.long_range_content, sizeof d.lrc);
        seq_total += R.seq_len;
        if (R.copy) copy_total += (int64_t)D.n_len_slots * R.seq_len;
        if (R.unmappable) unmap_total += (int64_t)D.n_sizes_all * R.seq_len;
        cand_total += g.count;
        pos_total += g.n_pos;
        cand_max = std::max(cand_max, g.count);
    }
    if (pos_total > INT32_MAX) return fail(MIPGEN_E_INVALID, "batch has %lld scan positions (max %d): split the design", (long long)pos_total, INT32_MAX);
    DIAG_LAP("layout");
    // encode + pack on the host
    uint8_t code_lut[256];
    for (int c = 0; c < 256; c++) code_lut[c] = base_code((char)c);
    std::vector<uint8_t> hb((size_t)std::max<int64_t>(seq_total, 1));
    std::vector<char> hl((size_t)std::max<int64_t>(seq_total, 1));
    std::vector<uint8_t> hu((size_t)std::max<int64_t>(unmap_total, 1));
    for (int i = 0; i < n; i++) {
        const mipgen_region& R = regions[i];
        const DevRegion& d = h->hregions[i];
        uint8_t* b = &hb[(size_t)d.seq_off];
        memcpy(&hl[(size_t)d.seq_off], R.seq, (size_t)R.seq_len);
        const char* ms = R.masked_seq ? R.masked_seq : R.seq;
        if (R.snp_class) for (int k = 0; k < R.seq_len; k++) b[k] = (uint8_t)(code_lut[(uint8_t)R.seq[k]] | (ms[k] == 'N' ? BASE_MASKED_BIT : 0) | ((R.snp_class[k] & 3) << BASE_SNP_SHIFT));
        else for (int k = 0; k < R.seq_len; k++) b[k] = (uint8_t)(code_lut[(uint8_t)R.seq[k]] | (ms[k] == 'N' ? BASE_MASKED_BIT : 0));
        if (R.unmappable) memcpy(&hu[(size_t)d.unmap_off], R.unmappable, (size_t)D.n_sizes_all * R.seq_len);
    }
    DIAG_LAP("pack");
    // inputs -> HBM (everything the kernels read stays resident for the whole batch)
    if (h->letters.reserve(hl.size())) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->letters.p, hl.data(), hl.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->regions.reserve((size_t)std::max(n, 1)) || h->bases.reserve(hb.size()) || h->copy.reserve((size_t)std::max<int64_t>(copy_total, 1)) || h->unmap.reserve(hu.size()) ||
        h->survivors.reserve((size_t)std::max<int64_t>(2 * pos_total, 1)) || h->emitted_per_region.reserve((size_t)std::max(n, 1)) ||
        h->pos_region.reserve((size_t)std::max<int64_t>(pos_total, 1)) || h->pos_local.reserve((size_t)std::max<int64_t>(pos_total, 1)))
        return MIPGEN_E_NOMEM;

    DIAG_LAP("input buffers");
    // ---- result windows: consecutive regions, at most `cap` candidates each (17 B per candidate: score, record, emitted flag) ----
    int64_t cap = h->window_cap;
    if (cap <= 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const int64_t have = (int64_t)(h->scores.cap * 8 + h->records.cap * 8 + h->emitted.cap) + (int64_t)free_b + (int64_t)h->pool.held();   // what the result arrays may grow into
        cap = std::max<int64_t>(1, (int64_t)((double)have * 0.85) / 17);
        // 2^30 candidates keep every kernel of a window busy for milliseconds to tenths of a second; result arrays beyond that only cost
        // allocation time (18 GB: 0.4-1 s on a cold device; 36 GB: 2 s) and delay the first window a pipelined caller can work on
        // (hipMalloc of 250 GB takes seconds) - unless one region alone is larger
        cap = std::min<int64_t>(cap, std::max<int64_t>((int64_t)1 << 30, cand_max));
    }
    if (cand_max > cap && h->window_cap > 0) cap = cand_max;       // a region is never split: one oversized region is its own window
    if (cand_max > cap) return fail(MIPGEN_E_NOMEM, "a single region has %lld dense candidates; device memory holds %lld", (long long)cand_max, (long long)cap);
    {
        Window w;
        for (int i = 0; i < n; i++) {
            const mipgen_grid& g = h->grids[i];
            if (w.r1 > w.r0 && w.n_cand + g.count > cap) {
                h->windows.push_back(w);
                Window nw; nw.r0 = nw.r1 = i; nw.cand0 = w.cand0 + w.n_cand; nw.pos0 = w.pos0 + w.n_pos;
                w = nw;
            }
            h->hregions[i].out_off = w.n_cand;                     // window-relative: the result arrays are reused window after window
            w.n_cand += g.count; w.n_pos += g.n_pos; w.r1 = i + 1;
        }
        h->windows.push_back(w);                                   // an empty batch is one empty window
    }
    int64_t win_cand_max = 0;
    for (const Window& w : h->windows) win_cand_max = std::max(win_cand_max, w.n_cand);

    // ---- collapse layout: bases a survivor's scan target can cover, per region ----
    std::vector<int64_t> rpos0((size_t)n + 1, 0);
    h->h_region_base0.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) {
        const DevRegion& d = h->hregions[i];
        const int max_scan = D.max_capture - d.k0 * D.inc - D.min_sum;
        const int64_t nb = (d.n_pos > 0 && d.n_sizes > 0) ? (int64_t)d.n_pos + max_scan - 1 : 0;
        rpos0[(size_t)i + 1] = rpos0[(size_t)i] + d.n_pos;
        h->h_region_base0[(size_t)i + 1] = h->h_region_base0[(size_t)i] + 2 * nb;
    }
    // collapse tiles: 128 bases each.  The scoring kernels' tile lists depend on the method and are laid out on first use (ensure_tiles)
    std::vector<CollapseTile> ct;
    for (Window& w : h->windows) {
        w.col_tile0 = (int)ct.size();
        w.base0 = h->h_region_base0[(size_t)w.r0]; w.n_base_entries = h->h_region_base0[(size_t)w.r1] - w.base0;
        w.log_tile0 = w.n_log_tiles = w.svr_tile0 = w.n_svr_tiles = w.ld_tile0 = w.n_ld_tiles = 0;
        for (int i = w.r0; i < w.r1; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            for (int64_t j0 = 0; 2 * j0 < h->h_region_base0[(size_t)i + 1] - h->h_region_base0[(size_t)i]; j0 += 128) { CollapseTile t = {i, (int32_t)j0}; ct.push_back(t); }
        }
        w.n_col_tiles = (int)ct.size() - w.col_tile0;
    }
    h->record_tiles_ready = h->logistic_tiles_ready = h->svr_tiles_ready = false;
    h->svr_batch_error.clear();
    h->ld_lds = 0; h->log_span_max = 0; h->svr_lds = 0;
    DIAG_LAP("tiles");
    if (h->scores.reserve((size_t)std::max<int64_t>(win_cand_max, 1)) || h->records.reserve((size_t)std::max<int64_t>(win_cand_max, 1)) ||
        h->emitted.reserve((size_t)std::max<int64_t>(win_cand_max, 1)))
        return MIPGEN_E_NOMEM;
    if (h->region_pos0.reserve((size_t)n + 1) || h->region_base0.reserve((size_t)n + 1) || h->col_tiles.reserve(std::max<size_t>(ct.size(), 1)) ||
        h->collapsed.reserve((size_t)std::max<int64_t>(h->h_region_base0[(size_t)n], 1)))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->region_pos0.p, rpos0.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->region_base0.p, h->h_region_base0.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    if (!ct.empty()) HIP_TRY(hipMemcpyAsync(h->col_tiles.p, ct.data(), ct.size() * sizeof(CollapseTile), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mipgen_launch_fill_pos_map(h->stream, h->region_pos0.p, n, pos_total, h->pos_region.p, h->pos_local.p));       // position -> (region, position in it)
    if (n > 0) HIP_TRY(hipMemcpyAsync(h->regions.p, h->hregions.data(), (size_t)n * sizeof(DevRegion), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->bases.p, hb.data(), hb.size(), hipMemcpyHostToDevice, h->stream));
    if (copy_total && !n_resident) {
        // copy tables: packed region by region into two pinned chunks and sent as they fill - no host image of the whole table (6.7 GB for the exome)
        int64_t max_block = 0;
        for (int i = 0; i < n; i++) if (regions[i].copy) max_block = std::max(max_block, (int64_t)D.n_len_slots * regions[i].seq_len);
        const int64_t chunk = std::min(copy_total, std::max<int64_t>((int64_t)1 << 20, max_block));          // int32 elements: 4 MiB, or the largest region's block
        PinnedPair<int32_t> pin;
        HIP_TRY(pin.alloc((size_t)chunk));
        int cur = 0;
        int64_t fill = 0, base = 0;
        auto flush = [&]() -> hipError_t {
            if (fill == 0) return hipSuccess;
            hipError_t e = hipMemcpyAsync(h->copy.p + base, pin.buf[cur], (size_t)fill * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = hipEventRecord(pin.done[cur], h->stream);
            if (e != hipSuccess) return e;
            pin.busy[cur] = true;
            base += fill; fill = 0; cur ^= 1;
            return pin.wait(cur);
        };
        for (int i = 0; i < n; i++) {
            const mipgen_region& R = regions[i];
            if (!R.copy) continue;
            const int64_t block = (int64_t)D.n_len_slots * R.seq_len;
            if (fill + block > chunk) HIP_TRY(flush());
            if (h->hregions[i].copy_off != base + fill) return fail(MIPGEN_E_INVALID, "internal: copy table offsets out of order");
            for (int len = 0; len <= MIPGEN_MAX_OLIGO; len++) {
                const int s = D.len_slot[len];
                if (s < 0) continue;
                int32_t* dst = pin.buf[cur] + fill + (int64_t)s * R.seq_len;
                if (R.copy[len]) memcpy(dst, R.copy[len], (size_t)R.seq_len * sizeof(int32_t));
                else memset(dst, 0, (size_t)R.seq_len * sizeof(int32_t));
            }
            fill += block;
        }
        HIP_TRY(flush());
        HIP_TRY(pin.wait(0)); HIP_TRY(pin.wait(1));
    }
    if (unmap_total) HIP_TRY(hipMemcpyAsync(h->unmap.p, hu.data(), hu.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));             // host staging vectors die here
    DIAG_LAP("buffers + copies");
    h->n_regions = n; h->n_cand = cand_total; h->total_pos = pos_total;
    h->scored = false; h->replayed = false; h->win_state.assign(h->windows.size(), 0); h->fmt_bytes = -1;
    h->ev_used.assign(h->windows.size(), 0);
#ifdef MIPGEN_DIAG
    fprintf(stderr, "[mipgen_accel] batch: %d regions, %lld candidates in %zu window(s)\n", n, (long long)cand_total, h->windows.size());
#endif
    if (grids_out) memcpy(grids_out, h->grids.data(), (size_t)n * sizeof(mipgen_grid));
    return MIPGEN_OK;
}

int64_t mipgen_accel_batch_candidates(const mipgen_accel* h) { return h ? h->n_cand : 0; }
int32_t mipgen_accel_window_count(const mipgen_accel* h) { return h ? (int32_t)h->windows.size() : 0; }

int mipgen_accel_window_info(const mipgen_accel* h, int32_t w, int32_t* first_region, int32_t* n_regions, int64_t* first_candidate,
                             int64_t* n_candidates, int64_t* first_position, int64_t* n_positions)
{
    if (!h || w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range", w);
    const Window& W = h->windows[(size_t)w];
    if (first_region) *first_region = W.r0;
    if (n_regions) *n_regions = W.r1 - W.r0;
    if (first_candidate) *first_candidate = W.cand0;
    if (n_candidates) *n_candidates = W.n_cand;
    if (first_position) *first_position = W.pos0;
    if (n_positions) *n_positions = W.n_pos;
    return MIPGEN_OK;
}

// Support-vector split of a dense SVR launch: a batch with few tiles is cut along the SV list so that it still fills the chip.
// Cost model in units of one SV-group iteration of a tile: a work unit pays the tile set-up (PH0) plus its share of the groups;
// units are dispatched to n_cu compute units in rounds.
static int pick_sv_split(int n_tiles, int n_sv, int n_cu)
{
    const int groups = (n_sv + SVR_GROUP - 1) / SVR_GROUP;
    const double PH0 = 6.0;
    int best = 1;
    double best_cost = 0;
    for (int s = 1; s <= 8; s++) {
        if (s > 1 && groups / s < 32) break;
        const int64_t units = (int64_t)n_tiles * s, rounds = (units + n_cu - 1) / n_cu;
        const double cost = (double)rounds * (PH0 + (double)((groups + s - 1) / s));
        if (s == 1 || cost < best_cost * 0.98) { best = s; best_cost = cost; }
    }
    return best;
}

static int ensure_events(mipgen_accel* h)
{
    const size_t want = 4 * h->windows.size();
    while (h->ev.size() < want) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    return 0;
}

// ---- tile lists of the scoring kernels: laid out when a method is first used on the resident batch (a 200,000-region logistic design
// never prices an SVR tile) ----
static int build_record_tiles(mipgen_accel* h)          // k_records / k_records_logistic: 8 positions per tile
{
    if (h->record_tiles_ready) return MIPGEN_OK;
    const DevParams& D = h->hp;
    const int Lmax = std::max(D.e_max, D.l_max);
    std::vector<LogTile> lt;
    int span_max = 0;
    for (Window& w : h->windows) {
        w.log_tile0 = (int)lt.size();
        for (int i = w.r0; i < w.r1; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            const int Cmax = D.max_capture - d.k0 * D.inc;
            const int NPL = 8;  // 8 positions per records tile: ~10 resident blocks per CU hide the per-candidate gathers (32: 2.6 waves/SIMD, 45 % slower)
            for (int p0 = 0; p0 < d.n_pos; p0 += NPL) {
                LogTile t = {i, p0, std::min(NPL, d.n_pos - p0), 0};
                lt.push_back(t);
                span_max = std::max(span_max, t.np + Cmax + Lmax);
            }
        }
        w.n_log_tiles = (int)lt.size() - w.log_tile0;
    }
    if (h->log_tiles.reserve(std::max<size_t>(lt.size(), 1))) return MIPGEN_E_NOMEM;
    if (!lt.empty()) HIP_TRY(hipMemcpy(h->log_tiles.p, lt.data(), lt.size() * sizeof(LogTile), hipMemcpyHostToDevice));
    h->log_span_max = span_max;
    h->record_tiles_ready = true;
    return MIPGEN_OK;
}

static int build_logistic_tiles(mipgen_accel* h)        // k_logistic_dense: a run of positions with all their capture sizes
{
    if (h->logistic_tiles_ready) return MIPGEN_OK;
    const DevParams& D = h->hp;
    const int Lmax = std::max(D.e_max, D.l_max);
    std::vector<SvrTile> ldt;
    size_t ld_lds = 0;
    bool ld_ok = true;
    const int64_t ld_np_cap = h->total_pos * 2 < 8192 ? 8 : (h->total_pos * 2 < 65536 ? 16 : 64);      // small batches: more, smaller tiles (fill 256 CUs)
    for (Window& w : h->windows) {
        w.ld_tile0 = (int)ldt.size();
        for (int i = w.r0; i < w.r1 && ld_ok; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            const int Cmax = D.max_capture - d.k0 * D.inc;
            // a tile = a run of positions with ALL their capture sizes (the kernel works them off in nkc runs of <= 9, staging the bases and
            // their prefix words once): sized for the widest run's tables and the first run's reach
            const int nkc = (d.n_sizes + 8) / 9;
            int kc_max = 0;
            for (int c = 0; c < nkc; c++) kc_max = std::max(kc_max, (int)((int64_t)d.n_sizes * (c + 1) / nkc) - (int)((int64_t)d.n_sizes * c / nkc));
            const int ssr = (kc_max - 1) * D.inc + D.max_sum - D.min_sum + 1, ssmax = Cmax - D.min_sum;
            const int ssmin_all = Cmax - (d.n_sizes - 1) * D.inc - D.max_sum;
            // the positions of a tile are worked off in `subs` sub-runs of np, the downstream-arm table sliding along (only np of its
            // np + ssr - 1 window starts are new per sub-run) and the bases / prefix words staged once; small batches keep one sub-run per tile
            // (they need every workgroup they can get)
            const int subs = h->ld_subruns > 0 ? h->ld_subruns : (h->total_pos >= 1000000 ? 3 : (h->total_pos >= 25000 ? 2 : 1));   // measured: 24 x 5 kb best at 2, 8,192 exons at 3
            int np = (int)std::min<int64_t>({ld_np_cap, (int64_t)d.n_pos, 64});
            size_t b = 0;
            for (; np >= 1; np--) {
                const int np_all = std::min(subs * np, d.n_pos);
                b = std::max(mipgen_logistic_dense_lds_bytes(np_all, np, ssr, ssmax, Lmax, D.e_max - D.e_min + 1, D.l_max - D.l_min + 1),
                             mipgen_logistic_dense_lds_bytes(np_all, np, ssr, ssmax, Lmax, D.l_max - D.l_min + 1, D.e_max - D.e_min + 1));
                if (b <= 80 * 1024) break;                             // two 512-thread workgroups per compute unit: one builds tables while the other scores
            }
            if (np < 1 || ssmin_all < 1) { ld_ok = false; break; }
            ld_lds = std::max(ld_lds, b);
            // the sub-run length travels in the tile's strand field (these tiles hold both strands)
            for (int p0 = 0; p0 < d.n_pos; p0 += subs * np) { SvrTile t = {i, np, p0, std::min(subs * np, d.n_pos - p0), 0, d.n_sizes, 0, 0}; ldt.push_back(t); }
        }
        w.n_ld_tiles = (int)ldt.size() - w.ld_tile0;
    }
    if (!ld_ok) {                                        // some region does not fit that kernel's LDS: the per-candidate kernel scores the batch
        for (Window& w : h->windows) { w.ld_tile0 = 0; w.n_ld_tiles = 0; }
        h->ld_lds = 0;
        h->logistic_tiles_ready = true;
        return build_record_tiles(h);
    }
    if (h->ld_tiles.reserve(std::max<size_t>(ldt.size(), 1))) return MIPGEN_E_NOMEM;
    if (!ldt.empty()) HIP_TRY(hipMemcpy(h->ld_tiles.p, ldt.data(), ldt.size() * sizeof(SvrTile), hipMemcpyHostToDevice));
    h->ld_lds = ld_lds;
    h->logistic_tiles_ready = true;
    return MIPGEN_OK;
}

static int build_svr_tiles(mipgen_accel* h)             // k_svr_dense (+ the record tiles of k_records, which runs before it)
{
    if (h->svr_tiles_ready) return MIPGEN_OK;
    if (int rc = build_record_tiles(h)) return rc;
    const DevParams& D = h->hp;
    const int Lmax = std::max(D.e_max, D.l_max);
    const int n_arm = std::max(h->geom.n_e, h->geom.n_l) | 1;
    std::vector<SvrTile> st, st_lvl;
    std::vector<double> st_cost;
    size_t svr_lds = 0;
    int max_levels = 1;                                  // capture-size runs of the region with the most of them
    // SVR tiles: capture sizes in nearly equal runs of <= 9, positions in runs as long as the tile's LDS allows (more positions per tile =
    // fewer factor-table entries per candidate and fewer idle candidate lanes).  Every split of the sizes is priced with the kernel's
    // instruction budget - ~47 VALU per (table entry, SV) against ~2.7 per (candidate, SV) at full lanes - and the cheapest one is laid
    // out.  The choice depends on (first size, number of sizes, positions) only: exome-shaped batches repeat the same few thousand shapes.
    struct Run { int ki0, kc, np; };
    struct Shape { std::vector<Run> runs; size_t lds = 0; };
    std::unordered_map<uint64_t, Shape> shapes;
    auto shape_of = [&](const DevRegion& d, bool few) -> const Shape& {
        const int lanes = 64 * (few ? h->geom_few.wpc : h->geom.wpc);
        const uint64_t key = ((uint64_t)(uint32_t)d.n_pos << 32) | ((uint64_t)(uint16_t)d.k0 << 16) | (uint64_t)(uint16_t)d.n_sizes | (few ? (uint64_t)1 << 15 : 0);
        auto it = shapes.find(key);
        if (it != shapes.end()) return it->second;
        Shape best;
        double best_cost = 0;
        const int Cmax = D.max_capture - d.k0 * D.inc;
        for (int kc_cap = std::min(9, d.n_sizes); kc_cap >= 1; kc_cap--) {
            const int nkc = (d.n_sizes + kc_cap - 1) / kc_cap;
            if (kc_cap < std::min(9, d.n_sizes) && nkc == (d.n_sizes + kc_cap) / (kc_cap + 1)) continue;   // same split as the previous cap
            std::vector<Run> runs;
            double ent = 0, slots = 0;
            size_t lds_r = 0;
            bool ok = true;
            for (int c = 0; c < nkc && ok; c++) {
                const int ki0 = (int)((int64_t)d.n_sizes * c / nkc), ki1 = (int)((int64_t)d.n_sizes * (c + 1) / nkc);
                const int kc = ki1 - ki0;
                const int Cmax_t = Cmax - ki0 * D.inc, Cmin_t = Cmax_t - (kc - 1) * D.inc;
                const int ssmax = Cmax_t - D.min_sum, ssmin = Cmin_t - D.max_sum, ssr = ssmax - ssmin + 1;
                int np = std::max(1, std::min(lanes / kc, d.n_pos));       // every lane of a pair chunk owns one (position, capture size)
                size_t lds_t = 0;
                for (; np >= 1; np--) {
                    lds_t = mipgen_svr_lds_bytes_tile(np, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64);
                    if (lds_t <= 160 * 1024) break;
                }
                if (np < 1) { ok = false; break; }
                // np = -inc (mod 32) makes the candidate steps' downstream-factor loads conflict free (see the kernel's lane mapping):
                // taken when it costs few positions
                { const int np_cf = np - ((np + D.inc) % 32); if (np_cf >= 1 && np_cf * 10 >= np * 9) { np = np_cf; lds_t = mipgen_svr_lds_bytes_tile(np, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64); } }
                // the same number of tiles, evenly filled (the last tile of a region is not a stub that costs a full table stage)
                { const int nt = (d.n_pos + np - 1) / np, np_even = (d.n_pos + nt - 1) / nt;
                  if (np_even < np) { np = np_even; lds_t = mipgen_svr_lds_bytes_tile(np, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64); } }
                runs.push_back({ki0, kc, np});
                lds_r = std::max(lds_r, lds_t);
                const double tiles = std::ceil((double)d.n_pos / np);
                ent += tiles * (np * (h->geom.n_e + h->geom.n_l) / 2.0 + (np + ssr - 1) * (h->geom.n_e + h->geom.n_l) / 2.0 + (double)np * ssr);
                slots += tiles * lanes;
            }
            if (!ok) continue;
            const double cost = 47.0 * ent + 2.7 * slots * D.n_pairs / 1.0;
            if (best.runs.empty() || cost < best_cost) { best.runs = runs; best_cost = cost; best.lds = lds_r; }
        }
        return shapes.emplace(key, std::move(best)).first->second;
    };
    for (Window& w : h->windows) {
        w.svr_tile0 = (int)st.size();
        // the few-sizes geometry is a launch of its own: taken when the window holds enough such regions to fill the chip a few times over
        // (a handful of them stay with the main launch: an extra launch ends with the tail of its last tile)
        int n_few_regions = 0;
        for (int i = w.r0; i < w.r1; i++) if (h->hregions[i].n_pos > 0 && h->hregions[i].n_sizes == 1) n_few_regions++;
        const bool win_few = h->have_few && n_few_regions >= h->n_cu;
        std::vector<uint8_t> st_few;                           // per tile of this window: few-sizes geometry?
        for (int i = w.r0; i < w.r1; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            // Regions of ONE capture size only, and with the shape the main launch would give them (their positions per tile are bound by the LDS -
            // 70-90, inside either geometry's lanes): a score's last bits depend on where its tile starts - the window sums are differences of
            // tile-relative prefix sums -, and a region must score bit-identically in any window or shard, whichever launch takes its tiles.
            const bool few = win_few && d.n_sizes == 1;
            const Shape& shape = shape_of(d, false);
            if (shape.runs.empty()) { svr_lds = (size_t)1 << 30; continue; }
            svr_lds = std::max(svr_lds, shape.lds);
            const int Cmax = D.max_capture - d.k0 * D.inc;
            for (size_t lvl = 0; lvl < shape.runs.size(); lvl++) {
                const Run& r = shape.runs[lvl];
                max_levels = std::max(max_levels, (int)lvl + 1);
                for (int p0 = 0; p0 < d.n_pos; p0 += r.np) {
                    const int npt = std::min(r.np, d.n_pos - p0);
                    // run time of the tile in wavefront-cycles per SV group (measured shares of the three stages): table entries, scan span,
                    // candidate steps (all lanes of the block step, whatever the tile holds)
                    const int Cmax_t = Cmax - r.ki0 * D.inc, ssmax = Cmax_t - D.min_sum, ssr = (r.kc - 1) * D.inc + D.max_sum - D.min_sum + 1;
                    const double ent = (npt + (npt + ssr - 1)) * (h->geom.n_e + h->geom.n_l) / 2.0 + (double)npt * ssr;
                    const double cost = 19.0 * ent + 100.0 * (npt + ssmax + 2 * Lmax) + 75000.0;
                    for (int s2 = 0; s2 < 2; s2++) { SvrTile t = {i, s2, p0, npt, r.ki0, r.kc, (int)lvl, 0}; st.push_back(t); st_cost.push_back(cost); st_few.push_back(few ? 1 : 0); }
                }
            }
        }
        // longest tiles first: workgroups are dispatched in index order as compute units free up, so the short tiles fill the end of the
        // launch (k_svr_dense takes tile blockIdx / n_split: consecutive tiles already land on different XCDs)
        {
            const size_t t0 = (size_t)w.svr_tile0, n = st.size() - t0;
            std::vector<size_t> order(n);
            for (size_t k = 0; k < n; k++) order[k] = k;
            // (the tiles of the few-sizes geometry behind the others: the window's second launch)
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return st_few[a] != st_few[b] ? st_few[a] < st_few[b] : st_cost[t0 + a] > st_cost[t0 + b]; });
            std::vector<SvrTile> sorted(n);
            for (size_t k = 0; k < n; k++) sorted[k] = st[t0 + order[k]];
            std::copy(sorted.begin(), sorted.end(), st.begin() + (ptrdiff_t)t0);
            w.n_svr_few = 0;
            for (size_t k = 0; k < n; k++) w.n_svr_few += st_few[k];
        }
        w.n_svr_tiles = (int)st.size() - w.svr_tile0;
        // the same tiles grouped by capture-size run (run 0 first, longest first inside a run): the launch order of the dynamic skip (kernels_skip.hip)
        {
            const size_t t0 = (size_t)w.svr_tile0, n = (size_t)w.n_svr_tiles;
            int wl = 0;
            for (size_t k = 0; k < n; k++) wl = std::max(wl, st[t0 + k].level + 1);
            w.lvl_tile0.assign(1, (int)st_lvl.size());
            for (int lvl = 0; lvl < wl; lvl++) {
                for (size_t k = 0; k < n; k++) if (st[t0 + k].level == lvl) st_lvl.push_back(st[t0 + k]);      // (order kept: run 0 ends with the few-sizes tiles)
                w.lvl_tile0.push_back((int)st_lvl.size());
            }
            w.lvl0_few = w.n_svr_few;                          // regions of one size have one run
        }
    }
    h->svr_batch_error.clear();
    if (svr_lds > 160 * 1024) {
        h->svr_batch_error = "an SVR tile needs more than 160 KiB of LDS: capture range / arm lists too wide";
        st.clear();
        for (Window& w : h->windows) { w.svr_tile0 = 0; w.n_svr_tiles = 0; }
        svr_lds = 0;
    }
    if (h->svr_tiles.reserve(std::max<size_t>(st.size(), 1))) return MIPGEN_E_NOMEM;
    if (!st.empty()) HIP_TRY(hipMemcpy(h->svr_tiles.p, st.data(), st.size() * sizeof(SvrTile), hipMemcpyHostToDevice));
    // the run-ordered copy of the tile list and the runs of every region, for the dynamic skip between capture-size runs (only regions of more
    // than nine capture sizes have a second run)
    h->svr_levels = st.empty() ? 1 : max_levels;
    if (h->svr_levels > 1) {
        std::vector<uint32_t> rb((size_t)h->n_regions * (size_t)max_levels, 0u);
        for (const SvrTile& t : st) rb[(size_t)t.region * (size_t)max_levels + (size_t)t.level] = (uint32_t)t.ki0 | ((uint32_t)t.kc << 16);
        if (h->svr_tiles_lvl.reserve(st_lvl.size()) || h->run_bounds.reserve(rb.size())) return MIPGEN_E_NOMEM;
        HIP_TRY(hipMemcpy(h->svr_tiles_lvl.p, st_lvl.data(), st_lvl.size() * sizeof(SvrTile), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->run_bounds.p, rb.data(), rb.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    h->svr_lds = svr_lds;
    h->svr_tiles_ready = true;
    return MIPGEN_OK;
}

static int ensure_tiles(mipgen_accel* h, int32_t method)
{
    // a parameter set the tiled SVR kernel cannot take still needs the record tiles: its dense grid goes through the list scorer (svr_window_via_list)
    if (method == MIPGEN_SCORE_SVR) return h->svr_geometry_error.empty() ? build_svr_tiles(h) : build_record_tiles(h);
    return build_logistic_tiles(h);
}

// After a stream synchronisation: did a print-exact re-score list overflow since the last check?  The surplus entries keep the dense kernel's
// value (still within 1e-5 - but their printed sixth digit is no longer guaranteed), so the download that would hand them out fails loudly.
static int pb_check(mipgen_accel* h)
{
    if (!h->pb_over || *h->pb_over == 0u) return MIPGEN_OK;
    const unsigned int v = *h->pb_over;
    *h->pb_over = 0u;
    return fail(MIPGEN_E_STATE, "print-exact re-score: %u scores on a 6-digit rounding boundary did not fit the re-score list "
                                "(mipgen_accel_set_print_exact(h, 0) waives the guarantee)", v);
}

// SVR scores within the dense / list kernels' error of a midpoint between two 6-significant-digit numbers (what the front end prints,
// mipgen.cpp:774) are re-scored by k_candidates in the reference's own operation order and overwritten, so that the printed digit is the
// reference's (its error against the reference's double is ~1e-16 relative: the libm exponential).  No host round trip: the list length
// stays on the device, the re-scoring grid is the list's capacity.  list = nullptr: the dense results of regions [r0, r1).
static int fix_print_boundaries(mipgen_accel* h, int r0, int r1, const mipgen_candidate* list, double* scores, const uint64_t* records, int64_t n)
{
    if (!h->print_exact || n <= 0 || h->n_sv <= 0) return MIPGEN_OK;
    const unsigned int cap = (unsigned int)std::min<int64_t>(n / 1024 + 4096, (int64_t)1 << 24);
    if (h->pb_cands.reserve(cap) || h->pb_idx.reserve(cap) || h->pb_scores.reserve(cap) || h->pb_count.reserve(1)) return MIPGEN_E_NOMEM;
    const double tol_rel = 1e-10, tol_abs = 1e-13 * std::max(1.0, h->sum_abs_coef);
    HIP_TRY(hipMemsetAsync(h->pb_count.p, 0, sizeof(unsigned int), h->stream));
    if (list) HIP_TRY(mipgen_launch_print_boundary_scan_list(h->stream, list, scores, records, (int)n, tol_rel, tol_abs, h->pb_cands.p, h->pb_idx.p, h->pb_count.p, cap));
    else HIP_TRY(mipgen_launch_print_boundary_scan(h->stream, h->dp, h->regions.p, r0, r1, scores, records, n, tol_rel, tol_abs, h->pb_cands.p, h->pb_idx.p, h->pb_count.p, cap, h->n_cu));
    HIP_TRY(mipgen_launch_candidates(h->stream, (int)cap, h->dp, h->regions.p, h->pb_cands.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->model.p, h->n_sv,
                                     h->gamma, h->rho, MIPGEN_SCORE_SVR, h->pb_scores.p, nullptr, nullptr, nullptr, 1, h->pb_count.p));
    HIP_TRY(mipgen_launch_scatter_scores(h->stream, h->pb_scores.p, h->pb_idx.p, (int)cap, h->pb_count.p, scores, h->pb_over));
    return MIPGEN_OK;
}

// the silent path (mipgen_accel_score_condense_all): only the condensed survivors of the window are ever printed, so only they are tested
// and re-scored (2 per scan position instead of the whole dense grid); the value goes into the survivor's score field
static int fix_print_boundaries_survivors(mipgen_accel* h, int w)
{
    const Window& W = h->windows[(size_t)w];
    const int64_t n = 2 * W.n_pos;
    if (!h->print_exact || n <= 0 || h->n_sv <= 0) return MIPGEN_OK;
    const unsigned int cap = (unsigned int)std::min<int64_t>(n / 256 + 1024, (int64_t)1 << 22);
    if (h->pb_cands.reserve(cap) || h->pb_idx.reserve(cap) || h->pb_scores.reserve(cap) || h->pb_count.reserve(1)) return MIPGEN_E_NOMEM;
    const double tol_rel = 1e-10, tol_abs = 1e-13 * std::max(1.0, h->sum_abs_coef);
    mipgen_survivor* surv = h->survivors.p + 2 * W.pos0;
    HIP_TRY(hipMemsetAsync(h->pb_count.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(mipgen_launch_print_boundary_scan_surv(h->stream, h->dp, h->regions.p, W.r0, W.r1, surv, n, W.cand0, tol_rel, tol_abs, h->pb_cands.p, h->pb_idx.p, h->pb_count.p, cap));
    HIP_TRY(mipgen_launch_candidates(h->stream, (int)cap, h->dp, h->regions.p, h->pb_cands.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->model.p, h->n_sv,
                                     h->gamma, h->rho, MIPGEN_SCORE_SVR, h->pb_scores.p, nullptr, nullptr, nullptr, 1, h->pb_count.p));
    HIP_TRY(mipgen_launch_scatter_surv_scores(h->stream, h->pb_scores.p, h->pb_idx.p, (int)cap, h->pb_count.p, surv, h->pb_over));
    return MIPGEN_OK;
}

// The dense grid of a window through the LIST scorer: the route of SVR parameter sets outside the tiled kernel's limits (scan size < 3,
// > 240 arm pairs, a tile beyond 160 KiB of LDS; h->svr_geometry_error / svr_batch_error say which) - the reference completes every
// -arm_lengths / -capture_increment / range it is given (mipgen.cpp:222-261, 427-444), so does this library, at the list scorer's rate
// (k_features_batch builds all 192 features per candidate, k_svr_gemm contracts them with the model on the FP64 matrix cores).
static int svr_window_via_list(mipgen_accel* h, const Window& W)
{
    // the list kernels stage an insert of at most 1,024 bases (kernels_misc.hip: MAX_INSERT): beyond that they would hand out the values of
    // bounds-skipped candidates - fail loudly instead
    if (h->hp.max_capture - h->hp.min_sum > 1024)
        return fail(MIPGEN_E_INVALID, "dense SVR scoring of scan sizes above 1024 bases is not supported (capture size %d, smallest arm sum %d)", h->hp.max_capture, h->hp.min_sum);
    const int64_t CH = (int64_t)1 << 19;                                  // candidates per chunk: 0.8 GB of features
    const size_t n_max = (size_t)std::min<int64_t>(CH, std::max<int64_t>(W.n_cand, 1));
    if (h->cand_in.reserve(n_max) || h->cand_records.reserve(n_max) || h->cand_feats.reserve(n_max * MIPGEN_N_FEATURES)) return MIPGEN_E_NOMEM;
    for (int64_t c0 = 0; c0 < W.n_cand; c0 += CH) {
        const int n = (int)std::min<int64_t>(CH, W.n_cand - c0);
        HIP_TRY(mipgen_launch_dense_candidates(h->stream, h->dp, h->regions.p, W.r0, W.r1, c0, n, h->cand_in.p));
        HIP_TRY(mipgen_launch_features_batch(h->stream, n, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->cand_records.p,
                                             h->cand_feats.p));
        HIP_TRY(mipgen_launch_svr_gemm(h->stream, n, h->cand_feats.p, h->cand_records.p, h->model_t.p, h->sv_norm.p, h->sv_coef.p, h->sv_center.p, h->n_sv_pad,
                                       h->gamma, h->rho, h->scores.p + c0));
        HIP_TRY(mipgen_launch_dense_list_fix(h->stream, n, h->records.p + c0, h->rho, h->s_guard, h->scores.p + c0));
    }
    return MIPGEN_OK;
}

static int score_window_impl(mipgen_accel* h, int w, int32_t method, bool fix_dense = true)
{
    if (int rc = ensure_tiles(h, method)) return rc;
    const Window& W = h->windows[(size_t)w];
    const bool svr_via_list = method == MIPGEN_SCORE_SVR && (!h->svr_geometry_error.empty() || !h->svr_batch_error.empty());
    if (svr_via_list && h->model_t.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    hipEvent_t* ev = nullptr;
    if (h->timing) { if (ensure_events(h)) return MIPGEN_E_HIP; ev = &h->ev[4 * (size_t)w]; }
    if (ev) HIP_TRY(hipEventRecord(ev[0], h->stream));
    if (method == MIPGEN_SCORE_LOGISTIC && h->ld_lds > 0)
        // records + logistic scores from per-window tables (kernels_logistic_dense.hip)
        HIP_TRY(mipgen_launch_logistic_dense(h->stream, W.n_ld_tiles, h->ld_lds, h->dp, h->regions.p, h->ld_tiles.p + W.ld_tile0, h->bases.p, h->copy.p,
                                             h->unmap.p, h->dconsts, h->scores.p, h->records.p));
    else
        HIP_TRY(mipgen_launch_records_logistic(h->stream, method == MIPGEN_SCORE_LOGISTIC, W.n_log_tiles, h->log_span_max, h->dp, h->regions.p,
                                               h->log_tiles.p + W.log_tile0, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->scores.p, h->records.p));
    if (ev) HIP_TRY(hipEventRecord(ev[1], h->stream));
    if (svr_via_list) {
        if (int rc = svr_window_via_list(h, W)) return rc;
        if (fix_dense) { if (int rc = fix_print_boundaries(h, W.r0, W.r1, nullptr, h->scores.p, h->records.p, W.n_cand)) return rc; }
    } else if (method == MIPGEN_SCORE_SVR) {
        const double gamma_l2e = h->gamma * 1.4426950408889634074;
        int split = h->sv_split > 0 ? h->sv_split : pick_sv_split(W.n_svr_tiles, h->n_sv, h->n_cu);
        split = std::max(1, std::min(split, (h->n_sv + SVR_GROUP - 1) / SVR_GROUP));
        if (split > 1 && h->partials.reserve((size_t)(split - 1) * (size_t)std::max<int64_t>(W.n_cand, 1))) return MIPGEN_E_NOMEM;
        const int levels = (int)W.lvl_tile0.size() - 1;
        if (h->dyn_skip && h->svr_levels > 1 && levels > 1) {
            // The runs of capture sizes in order, largest first; between two runs the enumeration state of every scan position (mipgen.cpp:426-497)
            // is advanced over the run just scored, and the tiles of the next run whose positions have all stopped (:430) are left out
            // (kernels_skip.hip).  One launch per run (no split along the SV list: these are large batches), one small synchronisation per run
            // for the number of tiles that are left.
            int max_run = 0;
            for (int l = 0; l < levels; l++) max_run = std::max(max_run, W.lvl_tile0[(size_t)l + 1] - W.lvl_tile0[(size_t)l]);
            if (h->run_pbs.reserve((size_t)W.n_pos) || h->run_state.reserve((size_t)W.n_pos) || h->run_keep.reserve((size_t)max_run + 1) ||
                h->run_offs.reserve((size_t)max_run + 1) || h->svr_tiles_kept.reserve((size_t)max_run) || h->skip_count.reserve(1))
                return MIPGEN_E_NOMEM;
            HIP_TRY(hipMemsetAsync(h->run_pbs.p, 0, (size_t)W.n_pos * sizeof(double), h->stream));
            HIP_TRY(hipMemsetAsync(h->run_state.p, 0, (size_t)W.n_pos, h->stream));
            if (!h->skip_count_valid) { HIP_TRY(hipMemsetAsync(h->skip_count.p, 0, sizeof(unsigned long long), h->stream)); h->skip_count_valid = true; }
            // a score this close to the limit could land on the other side after the print-exact re-score: such a position is never marked as stopped
            const double margin = 1e-6 * std::max(1.0, fabs(h->hp.upper)) + 1e-12 * std::max(1.0, h->sum_abs_coef);
            for (int l = 0; l < levels; l++) {
                const SvrTile* tl = h->svr_tiles_lvl.p + W.lvl_tile0[(size_t)l];
                int nt = W.lvl_tile0[(size_t)l + 1] - W.lvl_tile0[(size_t)l];
                if (l > 0 && nt > 0) {
                    HIP_TRY(mipgen_launch_svr_run_state(h->stream, W.n_pos, h->dp, h->regions.p, h->pos_region.p + W.pos0, h->pos_local.p + W.pos0, h->run_bounds.p,
                                                        h->svr_levels, l - 1, margin, h->scores.p, h->records.p, h->run_pbs.p, h->run_state.p));
                    HIP_TRY(mipgen_launch_svr_tile_keep(h->stream, nt, tl, h->region_pos0.p, W.pos0, h->run_state.p, h->run_keep.p));
                    size_t temp_bytes = 0;
                    HIP_TRY(mipgen_scan_i64(h->stream, nullptr, &temp_bytes, h->run_keep.p, h->run_offs.p, (int64_t)nt + 1));
                    if (h->fmt_temp.reserve(temp_bytes + 16)) return MIPGEN_E_NOMEM;
                    HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->run_keep.p, h->run_offs.p, (int64_t)nt + 1));
                    HIP_TRY(mipgen_launch_svr_tile_compact(h->stream, nt, tl, h->run_keep.p, h->run_offs.p, h->svr_tiles_kept.p, h->dp, h->regions.p, h->scores.p,
                                                           h->skip_count.p));
                    int64_t kept = 0;
                    HIP_TRY(hipMemcpyAsync(&kept, h->run_offs.p + nt, sizeof kept, hipMemcpyDeviceToHost, h->stream));
                    HIP_TRY(hipStreamSynchronize(h->stream));
                    tl = h->svr_tiles_kept.p; nt = (int)kept;
                }
                const int nt_few = l == 0 ? W.lvl0_few : 0;
                HIP_TRY(mipgen_launch_svr_dense(h->stream, nt - nt_few, nt_few, h->svr_lds, h->dp, &h->geom, &h->geom_few, h->regions.p, tl, h->bases.p, h->copy.p,
                                                (const double*)h->dconsts, h->model.p, h->n_sv, gamma_l2e, h->rho, h->s_guard, h->records.p, h->scores.p, W.n_cand, 1,
                                                nullptr));
            }
        } else
        HIP_TRY(mipgen_launch_svr_dense(h->stream, W.n_svr_tiles - W.n_svr_few, W.n_svr_few, h->svr_lds, h->dp, &h->geom, &h->geom_few, h->regions.p,
                                        h->svr_tiles.p + W.svr_tile0, h->bases.p, h->copy.p, (const double*)h->dconsts /* log10_tab is the first member */,
                                        h->model.p, h->n_sv, gamma_l2e, h->rho, h->s_guard, h->records.p, h->scores.p, W.n_cand, split, h->partials.p));
        if (fix_dense) { if (int rc = fix_print_boundaries(h, W.r0, W.r1, nullptr, h->scores.p, h->records.p, W.n_cand)) return rc; }
    }
    if (ev) { HIP_TRY(hipEventRecord(ev[2], h->stream)); h->ev_used[(size_t)w] |= 1; }
    h->cur_window = w; h->scored = true; h->replayed = false;
    if ((size_t)w < h->win_state.size()) h->win_state[(size_t)w] = 0;          // its survivors / collapse results are of older scores now
    return MIPGEN_OK;
}

static int replay_window_impl(mipgen_accel* h, bool want_mask)
{
    const int w = h->cur_window;
    const Window& W = h->windows[(size_t)w];
    // asynchronous on the handle's stream; buffers and the position map were laid out at upload
    if (W.r1 > W.r0) HIP_TRY(hipMemsetAsync(h->emitted_per_region.p + W.r0, 0, (size_t)(W.r1 - W.r0) * sizeof(unsigned long long), h->stream));
    // the per-candidate emitted flags are only materialised for callers that fetch them (the all_mips file); the fold itself keeps
    // them as lane masks in LDS
    if (want_mask && W.n_cand) HIP_TRY(hipMemsetAsync(h->emitted.p, 0, (size_t)W.n_cand, h->stream));
    HIP_TRY(mipgen_launch_replay_condense(h->stream, h->n_regions, (int)W.n_pos, h->dp, h->hp.n_pairs, h->hp.n_sizes_all, h->regions.p, h->pos_region.p + W.pos0,
                                          h->pos_local.p + W.pos0, h->scores.p, h->records.p, h->copy.p, W.cand0, want_mask ? h->emitted.p : nullptr,
                                          h->survivors.p + 2 * W.pos0, h->emitted_per_region.p));
    h->mask_valid = want_mask;
    if (h->timing && h->ev.size() >= 4 * ((size_t)w + 1)) { HIP_TRY(hipEventRecord(h->ev[4 * (size_t)w + 3], h->stream)); h->ev_used[(size_t)w] |= 2; }
    h->replayed = true;
    if ((size_t)w < h->win_state.size()) h->win_state[(size_t)w] = 1;
    return MIPGEN_OK;
}

static int collapse_window_impl(mipgen_accel* h)
{
    const Window& W = h->windows[(size_t)h->cur_window];
    HIP_TRY(mipgen_launch_collapse(h->stream, W.n_col_tiles, h->col_tiles.p + W.col_tile0, h->dp, h->regions.p, h->region_pos0.p, h->region_base0.p,
                                   h->survivors.p, h->copy.p, W.cand0, h->collapsed.p, h->hp.max_capture - h->hp.min_sum));
    if ((size_t)h->cur_window < h->win_state.size()) h->win_state[(size_t)h->cur_window] |= 2;
    return MIPGEN_OK;
}

static int check_scoring_args(mipgen_accel* h, int32_t method)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (method != MIPGEN_SCORE_LOGISTIC && method != MIPGEN_SCORE_SVR) return fail(MIPGEN_E_INVALID, "method must be logistic or svr");
    if (h->windows.empty()) return fail(MIPGEN_E_STATE, "no resident region batch");
    if (method == MIPGEN_SCORE_SVR && h->n_sv <= 0 && h->model.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    return MIPGEN_OK;
}

int mipgen_accel_score_window(mipgen_accel* h, int32_t w, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    if (w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range (%zu windows)", w, h->windows.size());
    HIP_TRY(hipSetDevice(h->device));
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    return score_window_impl(h, w, method);
}

int mipgen_accel_score_resident(mipgen_accel* h, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    if (h->windows.size() != 1)
        return fail(MIPGEN_E_STATE, "the resident batch spans %zu result windows: use mipgen_accel_score_window / mipgen_accel_score_condense_all", h->windows.size());
    return mipgen_accel_score_window(h, 0, method);
}

int mipgen_accel_score_condense_all(mipgen_accel* h, int32_t method)
{
    if (int rc = check_scoring_args(h, method)) return rc;
    HIP_TRY(hipSetDevice(h->device));
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    for (int w = 0; w < (int)h->windows.size(); w++) {
        if (int rc = score_window_impl(h, w, method, false)) return rc;
        if (int rc = replay_window_impl(h, false)) return rc;
        if (method == MIPGEN_SCORE_SVR) { if (int rc = fix_print_boundaries_survivors(h, w)) return rc; }
        if (int rc = collapse_window_impl(h)) return rc;
    }
    return MIPGEN_OK;
}

int mipgen_accel_set_dynamic_skip(mipgen_accel* h, int32_t on)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->dyn_skip = on != 0;
    return MIPGEN_OK;
}

int mipgen_accel_skip_state(mipgen_accel* h, uint8_t* state, double* previous_best, int64_t capacity)
{
    if (!h || capacity < 0) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (h->cur_window < 0 || !h->run_state.p) return fail(MIPGEN_E_STATE, "no window scored with the dynamic skip");
    const Window& W = h->windows[(size_t)h->cur_window];
    if (capacity < W.n_pos) return fail(MIPGEN_E_INVALID, "capacity %lld < %lld scan positions", (long long)capacity, (long long)W.n_pos);
    HIP_TRY(hipSetDevice(h->device));
    if (state) HIP_TRY(hipMemcpyAsync(state, h->run_state.p, (size_t)W.n_pos, hipMemcpyDeviceToHost, h->stream));
    if (previous_best) HIP_TRY(hipMemcpyAsync(previous_best, h->run_pbs.p, (size_t)W.n_pos * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_skipped_candidates(mipgen_accel* h, int64_t* n)
{
    if (!h || !n) return fail(MIPGEN_E_INVALID, "bad arguments");
    *n = 0;
    if (!h->skip_count_valid) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, h->skip_count.p, sizeof v, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemsetAsync(h->skip_count.p, 0, sizeof v, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *n = (int64_t)v;
    return MIPGEN_OK;
}

int mipgen_accel_result_device_ptrs(const mipgen_accel* h, void** scores_dev, void** records_dev)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (scores_dev) *scores_dev = h->scores.p;
    if (records_dev) *records_dev = h->records.p;
    return MIPGEN_OK;
}

int mipgen_accel_download_results(mipgen_accel* h, double* scores, uint64_t* records, int64_t first, int64_t count)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->scored || h->cur_window < 0) return fail(MIPGEN_E_STATE, "nothing scored yet");
    const Window& W = h->windows[(size_t)h->cur_window];
    if (first < W.cand0 || count < 0 || first + count > W.cand0 + W.n_cand)
        return fail(MIPGEN_E_INVALID, "range [%lld,+%lld) outside the scored window [%lld,+%lld)", (long long)first, (long long)count, (long long)W.cand0, (long long)W.n_cand);
    HIP_TRY(hipSetDevice(h->device));
    if (count == 0) { HIP_TRY(hipStreamSynchronize(h->stream)); return MIPGEN_OK; }
    const int64_t off = first - W.cand0;
    if (scores) HIP_TRY(hipMemcpyAsync(scores, h->scores.p + off, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (records) HIP_TRY(hipMemcpyAsync(records, h->records.p + off, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = pb_check(h)) return rc_pb;
    return MIPGEN_OK;
}

int mipgen_accel_score_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, int32_t method, mipgen_grid* grids_out,
                               double* scores, uint64_t* records, int64_t capacity)
{
    int rc = mipgen_accel_upload_regions(h, regions, n, grids_out);
    if (rc) return rc;
    if (h->n_cand > capacity) return fail(MIPGEN_E_INVALID, "result capacity %lld < %lld candidates", (long long)capacity, (long long)h->n_cand);
    for (int w = 0; w < (int)h->windows.size(); w++) {
        const Window& W = h->windows[(size_t)w];
        rc = mipgen_accel_score_window(h, w, method);
        if (rc) return rc;
        rc = mipgen_accel_download_results(h, scores ? scores + W.cand0 : nullptr, records ? records + W.cand0 : nullptr, W.cand0, W.n_cand);
        if (rc) return rc;
    }
    return MIPGEN_OK;
}

int mipgen_accel_score_candidates(mipgen_accel* h, const mipgen_candidate* cands, int32_t n, int32_t method, double* scores,
                                  uint64_t* records, double* features, mipgen_candidate_ints* ints)
{
    if (!h || n < 0 || (n > 0 && !cands)) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (method != MIPGEN_SCORE_LOGISTIC && method != MIPGEN_SCORE_SVR) return fail(MIPGEN_E_INVALID, "method must be logistic or svr");
    if (h->regions.p == nullptr) return fail(MIPGEN_E_STATE, "no resident region batch");
    if (method == MIPGEN_SCORE_SVR && h->model.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    if (n == 0) return MIPGEN_OK;
    for (int i = 0; i < n; i++) if (cands[i].region < 0 || cands[i].region >= h->n_regions) return fail(MIPGEN_E_INVALID, "candidate %d: region %d not resident", i, cands[i].region);
    HIP_TRY(hipSetDevice(h->device));
    if (h->cand_in.reserve((size_t)n) || h->cand_scores.reserve((size_t)n) || h->cand_records.reserve((size_t)n) ||
        (features && h->cand_feats.reserve((size_t)n * MIPGEN_N_FEATURES)) || (ints && h->cand_ints.reserve((size_t)n)))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->cand_in.p, cands, (size_t)n * sizeof(mipgen_candidate), hipMemcpyHostToDevice, h->stream));
    // long SVR lists (a mixed design re-scores every condensed survivor): features + records by k_features_batch (a wavefront per candidate),
    // then all candidate x support-vector distances through the FP64 matrix cores (k_svr_gemm) instead of one model walk per candidate
    const bool batched = method == MIPGEN_SCORE_SVR && n >= 256 && scores;
    if (batched && h->cand_feats.reserve((size_t)n * MIPGEN_N_FEATURES)) return MIPGEN_E_NOMEM;
    hipEvent_t le[3] = {nullptr, nullptr, nullptr};
    const bool time_list = batched && !ints && h->timing;
    if (time_list) { for (hipEvent_t& e : le) HIP_TRY(hipEventCreate(&e)); HIP_TRY(hipEventRecord(le[0], h->stream)); }
    if (batched && !ints)
        // lists: one wavefront per candidate for the features + records, then the matrix-core scorer
        HIP_TRY(mipgen_launch_features_batch(h->stream, n, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->cand_records.p,
                                             h->cand_feats.p));
    else
        HIP_TRY(mipgen_launch_candidates(h->stream, n, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts,
                                         h->model.p, h->n_sv, h->gamma, h->rho, method, batched ? nullptr : h->cand_scores.p, h->cand_records.p,
                                         (features || batched) ? h->cand_feats.p : nullptr, ints ? h->cand_ints.p : nullptr, 0, nullptr));
    if (time_list) HIP_TRY(hipEventRecord(le[1], h->stream));
    if (batched) HIP_TRY(mipgen_launch_svr_gemm(h->stream, n, h->cand_feats.p, h->cand_records.p, h->model_t.p, h->sv_norm.p, h->sv_coef.p, h->sv_center.p, h->n_sv_pad,
                                            h->gamma, h->rho, h->cand_scores.p));
    if (time_list) HIP_TRY(hipEventRecord(le[2], h->stream));
    if (batched) { if (int rc = fix_print_boundaries(h, 0, 0, h->cand_in.p, h->cand_scores.p, h->cand_records.p, n)) return rc; }
    if (scores) HIP_TRY(hipMemcpyAsync(scores, h->cand_scores.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (records) HIP_TRY(hipMemcpyAsync(records, h->cand_records.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    if (features) HIP_TRY(hipMemcpyAsync(features, h->cand_feats.p, (size_t)n * MIPGEN_N_FEATURES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (ints) HIP_TRY(hipMemcpyAsync(ints, h->cand_ints.p, (size_t)n * sizeof(mipgen_candidate_ints), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = pb_check(h)) return rc_pb;
    if (time_list) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, le[0], le[1]) == hipSuccess && hipEventElapsedTime(&b, le[1], le[2]) == hipSuccess) { h->list_feat_ms = a; h->list_svr_ms = b; }
        for (hipEvent_t e : le) (void)hipEventDestroy(e);
    }
    return MIPGEN_OK;
}

static void lrc_mers(LrcMers& M)
{
    static const char* mers[MIPGEN_N_LRC] = MIPGEN_FEATURE_MERS;
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
    for (int i = 0; i < MIPGEN_N_LRC; i++) {
        const int k = (int)strlen(mers[i]);
        int cd = 0, rc = 0;
        for (int j = 0; j < k; j++) cd = cd * 4 + code(mers[i][j]);
        for (int j = k - 1; j >= 0; j--) rc = rc * 4 + (3 - code(mers[i][j]));
        M.k[i] = (int8_t)k; M.code[i] = (int8_t)cd; M.rc[i] = (int8_t)(rc == cd ? -1 : rc);
    }
}

int mipgen_accel_long_range_content_batch(mipgen_accel* h, int32_t n, const char* const* extended_seqs, const int32_t* lens,
                                          const int32_t* chrom_seq_starts, const int32_t* chrom_seq_stops, double* out)
{
    if (!h || n < 0 || (n > 0 && (!extended_seqs || !lens || !chrom_seq_starts || !chrom_seq_stops || !out))) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (n == 0) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    LrcMers M;
    lrc_mers(M);
    std::vector<int64_t> offs((size_t)n);
    std::vector<int32_t> denoms((size_t)n);
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        if (lens[i] < 0 || (lens[i] > 0 && !extended_seqs[i])) return fail(MIPGEN_E_INVALID, "long-range sequence %d is missing", i);
        offs[(size_t)i] = total; total += lens[i];
        denoms[(size_t)i] = chrom_seq_stops[i] - chrom_seq_starts[i] + 2001;          // Featurev5.cpp:49,53
    }
    std::vector<char> packed((size_t)std::max<int64_t>(total, 1));
    for (int i = 0; i < n; i++) if (lens[i]) memcpy(&packed[(size_t)offs[(size_t)i]], extended_seqs[i], (size_t)lens[i]);
    if (h->lrc_seq.reserve(packed.size()) || h->lrc_out.reserve((size_t)n * MIPGEN_N_LRC) || h->lrc_offs.reserve((size_t)n) ||
        h->lrc_lens.reserve((size_t)n) || h->lrc_denoms.reserve((size_t)n))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->lrc_seq.p, packed.data(), packed.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->lrc_offs.p, offs.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->lrc_lens.p, lens, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->lrc_denoms.p, denoms.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mipgen_launch_long_range(h->stream, n, h->lrc_seq.p, h->lrc_offs.p, h->lrc_lens.p, h->lrc_denoms.p, &M, h->lrc_out.p));
    HIP_TRY(hipMemcpyAsync(out, h->lrc_out.p, (size_t)n * MIPGEN_N_LRC * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_long_range_content(mipgen_accel* h, const char* extended_seq, int32_t len, int32_t chrom_seq_start,
                                    int32_t chrom_seq_stop, double* out44)
{
    if (!h || !extended_seq || len < 0 || !out44) return fail(MIPGEN_E_INVALID, "bad arguments");
    return mipgen_accel_long_range_content_batch(h, 1, &extended_seq, &len, &chrom_seq_start, &chrom_seq_stop, out44);
}

int mipgen_accel_replay_condense(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->scored || h->cur_window < 0) return fail(MIPGEN_E_STATE, "replay requested before scoring");
    HIP_TRY(hipSetDevice(h->device));
    return replay_window_impl(h, true);
}

int mipgen_accel_download_replay(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors, int64_t survivor_capacity,
                                 uint8_t* emitted_mask, int64_t mask_capacity)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->replayed || h->cur_window < 0) return fail(MIPGEN_E_STATE, "mipgen_accel_replay_condense has not run on these scores");
    const Window& W = h->windows[(size_t)h->cur_window];
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = pb_check(h)) return rc_pb;
    if (emitted_per_region && W.r1 > W.r0)
        HIP_TRY(hipMemcpy(emitted_per_region, h->emitted_per_region.p + W.r0, (size_t)(W.r1 - W.r0) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (survivors) {
        if (survivor_capacity < 2 * W.n_pos) return fail(MIPGEN_E_INVALID, "survivor capacity too small");
        if (W.n_pos) HIP_TRY(hipMemcpy(survivors, h->survivors.p + 2 * W.pos0, (size_t)(2 * W.n_pos) * sizeof(mipgen_survivor), hipMemcpyDeviceToHost));
    }
    if (emitted_mask) {
        if (!h->mask_valid) return fail(MIPGEN_E_STATE, "the emitted mask is only kept by mipgen_accel_replay_condense, not by mipgen_accel_score_condense_all");
        if (mask_capacity < W.n_cand) return fail(MIPGEN_E_INVALID, "mask capacity too small");
        if (W.n_cand) HIP_TRY(hipMemcpy(emitted_mask, h->emitted.p, (size_t)W.n_cand, hipMemcpyDeviceToHost));
    }
    return MIPGEN_OK;
}

int mipgen_accel_collapse(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->replayed || h->cur_window < 0) return fail(MIPGEN_E_STATE, "collapse requested before replay + condense");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = collapse_window_impl(h)) return rc;
    return MIPGEN_OK;
}

int mipgen_accel_region_bases(const mipgen_accel* h, int32_t region, int64_t* first_entry, int32_t* n_bases)
{
    if (!h || region < 0 || region >= h->n_regions) return fail(MIPGEN_E_INVALID, "region %d out of range", region);
    if (first_entry) *first_entry = h->h_region_base0[(size_t)region];
    if (n_bases) *n_bases = (int32_t)((h->h_region_base0[(size_t)region + 1] - h->h_region_base0[(size_t)region]) / 2);
    return MIPGEN_OK;
}

int mipgen_accel_download_collapsed(mipgen_accel* h, int32_t window, int32_t* best_scan_index, int64_t capacity)
{
    if (!h || !best_scan_index) return fail(MIPGEN_E_INVALID, "bad arguments");
    // validity is per result window: every window asked for must have been collapsed since it was last scored
    for (size_t w = 0; w < h->win_state.size(); w++)
        if ((window < 0 || (size_t)window == w) && !(h->win_state[w] & 2))
            return fail(MIPGEN_E_STATE, "mipgen_accel_collapse / mipgen_accel_score_condense_all has not run on the current scores of window %zu", w);
    if (h->win_state.empty()) return fail(MIPGEN_E_STATE, "no resident region batch");
    int64_t first = 0, count = h->h_region_base0.empty() ? 0 : h->h_region_base0.back();
    if (window >= 0) {
        if (window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range", window);
        first = h->windows[(size_t)window].base0; count = h->windows[(size_t)window].n_base_entries;
    }
    if (capacity < count) return fail(MIPGEN_E_INVALID, "collapsed capacity %lld < %lld entries", (long long)capacity, (long long)count);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (count) HIP_TRY(hipMemcpy(best_scan_index, h->collapsed.p + first, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MIPGEN_OK;
}

// ---- section 8f-4: all_mips records formatted on the device -----------------------------------------------------------------------
int mipgen_accel_format_all_mips(mipgen_accel* h, const mipgen_record_names* names, const char* middle, int64_t first_index, int64_t* n_records, int64_t* n_bytes)
{
    if (!h || !middle || !n_records || !n_bytes) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (!h->replayed || !h->mask_valid || h->cur_window < 0) return fail(MIPGEN_E_STATE, "format_all_mips needs mipgen_accel_replay_condense on the scored window first");
    const Window& W = h->windows[(size_t)h->cur_window];
    const int nr = W.r1 - W.r0;
    if (nr > 0 && !names) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (strlen(middle) >= sizeof(((FmtConst*)0)->middle)) return fail(MIPGEN_E_INVALID, "middle sequence too long");
    HIP_TRY(hipSetDevice(h->device));
    FmtConst FC;
    memset(&FC, 0, sizeof FC);
    FC.middle_len = (int32_t)strlen(middle); memcpy(FC.middle, middle, (size_t)FC.middle_len);
    FC.n_regions = nr; FC.first_index = first_index;
    std::vector<FmtRegion> fr((size_t)std::max(nr, 1));
    std::string pool;
    int64_t n_rb = 0;
    for (int i = 0; i < nr; i++) {
        const DevRegion& d = h->hregions[(size_t)(W.r0 + i)];
        FmtRegion& f = fr[(size_t)i];
        const char* chr = names[i].chr ? names[i].chr : ""; const char* label = names[i].label ? names[i].label : "";
        f.chr_off = (int32_t)pool.size(); f.chr_len = (int32_t)strlen(chr); pool += chr;
        f.label_off = (int32_t)pool.size(); f.label_len = (int32_t)strlen(label); pool += label;
        f.feature_start = names[i].feature_start; f.feature_stop = names[i].feature_stop;
        f.rb0 = n_rb;
        n_rb += (int64_t)d.n_pos * d.n_sizes;
    }
    *n_records = 0; *n_bytes = 0; h->fmt_bytes = 0;
    if (n_rb == 0) return MIPGEN_OK;
    if (n_rb + 1 > INT32_MAX) return fail(MIPGEN_E_INVALID, "window too large for record formatting (%lld row blocks)", (long long)n_rb);
    if (h->fmt_regions.reserve(fr.size()) || h->fmt_pool.reserve(std::max<size_t>(pool.size(), 1)) || h->fmt_a.reserve((size_t)n_rb + 1) || h->fmt_b.reserve((size_t)n_rb + 1) ||
        h->fmt_c.reserve((size_t)n_rb + 1) || h->fmt_d.reserve((size_t)n_rb + 1))
        return MIPGEN_E_NOMEM;
    size_t temp_bytes = 0;
    HIP_TRY(mipgen_scan_i64(h->stream, nullptr, &temp_bytes, h->fmt_a.p, h->fmt_b.p, n_rb + 1));
    if (h->fmt_temp.reserve(temp_bytes + 16)) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->fmt_regions.p, fr.data(), fr.size() * sizeof(FmtRegion), hipMemcpyHostToDevice, h->stream));
    if (!pool.empty()) HIP_TRY(hipMemcpyAsync(h->fmt_pool.p, pool.data(), pool.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->fmt_a.p + n_rb, 0, sizeof(int64_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->fmt_c.p + n_rb, 0, sizeof(int64_t), h->stream));
    // records per row block -> ranks; bytes per row block -> offsets; then the bytes
    HIP_TRY(mipgen_launch_fmt_count(h->stream, n_rb, W.r0, &FC, h->fmt_regions.p, h->dp, h->regions.p, h->emitted.p, h->fmt_a.p));
    HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->fmt_a.p, h->fmt_b.p, n_rb + 1));
    HIP_TRY(mipgen_launch_fmt_records(h->stream, 0, n_rb, W.r0, &FC, h->fmt_regions.p, h->fmt_pool.p, h->dp, h->regions.p, h->letters.p, h->copy.p, h->scores.p,
                                      h->records.p, h->emitted.p, h->fmt_b.p, nullptr, h->fmt_c.p, nullptr));
    HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->fmt_c.p, h->fmt_d.p, n_rb + 1));
    int64_t totals[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&totals[0], h->fmt_b.p + n_rb, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(&totals[1], h->fmt_d.p + n_rb, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = pb_check(h)) return rc_pb;
    if (h->fmt_text.reserve((size_t)std::max<int64_t>(totals[1], 1))) return MIPGEN_E_NOMEM;
    HIP_TRY(mipgen_launch_fmt_records(h->stream, 1, n_rb, W.r0, &FC, h->fmt_regions.p, h->fmt_pool.p, h->dp, h->regions.p, h->letters.p, h->copy.p, h->scores.p,
                                      h->records.p, h->emitted.p, h->fmt_b.p, h->fmt_d.p, nullptr, h->fmt_text.p));
    *n_records = totals[0]; *n_bytes = totals[1]; h->fmt_bytes = totals[1];
    return MIPGEN_OK;
}

int mipgen_accel_download_text(mipgen_accel* h, char* dst, int64_t capacity)
{
    if (!h || (!dst && capacity > 0)) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (h->fmt_bytes < 0) return fail(MIPGEN_E_STATE, "mipgen_accel_format_all_mips has not run on this window");
    if (capacity < h->fmt_bytes) return fail(MIPGEN_E_INVALID, "text capacity %lld < %lld bytes", (long long)capacity, (long long)h->fmt_bytes);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->fmt_bytes) HIP_TRY(hipMemcpy(dst, h->fmt_text.p, (size_t)h->fmt_bytes, hipMemcpyDeviceToHost));
    return MIPGEN_OK;
}

int mipgen_accel_download_survivors(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors, int64_t survivor_capacity)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    // batch-wide survivors: every result window must have been replayed + condensed since it was last scored
    if (h->win_state.empty()) return fail(MIPGEN_E_STATE, "mipgen_accel_score_condense_all has not run on this batch");
    for (size_t w = 0; w < h->win_state.size(); w++)
        if (!(h->win_state[w] & 1)) return fail(MIPGEN_E_STATE, "mipgen_accel_score_condense_all has not run on this batch (window %zu holds no current survivors)", w);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = pb_check(h)) return rc_pb;
    if (emitted_per_region && h->n_regions)
        HIP_TRY(hipMemcpy(emitted_per_region, h->emitted_per_region.p, (size_t)h->n_regions * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (survivors) {
        if (survivor_capacity < 2 * h->total_pos) return fail(MIPGEN_E_INVALID, "survivor capacity too small");
        if (h->total_pos) HIP_TRY(hipMemcpy(survivors, h->survivors.p, (size_t)(2 * h->total_pos) * sizeof(mipgen_survivor), hipMemcpyDeviceToHost));
    }
    return MIPGEN_OK;
}

int mipgen_accel_survivors_device_ptr(const mipgen_accel* h, void** survivors_dev, int64_t* n_survivors)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (survivors_dev) *survivors_dev = h->survivors.p;
    if (n_survivors) *n_survivors = 2 * h->total_pos;
    return MIPGEN_OK;
}

// ---- mixed designs: every condensed survivor of the window scored last through the SVR, on the device ------------------------------------------
// The reference re-scores the MIPs its pick stage tests, one at a time (mipgen.cpp:1523-1527, 1533-1537, 1546-1550, 1873-1877); here all survivors
// of the window go through the list scorer in ONE call - candidate list built on the device from the survivor array (slot order), features +
// matrix-core SVR (short lists: the literal per-candidate kernel), print-exact re-score - and the values stay in HBM beside the survivors
// (slot for slot; NaN where a slot holds no survivor).  Same kernels, same values as mipgen_accel_score_candidates on the same list.
int mipgen_accel_rescore_survivors(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (!h->replayed || h->cur_window < 0) return fail(MIPGEN_E_STATE, "mipgen_accel_replay_condense has not run on these scores");
    if (h->model.p == nullptr) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    const int w = h->cur_window;
    const Window& W = h->windows[(size_t)w];
    const int64_t n = 2 * W.n_pos;
    HIP_TRY(hipSetDevice(h->device));
    if (h->surv_svr.reserve((size_t)std::max<int64_t>(2 * h->total_pos, 1))) return MIPGEN_E_NOMEM;
    if (n > 0) {
        if (h->rs_keep.reserve((size_t)n + 1) || h->rs_offs.reserve((size_t)n + 1) || h->rs_idx.reserve((size_t)n) || h->cand_in.reserve((size_t)n)) return MIPGEN_E_NOMEM;
        const mipgen_survivor* surv = h->survivors.p + 2 * W.pos0;
        double* svr = h->surv_svr.p + 2 * W.pos0;
        HIP_TRY(mipgen_launch_surv_keep(h->stream, surv, n, h->rs_keep.p, svr));
        size_t temp_bytes = 0;
        HIP_TRY(mipgen_scan_i64(h->stream, nullptr, &temp_bytes, h->rs_keep.p, h->rs_offs.p, n + 1));
        if (h->fmt_temp.reserve(temp_bytes + 16)) return MIPGEN_E_NOMEM;
        HIP_TRY(mipgen_scan_i64(h->stream, h->fmt_temp.p, &temp_bytes, h->rs_keep.p, h->rs_offs.p, n + 1));
        HIP_TRY(mipgen_launch_surv_candidates(h->stream, h->dp, h->regions.p, W.r0, W.r1, surv, n, W.cand0, h->rs_offs.p, h->cand_in.p, h->rs_idx.p));
        int64_t m = 0;
        HIP_TRY(hipMemcpyAsync(&m, h->rs_offs.p + n, sizeof m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (m > 0) {
            if (m > INT32_MAX) return fail(MIPGEN_E_INVALID, "too many survivors in one window for the list scorer");
            if (h->cand_scores.reserve((size_t)m) || h->cand_records.reserve((size_t)m)) return MIPGEN_E_NOMEM;
            const bool batched = m >= 256;                                   // as mipgen_accel_score_candidates decides
            if (batched) {
                if (h->cand_feats.reserve((size_t)m * MIPGEN_N_FEATURES)) return MIPGEN_E_NOMEM;
                HIP_TRY(mipgen_launch_features_batch(h->stream, (int)m, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts, h->cand_records.p,
                                                     h->cand_feats.p));
                HIP_TRY(mipgen_launch_svr_gemm(h->stream, (int)m, h->cand_feats.p, h->cand_records.p, h->model_t.p, h->sv_norm.p, h->sv_coef.p, h->sv_center.p, h->n_sv_pad,
                                               h->gamma, h->rho, h->cand_scores.p));
                if (int rc = fix_print_boundaries(h, 0, 0, h->cand_in.p, h->cand_scores.p, h->cand_records.p, m)) return rc;
            } else {
                HIP_TRY(mipgen_launch_candidates(h->stream, (int)m, h->dp, h->regions.p, h->cand_in.p, h->bases.p, h->copy.p, h->unmap.p, h->dconsts,
                                                 h->model.p, h->n_sv, h->gamma, h->rho, MIPGEN_SCORE_SVR, h->cand_scores.p, h->cand_records.p, nullptr, nullptr, 0, nullptr));
            }
            HIP_TRY(mipgen_launch_scatter_f64(h->stream, h->cand_scores.p, h->rs_idx.p, m, svr));
        }
    }
    if ((size_t)w < h->win_state.size()) h->win_state[(size_t)w] |= 4;
    return MIPGEN_OK;
}

int mipgen_accel_download_survivor_scores(mipgen_accel* h, int32_t window, double* svr, int64_t capacity)
{
    if (!h || !svr || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    if ((size_t)window >= h->win_state.size() || !(h->win_state[(size_t)window] & 4)) return fail(MIPGEN_E_STATE, "mipgen_accel_rescore_survivors has not run on the current survivors of window %d", window);
    const Window& W = h->windows[(size_t)window];
    if (capacity < 2 * W.n_pos) return fail(MIPGEN_E_INVALID, "capacity too small");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc_pb = pb_check(h)) return rc_pb;
    if (W.n_pos) HIP_TRY(hipMemcpy(svr, h->surv_svr.p + 2 * W.pos0, (size_t)(2 * W.n_pos) * sizeof(double), hipMemcpyDeviceToHost));
    return MIPGEN_OK;
}

// ---- device-side views of a result window: for a caller that moves results between devices itself (the RCCL gather of the multi-GPU front end) ----
int mipgen_accel_window_views(mipgen_accel* h, int32_t window, mipgen_window_views* out)
{
    if (!h || !out || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    const uint8_t st = (size_t)window < h->win_state.size() ? h->win_state[(size_t)window] : 0;
    if (!(st & 1)) return fail(MIPGEN_E_STATE, "window %d holds no current survivors (replay + condense first)", window);
    const Window& W = h->windows[(size_t)window];
    memset(out, 0, sizeof *out);
    out->emitted = h->emitted_per_region.p + W.r0; out->n_emitted = W.r1 - W.r0;
    out->survivors = h->survivors.p + 2 * W.pos0; out->n_survivors = 2 * W.n_pos;
    if (st & 2) { out->collapsed = h->collapsed.p + W.base0; out->n_collapsed = W.n_base_entries; }
    if (st & 4) out->survivor_svr = h->surv_svr.p + 2 * W.pos0;
    if (window == h->cur_window && h->fmt_bytes > 0) { out->text = h->fmt_text.p; out->n_text_bytes = h->fmt_bytes; }
    out->first_candidate = W.cand0;
    return MIPGEN_OK;
}

int mipgen_accel_synchronize(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return pb_check(h);
}

// ---- section 8f-3: arm-oligo copy numbers by exact k-mer counting (opt-in replacement of the bwa round trip) ----------------
namespace {

// device state of one counting run; `out` = int32 [n_k][total] over the concatenated region strings (kernels_kmer.hip: k_kmer_lookup)
struct KmerRun {
    KmerParams KP;
    int64_t total = 0;
    int64_t pad = 0;                          // 'N' bytes behind the concatenation (readers that run past the last region: kernels_window.hip)
    std::vector<int64_t> roff;                // start of every region in the concatenation (one separator after each), then `total`
    DevBuf<char> dq, dg;
    DevBuf<uint64_t> dkeys;
    DevBuf<unsigned int> dcounts;
    DevBuf<uint32_t> dfilter, dfolded;
    DevBuf<int32_t> out;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit KmerRun(DevPool* pool) { dq.pool = dg.pool = dkeys.pool = dcounts.pool = dfilter.pool = dfolded.pool = out.pool = pool; }
    ~KmerRun()
    {
        dq.release(); dg.release(); dkeys.release(); dcounts.release(); dfilter.release(); dfolded.release(); out.release();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// insert the regions' k-mers, stream the genome past them, look every region position up again: K.out is filled (on the stream) on return
int kmer_count_run(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                   const char* const* region_seqs, const int32_t* region_lens, int32_t n_lengths, const int32_t* lengths, KmerRun& K)
{
    DIAG_CLOCK("oligo copies");
    KmerParams& KP = K.KP;
    memset(&KP, 0, sizeof KP);
    KP.n_k = n_lengths;
    for (int i = 0; i < n_lengths; i++) {
        if (lengths[i] < 1 || lengths[i] > 31 || (i && lengths[i] <= lengths[i - 1])) return fail(MIPGEN_E_INVALID, "oligo lengths must be ascending and <= 31 (exact 2-bit keys)");
        KP.k[i] = lengths[i];
    }
    KP.kmax = lengths[n_lengths - 1];
    // region sequences, separated by one 'N' (no k-mer crosses it)
    int64_t total = 0;
    K.roff.resize((size_t)n_regions + 1);
    for (int r = 0; r < n_regions; r++) {
        if (region_lens[r] < 0 || !region_seqs[r]) return fail(MIPGEN_E_INVALID, "region %d: no sequence", r);
        K.roff[(size_t)r] = total; total += (int64_t)region_lens[r] + 1;
    }
    K.roff[(size_t)n_regions] = total;
    K.total = total;
    std::vector<char> q((size_t)(total + K.pad), 'N');
    for (int r = 0; r < n_regions; r++) memcpy(&q[(size_t)K.roff[(size_t)r]], region_seqs[r], (size_t)region_lens[r]);
    uint64_t cap = 1024;
    while (cap < 2 * (uint64_t)total) cap <<= 1;
    KP.cap_mask = cap - 1;
    // Bloom filter of the regions' canonical kmin-mers: ~32 bits per region position (2-3 % false positives), at least the size of its LDS fold
    KP.filter_bits = 18;
    while (KP.filter_bits < 30 && (1ull << KP.filter_bits) < 32ull * (uint64_t)total) KP.filter_bits++;
    int64_t gmax = 0;
    for (int c = 0; c < n_chrom; c++) gmax = std::max(gmax, chrom_lens[c]);
    const size_t tab = (size_t)cap * (size_t)n_lengths;
    if (K.dq.reserve((size_t)(total + K.pad)) || K.dg.reserve((size_t)std::max<int64_t>(gmax, 1)) || K.dkeys.reserve(tab) || K.dcounts.reserve(tab) ||
        K.dfilter.reserve((size_t)1 << (KP.filter_bits - 5)) || K.dfolded.reserve((size_t)1 << 13) || K.out.reserve((size_t)total * (size_t)n_lengths))
        return MIPGEN_E_NOMEM;
    DIAG_LAP("concatenate + device buffers");
    HIP_TRY(hipEventCreate(&K.e0)); HIP_TRY(hipEventCreate(&K.e1));
    HIP_TRY(hipMemcpyAsync(K.dq.p, q.data(), (size_t)(total + K.pad), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(K.dkeys.p, 0xFF, tab * sizeof(uint64_t), h->stream));
    HIP_TRY(hipMemsetAsync(K.dcounts.p, 0, tab * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(K.dfilter.p, 0, ((size_t)1 << (KP.filter_bits - 5)) * sizeof(uint32_t), h->stream));
    HIP_TRY(mipgen_launch_kmer_insert(h->stream, K.dq.p, total, &KP, K.dkeys.p, K.dfilter.p));
    HIP_TRY(mipgen_launch_kmer_fold(h->stream, K.dfilter.p, KP.filter_bits, K.dfolded.p));
    HIP_TRY(hipStreamSynchronize(h->stream));                          // q dies with this scope
    DIAG_LAP("insert");
    double ms_total = 0.0;
    int64_t gbytes = 0;
    for (int c = 0; c < n_chrom; c++) {                                // one streaming pass per chromosome: 1 byte per genome base
        if (chrom_lens[c] <= 0) continue;
        HIP_TRY(hipMemcpyAsync(K.dg.p, chrom_seqs[c], (size_t)chrom_lens[c], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipEventRecord(K.e0, h->stream));
        HIP_TRY(mipgen_launch_kmer_count(h->stream, K.dg.p, chrom_lens[c], &KP, K.dkeys.p, K.dfilter.p, K.dfolded.p, K.dcounts.p, h->n_cu));
        HIP_TRY(hipEventRecord(K.e1, h->stream));
        HIP_TRY(hipEventSynchronize(K.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, K.e0, K.e1));
        ms_total += ms; gbytes += chrom_lens[c];
    }
    DIAG_LAP("genome passes");
    HIP_TRY(mipgen_launch_kmer_lookup(h->stream, K.dq.p, total, &KP, K.dkeys.p, K.dcounts.p, K.out.p));
    h->kmer_count_ms = ms_total; h->kmer_genome_bytes = gbytes;
    return MIPGEN_OK;
}

}  // namespace

int mipgen_accel_count_oligo_copies(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                    const char* const* region_seqs, const int32_t* region_lens, int32_t n_lengths, const int32_t* lengths,
                                    int32_t* const* copy_out)
{
    if (!h || n_chrom < 0 || n_regions < 0 || n_lengths < 1 || n_lengths > MIPGEN_MAX_OLIGO || !lengths || (n_chrom && (!chrom_seqs || !chrom_lens)) ||
        (n_regions && (!region_seqs || !region_lens || !copy_out)))
        return fail(MIPGEN_E_INVALID, "bad arguments");
    if (n_regions == 0) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    KmerRun K(&h->pool);
    if (int rc = kmer_count_run(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, n_lengths, lengths, K)) return rc;
    // back to the host one oligo length at a time through two pinned buffers: the copy of length s + 1 runs under the scatter of length s
    const int64_t total = K.total;
    PinnedPair<int32_t> pin;
    HIP_TRY(pin.alloc((size_t)total));
    HIP_TRY(hipMemcpyAsync(pin.buf[0], K.out.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(pin.done[0], h->stream));
    pin.busy[0] = true;
    for (int s = 0; s < n_lengths; s++) {
        const int b = s & 1;
        if (s + 1 < n_lengths) {
            HIP_TRY(hipMemcpyAsync(pin.buf[b ^ 1], K.out.p + (size_t)(s + 1) * (size_t)total, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipEventRecord(pin.done[b ^ 1], h->stream));
            pin.busy[b ^ 1] = true;
        }
        HIP_TRY(pin.wait(b));
        for (int r = 0; r < n_regions; r++) {
            const int len = region_lens[r];
            if (!copy_out[r]) continue;
            int32_t* dst = copy_out[r] + (size_t)s * (size_t)len;
            memcpy(dst, pin.buf[b] + K.roff[(size_t)r], (size_t)len * sizeof(int32_t));
            for (int i = std::max(0, len - lengths[s]); i < len; i++) dst[i] = 0;      // oligos the reference never writes (mipgen.cpp:829): absent key -> 0
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_count_oligo_copies_resident(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                             const char* const* region_seqs, const int32_t* region_lens, int64_t* n_big, const mipgen_big_copy** big)
{
    if (!h || n_chrom < 0 || n_regions < 0 || (n_chrom && (!chrom_seqs || !chrom_lens)) || (n_regions && (!region_seqs || !region_lens)))
        return fail(MIPGEN_E_INVALID, "bad arguments");
    h->resident_lens.clear();
    h->big_copies.clear();
    if (n_big) *n_big = 0;
    if (big) *big = nullptr;
    if (n_regions == 0) return MIPGEN_OK;
    const DevParams& D = h->hp;
    std::vector<int32_t> lengths;                                      // slot order = ascending oligo length (create_handle)
    for (int len = 0; len <= MIPGEN_MAX_OLIGO; len++) if (D.len_slot[len] >= 0) lengths.push_back(len);
    if ((int)lengths.size() != D.n_len_slots || lengths.empty()) return fail(MIPGEN_E_INVALID, "internal: oligo length slots");
    HIP_TRY(hipSetDevice(h->device));
    KmerRun K(&h->pool);
    if (int rc = kmer_count_run(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, (int32_t)lengths.size(), lengths.data(), K)) return rc;
    const int64_t copy_total = (K.total - n_regions) * (int64_t)D.n_len_slots;
    const unsigned int big_cap = (unsigned int)std::min<int64_t>(std::max<int64_t>(K.total / 4, (int64_t)1 << 16), (int64_t)1 << 28);
    DevBuf<int64_t> droff;
    DevBuf<mipgen_big_copy> dbig;
    DevBuf<unsigned int> dn;
    struct Free { DevBuf<int64_t>& a; DevBuf<mipgen_big_copy>& b; DevBuf<unsigned int>& c; ~Free() { a.release(); b.release(); c.release(); } } free_all{droff, dbig, dn};
    if (h->copy.reserve((size_t)std::max<int64_t>(copy_total, 1)) || droff.reserve(K.roff.size()) || dbig.reserve(big_cap) || dn.reserve(1)) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(droff.p, K.roff.data(), K.roff.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(dn.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(mipgen_launch_kmer_place(h->stream, K.out.p, K.total, &K.KP, droff.p, n_regions, h->copy.p, dbig.p, dn.p, big_cap));
    unsigned int nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, dn.p, sizeof nb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (nb > big_cap) return fail(MIPGEN_E_NOMEM, "%u oligos with 65535 or more copies (list capacity %u): use mipgen_accel_count_oligo_copies", nb, big_cap);
    h->big_copies.resize(nb);
    if (nb) HIP_TRY(hipMemcpy(h->big_copies.data(), dbig.p, (size_t)nb * sizeof(mipgen_big_copy), hipMemcpyDeviceToHost));
    std::sort(h->big_copies.begin(), h->big_copies.end(), [](const mipgen_big_copy& x, const mipgen_big_copy& y) {
        return x.region != y.region ? x.region < y.region : (x.length != y.length ? x.length < y.length : x.start < y.start); });
    h->resident_lens.assign(region_lens, region_lens + n_regions);
    if (n_big) *n_big = (int64_t)nb;
    if (big) *big = h->big_copies.data();
    return MIPGEN_OK;
}


// ---- section 8f-3, second half: uniqueness of whole capture windows (kernels_window.hip) ---------------------------------------------------
extern "C" hipError_t mipgen_launch_window_spans(hipStream_t st, const char* q, const int64_t* roff, int n_regions, uint16_t* dist_bad, uint16_t* dist_end, uint16_t* dist_start);
extern "C" hipError_t mipgen_launch_seed_index(hipStream_t st, const char* q, int64_t total, int k, const uint64_t* keys, uint64_t cap_mask, unsigned int* rmult,
                                               unsigned int* rstart, unsigned int* rfill, uint32_t* rlist, unsigned int* alloc, int phase);
extern "C" hipError_t mipgen_launch_window_verify(hipStream_t st, const char* G, int64_t glen, const char* q, int64_t total, const int32_t* sizes, int n_sizes, int k,
                                                  const uint64_t* keys, uint64_t cap_mask, const unsigned int* counts, const uint32_t* filter, int filter_bits,
                                                  const unsigned int* rmult, const unsigned int* rstart, const uint32_t* rlist, const uint16_t* dist_start, unsigned int* ctr);
extern "C" hipError_t mipgen_launch_window_flags(hipStream_t st, const char* q, int64_t total, const int32_t* sizes, int n_sizes, int k, const uint64_t* keys,
                                                 uint64_t cap_mask, const unsigned int* counts, const uint16_t* dist_bad, const uint16_t* dist_end, const unsigned int* ctr,
                                                 uint8_t* unmap, const int64_t* roff, int n_regions, const int32_t* bounds, uint8_t* any);

// bounds == nullptr: the full flag image goes to unmap_out (mipgen_accel_window_uniqueness).  bounds != nullptr: the flags are restricted to the
// window starts the reference looks up on the device, the image stays in the handle (h->win_img) and any_out gets one byte per region
// (mipgen_accel_window_uniqueness_begin)
static int window_uniqueness_impl(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                  const char* const* region_seqs, const int32_t* region_lens, int32_t n_sizes, const int32_t* sizes, int32_t seed_len,
                                  uint8_t* const* unmap_out, const mipgen_window_bounds* bounds, uint8_t* any_out)
{
    if (!h || n_chrom < 0 || n_regions < 0 || n_sizes < 1 || n_sizes > 64 || !sizes || (n_chrom && (!chrom_seqs || !chrom_lens)) ||
        (n_regions && (!region_seqs || !region_lens || (!bounds && !unmap_out) || (bounds && !any_out))))
        return fail(MIPGEN_E_INVALID, "bad arguments");
    if (seed_len < 12 || seed_len > 31) return fail(MIPGEN_E_INVALID, "seed length %d: must be in [12, 31] (exact 2-bit keys)", seed_len);
    int max_size = 0;
    for (int i = 0; i < n_sizes; i++) {
        if (sizes[i] < 2 * seed_len || sizes[i] > 60000) return fail(MIPGEN_E_INVALID, "capture size %d: must be in [2 x seed length, 60000] (two disjoint seeds per window)", sizes[i]);
        max_size = std::max(max_size, sizes[i]);
    }
    if (n_regions == 0) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    KmerRun K(&h->pool);
    K.pad = (int64_t)max_size + 1;
    const int32_t lengths[1] = {seed_len};
    if (int rc = kmer_count_run(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, 1, lengths, K)) return rc;    // seeds + their genome loci
    const int64_t total = K.total;
    if (total >= ((int64_t)1 << 31)) return fail(MIPGEN_E_INVALID, "window uniqueness: more than 2^31 region bases in one call");
    const uint64_t cap = K.KP.cap_mask + 1;
    DevBuf<unsigned int> rmult, rstart, rfill, alloc, ctr;
    DevBuf<uint32_t> rlist;
    DevBuf<uint16_t> dbad, dend, dstart;
    DevBuf<int64_t> droff;
    DevBuf<uint8_t> dun, dany;
    DevBuf<int32_t> dbounds;
    h->win_img.release(); h->win_roff.clear(); h->win_lens.clear(); h->win_sizes = 0; h->win_total = 0;
    struct Free { std::vector<std::function<void()>> f; ~Free() { for (auto& g : f) g(); } } fr;     // DevBuf has no destructor: release on every exit
    fr.f = {[&] { rmult.release(); }, [&] { rstart.release(); }, [&] { rfill.release(); }, [&] { alloc.release(); }, [&] { ctr.release(); }, [&] { rlist.release(); },
            [&] { dbad.release(); }, [&] { dend.release(); }, [&] { dstart.release(); }, [&] { droff.release(); }, [&] { dun.release(); }, [&] { dany.release(); },
            [&] { dbounds.release(); }};
    if (bounds) {
        if (dany.reserve((size_t)n_regions) || dbounds.reserve(4 * (size_t)n_regions)) return MIPGEN_E_NOMEM;
        static_assert(sizeof(mipgen_window_bounds) == 16, "four int32");
        HIP_TRY(hipMemcpyAsync(dbounds.p, bounds, (size_t)n_regions * sizeof(mipgen_window_bounds), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(dany.p, 0, (size_t)n_regions, h->stream));
    }
    if (rmult.reserve(cap) || rstart.reserve(cap) || rfill.reserve(cap) || alloc.reserve(1) || ctr.reserve((size_t)n_sizes * (size_t)total) || rlist.reserve((size_t)total) ||
        dbad.reserve((size_t)total) || dend.reserve((size_t)total) || dstart.reserve((size_t)total) || droff.reserve(K.roff.size()) ||
        dun.reserve((size_t)n_sizes * (size_t)total))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemsetAsync(rmult.p, 0, cap * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(rfill.p, 0, cap * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(alloc.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(ctr.p, 0, (size_t)n_sizes * (size_t)total * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemcpyAsync(droff.p, K.roff.data(), K.roff.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mipgen_launch_window_spans(h->stream, K.dq.p, droff.p, n_regions, dbad.p, dend.p, dstart.p));
    for (int phase = 0; phase < 3; phase++)
        HIP_TRY(mipgen_launch_seed_index(h->stream, K.dq.p, total, seed_len, K.dkeys.p, K.KP.cap_mask, rmult.p, rstart.p, rfill.p, rlist.p, alloc.p, phase));
    for (int c = 0; c < n_chrom; c++) {                                // second genome pass: extend the loci of the repeated seeds
        if (chrom_lens[c] <= 0) continue;
        HIP_TRY(hipMemcpyAsync(K.dg.p, chrom_seqs[c], (size_t)chrom_lens[c], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(mipgen_launch_window_verify(h->stream, K.dg.p, chrom_lens[c], K.dq.p, total, sizes, n_sizes, seed_len, K.dkeys.p, K.KP.cap_mask, K.dcounts.p,
                                            K.dfilter.p, K.KP.filter_bits, rmult.p, rstart.p, rlist.p, dstart.p, ctr.p));
        HIP_TRY(hipStreamSynchronize(h->stream));                      // the next chromosome overwrites the genome buffer
    }
    HIP_TRY(mipgen_launch_window_flags(h->stream, K.dq.p, total, sizes, n_sizes, seed_len, K.dkeys.p, K.KP.cap_mask, K.dcounts.p, dbad.p, dend.p, ctr.p, dun.p,
                                       droff.p, n_regions, bounds ? dbounds.p : nullptr, bounds ? dany.p : nullptr));
    if (bounds) {
        // the image stays on the device: the caller fetches the few regions that have a flagged start (mipgen_accel_window_flags_region)
        HIP_TRY(hipMemcpyAsync(any_out, dany.p, (size_t)n_regions, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->win_img = dun; dun = DevBuf<uint8_t>();                     // ownership moves to the handle (released by _end / the next call / destroy)
        h->win_roff = K.roff; h->win_lens.assign(region_lens, region_lens + n_regions); h->win_sizes = n_sizes; h->win_total = total;
        return MIPGEN_OK;
    }
    std::vector<uint8_t> img((size_t)n_sizes * (size_t)total);
    HIP_TRY(hipMemcpyAsync(img.data(), dun.p, img.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int r = 0; r < n_regions; r++) {
        if (!unmap_out[r]) continue;
        const int len = region_lens[r];
        for (int c = 0; c < n_sizes; c++) memcpy(unmap_out[r] + (size_t)c * (size_t)len, &img[(size_t)c * (size_t)total + (size_t)K.roff[(size_t)r]], (size_t)len);
    }
    return MIPGEN_OK;
}

int mipgen_accel_window_uniqueness(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                   const char* const* region_seqs, const int32_t* region_lens, int32_t n_sizes, const int32_t* sizes, int32_t seed_len,
                                   uint8_t* const* unmap_out)
{
    return window_uniqueness_impl(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, n_sizes, sizes, seed_len, unmap_out, nullptr, nullptr);
}

int mipgen_accel_window_uniqueness_begin(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                         const char* const* region_seqs, const int32_t* region_lens, const mipgen_window_bounds* bounds,
                                         int32_t n_sizes, const int32_t* sizes, int32_t seed_len, uint8_t* any_out)
{
    if (!bounds || !any_out) return fail(MIPGEN_E_INVALID, "bad arguments");
    return window_uniqueness_impl(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, n_sizes, sizes, seed_len, nullptr, bounds, any_out);
}

int mipgen_accel_window_flags_region(mipgen_accel* h, int32_t region, uint8_t* out)
{
    if (!h || !out) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (!h->win_img.p || region < 0 || (size_t)region >= h->win_lens.size()) return fail(MIPGEN_E_STATE, "no window-uniqueness image in the handle for region %d", region);
    HIP_TRY(hipSetDevice(h->device));
    const size_t len = (size_t)h->win_lens[(size_t)region];
    if (len == 0) return MIPGEN_OK;
    HIP_TRY(hipMemcpy2DAsync(out, len, h->win_img.p + h->win_roff[(size_t)region], (size_t)h->win_total, len, (size_t)h->win_sizes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_window_uniqueness_end(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->win_img.release(); h->win_roff.clear(); h->win_lens.clear(); h->win_sizes = 0; h->win_total = 0;
    return MIPGEN_OK;
}

double mipgen_accel_last_kernel_ms(mipgen_accel* h, int32_t which)
{
    if (h && which == 4) return h->kmer_count_ms;
    if (h && which == 5) return h->list_svr_ms;
    if (h && which == 6) return h->list_feat_ms;
    if (!h || !h->timing) return -1.0;
    if (hipSetDevice(h->device) != hipSuccess) return -1.0;
    double total = 0.0;
    bool any = false;
    for (size_t w = 0; w < h->windows.size() && 4 * (w + 1) <= h->ev.size(); w++) {
        const uint8_t used = h->ev_used[w];
        if (!(used & 1)) continue;
        hipEvent_t* ev = &h->ev[4 * w];
        float ms = 0.f;
        hipError_t e = hipSuccess;
        if (which == 3) {                                                        // replay + condense (incl. its memsets)
            if (!(used & 2)) continue;
            if (hipEventSynchronize(ev[3]) != hipSuccess) return -1.0;
            e = hipEventElapsedTime(&ms, ev[2], ev[3]);
        } else {
            if (hipEventSynchronize(ev[2]) != hipSuccess) return -1.0;
            if (which == 0) e = hipEventElapsedTime(&ms, ev[1], ev[2]);          // SVR kernel (~0 for logistic)
            else if (which == 2) e = hipEventElapsedTime(&ms, ev[0], ev[1]);     // records / logistic kernel
            else e = hipEventElapsedTime(&ms, ev[0], ev[2]);
        }
        if (e != hipSuccess) return -1.0;
        total += ms; any = true;
    }
    return any ? total : -1.0;
}

int mipgen_accel_set_timing(mipgen_accel* h, int32_t enabled)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->timing = enabled != 0;
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    return MIPGEN_OK;
}

}  // extern "C"
