// stub_accel.cpp — a CPU stand-in for libmipgen_accel.so.  TEST INFRASTRUCTURE ONLY (never shipped, never linked by the product).
//
// It implements the part of include/mipgen_accel.h that libmipgen_host's accelerated tile_regions driver calls (mipgen_amd/host/design.cpp:
// device workers, result windows, channels, abort paths) by handing out what the ORACLE computes (oracle/mipgen_oracle.c), so that the driver's
// threads can run on a machine without a GPU under ThreadSanitizer / AddressSanitizer (tests/test_host_threads_cpu.py) and its output files can
// be compared with the reference's goldens there.  Knobs (environment, read here only):
//   STUB_ACCEL_DEVICES=n        what mipgen_accel_device_count() reports (default 1)
//   STUB_ACCEL_FAIL=dev:call:n  the n-th call (1-based) of mipgen_accel_<call> on a handle of device `dev` fails with MIPGEN_E_HIP
//   STUB_ACCEL_FAKE=1           PROFILING ONLY: nothing is scored - every scan position and strand gets a fabricated survivor (hashed capture size, arm pair
//                               and score; every copy number 1), the resident k-mer counter "counts" nothing, so that the HOST side of a 200,000-exon silent design
//                               (input stage, block dealing, selection stage) can be timed and profiled on a machine without a GPU (tools/exp/host_profile.sh)
// The collapse fold (mipgen.cpp:1616-1649) and the all_mips generation order (mipgen.cpp:421-491) are restated here; everything per candidate
// comes from the oracle.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mipgen_accel.h"
#include "../../oracle/mipgen_oracle.h"

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

struct RegionStore {
    mipgen_region r;
    std::string seq, masked;
    std::vector<std::vector<int32_t>> copy;
    std::vector<const int32_t*> copy_ptrs;
    std::vector<uint8_t> unmap, snp;
    bool has_masked = false, has_copy = false;
};

struct Win { int r0 = 0, r1 = 0; int64_t cand0 = 0, n_cand = 0, pos0 = 0, n_pos = 0, base0 = 0, n_base = 0; };

}  // namespace

struct mipgen_accel {
    mipgen_params P;
    int device = 0;
    mo_model* model = nullptr;
    int64_t window_cap = 0;
    std::vector<int32_t> window_breaks;
    std::vector<RegionStore> regions;
    std::vector<mipgen_grid> grids;
    std::vector<int64_t> region_pos0, region_base0;       // + totals at the end
    std::vector<Win> windows;
    int cur = -1;
    bool scored = false, replayed = false;
    std::vector<double> scores;                            // of the window scored last
    std::vector<uint64_t> records;
    std::vector<uint8_t> emitted;
    std::vector<int64_t> emitted_per_region;
    std::vector<mipgen_survivor> survivors;                // batch-wide
    std::vector<double> surv_svr;
    std::vector<int32_t> collapsed;
    std::vector<uint8_t> win_state;
    std::string text;
    bool have_text = false;
    std::map<std::string, int> calls;
};

namespace {

bool fake_mode() { static const bool f = getenv("STUB_ACCEL_FAKE") != nullptr; return f; }
inline uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

int max_scan_of(const mipgen_params& P, const mipgen_grid& g)
{
    int min_sum = 1 << 30;
    for (int i = 0; i < P.n_arm_pairs; i++) min_sum = std::min(min_sum, P.arm_ext[i] + P.arm_lig[i]);
    return P.max_capture_size - g.first_size_index * P.capture_increment - min_sum;
}

// STUB_ACCEL_FAIL=dev:call:n
int injected(mipgen_accel* h, const char* call)
{
    const int n = ++h->calls[call];
    const char* e = getenv("STUB_ACCEL_FAIL");
    if (!e) return 0;
    int dev = -1, nth = 0;
    char name[64] = "";
    if (sscanf(e, "%d:%63[^:]:%d", &dev, name, &nth) != 3) return 0;
    if (dev == h->device && nth == n && strcmp(name, call) == 0) return fail(MIPGEN_E_HIP, "stub accelerator: injected failure of %s (call %d, device %d)", call, n, dev);
    return 0;
}
#define INJECT(h, call) do { if (int rc_ = injected(h, call)) return rc_; } while (0)

mipgen_candidate cand_of(const mipgen_params& P, const mipgen_grid& g, int region, int64_t local)
{
    const int A = P.n_arm_pairs;
    const int a = (int)(local % A);
    const int64_t row = local / A;
    const int64_t rest = row >> 1;
    const int ki = (int)(rest % g.n_sizes), pi = (int)(rest / g.n_sizes);
    mipgen_candidate c;
    c.region = region; c.scan_start = g.first_pos + pi; c.capture_size = P.max_capture_size - (g.first_size_index + ki) * P.capture_increment;
    c.ext_len = P.arm_ext[a]; c.lig_len = P.arm_lig[a]; c.strand = (int)(row & 1);
    return c;
}

}  // namespace

extern "C" {

int mipgen_accel_abi_version(void) { return MIPGEN_ACCEL_ABI_VERSION; }
const char* mipgen_accel_last_error(void) { return g_err; }
int mipgen_accel_device_count(void)
{
    const char* e = getenv("STUB_ACCEL_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n > 0 ? n : 0;
}

int mipgen_accel_create(const mipgen_params* params, int device, void*, mipgen_accel** out)
{
    if (!params || !out) return fail(MIPGEN_E_INVALID, "null argument");
    if (params->abi_version != MIPGEN_ACCEL_ABI_VERSION) return fail(MIPGEN_E_INVALID, "abi_version %d != %d", params->abi_version, MIPGEN_ACCEL_ABI_VERSION);
    if (device < 0 || device >= mipgen_accel_device_count()) return fail(MIPGEN_E_NODEVICE, "stub accelerator: no device %d", device);
    mipgen_accel* h = new mipgen_accel();
    h->P = *params;
    if (h->P.capture_increment == 0) h->P.capture_increment = 1;
    h->device = device;
    if (int rc = injected(h, "create")) { delete h; return rc; }
    *out = h;
    return MIPGEN_OK;
}

void mipgen_accel_destroy(mipgen_accel* h)
{
    if (!h) return;
    if (h->model) mo_svm_free_model(h->model);
    delete h;
}

int mipgen_accel_load_model_file(mipgen_accel* h, const char* path)
{
    if (!h || !path) return fail(MIPGEN_E_INVALID, "null argument");
    INJECT(h, "load_model_file");
    if (h->model) { mo_svm_free_model(h->model); h->model = nullptr; }
    h->model = mo_svm_load_model(path);
    if (!h->model) return fail(MIPGEN_E_MODEL, "cannot open model file %s", path);
    return MIPGEN_OK;
}

int mipgen_accel_set_dynamic_skip(mipgen_accel* h, int32_t) { return h ? MIPGEN_OK : fail(MIPGEN_E_INVALID, "null handle"); }
int mipgen_accel_set_window_candidates(mipgen_accel* h, int64_t cap)
{
    if (!h || cap < 0) return fail(MIPGEN_E_INVALID, "bad argument");
    h->window_cap = cap;
    return MIPGEN_OK;
}

int mipgen_accel_set_window_breaks(mipgen_accel* h, const int32_t* first_regions, int32_t n)
{
    if (!h || n < 0 || (n > 0 && !first_regions)) return fail(MIPGEN_E_INVALID, "bad arguments");
    h->window_breaks.assign(first_regions, first_regions + n);
    return MIPGEN_OK;
}

int mipgen_accel_long_range_content_batch(mipgen_accel* h, int32_t n, const char* const* seqs, const int32_t* lens, const int32_t* starts, const int32_t* stops, double* out)
{
    if (!h || n < 0) return fail(MIPGEN_E_INVALID, "bad arguments");
    INJECT(h, "long_range_content_batch");
    for (int i = 0; i < n; i++) {
        const std::string s(seqs[i], (size_t)lens[i]);
        mo_long_range_content(s.c_str(), starts[i], stops[i], out + (size_t)i * MIPGEN_N_LRC);
    }
    return MIPGEN_OK;
}

int mipgen_accel_upload_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, mipgen_grid* grids_out)
{
    if (!h || n < 0 || (n > 0 && !regions)) return fail(MIPGEN_E_INVALID, "bad arguments");
    INJECT(h, "upload_regions");
    const mipgen_params& P = h->P;
    const int n_sizes_all = (P.max_capture_size - P.min_capture_size) / P.capture_increment + 1;
    h->regions.assign((size_t)n, RegionStore());
    h->grids.assign((size_t)n, mipgen_grid());
    h->region_pos0.assign((size_t)n + 1, 0); h->region_base0.assign((size_t)n + 1, 0);
    int64_t off = 0;
    for (int i = 0; i < n; i++) {
        RegionStore& S = h->regions[(size_t)i];
        const mipgen_region& R = regions[i];
        if (R.copy == MIPGEN_COPY_RESIDENT && !fake_mode()) return fail(MIPGEN_E_INVALID, "stub accelerator: resident copy tables are not supported");
        S.r = R;
        if (R.copy == MIPGEN_COPY_RESIDENT) S.r.copy = nullptr;
        S.seq.assign(R.seq, (size_t)R.seq_len); S.r.seq = S.seq.c_str();
        if (R.masked_seq) { S.masked.assign(R.masked_seq, (size_t)R.seq_len); S.r.masked_seq = S.masked.c_str(); }
        if (R.copy && R.copy != MIPGEN_COPY_RESIDENT) {
            S.copy.assign(MIPGEN_MAX_OLIGO + 1, std::vector<int32_t>());
            S.copy_ptrs.assign(MIPGEN_MAX_OLIGO + 1, nullptr);
            for (int l = 0; l <= MIPGEN_MAX_OLIGO; l++) if (R.copy[l]) { S.copy[(size_t)l].assign(R.copy[l], R.copy[l] + R.seq_len); S.copy_ptrs[(size_t)l] = S.copy[(size_t)l].data(); }
            S.r.copy = S.copy_ptrs.data();
        }
        if (R.unmappable) { S.unmap.assign(R.unmappable, R.unmappable + (size_t)n_sizes_all * (size_t)R.seq_len); S.r.unmappable = S.unmap.data(); }
        if (R.snp_class) { S.snp.assign(R.snp_class, R.snp_class + R.seq_len); S.r.snp_class = S.snp.data(); }
    }
    for (int i = 0; i < n; i++) {                            // (pointers into the vector's elements: taken after it stopped growing)
        RegionStore& S = h->regions[(size_t)i];
        S.r.seq = S.seq.c_str();
        if (!S.masked.empty() || regions[i].masked_seq) S.r.masked_seq = S.masked.c_str();
        if (!S.copy_ptrs.empty()) { for (int l = 0; l <= MIPGEN_MAX_OLIGO; l++) S.copy_ptrs[(size_t)l] = S.copy[(size_t)l].empty() ? nullptr : S.copy[(size_t)l].data(); S.r.copy = S.copy_ptrs.data(); }
        if (!S.unmap.empty()) S.r.unmappable = S.unmap.data();
        if (!S.snp.empty()) S.r.snp_class = S.snp.data();
        mipgen_grid g;
        memset(&g, 0, sizeof g);
        mo_grid(&P, &S.r, &g);
        g.offset = off; off += g.count;
        h->grids[(size_t)i] = g;
        h->region_pos0[(size_t)i + 1] = h->region_pos0[(size_t)i] + g.n_pos;
        const int nb = (g.n_pos > 0 && g.n_sizes > 0) ? g.n_pos + max_scan_of(P, g) - 1 : 0;
        h->region_base0[(size_t)i + 1] = h->region_base0[(size_t)i] + 2 * (int64_t)nb;
        if (grids_out) grids_out[i] = g;
    }
    h->windows.clear();
    for (int i = 0; i < n;) {
        Win w;
        w.r0 = i; w.cand0 = h->grids[(size_t)i].offset; w.pos0 = h->region_pos0[(size_t)i]; w.base0 = h->region_base0[(size_t)i];
        int64_t c = 0;
        int j = i;
        auto forced = [&](int k) { return std::find(h->window_breaks.begin(), h->window_breaks.end(), k) != h->window_breaks.end(); };
        while (j < n && (j == i || ((h->window_cap <= 0 || c + h->grids[(size_t)j].count <= h->window_cap) && !forced(j)))) { c += h->grids[(size_t)j].count; j++; }
        w.r1 = j; w.n_cand = c; w.n_pos = h->region_pos0[(size_t)j] - w.pos0; w.n_base = h->region_base0[(size_t)j] - w.base0;
        h->windows.push_back(w);
        i = j;
    }
    if (n == 0) { Win w; h->windows.push_back(w); }
    h->survivors.assign((size_t)(2 * h->region_pos0[(size_t)n]), mipgen_survivor{-1, 0.0, 0});
    h->surv_svr.assign(h->survivors.size(), std::numeric_limits<double>::quiet_NaN());
    h->collapsed.assign((size_t)h->region_base0[(size_t)n], -1);
    h->emitted_per_region.assign((size_t)n, 0);
    h->win_state.assign(h->windows.size(), 0);
    h->cur = -1; h->scored = h->replayed = false; h->have_text = false;
    return MIPGEN_OK;
}

int32_t mipgen_accel_window_count(const mipgen_accel* h) { return h ? (int32_t)h->windows.size() : 0; }
int mipgen_accel_window_info(const mipgen_accel* h, int32_t w, int32_t* fr, int32_t* nr, int64_t* fc, int64_t* nc, int64_t* fp, int64_t* np)
{
    if (!h || w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window out of range");
    const Win& W = h->windows[(size_t)w];
    if (fr) *fr = W.r0;
    if (nr) *nr = W.r1 - W.r0;
    if (fc) *fc = W.cand0;
    if (nc) *nc = W.n_cand;
    if (fp) *fp = W.pos0;
    if (np) *np = W.n_pos;
    return MIPGEN_OK;
}

int mipgen_accel_score_window(mipgen_accel* h, int32_t w, int32_t method)
{
    if (!h || w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window out of range");
    INJECT(h, "score_window");
    if (method == MIPGEN_SCORE_SVR && !h->model) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    const Win& W = h->windows[(size_t)w];
    if (fake_mode()) { h->cur = w; h->scored = true; h->replayed = false; h->have_text = false; h->win_state[(size_t)w] = 0; return MIPGEN_OK; }
    h->scores.assign((size_t)W.n_cand, 0.0); h->records.assign((size_t)W.n_cand, 0);
    for (int i = W.r0; i < W.r1; i++) {
        const mipgen_grid& g = h->grids[(size_t)i];
        if (g.count > 0) mo_score_region_dense(&h->P, &h->regions[(size_t)i].r, h->model, method, h->scores.data() + (g.offset - W.cand0), h->records.data() + (g.offset - W.cand0));
    }
    h->cur = w; h->scored = true; h->replayed = false; h->have_text = false;
    h->win_state[(size_t)w] = 0;
    return MIPGEN_OK;
}

int mipgen_accel_replay_condense(mipgen_accel* h);
int mipgen_accel_score_condense_window(mipgen_accel* h, int32_t w, int32_t method)      // ABI 6: what a silent front end calls per window
{
    if (!h || w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window out of range");
    INJECT(h, "score_condense_window");
    if (int rc = mipgen_accel_score_window(h, w, method)) return rc;
    return mipgen_accel_replay_condense(h);
}

int mipgen_accel_replay_condense(mipgen_accel* h)
{
    if (!h || !h->scored || h->cur < 0) return fail(MIPGEN_E_STATE, "replay requested before scoring");
    INJECT(h, "replay_condense");
    const Win& W = h->windows[(size_t)h->cur];
    if (fake_mode()) {
        // a fabricated survivor per scan position and strand: hashed capture size / arm pair / score (logistic-like in [0.3, 1), SVR-like in [0.8, 3)), copies 1
        const int A = h->P.n_arm_pairs;
        for (int i = W.r0; i < W.r1; i++) {
            const mipgen_grid& g = h->grids[(size_t)i];
            mipgen_survivor* sv = h->survivors.data() + 2 * h->region_pos0[(size_t)i];
            h->emitted_per_region[(size_t)i] = g.count;
            for (int pi = 0; pi < g.n_pos && g.n_sizes > 0; pi++)
                for (int s2 = 0; s2 < 2; s2++) {
                    const uint64_t hsh = mix64(((uint64_t)(uint32_t)i << 32) ^ ((uint64_t)pi << 1) ^ (uint64_t)s2 ^ 0x9e3779b97f4a7c15ULL);
                    const int ki = (int)(hsh % (uint64_t)g.n_sizes), a = (int)((hsh >> 20) % (uint64_t)A);
                    const double u = (double)((hsh >> 11) & 0xFFFFFFFFFFFFFull) / 4503599627370496.0;
                    mipgen_survivor& m = sv[2 * pi + s2];
                    m.cand_index = g.offset + (((int64_t)pi * g.n_sizes + ki) * 2 + s2) * A + a;
                    m.score = h->P.score_method == MIPGEN_SCORE_SVR ? 0.8 + 2.2 * u : 0.3 + 0.7 * u;
                    m.record = (uint64_t)1 | ((uint64_t)1 << 16) | ((uint64_t)MIPGEN_FLAG_VALID << 48) | ((uint64_t)(hsh >> 60) << 56);
                }
        }
        h->replayed = true; h->win_state[(size_t)h->cur] = 1;
        return MIPGEN_OK;
    }
    h->emitted.assign((size_t)W.n_cand, 0);
    for (int i = W.r0; i < W.r1; i++) {
        const mipgen_grid& g = h->grids[(size_t)i];
        const int64_t o = g.offset - W.cand0;
        mipgen_survivor* sv = h->survivors.data() + 2 * h->region_pos0[(size_t)i];
        h->emitted_per_region[(size_t)i] = 0;
        if (g.count <= 0) continue;
        h->emitted_per_region[(size_t)i] = mo_replay_region(&h->P, &h->regions[(size_t)i].r, h->scores.data() + o, h->records.data() + o, h->emitted.data() + o);
        mo_condense_region(&h->P, &h->regions[(size_t)i].r, h->scores.data() + o, h->records.data() + o, h->emitted.data() + o, sv);
        for (int64_t q = 0; q < 2 * (int64_t)g.n_pos; q++) if (sv[q].cand_index >= 0) sv[q].cand_index += g.offset;      // batch-wide, as the device hands them out
    }
    h->replayed = true;
    h->win_state[(size_t)h->cur] = 1;
    return MIPGEN_OK;
}

// collapse_mips (mipgen.cpp:1616-1649): survivors in scan-start order, '+' then '-'; a survivor is dropped by the copy / masked-arm filters (:1628-1629),
// else it takes every base of its scan target that is free, held by a survivor with more SNPs, or by one with the same SNPs and a lower score (:1634-1645)
int mipgen_accel_collapse(mipgen_accel* h)
{
    if (!h || !h->replayed || h->cur < 0) return fail(MIPGEN_E_STATE, "collapse requested before replay + condense");
    INJECT(h, "collapse");
    const mipgen_params& P = h->P;
    const Win& W = h->windows[(size_t)h->cur];
    for (int i = W.r0; i < W.r1; i++) {
        const mipgen_grid& g = h->grids[(size_t)i];
        const int64_t b0 = h->region_base0[(size_t)i], nb = (h->region_base0[(size_t)i + 1] - b0) / 2;
        int32_t* best = h->collapsed.data() + b0;
        std::fill(best, best + 2 * nb, -1);
        std::vector<int> snp_of((size_t)(2 * nb), 0);
        std::vector<double> score_of((size_t)(2 * nb), 0.0);
        const mipgen_survivor* sv = h->survivors.data() + 2 * h->region_pos0[(size_t)i];
        const int A = P.n_arm_pairs;
        for (int pi = 0; pi < g.n_pos; pi++)
            for (int s = 0; s < 2; s++) {
                const mipgen_survivor& m = sv[2 * pi + s];
                if (m.cand_index < 0) continue;
                const int64_t rel = m.cand_index - g.offset - (int64_t)pi * g.n_sizes * A * 2;
                const int a = (int)(rel % A), ki = (int)(rel / (2 * A));
                const int e = P.arm_ext[a], l = P.arm_lig[a];
                const int ss = P.max_capture_size - (g.first_size_index + ki) * P.capture_increment - e - l;
                const int64_t ec = (int64_t)MIPGEN_REC_EXT_COPY(m.record), lc = (int64_t)MIPGEN_REC_LIG_COPY(m.record);
                if (ec * lc > P.max_arm_copy_product || ec > P.target_arm_copy || lc > P.target_arm_copy) continue;
                if ((double)MIPGEN_REC_MASKED_N(m.record) / (double)(l + e) > P.masked_arm_threshold) continue;
                const int snp = (int)MIPGEN_REC_SNP_COUNT(m.record);
                for (int j = pi; j < pi + ss && j < nb; j++) {
                    const size_t k = (size_t)(2 * j + s);
                    if (best[k] < 0 || snp < snp_of[k] || (m.score > score_of[k] && snp == snp_of[k])) { best[k] = pi; snp_of[k] = snp; score_of[k] = m.score; }
                }
            }
    }
    h->win_state[(size_t)h->cur] |= 2;
    return MIPGEN_OK;
}

int mipgen_accel_region_bases(const mipgen_accel* h, int32_t region, int64_t* first_entry, int32_t* n_bases)
{
    if (!h || region < 0 || region >= (int32_t)h->grids.size()) return fail(MIPGEN_E_INVALID, "region out of range");
    if (first_entry) *first_entry = h->region_base0[(size_t)region];
    if (n_bases) *n_bases = (int32_t)((h->region_base0[(size_t)region + 1] - h->region_base0[(size_t)region]) / 2);
    return MIPGEN_OK;
}

int mipgen_accel_download_collapsed(mipgen_accel* h, int32_t window, int32_t* out, int64_t capacity)
{
    if (!h || !out || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    INJECT(h, "download_collapsed");
    if (!(h->win_state[(size_t)window] & 2)) return fail(MIPGEN_E_STATE, "collapse has not run on window %d", window);
    const Win& W = h->windows[(size_t)window];
    if (capacity < W.n_base) return fail(MIPGEN_E_INVALID, "collapsed capacity too small");
    if (W.n_base) memcpy(out, h->collapsed.data() + W.base0, (size_t)W.n_base * sizeof(int32_t));
    return MIPGEN_OK;
}

int mipgen_accel_download_results(mipgen_accel* h, double* scores, uint64_t* records, int64_t first, int64_t count)
{
    if (!h || !h->scored || h->cur < 0) return fail(MIPGEN_E_STATE, "nothing scored");
    const Win& W = h->windows[(size_t)h->cur];
    if (first < W.cand0 || first + count > W.cand0 + W.n_cand) return fail(MIPGEN_E_INVALID, "range outside the window scored last");
    if (scores) memcpy(scores, h->scores.data() + (first - W.cand0), (size_t)count * sizeof(double));
    if (records) memcpy(records, h->records.data() + (first - W.cand0), (size_t)count * sizeof(uint64_t));
    return MIPGEN_OK;
}

int mipgen_accel_download_replay(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors, int64_t survivor_capacity, uint8_t* mask, int64_t mask_capacity)
{
    if (!h || !h->replayed || h->cur < 0) return fail(MIPGEN_E_STATE, "replay + condense has not run");
    INJECT(h, "download_replay");
    const Win& W = h->windows[(size_t)h->cur];
    if (emitted_per_region) for (int i = W.r0; i < W.r1; i++) emitted_per_region[i - W.r0] = h->emitted_per_region[(size_t)i];
    if (survivors) {
        if (survivor_capacity < 2 * W.n_pos) return fail(MIPGEN_E_INVALID, "survivor capacity too small");
        if (W.n_pos) memcpy(survivors, h->survivors.data() + 2 * W.pos0, (size_t)(2 * W.n_pos) * sizeof(mipgen_survivor));
    }
    if (mask) {
        if (mask_capacity < W.n_cand) return fail(MIPGEN_E_INVALID, "mask capacity too small");
        if (W.n_cand) memcpy(mask, h->emitted.data(), (size_t)W.n_cand);
    }
    return MIPGEN_OK;
}

// print_details (mipgen.cpp:765-794) of every candidate the replay marked as constructed, in the reference's generation order: position, capture
// size, arm pair, plus then minus (mipgen.cpp:421-491)
int mipgen_accel_format_all_mips(mipgen_accel* h, const mipgen_record_names* names, const char* middle, int64_t first_index, int64_t* n_records, int64_t* n_bytes)
{
    if (!h || !middle || !n_records || !n_bytes) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (!h->replayed || h->cur < 0) return fail(MIPGEN_E_STATE, "format_all_mips needs replay + condense first");
    INJECT(h, "format_all_mips");
    const mipgen_params& P = h->P;
    const Win& W = h->windows[(size_t)h->cur];
    h->text.clear();
    int64_t idx = first_index;
    std::vector<char> buf(4 * MO_MAX_INSERT);
    for (int i = W.r0; i < W.r1; i++) {
        const mipgen_grid& g = h->grids[(size_t)i];
        const mipgen_record_names& nm = names[i - W.r0];
        const int64_t o = g.offset - W.cand0;
        const int A = P.n_arm_pairs;
        for (int64_t row = 0; row < (int64_t)g.n_pos * g.n_sizes; row++)
            for (int a = 0; a < A; a++)
                for (int s = 0; s < 2; s++) {
                    const int64_t k = (row * 2 + s) * A + a;
                    if (!h->emitted[(size_t)(o + k)]) continue;
                    const mipgen_candidate c = cand_of(P, g, i, k);
                    mo_designed d;
                    if (mo_design(&P, &h->regions[(size_t)i].r, &c, nullptr, &d)) return fail(MIPGEN_E_STATE, "stub accelerator: an emitted candidate fails the bounds skips");
                    idx++;
                    const int n = mo_print_details(nm.chr, nm.label, nm.feature_start + 1, nm.feature_stop, s, &d, h->scores[(size_t)(o + k)], middle, (int)idx, 0, buf.data(), (int)buf.size());
                    h->text.append(buf.data(), (size_t)n);
                }
    }
    *n_records = idx - first_index; *n_bytes = (int64_t)h->text.size();
    h->have_text = true;
    return MIPGEN_OK;
}

int mipgen_accel_download_text(mipgen_accel* h, char* dst, int64_t capacity)
{
    if (!h || !h->have_text) return fail(MIPGEN_E_STATE, "format_all_mips has not run on this window");
    INJECT(h, "download_text");
    if (capacity < (int64_t)h->text.size()) return fail(MIPGEN_E_INVALID, "text capacity too small");
    memcpy(dst, h->text.data(), h->text.size());
    return MIPGEN_OK;
}

int mipgen_accel_rescore_survivors(mipgen_accel* h)
{
    if (!h || !h->replayed || h->cur < 0) return fail(MIPGEN_E_STATE, "replay + condense has not run");
    INJECT(h, "rescore_survivors");
    if (!h->model) return fail(MIPGEN_E_MODEL, "SVR scoring requested but no model is loaded");
    const Win& W = h->windows[(size_t)h->cur];
    for (int i = W.r0; i < W.r1; i++) {
        const mipgen_grid& g = h->grids[(size_t)i];
        for (int64_t q = 2 * h->region_pos0[(size_t)i]; q < 2 * h->region_pos0[(size_t)i + 1]; q++) {
            const mipgen_survivor& sv = h->survivors[(size_t)q];
            h->surv_svr[(size_t)q] = std::numeric_limits<double>::quiet_NaN();
            if (sv.cand_index < 0) continue;
            const mipgen_candidate c = cand_of(h->P, g, i, sv.cand_index - g.offset);
            mo_designed d;
            if (mo_design(&h->P, &h->regions[(size_t)i].r, &c, nullptr, &d)) continue;
            h->surv_svr[(size_t)q] = mo_score_designed(&d, MIPGEN_SCORE_SVR, h->model, h->regions[(size_t)i].r.long_range_content, nullptr, nullptr);
        }
    }
    h->win_state[(size_t)h->cur] |= 4;
    return MIPGEN_OK;
}

int mipgen_accel_download_survivor_scores(mipgen_accel* h, int32_t window, double* svr, int64_t capacity)
{
    if (!h || !svr || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (!(h->win_state[(size_t)window] & 4)) return fail(MIPGEN_E_STATE, "rescore_survivors has not run on window %d", window);
    const Win& W = h->windows[(size_t)window];
    if (capacity < 2 * W.n_pos) return fail(MIPGEN_E_INVALID, "capacity too small");
    if (W.n_pos) memcpy(svr, h->surv_svr.data() + 2 * W.pos0, (size_t)(2 * W.n_pos) * sizeof(double));
    return MIPGEN_OK;
}

// the "device-side" views of a window the RCCL gather posts from (host memory here; tests/stub_accel/stub_rccl.cpp moves it on stream threads):
// the same fields as mipgen_amd/csrc/accel_score.hip hands out
int mipgen_accel_window_views(mipgen_accel* h, int32_t window, mipgen_window_views* out)
{
    if (!h || !out || window < 0 || window >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "bad arguments");
    INJECT(h, "window_views");
    const uint8_t st = h->win_state[(size_t)window];
    if (!(st & 1)) return fail(MIPGEN_E_STATE, "window %d holds no current survivors (replay + condense first)", window);
    const Win& W = h->windows[(size_t)window];
    memset(out, 0, sizeof *out);
    out->emitted = h->emitted_per_region.data() + W.r0; out->n_emitted = W.r1 - W.r0;
    out->survivors = h->survivors.data() + 2 * W.pos0; out->n_survivors = 2 * W.n_pos;
    if (st & 2) { out->collapsed = h->collapsed.data() + W.base0; out->n_collapsed = W.n_base; }
    if (st & 4) out->survivor_svr = h->surv_svr.data() + 2 * W.pos0;
    if (window == h->cur && h->have_text && !h->text.empty()) { out->text = h->text.data(); out->n_text_bytes = (int64_t)h->text.size(); }
    out->first_candidate = W.cand0;
    return MIPGEN_OK;
}
int mipgen_accel_synchronize(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    INJECT(h, "synchronize");
    return MIPGEN_OK;
}
// the GPU k-mer counter: not in the stub (those paths need a GPU)
int mipgen_accel_count_oligo_copies(mipgen_accel*, int32_t, const char* const*, const int64_t*, int32_t, const char* const*, const int32_t*, int32_t, const int32_t*, int32_t* const*)
{ return fail(MIPGEN_E_NODEVICE, "stub accelerator: no k-mer counter"); }
int mipgen_accel_count_oligo_copies_resident(mipgen_accel*, int32_t, const char* const*, const int64_t*, int32_t, const char* const*, const int32_t*, int64_t* n_big, const mipgen_big_copy** big)
{
    if (!fake_mode()) return fail(MIPGEN_E_NODEVICE, "stub accelerator: no k-mer counter");
    if (n_big) *n_big = 0;
    if (big) *big = nullptr;
    return MIPGEN_OK;
}
int mipgen_accel_window_uniqueness(mipgen_accel*, int32_t, const char* const*, const int64_t*, int32_t, const char* const*, const int32_t*, int32_t, const int32_t*, int32_t, uint8_t* const*)
{ return fail(MIPGEN_E_NODEVICE, "stub accelerator: no k-mer counter"); }
int mipgen_accel_window_uniqueness_begin(mipgen_accel*, int32_t, const char* const*, const int64_t*, int32_t n_regions, const char* const*, const int32_t*, const mipgen_window_bounds*, int32_t,
                                         const int32_t*, int32_t, uint8_t* any)
{
    if (!fake_mode()) return fail(MIPGEN_E_NODEVICE, "stub accelerator: no k-mer counter");
    if (any) memset(any, 0, (size_t)n_regions);
    return MIPGEN_OK;
}
int mipgen_accel_window_flags_region(mipgen_accel*, int32_t, uint8_t*) { return fail(MIPGEN_E_NODEVICE, "stub accelerator: no k-mer counter"); }
int mipgen_accel_window_uniqueness_end(mipgen_accel*) { return MIPGEN_OK; }

}  // extern "C"
