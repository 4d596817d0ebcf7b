// select.cpp — selection stage and record formatting of the drop-in front end.
// Semantics follow /root/reference/mipgen.cpp: collapse_mips :1616-1649, output_collapsed_mips :1651-1668,
// pick_mips :1506-1614, optimize_worst_in_region :1748-1820, translocate_down_region :1822-1908,
// manage_picked_mip :1910-1939, print_gaps / create_gap :1231-1278, print_details :765-794, headers :349-399.
// The per-(scan start, strand) survivors it starts from (condense_mips, :1670-1746) come from the accelerator.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>

#include "mipgen_host.hpp"
#include <chrono>

namespace mipgen {

// ---- candidates and records --------------------------------------------------------------------------------

Cand make_cand(const Options& o, const Region& r, const mipgen_grid& g, int64_t li, double score, uint64_t rec)
{
    // strand-major dense order: li = (((pi * n_sizes) + ki) * 2 + strand) * A + a   (include/mipgen_accel.h)
    const int64_t per_pos = (int64_t)g.n_sizes * 2 * (int64_t)o.arm_pairs.size();
    const int pi = (int)(li / per_pos);
    return make_cand_at(o, r, g, pi, (uint32_t)(li - (int64_t)pi * per_pos), score, rec);
}

// the same for a caller that knows the scan position: `within` = (ki * 2 + strand) * A + a, the candidate's index inside its position.  (The
// selection stage builds 10^8 of these per exome design: one 32-bit division instead of four 64-bit ones.)
Cand make_cand_at(const Options& o, const Region& r, const mipgen_grid& g, int pi, uint32_t within, double score, uint64_t rec)
{
    Cand c;
    const uint32_t A = (uint32_t)o.arm_pairs.size();
    const uint32_t row = within / A;
    const int a = (int)(within - row * A);
    c.strand = (int)(row & 1u);
    const int ki = (int)(row >> 1);
    c.ext_len = o.arm_pairs[(size_t)a].first; c.lig_len = o.arm_pairs[(size_t)a].second;
    c.capture = o.max_capture - (g.first_size_index + ki) * o.capture_increment;
    c.scan_start = g.first_pos + pi;
    c.scan_stop = c.scan_start + c.capture - c.ext_len - c.lig_len - 1;            // mipgen.cpp:449
    if (c.strand == 0) {                                                           // PlusSVMipv4.cpp:9-12
        c.ext_start = c.scan_start - c.ext_len; c.ext_stop = c.scan_start - 1;
        c.lig_start = c.scan_stop + 1; c.lig_stop = c.scan_stop + c.lig_len;
    } else {                                                                       // MinusSVMipv4.cpp:32-35
        c.ext_start = c.scan_stop + 1; c.ext_stop = c.scan_stop + c.ext_len;
        c.lig_start = c.scan_start - c.lig_len; c.lig_stop = c.scan_start - 1;
    }
    c.ext_copy = (int)MIPGEN_REC_EXT_COPY(rec); c.lig_copy = (int)MIPGEN_REC_LIG_COPY(rec);
    // the record's 16-bit copy fields saturate; bwa's X0 count does not (mipgen.cpp:586-587): a saturated field is read from the
    // region's own copy table, so printed and compared copies are the reference's
    auto true_copy = [&](int start, int len) -> int {
        if (r.copy_resident) {                                                     // the tables stayed in HBM: the host holds exactly the counts >= 65535
            auto it = r.big_copy.find({len, (int)((long)start - r.seq_start)});
            return it != r.big_copy.end() ? it->second : 0;
        }
        if ((size_t)len >= r.copy_ptr.size() || !r.copy_ptr[(size_t)len]) return 65535;
        const long rel = (long)start - r.seq_start;
        return rel >= 0 && rel < (long)r.seq.size() ? r.copy_ptr[(size_t)len][rel] : 0;
    };
    if (c.ext_copy == 65535) c.ext_copy = true_copy(c.ext_start, c.ext_len);
    if (c.lig_copy == 65535) c.lig_copy = true_copy(c.lig_start, c.lig_len);
    c.snp_count = (int)MIPGEN_REC_SNP_COUNT(rec);
    c.masked = (double)MIPGEN_REC_MASKED_N(rec) / (double)(c.lig_len + c.ext_len);  // mipgen.cpp:610
    c.score = score;
    const uint32_t f = MIPGEN_REC_FLAGS(rec);
    c.mapping_failed = (f & MIPGEN_FLAG_MAPPING) ? '1' : '0';
    c.snp_failed = (f & MIPGEN_FLAG_SNP) ? '1' : '0';
    c.masking_failed = (f & MIPGEN_FLAG_MASKING) ? '1' : '0';
    return c;
}

static std::string revcomp(const std::string& s)                                   // MinusSVMipv4.cpp:6-29
{
    std::string o;
    o.reserve(s.size());
    for (size_t i = s.size(); i-- > 0;) {
        char c = s[i];
        switch (c) { case 'G': c = 'C'; break; case 'C': c = 'G'; break; case 'A': c = 'T'; break; case 'T': c = 'A'; break; default: break; }
        o += c;
    }
    return o;
}

static std::string slice(const Region& r, int start, int len)
{
    const long rel = (long)start - r.seq_start;
    if (rel < 0 || rel > (long)r.seq.size()) return std::string();
    return r.seq.substr((size_t)rel, (size_t)len);
}

static char comp(char c)
{
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; default: return 0; }
}

// alternate-allele arms of a candidate with exactly one usable SNP (design_mip, mipgen.cpp:634-758)
static void snp_arms(const Region& r, const Tables& t, const Cand& c, std::string& ext, std::string& lig)
{
    auto sit = t.snps.find(r.chr);
    if (sit == t.snps.end()) return;
    for (int arm = 0; arm < 2; arm++) {
        const int a0 = arm == 0 ? c.ext_start : c.lig_start, a1 = arm == 0 ? c.ext_stop : c.lig_stop;
        std::string& seq = arm == 0 ? ext : lig;
        for (int i = a0; i <= a1; i++) {
            auto it = sit->second.find(i);
            if (it == sit->second.end()) continue;
            const std::string& al = it->second;
            if (!(al.size() == 2 && al[0] != 'N' && al[1] != 'N' && al[0] != '-' && al[1] != '-')) continue;
            const int rel = c.strand == 0 ? i - a0 : a1 - i;
            if (rel < 0 || rel >= (int)seq.size()) continue;
            if (seq[(size_t)rel] == al[0]) seq[(size_t)rel] = al[1];
            else if (comp(al[0]) && seq[(size_t)rel] == comp(al[0]) && comp(al[1])) seq[(size_t)rel] = comp(al[1]);
        }
    }
}

// print_details, mipgen.cpp:765-794.  One pass into a string (no stream objects: 10^5..10^6 picked / collapsed records per exome design).
static inline void put_int(std::string& s, long long v)
{
    char b[24];
    int n = 0;
    const bool neg = v < 0;
    unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) s += '-';
    while (n) s += b[--n];
}

std::string format_record(const Options& o, const Region& r, const Tables& t, const Cand& c, int64_t index, bool minor)
{
    std::string ext = slice(r, c.ext_start, c.ext_len), lig = slice(r, c.lig_start, c.lig_len), ins = slice(r, c.scan_start, c.scan_size());
    if (c.strand == 1) { ext = revcomp(ext); lig = revcomp(lig); ins = revcomp(ins); }
    if (minor) snp_arms(r, t, c, ext, lig);                                        // the SNP_b record carries the alternate arms (:1916-1919)
    char sc[64];
    if (std::isnan(c.score)) snprintf(sc, sizeof sc, "-nan");                      // x86 default NaN of the reference's inf-inf prints as "-nan"
    else snprintf(sc, sizeof sc, "%g", c.score);                                   // default ostream formatting: 6 significant digits
    const char st = c.strand == 0 ? '+' : '-';
    std::string ss;
    ss.reserve(2 * r.chr.size() + 2 * (ext.size() + lig.size()) + ins.size() + o.middle.size() + r.label.size() + 192);
    ss += r.chr; ss += ':'; put_int(ss, c.strand == 0 ? c.ext_start : c.lig_start); ss += '-'; put_int(ss, c.strand == 0 ? c.lig_stop : c.ext_stop); ss += '/';
    put_int(ss, c.ext_len); ss += ','; put_int(ss, c.lig_len); ss += '/'; ss += st; ss += '\t'; ss += sc; ss += '\t'; ss += r.chr; ss += '\t';
    put_int(ss, c.ext_start); ss += '\t'; put_int(ss, c.ext_stop); ss += '\t'; put_int(ss, c.ext_copy); ss += '\t'; ss += ext; ss += '\t';
    put_int(ss, c.lig_start); ss += '\t'; put_int(ss, c.lig_stop); ss += '\t'; put_int(ss, c.lig_copy); ss += '\t'; ss += lig; ss += '\t';
    put_int(ss, c.scan_start); ss += '\t'; put_int(ss, c.scan_stop); ss += '\t'; ss += ins; ss += '\t'; ss += lig; ss += o.middle; ss += ext; ss += '\t';
    put_int(ss, r.start - 1); ss += '\t'; put_int(ss, r.stop); ss += '\t'; ss += st; ss += '\t';
    ss += c.mapping_failed; ss += c.snp_failed; ss += c.masking_failed; ss += '\t'; ss += r.label; ss += '_';
    char num[32];
    snprintf(num, sizeof num, "%04lld", (long long)index);
    ss += num;
    if (c.snp_count == 1) { ss += "_SNP_"; ss += minor ? 'b' : 'a'; }
    ss += '\n';
    return ss;
}

static const char* k_cols =
    "_score\tchr\text_probe_start\text_probe_stop\text_probe_copy\text_probe_sequence\tlig_probe_start\tlig_probe_stop\tlig_probe_copy\t"
    "lig_probe_sequence\tmip_scan_start_position\tmip_scan_stop_position\tscan_target_sequence\tmip_sequence\tfeature_start_position\t"
    "feature_stop_position\tprobe_strand\tfailure_flags\tmip_name\n";

// file headers, mipgen.cpp:349-399 (all/collapsed say svr only for svr; picked/snp say logistic only for logistic)
void open_outputs(const Options& o, Outputs& out)
{
    const std::string pn = o.project_name;
    out.all.open(pn + ".all_mips.txt");
    if (!out.all.is_open()) { std::cerr << "[mipgen] all mips file could not be opened" << std::endl; throw 12; }
    out.all << ">mip_key\t" << (o.score_method == MIPGEN_SCORE_SVR ? "svr" : "logistic") << k_cols;
    out.collapsed.open(pn + ".collapsed_mips.txt");
    if (!out.collapsed.is_open()) { std::cerr << "[mipgen] file of collapsed mips could not be opened" << std::endl; throw 13; }
    out.collapsed << ">mip_key\t" << (o.score_method == MIPGEN_SCORE_SVR ? "svr" : "logistic") << k_cols;
    out.picked.open(pn + ".picked_mips.txt");
    if (!out.picked.is_open()) throw 14;
    out.picked << ">mip_key\t" << (o.score_method == MIPGEN_SCORE_LOGISTIC ? "logistic" : "svr") << k_cols;
    out.snp.open(pn + ".snp_mips.txt");
    if (!out.snp.is_open()) throw 15;
    out.snp << ">mip_key\t" << (o.score_method == MIPGEN_SCORE_LOGISTIC ? "logistic" : "svr") << k_cols;
}

// ---- the reference's random stream -------------------------------------------------------------------------
// glibc random_r.c, TYPE_3 (degree 31, separation 3), as initialised by srandom(1): the state a process that never calls
// srand() starts from.  r[i] = r[i-31] + r[i-3] (mod 2^32); rand() returns r >> 1.
GlibcRand::GlibcRand()
{
    int32_t st[34];
    st[0] = 1;
    for (int i = 1; i < 31; i++) {                       // 16807 * x mod (2^31 - 1), Schrage's method as glibc writes it
        const long hi = st[i - 1] / 127773, lo = st[i - 1] % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        st[i] = (int32_t)w;
    }
    for (int i = 0; i < 31; i++) r_[i] = (uint32_t)st[i];
    f_ = 3; b_ = 0;
    for (int i = 0; i < 310; i++) (void)next();          // glibc discards the first 10 * degree outputs
}

int GlibcRand::next()
{
    r_[f_] += r_[b_];
    const int out = (int)(r_[f_] >> 1);
    if (++f_ >= 31) f_ = 0;
    if (++b_ >= 31) b_ = 0;
    return out;
}

// ---- selection ---------------------------------------------------------------------------------------------

Selector::Selector(const Options& o, const Tables& t, Outputs& out) : o_(o), t_(t), out_(out)
{
    // index inside a scan position -> (capture size index, strand, arm lengths): within = (size index * 2 + strand) * pairs + pair
    const size_t A = o.arm_pairs.size();
    const int K = o.capture_increment > 0 ? (o.max_capture - o.min_capture) / o.capture_increment + 1 : 1;
    within_.resize((size_t)std::max(K, 1) * 2 * A);
    for (size_t w = 0; w < within_.size(); w++) {
        const size_t row = w / A, a = w - row * A;
        within_[w] = Within{(uint16_t)(row >> 1), (uint8_t)o.arm_pairs[a].first, (uint8_t)o.arm_pairs[a].second, (uint8_t)(row & 1)};
    }
}

// the record's 16-bit copy fields saturate; bwa's X0 count does not (mipgen.cpp:586-587): a saturated field is read from the region's own
// copy table (or the handle's list of counts >= 65535), so printed and compared copies are the reference's
int Selector::true_copy(int start, int len) const
{
    const Region& r = *r_;
    if (r.copy_resident) {
        auto it = r.big_copy.find({len, (int)((long)start - r.seq_start)});
        return it != r.big_copy.end() ? it->second : 0;
    }
    if ((size_t)len >= r.copy_ptr.size() || !r.copy_ptr[(size_t)len]) return 65535;
    const long rel = (long)start - r.seq_start;
    return rel >= 0 && rel < (long)r.seq.size() ? r.copy_ptr[(size_t)len][rel] : 0;
}

Cand Selector::cand_of(CandPtr m) const
{
    const int pi = slot_of(m) >> 1;
    return make_cand_at(o_, *r_, *g_, pi, (uint32_t)(m->cand_index - base_ - (int64_t)pi * per_pos_), m->score, m->record);
}

double Selector::rescore(CandPtr m)                          // test_mip->score = predict_value(...), in place (mipgen.cpp:1523-1527, 1873-1877)
{
    const double s = rs_->svr(cand_of(m));
    sv_copy_[(size_t)slot_of(m)].score = s;                  // mixed designs run on the selector's own copy of the survivors
    return s;
}

bool Selector::arm_used(const Geo& x, int strand) const
{
    const UsedBases& u = (*used_cur_)[(size_t)strand];
    return u.any(x.ext_start, x.ext_stop) || u.any(x.lig_start, x.lig_stop);
}

void Selector::run_region(const Region& r, const mipgen_grid& g, const mipgen_survivor* surv, int64_t index_base, Rescorer* rs,
                          double lower, double upper, const int32_t* collapsed, int32_t n_bases)
{
    r_ = &r; g_ = &g; rs_ = rs; lower_ = lower; upper_ = upper;
    used_cur_ = &used_[r.chr];
    const auto t0 = std::chrono::steady_clock::now();
    first_pos_ = g.first_pos; n_pos_ = std::max(g.n_pos, 0);
    base_ = index_base; per_pos_ = (int64_t)g.n_sizes * 2 * (int64_t)o_.arm_pairs.size();
    cap0_ = o_.max_capture - g.first_size_index * o_.capture_increment;
    if ((size_t)per_pos_ > within_.size()) throw 21;
    if (rs) { sv_copy_.assign(surv, surv + 2 * (size_t)n_pos_); sv_ = sv_copy_.data(); }     // re-scored in place
    else sv_ = surv;
    touched_.assign((size_t)n_pos_, 0);
    touched_outside_.clear();
    if (collapsed) { col_ = collapsed; n_bases_ = n_bases; }                                  // collapse_mips ran on the accelerator
    else collapse();
    const auto t1 = std::chrono::steady_clock::now();
    if (!o_.silent) output_collapsed();
    out_.progress << "mips collapsed! picking mips...\n";
    if (o_.score_method == MIPGEN_SCORE_MIXED) { lower_ = o_.svr_priority; upper_ = o_.svr_optimal; }     // mipgen.cpp:510-514
    const auto t2 = std::chrono::steady_clock::now();
    pick();
    const auto t3 = std::chrono::steady_clock::now();
    sv_ = nullptr; col_ = nullptr; n_bases_ = 0;
    const auto t4 = std::chrono::steady_clock::now();
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    stage_seconds[0] += sec(t0, t1); stage_seconds[1] += sec(t1, t2); stage_seconds[2] += sec(t2, t3); stage_seconds[3] += sec(t3, t4);
}

// collapse_mips, mipgen.cpp:1616-1649 (only when the accelerator's collapse result is not handed in)
void Selector::collapse()
{
    const int max_scan = std::max(o_.max_capture - o_.min_arm_sum, 1);
    n_bases_ = n_pos_ + max_scan;                            // the furthest scan target ends max_scan - 1 bases behind the last scan start
    col_own_.assign(2 * (size_t)n_bases_, -1);
    for (int pi = 0; pi < n_pos_; pi++) {
        for (int strand = 0; strand < 2; strand++) {
            const CandPtr m = sv_[2 * pi + strand].cand_index >= 0 ? sv_ + 2 * pi + strand : nullptr;
            if (!m) continue;
            const Geo x = geo(m);
            const int ec = ext_copy(m, x), lc = lig_copy(m, x);
            if ((long)ec * lc > o_.max_arm_copy || ec > o_.target_arm_copy || lc > o_.target_arm_copy) continue;
            if (masked(m, x) > o_.masked_arm_threshold) continue;
            for (int pos = x.scan_start; pos <= x.scan_stop; pos++) {
                const long j = (long)pos - first_pos_;
                if (j < 0 || j >= n_bases_) throw 21;
                int32_t& cur = col_own_[2 * (size_t)j + strand];
                if (cur < 0) { cur = pi; continue; }
                const CandPtr c = sv_ + 2 * (long)cur + strand;
                if (snp_count(m) < snp_count(c)) cur = pi;
                else if (m->score > c->score && snp_count(m) == snp_count(c)) cur = pi;
            }
        }
    }
    col_ = col_own_.data();
}

// output_collapsed_mips, mipgen.cpp:1651-1668
void Selector::output_collapsed()
{
    for (int32_t j = 0; j < n_bases_; j++)
        for (int strand = 0; strand < 2; strand++) {
            const CandPtr m = pos_m(first_pos_ + j, strand);
            if (!m) continue;
            if (m->cand_index < 0) throw 21;
            out_.collapsed_counter++;
            out_.collapsed << format_record(o_, *r_, t_, cand_of(m), out_.collapsed_counter, false);
        }
}

// optimize_worst_in_region, mipgen.cpp:1748-1820
Selector::CandPtr Selector::optimize_worst(PosSet& positions, int strand_to_use)
{
    CandPtr worst = nullptr;
    // a survivor is the best of ~100 consecutive bases: its arm test is repeated for every one of them by the reference, with the same
    // answer - the used-arm sets do not change inside this function - so the last answer per strand is kept
    CandPtr last[2] = {nullptr, nullptr};
    bool last_free[2] = {false, false};
    auto arm_free = [&](CandPtr m, int strand) -> bool {
        if (m != last[strand]) {
            if (m->cand_index < 0) throw 21;
            last[strand] = m; last_free[strand] = !arm_used(geo(m), strand);
        }
        return last_free[strand];
    };
    // consecutive bases are mostly covered by the same pair of survivors (a survivor is the best of ~100 bases): a base whose pair is the previous
    // base's changes nothing - its candidate was compared already and `<` is strict - so it costs two integer compares, not the body below
    int32_t last_c0 = -2, last_c1 = -2;
    positions.for_each([&](int pos) {
        const long j = (long)pos - first_pos_;
        if (j >= 0 && j < n_bases_) {
            const int32_t c0 = col_[2 * j], c1 = col_[2 * j + 1];
            if (c0 == last_c0 && c1 == last_c1) return;
            last_c0 = c0; last_c1 = c1;
        } else { last_c0 = -2; last_c1 = -2; }
        const CandPtr m0 = pos_m(pos, 0), m1 = pos_m(pos, 1);
        if (!m0 && !m1) return;
        CandPtr cur = nullptr, plus = nullptr, minus = nullptr;
        bool plus_set = false, minus_set = false;
        if (m0 && strand_to_use != 1) {
            plus = m0;
            plus_set = arm_free(plus, 0);
            if (plus_set) cur = plus;
        }
        if (m1 && strand_to_use != 0) {
            minus = m1;
            minus_set = arm_free(minus, 1);
            if (minus_set) cur = plus_set ? (plus->score > minus->score ? plus : minus) : minus;
        }
        if ((plus_set || minus_set) && (!worst || cur->score < worst->score)) worst = cur;
    });
    return worst;
}

// translocate_down_region, mipgen.cpp:1822-1908
Selector::CandPtr Selector::translocate(PosSet& positions, int strand_to_use)
{
    if (positions.empty()) return nullptr;                       // the reference dereferences begin() of an empty set here (UB); see DESIGN.md
    const int latest = positions.front();
    const int to_end = r_->stop_fl - latest;
    const int min_scan = o_.min_capture - o_.max_arm_sum;
    int earliest, prelim, dir;
    if (to_end < min_scan - 10 - o_.starting_mip_overlap) {
        earliest = r_->stop_fl - min_scan + 1; prelim = earliest; dir = 1;
        while (!scan_exists(prelim) && prelim <= latest) prelim++;
    } else {
        earliest = latest - o_.max_mip_overlap; prelim = latest - o_.starting_mip_overlap; dir = -1;
        while (!scan_exists(prelim) && prelim >= earliest) prelim--;
    }
    if (!scan_exists(prelim)) return nullptr;
    CandPtr next = nullptr;
    Geo next_x{};
    int next_snp = 0;
    int prev_extent = latest + 1;
    for (int chosen = prelim;
         (!next && prev_extent > latest && chosen < r_->stop_fl && chosen > r_->start_fl - o_.min_capture) ||
         (next && ((next->score < upper_ || next_snp > 0) && chosen >= earliest && chosen <= latest - o_.starting_mip_overlap &&
                   (dir == -1 || chosen + (next_x.scan_stop - next_x.scan_start + 1) > positions.back())));
         chosen += dir) {
        int strand_index, iterations;
        if (strand_to_use != -1) { strand_index = 1 - strand_to_use; iterations = 1; }
        else { strand_index = rand_.next() % 2; iterations = 2; }                 // the reference's libc rand(), never seeded (:1863)
        for (int i = 0; i < iterations; i++) {
            strand_index = 1 - strand_index;
            scan_touch(chosen);                                                   // operator[]: creates the position, as the reference does (:1869)
            const CandPtr test = scan_m(chosen, strand_index);
            if (!test) continue;
            if (o_.score_method == MIPGEN_SCORE_MIXED && rs_) rescore(test);      // in place (:1873-1877)
            const Geo x = geo(test);
            prev_extent = x.scan_stop;
            const int test_snp = snp_count(test);
            if (!next || test->score > next->score || test_snp < next_snp) {
                const int test_copy = std::max(ext_copy(test, x), lig_copy(test, x));
                if (test_copy > o_.target_arm_copy && next) {
                    if (test_copy > std::max(ext_copy(next, next_x), lig_copy(next, next_x))) continue;
                }
                const double test_masked = masked(test, x);
                if (test_masked > o_.masked_arm_threshold && next) {
                    if (test_masked > masked(next, next_x)) continue;
                }
                if (arm_used(x, strand_index)) continue;
                next = test; next_x = x; next_snp = test_snp;
            }
        }
    }
    return next;
}

// manage_picked_mip, mipgen.cpp:1910-1939
void Selector::manage_picked(CandPtr mp, PosSet& positions)
{
    const Cand c = cand_of(mp);
    const Cand* m = &c;
    out_.picked_counter++;
    // print_details of the picked MIP (+ its SNP record / the note of :1920-1923) on the writer thread, in pick order
    wlocal_.push_back(PickJob{r_, c, out_.picked_counter});
    if (wlocal_.size() >= 256) hand_over();
    const int other = 1 - m->strand;
    auto& used = *used_cur_;
    if (o_.seal_both) {
        used[other].insert_range(m->ext_start, m->ext_stop);
        used[other].insert_range(m->lig_start, m->lig_stop);
    } else if (o_.half_seal_both) {
        used[other].insert((m->ext_stop + m->ext_start) / 2);
        used[other].insert((m->lig_start + m->lig_stop) / 2);
    }
    used[m->strand].insert_range(m->ext_start, m->ext_stop);
    used[m->strand].insert_range(m->lig_start, m->lig_stop);
    positions.erase_range(m->scan_start, m->scan_stop);
}

void Selector::hand_over()
{
    if (wlocal_.empty()) return;
    std::unique_lock<std::mutex> lk(wm_);
    if (!wstarted_) { wstarted_ = true; writer_ = std::thread([this] { writer_loop(); }); }
    wcv_.wait(lk, [&] { return wq_.size() < 65536 || werr_; });
    if (werr_) { std::exception_ptr e = werr_; lk.unlock(); wlocal_.clear(); std::rethrow_exception(e); }
    const bool was_empty = wq_.empty();
    wq_.insert(wq_.end(), wlocal_.begin(), wlocal_.end());
    wlocal_.clear();
    if (was_empty) wcv_.notify_all();                                // (the writer only sleeps on an empty queue)
}

void Selector::writer_loop()
{
    try {
        for (;;) {
            std::deque<PickJob> batch;
            {
                std::unique_lock<std::mutex> lk(wm_);
                wcv_.wait(lk, [&] { return wstop_ || !wq_.empty(); });
                if (wq_.empty()) return;
                const bool was_full = wq_.size() >= 65536;
                batch.swap(wq_);
                if (was_full) wcv_.notify_all();                     // (the selection thread only sleeps on a full queue)
            }
            for (const PickJob& j : batch) {
                const Cand* m = &j.c;
                out_.picked << format_record(o_, *j.r, t_, *m, j.counter, false);
                if (m->snp_count == 1 && m->snp_failed == '0') out_.snp << format_record(o_, *j.r, t_, *m, j.counter, true);
                else if (m->snp_failed == '1') out_.snp << ">Alternate MIP(s) could not be generated for SNP in arms of MIP #" << j.counter << std::endl;
            }
        }
    } catch (...) {
        std::lock_guard<std::mutex> lk(wm_);
        werr_ = std::current_exception();
        wq_.clear();
        wcv_.notify_all();
    }
}

void Selector::finish()
{
    hand_over();
    {
        std::lock_guard<std::mutex> lk(wm_);
        if (!wstarted_) return;
        wstop_ = true;
        wcv_.notify_all();
    }
    if (writer_.joinable()) writer_.join();
    std::lock_guard<std::mutex> lk(wm_);
    wstarted_ = false; wstop_ = false;
    if (werr_) { std::exception_ptr e = werr_; werr_ = nullptr; std::rethrow_exception(e); }
}

Selector::~Selector()
{
    try { finish(); } catch (...) {}
}

// print_gaps, mipgen.cpp:1231-1259
void Selector::print_gaps(std::ofstream& f, const std::string& ext, const std::string& note, PosSet& positions)
{
    if (positions.empty()) return;
    if (!f.is_open()) f.open(o_.arg("-project_name") + ext);
    out_.progress << note << r_->chr << ":\n";
    int start = positions.front(), stop = start - 1;
    positions.for_each([&](int p) {
        if (p == stop + 1) stop++;
        else {
            out_.bad_design_count++;
            f << r_->chr << "\t" << start - 1 << "\t" << stop << std::endl;
            start = p; stop = p;
        }
    });
    out_.bad_design_count++;
    f << r_->chr << "\t" << start - 1 << "\t" << stop << std::endl;
}

// create_gap, mipgen.cpp:1261-1278
void Selector::create_gap(std::ofstream& f, const std::string& ext, const std::string& note, PosSet& positions)
{
    out_.bad_design_count++;
    if (!f.is_open()) f.open(o_.arg("-project_name") + ext);
    out_.progress << note << r_->chr << ":\n";
    const int start = positions.front(), stop = start + o_.max_capture / 2;
    out_.progress << r_->chr << "\t" << start - 1 << "\t" << stop << std::endl;
    for (int i = start; i <= stop; i++) positions.erase(i);
    f << r_->chr << "\t" << start - 1 << "\t" << stop << std::endl;
}

// pick_mips, mipgen.cpp:1506-1614
void Selector::pick()
{
    using clk = std::chrono::steady_clock;
    clk::time_point tq = fine_timing ? clk::now() : clk::time_point();
    auto lap = [&](int k) { if (fine_timing) { const auto n = clk::now(); pick_seconds[k] += std::chrono::duration<double>(n - tq).count(); tq = n; } };
    auto OW = [&](PosSet& ps, int st) { lap(0); CandPtr m = optimize_worst(ps, st); lap(1); return m; };
    auto TL = [&](PosSet& ps, int st) { lap(0); CandPtr m = translocate(ps, st); lap(2); return m; };
    auto MP = [&](CandPtr m, PosSet& ps) { lap(0); manage_picked(m, ps); lap(3); };
    PosSet pos, again, minus, minus_again;
    const int strand_to_use = o_.double_tile_strands_separately ? 0 : -1;
    pos.fill(r_->start_fl, r_->stop_fl);
    if (o_.double_tile) again.fill(r_->start_fl, r_->stop_fl);
    if (o_.double_tile_strands_separately) minus.fill(r_->start_fl, r_->stop_fl);
    if (o_.double_tile && o_.double_tile_strands_separately) minus_again.fill(r_->start_fl, r_->stop_fl);
    const bool mixed = o_.score_method == MIPGEN_SCORE_MIXED && rs_;
    CandPtr picked = OW(pos, strand_to_use);
    if (mixed && picked) rescore(picked);
    while (!pos.empty() && picked && picked->score < lower_) {
        MP(picked, pos);
        picked = OW(pos, strand_to_use);
        if (mixed && picked) rescore(picked);
    }
    if (o_.double_tile_strands_separately) {
        picked = OW(minus, 1);                                         // not re-scored here in the reference (:1541)
        while (!minus.empty() && picked && picked->score < lower_) {
            MP(picked, minus);
            picked = OW(minus, 1);
            if (mixed && picked) rescore(picked);
        }
    }
    bool extended;
    if (!pos.empty()) {
        do {
            picked = TL(pos, strand_to_use);
            extended = pos.back() - pos.front() > o_.max_capture;
            if (!picked && extended) create_gap(out_.gaps, ".coverage_failed.bed", "GAP INTRODUCED ON CHROMOSOME ", pos);
            if (picked) MP(picked, pos);
        } while (!pos.empty() && (picked || extended));
    }
    if (!minus.empty()) {
        do {
            picked = TL(minus, 1);
            extended = minus.back() - minus.front() > o_.max_capture;
            if (!picked && extended) create_gap(out_.minus_gaps, ".minus_strand_failed.bed", "GAP INTRODUCED ON MINUS STRAND OF CHROMOSOME ", minus);
            if (picked) MP(picked, minus);
        } while (!minus.empty() && (picked || extended));
    }
    if (o_.double_tile) {
        do {
            picked = TL(again, strand_to_use);
            extended = !again.empty() && again.back() - again.front() > o_.max_capture;
            if (!picked && extended) create_gap(out_.double_gaps, ".double_tile_failed.bed", "GAP INTRODUCED ON DOUBLE TILING OF CHROMOSOME ", again);
            if (picked) MP(picked, again);
        } while (!again.empty() && (picked || extended));
        if (o_.double_tile_strands_separately) {
            // the reference walks positions_to_scan_again here but manages / terminates on positions_to_scan_minus_again (:1597-1607)
            do {
                picked = TL(again, 1);
                extended = !again.empty() && again.back() - again.front() > o_.max_capture;
                if (!picked && extended && !minus_again.empty())
                    create_gap(out_.double_gaps, ".minus_strand_double_tile_failed.bed", "GAP INTRODUCED ON MINUS STRAND OF DOUBLE TILING OF CHROMOSOME ", minus_again);
                if (picked) MP(picked, minus_again);
            } while (!minus_again.empty() && (picked || extended));
        }
    }
    lap(0);
    print_gaps(out_.gaps, ".coverage_failed.bed", "BASES NOT COVERED ON CHROMOSOME ", pos);
    print_gaps(out_.double_gaps, ".double_tile_failed.bed", "BASES NOT DOUBLE TILED ON CHROMOSOME ", again);
    print_gaps(out_.minus_gaps, ".minus_strand_failed.bed", "BASES NOT COVERED ON MINUS STRAND OF CHROMOSOME ", minus);
    print_gaps(out_.double_minus_gaps, ".minus_strand_double_tile_failed.bed", "BASES NOT DOUBLE TILED ON MINUS STRAND OF CHROMOSOME ", minus_again);
    lap(4);
}

}  // namespace mipgen
