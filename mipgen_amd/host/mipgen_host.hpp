// mipgen_host.hpp — host side of the drop-in `mipgen` front end (C++17, no Boost).
//
// Mirrors the reference's operator interface for the hot path and the stages either side of it
// (citations into /root/reference):
//   Options            <- mipgen::set_default_args / parse_command_line / parse_arg_values   mipgen.cpp:164-276,1280-1501
//   Region             <- Featurev5                                                           Featurev5.h:7-28
//   load_regions       <- get_features_to_scan (+ compare_regions_to_scan)                    mipgen.cpp:37-67,981-1043
//   load_sequences     <- get_chr_fasta_sequence_from_genome_dir / _using_samtools            mipgen.cpp:1087-1229
//   load_masks/snps/copies <- get_masked_features_to_scan, load_snps/parse_vcf, check_copy_numbers/find_copy
//   Selector           <- collapse_mips, output_collapsed_mips, pick_mips, optimize_worst_in_region,
//                         translocate_down_region, manage_picked_mip, print_gaps, create_gap       mipgen.cpp:1231-1278,1506-1939
//   format_record      <- print_details                                                       mipgen.cpp:765-794
// Candidate construction + scoring (tile_regions' loop body, design_mip, get_score, get_parameters, predict_value)
// is NOT here: it is the accelerator's job (include/mipgen_accel.h).
#pragma once
#include <exception>
#include <deque>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <array>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/mipgen_accel.h"

namespace mipgen {

struct Options {
    std::map<std::string, std::string> args;     // raw option map, as the reference keeps it (mipgen.cpp:76)
    bool has_arm_lengths = false, has_arm_length_sums = false;
    // typed views (parse_arg_values, mipgen.cpp:190-276)
    int feature_flank = 0, ext_tag = 5, lig_tag = 0, max_arm_copy = 75, target_arm_copy = 20;
    int max_mip_overlap = 30, starting_mip_overlap = 0, max_capture = 0, min_capture = 0, capture_increment = 5;
    bool double_tile = false, double_tile_strands_separately = false, silent = false;
    bool seal_both = false, half_seal_both = false;
    double masked_arm_threshold = 0.5;
    bool masked_arm_threshold_bad = false;       // the option's text is no number: the reference only finds out at its first design_mip (mipgen.cpp:626)
    double logistic_priority = 0.9, logistic_optimal = 0.98, svr_priority = 1.5, svr_optimal = 2.2;
    int score_method = MIPGEN_SCORE_LOGISTIC;
    std::string middle;                          // universal_middle_mip_seq (mipgen.cpp:199-200)
    std::string regions_to_scan, project_name, bwa_genome_index, file_dir;
    std::vector<std::pair<int, int>> arm_pairs;  // (ext, lig) in enumeration order: arm sum desc, list order (mipgen.cpp:431-442)
    std::set<int> oligo_sizes;
    int max_arm_sum = 0, min_arm_sum = 0;
    std::string arg(const std::string& k) const { auto it = args.find(k); return it == args.end() ? std::string() : it->second; }
    bool has(const std::string& k) const { return args.count(k) != 0; }
    mipgen_params accel_params() const;
};

// returns "" on success, else the message the reference would print before `throw 1` (mipgen.cpp:140-145)
std::string parse_command_line(int argc, char** argv, Options& o);
// what boost::lexical_cast throws in the reference (every integer / real option and BED field): main() prints its text after "unable to tile sequences"
// The std::exceptions the REFERENCE itself raises on malformed input (a boost::lexical_cast, vector::at on a short BED line, std::string(NULL) for an
// option without its value): its main() prints their text and falls off its end - exit status 0 (mipgen.cpp:2033-2036).  Reproduced for exactly
// these (circumstance -1); any other std::exception is this port's own failure (an allocation, a logic error in the selection or gather code) and
// ends the run with circumstance -2 and exit status 1: a broken accelerated run must not look like a success to a pipeline.
struct ReferenceEquivalent { virtual ~ReferenceEquivalent() = default; };
struct BadLexicalCast : std::exception, ReferenceEquivalent {
    const char* what() const noexcept override { return "bad lexical cast: source type value could not be interpreted as target"; }
};
struct RefOutOfRange : std::out_of_range, ReferenceEquivalent { using std::out_of_range::out_of_range; };
struct RefLogicError : std::logic_error, ReferenceEquivalent { using std::logic_error::logic_error; };
inline int exception_circumstance(const std::exception& e) { return dynamic_cast<const ReferenceEquivalent*>(&e) ? -1 : -2; }
int lexical_int(const std::string& s);
// throws int (4 = invalid scoring method) exactly as parse_arg_values does
void finalize_options(Options& o);

struct Region {                                   // Featurev5
    std::string chr, label;
    int start = 0, stop = 0;                      // start_position, stop_position (1-based, unflanked)
    int start_fl = 0, stop_fl = 0;                // flanked
    std::string seq, masked;                      // chromosomal_sequence, masked_chromosomal_sequence
    int seq_start = 0, seq_stop = 0;
    double lrc[MIPGEN_N_LRC] = {0};
    // per-region slices of the global lookup tables, laid out as mipgen_region expects
    std::vector<std::vector<int32_t>> copy_store;
    std::vector<int32_t> copy_flat;               // -gpu_copy_counter on: [oligo size slot][position], filled by the accelerator in one piece
    std::vector<const int32_t*> copy_ptr;
    std::vector<uint8_t> unmappable, snp_class;
    std::string long_range_seq;                   // region +/- 1000 bases (svr / mixed only)
    // -gpu_copy_counter on, counted by the scoring handle itself: the tables stay in HBM (MIPGEN_COPY_RESIDENT) and the host only holds
    // the counts the 16-bit record fields cannot carry: (oligo length, offset in seq) -> copies >= 65535
    bool copy_resident = false;
    bool copy_deferred = false;                   // -gpu_copy_counter on: no tables yet (design.cpp counts them where they are needed)
    std::map<std::pair<int, int>, int> big_copy;
    bool copy_ready = false;                      // copy_flat already holds the oligo copy numbers (-gpu_copy_counter on)
};

struct Tables {                                   // global lookup tables the input stage fills (mipgen.cpp:81-83)
    std::map<std::string, std::map<int, std::string>> snps;                  // chr_snp_positions
    std::map<int, std::map<std::string, std::set<int>>> unmappable;          // unmappable_positions[size][chr]
    std::map<std::string, std::map<int, std::map<int, int>>> copies;         // copy_chr_start_stop[chr][start][stop]
    int snp_load_count = 0;
};

// input stage; each throws int on the reference's error paths
std::vector<Region> load_regions(const Options& o, bool* opened);                           // *opened false -> caller throws 6; no interval at all is not an error
bool load_sequences_from_genome_dir(const Options& o, std::vector<Region>& regs);
bool load_sequences_from_indexed_fasta(const Options& o, std::vector<Region>& regs);       // native faidx, no samtools fork
bool load_masks(const Options& o, std::vector<Region>& regs);
void load_snps(const Options& o, const std::vector<Region>& regs, Tables& t);
std::string check_copy_numbers(const Options& o, const std::vector<Region>& regs, Tables& t);   // "" on failure
void find_copy(const Options& o, Tables& t);
void attach_tables(const Options& o, const Tables& t, Region& r);                          // fill copy/unmappable/snp slices
// -gpu_copy_counter on: exact oligo copy numbers from the accelerator's k-mer counter instead of the bwa round trip; throws int on failure
void load_genome(const Options& o, const std::vector<Region>& regs, std::vector<std::string>& chroms);   // the genome behind the bwa index
void gpu_copy_numbers(const Options& o, const std::vector<std::string>& chroms, std::vector<Region>& regs, int r0 = 0, int r1 = -1, int device = 0);   // regions r0 .. r1 - 1 (r1 < 0: all), counted on `device`
// -gpu_copy_counter on: Region::unmappable of regions [r0, r1) from the accelerator's window uniqueness test (mipgen.cpp:806-823, 841-868)
void gpu_window_flags(const Options& o, mipgen_accel* h, const std::vector<std::string>& chroms, std::vector<Region>& regs, const std::vector<int>& idx); // host tables (copy_flat)
void attach_copy_tables(const Options& o, Region& r);                                                  // copy_ptr from copy_store / copy_flat
void fill_accel_region(const Region& r, mipgen_region& out, bool resident_copies = false);

// one candidate as the selection stage sees it (the fields of SVMipv4 it reads)
struct Cand {
    int scan_start = 0, scan_stop = 0, ext_start = 0, ext_stop = 0, lig_start = 0, lig_stop = 0;
    int ext_len = 0, lig_len = 0, strand = 0, capture = 0;
    int ext_copy = 0, lig_copy = 0, snp_count = 0;
    double masked = 0.0, score = 0.0;
    char mapping_failed = '0', snp_failed = '0', masking_failed = '0';
    int scan_size() const { return scan_stop - scan_start + 1; }
};
Cand make_cand(const Options& o, const Region& r, const mipgen_grid& g, int64_t local_index, double score, uint64_t record);
Cand make_cand_at(const Options& o, const Region& r, const mipgen_grid& g, int scan_index, uint32_t index_in_position, double score, uint64_t record);

struct Outputs {
    std::ofstream all, collapsed, picked, snp, progress, gaps, double_gaps, minus_gaps, double_minus_gaps;
    int64_t all_counter = 0, collapsed_counter = 0, picked_counter = 0;   // 64-bit: an exome design constructs > 2^31 candidates (the reference's int wraps)
    int bad_design_count = 0;
};
void open_outputs(const Options& o, Outputs& out);       // headers as mipgen.cpp:349-399
std::string format_record(const Options& o, const Region& r, const Tables& t, const Cand& c, int64_t index, bool minor);

// re-scoring hook for -score_method mixed (mipgen.cpp:1523-1527,1873-1877)
struct Rescorer { virtual double svr(const Cand& c) = 0; virtual ~Rescorer() {} };

// glibc's rand() as a never-seeded process sees it (TYPE_3 additive feedback generator, seed 1): the reference picks the strand it
// tries first with `rand() % 2` (mipgen.cpp:1863) and never calls srand, so its picks depend on this exact stream.  A private copy
// keeps the stream out of reach of anything else in the process (device runtime, worker threads) that might call rand().
class GlibcRand {
public:
    GlibcRand();
    int next();
private:
    uint32_t r_[34];
    int f_ = 3, b_ = 0;
};

// chr_strand_pos_used_arm_bases of one chromosome and strand (mipgen.cpp:97,1928-1937): the bases under the arms of the MIPs picked so far.
// The reference keeps a std::set<int> and asks it base by base; the same membership as a growing bitmap (positions are >= 0), with the
// range question answered a word at a time - at exome scale the set holds 10^7 bases and this test is the inner loop of the pick stage.
class UsedBases {
public:
    void insert(int p)
    {
        if (p < 0) return;
        const size_t w = (size_t)p >> 6;
        if (w >= bits_.size()) bits_.resize(std::max(w + 1, bits_.size() * 2), 0);
        bits_[w] |= 1ull << (p & 63);
    }
    void insert_range(int lo, int hi)               // every base of [lo, hi] (an arm of a picked MIP, mipgen.cpp:1928-1937), a word at a time
    {
        if (lo < 0) lo = 0;
        if (hi < lo) return;
        const size_t w0 = (size_t)lo >> 6, w1 = (size_t)hi >> 6;
        if (w1 >= bits_.size()) bits_.resize(std::max(w1 + 1, bits_.size() * 2), 0);
        const uint64_t first = ~0ull << (lo & 63), last = ~0ull >> (63 - (hi & 63));
        if (w0 == w1) { bits_[w0] |= first & last; return; }
        bits_[w0] |= first;
        for (size_t w = w0 + 1; w < w1; w++) bits_[w] = ~0ull;
        bits_[w1] |= last;
    }
    bool any(int lo, int hi) const                  // any used base in [lo, hi]
    {
        if (lo < 0) lo = 0;
        if (hi < lo) return false;
        size_t w0 = (size_t)lo >> 6, w1 = (size_t)hi >> 6;
        if (w0 >= bits_.size()) return false;
        if (w1 >= bits_.size()) { w1 = bits_.size() - 1; hi = (int)(w1 * 64 + 63); }
        const uint64_t first = ~0ull << (lo & 63), last = ~0ull >> (63 - (hi & 63));
        if (w0 == w1) return (bits_[w0] & first & last) != 0;
        if (bits_[w0] & first) return true;
        for (size_t w = w0 + 1; w < w1; w++) if (bits_[w]) return true;
        return (bits_[w1] & last) != 0;
    }
private:
    std::vector<uint64_t> bits_;
};

// positions_to_scan (and its three siblings) of pick_mips (mipgen.cpp:1508-1516): the reference fills a std::set<int> with the flanked
// region, base by base, and only ever erases from it.  Same membership and ascending order as a bitmap over that run; front() / back()
// are *begin() / *rbegin().
class PosSet {
public:
    void fill(int first, int last)                  // every position of [first, last]
    {
        first_ = first;
        const int n = std::max(last - first + 1, 0);
        bits_.assign(((size_t)n + 63) / 64, ~0ull);
        if (n & 63) bits_.back() = ~0ull >> (64 - (n & 63));
        count_ = n; lo_ = 0; hi_ = bits_.empty() ? 0 : bits_.size() - 1;
    }
    bool empty() const { return count_ == 0; }
    int front() const                               // undefined on an empty set, as *begin() is
    {
        while (lo_ < bits_.size() && !bits_[lo_]) lo_++;
        return lo_ < bits_.size() ? first_ + (int)(lo_ * 64) + __builtin_ctzll(bits_[lo_]) : first_;
    }
    int back() const
    {
        while (hi_ > 0 && !bits_[hi_]) hi_--;
        return !bits_.empty() && bits_[hi_] ? first_ + (int)(hi_ * 64) + 63 - __builtin_clzll(bits_[hi_]) : first_;
    }
    void erase(int p)
    {
        const long rel = (long)p - first_;
        if (rel < 0 || (size_t)rel >= bits_.size() * 64) return;
        uint64_t& w = bits_[(size_t)rel >> 6];
        const uint64_t bit = 1ull << (rel & 63);
        if (w & bit) { w &= ~bit; count_--; }
    }
    void erase_range(int lo, int hi)                // every position of [lo, hi] (the scan target of a picked MIP, mipgen.cpp:1938)
    {
        long a = (long)lo - first_, b = (long)hi - first_;
        const long n = (long)bits_.size() * 64;
        if (a < 0) a = 0;
        if (b >= n) b = n - 1;
        if (b < a) return;
        const size_t w0 = (size_t)a >> 6, w1 = (size_t)b >> 6;
        const uint64_t first = ~0ull << (a & 63), last = ~0ull >> (63 - (b & 63));
        for (size_t w = w0; w <= w1; w++) {
            uint64_t m = ~0ull;
            if (w == w0) m &= first;
            if (w == w1) m &= last;
            const uint64_t hit = bits_[w] & m;
            if (hit) { bits_[w] &= ~m; count_ -= __builtin_popcountll(hit); }
        }
    }
    template <class F> void for_each(F f) const     // ascending
    {
        for (size_t w = lo_; w < bits_.size(); w++)
            for (uint64_t b = bits_[w]; b; b &= b - 1) f(first_ + (int)(w * 64) + __builtin_ctzll(b));
    }
private:
    int first_ = 0, count_ = 0;
    mutable size_t lo_ = 0, hi_ = 0;                // words below lo_ / above hi_ are empty (the set only shrinks)
    std::vector<uint64_t> bits_;
};

// scan_strand_best_mip / pos_strand_best_mip of the region in hand (mipgen.cpp:1616-1746, 1748-1908): [position][strand] -> candidate.
// The reference nests two std::map and fills them with heap objects; here NOTHING is built per region: the accelerator's two arrays ARE
// the tables.  scan_strand_best_mip[p][s] is slot 2 * (p - first_pos) + s of the survivor array (present iff its cand_index >= 0),
// pos_strand_best_mip[b][s] is entry 2 * (b - first_pos) + s of the collapse array (the scan-start index of the survivor the fold kept, or
// -1), and a candidate is a pointer into the survivor array: its geometry follows from the slot (scan start, strand) and from a per-design
// table over the index inside the position (capture size, arm lengths) by a few additions whenever a field is read - the pick stage reads
// a handful of fields of a few hundred candidates per region, whereas materialising every survivor (80 bytes each, 10^8 of them per exome
// design) was half of the selection stage's time.  operator[] on the reference's map CREATES a position (:1869) and later find() calls
// see it: a per-region bitmap of touched scan positions reproduces that (a std::set for the positions outside the region's grid).
class Selector {
public:
    Selector(const Options& o, const Tables& t, Outputs& out);
    ~Selector();
    // The picked / snp records are formatted and written by a WRITER thread: the pick stage is what bounds a logistic design (and any design once the
    // scoring is spread over several devices), and print_details of 4-5 10^5 picked MIPs was a third of it.  One FIFO, one consumer: the files' order is
    // the pick order.  finish() drains the queue and joins the thread (idempotent; before the files are closed); it rethrows what the writer caught.
    void finish();
    // survivors: 2 per scan position ('+','-'); cand_index - index_base is the region-local dense index, cand_index < 0 = no survivor.
    // collapsed: optional result of the accelerator's collapse (2 entries per base from g.first_pos on: scan-start index of the
    // winning survivor per strand, or -1; n_bases of them); nullptr = collapse on the host
    void run_region(const Region& r, const mipgen_grid& g, const mipgen_survivor* survivors, int64_t index_base, Rescorer* rescorer,
                    double lower, double upper, const int32_t* collapsed = nullptr, int32_t n_bases = 0);
    double stage_seconds[4] = {0, 0, 0, 0};      // diagnostics (-gpu_timing on): table set-up (+ host collapse), collapsed output, pick, clean-up
    bool fine_timing = false;                    // -gpu_timing on: the pick stage by function (a clock read per call)
    double pick_seconds[5] = {0, 0, 0, 0, 0};    // position sets, optimize_worst, translocate, manage_picked (+ gaps), print_gaps
private:
    using CandPtr = const mipgen_survivor*;                                // into sv_: valid until the next region
    struct Geo { int scan_start, scan_stop, ext_start, ext_stop, lig_start, lig_stop, ext_len, lig_len, strand; };
    struct Within { uint16_t size_index; uint8_t ext_len, lig_len, strand; };   // per index inside a position: (capture size index * 2 + strand) * pairs + pair
    const Options& o_; const Tables& t_; Outputs& out_;
    std::vector<Within> within_;                                           // n_sizes_all * 2 * pairs entries (a few KB: stays in L1)
    std::map<std::string, std::array<UsedBases, 2>> used_;                 // chr_strand_pos_used_arm_bases (persists across regions)
    std::array<UsedBases, 2>* used_cur_ = nullptr;                         // ... of the chromosome of the region in hand
    // the region in hand
    const Region* r_ = nullptr; const mipgen_grid* g_ = nullptr; Rescorer* rs_ = nullptr; double lower_ = 0, upper_ = 0;
    const mipgen_survivor* sv_ = nullptr;                                  // 2 * n_pos slots
    std::vector<mipgen_survivor> sv_copy_;                                 // mixed designs re-score in place (:1873-1877): their own copy
    int64_t base_ = 0, per_pos_ = 0;
    int first_pos_ = 0, n_pos_ = 0, cap0_ = 0;                             // cap0_: capture size of size index 0 of this region
    const int32_t* col_ = nullptr; int32_t n_bases_ = 0;                   // pos_strand_best_mip as scan-start indices
    std::vector<int32_t> col_own_;                                         // ... when the collapse runs here
    std::vector<uint8_t> touched_;                                         // scan positions created by operator[] (:1869)
    std::set<int> touched_outside_;
    GlibcRand rand_;
    struct PickJob { const Region* r; Cand c; int64_t counter; };
    std::thread writer_;
    std::mutex wm_;
    std::condition_variable wcv_;
    std::deque<PickJob> wq_;
    std::vector<PickJob> wlocal_;                                          // the selection thread's own batch: handed over 256 records at a time (a lock + a wake-up
                                                                           // call per record cost as much as print_details itself did)
    void hand_over();
    bool wstop_ = false, wstarted_ = false;
    std::exception_ptr werr_;
    void writer_loop();
    // table views
    CandPtr scan_m(int pos, int s) const
    {
        const long pi = (long)pos - first_pos_;
        if (pi < 0 || pi >= n_pos_) return nullptr;
        const mipgen_survivor* m = sv_ + 2 * pi + s;
        return m->cand_index >= 0 ? m : nullptr;
    }
    bool scan_exists(int pos) const
    {
        const long pi = (long)pos - first_pos_;
        if (pi < 0 || pi >= n_pos_) return !touched_outside_.empty() && touched_outside_.count(pos) != 0;
        return sv_[2 * pi].cand_index >= 0 || sv_[2 * pi + 1].cand_index >= 0 || touched_[(size_t)pi] != 0;
    }
    void scan_touch(int pos)
    {
        const long pi = (long)pos - first_pos_;
        if (pi < 0 || pi >= n_pos_) touched_outside_.insert(pos); else touched_[(size_t)pi] = 1;
    }
    CandPtr pos_m(int pos, int s) const
    {
        const long j = (long)pos - first_pos_;
        if (j < 0 || j >= n_bases_) return nullptr;
        const int32_t pi = col_[2 * j + s];
        return pi >= 0 ? sv_ + 2 * (long)pi + s : nullptr;
    }
    // candidate views
    int slot_of(CandPtr m) const { return (int)(m - sv_); }
    Geo geo(CandPtr m) const
    {
        const int slot = slot_of(m), pi = slot >> 1;
        const Within& w = within_[(size_t)(m->cand_index - base_ - (int64_t)pi * per_pos_)];
        Geo x;
        x.ext_len = w.ext_len; x.lig_len = w.lig_len; x.strand = w.strand;
        const int capture = cap0_ - (int)w.size_index * o_.capture_increment;
        x.scan_start = first_pos_ + pi;
        x.scan_stop = x.scan_start + capture - x.ext_len - x.lig_len - 1;          // mipgen.cpp:449
        if (x.strand == 0) {                                                       // PlusSVMipv4.cpp:9-12
            x.ext_start = x.scan_start - x.ext_len; x.ext_stop = x.scan_start - 1;
            x.lig_start = x.scan_stop + 1; x.lig_stop = x.scan_stop + x.lig_len;
        } else {                                                                   // MinusSVMipv4.cpp:32-35
            x.ext_start = x.scan_stop + 1; x.ext_stop = x.scan_stop + x.ext_len;
            x.lig_start = x.scan_start - x.lig_len; x.lig_stop = x.scan_start - 1;
        }
        return x;
    }
    static int snp_count(CandPtr m) { return (int)MIPGEN_REC_SNP_COUNT(m->record); }
    double masked(CandPtr m, const Geo& x) const { return (double)MIPGEN_REC_MASKED_N(m->record) / (double)(x.lig_len + x.ext_len); }   // mipgen.cpp:610
    int ext_copy(CandPtr m, const Geo& x) const { const int c = (int)MIPGEN_REC_EXT_COPY(m->record); return c == 65535 ? true_copy(x.ext_start, x.ext_len) : c; }
    int lig_copy(CandPtr m, const Geo& x) const { const int c = (int)MIPGEN_REC_LIG_COPY(m->record); return c == 65535 ? true_copy(x.lig_start, x.lig_len) : c; }
    int true_copy(int start, int len) const;
    Cand cand_of(CandPtr m) const;                                         // the materialised record (printing, the SVR re-score hook)
    double rescore(CandPtr m);                                             // mixed designs: SVR score, stored in place
    void collapse();
    void output_collapsed();
    void pick();
    CandPtr optimize_worst(PosSet& positions, int strand_to_use);
    CandPtr translocate(PosSet& positions, int strand_to_use);
    void manage_picked(CandPtr m, PosSet& positions);
    void print_gaps(std::ofstream& f, const std::string& ext, const std::string& note, PosSet& positions);
    void create_gap(std::ofstream& f, const std::string& ext, const std::string& note, PosSet& positions);
    bool arm_used(const Geo& x, int strand) const;
};

}  // namespace mipgen
