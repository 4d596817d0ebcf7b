// design.cpp — libmipgen_host.so: the C ABI of include/mipgen_host.h over the host front end, and the accelerated tile_regions
// driver (one libmipgen_accel handle per GPU, sequential selection stage on the calling thread).
// Reference: /root/reference/mipgen.cpp:2021-2037 main, :293-400 query_sequences, :403-556 tile_regions.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <future>

#include "../../include/mipgen_host.h"
#include "mipgen_host.hpp"
#include "gather.hpp"

using namespace mipgen;

namespace {

thread_local char g_err[8192] = "";        // (holds the whole -doc text: the option documentation travels as the "usage" message, mipgen.cpp:140-145)
thread_local int g_circumstance = 0;

int fail(int code, int circumstance, const std::string& msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg.c_str());
    g_circumstance = circumstance;
    return code;
}

struct FnRescorer : Rescorer {
    mipgen_rescore_fn fn = nullptr;
    void* ctx = nullptr;
    int region = 0;
    double svr(const Cand& c) override
    {
        mipgen_candidate mc = {region, c.scan_start, c.capture, c.ext_len, c.lig_len, c.strand};
        return fn(ctx, region, &mc);
    }
};

}  // namespace

struct mipgen_design {
    Options o;
    std::vector<Region> regions;
    Tables tables;
    Outputs out;
    std::unique_ptr<Selector> selector;
    std::string model_path;
    int next_region = 0;
    bool closed = false;
    bool failed = false;                         // a selection / accelerator failure: the design does not announce its completion (mipgen.cpp:532-533 is only reached on success)
    // -gpu_copy_counter on: the genome the arm oligos are counted against.  The device workers of mipgen_design_tile_regions count their own
    // shard and keep the tables in HBM; host tables are only built for a caller that asks for the regions (mipgen_design_region).
    std::vector<std::string> genome;
    int api_device = 0;                          // the HIP device mipgen_design_region(s) counts the deferred copy numbers on (mipgen_design_set_api_device)
    bool copies_deferred = false;
    std::mutex copies_mu;
    // front-end knobs (mipgen_design_set_*; the command line's -gpus / -gpu_window_candidates / -gpu_timing extension options): no environment
    int n_devices = 0;                           // device workers of mipgen_design_run (0 = every visible device)
    int gather_rccl = 0;                         // -gpu_gather rccl: the workers' result windows travel to GPU 0 over RCCL / xGMI and come down ONE PCIe link
                                                 // (gather.cpp); pcie (default): every device's windows come down its own link
    int64_t window_candidates = 0;               // cap on the candidates of one result window (0 = the default policy)
    bool timing = false;                         // stage timings on stderr
    // the per-region "[mipgen] feature #N" lines of stderr (mipgen.cpp:417): the same bytes in the same order, written a few KB at a time -
    // one write() per region was 200,000 system calls on the critical path of an exome design
    std::string err_lines;
    void flush_err() { if (!err_lines.empty()) { std::cerr.write(err_lines.data(), (std::streamsize)err_lines.size()); err_lines.clear(); } }
};

extern "C" {

const char* mipgen_host_last_error(void) { return g_err; }
int mipgen_host_last_circumstance(void) { return g_circumstance; }

// -gpu_timing on / mipgen_design_set_timing: wall-clock seconds of the front end's stages on stderr (neither library reads the environment)
struct StageClock {
    bool on = false;
    explicit StageClock(bool on_) : on(on_) {}
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what)
    {
        const auto n = std::chrono::steady_clock::now();
        if (on) std::cerr << "[mipgen timing] " << what << ": " << std::chrono::duration<double>(n - t).count() << " s\n";
        t = n;
    }
};

int mipgen_design_open(int argc, const char* const* argv, mipgen_design** out_d)
{
    bool timing_opt = false;
    for (int i = 1; i + 1 < argc; i++) if (argv[i] && argv[i + 1] && std::string(argv[i]) == "-gpu_timing" && std::string(argv[i + 1]) == "on") timing_opt = true;
    StageClock clk(timing_opt);
    if (!out_d || argc < 1 || !argv) return fail(MIPGEN_HOST_E_USAGE, 1, "null argument");
    *out_d = nullptr;
    g_err[0] = 0; g_circumstance = 0;
    std::unique_ptr<mipgen_design> d(new mipgen_design());
    try {
        Options& o = d->o;
        const std::string status = parse_command_line(argc, const_cast<char**>(argv), o);
        if (!status.empty()) return fail(MIPGEN_HOST_E_USAGE, 1, status);
        const bool gpu_copies = o.arg("-gpu_copy_counter") == "on";
        if (!gpu_copies && std::system(o.arg("-bwa").c_str()) != 256) { std::cerr << "load bwa" << std::endl; throw 2; }            // mipgen.cpp:146-151
        if (o.arg("-trf") != "off" && std::system(o.arg("-trf").c_str()) != 65280) { std::cerr << "TRF directory invalid" << std::endl; throw 3; }
        finalize_options(o);
        d->timing = o.arg("-gpu_timing") == "on";
        d->n_devices = std::max(0, std::atoi(o.arg("-gpus").c_str()));
        d->gather_rccl = o.arg("-gpu_gather") == "rccl" ? 1 : 0;
        d->window_candidates = std::max<int64_t>(0, std::atoll(o.arg("-gpu_window_candidates").c_str()));
        d->model_path = o.file_dir + "mipgen_svr.model";                                                              // mipgen.cpp:409
        Outputs& out = d->out;
        out.progress.open(o.project_name + ".progress.txt");
        if (!out.progress.is_open()) { std::cerr << "progress file could not be opened" << std::endl; throw 5; }
        out.progress << "mipgen (MI355X accelerated hot path; libmipgen_accel ABI " << mipgen_accel_abi_version() << ")\n";
        for (auto& kv : o.args) out.progress << kv.first << " " << kv.second << std::endl;
        // ---- query_sequences ---------------------------------------------------------------------------------
        bool bed_opened = false;
        d->regions = load_regions(o, &bed_opened);
        if (!bed_opened) { std::cerr << "[mipgen] region file could not be opened" << std::endl; throw 6; }
        clk.lap("options + BED sort / merge");
        // -gpu_copy_counter: the genome behind the bwa index is read while the region sequences are sliced (it only looks at the regions' chromosome names)
        std::future<void> genome_ready;
        if (gpu_copies) genome_ready = std::async(std::launch::async, [&o, d_ = d.get()]() { load_genome(o, d_->regions, d_->genome); });
        out.progress << "successfully loaded features for mip design; retrieving chromosomal sequence\n";
        std::cerr << "[mipgen] features loaded; retrieving chromosomal sequence\n";
        if (o.has("-genome_dir")) { if (!load_sequences_from_genome_dir(o, d->regions)) { std::cerr << "[mipgen] chromosome fasta not acquired" << std::endl; throw 7; } }
        else if (!load_sequences_from_indexed_fasta(o, d->regions)) { std::cerr << "[mipgen] chromosome fasta not acquired" << std::endl; throw 9; }
        clk.lap("region sequences");
        if (!load_masks(o, d->regions)) std::cerr << "[mipgen] masked chromosome fasta not acquired; no repetitive bases?" << std::endl;
        out.progress << "successfully acquired chromosomal data for mip design; accessing snp file ...\n";
        std::cerr << "[mipgen] regions ready; accessing snp file\n";
        clk.lap("TRF masks");
        load_snps(o, d->regions, d->tables);
        clk.lap("SNPs");
        out.progress << "all " << d->tables.snp_load_count << " snps loaded; generating files for bwa\n";
        std::cerr << "[mipgen] all " << d->tables.snp_load_count << " snps loaded; generating files for bwa\n";
        if (gpu_copies) {
            genome_ready.get();                                                                                      // SURVEY.md section 8f-3 (load_genome, started above)
            d->copies_deferred = true;
            for (Region& r : d->regions) r.copy_deferred = true;
            out.progress << "exact oligo copy numbers and capture-window uniqueness counted on the accelerator\n";
            std::cerr << "[mipgen] oligo copy numbers counted on the accelerator (no bwa)\n";
        } else {
            const std::string copy_status = check_copy_numbers(o, d->regions, d->tables);
            if (copy_status.empty()) { std::cerr << "error with copy number analysis" << std::endl; throw 11; }
            out.progress << copy_status;
            find_copy(o, d->tables);
            out.progress << "bwa copy number analysis finished\n";
            std::cerr << "[mipgen] bwa copy number analysis finished\n";
        }
        clk.lap("arm copy numbers");
        open_outputs(o, out);
        for (Region& r : d->regions) attach_tables(o, d->tables, r);
        d->selector.reset(new Selector(d->o, d->tables, d->out));
        clk.lap("per-region tables");
    } catch (int e) {
        d->flush_err();
        char msg[96];
        snprintf(msg, sizeof msg, "unable to tile sequences due to circumstance %d", e);                             // mipgen.cpp:2029-2032
        return fail(e == 1 ? MIPGEN_HOST_E_USAGE : MIPGEN_HOST_E_INPUT, e, msg);
    } catch (std::exception& e) {
        return fail(MIPGEN_HOST_E_INPUT, exception_circumstance(e), std::string("unable to tile sequences\n") + e.what());
    }
    *out_d = d.release();
    return 0;
}

int mipgen_design_close(mipgen_design* d)
{
    if (!d) return 0;
    if (!d->closed) {
        d->flush_err();
        // the writer thread of the selection stage holds picked / snp records that are not in the files yet
        if (d->selector) { try { d->selector->finish(); } catch (std::exception& e) { d->failed = true; std::cerr << "[mipgen] writing the picked MIPs: " << e.what() << std::endl; } catch (...) { d->failed = true; } }
        Outputs& out = d->out;
        const Options& o = d->o;
        out.all.close(); out.collapsed.close(); out.picked.close(); out.snp.close();
        // the reference announces completion at the end of tile_regions (mipgen.cpp:532-538); a run that threw never gets there (:2029-2035)
        if (!d->failed && d->next_region == (int)d->regions.size()) {
            out.progress << "mip picking complete:\n" << o.project_name << ".picked_mips.txt\n and \n" << o.project_name << ".snps_mips.txt\n";
            std::cerr << "[mipgen] mip picking complete:\n" << o.project_name << ".picked_mips.txt\n and \n" << o.project_name << ".snp_mips.txt\n";
            if (out.bad_design_count > 0) {
                out.progress << "WARNING: There are " << out.bad_design_count << " gaps in covering supplied regions\n";
                std::cerr << "[mipgen] WARNING: There are " << out.bad_design_count << " gaps in covering supplied regions\n";
            }
        }
        out.progress.close();
    }
    delete d;
    return 0;
}

int mipgen_design_params(const mipgen_design* d, mipgen_params* out)
{
    if (!d || !out) return fail(MIPGEN_HOST_E_USAGE, 0, "null argument");
    try { *out = d->o.accel_params(); } catch (std::exception& e) { return fail(MIPGEN_HOST_E_USAGE, 0, e.what()); }
    return 0;
}
int32_t mipgen_design_score_method(const mipgen_design* d) { return d ? d->o.score_method : -1; }
int32_t mipgen_design_silent(const mipgen_design* d) { return d && d->o.silent ? 1 : 0; }
const char* mipgen_design_model_path(const mipgen_design* d) { return d ? d->model_path.c_str() : ""; }
int32_t mipgen_design_region_count(const mipgen_design* d) { return d ? (int32_t)d->regions.size() : 0; }

int mipgen_design_region(const mipgen_design* d, int32_t i, mipgen_region* out)
{
    if (!d || !out || i < 0 || i >= (int32_t)d->regions.size()) return fail(MIPGEN_HOST_E_USAGE, 0, "region index out of range");
    if (d->copies_deferred) {                                          // the caller scores through its own handle: it needs the tables on the host
        mipgen_design* md = const_cast<mipgen_design*>(d);
        std::lock_guard<std::mutex> lock(md->copies_mu);
        if (!md->regions[(size_t)i].copy_ready) {
            try { gpu_copy_numbers(md->o, md->genome, md->regions, 0, -1, md->api_device); } catch (int e) { return fail(MIPGEN_HOST_E_ACCEL, e, std::string("accelerator: ") + mipgen_accel_last_error()); }
            for (Region& r : md->regions) { r.copy_deferred = false; attach_copy_tables(md->o, r); }
        }
    }
    fill_accel_region(d->regions[(size_t)i], *out);
    return 0;
}

int mipgen_design_long_range_seq(const mipgen_design* d, int32_t i, const char** seq, int32_t* len)
{
    if (!d || !seq || !len || i < 0 || i >= (int32_t)d->regions.size()) return fail(MIPGEN_HOST_E_USAGE, 0, "region index out of range");
    *seq = d->regions[(size_t)i].long_range_seq.data();
    *len = (int32_t)d->regions[(size_t)i].long_range_seq.size();
    return 0;
}

int mipgen_design_set_long_range_content(mipgen_design* d, int32_t i, const double* lrc44)
{
    if (!d || !lrc44 || i < 0 || i >= (int32_t)d->regions.size()) return fail(MIPGEN_HOST_E_USAGE, 0, "region index out of range");
    memcpy(d->regions[(size_t)i].lrc, lrc44, sizeof(double) * MIPGEN_N_LRC);
    return 0;
}

int mipgen_design_regions(const mipgen_design* d, int32_t first, int32_t n, mipgen_region* out)
{
    if (!d || n < 0 || first < 0 || first + n > (int32_t)d->regions.size() || (n > 0 && !out)) return fail(MIPGEN_HOST_E_USAGE, 0, "region range out of bounds");
    if (d->copies_deferred && n > 0) {                                 // -gpu_copy_counter: host tables for THIS range only (a rank's shard), counted on the accelerator
        mipgen_design* md = const_cast<mipgen_design*>(d);
        std::lock_guard<std::mutex> lock(md->copies_mu);
        bool ready = true;
        for (int32_t k = 0; k < n; k++) ready = ready && md->regions[(size_t)(first + k)].copy_ready;
        if (!ready) {
            try { gpu_copy_numbers(md->o, md->genome, md->regions, first, first + n, md->api_device); } catch (int e) { return fail(MIPGEN_HOST_E_ACCEL, e, std::string("accelerator: ") + mipgen_accel_last_error()); }
            for (int32_t k = 0; k < n; k++) { Region& r = md->regions[(size_t)(first + k)]; r.copy_deferred = false; attach_copy_tables(md->o, r); }
        }
    }
    for (int32_t k = 0; k < n; k++) { const Region& r = d->regions[(size_t)(first + k)]; if (d->copies_deferred && !r.copy_ready) return fail(MIPGEN_HOST_E_USAGE, 0, "internal: copy tables"); fill_accel_region(r, out[k]); }
    return 0;
}

int mipgen_design_select_region(mipgen_design* d, int32_t i, const mipgen_grid* grid, const mipgen_survivor* survivors, int64_t emitted,
                                const double* scores, const uint64_t* records, const uint8_t* emitted_mask, mipgen_rescore_fn rescore, void* ctx)
{
    return mipgen_design_select_region_collapsed(d, i, grid, survivors, emitted, scores, records, emitted_mask, nullptr, 0, rescore, ctx);
}

namespace {
struct ArrayRescorer {                           // mixed designs through mipgen_design_select_regions: the SVR score of every survivor, parallel to the survivors
    const double* svr = nullptr;
    const mipgen_survivor* surv = nullptr;
    int64_t pos0 = 0;
    const mipgen_grid* g = nullptr;
    static double fn(void* ctx, int32_t, const mipgen_candidate* c)
    {
        auto* self = (ArrayRescorer*)ctx;
        const int64_t k = 2 * (self->pos0 + (c->scan_start - self->g->first_pos)) + c->strand;
        if (c->scan_start < self->g->first_pos || c->scan_start >= self->g->first_pos + self->g->n_pos || self->surv[k].cand_index < 0) {
            std::cerr << "[mipgen] re-score of a candidate that did not survive condense" << std::endl;
            throw 20;
        }
        return self->svr[k];
    }
};
}  // namespace

int mipgen_design_survivor_candidates(const mipgen_design* d, int32_t first, int32_t n, const mipgen_grid* grids, const mipgen_survivor* survivors,
                                      mipgen_candidate* cands, int64_t* where, int64_t capacity, int64_t* count)
{
    if (!d || !count || n < 0 || first < 0 || first + n > (int32_t)d->regions.size() || (n > 0 && (!grids || !survivors)) || (capacity > 0 && (!cands || !where)))
        return fail(MIPGEN_HOST_E_USAGE, 0, "bad arguments");
    const Options& o = d->o;
    int64_t m = 0, q0 = 0;
    for (int32_t bi = 0; bi < n; bi++) {
        const mipgen_grid& g = grids[bi];
        for (int64_t q = 2 * q0; q < 2 * (q0 + g.n_pos); q++) {
            const mipgen_survivor& sv = survivors[q];
            if (sv.cand_index < 0) continue;
            if (m < capacity) {
                const Cand c = make_cand(o, d->regions[(size_t)(first + bi)], g, sv.cand_index - g.offset, sv.score, sv.record);
                cands[m] = mipgen_candidate{bi, c.scan_start, c.capture, c.ext_len, c.lig_len, c.strand};
                where[m] = q;
            }
            m++;
        }
        q0 += g.n_pos;
    }
    *count = m;
    return m > capacity && capacity > 0 ? fail(MIPGEN_HOST_E_USAGE, 0, "candidate capacity too small") : 0;
}

int mipgen_design_select_regions(mipgen_design* d, int32_t first, int32_t n, const mipgen_grid* grids, const mipgen_survivor* survivors, const int64_t* emitted,
                                 const int32_t* collapsed, const int32_t* n_bases, const double* svr)
{
    if (!d || n < 0 || (n > 0 && (!grids || !survivors || !emitted)) || ((collapsed != nullptr) != (n_bases != nullptr)))
        return fail(MIPGEN_HOST_E_USAGE, 0, "bad arguments");
    if (d->o.score_method == MIPGEN_SCORE_MIXED && !svr) return fail(MIPGEN_HOST_E_USAGE, 0, "a mixed design needs the SVR scores of its survivors");
    int64_t pos = 0, col = 0;
    for (int32_t k = 0; k < n; k++) {                                  // silent designs: survivors only (2 per scan position, region after region)
        ArrayRescorer rs;
        rs.svr = svr; rs.surv = survivors; rs.pos0 = pos; rs.g = &grids[k];
        if (int rc = mipgen_design_select_region_collapsed(d, first + k, &grids[k], survivors + 2 * pos, emitted[k], nullptr, nullptr, nullptr,
                                                           collapsed ? collapsed + col : nullptr, collapsed ? n_bases[k] : 0, svr ? &ArrayRescorer::fn : nullptr, svr ? &rs : nullptr)) return rc;
        pos += grids[k].n_pos;
        if (collapsed) col += 2 * (int64_t)n_bases[k];
    }
    return 0;
}

int mipgen_design_select_region_collapsed(mipgen_design* d, int32_t i, const mipgen_grid* grid, const mipgen_survivor* survivors, int64_t emitted,
                                          const double* scores, const uint64_t* records, const uint8_t* emitted_mask, const int32_t* collapsed,
                                          int32_t n_bases, mipgen_rescore_fn rescore, void* ctx)
{
    if (!d || !grid || (!survivors && grid->n_pos > 0)) return fail(MIPGEN_HOST_E_USAGE, 0, "null argument");
    if (i != d->next_region) {
        char msg[128];
        snprintf(msg, sizeof msg, "regions are selected in order: expected region %d, got %d", d->next_region, i);
        return fail(MIPGEN_HOST_E_ORDER, 0, msg);
    }
    if (i >= (int32_t)d->regions.size()) return fail(MIPGEN_HOST_E_USAGE, 0, "region index out of range");
    const Options& o = d->o;
    const Region& r = d->regions[(size_t)i];
    Outputs& out = d->out;
    try {
        out.progress << "designing all mips for feature #" << i + 1 << '\n';        // flushed when the design closes (200,000 regions: no write per line)
        {
            char line[48];
            const int n = snprintf(line, sizeof line, "[mipgen] feature #%d\n", i + 1);
            d->err_lines.append(line, (size_t)n);
            if (d->err_lines.size() >= 4096) d->flush_err();
        }
        if (!o.silent && scores && records && emitted_mask) {
            // the reference's generation order: position, size, pair, plus then minus (mipgen.cpp:421-491)
            const int64_t An = (int64_t)o.arm_pairs.size();
            std::string buf;
            for (int64_t row = 0; row < (int64_t)grid->n_pos * grid->n_sizes; row++)
                for (int64_t a = 0; a < An; a++)
                    for (int s = 0; s < 2; s++) {
                        const int64_t k = (row * 2 + s) * An + a;
                        if (!emitted_mask[k]) continue;
                        out.all_counter++;
                        const Cand c = make_cand(o, r, *grid, k, scores[k], records[k]);
                        buf += format_record(o, r, d->tables, c, out.all_counter, false);
                        if (buf.size() > (1u << 20)) { out.all << buf; buf.clear(); }
                    }
            out.all << buf;
        } else out.all_counter += emitted;
        out.progress << "condensing feature #" << i + 1 << "\ncollapsing feature #" << i + 1 << '\n';
        const int method = o.score_method == MIPGEN_SCORE_SVR ? MIPGEN_SCORE_SVR : MIPGEN_SCORE_LOGISTIC;   // mixed scans with logistic (:467)
        const double lower = method == MIPGEN_SCORE_SVR ? o.svr_priority : o.logistic_priority;
        const double upper = method == MIPGEN_SCORE_SVR ? o.svr_optimal : o.logistic_optimal;
        FnRescorer rs_fn;
        rs_fn.fn = rescore; rs_fn.ctx = ctx; rs_fn.region = i;
        if (o.score_method == MIPGEN_SCORE_MIXED && !rescore) return fail(MIPGEN_HOST_E_USAGE, 0, "a mixed design needs the SVR re-score hook");
        // the survivor and collapse arrays ARE the selection stage's tables (cand_index - grid->offset = the region-local dense index)
        d->selector->run_region(r, *grid, survivors, grid->offset, o.score_method == MIPGEN_SCORE_MIXED ? &rs_fn : nullptr, lower, upper, collapsed, n_bases);
    } catch (int e) {
        d->flush_err();
        d->failed = true;
        char msg[96];
        snprintf(msg, sizeof msg, "unable to tile sequences due to circumstance %d", e);
        return fail(MIPGEN_HOST_E_INPUT, e, msg);
    } catch (std::exception& e) {
        d->flush_err();
        d->failed = true;
        return fail(MIPGEN_HOST_E_INPUT, exception_circumstance(e), std::string("unable to tile sequences\n") + e.what());
    }
    d->next_region = i + 1;
    if (d->next_region == (int)d->regions.size()) d->flush_err();
    return 0;
}

int mipgen_design_set_api_device(mipgen_design* d, int32_t device)
{
    if (!d || device < 0) return fail(MIPGEN_HOST_E_USAGE, 0, "bad argument");
    d->api_device = device;
    return 0;
}
int mipgen_design_set_devices(mipgen_design* d, int32_t n_devices)
{
    if (!d || n_devices < 0) return fail(MIPGEN_HOST_E_USAGE, 0, "bad argument");
    d->n_devices = n_devices;
    return 0;
}
int mipgen_design_set_gather(mipgen_design* d, int32_t rccl)
{
    if (!d) return fail(MIPGEN_HOST_E_USAGE, 0, "null argument");
    d->gather_rccl = rccl ? 1 : 0;
    return 0;
}
int mipgen_design_set_window_candidates(mipgen_design* d, int64_t max_candidates)
{
    if (!d || max_candidates < 0) return fail(MIPGEN_HOST_E_USAGE, 0, "bad argument");
    d->window_candidates = max_candidates;
    return 0;
}
int mipgen_design_set_timing(mipgen_design* d, int32_t on)
{
    if (!d) return fail(MIPGEN_HOST_E_USAGE, 0, "null argument");
    d->timing = on != 0;
    return 0;
}

int mipgen_design_counters(const mipgen_design* d, int64_t* all_mips, int64_t* collapsed, int64_t* picked, int64_t* gaps)
{
    if (!d) return fail(MIPGEN_HOST_E_USAGE, 0, "null argument");
    if (all_mips) *all_mips = d->out.all_counter;
    if (collapsed) *collapsed = d->out.collapsed_counter;
    if (picked) *picked = d->out.picked_counter;
    if (gaps) *gaps = d->out.bad_design_count;
    return 0;
}

}  // extern "C"

// ---- tile_regions on the accelerator ---------------------------------------------------------------------------------------------
namespace {

struct WindowResult {
    int r0 = 0, r1 = 0;                          // design-wide region range of the window
    std::vector<mipgen_grid> grids;              // one per region; offsets batch-wide (the worker's batch)
    std::vector<int64_t> emitted;
    std::vector<mipgen_survivor> surv;           // cand_index batch-wide, like the offsets of `grids` (the selection stage uses their difference only)
    std::vector<double> svr;                     // mixed designs: SVR score of every survivor (parallel to surv; NaN where none)
    std::vector<int32_t> collapsed;              // collapse_mips from the device: 2 entries per base, region after region
    std::vector<int64_t> col_off;                // first entry of every region of the window in `collapsed` (+ total)
    std::vector<char> text;                      // the window's all_mips records, formatted on the device, numbered from the worker's own first record
    // what the consumer reads: the vectors above (-gpu_gather pcie: the worker downloaded them) or the packed buffer of the RCCL gather
    const int64_t* emitted_p = nullptr;
    const mipgen_survivor* surv_p = nullptr;
    const double* svr_p = nullptr;
    const int32_t* collapsed_p = nullptr;
    const char* text_p = nullptr;
    int64_t text_n = 0;
    void point_at_own() { emitted_p = emitted.data(); surv_p = surv.data(); svr_p = svr.data(); collapsed_p = collapsed.data(); text_p = text.data(); text_n = (int64_t)text.size(); }
    // -gpu_gather rccl: the window's arrays in the worker's HBM (nothing was downloaded), the window's first candidate (survivor indices are batch-wide there)
    bool on_device = false;
    mipgen_window_views views;
    int64_t c0 = 0, n_surv = 0;
    int64_t n_rec = 0;                           // how far they advance the design-wide all_mip_counter
    int64_t own_before = 0;                      // records the worker had numbered before this window (it counts its own from 0)
    bool has_text = false;
    bool block_end = false;                      // the last window of one of the worker's region blocks: the consumer moves on to the next device
    bool last = false;
    int error = 0;
    std::string msg;
    // a result object goes round (worker -> channel -> selection thread -> back to the worker: Channel::give_back): its vectors keep their pages, so a
    // window costs no page faults after the first few (fresh vectors of 2 GB per exome were 0.5 s of the worker's time, a third of a logistic design's)
    void reuse()
    {
        r0 = r1 = 0; emitted_p = nullptr; surv_p = nullptr; svr_p = nullptr; collapsed_p = nullptr; text_p = nullptr; text_n = 0;
        on_device = false; views = mipgen_window_views(); c0 = n_surv = n_rec = own_before = 0; has_text = block_end = last = false; error = 0; msg.clear();
    }
};

// the first failure of a run, shared by the device workers (a worker that only learns of the abort reports this one)
struct RecordOrder {
    std::mutex m;
    bool abort = false;
    int err_code = 0;
    std::string err_msg;
    explicit RecordOrder(int) {}
    void stop(int code = 0, const std::string& msg = std::string())
    {
        std::lock_guard<std::mutex> lk(m);
        if (code && !err_code) { err_code = code; err_msg = msg; }
        abort = true;
    }
    void first_error(int* code, std::string* msg) { std::lock_guard<std::mutex> lk(m); *code = err_code; *msg = err_msg; }
};

struct Channel {                                 // worker -> consumer, at most two windows in flight per device
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::unique_ptr<WindowResult>> q;
    bool abort = false;
    bool aborted() { std::lock_guard<std::mutex> lk(m); return abort; }
    void push(std::unique_ptr<WindowResult> r)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return q.size() < 2 || abort; });
        q.push_back(std::move(r));
        cv.notify_all();
    }
    std::unique_ptr<WindowResult> pop()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !q.empty(); });
        auto r = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return r;
    }
    std::unique_ptr<WindowResult> try_pop()
    {
        std::unique_lock<std::mutex> lk(m);
        if (q.empty()) return nullptr;
        auto r = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return r;
    }
    void stop() { std::lock_guard<std::mutex> lk(m); abort = true; cv.notify_all(); }
    // result objects the selection thread has finished with, for the worker to fill again (at most four exist per device: one being filled, two in
    // flight, one being selected)
    std::vector<std::unique_ptr<WindowResult>> spare;
    void give_back(std::unique_ptr<WindowResult> r) { std::lock_guard<std::mutex> lk(m); if (spare.size() < 4) spare.push_back(std::move(r)); }
    std::unique_ptr<WindowResult> fresh()
    {
        std::unique_ptr<WindowResult> r;
        { std::lock_guard<std::mutex> lk(m); if (!spare.empty()) { r = std::move(spare.back()); spare.pop_back(); } }
        if (r) r->reuse(); else r.reset(new WindowResult());
        return r;
    }
    // -gpu_gather rccl: windows of this worker whose arrays the consumer has finished reading from the worker's HBM (the worker must not overwrite
    // its text buffer, nor destroy its handle, before)
    int transferred = 0;
    void mark_transferred() { std::lock_guard<std::mutex> lk(m); transferred++; cv.notify_all(); }
    bool wait_transferred(int n) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return transferred >= n || abort; }); return transferred >= n; }
};

struct SurvivorRescorer {                        // the SVR scores the worker computed for every survivor of the region
    const WindowResult* w = nullptr;
    int64_t pos0 = 0;
    const mipgen_grid* g = nullptr;
    static double fn(void* ctx, int32_t, const mipgen_candidate* c)
    {
        auto* self = (SurvivorRescorer*)ctx;
        const int64_t k = 2 * (self->pos0 + (c->scan_start - self->g->first_pos)) + c->strand;
        if (c->scan_start < self->g->first_pos || c->scan_start >= self->g->first_pos + self->g->n_pos || self->w->surv_p[(size_t)k].cand_index < 0) {
            std::cerr << "[mipgen] re-score of a candidate that did not survive condense" << std::endl;
            throw 20;
        }
        return self->w->svr_p[(size_t)k];
    }
};

// A device worker scores BLOCKS of consecutive regions - block b of the design goes to worker b mod N - so that the order in which the devices
// produce their result windows is the order in which the one selection thread consumes them (design order: mipgen.cpp:503-515, the rand() of :1863
// and the used-arm sets of :1925-1938 couple the regions): with two windows in flight per device every device keeps scoring while the others are
// being selected, whatever the number of windows.  (Contiguous shards did not: device k slept after three windows until devices 0..k-1 were consumed.)
typedef std::vector<std::pair<int, int>> Blocks;

void worker_body(mipgen_design* d, int device, int k_worker, const Blocks& blocks, Channel* ch, RecordOrder* order);

void worker(mipgen_design* d, int device, int k_worker, Blocks blocks, Channel* ch, RecordOrder* order)
{
    // nothing may leave a worker thread as an exception (std::terminate): a failed allocation of a multi-GB result vector ends the run
    // through the channel like any accelerator error
    try { worker_body(d, device, k_worker, blocks, ch, order); }
    catch (std::exception& e) {
        std::unique_ptr<WindowResult> r(new WindowResult());
        r->error = 19; r->msg = std::string("device worker: ") + e.what(); r->last = true;
        order->stop(r->error, r->msg);
        ch->push(std::move(r));
    }
}

void worker_body(mipgen_design* d, int device, int k_worker, const Blocks& blocks, Channel* ch, RecordOrder* order)
{
    // the worker's batch = its blocks one after the other; ridx[i] = the design-wide index of batch region i; a result window never spans two blocks
    std::vector<int> ridx;
    std::vector<int32_t> breaks;
    std::vector<char> block_ends_at;             // [i + 1] set: batch region i is the last of a block
    for (const auto& b : blocks) {
        if (!ridx.empty()) breaks.push_back((int32_t)ridx.size());
        for (int i = b.first; i < b.second; i++) ridx.push_back(i);
    }
    block_ends_at.assign(ridx.size() + 1, 0);
    for (int32_t b : breaks) block_ends_at[(size_t)b] = 1;
    block_ends_at[ridx.size()] = 1;
    int64_t all_before = 0;                      // all_mip_counter at the start of the next window
    auto fail_out = [&](int code, const std::string& msg) {
        std::unique_ptr<WindowResult> r(new WindowResult());
        r->error = code; r->msg = msg; r->last = true;
        order->stop(code, msg);
        ch->push(std::move(r));
    };
    const Options& o = d->o;
    const bool timing = d->timing;               // seconds per stage of this worker
    double t_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // create + model, long-range content, upload, score + replay + collapse (+ downloads), text, mixed re-scores, copy numbers
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](int k) { const auto n_ = std::chrono::steady_clock::now(); t_stage[k] += std::chrono::duration<double>(n_ - t_prev).count(); t_prev = n_; };
    mipgen_accel* h = nullptr;
    mipgen_params ap;
    try { ap = o.accel_params(); } catch (std::exception& e) { fail_out(17, e.what()); return; }
    if (mipgen_accel_create(&ap, device, nullptr, &h)) { fail_out(17, mipgen_accel_last_error()); return; }
    auto bail = [&](int code) { std::string m = mipgen_accel_last_error(); mipgen_accel_destroy(h); fail_out(code, m); };
    if (o.score_method != MIPGEN_SCORE_LOGISTIC && mipgen_accel_load_model_file(h, d->model_path.c_str())) { bail(18); return; }
    const int n = (int)ridx.size();
    lap(0);
    if (o.score_method != MIPGEN_SCORE_LOGISTIC) {                                                                   // mipgen.cpp:1171,1224
        std::vector<const char*> seqs((size_t)n);
        std::vector<int32_t> lens((size_t)n), starts((size_t)n), stops((size_t)n);
        std::vector<double> lrc((size_t)n * MIPGEN_N_LRC);
        for (int i = 0; i < n; i++) {
            const Region& r = d->regions[(size_t)ridx[(size_t)i]];
            seqs[(size_t)i] = r.long_range_seq.data(); lens[(size_t)i] = (int32_t)r.long_range_seq.size();
            starts[(size_t)i] = r.seq_start; stops[(size_t)i] = r.seq_stop;
        }
        if (mipgen_accel_long_range_content_batch(h, n, seqs.data(), lens.data(), starts.data(), stops.data(), lrc.data())) { bail(19); return; }
        for (int i = 0; i < n; i++) memcpy(d->regions[(size_t)ridx[(size_t)i]].lrc, &lrc[(size_t)i * MIPGEN_N_LRC], sizeof(double) * MIPGEN_N_LRC);
    }
    lap(1);
    // -gpu_copy_counter on: this handle counts the arm oligos of its own shard against the genome; the tables never leave its HBM
    // (host tables exist already if a caller asked for the regions before)
    const bool resident = d->copies_deferred && n > 0 && !d->regions[(size_t)ridx[0]].copy_ready;
    if (resident) {
        std::vector<const char*> cs, rs;
        std::vector<int64_t> cl;
        std::vector<int32_t> rl;
        for (const std::string& c : d->genome) { cs.push_back(c.data()); cl.push_back((int64_t)c.size()); }
        for (int i = 0; i < n; i++) { const Region& r = d->regions[(size_t)ridx[(size_t)i]]; rs.push_back(r.seq.data()); rl.push_back((int32_t)r.seq.size()); }
        int64_t n_big = 0;
        const mipgen_big_copy* big = nullptr;
        if (mipgen_accel_count_oligo_copies_resident(h, (int32_t)cs.size(), cs.data(), cl.data(), n, rs.data(), rl.data(), &n_big, &big)) { bail(11); return; }
        for (int i = 0; i < n; i++) { Region& r = d->regions[(size_t)ridx[(size_t)i]]; r.copy_resident = true; r.big_copy.clear(); }
        for (int64_t k = 0; k < n_big; k++) d->regions[(size_t)ridx[(size_t)big[k].region]].big_copy[{big[k].length, big[k].start}] = big[k].copies;
        lap(6);
        // the capture-window half of check_copy_numbers (mapping_failed, mipgen.cpp:615-625, 841-868) for the same shard
        try { gpu_window_flags(o, h, d->genome, d->regions, ridx); } catch (int) { bail(11); return; }
        lap(7);
    }
    lap(6);
    std::vector<mipgen_region> batch((size_t)n);
    for (int i = 0; i < n; i++) fill_accel_region(d->regions[(size_t)ridx[(size_t)i]], batch[(size_t)i], resident);
    std::vector<mipgen_grid> grids((size_t)n);
    // silent designs keep only survivors, so a window may fill the HBM; otherwise a window's all_mips text comes to the host (its dense results never do)
    mipgen_accel_set_window_candidates(h, d->window_candidates > 0 ? d->window_candidates : (o.silent ? 0 : (int64_t)64 << 20));   // (tests force several windows)
    if (mipgen_accel_set_window_breaks(h, breaks.data(), (int32_t)breaks.size())) { bail(19); return; }
    if (mipgen_accel_upload_regions(h, batch.data(), n, grids.data())) { bail(19); return; }
    // svr designs: mipgen.cpp:430 between the capture-size runs of the dense scorer - the tiles of a later run whose positions have all stopped are
    // never scored, exactly as the reference never constructs those candidates (nothing this front end reads lies behind a position's exit).
    // Only where it can pay: the runs are launches of their own (no split along the SV list, one small synchronisation per run), which a shard of
    // fewer than ~3e7 candidates does not fill the chip with - small designs keep the one launch per window with the automatic split.
    if (o.score_method == MIPGEN_SCORE_SVR) {
        int64_t shard_cand = 0;
        for (const mipgen_grid& g : grids) shard_cand += g.count;
        (void)mipgen_accel_set_dynamic_skip(h, shard_cand >= 30000000 ? 1 : 0);
    }
    lap(2);
    const int method = o.score_method == MIPGEN_SCORE_SVR ? MIPGEN_SCORE_SVR : MIPGEN_SCORE_LOGISTIC;               // mixed scans with logistic (:467)
    const int nw = mipgen_accel_window_count(h);
    // Non-silent designs get their all_mips records as text from the device.  The records are numbered design-wide
    // (mipgen.cpp:474,488,792): a worker numbers its own from 0 and the consumer - which drains the workers in order and knows by then how many
    // records the workers before this one wrote - adds that base to the last column while it copies the text (write_renumbered).  No counting
    // pass: every window is scored once, with any number of workers.
    const bool text = !o.silent;
    const bool rccl = d->gather_rccl != 0;       // -gpu_gather rccl: nothing is downloaded here; the consumer posts the window's arrays to GPU 0 (gather.cpp)
    for (int w = 0; w < nw; w++) {
        if (ch->aborted()) { mipgen_accel_destroy(h); return; }       // the selection stage failed: do not score what nobody will consume
        int32_t wr0 = 0, wn = 0;
        int64_t c0 = 0, nc = 0, p0 = 0, np = 0;
        mipgen_accel_window_info(h, w, &wr0, &wn, &c0, &nc, &p0, &np);
        std::unique_ptr<WindowResult> res = ch->fresh();
        res->r0 = wn > 0 ? ridx[(size_t)wr0] : 0; res->r1 = res->r0 + wn; res->last = w == nw - 1;
        res->block_end = block_ends_at[(size_t)(wr0 + wn)] != 0;
        if (wn > 0 && ridx[(size_t)(wr0 + wn - 1)] != res->r1 - 1) { mipgen_accel_destroy(h); fail_out(19, "internal: a result window spans two region blocks"); return; }
        res->grids.assign(grids.begin() + wr0, grids.begin() + wr0 + wn);
        if (!rccl) { res->emitted.resize((size_t)wn); res->surv.resize((size_t)(2 * np)); }
        // a silent design never reads a window's dense results: scored, replayed and condensed in one call, the print-exact re-score on the survivors only
        if (o.silent ? mipgen_accel_score_condense_window(h, w, method) : (mipgen_accel_score_window(h, w, method) || mipgen_accel_replay_condense(h))) { bail(19); return; }
        if (mipgen_accel_collapse(h)) { bail(19); return; }
        res->col_off.assign((size_t)wn + 1, 0);
        for (int bi = 0; bi < wn; bi++) {
            int64_t fe = 0; int32_t nb = 0;
            mipgen_accel_region_bases(h, wr0 + bi, &fe, &nb);
            res->col_off[(size_t)bi + 1] = res->col_off[(size_t)bi] + 2 * (int64_t)nb;
        }
        if (!rccl) {
        res->collapsed.resize((size_t)std::max<int64_t>(res->col_off[(size_t)wn], 1));
        if (mipgen_accel_download_collapsed(h, w, res->collapsed.data(), (int64_t)res->collapsed.size())) { bail(19); return; }
        }
        if (!rccl && mipgen_accel_download_replay(h, res->emitted.data(), res->surv.data(), (int64_t)res->surv.size(), nullptr, 0)) { bail(19); return; }
        lap(3);
        if (text) {
            // print_details on the device: the records leave the GPU as text, the dense results never do
            std::vector<mipgen_record_names> names((size_t)wn);
            for (int bi = 0; bi < wn; bi++) {
                const Region& r = d->regions[(size_t)(res->r0 + bi)];
                names[(size_t)bi] = mipgen_record_names{r.chr.c_str(), r.label.c_str(), r.start - 1, r.stop};
            }
            int64_t n_rec = 0, n_bytes = 0;
            // (rccl: the text buffer of the handle is scratch - the window before must have left it)
            if (rccl && !ch->wait_transferred(w)) { mipgen_accel_destroy(h); return; }
            if (mipgen_accel_format_all_mips(h, names.data(), o.middle.c_str(), all_before, &n_rec, &n_bytes)) { bail(19); return; }
            if (!rccl) {
            res->text.resize((size_t)n_bytes);
            if (mipgen_accel_download_text(h, res->text.data(), n_bytes)) { bail(19); return; }
            }
            res->has_text = true;
            res->n_rec = n_rec;
            res->own_before = all_before;
            all_before += n_rec;
        }
        lap(4);
        if (o.score_method == MIPGEN_SCORE_MIXED) {
            // every survivor of the window through the SVR in one list call ON THE DEVICE (the pick stage re-scores a subset of them, mipgen.cpp:1523-1527,
            // 1873-1877): the candidate list is built from the survivor array in HBM, only the scores come down
            if (mipgen_accel_rescore_survivors(h)) { bail(20); return; }
            if (!rccl) {
                res->svr.assign(res->surv.size(), std::numeric_limits<double>::quiet_NaN());
                if (!res->svr.empty() && mipgen_accel_download_survivor_scores(h, w, res->svr.data(), (int64_t)res->svr.size())) { bail(20); return; }
            }
        }
        lap(5);
        if (rccl) {
            // everything the handle has enqueued is done: the consumer's transfer stream may read the arrays
            if (mipgen_accel_synchronize(h) || mipgen_accel_window_views(h, w, &res->views)) { bail(19); return; }
            res->on_device = true; res->c0 = c0; res->n_surv = 2 * np;
        } else res->point_at_own();
        ch->push(std::move(res));
        t_prev = std::chrono::steady_clock::now();                 // (time blocked on the consumer is not the worker's)
    }
    if (rccl) (void)ch->wait_transferred(nw);                        // the consumer still reads this handle's arrays (or the run was aborted)
    mipgen_accel_destroy(h);
    if (timing) {
        std::ostringstream line;                                     // one write: the selection thread prints to stderr too
        line << "[mipgen timing] device " << device << " worker: create + model " << t_stage[0] << " s, long-range content " << t_stage[1]
             << " s, arm copy numbers (resident) " << t_stage[6] << " s, capture-window uniqueness " << t_stage[7] << " s, upload "
             << t_stage[2] << " s, score + replay + collapse + downloads " << t_stage[3] << " s, record text " << t_stage[4] << " s, mixed re-scores "
             << t_stage[5] << " s\n";
        std::cerr << line.str();
    }
}

}  // namespace

// Relative device time of a region (the shard weights): the candidates of its dense grid - capture sizes after the static skip of
// mipgen.cpp:429, positions of :421-425 - and, for the dense SVR scorer, the factor-table entries it builds per support vector, at the
// kernel's instruction budget (~47 VALU per table entry against ~2.7 per candidate; mipgen_amd/csrc/accel_tiles.hip: build_svr_tiles).  Exons
// with few capture sizes cost more per candidate than their dense-grid size says.  The same rule: mipgen_amd/dist.py: region_cost.
static int64_t region_cost(int start_fl, int stop_fl, int min_capture, int max_capture, int inc, int max_overlap, int n_pairs, int n_e, int n_l,
                           int max_sum, int min_sum, bool svr, int64_t* candidates = nullptr)
{
    const int K_all = (max_capture - min_capture) / inc + 1;
    int k0 = 0;
    while (k0 < K_all) {
        const int C = max_capture - k0 * inc;
        if (C > stop_fl - start_fl + max_overlap && C - inc >= min_capture) k0++; else break;
    }
    const int64_t K = K_all - k0;
    const int64_t n_pos = std::max(0, stop_fl - std::max(0, start_fl - max_capture + max_sum));
    const int64_t cand = n_pos * K * n_pairs * 2;
    if (candidates) *candidates = cand;
    if (!svr) return cand;
    const int64_t ssr = (std::min<int64_t>(K, 9) - 1) * inc + max_sum - min_sum + 1;            // scan sizes of one run of <= 9 capture sizes
    const int64_t runs = (K + 8) / 9;
    const int64_t ent = 2 * runs * (n_pos * (n_e + n_l) + n_pos * ssr);                          // both strands
    return (int64_t)(2.7 * (double)cand + 47.0 * (double)ent);
}

// all_mips text of a worker that numbered its records from 0: the last column is <label>_<index, at least four digits>[_SNP_a|_SNP_b] (print_details,
// mipgen.cpp:792); every index is raised by `base`.  ~0.1 us per record on the consumer thread, instead of a second scoring pass on the device.
static void write_renumbered(std::ostream& os, const char* text, size_t n_bytes, int64_t base)
{
    const char* p = text;
    const char* const end = p + n_bytes;
    std::string buf;
    buf.reserve((size_t)1 << 20);
    char num[32];
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;                                            // end of the line's text
        const char* dend = le;
        if (le - p >= 6 && le[-6] == '_' && le[-5] == 'S' && le[-4] == 'N' && le[-3] == 'P' && le[-2] == '_' && (le[-1] == 'a' || le[-1] == 'b')) dend = le - 6;
        const char* dbeg = dend;
        while (dbeg > p && dbeg[-1] >= '0' && dbeg[-1] <= '9') dbeg--;
        if (dbeg == dend || dbeg == p || dbeg[-1] != '_') { buf.append(p, (size_t)(le - p)); }   // (not a record line: copied as it is)
        else {
            int64_t v = 0;
            for (const char* q = dbeg; q < dend; q++) v = v * 10 + (*q - '0');
            buf.append(p, (size_t)(dbeg - p));
            const int k = snprintf(num, sizeof num, "%04lld", (long long)(v + base));
            buf.append(num, (size_t)k);
            buf.append(dend, (size_t)(le - dend));
        }
        if (nl) buf.push_back('\n');
        p = nl ? nl + 1 : end;
        if (buf.size() >= ((size_t)1 << 20) - 4096) { os.write(buf.data(), (std::streamsize)buf.size()); buf.clear(); }
    }
    if (!buf.empty()) os.write(buf.data(), (std::streamsize)buf.size());
}

extern "C" int mipgen_design_record_names(const mipgen_design* d, int32_t first, int32_t n, mipgen_record_names* out)
{
    if (!d || n < 0 || first < 0 || first + n > (int32_t)d->regions.size() || (n > 0 && !out)) return fail(MIPGEN_HOST_E_USAGE, 0, "region range out of bounds");
    for (int32_t k = 0; k < n; k++) {
        const Region& r = d->regions[(size_t)(first + k)];
        out[k] = mipgen_record_names{r.chr.c_str(), r.label.c_str(), r.start - 1, r.stop};
    }
    return 0;
}

extern "C" const char* mipgen_design_middle(const mipgen_design* d) { return d ? d->o.middle.c_str() : ""; }

extern "C" int mipgen_design_write_all_mips(mipgen_design* d, const char* text, int64_t n_bytes, int64_t renumber_base)
{
    if (!d || n_bytes < 0 || (n_bytes > 0 && !text) || renumber_base < 0) return fail(MIPGEN_HOST_E_USAGE, 0, "bad arguments");
    if (d->o.silent) return fail(MIPGEN_HOST_E_USAGE, 0, "a silent design has no all_mips records");
    if (n_bytes == 0) return 0;
    if (renumber_base) write_renumbered(d->out.all, text, (size_t)n_bytes, renumber_base);
    else d->out.all.write(text, (std::streamsize)n_bytes);
    return 0;
}

static int64_t region_weight_of(const mipgen_design* d, int i, int64_t* candidates = nullptr)
{
    const Options& o = d->o;
    std::set<int> es, ls;
    for (auto& pr : o.arm_pairs) { es.insert(pr.first); ls.insert(pr.second); }
    const Region& r = d->regions[(size_t)i];
    return region_cost(r.start_fl, r.stop_fl, o.min_capture, o.max_capture, o.capture_increment, o.max_mip_overlap, (int)o.arm_pairs.size(), (int)es.size(),
                       (int)ls.size(), o.max_arm_sum, o.min_arm_sum, o.score_method == MIPGEN_SCORE_SVR, candidates);
}

extern "C" int mipgen_design_region_weights(const mipgen_design* d, int64_t* weights, int32_t capacity)
{
    if (!d || !weights || capacity < (int32_t)d->regions.size()) return fail(MIPGEN_HOST_E_USAGE, 0, "bad arguments");
    for (int i = 0; i < (int)d->regions.size(); i++) weights[i] = region_weight_of(d, i);
    return 0;
}

extern "C" int mipgen_design_run(mipgen_design* d, int32_t n_devices)
{
    if (!d) return fail(MIPGEN_HOST_E_USAGE, 0, "null argument");
    const int n = (int)d->regions.size();
    if (n == 0) return 0;                        // a BED without intervals: nothing to tile, the run completes with header-only files like the reference's
    if (d->o.arm_pairs.empty() || d->o.max_capture < d->o.min_capture) {
        // every arm-sum list is empty (-arm_length_sums 39 -ext_min_length 18 -lig_min_length 22), or no capture size at all (-min_capture_size above
        // -max_capture_size: the loop of mipgen.cpp:427 has no iteration): the reference walks the scan positions, constructs nothing (:438) and hands
        // empty tables to the selection stage region after region (:503-515) - no accelerator call to make
        int rc0 = 0;
        for (int i = 0; i < n && rc0 == 0; i++) {
            const Region& r = d->regions[(size_t)i];
            mipgen_grid g = {};
            const int cur = std::max(0, r.start_fl - d->o.max_capture + d->o.max_arm_sum);                                  // :421-422
            g.first_pos = cur + 1; g.n_pos = std::max(0, r.stop_fl - cur);
            std::vector<mipgen_survivor> none(2 * (size_t)g.n_pos);
            for (auto& s : none) { s.cand_index = -1; s.score = 0.0; s.record = 0; }
            try { rc0 = mipgen_design_select_region_collapsed(d, i, &g, none.data(), 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr); }
            catch (int e) { rc0 = fail(MIPGEN_HOST_E_INPUT, e, "unable to tile sequences"); }
            catch (std::exception& e) { rc0 = fail(MIPGEN_HOST_E_INPUT, exception_circumstance(e), std::string("unable to tile sequences\n") + e.what()); }
        }
        d->flush_err();
        if (rc0) d->failed = true;
        return rc0;
    }
    if (d->o.masked_arm_threshold_bad) {
        // -masked_arm_threshold is no number: the reference's first design_mip throws boost::bad_lexical_cast (mipgen.cpp:626) - after the output
        // files were opened; their buffered headers never reach the disk (probed: four empty files, exit status 0)
        d->out.progress << "designing all mips for feature #1\n";
        d->err_lines += "[mipgen] feature #1\n";                        // mipgen.cpp:417: the first region's loop had started
        d->flush_err();
        Outputs& out = d->out;
        const std::string pn = d->o.project_name;
        out.all.close(); out.collapsed.close(); out.picked.close(); out.snp.close();
        for (const char* f : {".all_mips.txt", ".collapsed_mips.txt", ".picked_mips.txt", ".snp_mips.txt"}) { std::ofstream t(pn + f, std::ios::trunc); }
        d->failed = true;
        return fail(MIPGEN_HOST_E_INPUT, -1, std::string("unable to tile sequences\n") + BadLexicalCast().what());
    }
    const int visible = mipgen_accel_device_count();
    if (visible <= 0) return fail(MIPGEN_HOST_E_ACCEL, 17, "no HIP device: the accelerated front end has no CPU path");
    if (n_devices <= 0) n_devices = d->n_devices > 0 ? d->n_devices : visible;
    n_devices = std::max(1, std::min(n_devices, n));
    // Region blocks: consecutive regions of about equal device time (the kernels' cost model), dealt to the devices in turn - block b to device
    // b mod N - and consumed by the selection stage in design order.  One device: one block (the handle cuts its own result windows).  Several:
    // as many blocks per device as it has windows anyway (every window ends at a block's end), at least four where the design is large enough for
    // blocks of 2^26 candidates - small blocks start the selection stage early and even out what the cost model misses.
    std::vector<int64_t> weight((size_t)n);
    int64_t total = 0, cand_total = 0;
    for (int i = 0; i < n; i++) { int64_t c = 0; weight[(size_t)i] = region_weight_of(d, i, &c); total += weight[(size_t)i]; cand_total += c; }
    int n_blocks = 1;
    if (n_devices > 1) {
        const int64_t cap = d->window_candidates > 0 ? d->window_candidates : (d->o.silent ? (int64_t)1 << 30 : (int64_t)64 << 20);
        int64_t m = std::max<int64_t>(1, (cand_total + n_devices * cap - 1) / (n_devices * cap));
        m = std::max(m, std::min<int64_t>(4, cand_total / ((int64_t)n_devices << 26)));
        n_blocks = (int)std::min<int64_t>(n, (int64_t)n_devices * m);
    }
    std::vector<Blocks> blocks_of((size_t)n_devices);
    {
        int lo = 0;
        int64_t acc = 0;
        for (int b = 0; b < n_blocks; b++) {
            const int64_t target = (int64_t)((__int128)total * (b + 1) / n_blocks);
            int hi = lo;
            while (hi < n && (acc + weight[(size_t)hi] <= target || hi == lo) && (n - hi) > (n_blocks - 1 - b)) { acc += weight[(size_t)hi]; hi++; }
            if (b == n_blocks - 1) hi = n;
            blocks_of[(size_t)(b % n_devices)].push_back({lo, hi});
            lo = hi;
        }
    }
    if (d->timing) std::cerr << "[mipgen timing] tile_regions: " << n << " regions, " << cand_total << " dense candidates in " << n_blocks << " region blocks on "
                             << n_devices << " device worker(s)\n";
    // -gpu_gather rccl: one communicator rank per device worker, rank 0 = the root whose HBM the windows are gathered into
    std::unique_ptr<RcclGather> gather;
    std::thread gather_init;
    int gather_init_rc = 0;
    std::string gather_init_err;
    if (d->gather_rccl) {
        std::vector<int> devs;
        for (int k = 0; k < n_devices; k++) devs.push_back(k % visible);
        std::string gerr;
        if (RcclGather::check_devices(devs, &gerr)) { std::cerr << "[mipgen] " << gerr << std::endl; d->failed = true; return fail(MIPGEN_HOST_E_ACCEL, 21, gerr); }
        gather.reset(new RcclGather());
        // RCCL's own set-up takes seconds (5.4 s measured for one rank): it runs beside the workers' start-up (model, counters, upload, first window)
        // and is waited for before the first transfer is posted
        gather_init = std::thread([&gather, devs, &gather_init_rc, &gather_init_err] { gather_init_rc = gather->init(devs, &gather_init_err); });
    }
    std::vector<std::unique_ptr<Channel>> chans;
    std::vector<std::thread> threads;
    RecordOrder order(n_devices);
    for (int k = 0; k < n_devices; k++) {
        chans.emplace_back(new Channel());
        threads.emplace_back(worker, d, k % visible, k, blocks_of[(size_t)k], chans.back().get(), &order);
    }
    int rc = 0;
    StageClock clk(d->timing);
    if (d->selector) d->selector->fine_timing = d->timing;
    double t_wait = 0.0, t_select = 0.0;
    int64_t records_total = 0;                   // all_mips records written so far (the device text of all workers)
    // one result window through the selection stage: its all_mips text (a worker numbers its records from 0: shifted by what the workers before it
    // wrote), then region after region (mipgen.cpp:503-515)
    auto select_window = [&](WindowResult* w) {
        const auto ts0 = std::chrono::steady_clock::now();
        if (w->has_text) {
            // the worker numbered this window's records from its own count; the design's count is what every device has written before it
            const int64_t shift = records_total - w->own_before;
            if (shift) write_renumbered(d->out.all, w->text_p, (size_t)w->text_n, shift);
            else d->out.all.write(w->text_p, (std::streamsize)w->text_n);                           // (the first worker's numbers are the design's)
            records_total += w->n_rec;
        }
        int64_t pos0 = 0;
        for (int bi = 0; bi < w->r1 - w->r0 && rc == 0; bi++) {
            const mipgen_grid& g = w->grids[(size_t)bi];
            SurvivorRescorer rs;
            rs.w = w; rs.pos0 = pos0; rs.g = &g;
            try {
                rc = mipgen_design_select_region_collapsed(d, w->r0 + bi, &g, w->surv_p + 2 * pos0, w->emitted_p[(size_t)bi],
                                                 nullptr, nullptr, nullptr, w->collapsed_p + w->col_off[(size_t)bi],
                                                 (int32_t)((w->col_off[(size_t)bi + 1] - w->col_off[(size_t)bi]) / 2),
                                                 d->o.score_method == MIPGEN_SCORE_MIXED ? &SurvivorRescorer::fn : nullptr, &rs);
            } catch (int e) { rc = fail(MIPGEN_HOST_E_INPUT, e, "unable to tile sequences"); }
            catch (std::exception& e) { rc = fail(MIPGEN_HOST_E_INPUT, exception_circumstance(e), std::string("unable to tile sequences\n") + e.what()); }   // (mipgen.cpp:2033-2035)
            pos0 += g.n_pos;
        }
        t_select += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
    };
    auto worker_failed = [&](WindowResult* w) {
        d->flush_err();
        rc = fail(MIPGEN_HOST_E_ACCEL, w->error, "accelerator: " + w->msg);
        std::cerr << "[mipgen] " << g_err << std::endl;
    };
    if (!d->gather_rccl) {
        for (int b = 0; b < n_blocks && rc == 0; b++) {               // block after block = region after region; block b is device (b mod N)'s next
            const int k = b % n_devices;
            for (;;) {
                const auto tw0 = std::chrono::steady_clock::now();
                std::unique_ptr<WindowResult> w = chans[(size_t)k]->pop();
                t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
                if (w->error) { worker_failed(w.get()); break; }
                select_window(w.get());
                const bool done = w->block_end || w->last || rc;
                chans[(size_t)k]->give_back(std::move(w));                  // its vectors keep their pages for a later window
                if (done) break;
            }
        }
    } else {
        // -gpu_gather rccl: a window arrives as device pointers; its arrays are posted to GPU 0 with ONE grouped RCCL send / receive + one D2H copy
        // (asynchronous), and the selection of the window BEFORE it runs while that transfer is in flight (two receive slots)
        struct Pending { std::unique_ptr<WindowResult> w; int slot = 0; int worker = 0; size_t off[5] = {0, 0, 0, 0, 0}; };
        std::unique_ptr<Pending> pending;
        std::string gerr;
        int next_slot = 0;
        {
            const auto tw0 = std::chrono::steady_clock::now();
            gather_init.join();                                          // the communicators are needed from here on
            t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
            if (gather_init_rc) { d->flush_err(); rc = fail(MIPGEN_HOST_E_ACCEL, 21, gather_init_err); std::cerr << "[mipgen] " << g_err << std::endl; }
        }
        auto finish_pending = [&]() {
            if (!pending) return;
            std::unique_ptr<Pending> p = std::move(pending);
            const auto tw0 = std::chrono::steady_clock::now();
            const int grc = gather->wait(p->slot, &gerr);
            t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
            chans[(size_t)p->worker]->mark_transferred();               // the worker's arrays have been read
            if (grc) { d->flush_err(); rc = fail(MIPGEN_HOST_E_ACCEL, 21, gerr); std::cerr << "[mipgen] " << g_err << std::endl; return; }
            WindowResult* w = p->w.get();
            char* hb = gather->host(p->slot);
            w->emitted_p = (const int64_t*)(hb + p->off[0]);
            w->surv_p = (const mipgen_survivor*)(hb + p->off[1]);
            w->collapsed_p = (const int32_t*)(hb + p->off[2]);
            w->svr_p = (const double*)(hb + p->off[3]);
            w->text_p = hb + p->off[4]; w->text_n = w->views.n_text_bytes;
            select_window(w);
        };
        for (int b = 0; b < n_blocks && rc == 0; b++) {
            const int k = b % n_devices;
            for (;;) {
                std::unique_ptr<WindowResult> w = chans[(size_t)k]->try_pop();
                if (!w) {
                    if (pending) { finish_pending(); if (rc) break; continue; }   // nothing has arrived yet: the window in hand is selected meanwhile
                    const auto tw0 = std::chrono::steady_clock::now();
                    w = chans[(size_t)k]->pop();
                    t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
                }
                if (w->error) { worker_failed(w.get()); break; }
                std::unique_ptr<Pending> p(new Pending());
                const mipgen_window_views& v = w->views;
                const bool mixed = d->o.score_method == MIPGEN_SCORE_MIXED;
                GatherPiece pc[5] = {{v.emitted, (size_t)v.n_emitted * sizeof(int64_t), 0}, {v.survivors, (size_t)v.n_survivors * sizeof(mipgen_survivor), 0},
                                     {v.collapsed, (size_t)v.n_collapsed * sizeof(int32_t), 0}, {mixed ? v.survivor_svr : nullptr, mixed ? (size_t)v.n_survivors * sizeof(double) : 0, 0},
                                     {w->has_text ? v.text : nullptr, w->has_text ? (size_t)v.n_text_bytes : 0, 0}};
                size_t total = 0;
                for (int i = 0; i < 5; i++) { pc[i].offset = total; p->off[i] = total; total += (pc[i].bytes + 15) & ~(size_t)15; }
                if ((mixed && v.n_survivors && !v.survivor_svr) || (v.n_collapsed && !v.collapsed) || v.n_survivors != w->n_surv) { rc = fail(MIPGEN_HOST_E_ACCEL, 21, "rccl gather: a window without its arrays"); break; }
                p->slot = next_slot; next_slot ^= 1; p->worker = k;
                if (gather->post(k, pc, 5, total, p->slot, &gerr)) { d->flush_err(); rc = fail(MIPGEN_HOST_E_ACCEL, 21, gerr); std::cerr << "[mipgen] " << g_err << std::endl; break; }
                const bool last = w->last || w->block_end;
                p->w = std::move(w);
                finish_pending();                                      // the window before this one, while this one's transfer runs
                pending = std::move(p);
                if (last || rc) break;
            }
        }
        if (rc == 0) finish_pending();
        if (clk.on && gather) std::cerr << "[mipgen timing] rccl gather: " << gather->windows << " result windows, " << (double)gather->bytes_moved / 1e6
                                        << " MB to GPU 0 (one grouped send / receive + one D2H copy each): posting " << gather->seconds_posting
                                        << " s, waiting for a transfer that selection did not hide " << gather->seconds_waiting << " s\n";
        gather->destroy();                                             // (waits for a transfer still in flight: the workers' arrays are read until then)
        if (clk.on) std::cerr << "[mipgen timing] rccl gather: communicator set-up " << gather->seconds_init << " s, tear-down " << gather->seconds_destroy << " s\n";
    }
    d->flush_err();
    if (clk.on) std::cerr << "[mipgen timing] tile_regions: waiting for the device workers " << t_wait << " s, selection stage " << t_select << " s\n";
    if (clk.on && d->selector) std::cerr << "[mipgen timing] selection stage: tables " << d->selector->stage_seconds[0] << " s, collapsed output " << d->selector->stage_seconds[1]
                                          << " s, pick " << d->selector->stage_seconds[2] << " s, clean-up " << d->selector->stage_seconds[3] << " s\n"
                                          << "[mipgen timing] pick stage: position sets " << d->selector->pick_seconds[0] << " s, optimize_worst " << d->selector->pick_seconds[1]
                                          << " s, translocate " << d->selector->pick_seconds[2] << " s, manage_picked " << d->selector->pick_seconds[3]
                                          << " s, print_gaps " << d->selector->pick_seconds[4] << " s\n";
    for (auto& c : chans) c->stop();
    if (rc) order.stop();
    if (rc) for (auto& c : chans) { std::lock_guard<std::mutex> lk(c->m); c->q.clear(); }
    for (auto& t : threads) t.join();
    if (rc) d->failed = true;
    return rc;
}

// test hook: the first n values of the private rand() stream (compared with libc's in tests/test_host_select_cpu.py)
extern "C" int mipgen_host_rand_stream(int32_t* out, int32_t n)
{
    GlibcRand g;
    for (int i = 0; i < n; i++) out[i] = g.next();
    return 0;
}
