// main.cpp — the drop-in `mipgen` command line: same options, inputs and output files as the reference
// (/root/reference/mipgen.cpp:2021-2037 main, :293-400 query_sequences, :403-556 tile_regions), with candidate
// construction + scoring + replay/condense delegated to libmipgen_accel.so (HIP, gfx950) through its C ABI.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <array>
#include <map>

#include "mipgen_host.hpp"

using namespace mipgen;

namespace {

// -score_method mixed re-scores candidates with the SVR while picking (mipgen.cpp:1523-1527,1873-1877).  Every candidate the
// selection stage can reach is a survivor of the batch's replay + condense, so all of them are scored in ONE accelerator
// call per batch and served from a cache; a candidate that is not in the cache falls back to a single-candidate call.
struct AccelRescorer : Rescorer {
    mipgen_accel* h = nullptr;
    int region_in_batch = 0;
    typedef std::array<int32_t, 6> Key;                      // region in batch, scan start, capture, ext, lig, strand
    std::map<Key, double> cache;
    static Key key(int region, const mipgen_candidate& c) { return Key{region, c.scan_start, c.capture_size, c.ext_len, c.lig_len, c.strand}; }
    void prefetch(const std::vector<mipgen_candidate>& cands)
    {
        cache.clear();
        if (cands.empty()) return;
        std::vector<double> sc(cands.size());
        if (mipgen_accel_score_candidates(h, cands.data(), (int64_t)cands.size(), MIPGEN_SCORE_SVR, sc.data(), nullptr, nullptr, nullptr)) {
            std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl;
            throw 20;
        }
        for (size_t i = 0; i < cands.size(); i++) cache[key(cands[i].region, cands[i])] = sc[i];
    }
    double svr(const Cand& c) override
    {
        mipgen_candidate mc = {region_in_batch, c.scan_start, c.capture, c.ext_len, c.lig_len, c.strand};
        auto it = cache.find(key(region_in_batch, mc));
        if (it != cache.end()) return it->second;
        double s = 0.0;
        if (mipgen_accel_score_candidates(h, &mc, 1, MIPGEN_SCORE_SVR, &s, nullptr, nullptr, nullptr)) {
            std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl;
            throw 20;
        }
        return s;
    }
};

void accel_check(int rc, int code)
{
    if (rc) { std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl; throw code; }
}

int run(int argc, char** argv)
{
    Options o;
    const std::string status = parse_command_line(argc, argv, o);
    if (!status.empty()) { std::cerr << status << std::endl; throw 1; }
    if (std::system(o.arg("-bwa").c_str()) != 256) { std::cerr << "load bwa" << std::endl; throw 2; }            // mipgen.cpp:146-151
    if (o.arg("-trf") != "off" && std::system(o.arg("-trf").c_str()) != 65280) { std::cerr << "TRF directory invalid" << std::endl; throw 3; }
    finalize_options(o);

    Outputs out;
    out.progress.open(o.project_name + ".progress.txt");
    if (!out.progress.is_open()) { std::cerr << "progress file could not be opened" << std::endl; throw 5; }
    out.progress << "mipgen (MI355X accelerated hot path; libmipgen_accel ABI " << mipgen_accel_abi_version() << ")\n";
    for (auto& kv : o.args) out.progress << kv.first << " " << kv.second << std::endl;

    // ---- query_sequences -------------------------------------------------------------------------------------
    std::vector<Region> regions = load_regions(o);
    if (regions.empty()) { std::cerr << "[mipgen] region file could not be opened" << std::endl; throw 6; }
    out.progress << "successfully loaded features for mip design; retrieving chromosomal sequence\n";
    std::cerr << "[mipgen] features loaded; retrieving chromosomal sequence\n";
    if (o.has("-genome_dir")) { if (!load_sequences_from_genome_dir(o, regions)) { std::cerr << "[mipgen] chromosome fasta not acquired" << std::endl; throw 7; } }
    else if (!load_sequences_from_indexed_fasta(o, regions)) { std::cerr << "[mipgen] chromosome fasta not acquired" << std::endl; throw 9; }
    if (!load_masks(o, regions)) std::cerr << "[mipgen] masked chromosome fasta not acquired; no repetitive bases?" << std::endl;
    out.progress << "successfully acquired chromosomal data for mip design; accessing snp file ...\n";
    std::cerr << "[mipgen] regions ready; accessing snp file\n";
    Tables tables;
    load_snps(o, regions, tables);
    out.progress << "all " << tables.snp_load_count << " snps loaded; generating files for bwa\n";
    std::cerr << "[mipgen] all " << tables.snp_load_count << " snps loaded; generating files for bwa\n";
    const std::string copy_status = check_copy_numbers(o, regions, tables);
    if (copy_status.empty()) { std::cerr << "error with copy number analysis" << std::endl; throw 11; }
    out.progress << copy_status;
    find_copy(o, tables);
    out.progress << "bwa copy number analysis finished\n";
    std::cerr << "[mipgen] bwa copy number analysis finished\n";
    open_outputs(o, out);

    // ---- accelerator -------------------------------------------------------------------------------------------
    const mipgen_params ap = o.accel_params();
    mipgen_accel* accel = nullptr;
    accel_check(mipgen_accel_create(&ap, 0, nullptr, &accel), 17);
    if (o.score_method != MIPGEN_SCORE_LOGISTIC)
        accel_check(mipgen_accel_load_model_file(accel, (o.file_dir + "mipgen_svr.model").c_str()), 18);      // mipgen.cpp:409
    for (Region& r : regions) {
        attach_tables(o, tables, r);
        if (o.score_method != MIPGEN_SCORE_LOGISTIC)                                                           // mipgen.cpp:1171,1224
            accel_check(mipgen_accel_long_range_content(accel, r.long_range_seq.data(), (int)r.long_range_seq.size(), r.seq_start, r.seq_stop, r.lrc), 19);
    }

    // ---- tile_regions: batches of regions through the accelerator, then the sequential selection stage ----------
    const int method = o.score_method == MIPGEN_SCORE_SVR ? MIPGEN_SCORE_SVR : MIPGEN_SCORE_LOGISTIC;          // mixed scans with logistic (:467)
    const double lower = method == MIPGEN_SCORE_SVR ? o.svr_priority : o.logistic_priority;
    const double upper = method == MIPGEN_SCORE_SVR ? o.svr_optimal : o.logistic_optimal;
    Selector selector(o, tables, out);
    AccelRescorer rescorer;
    rescorer.h = accel;
    int64_t budget = 48LL << 20;                                             // dense candidates per batch (16 B each on the device)
    if (const char* e = std::getenv("MIPGEN_BATCH_CANDIDATES")) budget = std::max<int64_t>(1, std::atoll(e));   // tests force several batches
    size_t next = 0;
    int feature_counter = 0;
    while (next < regions.size()) {
        std::vector<mipgen_region> batch;
        std::vector<mipgen_grid> grids;
        const size_t first = next;
        int64_t est = 0;
        const int A = (int)o.arm_pairs.size(), K = (o.max_capture - o.min_capture) / o.capture_increment + 1;
        while (next < regions.size()) {
            const Region& r = regions[next];
            const int64_t c = (int64_t)(r.stop_fl - r.start_fl + o.max_capture) * K * A * 2;
            if (!batch.empty() && est + c > budget) break;
            est += c;
            batch.emplace_back();
            fill_accel_region(r, batch.back());
            next++;
        }
        grids.resize(batch.size());
        accel_check(mipgen_accel_upload_regions(accel, batch.data(), (int)batch.size(), grids.data()), 19);
        accel_check(mipgen_accel_score_resident(accel, method), 19);
        accel_check(mipgen_accel_replay_condense(accel), 19);
        const int64_t n_cand = mipgen_accel_batch_candidates(accel);
        int64_t n_pos = 0;
        for (auto& g : grids) n_pos += g.n_pos;
        std::vector<int64_t> emitted_n(batch.size());
        std::vector<mipgen_survivor> surv((size_t)(2 * n_pos));
        std::vector<uint8_t> mask;
        std::vector<double> scores;
        std::vector<uint64_t> records;
        if (!o.silent) {
            mask.resize((size_t)n_cand); scores.resize((size_t)n_cand); records.resize((size_t)n_cand);
            accel_check(mipgen_accel_download_results(accel, scores.data(), records.data(), 0, n_cand), 19);
        }
        accel_check(mipgen_accel_download_replay(accel, emitted_n.data(), surv.data(), (int64_t)surv.size(), o.silent ? nullptr : mask.data(), (int64_t)mask.size()), 19);
        if (o.score_method == MIPGEN_SCORE_MIXED) {
            // every survivor of the batch through the SVR in one call
            std::vector<mipgen_candidate> cands;
            int64_t q0 = 0;
            for (size_t bi = 0; bi < batch.size(); bi++) {
                const mipgen_grid& g = grids[bi];
                for (int64_t q = 2 * q0; q < 2 * (q0 + g.n_pos); q++) {
                    if (surv[(size_t)q].cand_index < 0) continue;
                    const Cand c = make_cand(o, regions[first + bi], g, surv[(size_t)q].cand_index - g.offset, surv[(size_t)q].score, surv[(size_t)q].record);
                    cands.push_back(mipgen_candidate{(int32_t)bi, c.scan_start, c.capture, c.ext_len, c.lig_len, c.strand});
                }
                q0 += g.n_pos;
            }
            rescorer.prefetch(cands);
        }
        int64_t pos0 = 0;
        for (size_t bi = 0; bi < batch.size(); bi++) {
            const Region& r = regions[first + bi];
            const mipgen_grid& g = grids[bi];
            feature_counter++;
            out.progress << "designing all mips for feature #" << feature_counter << std::endl;
            std::cerr << "[mipgen] feature #" << feature_counter << std::endl;
            if (!o.silent) {
                // the reference's generation order: position, size, pair, plus then minus (mipgen.cpp:421-491)
                const int64_t An = (int64_t)o.arm_pairs.size();
                for (int64_t row = 0; row < (int64_t)g.n_pos * g.n_sizes; row++)
                    for (int64_t a = 0; a < An; a++)
                        for (int s = 0; s < 2; s++) {
                            const int64_t i = (row * 2 + s) * An + a;
                            if (!mask[(size_t)(g.offset + i)]) continue;
                            out.all_counter++;
                            const Cand c = make_cand(o, r, g, i, scores[(size_t)(g.offset + i)], records[(size_t)(g.offset + i)]);
                            out.all << format_record(o, r, tables, c, out.all_counter, false);
                        }
            } else out.all_counter += (int)emitted_n[bi];
            out.progress << "condensing feature #" << feature_counter << "\ncollapsing feature #" << feature_counter << std::endl;
            std::vector<mipgen_survivor> rs(surv.begin() + 2 * pos0, surv.begin() + 2 * (pos0 + g.n_pos));
            for (auto& s : rs) if (s.cand_index >= 0) s.cand_index -= g.offset;       // region-local for make_cand
            rescorer.region_in_batch = (int)bi;
            selector.run_region(r, g, rs, o.score_method == MIPGEN_SCORE_MIXED ? &rescorer : nullptr, lower, upper);
            pos0 += g.n_pos;
        }
    }
    mipgen_accel_destroy(accel);
    out.all.close(); out.collapsed.close(); out.picked.close(); out.snp.close();
    out.progress << "mip picking complete:\n" << o.project_name << ".picked_mips.txt\n and \n" << o.project_name << ".snps_mips.txt\n";
    std::cerr << "[mipgen] mip picking complete:\n" << o.project_name << ".picked_mips.txt\n and \n" << o.project_name << ".snp_mips.txt\n";
    if (out.bad_design_count > 0) {
        out.progress << "WARNING: There are " << out.bad_design_count << " gaps in covering supplied regions\n";
        std::cerr << "[mipgen] WARNING: There are " << out.bad_design_count << " gaps in covering supplied regions\n";
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    try {
        return run(argc, argv);
    } catch (int e) {
        if (e != 1) std::cerr << "unable to tile sequences due to circumstance " << e << std::endl;       // mipgen.cpp:2029-2032
        return 1;
    } catch (std::exception& e) {
        std::cerr << "unable to tile sequences" << std::endl << e.what() << std::endl;
        return 1;
    }
}
