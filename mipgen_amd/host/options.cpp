// options.cpp — flag-compatible command line of the drop-in `mipgen` front end.
// Follows mipgen::set_default_args / parse_command_line / parse_arg_values (/root/reference/mipgen.cpp:164-276,1280-1501):
// same option names, defaults, required set, -file_of_parameters handling and error strings; the usage texts are our own.
#include <cmath>
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <sstream>
#include <stdexcept>

#include "mipgen_host.hpp"

namespace mipgen {

static const char* k_options[] = {
    "-regions_to_scan", "-feature_flank", "-genome_dir", "-project_name", "-bwa", "-bwa_genome_index", "-trf", "-tabix",
    "-check_copy_number", "-common_snps", "-arm_lengths", "-arm_length_sums", "-min_capture_size", "-download_tabix_index",
    "-max_capture_size", "-capture_increment", "-max_mip_overlap", "-starting_mip_overlap", "-stop_optimizing_scores_above",
    "-silent_mode", "-masked_arm_threshold", "-seal_both_strands", "-half_seal_both_strands", "-tag_sizes", "-ext_min_length",
    "-lig_min_length", "-bwa_threads", "-snp_file", "-double_tile_strand_unaware", "-double_tile_strands_separately",
    "-score_method", "-logistic_heuristic", "-file_of_parameters", "-logistic_priority_score", "-svr_priority_score",
    "-logistic_optimal_score", "-svr_optimal_score", "-max_arm_copy_product", "-target_arm_copy",
    "-gpu_copy_counter",           // extension (not in the reference): "on" = exact oligo copy numbers by k-mer counting on the GPU, no bwa
    "-gpus", "-gpu_window_candidates", "-gpu_timing",    // extensions: device workers (0 = all visible), result-window cap (tests), stage timings on stderr
    "-gpu_gather"};                                      // extension: pcie (default) | rccl - how the devices' result windows reach the selection stage

static bool known(const std::string& p)
{
    for (const char* k : k_options) if (p == k) return true;
    return false;
}

static const char* k_usage =
    "\nusage: mipgen (<-parameter_name> <parameter_value>)*\n"
    "MI355X-accelerated MIP designer (drop-in for shendurelab/MIPGEN); `mipgen -doc` lists every option\n"
    "required: -project_name <output prefix>  -bwa_genome_index <indexed reference fasta>  -regions_to_scan <BED>\n"
    "          -min_capture_size <int>  -max_capture_size <int>\n"
    "recommended: -snp_file <tabix-indexed VCF>\n";

static const char* k_doc =
    "\nusage: mipgen (<-parameter_name> <parameter_value>)*\n"
    "required\n"
    "  -project_name  -bwa_genome_index  -regions_to_scan  -min_capture_size  -max_capture_size\n"
    "oligo control\n"
    "  -arm_lengths e1:l1,e2:l2,...      explicit extension:ligation arm length pairs\n"
    "  -arm_length_sums s1,s2,...        all pairs adding up to these sums (default 40,41,42,43,44,45)\n"
    "  -ext_min_length n (16)  -lig_min_length n (18)  -tag_sizes ext,lig (5,0)\n"
    "  -masked_arm_threshold f (0.5)  -target_arm_copy n (20)  -max_arm_copy_product n (75)\n"
    "tools\n"
    "  -bwa path (bwa)  -tabix path (tabix)  -trf path (off)  -bwa_threads n (1)\n"
    "input\n"
    "  -genome_dir dir   per-chromosome fasta files chr<N>.fa\n"
    "  -snp_file vcf     SNPs to avoid\n"
    "  -file_of_parameters file   lines of `-name value`\n"
    "tiling\n"
    "  -feature_flank n (0)  -capture_increment n (5)  -logistic_heuristic off\n"
    "  -max_mip_overlap n (30)  -starting_mip_overlap n (0)  -check_copy_number off\n"
    "  -seal_both_strands on  -half_seal_both_strands on\n"
    "  -double_tile_strand_unaware on  -double_tile_strands_separately on\n"
    "scoring\n"
    "  -score_method logistic|svr|mixed (logistic)\n"
    "  -logistic_optimal_score f (0.98)  -svr_optimal_score f (2.2)\n"
    "  -logistic_priority_score f (0.9)  -svr_priority_score f (1.5)\n"
    "misc\n"
    "  -silent_mode on   skip the all_mips / collapsed_mips files\n"
    "  -gpu_copy_counter on   (extension) arm copy numbers = exact occurrences in the whole -bwa_genome_index fasta (else every chr*.fa of -genome_dir),\n"
    "                         capture-window uniqueness (mapping flag) = no other locus within one substitution; both on the GPU, bwa is not run\n"
    "  -gpus n   (extension) device workers, 0 = every visible GPU        -gpu_timing on   (extension) stage timings on stderr\n"
    "  -gpu_gather pcie|rccl   (extension) result windows come down every GPU's own PCIe link (default), or travel to GPU 0 over RCCL / xGMI first\n"
    "                          (rccl: experimental - run with one rank on hardware and with 2 / 4 ranks against a CPU stand-in of RCCL only; librccl is loaded on first use)\n"
    "limits of this build (the reference has none): arm lengths up to 64 bases, 256 arm-length pairs\n";

static void set_defaults(Options& o)
{
    auto& a = o.args;                       // mipgen.cpp:164-188
    a["-bwa"] = "bwa"; a["-feature_flank"] = "0"; a["-check_copy_number"] = "on"; a["-starting_mip_overlap"] = "0";
    a["-double_tile_strand_unaware"] = "off"; a["-double_tile_strands_separately"] = "off"; a["-silent_mode"] = "off";
    a["-tag_sizes"] = "5,0"; a["-trf"] = "off"; a["-tabix"] = "tabix"; a["-seal_both_strands"] = "off";
    a["-half_seal_both_strands"] = "off"; a["-masked_arm_threshold"] = "0.5"; a["-snp_file"] = "<none>";
    a["-capture_increment"] = "5"; a["-max_mip_overlap"] = "30"; a["-score_method"] = "logistic";
    a["-lig_min_length"] = "18"; a["-ext_min_length"] = "16"; a["-max_arm_copy_product"] = "75";
    a["-target_arm_copy"] = "20"; a["-bwa_threads"] = "1"; a["-gpu_copy_counter"] = "off";
    a["-gpus"] = "0"; a["-gpu_window_candidates"] = "0"; a["-gpu_timing"] = "off"; a["-gpu_gather"] = "pcie";
}

// boost::lexical_cast<int> as the reference uses it for every integer option and BED field: the WHOLE string must be a decimal integer that fits
// (no white space, no trailing characters); a failure is a std::exception that main() reports as "unable to tile sequences" + this text (mipgen.cpp:2033-2035)
int lexical_int(const std::string& s)
{
    if (s.empty() || std::isspace((unsigned char)s[0])) throw BadLexicalCast();
    errno = 0;
    char* end = nullptr;
    const long v = std::strtol(s.c_str(), &end, 10);
    if (end != s.c_str() + s.size() || errno == ERANGE || v < INT32_MIN || v > INT32_MAX) throw BadLexicalCast();
    return (int)v;
}

std::string parse_command_line(int argc, char** argv, Options& o)
{
    set_defaults(o);
    {
        std::string a0(argv[0]);
        size_t e = a0.find_last_of('/');
        o.file_dir = e == std::string::npos ? "" : a0.substr(0, e) + "/";      // mipgen.cpp:137-138
    }
    if (argc == 1) return k_usage;
    if (std::string(argv[1]) == "-doc") return k_doc;
    bool check_file = false;
    for (int i = 0; i < argc; i++) {
        if (argv[i][0] != '-') continue;
        std::string p(argv[i]);
        if (!known(p)) { std::cerr << "not found" << std::endl; return p + " not recognized as valid option\n"; }
        // an option without its value: the reference builds a std::string from argv[argc] = NULL (mipgen.cpp:1297) - a std::logic_error with this text
        if (i + 1 >= argc) throw RefLogicError("basic_string::_M_construct null not valid");
        o.args[p] = argv[i + 1];
        if (p == "-file_of_parameters") check_file = true;
        else if (p == "-arm_length_sums") o.has_arm_length_sums = true;
        else if (p == "-arm_lengths") o.has_arm_lengths = true;
    }
    if (check_file) {
        std::ifstream f(o.args["-file_of_parameters"]);
        if (!f.is_open()) { std::cerr << k_usage << std::endl; return "file of parameters could not be opened\n"; }
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] != '-') continue;
            size_t b = line.find_first_of(' ');
            std::string p = line.substr(0, b);
            if (!known(p)) return p + " not recognized as valid option\n";
            o.args[p] = b == std::string::npos ? std::string() : line.substr(b + 1);
            if (p == "-arm_length_sums") o.has_arm_length_sums = true;
            else if (p == "-arm_lengths") o.has_arm_lengths = true;
        }
    }
    for (const char* need : {"-regions_to_scan", "-project_name", "-max_capture_size", "-min_capture_size", "-bwa_genome_index"})
        if (!o.has(need)) { std::cerr << k_usage << std::endl; return std::string("required parameter ") + need + " not found"; }
    return "";
}

static int to_int(const std::string& s) { return lexical_int(s); }
static double to_double(const std::string& s)
{
    // boost::lexical_cast<double> as the reference's Boost behaves (probed on the real binary, tests/golden/error_cases.json): the whole string, no
    // leading white space, decimal only (strtod would take 0x1p-1), "inf" / "nan" / "infinity" in any case and a leading '+' accepted, a value
    // that overflows rejected, one that underflows (1e-320) accepted
    if (s.empty() || std::isspace((unsigned char)s[0])) throw BadLexicalCast();
    const size_t d0 = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (s.size() > d0 + 1 && s[d0] == '0' && (s[d0 + 1] == 'x' || s[d0 + 1] == 'X')) throw BadLexicalCast();
    errno = 0;
    char* end = nullptr;
    const double v = std::strtod(s.c_str(), &end);
    if (end != s.c_str() + s.size() || (errno == ERANGE && std::fabs(v) > 1.0)) throw BadLexicalCast();
    return v;
}
static std::vector<std::string> split(const std::string& s, char c)
{
    std::vector<std::string> out;
    std::string cur;
    for (char ch : s) { if (ch == c) { out.push_back(cur); cur.clear(); } else cur += ch; }
    out.push_back(cur);
    return out;
}

void finalize_options(Options& o)
{
    auto& a = o.args;
    o.feature_flank = to_int(a["-feature_flank"]);
    {
        size_t b = a["-tag_sizes"].find(',');
        o.ext_tag = to_int(a["-tag_sizes"].substr(0, b));
        o.lig_tag = to_int(a["-tag_sizes"].substr(b + 1));
    }
    o.max_arm_copy = to_int(a["-max_arm_copy_product"]);
    o.target_arm_copy = to_int(a["-target_arm_copy"]);
    o.middle = std::string(o.lig_tag, 'N') + "CTTCAGCTTCCCGATATCCGACGGTAGTGT" + std::string(o.ext_tag, 'N');   // mipgen.cpp:199-200
    o.double_tile = a["-double_tile_strand_unaware"] == "on";
    o.double_tile_strands_separately = a["-double_tile_strands_separately"] == "on";
    if (!o.has("-svr_optimal_score")) a["-svr_optimal_score"] = "2.2";
    if (!o.has("-svr_priority_score")) a["-svr_priority_score"] = "1.5";
    if (!o.has("-logistic_optimal_score")) a["-logistic_optimal_score"] = "0.98";
    if (!o.has("-logistic_priority_score")) a["-logistic_priority_score"] = "0.9";
    const std::string& sm = a["-score_method"];
    if (sm != "logistic" && sm != "svr" && sm != "mixed") { std::cerr << "invalid scoring method given" << std::endl; throw 4; }
    o.score_method = sm == "svr" ? MIPGEN_SCORE_SVR : (sm == "mixed" ? MIPGEN_SCORE_MIXED : MIPGEN_SCORE_LOGISTIC);

    std::map<int, std::list<int>> by_sum;                                      // arm_lengths_by_sum, mipgen.cpp:222-261
    if (o.has_arm_lengths)
        for (const std::string& pr : split(a["-arm_lengths"], ',')) {
            size_t b = pr.find(':');
            int e = to_int(pr.substr(0, b)), l = to_int(pr.substr(b + 1));
            o.oligo_sizes.insert(e); o.oligo_sizes.insert(l);
            by_sum[e + l].push_back(e);
        }
    if (o.has_arm_length_sums || !(o.has_arm_length_sums || o.has_arm_lengths)) {
        const std::string input = o.has_arm_length_sums ? a["-arm_length_sums"] : "40,41,42,43,44,45";
        const int lig_min = to_int(a["-lig_min_length"]), ext_min = to_int(a["-ext_min_length"]);
        for (const std::string& ss : split(input, ',')) {
            int sum = to_int(ss);
            for (int l = lig_min; l <= sum - ext_min && l <= 30; l++) {
                int e = sum - l;
                if (e <= 30) { by_sum[sum].push_back(e); o.oligo_sizes.insert(e); o.oligo_sizes.insert(l); }
            }
            by_sum[sum].sort();
        }
    }
    o.arm_pairs.clear();
    for (auto it = by_sum.rbegin(); it != by_sum.rend(); ++it)                 // sums descending (mipgen.cpp:431), list order (:438)
        for (int e : it->second) o.arm_pairs.push_back({e, it->first - e});
    if (by_sum.empty()) throw std::invalid_argument("no arm length pairs");
    o.max_arm_sum = by_sum.rbegin()->first; o.min_arm_sum = by_sum.begin()->first;

    o.logistic_priority = to_double(a["-logistic_priority_score"]); o.logistic_optimal = to_double(a["-logistic_optimal_score"]);
    o.svr_priority = to_double(a["-svr_priority_score"]); o.svr_optimal = to_double(a["-svr_optimal_score"]);
    o.regions_to_scan = a["-regions_to_scan"]; o.project_name = a["-project_name"]; o.bwa_genome_index = a["-bwa_genome_index"];
    o.max_mip_overlap = to_int(a["-max_mip_overlap"]); o.starting_mip_overlap = to_int(a["-starting_mip_overlap"]);
    o.max_capture = to_int(a["-max_capture_size"]); o.min_capture = to_int(a["-min_capture_size"]);
    o.capture_increment = to_int(a["-capture_increment"]);
    if (o.capture_increment == 0) o.capture_increment = 1;                      // mipgen.cpp:274
    // (the reference casts this option where it uses it - first in design_mip, mipgen.cpp:626 - so a malformed value ends the run THERE, with the
    // output files open: mipgen_design_run raises it at that point)
    try { o.masked_arm_threshold = to_double(a["-masked_arm_threshold"]); } catch (BadLexicalCast&) { o.masked_arm_threshold_bad = true; }
    o.silent = a["-silent_mode"] == "on";
    o.seal_both = a["-seal_both_strands"] == "on";
    o.half_seal_both = a["-half_seal_both_strands"] == "on";
}

mipgen_params Options::accel_params() const
{
    mipgen_params p = {};
    p.abi_version = MIPGEN_ACCEL_ABI_VERSION;
    p.score_method = score_method;
    p.min_capture_size = min_capture; p.max_capture_size = max_capture; p.capture_increment = capture_increment;
    p.max_mip_overlap = max_mip_overlap;
    if (arm_pairs.size() > MIPGEN_MAX_ARM_PAIRS) throw std::invalid_argument("too many arm length pairs");
    p.n_arm_pairs = (int)arm_pairs.size();
    for (size_t i = 0; i < arm_pairs.size(); i++) { p.arm_ext[i] = arm_pairs[i].first; p.arm_lig[i] = arm_pairs[i].second; }
    p.check_copy_number = arg("-check_copy_number") != "off";
    p.logistic_heuristic = arg("-logistic_heuristic") != "off";
    p.masked_arm_threshold = masked_arm_threshold;
    const bool svr = score_method == MIPGEN_SCORE_SVR;                          // mipgen.cpp:264-265
    p.upper_score_limit = svr ? svr_optimal : logistic_optimal;
    p.lower_score_limit = svr ? svr_priority : logistic_priority;
    p.max_arm_copy_product = max_arm_copy; p.target_arm_copy = target_arm_copy;
    // the largest / smallest KEY of the arm-sum map, which may hold an empty list (-arm_length_sums 30,41,62: mipgen.cpp:245-258 -> :421, :434)
    p.arm_sum_key_max = max_arm_sum; p.arm_sum_key_min = std::max(1, min_arm_sum);
    return p;
}

}  // namespace mipgen
