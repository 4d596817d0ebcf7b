// inputs.cpp — input stage of the drop-in front end: BED intervals, region sequences, TRF masks, SNPs, BWA copy numbers.
// Semantics follow /root/reference/mipgen.cpp:796-1229 (get_features_to_scan, get_chr_fasta_sequence_*,
// get_masked_features_to_scan, load_snps/parse_vcf, check_copy_numbers, find_copy); the external tools are invoked with
// the same command lines so that real bwa / tabix / trf (or the test stand-ins) see what the reference would hand them.
// One deliberate change: without -genome_dir the region sequences are sliced from the .fai-indexed
// reference directly instead of forking `samtools faidx` twice per region.
#include <dirent.h>
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <iostream>
#include <sstream>

#include "mipgen_host.hpp"

namespace mipgen {

static std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
static std::vector<std::string> split_ws(const std::string& s)            // boost::split(is_any_of(" \t"), token_compress_on)
{
    std::vector<std::string> out;
    std::string cur;
    bool in_sep = false;
    for (char c : s) {
        if (c == ' ' || c == '\t') { if (!in_sep) { out.push_back(cur); cur.clear(); in_sep = true; } }
        else { cur += c; in_sep = false; }
    }
    out.push_back(cur);
    return out;
}

// (compare_regions_to_scan, mipgen.cpp:37-67: header lines first, then chromosome name as a string - without a leading "chr" -, then the integer start;
// applied below from keys parsed once per line)

std::vector<Region> load_regions(const Options& o, bool* opened)
{
    std::vector<Region> out;
    std::ifstream fh(o.regions_to_scan);
    *opened = fh.is_open();                      // (a BED without intervals is not an error: the reference completes with header-only files, mipgen.cpp:986-987)
    if (!fh.is_open()) return out;
    std::cerr << "[mipgen] success on opening region file" << std::endl;
    std::vector<std::string> lines;
    std::string line;
    while (std::getline(fh, line)) {
        line = trim(line);
        if (line.size() > 1 && line[0] != '#') lines.push_back(line);
    }
    // list::sort is a stable merge sort.  Same comparator outcomes as bed_less on the lines themselves, from keys parsed once per line (the string
    // comparator allocated four substrings per comparison: 0.25 s of the 0.31 s this function took for 200,000 intervals)
    {
        struct Key { std::string chr; int start; bool header; };
        std::vector<Key> keys(lines.size());
        for (size_t i = 0; i < lines.size(); i++) {
            const std::string& a = lines[i];
            Key& k = keys[i];
            k.header = a[0] == '>' || a[0] == '#';
            const size_t e = a.find_first_of(" \t");
            k.chr = a.compare(0, 3, "chr") == 0 ? a.substr(3, e == std::string::npos ? std::string::npos : e - 3) : a.substr(0, e);
            k.start = 0;
            if (e != std::string::npos) {
                size_t b = a.find_first_not_of(" \t", e);
                if (b != std::string::npos) k.start = std::atoi(a.c_str() + b);        // atoi stops at the next separator, as the substring did
            }
        }
        std::vector<uint32_t> order(lines.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            const Key& a = keys[x]; const Key& b = keys[y];
            if (a.header) return true;
            if (a.chr != b.chr) return a.chr < b.chr;
            return a.start < b.start;
        });
        std::vector<std::string> sorted(lines.size());
        for (size_t i = 0; i < order.size(); i++) sorted[i].swap(lines[order[i]]);
        lines.swap(sorted);
    }
    std::string default_label = o.project_name;
    if (default_label.rfind('/') != std::string::npos) default_label = default_label.substr(default_label.rfind('/') + 1);
    for (const std::string& l : lines) {
        if (l[0] == '>' || l[0] == '#' || l.find_first_not_of(" \t\n") == std::string::npos) continue;
        std::vector<std::string> f = split_ws(l);
        if (f.size() < 3) {
            // bed_fields.at(1) / .at(2) of the reference (mipgen.cpp:1019,1021,1029): std::out_of_range with libstdc++'s text, reported by main()
            const std::string n = std::to_string(f.size());
            throw RefOutOfRange("vector::_M_range_check: __n (which is " + n + ") >= this->size() (which is " + n + ")");
        }
        const std::string label = f.size() > 3 ? f[3] : default_label;
        const std::string chr = f[0].substr(0, 3) == "chr" ? f[0].substr(3) : f[0];
        const int bs = lexical_int(f[1]), be = lexical_int(f[2]);                    // boost::lexical_cast<int>: strict
        if (!out.empty() && out.back().chr == chr && bs - out.back().stop - 2 * o.feature_flank < o.min_capture / 2) {   // :1019
            Region& p = out.back();
            p.stop = std::max(be, p.stop);
            p.label = label;
            p.stop_fl = p.stop + o.feature_flank;
        } else {
            Region r;
            r.chr = chr; r.label = label; r.start = bs + 1; r.stop = be;
            r.start_fl = r.start - o.feature_flank; r.stop_fl = r.stop + o.feature_flank;
            out.push_back(r);
        }
    }
    return out;
}

static void upper(std::string& s) { for (char& c : s) c = (char)std::toupper((unsigned char)c); }

// get_chr_fasta_sequence_from_genome_dir, mipgen.cpp:1180-1229
// A FASTA file at once: sequence lines joined (memchr / append), one string per '>' record (a file without a header line is one record)
static bool slurp_fasta(const std::string& path, std::vector<std::string>& records)
{
    FILE* fh = fopen(path.c_str(), "rb");
    if (!fh) return false;
    std::string buf;
    if (fseek(fh, 0, SEEK_END) == 0) { const long sz = ftell(fh); if (sz > 0) buf.resize((size_t)sz); rewind(fh); }
    size_t got = buf.empty() ? 0 : fread(&buf[0], 1, buf.size(), fh);
    if (buf.empty()) { char tmp[1 << 16]; size_t k; while ((k = fread(tmp, 1, sizeof tmp, fh)) > 0) buf.append(tmp, k); got = buf.size(); }
    fclose(fh);
    buf.resize(got);
    const size_t first = records.size();
    const char* p = buf.data(); const char* end = p + buf.size();
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* stop = nl ? nl : end;
        const char* next = nl ? nl + 1 : end;
        while (stop > p && stop[-1] == '\r') stop--;
        if (stop > p) {
            if (*p == '>') records.emplace_back();
            else {
                if (records.size() == first) records.emplace_back();
                records.back().append(p, (size_t)(stop - p));
            }
        }
        p = next;
    }
    return true;
}

bool load_sequences_from_genome_dir(const Options& o, std::vector<Region>& regs)
{
    std::ofstream fa(o.project_name + ".feature_sequences.fa");
    const std::string dir = o.arg("-genome_dir");
    // The chromosome files are read and joined ahead of the regions that slice them (up to four loads in flight: a 300 Mb genome is 0.5 s of
    // fread + line joining + toupper on one core).  One load per RUN of regions on the same chromosome, as the reference reloads on every change.
    struct Loaded { bool ok = false; std::string seq; };
    auto load_chr = [dir](std::string chr) {
        Loaded l;
        std::vector<std::string> recs;
        if (!slurp_fasta(dir + "/chr" + chr + ".fa", recs)) return l;
        for (std::string& rec : recs) { upper(rec); if (l.seq.empty()) l.seq.swap(rec); else l.seq += rec; }      // every sequence line of the file, as the reference joins them
        l.ok = true;
        return l;
    };
    std::vector<size_t> run_first;                                               // first region of every run
    for (size_t i = 0; i < regs.size(); i++) if (i == 0 || regs[i].chr != regs[i - 1].chr) run_first.push_back(i);
    std::deque<std::future<Loaded>> ahead;
    size_t next_run = 0;
    auto top_up = [&]() { while (next_run < run_first.size() && ahead.size() < 4) ahead.push_back(std::async(std::launch::async, load_chr, regs[run_first[next_run++]].chr)); };
    std::string chr_seq;
    size_t run = 0;
    for (size_t i = 0; i < regs.size(); i++) {
        Region& r = regs[i];
        if (run < run_first.size() && run_first[run] == i) {
            top_up();
            Loaded l = ahead.front().get();
            ahead.pop_front();
            run++;
            top_up();
            if (!l.ok) { std::cerr << "[mipgen] fasta file could not be opened" << std::endl; for (auto& f : ahead) f.wait(); return false; }
            chr_seq.swap(l.seq);
        }
        const int cs = r.start_fl - o.max_capture < 1 ? 1 : r.start_fl - o.max_capture;
        const int ce = r.stop_fl + o.max_capture + 15 > (int)chr_seq.size() ? (int)chr_seq.size() : r.stop_fl + o.max_capture + 15;
        const int len = ce - cs + 1;
        r.seq = chr_seq.substr((size_t)(cs - 1), (size_t)std::max(len, 0));
        r.seq_start = cs; r.seq_stop = ce;
        fa << ">" << r.chr << ':' << cs << "-" << ce << '\n' << r.seq << '\n';        // flushed when the file closes (200,000 regions: no write per line)
        if (o.score_method != MIPGEN_SCORE_LOGISTIC) {
            long s0 = (long)r.start_fl - o.max_capture - 1 - 1000;               // :1225 (the reference throws if this is negative)
            if (s0 < 0) s0 = 0;
            r.long_range_seq = s0 < (long)chr_seq.size() ? chr_seq.substr((size_t)s0, (size_t)(len + 2000)) : std::string();
        }
    }
    return true;
}

// replaces get_chr_fasta_sequence_using_samtools (mipgen.cpp:1087-1177): same coordinates ([start_fl-maxC, stop_fl+maxC+14],
// long range +/- 1000), read through the .fai index instead of `samtools faidx` child processes
bool load_sequences_from_indexed_fasta(const Options& o, std::vector<Region>& regs)
{
    struct Fai { long len, off; int bases, width; };
    std::map<std::string, Fai> fai;
    {
        std::ifstream f(o.bwa_genome_index + ".fai");
        if (!f.is_open()) { std::cerr << "[mipgen] " << o.bwa_genome_index << ".fai not found (index the reference with samtools faidx)" << std::endl; return false; }
        std::string name; Fai e;
        while (f >> name >> e.len >> e.off >> e.bases >> e.width) fai[name] = e;
    }
    FILE* fp = fopen(o.bwa_genome_index.c_str(), "rb");
    if (!fp) { std::cerr << "genome fasta could not be opened; check file path and permissions?" << std::endl; return false; }
    auto fetch = [&](const Fai& e, long a, long b) {                             // 1-based inclusive, clipped to the sequence
        std::string out;
        a = std::max(a, 1L); b = std::min(b, e.len);
        if (b < a) return out;
        out.reserve((size_t)(b - a + 1));
        long i = a - 1;
        fseek(fp, e.off + (i / e.bases) * e.width + (i % e.bases), SEEK_SET);
        while ((long)out.size() < b - a + 1) {
            int c = fgetc(fp);
            if (c == EOF) break;
            if (c == '\n' || c == '\r') continue;
            out += (char)std::toupper(c);
        }
        return out;
    };
    std::ofstream fa(o.project_name + ".feature_sequences.fa");
    for (Region& r : regs) {
        auto it = fai.find(r.chr);
        if (it == fai.end()) it = fai.find("chr" + r.chr);                      // mipgen.cpp:1106 adds "chr" when the reference uses it
        if (it == fai.end()) { std::cerr << "[mipgen] chromosome " << r.chr << " not in " << o.bwa_genome_index << ".fai" << std::endl; fclose(fp); return false; }
        const long a = (long)r.start_fl - o.max_capture, b = (long)r.stop_fl + o.max_capture + 14;
        r.seq = fetch(it->second, a, b);
        r.seq_start = (int)std::max(a, 1L);
        r.seq_stop = r.seq_start + (int)r.seq.size() - 1;
        fa << ">" << it->first << ':' << a << "-" << b << std::endl;
        for (size_t i = 0; i < r.seq.size(); i += 60) fa << r.seq.substr(i, 60) << std::endl;
        if (o.score_method != MIPGEN_SCORE_LOGISTIC) r.long_range_seq = fetch(it->second, a - 1000, b + 1000);
    }
    fclose(fp);
    return true;
}

// get_masked_features_to_scan, mipgen.cpp:1045-1084
bool load_masks(const Options& o, std::vector<Region>& regs)
{
    const std::string trf = o.arg("-trf");
    if (trf != "off") {
        int rc = std::system((trf + " " + o.project_name + ".feature_sequences.fa 2 7 7 80 10 14 100 -m -h").c_str());
        (void)rc;
    }
    const size_t ps = o.project_name.rfind('/');
    const std::string prefix = ps == std::string::npos ? o.project_name : o.project_name.substr(ps + 1);
    std::ifstream mf(prefix + ".feature_sequences.fa.2.7.7.80.10.14.100.mask");
    if (!mf.is_open() || trf == "off") {
        std::cerr << "[mipgen] no masked feature file found" << std::endl;
        for (Region& r : regs) r.masked = r.seq;
        return trf == "off";
    }
    std::string line;
    size_t idx = 0;
    Region* cur = nullptr;
    while (std::getline(mf, line)) {
        if (line.empty()) continue;
        if (line[0] == '>') { cur = idx < regs.size() ? &regs[idx] : nullptr; idx++; continue; }
        if (cur) cur->masked += line;
    }
    return true;
}

// load_snps + parse_vcf, mipgen.cpp:875-978
void load_snps(const Options& o, const std::vector<Region>& regs, Tables& t)
{
    if (o.arg("-snp_file") == "<none>" || regs.empty()) return;
    std::string query;
    int first = regs.front().start_fl - 500, last = regs.front().stop_fl + 500;
    std::string cur = regs.front().chr, prev;
    for (const Region& r : regs) {
        prev = cur; cur = r.chr;
        const int as = r.start_fl - 500, ae = r.stop_fl + 500;
        if (prev == cur && ae >= last && as <= last) last = ae;
        else {
            query += prev + ":" + std::to_string(first) + "-" + std::to_string(last) + " ";
            first = as; last = ae;
        }
    }
    query += cur + ":" + std::to_string(first) + "-" + std::to_string(last) + " ";
    if (std::system(o.arg("-tabix").c_str()) != 256) { std::cerr << "[mipgen] tabix not loaded" << std::endl; throw 16; }
    const std::string vcf = o.project_name + ".local_snp_data.vcf";
    int rc = std::system((o.arg("-tabix") + " " + o.arg("-snp_file") + " " + query + " > " + vcf).c_str());
    (void)rc;
    std::ifstream f(vcf);
    if (!f.is_open()) { std::cerr << "[mipgen] VCF file could not be opened" << std::endl; return; }
    std::string line;
    while (std::getline(f, line)) {
        if (line.size() < 2 || line[0] == '#') continue;
        size_t a = 0, b = line.find_first_of(" \t", 0);
        const std::string chr = line.substr(a, b);
        a = b + 1; b = line.find_first_of(" \t", a);
        const int pos = std::atoi(line.substr(a, b - a).c_str());
        a = b + 1; b = line.find_first_of(" \t", a);                            // id
        a = b + 1; b = line.find_first_of(" \t", a);
        const std::string ref = line.substr(a, b - a);
        a = b + 1; b = line.find_first_of(" \t", a);
        const std::string alt = line.substr(a, b == std::string::npos ? std::string::npos : b - a);
        const std::string alleles = ref + alt;
        if (ref.size() > 1) for (size_t i = 1; i < ref.size(); i++) t.snps[chr][pos + (int)i] = alleles;
        else t.snps[chr][pos] = alleles;
        t.snp_load_count++;
    }
}

// check_copy_numbers, mipgen.cpp:796-873
std::string check_copy_numbers(const Options& o, const std::vector<Region>& regs, Tables& t)
{
    const std::string pn = o.project_name, bwa = o.arg("-bwa");
    {
        std::ofstream fq(pn + ".all_sequences.fq"), arms(pn + ".oligo_copy_count.fq");
        if (!fq.is_open() || !arms.is_open()) { std::cerr << "[mipgen] copy_check files could not be opened" << std::endl; return ""; }
        for (const Region& r : regs) {
            for (int size = o.max_capture; size >= o.min_capture; size -= o.capture_increment) {
                for (int start = r.start_fl - size; start < r.stop_fl; start++) {
                    if (start > 0 && start + size - 1 <= r.seq_stop) {
                        const long rel = (long)start - r.seq_start;
                        if (rel < 0 || rel > (long)r.seq.size()) continue;         // the reference's substr would throw here
                        fq << "@" << size << "_" << r.chr << "_" << start << "\n" << r.seq.substr((size_t)rel, (size_t)size) << "\n+\n"
                           << std::string((size_t)size, '#') << "\n";
                    }
                }
            }
            for (int size : o.oligo_sizes) {
                if ((long)r.seq.size() - size <= 0) continue;
                for (size_t rel = 0; rel < r.seq.size() - (size_t)size; rel++) {
                    arms << "@chr" << r.chr << ":" << (r.seq_start + (long)rel) << "-" << (r.seq_start + (long)rel + size - 1) << "\n"
                         << r.seq.substr(rel, (size_t)size) << "\n+\n" << std::string((size_t)size, '#') << "\n";
                }
            }
        }
    }
    int rc = std::system((bwa + " aln -t " + o.arg("-bwa_threads") + " " + o.bwa_genome_index + " " + pn + ".all_sequences.fq > " + pn + ".all_sequences.sai").c_str());
    rc = std::system((bwa + " samse " + o.bwa_genome_index + " " + pn + ".all_sequences.sai " + pn + ".all_sequences.fq > " + pn + ".all_sequences.sam").c_str());
    (void)rc;
    std::ifstream sam(pn + ".all_sequences.sam");
    if (!sam.is_open()) return "";
    std::cerr << "[mipgen] sam file opened" << std::endl;
    int bad = 0;
    bool header = true;
    std::string line;
    while (std::getline(sam, line)) {
        const size_t n = line.size();
        if (header && line.find("X0:i:") >= n - 1) continue;                      // unsigned arithmetic as in the reference (:853)
        else if (header) header = false;
        if (n < 2) continue;
        if (line.find("X0:i:1") >= n - 1 || line.find("X1:i:0") >= n - 1) {       // :857
            const std::string key = line.substr(0, line.find('\t'));
            const size_t s1 = key.find('_'), s2 = key.rfind('_');
            const int size = std::atoi(key.substr(0, s1).c_str());
            const std::string chr = key.substr(s1 + 1, s2 - s1 - 1);
            const int pos = std::atoi(key.substr(s2 + 1).c_str());
            t.unmappable[size][chr].insert(pos);
            bad++;
        }
    }
    return std::to_string(bad) + " ambiguously mapping start positions must be avoided\n";
}

// find_copy, mipgen.cpp:558-596
void find_copy(const Options& o, Tables& t)
{
    const std::string pn = o.project_name, bwa = o.arg("-bwa");
    int rc = std::system((bwa + " aln -t " + o.arg("-bwa_threads") + " " + o.bwa_genome_index + " " + pn + ".oligo_copy_count.fq > " + pn + ".oligo_copy_count.sai").c_str());
    rc = std::system((bwa + " samse " + o.bwa_genome_index + " " + pn + ".oligo_copy_count.sai " + pn + ".oligo_copy_count.fq > " + pn + ".oligo_copy_count.sam").c_str());
    (void)rc;
    std::ifstream sam(pn + ".oligo_copy_count.sam");
    if (!sam.is_open()) return;
    std::cerr << "[mipgen] checking oligo copy" << std::endl;
    std::string line;
    while (std::getline(sam, line)) {
        if (line.size() <= 1 || line[0] == '@') continue;
        const size_t c0 = line.find("chr", 0) + 3, c1 = line.find(':', c0), s1 = line.find('-', c1 + 1), e1 = line.find('\t', s1 + 1);
        const std::string chr = line.substr(c0, c1 - c0);
        const int start = std::atoi(line.substr(c1 + 1, s1 - c1 - 1).c_str()), stop = std::atoi(line.substr(s1 + 1, e1 - s1 - 1).c_str());
        const size_t x0 = line.find("X0:i:", 0);
        if (x0 < line.size()) t.copies[chr][start][stop] = std::atoi(line.substr(x0 + 5, line.find('\t', x0 + 5) - (x0 + 5)).c_str());
        else t.copies[chr][start][stop] = 100;                                    // :591
    }
}

static char comp(char c)
{
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; default: return 0; }
}

void attach_copy_tables(const Options& o, Region& r)
{
    const size_t n = r.seq.size();
    r.copy_ptr.assign(MIPGEN_MAX_OLIGO + 1, nullptr);
    size_t k = 0;
    for (int len : o.oligo_sizes) {
        if (len < 0 || len > MIPGEN_MAX_OLIGO) continue;
        r.copy_ptr[(size_t)len] = r.copy_ready ? r.copy_flat.data() + k * n : r.copy_store[k].data();
        k++;
    }
}

// slices of the global tables in the layout of mipgen_region (include/mipgen_accel.h)
void attach_tables(const Options& o, const Tables& t, Region& r)
{
    const int n = (int)r.seq.size();
    r.copy_store.clear();
    r.copy_ptr.clear();
    auto cit = t.copies.find(r.chr);
    for (int len : o.oligo_sizes) {
        if (r.copy_ready || r.copy_deferred) break;
        if (len < 0 || len > MIPGEN_MAX_OLIGO) continue;
        r.copy_store.emplace_back((size_t)n, 0);                                 // absent key -> 0 (std::map::operator[], mipgen.cpp:612-613)
        std::vector<int32_t>& v = r.copy_store.back();
        if (cit != t.copies.end()) {
            auto lo = cit->second.lower_bound(r.seq_start), hi = cit->second.upper_bound(r.seq_stop);
            for (auto it = lo; it != hi; ++it) {
                auto e = it->second.find(it->first + len - 1);
                if (e != it->second.end()) v[(size_t)(it->first - r.seq_start)] = e->second;
            }
        }
    }
    if (!r.copy_deferred) attach_copy_tables(o, r);            // deferred: counted by the scoring handle (design.cpp: worker), or on the first mipgen_design_region
    r.unmappable.clear();
    if (!t.unmappable.empty()) {
        const int K = (o.max_capture - o.min_capture) / o.capture_increment + 1;
        r.unmappable.assign((size_t)K * n, 0);
        for (int k = 0; k < K; k++) {
            auto s = t.unmappable.find(o.max_capture - k * o.capture_increment);
            if (s == t.unmappable.end()) continue;
            auto c = s->second.find(r.chr);
            if (c == s->second.end()) continue;
            for (auto it = c->second.lower_bound(r.seq_start); it != c->second.end() && *it <= r.seq_stop; ++it)
                r.unmappable[(size_t)k * n + (size_t)(*it - r.seq_start)] = 1;
        }
    }
    r.snp_class.clear();
    auto sit = t.snps.find(r.chr);
    if (sit != t.snps.end() && !sit->second.empty()) {
        r.snp_class.assign((size_t)n, 0);
        for (auto it = sit->second.lower_bound(r.seq_start); it != sit->second.end() && it->first <= r.seq_stop; ++it) {
            const std::string& al = it->second;
            const char g = r.seq[(size_t)(it->first - r.seq_start)];
            bool ok = false;                                                      // can an alternate-allele arm be generated? (mipgen.cpp:644-682)
            if (al.size() == 2 && al[0] != 'N' && al[1] != 'N' && al[0] != '-' && al[1] != '-')
                ok = g == al[0] || (comp(al[0]) != 0 && g == comp(al[0]));
            r.snp_class[(size_t)(it->first - r.seq_start)] = ok ? 1 : 2;
        }
    }
}

// the arm-oligo copy numbers (mipgen.cpp:558-596, 825-835) as exact occurrence counts against the whole genome,
// counted by the accelerator in one streaming pass per chromosome - no FASTQ files, no bwa.
// The genome bwa would have searched = the whole -bwa_genome_index: that FASTA itself when it is readable, else EVERY chr*.fa of -genome_dir
// (not only the chromosomes that carry a region: an oligo's copies on the other chromosomes count, mipgen.cpp:560-561 aligns against the
// whole index).  Which files were counted against is reported on stderr.
void load_genome(const Options& o, const std::vector<Region>& regs, std::vector<std::string>& chroms)
{
    chroms.clear();
    auto read_fasta = [&](const std::string& path) { return slurp_fasta(path, chroms); };
    {
        std::ifstream probe(o.bwa_genome_index);
        char c = 0;
        if (probe.is_open() && probe.get(c) && c == '>') {
            if (!read_fasta(o.bwa_genome_index)) { std::cerr << "genome fasta could not be opened; check file path and permissions?" << std::endl; throw 9; }
            std::cerr << "[mipgen] -gpu_copy_counter: counting against the " << chroms.size() << " sequence(s) of " << o.bwa_genome_index << std::endl;
            return;
        }
    }
    if (!o.has("-genome_dir")) { std::cerr << "genome fasta could not be opened; check file path and permissions?" << std::endl; throw 9; }
    const std::string dir = o.arg("-genome_dir");
    std::vector<std::string> files;
    if (DIR* dh = opendir(dir.c_str())) {
        while (struct dirent* e = readdir(dh)) {
            const std::string n = e->d_name;
            if (n.size() > 6 && n.compare(0, 3, "chr") == 0 && n.compare(n.size() - 3, 3, ".fa") == 0) files.push_back(n);
        }
        closedir(dh);
    }
    std::sort(files.begin(), files.end());
    std::set<std::string> have(files.begin(), files.end());
    for (const Region& r : regs) if (!have.count("chr" + r.chr + ".fa")) { std::cerr << "[mipgen] fasta file could not be opened" << std::endl; throw 7; }
    for (const std::string& f : files) if (!read_fasta(dir + "/" + f)) { std::cerr << "[mipgen] fasta file could not be opened" << std::endl; throw 7; }
    std::cerr << "[mipgen] -gpu_copy_counter: -bwa_genome_index is not a readable FASTA; counting against the " << files.size() << " chr*.fa file(s) of " << dir
              << " (copies elsewhere in the genome are NOT seen)" << std::endl;
}

// -gpu_copy_counter on: the capture-window half of check_copy_numbers (mipgen.cpp:806-823, 841-868) through mipgen_accel_window_uniqueness, for the
// regions idx[] (design-wide indices, the order of the accelerator batch): Region::unmappable = 1 for the window starts the reference enumerates ([start_fl - C, stop_fl), :808-813) whose window is
// not unique within one substitution.  Skipped with -check_copy_number off (the flag is then never consulted, :619).
void gpu_window_flags(const Options& o, mipgen_accel* h, const std::vector<std::string>& chroms, std::vector<Region>& regs, const std::vector<int>& idx)
{
    const int nr = (int)idx.size();
    if (o.arg("-check_copy_number") == "off" || nr <= 0) return;
    const int K = (o.max_capture - o.min_capture) / o.capture_increment + 1;
    std::vector<int32_t> sizes((size_t)K);
    for (int k = 0; k < K; k++) sizes[(size_t)k] = o.max_capture - k * o.capture_increment;
    const int seed = std::min(31, std::max(12, o.oligo_sizes.empty() ? 30 : *o.oligo_sizes.rbegin()));
    if (o.min_capture < 2 * seed) {
        std::cerr << "[mipgen] -gpu_copy_counter: capture sizes below " << 2 * seed << " bases: the window uniqueness test is skipped" << std::endl;
        return;
    }
    std::vector<const char*> cs; std::vector<int64_t> cl;
    for (const std::string& c : chroms) { cs.push_back(c.data()); cl.push_back((int64_t)c.size()); }
    // The restriction to the window starts the reference looks up (current_mip_start of mipgen.cpp:808-813) happens on the device, and only the
    // regions that have a flagged start at all (a few percent) get a table: the K x 112 M bytes of an exome's flag image stay in HBM
    std::vector<const char*> rs; std::vector<int32_t> rl; std::vector<mipgen_window_bounds> bounds;
    for (int i : idx) {
        Region& r = regs[(size_t)i];
        rs.push_back(r.seq.data()); rl.push_back((int32_t)r.seq.size());
        bounds.push_back(mipgen_window_bounds{r.start_fl, r.stop_fl, r.seq_start, r.seq_stop});
        r.unmappable.clear();
    }
    std::vector<uint8_t> any((size_t)nr, 0);
    if (mipgen_accel_window_uniqueness_begin(h, (int32_t)cs.size(), cs.data(), cl.data(), nr, rs.data(), rl.data(), bounds.data(), K, sizes.data(), seed, any.data())) {
        std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl;
        throw 11;
    }
    for (int k = 0; k < nr; k++) {
        if (!any[(size_t)k]) continue;
        Region& r = regs[(size_t)idx[(size_t)k]];
        r.unmappable.assign((size_t)K * r.seq.size(), 0);
        if (mipgen_accel_window_flags_region(h, k, r.unmappable.data())) {
            std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl;
            (void)mipgen_accel_window_uniqueness_end(h);
            throw 11;
        }
    }
    (void)mipgen_accel_window_uniqueness_end(h);
}

// the counts as host tables (Region::copy_flat): for callers that score through their own accelerator handle
void gpu_copy_numbers(const Options& o, const std::vector<std::string>& chroms, std::vector<Region>& regs, int r0, int r1, int device)
{
    if (r1 < 0) r1 = (int)regs.size();                                // (the default: every region)
    if (r1 <= r0) return;
    std::vector<const char*> cs; std::vector<int64_t> cl;
    for (const std::string& c : chroms) { cs.push_back(c.data()); cl.push_back((int64_t)c.size()); }
    std::vector<int32_t> lengths(o.oligo_sizes.begin(), o.oligo_sizes.end());
    std::vector<const char*> rs; std::vector<int32_t> rl; std::vector<int32_t*> outp;
    for (int i = r0; i < r1; i++) {
        Region& r = regs[(size_t)i];
        rs.push_back(r.seq.data()); rl.push_back((int32_t)r.seq.size());
        r.copy_flat.assign(lengths.size() * r.seq.size(), 0);
        outp.push_back(r.copy_flat.data());
    }
    mipgen_params ap = o.accel_params();
    mipgen_accel* h = nullptr;
    if (mipgen_accel_create(&ap, device, nullptr, &h)) { std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl; throw 17; }
    const int rc = mipgen_accel_count_oligo_copies(h, (int32_t)cs.size(), cs.data(), cl.data(), r1 - r0, rs.data(), rl.data(),
                                                   (int32_t)lengths.size(), lengths.data(), outp.data());
    if (rc) { std::cerr << "[mipgen] accelerator: " << mipgen_accel_last_error() << std::endl; mipgen_accel_destroy(h); throw 11; }
    std::vector<int> idx;
    for (int i = r0; i < r1; i++) idx.push_back(i);
    try { gpu_window_flags(o, h, chroms, regs, idx); } catch (...) { mipgen_accel_destroy(h); throw; }
    mipgen_accel_destroy(h);
    for (int i = r0; i < r1; i++) regs[(size_t)i].copy_ready = true;
}

void fill_accel_region(const Region& r, mipgen_region& out, bool resident_copies)
{
    memset(&out, 0, sizeof out);
    out.start_flanked = r.start_fl; out.stop_flanked = r.stop_fl;
    out.seq_start = r.seq_start; out.seq_stop = r.seq_stop; out.seq_len = (int)r.seq.size();
    out.seq = r.seq.c_str();
    out.masked_seq = r.masked.size() == r.seq.size() ? r.masked.c_str() : nullptr;
    out.copy = resident_copies ? MIPGEN_COPY_RESIDENT : (r.copy_ptr.empty() ? nullptr : r.copy_ptr.data());
    out.unmappable = r.unmappable.empty() ? nullptr : r.unmappable.data();
    out.snp_class = r.snp_class.empty() ? nullptr : r.snp_class.data();
    memcpy(out.long_range_content, r.lrc, sizeof out.long_range_content);
}

}  // namespace mipgen
