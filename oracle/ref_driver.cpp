// Function-level driver for the REAL reference classes.  Test infrastructure only.
//
// This file is OUR code.  oracle/Makefile compiles it together with the reference's own
// SVMipv4.cpp / PlusSVMipv4.cpp / MinusSVMipv4.cpp / Featurev5.cpp / svm.cpp *where they lie* under
// /root/reference into oracle/_ref/libmipgen_refdrv.so.  It exposes the reference's per-candidate
// arithmetic through a C ABI so tests can pin the C restatement (mipgen_oracle.c) and the HIP path
// against full-precision (%.17g-grade) reference values: the CLI prints scores with only six
// significant digits (/root/reference/mipgen.cpp:774), too coarse for the 1e-5 gate.
//
// Sequences are passed as FORWARD-strand genomic substrings, exactly what tile_regions/design_mip hand
// to the setters (/root/reference/mipgen.cpp:461-462,602-603); the Minus class reverse-complements them
// itself (/root/reference/MinusSVMipv4.cpp:40-48).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "Featurev5.h"
#include "SVMipv4.h"
#include "PlusSVMipv4.h"
#include "MinusSVMipv4.h"
#include "svm.h"

// mipgen.cpp:33 defines this static member inside the monolith; a stand-alone link needs it too.
map<string, double> SVMipv4::junction_scores;

// The 44 long-range mers are a parameter table of the model (mipgen.cpp:32), restated as data.
static string k_feature_mers[44] = {"A","AA","AAA","AAC","AAG","AAT","AC","ACA","ACC","ACG","AG","AGA","AGC","AGG","AGT",
    "AT","ATA","ATC","ATG","CAG","CG","CGG","G","GAC","GAG","GC","GCG","GG","GGC","GGG","GTG","TA","TAA","TAC","TAG",
    "TC","TCC","TCG","TG","TGA","TGC","TGG","TTC","TTG"};

static SVMipv4* make_mip(int strand, const char* ext_fwd, const char* lig_fwd, const char* ins_fwd,
                         int ext_copy, int lig_copy, const char* middle)
{
    static bool init = false;
    if (!init) { SVMipv4::set_junction_scores(); init = true; }
    int e = (int)strlen(ext_fwd), l = (int)strlen(lig_fwd), n = (int)strlen(ins_fwd);
    // coordinates are irrelevant to the arithmetic; use scan_start = 1000
    SVMipv4* m;
    if (strand == 0) {
        PlusSVMipv4* p = new PlusSVMipv4("1", 1000, 1000 + n - 1, e, l);
        p->set_scan_target_seq(ins_fwd);
        m = p;
    } else {
        MinusSVMipv4* q = new MinusSVMipv4("1", 1000, 1000 + n - 1, e, l);
        q->set_scan_target_seq(ins_fwd);
        m = q;
    }
    m->set_ext_probe_seq(ext_fwd);
    m->set_lig_probe_seq(lig_fwd);
    m->mip_seq = m->lig_probe_sequence + string(middle) + m->ext_probe_sequence;   // mipgen.cpp:605
    m->ext_probe_copy = ext_copy;
    m->lig_probe_copy = lig_copy;
    return m;
}

extern "C" {

// SVMipv4::get_score (SVMipv4.cpp:114-248)
double ref_logistic(int strand, const char* ext_fwd, const char* lig_fwd, const char* ins_fwd,
                    int ext_copy, int lig_copy, const char* middle)
{
    SVMipv4* m = make_mip(strand, ext_fwd, lig_fwd, ins_fwd, ext_copy, lig_copy, middle);
    double s = m->get_score();
    delete m;
    return s;
}

// SVMipv4::get_parameters (SVMipv4.cpp:60-113); returns the vector length (192)
int ref_parameters(int strand, const char* ext_fwd, const char* lig_fwd, const char* ins_fwd,
                   int ext_copy, int lig_copy, const char* middle, const double* lrc44, double* out192)
{
    SVMipv4* m = make_mip(strand, ext_fwd, lig_fwd, ins_fwd, ext_copy, lig_copy, middle);
    vector<double> v;
    double lrc[44];
    memcpy(lrc, lrc44, sizeof(lrc));
    m->get_parameters(v, lrc);
    for (size_t i = 0; i < v.size() && i < 192; i++) out192[i] = v[i];
    int n = (int)v.size();
    delete m;
    return n;
}

// oriented sequences as the object holds them (for pinning reverse_comp, MinusSVMipv4.cpp:6-29)
void ref_oriented(int strand, const char* ext_fwd, const char* lig_fwd, const char* ins_fwd,
                  char* ext_out, char* lig_out, char* ins_out, char* junction_out)
{
    SVMipv4* m = make_mip(strand, ext_fwd, lig_fwd, ins_fwd, 1, 1, "");
    strcpy(ext_out, m->ext_probe_sequence.c_str());
    strcpy(lig_out, m->lig_probe_sequence.c_str());
    strcpy(ins_out, m->scan_target_sequence.c_str());
    strcpy(junction_out, m->ligation_junction.c_str());
    delete m;
}

// Featurev5::get_long_range_content (Featurev5.cpp:18-56)
void ref_long_range_content(const char* extended_seq, int chrom_seq_start, int chrom_seq_stop, double* out44)
{
    Featurev5 f("1", 1, 2, 0, "x");
    f.chromosomal_sequence_start_position = chrom_seq_start;
    f.chromosomal_sequence_stop_position = chrom_seq_stop;
    f.get_long_range_content(string(extended_seq), k_feature_mers);
    for (int i = 0; i < 44; i++) out44[i] = f.long_range_content[i];
}

void* ref_svm_load_model(const char* path) { return (void*)svm_load_model(path); }
int ref_svm_nsv(void* model) { return model ? ((svm_model*)model)->l : -1; }
double ref_svm_gamma(void* model) { return ((svm_model*)model)->param.gamma; }
double ref_svm_rho(void* model) { return ((svm_model*)model)->rho[0]; }
void ref_svm_free_model(void* model) { svm_model* m = (svm_model*)model; if (m) svm_free_and_destroy_model(&m); }

// The ten meaningful lines of mipgen::predict_value (mipgen.cpp:1948-2019), which is a private member of
// the monolith and cannot be linked: every index 1..n present (zeros included), terminator -1, svm_predict.
double ref_predict_dense(void* model, const double* x, int n)
{
    svm_node* nodes = (svm_node*)malloc((n + 1) * sizeof(svm_node));
    for (int i = 0; i < n; i++) { nodes[i].index = i + 1; nodes[i].value = x[i]; }
    nodes[n].index = -1;
    double s = svm_predict((svm_model*)model, nodes);
    free(nodes);
    return s;
}

// Same, but through the text hop the reference actually takes: each double printed with 17 significant
// digits (Boost 1.55 lexical_cast<string>(double): boost/detail/lcast_precision.hpp:81-93) and re-parsed
// with strtod (mipgen.cpp:1964,2008).  Tests use this to show the hop is lossless.
double ref_predict_text(void* model, const double* x, int n)
{
    svm_node* nodes = (svm_node*)malloc((n + 1) * sizeof(svm_node));
    char buf[64];
    for (int i = 0; i < n; i++) {
        snprintf(buf, sizeof buf, "%.17g", x[i]);
        nodes[i].index = i + 1;
        nodes[i].value = strtod(buf, NULL);
    }
    nodes[n].index = -1;
    double s = svm_predict((svm_model*)model, nodes);
    free(nodes);
    return s;
}

// A model trained AND written by the reference's own libsvm (svm_train, svm.cpp:2095; svm_save_model, svm.cpp:2644-2757): epsilon-SVR with an
// RBF kernel on n dense 192-feature rows (zeros omitted from the sparse nodes, as libsvm's own file readers produce them).  The file it leaves is
// genuine svm_save_model output - header keys, "%.8g"-style values, only the non-zero indices of every support vector - for pinning every
// loader (svm_load_model's grammar, svm.cpp:2779-2962) against.  Returns the number of support vectors, -1 on failure.
static void quiet_print(const char*) {}
int ref_svm_train_save(int n, const double* x, const double* y, double gamma, double cost, double epsilon, const char* out_path)
{
    svm_set_print_string_function(&quiet_print);
    svm_problem prob;
    prob.l = n;
    prob.y = (double*)malloc(sizeof(double) * n);
    prob.x = (svm_node**)malloc(sizeof(svm_node*) * n);
    std::vector<svm_node> pool;
    pool.reserve((size_t)n * 193);
    std::vector<size_t> first((size_t)n);
    for (int i = 0; i < n; i++) {
        prob.y[i] = y[i];
        first[(size_t)i] = pool.size();
        for (int j = 0; j < 192; j++) {
            const double v = x[(size_t)i * 192 + j];
            if (v != 0.0) { svm_node nd; nd.index = j + 1; nd.value = v; pool.push_back(nd); }
        }
        svm_node end; end.index = -1; end.value = 0.0; pool.push_back(end);
    }
    for (int i = 0; i < n; i++) prob.x[i] = &pool[first[(size_t)i]];
    svm_parameter par;
    memset(&par, 0, sizeof par);
    par.svm_type = EPSILON_SVR; par.kernel_type = RBF; par.degree = 3; par.gamma = gamma; par.coef0 = 0;
    par.cache_size = 100; par.eps = 1e-3; par.C = cost; par.nr_weight = 0; par.nu = 0.5; par.p = epsilon; par.shrinking = 1; par.probability = 0;
    const char* err = svm_check_parameter(&prob, &par);
    int nsv = -1;
    if (!err) {
        svm_model* m = svm_train(&prob, &par);
        if (m) {
            if (svm_save_model(out_path, m) == 0) nsv = m->l;
            svm_free_and_destroy_model(&m);
        }
    }
    free(prob.y); free(prob.x);
    return nsv;
}

}  // extern "C"
