/*
 * mipgen_oracle.c — CPU restatement (plain C11) of MIPgen's candidate enumeration + scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mipgen_oracle.h).  Each function cites the reference file:line it follows
 * (paths are under /root/reference).  Floating-point expressions keep the reference's evaluation order so
 * that, built with the same compiler family at -O2 without FP contraction, results are bit-identical to the
 * compiled reference on this CPU; the tests assert that.
 */
#define _GNU_SOURCE
#include "mipgen_oracle.h"
#include "../include/mipgen_logistic_model.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* strings                                                                                          */
/* ------------------------------------------------------------------------------------------------ */

/* reverse_comp, MinusSVMipv4.cpp:6-29 */
void mo_reverse_comp(const char* in, int n, char* out)
{
    int k = 0;
    for (int i = n - 1; i >= 0; i--) {
        char c = in[i];
        switch (c) {
            case 'G': c = 'C'; break;
            case 'C': c = 'G'; break;
            case 'A': c = 'T'; break;
            case 'T': c = 'A'; break;
            default: break;               /* N and anything else pass through (:24-25) */
        }
        out[k++] = c;
    }
    out[k] = '\0';
}

/* overlapping occurrences: for(offset = s.find(sub); ...; offset = s.find(sub, offset+1)) count++
 * SVMipv4.cpp:31-57, Featurev5.cpp:25-28 */
static double count_mer(const char* s, const char* sub)
{
    double count = 0;
    const char* p = strstr(s, sub);
    while (p) {
        count++;
        p = strstr(p + 1, sub);
    }
    return count;
}

static int count_char(const char* s, char c)
{
    int n = 0;
    for (; *s; s++) n += (*s == c);
    return n;
}

/* guard shared by get_parameters (SVMipv4.cpp:63) and get_score (:116):
 * ext.find("N") < e || lig.find("N") < l || mip_seq.find("-") < mip_seq.length() */
static int guard_trips(const char* ext, const char* lig, const char* mip_seq)
{
    if (strchr(ext, 'N') || strchr(lig, 'N')) return 1;
    if (mip_seq) return strchr(mip_seq, '-') != NULL;
    return strchr(ext, '-') != NULL || strchr(lig, '-') != NULL;
}

static int base_code(char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

static int junction_code(const char* lig)
{
    if (!lig[0] || !lig[1]) return 255;
    int a = base_code(lig[0]), b = base_code(lig[1]);
    if (a < 0 || b < 0) return 255;
    return 4 * a + b;
}

/* log10 of an int copy number with the reference's clamp, SVMipv4.cpp:109-110,173-174 */
static double log_copy(int copy)
{
    return copy > 100 ? 2 : log10((double)copy);
}

/* ------------------------------------------------------------------------------------------------ */
/* SVMipv4::get_parameters, SVMipv4.cpp:60-113                                                      */
/* ------------------------------------------------------------------------------------------------ */

static const char* k_arm_mers[20] = {"A","AA","AC","AG","AT","C","CA","CC","CG","CT","G","GA","GC","GG","GT","T","TA","TC","TG","TT"};
static const char* k_junctions[16] = {"AA","AC","AG","AT","CA","CC","CG","CT","GA","GC","GG","GT","TA","TC","TG","TT"};

/* the 84 insert mers are all 1-, 2-, 3-mers in lexicographic order (SVMipv4.cpp:70) */
static void insert_mer(int idx, char* out)
{
    /* enumerate: for b0: [b0], for b1: [b0 b1], for b2: [b0 b1 b2] */
    static const char B[4] = {'A','C','G','T'};
    int per_b0 = 1 + 4 * (1 + 4);      /* 21 */
    int b0 = idx / per_b0, r = idx % per_b0;
    out[0] = B[b0];
    if (r == 0) { out[1] = 0; return; }
    r -= 1;
    int b1 = r / 5, r2 = r % 5;
    out[1] = B[b1];
    if (r2 == 0) { out[2] = 0; return; }
    out[2] = B[r2 - 1];
    out[3] = 0;
}

void mo_get_parameters(const char* ext, const char* lig, const char* ins, const char* mip_seq,
                       int ext_copy, int lig_copy, const double* lrc44, double* out)
{
    int k = 0;
    int e = (int)strlen(ext), l = (int)strlen(lig), n = (int)strlen(ins);
    if (guard_trips(ext, lig, mip_seq)) {           /* :63-68 */
        for (int i = 0; i < MIPGEN_N_FEATURES; i++) out[i] = 0;
        return;
    }
    for (int i = 0; i < 20; i++) {                  /* extension, :72-79 */
        size_t ml = strlen(k_arm_mers[i]);
        if (strcmp(k_arm_mers[i], "T") == 0)
            out[k++] = (count_mer(ext, "G") + count_mer(ext, "C")) / (double)(size_t)((size_t)e - ml + 1);
        out[k++] = count_mer(ext, k_arm_mers[i]) / ((size_t)e - ml + 1.);
    }
    out[k++] = e;                                   /* :80 */
    for (int i = 0; i < MIPGEN_N_LRC; i++) out[k++] = lrc44[i];      /* :81-84 */
    for (int i = 0; i < 84; i++) {                  /* insert, :85-92 */
        char mer[4];
        insert_mer(i, mer);
        size_t ml = strlen(mer);
        if (strcmp(mer, "T") == 0)
            out[k++] = (count_mer(ins, "G") + count_mer(ins, "C")) / ((size_t)n - ml + 1.);
        out[k++] = count_mer(ins, mer) / ((size_t)n - ml + 1.);
    }
    out[k++] = n;                                   /* :93 (scan_size) */
    for (int i = 0; i < 20; i++) {                  /* ligation, :94-101 */
        size_t ml = strlen(k_arm_mers[i]);
        if (strcmp(k_arm_mers[i], "T") == 0)
            out[k++] = (count_mer(lig, "G") + count_mer(lig, "C")) / ((size_t)l - ml + 1.);
        out[k++] = count_mer(lig, k_arm_mers[i]) / ((size_t)l - ml + 1.);
    }
    out[k++] = l;                                   /* :102 */
    char lj[3] = {0, 0, 0};
    strncpy(lj, lig, 2);                            /* :103 */
    for (int i = 0; i < 16; i++) out[k++] = strcmp(lj, k_junctions[i]) == 0 ? 1 : 0;   /* :104-107 */
    out[k++] = log_copy(ext_copy);                  /* :109-112 */
    out[k++] = log_copy(lig_copy);
}

/* ------------------------------------------------------------------------------------------------ */
/* SVMipv4::get_score, SVMipv4.cpp:114-248                                                          */
/* ------------------------------------------------------------------------------------------------ */

static const mipgen_logistic_term k_terms[MIPGEN_LOGISTIC_NTERMS] = MIPGEN_LOGISTIC_TERMS;
static const double k_junction_scores[16] = MIPGEN_JUNCTION_SCORES;

static void fill_ints(const char* ext, const char* lig, const char* ins, int run_count,
                      int ext_copy, int lig_copy, mipgen_candidate_ints* ints)
{
    if (!ints) return;
    ints->ext_a = count_char(ext, 'A'); ints->ext_c = count_char(ext, 'C');
    ints->ext_g = count_char(ext, 'G'); ints->ext_t = count_char(ext, 'T');
    ints->lig_a = count_char(lig, 'A'); ints->lig_c = count_char(lig, 'C');
    ints->lig_g = count_char(lig, 'G'); ints->lig_t = count_char(lig, 'T');
    ints->ins_a = count_char(ins, 'A'); ints->ins_c = count_char(ins, 'C');
    ints->ins_g = count_char(ins, 'G'); ints->ins_t = count_char(ins, 'T');
    ints->run_count = run_count;
    ints->junction = junction_code(lig);
    ints->ext_copy = ext_copy; ints->lig_copy = lig_copy;
    ints->scan_size = (int)strlen(ins);
}

/* GC/AT run counter, SVMipv4.cpp:118-142 (returns run_count after the final ++) */
static int run_counter(const char* ins, int n)
{
    char last = ins[0];
    int run = 0;
    for (int i = 1; i < n; i++) {
        char cur = ins[i];
        if (cur == 'G' || cur == 'C') {
            if (last == 'G' || last == 'C') { }
            else { run++; last = cur; }
        } else {
            if (last == 'A' || last == 'T') { }
            else { run++; last = cur; }
        }
    }
    return run + 1;
}

double mo_get_score(const char* ext, const char* lig, const char* ins, const char* mip_seq,
                    int ext_copy, int lig_copy, mipgen_candidate_ints* ints)
{
    int n = (int)strlen(ins);
    int guard = guard_trips(ext, lig, mip_seq);
    int run = n > 0 ? run_counter(ins, n) : 1;
    fill_ints(ext, lig, ins, run, ext_copy, lig_copy, ints);
    if (guard) return -1000.0;                      /* :116 */

    double v[MLV_COUNT];
    double run_count = run;
    double scan_size = n;
    double ext_g = count_char(ext, 'G'), lig_g = count_char(lig, 'G'), tgt_g = count_char(ins, 'G');
    double ext_gc = (double)count_char(ext, 'C') + ext_g;
    double lig_gc = (double)count_char(lig, 'C') + lig_g;
    double tgt_gc = (double)count_char(ins, 'C') + tgt_g;
    double ext_a = count_char(ext, 'A'), lig_a = count_char(lig, 'A'), tgt_a = count_char(ins, 'A');
    double ext_len = (double)strlen(ext), lig_len = (double)strlen(lig);
    v[MLV_BPS] = scan_size / run_count;             /* :143; scan_size is an int there, promoted */
    v[MLV_TLEN] = n > 250 ? 250 : n;                /* :157 */
    v[MLV_ELEN] = ext_len; v[MLV_LLEN] = lig_len;
    v[MLV_EGC] = ext_gc / ext_len; v[MLV_LGC] = lig_gc / lig_len; v[MLV_TGC] = tgt_gc / n;   /* :159-161 */
    v[MLV_EG] = ext_g / ext_len;  v[MLV_LG] = lig_g / lig_len;   v[MLV_TG] = tgt_g / n;      /* :163-165 */
    v[MLV_EA] = ext_a / ext_len;  v[MLV_LA] = lig_a / lig_len;   v[MLV_TA] = tgt_a / n;      /* :167-169 */
    int jc = junction_code(lig);
    v[MLV_JS] = jc < 16 ? k_junction_scores[jc] : 0.0;                                        /* :171 */
    v[MLV_LEC] = log_copy(ext_copy); v[MLV_LLC] = log_copy(lig_copy);                        /* :173-174 */

    double exponent = MIPGEN_LOGISTIC_C0 - MIPGEN_LOGISTIC_C1;                               /* :177 */
    for (int i = 0; i < MIPGEN_LOGISTIC_NTERMS; i++) {
        const mipgen_logistic_term* t = &k_terms[i];
        double term;
        if (t->kind == MLT_LIN) term = t->coef * v[t->v1];
        else if (t->kind == MLT_BIL) term = t->coef * v[t->v1] * v[t->v2];
        else term = t->coef * pow(v[t->v1], 2);
        exponent = exponent + term;
    }
    return pow(MIPGEN_LOGISTIC_BASE, exponent) / (1 + pow(MIPGEN_LOGISTIC_BASE, exponent));  /* :247 */
}

/* ------------------------------------------------------------------------------------------------ */
/* Featurev5::get_long_range_content, Featurev5.cpp:18-56; mers mipgen.cpp:32                        */
/* ------------------------------------------------------------------------------------------------ */

static const char* k_feature_mers[MIPGEN_N_LRC] = MIPGEN_FEATURE_MERS;

void mo_long_range_content(const char* seq, int cs_start, int cs_stop, double* out)
{
    for (int i = 0; i < MIPGEN_N_LRC; i++) {
        const char* mer = k_feature_mers[i];
        double forward_count = count_mer(seq, mer);
        int ml = (int)strlen(mer);
        char rc[8];
        int k = 0;
        for (int j = ml - 1; j >= 0; j--) {          /* :32-41: only ACGT are appended */
            switch (mer[j]) {
                case 'G': rc[k++] = 'C'; break;
                case 'C': rc[k++] = 'G'; break;
                case 'A': rc[k++] = 'T'; break;
                case 'T': rc[k++] = 'A'; break;
                default: break;
            }
        }
        rc[k] = 0;
        if (strcmp(rc, mer) != 0) {
            double reverse_count = count_mer(seq, rc);
            out[i] = (forward_count + reverse_count) / (cs_stop - cs_start + 2001);          /* :49 */
        } else {
            out[i] = forward_count / (cs_stop - cs_start + 2001);                            /* :53 */
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* libsvm 3.17: svm_load_model (svm.cpp:2759-2973), svm_predict (svm.cpp:2504-2593, 329-368)          */
/* ------------------------------------------------------------------------------------------------ */

typedef struct { int index; double value; } mo_node;   /* svm.h:12-16 */

struct mo_model {
    int svm_type, kernel_type, degree;
    double gamma, coef0;
    int nr_class, l;
    double rho0;
    double* sv_coef;        /* sv_coef[0][l] */
    mo_node** SV;
    mo_node* x_space;
};

static const char* k_svm_types[] = {"c_svc", "nu_svc", "one_class", "epsilon_svr", "nu_svr", NULL};   /* svm.cpp:2631-2634 */
static const char* k_kernel_types[] = {"linear", "polynomial", "rbf", "sigmoid", "precomputed", NULL}; /* svm.cpp:2636-2639 */

static char* read_line(FILE* fp, char** buf, size_t* cap)
{
    ssize_t n = getline(buf, cap, fp);
    return n < 0 ? NULL : *buf;
}

mo_model* mo_svm_load_model(const char* path)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return NULL;                                         /* :2762 */
    mo_model* m = (mo_model*)calloc(1, sizeof(mo_model));
    char cmd[81];
    int ok = 1;
    while (ok) {                                                  /* header, :2779-2899 */
        if (fscanf(fp, "%80s", cmd) != 1) { ok = 0; break; }
        if (strcmp(cmd, "svm_type") == 0) {
            if (fscanf(fp, "%80s", cmd) != 1) { ok = 0; break; }
            int i; for (i = 0; k_svm_types[i]; i++) if (strcmp(k_svm_types[i], cmd) == 0) { m->svm_type = i; break; }
            if (!k_svm_types[i]) ok = 0;
        } else if (strcmp(cmd, "kernel_type") == 0) {
            if (fscanf(fp, "%80s", cmd) != 1) { ok = 0; break; }
            int i; for (i = 0; k_kernel_types[i]; i++) if (strcmp(k_kernel_types[i], cmd) == 0) { m->kernel_type = i; break; }
            if (!k_kernel_types[i]) ok = 0;
        } else if (strcmp(cmd, "degree") == 0) { if (fscanf(fp, "%d", &m->degree) != 1) ok = 0; }
        else if (strcmp(cmd, "gamma") == 0) { if (fscanf(fp, "%lf", &m->gamma) != 1) ok = 0; }
        else if (strcmp(cmd, "coef0") == 0) { if (fscanf(fp, "%lf", &m->coef0) != 1) ok = 0; }
        else if (strcmp(cmd, "nr_class") == 0) { if (fscanf(fp, "%d", &m->nr_class) != 1) ok = 0; }
        else if (strcmp(cmd, "total_sv") == 0) { if (fscanf(fp, "%d", &m->l) != 1) ok = 0; }
        else if (strcmp(cmd, "rho") == 0) {
            int n = m->nr_class * (m->nr_class - 1) / 2;
            for (int i = 0; i < n; i++) { double r; if (fscanf(fp, "%lf", &r) != 1) ok = 0; if (i == 0) m->rho0 = r; }
        } else if (strcmp(cmd, "label") == 0 || strcmp(cmd, "nr_sv") == 0) {
            for (int i = 0; i < m->nr_class; i++) { int d; if (fscanf(fp, "%d", &d) != 1) ok = 0; }
        } else if (strcmp(cmd, "probA") == 0 || strcmp(cmd, "probB") == 0) {
            int n = m->nr_class * (m->nr_class - 1) / 2;
            for (int i = 0; i < n; i++) { double d; if (fscanf(fp, "%lf", &d) != 1) ok = 0; }
        } else if (strcmp(cmd, "SV") == 0) {
            int c;
            while ((c = getc(fp)) != EOF && c != '\n') { }
            break;
        } else ok = 0;                                            /* unknown text, :2889-2899 */
    }
    if (!ok) { fclose(fp); free(m); return NULL; }

    /* SV lines: "coef idx:val idx:val ..." (:2936-2962) */
    long pos = ftell(fp);
    char* line = NULL; size_t cap = 0;
    size_t elements = 0;
    while (read_line(fp, &line, &cap)) for (char* p = line; *p; p++) if (*p == ':') elements++;
    elements += (size_t)m->l;
    fseek(fp, pos, SEEK_SET);
    m->sv_coef = (double*)calloc((size_t)(m->l > 0 ? m->l : 1), sizeof(double));
    m->SV = (mo_node**)calloc((size_t)(m->l > 0 ? m->l : 1), sizeof(mo_node*));
    m->x_space = (mo_node*)calloc(elements + 1, sizeof(mo_node));
    size_t j = 0;
    for (int i = 0; i < m->l; i++) {
        if (!read_line(fp, &line, &cap)) { m->l = i; break; }
        m->SV[i] = &m->x_space[j];
        char* save = NULL;
        char* p = strtok_r(line, " \t", &save);
        m->sv_coef[i] = p ? strtod(p, NULL) : 0.0;
        for (int k = 1; k < m->nr_class - 1; k++) strtok_r(NULL, " \t", &save);
        for (;;) {
            char* idx = strtok_r(NULL, ":", &save);
            char* val = strtok_r(NULL, " \t", &save);
            if (!val) break;
            m->x_space[j].index = (int)strtol(idx, NULL, 10);
            m->x_space[j].value = strtod(val, NULL);
            j++;
        }
        m->x_space[j++].index = -1;
    }
    free(line);
    fclose(fp);
    return m;
}

void mo_svm_free_model(mo_model* m)
{
    if (!m) return;
    free(m->sv_coef); free(m->SV); free(m->x_space); free(m);
}
int mo_svm_nsv(const mo_model* m) { return m->l; }
double mo_svm_gamma(const mo_model* m) { return m->gamma; }
double mo_svm_rho(const mo_model* m) { return m->rho0; }
int mo_svm_kernel_type(const mo_model* m) { return m->kernel_type; }
int mo_svm_svm_type(const mo_model* m) { return m->svm_type; }

int mo_svm_densify(const mo_model* m, double* sv, double* coef)
{
    int extra = 0;
    for (int i = 0; i < m->l; i++) {
        coef[i] = m->sv_coef[i];
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) sv[(size_t)i * MIPGEN_N_FEATURES + j] = 0.0;
        for (const mo_node* p = m->SV[i]; p->index != -1; p++) {
            if (p->index >= 1 && p->index <= MIPGEN_N_FEATURES) sv[(size_t)i * MIPGEN_N_FEATURES + p->index - 1] = p->value;
            else extra++;
        }
    }
    return extra;
}

/* Kernel::k_function RBF branch, svm.cpp:329-368: sparse merge walk */
static double k_rbf(const mo_node* x, const mo_node* y, double gamma)
{
    double sum = 0;
    while (x->index != -1 && y->index != -1) {
        if (x->index == y->index) {
            double d = x->value - y->value;
            sum += d * d;
            ++x; ++y;
        } else if (x->index > y->index) {
            sum += y->value * y->value;
            ++y;
        } else {
            sum += x->value * x->value;
            ++x;
        }
    }
    while (x->index != -1) { sum += x->value * x->value; ++x; }
    while (y->index != -1) { sum += y->value * y->value; ++y; }
    return exp(-gamma * sum);
}

/* predict_value (mipgen.cpp:1948-2019) builds nodes 1..192 (all present, zeros included; the %.17g text
 * hop is lossless) and calls svm_predict -> svm_predict_values SVR branch (svm.cpp:2507-2522). */
double mo_predict_value(const mo_model* m, const double* x192)
{
    mo_node x[MIPGEN_N_FEATURES + 1];
    for (int i = 0; i < MIPGEN_N_FEATURES; i++) { x[i].index = i + 1; x[i].value = x192[i]; }
    x[MIPGEN_N_FEATURES].index = -1;
    double sum = 0;
    for (int i = 0; i < m->l; i++) sum += m->sv_coef[i] * k_rbf(x, m->SV[i], m->gamma);
    sum -= m->rho0;
    return sum;
}

/* ------------------------------------------------------------------------------------------------ */
/* level 2: candidate construction + design_mip                                                     */
/* ------------------------------------------------------------------------------------------------ */

/* std::string::substr(pos, len) on the region string; returns 0 if pos is outside (the reference would throw) */
static int region_substr(const mipgen_region* R, const char* s, int pos0, int len, char* out)
{
    if (pos0 < 0 || pos0 > R->seq_len) return 0;
    int n = len;
    if (pos0 + n > R->seq_len) n = R->seq_len - pos0;
    if (n < 0) n = 0;
    memcpy(out, s + pos0, (size_t)n);
    out[n] = 0;
    return 1;
}

static int copy_lookup(const mipgen_region* R, int start, int len)
{
    if (!R->copy) return 1;                          /* "every oligo has copy 1" */
    if (len < 0 || len > MIPGEN_MAX_OLIGO || !R->copy[len]) return 0;
    int i = start - R->seq_start;
    if (i < 0 || i >= R->seq_len) return 0;          /* absent key -> 0, mipgen.cpp:612-613 */
    return R->copy[len][i];
}

static int size_index(const mipgen_params* P, int C)
{
    int inc = P->capture_increment ? P->capture_increment : 1;
    return (P->max_capture_size - C) / inc;
}

static char comp_base(char c)
{
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; default: return c; }
}

int mo_design(const mipgen_params* P, const mipgen_region* R, const mipgen_candidate* c,
              const char* alleles, mo_designed* d)
{
    memset(d, 0, sizeof(*d));
    int p = c->scan_start, C = c->capture_size, e = c->ext_len, l = c->lig_len;
    int S = e + l;
    /* bounds skips, mipgen.cpp:443-444 */
    if (p - e <= 0 || p - l <= 0) return 1;
    if (p + C - e - 1 > R->seq_stop || p + C - l - 1 > R->seq_stop) return 1;

    d->scan_start = p;
    d->scan_stop = p + C - S - 1;                    /* :449,456 */
    d->scan_size = d->scan_stop - d->scan_start + 1; /* SVMipv4.cpp:27 */
    if (c->strand == 0) {                            /* PlusSVMipv4.cpp:9-13 */
        d->ext_start = p - e; d->ext_stop = p - 1;
        d->lig_start = d->scan_stop + 1; d->lig_stop = d->scan_stop + l;
    } else {                                         /* MinusSVMipv4.cpp:32-36 */
        d->ext_start = d->scan_stop + 1; d->ext_stop = d->scan_stop + e;
        d->lig_start = p - l; d->lig_stop = p - 1;
    }
    d->mapping_failed = '0'; d->snp_failed = '0'; d->masking_failed = '0'; d->has_snp_mip = 0;

    char tmp[MO_MAX_INSERT];
    const char* masked = R->masked_seq ? R->masked_seq : R->seq;
    /* insert, mipgen.cpp:461-462 */
    if (d->scan_size >= (int)sizeof(tmp) || d->scan_size < 0) return 1;
    region_substr(R, R->seq, d->scan_start - R->seq_start, d->scan_size, tmp);
    if (c->strand == 0) strcpy(d->ins_seq, tmp); else mo_reverse_comp(tmp, (int)strlen(tmp), d->ins_seq);
    /* arms, mipgen.cpp:602-603 */
    region_substr(R, R->seq, d->ext_start - R->seq_start, e, tmp);
    if (c->strand == 0) strcpy(d->ext_seq, tmp); else mo_reverse_comp(tmp, (int)strlen(tmp), d->ext_seq);
    region_substr(R, R->seq, d->lig_start - R->seq_start, l, tmp);
    if (c->strand == 0) strcpy(d->lig_seq, tmp); else mo_reverse_comp(tmp, (int)strlen(tmp), d->lig_seq);
    strncpy(d->junction, d->lig_seq, 2); d->junction[2] = 0;

    /* masked fraction, mipgen.cpp:606-610 */
    char m1[MIPGEN_MAX_OLIGO + 1], m2[MIPGEN_MAX_OLIGO + 1];
    region_substr(R, masked, d->ext_start - R->seq_start, e, m1);
    region_substr(R, masked, d->lig_start - R->seq_start, l, m2);
    double ext_N = count_char(m1, 'N'), lig_N = count_char(m2, 'N');
    d->arm_fraction_masked = (ext_N + lig_N) / (l + e);
    d->masked_n = (int)(ext_N + lig_N);

    /* copies, mipgen.cpp:612-613 */
    d->ext_copy = copy_lookup(R, d->ext_start, e);
    d->lig_copy = copy_lookup(R, d->lig_start, l);

    /* mapping flag + early return, mipgen.cpp:615-625 */
    if (R->unmappable && P->check_copy_number) {
        int k = size_index(P, C);
        int mip_start = c->strand == 0 ? d->ext_start : d->lig_start;   /* get_mip_start */
        int i = mip_start - R->seq_start;
        if (i >= 0 && i < R->seq_len && R->unmappable[(size_t)k * R->seq_len + i]) {
            d->mapping_failed = '1';
            /* the reference returns here with masking_failed never assigned (uninitialised char);
             * this restatement reports '0' for it. */
            return 0;
        }
    }
    d->masking_failed = d->arm_fraction_masked > P->masked_arm_threshold ? '1' : '0';   /* :626-633 */

    /* SNP scan, mipgen.cpp:634-760 */
    strcpy(d->snp_ext_seq, d->ext_seq);
    strcpy(d->snp_lig_seq, d->lig_seq);
    for (int arm = 0; arm < 2; arm++) {
        int a0 = arm == 0 ? d->ext_start : d->lig_start;
        int a1 = arm == 0 ? d->ext_stop : d->lig_stop;
        char* arm_seq = arm == 0 ? d->ext_seq : d->lig_seq;
        for (int i = a0; i <= a1; i++) {
            int ri = i - R->seq_start;
            if (ri < 0 || ri >= R->seq_len) continue;
            int cls = 0;
            const char* al = NULL;
            if (alleles) {
                al = alleles + 2 * (size_t)ri;
                if (al[0] == 0) continue;
                cls = -1;                              /* decide below from the alleles */
            } else if (R->snp_class) {
                cls = R->snp_class[ri];
                if (!cls) continue;
            } else continue;
            d->snp_count++;
            int flag = 0;
            if (cls == -1) {
                int rel = c->strand == 0 ? i - a0 : a1 - i;             /* :647-654 */
                if (al[0] != '*' && al[0] != 'N' && al[1] != 'N' && al[0] != '-' && al[1] != '-') {   /* :644 */
                    char bases[MIPGEN_MAX_OLIGO + 1];
                    strcpy(bases, arm_seq);
                    if (bases[rel] == al[0]) { bases[rel] = al[1]; flag = 1; }
                    else if (bases[rel] == comp_base(al[0])) { bases[rel] = comp_base(al[1]); flag = 1; }
                    if (flag) {
                        if (arm == 0) { strcpy(d->snp_ext_seq, bases); strcpy(d->snp_lig_seq, d->lig_seq); }
                        else { strcpy(d->snp_ext_seq, d->ext_seq); strcpy(d->snp_lig_seq, bases); }
                    }
                }
            } else flag = (cls == 1);
            if (flag) d->has_snp_mip = 1; else d->snp_failed = '1';
        }
    }
    if (d->snp_count > 1) d->snp_failed = '1';         /* :759-760 */
    return 0;
}

double mo_score_designed(const mo_designed* d, int method, const mo_model* m, const double* lrc44,
                         double* features192, mipgen_candidate_ints* ints)
{
    mipgen_candidate_ints local;
    double s = mo_get_score(d->ext_seq, d->lig_seq, d->ins_seq, NULL, d->ext_copy, d->lig_copy, ints ? ints : &local);
    if (ints) {
        ints->masked_n = d->masked_n;
        ints->snp_count = d->snp_count;
    }
    if (method == MIPGEN_SCORE_LOGISTIC && !features192) return s;
    double x[MIPGEN_N_FEATURES];
    mo_get_parameters(d->ext_seq, d->lig_seq, d->ins_seq, NULL, d->ext_copy, d->lig_copy, lrc44, x);
    if (features192) memcpy(features192, x, sizeof(x));
    if (method == MIPGEN_SCORE_LOGISTIC) return s;
    return mo_predict_value(m, x);
}

static uint32_t sat(uint32_t v, uint32_t hi) { return v > hi ? hi : v; }

uint64_t mo_record_of(const mo_designed* d, int valid, const mipgen_candidate_ints* ints)
{
    if (!valid) return 0;
    uint32_t flags = MIPGEN_FLAG_VALID;
    if (guard_trips(d->ext_seq, d->lig_seq, NULL)) flags |= MIPGEN_FLAG_GUARD;
    if (d->mapping_failed == '1') flags |= MIPGEN_FLAG_MAPPING;
    if (d->masking_failed == '1') flags |= MIPGEN_FLAG_MASKING;
    if (d->snp_failed == '1') flags |= MIPGEN_FLAG_SNP;
    if (d->has_snp_mip) flags |= MIPGEN_FLAG_HAS_SNP_MIP;
    uint32_t ec = d->ext_copy < 0 ? 0 : sat((uint32_t)d->ext_copy, 65535);
    uint32_t lc = d->lig_copy < 0 ? 0 : sat((uint32_t)d->lig_copy, 65535);
    uint32_t mn = (uint32_t)d->masked_n; (void)ints;
    uint32_t jc = (uint32_t)junction_code(d->lig_seq);
    return (uint64_t)ec | ((uint64_t)lc << 16) | ((uint64_t)sat(mn, 255) << 32) |
           ((uint64_t)sat((uint32_t)d->snp_count, 255) << 40) | ((uint64_t)flags << 48) | ((uint64_t)jc << 56);
}

/* ------------------------------------------------------------------------------------------------ */
/* level 3: dense grid, replay, condense                                                            */
/* ------------------------------------------------------------------------------------------------ */

static int n_sizes_all(const mipgen_params* P)
{
    int inc = P->capture_increment ? P->capture_increment : 1;
    if (P->max_capture_size < P->min_capture_size) return 0;
    return (P->max_capture_size - P->min_capture_size) / inc + 1;
}

static int max_arm_sum(const mipgen_params* P)          /* *arm_length_sum_set.rbegin(), mipgen.cpp:421 (a key may hold an empty list) */
{
    if (P->arm_sum_key_max > 0) return P->arm_sum_key_max;
    int m = 0;
    for (int i = 0; i < P->n_arm_pairs; i++) { int s = P->arm_ext[i] + P->arm_lig[i]; if (s > m) m = s; }
    return m;
}
static int min_arm_sum(const mipgen_params* P)          /* *arm_length_sum_set.begin(), mipgen.cpp:434 */
{
    if (P->arm_sum_key_min > 0) return P->arm_sum_key_min;
    int m = INT_MAX;
    for (int i = 0; i < P->n_arm_pairs; i++) { int s = P->arm_ext[i] + P->arm_lig[i]; if (s < m) m = s; }
    return m;
}

int mo_grid(const mipgen_params* P, const mipgen_region* R, mipgen_grid* g)
{
    int inc = P->capture_increment ? P->capture_increment : 1;
    int cur = R->start_flanked - P->max_capture_size + max_arm_sum(P);   /* mipgen.cpp:421 */
    if (cur < 0) cur = 0;                                                 /* :422 */
    g->offset = 0;
    g->first_pos = cur + 1;                                               /* :423-425 */
    g->n_pos = R->stop_flanked - cur;
    if (g->n_pos < 0) g->n_pos = 0;
    int K = n_sizes_all(P), k0 = 0;
    /* static skip, :429: C > (stop_fl - start_fl) + max_mip_overlap && C - inc >= min_capture */
    while (k0 < K) {
        int C = P->max_capture_size - k0 * inc;
        if (C > R->stop_flanked - R->start_flanked + P->max_mip_overlap && C - inc >= P->min_capture_size) k0++;
        else break;
    }
    g->first_size_index = k0;
    g->n_sizes = K - k0;
    g->count = (int64_t)g->n_pos * g->n_sizes * P->n_arm_pairs * 2;
    return 0;
}

int mo_score_region_dense(const mipgen_params* P, const mipgen_region* R, const mo_model* m, int method,
                          double* scores, uint64_t* records)
{
    mipgen_grid g;
    mo_grid(P, R, &g);
    int inc = P->capture_increment ? P->capture_increment : 1;
    int64_t idx = 0;
    for (int pi = 0; pi < g.n_pos; pi++)
        for (int ki = 0; ki < g.n_sizes; ki++)
            for (int s = 0; s < 2; s++)                       /* strand-major inside a (position, size) row */
                for (int a = 0; a < P->n_arm_pairs; a++, idx++) {
                    mipgen_candidate c = {0, g.first_pos + pi, P->max_capture_size - (g.first_size_index + ki) * inc,
                                          P->arm_ext[a], P->arm_lig[a], s};
                    mo_designed d;
                    if (mo_design(P, R, &c, NULL, &d)) { scores[idx] = 0.0; records[idx] = 0; continue; }
                    mipgen_candidate_ints ints;
                    scores[idx] = mo_score_designed(&d, method, m, R->long_range_content, NULL, &ints);
                    records[idx] = mo_record_of(&d, 1, &ints);
                }
    return 0;
}

/* (int) conversion of a double as x86-64 cvttsd2si performs it (mipgen.cpp:496-497 assign a double score to
 * an int; NaN and out-of-range values yield INT_MIN there) */
static int to_int_x86(double v)
{
    if (isnan(v) || v >= 2147483648.0 || v <= -2147483649.0) return INT_MIN;
    return (int)v;
}

int64_t mo_replay_region(const mipgen_params* P, const mipgen_region* R, const double* scores,
                         const uint64_t* records, uint8_t* emitted)
{
    mipgen_grid g;
    mo_grid(P, R, &g);
    int A = P->n_arm_pairs;
    int min_sum = min_arm_sum(P);
    int64_t n_emitted = 0;
    memset(emitted, 0, (size_t)g.count);
    for (int pi = 0; pi < g.n_pos; pi++) {
        double previous_best_score = 0;                                   /* :426 */
        for (int ki = 0; ki < g.n_sizes; ki++) {                          /* :427 (sizes failing :429 are not in the grid) */
            if (previous_best_score > P->upper_score_limit) continue;     /* :430 */
            int a = 0;
            while (a < A) {                                               /* :431 one arm-sum list at a time */
                int sum = P->arm_ext[a] + P->arm_lig[a];
                int a_end = a;
                while (a_end < A && P->arm_ext[a_end] + P->arm_lig[a_end] == sum) a_end++;
                if (previous_best_score > P->upper_score_limit && sum != min_sum) { a = a_end; continue; }   /* :434 */
                int previous_minus_score = 0, previous_plus_score = 0;     /* :435-436 */
                int skip_ahead = 0;
                for (; a < a_end; a++) {
                    if (skip_ahead) continue;                              /* :440 */
                    int64_t idx = MIPGEN_PLUS_INDEX(pi, g.n_sizes, ki, A, a), idm = MIPGEN_MINUS_INDEX(pi, g.n_sizes, ki, A, a);
                    if (!(MIPGEN_REC_FLAGS(records[idx]) & MIPGEN_FLAG_VALID)) continue;     /* :443-444 */
                    emitted[idx] = 1; emitted[idm] = 1;
                    n_emitted += 2;
                    double plus = scores[idx], minus = scores[idm];
                    if (P->score_method == MIPGEN_SCORE_LOGISTIC && P->logistic_heuristic &&
                        plus < previous_plus_score && minus < previous_minus_score) skip_ahead = 1;   /* :494 */
                    previous_best_score = (minus > plus) ? minus : plus;   /* :495 */
                    previous_minus_score = to_int_x86(minus);              /* :496 */
                    previous_plus_score = to_int_x86(plus);                /* :497 */
                }
            }
        }
    }
    return n_emitted;
}

int mo_condense_region(const mipgen_params* P, const mipgen_region* R, const double* scores,
                       const uint64_t* records, const uint8_t* emitted, mipgen_survivor* out)
{
    mipgen_grid g;
    mo_grid(P, R, &g);
    int A = P->n_arm_pairs;
    int64_t per_pos = (int64_t)g.n_sizes * A * 2;
    for (int pi = 0; pi < g.n_pos; pi++) {
        /* chosen_* are declared per position, not reset per strand (mipgen.cpp:1677-1680) */
        int chosen_copy_count = 0;
        double chosen_masked = 0;
        for (int s = 0; s < 2; s++) {
            mipgen_survivor* best = &out[2 * (int64_t)pi + s];
            best->cand_index = -1; best->score = 0; best->record = 0;
            int skip_ahead = 0;
            /* the list is newest-first (push_front, :475,489): walk the dense order backwards */
            for (int64_t j = per_pos / 2 - 1; j >= 0; j--) {                /* j = (size index, arm pair) in generation order */
                int64_t idx = (int64_t)pi * per_pos + ((j / A) * 2 + s) * A + (j % A);
                if (!emitted[idx]) continue;
                if (skip_ahead) continue;                                                     /* :1687 */
                uint64_t r = records[idx];
                int a = (int)(j % A);
                /* the reference compares bwa's unbounded X0 counts (mipgen.cpp:586-587,612-613), not the record's saturating
                 * 16-bit copies: take them from the copy table through the candidate's geometry */
                int e = P->arm_ext[a], l = P->arm_lig[a];
                int C = P->max_capture_size - (g.first_size_index + (int)(j / A)) * (P->capture_increment ? P->capture_increment : 1);
                int p = g.first_pos + pi, ss = C - e - l;
                int ext_copy = copy_lookup(R, s ? p + ss : p - e, e), lig_copy = copy_lookup(R, s ? p - l : p + ss, l);
                if ((int64_t)ext_copy * lig_copy > P->max_arm_copy_product) continue;         /* :1689 */
                if (MIPGEN_REC_FLAGS(r) & MIPGEN_FLAG_MAPPING) continue;                      /* :1690 */
                int cur_copy = ext_copy > lig_copy ? ext_copy : lig_copy;                     /* :1692 */
                double cur_masked = (double)MIPGEN_REC_MASKED_N(r) / (P->arm_lig[a] + P->arm_ext[a]);   /* :610,1693 */
                int snp = (int)MIPGEN_REC_SNP_COUNT(r);
                double sc = scores[idx];
                if (best->cand_index < 0) {                                                   /* :1695-1700 */
                    best->cand_index = idx; best->score = sc; best->record = r;
                    chosen_masked = cur_masked; chosen_copy_count = cur_copy;
                } else if (cur_masked > P->masked_arm_threshold && cur_masked < chosen_masked) {   /* :1701-1706 */
                    best->cand_index = idx; best->score = sc; best->record = r;
                    chosen_masked = cur_masked; chosen_copy_count = cur_copy;
                } else {
                    if (cur_copy > P->target_arm_copy && cur_copy < chosen_copy_count) {      /* :1709-1714 */
                        best->cand_index = idx; best->score = sc; best->record = r;
                        chosen_masked = cur_masked; chosen_copy_count = cur_copy;
                    } else if (cur_copy <= P->target_arm_copy) {
                        if (sc < P->lower_score_limit && sc > best->score) {                  /* :1717-1722 */
                            best->cand_index = idx; best->score = sc; best->record = r;
                            chosen_masked = cur_masked; chosen_copy_count = cur_copy;
                        } else if (sc > P->lower_score_limit) {
                            int bsnp = (int)MIPGEN_REC_SNP_COUNT(best->record);
                            if (snp < bsnp) {                                                 /* :1725-1730 */
                                best->cand_index = idx; best->score = sc; best->record = r;
                                chosen_masked = cur_masked; chosen_copy_count = cur_copy;
                            } else if (snp == bsnp) {
                                if (sc > best->score) {                                       /* :1733-1737 */
                                    best->cand_index = idx; best->score = sc; best->record = r;
                                    if (sc > P->upper_score_limit) skip_ahead = 1;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* level 4: literal tile_regions loop with lazy scoring, mipgen.cpp:412-501                          */
/* ------------------------------------------------------------------------------------------------ */

int64_t mo_enumerate_region(const mipgen_params* P, const mipgen_region* R, const mo_model* m, int method,
                            const char* alleles, mo_emitted* out, int64_t capacity)
{
    int inc = P->capture_increment ? P->capture_increment : 1;
    int A = P->n_arm_pairs;
    int min_sum = min_arm_sum(P);
    mipgen_grid g;
    mo_grid(P, R, &g);
    int64_t n = 0;
    int cur = g.first_pos - 1;
    while (cur < R->stop_flanked) {                                       /* :423 */
        cur++;
        double previous_best_score = 0;
        for (int C = P->max_capture_size; C >= P->min_capture_size; C -= inc) {
            if (C > R->stop_flanked - R->start_flanked + P->max_mip_overlap && C - inc >= P->min_capture_size) continue;
            if (previous_best_score > P->upper_score_limit) continue;
            int a = 0;
            while (a < A) {
                int sum = P->arm_ext[a] + P->arm_lig[a];
                int a_end = a;
                while (a_end < A && P->arm_ext[a_end] + P->arm_lig[a_end] == sum) a_end++;
                if (previous_best_score > P->upper_score_limit && sum != min_sum) { a = a_end; continue; }
                int previous_minus_score = 0, previous_plus_score = 0, skip_ahead = 0;
                for (; a < a_end; a++) {
                    if (skip_ahead) continue;
                    double sc[2];
                    int skipped = 0;
                    for (int s = 0; s < 2 && !skipped; s++) {
                        mipgen_candidate c = {0, cur, C, P->arm_ext[a], P->arm_lig[a], s};
                        mo_designed d;
                        if (mo_design(P, R, &c, alleles, &d)) { skipped = 1; break; }
                        sc[s] = mo_score_designed(&d, method, m, R->long_range_content, NULL, NULL);
                        if (n < capacity) {
                            mo_emitted* o = &out[n];
                            o->scan_start = cur; o->capture_size = C; o->ext_len = c.ext_len; o->lig_len = c.lig_len;
                            o->strand = s; o->ext_copy = d.ext_copy; o->lig_copy = d.lig_copy; o->snp_count = d.snp_count;
                            o->score = sc[s];
                            o->flags[0] = d.mapping_failed; o->flags[1] = d.snp_failed; o->flags[2] = d.masking_failed; o->flags[3] = 0;
                            int ki = size_index(P, C) - g.first_size_index;
                            o->dense_index = s == 0 ? MIPGEN_PLUS_INDEX(cur - g.first_pos, g.n_sizes, ki, A, a) : MIPGEN_MINUS_INDEX(cur - g.first_pos, g.n_sizes, ki, A, a);
                        }
                        n++;
                    }
                    if (skipped) continue;
                    if (method == MIPGEN_SCORE_LOGISTIC && P->score_method == MIPGEN_SCORE_LOGISTIC && P->logistic_heuristic &&
                        sc[0] < previous_plus_score && sc[1] < previous_minus_score) skip_ahead = 1;
                    previous_best_score = (sc[1] > sc[0]) ? sc[1] : sc[0];
                    previous_minus_score = to_int_x86(sc[1]);
                    previous_plus_score = to_int_x86(sc[0]);
                }
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* print_details, mipgen.cpp:765-794                                                                */
/* ------------------------------------------------------------------------------------------------ */

int mo_print_details(const char* chr, const char* label, int feature_start, int feature_stop, int strand,
                     const mo_designed* d, double score, const char* middle, int mip_index, int minor,
                     char* buf, int bufsize)
{
    const char* st = strand == 0 ? "+" : "-";
    int e = (int)strlen(d->ext_seq), l = (int)strlen(d->lig_seq);
    char sc[64];
    snprintf(sc, sizeof sc, "%g", score);            /* default ostream formatting: 6 significant digits (:774) */
    char suffix[8] = "";
    if (d->snp_count == 1) snprintf(suffix, sizeof suffix, "_SNP_%s", minor ? "b" : "a");   /* :792 */
    return snprintf(buf, (size_t)bufsize,
        "%s:%d-%d/%d,%d/%s\t%s\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%s\t%s%s%s\t%d\t%d\t%s\t%c%c%c\t%s_%04d%s\n",
        chr, strand == 0 ? d->ext_start : d->lig_start, strand == 0 ? d->lig_stop : d->ext_stop, e, l, st,
        sc, chr, d->ext_start, d->ext_stop, d->ext_copy, d->ext_seq, d->lig_start, d->lig_stop, d->lig_copy, d->lig_seq,
        d->scan_start, d->scan_stop, d->ins_seq, d->lig_seq, middle, d->ext_seq,
        feature_start - 1, feature_stop, st, d->mapping_failed, d->snp_failed, d->masking_failed,
        label, mip_index, suffix);
}


/* ------------------------------------------------------------------------------------------------ */
/* capture-window uniqueness by brute force (see mipgen_oracle.h)             */
/* ------------------------------------------------------------------------------------------------ */
static int win_code(char c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

void mo_window_unmappable(const char* seq, int seq_len, int size, const char* const* chroms, const int64_t* chrom_lens, int n_chrom,
                          uint8_t* out, int32_t* x0_out, int32_t* x1_out)
{
    uint8_t* w = (uint8_t*)malloc((size_t)size), *wr = (uint8_t*)malloc((size_t)size);
    for (int i = 0; i < seq_len; i++) {
        out[i] = 0;
        if (x0_out) x0_out[i] = 0;
        if (x1_out) x1_out[i] = 0;
        if (i + size > seq_len) continue;                       /* the window does not fit: never written (mipgen.cpp:813) */
        int bad = 0;
        for (int j = 0; j < size; j++) { w[j] = (uint8_t)win_code(seq[i + j]); if (w[j] > 3) bad = 1; }
        if (bad) { out[i] = 1; continue; }
        for (int j = 0; j < size; j++) wr[j] = (uint8_t)(3 - w[size - 1 - j]);        /* reverse complement */
        long x0 = 0, x1 = 0;
        for (int c = 0; c < n_chrom; c++) {
            const char* g = chroms[c];
            for (int64_t p = 0; p + size <= chrom_lens[c]; p++) {
                for (int strand = 0; strand < 2; strand++) {
                    const uint8_t* q = strand ? wr : w;
                    int mm = 0;
                    for (int j = 0; j < size && mm < 2; j++) {
                        const int b = win_code(g[p + j]);
                        if (b > 3 || b != q[j]) mm++;
                    }
                    if (mm == 0) x0++; else if (mm == 1) x1++;
                }
            }
        }
        if (x0_out) x0_out[i] = (int32_t)x0;
        if (x1_out) x1_out[i] = (int32_t)x1;
        long lead = x0;
        while (lead >= 10) lead /= 10;
        out[i] = (lead == 1 && x1 == 0) ? 0 : 1;
    }
    free(w); free(wr);
}
