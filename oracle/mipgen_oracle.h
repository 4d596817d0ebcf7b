/*
 * mipgen_oracle.h — CPU restatement (plain C) of MIPgen's candidate enumeration + scoring hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (include/mipgen_accel.h, mipgen_amd/) never does.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py, tests/test_host_select_cpu.py) against the
 * real reference compiled from /root/reference by oracle/Makefile (oracle/_ref), and against the golden
 * vectors under tests/golden/ that were generated from that same build (tests/golden/make_golden.py).
 * The reference's own tests hold nothing for this path; the trained model
 * mipgen_svr.model is absent upstream (.MISSING_LARGE_BLOBS), so SVR parity is pinned for synthetic
 * libsvm models run through the reference's own svm_load_model/svm_predict.
 *
 * It shares only the POD boundary structs of include/mipgen_accel.h with the product.
 * file:line citations are into /root/reference.
 */
#ifndef MIPGEN_ORACLE_H
#define MIPGEN_ORACLE_H

#include <stdint.h>
#include "../include/mipgen_accel.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- level 1: per-candidate arithmetic on strand-oriented strings -------------------------------- */
/* reverse_comp (MinusSVMipv4.cpp:6-29): A<->T, C<->G, everything else passes through. out gets n+1 bytes. */
void mo_reverse_comp(const char* in, int n, char* out);

/* SVMipv4::get_score (SVMipv4.cpp:114-248).  ext/lig/ins are the strand-oriented sequences the object holds
 * (NUL-terminated); mip_seq may be NULL (= lig + middle + ext has no '-').  ints may be NULL. */
double mo_get_score(const char* ext, const char* lig, const char* ins, const char* mip_seq,
                    int ext_copy, int lig_copy, mipgen_candidate_ints* ints);

/* SVMipv4::get_parameters (SVMipv4.cpp:60-113): 192 doubles. */
void mo_get_parameters(const char* ext, const char* lig, const char* ins, const char* mip_seq,
                       int ext_copy, int lig_copy, const double* lrc44, double* out192);

/* Featurev5::get_long_range_content (Featurev5.cpp:18-56) with feature_mers (mipgen.cpp:32). */
void mo_long_range_content(const char* extended_seq, int chrom_seq_start, int chrom_seq_stop, double* out44);

/* ---- libsvm 3.17 model + SVR prediction (svm.cpp:2759-2973, 2504-2593, 329-368) ------------------- */
typedef struct mo_model mo_model;
mo_model* mo_svm_load_model(const char* path);           /* NULL on failure, as svm_load_model */
void mo_svm_free_model(mo_model* m);
int mo_svm_nsv(const mo_model* m);
double mo_svm_gamma(const mo_model* m);
double mo_svm_rho(const mo_model* m);
int mo_svm_kernel_type(const mo_model* m);               /* 2 = RBF */
int mo_svm_svm_type(const mo_model* m);                  /* 3 = EPSILON_SVR, 4 = NU_SVR */
/* densify: sv[n_sv][192] (absent indices 0; indices > 192 ignored and reported), coef[n_sv]; returns #entries with index>192 */
int mo_svm_densify(const mo_model* m, double* sv, double* coef);
/* mipgen::predict_value (mipgen.cpp:1948-2019) -> svm_predict: x has all 192 indices present. */
double mo_predict_value(const mo_model* m, const double* x192);

/* ---- level 2: one candidate of a region (tile_regions body + design_mip, mipgen.cpp:446-497,599-762) */
#define MO_MAX_INSERT 8192
typedef struct mo_designed {
    int32_t ext_start, ext_stop, lig_start, lig_stop, scan_start, scan_stop, scan_size;
    int32_t ext_copy, lig_copy, snp_count, masked_n;
    double arm_fraction_masked;
    char mapping_failed, snp_failed, masking_failed, has_snp_mip;
    char ext_seq[MIPGEN_MAX_OLIGO + 1];
    char lig_seq[MIPGEN_MAX_OLIGO + 1];
    char junction[3];
    char ins_seq[MO_MAX_INSERT];   /* the reference holds std::strings (no limit); this restatement covers scan sizes below MO_MAX_INSERT */
    /* alternate-allele arms when has_snp_mip (mipgen.cpp:683-689,745-751); needs the allele table, see mo_design */
    char snp_ext_seq[MIPGEN_MAX_OLIGO + 1];
    char snp_lig_seq[MIPGEN_MAX_OLIGO + 1];
} mo_designed;

/* alleles: optional per-base table of 2-char allele strings ("\0\0" = no SNP, else ref+alt as
 * chr_snp_positions holds for single-base records; multi-base records are flagged by first byte '*'),
 * indexed by (pos - seq_start), 2 bytes per base.  When NULL, region->snp_class is used for the counts and
 * no alternate sequences are produced.  Returns 0 if the candidate passes the bounds skips
 * (mipgen.cpp:443-444), 1 if it is skipped. */
int mo_design(const mipgen_params* P, const mipgen_region* R, const mipgen_candidate* c,
              const char* alleles, mo_designed* out);
/* score of a designed candidate with `method` (0 logistic / 1 svr) */
double mo_score_designed(const mo_designed* d, int method, const mo_model* m, const double* lrc44,
                         double* features192, mipgen_candidate_ints* ints);
uint64_t mo_record_of(const mo_designed* d, int valid, const mipgen_candidate_ints* ints);

/* ---- level 3: dense grid, replay, condense --------------------------------------------------------- */
/* grid geometry of a region (positions mipgen.cpp:421-425; static size skip :429); offset left 0 */
int mo_grid(const mipgen_params* P, const mipgen_region* R, mipgen_grid* g);
/* every dense-grid candidate of the region, in dense order */
int mo_score_region_dense(const mipgen_params* P, const mipgen_region* R, const mo_model* m, int method,
                          double* scores, uint64_t* records);
/* replay of the score-dependent control flow over dense scores (mipgen.cpp:426-437,440,494-497);
 * emitted[i] = 1 for candidates the reference would construct.  Returns the emitted count. */
int64_t mo_replay_region(const mipgen_params* P, const mipgen_region* R, const double* scores,
                         const uint64_t* records, uint8_t* emitted);
/* condense_mips fold (mipgen.cpp:1670-1746) over the emitted candidates; out has 2*n_pos entries
 * ([pos][strand]); cand_index is region-local. */
int mo_condense_region(const mipgen_params* P, const mipgen_region* R, const double* scores,
                       const uint64_t* records, const uint8_t* emitted, mipgen_survivor* out);

/* ---- level 4: literal restatement of the tile_regions loop (lazy scoring, mipgen.cpp:412-501) ------ */
typedef struct mo_emitted {
    int32_t scan_start, capture_size, ext_len, lig_len, strand;
    int32_t ext_copy, lig_copy, snp_count;
    double score;
    char flags[4];                   /* mapping, snp, masking + NUL, as print_details prints them (mipgen.cpp:791) */
    int64_t dense_index;             /* region-local dense-grid index */
} mo_emitted;
/* returns the number of emitted candidates; writes at most `capacity` of them */
int64_t mo_enumerate_region(const mipgen_params* P, const mipgen_region* R, const mo_model* m, int method,
                            const char* alleles, mo_emitted* out, int64_t capacity);

/* ---- output record (print_details, mipgen.cpp:765-794) --------------------------------------------- */
/* Writes one 20-column record; returns its length.  middle = universal_middle_mip_seq (mipgen.cpp:199-200). */
int mo_print_details(const char* chr, const char* label, int feature_start, int feature_stop, int strand,
                     const mo_designed* d, double score, const char* middle, int mip_index, int minor,
                     char* buf, int bufsize);

/* ---- checker of the opt-in capture-window uniqueness (mipgen_accel_window_uniqueness) ------------------------
 * The reference decides this through bwa (mipgen.cpp:806-823 writes the windows, :841-868 reads "X0:i:" / "X1:i:" back); bwa is absent here, so
 * parity against bwa itself stays unpinned.  This is the DEFINITION the device path is held to, by brute force: every window of
 * `size` bases starting at seq[i] is compared with every genome locus on both strands; X0 = loci at Hamming distance 0, X1 = loci at Hamming
 * distance exactly 1 (bytes outside ACGT never match).  bwa aln defaults modelled: -n 0.04 (>= 6 differences allowed for >= 120-base reads: never
 * binding at <= 1 difference), hits counted per locus and strand (X0 = best hits, X1 = hits with one more difference); NOT modelled: its gapped
 * one-difference alignments (-o 1 -e -1 -i 5 -d 16), the seed / queue heuristics (-l 32 -k 2 -m), the -R 30 cut-off, and its handling of N.
 * out[i] = 1 unless the reference's substring tests (:852) pass: decimal X0 starts with '1' and X1 == 0; 1 for a window with a non-ACGT byte;
 * 0 for windows that do not fit into seq (never written, :813). */
void mo_window_unmappable(const char* seq, int seq_len, int size, const char* const* chroms, const int64_t* chrom_lens, int n_chrom,
                          uint8_t* out, int32_t* x0_out, int32_t* x1_out);

#ifdef __cplusplus
}
#endif
#endif
