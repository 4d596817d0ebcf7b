/*
 * mipgen_accel.h — C-ABI drop-in boundary for the candidate-MIP enumeration + scoring hot path of MIPgen,
 * implemented as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (shendurelab/MIPGEN, C++03, single-threaded) has no FFI; the seam this library sits behind
 * is the body of mipgen::tile_regions' innermost loop and the mixed-mode re-scores
 * (file:line citations are into /root/reference):
 *
 *   reference call site / callee                                   replaced by
 *   -------------------------------------------------------------  -------------------------------------------
 *   mipgen.cpp:446-497  new Plus/MinusSVMipv4 + design_mip()        mipgen_accel_score_regions()  (dense grid)
 *                       + get_score() | get_parameters()+predict_value()
 *   mipgen.cpp:599-762  design_mip(Featurev5*, shared_ptr<SVMipv4>)  (same call; copies / masked / flags / SNP ints)
 *   SVMipv4.h:60        double SVMipv4::get_score()                  MIPGEN_SCORE_LOGISTIC
 *   SVMipv4.h:59        void SVMipv4::get_parameters(vector<double>&, double[])   mipgen_accel_score_candidates(.., features_out)
 *   mipgen.cpp:1948     double predict_value(vector<double>&, svm_model*)         MIPGEN_SCORE_SVR
 *   svm.h:78            svm_model* svm_load_model(const char*)      mipgen_accel_load_model_file()
 *   svm.h:88            double svm_predict(const svm_model*, const svm_node*)     (inside the SVR kernels)
 *   Featurev5.h:25      void Featurev5::get_long_range_content(string, string[])  mipgen_accel_long_range_content()
 *   mipgen.cpp:1525-1526,1535-1536,1548-1549,1875-1876  mixed-mode re-score        mipgen_accel_score_candidates()
 *   mipgen.cpp:426-437,494-497  score-dependent early exits (replay)               mipgen_accel_replay_condense()
 *   mipgen.cpp:1670-1746 condense_mips                                             mipgen_accel_replay_condense()
 *   mipgen.cpp:1616-1649 collapse_mips                                             mipgen_accel_collapse()
 *   mipgen.cpp:765-794   print_details (all_mips records)                          mipgen_accel_format_all_mips()
 *   mipgen.cpp:558-596,825-835 arm-oligo copy numbers through bwa                  mipgen_accel_count_oligo_copies()  (opt-in, exact matches)
 *   mipgen.cpp:412-524   tile_regions in -silent_mode (enumerate + score + condense of    mipgen_accel_score_condense_all()
 *                        every region, nothing kept per candidate)
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *   - every entry point returns 0 on success, a negative MIPGEN_E_* code otherwise;
 *     mipgen_accel_last_error() returns a human-readable message for the calling thread's last failure.
 *     (reference: C++ `throw <int>` caught in main → exit 1, mipgen.cpp:2029-2035; svm_load_model → NULL, svm.cpp:2762.)
 *   - a handle is single-owner and not thread-safe (the reference is not re-entrant either: SURVEY.md section 5);
 *     one handle per GPU.  The library owns device buffers and the densified model; the caller owns every
 *     host array it passes in or receives results into.  Nothing is transferred across the boundary.
 *   - there is NO CPU fallback: if no HIP device is usable every call fails with MIPGEN_E_NODEVICE.
 *   - coordinates are 1-based inclusive chromosome positions, as in the reference.
 */
#ifndef MIPGEN_ACCEL_H
#define MIPGEN_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  Struct layouts are unchanged since 1; 2 marks BEHAVIOUR changes of existing entry points that a caller written against 1
 * must know about (result windows): mipgen_accel_score_resident fails with MIPGEN_E_STATE on a batch of several result windows;
 * mipgen_accel_download_replay's emitted counts / survivors and mipgen_accel_download_results' range are those of the window scored last;
 * copy fields saturated at 65535 mean "look the count up"; survivor / collapse downloads are validated per result window.
 * 3: SVR requests on parameter sets outside the tiled kernel's limits succeed (list route) instead of failing with MIPGEN_E_INVALID; a download may fail
 * with MIPGEN_E_STATE when a print-exact re-score list overflowed; new entry points (mipgen_accel_window_uniqueness_begin / _flags_region / _end,
 * mipgen_accel_set_dynamic_skip / _skipped_candidates / _skip_state).
 * 4: new entry points only (mipgen_accel_rescore_survivors / _download_survivor_scores, mipgen_accel_window_views, mipgen_accel_synchronize).
 * 5: new entry point only (mipgen_accel_set_window_breaks: the multi-device front end deals its regions in blocks, a window never spans two). */
#define MIPGEN_ACCEL_ABI_VERSION 6

#define MIPGEN_MAX_ARM_PAIRS 256     /* flattened (ext,lig) list, enumeration order */
#define MIPGEN_N_FEATURES 192        /* SVMipv4.cpp:14 TOTAL_FEATURES */
#define MIPGEN_N_LRC 44              /* Featurev5.h:4 MER_NUM */
#define MIPGEN_MIN_OLIGO 1
#define MIPGEN_MAX_OLIGO 64          /* arm lengths accepted by this library (reference default 16..30) */

/* error codes */
#define MIPGEN_OK 0
#define MIPGEN_E_INVALID (-1)        /* bad argument / inconsistent sizes */
#define MIPGEN_E_NODEVICE (-2)       /* no usable HIP device: this library has no CPU path */
#define MIPGEN_E_HIP (-3)            /* a HIP runtime call or kernel failed */
#define MIPGEN_E_MODEL (-4)          /* libsvm model missing / unparsable / unsupported kernel type */
#define MIPGEN_E_NOMEM (-5)
#define MIPGEN_E_STATE (-6)          /* call sequence error (e.g. scoring before regions are resident) */

/* score_method (mipgen.cpp:182,217; "-score_method") */
#define MIPGEN_SCORE_LOGISTIC 0
#define MIPGEN_SCORE_SVR 1
#define MIPGEN_SCORE_MIXED 2         /* logistic during enumeration, SVR on picked/tested MIPs */

/*
 * Run-wide parameters, parsed once on the host (the reference re-reads map<string,string> args inside the
 * hot loop: mipgen.cpp:467,469,476,494,619,626).
 */
typedef struct mipgen_params {
    int32_t abi_version;             /* MIPGEN_ACCEL_ABI_VERSION */
    int32_t score_method;            /* MIPGEN_SCORE_* */
    int32_t min_capture_size;        /* -min_capture_size  (mipgen.cpp:272) */
    int32_t max_capture_size;        /* -max_capture_size  (mipgen.cpp:271) */
    int32_t capture_increment;       /* -capture_increment (mipgen.cpp:273-274; 0 is coerced to 1) */
    int32_t max_mip_overlap;         /* -max_mip_overlap   (static size skip, mipgen.cpp:429) */
    /* flattened arm-length pairs in enumeration order: arm sum descending, then list order within the sum
     * (mipgen.cpp:431-442; lists built at :222-261).  arm_sum_of[i] = arm_ext[i] + arm_lig[i]. */
    int32_t n_arm_pairs;
    int32_t arm_ext[MIPGEN_MAX_ARM_PAIRS];
    int32_t arm_lig[MIPGEN_MAX_ARM_PAIRS];
    int32_t check_copy_number;       /* 0 iff "-check_copy_number off" (mipgen.cpp:619) */
    int32_t logistic_heuristic;      /* 0 iff "-logistic_heuristic off" (mipgen.cpp:494) */
    double masked_arm_threshold;     /* -masked_arm_threshold (mipgen.cpp:626) */
    /* enumeration / condense thresholds in force while scanning (mipgen.cpp:264-265):
     * logistic and mixed use the logistic pair, svr the svr pair. */
    double upper_score_limit;
    double lower_score_limit;
    int32_t max_arm_copy_product;    /* -max_arm_copy_product (mipgen.cpp:197) */
    int32_t target_arm_copy;         /* -target_arm_copy (mipgen.cpp:198) */
    /* The reference keys its arm-length lists by sum and keeps a key whose list is EMPTY (-arm_length_sums 30,41,62 with the default minimum arm
     * lengths: 30 < 16 + 18 and 62 > 30 + 30 hold no pair, mipgen.cpp:245-258); the first scan position uses the LARGEST key (:421) and the one list
     * :434 never switches off is that of the SMALLEST key.  0 = the largest / smallest sum of the pairs above (every list non-empty). */
    int32_t arm_sum_key_max;
    int32_t arm_sum_key_min;
    int32_t reserved[4];
} mipgen_params;

/*
 * One merged BED interval +/- flank: the hot-path view of Featurev5 (Featurev5.h:7-28) plus the per-region
 * slices of the global lookup tables design_mip consults (mipgen.cpp:612-618,634-760).
 * All per-base arrays cover [seq_start, seq_stop] (index = position - seq_start), length seq_len.
 */
typedef struct mipgen_region {
    int32_t start_flanked;           /* Featurev5::start_position_flanked */
    int32_t stop_flanked;            /* Featurev5::stop_position_flanked */
    int32_t seq_start;               /* Featurev5::chromosomal_sequence_start_position */
    int32_t seq_stop;                /* Featurev5::chromosomal_sequence_stop_position */
    int32_t seq_len;                 /* strlen(seq); normally seq_stop - seq_start + 1 */
    int32_t reserved0;
    const char* seq;                 /* Featurev5::chromosomal_sequence (upper case) */
    const char* masked_seq;          /* Featurev5::masked_chromosomal_sequence, or NULL = same as seq */
    /* copy_chr_start_stop[chr][start][start+len-1] for oligo length len (mipgen.cpp:612-613):
     * copy[len] points to int32[seq_len] indexed by (start - seq_start); absent keys are 0, as
     * std::map::operator[] yields.  copy[len] may be NULL for lengths no arm pair uses, and the whole
     * pointer table may be NULL = "every oligo has copy 1".  MIPGEN_COPY_RESIDENT = the counts the handle itself
     * produced with mipgen_accel_count_oligo_copies_resident() for exactly this batch (they never left the device). */
    const int32_t* const* copy;      /* table of MIPGEN_MAX_OLIGO+1 pointers, indexed by oligo length */
    /* unmappable_positions[capture_size][chr] (mipgen.cpp:615-618): byte [k * seq_len + (pos - seq_start)] != 0
     * iff a MIP of capture size (max_capture_size - k*capture_increment) starting at pos is ambiguous.
     * NULL = none. */
    const uint8_t* unmappable;
    /* chr_snp_positions[chr] restricted to the region (mipgen.cpp:634-760): per base 0 = no SNP,
     * 1 = SNP for which an alternate-allele arm can be generated, 2 = SNP for which it cannot.  NULL = none. */
    const uint8_t* snp_class;
    double long_range_content[MIPGEN_N_LRC];   /* Featurev5::long_range_content (features 22..65) */
} mipgen_region;

/* Geometry of a region's dense candidate grid, as laid out in the result arrays. */
typedef struct mipgen_grid {
    int64_t offset;                  /* index of the region's first candidate in the batch-wide result arrays */
    int64_t count;                   /* n_pos * n_sizes * n_arm_pairs * 2 */
    int32_t first_pos;               /* first scan-start position p (mipgen.cpp:421-425) */
    int32_t n_pos;                   /* positions p = first_pos .. stop_flanked */
    int32_t first_size_index;        /* k0: capture sizes with k < k0 are removed by the static skip (mipgen.cpp:429) */
    int32_t n_sizes;                 /* surviving sizes C = max_capture - k*inc, k = k0 .. k0+n_sizes-1 */
} mipgen_grid;

/*
 * Dense-grid candidate index within a region (strand-major inside a (position, size) row, so that each strand's scores of a
 * row are contiguous in memory and every result store of the kernels is a full, coalesced segment):
 *     idx = ((((p - first_pos) * n_sizes + (k - k0)) * 2 + strand) * n_arm_pairs + a         strand 0 = '+', 1 = '-'
 * The reference's generation order (p asc, C desc, arm sum desc, ext asc, plus then minus: mipgen.cpp:421-491) is recovered by
 * visiting, for a = 0 .. n_arm_pairs-1, the pair (idx_plus(a), idx_plus(a) + n_arm_pairs); MIPGEN_PLUS_INDEX / MIPGEN_MINUS_INDEX.
 *
 * Integer record per candidate (uint64), the fields design_mip() produces:
 */
#define MIPGEN_PLUS_INDEX(pi, n_sizes, ki, n_pairs, a)  (((((int64_t)(pi) * (n_sizes)) + (ki)) * 2 + 0) * (n_pairs) + (a))
#define MIPGEN_MINUS_INDEX(pi, n_sizes, ki, n_pairs, a) (((((int64_t)(pi) * (n_sizes)) + (ki)) * 2 + 1) * (n_pairs) + (a))
#define MIPGEN_REC_EXT_COPY(r)   ((uint32_t)((r) & 0xFFFFu))            /* ext_probe_copy, saturated at 65535: a field that reads 65535 */
#define MIPGEN_REC_LIG_COPY(r)   ((uint32_t)(((r) >> 16) & 0xFFFFu))    /* lig_probe_copy,  means "look the count up in mipgen_region.copy" */
#define MIPGEN_REC_MASKED_N(r)   ((uint32_t)(((r) >> 32) & 0xFFu))      /* #N in masked ext + masked lig (mipgen.cpp:606-610) */
#define MIPGEN_REC_SNP_COUNT(r)  ((uint32_t)(((r) >> 40) & 0xFFu))      /* snp_count, saturated at 255 */
#define MIPGEN_REC_FLAGS(r)      ((uint32_t)(((r) >> 48) & 0xFFu))
#define MIPGEN_REC_JUNCTION(r)   ((uint32_t)(((r) >> 56) & 0xFFu))      /* 4*code(lj[0])+code(lj[1]), A<C<G<T; 255 if not ACGT */
#define MIPGEN_FLAG_VALID        0x01u   /* passes the bounds skips of mipgen.cpp:443-444 (a candidate the reference could emit) */
#define MIPGEN_FLAG_GUARD        0x02u   /* N in either arm or '-' in mip_seq: score -1000 / all-zero vector (SVMipv4.cpp:63,116) */
#define MIPGEN_FLAG_MAPPING      0x04u   /* mapping_failed == '1' (mipgen.cpp:615-625) */
#define MIPGEN_FLAG_MASKING      0x08u   /* masking_failed == '1' (mipgen.cpp:626-633) */
#define MIPGEN_FLAG_SNP          0x10u   /* snp_failed == '1'     (mipgen.cpp:690-693,753-760) */
#define MIPGEN_FLAG_HAS_SNP_MIP  0x20u   /* has_snp_mip           (mipgen.cpp:685,747) */

/* A single candidate addressed by coordinates, for sparse (mixed-mode) re-scoring and inspection. */
typedef struct mipgen_candidate {
    int32_t region;                  /* index into the resident region batch */
    int32_t scan_start;              /* p */
    int32_t capture_size;            /* C */
    int32_t ext_len;
    int32_t lig_len;
    int32_t strand;                  /* 0 '+', 1 '-' */
} mipgen_candidate;

/* Integer features of one candidate as the two scorers see them (strand-oriented sequences). */
typedef struct mipgen_candidate_ints {
    int32_t ext_a, ext_c, ext_g, ext_t;        /* base counts of the extension arm */
    int32_t lig_a, lig_c, lig_g, lig_t;
    int32_t ins_a, ins_c, ins_g, ins_t;        /* base counts of the insert (scan target) */
    int32_t run_count;                         /* GC/AT run counter incl. the final ++ (SVMipv4.cpp:118-142) */
    int32_t junction;                          /* as MIPGEN_REC_JUNCTION */
    int32_t ext_copy, lig_copy;                /* unsaturated */
    int32_t masked_n;
    int32_t snp_count;
    int32_t flags;
    int32_t scan_size;
} mipgen_candidate_ints;

/* Per (scan-start, strand) survivor of the reference's replay + condense_mips fold (mipgen.cpp:1670-1746). */
typedef struct mipgen_survivor {
    int64_t cand_index;              /* dense-grid index within the batch (mipgen_grid.offset + index in the region), or -1 if none survived */
    double score;
    uint64_t record;
} mipgen_survivor;

typedef struct mipgen_accel mipgen_accel;   /* opaque */

/* ---- lifecycle ------------------------------------------------------------------------------------ */
int mipgen_accel_abi_version(void);
const char* mipgen_accel_last_error(void);
/* number of usable HIP devices (0 if none); never fails */
int mipgen_accel_device_count(void);
/* device: HIP ordinal.  stream: a hipStream_t the caller owns (e.g. torch's current stream) or NULL for a private one. */
int mipgen_accel_create(const mipgen_params* params, int device, void* stream, mipgen_accel** out);
void mipgen_accel_destroy(mipgen_accel* h);

/* ---- model (svm.h:78 svm_load_model) -------------------------------------------------------------- */
/* Parses a libsvm 3.17 text model (grammar svm.cpp:2779-2962).  Only epsilon_svr/nu_svr + rbf are accepted;
 * anything else, or a missing file, is MIPGEN_E_MODEL (the reference dereferences NULL instead, svm.cpp:2507). */
int mipgen_accel_load_model_file(mipgen_accel* h, const char* path);
/* Same from memory: sv is row-major [n_sv][192] (absent libsvm indices = 0), coef[n_sv]. */
int mipgen_accel_set_model(mipgen_accel* h, int32_t n_sv, double gamma, double rho, const double* coef, const double* sv);
int mipgen_accel_model_info(const mipgen_accel* h, int32_t* n_sv, double* gamma, double* rho);

/* ---- region batch: host -> HBM -------------------------------------------------------------------- */
/* Copies the batch into device memory (sequence bytes, masked bytes, copy tables, SNP / mappability maps,
 * long-range content) and lays out the dense grids.  grids_out (n entries, caller-allocated) may be NULL.
 * Replaces any previously resident batch.
 *
 * Result windows.  The inputs of EVERY region of the batch stay resident (they are small: ~16-70 bytes per base of region).
 * The dense results (8 B score + 8 B record + 1 B emitted flag per candidate) are what fills HBM, so the batch is cut into
 * windows of consecutive regions whose dense results fit the result arrays; windows are scored one after the other into the
 * same arrays.  A batch that fits is one window (the usual case: 288 GB hold ~1.4e10 candidates).  mipgen_grid.offset stays
 * batch-wide; a region is never split. */
int mipgen_accel_upload_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, mipgen_grid* grids_out);
/* total dense-grid candidates of the resident batch */
int64_t mipgen_accel_batch_candidates(const mipgen_accel* h);
/* upper bound on the candidates of one result window for the following uploads (0 = automatic: what fits in free device memory, at most
 * 2^30 candidates unless a single region is larger) */
int mipgen_accel_set_window_candidates(mipgen_accel* h, int64_t max_candidates);
/* regions of the following uploads at which a new result window must start whatever the candidate bound says (ascending batch indices; n = 0
 * clears them): a front end that deals consecutive region blocks of one design to several devices in turn (the selection stage consumes the
 * blocks in design order, /root/reference/mipgen.cpp:503-515) uploads its blocks as ONE batch and never gets a window that spans two of them */
int mipgen_accel_set_window_breaks(mipgen_accel* h, const int32_t* first_regions, int32_t n);
int32_t mipgen_accel_window_count(const mipgen_accel* h);
/* regions [first_region, +n_regions), candidates [first_candidate, +n_candidates) and scan positions of window w; any output may be NULL */
int mipgen_accel_window_info(const mipgen_accel* h, int32_t w, int32_t* first_region, int32_t* n_regions, int64_t* first_candidate,
                             int64_t* n_candidates, int64_t* first_position, int64_t* n_positions);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* Scores the resident batch's dense grid with `method` (MIPGEN_SCORE_LOGISTIC or MIPGEN_SCORE_SVR) into
 * library-owned device arrays (double scores[], uint64 records[], both batch_candidates long).
 * Asynchronous on the handle's stream.
 * Every parameter set the reference accepts is scored (mipgen.cpp:222-261, 427-444: any -arm_lengths / -capture_increment / range): SVR
 * requests outside the tiled kernel's limits (scan sizes below 3, more than 240 arm pairs, a tile beyond 160 KiB of LDS) take the
 * list scorer over the window's dense index range - same results, a lower rate (any scan size: the list kernels pass an insert through LDS in
 * pieces of 1,024 bases). */
int mipgen_accel_score_resident(mipgen_accel* h, int32_t method);       /* single-window batches; else MIPGEN_E_STATE */
/* the same for result window w of a larger batch; the window's results replace the previous window's */
int mipgen_accel_score_window(mipgen_accel* h, int32_t w, int32_t method);
/* device pointers of the result arrays (element 0 = first candidate of the window scored last; valid until the next upload) */
int mipgen_accel_result_device_ptrs(const mipgen_accel* h, void** scores_dev, void** records_dev);
/* blocks until the stream is idle, then copies results to host arrays (either may be NULL); [first, first+count) are batch-wide
 * candidate indices and must lie inside the window scored last */
int mipgen_accel_download_results(mipgen_accel* h, double* scores, uint64_t* records, int64_t first, int64_t count);
/* The whole batch the way a -silent_mode design needs it (mipgen.cpp:412-524 without the all_mips / collapsed files): every window is
 * scored, replayed, condensed and collapsed back to back on the stream; only the survivors (2 per scan position), the per-base
 * collapse result and the per-region emitted counts are kept.  Asynchronous; fetch with mipgen_accel_download_survivors /
 * mipgen_accel_download_collapsed. */
int mipgen_accel_score_condense_all(mipgen_accel* h, int32_t method);
/* ONE window that way (ABI 6): scored, replayed and condensed; what mipgen_accel_score_window + mipgen_accel_replay_condense leave for a caller that will
 * never read the window's dense results (a -silent_mode front end that downloads window w while window w + 1 is scored): the print-exact re-score of
 * mipgen_accel_set_print_exact then tests the 2 survivors per scan position - the only scores such a design prints, mipgen.cpp:1917 - instead of every
 * dense candidate.  Follow with mipgen_accel_collapse / mipgen_accel_download_replay / mipgen_accel_download_collapsed as after mipgen_accel_replay_condense. */
int mipgen_accel_score_condense_window(mipgen_accel* h, int32_t w, int32_t method);
/* upload + score + download in one call: the literal replacement for the loop body of mipgen.cpp:446-497 */
int mipgen_accel_score_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, int32_t method,
                               mipgen_grid* grids_out, double* scores, uint64_t* records, int64_t capacity);

/* Sparse list against the resident batch (mixed-mode re-scores, mipgen.cpp:1525-1526,1875-1876; inspection).
 * Any output pointer may be NULL.  features: [n][192] as SVMipv4::get_parameters fills it. */
int mipgen_accel_score_candidates(mipgen_accel* h, const mipgen_candidate* cands, int32_t n, int32_t method,
                                  double* scores, uint64_t* records, double* features, mipgen_candidate_ints* ints);

/* Featurev5::get_long_range_content on the device: extended_seq covers the region +/- 1000 bases
 * (mipgen.cpp:1125-1128,1225); denominator = chrom_seq_stop - chrom_seq_start + 2001 (Featurev5.cpp:49,53). */
int mipgen_accel_long_range_content(mipgen_accel* h, const char* extended_seq, int32_t len,
                                    int32_t chrom_seq_start, int32_t chrom_seq_stop, double* out44);
/* the same for n regions in one launch (one workgroup per region); out is [n][44] */
int mipgen_accel_long_range_content_batch(mipgen_accel* h, int32_t n, const char* const* extended_seqs, const int32_t* lens,
                                          const int32_t* chrom_seq_starts, const int32_t* chrom_seq_stops, double* out);

/* Replays the reference's score-dependent enumeration control flow (mipgen.cpp:426-437,494-497) over the
 * scored dense grid on the device and folds condense_mips (mipgen.cpp:1670-1746) per (scan start, strand).
 * emitted_dev/ survivors are library-owned device arrays; results are fetched with the calls below. */
int mipgen_accel_replay_condense(mipgen_accel* h);
/* results of the window replayed last: per-region emitted-candidate counts (int64[regions of the window]), survivors (2 per scan
 * position of the window: '+','-') and the emitted mask (one byte per candidate of the window).  Any pointer may be NULL. */
int mipgen_accel_download_replay(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors,
                                 int64_t survivor_capacity, uint8_t* emitted_mask, int64_t mask_capacity);
/* after mipgen_accel_score_condense_all: emitted counts of every region (int64[n_regions]) and the survivors of every scan position
 * of the batch (2 per position, region order) */
int mipgen_accel_download_survivors(mipgen_accel* h, int64_t* emitted_per_region, mipgen_survivor* survivors, int64_t survivor_capacity);
/* device pointer of that survivor array (mipgen_survivor[n_survivors], batch order): the send buffer of the multi-GPU gather */
int mipgen_accel_survivors_device_ptr(const mipgen_accel* h, void** survivors_dev, int64_t* n_survivors);

/* collapse_mips (mipgen.cpp:1616-1649) on the device, over the survivors of the window replayed last: for every base a survivor's
 * scan target can cover (bases first_pos .. first_pos + n_bases - 1 of each region, n_bases = n_pos + largest scan size - 1) and each
 * strand, the scan-start index (0-based from the region's first_pos) of the survivor the reference's fold keeps, or -1.
 * mipgen_accel_score_condense_all runs it for every window. */
int mipgen_accel_collapse(mipgen_accel* h);
/* entries [first_entry, first_entry + 2 * n_bases) of the batch-wide collapsed array belong to `region`: [base][strand] */
int mipgen_accel_region_bases(const mipgen_accel* h, int32_t region, int64_t* first_entry, int32_t* n_bases);
/* window >= 0: the entries of that window's regions; window < 0: the whole batch (after mipgen_accel_score_condense_all) */
int mipgen_accel_download_collapsed(mipgen_accel* h, int32_t window, int32_t* best_scan_index, int64_t capacity);

/* ---- mixed designs: the SVR score of every condensed survivor, computed where the survivors are --------------------------------------
 * The reference re-scores the MIPs its pick stage tests one at a time (mipgen.cpp:1523-1527, 1533-1537, 1546-1550, 1873-1877: get_parameters +
 * predict_value, score overwritten in place).  mipgen_accel_rescore_survivors scores EVERY condensed survivor of the window scored + replayed
 * last in one list call on the device (the same kernels and values as mipgen_accel_score_candidates on that list) and keeps the values in the
 * handle, slot for slot beside the survivors (NaN where a slot holds no survivor); the selection stage looks them up. */
int mipgen_accel_rescore_survivors(mipgen_accel* h);
int mipgen_accel_download_survivor_scores(mipgen_accel* h, int32_t window, double* svr, int64_t capacity /* >= 2 * n_positions of the window */);

/* ---- device-side views of a result window ---------------------------------------------------------------------------------------------
 * For a caller that moves results between devices itself - the multi-GPU front end posts these arrays to GPU 0 with one grouped RCCL
 * send / receive per window instead of taking them down each device's own PCIe link.  Pointers into the handle's own device arrays, valid
 * until the next mipgen_accel_upload_regions / mipgen_accel_destroy (text: until the next mipgen_accel_format_all_mips); a field is NULL
 * when the window does not hold it.  mipgen_accel_synchronize waits for everything the handle has enqueued, so that another stream may read
 * them.  cand_index of the survivors is batch-wide: subtract first_candidate for the window-relative index of mipgen_accel_window_info. */
typedef struct mipgen_window_views {
    const void* emitted;             /* int64 [n_emitted]: emitted candidates per region of the window */
    const void* survivors;           /* mipgen_survivor [n_survivors] = 2 per scan position */
    const void* collapsed;           /* int32 [n_collapsed] = 2 per base, region after region (mipgen_accel_collapse) */
    const void* survivor_svr;        /* double [n_survivors] (mipgen_accel_rescore_survivors) */
    const void* text;                /* all_mips text of the last mipgen_accel_format_all_mips, if this is the window it ran on */
    int64_t n_emitted, n_survivors, n_collapsed, n_text_bytes;
    int64_t first_candidate;
} mipgen_window_views;
int mipgen_accel_window_views(mipgen_accel* h, int32_t window, mipgen_window_views* out);
int mipgen_accel_synchronize(mipgen_accel* h);

/* ---- section 8f-4: the all_mips records of a window, formatted on the device -----------------------------------------------------
 * print_details (mipgen.cpp:765-794) for every candidate the replay marked as constructed, in the reference's generation order
 * (position, capture size, arm pair, plus then minus), numbered from first_index + 1 (the running all_mip_counter, mipgen.cpp:474,488):
 * 20 tab-separated columns, scores as printf("%g"), copy numbers from the copy table, mip_sequence = lig + middle + ext.
 * Call after mipgen_accel_replay_condense on the scored window; names[i] describes region i of that window. */
typedef struct mipgen_record_names {
    const char* chr;                 /* Featurev5::chr */
    const char* label;               /* Featurev5::label */
    int32_t feature_start;           /* start_position - 1 (mipgen.cpp:788) */
    int32_t feature_stop;            /* stop_position (:789) */
} mipgen_record_names;
int mipgen_accel_format_all_mips(mipgen_accel* h, const mipgen_record_names* names, const char* middle, int64_t first_index,
                                 int64_t* n_records, int64_t* n_bytes);
/* the text of the last mipgen_accel_format_all_mips (n_bytes bytes, no terminator) */
int mipgen_accel_download_text(mipgen_accel* h, char* dst, int64_t capacity);

/* ---- section 8f-3 (opt-in): arm-oligo copy numbers without the bwa round trip ----------------------------------------------------
 * Replaces check_copy_numbers / find_copy's oligo half (mipgen.cpp:825-835 writes every arm oligo to a FASTQ file, :558-596 reads bwa's
 * X0:i best-hit count back): an oligo always matches itself, so its best hits are its exact occurrences on either strand, counted here in
 * ONE streaming pass over the genome (1 byte per base; canonical 2-bit k-mer keys, exact for lengths <= 31).
 *   chrom_seqs / chrom_lens   the genome the oligos are counted against (ASCII, any case; non-ACGT bytes break k-mers)
 *   region_seqs / region_lens the region strings (Featurev5::chromosomal_sequence)
 *   lengths                   the design's distinct oligo lengths, ascending
 *   copy_out[r]               int32 [n_lengths][region_lens[r]]: exactly the mipgen_region.copy slices (copy[len][start - seq_start]);
 *                             oligos with a non-ACGT byte get 100 (a read without an X0 tag, :589-592), oligos that would run past the
 *                             region string 0 (never written, :829)
 * The uniqueness test of whole capture windows (:841-868) is mipgen_accel_window_uniqueness() below. */
int mipgen_accel_count_oligo_copies(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens,
                                    int32_t n_regions, const char* const* region_seqs, const int32_t* region_lens,
                                    int32_t n_lengths, const int32_t* lengths, int32_t* const* copy_out);

/* The capture-window half of check_copy_numbers (mipgen.cpp:806-823 writes every capture window to a FASTQ file, :841-868 marks a window start
 * "unmappable" unless bwa's SAM line contains "X0:i:1" and "X1:i:0"; design_mip sets mapping_failed there, :615-625).  Model: a window taken from
 * the genome matches itself exactly, so X0 = its exact occurrences on either strand and X1 = its occurrences with exactly ONE SUBSTITUTION
 * (Hamming distance 1); bwa aln's gapped one-difference hits are not searched.  The substring tests of :852 are kept (X0 printed with a leading
 * '1' passes: 1, 10-19, ...).  Seed-and-extend from the k-mer table: a window within distance 1 of a locus contains its first or its second
 * seed_len-mer exactly, so only the loci of repeated seeds are extended (two streaming passes over the genome).
 *   sizes                     capture sizes (any order), each >= 2 * seed_len
 *   seed_len                  12..31; the design's longest arm oligo (30) is the natural choice
 *   unmap_out[r]              uint8 [n_sizes][region_lens[r]]: 1 = the window of sizes[c] starting at index i of the region string is not unique
 *                             (or holds a non-ACGT byte); 0 = unique, or the window does not fit into the region string.  The caller keeps
 *                             the starts the reference enumerates ([start_flanked - C, stop_flanked), :808-812) - exactly mipgen_region.unmappable. */
int mipgen_accel_window_uniqueness(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens,
                                   int32_t n_regions, const char* const* region_seqs, const int32_t* region_lens,
                                   int32_t n_sizes, const int32_t* sizes, int32_t seed_len, uint8_t* const* unmap_out);

/* The same test for a caller that wants exactly mipgen_region.unmappable: `bounds` gives, per region, the coordinates the reference's lookup is limited
 * by (mipgen.cpp:808-813: a window start is written only for current_mip_start in [start_flanked - C, stop_flanked), > 0, with the window inside the
 * region string [seq_start, seq_stop]); the flags of every other start are cleared ON THE DEVICE, the flag image stays in the handle, and any_out[r]
 * (one byte per region) says whether region r has a flagged start at all.  Only those regions need mipgen_accel_window_flags_region (out: uint8
 * [n_sizes][region_lens[r]]); mipgen_accel_window_uniqueness_end releases the image (so do the next _begin and mipgen_accel_destroy).  An exome design
 * has a flagged start in a few percent of its regions: 560 MB of flags stay where they were computed. */
typedef struct mipgen_window_bounds { int32_t start_flanked, stop_flanked, seq_start, seq_stop; } mipgen_window_bounds;
int mipgen_accel_window_uniqueness_begin(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens,
                                         int32_t n_regions, const char* const* region_seqs, const int32_t* region_lens, const mipgen_window_bounds* bounds,
                                         int32_t n_sizes, const int32_t* sizes, int32_t seed_len, uint8_t* any_out);
int mipgen_accel_window_flags_region(mipgen_accel* h, int32_t region, uint8_t* out);
int mipgen_accel_window_uniqueness_end(mipgen_accel* h);

/* The same counts, kept in the handle's device memory in the layout the scoring kernels read (288 GB of HBM: the tables of a whole exome
 * are 6.7 GB and would otherwise cross PCIe twice).  The oligo lengths are the ones the handle's arm pairs use.  The next
 * mipgen_accel_upload_regions must pass the SAME regions (count, order, seq_len) with copy = MIPGEN_COPY_RESIDENT in every one of them;
 * any other upload discards the resident tables.  Counts of 65535 and more - which the 16-bit record fields cannot carry and the host looks
 * up instead (MIPGEN_REC_EXT_COPY) - are returned as a list owned by the handle, valid until the next call on it
 * (start = 0-based offset of the oligo in the region string). */
#define MIPGEN_COPY_RESIDENT ((const int32_t* const*)(uintptr_t)1)
typedef struct mipgen_big_copy { int32_t region, length, start, copies; } mipgen_big_copy;
int mipgen_accel_count_oligo_copies_resident(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens,
                                             int32_t n_regions, const char* const* region_seqs, const int32_t* region_lens,
                                             int64_t* n_big, const mipgen_big_copy** big);

/* ---- tuning ---------------------------------------------------------------------------------------- */
/* A dense SVR launch with few tiles is split along the support-vector list so that it still fills the chip (partial sums are added
 * in a fixed order: results are deterministic for a given split).  0 = chosen per launch from the tile count (default), n >= 1 forces
 * n parts - e.g. to compare two differently sized batches bit for bit. */
int mipgen_accel_set_sv_split(mipgen_accel* h, int32_t n_split);
/* SVR scores that sit within the device kernels' error (~1e-12) of a midpoint between two 6-significant-digit numbers - the precision the
 * front end prints scores with (mipgen.cpp:774) - are re-scored in the reference's own operation order (svm.cpp:329-368, 2511-2515: index-order
 * sums, every operation rounded on its own) and overwritten, so that the printed digit is the reference's.  On by default; 0 switches it off
 * (measurements, tests of the mechanism).  Dense windows and candidate lists (mixed designs) alike.  Should a window hold more such scores
 * than the re-score list (1/1024 of its candidates + 4096), the next download of its results fails with MIPGEN_E_STATE instead of handing
 * out digits that are not guaranteed. */
/* (ABI 4: logistic scores take the same route - within 1e-11 of a midpoint they are re-scored with the 69 terms in the reference's order, SVMipv4.cpp:176-247.) */
int mipgen_accel_set_print_exact(mipgen_accel* h, int32_t on);
/* The reference stops constructing candidates at a scan position for good once a capture size starts with previous_best_score above the upper
 * score limit (mipgen.cpp:430).  A region of more than nine capture sizes is scored in runs of <= 9 sizes; with this switch on the runs are scored in
 * order and a tile of a later run is left out when every one of its positions has stopped before it - exactly the candidates the reference never
 * constructs.  Emitted masks, emitted counts, survivors, collapse results and all_mips records are unchanged (the replay never consults the rows
 * behind a position's exit); the DENSE scores of skipped tiles read NaN, which is why the switch is off by default for callers that fetch the dense
 * grid.  SVR scoring through the tiled kernel only.  How much it saves is a property of the model (how early the arm-sum lists' last pairs score above
 * the limit); mipgen_accel_skipped_candidates returns - and resets - the dense candidates left out since the last call. */
int mipgen_accel_set_dynamic_skip(mipgen_accel* h, int32_t on);
int mipgen_accel_skipped_candidates(mipgen_accel* h, int64_t* n);
/* inspection: per scan position of the window scored last with the switch on, the state after its last-but-one run (0 still constructing, 1 stopped,
 * 2 too close to the limit to call) and previous_best_score there; either output may be NULL */
int mipgen_accel_skip_state(mipgen_accel* h, uint8_t* state, double* previous_best, int64_t capacity);
/* The dense logistic kernel gives a workgroup `n` consecutive runs of scan positions (it stages the bases once and slides its downstream-arm
 * table from run to run).  0 = chosen from the batch size (1 for small batches, which need every workgroup they can get; 2 or 3 for large ones),
 * 1..8 forced.  Results do not depend on it. */
int mipgen_accel_set_logistic_subruns(mipgen_accel* h, int32_t n);

/* ---- instrumentation ------------------------------------------------------------------------------ */
/* HIP-event time (ms) of the kernels of the last scoring call (summed over its windows), measured on the handle's stream;
 * negative if unavailable.  which: 0 = dense SVR kernel, 1 = records + scoring kernels, 2 = records / logistic kernel,
 * 3 = replay + condense; 4 = genome pass of the last mipgen_accel_count_oligo_copies (always recorded);
 * 5 / 6 = the matrix-core SVR kernel / the feature kernel of the last mipgen_accel_score_candidates call on a list (>= 256 SVR candidates). */
double mipgen_accel_last_kernel_ms(mipgen_accel* h, int32_t which);
/* enable/disable per-call event timing (it inserts two hipEventRecord per call) */
int mipgen_accel_set_timing(mipgen_accel* h, int32_t enabled);

#ifdef __cplusplus
}
#endif
#endif /* MIPGEN_ACCEL_H */
