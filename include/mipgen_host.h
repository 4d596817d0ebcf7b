/*
 * mipgen_host.h — C ABI of the host side of the drop-in `mipgen` front end (libmipgen_host.so, C++17 inside):
 * the stages either side of the accelerated hot path.
 *
 *   reference (citations into /root/reference)                          here
 *   -------------------------------------------------------------------  ------------------------------------------------
 *   mipgen::parse_command_line / parse_arg_values  mipgen.cpp:190-276,1280-1501   mipgen_design_open (options)
 *   mipgen::query_sequences                        mipgen.cpp:293-400             mipgen_design_open (input stage: BED sort/merge,
 *       get_features_to_scan :981-1043, get_chr_fasta_sequence_* :1087-1229,          FASTA slices, TRF masks, SNPs, bwa copy tables,
 *       get_masked_features_to_scan :1045-1084, load_snps :875-978,                    output files + headers)
 *       check_copy_numbers :796-873, find_copy :558-596
 *   mipgen::tile_regions per-region tail           mipgen.cpp:503-555             mipgen_design_select_region (all_mips records,
 *       print_details :765-794, collapse_mips :1616-1668, pick_mips :1506-1614,        collapse unless the accelerator did it, pick, SNP
 *                                                                                       re-design, gap files)
 *       optimize_worst_in_region :1748-1820, translocate_down_region :1822-1908,
 *       manage_picked_mip :1910-1939, print_gaps / create_gap :1231-1278
 *   mipgen::tile_regions as a whole                mipgen.cpp:403-556             mipgen_design_run (drives libmipgen_accel on one
 *                                                                                  or several GPUs, consumes results in region order)
 *
 * The selection stage is sequential by construction: picked MIPs are numbered globally, the libc rand() stream and the used-arm
 * sets persist across regions (mipgen.cpp:97,1863,1912).  mipgen_design_select_region must therefore be called for region
 * 0, 1, 2, ... in order, by one thread: in a multi-GPU run that thread is rank 0, fed with the survivors every rank's
 * accelerator condensed (2 per scan position) - gathered over RCCL in the torch.distributed harness, or handed over in host
 * memory by the per-device worker threads of mipgen_design_run.
 *
 * Conventions: plain C; 0 on success, negative on failure with a message in mipgen_host_last_error(); a design is single-owner.
 */
#ifndef MIPGEN_HOST_H
#define MIPGEN_HOST_H

#include "mipgen_accel.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MIPGEN_HOST_E_USAGE (-101)     /* command line rejected; the message is what the reference prints before exit 1 */
#define MIPGEN_HOST_E_INPUT (-102)     /* input stage failed ("unable to tile sequences due to circumstance N", mipgen.cpp:2029-2032) */
#define MIPGEN_HOST_E_ORDER (-103)     /* regions must be selected in order */
#define MIPGEN_HOST_E_ACCEL (-104)     /* an accelerator call failed */

typedef struct mipgen_design mipgen_design;

const char* mipgen_host_last_error(void);
/* the reference's `throw <int>` code of the last failure (mipgen.cpp:2029-2035), 0 if none; -1 = a std::exception the reference raises on the same
 * input (boost::bad_lexical_cast, vector::at, std::string(NULL)): its main() reports it and exits with 0, and so does the drop-in command line;
 * -2 = a std::exception only this library can raise (allocation, internal error): reported the same way, exit status 1 */
int mipgen_host_last_circumstance(void);

/* argv as main() receives it (argv[0] locates mipgen_svr.model, mipgen.cpp:137-138,409).  Parses the options, runs the input stage
 * (shelling out to bwa / tabix / trf exactly as the reference does) and opens <project>.all_mips/collapsed_mips/picked_mips/snp_mips.txt. */
int mipgen_design_open(int argc, const char* const* argv, mipgen_design** out);
/* closes the output files, writes the tail of <project>.progress.txt */
int mipgen_design_close(mipgen_design* d);

int mipgen_design_params(const mipgen_design* d, mipgen_params* out);          /* what mipgen_accel_create needs */
int32_t mipgen_design_score_method(const mipgen_design* d);                    /* MIPGEN_SCORE_* as requested (mixed stays mixed) */
int32_t mipgen_design_silent(const mipgen_design* d);                          /* -silent_mode on */
const char* mipgen_design_model_path(const mipgen_design* d);                  /* dirname(argv[0]) + "mipgen_svr.model" */
int32_t mipgen_design_region_count(const mipgen_design* d);
/* Region i (after sort + merge) in the accelerator's input layout; the pointers stay valid until mipgen_design_close. */
int mipgen_design_region(const mipgen_design* d, int32_t i, mipgen_region* out);
/* regions first .. first + n - 1 at once (one call for a rank's whole shard) */
int mipgen_design_regions(const mipgen_design* d, int32_t first, int32_t n, mipgen_region* out);
/* region +/- 1000 bases for Featurev5::get_long_range_content (svr / mixed designs; empty otherwise) */
int mipgen_design_long_range_seq(const mipgen_design* d, int32_t i, const char** seq, int32_t* len);
int mipgen_design_set_long_range_content(mipgen_design* d, int32_t i, const double* lrc44);

/* SVR score of one candidate of region `region` for -score_method mixed (mipgen.cpp:1523-1527,1873-1877) */
typedef double (*mipgen_rescore_fn)(void* ctx, int32_t region, const mipgen_candidate* cand);

/*
 * The per-region tail of tile_regions for region i (call with i = 0, 1, 2, ...):
 *   grid        the region's dense-grid geometry as the accelerator laid it out
 *   survivors   2 * grid->n_pos records ('+','-' per scan position) from replay + condense; cand_index is batch-wide
 *               (grid->offset + index inside the region) or -1
 *   emitted     number of candidates the reference would have constructed in this region (numbers the records)
 *   scores / records / emitted_mask   the region's dense results (grid->count entries each) for the all_mips file; all three may
 *               be NULL in -silent_mode
 *   rescore     SVR re-score hook, used only by mixed designs; may be NULL otherwise
 */
int mipgen_design_select_region(mipgen_design* d, int32_t i, const mipgen_grid* grid, const mipgen_survivor* survivors,
                                int64_t emitted, const double* scores, const uint64_t* records, const uint8_t* emitted_mask,
                                mipgen_rescore_fn rescore, void* ctx);
/* The same with collapse_mips already done by the accelerator (mipgen_accel_collapse): collapsed = 2 entries per base, n_bases bases
 * from grid->first_pos on (mipgen_accel_region_bases), each the scan-start index of the winning survivor of that strand or -1. */
int mipgen_design_select_region_collapsed(mipgen_design* d, int32_t i, const mipgen_grid* grid, const mipgen_survivor* survivors,
                                          int64_t emitted, const double* scores, const uint64_t* records, const uint8_t* emitted_mask,
                                          const int32_t* collapsed, int32_t n_bases, mipgen_rescore_fn rescore, void* ctx);
/* The same for a run of regions of a SILENT design (no dense results): regions first .. first + n - 1 with their grids, their survivors one after the
 * other (2 * n_pos each) and their emitted counts - one call for a gathered batch (mipgen_amd/mp_design.py).  collapsed / n_bases: the accelerator's
 * collapse_mips results of the regions one after the other (mipgen_accel_download_collapsed: 2 * n_bases[k] entries for region first + k) or both NULL (the
 * host folds the survivors itself: ten times slower on exome-sized designs).  svr: mixed designs - the SVR score of every survivor, parallel to
 * `survivors` (what mipgen.cpp:1523-1527,1873-1877 re-score on demand); NULL otherwise. */
int mipgen_design_select_regions(mipgen_design* d, int32_t first, int32_t n, const mipgen_grid* grids, const mipgen_survivor* survivors, const int64_t* emitted,
                                 const int32_t* collapsed, const int32_t* n_bases, const double* svr);
/* Mixed designs: the survivors of regions first .. first + n - 1 (grids / survivors as above) as candidates for mipgen_accel_score_candidates on a handle
 * whose resident batch is exactly these regions (candidate.region = 0 .. n - 1), and where[j] = the survivor slot candidate j came from.
 * capacity 0: only *count is set. */
int mipgen_design_survivor_candidates(const mipgen_design* d, int32_t first, int32_t n, const mipgen_grid* grids, const mipgen_survivor* survivors,
                                      mipgen_candidate* cands, int64_t* where, int64_t capacity, int64_t* count);
/* Non-silent designs whose all_mips records were formatted on the accelerator (mipgen_accel_format_all_mips) by a caller of its own (mipgen_amd/mp_design.py):
 * what that call needs - the record names of regions first .. first + n - 1 (pointers into the design's strings, valid until close) and the -mip_middle
 * sequence - and the way back: append formatted records to the all_mips file, every record number (last column, `label_0042[_SNP_a]`) raised by
 * renumber_base (a rank numbers its records from 0; the base is what the ranks before it wrote). */
int mipgen_design_record_names(const mipgen_design* d, int32_t first, int32_t n, mipgen_record_names* out);
const char* mipgen_design_middle(const mipgen_design* d);
int mipgen_design_write_all_mips(mipgen_design* d, const char* text, int64_t n_bytes, int64_t renumber_base);
/* counters after the regions selected so far: all / collapsed / picked records written, gaps reported */
int mipgen_design_counters(const mipgen_design* d, int64_t* all_mips, int64_t* collapsed, int64_t* picked, int64_t* gaps);

/* Relative device time of every region (dense-grid candidates; for svr designs weighted with the factor-table entries of the dense SVR scorer): the
 * weights mipgen_design_run balances its contiguous device shards with, for a caller that shards the regions itself (one process per GPU:
 * mipgen_amd/mp_design.py).  capacity >= mipgen_design_region_count. */
int mipgen_design_region_weights(const mipgen_design* d, int64_t* weights, int32_t capacity);

/*
 * tile_regions on the accelerator: one libmipgen_accel handle per device, regions sharded over the devices in contiguous ranges
 * balanced by dense-grid size, each device scoring / replaying / condensing its result windows on its own thread while the calling
 * thread consumes the windows in region order through mipgen_design_select_region.  n_devices <= 0: mipgen_design_set_devices / the
 * command line's -gpus, else every visible HIP device.  Devices beyond the visible count wrap around (two handles on one GPU: a functional
 * test of the sharding).  Non-silent designs receive their all_mips records as text formatted on the device, with any number of workers
 * (records are numbered design-wide, mipgen.cpp:474,488,792: a worker numbers from 0, the consumer shifts the numbers by what the workers before it wrote).
 */
int mipgen_design_run(mipgen_design* d, int32_t n_devices);
/* Front-end knobs (the library reads no environment variables): device workers of mipgen_design_run (0 = every visible device; option -gpus),
 * cap on the candidates of one result window (0 = default policy; option -gpu_window_candidates; tests force several windows with it),
 * stage timings on stderr (option -gpu_timing on). */
int mipgen_design_set_devices(mipgen_design* d, int32_t n_devices);
/* -gpu_copy_counter on, regions fetched through mipgen_design_region(s): the HIP device their copy numbers are counted on (default 0; a rank of a
 * one-process-per-GPU run passes its own) */
int mipgen_design_set_api_device(mipgen_design* d, int32_t device);
int mipgen_design_set_window_candidates(mipgen_design* d, int64_t max_candidates);
int mipgen_design_set_timing(mipgen_design* d, int32_t on);
/* How the device workers' result windows reach the selection stage (option -gpu_gather pcie|rccl).  0 = pcie (default): every worker downloads its
 * windows over its own GPU's PCIe link.  1 = rccl: per window ONE grouped RCCL send / receive moves the window's emitted counts, condensed survivors,
 * collapse results (+ SVR re-scores of a mixed design, + all_mips text) from the worker's HBM into a packed buffer on GPU 0 over xGMI, one D2H copy
 * hands it to the selection stage while the workers score their next windows (the gather of /root/reference-equivalent per-region tables that
 * mipgen.cpp:503-515 consumes in region order).  Needs one visible GPU per device worker. */
int mipgen_design_set_gather(mipgen_design* d, int32_t rccl);

/* The selection stage's private copy of glibc's never-seeded rand() stream (mipgen.cpp:1863 picks the first strand with rand() % 2):
 * its first n values, for checking it against the C library's. */
int mipgen_host_rand_stream(int32_t* out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* MIPGEN_HOST_H */
