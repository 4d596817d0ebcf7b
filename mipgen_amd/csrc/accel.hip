// accel.hip — host side of libmipgen_accel.so: the C-ABI of include/mipgen_accel.h over the gfx950 kernels - lifecycle, model, region batch,
// instrumentation (scoring: accel_score.hip; tile lists: accel_tiles.hip; section 8f-3: accel_kmer.hip).
// No CPU scoring path exists in this library: every entry point either drives the GPU or fails.
#include "accel_internal.h"

thread_local char g_mipgen_accel_err[512] = "";

// ---- C ABI ---------------------------------------------------------------------------------------------------
extern "C" {

int mipgen_accel_abi_version(void) { return MIPGEN_ACCEL_ABI_VERSION; }
const char* mipgen_accel_last_error(void) { return g_err; }

int mipgen_accel_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mipgen_accel_create(const mipgen_params* params, int device, void* stream, mipgen_accel** out)
{
    if (!params || !out) return fail(MIPGEN_E_INVALID, "null argument");
    *out = nullptr;
    if (params->abi_version != MIPGEN_ACCEL_ABI_VERSION) return fail(MIPGEN_E_INVALID, "abi_version %d != %d", params->abi_version, MIPGEN_ACCEL_ABI_VERSION);
    if (params->n_arm_pairs < 1 || params->n_arm_pairs > MIPGEN_MAX_ARM_PAIRS) return fail(MIPGEN_E_INVALID, "n_arm_pairs out of range");
    if (params->capture_increment <= 0) return fail(MIPGEN_E_INVALID, "capture_increment must be positive (the reference coerces 0 to 1 before this call)");
    if (params->max_capture_size < params->min_capture_size) return fail(MIPGEN_E_INVALID, "max_capture_size < min_capture_size");
    int ndev = mipgen_accel_device_count();
    if (ndev <= 0) return fail(MIPGEN_E_NODEVICE, "no HIP device: libmipgen_accel has no CPU path");
    if (device < 0 || device >= ndev) return fail(MIPGEN_E_INVALID, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    mipgen_accel* h = new mipgen_accel();
    // the buffers that reach gigabytes share the handle's pool
    h->copy.pool = h->scores.pool = h->partials.pool = h->records.pool = h->emitted.pool = h->survivors.pool = h->collapsed.pool = h->pos_region.pool =
        h->pos_local.pool = h->bases.pool = h->letters.pool = h->unmap.pool = h->fmt_pool.pool = h->fmt_text.pool = h->fmt_temp.pool = &h->pool;
    h->device = device;
    h->params = *params;
    DevParams& D = h->hp;
    memset(&D, 0, sizeof D);
    D.min_capture = params->min_capture_size; D.max_capture = params->max_capture_size; D.inc = params->capture_increment;
    D.n_sizes_all = n_sizes_all(*params);
    D.n_pairs = params->n_arm_pairs;
    D.check_copy_number = params->check_copy_number; D.logistic_heuristic = params->logistic_heuristic;
    D.score_method = params->score_method;
    D.masked_arm_threshold = params->masked_arm_threshold; D.upper = params->upper_score_limit; D.lower = params->lower_score_limit;
    D.max_arm_copy_product = params->max_arm_copy_product; D.target_arm_copy = params->target_arm_copy;
    D.min_sum = 1 << 30; D.max_sum = 0; D.e_min = D.l_min = 1 << 30; D.e_max = D.l_max = 0;
    for (int i = 0; i <= MIPGEN_MAX_OLIGO; i++) D.len_slot[i] = -1;
    for (int i = 0; i < D.n_pairs; i++) {
        int e = params->arm_ext[i], l = params->arm_lig[i];
        if (e < 2 || l < 2 || e > MIPGEN_MAX_OLIGO || l > MIPGEN_MAX_OLIGO) { delete h; return fail(MIPGEN_E_INVALID, "arm length out of range [2,%d]", MIPGEN_MAX_OLIGO); }
        D.arm_ext[i] = (uint8_t)e; D.arm_lig[i] = (uint8_t)l;
        D.min_sum = std::min(D.min_sum, e + l); D.max_sum = std::max(D.max_sum, e + l);
        D.e_min = std::min(D.e_min, e); D.e_max = std::max(D.e_max, e);
        D.l_min = std::min(D.l_min, l); D.l_max = std::max(D.l_max, l);
        D.len_slot[e] = 0; D.len_slot[l] = 0;
    }
    if (params->min_capture_size - D.max_sum < 1) { delete h; return fail(MIPGEN_E_INVALID, "min_capture_size leaves no scan target (capture size <= arm length sum)"); }
    // keys of the reference's arm-sum map whose lists are empty (mipgen.cpp:245-258 -> :421, :434): they can only lie outside the pairs' own sums
    if ((params->arm_sum_key_max > 0 && params->arm_sum_key_max < D.max_sum) || (params->arm_sum_key_min > 0 && params->arm_sum_key_min > D.min_sum)) {
        delete h; return fail(MIPGEN_E_INVALID, "arm_sum_key_max / arm_sum_key_min inside the arm pairs' own sums [%d, %d]", D.min_sum, D.max_sum);
    }
    D.key_min_sum = params->arm_sum_key_min > 0 ? params->arm_sum_key_min : D.min_sum;
    // limits of the dense SVR kernel are reported when an SVR launch is requested: logistic designs are not bound by them
    if (params->min_capture_size - D.max_sum < 3) h->svr_geometry_error = "min_capture_size leaves a scan size < 3 (insert 3-mer frequencies are undefined)";
    int slot = 0;
    for (int i = 0; i <= MIPGEN_MAX_OLIGO; i++) if (D.len_slot[i] == 0) D.len_slot[i] = (int8_t)slot++;
    D.n_len_slots = slot;
    for (int i = 0; i < D.n_pairs;) {                         // arm-sum lists are contiguous runs (mipgen.cpp:431-438)
        int j = i, s = D.arm_ext[i] + D.arm_lig[i];
        while (j < D.n_pairs && D.arm_ext[j] + D.arm_lig[j] == s) j++;
        for (int k = i; k < j; k++) D.group_end[k] = (uint16_t)j;
        i = j;
    }
    // SVR thread geometry
    SvrGeom& G = h->geom;
    memset(&G, 0, sizeof G);
    G.nchunk = (D.n_pairs + SVR_MAX_CHUNK - 1) / SVR_MAX_CHUNK;
    G.chunk_len = (D.n_pairs + G.nchunk - 1) / G.nchunk;
    G.n_e = D.e_max - D.e_min + 1; G.n_l = D.l_max - D.l_min + 1;
    G.group = SVR_GROUP;
    G.wpc = 4;                                                   // waves per arm-pair chunk
    while (G.wpc > 1 && G.nchunk * G.wpc * 64 > SVR_MAX_THREADS) G.wpc--;
    if (G.nchunk * G.wpc * 64 > SVR_MAX_THREADS) {
        char msg[128];
        snprintf(msg, sizeof msg, "too many arm pairs for the dense SVR kernel (max %d)", SVR_MAX_CHUNK * (SVR_MAX_THREADS / 64));
        h->svr_geometry_error = msg;
        G.wpc = 1;
    }
    // Regions that keep ONE capture size after the static skip of mipgen.cpp:429 (short exons: half of an exome BED) hold few (position, size)
    // items per tile - the tile's positions are bound by its LDS, not by the lanes -, so with the main geometry half of the wavefronts own no
    // candidate while the others walk fifteen pairs each.  Their tiles run with twice the chunks of half the length on two wavefronts each.
    h->geom_few = G;
    h->have_few = false;
    if (h->svr_geometry_error.empty() && G.wpc == 4 && G.chunk_len > 8) {
        SvrGeom& F = h->geom_few;
        const int max_chunks = SVR_MAX_THREADS / 64 / 2;
        F.chunk_len = (D.n_pairs + max_chunks - 1) / max_chunks;
        F.nchunk = (D.n_pairs + F.chunk_len - 1) / F.chunk_len;
        F.wpc = 2;
        h->have_few = F.chunk_len < G.chunk_len && F.nchunk * F.wpc * 64 <= SVR_MAX_THREADS;
    }
    {
        const int n_arm = std::max(G.n_e, G.n_l) | 1;            // table row pitch in slots (as in the kernel)
        for (int strand = 0; strand < 2; strand++) {
            const int up_min = strand ? D.l_min : D.e_min, dn_min = strand ? D.e_min : D.l_min;
            for (int i = 0; i < MIPGEN_MAX_ARM_PAIRS + 16; i++) {
                const int a = std::min(i, D.n_pairs - 1);
                const int e = D.arm_ext[a], l = D.arm_lig[a];
                const uint32_t uU = (uint32_t)(((strand ? l : e) - up_min) * SVR_GROUP * 8);
                const uint32_t uD = (uint32_t)(((D.max_sum - (e + l)) * n_arm + ((strand ? e : l) - dn_min)) * SVR_GROUP * 8);
                D.pair_prog[strand][i] = uU | (uD << 16);
            }
        }
    }

    if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(MIPGEN_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    for (int i = 0; i < 101; i++) h->hconsts.log10_tab[i] = log10((double)i);    // host libm, as the reference computes it
    h->hconsts.ln_base = log(MIPGEN_LOGISTIC_BASE);
    hipError_t e1 = hipMalloc((void**)&h->dp, sizeof(DevParams));
    hipError_t e2 = hipMalloc((void**)&h->dconsts, sizeof(HostConsts));
    if (e1 != hipSuccess || e2 != hipSuccess) { mipgen_accel_destroy(h); return fail(MIPGEN_E_NOMEM, "hipMalloc failed"); }
    if (hipHostMalloc((void**)&h->pb_over, sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) { h->pb_over = nullptr; mipgen_accel_destroy(h); return fail(MIPGEN_E_NOMEM, "hipHostMalloc failed"); }
    *h->pb_over = 0u;
    hipError_t e3 = hipMemcpy(h->dp, &D, sizeof D, hipMemcpyHostToDevice);
    if (e3 == hipSuccess) e3 = hipMemcpy(h->dconsts, &h->hconsts, sizeof(HostConsts), hipMemcpyHostToDevice);
    if (e3 != hipSuccess) { mipgen_accel_destroy(h); return fail(MIPGEN_E_HIP, "hipMemcpy of the run parameters: %s", hipGetErrorString(e3)); }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cu = cus; }
    *out = h;
    return MIPGEN_OK;
}

void mipgen_accel_destroy(mipgen_accel* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->model.release(); h->model_t.release(); h->sv_norm.release(); h->sv_coef.release(); h->sv_center.release(); h->regions.release(); h->bases.release(); h->unmap.release(); h->copy.release();
    h->log_tiles.release(); h->svr_tiles.release(); h->ld_tiles.release(); h->scores.release(); h->records.release();
    h->emitted.release(); h->survivors.release(); h->emitted_per_region.release(); h->pos_region.release(); h->pos_local.release();
    h->letters.release(); h->fmt_regions.release(); h->fmt_pool.release(); h->fmt_text.release(); h->fmt_temp.release();
    h->fmt_a.release(); h->fmt_b.release(); h->fmt_c.release(); h->fmt_d.release();
    h->region_pos0.release(); h->region_base0.release(); h->col_tiles.release(); h->collapsed.release();
    h->cand_in.release(); h->cand_scores.release(); h->cand_feats.release(); h->cand_records.release(); h->cand_ints.release();
    h->win_img.release();
    h->pb_cands.release(); h->pb_idx.release(); h->pb_scores.release(); h->pb_count.release(); h->surv_svr.release(); h->rs_keep.release(); h->rs_offs.release(); h->rs_idx.release();
    h->sat_idx.release(); h->sat_count.release(); h->sat_cands.release(); h->sat_scores.release();
    h->svr_tiles_lvl.release(); h->svr_tiles_kept.release(); h->run_bounds.release(); h->run_pbs.release(); h->run_state.release(); h->run_keep.release();
    h->run_offs.release(); h->skip_count.release();
    h->lrc_seq.release(); h->lrc_out.release(); h->lrc_offs.release(); h->lrc_lens.release(); h->lrc_denoms.release(); h->partials.release();
    h->pool.clear();
    if (h->dp) (void)hipFree(h->dp);
    if (h->dconsts) (void)hipFree(h->dconsts);
    if (h->pb_over) (void)hipHostFree(h->pb_over);
    for (hipEvent_t e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// ---- model ----------------------------------------------------------------------------------------------------
// The list scorer's view of the model (kernels_svr_gemm.hip): support vectors centred on their mean and transposed to [feature][SV], the
// centred squared norms (+ the squares of libsvm indices > 192, which candidates never have: svm.cpp:359-363), the coefficients; the SV
// count padded to a multiple of 64 with zero-coefficient columns.
static int build_list_model(mipgen_accel* h, const std::vector<double>& rows, int n_sv)
{
    const int pad = std::max(64, (n_sv + 63) / 64 * 64);
    std::vector<double> center(MIPGEN_N_FEATURES, 0.0), mt((size_t)MIPGEN_N_FEATURES * pad, 0.0), sn((size_t)pad, 0.0), cf((size_t)pad, 0.0);
    for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
        double s = 0.0;
        for (int i = 0; i < n_sv; i++) s += rows[(size_t)i * SV_ROW + j];
        center[(size_t)j] = n_sv ? s / n_sv : 0.0;
    }
    for (int i = 0; i < n_sv; i++) {
        double q = 0.0;
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
            const double v = rows[(size_t)i * SV_ROW + j] - center[(size_t)j];
            mt[(size_t)j * pad + i] = v;
            q += v * v;
        }
        sn[(size_t)i] = q + rows[(size_t)i * SV_ROW + SVR_N_EXTRA];
        cf[(size_t)i] = rows[(size_t)i * SV_ROW + SVR_COEF];
    }
    if (h->model_t.reserve(mt.size()) || h->sv_norm.reserve(sn.size()) || h->sv_coef.reserve(cf.size()) || h->sv_center.reserve(center.size())) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpy(h->model_t.p, mt.data(), mt.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sv_norm.p, sn.data(), sn.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sv_coef.p, cf.data(), cf.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sv_center.p, center.data(), center.size() * sizeof(double), hipMemcpyHostToDevice));
    h->n_sv_pad = pad;
    return MIPGEN_OK;
}

int mipgen_accel_set_model(mipgen_accel* h, int32_t n_sv, double gamma, double rho, const double* coef, const double* sv)
{
    if (!h || n_sv < 0 || (n_sv > 0 && (!coef || !sv))) return fail(MIPGEN_E_INVALID, "bad model arguments");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<double> rows((size_t)std::max(n_sv, 1) * SV_ROW, 0.0);
    double s_guard = 0.0;
    for (int i = 0; i < n_sv; i++) {
        double* r = &rows[(size_t)i * SV_ROW];
        const double* x = sv + (size_t)i * MIPGEN_N_FEATURES;
        double ne = 0, ni = 0, nl = 0, nj = 0, tot = 0;
        for (int j = 0; j < MIPGEN_N_FEATURES; j++) {
            r[j] = x[j];
            double q = x[j] * x[j];
            tot += q;                                     // index order, as the sparse merge walk accumulates (svm.cpp:329-368)
            if (j <= 20) ne += q;
            else if (j >= F_INS && j <= 150) ni += q;
            else if (j >= F_LIG && j <= 172) nl += q;
            else if (j >= F_JUNC && j <= 189) nj += q;
        }
        r[SVR_COEF] = coef[i]; r[SVR_N_EXT] = ne; r[SVR_N_INS] = ni; r[SVR_N_LIG] = nl; r[SVR_N_JUNC] = nj;
        r[SVR_N_TOTAL] = tot; r[SVR_N_EXTRA] = 0.0;
        // what one base adds to a block's 1-mer window sum: its 1-mer value, plus the GC-fraction value for C and G (mer order A, AA, .., C, ..:
        // SVMipv4.cpp:69-70; the GC fraction sits before the "T" entry)
        const int blk_base[3] = {F_EXT, F_INS, F_LIG}, blk_gc[3] = {F_EXT_GC, F_INS_GC, F_LIG_GC}, blk_step[3] = {5, 21, 5}, blk_thr[3] = {15, 63, 15};
        for (int b = 0; b < 3; b++)
            for (int c = 0; c < 4; c++) {
                const int idx = blk_step[b] * c;
                r[SVR_C1 + 4 * b + c] = x[blk_base[b] + idx + (idx >= blk_thr[b])] + ((c == BASE_C || c == BASE_G) ? x[blk_gc[b]] : 0.0);
            }
        s_guard += coef[i] * exp(-gamma * tot);
    }
    if (h->model.reserve(rows.size())) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpy(h->model.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    h->n_sv = n_sv; h->gamma = gamma; h->rho = rho; h->s_guard = s_guard - rho;
    h->sum_abs_coef = 0.0;
    for (int i = 0; i < n_sv; i++) h->sum_abs_coef += fabs(coef[i]);
    return build_list_model(h, rows, n_sv);
}

// libsvm 3.17 text model (grammar: svm.cpp:2779-2962)
int mipgen_accel_load_model_file(mipgen_accel* h, const char* path)
{
    if (!h || !path) return fail(MIPGEN_E_INVALID, "null argument");
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(MIPGEN_E_MODEL, "cannot open model file %s (the reference would dereference NULL here, svm.cpp:2507)", path);
    char cmd[128];
    std::string svm_type, kernel_type;
    double gamma = 0, rho = 0;
    int nr_class = 2, total_sv = -1;
    bool ok = true, seen_sv = false;
    while (ok && fscanf(fp, "%127s", cmd) == 1) {
        std::string c(cmd);
        if (c == "svm_type") { ok = fscanf(fp, "%127s", cmd) == 1; svm_type = cmd; }
        else if (c == "kernel_type") { ok = fscanf(fp, "%127s", cmd) == 1; kernel_type = cmd; }
        else if (c == "degree") { int d; ok = fscanf(fp, "%d", &d) == 1; }
        else if (c == "gamma") ok = fscanf(fp, "%lf", &gamma) == 1;
        else if (c == "coef0") { double d; ok = fscanf(fp, "%lf", &d) == 1; }
        else if (c == "nr_class") ok = fscanf(fp, "%d", &nr_class) == 1;
        else if (c == "total_sv") ok = fscanf(fp, "%d", &total_sv) == 1;
        else if (c == "rho") { int n = nr_class * (nr_class - 1) / 2; for (int i = 0; i < n && ok; i++) { double r; ok = fscanf(fp, "%lf", &r) == 1; if (i == 0) rho = r; } }
        else if (c == "label" || c == "nr_sv") { for (int i = 0; i < nr_class && ok; i++) { int d; ok = fscanf(fp, "%d", &d) == 1; } }
        else if (c == "probA" || c == "probB") { int n = nr_class * (nr_class - 1) / 2; for (int i = 0; i < n && ok; i++) { double d; ok = fscanf(fp, "%lf", &d) == 1; } }
        else if (c == "SV") { int ch; while ((ch = getc(fp)) != EOF && ch != '\n') {} seen_sv = true; break; }
        else { fclose(fp); return fail(MIPGEN_E_MODEL, "unknown text in model file: [%s]", cmd); }
    }
    if (!ok || !seen_sv || total_sv < 0) { fclose(fp); return fail(MIPGEN_E_MODEL, "malformed model header in %s", path); }
    if (svm_type != "epsilon_svr" && svm_type != "nu_svr") { fclose(fp); return fail(MIPGEN_E_MODEL, "svm_type %s unsupported (SVR only)", svm_type.c_str()); }
    if (kernel_type != "rbf") { fclose(fp); return fail(MIPGEN_E_MODEL, "kernel_type %s unsupported (rbf only)", kernel_type.c_str()); }
    std::vector<double> sv((size_t)std::max(total_sv, 1) * MIPGEN_N_FEATURES, 0.0), coef((size_t)std::max(total_sv, 1), 0.0), extra((size_t)std::max(total_sv, 1), 0.0);
    char* line = nullptr; size_t cap = 0;
    int got = 0;
    while (got < total_sv && getline(&line, &cap, fp) >= 0) {
        char* save = nullptr;
        char* p = strtok_r(line, " \t\n", &save);
        if (!p) continue;
        coef[got] = strtod(p, nullptr);
        for (int k = 1; k < nr_class - 1; k++) strtok_r(nullptr, " \t", &save);
        for (;;) {
            char* idx = strtok_r(nullptr, ":", &save);
            char* val = strtok_r(nullptr, " \t\n", &save);
            if (!val) break;
            long j = strtol(idx, nullptr, 10);
            double v = strtod(val, nullptr);
            if (j >= 1 && j <= MIPGEN_N_FEATURES) sv[(size_t)got * MIPGEN_N_FEATURES + (j - 1)] = v;
            else if (j > MIPGEN_N_FEATURES) extra[got] += v * v;   // x has no such index: contributes sv^2 (svm.cpp:359-363)
        }
        got++;
    }
    free(line);
    fclose(fp);
    if (got != total_sv) return fail(MIPGEN_E_MODEL, "model file has %d SV lines, header says %d", got, total_sv);
    int rc = mipgen_accel_set_model(h, total_sv, gamma, rho, coef.data(), sv.data());
    if (rc) return rc;
    bool any_extra = false;
    for (double x : extra) any_extra |= x != 0.0;
    if (any_extra) {
        std::vector<double> rows((size_t)total_sv * SV_ROW);
        HIP_TRY(hipMemcpy(rows.data(), h->model.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
        double sg = 0;
        for (int i = 0; i < total_sv; i++) {
            rows[(size_t)i * SV_ROW + SVR_N_EXTRA] = extra[i];
            rows[(size_t)i * SV_ROW + SVR_N_TOTAL] += extra[i];
            sg += coef[i] * exp(-gamma * rows[(size_t)i * SV_ROW + SVR_N_TOTAL]);
        }
        HIP_TRY(hipMemcpy(h->model.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
        h->s_guard = sg - rho;
        return build_list_model(h, rows, total_sv);
    }
    return MIPGEN_OK;
}

int mipgen_accel_model_info(const mipgen_accel* h, int32_t* n_sv, double* gamma, double* rho)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    if (n_sv) *n_sv = h->n_sv;
    if (gamma) *gamma = h->gamma;
    if (rho) *rho = h->rho;
    return MIPGEN_OK;
}

// ---- region batch ---------------------------------------------------------------------------------------------
int mipgen_accel_set_window_candidates(mipgen_accel* h, int64_t max_candidates)
{
    if (!h || max_candidates < 0) return fail(MIPGEN_E_INVALID, "bad arguments");
    h->window_cap = max_candidates;
    return MIPGEN_OK;
}

int mipgen_accel_set_window_breaks(mipgen_accel* h, const int32_t* first_regions, int32_t n)
{
    if (!h || n < 0 || (n > 0 && !first_regions)) return fail(MIPGEN_E_INVALID, "bad arguments");
    for (int i = 0; i < n; i++)
        if (first_regions[i] < 0 || (i > 0 && first_regions[i] <= first_regions[i - 1])) return fail(MIPGEN_E_INVALID, "window breaks must be ascending batch indices");
    h->window_breaks.assign(first_regions, first_regions + n);
    return MIPGEN_OK;
}

int mipgen_accel_set_print_exact(mipgen_accel* h, int32_t on)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->print_exact = on != 0;
    return MIPGEN_OK;
}

int mipgen_accel_set_logistic_subruns(mipgen_accel* h, int32_t n)
{
    if (!h || n < 0 || n > 8) return fail(MIPGEN_E_INVALID, "bad arguments");
    h->ld_subruns = n;
    h->logistic_tiles_ready = false;                                   // laid out again on the next logistic call
    return MIPGEN_OK;
}

int mipgen_accel_set_sv_split(mipgen_accel* h, int32_t n_split)
{
    if (!h || n_split < 0 || n_split > 64) return fail(MIPGEN_E_INVALID, "n_split must be in [0, 64] (0 = automatic)");
    h->sv_split = n_split;
    return MIPGEN_OK;
}


int mipgen_accel_upload_regions(mipgen_accel* h, const mipgen_region* regions, int32_t n, mipgen_grid* grids_out)
{
    if (!h || n < 0 || (n > 0 && !regions)) return fail(MIPGEN_E_INVALID, "bad arguments");
    DIAG_CLOCK("upload");
    HIP_TRY(hipSetDevice(h->device));
    const mipgen_params& P = h->params;
    const DevParams& D = h->hp;
    h->cur_window = -1;
    h->windows.clear();
    h->hregions.assign((size_t)n, DevRegion());
    h->grids.assign((size_t)n, mipgen_grid());
    int64_t seq_total = 0, copy_total = 0, unmap_total = 0, cand_total = 0, pos_total = 0, cand_max = 0;
    // copy tables the handle counted itself (mipgen_accel_count_oligo_copies_resident): the batch must be the one they were counted for
    int n_resident = 0;
    for (int i = 0; i < n; i++) n_resident += regions[i].copy == MIPGEN_COPY_RESIDENT;
    if (n_resident) {
        bool same = n_resident == n && h->resident_lens.size() == (size_t)n;
        for (int i = 0; same && i < n; i++) same = h->resident_lens[(size_t)i] == regions[i].seq_len;
        if (!same) return fail(MIPGEN_E_INVALID, "MIPGEN_COPY_RESIDENT: the batch is not the one mipgen_accel_count_oligo_copies_resident counted (regions, order, seq_len)");
    } else {
        h->resident_lens.clear();                                      // `copy` is about to be overwritten (or unused)
    }
    for (int i = 0; i < n; i++) {
        const mipgen_region& R = regions[i];
        if (!R.seq || R.seq_len <= 0) return fail(MIPGEN_E_INVALID, "region %d has no sequence", i);
        if (R.seq_stop - R.seq_start + 1 != R.seq_len) return fail(MIPGEN_E_INVALID, "region %d: seq_len %d != seq_stop-seq_start+1 = %d", i, R.seq_len, R.seq_stop - R.seq_start + 1);
        DevRegion& d = h->hregions[i];
        mipgen_grid& g = h->grids[i];
        grid_of(P, D, R, &g);
        g.offset = cand_total;
        d.seq_off = seq_total;
        d.seq_len = R.seq_len; d.seq_start = R.seq_start; d.seq_stop = R.seq_stop;
        d.start_fl = R.start_flanked; d.stop_fl = R.stop_flanked;
        d.first_pos = g.first_pos; d.n_pos = g.n_pos; d.k0 = g.first_size_index; d.n_sizes = g.n_sizes;
        d.copy_off = R.copy ? copy_total : -1;
        d.unmap_off = R.unmappable ? unmap_total : -1;
        memcpy(d.lrc, R.long_range_content, sizeof d.lrc);
        seq_total += R.seq_len;
        if (R.copy) copy_total += (int64_t)D.n_len_slots * R.seq_len;
        if (R.unmappable) unmap_total += (int64_t)D.n_sizes_all * R.seq_len;
        cand_total += g.count;
        pos_total += g.n_pos;
        cand_max = std::max(cand_max, g.count);
    }
    if (pos_total > INT32_MAX) return fail(MIPGEN_E_INVALID, "batch has %lld scan positions (max %d): split the design", (long long)pos_total, INT32_MAX);
    DIAG_LAP("layout");
    // encode + pack on the host
    uint8_t code_lut[256];
    for (int c = 0; c < 256; c++) code_lut[c] = base_code((char)c);
    std::vector<uint8_t> hb((size_t)std::max<int64_t>(seq_total, 1));
    std::vector<char> hl((size_t)std::max<int64_t>(seq_total, 1));
    std::vector<uint8_t> hu((size_t)std::max<int64_t>(unmap_total, 1));
    for (int i = 0; i < n; i++) {
        const mipgen_region& R = regions[i];
        const DevRegion& d = h->hregions[i];
        uint8_t* b = &hb[(size_t)d.seq_off];
        memcpy(&hl[(size_t)d.seq_off], R.seq, (size_t)R.seq_len);
        const char* ms = R.masked_seq ? R.masked_seq : R.seq;
        if (R.snp_class) for (int k = 0; k < R.seq_len; k++) b[k] = (uint8_t)(code_lut[(uint8_t)R.seq[k]] | (ms[k] == 'N' ? BASE_MASKED_BIT : 0) | ((R.snp_class[k] & 3) << BASE_SNP_SHIFT));
        else for (int k = 0; k < R.seq_len; k++) b[k] = (uint8_t)(code_lut[(uint8_t)R.seq[k]] | (ms[k] == 'N' ? BASE_MASKED_BIT : 0));
        if (R.unmappable) memcpy(&hu[(size_t)d.unmap_off], R.unmappable, (size_t)D.n_sizes_all * R.seq_len);
    }
    DIAG_LAP("pack");
    // inputs -> HBM (everything the kernels read stays resident for the whole batch)
    if (h->letters.reserve(hl.size())) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->letters.p, hl.data(), hl.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->regions.reserve((size_t)std::max(n, 1)) || h->bases.reserve(hb.size()) || h->copy.reserve((size_t)std::max<int64_t>(copy_total, 1)) || h->unmap.reserve(hu.size()) ||
        h->survivors.reserve((size_t)std::max<int64_t>(2 * pos_total, 1)) || h->emitted_per_region.reserve((size_t)std::max(n, 1)) ||
        h->pos_region.reserve((size_t)std::max<int64_t>(pos_total, 1)) || h->pos_local.reserve((size_t)std::max<int64_t>(pos_total, 1)))
        return MIPGEN_E_NOMEM;

    DIAG_LAP("input buffers");
    // ---- result windows: consecutive regions, at most `cap` candidates each (17 B per candidate: score, record, emitted flag) ----
    int64_t cap = h->window_cap;
    if (cap <= 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const int64_t have = (int64_t)(h->scores.cap * 8 + h->records.cap * 8 + h->emitted.cap) + (int64_t)free_b + (int64_t)h->pool.held();   // what the result arrays may grow into
        cap = std::max<int64_t>(1, (int64_t)((double)have * 0.85) / 17);
        // 2^30 candidates keep every kernel of a window busy for milliseconds to tenths of a second; result arrays beyond that only cost
        // allocation time (18 GB: 0.4-1 s on a cold device; 36 GB: 2 s) and delay the first window a pipelined caller can work on
        // (hipMalloc of 250 GB takes seconds) - unless one region alone is larger
        cap = std::min<int64_t>(cap, std::max<int64_t>((int64_t)1 << 30, cand_max));
    }
    if (cand_max > cap && h->window_cap > 0) cap = cand_max;       // a region is never split: one oversized region is its own window
    if (cand_max > cap) return fail(MIPGEN_E_NOMEM, "a single region has %lld dense candidates; device memory holds %lld", (long long)cand_max, (long long)cap);
    {
        Window w;
        size_t nb = 0;                                             // next forced break (mipgen_accel_set_window_breaks)
        for (int i = 0; i < n; i++) {
            const mipgen_grid& g = h->grids[i];
            while (nb < h->window_breaks.size() && h->window_breaks[nb] < i) nb++;
            const bool forced = nb < h->window_breaks.size() && h->window_breaks[nb] == i;
            if (w.r1 > w.r0 && (forced || w.n_cand + g.count > cap)) {
                h->windows.push_back(w);
                Window nw; nw.r0 = nw.r1 = i; nw.cand0 = w.cand0 + w.n_cand; nw.pos0 = w.pos0 + w.n_pos;
                w = nw;
            }
            h->hregions[i].out_off = w.n_cand;                     // window-relative: the result arrays are reused window after window
            w.n_cand += g.count; w.n_pos += g.n_pos; w.r1 = i + 1;
        }
        h->windows.push_back(w);                                   // an empty batch is one empty window
    }
    int64_t win_cand_max = 0;
    for (const Window& w : h->windows) win_cand_max = std::max(win_cand_max, w.n_cand);

    // ---- collapse layout: bases a survivor's scan target can cover, per region ----
    std::vector<int64_t> rpos0((size_t)n + 1, 0);
    h->h_region_base0.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) {
        const DevRegion& d = h->hregions[i];
        const int max_scan = D.max_capture - d.k0 * D.inc - D.min_sum;
        const int64_t nb = (d.n_pos > 0 && d.n_sizes > 0) ? (int64_t)d.n_pos + max_scan - 1 : 0;
        rpos0[(size_t)i + 1] = rpos0[(size_t)i] + d.n_pos;
        h->h_region_base0[(size_t)i + 1] = h->h_region_base0[(size_t)i] + 2 * nb;
    }
    // collapse tiles: 128 bases each.  The scoring kernels' tile lists depend on the method and are laid out on first use (ensure_tiles)
    std::vector<CollapseTile> ct;
    for (Window& w : h->windows) {
        w.col_tile0 = (int)ct.size();
        w.base0 = h->h_region_base0[(size_t)w.r0]; w.n_base_entries = h->h_region_base0[(size_t)w.r1] - w.base0;
        w.log_tile0 = w.n_log_tiles = w.svr_tile0 = w.n_svr_tiles = w.ld_tile0 = w.n_ld_tiles = 0;
        for (int i = w.r0; i < w.r1; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            for (int64_t j0 = 0; 2 * j0 < h->h_region_base0[(size_t)i + 1] - h->h_region_base0[(size_t)i]; j0 += 128) { CollapseTile t = {i, (int32_t)j0}; ct.push_back(t); }
        }
        w.n_col_tiles = (int)ct.size() - w.col_tile0;
    }
    h->record_tiles_ready = h->logistic_tiles_ready = h->svr_tiles_ready = false;
    h->svr_batch_error.clear();
    h->ld_lds = 0; h->log_span_max = 0; h->svr_lds = 0;
    DIAG_LAP("tiles");
    if (h->scores.reserve((size_t)std::max<int64_t>(win_cand_max, 1)) || h->records.reserve((size_t)std::max<int64_t>(win_cand_max, 1)) ||
        h->emitted.reserve((size_t)std::max<int64_t>(win_cand_max, 1)))
        return MIPGEN_E_NOMEM;
    if (h->region_pos0.reserve((size_t)n + 1) || h->region_base0.reserve((size_t)n + 1) || h->col_tiles.reserve(std::max<size_t>(ct.size(), 1)) ||
        h->collapsed.reserve((size_t)std::max<int64_t>(h->h_region_base0[(size_t)n], 1)))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(h->region_pos0.p, rpos0.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->region_base0.p, h->h_region_base0.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    if (!ct.empty()) HIP_TRY(hipMemcpyAsync(h->col_tiles.p, ct.data(), ct.size() * sizeof(CollapseTile), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mipgen_launch_fill_pos_map(h->stream, h->region_pos0.p, n, pos_total, h->pos_region.p, h->pos_local.p));       // position -> (region, position in it)
    if (n > 0) HIP_TRY(hipMemcpyAsync(h->regions.p, h->hregions.data(), (size_t)n * sizeof(DevRegion), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->bases.p, hb.data(), hb.size(), hipMemcpyHostToDevice, h->stream));
    if (copy_total && !n_resident) {
        // copy tables: packed region by region into two pinned chunks and sent as they fill - no host image of the whole table (6.7 GB for the exome)
        int64_t max_block = 0;
        for (int i = 0; i < n; i++) if (regions[i].copy) max_block = std::max(max_block, (int64_t)D.n_len_slots * regions[i].seq_len);
        const int64_t chunk = std::min(copy_total, std::max<int64_t>((int64_t)1 << 20, max_block));          // int32 elements: 4 MiB, or the largest region's block
        PinnedPair<int32_t> pin;
        HIP_TRY(pin.alloc((size_t)chunk));
        int cur = 0;
        int64_t fill = 0, base = 0;
        auto flush = [&]() -> hipError_t {
            if (fill == 0) return hipSuccess;
            hipError_t e = hipMemcpyAsync(h->copy.p + base, pin.buf[cur], (size_t)fill * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = hipEventRecord(pin.done[cur], h->stream);
            if (e != hipSuccess) return e;
            pin.busy[cur] = true;
            base += fill; fill = 0; cur ^= 1;
            return pin.wait(cur);
        };
        for (int i = 0; i < n; i++) {
            const mipgen_region& R = regions[i];
            if (!R.copy) continue;
            const int64_t block = (int64_t)D.n_len_slots * R.seq_len;
            if (fill + block > chunk) HIP_TRY(flush());
            if (h->hregions[i].copy_off != base + fill) return fail(MIPGEN_E_INVALID, "internal: copy table offsets out of order");
            for (int len = 0; len <= MIPGEN_MAX_OLIGO; len++) {
                const int s = D.len_slot[len];
                if (s < 0) continue;
                int32_t* dst = pin.buf[cur] + fill + (int64_t)s * R.seq_len;
                if (R.copy[len]) memcpy(dst, R.copy[len], (size_t)R.seq_len * sizeof(int32_t));
                else memset(dst, 0, (size_t)R.seq_len * sizeof(int32_t));
            }
            fill += block;
        }
        HIP_TRY(flush());
        HIP_TRY(pin.wait(0)); HIP_TRY(pin.wait(1));
    }
    if (unmap_total) HIP_TRY(hipMemcpyAsync(h->unmap.p, hu.data(), hu.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));             // host staging vectors die here
    DIAG_LAP("buffers + copies");
    h->n_regions = n; h->n_cand = cand_total; h->total_pos = pos_total;
    h->scored = false; h->replayed = false; h->win_state.assign(h->windows.size(), 0); h->fmt_bytes = -1;
    h->ev_used.assign(h->windows.size(), 0);
#ifdef MIPGEN_DIAG
    fprintf(stderr, "[mipgen_accel] batch: %d regions, %lld candidates in %zu window(s)\n", n, (long long)cand_total, h->windows.size());
#endif
    if (grids_out) memcpy(grids_out, h->grids.data(), (size_t)n * sizeof(mipgen_grid));
    return MIPGEN_OK;
}

int64_t mipgen_accel_batch_candidates(const mipgen_accel* h) { return h ? h->n_cand : 0; }
int32_t mipgen_accel_window_count(const mipgen_accel* h) { return h ? (int32_t)h->windows.size() : 0; }

int mipgen_accel_window_info(const mipgen_accel* h, int32_t w, int32_t* first_region, int32_t* n_regions, int64_t* first_candidate,
                             int64_t* n_candidates, int64_t* first_position, int64_t* n_positions)
{
    if (!h || w < 0 || w >= (int32_t)h->windows.size()) return fail(MIPGEN_E_INVALID, "window %d out of range", w);
    const Window& W = h->windows[(size_t)w];
    if (first_region) *first_region = W.r0;
    if (n_regions) *n_regions = W.r1 - W.r0;
    if (first_candidate) *first_candidate = W.cand0;
    if (n_candidates) *n_candidates = W.n_cand;
    if (first_position) *first_position = W.pos0;
    if (n_positions) *n_positions = W.n_pos;
    return MIPGEN_OK;
}

// Support-vector split of a dense SVR launch: a batch with few tiles is cut along the SV list so that it still fills the chip.
// Cost model in units of one SV-group iteration of a tile: a work unit pays the tile set-up (PH0) plus its share of the groups;

double mipgen_accel_last_kernel_ms(mipgen_accel* h, int32_t which)
{
    if (h && which == 4) return h->kmer_count_ms;
    if (h && which == 5) return h->list_svr_ms;
    if (h && which == 6) return h->list_feat_ms;
    if (!h || !h->timing) return -1.0;
    if (hipSetDevice(h->device) != hipSuccess) return -1.0;
    double total = 0.0;
    bool any = false;
    for (size_t w = 0; w < h->windows.size() && 4 * (w + 1) <= h->ev.size(); w++) {
        const uint8_t used = h->ev_used[w];
        if (!(used & 1)) continue;
        hipEvent_t* ev = &h->ev[4 * w];
        float ms = 0.f;
        hipError_t e = hipSuccess;
        if (which == 3) {                                                        // replay + condense (incl. its memsets)
            if (!(used & 2)) continue;
            if (hipEventSynchronize(ev[3]) != hipSuccess) return -1.0;
            e = hipEventElapsedTime(&ms, ev[2], ev[3]);
        } else {
            if (hipEventSynchronize(ev[2]) != hipSuccess) return -1.0;
            if (which == 0) e = hipEventElapsedTime(&ms, ev[1], ev[2]);          // SVR kernel (~0 for logistic)
            else if (which == 2) e = hipEventElapsedTime(&ms, ev[0], ev[1]);     // records / logistic kernel
            else e = hipEventElapsedTime(&ms, ev[0], ev[2]);
        }
        if (e != hipSuccess) return -1.0;
        total += ms; any = true;
    }
    return any ? total : -1.0;
}

int mipgen_accel_set_timing(mipgen_accel* h, int32_t enabled)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->timing = enabled != 0;
    std::fill(h->ev_used.begin(), h->ev_used.end(), 0);
    return MIPGEN_OK;
}

}  // extern "C"
