// accel_tiles.hip — tile lists of the scoring kernels (k_records / k_logistic_dense / k_svr_dense), laid out on the host when a scoring method is
// first used on the resident batch; the support-vector split of small dense SVR launches.
#include "accel_internal.h"

extern "C" {

int mipgen_pick_sv_split(int n_tiles, int n_sv, int n_cu)
{
    const int groups = (n_sv + SVR_GROUP - 1) / SVR_GROUP;
    const double PH0 = 6.0;
    int best = 1;
    double best_cost = 0;
    for (int s = 1; s <= 8; s++) {
        if (s > 1 && groups / s < 32) break;
        const int64_t units = (int64_t)n_tiles * s, rounds = (units + n_cu - 1) / n_cu;
        const double cost = (double)rounds * (PH0 + (double)((groups + s - 1) / s));
        if (s == 1 || cost < best_cost * 0.98) { best = s; best_cost = cost; }
    }
    return best;
}

int mipgen_ensure_events(mipgen_accel* h)
{
    const size_t want = 4 * h->windows.size();
    while (h->ev.size() < want) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    return 0;
}

// ---- tile lists of the scoring kernels: laid out when a method is first used on the resident batch (a 200,000-region logistic design
// never prices an SVR tile) ----
static int build_record_tiles(mipgen_accel* h)          // k_records / k_records_logistic: 8 positions per tile
{
    if (h->record_tiles_ready) return MIPGEN_OK;
    const DevParams& D = h->hp;
    const int Lmax = std::max(D.e_max, D.l_max);
    std::vector<LogTile> lt;
    int span_max = 0;
    for (Window& w : h->windows) {
        w.log_tile0 = (int)lt.size();
        for (int i = w.r0; i < w.r1; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            const int Cmax = D.max_capture - d.k0 * D.inc;
            const int NPL = 8;  // 8 positions per records tile: ~10 resident blocks per CU hide the per-candidate gathers (32: 2.6 waves/SIMD, 45 % slower)
            for (int p0 = 0; p0 < d.n_pos; p0 += NPL) {
                LogTile t = {i, p0, std::min(NPL, d.n_pos - p0), 0};
                lt.push_back(t);
                span_max = std::max(span_max, t.np + Cmax + Lmax);
            }
        }
        w.n_log_tiles = (int)lt.size() - w.log_tile0;
    }
    if (h->log_tiles.reserve(std::max<size_t>(lt.size(), 1))) return MIPGEN_E_NOMEM;
    if (!lt.empty()) HIP_TRY(hipMemcpy(h->log_tiles.p, lt.data(), lt.size() * sizeof(LogTile), hipMemcpyHostToDevice));
    h->log_span_max = span_max;
    h->record_tiles_ready = true;
    return MIPGEN_OK;
}

static int build_logistic_tiles(mipgen_accel* h)        // k_logistic_dense: a run of positions with all their capture sizes
{
    if (h->logistic_tiles_ready) return MIPGEN_OK;
    const DevParams& D = h->hp;
    const int Lmax = std::max(D.e_max, D.l_max);
    std::vector<SvrTile> ldt;
    size_t ld_lds = 0;
    bool ld_ok = true;
    const int64_t ld_np_cap = h->total_pos * 2 < 8192 ? 8 : (h->total_pos * 2 < 65536 ? 16 : 64);      // small batches: more, smaller tiles (fill 256 CUs)
    for (Window& w : h->windows) {
        w.ld_tile0 = (int)ldt.size();
        for (int i = w.r0; i < w.r1 && ld_ok; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            const int Cmax = D.max_capture - d.k0 * D.inc;
            // a tile = a run of positions with ALL their capture sizes (the kernel works them off in nkc runs of <= 9, staging the bases and
            // their prefix words once): sized for the widest run's tables and the first run's reach
            const int nkc = (d.n_sizes + 8) / 9;
            int kc_max = 0;
            for (int c = 0; c < nkc; c++) kc_max = std::max(kc_max, (int)((int64_t)d.n_sizes * (c + 1) / nkc) - (int)((int64_t)d.n_sizes * c / nkc));
            const int ssr = (kc_max - 1) * D.inc + D.max_sum - D.min_sum + 1, ssmax = Cmax - D.min_sum;
            const int ssmin_all = Cmax - (d.n_sizes - 1) * D.inc - D.max_sum;
            // the positions of a tile are worked off in `subs` sub-runs of np, the downstream-arm table sliding along (only np of its
            // np + ssr - 1 window starts are new per sub-run) and the bases / prefix words staged once; small batches keep one sub-run per tile
            // (they need every workgroup they can get)
            const int subs = h->ld_subruns > 0 ? h->ld_subruns : (h->total_pos >= 1000000 ? 3 : (h->total_pos >= 25000 ? 2 : 1));   // measured: 24 x 5 kb best at 2, 8,192 exons at 3
            int np = (int)std::min<int64_t>({ld_np_cap, (int64_t)d.n_pos, 64});
            size_t b = 0;
            for (; np >= 1; np--) {
                const int np_all = std::min(subs * np, d.n_pos);
                b = std::max(mipgen_logistic_dense_lds_bytes(np_all, np, ssr, ssmax, Lmax, D.e_max - D.e_min + 1, D.l_max - D.l_min + 1),
                             mipgen_logistic_dense_lds_bytes(np_all, np, ssr, ssmax, Lmax, D.l_max - D.l_min + 1, D.e_max - D.e_min + 1));
                if (b <= 80 * 1024) break;                             // two 512-thread workgroups per compute unit: one builds tables while the other scores
            }
            if (np < 1 || ssmin_all < 1) { ld_ok = false; break; }
            ld_lds = std::max(ld_lds, b);
            // the sub-run length travels in the tile's strand field (these tiles hold both strands)
            for (int p0 = 0; p0 < d.n_pos; p0 += subs * np) { SvrTile t = {i, np, p0, std::min(subs * np, d.n_pos - p0), 0, d.n_sizes, 0, 0}; ldt.push_back(t); }
        }
        w.n_ld_tiles = (int)ldt.size() - w.ld_tile0;
    }
    if (!ld_ok) {                                        // some region does not fit that kernel's LDS: the per-candidate kernel scores the batch
        for (Window& w : h->windows) { w.ld_tile0 = 0; w.n_ld_tiles = 0; }
        h->ld_lds = 0;
        h->logistic_tiles_ready = true;
        return build_record_tiles(h);
    }
    if (h->ld_tiles.reserve(std::max<size_t>(ldt.size(), 1))) return MIPGEN_E_NOMEM;
    if (!ldt.empty()) HIP_TRY(hipMemcpy(h->ld_tiles.p, ldt.data(), ldt.size() * sizeof(SvrTile), hipMemcpyHostToDevice));
    h->ld_lds = ld_lds;
    h->logistic_tiles_ready = true;
    return MIPGEN_OK;
}

static int build_svr_tiles(mipgen_accel* h)             // k_svr_dense (+ the record tiles of k_records, which runs before it)
{
    if (h->svr_tiles_ready) return MIPGEN_OK;
    if (int rc = build_record_tiles(h)) return rc;
    const DevParams& D = h->hp;
    const int Lmax = std::max(D.e_max, D.l_max);
    const int n_arm = std::max(h->geom.n_e, h->geom.n_l) | 1;
    std::vector<SvrTile> st, st_lvl;
    std::vector<double> st_cost;
    size_t svr_lds = 0;
    int max_levels = 1;                                  // capture-size runs of the region with the most of them
    // SVR tiles: capture sizes in nearly equal runs of <= 9, positions in runs as long as the tile's LDS allows (more positions per tile =
    // fewer factor-table entries per candidate and fewer idle candidate lanes).  Every split of the sizes is priced with the kernel's
    // instruction budget - ~47 VALU per (table entry, SV) against ~2.7 per (candidate, SV) at full lanes - and the cheapest one is laid
    // out.  The choice depends on (first size, number of sizes, positions) only: exome-shaped batches repeat the same few thousand shapes.
    struct Run { int ki0, kc, np; };
    struct Shape { std::vector<Run> runs; size_t lds = 0; };
    std::unordered_map<uint64_t, Shape> shapes;
    auto shape_of = [&](const DevRegion& d, bool few) -> const Shape& {
        const int lanes = 64 * (few ? h->geom_few.wpc : h->geom.wpc);
        const uint64_t key = ((uint64_t)(uint32_t)d.n_pos << 32) | ((uint64_t)(uint16_t)d.k0 << 16) | (uint64_t)(uint16_t)d.n_sizes | (few ? (uint64_t)1 << 15 : 0);
        auto it = shapes.find(key);
        if (it != shapes.end()) return it->second;
        Shape best;
        double best_cost = 0;
        const int Cmax = D.max_capture - d.k0 * D.inc;
        for (int kc_cap = std::min(9, d.n_sizes); kc_cap >= 1; kc_cap--) {
            const int nkc = (d.n_sizes + kc_cap - 1) / kc_cap;
            if (kc_cap < std::min(9, d.n_sizes) && nkc == (d.n_sizes + kc_cap) / (kc_cap + 1)) continue;   // same split as the previous cap
            std::vector<Run> runs;
            double ent = 0, slots = 0;
            size_t lds_r = 0;
            bool ok = true;
            for (int c = 0; c < nkc && ok; c++) {
                const int ki0 = (int)((int64_t)d.n_sizes * c / nkc), ki1 = (int)((int64_t)d.n_sizes * (c + 1) / nkc);
                const int kc = ki1 - ki0;
                const int Cmax_t = Cmax - ki0 * D.inc, Cmin_t = Cmax_t - (kc - 1) * D.inc;
                const int ssmax = Cmax_t - D.min_sum, ssmin = Cmin_t - D.max_sum, ssr = ssmax - ssmin + 1;
                int np = std::max(1, std::min(lanes / kc, d.n_pos));       // every lane of a pair chunk owns one (position, capture size)
                size_t lds_t = 0;
                for (; np >= 1; np--) {
                    lds_t = mipgen_svr_lds_bytes_tile(np, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64);
                    if (lds_t <= 160 * 1024) break;
                }
                if (np < 1) { ok = false; break; }
                // np = -inc (mod 32) makes the candidate steps' downstream-factor loads conflict free (see the kernel's lane mapping):
                // taken when it costs few positions
                { const int np_cf = np - ((np + D.inc) % 32); if (np_cf >= 1 && np_cf * 10 >= np * 9) { np = np_cf; lds_t = mipgen_svr_lds_bytes_tile(np, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64); } }
                // the same number of tiles, evenly filled (the last tile of a region is not a stub that costs a full table stage)
                { const int nt = (d.n_pos + np - 1) / np, np_even = (d.n_pos + nt - 1) / nt;
                  if (np_even < np) { np = np_even; lds_t = mipgen_svr_lds_bytes_tile(np, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64); } }
                runs.push_back({ki0, kc, np});
                lds_r = std::max(lds_r, lds_t);
                const double tiles = std::ceil((double)d.n_pos / np);
                ent += tiles * (np * (h->geom.n_e + h->geom.n_l) / 2.0 + (np + ssr - 1) * (h->geom.n_e + h->geom.n_l) / 2.0 + (double)np * ssr);
                slots += tiles * lanes;
            }
            if (!ok) continue;
            const double cost = 47.0 * ent + 2.7 * slots * D.n_pairs / 1.0;
            if (best.runs.empty() || cost < best_cost) { best.runs = runs; best_cost = cost; best.lds = lds_r; }
        }
        return shapes.emplace(key, std::move(best)).first->second;
    };
    for (Window& w : h->windows) {
        w.svr_tile0 = (int)st.size();
        // the few-sizes geometry is a launch of its own: taken when the window holds enough such regions to fill the chip a few times over
        // (a handful of them stay with the main launch: an extra launch ends with the tail of its last tile)
        int n_few_regions = 0;
        for (int i = w.r0; i < w.r1; i++) if (h->hregions[i].n_pos > 0 && h->hregions[i].n_sizes == 1) n_few_regions++;
        const bool win_few = h->have_few && n_few_regions >= h->n_cu;
        std::vector<uint8_t> st_few;                           // per tile of this window: few-sizes geometry?
        for (int i = w.r0; i < w.r1; i++) {
            const DevRegion& d = h->hregions[i];
            if (d.n_pos <= 0 || d.n_sizes <= 0) continue;
            // Regions of ONE capture size only, and with the shape the main launch would give them (their positions per tile are bound by the LDS -
            // 70-90, inside either geometry's lanes): a score's last bits depend on where its tile starts - the window sums are differences of
            // tile-relative prefix sums -, and a region must score bit-identically in any window or shard, whichever launch takes its tiles.
            const bool few = win_few && d.n_sizes == 1;
            const Shape& shape = shape_of(d, false);
            if (shape.runs.empty()) { svr_lds = (size_t)1 << 30; continue; }
            svr_lds = std::max(svr_lds, shape.lds);
            const int Cmax = D.max_capture - d.k0 * D.inc;
            for (size_t lvl = 0; lvl < shape.runs.size(); lvl++) {
                const Run& r = shape.runs[lvl];
                max_levels = std::max(max_levels, (int)lvl + 1);
                for (int p0 = 0; p0 < d.n_pos; p0 += r.np) {
                    const int npt = std::min(r.np, d.n_pos - p0);
                    // run time of the tile in wavefront-cycles per SV group (measured shares of the three stages): table entries, scan span,
                    // candidate steps (all lanes of the block step, whatever the tile holds)
                    const int Cmax_t = Cmax - r.ki0 * D.inc, ssmax = Cmax_t - D.min_sum, ssr = (r.kc - 1) * D.inc + D.max_sum - D.min_sum + 1;
                    const double ent = (npt + (npt + ssr - 1)) * (h->geom.n_e + h->geom.n_l) / 2.0 + (double)npt * ssr;
                    const double cost = 19.0 * ent + 100.0 * (npt + ssmax + 2 * Lmax) + 75000.0;
                    const int fits = mipgen_svr_scores_fit_lds(npt, r.kc, D.n_pairs, ssr, ssmax, Lmax, n_arm, h->geom.group, h->geom.n_e, h->geom.n_l, h->geom.nchunk * h->geom.wpc * 64);
                    for (int s2 = 0; s2 < 2; s2++) { SvrTile t = {i, s2, p0, npt, r.ki0, r.kc, (int)lvl, fits}; st.push_back(t); st_cost.push_back(cost); st_few.push_back(few ? 1 : 0); }
                }
            }
        }
        // longest tiles first: workgroups are dispatched in index order as compute units free up, so the short tiles fill the end of the
        // launch (k_svr_dense takes tile blockIdx / n_split: consecutive tiles already land on different XCDs)
        {
            const size_t t0 = (size_t)w.svr_tile0, n = st.size() - t0;
            std::vector<size_t> order(n);
            for (size_t k = 0; k < n; k++) order[k] = k;
            // (the tiles of the few-sizes geometry behind the others: the window's second launch)
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return st_few[a] != st_few[b] ? st_few[a] < st_few[b] : st_cost[t0 + a] > st_cost[t0 + b]; });
            std::vector<SvrTile> sorted(n);
            for (size_t k = 0; k < n; k++) sorted[k] = st[t0 + order[k]];
            std::copy(sorted.begin(), sorted.end(), st.begin() + (ptrdiff_t)t0);
            w.n_svr_few = 0;
            for (size_t k = 0; k < n; k++) w.n_svr_few += st_few[k];
        }
        w.n_svr_tiles = (int)st.size() - w.svr_tile0;
        // the same tiles grouped by capture-size run (run 0 first, longest first inside a run): the launch order of the dynamic skip (kernels_skip.hip)
        {
            const size_t t0 = (size_t)w.svr_tile0, n = (size_t)w.n_svr_tiles;
            int wl = 0;
            for (size_t k = 0; k < n; k++) wl = std::max(wl, st[t0 + k].level + 1);
            w.lvl_tile0.assign(1, (int)st_lvl.size());
            for (int lvl = 0; lvl < wl; lvl++) {
                for (size_t k = 0; k < n; k++) if (st[t0 + k].level == lvl) st_lvl.push_back(st[t0 + k]);      // (order kept: run 0 ends with the few-sizes tiles)
                w.lvl_tile0.push_back((int)st_lvl.size());
            }
            w.lvl0_few = w.n_svr_few;                          // regions of one size have one run
        }
    }
    h->svr_batch_error.clear();
    if (svr_lds > 160 * 1024) {
        h->svr_batch_error = "an SVR tile needs more than 160 KiB of LDS: capture range / arm lists too wide";
        st.clear();
        for (Window& w : h->windows) { w.svr_tile0 = 0; w.n_svr_tiles = 0; }
        svr_lds = 0;
    }
    if (h->svr_tiles.reserve(std::max<size_t>(st.size(), 1))) return MIPGEN_E_NOMEM;
    if (!st.empty()) HIP_TRY(hipMemcpy(h->svr_tiles.p, st.data(), st.size() * sizeof(SvrTile), hipMemcpyHostToDevice));
    // the run-ordered copy of the tile list and the runs of every region, for the dynamic skip between capture-size runs (only regions of more
    // than nine capture sizes have a second run)
    h->svr_levels = st.empty() ? 1 : max_levels;
    if (h->svr_levels > 1) {
        std::vector<uint32_t> rb((size_t)h->n_regions * (size_t)max_levels, 0u);
        for (const SvrTile& t : st) rb[(size_t)t.region * (size_t)max_levels + (size_t)t.level] = (uint32_t)t.ki0 | ((uint32_t)t.kc << 16);
        if (h->svr_tiles_lvl.reserve(st_lvl.size()) || h->run_bounds.reserve(rb.size())) return MIPGEN_E_NOMEM;
        HIP_TRY(hipMemcpy(h->svr_tiles_lvl.p, st_lvl.data(), st_lvl.size() * sizeof(SvrTile), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->run_bounds.p, rb.data(), rb.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    h->svr_lds = svr_lds;
    h->svr_tiles_ready = true;
    return MIPGEN_OK;
}

int mipgen_ensure_tiles(mipgen_accel* h, int32_t method)
{
    // a parameter set the tiled SVR kernel cannot take still needs the record tiles: its dense grid goes through the list scorer (svr_window_via_list)
    if (method == MIPGEN_SCORE_SVR) return h->svr_geometry_error.empty() ? build_svr_tiles(h) : build_record_tiles(h);
    return build_logistic_tiles(h);
}

// After a stream synchronisation: did a print-exact re-score list overflow since the last check?  The surplus entries keep the dense kernel's

}  // extern "C"
