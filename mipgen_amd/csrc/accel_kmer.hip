// accel_kmer.hip — SURVEY.md section 8f-3 behind the C ABI: arm-oligo copy numbers by exact k-mer counting and capture-window uniqueness
// (kernels_kmer.hip, kernels_window.hip), the opt-in replacement of the reference's bwa round trips (mipgen.cpp:558-596, 796-873).
#include "accel_internal.h"

extern "C" {

// ---- section 8f-3: arm-oligo copy numbers by exact k-mer counting (opt-in replacement of the bwa round trip) ----------------
namespace {

// device state of one counting run; `out` = int32 [n_k][total] over the concatenated region strings (kernels_kmer.hip: k_kmer_lookup)
struct KmerRun {
    KmerParams KP;
    int64_t total = 0;
    int64_t pad = 0;                          // 'N' bytes behind the concatenation (readers that run past the last region: kernels_window.hip)
    std::vector<int64_t> roff;                // start of every region in the concatenation (one separator after each), then `total`
    DevBuf<char> dq, dg;
    DevBuf<uint64_t> dkeys;
    DevBuf<unsigned int> dcounts;
    DevBuf<uint32_t> dfilter, dfolded;
    DevBuf<int32_t> out;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    explicit KmerRun(DevPool* pool) { dq.pool = dg.pool = dkeys.pool = dcounts.pool = dfilter.pool = dfolded.pool = out.pool = pool; }
    ~KmerRun()
    {
        dq.release(); dg.release(); dkeys.release(); dcounts.release(); dfilter.release(); dfolded.release(); out.release();
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// insert the regions' k-mers, stream the genome past them, look every region position up again: K.out is filled (on the stream) on return
int kmer_count_run(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                   const char* const* region_seqs, const int32_t* region_lens, int32_t n_lengths, const int32_t* lengths, KmerRun& K)
{
    DIAG_CLOCK("oligo copies");
    KmerParams& KP = K.KP;
    memset(&KP, 0, sizeof KP);
    KP.n_k = n_lengths;
    for (int i = 0; i < n_lengths; i++) {
        if (lengths[i] < 1 || lengths[i] > 31 || (i && lengths[i] <= lengths[i - 1])) return fail(MIPGEN_E_INVALID, "oligo lengths must be ascending and <= 31 (exact 2-bit keys)");
        KP.k[i] = lengths[i];
    }
    KP.kmax = lengths[n_lengths - 1];
    // region sequences, separated by one 'N' (no k-mer crosses it)
    int64_t total = 0;
    K.roff.resize((size_t)n_regions + 1);
    for (int r = 0; r < n_regions; r++) {
        if (region_lens[r] < 0 || !region_seqs[r]) return fail(MIPGEN_E_INVALID, "region %d: no sequence", r);
        K.roff[(size_t)r] = total; total += (int64_t)region_lens[r] + 1;
    }
    K.roff[(size_t)n_regions] = total;
    K.total = total;
    std::vector<char> q((size_t)(total + K.pad), 'N');
    for (int r = 0; r < n_regions; r++) memcpy(&q[(size_t)K.roff[(size_t)r]], region_seqs[r], (size_t)region_lens[r]);
    uint64_t cap = 1024;
    while (cap < 2 * (uint64_t)total) cap <<= 1;
    KP.cap_mask = cap - 1;
    // Bloom filter of the regions' canonical kmin-mers: ~32 bits per region position (2-3 % false positives), at least the size of its LDS fold
    KP.filter_bits = 18;
    while (KP.filter_bits < 30 && (1ull << KP.filter_bits) < 32ull * (uint64_t)total) KP.filter_bits++;
    int64_t gmax = 0;
    for (int c = 0; c < n_chrom; c++) gmax = std::max(gmax, chrom_lens[c]);
    const size_t tab = (size_t)cap * (size_t)n_lengths;
    if (K.dq.reserve((size_t)(total + K.pad)) || K.dg.reserve((size_t)std::max<int64_t>(gmax, 1)) || K.dkeys.reserve(tab) || K.dcounts.reserve(tab) ||
        K.dfilter.reserve((size_t)1 << (KP.filter_bits - 5)) || K.dfolded.reserve((size_t)1 << 13) || K.out.reserve((size_t)total * (size_t)n_lengths))
        return MIPGEN_E_NOMEM;
    DIAG_LAP("concatenate + device buffers");
    HIP_TRY(hipEventCreate(&K.e0)); HIP_TRY(hipEventCreate(&K.e1));
    HIP_TRY(hipMemcpyAsync(K.dq.p, q.data(), (size_t)(total + K.pad), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(K.dkeys.p, 0xFF, tab * sizeof(uint64_t), h->stream));
    HIP_TRY(hipMemsetAsync(K.dcounts.p, 0, tab * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(K.dfilter.p, 0, ((size_t)1 << (KP.filter_bits - 5)) * sizeof(uint32_t), h->stream));
    HIP_TRY(mipgen_launch_kmer_insert(h->stream, K.dq.p, total, &KP, K.dkeys.p, K.dfilter.p));
    HIP_TRY(mipgen_launch_kmer_fold(h->stream, K.dfilter.p, KP.filter_bits, K.dfolded.p));
    HIP_TRY(hipStreamSynchronize(h->stream));                          // q dies with this scope
    DIAG_LAP("insert");
    double ms_total = 0.0;
    int64_t gbytes = 0;
    for (int c = 0; c < n_chrom; c++) {                                // one streaming pass per chromosome: 1 byte per genome base
        if (chrom_lens[c] <= 0) continue;
        HIP_TRY(hipMemcpyAsync(K.dg.p, chrom_seqs[c], (size_t)chrom_lens[c], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipEventRecord(K.e0, h->stream));
        HIP_TRY(mipgen_launch_kmer_count(h->stream, K.dg.p, chrom_lens[c], &KP, K.dkeys.p, K.dfilter.p, K.dfolded.p, K.dcounts.p, h->n_cu));
        HIP_TRY(hipEventRecord(K.e1, h->stream));
        HIP_TRY(hipEventSynchronize(K.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, K.e0, K.e1));
        ms_total += ms; gbytes += chrom_lens[c];
    }
    DIAG_LAP("genome passes");
    HIP_TRY(mipgen_launch_kmer_lookup(h->stream, K.dq.p, total, &KP, K.dkeys.p, K.dcounts.p, K.out.p));
    h->kmer_count_ms = ms_total; h->kmer_genome_bytes = gbytes;
    return MIPGEN_OK;
}

}  // namespace

int mipgen_accel_count_oligo_copies(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                    const char* const* region_seqs, const int32_t* region_lens, int32_t n_lengths, const int32_t* lengths,
                                    int32_t* const* copy_out)
{
    if (!h || n_chrom < 0 || n_regions < 0 || n_lengths < 1 || n_lengths > MIPGEN_MAX_OLIGO || !lengths || (n_chrom && (!chrom_seqs || !chrom_lens)) ||
        (n_regions && (!region_seqs || !region_lens || !copy_out)))
        return fail(MIPGEN_E_INVALID, "bad arguments");
    if (n_regions == 0) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    KmerRun K(&h->pool);
    if (int rc = kmer_count_run(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, n_lengths, lengths, K)) return rc;
    // back to the host one oligo length at a time through two pinned buffers: the copy of length s + 1 runs under the scatter of length s
    const int64_t total = K.total;
    PinnedPair<int32_t> pin;
    HIP_TRY(pin.alloc((size_t)total));
    HIP_TRY(hipMemcpyAsync(pin.buf[0], K.out.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(pin.done[0], h->stream));
    pin.busy[0] = true;
    for (int s = 0; s < n_lengths; s++) {
        const int b = s & 1;
        if (s + 1 < n_lengths) {
            HIP_TRY(hipMemcpyAsync(pin.buf[b ^ 1], K.out.p + (size_t)(s + 1) * (size_t)total, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipEventRecord(pin.done[b ^ 1], h->stream));
            pin.busy[b ^ 1] = true;
        }
        HIP_TRY(pin.wait(b));
        for (int r = 0; r < n_regions; r++) {
            const int len = region_lens[r];
            if (!copy_out[r]) continue;
            int32_t* dst = copy_out[r] + (size_t)s * (size_t)len;
            memcpy(dst, pin.buf[b] + K.roff[(size_t)r], (size_t)len * sizeof(int32_t));
            for (int i = std::max(0, len - lengths[s]); i < len; i++) dst[i] = 0;      // oligos the reference never writes (mipgen.cpp:829): absent key -> 0
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_count_oligo_copies_resident(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                             const char* const* region_seqs, const int32_t* region_lens, int64_t* n_big, const mipgen_big_copy** big)
{
    if (!h || n_chrom < 0 || n_regions < 0 || (n_chrom && (!chrom_seqs || !chrom_lens)) || (n_regions && (!region_seqs || !region_lens)))
        return fail(MIPGEN_E_INVALID, "bad arguments");
    h->resident_lens.clear();
    h->big_copies.clear();
    if (n_big) *n_big = 0;
    if (big) *big = nullptr;
    if (n_regions == 0) return MIPGEN_OK;
    const DevParams& D = h->hp;
    std::vector<int32_t> lengths;                                      // slot order = ascending oligo length (create_handle)
    for (int len = 0; len <= MIPGEN_MAX_OLIGO; len++) if (D.len_slot[len] >= 0) lengths.push_back(len);
    if ((int)lengths.size() != D.n_len_slots || lengths.empty()) return fail(MIPGEN_E_INVALID, "internal: oligo length slots");
    HIP_TRY(hipSetDevice(h->device));
    KmerRun K(&h->pool);
    if (int rc = kmer_count_run(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, (int32_t)lengths.size(), lengths.data(), K)) return rc;
    const int64_t copy_total = (K.total - n_regions) * (int64_t)D.n_len_slots;
    const unsigned int big_cap = (unsigned int)std::min<int64_t>(std::max<int64_t>(K.total / 4, (int64_t)1 << 16), (int64_t)1 << 28);
    DevBuf<int64_t> droff;
    DevBuf<mipgen_big_copy> dbig;
    DevBuf<unsigned int> dn;
    struct Free { DevBuf<int64_t>& a; DevBuf<mipgen_big_copy>& b; DevBuf<unsigned int>& c; ~Free() { a.release(); b.release(); c.release(); } } free_all{droff, dbig, dn};
    if (h->copy.reserve((size_t)std::max<int64_t>(copy_total, 1)) || droff.reserve(K.roff.size()) || dbig.reserve(big_cap) || dn.reserve(1)) return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemcpyAsync(droff.p, K.roff.data(), K.roff.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(dn.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(mipgen_launch_kmer_place(h->stream, K.out.p, K.total, &K.KP, droff.p, n_regions, h->copy.p, dbig.p, dn.p, big_cap));
    unsigned int nb = 0;
    HIP_TRY(hipMemcpyAsync(&nb, dn.p, sizeof nb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (nb > big_cap) return fail(MIPGEN_E_NOMEM, "%u oligos with 65535 or more copies (list capacity %u): use mipgen_accel_count_oligo_copies", nb, big_cap);
    h->big_copies.resize(nb);
    if (nb) HIP_TRY(hipMemcpy(h->big_copies.data(), dbig.p, (size_t)nb * sizeof(mipgen_big_copy), hipMemcpyDeviceToHost));
    std::sort(h->big_copies.begin(), h->big_copies.end(), [](const mipgen_big_copy& x, const mipgen_big_copy& y) {
        return x.region != y.region ? x.region < y.region : (x.length != y.length ? x.length < y.length : x.start < y.start); });
    h->resident_lens.assign(region_lens, region_lens + n_regions);
    if (n_big) *n_big = (int64_t)nb;
    if (big) *big = h->big_copies.data();
    return MIPGEN_OK;
}


// ---- section 8f-3, second half: uniqueness of whole capture windows (kernels_window.hip) ---------------------------------------------------
extern "C" hipError_t mipgen_launch_window_spans(hipStream_t st, const char* q, const int64_t* roff, int n_regions, uint16_t* dist_bad, uint16_t* dist_end, uint16_t* dist_start);
extern "C" hipError_t mipgen_launch_seed_index(hipStream_t st, const char* q, int64_t total, int k, const uint64_t* keys, uint64_t cap_mask, unsigned int* rmult,
                                               unsigned int* rstart, unsigned int* rfill, uint32_t* rlist, unsigned int* alloc, int phase);
extern "C" hipError_t mipgen_launch_window_verify(hipStream_t st, const char* G, int64_t glen, const char* q, int64_t total, const int32_t* sizes, int n_sizes, int k,
                                                  const uint64_t* keys, uint64_t cap_mask, const unsigned int* counts, const uint32_t* filter, int filter_bits,
                                                  const unsigned int* rmult, const unsigned int* rstart, const uint32_t* rlist, const uint16_t* dist_start, unsigned int* ctr);
extern "C" hipError_t mipgen_launch_window_flags(hipStream_t st, const char* q, int64_t total, const int32_t* sizes, int n_sizes, int k, const uint64_t* keys,
                                                 uint64_t cap_mask, const unsigned int* counts, const uint16_t* dist_bad, const uint16_t* dist_end, const unsigned int* ctr,
                                                 uint8_t* unmap, const int64_t* roff, int n_regions, const int32_t* bounds, uint8_t* any);

// bounds == nullptr: the full flag image goes to unmap_out (mipgen_accel_window_uniqueness).  bounds != nullptr: the flags are restricted to the
// window starts the reference looks up on the device, the image stays in the handle (h->win_img) and any_out gets one byte per region
// (mipgen_accel_window_uniqueness_begin)
static int window_uniqueness_impl(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                  const char* const* region_seqs, const int32_t* region_lens, int32_t n_sizes, const int32_t* sizes, int32_t seed_len,
                                  uint8_t* const* unmap_out, const mipgen_window_bounds* bounds, uint8_t* any_out)
{
    if (!h || n_chrom < 0 || n_regions < 0 || n_sizes < 1 || n_sizes > 64 || !sizes || (n_chrom && (!chrom_seqs || !chrom_lens)) ||
        (n_regions && (!region_seqs || !region_lens || (!bounds && !unmap_out) || (bounds && !any_out))))
        return fail(MIPGEN_E_INVALID, "bad arguments");
    if (seed_len < 12 || seed_len > 31) return fail(MIPGEN_E_INVALID, "seed length %d: must be in [12, 31] (exact 2-bit keys)", seed_len);
    int max_size = 0;
    for (int i = 0; i < n_sizes; i++) {
        if (sizes[i] < 2 * seed_len || sizes[i] > 60000) return fail(MIPGEN_E_INVALID, "capture size %d: must be in [2 x seed length, 60000] (two disjoint seeds per window)", sizes[i]);
        max_size = std::max(max_size, sizes[i]);
    }
    if (n_regions == 0) return MIPGEN_OK;
    HIP_TRY(hipSetDevice(h->device));
    KmerRun K(&h->pool);
    K.pad = (int64_t)max_size + 1;
    const int32_t lengths[1] = {seed_len};
    if (int rc = kmer_count_run(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, 1, lengths, K)) return rc;    // seeds + their genome loci
    const int64_t total = K.total;
    if (total >= ((int64_t)1 << 31)) return fail(MIPGEN_E_INVALID, "window uniqueness: more than 2^31 region bases in one call");
    const uint64_t cap = K.KP.cap_mask + 1;
    DevBuf<unsigned int> rmult, rstart, rfill, alloc, ctr;
    DevBuf<uint32_t> rlist;
    DevBuf<uint16_t> dbad, dend, dstart;
    DevBuf<int64_t> droff;
    DevBuf<uint8_t> dun, dany;
    DevBuf<int32_t> dbounds;
    h->win_img.release(); h->win_roff.clear(); h->win_lens.clear(); h->win_sizes = 0; h->win_total = 0;
    struct Free { std::vector<std::function<void()>> f; ~Free() { for (auto& g : f) g(); } } fr;     // DevBuf has no destructor: release on every exit
    fr.f = {[&] { rmult.release(); }, [&] { rstart.release(); }, [&] { rfill.release(); }, [&] { alloc.release(); }, [&] { ctr.release(); }, [&] { rlist.release(); },
            [&] { dbad.release(); }, [&] { dend.release(); }, [&] { dstart.release(); }, [&] { droff.release(); }, [&] { dun.release(); }, [&] { dany.release(); },
            [&] { dbounds.release(); }};
    if (bounds) {
        if (dany.reserve((size_t)n_regions) || dbounds.reserve(4 * (size_t)n_regions)) return MIPGEN_E_NOMEM;
        static_assert(sizeof(mipgen_window_bounds) == 16, "four int32");
        HIP_TRY(hipMemcpyAsync(dbounds.p, bounds, (size_t)n_regions * sizeof(mipgen_window_bounds), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(dany.p, 0, (size_t)n_regions, h->stream));
    }
    if (rmult.reserve(cap) || rstart.reserve(cap) || rfill.reserve(cap) || alloc.reserve(1) || ctr.reserve((size_t)n_sizes * (size_t)total) || rlist.reserve((size_t)total) ||
        dbad.reserve((size_t)total) || dend.reserve((size_t)total) || dstart.reserve((size_t)total) || droff.reserve(K.roff.size()) ||
        dun.reserve((size_t)n_sizes * (size_t)total))
        return MIPGEN_E_NOMEM;
    HIP_TRY(hipMemsetAsync(rmult.p, 0, cap * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(rfill.p, 0, cap * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(alloc.p, 0, sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemsetAsync(ctr.p, 0, (size_t)n_sizes * (size_t)total * sizeof(unsigned int), h->stream));
    HIP_TRY(hipMemcpyAsync(droff.p, K.roff.data(), K.roff.size() * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(mipgen_launch_window_spans(h->stream, K.dq.p, droff.p, n_regions, dbad.p, dend.p, dstart.p));
    for (int phase = 0; phase < 3; phase++)
        HIP_TRY(mipgen_launch_seed_index(h->stream, K.dq.p, total, seed_len, K.dkeys.p, K.KP.cap_mask, rmult.p, rstart.p, rfill.p, rlist.p, alloc.p, phase));
    for (int c = 0; c < n_chrom; c++) {                                // second genome pass: extend the loci of the repeated seeds
        if (chrom_lens[c] <= 0) continue;
        HIP_TRY(hipMemcpyAsync(K.dg.p, chrom_seqs[c], (size_t)chrom_lens[c], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(mipgen_launch_window_verify(h->stream, K.dg.p, chrom_lens[c], K.dq.p, total, sizes, n_sizes, seed_len, K.dkeys.p, K.KP.cap_mask, K.dcounts.p,
                                            K.dfilter.p, K.KP.filter_bits, rmult.p, rstart.p, rlist.p, dstart.p, ctr.p));
        HIP_TRY(hipStreamSynchronize(h->stream));                      // the next chromosome overwrites the genome buffer
    }
    HIP_TRY(mipgen_launch_window_flags(h->stream, K.dq.p, total, sizes, n_sizes, seed_len, K.dkeys.p, K.KP.cap_mask, K.dcounts.p, dbad.p, dend.p, ctr.p, dun.p,
                                       droff.p, n_regions, bounds ? dbounds.p : nullptr, bounds ? dany.p : nullptr));
    if (bounds) {
        // the image stays on the device: the caller fetches the few regions that have a flagged start (mipgen_accel_window_flags_region)
        HIP_TRY(hipMemcpyAsync(any_out, dany.p, (size_t)n_regions, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->win_img = dun; dun = DevBuf<uint8_t>();                     // ownership moves to the handle (released by _end / the next call / destroy)
        h->win_roff = K.roff; h->win_lens.assign(region_lens, region_lens + n_regions); h->win_sizes = n_sizes; h->win_total = total;
        return MIPGEN_OK;
    }
    std::vector<uint8_t> img((size_t)n_sizes * (size_t)total);
    HIP_TRY(hipMemcpyAsync(img.data(), dun.p, img.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int r = 0; r < n_regions; r++) {
        if (!unmap_out[r]) continue;
        const int len = region_lens[r];
        for (int c = 0; c < n_sizes; c++) memcpy(unmap_out[r] + (size_t)c * (size_t)len, &img[(size_t)c * (size_t)total + (size_t)K.roff[(size_t)r]], (size_t)len);
    }
    return MIPGEN_OK;
}

int mipgen_accel_window_uniqueness(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                   const char* const* region_seqs, const int32_t* region_lens, int32_t n_sizes, const int32_t* sizes, int32_t seed_len,
                                   uint8_t* const* unmap_out)
{
    return window_uniqueness_impl(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, n_sizes, sizes, seed_len, unmap_out, nullptr, nullptr);
}

int mipgen_accel_window_uniqueness_begin(mipgen_accel* h, int32_t n_chrom, const char* const* chrom_seqs, const int64_t* chrom_lens, int32_t n_regions,
                                         const char* const* region_seqs, const int32_t* region_lens, const mipgen_window_bounds* bounds,
                                         int32_t n_sizes, const int32_t* sizes, int32_t seed_len, uint8_t* any_out)
{
    if (!bounds || !any_out) return fail(MIPGEN_E_INVALID, "bad arguments");
    return window_uniqueness_impl(h, n_chrom, chrom_seqs, chrom_lens, n_regions, region_seqs, region_lens, n_sizes, sizes, seed_len, nullptr, bounds, any_out);
}

int mipgen_accel_window_flags_region(mipgen_accel* h, int32_t region, uint8_t* out)
{
    if (!h || !out) return fail(MIPGEN_E_INVALID, "bad arguments");
    if (!h->win_img.p || region < 0 || (size_t)region >= h->win_lens.size()) return fail(MIPGEN_E_STATE, "no window-uniqueness image in the handle for region %d", region);
    HIP_TRY(hipSetDevice(h->device));
    const size_t len = (size_t)h->win_lens[(size_t)region];
    if (len == 0) return MIPGEN_OK;
    HIP_TRY(hipMemcpy2DAsync(out, len, h->win_img.p + h->win_roff[(size_t)region], (size_t)h->win_total, len, (size_t)h->win_sizes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MIPGEN_OK;
}

int mipgen_accel_window_uniqueness_end(mipgen_accel* h)
{
    if (!h) return fail(MIPGEN_E_INVALID, "null handle");
    h->win_img.release(); h->win_roff.clear(); h->win_lens.clear(); h->win_sizes = 0; h->win_total = 0;
    return MIPGEN_OK;
}

}  // extern "C"
