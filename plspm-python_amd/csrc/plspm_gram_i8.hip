// plspm_gram_i8.hip -- host side, part 3: the digit planes of the resident data and the int8 MFMA Gram of bootstrap batches (resample counts
// + one exact integer product; kernels_gram_i8.h, kernels_gram_i8p.h), the cut of a batch into tile rows, plspm_bootstrap_prepare.
#include "host_internal.h"

#include "philox.h"
#include "wave_ops.h"
#include "kernels_gram_i8.h"
#include "kernels_gram_i8p.h"

// ------------------------------------------------------------------------------------------------ int8 digit-plane Gram (kernels_gram_i8.h)
static constexpr size_t kZsBudget = (size_t)24 << 30;       // digit planes of one data set: at most 24 GiB of the 288 GiB
// Which Gram a bootstrap call of B replicates takes: 1 = fp64 MFMA on the (row,count) lists, 2 = int8 digit planes.
// Can a non-metric bootstrap with on-device draws take its stop-rule passes' row multiplicities from the int8 counts of the digit-plane Gram
// (plspm_detail_bootstrap counts8_plan)?
static int cu_count_of(plspm_model* m) {
    if (!m->cu_count) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, m->device) == hipSuccess) m->cu_count = n;
    }
    return m->cu_count;
}

bool nm_counts8_possible(const plspm_model* m) {
    return m->nonmetric && m->tune.nm_counts8 != 0 && m->tune.resample_aux == 0 && !m->aux && m->tune.i8_shape == 16 && m->nmx_K == 0 &&
           nm_dense_lds(m, nullptr, nullptr) != 0 && (!m->stage2 || nm_dense_lds(m->stage2, nullptr, nullptr) != 0);
}
int choose_gram_path(const plspm_model* m, int64_t B) {
    if (m->tune.gram_path == 1) return 1;
    // every model's replicates start from the moment matrix of the uploaded columns (metric, mean-imputed, non-metric, categorical
    // indicator columns, incomplete rows zeroed, first stage of a HOC pair).  The LDS histogram bounds N; int32 accumulators need
    // 128 N < 2^31.
    // (int32 accumulators: |sum_i c_bi d_is| <= 128 sum_i c_bi = 128 N < 2^31, i.e. N < 2^24; the resample counts come from an LDS
    // histogram of 65,536 rows per workgroup, larger data sets take several windows per replicate)
    if (m->stage1 || m->N >= (1 << 24) || m->N < 2) return 1;
    // non-metric models beyond one 16-bit histogram window: the int8 route when their stop-rule passes can read the Gram's int8 counts
    // (on-device draws, and -- round 4 -- explicit index lists too: windowed 16-bit histograms, resample_i8_kernel; a chunk that carries a
    // multiplicity above 127 falls back to row lists from the global histogram, the fp64 Gram and the gathering pass, as for metric models);
    // else the fp64 route
    if (m->nonmetric && m->N > 65535 && !nm_counts8_possible(m)) return 1;
    const size_t zs_bytes = (size_t)(i8_kblocks(m->N) + I8_SLACK_KB) * (size_t)(((i8_pairs(m) + 31) / 32) * 2 * (m->tune.i8_slices ? m->tune.i8_slices : 7)) * 1024;
    if (zs_bytes > kZsBudget) return 1;
    if (m->tune.gram_path == 2) return 2;
    return B >= m->tune.i8_min_batch ? 2 : 1;
}

// Digit planes + pair tables of the resident data (once per upload / digit count).
// How many digit planes ("i8_slices" 0 = automatic).  S planes represent every product with an absolute error of at most
// 2^-(8S-1) max|z| of its column (zs_scale_kernel), so a replicate's sum is off by at most N 2^-(8S-1) max|z| (the multiplicities add up
// to N) -- the integer sum itself is exact.  A sequential fp64 accumulation of the same N terms carries the a-priori bound gamma_N sum|z|
// ~ N 2^-53 sum|z|.  S planes are therefore within the error bound of fp64 arithmetic on the same data whenever
//       sum_i |z_i|  >=  2^(54 - 8S) max_i |z_i|        in every pair column:      S = 7: always,   S = 6: sum >= 64 max.
// Automatic = 6 planes if every column clears that bar with a factor 4 to spare (sum >= 256 max; a resample re-weights the terms:
// sum_i c_i |z_i| scatters by a few percent around sum_i |z_i|), else 7.  Data sets of a few hundred rows stay at 7; 10,000 rows of
// anything bell-shaped reach several hundred.  Measured against 80-bit sums on the 10k x 60 benchmark data (tests/test_gpu_gram_i8.py,
// error relative to sqrt(M_pp M_qq)): seven planes 1e-16 (correctly rounded), six planes 3e-15, the blocked fp64 MFMA accumulation
// 1.6e-15 -- all nine orders below the 1e-6 the records are held to.
static int choose_slices(plspm_model* m, const unsigned long long* h, long npair, int* S_out) {      // h: [max bits | fixed-point sums | OR of the scaled integers], on the host
    double worst = 1e300;
    unsigned long long any = 0ull;
    for (long j = 0; j < npair; ++j) any |= h[2 * npair + j];
    for (long j = 0; j < npair; ++j) {
        const unsigned long long mb = h[j];
        if (mb == 0) continue;                                            // an all-zero column has nothing to round
        const int ef = (int)((mb >> 52) & 0x7ffull);
        if (ef < 64 || ef >= 0x7ff) { worst = 0.0; break; }               // tiny or non-finite maximum: not evaluated, full plane count
        double zmax;
        memcpy(&zmax, &mb, sizeof(double));
        const double sum = std::ldexp((double)h[npair + j], (ef - 1022) - 40);
        worst = std::min(worst, sum / zmax);
    }
    m->zs_ratio = worst;
    // (round 5: EIGHT planes when some column's sum barely exceeds its largest product -- sum < 2 max: one gross outlier row carries the column.  A
    //  replicate that does not draw that row sums products that are tiny against the column maximum the planes are scaled to; seven planes then
    //  leave ~1e-14 of ITS moment (measured: tests/test_gpu_gram_i8.py, a 900-sigma cell), eight stay below fp64's own rounding.  Same a-priori
    //  argument as for six / seven with the dominant row taken out of the sum: sum - max >= 2^(56 - 8S) max.)
    int S = worst >= 257.0 ? 6 : (worst >= 2.0 ? 7 : 8);
    if (m->tune.i8_min_slices > S) S = m->tune.i8_min_slices;              // "i8_min_slices": the automatic choice, but never fewer (Plspm(precision="strict"): 7)
    // planes that would be identically zero in the seven-plane decomposition carry nothing: dropping them changes no sum
    int zero_planes = 0;
    if (worst > 0.0) {
        zero_planes = 6;
        if (any) { int tz = 0; while (!((any >> tz) & 1ull)) ++tz; zero_planes = std::min(6, tz / 8); }
    }
    if (zero_planes > 0) S = std::min(S, 7 - zero_planes);                   // (data on a coarse binary grid are EXACT on fewer planes, whatever the rule above asked for)
    if (m->tune.i8_shape == 32) S = std::max(S, 5);                            // (the 32x32x32 layout is instantiated for 5 .. 8 planes)
    *S_out = S;
    return 0;
}

// Phase 1 of the digit planes (enqueue only -- plspm_bootstrap_prepare): pair tables, column maxima of the pair products and, for the
// automatic plane count, the two column statistics, copied to a pinned block behind an event.  Nothing here waits for the device: the
// plane count is read in phase 2 (prepare_zs), by which time the fit that was enqueued behind this has long synchronised the stream.
int prepare_zs_stats(plspm_model* m) {
    if (m->zs_valid || m->zs_stats_ready) return 0;
    const int C = m->Pg + 1;
    const long npair = i8_pairs(m);
    std::vector<int> tab(7 * (size_t)npair);
    int* hp = tab.data(); int* hq = hp + npair; int* hd = hq + 2 * npair;       // [p | q | k (device) | packed slot | dense slot | mirrored dense slot | uint16 count-matrix slot]
    int* hd1 = hd + npair; int* hd2 = hd1 + npair; int* hd16 = hd2 + npair;
    const int PSd = cov_ld(m->Pg), ld16 = (C + 7) & ~7;                          // (ld16: the pitch of run_nonmetric's uint16 count matrices)
    long j = 0;
    for (int p = 0; p < C; ++p)
        for (int q = p; q < C; ++q, ++j) {
            hp[j] = p; hq[j] = q; hd[j] = (int)packed_index(m->T, p, q);
            hd1[j] = p * PSd + q; hd2[j] = (p == q) ? -1 : q * PSd + p;
            hd16[j] = p * ld16 + q;
        }
    int rc;
    if ((rc = ensure(m, m->pair_tab, tab.size() * sizeof(int)))) return rc;
    if ((rc = ensure(m, m->pair_scale, (size_t)npair * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->zs_stat, 3 * (size_t)npair * sizeof(unsigned long long)))) return rc;
    if ((rc = plspm_detail_h2d(m, m->pair_tab.p, tab.data(), tab.size() * sizeof(int)))) return rc;
    int* d_p = (int*)m->pair_tab.p; int* d_q = d_p + npair;
    ProfScope ps(m, PLSPM_K_PACK);
    unsigned long long* d_max = (unsigned long long*)m->zs_stat.p;
    HIPCHK(m, hipMemsetAsync(d_max, 0, 3 * (size_t)npair * sizeof(unsigned long long), m->stream));
    const int RB = (int)std::max<size_t>(1, std::min<size_t>(64, (kMaxLds - 1024) / ((size_t)(C | 1) * sizeof(double))));
    const size_t lds = (size_t)RB * (C | 1) * sizeof(double);
    if ((rc = allow_lds(m, (const void*)zs_max_kernel, lds))) return rc;
    hipLaunchKernelGGL(zs_max_kernel, dim3((unsigned)((m->N + RB - 1) / RB)), dim3(256), lds, m->stream, (const double*)m->d_Xa, (long)m->N, m->PA, C, d_p, d_q, (int)npair, RB, d_max);
    if (m->tune.i8_slices == 0) {
        if ((rc = allow_lds(m, (const void*)zs_abssum_kernel, lds))) return rc;
        hipLaunchKernelGGL(zs_abssum_kernel, dim3((unsigned)((m->N + RB - 1) / RB)), dim3(256), lds, m->stream, (const double*)m->d_Xa, (long)m->N, m->PA, C, d_p, d_q, (int)npair, RB,
                           (const unsigned long long*)d_max, d_max + npair, d_max + 2 * npair);
        HIPCHK(m, hipGetLastError());
        const size_t bytes = 3 * (size_t)npair * sizeof(unsigned long long);
        if (m->h_zstat_cap < bytes) {
            if (m->h_zstat) plspm_hfree(m->h_zstat);
            m->h_zstat = nullptr; m->h_zstat_cap = 0;
            HIPCHK(m, plspm_hmalloc(&m->h_zstat, bytes));
            m->h_zstat_cap = bytes;
        }
        if (!m->ev_zstat) HIPCHK(m, hipEventCreateWithFlags(&m->ev_zstat, hipEventDisableTiming));
        HIPCHK(m, hipMemcpyAsync(m->h_zstat, d_max, bytes, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipEventRecord(m->ev_zstat, m->stream));
    }
    HIPCHK(m, hipGetLastError());
    m->zs_stats_ready = true; m->zs_stats_S = m->tune.i8_slices;
    return 0;
}

int prepare_zs(plspm_model* m) {
    if (m->zs_valid) return 0;
    int rc;
    if (m->zs_stats_ready && m->zs_stats_S != m->tune.i8_slices) m->zs_stats_ready = false;      // the option changed in between
    if ((rc = prepare_zs_stats(m))) return rc;
    int S = m->tune.i8_slices;
    const long npair = i8_pairs(m);
    const int npg = (int)((npair + 31) / 32) * 2;             // pair groups of 16, padded to whole workgroup tiles (two groups)
    const int KB = i8_kblocks(m->N);
    int* d_p = (int*)m->pair_tab.p; int* d_q = d_p + npair; int* d_k = d_q + npair;
    unsigned long long* d_max = (unsigned long long*)m->zs_stat.p;
    ProfScope ps(m, PLSPM_K_PACK);
    if (S == 0) {
        HIPCHK(m, hipEventSynchronize(m->ev_zstat));          // (long done when a fit ran behind plspm_bootstrap_prepare)
        if ((rc = choose_slices(m, (const unsigned long long*)m->h_zstat, npair, &S))) return rc;
    }
    hipLaunchKernelGGL(zs_scale_kernel, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, m->stream, d_max, (int)npair, S, d_k, (double*)m->pair_scale.p);
    // one plane (0/1 data): the product runs through the seven-plane main loop with the planes of a wave standing for seven consecutive
    // pair groups (gram_i8_kernel<.., IND>): the buffer is padded to whole tiles of 2 x 7 groups
    const bool ind = S == 1 && m->tune.i8_shape == 16 && m->tune.i8_ind != 0;
    const int npg_built = ind ? ((npg + 13) / 14) * 14 : npg;
    const int NT = npg_built * S;
    if ((rc = ensure(m, m->zs, (size_t)(KB + I8_SLACK_KB) * (size_t)NT * 1024))) return rc;      // (sized once the plane count is known: 0/1 data take one plane)
    const dim3 grid((unsigned)KB, (unsigned)((npg_built + 3) / 4));
#define ZSB(SS) hipLaunchKernelGGL((zs_build_kernel<SS>), grid, dim3(256), 0, m->stream, (const double*)m->d_Xa, (long)m->N, m->PA, d_p, d_q, d_k, (int)npair, npg_built, NT, m->tune.i8_shape, (uint4*)m->zs.p)
    switch (S) { case 1: ZSB(1); break; case 2: ZSB(2); break; case 3: ZSB(3); break; case 4: ZSB(4); break; case 5: ZSB(5); break; case 6: ZSB(6); break; case 7: ZSB(7); break; default: ZSB(8); break; }
#undef ZSB
    HIPCHK(m, hipGetLastError());
    m->zs_S = S; m->zs_KB = KB; m->zs_NT = NT; m->zs_npair = (int)npair; m->zs_npg = npg_built; m->zs_ind = ind;
    m->zs_valid = true; m->zs_stats_ready = false;
    return 0;
}

// Tile rows of the six-plane int8 Gram for `ct` count tiles (16 replicates each) x `ntx` pair tiles on `cus` CUs in 8 XCDs: `tall` rows of
// 20 count tiles and -- `mix` -- `shrt` rows of 16 in one launch of gram_i8_kernel<6, 4, ., 16, 20>, or (return false) the 256-replicate
// kernel.  Cost model = what the device does: an XCD's workgroups go in order to the CU that is free first (one workgroup per CU), so the
// makespan of a cut is that of list scheduling its tall tiles first, then its short ones, per XCD.  Costs in count-tile rows: 20 per
// tall tile, 16.6 per short one (the same DMA ring for 4/5 of the MFMAs), 16.35 per tile of the 256-replicate kernel (measured on 960 tiles
// of each kind, tools/i8_mix_calib.py: 0.391 / 0.3245 / 0.3195 ms).  Deterministic in (ct, ntx, cus): every rank of a job cuts alike -- and the sums are exact
// integers, so the cut never shows in a result.
// (profiles/r04_i8_mix_calib.jsonl: 960 tiles of each height, 0.3628 / 0.2990 ms with six planes, 0.3435 / 0.2700 with seven)
static constexpr double kI8pTall6 = 20.0, kI8pShort6 = 16.5, kI8pTall7 = 16.0, kI8pShort7 = 12.6;
// Tile heights and their costs: `rt_tall` / `rt_short` count tiles per row, `ca` / `cb` what a tile of each costs (any common unit).  The
// round-3 kernel: 20 / 16 at 20 / 16.6; gram_i8p_kernel: i8p_costs() below.
static bool i8_mix_plan(long ct, long ntx, int cus, bool mix, int* tall, int* shrt, int rt_tall = 20, int rt_short = 16, double ca = 20.0, double cb = 16.6, bool vs_rt16 = true) {
    const int per_xcd = std::max(1, cus / 8);
    // (tiles of one kind are interchangeable: after the tall ones the CUs of an XCD sit on at most two load levels, and the short ones raise
    //  the lowest level a whole group of CUs at a time -- a handful of steps per XCD instead of one per tile)
    auto xcd_span = [&](long na, long nb2, double ca, double cb) {
        const long c = per_xcd, q = na / c, r = na % c;
        double lv[3] = {q * ca, (q + 1) * ca, 0.0};
        long cnt[3] = {c - r, r, 0};
        int n = r ? 2 : 1;
        long left = nb2;
        while (left > 0) {
            int lo = 0;
            for (int k = 1; k < n; ++k) if (lv[k] < lv[lo]) lo = k;
            if (left >= cnt[lo]) { left -= cnt[lo]; lv[lo] += cb; }
            else { lv[n] = lv[lo] + cb; cnt[n] = left; cnt[lo] -= left; left = 0; ++n; }
            for (int k = 0; k < n; ++k)                      // merge equal levels (keeps n <= 2 before the last step)
                for (int k2 = k + 1; k2 < n; ++k2)
                    if (lv[k2] == lv[k]) { cnt[k] += cnt[k2]; lv[k2] = lv[n - 1]; cnt[k2] = cnt[n - 1]; --n; --k2; }
        }
        double worst = 0.0;
        for (int k = 0; k < n; ++k) if (cnt[k] > 0) worst = std::max(worst, lv[k]);
        return worst;
    };
    auto makespan = [&](long a, long b, double ca, double cb) {
        double worst = 0.0;
        const long ta = a * ntx, tb = b * ntx, pa = (ta + 7) / 8, pb = (tb + 7) / 8;
        long seen_a = -1, seen_b = -1;
        for (int x = 0; x < 8; ++x) {
            const long na = std::max(0L, std::min(pa, ta - x * pa)), nb2 = std::max(0L, std::min(pb, tb - x * pb));
            if (na == seen_a && nb2 == seen_b) continue;
            seen_a = na; seen_b = nb2;
            worst = std::max(worst, xcd_span(na, nb2, ca, cb));
        }
        return worst;
    };
    const long rows16 = (ct + 15) / 16, rows_s = (ct + rt_short - 1) / rt_short;
    const double ref16 = vs_rt16 ? makespan(rows16, 0, 16.35, 0.0) : 1e300;
    double best = 1e300;
    long ba = 0, bb = 0;
    for (long b = 0; b <= (mix ? std::min(rows_s, 48L) : 0L); ++b) {
        const long a = std::max(0L, (ct - (long)rt_short * b + rt_tall - 1) / rt_tall);
        if (a == 0 && b * rt_short < ct) continue;
        const double t = makespan(a, b, ca, cb);
        if (t < best - 1e-9) { best = t; ba = a; bb = b; }
        if (a == 0) break;
    }
    *tall = (int)ba; *shrt = (int)bb;
    return best <= ref16;
}

// Resample nb replicates into dense int8 counts and multiply with the digit planes: the nb moment matrices land at `out`.
// Explicit indices can carry a multiplicity above 127 (Philox draws of N >= 128 rows cannot, P < 1e-200): the host looks at the
// flag before the product and reports *fallback so that the caller takes the fp64 Gram for this chunk.
// out16 (round 5, may be null): where an all-indicator data set's products -- co-occurrence counts -- may leave as uint16 [nb][C][ld16], upper triangle, instead
// of fp64 slots at `out`; *wrote16 tells whether this call did (the one-plane launch on 0/1 data; any other launch writes `out` as ever).
int run_gram_i8(plspm_model* m, int64_t nb, uint64_t seed, int64_t rep0, const int32_t* d_idx, double* out, bool dense, bool* fallback,
                const void** counts, int* counts_MT, unsigned short* out16, bool* wrote16) {
    *fallback = false;
    if (counts) *counts = nullptr;
    if (wrote16) *wrote16 = false;
    const int S = m->zs_S, KB = m->zs_KB, NT = m->zs_NT;
    // workgroup tile of the product: 16 RT replicates x 32 pairs; narrow tiles (RT 12 / 8) only in the plain four-wave 16x16x64 launch
    // Six planes leave registers for a taller workgroup tile: 320 replicates x 32 pairs (`RT` 20: 30 accumulator tiles per wave with eight
    // waves) moves 9 % fewer LDS-DMA bytes and reads 12 % fewer fragments per MFMA than 256 x 32 -- 2.2 % on the step when the tile grid
    // fills the machine equally well (5,000 replicates: 960 tiles = 3.75 rounds against 1,200 = 4.69, both five tile-units per CU).  "i8_rt"
    // 0 (default) takes it when its rounds cost no more than those of the 256-replicate tile; the sums are exact either way.
#ifdef PLSPM_I8_EXPERIMENTS
    const bool var20 = m->tune.i8_variant < 0 || m->tune.i8_rt == 20;      // experiments build: the schedule variants exist for the 320-replicate tile too
#else
    const bool var20 = m->tune.i8_variant < 0;
#endif
    // private count fragments (kernels_gram_i8p.h, "i8_priv"): four waves, the counts straight into registers, only the digit blocks through LDS;
    // six planes: tile rows of 320 (tall) / 256 (short) replicates, seven planes: 256 / 192
#ifdef PLSPM_I8_EXPERIMENTS
    const bool priv_var = true;                                             // schedule variants / ablation probes of the kernel (tools/i8p_bench.py)
#else
    const bool priv_var = m->tune.i8_variant < 0;
#endif
    const bool priv = m->tune.i8_priv != 0 && m->tune.i8_shape == 16 && m->tune.i8_sched == 0 && priv_var && (S == 6 || S == 7) && !m->zs_ind &&
                      (m->tune.i8_rt == 0 || m->tune.i8_rt == (S == 6 ? 20 : 16));      // (an explicit other tile height names a round-3 kernel)
    bool wide20 = !priv && m->tune.i8_shape == 16 && m->tune.i8_sched == 0 && var20 && S == 6 && !m->zs_ind && (m->tune.i8_rt == 20 || m->tune.i8_rt == 0);
    const int RTtall = priv ? (S == 6 ? 20 : 16) : 20, RTshort = RTtall - 4;
    // "i8_rt" 0: the cut of the replicates into tile rows is planned (i8_mix_plan below): rows of 320 and -- eight-wave kernel -- rows of 256
    // in ONE launch, so that the last round of the machine is as full as the others (5,000 replicates x 60 pair tiles: 11 + 6 rows = 1,020
    // tiles, every CU three tall + one short = 76 count-tile rows, against 16 rows of 320 = 960 tiles, 80 on three CUs of four)
    int nty_tall = 0, nty_short = 0;
    if ((wide20 || priv) && m->tune.i8_rt == 0 && m->tune.i8_short < 0) {
        if (!cu_count_of(m)) return fail(m, PLSPM_E_STATE, "hipDeviceGetAttribute(multiprocessor count) failed");
        // (the plan of the last shape is kept: a bootstrap calls with the same B again and again, and the search costs of the order of a millisecond)
        // ("i8_cus": the cut for fewer CUs than the device has -- a launch that shares the chip with a collective's kernels)
        const long key[4] = {(long)((nb + 15) / 16), (long)(m->zs_npg / 2), (long)std::max(8, m->tune.i8_cus > 0 ? std::min(m->tune.i8_cus, m->cu_count) : m->cu_count),
                             priv ? 2L + S : (long)(m->tune.i8_waves == 8 && var20 && m->tune.i8_dma != 2)};
        if (!(m->mix_valid && std::equal(key, key + 4, m->mix_key))) {
            // (tile costs of gram_i8p_kernel, tools/i8_mix_calib.py on 960 tiles of each height: six planes 320 / 256 replicates, seven planes 256 / 192)
            if (priv) m->mix_wide = i8_mix_plan(key[0], key[1], (int)key[2], true, &m->mix_tall, &m->mix_short, RTtall, RTshort, S == 6 ? kI8pTall6 : kI8pTall7, S == 6 ? kI8pShort6 : kI8pShort7, false);
            else m->mix_wide = i8_mix_plan(key[0], key[1], (int)key[2], key[3] != 0, &m->mix_tall, &m->mix_short);
            std::copy(key, key + 4, m->mix_key); m->mix_valid = true;
        }
        if (!priv) wide20 = m->mix_wide;
        nty_tall = m->mix_tall; nty_short = m->mix_short;
    } else if (wide20 || priv) {
        // tall rows only, or -- "i8_short_rows" n >= 0 (test seam) -- n short rows behind as many tall ones as it takes
        const long ct = (nb + 15) / 16;
        if (m->tune.i8_short > 0 && (priv || (m->tune.i8_waves == 8 && var20 && m->tune.i8_dma != 2))) nty_short = (int)std::min<long>(m->tune.i8_short, (ct + RTshort - 1) / RTshort);
        nty_tall = (int)std::max(0L, (ct - (long)RTshort * nty_short + RTtall - 1) / RTtall);
    }
    const bool rows2 = wide20 || priv;                 // launches whose grid holds tile rows of two heights
    const bool narrow = wide20 || (!priv && m->tune.i8_rt == 8 && m->tune.i8_shape == 16 && m->tune.i8_waves == 4 && m->tune.i8_sched == 0 && m->tune.i8_variant < 0 && S == 7);
    const int RTg = rows2 ? RTtall : narrow ? 8 : 16;
    const bool ind = m->zs_ind && m->tune.i8_sched == 0 && m->tune.i8_variant < 0;
    const int nty = rows2 ? nty_tall + nty_short : (int)((nb + 16 * RTg - 1) / (16 * RTg)), MT = rows2 ? nty_tall * RTtall + nty_short * RTshort : nty * RTg, ntx = ind ? m->zs_npg / 14 : m->zs_npg / 2;
    // resample counts from an LDS histogram per (replicate, window of rows): 65,536 rows of 16-bit counters, or -- Philox draws of a data
    // set that would need more than one such window -- 131,072 rows of 8-bit counters (kernels_gram_i8.h resample_i8_kernel)
    // (round 5: 4-bit counters with an exact overflow test and an 8-bit second try, resample_i8_nib_kernel -- half the LDS again, three workgroups per CU at N = 100,000;
    //  option "i8_nibbles" 0: the 8-bit histogram, 2: every replicate through the second try -- tests)
    const bool hist_byte = KB > I8_HIST_KB && !d_idx;
    const bool hist_nib = hist_byte && m->tune.i8_nibbles != 0 && m->tune.i8_shape == 16;
    const int hist_kb = hist_nib ? I8_HIST_KB_NIB : (hist_byte ? I8_HIST_KB_BYTES : I8_HIST_KB);
    const size_t hist_bytes = (size_t)std::min(KB, hist_kb) * (hist_nib ? 8 : (hist_byte ? 16 : 32)) * sizeof(unsigned);
    const unsigned hist_windows = (unsigned)((KB + hist_kb - 1) / hist_kb);
    int rc;
    if ((rc = allow_lds(m, hist_nib ? (const void*)resample_i8_nib_kernel : (hist_byte ? (const void*)resample_i8_kernel<true> : (const void*)resample_i8_kernel<false>), hist_bytes))) return rc;
    const auto resample_k = hist_byte ? resample_i8_kernel<true> : resample_i8_kernel<false>;
    const int nib_slow = m->tune.i8_nibbles == 2 ? 1 : 0;
    m->last_i8_nibbles = hist_nib ? 1 : 0;
    // threads per workgroup: the histogram decides how many workgroups share a CU (160 KB of LDS); the VALU-bound Philox loop wants the
    // CU's wave slots filled either way (N = 100,000: one 128 KB histogram per CU -- 256 threads left three quarters of the SIMD time idle)
    const unsigned resample_threads = (unsigned)std::min(1024, std::max(256, 256 * (int)(8 / std::max<size_t>(1, (160 * 1024) / std::max<size_t>(1, hist_bytes)))));
    if (m->tune.resample_aux && !m->aux) {
        int lo = 0, hi = 0;
        HIPCHK(m, hipDeviceGetStreamPriorityRange(&lo, &hi));                // (numerically: lowest priority first)
        HIPCHK(m, hipStreamCreateWithPriority(&m->aux, hipStreamNonBlocking, m->tune.resample_aux == 2 ? 0 : (m->tune.resample_aux == 3 ? hi : lo)));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(m, hipEventCreateWithFlags(&m->ev_counts[k], hipEventDisableTiming));
            HIPCHK(m, hipEventCreateWithFlags(&m->ev_cdfree[k], hipEventDisableTiming));
        }
        if ((rc = ensure(m, m->err2, sizeof(int)))) return rc;
        HIPCHK(m, hipMemsetAsync(m->err2.p, 0, sizeof(int), m->aux));
    }
    // counts of this chunk: the buffer the Gram before last read; grown only with both streams idle
    const int slot = m->aux ? (m->cd_slot ^= 1) : 0;
    plspm_model::Buf& cd = slot ? m->cd1 : m->cd;
    const size_t cd_bytes = (size_t)MT * 16 * ((size_t)KB + I8_SLACK_KB) * 64;
    if (cd_bytes > cd.cap) { if (m->aux) HIPCHK(m, hipStreamSynchronize(m->aux)); if ((rc = ensure(m, cd, cd_bytes))) return rc; m->cdfree_set[slot] = false; }
    if (!d_idx && m->aux) {
        // Philox draws: on the low-priority stream, as soon as the Gram that last read this buffer is done -- i.e. beside the Gram and the
        // solver of the PREVIOUS call when the host runs ahead; this call's Gram waits for the counts by event
        if (m->cdfree_set[slot]) HIPCHK(m, hipStreamWaitEvent(m->aux, m->ev_cdfree[slot], 0));
        {
            ProfScope ps(m, PLSPM_K_RESAMPLE, m->aux);
            if (hist_nib) hipLaunchKernelGGL(resample_i8_nib_kernel, dim3((unsigned)nb, hist_windows), dim3(resample_threads), hist_bytes, m->aux, (int)m->N, KB, MT, seed, rep0, (uint4*)cd.p, (int*)m->err2.p, nib_slow);
            else
            hipLaunchKernelGGL(resample_k, dim3((unsigned)nb, hist_windows), dim3(resample_threads), hist_bytes, m->aux, (int)m->N, KB, MT, m->tune.i8_shape, d_idx, seed, rep0, (uint4*)cd.p, (int*)m->err2.p);
        }
        HIPCHK(m, hipEventRecord(m->ev_counts[slot], m->aux));
        HIPCHK(m, hipStreamWaitEvent(m->stream, m->ev_counts[slot], 0));
    } else {
        // explicit index lists (test / parity seam) arrive on the main stream: drawn there, and the host looks at the flag
        if (m->aux && m->cdfree_set[slot]) HIPCHK(m, hipStreamWaitEvent(m->stream, m->ev_cdfree[slot], 0));
        ProfScope ps(m, PLSPM_K_RESAMPLE);
        if (hist_nib) hipLaunchKernelGGL(resample_i8_nib_kernel, dim3((unsigned)nb, hist_windows), dim3(resample_threads), hist_bytes, m->stream, (int)m->N, KB, MT, seed, rep0, (uint4*)cd.p, (int*)m->err.p, nib_slow);
        else
        hipLaunchKernelGGL(resample_k, dim3((unsigned)nb, hist_windows), dim3(resample_threads), hist_bytes, m->stream, (int)m->N, KB, MT, m->tune.i8_shape, d_idx, seed, rep0, (uint4*)cd.p, (int*)m->err.p);
    }
    if (d_idx) {
        int* h_err = (int*)m->h_flag + 9;
        HIPCHK(m, hipMemcpyAsync(h_err, m->err.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (*h_err & 2) {
            const int keep = *h_err & 1;
            HIPCHK(m, hipMemcpyAsync(m->err.p, &keep, sizeof(int), hipMemcpyHostToDevice, m->stream));
            HIPCHK(m, hipStreamSynchronize(m->stream));
            *fallback = true;
            return 0;
        }
    }
    if (counts && m->tune.i8_shape == 16) { *counts = cd.p; *counts_MT = MT; }       // (16-row pieces: what nm_conv_dense_kernel<.., CNT8> reads)
    const int total = ntx * nty, per = rows2 ? (ntx * nty_tall + 7) / 8 + (ntx * nty_short + 7) / 8 : (total + 7) / 8;      // workgroups per XCD
    // packed: the tile-packed slots the LDS solver / impute kernel read; dense: [(Pg+1) x cov_ld(Pg)] row-major, upper triangle (rows solver)
    const bool to16 = out16 && wrote16 && ind && m->tune.i8_shape == 16;
    const int* d_dst = (const int*)m->pair_tab.p + (to16 ? 6 : (dense ? 4 : 3)) * (size_t)m->zs_npair;
    // (a mirrored second store per element cost 0.08 ms per 5,000 replicates of the metric benchmark: the rows solver reads the triangle
    //  instead.  Categorical problems, whose solver wants the full square: Gram 1.40 -> 2.07 ms per 1,000 problems with the mirrored
    //  stores against 0.4 ms saved in nmg_prepare's scatter -- not taken either)
    const int* d_dst2 = nullptr;
    const long out_stride = to16 ? (long)(m->Pg + 1) * ((m->Pg + 1 + 7) & ~7) : (dense ? cov_doubles(m->Pg) : packed_size(m->T));
    if (to16) { out = reinterpret_cast<double*>(out16); *wrote16 = true; }
    // persistent stream-K schedule (kernels_gram_i8.h gram_i8_sk_kernel): one workgroup per CU, whole CUs per XCD
    const bool sk = m->tune.i8_sched == 1 && m->tune.i8_shape == 16 && m->tune.i8_variant < 0 && S >= 5 && S <= 7;      // (S = 8 spills in the persistent kernel)
    int sk_grid = 0;
    if (sk) {
        if (!cu_count_of(m)) return fail(m, PLSPM_E_STATE, "hipDeviceGetAttribute(multiprocessor count) failed");
        sk_grid = std::max(8, (m->cu_count / 8) * 8);
        const size_t slot_bytes = (size_t)32 * S * 1024;                    // 16 count tiles x 2 pair groups x S planes x 1 KB of int32 per workgroup
        if ((size_t)sk_grid * slot_bytes > m->sk_partial.cap && (rc = ensure(m, m->sk_partial, (size_t)sk_grid * slot_bytes))) return rc;
        if (!m->sk_flags.p) {
            if ((rc = ensure(m, m->sk_flags, (size_t)2048 * 8 * sizeof(unsigned)))) return rc;
            HIPCHK(m, hipMemsetAsync(m->sk_flags.p, 0, (size_t)2048 * 8 * sizeof(unsigned), m->stream));
            m->sk_epoch = 0;
        }
    }
    // the buffer form of the LDS-DMA needs every byte offset of a workgroup's walk (incl. the slack k-blocks) below 4 GiB
    const bool dma_fits = (uint64_t)(KB + I8_SLACK_KB) * (uint64_t)std::max(MT, NT) * 1024ull < (1ull << 32);
    const bool dma_buffer = !priv && m->tune.i8_dma != 1 && dma_fits && m->tune.i8_shape == 16 && !sk && (!narrow || (wide20 && m->tune.i8_waves == 4)) && m->tune.i8_variant < 0;
    m->last_i8_dma = dma_buffer ? 2 : 1;
    m->last_i8_rt = RTg;
    m->last_i8_short = rows2 ? nty_short : 0;
    m->last_i8_priv = priv ? 1 : 0;
    m->last_i8_mt = MT;
    ProfScope ps(m, PLSPM_K_GRAM);
    // (release library: the eight-wave forms of the round-3 kernel only -- four waves measured equal, the 32x32x32 layout 16 % slower, the narrow
    //  128-replicate tile 2.6 % slower, the persistent stream-K launch neutral with a starvation hazard: DESIGN.md 7b; `make experiments` builds them)
#ifdef PLSPM_I8_EXPERIMENTS
#define I8_FOUR_WAVES (m->tune.i8_waves == 4)
#else
#define I8_FOUR_WAVES false
#endif
#define GI8SK(SS, WW)                                                                                                                        \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<SS, WW, I8_DEFAULT_VAR, 16>::LDS_BYTES;                                                              \
        if ((rc = allow_lds(m, (const void*)gram_i8_sk_kernel<SS, WW>, lds_bytes))) return rc;                                               \
        hipLaunchKernelGGL((gram_i8_sk_kernel<SS, WW>), dim3((unsigned)sk_grid), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p,    \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, \
                           (i32x4*)m->sk_partial.p, (unsigned*)m->sk_flags.p, ++m->sk_epoch, (int*)m->err.p);                                \
    }
#define GI8P(SS, MM, VV)                                                                                                                     \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8P<SS, MM, VV>::LDS_BYTES;                                                                             \
        auto kfn = nty_short ? gram_i8p_kernel<SS, MM, VV, true> : gram_i8p_kernel<SS, MM, VV, false>;                                       \
        if ((rc = allow_lds(m, (const void*)kfn, lds_bytes))) return rc;                                                                     \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(8 * per)), dim3(256), lds_bytes, m->stream, (const uint4*)cd.p,                               \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, (const double*)m->pair_scale.p, m->zs_npair, nb_launch, out, out_stride, nty_short); \
    }
    // round 5: the same tiles on ONE persistent workgroup per CU (gram_i8pp_kernel: tiles from a per-XCD counter, the next tile's prologue in front
    // of this tile's epilogue; "i8_persist" 0: the tiled launch of round 4, kept as the A/B reference)
#define GI8PP(SS, MM)                                                                                                                        \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8P<SS, MM, 64>::LDS_BYTES + 64;                                                                        \
        auto kfn = nty_short ? gram_i8pp_kernel<SS, MM, 64, true> : gram_i8pp_kernel<SS, MM, 64, false>;                                     \
        if ((rc = allow_lds(m, (const void*)kfn, lds_bytes))) return rc;                                                                     \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(8 * pp_wgs)), dim3(256), lds_bytes, m->stream, (const uint4*)cd.p,                            \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, (const double*)m->pair_scale.p, m->zs_npair, nb_launch, out, out_stride, nty_short, \
                           (GramI8PPCtl*)m->pp_ctl.p);                                                                                       \
    }
#ifdef PLSPM_I8_EXPERIMENTS
    // timing probe (results are garbage): "i8_nostore" 1 launches both forms of the private-count kernel with nrep = 0 -- every epilogue store (and the
    // fp64 recombination in front of it) is skipped; tools/persist_ab.py prices the epilogue of the tiled and of the persistent form with it
    const long nb_launch = m->tune.i8_nostore ? 0 : (long)nb;
#else
    const long nb_launch = (long)nb;
#endif
    const bool persist = priv && m->tune.i8_persist != 0 && (m->tune.i8_variant < 0 || m->tune.i8_variant == 64);
    int pp_wgs = 0;
    if (persist) {
        if (!cu_count_of(m)) return fail(m, PLSPM_E_STATE, "hipDeviceGetAttribute(multiprocessor count) failed");
        const int cus = std::max(8, m->tune.i8_cus > 0 ? std::min(m->tune.i8_cus, m->cu_count) : m->cu_count);
        pp_wgs = std::max(1, std::min(cus / 8, per));                       // one workgroup per CU of an XCD, never more than the XCD has tiles
        if (!m->pp_ctl.p) {
            if ((rc = ensure(m, m->pp_ctl, sizeof(GramI8PPCtl)))) return rc;
            HIPCHK(m, hipMemsetAsync(m->pp_ctl.p, 0, sizeof(GramI8PPCtl), m->stream));      // (once: the kernel leaves the counters at zero)
        }
    }
    m->last_i8_persist = persist ? 1 : 0;
    if (persist) {
        if (S == 6) GI8PP(6, 5) else GI8PP(7, 4)
    } else
    if (priv) {
#define GI8PX(VV) case VV: if (S == 6) GI8P(6, 5, VV) else GI8P(7, 4, VV) break;
#ifdef PLSPM_I8_EXPERIMENTS
        // "i8_variant" v >= 0: the template's VAR itself -- bits 0-3 ablations, bit 4 digit blocks through staging registers, bits 5-6 the filler
        // schedule (0 strides / 1 one per gap, DMAs last / 2 VMEM evenly spaced = the release kernel), bit 7 one barrier per two k-steps
        switch (m->tune.i8_variant < 0 ? 64 : m->tune.i8_variant) {
            GI8PX(0) GI8PX(1) GI8PX(2) GI8PX(4) GI8PX(8) GI8PX(15) GI8PX(16) GI8PX(32) GI8PX(128) GI8PX(192)
            GI8PX(64) GI8PX(65) GI8PX(66) GI8PX(68) GI8PX(72) GI8PX(69) GI8PX(77) GI8PX(79)
            default: return fail(m, PLSPM_E_ARG, "i8_variant: not a built variant of the private-count kernel"); }
#else
        switch (64) { GI8PX(64) }
#endif
    } else
#ifdef PLSPM_I8_EXPERIMENTS
    if (sk) {
        if (m->tune.i8_waves == 4) { switch (S) { case 5: GI8SK(5, 2) break; case 6: GI8SK(6, 2) break; default: GI8SK(7, 2) break; } }
        else { switch (S) { case 5: GI8SK(5, 4) break; case 6: GI8SK(6, 4) break; default: GI8SK(7, 4) break; } }
    } else
#endif
#define GI8VS(SS, WW, VV, SH)                                                                                                                \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<SS, WW, VV, SH>::LDS_BYTES;                                                                          \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<SS, WW, VV, SH>, lds_bytes))) return rc;                                          \
        hipLaunchKernelGGL((gram_i8_kernel<SS, WW, VV, SH>), dim3((unsigned)(8 * per)), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, 0); \
    }
#define GI8V(SS, WW, VV) GI8VS(SS, WW, VV, 16)
#define GI8(SS, WW) GI8V(SS, WW, I8_DEFAULT_VAR)
#define GI8B(SS, WW) GI8V(SS, WW, I8_DEFAULT_VAR + 800)
#ifdef PLSPM_I8_EXPERIMENTS       // every schedule variant of the 7-plane kernel (tools/i8_bench.py --variants; not in the release library)
#define GI8X(WW) switch (m->tune.i8_variant) { case 0: GI8V(7, WW, 0) break; case 3: GI8V(7, WW, 3) break; case 6: GI8V(7, WW, 6) break; case 4: GI8V(7, WW, 4) break; \
        case 803: GI8V(7, WW, 803) break; case 103: GI8V(7, WW, 103) break; case 203: GI8V(7, WW, 203) break; case 303: GI8V(7, WW, 303) break; case 403: GI8V(7, WW, 403) break; case 503: GI8V(7, WW, 503) break; case 703: GI8V(7, WW, 703) break; \
        case 12: GI8V(7, WW, 12) break; case 18: GI8V(7, WW, 18) break; case 21: GI8V(7, WW, 21) break; case 24: GI8V(7, WW, 24) break; case 30: GI8V(7, WW, 30) break; default: GI8V(7, WW, 33) break; }
    if (S == 7 && m->tune.i8_variant >= 0) { if (m->tune.i8_waves == 4) GI8X(2) else GI8X(4) } else
#endif
#define GI8RT(RR)                                                                                                                            \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<7, 2, I8_DEFAULT_VAR, 16, RR>::LDS_BYTES;                                                            \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<7, 2, I8_DEFAULT_VAR, 16, RR>, lds_bytes))) return rc;                             \
        hipLaunchKernelGGL((gram_i8_kernel<7, 2, I8_DEFAULT_VAR, 16, RR>), dim3((unsigned)(8 * per)), dim3(256), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, 0); \
    }
#define GI8IND(WW, VV)                                                                                                                       \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<7, WW, VV, 16, 16>::LDS_BYTES;                                                                       \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<7, WW, VV, 16, 16, true>, lds_bytes))) return rc;                                 \
        hipLaunchKernelGGL((gram_i8_kernel<7, WW, VV, 16, 16, true>), dim3((unsigned)(8 * per)), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, to16 ? -1 : 0); \
    }
    if (ind) {                   // one plane per pair group, seven groups per wave
#ifdef PLSPM_I8_EXPERIMENTS
        if (m->tune.i8_waves == 4) { if (dma_buffer) GI8IND(2, I8_DEFAULT_VAR + 800) else GI8IND(2, I8_DEFAULT_VAR) } else
#endif
        if (dma_buffer) GI8IND(4, I8_DEFAULT_VAR + 800) else GI8IND(4, I8_DEFAULT_VAR)
    } else
#define GI8RT20(WW, VV)                                                                                                                      \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<6, WW, VV, 16, 20>::LDS_BYTES;                                                                       \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<6, WW, VV, 16, 20>, lds_bytes))) return rc;                                       \
        if (nty_short && !GramI8<6, WW, VV, 16, 20>::MIX) return fail(m, PLSPM_E_STATE, "int8 Gram: short tile rows planned for a kernel form without them"); \
        hipLaunchKernelGGL((gram_i8_kernel<6, WW, VV, 16, 20>), dim3((unsigned)(8 * per)), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, nty_short); \
    }
#ifdef PLSPM_I8_EXPERIMENTS
#define GI8X20(VV) case VV: GI8RT20(4, VV) break;
    if (wide20 && m->tune.i8_variant >= 0 && m->tune.i8_waves == 8) {
        switch (m->tune.i8_variant) { GI8X20(0) GI8X20(1) GI8X20(3) GI8X20(4) GI8X20(5) GI8X20(6) GI8X20(7) GI8X20(12) GI8X20(13) GI8X20(18) GI8X20(21) GI8X20(103) GI8X20(203) GI8X20(403) GI8X20(703) default: return fail(m, PLSPM_E_ARG, "i8_variant: not built for the 320-replicate tile"); }
    } else
#endif
#ifdef PLSPM_I8_EXPERIMENTS
    if (wide20 && m->tune.i8_waves == 4) { if (dma_buffer) GI8RT20(2, I8_DEFAULT_VAR + 800) else GI8RT20(2, I8_DEFAULT_VAR) } else
    if (narrow && !wide20) GI8RT(8) else
    if (m->tune.i8_shape == 32 && S >= 5) {    // v_mfma_i32_32x32x32_i8: four waves (64 replicates x 32 pairs x S planes each)
        switch (S) { case 5: GI8VS(5, 2, I8_DEFAULT_VAR, 32) break; case 6: GI8VS(6, 2, I8_DEFAULT_VAR, 32) break; case 7: GI8VS(7, 2, I8_DEFAULT_VAR, 32) break; default: GI8VS(8, 2, I8_DEFAULT_VAR, 32) break; }
    } else
    if (m->tune.i8_waves == 4) {
        if (dma_buffer) { switch (S) { case 1: GI8B(1, 2) break; case 2: GI8B(2, 2) break; case 3: GI8B(3, 2) break; case 4: GI8B(4, 2) break; case 5: GI8B(5, 2) break; case 6: GI8B(6, 2) break; case 7: GI8B(7, 2) break; default: GI8B(8, 2) break; } }
        else { switch (S) { case 1: GI8(1, 2) break; case 2: GI8(2, 2) break; case 3: GI8(3, 2) break; case 4: GI8(4, 2) break; case 5: GI8(5, 2) break; case 6: GI8(6, 2) break; case 7: GI8(7, 2) break; default: GI8(8, 2) break; } }
    } else
#endif
    if (wide20) GI8RT20(4, I8_DEFAULT_VAR) else
    if (dma_buffer) {            // LDS-DMA as buffer_load ... lds (32-bit offsets from per-workgroup descriptors)
        switch (S) { case 1: GI8B(1, 4) break; case 2: GI8B(2, 4) break; case 3: GI8B(3, 4) break; case 4: GI8B(4, 4) break; case 5: GI8B(5, 4) break; case 6: GI8B(6, 4) break; case 7: GI8B(7, 4) break; default: GI8B(8, 4) break; }
    } else {
        switch (S) { case 1: GI8(1, 4) break; case 2: GI8(2, 4) break; case 3: GI8(3, 4) break; case 4: GI8(4, 4) break; case 5: GI8(5, 4) break; case 6: GI8(6, 4) break; case 7: GI8(7, 4) break; default: GI8(8, 4) break; }
    }
#undef GI8V
#undef GI8VS
#undef GI8
#undef GI8B
    HIPCHK(m, hipGetLastError());
    if (m->aux) { HIPCHK(m, hipEventRecord(m->ev_cdfree[slot], m->stream)); m->cdfree_set[slot] = true; }     // the counts buffer is free once this Gram has run
    return 0;
}

int64_t plspm_detail_round_units_peek(const plspm_model* m) {
    if (!m || !m->d_Xa || m->nonmetric || m->n_ind || !m->zs_valid || !m->cu_count || choose_gram_path(m, (int64_t)1 << 20) != 2) return 64;
    const int S = m->zs_S;
    if ((S != 6 && S != 7) || m->zs_ind || m->tune.i8_priv == 0) return 64;
    const int cus = std::max(8, m->tune.i8_cus > 0 ? std::min(m->tune.i8_cus, m->cu_count) : m->cu_count);
    const int ntx = std::max(1, m->zs_npg / 2), tall = (S == 6 ? 20 : 16) * 16;
    return (int64_t)std::max(1, cus / ntx) * tall;
}

int64_t plspm_detail_round_units(plspm_model* m) {
    if (!m || !m->d_Xa || m->nonmetric || m->n_ind || choose_gram_path(m, (int64_t)1 << 20) != 2) return 64;
    if (hipSetDevice(m->device) != hipSuccess || prepare_zs(m)) return 64;            // (the plane count decides the tile height; built once per upload anyway)
    if (!cu_count_of(m)) return 64;
    return plspm_detail_round_units_peek(m);
}

extern "C" {

int plspm_bootstrap_prepare(plspm_model_t* m) {
    if (!m) return PLSPM_E_ARG;
    if (!m->d_Xa || m->N < 2) return fail(m, PLSPM_E_STATE, "plspm_bootstrap_prepare: no data uploaded");
    HIPCHK(m, hipSetDevice(m->device));
    // what the first bootstrap call on this data would build before its first replicate: the digit planes of the pair products
    // (enqueue only; a model that takes the fp64 Gram has nothing to prepare)
    // (automatic plane count: the column statistics are enqueued and copied to pinned memory behind an event -- no host wait here; the
    //  planes are cut by the first bootstrap call, which finds the statistics on the host.  A fixed plane count has nothing to read back:
    //  everything is enqueued now)
    if (choose_gram_path(m, (int64_t)1 << 20) == 2) return m->tune.i8_slices == 0 ? prepare_zs_stats(m) : prepare_zs(m);
    return 0;
}

int plspm_gram_tile_plan(int64_t count_tiles, int64_t pair_tiles, int32_t cus, int32_t mix, int32_t* tall, int32_t* shrt) {
    if (!tall || !shrt || count_tiles < 1 || pair_tiles < 1 || cus < 1 || count_tiles > ((int64_t)1 << 26) || pair_tiles > ((int64_t)1 << 20)) return PLSPM_E_ARG;
    int a = 0, b = 0;
    const bool wide = i8_mix_plan((long)count_tiles, (long)pair_tiles, std::max(8, (int)cus), mix != 0, &a, &b);
    *tall = a; *shrt = b;
    return wide ? 1 : 0;
}

}  // extern "C"
