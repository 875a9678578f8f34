// solver_hoc.h -- second stage of the two-stage higher-order-construct estimation, on the moments.
//
// TEST NOTE: like solver_core.h this source is compiled twice -- for the GPU (plspm_nonmetric.hip / plspm_fit.hip) and for the std::thread emulation
// build that the CPU tests drive (tests/hostemu).
//
// Reference: Estimator.estimate (plspm/estimator.py:29-55).  Stage 1 estimates the model in which every higher order
// construct (HOC) is replaced by its constituent LVs (hoc_path_first_stage, estimator.py:60-74); stage 2 re-estimates the
// original path model with the stage-1 SCORES of the constituents as the HOC's manifest variables (Scale.NUM, estimator.py:43-52).
// The bootstrap repeats both stages on every resampled data set (bootstrap.py:57).
//
// A stage-1 score is an affine map of the uploaded columns (y_j = sum_{p in block j} c1_p x_p + k1_j, NmState::c_new / k_new),
// so the stage-2 moment matrix is a congruence of the replicate's stage-1 Gram M1 (aug: P1 columns + ones):
//     M2 = A' M1 A,     column a of A:  plain MV -> e_{p(a)};   HOC MV (constituent j) -> (c1 restricted to block j, k1_j on the ones row);
//     ones -> e_ones.
// The stage-1 LV order is the stage-2 (original) order with every HOC expanded in place, so stage-2 LV l stands for the
// stage-1 LVs [lv_first[l], lv_first[l+1]) and its columns for one contiguous range of stage-1 columns.
#pragma once
#include "solver_core.h"

namespace plspm {

struct HocDesc {
    int P1, L1, P2, L2, T1, T2;
    const int* boff1;       // [L1+1] stage-1 column blocks
    const int* boff2;       // [L2+1] stage-2 column blocks
    const int* lv_first;    // [L2+1] stage-1 LV range of every stage-2 LV
    const int* col2_lv1;    // [P2] for a HOC column: its stage-1 constituent LV j; for a plain column: -1
    const int* col2_p1;     // [P2] for a plain column: its stage-1 column; for a HOC column: -1
    int nh;                 // number of HOC columns
    const int* hcol;        // [nh] stage-2 column of every HOC column;   hidx[a] below is its inverse
    const int* hidx;        // [P2] index into hcol, or -1
};

// `V`: nh * (P1 + 1) doubles of scratch.  `ok`: stage 1 finished with status OK (otherwise the replicate is poisoned with NaN
// and stage 2 reports PLSPM_NONFINITE -- the reference drops such a replicate, bootstrap.py:65-66).
template <class Ex>
PLSPM_HD void hoc_second_stage_moments(Ex& ex, const HocDesc& hd, const double* M1, const double* c1, const double* k1, bool ok, double* M2, double* V) {
    const int P1 = hd.P1, P2 = hd.P2, Q1 = P1 + 1;
    // V[h][q] = sum_{p in supp(h)} alpha_p M1(p, q),   q = 0..P1 (P1 = ones)
    ex.par(hd.nh * Q1, [&](int e) {
        const int h = e / Q1, q = e - h * Q1;
        const int j = hd.col2_lv1[hd.hcol[h]];
        double s = k1[j] * M1[packed_index(hd.T1, P1, q)];
        for (int p = hd.boff1[j]; p < hd.boff1[j + 1]; ++p) s += c1[p] * M1[packed_index(hd.T1, p, q)];
        V[e] = s;
    });
    const int ntile = hd.T2 * (hd.T2 + 1) / 2;
    ex.par(ntile * 256, [&](int e) {
        const int tile = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        int t = 0, rem = tile;
        while (rem >= hd.T2 - t) { rem -= hd.T2 - t; ++t; }
        int a, b;
        packed_coords(hd.T2, t, t + rem, r, lane, a, b);
        double v = 0.0;
        if (a <= P2 && b <= P2) {
            if (!ok) v = NAN;
            else {
                const int ha = a < P2 ? hd.hidx[a] : -1, hb = b < P2 ? hd.hidx[b] : -1;
                const int pa = a < P2 ? hd.col2_p1[a] : P1, pb = b < P2 ? hd.col2_p1[b] : P1;      // ones -> ones
                if (ha < 0 && hb < 0) v = M1[packed_index(hd.T1, pa, pb)];
                else if (ha < 0) v = V[hb * Q1 + pa];
                else if (hb < 0) v = V[ha * Q1 + pb];
                else {
                    const int h1 = ha <= hb ? ha : hb, h2 = ha <= hb ? hb : ha;                     // one evaluation order: exactly symmetric
                    const int j = hd.col2_lv1[hd.hcol[h2]];
                    double s = k1[j] * V[h1 * Q1 + P1];
                    for (int q = hd.boff1[j]; q < hd.boff1[j + 1]; ++q) s += c1[q] * V[h1 * Q1 + q];
                    v = s;
                }
            }
        }
        M2[e] = v;
    });
}

// Score maps of stage 2 expressed on the STAGE-1 columns, in the layout nm_conv_kernel reads (an NmState head with P = P1,
// L = L2: scal[8] | a_old a_new (unused) | c_old c_new [P1] | k_old k_new [L2]): the streaming convergence pass of stage 2 then
// runs on the uploaded data unchanged.  st2: stage-2 state (its c / k are on the stage-2 columns).
template <class Ex>
PLSPM_HD void hoc_compose_score_maps(Ex& ex, const HocDesc& hd, const double* c1, const double* k1, const NmState& st2, double* pseudo) {
    const int P1 = hd.P1, L2 = hd.L2;
    double* co = pseudo + 8 + 2 * P1; double* cn = co + P1; double* ko = cn + P1; double* kn = ko + L2;
    ex.one([&]() { for (int i = 0; i < 8; ++i) pseudo[i] = st2.scal[i]; });
    ex.par(L2, [&](int l) {
        const int j0 = hd.lv_first[l], j1 = hd.lv_first[l + 1], a0 = hd.boff2[l];
        const bool plain = (hd.col2_lv1[a0] < 0);
        double so = st2.k_old[l], sn = st2.k_new[l];
        if (plain) {
            const int p0 = hd.boff1[j0];
            for (int a = a0; a < hd.boff2[l + 1]; ++a) { co[p0 + a - a0] = st2.c_old[a]; cn[p0 + a - a0] = st2.c_new[a]; }
        } else {
            for (int j = j0; j < j1; ++j) {
                const double fo = st2.c_old[a0 + j - j0], fn = st2.c_new[a0 + j - j0];
                so += fo * k1[j]; sn += fn * k1[j];
                for (int p = hd.boff1[j]; p < hd.boff1[j + 1]; ++p) { co[p] = fo * c1[p]; cn[p] = fn * c1[p]; }
            }
        }
        ko[l] = so; kn[l] = sn;
    });
}

}  // namespace plspm
