"""The compile-time schedule of the long-filter kernel's matrix phase (tsl-sdr_amd/csrc/mfm_v3l_plan.h) on the host: the header is
plain C++17, so the plans of the instances that matter are built by g++ and their invariants checked here -
  * every k-step's fragments are waited for with a count that covers exactly the LGKM operations issued behind its reads;
  * outside flush points a gap between two matrix instructions holds at most two fillers;
  * a column group's accumulators are read by the recombination no sooner than three matrix instructions behind their last write,
    and never after the next group but one has begun to overwrite them;
  * every filler (fragment requests, recombination, staging) is issued exactly once.
No GPU involved (mfm_kernel_v3l.hip's arithmetic is what tests/test_gpu_parity.py checks against the oracle)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECKER = r"""
#include <cstdio>
#include <cstdlib>
#include "mfm_v3l_plan.h"
static int bad = 0;
#define REQUIRE(c, ...) do { if (!(c)) { bad++; printf("FAIL %s: ", name); printf(__VA_ARGS__); printf("\n"); } } while (0)
template <int KQ, int NH, int NGC, int RB, bool IN8, int PF, bool DB, bool PI, bool CO, int NST, int STGM = (IN8 ? MFM3L_STG_I8 : MFM3L_STG_I16)>
void check(const char *name)
{
    static constexpr auto P = mfm3l_make_plan<KQ, NH, NGC, RB, IN8, PF, false, DB, PI, CO, NST, STGM>();
    constexpr int NMF = P.NMF, NS = NGC * KQ, RPK = IN8 ? 1 : 2, SPC = mfm3l_stg_ops(STGM);
    // emission order: prologue reads, then per matrix instruction [wait] MFMA fillers..., then the tail
    int nl = 0, rd_done[NS + 1] = {}, req_seen[NS + 1] = {}, stg_seen[9][20] = {}, rec_seen[NGC + 1] = {};
    for (int st = 0; st < PF && st < NS; st++) { nl += RPK; rd_done[st] = nl; req_seen[st] = RPK; }
    int last_write[NGC][2][3];
    for (auto &a : last_write) for (auto &b : a) for (int &c : b) c = -1;
    int first_mf_of_group[NGC + 2];
    for (int g = 0; g < NGC + 2; g++) first_mf_of_group[g] = NMF + 1000;
    for (int m = 0; m < NMF; m++) if (m < first_mf_of_group[P.mf[m].g]) first_mf_of_group[P.mf[m].g] = m;
    auto filler = [&](const mfm3l_fl &f, int m, bool in_tail) {
        switch (f.kind) {
        case MFM3L_F_RDH: nl++; req_seen[f.a]++; if (IN8) rd_done[f.a] = nl; break;
        case MFM3L_F_RDL: nl++; req_seen[f.a]++; rd_done[f.a] = nl; break;
        case MFM3L_F_TPW: nl++; break;
        case MFM3L_F_STG: stg_seen[f.a][f.b]++; if (mfm3l_stg_is_lds(STGM, f.b)) nl++; break;
        default: break;
        }
        if (f.kind == MFM3L_F_LA || f.kind == MFM3L_F_SH0 || f.kind == MFM3L_F_SH1 || f.kind == MFM3L_F_TPW) {
            if (f.c != MFM3L_G_PEND) {
                rec_seen[f.c]++;
                // its accumulators: finished, three matrix instructions ago (or 16 wait states in front: !DB and the tail)
                int lw = -1;
                for (int k = 0; k < 3; k++) if (last_write[f.c][f.a][k] > lw) lw = last_write[f.c][f.a][k];
                REQUIRE(lw >= 0, "group %d recombined before it was multiplied", f.c);
                if (DB && !in_tail) REQUIRE(m - lw >= 3, "m%d reads group %d's sums %d matrix instructions behind their last write", m, f.c, m - lw);
                // ... and the group after next has not begun (it would overwrite the set)
                if (DB && f.c + 2 < NGC) REQUIRE(m < first_mf_of_group[f.c + 2], "m%d: group %d's sums read after group %d began", m, f.c, f.c + 2);
            } else {
                rec_seen[NGC]++;
                if (NGC > 1) REQUIRE(m < first_mf_of_group[1], "the pending group's sums read after group 1 began (m%d)", m);
            }
        }
    };
    for (int m = 0; m < NMF; m++) {
        const mfm3l_mf d = P.mf[m];
        if (d.wait != 0xff) {
            REQUIRE(req_seen[d.step] == RPK, "step %d multiplied with %d of %d fragments requested", d.step, req_seen[d.step], RPK);
            const int n = nl - rd_done[d.step];
            REQUIRE(d.wait == (n > 15 ? 15 : n), "step %d: lgkmcnt(%d), %d operations are younger than its reads", d.step, d.wait, n);
        }
        last_write[d.g][d.r][d.prod == MFM3L_P_HH ? 0 : d.prod == MFM3L_P_LL ? 2 : 1] = m;
        int shadow = 0;
        bool flush = false;
        for (int i = P.gap_lo[m]; i < P.gap_hi[m]; i++) {
            if (P.fl[i].kind == MFM3L_F_NOP16) flush = true;
            filler(P.fl[i], m, flush);
            shadow++;
        }
        const bool last_of_group = m + 1 == NMF || P.mf[m + 1].g != d.g;
        if (!last_of_group) REQUIRE(shadow <= 2, "m%d has %d fillers in its gap", m, shadow);
    }
    bool nop = false;
    for (int i = P.tail_lo; i < P.tail_hi; i++) { if (P.fl[i].kind == MFM3L_F_NOP16) nop = true; filler(P.fl[i], NMF, nop); }
    for (int st = 0; st < NS; st++) REQUIRE(req_seen[st] == RPK, "step %d: %d fragment reads", st, req_seen[st]);
    for (int j = 0; j < NST; j++) for (int o = 0; o < SPC; o++) REQUIRE(stg_seen[j][o] == 1, "staging chunk %d op %d issued %d times", j, o, stg_seen[j][o]);
    const int nrec = mfm3l_rec_items(RB, IN8, NH);
    for (int g = 0; g < NGC; g++) {
        const int want = (g + 1 == NGC && CO) ? 0 : nrec;
        REQUIRE(rec_seen[g] == want, "group %d: %d of %d recombination fillers", g, rec_seen[g], want);
    }
    REQUIRE(rec_seen[NGC] == (PI ? nrec : 0), "pending group: %d recombination fillers", rec_seen[NGC]);
    printf("%s: %d matrix instructions, %d fillers, %d in gaps\n", name, NMF, P.total, P.shadowed);
}
int main()
{
    // configs[4]'s share (512 taps, D = 400: quarter-tile images, two row blocks, one accumulator set)
    check<16, 0, 1, 2, false, 2, false, false, false, 4>("kq16 nh0 ng1 rb2");
    check<16, 0, 1, 2, true, 2, false, false, false, 4>("kq16 nh0 ng1 rb2 in8");
    // north star's shape (128 taps, D = 96 on 128-channel slices): whole-tile image, two accumulator sets
    check<4, 2, 4, 2, false, 4, true, false, false, 4>("kq4 nh2 ng4 rb2");
    check<4, 2, 4, 2, true, 4, true, false, false, 4>("kq4 nh2 ng4 rb2 in8");
    // 256-tap POCSAG low-passes (etc/pocsag_1200khz_fs.json at D = 25: old-style staging; etc/pocsag_narrow.json at D = 100)
    check<11, 4, 4, 1, false, 4, true, false, false, 1, MFM3L_STG_I16_S>("kq11 nh4 ng4 rb1 split rows, one chunk");
    check<6, 2, 4, 1, true, 4, true, false, false, 4, MFM3L_STG_I8_S>("kq6 nh2 ng4 rb1 in8 split rows, four chunks");
    check<9, 4, 4, 1, false, 4, true, false, false, 4>("kq9 nh4 ng4 rb1");
    // 512 taps at D = 120 (etc/flex_25khz_lpf_3mhz.json): half-tile images, a group carried from the first to the second
    check<16, 2, 2, 1, false, 4, true, false, true, 4>("kq16 nh2 ng2 rb1 first image");
    check<16, 2, 2, 1, false, 4, true, true, false, 4>("kq16 nh2 ng2 rb1 second image");
    check<16, 16, 2, 1, false, 2, false, false, false, 8>("kq16 nh16 ng2 rb1 nch8");
    // the one-group image in front of a chunk
    check<4, 2, 1, 2, false, 4, false, false, false, 0>("front kq4");
    check<6, 0, 1, 1, true, 4, false, false, false, 0>("front kq6 in8");
    return bad ? 1 : 0;
}
"""


@pytest.mark.timeout(600)
def test_matrix_phase_plans_hold_their_invariants(tmp_path):
    src = tmp_path / "check_plan.cpp"
    src.write_text(CHECKER)
    exe = tmp_path / "check_plan"
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-fconstexpr-ops-limit=1000000000", "-fconstexpr-loop-limit=10000000",
                        "-I", os.path.join(ROOT, "tsl-sdr_amd", "csrc"), "-o", str(exe), str(src)],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout[-4000:]
    assert r.stdout.count("matrix instructions") == 12
