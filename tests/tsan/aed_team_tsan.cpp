// ThreadSanitizer harness for the helper team of the host window kernel (csrc/schur_host_team.h):
// schur_host.hip is host-only code, so it is compiled here as plain C++ with -fsanitize=thread and
// run on AED windows in the SN_AED_DUMP record format (int32 nw, float64 sub, thres, nw*nw column-major).
// Every window is reduced serially and with the team; the results must be bit-identical and the
// sanitizer silent.  Built and run by tests/test_schur_host_tsan.py (CPU suite).
#include "../../starneig_amd/csrc/schur_host.h"
#include "../../starneig_amd/csrc/tuning.h"
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace sn { Tuning const &tuning() { static Tuning t; return t; } }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s windows.bin helpers\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int const helpers = atoi(argv[2]);
    int windows = 0, bad = 0;
    for (;;) {
        int32_t nw; double sub, thres;
        if (fread(&nw, 4, 1, f) != 1) break;
        if (fread(&sub, 8, 1, f) != 1 || fread(&thres, 8, 1, f) != 1) break;
        std::vector<double> W((size_t)nw * nw);
        if (fread(W.data(), 8, W.size(), f) != W.size()) break;
        int const ld = nw + 8;
        std::vector<double> out[2][3];
        sn::host::AedResult res[2];
        for (int team = 0; team < 2; team++) {
            sn::host::helper_session(team != 0, helpers);
            std::vector<double> T((size_t)ld * nw, 0.0), Z((size_t)ld * nw, 0.0), spike(nw), sr(nw), si(nw);
            for (int j = 0; j < nw; j++) std::memcpy(&T[(size_t)j * ld], &W[(size_t)j * nw], (size_t)nw * 8);
            res[team] = sn::host::aed_window(nw, T.data(), ld, Z.data(), ld, sub, thres, spike.data(), sr.data(), si.data());
            sn::host::helper_session(false);
            out[team][0] = T; out[team][1] = Z; out[team][2] = spike;
        }
        bool same = res[0].deflated == res[1].deflated && res[0].shifts == res[1].shifts;
        for (int k = 0; k < 3 && same; k++)
            same = std::memcmp(out[0][k].data(), out[1][k].data(), out[0][k].size() * 8) == 0;
        printf("window %d: nw %d deflated %d, team %s serial\n", windows, nw, res[0].deflated, same ? "==" : "!=");
        bad += !same; windows++;
    }
    fclose(f);
    printf("%d windows, %d mismatches\n", windows, bad);
    return bad ? 1 : 0;
}
