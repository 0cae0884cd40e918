// Per-layer choice between the two fp32 MFMA convolution kernels and their tile shapes.
#include "conv_mfma.h"

namespace pf {

ConvChoice g_conv_force = {0, 0, 0, 0};
int g_opt_use_tuned = 1;   // pf_set_option("use_tuned_table", 0/1)

namespace {
struct Tuned {
    int ks, cin, cout, hout, wout, B;
    ConvChoice c;
};
// Measured winners for the shapes of the 1024x2048 bg network (tools/tune_convs.py --emit on an MI355X); rows are
// only kept where they beat the cost model by > 3 %.
const Tuned kTuned[] = {
#include "conv_tuned.inc"
};

// conv_s4 (packed-pair sources): measured per-layer shapes with every eligible layer on conv_s4 (tools/tune_s4.py --emit).
// kind 5 = run it on conv_s4 with (p0 = cout tiles per workgroup, p1 = 8x64-pixel tiles) if the plan can give it S4 sources,
// kind 0 = the table's fp32-source kernel measured faster in situ: do not ask for S4 sources.
const Tuned kTunedS4[] = {
    {0, 0, 0, 0, 0, 0, {0, 0, 0, 0}},
#include "conv_s4_tuned.inc"
};

// LDS bytes of a conv_wave workgroup (mirrors WaveCfg in conv_wave.hip)
size_t wave_lds_bytes(int ks, int mh, int nt, int wk) {
    const int kc = wave_kc(ks), ih = mh + ks - 1, iw = 16 + (ks == 3 ? 8 : 0);
    const int raw = ih * iw, plane = (raw + 15) / 32 * 32 + 16, nit = (kc * (plane / 4) + 63) / 64;
    const size_t ring = (size_t)wk * 2 * nit * 256, red = (size_t)wk * mh * nt * 256;
    return 4 * (ring > red ? ring : red);
}
}  // namespace

// A layer of the architecture at an image size the tables were not measured at.  Both tables are keyed on the exact shapes of the
// 1024x2048 network; what a row's choice depends on is how many tiles its launch has, i.e. its pixel count B * hout * wout, not
// the image size as such (every kernel clips its tiles at the image edge).  So the layer (ks, cin, cout) - each occurs at ONE
// resolution of the measured network, and the conv_s4 table has a row for every layer at every measured batch - is looked up
// with the batch size at which the measured launch had the same number of pixels: 512x1024 at B = 16 takes the B = 4 rows.
// Launches with less than half the pixels of the smallest measured one (B = 1) are outside what the rows know: heuristics.
// Returns false when the layer is not in the table at all (another architecture) or that small; *exact = its measured size is this one.
namespace {
bool measured_geometry(int ks, int cin, int cout, int hout, int wout, int B, int *ht, int *wt, double *b_eq, bool *exact) {
    // a row measured at exactly this size wins over any other resolution of the same (ks, cin, cout) (today every layer occurs at one
    // resolution - tests/test_host_logic.py asserts it on both tables - but a table that gains a second size must not remap the first)
    const Tuned *first = nullptr;
    for (const Tuned &t : kTunedS4) {
        if (t.B <= 0 || t.ks != ks || t.cin != cin || t.cout != cout) continue;
        if (t.hout == hout && t.wout == wout) {
            *ht = hout;
            *wt = wout;
            *exact = true;
            *b_eq = (double)B;
            return true;
        }
        if (!first) first = &t;
    }
    if (!first) return false;
    *ht = first->hout;
    *wt = first->wout;
    *exact = false;
    *b_eq = (double)B * hout * wout / ((double)first->hout * first->wout);
    return *b_eq >= 0.5;
}
}  // namespace

// Row of the conv_s4 table for this layer at the measured batch size nearest to B (in ratio); false: no row at all.
bool choose_s4(int ks, int cin, int cout, int hout, int wout, int B, ConvChoice *out) {
    int ht, wt;
    double b_eq;
    bool exact;
    if (!measured_geometry(ks, cin, cout, hout, wout, B, &ht, &wt, &b_eq, &exact)) return false;
    const Tuned *best = nullptr;
    double best_d = 0.0;
    for (const Tuned &t : kTunedS4) {
        if (t.ks != ks || t.cin != cin || t.cout != cout || t.hout != ht || t.wout != wt || t.B <= 0) continue;
        const double d = t.B > b_eq ? (double)t.B / b_eq : b_eq / (double)t.B;
        if (!best || d < best_d || (d == best_d && t.B > best->B)) { best = &t; best_d = d; }
    }
    if (!best) return false;
    *out = best->c;
    return true;
}

// conv_pair.hip (an odd HarDBlock layer inside its consumer) instead of two conv_s4 launches?  mode = plan option fuse_pairs: 0 never,
// 2 wherever the kernel exists (tests, A/B runs), 1 (default) where it MEASURED faster (MI355X, 1024x2048 network, per pair, same box:
// profiles/r06_experiments.md).  The fused form does 1.17-1.21x the matrix work of the two launches (the odd layer on 360 + 24 positions
// per 256 pixels) for 0.68x their bytes, and both forms run at the same ~60 % of the matrix pipe's sustained rate - the pipe and the
// LDS fragment reads bound them, not the DMA the fusion saves - so a full chip loses exactly the extra matrix work:
//      pixels of the launch (B * H * W)   524 288+ (B >= 4 at 256x512)    131 072 (B = 1 at 256x512, B = 4 at 128x256)     32 768
//      <2, 1> pairs (C <= 32, P <= 16)    1.06 - 1.31 x the two launches   0.84 - 0.96 x                                    0.97 - 1.15
//      <3, 1> / <*, 2> pairs              1.3 - 2.8 x (register spills at 128 registers per lane)
// It wins where the chip is NOT full: one launch of 512 eight-wave workgroups instead of two launches of 512 four-wave ones.
bool pair_wanted(int p_cin, int p_cout, int c_cin, int c_cout, int h, int w, int B, int mode) {
    if (mode == 4) return conv_pair_merged_supports(c_cout, p_cout);
    if (mode >= 2) return true;
    if (mode <= 0) return false;
    const long px = (long)B * h * w;
    return c_cout <= 32 && p_cout <= 16 && px >= 65536 && px <= 196608;
}

ConvChoice choose_conv(int ks, int stride, int cin, int cout, int hout, int wout, int B, int need, int use_tuned) {
    if (stride != 1) return ConvChoice{1, 0, 0, 0};           // conv_dma, shape by its cost model
    // The table was measured at B = 1, 2, 4, 8, 16 and (round 5, the strict-fp32 model at the headline's sub-batch) 32; a layer
    // without a row at one of those: the heuristics below won there.  Any other batch size takes the decision of the nearest
    // measured one (in ratio; the larger on a tie; beyond the largest: the largest): the choice depends on how many tiles the
    // launch has, which moves slowly with B.
    int Bt = B, ht = hout, wt = wout;
    if (use_tuned) {
        // (a layer at another image size than the measured one: the batch at which the measured launch had as many pixels)
        double b_eq = (double)B;
        bool exact = true;
        if (!measured_geometry(ks, cin, cout, hout, wout, B, &ht, &wt, &b_eq, &exact)) { ht = hout; wt = wout; b_eq = (double)B; }
        const int measured[6] = {1, 2, 4, 8, 16, 32};
        Bt = measured[5];
        for (int m : measured)
            if ((double)m >= b_eq) { Bt = (m > 1 && (double)m != b_eq && b_eq * b_eq < (double)m * (m / 2)) ? m / 2 : m; break; }
    }
    // (a layer without a B = 32 row: the cost model / heuristics below measured faster there - rows are kept only where they win by
    //  > 3 % - so it does NOT fall back to its B = 16 row)
    for (const Tuned &t : kTuned)
        if (use_tuned && t.ks == ks && t.cin == cin && t.cout == cout && t.hout == ht && t.wout == wt && t.B == Bt &&
            (!(need & 2) || t.c.kind != 2 || t.c.p0 != 1))
            return t.c;
    // Untuned shape.  Large stride-1 3x3 layers go to the split kernel: on every measured shape with >= 64x128
    // pixels x 4 images it beat the fp32-MFMA kernels by 1.3-2.2x (profiles/README.md); cout tiles per workgroup by
    // channel count, 8x64 tiles where the image is wide enough to still fill the chip.  (The executor falls back to
    // conv_dma when the layer needs a fused epilogue or split_f16 is off.)
    if (ks == 3 && (need == 0) && (wout & 3) == 0 && (long)B * hout * wout >= 32768) {
        const int nt = cout <= 16 ? 1 : (cout <= 32 ? 2 : 3);
        const long tiles_wide = (long)B * ((hout + 7) / 8) * ((wout + 63) / 64) * (((cout + 15) / 16 + nt - 1) / nt);
        return ConvChoice{4, nt, (nt <= 2 && tiles_wide >= 512) ? 1 : 0, 0};
    }
    // Images of >= 256x512 pixels keep the barrier-synchronised kernel (its big shared tiles move the
    // fewest bytes); below that the wave-autonomous kernel wins everywhere it was measured.  Pick the tile with the
    // most operand reuse (rows x cout tiles) that still puts >= 2 waves on every SIMD of the chip.
    // (1x1 layers from 32 768 pixels up too - round 5: at 512x1024, B = 16 the wave kernel took the 1x1 layers of the 32x64 level; it
    //  cannot write the packed-pair layout, so every 3x3 layer of that level fell back to the fp32-source split kernel: 0.56 of
    //  2.76 ms.  The measured table of the 1024x2048 network makes the same cut: packed pairs down to 32x64 at B = 16.)
    const long px = (long)B * hout * wout;
    if (px >= 131072 || (ks == 1 && px >= 32768)) return ConvChoice{1, 0, 0, 0};
    const int ntiles = (cout + 15) / 16;
    ConvChoice best{2, (need & 2) ? 2 : 1, 1, 8};
    long best_reuse = -1, best_waves = -1;
    const int mhs[3] = {4, 2, 1}, wks[3] = {2, 4, 8};
    for (int mh : mhs)
        for (int nt = 2; nt >= 1; --nt) {
            if ((need & 2) && mh == 1) continue;
            if (nt > ntiles) continue;
            for (int wk : wks) {
                if (wave_lds_bytes(ks, mh, nt, wk) > 64 * 1024) continue;
                const long waves = (long)B * ((hout + mh - 1) / mh) * ((wout + 15) / 16) * ((ntiles + nt - 1) / nt) * wk;
                const long reuse = waves >= 2048 ? mh * nt : 0;
                if (reuse > best_reuse || (reuse == best_reuse && reuse == 0 && waves > best_waves)) {
                    best = ConvChoice{2, mh, nt, wk};
                    best_reuse = reuse;
                    best_waves = waves;
                }
            }
        }
    return best;
}

}  // namespace pf

static long long *g_probe_dev = nullptr;
namespace pf {
long long *probe_buffer() {
    if (!g_probe_dev) {
        if (hipMalloc((void **)&g_probe_dev, 64 * sizeof(long long)) != hipSuccess) return nullptr;
        (void)hipMemset(g_probe_dev, 0, 64 * sizeof(long long));
        (void)hipMemset(g_probe_dev + 60, 0xFF, sizeof(long long));
    }
    return g_probe_dev;
}
}  // namespace pf

/* PF_PROBE builds: copy out the 64 in-kernel timestamps recorded by the last probed launch (synchronises). */
extern "C" int pf_debug_probe_read(long long *host64) {
    if (!g_probe_dev) return PF_EINVAL;
    if (hipMemcpy(host64, g_probe_dev, 64 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return PF_EHIP;
    (void)hipMemset(g_probe_dev, 0, 64 * sizeof(long long));
    (void)hipMemset(g_probe_dev + 60, 0xFF, sizeof(long long));   // slot 60 is an atomicMin target
    return PF_OK;
}

extern "C" int pf_debug_force_conv(int kind, int p0, int p1, int p2) {
    pf::g_conv_force = pf::ConvChoice{kind, p0, p1, p2};
    return PF_OK;
}
