/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the
 * product path (panoptic-forecasting_amd/, libpfhip.so).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Scalar C restatement of the reference's point-cloud warp + z-buffered splat:
 *   /root/reference/panoptic_forecasting/models/pc_transform/pc_transform_model.py:26-150
 * Parity pinning: checked bit-for-bit (seg, depth, result2d) against outputs of
 * the reference itself, run in the build container with a stand-in for the
 * absent torch_scatter — fixtures tests/golden/g1_*.npz, generator
 * tests/golden/make_golden.py.  The scatter_min tie rule (lowest source index
 * wins) restates pytorch_scatter 2.0.5's CPU loop from its published source;
 * the reference holds no vectors for it => that one rule is "parity unpinned".
 *
 * Floating-point contract (must be compiled with -ffp-contract=off, no -ffast-math):
 * every small mat-vec is  acc = 0; acc = acc + M[r][k]*v[k]  for k ascending,
 * with a separately rounded fp32 multiply and add, and every divide is an IEEE
 * fp32 divide — this is what ATen's CPU bmm does for these 3x3·3x1 / 4x4·4x1
 * products and it decides which pixel floor()/ceil() select.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>


static inline void matvec3(const float *M, const float *v, float *o) {
    for (int r = 0; r < 3; ++r) {
        float acc = 0.0f;
        for (int k = 0; k < 3; ++k) { float p = M[r * 3 + k] * v[k]; acc = acc + p; }
        o[r] = acc;
    }
}
static inline void matvec4(const float *M, const float *v, float *o) {
    for (int r = 0; r < 4; ++r) {
        float acc = 0.0f;
        for (int k = 0; k < 4; ++k) { float p = M[r * 4 + k] * v[k]; acc = acc + p; }
        o[r] = acc;
    }
}

/* pc_transform_model.py:42-78 for one pixel of one input frame. */
static inline void project_point(const float *Kinv, const float *E, const float *T, const float *Einv,
                                 const float *K, float u, float v, float d,
                                 float *uo, float *vo, float *zo) {
    float p2[3] = {u, v, 1.0f}, ray[3];
    matvec3(Kinv, p2, ray);                                        /* :54 */
    float pc[4] = {ray[0] * d, ray[1] * d, ray[2] * d, 1.0f};      /* :55-59 */
    float pv[4], pt[4], pc2[4];
    matvec4(E, pc, pv);                                            /* :63 */
    matvec4(T, pv, pt);                                            /* :68 */
    matvec4(Einv, pt, pc2);                                        /* :71 */
    float c3[3] = {pc2[0] / pc2[3], pc2[1] / pc2[3], pc2[2] / pc2[3]}; /* :72 */
    float q[3];
    matvec3(K, c3, q);                                             /* :74 */
    *zo = c3[2];                                                   /* :73 */
    *uo = q[0] / q[2];                                             /* :75 */
    *vo = q[1] / q[2];
}

static inline int64_t clampi(int64_t x, int64_t lo, int64_t hi) { return x < lo ? lo : (x > hi ? hi : x); }

/*
 * All arrays are host pointers, C-contiguous.
 *   depth [B,T,H,W] f32, mask [B,T,H,W] u8 (0/1), seg [B,T,H,W,C] u8 (C = seg_channels, 1 or 3)
 *   Kinv,K [B,3,3]; E,Einv [B,4,4]; Tt [B,T,4,4]
 *   out_seg [B,H,W,C] u8, out_depth [B,H,W] f32, out_result2d [B,T,H,W,2] i64 (nullable)
 * Debug taps (nullable): dbg_uvz [B,T,N,3] f32 (u', v', z before the sentinel),
 *   dbg_inds [B,4,T,N] i64 scatter bins, dbg_arg [B,N] i64 winner e (4P when empty),
 *   dbg_zmax [1] f32 the batch-global max.
 * Returns 0, or -1 on allocation failure.
 */
int pfo_warp_splat(const float *depth, const uint8_t *mask, const uint8_t *seg, int seg_channels,
                   const float *Kinv, const float *E, const float *Tt, const float *Einv, const float *K,
                   int B, int T, int H, int W,
                   uint8_t *out_seg, float *out_depth, int64_t *out_result2d,
                   float *dbg_uvz, int64_t *dbg_inds, int64_t *dbg_arg, float *dbg_zmax) {
    const int64_t N = (int64_t)H * W, P = (int64_t)T * N, E4 = 4 * P;
    const int C = seg_channels;
    float *u = malloc(sizeof(float) * B * P), *v = malloc(sizeof(float) * B * P);
    float *z = malloc(sizeof(float) * B * P);
    uint8_t *valid = malloc((size_t)B * P);
    float *best = malloc(sizeof(float) * N);
    int64_t *arg = malloc(sizeof(int64_t) * N);
    if (!u || !v || !z || !valid || !best || !arg) { free(u); free(v); free(z); free(valid); free(best); free(arg); return -1; }

    /* steps 1-4 (:41-78) + validity (:83-89) + batch-global max (:105) */
    float zmax = -INFINITY;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const float *Tm = Tt + ((int64_t)b * T + t) * 16;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    int64_t n = (int64_t)y * W + x, i = ((int64_t)b * T + t) * N + n;
                    float uu, vv, zz;
                    project_point(Kinv + b * 9, E + b * 16, Tm, Einv + b * 16, K + b * 9,
                                  (float)x, (float)y, depth[i], &uu, &vv, &zz);
                    u[i] = uu; v[i] = vv; z[i] = zz;
                    int inb = (uu >= 0.0f) && (uu < (float)W) && (vv >= 0.0f) && (vv < (float)H);
                    valid[i] = (uint8_t)((mask[i] != 0) && (zz > 0.0f) && inb);
                    if (zz > zmax) zmax = zz;
                    if (dbg_uvz) { dbg_uvz[i * 3] = uu; dbg_uvz[i * 3 + 1] = vv; dbg_uvz[i * 3 + 2] = zz; }
                }
        }
    const float sentinel = zmax + 1.0f;                           /* :105 */
    if (dbg_zmax) *dbg_zmax = zmax;

    for (int b = 0; b < B; ++b) {
        const float *ub = u + b * P, *vb = v + b * P;
        float *zb = z + b * P;
        const uint8_t *vab = valid + b * P;
        for (int64_t p = 0; p < P; ++p) if (!vab[p]) zb[p] = sentinel;
        /* scatter_min (:118-119): out=+max, arg=4P; sequential e, strict '<' */
        for (int64_t n = 0; n < N; ++n) { best[n] = INFINITY; arg[n] = E4; }
        for (int r = 0; r < 4; ++r)
            for (int64_t p = 0; p < P; ++p) {
                float fu = (r & 2) ? ceilf(ub[p]) : floorf(ub[p]);   /* :107-110: r = 0 ff, 1 fc, 2 cf, 3 cc */
                float fv = (r & 1) ? ceilf(vb[p]) : floorf(vb[p]);
                int64_t xi = clampi((int64_t)fu, 0, W - 1);          /* :113-114 */
                int64_t yi = clampi((int64_t)fv, 0, H - 1);
                int64_t bin = yi * W + xi, e = (int64_t)r * P + p;   /* :112,:117 */
                if (dbg_inds) dbg_inds[(int64_t)b * E4 + e] = bin;
                if (r == 0 && out_result2d) {                        /* :147 */
                    out_result2d[((int64_t)b * P + p) * 2] = xi;
                    out_result2d[((int64_t)b * P + p) * 2 + 1] = yi;
                }
                if (zb[p] < best[bin]) { best[bin] = zb[p]; arg[bin] = e; }
            }
        /* winner gather (:120-139) */
        for (int64_t n = 0; n < N; ++n) {
            uint8_t *os = out_seg + ((int64_t)b * N + n) * C;
            if (dbg_arg) dbg_arg[(int64_t)b * N + n] = arg[n];
            if (arg[n] < E4) {
                int64_t p = arg[n] % P;
                out_depth[(int64_t)b * N + n] = zb[p];
                for (int c = 0; c < C; ++c)
                    os[c] = vab[p] ? seg[((int64_t)b * P + p) * C + c] : 0;   /* :133 */
            } else {
                out_depth[(int64_t)b * N + n] = -1.0f;                    /* :136-138 */
                for (int c = 0; c < C; ++c) os[c] = 0;
            }
        }
    }
    free(u); free(v); free(z); free(valid); free(best); free(arg);
    return 0;
}
