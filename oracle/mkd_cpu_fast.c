/*
 * mkd_cpu_fast.c -- the MKD descriptor path organised for a CPU (AVX2 + FMA): what bench.py times as `cpu_baseline`.
 *
 * TEST / BENCHMARK INFRASTRUCTURE ONLY, like mkd_oracle.c: nothing under local-features_amd/ (the product) may include,
 * link or call this file.  It exists because the oracle proper follows the SHADERS' structure (fourteen pooling passes
 * that each recompute cos / sin per pixel, subgroup-shaped partial sums) -- the right thing for a checker, a soft target
 * for a baseline.  Here the same arithmetic is arranged the way one would write it for the host:
 *
 *   per patch: blur, gradient, magnitude, the shader's polynomial atan2 (branch-free), ONE sine / cosine per pixel
 *   (polynomial, 1e-7), harmonics by angle addition, the relative-angle streams by rotation with cos / sin (k phi)
 *   tables, then the pooling as 238 dot products of 1024 in register-blocked 7 x 2 tiles of 8-wide FMAs, normalisation,
 *   and the whitening as 128 dot products of 238.
 *
 * It implements the reference's stages with the same formulas (the file:line citations are those of mkd_oracle.c, whose
 * functions of the same name it mirrors) and must agree with the oracle to 2e-5 relative L2 per descriptor
 * (tests/test_oracle.py); it is NOT a parity reference itself.  The blur is evaluated as an fma chain (the oracle's
 * BLUR_CONTRACT reading, which is also the product's).
 *
 * Built with -ffp-contract=fast (mkd_oracle.c is built with =off), into the same libmkd_oracle.so.
 */
#include <math.h>
#include <pthread.h>
#include <string.h>

#define PS 32
#define NPX (PS * PS)
#define DIMS_IN 7
#define D_CART 9
#define D_POLAR 25
#define RAW 238
#define OUT 128
#define ATAN_SHADER 0

typedef struct {   /* ConstantData, shaders/common.glsl:34-40 (same layout as in mkd_oracle.c) */
    float gradient_angle[NPX];
    float embedding_polar[D_POLAR * NPX];
    float embedding_cartesian[D_CART * NPX];
    float mean_vec[RAW];
    float eigen_vecs[OUT * RAW];
} mkd_consts;

typedef float v8 __attribute__((vector_size(32), aligned(4)));

static const float VM_N3_K8[4] = {0.37872374f, 0.51796234f, 0.46882015f, 0.39798096f};   /* mkd_ref.rs:7 */

/* shaders/mkd/patch_gradients.glsl:72-104: 5-tap blur (fma chain, taps in the shader's order), gx = left - right,
 * gy = down - up with the border replicated, mag = (gx^2 + gy^2 + 1e-8)^(1/4) */
static void gradients(const float *patch, float *gx, float *gy, float *mag)
{
    static const float k[5] = {0.0096f, 0.2054f, 0.5699f, 0.2054f, 0.0096f};
    float tmp[PS][PS + 4], bl[PS + 2][PS + 2];
    for (int y = 0; y < PS; y++) {
        const float *r[5];
        for (int i = 0; i < 5; i++) {
            int yy = y + i - 2;
            yy = yy < 0 ? 0 : (yy > PS - 1 ? PS - 1 : yy);
            r[i] = patch + yy * PS;
        }
        for (int x = 0; x < PS; x++) {
            float s = 0.f;
            for (int i = 0; i < 5; i++) s = __builtin_fmaf(k[i], r[i][x], s);
            tmp[y][x + 2] = s;
        }
        tmp[y][0] = tmp[y][1] = tmp[y][2];
        tmp[y][PS + 2] = tmp[y][PS + 3] = tmp[y][PS + 1];
    }
    for (int y = 0; y < PS; y++) {
        for (int x = 0; x < PS; x++) {
            float s = 0.f;
            for (int i = 0; i < 5; i++) s = __builtin_fmaf(k[i], tmp[y][x + i], s);
            bl[y + 1][x + 1] = s;
        }
        bl[y + 1][0] = bl[y + 1][1];
        bl[y + 1][PS + 1] = bl[y + 1][PS];
    }
    memcpy(bl[0], bl[1], sizeof bl[0]);
    memcpy(bl[PS + 1], bl[PS], sizeof bl[0]);
    for (int y = 0; y < PS; y++)
        for (int x = 0; x < PS; x++) {
            /* at the border the replicated texel is the centre one: left - right = b[x] - b[x+1] etc. */
            const float l = x == 0 ? bl[y + 1][1] : bl[y + 1][x], rr = x == PS - 1 ? bl[y + 1][PS] : bl[y + 1][x + 2];
            const float dn = y == PS - 1 ? bl[PS][x + 1] : bl[y + 2][x + 1], up = y == 0 ? bl[1][x + 1] : bl[y][x + 1];
            const float a = l - rr, b = dn - up;
            gx[y * PS + x] = a;
            gy[y * PS + x] = b;
            mag[y * PS + x] = sqrtf(sqrtf(a * a + b * b + 1e-8f));
        }
}

/* shaders/atan2.glsl (the oracle's mkd_oracle_atan2_shader), branch-free; returns the ORIENTATION -atan2 */
static inline float neg_atan2_shader(float x, float y)
{
    const float A1 = 0.99997726f, A3 = -0.33262347f, A5 = 0.19354346f, A7 = -0.11643287f, A9 = 0.05265332f,
                A11 = -0.0117212f, PI_F = 3.1415927f, FRAC_PI_2 = 1.5707964f;
    const int swap = fabsf(x) < fabsf(y);
    const float num = swap ? x : y, den = swap ? y : x;
    const float a = den == 0.f ? 0.f : num / den;                 /* x == y == 0 -> 0 */
    const float q = a * a;
    const float p = a * (A1 + q * (A3 + q * (A5 + q * (A7 + q * (A9 + q * A11)))));
    const float sa = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);      /* sign(0) = 0: the gx == 0 quirk */
    float res = swap ? FRAC_PI_2 * sa - p : p;
    res += x < 0.f ? (y < 0.f ? -PI_F : PI_F) : 0.f;
    return -res;
}

/* sine and cosine of t in [-2 pi, 2 pi], 1e-7 (Cody-Waite reduction by pi/2, minimax polynomials on [-pi/4, pi/4]) */
static inline void sincos_poly(float t, float *s, float *c)
{
    const float qf = nearbyintf(t * 0.63661975f);
    float r = __builtin_fmaf(-qf, 1.5707962513f, t);
    r = __builtin_fmaf(-qf, 7.5497894159e-08f, r);
    const float r2 = r * r;
    const float sn = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
    const float cs = 1.f - 0.5f * r2 + r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    const int n = (int)qf & 3;
    const float a = (n & 1) ? cs : sn, b = (n & 1) ? sn : cs;
    *s = (n & 2) ? -a : a;
    *c = ((n + 1) & 2) ? -b : b;
}

/* out[s * D + j] = sum_px A[s][px] * E[j][px]   (embedding.glsl:53-121), 7 x 2 register tiles of 8-wide FMAs */
static void pool(const float *A, const float *E, int D, float *out)
{
    for (int j = 0; j < D; j += 2) {
        const float *e0 = E + j * NPX, *e1 = E + (j + 1 < D ? j + 1 : j) * NPX;
        v8 acc[DIMS_IN][2];
        for (int s = 0; s < DIMS_IN; s++) acc[s][0] = acc[s][1] = (v8){0, 0, 0, 0, 0, 0, 0, 0};
        for (int px = 0; px < NPX; px += 8) {
            const v8 b0 = *(const v8 *)(e0 + px), b1 = *(const v8 *)(e1 + px);
            for (int s = 0; s < DIMS_IN; s++) {
                const v8 a = *(const v8 *)(A + s * NPX + px);
                acc[s][0] += a * b0;
                acc[s][1] += a * b1;
            }
        }
        for (int s = 0; s < DIMS_IN; s++)
            for (int jj = 0; jj < 2 && j + jj < D; jj++) {
                const v8 v = acc[s][jj];
                out[s * D + j + jj] = ((v[0] + v[4]) + (v[1] + v[5])) + ((v[2] + v[6]) + (v[3] + v[7]));
            }
    }
}

static float dot(const float *a, const float *b, int n)
{
    v8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= n; i += 8) acc += *(const v8 *)(a + i) * *(const v8 *)(b + i);
    float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    for (; i < n; i++) s += a[i] * b[i];
    return s;
}

typedef struct {
    const mkd_consts *c;
    const float *cphi, *sphi;   /* [3][NPX]: cos / sin (k phi), k = 1..3 */
    const float *patches;
    float *desc;
    long begin, end;
    int atan_mode;
} job_t;

static void describe_one(const job_t *jb, const float *patch, float *desc)
{
    const mkd_consts *c = jb->c;
    float gx[NPX], gy[NPX], mag[NPX];
    float ab[DIMS_IN * NPX], rl[DIMS_IN * NPX];   /* streams: absolute angle (cartesian kernels), relative (polar) */
    gradients(patch, gx, gy, mag);
    float th[NPX];
    if (jb->atan_mode == ATAN_SHADER)
        for (int px = 0; px < NPX; px++) th[px] = neg_atan2_shader(gx[px], gy[px]);
    else
        for (int px = 0; px < NPX; px++) th[px] = -atan2f(gy[px], gx[px]);
    const float k0 = VM_N3_K8[0], k1 = VM_N3_K8[1], k2 = VM_N3_K8[2], k3 = VM_N3_K8[3];
    const float *restrict cp1 = jb->cphi, *restrict cp2 = jb->cphi + NPX, *restrict cp3 = jb->cphi + 2 * NPX;
    const float *restrict sp1 = jb->sphi, *restrict sp2 = jb->sphi + NPX, *restrict sp3 = jb->sphi + 2 * NPX;
    for (int px = 0; px < NPX; px++) {   /* (no calls, no inner loops: one vectorised pass) */
        float s1, c1;
        sincos_poly(th[px], &s1, &c1);
        /* embedding.glsl:34-51 von_mises_n3k8: [c0, c_k cos k t, c_k sin k t] * mag; k t by angle addition */
        const float c2 = c1 * c1 - s1 * s1, s2 = 2.f * s1 * c1;
        const float c3 = c2 * c1 - s2 * s1, s3 = s2 * c1 + c2 * s1;
        const float m = mag[px];
        const float a1 = m * k1 * c1, a2 = m * k2 * c2, a3 = m * k3 * c3;
        const float b1 = m * k1 * s1, b2 = m * k2 * s2, b3 = m * k3 * s3;
        ab[px] = rl[px] = m * k0;
        ab[1 * NPX + px] = a1; ab[2 * NPX + px] = a2; ab[3 * NPX + px] = a3;
        ab[4 * NPX + px] = b1; ab[5 * NPX + px] = b2; ab[6 * NPX + px] = b3;
        /* embedding.glsl:70-77: the polar kernels see t + phi(px) */
        rl[1 * NPX + px] = a1 * cp1[px] - b1 * sp1[px];
        rl[2 * NPX + px] = a2 * cp2[px] - b2 * sp2[px];
        rl[3 * NPX + px] = a3 * cp3[px] - b3 * sp3[px];
        rl[4 * NPX + px] = b1 * cp1[px] + a1 * sp1[px];
        rl[5 * NPX + px] = b2 * cp2[px] + a2 * sp2[px];
        rl[6 * NPX + px] = b3 * cp3[px] + a3 * sp3[px];
    }
    float polar[DIMS_IN * D_POLAR], cart[DIMS_IN * D_CART], raw[RAW + 2];
    pool(rl, c->embedding_polar, D_POLAR, polar);
    pool(ab, c->embedding_cartesian, D_CART, cart);
    /* normalize.glsl:22-142: each block to unit norm, then the concatenation */
    const float np = sqrtf(dot(polar, polar, DIMS_IN * D_POLAR)), nc = sqrtf(dot(cart, cart, DIMS_IN * D_CART));
    float s = 0.f;
    for (int i = 0; i < RAW; i++) {
        raw[i] = i < DIMS_IN * D_POLAR ? polar[i] / np : cart[i - DIMS_IN * D_POLAR] / nc;
        s += raw[i] * raw[i];
    }
    const float n = sqrtf(s);
    /* whitening.glsl:22-77, normalize_final.glsl:17-59 */
    for (int i = 0; i < RAW; i++) raw[i] = raw[i] / n - c->mean_vec[i];
    float q = 0.f;
    for (int r = 0; r < OUT; r++) {
        desc[r] = dot(raw, c->eigen_vecs + r * RAW, RAW);
        q += desc[r] * desc[r];
    }
    const float qn = sqrtf(q);
    for (int r = 0; r < OUT; r++) desc[r] = desc[r] / qn;
}

static void *worker(void *arg)
{
    const job_t *j = (const job_t *)arg;
    for (long i = j->begin; i < j->end; i++) describe_one(j, j->patches + i * NPX, j->desc + i * OUT);
    return NULL;
}

void mkd_cpu_fast_describe_patches(const mkd_consts *c, const float *patches, long n, float *desc, int atan_mode,
                                   int nthreads)
{
    static float cphi[3 * NPX], sphi[3 * NPX];
    for (int k = 0; k < 3; k++)
        for (int px = 0; px < NPX; px++) {
            cphi[k * NPX + px] = cosf((float)(k + 1) * c->gradient_angle[px]);
            sphi[k * NPX + px] = sinf((float)(k + 1) * c->gradient_angle[px]);
        }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){c, cphi, sphi, patches, desc, n * t / nthreads, n * (t + 1) / nthreads, atan_mode & 1};
        if (nthreads == 1) worker(&jobs[t]);
        else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
