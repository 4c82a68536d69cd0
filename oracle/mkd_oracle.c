/*
 * mkd_oracle.c -- CPU restatement of the reference's MKD descriptor path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under local-features_amd/ (the product) may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline ("port").
 *
 * PARITY UNPINNED by the reference: tnibler/local-features ships no golden
 * vectors, no known-answer tests and no runnable implementation of this path in
 * this environment (Rust + Vulkan; its only tests are commented out and name
 * absent files, local_features/src/mkd_ref.rs:393-453).  What pins this file
 * instead is tools/gen_golden.py: an independent NumPy float64 restatement
 * written from the same sources, which must agree with this one to < 1e-5
 * relative L2 per descriptor (tests/test_oracle.py), and whose outputs are
 * committed under tests/golden/.
 *
 * Every function cites the reference file:line it follows.  Paths are relative
 * to /root/reference/local_features/src/ ("shaders/" = vulkan/shaders/).
 *
 * All arithmetic is IEEE f32; build with -ffp-contract=off so that a*b+c is a
 * rounded multiply followed by a rounded add, as the Rust CPU twin does.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PS 32           /* PATCH_SIZE            shaders/common.glsl:19 */
#define NPX (PS * PS)
#define DIMS_IN 7       /* DIMS_INPUT            shaders/common.glsl:22 */
#define D_CART 9        /* DIMS_EMB_CARTESIAN    shaders/common.glsl:23 */
#define D_POLAR 25      /* DIMS_EMB_POLAR        shaders/common.glsl:24 */
#define RAW 238         /* DESCRIPTOR_SIZE       shaders/common.glsl:33 */
#define OUT 128         /* PCAD_DESCRIPTOR_SIZE  shaders/common.glsl:20 */

#define ATAN_SHADER 0   /* polynomial atan2 incl. its quirks (parity gate)  */
#define ATAN_LIBM 1     /* exact atan2f, as the CPU twin mkd_ref.rs:140     */
/* OR-ed into atan_mode: evaluate the blur's `sum += k[i] * p` (patch_gradients.glsl:72-92) as fma(k[i], p, sum).
 * The shader carries no `precise` qualifier, so a GLSL compiler may contract or not; both are the reference.
 * It matters only through atan2.glsl's discontinuity at x == 0 (a pixel whose gx rounds to exactly 0 gets angle 0
 * instead of +-pi/2): see mkd_oracle_quirk_pixels. */
#define BLUR_CONTRACT 2

/* mkd_ref.rs:7-9 */
static const float VM_N3_K8[4] = {0.37872374f, 0.51796234f, 0.46882015f, 0.39798096f};
static const float VM_N1_K1[2] = {0.618176f, 0.6934725f};
static const float VM_N2_K8[3] = {0.37872374f, 0.51796234f, 0.46882015f};

/* Layout of ConstantData, shaders/common.glsl:34-40 */
typedef struct {
    float gradient_angle[NPX];
    float embedding_polar[D_POLAR * NPX];
    float embedding_cartesian[D_CART * NPX];
    float mean_vec[RAW];
    float eigen_vecs[OUT * RAW]; /* W_T row-major [128][238] */
} mkd_consts;

/* ------------------------------------------------------------------------- */
/* LUT builders: mkd_ref.rs:173-267                                          */
/* ------------------------------------------------------------------------- */

/* mkd_ref.rs:173-185 mesh_grid: grid[0]=x coordinate, grid[1]=y coordinate */
static void mesh_grid(float *gx, float *gy)
{
    for (int y = 0; y < PS; y++)
        for (int x = 0; x < PS; x++) {
            gx[y * PS + x] = 2.f * (float)x / ((float)PS - 1.f) - 1.f;
            gy[y * PS + x] = 2.f * (float)y / ((float)PS - 1.f) - 1.f;
        }
}

/* mkd_ref.rs:133-144 cart2pol: mag=sqrt(x^2+y^2+eps), angle=-atan2(y,x) */
static void cart2pol(const float *x, const float *y, float *mag, float *ang)
{
    const float eps = 1e-8f;
    for (int i = 0; i < NPX; i++) {
        mag[i] = sqrtf(x[i] * x[i] + y[i] * y[i] + eps);
        ang[i] = -atan2f(y[i], x[i]);
    }
}

/* mkd_ref.rs:146-171 von_mises: [c0, c_k cos(k t), c_k sin(k t)] k=1..n */
static void von_mises(const float *t, const float *coeffs, int n, float *out /*[2n+1][NPX]*/)
{
    for (int i = 0; i < NPX; i++) {
        out[i] = 1.f * coeffs[0];
        for (int k = 1; k <= n; k++) {
            const float a = t[i] * (float)k;
            out[k * NPX + i] = cosf(a) * coeffs[k];
            out[(n + k) * NPX + i] = sinf(a) * coeffs[k];
        }
    }
}

/* mkd_ref.rs:259-267 gaussian_weighting */
static void gaussian_weighting(float *g)
{
    float gx[NPX], gy[NPX], norm[NPX];
    mesh_grid(gx, gy);
    float mx = 0.f;
    for (int i = 0; i < NPX; i++) {
        norm[i] = sqrtf(gx[i] * gx[i] + gy[i] * gy[i]);
        if (norm[i] > mx) mx = norm[i];
    }
    for (int i = 0; i < NPX; i++) {
        const float nn = norm[i] / mx;
        g[i] = expf((nn * nn) / -(1.f * 1.f));
    }
}

/* mkd_ref.rs:210-231 spatial_kernel_embedding_cart, times the Gaussian
 * (vulkan/mod.rs:1614-1615) */
static void build_embedding_cart(float *ec /*[9][NPX]*/)
{
    float gx[NPX], gy[NPX], g[NPX];
    static float ea[3 * NPX], eb[3 * NPX];
    mesh_grid(gx, gy);
    for (int i = 0; i < NPX; i++) {
        gx[i] *= 1.57079632679489661923f;
        gy[i] *= 1.57079632679489661923f;
    }
    von_mises(gx, VM_N1_K1, 1, ea);
    von_mises(gy, VM_N1_K1, 1, eb);
    gaussian_weighting(g);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            for (int i = 0; i < NPX; i++)
                ec[(a * 3 + b) * NPX + i] = (ea[a * NPX + i] * eb[b * NPX + i]) * g[i];
}

/* mkd_ref.rs:233-257 spatial_kernel_embedding_polar, times the Gaussian
 * (vulkan/mod.rs:1616-1617) */
static void build_embedding_polar(float *ep /*[25][NPX]*/)
{
    float gx[NPX], gy[NPX], rho[NPX], phi[NPX], g[NPX];
    static float ea[5 * NPX], eb[5 * NPX];
    mesh_grid(gx, gy);
    cart2pol(gx, gy, rho, phi);
    for (int i = 0; i < NPX; i++) {
        rho[i] = rho[i] * 3.14159265358979323846f / 1.41421356237309504880f;
        phi[i] = phi[i] * -1.f;
    }
    von_mises(phi, VM_N2_K8, 2, ea);
    von_mises(rho, VM_N2_K8, 2, eb);
    gaussian_weighting(g);
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            for (int i = 0; i < NPX; i++)
                ep[(a * 5 + b) * NPX + i] = (ea[a * NPX + i] * eb[b * NPX + i]) * g[i];
}

/* vulkan/mod.rs:1594-1619 upload_constant_data: everything the shaders read */
void mkd_oracle_build_consts(const float *mean, const float *eigvals, const float *eigvecs,
                             mkd_consts *c)
{
    float gx[NPX], gy[NPX], mag[NPX];
    mesh_grid(gx, gy);
    cart2pol(gx, gy, mag, c->gradient_angle); /* mod.rs:1618-1619 */
    build_embedding_polar(c->embedding_polar);
    build_embedding_cart(c->embedding_cartesian);
    memcpy(c->mean_vec, mean, sizeof(float) * RAW);
    /* mod.rs:1604-1612: t=0.7, m=-0.5t, W = eigvecs[:, :128]*eigvals[:128]^m, stored transposed */
    const float t = 0.7f;
    const float m = -0.5f * t;
    for (int r = 0; r < OUT; r++) {
        const float s = powf(eigvals[r], m);
        for (int col = 0; col < RAW; col++)
            c->eigen_vecs[r * RAW + col] = eigvecs[col * RAW + r] * s;
    }
}

/* ------------------------------------------------------------------------- */
/* safetensors reader: mkd_ref.rs:352-391 (format: 8-byte LE header length,   */
/* JSON header, raw little-endian F32 tensors; crate safetensors 0.5.2,       */
/* Cargo.lock:3214-3216, is not in the reference tree -- format restated)     */
/* ------------------------------------------------------------------------- */
static int st_find(const char *hdr, const char *name, long *begin, long *end)
{
    char key[64];
    snprintf(key, sizeof key, "\"%s\"", name);
    const char *p = strstr(hdr, key);
    if (!p) return -1;
    p = strstr(p, "\"data_offsets\"");
    if (!p) return -1;
    p = strchr(p, '[');
    if (!p) return -1;
    if (sscanf(p, "[%ld,%ld]", begin, end) != 2) return -1;
    return 0;
}

int mkd_oracle_load_pca(const char *path, float *mean, float *eigvals, float *eigvecs)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t hlen = 0;
    if (fread(&hlen, 8, 1, f) != 1 || hlen > (1u << 20)) { fclose(f); return -2; }
    char *hdr = (char *)calloc(hlen + 1, 1);
    if (fread(hdr, 1, hlen, f) != hlen) { free(hdr); fclose(f); return -2; }
    const struct { const char *name; float *dst; long n; } t[3] = {
        {"mean", mean, RAW}, {"eigvals", eigvals, RAW}, {"eigvecs", eigvecs, RAW * RAW}};
    int rc = 0;
    for (int i = 0; i < 3 && rc == 0; i++) {
        long b, e;
        if (st_find(hdr, t[i].name, &b, &e) || e - b != t[i].n * 4) { rc = -3; break; }
        fseek(f, 8 + (long)hlen + b, SEEK_SET);
        if (fread(t[i].dst, 4, t[i].n, f) != (size_t)t[i].n) rc = -4;
    }
    free(hdr);
    fclose(f);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* shaders/atan2.glsl:19-46.  Signature atan2(x, y): angle of the point (x,y) */
/* ------------------------------------------------------------------------- */
static float glsl_sign(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

float mkd_oracle_atan2_shader(float x, float y)
{
    const float A1 = 0.99997726f, A3 = -0.33262347f, A5 = 0.19354346f, A7 = -0.11643287f,
                A9 = 0.05265332f, A11 = -0.0117212f;
    const float PI_F = 3.1415927f, FRAC_PI_2 = 1.5707964f;
    if (x == 0.f && y == 0.f) return 0.f;
    const int swap = fabsf(x) < fabsf(y);
    const float a = swap ? (x / y) : (y / x);
    const float asq = a * a;
    const float atan_res =
        a * (A1 + asq * (A3 + asq * (A5 + asq * (A7 + asq * (A9 + asq * A11)))));
    /* sign(a) is 0 when a == 0: atan2(0, y!=0) returns 0, not +-pi/2 (lines 33-38) */
    const float res = swap ? (FRAC_PI_2 * glsl_sign(a) - atan_res) : atan_res;
    if (glsl_sign(x) == -1.f) {
        /* bits(sign(y)) | bits(1.0): -1 for y<0, +1 otherwise (y==0 -> +1) */
        const float sign_y = (glsl_sign(y) == -1.f) ? -1.f : 1.f;
        return PI_F * sign_y + res;
    }
    return res;
}

/* ------------------------------------------------------------------------- */
/* shaders/mkd/patch_gradients.glsl:20-28,72-104 (blur + gradient + polar)    */
/* CPU twin: mkd_ref.rs:82-144,304-307                                        */
/* ------------------------------------------------------------------------- */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void blur_patch(const float *patch, float *blur, int contract)
{
    static const float k[5] = {0.0096f, 0.2054f, 0.5699f, 0.2054f, 0.0096f};
    float tmp[NPX];
    for (int y = 0; y < PS; y++) /* vertical pass, lines 72-81 */
        for (int x = 0; x < PS; x++) {
            float sum = 0.f;
            for (int i = 0; i < 5; i++) {
                const int yy = clampi(y + i - 2, 0, PS - 1);
                sum = contract ? fmaf(k[i], patch[yy * PS + x], sum) : sum + k[i] * patch[yy * PS + x];
            }
            tmp[y * PS + x] = sum;
        }
    for (int y = 0; y < PS; y++) /* horizontal pass, lines 83-92 */
        for (int x = 0; x < PS; x++) {
            float sum = 0.f;
            for (int i = 0; i < 5; i++) {
                const int xx = clampi(x + i - 2, 0, PS - 1);
                sum = contract ? fmaf(k[i], tmp[y * PS + xx], sum) : sum + k[i] * tmp[y * PS + xx];
            }
            blur[y * PS + x] = sum;
        }
}

/* Number of pixels of a patch that sit on atan2.glsl's discontinuity: |gx| <= tol (a few ulps of the blurred values)
 * in either of the two ways the blur may round.  At gx == 0 the shader's angle is 0 whatever gy is, next to it the angle
 * is +-pi/2 (gy != 0), or pi for a gx just below zero with gy == 0 -- so every such pixel counts, except the one case in
 * which nothing can flip: a gradient that is exactly null in both readings.  A descriptor of such a patch is decided by
 * last-bit rounding the reference leaves to its GLSL compiler; parity tests hold those patches to the contracted variant
 * when the input bits are shared and set them aside when they are not. */
int mkd_oracle_quirk_pixels(const float *patch, float tol)
{
    float blur[2][NPX];
    blur_patch(patch, blur[0], 0);
    blur_patch(patch, blur[1], 1);
    int n = 0;
    for (int y = 0; y < PS; y++)
        for (int x = 0; x < PS; x++) {
            int near = 0, null_both = 1;
            for (int v = 0; v < 2; v++) {
                const float *b = blur[v];
                const float gx = b[y * PS + clampi(x, 1, PS - 1) - 1] - b[y * PS + clampi(x, 0, PS - 2) + 1];
                const float gy = b[(clampi(y, 0, PS - 2) + 1) * PS + x] - b[(clampi(y, 1, PS - 1) - 1) * PS + x];
                near |= fabsf(gx) <= tol;
                null_both &= gx == 0.f && gy == 0.f;
            }
            n += near && !null_both;
        }
    return n;
}

void mkd_oracle_patch_gradients(const float *patch, float *mag, float *angle, int atan_mode)
{
    float blur[NPX];
    blur_patch(patch, blur, atan_mode & BLUR_CONTRACT);
    atan_mode &= ~BLUR_CONTRACT;
    for (int y = 0; y < PS; y++)
        for (int x = 0; x < PS; x++) {
            /* lines 94-96: gx = left - right, gy = down - up, replicate border */
            const float gx = blur[y * PS + clampi(x, 1, PS - 1) - 1] -
                             blur[y * PS + clampi(x, 0, PS - 2) + 1];
            const float gy = blur[(clampi(y, 0, PS - 2) + 1) * PS + x] -
                             blur[(clampi(y, 1, PS - 1) - 1) * PS + x];
            /* lines 98-101: mag = sqrt(sqrt(.)), the outer sqrt hoisted from the embedding */
            mag[y * PS + x] = sqrtf(sqrtf(gx * gx + gy * gy + 1e-8f));
            angle[y * PS + x] = (atan_mode == ATAN_SHADER) ? -mkd_oracle_atan2_shader(gx, gy)
                                                           : -atan2f(gy, gx);
        }
}

/* shaders/mkd/embedding.glsl:34-51 von_mises_n3k8 */
static float von_mises_n3k8(int i, float ori, float mag)
{
    if (i == 0) return VM_N3_K8[0] * mag;
    if (i < 4) return cosf((float)i * ori) * VM_N3_K8[i] * mag;
    return sinf(((float)i - 3.f) * ori) * VM_N3_K8[i - 3] * mag;
}

/* Sum 64 per-thread partial sums the way a 64-wide subgroupAdd tree would.
 * The reference leaves the order to the driver (embedding.glsl:90-111). */
static float tree64(float *v)
{
    for (int s = 32; s > 0; s >>= 1)
        for (int i = 0; i < s; i++) v[i] += v[i + s];
    return v[0];
}

/* shaders/mkd/embedding.glsl:53-121: pooled[i*D+j] = sum_px vm_i(px) * E_j(px).
 * One 32x2 workgroup per in_dim: thread (lx,ly) walks rows ly, ly+2, ... */
static void embed_pool(const float *mag, const float *angle, const float *phi /*or NULL*/,
                       const float *lut, int D, float *out /*[7*D]*/)
{
    for (int i = 0; i < DIMS_IN; i++) {
        static __thread float part[D_POLAR][64];
        memset(part, 0, sizeof part);
        for (int row = 0; row < PS; row += 2)
            for (int ly = 0; ly < 2; ly++)
                for (int lx = 0; lx < PS; lx++) {
                    const int px = (row + ly) * PS + lx;
                    const float ori = phi ? angle[px] + phi[px] : angle[px];
                    const float g = von_mises_n3k8(i, ori, mag[px]);
                    for (int j = 0; j < D; j++) part[j][ly * PS + lx] += g * lut[j * NPX + px];
                }
        for (int j = 0; j < D; j++) out[i * D + j] = tree64(part[j]);
    }
}

/* shaders/mkd/normalize.glsl:22-142; CPU twin mkd_ref.rs:313-326 */
static void normalize_raw(const float *polar, const float *cart, float *raw)
{
    float sp = 0.f, sc = 0.f;
    for (int i = 0; i < DIMS_IN * D_POLAR; i++) sp += polar[i] * polar[i];
    for (int i = 0; i < DIMS_IN * D_CART; i++) sc += cart[i] * cart[i];
    const float np = sqrtf(sp), nc = sqrtf(sc);
    float s = 0.f;
    for (int i = 0; i < RAW; i++) {
        const float v = i < DIMS_IN * D_POLAR ? polar[i] / np : cart[i - DIMS_IN * D_POLAR] / nc;
        raw[i] = v;
        s += v * v;
    }
    const float n = sqrtf(s);
    for (int i = 0; i < RAW; i++) raw[i] = raw[i] / n;
}

/* shaders/mkd/whitening.glsl:22-77 + normalize_final.glsl:17-59 */
static void whiten(const mkd_consts *c, const float *raw, float *out)
{
    float vec[RAW];
    for (int i = 0; i < RAW; i++) vec[i] = raw[i] - c->mean_vec[i];
    float s = 0.f;
    for (int r = 0; r < OUT; r++) {
        float acc = 0.f;
        for (int col = 0; col < RAW; col += 64) { /* one subgroupAdd per 64-column chunk */
            float p[64];
            for (int l = 0; l < 64; l++)
                p[l] = (col + l < RAW) ? vec[col + l] * c->eigen_vecs[r * RAW + col + l] : 0.f;
            acc += tree64(p);
        }
        out[r] = acc;
        s += acc * acc;
    }
    const float n = sqrtf(s);
    for (int r = 0; r < OUT; r++) out[r] = out[r] / n;
}

/* One patch through rows 3-9 of SURVEY.md section 8(a). raw238 may be NULL. */
void mkd_oracle_describe_patch(const mkd_consts *c, const float *patch, float *desc128,
                               float *raw238, int atan_mode)
{
    float mag[NPX], ang[NPX], polar[DIMS_IN * D_POLAR], cart[DIMS_IN * D_CART], raw[RAW];
    mkd_oracle_patch_gradients(patch, mag, ang, atan_mode);
    embed_pool(mag, ang, c->gradient_angle, c->embedding_polar, D_POLAR, polar);
    embed_pool(mag, ang, NULL, c->embedding_cartesian, D_CART, cart);
    normalize_raw(polar, cart, raw);
    if (raw238) memcpy(raw238, raw, sizeof raw);
    if (desc128) whiten(c, raw, desc128);
}

typedef struct {
    const mkd_consts *c;
    const float *patches;
    float *desc, *raw;
    long begin, end;
    int atan_mode;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (long i = j->begin; i < j->end; i++)
        mkd_oracle_describe_patch(j->c, j->patches + i * NPX, j->desc ? j->desc + i * OUT : NULL,
                                  j->raw ? j->raw + i * RAW : NULL, j->atan_mode);
    return NULL;
}

/* Batch entry point; nthreads>1 partitions patches over pthreads (CPU baseline). */
void mkd_oracle_describe_patches(const mkd_consts *c, const float *patches, long n, float *desc128,
                                 float *raw238, int atan_mode, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    job_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){c, patches, desc128, raw238, n * t / nthreads, n * (t + 1) / nthreads,
                          atan_mode};
        if (nthreads == 1) worker(&jobs[t]);
        else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------- */
/* Keypoint mode: patch pyramid + rotated/scaled bilinear sampling            */
/* ------------------------------------------------------------------------- */

/* Sampler: linear filter, MirroredRepeat (vulkan/mod.rs:940-943).  Exact-f32
 * restatement of VK bilinear filtering at unnormalised coordinate u (texel
 * centres at i+0.5): texels floor(u-0.5), +1, weight frac(u-0.5).  The weight
 * precision of real samplers is implementation-defined. */
static int mirror(int i, int n)
{
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

static float tex_bilinear(const float *img, int w, int h, float u, float v)
{
    const float fu = u - 0.5f, fv = v - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const float ax = fu - x0f, ay = fv - y0f;
    const int x0 = mirror((int)x0f, w), x1 = mirror((int)x0f + 1, w);
    const int y0 = mirror((int)y0f, h), y1 = mirror((int)y0f + 1, h);
    const float t00 = img[y0 * w + x0], t10 = img[y0 * w + x1];
    const float t01 = img[y1 * w + x0], t11 = img[y1 * w + x1];
    const float top = t00 * (1.f - ax) + t10 * ax;
    const float bot = t01 * (1.f - ax) + t11 * ax;
    return top * (1.f - ay) + bot * ay;
}

/* shaders/blur.glsl:18-64: sigma 0.6 Gaussian as 3 bilinear fetches per pass,
 * H pass (layer0 -> layer1) then V pass (layer1 -> layer0);
 * dispatch vulkan/tasks_detect.rs:150-161 */
static void blur_sigma06(const float *in, int w, int h, float *tmp, float *out)
{
    const float W0 = 0.66381836f, W1 = 0.16809084f, OFF = 0.015267163f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float cx = (float)x + 0.5f, cy = (float)y + 0.5f, o = 1.f + OFF;
            float sum = tex_bilinear(in, w, h, cx, cy) * W0;
            sum += (tex_bilinear(in, w, h, cx - o, cy) + tex_bilinear(in, w, h, cx + o, cy)) * W1;
            tmp[y * w + x] = sum;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float cx = (float)x + 0.5f, cy = (float)y + 0.5f, o = 1.f + OFF;
            float sum = tex_bilinear(tmp, w, h, cx, cy) * W0;
            sum += (tex_bilinear(tmp, w, h, cx, cy - o) + tex_bilinear(tmp, w, h, cx, cy + o)) * W1;
            out[y * w + x] = sum;
        }
}

/* shaders/swt.glsl:18-57 with in_level=0: B3-spline [1,4,6,4,1]/16, dilation 1,
 * H (layer 0 -> scratch) then V (scratch -> layer 1).  Taps land on texel
 * centres, so the fetches are plain texel reads with mirrored addressing. */
static void swt_level0(const float *in, int w, int h, float *tmp, float *out)
{
    const float K0 = 6.f / 16.f, K1 = 4.f / 16.f, K2 = 1.f / 16.f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sum = in[y * w + x] * K0;
            sum += in[y * w + mirror(x - 2, w)] * K2;
            sum += in[y * w + mirror(x - 1, w)] * K1;
            sum += in[y * w + mirror(x + 1, w)] * K1;
            sum += in[y * w + mirror(x + 2, w)] * K2;
            tmp[y * w + x] = sum;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float sum = tmp[y * w + x] * K0;
            sum += tmp[mirror(y - 1, h) * w + x] * K1;
            sum += tmp[mirror(y - 2, h) * w + x] * K2;
            sum += tmp[mirror(y + 2, h) * w + x] * K2;
            sum += tmp[mirror(y + 1, h) * w + x] * K1;
            out[y * w + x] = sum;
        }
}

/* Level geometry: content of level l is (w >> l) x (h >> l)
 * (vulkan/patch_pyramid.rs:179-180, mip sizes vulkan/mod.rs:840-852 with
 * max_image == image, which every caller in the reference uses). */
int mkd_oracle_pyramid_levels(int w, int h)
{
    /* vulkan/mod.rs:372-373: ceil(log2(min(w,h))) */
    const float m = (float)(w < h ? w : h);
    return (int)roundf(ceilf(log2f(m)));
}

long mkd_oracle_pyramid_floats(int w, int h)
{
    long n = 0;
    const int L = mkd_oracle_pyramid_levels(w, h);
    for (int l = 0; l < L; l++) {
        const int lw = (w >> l) > 0 ? (w >> l) : 1, lh = (h >> l) > 0 ? (h >> l) : 1;
        n += (long)lw * lh;
    }
    return n;
}

/* Build the patch pyramid into `pyr` (levels packed back to back, level l is
 * (w>>l)x(h>>l) row-major).  vulkan/patch_pyramid.rs:38-156,232-289,
 * shaders/blur_pyramid.glsl:24-62. */
void mkd_oracle_build_pyramid(const float *img, int w, int h, float *pyr)
{
    const int L = mkd_oracle_pyramid_levels(w, h);
    float *tmp = (float *)malloc(sizeof(float) * w * h);
    float *coarse1 = (float *)malloc(sizeof(float) * w * h);
    float *lvl0 = pyr;
    blur_sigma06(img, w, h, tmp, lvl0); /* level 0 = coarse layer 0 (Nearest blit 1:1) */
    long off = (long)w * h;
    int pw = w, ph = h;
    const float *prev = lvl0;
    for (int l = 1; l < L; l++) {
        const int lw = (w >> l) > 0 ? (w >> l) : 1, lh = (h >> l) > 0 ? (h >> l) : 1;
        float *dst = pyr + off;
        if (l == 1) {
            /* coarse layer 1 = one a-trous pass over layer 0, then a Nearest blit
             * from [0,w)x[0,h) to [0,w/2)x[0,h/2): source texel floor((i+.5)*w/(w/2))
             * (patch_pyramid.rs:74-96,251-285) */
            swt_level0(lvl0, w, h, tmp, coarse1);
            const int dw = w / 2, dh = h / 2;
            for (int y = 0; y < lh; y++)
                for (int x = 0; x < lw; x++) {
                    int sx = (int)floorf(((float)x + 0.5f) * (float)w / (float)dw);
                    int sy = (int)floorf(((float)y + 0.5f) * (float)h / (float)dh);
                    if (sx > w - 1) sx = w - 1;
                    if (sy > h - 1) sy = h - 1;
                    dst[y * lw + x] = coarse1[sy * w + sx];
                }
        } else if (pw / 2 >= 1 && ph / 2 >= 1) {
            /* blur_pyramid.glsl: H pass at full resolution of level l-1 ... */
            const float W0 = 0.375f, W1 = 0.3125f, O = 1.2f;
            for (int y = 0; y < ph; y++)
                for (int x = 0; x < pw; x++) {
                    const float cx = (float)x + 0.5f, cy = (float)y + 0.5f;
                    const float s = tex_bilinear(prev, pw, ph, cx, cy) * W0 +
                                    (tex_bilinear(prev, pw, ph, cx - O, cy) +
                                     tex_bilinear(prev, pw, ph, cx + O, cy)) * W1;
                    tmp[y * pw + x] = s;
                }
            /* ... then V pass centred on texel (2x, 2y) of the H result */
            for (int y = 0; y < lh; y++)
                for (int x = 0; x < lw; x++) {
                    const float cx = 2.f * (float)x + 0.5f, cy = 2.f * (float)y + 0.5f;
                    const float s = tex_bilinear(tmp, pw, ph, cx, cy) * W0 +
                                    (tex_bilinear(tmp, pw, ph, cx, cy - O) +
                                     tex_bilinear(tmp, pw, ph, cx, cy + O)) * W1;
                    dst[y * lw + x] = s;
                }
        } else {
            for (long i = 0; i < (long)lw * lh; i++) dst[i] = prev[0];
        }
        prev = dst;
        pw = lw;
        ph = lh;
        off += (long)lw * lh;
    }
    free(tmp);
    free(coarse1);
}

/* shaders/mkd/patch_gradients.glsl:42-70: sample one 32x32 patch.
 * kp = {x, y, size, angle_deg}.  Level clamped to [0, L-1] (the detector never
 * emits sizes below 1.64). */
/* `contract` != 0: the second legitimate reading of patch_gradients.glsl:60-67 -- the shader is not `precise`, so its
 * compiler may fuse `dx*ca - dy*sa` and `xx*r + x/2^L` into fma (one rounding instead of two).  The two readings place a
 * sample up to one ulp of its coordinate apart (2.4e-4 texel for coordinates in [2048, 4096)): the whole of the thinnest
 * parity margin of the -m gpu run (profiles/r05_parity_report.txt).  The HIP sampler (csrc/mkd_sample.h) is this reading. */
static void sample_patch_reading(const float *pyr, int w, int h, const float *kp, float patch_scale_factor, float *patch,
                                 int contract);

void mkd_oracle_sample_patch(const float *pyr, int w, int h, const float *kp,
                             float patch_scale_factor, float *patch)
{
    sample_patch_reading(pyr, w, h, kp, patch_scale_factor, patch, 0);
}

static void sample_patch_reading(const float *pyr, int w, int h, const float *kp, float patch_scale_factor, float *patch,
                                 int contract)
{
    const int L = mkd_oracle_pyramid_levels(w, h);
    const float scale = kp[2] * patch_scale_factor / (float)PS;
    const float log2_scale = log2f(scale);
    float lvl = floorf(log2_scale);
    if (lvl < 0.f) lvl = 0.f;
    if (lvl > (float)(L - 1)) lvl = (float)(L - 1);
    const float rem_scale = exp2f(log2_scale - lvl);
    const int l = (int)lvl;
    long off = 0;
    for (int i = 0; i < l; i++) {
        const int lw = (w >> i) > 0 ? (w >> i) : 1, lh = (h >> i) > 0 ? (h >> i) : 1;
        off += (long)lw * lh;
    }
    const int lw = (w >> l) > 0 ? (w >> l) : 1, lh = (h >> l) > 0 ? (h >> l) : 1;
    const float *img = pyr + off;
    const float ang = kp[3] * (3.14159265358979323846f / 180.f); /* radians() */
    const float ca = cosf(ang), sa = sinf(ang);
    const float inv = 1.f / exp2f(lvl);
    for (int ly = 0; ly < PS; ly++)
        for (int lx = 0; lx < PS; lx++) {
            const float dx = (float)lx - 16.f, dy = (float)ly - 16.f;
            const float xx = contract ? fmaf(dx, ca, -dy * sa) : dx * ca - dy * sa;
            const float yy = contract ? fmaf(dx, sa, dy * ca) : dx * sa + dy * ca;
            const float sx = contract ? fmaf(xx, rem_scale, kp[0] * inv) : xx * rem_scale + kp[0] * inv;
            const float sy = contract ? fmaf(yy, rem_scale, kp[1] * inv) : yy * rem_scale + kp[1] * inv;
            /* textureLod at ((s+0.5)/size): unnormalised coordinate s+0.5 */
            patch[ly * PS + lx] = tex_bilinear(img, lw, lh, sx + 0.5f, sy + 0.5f);
        }
}

void mkd_oracle_sample_patches(const float *pyr, int w, int h, const float *kps /*[n][4]*/, long n,
                               float patch_scale_factor, float *patches)
{
    for (long i = 0; i < n; i++)
        mkd_oracle_sample_patch(pyr, w, h, kps + 4 * i, patch_scale_factor, patches + i * NPX);
}

void mkd_oracle_sample_patches_reading(const float *pyr, int w, int h, const float *kps /*[n][4]*/, long n,
                                       float patch_scale_factor, int contract, float *patches)
{
    for (long i = 0; i < n; i++)
        sample_patch_reading(pyr, w, h, kps + 4 * i, patch_scale_factor, patches + i * NPX, contract);
}

unsigned long mkd_oracle_sizeof_consts(void) { return sizeof(mkd_consts); }

/* ------------------------------------------------------------------------- */
/* Keypoint orientation: shaders/keypoint_orientation.glsl */
/* ------------------------------------------------------------------------- */

/* Coarse stack: layer 0 = sigma-0.6 blur of the input (shaders/blur.glsl), layer l+1 = B3-spline a-trous
 * pass with dilation 2^l over layer l (shaders/swt.glsl:24-58, H then V; vulkan/mod.rs:1093-1130).
 * stack: [n_layers][h][w]. */
void mkd_oracle_build_coarse_stack(const float *img, int w, int h, int n_layers, float *stack)
{
    float *tmp = (float *)malloc(sizeof(float) * w * h);
    blur_sigma06(img, w, h, tmp, stack);
    const float K0 = 6.f / 16.f, K1 = 4.f / 16.f, K2 = 1.f / 16.f;
    for (int l = 0; l + 1 < n_layers; l++) {
        const float *in = stack + (long)l * w * h;
        float *out = stack + (long)(l + 1) * w * h;
        const int d = 1 << l;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float sum = in[y * w + x] * K0;
                sum += in[y * w + mirror(x - 2 * d, w)] * K2;
                sum += in[y * w + mirror(x - d, w)] * K1;
                sum += in[y * w + mirror(x + d, w)] * K1;
                sum += in[y * w + mirror(x + 2 * d, w)] * K2;
                tmp[y * w + x] = sum;
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float sum = tmp[y * w + x] * K0;
                sum += tmp[mirror(y - d, h) * w + x] * K1;
                sum += tmp[mirror(y - 2 * d, h) * w + x] * K2;
                sum += tmp[mirror(y + 2 * d, h) * w + x] * K2;
                sum += tmp[mirror(y + d, h) * w + x] * K1;
                out[y * w + x] = sum;
            }
    }
    free(tmp);
}

/* One extremum {x, y, size}: angles (degrees, 360 - 10*bin) of every histogram peak >= 0.8 max, in ascending
 * bin order.  Returns the number of angles written (<= max_out).  keypoint_orientation.glsl:36-171.
 * Out-of-image loads return 0 (the shader's valid_px admits y == height; robust image access yields 0). */
int mkd_oracle_orient_one(const float *stack, int w, int h, int n_layers, const float *ex, float *angles,
                          int max_out)
{
    const float SIGMA_RADIUS = sqrtf(2.0f), FIRST = 0.82f, PI_F = 3.1415927f;
    const int R = 7, PS15 = 15, NB = 36;
    const int kx = (int)ex[0], ky = (int)ex[1];
    const float size = ex[2];
    int level = (int)roundf(log2f(size / (FIRST * SIGMA_RADIUS)));
    if (level < 0) level = 0;
    if (level > n_layers - 1) level = n_layers - 1;
    const int step = 1 << level;
    const int radius = (int)roundf(3.f * 1.5f * size / SIGMA_RADIUS);
    const float sigma = 1.5f * size / SIGMA_RADIUS;
    const float *img = stack + (long)level * w * h;
    float patch[15 * 15], weight[15 * 15];
    int bins[15 * 15], ingrad[15 * 15];
    for (int ly = 0; ly < PS15; ly++)
        for (int lx = 0; lx < PS15; lx++) {
            const int xd = (lx - R) * step, yd = (ly - R) * step;
            const int xi = kx + xd, yi = ky + yd;
            const int valid = 0 <= xi && xi < w && 0 <= yi && yi <= h;
            ingrad[ly * PS15 + lx] = valid && abs(xd) <= radius && abs(yd) <= radius;
            patch[ly * PS15 + lx] = (valid && yi < h) ? img[yi * w + xi] : 0.f;
        }
    for (int ly = 0; ly < PS15; ly++)
        for (int lx = 0; lx < PS15; lx++) {
            const int i = ly * PS15 + lx;
            bins[i] = NB + 1;
            weight[i] = 0.f;
            if (!ingrad[i] || lx == 0 || lx == PS15 - 1 || ly == 0 || ly == PS15 - 1) continue;
            const float gx = patch[i + 1] - patch[i - 1];
            const float gy = patch[i - PS15] - patch[i + PS15];
            if (gx == 0.f && gy == 0.f) continue;
            const float mag = sqrtf(gx * gx + gy * gy);
            const float fx = (float)(lx - R) * (float)step, fy = (float)(ly - R) * (float)step;
            const float dist = fx * fx + fy * fy;
            weight[i] = expf(-dist / (2.f * sigma * sigma)) * mag;
            const float ang = mkd_oracle_atan2_shader(gx, gy);
            const int rb = (int)roundf(ang * ((float)NB / (2.f * PI_F)));
            bins[i] = rb < 0 ? rb + NB : (rb >= NB ? rb - NB : rb);
        }
    float raw[36 + 4] = {0}, hist[36];
    for (int i = 0; i < PS15 * PS15; i++) /* one thread, row-major: lines 114-124 */
        if (bins[i] < NB) raw[2 + bins[i]] += weight[i];
    raw[1] = raw[NB + 1];
    raw[0] = raw[NB];
    raw[NB + 2] = raw[2];
    raw[NB + 3] = raw[3];
    float mx = 0.f;
    for (int b = 0; b < NB; b++) {
        const int rb = b + 2;
        hist[b] = (raw[rb - 2] + raw[rb + 2]) * (1.0f / 16.0f) + (raw[rb - 1] + raw[rb + 1]) * (4.0f / 16.0f) +
                  raw[rb] * (6.0f / 16.0f);
        if (hist[b] > mx) mx = hist[b];
    }
    const float thresh = mx * 0.8f;
    int n = 0;
    for (int b = 0; b < NB; b++) {
        const float hv = hist[b], left = hist[b > 0 ? b - 1 : NB - 1], right = hist[(b + 1) % NB];
        if (left < hv && right < hv && thresh <= hv) {
            const float interp = (left - right) / (left - 2.f * hv + right);
            const float rbin = (float)b + interp / 2.0f;
            const float bin = rbin < 0.f ? rbin + NB : (rbin > NB ? rbin - NB : rbin);
            if (n < max_out) angles[n] = 360.0f - (360.0f / (float)NB) * bin;
            n++;
        }
    }
    return n < max_out ? n : max_out;
}

/* extrema [n][4] (x, y, size, response) -> keypoints [..][5] (x, y, size, angle, response), ordered by
 * extremum then bin (the reference appends atomically, i.e. in no particular order).  Returns the count. */
long mkd_oracle_orient(const float *stack, int w, int h, int n_layers, const float *extrema, long n,
                       float *kps, long max_kps)
{
    long m = 0;
    for (long i = 0; i < n; i++) {
        float ang[36];
        const int c = mkd_oracle_orient_one(stack, w, h, n_layers, extrema + 4 * i, ang, 36);
        for (int j = 0; j < c && m < max_kps; j++, m++) {
            kps[5 * m + 0] = extrema[4 * i + 0];
            kps[5 * m + 1] = extrema[4 * i + 1];
            kps[5 * m + 2] = extrema[4 * i + 2];
            kps[5 * m + 3] = ang[j];
            kps[5 * m + 4] = extrema[4 * i + 3];
        }
    }
    return m;
}

/* ------------------------------------------------------------------------- */
/* Detector: DoG + 3-D extremum scan + refinement + edge test    */
/* shaders/swt_sub.glsl:17-30, shaders/scan_extrema.glsl:36-241,               */
/* dispatch vulkan/tasks_detect.rs:264-316, constants vulkan/mod.rs:76,395-407 */
/* ------------------------------------------------------------------------- */

/* fine[l] = coarse[l] - coarse[l+1], l = 0 .. n_layers-2 (swt_sub.glsl:24-29; texel-centre fetches) */
void mkd_oracle_dog(const float *stack, int w, int h, int n_layers, float *fine)
{
    const long px = (long)w * h;
    for (int l = 0; l + 1 < n_layers; l++)
        for (long i = 0; i < px; i++) fine[l * px + i] = stack[l * px + i] - stack[(l + 1) * px + i];
}

/* Extremum scan over the DoG volume fine[n_fine][h][w].  The shader works in 4x4x4 cubes (one workgroup each,
 * origin (border, border, 1 + skip_layers)), keeps at most 8 candidates per cube (max_wg_extrema, line 28; WHICH
 * 8 is left to the atomics) and appends survivors with a global atomic, i.e. in no order.  Order defined here:
 * cubes in raster order (z, y, x), candidates of a cube by local index x + 4 y + 16 z; a cube with more than
 * 8 candidates keeps the first 8 in that order.
 * extrema [max_out][4] = (x + offset_x, y + offset_y, size, contrast).  Returns the number written; *n_total
 * (may be NULL) receives the number found (reference: n_scanned_extrema, mod.rs:625). */
long mkd_oracle_scan_extrema(const float *fine, int w, int h, int n_fine, int border, int skip_layers,
                             float contrast_threshold, float *extrema, long max_out, long *n_total)
{
    const long px = (long)w * h;
    const int b1 = border > 1 ? border : 1;
    const int gx = (w - 2 * border + 3) / 4, gy = (h - 2 * border + 3) / 4;
    const int gz = (n_fine - 2 - skip_layers + 3) / 4;
    long found = 0;
    if (w - 2 * border <= 0 || h - 2 * border <= 0 || n_fine - 2 - skip_layers <= 0) {
        if (n_total) *n_total = 0;
        return 0;
    }
#define AT(z, y, x) fine[(long)(z) * px + (long)(y) * w + (x)]
    for (int cz = 0; cz < gz; cz++)
        for (int cy = 0; cy < gy; cy++)
            for (int cx = 0; cx < gx; cx++) {
                int cand = 0;
                for (int li = 0; li < 64 && cand < 8; li++) {
                    const int x = cx * 4 + (li & 3) + border, y = cy * 4 + ((li >> 2) & 3) + border;
                    const int z = cz * 4 + (li >> 4) + 1 + skip_layers;
                    /* is_extremum, lines 86-126 */
                    if (x < b1 || x >= w - b1 || y < b1 || y >= h - b1 || z <= 0 || z >= n_fine - 1) continue;
                    const float val = AT(z, y, x);
                    if (fabsf(val) <= contrast_threshold) continue;
                    const float sgn = glsl_sign(val);
                    int ok = 1;
                    for (int dz = -1; dz <= 1 && ok; dz++)
                        for (int dy = -1; dy <= 1 && ok; dy++)
                            for (int dx = -1; dx <= 1; dx++)
                                if ((dz || dy || dx) && !(sgn * val >= sgn * AT(z + dz, y + dy, x + dx))) {
                                    ok = 0;
                                    break;
                                }
                    if (!ok) continue;
                    cand++; /* occupies one of the cube's 8 slots whether or not it survives refinement */
                    /* lines 165-236 */
                    const float dds = (AT(z + 1, y, x) - AT(z - 1, y, x)) / 2.0f;
                    const float ddy = (AT(z, y + 1, x) - AT(z, y - 1, x)) / 2.0f;
                    const float ddx = (AT(z, y, x + 1) - AT(z, y, x - 1)) / 2.0f;
                    const float value2x = AT(z, y, x) * 2.0f;
                    const float h11 = AT(z + 1, y, x) + AT(z - 1, y, x) - value2x;
                    const float h22 = AT(z, y + 1, x) + AT(z, y - 1, x) - value2x;
                    const float h33 = AT(z, y, x + 1) + AT(z, y, x - 1) - value2x;
                    const float h12 = (AT(z + 1, y + 1, x) - AT(z - 1, y + 1, x) - AT(z + 1, y - 1, x) +
                                       AT(z - 1, y - 1, x)) / 4.0f;
                    const float h13 = (AT(z + 1, y, x + 1) - AT(z - 1, y, x + 1) - AT(z + 1, y, x - 1) +
                                       AT(z - 1, y, x - 1)) / 4.0f;
                    const float h23 = (AT(z, y + 1, x + 1) - AT(z, y + 1, x - 1) - AT(z, y - 1, x + 1) +
                                       AT(z, y - 1, x - 1)) / 4.0f;
                    const float det = h11 * h22 * h33 - h11 * h23 * h23 - h12 * h12 * h33 +
                                      2.f * h12 * h13 * h23 - h13 * h13 * h22;
                    const float hinv11 = (h22 * h33 - h23 * h23) / det;
                    const float hinv12 = (h13 * h23 - h12 * h33) / det;
                    const float hinv13 = (h12 * h23 - h13 * h22) / det;
                    const float hinv22 = (h11 * h33 - h13 * h13) / det;
                    const float hinv23 = (h12 * h13 - h11 * h23) / det;
                    const float hinv33 = (h11 * h22 - h12 * h12) / det;
                    const float os = -(hinv11 * dds + hinv12 * ddy + hinv13 * ddx);
                    const float oy = -(hinv12 * dds + hinv22 * ddy + hinv23 * ddx);
                    const float ox = -(hinv13 * dds + hinv23 * ddy + hinv33 * ddx);
                    /* an offset beyond half a cell moves x, y, z and emits nothing (lines 200-203); NaN offsets
                     * (det == 0) compare false and fall through to the else branch like in the shader */
                    if (fabsf(ox) > 0.5f || fabsf(oy) > 0.5f || fabsf(os) > 0.5f) continue;
                    if (x < border || w - border <= x || y < border || h - border <= y || z < 1 || n_fine - 1 <= z)
                        continue;
                    const float interp = os * dds + oy * ddy + ox * ddx;
                    const float contrast = fabsf(AT(z, y, x) + interp / 2.0f);
                    const float denom = (h22 + h33) * (h22 + h33);
                    if (denom == 0.f) continue;
                    const float cm = 1.f - 4.f * (h22 * h33 - h23 * h23) / denom;
                    if (0.7f <= cm && cm <= 1.5f) continue; /* edge-like: anisotropic hessian */
                    const float size = 0.82f * sqrtf(2.0f) * exp2f((float)z + os);
                    if (found < max_out) {
                        extrema[4 * found + 0] = (float)x + ox;
                        extrema[4 * found + 1] = (float)y + oy;
                        extrema[4 * found + 2] = size;
                        extrema[4 * found + 3] = contrast;
                    }
                    found++;
                }
            }
#undef AT
    if (n_total) *n_total = found;
    return found < max_out ? found : max_out;
}

/* TopKContrastFilter::filter, vulkan/mod.rs:1753-1786: blobs with size >= min_size; if more than n remain, those
 * whose contrast reaches the (n+1)-th largest, in index order, until n are taken.  (The reference indexes its
 * copy of the contrasts by the ORIGINAL index while that copy holds only the blobs that passed min_size; the
 * two coincide when min_size filters nothing, which is how its callers use it.  This restatement indexes by
 * position, the evident intent.)  Returns the number of indices written. */
static int cmp_desc(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? 1 : (x > y ? -1 : 0);
}

long mkd_oracle_topk_filter(const float *extrema, long n_in, long n_keep, float min_size, unsigned *indices)
{
    long m = 0;
    float *c = (float *)malloc(sizeof(float) * (n_in > 0 ? n_in : 1));
    unsigned *idx = (unsigned *)malloc(sizeof(unsigned) * (n_in > 0 ? n_in : 1));
    for (long i = 0; i < n_in; i++)
        if (extrema[4 * i + 2] >= min_size) {
            idx[m] = (unsigned)i;
            c[m++] = fabsf(extrema[4 * i + 3]);
        }
    long out = 0;
    if (m <= n_keep) {
        for (long i = 0; i < m; i++) indices[out++] = idx[i];
    } else {
        float *sorted = (float *)malloc(sizeof(float) * m);
        memcpy(sorted, c, sizeof(float) * m);
        qsort(sorted, m, sizeof(float), cmp_desc);
        const float cutoff = sorted[n_keep]; /* order_stat::kth(.., n): element n of the ascending -|c| */
        for (long i = 0; i < m && out < n_keep; i++)
            if (c[i] >= cutoff) indices[out++] = idx[i];
        free(sorted);
    }
    free(c);
    free(idx);
    return out;
}

/* ------------------------------------------------------------------------- */
/* Brute-force matcher: examples/match_images/src/main.rs:8-27   */
/* ------------------------------------------------------------------------- */

/* `(&vb * &va).sum()`: ndarray (0.16, a dependency, not in the tree) sums a contiguous f32 array with eight
 * running partial sums combined as ((p0+p4)+(p1+p5))+((p2+p6)+(p3+p7)), then the tail; 128 = 16 x 8, no tail. */
static float dot128(const float *a, const float *b)
{
    float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 128; i += 8)
        for (int j = 0; j < 8; j++) p[j] += a[i + j] * b[i + j];
    float acc = 0.f;
    acc += p[0] + p[4];
    acc += p[1] + p[5];
    acc += p[2] + p[6];
    acc += p[3] + p[7];
    return acc;
}

typedef struct {
    const float *a, *b;
    long nb, begin, end;
    const unsigned *excl_lo, *excl_hi;
    float ratio;
    int *match;
    float *best, *second;
} match_job_t;

static void *match_worker(void *arg)
{
    match_job_t *j = (match_job_t *)arg;
    for (long i = j->begin; i < j->end; i++) {
        /* stable ascending sort by similarity, take the last two (lines 15-21): the best is the HIGHEST index among
         * equal maxima, the second the next one down, possibly an equal value */
        float s1 = -INFINITY, s2 = -INFINITY;
        long i1 = -1;
        for (long k = 0; k < j->nb; k++) {
            if (j->excl_lo && (unsigned long)k >= j->excl_lo[i] && (unsigned long)k < j->excl_hi[i]) continue;
            const float s = dot128(j->a + i * 128, j->b + k * 128);
            if (s >= s1) {
                s2 = s1;
                s1 = s;
                i1 = k;
            } else if (s > s2) {
                s2 = s;
            }
        }
        /* line 22: sim[first] * 0.8 > sim[second]; ratio <= 0 (an extension of the ABI) skips the test */
        j->match[i] = (i1 >= 0 && (j->ratio <= 0.f || s1 * j->ratio > s2)) ? (int)i1 : -1;
        if (j->best) j->best[i] = s1;
        if (j->second) j->second[i] = s2;
    }
    return NULL;
}

/* a [na][128] against b [nb][128]: match[i] = index of a_i's best b if it passes Lowe's ratio test, else -1.
 * excl_lo/excl_hi (may be NULL): b indices [excl_lo[i], excl_hi[i]) are skipped for a_i -- the "cross-image" form
 * of BASELINE configs[3], where a descriptor is not matched against its own image; the reference has two images
 * and no such notion.  The reference needs nb >= 2 (it indexes idxs[len - 2]); fewer candidates give -1 here. */
void mkd_oracle_match(const float *a, long na, const float *b, long nb, float ratio, const unsigned *excl_lo,
                      const unsigned *excl_hi, int *match, float *best, float *second, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    match_job_t jobs[256];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (match_job_t){a, b, nb, na * t / nthreads, na * (t + 1) / nthreads, excl_lo, excl_hi, ratio,
                                match, best, second};
        if (nthreads == 1) match_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, match_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}
