/* ORACLE — test infrastructure only. Never linked into, imported by or executed
 * from the product path (learned-pmctf_amd/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * pm_math.h: the scalar arithmetic spec ("PM-F32") shared by every oracle
 * primitive.  All values are IEEE-754 binary32, round-to-nearest-even, no
 * contraction other than the fmaf() calls written here (compile with
 * -ffp-contract=off).  The HIP kernels restate exactly these operation
 * sequences, which is what makes HIP-vs-oracle comparisons bit-exact.
 *
 * The functions replace the libm/ATen transcendentals the reference calls:
 *   torch.tanh     pMCTF/layers/lifting_1d.py:39,42 ; pMCTF/layers/long_context.py:26,32
 *   torch.sigmoid  pMCTF/layers/long_context.py:24,25,31
 *   torch.log      pMCTF/entropy_models/entropy_models.py:271
 * They agree with libm to <= 2 ulp (checked in tests/test_oracle_math.py); the
 * reference's results differ from ours only by that rounding noise.
 */
#ifndef PM_MATH_H
#define PM_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float pm_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t pm_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x)-1 for the reduced argument and the power of two: returns q with
 * exp(x) = (1+q) * 2^n, |r| <= ln2/2.  Degree-7 Taylor/Horner in fmaf. */
static inline float pm_exp_core(float x, int *n_out) {
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    float nf = rintf(x * 1.44269504088896341f);
    float r = fmaf(nf, -0.693145751953125f, x);
    r = fmaf(nf, -1.42860682030941723212e-6f, r);
    float p = 1.9841270e-4f;                 /* 1/5040 */
    p = fmaf(p, r, 1.3888889e-3f);           /* 1/720  */
    p = fmaf(p, r, 8.3333333e-3f);           /* 1/120  */
    p = fmaf(p, r, 4.1666667e-2f);           /* 1/24   */
    p = fmaf(p, r, 1.6666667e-1f);           /* 1/6    */
    p = fmaf(p, r, 0.5f);
    float r2 = r * r;
    float q = fmaf(p, r2, r);                /* exp(r) - 1 */
    *n_out = (int)nf;
    return q;
}

static inline float pm_expf(float x) {
    int n;
    float q = pm_exp_core(x, &n);
    float s = pm_u2f((uint32_t)(n + 127) << 23);   /* 2^n, n in [-126,127] */
    return (q + 1.0f) * s;
}

/* tanh(x) = -em1/(em1+2), em1 = expm1(-2|x|) */
static inline float pm_tanhf(float x) {
    float a = fabsf(x);
    int n;
    float q = pm_exp_core(-2.0f * a, &n);
    float em1;
    if (n == 0) {
        em1 = q;
    } else {
        float s = pm_u2f((uint32_t)(n + 127) << 23);
        em1 = (q + 1.0f) * s - 1.0f;
    }
    float t = -em1 / (em1 + 2.0f);
    return copysignf(t, x);
}

static inline float pm_sigmoidf(float x) {
    return 1.0f / (1.0f + pm_expf(-x));
}

/* natural log for normal positive x (callers clamp to >= 1e-5) */
static inline float pm_logf(float x) {
    uint32_t u = pm_f2u(x);
    int e = (int)(u >> 23) - 127;
    float m = pm_u2f((u & 0x007fffffu) | 0x3f800000u);   /* [1,2) */
    if (m > 1.41421356237f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 0.2222222222f;                 /* 2/9 */
    p = fmaf(p, z, 0.2857142857f);           /* 2/7 */
    p = fmaf(p, z, 0.4f);                    /* 2/5 */
    p = fmaf(p, z, 0.6666666667f);           /* 2/3 */
    float lm = fmaf(p * z, s, 2.0f * s);     /* 2s + s*z*p */
    float ef = (float)e;
    float lo = fmaf(ef, 9.0580006145e-6f, lm);      /* ln2_lo */
    return fmaf(ef, 0.693138123f, lo);              /* ln2_hi */
}

#endif
