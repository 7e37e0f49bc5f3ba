/* hpt_oracle.c — CPU oracle: a plain-C restatement of pbrt-v2's SamplerRenderer + PathIntegrator
 * hot path.  TEST INFRASTRUCTURE ONLY (see hpt_oracle.h).  Nothing here is product code.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  Float expressions keep the reference's evaluation order and its
 * float/double promotions (e.g. Cross() in double, geometry.h:477-484; double literals in
 * comparisons) so that, compiled with the same gcc/-O2/libm as oracle/_ref, the MT_REPLAY mode
 * reproduces the reference binary's images (tests/test_oracle_pin.py).
 *
 * Compile: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile.oracle).
 */
#include "hpt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* pbrt.h:190-197: M_PI is redefined as a FLOAT literal; INFINITY is libm's +inf on Linux. */
#define PI_F 3.14159265358979323846f
#define INV_PI_F 0.31830988618379067154f
#define INV_TWOPI_F 0.15915494309189533577f
static const float OneMinusEpsilon = 0x1.fffffep-1; /* montecarlo.h:50 */

typedef struct { float x, y, z; } v3;
typedef struct { float c[3]; } rgb;

/* ---- geometry helpers (core/geometry.h) ------------------------------------------------ */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, float f) { return V(a.x * f, a.y * f, a.z * f); }      /* :87-89  */
static inline v3 vdiv(v3 a, float f) { float inv = 1.f / f; return V(a.x * inv, a.y * inv, a.z * inv); } /* :94-98 */
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  /* :463-466 */
static inline float absdot(v3 a, v3 b) { return fabsf(dot(a, b)); }
static inline v3 cross(v3 a, v3 b) {                                               /* :475-484 (double!) */
    double v1x = a.x, v1y = a.y, v1z = a.z, v2x = b.x, v2y = b.y, v2z = b.z;
    v3 r;
    r.x = (float)((v1y * v2z) - (v1z * v2y));
    r.y = (float)((v1z * v2x) - (v1x * v2z));
    r.z = (float)((v1x * v2y) - (v1y * v2x));
    return r;
}
static inline float vlen2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline float vlen(v3 a) { return sqrtf(vlen2(a)); }
static inline v3 normalize(v3 a) { return vdiv(a, vlen(a)); }                       /* :507 */
static inline float dist2(v3 a, v3 b) { return vlen2(vsub(a, b)); }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); } /* pbrt.h:203-207 */
static inline float maxf(float a, float b) { return a < b ? b : a; }  /* std::max(a,b) */
static inline float minf(float a, float b) { return b < a ? b : a; }  /* std::min(a,b) */
static void coordinate_system(v3 v1, v3 *v2, v3 *v3o) {                            /* :508-518 */
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float invLen = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = V(-v1.z * invLen, 0.f, v1.x * invLen);
    } else {
        float invLen = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = V(0.f, v1.z * invLen, -v1.y * invLen);
    }
    *v3o = cross(v1, *v2);
}
static inline float spherical_theta(v3 v) { return acosf(clampf(v.z, -1.f, 1.f)); } /* :640-642 */
static inline float spherical_phi(v3 v) {                                           /* :645-648 */
    float p = atan2f(v.y, v.x);
    return (p < 0.f) ? p + 2.f * PI_F : p;
}

/* Transform::operator() (core/transform.h:192-246); m row-major */
static inline v3 xf_point(const float *m, v3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1.) return V(xp, yp, zp);
    return vdiv(V(xp, yp, zp), wp);   /* Point::operator/ : inv multiply (geometry.h:190-193) */
}
static inline v3 xf_vec(const float *m, v3 v) {
    float x = v.x, y = v.y, z = v.z;
    return V(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z,
             m[8] * x + m[9] * y + m[10] * z);
}
static inline v3 xf_normal(const float *minv, v3 n) { /* transform.h:230-236: transpose of mInv */
    float x = n.x, y = n.y, z = n.z;
    return V(minv[0] * x + minv[4] * y + minv[8] * z, minv[1] * x + minv[5] * y + minv[9] * z,
             minv[2] * x + minv[6] * y + minv[10] * z);
}


/* ---- 4x4 matrices, quaternions, AnimatedTransform (core/transform.{h,cpp}, core/quaternion.cpp) ---- */
typedef struct { float m[16]; } mat4;
static mat4 m4_identity(void) { mat4 r; memset(&r, 0, sizeof(r)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
static mat4 m4_mul(const mat4 *a, const mat4 *b) { /* Matrix4x4::Mul transform.h:83-92 */
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[4 * i + j] = a->m[4 * i + 0] * b->m[0 + j] + a->m[4 * i + 1] * b->m[4 + j] +
                             a->m[4 * i + 2] * b->m[8 + j] + a->m[4 * i + 3] * b->m[12 + j];
    return r;
}
static mat4 m4_transpose(const mat4 *a) { mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[4 * i + j] = a->m[4 * j + i]; return r; }
static mat4 m4_inverse(const mat4 *in) { /* Inverse(Matrix4x4) transform.cpp:76-135 (Gauss-Jordan, full pivoting) */
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    float minv[4][4];
    memcpy(minv, in->m, 16 * sizeof(float));
    for (int i = 0; i < 4; i++) {
        int irow = -1, icol = -1;
        float big = 0.;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        if (fabsf(minv[j][k]) >= big) { big = (float)fabsf(minv[j][k]); irow = j; icol = k; }
                    }
                }
            }
        }
        ++ipiv[icol];
        if (irow != icol) for (int k = 0; k < 4; ++k) { float t = minv[irow][k]; minv[irow][k] = minv[icol][k]; minv[icol][k] = t; }
        indxr[i] = irow; indxc[i] = icol;
        float pivinv = 1.f / minv[icol][icol];
        minv[icol][icol] = 1.f;
        for (int j = 0; j < 4; j++) minv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                float save = minv[j][icol];
                minv[j][icol] = 0;
                for (int k = 0; k < 4; k++) minv[j][k] -= minv[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) { float t = minv[k][indxr[j]]; minv[k][indxr[j]] = minv[k][indxc[j]]; minv[k][indxc[j]] = t; }
    }
    mat4 r; memcpy(r.m, minv, 16 * sizeof(float));
    return r;
}
static int m4_is_identity(const mat4 *a) { /* Transform::IsIdentity transform.h:138-147 */
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (a->m[4 * i + j] != (i == j ? 1.f : 0.f)) return 0;
    return 1;
}
typedef struct { mat4 m, minv; } xform;
typedef struct { v3 v; float w; } quat;
static inline float qdot(quat a, quat b) { return dot(a.v, b.v) + a.w * b.w; }          /* quaternion.h:104-106 */
static inline quat qscale(quat q, float f) { quat r; r.v = vmul(q.v, f); r.w = q.w * f; return r; }
static inline quat qadd(quat a, quat b) { quat r; r.v = vadd(a.v, b.v); r.w = a.w + b.w; return r; }
static inline quat qsub(quat a, quat b) { quat r; r.v = vsub(a.v, b.v); r.w = a.w - b.w; return r; }
static inline quat qnormalize(quat q) { float f = sqrtf(qdot(q, q)); quat r; r.v = vdiv(q.v, f); r.w = q.w / f; return r; } /* :109-111; v /= f is inv-multiply, w /= f divides */
static quat slerp(float t, quat q1, quat q2) { /* quaternion.cpp:95-107 */
    float cosTheta = qdot(q1, q2);
    if (cosTheta > .9995f) return qnormalize(qadd(qscale(q1, 1.f - t), qscale(q2, t)));
    float theta = acosf(clampf(cosTheta, -1.f, 1.f));
    float thetap = theta * t;
    quat qperp = qnormalize(qsub(q2, qscale(q1, cosTheta)));
    return qadd(qscale(q1, cosf(thetap)), qscale(qperp, sinf(thetap)));
}
static xform quat_to_transform(quat q) { /* Quaternion::ToTransform quaternion.cpp:39-59 */
    float xx = q.v.x * q.v.x, yy = q.v.y * q.v.y, zz = q.v.z * q.v.z;
    float xy = q.v.x * q.v.y, xz = q.v.x * q.v.z, yz = q.v.y * q.v.z;
    float wx = q.v.x * q.w, wy = q.v.y * q.w, wz = q.v.z * q.w;
    mat4 m = m4_identity();
    m.m[0] = 1.f - 2.f * (yy + zz); m.m[1] = 2.f * (xy + wz);       m.m[2] = 2.f * (xz - wy);
    m.m[4] = 2.f * (xy - wz);       m.m[5] = 1.f - 2.f * (xx + zz); m.m[6] = 2.f * (yz + wx);
    m.m[8] = 2.f * (xz + wy);       m.m[9] = 2.f * (yz - wx);       m.m[10] = 1.f - 2.f * (xx + yy);
    xform r; r.m = m4_transpose(&m); r.minv = m;
    return r;
}
static xform xf_mul(const xform *a, const xform *b) { /* Transform::operator* transform.cpp:286-290 */
    xform r; r.m = m4_mul(&a->m, &b->m); r.minv = m4_mul(&b->minv, &a->minv);
    return r;
}
/* AnimatedTransform::Interpolate (core/transform.cpp:371-396) */
static xform anim_interpolate(const hpt_instance *in, float time) {
    xform r;
    if (!in->actually_animated || time <= in->start_time) { memcpy(r.m.m, in->w2p_m[0], 64); memcpy(r.minv.m, in->w2p_minv[0], 64); return r; }
    if (time >= in->end_time) { memcpy(r.m.m, in->w2p_m[1], 64); memcpy(r.minv.m, in->w2p_minv[1], 64); return r; }
    float dt = (time - in->start_time) / (in->end_time - in->start_time);
    v3 T0 = V(in->T[0][0], in->T[0][1], in->T[0][2]), T1 = V(in->T[1][0], in->T[1][1], in->T[1][2]);
    v3 trans = vadd(vmul(T0, 1.f - dt), vmul(T1, dt));
    quat q0 = {V(in->R[0][0], in->R[0][1], in->R[0][2]), in->R[0][3]}, q1 = {V(in->R[1][0], in->R[1][1], in->R[1][2]), in->R[1][3]};
    quat rotate = slerp(dt, q0, q1);
    mat4 scale = m4_identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            scale.m[4 * i + j] = (1.f - dt) * in->S[0][4 * i + j] + dt * in->S[1][4 * i + j];
    xform A; A.m = m4_identity(); A.minv = m4_identity();  /* Translate(trans) transform.cpp:145-155 */
    A.m.m[3] = trans.x; A.m.m[7] = trans.y; A.m.m[11] = trans.z;
    A.minv.m[3] = -trans.x; A.minv.m[7] = -trans.y; A.minv.m[11] = -trans.z;
    xform B = quat_to_transform(rotate);
    xform C; C.m = scale; C.minv = m4_inverse(&scale);      /* Transform(const Matrix4x4&) transform.h:111-113 */
    xform AB = xf_mul(&A, &B);
    return xf_mul(&AB, &C);
}

/* ---- Spectrum = RGBSpectrum (pbrt.h:156, core/spectrum.h) -------------------------------- */
static inline rgb S(float v) { rgb r = {{v, v, v}}; return r; }
static inline rgb sadd(rgb a, rgb b) { rgb r = {{a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]}}; return r; }
static inline rgb smul(rgb a, rgb b) { rgb r = {{a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2]}}; return r; }
static inline rgb sscale(rgb a, float f) { rgb r = {{a.c[0] * f, a.c[1] * f, a.c[2] * f}}; return r; }
static inline rgb sdivf(rgb a, float f) { rgb r = {{a.c[0] / f, a.c[1] / f, a.c[2] / f}}; return r; } /* :182-189 true division */
static inline int sblack(rgb a) { return a.c[0] == 0. && a.c[1] == 0. && a.c[2] == 0.; }  /* :205-209 */
static inline float sy(rgb a) { return 0.212671f * a.c[0] + 0.715160f * a.c[1] + 0.072169f * a.c[2]; } /* :424-427 */
static inline rgb sclamp0(rgb a) { rgb r = {{clampf(a.c[0], 0, INFINITY), clampf(a.c[1], 0, INFINITY), clampf(a.c[2], 0, INFINITY)}}; return r; }

/* ---- RNG: MT19937 (core/rng.cpp:43-107) ------------------------------------------------ */
#define MT_N 624
#define MT_M 397
typedef struct { uint32_t mt[MT_N]; int mti; } mt_rng;
static void mt_seed(mt_rng *r, uint32_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < MT_N; i++)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->mti = MT_N;
}
static uint32_t mt_uint(mt_rng *r) {
    static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
    uint32_t y;
    if (r->mti >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (r->mt[kk] & 0x80000000u) | (r->mt[kk + 1] & 0x7fffffffu);
            r->mt[kk] = r->mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (r->mt[kk] & 0x80000000u) | (r->mt[kk + 1] & 0x7fffffffu);
            r->mt[kk] = r->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (r->mt[MT_N - 1] & 0x80000000u) | (r->mt[0] & 0x7fffffffu);
        r->mt[MT_N - 1] = r->mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        r->mti = 0;
    }
    y = r->mt[r->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static float mt_float(mt_rng *r) { return (mt_uint(r) & 0xffffff) / (float)(1 << 24); } /* rng.cpp:59-65 */
void orc_mt_fill(uint32_t seed, uint32_t *out, int n) {
    mt_rng r; mt_seed(&r, seed);
    for (int i = 0; i < n; i++) out[i] = mt_uint(&r);
}

/* ---- (0,2)-sequences (core/montecarlo.h:274-327) ---------------------------------------- */
static float van_der_corput(uint32_t n, uint32_t scramble) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    n ^= scramble;
    return minf(((n >> 8) & 0xffffff) / (float)(1 << 24), OneMinusEpsilon);
}
static float sobol2(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 0x1) scramble ^= v;
    return minf(((scramble >> 8) & 0xffffff) / (float)(1 << 24), OneMinusEpsilon);
}
static void shuffle_f(float *samp, uint32_t count, uint32_t dims, mt_rng *rng) { /* montecarlo.h:174-181 */
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t other = i + (mt_uint(rng) % (count - i));
        for (uint32_t j = 0; j < dims; ++j) {
            float t = samp[dims * i + j]; samp[dims * i + j] = samp[dims * other + j]; samp[dims * other + j] = t;
        }
    }
}
static void ld_shuffle_scrambled_1d(int nSamples, int nPixel, float *samples, mt_rng *rng) { /* :307-315 */
    uint32_t scramble = mt_uint(rng);
    for (int i = 0; i < nSamples * nPixel; ++i) samples[i] = van_der_corput(i, scramble);
    for (int i = 0; i < nPixel; ++i) shuffle_f(samples + i * nSamples, nSamples, 1, rng);
    shuffle_f(samples, nPixel, nSamples, rng);
}
static void ld_shuffle_scrambled_2d(int nSamples, int nPixel, float *samples, mt_rng *rng) { /* :318-326 */
    uint32_t scramble[2];
    scramble[0] = mt_uint(rng); scramble[1] = mt_uint(rng);
    for (int i = 0; i < nSamples * nPixel; ++i) {
        samples[2 * i] = van_der_corput(i, scramble[0]);
        samples[2 * i + 1] = sobol2(i, scramble[1]);
    }
    for (int i = 0; i < nPixel; ++i) shuffle_f(samples + 2 * i * nSamples, nSamples, 2, rng);
    shuffle_f(samples, nPixel, 2 * nSamples, rng);
}

/* Sample layout of PathIntegrator::RequestSamples (integrators/path.cpp:41-49) followed by
 * EmissionIntegrator::RequestSamples (integrators/emission.cpp:41-42):
 *   1D arrays (count 1 each): per depth i<3: [4i]=light component, [4i+1]=light number,
 *                             [4i+2]=bsdf component, [4i+3]=path component; [12],[13]=emission
 *   2D arrays: per depth i<3: [3i]=light position, [3i+1]=bsdf direction, [3i+2]=path direction */
#define N1D_PATH 12
#define N1D_ALL 14
#define N2D 9
#define SAMPLE_FLOATS 35 /* 5 camera + 12 + 18 */
typedef struct {
    float imageX, imageY, lensU, lensV, time;
    float oneD[N1D_ALL];
    float twoD[N2D][2];
} cam_sample;

/* LDPixelSample (core/montecarlo.cpp:200-252): consumes the tile RNG exactly as the reference. */
static void ld_pixel_sample_mt(int xPos, int yPos, float shutterOpen, float shutterClose,
                               int n, cam_sample *samples, float *buf, mt_rng *rng) {
    float *imageSamples = buf; buf += 2 * n;
    float *lensSamples = buf; buf += 2 * n;
    float *timeSamples = buf; buf += n;
    float *oneD[N1D_ALL], *twoD[N2D];
    for (int i = 0; i < N1D_ALL; ++i) { oneD[i] = buf; buf += n; }
    for (int i = 0; i < N2D; ++i) { twoD[i] = buf; buf += 2 * n; }
    ld_shuffle_scrambled_2d(1, n, imageSamples, rng);
    ld_shuffle_scrambled_2d(1, n, lensSamples, rng);
    ld_shuffle_scrambled_1d(1, n, timeSamples, rng);
    for (int i = 0; i < N1D_ALL; ++i) ld_shuffle_scrambled_1d(1, n, oneD[i], rng);
    for (int i = 0; i < N2D; ++i) ld_shuffle_scrambled_2d(1, n, twoD[i], rng);
    for (int i = 0; i < n; ++i) {
        samples[i].imageX = xPos + imageSamples[2 * i];
        samples[i].imageY = yPos + imageSamples[2 * i + 1];
        samples[i].time = (1.f - timeSamples[i]) * shutterOpen + timeSamples[i] * shutterClose; /* Lerp pbrt.h:198 */
        samples[i].lensU = lensSamples[2 * i];
        samples[i].lensV = lensSamples[2 * i + 1];
        for (int j = 0; j < N1D_ALL; ++j) samples[i].oneD[j] = oneD[j][i];
        for (int j = 0; j < N2D; ++j) { samples[i].twoD[j][0] = twoD[j][2 * i]; samples[i].twoD[j][1] = twoD[j][2 * i + 1]; }
    }
}

/* ---- HPT_SAMPLER_LD_HASH: the production sampler's definition --------
 * Same (0,2)-sequence structure as LDPixelSample; the scramble words and the within-pixel
 * sample permutations come from a stateless hash of (pixel index, seed) instead of the tile's
 * serial MT19937 stream, so sample i of any pixel is O(1) computable. */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
static inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = fmix32(a + 0x9e3779b9u);
    h = fmix32(h ^ (b + 0x85ebca6bu));
    h = fmix32(h ^ (c + 0xc2b2ae35u));
    return h;
}
/* bijection on [0, 2^m): w = 2^m - 1 */
static inline uint32_t perm_pow2(uint32_t i, uint32_t w, uint32_t key) {
    uint32_t x = i & w;
    x ^= key & w;          x = (x * 0xe170893du) & w;          x ^= x >> 4;
    x ^= (key >> 8) & w;   x = (x * 0x0929eb3fu) & w;          x ^= x >> 2;
    x ^= (key >> 16) & w;  x = (x * ((key >> 3) | 1u)) & w;    x ^= x >> 3;
    x = (x + (key >> 24)) & w;
    return x;
}
/* array ids: 0=image 1=lens 2=time 3..14=oneD[0..11] 15..23=twoD[0..8]; scramble word ids:
 * image 0,1 lens 2,3 time 4 oneD[j] 5+j twoD[j] 17+2j, 18+2j */
static void ld_hash_sample(uint32_t pixelIndex, uint32_t seed, int xPos, int yPos, float shutterOpen,
                           float shutterClose, uint32_t spp, uint32_t i, cam_sample *s) {
    uint32_t pk = hash3(pixelIndex, seed, 0x50495845u);
    uint32_t w = spp - 1;
#define SCR(k) hash3(pk, (uint32_t)(k), 1u)
#define PERM(a) perm_pow2(i, w, hash3(pk, (uint32_t)(a), 2u))
    uint32_t n;
    n = PERM(0); s->imageX = xPos + van_der_corput(n, SCR(0)); s->imageY = yPos + sobol2(n, SCR(1));
    n = PERM(1); s->lensU = van_der_corput(n, SCR(2)); s->lensV = sobol2(n, SCR(3));
    n = PERM(2); { float t = van_der_corput(n, SCR(4)); s->time = (1.f - t) * shutterOpen + t * shutterClose; }
    for (int j = 0; j < N1D_PATH; ++j) { n = PERM(3 + j); s->oneD[j] = van_der_corput(n, SCR(5 + j)); }
    s->oneD[12] = s->oneD[13] = 0.f;
    for (int j = 0; j < N2D; ++j) {
        n = PERM(15 + j);
        s->twoD[j][0] = van_der_corput(n, SCR(17 + 2 * j));
        s->twoD[j][1] = sobol2(n, SCR(18 + 2 * j));
    }
#undef SCR
#undef PERM
}
/* draws for bounces >= SAMPLE_DEPTH and Russian roulette, LD_HASH mode */
typedef struct { int mode; mt_rng *mt; uint32_t key; uint32_t counter; } draw_src;
static inline float draw_float(draw_src *d) {
    if (d->mode == HPT_SAMPLER_MT_REPLAY) return mt_float(d->mt);
    uint32_t h = fmix32(d->key + 0x9e3779b9u * (d->counter++));
    return (h & 0xffffff) / (float)(1 << 24);
}

/* ---- DirectLightingIntegrator sample layout (integrators/directlighting.cpp:54-77) --------------------
 * Strategy "all": per light i, LightSampleOffsets(n_i) then BSDFSampleOffsets(n_i), n_i = LDSampler::RoundSize
 * (RoundUpPow2) of Light::nSamples — each adds a 1D component array and a 2D array (core/light.cpp:64-68,
 * core/reflection.cpp:502-506).  Strategy "one": light (1), light number (1), bsdf (1).  Then
 * EmissionIntegrator::RequestSamples' two 1D arrays (integrators/emission.cpp:41-42), which the MT stream pays for.
 * A pixel sample is a vector of `fps` floats: 1D array j element k at o1[j] + k, 2D array j at o2[j] + 2k. */
typedef struct { int integrator, nl, n1d, n2d, fps; int *ns, *c1, *c2, *o1, *o2; } dl_layout;
static uint32_t round_up_pow2(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; } /* pbrt.h:264-271 */
static void dl_layout_init(const hpt_scene_desc *d, int integrator, dl_layout *L) {
    L->integrator = integrator; L->nl = d->n_lights;
    int all = integrator == HPT_INTEGRATOR_DIRECT_ALL;
    int groups = all ? d->n_lights : 1;
    L->n1d = (all ? 2 * groups : 3) + 2; L->n2d = 2 * groups;
    L->ns = (int *)calloc((size_t)groups + 1, sizeof(int));
    L->c1 = (int *)calloc((size_t)L->n1d + 1, sizeof(int)); L->o1 = (int *)calloc((size_t)L->n1d + 1, sizeof(int));
    L->c2 = (int *)calloc((size_t)L->n2d + 1, sizeof(int)); L->o2 = (int *)calloc((size_t)L->n2d + 1, sizeof(int));
    int k1 = 0, k2 = 0;
    for (int i = 0; i < groups; ++i) {
        int n = 1;
        if (all) { int ns = d->lights[i].nsamples; if (ns < 1) ns = 1; n = (int)round_up_pow2((uint32_t)ns); } /* Light ctor: max(1, ns) */
        L->ns[i] = n;
        L->c1[k1++] = n; L->c2[k2++] = n;          /* light component, light position */
        if (!all) L->c1[k1++] = 1;                /* light number */
        L->c1[k1++] = n; L->c2[k2++] = n;          /* bsdf component, bsdf direction */
    }
    L->c1[k1++] = 1; L->c1[k1++] = 1;              /* emission integrator */
    int off = 0;
    for (int j = 0; j < L->n1d; ++j) { L->o1[j] = off; off += L->c1[j]; }
    for (int j = 0; j < L->n2d; ++j) { L->o2[j] = off; off += 2 * L->c2[j]; }
    L->fps = off;
}
static void dl_layout_free(dl_layout *L) { free(L->ns); free(L->c1); free(L->c2); free(L->o1); free(L->o2); }
/* floats LDPixelSample needs for one pixel (LDSampler ctor, samplers/lowdiscrepancy.cpp:52-58) */
static size_t dl_buf_floats(const dl_layout *L, int spp) { return (size_t)spp * (size_t)(5 + L->fps); }

/* LDPixelSample (core/montecarlo.cpp:200-252) for an arbitrary array layout; vals: spp x fps */
static void ld_pixel_sample_mt_dl(const dl_layout *L, int xPos, int yPos, float shutterOpen, float shutterClose,
                                  int n, cam_sample *cams, float *vals, float *buf, mt_rng *rng) {
    float *imageSamples = buf; buf += 2 * n;
    float *lensSamples = buf; buf += 2 * n;
    float *timeSamples = buf; buf += n;
    float *arrays = buf;
    ld_shuffle_scrambled_2d(1, n, imageSamples, rng);
    ld_shuffle_scrambled_2d(1, n, lensSamples, rng);
    ld_shuffle_scrambled_1d(1, n, timeSamples, rng);
    float *b = arrays;
    for (int j = 0; j < L->n1d; ++j) { ld_shuffle_scrambled_1d(L->c1[j], n, b, rng); b += L->c1[j] * n; }
    for (int j = 0; j < L->n2d; ++j) { ld_shuffle_scrambled_2d(L->c2[j], n, b, rng); b += 2 * L->c2[j] * n; }
    for (int i = 0; i < n; ++i) {
        cams[i].imageX = xPos + imageSamples[2 * i];
        cams[i].imageY = yPos + imageSamples[2 * i + 1];
        cams[i].time = (1.f - timeSamples[i]) * shutterOpen + timeSamples[i] * shutterClose;
        cams[i].lensU = lensSamples[2 * i];
        cams[i].lensV = lensSamples[2 * i + 1];
        float *v = vals + (size_t)i * L->fps;
        b = arrays;
        for (int j = 0; j < L->n1d; ++j) { for (int k = 0; k < L->c1[j]; ++k) v[L->o1[j] + k] = b[L->c1[j] * i + k]; b += L->c1[j] * n; }
        for (int j = 0; j < L->n2d; ++j) { for (int k = 0; k < 2 * L->c2[j]; ++k) v[L->o2[j] + k] = b[2 * L->c2[j] * i + k]; b += 2 * L->c2[j] * n; }
    }
}
/* HPT_SAMPLER_LD_HASH for the same layout.  An array of count c is, as in LDShuffleScrambled*D, one scrambled
 * (0,2)-sequence of spp * c points cut into spp blocks of c: pixel sample i owns block PERM_a(i) and visits its c
 * points in a keyed order that differs per pixel sample.  Array ids: 1D array j -> 3 + j, 2D array j -> 3 + n1d + j;
 * scramble words: 1D j -> 5 + j, 2D j -> 5 + n1d + 2j, +1. */
static void ld_hash_sample_dl(const dl_layout *L, uint32_t pixelIndex, uint32_t seed, int xPos, int yPos, float shutterOpen,
                              float shutterClose, uint32_t spp, uint32_t i, cam_sample *s, float *v) {
    uint32_t pk = hash3(pixelIndex, seed, 0x50495845u);
    uint32_t w = spp - 1;
#define SCR(k) hash3(pk, (uint32_t)(k), 1u)
#define PERM(a) perm_pow2(i, w, hash3(pk, (uint32_t)(a), 2u))
#define WITHIN(a, k, c) perm_pow2((uint32_t)(k), (uint32_t)(c) - 1u, hash3(hash3(pk, (uint32_t)(a), 4u), i, 5u))
    uint32_t n;
    n = PERM(0); s->imageX = xPos + van_der_corput(n, SCR(0)); s->imageY = yPos + sobol2(n, SCR(1));
    n = PERM(1); s->lensU = van_der_corput(n, SCR(2)); s->lensV = sobol2(n, SCR(3));
    n = PERM(2); { float t = van_der_corput(n, SCR(4)); s->time = (1.f - t) * shutterOpen + t * shutterClose; }
    for (int j = 0; j < L->n1d - 2; ++j) {                 /* the emission integrator's arrays are not generated */
        uint32_t c = (uint32_t)L->c1[j], blk = PERM(3 + j);
        for (uint32_t k = 0; k < c; ++k) v[L->o1[j] + (int)k] = van_der_corput(blk * c + WITHIN(3 + j, k, c), SCR(5 + j));
    }
    v[L->o1[L->n1d - 2]] = v[L->o1[L->n1d - 1]] = 0.f;
    for (int j = 0; j < L->n2d; ++j) {
        uint32_t c = (uint32_t)L->c2[j], blk = PERM(3 + L->n1d + j);
        for (uint32_t k = 0; k < c; ++k) {
            uint32_t q = blk * c + WITHIN(3 + L->n1d + j, k, c);
            v[L->o2[j] + 2 * (int)k] = van_der_corput(q, SCR(5 + L->n1d + 2 * j));
            v[L->o2[j] + 2 * (int)k + 1] = sobol2(q, SCR(6 + L->n1d + 2 * j));
        }
    }
#undef SCR
#undef PERM
#undef WITHIN
}

/* ---- scene ------------------------------------------------------------------------------ */
typedef struct { v3 pmin, pmax; } bbox;
static inline bbox bbox_empty(void) { bbox b; b.pmin = V(INFINITY, INFINITY, INFINITY); b.pmax = V(-INFINITY, -INFINITY, -INFINITY); return b; }
static inline bbox bbox_union_p(bbox b, v3 p) {
    b.pmin = V(minf(b.pmin.x, p.x), minf(b.pmin.y, p.y), minf(b.pmin.z, p.z));
    b.pmax = V(maxf(b.pmax.x, p.x), maxf(b.pmax.y, p.y), maxf(b.pmax.z, p.z));
    return b;
}
static inline bbox bbox_union(bbox a, bbox b) { return bbox_union_p(bbox_union_p(a, b.pmin), b.pmax); }
static inline float bbox_area(bbox b) { v3 d = vsub(b.pmax, b.pmin); return 2.f * (d.x * d.y + d.x * d.z + d.y * d.z); }
static inline int bbox_maxext(bbox b) { v3 d = vsub(b.pmax, b.pmin); if (d.x > d.y && d.x > d.z) return 0; else if (d.y > d.z) return 1; else return 2; }
static inline float v3c(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

typedef struct { /* LinearBVHNode, accelerators/bvh.cpp:113-123 */
    bbox bounds;
    uint32_t offset; /* primitivesOffset | secondChildOffset */
    uint8_t nprims, axis, pad[2];
} lnode;

typedef struct { int32_t *ordered; lnode *nodes; int64_t nnodes; } bvh_t;
struct orc_scene {
    hpt_scene_desc d; /* deep copy */
    int64_t ntris;            /* prim ids: [0,ntris) triangles of ALL meshes, then quadrics, then instances */
    int64_t *mesh_base;       /* first global prim id of each mesh */
    int32_t *prim_mesh;       /* per triangle prim: mesh */
    int64_t nprims;           /* ntris + n_quadrics + n_instances */
    bvh_t top;                /* Scene::aggregate */
    bvh_t *inst;              /* TransformedPrimitive::primitive of each instance */
};

typedef struct { v3 o, d; float mint, maxt, time; int depth; } ray_t;
static inline v3 ray_at(const ray_t *r, float t) { return vadd(r->o, vmul(r->d, t)); } /* geometry.h:333 */

/* DifferentialGeometry subset (core/diffgeom.cpp:40-55) */
typedef struct { v3 p, nn, dpdu, dpdv; float u, v; } dgeom;
static void dg_init(dgeom *dg, v3 P, v3 dpdu, v3 dpdv, float u, float v, int flip) {
    dg->p = P; dg->dpdu = dpdu; dg->dpdv = dpdv;
    dg->nn = normalize(cross(dpdu, dpdv));
    dg->u = u; dg->v = v;
    if (flip) dg->nn = vmul(dg->nn, -1.f);
}

typedef struct { dgeom dg; int64_t prim; float rayEpsilon; float t, b1, b2; int inst; mat4 w2p_m; } isect_t;

static void tri_verts(const orc_scene *s, int64_t prim, const hpt_mesh **mo, int *vi, v3 *p1, v3 *p2, v3 *p3) {
    int m = s->prim_mesh[prim];
    const hpt_mesh *me = &s->d.meshes[m];
    int64_t tri = prim - s->mesh_base[m];
    const int32_t *idx = s->d.ipool + me->idx_off + 3 * tri;
    const float *P = s->d.fpool + me->p_off;
    vi[0] = idx[0]; vi[1] = idx[1]; vi[2] = idx[2];
    *p1 = V(P[3 * vi[0]], P[3 * vi[0] + 1], P[3 * vi[0] + 2]);
    *p2 = V(P[3 * vi[1]], P[3 * vi[1] + 1], P[3 * vi[1] + 2]);
    *p3 = V(P[3 * vi[2]], P[3 * vi[2] + 1], P[3 * vi[2] + 2]);
    *mo = me;
}
static void tri_uvs(const orc_scene *s, const hpt_mesh *me, const int *vi, float uv[3][2]) { /* trianglemesh.h:86-100 */
    if (me->uv_off >= 0) {
        const float *U = s->d.fpool + me->uv_off;
        for (int k = 0; k < 3; ++k) { uv[k][0] = U[2 * vi[k]]; uv[k][1] = U[2 * vi[k] + 1]; }
    } else {
        uv[0][0] = 0.; uv[0][1] = 0.; uv[1][0] = 1.; uv[1][1] = 0.; uv[2][0] = 1.; uv[2][1] = 1.;
    }
}

/* Triangle::Intersect / IntersectP (shapes/trianglemesh.cpp:127-208, 211-281).
 * want_dg: build the DifferentialGeometry (the reference does so on every accepted hit). */
static int tri_intersect(const orc_scene *s, int64_t prim, const ray_t *ray, isect_t *is, int want_dg, uint64_t *st) {
    const hpt_mesh *me; int vi[3]; v3 p1, p2, p3;
    tri_verts(s, prim, &me, vi, &p1, &p2, &p3);
    if (st) st[4]++;
    v3 e1 = vsub(p2, p1), e2 = vsub(p3, p1);
    v3 s1 = cross(ray->d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.) return 0;
    float invDivisor = 1.f / divisor;
    v3 sv = vsub(ray->o, p1);
    float b1 = dot(sv, s1) * invDivisor;
    if (b1 < 0. || b1 > 1.) return 0;
    v3 s2 = cross(sv, e1);
    float b2 = dot(ray->d, s2) * invDivisor;
    if (b2 < 0. || b1 + b2 > 1.) return 0;
    float t = dot(e2, s2) * invDivisor;
    if (t < ray->mint || t > ray->maxt) return 0;
    if (!want_dg) return 1;
    float uvs[3][2];
    tri_uvs(s, me, vi, uvs);
    float du1 = uvs[0][0] - uvs[2][0], du2 = uvs[1][0] - uvs[2][0];
    float dv1 = uvs[0][1] - uvs[2][1], dv2 = uvs[1][1] - uvs[2][1];
    v3 dp1 = vsub(p1, p3), dp2 = vsub(p2, p3);
    float determinant = du1 * dv2 - dv1 * du2;
    v3 dpdu, dpdv;
    if (determinant == 0.f)
        coordinate_system(normalize(cross(e2, e1)), &dpdu, &dpdv);
    else {
        float invdet = 1.f / determinant;
        dpdu = vmul(vsub(vmul(dp1, dv2), vmul(dp2, dv1)), invdet);
        dpdv = vmul(vadd(vmul(dp1, -du2), vmul(dp2, du1)), invdet);
    }
    float b0 = 1 - b1 - b2;
    float tu = b0 * uvs[0][0] + b1 * uvs[1][0] + b2 * uvs[2][0];
    float tv = b0 * uvs[0][1] + b1 * uvs[1][1] + b2 * uvs[2][1];
    dg_init(&is->dg, ray_at(ray, t), dpdu, dpdv, tu, tv, me->reverse_orientation ^ me->swaps_handedness);
    is->prim = prim; is->t = t; is->b1 = b1; is->b2 = b2;
    is->rayEpsilon = 1e-3f * t;
    return 1;
}

/* Quadratic (core/pbrt.h:309-323) */
static int quadratic(float A, float B, float C, float *t0, float *t1) {
    float discrim = B * B - 4.f * A * C;
    if (discrim < 0.) return 0;
    float rootDiscrim = sqrtf(discrim);
    float q;
    if (B < 0) q = -.5f * (B - rootDiscrim);
    else q = -.5f * (B + rootDiscrim);
    *t0 = q / A;
    *t1 = C / q;
    if (*t0 > *t1) { float tmp = *t0; *t0 = *t1; *t1 = tmp; }
    return 1;
}

/* Sphere::Intersect (shapes/sphere.cpp:58-157) / Disk::Intersect (shapes/disk.cpp:56-102).
 * The ray is given in WORLD space and transformed by WorldToObject as the reference does. */
static int quadric_intersect(const hpt_quadric *q, const ray_t *r, float *tHit, float *rayEps, dgeom *dg) {
    ray_t ray = *r;
    ray.o = xf_point(q->o2w_inv, r->o);   /* WorldToObject->m == ObjectToWorld->mInv */
    ray.d = xf_vec(q->o2w_inv, r->d);
    int flip = q->reverse_orientation ^ q->swaps_handedness;
    if (q->kind == HPT_QUADRIC_SPHERE) {
        float radius = q->radius, phiMax = q->phi_max, zmin = q->zmin, zmax = q->zmax;
        float A = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
        float B = 2 * (ray.d.x * ray.o.x + ray.d.y * ray.o.y + ray.d.z * ray.o.z);
        float C = ray.o.x * ray.o.x + ray.o.y * ray.o.y + ray.o.z * ray.o.z - radius * radius;
        float t0, t1;
        if (!quadratic(A, B, C, &t0, &t1)) return 0;
        if (t0 > ray.maxt || t1 < ray.mint) return 0;
        float thit = t0;
        if (t0 < ray.mint) { thit = t1; if (thit > ray.maxt) return 0; }
        v3 phit = ray_at(&ray, thit);
        if (phit.x == 0.f && phit.y == 0.f) phit.x = 1e-5f * radius;
        float phi = atan2f(phit.y, phit.x);
        if (phi < 0.) phi += 2.f * PI_F;
        if ((zmin > -radius && phit.z < zmin) || (zmax < radius && phit.z > zmax) || phi > phiMax) {
            if (thit == t1) return 0;
            if (t1 > ray.maxt) return 0;
            thit = t1;
            phit = ray_at(&ray, thit);
            if (phit.x == 0.f && phit.y == 0.f) phit.x = 1e-5f * radius;
            phi = atan2f(phit.y, phit.x);
            if (phi < 0.) phi += 2.f * PI_F;
            if ((zmin > -radius && phit.z < zmin) || (zmax < radius && phit.z > zmax) || phi > phiMax) return 0;
        }
        if (dg) {
            float u = phi / phiMax;
            float theta = acosf(clampf(phit.z / radius, -1.f, 1.f));
            float v = (theta - q->theta_min) / (q->theta_max - q->theta_min);
            float zradius = sqrtf(phit.x * phit.x + phit.y * phit.y);
            float invzradius = 1.f / zradius;
            float cosphi = phit.x * invzradius, sinphi = phit.y * invzradius;
            v3 dpdu = V(-phiMax * phit.y, phiMax * phit.x, 0);
            v3 dpdv = vmul(V(phit.z * cosphi, phit.z * sinphi, -radius * sinf(theta)), (q->theta_max - q->theta_min));
            dg_init(dg, xf_point(q->o2w, phit), xf_vec(q->o2w, dpdu), xf_vec(q->o2w, dpdv), u, v, flip);
        }
        *tHit = thit;
        *rayEps = 5e-4f * thit;
        return 1;
    } else { /* disk */
        if (fabsf(ray.d.z) < 1e-7) return 0;
        float thit = (q->height - ray.o.z) / ray.d.z;
        if (thit < ray.mint || thit > ray.maxt) return 0;
        v3 phit = ray_at(&ray, thit);
        float d2 = phit.x * phit.x + phit.y * phit.y;
        if (d2 > q->radius * q->radius || d2 < q->inner_radius * q->inner_radius) return 0;
        float phi = atan2f(phit.y, phit.x);
        if (phi < 0) phi += 2. * PI_F;
        if (phi > q->phi_max) return 0;
        if (dg) {
            float u = phi / q->phi_max;
            float R = sqrtf(d2);
            float oneMinusV = ((R - q->inner_radius) / (q->radius - q->inner_radius));
            float v = 1.f - oneMinusV;
            v3 dpdu = V(-q->phi_max * phit.y, q->phi_max * phit.x, 0.);
            v3 dpdv = V(phit.x, phit.y, 0.);
            dpdv = vmul(dpdv, (q->radius - q->inner_radius) / R);
            dg_init(dg, xf_point(q->o2w, phit), xf_vec(q->o2w, dpdu), xf_vec(q->o2w, dpdv), u, v, flip);
        }
        *tHit = thit;
        *rayEps = 5e-4f * thit;
        return 1;
    }
}
static float quadric_area(const hpt_quadric *q) {
    if (q->kind == HPT_QUADRIC_SPHERE) return q->phi_max * q->radius * (q->zmax - q->zmin);   /* sphere.cpp:212 */
    return q->phi_max * 0.5f * (q->radius * q->radius - q->inner_radius * q->inner_radius);  /* disk.cpp:131-134 */
}
static bbox quadric_world_bound(const hpt_quadric *q) { /* Shape::WorldBound -> Transform(BBox) transform.cpp:272-283 */
    v3 lo, hi;
    if (q->kind == HPT_QUADRIC_SPHERE) { lo = V(-q->radius, -q->radius, q->zmin); hi = V(q->radius, q->radius, q->zmax); }
    else { lo = V(-q->radius, -q->radius, q->height); hi = V(q->radius, q->radius, q->height); }
    bbox b = bbox_empty();
    for (int k = 0; k < 8; ++k)
        b = bbox_union_p(b, xf_point(q->o2w, V(k & 1 ? hi.x : lo.x, k & 2 ? hi.y : lo.y, k & 4 ? hi.z : lo.z)));
    return b;
}

static int bvh_intersect(const orc_scene *s, const bvh_t *bvh, ray_t *ray, isect_t *is, int anyhit, uint64_t *st);
/* GeometricPrimitive::Intersect (core/primitive.cpp:163-176): shrinks ray.maxt on hit;
 * TransformedPrimitive::Intersect (core/primitive.cpp:95-119) for instance prims */
static int prim_intersect(const orc_scene *s, int64_t prim, ray_t *ray, isect_t *is, uint64_t *st) {
    if (prim < s->ntris) {
        if (!tri_intersect(s, prim, ray, is, 1, st)) return 0;
        ray->maxt = is->t;
        is->inst = -1;
        return 1;
    }
    if (prim < s->ntris + s->d.n_quadrics) {
        float thit, eps;
        if (!quadric_intersect(&s->d.quadrics[prim - s->ntris], ray, &thit, &eps, &is->dg)) return 0;
        is->prim = prim; is->t = thit; is->rayEpsilon = eps; is->b1 = is->b2 = 0.f; is->inst = -1;
        ray->maxt = thit;
        return 1;
    }
    int k = (int)(prim - s->ntris - s->d.n_quadrics);
    xform w2p = anim_interpolate(&s->d.instances[k], ray->time);
    ray_t r2 = *ray;                               /* Ray ray = w2p(r) transform.h:237-243 */
    r2.o = xf_point(w2p.m.m, ray->o);
    r2.d = xf_vec(w2p.m.m, ray->d);
    if (!bvh_intersect(s, &s->inst[k], &r2, is, 0, st)) return 0;
    ray->maxt = r2.maxt;
    is->inst = k;
    is->w2p_m = w2p.m;       /* isect->WorldToObject = identity * w2p ; ObjectToWorld = Inverse(...) */
    if (!m4_is_identity(&w2p.m)) {
        /* PrimitiveToWorld = Inverse(w2p): m = w2p.mInv, mInv = w2p.m */
        is->dg.p = xf_point(w2p.minv.m, is->dg.p);
        is->dg.nn = normalize(xf_normal(w2p.m.m, is->dg.nn));
        is->dg.dpdu = xf_vec(w2p.minv.m, is->dg.dpdu);
        is->dg.dpdv = xf_vec(w2p.minv.m, is->dg.dpdv);
    }
    return 1;
}
static int prim_intersect_p(const orc_scene *s, int64_t prim, const ray_t *ray, uint64_t *st) {
    if (prim < s->ntris) { isect_t tmp; return tri_intersect(s, prim, ray, &tmp, 0, st); }
    if (prim < s->ntris + s->d.n_quadrics) {
        float thit, eps;
        return quadric_intersect(&s->d.quadrics[prim - s->ntris], ray, &thit, &eps, NULL);
    }
    /* TransformedPrimitive::IntersectP (primitive.cpp:122-124) via AnimatedTransform::operator()(Ray)
     * (transform.cpp:416-427): same boundary handling as Interpolate */
    int k = (int)(prim - s->ntris - s->d.n_quadrics);
    xform w2p = anim_interpolate(&s->d.instances[k], ray->time);
    ray_t r2 = *ray;
    r2.o = xf_point(w2p.m.m, ray->o);
    r2.d = xf_vec(w2p.m.m, ray->d);
    isect_t tmp;
    return bvh_intersect(s, &s->inst[k], &r2, &tmp, 1, st);
}

/* ---- BVH build (accelerators/bvh.cpp:153-395, SAH, maxPrimsInNode = 4) ------------------- */
typedef struct { int32_t prim; v3 centroid; bbox bounds; } binfo;
typedef struct bnode { bbox bounds; struct bnode *c[2]; uint32_t axis, first, n; } bnode;
typedef struct { const orc_scene *s; binfo *bd; int32_t *ordered; int64_t nordered; int64_t total; } bctx;

static int64_t partition_pred(binfo *a, int64_t lo, int64_t hi, int (*pred)(const binfo *, void *), void *ctx) {
    int64_t i = lo;
    for (int64_t j = lo; j < hi; ++j)
        if (pred(&a[j], ctx)) { binfo t = a[i]; a[i] = a[j]; a[j] = t; ++i; }
    return i;
}
static void nth_element_dim(binfo *a, int64_t lo, int64_t nth, int64_t hi, int dim) {
    while (hi - lo > 1) {
        float pivot = v3c(a[lo + (hi - lo) / 2].centroid, dim);
        int64_t i = lo, j = hi - 1;
        while (i <= j) {
            while (v3c(a[i].centroid, dim) < pivot) ++i;
            while (v3c(a[j].centroid, dim) > pivot) --j;
            if (i <= j) { binfo t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
        }
        if (nth <= j) hi = j + 1;
        else if (nth >= i) lo = i;
        else return;
    }
}
typedef struct { int split, nb, dim; bbox cb; } bucket_ctx;
static int cmp_bucket(const binfo *p, void *vc) { /* CompareToBucket bvh.cpp:104-110 */
    bucket_ctx *c = (bucket_ctx *)vc;
    int b = c->nb * ((v3c(p->centroid, c->dim) - v3c(c->cb.pmin, c->dim)) / (v3c(c->cb.pmax, c->dim) - v3c(c->cb.pmin, c->dim)));
    if (b == c->nb) b = c->nb - 1;
    return b <= c->split;
}
static bnode *bvh_leaf(bctx *c, bnode *node, int64_t start, int64_t end, bbox bb) {
    node->first = (uint32_t)c->nordered; node->n = (uint32_t)(end - start); node->bounds = bb; node->c[0] = node->c[1] = NULL;
    for (int64_t i = start; i < end; ++i) c->ordered[c->nordered++] = c->bd[i].prim;
    return node;
}
static bnode *bvh_build(bctx *c, int64_t start, int64_t end) {
    const uint32_t maxPrimsInNode = 4;
    c->total++;
    bnode *node = (bnode *)calloc(1, sizeof(bnode));
    bbox bb = bbox_empty();
    for (int64_t i = start; i < end; ++i) bb = bbox_union(bb, c->bd[i].bounds);
    uint32_t nPrimitives = (uint32_t)(end - start);
    if (nPrimitives == 1) return bvh_leaf(c, node, start, end, bb);
    bbox cb = bbox_empty();
    for (int64_t i = start; i < end; ++i) cb = bbox_union_p(cb, c->bd[i].centroid);
    int dim = bbox_maxext(cb);
    int64_t mid = (start + end) / 2;
    if (v3c(cb.pmax, dim) == v3c(cb.pmin, dim)) {
        if (nPrimitives <= maxPrimsInNode) return bvh_leaf(c, node, start, end, bb);
        node->axis = dim; node->bounds = bb; node->n = 0;
        node->c[0] = bvh_build(c, start, mid); node->c[1] = bvh_build(c, mid, end);
        return node;
    }
    if (nPrimitives <= 4) {
        nth_element_dim(c->bd, start, mid, end, dim);
    } else {
        enum { nBuckets = 12 };
        int count[nBuckets]; bbox bounds[nBuckets];
        for (int i = 0; i < nBuckets; ++i) { count[i] = 0; bounds[i] = bbox_empty(); }
        for (int64_t i = start; i < end; ++i) {
            int b = nBuckets * ((v3c(c->bd[i].centroid, dim) - v3c(cb.pmin, dim)) / (v3c(cb.pmax, dim) - v3c(cb.pmin, dim)));
            if (b == nBuckets) b = nBuckets - 1;
            count[b]++; bounds[b] = bbox_union(bounds[b], c->bd[i].bounds);
        }
        float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            bbox b0 = bbox_empty(), b1 = bbox_empty(); int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) { b0 = bbox_union(b0, bounds[j]); c0 += count[j]; }
            for (int j = i + 1; j < nBuckets; ++j) { b1 = bbox_union(b1, bounds[j]); c1 += count[j]; }
            cost[i] = .125f + (c0 * bbox_area(b0) + c1 * bbox_area(b1)) / bbox_area(bb);
        }
        float minCost = cost[0]; int minSplit = 0;
        for (int i = 1; i < nBuckets - 1; ++i) if (cost[i] < minCost) { minCost = cost[i]; minSplit = i; }
        if (nPrimitives > maxPrimsInNode || minCost < nPrimitives) {
            bucket_ctx bc = {minSplit, nBuckets, dim, cb};
            mid = partition_pred(c->bd, start, end, cmp_bucket, &bc);
            if (mid == start || mid == end) { mid = (start + end) / 2; nth_element_dim(c->bd, start, mid, end, dim); }
        } else
            return bvh_leaf(c, node, start, end, bb);
    }
    node->axis = dim; node->bounds = bb; node->n = 0;
    node->c[0] = bvh_build(c, start, mid); node->c[1] = bvh_build(c, mid, end);
    return node;
}
static uint32_t bvh_flatten(lnode *nodes, bnode *n, uint32_t *off) { /* bvh.cpp:377-395 */
    lnode *ln = &nodes[*off];
    ln->bounds = n->bounds;
    uint32_t my = (*off)++;
    if (n->n > 0) { ln->offset = n->first; ln->nprims = (uint8_t)n->n; }
    else {
        ln->axis = (uint8_t)n->axis; ln->nprims = 0;
        bvh_flatten(nodes, n->c[0], off);
        ln->offset = bvh_flatten(nodes, n->c[1], off);
    }
    free(n);
    return my;
}

orc_scene *orc_scene_create(const hpt_scene_desc *desc) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->d = *desc;
#define DUP(field, n, T) do { if ((n) > 0) { T *p_ = (T *)malloc(sizeof(T) * (size_t)(n)); memcpy(p_, desc->field, sizeof(T) * (size_t)(n)); s->d.field = p_; } else s->d.field = NULL; } while (0)
    DUP(meshes, desc->n_meshes, hpt_mesh); DUP(quadrics, desc->n_quadrics, hpt_quadric);
    DUP(materials, desc->n_materials, hpt_material); DUP(lights, desc->n_lights, hpt_light);
    DUP(instances, desc->n_instances, hpt_instance);
    DUP(fpool, desc->n_f, float); DUP(ipool, desc->n_i, int32_t);
#undef DUP
    s->mesh_base = (int64_t *)calloc((size_t)desc->n_meshes + 1, sizeof(int64_t));
    for (int m = 0; m < desc->n_meshes; ++m) { s->mesh_base[m] = s->ntris; s->ntris += desc->meshes[m].ntris; }
    s->nprims = s->ntris + desc->n_quadrics + desc->n_instances;
    s->prim_mesh = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->ntris + 1));
    for (int m = 0; m < desc->n_meshes; ++m)
        for (int64_t t = 0; t < desc->meshes[m].ntris; ++t) s->prim_mesh[s->mesh_base[m] + t] = m;
    s->inst = (bvh_t *)calloc((size_t)desc->n_instances + 1, sizeof(bvh_t));
    if (s->nprims == 0) return s;
    /* one BVH per instance over its own triangles, then the top level over world triangles,
     * quadrics and instances (bounded by their MotionBounds) */
    for (int k = -1 + 0; k < desc->n_instances + 1; ++k) {
        int which = (k == desc->n_instances) ? -1 : k;   /* instances first, top level (-1) last */
        if (k == -1) continue;
        binfo *bd = (binfo *)malloc(sizeof(binfo) * (size_t)s->nprims);
        int64_t n = 0;
        for (int64_t p = 0; p < s->ntris; ++p) {
            if (s->d.meshes[s->prim_mesh[p]].instance != which) continue;
            const hpt_mesh *me; int vi[3]; v3 p1, p2, p3;
            tri_verts(s, p, &me, vi, &p1, &p2, &p3);
            bd[n].prim = (int32_t)p;
            bd[n].bounds = bbox_union_p(bbox_union_p(bbox_union_p(bbox_empty(), p1), p2), p3);
            ++n;
        }
        if (which == -1) {
            for (int q = 0; q < desc->n_quadrics; ++q) { bd[n].prim = (int32_t)(s->ntris + q); bd[n].bounds = quadric_world_bound(&s->d.quadrics[q]); ++n; }
            for (int i = 0; i < desc->n_instances; ++i) {
                const float *bb = s->d.instances[i].bounds;
                bd[n].prim = (int32_t)(s->ntris + desc->n_quadrics + i);
                bd[n].bounds.pmin = V(bb[0], bb[1], bb[2]); bd[n].bounds.pmax = V(bb[3], bb[4], bb[5]);
                ++n;
            }
        }
        for (int64_t i = 0; i < n; ++i)
            bd[i].centroid = vadd(vmul(bd[i].bounds.pmin, .5f), vmul(bd[i].bounds.pmax, .5f)); /* bvh.cpp:47 */
        bvh_t *out = which == -1 ? &s->top : &s->inst[which];
        if (n > 0) {
            bctx c; c.s = s; c.bd = bd; c.ordered = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); c.nordered = 0; c.total = 0;
            bnode *root = bvh_build(&c, 0, n);
            out->ordered = c.ordered; out->nnodes = c.total;
            out->nodes = (lnode *)calloc((size_t)c.total, sizeof(lnode));
            uint32_t off = 0;
            bvh_flatten(out->nodes, root, &off);
        }
        free(bd);
    }
    return s;
}
void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (int k = 0; k < s->d.n_instances; ++k) { free(s->inst[k].ordered); free(s->inst[k].nodes); }
    free((void *)s->d.meshes); free((void *)s->d.quadrics); free((void *)s->d.materials); free((void *)s->d.lights);
    free((void *)s->d.instances); free((void *)s->d.fpool); free((void *)s->d.ipool);
    free(s->mesh_base); free(s->prim_mesh); free(s->top.ordered); free(s->top.nodes); free(s->inst); free(s);
}

/* slab test (accelerators/bvh.cpp:126-148) */
static inline int box_hit(const bbox *b, const ray_t *ray, v3 invDir, const uint32_t neg[3]) {
    const v3 *bb = &b->pmin; /* bounds[0]=pMin, bounds[1]=pMax */
    float tmin = (bb[neg[0]].x - ray->o.x) * invDir.x;
    float tmax = (bb[1 - neg[0]].x - ray->o.x) * invDir.x;
    float tymin = (bb[neg[1]].y - ray->o.y) * invDir.y;
    float tymax = (bb[1 - neg[1]].y - ray->o.y) * invDir.y;
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (bb[neg[2]].z - ray->o.z) * invDir.z;
    float tzmax = (bb[1 - neg[2]].z - ray->o.z) * invDir.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < ray->maxt) && (tmax > ray->mint);
}
/* BVHAccel::Intersect (bvh.cpp:403-454) / IntersectP (:457-503) */
static int bvh_intersect(const orc_scene *s, const bvh_t *bvh, ray_t *ray, isect_t *is, int anyhit, uint64_t *st) {
    if (!bvh->nodes) return 0;
    int hit = 0;
    v3 invDir = V(1.f / ray->d.x, 1.f / ray->d.y, 1.f / ray->d.z);
    uint32_t neg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    uint32_t todoOffset = 0, nodeNum = 0, todo[64];
    while (1) {
        const lnode *node = &bvh->nodes[nodeNum];
        if (st) st[3]++;
        if (box_hit(&node->bounds, ray, invDir, neg)) {
            if (node->nprims > 0) {
                for (uint32_t i = 0; i < node->nprims; ++i) {
                    int64_t prim = bvh->ordered[node->offset + i];
                    if (anyhit) { if (prim_intersect_p(s, prim, ray, st)) return 1; }
                    else if (prim_intersect(s, prim, ray, is, st)) hit = 1;
                }
                if (todoOffset == 0) break;
                nodeNum = todo[--todoOffset];
            } else {
                if (neg[node->axis]) { todo[todoOffset++] = nodeNum + 1; nodeNum = node->offset; }
                else { todo[todoOffset++] = node->offset; nodeNum = nodeNum + 1; }
            }
        } else {
            if (todoOffset == 0) break;
            nodeNum = todo[--todoOffset];
        }
    }
    return hit;
}
static int scene_intersect(const orc_scene *s, ray_t *ray, isect_t *is, int anyhit, uint64_t *st) { /* Scene::Intersect/IntersectP core/scene.h:50-61 */
    if (st) st[anyhit ? 2 : 1]++;
    return bvh_intersect(s, &s->top, ray, is, anyhit, st);
}

/* ---- BSDF (core/reflection.cpp) ---------------------------------------------------------- */
enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16 };
enum { BX_LAMBERT = 1, BX_MICROFACET = 2, BX_IRREG = 3, BX_MICROFACET_COND = 4, BX_FRESNELBLEND = 5 };
/* R: reflectance (Rd for FresnelBlend); R2: k for the conductor / Rs for FresnelBlend; eta in R for the conductor */
typedef struct { int kind, type; rgb R; float exponent; const hpt_material *mat; rgb R2; float ey; } bxdf_t;
typedef struct { v3 nn, ng, sn, tn; int n; bxdf_t bx[2]; v3 p; } bsdf_t;

static inline float cos_theta(v3 w) { return w.z; }
static inline float abs_cos_theta(v3 w) { return fabsf(w.z); }
static inline float sin_theta2(v3 w) { return maxf(0.f, 1.f - cos_theta(w) * cos_theta(w)); }
static inline float sin_theta(v3 w) { return sqrtf(sin_theta2(w)); }
static inline int same_hemisphere(v3 w, v3 wp) { return w.z * wp.z > 0.f; }

static inline v3 w2l(const bsdf_t *b, v3 v) { return V(dot(v, b->sn), dot(v, b->tn), dot(v, b->nn)); } /* reflection.h:166-168 */
static inline v3 l2w(const bsdf_t *b, v3 v) {                                                           /* :169-173 */
    return V(b->sn.x * v.x + b->tn.x * v.y + b->nn.x * v.z, b->sn.y * v.x + b->tn.y * v.y + b->nn.y * v.z,
             b->sn.z * v.x + b->tn.z * v.y + b->nn.z * v.z);
}
/* BSDF ctor (reflection.cpp:601-609) */
static void bsdf_frame(bsdf_t *b, v3 nn_shading, v3 dpdu_shading, v3 ng) {
    b->ng = ng; b->nn = nn_shading; b->sn = normalize(dpdu_shading); b->tn = cross(b->nn, b->sn); b->n = 0;
}
/* Material::GetBSDF: matte.cpp:42-63, plastic.cpp:42-66, measured.cpp:194-210 */
static void bsdf_add_material(bsdf_t *b, const hpt_material *m) {
    rgb kd = {{m->kd[0], m->kd[1], m->kd[2]}}, ks = {{m->ks[0], m->ks[1], m->ks[2]}};
    if (m->kind == HPT_MAT_MATTE) {
        if (!sblack(kd)) { bxdf_t x = {BX_LAMBERT, BSDF_REFLECTION | BSDF_DIFFUSE, kd, 0.f, m}; b->bx[b->n++] = x; }
    } else if (m->kind == HPT_MAT_PLASTIC) {
        if (!sblack(kd)) { bxdf_t x = {BX_LAMBERT, BSDF_REFLECTION | BSDF_DIFFUSE, kd, 0.f, m}; b->bx[b->n++] = x; }
        if (!sblack(ks)) {
            float e = 1.f / m->roughness;
            if (e > 10000.f || isnan(e)) e = 10000.f; /* Blinn ctor reflection.h:424 */
            bxdf_t x = {BX_MICROFACET, BSDF_REFLECTION | BSDF_GLOSSY, ks, e, m}; b->bx[b->n++] = x;
        }
    } else if (m->kind == HPT_MAT_MEASURED_IRREG) {
        bxdf_t x = {BX_IRREG, BSDF_REFLECTION | BSDF_GLOSSY, S(0.f), 0.f, m}; b->bx[b->n++] = x; /* reflection.h:464-466 */
    } else if (m->kind == HPT_MAT_METAL) { /* metal.cpp:51-68: Microfacet(1., FresnelConductor(eta,k), Blinn(1/rough)) */
        float e = 1.f / m->roughness;
        if (e > 10000.f || isnan(e)) e = 10000.f;
        bxdf_t x; memset(&x, 0, sizeof(x));
        x.kind = BX_MICROFACET_COND; x.type = BSDF_REFLECTION | BSDF_GLOSSY; x.exponent = e; x.mat = m;
        x.R.c[0] = m->eta[0]; x.R.c[1] = m->eta[1]; x.R.c[2] = m->eta[2];
        x.R2.c[0] = m->k[0]; x.R2.c[1] = m->k[1]; x.R2.c[2] = m->k[2];
        b->bx[b->n++] = x;
    } else if (m->kind == HPT_MAT_SUBSTRATE) { /* substrate.cpp:42-58: FresnelBlend(d, s, Anisotropic(1/u, 1/v)) */
        if (!sblack(kd) || !sblack(ks)) {
            bxdf_t x; memset(&x, 0, sizeof(x));
            x.kind = BX_FRESNELBLEND; x.type = BSDF_REFLECTION | BSDF_GLOSSY; x.mat = m; x.R = kd; x.R2 = ks;
            float ex = 1.f / m->nu, ey = 1.f / m->nv;           /* Anisotropic ctor reflection.h:439-443 */
            if (ex > 10000.f || isnan(ex)) ex = 10000.f;
            if (ey > 10000.f || isnan(ey)) ey = 10000.f;
            x.exponent = ex; x.ey = ey;
            b->bx[b->n++] = x;
        }
    }
}
/* FresnelDielectric::Evaluate(1.5, 1) (reflection.cpp:115-135) + FrDiel (:60-67); scalar: all
 * three channels are equal because eta_i/eta_t are scalar spectra */
static float fresnel_dielectric(float cosi, float eta_i, float eta_t) {
    cosi = clampf(cosi, -1.f, 1.f);
    int entering = cosi > 0.;
    float ei = eta_i, et = eta_t;
    if (!entering) { float t = ei; ei = et; et = t; }
    float sint = ei / et * sqrtf(maxf(0.f, 1.f - cosi * cosi));
    if (sint >= 1.) return 1.;
    float cost = sqrtf(maxf(0.f, 1.f - sint * sint));
    float ci = fabsf(cosi);
    float Rparl = ((et * ci) - (ei * cost)) / ((et * ci) + (ei * cost));
    float Rperp = ((ei * ci) - (et * cost)) / ((ei * ci) + (et * cost));
    return (Rparl * Rparl + Rperp * Rperp) / 2.f;
}
/* KdTree::privateLookup (core/kdtree.h:159-183) + IrregIsoProc (reflection.cpp:42-55) */
typedef struct { rgb v; float sumWeights; int nFound; } irreg_proc;
static void kd_lookup(const orc_scene *s, const hpt_material *m, uint32_t nodeNum, v3 p, irreg_proc *proc, float *maxDist2) {
    const float *split = s->d.fpool + m->kd_split_off;
    const int32_t *bits = s->d.ipool + m->kd_bits_off;
    const float *data = s->d.fpool + m->kd_data_off;
    uint32_t b = (uint32_t)bits[nodeNum];
    int axis = b & 3; uint32_t hasLeft = (b >> 2) & 1, right = b >> 3, nNodes = (uint32_t)m->kd_nnodes;
    if (axis != 3) {
        float pa = v3c(p, axis);
        float d2 = (pa - split[nodeNum]) * (pa - split[nodeNum]);
        if (pa <= split[nodeNum]) {
            if (hasLeft) kd_lookup(s, m, nodeNum + 1, p, proc, maxDist2);
            if (d2 < *maxDist2 && right < nNodes) kd_lookup(s, m, right, p, proc, maxDist2);
        } else {
            if (right < nNodes) kd_lookup(s, m, right, p, proc, maxDist2);
            if (d2 < *maxDist2 && hasLeft) kd_lookup(s, m, nodeNum + 1, p, proc, maxDist2);
        }
    }
    v3 np = V(data[6 * nodeNum], data[6 * nodeNum + 1], data[6 * nodeNum + 2]);
    float d2 = dist2(np, p);
    if (d2 < *maxDist2) {
        float weight = expf(-100.f * d2);
        rgb sv = {{data[6 * nodeNum + 3], data[6 * nodeNum + 4], data[6 * nodeNum + 5]}};
        proc->v = sadd(proc->v, sscale(sv, weight));
        proc->sumWeights += weight;
        ++proc->nFound;
    }
}
static rgb irreg_f(const orc_scene *s, const hpt_material *m, v3 wo, v3 wi) { /* reflection.cpp:247-272 */
    float cosi = cos_theta(wi), coso = cos_theta(wo);
    float sini = sin_theta(wi), sino = sin_theta(wo);
    float phii = spherical_phi(wi), phio = spherical_phi(wo);
    float dphi = phii - phio;
    if (dphi < 0.) dphi += 2.f * PI_F;
    if (dphi > 2.f * PI_F) dphi -= 2.f * PI_F;
    if (dphi > PI_F) dphi = 2.f * PI_F - dphi;
    v3 mpt = V(sini * sino, dphi / PI_F, cosi * coso);
    float lastMaxDist2 = .001f;
    while (1) {
        irreg_proc proc; proc.v = S(0.f); proc.sumWeights = 0.f; proc.nFound = 0;
        float maxDist2 = lastMaxDist2;
        kd_lookup(s, m, 0, mpt, &proc, &maxDist2);
        if (proc.nFound > 2 || lastMaxDist2 > 1.5f) return sdivf(sclamp0(proc.v), proc.sumWeights);
        lastMaxDist2 *= 2.f;
    }
}
/* FrCond (reflection.cpp:70-79) via FresnelConductor::Evaluate (:110-112) */
static rgb fr_cond(float cosi, rgb eta, rgb k) {
    cosi = fabsf(cosi);
    rgb r;
    for (int c = 0; c < 3; ++c) {
        float e = eta.c[c], kk = k.c[c];
        float tmp = (e * e + kk * kk) * cosi * cosi;
        float Rparl2 = (tmp - (2.f * e * cosi) + 1) / (tmp + (2.f * e * cosi) + 1);
        float tmp_f = e * e + kk * kk;
        float Rperp2 = (tmp_f - (2.f * e * cosi) + cosi * cosi) / (tmp_f + (2.f * e * cosi) + cosi * cosi);
        r.c[c] = (Rparl2 + Rperp2) / 2.f;
    }
    return r;
}
/* Anisotropic::D / Pdf / Sample_f (reflection.h:444-451, reflection.cpp:377-443) */
static float aniso_D(float ex, float ey, v3 wh) {
    float costhetah = abs_cos_theta(wh);
    float d = 1.f - costhetah * costhetah;
    if (d == 0.f) return 0.f;
    float e = (ex * wh.x * wh.x + ey * wh.y * wh.y) / d;
    return sqrtf((ex + 2.f) * (ey + 2.f)) * INV_TWOPI_F * powf(costhetah, e);
}
static float aniso_pdf_wh(float ex, float ey, v3 wo, v3 wh) {
    float costhetah = abs_cos_theta(wh);
    float ds = 1.f - costhetah * costhetah;
    float p = 0.f;
    if (ds > 0.f && dot(wo, wh) > 0.f) {
        float e = (ex * wh.x * wh.x + ey * wh.y * wh.y) / ds;
        float d = sqrtf((ex + 1.f) * (ey + 1.f)) * INV_TWOPI_F * powf(costhetah, e);
        p = d / (4.f * dot(wo, wh));
    }
    return p;
}
static void aniso_first_quadrant(float ex, float ey, float u1, float u2, float *phi, float *costheta) {
    if (ex == ey) *phi = PI_F * u1 * 0.5f;
    else *phi = atanf(sqrtf((ex + 1.f) / (ey + 1.f)) * tanf(PI_F * u1 * 0.5f));
    float cosphi = cosf(*phi), sinphi = sinf(*phi);
    *costheta = powf(u2, 1.f / (ex * cosphi * cosphi + ey * sinphi * sinphi + 1));
}
static void aniso_sample(float ex, float ey, v3 wo, v3 *wi, float u1, float u2, float *pdf) {
    float phi, costheta;
    if (u1 < .25f) aniso_first_quadrant(ex, ey, 4.f * u1, u2, &phi, &costheta);
    else if (u1 < .5f) { u1 = 4.f * (.5f - u1); aniso_first_quadrant(ex, ey, u1, u2, &phi, &costheta); phi = PI_F - phi; }
    else if (u1 < .75f) { u1 = 4.f * (u1 - .5f); aniso_first_quadrant(ex, ey, u1, u2, &phi, &costheta); phi += PI_F; }
    else { u1 = 4.f * (1.f - u1); aniso_first_quadrant(ex, ey, u1, u2, &phi, &costheta); phi = 2.f * PI_F - phi; }
    float sintheta = sqrtf(maxf(0.f, 1.f - costheta * costheta));
    v3 wh = V(sintheta * cosf(phi), sintheta * sinf(phi), costheta);
    if (!same_hemisphere(wo, wh)) wh = vneg(wh);
    *wi = vadd(vneg(wo), vmul(wh, 2.f * dot(wo, wh)));
    *pdf = aniso_pdf_wh(ex, ey, wo, wh);
}
static void concentric_sample_disk(float u1, float u2, float *dx, float *dy);
static rgb bxdf_f(const orc_scene *s, const bxdf_t *x, v3 wo, v3 wi) {
    if (x->kind == BX_LAMBERT) return sscale(x->R, INV_PI_F);                 /* reflection.cpp:173-175 */
    if (x->kind == BX_MICROFACET) {                                           /* :211-222 */
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
        if (cosThetaI == 0.f || cosThetaO == 0.f) return S(0.f);
        v3 wh = vadd(wi, wo);
        if (wh.x == 0. && wh.y == 0. && wh.z == 0.) return S(0.f);
        wh = normalize(wh);
        float cosThetaH = dot(wi, wh);
        float F = fresnel_dielectric(cosThetaH, 1.5f, 1.f);
        float D = (x->exponent + 2) * INV_TWOPI_F * powf(abs_cos_theta(wh), x->exponent); /* Blinn::D reflection.h:427-430 */
        float NdotWh = abs_cos_theta(wh), NdotWo = abs_cos_theta(wo), NdotWi = abs_cos_theta(wi);
        float WOdotWh = absdot(wo, wh);
        float G = minf(1.f, minf((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh))); /* :403-410 */
        return sdivf(smul(sscale(sscale(x->R, D), G), S(F)), (4.f * cosThetaI * cosThetaO));
    }
    if (x->kind == BX_MICROFACET_COND) { /* Microfacet::f with FresnelConductor, R = 1 (reflection.cpp:211-222) */
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
        if (cosThetaI == 0.f || cosThetaO == 0.f) return S(0.f);
        v3 wh = vadd(wi, wo);
        if (wh.x == 0. && wh.y == 0. && wh.z == 0.) return S(0.f);
        wh = normalize(wh);
        float cosThetaH = dot(wi, wh);
        rgb F = fr_cond(cosThetaH, x->R, x->R2);
        float D = (x->exponent + 2) * INV_TWOPI_F * powf(abs_cos_theta(wh), x->exponent);
        float NdotWh = abs_cos_theta(wh), NdotWo = abs_cos_theta(wo), NdotWi = abs_cos_theta(wi);
        float WOdotWh = absdot(wo, wh);
        float G = minf(1.f, minf((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh)));
        return sdivf(smul(sscale(sscale(S(1.f), D), G), F), (4.f * cosThetaI * cosThetaO));
    }
    if (x->kind == BX_FRESNELBLEND) { /* FresnelBlend::f (reflection.cpp:232-244) */
        rgb one_minus_rs = {{1.f - x->R2.c[0], 1.f - x->R2.c[1], 1.f - x->R2.c[2]}};
        rgb diffuse = sscale(sscale(smul(sscale(x->R, (28.f / (23.f * PI_F))), one_minus_rs),
                                    (1.f - powf(1.f - .5f * abs_cos_theta(wi), 5))), (1.f - powf(1.f - .5f * abs_cos_theta(wo), 5)));
        v3 wh = vadd(wi, wo);
        if (wh.x == 0. && wh.y == 0. && wh.z == 0.) return S(0.f);
        wh = normalize(wh);
        float sc = aniso_D(x->exponent, x->ey, wh) / (4.f * absdot(wi, wh) * maxf(abs_cos_theta(wi), abs_cos_theta(wo)));
        float pw = powf(1 - dot(wi, wh), 5.f);
        rgb schlick = sadd(x->R2, sscale(one_minus_rs, pw));   /* SchlickFresnel reflection.h:468-470 */
        return sadd(diffuse, sscale(schlick, sc));
    }
    return irreg_f(s, x->mat, wo, wi);
}
static float blinn_pdf(float exponent, v3 wo, v3 wi) { /* reflection.cpp:366-374 */
    v3 wh = normalize(vadd(wo, wi));
    float costheta = abs_cos_theta(wh);
    float p = ((exponent + 1.f) * powf(costheta, exponent)) / (2.f * PI_F * 4.f * dot(wo, wh));
    if (dot(wo, wh) <= 0.f) p = 0.f;
    return p;
}
static float bxdf_pdf(const bxdf_t *x, v3 wo, v3 wi) {
    if (x->kind == BX_FRESNELBLEND) { /* FresnelBlend::Pdf (reflection.cpp:465-468) */
        if (!same_hemisphere(wo, wi)) return 0.f;
        return .5f * (abs_cos_theta(wi) * INV_PI_F + aniso_pdf_wh(x->exponent, x->ey, wo, normalize(vadd(wo, wi))));
    }
    if (x->kind == BX_MICROFACET || x->kind == BX_MICROFACET_COND) { if (!same_hemisphere(wo, wi)) return 0.f; return blinn_pdf(x->exponent, wo, wi); } /* :340-343 */
    return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * INV_PI_F : 0.f;                                                /* :321-323 */
}
static void concentric_sample_disk(float u1, float u2, float *dx, float *dy) { /* montecarlo.cpp:306-348 */
    float r, theta;
    float sx = 2 * u1 - 1, sy = 2 * u2 - 1;
    if (sx == 0.0 && sy == 0.0) { *dx = 0.0; *dy = 0.0; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; if (sy > 0.0) theta = sy / r; else theta = 8.0f + sy / r; }
        else { r = sy; theta = 2.0f - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
        else { r = -sy; theta = 6.0f + sx / r; }
    }
    theta *= PI_F / 4.f;
    *dx = r * cosf(theta);
    *dy = r * sinf(theta);
}
static rgb bxdf_sample_f(const orc_scene *s, const bxdf_t *x, v3 wo, v3 *wi, float u1, float u2, float *pdf) {
    if (x->kind == BX_FRESNELBLEND) { /* FresnelBlend::Sample_f (reflection.cpp:446-462) */
        if (u1 < .5) {
            u1 = 2.f * u1;
            v3 w; concentric_sample_disk(u1, u2, &w.x, &w.y);
            w.z = sqrtf(maxf(0.f, 1.f - w.x * w.x - w.y * w.y));
            if (wo.z < 0.) w.z *= -1.f;
            *wi = w;
        } else {
            u1 = 2.f * (u1 - .5f);
            aniso_sample(x->exponent, x->ey, wo, wi, u1, u2, pdf);
            if (!same_hemisphere(wo, *wi)) return S(0.f);
        }
        *pdf = bxdf_pdf(x, wo, *wi);
        return bxdf_f(s, x, wo, *wi);
    }
    if (x->kind == BX_MICROFACET || x->kind == BX_MICROFACET_COND) { /* Microfacet::Sample_f :332-337 + Blinn::Sample_f :346-363 */
        float costheta = powf(u1, 1.f / (x->exponent + 1));
        float sintheta = sqrtf(maxf(0.f, 1.f - costheta * costheta));
        float phi = u2 * 2.f * PI_F;
        v3 wh = V(sintheta * cosf(phi), sintheta * sinf(phi), costheta); /* SphericalDirection geometry.h:624-629 */
        if (!same_hemisphere(wo, wh)) wh = vneg(wh);
        *wi = vadd(vneg(wo), vmul(wh, 2.f * dot(wo, wh)));
        float bp = ((x->exponent + 1.f) * powf(costheta, x->exponent)) / (2.f * PI_F * 4.f * dot(wo, wh));
        if (dot(wo, wh) <= 0.f) bp = 0.f;
        *pdf = bp;
        if (!same_hemisphere(wo, *wi)) return S(0.f);
        return bxdf_f(s, x, wo, *wi);
    }
    /* BxDF::Sample_f :311-318: cosine hemisphere */
    v3 w; concentric_sample_disk(u1, u2, &w.x, &w.y);
    w.z = sqrtf(maxf(0.f, 1.f - w.x * w.x - w.y * w.y));
    if (wo.z < 0.) w.z *= -1.f;
    *wi = w;
    *pdf = bxdf_pdf(x, wo, w);
    return bxdf_f(s, x, wo, w);
}
static inline int bx_match(const bxdf_t *x, int flags) { return (x->type & flags) == x->type; }
/* BSDF::f (reflection.cpp:612-626) */
static rgb bsdf_f(const orc_scene *s, const bsdf_t *b, v3 woW, v3 wiW, int flags) {
    v3 wi = w2l(b, wiW), wo = w2l(b, woW);
    if (dot(wiW, b->ng) * dot(woW, b->ng) > 0) flags &= ~BSDF_TRANSMISSION;
    else flags &= ~BSDF_REFLECTION;
    rgb f = S(0.f);
    for (int i = 0; i < b->n; ++i) if (bx_match(&b->bx[i], flags)) f = sadd(f, bxdf_f(s, &b->bx[i], wo, wi));
    return f;
}
/* BSDF::Pdf (reflection.cpp:583-598) */
static float bsdf_pdf(const bsdf_t *b, v3 woW, v3 wiW, int flags) {
    if (b->n == 0.) return 0.;
    v3 wo = w2l(b, woW), wi = w2l(b, wiW);
    float pdf = 0.f; int matching = 0;
    for (int i = 0; i < b->n; ++i) if (bx_match(&b->bx[i], flags)) { ++matching; pdf += bxdf_pdf(&b->bx[i], wo, wi); }
    return matching > 0 ? pdf / matching : 0.f;
}
/* BSDF::Sample_f (reflection.cpp:522-580) */
static rgb bsdf_sample_f(const orc_scene *s, const bsdf_t *b, v3 woW, v3 *wiW, float u1, float u2, float uComp,
                         float *pdf, int flags, int *sampledType) {
    int matching = 0;
    for (int i = 0; i < b->n; ++i) if (bx_match(&b->bx[i], flags)) ++matching;
    if (matching == 0) { *pdf = 0.f; *sampledType = 0; return S(0.f); }
    int which = (int)floorf(uComp * matching);
    if (which > matching - 1) which = matching - 1;
    const bxdf_t *bx = NULL; int count = which;
    for (int i = 0; i < b->n; ++i) if (bx_match(&b->bx[i], flags) && count-- == 0) { bx = &b->bx[i]; break; }
    v3 wo = w2l(b, woW), wi;
    *pdf = 0.f;
    rgb f = bxdf_sample_f(s, bx, wo, &wi, u1, u2, pdf);
    if (*pdf == 0.f) { *sampledType = 0; return S(0.f); }
    *sampledType = bx->type;
    *wiW = l2w(b, wi);
    if (!(bx->type & BSDF_SPECULAR) && matching > 1)
        for (int i = 0; i < b->n; ++i) if (&b->bx[i] != bx && bx_match(&b->bx[i], flags)) *pdf += bxdf_pdf(&b->bx[i], wo, wi);
    if (matching > 1) *pdf /= matching;
    if (!(bx->type & BSDF_SPECULAR)) {
        f = S(0.f);
        if (dot(*wiW, b->ng) * dot(woW, b->ng) > 0) flags &= ~BSDF_TRANSMISSION;
        else flags &= ~BSDF_REFLECTION;
        for (int i = 0; i < b->n; ++i) if (bx_match(&b->bx[i], flags)) f = sadd(f, bxdf_f(s, &b->bx[i], wo, wi));
    }
    return f;
}

/* Intersection::GetBSDF -> GeometricPrimitive::GetBSDF -> Triangle::GetShadingGeometry
 * (core/intersection.cpp:41-48, core/primitive.cpp:184-190, shapes/trianglemesh.cpp:293-368) */
static void get_bsdf(const orc_scene *s, const isect_t *is, bsdf_t *b) {
    const dgeom *dg = &is->dg;
    v3 ns_nn = dg->nn, ns_dpdu = dg->dpdu;
    const hpt_material *mat;
    if (is->prim < s->ntris) {
        const hpt_mesh *me; int vi[3]; v3 p1, p2, p3;
        tri_verts(s, is->prim, &me, vi, &p1, &p2, &p3);
        mat = &s->d.materials[me->material];
        if (me->n_off >= 0) {
            float bb[3], uv[3][2];
            tri_uvs(s, me, vi, uv);
            float A00 = uv[1][0] - uv[0][0], A01 = uv[2][0] - uv[0][0], A10 = uv[1][1] - uv[0][1], A11 = uv[2][1] - uv[0][1];
            float C0 = dg->u - uv[0][0], C1 = dg->v - uv[0][1];
            /* SolveLinearSystem2x2 core/transform.cpp:39-49 */
            float det = A00 * A11 - A01 * A10;
            int ok = 1;
            if (fabsf(det) < 1e-10f) ok = 0;
            else {
                bb[1] = (A11 * C0 - A01 * C1) / det;
                bb[2] = (A00 * C1 - A10 * C0) / det;
                if (isnan(bb[1]) || isnan(bb[2])) ok = 0;
            }
            if (!ok) bb[0] = bb[1] = bb[2] = 1.f / 3.f;
            else bb[0] = 1.f - bb[1] - bb[2];
            const float *N = s->d.fpool + me->n_off;
            v3 n0 = V(N[3 * vi[0]], N[3 * vi[0] + 1], N[3 * vi[0] + 2]);
            v3 n1 = V(N[3 * vi[1]], N[3 * vi[1] + 1], N[3 * vi[1] + 2]);
            v3 n2 = V(N[3 * vi[2]], N[3 * vi[2] + 1], N[3 * vi[2] + 2]);
            /* b[0]*n0 + b[1]*n1 + b[2]*n2 : Normal operator*(float f, Normal) = (f*x..), left-to-right + */
            v3 nsum = vadd(vadd(V(bb[0] * n0.x, bb[0] * n0.y, bb[0] * n0.z), V(bb[1] * n1.x, bb[1] * n1.y, bb[1] * n1.z)),
                           V(bb[2] * n2.x, bb[2] * n2.y, bb[2] * n2.z));
            /* obj2world = isect.ObjectToWorld: for an instance hit its mInv is w2p.m (primitive.cpp:104-107) */
            v3 ns = normalize(xf_normal(is->inst >= 0 ? is->w2p_m.m : me->o2w_inv, nsum));
            v3 ss = normalize(dg->dpdu);
            v3 ts = cross(ss, ns);
            if (vlen2(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, &ss, &ts);
            /* dgShading = DifferentialGeometry(dg.p, ss, ts, ...) : nn = Normalize(Cross(ss, ts)), flipped */
            dgeom dgs;
            dg_init(&dgs, dg->p, ss, ts, dg->u, dg->v, me->reverse_orientation ^ me->swaps_handedness);
            ns_nn = dgs.nn; ns_dpdu = dgs.dpdu;
        }
    } else mat = &s->d.materials[s->d.quadrics[is->prim - s->ntris].material];
    bsdf_frame(b, ns_nn, ns_dpdu, dg->nn);
    b->p = dg->p;
    bsdf_add_material(b, mat);
}

/* ---- lights ------------------------------------------------------------------------------ */
static int prim_arealight(const orc_scene *s, int64_t prim) {
    if (prim < s->ntris) return s->d.meshes[s->prim_mesh[prim]].arealight;
    return s->d.quadrics[prim - s->ntris].arealight;
}
/* DiffuseAreaLight::L (lights/diffuse.h:51-53) via Intersection::Le (core/intersection.cpp:61-64) */
static rgb area_L(const hpt_light *l, v3 n, v3 w) {
    rgb Le = {{l->intensity[0], l->intensity[1], l->intensity[2]}};
    return dot(n, w) > 0.f ? Le : S(0.f);
}
static rgb isect_Le(const orc_scene *s, const isect_t *is, v3 w) {
    int al = prim_arealight(s, is->prim);
    return al >= 0 ? area_L(&s->d.lights[al], is->dg.nn, w) : S(0.f);
}
/* MIPMap::Lookup(s,t,width=0) -> triangle(0,s,t) (core/mipmap.h:238-269), TEXTURE_REPEAT Texel (:204-223) */
static int mod_i(int a, int b) { int n = (int)(a / b); a -= n * b; if (a < 0) a += b; return a; } /* pbrt.h:217-222 */
static rgb env_texel(const orc_scene *s, const hpt_light *l, int si, int ti) {
    si = mod_i(si, l->env_w); ti = mod_i(ti, l->env_h);
    const float *t = s->d.fpool + l->tex_off + 3 * ((int64_t)ti * l->env_w + si);
    rgb r = {{t[0], t[1], t[2]}};
    return r;
}
static rgb env_lookup(const orc_scene *s, const hpt_light *l, float sc, float tc) {
    sc = sc * l->env_w - 0.5f;
    tc = tc * l->env_h - 0.5f;
    int s0 = (int)floorf(sc), t0 = (int)floorf(tc);
    float ds = sc - s0, dt = tc - t0;
    return sadd(sadd(sadd(sscale(env_texel(s, l, s0, t0), (1.f - ds) * (1.f - dt)), sscale(env_texel(s, l, s0, t0 + 1), (1.f - ds) * dt)),
                     sscale(env_texel(s, l, s0 + 1, t0), ds * (1.f - dt))), sscale(env_texel(s, l, s0 + 1, t0 + 1), ds * dt));
}
/* Light::Le (core/light.cpp:58-60) / InfiniteAreaLight::Le (lights/infinite.cpp:117-122) */
static rgb light_Le(const orc_scene *s, const hpt_light *l, v3 d) {
    if (l->kind != HPT_LIGHT_INFINITE) return S(0.f);
    v3 wh = normalize(xf_vec(l->l2w_inv, d));
    float sc = spherical_phi(wh) * INV_TWOPI_F;
    float tc = spherical_theta(wh) * INV_PI_F;
    return env_lookup(s, l, sc, tc);
}
/* Distribution1D::SampleContinuous (core/montecarlo.h:80-97): upper_bound on cdf[0..count] */
static float dist1d_sample(const float *func, const float *cdf, float funcInt, int count, float u, float *pdf, int *off) {
    int lo = 0, hi = count + 1; /* first element > u */
    while (lo < hi) { int mid = (lo + hi) / 2; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
    int offset = lo - 1; if (offset < 0) offset = 0;
    if (off) *off = offset;
    float du = (u - cdf[offset]) / (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = func[offset] / funcInt;
    return (offset + du) / count;
}

typedef struct { v3 wi; float pdf; rgb Li; ray_t shadow; } lsample;

/* Sphere::Sample(p,u1,u2) (shapes/sphere.cpp:236-261), Disk::Sample (disk.cpp:147-156) */
static v3 quadric_sample(const hpt_quadric *q, v3 p, float u1, float u2, v3 *ns) {
    if (q->kind == HPT_QUADRIC_DISK) {
        v3 pd; concentric_sample_disk(u1, u2, &pd.x, &pd.y);
        pd.x *= q->radius; pd.y *= q->radius; pd.z = q->height;
        *ns = normalize(xf_normal(q->o2w_inv, V(0, 0, 1)));
        if (q->reverse_orientation) *ns = vmul(*ns, -1.f);
        return xf_point(q->o2w, pd);
    }
    v3 Pcenter = xf_point(q->o2w, V(0, 0, 0));
    v3 wc = normalize(vsub(Pcenter, p));
    v3 wcX, wcY; coordinate_system(wc, &wcX, &wcY);
    if (dist2(p, Pcenter) - q->radius * q->radius < 1e-4f) { /* Sphere::Sample(u1,u2) :227-233 + UniformSampleSphere montecarlo.cpp:283-290 */
        float z = 1.f - 2.f * u1;
        float r = sqrtf(maxf(0.f, 1.f - z * z));
        float phi = 2.f * PI_F * u2;
        v3 ps = vmul(V(r * cosf(phi), r * sinf(phi), z), q->radius); /* Point(0,0,0) + radius * v */
        *ns = normalize(xf_normal(q->o2w_inv, ps));
        if (q->reverse_orientation) *ns = vmul(*ns, -1.f);
        return xf_point(q->o2w, ps);
    }
    float sinThetaMax2 = q->radius * q->radius / dist2(p, Pcenter);
    float cosThetaMax = sqrtf(maxf(0.f, 1.f - sinThetaMax2));
    /* UniformSampleCone(u1,u2,costhetamax,x,y,z) montecarlo.cpp:413-420 */
    float costheta = (1.f - u1) * cosThetaMax + u1 * 1.f;
    float sintheta = sqrtf(1.f - costheta * costheta);
    float phi = u2 * 2.f * PI_F;
    v3 dir = vadd(vadd(vmul(wcX, cosf(phi) * sintheta), vmul(wcY, sinf(phi) * sintheta)), vmul(wc, costheta));
    ray_t r; r.o = p; r.d = dir; r.mint = 1e-3f; r.maxt = INFINITY; r.time = 0.f; r.depth = 0;
    float thit, eps; dgeom dgs;
    if (!quadric_intersect(q, &r, &thit, &eps, &dgs)) thit = dot(vsub(Pcenter, p), normalize(r.d));
    v3 ps = ray_at(&r, thit);
    *ns = normalize(vsub(ps, Pcenter));
    if (q->reverse_orientation) *ns = vmul(*ns, -1.f);
    return ps;
}
/* Shape::Pdf(p,wi) (core/shape.cpp:86-99), Sphere::Pdf (sphere.cpp:264-274) */
static float quadric_pdf(const hpt_quadric *q, v3 p, v3 wi) {
    if (q->kind == HPT_QUADRIC_SPHERE) {
        v3 Pcenter = xf_point(q->o2w, V(0, 0, 0));
        if (!(dist2(p, Pcenter) - q->radius * q->radius < 1e-4f)) {
            float sinThetaMax2 = q->radius * q->radius / dist2(p, Pcenter);
            float cosThetaMax = sqrtf(maxf(0.f, 1.f - sinThetaMax2));
            return 1.f / (2.f * PI_F * (1.f - cosThetaMax)); /* UniformConePdf montecarlo.cpp:400-402 */
        }
    }
    ray_t ray; ray.o = p; ray.d = wi; ray.mint = 1e-3f; ray.maxt = INFINITY; ray.time = 0.f; ray.depth = -1;
    float thit, eps; dgeom dgl;
    if (!quadric_intersect(q, &ray, &thit, &eps, &dgl)) return 0.;
    float pdf = dist2(p, ray_at(&ray, thit)) / (absdot(dgl.nn, vneg(wi)) * quadric_area(q));
    if (isinf(pdf)) pdf = 0.f;
    return pdf;
}
/* Light::Pdf(p, wi) */
static float light_pdf(const orc_scene *s, const hpt_light *l, v3 p, v3 wi) {
    if (l->kind == HPT_LIGHT_DIFFUSE_AREA) { /* ShapeSet::Pdf core/light.cpp:157-162 */
        float pdf = 0.f;
        pdf += l->area * quadric_pdf(&s->d.quadrics[l->quadric], p, wi);
        float sumArea = 0.f; sumArea += l->area;
        return pdf / sumArea;
    }
    if (l->kind == HPT_LIGHT_INFINITE) { /* lights/infinite.cpp:224-234 + Distribution2D::Pdf montecarlo.h:153-161 */
        v3 w = xf_vec(l->l2w_inv, wi);
        float theta = spherical_theta(w), phi = spherical_phi(w);
        float sintheta = sinf(theta);
        if (sintheta == 0.f) return 0.f;
        float u = phi * INV_TWOPI_F, v = theta * INV_PI_F;
        int iu = (int)(u * l->env_w); if (iu < 0) iu = 0; if (iu > l->env_w - 1) iu = l->env_w - 1;
        int iv = (int)(v * l->env_h); if (iv < 0) iv = 0; if (iv > l->env_h - 1) iv = l->env_h - 1;
        const float *cf = s->d.fpool + l->cond_func_off, *ci = s->d.fpool + l->cond_int_off, *mf = s->d.fpool + l->marg_func_off;
        float dp;
        if (ci[iv] * l->marg_int == 0.f) dp = 0.f;
        else dp = (cf[(int64_t)iv * l->env_w + iu] * mf[iv]) / (ci[iv] * l->marg_int);
        return dp / (2.f * PI_F * PI_F * sintheta);
    }
    return 0.f; /* PointLight::Pdf lights/point.cpp */
}
/* Light::Sample_L(p, pEpsilon, ls, time, &wi, &pdf, &visibility) */
static rgb light_sample_L(const orc_scene *s, const hpt_light *l, v3 p, float pEps, float uPos0, float uPos1, float uComp,
                          float time, lsample *o) {
    (void)uComp;
    o->shadow.time = time; o->shadow.depth = 0;
    if (l->kind == HPT_LIGHT_POINT) { /* lights/point.cpp:50-57 */
        v3 lp = V(l->pos[0], l->pos[1], l->pos[2]);
        o->wi = normalize(vsub(lp, p));
        o->pdf = 1.f;
        float d = vlen(vsub(p, lp)); /* VisibilityTester::SetSegment core/light.h:87-92: Distance(p1,p2) */
        o->shadow.o = p; o->shadow.d = vdiv(vsub(lp, p), d); o->shadow.mint = pEps; o->shadow.maxt = d * (1.f - 0.f);
        rgb I = {{l->intensity[0], l->intensity[1], l->intensity[2]}};
        return sdivf(I, dist2(lp, p));
    }
    if (l->kind == HPT_LIGHT_DIFFUSE_AREA) { /* lights/diffuse.cpp:69-81 */
        const hpt_quadric *q = &s->d.quadrics[l->quadric];
        v3 ns; v3 ps = quadric_sample(q, p, uPos0, uPos1, &ns);
        o->wi = normalize(vsub(ps, p));
        o->pdf = light_pdf(s, l, p, o->wi);
        float d = vlen(vsub(p, ps));
        o->shadow.o = p; o->shadow.d = vdiv(vsub(ps, p), d); o->shadow.mint = pEps; o->shadow.maxt = d * (1.f - 1e-3f);
        return area_L(l, ns, vneg(o->wi));
    }
    /* InfiniteAreaLight::Sample_L lights/infinite.cpp:195-221 */
    const float *cf = s->d.fpool + l->cond_func_off, *cc = s->d.fpool + l->cond_cdf_off, *ci = s->d.fpool + l->cond_int_off;
    const float *mf = s->d.fpool + l->marg_func_off, *mc = s->d.fpool + l->marg_cdf_off;
    float uv[2], pdfs[2]; int v;
    uv[1] = dist1d_sample(mf, mc, l->marg_int, l->env_h, uPos1, &pdfs[1], &v);
    uv[0] = dist1d_sample(cf + (int64_t)v * l->env_w, cc + (int64_t)v * (l->env_w + 1), ci[v], l->env_w, uPos0, &pdfs[0], NULL);
    float mapPdf = pdfs[0] * pdfs[1];
    if (mapPdf == 0.f) { o->pdf = 0.f; return S(0.f); }
    float theta = uv[1] * PI_F, phi = uv[0] * 2.f * PI_F;
    float costheta = cosf(theta), sintheta = sinf(theta);
    float sinphi = sinf(phi), cosphi = cosf(phi);
    o->wi = xf_vec(l->l2w, V(sintheta * cosphi, sintheta * sinphi, costheta));
    o->pdf = mapPdf / (2.f * PI_F * PI_F * sintheta);
    if (sintheta == 0.f) o->pdf = 0.f;
    o->shadow.o = p; o->shadow.d = o->wi; o->shadow.mint = pEps; o->shadow.maxt = INFINITY; /* SetRay light.h:93-96 */
    return env_lookup(s, l, uv[0], uv[1]);
}
static inline float power_heuristic(int nf, float fPdf, int ng, float gPdf) { /* montecarlo.h:266-269 */
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}

/* EstimateDirect (core/integrator.cpp:117-174) */
static rgb estimate_direct(const orc_scene *s, int lightNum, v3 p, v3 n, v3 wo, float rayEpsilon, float time,
                           const bsdf_t *bsdf, const float ls[3], const float bs[3], uint64_t *st) {
    const hpt_light *light = &s->d.lights[lightNum];
    const int flags = (BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_DIFFUSE | BSDF_GLOSSY); /* BSDF_ALL & ~BSDF_SPECULAR */
    int isDelta = light->kind == HPT_LIGHT_POINT;
    rgb Ld = S(0.f);
    lsample lsmp; float bsdfPdf;
    rgb Li = light_sample_L(s, light, p, rayEpsilon, ls[0], ls[1], ls[2], time, &lsmp);
    float lightPdf = lsmp.pdf;
    v3 wi = lsmp.wi;
    if (lightPdf > 0. && !sblack(Li)) {
        rgb f = bsdf_f(s, bsdf, wo, wi, flags);
        if (!sblack(f)) {
            isect_t tmp;
            ray_t sr = lsmp.shadow;
            if (!scene_intersect(s, &sr, &tmp, 1, st)) {
                if (isDelta) Ld = sadd(Ld, sscale(smul(f, Li), (absdot(wi, n) / lightPdf)));
                else {
                    bsdfPdf = bsdf_pdf(bsdf, wo, wi, flags);
                    float weight = power_heuristic(1, lightPdf, 1, bsdfPdf);
                    Ld = sadd(Ld, sscale(smul(f, Li), (absdot(wi, n) * weight / lightPdf)));
                }
            }
        }
    }
    if (!isDelta) {
        int sampledType;
        rgb f = bsdf_sample_f(s, bsdf, wo, &wi, bs[0], bs[1], bs[2], &bsdfPdf, flags, &sampledType);
        if (!sblack(f) && bsdfPdf > 0.) {
            float weight = 1.f;
            if (!(sampledType & BSDF_SPECULAR)) {
                lightPdf = light_pdf(s, light, p, wi);
                if (lightPdf == 0.) return Ld;
                weight = power_heuristic(1, bsdfPdf, 1, lightPdf);
            }
            isect_t lightIsect;
            rgb Li2 = S(0.f);
            ray_t ray; ray.o = p; ray.d = wi; ray.mint = rayEpsilon; ray.maxt = INFINITY; ray.time = time; ray.depth = 0;
            if (scene_intersect(s, &ray, &lightIsect, 0, st)) {
                if (prim_arealight(s, lightIsect.prim) == lightNum) Li2 = isect_Le(s, &lightIsect, vneg(wi));
            } else Li2 = light_Le(s, light, ray.d);
            if (!sblack(Li2)) Ld = sadd(Ld, sdivf(sscale(sscale(smul(f, Li2), absdot(wi, n)), weight), bsdfPdf));
        }
    }
    return Ld;
}

/* PathIntegrator::Li (integrators/path.cpp:52-123) on top of SamplerRenderer::Li
 * (renderers/samplerrenderer.cpp:320-342) */
static rgb path_Li(const orc_scene *s, ray_t ray, const cam_sample *sample, int maxDepth, draw_src *rng, uint64_t *st) {
    isect_t isect;
    rgb L = S(0.f);
    int nLights = s->d.n_lights;
    if (!scene_intersect(s, &ray, &isect, 0, st)) {
        for (int i = 0; i < nLights; ++i) L = sadd(L, light_Le(s, &s->d.lights[i], ray.d));
        return L;
    }
    rgb pathThroughput = S(1.f);
    int specularBounce = 0;
    for (int bounces = 0;; ++bounces) {
        if (bounces == 0 || specularBounce) L = sadd(L, smul(pathThroughput, isect_Le(s, &isect, vneg(ray.d))));
        bsdf_t bsdf;
        get_bsdf(s, &isect, &bsdf);
        v3 p = bsdf.p, n = bsdf.nn;
        v3 wo = vneg(ray.d);
        /* UniformSampleOneLight (core/integrator.cpp:82-114) */
        if (nLights > 0) {
            float ln, ls[3], bs[3];
            if (bounces < 3) {
                ln = sample->oneD[4 * bounces + 1];
                ls[0] = sample->twoD[3 * bounces][0]; ls[1] = sample->twoD[3 * bounces][1]; ls[2] = sample->oneD[4 * bounces];
                bs[0] = sample->twoD[3 * bounces + 1][0]; bs[1] = sample->twoD[3 * bounces + 1][1]; bs[2] = sample->oneD[4 * bounces + 2];
            } else {
                ln = draw_float(rng);
                ls[0] = draw_float(rng); ls[1] = draw_float(rng); ls[2] = draw_float(rng);
                bs[0] = draw_float(rng); bs[1] = draw_float(rng); bs[2] = draw_float(rng);
            }
            int lightNum = (int)floorf(ln * nLights);
            if (lightNum > nLights - 1) lightNum = nLights - 1;
            rgb Ld = estimate_direct(s, lightNum, p, n, wo, isect.rayEpsilon, ray.time, &bsdf, ls, bs, st);
            L = sadd(L, smul(pathThroughput, sscale(Ld, (float)nLights)));
        }
        float ps[3];
        if (bounces < 3) { ps[0] = sample->twoD[3 * bounces + 2][0]; ps[1] = sample->twoD[3 * bounces + 2][1]; ps[2] = sample->oneD[4 * bounces + 3]; }
        else { ps[0] = draw_float(rng); ps[1] = draw_float(rng); ps[2] = draw_float(rng); }
        v3 wi; float pdf; int flags;
        rgb f = bsdf_sample_f(s, &bsdf, wo, &wi, ps[0], ps[1], ps[2], &pdf,
                              BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_DIFFUSE | BSDF_GLOSSY | BSDF_SPECULAR, &flags);
        if (sblack(f) || pdf == 0.) break;
        specularBounce = (flags & BSDF_SPECULAR) != 0;
        pathThroughput = smul(pathThroughput, sdivf(sscale(f, absdot(wi, n)), pdf));
        { ray_t nr; nr.o = p; nr.d = wi; nr.mint = isect.rayEpsilon; nr.maxt = INFINITY; nr.time = ray.time; nr.depth = ray.depth + 1; ray = nr; }
        if (bounces > 3) {
            float continueProbability = minf(.5f, sy(pathThroughput));
            if (draw_float(rng) > continueProbability) break;
            pathThroughput = sdivf(pathThroughput, continueProbability);
        }
        if (bounces == maxDepth) break;
        if (!scene_intersect(s, &ray, &isect, 0, st)) {
            if (specularBounce) for (int i = 0; i < nLights; ++i) L = sadd(L, smul(pathThroughput, light_Le(s, &s->d.lights[i], ray.d)));
            break;
        }
    }
    return L;
}

/* DirectLightingIntegrator::Li (integrators/directlighting.cpp:80-121) on top of SamplerRenderer::Li
 * (renderers/samplerrenderer.cpp:320-342).  SpecularReflect / SpecularTransmit (:111-118) sample BSDF_SPECULAR
 * lobes only (core/integrator.cpp:177-258); no material of this path has one, so both return black — but each
 * constructs a BSDFSample(rng) first (core/reflection.h:135-139): three RandomFloat() draws apiece from the tile's
 * generator whenever ray.depth + 1 < maxDepth, which the MT_REPLAY stream has to pay for. */
static rgb direct_Li(const orc_scene *s, ray_t ray, const dl_layout *Lt, const float *v, int maxDepth, draw_src *rng, uint64_t *st) {
    isect_t isect;
    rgb L = S(0.f);
    int nLights = s->d.n_lights;
    if (!scene_intersect(s, &ray, &isect, 0, st)) {
        for (int i = 0; i < nLights; ++i) L = sadd(L, light_Le(s, &s->d.lights[i], ray.d));
        return L;
    }
    bsdf_t bsdf;
    get_bsdf(s, &isect, &bsdf);
    v3 wo = vneg(ray.d);
    v3 p = bsdf.p, n = bsdf.nn;
    L = sadd(L, isect_Le(s, &isect, wo));
    if (nLights > 0) {
        if (Lt->integrator == HPT_INTEGRATOR_DIRECT_ALL) {        /* UniformSampleAllLights core/integrator.cpp:47-79 */
            rgb La = S(0.f);
            for (int i = 0; i < nLights; ++i) {
                int nSamples = Lt->ns[i];
                rgb Ld = S(0.f);
                for (int j = 0; j < nSamples; ++j) {
                    float ls[3], bs[3];
                    ls[0] = v[Lt->o2[2 * i] + 2 * j]; ls[1] = v[Lt->o2[2 * i] + 2 * j + 1]; ls[2] = v[Lt->o1[2 * i] + j];
                    bs[0] = v[Lt->o2[2 * i + 1] + 2 * j]; bs[1] = v[Lt->o2[2 * i + 1] + 2 * j + 1]; bs[2] = v[Lt->o1[2 * i + 1] + j];
                    Ld = sadd(Ld, estimate_direct(s, i, p, n, wo, isect.rayEpsilon, ray.time, &bsdf, ls, bs, st));
                }
                La = sadd(La, sdivf(Ld, (float)nSamples));
            }
            L = sadd(L, La);
        } else {                                                   /* UniformSampleOneLight :82-114 */
            int lightNum = (int)floorf(v[Lt->o1[1]] * nLights);
            if (lightNum > nLights - 1) lightNum = nLights - 1;
            float ls[3], bs[3];
            ls[0] = v[Lt->o2[0]]; ls[1] = v[Lt->o2[0] + 1]; ls[2] = v[Lt->o1[0]];
            bs[0] = v[Lt->o2[1]]; bs[1] = v[Lt->o2[1] + 1]; bs[2] = v[Lt->o1[2]];
            rgb Ld = estimate_direct(s, lightNum, p, n, wo, isect.rayEpsilon, ray.time, &bsdf, ls, bs, st);
            L = sadd(L, sscale(Ld, (float)nLights));
        }
    }
    if (ray.depth + 1 < maxDepth) for (int k = 0; k < 6; ++k) (void)draw_float(rng);
    return L;
}

/* PerspectiveCamera::GenerateRayDifferential, lensRadius == 0 (cameras/perspective.cpp:81-138) */
static void camera_ray(const hpt_camera *cam, const cam_sample *cs, ray_t *ray) {
    v3 Pcamera = xf_point(cam->raster_to_camera, V(cs->imageX, cs->imageY, 0));
    v3 dir = normalize(Pcamera);
    ray->o = V(0, 0, 0); ray->d = dir; ray->mint = 0.f; ray->maxt = INFINITY;
    if (cam->lens_radius > 0.) {
        float lensU, lensV;
        concentric_sample_disk(cs->lensU, cs->lensV, &lensU, &lensV);
        lensU *= cam->lens_radius; lensV *= cam->lens_radius;
        float ft = cam->focal_distance / ray->d.z;
        v3 Pfocus = ray_at(ray, ft);
        ray->o = V(lensU, lensV, 0.f);
        ray->d = normalize(vsub(Pfocus, ray->o));
    }
    ray->time = cs->time; ray->depth = 0;
    ray->o = xf_point(cam->camera_to_world, ray->o);
    ray->d = xf_vec(cam->camera_to_world, ray->d);
}

/* ImageFilm::AddSample, box filter (film/image.cpp:77-137) */
static void film_add(float *film, const hpt_render_desc *rd, float imageX, float imageY, rgb L) {
    float dimageX = imageX - 0.5f, dimageY = imageY - 0.5f;
    int x0 = (int)ceilf(dimageX - 0.5f), x1 = (int)floorf(dimageX + 0.5f);
    int y0 = (int)ceilf(dimageY - 0.5f), y1 = (int)floorf(dimageY + 0.5f);
    if (x0 < rd->x_start) x0 = rd->x_start;
    if (x1 > rd->x_start + rd->x_count - 1) x1 = rd->x_start + rd->x_count - 1;
    if (y0 < rd->y_start) y0 = rd->y_start;
    if (y1 > rd->y_start + rd->y_count - 1) y1 = rd->y_start + rd->y_count - 1;
    if ((x1 - x0) < 0 || (y1 - y0) < 0) return;
    float xyz[3]; /* RGBToXYZ spectrum.h:58-62 */
    xyz[0] = 0.412453f * L.c[0] + 0.357580f * L.c[1] + 0.180423f * L.c[2];
    xyz[1] = 0.212671f * L.c[0] + 0.715160f * L.c[1] + 0.072169f * L.c[2];
    xyz[2] = 0.019334f * L.c[0] + 0.119193f * L.c[1] + 0.950227f * L.c[2];
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            float *px = film + 4 * ((int64_t)(y - rd->y_start) * rd->x_count + (x - rd->x_start));
            const float filterWt = 1.f; /* BoxFilter::Evaluate filters/box.cpp:44-47 */
#ifdef _OPENMP
#pragma omp atomic
#endif
            px[0] += filterWt * xyz[0];
#ifdef _OPENMP
#pragma omp atomic
#endif
            px[1] += filterWt * xyz[1];
#ifdef _OPENMP
#pragma omp atomic
#endif
            px[2] += filterWt * xyz[2];
#ifdef _OPENMP
#pragma omp atomic
#endif
            px[3] += filterWt;
        }
}
/* radiance sanity (samplerrenderer.cpp:214-228) */
static rgb sanitize(rgb L, uint64_t *bad) {
    if (isnan(L.c[0]) || isnan(L.c[1]) || isnan(L.c[2])) { if (bad) (*bad)++; return S(0.f); }
    if (sy(L) < -1e-5) { if (bad) (*bad)++; return S(0.f); }
    if (isinf(sy(L))) { if (bad) (*bad)++; return S(0.f); }
    return L;
}
/* Sampler::ComputeSubWindow (core/sampler.cpp:55-74) */
static void compute_sub_window(int xs, int xe, int ys, int ye, int num, int count, int *nx0, int *nx1, int *ny0, int *ny1) {
    int dx = xe - xs, dy = ye - ys;
    int nx = count, ny = 1;
    while ((nx & 0x1) == 0 && 2 * dx * ny < dy * nx) { nx >>= 1; ny <<= 1; }
    int xo = num % nx, yo = num / nx;
    float tx0 = (float)xo / (float)nx, tx1 = (float)(xo + 1) / (float)nx;
    float ty0 = (float)yo / (float)ny, ty1 = (float)(yo + 1) / (float)ny;
    *nx0 = (int)floorf((1.f - tx0) * xs + tx0 * xe);
    *nx1 = (int)floorf((1.f - tx1) * xs + tx1 * xe);
    *ny0 = (int)floorf((1.f - ty0) * ys + ty0 * ye);
    *ny1 = (int)floorf((1.f - ty1) * ys + ty1 * ye);
}

int orc_render(const orc_scene *s, const hpt_camera *cam, const hpt_render_desc *rd, float *film, int nthreads, uint64_t *stats) {
    int spp = rd->spp;
    if (spp <= 0 || (spp & (spp - 1))) return HPT_E_INVALID;
    memset(film, 0, sizeof(float) * 4 * (size_t)rd->x_count * rd->y_count);
    /* sample extent == pixel extent for the box filter (film/image.cpp:157-166) */
    int xs = rd->x_start, xe = rd->x_start + rd->x_count, ys = rd->y_start, ye = rd->y_start + rd->y_count;
    uint64_t tot[6] = {0, 0, 0, 0, 0, 0};
    const int direct = rd->integrator != HPT_INTEGRATOR_PATH;
    if (rd->integrator < HPT_INTEGRATOR_PATH || rd->integrator > HPT_INTEGRATOR_DIRECT_ONE) return HPT_E_INVALID;
    dl_layout Lt; memset(&Lt, 0, sizeof(Lt));
    if (direct) dl_layout_init(&s->d, rd->integrator, &Lt);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    if (rd->sampler_mode == HPT_SAMPLER_MT_REPLAY) {
        int ntasks = rd->ntasks;
#pragma omp parallel
        {
            uint64_t st[6] = {0, 0, 0, 0, 0, 0};
            cam_sample *samples = (cam_sample *)malloc(sizeof(cam_sample) * (size_t)spp);
            float *buf = (float *)malloc(sizeof(float) * (direct ? dl_buf_floats(&Lt, spp) : (size_t)spp * (5 + N1D_ALL + 2 * N2D)));
            float *vals = direct ? (float *)malloc(sizeof(float) * (size_t)spp * (size_t)Lt.fps) : NULL;
            mt_rng rng;
#pragma omp for schedule(dynamic, 1)
            for (int task = 0; task < ntasks; ++task) {
                int x0, x1, y0, y1;
                compute_sub_window(xs, xe, ys, ye, task, ntasks, &x0, &x1, &y0, &y1);
                if (x0 == x1 || y0 == y1) continue;
                mt_seed(&rng, (uint32_t)task); /* RNG rng(taskNum) samplerrenderer.cpp:168 */
                draw_src ds; ds.mode = HPT_SAMPLER_MT_REPLAY; ds.mt = &rng; ds.key = 0; ds.counter = 0;
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) {
                        if (direct) ld_pixel_sample_mt_dl(&Lt, x, y, cam->shutter_open, cam->shutter_close, spp, samples, vals, buf, &rng);
                        else ld_pixel_sample_mt(x, y, cam->shutter_open, cam->shutter_close, spp, samples, buf, &rng);
                        for (int i = 0; i < spp; ++i) {
                            ray_t ray; camera_ray(cam, &samples[i], &ray);
                            rgb L = direct ? direct_Li(s, ray, &Lt, vals + (size_t)i * Lt.fps, rd->maxdepth, &ds, st)
                                           : path_Li(s, ray, &samples[i], rd->maxdepth, &ds, st);
                            L = sanitize(L, &st[5]);
                            film_add(film, rd, samples[i].imageX, samples[i].imageY, L);
                            st[0]++;
                        }
                    }
            }
            free(samples); free(buf); free(vals);
#pragma omp critical
            for (int k = 0; k < 6; ++k) tot[k] += st[k];
        }
    } else {
#pragma omp parallel
        {
            uint64_t st[6] = {0, 0, 0, 0, 0, 0};
            float *vals = direct ? (float *)malloc(sizeof(float) * (size_t)Lt.fps) : NULL;
#pragma omp for schedule(dynamic, 1)
            for (int y = ys; y < ye; ++y)
                for (int x = xs; x < xe; ++x) {
                    uint32_t pixelIndex = (uint32_t)y * (uint32_t)rd->xres + (uint32_t)x;
                    uint32_t pk = hash3(pixelIndex, rd->seed, 0x50495845u);
                    for (int i = 0; i < spp; ++i) {
                        cam_sample cs;
                        if (direct) ld_hash_sample_dl(&Lt, pixelIndex, rd->seed, x, y, cam->shutter_open, cam->shutter_close, (uint32_t)spp, (uint32_t)i, &cs, vals);
                        else ld_hash_sample(pixelIndex, rd->seed, x, y, cam->shutter_open, cam->shutter_close, (uint32_t)spp, (uint32_t)i, &cs);
                        draw_src ds; ds.mode = HPT_SAMPLER_LD_HASH; ds.mt = NULL; ds.key = hash3(pk, (uint32_t)i, 3u); ds.counter = 0;
                        ray_t ray; camera_ray(cam, &cs, &ray);
                        rgb L = direct ? direct_Li(s, ray, &Lt, vals, rd->maxdepth, &ds, st) : path_Li(s, ray, &cs, rd->maxdepth, &ds, st);
                        L = sanitize(L, &st[5]);
                        film_add(film, rd, cs.imageX, cs.imageY, L);
                        st[0]++;
                    }
                }
            free(vals);
#pragma omp critical
            for (int k = 0; k < 6; ++k) tot[k] += st[k];
        }
    }
    if (direct) dl_layout_free(&Lt);
    if (stats) for (int k = 0; k < 6; ++k) stats[k] = tot[k];
    return HPT_OK;
}

/* ---- function-level entry points ----------------------------------------------------------- */
int orc_intersect(const orc_scene *s, const float *rays, int64_t n, int anyhit, float *out_hit, int32_t *out_prim) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *r = rays + 8 * i;
        ray_t ray; ray.o = V(r[0], r[1], r[2]); ray.d = V(r[3], r[4], r[5]); ray.mint = r[6]; ray.maxt = r[7]; ray.time = 0; ray.depth = 0;
        isect_t is; memset(&is, 0, sizeof(is));
        int hit = scene_intersect(s, &ray, &is, anyhit, NULL);
        if (anyhit) { out_prim[i] = hit ? 0 : -1; out_hit[4 * i] = out_hit[4 * i + 1] = out_hit[4 * i + 2] = out_hit[4 * i + 3] = 0.f; }
        else if (hit) { out_prim[i] = (int32_t)is.prim; out_hit[4 * i] = is.t; out_hit[4 * i + 1] = is.b1; out_hit[4 * i + 2] = is.b2; out_hit[4 * i + 3] = is.rayEpsilon; }
        else { out_prim[i] = -1; out_hit[4 * i] = out_hit[4 * i + 1] = out_hit[4 * i + 2] = out_hit[4 * i + 3] = 0.f; }
    }
    return HPT_OK;
}
int orc_bsdf(const orc_scene *s, int material, const float *in, int64_t n, float *out) {
    if (material < 0 || material >= s->d.n_materials) return HPT_E_INVALID;
    const int flags = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_DIFFUSE | BSDF_GLOSSY;
    for (int64_t i = 0; i < n; ++i) {
        const float *q = in + 16 * i; float *o = out + 12 * i;
        v3 wo = V(q[0], q[1], q[2]), wi = V(q[3], q[4], q[5]);
        v3 nn = V(q[9], q[10], q[11]), dpdu = V(q[12], q[13], q[14]);
        bsdf_t b; bsdf_frame(&b, nn, dpdu, vmul(nn, q[15]));
        bsdf_add_material(&b, &s->d.materials[material]);
        rgb f = bsdf_f(s, &b, wo, wi, flags);
        float pdf = bsdf_pdf(&b, wo, wi, flags);
        v3 swi = V(0, 0, 0); float spdf = 0.f; int stype = 0;
        rgb sf = bsdf_sample_f(s, &b, wo, &swi, q[6], q[7], q[8], &spdf, flags, &stype);
        o[0] = f.c[0]; o[1] = f.c[1]; o[2] = f.c[2]; o[3] = pdf;
        o[4] = swi.x; o[5] = swi.y; o[6] = swi.z; o[7] = sf.c[0]; o[8] = sf.c[1]; o[9] = sf.c[2]; o[10] = spdf; o[11] = (float)stype;
    }
    return HPT_OK;
}
int orc_sampler(const hpt_render_desc *rd, int x, int y, float *out) {
    uint32_t pixelIndex = (uint32_t)y * (uint32_t)rd->xres + (uint32_t)x;
    for (int i = 0; i < rd->spp; ++i) {
        cam_sample cs;
        ld_hash_sample(pixelIndex, rd->seed, x, y, 0.f, 1.f, (uint32_t)rd->spp, (uint32_t)i, &cs);
        float *o = out + SAMPLE_FLOATS * i;
        o[0] = cs.imageX; o[1] = cs.imageY; o[2] = cs.lensU; o[3] = cs.lensV; o[4] = cs.time;
        for (int j = 0; j < N1D_PATH; ++j) o[5 + j] = cs.oneD[j];
        for (int j = 0; j < N2D; ++j) { o[17 + 2 * j] = cs.twoD[j][0]; o[18 + 2 * j] = cs.twoD[j][1]; }
    }
    return HPT_OK;
}
