// molann_math.h - per-frame arithmetic of the molann forward path, written once as
// __host__ __device__ code: the gfx950 kernels inline it, and the molann_selftest_* hooks compile the
// same source for the host so the CPU test-suite can check it against the oracle.
//
//   features      molann/ann.py:323-354   (angle / bond / dihedral / position)
//   kabsch        molann/ann.py:187-195   (rotation from the 3x3 covariance)
//   activations   molann/ann.py:37,64     (the module placed between the Linear layers)
#pragma once

#if !defined(__HIPCC_RTC__) // hipRTC provides the runtime declarations and libm itself
#include <hip/hip_runtime.h>
#include <math.h>
#endif

#define MOLANN_HD __host__ __device__ __forceinline__

namespace molann {

// item types of the expanded feature table (one output column group each)
enum ItemType : int {
    IT_ANGLE_COS = 0,   // ann.py:328-332, use_angle_value = False
    IT_BOND = 1,        // ann.py:334-336
    IT_DIHEDRAL_CS = 2, // ann.py:344-351, [cos, sin]
    IT_POSITION = 3,    // ann.py:353-354, one atom (3 columns); a k-atom feature is k items
    IT_ANGLE_VAL = 4,   // ann.py:330 acos
    IT_DIHEDRAL_VAL = 5 // ann.py:349 atan2
};

struct V3 {
    float x, y, z;
};

MOLANN_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MOLANN_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MOLANN_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
MOLANN_HD float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
MOLANN_HD V3 cross(V3 a, V3 b) {
    return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// ---- fast scalar helpers (device: one hardware instruction; host: libm) -----------------------
MOLANN_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
MOLANN_HD float fast_rsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
// sqrt(x) as x * rsq(x) (1-2 ulp): exact 0 at 0 like sqrtf, NaN for negative / NaN input
MOLANN_HD float fast_sqrt(float x) { return x == 0.0f ? 0.0f : x * fast_rsq(x); }
MOLANN_HD float fast_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return expf(x);
#endif
}

// ---- features ---------------------------------------------------------------------------------
// bond length |x2 - x1| (ann.py:335-336)
MOLANN_HD float feat_bond(V3 a0, V3 a1) {
    V3 r = a1 - a0;
    return fast_sqrt(dot(r, r));
}

// cosine of the angle at the SECOND atom (ann.py:324-328); no clamp, as in the reference
MOLANN_HD float feat_angle_cos(V3 a0, V3 a1, V3 a2) {
    V3 r21 = a0 - a1;
    V3 r23 = a2 - a1;
    // dot / (|r21| |r23|) with one reciprocal square root of the product of the squared lengths
    return dot(r21, r23) * fast_rsq(dot(r21, r21) * dot(r23, r23));
}

// unnormalised (cos, sin) of the dihedral 1-2-3-4 (ann.py:339-345)
MOLANN_HD void feat_dihedral_raw(V3 a0, V3 a1, V3 a2, V3 a3, float& c, float& s) {
    V3 r12 = a1 - a0;
    V3 r23 = a2 - a1;
    V3 r34 = a3 - a2;
    V3 n1 = cross(r12, r23);
    V3 n2 = cross(r23, r34);
    c = dot(n1, n2);
    s = dot(n1, r34) * fast_sqrt(dot(r23, r23));
}

// One item of the feature table on up to four atoms; writes 1..3 values, returns how many.
MOLANN_HD int eval_item(int type, V3 a0, V3 a1, V3 a2, V3 a3, float (&out)[3]) {
    switch (type) {
    case IT_ANGLE_COS:
        out[0] = feat_angle_cos(a0, a1, a2);
        return 1;
    case IT_ANGLE_VAL:
        out[0] = acosf(feat_angle_cos(a0, a1, a2)); // ann.py:330
        return 1;
    case IT_BOND:
        out[0] = feat_bond(a0, a1);
        return 1;
    case IT_DIHEDRAL_CS: {
        float c, s;
        feat_dihedral_raw(a0, a1, a2, a3, c, s);
        const float inv_radius = fast_rsq(fmaf(c, c, s * s)); // 1 / radius, ann.py:346
        out[0] = c * inv_radius;                              // ann.py:351 cos first, then sin
        out[1] = s * inv_radius;
        return 2;
    }
    case IT_DIHEDRAL_VAL: {
        float c, s;
        feat_dihedral_raw(a0, a1, a2, a3, c, s);
        out[0] = atan2f(s, c); // ann.py:349
        return 1;
    }
    default: // IT_POSITION
        out[0] = a0.x;
        out[1] = a0.y;
        out[2] = a0.z;
        return 3;
    }
}

MOLANN_HD int item_width(int type) { return type == IT_POSITION ? 3 : (type == IT_DIHEDRAL_CS ? 2 : 1); }
MOLANN_HD int item_atoms(int type) {
    return (type == IT_BOND) ? 2 : (type == IT_POSITION) ? 1 : (type == IT_ANGLE_COS || type == IT_ANGLE_VAL) ? 3 : 4;
}

// ---- activations ------------------------------------------------------------------------------
// tanh(x) = 1 - 2 / (1 + e^{2x}): five instructions, two of them transcendental.  Absolute error
// <= ~1.5e-7 everywhere (the 1 - 2r cancellation near 0 costs relative, not absolute, accuracy; the
// parity bar of this path is absolute, 1e-5).  +-inf and NaN behave like tanhf.
MOLANN_HD float act_tanh(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float t = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f); // e^{2x} = 2^{2x log2 e}
#else
    const float t = exp2f(x * 2.88539008177792681472f);
#endif
    return fmaf(-2.0f, fast_rcp(1.0f + t), 1.0f);
}

MOLANN_HD float act_sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

MOLANN_HD float apply_activation(int act, float v) {
    switch (act) {
    case 0: return act_tanh(v);                       // torch.nn.Tanh (the default, ann.py:37)
    case 1: return fmaxf(v, 0.0f) + (v != v ? v : 0.0f); // ReLU, NaN-propagating like torch
    case 2: return act_sigmoid(v);                    // Sigmoid
    case 3: return v;                                 // Identity
    case 4: return v > 0.0f ? v : expm1f(v);          // ELU(alpha=1)
    case 5: return v * act_sigmoid(v);                // SiLU
    case 6: return v > 20.0f ? v : log1pf(expf(v));   // Softplus(beta=1, threshold=20)
    case 7: return v > 0.0f ? v : 0.01f * v;          // LeakyReLU(0.01)
    case 8: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); // GELU (erf)
    default: return v;
    }
}

// ---- Kabsch rotation --------------------------------------------------------------------------
// Input: H[a][b] = sum_i p_i[a] * ref_i[b] over the align atoms (p centred frame coordinates, ref the
// centred reference) = `prod` of ann.py:187, accumulated in fp64 by the callers (the products of fp32
// coordinates are exact in fp64, so H carries no rounding of its own), and
// e0 >= (sum|p|^2 + sum|ref|^2)/2, an upper bound on the largest eigenvalue used as the Newton start.
// Output: R (row-major) with aligned_row = p_row . R, equal to U diag(1,1,sign det(U Vh)) Vh of
// ann.py:188-195.
//
// Method (not the reference's LAPACK SVD): the optimal proper rotation is the top eigenvector of
// Horn's symmetric 4x4 matrix K(H) read as a unit quaternion.  Its eigenvalues are
// {s1+s2+d*s3, s1-s2-d*s3, -s1+s2-d*s3, -s1-s2+d*s3} (s = singular values, d = sign det H), so the top
// one is simple exactly when the rotation is well defined (s2 + d*s3 > 0): rank-2 covariances (planar
// reference, three align atoms) and reflections need no special case.  lambda_max by Newton on the
// characteristic quartic from above (monotone), eigenvector = the column of adj(K - lambda I) with the
// largest diagonal entry.  fp64 throughout: ~250 flops per frame, against ~60 flops to form H.
MOLANN_HD float tfma(float a, float b, float c) { return fmaf(a, b, c); }
MOLANN_HD double tfma(double a, double b, double c) { return fma(a, b, c); }
MOLANN_HD float tabs(float a) { return fabsf(a); }
MOLANN_HD double tabs(double a) { return fabs(a); }

// T = double: what every plan with position items (and AlignmentLayer.forward) uses.  T = float: the same
// algorithm at the reference's own precision (its SVD runs in fp32) for plans whose items are all invariant
// under rigid motion (bond / angle / dihedral): their outputs do not depend on how accurate R is, only on R
// being a proper rotation, which the normalised quaternion guarantees in either precision.
template <typename T, typename RT = float>
MOLANN_HD void kabsch_rotation_t(const T (&H)[9], T e0, RT (&R)[9]) {
    constexpr bool F32 = sizeof(T) == 4;
    const T tiny = F32 ? (T)1e-30f : (T)1e-30, huge = F32 ? (T)1e30f : (T)1e30;
    T fro2 = (T)0;
#pragma unroll
    for (int i = 0; i < 9; ++i) fro2 = tfma(H[i], H[i], fro2);
    if (!(fro2 > tiny) || !(fro2 < huge)) { // zero / non-finite covariance: no rotation
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? (RT)1 : (RT)0;
        return;
    }
    // scale so that |h|_F ~= 1: the rotation does not depend on the scale, so an fp32 rsqrt is enough
#if defined(__HIP_DEVICE_COMPILE__)
    const T s = (T)__builtin_amdgcn_rsqf((float)fro2);
#else
    const T s = (T)(1.0f / sqrtf((float)fro2));
#endif
    const T hxx = H[0] * s, hxy = H[1] * s, hxz = H[2] * s;
    const T hyx = H[3] * s, hyy = H[4] * s, hyz = H[5] * s;
    const T hzx = H[6] * s, hzy = H[7] * s, hzz = H[8] * s;

    const T k00 = hxx + hyy + hzz, k01 = hyz - hzy, k02 = hzx - hxz, k03 = hxy - hyx;
    const T k11 = hxx - hyy - hzz, k12 = hxy + hyx, k13 = hzx + hxz;
    const T k22 = -hxx + hyy - hzz, k23 = hyz + hzy;
    const T k33 = -hxx - hyy + hzz;

    // characteristic polynomial l^4 + c2 l^2 + c1 l + c0 (trace K = 0)
    const T c2 = (T)-2 * (hxx * hxx + hxy * hxy + hxz * hxz + hyx * hyx + hyy * hyy + hyz * hyz + hzx * hzx +
                              hzy * hzy + hzz * hzz);
    const T det_h = hxx * (hyy * hzz - hyz * hzy) - hxy * (hyx * hzz - hyz * hzx) + hxz * (hyx * hzy - hyy * hzx);
    const T c1 = (T)-8 * det_h;
    T c0;
    {
        const T s0 = k00 * k11 - k01 * k01, s1 = k00 * k12 - k01 * k02, s2 = k00 * k13 - k01 * k03;
        const T s3 = k01 * k12 - k11 * k02, s4 = k01 * k13 - k11 * k03, s5 = k02 * k13 - k12 * k03;
        const T d5 = k22 * k33 - k23 * k23, d4 = k12 * k33 - k13 * k23, d3 = k12 * k23 - k13 * k22;
        const T d2 = k02 * k33 - k03 * k23, d1 = k02 * k23 - k03 * k22, d0 = k02 * k13 - k03 * k12;
        c0 = s0 * d5 - s1 * d4 + s2 * d3 + s3 * d2 - s4 * d1 + s5 * d0;
    }

    // Newton from above; lambda_max <= s1+s2+s3 <= sqrt(3) |h|_F and <= e0 * s
    constexpr T sqrt3 = (T)1.7320508075688772;
    T lam0 = e0 * s < sqrt3 ? e0 * s : sqrt3;
    if (!(lam0 > (T)0)) lam0 = sqrt3;
    // A frame that resembles the reference starts within ~1e-2 of the root (e0 is then tight) and Newton converges
    // quadratically: NFIX unconditional steps - straight-line code, no exec-mask bookkeeping, nothing for the compiler
    // to split into blocks - leave the last step below the tolerance.  Whatever has not converged by then (or went
    // non-finite) takes the guarded loop below; a wave skips it when all its frames are done.
    constexpr int NFIX = F32 ? 4 : 5;
    const T tol = F32 ? (T)1e-6f : (T)1e-14;
    T lam = lam0, step = (T)0;
#pragma unroll
    for (int it = 0; it < NFIX; ++it) {
        const T l2 = lam * lam;
        const T p = tfma(l2 + c2, l2, tfma(c1, lam, c0));
        const T dp = tfma(tfma((T)4, l2, (T)2 * c2), lam, c1);
        step = p * (T)fast_rcp((float)dp);   // Newton corrects itself: an fp32-accurate reciprocal is enough
        lam = lam - step;
    }
    bool done = (step == step) && (tabs(lam) < (T)4) && !(tabs(step) > tol * tabs(lam));
#if defined(__HIP_DEVICE_COMPILE__)
    if (!__all(done)) {
#else
    if (!done) {
#endif
        if (!done && !((lam == lam) && lam > (T)0 && lam <= lam0)) lam = lam0;   // left the monotone path: start again
        // bounded: every lane reaches the exit.  fp32: the residual's own rounding (~1e-7 / p'(lam)) keeps an
        // ill-conditioned frame from ever meeting the step test, so the cap is what ends it there.
#pragma unroll 1
        for (int it = 0; it < (F32 ? 12 : 48); ++it) {
            if (!done) {
                const T l2 = lam * lam;
                const T p = (l2 + c2) * l2 + c1 * lam + c0;
                const T dp = ((T)4 * l2 + (T)2 * c2) * lam + c1;
                const T st = p * (T)fast_rcp((float)dp);
                const T nl = lam - st;
                const bool finite = (st == st) && (tabs(nl) < (T)4);
                if (finite) lam = nl;
                if (!finite || !(tabs(st) > tol * tabs(nl))) done = true;
            }
#if defined(__HIP_DEVICE_COMPILE__)
            if (__all(done)) break; // wave-uniform exit
#else
            if (done) break;
#endif
        }
    }

    // adj(K - lam I) = const * q q^T
    const T m00 = k00 - lam, m11 = k11 - lam, m22 = k22 - lam, m33 = k33 - lam;
    const T s0 = m00 * m11 - k01 * k01, s1 = m00 * k12 - k01 * k02, s2 = m00 * k13 - k01 * k03;
    const T s3 = k01 * k12 - m11 * k02, s4 = k01 * k13 - m11 * k03, s5 = k02 * k13 - k12 * k03;
    const T d5 = m22 * m33 - k23 * k23, d4 = k12 * m33 - k13 * k23, d3 = k12 * k23 - k13 * m22;
    const T d2 = k02 * m33 - k03 * k23, d1 = k02 * k23 - k03 * m22;
    const T a00 = m11 * d5 - k12 * d4 + k13 * d3;
    const T a01 = -k01 * d5 + k02 * d4 - k03 * d3;
    const T a02 = k13 * s5 - k23 * s4 + m33 * s3;
    const T a03 = -k12 * s5 + m22 * s4 - k23 * s3;
    const T a11 = m00 * d5 - k02 * d2 + k03 * d1;
    const T a12 = -k03 * s5 + k23 * s2 - m33 * s1;
    const T a13 = k02 * s5 - m22 * s2 + k23 * s1;
    const T a22 = k03 * s4 - k13 * s2 + m33 * s0;
    const T a23 = -k02 * s4 + k12 * s2 - k23 * s0;
    const T a33 = k02 * s3 - k12 * s1 + m22 * s0;

    // column with the largest |diagonal| (q_j^2 >= 1/4 there)
    T q0 = a00, q1 = a01, q2 = a02, q3 = a03, best = tabs(a00);
    if (tabs(a11) > best) { best = tabs(a11); q0 = a01; q1 = a11; q2 = a12; q3 = a13; }
    if (tabs(a22) > best) { best = tabs(a22); q0 = a02; q1 = a12; q2 = a22; q3 = a23; }
    if (tabs(a33) > best) { best = tabs(a33); q0 = a03; q1 = a13; q2 = a23; q3 = a33; }
    const T n2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    if (!(n2 > tiny) || !(n2 < huge)) { // K - lam I numerically zero: degenerate input
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? (RT)1 : (RT)0;
        return;
    }
    // 1/n2: fp32 reciprocal seed + one Newton step in fp64 (relative error ~1e-14)
    T inv = (T)fast_rcp((float)n2);
    inv = inv * ((T)2 - n2 * inv);
    if (sizeof(RT) == 8) inv = inv * ((T)2 - n2 * inv);   // rotation wanted in double: one more step (1e-14 -> rounding)
    const T ww = q0 * q0 * inv, xx = q1 * q1 * inv, yy = q2 * q2 * inv, zz = q3 * q3 * inv;
    const T wx = q0 * q1 * inv, wy = q0 * q2 * inv, wz = q0 * q3 * inv;
    const T xy = q1 * q2 * inv, xz = q1 * q3 * inv, yz = q2 * q3 * inv;
    // Q (column convention, maps frame -> reference); R = Q^T for row vectors
    R[0] = (RT)(ww + xx - yy - zz);
    R[1] = (RT)((T)2 * (xy + wz));
    R[2] = (RT)((T)2 * (xz - wy));
    R[3] = (RT)((T)2 * (xy - wz));
    R[4] = (RT)(ww - xx + yy - zz);
    R[5] = (RT)((T)2 * (yz + wx));
    R[6] = (RT)((T)2 * (xz + wy));
    R[7] = (RT)((T)2 * (yz - wx));
    R[8] = (RT)(ww - xx - yy + zz);
}

MOLANN_HD void kabsch_rotation(const double (&H)[9], double e0, float (&R)[9]) { kabsch_rotation_t<double>(H, e0, R); }
MOLANN_HD void kabsch_rotation_f32(const float (&H)[9], float e0, float (&R)[9]) { kabsch_rotation_t<float>(H, e0, R); }

// y = p . R  (row vector times row-major R)
MOLANN_HD V3 rotate(V3 p, const float (&R)[9]) {
    return v3(fmaf(p.z, R[6], fmaf(p.y, R[3], p.x * R[0])), fmaf(p.z, R[7], fmaf(p.y, R[4], p.x * R[1])),
              fmaf(p.z, R[8], fmaf(p.y, R[5], p.x * R[2])));
}

// The atoms of one item as the feature sees them in the ALIGNED frame y = ((p - c0) - dl) R (ann.py:197, 565).  A position
// item needs y itself.  Bond / angle / dihedral items are translation invariant: they are evaluated about the item's second
// atom, y_k - y_1 = (p_k - p_1) R, so the short vectors between neighbouring atoms are formed from the INPUT coordinates
// (exact up to the rounding of a vector of a few Angstrom) rather than from aligned coordinates that each carry the
// rounding of their distance to the centroid - 1e-5 A at 100 A, which is what a 5000-atom frame's outer atoms have, and
// what an ill-conditioned dihedral then amplifies.  Same values in exact arithmetic; used by the large-frame kernels.
MOLANN_HD void align_item_atoms(int type, V3& p0, V3& p1, V3& p2, V3& p3, V3 c0, V3 dl, const float (&R)[9]) {
    if (type == IT_POSITION) {
        p0 = rotate((p0 - c0) - dl, R);
        return;
    }
    p0 = rotate(p0 - p1, R);
    p2 = rotate(p2 - p1, R);
    p3 = rotate(p3 - p1, R);
    p1 = v3(0.f, 0.f, 0.f);
}

// =================================================================================================
// float64 instantiation of the forward (the reference follows x.dtype: `model.double()(x.double())`, ann.py:187-197,
// 323-354).  Plain double arithmetic with libm - no fast approximations: the bar is 1e-10 against the reference's
// own float64 run.
// =================================================================================================
struct V3d {
    double x, y, z;
};
MOLANN_HD V3d v3d(double x, double y, double z) { V3d r; r.x = x; r.y = y; r.z = z; return r; }
MOLANN_HD V3d operator-(V3d a, V3d b) { return v3d(a.x - b.x, a.y - b.y, a.z - b.z); }
MOLANN_HD double dot(V3d a, V3d b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
MOLANN_HD V3d cross(V3d a, V3d b) {
    return v3d(fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x)));
}
MOLANN_HD V3d rotate(V3d p, const double (&R)[9]) {
    return v3d(fma(p.z, R[6], fma(p.y, R[3], p.x * R[0])), fma(p.z, R[7], fma(p.y, R[4], p.x * R[1])),
               fma(p.z, R[8], fma(p.y, R[5], p.x * R[2])));
}
// eval_item in double (same item types, same column order)
MOLANN_HD int eval_item_f64(int type, V3d a0, V3d a1, V3d a2, V3d a3, double (&out)[3]) {
    switch (type) {
    case IT_ANGLE_COS:
    case IT_ANGLE_VAL: {
        const V3d r21 = a0 - a1, r23 = a2 - a1;
        const double c = dot(r21, r23) / (sqrt(dot(r21, r21)) * sqrt(dot(r23, r23)));   // ann.py:324-328, no clamp
        out[0] = type == IT_ANGLE_VAL ? acos(c) : c;
        return 1;
    }
    case IT_BOND: {
        const V3d r = a1 - a0;
        out[0] = sqrt(dot(r, r));
        return 1;
    }
    case IT_DIHEDRAL_CS:
    case IT_DIHEDRAL_VAL: {
        const V3d r12 = a1 - a0, r23 = a2 - a1, r34 = a3 - a2;
        const V3d n1 = cross(r12, r23), n2 = cross(r23, r34);
        const double c = dot(n1, n2), sn = dot(n1, r34) * sqrt(dot(r23, r23));
        if (type == IT_DIHEDRAL_VAL) { out[0] = atan2(sn, c); return 1; }
        const double radius = sqrt(fma(c, c, sn * sn));                              // ann.py:346
        out[0] = c / radius;
        out[1] = sn / radius;
        return 2;
    }
    default:
        out[0] = a0.x; out[1] = a0.y; out[2] = a0.z;
        return 3;
    }
}
MOLANN_HD double apply_activation_f64(int act, double v) {
    switch (act) {
    case 0: return tanh(v);
    case 1: return v > 0.0 ? v : (v != v ? v : 0.0);
    case 2: return 1.0 / (1.0 + exp(-v));
    case 3: return v;
    case 4: return v > 0.0 ? v : expm1(v);
    case 5: return v / (1.0 + exp(-v));
    case 6: return v > 20.0 ? v : log1p(exp(v));
    case 7: return v > 0.0 ? v : 0.01 * v;
    case 8: return 0.5 * v * (1.0 + erf(v * 0.70710678118654752));
    default: return v;
    }
}

// =================================================================================================
// reverse mode (backward of the path; the reference relies on torch autograd for all of it)
// =================================================================================================
MOLANN_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
MOLANN_HD void axpy(V3& acc, float s, V3 a) { acc.x = fmaf(s, a.x, acc.x); acc.y = fmaf(s, a.y, acc.y); acc.z = fmaf(s, a.z, acc.z); }

// Gradient of one feature item with respect to its (aligned) atoms: g[] is dL/d(output columns of the item),
// ga0..ga3 are ACCUMULATED into.  Mirrors eval_item; formulas are the reverse-mode of ann.py:323-354.
MOLANN_HD void eval_item_backward(int type, V3 a0, V3 a1, V3 a2, V3 a3, const float (&g)[3], V3& ga0, V3& ga1, V3& ga2,
                                  V3& ga3) {
    switch (type) {
    case IT_BOND: {
        const V3 r = a1 - a0;
        const float d2 = dot(r, r);
        const float inv = d2 > 0.0f ? fast_rsq(d2) : 0.0f; // |r| = 0: subgradient 0
        axpy(ga1, g[0] * inv, r);
        axpy(ga0, -g[0] * inv, r);
        return;
    }
    case IT_ANGLE_COS:
    case IT_ANGLE_VAL: {
        const V3 u = a0 - a1, v = a2 - a1;
        const float uu = dot(u, u), vv = dot(v, v), uv = dot(u, v);
        const float inv_uv = fast_rsq(uu * vv); // 1 / (|u||v|)
        const float c = uv * inv_uv;
        float gc = g[0];
        if (type == IT_ANGLE_VAL) gc = -g[0] * fast_rsq(fmaxf(1.0f - c * c, 1e-30f)); // d acos(c) = -dc / sqrt(1 - c^2)
        // dc/du = v/(|u||v|) - c u/|u|^2 ,  dc/dv = u/(|u||v|) - c v/|v|^2
        V3 gu = v3(0.f, 0.f, 0.f), gv = v3(0.f, 0.f, 0.f);
        axpy(gu, gc * inv_uv, v); axpy(gu, -gc * c * fast_rcp(uu), u);
        axpy(gv, gc * inv_uv, u); axpy(gv, -gc * c * fast_rcp(vv), v);
        ga0 = ga0 + gu;
        ga2 = ga2 + gv;
        ga1 = ga1 - (gu + gv);
        return;
    }
    case IT_DIHEDRAL_CS:
    case IT_DIHEDRAL_VAL: {
        const V3 r12 = a1 - a0, r23 = a2 - a1, r34 = a3 - a2;
        const V3 n1 = cross(r12, r23), n2 = cross(r23, r34);
        const float L2 = dot(r23, r23);
        const float L = fast_sqrt(L2);
        const float n1r34 = dot(n1, r34);
        const float C = dot(n1, n2), S = n1r34 * L;
        const float rad2 = fmaf(C, C, S * S);
        float gC, gS;
        if (type == IT_DIHEDRAL_CS) { // outputs (C, S) / rad
            const float inv_rad = fast_rsq(rad2);
            const float proj = (g[0] * C + g[1] * S) * inv_rad * inv_rad * inv_rad;
            gC = g[0] * inv_rad - proj * C;
            gS = g[1] * inv_rad - proj * S;
        } else { // atan2(S, C): d phi = (C dS - S dC) / rad^2
            const float inv_rad2 = fast_rcp(rad2);
            gC = -g[0] * S * inv_rad2;
            gS = g[0] * C * inv_rad2;
        }
        // C = n1.n2 ; S = (n1.r34) L
        V3 gn1 = gC * n2; axpy(gn1, gS * L, r34);
        const V3 gn2 = gC * n1;
        V3 gr34 = (gS * L) * n1;
        V3 gr23 = (L > 0.0f ? gS * n1r34 * fast_rcp(L) : 0.0f) * r23;
        // n1 = r12 x r23 ; n2 = r23 x r34      (n = a x b  =>  ga = b x gn , gb = gn x a)
        const V3 gr12 = cross(r23, gn1);
        gr23 = gr23 + cross(gn1, r12) + cross(r34, gn2);
        gr34 = gr34 + cross(gn2, r23);
        ga0 = ga0 - gr12;
        ga1 = ga1 + (gr12 - gr23);
        ga2 = ga2 + (gr23 - gr34);
        ga3 = ga3 + gr34;
        return;
    }
    default: // IT_POSITION
        ga0 = ga0 + v3(g[0], g[1], g[2]);
        return;
    }
}

// Derivative of the activation expressed with its OUTPUT h (and input z where the output is not enough).
MOLANN_HD float act_derivative(int act, float z, float h) {
    switch (act) {
    case 0: return fmaf(-h, h, 1.0f);          // tanh' = 1 - tanh^2
    case 1: return z > 0.0f ? 1.0f : 0.0f;     // ReLU
    case 2: return h * (1.0f - h);             // sigmoid' = s (1 - s)
    case 3: return 1.0f;                       // identity
    case 5: { const float s = act_sigmoid(z); return s * fmaf(z, 1.0f - s, 1.0f); } // SiLU' = s (1 + z (1 - s))
    case 7: return z > 0.0f ? 1.0f : 0.01f;    // LeakyReLU
    default: return 1.0f;
    }
}

// Backward of kabsch_rotation: given H, the rotation R it produced and G_R = dL/dR, returns G_H = dL/dH.
// With S = R^T H (symmetric at the optimum) a perturbation dH turns R by dR = R [w]x where
// (tr(S) I - S) w = vee(R^T dH - dH^T R); hence G_H = R [n]x, n = (tr(S) I - S)^-1 vee(M - M^T), M = R^T G_R.
// ([v]x = cross-product matrix of v.)  3x3 products and one 3x3 symmetric solve in T: double, or float where the forward's
// Kabsch is float as well (plans whose items are all invariant under rigid motion).
template <typename T, typename RT = float>
MOLANN_HD void kabsch_rotation_backward_t(const T (&H)[9], const RT (&R)[9], const RT (&GR)[9], RT (&GH)[9]) {
    T S[9], M[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            T s = (T)0, m = (T)0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { s = tfma((T)R[3 * k + a], H[3 * k + b], s); m = tfma((T)R[3 * k + a], (T)GR[3 * k + b], m); }
            S[3 * a + b] = s;
            M[3 * a + b] = m;
        }
    const T s01 = (T)0.5 * (S[1] + S[3]), s02 = (T)0.5 * (S[2] + S[6]), s12 = (T)0.5 * (S[5] + S[7]);
    const T tr = S[0] + S[4] + S[8];
    // B = tr I - S  (symmetric)
    const T b00 = tr - S[0], b11 = tr - S[4], b22 = tr - S[8], b01 = -s01, b02 = -s02, b12 = -s12;
    const T m0 = M[7] - M[5], m1 = M[2] - M[6], m2 = M[3] - M[1]; // vee(M - M^T)
    // n = B^-1 m by the adjugate
    const T c00 = b11 * b22 - b12 * b12, c01 = b02 * b12 - b01 * b22, c02 = b01 * b12 - b02 * b11;
    const T c11 = b00 * b22 - b02 * b02, c12 = b01 * b02 - b00 * b12, c22 = b00 * b11 - b01 * b01;
    const T det = b00 * c00 + b01 * c01 + b02 * c02;
    constexpr T tiny = sizeof(T) == 8 ? (T)1e-300 : (T)1e-30;
    const T inv = (det > tiny || det < -tiny) ? (T)1 / det : (T)0; // ill-defined rotation: no gradient through R
    const T n0 = (c00 * m0 + c01 * m1 + c02 * m2) * inv;
    const T n1 = (c01 * m0 + c11 * m1 + c12 * m2) * inv;
    const T n2 = (c02 * m0 + c12 * m1 + c22 * m2) * inv;
    // G_H = R [n]x ,  [n]x = [[0,-n2,n1],[n2,0,-n0],[-n1,n0,0]]
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T r0 = R[3 * a], r1 = R[3 * a + 1], r2 = R[3 * a + 2];
        GH[3 * a + 0] = (RT)(r1 * n2 - r2 * n1);
        GH[3 * a + 1] = (RT)(r2 * n0 - r0 * n2);
        GH[3 * a + 2] = (RT)(r0 * n1 - r1 * n0);
    }
}

MOLANN_HD void kabsch_rotation_backward(const double (&H)[9], const float (&R)[9], const float (&GR)[9], float (&GH)[9]) {
    kabsch_rotation_backward_t<double>(H, R, GR, GH);
}

// ---- float64 reverse mode of the feature items (the reference differentiates its float64 forward with autograd too) ----
MOLANN_HD V3d operator+(V3d a, V3d b) { return v3d(a.x + b.x, a.y + b.y, a.z + b.z); }
MOLANN_HD V3d operator*(double s, V3d a) { return v3d(s * a.x, s * a.y, s * a.z); }
MOLANN_HD void axpy(V3d& acc, double s, V3d a) { acc.x = fma(s, a.x, acc.x); acc.y = fma(s, a.y, acc.y); acc.z = fma(s, a.z, acc.z); }

// eval_item_backward in double: the same formulas with exact divisions and square roots
MOLANN_HD void eval_item_backward_f64(int type, V3d a0, V3d a1, V3d a2, V3d a3, const double (&g)[3], V3d& ga0, V3d& ga1, V3d& ga2,
                                      V3d& ga3) {
    switch (type) {
    case IT_BOND: {
        const V3d r = a1 - a0;
        const double d2 = dot(r, r);
        const double inv = d2 > 0.0 ? 1.0 / sqrt(d2) : 0.0;
        axpy(ga1, g[0] * inv, r);
        axpy(ga0, -g[0] * inv, r);
        return;
    }
    case IT_ANGLE_COS:
    case IT_ANGLE_VAL: {
        const V3d u = a0 - a1, v = a2 - a1;
        const double uu = dot(u, u), vv = dot(v, v), uv = dot(u, v);
        const double inv_uv = 1.0 / (sqrt(uu) * sqrt(vv));
        const double c = uv * inv_uv;
        double gc = g[0];
        if (type == IT_ANGLE_VAL) gc = -g[0] / sqrt(1.0 - c * c);
        V3d gu = v3d(0., 0., 0.), gv = v3d(0., 0., 0.);
        axpy(gu, gc * inv_uv, v); axpy(gu, -gc * c / uu, u);
        axpy(gv, gc * inv_uv, u); axpy(gv, -gc * c / vv, v);
        ga0 = ga0 + gu;
        ga2 = ga2 + gv;
        ga1 = ga1 - (gu + gv);
        return;
    }
    case IT_DIHEDRAL_CS:
    case IT_DIHEDRAL_VAL: {
        const V3d r12 = a1 - a0, r23 = a2 - a1, r34 = a3 - a2;
        const V3d n1 = cross(r12, r23), n2 = cross(r23, r34);
        const double L2 = dot(r23, r23);
        const double L = sqrt(L2);
        const double n1r34 = dot(n1, r34);
        const double C = dot(n1, n2), S = n1r34 * L;
        const double rad2 = fma(C, C, S * S);
        double gC, gS;
        if (type == IT_DIHEDRAL_CS) {
            const double inv_rad = 1.0 / sqrt(rad2);
            const double proj = (g[0] * C + g[1] * S) * inv_rad * inv_rad * inv_rad;
            gC = g[0] * inv_rad - proj * C;
            gS = g[1] * inv_rad - proj * S;
        } else {
            gC = -g[0] * S / rad2;
            gS = g[0] * C / rad2;
        }
        V3d gn1 = gC * n2; axpy(gn1, gS * L, r34);
        const V3d gn2 = gC * n1;
        V3d gr34 = (gS * L) * n1;
        V3d gr23 = (L > 0.0 ? gS * n1r34 / L : 0.0) * r23;
        const V3d gr12 = cross(r23, gn1);
        gr23 = gr23 + cross(gn1, r12) + cross(r34, gn2);
        gr34 = gr34 + cross(gn2, r23);
        ga0 = ga0 - gr12;
        ga1 = ga1 + (gr12 - gr23);
        ga2 = ga2 + (gr23 - gr34);
        ga3 = ga3 + gr34;
        return;
    }
    default: // IT_POSITION
        ga0 = ga0 + v3d(g[0], g[1], g[2]);
        return;
    }
}

} // namespace molann
