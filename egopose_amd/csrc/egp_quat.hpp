// Device quaternion algebra for the reward / observation kernels, (w,x,y,z) order.
// Semantics follow the reference's vendored transformations.py and utils/math.py (file:line per
// function); written for registers -- no arrays that could land in scratch.
#pragma once
#include <hip/hip_runtime.h>

namespace egp {

template <typename T> struct Q4 { T w, x, y, z; };
template <typename T> struct V3 { T x, y, z; };

template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ float t_sqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }
template <typename T> __device__ __forceinline__ T t_acos(T v);
template <> __device__ __forceinline__ float t_acos<float>(float v) { return acosf(v); }
template <> __device__ __forceinline__ double t_acos<double>(double v) { return acos(v); }
template <typename T> __device__ __forceinline__ T t_exp(T v);
template <> __device__ __forceinline__ float t_exp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double t_exp<double>(double v) { return exp(v); }
template <typename T> __device__ __forceinline__ T t_pow(T a, T b);
template <> __device__ __forceinline__ float t_pow<float>(float a, float b) { return powf(a, b); }
template <> __device__ __forceinline__ double t_pow<double>(double a, double b) { return pow(a, b); }
template <typename T> __device__ __forceinline__ void t_sincos(T v, T *s, T *c);
template <> __device__ __forceinline__ void t_sincos<float>(float v, float *s, float *c) { sincosf(v, s, c); }
// float64 sincos for the kernels' joint angles and half-angles: Cody-Waite reduction by pi/2 in two FMAs (exact enough while
// |n| < 2^18) and the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4] (< 1 ulp each) -- ~35 instructions where the
// library routine, which also carries a Payne-Hanek path for huge arguments, runs ~120. The reward kernel is float64-VALU
// bound (VALUBusy 87 %) and spends 12 of these per body lane. Arguments beyond 2^18 (and NaN / inf) take the library routine.
__device__ __forceinline__ void egp_sincos_f64(double x, double *s, double *c) {
    if (!(fabs(x) < 262144.0)) { sincos(x, s, c); return; }
    const double n = rint(x * 6.36619772367581382433e-01);                 // x * 2 / pi
    double r = fma(-n, 1.57079632679489655800e+00, x);                      // pi/2 = hi + lo
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                 -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double sr = fma(r * z, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                        2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cr = 1.0 - fma(0.5, z, -(z * z) * pc);
    const int q = (int)n & 3;
    const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}
template <> __device__ __forceinline__ void t_sincos<double>(double v, double *s, double *c) { egp_sincos_f64(v, s, c); }

// quaternion_multiply  (utils/transformation.py:1379-1393)
template <typename T> __device__ __forceinline__ Q4<T> qmul(const Q4<T> &a, const Q4<T> &b) {
    Q4<T> r;
    r.w = -a.x * b.x - a.y * b.y - a.z * b.z + a.w * b.w;
    r.x = a.x * b.w + a.y * b.z - a.z * b.y + a.w * b.x;
    r.y = -a.x * b.z + a.y * b.w + a.z * b.x + a.w * b.y;
    r.z = a.x * b.y - a.y * b.x + a.z * b.w + a.w * b.z;
    return r;
}

// quaternion_inverse: conjugate / (q.q)  (utils/transformation.py:1410-1421)
// (one division and four products instead of four divisions -- a float64 division is ~15 instructions around a quarter-rate
//  v_rcp_f64, and the reward kernel, float64-VALU bound, inverted two quaternions per body lane; <= 1 ulp from the quotient)
template <typename T> __device__ __forceinline__ Q4<T> qinv(const Q4<T> &q) {
    const T n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    Q4<T> r;
    if (sizeof(T) == 8) {
        const T inv = T(1) / n;
        r.w = q.w * inv; r.x = -q.x * inv; r.y = -q.y * inv; r.z = -q.z * inv;
    } else {                       // float32: the quotient itself (a float32 division is cheap, and the product costs a second rounding)
        r.w = q.w / n; r.x = -q.x / n; r.y = -q.y / n; r.z = -q.z / n;
    }
    return r;
}

// quaternion_from_euler(ai, aj, ak, 'sxyz')  (utils/transformation.py:1194-1248)
template <typename T> __device__ __forceinline__ Q4<T> q_from_euler_sxyz(T ai, T aj, T ak) {
    T si, ci, sj, cj, sk, ck;
    t_sincos<T>(ai * T(0.5), &si, &ci);
    t_sincos<T>(aj * T(0.5), &sj, &cj);
    t_sincos<T>(ak * T(0.5), &sk, &ck);
    const T cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
    Q4<T> q;
    q.w = cj * cc + sj * ss;
    q.x = cj * sc - sj * cs;
    q.y = cj * ss + sj * cc;
    q.z = cj * cs - sj * sc;
    return q;
}

// get_heading_q: zero x,y and renormalise  (utils/math.py:62-67)
template <typename T> __device__ __forceinline__ Q4<T> heading_q(const Q4<T> &q) {
    const T n = t_sqrt<T>(q.w * q.w + q.z * q.z);
    Q4<T> h;
    h.w = q.w / n; h.x = T(0); h.y = T(0); h.z = q.z / n;
    return h;
}

// de_heading: inverse(heading_q(q)) * q  (utils/math.py:80-81)
template <typename T> __device__ __forceinline__ Q4<T> de_heading(const Q4<T> &q) {
    return qmul(qinv(heading_q(q)), q);
}

// rows of quaternion_matrix(q)[:3,:3] applied transposed: R^T v  (utils/transformation.py:1267-1291,
// transform_vec utils/math.py:47-59). Identity when |q|^2 < 4 eps.
template <typename T> __device__ __forceinline__ V3<T> rotate_T(const Q4<T> &q, const V3<T> &v) {
    const T n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    const T eps4 = sizeof(T) == 8 ? T(8.881784197001252e-16) : T(4.76837158203125e-07);
    if (n < eps4) return v;
    const T s = T(2) / n;          // (q*sqrt(2/n)) outer itself == s * q_a q_b
    const T xx = s * q.x * q.x, yy = s * q.y * q.y, zz = s * q.z * q.z;
    const T xy = s * q.x * q.y, xz = s * q.x * q.z, yz = s * q.y * q.z;
    const T wx = s * q.w * q.x, wy = s * q.w * q.y, wz = s * q.w * q.z;
    // R = [[1-yy-zz, xy-wz, xz+wy],[xy+wz, 1-xx-zz, yz-wx],[xz-wy, yz+wx, 1-xx-yy]];  out = R^T v
    V3<T> o;
    o.x = (T(1) - yy - zz) * v.x + (xy + wz) * v.y + (xz - wy) * v.z;
    o.y = (xy - wz) * v.x + (T(1) - xx - zz) * v.y + (yz + wx) * v.z;
    o.z = (xz + wy) * v.x + (yz - wx) * v.y + (T(1) - xx - yy) * v.z;
    return o;
}

// rotation_from_quaternion(q, separate=True): axis, angle with the reference's 1-w<1e-8 identity
// branch, no clamp on w, no wrap  (utils/transformation.py:348-356)
template <typename T> __device__ __forceinline__ void rot_axis_angle(const Q4<T> &q, V3<T> *axis, T *angle) {
    if (T(1) - q.w < T(1e-8)) {
        axis->x = T(1); axis->y = T(0); axis->z = T(0);
        *angle = T(0);
    } else if (sizeof(T) == 8) {
        const T inv = T(1) / t_sqrt<T>(T(1) - q.w * q.w);
        axis->x = q.x * inv; axis->y = q.y * inv; axis->z = q.z * inv;
        *angle = T(2) * t_acos<T>(q.w);
    } else {
        // float32 variant: sqrt(1-w^2) and acos(w) lose all digits for small rotations; for a unit
        // quaternion |xyz| == sqrt(1-w^2) and atan2(|xyz|, w) == acos(w), both well conditioned.
        // (w can round to 1 - 1 ulp for q q^-1 while the vector part is exactly zero: the float64 test above does not
        //  catch that in float32, and 0 / 0 must not become the axis)
        const T s = t_sqrt<T>(q.x * q.x + q.y * q.y + q.z * q.z);
        if (s == T(0)) {
            axis->x = T(1); axis->y = T(0); axis->z = T(0);
            *angle = T(0);
        } else {
            axis->x = q.x / s; axis->y = q.y / s; axis->z = q.z / s;
            *angle = T(2) * atan2f((float)s, (float)q.w);
        }
    }
}

template <typename T> __device__ __forceinline__ T clamp1(T v) { return v < T(-1) ? T(-1) : (v > T(1) ? T(1) : v); }

// multi_quat_norm: arccos(clip(w, -1, 1)) = HALF the rotation angle  (utils/math.py:96-100)
template <typename T> __device__ __forceinline__ T half_angle(const Q4<T> &q) {
    if (sizeof(T) == 8) return t_acos<T>(clamp1<T>(q.w));
    const T s = t_sqrt<T>(q.x * q.x + q.y * q.y + q.z * q.z);   // float32: see rot_axis_angle
    return (T)atan2f((float)s, (float)q.w);
}

}  // namespace egp
