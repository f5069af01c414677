// velocity.hip -- stage C: predict_U -> calculate_velocity -> calculate_shift ->
// extrapolate_cell_at_t and the `dmat` transform of estimate_transition_prob, fused into one
// streaming pass (analysis.py:1321-1439, 1538, 1575-1601).  The reference materialises five
// (G,C) fp64 temporaries (48 GB at 50k x 30k); here each requested output is written once and
// everything else stays in registers.  HBM-bound: 2 reads + (#outputs) writes of G*s per cell.
#include "common.h"

namespace vcy {

template <typename T> __device__ __forceinline__ T vsqrt(T x);
template <> __device__ __forceinline__ float vsqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> __device__ __forceinline__ double vsqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T vlog10(T x);
template <> __device__ __forceinline__ float vlog10<float>(float x) { return __builtin_amdgcn_logf(x) * 0.30102999566398120f; }
template <> __device__ __forceinline__ double vlog10<double>(double x) { return log10(x); }

struct VelArgs {
    const void *Sx, *Ux;
    const float *gamma, *q;
    const double *eps_thr;
    void *Upred, *velocity, *delta_S, *Sx_t, *dmat;
    int64_t C, ld;
    int G;
    double dt_shift, dt_extrap, used_dt, psc;
    int assumption, clip, transform;
};

// np.sign(D) * f(|D| + psc)  (analysis.py:1577, 1597): sign(0) = 0 -> dmat = 0 where D == 0.
template <typename T> __device__ __forceinline__ T dmat_of(T D, int transform, T psc)
{
    if (transform == VCY_LINEAR) return D;
    const T a = fabs(D) + psc;
    const T f = transform == VCY_SQRT ? vsqrt<T>(a) : vlog10<T>(a);
    return D > T(0) ? f : (D < T(0) ? -f : T(0) * f);
}

// thread = one 16-byte gene vector, fixed for the thread's lifetime (gamma, q, eps live in registers);
// blockIdx.y = block of cells walked with 4 rows of loads in flight.  Lanes map to adjacent gene
// vectors, so every load/store instruction covers one contiguous 1 KiB row segment.
constexpr int VEL_CB = 64;

template <typename T> __global__ __launch_bounds__(256) void k_velocity_chain(VelArgs a)
{
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    const int nvec = (int)(a.ld / N);
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvec) return;
    const T *Sx = (const T *)a.Sx, *Ux = (const T *)a.Ux;
    T gm[N], qq[N];
    double eps[N];
    bool live[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int g = v * N + k;
        live[k] = g < a.G;
        gm[k] = live[k] ? (T)a.gamma[g] : T(0);
        qq[k] = (live[k] && a.q) ? (T)a.q[g] : T(0);
        eps[k] = (live[k] && a.eps_thr) ? a.eps_thr[g] : -1.0;
    }
    float egt32[N];
#pragma unroll
    for (int k = 0; k < N; ++k)   // the reference forms exp(-gammas*dt) and (1 - egt) in FLOAT32 (gammas is a float32 array
        egt32[k] = (a.assumption == 1 && live[k]) ? expf(-(a.gamma[v * N + k] * (float)a.dt_shift)) : 0.f;   // and numpy keeps float32)
    const int64_t per = (a.C + VEL_CB - 1) / VEL_CB;
    const int64_t c0 = blockIdx.y * per, c1 = c0 + per < a.C ? c0 + per : a.C;

    auto one = [&](int64_t c, const V &sv, const V &uv) {
        const int64_t o = c * a.ld + (int64_t)v * N;
        const T *sp = reinterpret_cast<const T *>(&sv);
        const T *up = reinterpret_cast<const T *>(&uv);
        V o_up, o_vel, o_ds, o_st, o_dm;
        T *pu = reinterpret_cast<T *>(&o_up), *pv = reinterpret_cast<T *>(&o_vel), *pd = reinterpret_cast<T *>(&o_ds),
          *pt = reinterpret_cast<T *>(&o_st), *pm = reinterpret_cast<T *>(&o_dm);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            T upred = T(0), vel = T(0), ds = T(0), st = T(0), dm = T(0);
            if (live[k]) {
                const T s = sp[k], u = up[k];
                upred = gm[k] * s + qq[k];                         // analysis.py:1346
                vel = u - upred;                                   // :1369
                if (eps[k] >= 0.0 && fabs((double)vel) < eps[k]) vel = T(0);   // :1377-1379
                if (a.assumption == 0) ds = (T)a.dt_shift * vel;   // :1399
                else {                                             // :1403-1406
                    T uo = u - qq[k];
                    uo = uo < T(0) ? T(0) : uo;
                    const T egt = (T)egt32[k], omegt = (T)(1.0f - egt32[k]);
                    ds = s * egt + omegt * uo / gm[k] - s;
                }
                st = s + (T)a.dt_extrap * ds;                      // :1429
                if (a.clip) st = st < T(0) ? T(0) : st;            // :1431
                const T D = (s + (T)a.used_dt * ds) - s;           // :1538 + :1576/1596 (hi_dim_t - hi_dim)
                dm = dmat_of<T>(D, a.transform, (T)a.psc);
            }
            pu[k] = upred; pv[k] = vel; pd[k] = ds; pt[k] = st; pm[k] = dm;
        }
        if (a.Upred) *reinterpret_cast<V *>((T *)a.Upred + o) = o_up;
        if (a.velocity) *reinterpret_cast<V *>((T *)a.velocity + o) = o_vel;
        if (a.delta_S) *reinterpret_cast<V *>((T *)a.delta_S + o) = o_ds;
        if (a.Sx_t) *reinterpret_cast<V *>((T *)a.Sx_t + o) = o_st;
        if (a.dmat) *reinterpret_cast<V *>((T *)a.dmat + o) = o_dm;
    };
    int64_t c = c0;
    for (; c + 3 < c1; c += 4) {
        V sv[4], uv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sv[u] = reinterpret_cast<const V *>(Sx + (c + u) * a.ld)[v];
            uv[u] = reinterpret_cast<const V *>(Ux + (c + u) * a.ld)[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(c + u, sv[u], uv[u]);
    }
    for (; c < c1; ++c) one(c, reinterpret_cast<const V *>(Sx + c * a.ld)[v], reinterpret_cast<const V *>(Ux + c * a.ld)[v]);
}
// One stage of the chain from its STORED predecessor, the way the reference's methods read self.Upred / self.velocity /
// self.delta_S (analysis.py:1369, 1399, 1429-1431) - so that an intermediate the user has edited propagates:
//     out = a * x + b * y      (two roundings of the products, then the sum: numpy's evaluation order, no FMA contraction)
//     |out| < zero_below[g] -> 0   (the eps rule of calculate_velocity, :1377-1379)        clip: max(out, 0)  (:1431)
template <typename T>
__global__ __launch_bounds__(256) void k_lincomb(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ out, T a, T b,
                                                  const double *__restrict__ zero_below, int clip, int64_t C, int G, int64_t ld)
{
#pragma clang fp contract(off)
    using V = typename Vec<T>::type;
    constexpr int N = Vec<T>::N;
    const int64_t nvec = ld / N, total = C * nvec;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = t / nvec;
        const int v = (int)(t - c * nvec);
        const V xv = reinterpret_cast<const V *>(x + c * ld)[v];
        V yv = xv, ov;
        if (y) yv = reinterpret_cast<const V *>(y + c * ld)[v];
        const T *xp = reinterpret_cast<const T *>(&xv), *yp = reinterpret_cast<const T *>(&yv);
        T *op = reinterpret_cast<T *>(&ov);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int g = v * N + k;
            T r = T(0);
            if (g < G) {
                const T p = a * xp[k];
                r = y ? p + b * yp[k] : p;
                if (zero_below && fabs((double)r) < zero_below[g]) r = T(0);
                if (clip) r = r < T(0) ? T(0) : r;
            }
            op[k] = r;
        }
        reinterpret_cast<V *>(out + c * ld)[v] = ov;
    }
}
}  // namespace vcy

using namespace vcy;

extern "C" int vcy_lincomb(const void *x, const void *y, void *out, double a, double b, const double *zero_below, int clip, int64_t C,
                           int64_t G, int64_t ld, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(x && out && C > 0 && G > 0 && ld >= G, "lincomb: bad arguments");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "lincomb: bad dtype");
    VCY_REQUIRE(ld % (dtype == VCY_F32 ? 4 : 2) == 0, "lincomb: ld must keep rows 16-byte aligned");
    const int64_t total = C * (ld / (dtype == VCY_F32 ? 4 : 2));
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_lincomb<float>, dim3(blocks), dim3(256), 0, st, (const float *)x, (const float *)y, (float *)out, (float)a, (float)b, zero_below, clip, C, (int)G, ld);
    else hipLaunchKernelGGL(k_lincomb<double>, dim3(blocks), dim3(256), 0, st, (const double *)x, (const double *)y, (double *)out, a, b, zero_below, clip, C, (int)G, ld);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}

extern "C" int vcy_velocity_chain(const void *Sx_sz, const void *Ux_sz, const float *gamma, const float *q, const double *eps_thr,
                                  void *Upred, void *velocity, void *delta_S, void *Sx_sz_t, void *dmat, int64_t C, int64_t G,
                                  int64_t ld, double dt_shift, double dt_extrap, double used_dt, int assumption, int clip,
                                  int transform, double psc, int dtype, vcy_stream stream)
{
    VCY_REQUIRE(Sx_sz && Ux_sz && gamma, "velocity_chain: null pointer");
    VCY_REQUIRE(C > 0 && G > 0 && ld >= G, "velocity_chain: bad shape");
    VCY_REQUIRE(dtype == VCY_F32 || dtype == VCY_F64, "velocity_chain: bad dtype");
    VCY_REQUIRE(ld % (dtype == VCY_F32 ? 4 : 2) == 0, "velocity_chain: ld must keep rows 16-byte aligned");
    VCY_REQUIRE(transform >= VCY_LINEAR && transform <= VCY_LOG10, "velocity_chain: bad transform");
    VCY_REQUIRE(assumption == 0 || assumption == 1, "velocity_chain: bad assumption");
    VelArgs a{Sx_sz, Ux_sz, gamma, q, eps_thr, Upred, velocity, delta_S, Sx_sz_t, dmat, C, ld, (int)G,
              dt_shift, dt_extrap, used_dt, psc, assumption, clip, transform};
    const int N = dtype == VCY_F32 ? 4 : 2;
    const dim3 grid((unsigned)((ld / N + 255) / 256), VEL_CB);
    hipStream_t st = as_stream(stream);
    if (dtype == VCY_F32) hipLaunchKernelGGL(k_velocity_chain<float>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_velocity_chain<double>, grid, dim3(256), 0, st, a);
    VCY_LAUNCH_CHECK();
    return VCY_OK;
}
