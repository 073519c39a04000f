// bias_act for gfx950: y = clamp(act(x + b) * gain) and its 1st/2nd-order gradient forms.
//
// Replaces bias_act_plugin.bias_act (reference torch_utils/ops/bias_act.cpp:36, bias_act.cu:27-151);
// semantics follow torch_utils/ops/bias_act.py:92-122 (forward) and the autograd wrappers :128-209.
//
// The op is a pure HBM stream (2-5 tensors in, 1 out), so the kernel is organised around 16-byte
// per-lane accesses (1 KiB per wave instruction), four vectors per tensor in flight per lane, a grid of
// <= 8 workgroups per CU striding over the tensor, and the bias index resolved once per vector (or once per
// workgroup) whenever the bias stride allows it.

#include "common.h"

namespace {

using namespace gnerf;

constexpr int kThreads = 256;

template <class A> struct Consts;
template <> struct Consts<float>  { static __device__ float  exp_range() { return 80.f; } };
template <> struct Consts<double> { static __device__ double exp_range() { return 80.0; } };

template <class A> __device__ __forceinline__ A act_exp(A v);
template <> __device__ __forceinline__ float  act_exp<float>(float v)   { return expf(v); }
template <> __device__ __forceinline__ double act_exp<double>(double v) { return exp(v); }
template <class A> __device__ __forceinline__ A act_log1p(A v);
template <> __device__ __forceinline__ float  act_log1p<float>(float v)   { return log1pf(v); }
template <> __device__ __forceinline__ double act_log1p<double>(double v) { return log1p(v); }
template <class A> __device__ __forceinline__ A act_tanh(A v);
template <> __device__ __forceinline__ float  act_tanh<float>(float v)   { return tanhf(v); }
template <> __device__ __forceinline__ double act_tanh<double>(double v) { return tanh(v); }

struct Args {
    const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
    int64_t numel; unsigned size_b; unsigned step_b;
    int grad; float alpha, gain, clamp;
};

// One element.  `in` is x (grad 0), dy (grad 1) or d_dx (grad 2); `pre` is xref + b (or unused);
// `yy` is the saved forward output divided by gain; `dy2` multiplies the result (1 when absent).
template <class A, int ACT>
__device__ __forceinline__ A eval(A in, A bias, A xref, A yref, A dy2, int grad, A alpha, A gain, A clamp) {
    const A one = A(1), two = A(2), zero = A(0);
    const A selu_scale = A(1.0507009873554804934193349852946);
    const A selu_alpha = A(1.6732632423543772848170429916717);
    const A yy = (gain != zero) ? yref / gain : zero;
    A out = zero;
    if (grad == 0) {
        const A v = in + bias;
        if (ACT == 1) out = v;
        if (ACT == 2) out = v > zero ? v : zero;
        if (ACT == 3) out = v > zero ? v : v * alpha;
        if (ACT == 4) out = act_tanh<A>(v);
        if (ACT == 5) out = one / (one + act_exp<A>(-v));
        if (ACT == 6) out = v >= zero ? v : act_exp<A>(v) - one;
        if (ACT == 7) out = v >= zero ? selu_scale * v : (selu_scale * selu_alpha) * (act_exp<A>(v) - one);
        if (ACT == 8) out = v > A(20) ? v : act_log1p<A>(act_exp<A>(v));
        if (ACT == 9) out = v / (one + act_exp<A>(-v));
    } else {
        const A pre = xref + bias;      // only meaningful for swish (its gradient is written in terms of x)
        if (grad == 1) {
            if (ACT == 1) out = in;
            if (ACT == 2) out = yy > zero ? in : zero;
            if (ACT == 3) out = yy > zero ? in : in * alpha;
            if (ACT == 4) out = in * (one - yy * yy);
            if (ACT == 5) out = in * yy * (one - yy);
            if (ACT == 6) out = yy >= zero ? in : in * (yy + one);
            if (ACT == 7) out = yy >= zero ? in * selu_scale : in * (yy + selu_scale * selu_alpha);
            if (ACT == 8) out = in * (one - act_exp<A>(-yy));
        } else {
            if (ACT == 4) out = in * (one - yy * yy) * (-two * yy);
            if (ACT == 5) out = in * yy * (one - yy) * (one - two * yy);
            if (ACT == 6) out = yy >= zero ? zero : in * (yy + one);
            if (ACT == 7) out = yy >= zero ? zero : in * (yy + selu_scale * selu_alpha);
            if (ACT == 8) { const A e = act_exp<A>(-yy); out = in * e * (one - e); }
        }
        if (ACT == 9) {
            // sigmoid s = 1/(1+exp(-pre));  swish' = s + pre*s*(1-s);  swish'' = s*(1-s)*(2 + pre*(1-2s))
            const A s = one / (one + act_exp<A>(-pre));
            const A ds = s * (one - s);
            out = (grad == 1) ? in * (s + pre * ds) : in * ds * (two + pre * (one - two * s));
            yref = pre * s * gain;      // the clamp mask below needs the forward output
        }
    }
    out *= gain * dy2;
    if (clamp >= zero) {
        if (grad == 0) {                                        // NaN -> -clamp, as the reference's kernel has it (bias_act.cu:143)
            if constexpr (sizeof(A) == 4) out = __builtin_amdgcn_fmed3f(out, -clamp, clamp);
            else                          out = (out > -clamp && out < clamp) ? out : (out >= zero ? clamp : -clamp);
        }
        else           out = (yref > -clamp && yref < clamp) ? out : zero;
    }
    return out;
}

template <class T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

// Both kernels move a batch of vectors per lane per wait: all loads of a batch are issued, then the arithmetic, then all
// stores.  Loads and stores share one completion counter (vmcnt) on this hardware, and with both kinds pending the compiler
// must wait for everything -- a load -> store -> load -> store loop waits for each store's acknowledgement before the next
// load's data can be used.  Out-of-range slots of the last batch load a valid address and skip the store.
// Batch size by element type, measured on [4,128,512,512], [4,256,256,256] and [1,128,512,512] (sweep 1, 2, 3, 4, 8): two 16-byte
// vectors for fp16 (4.75 -> 5.0, 5.4 -> 5.7-6.0, 4.85 -> 5.3-5.4 TB/s against four), one for fp32 / fp64 (6.0 -> 6.3, 5.7 -> 6.2,
// 6.6 -> 6.9 TB/s): enough workgroups are resident for the memory system without deeper per-lane batches, which only add registers.
template <class T> __host__ __device__ constexpr int batch_of() { return sizeof(T) == 2 ? 2 : 1; }

template <class T, int ACT, int VEC, int kBatch>
__device__ __forceinline__ void act_batch(const Args& a, const int64_t (&i0)[kBatch], const bool (&ok)[kBatch],
                                          const typename Arith<T>::type (&bias)[kBatch], bool bias_per_elem,
                                          const typename Arith<T>::type* bias_vec = nullptr) {
    typedef typename Arith<T>::type A;
    typedef Pack<T, VEC> P;
    const A alpha = A(a.alpha), gain = A(a.gain), clamp = A(a.clamp);
    const T* __restrict__ x = static_cast<const T*>(a.x);
    const T* __restrict__ b = static_cast<const T*>(a.b);
    const T* __restrict__ xref = static_cast<const T*>(a.xref);
    const T* __restrict__ yref = static_cast<const T*>(a.yref);
    const T* __restrict__ dy = static_cast<const T*>(a.dy);
    T* __restrict__ y = static_cast<T*>(a.y);
    P px[kBatch], pxr[kBatch], pyr[kBatch], pdy[kBatch], po[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; t++) {
        px[t] = *reinterpret_cast<const P*>(x + i0[t]);
        if (xref) pxr[t] = *reinterpret_cast<const P*>(xref + i0[t]);
        if (yref) pyr[t] = *reinterpret_cast<const P*>(yref + i0[t]);
        if (dy)   pdy[t] = *reinterpret_cast<const P*>(dy + i0[t]);
    }
#pragma unroll
    for (int t = 0; t < kBatch; t++) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            A bv = bias[t];
            if (bias_vec) bv = bias_vec[k];
            else if (bias_per_elem) bv = load_as<T>(b, (unsigned(i0[t] + k) / a.step_b) % a.size_b);
            const A r = eval<A, ACT>(load_as<T>(px[t].v, k), bv,
                                     xref ? load_as<T>(pxr[t].v, k) : A(0), yref ? load_as<T>(pyr[t].v, k) : A(0),
                                     dy ? load_as<T>(pdy[t].v, k) : A(1), a.grad, alpha, gain, clamp);
            store_as<T>(po[t].v, k, r);
        }
    }
#pragma unroll
    for (int t = 0; t < kBatch; t++)
        if (ok[t]) *reinterpret_cast<P*>(y + i0[t]) = po[t];
}

template <class T, int ACT, int VEC>
__global__ __launch_bounds__(kThreads) void bias_act_kernel(Args a) {
    typedef typename Arith<T>::type A;
    constexpr int kBatch = batch_of<T>();
    const A alpha = A(a.alpha), gain = A(a.gain), clamp = A(a.clamp);
    const T* x = static_cast<const T*>(a.x);
    const T* b = static_cast<const T*>(a.b);
    const T* xref = static_cast<const T*>(a.xref);
    const T* yref = static_cast<const T*>(a.yref);
    const T* dy = static_cast<const T*>(a.dy);
    T* y = static_cast<T*>(a.y);
    const int64_t nvec = a.numel / VEC;
    const bool bias_per_vec = (b != nullptr) && (a.step_b % VEC == 0);
    const bool bias_per_elem = (b != nullptr) && !bias_per_vec;
    const int64_t stride = int64_t(gridDim.x) * kThreads;
    for (int64_t iv = int64_t(blockIdx.x) * kThreads + threadIdx.x; iv < nvec; iv += stride * kBatch) {
        int64_t i0[kBatch];
        bool ok[kBatch];
        A bias[kBatch];
#pragma unroll
        for (int t = 0; t < kBatch; t++) {
            const int64_t v = iv + t * stride;
            ok[t] = v < nvec;
            i0[t] = (ok[t] ? v : iv) * VEC;
            bias[t] = bias_per_vec ? load_as<T>(b, (unsigned(i0[t]) / a.step_b) % a.size_b) : A(0);
        }
        act_batch<T, ACT, VEC, kBatch>(a, i0, ok, bias, bias_per_elem);
    }
    // ragged tail (fewer than VEC elements), one lane each
    const int64_t tail0 = nvec * VEC;
    const int64_t it = tail0 + int64_t(blockIdx.x) * kThreads + threadIdx.x;
    if (it < a.numel) {
        const A bv = b ? load_as<T>(b, (unsigned(it) / a.step_b) % a.size_b) : A(0);
        const A r = eval<A, ACT>(load_as<T>(x, it), bv, xref ? load_as<T>(xref, it) : A(0), yref ? load_as<T>(yref, it) : A(0),
                                 dy ? load_as<T>(dy, it) : A(1), a.grad, alpha, gain, clamp);
        store_as<T>(y, it, r);
    }
}

// Row form for the common NCHW case (bias stride = H*W, large and a multiple of the vector width): the tensor is
// [rows = numel / step_b][step_b] and the bias is constant along a row, so blockIdx.y = row makes the bias index a
// per-workgroup scalar -- no per-vector integer division, which otherwise costs as much as the activation itself
// (measured on [4,128,512,512]: fp16 3.5 -> 4.8 TB/s, fp32 5.0 -> 5.7 TB/s).
template <class T, int ACT, int VEC>
__global__ __launch_bounds__(kThreads) void bias_act_rows_kernel(Args a) {
    typedef typename Arith<T>::type A;
    constexpr int kBatch = batch_of<T>(), kRowVecsPerLane = kBatch;
    const unsigned row = blockIdx.y;
    const A bv = load_as<T>(static_cast<const T*>(a.b), row % a.size_b);
    const int64_t row0 = int64_t(row) * a.step_b;
    const unsigned nvec_row = a.step_b / VEC;
    const unsigned v0 = blockIdx.x * kRowVecsPerLane * kThreads + threadIdx.x;
    if (v0 >= nvec_row) return;
    int64_t i0[kBatch];
    bool ok[kBatch];
    A bias[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; t++) {
        const unsigned v = v0 + t * kThreads;
        ok[t] = v < nvec_row;
        i0[t] = row0 + int64_t(ok[t] ? v : v0) * VEC;
        bias[t] = bv;
    }
    act_batch<T, ACT, VEC, kBatch>(a, i0, ok, bias, false);
}

// Channels-last form (bias along the fastest axis: step_b = 1, e.g. a channels_last activation tensor biased per channel).  A 16-byte
// vector then covers VEC consecutive channels, and with a grid stride that is a multiple of the vectors per pixel a lane meets the SAME
// channels on every trip: its VEC biases are loaded once and live in registers.  The general kernel resolves the bias per ELEMENT here
// (an integer modulo and a load each): 3.5 TB/s on the superresolution's [1,128,512,512] fp16 layers against the row form's 4.8+.
template <class T, int ACT, int VEC>
__global__ __launch_bounds__(kThreads) void bias_act_cl_kernel(Args a) {
    typedef typename Arith<T>::type A;
    constexpr int kBatch = batch_of<T>();
    const int64_t nvec = a.numel / VEC;                                   // numel % size_b == 0 and size_b % VEC == 0: no ragged tail
    const int64_t stride = int64_t(gridDim.x) * kThreads;                 // a multiple of size_b / VEC (the launcher checks)
    const int64_t iv0 = int64_t(blockIdx.x) * kThreads + threadIdx.x;
    A bvec[VEC];
    const unsigned c0 = unsigned((iv0 * VEC) % a.size_b);
#pragma unroll
    for (int k = 0; k < VEC; k++) bvec[k] = load_as<T>(static_cast<const T*>(a.b), c0 + k);
    for (int64_t iv = iv0; iv < nvec; iv += stride * kBatch) {
        int64_t i0[kBatch];
        bool ok[kBatch];
        A bias[kBatch];
#pragma unroll
        for (int t = 0; t < kBatch; t++) {
            const int64_t v = iv + t * stride;
            ok[t] = v < nvec;
            i0[t] = (ok[t] ? v : iv) * VEC;
            bias[t] = A(0);
        }
        act_batch<T, ACT, VEC, kBatch>(a, i0, ok, bias, false, bvec);
    }
}

template <class T, int VEC>
int launch_act(const Args& a, int act, hipStream_t stream) {
    const int64_t nvec = (a.numel + VEC - 1) / VEC;
    int64_t blocks = (nvec + kThreads - 1) / kThreads;
    const int64_t cap = int64_t(kNumCU) * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 g((unsigned)blocks), t(kThreads);
    if (VEC > 1 && a.b && a.step_b >= 2048 && a.step_b % VEC == 0 && a.numel % a.step_b == 0 && a.numel / a.step_b <= 65535) {
        const unsigned nvec_row = a.step_b / VEC;
        constexpr int kRowVecsPerLane = batch_of<T>();
        dim3 gr((nvec_row + kThreads * kRowVecsPerLane - 1) / (kThreads * kRowVecsPerLane), (unsigned)(a.numel / a.step_b));
        switch (act) {
#define GNERF_CASE(A_) case A_: hipLaunchKernelGGL((bias_act_rows_kernel<T, A_, VEC>), gr, t, 0, stream, a); break;
            GNERF_CASE(1) GNERF_CASE(2) GNERF_CASE(3) GNERF_CASE(4) GNERF_CASE(5)
            GNERF_CASE(6) GNERF_CASE(7) GNERF_CASE(8) GNERF_CASE(9)
#undef GNERF_CASE
            default: return fail(GNERF_E_ARG, "bias_act: unknown activation %d", act);
        }
        return check_launch("bias_act(rows)");
    }
    if (VEC > 1 && a.b && a.step_b == 1 && a.size_b % VEC == 0 && a.numel % a.size_b == 0 && (blocks * kThreads) % (a.size_b / VEC) == 0) {
        switch (act) {
#define GNERF_CASE(A_) case A_: hipLaunchKernelGGL((bias_act_cl_kernel<T, A_, VEC>), g, t, 0, stream, a); break;
            GNERF_CASE(1) GNERF_CASE(2) GNERF_CASE(3) GNERF_CASE(4) GNERF_CASE(5)
            GNERF_CASE(6) GNERF_CASE(7) GNERF_CASE(8) GNERF_CASE(9)
#undef GNERF_CASE
            default: return fail(GNERF_E_ARG, "bias_act: unknown activation %d", act);
        }
        return check_launch("bias_act(channels_last)");
    }
    switch (act) {
#define GNERF_CASE(A_) case A_: hipLaunchKernelGGL((bias_act_kernel<T, A_, VEC>), g, t, 0, stream, a); break;
        GNERF_CASE(1) GNERF_CASE(2) GNERF_CASE(3) GNERF_CASE(4) GNERF_CASE(5)
        GNERF_CASE(6) GNERF_CASE(7) GNERF_CASE(8) GNERF_CASE(9)
#undef GNERF_CASE
        default: return fail(GNERF_E_ARG, "bias_act: unknown activation %d", act);
    }
    return check_launch("bias_act");
}

bool aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int gnerf_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                              void* y, int dtype, int64_t numel, int size_b, int64_t step_b,
                              int grad, int act, float alpha, float gain, float clamp, gnerf_stream_t stream) {
    using namespace gnerf;
    if (numel == 0) return GNERF_OK;
    if (!x || !y) return fail(GNERF_E_ARG, "bias_act: x and y must not be null");
    if (numel < 0 || numel > INT32_MAX) return fail(GNERF_E_ARG, "bias_act: x is too large (%lld elements)", (long long)numel);
    if (grad < 0 || grad > 2) return fail(GNERF_E_ARG, "bias_act: grad must be 0, 1 or 2");
    if (b && (size_b <= 0 || step_b <= 0)) return fail(GNERF_E_ARG, "bias_act: bias given with size %d, step %lld", size_b, (long long)step_b);
    if (act < 1 || act > 9) return fail(GNERF_E_ARG, "bias_act: unknown activation %d", act);
    Args a{x, b, xref, yref, dy, y, numel, b ? unsigned(size_b) : 1u, b ? unsigned(step_b) : 1u, grad, alpha, gain, clamp};
    const bool vec_ok = aligned16(x) && aligned16(y) && aligned16(xref) && aligned16(yref) && aligned16(dy);
    hipStream_t s = as_stream(stream);
    switch (dtype) {
        case GNERF_F32: return vec_ok ? launch_act<float, 4>(a, act, s) : launch_act<float, 1>(a, act, s);
        case GNERF_F16: return vec_ok ? launch_act<__half, 8>(a, act, s) : launch_act<__half, 1>(a, act, s);
        case GNERF_F64: return vec_ok ? launch_act<double, 2>(a, act, s) : launch_act<double, 1>(a, act, s);
        default: return fail(GNERF_E_ARG, "bias_act: unsupported dtype code %d", dtype);
    }
}
