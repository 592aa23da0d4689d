// Dense optimiser + L2 + loss reduction kernels, gfx950.  Pure HBM streaming.
//
// The reference updates EVERY element of EVERY table EVERY step: the L2 term
// lambda/(2B)*||p||^2 (sert/models.py:764-795) gives every parameter a non-zero
// gradient, and lasagne.updates.adam / adadelta (models.py:922 / :820, applied
// :548-549) are dense element-wise maps.  One fused kernel per tensor:
//   g  = grad + (lambda/B) * p          (b: no L2 [upstream: regularizable=False])
//   sumsq += p^2  (pre-update, for the loss the step returns)
//   state/param update
#pragma once
#include "common.h"

namespace sert {

constexpr int kOptBlocks = 1024;  // fixed grid => fixed reduction tree => deterministic

struct AdamArgs {
    float l2k;   // lambda / B
    float a_t;   // lr*sqrt(1-b2^t)/(1-b1^t), evaluated on the host in fp32
    float b1, b2, eps;
};

// Lasagne 0.1 adam [upstream]:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p = p - a_t*m/(sqrt(v)+eps)
template <bool STORE_G>
__global__ __launch_bounds__(256) void adam_l2(float* __restrict__ p, float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v,
                                               size_t count, AdamArgs a,
                                               float* __restrict__ sumsq_partial) {
    __shared__ float red[4];
    float ss = 0.f;
    const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gv = g[i] + a.l2k * pv;
        ss += pv * pv;
        const float mv = a.b1 * m[i] + omb1 * gv;
        const float vv = a.b2 * v[i] + omb2 * gv * gv;
        m[i] = mv;
        v[i] = vv;
        p[i] = pv - a.a_t * mv / (sqrtf(vv) + a.eps);
        if (STORE_G) g[i] = gv;
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = tot;
}

struct AdadeltaArgs {
    float l2k, lr, rho, eps;
};

// Lasagne 0.1 adadelta [upstream]:
//   a = rho*a + (1-rho)*g^2 ; u = g*sqrt(d+eps)/sqrt(a+eps) ; p = p - lr*u ;
//   d = rho*d + (1-rho)*u^2
template <bool STORE_G>
__global__ __launch_bounds__(256) void adadelta_l2(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ accu,
                                                   float* __restrict__ delta, size_t count,
                                                   AdadeltaArgs a,
                                                   float* __restrict__ sumsq_partial) {
    __shared__ float red[4];
    float ss = 0.f;
    const float omr = 1.0f - a.rho;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gv = g[i] + a.l2k * pv;
        ss += pv * pv;
        const float av = a.rho * accu[i] + omr * gv * gv;
        const float dv = delta[i];
        const float u = gv * sqrtf(dv + a.eps) / sqrtf(av + a.eps);
        accu[i] = av;
        p[i] = pv - a.lr * u;
        delta[i] = a.rho * dv + omr * u * u;
        if (STORE_G) g[i] = gv;
    }
    const float tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = tot;
}

// partial[b] = sum of block b's strided share of in[0..count)
__global__ __launch_bounds__(256) void sum_partial(const float* __restrict__ in, size_t count,
                                                   float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        s += in[i];
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// loss = (sum rowloss)/B + lambda/(2B) * sum of squares of the regularised
// tensors (models.py:278-282, :773-791).  Single block, fp64 accumulation of
// the (few thousand) partials, fixed order.
//   parts: [loss partials (n_loss)] [sumsq partials (n_sq)]
__global__ __launch_bounds__(256) void finalize_loss(const float* __restrict__ loss_partials,
                                                     int n_loss,
                                                     const float* __restrict__ sq_partials,
                                                     int n_sq, float inv_batch, float reg_scale,
                                                     float* __restrict__ out) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n_loss; i += 256) a += (double)loss_partials[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double loss_sum = red[0];
    __syncthreads();
    double b = 0.0;
    for (int i = threadIdx.x; i < n_sq; i += 256) b += (double)sq_partials[i];
    red[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float data = (float)loss_sum * inv_batch;
        const float reg = reg_scale * (float)red[0];
        out[0] = data + reg;
        out[1] = data;
        out[2] = reg;
    }
}

}  // namespace sert
