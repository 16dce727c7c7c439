// Default boxes, IoU matching, hard-negative mining and the MultiBox loss (forward + analytic backward).
// Reference: model/ssd.py:108-133 (default boxes), 181-328 (loss and helpers).
//
// Built with -ffp-contract=off: the match mask must reproduce the reference's separately-rounded fp32 ops
// (ssd.py:247-250) bit for bit, so no FMA contraction and IEEE division in this translation unit.
//
// Kernels
//   default_boxes_kernel   one thread per box, fp64 arithmetic rounded once to fp32 (python double -> torch.Tensor)
//   multibox_terms_kernel  one thread per (image, prior): GT rows staged in LDS, per-prior log-softmax in registers,
//                          emits l_loc / l_conf_pos / l_conf_neg (N,P) and the per-image positive count (ballot+atomic)
//   multibox_select_kernel one 1024-thread block per image: both (N,P) rows staged in LDS as order-preserving bits,
//                          exact (k+1)-th largest by 4-pass radix select (LDS histograms), then the masked sum
//   multibox_bwd_kernel    one thread per (image, prior): recomputes the match and writes dL/d outputs
#include "common.h"

namespace {

struct BoxMaps {
    int m[8], n[8], a[8], off[9];
    int count;
};

__global__ void default_boxes_kernel(float* __restrict__ out, BoxMaps maps, double s_min, double s_max, int total) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    int k = 0;
    while (k + 1 < maps.count && p >= maps.off[k + 1]) ++k;
    int local = p - maps.off[k];
    int an = maps.a[k];
    int a = local % an;
    int cell = local / an;
    int j = cell % maps.n[k];
    int i = cell / maps.n[k];
    int mm = maps.count;
    // s_(k) = s_min + (s_max - s_min) * (k - 1) / (m - 1), k is 1-based in the reference (ssd.py:114-115)
    double sk = s_min + (s_max - s_min) * (double)k / (double)(mm - 1);
    double sk1 = s_min + (s_max - s_min) * (double)(k + 1) / (double)(mm - 1);
    double w, h;
    if (a == an - 1) {  // 'add'
        w = h = sqrt(sk * sk1);
    } else {
        // aspects [1, 2, 1/2, 3, 1/3]
        double asp = (a == 0) ? 1.0 : (a == 1) ? 2.0 : (a == 2) ? (1.0 / 2.0) : (a == 3) ? 3.0 : (1.0 / 3.0);
        w = sk * sqrt(asp);
        h = sk * sqrt(1.0 / asp);
    }
    float4 r;
    r.x = (float)(((double)i + 0.5) / (double)maps.m[k]);  // cx follows the ROW index
    r.y = (float)(((double)j + 0.5) / (double)maps.n[k]);
    r.z = (float)w;
    r.w = (float)h;
    reinterpret_cast<float4*>(out)[p] = r;
}

// ---- shared per-(prior, gt) arithmetic ---------------------------------------------------------------------
// SSD._match, ssd.py:242-250.  Every operation below is one rounded fp32 op, in the reference's order.
__device__ __forceinline__ bool match_one(float g_cx, float g_cy, float g_w, float g_h, float d_cx, float d_cy, float d_w,
                                          float d_h, float threshold) {
    float gw2 = g_w / 2.0f, dw2 = d_w / 2.0f, gh2 = g_h / 2.0f, dh2 = d_h / 2.0f;
    float w = fminf(g_cx + gw2, d_cx + dw2) - fmaxf(g_cx - gw2, d_cx - dw2);
    w = fmaxf(w, 0.0f);
    float h = fminf(g_cy + gh2, d_cy + dh2) - fmaxf(g_cy - gh2, d_cy - dh2);
    h = fmaxf(h, 0.0f);
    float ga = g_w * g_h;
    float wh = w * h;
    float iou = (ga > 0.0f) ? wh / (ga + d_w * d_h - wh) : ga;
    return iou > threshold;
}

__device__ __forceinline__ float smooth_l1(float x) {  // ssd.py:283
    float ax = fabsf(x);
    return ax < 1.0f ? 0.5f * x * x : ax - 0.5f;
}
__device__ __forceinline__ float smooth_l1_grad(float x) {
    float ax = fabsf(x);
    return ax < 1.0f ? x : (x > 0.0f ? 1.0f : -1.0f);
}

template <int NC>
__device__ __forceinline__ void row_log_softmax(const float* __restrict__ row, float (&lsm)[NC]) {
    float mx = row[0];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        lsm[c] = row[c];
        mx = fmaxf(mx, lsm[c]);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        lsm[c] -= mx;
        s += expf(lsm[c]);
    }
    float ls = logf(s);
#pragma unroll
    for (int c = 0; c < NC; ++c) lsm[c] -= ls;
}

template <int NC>
__global__ __launch_bounds__(256) void multibox_terms_kernel(const float* __restrict__ outputs, const float* __restrict__ targets,
                                                             const float* __restrict__ dboxes, int P, int G, float alpha,
                                                             float threshold, float* __restrict__ terms, int* __restrict__ cnt,
                                                             int N) {
    extern __shared__ float gt[];  // (G, 4+NC)
    constexpr int ROW = 4 + NC;
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < G * ROW; i += blockDim.x) gt[i] = targets[(size_t)n * G * ROW + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool any = false;
    if (p < P) {
        const float4 d = reinterpret_cast<const float4*>(dboxes)[p];
        const float* row = outputs + ((size_t)n * P + p) * ROW;
        float o0 = row[0], o1 = row[1], o2 = row[2], o3 = row[3];
        float lsm[NC];
        row_log_softmax<NC>(row + 4, lsm);
        float l_loc = 0.0f, l_pos = 0.0f;
        for (int g = 0; g < G; ++g) {
            const float* t = gt + g * ROW;
            float g_cx = t[0], g_cy = t[1], g_w = t[2], g_h = t[3];
            if (!match_one(g_cx, g_cy, g_w, g_h, d.x, d.y, d.z, d.w, threshold)) continue;
            any = true;
            // SSD._calc_delta, ssd.py:267-270
            float h0 = (g_cx - d.x) / d.z;
            float h1 = (g_cy - d.y) / d.w;
            float h2 = g_w > 0.0f ? logf(g_w / d.z) : g_w;
            float h3 = g_h > 0.0f ? logf(g_h / d.w) : g_h;
            float s4 = ((smooth_l1(o0 - h0) + smooth_l1(o1 - h1)) + smooth_l1(o2 - h2)) + smooth_l1(o3 - h3);
            l_loc += s4;
            float ce = 0.0f;  // -(gt * log_softmax).sum(), ssd.py:298
#pragma unroll
            for (int c = 0; c < NC; ++c) ce += t[4 + c] * lsm[c];
            l_pos += -ce;
        }
        float l_neg = any ? 0.0f : -lsm[0];
        size_t np = (size_t)N * P;
        size_t idx = (size_t)n * P + p;
        terms[idx] = l_loc;
        terms[np + idx] = l_pos;
        terms[2 * np + idx] = l_neg;
    }
    unsigned long long b = __ballot(any);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt[n], (int)__popcll(b));
}

__device__ __forceinline__ unsigned canon_bits(float v) {
    // all loss terms are >= 0 (or -0.0 / NaN); canonicalise -0 so unsigned order == float order
    if (v == 0.0f) v = 0.0f;
    return __float_as_uint(v);
}

// exact (want)-th largest (1-based) of vals[0..P) by MSB-first radix select; all threads return the same bits
__device__ unsigned radix_select_desc(const unsigned* __restrict__ vals, int P, int want, unsigned* hist, unsigned* bcast) {
    unsigned prefix = 0, mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            unsigned v = vals[i];
            if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // wave 0: suffix sums over 256 bins, 4 bins per lane (lane l owns bins 4l..4l+3)
            int l = threadIdx.x;
            unsigned h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            unsigned own = h0 + h1 + h2 + h3;
            unsigned above = 0;  // sum over lanes > l
            unsigned run = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                unsigned t = __shfl_down(run, o, 64);
                if (l + o < 64) run += t;
            }
            above = run - own;  // elements in bins above this lane's bins
            unsigned w = (unsigned)want;
            // this lane holds the target bin iff above < want <= above + own
            if (above < w && w <= above + own) {
                unsigned c = above;
                int bin;
                if (c + h3 >= w) bin = 4 * l + 3;
                else {
                    c += h3;
                    if (c + h2 >= w) bin = 4 * l + 2;
                    else {
                        c += h2;
                        if (c + h1 >= w) bin = 4 * l + 1;
                        else { c += h1; bin = 4 * l; }
                    }
                }
                bcast[0] = (unsigned)bin;
                bcast[1] = w - c;  // rank inside the bin
            }
        }
        __syncthreads();
        prefix |= bcast[0] << shift;
        mask |= 255u << shift;
        want = (int)bcast[1];
        __syncthreads();
    }
    return prefix;
}

// `single` (P too large for two staged rows, e.g. SSD512's 24,564 priors): one LDS row, staged twice (positives, then negatives)
__global__ __launch_bounds__(1024) void multibox_select_kernel(const float* __restrict__ terms, const int* __restrict__ cnt, int N,
                                                               int P, float alpha, float* __restrict__ sel, int single) {
    extern __shared__ unsigned lds_u[];
    const int rows = single ? 1 : 2;
    unsigned* vpos = lds_u;                          // P
    unsigned* vneg = single ? lds_u : lds_u + P;     // P
    unsigned* hist = lds_u + rows * P;               // 256
    unsigned* bcast = hist + 256;                    // 2
    double* red = reinterpret_cast<double*>(bcast + 2 + ((rows * P + 258) & 1));  // 16 doubles, 8B aligned
    const int n = blockIdx.x;
    const size_t np = (size_t)N * P;
    const float* l_loc = terms + (size_t)n * P;
    const float* l_pos = terms + np + (size_t)n * P;
    const float* l_neg = terms + 2 * np + (size_t)n * P;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        vpos[i] = canon_bits(l_pos[i]);
        if (!single) vneg[i] = canon_bits(l_neg[i]);
    }
    // SSD._split_pos_neg, ssd.py:300-311 (integer exact)
    int pos = cnt[n], neg = P - pos;
    int pos_k, neg_k;
    if (pos * 3 > neg) { pos_k = neg / 3; neg_k = neg; } else { pos_k = pos; neg_k = pos * 3; }
    __syncthreads();
    // SSD._k_plus_1_th_value, ssd.py:313-328: kthvalue(P-k) ascending == (k+1)-th largest; k == 0 -> max == 1st largest
    unsigned tp = radix_select_desc(vpos, P, pos_k + 1, hist, bcast);
    if (single) {
        __syncthreads();
        for (int i = threadIdx.x; i < P; i += blockDim.x) vneg[i] = canon_bits(l_neg[i]);
        __syncthreads();
    }
    unsigned tn = radix_select_desc(vneg, P, neg_k + 1, hist, bcast);
    float thr_pos = __uint_as_float(tp), thr_neg = __uint_as_float(tn);
    // ssd.py:226-227 masked sum for this image
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        float lp = __uint_as_float(canon_bits(l_pos[i])), ln = __uint_as_float(vneg[i]);
        float v = 0.0f;
        if (lp > thr_pos) v += alpha * l_loc[i] + lp;
        if (ln > thr_neg) v += ln;
        acc += (double)v;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
        float inv = pos_k > 0 ? 1.0f / (float)pos_k : 0.0f;
        float* o = sel + (size_t)n * 8;
        o[0] = thr_pos; o[1] = thr_neg; o[2] = (float)pos_k; o[3] = (float)neg_k;
        o[4] = inv; o[5] = (float)s * inv; o[6] = (float)pos; o[7] = 0.0f;
    }
}

__global__ void multibox_mean_kernel(const float* __restrict__ sel, int N, float* __restrict__ loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 64) s += (double)sel[(size_t)i * 8 + 5];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)N);
}

template <int NC>
__global__ __launch_bounds__(256) void multibox_bwd_kernel(const float* __restrict__ outputs, const float* __restrict__ targets,
                                                           const float* __restrict__ dboxes, int N, int P, int G, float alpha,
                                                           float threshold, const float* __restrict__ terms,
                                                           const float* __restrict__ sel, const float* __restrict__ grad_out,
                                                           float* __restrict__ grad) {
    extern __shared__ float gt[];
    constexpr int ROW = 4 + NC;
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < G * ROW; i += blockDim.x) gt[i] = targets[(size_t)n * G * ROW + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t np = (size_t)N * P, idx = (size_t)n * P + p;
    const float* s = sel + (size_t)n * 8;
    const float coef = s[4] * grad_out[0] / (float)N;
    const bool pos_valid = terms[np + idx] > s[0];
    const bool neg_valid = terms[2 * np + idx] > s[1];
    const float4 d = reinterpret_cast<const float4*>(dboxes)[p];
    const float* row = outputs + idx * ROW;
    float o0 = row[0], o1 = row[1], o2 = row[2], o3 = row[3];
    float sm[NC];
    row_log_softmax<NC>(row + 4, sm);
#pragma unroll
    for (int c = 0; c < NC; ++c) sm[c] = expf(sm[c]);
    float gl0 = 0.f, gl1 = 0.f, gl2 = 0.f, gl3 = 0.f;
    float gc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) gc[c] = 0.0f;
    bool any = false;
    for (int g = 0; g < G; ++g) {
        const float* t = gt + g * ROW;
        float g_cx = t[0], g_cy = t[1], g_w = t[2], g_h = t[3];
        if (!match_one(g_cx, g_cy, g_w, g_h, d.x, d.y, d.z, d.w, threshold)) continue;
        any = true;
        if (!pos_valid) continue;
        float h0 = (g_cx - d.x) / d.z;
        float h1 = (g_cy - d.y) / d.w;
        float h2 = g_w > 0.0f ? logf(g_w / d.z) : g_w;
        float h3 = g_h > 0.0f ? logf(g_h / d.w) : g_h;
        gl0 += smooth_l1_grad(o0 - h0);
        gl1 += smooth_l1_grad(o1 - h1);
        gl2 += smooth_l1_grad(o2 - h2);
        gl3 += smooth_l1_grad(o3 - h3);
        float tsum = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) tsum += t[4 + c];
#pragma unroll
        for (int c = 0; c < NC; ++c) gc[c] += sm[c] * tsum - t[4 + c];
    }
    if (!any && neg_valid) {
#pragma unroll
        for (int c = 0; c < NC; ++c) gc[c] += sm[c] - (c == 0 ? 1.0f : 0.0f);
    }
    float* go = grad + idx * ROW;
    const float ca = coef * alpha;
    go[0] = gl0 * ca; go[1] = gl1 * ca; go[2] = gl2 * ca; go[3] = gl3 * ca;
#pragma unroll
    for (int c = 0; c < NC; ++c) go[4 + c] = gc[c] * coef;
}

// ---- any number of classes (SSD(num_classes) is parametric in the reference: ssd.py:9,70-77) --------------------------------
// Same arithmetic, in the same order, as the C = 21 instantiations above (so the two agree bit for bit at C = 21); the per-class
// values are recomputed from the logits instead of being held in registers, because the register arrays need a compile-time size.
struct RowLsm {                  // log_softmax of one row, evaluated per class: (x[c] - max) - log(sum(exp(x - max)))
    const float* x; float mx, ls;
    __device__ __forceinline__ RowLsm(const float* row, int nc) : x(row) {
        mx = row[0];
        for (int c = 0; c < nc; ++c) mx = fmaxf(mx, row[c]);
        float s = 0.0f;
        for (int c = 0; c < nc; ++c) s += expf(row[c] - mx);
        ls = logf(s);
    }
    __device__ __forceinline__ float at(int c) const { return (x[c] - mx) - ls; }
};

__global__ __launch_bounds__(256) void multibox_terms_rt_kernel(const float* __restrict__ outputs, const float* __restrict__ targets,
                                                                const float* __restrict__ dboxes, int P, int G, int NC, float alpha,
                                                                float threshold, float* __restrict__ terms, int* __restrict__ cnt, int N) {
    extern __shared__ float gt[];  // (G, 4+NC)
    const int ROW = 4 + NC;
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < G * ROW; i += blockDim.x) gt[i] = targets[(size_t)n * G * ROW + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool any = false;
    if (p < P) {
        const float4 d = reinterpret_cast<const float4*>(dboxes)[p];
        const float* row = outputs + ((size_t)n * P + p) * ROW;
        float o0 = row[0], o1 = row[1], o2 = row[2], o3 = row[3];
        const RowLsm lsm(row + 4, NC);
        float l_loc = 0.0f, l_pos = 0.0f;
        for (int g = 0; g < G; ++g) {
            const float* t = gt + g * ROW;
            float g_cx = t[0], g_cy = t[1], g_w = t[2], g_h = t[3];
            if (!match_one(g_cx, g_cy, g_w, g_h, d.x, d.y, d.z, d.w, threshold)) continue;
            any = true;
            float h0 = (g_cx - d.x) / d.z;
            float h1 = (g_cy - d.y) / d.w;
            float h2 = g_w > 0.0f ? logf(g_w / d.z) : g_w;
            float h3 = g_h > 0.0f ? logf(g_h / d.w) : g_h;
            float s4 = ((smooth_l1(o0 - h0) + smooth_l1(o1 - h1)) + smooth_l1(o2 - h2)) + smooth_l1(o3 - h3);
            l_loc += s4;
            float ce = 0.0f;
            for (int c = 0; c < NC; ++c) ce += t[4 + c] * lsm.at(c);
            l_pos += -ce;
        }
        float l_neg = any ? 0.0f : -lsm.at(0);
        size_t np = (size_t)N * P;
        size_t idx = (size_t)n * P + p;
        terms[idx] = l_loc;
        terms[np + idx] = l_pos;
        terms[2 * np + idx] = l_neg;
    }
    unsigned long long b = __ballot(any);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt[n], (int)__popcll(b));
}

__global__ __launch_bounds__(256) void multibox_bwd_rt_kernel(const float* __restrict__ outputs, const float* __restrict__ targets,
                                                              const float* __restrict__ dboxes, int N, int P, int G, int NC, float alpha,
                                                              float threshold, const float* __restrict__ terms,
                                                              const float* __restrict__ sel, const float* __restrict__ grad_out,
                                                              float* __restrict__ grad) {
    extern __shared__ float gt[];   // (G, 4+NC) rows, then G class-sums
    const int ROW = 4 + NC;
    float* tsum = gt + G * ROW;
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < G * ROW; i += blockDim.x) gt[i] = targets[(size_t)n * G * ROW + i];
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float v = 0.0f;
        for (int c = 0; c < NC; ++c) v += gt[g * ROW + 4 + c];
        tsum[g] = v;
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t np = (size_t)N * P, idx = (size_t)n * P + p;
    const float* s = sel + (size_t)n * 8;
    const float coef = s[4] * grad_out[0] / (float)N;
    const bool pos_valid = terms[np + idx] > s[0];
    const bool neg_valid = terms[2 * np + idx] > s[1];
    const float4 d = reinterpret_cast<const float4*>(dboxes)[p];
    const float* row = outputs + idx * ROW;
    float o0 = row[0], o1 = row[1], o2 = row[2], o3 = row[3];
    const RowLsm lsm(row + 4, NC);
    float gl0 = 0.f, gl1 = 0.f, gl2 = 0.f, gl3 = 0.f;
    bool any = false;
    for (int g = 0; g < G; ++g) {
        const float* t = gt + g * ROW;
        float g_cx = t[0], g_cy = t[1], g_w = t[2], g_h = t[3];
        if (!match_one(g_cx, g_cy, g_w, g_h, d.x, d.y, d.z, d.w, threshold)) continue;
        any = true;
        if (!pos_valid) continue;
        float h0 = (g_cx - d.x) / d.z;
        float h1 = (g_cy - d.y) / d.w;
        float h2 = g_w > 0.0f ? logf(g_w / d.z) : g_w;
        float h3 = g_h > 0.0f ? logf(g_h / d.w) : g_h;
        gl0 += smooth_l1_grad(o0 - h0);
        gl1 += smooth_l1_grad(o1 - h1);
        gl2 += smooth_l1_grad(o2 - h2);
        gl3 += smooth_l1_grad(o3 - h3);
    }
    float* go = grad + idx * ROW;
    const float ca = coef * alpha;
    go[0] = gl0 * ca; go[1] = gl1 * ca; go[2] = gl2 * ca; go[3] = gl3 * ca;
    for (int c = 0; c < NC; ++c) {
        const float smc = expf(lsm.at(c));
        float gc = 0.0f;
        if (any && pos_valid) {
            for (int g = 0; g < G; ++g) {
                const float* t = gt + g * ROW;
                if (match_one(t[0], t[1], t[2], t[3], d.x, d.y, d.z, d.w, threshold)) gc += smc * tsum[g] - t[4 + c];
            }
        }
        if (!any && neg_valid) gc += smc - (c == 0 ? 1.0f : 0.0f);
        go[4 + c] = gc * coef;
    }
}

__global__ void match_mask_kernel(const float* __restrict__ targets, const float* __restrict__ dboxes, int P, int G, int row,
                                  float threshold, uint8_t* __restrict__ mask) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float4 d = reinterpret_cast<const float4*>(dboxes)[p];
    for (int g = 0; g < G; ++g) {
        const float* t = targets + ((size_t)n * G + g) * row;
        mask[((size_t)n * P + p) * G + g] = match_one(t[0], t[1], t[2], t[3], d.x, d.y, d.z, d.w, threshold) ? 1 : 0;
    }
}

}  // namespace

extern "C" {

int ssdhip_default_boxes(float* out, const int32_t* maps_host, int n_maps, double s_min, double s_max, void* stream) {
    SSDHIP_REQUIRE(out && maps_host && n_maps >= 2 && n_maps <= 8, "default_boxes: 2..8 maps");
    BoxMaps maps;
    maps.count = n_maps;
    int total = 0;
    for (int k = 0; k < n_maps; ++k) {
        maps.m[k] = maps_host[3 * k];
        maps.n[k] = maps_host[3 * k + 1];
        maps.a[k] = maps_host[3 * k + 2];
        SSDHIP_REQUIRE(maps.a[k] == 4 || maps.a[k] == 6, "default_boxes: anchors per cell must be 4 or 6");
        maps.off[k] = total;
        total += maps.m[k] * maps.n[k] * maps.a[k];
    }
    maps.off[n_maps] = total;
    hipLaunchKernelGGL(default_boxes_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, maps, s_min, s_max, total);
    SSDHIP_CHECK_LAUNCH();
    return total;
}

int ssdhip_multibox_loss_fwd(const float* outputs, const float* targets, const float* dboxes, int N, int P, int G, int C,
                             float alpha, float threshold, float* terms, int32_t* cnt, float* sel, float* loss, void* stream) {
    SSDHIP_REQUIRE(outputs && targets && dboxes && terms && cnt && sel && loss, "multibox_loss_fwd: null pointer");
    SSDHIP_REQUIRE(N > 0 && P > 0 && P <= 32768 && G >= 0, "multibox_loss_fwd: need 0 < P <= 32768");
    SSDHIP_REQUIRE(C >= 2 && C <= 1024, "multibox_loss_fwd: 2 <= C <= 1024 (void + at least one class)");
    SSDHIP_REQUIRE(N <= 65535, "multibox_loss_fwd: N <= 65535");
    SSDHIP_REQUIRE((size_t)G * (4 + C) * 4 <= 64 * 1024, "multibox_loss_fwd: G too large for LDS staging");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(cnt, 0, sizeof(int32_t) * N, st) != hipSuccess) { ssdhip_set_error("memset failed"); return SSDHIP_ELAUNCH; }
    size_t lds = (size_t)(G > 0 ? G : 1) * (4 + C) * sizeof(float);
    if (C == 21)       // VOC (train.py:77): per-class values in registers
        hipLaunchKernelGGL(multibox_terms_kernel<21>, dim3(cdiv(P, 256), N), dim3(256), lds, st, outputs, targets, dboxes, P, G, alpha,
                           threshold, terms, cnt, N);
    else
        hipLaunchKernelGGL(multibox_terms_rt_kernel, dim3(cdiv(P, 256), N), dim3(256), lds, st, outputs, targets, dboxes, P, G, C, alpha,
                           threshold, terms, cnt, N);
    SSDHIP_CHECK_LAUNCH();
    const int single = P > 16384;                       // two staged rows fit the 160 KiB LDS up to P = 16,384
    size_t lds2 = ((size_t)(single ? 1 : 2) * P + 258 + 2) * 4 + 16 * 8 + 8;
    static DeviceOnce lds_attr;                        // the attribute belongs to the kernel's per-device function object
    if (const int dev = lds_attr.pending(); dev != DeviceOnce::kDone) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(multibox_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess) { ssdhip_set_error("multibox_loss_fwd: cannot raise the dynamic LDS limit"); return SSDHIP_ELAUNCH; }
        lds_attr.mark(dev);
    }
    hipLaunchKernelGGL(multibox_select_kernel, dim3(N), dim3(1024), lds2, st, terms, cnt, N, P, alpha, sel, single);
    SSDHIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(multibox_mean_kernel, dim3(1), dim3(64), 0, st, sel, N, loss);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_multibox_loss_bwd(const float* outputs, const float* targets, const float* dboxes, int N, int P, int G, int C,
                             float alpha, float threshold, const float* terms, const float* sel, const float* grad_out,
                             float* grad, void* stream) {
    SSDHIP_REQUIRE(outputs && targets && dboxes && terms && sel && grad_out && grad, "multibox_loss_bwd: null pointer");
    SSDHIP_REQUIRE(N > 0 && N <= 65535 && P > 0 && C >= 2 && C <= 1024, "multibox_loss_bwd: need 0 < N <= 65535, P > 0, 2 <= C <= 1024");
    SSDHIP_REQUIRE((size_t)G * (4 + C) * 4 <= 64 * 1024, "multibox_loss_bwd: G too large for LDS staging");
    size_t lds = (size_t)(G > 0 ? G : 1) * (4 + C) * sizeof(float);
    if (C == 21)
        hipLaunchKernelGGL(multibox_bwd_kernel<21>, dim3(cdiv(P, 256), N), dim3(256), lds, (hipStream_t)stream, outputs, targets, dboxes,
                           N, P, G, alpha, threshold, terms, sel, grad_out, grad);
    else
        hipLaunchKernelGGL(multibox_bwd_rt_kernel, dim3(cdiv(P, 256), N), dim3(256), lds + (size_t)(G > 0 ? G : 1) * sizeof(float),
                           (hipStream_t)stream, outputs, targets, dboxes, N, P, G, C, alpha, threshold, terms, sel, grad_out, grad);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

int ssdhip_match_mask(const float* targets, const float* dboxes, int N, int P, int G, int row, float threshold, uint8_t* mask,
                      void* stream) {
    SSDHIP_REQUIRE(targets && dboxes && mask && N > 0 && P > 0 && G > 0 && row >= 4, "match_mask: bad arguments");
    hipLaunchKernelGGL(match_mask_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, (hipStream_t)stream, targets, dboxes, P, G, row,
                       threshold, mask);
    SSDHIP_CHECK_LAUNCH();
    return SSDHIP_OK;
}

}  // extern "C"
